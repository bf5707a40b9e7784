/* Shared part of the drop-in C-ABI: the helper symbols that more than one of the reference's libzk_*.so exports.
 * Argument encoding (identical to the reference): every char* is a NUL-terminated "0x"-prefixed
 * lowercase hex string as produced by go-ethereum's common.ToHex (GETH/zktx/zktx.go:124-126); it is parsed like
 * uint256S / uint160S (libsnark-vnt/src/send/uint256.h:222-248).  Returned strings are heap buffers owned by the
 * caller (the Go side never frees them); hashes are 64 hex characters + NUL, proofs 512 hex characters + NUL.
 */
#ifndef ZK_COMMON_H
#define ZK_COMMON_H
#include <stdbool.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* replaces libsnark-vnt/src/mint/mintcgo.cpp:239-252 (same body in sendcgo.cpp:241-254, depositcgo.cpp:256-269, redeemcgo.cpp:240-253) */
char *genCMT(uint64_t value, char *sn_string, char *r_string);
/* replaces libsnark-vnt/src/send/sendcgo.cpp:274-286 (same body in all four libraries) */
char *computePRF(char *sk_string, char *r_string);
#ifdef __cplusplus
}
#endif
#endif
