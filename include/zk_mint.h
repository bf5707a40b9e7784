/* Drop-in for libsnark-vnt/src/mint/mintcgo.hpp (copied by the reference into go-ethereum/zktx/mintcgo.hpp and bound by
 * `#cgo LDFLAGS: -lzk_mint ...` in go-ethereum/zktx/zktx.go:3-11). */
#ifndef ZK_MINT_H
#define ZK_MINT_H
#include "zk_common.h"
#ifdef __cplusplus
extern "C" {
#endif
/* replaces libsnark-vnt/src/mint/mintcgo.cpp:268-321; key file /usr/local/prfKey/mintpk.txt (ZK_PRFKEY_DIR overrides the directory) */
char *genMintproof(uint64_t value, uint64_t value_old, char *sn_old_string, char *r_old_string, char *sn_string, char *r_string,
                   char *cmtA_old_string, char *cmtA_string, uint64_t value_s, char *sk_string);
/* replaces libsnark-vnt/src/mint/mintcgo.cpp:323-417; key file /usr/local/prfKey/mintvk.txt */
bool verifyMintproof(char *data, char *cmtA_old_string, char *sn_old_string, char *cmtA_string, uint64_t value_s);
#ifdef __cplusplus
}
#endif
#endif
