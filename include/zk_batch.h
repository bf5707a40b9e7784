/* Optional extension of the drop-in surface: batch verification.
 *
 * Not part of the reference: go-ethereum calls verifyMintproof / verifySendproof / verifyDepositproof / verifyRedeemproof once per transaction, twice per
 * transaction over its life (core/tx_pool.go:612-645 when it enters the pool, core/state_processor.go:106-163 when its block is processed), and the reference loads
 * the verification key from disk on every call (e.g. libsnark-vnt/src/send/sendcgo.cpp:380).  verifyBatch takes all ZK transactions of a block (or of a pool
 * sweep) in one cgo call; its decisions are exactly those of the four verify symbols (r1cs_gg_ppzksnark_verifier_strong_IC, r1cs_gg_ppzksnark.tcc:614-623, on the
 * statement packed as X_gadget::witness_map does).  Large groups run on the GPU (kernel K9, one lane per proof), small ones on the host.
 *
 * Exported by libzkgpu.so only (the four libzk_*.so keep exactly the reference's symbol sets): a caller that wants it adds -lzkgpu to its link line.
 */
#ifndef ZK_BATCH_H
#define ZK_BATCH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { ZK_KIND_MINT = 0, ZK_KIND_SEND = 1, ZK_KIND_DEPOSIT = 2, ZK_KIND_REDEEM = 3 };

/* One proof and its statement.  `proof`: the 512 hex characters gen*proof returned.  `args`: the "0x…" hex strings of the kind's verify symbol in its argument order —
 *   mint / redeem : cmtA_old, sn_old, cmtA                    (+ value_s)      mintcgo.hpp:24, redeemcgo.hpp:24
 *   send          : cmtA_old, sn_old, cmtS, cmtA_new                           sendcgo.hpp:30
 *   deposit       : RT, pk, cmtb_old, snold, cmtb, sns                         depositcgo.hpp:33
 * unused entries may be NULL. */
typedef struct { int kind; const char *proof; const char *args[6]; uint64_t value_s; } zk_verify_item;

/* ok[i] = 1 if item i is accepted, 0 otherwise (malformed records and unknown kinds are rejected).  Returns the number of accepted proofs, or -1 if no
 * decision could be made (key file missing, device failure): every ok[i] is 0 then. */
int verifyBatch(const zk_verify_item *items, int n, unsigned char *ok);

#ifdef __cplusplus
}
#endif
#endif
