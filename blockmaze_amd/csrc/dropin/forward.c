/* Thin drop-in library: defines the symbols of one of the reference's libzk_X.so and forwards them to libzkgpu.so, so that the unchanged
 * `#cgo LDFLAGS: -L/usr/local/lib -lzk_mint -lzk_send -lzk_deposit -lzk_redeem -lff -lsnark ...` line of go-ethereum/zktx/zktx.go:4 links.
 * Built four times with -DZK_LIB_MINT / _SEND / _DEPOSIT / _REDEEM (symbol sets: SURVEY.md §8b). */
#include <stdbool.h>
#include <stdint.h>
#define FWD(ret, name, params, args) extern ret zkgpu_abi_##name params; ret name params { return zkgpu_abi_##name args; }
FWD(char *, genCMT, (uint64_t v, char *a, char *b), (v, a, b))
FWD(char *, computePRF, (char *a, char *b), (a, b))
#if defined(ZK_LIB_SEND) || defined(ZK_LIB_DEPOSIT)
FWD(char *, genCMTS, (uint64_t v, char *a, char *b, char *c), (v, a, b, c))
#endif
#ifdef ZK_LIB_SEND
FWD(char *, computeCRH, (char *a, char *b), (a, b))
FWD(char *, genSendproof, (uint64_t a, char *b, char *c, char *d, char *e, char *f, uint64_t g, char *h, uint64_t i, char *j, char *k, char *l, char *m, char *n), (a, b, c, d, e, f, g, h, i, j, k, l, m, n))
FWD(bool, verifySendproof, (char *a, char *b, char *c, char *d, char *e), (a, b, c, d, e))
#endif
#ifdef ZK_LIB_MINT
FWD(char *, genMintproof, (uint64_t a, uint64_t b, char *c, char *d, char *e, char *f, char *g, char *h, uint64_t i, char *j), (a, b, c, d, e, f, g, h, i, j))
FWD(bool, verifyMintproof, (char *a, char *b, char *c, char *d, uint64_t e), (a, b, c, d, e))
#endif
#ifdef ZK_LIB_REDEEM
FWD(char *, genRedeemproof, (uint64_t a, uint64_t b, char *c, char *d, char *e, char *f, char *g, char *h, uint64_t i, char *j), (a, b, c, d, e, f, g, h, i, j))
FWD(bool, verifyRedeemproof, (char *a, char *b, char *c, char *d, uint64_t e), (a, b, c, d, e))
#endif
#ifdef ZK_LIB_DEPOSIT
FWD(char *, genRoot, (char *a, int n), (a, n))
FWD(char *, genDepositproof, (uint64_t a, uint64_t b, char *c, char *d, char *e, char *f, char *g, char *h, char *i, char *j, uint64_t k, char *l, char *m, char *n, char *o, int p, char *q, char *r), (a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p, q, r))
FWD(bool, verifyDepositproof, (char *a, char *b, char *c, char *d, char *e, char *f, char *g), (a, b, c, d, e, f, g))
#endif
