/* C stand-in for go-ethereum/zktx/zktx.go (no Go toolchain in this image): includes the drop-in headers the way zktx.go
 * includes the copied *cgo.hpp files, links with the reference's cgo LDFLAGS (-lzk_mint -lzk_send -lzk_deposit -lzk_redeem
 * -lff -lsnark; zktx.go:4) and marshals arguments as zktx.go does: common.ToHex strings ("0x" + lowercase hex) and uint64.
 *   dropin_driver hashes            -> prints computePRF / genCMT / computeCRH / genCMTS / genRoot on fixed inputs
 *   dropin_driver send              -> genSendproof + verifySendproof on the reference's send fixture (needs keys in $ZK_PRFKEY_DIR and a GPU)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/zk_mint.h"
#include "../include/zk_send.h"
#include "../include/zk_deposit.h"
#include "../include/zk_redeem.h"

static char *tohex(const unsigned char *b, int n) { char *s = malloc(2 * n + 3); s[0] = '0'; s[1] = 'x'; for (int i = 0; i < n; i++) sprintf(s + 2 + 2 * i, "%02x", b[i]); return s; }
static char *with0x(const char *h) { char *s = malloc(strlen(h) + 3); strcpy(s, "0x"); strcat(s, h); return s; }   /* results come back without 0x, zktx.go re-encodes them */

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  unsigned char sk[32], r[32], pk[20] = {0x00, 0x11, 0x22, 0x33, 0x44, 0x55, 0x66, 0x77, 0x88, 0x99, 0xaa, 0xbb, 0xcc, 0xdd, 0xee, 0xff, 0x00, 0x11, 0x22, 0x33}, zero[32] = {0};
  for (int i = 0; i < 32; i++) { sk[i] = 1; r[i] = (unsigned char)i; }
  if (!strcmp(argv[1], "hashes")) {
    printf("computePRF %s\n", computePRF(tohex(sk, 32), tohex(r, 32)));
    printf("genCMT0 %s\n", genCMT(0, tohex(zero, 32), tohex(zero, 32)));
    printf("computeCRH %s\n", computeCRH(tohex(pk, 20), tohex(r, 32)));
    printf("genCMTS %s\n", genCMTS(77, tohex(pk, 20), tohex(r, 32), tohex(sk, 32)));
    printf("genRoot0 %s\n", genRoot("", 0));
    return 0;
  }
  if (!strcmp(argv[1], "send")) {   /* libsnark-vnt/src/send/main.cpp:123-142: value_old 22, value_s 8, value 14, sk 1, r_old 123456, r 12, pk_sender 456, pk_recv 123 */
    char *skh = "0x1", *r_old = "0x123456", *rr = "0x12", *pks = "0x456", *pkr = "0x123";
    char *sn_old = with0x(computePRF(skh, r_old)), *cmtA_old = with0x(genCMT(22, sn_old, r_old)), *r_s = with0x(computeCRH(pks, rr)), *sn = with0x(computePRF(skh, rr));
    char *cmtS = with0x(genCMTS(8, pkr, r_s, sn_old)), *cmtA = with0x(genCMT(14, sn, rr));
    char *proof = genSendproof(22, r_s, sn_old, r_old, cmtS, cmtA_old, 8, pkr, 14, sn, rr, cmtA, skh, pks);   /* argument order of zktx.go:426 */
    printf("proof_len %zu head %.10s\n", strlen(proof), proof);
    printf("verify %d\n", (int)verifySendproof(proof, cmtA_old, sn_old, cmtS, cmtA));
    printf("verify_wrong %d\n", (int)verifySendproof(proof, cmtA, sn_old, cmtS, cmtA_old));
    return 0;
  }
  return 2;
}
