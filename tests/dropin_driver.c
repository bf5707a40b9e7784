/* C stand-in for go-ethereum/zktx/zktx.go (no Go toolchain in this image): includes the drop-in headers the way zktx.go
 * includes the copied *cgo.hpp files, links with the reference's cgo LDFLAGS (-lzk_mint -lzk_send -lzk_deposit -lzk_redeem
 * -lff -lsnark; zktx.go:4) and marshals arguments as zktx.go does: common.ToHex strings ("0x" + lowercase hex) and uint64.
 *   dropin_driver hashes            -> prints computePRF / genCMT / computeCRH / genCMTS / genRoot on fixed inputs
 *   dropin_driver send              -> genSendproof + verifySendproof on the reference's send fixture (needs keys in $ZK_PRFKEY_DIR and a GPU)
 *   dropin_driver mint|redeem|deposit -> the same for the fixtures of mint/main.cpp:121-129, redeem/main.cpp:121-129, deposit/main.cpp:131-167 (genRoot with n = 16 on the way)
 *   dropin_driver roots             -> genRoot with n = 0, 1 and 16 (SURVEY.md §8c golden values)
 *   dropin_driver verifybench T N   -> one send proof, then T pthreads calling verifySendproof N times each (geth's goroutines: one proof a call); calls/s on stderr
 *   dropin_driver threads           -> 8 pthreads, each proving and verifying all four circuits twice through the thin libraries: cgo calls arrive on arbitrary OS threads (zktx.go:406-430)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>
#include "../include/zk_mint.h"
#include "../include/zk_send.h"
#include "../include/zk_deposit.h"
#include "../include/zk_redeem.h"

static char *tohex(const unsigned char *b, int n) { char *s = malloc(2 * n + 3); s[0] = '0'; s[1] = 'x'; for (int i = 0; i < n; i++) sprintf(s + 2 + 2 * i, "%02x", b[i]); return s; }
static char *with0x(const char *h) { char *s = malloc(strlen(h) + 3); strcpy(s, "0x"); strcat(s, h); return s; }   /* results come back without 0x, zktx.go re-encodes them */

static char *leaves16(const char *cmtS) {   /* deposit/main.cpp:154-167: hex values 1..9, cmtS, 0x11..0x16 as 66-character items */
  static const char *v[16] = {"1", "2", "3", "4", "5", "6", "7", "8", "9", 0, "11", "12", "13", "14", "15", "16"}; char *s = malloc(16 * 66 + 1); s[0] = 0;
  for (int i = 0; i < 16; i++) { char item[67]; if (v[i]) snprintf(item, sizeof item, "0x%64s", v[i]); else snprintf(item, sizeof item, "0x%s", cmtS); for (char *p = item + 2; *p == ' '; p++) *p = '0'; strcat(s, item); }
  return s; }
static int run_mint(int redeem) {   /* mint: value 13, old 6, s 7; redeem: value 13, old 20, s 7; sk 1, r_old 123456, r 123 */
  char *skh = "0x1", *r_old = "0x123456", *rr = "0x123"; unsigned long long v = 13, vo = redeem ? 20 : 6, vs = 7;
  char *sn_old = with0x(computePRF(skh, r_old)), *cmtA_old = with0x(genCMT(vo, sn_old, r_old)), *sn = with0x(computePRF(skh, rr)), *cmtA = with0x(genCMT(v, sn, rr));
  char *proof = redeem ? genRedeemproof(v, vo, sn_old, r_old, sn, rr, cmtA_old, cmtA, vs, skh) : genMintproof(v, vo, sn_old, r_old, sn, rr, cmtA_old, cmtA, vs, skh);
  int ok = redeem ? verifyRedeemproof(proof, cmtA_old, sn_old, cmtA, vs) : verifyMintproof(proof, cmtA_old, sn_old, cmtA, vs), bad = redeem ? verifyRedeemproof(proof, cmtA_old, sn_old, cmtA, vs + 1) : verifyMintproof(proof, cmtA_old, sn_old, cmtA, vs + 1);
  printf("%s proof_len %zu head %.10s verify %d verify_wrong %d\n", redeem ? "redeem" : "mint", strlen(proof), proof, ok, bad); return strlen(proof) == 512 && ok && !bad; }
static int run_deposit(void) {   /* value 264, old 255, s 9, r_s 123, sn_A_old 123, r 12, pk_recv 123; 16 leaves with cmtS at index 9 */
  char *skh = "0x1", *r_old = "0x123456", *rr = "0x12", *r_s = "0x123", *sn_A_old = "0x123", *pkr = "0x123";
  char *sn_old = with0x(computePRF(skh, r_old)), *cmtB_old = with0x(genCMT(255, sn_old, r_old)), *sn_s = with0x(computePRF(skh, r_s)), *cmtS = with0x(genCMTS(9, pkr, r_s, sn_A_old)), *sn = with0x(computePRF(skh, rr)), *cmtB = with0x(genCMT(264, sn, rr));
  char *arr = leaves16(cmtS + 2), *rt = with0x(genRoot(arr, 16));
  char *proof = genDepositproof(264, 255, sn_old, r_old, sn, rr, sn_s, r_s, cmtB_old, cmtB, 9, pkr, sn_A_old, cmtS, arr, 16, rt, skh);
  int ok = verifyDepositproof(proof, rt, pkr, cmtB_old, sn_old, cmtB, sn_s), bad = verifyDepositproof(proof, cmtB, pkr, cmtB_old, sn_old, cmtB, sn_s);
  printf("deposit root %s proof_len %zu head %.10s verify %d verify_wrong %d\n", rt + 2, strlen(proof), proof, ok, bad); return strlen(proof) == 512 && ok && !bad; }
static int run_send(void) {   /* libsnark-vnt/src/send/main.cpp:123-142: value_old 22, value_s 8, value 14, sk 1, r_old 123456, r 12, pk_sender 456, pk_recv 123 */
  char *skh = "0x1", *r_old = "0x123456", *rr = "0x12", *pks = "0x456", *pkr = "0x123";
  char *sn_old = with0x(computePRF(skh, r_old)), *cmtA_old = with0x(genCMT(22, sn_old, r_old)), *r_s = with0x(computeCRH(pks, rr)), *sn = with0x(computePRF(skh, rr));
  char *cmtS = with0x(genCMTS(8, pkr, r_s, sn_old)), *cmtA = with0x(genCMT(14, sn, rr));
  char *proof = genSendproof(22, r_s, sn_old, r_old, cmtS, cmtA_old, 8, pkr, 14, sn, rr, cmtA, skh, pks);   /* argument order of zktx.go:426 */
  int ok = verifySendproof(proof, cmtA_old, sn_old, cmtS, cmtA), bad = verifySendproof(proof, cmtA, sn_old, cmtS, cmtA_old);
  printf("proof_len %zu head %.10s\nverify %d\nverify_wrong %d\n", strlen(proof), proof, ok, bad); return strlen(proof) == 512 && ok && !bad; }
static void *thread_main(void *arg) { long good = 0; for (int rep = 0; rep < 2; rep++) good += run_mint(0) + run_send() + run_deposit() + run_mint(1); *(long *)arg = good; return 0; }

struct vb { char *proof, *a, *b, *c, *d; int n; long good; };
static void *vb_main(void *arg) { struct vb *v = arg; for (int i = 0; i < v->n; i++) v->good += verifySendproof(v->proof, v->a, v->b, v->c, v->d) ? 1 : 0; return 0; }
static int verify_bench(int T, int N) {
  char *skh = "0x1", *r_old = "0x123456", *rr = "0x12", *pks = "0x456", *pkr = "0x123";
  char *sn_old = with0x(computePRF(skh, r_old)), *cmtA_old = with0x(genCMT(22, sn_old, r_old)), *r_s = with0x(computeCRH(pks, rr)), *sn = with0x(computePRF(skh, rr));
  char *cmtS = with0x(genCMTS(8, pkr, r_s, sn_old)), *cmtA = with0x(genCMT(14, sn, rr));
  char *proof = genSendproof(22, r_s, sn_old, r_old, cmtS, cmtA_old, 8, pkr, 14, sn, rr, cmtA, skh, pks);
  for (int i = 0; i < 20; i++) if (!verifySendproof(proof, cmtA_old, sn_old, cmtS, cmtA)) return 0;
  if (verifySendproof(proof, cmtA, sn_old, cmtS, cmtA_old)) return 0;
  pthread_t th[64]; struct vb v[64]; if (T > 64) T = 64; struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i < T; i++) { v[i] = (struct vb){proof, cmtA_old, sn_old, cmtS, cmtA, N, 0}; pthread_create(&th[i], 0, vb_main, &v[i]); }
  long good = 0; for (int i = 0; i < T; i++) { pthread_join(th[i], 0); good += v[i].good; }
  clock_gettime(CLOCK_MONOTONIC, &t1); double dt = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  fprintf(stderr, "verifySendproof from %d thread(s), %d calls each: %.3f ms per call, %.0f verifications/s, %ld of %ld accepted\n", T, N, 1e3 * dt / N, T * (double)N / dt, good, (long)T * N);
  return good == (long)T * N; }

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  if (!strcmp(argv[1], "verifybench")) return verify_bench(argc > 2 ? atoi(argv[2]) : 8, argc > 3 ? atoi(argv[3]) : 500) ? 0 : 1;
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
  if (!strcmp(argv[1], "roots")) {   /* the host-only symbol of libzk_deposit.so: needs no key and no GPU */
    char one[67]; snprintf(one, sizeof one, "0x%064d", 1); char *cmtS = genCMTS(9, "0x123", "0x123", "0x123");
    printf("genRoot0 %s\ngenRoot1 %s\ngenRoot16 %s\n", genRoot("", 0), genRoot(one, 1), genRoot(leaves16(cmtS), 16)); return 0; }
  if (!strcmp(argv[1], "send")) return run_send() ? 0 : 1;
  if (!strcmp(argv[1], "mint")) return run_mint(0) ? 0 : 1;
  if (!strcmp(argv[1], "redeem")) return run_mint(1) ? 0 : 1;
  if (!strcmp(argv[1], "deposit")) return run_deposit() ? 0 : 1;
  if (!strcmp(argv[1], "threads")) { pthread_t th[8]; long good[8]; for (int i = 0; i < 8; i++) pthread_create(&th[i], 0, thread_main, &good[i]); long total = 0; for (int i = 0; i < 8; i++) { pthread_join(th[i], 0); total += good[i]; }
    printf("threads_good %ld of %d\n", total, 8 * 2 * 4); return total == 8 * 2 * 4 ? 0 : 1; }
  return 2;
}
