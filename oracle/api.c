/* TEST INFRASTRUCTURE (see oracle.h).  Flat-array entry points for ctypes (oracle/pyoracle.py).
 * Encoding at this boundary: field element = 4 x u64 LE limbs, canonical; G1 affine = 8 u64 (x, y); G2 affine = 16 u64
 * (x.c0, x.c1, y.c0, y.c1); the point at infinity is all-zero words ((0,0) is on neither curve). */
#include "oracle.h"
#include "constants.h"
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
static const fctx *ctx(int f) { return f ? &FQ : &FR; }
static void g1_in(g1 *o, const u64 *p) { int z = 1; for (int i = 0; i < 8; i++) if (p[i]) z = 0; if (z) { g1_zero(o); return; } fe_from_canon(&o->X, p, &FQ); fe_from_canon(&o->Y, p + 4, &FQ); fe_one(&o->Z, &FQ); }
static void g1_out(u64 *p, const g1 *a) { g1 t = *a; if (g1_is_zero(&t)) { memset(p, 0, 64); return; } g1_to_affine(&t); fe_to_canon(p, &t.X, &FQ); fe_to_canon(p + 4, &t.Y, &FQ); }
static void fe2_in(fe2 *o, const u64 *p) { fe_from_canon(&o->c0, p, &FQ); fe_from_canon(&o->c1, p + 4, &FQ); }
static void fe2_out(u64 *p, const fe2 *a) { fe_to_canon(p, &a->c0, &FQ); fe_to_canon(p + 4, &a->c1, &FQ); }
static void g2_in(g2 *o, const u64 *p) { int z = 1; for (int i = 0; i < 16; i++) if (p[i]) z = 0; if (z) { g2_zero(o); return; } fe2_in(&o->X, p); fe2_in(&o->Y, p + 8); fe2_one(&o->Z); }
static void g2_out(u64 *p, const g2 *a) { g2 t = *a; if (g2_is_zero(&t)) { memset(p, 0, 128); return; } g2_to_affine(&t); fe2_out(p, &t.X); fe2_out(p + 8, &t.Y); }

/* op: 0 mul 1 add 2 sub 3 inv(a) 4 sqr(a) 5 neg(a) 6 sqrt(a) (Fq only; returns 0 words if non-residue) */
void o_field_op(int field, int op, const u64 *a, const u64 *b, u64 *out, size_t n) { const fctx *F = ctx(field);
  for (size_t i = 0; i < n; i++) { fe x, y, r; fe_from_canon(&x, a + 4 * i, F); if (b) fe_from_canon(&y, b + 4 * i, F); else fe_zero(&y);
    switch (op) { case 0: fe_mul(&r, &x, &y, F); break; case 1: fe_add(&r, &x, &y, F); break; case 2: fe_sub(&r, &x, &y, F); break; case 3: fe_inv(&r, &x, F); break;
      case 4: fe_sqr(&r, &x, F); break; case 5: fe_neg(&r, &x, F); break; default: if (!fq_sqrt(&r, &x)) fe_zero(&r); }
    fe_to_canon(out + 4 * i, &r, F); } }
void o_fe_to_mont(int field, const u64 *a, u64 *out, size_t n) { for (size_t i = 0; i < n; i++) { fe x; fe_from_canon(&x, a + 4 * i, ctx(field)); memcpy(out + 4 * i, x.l, 32); } }
void o_fe_from_mont(int field, const u64 *a, u64 *out, size_t n) { for (size_t i = 0; i < n; i++) { fe x; memcpy(x.l, a + 4 * i, 32); fe_to_canon(out + 4 * i, &x, ctx(field)); } }
/* op: 0 mul 1 sqr 2 inv 3 sqrt 4 frobenius(1) */
int o_fq2_op(int op, const u64 *a, const u64 *b, u64 *out) { fe2 x, y, r; fe2_in(&x, a); if (b) fe2_in(&y, b); int ok = 1;
  switch (op) { case 0: fe2_mul(&r, &x, &y); break; case 1: fe2_sqr(&r, &x); break; case 2: fe2_inv(&r, &x); break; case 3: ok = fe2_sqrt(&r, &x); break; default: fe2_frob(&r, &x, 1); }
  if (ok) fe2_out(out, &r); return ok; }
/* op: 0 add 1 dbl(a) 2 madd 3 scalar mul (k) 4 neg */
void o_g1_op(int op, const u64 *a, const u64 *b, const u64 *k, u64 *out) { g1 x, y, r; g1_in(&x, a); if (b) g1_in(&y, b);
  switch (op) { case 0: g1_add(&r, &x, &y); break; case 1: g1_dbl(&r, &x); break; case 2: g1_madd(&r, &x, &y); break; case 3: g1_mul(&r, &x, k); break; default: g1_neg(&r, &x); } g1_out(out, &r); }
void o_g2_op(int op, const u64 *a, const u64 *b, const u64 *k, u64 *out) { g2 x, y, r; g2_in(&x, a); if (b) g2_in(&y, b);
  switch (op) { case 0: g2_add(&r, &x, &y); break; case 1: g2_dbl(&r, &x); break; case 2: g2_madd(&r, &x, &y); break; case 3: g2_mul(&r, &x, k); break; default: g2_neg(&r, &x); } g2_out(out, &r); }
void o_g1_gen(u64 *out) { g1 g; g1_gen(&g); g1_out(out, &g); }
void o_g2_gen(u64 *out) { g2 g; g2_gen(&g); g2_out(out, &g); }
int o_g1_on_curve(const u64 *a) { g1 x; g1_in(&x, a); return g1_on_curve(&x); }
int o_g2_on_curve(const u64 *a) { g2 x; g2_in(&x, a); return g2_on_curve(&x); }
/* consecutive multiples: out[i] = (k0 + i)·G  (cheap test bases; the harness builds its MSM bases the same way) */
void o_g1_consecutive(const u64 *k0, size_t n, u64 *out) { g1 g, p; g1_gen(&g); g1_mul(&p, &g, k0); for (size_t i = 0; i < n; i++) { g1_out(out + 8 * i, &p); g1_add(&p, &p, &g); } }
void o_g2_consecutive(const u64 *k0, size_t n, u64 *out) { g2 g, p; g2_gen(&g); g2_mul(&p, &g, k0); for (size_t i = 0; i < n; i++) { g2_out(out + 16 * i, &p); g2_add(&p, &p, &g); } }

/* mode 0: BDLO12 on all pairs; mode 1: multi_exp_with_mixed_addition */
void o_msm_g1(const u64 *pts, const u64 *k, size_t n, int mode, u64 *out) { g1 *b = malloc(sizeof(g1) * (n + 1)); for (size_t i = 0; i < n; i++) g1_in(&b[i], pts + 8 * i);
  g1 r; if (mode) msm_g1_mixed(&r, b, (const u64 (*)[4])k, n); else msm_g1_bdlo12(&r, b, (const u64 (*)[4])k, n); g1_out(out, &r); free(b); }
void o_msm_g2(const u64 *pts, const u64 *k, size_t n, int mode, u64 *out) { g2 *b = malloc(sizeof(g2) * (n + 1)); for (size_t i = 0; i < n; i++) g2_in(&b[i], pts + 16 * i);
  g2 r; if (mode) msm_g2_mixed(&r, b, (const u64 (*)[4])k, n); else msm_g2_bdlo12(&r, b, (const u64 (*)[4])k, n); g2_out(out, &r); free(b); }

size_t o_domain_size(size_t min_size) { domain d; return domain_init(&d, min_size) ? 0 : d.m; }
/* op: 0 FFT 1 iFFT 2 cosetFFT(g=5) 3 icosetFFT 4 divide_by_Z_on_coset 5 lagrange(t) -> out[m] 6 Z(t) -> out[1] 7 add_poly_Z(t) into zero H -> out[m+1] */
int o_domain_op(int op, size_t min_size, const u64 *in, const u64 *t, u64 *out) { domain d; if (domain_init(&d, min_size)) return -1; size_t m = d.m;
  fe g; memcpy(g.l, FR_MULT_GEN, 32); fe tt; if (t) fe_from_canon(&tt, t, &FR);
  size_t n = op == 6 ? 1 : op == 7 ? m + 1 : m; fe *a = calloc(n, sizeof(fe)); if (op <= 4) for (size_t i = 0; i < m; i++) fe_from_canon(&a[i], in + 4 * i, &FR);
  switch (op) { case 0: domain_fft(&d, a); break; case 1: domain_ifft(&d, a); break; case 2: domain_coset_fft(&d, a, &g); break; case 3: domain_icoset_fft(&d, a, &g); break;
    case 4: domain_divide_by_Z_on_coset(&d, a); break; case 5: domain_lagrange(&d, a, &tt); break; case 6: domain_vanishing(&d, a, &tt); break; default: domain_add_poly_Z(&d, &tt, a); }
  for (size_t i = 0; i < n; i++) fe_to_canon(out + 4 * i, &a[i], &FR); free(a); return 0; }

/* ---- R1CS (CSR with canonical coefficients) ---- */
static void r1cs_in(r1cs *cs, fe **owned, size_t n_inputs, size_t n_vars, size_t n_cons, const uint32_t *rp[3], const uint32_t *col[3], const u64 *co[3]) {
  cs->n_inputs = n_inputs; cs->n_vars = n_vars; cs->n_cons = n_cons;
  for (int m = 0; m < 3; m++) { size_t nnz = rp[m][n_cons]; cs->nnz[m] = nnz; cs->rowptr[m] = rp[m]; cs->col[m] = col[m]; owned[m] = malloc(sizeof(fe) * (nnz + 1));
    for (size_t k = 0; k < nnz; k++) fe_from_canon(&owned[m][k], co[m] + 4 * k, &FR); cs->coeff[m] = owned[m]; } }
#define R1CS_ARGS size_t n_inputs, size_t n_vars, size_t n_cons, const uint32_t *rpA, const uint32_t *colA, const u64 *coA, const uint32_t *rpB, const uint32_t *colB, const u64 *coB, const uint32_t *rpC, const uint32_t *colC, const u64 *coC
#define R1CS_LOAD r1cs cs; fe *own[3]; { const uint32_t *rp[3] = {rpA, rpB, rpC}, *cl[3] = {colA, colB, colC}; const u64 *co[3] = {coA, coB, coC}; r1cs_in(&cs, own, n_inputs, n_vars, n_cons, rp, cl, co); }
#define R1CS_FREE free(own[0]); free(own[1]); free(own[2]);
static fe *z_in(const u64 *z, size_t n) { fe *w = malloc(sizeof(fe) * (n + 1)); for (size_t i = 0; i < n; i++) fe_from_canon(&w[i], z + 4 * i, &FR); return w; }

int o_r1cs_is_satisfied(R1CS_ARGS, const u64 *z) { R1CS_LOAD fe *w = z_in(z, n_vars); int ok = r1cs_is_satisfied(&cs, w); free(w); R1CS_FREE return ok; }
/* H_out: (m+1) x 4 words */
size_t o_witness_map(R1CS_ARGS, const u64 *z, u64 *H_out) { R1CS_LOAD fe *w = z_in(z, n_vars); domain d; domain_init(&d, n_cons + n_inputs + 1);
  fe *H = malloc(sizeof(fe) * (d.m + 1)); r1cs_witness_map(&cs, w, H, &d); for (size_t i = 0; i <= d.m; i++) fe_to_canon(H_out + 4 * i, &H[i], &FR); free(H); free(w); R1CS_FREE return d.m; }

/* proving key as flat arrays: head = alpha_g1(8) beta_g1(8) delta_g1(8) beta_g2(16) delta_g2(16) */
static void pk_in(proving_key *pk, const u64 *head, const u64 *A, size_t nA, const uint32_t *Bidx, const u64 *B2, const u64 *B1, size_t nB, const u64 *H, size_t nH, const u64 *L, size_t nL) {
  g1_in(&pk->alpha_g1, head); g1_in(&pk->beta_g1, head + 8); g1_in(&pk->delta_g1, head + 16); g2_in(&pk->beta_g2, head + 24); g2_in(&pk->delta_g2, head + 40);
  pk->nA = nA; pk->A = malloc(sizeof(g1) * (nA + 1)); for (size_t i = 0; i < nA; i++) g1_in(&pk->A[i], A + 8 * i);
  pk->nB = nB; pk->B_idx = (uint32_t *)Bidx; pk->B_g2 = malloc(sizeof(g2) * (nB + 1)); pk->B_g1 = malloc(sizeof(g1) * (nB + 1)); for (size_t i = 0; i < nB; i++) { g2_in(&pk->B_g2[i], B2 + 16 * i); g1_in(&pk->B_g1[i], B1 + 8 * i); }
  pk->nH = nH; pk->H = malloc(sizeof(g1) * (nH + 1)); for (size_t i = 0; i < nH; i++) g1_in(&pk->H[i], H + 8 * i);
  pk->nL = nL; pk->L = malloc(sizeof(g1) * (nL + 1)); for (size_t i = 0; i < nL; i++) g1_in(&pk->L[i], L + 8 * i); }
static void pk_free(proving_key *pk) { free(pk->A); free(pk->B_g2); free(pk->B_g1); free(pk->H); free(pk->L); }
/* proof_out: A(8) B(16) C(8) */
void o_prove(R1CS_ARGS, const u64 *z, const u64 *head, const u64 *A, size_t nA, const uint32_t *Bidx, const u64 *B2, const u64 *B1, size_t nB, const u64 *H, size_t nH, const u64 *L, size_t nL,
             const u64 *r, const u64 *s, u64 *proof_out) { R1CS_LOAD fe *w = z_in(z, n_vars); proving_key pk; pk_in(&pk, head, A, nA, Bidx, B2, B1, nB, H, nH, L, nL);
  fe rr, ss; fe_from_canon(&rr, r, &FR); fe_from_canon(&ss, s, &FR); proof p; groth16_prove(&p, &pk, &cs, w, &rr, &ss);
  g1_out(proof_out, &p.A); g2_out(proof_out + 8, &p.B); g1_out(proof_out + 24, &p.C); pk_free(&pk); free(w); R1CS_FREE }
/* toxic: t alpha beta gamma delta g1k g2k (7 x 4 words).  Outputs sized by the caller: A (n_vars+1), Bidx/B2/B1 (n_vars+1 capacity), H (m-1), L (n_vars-n_inputs), IC (n_inputs+1), gt (48 words) */
size_t o_setup(R1CS_ARGS, const u64 *toxic, u64 *head, u64 *A, uint32_t *Bidx, u64 *B2, u64 *B1, u64 *H, u64 *L, u64 *vk_gt, u64 *vk_gamma_g2, u64 *IC) {
  R1CS_LOAD fe tx[7]; for (int i = 0; i < 7; i++) fe_from_canon(&tx[i], toxic + 4 * i, &FR); proving_key pk; verifying_key vk;
  groth16_setup(&pk, &vk, &cs, &tx[0], &tx[1], &tx[2], &tx[3], &tx[4], &tx[5], &tx[6]);
  g1_out(head, &pk.alpha_g1); g1_out(head + 8, &pk.beta_g1); g1_out(head + 16, &pk.delta_g1); g2_out(head + 24, &pk.beta_g2); g2_out(head + 40, &pk.delta_g2);
  for (size_t i = 0; i < pk.nA; i++) g1_out(A + 8 * i, &pk.A[i]); for (size_t i = 0; i < pk.nB; i++) { Bidx[i] = pk.B_idx[i]; g2_out(B2 + 16 * i, &pk.B_g2[i]); g1_out(B1 + 8 * i, &pk.B_g1[i]); }
  for (size_t i = 0; i < pk.nH; i++) g1_out(H + 8 * i, &pk.H[i]); for (size_t i = 0; i < pk.nL; i++) g1_out(L + 8 * i, &pk.L[i]);
  for (size_t i = 0; i < vk.nIC; i++) g1_out(IC + 8 * i, &vk.IC[i]); g2_out(vk_gamma_g2, &vk.gamma_g2);
  const fe *gt = (const fe *)&vk.alpha_g1_beta_g2; for (int i = 0; i < 12; i++) fe_to_canon(vk_gt + 4 * i, &gt[i], &FQ);
  size_t nB = pk.nB; free(pk.B_idx); pk_free(&pk); free(vk.IC); R1CS_FREE return nB; }
/* gt: 12 Fq coefficients in the order c0.c0.c0 c0.c0.c1 c0.c1.c0 ... c1.c2.c1 */
int o_verify(const u64 *vk_gt, const u64 *gamma_g2, const u64 *delta_g2, const u64 *IC, size_t nIC, const u64 *inputs, size_t n_inputs, const u64 *proof_in) {
  verifying_key vk; fe *gt = (fe *)&vk.alpha_g1_beta_g2; for (int i = 0; i < 12; i++) fe_from_canon(&gt[i], vk_gt + 4 * i, &FQ);
  g2_in(&vk.gamma_g2, gamma_g2); g2_in(&vk.delta_g2, delta_g2); vk.nIC = nIC; vk.IC = malloc(sizeof(g1) * (nIC + 1)); for (size_t i = 0; i < nIC; i++) g1_in(&vk.IC[i], IC + 8 * i);
  fe *in = z_in(inputs, n_inputs); proof p; g1_in(&p.A, proof_in); g2_in(&p.B, proof_in + 8); g1_in(&p.C, proof_in + 24);
  int ok = groth16_verify(&vk, in, n_inputs, &p); free(in); free(vk.IC); return ok; }
void o_pairing(const u64 *P, const u64 *Q, u64 *out) { g1 p; g2 q; g1_in(&p, P); g2_in(&q, Q); fe12 r; pairing_reduced(&r, &p, &q); const fe *c = (const fe *)&r; for (int i = 0; i < 12; i++) fe_to_canon(out + 4 * i, &c[i], &FQ); }
