/* TEST INFRASTRUCTURE (see oracle.h).  R1CS -> QAP witness map and the Groth16 generator / prover / verifier.
 * Follows SNARK/reductions/r1cs_to_qap/r1cs_to_qap.tcc:105-176 (instance map with evaluation), :206-334 (witness map),
 * SNARK/zk_proof_systems/ppzksnark/r1cs_gg_ppzksnark/r1cs_gg_ppzksnark.tcc:212-388 (generator), :391-506 (prover),
 * :509-623 (verifier), SNARK/relations/constraint_satisfaction_problems/r1cs/r1cs.tcc:129-166 (is_satisfied). */
#include "oracle.h"
#include "constants.h"
#include <stdlib.h>
#include <string.h>
#define R (&FR)

static void lc_eval(fe *o, const r1cs *cs, int m, size_t row, const fe *z) { /* z without ONE; index 0 = ONE */
  fe acc, one, t; fe_zero(&acc); fe_one(&one, R);
  for (uint32_t k = cs->rowptr[m][row]; k < cs->rowptr[m][row + 1]; k++) { uint32_t c = cs->col[m][k]; fe_mul(&t, &cs->coeff[m][k], c ? &z[c - 1] : &one, R); fe_add(&acc, &acc, &t, R); }
  *o = acc; }
int r1cs_is_satisfied(const r1cs *cs, const fe *z) { fe a, b, c, ab;
  for (size_t i = 0; i < cs->n_cons; i++) { lc_eval(&a, cs, 0, i, z); lc_eval(&b, cs, 1, i, z); lc_eval(&c, cs, 2, i, z); fe_mul(&ab, &a, &b, R); if (!fe_eq(&ab, &c)) return 0; } return 1; }

void r1cs_witness_map(const r1cs *cs, const fe *z, fe *H, const domain *d) { /* d1 = d2 = d3 = 0 as the prover calls it (:402) */
  size_t m = d->m, nc = cs->n_cons; fe *aA = calloc(m, sizeof(fe)), *aB = calloc(m, sizeof(fe)), *aC = calloc(m, sizeof(fe)); fe g; memcpy(g.l, FR_MULT_GEN, 32);
  for (size_t i = 0; i <= cs->n_inputs; i++) { if (i) aA[i + nc] = z[i - 1]; else fe_one(&aA[nc], R); }
  for (size_t i = 0; i < nc; i++) { lc_eval(&aA[i], cs, 0, i, z); lc_eval(&aB[i], cs, 1, i, z); }
  domain_ifft(d, aA); domain_ifft(d, aB);
  memset(H, 0, sizeof(fe) * (m + 1));                                   /* ZK patch with d's = 0 leaves H = 0 */
  domain_coset_fft(d, aA, &g); domain_coset_fft(d, aB, &g);
  for (size_t i = 0; i < m; i++) fe_mul(&aA[i], &aA[i], &aB[i], R);
  for (size_t i = 0; i < nc; i++) lc_eval(&aC[i], cs, 2, i, z);
  domain_ifft(d, aC); domain_coset_fft(d, aC, &g);
  for (size_t i = 0; i < m; i++) fe_sub(&aA[i], &aA[i], &aC[i], R);
  domain_divide_by_Z_on_coset(d, aA); domain_icoset_fft(d, aA, &g);
  for (size_t i = 0; i < m; i++) fe_add(&H[i], &H[i], &aA[i], R);
  free(aA); free(aB); free(aC); }

void groth16_prove(proof *o, const proving_key *pk, const r1cs *cs, const fe *z, const fe *r, const fe *s) {
  domain d; domain_init(&d, cs->n_cons + cs->n_inputs + 1); size_t m = d.m, nv = cs->n_vars, ni = cs->n_inputs;
  fe *H = malloc(sizeof(fe) * (m + 1)); r1cs_witness_map(cs, z, H, &d);
  uint64_t (*zc)[4] = malloc(32 * (nv + 1)); zc[0][0] = 1; zc[0][1] = zc[0][2] = zc[0][3] = 0; for (size_t i = 0; i < nv; i++) fe_to_canon(zc[i + 1], &z[i], R);
  uint64_t (*hc)[4] = malloc(32 * (m + 1)); for (size_t i = 0; i <= m; i++) fe_to_canon(hc[i], &H[i], R);
  g1 eA, eB1, eH, eL; g2 eB2;
  msm_g1_mixed(&eA, pk->A, (const uint64_t (*)[4])zc, nv + 1);                                   /* :442-449 */
  { uint64_t (*zb)[4] = malloc(32 * (pk->nB ? pk->nB : 1)); for (size_t j = 0; j < pk->nB; j++) memcpy(zb[j], zc[pk->B_idx[j]], 32);          /* kc_multiexp.tcc:21-85 */
    msm_g2_mixed(&eB2, pk->B_g2, (const uint64_t (*)[4])zb, pk->nB); msm_g1_mixed(&eB1, pk->B_g1, (const uint64_t (*)[4])zb, pk->nB); free(zb); }
  msm_g1_bdlo12(&eH, pk->H, (const uint64_t (*)[4])hc, m - 1);                                  /* :466-473 */
  msm_g1_mixed(&eL, pk->L, (const uint64_t (*)[4])(zc + ni + 1), nv - ni);                        /* :477-484 */
  uint64_t rc[4], sc[4], rsc[4]; fe rs; fe_to_canon(rc, r, R); fe_to_canon(sc, s, R); fe_mul(&rs, r, s, R); fe_to_canon(rsc, &rs, R);
  g1 t, gA, gB1, gC; g2 t2, gB2;
  g1_mul(&t, &pk->delta_g1, rc); g1_add(&gA, &pk->alpha_g1, &eA); g1_add(&gA, &gA, &t);           /* :488 */
  g1_mul(&t, &pk->delta_g1, sc); g1_add(&gB1, &pk->beta_g1, &eB1); g1_add(&gB1, &gB1, &t);        /* :491 */
  g2_mul(&t2, &pk->delta_g2, sc); g2_add(&gB2, &pk->beta_g2, &eB2); g2_add(&gB2, &gB2, &t2);      /* :492 */
  g1_add(&gC, &eH, &eL); g1_mul(&t, &gA, sc); g1_add(&gC, &gC, &t); g1_mul(&t, &gB1, rc); g1_add(&gC, &gC, &t);
  g1_mul(&t, &pk->delta_g1, rsc); g1_neg(&t, &t); g1_add(&gC, &gC, &t);                           /* :495 */
  g1_to_affine(&gA); g2_to_affine(&gB2); g1_to_affine(&gC); o->A = gA; o->B = gB2; o->C = gC;
  free(H); free(zc); free(hc); }

int groth16_verify(const verifying_key *vk, const fe *inputs, size_t n_inputs, const proof *p) {
  if (vk->nIC != n_inputs + 1) return 0;                                                          /* strong IC :584-590 */
  g1 acc = vk->IC[0], t; uint64_t c[4];
  for (size_t i = 0; i < n_inputs; i++) { fe_to_canon(c, &inputs[i], R); g1_mul(&t, &vk->IC[i + 1], c); g1_add(&acc, &acc, &t); }   /* accumulate_chunk */
  int ok = g1_on_curve(&p->A) && g2_on_curve(&p->B) && g1_on_curve(&p->C);                        /* is_well_formed: on-curve only */
  fe12 q1, q2a, q2b, q2, f, res; pairing_miller(&q1, &p->A, &p->B); pairing_miller(&q2a, &acc, &vk->gamma_g2); pairing_miller(&q2b, &p->C, &vk->delta_g2);
  fe12_mul(&q2, &q2a, &q2b); fe12_conj(&q2, &q2); fe12_mul(&f, &q1, &q2); pairing_final_exp(&res, &f);   /* :556-560 */
  return ok && fe12_eq(&res, &vk->alpha_g1_beta_g2); }

void groth16_setup(proving_key *pk, verifying_key *vk, const r1cs *cs, const fe *t, const fe *alpha, const fe *beta,
                   const fe *gamma, const fe *delta, const fe *g1_gen_k, const fe *g2_gen_k) {
  domain d; domain_init(&d, cs->n_cons + cs->n_inputs + 1); size_t m = d.m, nv = cs->n_vars, ni = cs->n_inputs, nc = cs->n_cons;
  fe *u = malloc(sizeof(fe) * m), *At = calloc(nv + 1, sizeof(fe)), *Bt = calloc(nv + 1, sizeof(fe)), *Ct = calloc(nv + 1, sizeof(fe)); fe Zt, x;
  domain_vanishing(&d, &Zt, t); domain_lagrange(&d, u, t);
  for (size_t i = 0; i <= ni; i++) At[i] = u[nc + i];                                             /* r1cs_to_qap.tcc:128-131 */
  fe *Mt[3] = {At, Bt, Ct};
  for (int mm = 0; mm < 3; mm++) for (size_t i = 0; i < nc; i++) for (uint32_t k = cs->rowptr[mm][i]; k < cs->rowptr[mm][i + 1]; k++) {
    fe_mul(&x, &u[i], &cs->coeff[mm][k], R); fe_add(&Mt[mm][cs->col[mm][k]], &Mt[mm][cs->col[mm][k]], &x, R); }
  fe gi, di; fe_inv(&gi, gamma, R); fe_inv(&di, delta, R);
  g1 G1g; g2 G2g; uint64_t c[4]; { g1 g; g1_gen(&g); fe_to_canon(c, g1_gen_k, R); g1_mul(&G1g, &g, c); g2 h; g2_gen(&h); fe_to_canon(c, g2_gen_k, R); g2_mul(&G2g, &h, c); }
#define G1MUL(dst, scalar) do { fe_to_canon(c, (scalar), R); g1_mul((dst), &G1g, c); g1_to_affine(dst); } while (0)
#define G2MUL(dst, scalar) do { fe_to_canon(c, (scalar), R); g2_mul((dst), &G2g, c); g2_to_affine(dst); } while (0)
  G1MUL(&pk->alpha_g1, alpha); G1MUL(&pk->beta_g1, beta); G2MUL(&pk->beta_g2, beta); G1MUL(&pk->delta_g1, delta); G2MUL(&pk->delta_g2, delta);
  pk->nA = nv + 1; pk->A = malloc(sizeof(g1) * pk->nA); for (size_t i = 0; i <= nv; i++) G1MUL(&pk->A[i], &At[i]);
  pk->nB = 0; for (size_t i = 0; i <= nv; i++) if (!fe_is_zero(&Bt[i])) pk->nB++;
  pk->B_idx = malloc(4 * (pk->nB + 1)); pk->B_g2 = malloc(sizeof(g2) * (pk->nB + 1)); pk->B_g1 = malloc(sizeof(g1) * (pk->nB + 1));
  for (size_t i = 0, j = 0; i <= nv; i++) if (!fe_is_zero(&Bt[i])) { pk->B_idx[j] = (uint32_t)i; G2MUL(&pk->B_g2[j], &Bt[i]); G1MUL(&pk->B_g1[j], &Bt[i]); j++; }
  pk->nH = m - 1; pk->H = malloc(sizeof(g1) * pk->nH); { fe coeff, ti; fe_mul(&coeff, &Zt, &di, R); fe_one(&ti, R);     /* :330 batch_exp_with_coeff(Zt/delta, Ht) */
    for (size_t i = 0; i < m - 1; i++) { fe_mul(&x, &coeff, &ti, R); G1MUL(&pk->H[i], &x); fe_mul(&ti, &ti, t, R); } }
  pk->nL = nv - ni; pk->L = malloc(sizeof(g1) * (pk->nL + 1));
  for (size_t i = 0; i < pk->nL; i++) { size_t j = ni + 1 + i; fe a, b; fe_mul(&a, beta, &At[j], R); fe_mul(&b, alpha, &Bt[j], R); fe_add(&a, &a, &b, R); fe_add(&a, &a, &Ct[j], R); fe_mul(&a, &a, &di, R); G1MUL(&pk->L[i], &a); }
  vk->nIC = ni + 1; vk->IC = malloc(sizeof(g1) * vk->nIC);
  for (size_t i = 0; i <= ni; i++) { fe a, b; fe_mul(&a, beta, &At[i], R); fe_mul(&b, alpha, &Bt[i], R); fe_add(&a, &a, &b, R); fe_add(&a, &a, &Ct[i], R); fe_mul(&a, &a, &gi, R); G1MUL(&vk->IC[i], &a); }
  G2MUL(&vk->gamma_g2, gamma); vk->delta_g2 = pk->delta_g2; pairing_reduced(&vk->alpha_g1_beta_g2, &pk->alpha_g1, &pk->beta_g2);
  free(u); free(At); free(Bt); free(Ct); }
