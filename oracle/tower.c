/* TEST INFRASTRUCTURE (see oracle.h).  Tower Fq2 / Fq6 / Fq12 over alt_bn128's Fq.
 * Follows FF/algebra/fields/fp2.tcc:73-200 (u^2 = -1), fp6_3over2.tcc:94-170 (v^3 = 9+u),
 * fp12_2over3over2.tcc:91-365 (w^2 = v; cyclotomic_squared :180-246, mul_by_024 :248-331, cyclotomic_exp :345-370).
 * Only values matter (field elements are unique), so products are written schoolbook rather than Karatsuba. */
#include "oracle.h"
#include "constants.h"
#include <string.h>
#define Q (&FQ)

void fe2_zero(fe2 *o) { memset(o, 0, sizeof *o); }
void fe2_one(fe2 *o) { fe_one(&o->c0, Q); fe_zero(&o->c1); }
int fe2_is_zero(const fe2 *a) { return fe_is_zero(&a->c0) && fe_is_zero(&a->c1); }
int fe2_eq(const fe2 *a, const fe2 *b) { return fe_eq(&a->c0, &b->c0) && fe_eq(&a->c1, &b->c1); }
void fe2_add(fe2 *o, const fe2 *a, const fe2 *b) { fe_add(&o->c0, &a->c0, &b->c0, Q); fe_add(&o->c1, &a->c1, &b->c1, Q); }
void fe2_sub(fe2 *o, const fe2 *a, const fe2 *b) { fe_sub(&o->c0, &a->c0, &b->c0, Q); fe_sub(&o->c1, &a->c1, &b->c1, Q); }
void fe2_neg(fe2 *o, const fe2 *a) { fe_neg(&o->c0, &a->c0, Q); fe_neg(&o->c1, &a->c1, Q); }
void fe2_dbl(fe2 *o, const fe2 *a) { fe2_add(o, a, a); }
void fe2_mul(fe2 *o, const fe2 *a, const fe2 *b) { /* (a0 + a1 u)(b0 + b1 u) = a0b0 - a1b1 + (a0b1 + a1b0) u */
  fe t0, t1, t2, t3; fe_mul(&t0, &a->c0, &b->c0, Q); fe_mul(&t1, &a->c1, &b->c1, Q); fe_mul(&t2, &a->c0, &b->c1, Q); fe_mul(&t3, &a->c1, &b->c0, Q);
  fe_sub(&o->c0, &t0, &t1, Q); fe_add(&o->c1, &t2, &t3, Q); }
void fe2_sqr(fe2 *o, const fe2 *a) { fe2 t = *a; fe2_mul(o, &t, &t); }
void fe2_mul_fq(fe2 *o, const fe2 *a, const fe *b) { fe_mul(&o->c0, &a->c0, b, Q); fe_mul(&o->c1, &a->c1, b, Q); }
void fe2_mul_xi(fe2 *o, const fe2 *a) { /* (9+u)(a0 + a1 u) = 9a0 - a1 + (9a1 + a0) u */
  fe nine, t0, t1; fe_from_u64(&nine, 9, Q); fe_mul(&t0, &a->c0, &nine, Q); fe_mul(&t1, &a->c1, &nine, Q);
  fe r0, r1; fe_sub(&r0, &t0, &a->c1, Q); fe_add(&r1, &t1, &a->c0, Q); o->c0 = r0; o->c1 = r1; }
void fe2_inv(fe2 *o, const fe2 *a) { /* fp2.tcc:121-135 */
  fe t0, t1, t2, t3; fe_sqr(&t0, &a->c0, Q); fe_sqr(&t1, &a->c1, Q); fe_add(&t2, &t0, &t1, Q); fe_inv(&t3, &t2, Q);
  fe r0, r1; fe_mul(&r0, &a->c0, &t3, Q); fe_mul(&r1, &a->c1, &t3, Q); fe_neg(&r1, &r1, Q); o->c0 = r0; o->c1 = r1; }
void fe2_frob(fe2 *o, const fe2 *a, unsigned p) { o->c0 = a->c0; if (p & 1) fe_neg(&o->c1, &a->c1, Q); else o->c1 = a->c1; }
static void fe2_pow(fe2 *o, const fe2 *a, const uint64_t *e, int elimbs) { fe2 r, b = *a; fe2_one(&r); int found = 0;
  for (int i = elimbs * 64 - 1; i >= 0; i--) { if (found) fe2_sqr(&r, &r); if ((e[i / 64] >> (i % 64)) & 1) { found = 1; fe2_mul(&r, &r, &b); } } *o = r; }
/* fp2.tcc:146-200 Tonelli-Shanks with s = 4 (alt_bn128_init.cpp:149-155).  Returns 0 for non-squares (the reference loops forever). */
int fe2_sqrt(fe2 *o, const fe2 *a) {
  fe2 one, z, w, x, b; fe2_one(&one); size_t v = FQ2_S; memcpy(&z, FQ2_NQR_TO_T, sizeof z);
  if (fe2_is_zero(a)) { fe2_zero(o); return 1; }
  fe2_pow(&w, a, FQ2_T_MINUS_1_OVER_2, 8); fe2_mul(&x, a, &w); fe2_mul(&b, &x, &w);
  { fe2 chk = b; for (size_t i = 0; i + 1 < v; i++) fe2_sqr(&chk, &chk); if (!fe2_eq(&chk, &one)) return 0; }
  while (!fe2_eq(&b, &one)) { size_t m = 0; fe2 b2m = b; while (!fe2_eq(&b2m, &one)) { fe2_sqr(&b2m, &b2m); m++; }
    int j = (int)v - (int)m - 1; w = z; while (j > 0) { fe2_sqr(&w, &w); --j; }
    fe2_sqr(&z, &w); fe2_mul(&b, &b, &z); fe2_mul(&x, &x, &w); v = m; }
  *o = x; return 1; }

/* ---- Fq6 ---- */
void fe6_add(fe6 *o, const fe6 *a, const fe6 *b) { fe2_add(&o->c0, &a->c0, &b->c0); fe2_add(&o->c1, &a->c1, &b->c1); fe2_add(&o->c2, &a->c2, &b->c2); }
void fe6_sub(fe6 *o, const fe6 *a, const fe6 *b) { fe2_sub(&o->c0, &a->c0, &b->c0); fe2_sub(&o->c1, &a->c1, &b->c1); fe2_sub(&o->c2, &a->c2, &b->c2); }
void fe6_neg(fe6 *o, const fe6 *a) { fe2_neg(&o->c0, &a->c0); fe2_neg(&o->c1, &a->c1); fe2_neg(&o->c2, &a->c2); }
void fe6_mul(fe6 *o, const fe6 *a, const fe6 *b) { /* v^3 = xi */
  fe2 t, r0, r1, r2, s;
  fe2_mul(&r0, &a->c0, &b->c0); fe2_mul(&t, &a->c1, &b->c2); fe2_mul(&s, &a->c2, &b->c1); fe2_add(&t, &t, &s); fe2_mul_xi(&t, &t); fe2_add(&r0, &r0, &t);
  fe2_mul(&r1, &a->c0, &b->c1); fe2_mul(&t, &a->c1, &b->c0); fe2_add(&r1, &r1, &t); fe2_mul(&t, &a->c2, &b->c2); fe2_mul_xi(&t, &t); fe2_add(&r1, &r1, &t);
  fe2_mul(&r2, &a->c0, &b->c2); fe2_mul(&t, &a->c1, &b->c1); fe2_add(&r2, &r2, &t); fe2_mul(&t, &a->c2, &b->c0); fe2_add(&r2, &r2, &t);
  o->c0 = r0; o->c1 = r1; o->c2 = r2; }
void fe6_sqr(fe6 *o, const fe6 *a) { fe6 t = *a; fe6_mul(o, &t, &t); }
void fe6_mul_by_v(fe6 *o, const fe6 *a) { fe2 t; fe2_mul_xi(&t, &a->c2); fe2 c0 = a->c0, c1 = a->c1; o->c0 = t; o->c1 = c0; o->c2 = c1; } /* Fp12 mul_by_non_residue */
void fe6_inv(fe6 *o, const fe6 *a) { /* fp6_3over2.tcc:128-146 */
  fe2 t0, t1, t2, t3, t4, t5, c0, c1, c2, t6, s;
  fe2_sqr(&t0, &a->c0); fe2_sqr(&t1, &a->c1); fe2_sqr(&t2, &a->c2); fe2_mul(&t3, &a->c0, &a->c1); fe2_mul(&t4, &a->c0, &a->c2); fe2_mul(&t5, &a->c1, &a->c2);
  fe2_mul_xi(&s, &t5); fe2_sub(&c0, &t0, &s); fe2_mul_xi(&s, &t2); fe2_sub(&c1, &s, &t3); fe2_sub(&c2, &t1, &t4);
  fe2_mul(&t6, &a->c2, &c1); fe2_mul(&s, &a->c1, &c2); fe2_add(&t6, &t6, &s); fe2_mul_xi(&t6, &t6); fe2_mul(&s, &a->c0, &c0); fe2_add(&t6, &t6, &s); fe2_inv(&t6, &t6);
  fe2_mul(&o->c0, &t6, &c0); fe2_mul(&o->c1, &t6, &c1); fe2_mul(&o->c2, &t6, &c2); }
void fe6_frob(fe6 *o, const fe6 *a, unsigned p) { fe2 t; fe2_frob(&o->c0, &a->c0, p);
  fe2_frob(&t, &a->c1, p); fe2_mul(&o->c1, (const fe2 *)FQ6_FROB_C1[p % 6], &t); fe2_frob(&t, &a->c2, p); fe2_mul(&o->c2, (const fe2 *)FQ6_FROB_C2[p % 6], &t); }

/* ---- Fq12 ---- */
void fe12_one(fe12 *o) { memset(o, 0, sizeof *o); fe2_one(&o->c0.c0); }
int fe12_eq(const fe12 *a, const fe12 *b) { return memcmp(a, b, sizeof *a) == 0; }
void fe12_mul(fe12 *o, const fe12 *a, const fe12 *b) { fe6 aA, bB, t, s, r0, r1;
  fe6_mul(&aA, &a->c0, &b->c0); fe6_mul(&bB, &a->c1, &b->c1); fe6_mul_by_v(&t, &bB); fe6_add(&r0, &aA, &t);
  fe6_mul(&t, &a->c0, &b->c1); fe6_mul(&s, &a->c1, &b->c0); fe6_add(&r1, &t, &s); o->c0 = r0; o->c1 = r1; }
void fe12_sqr(fe12 *o, const fe12 *a) { fe12 t = *a; fe12_mul(o, &t, &t); }
void fe12_inv(fe12 *o, const fe12 *a) { /* fp12_2over3over2.tcc:147-161 */
  fe6 t0, t1, t2, t3; fe6_sqr(&t0, &a->c0); fe6_sqr(&t1, &a->c1); fe6_mul_by_v(&t2, &t1); fe6_sub(&t2, &t0, &t2); fe6_inv(&t3, &t2);
  fe6 r0, r1; fe6_mul(&r0, &a->c0, &t3); fe6_mul(&r1, &a->c1, &t3); fe6_neg(&r1, &r1); o->c0 = r0; o->c1 = r1; }
void fe12_conj(fe12 *o, const fe12 *a) { o->c0 = a->c0; fe6_neg(&o->c1, &a->c1); } /* unitary_inverse :170-175 */
void fe12_frob(fe12 *o, const fe12 *a, unsigned p) { fe6 t; fe6_frob(&o->c0, &a->c0, p); fe6_frob(&t, &a->c1, p); const fe2 *c = (const fe2 *)FQ12_FROB_C1[p % 12];
  fe2_mul(&o->c1.c0, c, &t.c0); fe2_mul(&o->c1.c1, c, &t.c1); fe2_mul(&o->c1.c2, c, &t.c2); }
/* In the cyclotomic subgroup the Granger-Scott squaring (:180-246) equals the plain square as a field element. */
void fe12_cyclo_sqr(fe12 *o, const fe12 *a) { fe12_sqr(o, a); }
void fe12_cyclo_exp(fe12 *o, const fe12 *a, uint64_t e) { fe12 r, b = *a; fe12_one(&r); int found = 0; /* :345-370 */
  for (int j = 63; j >= 0; j--) { if (found) fe12_cyclo_sqr(&r, &r); if ((e >> j) & 1) { found = 1; fe12_mul(&r, &r, &b); } } *o = r; }
void fe12_mul_by_024(fe12 *o, const fe12 *a, const fe2 *ell_0, const fe2 *ell_VW, const fe2 *ell_VV) { /* "OLD: naive implementation" comment at :252-257 */
  fe12 s; memset(&s, 0, sizeof s); s.c0.c0 = *ell_0; s.c0.c2 = *ell_VV; s.c1.c1 = *ell_VW; fe12_mul(o, a, &s); }
