/* TEST INFRASTRUCTURE (see oracle.h).  Prime-field arithmetic for Fr and Fq of alt_bn128.
 * Follows FF/algebra/fields/fp.tcc: Montgomery representation with R = 2^256 on 4x64-bit limbs
 * (mul_reduce :23-190, add :310-400, sub :402-520, squared :594, inverse :688, sqrt :724) and
 * FF/algebra/fields/bigint.tcc (num_bits :104-123, test_bit :137-151). */
#include "oracle.h"
#include "constants.h"
#include <string.h>

typedef unsigned __int128 u128;
const fctx FR = {FR_MOD, FR_R1, FR_R2, FR_R3, FR_INV};
const fctx FQ = {FQ_MOD, FQ_R1, FQ_R2, FQ_R3, FQ_INV};

static int geq(const uint64_t *a, const uint64_t *b) { for (int i = 3; i >= 0; i--) { if (a[i] != b[i]) return a[i] > b[i]; } return 1; }
static uint64_t sub_n(uint64_t *o, const uint64_t *a, const uint64_t *b) { uint64_t br = 0;
  for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; o[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; } return br; }
static uint64_t add_n(uint64_t *o, const uint64_t *a, const uint64_t *b) { uint64_t c = 0;
  for (int i = 0; i < 4; i++) { u128 s = (u128)a[i] + b[i] + c; o[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } return c; }

/* fp.tcc:23-190 mul_reduce — CIOS: interleave one row of a*b with one Montgomery reduction step */
void fe_mul(fe *o, const fe *a, const fe *b, const fctx *F) {
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)a->l[j] * b->l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t k = t[0] * F->inv;
    c = (u128)k * F->mod[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128)k * F->mod[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
  }
  if (t[4] || geq(t, F->mod)) sub_n(o->l, t, F->mod); else memcpy(o->l, t, 32);
}
void fe_sqr(fe *o, const fe *a, const fctx *F) { fe_mul(o, a, a, F); }
void fe_add(fe *o, const fe *a, const fe *b, const fctx *F) { uint64_t t[4]; uint64_t c = add_n(t, a->l, b->l);
  if (c || geq(t, F->mod)) sub_n(o->l, t, F->mod); else memcpy(o->l, t, 32); }
void fe_sub(fe *o, const fe *a, const fe *b, const fctx *F) { uint64_t t[4]; if (sub_n(t, a->l, b->l)) add_n(t, t, F->mod); memcpy(o->l, t, 32); }
void fe_neg(fe *o, const fe *a, const fctx *F) { if (fe_is_zero(a)) { fe_zero(o); return; } uint64_t t[4]; sub_n(t, F->mod, a->l); memcpy(o->l, t, 32); }
void fe_dbl(fe *o, const fe *a, const fctx *F) { fe_add(o, a, a, F); }
int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
int fe_eq(const fe *a, const fe *b) { return memcmp(a->l, b->l, 32) == 0; }
void fe_zero(fe *o) { memset(o, 0, sizeof *o); }
void fe_one(fe *o, const fctx *F) { memcpy(o->l, F->r1, 32); }
void fe_from_canon(fe *o, const uint64_t c[4], const fctx *F) { fe t, r2; memcpy(t.l, c, 32); memcpy(r2.l, F->r2, 32); fe_mul(o, &t, &r2, F); }
void fe_from_u64(fe *o, uint64_t v, const fctx *F) { uint64_t c[4] = {v, 0, 0, 0}; fe_from_canon(o, c, F); }
void fe_to_canon(uint64_t c[4], const fe *a, const fctx *F) { fe one = {{1, 0, 0, 0}}, t; fe_mul(&t, a, &one, F); memcpy(c, t.l, 32); }
/* square-and-multiply, MSB first (FF/algebra/exponentiation/exponentiation.tcc:24-50) */
void fe_pow(fe *o, const fe *a, const uint64_t *e, int elimbs, const fctx *F) { fe r, base = *a; fe_one(&r, F); int found = 0;
  for (int i = elimbs * 64 - 1; i >= 0; i--) { if (found) fe_sqr(&r, &r, F); if ((e[i / 64] >> (i % 64)) & 1) { found = 1; fe_mul(&r, &r, &base, F); } } *o = r; }
/* fp.tcc:688 computes the inverse with mpn_gcdext; a^(p-2) gives the same field element */
void fe_inv(fe *o, const fe *a, const fctx *F) { uint64_t e[4]; uint64_t two[4] = {2, 0, 0, 0}; sub_n(e, F->mod, two); fe_pow(o, a, e, 4, F); }
/* fp.tcc:724 is Tonelli-Shanks; for q = 3 mod 4 (Fq::s = 1, alt_bn128_init.cpp:131) it yields a^((q+1)/4) up to sign.
 * The reference's loop: z = nqr_to_t = -1, w = a^((t-1)/2), x = a*w, b = x*w; if b != 1 then x *= z... we follow it literally. */
int fq_sqrt(fe *o, const fe *a) {
  fe one, w, x, b, z; fe_one(&one, &FQ);
  uint64_t euler[4]; memcpy(euler, FQ_EULER, 32);              /* t = (q-1)/2 since s = 1 */
  uint64_t tm1o2[4]; for (int i = 0; i < 4; i++) tm1o2[i] = (euler[i] >> 1) | (i < 3 ? euler[i + 1] << 63 : 0); /* (t-1)/2, t odd */
  fe_neg(&z, &one, &FQ);                                         /* nqr_to_t = 3^t = -1 */
  fe_pow(&w, a, tm1o2, 4, &FQ); fe_mul(&x, a, &w, &FQ); fe_mul(&b, &x, &w, &FQ);   /* b = a^t */
  if (fe_eq(&b, &one)) { *o = x; return 1; }
  /* v = s = 1: b^(2^0) != 1 means a is a non-residue unless b == -1 with m = ... ; with s = 1, b in {1,-1}; -1 => non-residue */
  return 0; }
int canon_is_zero(const uint64_t c[4]) { return (c[0] | c[1] | c[2] | c[3]) == 0; }
int canon_is_one(const uint64_t c[4]) { return c[0] == 1 && (c[1] | c[2] | c[3]) == 0; }
int canon_num_bits(const uint64_t c[4]) { for (int i = 3; i >= 0; i--) if (c[i]) return 64 * i + 64 - __builtin_clzll(c[i]); return 0; }
int canon_test_bit(const uint64_t c[4], int i) { return i < 256 ? (int)((c[i / 64] >> (i % 64)) & 1) : 0; }
