/* TEST INFRASTRUCTURE (see oracle.h).  alt_bn128 G1 / G2 group law — body in curve_body.inc. */
#include "oracle.h"
#include "constants.h"
#include <string.h>
static void fq_add(fe *o, const fe *a, const fe *b) { fe_add(o, a, b, &FQ); }
static void fq_sub(fe *o, const fe *a, const fe *b) { fe_sub(o, a, b, &FQ); }
static void fq_mul(fe *o, const fe *a, const fe *b) { fe_mul(o, a, b, &FQ); }
static void fq_sqr(fe *o, const fe *a) { fe_sqr(o, a, &FQ); }
static void fq_dbl(fe *o, const fe *a) { fe_dbl(o, a, &FQ); }
static void fq_neg(fe *o, const fe *a) { fe_neg(o, a, &FQ); }
static void fq_inv(fe *o, const fe *a) { fe_inv(o, a, &FQ); }
static void fq_one(fe *o) { fe_one(o, &FQ); }
static void fq_zero(fe *o) { fe_zero(o); }
static int fq_is_zero(const fe *a) { return fe_is_zero(a); }
static int fq_eq(const fe *a, const fe *b) { return fe_eq(a, b); }

#define G(x) g1_##x
#define F(x) fq_##x
#define GT g1
#define FT fe
#define COEFF_B ((const fe *)FQ_COEFF_B)
#include "curve_body.inc"
#undef G
#undef F
#undef GT
#undef FT
#undef COEFF_B

#define G(x) g2_##x
#define F(x) fe2_##x
#define GT g2
#define FT fe2
#define COEFF_B ((const fe2 *)TWIST_COEFF_B)
#include "curve_body.inc"

void g1_gen(g1 *o) { fe_from_u64(&o->X, 1, &FQ); fe_from_u64(&o->Y, 2, &FQ); fe_one(&o->Z, &FQ); }            /* alt_bn128_init.cpp:203-205 */
void g2_gen(g2 *o) { memcpy(&o->X, G2_GEN[0], sizeof(fe2)); memcpy(&o->Y, G2_GEN[1], sizeof(fe2)); fe2_one(&o->Z); } /* :262-269 */
void g2_mul_by_q(g2 *o, const g2 *a) { fe2 t; /* alt_bn128_g2.cpp:367-372 */
  fe2_frob(&t, &a->X, 1); fe2_mul(&o->X, (const fe2 *)TWIST_MUL_BY_Q[0], &t); fe2_frob(&t, &a->Y, 1); fe2_mul(&o->Y, (const fe2 *)TWIST_MUL_BY_Q[1], &t); fe2_frob(&o->Z, &a->Z, 1); }
