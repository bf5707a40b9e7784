/* TEST INFRASTRUCTURE (see oracle.h).  Evaluation domains over Fr.
 * Follows FQFFT/evaluation_domain/get_evaluation_domain.tcc:33-52 (selection; only basic and step radix-2 are
 * reachable at BlockMaze's sizes), domains/basic_radix2_domain.tcc:25-112, basic_radix2_domain_aux.tcc:44-79
 * (serial FFT), :171-180 (_multiply_by_coset), :182-236 (Lagrange), domains/step_radix2_domain.tcc:20-260,
 * FF/algebra/fields/field_utils.tcc:36-51 (get_root_of_unity). */
#include "oracle.h"
#include "constants.h"
#include <stdlib.h>
#include <string.h>
#define R (&FR)
static size_t ceil_log2(size_t n) { size_t r = ((n & (n - 1)) == 0 ? 0 : 1); while (n > 1) { n >>= 1; r++; } return r; }
static void fr_pow_u64(fe *o, const fe *a, uint64_t e) { uint64_t ee[1] = {e}; fe_pow(o, a, ee, 1, R); }

void fr_root_of_unity(fe *o, size_t n) { /* field_utils.tcc:36-51: omega = root_of_unity^(2^(s - log n)) */
  size_t logn = ceil_log2(n); fe w; memcpy(w.l, FR_ROOT_OF_UNITY, 32); for (size_t i = FR_S; i > logn; --i) fe_sqr(&w, &w, R); *o = w; }

static size_t bitrev(size_t x, size_t bits) { size_t r = 0; for (size_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }
static void radix2_fft(fe *a, size_t n, const fe *omega) { /* basic_radix2_domain_aux.tcc:44-79 */
  size_t logn = ceil_log2(n);
  for (size_t k = 0; k < n; k++) { size_t rk = bitrev(k, logn); if (k < rk) { fe t = a[k]; a[k] = a[rk]; a[rk] = t; } }
  size_t m = 1;
  for (size_t s = 1; s <= logn; s++) { fe w_m; fr_pow_u64(&w_m, omega, n / (2 * m));
    for (size_t k = 0; k < n; k += 2 * m) { fe w; fe_one(&w, R);
      for (size_t j = 0; j < m; j++) { fe t; fe_mul(&t, &w, &a[k + j + m], R); fe_sub(&a[k + j + m], &a[k + j], &t, R); fe_add(&a[k + j], &a[k + j], &t, R); fe_mul(&w, &w, &w_m, R); } }
    m *= 2; } }
static void mul_by_coset(fe *a, size_t n, const fe *g) { fe u = *g; for (size_t i = 1; i < n; i++) { fe_mul(&a[i], &a[i], &u, R); fe_mul(&u, &u, g, R); } }

int domain_init(domain *d, size_t min_size) {
  memset(d, 0, sizeof *d); if (min_size <= 1) return -1;
  size_t lg = ceil_log2(min_size);
  if (min_size == ((size_t)1 << lg)) { if (lg > FR_S) return -1; d->m = min_size; d->kind = 0; fr_root_of_unity(&d->omega, min_size); return 0; }
  /* extended_radix2 needs logm == s+1: unreachable here.  step_radix2(min_size): */
  size_t big = (size_t)1 << (lg - 1), small = min_size - big, rounded_small = (size_t)1 << ceil_log2(small);
  size_t m = (small == rounded_small) ? min_size : big + rounded_small;   /* get_evaluation_domain.tcc:37-47 */
  if (m == ((size_t)1 << ceil_log2(m))) { d->m = m; d->kind = 0; fr_root_of_unity(&d->omega, m); return 0; }   /* big + rounded_small became a power of two */
  d->m = m; d->kind = 1; d->big_m = (size_t)1 << (ceil_log2(m) - 1); d->small_m = m - d->big_m;
  fr_root_of_unity(&d->omega, (size_t)1 << ceil_log2(m)); fe_sqr(&d->big_omega, &d->omega, R); fr_root_of_unity(&d->small_omega, d->small_m); return 0; }

static void step_fft(const domain *d, fe *a) { /* step_radix2_domain.tcc:39-78 */
  size_t B = d->big_m, S = d->small_m; fe *c = malloc(sizeof(fe) * B), *dd = malloc(sizeof(fe) * B), *e = calloc(S, sizeof(fe)); fe omega_i; fe_one(&omega_i, R);
  for (size_t i = 0; i < B; i++) { fe t; if (i < S) { fe_add(&c[i], &a[i], &a[i + B], R); fe_sub(&t, &a[i], &a[i + B], R); } else { c[i] = a[i]; t = a[i]; }
    fe_mul(&dd[i], &omega_i, &t, R); fe_mul(&omega_i, &omega_i, &d->omega, R); }
  size_t compr = B / S; for (size_t i = 0; i < S; i++) for (size_t j = 0; j < compr; j++) fe_add(&e[i], &e[i], &dd[i + j * S], R);
  radix2_fft(c, B, &d->big_omega); radix2_fft(e, S, &d->small_omega);
  memcpy(a, c, sizeof(fe) * B); memcpy(a + B, e, sizeof(fe) * S); free(c); free(dd); free(e); }
static void step_ifft(const domain *d, fe *a) { /* step_radix2_domain.tcc:80-153 */
  size_t B = d->big_m, S = d->small_m; fe *U0 = malloc(sizeof(fe) * B), *U1 = malloc(sizeof(fe) * S), *tmp = malloc(sizeof(fe) * B);
  memcpy(U0, a, sizeof(fe) * B); memcpy(U1, a + B, sizeof(fe) * S);
  fe wi, t; fe_inv(&wi, &d->big_omega, R); radix2_fft(U0, B, &wi); fe_inv(&wi, &d->small_omega, R); radix2_fft(U1, S, &wi);
  fe_from_u64(&t, B, R); fe_inv(&t, &t, R); for (size_t i = 0; i < B; i++) fe_mul(&U0[i], &U0[i], &t, R);
  fe_from_u64(&t, S, R); fe_inv(&t, &t, R); for (size_t i = 0; i < S; i++) fe_mul(&U1[i], &U1[i], &t, R);
  fe omega_i; fe_one(&omega_i, R); for (size_t i = 0; i < B; i++) { fe_mul(&tmp[i], &U0[i], &omega_i, R); fe_mul(&omega_i, &omega_i, &d->omega, R); }
  for (size_t i = S; i < B; i++) a[i] = U0[i];
  size_t compr = B / S; for (size_t i = 0; i < S; i++) for (size_t j = 1; j < compr; j++) fe_sub(&U1[i], &U1[i], &tmp[i + j * S], R);
  fe omega_inv, oi; fe_inv(&omega_inv, &d->omega, R); fe_one(&oi, R); for (size_t i = 0; i < S; i++) { fe_mul(&U1[i], &U1[i], &oi, R); fe_mul(&oi, &oi, &omega_inv, R); }
  fe over_two; fe_from_u64(&over_two, 2, R); fe_inv(&over_two, &over_two, R);
  for (size_t i = 0; i < S; i++) { fe s; fe_add(&s, &U0[i], &U1[i], R); fe_mul(&a[i], &s, &over_two, R); fe_sub(&s, &U0[i], &U1[i], R); fe_mul(&a[B + i], &s, &over_two, R); }
  free(U0); free(U1); free(tmp); }

void domain_fft(const domain *d, fe *a) { if (d->kind == 0) radix2_fft(a, d->m, &d->omega); else step_fft(d, a); }
void domain_ifft(const domain *d, fe *a) { if (d->kind == 1) { step_ifft(d, a); return; }
  fe wi, s; fe_inv(&wi, &d->omega, R); radix2_fft(a, d->m, &wi); fe_from_u64(&s, d->m, R); fe_inv(&s, &s, R); for (size_t i = 0; i < d->m; i++) fe_mul(&a[i], &a[i], &s, R); }
void domain_coset_fft(const domain *d, fe *a, const fe *g) { mul_by_coset(a, d->m, g); domain_fft(d, a); }
void domain_icoset_fft(const domain *d, fe *a, const fe *g) { fe gi; domain_ifft(d, a); fe_inv(&gi, g, R); mul_by_coset(a, d->m, &gi); }
void domain_vanishing(const domain *d, fe *o, const fe *t) { fe one, a, b, c; fe_one(&one, R);
  if (d->kind == 0) { fr_pow_u64(&a, t, d->m); fe_sub(o, &a, &one, R); return; }
  fr_pow_u64(&a, t, d->big_m); fe_sub(&a, &a, &one, R); fr_pow_u64(&b, t, d->small_m); fr_pow_u64(&c, &d->omega, d->small_m); fe_sub(&b, &b, &c, R); fe_mul(o, &a, &b, R); }
void domain_add_poly_Z(const domain *d, const fe *coeff, fe *H) {
  if (d->kind == 0) { fe_add(&H[d->m], &H[d->m], coeff, R); fe_sub(&H[0], &H[0], coeff, R); return; }
  fe w, cw; fr_pow_u64(&w, &d->omega, d->small_m); fe_mul(&cw, coeff, &w, R);
  fe_add(&H[d->m], &H[d->m], coeff, R); fe_sub(&H[d->big_m], &H[d->big_m], &cw, R); fe_sub(&H[d->small_m], &H[d->small_m], coeff, R); fe_add(&H[0], &H[0], &cw, R); }
void domain_divide_by_Z_on_coset(const domain *d, fe *P) { fe coset, one; memcpy(coset.l, FR_MULT_GEN, 32); fe_one(&one, R);
  if (d->kind == 0) { fe z; domain_vanishing(d, &z, &coset); fe_inv(&z, &z, R); for (size_t i = 0; i < d->m; i++) fe_mul(&P[i], &P[i], &z, R); return; }
  size_t B = d->big_m, S = d->small_m; fe Z0, cS, wS, cSZ0, wSZ0, w2S, elt, t;
  fr_pow_u64(&Z0, &coset, B); fe_sub(&Z0, &Z0, &one, R); fr_pow_u64(&cS, &coset, S); fe_mul(&cSZ0, &cS, &Z0, R); fr_pow_u64(&wS, &d->omega, S); fe_mul(&wSZ0, &wS, &Z0, R);
  fr_pow_u64(&w2S, &d->omega, 2 * S); fe_one(&elt, R);
  for (size_t i = 0; i < B; i++) { fe_mul(&t, &cSZ0, &elt, R); fe_sub(&t, &t, &wSZ0, R); fe_inv(&t, &t, R); fe_mul(&P[i], &P[i], &t, R); fe_mul(&elt, &elt, &w2S, R); }
  fe cw, a, b, Z1; fe_mul(&cw, &coset, &d->omega, R); fr_pow_u64(&a, &cw, B); fe_sub(&a, &a, &one, R); fr_pow_u64(&b, &cw, S); fe_sub(&b, &b, &wS, R); fe_mul(&Z1, &a, &b, R); fe_inv(&Z1, &Z1, R);
  for (size_t i = 0; i < S; i++) fe_mul(&P[B + i], &P[B + i], &Z1, R); }
static void radix2_lagrange(fe *u, size_t m, const fe *t) { /* basic_radix2_domain_aux.tcc:182-236 */
  fe one; fe_one(&one, R); if (m == 1) { u[0] = one; return; }
  fe omega, tm; fr_root_of_unity(&omega, m); fr_pow_u64(&tm, t, m); memset(u, 0, sizeof(fe) * m);
  if (fe_eq(&tm, &one)) { fe wi = one; for (size_t i = 0; i < m; i++) { if (fe_eq(&wi, t)) { u[i] = one; return; } fe_mul(&wi, &wi, &omega, R); } }
  fe Z, l, r, minv, d; fe_sub(&Z, &tm, &one, R); fe_from_u64(&minv, m, R); fe_inv(&minv, &minv, R); fe_mul(&l, &Z, &minv, R); r = one;
  for (size_t i = 0; i < m; i++) { fe_sub(&d, t, &r, R); fe_inv(&d, &d, R); fe_mul(&u[i], &l, &d, R); fe_mul(&l, &l, &omega, R); fe_mul(&r, &r, &omega, R); } }
void domain_lagrange(const domain *d, fe *u, const fe *t) {
  if (d->kind == 0) { radix2_lagrange(u, d->m, t); return; }
  size_t B = d->big_m, S = d->small_m; fe *ib = malloc(sizeof(fe) * B), *is = malloc(sizeof(fe) * S); fe oinv, tt, one; fe_one(&one, R);
  radix2_lagrange(ib, B, t); fe_inv(&oinv, &d->omega, R); fe_mul(&tt, t, &oinv, R); radix2_lagrange(is, S, &tt);
  fe L0, wS, bwS, elt, a; fr_pow_u64(&a, t, S); fr_pow_u64(&wS, &d->omega, S); fe_sub(&L0, &a, &wS, R); fr_pow_u64(&bwS, &d->big_omega, S); elt = one;
  for (size_t i = 0; i < B; i++) { fe_sub(&a, &elt, &wS, R); fe_inv(&a, &a, R); fe_mul(&u[i], &ib[i], &L0, R); fe_mul(&u[i], &u[i], &a, R); fe_mul(&elt, &elt, &bwS, R); }
  fe L1, b; fr_pow_u64(&a, t, B); fe_sub(&a, &a, &one, R); fr_pow_u64(&b, &d->omega, B); fe_sub(&b, &b, &one, R); fe_inv(&b, &b, R); fe_mul(&L1, &a, &b, R);
  for (size_t i = 0; i < S; i++) fe_mul(&u[B + i], &L1, &is[i], R);
  free(ib); free(is); }
