/* TEST INFRASTRUCTURE (see oracle.h).  Reader for the proving / verification key files the reference writes
 * (`Xpk.txt`, `Xvk.txt`; writeToFile at libsnark-vnt/src/send/sendcgo.cpp:50-62).  The format is a hybrid
 * (SURVEY.md §5.6): group elements come from libff.so's non-template operators built with
 * BINARY_OUTPUT + MONTGOMERY_OUTPUT + point compression (alt_bn128_g1.cpp:404-465, alt_bn128_g2.cpp:418-470:
 * ASCII is_zero flag, raw Montgomery limbs of X, ASCII LSB of canonical Y), everything else from templates instantiated
 * in the src/ TU without those flags: decimal text, "\n" / " " separators (r1cs_gg_ppzksnark.tcc:52-66,100-108,
 * sparse_vector.tcc:272-288, accumulation_vector.tcc:63-69, knowledge_commitment.tcc:121-125, variable.tcc:411-421,
 * r1cs.tcc:242-254, fp.tcc:779-790).  Point decompression: y = sqrt(x^3 + b), negated if its LSB differs. */
#include "oracle.h"
#include "constants.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
typedef struct { const unsigned char *p, *end; int err; } cur;

static void skip_ws(cur *c) { while (c->p < c->end && (*c->p == ' ' || *c->p == '\n' || *c->p == '\r' || *c->p == '\t')) c->p++; }
static void rd_dec(cur *c, u64 out[4]) { skip_ws(c); memset(out, 0, 32); int nd = 0;
  while (c->p < c->end && *c->p >= '0' && *c->p <= '9') { unsigned __int128 carry = *c->p - '0'; for (int i = 0; i < 4; i++) { unsigned __int128 v = (unsigned __int128)out[i] * 10 + carry; out[i] = (u64)v; carry = v >> 64; } c->p++; nd++; }
  if (!nd) c->err = 1; }
static size_t rd_size(cur *c) { u64 v[4]; rd_dec(c, v); return (size_t)v[0]; }
static void eat(cur *c, char ch) { if (c->p < c->end && *c->p == (unsigned char)ch) c->p++; else c->err = 1; }
static void rd_fq_mont(cur *c, fe *o) { if (c->end - c->p < 32) { c->err = 1; return; } memcpy(o->l, c->p, 32); c->p += 32; }
static void rd_g1(cur *c, u64 *out8) { /* -> canonical affine, all-zero for infinity */
  if (c->end - c->p < 34) { c->err = 1; return; } int is_zero = *c->p++ - '0'; fe X; rd_fq_mont(c, &X); int lsb = *c->p++ - '0';
  if (is_zero) { memset(out8, 0, 64); return; }
  fe x2, y2, Y; fe_sqr(&x2, &X, &FQ); fe_mul(&y2, &x2, &X, &FQ); fe_add(&y2, &y2, (const fe *)FQ_COEFF_B, &FQ); if (!fq_sqrt(&Y, &y2)) { c->err = 2; return; }
  fe_to_canon(out8 + 4, &Y, &FQ); if ((int)(out8[4] & 1) != lsb) { fe_neg(&Y, &Y, &FQ); fe_to_canon(out8 + 4, &Y, &FQ); } fe_to_canon(out8, &X, &FQ); }
static void rd_g2(cur *c, u64 *out16) {
  if (c->end - c->p < 66) { c->err = 1; return; } int is_zero = *c->p++ - '0'; fe2 X; rd_fq_mont(c, &X.c0); rd_fq_mont(c, &X.c1); int lsb = *c->p++ - '0';
  if (is_zero) { memset(out16, 0, 128); return; }
  fe2 x2, y2, Y; fe2_sqr(&x2, &X); fe2_mul(&y2, &x2, &X); fe2_add(&y2, &y2, (const fe2 *)TWIST_COEFF_B); if (!fe2_sqrt(&Y, &y2)) { c->err = 2; return; }
  u64 y0[4]; fe_to_canon(y0, &Y.c0, &FQ); if ((int)(y0[0] & 1) != lsb) fe2_neg(&Y, &Y);
  fe_to_canon(out16, &X.c0, &FQ); fe_to_canon(out16 + 4, &X.c1, &FQ); fe_to_canon(out16 + 8, &Y.c0, &FQ); fe_to_canon(out16 + 12, &Y.c1, &FQ); }
static unsigned char *slurp(const char *path, size_t *n) { FILE *f = fopen(path, "rb"); if (!f) return NULL; fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  unsigned char *b = malloc(sz + 1); if (fread(b, 1, sz, f) != (size_t)sz) { free(b); fclose(f); return NULL; } fclose(f); *n = sz; return b; }

typedef struct {
  u64 head[56]; size_t nA, nB, nH, nL, B_domain; u64 *A, *B_g2, *B_g1, *H, *L; uint32_t *B_idx;
  size_t n_inputs, n_vars, n_cons, nnz[3]; uint32_t *rowptr[3], *col[3]; u64 *coeff[3]; int err;
} parsed_pk;
static u64 *rd_g1_vec(cur *c, size_t *n) { *n = rd_size(c); eat(c, '\n'); u64 *v = malloc(64 * (*n + 1)); for (size_t i = 0; i < *n && !c->err; i++) rd_g1(c, v + 8 * i); return v; }

parsed_pk *o_pk_parse(const char *path) { size_t n; unsigned char *buf = slurp(path, &n); if (!buf) return NULL; cur c = {buf, buf + n, 0}; parsed_pk *pk = calloc(1, sizeof *pk);
  rd_g1(&c, pk->head); eat(&c, '\n'); rd_g1(&c, pk->head + 8); eat(&c, '\n'); rd_g2(&c, pk->head + 24); eat(&c, '\n'); rd_g1(&c, pk->head + 16); eat(&c, '\n'); rd_g2(&c, pk->head + 40); eat(&c, '\n');
  pk->A = rd_g1_vec(&c, &pk->nA);
  pk->B_domain = rd_size(&c); size_t ni = rd_size(&c); pk->B_idx = malloc(4 * (ni + 1)); for (size_t i = 0; i < ni; i++) pk->B_idx[i] = (uint32_t)rd_size(&c);
  pk->nB = rd_size(&c); eat(&c, '\n'); if (pk->nB != ni) c.err = 3; pk->B_g2 = malloc(128 * (pk->nB + 1)); pk->B_g1 = malloc(64 * (pk->nB + 1));
  for (size_t i = 0; i < pk->nB && !c.err; i++) { rd_g2(&c, pk->B_g2 + 16 * i); eat(&c, ' '); rd_g1(&c, pk->B_g1 + 8 * i); eat(&c, '\n'); }
  pk->H = rd_g1_vec(&c, &pk->nH); pk->L = rd_g1_vec(&c, &pk->nL);
  pk->n_inputs = rd_size(&c); pk->n_vars = pk->n_inputs + rd_size(&c); pk->n_cons = rd_size(&c);
  size_t cap[3]; for (int m = 0; m < 3; m++) { cap[m] = 4 * pk->n_cons + 16; pk->rowptr[m] = malloc(4 * (pk->n_cons + 1)); pk->rowptr[m][0] = 0; pk->col[m] = malloc(4 * cap[m]); pk->coeff[m] = malloc(32 * cap[m]); }
  for (size_t i = 0; i < pk->n_cons && !c.err; i++) for (int m = 0; m < 3; m++) { size_t nt = rd_size(&c);
    if (pk->nnz[m] + nt > cap[m]) { cap[m] = 2 * (pk->nnz[m] + nt); pk->col[m] = realloc(pk->col[m], 4 * cap[m]); pk->coeff[m] = realloc(pk->coeff[m], 32 * cap[m]); }
    for (size_t k = 0; k < nt; k++) { pk->col[m][pk->nnz[m]] = (uint32_t)rd_size(&c); rd_dec(&c, pk->coeff[m] + 4 * pk->nnz[m]); pk->nnz[m]++; }
    pk->rowptr[m][i + 1] = (uint32_t)pk->nnz[m]; }
  pk->err = c.err; free(buf); return pk; }
void o_pk_sizes(const parsed_pk *pk, size_t *out) { size_t v[12] = {pk->nA, pk->nB, pk->nH, pk->nL, pk->B_domain, pk->n_inputs, pk->n_vars, pk->n_cons, pk->nnz[0], pk->nnz[1], pk->nnz[2], (size_t)pk->err}; memcpy(out, v, sizeof v); }
/* which: 0 head 1 A 2 B_g2 3 B_g1 4 H 5 L 6 B_idx 7.. rowptr/col/coeff for A,B,C */
const void *o_pk_ptr(const parsed_pk *pk, int which) { switch (which) { case 0: return pk->head; case 1: return pk->A; case 2: return pk->B_g2; case 3: return pk->B_g1; case 4: return pk->H; case 5: return pk->L; case 6: return pk->B_idx;
    default: { int m = (which - 7) / 3, k = (which - 7) % 3; return k == 0 ? (void *)pk->rowptr[m] : k == 1 ? (void *)pk->col[m] : (void *)pk->coeff[m]; } } }
void o_pk_free(parsed_pk *pk) { free(pk->A); free(pk->B_g2); free(pk->B_g1); free(pk->H); free(pk->L); free(pk->B_idx); for (int m = 0; m < 3; m++) { free(pk->rowptr[m]); free(pk->col[m]); free(pk->coeff[m]); } free(pk); }

/* vk: gt(12 decimal) \n gamma_g2 \n delta_g2 \n IC0 \n domain \n n \n idx.. n \n {G1 \n}n \n \n   -> gt 48 words, gamma 16, delta 16, IC (nIC x 8).  Returns nIC or 0 on error. */
size_t o_vk_parse(const char *path, u64 *gt, u64 *gamma_g2, u64 *delta_g2, u64 *IC, size_t ic_cap) { size_t n; unsigned char *buf = slurp(path, &n); if (!buf) return 0; cur c = {buf, buf + n, 0};
  for (int i = 0; i < 12; i++) rd_dec(&c, gt + 4 * i); eat(&c, '\n'); rd_g2(&c, gamma_g2); eat(&c, '\n'); rd_g2(&c, delta_g2); eat(&c, '\n');
  if (ic_cap < 1) { free(buf); return 0; } rd_g1(&c, IC); eat(&c, '\n'); size_t dom = rd_size(&c); size_t ni = rd_size(&c); for (size_t i = 0; i < ni; i++) if (rd_size(&c) != i) c.err = 4;
  size_t nv = rd_size(&c); eat(&c, '\n'); if (nv != ni || dom != ni || nv + 1 > ic_cap) c.err = 5; for (size_t i = 0; i < nv && !c.err; i++) { rd_g1(&c, IC + 8 * (i + 1)); eat(&c, '\n'); }
  free(buf); return c.err ? 0 : nv + 1; }
