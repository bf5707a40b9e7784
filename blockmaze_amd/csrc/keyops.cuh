// Key-side kernels: batched point decompression when a reference-format proving key is loaded (kernel K8, SURVEY.md §8a
// row S1; replaces the per-point sqrt of alt_bn128_g1.cpp:420-465 / alt_bn128_g2.cpp:433-470) and fixed-base batch
// exponentiation for key generation (replaces libff's window-table batch_exp, multiexp.tcc:547-646).
#pragma once
#include <hip/hip_runtime.h>
#include "curve.cuh"

namespace zk {

struct FqPowConsts { uint32_t sqrt_exp[8]; uint32_t qm3o4[8]; uint32_t qm1o2[8]; };   // (q+1)/4, (q-3)/4, (q-1)/2

// y = sqrt(x^3 + 3) with the parity of the canonical y chosen by lsb; flags: bit0 = lsb of y, bit1 = point is zero.  ok[i] = 0 if x^3+3 is a non-residue.
__global__ void k_g1_decompress(const Fq *__restrict__ xs, const uint8_t *__restrict__ flags, Affine<Fq> *__restrict__ out, uint32_t n, FqPowConsts pc,
    uint32_t *bad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  if (flags[i] & 2) { out[i] = Affine<Fq>::inf(); return; }
  Fq x = xs[i], three = Fq::from_u64(3), y2 = x.sqr() * x + three, y = y2.pow(pc.sqrt_exp);
  if (y.sqr() != y2) { atomicAdd(bad, 1u); out[i] = Affine<Fq>::inf(); return; }
  if ((y.from_mont().l[0] & 1u) != (uint32_t)(flags[i] & 1)) y = y.neg();
  out[i] = {x, y};
}

__device__ __noinline__ Fq2 fq2_pow(const Fq2 &a, const uint32_t e[8]) { Fq2 r = Fq2::one(); bool found = false;
  for (int i = 255; i >= 0; i--) { if (found) r = r.sqr(); if ((e[i >> 5] >> (i & 31)) & 1) { found = true; r = r * a; } } return r; }
// square root in Fq2 = Fq[u]/(u^2+1), q = 3 mod 4 (Adj, Rodriguez-Henriquez, "Square root computation over even extension fields", Alg. 9)
__device__ __noinline__ bool fq2_sqrt(const Fq2 &a, Fq2 &out, const FqPowConsts &pc) {
  if (a.is_zero()) { out = a; return true; }
  Fq2 a1 = fq2_pow(a, pc.qm3o4), x0 = a1 * a, alpha = a1 * x0;
  Fq2 a0 = Fq2{alpha.c0, alpha.c1.neg()} * alpha;                       // alpha^(q+1)
  Fq2 minus_one = Fq2::one().neg();
  if (a0 == minus_one) return false;
  if (alpha == minus_one) out = Fq2{x0.c1.neg(), x0.c0};               // u * x0
  else { Fq2 b = fq2_pow(Fq2::one() + alpha, pc.qm1o2); out = b * x0; }
  return true;
}
__global__ void k_g2_decompress(const Fq2 *__restrict__ xs, const uint8_t *__restrict__ flags, Affine<Fq2> *__restrict__ out, uint32_t n, FqPowConsts pc,
    Fq2 twist_b, uint32_t *bad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  if (flags[i] & 2) { out[i] = Affine<Fq2>::inf(); return; }
  Fq2 x = xs[i], y2 = x.sqr() * x + twist_b, y;
  if (!fq2_sqrt(y2, y, pc) || y.sqr() != y2) { atomicAdd(bad, 1u); out[i] = Affine<Fq2>::inf(); return; }
  if ((y.c0.from_mont().l[0] & 1u) != (uint32_t)(flags[i] & 1)) y = y.neg();
  out[i] = {x, y};
}

// out[i] = scalars[i] * G for a fixed G, from an 8-bit window table: table[w * 255 + (d - 1)] = d * 2^(8w) * G (affine), w < 32.
// scalars canonical.  Result affine (one Fermat inversion per point; infinity for scalar 0).
template <class F>
__global__ void __launch_bounds__(128) k_fixed_base_mul(const Affine<F> *__restrict__ table, const Fr *__restrict__ scalars, Affine<F> *__restrict__ out,
    uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  Fr k = scalars[i]; XYZZ<F> acc = XYZZ<F>::inf();
  for (int w = 0; w < 32; w++) { uint32_t d = (k.l[w >> 2] >> ((w & 3) * 8)) & 0xffu; if (d) acc.madd_inl(table[w * 255 + d - 1]); }
  if (acc.is_inf()) { out[i] = Affine<F>::inf(); return; }
  F zi = acc.ZZ.inv(), z3i = acc.ZZZ.inv(); out[i] = {acc.X * zi, acc.Y * z3i};
}

}  // namespace zk
