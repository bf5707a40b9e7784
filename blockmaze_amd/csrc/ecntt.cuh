// Key-load transform of the H query into the Lagrange basis of the coset: an inverse NTT over group elements (see below).  Included by gpu.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include "curve.cuh"

namespace zk {

// ---- H query in the Lagrange basis of the coset (key load) ---------------------------------------------------------------
// The prover needs sum_i h_i H_i where h = icosetFFT(v) are the coefficients of the quotient polynomial and v its values on the coset (r1cs_to_qap.tcc:310-322,
// r1cs_gg_ppzksnark.tcc:466-473).  h_i = (g^-i / m) sum_j v_j w^(-ij), hence  sum_i h_i H_i = sum_j v_j P_j  with  P_j = sum_i w^(-ij) * ((g^-i / m) H_i):
// the inverse DFT, over GROUP ELEMENTS, of the scaled query (H_(m-1) := 0, its coefficient is zero anyway).  Computing P once per key (m/2 * log m point
// multiplications by twiddles) removes the seventh transform from every proof; the group element, and so the proof bytes, are the same.
// k * p, MSB-first double-and-add over the canonical bits of k
template <class F> __device__ __forceinline__ XYZZ<F> xyzz_mul_fr(const XYZZ<F> &p, const Fr &k_mont) {
  const Fr k = k_mont.from_mont(); XYZZ<F> r = XYZZ<F>::inf(); bool started = false;
#pragma unroll 1
  for (int i = 255; i >= 0; i--) {
    if (started) r = r.dbl_inl();
    if ((k.l[i >> 5] >> (i & 31)) & 1) {
      if (started) r.add_inl(p);
      else {
        r = p;
        started = true;
      }
    }
  }
  return r;
}
// data[i] = scale[i] * H_i for i < n_in, infinity for n_in <= i < m
__global__ void __launch_bounds__(64) k_ecntt_prescale(const Affine<Fq> *__restrict__ h, uint32_t n_in, const Fr *__restrict__ scale, uint32_t m,
    XYZZ<Fq> *__restrict__ data) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= m) return;
  data[i] = i < n_in ? xyzz_mul_fr(XYZZ<Fq>::from_affine(h[i]), scale[i]) : XYZZ<Fq>::inf();
}
// one decimation-in-frequency stage s (s = log m ... 1): (u, v) -> (u + v, w^j (u - v)); tw[j] = w^j for j < m/2 (here the inverse root's table); natural order
// in, bit-reversed out
__global__ void __launch_bounds__(64) k_ecntt_stage(XYZZ<Fq> *__restrict__ data, const Fr *__restrict__ tw, int logm, int s) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, half_m = 1u << (logm - 1); if (b >= half_m) return;
  const uint32_t half = 1u << (s - 1), j = b & (half - 1), i0 = ((b >> (s - 1)) << s) + j, i1 = i0 + half;
  XYZZ<Fq> u = data[i0], v = data[i1], sum = u; sum.add_inl(v); XYZZ<Fq> d = u; d.add_inl(v.neg());
  if (j) d = xyzz_mul_fr(d, tw[j << (logm - s)]);
  data[i0] = sum; data[i1] = d;
}
// Step-radix-2 domains (m = B + S, step_radix2_domain.tcc:79-140): the inverse transform is  Post . (iFFT_B (+) iFFT_S)  with the recombination pass
//   u1' = w^-i (U1[i] - sum_{k>=1} w^(i+kS) U0[i+kS]);  out[i] = (U0[i] + u1')/2;  out[B+i] = (U0[i] - u1')/2  (i < S);  out[i] = U0[i]  (S <= i < B),
// so the query needs the TRANSPOSE applied to its (coset-scaled) points Q: first Post^T below, with the 1/B and 1/S of the two inverse transforms folded in,
// then the two inverse DFTs over group elements (k_ecntt_stage on each part).
//   R0[j] = (Q[j] + Q[B+j]) / (2B)                                  j < S
//   R0[j] = Q[j] / B + (w^(j-i) / (2B)) (Q[B+i] - Q[i])              S <= j < B, i = j mod S
//   R1[i] = (w^-i / (2S)) (Q[i] - Q[B+i])                            i < S
__global__ void __launch_bounds__(64) k_ecntt_step_pre(const XYZZ<Fq> *__restrict__ Q, XYZZ<Fq> *__restrict__ R, const Fr *__restrict__ wpow,
    const Fr *__restrict__ winvpow, Fr half_inv_b, Fr inv_b, Fr half_inv_s, uint32_t B, uint32_t S) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= B + S) return;
  if (t < B) { const uint32_t j = t, i = j % S;
    if (j < S) { XYZZ<Fq> a = Q[j]; a.add_inl(Q[B + j]); R[j] = xyzz_mul_fr(a, half_inv_b); }
    else { XYZZ<Fq> d = Q[B + i]; d.add_inl(Q[i].neg()); XYZZ<Fq> r = xyzz_mul_fr(Q[j], inv_b); r.add_inl(xyzz_mul_fr(d, wpow[j - i] * half_inv_b)); R[j] = r; }
  } else { const uint32_t i = t - B; XYZZ<Fq> d = Q[i]; d.add_inl(Q[B + i].neg()); R[B + i] = xyzz_mul_fr(d, winvpow[i] * half_inv_s); }
}
// out[bitrev(p)] = affine(data[p])
// (data / out already offset to the sub-transform)
__global__ void __launch_bounds__(64) k_ecntt_finish(const XYZZ<Fq> *__restrict__ data, int logm, Affine<Fq> *__restrict__ out) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; if (p >= (1u << logm)) return; const XYZZ<Fq> q = data[p]; const uint32_t r = __brev(p) >> (32 - logm);
  if (q.is_inf()) { out[r] = Affine<Fq>::inf(); return; }
  Fq t = (q.ZZ * q.ZZZ).inv(); out[r] = {q.X * (t * q.ZZZ), q.Y * (t * q.ZZ)};
}

// ---- the C polynomial folded into the L query (key load, radix-2 domains) ----------------------------------------------- With P the coset-Lagrange H query,
// the H term of the proof is sum_j zinv (A_j B_j - C_j) P_j (A_j, B_j, C_j: values on the coset). The C part is LINEAR in the assignment: C_j = (cosetFFT .
// iFFT)(c)_j with c_k = <C_k, z>, so sum_j zinv C_j P_j = sum_k c_k U_k = sum_v z_v (sum_k C_kv U_k) with U = iFFT^T cosetFFT^T (zinv P) — two more DFTs over
// group elements and one sparse pass at key load. Subtracting the per-variable points from the L query (extended to all variables) removes the two transforms
// of C, and the C rows, from every proof; the proof's C element is the same group element.
__device__ __forceinline__ XYZZ<Fq> xyzz_mul_canon(const XYZZ<Fq> &p, const Fr &k) {   // like xyzz_mul_fr for a canonical (non-Montgomery) scalar
  XYZZ<Fq> r = XYZZ<Fq>::inf(); bool started = false;
#pragma unroll 1
  for (int i = 255; i >= 0; i--) {
    if (started) r = r.dbl_inl();
    if ((k.l[i >> 5] >> (i & 31)) & 1) {
      if (started) r.add_inl(p);
      else {
        r = p;
        started = true;
      }
    }
  }
  return r;
}
// data[i] = k * P_i (one scalar for all)
__global__ void __launch_bounds__(64) k_ecntt_scale_const(const Affine<Fq> *__restrict__ pts, Fr k_mont, uint32_t m, XYZZ<Fq> *__restrict__ data) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= m) return; data[i] = xyzz_mul_fr(XYZZ<Fq>::from_affine(pts[i]), k_mont);
}
// out[bitrev(p)] = scale[bitrev(p)] * in[p]: back to natural order between two transforms, with the diagonal factor in between
__global__ void __launch_bounds__(64) k_ecntt_permute_scale(const XYZZ<Fq> *__restrict__ in, const Fr *__restrict__ scale, int logm,
    XYZZ<Fq> *__restrict__ out) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (1u << logm)) return;
  const uint32_t r = __brev(p) >> (32 - logm);
  out[r] = xyzz_mul_fr(in[p], scale[r]);
}
// Step domains: the forward transform is  (DFT_B (+) DFT_S) . Pre  with  c[i] = a[i] + [i<S] a[i+B],  d[t] = w^t (a[t] - [t<S] a[t+B]),  e[i] = sum_j d[i + jS]
// (step_radix2_domain.tcc:39-77), preceded by the coset factor g^i.  Its transpose on points: the two DFTs of the parts, then
//   T[t] = g^t (C'[t] + w^t E'[t mod S])  (t < B),      T[B+t] = g^(B+t) (C'[t] - w^t E'[t])  (t < S),
// C' / E' being the transformed parts (held bit-reversed after the DIF stages).
__global__ void __launch_bounds__(64) k_ecntt_step_fwd_T(const XYZZ<Fq> *__restrict__ Cb, const XYZZ<Fq> *__restrict__ Eb, int logB, int logS,
    const Fr *__restrict__ wpow, const Fr *__restrict__ gpow, XYZZ<Fq> *__restrict__ T) {
  const uint32_t B = 1u << logB, S = 1u << logS; uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= B) return;
  const XYZZ<Fq> c = Cb[__brev(t) >> (32 - logB)]; XYZZ<Fq> we = xyzz_mul_fr(Eb[__brev(t % S) >> (32 - logS)], wpow[t]);
  XYZZ<Fq> lo = c; lo.add_inl(we); T[t] = xyzz_mul_fr(lo, gpow[t]);
  if (t < S) { XYZZ<Fq> hi = c; hi.add_inl(we.neg()); T[B + t] = xyzz_mul_fr(hi, gpow[B + t]); }
}
// data[i] = scale[i] * P_i
__global__ void __launch_bounds__(64) k_ecntt_scale_table(const Affine<Fq> *__restrict__ pts, const Fr *__restrict__ scale, uint32_t m,
    XYZZ<Fq> *__restrict__ data) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= m) return; data[i] = xyzz_mul_fr(XYZZ<Fq>::from_affine(pts[i]), scale[i]);
}
// natural-order copy of a bit-reversed array
__global__ void __launch_bounds__(64) k_ecntt_unpermute(const XYZZ<Fq> *__restrict__ in, int logm, XYZZ<Fq> *__restrict__ out) {
  uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; if (p >= (1u << logm)) return; out[__brev(p) >> (32 - logm)] = in[p];
}
// Lstar[v] = (v > n_inputs ? L[v - n_inputs - 1] : 0) - sum over the entries (k, coefficient) of column v of C of coefficient * U[k]; U is held bit-reversed.
// kind[e]: 0 = +1, 1 = -1, 2 = general (coef[e], canonical)
__global__ void __launch_bounds__(64) k_fold_c_columns(const uint32_t *__restrict__ colptr, const uint32_t *__restrict__ rowidx,
    const uint8_t *__restrict__ kind, const Fr *__restrict__ coef,
                                                       const XYZZ<Fq> *__restrict__ U_bitrev, int logm /* 0: U is in natural order */,
                                                           const Affine<Fq> *__restrict__ L, uint32_t n_inputs, uint32_t n_all, Affine<Fq> *__restrict__ out) {
  uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; if (v >= n_all) return; XYZZ<Fq> acc = XYZZ<Fq>::inf();
#pragma unroll 1
  for (uint32_t e = colptr[v]; e < colptr[v + 1]; e++) { XYZZ<Fq> u = U_bitrev[logm ? __brev(rowidx[e]) >> (32 - logm) : rowidx[e]]; uint8_t kd = kind[e];
    if (kd == 2) u = xyzz_mul_canon(u, coef[e]); else if (kd == 1) u = u.neg(); acc.add_inl(u); }
  acc = acc.neg(); if (v > n_inputs) acc.madd_inl(L[v - n_inputs - 1]);
  if (acc.is_inf()) { out[v] = Affine<Fq>::inf(); return; }
  Fq t = (acc.ZZ * acc.ZZZ).inv(); out[v] = {acc.X * (t * acc.ZZZ), acc.Y * (t * acc.ZZ)};
}

}  // namespace zk
