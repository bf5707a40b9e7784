// Fold and tail of the G2 witness MSM (the B query's G2 half) on nine 29-bit limbs, with the point spread over EIGHT lanes (round 5).
//
// Replaces k_wacc_fold<Fq2> / k_wtail<Fq2> (msm.cuh): quad-cooperative additions on 8 x 32-bit limbs in which every lane holds a whole Fq2 point — 256 VGPRs + 95
// AGPRs, one wave per SIMD, 233-325 us of kernel time right beside the transforms of every proof, whose workgroups could not share a compute unit with them.
// What they compute is unchanged (libsnark: the G2 half of kc_multi_exp_with_mixed_addition, SNARK/knowledge_commitment/kc_multiexp.tcc:21-85, as bucket sums):
//   fold   bucket b = the sum of its lanes' partial sums (k_wacc_lanes_g2_29); 256 of the ones' lanes per extra workgroup
//   tail   S_s = the sum of the buckets whose weight b + 1 has bit s (the host's Horner rule finishes), bucket NB - 1, the sum of the ones' partial sums
//
// Layout: an XYZZ point over Fq2 is eight Fq values; lane e of an aligned group of eight holds ONE of them — coordinate k = e & 3 of (X, Y, ZZ, ZZZ), component
// h = e >> 2 (c0 on the first quad, c1 on the second) — nine registers.  Inside each quad the choreography is quad29_add's (htail29.cuh: four rounds of one product
// per lane, add-2008-s); an Fq2 product needs the partner component from lane e ^ 4 (row_half_mirror + quad_perm: two DPP moves a limb) and is formed with ONE
// Montgomery reduction per component (Fq29::mul2, gen_field29.py):
//   component 0:  a0 b0 + (K_6 - a1) b1            component 1:  a0 b1 + a1 b0
// — 243 multiply-adds a lane and round where Karatsuba on one lane takes 3 x 162.  Differences and their constants are those of the G1 form; value bounds and the
// column bound of mul2 with the limb sizes used here: gen_field29.py, check_bounds_oct (X < 5.5 p, Y < 4.1 p, ZZ / ZZZ components < 5.7 p in and out).
// Operand = +-the other operand leaves ZZ = 0 (mod p): k_wtail_g2_29 looks for it in every result and raises the flag that sends the MSM to the general path.
#pragma once
#include "htail29.cuh"

namespace zk {

// a G2 partial sum / bucket sum in HBM: eight slots of 12 words (nine limbs + padding), slot e = 4 h + k as above; all-zero ZZ limbs (slots 2 and 6) = infinity
struct Point29Rec2 { uint32_t w[96]; };

// lane e reads lane e ^ 4 of its group of eight
__device__ __forceinline__ uint32_t oct29_partner_u32(uint32_t v) {
  const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false);        // row_half_mirror: lane i reads lane 7 - i
  return (uint32_t)__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xf, 0xf, false);            // quad_perm [3,2,1,0]: lane i reads lane i ^ 3
}
__device__ __forceinline__ Fq29 oct29_partner(const Fq29 &v) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = oct29_partner_u32(v.l[i]);
  return r;
}
// K_6 - v, normalized: v below 6 p with limbs below 3 * (2^29 + 8)
__device__ __forceinline__ Fq29 fq29_neg6(const Fq29 &v) {
  Fq29 z;
#pragma unroll
  for (int i = 0; i < 9; i++) z.l[i] = 0;
  return Fq29::sub<6>(z, v);
}
// this lane's component of the Fq2 product x y (x: limbs up to 2^30 + 8, value below 6 p per component; y: normalized)
__device__ __forceinline__ Fq29 oct29_mul(const Fq29 &x, const Fq29 &y, bool h0) {
  const Fq29 xp = oct29_partner(x), yp = oct29_partner(y);
  return Fq29::mul2(sel29(h0, x, xp), y, sel29(h0, fq29_neg6(xp), x), yp);
}
struct OPoint29 { Fq29 c; bool inf; };

__device__ __forceinline__ OPoint29 oct29_add(const OPoint29 &A, const OPoint29 &B, int k, bool h0) {
  const Fq29 &a = A.c, &b = B.c;
  // round 1
  const Fq29 m1 = oct29_mul(a, quad29_perm<0x4E>(b), h0);                                // U1 | S1 | U2 | S2
  const Fq29 d = Fq29::sub<2>(quad29_perm<0x4E>(m1), m1);                                // P | R | -P | -R
  // round 2
  const bool low = k < 2;
  const Fq29 m2 = oct29_mul(sel29(low, d, a), sel29(low, d, b), h0);                     // PP | RR | ZZ12 | ZZZ12
  // round 3
  const Fq29 pp = quad29_perm<0x00>(m2), u1 = quad29_perm<0x00>(m1);
  const Fq29 m3 = oct29_mul(sel29(k == 0, d, sel29(k == 1, u1, m2)), pp, h0);            // PPP | Q | ZZ3 | -
  const Fq29 ppp = quad29_perm<0x00>(m3);
  Fq29 s;
#pragma unroll
  for (int i = 0; i < 9; i++) s.l[i] = ppp.l[i] + 2u * m3.l[i];                           // lane 1: PPP + 2 Q
  const Fq29 x3 = Fq29::sub<4>(m2, s);                                                   // lane 1: X3 = RR - PPP - 2 Q
  const Fq29 t = Fq29::sub<6>(m3, x3);                                                   // lane 1: Q - X3
  // round 4
  const Fq29 zzz12 = quad29_perm<0xFF>(m2);
  const Fq29 s1 = Fq29::add_raw(m1, d);                                                  // lane 3: S2 + (2p + S1 - S2) = S1 + 2p
  const Fq29 m4 = oct29_mul(sel29(k == 0, zzz12, sel29(k == 1, d, s1)), sel29(k == 1, t, ppp), h0);   // ZZZ3 | R (Q - X3) | - | S1 PPP
  const Fq29 m4x = quad29_perm<0x2C>(m4);                                                // [0,3,2,0]: lane 1 reads lane 3, lane 3 reads lane 0
  const Fq29 y3 = Fq29::sub<2>(m4, m4x);                                                 // lane 1: Y3
  const Fq29 x3b = quad29_perm<0x55>(x3);                                                // X3 for lane 0
  OPoint29 r;
  r.c = sel29(k == 0, x3b, sel29(k == 1, y3, sel29(k == 2, m3, m4x)));                   // X3 | Y3 | ZZ3 | ZZZ3
  r.c = sel29(B.inf, a, sel29(A.inf, b, r.c));                                           // an operand at infinity: the other one
  r.inf = A.inf && B.inf;
  return r;
}

// this lane's slot of a record; the eight lanes agree on "infinity" by looking at both components of ZZ
__device__ __forceinline__ OPoint29 oct29_load(const Point29Rec2 *rec, int e) {
  const uint4 *src = reinterpret_cast<const uint4 *>(rec->w + 12 * e);
  const uint4 v0 = src[0], v1 = src[1], v2 = src[2];
  OPoint29 p;
  p.c.l[0] = v0.x; p.c.l[1] = v0.y; p.c.l[2] = v0.z; p.c.l[3] = v0.w; p.c.l[4] = v1.x; p.c.l[5] = v1.y; p.c.l[6] = v1.z; p.c.l[7] = v1.w; p.c.l[8] = v2.x;
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o |= p.c.l[i];
  const uint32_t zz = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0xAA, 0xf, 0xf, false);   // [2,2,2,2]: this quad's ZZ component
  p.inf = (zz | oct29_partner_u32(zz)) == 0;
  return p;
}
__device__ __forceinline__ void oct29_store(Point29Rec2 *rec, const OPoint29 &p, int e) {
  const uint32_t z = p.inf ? 0u : ~0u;
  uint4 *dst = reinterpret_cast<uint4 *>(rec->w + 12 * e);
  dst[0] = make_uint4(p.c.l[0] & z, p.c.l[1] & z, p.c.l[2] & z, p.c.l[3] & z);
  dst[1] = make_uint4(p.c.l[4] & z, p.c.l[5] & z, p.c.l[6] & z, p.c.l[7] & z);
  dst[2] = make_uint4(p.c.l[8] & z, 0u, 0u, 0u);
}
__device__ __forceinline__ OPoint29 oct29_inf() {
  OPoint29 p;
#pragma unroll
  for (int i = 0; i < 9; i++) p.c.l[i] = 0;
  p.inf = true;
  return p;
}
__device__ __forceinline__ OPoint29 oct29_shfl_down(const OPoint29 &p, int lanes) {
  OPoint29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.c.l[i] = __shfl_down(p.c.l[i], lanes, 64);
  r.inf = __shfl_down((int)p.inf, lanes, 64) != 0;
  return r;
}
// tree over the groups of eight of a workgroup of up to 512 threads: the 8 groups of a wave by shuffles, the waves through LDS (one record each). `live` = how
// many groups (the first ones) hold something; the sum is valid in group 0.
__device__ __forceinline__ OPoint29 block_oct29_tree(OPoint29 acc, Point29Rec2 *lds, uint32_t live) {
  const uint32_t q = threadIdx.x >> 3, wq = q & 7, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const int e = threadIdx.x & 7, k = e & 3; const bool h0 = e < 4;
#pragma unroll 1
  for (int dq = 4; dq >= 1; dq >>= 1) {
    if (wave * 8 + dq >= live) continue;
    const OPoint29 o = oct29_shfl_down(acc, 8 * dq);
    if (wq + dq < 8) acc = oct29_add(acc, o, k, h0);
  }
  if (n_waves == 1) return acc;
  if ((threadIdx.x & 63) < 8) oct29_store(lds + wave, acc, e);
  __syncthreads();
  if (wave == 0) {
    acc = oct29_inf();
    if (q < n_waves) acc = oct29_load(lds + q, e);
#pragma unroll 1
    for (int dq = 4; dq >= 1; dq >>= 1) {
      if ((uint32_t)dq * 8 >= live || (uint32_t)dq >= n_waves) continue;
      const OPoint29 o = oct29_shfl_down(acc, 8 * dq);
      if (q + dq < n_waves) acc = oct29_add(acc, o, k, h0);
    }
  }
  return acc;
}
// group 0 of a workgroup: the result as 8 x 32-bit words per component (a product with 2^256 mod p each; XYZZ<Fq2>: X.c0 X.c1 Y.c0 Y.c1 ZZ.c0 ...); ZZ = 0 (mod p)
// in both components of something that is not the point at infinity raises the flag of the fused path
__device__ __forceinline__ void oct29_emit(const OPoint29 &acc, int e, XYZZ<Fq2> *slot, MsmCounters *cnt) {
  const int k = e & 3, h = e >> 2;
  Fq out = Fq::zero();
  uint32_t zero = 0;
  if (!acc.inf) {
    acc.c.to_words(out.l);
    zero = out.is_zero_lazy() ? 1u : 0u;
    out = out.normalize();
  }
  const uint32_t both = zero & oct29_partner_u32(zero);
  if (e == 2 && both && !acc.inf) atomicOr(&cnt->pad[0], 2u);
  reinterpret_cast<Fq *>(slot)[2 * k + h] = out;
}

constexpr uint32_t OCT_BLOCK = 512;                                                     // 64 groups of eight
template <int UNIT>
__global__ void __launch_bounds__(OCT_BLOCK) k_wfold_g2_29(const Point29Rec2 *__restrict__ partial, const uint32_t *__restrict__ lane_off, uint32_t NB,
    uint32_t bucket_lanes, Point29Rec2 *__restrict__ out) {
  zk_take_prio(NB);
  __shared__ Point29Rec2 lds[OCT_BLOCK / 64];
  const uint32_t b = blockIdx.x, q = threadIdx.x >> 3, nq = blockDim.x >> 3;
  const int e = threadIdx.x & 7, k = e & 3; const bool h0 = e < 4;
  uint32_t beg, len;
  if (b < NB) { beg = lane_off[b]; len = lane_off[b + 1] - beg; } else { beg = bucket_lanes + (b - NB) * 256; len = 256; }
  OPoint29 acc = oct29_inf();
  if (q < len) {
    acc = oct29_load(partial + beg + q, e);
#pragma unroll 1
    for (uint32_t j = q + nq; j < len; j += nq) acc = oct29_add(acc, oct29_load(partial + beg + j, e), k, h0);
  }
  acc = block_oct29_tree(acc, lds, min(len, nq));
  if (threadIdx.x < 8) oct29_store(out + b, acc, e);
}
// workgroup s < top: S_s; workgroup top: bucket NB - 1 and the unused slots; workgroup top + 1: the sum of the ones' partial sums (slot `slots`); the last
// workgroup to finish hands the counters to the host
template <int UNIT>
__global__ void __launch_bounds__(OCT_BLOCK) k_wtail_g2_29(const Point29Rec2 *__restrict__ buckets, uint32_t NB, const Point29Rec2 *__restrict__ ones_partial,
    uint32_t n_ones_partial, uint32_t slots, XYZZ<Fq2> *__restrict__ res, MsmCounters *cnt, uint4 *copy_dst, uint32_t ticket_shift) {
  zk_take_prio(NB);
  __shared__ Point29Rec2 lds[OCT_BLOCK / 64];
  const uint32_t q = threadIdx.x >> 3, nq = OCT_BLOCK / 8, s_ = blockIdx.x, half = NB >> 1;
  uint32_t top = 0;
  while ((1u << top) < NB) top++;
  const int e = threadIdx.x & 7, k = e & 3; const bool h0 = e < 4;
  OPoint29 acc = oct29_inf();
  uint32_t slot = s_;
  if (s_ == top + 1) {                                                                  // the ones
    if (q < n_ones_partial) {
      acc = oct29_load(ones_partial + q, e);
#pragma unroll 1
      for (uint32_t j = q + nq; j < n_ones_partial; j += nq) acc = oct29_add(acc, oct29_load(ones_partial + j, e), k, h0);
    }
    acc = block_oct29_tree(acc, lds, min(n_ones_partial, nq));
    slot = slots;
  } else if (s_ == top) {                                                               // weight NB: one bucket; slots above `top` hold the point at infinity
    if (q == 0) acc = oct29_load(buckets + NB - 1, e);
    if (threadIdx.x >= 8 && threadIdx.x < 8 * (slots - top)) reinterpret_cast<Fq *>(res + top + (threadIdx.x >> 3))[2 * k + (e >> 2)] = Fq::zero();
  } else {
    auto bucket_of = [&](uint32_t i) { return (((i >> s_) << (s_ + 1)) | (1u << s_) | (i & ((1u << s_) - 1))) - 1; };
    if (q < half) {
      acc = oct29_load(buckets + bucket_of(q), e);
#pragma unroll 1
      for (uint32_t j = q + nq; j < half; j += nq) acc = oct29_add(acc, oct29_load(buckets + bucket_of(j), e), k, h0);
    }
    acc = block_oct29_tree(acc, lds, min(half, nq));
  }
  if (threadIdx.x < 8) oct29_emit(acc, e, res + slot, cnt);
  __syncthreads();                                                                      // (the flag of oct29_emit before the ticket, whatever lane raised it)
  if (threadIdx.x == 0) {
    __threadfence();
    if (((atomicAdd(&cnt->pad[1], 1u << ticket_shift) >> ticket_shift) & 0xffffu) == gridDim.x - 1) {   // (the follower of a shared sort counts in the high half)
      __threadfence();
      *copy_dst = *reinterpret_cast<const uint4 *>(cnt);
      atomicAnd(&cnt->pad[1], ~(0xffffu << ticket_shift));
    }
  }
}

}  // namespace zk
