// Weighted bucket sum of the H query on nine 29-bit limbs: sum_b (b + 1) S_b over NB = 2^top buckets, after k_hacc_combine29 (msm.cuh).
//
// Replaces libff's running-sum reduction of the bucket array (FF/algebra/scalar_multiplication/multiexp.tcc:244-278) — a chain of 2 NB dependent additions
// there — and round 2's k_bitsum_chunks / k_bitsum_final (sums by weight bit on 8 x 32-bit limbs: top * NB / 2 = 245 K quad additions for send, 13 % of a
// proof's VALU instructions).
//
// Same result slots as before — T_s = the sum of the buckets whose weight w = b + 1 has bit s, s = 0 .. top, finished by the host's Horner rule — but formed in
// two levels:
//   w = hi * L + lo  (L = 2^lo_bits)      C_lo = sum over hi of S_(hi, lo)      R_hi = sum over lo of S_(hi, lo)
//   T_s = sum of the C_lo whose lo has bit s              (s <  lo_bits)
//   T_s = sum of the R_hi whose hi has bit s - lo_bits    (s >= lo_bits;  T_top = R_(2^hi_bits) = bucket NB - 1 alone)
// i.e. 2 NB additions for the marginal sums and a few hundred for the bit sums instead of top * NB / 2: 3.7 x fewer for NB = 32,768, and a dependent chain of 7
// + 7 additions instead of 16 + 6. Everything stays on the 29-bit limbs of the accumulation (field29_gfx950.inc); the one conversion per result slot happens at
// the very end.
//
// The additions are quad-cooperative like curve.cuh's, with the point SPREAD over the quad: lane k of a DPP quad holds coordinate k of (X, Y, ZZ, ZZZ) — nine
// registers a point instead of 36 — and fetches what it needs from its neighbours with quad_perm moves. One addition = four rounds of ONE product per lane
// (add-2008-s):
//   round 1   U1 = X1 ZZ2 | S1 = Y1 ZZZ2 | U2 = X2 ZZ1 | S2 = Y2 ZZZ1                     (own coordinate of a  x  the partner lane's coordinate of b)
//   round 2 P^2 | R^2 | ZZ1 ZZ2 | ZZZ1 ZZZ2 (P = U2 - U1 on lane 0, R = S2 - S1 on lane 1; lanes 2, 3 hold -P, -R)
//   round 3   P PP | U1 PP | ZZ12 PP | (idle)                                              -> PPP, Q, ZZ3
//   round 4 ZZZ12 PPP | R (Q - X3) | (idle) | S1 PPP -> ZZZ3, and Y3 = R (Q - X3) - S1 PPP; S1 is rebuilt on lane 3 as S2 + (-R)
// Value bounds: the formulas and reduction constants are those of xyzz29_add (msm.cuh), whose interval arithmetic lives in gen_field29.py (check_bounds_add);
// the one difference — S1 enters its product as S2 + (2p + S1 - S2) < S1 + 2p — is covered by check_bounds_add_quad there.
// Operand = +-the other operand leaves ZZ = 0 (mod p) for good, exactly as in the accumulation: k_hbits29 looks for it in every result slot and raises the flag
// that sends the MSM to the general path (complete formulas). The point at infinity is a record whose ZZ limbs are all zero; it is tracked as a quad-uniform
// flag beside the limbs.
#pragma once
#include "curve.cuh"
#include "field29.cuh"

namespace zk {

// bucket sums and marginal sums in HBM: four coordinate slots of 12 words (nine limbs + padding: every lane of a quad moves its slot as three 16-byte words)
struct Point29Rec { uint32_t w[48]; };

template <int CTRL> __device__ __forceinline__ Fq29 quad29_perm(const Fq29 &v) {   // quad_perm: lane i of every quad reads lane (CTRL >> 2i) & 3
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], CTRL, 0xf, 0xf, false);
  return r;
}
__device__ __forceinline__ Fq29 sel29(bool c, const Fq29 &a, const Fq29 &b) {
  Fq29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}
// a point spread over a quad: this lane's coordinate, and whether the point is the point at infinity (the same value in all four lanes)
struct QPoint29 { Fq29 c; bool inf; };

// OPP (the witness MSMs' folds and tails): an operand equal to MINUS the other one is noticed — P^2 = 0 (mod p) on lane 0 while R^2 on lane 1 is not — and the sum is
// the point at infinity, as it should be.  A key's query holds equal points (variables that enter the same constraints with the same coefficients), their small signed
// digits cancel inside lanes, and now and then two lanes of a bucket are left with one such point each, with opposite signs (found in round 6 from a state dump of
// mint's A query, tools/degenerate_dump.py: 2 to 4 events in 100,000 mint proofs under concurrent callers, each one MSM repeated on the general path).  An operand
// EQUAL to the other (a third of the events) is added by way of the generator, see below: the witness folds are complete.
template <bool OPP = false> __device__ __forceinline__ QPoint29 quad29_add(const QPoint29 &A, const QPoint29 &B, int k) {
  const Fq29 &a = A.c, &b = B.c;
  // round 1
  const Fq29 m1 = Fq29::mul(a, quad29_perm<0x4E>(b));                                  // [2,3,0,1]:  U1 | S1 | U2 | S2
  const Fq29 d = Fq29::sub<2>(quad29_perm<0x4E>(m1), m1);                              // P | R | -P | -R
  // round 2
  const bool low = k < 2;
  const Fq29 m2 = Fq29::mul(sel29(low, d, a), sel29(low, d, b));                       // PP | RR | ZZ12 | ZZZ12
  bool opposite = false, zf_p = false, zf_r = false;
  if (OPP) { uint32_t zero_or = 0, p_xor = 0;                                          // (a product's result — exact limbs, below 2 p — is 0 mod p iff it is 0 or p)
#pragma unroll
    for (int i = 0; i < 9; i++) { zero_or |= m2.l[i]; p_xor |= m2.l[i] ^ Fq29::P29[i]; }
    const int zf = (zero_or == 0 || p_xor == 0) ? 1 : 0;
    zf_p = __builtin_amdgcn_update_dpp(0, zf, 0x00, 0xf, 0xf, false) != 0; zf_r = __builtin_amdgcn_update_dpp(0, zf, 0x55, 0xf, 0xf, false) != 0;
    opposite = zf_p && !zf_r && !A.inf && !B.inf; }
  // round 3
  const Fq29 pp = quad29_perm<0x00>(m2), u1 = quad29_perm<0x00>(m1);
  const Fq29 m3 = Fq29::mul(sel29(k == 0, d, sel29(k == 1, u1, m2)), pp);              // PPP | Q | ZZ3 | -
  const Fq29 ppp = quad29_perm<0x00>(m3);
  Fq29 s;
#pragma unroll
  for (int i = 0; i < 9; i++) s.l[i] = ppp.l[i] + 2u * m3.l[i];                         // lane 1: PPP + 2 Q
  const Fq29 x3 = Fq29::sub<4>(m2, s);                                                 // lane 1: X3 = RR - PPP - 2 Q
  const Fq29 t = Fq29::sub<6>(m3, x3);                                                 // lane 1: Q - X3
  // round 4
  const Fq29 zzz12 = quad29_perm<0xFF>(m2);
  const Fq29 s1 = Fq29::add_raw(m1, d);                                                // lane 3: S2 + (2p + S1 - S2) = S1 + 2p  (limbs below 2^30 + 8)
  const Fq29 m4 = Fq29::mul(sel29(k == 0, zzz12, sel29(k == 1, d, s1)), sel29(k == 1, t, ppp));   // ZZZ3 | R (Q - X3) | - | S1 PPP
  const Fq29 m4x = quad29_perm<0x2C>(m4);                                              // [0,3,2,0]: lane 1 reads lane 3, lane 3 reads lane 0
  const Fq29 y3 = Fq29::sub<2>(m4, m4x);                                               // lane 1: Y3
  const Fq29 x3b = quad29_perm<0x55>(x3);                                              // X3 for lane 0
  QPoint29 r;
  r.c = sel29(k == 0, x3b, sel29(k == 1, y3, sel29(k == 2, m3, m4x)));                 // X3 | Y3 | ZZ3 | ZZZ3
  r.c = sel29(B.inf, a, sel29(A.inf, b, r.c));                                         // an operand at infinity: the other one
  r.inf = (A.inf && B.inf) || opposite;
  if (OPP) {
    // an operand EQUAL to the other one (P^2 = R^2 = 0): A + B = (A + G) + (B - G) with the curve's generator G = (1, 2) — three additions none of which is
    // degenerate (A = +-G aside), on the same formulas and bounds; the branch is taken by a whole wave, once in some hundred thousand proofs
    const bool equal = zf_p && zf_r && !A.inf && !B.inf;
    if (__ballot(equal)) {
      const Fq29 one = Fq29::one(), two = Fq29::add_raw(one, one).norm(); QPoint29 G, Gm;
      G.c = k == 1 ? two : one; G.inf = false; Gm.c = k == 1 ? Fq29::cond_neg(two, true).norm() : one; Gm.inf = false;
      const QPoint29 t3 = quad29_add<false>(quad29_add<false>(A, G, k), quad29_add<false>(B, Gm, k), k);
      if (equal) r = t3;
    }
  }
  return r;
}

// this lane's coordinate slot of a record; the quad agrees on "infinity" by looking at lane 2's ZZ limbs
__device__ __forceinline__ QPoint29 quad29_load(const Point29Rec *rec, int k) {
  const uint4 *src = reinterpret_cast<const uint4 *>(rec->w + 12 * k);
  const uint4 v0 = src[0], v1 = src[1], v2 = src[2];
  QPoint29 p;
  p.c.l[0] = v0.x; p.c.l[1] = v0.y; p.c.l[2] = v0.z; p.c.l[3] = v0.w; p.c.l[4] = v1.x; p.c.l[5] = v1.y; p.c.l[6] = v1.z; p.c.l[7] = v1.w; p.c.l[8] = v2.x;
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) o |= p.c.l[i];
  p.inf = __builtin_amdgcn_update_dpp(0, (int)o, 0xAA, 0xf, 0xf, false) == 0;           // [2,2,2,2]: ZZ
  return p;
}
__device__ __forceinline__ void quad29_store(Point29Rec *rec, const QPoint29 &p, int k) {
  const uint32_t z = p.inf ? 0u : ~0u;                                                  // the point at infinity is stored as all-zero limbs
  uint4 *dst = reinterpret_cast<uint4 *>(rec->w + 12 * k);
  dst[0] = make_uint4(p.c.l[0] & z, p.c.l[1] & z, p.c.l[2] & z, p.c.l[3] & z);
  dst[1] = make_uint4(p.c.l[4] & z, p.c.l[5] & z, p.c.l[6] & z, p.c.l[7] & z);
  dst[2] = make_uint4(p.c.l[8] & z, 0u, 0u, 0u);
}
__device__ __forceinline__ QPoint29 quad29_inf() {
  QPoint29 p;
#pragma unroll
  for (int i = 0; i < 9; i++) p.c.l[i] = 0;
  p.inf = true;
  return p;
}
__device__ __forceinline__ QPoint29 quad29_shfl_down(const QPoint29 &p, int lanes) {
  QPoint29 r;
#pragma unroll
  for (int i = 0; i < 9; i++) r.c.l[i] = __shfl_down(p.c.l[i], lanes, 64);
  r.inf = __shfl_down((int)p.inf, lanes, 64) != 0;
  return r;
}
// tree over the quads of a workgroup of up to 256 threads: the 16 quads of a wave by shuffles, the waves through LDS (one record each). `live` = how many quads
// (the first ones) hold something: levels whose partner quads are all empty are skipped, whole waves at a time. The sum is valid in quad 0.
template <bool OPP = false> __device__ __forceinline__ QPoint29 block_quad29_tree(QPoint29 acc, Point29Rec *lds, uint32_t live) {
  const uint32_t q = threadIdx.x >> 2, wq = q & 15, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const int k = threadIdx.x & 3;
#pragma unroll 1
  for (int dq = 8; dq >= 1; dq >>= 1) {
    if (wave * 16 + dq >= live) continue;
    const QPoint29 o = quad29_shfl_down(acc, 4 * dq);
    if (wq + dq < 16) acc = quad29_add<OPP>(acc, o, k);
  }
  if (n_waves == 1) return acc;
  if ((threadIdx.x & 63) < 4) quad29_store(lds + wave, acc, k);
  __syncthreads();
  if (wave == 0) {
    acc = quad29_inf();
    if (q < n_waves) acc = quad29_load(lds + q, k);
#pragma unroll 1
    for (int dq = 2; dq >= 1; dq >>= 1) {
      if ((uint32_t)dq * 16 >= live || (uint32_t)dq >= n_waves) continue;
      const QPoint29 o = quad29_shfl_down(acc, 4 * dq);
      if (q + dq < n_waves) acc = quad29_add<OPP>(acc, o, k);
    }
  }
  return acc;
}

__device__ __forceinline__ void quad29_emit(const QPoint29 &acc, int k, XYZZ<Fq> *slot, MsmCounters *cnt, uint32_t which = 0);
// how the weights are cut: w = hi * 2^lo_bits + lo. A row (2^lo_bits consecutive buckets) is added up in `row_chunks` pieces of at most HTAIL_CHUNK buckets, a
// workgroup each, so that no quad of the marginal kernel holds more than two buckets: 1 + 6 dependent additions for rows and columns alike.
constexpr uint32_t HTAIL_CHUNK = 128;
struct HtailShape { uint32_t top, lo_bits, hi_bits, row_chunks; };
__host__ __device__ inline HtailShape htail_shape(uint32_t NB) {
  HtailShape s; s.top = 0;
  while ((1u << s.top) < NB) s.top++;
  s.lo_bits = (s.top + 1) / 2; s.hi_bits = s.top - s.lo_bits;
  s.row_chunks = (1u << s.lo_bits) > HTAIL_CHUNK ? (1u << s.lo_bits) / HTAIL_CHUNK : 1u;
  return s;
}
// marginal sums: marg[lo] = C_lo for lo = 1 .. L - 1 (slot 0 unused); marg[L + hi * row_chunks + j] = piece j of R_hi for hi = 1 .. H - 1; marg[L + H *
// row_chunks] = R_H = bucket NB - 1 (the only bucket of that row)
__host__ __device__ inline uint32_t htail_marg_count(const HtailShape &s) { return (1u << s.lo_bits) + ((1u << s.hi_bits)) * s.row_chunks + 1; }
__host__ __device__ inline uint32_t htail_marg_blocks(const HtailShape &s) { return ((1u << s.lo_bits) - 1) + ((1u << s.hi_bits) - 1) * s.row_chunks + 1; }

// k_hmarg29: workgroup g < L - 1: column lo = g + 1 — the H buckets of weight hi * L + lo; then (H - 1) * row_chunks workgroups: piece j of row hi — buckets of
// weight hi * L + lo, lo = j * L / row_chunks .. (consecutive); the last workgroup copies bucket NB - 1 into the slot of R_H. A quad takes one or two buckets,
// then the tree.
template <int UNIT>   // (a template only so that the one translation unit that launches it instantiates it)
__global__ void __launch_bounds__(256) k_hmarg29(const Point29Rec *__restrict__ buckets, uint32_t NB, Point29Rec *__restrict__ marg) {
  zk_take_prio(NB);
  __shared__ Point29Rec lds[4];
  const HtailShape sh = htail_shape(NB);
  const uint32_t L = 1u << sh.lo_bits, H = 1u << sh.hi_bits, RC = sh.row_chunks, g = blockIdx.x, q = threadIdx.x >> 2;
  const int k = threadIdx.x & 3;
  if (g == (L - 1) + (H - 1) * RC) {
    if (threadIdx.x < 4) quad29_store(marg + L + H * RC, quad29_load(buckets + NB - 1, k), k);
    return;
  }
  const bool column = g < L - 1;
  const uint32_t rg = g - (L - 1), hi = rg / RC + 1, piece = rg % RC;                 // (rows only)
  const uint32_t count = column ? H : L / RC;                                          // buckets to add up
  const uint32_t first = column ? g + 1 : hi * L + piece * (L / RC);                   // weight of the first one ...
  const uint32_t step = column ? L : 1;                                                // ... and the distance to the next
  QPoint29 acc = quad29_inf();
  const uint32_t nq = blockDim.x >> 2;
  if (q < count) {
    acc = quad29_load(buckets + (first + q * step - 1), k);
#pragma unroll 1
    for (uint32_t j = q + nq; j < count; j += nq) acc = quad29_add(acc, quad29_load(buckets + (first + j * step - 1), k), k);
  }
  acc = block_quad29_tree(acc, lds, min(count, nq));
  if (threadIdx.x < 4) quad29_store(marg + (column ? g + 1 : L + hi * RC + piece), acc, k);
}

// k_hbits29: workgroup s: T_s from the marginal sums — for s < lo_bits the C_lo with bit s of lo set ("i with a one inserted at bit s", i < L / 2), for lo_bits
// <= s < top the pieces of the R_hi with bit s - lo_bits of hi set (i < H / 2, row_chunks pieces each), for s = top the one record R_H. The result leaves the
// 29-bit domain here: every lane converts its coordinate (a product with 2^256 mod p: the lazy 8 x 32-bit form of field.cuh, normalized) and stores its 32
// bytes of res[s] — pinned host memory. A result with ZZ = 0 (mod p) that is not the point at infinity raises the flag of the one-pass path
// (MsmCounters::pad[0]); the last workgroup to finish hands the counters to the host.
template <int UNIT>
__global__ void __launch_bounds__(256) k_hbits29(const Point29Rec *__restrict__ marg, uint32_t NB, XYZZ<Fq> *__restrict__ res, MsmCounters *cnt,
    uint4 *copy_dst) {
  zk_take_prio(NB);
  __shared__ Point29Rec lds[4];
  const HtailShape sh = htail_shape(NB);
  const uint32_t L = 1u << sh.lo_bits, H = 1u << sh.hi_bits, s_ = blockIdx.x, q = threadIdx.x >> 2;
  const int k = threadIdx.x & 3;
  QPoint29 acc = quad29_inf();
  if (s_ == sh.top) {
    if (q == 0) acc = quad29_load(marg + L + H * sh.row_chunks, k);
  } else {
    const bool cols = s_ < sh.lo_bits;
    const uint32_t bit = cols ? s_ : s_ - sh.lo_bits, RC = cols ? 1u : sh.row_chunks, count = ((cols ? L : H) >> 1) * RC;
    auto with_bit = [&](uint32_t i) { return ((i >> bit) << (bit + 1)) | (1u << bit) | (i & ((1u << bit) - 1)); };   // "i with a one inserted at `bit`"
    auto item = [&](uint32_t j) { return cols ? with_bit(j) : L + with_bit(j / RC) * RC + j % RC; };
    if (q < count) {
      acc = quad29_load(marg + item(q), k);
#pragma unroll 1
      for (uint32_t j = q + 64; j < count; j += 64) acc = quad29_add(acc, quad29_load(marg + item(j), k), k);
    }
    acc = block_quad29_tree(acc, lds, min(count, 64u));
  }
  // (ZZ = 0 mod p in something that is not the point at infinity raises the flag)
  if (threadIdx.x < 4) quad29_emit(acc, k, res + s_, cnt, s_);
  // (lane 2 of the quad may have raised the degenerate-sum flag with an atomic of its own: every lane's writes are ordered before the ticket below by this barrier and
  // lane 0's fence, whatever wave the lanes sit in)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    // (the ticket is left at zero: MSMs that share a sort share these counters)
    if (atomicAdd(&cnt->pad[1], 1u) == gridDim.x - 1) {
      __threadfence();
      *copy_dst = *reinterpret_cast<const uint4 *>(cnt);
      cnt->pad[1] = 0;
    }
  }
}

// ---- the G1 witness MSMs (A, L*, B1) on the same arithmetic (round 4)
// ------------------------------------------------------------------------------------------------ k_wacc_lanes29 (msm.cuh) leaves one Point29Rec per lane;
// here the two cooperative stages that follow it, the 29-bit forms of k_wacc_fold and k_wtail:
//   k_wfold29   workgroup b < NB: bucket b = the sum of its lanes' partial sums (lane_off says where they lie); workgroup NB + g: 256 of the ones lanes;
//   k_wtail29 workgroup s < top: S_s = the sum of the NB / 2 buckets whose weight has bit s (a quad each, then the tree); workgroup top: bucket NB - 1 and the
//               unused
//               slots; workgroup top + 1: the sum of the ones' partial sums. Every result leaves the 29-bit domain through the conversion of k_hbits29, a
//               degenerate one
//               (ZZ = 0 mod p) raises the flag that sends the MSM to the general path, the last workgroup to finish hands the counters to the host.
// quad 0 of a workgroup: the result as 8 x 32-bit words
__device__ __forceinline__ void quad29_emit(const QPoint29 &acc, int k, XYZZ<Fq> *slot, MsmCounters *cnt, uint32_t which) {
  Fq out = Fq::zero();
  bool bad = false;
  if (!acc.inf) {
    acc.c.to_words(out.l);
    bad = k == 2 && out.is_zero_lazy();
    out = out.normalize();
  }
  if (bad) atomicOr(&cnt->pad[0], 2u | (1u << (8 + (which < 23u ? which : 23u))));                      // (bit 1: a degenerate sum — bits 8 .. 31: in which result slot(s); bit 0: a sort region overflowed)
  reinterpret_cast<Fq *>(slot)[k] = out;
}
template <int UNIT>
__global__ void __launch_bounds__(256) k_wfold29(const Point29Rec *__restrict__ partial, const uint32_t *__restrict__ lane_off, uint32_t NB,
    uint32_t bucket_lanes, Point29Rec *__restrict__ out) {
  zk_take_prio(NB);
  __shared__ Point29Rec lds[4];
  const uint32_t b = blockIdx.x, q = threadIdx.x >> 2;
  const int k = threadIdx.x & 3;
  uint32_t beg, len;
  if (b < NB) { beg = lane_off[b]; len = lane_off[b + 1] - beg; } else { beg = bucket_lanes + (b - NB) * 256; len = 256; }
  QPoint29 acc = quad29_inf();
  const uint32_t nq = blockDim.x >> 2;
  if (q < len) {
    acc = quad29_load(partial + beg + q, k);
#pragma unroll 1
    for (uint32_t j = q + nq; j < len; j += nq) acc = quad29_add<true>(acc, quad29_load(partial + beg + j, k), k);
  }
  acc = block_quad29_tree<true>(acc, lds, min(len, nq));
  if (threadIdx.x < 4) quad29_store(out + b, acc, k);
}
template <int UNIT>
__global__ void __launch_bounds__(256) k_wtail29(const Point29Rec *__restrict__ buckets, uint32_t NB, const Point29Rec *__restrict__ ones_partial,
    uint32_t n_ones_partial, uint32_t slots, XYZZ<Fq> *__restrict__ res,
                                                 MsmCounters *cnt, uint4 *copy_dst, uint32_t ticket_shift) {
  zk_take_prio(NB);
  __shared__ Point29Rec lds[4];
  const uint32_t q = threadIdx.x >> 2, s_ = blockIdx.x, half = NB >> 1;
  uint32_t top = 0;
  while ((1u << top) < NB) top++;
  const int k = threadIdx.x & 3;
  QPoint29 acc = quad29_inf();
  uint32_t slot = s_;
  if (s_ == top + 1) {                                                                  // the ones
    if (q < n_ones_partial) {
      acc = quad29_load(ones_partial + q, k);
#pragma unroll 1
      for (uint32_t j = q + 64; j < n_ones_partial; j += 64) acc = quad29_add<true>(acc, quad29_load(ones_partial + j, k), k);
    }
    acc = block_quad29_tree<true>(acc, lds, min(n_ones_partial, 64u));
    slot = slots;
  } else if (s_ == top) {                                                               // weight NB: one bucket; slots above `top` hold the point at infinity
    if (q == 0) acc = quad29_load(buckets + NB - 1, k);
    if (threadIdx.x >= 4 && threadIdx.x < 4 * (slots - top)) reinterpret_cast<Fq *>(res + top + (threadIdx.x >> 2))[k] = Fq::zero();
  } else {
    auto bucket_of = [&](uint32_t i) { return (((i >> s_) << (s_ + 1)) | (1u << s_) | (i & ((1u << s_) - 1))) - 1; };
    if (q < half) {
      acc = quad29_load(buckets + bucket_of(q), k);
#pragma unroll 1
      for (uint32_t j = q + 64; j < half; j += 64) acc = quad29_add<true>(acc, quad29_load(buckets + bucket_of(j), k), k);
    }
    acc = block_quad29_tree<true>(acc, lds, min(half, 64u));
  }
  if (threadIdx.x < 4) quad29_emit(acc, k, res + slot, cnt, s_ == top + 1 ? 23u : s_);   // (slot 23 of the flag word: the sum of the ones)
  // (lane 2 of the quad may have raised the degenerate-sum flag with an atomic of its own: every lane's writes are ordered before the ticket below by this barrier and
  // lane 0's fence, whatever wave the lanes sit in)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    // (the ticket is left at zero: MSMs that share a sort share these counters — and may run side by side on two streams: the leader counts in the low half of
    // the word, the follower in the high half, ticket_shift = 0 / 16)
    if (((atomicAdd(&cnt->pad[1], 1u << ticket_shift) >> ticket_shift) & 0xffffu) == gridDim.x - 1) {
      __threadfence();
      *copy_dst = *reinterpret_cast<const uint4 *>(cnt);
      atomicAnd(&cnt->pad[1], ~(0xffffu << ticket_shift));
    }
  }
}

}  // namespace zk
