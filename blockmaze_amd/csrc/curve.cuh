// alt_bn128 G1 (over Fq) and G2 (over Fq2) group law for the MSM kernels.
//
// The reference works in Jacobian coordinates (FF/algebra/curves/alt_bn128/alt_bn128_g1.cpp:139-358,
// alt_bn128_g2.cpp:149-365).  A group element has a unique affine form, so any complete set of formulas gives the same
// proof bytes; the kernels use extended Jacobian (X, Y, ZZ, ZZZ) with x = X/ZZ, y = Y/ZZZ because its mixed addition is
// the cheapest for bucket accumulation (8M + 2S, EFD madd-2008-s) and both curves have a = 0.
// Key points live in HBM as affine Montgomery pairs; the point at infinity is the all-zero pair ((0,0) is on neither
// curve: y^2 = x^3 + 3, resp. y^2 = x^3 + 3/(9+u)).
#pragma once
#include "field.cuh"

namespace zk {

template <class F>
struct Affine {
  F x, y;
  ZK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
  static ZK_HD Affine inf() { return {F::zero(), F::zero()}; }
  ZK_HD Affine neg() const { return {x, y.neg()}; }
};

template <class F>
struct XYZZ {
  F X, Y, ZZ, ZZZ;
  static ZK_HD XYZZ inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
  ZK_HD bool is_inf() const { return ZZ.is_zero(); }
  static ZK_HD XYZZ from_affine(const Affine<F> &p) { if (p.is_inf()) return inf(); return {p.x, p.y, F::one(), F::one()}; }
  ZK_HD XYZZ neg() const { return {X, Y.neg(), ZZ, ZZZ}; }

  // doubling of an affine point (EFD mdbl-2008-s-1)
  static ZK_HD XYZZ dbl_affine_inl(const Affine<F> &p) {      // (an affine point at infinity is (0,0): U = 0 gives ZZ = 0, i.e. infinity, without a test)
    F U = p.y.dbl(), V = U.sqr(), W = U * V, S = p.x * V, X2 = p.x.sqr(), M = X2.dbl() + X2;
    XYZZ r; r.X = M.sqr() - S.dbl(); r.Y = M * (S - r.X) - W * p.y; r.ZZ = V; r.ZZZ = W; return r;
  }
  // general doubling (EFD dbl-2008-s-1, a = 0)
  ZK_HD XYZZ dbl_inl() const {                                 // (ZZ = 0 gives ZZ3 = V*ZZ = 0: infinity stays infinity without a test)
    F U = Y.dbl(), V = U.sqr(), W = U * V, S = X * V, X2 = X.sqr(), M = X2.dbl() + X2;
    XYZZ r; r.X = M.sqr() - S.dbl(); r.Y = M * (S - r.X) - W * Y; r.ZZ = V * ZZ; r.ZZZ = W * ZZZ; return r;
  }
  // mixed addition acc += p (EFD madd-2008-s), complete: handles acc = inf, p = inf, p = +-acc.  Everything here is inlined into the kernels: an out-of-line
  // call passes `this` through scratch memory (measured: 2.5 GB of scratch traffic per H accumulation), and hipcc 7.2 was seen to drop the save/restore of
  // callee-saved SGPRs in large out-of-line functions holding the inline-asm field product (the caller's second call then jumped to a clobbered address)
  ZK_HD void madd_inl(const Affine<F> &p) {
    if (p.is_inf()) return;
    if (is_inf()) { X = p.x; Y = p.y; ZZ = F::one(); ZZZ = F::one(); return; }
    F U2 = p.x * ZZ, S2 = p.y * ZZZ, Pv = U2 - X, Rv = S2 - Y;
    // (inlined: an out-of-line call takes p by address, which pins it in scratch memory for every iteration of the caller's loop)
    if (Pv.is_zero()) {
      if (Rv.is_zero()) *this = dbl_affine_inl(p);
      else *this = inf();
      return;
    }
    F PP = Pv.sqr(), PPP = Pv * PP, Q = X * PP;
    F X3 = Rv.sqr() - PPP - Q.dbl();
    Y = Rv * (Q - X3) - Y * PPP; X = X3; ZZ = ZZ * PP; ZZZ = ZZZ * PPP;
  }
  // general addition acc += o (EFD add-2008-s), complete
  ZK_HD void add_inl(const XYZZ &o) {
    if (o.is_inf()) return;
    if (is_inf()) { *this = o; return; }
    F U1 = X * o.ZZ, U2 = o.X * ZZ, S1 = Y * o.ZZZ, S2 = o.Y * ZZZ, Pv = U2 - U1, Rv = S2 - S1;
    if (Pv.is_zero()) { if (Rv.is_zero()) *this = dbl_inl(); else *this = inf(); return; }
    F PP = Pv.sqr(), PPP = Pv * PP, Q = U1 * PP;
    F X3 = Rv.sqr() - PPP - Q.dbl();
    Y = Rv * (Q - X3) - S1 * PPP; X = X3; ZZ = ZZ * o.ZZ * PP; ZZZ = ZZZ * o.ZZZ * PPP;
  }
  // k * this for a small scalar (double-and-add, MSB first) — used for segment offsets in the bucket reduction
  ZK_HD XYZZ mul_small(uint32_t k) const {
    XYZZ r = inf(); bool found = false;
    for (int i = 31; i >= 0; i--) { if (found) r = r.dbl_inl(); if ((k >> i) & 1) { found = true; r.add_inl(*this); } }
    return r;
  }
};

#if defined(__HIPCC__)
// ---- quad-cooperative group law ------------------------------------------------------------------------------------------
// The tails of an MSM (bucket reduction, tree sums) are chains of dependent additions on few points: one lane per point leaves the chip idle and each
// addition costs 14 sequential field multiplications (a single wave issues one in ~0.9 us).  Here FOUR adjacent lanes (a DPP quad) hold the same point
// and share an addition: every lane computes one of the products of a round, quad_perm broadcasts hand the four products back to all lanes, and the cheap
// additions/subtractions are done redundantly.  add = 4 rounds (14 products), dbl = 3 rounds (9 products).  All data-dependent branches are uniform within
// a quad because the state is replicated.  k = lane & 3.
template <int K, class P> __device__ __forceinline__ Fp<P> quad_pick(const Fp<P> &v) {
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.l[i], K * 0x55, 0xf, 0xf, false);   // quad_perm:[K,K,K,K]
  return r;
}
template <int K> __device__ __forceinline__ Fq2 quad_pick(const Fq2 &v) { return {quad_pick<K>(v.c0), quad_pick<K>(v.c1)}; }
template <class P> __device__ __forceinline__ Fp<P> lane_sel(bool c, const Fp<P> &a, const Fp<P> &b) {
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = c ? a.l[i] : b.l[i];
  return r;
}
__device__ __forceinline__ Fq2 lane_sel(bool c, const Fq2 &a, const Fq2 &b) { return {lane_sel(c, a.c0, b.c0), lane_sel(c, a.c1, b.c1)}; }
template <class F> __device__ __forceinline__ F quad_sel(int k, const F &a, const F &b, const F &c, const F &d) {
  return lane_sel(k < 2, lane_sel(k == 0, a, b), lane_sel(k == 2, c, d));
}

template <class F> __device__ __forceinline__ XYZZ<F> xyzz_sel(bool c, const XYZZ<F> &a, const XYZZ<F> &b) {
  return {lane_sel(c, a.X, b.X), lane_sel(c, a.Y, b.Y), lane_sel(c, a.ZZ, b.ZZ), lane_sel(c, a.ZZZ, b.ZZZ)};
}
// The special cases (an operand at infinity, equal or opposite operands) are resolved with limb-wise selects AFTER the general formulas have run on
// whatever the inputs were: an early `return a` / `return b` makes the compiler park both points in scratch memory and select between the two copies
// by address, which put a memory round trip into every addition of the dependent chain.

// dbl-2008-s-1 (a = 0):  round 1: V = U^2 | X^2      round 2: W = U*V | S = X*V | M^2 | V*ZZ      round 3: M*(S - X3) | W*Y | W*ZZZ
// (the point at infinity needs no special case: ZZ = 0 gives ZZ3 = V*ZZ = 0)
template <class F> __device__ __forceinline__ XYZZ<F> quad_dbl_inl(const XYZZ<F> &a, int k) {
  F U = a.Y.dbl(), m = lane_sel(k == 0, U, a.X); m = m * m;
  F V = quad_pick<0>(m), X2 = quad_pick<1>(m), M = X2.dbl() + X2;
  m = quad_sel(k, U, a.X, M, V) * quad_sel(k, V, V, M, a.ZZ);
  F W = quad_pick<0>(m), S = quad_pick<1>(m), MM = quad_pick<2>(m); XYZZ<F> r; r.ZZ = quad_pick<3>(m); r.X = MM - S.dbl();
  m = quad_sel(k, M, W, W, W) * quad_sel(k, S - r.X, a.Y, a.ZZZ, a.ZZZ);
  r.Y = quad_pick<0>(m) - quad_pick<1>(m); r.ZZZ = quad_pick<2>(m); return r;
}
// add-2008-s:  round 1: U1 = X1*ZZ2 | U2 = X2*ZZ1 | S1 = Y1*ZZZ2 | S2 = Y2*ZZZ1      round 2: P^2 | R^2 | ZZ1*ZZ2 | ZZZ1*ZZZ2
//              round 3: P*PP | U1*PP | ZZ12*PP      round 4: R*(Q - X3) | S1*PPP | ZZZ12*PPP
template <class F> __device__ __forceinline__ XYZZ<F> quad_add(const XYZZ<F> &a, const XYZZ<F> &b, int k) {
  const bool a_inf = a.is_inf(), b_inf = b.is_inf();
  F m = quad_sel(k, a.X, b.X, a.Y, b.Y) * quad_sel(k, b.ZZ, a.ZZ, b.ZZZ, a.ZZZ);
  F U1 = quad_pick<0>(m), S1 = quad_pick<2>(m), Pv = quad_pick<1>(m) - U1, Rv = quad_pick<3>(m) - S1;
  m = quad_sel(k, Pv, Rv, a.ZZ, a.ZZZ) * quad_sel(k, Pv, Rv, b.ZZ, b.ZZZ);
  F PP = quad_pick<0>(m), RR = quad_pick<1>(m), ZZ12 = quad_pick<2>(m), ZZZ12 = quad_pick<3>(m);
  m = quad_sel(k, Pv, U1, ZZ12, ZZ12) * PP;
  F PPP = quad_pick<0>(m), Q = quad_pick<1>(m); XYZZ<F> r; r.ZZ = quad_pick<2>(m); r.X = RR - PPP - Q.dbl();
  m = quad_sel(k, Rv, S1, ZZZ12, ZZZ12) * lane_sel(k == 0, Q - r.X, PPP);
  r.Y = quad_pick<0>(m) - quad_pick<1>(m); r.ZZZ = quad_pick<2>(m);
  // b = +-a: rare, uniform within the quad; inlined so that `a` never has its address taken
  if (Pv.is_zero() && !a_inf && !b_inf) {
    if (Rv.is_zero()) r = quad_dbl_inl(a, k);
    else r = XYZZ<F>::inf();
  }
  return xyzz_sel(b_inf, a, xyzz_sel(a_inf, b, r));
}
// madd-2008-s (affine operand): round 1: U2 = X2*ZZ1 | S2 = Y2*ZZZ1 round 2: P^2 | R^2 round 3: P*PP | X1*PP | ZZ1*PP round 4: R*(Q - X3) | Y1*PPP | ZZZ1*PPP
template <class F> __device__ __forceinline__ XYZZ<F> quad_madd(const XYZZ<F> &a, const Affine<F> &p, int k) {
  const bool a_inf = a.is_inf(), p_inf = p.is_inf();
  F m = lane_sel(k == 0, p.x, p.y) * lane_sel(k == 0, a.ZZ, a.ZZZ);
  F Pv = quad_pick<0>(m) - a.X, Rv = quad_pick<1>(m) - a.Y;
  m = lane_sel(k == 0, Pv, Rv); m = m * m;
  F PP = quad_pick<0>(m), RR = quad_pick<1>(m);
  m = quad_sel(k, Pv, a.X, a.ZZ, a.ZZ) * PP;
  F PPP = quad_pick<0>(m), Q = quad_pick<1>(m); XYZZ<F> r; r.ZZ = quad_pick<2>(m); r.X = RR - PPP - Q.dbl();
  m = quad_sel(k, Rv, a.Y, a.ZZZ, a.ZZZ) * lane_sel(k == 0, Q - r.X, PPP);
  r.Y = quad_pick<0>(m) - quad_pick<1>(m); r.ZZZ = quad_pick<2>(m);
  if (Pv.is_zero() && !a_inf && !p_inf) { if (Rv.is_zero()) r = quad_dbl_inl(a, k); else r = XYZZ<F>::inf(); }
  XYZZ<F> lifted = {p.x, p.y, F::one(), F::one()};
  return xyzz_sel(p_inf, a, xyzz_sel(a_inf, lifted, r));
}
// n * a for a small scalar, MSB first
template <class F> __device__ __forceinline__ XYZZ<F> quad_mul_small(const XYZZ<F> &a, uint32_t n, int k) {
  XYZZ<F> r = XYZZ<F>::inf(); if (!n) return r;
#pragma unroll 1
  for (int i = 31 - __clz(n); i >= 0; i--) { r = quad_dbl_inl(r, k); if ((n >> i) & 1) r = quad_add(r, a, k); }
  return r;
}
// n * p for an affine point and a small scalar n >= 1, MSB first: the top bit lifts p, every further bit costs a doubling and, if set, a mixed addition
template <class F> __device__ __forceinline__ XYZZ<F> quad_mul_small_affine(const Affine<F> &p, uint32_t n, int k) {
  XYZZ<F> r = XYZZ<F>::from_affine(p);
#pragma unroll 1
  for (int i = 30 - __clz(n); i >= 0; i--) { r = quad_dbl_inl(r, k); if ((n >> i) & 1) r = quad_madd(r, p, k); }
  return r;
}
#endif

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;

}  // namespace zk
