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
  static ZK_NI XYZZ dbl_affine(const Affine<F> &p) {
    if (p.is_inf()) return inf();
    F U = p.y.dbl(), V = U.sqr(), W = U * V, S = p.x * V, X2 = p.x.sqr(), M = X2.dbl() + X2;
    XYZZ r; r.X = M.sqr() - S.dbl(); r.Y = M * (S - r.X) - W * p.y; r.ZZ = V; r.ZZZ = W; return r;
  }
  // general doubling (EFD dbl-2008-s-1, a = 0)
  ZK_NI XYZZ dbl() const {
    if (is_inf()) return *this;
    F U = Y.dbl(), V = U.sqr(), W = U * V, S = X * V, X2 = X.sqr(), M = X2.dbl() + X2;
    XYZZ r; r.X = M.sqr() - S.dbl(); r.Y = M * (S - r.X) - W * Y; r.ZZ = V * ZZ; r.ZZZ = W * ZZZ; return r;
  }
  // mixed addition acc += p (EFD madd-2008-s), complete: handles acc = inf, p = inf, p = +-acc.  The *_inl forms are for the hot loops, where the
  // accumulator has to stay in VGPRs: an out-of-line call passes `this` through scratch memory (measured: 2.5 GB of scratch traffic per H accumulation)
  ZK_NI void madd(const Affine<F> &p) { madd_inl(p); }
  ZK_HD void madd_inl(const Affine<F> &p) {
    if (p.is_inf()) return;
    if (is_inf()) { X = p.x; Y = p.y; ZZ = F::one(); ZZZ = F::one(); return; }
    F U2 = p.x * ZZ, S2 = p.y * ZZZ, Pv = U2 - X, Rv = S2 - Y;
    if (Pv.is_zero()) { if (Rv.is_zero()) *this = dbl_affine(p); else *this = inf(); return; }
    F PP = Pv.sqr(), PPP = Pv * PP, Q = X * PP;
    F X3 = Rv.sqr() - PPP - Q.dbl();
    Y = Rv * (Q - X3) - Y * PPP; X = X3; ZZ = ZZ * PP; ZZZ = ZZZ * PPP;
  }
  // general addition acc += o (EFD add-2008-s), complete
  ZK_NI void add(const XYZZ &o) { add_inl(o); }
  ZK_HD void add_inl(const XYZZ &o) {
    if (o.is_inf()) return;
    if (is_inf()) { *this = o; return; }
    F U1 = X * o.ZZ, U2 = o.X * ZZ, S1 = Y * o.ZZZ, S2 = o.Y * ZZZ, Pv = U2 - U1, Rv = S2 - S1;
    if (Pv.is_zero()) { if (Rv.is_zero()) *this = dbl(); else *this = inf(); return; }
    F PP = Pv.sqr(), PPP = Pv * PP, Q = U1 * PP;
    F X3 = Rv.sqr() - PPP - Q.dbl();
    Y = Rv * (Q - X3) - S1 * PPP; X = X3; ZZ = ZZ * o.ZZ * PP; ZZZ = ZZZ * o.ZZZ * PPP;
  }
  // k * this for a small scalar (double-and-add, MSB first) — used for segment offsets in the bucket reduction
  ZK_NI XYZZ mul_small(uint32_t k) const {
    XYZZ r = inf(); bool found = false;
    for (int i = 31; i >= 0; i--) { if (found) r = r.dbl(); if ((k >> i) & 1) { found = true; r.add(*this); } }
    return r;
  }
};

using G1Affine = Affine<Fq>;
using G2Affine = Affine<Fq2>;
using G1XYZZ = XYZZ<Fq>;
using G2XYZZ = XYZZ<Fq2>;

}  // namespace zk
