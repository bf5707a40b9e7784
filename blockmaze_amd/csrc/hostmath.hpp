// Host-side alt_bn128 arithmetic (4 x 64-bit limbs, Montgomery R = 2^256) for the parts of the path that are a few
// hundred dependent operations long and therefore belong on a CPU core, not a GPU lane: the Horner combine of MSM
// window sums, the proof assembly with (r, s) (r1cs_gg_ppzksnark.tcc:487-495), QAP evaluation during key generation
// (r1cs_to_qap.tcc:105-176) and the pairing check of the verifier (alt_bn128_pairing.cpp).
// Byte layout of an element is identical to the device's 8 x 32-bit limbs, so buffers move between the two unchanged.
#pragma once
#include <cstdint>
#include <cstring>
#include "field_params.h"

namespace zk { namespace host {

typedef unsigned __int128 u128;

template <class P>
struct HFp {
  uint64_t l[4];
  static constexpr uint64_t mod(int i) { return (uint64_t)P::MOD[2 * i] | ((uint64_t)P::MOD[2 * i + 1] << 32); }
  static uint64_t inv64() {
    static const uint64_t v = [] {
      uint64_t p = mod(0), x = 1;
      for (int i = 0; i < 6; i++) x *= 2 - p * x;
      return (uint64_t)0 - x;
    }();
    return v;
  }
  static HFp zero() { HFp r; memset(r.l, 0, 32); return r; }
  static HFp one() { HFp r; memcpy(r.l, P::R1, 32); return r; }
  static HFp r2() { HFp r; memcpy(r.l, P::R2, 32); return r; }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const HFp &b) const { return memcmp(l, b.l, 32) == 0; }
  bool operator!=(const HFp &b) const { return !(*this == b); }
  static bool geq_mod(const uint64_t *a) { for (int i = 3; i >= 0; i--) { if (a[i] != mod(i)) return a[i] > mod(i); } return true; }
  static void sub_mod(uint64_t *a) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)a[i] - mod(i) - br;
      a[i] = (uint64_t)d;
      br = (uint64_t)(d >> 64) & 1;
    }
  }
  friend HFp operator+(const HFp &a, const HFp &b) {
    HFp r;
    uint64_t c = 0;
    for (int i = 0; i < 4; i++) {
      u128 s = (u128)a.l[i] + b.l[i] + c;
      r.l[i] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    if (c || geq_mod(r.l)) sub_mod(r.l);
    return r;
  }
  friend HFp operator-(const HFp &a, const HFp &b) {
    HFp r;
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)a.l[i] - b.l[i] - br;
      r.l[i] = (uint64_t)d;
      br = (uint64_t)(d >> 64) & 1;
    }
    if (br) { uint64_t c = 0; for (int i = 0; i < 4; i++) { u128 s = (u128)r.l[i] + mod(i) + c; r.l[i] = (uint64_t)s; c = (uint64_t)(s >> 64); } } return r; }
  HFp neg() const { return is_zero() ? *this : zero() - *this; }
  HFp dbl() const { return *this + *this; }
  friend HFp operator*(const HFp &a, const HFp &b) {
    uint64_t t[5] = {0, 0, 0, 0, 0}; const uint64_t ninv = inv64();
    for (int i = 0; i < 4; i++) {
      u128 c = 0; for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[4] = (uint64_t)c; uint64_t t5 = (uint64_t)(c >> 64);
      uint64_t k = t[0] * ninv; c = (u128)k * mod(0) + t[0]; c >>= 64;
      for (int j = 1; j < 4; j++) { c += (u128)k * mod(j) + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
      c += t[4]; t[3] = (uint64_t)c; t[4] = t5 + (uint64_t)(c >> 64);
    }
    HFp r; memcpy(r.l, t, 32); if (t[4] || geq_mod(r.l)) sub_mod(r.l); return r;
  }
  HFp sqr() const { return *this * *this; }
  HFp to_mont() const { return *this * r2(); }
  HFp from_mont() const { HFp o = zero(); o.l[0] = 1; return *this * o; }
  static HFp from_u64(uint64_t v) { HFp r = zero(); r.l[0] = v; return r.to_mont(); }
  HFp pow(const uint64_t *e, int limbs) const {
    HFp r = one();
    bool found = false;
    for (int i = limbs * 64 - 1; i >= 0; i--) {
      if (found) r = r.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) {
        found = true;
        r = r * *this;
      }
    }
    return r;
  }
  HFp pow_u64(uint64_t e) const { return pow(&e, 1); }
  HFp inv() const {
    uint64_t e[4];
    uint64_t br = 2;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)mod(i) - br;
      e[i] = (uint64_t)d;
      br = (uint64_t)(d >> 64) & 1;
    }
    return pow(e, 4);
  }
  bool canonical_lsb() const { return from_mont().l[0] & 1; }
};
using HFr = HFp<FrParams>;
using HFq = HFp<FqParams>;

// (q+1)/4 square root, q = 3 mod 4 (fp.tcc:724 with s = 1).  false for non-residues.
inline bool fq_sqrt(const HFq &a, HFq &out) {
  uint64_t e[4];
  {
    uint64_t c = 1;
    for (int i = 0; i < 4; i++) {
      u128 s = (u128)HFq::mod(i) + c;
      e[i] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 2) | (i < 3 ? e[i + 1] << 62 : 0);
  }
  HFq x = a.pow(e, 4); if (x.sqr() != a) return false; out = x; return true; }

struct HFq2 {
  HFq c0, c1;
  static HFq2 zero() { return {HFq::zero(), HFq::zero()}; }
  static HFq2 one() { return {HFq::one(), HFq::zero()}; }
  bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  bool operator==(const HFq2 &b) const { return c0 == b.c0 && c1 == b.c1; }
  bool operator!=(const HFq2 &b) const { return !(*this == b); }
  friend HFq2 operator+(const HFq2 &a, const HFq2 &b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  friend HFq2 operator-(const HFq2 &a, const HFq2 &b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  HFq2 neg() const { return {c0.neg(), c1.neg()}; }
  HFq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  friend HFq2 operator*(const HFq2 &a, const HFq2 &b) {
    HFq aA = a.c0 * b.c0, bB = a.c1 * b.c1, s = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {aA - bB, s - aA - bB};
  }
  HFq2 sqr() const { HFq ab = c0 * c1; return {(c0 + c1) * (c0 - c1), ab.dbl()}; }
  HFq2 mul_fq(const HFq &k) const { return {c0 * k, c1 * k}; }
  HFq2 mul_xi() const { HFq n0 = c0.dbl().dbl().dbl() + c0, n1 = c1.dbl().dbl().dbl() + c1; return {n0 - c1, n1 + c0}; }   // (9+u)
  HFq2 inv() const { HFq t = (c0.sqr() + c1.sqr()).inv(); return {c0 * t, (c1 * t).neg()}; }
  HFq2 frob(unsigned p) const { return (p & 1) ? HFq2{c0, c1.neg()} : *this; }
  HFq2 pow(const uint64_t *e, int limbs) const {
    HFq2 r = one();
    bool found = false;
    for (int i = limbs * 64 - 1; i >= 0; i--) {
      if (found) r = r.sqr();
      if ((e[i / 64] >> (i % 64)) & 1) {
        found = true;
        r = r * *this;
      }
    }
    return r;
  }
};

// Jacobian points, formulas as in the reference (alt_bn128_g1.cpp:139-358): add-2007-bl, madd-2007-bl, dbl-2009-l
template <class F>
struct HPoint {
  F X, Y, Z;
  static HPoint inf() { return {F::zero(), F::one(), F::zero()}; }
  bool is_inf() const { return Z.is_zero(); }
  static HPoint from_affine(const F &x, const F &y) { if (x.is_zero() && y.is_zero()) return inf(); return {x, y, F::one()}; }
  HPoint neg() const { return {X, Y.neg(), Z}; }
  HPoint dbl() const { if (is_inf()) return *this; F A = X.sqr(), B = Y.sqr(), C = B.sqr(), D = ((X + B).sqr() - A - C).dbl(), E = A.dbl() + A, Fv = E.sqr();
    HPoint r; r.X = Fv - D.dbl(); r.Y = E * (D - r.X) - C.dbl().dbl().dbl(); r.Z = (Y * Z).dbl(); return r; }
  HPoint add(const HPoint &o) const { if (is_inf()) return o; if (o.is_inf()) return *this;
    F Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr(), U1 = X * Z2Z2, U2 = o.X * Z1Z1, S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
    if (U1 == U2) { if (S1 == S2) return dbl(); return inf(); }
    F H = U2 - U1, I = H.dbl().sqr(), J = H * I, r = (S2 - S1).dbl(), V = U1 * I; HPoint R;
    R.X = r.sqr() - J - V.dbl(); R.Y = r * (V - R.X) - (S1 * J).dbl(); R.Z = ((Z + o.Z).sqr() - Z1Z1 - Z2Z2) * H; return R; }
  HPoint mul(const uint64_t k[4]) const {
    HPoint r = inf();
    bool found = false;
    for (int i = 255; i >= 0; i--) {
      if (found) r = r.dbl();
      if ((k[i / 64] >> (i % 64)) & 1) {
        found = true;
        r = r.add(*this);
      }
    }
    return r;
  }
  void to_affine(F &x, F &y) const { if (is_inf()) { x = F::zero(); y = F::zero(); return; } F zi = Z.inv(), z2 = zi.sqr(); x = X * z2; y = Y * z2 * zi; }
  // from the device's extended-Jacobian (X, Y, ZZ, ZZZ): Z := ZZZ/ZZ is avoided by mapping to (X*ZZ, Y*ZZZ^? ...) — use affine-free identity:
  // (X, Y, ZZ, ZZZ) with ZZ^3 = ZZZ^2 equals Jacobian (X*ZZ^... ) — simplest exact map: Z = ZZZ/ZZ would need an inversion, so scale instead:
  // Jacobian (X', Y', Z') = (X*ZZ, Y*ZZZ, ZZ) since x = X'/Z'^2 = X/ZZ and y = Y'/Z'^3 = Y*ZZZ/ZZ^3 = Y/ZZZ.
  static HPoint from_xyzz(const F &X, const F &Y, const F &ZZ, const F &ZZZ) { if (ZZ.is_zero()) return inf(); return {X * ZZ, Y * ZZZ, ZZ}; }
};
using HG1 = HPoint<HFq>;
using HG2 = HPoint<HFq2>;

} }  // namespace zk::host
