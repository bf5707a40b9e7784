// Prime-field arithmetic for alt_bn128's Fr / Fq on CDNA4 VALU (and on the host, for the same code path in tests of
// small host-side steps).  Elements are 8 x 32-bit limbs in Montgomery form with R = 2^256 — the reference's
// representation (FF/algebra/fields/fp.hpp, fp.tcc:23-190 mul_reduce; :310-520 add/sub), re-limbed for a 32-bit ALU:
// gfx950 multiplies 32x32->64 with a 64-bit addend in one VALU op (v_mad_u64_u32), so the product rows are built from
// that primitive.  Both moduli are < 2^254, which leaves two spare bits: sums of two reduced values never overflow
// 256 bits and the Montgomery accumulator never needs a ninth limb.
#pragma once
#include <cstdint>
#include "field_params.h"

#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_NI __host__ __device__ __noinline__      // long, cold or many-call-site routines: one out-of-line copy keeps
#else                                               // code size (and hipcc's compile time) bounded
#define ZK_HD inline
#define ZK_NI
#endif

namespace zk {


template <class P>
struct Fp {
  uint32_t l[8];

  static ZK_HD Fp zero() { Fp r; for (int i = 0; i < 8; i++) r.l[i] = 0; return r; }
  static ZK_HD Fp one() { Fp r; for (int i = 0; i < 8; i++) r.l[i] = P::R1[i]; return r; }
  static ZK_HD Fp r2() { Fp r; for (int i = 0; i < 8; i++) r.l[i] = P::R2[i]; return r; }
  static ZK_HD uint32_t modulus_limb0() { return P::MOD[0]; }
  static ZK_HD uint32_t modulus_limb(int i) { return P::MOD[i]; }

  ZK_HD bool is_zero() const { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= l[i]; return o == 0; }
  ZK_HD bool operator==(const Fp &b) const { uint32_t o = 0; for (int i = 0; i < 8; i++) o |= l[i] ^ b.l[i]; return o == 0; }
  ZK_HD bool operator!=(const Fp &b) const { return !(*this == b); }

  // Device code: every carry chain is one v_addc / v_subb per limb (field_mul_gfx950.inc, generated); hipcc lowers the limb loops of the host versions below to
  // 64-bit adds and sign extensions, about five instructions per limb (measured in the H-query accumulation: 1,100 of 4,215 instructions per mixed addition
  // were such code).
#if defined(__HIP_DEVICE_COMPILE__)
#include "field_mul_gfx950.inc"   // mul_raw / sqr_raw / add_raw / sub_fix / cond_sub / cond_neg / neg_masked (gen_field_mul.py)
  static __device__ __forceinline__ Fp reduce_once(const Fp &a) { Fp r = a; cond_sub<1>(r); return r; }          // r = a - p if a >= p (a < 2p)
  friend __device__ __forceinline__ Fp operator+(const Fp &a, const Fp &b) { Fp s = a; add_raw(s, b); cond_sub<1>(s); return s; }   // no carry out: 2p < 2^256
  friend __device__ __forceinline__ Fp operator-(const Fp &a, const Fp &b) { Fp d = a; sub_fix<1>(d, b); return d; }
  friend __device__ __forceinline__ Fp operator*(const Fp &a, const Fp &b) { Fp r = mul_raw(a, b); cond_sub<1>(r); return r; }
  // dedicated squaring: 36 limb products instead of 64 (fp.tcc:594 squared())
  __device__ __forceinline__ Fp sqr() const {
    Fp r = sqr_raw(*this);
    cond_sub<1>(r);
    return r;
  }
  // ---- the lazy domain: values in [0, 2p)
  // ---------------------------------------------------------------------------------------------------------------------- Both moduli leave two spare bits (p
  // < 2^254). A Montgomery product of a, b < 2p is (ab + mp)/R < (4p^2 + Rp)/R < 2p because R = 2^256 > 4p, so products need no final subtraction when they
  // feed further products; a difference stays in [0, 2p) when 2p is added after a borrow. normalize() brings a value back to [0, p). The 29-bit kernels of
  // msm.cuh hand their results over in this domain (k_hacc_combine29); round 3's first version of the H accumulation lived in it (tools/mul_probe.hip compares
  // the two product forms).
  static __device__ __forceinline__ Fp mul_lazy(const Fp &a, const Fp &b) { return mul_raw(a, b); }
  static __device__ __forceinline__ Fp sqr_lazy(const Fp &a) { return sqr_raw(a); }
  static __device__ __forceinline__ Fp sub_lazy(const Fp &a, const Fp &b) { Fp d = a; sub_fix<2>(d, b); return d; }
  __device__ __forceinline__ Fp normalize() const { return reduce_once(*this); }
  // 0 or p
  __device__ __forceinline__ bool is_zero_lazy() const {
    uint32_t o = 0, q = 0;
    for (int i = 0; i < 8; i++) {
      o |= l[i];
      q |= l[i] ^ P::MOD[i];
    }
    return o == 0 || q == 0;
  }
#else
  // r = a - MOD if a >= MOD (a < 2*MOD)
  static ZK_HD Fp reduce_once(const Fp &a) {
    Fp d; uint64_t br = 0;
    for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)a.l[i] - P::MOD[i] - br; d.l[i] = (uint32_t)t; br = (t >> 32) & 1; }
    Fp r;
    for (int i = 0; i < 8; i++) r.l[i] = br ? a.l[i] : d.l[i];
    return r;
  }
  friend ZK_HD Fp operator+(const Fp &a, const Fp &b) {
    Fp s; uint64_t c = 0;
    for (int i = 0; i < 8; i++) { c += (uint64_t)a.l[i] + b.l[i]; s.l[i] = (uint32_t)c; c >>= 32; }
    return reduce_once(s);   // no carry out: 2*MOD < 2^256
  }
  friend ZK_HD Fp operator-(const Fp &a, const Fp &b) {
    Fp d; uint64_t br = 0;
    for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)a.l[i] - b.l[i] - br; d.l[i] = (uint32_t)t; br = (t >> 32) & 1; }
    uint32_t mask = (uint32_t)0 - (uint32_t)br; uint64_t c = 0;
    for (int i = 0; i < 8; i++) { c += (uint64_t)d.l[i] + (P::MOD[i] & mask); d.l[i] = (uint32_t)c; c >>= 32; }
    return d;
  }
  // Montgomery product a*b/R mod p, host reference of the device's column product: coarsely integrated operand scanning, one row of a*b_i followed by one
  // reduction row m*p.
  friend ZK_HD Fp operator*(const Fp &a, const Fp &b) {
    uint32_t t[8];
    for (int j = 0; j < 8; j++) t[j] = 0;
    for (int i = 0; i < 8; i++) {
      uint64_t c = 0;
      for (int j = 0; j < 8; j++) { uint64_t s = (uint64_t)a.l[j] * b.l[i] + t[j] + c; t[j] = (uint32_t)s; c = s >> 32; }
      uint32_t t8 = (uint32_t)c;
      uint32_t m = t[0] * P::INV;
      uint64_t s = (uint64_t)m * P::MOD[0] + t[0]; c = s >> 32;
      for (int j = 1; j < 8; j++) { s = (uint64_t)m * P::MOD[j] + t[j] + c; t[j - 1] = (uint32_t)s; c = s >> 32; }
      t[7] = t8 + (uint32_t)c;   // < 2^32 because the running value stays < 2p < 2^255
    }
    Fp r;
    for (int j = 0; j < 8; j++) r.l[j] = t[j];
    return reduce_once(r);
  }
  ZK_HD Fp sqr() const { return (*this) * (*this); }
  // (host pass of hipcc and plain g++: the lazy-domain entry points exist so that kernel bodies parse; canonical values are lazy values)
  static ZK_HD Fp mul_lazy(const Fp &a, const Fp &b) { return a * b; }
  static ZK_HD Fp sqr_lazy(const Fp &a) { return a * a; }
  static ZK_HD Fp sub_lazy(const Fp &a, const Fp &b) { return a - b; }
  static ZK_HD void neg_masked(Fp &a, uint32_t m) { if (m) a = zero() - a; }
  static ZK_HD void add_raw(Fp &a, const Fp &b) { a = a + b; }
  ZK_HD Fp normalize() const { return *this; }
  ZK_HD bool is_zero_lazy() const { return is_zero(); }
#endif
  // 0 - 0 borrows nothing, so zero stays zero; (a `cond ? *this : ...` here makes the compiler select between two memory copies and pins the operand in
  // scratch)
  ZK_HD Fp neg() const {
    return zero() - *this;
  }
  ZK_HD Fp dbl() const { return *this + *this; }

  ZK_HD Fp to_mont() const { return (*this) * r2(); }              // canonical -> Montgomery
  ZK_HD Fp from_mont() const { Fp o = zero(); o.l[0] = 1; return (*this) * o; }   // Montgomery -> canonical

  static ZK_NI Fp mul_outofline(const Fp &a, const Fp &b) { return a * b; }
  // a^e, e given as 8 x 32-bit limbs (plain integer), MSB-first square-and-multiply
  ZK_NI Fp pow(const uint32_t e[8]) const {
    Fp r = one(); bool found = false;
    for (int i = 255; i >= 0; i--) { if (found) r = mul_outofline(r, r); if ((e[i >> 5] >> (i & 31)) & 1) { found = true; r = mul_outofline(r, *this); } }
    return r;
  }
  ZK_HD Fp pow_u64(uint64_t e) const { uint32_t ee[8] = {(uint32_t)e, (uint32_t)(e >> 32), 0, 0, 0, 0, 0, 0}; return pow(ee); }
  // Fermat inverse (the reference uses mpn_gcdext, fp.tcc:688; the value is the same)
  ZK_NI Fp inv() const {
    uint32_t e[8]; uint64_t br = 2;
    for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)P::MOD[i] - br; e[i] = (uint32_t)t; br = (t >> 32) & 1; }
    return pow(e);
  }
  static ZK_HD Fp from_u64(uint64_t v) { Fp r = zero(); r.l[0] = (uint32_t)v; r.l[1] = (uint32_t)(v >> 32); return r.to_mont(); }
};

using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

// Fq2 = Fq[u]/(u^2 + 1)  (FF/algebra/fields/fp2.tcc:73-135; non-residue -1, alt_bn128_init.cpp:152)
struct Fq2 {
  Fq c0, c1;
  static ZK_HD Fq2 zero() { return {Fq::zero(), Fq::zero()}; }
  static ZK_HD Fq2 one() { return {Fq::one(), Fq::zero()}; }
  ZK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
  ZK_HD bool operator==(const Fq2 &b) const { return c0 == b.c0 && c1 == b.c1; }
  ZK_HD bool operator!=(const Fq2 &b) const { return !(*this == b); }
  friend ZK_HD Fq2 operator+(const Fq2 &a, const Fq2 &b) { return {a.c0 + b.c0, a.c1 + b.c1}; }
  friend ZK_HD Fq2 operator-(const Fq2 &a, const Fq2 &b) { return {a.c0 - b.c0, a.c1 - b.c1}; }
  ZK_HD Fq2 neg() const { return {c0.neg(), c1.neg()}; }
  ZK_HD Fq2 dbl() const { return {c0.dbl(), c1.dbl()}; }
  // Karatsuba: 3 base-field products
  friend ZK_HD Fq2 operator*(const Fq2 &a, const Fq2 &b) {
    Fq aA = a.c0 * b.c0, bB = a.c1 * b.c1, s = (a.c0 + a.c1) * (b.c0 + b.c1);
    return {aA - bB, s - aA - bB};
  }
  // complex squaring: 2 base-field products
  ZK_HD Fq2 sqr() const { Fq ab = c0 * c1; return {(c0 + c1) * (c0 - c1), ab.dbl()}; }
  ZK_HD Fq2 mul_fq(const Fq &k) const { return {c0 * k, c1 * k}; }
  // times xi = 9 + u, the Fq6 non-residue (alt_bn128_init.cpp:158)
  ZK_HD Fq2 mul_xi() const {
    Fq a = c0.dbl().dbl().dbl() + c0, b = c1.dbl().dbl().dbl() + c1;
    return {a - c1, b + c0};
  }
  ZK_HD Fq2 frob(unsigned p) const { return (p & 1) ? Fq2{c0, c1.neg()} : *this; }                                          // x -> x^(q^p)
  ZK_HD Fq2 inv() const { Fq t = (c0.sqr() + c1.sqr()).inv(); return {c0 * t, (c1 * t).neg()}; }
};

}  // namespace zk
