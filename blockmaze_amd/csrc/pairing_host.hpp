// Optimal-ate pairing on alt_bn128 for the verifier (row V1 of SURVEY.md §8a), host side.
//
// Restates FF/algebra/curves/alt_bn128/alt_bn128_pairing.cpp: G2 precomputation :305-366 (doubling / mixed-addition
// steps :242-293 over the bits of 6z+2, then the two Frobenius corrections), miller_loop :368-418,
// final_exponentiation :110-238.  The last chunk of the final exponentiation raises to
// 2z(6z^2+3z+1)(q^4-q^2+1)/r — a fixed multiple of the textbook hard part — so vk.alpha_g1_beta_g2 values written by the
// reference only compare equal if exactly this chain is used.  Tower: Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-(9+u)),
// Fq12 = Fq6[w]/(w^2-v)  (fp6_3over2.tcc, fp12_2over3over2.tcc); Frobenius constants are derived here from xi = 9+u.
#pragma once
#include <vector>
#include "hostmath.hpp"

namespace zk { namespace host {

struct HFq6 {
  HFq2 c0, c1, c2;
  static HFq6 zero() { return {HFq2::zero(), HFq2::zero(), HFq2::zero()}; }
  static HFq6 one() { return {HFq2::one(), HFq2::zero(), HFq2::zero()}; }
  friend HFq6 operator+(const HFq6 &a, const HFq6 &b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
  friend HFq6 operator-(const HFq6 &a, const HFq6 &b) { return {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
  HFq6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
  friend HFq6 operator*(const HFq6 &a, const HFq6 &b) {   // Karatsuba, fp6_3over2.tcc:94-108
    HFq2 aA = a.c0 * b.c0, bB = a.c1 * b.c1, cC = a.c2 * b.c2;
    return {aA + ((a.c1 + a.c2) * (b.c1 + b.c2) - bB - cC).mul_xi(), (a.c0 + a.c1) * (b.c0 + b.c1) - aA - bB + cC.mul_xi(),
        (a.c0 + a.c2) * (b.c0 + b.c2) - aA + bB - cC};
  }
  HFq6 sqr() const { return *this * *this; }
  HFq6 mul_by_v() const { return {c2.mul_xi(), c0, c1}; }                 // Fp12 mul_by_non_residue
  HFq6 mul_fq2(const HFq2 &k) const { return {c0 * k, c1 * k, c2 * k}; }
  HFq6 inv() const {                                                       // fp6_3over2.tcc:128-146
    HFq2 t0 = c0.sqr(), t1 = c1.sqr(), t2 = c2.sqr(), t3 = c0 * c1, t4 = c0 * c2, t5 = c1 * c2, d0 = t0 - t5.mul_xi(), d1 = t2.mul_xi() - t3, d2 = t1 - t4;
    HFq2 t6 = (c0 * d0 + (c2 * d1 + c1 * d2).mul_xi()).inv(); return {t6 * d0, t6 * d1, t6 * d2}; }
  bool operator==(const HFq6 &o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
};

struct FrobeniusTables { HFq2 fq6_c1[6], fq6_c2[6], fq12_c1[12], twist_mul_by_q_x, twist_mul_by_q_y; };
const FrobeniusTables &frobenius_tables();     // xi^((q^i-1)/3), xi^(2(q^i-1)/3), xi^((q^i-1)/6); alt_bn128_init.cpp:160-201
inline HFq6 fq6_frob(const HFq6 &a, unsigned p) {
  const FrobeniusTables &t = frobenius_tables();
  return {a.c0.frob(p), t.fq6_c1[p % 6] * a.c1.frob(p), t.fq6_c2[p % 6] * a.c2.frob(p)};
}

struct HFq12 {
  HFq6 c0, c1;
  static HFq12 one() { return {HFq6::one(), HFq6::zero()}; }
  friend HFq12 operator*(const HFq12 &a, const HFq12 &b) {
    HFq6 aA = a.c0 * b.c0, bB = a.c1 * b.c1;
    return {aA + bB.mul_by_v(), (a.c0 + a.c1) * (b.c0 + b.c1) - aA - bB};
  }
  HFq12 sqr() const { return *this * *this; }
  HFq12 inv() const { HFq6 t = (c0.sqr() - c1.sqr().mul_by_v()).inv(); return {c0 * t, (c1 * t).neg()}; }
  HFq12 conj() const { return {c0, c1.neg()}; }                            // unitary_inverse
  HFq12 frob(unsigned p) const { const FrobeniusTables &t = frobenius_tables(); return {fq6_frob(c0, p), fq6_frob(c1, p).mul_fq2(t.fq12_c1[p % 12])}; }
  HFq12 cyclo_exp(uint64_t e) const {
    HFq12 r = one();
    bool found = false;
    for (int i = 63; i >= 0; i--) {
      if (found) r = r.sqr();
      if ((e >> i) & 1) {
        found = true;
        r = r * *this;
      }
    }
    return r;
  }
  HFq12 mul_by_024(const HFq2 &ell_0, const HFq2 &ell_VW, const HFq2 &ell_VV) const {
    HFq12 s{{ell_0, HFq2::zero(), ell_VV}, {HFq2::zero(), ell_VW, HFq2::zero()}};
    return *this * s;
  }
  bool operator==(const HFq12 &o) const { return c0 == o.c0 && c1 == o.c1; }
};

struct EllCoeffs { HFq2 ell_0, ell_VW, ell_VV; };
typedef std::vector<EllCoeffs> G2Precomp;
G2Precomp precompute_g2(const HFq2 &qx, const HFq2 &qy);                                   // affine Q
HFq12 miller_loop(const HFq &px, const HFq &py, const G2Precomp &q);                      // affine P
HFq12 final_exponentiation(const HFq12 &f);
inline HFq12 reduced_pairing(const HFq &px, const HFq &py, const HFq2 &qx, const HFq2 &qy) {
  return final_exponentiation(miller_loop(px, py, precompute_g2(qx, qy)));
}
bool g1_on_curve(const HFq &x, const HFq &y);
bool g2_on_curve(const HFq2 &x, const HFq2 &y);

} }  // namespace zk::host
