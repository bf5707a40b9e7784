// see pairing_host.hpp
#include "pairing_host.hpp"

namespace zk { namespace host {

static const uint64_t BN_Z = 4965661367192848881ull;                         // alt_bn128_init.cpp:327 (final_exponent_z)
static const uint64_t ATE_LOOP[2] = {0x9d797039be763ba8ull, 0x1ull};          // 6z+2 = 29793968203157093288 (alt_bn128_init.cpp:324)

const FrobeniusTables &frobenius_tables() {
  static const FrobeniusTables T = [] {
    FrobeniusTables t; HFq2 xi{HFq::from_u64(9), HFq::one()};
    // (q-1)/6 as an integer: q = 1 mod 6
    uint64_t e[4]; { uint64_t br = 1; for (int i = 0; i < 4; i++) { u128 d = (u128)HFq::mod(i) - br; e[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
                     u128 rem = 0; for (int i = 3; i >= 0; i--) { u128 cur = (rem << 64) | e[i]; e[i] = (uint64_t)(cur / 6); rem = cur % 6; } }
    HFq2 g1 = xi.pow(e, 4);                                                  // xi^((q-1)/6)
    t.fq12_c1[0] = HFq2::one(); for (int i = 1; i < 12; i++) t.fq12_c1[i] = t.fq12_c1[i - 1] * g1.frob(i - 1);
    for (int i = 0; i < 6; i++) { t.fq6_c1[i] = t.fq12_c1[i].sqr(); t.fq6_c2[i] = t.fq6_c1[i].sqr(); }
    t.twist_mul_by_q_x = t.fq6_c1[1]; t.twist_mul_by_q_y = t.fq12_c1[1] * t.fq12_c1[1].sqr(); return t; }();
  return T;
}
static const HFq2 &twist_b() { static const HFq2 b = HFq2{HFq::from_u64(3), HFq::zero()} * HFq2{HFq::from_u64(9), HFq::one()}.inv(); return b; }
bool g1_on_curve(const HFq &x, const HFq &y) { if (x.is_zero() && y.is_zero()) return true; return y.sqr() == x.sqr() * x + HFq::from_u64(3); }
bool g2_on_curve(const HFq2 &x, const HFq2 &y) { if (x.is_zero() && y.is_zero()) return true; return y.sqr() == x.sqr() * x + twist_b(); }

struct G2Proj { HFq2 X, Y, Z; };
static void doubling_step(const HFq &two_inv, G2Proj &c, EllCoeffs &o) {      // alt_bn128_pairing.cpp:242-268
  HFq2 A = (c.X * c.Y).mul_fq(two_inv), B = c.Y.sqr(), C = c.Z.sqr(), D = C + C + C, E = twist_b() * D, F = E + E + E, G = (B + F).mul_fq(two_inv),
      H = (c.Y + c.Z).sqr() - (B + C), I = E - B, J = c.X.sqr(), E2 = E.sqr();
  c.X = A * (B - F); c.Y = G.sqr() - (E2 + E2 + E2); c.Z = B * H; o.ell_0 = I.mul_xi(); o.ell_VW = H.neg(); o.ell_VV = J + J + J; }
static void mixed_addition_step(const HFq2 &x2, const HFq2 &y2, G2Proj &c, EllCoeffs &o) {   // :270-293
  HFq2 D = c.X - x2 * c.Z, E = c.Y - y2 * c.Z, F = D.sqr(), G = E.sqr(), H = D * F, I = c.X * F, J = H + c.Z * G - (I + I);
  HFq2 Y1 = c.Y; c.X = D * J; c.Y = E * (I - J) - H * Y1; c.Z = c.Z * H; o.ell_0 = (E * x2 - D * y2).mul_xi(); o.ell_VV = E.neg(); o.ell_VW = D; }
G2Precomp precompute_g2(const HFq2 &qx, const HFq2 &qy) {                    // :305-366
  G2Precomp out; HFq two_inv = HFq::from_u64(2).inv(); G2Proj R{qx, qy, HFq2::one()}; bool found = false; EllCoeffs c;
  for (int i = 127; i >= 0; i--) { bool bit = (ATE_LOOP[i / 64] >> (i % 64)) & 1; if (!found) { found |= bit; continue; }
    doubling_step(two_inv, R, c); out.push_back(c); if (bit) { mixed_addition_step(qx, qy, R, c); out.push_back(c); } }
  const FrobeniusTables &t = frobenius_tables();
  // mul_by_q, alt_bn128_g2.cpp:367-372
  HFq2 q1x = t.twist_mul_by_q_x * qx.frob(1), q1y = t.twist_mul_by_q_y * qy.frob(1), q2x = t.twist_mul_by_q_x * q1x.frob(1),
      q2y = (t.twist_mul_by_q_y * q1y.frob(1)).neg();
  mixed_addition_step(q1x, q1y, R, c); out.push_back(c); mixed_addition_step(q2x, q2y, R, c); out.push_back(c); return out; }
HFq12 miller_loop(const HFq &px, const HFq &py, const G2Precomp &q) {        // :368-418
  HFq12 f = HFq12::one(); bool found = false; size_t idx = 0;
  for (int i = 127; i >= 0; i--) { bool bit = (ATE_LOOP[i / 64] >> (i % 64)) & 1; if (!found) { found |= bit; continue; }
    const EllCoeffs &c = q[idx++]; f = f.sqr().mul_by_024(c.ell_0, c.ell_VW.mul_fq(py), c.ell_VV.mul_fq(px));
    if (bit) { const EllCoeffs &d = q[idx++]; f = f.mul_by_024(d.ell_0, d.ell_VW.mul_fq(py), d.ell_VV.mul_fq(px)); } }
  for (int k = 0; k < 2; k++) { const EllCoeffs &c = q[idx++]; f = f.mul_by_024(c.ell_0, c.ell_VW.mul_fq(py), c.ell_VV.mul_fq(px)); }
  return f; }
static HFq12 exp_by_neg_z(const HFq12 &a) { return a.cyclo_exp(BN_Z).conj(); }
HFq12 final_exponentiation(const HFq12 &elt) {                               // :110-238
  HFq12 C0 = elt.conj() * elt.inv(), first = C0.frob(2) * C0;
  HFq12 A = exp_by_neg_z(first), B = A.sqr(), C = B.sqr(), D = C * B, E = exp_by_neg_z(D), F = E.sqr(), G = exp_by_neg_z(F), H = D.conj(), I = G.conj(),
      J = I * E, K = J * H, L = K * B, M = K * E, N = M * first,
        O = L.frob(1), P = O * N, Q = K.frob(2), R = Q * P, S = first.conj(), T = S * L, U = T.frob(3);
  return U * R; }

} }  // namespace zk::host
