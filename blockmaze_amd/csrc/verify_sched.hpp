// The Groth16 pairing check as a SCHEDULE of Fq operations for one wave per proof (kernel K9, second generation).
//
// The first K9 (round 1-2) gave every proof one lane that interpreted ~1,000 Fq12-level instructions: ~26,000 dependent field products, 46 ms per launch whatever the
// batch.  The check has plenty of parallelism below the Fq12 level — an Fq12 product is 144 independent Fq products — so here the whole check
// (r1cs_gg_ppzksnark_verifier_strong_IC, SNARK/.../r1cs_gg_ppzksnark.tcc:509-623, over libff's optimal-ate pairing, FF/algebra/curves/alt_bn128/alt_bn128_pairing.cpp:
// doubling / mixed-addition steps :242-293, miller_loop :368-418, final_exponentiation :110-238) is written ONCE, on the host, as a straight-line program over
// field-element values, and compiled into ROUNDS: in a round every lane of the wave executes one operation of the same kind —
//     MUL   slot[dst] = slot[a] * slot[b]                          (one Montgomery product)
//     LIN   slot[dst] = sum of up to 12 terms +-(2^s * slot[src])  (what the tower's additions, the xi = 9 + u products and the Karatsuba-free recombinations amount to)
// — on values that live in LDS (32 bytes a slot) or in a per-key table of constants in global memory (line coefficients of the vk's gamma and delta, Frobenius
// constants, alpha_g1_beta_g2).  Dependencies fix a value's level; values of a level are packed into rounds of 64; slots are reused once a value's last reader has run.
// The program does not depend on the proof (nor, apart from the constants, on the key): no divergence, any number of proofs per launch, one wave each.
// The same builder drives a host simulator (simulate()), so the schedule is checked on the CPU against the host verifier before it ever reaches a GPU
// (tests/test_verifier_cpu.py).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <utility>
#include <vector>
#include "pairing_host.hpp"

namespace zk { namespace vsched {
using host::HFq; using host::HFq2;

constexpr uint32_t CONST_FLAG = 0x8000u, MAX_TERMS = 12, LANES = 256;   // a round: one operation per lane of a 256-thread workgroup (four waves per proof: a dense Fq12 product's 144 Fq products are ONE round)
inline uint32_t hdr(uint32_t kind, uint32_t count, uint32_t nterms, uint32_t wpl) { return kind | count << 4 | nterms << 14 | wpl << 19; }
inline void unhdr(uint32_t h, uint32_t &kind, uint32_t &count, uint32_t &nterms, uint32_t &wpl) { kind = h & 15; count = (h >> 4) & 1023; nterms = (h >> 14) & 31; wpl = (h >> 19) & 31; }
enum : uint32_t { K_MUL = 1, K_LIN = 2 };
// per-proof inputs: fixed slots 0 .. N_INPUTS-1 (Montgomery form)
enum : int { IN_AX = 0, IN_AY, IN_BX0, IN_BX1, IN_BY0, IN_BY1, IN_CX, IN_CY, IN_NACCX, IN_NACCY, IN_NACCW, N_INPUTS };
// (IN_NACC*: the negated input accumulator -acc = (x, -y) given as (x w, -y w, w) for any non-zero w in Fq — the gamma line values are evaluated times w, a factor the final
//  exponentiation kills — so that the accumulation kernel needs no inversion: w = ZZ ZZZ of its extended Jacobian sum)
constexpr int N_RESULT = 12, N_CHECK = 4;   // outputs: the GT value (tower order), then four values that must be zero (A, C on the curve; B on the twist: two components)

struct Schedule {
  std::vector<uint32_t> prog;        // rounds: [hdr(kind, count, nterms, words_per_lane), the next round's hdr, 0, 0], then count * words_per_lane words: dst, then the terms (src | neg << 16 | shift << 17); MUL: dst, a, b, 0
  std::vector<HFq> consts;           // CONST_FLAG | index
  uint32_t n_rounds = 0, n_slots = 0, n_mul = 0, n_lin = 0, out_slot[N_RESULT + N_CHECK] = {0}, alpha_beta_const = 0;   // alpha_beta_const: first of 12 constants holding vk.alpha_g1_beta_g2
};

class Builder {
 public:
  struct Node { uint8_t kind; int a, b; std::vector<std::pair<int, int>> terms; /* LIN: (signed power-of-two coefficient, node) */ int level; uint32_t cidx; };
  std::vector<Node> nodes; std::vector<HFq> consts; std::map<std::array<uint32_t, 8>, int> const_of; std::map<std::pair<int, int>, int> mul_of; int zero_node, one_node;
  Builder() { for (int i = 0; i < N_INPUTS; i++) nodes.push_back(Node{0, i, 0, {}, 0, 0}); zero_node = constant(HFq::zero()); one_node = constant(HFq::one()); }
  int input(int i) const { return i; }
  int constant(const HFq &v) { std::array<uint32_t, 8> k; memcpy(k.data(), v.l, 32); auto it = const_of.find(k); if (it != const_of.end()) return it->second;
    consts.push_back(v); nodes.push_back(Node{3, 0, 0, {}, 0, (uint32_t)consts.size() - 1}); return const_of[k] = (int)nodes.size() - 1; }
  bool is_zero(int n) const { return n == zero_node; }
  int mul(int a, int b) { if (is_zero(a) || is_zero(b)) return zero_node; if (a == one_node) return b; if (b == one_node) return a; if (a > b) std::swap(a, b);
    auto it = mul_of.find({a, b}); if (it != mul_of.end()) return it->second;
    nodes.push_back(Node{K_MUL, a, b, {}, 1 + std::max(nodes[a].level, nodes[b].level), 0}); return mul_of[{a, b}] = (int)nodes.size() - 1; }
  // sum of coef * node with small integer coefficients; coefficients are split into signed powers of two (non-adjacent form), long sums into a tree of LINs
  int lin(const std::vector<std::pair<int, int>> &in) {
    std::map<int, int> coef; for (auto &t : in) if (!is_zero(t.second) && t.first) coef[t.second] += t.first;
    std::vector<std::pair<int, int>> terms;
    for (auto &kv : coef) { int c = kv.second; if (!c) continue; const int sgn = c < 0 ? -1 : 1; unsigned m = (unsigned)(c < 0 ? -c : c); int s = 0;
      while (m) { if (m & 1) { int d = (m & 3) == 3 ? -1 : 1; terms.push_back({sgn * d * (1 << s), kv.first}); m = (unsigned)((int)m - d); } m >>= 1; s++; if (s > 4) throw std::runtime_error("verify schedule: coefficient too large"); } }
    if (terms.empty()) return zero_node;
    if (terms.size() == 1 && terms[0].first == 1) return terms[0].second;
    while (terms.size() > MAX_TERMS) { std::vector<std::pair<int, int>> next; for (size_t i = 0; i < terms.size(); i += MAX_TERMS) { std::vector<std::pair<int, int>> chunk(terms.begin() + i, terms.begin() + std::min(terms.size(), i + MAX_TERMS)); next.push_back({1, raw_lin(chunk)}); } terms = next; }
    return raw_lin(terms); }
  int add(int a, int b) { return lin({{1, a}, {1, b}}); }
  int sub(int a, int b) { return lin({{1, a}, {-1, b}}); }
  int neg(int a) { return lin({{-1, a}}); }
 private:
  int raw_lin(const std::vector<std::pair<int, int>> &terms) { int lv = 0; for (auto &t : terms) lv = std::max(lv, nodes[t.second].level); nodes.push_back(Node{K_LIN, 0, 0, terms, lv + 1, 0}); return (int)nodes.size() - 1; }
};

// ---- the tower on node ids -------------------------------------------------------------------------------------------------------------------------------------
struct F2 { int c0, c1; };
typedef std::array<F2, 6> F12;   // coefficients of w^0 .. w^5 over Fq2, w^6 = xi = 9 + u;  tower (c_i.c_j) <-> w^(2j + i)
struct Tower {
  Builder &b; explicit Tower(Builder &b) : b(b) {}
  F2 zero2() { return {b.zero_node, b.zero_node}; } F2 one2() { return {b.one_node, b.zero_node}; }
  F2 k2(const HFq2 &v) { return {b.constant(v.c0), b.constant(v.c1)}; }
  F2 add(const F2 &x, const F2 &y) { return {b.add(x.c0, y.c0), b.add(x.c1, y.c1)}; }
  F2 sub(const F2 &x, const F2 &y) { return {b.sub(x.c0, y.c0), b.sub(x.c1, y.c1)}; }
  F2 neg(const F2 &x) { return {b.neg(x.c0), b.neg(x.c1)}; }
  F2 conj(const F2 &x) { return {x.c0, b.neg(x.c1)}; }
  F2 mul(const F2 &x, const F2 &y) { return {b.lin({{1, b.mul(x.c0, y.c0)}, {-1, b.mul(x.c1, y.c1)}}), b.lin({{1, b.mul(x.c0, y.c1)}, {1, b.mul(x.c1, y.c0)}})}; }
  F2 sqr(const F2 &x) { return {b.lin({{1, b.mul(x.c0, x.c0)}, {-1, b.mul(x.c1, x.c1)}}), b.lin({{2, b.mul(x.c0, x.c1)}})}; }
  F2 mul_fq(const F2 &x, int k) { return {b.mul(x.c0, k), b.mul(x.c1, k)}; }
  F2 scale(const F2 &x, int k) { return {b.lin({{k, x.c0}}), b.lin({{k, x.c1}})}; }                       // small integer multiple
  F2 mul_xi(const F2 &x) { return {b.lin({{9, x.c0}, {-1, x.c1}}), b.lin({{9, x.c1}, {1, x.c0}})}; }
  // sum_i s_i * x_i over Fq2 with small integer coefficients, one LIN per component
  F2 lin2(const std::vector<std::pair<int, F2>> &t) { std::vector<std::pair<int, int>> a, c; for (auto &p : t) { a.push_back({p.first, p.second.c0}); c.push_back({p.first, p.second.c1}); } return {b.lin(a), b.lin(c)}; }
  F2 inv(const F2 &x) { int n = b.add(b.mul(x.c0, x.c0), b.mul(x.c1, x.c1)), t = fq_inv(n); return {b.mul(x.c0, t), b.neg(b.mul(x.c1, t))}; }
  // x^(q-2) by 4-bit windows (Fermat; the reference uses mpn_gcdext, fp.tcc:688 — same value): a chain of ~330 dependent products, one lane busy
  int fq_inv(int x) { int tab[16]; tab[0] = b.one_node; tab[1] = x; for (int i = 2; i < 16; i++) tab[i] = b.mul(tab[i - 1], x);
    uint32_t e[8]; uint64_t br = 2; for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)FqParams::MOD[i] - br; e[i] = (uint32_t)t; br = (t >> 32) & 1; }
    int r = -1; for (int w = 63; w >= 0; w--) { if (r >= 0) for (int k = 0; k < 4; k++) r = b.mul(r, r); const uint32_t d = (e[w >> 3] >> ((w & 7) * 4)) & 15; if (d) r = r < 0 ? tab[d] : b.mul(r, tab[d]); }
    return r; }

  F12 one12() { F12 r; for (auto &c : r) c = zero2(); r[0] = one2(); return r; }
  F12 mul(const F12 &x, const F12 &y) {   // 36 Fq2 products (fewer when an operand is sparse: products with the zero node vanish), then per coefficient: lo + xi * hi
    F12 r;
    for (int k = 0; k < 6; k++) { std::vector<std::pair<int, int>> lo0, lo1, hi0, hi1;
      for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { if (i + j != k && i + j != k + 6) continue; const bool hi = i + j == k + 6;
        int p00 = b.mul(x[i].c0, y[j].c0), p11 = b.mul(x[i].c1, y[j].c1), p01 = b.mul(x[i].c0, y[j].c1), p10 = b.mul(x[i].c1, y[j].c0);
        (hi ? hi0 : lo0).push_back({1, p00}); (hi ? hi0 : lo0).push_back({-1, p11}); (hi ? hi1 : lo1).push_back({1, p01}); (hi ? hi1 : lo1).push_back({1, p10}); }
      int l0 = b.lin(lo0), l1 = b.lin(lo1), h0 = b.lin(hi0), h1 = b.lin(hi1);
      r[k] = {b.lin({{1, l0}, {9, h0}, {-1, h1}}), b.lin({{1, l1}, {9, h1}, {1, h0}})}; }
    return r; }
  F12 conj(const F12 &x) { F12 r = x; for (int e = 1; e < 6; e += 2) r[e] = neg(x[e]); return r; }                                                  // unitary inverse: c1 -> -c1, i.e. the odd powers of w
  F12 frob(const F12 &x, unsigned p) { const host::FrobeniusTables &t = host::frobenius_tables(); F12 r;                                            // fp12_2over3over2.tcc:139-147 over fp6_3over2.tcc:148-157
    for (int e = 0; e < 6; e++) { const int i = e & 1, j = e >> 1; HFq2 k = HFq2::one(); if (j == 1) k = t.fq6_c1[p % 6]; if (j == 2) k = t.fq6_c2[p % 6]; if (i) k = k * t.fq12_c1[p % 12];
      F2 v = (p & 1) ? conj(x[e]) : x[e]; r[e] = (j == 0 && i == 0) ? v : mul(v, k2(k)); }
    return r; }
  // Fq6 = Fq2[v]/(v^3 - xi) on {c0, c1, c2}: only what the Fq12 inverse needs
  typedef std::array<F2, 3> F6;
  F6 mul6(const F6 &x, const F6 &y) { F2 aA = mul(x[0], y[0]), bB = mul(x[1], y[1]), cC = mul(x[2], y[2]);
    return {add(aA, mul_xi(sub(sub(mul(add(x[1], x[2]), add(y[1], y[2])), bB), cC))), add(sub(sub(mul(add(x[0], x[1]), add(y[0], y[1])), aA), bB), mul_xi(cC)), add(sub(mul(add(x[0], x[2]), add(y[0], y[2])), aA), sub(bB, cC))}; }
  F6 mul_by_v(const F6 &x) { return {mul_xi(x[2]), x[0], x[1]}; }
  F12 inv(const F12 &x) {   // fp12_2over3over2.tcc:128-137 over fp6_3over2.tcc:128-146
    F6 c0 = {x[0], x[2], x[4]}, c1 = {x[1], x[3], x[5]}, s0 = mul6(c0, c0), s1 = mul6(c1, c1), m = mul_by_v(s1), t = {sub(s0[0], m[0]), sub(s0[1], m[1]), sub(s0[2], m[2])};
    F2 t0 = sqr(t[0]), t1 = sqr(t[1]), t2 = sqr(t[2]), t3 = mul(t[0], t[1]), t4 = mul(t[0], t[2]), t5 = mul(t[1], t[2]), d0 = sub(t0, mul_xi(t5)), d1 = sub(mul_xi(t2), t3), d2 = sub(t1, t4);
    F2 t6 = inv(add(mul(t[0], d0), mul_xi(add(mul(t[2], d1), mul(t[1], d2))))); F6 ti = {mul(t6, d0), mul(t6, d1), mul(t6, d2)}, r0 = mul6(c0, ti), r1 = mul6(c1, ti);
    return {r0[0], neg(r1[0]), r0[1], neg(r1[1]), r0[2], neg(r1[2])}; }
  // the operand of mul_by_024 (fp12_2over3over2.tcc:240-335): (ell_0, 0, ell_VV | 0, ell_VW, 0) in the tower = w^0, w^4, w^3
  F12 sparse(const F2 &ell_0, const F2 &ell_VW, const F2 &ell_VV) { F12 r; for (auto &c : r) c = zero2(); r[0] = ell_0; r[3] = ell_VW; r[4] = ell_VV; return r; }
};

static const uint64_t BN_Z = 4965661367192848881ull;                         // alt_bn128_init.cpp:327 (final_exponent_z)
static const uint64_t ATE_LOOP[2] = {0x9d797039be763ba8ull, 0x1ull};          // 6z+2 (alt_bn128_init.cpp:324)

// The whole check for one verifying key.  gamma / delta: the precomputed line coefficients of the vk's G2 points (host::precompute_g2).
inline Schedule build(const host::HFq12 &alpha_g1_beta_g2, const host::G2Precomp &gamma, const host::G2Precomp &delta) {
  Builder b; Tower T(b); const host::FrobeniusTables &ft = host::frobenius_tables();
  const HFq2 twist_b_v = HFq2{HFq::from_u64(3), HFq::zero()} * HFq2{HFq::from_u64(9), HFq::one()}.inv(); const int two_inv = b.constant(HFq::from_u64(2).inv()), three = b.constant(HFq::from_u64(3)); const F2 twist_b = T.k2(twist_b_v);
  const int ax = b.input(IN_AX), ay = b.input(IN_AY), cx = b.input(IN_CX), cy = b.input(IN_CY), nx = b.input(IN_NACCX), ny = b.input(IN_NACCY), nw = b.input(IN_NACCW); const F2 bx = {b.input(IN_BX0), b.input(IN_BX1)}, by = {b.input(IN_BY0), b.input(IN_BY1)};
  const int ncy = b.neg(cy);
  // is_well_formed: on-curve residues (alt_bn128_g1.cpp:92-117, alt_bn128_g2.cpp:98-127), must come out zero
  int chk[N_CHECK]; chk[0] = b.lin({{1, b.mul(ay, ay)}, {-1, b.mul(b.mul(ax, ax), ax)}, {-1, three}}); chk[1] = b.lin({{1, b.mul(cy, cy)}, {-1, b.mul(b.mul(cx, cx), cx)}, {-1, three}});
  { F2 r = T.sub(T.sub(T.sqr(by), T.mul(T.sqr(bx), bx)), twist_b); chk[2] = r.c0; chk[3] = r.c1; }
  // miller_loop :368-418 for e(A, B) with the running G2 point in homogeneous projective coordinates (:242-293), and the precomputed lines of gamma (at -acc) and delta (at -C)
  F2 X = bx, Y = by, Z = T.one2(); F12 f = T.one12(); size_t idx = 0;
  const F2 q1x = T.mul(T.k2(ft.twist_mul_by_q_x), T.conj(bx)), q1y = T.mul(T.k2(ft.twist_mul_by_q_y), T.conj(by)), q2x = T.mul(T.k2(ft.twist_mul_by_q_x), T.conj(q1x)), q2y = T.neg(T.mul(T.k2(ft.twist_mul_by_q_y), T.conj(q1y)));   // mul_by_q, alt_bn128_g2.cpp:367-372
  // the three sparse line values of a step — the running point's own, and the precomputed ones of gamma (at -acc) and delta (at -C) — are multiplied TOGETHER off the
  // chain of f: they depend on the G2 chain and on constants only, so the accumulator's chain is f <- f^2 * L per bit instead of four products in sequence
  auto lines = [&](const F12 &own) -> F12 {
    if (idx >= gamma.size() || idx >= delta.size()) throw std::runtime_error("verify schedule: line count");
    const host::EllCoeffs &g = gamma[idx], &d = delta[idx]; idx++;
    const F12 lg = T.sparse(T.mul_fq(T.k2(g.ell_0), nw), T.mul_fq(T.k2(g.ell_VW), ny), T.mul_fq(T.k2(g.ell_VV), nx)), ld = T.sparse(T.k2(d.ell_0), T.mul_fq(T.k2(d.ell_VW), ncy), T.mul_fq(T.k2(d.ell_VV), cx));
    return T.mul(own, T.mul(lg, ld)); };
  auto dbl_step = [&]() {   // doubling_step_for_flipped_miller_loop :242-268
    F2 A = T.mul_fq(T.mul(X, Y), two_inv), B = T.sqr(Y), C = T.sqr(Z), D = T.scale(C, 3), E = T.mul(twist_b, D), F = T.scale(E, 3), G = T.mul_fq(T.add(B, F), two_inv), H = T.scale(T.mul(Y, Z), 2) /* = (Y + Z)^2 - (B + C) */, I = T.sub(E, B), J = T.sqr(X), E2 = T.sqr(E);
    X = T.mul(A, T.sub(B, F)); Y = T.sub(T.sqr(G), T.scale(E2, 3)); Z = T.mul(B, H);
    return T.sparse(T.mul_xi(I), T.mul_fq(T.neg(H), ay), T.mul_fq(T.scale(J, 3), ax)); };
  auto add_step = [&](const F2 &x2, const F2 &y2) {   // mixed_addition_step_for_flipped_miller_loop :270-293
    F2 D = T.sub(X, T.mul(x2, Z)), E = T.sub(Y, T.mul(y2, Z)), F = T.sqr(D), G = T.sqr(E), H = T.mul(D, F), I = T.mul(X, F), J = T.sub(T.add(H, T.mul(Z, G)), T.scale(I, 2)), Y1 = Y;
    X = T.mul(D, J); Y = T.sub(T.mul(E, T.sub(I, J)), T.mul(H, Y1)); Z = T.mul(Z, H);
    return T.sparse(T.mul_xi(T.sub(T.mul(E, x2), T.mul(D, y2))), T.mul_fq(D, ay), T.mul_fq(T.neg(E), ax)); };
  bool found = false;
  for (int i = 127; i >= 0; i--) { const bool bit = (ATE_LOOP[i / 64] >> (i % 64)) & 1; if (!found) { found |= bit; continue; }
    F12 L = lines(dbl_step()); if (bit) L = T.mul(L, lines(add_step(bx, by))); f = T.mul(T.mul(f, f), L); }
  { F12 L1 = lines(add_step(q1x, q1y)), L2 = lines(add_step(q2x, q2y)); f = T.mul(f, T.mul(L1, L2)); }
  if (idx != gamma.size() || idx != delta.size()) throw std::runtime_error("verify schedule: line count");
  // final_exponentiation :110-238: first chunk f^((q^6 - 1)(q^2 + 1)), then the last chunk's chain with three exponentiations by -z.  Squarings in the cyclotomic subgroup
  // are plain squarings here: with all 144 coefficient products of a round in parallel the Granger-Scott form saves nothing.
  auto exp_neg_z = [&](const F12 &src) {   // alt_bn128_pairing.cpp:84-96: conj(src^z), src in the cyclotomic subgroup.  Width-3 NAF of z: 18 non-zero digits in {+-1, +-3}; a negative digit is a conjugation
    std::vector<int> dig; for (uint64_t k = BN_Z; k;) { int t = 0; if (k & 1) { t = (int)(k & 7); if (t >= 4) t -= 8; k -= (uint64_t)(int64_t)t; } dig.push_back(t); k >>= 1; }
    const F12 s3 = T.mul(T.mul(src, src), src), sc = T.conj(src), s3c = T.conj(s3); F12 r = T.one12(); bool started = false;
    for (size_t i = dig.size(); i-- > 0;) { const int d = dig[i]; if (started) r = T.mul(r, r); if (!d) continue; const F12 &m = d == 1 ? src : d == -1 ? sc : d == 3 ? s3 : s3c; r = started ? T.mul(r, m) : m; started = true; }
    return T.conj(r); };
  F12 c0 = T.mul(T.conj(f), T.inv(f)), first = T.mul(T.frob(c0, 2), c0);
  F12 A = exp_neg_z(first), Bq = T.mul(A, A), Cq = T.mul(Bq, Bq), D = T.mul(Cq, Bq), E = exp_neg_z(D), Fq_ = T.mul(E, E), G = exp_neg_z(Fq_), H = T.conj(D), I = T.conj(G), J = T.mul(I, E), K = T.mul(J, H), L = T.mul(K, Bq), M = T.mul(K, E), N = T.mul(M, first),
      O = T.frob(L, 1), P = T.mul(O, N), Q = T.frob(K, 2), R = T.mul(Q, P), S = T.conj(first), Tt = T.mul(S, L), U = T.frob(Tt, 3), V = T.mul(U, R);
  // outputs in tower order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 — each c0 then c1): what vk.alpha_g1_beta_g2 is compared with
  std::vector<int> outs; for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) { const F2 &c = V[2 * j + i]; outs.push_back(c.c0); outs.push_back(c.c1); } for (int k = 0; k < N_CHECK; k++) outs.push_back(chk[k]);

  // ---- liveness, rounds, slots -----------------------------------------------------------------------------------------------------------------------------------
  const size_t nn = b.nodes.size(); std::vector<char> live(nn, 0); for (int o : outs) live[o] = 1;
  for (size_t n = nn; n-- > 0;) { if (!live[n]) continue; const Builder::Node &nd = b.nodes[n]; if (nd.kind == K_MUL) { live[nd.a] = live[nd.b] = 1; } else if (nd.kind == K_LIN) for (auto &t : nd.terms) live[t.second] = 1; }
  // an output that is an input or a constant (degenerate) still needs a slot of its own: route it through a LIN — cannot happen for this program, checked
  for (int o : outs) if (b.nodes[o].kind != K_MUL && b.nodes[o].kind != K_LIN) throw std::runtime_error("verify schedule: output is not a computed value");
  int max_level = 0; for (size_t n = 0; n < nn; n++) if (live[n]) max_level = std::max(max_level, b.nodes[n].level);
  // as LATE as possible: a value is placed one level before its first reader (outputs at the last level).  Scheduled as early as possible, everything off the critical
  // chain — the G2 point's chain, the products of the line values — ran hundreds of levels ahead of its readers and the live values did not fit the LDS (3,769 slots).
  { std::vector<int> alap(nn, max_level + 1); for (int o : outs) alap[o] = max_level;
    for (size_t n = nn; n-- > 0;) { if (!live[n]) continue; Builder::Node &nd = b.nodes[n]; if (alap[n] > max_level) alap[n] = max_level; const int lv = alap[n];
      if (nd.kind == K_MUL) { alap[nd.a] = std::min(alap[nd.a], lv - 1); alap[nd.b] = std::min(alap[nd.b], lv - 1); } else if (nd.kind == K_LIN) for (auto &t : nd.terms) alap[t.second] = std::min(alap[t.second], lv - 1); }
    for (size_t n = 0; n < nn; n++) if (live[n] && (b.nodes[n].kind == K_MUL || b.nodes[n].kind == K_LIN)) { if (alap[n] < b.nodes[n].level) throw std::runtime_error("verify schedule: level"); b.nodes[n].level = alap[n]; } }
  std::vector<std::vector<int>> mul_at(max_level + 1), lin_at(max_level + 1);
  for (size_t n = 0; n < nn; n++) if (live[n]) { if (b.nodes[n].kind == K_MUL) mul_at[b.nodes[n].level].push_back((int)n); else if (b.nodes[n].kind == K_LIN) lin_at[b.nodes[n].level].push_back((int)n); }
  struct Round { uint32_t kind; std::vector<int> ns; }; std::vector<Round> rounds;
  for (int lv = 1; lv <= max_level; lv++) for (int kind = K_MUL; kind <= (int)K_LIN; kind++) { const std::vector<int> &v = kind == (int)K_MUL ? mul_at[lv] : lin_at[lv];
    for (size_t i = 0; i < v.size(); i += LANES) rounds.push_back(Round{(uint32_t)kind, std::vector<int>(v.begin() + i, v.begin() + std::min(v.size(), i + LANES))}); }
  std::vector<int> last_use(nn, -1); std::vector<char> is_out(nn, 0); for (int o : outs) is_out[o] = 1;
  for (size_t r = 0; r < rounds.size(); r++) for (int n : rounds[r].ns) { const Builder::Node &nd = b.nodes[n]; if (nd.kind == K_MUL) { last_use[nd.a] = last_use[nd.b] = (int)r; } else for (auto &t : nd.terms) last_use[t.second] = (int)r; }
  Schedule sc; sc.consts = b.consts; std::vector<int> slot(nn, -1); for (int i = 0; i < N_INPUTS; i++) slot[i] = i; std::vector<int> free_slots; uint32_t next_slot = N_INPUTS;
  for (size_t k = 0; k < 12; k++) { const HFq2 *c2 = k < 6 ? &alpha_g1_beta_g2.c0.c0 + k / 2 : &alpha_g1_beta_g2.c1.c0 + (k - 6) / 2; const HFq &v = (k & 1) ? c2->c1 : c2->c0; sc.consts.push_back(v); if (k == 0) sc.alpha_beta_const = (uint32_t)sc.consts.size() - 1; }   // (appended, not deduplicated: twelve consecutive entries)
  if (sc.consts.size() >= CONST_FLAG) throw std::runtime_error("verify schedule: too many constants");
  auto ref = [&](int n) -> uint32_t { const Builder::Node &nd = b.nodes[n]; if (nd.kind == 3) return CONST_FLAG | nd.cidx; if (slot[n] < 0) throw std::runtime_error("verify schedule: value read before it was written"); return (uint32_t)slot[n]; };
  std::vector<size_t> hdr_pos;
  for (size_t r = 0; r < rounds.size(); r++) { const Round &rd = rounds[r]; uint32_t nterms = 2;
    if (rd.kind == K_LIN) { nterms = 1; for (int n : rd.ns) nterms = std::max<uint32_t>(nterms, (uint32_t)b.nodes[n].terms.size()); }
    const uint32_t wpl = rd.kind == K_MUL ? 4 : ((1 + nterms + 3) & ~3u); hdr_pos.push_back(sc.prog.size()); sc.prog.push_back(hdr(rd.kind, (uint32_t)rd.ns.size(), nterms, wpl)); sc.prog.insert(sc.prog.end(), 3, 0u);   // (four header words: every lane's words stay 16-byte aligned; word 1 = the NEXT round's header, filled in below)
    std::vector<uint32_t> srcs; for (int n : rd.ns) { const Builder::Node &nd = b.nodes[n]; std::vector<uint32_t> w(wpl, 0);   // (operands are resolved BEFORE this round's destinations are assigned)
      if (rd.kind == K_MUL) { w[1] = ref(nd.a); w[2] = ref(nd.b); sc.n_mul++; }
      else { for (uint32_t t = 0; t < nterms; t++) { if (t < nd.terms.size()) { int c = nd.terms[t].first; const uint32_t ng = c < 0; unsigned m = (unsigned)(c < 0 ? -c : c), sh = 0; while (m > 1) { m >>= 1; sh++; } w[1 + t] = ref(nd.terms[t].second) | ng << 16 | sh << 17; } else w[1 + t] = CONST_FLAG | b.nodes[b.zero_node].cidx; } sc.n_lin++; }
      srcs.insert(srcs.end(), w.begin(), w.end()); }
    size_t pos = 0; for (int n : rd.ns) { int s; if (!free_slots.empty()) { s = free_slots.back(); free_slots.pop_back(); } else s = (int)next_slot++; slot[n] = s; srcs[pos] = (uint32_t)s; pos += wpl; }
    sc.prog.insert(sc.prog.end(), srcs.begin(), srcs.end());
    for (int n : rd.ns) { const Builder::Node &nd = b.nodes[n]; auto release = [&](int o) { if (b.nodes[o].kind != 3 && o >= N_INPUTS && last_use[o] == (int)r && !is_out[o] && slot[o] >= 0) { free_slots.push_back(slot[o]); slot[o] = -2; } };
      if (nd.kind == K_MUL) { release(nd.a); release(nd.b); } else for (auto &t : nd.terms) release(t.second); }
  }
  for (size_t r = 0; r + 1 < hdr_pos.size(); r++) sc.prog[hdr_pos[r] + 1] = sc.prog[hdr_pos[r + 1]];   // look-ahead: a round's header also carries the next one's, so the kernel can fetch the next round's words without waiting for a header load
  sc.n_rounds = (uint32_t)rounds.size(); sc.n_slots = next_slot; for (size_t k = 0; k < outs.size(); k++) sc.out_slot[k] = (uint32_t)slot[outs[k]];
  return sc;
}

// Host interpreter of a schedule: what the kernel does, on HFq.  in: the N_INPUTS values of one proof.  Returns the output values (N_RESULT + N_CHECK).
inline std::vector<HFq> simulate(const Schedule &S, const HFq *in) {
  std::vector<HFq> slots(S.n_slots, HFq::zero()); for (int i = 0; i < N_INPUTS; i++) slots[i] = in[i];
  auto val = [&](uint32_t r) { return (r & CONST_FLAG) ? S.consts[r & (CONST_FLAG - 1)] : slots[r & 0xffff]; };
  size_t pc = 0; std::vector<std::pair<uint32_t, HFq>> writes;
  for (uint32_t r = 0; r < S.n_rounds; r++) { uint32_t kind, count, nterms, wpl; unhdr(S.prog[pc], kind, count, nterms, wpl); pc += 4; writes.clear();
    for (uint32_t l = 0; l < count; l++) { const uint32_t *w = &S.prog[pc + (size_t)l * wpl]; HFq v;
      if (kind == K_MUL) v = val(w[1]) * val(w[2]);
      else { v = HFq::zero(); for (uint32_t t = 0; t < nterms; t++) { HFq x = val(w[1 + t] & 0xffff); for (uint32_t s = 0; s < ((w[1 + t] >> 17) & 7); s++) x = x + x; v = ((w[1 + t] >> 16) & 1) ? v - x : v + x; } }
      writes.push_back({w[0], v}); }
    for (auto &wv : writes) slots[wv.first] = wv.second;   // (all reads of a round happen before its writes, as in the lock-step wave)
    pc += (size_t)count * wpl; }
  std::vector<HFq> out; for (int k = 0; k < N_RESULT + N_CHECK; k++) out.push_back(slots[S.out_slot[k]]); return out;
}

} }  // namespace zk::vsched
