// The Groth16 pairing check as a SCHEDULE of Fq operations for one workgroup per proof (kernel K9, third generation).
//
// The first K9 (round 1-2) gave every proof one lane that interpreted ~1,000 Fq12-level instructions: ~26,000 dependent field products, 46 ms per launch
// whatever the batch. The check has plenty of parallelism below the Fq12 level — an Fq12 product is 144 independent Fq products — so here the whole check
// (r1cs_gg_ppzksnark_verifier_strong_IC, SNARK/.../r1cs_gg_ppzksnark.tcc:509-623, over libff's optimal-ate pairing,
// FF/algebra/curves/alt_bn128/alt_bn128_pairing.cpp: doubling / mixed-addition steps :242-293, miller_loop :368-418, final_exponentiation :110-238) is written
// ONCE, on the host, as a straight-line program over field-element values, and compiled into ROUNDS: in a round every lane of the 256-thread workgroup executes
// one operation of the same kind —
//     MUL    slot[dst] = slot[a] * slot[b]                         (one Montgomery product per lane)
//     LIN8   slot[dst] = sum of up to 24 terms c * slot[src], |c| <= 255   (eight lanes per value, three terms each, then a tree over the eight lanes)
//     LIN1   the same with up to 3 terms, one lane per value
// — on values that live in LDS. Round 3, second half: the values are kept on NINE 29-BIT LIMBS (Montgomery radix 2^261, gen_field29.py — the representation of
// the H accumulation): a product is 162 multiply-adds and no carry instruction, and a linear combination is a multiply-add per limb and term on 64-bit limb
// accumulators with NO modular correction per term — one Barrett-like step at the end brings the value back below 4.1 p (l29::barrett; the pipeline is modelled
// and checked on integers in gen_field29.py: lin_check). Because sums are that cheap, the tower's two levels of recombination after a product (Karatsuba-free
// sums, then lo + xi * hi) are FLATTENED into one linear combination of the products (Builder::lin substitutes the terms of an operand that is itself a linear
// combination): an Fq12 product is two rounds. Dependencies fix a value's level; slots are reused once a value's last reader has run. The program does not
// depend on the proof (nor, apart from the constants, on the key): no divergence, any number of proofs per launch.
// The same builder drives two host interpreters, so the schedule is checked on the CPU before it ever reaches a GPU (tests/test_verifier_cpu.py): simulate() on
// the host field type (what the program MEANS), and simulate29() on the device's limb arithmetic, operation by operation, with every bound asserted (what the
// kernel DOES).
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>
#include "pairing_host.hpp"
#include "field29_params.h"
#if defined(__HIPCC__)
#define ZK_HD __host__ __device__ __forceinline__
#define ZK_UNROLL _Pragma("unroll")
#else
#define ZK_HD inline
#define ZK_UNROLL
#endif

namespace zk {
// ---- the limb arithmetic of a linear combination, shared by the kernel (pairing.cuh) and the host model (simulate29)
// --------------------------------------------
namespace l29 {
constexpr uint32_t STRIDE = 12;          // dwords per value in LDS / in the table of constants: nine limbs + padding to 48 bytes (three 16-byte reads)
// acc += c * (neg ? K6 - x : x), limb by limb.  x: limbs below 2^29 + 8 (top limb: the rest), value below 6 p
ZK_HD void term(uint64_t (&acc)[9], const uint32_t (&x)[9], uint32_t c, bool neg) {
  ZK_UNROLL for (int i = 0; i < 9; i++) { const uint32_t y = neg ? p29::K6[i] - x[i] : x[i]; acc[i] += (uint64_t)c * y; } }
// one parallel carry step from 64-bit accumulators to 32-bit limbs (the top limb modulo 2^32)
ZK_HD void norm64(const uint64_t (&acc)[9], uint32_t (&l)[9]) { l[0] = (uint32_t)acc[0] & p29::MASK;
  ZK_UNROLL for (int i = 1; i < 8; i++) l[i] = ((uint32_t)acc[i] & p29::MASK) + (uint32_t)(acc[i - 1] >> 29);
  l[8] = (uint32_t)acc[8] + (uint32_t)(acc[7] >> 29); }
ZK_HD void norm32(uint32_t (&l)[9]) { uint32_t r[9]; r[0] = l[0] & p29::MASK;
  ZK_UNROLL for (int i = 1; i < 8; i++) r[i] = (l[i] & p29::MASK) + (l[i - 1] >> 29);
  r[8] = l[8] + (l[7] >> 29);
  ZK_UNROLL for (int i = 0; i < 9; i++) l[i] = r[i]; }
// l: limbs below 2^31, value V below LIN_MAX_UNITS p. V - q p with q = max(floor-estimate of V / p from the top limb - 1, 0): limbs below 2^29 + 2, value below
// 4.1 p (and at least p when q > 0). V + q (2^264 - p) with the top limb modulo 2^32: the q multiples of 2^264 fall out (gen_field29.py: barrett, lin_check)
ZK_HD void barrett(uint32_t (&l)[9]) { uint32_t q = (uint32_t)(((uint64_t)l[8] * p29::MU) >> 32) >> (p29::MU_SHIFT - 32); q = q ? q - 1 : 0; uint64_t acc[9];
  ZK_UNROLL for (int i = 0; i < 9; i++) acc[i] = (uint64_t)l[i] + (uint64_t)q * p29::NP[i];
  norm64(acc, l); norm32(l); }
// is the value (below 5 p, limbs below 2^30) a multiple of p?  Exact carry propagation, then the five candidates
ZK_HD bool multiple_of_p(const uint32_t (&v)[9]) { uint32_t f[9];
  ZK_UNROLL for (int i = 0; i < 9; i++) f[i] = v[i];
  ZK_UNROLL for (int i = 0; i < 8; i++) { f[i + 1] += f[i] >> 29; f[i] &= p29::MASK; }
  bool any = false;
  ZK_UNROLL for (int k = 0; k < 5; k++) { bool eq = true; ZK_UNROLL for (int i = 0; i < 9; i++) eq = eq && f[i] == p29::KP[k][i]; any = any || eq; }
  return any; }
// the nine limbs of a 256-bit integer given as eight 32-bit words
ZK_HD void unpack(const uint32_t (&w)[8], uint32_t (&l)[9]) {
  ZK_UNROLL for (int i = 0; i < 9; i++) {
    const int bit = 29 * i, j = bit >> 5, s = bit & 31;
    uint32_t v = w[j] >> s;
    if (s > 3 && j + 1 < 8) v |= w[j + 1] << (32 - s);
    l[i] = i < 8 ? (v & p29::MASK) : v;
  }
}
// Montgomery form 2^256 -> 2^261: times 32, reduced (the kernel's input conversion)
ZK_HD void lift(const uint32_t (&w)[8], uint32_t (&l)[9]) {
  uint32_t x[9];
  unpack(w, x);
  uint64_t acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  term(acc, x, 32u, false);
  norm64(acc, l);
  barrett(l);
}
// host model of Fq29::mul (field29_gfx950.inc): product scanning with the Montgomery multiples m_k p folded into the same 64-bit columns; every column asserted
// below 2^64
inline void mul_model(const uint32_t (&a)[9], const uint32_t (&b)[9], uint32_t (&r)[9]) { unsigned __int128 acc = 0; uint32_t m[9];
  for (int k = 0; k < 17; k++) {
    for (int i = k > 8 ? k - 8 : 0; i <= (k < 8 ? k : 8); i++) acc += (unsigned __int128)a[i] * b[k - i];
    if (k < 9) {
      for (int i = 0; i < k; i++) acc += (unsigned __int128)m[i] * p29::P[k - i];
      m[k] = ((uint32_t)acc * p29::INV) & p29::MASK;
      acc += (unsigned __int128)m[k] * p29::P[0];
      if (acc >> 64) throw std::runtime_error("mul29: column overflow");
      if ((uint32_t)acc & p29::MASK) throw std::runtime_error("mul29: reduction");
      acc >>= 29;
    }
    else {
      for (int i = k - 8; i < 9; i++) acc += (unsigned __int128)m[i] * p29::P[k - i];
      if (acc >> 64) throw std::runtime_error("mul29: column overflow");
      r[k - 9] = (uint32_t)acc & p29::MASK;
      acc >>= 29;
    }
  }
  if (acc >> 32) throw std::runtime_error("mul29: top limb overflow");
  r[8] = (uint32_t)acc; }
}  // namespace l29

namespace vsched {
using host::HFq; using host::HFq2;

// a round: one operation per lane of a 256-thread workgroup (a dense Fq12 product's 144 Fq products are ONE round)
constexpr uint32_t CONST_FLAG = 0x8000u, LIN_GROUP = 8, TERMS_PER_LANE = 3, MAX_TERMS = LIN_GROUP * TERMS_PER_LANE, MAX_COEF = 255, LANES = 256;
// instruction words per lane: MUL dst, a, b, 0; LIN dst, then three terms (src | neg << 16 | |c| << 17; c = 0: no term)
constexpr uint32_t WPL = 4;
inline uint32_t hdr(uint32_t kind, uint32_t count) { return kind | count << 4; }
inline void unhdr(uint32_t h, uint32_t &kind, uint32_t &count) { kind = h & 15; count = (h >> 4) & 1023; }
enum : uint32_t { K_MUL = 1, K_LIN8 = 2, K_LIN1 = 3 };
// per-proof inputs: fixed slots 0 .. N_INPUTS-1
enum : int { IN_AX = 0, IN_AY, IN_BX0, IN_BX1, IN_BY0, IN_BY1, IN_CX, IN_CY, IN_NACCX, IN_NACCY, IN_NACCW, N_INPUTS };
// (IN_NACC*: the negated input accumulator -acc = (x, -y) given as (x w, -y w, w) for any non-zero w in Fq — the gamma line values are evaluated times w, a
// factor the final
//  exponentiation kills — so that the accumulation kernel needs no inversion: w = ZZ ZZZ of its extended Jacobian sum)
// outputs, all of which must be ZERO (mod p): the GT value minus vk.alpha_g1_beta_g2 (tower order), then A, C on the curve and B on the twist (two components)
constexpr int N_RESULT = 12, N_CHECK = 4;

struct Schedule {
  std::vector<uint32_t> prog;        // rounds: [hdr(kind, lanes), the next round's hdr, 0, 0], then lanes * WPL words
  std::vector<HFq> consts;           // CONST_FLAG | index
  uint32_t n_rounds = 0, n_slots = 0, n_mul = 0, n_lin = 0, rounds_of_kind[4] = {0, 0, 0, 0}, out_slot[N_RESULT + N_CHECK] = {0};
};

class Builder {
 public:
  struct Node { uint8_t kind; int a, b; std::vector<std::pair<int, int>> terms; /* LIN: (integer coefficient, node) */ int level; uint32_t cidx; };
  std::vector<Node> nodes;
  std::vector<HFq> consts;
  std::map<std::array<uint32_t, 8>, int> const_of;
  std::map<std::pair<int, int>, int> mul_of;
  int zero_node, one_node;
  Builder() { for (int i = 0; i < N_INPUTS; i++) nodes.push_back(Node{0, i, 0, {}, 0, 0}); zero_node = constant(HFq::zero()); one_node = constant(HFq::one()); }
  int input(int i) const { return i; }
  int constant(const HFq &v) { std::array<uint32_t, 8> k; memcpy(k.data(), v.l, 32); auto it = const_of.find(k); if (it != const_of.end()) return it->second;
    consts.push_back(v); nodes.push_back(Node{3, 0, 0, {}, 0, (uint32_t)consts.size() - 1}); return const_of[k] = (int)nodes.size() - 1; }
  bool is_zero(int n) const { return n == zero_node; }
  int mul(int a, int b) { if (is_zero(a) || is_zero(b)) return zero_node; if (a == one_node) return b; if (b == one_node) return a; if (a > b) std::swap(a, b);
    auto it = mul_of.find({a, b}); if (it != mul_of.end()) return it->second;
    nodes.push_back(Node{K_MUL, a, b, {}, 1 + std::max(nodes[a].level, nodes[b].level), 0}); return mul_of[{a, b}] = (int)nodes.size() - 1; }
  // sum of coef * node with small integer coefficients. An operand that is itself a linear combination is replaced by its terms (one level of sums instead of
  // two) as long as the result keeps to MAX_TERMS terms, |c| <= MAX_COEF and sum |c| * 6 p below the Barrett step's range; longer sums become a tree of LINs
  int lin(const std::vector<std::pair<int, int>> &in) {
    std::map<int, long> direct; for (auto &t : in) if (!is_zero(t.second) && t.first) direct[t.second] += t.first;
    std::map<int, long> flat;
    for (auto &kv : direct) {
      if (!kv.second) continue;
      const Node &nd = nodes[kv.first];
      if (nd.kind == K_LIN8) for (auto &t : nd.terms) flat[t.second] += kv.second * t.first;
      else flat[kv.first] += kv.second;
    }
    auto fits = [](const std::map<int, long> &m, bool any_length) {
      long units = 0;
      size_t n = 0;
      for (auto &kv : m) {
        if (!kv.second) continue;
        n++;
        const long c = kv.second < 0 ? -kv.second : kv.second;
        if (c > (long)MAX_COEF) return false;
        units += c;
      }
      return units * 6 < (long)p29::LIN_MAX_UNITS && (any_length || n <= MAX_TERMS);
    };
    const std::map<int, long> &use = fits(flat, false) ? flat : direct;
    if (!fits(use, true)) throw std::runtime_error("verify schedule: a linear combination exceeds the range of the reduction step");
    std::vector<std::pair<int, int>> terms; for (auto &kv : use) if (kv.second) terms.push_back({(int)kv.second, kv.first});
    if (terms.empty()) return zero_node;
    if (terms.size() == 1 && terms[0].first == 1) return terms[0].second;
    while (terms.size() > MAX_TERMS) {
      std::vector<std::pair<int, int>> next;
      for (size_t i = 0; i < terms.size(); i += MAX_TERMS) {
        std::vector<std::pair<int, int>> chunk(terms.begin() + i, terms.begin() + std::min(terms.size(), i + MAX_TERMS));
        next.push_back({1, raw_lin(chunk)});
      }
      terms = next;
    }
    return raw_lin(terms); }
  int add(int a, int b) { return lin({{1, a}, {1, b}}); }
  int sub(int a, int b) { return lin({{1, a}, {-1, b}}); }
  int neg(int a) { return lin({{-1, a}}); }
 private:
  int raw_lin(const std::vector<std::pair<int, int>> &terms) {
    int lv = 0;
    for (auto &t : terms) lv = std::max(lv, nodes[t.second].level);
    nodes.push_back(Node{K_LIN8, 0, 0, terms, lv + 1, 0});
    return (int)nodes.size() - 1;
  }
};

// ---- the tower on node ids
// -------------------------------------------------------------------------------------------------------------------------------------
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
  F2 mul(const F2 &x, const F2 &y) {
    return {b.lin({{1, b.mul(x.c0, y.c0)}, {-1, b.mul(x.c1, y.c1)}}), b.lin({{1, b.mul(x.c0, y.c1)}, {1, b.mul(x.c1, y.c0)}})};
  }
  F2 sqr(const F2 &x) { return {b.lin({{1, b.mul(x.c0, x.c0)}, {-1, b.mul(x.c1, x.c1)}}), b.lin({{2, b.mul(x.c0, x.c1)}})}; }
  F2 mul_fq(const F2 &x, int k) { return {b.mul(x.c0, k), b.mul(x.c1, k)}; }
  F2 scale(const F2 &x, int k) { return {b.lin({{k, x.c0}}), b.lin({{k, x.c1}})}; }                       // small integer multiple
  F2 mul_xi(const F2 &x) { return {b.lin({{9, x.c0}, {-1, x.c1}}), b.lin({{9, x.c1}, {1, x.c0}})}; }
  // sum_i s_i * x_i over Fq2 with small integer coefficients, one LIN per component
  F2 lin2(const std::vector<std::pair<int, F2>> &t) {
    std::vector<std::pair<int, int>> a, c;
    for (auto &p : t) {
      a.push_back({p.first, p.second.c0});
      c.push_back({p.first, p.second.c1});
    }
    return {b.lin(a), b.lin(c)};
  }
  F2 inv(const F2 &x) { int n = b.add(b.mul(x.c0, x.c0), b.mul(x.c1, x.c1)), t = fq_inv(n); return {b.mul(x.c0, t), b.neg(b.mul(x.c1, t))}; }
  // x^(q-2) by 4-bit windows (Fermat; the reference uses mpn_gcdext, fp.tcc:688 — same value): a chain of ~330 dependent products, one lane busy
  int fq_inv(int x) { int tab[16]; tab[0] = b.one_node; tab[1] = x; for (int i = 2; i < 16; i++) tab[i] = b.mul(tab[i - 1], x);
    uint32_t e[8]; uint64_t br = 2; for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)FqParams::MOD[i] - br; e[i] = (uint32_t)t; br = (t >> 32) & 1; }
    int r = -1;
    for (int w = 63; w >= 0; w--) {
      if (r >= 0) for (int k = 0; k < 4; k++) r = b.mul(r, r);
      const uint32_t d = (e[w >> 3] >> ((w & 7) * 4)) & 15;
      if (d) r = r < 0 ? tab[d] : b.mul(r, tab[d]);
    }
    return r; }

  F12 one12() { F12 r; for (auto &c : r) c = zero2(); r[0] = one2(); return r; }
  // 36 Fq2 products (fewer when an operand is sparse: products with the zero node vanish), then per coefficient: lo + xi * hi
  F12 mul(const F12 &x, const F12 &y) {
    F12 r;
    for (int k = 0; k < 6; k++) { std::vector<std::pair<int, int>> lo0, lo1, hi0, hi1;
      for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { if (i + j != k && i + j != k + 6) continue; const bool hi = i + j == k + 6;
        int p00 = b.mul(x[i].c0, y[j].c0), p11 = b.mul(x[i].c1, y[j].c1), p01 = b.mul(x[i].c0, y[j].c1), p10 = b.mul(x[i].c1, y[j].c0);
        (hi ? hi0 : lo0).push_back({1, p00});
        (hi ? hi0 : lo0).push_back({-1, p11});
        (hi ? hi1 : lo1).push_back({1, p01});
        (hi ? hi1 : lo1).push_back({1, p10});
      }
      int l0 = b.lin(lo0), l1 = b.lin(lo1), h0 = b.lin(hi0), h1 = b.lin(hi1);
      r[k] = {b.lin({{1, l0}, {9, h0}, {-1, h1}}), b.lin({{1, l1}, {9, h1}, {1, h0}})}; }
    return r; }
  // unitary inverse: c1 -> -c1, i.e. the odd powers of w
  F12 conj(const F12 &x) {
    F12 r = x;
    for (int e = 1; e < 6; e += 2) r[e] = neg(x[e]);
    return r;
  }
  // fp12_2over3over2.tcc:139-147 over fp6_3over2.tcc:148-157
  F12 frob(const F12 &x, unsigned p) {
    const host::FrobeniusTables &t = host::frobenius_tables();
    F12 r;
    for (int e = 0; e < 6; e++) {
      const int i = e & 1, j = e >> 1;
      HFq2 k = HFq2::one();
      if (j == 1) k = t.fq6_c1[p % 6];
      if (j == 2) k = t.fq6_c2[p % 6];
      if (i) k = k * t.fq12_c1[p % 12];
      F2 v = (p & 1) ? conj(x[e]) : x[e]; r[e] = (j == 0 && i == 0) ? v : mul(v, k2(k)); }
    return r; }
  // Fq6 = Fq2[v]/(v^3 - xi) on {c0, c1, c2}: only what the Fq12 inverse needs
  typedef std::array<F2, 3> F6;
  F6 mul6(const F6 &x, const F6 &y) { F2 aA = mul(x[0], y[0]), bB = mul(x[1], y[1]), cC = mul(x[2], y[2]);
    return {add(aA, mul_xi(sub(sub(mul(add(x[1], x[2]), add(y[1], y[2])), bB), cC))), add(sub(sub(mul(add(x[0], x[1]), add(y[0], y[1])), aA), bB), mul_xi(cC)),
        add(sub(mul(add(x[0], x[2]), add(y[0], y[2])), aA), sub(bB, cC))};
  }
  F6 mul_by_v(const F6 &x) { return {mul_xi(x[2]), x[0], x[1]}; }
  F12 inv(const F12 &x) {   // fp12_2over3over2.tcc:128-137 over fp6_3over2.tcc:128-146
    F6 c0 = {x[0], x[2], x[4]}, c1 = {x[1], x[3], x[5]}, s0 = mul6(c0, c0), s1 = mul6(c1, c1), m = mul_by_v(s1), t = {sub(s0[0], m[0]), sub(s0[1], m[1]),
        sub(s0[2], m[2])};
    F2 t0 = sqr(t[0]), t1 = sqr(t[1]), t2 = sqr(t[2]), t3 = mul(t[0], t[1]), t4 = mul(t[0], t[2]), t5 = mul(t[1], t[2]), d0 = sub(t0, mul_xi(t5)),
        d1 = sub(mul_xi(t2), t3), d2 = sub(t1, t4);
    F2 t6 = inv(add(mul(t[0], d0), mul_xi(add(mul(t[2], d1), mul(t[1], d2)))));
    F6 ti = {mul(t6, d0), mul(t6, d1), mul(t6, d2)}, r0 = mul6(c0, ti), r1 = mul6(c1, ti);
    return {r0[0], neg(r1[0]), r0[1], neg(r1[1]), r0[2], neg(r1[2])}; }
  // the operand of mul_by_024 (fp12_2over3over2.tcc:240-335): (ell_0, 0, ell_VV | 0, ell_VW, 0) in the tower = w^0, w^4, w^3
  F12 sparse(const F2 &ell_0, const F2 &ell_VW, const F2 &ell_VV) {
    F12 r;
    for (auto &c : r) c = zero2();
    r[0] = ell_0;
    r[3] = ell_VW;
    r[4] = ell_VV;
    return r;
  }
};

static const uint64_t BN_Z = 4965661367192848881ull;                         // alt_bn128_init.cpp:327 (final_exponent_z)
static const uint64_t ATE_LOOP[2] = {0x9d797039be763ba8ull, 0x1ull};          // 6z+2 (alt_bn128_init.cpp:324)

// The whole check for one verifying key.  gamma / delta: the precomputed line coefficients of the vk's G2 points (host::precompute_g2).
inline Schedule build(const host::HFq12 &alpha_g1_beta_g2, const host::G2Precomp &gamma, const host::G2Precomp &delta) {
  Builder b; Tower T(b); const host::FrobeniusTables &ft = host::frobenius_tables();
  const HFq2 twist_b_v = HFq2{HFq::from_u64(3), HFq::zero()} * HFq2{HFq::from_u64(9), HFq::one()}.inv();
  const int two_inv = b.constant(HFq::from_u64(2).inv()), three = b.constant(HFq::from_u64(3));
  const F2 twist_b = T.k2(twist_b_v);
  const int ax = b.input(IN_AX), ay = b.input(IN_AY), cx = b.input(IN_CX), cy = b.input(IN_CY), nx = b.input(IN_NACCX), ny = b.input(IN_NACCY),
      nw = b.input(IN_NACCW);
  const F2 bx = {b.input(IN_BX0), b.input(IN_BX1)}, by = {b.input(IN_BY0), b.input(IN_BY1)};
  const int ncy = b.neg(cy);
  // is_well_formed: on-curve residues (alt_bn128_g1.cpp:92-117, alt_bn128_g2.cpp:98-127), must come out zero
  int chk[N_CHECK];
  chk[0] = b.lin({{1, b.mul(ay, ay)}, {-1, b.mul(b.mul(ax, ax), ax)}, {-1, three}});
  chk[1] = b.lin({{1, b.mul(cy, cy)}, {-1, b.mul(b.mul(cx, cx), cx)}, {-1, three}});
  { F2 r = T.sub(T.sub(T.sqr(by), T.mul(T.sqr(bx), bx)), twist_b); chk[2] = r.c0; chk[3] = r.c1; }
  // miller_loop :368-418 for e(A, B) with the running G2 point in homogeneous projective coordinates (:242-293), and the precomputed lines of gamma (at -acc)
  // and delta (at -C)
  F2 X = bx, Y = by, Z = T.one2(); F12 f = T.one12(); size_t idx = 0;
  // mul_by_q, alt_bn128_g2.cpp:367-372
  const F2 q1x = T.mul(T.k2(ft.twist_mul_by_q_x), T.conj(bx)), q1y = T.mul(T.k2(ft.twist_mul_by_q_y), T.conj(by)), q2x = T.mul(T.k2(ft.twist_mul_by_q_x),
      T.conj(q1x)), q2y = T.neg(T.mul(T.k2(ft.twist_mul_by_q_y), T.conj(q1y)));
  // the three sparse line values of a step — the running point's own, and the precomputed ones of gamma (at -acc) and delta (at -C) — are multiplied TOGETHER
  // off the chain of f: they depend on the G2 chain and on constants only, so the accumulator's chain is f <- f^2 * L per bit instead of four products in
  // sequence
  auto lines = [&](const F12 &own) -> F12 {
    if (idx >= gamma.size() || idx >= delta.size()) throw std::runtime_error("verify schedule: line count");
    const host::EllCoeffs &g = gamma[idx], &d = delta[idx]; idx++;
    const F12 lg = T.sparse(T.mul_fq(T.k2(g.ell_0), nw), T.mul_fq(T.k2(g.ell_VW), ny), T.mul_fq(T.k2(g.ell_VV), nx)), ld = T.sparse(T.k2(d.ell_0),
        T.mul_fq(T.k2(d.ell_VW), ncy), T.mul_fq(T.k2(d.ell_VV), cx));
    return T.mul(own, T.mul(lg, ld)); };
  auto dbl_step = [&]() {   // doubling_step_for_flipped_miller_loop :242-268
    F2 A = T.mul_fq(T.mul(X, Y), two_inv), B = T.sqr(Y), C = T.sqr(Z), D = T.scale(C, 3), E = T.mul(twist_b, D), F = T.scale(E, 3), G = T.mul_fq(T.add(B, F),
        two_inv), H = T.scale(T.mul(Y, Z), 2) /* = (Y + Z)^2 - (B + C) */, I = T.sub(E, B), J = T.sqr(X), E2 = T.sqr(E);
    X = T.mul(A, T.sub(B, F)); Y = T.sub(T.sqr(G), T.scale(E2, 3)); Z = T.mul(B, H);
    return T.sparse(T.mul_xi(I), T.mul_fq(T.neg(H), ay), T.mul_fq(T.scale(J, 3), ax)); };
  auto add_step = [&](const F2 &x2, const F2 &y2) {   // mixed_addition_step_for_flipped_miller_loop :270-293
    F2 D = T.sub(X, T.mul(x2, Z)), E = T.sub(Y, T.mul(y2, Z)), F = T.sqr(D), G = T.sqr(E), H = T.mul(D, F), I = T.mul(X, F), J = T.sub(T.add(H, T.mul(Z, G)),
        T.scale(I, 2)), Y1 = Y;
    X = T.mul(D, J); Y = T.sub(T.mul(E, T.sub(I, J)), T.mul(H, Y1)); Z = T.mul(Z, H);
    return T.sparse(T.mul_xi(T.sub(T.mul(E, x2), T.mul(D, y2))), T.mul_fq(D, ay), T.mul_fq(T.neg(E), ax)); };
  bool found = false;
  for (int i = 127; i >= 0; i--) { const bool bit = (ATE_LOOP[i / 64] >> (i % 64)) & 1; if (!found) { found |= bit; continue; }
    F12 L = lines(dbl_step()); if (bit) L = T.mul(L, lines(add_step(bx, by))); f = T.mul(T.mul(f, f), L); }
  { F12 L1 = lines(add_step(q1x, q1y)), L2 = lines(add_step(q2x, q2y)); f = T.mul(f, T.mul(L1, L2)); }
  if (idx != gamma.size() || idx != delta.size()) throw std::runtime_error("verify schedule: line count");
  // final_exponentiation :110-238: first chunk f^((q^6 - 1)(q^2 + 1)), then the last chunk's chain with three exponentiations by -z. Squarings in the
  // cyclotomic subgroup are plain squarings here: with all 144 coefficient products of a round in parallel the Granger-Scott form saves nothing.
  // alt_bn128_pairing.cpp:84-96: conj(src^z), src in the cyclotomic subgroup. Width-3 NAF of z: 18 non-zero digits in {+-1, +-3}; a negative digit is a
  // conjugation
  auto exp_neg_z = [&](const F12 &src) {
    std::vector<int> dig;
    for (uint64_t k = BN_Z; k;) {
      int t = 0;
      if (k & 1) {
        t = (int)(k & 7);
        if (t >= 4) t -= 8;
        k -= (uint64_t)(int64_t)t;
      }
      dig.push_back(t);
      k >>= 1;
    }
    const F12 s3 = T.mul(T.mul(src, src), src), sc = T.conj(src), s3c = T.conj(s3); F12 r = T.one12(); bool started = false;
    for (size_t i = dig.size(); i-- > 0;) {
      const int d = dig[i];
      if (started) r = T.mul(r, r);
      if (!d) continue;
      const F12 &m = d == 1 ? src : d == -1 ? sc : d == 3 ? s3 : s3c;
      r = started ? T.mul(r, m) : m;
      started = true;
    }
    return T.conj(r); };
  F12 c0 = T.mul(T.conj(f), T.inv(f)), first = T.mul(T.frob(c0, 2), c0);
  F12 A = exp_neg_z(first), Bq = T.mul(A, A), Cq = T.mul(Bq, Bq), D = T.mul(Cq, Bq), E = exp_neg_z(D), Fq_ = T.mul(E, E), G = exp_neg_z(Fq_), H = T.conj(D),
      I = T.conj(G), J = T.mul(I, E), K = T.mul(J, H), L = T.mul(K, Bq), M = T.mul(K, E), N = T.mul(M, first),
      O = T.frob(L, 1), P = T.mul(O, N), Q = T.frob(K, 2), R = T.mul(Q, P), S = T.conj(first), Tt = T.mul(S, L), U = T.frob(Tt, 3), V = T.mul(U, R);
  // outputs in tower order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 — each c0 then c1) MINUS vk.alpha_g1_beta_g2: sixteen values that must all be zero
  std::vector<int> outs;
  {
    for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) {
      const F2 &c = V[2 * j + i];
      const HFq2 *c2 = i == 0 ? &alpha_g1_beta_g2.c0.c0 + j : &alpha_g1_beta_g2.c1.c0 + j;
      outs.push_back(b.lin({{1, c.c0}, {-1, b.constant(c2->c0)}})); outs.push_back(b.lin({{1, c.c1}, {-1, b.constant(c2->c1)}})); } }
  for (int k = 0; k < N_CHECK; k++) outs.push_back(chk[k]);

  // ---- liveness, rounds, slots
  // -----------------------------------------------------------------------------------------------------------------------------------
  const size_t nn = b.nodes.size(); std::vector<char> live(nn, 0); for (int o : outs) live[o] = 1;
  for (size_t n = nn; n-- > 0;) {
    if (!live[n]) continue;
    const Builder::Node &nd = b.nodes[n];
    if (nd.kind == K_MUL) {
      live[nd.a] = live[nd.b] = 1;
    } else if (nd.kind == K_LIN8) for (auto &t : nd.terms) live[t.second] = 1;
  }
  for (int o : outs) if (b.nodes[o].kind != K_LIN8) throw std::runtime_error("verify schedule: output is not a computed difference");
  {
    std::vector<int> seen(outs);
    std::sort(seen.begin(), seen.end());
    if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) throw std::runtime_error("verify schedule: two outputs share a value");
  }
  int max_level = 0; for (size_t n = 0; n < nn; n++) if (live[n]) max_level = std::max(max_level, b.nodes[n].level);
  // as LATE as possible: a value is placed one level before its first reader (outputs at the last level). Scheduled as early as possible, everything off the
  // critical chain — the G2 point's chain, the products of the line values — ran hundreds of levels ahead of its readers and the live values did not fit the
  // LDS (3,769 slots).
  { std::vector<int> alap(nn, max_level + 1); for (int o : outs) alap[o] = max_level;
    for (size_t n = nn; n-- > 0;) {
      if (!live[n]) continue;
      Builder::Node &nd = b.nodes[n];
      if (alap[n] > max_level) alap[n] = max_level;
      const int lv = alap[n];
      if (nd.kind == K_MUL) {
        alap[nd.a] = std::min(alap[nd.a], lv - 1);
        alap[nd.b] = std::min(alap[nd.b], lv - 1);
      } else if (nd.kind == K_LIN8) for (auto &t : nd.terms) alap[t.second] = std::min(alap[t.second], lv - 1);
    }
    for (size_t n = 0; n < nn; n++) if (live[n] && (b.nodes[n].kind == K_MUL || b.nodes[n].kind == K_LIN8)) {
      if (alap[n] < b.nodes[n].level) throw std::runtime_error("verify schedule: level");
      b.nodes[n].level = alap[n];
    }
  }
  std::vector<std::vector<int>> mul_at(max_level + 1), lin_at(max_level + 1);
  for (size_t n = 0; n < nn; n++) if (live[n]) {
    if (b.nodes[n].kind == K_MUL) mul_at[b.nodes[n].level].push_back((int)n);
    else if (b.nodes[n].kind == K_LIN8) lin_at[b.nodes[n].level].push_back((int)n);
  }
  // rounds of a level: products, 256 a round; then the sums. A sum of up to three terms needs one lane (LIN1, 256 a round), a longer one eight (LIN8, 32 a
  // round); short sums ride in the spare groups of the level's LIN8 rounds when they all fit there (a round costs more than the lanes it leaves idle)
  struct Round { uint32_t kind; std::vector<int> ns; }; std::vector<Round> rounds;
  for (int lv = 1; lv <= max_level; lv++) {
    for (size_t i = 0; i < mul_at[lv].size(); i += LANES) rounds.push_back(Round{K_MUL, std::vector<int>(mul_at[lv].begin() + i,
        mul_at[lv].begin() + std::min(mul_at[lv].size(), i + LANES))});
    std::vector<int> wide, narrow; for (int n : lin_at[lv]) (b.nodes[n].terms.size() > TERMS_PER_LANE ? wide : narrow).push_back(n);
    const size_t per8 = LANES / LIN_GROUP, spare = wide.empty() ? 0 : (per8 - wide.size() % per8) % per8;
    if (!wide.empty() && narrow.size() <= spare) { wide.insert(wide.end(), narrow.begin(), narrow.end()); narrow.clear(); }
    for (size_t i = 0; i < wide.size(); i += per8) rounds.push_back(Round{K_LIN8, std::vector<int>(wide.begin() + i, wide.begin() + std::min(wide.size(),
        i + per8))});
    for (size_t i = 0; i < narrow.size(); i += LANES) rounds.push_back(Round{K_LIN1, std::vector<int>(narrow.begin() + i,
        narrow.begin() + std::min(narrow.size(), i + LANES))});
  }
  std::vector<int> last_use(nn, -1); std::vector<char> is_out(nn, 0); for (int o : outs) is_out[o] = 1;
  for (size_t r = 0; r < rounds.size(); r++) for (int n : rounds[r].ns) {
    const Builder::Node &nd = b.nodes[n];
    if (nd.kind == K_MUL) {
      last_use[nd.a] = last_use[nd.b] = (int)r;
    } else for (auto &t : nd.terms) last_use[t.second] = (int)r;
  }
  Schedule sc;
  sc.consts = b.consts;
  std::vector<int> slot(nn, -1);
  for (int i = 0; i < N_INPUTS; i++) slot[i] = i;
  std::vector<int> free_slots;
  uint32_t next_slot = N_INPUTS;
  if (sc.consts.size() >= CONST_FLAG) throw std::runtime_error("verify schedule: too many constants");
  auto ref = [&](int n) -> uint32_t { const Builder::Node &nd = b.nodes[n]; if (nd.kind == 3) return CONST_FLAG | nd.cidx;
      if (slot[n] < 0) throw std::runtime_error("verify schedule: value read before it was written"); return (uint32_t)slot[n]; };
  const uint32_t no_term = CONST_FLAG | b.nodes[b.zero_node].cidx;   // coefficient 0 on the constant zero
  auto term_word = [&](const std::pair<int, int> &t) -> uint32_t { const uint32_t c = (uint32_t)(t.first < 0 ? -t.first : t.first);
      if (!c || c > MAX_COEF) throw std::runtime_error("verify schedule: coefficient"); return ref(t.second) | (t.first < 0 ? 1u << 16 : 0u) | c << 17; };
  std::vector<size_t> hdr_pos;
  for (size_t r = 0; r < rounds.size(); r++) {
    const Round &rd = rounds[r];
    const uint32_t lanes_per = rd.kind == K_LIN8 ? LIN_GROUP : 1, lanes = (uint32_t)rd.ns.size() * lanes_per;
    // (four header words: every lane's words stay 16-byte aligned; word 1 = the NEXT round's header, filled in below)
    hdr_pos.push_back(sc.prog.size());
    sc.prog.push_back(hdr(rd.kind, lanes));
    sc.rounds_of_kind[rd.kind]++;
    sc.prog.insert(sc.prog.end(), 3, 0u);
    std::vector<uint32_t> words((size_t)lanes * WPL, 0u);   // (operands are resolved BEFORE this round's destinations are assigned)
    for (size_t k = 0; k < rd.ns.size(); k++) { const Builder::Node &nd = b.nodes[rd.ns[k]];
      if (rd.kind == K_MUL) { words[k * WPL + 1] = ref(nd.a); words[k * WPL + 2] = ref(nd.b); sc.n_mul++; }
      else {
        for (uint32_t l = 0; l < lanes_per; l++) for (uint32_t t = 0; t < TERMS_PER_LANE; t++) {
          const size_t idx = (size_t)t * lanes_per + l;
          words[(k * lanes_per + l) * WPL + 1 + t] = idx < nd.terms.size() ? term_word(nd.terms[idx]) : no_term;
        }
        sc.n_lin++;
      }
    }
    // destinations come from the slots freed in EARLIER rounds only (this round's operands are released below, after the assignment): no lane writes a slot
    // that another lane of the same round still reads, so the kernel needs one barrier per round, not two
    for (size_t k = 0; k < rd.ns.size(); k++) {
      int s;
      if (!free_slots.empty()) {
        s = free_slots.back();
        free_slots.pop_back();
      } else s = (int)next_slot++;
      slot[rd.ns[k]] = s;
      for (uint32_t l = 0; l < lanes_per; l++) words[(k * lanes_per + l) * WPL] = (uint32_t)s;
    }
    { std::vector<char> is_dst(next_slot, 0); for (int n : rd.ns) is_dst[slot[n]] = 1;
      for (size_t q = 0; q < words.size(); q++) {
        const uint32_t pos = (uint32_t)(q % WPL), wv = words[q];
        const bool operand = pos && (rd.kind == K_MUL ? pos < 3 : (wv >> 17) != 0);
        if (operand && !(wv & CONST_FLAG) && is_dst[wv & 0x7fffu]) throw std::runtime_error("verify schedule: a round writes a slot it reads"); } }
    sc.prog.insert(sc.prog.end(), words.begin(), words.end());
    for (int n : rd.ns) {
      const Builder::Node &nd = b.nodes[n];
      auto release = [&](int o) {
        if (b.nodes[o].kind != 3 && o >= N_INPUTS && last_use[o] == (int)r && !is_out[o] && slot[o] >= 0) {
          free_slots.push_back(slot[o]);
          slot[o] = -2;
        }
      };
      if (nd.kind == K_MUL) { release(nd.a); release(nd.b); } else for (auto &t : nd.terms) release(t.second); }
  }
  // look-ahead: a round's header also carries the next one's, so the kernel can fetch the next round's words without waiting for a header load
  for (size_t r = 0; r + 1 < hdr_pos.size(); r++) sc.prog[hdr_pos[r] + 1] = sc.prog[hdr_pos[r + 1]];
  sc.n_rounds = (uint32_t)rounds.size(); sc.n_slots = next_slot; for (size_t k = 0; k < outs.size(); k++) sc.out_slot[k] = (uint32_t)slot[outs[k]];
  return sc;
}

// Host interpreter of a schedule on the host field type: what the program means. in: the N_INPUTS values of one proof. Returns the output values (N_RESULT +
// N_CHECK), all zero for a valid proof.
inline std::vector<HFq> simulate(const Schedule &S, const HFq *in) {
  std::vector<HFq> slots(S.n_slots, HFq::zero()); for (int i = 0; i < N_INPUTS; i++) slots[i] = in[i];
  auto val = [&](uint32_t r) { return (r & CONST_FLAG) ? S.consts[r & (CONST_FLAG - 1)] : slots[r & 0xffff]; };
  auto times = [](HFq x, uint32_t c) { HFq r = HFq::zero(); for (; c; c >>= 1) { if (c & 1) r = r + x; x = x + x; } return r; };
  size_t pc = 0; std::vector<std::pair<uint32_t, HFq>> writes;
  for (uint32_t r = 0; r < S.n_rounds; r++) {
    uint32_t kind, count;
    unhdr(S.prog[pc], kind, count);
    pc += 4;
    writes.clear();
    const uint32_t per = kind == K_LIN8 ? LIN_GROUP : 1;
    for (uint32_t g = 0; g < count / per; g++) { const uint32_t *w = &S.prog[pc + (size_t)g * per * WPL]; HFq v = HFq::zero();
      if (kind == K_MUL) v = val(w[1]) * val(w[2]);
      else for (uint32_t l = 0; l < per; l++) for (uint32_t t = 0; t < TERMS_PER_LANE; t++) {
        const uint32_t e = w[l * WPL + 1 + t];
        const HFq x = times(val(e & 0xffff), e >> 17);
        v = ((e >> 16) & 1) ? v - x : v + x;
      }
      writes.push_back({w[0], v}); }
    for (auto &wv : writes) slots[wv.first] = wv.second;   // (all reads of a round happen before its writes, as in the lock-step workgroup)
    pc += (size_t)count * WPL; }
  std::vector<HFq> out; for (int k = 0; k < N_RESULT + N_CHECK; k++) out.push_back(slots[S.out_slot[k]]); return out;
}

// The constants as the kernel wants them: c 2^261 mod p (the host type holds c 2^256: five doublings), canonical, nine limbs in a 12-word record
inline std::vector<uint32_t> consts29(const Schedule &S) { std::vector<uint32_t> out(S.consts.size() * l29::STRIDE, 0u);
  for (size_t k = 0; k < S.consts.size(); k++) {
    HFq v = S.consts[k];
    for (int i = 0; i < 5; i++) v = v + v;
    uint32_t w[8], l[9];
    memcpy(w, v.l, 32);
    l29::unpack(w, l);
    memcpy(&out[k * l29::STRIDE], l, 36);
  }
  return out; }
// Host interpreter on the DEVICE's arithmetic: the kernel's operations limb by limb (l29::term / norm64 / the lane tree / barrett, and the model of the 29-bit
// product), every intermediate bound asserted. in_words: the N_INPUTS values as eight 32-bit words each (Montgomery 2^256, canonical — what the kernel is
// handed). Returns for each output whether it is a multiple of p (what the kernel tests).
inline std::vector<bool> simulate29(const Schedule &S, const uint32_t (*in_words)[8]) {
  typedef std::array<uint32_t, 9> V; const std::vector<uint32_t> c29 = consts29(S); std::vector<V> slots(S.n_slots, V{});
  auto bad = [](const char *what) { throw std::runtime_error(std::string("verify schedule (29-bit model): ") + what); };
  // (p >> 232 is just above 3 * 2^20: 5 * 2^22 bounds 6.6 p)
  auto check_stored = [&](const uint32_t (&l)[9]) {
    for (int i = 0; i < 8; i++) if (l[i] >= (1u << 29) + 8) bad("a stored limb is not normalized");
    if (l[8] >= (5u << 22)) bad("a stored value is not below 5 p");
  };
  for (int i = 0; i < N_INPUTS; i++) { uint32_t l[9]; l29::lift(in_words[i], l); check_stored(l); memcpy(slots[i].data(), l, 36); }
  auto load = [&](uint32_t r, uint32_t (&x)[9]) {
    if (r & CONST_FLAG) memcpy(x, &c29[(size_t)(r & (CONST_FLAG - 1)) * l29::STRIDE], 36);
    else memcpy(x, slots[r & 0xffff].data(), 36);
  };
  size_t pc = 0; std::vector<std::pair<uint32_t, V>> writes;
  for (uint32_t r = 0; r < S.n_rounds; r++) {
    uint32_t kind, count;
    unhdr(S.prog[pc], kind, count);
    pc += 4;
    writes.clear();
    const uint32_t per = kind == K_LIN8 ? LIN_GROUP : 1;
    for (uint32_t g = 0; g < count / per; g++) { const uint32_t *w = &S.prog[pc + (size_t)g * per * WPL]; uint32_t res[9];
      if (kind == K_MUL) { uint32_t a[9], bb[9]; load(w[1], a); load(w[2], bb); l29::mul_model(a, bb, res); }
      else { uint32_t part[LIN_GROUP][9];
        for (uint32_t l = 0; l < per; l++) { uint64_t acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
          for (uint32_t t = 0; t < TERMS_PER_LANE; t++) {
            const uint32_t e = w[l * WPL + 1 + t];
            uint32_t x[9];
            load(e & 0xffff, x);
            const bool neg = (e >> 16) & 1;
            if (neg) for (int i = 0; i < 9; i++) if (x[i] > p29::K6[i]) bad("K6 - x has a negative limb");
            l29::term(acc, x, e >> 17, neg);
          }
          for (int i = 0; i < 9; i++) { if (acc[i] >> 40) bad("a limb accumulator is above 2^40"); }
          l29::norm64(acc, part[l]); }
        if (per == LIN_GROUP) {   // the kernel's tree: lanes 4..7 += lanes 0..3, a carry step, then lane 6 += 4, 7 += 5, 7 += 6
          uint32_t hi[4][9];
          for (int j = 0; j < 4; j++) for (int i = 0; i < 9; i++) {
            const uint64_t s2 = (uint64_t)part[j][i] + part[j + 4][i];
            if (s2 >> 32) bad("tree level 1 overflows");
            hi[j][i] = (uint32_t)s2;
          }
          for (int i = 0; i < 9; i++) {
            const uint64_t s2 = (uint64_t)hi[0][i] + hi[2][i], s3 = (uint64_t)hi[1][i] + hi[3][i];
            if ((s2 | s3) >> 32) bad("tree level 2 overflows");
            hi[2][i] = (uint32_t)s2;
            hi[3][i] = (uint32_t)s3;
          }
          l29::norm32(hi[2]); l29::norm32(hi[3]);
          for (int i = 0; i < 9; i++) {
            const uint64_t s2 = (uint64_t)hi[2][i] + hi[3][i];
            if (s2 >> 32) bad("tree level 3 overflows");
            res[i] = (uint32_t)s2;
          }
        }
        else memcpy(res, part[0], 36);
        for (int i = 0; i < 8; i++) if (res[i] >> 31) bad("a limb entering the reduction step is above 2^31");
        l29::barrett(res); }
      check_stored(res); V v; memcpy(v.data(), res, 36); writes.push_back({w[0], v}); }
    for (auto &wv : writes) slots[wv.first] = wv.second;
    pc += (size_t)count * WPL; }
  std::vector<bool> out;
  for (int k = 0; k < N_RESULT + N_CHECK; k++) {
    uint32_t l[9];
    memcpy(l, slots[S.out_slot[k]].data(), 36);
    out.push_back(l29::multiple_of_p(l));
  }
  return out;
}

} }  // namespace zk::vsched
