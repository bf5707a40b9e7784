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
#include <functional>
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

// A ROUND is one operation per lane of a 256-thread workgroup, decided WAVE by wave (round 6): each of the four waves of a round executes one kind of operation
// (products / eight-lane sums / one-lane sums / nothing), so a level of the dependency graph that holds 78 products of one chain and a dozen sums of another is
// still one round.  The program is a dense array: round r, lane l -> four words at (r * 256 + l) * 4, no headers, no counts — the kernel's fetch of round r + D
// depends on nothing but r (a ring of D rounds in flight; the first generation's "next round" fetch sat behind a header load and a branch and was in effect
// synchronous: ~0.5 us of every round was the wait for it).
//     word 0: destination slot | live << 16 | store << 17 | kind << 30 (the wave's kind, in every lane's word, idle lanes included)
//     MUL    slot[dst] = slot[a] * slot[b]                         (words 1, 2: a, b)
//     LIN8   slot[dst] = sum of up to 24 terms c * slot[src], |c| <= 255   (eight lanes per value, three terms each — words 1..3 —, then a tree over the eight lanes; the
//                                                                            group's last lane stores)
//     LIN1   the same with up to 3 terms, one lane per value
//     term word: src | neg << 16 | |c| << 17; c = 0: no term
constexpr uint32_t CONST_FLAG = 0x8000u, LIN_GROUP = 8, TERMS_PER_LANE = 3, MAX_TERMS = LIN_GROUP * TERMS_PER_LANE, MAX_COEF = 255, LANES = 256, WAVE = 64,
    WAVES = LANES / WAVE;
constexpr uint32_t WPL = 4, LIVE_BIT = 1u << 16, STORE_BIT = 1u << 17, KIND_SHIFT = 30;
constexpr uint32_t PREFETCH_ROUNDS = 4;   // the kernel's ring of rounds in flight: the program is padded to a multiple of this many rounds, plus as many idle rounds again
enum : uint32_t { K_IDLE = 0, K_MUL = 1, K_LIN8 = 2, K_LIN1 = 3 };
// per-proof inputs: fixed slots 0 .. N_INPUTS-1
enum : int { IN_AX = 0, IN_AY, IN_BX0, IN_BX1, IN_BY0, IN_BY1, IN_CX, IN_CY, IN_NACCX, IN_NACCY, IN_NACCW, N_INPUTS };
// (IN_NACC*: the negated input accumulator -acc = (x, -y) given as (x w, -y w, w) for any non-zero w in Fq — the gamma line values are evaluated times w, a
// factor the final exponentiation kills — so that the accumulation kernel needs no inversion: w = ZZ ZZZ of its extended Jacobian sum)
// outputs 0 .. 15 must be ZERO (mod p): the GT comparison (tower order), then A, C on the curve and B on the twist (two components); output 16 must NOT be zero:
// the norm of the Miller value (below)
constexpr int N_RESULT = 12, N_CHECK = 4, N_ZERO = N_RESULT + N_CHECK, OUT_NONZERO = N_ZERO, N_OUT = N_ZERO + 1;

struct Schedule {
  std::vector<uint32_t> prog;        // n_rounds_padded * LANES * WPL words
  std::vector<HFq> consts;           // CONST_FLAG | index
  uint32_t n_rounds = 0 /* with work */, n_rounds_padded = 0, n_slots = 0, n_mul = 0, n_lin = 0, waves_of_kind[4] = {0, 0, 0, 0}, out_slot[N_OUT] = {0};
  uint32_t n_levels = 0; std::vector<uint32_t> round_level;   // diagnostics: the dependency depth, and the level every round belongs to
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

  F12 one12() { F12 r; for (auto &c : r) c = zero2(); r[0] = one2(); return r; }
  // 36 Fq2 products (fewer when an operand is sparse: products with the zero node vanish; a square's symmetric products are one node), then per coefficient:
  // lo + xi * hi
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
  // Fq6 = Fq2[v]/(v^3 - xi) on {c0, c1, c2}: only what the inverse-free Fq12 "inverse" needs. Schoolbook (nine Fq2 products, ONE level of sums — the lanes are
  // there, the depth is what counts): (a0 b0 + xi (a1 b2 + a2 b1), a0 b1 + a1 b0 + xi a2 b2, a0 b2 + a1 b1 + a2 b0)
  typedef std::array<F2, 3> F6;
  F6 mul6_xi(const F6 &x, const F6 &y) {
    const F2 a = mul(x[0], y[0]), bb = mul(x[1], y[2]), c = mul(x[2], y[1]), d = mul(x[0], y[1]), e = mul(x[1], y[0]), f = mul(x[2], y[2]), g = mul(x[0], y[2]),
        h = mul(x[1], y[1]), i = mul(x[2], y[0]);
    // xi * t = (9 t0 - t1, 9 t1 + t0)
    auto plus_xi = [&](const std::vector<F2> &plain, const std::vector<F2> &by_xi) { std::vector<std::pair<int, int>> r0, r1;
      for (auto &p : plain) { r0.push_back({1, p.c0}); r1.push_back({1, p.c1}); }
      for (auto &t : by_xi) { r0.push_back({9, t.c0}); r0.push_back({-1, t.c1}); r1.push_back({9, t.c1}); r1.push_back({1, t.c0}); }
      return F2{b.lin(r0), b.lin(r1)}; };
    return {plus_xi({a}, {bb, c}), plus_xi({d, e}, {f}), plus_xi({g, h, i}, {})};
  }
  F6 mul_by_v(const F6 &x) { return {mul_xi(x[2]), x[0], x[1]}; }
  // The inverse of the Miller value WITHOUT an inversion: x^-1 = P / n with P in Fq12 and n in Fq, both made of products only — n is the norm of x down to Fq
  // (fp12_2over3over2.tcc:128-137 over fp6_3over2.tcc:128-146 over fp2.tcc's inverse, every division carried along as a denominator). Returns P, sets n.
  F12 pseudo_inverse(const F12 &x, int &n) {
    F6 c0 = {x[0], x[2], x[4]}, c1 = {x[1], x[3], x[5]}, s0 = mul6_xi(c0, c0), s1 = mul6_xi(c1, c1), m = mul_by_v(s1), t = {sub(s0[0], m[0]), sub(s0[1], m[1]),
        sub(s0[2], m[2])};
    F2 t0 = sqr(t[0]), t1 = sqr(t[1]), t2 = sqr(t[2]), t3 = mul(t[0], t[1]), t4 = mul(t[0], t[2]), t5 = mul(t[1], t[2]), d0 = sub(t0, mul_xi(t5)),
        d1 = sub(mul_xi(t2), t3), d2 = sub(t1, t4);
    F2 N = add(mul(t[0], d0), mul_xi(add(mul(t[2], d1), mul(t[1], d2))));                 // the norm of t down to Fq2; t^-1 = (d0, d1, d2) / N
    n = b.add(b.mul(N.c0, N.c0), b.mul(N.c1, N.c1));                                      // N^-1 = conj(N) / n
    const F2 Nc = conj(N); F6 ti = {mul(Nc, d0), mul(Nc, d1), mul(Nc, d2)}, r0 = mul6_xi(c0, ti), r1 = mul6_xi(c1, ti);
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

// little-endian 32-bit digits, enough for 2 (48 z^3 + 30 z^2 + 28 z + 2)
struct Big { std::vector<uint32_t> d;
  static Big of(uint64_t v) { Big r; r.d = {(uint32_t)v, (uint32_t)(v >> 32)}; return r; }
  Big operator*(const Big &o) const { Big r; r.d.assign(d.size() + o.d.size(), 0u);
    for (size_t i = 0; i < d.size(); i++) { uint64_t c = 0; for (size_t j = 0; j < o.d.size() || c; j++) { const uint64_t t = (uint64_t)r.d[i + j] + c + (j < o.d.size() ? (uint64_t)d[i] * o.d[j] : 0); r.d[i + j] = (uint32_t)t; c = t >> 32; } }
    return r; }
  Big operator+(const Big &o) const { Big r; r.d.assign(std::max(d.size(), o.d.size()) + 1, 0u); uint64_t c = 0;
    for (size_t i = 0; i < r.d.size(); i++) { c += (i < d.size() ? d[i] : 0u); c += (i < o.d.size() ? o.d[i] : 0u); r.d[i] = (uint32_t)c; c >>= 32; } return r; }
  int bits() const { for (size_t i = d.size(); i-- > 0;) if (d[i]) return (int)(32 * i + 32 - __builtin_clz(d[i])); return 0; }
  bool bit(int i) const { return (size_t)(i >> 5) < d.size() && ((d[i >> 5] >> (i & 31)) & 1); }
};

// The whole check for one verifying key.  gamma / delta: the precomputed line coefficients of the vk's G2 points (host::precompute_g2).
inline Schedule build(const host::HFq12 &alpha_g1_beta_g2, const host::G2Precomp &gamma, const host::G2Precomp &delta) {
  Builder b; Tower T(b); const host::FrobeniusTables &ft = host::frobenius_tables();
  const HFq2 twist_b_v = HFq2{HFq::from_u64(3), HFq::zero()} * HFq2{HFq::from_u64(9), HFq::one()}.inv();
  const int three = b.constant(HFq::from_u64(3));
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
  // and delta (at -C).
  // The chain of the running point is the Miller loop's critical path (the accumulator f needs four levels a bit, f <- f^2 * L): it is written for DEPTH here.
  // State (X, Y, Z, W) with W = 3 b' Z carried along (b' = the twist's coefficient): E = b' * 3 Z^2 of :247-249 is the product Z W — no constant product on the
  // chain — and the halvings of :245,251 are dropped: the doubling leaves 4 (X3, Y3, Z3), a representative of the same projective point, and every line of a
  // later step comes out times a power of 4, a factor in Fq that the final exponentiation kills.  A doubling is two product levels (the degree of its
  // formulas, 4, allows no less), a mixed addition three (:270-293 in four: H = D F, then J, then X3 = D J — here X3 = F^2 + (D Z) G - 2 (D X) F etc., the same
  // values from products of the previous level's products).
  F2 X = bx, Y = by, Z = T.one2(), W = T.k2(twist_b_v + twist_b_v + twist_b_v); F12 f = T.one12(); size_t idx = 0;
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
  auto dbl_step = [&]() {   // doubling_step_for_flipped_miller_loop :242-268, times 4 (see above)
    const F2 XY = T.mul(X, Y), B = T.sqr(Y), ZW = T.mul(Z, W) /* E */, YZ = T.mul(Y, Z) /* H / 2 */, J = T.sqr(X), YW = T.mul(Y, W),
        BmF = T.lin2({{1, B}, {-3, ZW}}), BpF = T.lin2({{1, B}, {3, ZW}}), I = T.sub(ZW, B);
    const F12 line = T.sparse(T.mul_xi(I), T.mul_fq(T.scale(YZ, -2), ay), T.mul_fq(T.scale(J, 3), ax));
    X = T.scale(T.mul(XY, BmF), 2); Y = T.lin2({{1, T.sqr(BpF)}, {-12, T.sqr(ZW)}}); Z = T.scale(T.mul(B, YZ), 8); W = T.scale(T.mul(B, YW), 8);
    return line; };
  auto add_step = [&](const F2 &x2, const F2 &y2) {   // mixed_addition_step_for_flipped_miller_loop :270-293, the same values in three product levels
    const F2 D = T.sub(X, T.mul(x2, Z)), E = T.sub(Y, T.mul(y2, Z)), F = T.sqr(D), G = T.sqr(E), DX = T.mul(D, X), DZ = T.mul(D, Z), DW = T.mul(D, W),
        EX = T.mul(E, X), ED = T.mul(E, D), DY = T.mul(D, Y), EZ = T.mul(E, Z), Wn = T.lin2({{3, EX}, {-1, ED}, {-1, DY}});
    const F12 line = T.sparse(T.mul_xi(T.sub(T.mul(E, x2), T.mul(D, y2))), T.mul_fq(D, ay), T.mul_fq(T.neg(E), ax));
    X = T.lin2({{1, T.sqr(F)}, {1, T.mul(DZ, G)}, {-2, T.mul(DX, F)}}); Y = T.sub(T.mul(F, Wn), T.mul(EZ, G)); Z = T.mul(DZ, F); W = T.mul(DW, F);
    return line; };
  bool found = false;
  for (int i = 127; i >= 0; i--) { const bool bit = (ATE_LOOP[i / 64] >> (i % 64)) & 1; if (!found) { found |= bit; continue; }
    F12 L = lines(dbl_step()); if (bit) L = T.mul(L, lines(add_step(bx, by))); f = T.mul(T.mul(f, f), L); }
  { F12 L1 = lines(add_step(q1x, q1y)), L2 = lines(add_step(q2x, q2y)); f = T.mul(f, T.mul(L1, L2)); }
  if (idx != gamma.size() || idx != delta.size()) throw std::runtime_error("verify schedule: line count");
  // final_exponentiation :110-238 WITHOUT the inversion of its first chunk (:116-119, the one Fq inversion of the whole check: 254 dependent squarings with one
  // lane busy, a fifth of the first generation's rounds).  f^-1 = P / n, n in Fq (Tower::pseudo_inverse); the chain is run on c0' = conj(f) P = f^(q^6-1) n instead
  // of f^(q^6-1).  Every later operation — product, conjugation, Frobenius map — is multiplicative and fixes Fq, so the scalar rides along: the result is
  // V' = V n^e with an exponent e that the chain itself determines (below), and V == alpha_beta  <=>  V' == alpha_beta n^e, provided n != 0 — which holds unless
  // f = 0 (the norm of a field element) and is an output of its own: a proof whose Miller value is zero is rejected, as it was when 0^-1 came out 0.  The power
  // n^e is an Fq chain of one lane per level beside the last chunk's Fq12 chain: off the critical path.
  // Squarings in the cyclotomic subgroup are plain squarings here (with all coefficient products of a round in parallel the Granger-Scott form saves nothing, and
  // c0' is not in the subgroup).
  // alt_bn128_pairing.cpp:84-96: conj(src^z).  The critical path is the 62 squarings; the products with the powers src^(2^i) of the one bits are a second chain
  // that keeps pace with the first (a product per squaring at most) and joins it after the last squaring: 63 levels of Fq12 products instead of 80 for the
  // width-3 NAF chain of the first generation.  Plain bits (no negative digits): a conjugate would be the inverse for the subgroup part but not for the scalar.
  auto exp_neg_z = [&](const F12 &src) {
    F12 pw = src, acc{}; bool started = false;
    for (int i = 0; i < 64; i++) {
      if (!(BN_Z >> i)) break;
      if (i) pw = T.mul(pw, pw);
      if ((BN_Z >> i) & 1) { acc = started ? T.mul(acc, pw) : pw; started = true; }
    }
    return T.conj(acc); };
  int n = 0; const F12 P = T.pseudo_inverse(f, n);
  F12 c0 = T.mul(T.conj(f), P), first = T.mul(T.frob(c0, 2), c0);
  F12 A = exp_neg_z(first), Bq = T.mul(A, A), Cq = T.mul(Bq, Bq), D = T.mul(Cq, Bq), E = exp_neg_z(D), Fq_ = T.mul(E, E), G = exp_neg_z(Fq_), H = T.conj(D),
      I = T.conj(G), J = T.mul(I, E), K = T.mul(J, H), L = T.mul(K, Bq), M = T.mul(K, E), N = T.mul(M, first),
      O = T.frob(L, 1), Pp = T.mul(O, N), Q = T.frob(K, 2), R = T.mul(Q, Pp), S = T.conj(first), Tt = T.mul(S, L), U = T.frob(Tt, 3), V = T.mul(U, R);
  // the scalar's exponent: first' = first n^2 (n^(q^2) = n); through the chain above — a power z per exp_neg_z, 1 per conjugation and Frobenius map —
  // A: z, B: 2z, C: 4z, D: 6z, E: 6z^2, F: 12z^2, G: 12z^3, J: 12z^3 + 6z^2, K: J + 6z, L: K + 2z, M: K + 6z^2, N: M + 1, P: L + N, R: K + P, T: L + 1, V: T + R
  // = 48 z^3 + 30 z^2 + 28 z + 2, times the 2 of n^2
  const Big zb = Big::of(BN_Z), e_big = (Big::of(48) * zb * zb * zb + Big::of(30) * zb * zb + Big::of(28) * zb + Big::of(2)) * Big::of(2);
  int s = -1;
  { int pw = n; const int nb = e_big.bits();
    for (int i = 0; i < nb; i++) { if (i) pw = b.mul(pw, pw); if (e_big.bit(i)) s = s < 0 ? pw : b.mul(s, pw); } }
  // outputs in tower order (c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 — each c0 then c1): V' MINUS vk.alpha_g1_beta_g2 * n^e — sixteen values that must all be
  // zero, then n, which must not
  std::vector<int> outs;
  {
    for (int i = 0; i < 2; i++) for (int j = 0; j < 3; j++) {
      const F2 &c = V[2 * j + i];
      const HFq2 *c2 = i == 0 ? &alpha_g1_beta_g2.c0.c0 + j : &alpha_g1_beta_g2.c1.c0 + j;
      outs.push_back(b.lin({{1, c.c0}, {-1, b.mul(s, b.constant(c2->c0))}})); outs.push_back(b.lin({{1, c.c1}, {-1, b.mul(s, b.constant(c2->c1))}})); } }
  for (int k = 0; k < N_CHECK; k++) outs.push_back(chk[k]);
  outs.push_back(n);

  // ---- liveness, levels, rounds, slots
  // -----------------------------------------------------------------------------------------------------------------------------------
  const size_t nn = b.nodes.size(); std::vector<char> live(nn, 0); for (int o : outs) live[o] = 1;
  for (size_t k = nn; k-- > 0;) {
    if (!live[k]) continue;
    const Builder::Node &nd = b.nodes[k];
    if (nd.kind == K_MUL) {
      live[nd.a] = live[nd.b] = 1;
    } else if (nd.kind == K_LIN8) for (auto &t : nd.terms) live[t.second] = 1;
  }
  for (int o : outs) if (b.nodes[o].kind != K_LIN8) throw std::runtime_error("verify schedule: output is not a computed sum");
  {
    std::vector<int> seen(outs);
    std::sort(seen.begin(), seen.end());
    if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) throw std::runtime_error("verify schedule: two outputs share a value");
  }
  int max_level = 0; for (size_t k = 0; k < nn; k++) if (live[k]) max_level = std::max(max_level, b.nodes[k].level);
  // Levels.  A value may sit anywhere between its earliest level (its operands' depth) and one level before its first reader.  As EARLY as possible, everything
  // off the critical chain — the products of the line values, the powers of n — runs hundreds of levels ahead of its readers and the live values do not fit the
  // LDS; as LATE as possible (the first generation), the accumulator's chain f <- f^2 L, which only needs four levels a bit, is compressed into the last 256
  // levels of the running point's 480 and every one of those levels then needs two rounds.  So: late, but only while the level still has room in its ONE round
  // (four waves) — the values on the critical path first (they have no choice), then the others from the outputs backwards, each at the latest level below its
  // readers that has a free lane for it, falling back towards its earliest level; if none has room, at the latest (that level then takes a second round).
  { std::vector<int> alap(nn, max_level + 1); for (int o : outs) alap[o] = max_level;
    for (size_t k = nn; k-- > 0;) {
      if (!live[k]) continue;
      const Builder::Node &nd = b.nodes[k];
      if (alap[k] > max_level) alap[k] = max_level;
      const int lv = alap[k];
      if (nd.kind == K_MUL) {
        alap[nd.a] = std::min(alap[nd.a], lv - 1);
        alap[nd.b] = std::min(alap[nd.b], lv - 1);
      } else if (nd.kind == K_LIN8) for (auto &t : nd.terms) alap[t.second] = std::min(alap[t.second], lv - 1);
    }
    struct Load { uint32_t mul = 0, wide = 0, narrow = 0; };
    std::vector<Load> load(max_level + 2);
    auto waves_of = [](const Load &l) { const uint32_t per8 = WAVE / LIN_GROUP, w8 = (l.wide + per8 - 1) / per8, spare = w8 * per8 - l.wide;
      return (l.mul + WAVE - 1) / WAVE + w8 + (l.narrow <= spare ? 0u : (l.narrow + WAVE - 1) / WAVE); };
    auto with = [&](Load l, const Builder::Node &nd) { if (nd.kind == K_MUL) l.mul++; else if (nd.terms.size() > TERMS_PER_LANE) l.wide++; else l.narrow++; return l; };
    auto is_op = [&](size_t k) { return live[k] && (b.nodes[k].kind == K_MUL || b.nodes[k].kind == K_LIN8); };
    std::vector<int> place(nn, -1), ub(nn, max_level + 1);
    auto put = [&](size_t k, int lv) { place[k] = lv; load[lv] = with(load[lv], b.nodes[k]); const Builder::Node &nd = b.nodes[k];
      if (nd.kind == K_MUL) { ub[nd.a] = std::min(ub[nd.a], lv - 1); ub[nd.b] = std::min(ub[nd.b], lv - 1); } else for (auto &t : nd.terms) ub[t.second] = std::min(ub[t.second], lv - 1); };
    for (size_t k = 0; k < nn; k++) if (is_op(k)) { if (alap[k] < b.nodes[k].level) throw std::runtime_error("verify schedule: level"); if (alap[k] == b.nodes[k].level) put(k, alap[k]); }
    for (int o : outs) ub[o] = std::min(ub[o], max_level);
    for (size_t k = nn; k-- > 0;) { if (!is_op(k) || place[k] >= 0) continue;
      const int hi = std::min(ub[k], max_level), lo = b.nodes[k].level; if (hi < lo) throw std::runtime_error("verify schedule: no level for a value");
      int lv = hi; for (int c = hi; c >= lo; c--) if (waves_of(with(load[c], b.nodes[k])) <= WAVES) { lv = c; break; }
      put(k, lv); }
    for (size_t k = 0; k < nn; k++) if (is_op(k)) {
      const Builder::Node &nd = b.nodes[k]; const int lv = place[k];
      auto before = [&](int o) { const uint8_t kd = b.nodes[o].kind; if ((kd == K_MUL || kd == K_LIN8) && place[o] >= lv) throw std::runtime_error("verify schedule: order of levels"); };
      if (nd.kind == K_MUL) { before(nd.a); before(nd.b); } else for (auto &t : nd.terms) before(t.second);
    }
    for (size_t k = 0; k < nn; k++) if (is_op(k)) b.nodes[k].level = place[k];
  }
  std::vector<std::vector<int>> mul_at(max_level + 1), lin_at(max_level + 1);
  for (size_t k = 0; k < nn; k++) if (live[k]) {
    if (b.nodes[k].kind == K_MUL) mul_at[b.nodes[k].level].push_back((int)k);
    else if (b.nodes[k].kind == K_LIN8) lin_at[b.nodes[k].level].push_back((int)k);
  }
  // A level's operations are cut into WAVES — 64 products, or 8 eight-lane sums, or 64 one-lane sums (up to three terms; they ride in the spare groups of the
  // level's eight-lane waves when they all fit there) — and four waves make a round, whatever their kinds.
  struct WaveOp { uint32_t kind; std::vector<int> ns; }; struct Round { std::vector<WaveOp> waves; }; std::vector<Round> rounds; std::vector<uint32_t> round_level;
  for (int lv = 1; lv <= max_level; lv++) {
    std::vector<WaveOp> ws;
    for (size_t i = 0; i < mul_at[lv].size(); i += WAVE) ws.push_back(WaveOp{K_MUL, std::vector<int>(mul_at[lv].begin() + i,
        mul_at[lv].begin() + std::min(mul_at[lv].size(), i + WAVE))});
    std::vector<int> wide, narrow; for (int k : lin_at[lv]) (b.nodes[k].terms.size() > TERMS_PER_LANE ? wide : narrow).push_back(k);
    const size_t per8 = WAVE / LIN_GROUP, spare = wide.empty() ? 0 : (per8 - wide.size() % per8) % per8;
    if (!wide.empty() && narrow.size() <= spare) { wide.insert(wide.end(), narrow.begin(), narrow.end()); narrow.clear(); }
    for (size_t i = 0; i < wide.size(); i += per8) ws.push_back(WaveOp{K_LIN8, std::vector<int>(wide.begin() + i, wide.begin() + std::min(wide.size(),
        i + per8))});
    for (size_t i = 0; i < narrow.size(); i += WAVE) ws.push_back(WaveOp{K_LIN1, std::vector<int>(narrow.begin() + i,
        narrow.begin() + std::min(narrow.size(), i + WAVE))});
    for (size_t i = 0; i < ws.size(); i += WAVES) { Round rd; rd.waves.assign(ws.begin() + i, ws.begin() + std::min(ws.size(), i + WAVES)); rounds.push_back(rd);
      round_level.push_back((uint32_t)lv); }
  }
  std::vector<int> last_use(nn, -1); std::vector<char> is_out(nn, 0); for (int o : outs) is_out[o] = 1;
  for (size_t r = 0; r < rounds.size(); r++) for (const WaveOp &wv : rounds[r].waves) for (int k : wv.ns) {
    const Builder::Node &nd = b.nodes[k];
    if (nd.kind == K_MUL) {
      last_use[nd.a] = last_use[nd.b] = (int)r;
    } else for (auto &t : nd.terms) last_use[t.second] = (int)r;
  }
  Schedule sc; sc.round_level = round_level; sc.n_levels = (uint32_t)max_level;
  sc.consts = b.consts;
  std::vector<int> slot(nn, -1);
  for (int i = 0; i < N_INPUTS; i++) slot[i] = i;
  std::vector<int> free_slots;
  uint32_t next_slot = N_INPUTS;
  if (sc.consts.size() >= CONST_FLAG) throw std::runtime_error("verify schedule: too many constants");
  auto ref = [&](int k) -> uint32_t { const Builder::Node &nd = b.nodes[k]; if (nd.kind == 3) return CONST_FLAG | nd.cidx;
      if (slot[k] < 0) throw std::runtime_error("verify schedule: value read before it was written"); return (uint32_t)slot[k]; };
  const uint32_t no_term = CONST_FLAG | b.nodes[b.zero_node].cidx;   // coefficient 0 on the constant zero
  auto term_word = [&](const std::pair<int, int> &t) -> uint32_t { const uint32_t c = (uint32_t)(t.first < 0 ? -t.first : t.first);
      if (!c || c > MAX_COEF) throw std::runtime_error("verify schedule: coefficient"); return ref(t.second) | (t.first < 0 ? 1u << 16 : 0u) | c << 17; };
  const size_t padded = (rounds.size() + PREFETCH_ROUNDS - 1) / PREFETCH_ROUNDS * PREFETCH_ROUNDS + PREFETCH_ROUNDS;
  sc.prog.assign(padded * LANES * WPL, 0u);
  for (size_t r = 0; r < rounds.size(); r++) {
    const Round &rd = rounds[r]; uint32_t *words = &sc.prog[r * LANES * WPL];
    // (operands are resolved BEFORE this round's destinations are assigned)
    for (size_t wv = 0; wv < rd.waves.size(); wv++) { const WaveOp &op = rd.waves[wv]; const uint32_t lanes_per = op.kind == K_LIN8 ? LIN_GROUP : 1;
      sc.waves_of_kind[op.kind]++;
      for (uint32_t l = 0; l < WAVE; l++) { uint32_t *w = words + (wv * WAVE + l) * WPL; w[0] = op.kind << KIND_SHIFT; w[1] = w[2] = w[3] = op.kind == K_MUL ? 0u : no_term; }
      for (size_t k = 0; k < op.ns.size(); k++) { const Builder::Node &nd = b.nodes[op.ns[k]];
        if (op.kind == K_MUL) { uint32_t *w = words + (wv * WAVE + k) * WPL; w[0] |= LIVE_BIT | STORE_BIT; w[1] = ref(nd.a); w[2] = ref(nd.b); w[3] = 0; sc.n_mul++; }
        else {
          for (uint32_t l = 0; l < lanes_per; l++) { uint32_t *w = words + (wv * WAVE + k * lanes_per + l) * WPL; w[0] |= LIVE_BIT | (l == lanes_per - 1 ? STORE_BIT : 0u);
            for (uint32_t t = 0; t < TERMS_PER_LANE; t++) { const size_t ti = (size_t)t * lanes_per + l; w[1 + t] = ti < nd.terms.size() ? term_word(nd.terms[ti]) : no_term; } }
          sc.n_lin++;
        }
      }
    }
    // destinations come from the slots freed in EARLIER rounds only (this round's operands are released below, after the assignment): no lane writes a slot
    // that another lane of the same round still reads, so the kernel needs one barrier per round, not two
    std::vector<int> dsts;
    for (size_t wv = 0; wv < rd.waves.size(); wv++) { const WaveOp &op = rd.waves[wv]; const uint32_t lanes_per = op.kind == K_LIN8 ? LIN_GROUP : 1;
      for (size_t k = 0; k < op.ns.size(); k++) {
        int s2;
        if (!free_slots.empty()) {
          s2 = free_slots.back();
          free_slots.pop_back();
        } else s2 = (int)next_slot++;
        if (s2 >= (int)CONST_FLAG) throw std::runtime_error("verify schedule: too many live values");
        slot[op.ns[k]] = s2; dsts.push_back(s2);
        for (uint32_t l = 0; l < lanes_per; l++) words[(wv * WAVE + k * lanes_per + l) * WPL] |= (uint32_t)s2;
      }
    }
    { std::vector<char> is_dst(next_slot, 0); for (int s2 : dsts) is_dst[s2] = 1;
      for (size_t q = 0; q < (size_t)LANES * WPL; q++) {
        const uint32_t pos = (uint32_t)(q % WPL), wv = words[q], kind = words[q - pos] >> KIND_SHIFT; if (!pos || !(words[q - pos] & LIVE_BIT)) continue;
        const bool operand = kind == K_MUL ? pos < 3 : (wv >> 17) != 0;
        if (operand && !(wv & CONST_FLAG) && is_dst[wv & 0x7fffu]) throw std::runtime_error("verify schedule: a round writes a slot it reads"); } }
    for (const WaveOp &op : rd.waves) for (int k : op.ns) {
      const Builder::Node &nd = b.nodes[k];
      auto release = [&](int o) {
        if (b.nodes[o].kind != 3 && o >= N_INPUTS && last_use[o] == (int)r && !is_out[o] && slot[o] >= 0) {
          free_slots.push_back(slot[o]);
          slot[o] = -2;
        }
      };
      if (nd.kind == K_MUL) { release(nd.a); release(nd.b); } else for (auto &t : nd.terms) release(t.second); }
  }
  sc.n_rounds = (uint32_t)rounds.size(); sc.n_rounds_padded = (uint32_t)padded; sc.n_slots = next_slot;
  for (size_t k = 0; k < outs.size(); k++) sc.out_slot[k] = (uint32_t)slot[outs[k]];
  return sc;
}

// the operations of a round in program order: fn(kind, w) for every live MUL lane / LIN group (w: the group's first lane's words; per: lanes of the group)
template <class Fn> inline void for_each_op(const Schedule &S, uint32_t r, Fn fn) {
  for (uint32_t wv = 0; wv < WAVES; wv++) { const uint32_t *base = &S.prog[((size_t)r * LANES + wv * WAVE) * WPL]; const uint32_t kind = base[0] >> KIND_SHIFT;
    if (kind == K_IDLE) continue; const uint32_t per = kind == K_LIN8 ? LIN_GROUP : 1;
    for (uint32_t g = 0; g < WAVE / per; g++) { const uint32_t *w = base + (size_t)g * per * WPL; if (!(w[0] & LIVE_BIT)) continue;
      if ((w[0] >> KIND_SHIFT) != kind) throw std::runtime_error("verify schedule: a wave of two kinds");
      fn(kind, per, w); } }
}

// Host interpreter of a schedule on the host field type: what the program means. in: the N_INPUTS values of one proof. Returns the output values (N_OUT): the
// first N_ZERO all zero and the last one not, for a valid proof.
inline std::vector<HFq> simulate(const Schedule &S, const HFq *in) {
  std::vector<HFq> slots(S.n_slots, HFq::zero()); for (int i = 0; i < N_INPUTS; i++) slots[i] = in[i];
  auto val = [&](uint32_t r) { return (r & CONST_FLAG) ? S.consts[r & (CONST_FLAG - 1)] : slots[r & 0x7fff]; };
  auto times = [](HFq x, uint32_t c) { HFq r = HFq::zero(); for (; c; c >>= 1) { if (c & 1) r = r + x; x = x + x; } return r; };
  std::vector<std::pair<uint32_t, HFq>> writes;
  for (uint32_t r = 0; r < S.n_rounds_padded; r++) {
    writes.clear();
    for_each_op(S, r, [&](uint32_t kind, uint32_t per, const uint32_t *w) { HFq v = HFq::zero();
      if (kind == K_MUL) v = val(w[1]) * val(w[2]);
      else for (uint32_t l = 0; l < per; l++) for (uint32_t t = 0; t < TERMS_PER_LANE; t++) {
        const uint32_t e = w[l * WPL + 1 + t];
        const HFq x = times(val(e & 0xffff), e >> 17);
        v = ((e >> 16) & 1) ? v - x : v + x;
      }
      writes.push_back({w[0] & 0x7fffu, v}); });
    for (auto &wv : writes) slots[wv.first] = wv.second;   // (all reads of a round happen before its writes, as in the lock-step workgroup)
  }
  std::vector<HFq> out; for (int k = 0; k < N_OUT; k++) out.push_back(slots[S.out_slot[k]]); return out;
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
// after_round (optional): called with the round's number and all values once its writes are done
inline std::vector<bool> simulate29(const Schedule &S, const uint32_t (*in_words)[8],
    const std::function<void(uint32_t, const std::vector<std::array<uint32_t, 9>> &)> &after_round = nullptr) {
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
    else memcpy(x, slots[r & 0x7fff].data(), 36);
  };
  std::vector<std::pair<uint32_t, V>> writes;
  for (uint32_t r = 0; r < S.n_rounds_padded; r++) {
    writes.clear();
    for_each_op(S, r, [&](uint32_t kind, uint32_t per, const uint32_t *w) { uint32_t res[9];
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
      check_stored(res); V v; memcpy(v.data(), res, 36); writes.push_back({w[0] & 0x7fffu, v}); });
    for (auto &wv : writes) slots[wv.first] = wv.second;
    if (after_round) after_round(r, slots);
  }
  std::vector<bool> out;
  for (int k = 0; k < N_OUT; k++) {
    uint32_t l[9];
    memcpy(l, slots[S.out_slot[k]].data(), 36);
    out.push_back(l29::multiple_of_p(l));
  }
  return out;
}

} }  // namespace zk::vsched
