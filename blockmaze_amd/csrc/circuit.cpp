// gadgetlib1 subset restated on circuit::Board — see circuit.hpp for the source map.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <stdexcept>
#include "circuit.hpp"
#include <chrono>

namespace zk { namespace circuit {

// ---- Board -------------------------------------------------------------------------------------------------------------
void Board::push(int m, const LC &lc) {
  // canonical row: sorted by variable, duplicates merged, zero coefficients dropped
  std::vector<Term> t = lc.t; std::stable_sort(t.begin(), t.end(), [](const Term &a, const Term &b) { return a.v < b.v; });
  for (size_t i = 0; i < t.size();) { HFr c = t[i].c; size_t j = i + 1; while (j < t.size() && t[j].v == t[i].v) { c = c + t[j].c; j++; }
    if (!c.is_zero()) { cs.col[m].push_back(t[i].v); HFr cc = c.from_mont(); Fe32 f; memcpy(&f, cc.l, 32); cs.coeff[m].push_back(f); } i = j; }
  cs.rowptr[m].push_back((uint32_t)cs.col[m].size());
}
void Board::constraint(const LC &a, const LC &b, const LC &c) { if (!emit) return; push(0, a); push(1, b); push(2, c); }

// the operands are bits: assemble the integer natively, one conversion to Montgomery form at the end
HFr pack_bits_value(const Board &b, const LCArray &bits) {
  if (bits.size() <= 253) {
    HFr c = HFr::zero();
    for (size_t i = 0; i < bits.size(); i++) if (b.eval_bit(bits[i])) c.l[i / 64] |= 1ull << (i % 64);
    return c.to_mont();
  }
  HFr r = HFr::zero(); for (size_t i = bits.size(); i-- > 0;) { r = r.dbl(); r = r + b.eval(bits[i]); } return r; }
LC packing_sum(const LCArray &bits) { LC r; HFr two_i = HFr::one(); for (const LC &x : bits) { r.add(x.scaled(two_i)); two_i = two_i.dbl(); } return r; }
void fill_bits_of_value(Board &b, const VarArray &bits, const HFr &value) {
  HFr c = value.from_mont();
  for (size_t i = 0; i < bits.size(); i++) b.set_bit(bits[i], i < 256 && ((c.l[i / 64] >> (i % 64)) & 1));
}
void boolean_constraint(Board &b, const LC &x) { b.constraint(x, LC::constant(HFr::one()) - x, LC()); }
void Packing::constraints(bool enforce_bitness) {
  b.constraint(LC::constant(HFr::one()), packing_sum(bits), packed);
  if (enforce_bitness) for (const LC &x : bits) boolean_constraint(b, x);
}

static const uint32_t SHA256_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74,
        0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3,
        0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819,
        0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa,
        0xa4506ceb, 0xbef9a3f7, 0xc67178f2
  };
static const uint32_t SHA256_H[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

LCArray sha256_default_iv() { LCArray r; r.reserve(256);
  for (int i = 0; i < 256; i++) { int bit = (SHA256_H[i / 32] >> (31 - (i % 32))) & 1; r.push_back(bit ? LC::constant(HFr::one()) : LC()); } return r; }

// ---- sha256_aux.tcc ------------------------------------------------------------------------------------------------------
namespace {
const LC ONE_LC = LC::constant(HFr::one());
const HFr TWO = HFr::from_u64(2);

// Where a gadget's bit array lives on the board: `first` if its variables are consecutive and ascending (alloc_array, or allocations made one after the other),
// so that the native witness path writes the bits of a word with Board::set_bits_run instead of one indexed byte store per bit
struct BitRun {
  Var first = 0; uint32_t count = 0; bool consecutive = false;
  BitRun() {}
  explicit BitRun(const VarArray &v) : first(v.empty() ? 0 : v[0]), count((uint32_t)v.size()), consecutive(!v.empty() && v[0] != 0) {
    for (size_t i = 0; i < v.size(); i++) consecutive = consecutive && v[i] == v[0] + i;
  }
  void write(Board &b, const VarArray &vars, uint64_t bits) const {
    if (consecutive) b.set_bits_run(first, bits, count);
    else for (size_t i = 0; i < vars.size(); i++) b.set_bit(vars[i], (bits >> i) & 1);
  }
};
struct LastBits {     // X has X_bits bits; result = the low |result_bits| of them
  Board &b; Var X; size_t X_bits; Var result; VarArray result_bits, full_bits, high_bits; BitRun low_run, high_run;
  LastBits(Board &b, Var X, size_t X_bits, Var result, const VarArray &result_bits) : b(b), X(X), X_bits(X_bits), result(result), result_bits(result_bits),
      full_bits(result_bits) {
    for (size_t i = result_bits.size(); i < X_bits; i++) { high_bits.push_back(b.alloc()); full_bits.push_back(high_bits.back()); }
    low_run = BitRun(result_bits); high_run = BitRun(high_bits); }
  void constraints() { Packing(b, to_lcs(full_bits), X).constraints(true); Packing(b, to_lcs(result_bits), result).constraints(false); }
  void witness() { fill_bits_of_value(b, full_bits, b.get(X)); b.set(result, pack_bits_value(b, to_lcs(result_bits))); }
  // native form: X is known as an integer (< 2^36)
  // (write_low = false: the result's bits are on the board already — Sha256Compression::witness_output_only — and other threads may be reading them: not touched again)
  uint32_t witness_native(uint64_t x, bool write_low = true) {
    b.set_small(X, x);
    if (write_low) low_run.write(b, result_bits, x);
    high_run.write(b, high_bits, x >> result_bits.size());
    const uint64_t r = result_bits.size() >= 64 ? x : x & ((1ull << result_bits.size()) - 1);
    b.set_small(result, r);
    return (uint32_t)r;
  }
};
inline uint32_t rotr32(uint32_t x, unsigned n) { return (x >> n) | (x << (32 - n)); }
struct Xor3 {
  Board &b; LC A, B, C; bool c_zero; Var out, tmp = 0;
  Xor3(Board &b, const LC &A, const LC &B, const LC &C, bool c_zero, Var out) : b(b), A(A), B(B), C(C), c_zero(c_zero), out(out) {
    if (!c_zero) tmp = b.alloc();
  }
  void constraints() {
    if (c_zero) b.constraint(A.scaled(TWO), B, A + B - LC(out));
    else { b.constraint(A.scaled(TWO), B, A + B - LC(tmp)); b.constraint(LC(tmp).scaled(TWO), C, LC(tmp) + C - LC(out)); } }
  void witness() { bool a = b.eval_bit(A), bb = b.eval_bit(B);
    if (c_zero) b.set_bit(out, a ^ bb); else { bool t = a ^ bb, c = b.eval_bit(C); b.set_bit(tmp, t); b.set_bit(out, t ^ c); } }
};
inline const LC &rotr(const LCArray &A, size_t i, size_t k) { return A[(i + k) % 32]; }
struct SmallSigma {
  Board &b; Var result; VarArray result_bits, tmp_bits; BitRun res_run, tmp_run; std::vector<Xor3> x; unsigned r1, r2, sh;
  uint32_t witness_native(uint32_t w) { uint32_t t = rotr32(w, r1) ^ rotr32(w, r2), res = t ^ (w >> sh);
    // (the 32 - sh low positions have a tmp variable: bits 0 .. 31 - sh of t)
    tmp_run.write(b, tmp_bits, t);
    res_run.write(b, result_bits, res);
    b.set_small(result, res);
    return res;
  }
  SmallSigma(Board &b, const LCArray &W, Var result, size_t rot1, size_t rot2, size_t shift) : b(b), result(result), result_bits(b.alloc_array(32)),
      r1((unsigned)rot1), r2((unsigned)rot2), sh((unsigned)shift) {
    for (size_t i = 0; i < 32; i++) {
      x.emplace_back(b, rotr(W, i, rot1), rotr(W, i, rot2), (i + shift < 32 ? W[i + shift] : ONE_LC), i + shift >= 32, result_bits[i]);
      if (!x.back().c_zero) {
        if (tmp_bits.size() != i) throw std::logic_error("SmallSigma: tmp variables are not the low positions");
        tmp_bits.push_back(x.back().tmp);
      }
    }
    res_run = BitRun(result_bits); tmp_run = BitRun(tmp_bits); }
  void constraints() { for (auto &g : x) g.constraints(); Packing(b, to_lcs(result_bits), result).constraints(false); }
  void witness() { for (auto &g : x) g.witness(); b.set(result, pack_bits_value(b, to_lcs(result_bits))); }
};
struct BigSigma {
  Board &b; Var result; VarArray result_bits, tmp_bits; BitRun res_run, tmp_run; std::vector<Xor3> x; unsigned q1, q2, q3;
  uint32_t witness_native(uint32_t w) { uint32_t t = rotr32(w, q1) ^ rotr32(w, q2), res = t ^ rotr32(w, q3);
    tmp_run.write(b, tmp_bits, t); res_run.write(b, result_bits, res); b.set_small(result, res); return res; }
  BigSigma(Board &b, const LCArray &W, Var result, size_t r1, size_t r2, size_t r3) : b(b), result(result), result_bits(b.alloc_array(32)), q1((unsigned)r1),
      q2((unsigned)r2), q3((unsigned)r3) {
    for (size_t i = 0; i < 32; i++) {
      x.emplace_back(b, rotr(W, i, r1), rotr(W, i, r2), rotr(W, i, r3), false, result_bits[i]);
      tmp_bits.push_back(x.back().tmp);
    }
    res_run = BitRun(result_bits); tmp_run = BitRun(tmp_bits); }
  void constraints() { for (auto &g : x) g.constraints(); Packing(b, to_lcs(result_bits), result).constraints(false); }
  void witness() { for (auto &g : x) g.witness(); b.set(result, pack_bits_value(b, to_lcs(result_bits))); }
};
struct Choice {
  Board &b; LCArray X, Y, Z; Var result; VarArray result_bits; BitRun res_run;
  Choice(Board &b, const LCArray &X, const LCArray &Y, const LCArray &Z, Var result) : b(b), X(X), Y(Y), Z(Z), result(result), result_bits(b.alloc_array(32)) {
    res_run = BitRun(result_bits);
  }
  void constraints() {
    for (size_t i = 0; i < 32; i++) b.constraint(X[i], Y[i] - Z[i], LC(result_bits[i]) - Z[i]);
    Packing(b, to_lcs(result_bits), result).constraints(false);
  }
  void witness() {
    for (size_t i = 0; i < 32; i++) {
      bool x = b.eval_bit(X[i]);
      b.set_bit(result_bits[i], x ? b.eval_bit(Y[i]) : b.eval_bit(Z[i]));
    }
    b.set(result, pack_bits_value(b, to_lcs(result_bits)));
  }
  uint32_t witness_native(uint32_t e, uint32_t f, uint32_t g) {
    uint32_t res = (e & f) ^ (~e & g);
    res_run.write(b, result_bits, res);
    b.set_small(result, res);
    return res;
  }
};
struct Majority {
  Board &b; LCArray X, Y, Z; Var result; VarArray result_bits; BitRun res_run;
  Majority(Board &b, const LCArray &X, const LCArray &Y, const LCArray &Z, Var result) : b(b), X(X), Y(Y), Z(Z), result(result),
      result_bits(b.alloc_array(32)) {
    res_run = BitRun(result_bits);
  }
  void constraints() {
    for (size_t i = 0; i < 32; i++) {
      boolean_constraint(b, LC(result_bits[i]));
      LC s = X[i] + Y[i] + Z[i] - LC(result_bits[i]).scaled(TWO);
      b.constraint(s, ONE_LC - s, LC());
    }
    Packing(b, to_lcs(result_bits), result).constraints(false); }
  void witness() {
    for (size_t i = 0; i < 32; i++) {
      int v = (int)b.eval_bit(X[i]) + (int)b.eval_bit(Y[i]) + (int)b.eval_bit(Z[i]);
      b.set_bit(result_bits[i], v >= 2);
    }
    b.set(result, pack_bits_value(b, to_lcs(result_bits)));
  }
  uint32_t witness_native(uint32_t x, uint32_t y, uint32_t z) {
    uint32_t res = (x & y) ^ (x & z) ^ (y & z);
    res_run.write(b, result_bits, res);
    b.set_small(result, res);
    return res;
  }
};

// ---- sha256_components.tcc ---------------------------------------------------------------------------------------------
struct MessageSchedule {
  Board &b; VarArray packed_W; std::vector<VarArray> W_bits; VarArray sigma0, sigma1, unreduced_W; std::vector<SmallSigma> cs0, cs1; std::vector<LastBits> red;
  MessageSchedule(Board &b, const VarArray &M, const VarArray &packed_W) : b(b), packed_W(packed_W), W_bits(64), sigma0(64), sigma1(64), unreduced_W(64) {
    for (size_t i = 0; i < 16; i++) { W_bits[i].resize(32); for (size_t k = 0; k < 32; k++) W_bits[i][k] = M[32 * i + 31 - k]; }
    cs0.reserve(48); cs1.reserve(48); red.reserve(48);
    for (size_t i = 16; i < 64; i++) { sigma0[i] = b.alloc(); sigma1[i] = b.alloc();
      cs0.emplace_back(b, to_lcs(W_bits[i - 15]), sigma0[i], 7, 18, 3); cs1.emplace_back(b, to_lcs(W_bits[i - 2]), sigma1[i], 17, 19, 10);
      unreduced_W[i] = b.alloc(); W_bits[i] = b.alloc_array(32); red.emplace_back(b, unreduced_W[i], 32 + 2, packed_W[i], W_bits[i]); } }
  void constraints() { for (size_t i = 0; i < 16; i++) Packing(b, to_lcs(W_bits[i]), packed_W[i]).constraints(false);
    for (size_t i = 16; i < 64; i++) { cs0[i - 16].constraints(); cs1[i - 16].constraints();
      b.constraint(ONE_LC, LC(sigma0[i]) + LC(sigma1[i]) + LC(packed_W[i - 16]) + LC(packed_W[i - 7]), LC(unreduced_W[i])); red[i - 16].constraints(); } }
  void witness_native(uint32_t W[64]) {   // W[0..15] given; the message bits themselves are inputs of the gadget and already assigned
    for (size_t i = 0; i < 16; i++) b.set_small(packed_W[i], W[i]);
    for (size_t i = 16; i < 64; i++) {
      uint32_t s0 = cs0[i - 16].witness_native(W[i - 15]), s1 = cs1[i - 16].witness_native(W[i - 2]);
      W[i] = red[i - 16].witness_native((uint64_t)s0 + s1 + W[i - 16] + W[i - 7]);
    }
  }
  void witness() { for (size_t i = 0; i < 16; i++) b.set(packed_W[i], pack_bits_value(b, to_lcs(W_bits[i])));
    for (size_t i = 16; i < 64; i++) {
      cs0[i - 16].witness();
      cs1[i - 16].witness();
      b.set(unreduced_W[i], b.get(sigma0[i]) + b.get(sigma1[i]) + b.get(packed_W[i - 16]) + b.get(packed_W[i - 7]));
      red[i - 16].witness();
    }
  }
};
struct RoundFunction {
  Board &b; LCArray a, bb, c, d, e, f, g, h; Var W; uint32_t K; VarArray new_a, new_e;
  Var sigma0, sigma1, choice, majority, packed_d, packed_h, unreduced_new_a, unreduced_new_e, packed_new_a, packed_new_e;
  std::unique_ptr<BigSigma> s0, s1; std::unique_ptr<Choice> ch; std::unique_ptr<Majority> mj; std::unique_ptr<LastBits> ra, re;
  RoundFunction(Board &b, const LCArray &a, const LCArray &bb, const LCArray &c, const LCArray &d, const LCArray &e, const LCArray &f, const LCArray &g,
      const LCArray &h,
                Var W, uint32_t K, const VarArray &new_a, const VarArray &new_e) : b(b), a(a), bb(bb), c(c), d(d), e(e), f(f), g(g), h(h), W(W), K(K),
                    new_a(new_a), new_e(new_e) {
    sigma0 = b.alloc(); sigma1 = b.alloc(); s0.reset(new BigSigma(b, a, sigma0, 2, 13, 22)); s1.reset(new BigSigma(b, e, sigma1, 6, 11, 25));
    choice = b.alloc(); ch.reset(new Choice(b, e, f, g, choice)); majority = b.alloc(); mj.reset(new Majority(b, a, bb, c, majority));
    packed_d = b.alloc(); packed_h = b.alloc(); unreduced_new_a = b.alloc(); unreduced_new_e = b.alloc(); packed_new_a = b.alloc(); packed_new_e = b.alloc();
    ra.reset(new LastBits(b, unreduced_new_a, 32 + 3, packed_new_a, new_a)); re.reset(new LastBits(b, unreduced_new_e, 32 + 3, packed_new_e, new_e)); }
  void constraints() { s0->constraints(); s1->constraints(); ch->constraints(); mj->constraints();
    b.constraint(ONE_LC, packing_sum(d), LC(packed_d)); b.constraint(ONE_LC, packing_sum(h), LC(packed_h));
    LC k = LC::constant_u64(K);
    b.constraint(ONE_LC, LC(packed_h) + LC(sigma1) + LC(choice) + k + LC(W) + LC(sigma0) + LC(majority), LC(unreduced_new_a));
    b.constraint(ONE_LC, LC(packed_d) + LC(packed_h) + LC(sigma1) + LC(choice) + k + LC(W), LC(unreduced_new_e));
    ra->constraints(); re->constraints(); }
  void witness_native(uint32_t wa, uint32_t wb, uint32_t wc, uint32_t wd, uint32_t we, uint32_t wf, uint32_t wg, uint32_t wh, uint32_t Wi, uint32_t &na,
      uint32_t &ne) {
    uint32_t S0 = s0->witness_native(wa), S1 = s1->witness_native(we), chv = ch->witness_native(we, wf, wg), mjv = mj->witness_native(wa, wb, wc);
    b.set_small(packed_d, wd); b.set_small(packed_h, wh);
    na = ra->witness_native((uint64_t)wh + S1 + chv + K + Wi + S0 + mjv); ne = re->witness_native((uint64_t)wd + wh + S1 + chv + K + Wi); }
  void witness() {
    s0->witness();
    s1->witness();
    ch->witness();
    mj->witness();
    b.set(packed_d, pack_bits_value(b, d));
    b.set(packed_h, pack_bits_value(b, h));
    HFr k = HFr::from_u64(K);
    b.set(unreduced_new_a, b.get(packed_h) + b.get(sigma1) + b.get(choice) + k + b.get(W) + b.get(sigma0) + b.get(majority));
    b.set(unreduced_new_e, b.get(packed_d) + b.get(packed_h) + b.get(sigma1) + b.get(choice) + k + b.get(W)); ra->witness(); re->witness(); }
};
}  // namespace

// ---- sha256_gadget.tcc:20-140 -----------------------------------------------------------------------------------------
struct Sha256Compression::Impl {
  Board &b;
  LCArray prev;
  VarArray block;
  VarArray packed_W;
  std::unique_ptr<MessageSchedule> ms;
  std::vector<std::unique_ptr<RoundFunction>> rounds;
  VarArray unreduced_output, reduced_output, output;
  std::vector<LastBits> reduce;
  Impl(Board &b, const LCArray &prev, const VarArray &block, const VarArray &output) : b(b), prev(prev), block(block), output(output) {
    packed_W = b.alloc_array(64); ms.reset(new MessageSchedule(b, block, packed_W));
    auto word = [&](int w) { LCArray r(32); for (int k = 0; k < 32; k++) r[k] = prev[32 * w + 31 - k]; return r; };   // word w, little-endian bits
    std::vector<LCArray> ra{word(0)}, rb{word(1)}, rc{word(2)}, rd{word(3)}, re{word(4)}, rf{word(5)}, rg{word(6)}, rh{word(7)};
    for (size_t i = 0; i < 64; i++) {
      rh.push_back(rg[i]); rg.push_back(rf[i]); rf.push_back(re[i]); rd.push_back(rc[i]); rc.push_back(rb[i]); rb.push_back(ra[i]);
      VarArray na = b.alloc_array(32); ra.push_back(to_lcs(na)); VarArray ne = b.alloc_array(32); re.push_back(to_lcs(ne));
      rounds.emplace_back(new RoundFunction(b, ra[i], rb[i], rc[i], rd[i], re[i], rf[i], rg[i], rh[i], packed_W[i], SHA256_K[i], na, ne));
    }
    unreduced_output = b.alloc_array(8); reduced_output = b.alloc_array(8); reduce.reserve(8);
    for (size_t i = 0; i < 8; i++) {
      VarArray ob(32);
      for (size_t k = 0; k < 32; k++) ob[k] = output[32 * i + 31 - k];
      reduce.emplace_back(b, unreduced_output[i], 32 + 1, reduced_output[i], ob);
    }
  }
};
Sha256Compression::Sha256Compression(Board &b, const LCArray &prev, const VarArray &block, const VarArray &output) : impl(new Impl(b, prev, block, output)) {}
void Sha256Compression::constraints() { Impl &s = *impl; s.ms->constraints(); for (auto &r : s.rounds) r->constraints();
  for (size_t i = 0; i < 4; i++) { s.b.constraint(ONE_LC, LC(s.rounds[3 - i]->packed_d) + LC(s.rounds[63 - i]->packed_new_a), LC(s.unreduced_output[i]));
                                   s.b.constraint(ONE_LC, LC(s.rounds[3 - i]->packed_h) + LC(s.rounds[63 - i]->packed_new_e), LC(s.unreduced_output[4 + i])); }
  for (auto &r : s.reduce) r.constraints(); }
// native SHA-256 arithmetic; every variable gets exactly the value the gadget-by-gadget evaluation (witness_reference) assigns
void Sha256Compression::witness(bool outputs_written) {
  Impl &s = *impl; Board &b = s.b; uint32_t W[64], st[8];
  for (int i = 0; i < 16; i++) { uint32_t w = 0; for (int k = 0; k < 32; k++) w |= (uint32_t)b.bit(s.block[32 * i + 31 - k]) << k; W[i] = w; }
  for (int i = 0; i < 8; i++) { uint32_t w = 0; for (int k = 0; k < 32; k++) w |= (uint32_t)b.eval_bit(s.prev[32 * i + 31 - k]) << k; st[i] = w; }
  s.ms->witness_native(W);
  uint32_t a = st[0], bb = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7], hist_d[64], hist_h[64], hist_na[64], hist_ne[64];
  for (int i = 0; i < 64; i++) {
    uint32_t na, ne;
    hist_d[i] = d;
    hist_h[i] = h;
    s.rounds[i]->witness_native(a, bb, c, d, e, f, g, h, W[i], na, ne);
    hist_na[i] = na;
    hist_ne[i] = ne;
    h = g;
    g = f;
    f = e;
    e = ne;
    d = c;
    c = bb;
    bb = a;
    a = na;
  }
  for (int i = 0; i < 4; i++) {
    s.reduce[i].witness_native((uint64_t)hist_d[3 - i] + hist_na[63 - i], !outputs_written);
    s.reduce[4 + i].witness_native((uint64_t)hist_h[3 - i] + hist_ne[63 - i], !outputs_written);
  }
  for (int i = 0; i < 8; i++) b.set(s.unreduced_output[i], b.get(s.reduce[i].X));
}
// Only the 256 output bits, from the block and the previous state as they stand on the board (FIPS 180-4 6.2.2 on plain words: a microsecond): what a LATER
// compression reads.  With the outputs of a chain written first, all its compressions can fill in their ~25,000 internal variables side by side; witness(true)
// then leaves the output bits alone.  (It must: the first version wrote them a second time — the same values — while other threads were reading them, and 1 proof in
// 1,000 came out with ONE bit of a later compression's input state read differently: the board of the failing proof against a sequential witness of the same
// statement, tools/soak_send.py.  No variable is written while another thread may read it.)
void Sha256Compression::witness_output_only() {
  Impl &s = *impl; Board &b = s.b; uint32_t W[64], st[8], v[8];
  for (int i = 0; i < 16; i++) { uint32_t w = 0; for (int k = 0; k < 32; k++) w |= (uint32_t)b.bit(s.block[32 * i + 31 - k]) << k; W[i] = w; }
  for (int i = 0; i < 8; i++) { uint32_t w = 0; for (int k = 0; k < 32; k++) w |= (uint32_t)b.eval_bit(s.prev[32 * i + 31 - k]) << k; st[i] = w; v[i] = w; }
  auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
  for (int i = 16; i < 64; i++) W[i] = W[i - 16] + (rotr(W[i - 15], 7) ^ rotr(W[i - 15], 18) ^ (W[i - 15] >> 3)) + W[i - 7] + (rotr(W[i - 2], 17) ^ rotr(W[i - 2], 19) ^ (W[i - 2] >> 10));
  for (int i = 0; i < 64; i++) {
    const uint32_t t1 = v[7] + (rotr(v[4], 6) ^ rotr(v[4], 11) ^ rotr(v[4], 25)) + ((v[4] & v[5]) ^ (~v[4] & v[6])) + SHA256_K[i] + W[i],
        t2 = (rotr(v[0], 2) ^ rotr(v[0], 13) ^ rotr(v[0], 22)) + ((v[0] & v[1]) ^ (v[0] & v[2]) ^ (v[1] & v[2]));
    v[7] = v[6]; v[6] = v[5]; v[5] = v[4]; v[4] = v[3] + t1; v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = t1 + t2;
  }
  for (int i = 0; i < 8; i++) { const uint32_t o = st[i] + v[i]; for (int k = 0; k < 32; k++) b.set_bit(s.output[32 * i + 31 - k], (o >> k) & 1); }
}
void Sha256Compression::witness_reference() { Impl &s = *impl; s.ms->witness(); for (auto &r : s.rounds) r->witness();
  for (size_t i = 0; i < 4; i++) { s.b.set(s.unreduced_output[i], s.b.get(s.rounds[3 - i]->packed_d) + s.b.get(s.rounds[63 - i]->packed_new_a));
                                   s.b.set(s.unreduced_output[4 + i], s.b.get(s.rounds[3 - i]->packed_h) + s.b.get(s.rounds[63 - i]->packed_new_e)); }
  for (auto &r : s.reduce) r.witness(); }


// ---- the helper pool of run_parallel ----------------------------------------------------------------------------------------
namespace {
struct Batch { std::mutex m; std::condition_variable cv; size_t remaining; std::exception_ptr err; };
struct Job { std::function<void()> fn; Batch *batch; };
class TaskPool {
 public:
  // (never destroyed: the threads may outlive static destruction of a host process that exits from another thread)
  static TaskPool &instance() {
    static TaskPool *p = new TaskPool;
    return *p;
  }
  void run(std::vector<std::function<void()>> &tasks) {
    if (tasks.size() <= 1 || threads_.empty()) { for (auto &t : tasks) t(); return; }
    Batch b; b.remaining = tasks.size();
    { std::lock_guard<std::mutex> lk(m_); for (auto &t : tasks) q_.push_back(Job{std::move(t), &b}); queued_.fetch_add((int)tasks.size(), std::memory_order_release); } cv_.notify_all();
    // the caller works as well (possibly on another caller's tasks)
    for (;;) {
      Job j;
      {
        std::lock_guard<std::mutex> lk(m_);
        if (q_.empty()) break;
        j = std::move(q_.front());
        q_.pop_front(); queued_.fetch_sub(1, std::memory_order_release);
      }
      execute(j);
    }
    std::unique_lock<std::mutex> lk(b.m); b.cv.wait(lk, [&] { return b.remaining == 0; }); if (b.err) std::rethrow_exception(b.err);
  }
 private:
  TaskPool() {
    const char *e = getenv("ZK_WITNESS_THREADS");
    long n = e ? atol(e) : 4;
    unsigned hw = std::thread::hardware_concurrency();
    if (hw && (long)hw / 2 < n + 1) n = (long)hw / 2 - 1;
    if (n < 0) n = 0;
    if (n > 15) n = 15;
    for (long i = 0; i < n; i++) threads_.emplace_back([this] { loop(); }); for (auto &t : threads_) t.detach(); }
  static void execute(Job &j) {
    std::exception_ptr e;
    try {
      j.fn();
    }
    catch (...) {
      e = std::current_exception();
    }
    std::lock_guard<std::mutex> lk(j.batch->m);
    if (e && !j.batch->err) j.batch->err = e;
    if (--j.batch->remaining == 0) j.batch->cv.notify_all();
  }
  // A helper that has been woken — by a job or by nudge(), which a proof call makes on its way in, a few dozen microseconds before it has tasks — polls for
  // SPIN_US (an atomic count of queued jobs: no lock while there is nothing) before it sleeps again: waking a sleeping thread costs 30-50 us, which a witness of
  // 0.1 ms cannot afford twice.
  void loop() {
    using clock = std::chrono::steady_clock; uint64_t seen = 0; clock::time_point awake_until = clock::now();
    for (;;) {
      Job j; bool have = false;
      if (queued_.load(std::memory_order_acquire) == 0 && clock::now() < awake_until) { __builtin_ia32_pause(); continue; }
      {
        std::unique_lock<std::mutex> lk(m_);
        if (q_.empty() && clock::now() >= awake_until) { cv_.wait(lk, [&] { return !q_.empty() || nudged_ != seen; }); seen = nudged_; awake_until = clock::now() + std::chrono::microseconds(SPIN_US); }
        if (!q_.empty()) { j = std::move(q_.front()); q_.pop_front(); queued_.fetch_sub(1, std::memory_order_release); have = true; }
      }
      if (have) { execute(j); awake_until = clock::now() + std::chrono::microseconds(SPIN_US); }
    }
  }
  static constexpr int SPIN_US = 150;
  uint64_t nudged_ = 0; std::atomic<int> queued_{0};
 public:
  void nudge() { { std::lock_guard<std::mutex> lk(m_); nudged_++; } cv_.notify_all(); }
 private:
  std::mutex m_; std::condition_variable cv_; std::deque<Job> q_; std::vector<std::thread> threads_;
};
}  // namespace
void run_parallel(std::vector<std::function<void()>> tasks) { TaskPool::instance().run(tasks); }
void wake_helpers() { TaskPool::instance().nudge(); }

}  }  // namespace zk::circuit
