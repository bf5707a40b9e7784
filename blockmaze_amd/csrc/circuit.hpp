// R1CS circuit construction and witness generation for BlockMaze's four circuits (host side, row W1 of SURVEY.md §8a).
//
// The reference builds its circuits with libsnark's gadgetlib1 (protoboard + gadget objects) every time a proof is
// made (libsnark-vnt/src/send/sendcgo.cpp:203-207).  Variable numbering is fixed by the order in which gadget
// constructors allocate, constraint numbering by the order of the generate_r1cs_constraints() calls; both are part of
// the proving key (the key stores the constraint system and one group element per variable), so a drop-in prover has
// to reproduce them exactly.  This file restates the needed subset of gadgetlib1 with the same three phases
// (construct = allocate, constraints(), witness()) on a much lighter board: variables are plain indices, linear
// combinations are short term lists, constraints go straight into CSR arrays, values are host field elements.
//
// Restated gadgets and their sources (all under libsnark-vnt/depends/libsnark/libsnark/gadgetlib1/):
//   Board                      protoboard.tcc:19-130, pb_variable.tcc:27-133
//   Packing / MultiPacking     gadgets/basic_gadgets.tcc:31-108
//   Disjunction                gadgets/basic_gadgets.tcc:197-261
//   Digest (bit array)         gadgets/hashes/hash_io.tcc:17-60
//   LastBits, Xor3, SmallSigma, BigSigma, Choice, Majority      gadgets/hashes/sha256/sha256_aux.tcc:20-291
//   MessageSchedule, RoundFunction                              gadgets/hashes/sha256/sha256_components.tcc:52-243
//   Sha256Compression                                           gadgets/hashes/sha256/sha256_gadget.tcc:20-140
//   MerkleRead                 gadgets/merkle_tree/merkle_tree_check_read_gadget.tcc, merkle_authentication_path_variable.tcc,
//                              gadgets/hashes/digest_selector_gadget.tcc
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <vector>
#include "gpu.hpp"
#include "hostmath.hpp"

namespace zk { namespace circuit {

using host::HFr;
typedef uint32_t Var;                 // 0 is the constant ONE
typedef std::vector<Var> VarArray;

struct Term { Var v; HFr c; bool one; };   // one: c == 1 (evaluation skips the multiply)
struct LC {
  std::vector<Term> t;
  LC() {}
  LC(Var v) { t.push_back({v, HFr::one(), true}); }
  static LC constant(const HFr &c) { LC r; r.t.push_back({0, c, c == HFr::one()}); return r; }
  static LC constant_u64(uint64_t c) { return constant(HFr::from_u64(c)); }
  LC &add(const LC &o) { t.insert(t.end(), o.t.begin(), o.t.end()); return *this; }
  LC &sub(const LC &o) { for (const Term &x : o.t) t.push_back({x.v, x.c.neg(), false}); return *this; }
  LC &add_term(Var v, const HFr &c) { t.push_back({v, c, c == HFr::one()}); return *this; }
  LC scaled(const HFr &k) const { LC r; for (const Term &x : t) { HFr c = x.c * k; r.t.push_back({x.v, c, c == HFr::one()}); } return r; }
  friend LC operator+(LC a, const LC &b) { a.add(b); return a; }
  friend LC operator-(LC a, const LC &b) { a.sub(b); return a; }
};
typedef std::vector<LC> LCArray;
inline LCArray to_lcs(const VarArray &v) { LCArray r; r.reserve(v.size()); for (Var x : v) r.emplace_back(x); return r; }

class Board {
 public:
  explicit Board(bool emit_constraints) : emit(emit_constraints) {
    tag.push_back(1);
    wide.push_back(HFr::one()); ever_wide.push_back(0);
    for (int m = 0; m < 3; m++) cs.rowptr[m].push_back(0);
  }
  bool emit;                                   // false: witness-only pass (allocation still happens, constraints are skipped)
  // The assignment, variable 0 = ONE. 97 % of a BlockMaze witness is 0 or 1 (bits of SHA-256 states), so a value is ONE BYTE — tag 0 / 1 — unless it is
  // something else (tag 2, value in wide[]): a SHA round writes ~900 bytes instead of 28 KB of field elements, and the prover takes tags + wide values as they
  // are (Prover::set_witness_tagged: no scan of a 7 MB vector to find the zeros and ones again).
  std::vector<uint8_t> tag; std::vector<HFr> wide;
  // Which variables have EVER held something else than 0 / 1 in this object: the calls that write such values sit at fixed places of the gadgets, so after one
  // assignment the set is complete — the prover's hand-over gathers exactly these values (a fixed list: no scan of the tags for them) and uploads the tag bytes
  // as they are (Prover::set_witness_board).  wide_marks counts the first-time marks (the hashers run on several threads: a lost increment only matters if no
  // increment at all were seen, and a mark and its increment are made by the same thread before the pool joins).
  std::vector<uint8_t> ever_wide; uint32_t wide_marks = 0;
  void mark_wide(Var v) { if (!__atomic_load_n(&ever_wide[v], __ATOMIC_RELAXED)) { __atomic_store_n(&ever_wide[v], (uint8_t)1, __ATOMIC_RELAXED); __atomic_fetch_add(&wide_marks, 1u, __ATOMIC_RELAXED); } }
  R1csHost cs;
  Var alloc() { tag.push_back(0); wide.push_back(HFr::zero()); ever_wide.push_back(0); return (Var)(tag.size() - 1); }
  VarArray alloc_array(size_t n) { VarArray a(n); for (size_t i = 0; i < n; i++) a[i] = alloc(); return a; }
  void set_input_sizes(size_t n) { cs.n_inputs = n; }
  size_t num_variables() const { return tag.size() - 1; }
  void constraint(const LC &a, const LC &b, const LC &c);
  // tag 6 (round 4): a SMALL integer (below 2^64) kept as it is in wide[v].l[0] — the packed words and sums of the SHA-256 gadgets, ~850 per compression: the
  // native witness path used to pay a Montgomery product for each of them on the calling thread; now whoever needs the field element converts (get / eval here,
  // rarely; the prover's hand-over, on its four scan threads). TAG_WIDE and TAG_SMALL both have bit 1 set ("neither 0 nor 1") and bit 0 clear.
  static constexpr uint8_t TAG_WIDE = 2, TAG_SMALL = 6;
  HFr get(Var v) const {
    const uint8_t t = tag[v];
    return t == TAG_WIDE ? wide[v] : t == TAG_SMALL ? HFr::from_u64(wide[v].l[0]) : t ? HFr::one() : HFr::zero();
  }
  void set(Var v, const HFr &x) { if (x.is_zero()) tag[v] = 0; else if (x == HFr::one()) tag[v] = 1; else { wide[v] = x; tag[v] = TAG_WIDE; mark_wide(v); } }
  void set_small(Var v, uint64_t x) { if (x < 2) tag[v] = (uint8_t)x; else { wide[v].l[0] = x; tag[v] = TAG_SMALL; mark_wide(v); } }
  // bit i of `bits` -> variable first + i, i < count <= 64: the variables of a gadget's bit array are consecutive, eight of them are one 8-byte store
  void set_bits_run(Var first, uint64_t bits, size_t count) {
    uint8_t *t = tag.data() + first; size_t i = 0;
    // byte j of the store = bit j of x
    for (; i + 8 <= count; i += 8) {
      const uint64_t x = (bits >> i) & 0xff, sel = (x * 0x0101010101010101ull) & 0x8040201008040201ull,
          spread = ((sel + 0x7f7f7f7f7f7f7f7full) >> 7) & 0x0101010101010101ull;
      memcpy(t + i, &spread, 8);
    }
    for (; i < count; i++) t[i] = (uint8_t)((bits >> i) & 1);
  }
  HFr eval(const LC &lc) const {
    if (lc.t.size() == 1 && lc.t[0].one) return get(lc.t[0].v);
    HFr s = HFr::zero();
    for (const Term &x : lc.t) {
      const uint8_t t = tag[x.v];
      if (t == 0) continue;
      s = s + (x.one ? get(x.v) : t == 1 ? x.c : x.c * get(x.v));
    }
    return s;
  }
  bool eval_bit(const LC &lc) const {
    if (lc.t.size() == 1 && lc.t[0].one) return tag[lc.t[0].v] != 0;
    if (lc.t.empty()) return false;
    return !eval(lc).is_zero();
  }
  void set_bit(Var v, bool b) { if (v) tag[v] = (uint8_t)b; }   // writes to ONE are dropped (see LessCmp)
  bool bit(Var v) const { return tag[v] != 0; }
  void finish() { cs.n_vars = num_variables(); cs.n_cons = cs.rowptr[0].size() - 1; }
 private:
  void push(int m, const LC &lc);
};

// value of bits[i] * 2^i, bits little-endian
HFr pack_bits_value(const Board &b, const LCArray &bits);
LC packing_sum(const LCArray &bits);
void fill_bits_of_value(Board &b, const VarArray &bits, const HFr &value);     // fill_with_bits_of_field_element
void boolean_constraint(Board &b, const LC &x);                                 // basic_gadgets.tcc:17-22

struct Packing {      // packed = sum bits[i] 2^i
  Board &b; LCArray bits; LC packed; bool packed_is_var; Var packed_var;
  Packing(Board &b, const LCArray &bits, Var packed) : b(b), bits(bits), packed(packed), packed_is_var(true), packed_var(packed) {}
  void constraints(bool enforce_bitness);
  void witness_from_bits() { b.set(packed_var, pack_bits_value(b, bits)); }
};

struct Digest { Board &b; VarArray bits; Digest(Board &b, size_t n) : b(b), bits(b.alloc_array(n)) {} void constraints() {
    for (Var v : bits) boolean_constraint(b, LC(v)); }
  void fill(const std::vector<bool> &v) { for (size_t i = 0; i < bits.size(); i++) b.set_bit(bits[i], v[i]); }
  std::vector<bool> get() const { std::vector<bool> r(bits.size()); for (size_t i = 0; i < bits.size(); i++) r[i] = b.bit(bits[i]); return r; } };

struct Sha256Compression {
  struct Impl; std::shared_ptr<Impl> impl;
  // prev_output: 256 LCs (MSB-first words), block: 512 variables, output: 256 variables
  Sha256Compression(Board &b, const LCArray &prev_output, const VarArray &block, const VarArray &output);
  void constraints(); void witness(bool outputs_written = false);   // outputs_written: witness_output_only() has run — the output bits are not written a second time
  void witness_output_only();  // the 256 output bits alone, natively from the inputs on the board (what a later compression of a chain reads)
  void witness_reference();   // gadget-by-gadget evaluation exactly as libsnark does it; kept as the cross-check of the native path
};
LCArray sha256_default_iv();          // sha256_components.tcc:38-56

// Witness generation is 9 (send) to 18 (deposit) SHA-256 compression gadgets of 24,792 variables each, most of them independent of one another: run_parallel
// hands the tasks of one wave to a small process-wide pool of helper threads (the caller works too) and returns when all of them are done. Tasks of one call
// must write disjoint variables and read nothing another task of the same call writes — the circuits below order their waves so that this holds and the result
// equals the sequential order's.
void run_parallel(std::vector<std::function<void()>> tasks);
void wake_helpers();   // a proof call on its way in: the pool's threads are awake and polling by the time its tasks arrive

}  }  // namespace zk::circuit
