// BlockMaze statement circuits — see blockmaze_circuits.hpp.  Allocation and constraint order follow the reference's
// gadget constructors / generate_r1cs_constraints() line by line (cited per class); the code itself is written against
// circuit::Board.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include "blockmaze_circuits.hpp"

namespace zk {
using namespace circuit;

const char *circuit_name(CircuitKind k) { static const char *n[] = {"mint", "send", "deposit", "redeem"}; return n[(int)k]; }

// ---- bit order helpers (src/send/circuit/utils.tcc:23-66, src/send/util.h:96-107) --------------------------------------
std::vector<bool> blob_bits(const uint8_t *b, size_t nbytes) {
  std::vector<bool> v(nbytes * 8);
  for (size_t i = 0; i < nbytes; i++) for (int j = 0; j < 8; j++) v[i * 8 + j] = (b[i] >> (7 - j)) & 1;
  return v;
}
std::vector<bool> u64_bits(uint64_t x) { uint8_t le[8]; for (int i = 0; i < 8; i++) le[i] = (uint8_t)(x >> (8 * i)); return blob_bits(le, 8); }
std::vector<Fe32> pack_public_bits(const std::vector<bool> &bits) {
  const size_t chunk = 253; size_t n = (bits.size() + chunk - 1) / chunk; std::vector<Fe32> r(n);
  for (size_t i = 0; i < n; i++) {
    memset(&r[i], 0, 32);
    for (size_t j = 0; j < chunk && i * chunk + j < bits.size(); j++) if (bits[i * chunk + j]) r[i].l[j / 32] |= 1u << (j % 32);
  }
  return r; }

void Circuit::export_assignment(std::vector<Fe32> &z) const {
  size_t n = board.num_variables();
  z.resize(n);
  for (size_t i = 0; i < n; i++) {
    HFr c = board.get(i + 1).from_mont();
    memcpy(&z[i], c.l, 32);
  }
}

namespace {
const LC ONE_LC = LC::constant(HFr::one());

// value from bits "by order" (pb_variable.tcc:119-133): big-endian bytes, MSB-first bits == the integer itself for a
// 64-bit value stored as little-endian bytes
HFr value_by_order(const Board &b, const VarArray &bits) { HFr r = HFr::zero(); size_t n = bits.size();
  for (size_t i = 0; i < n / 8; i++) for (size_t j = 0; j < 8; j++) { r = r.dbl(); r = r + b.get(bits[n - 1 - i * 8 - (7 - j)]); } return r; }
void fill(Board &b, const VarArray &vars, const std::vector<bool> &bits) { for (size_t i = 0; i < vars.size(); i++) b.set_bit(vars[i], bits[i]); }

struct MultiPacking {   // basic_gadgets.tcc:60-108, chunk = Fr capacity = 253 bits
  Board &b; std::vector<Packing> packers;
  MultiPacking(Board &b, const VarArray &bits, const VarArray &packed) : b(b) { const size_t chunk = 253;
    for (size_t i = 0; i < packed.size(); i++) {
      VarArray part(bits.begin() + i * chunk, bits.begin() + std::min((i + 1) * chunk, bits.size()));
      packers.emplace_back(b, to_lcs(part), packed[i]);
    }
  }
  void constraints(bool bitness) { for (auto &p : packers) p.constraints(bitness); }
  void witness_from_bits() { for (auto &p : packers) p.witness_from_bits(); }
};
struct Disjunction {    // basic_gadgets.tcc:197-261
  Board &b; VarArray inputs; Var output, inv;
  Disjunction(Board &b, const VarArray &inputs, Var output) : b(b), inputs(inputs), output(output), inv(b.alloc()) {}
  void constraints() { LC sum; for (Var v : inputs) sum.add(LC(v)); b.constraint(LC(inv), sum, LC(output)); b.constraint(ONE_LC - LC(output), sum, LC()); }
  void witness() {
    HFr sum = HFr::zero();
    for (Var v : inputs) sum = sum + b.get(v);
    if (sum.is_zero()) {
      b.set(inv, HFr::zero());
      b.set(output, HFr::zero());
    } else {
      b.set(inv, sum.inv());
      b.set(output, HFr::one());
    }
  }
};
// less_comparison_gadget (src/send/circuit/comparison.tcc:5-96): proves A <= B for 64-bit values.  alpha[64] is the
// constant ONE (alpha.emplace_back(0) appends variable index 0), so packed(alpha) = 2^64 + B - A forces B - A >= 0.
struct LessCmp {
  Board &b; VarArray alpha; Var alpha_packed, not_all_zeros; LC A, B; std::unique_ptr<Disjunction> dis;
  LessCmp(Board &b, const LC &A, const LC &B) : b(b), A(A), B(B) {
    alpha = b.alloc_array(64);
    alpha.push_back(0);
    alpha_packed = b.alloc();
    not_all_zeros = b.alloc();
    dis.reset(new Disjunction(b, VarArray(alpha.begin(), alpha.begin() + 64), not_all_zeros)); }
  void constraints() { boolean_constraint(b, LC(not_all_zeros)); Packing(b, to_lcs(alpha), alpha_packed).constraints(true);
    b.constraint(ONE_LC, LC::constant(HFr::from_u64(2).pow_u64(64)) + B - A, LC(alpha_packed));
    dis->constraints();
    b.constraint(ONE_LC, LC(not_all_zeros), LC(not_all_zeros));
  }
  void witness() {
    b.set(alpha_packed, HFr::from_u64(2).pow_u64(64) + b.eval(B) - b.eval(A));
    fill_bits_of_value(b, alpha, b.get(alpha_packed));
    dis->witness();
  }
};

// two-/one-block SHA-256 wrappers with hard-wired padding (src/send/circuit/commitment.tcc); `pad` is the bit pattern of
// the padding, realised with the constant ONE and the variable ZERO (utils.tcc:3-12)
VarArray pad_bits(Var ZERO, size_t msg_bits, size_t total_bits) { VarArray p(total_bits - msg_bits, ZERO); p[0] = 0 /* ONE */;
  for (int i = 0; i < 64; i++) if ((msg_bits >> i) & 1) p[p.size() - 1 - i] = 0; return p; }
VarArray concat(std::initializer_list<VarArray> parts) { VarArray r; for (auto &p : parts) r.insert(r.end(), p.begin(), p.end()); return r; }
struct ShaTwoBlock {   // CMTA / CMTS / PRF: intermediate digest allocated first, then hasher1, hasher2
  Board &b; Digest inter; std::unique_ptr<Sha256Compression> h1, h2;
  ShaTwoBlock(Board &b, const VarArray &message_and_padding /* 1024 bits */, const VarArray &out) : b(b), inter(b, 256) {
    VarArray b1(message_and_padding.begin(), message_and_padding.begin() + 512), b2(message_and_padding.begin() + 512, message_and_padding.end());
    h1.reset(new Sha256Compression(b, sha256_default_iv(), b1, inter.bits)); h2.reset(new Sha256Compression(b, to_lcs(inter.bits), b2, out)); }
  void constraints() { inter.constraints(); h1->constraints(); h2->constraints(); }
  void witness() { h1->witness(); h2->witness(); }
  // (round 6) the outputs first — the intermediate digest, then the result, natively — so that both compressions, and whoever reads the result, can run side by side
  void outputs_first() { h1->witness_output_only(); h2->witness_output_only(); }
  void add_tasks(std::vector<std::function<void()>> &t) { Sha256Compression *a = h1.get(), *c = h2.get(); t.push_back([a] { a->witness(true); }); t.push_back([c] { c->witness(true); }); }
};
struct ShaOneBlock {   // CRH
  std::unique_ptr<Sha256Compression> h1;
  ShaOneBlock(Board &b, const VarArray &block, const VarArray &out) { h1.reset(new Sha256Compression(b, sha256_default_iv(), block, out)); }
  void constraints() { h1->constraints(); }
  void witness() { h1->witness(); }
};
std::unique_ptr<ShaTwoBlock> make_cmta(Board &b, Var ZERO, const VarArray &v, const VarArray &sn, const VarArray &r, const VarArray &out) {
  return std::unique_ptr<ShaTwoBlock>(new ShaTwoBlock(b, concat({v, sn, r, pad_bits(ZERO, 576, 1024)}), out));
}
std::unique_ptr<ShaTwoBlock> make_cmts(Board &b, Var ZERO, const VarArray &v, const VarArray &pk, const VarArray &r, const VarArray &sn_old,
    const VarArray &out) {
  return std::unique_ptr<ShaTwoBlock>(new ShaTwoBlock(b, concat({v, pk, r, sn_old, pad_bits(ZERO, 736, 1024)}), out));
}
std::unique_ptr<ShaTwoBlock> make_prf(Board &b, Var ZERO, const VarArray &sk, const VarArray &r, const VarArray &out) {
  return std::unique_ptr<ShaTwoBlock>(new ShaTwoBlock(b, concat({sk, r, pad_bits(ZERO, 512, 1024)}), out));
}
std::unique_ptr<ShaOneBlock> make_crh(Board &b, Var ZERO, const VarArray &pk, const VarArray &r, const VarArray &out) {
  return std::unique_ptr<ShaOneBlock>(new ShaOneBlock(b, concat({pk, r, pad_bits(ZERO, 416, 512)}), out));
}

void bool64(Board &b, const VarArray &v) { for (Var x : v) boolean_constraint(b, LC(x)); }

// ======================================================================================================================
// send (src/send/circuit/gadget.tcc:27-326, note.tcc, less_cmp.tcc)
// ======================================================================================================================
struct SendCircuit : Circuit {
  VarArray packed_inputs, unpacked;
  std::unique_ptr<Digest> cmtA_old, sn_old, cmtS, cmtA, r_old, pk_recv, pk_sender, r_s, sn, r, sk;
  std::unique_ptr<MultiPacking> unpacker;
  Var ZERO; VarArray value_old, value_s, value;
  Var l_value_old_packed, l_value_s_packed; std::unique_ptr<LessCmp> less;          // note_gadget_with_comparison_for_value_old
  Var s_value_old_packed, s_value_s_packed, s_value_packed;                           // note_gadget_with_packing_and_SUB
  std::unique_ptr<ShaOneBlock> crh; std::unique_ptr<ShaTwoBlock> prf, cmt_old, cmt_s, cmt_new;
  explicit SendCircuit(bool emit) : Circuit(emit) {
    Board &b = board;
    packed_inputs = b.alloc_array(5); b.set_input_sizes(5);                                                       // gadget.tcc:87-88
    auto alloc256 = [&](std::unique_ptr<Digest> &d) { d.reset(new Digest(b, 256)); unpacked.insert(unpacked.end(), d->bits.begin(), d->bits.end()); };
    alloc256(cmtA_old); alloc256(sn_old); alloc256(cmtS); alloc256(cmtA);                                           // :90-93
    unpacker.reset(new MultiPacking(b, unpacked, packed_inputs));
    ZERO = b.alloc();                                                                                               // :108
    value_old = b.alloc_array(64); r_old.reset(new Digest(b, 256));                                                 // :110-111
    value_s = b.alloc_array(64); pk_recv.reset(new Digest(b, 160)); pk_sender.reset(new Digest(b, 160)); r_s.reset(new Digest(b, 256));   // :114-117
    value = b.alloc_array(64); sn.reset(new Digest(b, 256)); r.reset(new Digest(b, 256)); sk.reset(new Digest(b, 256));                   // :119-123
    // note.tcc:35-37, less_cmp.tcc:23-24
    l_value_old_packed = b.alloc();
    l_value_s_packed = b.alloc();
    less.reset(new LessCmp(b, LC(l_value_s_packed), LC(l_value_old_packed)));
    // note.tcc:35-37,116
    s_value_old_packed = b.alloc();
    s_value_s_packed = b.alloc();
    s_value_packed = b.alloc();
    crh = make_crh(b, ZERO, pk_sender->bits, r->bits, r_s->bits);                                                    // gadget.tcc:153-159
    prf = make_prf(b, ZERO, sk->bits, r->bits, sn->bits);
    cmt_old = make_cmta(b, ZERO, value_old, sn_old->bits, r_old->bits, cmtA_old->bits);
    cmt_s = make_cmts(b, ZERO, value_s, pk_recv->bits, r_s->bits, sn_old->bits, cmtS->bits);
    cmt_new = make_cmta(b, ZERO, value, sn->bits, r->bits, cmtA->bits);
    if (emit) emit_constraints();
    b.finish();
  }
  // note.tcc:40-62
  void note_packing_constraints() {
    Board &b = board;
    bool64(b, value_old);
    bool64(b, value_s);
    sn_old->constraints();
    r_old->constraints();
    pk_recv->constraints();
    r_s->constraints();
  }
  void emit_constraints() { Board &b = board;                                                                         // gadget.tcc:196-225
    unpacker->constraints(true);
    note_packing_constraints(); less->constraints();                                                                  // lessCMP
    // noteSUB (note.tcc:119-140)
    note_packing_constraints();
    bool64(b, value);
    sn->constraints();
    r->constraints();
    sk->constraints();
    pk_sender->constraints();
    b.constraint(ONE_LC, LC(s_value_old_packed) - LC(s_value_s_packed), LC(s_value_packed));
    b.constraint(ONE_LC, LC(ZERO), LC());                                                                             // ZERO == 0
    r_s->constraints(); crh->constraints(); sn->constraints(); prf->constraints(); sn_old->constraints();
    cmtA_old->constraints(); cmt_old->constraints(); cmtS->constraints(); cmt_s->constraints(); cmtA->constraints(); cmt_new->constraints(); }
  void assign(const SendInputs &in) { Board &b = board; circuit::wake_helpers();                                    // gadget.tcc:228-271
    static const bool tr = getenv("ZK_TRACE_WITNESS") != nullptr;
    auto now = [] {
      return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    double t0 = now(), t1 = 0, t2 = 0, t3 = 0;
    auto note_fill = [&](Var vo_packed, Var vs_packed) { fill(b, value_old, u64_bits(in.value_old)); b.set(vo_packed, value_by_order(b, value_old));
      sn_old->fill(blob_bits(in.sn_old.b, 32));
      r_old->fill(blob_bits(in.r_old.b, 32));
      fill(b, value_s, u64_bits(in.value_s));
      b.set(vs_packed, value_by_order(b, value_s));
      pk_recv->fill(blob_bits(in.pk_recv.b, 20)); r_s->fill(blob_bits(in.r_s.b, 32)); };
    note_fill(l_value_old_packed, l_value_s_packed); less->witness();
    note_fill(s_value_old_packed, s_value_s_packed); fill(b, value, u64_bits(in.value)); b.set(s_value_packed, value_by_order(b, value));
    sn->fill(blob_bits(in.sn.b, 32)); r->fill(blob_bits(in.r.b, 32)); sk->fill(blob_bits(in.sk.b, 32)); pk_sender->fill(blob_bits(in.pk_sender.b, 20));
    b.set(ZERO, HFr::zero()); t1 = now();
    // sequential order of the reference: crh (writes r_s), prf (writes sn), cmt_old, cmt_s (reads r_s), cmt_new (reads sn). Two waves of independent hashers
    // give the same board:
    // (round 6) ONE wave of nine compressions: the outputs that a later compression reads — r_s, sn, and every two-block hasher's intermediate digest — are
    // written first (natively: a microsecond each, in the reference's order), then every compression fills in its own 25,000 variables beside the others and
    // leaves the outputs alone (Sha256Compression::witness(true)).
    crh->h1->witness_output_only(); prf->outputs_first(); cmt_old->outputs_first(); cmt_s->outputs_first(); cmt_new->outputs_first();
    { std::vector<std::function<void()>> t; cmt_s->add_tasks(t); cmt_new->add_tasks(t); prf->add_tasks(t); cmt_old->add_tasks(t); Sha256Compression *c0 = crh->h1.get(); t.push_back([c0] { c0->witness(true); });
      run_parallel(std::move(t)); }
    t2 = now();
    cmtA_old->fill(blob_bits(in.cmtA_old.b, 32)); cmtS->fill(blob_bits(in.cmtS.b, 32)); cmtA->fill(blob_bits(in.cmtA.b, 32));
    unpacker->witness_from_bits(); t3 = now(); if (tr) fprintf(stderr, "trace-witness: fills %.3f hashers %.3f rest %.3f ms\n", t1 - t0, t2 - t1, t3 - t2); }
};

// ======================================================================================================================
// mint (src/mint/circuit/gadget.tcc, note.tcc, add_cmp.tcc) and redeem (src/redeem/circuit/gadget.tcc, note.tcc, sub_cmp.tcc)
// ======================================================================================================================
struct MintRedeemCircuit : Circuit {
  bool redeem; VarArray packed_inputs, unpacked; std::unique_ptr<Digest> cmtA_old, sn_old, cmtA, sk, r, r_old, sn; std::unique_ptr<MultiPacking> unpacker;
  Var ZERO;
  VarArray value, value_old, value_s;
  Var value_packed, value_old_packed, value_s_packed;
  std::unique_ptr<LessCmp> less;
  std::unique_ptr<ShaTwoBlock> prf, cmt_old, cmt_new;
  MintRedeemCircuit(bool emit, bool redeem) : Circuit(emit), redeem(redeem) { Board &b = board;
    packed_inputs = b.alloc_array(4); b.set_input_sizes(4);
    auto alloc256 = [&](std::unique_ptr<Digest> &d) { d.reset(new Digest(b, 256)); unpacked.insert(unpacked.end(), d->bits.begin(), d->bits.end()); };
    // mint/gadget.tcc:84-88
    alloc256(cmtA_old);
    alloc256(sn_old);
    alloc256(cmtA);
    value_s = b.alloc_array(64);
    unpacked.insert(unpacked.end(), value_s.begin(), value_s.end());
    unpacker.reset(new MultiPacking(b, unpacked, packed_inputs));
    ZERO = b.alloc();
    value = b.alloc_array(64);
    value_old = b.alloc_array(64);
    sk.reset(new Digest(b, 256));
    r.reset(new Digest(b, 256));
    r_old.reset(new Digest(b, 256));
    sn.reset(new Digest(b, 256));
    value_packed = b.alloc(); value_old_packed = b.alloc(); value_s_packed = b.alloc();                                 // note.tcc:39-43
    if (redeem) less.reset(new LessCmp(b, LC(value_s_packed), LC(value_old_packed)));                                   // sub_cmp.tcc:25-27
    prf = make_prf(b, ZERO, sk->bits, r->bits, sn->bits);
    cmt_old = make_cmta(b, ZERO, value_old, sn_old->bits, r_old->bits, cmtA_old->bits);
    cmt_new = make_cmta(b, ZERO, value, sn->bits, r->bits, cmtA->bits);
    if (emit) emit_constraints();
    b.finish(); }
  void emit_constraints() { Board &b = board; unpacker->constraints(true);
    bool64(b, value_old); bool64(b, value_s); bool64(b, value); sk->constraints(); r->constraints(); r_old->constraints();
    // redeem/note.tcc:50-79, sub_cmp.tcc:30-37
    if (redeem) {
      sn->constraints();
      sn_old->constraints();
      b.constraint(ONE_LC, LC(value_old_packed) - LC(value_s_packed), LC(value_packed));
      less->constraints();
    }
    // add_cmp.tcc:23-29
    else b.constraint(ONE_LC, LC(value_old_packed) + LC(value_s_packed), LC(value_packed));
    b.constraint(ONE_LC, LC(ZERO), LC());
    sn->constraints();
    prf->constraints();
    sn_old->constraints();
    cmtA_old->constraints();
    cmt_old->constraints();
    cmtA->constraints();
    cmt_new->constraints();
  }
  template <class In> void assign(const In &in) { Board &b = board; circuit::wake_helpers();
    fill(b, value, u64_bits(in.value));
    b.set(value_packed, value_by_order(b, value));
    fill(b, value_old, u64_bits(in.value_old));
    b.set(value_old_packed, value_by_order(b, value_old));
    fill(b, value_s, u64_bits(in.value_s)); b.set(value_s_packed, value_by_order(b, value_s));
    sk->fill(blob_bits(in.sk.b, 32)); r->fill(blob_bits(in.r.b, 32)); r_old->fill(blob_bits(in.r_old.b, 32));
    if (redeem) { sn->fill(blob_bits(in.sn.b, 32)); sn_old->fill(blob_bits(in.sn_old.b, 32)); less->witness(); }
    b.set(ZERO, HFr::zero());
    // (no hasher writes sn_old: filling it before the first wave changes nothing)
    if (!redeem) sn_old->fill(blob_bits(in.sn_old.b, 32));
    // prf writes sn, cmt_old reads sn_old / r_old / value_old; mint/gadget.tcc:213-221: mint overwrites the computed serial number with the given one before cmt_new
    // reads it.  (round 6) One wave of six compressions: every output that a later step reads is written first, natively, in the reference's order.
    prf->outputs_first(); cmt_old->outputs_first();
    if (!redeem) sn->fill(blob_bits(in.sn.b, 32));
    cmt_new->outputs_first();
    { std::vector<std::function<void()>> t; cmt_new->add_tasks(t); prf->add_tasks(t); cmt_old->add_tasks(t); run_parallel(std::move(t)); }
    if (!redeem) sn->fill(blob_bits(in.sn.b, 32));                   // (prf's second compression wrote its own result again: the given one stands, as in the reference)
    cmtA_old->fill(blob_bits(in.cmtA_old.b, 32)); cmtA->fill(blob_bits(in.cmtA.b, 32)); unpacker->witness_from_bits(); }
};


// ======================================================================================================================
// deposit (src/deposit/circuit/gadget.tcc:24-368, note.tcc, merkle.tcc) over libsnark's Merkle gadgets
// (gadgets/merkle_tree/merkle_tree_check_read_gadget.tcc:20-112, merkle_authentication_path_variable.tcc:17-50,
//  gadgets/hashes/digest_selector_gadget.tcc:17-64, gadgets/basic_gadgets.tcc:110-194 bit_vector_copy / field_vector_copy)
// ======================================================================================================================
struct MerkleRead {
  Board &b;
  size_t depth;
  VarArray positions;
  std::vector<Digest> left, right, internal;
  std::unique_ptr<Digest> computed_root;
  std::vector<std::unique_ptr<Sha256Compression>> hashers;
  VarArray leaf, root; Var enforce; VarArray packed_source, packed_target; std::unique_ptr<MultiPacking> pack_source, pack_target; bool own_positions;
  // given_positions empty: BlockMaze's merkle_tree_gadget (allocates and boolean-constrains the position bits itself, merkle.tcc:20,41-50)
  MerkleRead(Board &b, size_t depth, const VarArray &leaf, const VarArray &root, Var enforce, const VarArray &given_positions = VarArray()) : b(b),
      depth(depth), leaf(leaf), root(root), enforce(enforce), own_positions(given_positions.empty()) {
    positions = own_positions ? b.alloc_array(depth) : given_positions;
    for (size_t i = 0; i < depth; i++) { left.emplace_back(b, 256); right.emplace_back(b, 256); }              // merkle_authentication_path_variable.tcc:22-26
    for (size_t i = 0; i + 1 < depth; i++) internal.emplace_back(b, 256);                                    // merkle_tree_check_read_gadget.tcc:49-52
    computed_root.reset(new Digest(b, 256));
    for (size_t i = 0; i < depth; i++) hashers.emplace_back(new Sha256Compression(b, sha256_default_iv(), concat({left[i].bits, right[i].bits}),
        i == 0 ? computed_root->bits : internal[i - 1].bits));
    // bit_vector_copy_gadget, chunk 253
    packed_source = b.alloc_array(2);
    pack_source.reset(new MultiPacking(b, computed_root->bits, packed_source));
    packed_target = b.alloc_array(2);
    pack_target.reset(new MultiPacking(b, root, packed_target));
  }
  const VarArray &input_of(size_t i) const { return i + 1 < depth ? internal[i].bits : leaf; }
  void constraints() {
    if (own_positions) for (Var p : positions) boolean_constraint(b, LC(p));                                // merkle.tcc:41-50
    for (size_t i = 0; i < depth; i++) { left[i].constraints(); right[i].constraints(); }
    for (auto &h : hashers) h->constraints();
    // digest_selector: is_right * (right - left) = input - left
    for (size_t i = 0; i < depth; i++) {
      Var is_right = positions[depth - 1 - i];
      const VarArray &in = input_of(i);
      for (size_t k = 0; k < 256; k++) b.constraint(LC(is_right), LC(right[i].bits[k]) - LC(left[i].bits[k]), LC(in[k]) - LC(left[i].bits[k])); }
    pack_source->constraints(false); pack_target->constraints(false);
    for (size_t k = 0; k < 2; k++) b.constraint(LC(enforce), LC(packed_source[k]) - LC(packed_target[k]), LC());       // field_vector_copy_gadget
  }
  // path: siblings from the leaf level upwards; index_bits[d] = bit d of the leaf position
  void witness(const std::vector<Blob256> &path, const std::vector<bool> &index_bits, std::vector<std::function<void()>> *tasks = nullptr) {
    for (size_t d = 0; d < depth; d++) b.set_bit(positions[d], index_bits[d]);                               // fill_with_bits_of_ulong(path_index)
    // authvars (address bit depth-1-i)
    for (size_t i = 0; i < depth; i++) {
      size_t level = depth - 1 - i;
      std::vector<bool> sib = blob_bits(path[level].b, 32);
      if (index_bits[level]) left[i].fill(sib);
      else right[i].fill(sib);
    }
    // (tasks: the compressions are handed back instead of being run here — every level's digest is written first, natively, up the whole path, so that the
    // levels can fill in their internals side by side with each other and with the caller's other hashers; finish() after they have run)
    for (size_t i = depth; i-- > 0;) {
      bool is_right = b.bit(positions[depth - 1 - i]);
      const VarArray &in = input_of(i);
      Digest &dst = is_right ? right[i] : left[i];
      for (size_t k = 0; k < 256; k++) b.set_bit(dst.bits[k], b.bit(in[k]));
      if (tasks) { hashers[i]->witness_output_only(); Sha256Compression *h = hashers[i].get(); tasks->push_back([h] { h->witness(true); }); } else hashers[i]->witness(); }
    if (!tasks) finish();
  }
  void finish() {
    if (b.bit(enforce)) for (size_t k = 0; k < 256; k++) b.set_bit(root[k], b.bit(computed_root->bits[k]));
    pack_source->witness_from_bits(); pack_target->witness_from_bits();
  }
};
struct DepositCircuit : Circuit {
  size_t depth;
  VarArray packed_inputs, unpacked;
  std::unique_ptr<Digest> rt, pk_recv, cmtB_old, sn_old, cmtB, sn_s, r_s, sn_A_old, cmtS, r_old, sn, r, sk;
  std::unique_ptr<MultiPacking> unpacker;
  Var value_enforce, ZERO; VarArray value_s, value_old, value; Var value_s_packed, value_old_packed, value_packed;
  std::unique_ptr<ShaTwoBlock> prf_sn, prf_sn_s, cmt_s, cmt_old, cmt_new; std::unique_ptr<MerkleRead> merkle;
  DepositCircuit(bool emit, size_t depth) : Circuit(emit), depth(depth) { Board &b = board;
    packed_inputs = b.alloc_array(6); b.set_input_sizes(6);
    auto alloc_in = [&](std::unique_ptr<Digest> &d, size_t n) { d.reset(new Digest(b, n)); unpacked.insert(unpacked.end(), d->bits.begin(), d->bits.end()); };
    alloc_in(rt, 256); alloc_in(pk_recv, 160); alloc_in(cmtB_old, 256); alloc_in(sn_old, 256); alloc_in(cmtB, 256); alloc_in(sn_s, 256);   // gadget.tcc:91-96
    unpacker.reset(new MultiPacking(b, unpacked, packed_inputs));
    value_enforce = b.alloc();
    ZERO = b.alloc();
    value_s = b.alloc_array(64);
    r_s.reset(new Digest(b, 256));
    sn_A_old.reset(new Digest(b, 256));
    cmtS.reset(new Digest(b, 256));
    value_old = b.alloc_array(64);
    r_old.reset(new Digest(b, 256));
    value = b.alloc_array(64);
    sn.reset(new Digest(b, 256));
    r.reset(new Digest(b, 256));
    sk.reset(new Digest(b, 256));
    value_s_packed = b.alloc(); value_old_packed = b.alloc(); value_packed = b.alloc();                      // note.tcc (note_gadget_with_packing_and_ADD ctor)
    prf_sn = make_prf(b, ZERO, sk->bits, r->bits, sn->bits); prf_sn_s = make_prf(b, ZERO, sk->bits, r_s->bits, sn_s->bits);
    cmt_s = make_cmts(b, ZERO, value_s, pk_recv->bits, r_s->bits, sn_A_old->bits, cmtS->bits);
    cmt_old = make_cmta(b, ZERO, value_old, sn_old->bits, r_old->bits, cmtB_old->bits);
    cmt_new = make_cmta(b, ZERO, value, sn->bits, r->bits, cmtB->bits);
    merkle.reset(new MerkleRead(b, depth, cmtS->bits, rt->bits, value_enforce));
    if (emit) emit_constraints();
    b.finish(); }
  void emit_constraints() { Board &b = board;                                                               // gadget.tcc:196-233
    unpacker->constraints(true);
    bool64(b, value_s); bool64(b, value_old); bool64(b, value); b.constraint(ONE_LC, LC(value_old_packed) + LC(value_s_packed), LC(value_packed));
    pk_recv->constraints();
    r_s->constraints();
    sn_A_old->constraints();
    sn_old->constraints();
    r_old->constraints();
    sn->constraints();
    r->constraints();
    sk->constraints();
    b.constraint(ONE_LC, LC(ZERO), LC());
    sn_s->constraints(); prf_sn_s->constraints(); sn->constraints(); prf_sn->constraints(); sn_old->constraints();
    cmtS->constraints(); cmt_s->constraints(); cmtB_old->constraints(); cmt_old->constraints(); cmtB->constraints(); cmt_new->constraints();
    rt->constraints(); boolean_constraint(b, LC(value_enforce)); merkle->constraints(); }
  void assign(const DepositInputs &in) { Board &b = board; circuit::wake_helpers();                       // gadget.tcc:235-298
    fill(b, value_s, u64_bits(in.value_s));
    b.set(value_s_packed, value_by_order(b, value_s));
    fill(b, value_old, u64_bits(in.value_old));
    b.set(value_old_packed, value_by_order(b, value_old));
    fill(b, value, u64_bits(in.value)); b.set(value_packed, value_by_order(b, value));
    pk_recv->fill(blob_bits(in.pk_recv.b, 20));
    r_s->fill(blob_bits(in.r_s.b, 32));
    sn_A_old->fill(blob_bits(in.sn_A_old.b, 32));
    sn_old->fill(blob_bits(in.sn_old.b, 32));
    r_old->fill(blob_bits(in.r_old.b, 32));
    sn->fill(blob_bits(in.sn.b, 32)); r->fill(blob_bits(in.r.b, 32)); sk->fill(blob_bits(in.sk.b, 32));
    b.set_bit(value_enforce, in.value_s != 0); b.set(ZERO, HFr::zero());
    if (in.path.size() != depth || in.index_bits.size() != depth) throw std::runtime_error("deposit: Merkle path length does not match the tree depth");
    // reference order: prf_sn (writes sn), prf_sn_s (writes sn_s), sn_s := given, cmt_s, cmt_old, cmt_new (reads sn), the three commitments := given, merkle
    // (reads cmtS).
    // (round 6) ONE wave of 10 + depth compressions: every value a later step reads — sn, sn_s, the two-block hashers' intermediate digests, the digests up the
    // Merkle path — is written first (natively, in the reference's order, the given values where the reference lets them stand); the compressions then fill in their
    // internals side by side, and the given values are written once more where a compression has written its own result over them.
    prf_sn->outputs_first(); prf_sn_s->outputs_first();
    sn_s->fill(blob_bits(in.sn_s.b, 32));
    cmt_s->outputs_first(); cmt_old->outputs_first(); cmt_new->outputs_first();
    cmtS->fill(blob_bits(in.cmtS.b, 32));
    cmtB_old->fill(blob_bits(in.cmtB_old.b, 32));
    { std::vector<std::function<void()>> t; merkle->witness(in.path, in.index_bits, &t); cmt_new->add_tasks(t); prf_sn->add_tasks(t); prf_sn_s->add_tasks(t); cmt_s->add_tasks(t);
      cmt_old->add_tasks(t); run_parallel(std::move(t)); }
    sn_s->fill(blob_bits(in.sn_s.b, 32)); cmtS->fill(blob_bits(in.cmtS.b, 32)); cmtB_old->fill(blob_bits(in.cmtB_old.b, 32));
    merkle->finish(); cmtB->fill(blob_bits(in.cmtB.b, 32));
    rt->fill(blob_bits(in.rt.b, 32)); unpacker->witness_from_bits(); }
};

// libsnark's merkle_tree_check_read_gadget composed as its own self-test does (merkle_tree_check_read_gadget.tcc:131-196): address bits, leaf, root,
// path variable, then the gadget with read_successful = ONE
struct MerkleTestCircuit : Circuit {
  size_t depth; VarArray address; std::unique_ptr<Digest> leaf, root; std::unique_ptr<MerkleRead> ml;
  MerkleTestCircuit(bool emit, size_t depth) : Circuit(emit), depth(depth) {
    Board &b = board;
    address = b.alloc_array(depth);
    leaf.reset(new Digest(b, 256));
    root.reset(new Digest(b, 256));
    ml.reset(new MerkleRead(b, depth, leaf->bits, root->bits, 0 /* ONE */, address)); if (emit) ml->constraints(); b.finish(); }
};

// libsnark's own two-to-one hash test circuit (test_sha256_gadget.cpp:20-41)
struct Sha256TwoToOne : Circuit {
  std::unique_ptr<Digest> left, right, output; std::unique_ptr<Sha256Compression> f;
  explicit Sha256TwoToOne(bool emit) : Circuit(emit) {
    Board &b = board;
    left.reset(new Digest(b, 256));
    right.reset(new Digest(b, 256));
    output.reset(new Digest(b, 256));
    f.reset(new Sha256Compression(b, sha256_default_iv(), concat({left->bits, right->bits}), output->bits)); if (emit) f->constraints(); b.finish(); }
};
}  // namespace

std::unique_ptr<Circuit> make_send_circuit(bool emit) { return std::unique_ptr<Circuit>(new SendCircuit(emit)); }
void assign_send(Circuit &c, const SendInputs &in) { static_cast<SendCircuit &>(c).assign(in); }
std::unique_ptr<Circuit> make_mint_circuit(bool emit) { return std::unique_ptr<Circuit>(new MintRedeemCircuit(emit, false)); }
void assign_mint(Circuit &c, const MintInputs &in) { static_cast<MintRedeemCircuit &>(c).assign(in); }
std::unique_ptr<Circuit> make_redeem_circuit(bool emit) { return std::unique_ptr<Circuit>(new MintRedeemCircuit(emit, true)); }
void assign_redeem(Circuit &c, const RedeemInputs &in) { static_cast<MintRedeemCircuit &>(c).assign(in); }
// test circuit: the "value_s <= value_old" block of send's and redeem's note gadgets on its own (note.tcc:35-59,78-83 + less_cmp.tcc:23-34): compared with the
// reference's comparison.tcc compiled for real (oracle/ref_harness.cpp cmd_lesscmp)
struct LessCmpTestCircuit : Circuit {
  VarArray value_old, value_s; Var value_old_packed, value_s_packed; std::unique_ptr<LessCmp> less;
  explicit LessCmpTestCircuit(bool emit) : Circuit(emit) {
    Board &b = board;
    value_old = b.alloc_array(64);
    value_s = b.alloc_array(64);
    value_old_packed = b.alloc();
    value_s_packed = b.alloc();
    less.reset(new LessCmp(b, LC(value_s_packed), LC(value_old_packed)));
    if (emit) {
      bool64(b, value_old);
      bool64(b, value_s);
      less->constraints();
    }
    b.finish();
  }
  void assign(uint64_t v_old, uint64_t v_s) {
    Board &b = board;
    fill(b, value_old, u64_bits(v_old));
    b.set(value_old_packed, value_by_order(b, value_old));
    fill(b, value_s, u64_bits(v_s));
    b.set(value_s_packed, value_by_order(b, value_s));
    less->witness();
  }
};
// test circuit: one sha256_CMTA_gadget (commitment.tcc:12-110) on its own — ZERO, value, sn, r, output digest, then the gadget; compared with the same
// composition built from libsnark's own classes (oracle/ref_harness.cpp cmd_cmta)
struct CmtaTestCircuit : Circuit {
  Var ZERO; VarArray v, sn, r; std::unique_ptr<Digest> out; std::unique_ptr<ShaTwoBlock> g;
  explicit CmtaTestCircuit(bool emit) : Circuit(emit) {
    Board &b = board;
    ZERO = b.alloc();
    v = b.alloc_array(64);
    sn = b.alloc_array(256);
    r = b.alloc_array(256);
    out.reset(new Digest(b, 256));
    g = make_cmta(b, ZERO, v, sn, r, out->bits);
    if (emit) { b.constraint(ONE_LC, LC(ZERO), LC()); g->constraints(); } b.finish(); }
  void assign(const std::vector<bool> &bv, const std::vector<bool> &bsn, const std::vector<bool> &br) {
    Board &b = board;
    b.set(ZERO, HFr::zero());
    fill(b, v, bv);
    fill(b, sn, bsn);
    fill(b, r, br);
    g->witness();
  }
};
// test circuits: the CMTS / PRF / CRH blocks on their own (commitment.tcc:100-320) — ZERO, the inputs, the output digest, then the gadget; compared with the
// same compositions built from libsnark's own classes (oracle/ref_harness.cpp cmd_hashblock). which: 0 CMTS (64 | 160 | 256 | 256 bits), 1 PRF (256 | 256), 2
// CRH (160 | 256)
struct HashBlockTestCircuit : Circuit {
  int which; Var ZERO; std::vector<VarArray> in; std::unique_ptr<Digest> out; std::unique_ptr<ShaTwoBlock> g2; std::unique_ptr<ShaOneBlock> g1;
  static std::vector<size_t> widths(int which) {
    return which == 0 ? std::vector<size_t>{64, 160, 256, 256} : which == 1 ? std::vector<size_t>{256, 256} : std::vector<size_t>{160, 256};
  }
  HashBlockTestCircuit(bool emit, int which) : Circuit(emit), which(which) {
    Board &b = board;
    ZERO = b.alloc();
    for (size_t w : widths(which)) in.push_back(b.alloc_array(w));
    out.reset(new Digest(b, 256));
    if (which == 0) g2 = make_cmts(b, ZERO, in[0], in[1], in[2], in[3], out->bits);
    else if (which == 1) g2 = make_prf(b, ZERO, in[0], in[1], out->bits);
    else g1 = make_crh(b, ZERO, in[0], in[1], out->bits);
    if (emit) { b.constraint(ONE_LC, LC(ZERO), LC()); if (g2) g2->constraints(); else g1->constraints(); } b.finish(); }
  void assign(const std::vector<bool> &bits) {
    Board &b = board;
    b.set(ZERO, HFr::zero());
    size_t pos = 0;
    for (auto &a : in) {
      fill(b, a, std::vector<bool>(bits.begin() + pos, bits.begin() + pos + a.size()));
      pos += a.size();
    }
    if (g2) g2->witness();
    else g1->witness();
  }
};
// test circuit: the public-input unpacker of the four circuits on its own (e.g. send/circuit/gadget.tcc:87-106,198: libsnark's multipacking_gadget over the unpacked bits with
// chunks of Fr's capacity, constraints with enforce_bitness) — packed inputs, then the bits; compared with libsnark's own class (oracle/ref_harness.cpp cmd_unpacker)
struct UnpackerTestCircuit : Circuit {
  VarArray packed, bits; std::unique_ptr<MultiPacking> up;
  UnpackerTestCircuit(bool emit, size_t nbits) : Circuit(emit) {
    Board &b = board;
    packed = b.alloc_array((nbits + 252) / 253); b.set_input_sizes(packed.size());
    bits = b.alloc_array(nbits);
    up.reset(new MultiPacking(b, bits, packed));
    if (emit) up->constraints(true);
    b.finish();
  }
};
std::unique_ptr<Circuit> make_unpacker_test_circuit(bool emit, size_t nbits) { return std::unique_ptr<Circuit>(new UnpackerTestCircuit(emit, nbits)); }
void assign_unpacker_test(Circuit &c, const std::vector<bool> &bits) { auto &u = static_cast<UnpackerTestCircuit &>(c); fill(u.board, u.bits, bits); u.up->witness_from_bits(); }
std::unique_ptr<Circuit> make_hashblock_test_circuit(bool emit, int which) { return std::unique_ptr<Circuit>(new HashBlockTestCircuit(emit, which)); }
size_t hashblock_input_bits(int which) { size_t n = 0; for (size_t w : HashBlockTestCircuit::widths(which)) n += w; return n; }
void assign_hashblock_test(Circuit &c, const std::vector<bool> &bits) { static_cast<HashBlockTestCircuit &>(c).assign(bits); }
std::unique_ptr<Circuit> make_cmta_test_circuit(bool emit) { return std::unique_ptr<Circuit>(new CmtaTestCircuit(emit)); }
void assign_cmta_test(Circuit &c, const std::vector<bool> &v, const std::vector<bool> &sn, const std::vector<bool> &r) {
  static_cast<CmtaTestCircuit &>(c).assign(v, sn, r);
}
std::unique_ptr<Circuit> make_lesscmp_test_circuit(bool emit) { return std::unique_ptr<Circuit>(new LessCmpTestCircuit(emit)); }
void assign_lesscmp_test(Circuit &c, uint64_t value_old, uint64_t value_s) { static_cast<LessCmpTestCircuit &>(c).assign(value_old, value_s); }
std::unique_ptr<Circuit> make_sha256_two_to_one(bool emit) { return std::unique_ptr<Circuit>(new Sha256TwoToOne(emit)); }
void assign_sha256_two_to_one(Circuit &c, const std::vector<bool> &l, const std::vector<bool> &r) {
  auto &s = static_cast<Sha256TwoToOne &>(c);
  s.left->fill(l);
  s.right->fill(r);
  s.f->witness();
}

}  // namespace zk
namespace zk {
std::unique_ptr<Circuit> make_merkle_test_circuit(bool emit, size_t depth) { return std::unique_ptr<Circuit>(new MerkleTestCircuit(emit, depth)); }
void assign_merkle_test(Circuit &c, const Blob256 &leaf, const std::vector<Blob256> &path, const std::vector<bool> &index_bits, const Blob256 &root) {
  auto &m = static_cast<MerkleTestCircuit &>(c);
  m.leaf->fill(blob_bits(leaf.b, 32)); m.ml->witness(path, index_bits); m.root->fill(blob_bits(root.b, 32)); }
std::unique_ptr<Circuit> make_deposit_circuit(bool emit, size_t tree_depth) { return std::unique_ptr<Circuit>(new DepositCircuit(emit, tree_depth)); }
void assign_deposit(Circuit &c, const DepositInputs &in) { static_cast<DepositCircuit &>(c).assign(in); }
}
