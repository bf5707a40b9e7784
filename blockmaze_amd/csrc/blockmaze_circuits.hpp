// BlockMaze's four statement circuits (mint / send / deposit / redeem) on circuit::Board, and the host-side note
// hashing they are checked against.  Sources: libsnark-vnt/src/{mint,send,deposit,redeem}/circuit/*.tcc (gadget
// composition and allocation order), libsnark-vnt/src/*/Note.h, util.h, uint256.h (host hashing and byte order).
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
#include "circuit.hpp"

namespace zk {

// 256-/160-bit opaque blobs in the reference's byte order: byte k is the k-th least significant byte of the hex value
// (base_blob::SetHex, libsnark-vnt/src/send/uint256.h:222-248)
struct Blob256 { uint8_t b[32]; };
struct Blob160 { uint8_t b[20]; };
Blob256 blob256_from_hex(const char *s);          // uint256S: skips blanks and "0x", reads hex digits from the end, zero-extends
Blob160 blob160_from_hex(const char *s);
std::string blob_to_hex(const uint8_t *b, size_t n);   // GetHex: bytes reversed, lowercase (uint256.h:190-196)

void sha256(const uint8_t *msg, size_t len, uint8_t out[32]);
void sha256_compress_raw(const uint8_t left[32], const uint8_t right[32], uint8_t out[32]);   // SHA256Compress::combine (IncrementalMerkleTree.tcc:14-24)
Blob256 note_cm(uint64_t value, const Blob256 &sn, const Blob256 &r);                          // Note::cm   (send/Note.h:30-44)
Blob256 note_s_cm(uint64_t value, const Blob160 &pk, const Blob256 &r, const Blob256 &sn_old); // NoteS::cm  (send/Note.h:63-78)
Blob256 compute_prf(const Blob256 &sk, const Blob256 &r);                                      // util.h:233-245
Blob256 compute_crh(const Blob160 &pk, const Blob256 &r);                                      // util.h:247-258
// root of the depth-8 incremental Merkle tree over the first n leaves, empty leaves = 0 (depositcgo.cpp:302-325); path of leaf `index`
Blob256 merkle_root(const std::vector<Blob256> &leaves, size_t depth);
std::vector<Blob256> merkle_path(const std::vector<Blob256> &leaves, size_t depth, size_t index, std::vector<bool> &index_bits);

enum class CircuitKind { Mint = 0, Send = 1, Deposit = 2, Redeem = 3 };
const char *circuit_name(CircuitKind k);

struct MintInputs { uint64_t value, value_old, value_s; Blob256 sn_old, r_old, sn, r, cmtA_old, cmtA, sk; };
struct RedeemInputs { uint64_t value, value_old, value_s; Blob256 sn_old, r_old, sn, r, cmtA_old, cmtA, sk; };
struct SendInputs { uint64_t value_old, value_s, value; Blob256 sn_old, r_old, r_s, sn, r, cmtA_old, cmtS, cmtA, sk; Blob160 pk_recv, pk_sender; };
struct DepositInputs { uint64_t value, value_old, value_s; Blob256 sn_old, r_old, sn, r, sn_s, r_s, sn_A_old, cmtB_old, cmtB, cmtS, sk, rt; Blob160 pk_recv;
    std::vector<Blob256> path; std::vector<bool> index_bits; };

// A circuit instance: construct once (allocates variables; with emit = true also emits the constraint system), then
// assign() any number of witnesses.
class Circuit {
 public:
  virtual ~Circuit() {}
  circuit::Board board;
  explicit Circuit(bool emit) : board(emit) {}
  const R1csHost &r1cs() const { return board.cs; }
  // full assignment without ONE, canonical 32-byte little-endian values
  void export_assignment(std::vector<Fe32> &z) const;
  size_t num_inputs() const { return board.cs.n_inputs; }
};
std::unique_ptr<Circuit> make_send_circuit(bool emit);
void assign_send(Circuit &c, const SendInputs &in);
std::unique_ptr<Circuit> make_mint_circuit(bool emit);
void assign_mint(Circuit &c, const MintInputs &in);
std::unique_ptr<Circuit> make_redeem_circuit(bool emit);
void assign_redeem(Circuit &c, const RedeemInputs &in);
std::unique_ptr<Circuit> make_deposit_circuit(bool emit, size_t tree_depth);
void assign_deposit(Circuit &c, const DepositInputs &in);
// test circuit: libsnark's sha256_two_to_one_hash_gadget exactly as gadgets/hashes/sha256/tests/test_sha256_gadget.cpp:20-41 builds it
std::unique_ptr<Circuit> make_sha256_two_to_one(bool emit);
void assign_sha256_two_to_one(Circuit &c, const std::vector<bool> &left, const std::vector<bool> &right);

// test circuit: BlockMaze's less_comparison_gadget block (send/circuit/comparison.tcc:5-96 as composed by note.tcc / less_cmp.tcc)
std::unique_ptr<Circuit> make_lesscmp_test_circuit(bool emit);
void assign_lesscmp_test(Circuit &c, uint64_t value_old, uint64_t value_s);

// test circuit: one sha256_CMTA_gadget (send/circuit/commitment.tcc:12-110): two chained compressions with hard-wired padding; bit vectors of 64 / 256 / 256
// entries
std::unique_ptr<Circuit> make_cmta_test_circuit(bool emit);
std::unique_ptr<Circuit> make_hashblock_test_circuit(bool emit, int which /* 0 CMTS, 1 PRF, 2 CRH */);
size_t hashblock_input_bits(int which);
void assign_hashblock_test(Circuit &c, const std::vector<bool> &bits);
void assign_cmta_test(Circuit &c, const std::vector<bool> &v, const std::vector<bool> &sn, const std::vector<bool> &r);

// test circuit: the public-input unpacker (multipacking_gadget, enforce_bitness) over nbits bits
std::unique_ptr<Circuit> make_unpacker_test_circuit(bool emit, size_t nbits);
void assign_unpacker_test(Circuit &c, const std::vector<bool> &bits);

// test circuit: libsnark's merkle_tree_check_read_gadget as its self-test composes it (merkle_tree_check_read_gadget.tcc:131-196)
std::unique_ptr<Circuit> make_merkle_test_circuit(bool emit, size_t depth);
void assign_merkle_test(Circuit &c, const Blob256 &leaf, const std::vector<Blob256> &path /* leaf level first */, const std::vector<bool> &index_bits,
    const Blob256 &root);

// public inputs of a statement packed the way the verifier side does it (X_gadget::witness_map, e.g. send/circuit/gadget.tcc:274-291;
// pack_bit_vector_into_field_element_vector, field_utils.tcc:78-102): 253-bit chunks, little-endian within a chunk
std::vector<Fe32> pack_public_bits(const std::vector<bool> &bits);
std::vector<bool> blob_bits(const uint8_t *b, size_t nbytes);     // uint256_to_bool_vector: byte order of the blob, MSB first inside a byte
std::vector<bool> u64_bits(uint64_t v);                           // uint64_to_bool_vector: little-endian bytes, MSB first inside a byte

}  // namespace zk
