// Groth16 (libsnark's r1cs_gg_ppzksnark) on the MI355X engine: key containers, the reference's key-file format, key
// generation, the prover pipeline and the verifier.
// Reference: SNARK/zk_proof_systems/ppzksnark/r1cs_gg_ppzksnark/r1cs_gg_ppzksnark.{hpp,tcc}.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "gpu.hpp"
#include "pairing_host.hpp"

namespace zk {

// All group elements affine, Montgomery form; the all-zero record is the point at infinity.
struct ProvingKeyHost {                           // r1cs_gg_ppzksnark.hpp:72-110
  G1AffineRaw alpha_g1, beta_g1, delta_g1; G2AffineRaw beta_g2, delta_g2;
  std::vector<G1AffineRaw> A;                     // n_vars + 1 (entries may be infinity)
  std::vector<uint32_t> B_idx; std::vector<G2AffineRaw> B_g2; std::vector<G1AffineRaw> B_g1;   // sparse knowledge-commitment vector
  std::vector<G1AffineRaw> H;                     // m - 1
  // likewise: n_vars + 1 points, the L query extended to all variables minus the C polynomial's share of the H term (ecntt.cuh)
  mutable std::vector<G1AffineRaw> L_star;
  mutable std::vector<G1AffineRaw> H_lagrange;    // filled by the first Prover built on this key: H in the Lagrange basis of the coset, m points (ecntt.cuh)
  std::vector<G1AffineRaw> L;                     // n_vars - n_inputs
  R1csHost cs;                                    // as stored in the key (A/B already swapped if the generator found it beneficial)
};
struct VerifyingKeyHost {                         // r1cs_gg_ppzksnark.hpp:170-200
  host::HFq12 alpha_g1_beta_g2; G2AffineRaw gamma_g2, delta_g2; std::vector<G1AffineRaw> IC;   // IC[0] = gamma_ABC_g1.first
};
struct Proof { G1AffineRaw A; G2AffineRaw B; G1AffineRaw C; };   // affine, Montgomery

// ---- the reference's on-disk format (hybrid binary/decimal) ---------------------------------------------
ProvingKeyHost load_proving_key(const std::string &path);       // point decompression runs on the GPU
VerifyingKeyHost load_verifying_key(const std::string &path);   // host only
void save_proving_key(const std::string &path, const ProvingKeyHost &pk);
void save_verifying_key(const std::string &path, const VerifyingKeyHost &vk);

// ---- the fast key container ------------------------------------------------------------------------------ The reference parses its key
// file on every call (sendcgo.cpp:64-81: 54 s for send); this engine parses it once per process — still 0.9 s: 77 MB of decimal text, 1.08 M square roots, and
// the key transforms of ecntt.cuh. The container holds the RESULT of all that as raw, 64-byte aligned arrays (uncompressed affine Montgomery points with H
// already in the coset's Lagrange basis and C folded into L, the constraint system in CSR form), so a later process start maps the file and copies:
//   header: magic "ZKGPUKC1", flags, the source key file's size and mtime (a stale or foreign container is ignored), shape, payload length, 64-bit checksum
//   payload: alpha_g1 beta_g1 delta_g1 | beta_g2 delta_g2 | A | B_idx | B_g1 | B_g2 | H_lagrange | L_star | rowptr[3] col[3] coeff[3]
// Written next to the key file as <key>.gpucache (or under $ZK_KEY_CACHE_DIR) after the first load from text; ZK_KEY_CACHE=0 disables both reading and writing.
struct KeyStamp { int64_t size = -1, mtime_s = 0, mtime_ns = 0; bool operator==(const KeyStamp &o) const {
    return size == o.size && mtime_s == o.mtime_s && mtime_ns == o.mtime_ns; } };
bool key_stamp_of(const std::string &path, KeyStamp &out);
std::string key_container_path(const std::string &pk_path);            // "" if the cache is disabled
// pk must carry H_lagrange and L_star (filled by the first Prover built on it); atomic (temporary file + rename)
void save_key_container(const std::string &path, const ProvingKeyHost &pk, const KeyStamp &source);
// false: missing, stale, truncated, wrong checksum or wrong version — the caller falls back to the text key
bool load_key_container(const std::string &path, const KeyStamp &source, ProvingKeyHost &pk);
// text key or its container, whichever is valid; `from_container` tells which. After building the first Prover on a key that came from text, call
// save_key_container.
ProvingKeyHost load_proving_key_fast(const std::string &pk_path, bool &from_container);

// ---- generator (r1cs_gg_ppzksnark.tcc:212-388) --------------------------------------------------------------------------
struct ToxicWaste { host::HFr t, alpha, beta, gamma, delta, g1_scalar, g2_scalar; static ToxicWaste random(); static ToxicWaste from_seed(uint64_t seed); };
void generate_keys(const R1csHost &cs, const ToxicWaste &tw, ProvingKeyHost &pk, VerifyingKeyHost &vk);

// ---- prover (r1cs_gg_ppzksnark.tcc:391-506) -------------------------------------------------------------------------------
class Prover {                                    // a proving key resident in HBM
 public:
  // shard_rank / shard_world: this object holds only the contiguous slice [n*rank/world, n*(rank+1)/world) of every query (kernel K7);
  // a sharded prover produces partial sums (prove_partial), any process then adds the ranks' partials and assembles the proof (finish_from_partials)
  // device_slot: index into the process's device list (ZK_DEVICES)
  explicit Prover(const ProvingKeyHost &pk, size_t shard_rank = 0, size_t shard_world = 1, int device_slot = 0);
  ~Prover();
  // a second prover on the same key: shares the peer's immutable device state (query tables, twiddles, constraint system: 1.8 GB for send) and owns only its
  // streams, sort / bucket workspaces and vectors (about 0.25 GB), so a pool of provers per key costs little HBM and no second key load
  explicit Prover(const Prover &peer);
  int device_slot() const;
  size_t num_variables() const; size_t num_inputs() const; size_t domain_size() const;
  // z: full assignment without ONE (canonical).  r, s: prover randomness (canonical; nullptr = fresh CSPRNG values).
  // Returns false if z does not satisfy the constraint system (the reference then emits its default proof, sendcgo.cpp:209-214).
  bool prove(const Fe32 *z, const Fe32 *r, const Fe32 *s, Proof &out) { set_witness(z, false); return prove_resident(r, s, out); }
  // the two halves of prove(): hand the assignment over (canonical, or already in Montgomery form as the circuit boards hold it), then prove from HBM
  void set_witness(const Fe32 *z, bool montgomery);
  // the same assignment as one byte per entry — 0, 1, 2 = "see wide[i]" (Montgomery form) or 6 = "the small integer in the low 64 bits of wide[i]" — entry 0
  // being the constant ONE (circuit::Board's own form)
  void set_witness_tagged(const uint8_t *tag, const Fe32 *wide);
  // The same from a circuit board that also says which variables EVER held something else than 0 / 1 (circuit::Board::ever_wide, marks = its counter of such
  // first-time marks): the tag bytes go up as they are, the values of exactly those variables in a fixed order — no scan, no compaction on the host (0.085 -> 0.03 ms for send)
  // tag_dev / wide_dev (optional): the device's addresses of the two arrays where the board's memory is pinned and mapped (gpu_host_register): the device then reads
  // them in place — the calling thread only launches
  void set_witness_board(const uint8_t *tag, const Fe32 *wide, const uint8_t *ever_wide, uint32_t marks, const uint8_t *tag_dev = nullptr, const Fe32 *wide_dev = nullptr);
  bool prove_resident(const Fe32 *r, const Fe32 *s, Proof &out);
  // Inputs resident in HBM: stash_witness() keeps the assignment that was handed over last in device memory — the RAW vector only, (n + 1) x 32 B in
  // Montgomery form as libsnark holds it, nothing derived from it — and returns its slot (a dropped slot is reused); prove_stashed(slot, ...) proves it in place:
  // tags and the list of the other values are derived by a device kernel inside the call (k_classify_witness), no host buffer, no PCIe.  bench.py's `value`: a
  // batch of distinct statements uploaded before the timed region starts.  drop_stash(slot) frees one, drop_stash(-1) all of them.
  size_t stash_witness();
  void drop_stash(size_t slot);
  size_t stash_count() const;
  void read_stash(size_t slot, Fe32 *out);   // the kept assignment, canonical, n_vars x 32 bytes (tests, diagnostics)
  size_t equal_column_groups() const;   // groups of variables with identical columns found in the key (their values are folded at the head of every proof)
  bool prove_stashed(size_t slot, const Fe32 *r, const Fe32 *s, Proof &out);
  // partial multi-exponentiation results of this shard, affine canonical: eA(64) eB1(64) eH(64) eL(64) eB2(128) = 384 bytes.  false if z is unsatisfying.
  static constexpr size_t PARTIAL_BYTES = 384;
  bool prove_partial(uint8_t out[PARTIAL_BYTES]);
  // sum `n` shard records and assemble the proof (r1cs_gg_ppzksnark.tcc:487-495); needs only the key's alpha/beta/delta, no device work
  void finish_from_partials(const uint8_t *records, size_t n, const Fe32 *r, const Fe32 *s, Proof &out);
  struct Timings { double upload_ms, qap_ms, msm_ms, finish_ms, total_ms; } last{};
  long last_failed_row = -1;   // a constraint the last proof found violated (prove* returned false)
  struct Impl; std::unique_ptr<Impl> impl;
};

// ---- verifier (r1cs_gg_ppzksnark.tcc:509-623) -------------------------------------------------------------------------------
bool verify_proof(const VerifyingKeyHost &vk, const Fe32 *inputs /* canonical */, size_t n_inputs, const Proof &proof);
// The key-dependent half of the check done once (libsnark's r1cs_gg_ppzksnark_verifier_process_vk, :509-522, plus window tables of the IC points): the line
// coefficients of gamma_g2 and delta_g2, and 2^(8w) d IC_j for every byte value d.  A call then costs three Miller loops, one final exponentiation and 32
// mixed additions per public input instead of a 254-bit scalar multiplication each — 2.5 -> about 1.3 ms per proof on the host.
struct PreparedVerifyingKey { VerifyingKeyHost vk; host::G2Precomp gamma, delta; std::vector<host::HFq> ic_x, ic_y;
    /* [j][w*255 + d-1], affine; (0,0) = infinity */ };
std::shared_ptr<PreparedVerifyingKey> prepare_verifying_key(const VerifyingKeyHost &vk);
bool verify_proof(const PreparedVerifyingKey &pvk, const Fe32 *inputs /* canonical */, size_t n_inputs, const Proof &proof);

// the GPU verifier's schedule interpreted on the host (test entry)
bool verify_by_schedule_on_host(const PreparedVerifyingKey &pvk, const Fe32 *inputs, size_t n_inputs, const Proof &proof, uint32_t stats[8]);
// kernel K9's LDS values after every `every`-th round against vsched::simulate29 on the same inputs: -1 = all equal, else the first differing round (GPU tests)
long verify_schedule_trace_on_device(BatchVerifier &bv, const PreparedVerifyingKey &pvk, const Fe32 *inputs, size_t n_inputs, const Proof &proof, uint32_t every,
    uint32_t *bad_slot, uint8_t *device_ok);
// the same decision for n proofs at once on the GPU (kernel K9): one BatchVerifier per verifying key
std::unique_ptr<BatchVerifier> make_batch_verifier(const VerifyingKeyHost &vk);
static_assert(sizeof(Proof) == 256, "proof record");

// proof <-> the 512-hex-character form of the cgo wrappers (sendcgo.cpp:113-188, :388-448)
std::string proof_to_hex(const Proof &p);
bool proof_from_hex(const char *hex, Proof &p);   // reads exactly 512 characters; false on a non-hex character
Proof default_proof();                            // (G1::one, G2::one, G1::one) — r1cs_gg_ppzksnark.hpp:309-315

// host-only self-test of the hand-over's block classifiers (scalar against AVX2 forms)
void test_scan_blocks(const uint8_t tags[64], const uint64_t elems[256], const uint64_t one[4], uint64_t out[10]);
// auxiliary variables whose columns are identical in A, B and C, in groups of two or more (host only; the prover folds their values, k_merge_equal_columns)
std::vector<std::vector<uint32_t>> equal_column_groups(const R1csHost &cs);
int test_cgroup_quota(const char *root);   // host-only: the CPU quota (CPUs, rounded up) a cgroup tree states — cpu.max (v2) or cpu/cpu.cfs_*_us (v1) under `root`; 0 = none
int test_scan_pool(int callers, int rounds);   // host-only self-test of the hand-over's scan pool: rounds that ran on the pool, -1 on a miscount
}  // namespace zk
