/* zkgpu.h — C-ABI of libzkgpu.so, the MI355X-native engine behind BlockMaze's libzk_{mint,send,deposit,redeem}.so.
 *
 * Two layers:
 *   1. the drop-in layer: the exact cgo symbols of the reference (declared in zk_mint.h / zk_send.h / zk_deposit.h /
 *      zk_redeem.h next to this file), exported by libzkgpu.so and re-exported by the four thin libzk_*.so;
 *   2. this header: the building blocks underneath, exposed so that parity tests and the benchmark can drive each
 *      kernel on its own.  They correspond to the reference's C++ entry points one level below the cgo wrappers:
 *        zkgpu_msm_g1/g2      <- libff::multi_exp / multi_exp_with_mixed_addition
 *                                (libsnark-vnt/depends/libsnark/depends/libff/libff/algebra/scalar_multiplication/multiexp.tcc:403-496)
 *        zkgpu_domain_*       <- libfqfft::evaluation_domain::{FFT,iFFT,cosetFFT,icosetFFT}
 *                                (depends/libfqfft/libfqfft/evaluation_domain/domains/basic_radix2_domain.tcc:48-88, step_radix2_domain.tcc:39-167)
 *        zkgpu_witness_map    <- libsnark::r1cs_to_qap_witness_map (libsnark/reductions/r1cs_to_qap/r1cs_to_qap.tcc:206-334)
 *        zkgpu_prover_*       <- libsnark::r1cs_gg_ppzksnark_prover (zk_proof_systems/ppzksnark/r1cs_gg_ppzksnark/r1cs_gg_ppzksnark.tcc:391-506)
 *                                + the key loading of libsnark-vnt/src/send/sendcgo.cpp:64-81,345
 *
 * Conventions: all pointers are HOST pointers.  A field element is 32 bytes, little-endian, CANONICAL (not Montgomery).
 * G1 affine = x | y (64 bytes), G2 affine = x.c0 | x.c1 | y.c0 | y.c1 (128 bytes); the point at infinity is all zero bytes.
 * Every function returns 0 on success and a negative code on failure; zkgpu_last_error() gives the message.
 * There is NO CPU fallback: without a HIP device every compute entry point fails with ZKGPU_ERR_NO_DEVICE.
 */
#ifndef ZKGPU_H
#define ZKGPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZKGPU_OK 0
#define ZKGPU_ERR_NO_DEVICE (-1)
#define ZKGPU_ERR_ARG (-2)
#define ZKGPU_ERR_RUNTIME (-3)
#define ZKGPU_ERR_UNSATISFIED (-4)

const char *zkgpu_last_error(void);
const char *zkgpu_version(void);
int zkgpu_device_count(void);                       /* number of visible HIP devices (0 on a CPU-only host) */
int zkgpu_device_numa_node(int device);             /* NUMA node of the host socket visible device `device` hangs off (sysfs), -1 unknown: what a rank launcher binds its process to */
int zkgpu_init(void);                               /* create the device context now instead of lazily */

/* ---- device arithmetic probes (parity tests of the __device__ field / curve code) ---------------------------------- */
/* field: 0 = Fr, 1 = Fq.  op: 0 mul, 1 add, 2 sub, 3 inverse(a), 4 square(a), 5 negate(a) */
int zkgpu_test_field_op(int field, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
/* (field ops 6..9 probe the lazy domain of field.cuh: 6 mul, 7 square, 8 sub on operands pushed towards 2p, 9 masked negation; results normalized) */
/* op: 0 mul, 1 square(a), 2 inverse(a) on Fq2 (64-byte elements c0 | c1) */
int zkgpu_test_fq2_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
/* group: 1 = G1, 2 = G2.  op: 0 general add, 1 double(a), 2 mixed add (b affine), 3 a*k for 32-bit k (k in b's first 4 bytes) */
int zkgpu_test_group_op(int group, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);

/* ---- multi-scalar multiplication ------------------------------------------------------------------------------------ */
/* window_bits 0 = choose from n.  filter_ones: bit 0 = treat scalars 0 / 1 specially like multi_exp_with_mixed_addition; bit 1 = the scalars are known to be
   uniform (H query): one-pass sort with fixed slots per bucket, falling back to the two-pass sort if a bucket overflows (results are the same either way). */
int zkgpu_msm_g1(const uint8_t *points, const uint8_t *scalars, size_t n, int window_bits, int filter_ones, uint8_t out[64]);
int zkgpu_msm_g2(const uint8_t *points, const uint8_t *scalars, size_t n, int window_bits, int filter_ones, uint8_t out[128]);

/* resident form for benchmarking: bases stay in HBM, scalars are uploaded once, run() times only the kernels */
typedef struct zkgpu_msm zkgpu_msm;
zkgpu_msm *zkgpu_msm_create(int group, const uint8_t *points, size_t n, int window_bits, int filter_ones);
int zkgpu_msm_set_scalars(zkgpu_msm *h, const uint8_t *scalars, size_t n);
int zkgpu_msm_run(zkgpu_msm *h, uint8_t *out);     /* out: 64 or 128 bytes */
void zkgpu_msm_destroy(zkgpu_msm *h);

/* ---- evaluation domains --------------------------------------------------------------------------------------------- */
size_t zkgpu_domain_size(size_t min_size);          /* m chosen by get_evaluation_domain for this minimum size, 0 if none */
/* op: 0 FFT, 1 iFFT, 2 cosetFFT (g = 5), 3 icosetFFT.  data: m elements, transformed in place */
int zkgpu_domain_transform(size_t min_size, int op, uint8_t *data);

/* ---- R1CS / QAP ------------------------------------------------------------------------------------------------------- */
typedef struct zkgpu_r1cs zkgpu_r1cs;
/* CSR per matrix: rowptr (n_cons+1), col (nnz, 0 = the constant ONE), coeff (nnz x 32 bytes canonical) */
zkgpu_r1cs *zkgpu_r1cs_create(size_t n_inputs, size_t n_vars, size_t n_cons, const uint32_t *const rowptr[3], const uint32_t *const col[3], const uint8_t *const coeff[3]);
void zkgpu_r1cs_destroy(zkgpu_r1cs *cs);
/* z: n_vars elements (without ONE).  h_out: (m+1) elements, m = zkgpu_domain_size(n_cons + n_inputs + 1).  Returns ZKGPU_ERR_UNSATISFIED if z violates a constraint. */
int zkgpu_witness_map(zkgpu_r1cs *cs, const uint8_t *z, uint8_t *h_out);

/* ---- circuits, keys, resident prover, verifier -------------------------------------------------------------------- */
/* kind: 0 mint, 1 send, 2 deposit, 3 redeem (100 / 101 / 102 / 103 = test circuits: libsnark's sha256 two-to-one, Merkle check-read, BlockMaze's less-comparison block, one sha256_CMTA_gadget; 104..106 the CMTS / PRF / CRH blocks; 107 the public-input unpacker over tree_depth bits).  tree_depth only matters for deposit (reference: 8) and kind 107. */
/* writes the circuit's constraint system as an "R1CSBM01" file: magic, u64 n_inputs / n_vars / n_cons, then per matrix u64 nnz, u32 rowptr[n_cons+1], u32 col[nnz], 32-byte coeff[nnz] */
int zkgpu_circuit_export(int kind, int tree_depth, const char *r1cs_path);
/* witness files: u64 n, then n 32-byte canonical values (the full assignment without ONE).  Same arguments as the gen*proof symbols. */
int zkgpu_witness_sha256(const uint8_t left[32], const uint8_t right[32], const char *wit_path);
int zkgpu_witness_lesscmp(uint64_t value_old, uint64_t value_s, const char *wit_path);
int zkgpu_witness_unpacker(int nbits, const uint8_t *bits /* nbits bytes of 0 / 1 */, const char *wit_path);
int zkgpu_witness_hashblock(int which /* 0 CMTS, 1 PRF, 2 CRH: circuit kinds 104..106 of zkgpu_circuit_export */, const uint8_t *bits /* 736 / 512 / 416 bytes of 0 / 1 */, const char *wit_path);
int zkgpu_witness_cmta(const uint8_t *bits /* 576 bytes of 0 / 1: value[64], sn[256], r[256] */, const char *wit_path);
int zkgpu_witness_send(uint64_t value_A, char *r_s, char *sn, char *r, char *cmt_s, char *cmtA, uint64_t value_s, char *pk_recv, uint64_t value_A_new, char *sn_A_new,
                       char *r_A_new, char *cmt_A_new, char *sk, char *pk_sender, const char *wit_path);
int zkgpu_witness_deposit(uint64_t value, uint64_t value_old, char *sn_old, char *r_old, char *sn, char *r, char *sns, char *rs, char *cmtB_old, char *cmtB, uint64_t value_s, char *pk,
                          char *sn_A_old, char *cmtS, char *cmtarray, int n, char *sk, int tree_depth, const char *wit_path);
/* kind 101 test circuit: leaf, `depth` siblings (leaf level first, 32 bytes each in hashing byte order) and the leaf position; the root is computed */
int zkgpu_witness_merkle(int depth, const uint8_t leaf[32], const uint8_t *siblings, uint64_t position, const char *wit_path);
int zkgpu_witness_mint_redeem(int redeem, uint64_t value, uint64_t value_old, char *sn_old, char *r_old, char *sn, char *r, char *cmtA_old, char *cmtA, uint64_t value_s, char *sk, const char *wit_path);
/* key generation (r1cs_gg_ppzksnark_generator, r1cs_gg_ppzksnark.tcc:212-388; the *_key executables of libsnark-vnt/src/X/getpvk.cpp).  seed 0 = fresh randomness from the
 * OS; any other seed gives reproducible TEST keys.  Files are written in the reference's key-file format. */
int zkgpu_keygen(int kind, int tree_depth, uint64_t seed, const char *pk_path, const char *vk_path);
int zkgpu_keygen_from_r1cs(const char *r1cs_path, uint64_t seed, const char *pk_path, const char *vk_path);
/* resident prover over a reference-format proving key file */
typedef struct zkgpu_prover zkgpu_prover;
zkgpu_prover *zkgpu_prover_load(const char *pk_path);   /* parses the reference-format key file, or maps its container <pk_path>.gpucache when that is valid; writes the container after a load from text */
int zkgpu_key_container_valid(const char *pk_path);
int zkgpu_test_device_plan(const char *spec, int n_visible, int fallback, int per_device, int *out_devices, int *out_order, int n_order);   /* multi-device pool planning (pure host logic) */
int zkgpu_test_pool_plan(int n_devices, int spill, const int *release_before, int n_calls, int *out_dev);   /* acquire_prover's device choice replayed on the host (capi_zk.cpp) */
int zkgpu_test_scan_blocks(const uint8_t *tags64, const uint64_t *elems64x4, const uint64_t *one4, uint64_t *out10);   /* the hand-over's block classifiers: out[0..2] / [3..5] tag masks scalar / fast, out[6..7] / [8..9] element masks scalar / fast */
int zkgpu_test_equal_columns(const char *r1cs_path, uint32_t *out, size_t cap);   /* host only: groups of variables with identical columns in A, B and C, flattened [size, members ...]; returns the words written / needed */
int zkgpu_test_cgroup_quota(const char *root);   /* the CPU quota the library would respect when sizing its helper pools (host only): CPUs, rounded up, 0 = none */
int zkgpu_test_scan_pool(int callers, int rounds);   /* the hand-over's scan pool driven from several threads at once (host only): rounds served by the pool, -1 on a miscount */
int zkgpu_test_lane_plan(int n_slots, int kinds, int per_kind, int *out_lanes_per_slot);   /* the stream-lane planner with its per-device quota (gpu.hip) */
int zkgpu_test_key_container(const char *path, size_t n_vars, size_t n_cons, size_t m);   /* host-only self-test of the container reader / writer; 0 = passed */     /* 1 if a valid container (matching size / mtime of the key file, checksum) is in place */
/* MSM sharding across GPUs (one process per GPU): a shard holds the contiguous slice rank/world of every query of the key.  prove_partial() runs the whole
 * device pipeline on the resident witness and returns this shard's five partial sums (affine canonical: eA 64 | eB1 64 | eH 64 | eL 64 | eB2 128 = 384 bytes);
 * the records of all ranks are exchanged by the caller (one all-gather) and zkgpu_prover_finish() adds them and assembles the proof on the host. */
zkgpu_prover *zkgpu_prover_load_shard(const char *pk_path, size_t shard_rank, size_t shard_world);
int zkgpu_prover_prove_partial(zkgpu_prover *h, uint8_t out[384]);
int zkgpu_prover_finish(zkgpu_prover *h, const uint8_t *records, size_t n_records, const uint8_t *r, const uint8_t *s, char proof_hex[513]);
void zkgpu_prover_destroy(zkgpu_prover *h);
int zkgpu_prover_info(zkgpu_prover *h, size_t out[3]);          /* n_vars, n_inputs, domain size m */
/* z: n_vars elements; r, s: 32-byte canonical prover randomness or NULL for fresh values.  proof_hex: 512 hex characters + NUL. */
int zkgpu_prover_prove(zkgpu_prover *h, const uint8_t *z, const uint8_t *r, const uint8_t *s, char proof_hex[513]);
/* the two halves of zkgpu_prover_prove: upload the assignment into HBM (returns when it is resident), then prove from there any number of times */
int zkgpu_prover_set_witness(zkgpu_prover *h, const uint8_t *z);
int zkgpu_prover_prove_resident(zkgpu_prover *h, const uint8_t *r, const uint8_t *s, char proof_hex[513]);
/* inputs resident in HBM: keep the assignment handed over last in device memory — the RAW vector, (n_vars + 1) x 32 bytes, nothing derived from it; its slot is
 * returned, a dropped slot is reused — / prove a kept assignment in place: the tags and the list of values other than 0 and 1 (the classification of libsnark's
 * multi_exp_with_mixed_addition, multiexp.tcc:443-496) are derived by a device kernel INSIDE the call; no host buffer, no copy.  What bench.py's `value` times:
 * distinct statements uploaded before the timed region.  drop_stash frees a slot (0xffffffff: all of them); ZKGPU_ERR_* if no assignment was handed over / no such slot. */
int zkgpu_prover_stash_witness(zkgpu_prover *h, uint32_t *slot);
int zkgpu_prover_drop_stash(zkgpu_prover *h, uint32_t slot);
int zkgpu_prover_stash_count(zkgpu_prover *h, uint32_t *count);
/* a kept assignment copied back to the host in the layout zkgpu_prover_set_witness takes (n_vars x 32 bytes, canonical): tests and diagnostics — after a proof has read
 * the slot in place, variables with equal columns hold their folded (equivalent) values */
int zkgpu_prover_read_stash(zkgpu_prover *h, uint32_t slot, uint8_t *z_out);
/* groups of variables whose columns coincide in A, B and C found in this key (their values are folded into one place at the head of every proof: an equivalent assignment,
 * no equal points meeting in an incomplete addition); and, process-wide, how often a fast MSM path raised its flag and the MSM was repeated on the general path */
int zkgpu_prover_equal_column_groups(zkgpu_prover *h, uint32_t *count);
uint64_t zkgpu_general_path_repeats(void);
/* key queries this process loaded WITHOUT their fixed-base tables because the tables did not fit the device's free memory: proofs on such a key are several times slower —
 * a deployment can read the degradation here (and on stderr) instead of guessing it from the proof rate */
uint64_t zkgpu_queries_without_tables(void);
int zkgpu_prover_prove_stashed(zkgpu_prover *h, uint32_t slot, const uint8_t *r, const uint8_t *s, char proof_hex[513]);
/* A second prover object on the same resident key: shares the immutable device tables of `h` (1.8 GB for send), owns its streams and workspaces (about 0.25 GB).
 * Objects may be used from different threads at the same time; their proofs overlap on the device. */
zkgpu_prover *zkgpu_prover_clone(zkgpu_prover *h);
/* n proofs against one resident key in one call (BASELINE.json configs[2]: a batch of independent statements).  zs: n assignments of n_vars 32-byte canonical values each,
 * back to back; rs: n pairs (r, s) of 32-byte canonical values, or NULL for fresh randomness; proofs_hex: n records of 513 bytes.  The proofs are spread over
 * ZK_BATCH_LANES (default 4) prover objects sharing the key's tables, one host thread each, so that the packing of witness i+1 and the latency-bound tails of proof i
 * overlap the kernels of the others.  Every proof is byte-identical to what zkgpu_prover_prove returns for the same (z, r, s).  ZKGPU_ERR_UNSATISFIED if any assignment
 * violates the constraint system (the error text lists which); the other records are still filled. */
int zkgpu_prover_prove_batch(zkgpu_prover *h, const uint8_t *zs, size_t n, const uint8_t *rs, char *proofs_hex);
int zkgpu_prover_timings(zkgpu_prover *h, double out[5]);       /* ms of the last prove(): upload+rows, (unused), device kernels, host finish, total */
/* per-stage device timing with HIP events on the compute stream (for the benchmark's roofline leg).  report: JSON {"stage": {"ms_total": x, "count": n}, ...} */
int zkgpu_profile_enable(int on);
int zkgpu_profile_report(char *buf, size_t cap);
/* 1 = accept, 0 = reject, negative = error.  inputs: n_inputs canonical field elements (the packed public input) */
int zkgpu_verify(const char *vk_path, const char *proof_hex, const uint8_t *inputs, size_t n_inputs);
/* the same decision for n proofs in one GPU launch (kernel K9; r1cs_gg_ppzksnark_verifier_strong_IC, r1cs_gg_ppzksnark.tcc:509-623, one lane per proof).
   proofs_hex: n * 512 characters (no separators); inputs: n * n_inputs canonical field elements of 32 bytes; ok[i] = 1 accept / 0 reject.  Returns ZKGPU_OK or an error */
/* test entry: the GPU verifier's operation schedule interpreted on the host (no device needed); returns 1 accept / 0 reject; stats[8] (optional): rounds, slots, products, linear operations, constants, rounds of products / eight-lane sums / one-lane sums */
int zkgpu_test_verify_schedule(const char *vk_path, const char *proof_hex, const uint8_t *inputs, size_t n_inputs, uint32_t *stats);
/* out[0] = small verification calls (up to 64 proofs) taken by this key's GPU verifier, out[1] = kernel launches made for them: calls that meet — go-ethereum verifies
 * from many goroutines, one proof a call — share a launch */
int zkgpu_verify_counters(const char *vk_path, uint64_t out[2]);
/* test entry (needs a GPU): kernel K9's LDS values after every `every`-th round of its schedule against the host model of the same 29-bit limb arithmetic, on one proof.
 * out[0] = the first round whose values differ or -1, out[1] = the slot, out[2] = the kernel's verdict (1 accept, 0 reject, 2 handed back to the host verifier) */
int zkgpu_test_verify_trace(const char *vk_path, const char *proof_hex, const uint8_t *inputs, size_t n_inputs, uint32_t every, long out[3]);
int zkgpu_verify_batch(const char *vk_path, const char *proofs_hex, const uint8_t *inputs, size_t n_inputs, size_t n, uint8_t *ok);

#ifdef __cplusplus
}
#endif
#endif
