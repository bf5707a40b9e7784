// Host-side interface to the HIP kernels (implemented in gpu.hip).  Plain pointers and sizes only; device buffers are
// owned by the objects declared here.  Everything runs on one HIP stream per GpuContext; results that the host needs
// come back through pinned buffers after a single stream synchronisation.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "hostmath.hpp"
#include "pairing_host.hpp"

namespace zk {

struct GpuError : std::runtime_error { using std::runtime_error::runtime_error; };
struct GpuOutOfMemory : GpuError { using GpuError::GpuError; };   // a device allocation that did not fit (DevBuf): the one failure a caller may answer by asking for less
uint64_t queries_without_tables();   // key queries loaded without their fixed-base tables because the tables did not fit the device (the prover is then several times slower on them)
void note_query_without_tables();

// raw 32-byte field element / point records as they sit in HBM (Montgomery form, little-endian limbs)
struct Fe32 { uint32_t l[8]; };
struct G1AffineRaw { Fe32 x, y; };            // 64 B
struct G2AffineRaw { Fe32 x0, x1, y0, y1; };  // 128 B

class GpuContext;
GpuContext &gpu();                             // lazily initialised process-wide context (device from ZK_DEVICE / LOCAL_RANK)
int gpu_device_numa_node(int device);            // NUMA node of the socket the device hangs off, -1 unknown (sysfs)
bool gpu_available();                          // false if no HIP device is visible
int lane_plan_simulate(int n_slots, int kinds, int per_kind, int *out_lanes_per_slot);   // the lane planner on a private table (CPU tests)
// independent stream sets (gpu.hip); a lane belongs to one device of the list below
int gpu_lane_acquire(int device_slot = 0);
void gpu_lane_release(int lane);
int gpu_lane_current();
void gpu_lane_select(int lane);
// The devices this process works on: ZK_DEVICES = "all" or a comma-separated list of HIP device indices; without it the single device ZK_DEVICE / LOCAL_RANK
// (default 0) as before. Slot k of the list is what device_slot arguments mean. A plain go-ethereum process calling the cgo symbols from many goroutines
// thereby uses every GPU of the node: the key pool holds provers on each device (capi_zk.cpp) and spreads the callers over them.
int gpu_device_slots(); int gpu_slot_of_lane(int lane);
std::vector<int> parse_device_list(const char *spec, int n_visible, int fallback_device);   // pure function (unit-tested on the CPU)
struct LaneScope { int prev; explicit LaneScope(int lane) : prev(gpu_lane_current()) { gpu_lane_select(lane); } ~LaneScope() { gpu_lane_select(prev);
    } LaneScope(const LaneScope &) = delete; };

template <class T> class DevBuf {              // RAII device allocation
 public:
  DevBuf() = default; explicit DevBuf(size_t n); ~DevBuf(); DevBuf(DevBuf &&o) noexcept; DevBuf &operator=(DevBuf &&o) noexcept;
  DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
  T *get() const { return p_; } size_t size() const { return n_; }
  void upload(const T *host, size_t n); void download(T *host, size_t n) const; void zero();
 private:
  T *p_ = nullptr; size_t n_ = 0;
};

template <class T> class PinnedBuf {           // page-locked host staging buffer
 public:
  PinnedBuf() = default; explicit PinnedBuf(size_t n); ~PinnedBuf(); PinnedBuf(const PinnedBuf &) = delete; PinnedBuf &operator=(const PinnedBuf &) = delete;
  PinnedBuf &operator=(PinnedBuf &&o) noexcept { if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; } return *this; }
  T *get() const { return p_; } size_t size() const { return n_; }
 private:
  void release(); T *p_ = nullptr; size_t n_ = 0;
};
void upload_async(void *dev, const void *pinned_host, size_t bytes);   // on the main stream, no synchronisation
void copy_dev_async(void *dst, const void *src, size_t bytes);          // device to device, on the main stream, no synchronisation

// A fixed set of base points resident in HBM (one query of a proving key) plus the reusable MSM workspace for it.
struct WsortBuffers;                           // the sorted witness digits an MSM leaves for the MSMs over the same scalar vector (msm_impl.hpp)
// An assignment that arrived in compact form (Prover::set_witness*), as the witness MSMs' sort (msm.cuh: k_wsort_tagged) wants it: device pointers, valid for
// the run that follows
struct WitnessTags {
  const uint8_t *tags = nullptr;        // one byte per variable (variable 0 = ONE): 0 the value is zero, 1 it is one, 2 anything else
  const uint32_t *other_vars = nullptr; // the variables tagged 2, ascending
  uint32_t n_other = 0;                // the list's length — or, with n_other_dev set, an upper bound of it (it sizes the launch)
  const uint32_t *n_other_dev = nullptr; // the length in device memory, where the list was made on the device (k_classify_witness): no trip to the host
  // indexed queries (B): position of a variable in the query's index list, 0xffffffff if it has no point; null for plain queries
  const uint32_t *var_pos = nullptr;
  uint32_t base = 0;                    // plain query: point i belongs to variable base + i; indexed query: first position of this slice of the index list
};
class MsmG1 {
 public:
  // tables: precompute 2^(cw) P if the size cap allows; uniform: one-pass sort with overflow fallback (msm_impl.hpp)
  MsmG1(const G1AffineRaw *host_points, size_t n, int window_bits, bool filter_ones, bool fixed_base_tables = true, bool uniform_scalars = false);
  // shares the peer's resident points / fixed-base table (immutable); owns only its workspace
  MsmG1(const MsmG1 &peer, bool filter_ones, bool uniform_scalars);
  ~MsmG1();
  // scalars_dev: Fr (Montgomery) on the device.  scalar_index_dev: optional gather map (point i uses scalars[index[i]]).
  // Enqueues the kernels; result() synchronises and finishes the combine on the host.
  void run(const Fe32 *scalars_dev, const uint32_t *scalar_index_dev);
  // the same with the whole assignment z_all (variable 0 = ONE) and its tags: the three-launch witness path then sorts from the tags (other paths read the
  // scalars as run() does)
  void run_tagged(const Fe32 *z_all_dev, const WitnessTags &wt, const uint32_t *scalar_index_dev);
  // scalars a_i * b_i * z (z: one element, or a table of n) formed inside the sort kernel; only when one_pass_sort() (uniform-scalar MSM with fixed-base
  // tables)
  bool one_pass_sort() const; void run_product(const Fe32 *a_dev, const Fe32 *b_dev, const Fe32 *z_dev, bool z_is_table);
  void set_crowded(bool other_proofs_in_flight);   // uniform-scalar path: longer accumulation runs while the chip is shared with other proofs (msm_impl.hpp: h_run_crowded)
  host::HG1 result();
  // MSMs over the same scalar vector (same length, same window) can share one sort of its digits: the follower must be run after the leader, on the leader's
  // stream or on another one (it then waits for the leader's event). false if either side is not on the three-launch witness path.
  std::shared_ptr<WsortBuffers> sort_handle() const; bool share_sort_with(const std::shared_ptr<WsortBuffers> &leader);
  // split: the scalar-one sum runs on a stream of its own beside the bucket path
  size_t size() const;
  const G1AffineRaw *points_dev() const;
  void set_label(const char *l);
  void set_stream(int aux /* -1 main, 0..3 auxiliary */);
  void split_ones_path();
  struct Impl; std::unique_ptr<Impl> impl;
};
class MsmG2 {
 public:
  MsmG2(const G2AffineRaw *host_points, size_t n, int window_bits, bool filter_ones, bool fixed_base_tables = true, bool uniform_scalars = false);
  MsmG2(const MsmG2 &peer, bool filter_ones, bool uniform_scalars);
  ~MsmG2();
  void run(const Fe32 *scalars_dev, const uint32_t *scalar_index_dev);
  void run_tagged(const Fe32 *z_all_dev, const WitnessTags &wt, const uint32_t *scalar_index_dev);
  std::shared_ptr<WsortBuffers> sort_handle() const; bool share_sort_with(const std::shared_ptr<WsortBuffers> &leader);
  host::HG2 result(); void set_label(const char *l); void set_stream(int aux); void split_ones_path();
  struct Impl; std::unique_ptr<Impl> impl;
};

// Batched Groth16 verification on the GPU (kernel K9, pairing.cuh): one lane per proof, the key-dependent tables and the verification program resident in HBM.
class BatchVerifier {
 public:
  // all Montgomery, affine
  BatchVerifier(const host::HFq12 &alpha_g1_beta_g2, const G2AffineRaw &gamma_g2, const G2AffineRaw &delta_g2, const G1AffineRaw *ic, size_t n_ic);
  ~BatchVerifier();
  size_t num_inputs() const; size_t program_length() const;
  void counters(uint64_t out[2]) const;   // small calls taken / launches made for them (calls that meet share a launch)
  // proofs_mont: n records of 256 bytes (A.x A.y | B.x.c0 B.x.c1 B.y.c0 B.y.c1 | C.x C.y, Montgomery); inputs: n * num_inputs() canonical field elements; ok[i]
  // = 1 accept / 0 reject
  void verify(const void *proofs_mont, const Fe32 *inputs_canonical, size_t n, uint8_t *ok);
  uint8_t trace(const void *proof_mont, const Fe32 *inputs_canonical, uint32_t every, std::vector<uint32_t> &values, uint8_t nacc_out[96]);   // tests (gpu_verify.hip)
  struct Impl; std::unique_ptr<Impl> impl;
};

// Evaluation domain of size m = 2^k or 2^k + 2^r (libfqfft get_evaluation_domain, get_evaluation_domain.tcc:33-52) with
// its twiddle / coset tables resident in HBM.
struct R1csHost;
class Domain {
 public:
  explicit Domain(size_t min_size); explicit Domain(const Domain &peer); ~Domain();   // the copy shares the twiddle / coset tables and owns its scratch space
  size_t m() const; bool is_step() const;
  // in place on `batch` device vectors of m elements each, `stride` elements apart (Montgomery form)
  void fft(Fe32 *data, int batch, size_t stride); void ifft(Fe32 *data, int batch, size_t stride);
  void coset_fft(Fe32 *data, int batch, size_t stride); void icoset_fft(Fe32 *data, int batch, size_t stride);
  // = ifft(); coset_fft(); on a step domain the passes between the two transforms are one kernel
  void ifft_then_coset_fft(Fe32 *data, int batch, size_t stride);
 private: void fft_with_factors(Fe32 *data, int batch, size_t stride, const Fe32 *factors); public:
  // key load: the H query (n_in = m - 1 affine points) re-expressed so that sum_j v_j out_j = sum_i icosetFFT(v)_i h_i: the prover then skips the last
  // transform (ecntt.cuh)
  bool supports_h_lagrange() const; void h_query_to_coset_lagrange(const G1AffineRaw *h, size_t n_in, G1AffineRaw *out /* m points */);
  // key load, radix-2 domains: out (n_vars + 1 points) = the L query extended to all variables minus the C polynomial's share of the H term; the prover then
  // transforms A and B only (ecntt.cuh)
  bool supports_c_fold() const;
  void fold_c_into_l(const G1AffineRaw *h_lagrange /* m */, const R1csHost &cs, const G1AffineRaw *L /* n_vars - n_inputs */, G1AffineRaw *out);
  // a = (a*b - c) / Z on the coset (c may be null: a = a*b / Z); zinv_dev(): 1/Z on the coset, one element or (step domains) a table of m
  const Fe32 *zinv_dev() const; bool zinv_is_table() const;
  void qap_pointwise(Fe32 *a, const Fe32 *b, const Fe32 *c);
  struct Impl; std::unique_ptr<Impl> impl;
};

// R1CS in CSR form resident in HBM (three matrices, coefficient table) — kernel K1
struct R1csHost {                               // as parsed from a key file or emitted by a circuit
  size_t n_inputs = 0, n_vars = 0, n_cons = 0;  // n_vars excludes ONE
  std::vector<uint32_t> rowptr[3], col[3]; std::vector<Fe32> coeff[3];   // coeff canonical (non-Montgomery), col 0 = ONE
};
class R1csDev {
 public:
  explicit R1csDev(const R1csHost &h); explicit R1csDev(const R1csDev &peer); ~R1csDev();   // the copy shares the CSR arrays and owns its failure word
  // z_dev: n_vars+1 Fr (Montgomery, z[0] = 1).  abc: 3 vectors of m (zero padded, aA[n_cons + i] = z_i for i <= n_inputs; r1cs_to_qap.tcc:227-230)
  // tags: one byte per variable (0 / 1 / 2 = other) when the assignment came in compact form; write_c = false: the C vector is not stored (it is folded into
  // the L query)
  void eval(const Fe32 *z_dev, Fe32 *abc, size_t m, const uint8_t *tags_dev = nullptr, bool write_c = true);
  bool satisfied(const Fe32 *abc, size_t m);    // synchronises
  // eval() also tests a*b == c row by row; true if the last eval() found every constraint satisfied (read after the main stream has been synchronised)
  bool check_result() const;
  uint32_t failed_row() const;   // a constraint the last eval() found violated (while check_result() is false)
  struct Impl; std::unique_ptr<Impl> impl;
};

void fr_to_mont_dev(Fe32 *a, size_t n); void fr_from_mont_dev(Fe32 *a, size_t n);
// compact assignment upload (ntt.cuh: k_expand_witness). packed = [ones bitmap | other bitmap | (canon bitmap, canon == 2) | block offsets | values at
// expand_values_offset()]; canon: 0 all values in Montgomery form, 1 all canonical, 2 the third bitmap says which are canonical. tags_out / other_vars_out: see
// WitnessTags
inline size_t expand_values_offset(size_t words, int canon) { return (((canon == 2 ? 28 : 20) * words + 31) / 32) * 32; }
// compact assignment upload (ntt.cuh: k_expand_witness)
void expand_witness_dev(const uint8_t *packed_dev, size_t words, const Fe32 &one_value, int canon, size_t n, Fe32 *out, uint8_t *tags_out = nullptr,
    uint32_t *other_vars_out = nullptr);
// the same tags and list from an assignment that already lies in device memory (ntt.cuh: k_classify_witness); counters = two words, alternating by parity
// a circuit board's hand-over (ntt.cuh: k_expand_board): tag bytes and the candidates' values already on the device -> z, tags, the list of other values
// (vals_by_var: cand_vals is the board's whole wide array — entry = variable number — instead of the gathered list)
void expand_board_dev(const uint8_t *board_tags, size_t n, const uint32_t *cand, const Fe32 *cand_vals, size_t n_cand, Fe32 *z, uint8_t *tags, uint32_t *other_vars,
    uint32_t *counters, int parity, bool vals_by_var = false);
// groups of variables with equal columns folded into one place each (ntt.cuh: k_merge_equal_columns); tags may be null; on the main stream
void merge_equal_columns_dev(Fe32 *z, uint8_t *tags, const uint32_t *grp_ptr, const uint32_t *grp_mem, size_t n_groups);
void classify_witness_dev(const Fe32 *z, size_t n, uint8_t *tags_out, uint32_t *other_vars_out, uint32_t *counters, int parity);

// Key loading: y-coordinates of compressed points (x Montgomery; flags bit0 = parity of canonical y, bit1 = point at infinity). Throws if an x is not on the
// curve.
void decompress_g1(const Fe32 *xs, const uint8_t *flags, size_t n, G1AffineRaw *out);
void decompress_g2(const Fe32 *xs /* 2 per point */, const uint8_t *flags, size_t n, G2AffineRaw *out);
// Key generation: out[i] = scalars[i] * base (scalars canonical), results affine Montgomery
void fixed_base_mul_g1(const host::HG1 &base, const Fe32 *scalars, size_t n, G1AffineRaw *out);
void fixed_base_mul_g2(const host::HG2 &base, const Fe32 *scalars, size_t n, G2AffineRaw *out);
// host memory the device reads in place: pins the pages of [p, p + bytes) and returns the device's address of p, or null if that is not possible (callers then stage);
// gpu_host_unregister(p) undoes it
void *gpu_host_register(void *p, size_t bytes);
void gpu_host_unregister(void *p);
void gpu_sync();            // all streams
void gpu_fork_aux();        // auxiliary streams wait for everything queued on the main stream so far
void gpu_fork_one(int aux); // the same for one auxiliary stream
// the two halves of a fork: record the point on the main stream once, let each auxiliary stream wait for it (from any thread)
void gpu_fork_record();
void gpu_fork_wait(int aux);
void gpu_join_aux();        // the main stream waits for everything queued on the auxiliary streams
bool profiling_enabled();
uint64_t general_path_repeats();   // how often a fast MSM path raised its flag and the MSM was repeated on the general path, process-wide (tests, soak runs)
void note_general_path_repeat();
// per-stage device timing (HIP events on the compute stream); report = JSON object {stage: {ms_total, count}}
void profile_enable(bool on); std::string profile_report();

}  // namespace zk
