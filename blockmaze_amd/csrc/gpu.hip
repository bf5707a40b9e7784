// HIP side of the prover: owns the device, the stream, and the kernel launches declared in gpu.hpp.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
#include <algorithm>
#include <mutex>
#include <atomic>
#include <mutex>
#include "gpu_internal.hpp"
#include "ntt.cuh"
#include "ecntt.cuh"

// Every prover runs five HIP streams at once (the critical chain + four witness MSMs). The runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default
// 4), and two streams sharing a queue run one after the other (measured: the B1 MSM then ends at 2.7 ms instead of 1.4 ms). The runtime reads the variable when
// its first API call initialises it, so it has to be in the environment before ANY HIP call of the process: a load-time constructor with the earliest user
// priority does that — it runs when the dynamic loader maps libzkgpu.so (program start for a cgo binary linked against libzk_*.so), before this library's own
// code-object registration and long before gpu_available(). A host that wants another value exports the variable itself (it is not overwritten). A host that
// dlopen()s the library late — after it has started threads that call getenv, or after its own first HIP call — must export GPU_MAX_HW_QUEUES itself before
// that call: setenv is not thread safe and comes too late then.
__attribute__((constructor(101))) static void zkgpu_load_time_environment() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

namespace zk {

// Lanes: independent sets of streams (one main + four auxiliary) so that several provers can have a proof in flight at the same time; a thread works on the
// lane it selected with LaneScope (lane 0 unless told otherwise). Contexts are created on first use and live for the life of the process. (lanes beyond the
// hardware queues share queues) Sized for 8 devices x 4 circuit kinds x 7 pool members with a lane each; a single device never binds more than 31.
constexpr int MAX_LANES_PER_DEVICE = 31, MAX_LANES = 1 + 8 * 28;
static uint8_t g_lane_claimed[MAX_LANES];   // a lane that was ever lent keeps its device slot (guarded by g_lane_mutex)
static std::atomic<GpuContext *> g_lanes[MAX_LANES];
static std::atomic<int> g_lane_slot[MAX_LANES];
static std::mutex g_lane_mutex;
static thread_local int t_lane = 0;
static std::atomic<unsigned> g_next_lane{0};
std::vector<int> parse_device_list(const char *spec, int n_visible, int fallback_device) {
  std::vector<int> out; if (n_visible <= 0) return out;
  if (!spec || !*spec) { out.push_back(((fallback_device % n_visible) + n_visible) % n_visible); return out; }
  if (!strcmp(spec, "all")) { for (int i = 0; i < n_visible; i++) out.push_back(i); return out; }
  for (const char *p = spec; *p;) {
    char *end = nullptr;
    long v = strtol(p, &end, 10);
    if (end == p) break;
    if (v >= 0 && v < n_visible) {
      bool dup = false;
      for (int x : out) dup |= x == (int)v;
      if (!dup) out.push_back((int)v);
    }
    p = *end == ',' ? end + 1 : end;
    if (*end && *end != ',') break;
  }
  if (out.empty()) out.push_back(0);
  return out; }
static const std::vector<int> &device_list() {
  static const std::vector<int> l = [] {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    const char *e = getenv("ZK_DEVICE");
    if (!e) e = getenv("LOCAL_RANK");
    return parse_device_list(getenv("ZK_DEVICES"), n, e ? atoi(e) : 0); }(); return l; }
int gpu_device_slots() { return (int)device_list().size(); }
int gpu_slot_of_lane(int lane) { return lane < 0 || lane >= MAX_LANES ? 0 : g_lane_slot[lane].load(); }
GpuContext &gpu() {
  GpuContext *c = g_lanes[t_lane].load(std::memory_order_acquire);
  if (!c) { std::lock_guard<std::mutex> lk(g_lane_mutex); c = g_lanes[t_lane].load(std::memory_order_acquire);
    if (!c) {
      const std::vector<int> &l = device_list();
      if (l.empty()) throw GpuError("no HIP device visible: the prover's HIP path cannot run (there is no CPU fallback)");
      c = new GpuContext(l[(size_t)g_lane_slot[t_lane].load() % l.size()]); g_lanes[t_lane].store(c, std::memory_order_release); } }
  hipSetDevice(c->device); return *c;
}
// Lanes 1.. are lent to provers; lane 0 (device slot 0) stays with everything else. A lane belongs to the device slot of its first user for good (its streams
// live on that device). A lane is lent to ONE prover at a time and handed back by its destructor (gpu_lane_release): a process that reloads keys or clones
// provers for ever keeps cycling through the same stream sets instead of running out of them.
// Every device slot may bind at most lane_quota(D) lanes, so that D devices can never starve one another whatever the order in which their pools are built (the
// cgo layer builds them lazily, one device at a time, every pool member of every circuit kind taking a lane); past its quota a slot shares its least-used lane
// — provers on one lane share its streams: still correct, merely serialised.
static int g_lane_users[MAX_LANES];
int lane_quota(int n_slots) { const int q = (MAX_LANES - 1) / std::max(1, n_slots); return std::max(1, std::min(q, MAX_LANES_PER_DEVICE)); }
// The planner, free of HIP state so that a CPU test can drive it (zkgpu_test_lane_plan): users[l] = provers holding lane l, slot[l] = its device slot, bound[l]
// = whether the lane belongs to a slot yet. Returns the lane to use (and binds / counts it), or -1.
int lane_plan_pick(int *users, int *slot, uint8_t *bound, int device_slot, int n_slots) {
  int mine = 0;
  for (int lane = 1; lane < MAX_LANES; lane++) if (bound[lane] && slot[lane] == device_slot) mine++;
  for (int lane = 1; lane < MAX_LANES; lane++)                                           // a free lane of this slot: its context exists
    if (bound[lane] && slot[lane] == device_slot && users[lane] == 0) { users[lane] = 1; return lane; }
  if (mine < lane_quota(n_slots))
    for (int lane = 1; lane < MAX_LANES; lane++)                                         // a lane nobody owns yet
      if (!bound[lane] && users[lane] == 0) { users[lane] = 1; slot[lane] = device_slot; bound[lane] = 1; return lane; }
  int best = -1;
  for (int lane = 1; lane < MAX_LANES; lane++)                                           // quota reached (or nothing left): share this slot's least-used lane
    if (bound[lane] && slot[lane] == device_slot && (best < 0 || users[lane] < users[best])) best = lane;
  if (best >= 0) users[best]++;
  return best;
}
int lane_plan_simulate(int n_slots, int kinds, int per_kind, int *out_lanes_per_slot) {
  if (n_slots < 1 || n_slots > 64) return -1;
  std::vector<int> users(MAX_LANES, 0), slot(MAX_LANES, 0); std::vector<uint8_t> bound(MAX_LANES, 0); int worst = 0;
  // lazily, one device at a time: the order that used to starve the later devices
  for (int d = 0; d < n_slots; d++)
    for (int k = 0; k < kinds * per_kind; k++) {
      const int lane = lane_plan_pick(users.data(), slot.data(), bound.data(), d, n_slots);
      if (lane < 0) return -1;
      worst = std::max(worst, users[lane]);
    }
  for (int d = 0; d < n_slots; d++) {
    int c = 0;
    for (int lane = 1; lane < MAX_LANES; lane++) c += bound[lane] && slot[lane] == d;
    if (out_lanes_per_slot) out_lanes_per_slot[d] = c;
  }
  return worst;
}
int gpu_lane_acquire(int device_slot) {
  std::lock_guard<std::mutex> lk(g_lane_mutex);
  int slot[MAX_LANES]; uint8_t bound[MAX_LANES];
  for (int lane = 0; lane < MAX_LANES; lane++) {
    slot[lane] = g_lane_slot[lane].load();
    bound[lane] = g_lanes[lane].load() != nullptr || g_lane_users[lane] > 0 || g_lane_claimed[lane];
  }
  const int lane = lane_plan_pick(g_lane_users, slot, bound, device_slot, gpu_device_slots());
  if (lane < 0) throw GpuError("no stream lane left for device slot " + std::to_string(device_slot));
  g_lane_slot[lane].store(device_slot); g_lane_claimed[lane] = 1; return lane; }
void gpu_lane_release(int lane) {
  if (lane <= 0 || lane >= MAX_LANES) return;
  std::lock_guard<std::mutex> lk(g_lane_mutex);
  if (g_lane_users[lane] > 0) g_lane_users[lane]--;
}
int gpu_lane_current() { return t_lane; }
void gpu_lane_select(int lane) { t_lane = lane < 0 || lane >= MAX_LANES ? 0 : lane; }
bool gpu_available() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess && n > 0; }
// NUMA node of the host socket visible device `device` hangs off (sysfs entry of its PCI function), -1 unknown. A rank launcher binds its process to that
// node's CPUs (bench.py); the library itself only moves its scan threads next to the caller's buffer (hostnuma.hpp).
int gpu_device_numa_node(int device) {
  char id[64] = {0};
  if (hipDeviceGetPCIBusId(id, (int)sizeof(id), device) != hipSuccess) return -1;
  for (char *c = id; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');      // sysfs spells the address in lower case
  char path[160];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", id);
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}
void gpu_sync() { HIP_CHECK(hipStreamSynchronize(gpu().stream)); for (int i = 0; i < 4; i++) HIP_CHECK(hipStreamSynchronize(gpu().aux[i])); }
void gpu_join_aux() {
  GpuContext &g = gpu();
  for (int i = 0; i < 4; i++) {
    HIP_CHECK(hipEventRecord(g.join_event[i], g.aux[i]));
    HIP_CHECK(hipStreamWaitEvent(g.stream, g.join_event[i], 0));
  }
}
bool profiling_enabled();
void gpu_fork_record() { GpuContext &g = gpu(); HIP_CHECK(hipEventRecord(g.fork_event, g.stream)); }
// may be called from the thread that submits to that stream
void gpu_fork_wait(int i) {
  GpuContext &g = gpu();
  HIP_CHECK(hipStreamWaitEvent(g.aux[i & 3], g.fork_event, 0));
}
// (the stream's join event is free at this point of a proof)
void gpu_fork_one(int i) {
  GpuContext &g = gpu();
  HIP_CHECK(hipEventRecord(g.join_event[i & 3], g.stream));
  HIP_CHECK(hipStreamWaitEvent(g.aux[i & 3], g.join_event[i & 3], 0));
}
void gpu_fork_aux() {
  GpuContext &g = gpu();
  HIP_CHECK(hipEventRecord(g.fork_event, g.stream));
  for (int i = 0; i < 4; i++) HIP_CHECK(hipStreamWaitEvent(g.aux[i], g.fork_event, 0));
}
hipStream_t gpu_stream() { return gpu().stream; }

// ---- optional per-stage timing with HIP events on the compute stream (bench.py's roofline leg; off by default) ----------
struct StageTimer {
  struct Span { std::string name; hipEvent_t a, b; };
  bool enabled = false; std::mutex mu;   // (stages are opened from the prover's submit threads as well)
  std::vector<Span> open; std::vector<hipEvent_t> pool; std::map<std::string, std::pair<double, long>> acc;
  hipEvent_t get() { if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; } hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e; }
  struct Open { hipStream_t st; };
  std::vector<hipStream_t> open_streams;
  size_t begin(const char *name, hipStream_t st) {
    if (!enabled) return (size_t)-1;
    std::lock_guard<std::mutex> lk(mu);
    Span s{name, get(), get()};
    HIP_CHECK(hipEventRecord(s.a, st));
    open.push_back(s);
    open_streams.push_back(st);
    return open.size() - 1;
  }
  void end(size_t id) { if (id == (size_t)-1) return; std::lock_guard<std::mutex> lk(mu); HIP_CHECK(hipEventRecord(open[id].b, open_streams[id])); }
  void collect() { std::lock_guard<std::mutex> lk(mu); if (open.empty()) return; HIP_CHECK(hipDeviceSynchronize());
    for (Span &s : open) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
        auto &e = acc[s.name];
        e.first += ms;
        e.second++;
      }
      pool.push_back(s.a);
      pool.push_back(s.b);
    }
    open.clear();
    open_streams.clear();
  }
};
static StageTimer g_timer;
Stage::Stage(const char *n, hipStream_t st) : id(g_timer.begin(n, st ? st : gpu().stream)) {}
Stage::~Stage() { g_timer.end(id); }
bool profiling_enabled() { return g_timer.enabled; }
void profile_enable(bool on) { g_timer.collect(); g_timer.enabled = on; g_timer.acc.clear(); }
std::string profile_report() { g_timer.collect(); std::string o = "{"; bool first = true;
  for (auto &kv : g_timer.acc) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s\"%s\": {\"ms_total\": %.6f, \"count\": %ld}", first ? "" : ", ", kv.first.c_str(), kv.second.first, kv.second.second);
    o += buf;
    first = false;
  }
  return o + "}";
}

static bool alloc_poison() { static const bool on = [] { const char *e = getenv("ZK_DEBUG_POISON_ALLOC"); return e && *e && *e != '0'; }(); return on; }
template <class T> DevBuf<T>::DevBuf(size_t n) : n_(n) {
  gpu();
  if (n) {
    hipError_t e = hipMalloc((void **)&p_, n * sizeof(T));
    if (e != hipSuccess) {
      p_ = nullptr;
      if (e == hipErrorOutOfMemory) throw GpuOutOfMemory("hipMalloc of " + std::to_string(n * sizeof(T)) + " bytes: " + hipGetErrorString(e));
      throw GpuError("hipMalloc of " + std::to_string(n * sizeof(T)) + " bytes: " + hipGetErrorString(e));
    }
    // (diagnostic: ZK_DEBUG_POISON_ALLOC=1 fills every new allocation with 0xA5 bytes — whether fresh device memory reads as zeros depends on the host's driver, so
    // a kernel that reads a word nobody wrote can pass on one box and fail on the next; with the pattern it fails everywhere.  tests/test_gpu_groth16.py runs with it)
    if (alloc_poison()) { if (hipMemset(p_, 0xA5, n * sizeof(T)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) throw GpuError("hipMemset (ZK_DEBUG_POISON_ALLOC)"); }
  }
}
template <class T> DevBuf<T>::~DevBuf() { if (p_) hipFree(p_); }
template <class T> DevBuf<T>::DevBuf(DevBuf &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
template <class T> DevBuf<T> &DevBuf<T>::operator=(DevBuf &&o) noexcept {
  if (this != &o) {
    if (p_) hipFree(p_);
    p_ = o.p_;
    n_ = o.n_;
    o.p_ = nullptr;
    o.n_ = 0;
  }
  return *this;
}
template <class T> void DevBuf<T>::upload(const T *h, size_t n) {
  HIP_CHECK(hipMemcpyAsync(p_, h, n * sizeof(T), hipMemcpyHostToDevice, gpu().stream));
  HIP_CHECK(hipStreamSynchronize(gpu().stream));
}
template <class T> void DevBuf<T>::download(T *h, size_t n) const {
  HIP_CHECK(hipMemcpyAsync(h, p_, n * sizeof(T), hipMemcpyDeviceToHost, gpu().stream));
  HIP_CHECK(hipStreamSynchronize(gpu().stream));
}
template <class T> void DevBuf<T>::zero() { if (n_) HIP_CHECK(hipMemsetAsync(p_, 0, n_ * sizeof(T), gpu().stream)); }
template class DevBuf<uint8_t>;
template class DevBuf<uint32_t>;
template class DevBuf<Fe32>;
template class DevBuf<G1AffineRaw>;
template class DevBuf<G2AffineRaw>;

template <class T> PinnedBuf<T>::PinnedBuf(size_t n) : n_(n) { gpu(); if (n) { HIP_CHECK(hipHostMalloc((void **)&p_, n * sizeof(T))); if (alloc_poison()) memset(p_, 0xA5, n * sizeof(T)); } }
template <class T> PinnedBuf<T>::~PinnedBuf() { release(); }
template <class T> void PinnedBuf<T>::release() { if (p_) hipHostFree(p_); p_ = nullptr; n_ = 0; }
template class PinnedBuf<Fe32>;
void upload_async(void *dev, const void *host, size_t bytes) { HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, gpu().stream)); }
// device to device by a kernel of this library: in order with the kernels around it on the compute queue — no copy engine, no choice the runtime makes per size
// (what keeps a statement in HBM, Prover::stash_witness, is made of this and nothing else)
static __global__ void k_copy_words16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void copy_dev_async(void *dst, const void *src, size_t bytes) {
  if (!bytes) return;
  if (bytes % 16 || (reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) % 16) { HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, gpu().stream)); return; }
  const size_t n16 = bytes / 16; hipLaunchKernelGGL(k_copy_words16, dim3((unsigned)std::min<size_t>(cdiv(n16, 256), 4096)), dim3(256), 0, gpu().stream, (const uint4 *)src, (uint4 *)dst, n16);
}


// ======================================================================================================================
// Evaluation domains
// ======================================================================================================================
static size_t ceil_log2(size_t n) { size_t r = ((n & (n - 1)) == 0 ? 0 : 1); while (n > 1) { n >>= 1; r++; } return r; }   // FF/common/utils.cpp:32-45
using host::HFr;
// field_utils.tcc:36-51
static HFr fr_root_of_unity(size_t n) {
  HFr w;
  memcpy(w.l, FR_ROOT_OF_UNITY_2_28, 32);
  for (size_t i = 28; i > ceil_log2(n); --i) w = w.sqr();
  return w;
}
static HFr fr_coset_gen() { HFr g; memcpy(g.l, FR_COSET_GEN, 32); return g; }

// tables for one power-of-two transform size n with root w: tw[j] = w^j, itw[j] = w^-j (j < n/2)
// tw261 / itw261: the same powers times 2^261 instead of 2^256 (canonical integers, 8 words): the factor form of the tile kernels, which compute on 29-bit
// limbs (ntt.cuh)
struct Radix2Tables {
  int logn; size_t n; DevBuf<Fe32> tw, itw, tw261, itw261;
  Radix2Tables(size_t n_, const HFr &w) : logn((int)ceil_log2(n_)), n(n_), tw(n_ / 2 ? n_ / 2 : 1), itw(n_ / 2 ? n_ / 2 : 1), tw261(n_ / 2 ? n_ / 2 : 1),
      itw261(n_ / 2 ? n_ / 2 : 1) {
    std::vector<Fe32> a(n / 2 ? n / 2 : 1), b(a.size()), a5(a.size()), b5(a.size()); HFr wi = w.inv(), x = HFr::one(), y = HFr::one();
    // the Montgomery integer of v is v 2^256: five doublings give v 2^261 (mod r)
    auto times32 = [](HFr v) {
      for (int i = 0; i < 5; i++) v = v + v;
      return v;
    };
    for (size_t j = 0; j < std::max<size_t>(n / 2, 1); j++) {
      memcpy(&a[j], x.l, 32);
      memcpy(&b[j], y.l, 32);
      HFr x5 = times32(x), y5 = times32(y);
      memcpy(&a5[j], x5.l, 32);
      memcpy(&b5[j], y5.l, 32);
      x = x * w;
      y = y * wi;
    }
    tw.upload(a.data(), a.size()); itw.upload(b.data(), b.size()); tw261.upload(a5.data(), a5.size()); itw261.upload(b5.data(), b5.size());
  }
};
static std::vector<Fe32> geometric_table(size_t n, const HFr &first, const HFr &ratio) {
  std::vector<Fe32> t(n);
  HFr x = first;
  for (size_t i = 0; i < n; i++) {
    memcpy(&t[i], x.l, 32);
    x = x * ratio;
  }
  return t;
}

// in-place radix-2 transform of `batch` vectors: data = post * NTT(pre * data), natural order in and out.  Up to 2^22 points: two LDS-tiled passes
// (k_ntt_cols: data -> scratch, k_ntt_rows: scratch -> data; one pass in place when the whole vector fits a tile); beyond that the stage-per-launch path.
// two columns per tile since the tiles compute on 29-bit limbs (1.01 against 1.04 ms per send proof; one column was best for the 32-bit passes)
static int ntt_pref_log_c() {
  static const int v = [] {
    const char *e = getenv("ZK_NTT_LOGC");
    int x = e ? atoi(e) : 1;
    return x < 0 ? 0 : x > 3 ? 3 : x; }(); return v; }
// f 2^261 mod r as a canonical integer (the host type holds f 2^256: five doublings)
static Fe32 fe261(HFr v) {
  for (int i = 0; i < 5; i++) v = v + v;
  Fe32 o;
  memcpy(&o, v.l, 32);
  return o;
}
// One in-place radix-2 transform of `batch` vectors: data = post * NTT(pre * data) * scale, natural order in and out. The tile kernels take the per-element
// factor as pre261 (f 2^261) and the constant as scale261; pre_scale (f 2^256) is the same table for the stage-per-launch path beyond 2^22 points, which folds
// a constant scale into it.
struct NttCall { Fe32 *data, *scratch; const Fe32 *tw, *tw261; int logn; const Fe32 *pre_scale, *pre261; Fe32 scale261; const Fe32 *post_scale;
    size_t stride, scratch_stride; const Fe32 *out261 = nullptr; };   // out261: two-pass range only (NttJob::out261)
// radix-4 passes measured best with two vectors per launch (twice the waves of radix 8: the passes are latency bound), radix 8 with three
static int ntt_prio_bits() { static const int v = (int)(zk_prio_bits(ZKP_NTT) >> 24) << 8; return v; }
static int ntt_radix_log() {
  static const int rl = [] {
    const char *e = getenv("ZK_NTT_RADIX_LOG");
    int x = e ? atoi(e) : 2;
    return x < 1 ? 1 : x > 3 ? 3 : x;
  }();
  return rl;
}
static bool ntt_two_pass(int logn) { return logn > NTT_TILE_LOG && logn <= 2 * NTT_TILE_LOG; }
// one butterfly group per thread and pass, 64..256 threads
static unsigned ntt_threads_for(int logN, int logC) {
  int g = logN + logC - ntt_radix_log();
  return 1u << (g < 6 ? 6 : g > 8 ? 8 : g);
}
// padded tile (ntt_pad) + twiddle table, 36 bytes an element
static size_t ntt_lds_for(int logN, int logC) {
  size_t e = (size_t)1 << (logN + logC);
  return sizeof(Fr29) * (e + (e >> 4) + 1) + (sizeof(Fr29) << logN) / 2;
}
static void ntt_raise_lds() {
  static std::atomic<uint64_t> cols_done{0}, rows_done{0};
  zk_raise_dynamic_lds((const void *)k_ntt_cols, 128 * 1024, cols_done); zk_raise_dynamic_lds((const void *)k_ntt_rows, 128 * 1024, rows_done);
}
// up to two transforms of the two-pass range (2^12 .. 2^22 points) in ONE column launch and ONE row launch (k_ntt_cols: data -> scratch, k_ntt_rows: scratch ->
// data)
static void ntt_two_pass_launch(const NttCall *calls, int n_calls, int batch) {
  hipStream_t s = gpu().stream;
  ntt_raise_lds();
  NttJob cj[2], rj[2];
  memset(cj, 0, sizeof cj);
  memset(rj, 0, sizeof rj);
  unsigned tc = 64, tr = 64;
  size_t lc = 0, lr = 0;
  for (int k = 0; k < n_calls; k++) {
    const NttCall &c = calls[k];
    const int l1 = c.logn / 2, l2 = c.logn - l1, c1 = std::min(ntt_pref_log_c(), std::min(NTT_TILE_LOG - l1, l2)), c2 = std::min(ntt_pref_log_c(),
        std::min(NTT_TILE_LOG - l2, l1));
    Fr sc; memcpy(&sc, &c.scale261, 32);
    cj[k] = NttJob{(const Fr *)c.data, (Fr *)c.scratch, (const Fr *)c.pre261, (const Fr *)c.tw261, sc, c.logn, l1, c1, 1u << (l2 - c1), c.stride,
        c.scratch_stride, nullptr};
    rj[k] = NttJob{(const Fr *)c.scratch, (Fr *)c.data, (const Fr *)c.post_scale, (const Fr *)c.tw261, sc, c.logn, l1, c2, 1u << (l1 - c2), c.scratch_stride,
        c.stride, (const Fr *)c.out261};
    tc = std::max(tc, ntt_threads_for(l1, c1));
    tr = std::max(tr, ntt_threads_for(l2, c2));
    lc = std::max(lc, ntt_lds_for(l1, c1));
    lr = std::max(lr, ntt_lds_for(l2, c2));
  }
  hipLaunchKernelGGL(k_ntt_cols, dim3(cj[0].tiles + cj[1].tiles, batch), dim3(tc), lc, s, cj[0], cj[1], ntt_radix_log() | ntt_prio_bits());
  hipLaunchKernelGGL(k_ntt_rows, dim3(rj[0].tiles + rj[1].tiles, batch), dim3(tr), lr, s, rj[0], rj[1], ntt_radix_log() | ntt_prio_bits());
}
static void radix2_transform(const NttCall &c, int batch) {
  hipStream_t s = gpu().stream;
  size_t n = (size_t)1 << c.logn;
  Fe32 *data = c.data, *scratch = c.scratch;
  const int logn = c.logn;
  const size_t stride = c.stride, scratch_stride = c.scratch_stride;
  if (logn <= NTT_TILE_LOG) {          // n2 = 1: the column pass alone is the whole transform
    if (c.post_scale) throw GpuError("ntt: post scale on a single-pass transform");
    ntt_raise_lds();
    Fr sc;
    memcpy(&sc, &c.scale261, 32);
    if (c.out261) throw GpuError("ntt: output factors on a single-pass transform");
    NttJob j{(const Fr *)data, (Fr *)data, (const Fr *)c.pre261, (const Fr *)c.tw261, sc, logn, logn, 0, 1u, stride, stride, nullptr}, none;
    memset(&none, 0, sizeof none);
    hipLaunchKernelGGL(k_ntt_cols, dim3(1, batch), dim3(ntt_threads_for(logn, 0)), ntt_lds_for(logn, 0), s, j, none, ntt_radix_log());
    return;
  }
  if (ntt_two_pass(logn)) { ntt_two_pass_launch(&c, 1, batch); return; }
  if (c.out261) throw GpuError("ntt: output factors outside the two-pass range");
  hipLaunchKernelGGL(k_ntt_bitrev_scale, dim3(cdiv(n, 256), batch), dim3(256), 0, s, (const Fr *)data, (Fr *)scratch, (const Fr *)c.pre_scale, logn, stride,
      scratch_stride);
  int L = NTT_LOCAL_LOG;
  hipLaunchKernelGGL(k_ntt_local, dim3((unsigned)(n >> L), batch), dim3(NTT_LOCAL_THREADS), sizeof(Fr) << L, s, (Fr *)scratch, (const Fr *)c.tw, logn, L,
      scratch_stride);
  for (int st = L + 1; st <= logn; st++) hipLaunchKernelGGL(k_ntt_stage, dim3(cdiv(n / 2, 256), batch), dim3(256), 0, s, (Fr *)scratch, (const Fr *)c.tw, logn,
      st, scratch_stride);
  for (int b = 0; b < batch; b++) HIP_CHECK(hipMemcpyAsync(data + b * stride, scratch + b * scratch_stride, n * sizeof(Fe32), hipMemcpyDeviceToDevice, s));
  if (c.post_scale) hipLaunchKernelGGL(k_fr_mul_table, dim3(cdiv(n, 256), batch), dim3(256), 0, s, (Fr *)data, (const Fr *)c.post_scale, (uint32_t)n, stride);
}
// the two transforms of a step-radix-2 domain: one pair of launches when both are in the two-pass range, one after the other otherwise
static void radix2_transform_pair(const NttCall &a, const NttCall &b, int batch) {
  if (ntt_two_pass(a.logn) && ntt_two_pass(b.logn)) { NttCall both[2] = {a, b}; ntt_two_pass_launch(both, 2, batch); return; }
  radix2_transform(a, batch); radix2_transform(b, batch);
}

// ---- step-radix-2 helper kernels (domains/step_radix2_domain.tcc:39-153) ---------------------------------------------
// forward pre-pass: c[i] = a[i] + a[i+B] (i<S) else a[i];  d[i] = w^i * (a[i] - a[i+B] (i<S) else a[i]);  e[i] = sum_j d[i + j*S]
// (in place: c overwrites a[0..B); blockIdx.y = vector of the batch)
// (cf, optional: the coset factors g^i of cosetFFT, multiplied in on the way — one pass over the vector and one launch less than a separate table
// multiplication)
__global__ void k_step_fwd_pre(Fr *a_all, Fr *__restrict__ dbuf_all, const Fr *__restrict__ wpow, const Fr *__restrict__ cf, uint32_t B, uint32_t S,
    size_t stride) {
  zk_take_prio(S);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  Fr *a = a_all + blockIdx.y * stride, *dbuf = dbuf_all + (size_t)blockIdx.y * B;
  Fr x = a[i];
  if (cf) x = x * cf[i];
  if (i < S) { Fr y = a[i + B]; if (cf) y = y * cf[i + B]; a[i] = x + y; dbuf[i] = wpow[i] * (x - y); } else { if (cf) a[i] = x; dbuf[i] = wpow[i] * x; }
}
__global__ void k_step_fold(const Fr *__restrict__ dbuf_all, Fr *__restrict__ a_all, uint32_t B, uint32_t S, size_t stride) {   // e overwrites a[B..B+S)
  zk_take_prio(S);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S) return;
  const Fr *dbuf = dbuf_all + (size_t)blockIdx.y * B;
  Fr acc = Fr::zero();
  for (uint32_t j = i; j < B; j += S) acc = acc + dbuf[j];
  a_all[blockIdx.y * stride + B + i] = acc;
}
// inverse post-pass.  U0 (B values, already scaled by 1/B), U1 (S values, scaled by 1/S):
//   tmp[i] = U0[i]*w^i ; U1[i] -= sum_{j>=1} tmp[i + j*S] ; U1[i] *= w^-i ; a[i] = (U0[i]+U1[i])/2 (i<S) ; a[B+i] = (U0[i]-U1[i])/2 ; a[i] = U0[i] (S<=i<B)
// In place on a = [U0 (B) | U1 (S)]: thread i < S reads U0[i], U0[i + kS], U1[i] and writes a[i], a[B+i]; entries a[S..B) = U0[S..B) stay as they are.
__global__ void k_step_inv_post(Fr *a_all, const Fr *__restrict__ wpow, const Fr *__restrict__ winvpow, Fr half, uint32_t B, uint32_t S, size_t stride) {
  zk_take_prio(S);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= S) return; Fr *a = a_all + blockIdx.y * stride;
  Fr u1 = a[B + i]; for (uint32_t j = i + S; j < B; j += S) u1 = u1 - a[j] * wpow[j];
  u1 = u1 * winvpow[i]; Fr u0 = a[i]; a[i] = (u0 + u1) * half; a[B + i] = (u0 - u1) * half;
}

// iFFT immediately followed by cosetFFT (what the witness map does to A and B): the inverse transform's recombination pass and the forward transform's factor,
// pre-pass and fold touch the same elements — thread i < S owns the index class {i, i + S, i + 2S, ... < B} and B + i —, so they are ONE pass over the vectors
// instead of three (k_step_inv_post, k_step_fwd_pre with the coset factors, k_step_fold) and one launch instead of three
__global__ void k_step_inv_fwd(Fr *a_all, const Fr *__restrict__ wpow, const Fr *__restrict__ winvpow, Fr half, const Fr *__restrict__ cf, uint32_t B,
    uint32_t S, size_t stride) {
  zk_take_prio(S);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= S) return; Fr *a = a_all + blockIdx.y * stride;
  Fr u1 = a[B + i]; for (uint32_t j = i + S; j < B; j += S) u1 = u1 - a[j] * wpow[j];
  // the inverse transform's a[i], a[B + i]; a[S..B) are final as they are
  u1 = u1 * winvpow[i];
  const Fr u0 = a[i], lo = (u0 + u1) * half, hi = (u0 - u1) * half;
  const Fr x = lo * cf[i], y = hi * cf[B + i]; a[i] = x + y; Fr e = wpow[i] * (x - y);            // the forward pre-pass on g^i a[i]: c[i], d[i]
  for (uint32_t j = i + S; j < B; j += S) { const Fr xx = a[j] * cf[j]; a[j] = xx; e = e + wpow[j] * xx; }
  a[B + i] = e;                                                                                  // the fold: e[i] = sum_j d[i + jS]
}

struct DomainTables {                                            // immutable per key: shared by every Domain object copied from the first
  size_t m = 0; bool step = false; size_t B = 0, S = 0;           // step: m = B + S
  std::unique_ptr<Radix2Tables> big, small;                       // basic: only `big` (size m)
  DevBuf<Fe32> coset_fwd, coset_inv, zinv, wpow, winvpow;
  DevBuf<Fe32> scale_big, scale_small;                            // 1/n as a table: the pre-scale of the stage-per-launch path (beyond 2^22 points)
  // the tile kernels' factor forms (f 2^261): g^i per element, the constants 1, 1/B (or 1/m), 1/S
  DevBuf<Fe32> coset_fwd261;
  // radix-2 domains in the two-pass range: 1/m g^i 2^261, the inverse transform's output factors when the coset transform follows
  DevBuf<Fe32> inv_coset261;
  Fe32 one261, inv_big261, inv_small261;
  HFr half;
};
struct Domain::Impl {
  std::shared_ptr<DomainTables> t; DomainTables &d_;              // (d_ keeps the code below unchanged: every table is reached through it)
  size_t &m;
  bool &step;
  size_t &B, &S;
  std::unique_ptr<Radix2Tables> &big, &small;
  DevBuf<Fe32> &coset_fwd, &coset_inv, &zinv, &wpow, &winvpow, &scale_big, &scale_small, &coset_fwd261;
  Fe32 &one261, &inv_big261, &inv_small261;
  HFr &half;
  DevBuf<Fe32> scratch; size_t scratch_stride = 0;                // per object
  explicit Impl(std::shared_ptr<DomainTables> tt) : t(tt), d_(*t), m(d_.m), step(d_.step), B(d_.B), S(d_.S), big(d_.big), small(d_.small),
      coset_fwd(d_.coset_fwd), coset_inv(d_.coset_inv), zinv(d_.zinv), wpow(d_.wpow), winvpow(d_.winvpow),
                                                        scale_big(d_.scale_big), scale_small(d_.scale_small), coset_fwd261(d_.coset_fwd261), one261(d_.one261),
                                                            inv_big261(d_.inv_big261), inv_small261(d_.inv_small261), half(d_.half) {
                                                        }
};
Domain::Domain(const Domain &peer) : impl(new Impl(peer.impl->t)) {
  Impl &d = *impl;
  d.scratch_stride = d.m;
  d.scratch = DevBuf<Fe32>(d.step ? 6 * d.B : 3 * d.m);
}

Domain::Domain(size_t min_size) : impl(new Impl(std::make_shared<DomainTables>())) {
  Impl &d = *impl; if (min_size <= 1) throw GpuError("domain: size");
  size_t lg = ceil_log2(min_size);
  if (min_size == ((size_t)1 << lg)) { d.m = min_size; }
  else { size_t big = (size_t)1 << (lg - 1), small = min_size - big, rs = (size_t)1 << ceil_log2(small); d.m = (small == rs) ? min_size : big + rs;
         if (d.m != ((size_t)1 << ceil_log2(d.m))) { d.step = true; d.B = (size_t)1 << (ceil_log2(d.m) - 1); d.S = d.m - d.B; } }
  if (ceil_log2(d.m) > 28) throw GpuError("domain: too large for Fr's 2-adicity");
  HFr g = fr_coset_gen(), ginv = g.inv(), one = HFr::one();
  d.half = HFr::from_u64(2).inv();
  if (!d.step) {
    HFr w = fr_root_of_unity(d.m), minv = HFr::from_u64(d.m).inv(); d.big.reset(new Radix2Tables(d.m, w));
    // cosetFFT: multiply by g^i then FFT.  icosetFFT: iFFT (incl. 1/m) then multiply by g^-i  (basic_radix2_domain.tcc:74-88)
    auto cf = geometric_table(d.m, one, g), ci = geometric_table(d.m, minv, ginv), sc = geometric_table(d.m, minv, one);
    d.coset_fwd = DevBuf<Fe32>(d.m); d.coset_fwd.upload(cf.data(), d.m); d.coset_inv = DevBuf<Fe32>(d.m); d.coset_inv.upload(ci.data(), d.m);
    d.scale_big = DevBuf<Fe32>(d.m); d.scale_big.upload(sc.data(), d.m); d.inv_big261 = fe261(minv); d.inv_small261 = fe261(one);
    HFr z = (g.pow_u64(d.m) - one).inv(); Fe32 zz; memcpy(&zz, z.l, 32); d.zinv = DevBuf<Fe32>(1); d.zinv.upload(&zz, 1);       // divide_by_Z_on_coset :103-112
  } else {
    HFr w = fr_root_of_unity((size_t)1 << ceil_log2(d.m)), wb = w.sqr(), ws = fr_root_of_unity(d.S), winv = w.inv();   // step_radix2_domain.tcc:20-37
    d.big.reset(new Radix2Tables(d.B, wb)); d.small.reset(new Radix2Tables(d.S, ws));
    auto cf = geometric_table(d.m, one, g), ci = geometric_table(d.m, one, ginv), wp = geometric_table(d.B, one, w), wip = geometric_table(d.S, one, winv);
    auto sb = geometric_table(d.B, HFr::from_u64(d.B).inv(), one), ss = geometric_table(d.S, HFr::from_u64(d.S).inv(), one);
    d.coset_fwd = DevBuf<Fe32>(d.m); d.coset_fwd.upload(cf.data(), d.m); d.coset_inv = DevBuf<Fe32>(d.m); d.coset_inv.upload(ci.data(), d.m);
    d.wpow = DevBuf<Fe32>(d.B); d.wpow.upload(wp.data(), d.B); d.winvpow = DevBuf<Fe32>(d.S); d.winvpow.upload(wip.data(), d.S);
    d.scale_big = DevBuf<Fe32>(d.B);
    d.scale_big.upload(sb.data(), d.B);
    d.scale_small = DevBuf<Fe32>(d.S);
    d.scale_small.upload(ss.data(), d.S);
    d.inv_big261 = fe261(HFr::from_u64(d.B).inv());
    d.inv_small261 = fe261(HFr::from_u64(d.S).inv());
    // divide_by_Z_on_coset (:242-260): P[i] /= (g^S * Z0 * w^(2S i) - w^S * Z0) for i < B ; P[B+i] /= Z1
    std::vector<Fe32> zt(d.m);
    HFr Z0 = g.pow_u64(d.B) - one, cSZ0 = g.pow_u64(d.S) * Z0, wS = w.pow_u64(d.S), wSZ0 = wS * Z0, w2S = w.pow_u64(2 * d.S), elt = one;
    // batch inversion of the B denominators
    std::vector<HFr> den(d.B), pre(d.B); for (size_t i = 0; i < d.B; i++) { den[i] = cSZ0 * elt - wSZ0; elt = elt * w2S; }
    HFr acc = one; for (size_t i = 0; i < d.B; i++) { pre[i] = acc; acc = acc * den[i]; } HFr ai = acc.inv();
    for (size_t i = d.B; i-- > 0;) { HFr v = ai * pre[i]; ai = ai * den[i]; memcpy(&zt[i], v.l, 32); }
    HFr cw = g * w, Z1 = ((cw.pow_u64(d.B) - one) * (cw.pow_u64(d.S) - wS)).inv(); for (size_t i = 0; i < d.S; i++) memcpy(&zt[d.B + i], Z1.l, 32);
    d.zinv = DevBuf<Fe32>(d.m); d.zinv.upload(zt.data(), d.m);
  }
  d.one261 = fe261(one);
  {
    auto cf = geometric_table(d.m, one, g);
    for (auto &f : cf) {
      HFr v;
      memcpy(v.l, &f, 32);
      f = fe261(v);
    }
    d.coset_fwd261 = DevBuf<Fe32>(d.m);
    d.coset_fwd261.upload(cf.data(), d.m);
    if (!d.step && ntt_two_pass(d.big->logn)) {
      auto ic = geometric_table(d.m, HFr::from_u64(d.m).inv(), g);
      for (auto &f : ic) {
        HFr v;
        memcpy(v.l, &f, 32);
        f = fe261(v);
      }
      d.t->inv_coset261 = DevBuf<Fe32>(d.m);
      d.t->inv_coset261.upload(ic.data(), d.m);
    }
  }
  d.scratch_stride = d.m; d.scratch = DevBuf<Fe32>(d.step ? 6 * d.B : 3 * d.m);
}
Domain::~Domain() = default;
size_t Domain::m() const { return impl->m; }
bool Domain::is_step() const { return impl->step; }

static void mul_table(Fe32 *a, const Fe32 *t, size_t n, int batch, size_t stride) {
  hipLaunchKernelGGL(k_fr_mul_table, dim3(cdiv(n, 256), batch), dim3(256), 0, gpu().stream, (Fr *)a, (const Fr *)t, (uint32_t)n, stride);
}

void Domain::fft(Fe32 *data, int batch, size_t stride) { fft_with_factors(data, batch, stride, nullptr); }
void Domain::fft_with_factors(Fe32 *data, int batch, size_t stride, const Fe32 *cf) {   // step domains only: cf = the coset factors, folded into the pre-pass
  Stage st("ntt.forward");
  Impl &d = *impl;
  if (batch > 3) throw GpuError("domain: batch > 3");
  if (!d.step && cf) throw GpuError("domain: factors on a basic domain go through coset_fft");
  if (!d.step) {
    radix2_transform(NttCall{data, d.scratch.get(), d.big->tw.get(), d.big->tw261.get(), d.big->logn, nullptr, nullptr, d.one261, nullptr, stride,
        d.scratch_stride}, batch);
    return;
  }
  // step_radix2_domain::FFT (:39-77): c / d / e pre-pass in place, then a B-point and an S-point transform of every vector; scratch = [d: 3B | transform
  // scratch: 3B]
  hipStream_t s = gpu().stream; Fe32 *dbuf = d.scratch.get(), *tmp = d.scratch.get() + 3 * d.B;
  hipLaunchKernelGGL(k_step_fwd_pre, dim3(cdiv(d.B, 256), batch), dim3(256), 0, s, (Fr *)data, (Fr *)dbuf, (const Fr *)d.wpow.get(), (const Fr *)cf,
      (uint32_t)d.B, zk_with_prio(d.S, ZKP_STEP), stride);
  hipLaunchKernelGGL(k_step_fold, dim3(cdiv(d.S, 256), batch), dim3(256), 0, s, (const Fr *)dbuf, (Fr *)data, (uint32_t)d.B, zk_with_prio(d.S, ZKP_STEP), stride);
  // (dbuf is free again after the fold: the S-point transform's scratch)
  radix2_transform_pair(NttCall{data, tmp, d.big->tw.get(), d.big->tw261.get(), d.big->logn, nullptr, nullptr, d.one261, nullptr, stride, d.B},
      NttCall{data + d.B, dbuf, d.small->tw.get(), d.small->tw261.get(), d.small->logn, nullptr, nullptr, d.one261, nullptr, stride, d.B}, batch);
}
void Domain::ifft(Fe32 *data, int batch, size_t stride) {
  Stage st("ntt.inverse"); Impl &d = *impl; if (batch > 3) throw GpuError("domain: batch > 3");
  // 1/m: the row pass's final factor
  if (!d.step) {
    radix2_transform(NttCall{data, d.scratch.get(), d.big->itw.get(), d.big->itw261.get(), d.big->logn, d.scale_big.get(), nullptr, d.inv_big261, nullptr,
        stride, d.scratch_stride}, batch);
    return;
  }
  // step_radix2_domain::iFFT (:79-140): both inverse transforms in place (1/B, 1/S folded into their loads), then the recombination pass
  hipStream_t s = gpu().stream; Fe32 *tmp = d.scratch.get() + 3 * d.B;
  radix2_transform_pair(NttCall{data, tmp, d.big->itw.get(), d.big->itw261.get(), d.big->logn, d.scale_big.get(), nullptr, d.inv_big261, nullptr, stride, d.B},
      NttCall{data + d.B, d.scratch.get(), d.small->itw.get(), d.small->itw261.get(), d.small->logn, d.scale_small.get(), nullptr, d.inv_small261, nullptr,
      stride, d.B}, batch);
  Fr half; memcpy(&half, d.half.l, 32);
  hipLaunchKernelGGL(k_step_inv_post, dim3(cdiv(d.S, 256), batch), dim3(256), 0, s, (Fr *)data, (const Fr *)d.wpow.get(), (const Fr *)d.winvpow.get(), half,
      (uint32_t)d.B, zk_with_prio(d.S, ZKP_STEP), stride);
}
void Domain::ifft_then_coset_fft(Fe32 *data, int batch, size_t stride) {
  Impl &d = *impl;
  if (batch > 3) throw GpuError("domain: batch > 3");
  if (!d.step) {
    // (without the folded table: the two transforms as they are called one by one, g^i multiplied in by the forward transform's column pass)
    if (!d.t->inv_coset261.get()) { ifft(data, batch, stride); coset_fft(data, batch, stride); return; }
    { Stage st("ntt.inverse");      // 1/m and g^i leave with the row pass's last product
      radix2_transform(NttCall{data, d.scratch.get(), d.big->itw.get(), d.big->itw261.get(), d.big->logn, nullptr, nullptr, d.inv_big261, nullptr, stride,
          d.scratch_stride, d.t->inv_coset261.get()}, batch); }
    { Stage st("ntt.forward");
      radix2_transform(NttCall{data, d.scratch.get(), d.big->tw.get(), d.big->tw261.get(), d.big->logn, nullptr, nullptr, d.one261, nullptr, stride,
          d.scratch_stride}, batch); }
    return;
  }
  hipStream_t s = gpu().stream;
  Fe32 *dbuf = d.scratch.get(), *tmp = d.scratch.get() + 3 * d.B;
  Fr half;
  memcpy(&half, d.half.l, 32);
  { Stage st("ntt.inverse");
    radix2_transform_pair(NttCall{data, tmp, d.big->itw.get(), d.big->itw261.get(), d.big->logn, d.scale_big.get(), nullptr, d.inv_big261, nullptr, stride,
        d.B}, NttCall{data + d.B, dbuf, d.small->itw.get(), d.small->itw261.get(), d.small->logn, d.scale_small.get(), nullptr, d.inv_small261, nullptr,
        stride, d.B}, batch);
    hipLaunchKernelGGL(k_step_inv_fwd, dim3(cdiv(d.S, 256), batch), dim3(256), 0, s, (Fr *)data, (const Fr *)d.wpow.get(), (const Fr *)d.winvpow.get(), half,
        (const Fr *)d.coset_fwd.get(), (uint32_t)d.B, zk_with_prio(d.S, ZKP_STEP), stride);
  }
  { Stage st("ntt.forward");
    radix2_transform_pair(NttCall{data, tmp, d.big->tw.get(), d.big->tw261.get(), d.big->logn, nullptr, nullptr, d.one261, nullptr, stride, d.B},
        NttCall{data + d.B, dbuf, d.small->tw.get(), d.small->tw261.get(), d.small->logn, nullptr, nullptr, d.one261, nullptr, stride, d.B}, batch);
  }
}
void Domain::coset_fft(Fe32 *data, int batch, size_t stride) {
  Impl &d = *impl; if (batch > 3) throw GpuError("domain: batch > 3");
  // g^i folded into the load
  if (!d.step) {
    Stage st("ntt.forward");
    radix2_transform(NttCall{data, d.scratch.get(), d.big->tw.get(), d.big->tw261.get(), d.big->logn, d.coset_fwd.get(), d.coset_fwd261.get(), d.one261,
        nullptr, stride, d.scratch_stride}, batch);
    return;
  }
  fft_with_factors(data, batch, stride, d.coset_fwd.get());
}
void Domain::icoset_fft(Fe32 *data, int batch, size_t stride) {
  Impl &d = *impl;
  if (!d.step) { Stage st("ntt.inverse");   // coset_inv carries 1/m; it is folded into the store unless the transform is a single pass
    if (d.big->logn <= NTT_TILE_LOG) {
      radix2_transform(NttCall{data, d.scratch.get(), d.big->itw.get(), d.big->itw261.get(), d.big->logn, nullptr, nullptr, d.one261, nullptr, stride,
          d.scratch_stride}, batch);
      mul_table(data, d.coset_inv.get(), d.m, batch, stride);
    }
    else radix2_transform(NttCall{data, d.scratch.get(), d.big->itw.get(), d.big->itw261.get(), d.big->logn, nullptr, nullptr, d.one261, d.coset_inv.get(),
        stride, d.scratch_stride}, batch);
    return; }
  ifft(data, batch, stride); mul_table(data, d.coset_inv.get(), d.m, batch, stride);
}
// see ecntt.cuh: the H query (m - 1 points) in the Lagrange basis of the coset (m points), for both kinds of domain
bool Domain::supports_h_lagrange() const { return true; }
void Domain::h_query_to_coset_lagrange(const G1AffineRaw *h, size_t n_in, G1AffineRaw *out) {
  Impl &d = *impl; if (n_in > d.m) throw GpuError("domain: h_query_to_coset_lagrange"); hipStream_t s = gpu().stream; const size_t m = d.m;
  DevBuf<G1AffineRaw> din(n_in ? n_in : 1), dout(m);
  DevBuf<uint8_t> data(m * sizeof(XYZZ<Fq>));
  if (n_in) din.upload(h, n_in);
  XYZZ<Fq> *X = (XYZZ<Fq> *)data.get();
  // unscaled inverse DFT over group elements, natural order in, natural order out (bit reversal in the final conversion)
  auto idft = [&](XYZZ<Fq> *part, const Radix2Tables &t, Affine<Fq> *o) {
    for (int st = t.logn; st >= 1; st--) hipLaunchKernelGGL(k_ecntt_stage, dim3(cdiv(t.n / 2, 64)), dim3(64), 0, s, part, (const Fr *)t.itw.get(), t.logn, st);
    hipLaunchKernelGGL(k_ecntt_finish, dim3(cdiv(t.n, 64)), dim3(64), 0, s, (const XYZZ<Fq> *)part, t.logn, o); };
  // basic: g^-i / m; step: g^-i
  hipLaunchKernelGGL(k_ecntt_prescale, dim3(cdiv(m, 64)), dim3(64), 0, s, (const Affine<Fq> *)din.get(), (uint32_t)n_in, (const Fr *)d.coset_inv.get(),
      (uint32_t)m, X);
  if (!d.step) idft(X, *d.big, (Affine<Fq> *)dout.get());
  else {
    DevBuf<uint8_t> data2(m * sizeof(XYZZ<Fq>)); XYZZ<Fq> *R = (XYZZ<Fq> *)data2.get();
    HFr ib = HFr::from_u64(d.B).inv(), is = HFr::from_u64(d.S).inv(), hib = d.half * ib, his = d.half * is;
    Fr f_hib, f_ib, f_his;
    memcpy(&f_hib, hib.l, 32);
    memcpy(&f_ib, ib.l, 32);
    memcpy(&f_his, his.l, 32);
    hipLaunchKernelGGL(k_ecntt_step_pre, dim3(cdiv(m, 64)), dim3(64), 0, s, (const XYZZ<Fq> *)X, R, (const Fr *)d.wpow.get(), (const Fr *)d.winvpow.get(),
        f_hib, f_ib, f_his, (uint32_t)d.B, (uint32_t)d.S);
    idft(R, *d.big, (Affine<Fq> *)dout.get()); idft(R + d.B, *d.small, (Affine<Fq> *)dout.get() + d.B);
  }
  HIP_CHECK(hipGetLastError()); dout.download(out, m);
}
// see ecntt.cuh: Lstar (n_vars + 1 points) = L extended to all variables minus the C polynomial's share of the H term
bool Domain::supports_c_fold() const { return true; }
void Domain::fold_c_into_l(const G1AffineRaw *h_lagrange, const R1csHost &cs, const G1AffineRaw *L, G1AffineRaw *out) {
  // (step domains: logm is the size of the big part)
  Impl &d = *impl;
  hipStream_t s = gpu().stream;
  const size_t m = d.m, n_all = cs.n_vars + 1;
  const int logm = d.big->logn;
  // column form of C; coefficients classified so that +-1 cost an addition only
  std::vector<uint32_t> colptr(n_all + 1, 0); const std::vector<uint32_t> &rp = cs.rowptr[2], &cl = cs.col[2]; size_t nnz = cl.size();
  for (size_t e = 0; e < nnz; e++) colptr[cl[e] + 1]++; for (size_t v = 0; v < n_all; v++) colptr[v + 1] += colptr[v];
  std::vector<uint32_t> rowidx(nnz ? nnz : 1), fillp(colptr.begin(), colptr.end() - 1);
  std::vector<uint8_t> kind(nnz ? nnz : 1);
  std::vector<Fe32> coef(nnz ? nnz : 1);
  Fe32 one{}; one.l[0] = 1; Fe32 minus_one; { HFr mo = HFr::one().neg().from_mont(); memcpy(&minus_one, mo.l, 32); }
  for (size_t k = 0; k < cs.n_cons; k++) for (uint32_t e = rp[k]; e < rp[k + 1]; e++) {
    uint32_t pos = fillp[cl[e]]++;
    rowidx[pos] = (uint32_t)k;
    coef[pos] = cs.coeff[2][e];
    kind[pos] = !memcmp(&cs.coeff[2][e], &one, 32) ? 0 : !memcmp(&cs.coeff[2][e], &minus_one, 32) ? 1 : 2; }
  DevBuf<uint32_t> d_colptr(n_all + 1), d_rowidx(rowidx.size()); DevBuf<uint8_t> d_kind(kind.size()); DevBuf<Fe32> d_coef(coef.size());
  d_colptr.upload(colptr.data(), colptr.size());
  d_rowidx.upload(rowidx.data(), rowidx.size());
  d_kind.upload(kind.data(), kind.size());
  d_coef.upload(coef.data(), coef.size());
  DevBuf<G1AffineRaw> dp(m), dl(cs.n_vars - cs.n_inputs ? cs.n_vars - cs.n_inputs : 1), dout(n_all);
  dp.upload(h_lagrange, m);
  if (cs.n_vars > cs.n_inputs) dl.upload(L, cs.n_vars - cs.n_inputs);
  DevBuf<uint8_t> b0(m * sizeof(XYZZ<Fq>)), b1(m * sizeof(XYZZ<Fq>)); XYZZ<Fq> *X = (XYZZ<Fq> *)b0.get(), *Y = (XYZZ<Fq> *)b1.get();
  auto stages = [&](XYZZ<Fq> *x, const Fe32 *tw, int lg) {
    for (int st = lg; st >= 1; st--) hipLaunchKernelGGL(k_ecntt_stage, dim3(cdiv(((size_t)1 << lg) / 2, 64)), dim3(64), 0, s, x, (const Fr *)tw, lg, st);
  };
  auto columns = [&](const XYZZ<Fq> *U, int lg) { hipLaunchKernelGGL(k_fold_c_columns, dim3(cdiv(n_all, 64)), dim3(64), 0, s, d_colptr.get(), d_rowidx.get(),
      d_kind.get(), (const Fr *)d_coef.get(), U, lg,
                     (const Affine<Fq> *)dl.get(), (uint32_t)cs.n_inputs, (uint32_t)n_all, (Affine<Fq> *)dout.get()); };
  if (!d.step) {
    HFr zinv; { Fe32 z; d.zinv.download(&z, 1); memcpy(zinv.l, &z, 32); } Fr fz; memcpy(&fz, zinv.l, 32);
    // g^i / m
    std::vector<Fe32> gm = geometric_table(m, HFr::from_u64(m).inv(), fr_coset_gen());
    DevBuf<Fe32> d_gm(m);
    d_gm.upload(gm.data(), m);
    // zinv * P
    hipLaunchKernelGGL(k_ecntt_scale_const, dim3(cdiv(m, 64)), dim3(64), 0, s, (const Affine<Fq> *)dp.get(), fz, (uint32_t)m, X);
    // cosetFFT^T = D_g . DFT_w: the DFT ...
    stages(X, d.big->tw.get(), logm);
    // ... then g^i, with the 1/m of the inverse transform
    hipLaunchKernelGGL(k_ecntt_permute_scale, dim3(cdiv(m, 64)), dim3(64), 0, s, (const XYZZ<Fq> *)X, (const Fr *)d_gm.get(), logm, Y);
    // iFFT^T = (1/m) DFT_(1/w); result bit-reversed
    stages(Y, d.big->itw.get(), logm);
    columns(Y, logm);
  } else {
    const int lb = d.big->logn, ls = d.small->logn;
    // zinv[j] * P_j (divide_by_Z_on_coset is a table here)
    hipLaunchKernelGGL(k_ecntt_scale_table, dim3(cdiv(m, 64)), dim3(64), 0, s, (const Affine<Fq> *)dp.get(), (const Fr *)d.zinv.get(), (uint32_t)m, X);
    // the two DFTs of the forward transform's transpose
    stages(X, d.big->tw.get(), lb);
    stages(X + d.B, d.small->tw.get(), ls);
    // Pre^T and g^i
    hipLaunchKernelGGL(k_ecntt_step_fwd_T, dim3(cdiv(d.B, 64)), dim3(64), 0, s, (const XYZZ<Fq> *)X, (const XYZZ<Fq> *)(X + d.B), lb, ls,
        (const Fr *)d.wpow.get(), (const Fr *)d.coset_fwd.get(), Y);
    // the inverse transform's transpose, as for the H query: Post^T with 1/B, 1/S, then the inverse DFTs of the parts
    HFr ib = HFr::from_u64(d.B).inv(), is = HFr::from_u64(d.S).inv(), hib = d.half * ib, his = d.half * is;
    Fr f_hib, f_ib, f_his;
    memcpy(&f_hib, hib.l, 32);
    memcpy(&f_ib, ib.l, 32);
    memcpy(&f_his, his.l, 32);
    hipLaunchKernelGGL(k_ecntt_step_pre, dim3(cdiv(m, 64)), dim3(64), 0, s, (const XYZZ<Fq> *)Y, X, (const Fr *)d.wpow.get(), (const Fr *)d.winvpow.get(),
        f_hib, f_ib, f_his, (uint32_t)d.B, (uint32_t)d.S);
    stages(X, d.big->itw.get(), lb); stages(X + d.B, d.small->itw.get(), ls);
    hipLaunchKernelGGL(k_ecntt_unpermute, dim3(cdiv(d.B, 64)), dim3(64), 0, s, (const XYZZ<Fq> *)X, lb, Y);
    hipLaunchKernelGGL(k_ecntt_unpermute, dim3(cdiv(d.S, 64)), dim3(64), 0, s, (const XYZZ<Fq> *)(X + d.B), ls, Y + d.B);
    columns(Y, 0);
  }
  HIP_CHECK(hipGetLastError()); dout.download(out, n_all);
}
const Fe32 *Domain::zinv_dev() const { return impl->zinv.get(); }
bool Domain::zinv_is_table() const { return impl->step; }
void Domain::qap_pointwise(Fe32 *a, const Fe32 *b, const Fe32 *c) {
  hipLaunchKernelGGL(k_qap_pointwise, dim3(cdiv(impl->m, 256)), dim3(256), 0, gpu().stream, (Fr *)a, (const Fr *)b, (const Fr *)c,
      (const Fr *)impl->zinv.get(), impl->step ? 1 : 0, (uint32_t)impl->m);
}

// packed: [ones bitmap | other bitmap | (canon bitmap) | block offsets | values] already on the device (layout of Prover::set_witness / set_witness_tagged).
// canon: 0 = every value is in Montgomery form, 1 = every value is canonical (the "other" bitmap doubles as the list of values to convert), 2 = a third bitmap
// says which
void expand_witness_dev(const uint8_t *packed, size_t words, const Fe32 &one_value, int canon, size_t n, Fe32 *out, uint8_t *tags, uint32_t *other_vars) {
  const uint64_t *ones = (const uint64_t *)packed, *other = ones + words, *third = other + words; const size_t nbm = canon == 2 ? 3 : 2;
  const uint32_t *off = (const uint32_t *)(ones + nbm * words); const Fr *vals = (const Fr *)(packed + expand_values_offset(words, canon));
  Fr one; memcpy(&one, &one_value, 32);
  hipLaunchKernelGGL(k_expand_witness, dim3(cdiv(n, 256)), dim3(256), 0, gpu().stream, ones, other,
      canon == 0 ? (const uint64_t *)nullptr : canon == 1 ? other : third, off, vals, one, zk_with_prio(n, ZKP_EXPAND), (Fr *)out, tags, other_vars);   // (the priority rides in bits 24..: not for 16 M variables and more)
}
// tags and the list of other values of an assignment that already lies in device memory (k_classify_witness): counters[parity] receives the list's length,
// counters[parity ^ 1] is cleared for the next call
void classify_witness_dev(const Fe32 *z, size_t n, uint8_t *tags, uint32_t *other_vars, uint32_t *counters, int parity) {
  Fr one; memcpy(&one, FrParams::R1, 32);
  hipLaunchKernelGGL(k_classify_witness, dim3(cdiv(n, 256 * CLASSIFY_PER_LANE)), dim3(256), 0, gpu().stream, (const Fr *)z, one, zk_with_prio(n, ZKP_EXPAND), tags, other_vars,
      counters + (parity & 1), counters + ((parity & 1) ^ 1));
}
// a circuit board's assignment from its tag bytes and its candidates' values, both already on the device (k_expand_board); counters as in classify_witness_dev
void expand_board_dev(const uint8_t *board_tags, size_t n, const uint32_t *cand, const Fe32 *cand_vals, size_t n_cand, Fe32 *z, uint8_t *tags, uint32_t *other_vars,
    uint32_t *counters, int parity, bool vals_by_var) {
  Fr one; memcpy(&one, FrParams::R1, 32); const unsigned tb = cdiv(n, 256);
  hipLaunchKernelGGL(k_expand_board, dim3(tb + cdiv(n_cand, 256)), dim3(256), 0, gpu().stream, board_tags, zk_with_prio(n, ZKP_EXPAND), tb, cand, (const Fr *)cand_vals,
      (uint32_t)n_cand | (vals_by_var ? 0x80000000u : 0u), one, (Fr *)z, tags, other_vars, counters + (parity & 1), counters + ((parity & 1) ^ 1));
}
void *gpu_host_register(void *p, size_t bytes) {
  if (!p || !bytes) return nullptr; gpu();
  if (hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  void *d = nullptr; if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || !d) { (void)hipGetLastError(); hipHostUnregister(p); return nullptr; }
  return d;
}
void gpu_host_unregister(void *p) { if (p) { (void)hipHostUnregister(p); (void)hipGetLastError(); } }
static std::atomic<uint64_t> g_general_path_repeats{0}, g_queries_without_tables{0};
uint64_t queries_without_tables() { return g_queries_without_tables.load(std::memory_order_relaxed); }
void note_query_without_tables() { g_queries_without_tables.fetch_add(1, std::memory_order_relaxed); }
uint64_t general_path_repeats() { return g_general_path_repeats.load(std::memory_order_relaxed); }
void note_general_path_repeat() { g_general_path_repeats.fetch_add(1, std::memory_order_relaxed); }
void merge_equal_columns_dev(Fe32 *z, uint8_t *tags, const uint32_t *grp_ptr, const uint32_t *grp_mem, size_t n_groups) {
  if (n_groups) hipLaunchKernelGGL(k_merge_equal_columns, dim3(cdiv(n_groups, 64)), dim3(64), 0, gpu().stream, (Fr *)z, tags, grp_ptr, grp_mem, (uint32_t)n_groups);
}
void fr_to_mont_dev(Fe32 *a, size_t n) { if (n) hipLaunchKernelGGL(k_fr_to_mont, dim3(cdiv(n, 256)), dim3(256), 0, gpu().stream, (Fr *)a, (uint32_t)n); }
void fr_from_mont_dev(Fe32 *a, size_t n) { if (n) hipLaunchKernelGGL(k_fr_from_mont, dim3(cdiv(n, 256)), dim3(256), 0, gpu().stream, (Fr *)a, (uint32_t)n); }

// ======================================================================================================================
// R1CS rows
// ======================================================================================================================
// immutable per key
struct R1csArrays { size_t n_inputs, n_vars, n_cons; DevBuf<uint32_t> rowptr[3], col[3], cid[3], long_rows[3]; size_t n_long[3] = {0, 0, 0}; DevBuf<Fe32> ctab;
    DevBuf<uint32_t> long_any; size_t n_long_any = 0; };
struct R1csDev::Impl {
  std::shared_ptr<R1csArrays> a;
  size_t &n_inputs, &n_vars, &n_cons;
  DevBuf<uint32_t> (&rowptr)[3], (&col)[3], (&cid)[3], (&long_rows)[3];
  size_t (&n_long)[3];
  DevBuf<Fe32> &ctab;
  DevBuf<uint32_t> &long_any;
  size_t &n_long_any;
  // per object. h_fail: mapped host word the prover's row kernels store the evaluation number to when a constraint is violated
  DevBuf<uint32_t> flag;
  uint32_t *h_flag = nullptr;
  uint32_t *h_fail = nullptr, *d_fail = nullptr, seq = 0;
  explicit Impl(std::shared_ptr<R1csArrays> aa) : a(aa), n_inputs(a->n_inputs), n_vars(a->n_vars), n_cons(a->n_cons), rowptr(a->rowptr), col(a->col),
      cid(a->cid), long_rows(a->long_rows), n_long(a->n_long), ctab(a->ctab), long_any(a->long_any), n_long_any(a->n_long_any) {
  }
  void own_words() {
    flag = DevBuf<uint32_t>(1);
    HIP_CHECK(hipHostMalloc((void **)&h_flag, 4));
    HIP_CHECK(hipHostMalloc((void **)&h_fail, 8, hipHostMallocMapped));   // [evaluation number of the last violation, a row that violated]
    h_fail[0] = 0; h_fail[1] = 0;
    HIP_CHECK(hipHostGetDevicePointer((void **)&d_fail, h_fail, 0));
  }
  ~Impl() { if (h_flag) hipHostFree(h_flag); if (h_fail) hipHostFree(h_fail); }
};
R1csDev::R1csDev(const R1csDev &peer) : impl(new Impl(peer.impl->a)) { impl->own_words(); }
R1csDev::R1csDev(const R1csHost &h) : impl(new Impl(std::make_shared<R1csArrays>())) {
  Impl &d = *impl; d.n_inputs = h.n_inputs; d.n_vars = h.n_vars; d.n_cons = h.n_cons;
  // coefficient table: slot 0 = +1, slot 1 = -1 (handled without a multiply), the rest in order of first appearance
  std::vector<Fe32> tab(2); { HFr one = HFr::one(), m1 = one.neg(); memcpy(&tab[0], one.l, 32); memcpy(&tab[1], m1.l, 32); }
  struct Key { uint32_t l[8]; bool operator<(const Key &o) const { return memcmp(l, o.l, 32) < 0; } };
  std::vector<std::pair<Key, uint32_t>> seen;   // small: linear probing over a sorted vector would be overkill; use a simple hash map below
  // open-addressing hash on the low 64 bits
  size_t cap = 1 << 12; std::vector<int64_t> slots(cap, -1); std::vector<Fe32> keys; std::vector<uint32_t> vals;
  auto lookup = [&](const Fe32 &c) -> uint32_t {
    uint64_t hsh = ((uint64_t)c.l[1] << 32 | c.l[0]) * 0x9E3779B97F4A7C15ull ^ c.l[7]; size_t pos = hsh & (cap - 1);
    for (;;) { int64_t s = slots[pos]; if (s < 0) break; if (!memcmp(&keys[s], &c, 32)) return vals[s]; pos = (pos + 1) & (cap - 1); }
    if (keys.size() * 2 >= cap) throw GpuError("r1cs: more than 2048 distinct coefficients");
    HFr v; memcpy(v.l, &c, 32); v = v.to_mont(); Fe32 m; memcpy(&m, v.l, 32); uint32_t id;
    if (!memcmp(&m, &tab[0], 32)) id = 0; else if (!memcmp(&m, &tab[1], 32)) id = 1; else { id = (uint32_t)tab.size(); tab.push_back(m); }
    slots[pos] = (int64_t)keys.size(); keys.push_back(c); vals.push_back(id); return id; };
  for (int m = 0; m < 3; m++) {
    if (h.rowptr[m].size() != h.n_cons + 1) throw GpuError("r1cs: rowptr size");
    std::vector<uint32_t> ids(h.col[m].size()); for (size_t k = 0; k < ids.size(); k++) ids[k] = lookup(h.coeff[m][k]);
    d.rowptr[m] = DevBuf<uint32_t>(h.n_cons + 1); d.rowptr[m].upload(h.rowptr[m].data(), h.n_cons + 1);
    d.col[m] = DevBuf<uint32_t>(ids.size() + 1); d.cid[m] = DevBuf<uint32_t>(ids.size() + 1);
    if (!ids.empty()) { d.col[m].upload(h.col[m].data(), ids.size()); d.cid[m].upload(ids.data(), ids.size()); }
    std::vector<uint32_t> lr; for (size_t i = 0; i < h.n_cons; i++) if (h.rowptr[m][i + 1] - h.rowptr[m][i] > R1CS_LONG_ROW) lr.push_back((uint32_t)i);
    d.n_long[m] = lr.size(); d.long_rows[m] = DevBuf<uint32_t>(lr.size() + 1); if (!lr.empty()) d.long_rows[m].upload(lr.data(), lr.size());
  }
  {
    std::vector<uint32_t> lr;
    for (size_t i = 0; i < h.n_cons; i++) {
      bool lg = false;
      for (int m = 0; m < 3; m++) lg |= h.rowptr[m][i + 1] - h.rowptr[m][i] > R1CS_LONG_ROW;
      if (lg) lr.push_back((uint32_t)i);
    }
    d.n_long_any = lr.size(); d.long_any = DevBuf<uint32_t>(lr.size() + 1); if (!lr.empty()) d.long_any.upload(lr.data(), lr.size()); }
  d.ctab = DevBuf<Fe32>(tab.size()); d.ctab.upload(tab.data(), tab.size()); d.own_words();
}
R1csDev::~R1csDev() = default;
void R1csDev::eval(const Fe32 *z, Fe32 *abc, size_t m, const uint8_t *tags, bool write_c) {
  Stage st("r1cs.rows"); Impl &d = *impl; hipStream_t s = gpu().stream; if (m < d.n_cons + d.n_inputs + 1) throw GpuError("r1cs: domain too small");
  R1csMatrices M; for (int mm = 0; mm < 3; mm++) { M.rowptr[mm] = d.rowptr[mm].get(); M.col[mm] = d.col[mm].get(); M.cid[mm] = d.cid[mm].get(); }
  if (++d.seq == 0) d.seq = 1;
  if (tags) {   // the assignment came in compact form: a byte per variable says 0 / 1 / other (k_r1cs_rows_tagged)
    const uint32_t sb = (uint32_t)cdiv(m, 256);
    hipLaunchKernelGGL(k_r1cs_rows_tagged, dim3(sb + (unsigned)cdiv(d.n_long_any, 4)), dim3(256), 0, s, M, (const Fr *)d.ctab.get(), (const Fr *)z, tags,
        (uint32_t)d.n_cons, zk_with_prio(d.n_inputs, ZKP_ROWS), (uint32_t)m,
                       (const uint32_t *)d.long_any.get(), (uint32_t)d.n_long_any, sb, write_c ? 1 : 0, (Fr *)abc, d.seq, d.d_fail);
    return; }
  // rows of more than 16 terms exist: the one-launch form (short rows and one wave per long row) const uint32_t sb = (uint32_t)cdiv(m, 256);
  if (d.n_long_any) {
    const uint32_t sb = (uint32_t)cdiv(m, 256);
    hipLaunchKernelGGL(k_r1cs_rows_all, dim3(sb + (unsigned)cdiv(d.n_long_any, 4)), dim3(256), 0, s, M, (const Fr *)d.ctab.get(), (const Fr *)z,
        (uint32_t)d.n_cons, (uint32_t)d.n_inputs, (uint32_t)m, (const uint32_t *)d.long_any.get(), (uint32_t)d.n_long_any, sb, (Fr *)abc, d.seq, d.d_fail);
    return; }
  hipLaunchKernelGGL(k_r1cs_rows3, dim3(cdiv(m, 256)), dim3(256), 0, s, M, (const Fr *)d.ctab.get(), (const Fr *)z, (uint32_t)d.n_cons, (uint32_t)d.n_inputs,
      (uint32_t)m, (Fr *)abc, d.seq, d.d_fail);
  if (d.n_long_any) hipLaunchKernelGGL(k_r1cs_long_rows3, dim3((unsigned)d.n_long_any), dim3(64), 0, s, d.long_any.get(), M, (const Fr *)d.ctab.get(),
      (const Fr *)z, (uint32_t)m, (Fr *)abc, d.seq, d.d_fail);
}
bool R1csDev::check_result() const { return *impl->h_fail != impl->seq; }
uint32_t R1csDev::failed_row() const { return impl->h_fail[1]; }   // a constraint the last evaluation found violated (meaningful while check_result() is false)   // valid once the main stream has been synchronised after eval()
bool R1csDev::satisfied(const Fe32 *abc, size_t m) {
  Impl &d = *impl; hipStream_t s = gpu().stream; d.flag.zero();
  if (d.n_cons) hipLaunchKernelGGL(k_r1cs_check, dim3(cdiv(d.n_cons, 256)), dim3(256), 0, s, (const Fr *)abc, (const Fr *)(abc + m), (const Fr *)(abc + 2 * m),
      (uint32_t)d.n_cons, d.flag.get());
  HIP_CHECK(hipMemcpyAsync(d.h_flag, d.flag.get(), 4, hipMemcpyDeviceToHost, s)); HIP_CHECK(hipStreamSynchronize(s)); return *d.h_flag == 0;
}

}  // namespace zk
