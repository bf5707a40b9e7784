// HIP side of the prover: owns the device, the stream, and the kernel launches declared in gpu.hpp.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include "gpu.hpp"
#include "msm.cuh"
#include "ntt.cuh"
#include "keyops.cuh"

namespace zk {

#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw GpuError(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

class GpuContext {
 public:
  int device = 0; hipStream_t stream = nullptr; hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr}; hipEvent_t fork_event = nullptr; hipDeviceProp_t prop;
  GpuContext() {
    int n = 0; if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw GpuError("no HIP device visible: the prover's HIP path cannot run (there is no CPU fallback)");
    const char *e = getenv("ZK_DEVICE"); if (!e) e = getenv("LOCAL_RANK"); device = e ? atoi(e) % n : 0;
    HIP_CHECK(hipSetDevice(device)); HIP_CHECK(hipGetDeviceProperties(&prop, device)); HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; i++) HIP_CHECK(hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking)); HIP_CHECK(hipEventCreateWithFlags(&fork_event, hipEventDisableTiming));
  }
};
GpuContext &gpu() { static GpuContext ctx; hipSetDevice(ctx.device); return ctx; }
bool gpu_available() { int n = 0; return hipGetDeviceCount(&n) == hipSuccess && n > 0; }
void gpu_sync() { HIP_CHECK(hipStreamSynchronize(gpu().stream)); for (int i = 0; i < 4; i++) HIP_CHECK(hipStreamSynchronize(gpu().aux[i])); }
void gpu_fork_aux() { GpuContext &g = gpu(); HIP_CHECK(hipEventRecord(g.fork_event, g.stream)); for (int i = 0; i < 4; i++) HIP_CHECK(hipStreamWaitEvent(g.aux[i], g.fork_event, 0)); }
hipStream_t gpu_stream() { return gpu().stream; }

// ---- optional per-stage timing with HIP events on the compute stream (bench.py's roofline leg; off by default) ----------
struct StageTimer {
  struct Span { std::string name; hipEvent_t a, b; };
  bool enabled = false; std::vector<Span> open; std::vector<hipEvent_t> pool; std::map<std::string, std::pair<double, long>> acc;
  hipEvent_t get() { if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; } hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e; }
  struct Open { hipStream_t st; };
  std::vector<hipStream_t> open_streams;
  size_t begin(const char *name, hipStream_t st) { if (!enabled) return (size_t)-1; Span s{name, get(), get()}; HIP_CHECK(hipEventRecord(s.a, st)); open.push_back(s); open_streams.push_back(st); return open.size() - 1; }
  void end(size_t id) { if (id == (size_t)-1) return; HIP_CHECK(hipEventRecord(open[id].b, open_streams[id])); }
  void collect() { if (open.empty()) return; HIP_CHECK(hipDeviceSynchronize());
    for (Span &s : open) { float ms = 0; if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { auto &e = acc[s.name]; e.first += ms; e.second++; } pool.push_back(s.a); pool.push_back(s.b); } open.clear(); open_streams.clear(); }
};
static StageTimer g_timer;
struct Stage { size_t id; explicit Stage(const char *n, hipStream_t st = nullptr) : id(g_timer.begin(n, st ? st : gpu().stream)) {} ~Stage() { g_timer.end(id); } };
void profile_enable(bool on) { g_timer.collect(); g_timer.enabled = on; g_timer.acc.clear(); }
std::string profile_report() { g_timer.collect(); std::string o = "{"; bool first = true;
  for (auto &kv : g_timer.acc) { char buf[256]; snprintf(buf, sizeof buf, "%s\"%s\": {\"ms_total\": %.6f, \"count\": %ld}", first ? "" : ", ", kv.first.c_str(), kv.second.first, kv.second.second); o += buf; first = false; } return o + "}"; }

template <class T> DevBuf<T>::DevBuf(size_t n) : n_(n) { gpu(); if (n) HIP_CHECK(hipMalloc((void **)&p_, n * sizeof(T))); }
template <class T> DevBuf<T>::~DevBuf() { if (p_) hipFree(p_); }
template <class T> DevBuf<T>::DevBuf(DevBuf &&o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
template <class T> DevBuf<T> &DevBuf<T>::operator=(DevBuf &&o) noexcept { if (this != &o) { if (p_) hipFree(p_); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; } return *this; }
template <class T> void DevBuf<T>::upload(const T *h, size_t n) { HIP_CHECK(hipMemcpyAsync(p_, h, n * sizeof(T), hipMemcpyHostToDevice, gpu().stream)); HIP_CHECK(hipStreamSynchronize(gpu().stream)); }
template <class T> void DevBuf<T>::download(T *h, size_t n) const { HIP_CHECK(hipMemcpyAsync(h, p_, n * sizeof(T), hipMemcpyDeviceToHost, gpu().stream)); HIP_CHECK(hipStreamSynchronize(gpu().stream)); }
template <class T> void DevBuf<T>::zero() { if (n_) HIP_CHECK(hipMemsetAsync(p_, 0, n_ * sizeof(T), gpu().stream)); }
template class DevBuf<uint8_t>; template class DevBuf<uint32_t>; template class DevBuf<Fe32>; template class DevBuf<G1AffineRaw>; template class DevBuf<G2AffineRaw>;

template <class T> PinnedBuf<T>::PinnedBuf(size_t n) : n_(n) { gpu(); if (n) HIP_CHECK(hipHostMalloc((void **)&p_, n * sizeof(T))); }
template <class T> PinnedBuf<T>::~PinnedBuf() { release(); }
template <class T> void PinnedBuf<T>::release() { if (p_) hipHostFree(p_); p_ = nullptr; n_ = 0; }
template class PinnedBuf<Fe32>;
void upload_async(void *dev, const void *host, size_t bytes) { HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, gpu().stream)); }

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// exclusive scan of a uint32 array on the stream
struct Scanner {
  DevBuf<uint32_t> block_sums; size_t cap;
  explicit Scanner(size_t n) : block_sums(cdiv(n, SCAN_BLOCK * SCAN_ITEMS) + 1), cap(n) {}
  void run(const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
    unsigned nb = cdiv(n, SCAN_BLOCK * SCAN_ITEMS);
    hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, out, block_sums.get(), (uint32_t)n);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, block_sums.get(), nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_BLOCK), 0, s, out, block_sums.get(), (uint32_t)n);
  }
};

// ======================================================================================================================
// MSM
// ======================================================================================================================
template <class F, class RawAffine>
struct MsmImpl {
  size_t n; int c, W; uint32_t NB; bool filter_ones; uint32_t seg, n_ones_threads; std::string label = "msm"; int stream_id = -1;   // -1: main stream, 0..3: auxiliary stream
  DevBuf<RawAffine> points; DevBuf<uint8_t> inf; bool any_inf = false;
  DevBuf<uint32_t> hist, offsets, fill, entries, ones, ntasks, task_off; DevBuf<uint8_t> counters; Scanner scanner, task_scanner; uint32_t max_tasks;
  DevBuf<uint8_t> buckets, partials, seg_out, seg_l2, ones_partial, ones_l2, result;   // XYZZ<F> arrays, kept as bytes to stay out of the header
  XYZZ<F> *h_result = nullptr;                                      // pinned: W window sums + ones sum
  MsmCounters *h_cnt = nullptr;

  MsmImpl(const RawAffine *host_points, size_t n_, int c_, bool fo)
      : n(n_), c(c_), W(msm_num_windows(c_)), NB(1u << (c_ - 1)), filter_ones(fo), points(n_ ? n_ : 1), inf(n_ ? n_ : 1),
        hist((size_t)W * NB), offsets((size_t)W * NB), fill((size_t)W * NB), entries((n_ ? n_ : 1) * (size_t)W), ones(n_ ? n_ : 1), ntasks((size_t)W * NB + 1), task_off((size_t)W * NB + 1), counters(sizeof(MsmCounters)),
        scanner((size_t)W * NB), task_scanner((size_t)W * NB + 1) {
    if (c < 6 || c > 20 || W > MSM_MAX_WINDOWS) throw GpuError("msm: unsupported window size");
    seg = NB >= 4096 ? 8 : 4; n_ones_threads = 16384;
    std::vector<uint8_t> flags(n ? n : 1, 0); const uint8_t zero[sizeof(RawAffine)] = {0};
    for (size_t i = 0; i < n; i++) if (!memcmp(&host_points[i], zero, sizeof(RawAffine))) { flags[i] = 1; any_inf = true; }
    if (n) { points.upload(host_points, n); inf.upload(flags.data(), n); }
    max_tasks = (uint32_t)((n * (size_t)W) / MSM_TASK + (size_t)W * NB + 1);
    buckets = DevBuf<uint8_t>((size_t)W * NB * sizeof(XYZZ<F>)); partials = DevBuf<uint8_t>((size_t)max_tasks * sizeof(XYZZ<F>));
    seg_out = DevBuf<uint8_t>((size_t)W * (NB / seg) * sizeof(XYZZ<F>)); seg_l2 = DevBuf<uint8_t>((size_t)W * cdiv(NB / seg, 64) * sizeof(XYZZ<F>));
    ones_partial = DevBuf<uint8_t>((size_t)n_ones_threads * sizeof(XYZZ<F>)); ones_l2 = DevBuf<uint8_t>((size_t)(n_ones_threads / 64) * sizeof(XYZZ<F>));
    result = DevBuf<uint8_t>((size_t)(W + 1) * sizeof(XYZZ<F>));
    HIP_CHECK(hipHostMalloc((void **)&h_result, (size_t)(W + 1) * sizeof(XYZZ<F>))); HIP_CHECK(hipHostMalloc((void **)&h_cnt, sizeof(MsmCounters)));
  }
  ~MsmImpl() { if (h_result) hipHostFree(h_result); if (h_cnt) hipHostFree(h_cnt); }
  hipStream_t stream() { return stream_id < 0 ? gpu().stream : gpu().aux[stream_id & 3]; }

  void run(const Fe32 *scalars, const uint32_t *scalar_index) {
    hipStream_t s = stream(); size_t nbk = (size_t)W * NB; const uint8_t *infp = any_inf ? inf.get() : nullptr; MsmCounters *cnt = (MsmCounters *)counters.get();
    HIP_CHECK(hipMemsetAsync(hist.get(), 0, nbk * 4, s)); HIP_CHECK(hipMemsetAsync(fill.get(), 0, nbk * 4, s)); HIP_CHECK(hipMemsetAsync(cnt, 0, sizeof(MsmCounters), s));
    { Stage st((label + ".sort").c_str(), s);
      if (n) {
        hipLaunchKernelGGL(k_msm_classify<0>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const Fr *)scalars, scalar_index, infp, (uint32_t)n, c, W, (int)filter_ones, hist.get(), ones.get(), cnt);
        scanner.run(hist.get(), offsets.get(), nbk, s);
        hipLaunchKernelGGL(k_msm_scatter<0>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const Fr *)scalars, scalar_index, infp, (uint32_t)n, c, W, (int)filter_ones, offsets.get(), fill.get(), entries.get());
      }
      hipLaunchKernelGGL(k_msm_plan, dim3(cdiv(nbk + 1, 256)), dim3(256), 0, s, hist.get(), (uint32_t)nbk, ntasks.get());
      task_scanner.run(ntasks.get(), task_off.get(), nbk + 1, s);
    }
    { Stage st((label + ".accumulate").c_str(), s);
      hipLaunchKernelGGL((k_msm_accumulate_tasks<F>), dim3(cdiv(max_tasks, 256)), dim3(256), 0, s, (const Affine<F> *)points.get(), entries.get(), offsets.get(), hist.get(), task_off.get(), (uint32_t)nbk, max_tasks,
                         (XYZZ<F> *)buckets.get(), (XYZZ<F> *)partials.get());
    }
    { Stage st((label + ".combine").c_str(), s);
      hipLaunchKernelGGL((k_msm_combine_tasks<F, 8>), dim3(cdiv(nbk * 8, 256)), dim3(256), 0, s, hist.get(), task_off.get(), (uint32_t)nbk, (const XYZZ<F> *)partials.get(), (XYZZ<F> *)buckets.get());
    }
    { Stage st_red((label + ".reduce").c_str(), s);
      uint32_t spw = NB / seg, nseg = (uint32_t)W * spw;
      hipLaunchKernelGGL((k_msm_reduce_segments<F>), dim3(cdiv(nseg, 64)), dim3(64), 0, s, (const XYZZ<F> *)buckets.get(), NB, seg, nseg, (XYZZ<F> *)seg_out.get());
      if (spw > 64) { uint32_t g = cdiv(spw, 64);   // two-level tree per window keeps the dependent chain short
        hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(W * g), dim3(64), 0, s, (const XYZZ<F> *)seg_out.get(), 64u, (XYZZ<F> *)seg_l2.get());
        hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(W), dim3(64), 0, s, (const XYZZ<F> *)seg_l2.get(), g, (XYZZ<F> *)result.get());
      } else hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(W), dim3(64), 0, s, (const XYZZ<F> *)seg_out.get(), spw, (XYZZ<F> *)result.get());
    }
    XYZZ<F> *ones_dst = (XYZZ<F> *)result.get() + W;
    if (filter_ones && n) { Stage st((label + ".ones").c_str(), s);
      hipLaunchKernelGGL((k_msm_sum_ones<F>), dim3(cdiv(n_ones_threads, 256)), dim3(256), 0, s, (const Affine<F> *)points.get(), ones.get(), cnt, n_ones_threads, (XYZZ<F> *)ones_partial.get());
      hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(n_ones_threads / 64), dim3(64), 0, s, (const XYZZ<F> *)ones_partial.get(), 64u, (XYZZ<F> *)ones_l2.get());
      hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(1), dim3(64), 0, s, (const XYZZ<F> *)ones_l2.get(), n_ones_threads / 64, ones_dst);
    } else HIP_CHECK(hipMemsetAsync(ones_dst, 0, sizeof(XYZZ<F>), s));
    HIP_CHECK(hipMemcpyAsync(h_result, result.get(), (size_t)(W + 1) * sizeof(XYZZ<F>), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(h_cnt, cnt, sizeof(MsmCounters), hipMemcpyDeviceToHost, s));
  }
};

template <class HF> static HF load_hf(const void *p);
template <> host::HFq load_hf<host::HFq>(const void *p) { host::HFq r; memcpy(r.l, p, 32); return r; }
template <> host::HFq2 load_hf<host::HFq2>(const void *p) { host::HFq2 r; memcpy(r.c0.l, p, 32); memcpy(r.c1.l, (const char *)p + 32, 32); return r; }

// Horner combine of the window sums (c doublings per window) plus the ones-sum, on the host
template <class HF, class F> static host::HPoint<HF> combine(const XYZZ<F> *res, int W, int c) {
  auto get = [&](int i) { const char *b = (const char *)&res[i]; size_t fs = sizeof(F);
    return host::HPoint<HF>::from_xyzz(load_hf<HF>(b), load_hf<HF>(b + fs), load_hf<HF>(b + 2 * fs), load_hf<HF>(b + 3 * fs)); };
  host::HPoint<HF> acc = host::HPoint<HF>::inf();
  for (int w = W - 1; w >= 0; w--) { if (!acc.is_inf()) for (int i = 0; i < c; i++) acc = acc.dbl(); acc = acc.add(get(w)); }
  return acc.add(get(W));
}

struct MsmG1::Impl : MsmImpl<Fq, G1AffineRaw> { using MsmImpl::MsmImpl; };
struct MsmG2::Impl : MsmImpl<Fq2, G2AffineRaw> { using MsmImpl::MsmImpl; };
MsmG1::MsmG1(const G1AffineRaw *p, size_t n, int c, bool fo) : impl(new Impl(p, n, c, fo)) {}
MsmG1::~MsmG1() = default;
void MsmG1::run(const Fe32 *s, const uint32_t *idx) { impl->run(s, idx); }
void MsmG1::set_label(const char *l) { impl->label = l; }
void MsmG2::set_label(const char *l) { impl->label = l; }
void MsmG1::set_stream(int aux) { impl->stream_id = aux; }
void MsmG2::set_stream(int aux) { impl->stream_id = aux; }
host::HG1 MsmG1::result() { HIP_CHECK(hipStreamSynchronize(impl->stream())); return combine<host::HFq, Fq>(impl->h_result, impl->W, impl->c); }
size_t MsmG1::size() const { return impl->n; }
const G1AffineRaw *MsmG1::points_dev() const { return impl->points.get(); }
MsmG2::MsmG2(const G2AffineRaw *p, size_t n, int c, bool fo) : impl(new Impl(p, n, c, fo)) {}
MsmG2::~MsmG2() = default;
void MsmG2::run(const Fe32 *s, const uint32_t *idx) { impl->run(s, idx); }
host::HG2 MsmG2::result() { HIP_CHECK(hipStreamSynchronize(impl->stream())); return combine<host::HFq2, Fq2>(impl->h_result, impl->W, impl->c); }

// ======================================================================================================================
// Evaluation domains
// ======================================================================================================================
static size_t ceil_log2(size_t n) { size_t r = ((n & (n - 1)) == 0 ? 0 : 1); while (n > 1) { n >>= 1; r++; } return r; }   // FF/common/utils.cpp:32-45
using host::HFr;
static HFr fr_root_of_unity(size_t n) { HFr w; memcpy(w.l, FR_ROOT_OF_UNITY_2_28, 32); for (size_t i = 28; i > ceil_log2(n); --i) w = w.sqr(); return w; }   // field_utils.tcc:36-51
static HFr fr_coset_gen() { HFr g; memcpy(g.l, FR_COSET_GEN, 32); return g; }

// tables for one power-of-two transform size n with root w: tw[j] = w^j, itw[j] = w^-j (j < n/2)
struct Radix2Tables {
  int logn; size_t n; DevBuf<Fe32> tw, itw;
  Radix2Tables(size_t n_, const HFr &w) : logn((int)ceil_log2(n_)), n(n_), tw(n_ / 2 ? n_ / 2 : 1), itw(n_ / 2 ? n_ / 2 : 1) {
    std::vector<Fe32> a(n / 2 ? n / 2 : 1), b(a.size()); HFr wi = w.inv(), x = HFr::one(), y = HFr::one();
    for (size_t j = 0; j < n / 2; j++) { memcpy(&a[j], x.l, 32); memcpy(&b[j], y.l, 32); x = x * w; y = y * wi; }
    if (n / 2 == 0) { memcpy(&a[0], x.l, 32); memcpy(&b[0], y.l, 32); }
    tw.upload(a.data(), a.size()); itw.upload(b.data(), b.size());
  }
};
static std::vector<Fe32> geometric_table(size_t n, const HFr &first, const HFr &ratio) { std::vector<Fe32> t(n); HFr x = first; for (size_t i = 0; i < n; i++) { memcpy(&t[i], x.l, 32); x = x * ratio; } return t; }

// in-place radix-2 transform of `batch` vectors: out = DIT(bitrev(in * pre_scale)); scratch holds the permuted copy
static void radix2_transform(Fe32 *data, Fe32 *scratch, const Fe32 *tw, int logn, const Fe32 *pre_scale, int batch, size_t stride, size_t scratch_stride) {
  hipStream_t s = gpu().stream; size_t n = (size_t)1 << logn;
  hipLaunchKernelGGL(k_ntt_bitrev_scale, dim3(cdiv(n, 256), batch), dim3(256), 0, s, (const Fr *)data, (Fr *)scratch, (const Fr *)pre_scale, logn, stride, scratch_stride);
  int L = logn < NTT_LOCAL_LOG ? logn : NTT_LOCAL_LOG;
  if (L > 0) hipLaunchKernelGGL(k_ntt_local, dim3((unsigned)(n >> L), batch), dim3(NTT_LOCAL_THREADS), sizeof(Fr) << L, s, (Fr *)scratch, (const Fr *)tw, logn, L, scratch_stride);
  for (int st = L + 1; st <= logn; st++) hipLaunchKernelGGL(k_ntt_stage, dim3(cdiv(n / 2, 256), batch), dim3(256), 0, s, (Fr *)scratch, (const Fr *)tw, logn, st, scratch_stride);
  for (int b = 0; b < batch; b++) HIP_CHECK(hipMemcpyAsync(data + b * stride, scratch + b * scratch_stride, n * sizeof(Fe32), hipMemcpyDeviceToDevice, s));
}

// ---- step-radix-2 helper kernels (domains/step_radix2_domain.tcc:39-153) ---------------------------------------------
// forward pre-pass: c[i] = a[i] + a[i+B] (i<S) else a[i];  d[i] = w^i * (a[i] - a[i+B] (i<S) else a[i]);  e[i] = sum_j d[i + j*S]
__global__ void k_step_fwd_pre(const Fr *__restrict__ a, Fr *__restrict__ cbuf, Fr *__restrict__ dbuf, const Fr *__restrict__ wpow, uint32_t B, uint32_t S) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= B) return; Fr x = a[i];
  if (i < S) { Fr y = a[i + B]; cbuf[i] = x + y; dbuf[i] = wpow[i] * (x - y); } else { cbuf[i] = x; dbuf[i] = wpow[i] * x; }
}
__global__ void k_step_fold(const Fr *__restrict__ dbuf, Fr *__restrict__ e, uint32_t B, uint32_t S) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= S) return; Fr acc = Fr::zero(); for (uint32_t j = i; j < B; j += S) acc = acc + dbuf[j]; e[i] = acc;
}
// inverse post-pass.  U0 (B values, already scaled by 1/B), U1 (S values, scaled by 1/S):
//   tmp[i] = U0[i]*w^i ; U1[i] -= sum_{j>=1} tmp[i + j*S] ; U1[i] *= w^-i ; a[i] = (U0[i]+U1[i])/2 (i<S) ; a[B+i] = (U0[i]-U1[i])/2 ; a[i] = U0[i] (S<=i<B)
__global__ void k_step_inv_post(const Fr *__restrict__ U0, const Fr *__restrict__ U1, Fr *__restrict__ a, const Fr *__restrict__ wpow, const Fr *__restrict__ winvpow, Fr half, uint32_t B, uint32_t S) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= B) return;
  if (i >= S) { a[i] = U0[i]; return; }
  Fr u1 = U1[i]; for (uint32_t j = i + S; j < B; j += S) u1 = u1 - U0[j] * wpow[j];
  u1 = u1 * winvpow[i]; Fr u0 = U0[i]; a[i] = (u0 + u1) * half; a[B + i] = (u0 - u1) * half;
}

struct Domain::Impl {
  size_t m = 0; bool step = false; size_t B = 0, S = 0;           // step: m = B + S
  std::unique_ptr<Radix2Tables> big, small;                       // basic: only `big` (size m)
  DevBuf<Fe32> inv_n_big, inv_n_small;                            // constant tables 1/n folded into pre-scales where possible
  DevBuf<Fe32> coset_fwd, coset_inv, zinv, wpow, winvpow, scratch;
  DevBuf<Fe32> scale_big, scale_small;                            // 1/B, 1/S as m-long constant tables are avoided: k_fr_mul_table with per-element tables below
  HFr half;
  size_t scratch_stride = 0;
};

Domain::Domain(size_t min_size) : impl(new Impl) {
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
    d.scale_big = DevBuf<Fe32>(d.m); d.scale_big.upload(sc.data(), d.m);
    HFr z = (g.pow_u64(d.m) - one).inv(); Fe32 zz; memcpy(&zz, z.l, 32); d.zinv = DevBuf<Fe32>(1); d.zinv.upload(&zz, 1);       // divide_by_Z_on_coset :103-112
  } else {
    HFr w = fr_root_of_unity((size_t)1 << ceil_log2(d.m)), wb = w.sqr(), ws = fr_root_of_unity(d.S), winv = w.inv();   // step_radix2_domain.tcc:20-37
    d.big.reset(new Radix2Tables(d.B, wb)); d.small.reset(new Radix2Tables(d.S, ws));
    auto cf = geometric_table(d.m, one, g), ci = geometric_table(d.m, one, ginv), wp = geometric_table(d.B, one, w), wip = geometric_table(d.S, one, winv);
    auto sb = geometric_table(d.B, HFr::from_u64(d.B).inv(), one), ss = geometric_table(d.S, HFr::from_u64(d.S).inv(), one);
    d.coset_fwd = DevBuf<Fe32>(d.m); d.coset_fwd.upload(cf.data(), d.m); d.coset_inv = DevBuf<Fe32>(d.m); d.coset_inv.upload(ci.data(), d.m);
    d.wpow = DevBuf<Fe32>(d.B); d.wpow.upload(wp.data(), d.B); d.winvpow = DevBuf<Fe32>(d.S); d.winvpow.upload(wip.data(), d.S);
    d.scale_big = DevBuf<Fe32>(d.B); d.scale_big.upload(sb.data(), d.B); d.scale_small = DevBuf<Fe32>(d.S); d.scale_small.upload(ss.data(), d.S);
    // divide_by_Z_on_coset (:242-260): P[i] /= (g^S * Z0 * w^(2S i) - w^S * Z0) for i < B ; P[B+i] /= Z1
    std::vector<Fe32> zt(d.m); HFr Z0 = g.pow_u64(d.B) - one, cSZ0 = g.pow_u64(d.S) * Z0, wS = w.pow_u64(d.S), wSZ0 = wS * Z0, w2S = w.pow_u64(2 * d.S), elt = one;
    // batch inversion of the B denominators
    std::vector<HFr> den(d.B), pre(d.B); for (size_t i = 0; i < d.B; i++) { den[i] = cSZ0 * elt - wSZ0; elt = elt * w2S; }
    HFr acc = one; for (size_t i = 0; i < d.B; i++) { pre[i] = acc; acc = acc * den[i]; } HFr ai = acc.inv();
    for (size_t i = d.B; i-- > 0;) { HFr v = ai * pre[i]; ai = ai * den[i]; memcpy(&zt[i], v.l, 32); }
    HFr cw = g * w, Z1 = ((cw.pow_u64(d.B) - one) * (cw.pow_u64(d.S) - wS)).inv(); for (size_t i = 0; i < d.S; i++) memcpy(&zt[d.B + i], Z1.l, 32);
    d.zinv = DevBuf<Fe32>(d.m); d.zinv.upload(zt.data(), d.m);
  }
  d.scratch_stride = d.m; d.scratch = DevBuf<Fe32>(3 * d.m);
}
Domain::~Domain() = default;
size_t Domain::m() const { return impl->m; }
bool Domain::is_step() const { return impl->step; }

static void mul_table(Fe32 *a, const Fe32 *t, size_t n, int batch, size_t stride) { hipLaunchKernelGGL(k_fr_mul_table, dim3(cdiv(n, 256), batch), dim3(256), 0, gpu().stream, (Fr *)a, (const Fr *)t, (uint32_t)n, stride); }

void Domain::fft(Fe32 *data, int batch, size_t stride) {
  Stage st("ntt.forward"); Impl &d = *impl; if (batch > 3) throw GpuError("domain: batch > 3");
  if (!d.step) { radix2_transform(data, d.scratch.get(), d.big->tw.get(), d.big->logn, nullptr, batch, stride, d.scratch_stride); return; }
  hipStream_t s = gpu().stream;
  for (int b = 0; b < batch; b++) {   // scratch layout per vector: [c (B) | e (S)] in slot 0, d (B) in slot 1, bitrev scratch in slot 2
    Fe32 *a = data + b * stride, *cb = d.scratch.get(), *db = d.scratch.get() + d.m, *tmp = d.scratch.get() + 2 * d.m, *e = cb + d.B;
    hipLaunchKernelGGL(k_step_fwd_pre, dim3(cdiv(d.B, 256)), dim3(256), 0, s, (const Fr *)a, (Fr *)cb, (Fr *)db, (const Fr *)d.wpow.get(), (uint32_t)d.B, (uint32_t)d.S);
    hipLaunchKernelGGL(k_step_fold, dim3(cdiv(d.S, 256)), dim3(256), 0, s, (const Fr *)db, (Fr *)e, (uint32_t)d.B, (uint32_t)d.S);
    radix2_transform(cb, tmp, d.big->tw.get(), d.big->logn, nullptr, 1, 0, 0); radix2_transform(e, tmp, d.small->tw.get(), d.small->logn, nullptr, 1, 0, 0);
    HIP_CHECK(hipMemcpyAsync(a, cb, d.m * sizeof(Fe32), hipMemcpyDeviceToDevice, s));
  }
}
void Domain::ifft(Fe32 *data, int batch, size_t stride) {
  Stage st("ntt.inverse"); Impl &d = *impl; if (batch > 3) throw GpuError("domain: batch > 3");
  if (!d.step) { radix2_transform(data, d.scratch.get(), d.big->itw.get(), d.big->logn, nullptr, batch, stride, d.scratch_stride); mul_table(data, d.scale_big.get(), d.m, batch, stride); return; }
  hipStream_t s = gpu().stream;
  for (int b = 0; b < batch; b++) {
    Fe32 *a = data + b * stride, *U = d.scratch.get(), *tmp = d.scratch.get() + 2 * d.m;
    HIP_CHECK(hipMemcpyAsync(U, a, d.m * sizeof(Fe32), hipMemcpyDeviceToDevice, s));
    radix2_transform(U, tmp, d.big->itw.get(), d.big->logn, nullptr, 1, 0, 0); radix2_transform(U + d.B, tmp, d.small->itw.get(), d.small->logn, nullptr, 1, 0, 0);
    mul_table(U, d.scale_big.get(), d.B, 1, 0); mul_table(U + d.B, d.scale_small.get(), d.S, 1, 0);
    Fr half; memcpy(&half, d.half.l, 32);
    hipLaunchKernelGGL(k_step_inv_post, dim3(cdiv(d.B, 256)), dim3(256), 0, s, (const Fr *)U, (const Fr *)(U + d.B), (Fr *)a, (const Fr *)d.wpow.get(), (const Fr *)d.winvpow.get(), half, (uint32_t)d.B, (uint32_t)d.S);
  }
}
void Domain::coset_fft(Fe32 *data, int batch, size_t stride) { mul_table(data, impl->coset_fwd.get(), impl->m, batch, stride); fft(data, batch, stride); }
void Domain::icoset_fft(Fe32 *data, int batch, size_t stride) {
  Impl &d = *impl;
  if (!d.step) { Stage st("ntt.inverse"); radix2_transform(data, d.scratch.get(), d.big->itw.get(), d.big->logn, nullptr, batch, stride, d.scratch_stride); mul_table(data, d.coset_inv.get(), d.m, batch, stride); return; }   // coset_inv carries 1/m
  ifft(data, batch, stride); mul_table(data, d.coset_inv.get(), d.m, batch, stride);
}
void Domain::qap_pointwise(Fe32 *a, const Fe32 *b, const Fe32 *c) {
  hipLaunchKernelGGL(k_qap_pointwise, dim3(cdiv(impl->m, 256)), dim3(256), 0, gpu().stream, (Fr *)a, (const Fr *)b, (const Fr *)c, (const Fr *)impl->zinv.get(), impl->step ? 1 : 0, (uint32_t)impl->m);
}

void fr_to_mont_dev(Fe32 *a, size_t n) { if (n) hipLaunchKernelGGL(k_fr_to_mont, dim3(cdiv(n, 256)), dim3(256), 0, gpu().stream, (Fr *)a, (uint32_t)n); }
void fr_from_mont_dev(Fe32 *a, size_t n) { if (n) hipLaunchKernelGGL(k_fr_from_mont, dim3(cdiv(n, 256)), dim3(256), 0, gpu().stream, (Fr *)a, (uint32_t)n); }

// ======================================================================================================================
// R1CS rows
// ======================================================================================================================
struct R1csDev::Impl {
  size_t n_inputs, n_vars, n_cons; DevBuf<uint32_t> rowptr[3], col[3], cid[3]; DevBuf<Fe32> ctab; DevBuf<uint32_t> flag; uint32_t *h_flag = nullptr;
  ~Impl() { if (h_flag) hipHostFree(h_flag); }
};
R1csDev::R1csDev(const R1csHost &h) : impl(new Impl) {
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
  }
  d.ctab = DevBuf<Fe32>(tab.size()); d.ctab.upload(tab.data(), tab.size()); d.flag = DevBuf<uint32_t>(1); HIP_CHECK(hipHostMalloc((void **)&d.h_flag, 4));
}
R1csDev::~R1csDev() = default;
void R1csDev::eval(const Fe32 *z, Fe32 *abc, size_t m) {
  Stage st("r1cs.rows"); Impl &d = *impl; hipStream_t s = gpu().stream; if (m < d.n_cons + d.n_inputs + 1) throw GpuError("r1cs: domain too small");
  HIP_CHECK(hipMemsetAsync(abc, 0, 3 * m * sizeof(Fe32), s));
  for (int mm = 0; mm < 3; mm++) if (d.n_cons) hipLaunchKernelGGL(k_r1cs_rows, dim3(cdiv(d.n_cons, 256)), dim3(256), 0, s, d.rowptr[mm].get(), d.col[mm].get(), d.cid[mm].get(), (const Fr *)d.ctab.get(), (const Fr *)z, (uint32_t)d.n_cons, (Fr *)(abc + mm * m));
  HIP_CHECK(hipMemcpyAsync(abc + d.n_cons, z, (d.n_inputs + 1) * sizeof(Fe32), hipMemcpyDeviceToDevice, s));   // input-consistency rows (r1cs_to_qap.tcc:227-230)
}
void R1csDev::check_async(const Fe32 *abc, size_t m) {
  Impl &d = *impl; hipStream_t s = gpu().stream; HIP_CHECK(hipMemsetAsync(d.flag.get(), 0, 4, s));
  if (d.n_cons) hipLaunchKernelGGL(k_r1cs_check, dim3(cdiv(d.n_cons, 256)), dim3(256), 0, s, (const Fr *)abc, (const Fr *)(abc + m), (const Fr *)(abc + 2 * m), (uint32_t)d.n_cons, d.flag.get());
  HIP_CHECK(hipMemcpyAsync(d.h_flag, d.flag.get(), 4, hipMemcpyDeviceToHost, s));
}
bool R1csDev::check_result() const { return *impl->h_flag == 0; }
bool R1csDev::satisfied(const Fe32 *abc, size_t m) {
  Impl &d = *impl; hipStream_t s = gpu().stream; d.flag.zero();
  if (d.n_cons) hipLaunchKernelGGL(k_r1cs_check, dim3(cdiv(d.n_cons, 256)), dim3(256), 0, s, (const Fr *)abc, (const Fr *)(abc + m), (const Fr *)(abc + 2 * m), (uint32_t)d.n_cons, d.flag.get());
  HIP_CHECK(hipMemcpyAsync(d.h_flag, d.flag.get(), 4, hipMemcpyDeviceToHost, s)); HIP_CHECK(hipStreamSynchronize(s)); return *d.h_flag == 0;
}

// ======================================================================================================================
// Key-side operations
// ======================================================================================================================
static FqPowConsts fq_pow_consts() {
  FqPowConsts pc; auto shr = [](const uint32_t *a, int k, uint32_t *o) { for (int i = 0; i < 8; i++) o[i] = (a[i] >> k) | (i < 7 ? a[i + 1] << (32 - k) : 0); };
  uint32_t t[8]; uint64_t c = 1; for (int i = 0; i < 8; i++) { c += FqParams::MOD[i]; t[i] = (uint32_t)c; c >>= 32; } shr(t, 2, pc.sqrt_exp);     // (q+1)/4
  uint64_t b = 3; for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)FqParams::MOD[i] - b; t[i] = (uint32_t)d; b = (d >> 32) & 1; } shr(t, 2, pc.qm3o4);   // (q-3)/4
  b = 1; for (int i = 0; i < 8; i++) { uint64_t d = (uint64_t)FqParams::MOD[i] - b; t[i] = (uint32_t)d; b = (d >> 32) & 1; } shr(t, 1, pc.qm1o2);            // (q-1)/2
  return pc;
}
static void check_bad(DevBuf<uint32_t> &bad, const char *what) { uint32_t h = 0; bad.download(&h, 1); if (h) throw GpuError(std::string(what) + ": " + std::to_string(h) + " x-coordinates are not on the curve"); }
void decompress_g1(const Fe32 *xs, const uint8_t *flags, size_t n, G1AffineRaw *out) {
  if (!n) return; DevBuf<Fe32> dx(n); DevBuf<uint8_t> df(n); DevBuf<G1AffineRaw> dout(n); DevBuf<uint32_t> bad(1); dx.upload(xs, n); df.upload(flags, n); bad.zero();
  hipLaunchKernelGGL(k_g1_decompress, dim3(cdiv(n, 128)), dim3(128), 0, gpu().stream, (const Fq *)dx.get(), df.get(), (Affine<Fq> *)dout.get(), (uint32_t)n, fq_pow_consts(), bad.get());
  HIP_CHECK(hipGetLastError()); dout.download(out, n); check_bad(bad, "G1 decompression");
}
void decompress_g2(const Fe32 *xs, const uint8_t *flags, size_t n, G2AffineRaw *out) {
  if (!n) return; DevBuf<Fe32> dx(2 * n); DevBuf<uint8_t> df(n); DevBuf<G2AffineRaw> dout(n); DevBuf<uint32_t> bad(1); dx.upload(xs, 2 * n); df.upload(flags, n); bad.zero();
  host::HFq2 tb = host::HFq2{host::HFq::from_u64(3), host::HFq::zero()} * host::HFq2{host::HFq::from_u64(9), host::HFq::one()}.inv();   // 3 / (9 + u)  (alt_bn128_init.cpp:193)
  Fq2 twist_b; memcpy(&twist_b.c0, tb.c0.l, 32); memcpy(&twist_b.c1, tb.c1.l, 32);
  hipLaunchKernelGGL(k_g2_decompress, dim3(cdiv(n, 64)), dim3(64), 0, gpu().stream, (const Fq2 *)dx.get(), df.get(), (Affine<Fq2> *)dout.get(), (uint32_t)n, fq_pow_consts(), twist_b, bad.get());
  HIP_CHECK(hipGetLastError()); dout.download(out, n); check_bad(bad, "G2 decompression");
}
template <class HF, class Raw> static void store_affine(const host::HPoint<HF> &p, Raw &o) { HF x, y; p.to_affine(x, y); memcpy(&o, &x, sizeof(HF)); memcpy((char *)&o + sizeof(HF), &y, sizeof(HF)); }
template <class HF, class F, class Raw> static void fixed_base_mul(const host::HPoint<HF> &base, const Fe32 *scalars, size_t n, Raw *out) {
  if (!n) return; std::vector<Raw> table(32 * 255); host::HPoint<HF> wbase = base;
  for (int w = 0; w < 32; w++) { host::HPoint<HF> acc = wbase; for (int d = 1; d <= 255; d++) { store_affine(acc, table[w * 255 + d - 1]); acc = acc.add(wbase); } wbase = acc; }   // after 255 additions acc = 256 * wbase
  DevBuf<Raw> dt(table.size()), dout(n); DevBuf<Fe32> ds(n); dt.upload(table.data(), table.size()); ds.upload(scalars, n);
  hipLaunchKernelGGL((k_fixed_base_mul<F>), dim3(cdiv(n, 128)), dim3(128), 0, gpu().stream, (const Affine<F> *)dt.get(), (const Fr *)ds.get(), (Affine<F> *)dout.get(), (uint32_t)n);
  HIP_CHECK(hipGetLastError()); dout.download(out, n);
}
void fixed_base_mul_g1(const host::HG1 &base, const Fe32 *scalars, size_t n, G1AffineRaw *out) { fixed_base_mul<host::HFq, Fq, G1AffineRaw>(base, scalars, n, out); }
void fixed_base_mul_g2(const host::HG2 &base, const Fe32 *scalars, size_t n, G2AffineRaw *out) { fixed_base_mul<host::HFq2, Fq2, G2AffineRaw>(base, scalars, n, out); }

}  // namespace zk
