// Host-side driver of the MSM kernels, shared by the G1 and G2 translation units.
#pragma once
#include <cstring>
#include <memory>
#include "gpu_internal.hpp"

// window sizes with a compiled-in digit walk (msm.cuh: for_each_digit); every other size takes the runtime path (C = 0)
#define ZK_MSM_DISPATCH_C(cval, glvval, CALL) do { if (!(glvval) && (cval) == 8) { CALL(8); } else if (!(glvval) && (cval) == 16) { CALL(16); } else { CALL(0); } } while (0)

namespace zk {
// what k_wsort leaves behind (msm.cuh, "witness MSMs in three launches"): one set per scalar vector, shared by the MSMs over that vector (A and L*; B1 and B2 — different
// curve groups, hence no template parameter here).  The leader runs the sort; a follower queued behind it on the same stream, or waiting for `sorted`, only accumulates.
struct WsortBuffers { uint32_t NB = 0, cap = 0; size_t n = 0; DevBuf<uint32_t> fill /* 2 x NB: the runs alternate */, entries /* NB x cap */, ones /* n */, counters /* 2 MsmCounters */; int parity = 0; bool shared = false; hipEvent_t sorted = nullptr; int leader_stream = -1;
  ~WsortBuffers() { if (sorted) hipEventDestroy(sorted); } };
template <class F, class RawAffine>
struct MsmImpl {
  size_t n; int c, W, WB; uint32_t NB;   // W digit windows; WB bucket arrays (1 when the multiples 2^(cw) P are precomputed, else W)
   bool filter_ones; uint32_t seg, n_ones_quads; std::string label = "msm"; int stream_id = -1;   // -1: main stream, 0..3: auxiliary stream
  // the key's points (with the fixed-base table when there is one) are immutable and shared by every prover object of the key on this device; everything else below is per-object workspace
  struct Bases { size_t n = 0; int c = 0, W = 0, WB = 0; bool glv = false, any_inf = false; DevBuf<RawAffine> points, points261 /* the same table with coordinates x 2^261: what k_hacc_runs29 gathers from (G1, uniform scalars, fixed-base table) */; DevBuf<uint8_t> inf; };
  std::shared_ptr<const Bases> bases; const DevBuf<RawAffine> &points; const DevBuf<uint8_t> &inf; bool any_inf = false;
  bool direct = false, offsets_direct = false; uint32_t cap = 0, task = MSM_TASK;   // task: sorted entries per accumulation lane
  bool split_ones = false; hipStream_t ones_stream = nullptr; hipEvent_t ev_classified = nullptr, ev_ones = nullptr;   // the ones path on a stream of its own, beside the bucket path (the G2 MSM: both are long chains)
  const Fe32 *prod_b = nullptr, *prod_z = nullptr; bool prod_z_table = false; DevBuf<Fe32> prod_tmp;   // scalars given as a product a*b*z (run_product)
  uint32_t h_slices = [] { const char *e = getenv("ZK_MSM_H_SLICES"); int v = e ? atoi(e) : 12; return (uint32_t)(v < 1 ? 1 : v > 64 ? 64 : v); }();   // slices per bucket of the H query's accumulation (measured, accumulate + combine inside a proof: 5: 0.66, 6: 0.69, 8: 0.67, 12: 0.64, 16: 0.655, 32: 0.74, 64: 1.0 ms)
  uint32_t h_combine_lq = [] { const char *e = getenv("ZK_MSM_H_COMBINE_LQ"); return (uint32_t)(e ? atoi(e) & 3 : 0); }();      // 2^lq quads per bucket in the combine (slices must be divisible by it)
  bool wfused = false, wacc_quads = false, ws_leader = true, overflow_noted = false; std::shared_ptr<WsortBuffers> ws;   // witness MSMs in three launches (k_wsort / k_wacc / k_wtail); needs the fixed-base tables and at most 128 buckets
  bool sparse = false; DevBuf<uint8_t> others; uint32_t others_cap = 0;   // witness MSMs without buckets (k_wmsm_classify / k_wmsm_sum, msm.cuh): needs the fixed-base tables
  bool hsort = false; HsortShape hs{0, 0, 0, 0}; DevBuf<uint32_t> group_fill, mid, group_n; bool hruns = false; uint32_t h_run = 16, h_maxp = 16;   // hruns: accumulation over fixed-length runs (k_hacc_runs), h_run entries per lane, at most h_maxp pieces per bucket
    // group-binned one-pass sort (k_hsort_bin / k_hsort_group, msm.cuh)
  const Fe32 *last_scalars = nullptr; const uint32_t *last_index = nullptr;   // one-pass sort (k_msm_scatter_direct) for uniform scalars
  DevBuf<uint32_t> zeroed;                                          // [hist | fill | counters]: cleared by one memset per run
  DevBuf<uint32_t> offsets, entries, ones, ntasks, task_off, order, rank_of, block_hist, block_off, cls_start; uint32_t bsort_blocks; std::unique_ptr<Scanner> bsort_scanner; Scanner scanner, task_scanner; uint32_t max_tasks;
  DevBuf<uint32_t> lane_off;   // fused witness path: where each bucket's lanes start
  DevBuf<uint8_t> buckets, partials, seg_out, seg_l2, ones_partial, ones_l2, result;   // XYZZ<F> arrays, kept as bytes to stay out of the header; result = W window sums, the ones sum, the counters
  uint8_t *h_result = nullptr, *h_result_dev = nullptr;             // pinned host memory the last kernels of an MSM write their sums into (device address of the same pages); only under ZK_MSM_MAPPED_RESULT=1, otherwise `result` is copied
  // Host tail (optional, ZK_MSM_HOST_TAIL=1; one bucket array, i.e. fixed-base tables): the last, purely dependent additions of an MSM done by the host, which idles
  // while the device works (the MSM's submit thread does them, groth16.cpp).  Frees the device of its slowest tiny kernels, but is no faster (see the constructor).  tail = [T0: 256 sums | T1: 64 partial sums of the scalar-one path | counters].  Witness MSMs (at most
  // 256 buckets): T0 IS the bucket array and the host does the weighted running-sum reduction (2 * 128 additions); H query: T0 holds the sums of 256 segments each.
  static constexpr uint32_t TAIL_T0 = 256, TAIL_T1 = 64; int tail_mode = 0; uint32_t tail_n0 = 0, tail_n1 = 0; DevBuf<uint8_t> tail; uint8_t *h_tail = nullptr;   // mode 1: weighted buckets, mode 2: plain sums
  static size_t tail_bytes() { return (size_t)(TAIL_T0 + TAIL_T1) * sizeof(XYZZ<F>) + sizeof(MsmCounters); }
  XYZZ<F> *bucket_array() { return tail_mode == 1 ? (XYZZ<F> *)tail.get() : reinterpret_cast<XYZZ<F> *>(buckets.get()); }
  const XYZZ<F> *host_tail0() const { return (const XYZZ<F> *)h_tail; } const XYZZ<F> *host_tail1() const { return (const XYZZ<F> *)h_tail + TAIL_T0; }
  static constexpr uint32_t HEAVY_BLOCKS = 256, GROUP = 256;

  void share_sort(const std::shared_ptr<WsortBuffers> &leader_ws) { if (!wfused || !leader_ws || leader_ws->NB != NB || leader_ws->n != n) throw GpuError("msm: this MSM cannot share the sort (different size or path)"); ws = leader_ws; ws_leader = false; ws->shared = true; }
  uint32_t *hist() { return zeroed.get(); }
  uint32_t *fill() { return zeroed.get() + (size_t)WB * NB; }
  int parity = 0;                                                   // which of the two counter slots the current run uses
  MsmCounters *counters() { return (MsmCounters *)(zeroed.get() + 2 * (size_t)WB * NB) + parity; }
  MsmCounters *counters_next() { return (MsmCounters *)(zeroed.get() + 2 * (size_t)WB * NB) + (parity ^ 1); }
  int RS = 0;   // result slots before the ones slot: WB window sums, or (bitsum) log2(NB) + 1 sums by weight bit
  bool bitsum = false;
  size_t result_bytes() const { return (size_t)(RS + 1) * sizeof(XYZZ<F>) + sizeof(MsmCounters); }
  const XYZZ<F> *host_sums() const { return (const XYZZ<F> *)h_result; }
  const MsmCounters *host_counters() const { return tail_mode ? (const MsmCounters *)(h_tail + (size_t)(TAIL_T0 + TAIL_T1) * sizeof(XYZZ<F>)) : (const MsmCounters *)(h_result + (size_t)(RS + 1) * sizeof(XYZZ<F>)); }

  // Precomputed multiples 2^(cw) P unless switched off (ZK_MSM_PRECOMPUTE=0), the table would pass ZK_MSM_PRECOMPUTE_MAX_MB (default 768 MB: measured on MI355X, the
  // random 64-byte gathers from a table far beyond the 256 MB Infinity Cache cost more than the smaller bucket reduction saves — deposit at depth 32: 9.1 ms
  // without, 13.1 ms with 2-9 GB tables; send / mint / deposit-8 gain 1-5 %), or its indices would not fit the 31 bits of a sorted entry.
  static bool use_precompute(size_t n_, int W_) { static const bool on = [] { const char *e = getenv("ZK_MSM_PRECOMPUTE"); return !e || atoi(e) != 0; }();
    static const size_t cap = [] { const char *e = getenv("ZK_MSM_PRECOMPUTE_MAX_MB"); return (size_t)(e ? atol(e) : 768) << 20; }();
    return on && n_ > 0 && W_ > 1 && n_ * (size_t)W_ < (1ull << 31) && n_ * (size_t)W_ * sizeof(RawAffine) <= cap; }
  static int windows_for(int c_, bool glv_) { return glv_ ? 132 / c_ + 1 : msm_num_windows(c_); }   // GLV halves are below 2^128 (measured bound 2^127; four spare bits)
  static bool glv_possible(size_t n_, int c_, bool fo, bool tables, bool uniform_hint, bool glv_hint) { return glv_hint && uniform_hint && tables && !fo && sizeof(F) == 32 && use_precompute(n_, windows_for(c_, true)) && getenv("ZK_MSM_NO_GLV") == nullptr; }
  bool glv = false; DevBuf<Fe32> beta;   // GLV: two half-length scalars per point, the second half addresses lambda*P = (beta*x, y)
  static std::shared_ptr<const Bases> make_bases(const RawAffine *host_points, size_t n_, int c_, bool fo, bool tables, bool uniform_hint, bool glv_hint) {
    auto b = std::make_shared<Bases>(); b->n = n_; b->c = c_; b->glv = glv_possible(n_, c_, fo, tables, uniform_hint, glv_hint); b->W = windows_for(c_, b->glv); b->WB = tables && use_precompute(n_, b->W) ? 1 : b->W;
    if (c_ < 6 || c_ > 20 || b->W > MSM_MAX_WINDOWS) throw GpuError("msm: unsupported window size");
    if (b->glv && (b->WB != 1 || n_ * (size_t)b->W >= (1ull << 30))) { b->glv = false; b->W = windows_for(c_, false); b->WB = tables && use_precompute(n_, b->W) ? 1 : b->W; }
    b->points = DevBuf<RawAffine>((n_ ? n_ : 1) * (size_t)(b->WB == 1 ? b->W : 1)); b->inf = DevBuf<uint8_t>(n_ ? n_ : 1);
    std::vector<uint8_t> flags(n_ ? n_ : 1, 0); const uint8_t zero[sizeof(RawAffine)] = {0};
    for (size_t i = 0; i < n_; i++) if (!memcmp(&host_points[i], zero, sizeof(RawAffine))) { flags[i] = 1; b->any_inf = true; }
    if (n_) { b->points.upload(host_points, n_); b->inf.upload(flags.data(), n_); }
    if (b->WB == 1 && b->W > 1 && n_) {   // table[w*n + i] = 2^(c*w) * P_i (k_msm_precompute); the scratch arrays live only for this launch
      DevBuf<uint8_t> tmp((size_t)(b->W - 1) * n_ * sizeof(XYZZ<F>)), pref((size_t)(b->W - 1) * n_ * sizeof(F));
      hipLaunchKernelGGL((k_msm_precompute<F>), dim3(cdiv(n_, 64)), dim3(64), 0, gpu().stream, (Affine<F> *)b->points.get(), (uint32_t)n_, c_, b->W, (XYZZ<F> *)tmp.get(), (F *)pref.get());
      HIP_CHECK(hipGetLastError()); HIP_CHECK(hipStreamSynchronize(gpu().stream));
    }
    if constexpr (sizeof(F) == 32) if (uniform_hint && !fo && b->WB == 1 && n_ && !b->glv && !(getenv("ZK_MSM_HACC") && strcmp(getenv("ZK_MSM_HACC"), "runs29"))) { const size_t tn = n_ * (size_t)b->W; b->points261 = DevBuf<RawAffine>(tn);
      hipLaunchKernelGGL(k_table_to_r261, dim3(cdiv(tn, 256)), dim3(256), 0, gpu().stream, (const Affine<Fq> *)b->points.get(), (Affine<Fq> *)b->points261.get(), tn); HIP_CHECK(hipGetLastError()); HIP_CHECK(hipStreamSynchronize(gpu().stream)); }
    return b;
  }
  MsmImpl(const RawAffine *host_points, size_t n_, int c_, bool fo, bool tables = true, bool uniform_hint = false, bool glv_hint = false) : MsmImpl(make_bases(host_points, n_, c_, fo, tables, uniform_hint, glv_hint), fo, uniform_hint) {}
  MsmImpl(std::shared_ptr<const Bases> shared, bool fo, bool uniform_hint)
      : n(shared->n), c(shared->c), W(shared->W), WB(shared->WB), NB(1u << (shared->c - 1)), filter_ones(fo), bases(shared), points(shared->points), inf(shared->inf), any_inf(shared->any_inf),
        zeroed(2 * (size_t)WB * NB + 2 * sizeof(MsmCounters) / 4), offsets((size_t)WB * NB), entries((n ? n : 1) * (size_t)W * 2), ones(n ? n : 1), ntasks((size_t)WB * NB + 1), task_off((size_t)WB * NB + 1), cls_start(BSORT_CLASSES),
        scanner((size_t)WB * NB), task_scanner((size_t)WB * NB + 1) {
    glv = shared->glv;
    if (glv) { DevBuf<Fe32> b(1); Fe32 bm; memcpy(&bm, GLV_BETA_MONT, 32); b.upload(&bm, 1); beta = std::move(b); }
    { const char *e = getenv("ZK_MSM_SEG"); uint32_t big = e ? (uint32_t)atoi(e) : 16; if (big < 2 || big > 256 || (big & (big - 1))) big = 16; seg = NB >= 4096 ? (WB == 1 ? 4 : big) : 4; }   // one bucket array: few segments, keep the dependent chain short
    n_ones_quads = 16384;
    if (!wfused && WB == 1 && n && getenv("ZK_MSM_HOST_TAIL") != nullptr && atoi(getenv("ZK_MSM_HOST_TAIL")) != 0) {   // opt-in: measured on the GPU box's host a Jacobian addition costs 1.5 us, so the 320 additions of a witness MSM's tail take 0.5 ms of a thread against 0.15-0.25 ms for the quad kernels
      tail_mode = NB <= TAIL_T0 ? 1 : (NB / seg) / GROUP <= TAIL_T0 && NB / seg > GROUP ? 2 : 0;
      if (tail_mode) { tail = DevBuf<uint8_t>(tail_bytes()); tail.zero(); HIP_CHECK(hipHostMalloc((void **)&h_tail, tail_bytes())); memset(h_tail, 0, tail_bytes()); } }
    if (filter_ones && WB == 1 && n && NB >= 16 && NB <= WFUSED_MAX_BUCKETS && n * (size_t)W < (1ull << 31) && !(getenv("ZK_MSM_WFUSED") && atoi(getenv("ZK_MSM_WFUSED")) == 0)) {
      wfused = true; tail_mode = 0; { const char *e = getenv("ZK_MSM_WACC"); wacc_quads = e ? !strcmp(e, "quads") : sizeof(F) > 32; }   // G2 accumulates by quads, G1 by lanes (msm.cuh)
      ws = std::make_shared<WsortBuffers>(); ws->NB = NB; ws->n = n; ws->cap = getenv("ZK_MSM_DIRECT_CAP") ? 2 : (uint32_t)std::min<size_t>(std::max<size_t>(8192, n / 8), 1u << 17);   // (test hook: a tiny region forces the overflow fallback)
      ws->fill = DevBuf<uint32_t>(2 * NB); ws->fill.zero(); ws->entries = DevBuf<uint32_t>((size_t)NB * ws->cap); ws->ones = DevBuf<uint32_t>(n); ws->counters = DevBuf<uint32_t>(2 * sizeof(MsmCounters) / 4); ws->counters.zero();
      HIP_CHECK(hipEventCreateWithFlags(&ws->sorted, hipEventDisableTiming)); }
    if (filter_ones && WB == 1 && n && getenv("ZK_MSM_SPARSE") != nullptr && atoi(getenv("ZK_MSM_SPARSE")) != 0) {   // opt-in (measured: chains 2-3x shorter, but 3x the field products of the bucket path, which the other streams then miss: 1.82 vs 1.74 ms per send proof)
      sparse = true; wfused = false; tail_mode = 0; others_cap = (uint32_t)std::min<size_t>((size_t)n * W, std::max<size_t>((size_t)n * 2, 1u << 16)); others = DevBuf<uint8_t>((size_t)others_cap * sizeof(uint2)); }   // room for two non-zero digits per scalar on average (a witness has ~0.16); more raises the overflow flag and the bucket path below runs instead
    max_tasks = (uint32_t)std::max((n * (size_t)W * 2) / MSM_TASK + (size_t)WB * NB + 1, (size_t)WB * NB * h_slices);
    if (uniform_hint && WB == 1 && !filter_ones && n && getenv("ZK_MSM_NO_DIRECT_SORT") == nullptr) {   // slots per bucket: twice the expected load (+64), a power of two
      size_t lam = (n * (size_t)msm_num_windows(c)) / NB, want = 2 * lam + 64; cap = 64; while (cap < want) cap <<= 1;
      if (const char *e = getenv("ZK_MSM_DIRECT_CAP")) { int v = atoi(e); if (v >= 1 && v <= 4080) cap = (uint32_t)v; }   // test hook: a tiny capacity forces the overflow fallback
      // group-binned variant: G groups of 2^low buckets, about 16 K entries per group (one workgroup sorts a group in registers + LDS)
      { size_t total = n * (size_t)W; uint32_t G = 256; while (G < HSORT_GROUPS && total / G > 16384) G <<= 1; uint32_t low = 0; while ((G << low) < NB) low++; uint32_t ib = 1; while (((size_t)1 << ib) < total) ib++;
        size_t region = ((total / G) * 5 / 4 + 1024 + 255) & ~(size_t)255; if (getenv("ZK_MSM_DIRECT_CAP")) region = 256;   // (test hook: regions far too small force the overflow fallback)
        if (!glv && W <= (int)HSORT_STAGE_W && getenv("ZK_MSM_NO_HSORT") == nullptr && (G << low) == NB && low >= 1 && low <= 10 && low + 1 + ib <= 32 && region <= (size_t)HSORT_GROUP_THREADS * HSORT_MAX_PER_THREAD) {
          hsort = true; hs = HsortShape{G, low, ib, (uint32_t)region}; group_fill = DevBuf<uint32_t>(G); group_fill.zero(); group_n = DevBuf<uint32_t>(G); group_n.zero(); mid = DevBuf<uint32_t>((size_t)G * region); direct = true; entries = DevBuf<uint32_t>((size_t)G * region);
          if (sizeof(F) == 32 && !(getenv("ZK_MSM_HACC") && !strcmp(getenv("ZK_MSM_HACC"), "slices"))) { hruns = true; const char *e = getenv("ZK_MSM_H_RUN"); int v = e ? atoi(e) : 12; h_run = (uint32_t)(v < 4 ? 4 : v > 64 ? 64 : v);
            const size_t lam = total / NB; h_maxp = (uint32_t)((lam + lam / 2 + 32 + h_run - 1) / h_run + 2); } } }   // pieces per bucket: room for 1.5x the expected load + 32 entries
      if (!hsort && (size_t)NB * cap <= (1ull << 28)) { direct = true; entries = DevBuf<uint32_t>((size_t)NB * cap);
        const char *e = getenv("ZK_MSM_DIRECT_TASK"); int tv = e ? atoi(e) : 16; if (lam >= 64 && (tv == 16 || tv == 32 || tv == 64)) task = (uint32_t)tv; } }   // (measured: 32 halves the combine but costs as much in the accumulation, which then has too few lanes)
    { size_t nbk = (size_t)WB * NB; bsort_blocks = cdiv(nbk, BSORT_BLOCK); order = DevBuf<uint32_t>(nbk); rank_of = DevBuf<uint32_t>(nbk); block_hist = DevBuf<uint32_t>((size_t)bsort_blocks * BSORT_CLASSES); block_off = DevBuf<uint32_t>((size_t)bsort_blocks * BSORT_CLASSES); bsort_scanner.reset(new Scanner((size_t)bsort_blocks * BSORT_CLASSES)); }
    buckets = DevBuf<uint8_t>((size_t)WB * NB * sizeof(XYZZ<F>)); partials = DevBuf<uint8_t>(std::max<size_t>(std::max<size_t>(max_tasks, hruns ? ((size_t)NB * h_maxp * 144 + sizeof(XYZZ<F>) - 1) / sizeof(XYZZ<F>) : 0), wfused ? (size_t)WFUSED_BUCKET_LANES + WFUSED_ONES_LANES : 0) * sizeof(XYZZ<F>));
    seg_out = DevBuf<uint8_t>(std::max<size_t>((size_t)WB * (NB / seg), (size_t)32 * cdiv(NB, 512)) * sizeof(XYZZ<F>)); /* (also the chunk sums of the bit-sum tail: log2(NB) x NB/512) */ seg_l2 = DevBuf<uint8_t>((size_t)WB * cdiv(NB / seg, GROUP) * sizeof(XYZZ<F>));
    ones_partial = DevBuf<uint8_t>(std::max<size_t>(n_ones_quads, wfused ? (size_t)NB + WFUSED_ONES_BLOCKS : 0) * sizeof(XYZZ<F>)); if (wfused) lane_off = DevBuf<uint32_t>(WFUSED_MAX_BUCKETS + 1); ones_l2 = DevBuf<uint8_t>((size_t)cdiv(n_ones_quads, GROUP) * sizeof(XYZZ<F>));
    RS = WB; if (wfused) { bitsum = true; RS = WTAIL_SLOTS; }   // k_wtail leaves eight sums by weight bit
    else if (hsort && WB == 1 && NB >= 512 && tail_mode == 0 && !(getenv("ZK_MSM_H_BITSUM") && atoi(getenv("ZK_MSM_H_BITSUM")) == 0)) { bitsum = true; RS = 1; while ((1u << (RS - 1)) < NB) RS++; }   // RS = log2(NB) + 1
    zeroed.zero(); result = DevBuf<uint8_t>(result_bytes()); result.zero();          // the ones slot stays the point at infinity when the ones path is off
    HIP_CHECK(hipHostMalloc((void **)&h_result, result_bytes())); memset(h_result, 0, result_bytes());
    if (getenv("ZK_MSM_MAPPED_RESULT") && atoi(getenv("ZK_MSM_MAPPED_RESULT")) != 0) HIP_CHECK(hipHostGetDevicePointer((void **)&h_result_dev, h_result, 0));   // opt-in: saves the copy's blit kernel (25-40 us on the stream); measured 1.433 vs 1.437 ms per proof, i.e. nothing
    HIP_CHECK(hipStreamSynchronize(gpu().stream));
  }
  ~MsmImpl() { if (h_result) hipHostFree(h_result); if (h_tail) hipHostFree(h_tail); if (ones_stream) hipStreamDestroy(ones_stream); if (ev_classified) hipEventDestroy(ev_classified); if (ev_ones) hipEventDestroy(ev_ones); }
  void enable_split_ones() { if (split_ones || !filter_ones) return; HIP_CHECK(hipStreamCreateWithFlags(&ones_stream, hipStreamNonBlocking)); HIP_CHECK(hipEventCreateWithFlags(&ev_classified, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&ev_ones, hipEventDisableTiming)); split_ones = true; }
  hipStream_t stream() { return stream_id < 0 ? gpu().stream : gpu().aux[stream_id & 3]; }
  // after the stream has been synchronised: did a bucket of the one-pass sort overflow?  Then repeat the last run on the two-pass path (synchronously).
  void finish_sync() { HIP_CHECK(hipStreamSynchronize(stream()));
    if (wfused && host_counters()->pad[0]) { if (!overflow_noted) { overflow_noted = true; fprintf(stderr, "libzkgpu: %s: a bucket of the witness sort overflowed (%u slots), general MSM path used\n", label.c_str(), ws->cap); }
      wfused = false; run_impl(last_scalars, last_index); HIP_CHECK(hipStreamSynchronize(stream())); wfused = true; }   // a bucket's region overflowed: the general path handles any input
    if (sparse && host_counters()->pad[0]) { sparse = false; run_impl(last_scalars, last_index); HIP_CHECK(hipStreamSynchronize(stream())); sparse = true; }   // more digits than the list holds: the bucket path handles any input
    if (direct && host_counters()->pad[0]) { direct = false; offsets_direct = false; const Fe32 *sc = last_scalars; const bool was_hsort = hsort; hsort = false;
      if (was_hsort) { HIP_CHECK(hipMemsetAsync(zeroed.get(), 0, 2 * (size_t)WB * NB * sizeof(uint32_t), stream())); HIP_CHECK(hipMemsetAsync(group_fill.get(), 0, hs.groups * sizeof(uint32_t), stream())); }   // hist() held the bucket counts of the group sort; the two-pass path wants it cleared
      if (prod_b) { if (prod_tmp.size() < n) prod_tmp = DevBuf<Fe32>(n);   // materialise the product for the two-pass path
        hipLaunchKernelGGL(k_fr_mul3, dim3(cdiv(n, 256)), dim3(256), 0, stream(), (const Fr *)last_scalars, (const Fr *)prod_b, (const Fr *)prod_z, (int)prod_z_table, (uint32_t)n, (Fr *)prod_tmp.get()); sc = prod_tmp.get(); }
      const Fe32 *pb = prod_b; prod_b = nullptr; run_impl(sc, last_index); HIP_CHECK(hipStreamSynchronize(stream())); prod_b = pb; direct = true; hsort = was_hsort; } }
  // sum_i (a_i b_i z) P_i; only with the one-pass sort (is_direct()), where the product is formed inside the sort kernel
  void run_product(const Fe32 *a, const Fe32 *b, const Fe32 *z, bool z_is_table) { if (!direct) throw GpuError("msm: run_product needs the one-pass sort"); prod_b = b; prod_z = z; prod_z_table = z_is_table; run_impl(a, nullptr); }
  void run(const Fe32 *scalars, const uint32_t *scalar_index) { prod_b = nullptr; run_impl(scalars, scalar_index); }

  void run_impl(const Fe32 *scalars, const uint32_t *scalar_index) {
    hipStream_t s = stream(); size_t nbk = (size_t)WB * NB; const uint32_t hist_stride = WB == 1 ? 0 : NB, point_stride = WB == 1 && W > 1 ? (uint32_t)n : 0; const uint8_t *infp = any_inf ? inf.get() : nullptr; MsmCounters *cnt = counters();
    const uint32_t bucket_u4 = sizeof(XYZZ<F>) / 16; XYZZ<F> *res = h_result_dev ? (XYZZ<F> *)h_result_dev : (XYZZ<F> *)result.get();
    // (histogram and slot counters were cleared by the constructor and are left cleared by every run (k_msm_combine_tasks); the MsmCounters alternate between two slots)
    parity ^= 1; cnt = counters(); tail_n1 = 0;
    bool ones_forked = false;
    XYZZ<F> *const t0 = (XYZZ<F> *)tail.get(), *const t1 = t0 ? t0 + TAIL_T0 : nullptr; uint4 *const tail_cnt = t0 ? (uint4 *)(t0 + TAIL_T0 + TAIL_T1) : nullptr;
    auto ones_path = [&](hipStream_t os) { Stage st((label + ".ones").c_str(), os); uint32_t g = cdiv(n_ones_quads, GROUP);
      hipLaunchKernelGGL((k_msm_sum_ones<F>), dim3(cdiv((size_t)n_ones_quads * 4, 256)), dim3(256), 0, os, (const Affine<F> *)points.get(), ones.get(), cnt, n_ones_quads, (XYZZ<F> *)ones_partial.get());
      if (tail_mode && g <= TAIL_T1) { tail_n1 = g; hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(g), dim3(256), 0, os, (const XYZZ<F> *)ones_partial.get(), GROUP, n_ones_quads, t1, (uint4 *)cnt, tail_cnt); return; }   // the host adds the g partial sums; block 0 carries the counters along
      hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(g), dim3(256), 0, os, (const XYZZ<F> *)ones_partial.get(), GROUP, n_ones_quads, (XYZZ<F> *)ones_l2.get(), (uint4 *)nullptr, (uint4 *)nullptr);
      hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(1), dim3(256), 0, os, (const XYZZ<F> *)ones_l2.get(), g, g, res + RS, (uint4 *)nullptr, (uint4 *)nullptr); };
    last_scalars = scalars; last_index = scalar_index;
    if (wfused) {
      parity ^= 1;   // (undo the flip above: this path has its own counters, and the general path — which may follow as the overflow fallback — relies on strict alternation of its two slots)
      WsortBuffers &w = *ws; MsmCounters *wc = (MsmCounters *)w.counters.get();
      if (ws_leader) { w.parity ^= 1; w.leader_stream = stream_id; Stage st((label + ".sort").c_str(), s); const uint8_t *winf = w.shared ? nullptr : infp;   // a shared sort keeps every point: the tables differ in which points are at infinity, and the additions skip those
#define ZK_CALL(CC) hipLaunchKernelGGL(k_wsort<CC>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const Fr *)scalars, scalar_index, winf, (uint32_t)n, c, W, point_stride, NB, w.cap, w.fill.get() + (size_t)w.parity * NB, w.fill.get() + (size_t)(w.parity ^ 1) * NB, w.entries.get(), w.ones.get(), wc + w.parity, wc + (w.parity ^ 1))
        ZK_MSM_DISPATCH_C(c, false, ZK_CALL);
#undef ZK_CALL
        if (w.shared) HIP_CHECK(hipEventRecord(w.sorted, s)); }
      else if (w.leader_stream != stream_id) HIP_CHECK(hipStreamWaitEvent(s, w.sorted, 0));   // (a follower on the leader's stream is simply queued behind it)
      uint4 *csrc = (uint4 *)(wc + w.parity); uint4 *cdst = (uint4 *)(res + RS + 1);
      XYZZ<F> *l2 = (XYZZ<F> *)ones_partial.get(); uint32_t n_op;                          // l2: [NB bucket sums | n_op partial sums of the ones]
      { Stage st((label + ".accumulate").c_str(), s); const uint32_t *fl = w.fill.get() + (size_t)w.parity * NB;
        if (wacc_quads) { n_op = WFUSED_ONES_BLOCKS; hipLaunchKernelGGL((k_wacc_quads<F>), dim3(NB + WFUSED_ONES_BLOCKS), dim3(256), 0, s, (const Affine<F> *)points.get(), w.entries.get(), fl, w.cap, NB, w.ones.get(), wc + w.parity, l2); }
        else { n_op = WFUSED_ONES_GROUPS; XYZZ<F> *l1 = (XYZZ<F> *)partials.get();
          hipLaunchKernelGGL((k_wacc_lanes<F>), dim3((WFUSED_BUCKET_LANES + WFUSED_ONES_LANES) / 256), dim3(256), 0, s, (const Affine<F> *)points.get(), w.entries.get(), fl, w.cap, NB, w.ones.get(), wc + w.parity, l1, lane_off.get());
          hipLaunchKernelGGL((k_wacc_fold<F>), dim3(NB + WFUSED_ONES_GROUPS), dim3(256), 0, s, (const XYZZ<F> *)l1, (const uint32_t *)lane_off.get(), NB, l2); } }
      { Stage st((label + ".reduce").c_str(), s); hipLaunchKernelGGL((k_wtail<F>), dim3(2), dim3(256), 0, s, (const XYZZ<F> *)l2, NB, (const XYZZ<F> *)l2 + NB, n_op, res, csrc, cdst); }
      if (!h_result_dev) HIP_CHECK(hipMemcpyAsync(h_result, result.get(), result_bytes(), hipMemcpyDeviceToHost, s)); return;
    }
    if (sparse) {
      const uint32_t nq = 16384, nblk = nq / 64; uint4 *csrc = (uint4 *)cnt; uint4 *cdst = (uint4 *)(res + RS + 1);
      { Stage st((label + ".sort").c_str(), s);
#define ZK_CALL(CC) hipLaunchKernelGGL(k_wmsm_classify<CC>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const Fr *)scalars, scalar_index, infp, (uint32_t)n, c, W, point_stride, ones.get(), (uint2 *)others.get(), others_cap, cnt, counters_next())
        ZK_MSM_DISPATCH_C(c, false, ZK_CALL);
#undef ZK_CALL
      }
      { Stage st((label + ".accumulate").c_str(), s); hipLaunchKernelGGL((k_wmsm_sum<F>), dim3(nblk), dim3(256), 0, s, (const Affine<F> *)points.get(), ones.get(), (const uint2 *)others.get(), others_cap, cnt, nq, (XYZZ<F> *)ones_partial.get()); }
      { Stage st((label + ".reduce").c_str(), s); hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(1), dim3(256), 0, s, (const XYZZ<F> *)ones_partial.get(), nblk, nblk, res, csrc, cdst); }
      if (!h_result_dev) HIP_CHECK(hipMemcpyAsync(h_result, result.get(), result_bytes(), hipMemcpyDeviceToHost, s)); return;
    }
    const bool hs_run = direct && hsort && scalar_index == nullptr;
    if (hs_run) { Stage st((label + ".sort").c_str(), s);
#define ZK_CALL(CC) hipLaunchKernelGGL(k_hsort_bin<CC>, dim3(cdiv(n, HSORT_TILE)), dim3(HSORT_BIN_THREADS), 0, s, (const Fr *)scalars, (const Fr *)prod_b, (const Fr *)prod_z, (int)prod_z_table, infp, (uint32_t)n, c, W, point_stride, hs, group_fill.get(), mid.get(), cnt, counters_next())
      ZK_MSM_DISPATCH_C(c, false, ZK_CALL);
#undef ZK_CALL
      hipLaunchKernelGGL(k_hsort_group, dim3(hs.groups), dim3(HSORT_GROUP_THREADS), 0, s, (const uint32_t *)mid.get(), group_fill.get(), hs, entries.get(), hist(), offsets.get(), group_n.get(), (int)hruns);
    } else
    if (direct) { Stage st((label + ".sort").c_str(), s);
      if (hsort) throw GpuError("msm: the group-binned sort takes no scalar index");
      if (!offsets_direct) { std::vector<uint32_t> o(nbk); for (size_t b = 0; b < nbk; b++) o[b] = (uint32_t)(b * cap); offsets.upload(o.data(), nbk);
        for (size_t b = 0; b < nbk; b++) o[b] = (uint32_t)b; order.upload(o.data(), nbk); rank_of.upload(o.data(), nbk); offsets_direct = true; }   // identity ranking: uniform buckets need no size ordering
      hipLaunchKernelGGL(k_msm_scatter_direct<0>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const Fr *)scalars, scalar_index, infp, (uint32_t)n, c, W, point_stride, cap, hist(), entries.get(), cnt, counters_next(), (const Fr *)prod_b, (const Fr *)prod_z, (int)prod_z_table, (int)glv);
      if (nbk <= PLAN_DIRECT_MAX && cap <= 4080) hipLaunchKernelGGL(k_msm_plan_direct, dim3(1), dim3(PLAN_THREADS), 0, s, hist(), (uint32_t)nbk, cap, task, task_off.get(), cls_start.get());
      else {
        hipLaunchKernelGGL(k_bsort_hist, dim3(bsort_blocks), dim3(BSORT_BLOCK), 0, s, hist(), (uint32_t)nbk, bsort_blocks, block_hist.get(), cap);
        bsort_scanner->run(block_hist.get(), block_off.get(), (size_t)bsort_blocks * BSORT_CLASSES, s);
        hipLaunchKernelGGL(k_bsort_scatter, dim3(bsort_blocks), dim3(BSORT_BLOCK), 0, s, hist(), (uint32_t)nbk, bsort_blocks, block_off.get(), order.get(), rank_of.get(), ntasks.get(), cls_start.get(), (uint4 *)bucket_array(), bucket_u4);
        task_scanner.run(ntasks.get(), task_off.get(), nbk + 1, s);
      }
    } else
    { Stage st((label + ".sort").c_str(), s);
#define ZK_CALL(CC) hipLaunchKernelGGL(k_msm_classify<CC>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const Fr *)scalars, scalar_index, infp, (uint32_t)n, c, W, (int)filter_ones, hist_stride, (uint32_t)nbk, (int)glv, hist(), ones.get(), cnt, counters_next())
      if (n) ZK_MSM_DISPATCH_C(c, glv, ZK_CALL);
#undef ZK_CALL
      if (filter_ones && n && split_ones) { HIP_CHECK(hipEventRecord(ev_classified, s)); HIP_CHECK(hipStreamWaitEvent(ones_stream, ev_classified, 0)); ones_path(ones_stream); HIP_CHECK(hipEventRecord(ev_ones, ones_stream)); ones_forked = true; }
      if (nbk <= PLAN_SMALL_MAX) {
        hipLaunchKernelGGL(k_msm_plan_small, dim3(1), dim3(PLAN_THREADS), 0, s, hist(), (uint32_t)nbk, offsets.get(), order.get(), rank_of.get(), task_off.get(), cls_start.get(), (uint4 *)bucket_array(), bucket_u4);
      } else {
        scanner.run(hist(), offsets.get(), nbk, s);
        hipLaunchKernelGGL(k_bsort_hist, dim3(bsort_blocks), dim3(BSORT_BLOCK), 0, s, hist(), (uint32_t)nbk, bsort_blocks, block_hist.get(), 0u);
        bsort_scanner->run(block_hist.get(), block_off.get(), (size_t)bsort_blocks * BSORT_CLASSES, s);
        hipLaunchKernelGGL(k_bsort_scatter, dim3(bsort_blocks), dim3(BSORT_BLOCK), 0, s, hist(), (uint32_t)nbk, bsort_blocks, block_off.get(), order.get(), rank_of.get(), ntasks.get(), cls_start.get(), (uint4 *)bucket_array(), bucket_u4);
        task_scanner.run(ntasks.get(), task_off.get(), nbk + 1, s);
      }
#define ZK_CALL(CC) hipLaunchKernelGGL(k_msm_scatter<CC>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const Fr *)scalars, scalar_index, infp, (uint32_t)n, c, W, (int)filter_ones, hist_stride, point_stride, (uint32_t)nbk, (int)glv, offsets.get(), fill(), entries.get())
      if (n) ZK_MSM_DISPATCH_C(c, glv, ZK_CALL);
#undef ZK_CALL
    }
    if (hs_run && hruns) {
      if constexpr (sizeof(F) == 32) {
      { Stage st((label + ".accumulate").c_str(), s); const dim3 grid(cdiv(cdiv(n * (size_t)W, h_run) + hs.groups, 256));   // one lane per run: at most ceil(entries / run) + one short run per group
        if (bases->points261.size()) { if (any_inf) hipLaunchKernelGGL(k_hacc_runs29<1>, grid, dim3(256), 0, s, (const Affine<Fq> *)bases->points261.get(), entries.get(), group_n.get(), offsets.get(), hs, h_run, h_maxp, (Piece29 *)partials.get(), cnt);
          else hipLaunchKernelGGL(k_hacc_runs29<0>, grid, dim3(256), 0, s, (const Affine<Fq> *)bases->points261.get(), entries.get(), group_n.get(), offsets.get(), hs, h_run, h_maxp, (Piece29 *)partials.get(), cnt); }
        else if (any_inf) hipLaunchKernelGGL((k_hacc_runs<1, F>), grid, dim3(256), 0, s, (const Affine<F> *)points.get(), entries.get(), group_n.get(), offsets.get(), hs, h_run, h_maxp, (XYZZ<F> *)partials.get(), cnt);
        else hipLaunchKernelGGL((k_hacc_runs<0, F>), grid, dim3(256), 0, s, (const Affine<F> *)points.get(), entries.get(), group_n.get(), offsets.get(), hs, h_run, h_maxp, (XYZZ<F> *)partials.get(), cnt); }
      { Stage st((label + ".combine").c_str(), s); static const uint32_t ll = [] { const char *e = getenv("ZK_MSM_H_COMBINE_LANES"); int v = e ? atoi(e) : 1; return (uint32_t)(v < 0 ? 0 : v > 3 ? 3 : v); }();
        if (bases->points261.size()) hipLaunchKernelGGL(k_hacc_combine29, dim3(cdiv(nbk << ll, 256)), dim3(256), 0, s, (const Piece29 *)partials.get(), offsets.get(), hist(), hs, h_run, h_maxp, (uint32_t)nbk, ll, (XYZZ<Fq> *)bucket_array(), cnt);
        else hipLaunchKernelGGL((k_hacc_combine<F>), dim3(cdiv(nbk << ll, 256)), dim3(256), 0, s, (const XYZZ<F> *)partials.get(), offsets.get(), hist(), hs, h_run, h_maxp, (uint32_t)nbk, ll, bucket_array()); }
      }
    } else
    if (hs_run) {
      { Stage st((label + ".accumulate").c_str(), s); static const int av = [] { const char *e = getenv("ZK_ACC_VARIANT"); return e ? atoi(e) & 3 : 0; }();
#define ZK_ACC(V) hipLaunchKernelGGL((k_msm_accumulate_slices<V, F>), dim3(cdiv(nbk * h_slices, 256)), dim3(256), 0, s, (const Affine<F> *)points.get(), entries.get(), offsets.get(), hist(), (uint32_t)nbk, h_slices, (XYZZ<F> *)partials.get())
        if (av == 0) ZK_ACC(0); else if (av == 1) ZK_ACC(1); else if (av == 2) ZK_ACC(2); else ZK_ACC(3);
#undef ZK_ACC
      }
      static const int comb_ll = [] { const char *e = getenv("ZK_MSM_H_COMBINE_LANES"); return e ? atoi(e) : 1; }();   // >= 0: lane-serial combine with 2^ll lanes per bucket (default two lanes: 77 vs 103 us inside a proof for the quad form, which -1 selects)
      if (comb_ll >= 0 && comb_ll <= 4 && h_slices % (1u << comb_ll) == 0) { Stage st((label + ".combine").c_str(), s); hipLaunchKernelGGL((k_msm_combine_lanes<F>), dim3(cdiv(nbk << comb_ll, 256)), dim3(256), 0, s, (const XYZZ<F> *)partials.get(), (uint32_t)nbk, h_slices, (uint32_t)comb_ll, bucket_array()); }
      else
      { Stage st((label + ".combine").c_str(), s); hipLaunchKernelGGL((k_msm_combine_slices<F>), dim3(cdiv((nbk * 4) << h_combine_lq, 256)), dim3(256), 0, s, (const XYZZ<F> *)partials.get(), (uint32_t)nbk, h_slices, h_combine_lq, bucket_array()); }
    } else {
    { Stage st((label + ".accumulate").c_str(), s);
      hipLaunchKernelGGL((k_msm_accumulate_tasks<F>), dim3(cdiv(max_tasks, 256)), dim3(256), 0, s, (const Affine<F> *)points.get(), entries.get(), offsets.get(), hist(), order.get(), task_off.get(), (uint32_t)nbk, max_tasks, direct ? task : MSM_TASK, glv ? (const F *)beta.get() : (const F *)nullptr,
                         bucket_array(), (XYZZ<F> *)partials.get());
    }
    { Stage st((label + ".combine").c_str(), s);
      const uint32_t heavy = direct && cap <= COMBINE_QUAD_MAX * task ? 0u : HEAVY_BLOCKS;   // one-pass sort: no bucket can hold more than `cap` entries, so none needs a whole workgroup
      hipLaunchKernelGGL((k_msm_combine_tasks<F>), dim3(heavy + cdiv(nbk, 64)), dim3(256), 0, s, order.get(), task_off.get(), cls_start.get(), heavy, (const XYZZ<F> *)partials.get(), bucket_array(), zeroed.get(), (uint32_t)(2 * nbk), (int)(direct && nbk <= PLAN_DIRECT_MAX && cap <= 4080));
    }
    }
    const bool ones_runs = filter_ones && n;
    if (tail_mode == 1) { tail_n0 = NB; if (!ones_runs) HIP_CHECK(hipMemcpyAsync(tail_cnt, cnt, sizeof(MsmCounters), hipMemcpyDeviceToDevice, s)); }   // the buckets already sit in the tail buffer: nothing left to launch
    else if (bitsum && hs_run) { Stage st_red((label + ".reduce").c_str(), s); static const uint32_t per_env = [] { const char *e = getenv("ZK_MSM_BITSUM_PER"); int v = e ? atoi(e) : 8; return (uint32_t)(v == 4 || v == 8 || v == 16 || v == 32 ? v : 8); }();
      uint32_t per = per_env; while (per > 4 && (NB / 2) % (64 * per)) per >>= 1; const uint32_t top = (uint32_t)RS - 1, chunks = (NB / 2) / (64 * per);   // sums by weight bit; the host's Horner rule does the rest (combine() with c = 1)
      hipLaunchKernelGGL((k_bitsum_chunks<F>), dim3(chunks, top), dim3(256), 0, s, (const XYZZ<F> *)bucket_array(), per, (XYZZ<F> *)seg_out.get());
      hipLaunchKernelGGL((k_bitsum_final<F>), dim3(top + 1), dim3(256), 0, s, (const XYZZ<F> *)seg_out.get(), chunks, top, (const XYZZ<F> *)bucket_array(), NB, res, (uint4 *)cnt, (uint4 *)(res + RS + 1)); }
    else { Stage st_red((label + ".reduce").c_str(), s);
      if (RS > WB) HIP_CHECK(hipMemsetAsync(res + WB, 0, (size_t)(RS - WB) * sizeof(XYZZ<F>), s));   // (a bit-sum MSM on its fallback path: one window sum in slot 0, the other slots at infinity)
      uint32_t spw = NB / seg, nseg = (uint32_t)WB * spw; uint4 *csrc = (uint4 *)cnt; uint4 *cdst = (uint4 *)(res + RS + 1);
      hipLaunchKernelGGL((k_msm_reduce_segments<F>), dim3(cdiv(nseg, 16)), dim3(64), 0, s, (const XYZZ<F> *)bucket_array(), NB, seg, nseg, (XYZZ<F> *)seg_out.get());
      if (tail_mode == 2) { uint32_t g = spw / GROUP; tail_n0 = g; hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(g), dim3(256), 0, s, (const XYZZ<F> *)seg_out.get(), GROUP, nseg, t0, ones_runs ? (uint4 *)nullptr : csrc, ones_runs ? (uint4 *)nullptr : tail_cnt); }
      else if (spw > GROUP) { uint32_t g = spw / GROUP;   // two-level tree per window keeps the dependent chain short (spw is a power of two)
        hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(WB * g), dim3(256), 0, s, (const XYZZ<F> *)seg_out.get(), GROUP, nseg, (XYZZ<F> *)seg_l2.get(), (uint4 *)nullptr, (uint4 *)nullptr);
        hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(WB), dim3(256), 0, s, (const XYZZ<F> *)seg_l2.get(), g, (uint32_t)WB * g, res, csrc, cdst);
      } else hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(WB), dim3(256), 0, s, (const XYZZ<F> *)seg_out.get(), spw, nseg, res, csrc, cdst);
    }
    if (filter_ones && n && !split_ones) ones_path(s);
    if (ones_forked) HIP_CHECK(hipStreamWaitEvent(s, ev_ones, 0));
    if (tail_mode) HIP_CHECK(hipMemcpyAsync(h_tail, tail.get(), tail_bytes(), hipMemcpyDeviceToHost, s));
    else if (!h_result_dev) HIP_CHECK(hipMemcpyAsync(h_result, result.get(), result_bytes(), hipMemcpyDeviceToHost, s));
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

// the host tail (MsmImpl::tail_mode): mode 1 — sum_b (b + 1) B_b over the bucket array by running sums; mode 2 — plain sum; plus the partial sums of the scalar-one path
template <class HF, class F, class Impl> static host::HPoint<HF> host_tail_sum(const Impl &m) {
  auto get = [&](const XYZZ<F> *arr, uint32_t i) { const char *b = (const char *)&arr[i]; size_t fs = sizeof(F); return host::HPoint<HF>::from_xyzz(load_hf<HF>(b), load_hf<HF>(b + fs), load_hf<HF>(b + 2 * fs), load_hf<HF>(b + 3 * fs)); };
  host::HPoint<HF> acc = host::HPoint<HF>::inf();
  if (m.tail_mode == 1) { host::HPoint<HF> run = host::HPoint<HF>::inf(); for (uint32_t b = m.tail_n0; b-- > 0;) { run = run.add(get(m.host_tail0(), b)); acc = acc.add(run); } }
  else for (uint32_t i = 0; i < m.tail_n0; i++) acc = acc.add(get(m.host_tail0(), i));
  for (uint32_t i = 0; i < m.tail_n1; i++) acc = acc.add(get(m.host_tail1(), i));
  return acc;
}

}  // namespace zk
