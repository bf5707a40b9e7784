// Host-side driver of the MSM kernels, shared by the G1 and G2 translation units.
#pragma once
#include <cstring>
#include <memory>
#include <atomic>
#include <cstdio>
#include "gpu_internal.hpp"

// window sizes with a compiled-in digit walk (msm.cuh: for_each_digit); every other size takes the runtime path (C = 0). An instantiation with C > 0 ignores
// the runtime c / W it is handed: ZK_MSM_CHECK_C (below, every call site) makes sure they agree with it.
#define ZK_MSM_DISPATCH_C(cval, CALL) do { if ((cval) == 8) { ZK_MSM_CHECK_C(8); CALL(8); } else if ((cval) == 16) { ZK_MSM_CHECK_C(16); CALL(16); } else { CALL(0); } } while (0)
#define ZK_MSM_CHECK_C(CC) do { if (c != (CC) || W != msm_num_windows(CC)) throw GpuError("msm: compiled-in digit walk launched with a different window size"); } while (0)

namespace zk {
// what k_wsort leaves behind (msm.cuh, "witness MSMs in three launches"): one set per scalar vector, shared by the MSMs over that vector (A and L*; B1 and B2 —
// different curve groups, hence no template parameter here). The leader runs the sort; a follower queued behind it on the same stream, or waiting for `sorted`,
// only accumulates.
struct WsortBuffers { uint32_t NB = 0, cap = 0; size_t n = 0; DevBuf<uint32_t> fill /* 2 x NB: the runs alternate */, entries /* NB x cap */, ones /* n */,
    counters /* 2 MsmCounters */; int parity = 0; bool shared = false; hipEvent_t sorted = nullptr; int leader_stream = -1;
  ~WsortBuffers() { if (sorted) hipEventDestroy(sorted); } };
template <class F, class RawAffine>
struct MsmImpl {
  size_t n; int c, W, WB; uint32_t NB;   // W digit windows; WB bucket arrays (1 when the multiples 2^(cw) P are precomputed, else W)
  bool filter_ones; uint32_t seg, n_ones_quads; std::string label = "msm"; int stream_id = -1;   // -1: main stream, 0..3: auxiliary stream
  // the key's points (with the fixed-base table when there is one) are immutable and shared by every prover object of the key on this device; everything else
  // below is per-object workspace
  struct Bases { size_t n = 0; int c = 0, W = 0, WB = 0; bool any_inf = false, table261_only = false /* `points` holds the n base points only: the table lives in points261 */; DevBuf<RawAffine> points,
      points261 /* the same table with coordinates x 2^261: what k_hacc_runs29 gathers from (G1, uniform scalars, fixed-base table) and k_wacc_lanes_g2_29 (G2 witness MSM) */, groups261 /* ones_groups in that form (G2) */, ones_groups /* 15 subset sums per four consecutive points: what the fused witness path adds for the scalars equal to one */;
      DevBuf<uint8_t> inf;
    };
  std::shared_ptr<const Bases> bases; const DevBuf<RawAffine> &points; const DevBuf<uint8_t> &inf; bool any_inf = false;
  // the ones path on a stream of its own, beside the bucket path (general path of the G2 MSM: both are long chains)
  bool split_ones = false;
  hipStream_t ones_stream = nullptr;
  hipEvent_t ev_classified = nullptr, ev_ones = nullptr;
  const Fe32 *prod_b = nullptr, *prod_z = nullptr; bool prod_z_table = false; DevBuf<Fe32> prod_tmp;   // scalars given as a product a*b*z (run_product)
  // H query (uniform scalars, one bucket array): group-binned one-pass sort, accumulation over runs of h_run entries on 29-bit limbs, at most h_maxp pieces per
  // bucket
  // (run length swept on MI355X, accumulate + combine inside a send proof: 8: 0.397, 12: 0.382, 16: 0.402 ms, profiles/r03f_ab.txt; again in round 4 with the
  // 29-bit tail, whole proof: 12: 1.038, 16: 1.052, 20: 1.057, 24: 1.090 ms, profiles/r04a_hrun_sweep.txt; window 15 / 16 / 17: 0.992 / 0.952 / 0.976 ms,
  // profiles/r04d_ab.txt)
  bool hsort = false;
  HsortShape hs{0, 0, 0, 0};
  DevBuf<uint32_t> group_fill, mid, group_n;
  uint32_t h_run = 11, h_maxp = 16;
  // Several proofs in flight (round 5): the chip is full whatever one accumulation's wave count is, so the runs are made longer — fewer pieces per bucket, i.e.
  // fewer general additions in the combine (12.6 -> 4.2 pieces a bucket at 40: -6 % of a proof's instructions; six provers in flight 1,730 -> 1,875 proofs/s,
  // prove_batch(64) 1,680 -> 1,860; 24 / 32 / 48: 1,843 / 1,858 / 1,835; profiles/r05_hacc_sweeps.txt).  One proof at a time keeps the short runs: its latency is the accumulation's last partial round.
  uint32_t h_run_crowded = 40; bool crowded = false;
  // witness MSMs in three launches (k_wsort / k_wacc / k_wtail); needs the fixed-base tables and at most 128 buckets
  bool wfused = false, ws_leader = true, overflow_noted = false; std::shared_ptr<WsortBuffers> ws;
  const Fe32 *last_scalars = nullptr; const uint32_t *last_index = nullptr;
  // this run's assignment came in compact form (run_tagged): the witness sort reads tags instead of scalars
  WitnessTags wtags;
  const Fe32 *z_all = nullptr;
  bool tagged = false;
  DevBuf<uint32_t> zeroed;                                          // [hist | fill | counters]: cleared once; every run leaves them cleared
  DevBuf<uint32_t> offsets, entries, ones, ntasks, task_off, order, rank_of, block_hist, block_off, cls_start;
  uint32_t bsort_blocks;
  std::unique_ptr<Scanner> bsort_scanner;
  Scanner scanner, task_scanner;
  uint32_t max_tasks;
  DevBuf<uint32_t> lane_off;   // fused witness path: where each bucket's lanes start
  DevBuf<uint8_t> hb29, hmarg; bool htail29 = false;   // H path: bucket sums and marginal sums on 29-bit limbs (htail29.cuh: Point29Rec)
  // XYZZ<F> arrays (partials: Piece29 on the H path), kept as bytes to stay out of the header
  DevBuf<uint8_t> buckets, partials, seg_out, seg_l2, ones_partial, ones_l2;
  // the MSM's result — RS sums, the ones sum, the counters — in pinned host memory, and that memory's device address (the kernels write there directly)
  uint8_t *h_result = nullptr, *res_dev = nullptr;
  static constexpr uint32_t HEAVY_BLOCKS = 256, GROUP = 256;

  void share_sort(const std::shared_ptr<WsortBuffers> &leader_ws) {
    if (!wfused || !leader_ws || leader_ws->NB != NB || leader_ws->n != n) throw GpuError("msm: this MSM cannot share the sort (different size or path)");
    ws = leader_ws;
    ws_leader = false;
    ws->shared = true;
  }
  uint32_t *hist() { return zeroed.get(); }
  uint32_t *fill() { return zeroed.get() + (size_t)WB * NB; }
  int parity = 0;                                                   // which of the two counter slots the current run uses
  MsmCounters *counters() { return (MsmCounters *)(zeroed.get() + 2 * (size_t)WB * NB) + parity; }
  MsmCounters *counters_next() { return (MsmCounters *)(zeroed.get() + 2 * (size_t)WB * NB) + (parity ^ 1); }
  int RS = 0;   // result slots before the ones slot: WB window sums, or (bitsum) log2(NB) + 1 sums by weight bit
  bool bitsum = false;
  size_t result_bytes() const { return (size_t)(RS + 1) * sizeof(XYZZ<F>) + sizeof(MsmCounters); }
  const XYZZ<F> *host_sums() const { return (const XYZZ<F> *)h_result; }
  const MsmCounters *host_counters() const { return (const MsmCounters *)(h_result + (size_t)(RS + 1) * sizeof(XYZZ<F>)); }
  XYZZ<F> *bucket_array() { return reinterpret_cast<XYZZ<F> *>(buckets.get()); }

  // Precomputed multiples 2^(cw) P unless switched off (ZK_MSM_PRECOMPUTE=0), the table would pass ZK_MSM_PRECOMPUTE_MAX_MB, or its indices would not fit the
  // 31 bits of a sorted entry. The cap was 768 MB through round 3 — round 1 had measured the depth-32 deposit key slower with its 2-9 GB of tables (13.1
  // against 9.1 ms), on the general-path kernels of that time. With tables an MSM takes the fused paths (one bucket array: k_hsort_* / k_hacc_runs29 for the H
  // query, k_wsort / k_wacc_* / k_wtail for the witness MSMs), which is worth far more than the gathers from a multi-GB table cost: depth-32 deposit 6.6 ms at
  // 768 MB, 4.1 ms at 1.4 GB (H only), 3.27 ms from 3 GB on (all five queries; profiles/r04g_deposit32_caps.txt). Default since round 4: 8 GB per table — a
  // key's tables add up to ~14 GB at depth 32, of 288.
  static bool use_precompute(size_t n_, int W_) { static const bool on = [] { const char *e = getenv("ZK_MSM_PRECOMPUTE"); return !e || atoi(e) != 0; }();
    static const size_t cap = [] { const char *e = getenv("ZK_MSM_PRECOMPUTE_MAX_MB"); return (size_t)(e ? atol(e) : 8192) << 20; }();
    return on && n_ > 0 && W_ > 1 && n_ * (size_t)W_ < (1ull << 31) && n_ * (size_t)W_ * sizeof(RawAffine) <= cap; }
  // What a query's tables take while they are built: both coordinate forms of the table, the subset sums of the ones (15 / 4 records a point, twice), and the
  // precompute kernel's scratch.  Tables are an optimisation: with less free HBM than that (a smaller device, many keys x ZK_DEVICES pools, another tenant) the
  // query is loaded without them — the window-by-window path — instead of failing the key load.
  static bool tables_fit_device(size_t n_, int W_, bool fo) {
    static const double share = [] { const char *e = getenv("ZK_MSM_TABLES_FREE_SHARE"); const double v = e ? atof(e) : 0.6; return v > 0 && v <= 1 ? v : 0.6; }();
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return true;
    const size_t table = n_ * (size_t)W_ * sizeof(RawAffine), scratch = (size_t)(W_ - 1) * n_ * (sizeof(XYZZ<F>) + sizeof(F)), groups = fo ? (n_ + 3) / 4 * 15 * sizeof(RawAffine) : 0;
    const bool ok = (double)(2 * table + 2 * groups + scratch) <= share * (double)fr;
    if (!ok) { static std::atomic<int> noted{0}; if (noted.fetch_add(1) < 4) fprintf(stderr,
        "libzkgpu: %.1f GB of fixed-base tables for a query of %zu points do not fit the device's free memory (%.1f GB): loading it without tables\n",
        (2 * table + 2 * groups + scratch) / 1e9, n_, fr / 1e9); }
    return ok;
  }
  static std::shared_ptr<const Bases> make_bases(const RawAffine *host_points, size_t n_, int c_, bool fo, bool tables, bool uniform_hint) {
    if (tables && use_precompute(n_, msm_num_windows(c_))) {
      if (!tables_fit_device(n_, msm_num_windows(c_), fo)) { tables = false; note_query_without_tables(); }
      else {
        // (an allocation can still fail half-way — another process took the memory meanwhile: once more without tables.  ONLY an allocation that did not fit: a
        // kernel fault or a failed synchronisation while the tables are built is an error of the key load, not a reason to go on several times slower in silence)
        try { return make_bases_impl(host_points, n_, c_, fo, true, uniform_hint); }
        catch (const GpuOutOfMemory &e) {
          (void)hipGetLastError();
          fprintf(stderr, "libzkgpu: building the fixed-base tables of a query of %zu points failed (%s): loading it without tables\n", n_, e.what());
          tables = false; note_query_without_tables();
        }
      }
    }
    return make_bases_impl(host_points, n_, c_, fo, tables, uniform_hint);
  }
  static std::shared_ptr<const Bases> make_bases_impl(const RawAffine *host_points, size_t n_, int c_, bool fo, bool tables, bool uniform_hint) {
    auto b = std::make_shared<Bases>(); b->n = n_; b->c = c_; b->W = msm_num_windows(c_); b->WB = tables && use_precompute(n_, b->W) ? 1 : b->W;
    if (c_ < 6 || c_ > 20 || b->W > MSM_MAX_WINDOWS) throw GpuError("msm: unsupported window size");
    b->points = DevBuf<RawAffine>((n_ ? n_ : 1) * (size_t)(b->WB == 1 ? b->W : 1)); b->inf = DevBuf<uint8_t>(n_ ? n_ : 1);
    std::vector<uint8_t> flags(n_ ? n_ : 1, 0); const uint8_t zero[sizeof(RawAffine)] = {0};
    for (size_t i = 0; i < n_; i++) if (!memcmp(&host_points[i], zero, sizeof(RawAffine))) { flags[i] = 1; b->any_inf = true; }
    if (n_) { b->points.upload(host_points, n_); b->inf.upload(flags.data(), n_); }
    if (b->WB == 1 && b->W > 1 && n_) {   // table[w*n + i] = 2^(c*w) * P_i (k_msm_precompute); the scratch arrays live only for this launch
      DevBuf<uint8_t> tmp((size_t)(b->W - 1) * n_ * sizeof(XYZZ<F>)), pref((size_t)(b->W - 1) * n_ * sizeof(F));
      hipLaunchKernelGGL((k_msm_precompute<F>), dim3(cdiv(n_, 64)), dim3(64), 0, gpu().stream, (Affine<F> *)b->points.get(), (uint32_t)n_, c_, b->W,
          (XYZZ<F> *)tmp.get(), (F *)pref.get());
      HIP_CHECK(hipGetLastError()); HIP_CHECK(hipStreamSynchronize(gpu().stream));
    }
    // the H query's table once more, in the 29-bit kernels' Montgomery radix
    if constexpr (sizeof(F) == 32) if (uniform_hint && !fo && b->WB == 1 && n_) {
      const size_t tn = n_ * (size_t)b->W;
      b->points261 = DevBuf<RawAffine>(tn);
      hipLaunchKernelGGL(k_table_to_r261, dim3(cdiv(tn, 256)), dim3(256), 0, gpu().stream, (const Affine<Fq> *)b->points.get(),
          (Affine<Fq> *)b->points261.get(), tn);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(gpu().stream));
    }
    // subset sums of four consecutive points for the scalars equal to one (k_ones_groups)
    if (fo && b->WB == 1 && n_) {
      const size_t ng = (n_ + 3) / 4;
      b->ones_groups = DevBuf<RawAffine>(ng * 15);
      hipLaunchKernelGGL((k_ones_groups<F>), dim3(cdiv(ng, 64)), dim3(64), 0, gpu().stream, (const Affine<F> *)b->points.get(), (uint32_t)n_,
          (Affine<F> *)b->ones_groups.get());
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(gpu().stream));
    }
    // a G1 witness MSM accumulates, folds and sums on 29-bit limbs: both tables once more with coordinates x 2^261
    if constexpr (sizeof(F) == 32) if (fo && b->WB == 1 && n_) {
      const size_t tn = n_ * (size_t)b->W, gn = b->ones_groups.size(); b->points261 = DevBuf<RawAffine>(tn); b->groups261 = DevBuf<RawAffine>(gn ? gn : 1);
      hipLaunchKernelGGL(k_table_to_r261, dim3(cdiv(tn, 256)), dim3(256), 0, gpu().stream, (const Affine<Fq> *)b->points.get(),
          (Affine<Fq> *)b->points261.get(), tn);
      if (gn) hipLaunchKernelGGL(k_table_to_r261, dim3(cdiv(gn, 256)), dim3(256), 0, gpu().stream, (const Affine<Fq> *)b->ones_groups.get(),
          (Affine<Fq> *)b->groups261.get(), gn);
      HIP_CHECK(hipGetLastError()); HIP_CHECK(hipStreamSynchronize(gpu().stream)); }
    // the G2 witness MSM accumulates on 29-bit limbs: both tables once more with coordinates x 2^261
    if constexpr (sizeof(F) == 64) if (fo && b->WB == 1 && n_) {
      const size_t tn = n_ * (size_t)b->W, gn = b->ones_groups.size(); b->points261 = DevBuf<RawAffine>(tn); b->groups261 = DevBuf<RawAffine>(gn ? gn : 1);
      hipLaunchKernelGGL(k_table_to_r261_g2, dim3(cdiv(tn, 256)), dim3(256), 0, gpu().stream, (const Affine<Fq2> *)b->points.get(),
          (Affine<Fq2> *)b->points261.get(), tn);
      if (gn) hipLaunchKernelGGL(k_table_to_r261_g2, dim3(cdiv(gn, 256)), dim3(256), 0, gpu().stream, (const Affine<Fq2> *)b->ones_groups.get(),
          (Affine<Fq2> *)b->groups261.get(), gn);
      HIP_CHECK(hipGetLastError()); HIP_CHECK(hipStreamSynchronize(gpu().stream)); }
    // Round 5: ONE coordinate form per table.  Whatever the fused paths gather from is in points261 / groups261; the x 2^256 table next to it served the general
    // path only (fallbacks: an overflowing one-pass sort, a degenerate sum), which now reads points261 and converts on the fly (k_msm_accumulate_tasks<F, 1>).
    // Kept in the x 2^256 form: the n base points (the general path's sum over the scalars equal to one).  Send: 4.1 -> 2.1 GB per key, deposit-32: 14 -> 7 GB.
    if (b->points261.size() == n_ * (size_t)b->W && b->WB == 1 && b->W > 1 && n_) {
      HIP_CHECK(hipStreamSynchronize(gpu().stream));
      DevBuf<RawAffine> base(n_);
      HIP_CHECK(hipMemcpyAsync(base.get(), b->points.get(), n_ * sizeof(RawAffine), hipMemcpyDeviceToDevice, gpu().stream));
      HIP_CHECK(hipStreamSynchronize(gpu().stream));
      b->points = std::move(base); b->ones_groups = DevBuf<RawAffine>(); b->table261_only = true;
    }
    return b;
  }
  MsmImpl(const RawAffine *host_points, size_t n_, int c_, bool fo, bool tables = true, bool uniform_hint = false) : MsmImpl(make_bases(host_points, n_, c_,
      fo, tables, uniform_hint), fo, uniform_hint) {
  }
  MsmImpl(std::shared_ptr<const Bases> shared, bool fo, bool uniform_hint)
      : n(shared->n), c(shared->c), W(shared->W), WB(shared->WB), NB(1u << (shared->c - 1)), filter_ones(fo), bases(shared), points(shared->points),
          inf(shared->inf), any_inf(shared->any_inf),
        zeroed(2 * (size_t)WB * NB + 2 * sizeof(MsmCounters) / 4), offsets((size_t)WB * NB), entries((n ? n : 1) * (size_t)W * 2), ones(n ? n : 1),
            ntasks((size_t)WB * NB + 1), task_off((size_t)WB * NB + 1), cls_start(BSORT_CLASSES),
        scanner((size_t)WB * NB), task_scanner((size_t)WB * NB + 1) {
    // buckets per segment of the weighted reduction on the general path (one bucket array: few segments, short dependent chain)
    seg = NB >= 4096 ? (WB == 1 ? 4 : 16) : 4;
    n_ones_quads = 16384;
    if (filter_ones && WB == 1 && n && NB >= 16 && NB <= WFUSED_MAX_BUCKETS && n * (size_t)W < (1ull << 31)) {
      wfused = true;
      // (test hook: a tiny region forces the overflow fallback)
      ws = std::make_shared<WsortBuffers>();
      ws->NB = NB;
      ws->n = n;
      ws->cap = getenv("ZK_MSM_DIRECT_CAP") ? 2 : (uint32_t)std::min<size_t>(std::max<size_t>(8192, n / 8), 1u << 17);
      ws->fill = DevBuf<uint32_t>(2 * NB);
      ws->fill.zero();
      ws->entries = DevBuf<uint32_t>((size_t)NB * ws->cap);
      ws->ones = DevBuf<uint32_t>(n);
      ws->counters = DevBuf<uint32_t>(2 * sizeof(MsmCounters) / 4);
      ws->counters.zero();
      HIP_CHECK(hipEventCreateWithFlags(&ws->sorted, hipEventDisableTiming)); }
    max_tasks = (uint32_t)((n * (size_t)W * 2) / MSM_TASK + (size_t)WB * NB + 1);
    // (tuning knob: entries per lane of the H accumulation)
    // Short runs mean more pieces for the combine, long runs a longer drain at the end of the launch (waves are handed out as slots fall free: alone on the
    // chip, runs of 11 lose nothing to the last round, runs of 14 cost 25 us) — and inside a proof the witness MSMs hold part of the chip meanwhile.  Swept in steps of one inside whole proofs, on the prover's own device clock, after the wave priorities of round 5 changed what else holds the chip
    // meanwhile (profiles/r05_hacc_sweeps.txt, two passes of 100 proofs per point): send 0.751-0.764 ms at 11, 0.737-0.747 at 14; mint / redeem flat within
    // noise from 11 to 16; deposit at depth 8 (8.4 M entries) best at 11 (1.249 against 1.260-1.264 at 14); deposit at depth 32 (18.9 M) 2.97 at 11-13,
    // 2.89-2.91 at 15. (A stand-alone measurement of the MSM picks 13 everywhere: wrong inside a proof, where the witness MSMs hold part of the chip.)
    {
      const size_t total = n * (size_t)W;
      h_run = total > ((size_t)12 << 20) ? 15 : total > ((size_t)6 << 20) ? 11 : 14;
    }
    if (const char *e = getenv("ZK_MSM_H_RUN")) {
      const int v = atoi(e);
      if (v >= 4 && v <= 64) h_run = (uint32_t)v;
    }
    // (G1, uniform scalars — G2 keeps a 261-form table for another purpose: k_wacc_lanes_g2_29) group-binned one-pass sort: G groups of 2^low buckets, about 16
    // K entries per group (one workgroup sorts a group in registers + LDS)
    if (sizeof(F) == 32 && uniform_hint && bases->points261.size()) {
      size_t total = n * (size_t)W;
      uint32_t G = 256;
      while (G < HSORT_GROUPS && total / G > 16384) G <<= 1;   // (as large as a workgroup's registers + LDS take: groups of 8 K / 4 K entries cost a send proof 9 / 24 us)
      uint32_t low = 0;
      while ((G << low) < NB) low++;
      uint32_t ib = 1;
      while (((size_t)1 << ib) < total) ib++;
      // (test hook: regions far too small force the overflow fallback)
      size_t region = ((total / G) * 5 / 4 + 1024 + 255) & ~(size_t)255;
      if (getenv("ZK_MSM_DIRECT_CAP")) region = 256;
      if (W <= (int)HSORT_STAGE_W && (G << low) == NB && low >= 1 && low <= 10 && low + 1 + ib <= 32 &&
          region <= (size_t)HSORT_GROUP_THREADS * HSORT_MAX_PER_THREAD) {
        // (at least n W entries: the general path, should this one fall back to it, sorts into the same array)
        hsort = true;
        hs = HsortShape{G, low, ib, (uint32_t)region};
        group_fill = DevBuf<uint32_t>(G);
        group_fill.zero();
        group_n = DevBuf<uint32_t>(G);
        group_n.zero();
        mid = DevBuf<uint32_t>((size_t)G * region);
        entries = DevBuf<uint32_t>(std::max((size_t)G * region, total));
        // pieces per bucket: room for 1.5x the expected load + 32 entries
        const size_t lam = total / NB;
        h_maxp = (uint32_t)((lam + lam / 2 + 32 + h_run - 1) / h_run + 2);
      }
    }
    {
      size_t nbk = (size_t)WB * NB;
      bsort_blocks = cdiv(nbk, BSORT_BLOCK);
      order = DevBuf<uint32_t>(nbk);
      rank_of = DevBuf<uint32_t>(nbk);
      block_hist = DevBuf<uint32_t>((size_t)bsort_blocks * BSORT_CLASSES);
      block_off = DevBuf<uint32_t>((size_t)bsort_blocks * BSORT_CLASSES);
      bsort_scanner.reset(new Scanner((size_t)bsort_blocks * BSORT_CLASSES));
    }
    buckets = DevBuf<uint8_t>((size_t)WB * NB * sizeof(XYZZ<F>));
    partials = DevBuf<uint8_t>(std::max<size_t>(std::max<size_t>((size_t)max_tasks * sizeof(XYZZ<F>), hsort ? (size_t)NB * h_maxp * sizeof(Piece29) : 0),
        wfused ? ((size_t)WFUSED_BUCKET_LANES + WFUSED_ONES_LANES) * std::max(std::max(sizeof(XYZZ<F>), sizeof(Point29Rec)), sizeof(F) == 64 ? sizeof(Point29Rec2) : (size_t)0) : 0));
    seg_out = DevBuf<uint8_t>(std::max<size_t>((size_t)WB * (NB / seg), (size_t)32 * cdiv(NB, 512)) * sizeof(XYZZ<F>));
    seg_l2 = DevBuf<uint8_t>((size_t)WB * cdiv(NB / seg, GROUP) * sizeof(XYZZ<F>));
    ones_partial = DevBuf<uint8_t>(std::max<size_t>(n_ones_quads, wfused ? (size_t)NB + WFUSED_ONES_GROUPS : 0) * std::max(std::max(sizeof(XYZZ<F>),
        sizeof(Point29Rec)), sizeof(F) == 64 ? sizeof(Point29Rec2) : (size_t)0));
    if (wfused) lane_off = DevBuf<uint32_t>(WFUSED_MAX_BUCKETS + 1);
    ones_l2 = DevBuf<uint8_t>((size_t)cdiv(n_ones_quads, GROUP) * sizeof(XYZZ<F>));
    RS = WB; if (wfused) { bitsum = true; RS = WTAIL_SLOTS; }   // k_wtail leaves eight sums by weight bit
    else if (hsort && NB >= 512) { bitsum = true; RS = 1; while ((1u << (RS - 1)) < NB) RS++;   // RS = log2(NB) + 1
      htail29 = true;
      hb29 = DevBuf<uint8_t>((size_t)NB * sizeof(Point29Rec));
      hmarg = DevBuf<uint8_t>((size_t)htail_marg_count(htail_shape(NB)) * sizeof(Point29Rec));
    }
    // (the ones slot of the result stays the point at infinity when the ones path is off: h_result is cleared below)
    zeroed.zero();
    HIP_CHECK(hipHostMalloc((void **)&h_result, result_bytes())); memset(h_result, 0, result_bytes());
    // the last kernel of an MSM writes its few sums straight into the pinned host copy: no copy kernel behind it (1.125 -> 1.10 ms median per proof; round 2
    // had measured no difference, at 1.45 ms)
    {
      void *d = nullptr;
      HIP_CHECK(hipHostGetDevicePointer(&d, h_result, 0));
      res_dev = (uint8_t *)d;
    }
    HIP_CHECK(hipStreamSynchronize(gpu().stream));
  }
  ~MsmImpl() {
    if (h_result) hipHostFree(h_result);
    if (ones_stream) hipStreamDestroy(ones_stream);
    if (ev_classified) hipEventDestroy(ev_classified);
    if (ev_ones) hipEventDestroy(ev_ones);
  }
  void enable_split_ones() {
    if (split_ones || !filter_ones) return;
    HIP_CHECK(hipStreamCreateWithFlags(&ones_stream, hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&ev_classified, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&ev_ones, hipEventDisableTiming));
    split_ones = true;
  }
  hipStream_t stream() { return stream_id < 0 ? gpu().stream : gpu().aux[stream_id & 3]; }
  // after the stream has been synchronised: did a region of a one-pass sort overflow (or did the 29-bit accumulation meet an operand equal to +-its
  // accumulator)? Then repeat the last run on the general two-pass path (synchronously): any input stays correct, only the expected ones are fast.
  void finish_sync() { HIP_CHECK(hipStreamSynchronize(stream()));
    if (wfused && host_counters()->pad[0]) {
      if (!overflow_noted) {
        overflow_noted = true;
        const uint32_t why = host_counters()->pad[0];
        fprintf(stderr, "libzkgpu: %s: %s%s(flag 0x%x: bits 8.. say which result slots, 23 = the sum of the ones; %u slots a bucket), general MSM path used\n", label.c_str(), (why & 1u) ? "a bucket of the witness sort overflowed " : "",
            (why & 2u) ? "a sum of the fold / tail met an operand equal to +-its partner (ZZ = 0) " : "", why, ws->cap);
      }
      note_general_path_repeat();
      // (diagnostic: ZK_DEBUG_DUMP_DEGENERATE=<file prefix> writes what the fast path held when it raised its flag — bucket fills, entries, lane offsets, the lanes'
      //  partial sums, the bucket sums — for tools/degenerate_dump.py; G1 witness MSMs only)
      if constexpr (sizeof(F) == 32) { static const char *dump = getenv("ZK_DEBUG_DUMP_DEGENERATE"); static std::atomic<int> dumps{0};
        if (dump && dumps.fetch_add(1) < 8) { const int k = dumps.load();
          std::vector<uint32_t> fl(2 * NB), lo(NB + 1), en((size_t)NB * ws->cap); ws->fill.download(fl.data(), fl.size()); lane_off.download(lo.data(), lo.size()); ws->entries.download(en.data(), en.size());
          std::vector<uint8_t> p1((size_t)WFUSED_BUCKET_LANES * sizeof(Point29Rec)), p2((size_t)NB * sizeof(Point29Rec)); HIP_CHECK(hipMemcpy(p1.data(), partials.get(), p1.size(), hipMemcpyDeviceToHost));
          HIP_CHECK(hipMemcpy(p2.data(), ones_partial.get(), p2.size(), hipMemcpyDeviceToHost));
          const std::string path = std::string(dump) + "_" + label + "_" + std::to_string(k) + ".bin"; FILE *f = fopen(path.c_str(), "wb");
          if (f) { const uint32_t hdr[8] = {NB, ws->cap, (uint32_t)ws->parity, host_counters()->pad[0], (uint32_t)n, (uint32_t)WFUSED_BUCKET_LANES, (uint32_t)sizeof(Point29Rec), 0};
            fwrite(hdr, 4, 8, f); fwrite(fl.data(), 4, fl.size(), f); fwrite(lo.data(), 4, lo.size(), f);
            for (uint32_t b = 0; b < NB; b++) fwrite(en.data() + (size_t)b * ws->cap, 4, std::min(std::max(fl[b], fl[NB + b]), ws->cap), f);
            fwrite(p1.data(), 1, p1.size(), f); fwrite(p2.data(), 1, p2.size(), f); fclose(f); fprintf(stderr, "libzkgpu: %s: state written to %s\n", label.c_str(), path.c_str()); } } }
      wfused = false; run_impl(last_scalars, last_index); HIP_CHECK(hipStreamSynchronize(stream())); wfused = true; }
    if (hsort && host_counters()->pad[0]) { const Fe32 *sc = last_scalars; hsort = false; note_general_path_repeat();
      // hist() held the bucket counts of the group sort; the two-pass path wants it cleared
      HIP_CHECK(hipMemsetAsync(zeroed.get(), 0, 2 * (size_t)WB * NB * sizeof(uint32_t), stream()));
      HIP_CHECK(hipMemsetAsync(group_fill.get(), 0, hs.groups * sizeof(uint32_t), stream()));
      if (prod_b) { if (prod_tmp.size() < n) prod_tmp = DevBuf<Fe32>(n);   // materialise the product for the two-pass path
        hipLaunchKernelGGL(k_fr_mul3, dim3(cdiv(n, 256)), dim3(256), 0, stream(), (const Fr *)last_scalars, (const Fr *)prod_b, (const Fr *)prod_z,
            (int)prod_z_table, (uint32_t)n, (Fr *)prod_tmp.get());
        sc = prod_tmp.get();
      }
      const Fe32 *pb = prod_b; prod_b = nullptr; run_impl(sc, last_index); HIP_CHECK(hipStreamSynchronize(stream())); prod_b = pb; hsort = true; } }
  // sum_i (a_i b_i z) P_i; only on the H path, where the product is formed inside the sort kernel
  void run_product(const Fe32 *a, const Fe32 *b, const Fe32 *z, bool z_is_table) {
    if (!hsort) throw GpuError("msm: run_product needs the one-pass sort");
    prod_b = b;
    prod_z = z;
    prod_z_table = z_is_table;
    run_impl(a, nullptr);
  }
  void run(const Fe32 *scalars, const uint32_t *scalar_index) { prod_b = nullptr; tagged = false; run_impl(scalars, scalar_index); }
  // z_all_: the whole assignment (variable 0 = ONE); a plain query reads its scalars from z_all_ + wt.base, an indexed one through scalar_index
  void run_tagged(const Fe32 *z_all_, const WitnessTags &wt, const uint32_t *scalar_index) {
    prod_b = nullptr; wtags = wt; z_all = z_all_; tagged = wt.tags != nullptr && wt.other_vars != nullptr;
    run_impl(scalar_index ? z_all_ : z_all_ + wt.base, scalar_index); }

  void run_impl(const Fe32 *scalars, const uint32_t *scalar_index) {
    hipStream_t s = stream();
    size_t nbk = (size_t)WB * NB;
    const uint32_t hist_stride = WB == 1 ? 0 : NB, point_stride = WB == 1 && W > 1 ? (uint32_t)n : 0;
    const uint8_t *infp = any_inf ? inf.get() : nullptr;
    const uint32_t bucket_u4 = sizeof(XYZZ<F>) / 16; XYZZ<F> *res = (XYZZ<F> *)res_dev;
    // (histogram and slot counters were cleared by the constructor and are left cleared by every run (k_msm_combine_tasks); the MsmCounters alternate between
    // two slots)
    parity ^= 1; MsmCounters *cnt = counters();
    bool ones_forked = false;
    auto ones_path = [&](hipStream_t os) { Stage st((label + ".ones").c_str(), os); uint32_t g = cdiv(n_ones_quads, GROUP);
      hipLaunchKernelGGL((k_msm_sum_ones<F>), dim3(cdiv((size_t)n_ones_quads * 4, 256)), dim3(256), 0, os, (const Affine<F> *)points.get(), ones.get(), cnt,
          n_ones_quads, (XYZZ<F> *)ones_partial.get());
      hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(g), dim3(256), 0, os, (const XYZZ<F> *)ones_partial.get(), GROUP, n_ones_quads, (XYZZ<F> *)ones_l2.get(),
          (uint4 *)nullptr, (uint4 *)nullptr);
      hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(1), dim3(256), 0, os, (const XYZZ<F> *)ones_l2.get(), g, g, res + RS, (uint4 *)nullptr, (uint4 *)nullptr);
    };
    last_scalars = scalars; last_index = scalar_index;
    if (wfused) {
      // (undo the flip above: this path has its own counters, and the general path — which may follow as the overflow fallback — relies on strict alternation
      // of its two slots)
      parity ^= 1;
      WsortBuffers &w = *ws; MsmCounters *wc = (MsmCounters *)w.counters.get();
      // a shared sort keeps every point: the tables differ in which points are at infinity, and the additions skip those
      if (ws_leader) {
        w.parity ^= 1;
        w.leader_stream = stream_id;
        Stage st((label + ".sort").c_str(), s);
        const uint8_t *winf = w.shared ? nullptr : infp;
        // the sort from the assignment's tags (k_wsort_tagged)
        if (tagged) {
          const uint32_t ones_blocks = cdiv(n, 256), other_blocks = cdiv(wtags.n_other, 256);
#define ZK_CALL(CC) hipLaunchKernelGGL(k_wsort_tagged<CC>, dim3(ones_blocks + other_blocks), dim3(256), 0, s, (const Fr *)z_all, wtags, scalar_index, winf, (uint32_t)n, c, W, point_stride, NB, w.cap, ones_blocks, w.fill.get() + (size_t)w.parity * NB, w.fill.get() + (size_t)(w.parity ^ 1) * NB, w.entries.get(), w.ones.get(), wc + w.parity, wc + (w.parity ^ 1))
          ZK_MSM_DISPATCH_C(c, ZK_CALL);
#undef ZK_CALL
        } else {
#define ZK_CALL(CC) hipLaunchKernelGGL(k_wsort<CC>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const Fr *)scalars, scalar_index, winf, (uint32_t)n, c, W, point_stride, NB, w.cap, w.fill.get() + (size_t)w.parity * NB, w.fill.get() + (size_t)(w.parity ^ 1) * NB, w.entries.get(), w.ones.get(), wc + w.parity, wc + (w.parity ^ 1))
        ZK_MSM_DISPATCH_C(c, ZK_CALL);
#undef ZK_CALL
        }
        if (w.shared) HIP_CHECK(hipEventRecord(w.sorted, s)); }
      else if (w.leader_stream != stream_id) HIP_CHECK(hipStreamWaitEvent(s, w.sorted, 0));   // (a follower on the leader's stream is simply queued behind it)
      uint4 *cdst = (uint4 *)(res + RS + 1);
      // accumulate / fold / tail. G1 (A, L*, B1): on 29-bit limbs throughout (msm.cuh: k_wacc_lanes29; htail29.cuh: k_wfold29, k_wtail29). G2 (B2): the lanes
      // on 29-bit limbs over Fq2 (k_wacc_lanes_g2_29), fold and tail quad-cooperative on 8 x 32-bit limbs (k_wacc_fold<Fq2>, k_wtail<Fq2>).
      const uint32_t *fl = w.fill.get() + (size_t)w.parity * NB;
      const uint32_t top = 31 - (uint32_t)__builtin_clz(NB);
      const dim3 lanes_grid((WFUSED_BUCKET_LANES + WFUSED_ONES_LANES) / 256), fold_grid(NB + WFUSED_ONES_GROUPS);
      if constexpr (sizeof(F) == 32) {
        Point29Rec *p1 = (Point29Rec *)partials.get(), *p2 = (Point29Rec *)ones_partial.get();
        { Stage st((label + ".accumulate").c_str(), s);
          hipLaunchKernelGGL(k_wacc_lanes29<0>, lanes_grid, dim3(256), 0, s, (const Affine<Fq> *)bases->points261.get(),
              (const Affine<Fq> *)bases->groups261.get(), w.entries.get(), fl, w.cap, zk_with_prio(NB, ZKP_WLANES), w.ones.get(), wc + w.parity, p1, lane_off.get());
          hipLaunchKernelGGL(k_wfold29<0>, fold_grid, dim3(256), 0, s, (const Point29Rec *)p1, (const uint32_t *)lane_off.get(), zk_with_prio(NB, ZKP_WIT),
              (uint32_t)WFUSED_BUCKET_LANES, p2);
        }
        { Stage st((label + ".reduce").c_str(), s);
          hipLaunchKernelGGL(k_wtail29<0>, dim3(top + 2), dim3(256), 0, s, (const Point29Rec *)p2, zk_with_prio(NB, ZKP_WIT), (const Point29Rec *)p2 + NB,
              (uint32_t)WFUSED_ONES_GROUPS, (uint32_t)WTAIL_SLOTS, (XYZZ<Fq> *)res, wc + w.parity, cdst, ws_leader ? 0u : 16u);
        }
      } else {
        // round 5: fold and tail on 29-bit limbs with the point spread over eight lanes (oct29.cuh: k_wfold_g2_29, k_wtail_g2_29)
        Point29Rec2 *p1 = (Point29Rec2 *)partials.get(), *p2 = (Point29Rec2 *)ones_partial.get();
        { Stage st((label + ".accumulate").c_str(), s);
          hipLaunchKernelGGL(k_wacc_lanes_g2_29<0>, lanes_grid, dim3(256), 0, s, (const Affine<Fq2> *)bases->points261.get(),
              (const Affine<Fq2> *)bases->groups261.get(), w.entries.get(), fl, w.cap, zk_with_prio(NB, ZKP_WLANES), w.ones.get(), wc + w.parity, p1, lane_off.get());
          hipLaunchKernelGGL(k_wfold_g2_29<0>, fold_grid, dim3(OCT_BLOCK), 0, s, (const Point29Rec2 *)p1, (const uint32_t *)lane_off.get(), zk_with_prio(NB, ZKP_WIT),
              (uint32_t)WFUSED_BUCKET_LANES, p2);
        }
        { Stage st((label + ".reduce").c_str(), s);
          hipLaunchKernelGGL(k_wtail_g2_29<0>, dim3(top + 2), dim3(OCT_BLOCK), 0, s, (const Point29Rec2 *)p2, zk_with_prio(NB, ZKP_WIT), (const Point29Rec2 *)p2 + NB,
              (uint32_t)WFUSED_ONES_GROUPS, (uint32_t)WTAIL_SLOTS, (XYZZ<Fq2> *)res, wc + w.parity, cdst, ws_leader ? 0u : 16u);
        }
      }
      return;
    }
    const bool hs_run = hsort && scalar_index == nullptr;
    if (hs_run) {
      if constexpr (sizeof(F) == 32) {
      { Stage st((label + ".sort").c_str(), s);
        HsortShape hs_prio = hs; hs_prio.low_bits |= zk_prio_bits(ZKP_HSORT);
#define ZK_CALL(CC) hipLaunchKernelGGL(k_hsort_bin<CC>, dim3(cdiv(n, HSORT_TILE)), dim3(HSORT_BIN_THREADS), 0, s, (const Fr *)scalars, (const Fr *)prod_b, (const Fr *)prod_z, (int)prod_z_table, infp, (uint32_t)n, c, W, point_stride, hs_prio, group_fill.get(), mid.get(), cnt, counters_next())
        ZK_MSM_DISPATCH_C(c, ZK_CALL);
#undef ZK_CALL
        static std::atomic<uint64_t> lds_done{0}; zk_raise_dynamic_lds(reinterpret_cast<const void *>(&k_hsort_group), 128 * 1024, lds_done);
        hipLaunchKernelGGL(k_hsort_group, dim3(hs.groups), dim3(HSORT_GROUP_THREADS), 4 * hs.region, s, (const uint32_t *)mid.get(), group_fill.get(), hs_prio, entries.get(),
            hist(), offsets.get(), group_n.get());
      }
      // one lane per run: at most ceil(entries / run) + one short run per group
      {
        Stage st((label + ".accumulate").c_str(), s);
        const uint32_t h_run = crowded ? std::max(this->h_run, h_run_crowded) : this->h_run;   // (pieces are laid out by h_maxp, sized for the shorter run)
        const dim3 grid(cdiv(cdiv(n * (size_t)W, h_run) + hs.groups, 256));
        if (any_inf) hipLaunchKernelGGL(k_hacc_runs29<1>, grid, dim3(256), 0, s, (const Affine<Fq> *)bases->points261.get(), entries.get(), group_n.get(),
            offsets.get(), hs, h_run, zk_with_prio(h_maxp, ZKP_HACC), (Piece29 *)partials.get(), cnt);
        else hipLaunchKernelGGL(k_hacc_runs29<0>, grid, dim3(256), 0, s, (const Affine<Fq> *)bases->points261.get(), entries.get(), group_n.get(),
            offsets.get(), hs, h_run, zk_with_prio(h_maxp, ZKP_HACC), (Piece29 *)partials.get(), cnt);
      }
      // 2 / 4 / 8 lanes per bucket: about six pieces a lane (send: 12 pieces, two lanes; deposit at depth 32: 48 pieces — two lanes took 256 us there)
      {
        Stage st((label + ".combine").c_str(), s);
        const uint32_t h_run = crowded ? std::max(this->h_run, h_run_crowded) : this->h_run;
        const size_t pieces = n * (size_t)W / NB / h_run;
        const uint32_t ll = pieces > 40 ? 3 : pieces > 18 ? 2 : 1;   // (send, runs of 14: 9 pieces; two and four lanes measure the same, eight lose 30 us)
        hipLaunchKernelGGL(k_hacc_combine29, dim3(cdiv(nbk << ll, 256)), dim3(256), 0, s, (const Piece29 *)partials.get(), offsets.get(), hist(), hs, h_run,
            h_maxp, (uint32_t)nbk, zk_with_prio(ll, ZKP_HTAIL), (XYZZ<Fq> *)bucket_array(), htail29 ? (Point29Rec *)hb29.get() : nullptr, cnt);
      }
      }
    } else
    { Stage st((label + ".sort").c_str(), s);
      if (hsort) throw GpuError("msm: the group-binned sort takes no scalar index");
#define ZK_CALL(CC) hipLaunchKernelGGL(k_msm_classify<CC>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const Fr *)scalars, scalar_index, infp, (uint32_t)n, c, W, (int)filter_ones, hist_stride, (uint32_t)nbk, hist(), ones.get(), cnt, counters_next())
      if (n) ZK_MSM_DISPATCH_C(c, ZK_CALL);
#undef ZK_CALL
      if (filter_ones && n && split_ones) {
        HIP_CHECK(hipEventRecord(ev_classified, s));
        HIP_CHECK(hipStreamWaitEvent(ones_stream, ev_classified, 0));
        ones_path(ones_stream);
        HIP_CHECK(hipEventRecord(ev_ones, ones_stream));
        ones_forked = true;
      }
      if (nbk <= PLAN_SMALL_MAX) {
        hipLaunchKernelGGL(k_msm_plan_small, dim3(1), dim3(PLAN_THREADS), 0, s, hist(), (uint32_t)nbk, offsets.get(), order.get(), rank_of.get(),
            task_off.get(), cls_start.get(), (uint4 *)bucket_array(), bucket_u4);
      } else {
        scanner.run(hist(), offsets.get(), nbk, s);
        hipLaunchKernelGGL(k_bsort_hist, dim3(bsort_blocks), dim3(BSORT_BLOCK), 0, s, hist(), (uint32_t)nbk, bsort_blocks, block_hist.get(), 0u);
        bsort_scanner->run(block_hist.get(), block_off.get(), (size_t)bsort_blocks * BSORT_CLASSES, s);
        hipLaunchKernelGGL(k_bsort_scatter, dim3(bsort_blocks), dim3(BSORT_BLOCK), 0, s, hist(), (uint32_t)nbk, bsort_blocks, block_off.get(), order.get(),
            rank_of.get(), ntasks.get(), cls_start.get(), (uint4 *)bucket_array(), bucket_u4);
        task_scanner.run(ntasks.get(), task_off.get(), nbk + 1, s);
      }
#define ZK_CALL(CC) hipLaunchKernelGGL(k_msm_scatter<CC>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const Fr *)scalars, scalar_index, infp, (uint32_t)n, c, W, (int)filter_ones, hist_stride, point_stride, (uint32_t)nbk, offsets.get(), fill(), entries.get())
      if (n) ZK_MSM_DISPATCH_C(c, ZK_CALL);
#undef ZK_CALL
    }
    if (!hs_run) {
      { Stage st((label + ".accumulate").c_str(), s);
        if (bases->table261_only) hipLaunchKernelGGL((k_msm_accumulate_tasks<F, 1>), dim3(cdiv(max_tasks, 256)), dim3(256), 0, s, (const Affine<F> *)bases->points261.get(), entries.get(),
            offsets.get(), hist(), order.get(), task_off.get(), (uint32_t)nbk, max_tasks, MSM_TASK, bucket_array(), (XYZZ<F> *)partials.get());
        else hipLaunchKernelGGL((k_msm_accumulate_tasks<F, 0>), dim3(cdiv(max_tasks, 256)), dim3(256), 0, s, (const Affine<F> *)points.get(), entries.get(),
            offsets.get(), hist(), order.get(), task_off.get(), (uint32_t)nbk, max_tasks, MSM_TASK, bucket_array(), (XYZZ<F> *)partials.get());
      }
      { Stage st((label + ".combine").c_str(), s);
        hipLaunchKernelGGL((k_msm_combine_tasks<F>), dim3(HEAVY_BLOCKS + cdiv(nbk, 64)), dim3(256), 0, s, order.get(), task_off.get(), cls_start.get(),
            HEAVY_BLOCKS, (const XYZZ<F> *)partials.get(), bucket_array(), zeroed.get(), (uint32_t)(2 * nbk), 0);
      }
    }
    // marginal sums, then the sums by weight bit (htail29.cuh)
    if (bitsum && hs_run && htail29) {
      if constexpr (sizeof(F) == 32) {
        Stage st_red((label + ".reduce").c_str(), s);
      const HtailShape ts = htail_shape(NB);
      // one wave per row piece / column of up to 128 buckets (7 additions in a row, then the wave's tree of 4): as fast as four waves with 1 + 6 — a wave alone
      // on its SIMD issues an instruction every 6-8 cycles whatever it does — at 40 % of their instructions, which is what counts with other proofs in flight
      // (profiles/r05_hacc_sweeps.txt)
      const uint32_t marg_block = std::max(1u << ts.hi_bits, (1u << ts.lo_bits) / ts.row_chunks) <= 128 ? 64 : 256;
      hipLaunchKernelGGL(k_hmarg29<0>, dim3(htail_marg_blocks(ts)), dim3(marg_block), 0, s, (const Point29Rec *)hb29.get(), zk_with_prio(NB, ZKP_HTAIL), (Point29Rec *)hmarg.get());
      hipLaunchKernelGGL(k_hbits29<0>, dim3(ts.top + 1), dim3(256), 0, s, (const Point29Rec *)hmarg.get(), zk_with_prio(NB, ZKP_HTAIL), (XYZZ<Fq> *)res, cnt, (uint4 *)(res + RS + 1));
    }
  }
    else { Stage st_red((label + ".reduce").c_str(), s);
      // (a bit-sum MSM on its fallback path: one window sum in slot 0, the other slots at infinity; the host may write here: the run that used these slots was
      // synchronised before this one started)
      if (RS > WB) memset(h_result + (size_t)WB * sizeof(XYZZ<F>), 0, (size_t)(RS - WB) * sizeof(XYZZ<F>));
      uint32_t spw = NB / seg, nseg = (uint32_t)WB * spw; uint4 *csrc = (uint4 *)cnt; uint4 *cdst = (uint4 *)(res + RS + 1);
      hipLaunchKernelGGL((k_msm_reduce_segments<F>), dim3(cdiv(nseg, 16)), dim3(64), 0, s, (const XYZZ<F> *)bucket_array(), NB, seg, nseg,
          (XYZZ<F> *)seg_out.get());
      if (spw > GROUP) { uint32_t g = spw / GROUP;   // two-level tree per window keeps the dependent chain short (spw is a power of two)
        hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(WB * g), dim3(256), 0, s, (const XYZZ<F> *)seg_out.get(), GROUP, nseg, (XYZZ<F> *)seg_l2.get(),
            (uint4 *)nullptr, (uint4 *)nullptr);
        hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(WB), dim3(256), 0, s, (const XYZZ<F> *)seg_l2.get(), g, (uint32_t)WB * g, res, csrc, cdst);
      } else hipLaunchKernelGGL((k_xyzz_group_sum<F>), dim3(WB), dim3(256), 0, s, (const XYZZ<F> *)seg_out.get(), spw, nseg, res, csrc, cdst);
    }
    if (filter_ones && n && !split_ones) ones_path(s);
    if (ones_forked) HIP_CHECK(hipStreamWaitEvent(s, ev_ones, 0));
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

}  // namespace zk
