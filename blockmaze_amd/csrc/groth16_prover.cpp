// Groth16 host code, part 3 of 4: the prover (hand-over of the assignment, the device pipeline's submission, proof assembly, shards, stashes).
// see groth16.hpp
#include <sched.h>
#include <sys/random.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <array>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <climits>
#include "groth16_common.hpp"
#include "verify_sched.hpp"

namespace zk {
// ======================================================================================================================
// prover
// ====================================================================================================================== A helper thread that lives as long as
// its prover: submitting a witness MSM (about a dozen launches, several microseconds of host time each) must not cost a thread creation per proof on the
// critical path. post() hands over a job, wait() blocks until it has run and rethrows what it threw. CPUs this process may run on (its affinity mask, not the
// machine's size): a rank that a launcher pinned to two cores of a 256-thread host must not start sixteen polling helpers.  A cgroup CPU quota counts as well
// (cpu.max of cgroup v2, cpu.cfs_quota_us / cpu.cfs_period_us of v1): go-ethereum under a Kubernetes CPU limit without a cpuset sees every CPU of the host in its
// affinity mask, and fifteen spinning helpers would get the process throttled on the proof's critical path.
static unsigned cgroup_cpu_quota(const char *root = "/sys/fs/cgroup") {   // 0: none
  const std::string r(root), v2 = r + "/cpu.max", v1q = r + "/cpu/cpu.cfs_quota_us", v1p = r + "/cpu/cpu.cfs_period_us";
  auto read2 = [](const char *path, long long &a, long long &b, bool two) -> bool {
    FILE *f = fopen(path, "r"); if (!f) return false; char t[64] = {0}; bool ok;
    if (two) { ok = fscanf(f, "%63s %lld", t, &b) == 2; if (ok) { if (!strcmp(t, "max")) a = -1; else a = atoll(t); } }
    else ok = fscanf(f, "%lld", &a) == 1;
    fclose(f); return ok; };
  long long quota = -1, period = 0;
  if (read2(v2.c_str(), quota, period, true)) { if (quota > 0 && period > 0) return (unsigned)std::max<long long>(1, (quota + period - 1) / period); return 0; }
  if (read2(v1q.c_str(), quota, period, false) && quota > 0) { long long per = 0, dummy = 0;
    if (read2(v1p.c_str(), per, dummy, false) && per > 0) return (unsigned)std::max<long long>(1, (quota + per - 1) / per); }
  return 0;
}
int test_cgroup_quota(const char *root) { return (int)cgroup_cpu_quota(root); }   // (host-only test hook: the quota a cgroup directory tree states, 0 = none)
static unsigned usable_cpus() {
  static const unsigned v = [] {
    unsigned n = 0; cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = (unsigned)CPU_COUNT(&set);
    if (!n) { const unsigned hw = std::thread::hardware_concurrency(); n = hw ? hw : 1u; }
    const unsigned q = cgroup_cpu_quota();
    return q && q < n ? q : n;
  }();
  return v;
}
class SubmitWorker {
 public:
  explicit SubmitWorker(int lane) : lane_(lane), th_([this] { loop(); }) {}
  ~SubmitWorker() { { std::lock_guard<std::mutex> lk(m_); quit_ = true; } posted_.fetch_add(1, std::memory_order_release); cv_.notify_all(); th_.join(); }
  // (both flags change under the lock: the worker clears running_ under it too, so a job that is picked up by a spurious wake-up cannot finish before running_
  // is set)
  void post(std::function<void()> job) {
    {
      std::lock_guard<std::mutex> lk(m_);
      job_ = std::move(job);
      busy_ = true;
      err_ = nullptr;
      running_.store(true, std::memory_order_release);
      posted_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
  }
  void wait() {
    spin([this] { return !running_.load(std::memory_order_acquire); });
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [this] { return !busy_; });
    if (err_) {
      std::exception_ptr e = err_;
      err_ = nullptr;
      std::rethrow_exception(e);
    }
  }
 private:
  // A proof hands this thread two jobs a fraction of a millisecond apart (its share of the hand-over scan, then a witness MSM), and the next proof follows as
  // soon: the thread polls for SPIN_US before it goes to sleep on the condition variable, and so does a waiter — a futex wake-up costs 10-50 us, on the
  // critical path every time. An idle prover sleeps.
  // ZK_SPIN_US overrides the 250 us; a host with fewer than four cores gets 0 (no polling: there the spinners would compete with the threads that submit
  // kernels).
  static int spin_us() {
    static const int v = [] {
      const char *e = getenv("ZK_SPIN_US");
      if (e) return std::max(0, atoi(e));
      return usable_cpus() >= 4 ? 250 : 0;
    }();
    return v;
  }
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  template <class Pred> static void spin(Pred ready) {
    const int us = spin_us();
    if (us <= 0) return;
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(us);
    for (int k = 0; !ready(); k++) { if ((k & 63) == 63 && std::chrono::steady_clock::now() > t_end) return; cpu_relax(); } }
  void loop() { LaneScope lane_scope(lane_); std::unique_lock<std::mutex> lk(m_); uint32_t seen = 0;
    for (;;) { if (!(quit_ || (busy_ && job_))) { lk.unlock(); spin([&] { return posted_.load(std::memory_order_acquire) != seen; }); lk.lock(); }
      cv_.wait(lk, [this] { return quit_ || (busy_ && job_); });
      if (quit_) return;
      seen = posted_.load(std::memory_order_acquire);
      std::function<void()> j = std::move(job_);
      job_ = nullptr;
      lk.unlock();
      std::exception_ptr e;
      try {
        j();
      }
      catch (...) {
        e = std::current_exception();
      }
      lk.lock();
      err_ = e;
      busy_ = false;
      running_.store(false, std::memory_order_release);
      done_.notify_all();
    }
  }
  int lane_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  std::function<void()> job_;
  bool busy_ = false, quit_ = false;
  std::atomic<bool> running_{false};
  std::atomic<uint32_t> posted_{0};
  std::exception_ptr err_;
  std::thread th_;
};
struct Prover::Impl {
  bool h_lagrange = false;                                     // the H query is held in the coset's Lagrange basis: no inverse coset transform per proof
  // the C polynomial's share of the H term lives in the (extended) L query: A and B are the only vectors transformed
  bool c_fold = false;
  int lane = 0;                                                // this prover's stream set: provers on different lanes overlap on the device
  size_t nv, ni, m; size_t a0 = 0, l0 = 0, b0 = 0, h0 = 0;   // first element of this shard in each query
  HG1 alpha_g1, beta_g1, delta_g1; HG2 beta_g2, delta_g2;
  std::unique_ptr<MsmG1> A, B1, H, L;
  std::unique_ptr<MsmG2> B2;
  std::unique_ptr<R1csDev> cs;
  std::unique_ptr<Domain> dom;
  std::shared_ptr<DevBuf<uint32_t>> B_idx;
  DevBuf<Fe32> z, abc;
  DevBuf<uint8_t> packed, tags;
  DevBuf<uint32_t> other_vars;
  std::shared_ptr<DevBuf<uint32_t>> B_pos /* inverse of the B query's index list */;
  uint32_t n_other = 0;
  bool tags_valid = false /* the assignment on the device came in compact form: tags holds 0 / 1 / 2 per variable */;
  // assignments kept in HBM (Prover::stash_witness): the RAW vector only — n x 32 B, Montgomery form as libsnark holds it; tags and the list of other values are
  // derived on the device inside every prove_stashed call.  z_cur: the vector the running proof reads (this prover's z, or a stash in place — no copy)
  struct Stash { DevBuf<Fe32> z; };
  std::vector<std::unique_ptr<Stash>> stashes;
  const Fe32 *z_cur = nullptr; bool z_set = false;
  bool one_stream = false;                                     // diagnostic (ZK_MSM_ONE_STREAM at construction): every kernel on the main stream, submitted by the calling thread in order
  // groups of variables with equal columns (equal_column_groups; k_merge_equal_columns folds each into one place at the head of every proof); shared by the clones
  std::shared_ptr<DevBuf<uint32_t>> merge_ptr, merge_mem; size_t n_merge_groups = 0;
  // set_witness_board: the board's candidates (variables that ever held a value other than 0 / 1), as of the board's mark counter cand_marks
  std::vector<uint32_t> cand; DevBuf<uint32_t> cand_dev; uint32_t cand_marks = 0; bool cand_valid = false;
  DevBuf<uint32_t> other_count;                                // two words, alternating: the length of a list made on the device (k_classify_witness)
  int classify_parity = 0; bool n_other_on_device = false;
  PinnedBuf<Fe32> z_host;
  std::unique_ptr<SubmitWorker> workers[4];
  // submit thread t (1 .. 3) of this prover, idle while the assignment is handed over: the scan's helpers when the process's ScanPool is taken by another
  // prover
  SubmitWorker &scan_worker(size_t t) {
    if (!workers[t]) workers[t].reset(new SubmitWorker(lane));
    return *workers[t];
  }
  // The submit thread of a witness MSM also waits for its stream and finishes the MSM on the host (Horner combine, or the host tail of msm_impl.hpp): four
  // threads do that side by side while the H chain is still running. pending[j]: job j (order B2, L, A, B1) was posted and its result slot is not valid before
  // workers[j]->wait().
  HG2 rB2; HG1 rL, rA, rB1; bool pending[4] = {false, false, false, false}, inline_result[4] = {false, false, false, false};
  // L* rides on A's sort and job, B2 on B1's (same scalar vectors: msm.cuh, "witness MSMs in three launches")
  bool pair_AL = false, pair_B = false, b2_first = false;
  int owner(int j) const { return j == 1 && pair_AL ? 2 : j == 0 && pair_B ? 3 : j; }
  void settle(int j) {
    const int o = owner(j);
    if (pending[o]) {
      pending[o] = false;
      workers[o]->wait();
    }
    if (inline_result[j]) {
      inline_result[j] = false;
      switch (j) {
        case 0: rB2 = B2->result();
        break;
        case 1: rL = L->result();
        break;
        case 2: rA = A->result();
        break;
        default: rB1 = B1->result();
      }
    }
  }
  void settle_all_quietly() { for (int j = 0; j < 4; j++) { try { settle(j); } catch (...) {} } }
  ~Impl() { settle_all_quietly(); for (auto &w : workers) w.reset(); gpu_lane_release(lane); }
};
static int env_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
static void shard_range(size_t n, size_t rank, size_t world, size_t &b, size_t &e) {
  size_t base = n / world, rem = n % world;
  b = rank * base + (rank < rem ? rank : rem);
  e = b + base + (rank < rem ? 1 : 0);
}
// Auxiliary variables whose columns are identical in A, B and C (same rows, same coefficients): groups of two or more, members ascending.  One pass over the
// matrices hashes every column (row, coefficient, matrix in row order), equal hashes are then compared entry by entry.
std::vector<std::vector<uint32_t>> equal_column_groups(const R1csHost &cs) {
  const size_t nv = cs.n_vars + 1; std::vector<uint64_t> h(nv, 0); std::vector<uint32_t> cnt(nv, 0);
  auto mix = [](uint64_t a, uint64_t b) { a ^= b + 0x9e3779b97f4a7c15ull + (a << 6) + (a >> 2); a *= 0xff51afd7ed558ccdull; return a ^ (a >> 33); };
  for (int m = 0; m < 3; m++) for (size_t r = 0; r < cs.n_cons; r++) for (uint32_t k = cs.rowptr[m][r]; k < cs.rowptr[m][r + 1]; k++) {
    const uint32_t c = cs.col[m][k]; if (c >= nv) throw std::runtime_error("r1cs: column index"); uint64_t e = mix((uint64_t)m << 40 | r, 0);
    const uint32_t *w = cs.coeff[m][k].l; for (int i = 0; i < 8; i += 2) e = mix(e, (uint64_t)w[i] | (uint64_t)w[i + 1] << 32);
    h[c] = mix(h[c], e); cnt[c]++; }
  std::vector<uint32_t> cand; for (size_t v = cs.n_inputs + 1; v < nv; v++) if (cnt[v]) cand.push_back((uint32_t)v);
  std::sort(cand.begin(), cand.end(), [&](uint32_t a, uint32_t b) { return h[a] != h[b] ? h[a] < h[b] : a < b; });
  std::vector<uint32_t> suspects; for (size_t i = 0; i < cand.size(); i++) if ((i && h[cand[i - 1]] == h[cand[i]]) || (i + 1 < cand.size() && h[cand[i + 1]] == h[cand[i]])) suspects.push_back(cand[i]);
  if (suspects.empty()) return {};
  // the suspects' columns, explicitly: (matrix, row, coefficient) in order
  std::vector<int> which(nv, -1); for (size_t i = 0; i < suspects.size(); i++) which[suspects[i]] = (int)i;
  struct Ent { uint32_t m, r; Fe32 c; }; std::vector<std::vector<Ent>> cols(suspects.size());
  for (uint32_t m = 0; m < 3; m++) for (size_t r = 0; r < cs.n_cons; r++) for (uint32_t k = cs.rowptr[m][r]; k < cs.rowptr[m][r + 1]; k++) { const int w = which[cs.col[m][k]];
    if (w >= 0) cols[w].push_back(Ent{m, (uint32_t)r, cs.coeff[m][k]}); }
  auto same = [&](int a, int b) { if (cols[a].size() != cols[b].size()) return false;
    for (size_t i = 0; i < cols[a].size(); i++) if (cols[a][i].m != cols[b][i].m || cols[a][i].r != cols[b][i].r || memcmp(&cols[a][i].c, &cols[b][i].c, 32)) return false; return true; };
  std::vector<std::vector<uint32_t>> groups; std::vector<char> used(suspects.size(), 0);
  for (size_t i = 0; i < suspects.size(); i++) { if (used[i]) continue; std::vector<uint32_t> g{suspects[i]};
    for (size_t j = i + 1; j < suspects.size() && h[suspects[j]] == h[suspects[i]]; j++) if (!used[j] && same((int)i, (int)j)) { used[j] = 1; g.push_back(suspects[j]); }
    if (g.size() > 1) { std::sort(g.begin(), g.end()); groups.push_back(g); } }
  std::sort(groups.begin(), groups.end());
  return groups;
}
// streams, labels and the per-object vectors (everything that is not shared between the provers of one key)
static void finish_setup(Prover::Impl &p) {
  // diagnostic: everything on the main stream, so that a kernel trace shows every kernel's stand-alone duration
  const bool one_stream = p.one_stream = env_int("ZK_MSM_ONE_STREAM", 0) != 0;
  // MSMs over the same scalars share one sort: L* follows A, the two halves of the B query follow each other
  {
    p.pair_AL = p.c_fold && p.a0 == p.l0 && p.L->share_sort_with(p.A->sort_handle());
    // Since the end of round 4 the G1 half leads and the G2 half follows (round 4's measurement, profiles/r04y_b1_first.txt; the switch is gone): both halves are then done 0.65 ms into the
    // call instead of 0.76 — less of the chain falls beside the H accumulation, which stretched it —, the host has its last scalar multiple (r * B1) ready 0.16 instead of
    // 0.06 ms before the device finishes, and the device side is 4 us shorter (tools/trace_tail.py, profiles/r04y_b1_first.txt)
    const bool b1_first = true;
    p.b2_first = !b1_first;
    p.pair_B = b1_first ? p.B2->share_sort_with(p.B1->sort_handle()) : p.B1->share_sort_with(p.B2->sort_handle());
  }
  // the witness MSMs only need z: they overlap the QAP / H chain on the main stream
  if (!one_stream) {
    // (the follower of a shared sort stays on its leader's stream: on a stream of its own — four witness chains side by side, measured in round 5 — the H
    // accumulation gets the chip to itself again, 0.33 -> 0.29 ms, but the head of the chain loses more: device side 0.759 -> 0.827 ms, profiles/r05_priorities.txt)
    p.A->set_stream(0);
    p.L->set_stream(p.pair_AL ? 0 : 1);
    p.B1->set_stream(2);
    p.B2->set_stream(p.pair_B ? 2 : 3);
    if (!p.pair_B) p.B2->split_ones_path();
  }
  p.A->set_label("msm_A"); p.L->set_label("msm_L"); p.B1->set_label("msm_B1"); p.B2->set_label("msm_B2"); p.H->set_label("msm_H");
  p.z = DevBuf<Fe32>(p.nv + 1);
  p.abc = DevBuf<Fe32>(3 * p.m);
  p.z_host = PinnedBuf<Fe32>(p.nv + 1 + 8);
  p.packed = DevBuf<uint8_t>(32 * (p.nv + 1 + 8));
  p.tags = DevBuf<uint8_t>(p.nv + 1 + 64);
  // (room for every variable: a list made on the device — k_classify_witness — cannot ask the host to take another path when an assignment is not mostly bits)
  p.other_vars = DevBuf<uint32_t>(p.nv + 1 + 64);
  p.other_count = DevBuf<uint32_t>(2); const uint32_t zero2[2] = {0, 0}; p.other_count.upload(zero2, 2);
  p.z_cur = p.z.get();
}
Prover::Prover(const ProvingKeyHost &pk, size_t shard_rank, size_t shard_world, int device_slot) : impl(new Impl) {
  Impl &p = *impl;
  p.lane = gpu_lane_acquire(device_slot);
  LaneScope lane_scope(p.lane);
  p.nv = pk.cs.n_vars;
  p.ni = pk.cs.n_inputs;
  if (shard_world == 0 || shard_rank >= shard_world) throw std::runtime_error("prover: bad shard");
  p.cs.reset(new R1csDev(pk.cs));
  p.dom.reset(new Domain(pk.cs.n_cons + p.ni + 1));
  p.m = p.dom->m();
  // a key from the container carries H and L only in their transformed form
  const bool transformed = pk.H_lagrange.size() == p.m && pk.L_star.size() == p.nv + 1;
  if (pk.A.size() != p.nv + 1 || (!transformed && (pk.H.size() != p.m - 1 || pk.L.size() != p.nv -
      p.ni))) throw std::runtime_error("proving key: query sizes do not match the constraint system");
  if (transformed && pk.H.empty() && !(env_int("ZK_H_LAGRANGE", 1) != 0 && env_int("ZK_FOLD_C",
      1) != 0)) throw std::runtime_error("proving key: loaded from a container of transformed queries, which ZK_H_LAGRANGE=0 / ZK_FOLD_C=0 cannot use (set ZK_KEY_CACHE=0)");
  p.alpha_g1 = g1_of(pk.alpha_g1);
  p.beta_g1 = g1_of(pk.beta_g1);
  p.delta_g1 = g1_of(pk.delta_g1);
  p.beta_g2 = g2_of(pk.beta_g2);
  p.delta_g2 = g2_of(pk.delta_g2);
  int cw = env_int("ZK_MSM_WITNESS_WINDOW", 8), ch = env_int("ZK_MSM_H_WINDOW", 16);
  size_t e; shard_range(pk.A.size(), shard_rank, shard_world, p.a0, e); size_t nA = e - p.a0;
  // H query: in the coset's Lagrange basis when the domain allows it (then the seventh transform of every proof is skipped, ecntt.cuh); computed once per key
  // object
  shard_range(pk.B_idx.size(), shard_rank, shard_world, p.b0, e);
  size_t nB = e - p.b0;
  const std::vector<G1AffineRaw> *Hq = &pk.H; p.h_lagrange = env_int("ZK_H_LAGRANGE", 1) != 0 && p.dom->supports_h_lagrange();
  if (p.h_lagrange) {
    if (pk.H_lagrange.size() != p.m) {
      pk.H_lagrange.resize(p.m);
      p.dom->h_query_to_coset_lagrange(pk.H.data(), pk.H.size(), pk.H_lagrange.data());
    }
    Hq = &pk.H_lagrange;
  }
  shard_range(Hq->size(), shard_rank, shard_world, p.h0, e); size_t nH = e - p.h0;
  // L query: extended to all variables with the C polynomial folded in when the domain allows it (ecntt.cuh); computed once per key object
  const std::vector<G1AffineRaw> *Lq = &pk.L; p.c_fold = p.h_lagrange && env_int("ZK_FOLD_C", 1) != 0 && p.dom->supports_c_fold();
  if (p.c_fold) {
    if (pk.L_star.size() != p.nv + 1) {
      pk.L_star.resize(p.nv + 1);
      p.dom->fold_c_into_l(pk.H_lagrange.data(), pk.cs, pk.L.data(), pk.L_star.data());
    }
    Lq = &pk.L_star;
  }
  shard_range(Lq->size(), shard_rank, shard_world, p.l0, e); size_t nL = e - p.l0;
  p.A.reset(new MsmG1(pk.A.data() + p.a0, nA, cw, true)); p.L.reset(new MsmG1(Lq->data() + p.l0, nL, cw, true));
  // (with tables the H accumulation gathers from a table 16x larger, but the weighted bucket sum shrinks by the number of windows)
  p.B1.reset(new MsmG1(pk.B_g1.data() + p.b0, nB, cw, true));
  p.B2.reset(new MsmG2(pk.B_g2.data() + p.b0, nB, cw, true));
  p.H.reset(new MsmG1(Hq->data() + p.h0, nH, ch, false, env_int("ZK_MSM_H_TABLES", 1) != 0, true));
  finish_setup(p);
  p.B_idx = std::make_shared<DevBuf<uint32_t>>(pk.B_idx.size() + 1); if (!pk.B_idx.empty()) p.B_idx->upload(pk.B_idx.data(), pk.B_idx.size());
  // variable -> its position in the B query (k_wsort_tagged)
  {
    std::vector<uint32_t> pos(p.nv + 1, 0xffffffffu);
    for (size_t j = 0; j < pk.B_idx.size(); j++) pos[pk.B_idx[j]] = (uint32_t)j;
    p.B_pos = std::make_shared<DevBuf<uint32_t>>(pos.size()); p.B_pos->upload(pos.data(), pos.size()); }
  if (env_int("ZK_MERGE_EQUAL_COLUMNS", 1) != 0) {
    const std::vector<std::vector<uint32_t>> groups = zk::equal_column_groups(pk.cs); std::vector<uint32_t> ptr{0}, mem;
    for (const auto &g : groups) { mem.insert(mem.end(), g.begin(), g.end()); ptr.push_back((uint32_t)mem.size()); }
    p.n_merge_groups = groups.size();
    if (p.n_merge_groups) { p.merge_ptr = std::make_shared<DevBuf<uint32_t>>(ptr.size()); p.merge_ptr->upload(ptr.data(), ptr.size());
      p.merge_mem = std::make_shared<DevBuf<uint32_t>>(mem.size()); p.merge_mem->upload(mem.data(), mem.size()); }
  }
}
Prover::Prover(const Prover &peer) : impl(new Impl) {
  // same device as the peer: the shared tables live there
  Impl &p = *impl;
  const Impl &o = *peer.impl;
  p.lane = gpu_lane_acquire(gpu_slot_of_lane(o.lane));
  LaneScope lane_scope(p.lane);
  p.h_lagrange = o.h_lagrange; p.c_fold = o.c_fold; p.nv = o.nv; p.ni = o.ni; p.m = o.m; p.a0 = o.a0; p.l0 = o.l0; p.b0 = o.b0; p.h0 = o.h0;
  p.alpha_g1 = o.alpha_g1; p.beta_g1 = o.beta_g1; p.delta_g1 = o.delta_g1; p.beta_g2 = o.beta_g2; p.delta_g2 = o.delta_g2;
  p.cs.reset(new R1csDev(*o.cs)); p.dom.reset(new Domain(*o.dom)); p.B_idx = o.B_idx; p.B_pos = o.B_pos;
  p.merge_ptr = o.merge_ptr; p.merge_mem = o.merge_mem; p.n_merge_groups = o.n_merge_groups;
  p.A.reset(new MsmG1(*o.A, true, false));
  p.L.reset(new MsmG1(*o.L, true, false));
  p.B1.reset(new MsmG1(*o.B1, true, false));
  p.B2.reset(new MsmG2(*o.B2, true, false));
  p.H.reset(new MsmG1(*o.H, false, true));
  finish_setup(p);
}
Prover::~Prover() { if (impl) { LaneScope lane_scope(impl->lane); try { gpu_sync(); } catch (...) {} impl.reset(); } }
int Prover::device_slot() const { return gpu_slot_of_lane(impl->lane); }
size_t Prover::num_variables() const { return impl->nv; }
size_t Prover::num_inputs() const { return impl->ni; }
size_t Prover::domain_size() const { return impl->m; }

// 64 consecutive field elements -> two bit masks: "equals one" and "neither zero nor one" (Prover::set_witness). The scan of a 7.3 MB assignment is on the
// critical path of every host-buffer proof; with 256-bit loads an element is three instructions instead of a dozen 64-bit ones.
static void classify_block64_scalar(const uint64_t *v, const uint64_t (&o1)[4], uint64_t &mo, uint64_t &mx) {
  for (size_t i = 0; i < 64; i++, v += 4) {
    const uint64_t nz = (v[0] | v[1] | v[2] | v[3]) != 0, is1 = ((v[0] ^ o1[0]) | (v[1] ^ o1[1]) | (v[2] ^ o1[2]) | (v[3] ^ o1[3])) == 0;
    mo |= is1 << i; mx |= (nz & (is1 ^ 1)) << i; } }
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void classify_block64_avx2(const uint64_t *v, const uint64_t (&o1)[4], uint64_t &mo, uint64_t &mx) {
  const __m256i one = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(o1));
  for (size_t i = 0; i < 64; i++, v += 4) {
    const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(v)), d = _mm256_xor_si256(x, one);
    const uint64_t z = (uint64_t)_mm256_testz_si256(x, x), is1 = (uint64_t)_mm256_testz_si256(d, d);            // testz: 1 if all bits are zero
    mo |= is1 << i; mx |= ((z | is1) ^ 1) << i; } }
static void classify_block64(const uint64_t *v, const uint64_t (&o1)[4], uint64_t &mo, uint64_t &mx) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) classify_block64_avx2(v, o1, mo, mx); else classify_block64_scalar(v, o1, mo, mx); }
#else
static void classify_block64(const uint64_t *v, const uint64_t (&o1)[4], uint64_t &mo, uint64_t &mx) { classify_block64_scalar(v, o1, mo, mx); }
#endif
// 64 tag bytes of a circuit board -> three bit masks (bit 0: "is one", bit 1: "has a value", bit 2: "that value is a small integer"),
// Prover::set_witness_tagged
static void tags_block64_scalar(const uint8_t *tag, uint64_t &mo, uint64_t &mx, uint64_t &mc) {
  constexpr uint64_t LSB = 0x0101010101010101ull, GATHER = 0x0102040810204080ull;              // (y & LSB) * GATHER >> 56: the low bits of 8 bytes as one byte
  for (size_t k = 0; k < 8; k++) { uint64_t x; memcpy(&x, tag + 8 * k, 8);
    mo |= (((x & LSB) * GATHER) >> 56) << (8 * k);
    mx |= ((((x >> 1) & LSB) * GATHER) >> 56) << (8 * k);
    mc |= ((((x >> 2) & LSB) * GATHER) >> 56) << (8 * k);
  }
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void tags_block64_avx2(const uint8_t *tag, uint64_t &mo, uint64_t &mx, uint64_t &mc) {
  const __m256i lo = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(tag)), hi = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(tag + 32));
  // bit b of every byte -> the byte's sign bit (a 16-bit shift by 7 - b: what spills over from the lower byte lands below the sign bit) -> movemask
  mo |= (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(lo, 7)) | (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(hi, 7)) << 32;
  mx |= (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(lo, 6)) | (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(hi, 6)) << 32;
  mc |= (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(lo, 5)) | (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(hi, 5)) << 32; }
static bool host_has_avx2() { static const bool v = __builtin_cpu_supports("avx2"); return v; }
#else
static bool host_has_avx2() { return false; }
#endif
// host-only self-test of the two block classifiers (tests/test_key_container_cpu.py is the model: pure host logic reachable through the C-ABI): the AVX2 forms
// must give the scalar forms' masks. out[0..2] / out[3..5]: tag masks scalar / fast; out[6..7] / out[8..9]: element masks (is one, has another value) scalar /
// fast
void test_scan_blocks(const uint8_t tags[64], const uint64_t elems[256], const uint64_t one[4], uint64_t out[10]) {
  for (int i = 0; i < 10; i++) out[i] = 0;
  tags_block64_scalar(tags, out[0], out[1], out[2]);
#if defined(__x86_64__)
  if (host_has_avx2()) tags_block64_avx2(tags, out[3], out[4], out[5]); else
#endif
  tags_block64_scalar(tags, out[3], out[4], out[5]);
  uint64_t o1[4] = {one[0], one[1], one[2], one[3]};
  classify_block64_scalar(elems, o1, out[6], out[7]); classify_block64(elems, o1, out[8], out[9]);
}
// The threads that share a hand-over scan: ONE pool for the process, woken by ONE broadcast.  (Round 3 measured six and eight threads no faster than four — on
// assignments that sat in the host's last-level cache.  bench.py cycles through 400 MB of distinct assignments: each scan then streams 7.3 MB from DRAM, a core
// sustains ~10 GB/s of that, and eight threads halve the 0.17 ms; sixteen gain another 5-8 % per proof on the GPU boxes — 256 hardware threads visible,
// profiles/r04v_scan.txt; hosts with fewer than 32 / 12 hardware threads keep eight / four.)
// Until the end of round 4 every prover object kept twelve scan threads of its own and posted a job to each of them: fifteen mutex + futex round trips, 30-39
// us of the calling thread's time before it scanned a single word (tools/handover_trace.py, profiles/r04y_handover_sweep.txt: a third of the hand-over). Here
// the caller publishes the job, bumps an epoch and wakes everybody with one notify_all; the helpers — polling for ZK_SPIN_US after their last scan, asleep on
// the condition variable after that — take chunks until none are left. The caller scans too and does NOT wait for helpers that never woke up in time: it closes
// the round (no new helper may enter) and waits only for those inside. One scan at a time: a prover that finds the pool taken (several proofs in flight) scans
// with its own three submit threads as before.
class ScanPool {
 public:
  static constexpr size_t TMAX = 16;
  // threads that share a scan when the pool is free (the caller included)
  static size_t crew() {
    static const size_t v = [] {
      const char *e = getenv("ZK_SCAN_THREADS");
      const unsigned hw = usable_cpus();
      const size_t t = e ? (size_t)atoi(e) : (hw >= 32 ? 16 : hw >= 12 ? 8 : hw >= 4 ? 4 : 1);
      return t < 1 ? (size_t)1 : t > TMAX ? TMAX : t;
    }();
    return v;
  }
  static ScanPool &get() { static ScanPool pool; return pool; }
  // Runs job() on the caller and on the pool's helpers, returns when nobody is inside job() any more. false: the pool is busy with another prover's scan
  // (nothing ran).
  bool run(const std::function<void()> &job) {
    if (taken_.exchange(true, std::memory_order_acquire)) return false;
    // (released on every way out: start_threads() can throw std::system_error when the process is out of threads, job() may throw)
    struct Release { ScanPool &p; ~Release() { p.job_ = nullptr; p.taken_.store(false, std::memory_order_release); } } release{*this};
    start_threads();
    job_ = &job;
    state_.store(0, std::memory_order_release);                    // open: helpers may enter
    // (no lock around the bump: the calling thread is on its proof's critical path and must not wait for a helper that was preempted while it held the mutex; a helper
    // that misses this wake-up between its check and its sleep misses this round, nothing else — nobody waits for a helper that is not inside)
    epoch_.fetch_add(1, std::memory_order_release);
    cv_.notify_all();
    // (whatever job() does on this thread, nobody may leave while a helper is still inside it: close the round and wait before the exception travels on)
    std::exception_ptr err;
    try { job(); } catch (...) { err = std::current_exception(); }
    uint32_t st = state_.fetch_or(CLOSED, std::memory_order_acq_rel) | CLOSED;   // closed: a helper that wakes up now stays out
    for (int k = 0; st != CLOSED; k++) {
      if ((k & 255) == 255) std::this_thread::yield(); else cpu_relax();
      st = state_.load(std::memory_order_acquire);
    }
    if (err) std::rethrow_exception(err);
    return true;
  }
  // wake the helpers without a job (they find the round closed and poll for the next one): called where a scan is expected soon
  void nudge() {
    if (threads_started_.load(std::memory_order_acquire) == 0 || taken_.load(std::memory_order_acquire)) return;
    // ONE helper is woken here (a notify_all with fifteen sleepers costs the calling thread 15 us, on the critical path of its proof); that helper wakes the
    // others
    // the bump under the mutex when it is free (off the critical path, and then no helper can sit between its check and its sleep and miss this one wake-up —
    // with notify_one nobody else would poll for the next scan); contended, without it as in run()
    std::unique_lock<std::mutex> lk(m_, std::try_to_lock);
    epoch_.fetch_add(1, std::memory_order_release);
    if (lk.owns_lock()) lk.unlock();
    cv_.notify_one();
  }
  ~ScanPool() {
    { std::lock_guard<std::mutex> lk(m_); quit_.store(true); epoch_.fetch_add(1, std::memory_order_release); }
    cv_.notify_all();
    for (auto &t : threads_) t.join();
  }
 private:
  static constexpr uint32_t CLOSED = 1u << 31;
  static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
  }
  static int spin_us() {
    static const int v = [] {
      const char *e = getenv("ZK_SPIN_US");
      if (e) return std::max(0, atoi(e));
      return usable_cpus() >= 4 ? 250 : 0;
    }();
    return v;
  }
  void start_threads() {                                            // (called with taken_ held: one caller at a time)
    if (!threads_.empty() || crew() < 2) return;
    try { for (size_t i = 0; i + 1 < crew(); i++) threads_.emplace_back([this] { loop(); }); }
    catch (const std::system_error &) { if (threads_.empty()) throw; }   // (a smaller crew is a crew)
    threads_started_.store(threads_.size(), std::memory_order_release);
  }
  void loop() {
    uint32_t seen = 0;
    for (;;) {
      // poll for a while (the next proof's scan follows within a millisecond when proofs come back to back), then sleep
      const auto t_end = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us());
      for (int k = 0; spin_us() > 0 && epoch_.load(std::memory_order_acquire) == seen; k++) {
        if ((k & 63) == 63 && std::chrono::steady_clock::now() > t_end) break;
        cpu_relax();
      }
      if (epoch_.load(std::memory_order_acquire) == seen) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return quit_.load() || epoch_.load(std::memory_order_acquire) != seen; });
        lk.unlock();
        cv_.notify_all();                                           // (woken from sleep: pass it on — the caller may have woken only this one, see nudge())
      }
      if (quit_.load()) return;
      seen = epoch_.load(std::memory_order_acquire);
      // enter the round unless it is closed already (the caller finished without us)
      uint32_t st = state_.load(std::memory_order_acquire);
      bool inside = false;
      while (!(st & CLOSED)) {
        if (state_.compare_exchange_weak(st, st + 1, std::memory_order_acq_rel)) { inside = true; break; }
      }
      if (!inside) continue;
      (*job_)();                                                    // (valid: the caller does not leave run() while anybody is inside)
      state_.fetch_sub(1, std::memory_order_acq_rel);
    }
  }
  std::atomic<bool> taken_{false};
  std::atomic<size_t> threads_started_{0};
  std::atomic<uint32_t> epoch_{0}, state_{CLOSED};
  const std::function<void()> *job_ = nullptr;
  std::mutex m_;
  std::condition_variable cv_;
  std::atomic<bool> quit_{false};
  std::vector<std::thread> threads_;
};
// host-only self-test of the pool (tests/test_device_plan_cpu.py, and tests/sanitize_driver.cpp under ASan / UBSan): `callers` threads each run `rounds` rounds
// of a chunk-counting job — through the pool when they get it, alone when it is taken — and every chunk of every round must have been counted exactly once.
// Returns the number of rounds that ran on the pool, -1 on a miscount.
int test_scan_pool(int callers, int rounds) {
  std::atomic<int> pooled{0}, bad{0};
  auto caller = [&] {
    constexpr size_t N = 3000;
    std::vector<std::atomic<uint8_t>> hits(N);
    for (int r = 0; r < rounds; r++) {
      for (auto &h : hits) h.store(0, std::memory_order_relaxed);
      std::atomic<size_t> next{0};
      const std::function<void()> job = [&] {
        for (;;) {
          const size_t i = next.fetch_add(1, std::memory_order_relaxed);
          if (i >= N) break;
          hits[i].fetch_add(1, std::memory_order_relaxed);
        }
      };
      if (ScanPool::get().run(job)) pooled.fetch_add(1); else job();
      for (auto &h : hits) if (h.load(std::memory_order_relaxed) != 1) bad.fetch_add(1);
      if (r % 7 == 3) ScanPool::get().nudge();
    }
  };
  std::vector<std::thread> th;
  for (int c = 1; c < callers; c++) th.emplace_back(caller);
  caller();
  for (auto &t : th) t.join();
  return bad.load() ? -1 : pooled.load();
}
// Calls of this process that are handing over an assignment or proving right now. The scan pool, and the wake-up that precedes the next scan, are for a caller
// that has the prover to itself (one proof after the other: bench.py's loop, a node proving its own transactions one by one); with several proofs in flight the
// helpers would only take the cores from the other callers' witness generators and submit threads (a soak of six genSendproof callers: 1,260 proofs/s with the
// pool used by whoever found it free, 1,390 with three submit threads per caller).
static std::atomic<int> g_calls_busy{0};
struct BusyCall {
  int others;
  BusyCall() : others(g_calls_busy.fetch_add(1, std::memory_order_acq_rel)) {}
  ~BusyCall() { g_calls_busy.fetch_sub(1, std::memory_order_acq_rel); }
  bool alone() const { return others == 0; }
};
void Prover::set_witness(const Fe32 *z, bool montgomery) {
  Impl &p = *impl; LaneScope lane_scope(p.lane); BusyCall busy; double t0 = now_ms(); const size_t n = p.nv + 1, words = (n + 63) / 64;
  Fe32 one; if (montgomery) memcpy(&one, FrParams::R1, 32); else { memset(&one, 0, 32); one.l[0] = 1; }
  // compact form (k_expand_witness): bitmaps of the entries equal to one / to anything else than 0 and 1, offsets, and the "anything else" values only. The
  // scan of the 7 MB assignment is memory-bound on one core (0.3 ms for send), so the prover's submit threads — idle at this point of a proof — and, on hosts
  // with many cores, a dozen scan threads share it chunk by chunk (below); the per-word offsets make the value area look like one list to the kernel.
  uint8_t *pk = reinterpret_cast<uint8_t *>(p.z_host.get());
  uint64_t *ones = (uint64_t *)pk, *other = ones + words;
  uint32_t *off = (uint32_t *)(other + words);
  const size_t vals_at = expand_values_offset(words, 1); Fe32 *vals = (Fe32 *)(pk + vals_at); const size_t max_other = n / 4;
  // zz + 4 i = entry i of [ONE, z_1 .. z_n]; entry 0 is handled apart
  uint64_t o1[4];
  memcpy(o1, &one, 32);
  const uint64_t *zz = reinterpret_cast<const uint64_t *>(z) - 4;
  // The words are handed out in chunks of 32 (2,048 entries = 64 KB of the assignment) from one counter instead of being cut into T equal parts: with equal
  // parts the hand-over takes as long as its SLOWEST thread, and on the two-socket GPU hosts some of the sixteen threads always sit on the other socket from
  // the caller's buffer, share a core or meet another tenant (0.13 to 0.30 ms from process to process for one and the same build,
  // profiles/r04w_host_placement.txt); with chunks a slow thread simply takes fewer. A chunk is classified first (its masks stay on the stack), reserves room
  // for its values with ONE atomic on the shared cursor of the value area, and copies them there: off[w] is an absolute position, so k_expand_witness does not
  // care in which order the chunks arrived, and nothing has to be closed up afterwards.
  // (16 / 8 / 4 words per chunk: 0.10-0.16 / 0.15-0.35 / 0.39-0.56 ms against 0.07-0.15 — the two counters are shared across sockets;
  // profiles/r04y_chunk_sweep.txt)
  constexpr size_t CHUNK_WORDS = 32, CHUNK_MAX = CHUNK_WORDS;
  const size_t n_chunks = (words + CHUNK_WORDS - 1) / CHUNK_WORDS;
  std::atomic<size_t> next_chunk{0}, value_cursor{0};
  std::atomic<bool> overflow{false};
  auto scan = [&](size_t) {
    for (;;) {
      const size_t ch = next_chunk.fetch_add(1, std::memory_order_relaxed);
      if (ch >= n_chunks || overflow.load(std::memory_order_relaxed)) break;
      const size_t w0 = ch * CHUNK_WORDS, w1 = std::min(words, w0 + CHUNK_WORDS);
      uint64_t mo[CHUNK_MAX], mx[CHUNK_MAX];
      size_t cnt = 0;
      for (size_t w = w0; w < w1; w++) {
        uint64_t o = 0, x = 0;
        const size_t lo = 64 * w, hi = lo + 64 < n ? lo + 64 : n;
        // a whole block: 256-bit loads where the host has them
        if (lo && hi - lo == 64) classify_block64(zz + 4 * lo, o1, o, x);
        // branch-free classification of a ragged block
        else for (size_t i = lo ? lo : 1; i < hi; i++) {
          const uint64_t *v = zz + 4 * i;
          const uint64_t nz = (v[0] | v[1] | v[2] | v[3]) != 0, is1 = ((v[0] ^ o1[0]) | (v[1] ^ o1[1]) | (v[2] ^ o1[2]) | (v[3] ^ o1[3])) == 0;
          o |= is1 << (i - lo);
          x |= (nz & (is1 ^ 1)) << (i - lo);
        }
        if (!lo) o |= 1;                                                                                           // the constant ONE
        mo[w - w0] = o;
        mx[w - w0] = x;
        cnt += (size_t)__builtin_popcountll(x);
      }
      size_t at = cnt ? value_cursor.fetch_add(cnt, std::memory_order_relaxed) : 0;
      // too many other values: the call takes the dense path
      if (at + cnt > max_other) {
        overflow.store(true, std::memory_order_relaxed);
        break;
      }
      for (size_t w = w0; w < w1; w++) {
        const size_t lo = 64 * w;
        off[w] = (uint32_t)at;
        // (the chunk's 64 KB are still in this core's cache)
        for (uint64_t m = mx[w - w0]; m; m &= m - 1) memcpy(&vals[at++], zz + 4 * (lo + (size_t)__builtin_ctzll(m)), 32);
        ones[w] = mo[w - w0];
        other[w] = mx[w - w0];
      }
    }
  };
  static const bool threaded = [] { const char *e = getenv("ZK_SUBMIT_THREADS"); return !e || atoi(e) > 0; }();
  auto worker = [&](size_t t) -> SubmitWorker & { return p.scan_worker(t); };
  static const bool trace = getenv("ZK_TRACE_TIMES") != nullptr;
  double t_posted = t0, t_own = t0, t_joined = t0;
  size_t T = 1;
  if (threaded && words >= 512) {
    // the process's scan pool when it is free; with several proofs in flight whoever finds it taken shares the scan with its own three submit threads
    const std::function<void()> job = [&] { scan(0); };
    T = ScanPool::crew();
    if (T < 2 || !busy.alone() || !ScanPool::get().run(job)) {
      T = 4;
      // (the workers are constructed before anything is posted, and this frame — which the posted jobs refer to — is not left before all of them are back)
      SubmitWorker *ws[4] = {nullptr, &worker(1), &worker(2), &worker(3)};
      for (size_t t = 1; t < T; t++) ws[t]->post([&scan, t] { scan(t); });
      if (trace) t_posted = now_ms();
      std::exception_ptr own;
      try { scan(0); } catch (...) { own = std::current_exception(); }
      if (trace) t_own = now_ms();
      for (size_t t = 1; t < T; t++) { try { ws[t]->wait(); } catch (...) { if (!own) own = std::current_exception(); } }
      if (own) std::rethrow_exception(own);
    }
    if (trace) { t_joined = now_ms(); if (t_own == t0) t_own = t_joined; }
  }
  else scan(0);
  // (test switch: the plain-copy branch below, which no BlockMaze assignment reaches on its own)
  static const bool force_dense = getenv("ZK_WITNESS_DENSE") != nullptr;
  const bool compact = !force_dense && !overflow.load();
  // ONE copy carries bitmaps, offsets and values (a few hundred KB); the few values that are not 0 or 1 are brought into Montgomery form by the expanding
  // kernel itself
  if (compact) {
    const size_t total = value_cursor.load();
    // (letting the kernel read the pinned staging area itself, no copy, was measured: no faster)
    Fe32 one_mont;
    memcpy(&one_mont, FrParams::R1, 32);
    upload_async(p.packed.get(), pk, vals_at + 32 * total);
    expand_witness_dev(p.packed.get(), words, one_mont, montgomery ? 0 : 1, n, p.z.get(), p.tags.get(), p.other_vars.get());
    p.tags_valid = true;
    p.n_other = (uint32_t)total; p.n_other_on_device = false;
    if (trace) fprintf(stderr, "trace-handover-host: threads %zu post %.3f own scan %.3f join %.3f copy + expand calls %.3f ms\n", T, t_posted - t0,
        t_own - t_posted, t_joined - t_own,
        now_ms() - t_joined);
  }
  // dense assignment: plain copy
  else {
    Fe32 *h = p.z_host.get();
    h[0] = one;
    memcpy(&h[1], z, 32 * p.nv);
    upload_async(p.z.get(), h, 32 * n);
    if (!montgomery) fr_to_mont_dev(p.z.get(), n);
    p.tags_valid = false;
  }
  p.z_cur = p.z.get(); p.z_set = true;
  last.upload_ms = now_ms() - t0;
}
void Prover::set_witness_tagged(const uint8_t *tag, const Fe32 *wide) {
  Impl &p = *impl;
  LaneScope lane_scope(p.lane);
  BusyCall busy;
  double t0 = now_ms();
  const size_t n = p.nv + 1, words = (n + 63) / 64;
  Fe32 one;
  memcpy(&one, FrParams::R1, 32);
  // the layout set_witness builds, with a third bitmap: the values that are still canonical (the board's small integers)
  uint8_t *pk = reinterpret_cast<uint8_t *>(p.z_host.get());
  uint64_t *ones = (uint64_t *)pk, *other = ones + words, *canon = other + words;
  uint32_t *off = (uint32_t *)(canon + words);
  const size_t vals_at = expand_values_offset(words, 2); Fe32 *vals = (Fe32 *)(pk + vals_at); const size_t max_other = n / 4;
  const bool avx2 = host_has_avx2();
  // like set_witness: the prover's submit threads — idle at this point of a call — share the words chunk by chunk (0.28 -> 0.1 ms for send on the GPU box's
  // host with four threads). A chunk of 32 words is classified first — every value it will need is prefetched on the way: the 7,600 values lie scattered over a
  // 7 MB array of board entries, one cache miss each —, reserves its run of the value area with one atomic, then copies the values, which have had the whole
  // chunk's time to arrive.
  constexpr size_t CHUNK_WORDS = 32;
  const size_t n_chunks = (words + CHUNK_WORDS - 1) / CHUNK_WORDS;
  std::atomic<size_t> next_chunk{0}, value_cursor{0};
  std::atomic<bool> overflow{false};
  auto scan = [&](size_t) {
    for (;;) {
      const size_t ch = next_chunk.fetch_add(1, std::memory_order_relaxed);
      if (ch >= n_chunks || overflow.load(std::memory_order_relaxed)) break;
      const size_t w0 = ch * CHUNK_WORDS, w1 = std::min(words, w0 + CHUNK_WORDS);
      uint64_t mo[CHUNK_WORDS], mx[CHUNK_WORDS], mc[CHUNK_WORDS];
      size_t cnt = 0;
      for (size_t w = w0; w < w1; w++) {
        const size_t lo = 64 * w, hi = lo + 64 < n ? lo + 64 : n;
        uint64_t o = 0, x = 0, c = 0;
        if (hi - lo == 64) {
#if defined(__x86_64__)
          if (avx2) tags_block64_avx2(tag + lo, o, x, c); else
#endif
          tags_block64_scalar(tag + lo, o, x, c);
        }
        else for (size_t i = lo; i < hi; i++) {
          o |= (uint64_t)(tag[i] & 1) << (i - lo);
          x |= (uint64_t)((tag[i] >> 1) & 1) << (i - lo);
          c |= (uint64_t)((tag[i] >> 2) & 1) << (i - lo);
        }
        for (uint64_t m = x; m; m &= m - 1) __builtin_prefetch(&wide[lo + (size_t)__builtin_ctzll(m)]);
        mo[w - w0] = o;
        mx[w - w0] = x;
        mc[w - w0] = c;
        cnt += (size_t)__builtin_popcountll(x);
      }
      size_t at = cnt ? value_cursor.fetch_add(cnt, std::memory_order_relaxed) : 0;
      if (at + cnt > max_other) { overflow.store(true, std::memory_order_relaxed); break; }
      for (size_t w = w0; w < w1; w++) {
        const size_t lo = 64 * w;
        const uint64_t c = mc[w - w0];
        off[w] = (uint32_t)at;
        for (uint64_t m = mx[w - w0]; m; m &= m - 1) {
          const size_t i = lo + (size_t)__builtin_ctzll(m);
          Fe32 &dst = vals[at++];
          // a small integer the board kept as it was (circuit::Board::TAG_SMALL, only its low 64 bits are meaningful): the device converts it
          if ((c >> (i - lo)) & 1) {
            memset(&dst, 0, 32);
            dst.l[0] = wide[i].l[0];
            dst.l[1] = wide[i].l[1];
          }
          else dst = wide[i];
        }
        ones[w] = mo[w - w0];
        other[w] = mx[w - w0];
        canon[w] = c;
      }
    }
  };
  static const bool threaded = [] { const char *e = getenv("ZK_SUBMIT_THREADS"); return !e || atoi(e) > 0; }();
  auto worker = [&](size_t t) -> SubmitWorker & { return p.scan_worker(t); };
  if (threaded && words >= 512) {
    const std::function<void()> job = [&] { scan(0); };
    if (ScanPool::crew() < 2 || !busy.alone() || !ScanPool::get().run(job)) {
      SubmitWorker *ws[4] = {nullptr, &worker(1), &worker(2), &worker(3)};     // (constructed before anything is posted; see set_witness)
      for (size_t t = 1; t < 4; t++) ws[t]->post([&scan, t] { scan(t); });
      std::exception_ptr own;
      try { scan(0); } catch (...) { own = std::current_exception(); }
      for (size_t t = 1; t < 4; t++) { try { ws[t]->wait(); } catch (...) { if (!own) own = std::current_exception(); } }
      if (own) std::rethrow_exception(own);
    }
  }
  else scan(0);
  const bool fits = !overflow.load();
  static const bool force_dense = getenv("ZK_WITNESS_DENSE") != nullptr;
  static const bool trace = getenv("ZK_TRACE_TIMES") != nullptr;
  const double t1 = now_ms();
  if (fits && !force_dense) {
    const size_t n_other = value_cursor.load();
    const double t2 = now_ms(); upload_async(p.packed.get(), pk, vals_at + 32 * n_other); const double t3 = now_ms();
    expand_witness_dev(p.packed.get(), words, one, 2, n, p.z.get(), p.tags.get(), p.other_vars.get()); p.tags_valid = true; p.n_other = (uint32_t)n_other; p.n_other_on_device = false;
    if (trace) fprintf(stderr, "trace-handover: scan %.3f (close-up: none) %.3f copy call %.3f (%zu bytes) expand launch %.3f ms\n", t1 - t0, t2 - t1, t3 - t2,
        vals_at + 32 * n_other, now_ms() - t3);
  }
  // a dense assignment (never a BlockMaze one)
  else {
    Fe32 *h = p.z_host.get();
    Fe32 zero;
    memset(&zero, 0, 32);
    for (size_t i = 0; i < n; i++) {
      if (tag[i] == 6) {
        const HFr v = HFr::from_u64(wide[i].l[0] | (uint64_t)wide[i].l[1] << 32);
        memcpy(&h[i], v.l, 32);
      } else h[i] = tag[i] == 2 ? wide[i] : tag[i] ? one : zero;
    }
    upload_async(p.z.get(), h, 32 * n);
    p.tags_valid = false;
  }
  p.z_cur = p.z.get(); p.z_set = true;
  last.upload_ms = now_ms() - t0;
}
void Prover::set_witness_board(const uint8_t *tag, const Fe32 *wide, const uint8_t *ever_wide, uint32_t marks, const uint8_t *tag_dev, const Fe32 *wide_dev) {
  { static const bool force_dense = getenv("ZK_WITNESS_DENSE") != nullptr; if (force_dense) { set_witness_tagged(tag, wide); return; } }   // (test switch: the plain-copy branch)
  Impl &p = *impl; LaneScope lane_scope(p.lane); BusyCall busy; const double t0 = now_ms(); const size_t n = p.nv + 1;
  if (!p.cand_valid || p.cand_marks != marks) {                 // (first hand-overs of a circuit object only: afterwards the set is complete)
    p.cand.clear(); for (size_t i = 1; i < n; i++) if (ever_wide[i]) p.cand.push_back((uint32_t)i);
    if (p.cand_dev.size() < p.cand.size() + 1) p.cand_dev = DevBuf<uint32_t>(p.cand.size() + p.cand.size() / 8 + 64);
    gpu_sync(); if (!p.cand.empty()) p.cand_dev.upload(p.cand.data(), p.cand.size());
    p.cand_marks = marks; p.cand_valid = true;
  }
  const size_t nc = p.cand.size(), tags_bytes = (n + 31) & ~(size_t)31;
  static const bool trace = getenv("ZK_TRACE_TIMES") != nullptr;
  if (tag_dev && wide_dev) {                                    // the board's memory is mapped: the kernel reads the tag bytes and its candidates' values over PCIe itself
    p.classify_parity ^= 1;
    expand_board_dev(tag_dev, n, p.cand_dev.get(), wide_dev, nc, p.z.get(), p.tags.get(), p.other_vars.get(), p.other_count.get(), p.classify_parity, true);
    p.tags_valid = true; p.n_other = (uint32_t)nc; p.n_other_on_device = true; p.z_cur = p.z.get(); p.z_set = true;
    if (trace) fprintf(stderr, "trace-handover-board: %zu candidates read in place, host %.3f ms\n", nc, now_ms() - t0);
    last.upload_ms = now_ms() - t0; return;
  }
  if (tags_bytes + 32 * nc > 32 * (p.nv + 1 + 8)) { set_witness_tagged(tag, wide); return; }          // (does not fit the staging area: not a BlockMaze board)
  // pinned staging area and its device twin: [tag bytes | candidate values]; one copy
  uint8_t *h = reinterpret_cast<uint8_t *>(p.z_host.get()); memcpy(h, tag, n); Fe32 *vals = reinterpret_cast<Fe32 *>(h + tags_bytes);
  const uint32_t *c = p.cand.data();
  for (size_t j = 0; j < nc; j++) { if (j + 16 < nc) __builtin_prefetch(&wide[c[j + 16]]); vals[j] = wide[c[j]]; }
  upload_async(p.packed.get(), h, tags_bytes + 32 * nc);
  p.classify_parity ^= 1;
  expand_board_dev(p.packed.get(), n, p.cand_dev.get(), reinterpret_cast<const Fe32 *>(p.packed.get() + tags_bytes), nc, p.z.get(), p.tags.get(), p.other_vars.get(),
      p.other_count.get(), p.classify_parity);
  p.tags_valid = true; p.n_other = (uint32_t)nc; p.n_other_on_device = true; p.z_cur = p.z.get(); p.z_set = true;
  if (trace) fprintf(stderr, "trace-handover-board: %zu candidates, %zu bytes, host %.3f ms\n", nc, tags_bytes + 32 * nc, now_ms() - t0);
  last.upload_ms = now_ms() - t0;
}
struct RsTerms { HFr r, s; HG1 r_delta, s_delta, rs_delta_neg; HG2 s_delta2; };
// everything that depends only on (r, s) and the key (:488-495)
static RsTerms rs_terms(const Fe32 *r_in, const Fe32 *s_in, const HG1 &delta_g1, const HG2 &delta_g2) {
  // canonical scalars
  RsTerms t;
  t.r = r_in ? fr_of(*r_in) : random_fr().from_mont();
  t.s = s_in ? fr_of(*s_in) : random_fr().from_mont();
  HFr rs = (t.r.to_mont() * t.s.to_mont()).from_mont();
  t.r_delta = delta_g1.mul(t.r.l); t.s_delta = delta_g1.mul(t.s.l); t.rs_delta_neg = delta_g1.mul(rs.l).neg(); t.s_delta2 = delta_g2.mul(t.s.l); return t; }
static void enqueue_all(Prover::Impl &p) {
  // The witness MSMs (auxiliary streams; order B2, L, A, B1) are released AT ONCE (release point 0 of 0 .. 4 = after all transforms): their fork event sits
  // right behind the hand-over kernels and the submit threads — still polling after their share of the hand-over scan — are woken as soon as the row kernel is
  // launched. Rounds 1-2 released them after the row kernels (their five full-chip classify kernels then fought the row and transform kernels for the CUs);
  // since the witness path is one light sort per pair, starting it beside the gather-bound row kernel is worth 4 % of a host-buffer proof (1.095 -> 1.05 ms
  // median, tools/ab_steps.sh); later release points only move the contention into the transforms and the H accumulation (profiles/r03i_ab_start.txt, and again
  // at the end of round 3: 1.06-1.19 against 1.01 ms).
  // (release point 0 for all four jobs: the measurement switch of rounds 3-4 is gone)
  const std::array<int, 4> start{0, 0, 0, 0};
  // about 80 launches per proof, and the runtime takes several microseconds of host time for each: helper threads submit the four witness MSMs
  // (auxiliary streams) while this one submits the critical chain
  static const bool threaded = [] { const char *e = getenv("ZK_SUBMIT_THREADS"); return !e || atoi(e) > 0; }();
#ifdef ZKGPU_TEST_HOOKS   // diagnostic builds only (make HOOKS=1; tools/inflight_probe.py): 'w' drops the witness MSMs, 'h' the H query, 'n' the transforms — the proofs are then WRONG; shows what share of the machine each part takes
  static const char *dbg_skip = getenv("ZK_DEBUG_SKIP");
  const bool skip_w = dbg_skip && strchr(dbg_skip, 'w'), skip_h = dbg_skip && strchr(dbg_skip, 'h'), skip_n = dbg_skip && strchr(dbg_skip, 'n');
#else
  constexpr bool skip_w = false, skip_h = false, skip_n = false;
#endif
  Prover::Impl *pp = &p;
  // (an assignment that arrived in compact form: the witness MSMs sort from its tags — WitnessTags, k_wsort_tagged; a dense hand-over
  // — ZK_WITNESS_DENSE, or an assignment that is not mostly zeros and ones — keeps the scalar-reading sort)
  const bool tg = p.tags_valid;
  auto wt = [pp, tg](const uint32_t *var_pos, size_t base) {
    WitnessTags t;
    if (tg) {
      t.tags = pp->tags.get();
      t.other_vars = pp->other_vars.get();
      t.n_other = pp->n_other;
      if (pp->n_other_on_device) t.n_other_dev = pp->other_count.get() + (pp->classify_parity & 1);
    }
    t.var_pos = var_pos;
    t.base = (uint32_t)base;
    return t;
  };
  const WitnessTags wtB = wt(p.B_pos->get(), p.b0), wtL = wt(nullptr, (p.c_fold ? 0 : p.ni + 1) + p.l0), wtA = wt(nullptr, p.a0);
  // r1cs_gg_ppzksnark.tcc:442-462,477-484
  auto runB2 = [pp, wtB] {
    pp->B2->run_tagged(pp->z_cur, wtB, pp->B_idx->get() + pp->b0);
  };
  auto runL = [pp, wtL] {
    pp->L->run_tagged(pp->z_cur, wtL, nullptr);
  };
  auto runA = [pp, wtA] {
    pp->A->run_tagged(pp->z_cur, wtA, nullptr);
  };
  auto runB1 = [pp, wtB] {
    pp->B1->run_tagged(pp->z_cur, wtB, pp->B_idx->get() + pp->b0);
  };
  // job order: B2, L, A, B1 (longest first).  A follower of a shared sort is queued behind its leader by the leader's job: its own slot stays empty.
  // the G2 MSM first: its long accumulation then overlaps the transforms, not the H accumulation
  std::function<void()> jobs[4] = { runB2, runL, [pp, runA, runL] { runA(); if (pp->pair_AL) runL(); }, [pp, runB1, runB2] { if (pp->pair_B && pp->b2_first) {
      runB2(); runB1(); } else { runB1(); if (pp->pair_B) runB2(); } } };
  std::function<void()> finish[4] = { [pp] { pp->rB2 = pp->B2->result(); }, [pp] { pp->rL = pp->L->result(); }, [pp] { pp->rA = pp->A->result();
      if (pp->pair_AL) pp->rL = pp->L->result(); }, [pp] { pp->rB1 = pp->B1->result(); if (pp->pair_B) pp->rB2 = pp->B2->result(); } };
  const bool job_used[4] = {!p.pair_B, !p.pair_AL, true, true};
  const int job_stream[4] = {3, 1, 0, 2};                       // the auxiliary stream each MSM was bound to in the constructor (set_stream)
  const bool use_threads = threaded && !p.one_stream;          // (one stream: the calling thread submits everything itself, so that the stream's order is the program's)
  p.settle_all_quietly();                                       // (nothing is pending unless an earlier proof was abandoned by an exception)
  // phase 0: record the fork event (one event; each stream's wait is issued by the thread that feeds it); phase 1: hand the jobs to the submit threads. The
  // main chain's next launch goes in between: waking the threads costs this one ~10 us, which the device would otherwise spend idle behind the row kernel
  auto release = [&](int point, int phase = 2) {
    bool any = false;
    for (int j = 0; j < 4; j++) any |= start[j] == point;
    if (!any) return;
    if (phase != 1) gpu_fork_record();
    if (phase == 0) return;
    for (int j = 0; j < 4; j++) if (start[j] == point && !skip_w && job_used[j]) {
      const int sj = job_stream[j];
      std::function<void()> job = jobs[j], fin = finish[j];
      if (use_threads) {
        if (!p.workers[j]) p.workers[j].reset(new SubmitWorker(p.lane));
        p.workers[j]->post([sj, job, fin] { gpu_fork_wait(sj); job(); fin(); });
        p.pending[j] = true;
      } else {
        gpu_fork_wait(sj);
        job();
        p.inline_result[j] = true;
        if (j == 2 && p.pair_AL) p.inline_result[1] = true;
        if (j == 3 && p.pair_B) p.inline_result[0] = true;
      }
    }
  };
  // (the assignment the proof reads, made equivalent: equal columns folded — on the main stream ahead of the fork, so every witness MSM sees it)
  if (p.n_merge_groups) merge_equal_columns_dev(const_cast<Fe32 *>(p.z_cur), p.tags_valid ? p.tags.get() : nullptr, p.merge_ptr->get(), p.merge_mem->get(), p.n_merge_groups);
  release(0, 0);
  p.cs->eval(p.z_cur, p.abc.get(), p.m, p.tags_valid ? p.tags.get() : nullptr, !p.c_fold); release(0, 1); release(1, 0);
  // r1cs_to_qap_witness_map with d1 = d2 = d3 = 0 (r1cs_to_qap.tcc:239-322); the row kernels test a*b == c on the way
  const int nvec = p.c_fold ? 2 : 3;                          // A, B (and C unless it is folded into the L query)
  // iFFT, then cosetFFT (a step domain runs the passes between the two as one kernel)
  if (!skip_n) p.dom->ifft_then_coset_fft(p.abc.get(), nvec, p.m);
  release(1, 1);
  release(2);
  const bool fuse_pointwise = p.c_fold && p.H->one_pass_sort();   // zinv*a*b is then formed inside the H query's sort kernel
  if (!fuse_pointwise) p.dom->qap_pointwise(p.abc.get(), p.abc.get() + p.m, p.c_fold ? nullptr : p.abc.get() + 2 * p.m); release(3);
  if (!p.h_lagrange) p.dom->icoset_fft(p.abc.get(), 1, p.m);
  release(4);
  if (skip_h) return;
  if (fuse_pointwise) p.H->run_product(p.abc.get() + p.h0, p.abc.get() + p.m + p.h0, p.dom->zinv_dev() + (p.dom->zinv_is_table() ? p.h0 : 0),
      p.dom->zinv_is_table());
  else p.H->run(p.abc.get() + p.h0, nullptr);                                                                             // :466-473
}   // (the witness MSMs' jobs keep running: Impl::settle(j) waits for job j where its result is needed)
// one proof's device work.  (Replaying the five-stream DAG from a captured hipGraph was measured slower than eager submission from the five submit threads on
// ROCm 7.2 / MI355X — 4.65 vs 3.70 ms per proof in round 1 — and was removed.)
static void run_device(Prover::Impl &p) { enqueue_all(p); }
// proof assembly (r1cs_gg_ppzksnark.tcc:487-495)
static void assemble(const Prover::Impl &p, const RsTerms &t, const HG1 &eA, const HG1 &eB1, const HG2 &eB2, const HG1 &eH, const HG1 &eL, Proof &out) {
  HG1 gA = p.alpha_g1.add(eA).add(t.r_delta);                                                                            // :488
  HG1 gB1 = p.beta_g1.add(eB1).add(t.s_delta); HG2 gB2 = p.beta_g2.add(eB2).add(t.s_delta2);                            // :491-492
  HG1 gC = eH.add(eL).add(gA.mul(t.s.l)).add(gB1.mul(t.r.l)).add(t.rs_delta_neg);                                       // :495
  out.A = raw_of(gA); out.B = raw_of(gB2); out.C = raw_of(gC); }
size_t Prover::stash_witness() {
  Impl &p = *impl; LaneScope lane_scope(p.lane); const size_t n = p.nv + 1;
  if (!p.z_set) throw std::runtime_error("stash_witness: no assignment has been handed over to this prover");
  // a slot that was dropped is used again before the list grows
  size_t slot = 0; while (slot < p.stashes.size() && p.stashes[slot]) slot++;
  std::unique_ptr<Impl::Stash> st(new Impl::Stash()); st->z = DevBuf<Fe32>(n);
  copy_dev_async(st->z.get(), p.z.get(), 32 * n); gpu_sync();
  if (slot == p.stashes.size()) p.stashes.push_back(std::move(st)); else p.stashes[slot] = std::move(st);
  return slot;
}
void Prover::drop_stash(size_t slot) {
  Impl &p = *impl; LaneScope lane_scope(p.lane);
  if (slot == (size_t)-1) { gpu_sync(); p.stashes.clear(); p.z_cur = p.z.get(); return; }
  if (slot >= p.stashes.size() || !p.stashes[slot]) throw std::runtime_error("drop_stash: no such slot");
  gpu_sync(); if (p.z_cur == p.stashes[slot]->z.get()) p.z_cur = p.z.get();
  p.stashes[slot].reset();
  while (!p.stashes.empty() && !p.stashes.back()) p.stashes.pop_back();
}
// a kept assignment back on the host, canonical (what set_witness was given — or its equivalent with equal columns folded once a proof has read it in place)
void Prover::read_stash(size_t slot, Fe32 *out) {
  Impl &p = *impl; LaneScope lane_scope(p.lane); const size_t n = p.nv + 1;
  if (slot >= p.stashes.size() || !p.stashes[slot]) throw std::runtime_error("read_stash: no such slot");
  gpu_sync(); DevBuf<Fe32> t(n); copy_dev_async(t.get(), p.stashes[slot]->z.get(), 32 * n); fr_from_mont_dev(t.get(), n); gpu_sync();
  std::vector<Fe32> h(n); t.download(h.data(), n); memcpy(out, h.data() + 1, 32 * p.nv);
}
size_t Prover::equal_column_groups() const { return impl->n_merge_groups; }
size_t Prover::stash_count() const { size_t k = 0; for (const auto &s : impl->stashes) k += s ? 1 : 0; return k; }
bool Prover::prove_stashed(size_t slot, const Fe32 *r_in, const Fe32 *s_in, Proof &out) {
  {
    Impl &p = *impl; LaneScope lane_scope(p.lane); const double t0 = now_ms();
    if (slot >= p.stashes.size() || !p.stashes[slot]) throw std::runtime_error("prove_stashed: no such slot");
    const Impl::Stash &st = *p.stashes[slot]; const size_t n = p.nv + 1;
    // What is resident is the raw vector.  Everything the prover derives from it — the tag byte per variable, the list of the values that are neither 0 nor 1 (what
    // multi_exp_with_mixed_addition's classification does per call, multiexp.tcc:443-496) — is made HERE, inside the call, by one streaming kernel on the prover's
    // main stream ahead of the row kernel; the proof reads the stash in place.  The list's length stays on the device (WitnessTags::n_other_dev): the sort's launch
    // is sized by the bound n.
    p.classify_parity ^= 1;
    classify_witness_dev(st.z.get(), n, p.tags.get(), p.other_vars.get(), p.other_count.get(), p.classify_parity);
    p.z_cur = st.z.get(); p.n_other = (uint32_t)n; p.n_other_on_device = true; p.tags_valid = true; last.upload_ms = now_ms() - t0;
  }
  return prove_resident(r_in, s_in, out);
}
bool Prover::prove_resident(const Fe32 *r_in, const Fe32 *s_in, Proof &out) {
  Impl &p = *impl; LaneScope lane_scope(p.lane); BusyCall busy; double t1 = now_ms(); p.H->set_crowded(!busy.alone()); run_device(p);
  // host work overlapped with the kernels
  RsTerms t = rs_terms(r_in, s_in, p.delta_g1, p.delta_g2);
  double t2 = now_ms();
  // The witness MSMs finish well before the H chain (row products, 7 transforms, the largest MSM).  Their Horner combines, the two scalar multiples and the
  // affine conversions of A and B run on the host meanwhile, in the order the streams complete (each result() waits for its own stream only).
  static const bool trace = getenv("ZK_TRACE_TIMES") != nullptr;
  struct Settle { Impl &p; ~Settle() { p.settle_all_quietly(); } } settle_guard{p};   // no job outlives this call, whatever throws
  // :488 and s*A of :495
  p.settle(2);
  HG1 gA = p.alpha_g1.add(p.rA).add(t.r_delta), c_part = gA.mul(t.s.l).add(t.rs_delta_neg);
  out.A = raw_of(gA);
  double ta = now_ms();
  p.settle(1); c_part = c_part.add(p.rL); double tl = now_ms();
  // :491 and r*B1 of :495
  p.settle(3);
  HG1 gB1 = p.beta_g1.add(p.rB1).add(t.s_delta);
  c_part = c_part.add(gB1.mul(t.r.l));
  double tb1 = now_ms();
  p.settle(0); HG2 gB2 = p.beta_g2.add(p.rB2).add(t.s_delta2); out.B = raw_of(gB2); double tb2 = now_ms();                                              // :492
  gpu_sync(); double t3 = now_ms();
  // the next proof's hand-over is usually microseconds away (proofs come back to back): the scan helpers are woken now — they poll for ZK_SPIN_US before they
  // sleep again — while this thread finishes the proof (without it they are woken by the scan itself, 30-50 us late: profiles/r04y_nudge_ab.txt)
  if (busy.alone() && g_calls_busy.load(std::memory_order_acquire) == 1) ScanPool::get().nudge();
  // (rocprofv3's kernel trace is stamped with CLOCK_BOOTTIME: t1_boot places this proof on its time line)
  if (trace) {
    timespec bt;
    clock_gettime(CLOCK_BOOTTIME, &bt);
    const double boot_ms = bt.tv_sec * 1e3 + bt.tv_nsec * 1e-6;
    fprintf(stderr, "trace: enqueue %.3f A %.3f L %.3f B1 %.3f B2 %.3f sync %.3f upload %.3f t1_boot %.4f\n", t2 - t1, ta - t1, tl - t1, tb1 - t1, tb2 - t1,
        t3 - t1, last.upload_ms, boot_ms - (now_ms() - t1));
  }
  if (!p.cs->check_result()) { last_failed_row = (long)p.cs->failed_row(); return false; }
  out.C = raw_of(p.H->result().add(c_part)); double t4 = now_ms();                                                                                    // :495
  last.qap_ms = t2 - t1; last.msm_ms = t3 - t1; last.finish_ms = t4 - t3; last.total_ms = last.upload_ms + (t4 - t1); return true;
}
static void put_canon_g1(const HG1 &p, uint8_t *o) {
  HFq x, y;
  p.to_affine(x, y);
  x = x.from_mont();
  y = y.from_mont();
  memcpy(o, x.l, 32);
  memcpy(o + 32, y.l, 32);
}
static HG1 get_canon_g1(const uint8_t *o) {
  HFq x, y;
  memcpy(x.l, o, 32);
  memcpy(y.l, o + 32, 32);
  if (x.is_zero() && y.is_zero()) return HG1::inf();
  return HG1::from_affine(x.to_mont(), y.to_mont());
}
bool Prover::prove_partial(uint8_t out[PARTIAL_BYTES]) {
  Impl &p = *impl; LaneScope lane_scope(p.lane); run_device(p); for (int j = 0; j < 4; j++) p.settle(j); gpu_sync(); if (!p.cs->check_result()) { last_failed_row = (long)p.cs->failed_row(); return false; }
  put_canon_g1(p.rA, out); put_canon_g1(p.rB1, out + 64); put_canon_g1(p.H->result(), out + 128); put_canon_g1(p.rL, out + 192);
  HFq2 x, y;
  p.rB2.to_affine(x, y);
  HFq v[4] = {x.c0.from_mont(), x.c1.from_mont(), y.c0.from_mont(), y.c1.from_mont()};
  for (int k = 0; k < 4; k++) memcpy(out + 256 + 32 * k, v[k].l, 32);
  return true;
}
void Prover::finish_from_partials(const uint8_t *records, size_t n, const Fe32 *r_in, const Fe32 *s_in, Proof &out) {
  Impl &p = *impl; HG1 eA = HG1::inf(), eB1 = HG1::inf(), eH = HG1::inf(), eL = HG1::inf(); HG2 eB2 = HG2::inf();
  for (size_t k = 0; k < n; k++) {
    const uint8_t *q = records + k * PARTIAL_BYTES;
    eA = eA.add(get_canon_g1(q));
    eB1 = eB1.add(get_canon_g1(q + 64));
    eH = eH.add(get_canon_g1(q + 128));
    eL = eL.add(get_canon_g1(q + 192));
    HFq v[4];
    bool z = true;
    for (int i = 0; i < 4; i++) {
      memcpy(v[i].l, q + 256 + 32 * i, 32);
      if (!v[i].is_zero()) z = false;
      v[i] = v[i].to_mont();
    }
    if (!z) eB2 = eB2.add(HG2::from_affine(HFq2{v[0], v[1]}, HFq2{v[2], v[3]}));
  }
  assemble(p, rs_terms(r_in, s_in, p.delta_g1, p.delta_g2), eA, eB1, eB2, eH, eL, out);
}

}  // namespace zk
