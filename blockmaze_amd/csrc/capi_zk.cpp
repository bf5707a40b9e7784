// The drop-in layer: the C symbols go-ethereum/zktx binds through cgo (include/zk_{common,mint,send,deposit,redeem}.h).
// Behavioural contract restated from libsnark-vnt/src/{mint,send,deposit,redeem}/*cgo.cpp:
//   * hex strings in, freshly allocated NUL-terminated hex strings out (the Go side never frees them);
//   * a witness that violates the circuit yields the hex of the default proof (G1::one, G2::one, G1::one), whose first
//     characters are "000000..." — the failure sentinel go-ethereum checks (internal/ethapi/api.go:1690);
//   * keys are looked up under /usr/local/prfKey/ (override: ZK_PRFKEY_DIR); unlike the reference they are parsed once
//     and kept resident in HBM, re-read only when the file's size or mtime changes;
//   * no exception, abort or signal handler ever crosses this boundary; calls may arrive concurrently on any thread.
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "../../include/zk_deposit.h"
#include "../../include/zk_mint.h"
#include "../../include/zk_redeem.h"
#include "../../include/zk_send.h"
#include "../../include/zk_batch.h"
#include "../../include/zkgpu.h"
#include "blockmaze_circuits.hpp"
#include "groth16.hpp"

using namespace zk;
extern std::mutex g_gpu_mutex;
extern void zkgpu_set_error(const std::string &s);

namespace {
char *dup_string(const std::string &s) { char *p = (char *)malloc(s.size() + 1); if (p) memcpy(p, s.c_str(), s.size() + 1); return p; }
char *hash_out(const Blob256 &h) { return dup_string(blob_to_hex(h.b, 32)); }
std::string key_dir() { const char *e = getenv("ZK_PRFKEY_DIR"); return e && *e ? e : "/usr/local/prfKey"; }
std::string key_path(CircuitKind k, bool pk) { return key_dir() + "/" + circuit_name(k) + (pk ? "pk.txt" : "vk.txt"); }

struct FileStamp { off_t size = -1; time_t mtime = 0; long mtime_ns = 0; bool operator==(const FileStamp &o) const {
    return size == o.size && mtime == o.mtime && mtime_ns == o.mtime_ns; } };
bool stamp_of(const std::string &p, FileStamp &s) {
  struct stat st;
  if (stat(p.c_str(), &st)) return false;
  s.size = st.st_size;
  s.mtime = st.st_mtim.tv_sec;
  s.mtime_ns = st.st_mtim.tv_nsec;
  return true;
}

// One proving key = a small pool of provers (ZK_PROVERS_PER_KEY, default 6), each with its own circuit board, device buffers and stream set: cgo calls
// that arrive concurrently (tx pool, RPC goroutines, block processing) overlap on the GPU instead of queueing behind one mutex.
// (tag_dev / wide_dev: the device's addresses of the board's two arrays, pinned and mapped once per unit — the hand-over's kernel reads them in place; null: staged)
struct ProverUnit { std::shared_ptr<Prover> prover; std::unique_ptr<Circuit> circuit; std::mutex busy; const uint8_t *tag_dev = nullptr; const Fe32 *wide_dev = nullptr;
  void map_board() { static const bool on = [] { const char *e = getenv("ZK_HANDOVER_MAPPED"); return !e || atoi(e) != 0; }(); if (!on) return;
    circuit::Board &b = circuit->board; tag_dev = (const uint8_t *)gpu_host_register(b.tag.data(), b.tag.size());
    wide_dev = tag_dev ? (const Fe32 *)gpu_host_register(b.wide.data(), b.wide.size() * sizeof(b.wide[0])) : nullptr;
    if (tag_dev && !wide_dev) { gpu_host_unregister(b.tag.data()); tag_dev = nullptr; } }
  ~ProverUnit() { if (circuit && tag_dev) { gpu_host_unregister(circuit->board.tag.data()); gpu_host_unregister(circuit->board.wide.data()); } } };
typedef std::vector<std::shared_ptr<ProverUnit>> UnitList;
// A reload (the key file's size or mtime changed) never touches the old list: it publishes a NEW one, and the old units die when the last proof running on them
// lets go of its reference — a caller can therefore never see a destroyed unit or mutex, however the reload interleaves with proofs in flight.
struct ProverSlot { FileStamp stamp; std::vector<std::shared_ptr<const UnitList>> units /* one list per device slot, built on first use */;
    std::vector<uint8_t> building /* a caller is loading this device's pool */; std::atomic<unsigned> next{0}; };
struct VkSlot { FileStamp stamp; std::shared_ptr<PreparedVerifyingKey> vk; std::shared_ptr<BatchVerifier> gpu; };
std::mutex g_cache_mutex; std::map<std::string, ProverSlot> g_provers; std::map<std::string, VkSlot> g_vks;

std::unique_ptr<Circuit> make_circuit(CircuitKind k, bool emit) {
  switch (k) {
    case CircuitKind::Mint: return make_mint_circuit(emit);
    case CircuitKind::Send: return make_send_circuit(emit);
    case CircuitKind::Redeem: return make_redeem_circuit(emit);
    default: return make_deposit_circuit(emit, 8);
  }
}

// after the first load from text: leave the container behind for the next process start (a read-only key directory simply goes without). `before` is the key
// file's stamp taken BEFORE it was read: the container is written under that stamp, and only if the file still carries it — a key replaced while it was being
// parsed must not leave the old key's tables behind under the new file's size and mtime.
void write_container_quietly(const std::string &pk_path, const ProvingKeyHost &pk, const FileStamp &before) {
  std::string cp = key_container_path(pk_path);
  FileStamp now;
  if (cp.empty() || pk.H_lagrange.empty() || pk.L_star.empty() || !stamp_of(pk_path, now) || !(now == before)) return;
  KeyStamp ks; ks.size = before.size; ks.mtime_s = before.mtime; ks.mtime_ns = before.mtime_ns;
  try { save_key_container(cp, pk, ks); } catch (const std::exception &) {} }
// A unit of the key's pool, locked for the caller (the reference keeps the unit alive, the lock is released first: members are destroyed in reverse order)
struct HeldUnit { std::shared_ptr<ProverUnit> unit; std::unique_lock<std::mutex> lock; };
}  // namespace
// Which device of the list (ZK_DEVICES) a gen*proof call goes to — pure logic, driven by the CPU tests through zkgpu_test_pool_plan. loaded[d]: device d holds
// a pool of this key; building[d]: some caller is loading one there right now; busy[d]: proofs running on it. Policy: the least busy loaded device; but as soon
// as every loaded device already runs `spill` proofs (default 1) and a device without a pool is left, that one is taken (the caller builds its pool: 0.1 s from
// the key's container) — a process that never has two proofs in flight keeps one copy of one key on one GPU, one with many concurrent callers spreads over all
// GPUs of the node before two proofs share a device. A pool that somebody else is building is never waited for while a loaded device exists. Returns the
// device; -1 = wait for a build to finish.
int zk_pool_pick_device(const uint8_t *loaded, const uint8_t *building, const int *busy, int D, int spill, unsigned turn) {
  int best = -1, fresh = -1; bool any_building = false;
  for (int i = 0; i < D; i++) {
    const int d = (int)((turn + (unsigned)i) % (unsigned)D);                            // ties go round by turn
    if (loaded[d]) { if (best < 0 || busy[d] < busy[best]) best = d; }
    else if (building[d]) any_building = true;
  }
  for (int d = 0; d < D && fresh < 0; d++) if (!loaded[d] && !building[d]) fresh = d;   // pools appear in list order
  if (best < 0) return fresh >= 0 && !any_building ? fresh : -1;                         // nothing loaded yet: the first caller builds, the others wait for it
  if (busy[best] >= spill && fresh >= 0) return fresh;
  return best;
}
namespace {
std::condition_variable g_pool_cv;   // signalled under g_cache_mutex whenever a pool build ends
// Loads the key on first use or when the file changed. g_cache_mutex guards the slot table only; a pool is BUILT outside it (under g_gpu_mutex, which
// serialises key loads and the other set-up work of the device), so callers that can be served by a loaded device never queue behind a key load. A prover's
// helper threads start with its first proof (groth16_prover.cpp).
HeldUnit acquire_prover(CircuitKind k) {
  std::string path = key_path(k, true); FileStamp st; if (!stamp_of(path, st)) throw std::runtime_error("proving key not found: " + path);
  const int D = std::max(1, gpu_device_slots());
  static const int spill = [] { const char *e = getenv("ZK_SPILL_BUSY"); int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
  std::shared_ptr<const UnitList> list; unsigned turn = 0;
  { std::unique_lock<std::mutex> lk(g_cache_mutex); ProverSlot &slot = g_provers[path];
    // (a changed key file: new lists; the old provers die with the last proof running on them)
    if ((int)slot.units.size() != D || !(slot.stamp == st)) {
      slot.units.assign(D, nullptr);
      slot.building.assign(D, 0);
      slot.stamp = st;
    }
    turn = slot.next.fetch_add(1);
    for (;;) {
      std::vector<uint8_t> loaded(D, 0); std::vector<int> busy(D, 0);
      for (int d = 0; d < D; d++) if (slot.units[d]) {
        loaded[d] = 1;
        for (auto &u : *slot.units[d]) {
          std::unique_lock<std::mutex> t(u->busy, std::try_to_lock);
          if (!t.owns_lock()) busy[d]++;
        }
      }
      const int dev = zk_pool_pick_device(loaded.data(), slot.building.data(), busy.data(), D, spill, turn);
      if (dev < 0) {
        g_pool_cv.wait(lk);
        if (!(slot.stamp == st)) throw std::runtime_error("proving key changed while it was being loaded: " + path);
        continue;
      }
      if (slot.units[dev]) { list = slot.units[dev]; break; }
      slot.building[dev] = 1; lk.unlock();
      std::shared_ptr<UnitList> fresh; std::exception_ptr err;
      try { std::lock_guard<std::mutex> gl(g_gpu_mutex);
        bool cached = false;
        ProvingKeyHost pk = load_proving_key_fast(path, cached);
        const char *e = getenv("ZK_PROVERS_PER_KEY");
        int n = e ? atoi(e) : 6;
        if (n < 1) n = 1;
        if (n > 7) n = 7;
        fresh = std::make_shared<UnitList>(); std::shared_ptr<Prover> first;
        for (int i = 0; i < n; i++) { auto u = std::make_shared<ProverUnit>();
          // the pool's members share the first one's device tables
          if (i == 0) {
            u->prover.reset(new Prover(pk, 0, 1, dev));
            first = u->prover;
          } else u->prover.reset(new Prover(*first));
          u->circuit = make_circuit(k, false); u->map_board();
          if (u->circuit->board.num_variables() != u->prover->num_variables() ||
              u->circuit->num_inputs() != u->prover->num_inputs()) throw std::runtime_error("proving key does not belong to the " +
              std::string(circuit_name(k)) + " circuit: " + path);
          fresh->push_back(std::move(u)); }
        if (!cached) write_container_quietly(path, pk, st);
      } catch (...) { err = std::current_exception(); fresh.reset(); }
      lk.lock(); ProverSlot &again = g_provers[path];                                    // (std::map: the reference stays valid, looked up again for clarity)
      if (again.stamp == st && (int)again.building.size() == D) { again.building[dev] = 0; if (fresh) again.units[dev] = fresh; }
      g_pool_cv.notify_all();
      if (err) std::rethrow_exception(err);
      list = fresh; break;
    } }
  // first free member; otherwise wait for one (by turn)
  for (const auto &u : *list) { std::unique_lock<std::mutex> lk(u->busy, std::try_to_lock); if (lk.owns_lock()) return HeldUnit{u, std::move(lk)}; }
  const std::shared_ptr<ProverUnit> &u = (*list)[(turn / (unsigned)D) % list->size()]; return HeldUnit{u, std::unique_lock<std::mutex>(u->busy)};
}
std::shared_ptr<PreparedVerifyingKey> vk_for_path(const std::string &path) {
  FileStamp st; if (!stamp_of(path, st)) throw std::runtime_error("verification key not found: " + path);
  std::lock_guard<std::mutex> lk(g_cache_mutex); VkSlot &slot = g_vks[path];
  if (!slot.vk || !(slot.stamp == st)) { slot.vk = prepare_verifying_key(load_verifying_key(path)); slot.gpu.reset(); slot.stamp = st; }
  return slot.vk;
}
std::shared_ptr<PreparedVerifyingKey> vk_for(CircuitKind k) { return vk_for_path(key_path(k, false)); }
// the key's batched GPU verifier (kernel K9), built on first use; the caller holds g_gpu_mutex
std::shared_ptr<BatchVerifier> gpu_verifier_for_path(const std::string &path) {
  std::shared_ptr<PreparedVerifyingKey> vk = vk_for_path(path); std::lock_guard<std::mutex> lk(g_cache_mutex); VkSlot &slot = g_vks[path];
  if (!slot.gpu) slot.gpu = std::shared_ptr<BatchVerifier>(make_batch_verifier(vk->vk).release());
  return slot.gpu;
}
#ifdef ZKGPU_TEST_HOOKS
// test builds only (make TEST_HOOKS=1): ZK_FIXED_RS="<r hex>:<s hex>" makes proofs reproducible. The release library does not contain this code: an environment
// variable must never be able to remove the zero-knowledge property (the reference draws r, s from std::random_device, r1cs_gg_ppzksnark.tcc:418-419).
bool parse_fixed_rs(Fe32 &r, Fe32 &s) {
  const char *e = getenv("ZK_FIXED_RS"); if (!e) return false; const char *colon = strchr(e, ':'); if (!colon) return false;
  auto parse = [](const char *b, const char *en, Fe32 &o) {
    memset(&o, 0, sizeof o);
    int n = 0;
    for (const char *p = en; p-- > b;) {
      char ch = *p;
      int d = ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : -1;
      if (d < 0 || n >= 64) return false;
      o.l[n / 8] |= (uint32_t)d << (4 * (n % 8));
      n++;
    }
    return n > 0;
  };
  return parse(e, colon, r) && parse(colon + 1, e + strlen(e), s);
}
#else
inline bool parse_fixed_rs(Fe32 &, Fe32 &) { return false; }
#endif

// ---- one proof's MSMs cut over several GPUs behind the cgo symbols (kernel K7; round 5) --------------------------------------------------------------
// ZK_SHARD_DEVICES=k: every gen*proof call of the process runs on k shard provers — shard j holds the contiguous slice j of every query of the key on device slot
// j mod (number of device slots, ZK_DEVICES) —, each on a thread of its own: the assignment goes to every shard, each runs the replicated row / transform pipeline
// and its slice of the five MSMs (Prover::prove_partial: five partial sums, 384 bytes, written by the kernels into pinned host memory), the calling thread adds the
// k records and assembles the proof (finish_from_partials: host work only).  No collective, no torch: what a go-ethereum process can use.  It pays where one MSM is
// much longer than the latency floor of its tails (the deposit circuit at depth 32); for the four deployed circuits proof-level spreading (ZK_DEVICES) is faster.
// One proof at a time per key (the shard set is locked for the call).  On a box with one GPU all shards share it: the same code path, testable
// (tests/test_gpu_groth16.py::test_cgo_symbols_with_sharded_msms).
struct ShardSet { FileStamp stamp; std::vector<std::unique_ptr<Prover>> shards; std::unique_ptr<Circuit> circuit; std::mutex busy; };
std::mutex g_shard_mutex; std::map<std::string, std::shared_ptr<ShardSet>> g_shard_sets;
static size_t shard_count() { static const size_t k = [] { const char *e = getenv("ZK_SHARD_DEVICES"); long v = e ? atol(e) : 0; return (size_t)(v < 2 ? 0 : v > 64 ? 64 : v); }(); return k; }
static std::shared_ptr<ShardSet> shard_set_for(CircuitKind k) {
  const std::string path = key_path(k, true); FileStamp st; if (!stamp_of(path, st)) throw std::runtime_error("proving key not found: " + path);
  std::lock_guard<std::mutex> lk(g_shard_mutex); std::shared_ptr<ShardSet> &slot = g_shard_sets[path];
  if (!slot || !(slot->stamp == st)) {
    std::lock_guard<std::mutex> gl(g_gpu_mutex);
    bool cached = false; ProvingKeyHost pk = load_proving_key_fast(path, cached); const size_t K = shard_count(), D = (size_t)std::max(1, gpu_device_slots());
    auto fresh = std::make_shared<ShardSet>(); fresh->stamp = st; fresh->circuit = make_circuit(k, false);
    for (size_t j = 0; j < K; j++) fresh->shards.emplace_back(new Prover(pk, j, K, (int)(j % D)));
    if (fresh->circuit->board.num_variables() != fresh->shards[0]->num_variables() || fresh->circuit->num_inputs() != fresh->shards[0]->num_inputs())
      throw std::runtime_error("proving key does not belong to the " + std::string(circuit_name(k)) + " circuit: " + path);
    if (!cached) write_container_quietly(path, pk, st);
    slot = fresh;
  }
  return slot;
}
// false: the assignment does not satisfy the constraint system
static bool prove_sharded(ShardSet &set, const Fe32 *r, const Fe32 *s, Proof &proof) {
  const size_t K = set.shards.size(); std::vector<uint8_t> rec(K * Prover::PARTIAL_BYTES); std::vector<uint8_t> ok(K, 0); std::vector<std::string> errs(K); std::vector<std::thread> th;
  const uint8_t *tag = set.circuit->board.tag.data(); const Fe32 *wide = reinterpret_cast<const Fe32 *>(set.circuit->board.wide.data());
  auto work = [&](size_t j) {
    try { set.shards[j]->set_witness_tagged(tag, wide); ok[j] = set.shards[j]->prove_partial(rec.data() + j * Prover::PARTIAL_BYTES) ? 1 : 0; }
    catch (const std::exception &e) { errs[j] = e.what(); } catch (...) { errs[j] = "unknown error"; } };
  // (a thread that cannot be started — the process is out of threads — must not leave its started siblings joinable when the vector dies: that would end the host
  // process; they are joined, then the error travels up like any other)
  try { for (size_t j = 1; j < K; j++) th.emplace_back(work, j); }
  catch (...) { for (auto &t : th) t.join(); throw; }
  work(0); for (auto &t : th) t.join();
  for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
  for (size_t j = 0; j < K; j++) if (!ok[j]) return false;
  set.shards[0]->finish_from_partials(rec.data(), K, r, s, proof); return true;   // (r, s null: fresh randomness, drawn inside)
}

static std::atomic<int> g_proofs_in_flight{0};   // genXproof calls of this process that are between acquiring a prover and returning
// shared tail of the gen*proof functions: assign() has filled the circuit's board
template <class AssignFn> char *generate(CircuitKind k, AssignFn assign) {
  try {
    if (!gpu_available()) {
      zkgpu_set_error("no HIP device visible; libzkgpu has no CPU fallback");
      fprintf(stderr, "libzkgpu: no HIP device visible, cannot generate %s proof\n", circuit_name(k));
      return dup_string(proof_to_hex(default_proof()));
    }
    static const bool trace = getenv("ZK_TRACE_TIMES") != nullptr;
    auto now = [] {
      return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    };
    struct InFlight { InFlight() { g_proofs_in_flight.fetch_add(1, std::memory_order_relaxed); } ~InFlight() {
        g_proofs_in_flight.fetch_sub(1, std::memory_order_relaxed); } } in_flight;
    if (shard_count()) {
      std::shared_ptr<ShardSet> set = shard_set_for(k); std::lock_guard<std::mutex> one(set->busy); assign(*set->circuit);
      printf("Trying to generate %s proof...\n", circuit_name(k)); fflush(stdout);
      Fe32 r, s; const bool fixed = parse_fixed_rs(r, s); Proof proof;
      if (!prove_sharded(*set, fixed ? &r : nullptr, fixed ? &s : nullptr, proof)) { printf("can not generate %s proof\n", circuit_name(k)); fflush(stdout); proof = default_proof(); }
      return dup_string(proof_to_hex(proof));
    }
    double t0 = now(); HeldUnit held = acquire_prover(k); ProverUnit &slot = *held.unit; double t1 = now(); assign(*slot.circuit); double t2 = now();
    printf("Trying to generate %s proof...\n", circuit_name(k)); fflush(stdout);
    Fe32 r, s; bool fixed = parse_fixed_rs(r, s); Proof proof;
    // the board's own form (one byte per 0 / 1, Montgomery values for the rest): no conversion, no scan
    { const circuit::Board &bd = slot.circuit->board;
      slot.prover->set_witness_board(bd.tag.data(), reinterpret_cast<const Fe32 *>(bd.wide.data()), bd.ever_wide.data(), bd.wide_marks, slot.tag_dev, slot.wide_dev); }
    double t3 = now();
    if (!slot.prover->prove_resident(fixed ? &r : nullptr, fixed ? &s : nullptr, proof)) {
      fprintf(stderr, "libzkgpu: %s: the statement's assignment violates the constraint system (constraint %ld among others): no proof\n", circuit_name(k), slot.prover->last_failed_row);
      printf("can not generate %s proof\n", circuit_name(k));
      fflush(stdout);
      proof = default_proof();
    }
    double t4 = now(); char *out = dup_string(proof_to_hex(proof));
    if (trace) fprintf(stderr, "trace-abi: acquire %.3f witness %.3f upload %.3f prove %.3f hex %.3f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3, now() - t4);
    return out;
  }
  catch (const std::exception &e) {
    zkgpu_set_error(e.what());
    fprintf(stderr, "libzkgpu: %s\n", e.what());
    return dup_string(proof_to_hex(default_proof()));
  }
  catch (...) { zkgpu_set_error("unknown error"); return dup_string(proof_to_hex(default_proof())); }
}
// The verdicts for m proofs of ONE circuit kind — where every verification of the cgo layer ends up, the single-proof verifyXproof symbols (m = 1) and
// verifyBatch alike. From ZK_VERIFY_GPU_MIN proofs on (default 1: since kernel K9 keeps its values on 29-bit limbs a proof takes 2.1 ms on ONE compute unit of
// the device, 2.5 ms on a host core) the records go to the device in one launch — one workgroup per proof, concurrent callers on separate streams
// (gpu_verify.hip) —, otherwise, or when the process sees no device, to the prepared host verifier. res[j]: 1 accept, 0 reject. A record the device hands back
// (input accumulator at infinity) is decided by the host verifier.
#ifndef ZK_VERIFY_WHILE_PROVING_DEFAULT
#define ZK_VERIFY_WHILE_PROVING_DEFAULT true
#endif
void verify_group(CircuitKind kind, const Proof *ps, const uint8_t *parsed, const Fe32 *inputs, size_t ni, size_t m, uint8_t *res) {
  static const size_t gpu_min = [] { const char *e = getenv("ZK_VERIFY_GPU_MIN"); long v = e ? atol(e) : 1; return (size_t)(v < 1 ? 1 : v); }();
  const std::string path = key_path(kind, false);
  // A single proof also goes to the device while provers of this process are at work (round 6; rounds 3-5 sent it to the host verifier then: K9 took 2.1 ms idle and
  // 2.8 ms beside four provers, the host 1.87).  K9 now takes 0.74 ms and its waves run at priority 3: verifySendproof 0.85 ms on an idle GPU, 0.92 ms (median; p90 1.09)
  // beside one busy prover, 1.01 ms (p90 2.1) beside four, against 1.87-1.90 ms on a host core (profiles/r06_verify_under_load.txt).  ZK_VERIFY_WHILE_PROVING=0: the
  // host verifier while a proof is in flight, as before.
  bool decided = false;
  // (ZK_VERIFY_WHILE_PROVING: 1 = the device also while provers are at work, 0 = the host verifier then; measured again in round 6, profiles/r06_verify_under_load.txt)
  static const bool while_proving = [] { const char *e = getenv("ZK_VERIFY_WHILE_PROVING"); return e ? atoi(e) != 0 : ZK_VERIFY_WHILE_PROVING_DEFAULT; }();
  if (m >= gpu_min && (m >= 2 || while_proving || g_proofs_in_flight.load(std::memory_order_relaxed) == 0) && gpu_available()) {
    // The device may only ever be FASTER than the host verifier, never a different judge: anything that goes wrong on this branch — building the key's
    // verifier, an allocation, a launch or stream error, the test hook below — is logged and the whole group is decided by the prepared host verifier instead.
    // A transient GPU fault must not reject a valid transaction (the reference's verifier is pure host code, r1cs_gg_ppzksnark.tcc:584-590).
    try {
      // (tests: makes this branch throw, so that the fallback below is exercised on a healthy GPU)
      static const bool fail_hook = getenv("ZK_TEST_FAIL_GPU_VERIFY") != nullptr;
      if (fail_hook) throw std::runtime_error("ZK_TEST_FAIL_GPU_VERIFY is set");
      std::shared_ptr<BatchVerifier> v;
      { std::lock_guard<std::mutex> lk(g_gpu_mutex); v = gpu_verifier_for_path(path); }  // (building a key's verifier is serialised; using it is not)
      if (v->num_inputs() == ni) {
        std::vector<uint8_t> dev(m, 0); v->verify(ps, inputs, m, dev.data());            // (into a scratch vector: a throw half-way leaves `res` untouched)
        for (size_t j = 0; j < m; j++) res[j] = dev[j] == 2 ? (parsed[j] && verify_proof(*vk_for_path(path), inputs + j * ni, ni, ps[j])) : dev[j];
      // strong IC: a wrong input count rejects (r1cs_gg_ppzksnark.tcc:584-590)
      } else for (size_t j = 0; j < m; j++) res[j] = 0;
      decided = true;
    } catch (const std::exception &e) {
      static std::atomic<int> noted{0};
      if (noted.fetch_add(1, std::memory_order_relaxed) < 8) fprintf(stderr, "libzkgpu: GPU verifier failed (%s); deciding %zu proof(s) on the host\n",
          e.what(), m);
    }
  }
  if (!decided) {
    std::shared_ptr<PreparedVerifyingKey> vk = vk_for_path(path);
    for (size_t j = 0; j < m; j++) res[j] = parsed[j] && verify_proof(*vk, inputs + j * ni, ni, ps[j]);
  }
  for (size_t j = 0; j < m; j++) res[j] = parsed[j] && res[j] == 1;
}
bool verify(CircuitKind k, const char *data, const std::vector<bool> &public_bits) {
  bool ok = false;
  try {
    Proof p;
    if (data && strnlen(data, 512) == 512 && proof_from_hex(data, p)) {
      std::vector<Fe32> inputs = pack_public_bits(public_bits);
      uint8_t parsed = 1, res = 0;
      verify_group(k, &p, &parsed, inputs.data(), inputs.size(), 1, &res);
      ok = res == 1;
    }
  }
  catch (const std::exception &e) { zkgpu_set_error(e.what()); fprintf(stderr, "libzkgpu: %s\n", e.what()); ok = false; } catch (...) { ok = false; }
  printf("Verifying %s proof %s!!!\n", circuit_name(k), ok ? "successfully" : "unsuccessfully"); fflush(stdout); return ok;
}
void append(std::vector<bool> &v, const std::vector<bool> &w) { v.insert(v.end(), w.begin(), w.end()); }
// the statement of a proof as the verifier packs it (X_gadget::witness_map): args in the order of the kind's verifyXproof symbol
std::vector<bool> public_bits(CircuitKind k, const char *const *a, uint64_t value_s) {
  std::vector<bool> bits; auto h256 = [&](const char *s) { append(bits, blob_bits(blob256_from_hex(s ? s : "").b, 32)); };
  switch (k) {
    // cmtA_old, sn_old, cmtA, value_s (mint/circuit/gadget.tcc:252-269)
    case CircuitKind::Mint: case CircuitKind::Redeem: h256(a[0]);
    h256(a[1]);
    h256(a[2]);
    append(bits, u64_bits(value_s));
    break;
    // cmtA_old, sn_old, cmtS, cmtA_new (send/circuit/gadget.tcc:274-291)
    case CircuitKind::Send: h256(a[0]);
    h256(a[1]);
    h256(a[2]);
    h256(a[3]);
    break;
    // RT, pk, cmtb_old, sn_old, cmtb, sns (deposit/circuit/gadget.tcc:301-323)
    default: h256(a[0]);
    append(bits, blob_bits(blob160_from_hex(a[1] ? a[1] : "").b, 20));
    h256(a[2]);
    h256(a[3]);
    h256(a[4]);
    h256(a[5]);
    break;
  }
  return bits; }
}  // namespace

template <class Fn> static int guarded(Fn fn) {
  try {
    if (!gpu_available()) {
      zkgpu_set_error("no HIP device visible; libzkgpu has no CPU fallback");
      return ZKGPU_ERR_NO_DEVICE;
    }
    std::lock_guard<std::mutex> lk(g_gpu_mutex);
    return fn();
  }
  catch (const std::exception &e) {
    zkgpu_set_error(e.what());
    return ZKGPU_ERR_RUNTIME;
  }
  catch (...) {
    zkgpu_set_error("unknown error");
    return ZKGPU_ERR_RUNTIME;
  }
}
template <class Fn> static int guarded_host(Fn fn) {
  try {
    return fn();
  }
  catch (const std::exception &e) {
    zkgpu_set_error(e.what());
    return ZKGPU_ERR_RUNTIME;
  }
  catch (...) {
    zkgpu_set_error("unknown error");
    return ZKGPU_ERR_RUNTIME;
  }
}

// lanes: the extra prover objects of prove_batch (share p's tables), created on first use // proofs on different prover objects may run concurrently (each has
// its own streams); one object is used by one thread at a time
struct zkgpu_prover { std::shared_ptr<Prover> p; std::mutex m; std::vector<std::shared_ptr<Prover>> lanes; };
template <class Fn> static int guarded_prover(zkgpu_prover *h, Fn fn) {
  try {
    if (!gpu_available()) {
      zkgpu_set_error("no HIP device visible; libzkgpu has no CPU fallback");
      return ZKGPU_ERR_NO_DEVICE;
    }
    if (!h) return ZKGPU_ERR_ARG;
    std::lock_guard<std::mutex> lk(h->m);
    return fn();
  }
  catch (const std::exception &e) {
    zkgpu_set_error(e.what());
    return ZKGPU_ERR_RUNTIME;
  }
  catch (...) {
    zkgpu_set_error("unknown error");
    return ZKGPU_ERR_RUNTIME;
  }
}

extern "C" {
char *zkgpu_abi_genCMT(uint64_t value, char *sn_string, char *r_string) {
  return hash_out(note_cm(value, blob256_from_hex(sn_string), blob256_from_hex(r_string)));
}
char *zkgpu_abi_genCMTS(uint64_t value_s, char *pk_string, char *r_s_string, char *sn_old_string) {
  return hash_out(note_s_cm(value_s, blob160_from_hex(pk_string), blob256_from_hex(r_s_string), blob256_from_hex(sn_old_string)));
}
char *zkgpu_abi_computePRF(char *sk_string, char *r_string) { return hash_out(compute_prf(blob256_from_hex(sk_string), blob256_from_hex(r_string))); }
char *zkgpu_abi_computeCRH(char *pk_string, char *r_string) { return hash_out(compute_crh(blob160_from_hex(pk_string), blob256_from_hex(r_string))); }
// boost::array<uint256, 256> (depositcgo.cpp:304)
static std::vector<Blob256> parse_cmtarray(const char *cmtarray, int n) {
  std::vector<Blob256> leaves;
  std::string s = cmtarray ? cmtarray : "";
  if (n > 256) n = 256;
  for (int i = 0; i < n; i++) leaves.push_back(blob256_from_hex((size_t)i * 66 < s.size() ? s.substr((size_t)i * 66, 66).c_str() : "")); return leaves; }
char *zkgpu_abi_genRoot(char *cmtarray, int n) { return hash_out(merkle_root(parse_cmtarray(cmtarray, n), 8)); }

char *zkgpu_abi_genMintproof(uint64_t value, uint64_t value_old, char *sn_old, char *r_old, char *sn, char *r, char *cmtA_old, char *cmtA, uint64_t value_s,
    char *sk) {
  MintInputs in{value, value_old, value_s, blob256_from_hex(sn_old), blob256_from_hex(r_old), blob256_from_hex(sn), blob256_from_hex(r),
      blob256_from_hex(cmtA_old), blob256_from_hex(cmtA), blob256_from_hex(sk)};
  return generate(CircuitKind::Mint, [&](Circuit &c) { assign_mint(c, in); }); }
// mint_gadget::witness_map (mint/circuit/gadget.tcc:252-269)
bool zkgpu_abi_verifyMintproof(char *data, char *cmtA_old, char *sn_old, char *cmtA, uint64_t value_s) {
  const char *a[3] = {cmtA_old, sn_old, cmtA}; return verify(CircuitKind::Mint, data, public_bits(CircuitKind::Mint, a, value_s)); }
char *zkgpu_abi_genRedeemproof(uint64_t value, uint64_t value_old, char *sn_old, char *r_old, char *sn, char *r, char *cmtA_old, char *cmtA, uint64_t value_s,
    char *sk) {
  RedeemInputs in{value, value_old, value_s, blob256_from_hex(sn_old), blob256_from_hex(r_old), blob256_from_hex(sn), blob256_from_hex(r),
      blob256_from_hex(cmtA_old), blob256_from_hex(cmtA), blob256_from_hex(sk)};
  return generate(CircuitKind::Redeem, [&](Circuit &c) { assign_redeem(c, in); }); }
bool zkgpu_abi_verifyRedeemproof(char *data, char *cmtA_old, char *sn_old, char *cmtA, uint64_t value_s) {
  const char *a[3] = {cmtA_old, sn_old, cmtA}; return verify(CircuitKind::Redeem, data, public_bits(CircuitKind::Redeem, a, value_s)); }

static SendInputs send_inputs(uint64_t value_A, char *r_s, char *sn, char *r, char *cmt_s, char *cmtA, uint64_t value_s, char *pk_recv, uint64_t value_A_new,
    char *sn_A_new, char *r_A_new, char *cmt_A_new, char *sk, char *pk_sender) {
  SendInputs in;   // sendcgo.cpp:317-333: note_old = (value_A, sn, r), notes = (value_s, pk_recv, r_s, sn), note_new = (value_A_new, sn_A_new, r_A_new)
  in.value_old = value_A;
  in.value_s = value_s;
  in.value = value_A_new;
  in.sn_old = blob256_from_hex(sn);
  in.r_old = blob256_from_hex(r);
  in.r_s = blob256_from_hex(r_s);
  in.sn = blob256_from_hex(sn_A_new);
  in.r = blob256_from_hex(r_A_new);
  in.cmtA_old = blob256_from_hex(cmtA);
  in.cmtS = blob256_from_hex(cmt_s);
  in.cmtA = blob256_from_hex(cmt_A_new);
  in.sk = blob256_from_hex(sk);
  in.pk_recv = blob160_from_hex(pk_recv);
  in.pk_sender = blob160_from_hex(pk_sender);
  return in;
}
char *zkgpu_abi_genSendproof(uint64_t value_A, char *r_s, char *sn, char *r, char *cmt_s, char *cmtA, uint64_t value_s, char *pk_recv, uint64_t value_A_new,
    char *sn_A_new, char *r_A_new, char *cmt_A_new, char *sk, char *pk_sender) {
  SendInputs in = send_inputs(value_A, r_s, sn, r, cmt_s, cmtA, value_s, pk_recv, value_A_new, sn_A_new, r_A_new, cmt_A_new, sk, pk_sender);
  return generate(CircuitKind::Send, [&](Circuit &c) { assign_send(c, in); }); }
// send_gadget::witness_map (send/circuit/gadget.tcc:274-291)
bool zkgpu_abi_verifySendproof(char *data, char *cmtA_old, char *sn_old, char *cmtS, char *cmtA_new) {
  const char *a[4] = {cmtA_old, sn_old, cmtS, cmtA_new}; return verify(CircuitKind::Send, data, public_bits(CircuitKind::Send, a, 0)); }

// depositcgo.cpp:327-444: the Merkle path of cmtS is rebuilt from cmtarray (the tree holds the leaves up to and including the first occurrence of cmtS plus
// everything appended afterwards, i.e. all n leaves); RT is ignored and the root recomputed
static DepositInputs deposit_inputs(uint64_t value, uint64_t value_old, char *sn_old, char *r_old, char *sn, char *r, char *sns, char *rs, char *cmtB_old,
    char *cmtB, uint64_t value_s, char *pk, char *sn_A_old, char *cmtS, char *cmtarray, int n, char *sk, size_t depth) {
  DepositInputs in;
  in.value = value;
  in.value_old = value_old;
  in.value_s = value_s;
  in.sn_old = blob256_from_hex(sn_old);
  in.r_old = blob256_from_hex(r_old);
  in.sn = blob256_from_hex(sn);
  in.r = blob256_from_hex(r);
  in.sn_s = blob256_from_hex(sns);
  in.r_s = blob256_from_hex(rs);
  in.cmtB_old = blob256_from_hex(cmtB_old);
  in.cmtB = blob256_from_hex(cmtB);
  in.cmtS = blob256_from_hex(cmtS);
  in.sk = blob256_from_hex(sk);
  in.pk_recv = blob160_from_hex(pk);
  in.sn_A_old = blob256_from_hex(sn_A_old);
  std::vector<Blob256> leaves = parse_cmtarray(cmtarray, n);
  size_t index = 0;
  bool found = false;
  for (size_t i = 0; i < leaves.size(); i++) if (!memcmp(leaves[i].b, in.cmtS.b, 32)) {
    index = i;
    found = true;
    break;
  }
  // the reference throws out of IncrementalMerkleTree::path() here (IncrementalMerkleTree.tcc:214-216), taking the Go process with it
  if (!found) throw std::runtime_error("cmtS is not among the commitments of cmtarray");
  in.path = merkle_path(leaves, depth, index, in.index_bits); in.rt = merkle_root(leaves, depth); return in; }
char *zkgpu_abi_genDepositproof(uint64_t value, uint64_t value_old, char *sn_old, char *r_old, char *sn, char *r, char *sns, char *rs, char *cmtB_old,
    char *cmtB, uint64_t value_s, char *pk, char *sn_A_old, char *cmtS, char *cmtarray, int n, char *RT, char *sk) {
  (void)RT;
  return generate(CircuitKind::Deposit, [&](Circuit &c) { assign_deposit(c, deposit_inputs(value, value_old, sn_old, r_old, sn, r, sns, rs, cmtB_old, cmtB,
      value_s, pk, sn_A_old, cmtS, cmtarray, n, sk, 8)); });
}
// deposit_gadget::witness_map (deposit/circuit/gadget.tcc:301-323)
bool zkgpu_abi_verifyDepositproof(char *data, char *RT, char *pk, char *cmtb_old, char *snold, char *cmtb, char *sns) {
  const char *a[6] = {RT, pk, cmtb_old, snold, cmtb, sns}; return verify(CircuitKind::Deposit, data, public_bits(CircuitKind::Deposit, a, 0)); }

// ---- engine-level entry points for keys, circuits and the resident prover (include/zkgpu.h) ---------------------------
static void write_r1cs_file(const char *path, const R1csHost &cs) {
  FILE *f = fopen(path, "wb");
  if (!f) throw std::runtime_error(std::string("cannot write ") + path);
  uint64_t hdr[3] = {cs.n_inputs, cs.n_vars, cs.n_cons}; fwrite("R1CSBM01", 1, 8, f); fwrite(hdr, 8, 3, f);
  for (int m = 0; m < 3; m++) {
    uint64_t nnz = cs.col[m].size();
    fwrite(&nnz, 8, 1, f);
    fwrite(cs.rowptr[m].data(), 4, cs.rowptr[m].size(), f);
    fwrite(cs.col[m].data(), 4, nnz, f);
    fwrite(cs.coeff[m].data(), 32, nnz, f);
  }
  fclose(f);
}
static R1csHost read_r1cs_file(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + path);
  char mg[8];
  uint64_t hdr[3];
  R1csHost cs;
  if (fread(mg, 1, 8, f) != 8 || memcmp(mg, "R1CSBM01", 8) || fread(hdr, 8, 3, f) != 3) {
    fclose(f);
    throw std::runtime_error("bad R1CS file");
  }
  cs.n_inputs = hdr[0];
  cs.n_vars = hdr[1];
  cs.n_cons = hdr[2];
  for (int m = 0; m < 3; m++) {
    uint64_t nnz;
    if (fread(&nnz, 8, 1, f) != 1) {
      fclose(f);
      throw std::runtime_error("bad R1CS file");
    }
    cs.rowptr[m].resize(cs.n_cons + 1);
    cs.col[m].resize(nnz);
    cs.coeff[m].resize(nnz);
    if (fread(cs.rowptr[m].data(), 4, cs.n_cons + 1, f) != cs.n_cons + 1 || fread(cs.col[m].data(), 4, nnz, f) != nnz || fread(cs.coeff[m].data(), 32, nnz,
        f) != nnz) {
      fclose(f);
      throw std::runtime_error("truncated R1CS file");
    }
  }
  fclose(f);
  return cs;
}
static void write_witness_file(const char *path, const std::vector<Fe32> &z) {
  FILE *f = fopen(path, "wb");
  if (!f) throw std::runtime_error(std::string("cannot write ") + path);
  uint64_t n = z.size();
  fwrite(&n, 8, 1, f);
  fwrite(z.data(), 32, n, f);
  fclose(f);
}

int zkgpu_circuit_export(int kind, int tree_depth, const char *r1cs_path) {
  return guarded_host([&] { std::unique_ptr<Circuit> c = kind == 100 ? make_sha256_two_to_one(true) : kind == 101 ? make_merkle_test_circuit(true,
      tree_depth) : kind == 102 ? make_lesscmp_test_circuit(true) : kind == 103 ? make_cmta_test_circuit(true) : kind >= 104 && kind <= 106 ?
      make_hashblock_test_circuit(true, kind - 104) : kind == 107 ? make_unpacker_test_circuit(true, (size_t)tree_depth) : kind == (int)CircuitKind::Deposit ? make_deposit_circuit(true,
      tree_depth) : make_circuit((CircuitKind)kind, true); write_r1cs_file(r1cs_path, c->r1cs()); return ZKGPU_OK; });
}
/* bits: 64 + 256 + 256 bytes, each 0 or 1, in the circuit's bit order */
int zkgpu_witness_cmta(const uint8_t *bits, const char *wit_path) {
  return guarded_host([&] { auto c = make_cmta_test_circuit(false); std::vector<bool> v(bits, bits + 64), sn(bits + 64, bits + 320), r(bits + 320, bits + 576);
      assign_cmta_test(*c, v, sn, r); std::vector<Fe32> z; c->export_assignment(z); write_witness_file(wit_path, z); return ZKGPU_OK; });
}
/* which: 0 CMTS (736 input bits), 1 PRF (512), 2 CRH (416); bits: one byte (0 / 1) per input bit, in the block's message order */
int zkgpu_witness_hashblock(int which, const uint8_t *bits, const char *wit_path) { return guarded_host([&] {
    if (which < 0 || which > 2) throw std::runtime_error("hashblock: which must be 0, 1 or 2"); auto c = make_hashblock_test_circuit(false, which);
  assign_hashblock_test(*c, std::vector<bool>(bits, bits + hashblock_input_bits(which))); std::vector<Fe32> z; c->export_assignment(z);
      write_witness_file(wit_path, z); return ZKGPU_OK; }); }
/* bits: nbits bytes, each 0 or 1 */
int zkgpu_witness_unpacker(int nbits, const uint8_t *bits, const char *wit_path) {
  return guarded_host([&] { auto c = make_unpacker_test_circuit(false, (size_t)nbits); assign_unpacker_test(*c, std::vector<bool>(bits, bits + nbits)); std::vector<Fe32> z;
      c->export_assignment(z); write_witness_file(wit_path, z); return ZKGPU_OK; });
}
int zkgpu_witness_lesscmp(uint64_t value_old, uint64_t value_s, const char *wit_path) {
  return guarded_host([&] { auto c = make_lesscmp_test_circuit(false); assign_lesscmp_test(*c, value_old, value_s); std::vector<Fe32> z;
      c->export_assignment(z); write_witness_file(wit_path, z); return ZKGPU_OK; });
}
int zkgpu_witness_sha256(const uint8_t left[32], const uint8_t right[32], const char *wit_path) {
  return guarded_host([&] { auto c = make_sha256_two_to_one(false); assign_sha256_two_to_one(*c, blob_bits(left, 32), blob_bits(right, 32));
      std::vector<Fe32> z; c->export_assignment(z); write_witness_file(wit_path, z); return ZKGPU_OK; });
}
/* Merkle test circuit: leaf and depth siblings (leaf level first, 32 bytes each, in hashing byte order), position of the leaf; the root is computed */
int zkgpu_witness_merkle(int depth, const uint8_t leaf[32], const uint8_t *siblings, uint64_t position, const char *wit_path) { return guarded_host([&] {
    auto c = make_merkle_test_circuit(false, depth);
  Blob256 lf; memcpy(lf.b, leaf, 32); std::vector<Blob256> path(depth); std::vector<bool> idx(depth); Blob256 cur = lf;
  for (int d = 0; d < depth; d++) {
    memcpy(path[d].b, siblings + 32 * d, 32);
    idx[d] = (position >> d) & 1;
    Blob256 nx;
    if (idx[d]) sha256_compress_raw(path[d].b, cur.b, nx.b);
    else sha256_compress_raw(cur.b, path[d].b, nx.b);
    cur = nx;
  }
  assign_merkle_test(*c, lf, path, idx, cur); std::vector<Fe32> z; c->export_assignment(z); write_witness_file(wit_path, z); return ZKGPU_OK; }); }
int zkgpu_witness_deposit(uint64_t value, uint64_t value_old, char *sn_old, char *r_old, char *sn, char *r, char *sns, char *rs, char *cmtB_old, char *cmtB,
    uint64_t value_s, char *pk, char *sn_A_old, char *cmtS, char *cmtarray, int n, char *sk, int tree_depth, const char *wit_path) {
  return guarded_host([&] { DepositInputs in = deposit_inputs(value, value_old, sn_old, r_old, sn, r, sns, rs, cmtB_old, cmtB, value_s, pk, sn_A_old, cmtS,
      cmtarray, n, sk, (size_t)tree_depth);
    auto c = make_deposit_circuit(false, (size_t)tree_depth); assign_deposit(*c, in); std::vector<Fe32> z; c->export_assignment(z);
        write_witness_file(wit_path, z); return ZKGPU_OK; }); }
int zkgpu_witness_send(uint64_t value_A, char *r_s, char *sn, char *r, char *cmt_s, char *cmtA, uint64_t value_s, char *pk_recv, uint64_t value_A_new,
    char *sn_A_new, char *r_A_new, char *cmt_A_new, char *sk, char *pk_sender, const char *wit_path) {
  return guarded_host([&] { auto c = make_send_circuit(false); assign_send(*c, send_inputs(value_A, r_s, sn, r, cmt_s, cmtA, value_s, pk_recv, value_A_new,
      sn_A_new, r_A_new, cmt_A_new, sk, pk_sender)); std::vector<Fe32> z; c->export_assignment(z); write_witness_file(wit_path, z); return ZKGPU_OK; });
}
int zkgpu_witness_mint_redeem(int redeem, uint64_t value, uint64_t value_old, char *sn_old, char *r_old, char *sn, char *r, char *cmtA_old, char *cmtA,
    uint64_t value_s, char *sk, const char *wit_path) {
  return guarded_host([&] { MintInputs in{value, value_old, value_s, blob256_from_hex(sn_old), blob256_from_hex(r_old), blob256_from_hex(sn),
      blob256_from_hex(r), blob256_from_hex(cmtA_old), blob256_from_hex(cmtA), blob256_from_hex(sk)};
    auto c = redeem ? make_redeem_circuit(false) : make_mint_circuit(false);
    if (redeem) {
      RedeemInputs ri{in.value, in.value_old, in.value_s, in.sn_old, in.r_old, in.sn, in.r, in.cmtA_old, in.cmtA, in.sk};
      assign_redeem(*c, ri);
    } else assign_mint(*c, in);
    std::vector<Fe32> z; c->export_assignment(z); write_witness_file(wit_path, z); return ZKGPU_OK; }); }

int zkgpu_debug_time_send_witness(double out[3]) { return guarded_host([&] { auto now = [] {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  SendInputs in{};
  in.value_old = 22;
  in.value_s = 8;
  in.value = 14;
  double t0 = now();
  auto c = make_send_circuit(false);
  double t1 = now();
  assign_send(*c, in);
  double t2 = now();
  std::vector<Fe32> z;
  c->export_assignment(z);
  double t3 = now();
  assign_send(*c, in); double t4 = now(); out[0] = t1 - t0; out[1] = t4 - t3; out[2] = t3 - t2; return ZKGPU_OK; }); }
/* host only: the groups of variables with identical columns of an R1CS file, flattened as [size, members ...] per group; returns the number of words written (or needed) */
int zkgpu_test_equal_columns(const char *r1cs_path, uint32_t *out, size_t cap) { int res = 0; int rc = guarded_host([&] {
    if (!r1cs_path) return ZKGPU_ERR_ARG; const R1csHost cs = read_r1cs_file(r1cs_path); size_t at = 0;
    for (const auto &g : equal_column_groups(cs)) { if (out && at < cap) out[at] = (uint32_t)g.size(); at++; for (uint32_t v : g) { if (out && at < cap) out[at] = v; at++; } }
    res = (int)at; return ZKGPU_OK; }); return rc == ZKGPU_OK ? res : rc; }
int zkgpu_keygen_from_r1cs(const char *r1cs_path, uint64_t seed, const char *pk_path, const char *vk_path) { return guarded([&] {
    R1csHost cs = read_r1cs_file(r1cs_path); ProvingKeyHost pk; VerifyingKeyHost vk;
  generate_keys(cs, seed ? ToxicWaste::from_seed(seed) : ToxicWaste::random(), pk, vk); save_proving_key(pk_path, pk); save_verifying_key(vk_path, vk);
      return ZKGPU_OK; }); }
int zkgpu_keygen(int kind, int tree_depth, uint64_t seed, const char *pk_path, const char *vk_path) { return guarded([&] {
    std::unique_ptr<Circuit> c = kind == (int)CircuitKind::Deposit ? make_deposit_circuit(true, tree_depth) : make_circuit((CircuitKind)kind, true);
  ProvingKeyHost pk; VerifyingKeyHost vk; generate_keys(c->r1cs(), seed ? ToxicWaste::from_seed(seed) : ToxicWaste::random(), pk, vk);
      save_proving_key(pk_path, pk); save_verifying_key(vk_path, vk); return ZKGPU_OK; }); }

zkgpu_prover *zkgpu_prover_load_shard(const char *pk_path, size_t shard_rank, size_t shard_world) { zkgpu_prover *h = nullptr; guarded([&] { FileStamp before;
    if (!stamp_of(pk_path, before)) throw std::runtime_error(std::string("proving key not found: ") + pk_path); bool cached = false;
    ProvingKeyHost pk = load_proving_key_fast(pk_path, cached); std::unique_ptr<zkgpu_prover> p(new zkgpu_prover);
    p->p.reset(new Prover(pk, shard_rank, shard_world));
  if (!cached) write_container_quietly(pk_path, pk, before); h = p.release(); return ZKGPU_OK; }); return h; }
/* pure host logic of the multi-device pool, for the CPU tests: parses `spec` as ZK_DEVICES would be (n_visible devices, `fallback` = ZK_DEVICE / LOCAL_RANK) into out_devices (returns
 * the count), and writes into out_order the device that each of n_order callers arriving AT THE SAME TIME (nobody has finished yet) is sent to by acquire_prover's policy */
int zkgpu_test_device_plan(const char *spec, int n_visible, int fallback, int per_device, int *out_devices, int *out_order, int n_order) {
  std::vector<int> l = parse_device_list(spec, n_visible, fallback); for (size_t i = 0; i < l.size(); i++) out_devices[i] = l[i];
  const int D = std::max<int>(1, (int)l.size()); std::vector<int> rel(n_order, -1); (void)per_device;
  zkgpu_test_pool_plan(D, 1, rel.data(), n_order, out_order); return (int)l.size(); }
/* acquire_prover's device choice replayed on the host: call i first lets go of the proof that call release_before[i] started (-1: nobody finishes), then picks its device
 * (a pool that has to be built counts as loaded from then on).  out_dev[i] = the device slot.  Returns the number of pools built. */
int zkgpu_test_pool_plan(int D, int spill, const int *release_before, int n_calls, int *out_dev) {
  if (D < 1 || D > 64 || n_calls < 0) return -1;
  std::vector<uint8_t> loaded(D, 0), building(D, 0); std::vector<int> busy(D, 0); int built = 0;
  for (int i = 0; i < n_calls; i++) {
    const int r = release_before ? release_before[i] : -1; if (r >= 0 && r < i && out_dev[r] >= 0 && busy[out_dev[r]] > 0) busy[out_dev[r]]--;
    const int d = zk_pool_pick_device(loaded.data(), building.data(), busy.data(), D, spill < 1 ? 1 : spill, (unsigned)i); out_dev[i] = d; if (d < 0) return -1;
    if (!loaded[d]) { loaded[d] = 1; built++; } busy[d]++; }
  return built; }
/* the stream-lane planner of gpu.hip replayed on the host: n_slots devices whose pools are built one device at a time, `kinds` circuit kinds x per_kind members each taking
 * a lane.  Returns -1 if any member is left without a lane, else the largest number of provers sharing one lane; out_lanes_per_slot[d] = lanes bound to device slot d. */
int zkgpu_test_lane_plan(int n_slots, int kinds, int per_kind, int *out_lanes_per_slot) {
  return lane_plan_simulate(n_slots, kinds, per_kind, out_lanes_per_slot); }
/* the hand-over's block classifiers, scalar against the forms the host's CPU selects (AVX2 where it has it): see groth16_prover.cpp: test_scan_blocks */
int zkgpu_test_scan_blocks(const uint8_t *tags64, const uint64_t *elems64x4, const uint64_t *one4, uint64_t *out10) {
  return guarded_host([&] { test_scan_blocks(tags64, elems64x4, one4, out10); return ZKGPU_OK; });
}
/* the hand-over's scan pool (groth16_prover.cpp: ScanPool) driven from `callers` threads at once, host only: rounds that ran on the pool (>= 0), -1 if a chunk was counted twice or not at all */
int zkgpu_test_cgroup_quota(const char *root) { int out = -1; guarded_host([&] { out = test_cgroup_quota(root); return ZKGPU_OK; }); return out; }
int zkgpu_test_scan_pool(int callers, int rounds) { int out = -1; guarded_host([&] { out = test_scan_pool(callers, rounds); return ZKGPU_OK; }); return out; }
/* host-only self-test of the container code (tests/test_key_container_cpu.py): a synthetic transformed key of the given shape is written, mapped back and compared; then the
 * file is truncated, a payload byte is flipped, and the source stamp is changed — each must make the loader refuse.  Returns 0 if every step behaved. */
int zkgpu_test_key_container(const char *path, size_t n_vars, size_t n_cons, size_t m) { int rc = -1; guarded_host([&] {
  ProvingKeyHost pk;
  uint64_t s = 0x1234;
  auto rnd = [&](void *p, size_t n) {
    uint8_t *b = (uint8_t *)p;
    for (size_t i = 0; i < n; i++) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      b[i] = (uint8_t)(s >> 56);
    }
  };
  pk.cs.n_inputs = 3;
  pk.cs.n_vars = n_vars;
  pk.cs.n_cons = n_cons;
  pk.A.resize(n_vars + 1);
  pk.L_star.resize(n_vars + 1);
  pk.H_lagrange.resize(m);
  size_t nB = n_vars / 2 + 1;
  pk.B_idx.resize(nB);
  pk.B_g1.resize(nB);
  pk.B_g2.resize(nB);
  rnd(&pk.alpha_g1, 64);
  rnd(&pk.beta_g1, 64);
  rnd(&pk.delta_g1, 64);
  rnd(&pk.beta_g2, 128);
  rnd(&pk.delta_g2, 128);
  rnd(pk.A.data(), pk.A.size() * 64);
  rnd(pk.L_star.data(), pk.L_star.size() * 64);
  rnd(pk.H_lagrange.data(), m * 64);
  rnd(pk.B_g1.data(), nB * 64);
  rnd(pk.B_g2.data(), nB * 128);
  for (size_t i = 0; i < nB; i++) pk.B_idx[i] = (uint32_t)(2 * i);
  for (int k = 0; k < 3; k++) {
    pk.cs.rowptr[k].resize(n_cons + 1);
    pk.cs.rowptr[k][0] = 0;
    for (size_t i = 0; i < n_cons; i++) pk.cs.rowptr[k][i + 1] = pk.cs.rowptr[k][i] + (uint32_t)((i + k) % 3);
    size_t nnz = pk.cs.rowptr[k][n_cons];
    pk.cs.col[k].resize(nnz);
    pk.cs.coeff[k].resize(nnz);
    for (size_t e = 0; e < nnz; e++) pk.cs.col[k][e] = (uint32_t)(e % (n_vars + 1));
    rnd(pk.cs.coeff[k].data(), nnz * 32);
  }
  KeyStamp st{12345, 1700000000, 42}; save_key_container(path, pk, st); ProvingKeyHost q;
  auto same = [&](const ProvingKeyHost &a, const ProvingKeyHost &b) {
    bool ok = !memcmp(&a.alpha_g1, &b.alpha_g1, 64) && !memcmp(&a.delta_g2, &b.delta_g2,
        128) && a.B_idx == b.B_idx && a.cs.n_cons == b.cs.n_cons && a.cs.n_vars == b.cs.n_vars && a.cs.n_inputs == b.cs.n_inputs;
    ok = ok && a.A.size() == b.A.size() && !memcmp(a.A.data(), b.A.data(), a.A.size() * 64) && a.L_star.size() == b.L_star.size() && !memcmp(a.L_star.data(),
        b.L_star.data(), a.L_star.size() * 64) && a.H_lagrange.size() == b.H_lagrange.size() && !memcmp(a.H_lagrange.data(), b.H_lagrange.data(),
        a.H_lagrange.size() * 64);
    ok = ok && a.B_g2.size() == b.B_g2.size() && !memcmp(a.B_g2.data(), b.B_g2.data(), a.B_g2.size() * 128) && !memcmp(a.B_g1.data(), b.B_g1.data(),
        a.B_g1.size() * 64);
    for (int k = 0; k < 3 && ok; k++) ok = a.cs.rowptr[k] == b.cs.rowptr[k] && a.cs.col[k] == b.cs.col[k] && a.cs.coeff[k].size() == b.cs.coeff[k].size() &&
        !memcmp(a.cs.coeff[k].data(), b.cs.coeff[k].data(), a.cs.coeff[k].size() * 32);
    return ok;
  };
  if (!load_key_container(path, st, q) || !same(pk, q)) { rc = 1; return ZKGPU_OK; }
  // the key file changed: stale
  KeyStamp other = st;
  other.mtime_ns++;
  if (load_key_container(path, other, q)) {
    rc = 2;
    return ZKGPU_OK;
  }
  struct stat sb; if (stat(path, &sb)) { rc = 3; return ZKGPU_OK; }
  // bit rot: checksum
  {
    FILE *f = fopen(path, "r+b");
    fseek(f, (long)(sb.st_size / 2), SEEK_SET);
    int ch = fgetc(f);
    fseek(f, (long)(sb.st_size / 2), SEEK_SET);
    fputc(ch ^ 1, f);
    fclose(f);
    if (load_key_container(path, st, q)) {
      rc = 4;
      return ZKGPU_OK;
    }
    f = fopen(path, "r+b");
    fseek(f, (long)(sb.st_size / 2), SEEK_SET);
    fputc(ch, f);
    fclose(f);
    if (!load_key_container(path, st, q)) {
      rc = 5;
      return ZKGPU_OK;
    }
  }
  // truncated
  if (truncate(path, sb.st_size - 64)) {
    rc = 6;
    return ZKGPU_OK;
  }
  if (load_key_container(path, st, q)) {
    rc = 7;
    return ZKGPU_OK;
  }
  rc = 0; return ZKGPU_OK; }); return rc; }
/* 1 if a valid container exists for this key file (what the next load will use), 0 if not */
int zkgpu_key_container_valid(const char *pk_path) {
  int r = 0;
  guarded_host([&] { KeyStamp ks; ProvingKeyHost pk; std::string cp = key_container_path(pk_path);
      r = !cp.empty() && key_stamp_of(pk_path, ks) && load_key_container(cp, ks, pk) ? 1 : 0; return ZKGPU_OK; });
  return r;
}
zkgpu_prover *zkgpu_prover_load(const char *pk_path) { return zkgpu_prover_load_shard(pk_path, 0, 1); }
int zkgpu_prover_prove_partial(zkgpu_prover *h, uint8_t out[384]) {
  return guarded_prover(h, [&] { if (!h) return ZKGPU_ERR_ARG; if (!h->p->prove_partial(out)) {
      zkgpu_set_error("assignment does not satisfy the constraint system (constraint " + std::to_string(h->p->last_failed_row) + " among those violated)"); return ZKGPU_ERR_UNSATISFIED; } return ZKGPU_OK; });
}
int zkgpu_prover_finish(zkgpu_prover *h, const uint8_t *records, size_t n, const uint8_t *r, const uint8_t *s, char proof_hex[513]) { return guarded_host([&] {
    if (!h) return ZKGPU_ERR_ARG; Proof p; h->p->finish_from_partials(records, n, (const Fe32 *)r, (const Fe32 *)s, p);
  std::string hx = proof_to_hex(p); memcpy(proof_hex, hx.c_str(), 513); return ZKGPU_OK; }); }
zkgpu_prover *zkgpu_prover_clone(zkgpu_prover *h) {
  zkgpu_prover *out = nullptr;
  guarded([&] { if (!h) return ZKGPU_ERR_ARG; std::unique_ptr<zkgpu_prover> p(new zkgpu_prover); p->p.reset(new Prover(*h->p)); out = p.release();
      return ZKGPU_OK; });
  return out;
}
int zkgpu_prover_prove_batch(zkgpu_prover *h, const uint8_t *zs, size_t n, const uint8_t *rs, char *proofs_hex) { return guarded_prover(h, [&] {
  if (!h || (n && (!zs || !proofs_hex))) return ZKGPU_ERR_ARG; if (!n) return ZKGPU_OK;
  static const size_t want = [] { const char *e = getenv("ZK_BATCH_LANES"); long v = e ? atol(e) : 6; return (size_t)(v < 1 ? 1 : v > 7 ? 7 : v); }();
  const size_t K = std::min(want, n);
  {
    std::lock_guard<std::mutex> lk(g_gpu_mutex);
    while (h->lanes.size() + 1 < K) h->lanes.push_back(std::make_shared<Prover>(*h->p));
  }
  const size_t zbytes = 32 * h->p->num_variables(); std::vector<uint8_t> bad(n, 0); std::vector<std::string> errs(K); std::vector<std::thread> th; std::atomic<size_t> next{0};
  auto work = [&](size_t lane) { Prover &pv = lane ? *h->lanes[lane - 1] : *h->p;
    try {
      // (the next statement nobody has taken: a lane that is slowed down — a preempted helper, the lane that shares its SIMDs with another's H accumulation —
      // simply takes fewer, and the batch ends when the last statement does, not when the unluckiest lane has worked off its fixed share)
      for (size_t i = next.fetch_add(1, std::memory_order_relaxed); i < n; i = next.fetch_add(1, std::memory_order_relaxed)) {
        Proof pr;
        const Fe32 *r = rs ? (const Fe32 *)(rs + 64 * i) : nullptr, *s_ = rs ? (const Fe32 *)(rs + 64 * i + 32) : nullptr;
        if (!pv.prove((const Fe32 *)(zs + zbytes * i), r, s_, pr)) {
          bad[i] = 1;
          pr = default_proof();
        }
        std::string hx = proof_to_hex(pr);
        memcpy(proofs_hex + 513 * i, hx.c_str(), 513);
      }
    }
    catch (const std::exception &e) { errs[lane] = e.what(); } catch (...) { errs[lane] = "unknown error"; } };
  for (size_t lane = 1; lane < K; lane++) th.emplace_back(work, lane);
  work(0); for (auto &t : th) t.join();
  for (auto &e : errs) if (!e.empty()) throw std::runtime_error(e);
  std::string which; for (size_t i = 0; i < n; i++) if (bad[i]) which += (which.empty() ? "" : ", ") + std::to_string(i);
  if (!which.empty()) { zkgpu_set_error("assignment does not satisfy the constraint system: batch record(s) " + which); return ZKGPU_ERR_UNSATISFIED; }
  return ZKGPU_OK; }); }
void zkgpu_prover_destroy(zkgpu_prover *h) { guarded([&] { delete h; return ZKGPU_OK; }); }
int zkgpu_prover_info(zkgpu_prover *h, size_t out[3]) {
  if (!h) return ZKGPU_ERR_ARG;
  out[0] = h->p->num_variables();
  out[1] = h->p->num_inputs();
  out[2] = h->p->domain_size();
  return ZKGPU_OK;
}
int zkgpu_prover_prove(zkgpu_prover *h, const uint8_t *z, const uint8_t *r, const uint8_t *s, char proof_hex[513]) { return guarded_prover(h, [&] {
    if (!h) return ZKGPU_ERR_ARG; Proof p;
  if (!h->p->prove((const Fe32 *)z, (const Fe32 *)r, (const Fe32 *)s, p)) { zkgpu_set_error("assignment does not satisfy the constraint system (constraint " + std::to_string(h->p->last_failed_row) + " among those violated)");
      return ZKGPU_ERR_UNSATISFIED; } std::string hx = proof_to_hex(p); memcpy(proof_hex, hx.c_str(), 513); return ZKGPU_OK; }); }
int zkgpu_prover_set_witness(zkgpu_prover *h, const uint8_t *z) {
  return guarded_prover(h, [&] { if (!h) return ZKGPU_ERR_ARG; h->p->set_witness((const Fe32 *)z, false); gpu_sync(); return ZKGPU_OK; });
}
int zkgpu_prover_prove_resident(zkgpu_prover *h, const uint8_t *r, const uint8_t *s, char proof_hex[513]) { return guarded_prover(h, [&] {
    if (!h) return ZKGPU_ERR_ARG; Proof p;
  if (!h->p->prove_resident((const Fe32 *)r, (const Fe32 *)s, p)) { zkgpu_set_error("assignment does not satisfy the constraint system (constraint " + std::to_string(h->p->last_failed_row) + " among those violated)");
      return ZKGPU_ERR_UNSATISFIED; } std::string hx = proof_to_hex(p); memcpy(proof_hex, hx.c_str(), 513); return ZKGPU_OK; }); }
int zkgpu_prover_stash_witness(zkgpu_prover *h, uint32_t *slot) {
  return guarded_prover(h, [&] { if (!h || !slot) return ZKGPU_ERR_ARG; *slot = (uint32_t)h->p->stash_witness(); return ZKGPU_OK; });
}
int zkgpu_prover_drop_stash(zkgpu_prover *h, uint32_t slot) {
  return guarded_prover(h, [&] { if (!h) return ZKGPU_ERR_ARG; h->p->drop_stash(slot == 0xffffffffu ? (size_t)-1 : (size_t)slot); return ZKGPU_OK; });
}
/* process-wide: how often a fast MSM path raised its flag and the MSM was repeated on the general path (soak runs, tests) */
uint64_t zkgpu_general_path_repeats(void) { return general_path_repeats(); }
uint64_t zkgpu_queries_without_tables(void) { return queries_without_tables(); }
int zkgpu_prover_equal_column_groups(zkgpu_prover *h, uint32_t *count) { if (!h || !count) return ZKGPU_ERR_ARG; *count = (uint32_t)h->p->equal_column_groups(); return ZKGPU_OK; }
int zkgpu_prover_read_stash(zkgpu_prover *h, uint32_t slot, uint8_t *z_out) {
  return guarded_prover(h, [&] { if (!h || !z_out) return ZKGPU_ERR_ARG; h->p->read_stash(slot, (Fe32 *)z_out); return ZKGPU_OK; });
}
int zkgpu_prover_stash_count(zkgpu_prover *h, uint32_t *count) {
  if (!h || !count) return ZKGPU_ERR_ARG;
  *count = (uint32_t)h->p->stash_count(); return ZKGPU_OK;
}
int zkgpu_prover_prove_stashed(zkgpu_prover *h, uint32_t slot, const uint8_t *r, const uint8_t *s, char proof_hex[513]) { return guarded_prover(h, [&] {
    if (!h) return ZKGPU_ERR_ARG; Proof p;
  if (!h->p->prove_stashed(slot, (const Fe32 *)r, (const Fe32 *)s, p)) { zkgpu_set_error("assignment does not satisfy the constraint system (constraint " + std::to_string(h->p->last_failed_row) + " among those violated)");
      return ZKGPU_ERR_UNSATISFIED; } std::string hx = proof_to_hex(p); memcpy(proof_hex, hx.c_str(), 513); return ZKGPU_OK; }); }
int zkgpu_prover_timings(zkgpu_prover *h, double out[5]) {
  if (!h) return ZKGPU_ERR_ARG;
  out[0] = h->p->last.upload_ms;
  out[1] = h->p->last.qap_ms;
  out[2] = h->p->last.msm_ms;
  out[3] = h->p->last.finish_ms;
  out[4] = h->p->last.total_ms;
  return ZKGPU_OK;
}
int zkgpu_profile_enable(int on) { return guarded([&] { profile_enable(on != 0); return ZKGPU_OK; }); }
int zkgpu_profile_report(char *buf, size_t cap) {
  return guarded([&] { std::string r = profile_report(); if (r.size() + 1 > cap) return ZKGPU_ERR_ARG; memcpy(buf, r.c_str(), r.size() + 1); return ZKGPU_OK;
      });
}
// host verifier on the prepared key (cached by the file's size and mtime)
int zkgpu_verify(const char *vk_path, const char *proof_hex, const uint8_t *inputs, size_t n_inputs) { int res = 0; int rc = guarded_host([&] {
    std::shared_ptr<PreparedVerifyingKey> vk = vk_for_path(vk_path); Proof p;
  if (!proof_hex || strnlen(proof_hex, 512) < 512 || !proof_from_hex(proof_hex, p)) { res = 0; return ZKGPU_OK;
      } res = verify_proof(*vk, (const Fe32 *)inputs, n_inputs, p) ? 1 : 0; return ZKGPU_OK; }); return rc == ZKGPU_OK ? res : rc; }
/* test entry: the decision of the GPU verifier's schedule (verify_sched.hpp), interpreted on the HOST — no device needed; stats[8] (optional): rounds, slots, products, linear operations, constants, rounds of products / eight-lane sums / one-lane sums */
int zkgpu_test_verify_schedule(const char *vk_path, const char *proof_hex, const uint8_t *inputs, size_t n_inputs, uint32_t *stats) { int res = 0;
    int rc = guarded_host([&] { std::shared_ptr<PreparedVerifyingKey> vk = vk_for_path(vk_path); Proof p;
  if (!proof_hex || strnlen(proof_hex, 512) < 512 || !proof_from_hex(proof_hex, p)) { res = 0; return ZKGPU_OK;
      } res = verify_by_schedule_on_host(*vk, (const Fe32 *)inputs, n_inputs, p, stats) ? 1 : 0; return ZKGPU_OK; }); return rc == ZKGPU_OK ? res : rc; }
/* small verification calls taken by the key's GPU verifier / launches made for them (calls that meet share a launch) */
int zkgpu_verify_counters(const char *vk_path, uint64_t out[2]) { return guarded([&] { if (!vk_path || !out) return ZKGPU_ERR_ARG; gpu_verifier_for_path(vk_path)->counters(out); return ZKGPU_OK; }); }
/* test entry (GPU): kernel K9's values after every `every`-th round of its schedule against the host model of the same limb arithmetic.  out[0] = first differing round or -1,
 * out[1] = the slot, out[2] = the kernel's verdict (1 accept, 0 reject, 2 handed back) */
int zkgpu_test_verify_trace(const char *vk_path, const char *proof_hex, const uint8_t *inputs, size_t n_inputs, uint32_t every, long out[3]) { return guarded([&] {
  if (!vk_path || !proof_hex || !out || !every) return ZKGPU_ERR_ARG; Proof p; if (strnlen(proof_hex, 512) < 512 || !proof_from_hex(proof_hex, p)) return ZKGPU_ERR_ARG;
  std::shared_ptr<BatchVerifier> v = gpu_verifier_for_path(vk_path); std::shared_ptr<PreparedVerifyingKey> vk = vk_for_path(vk_path); uint32_t slot = 0; uint8_t ok = 0;
  out[0] = verify_schedule_trace_on_device(*v, *vk, (const Fe32 *)inputs, n_inputs, p, every, &slot, &ok); out[1] = slot; out[2] = ok; return ZKGPU_OK; }); }
// batched verification on the GPU (kernel K9).  proofs_hex: n * 512 characters; inputs: n * n_inputs canonical field elements; ok[i] = 1 accept / 0 reject
// (a record that is not 512 hex digits is rejected without reaching the device; coordinates are taken modulo q like proof_from_hex in zkgpu_verify)
int zkgpu_verify_batch(const char *vk_path, const char *proofs_hex, const uint8_t *inputs, size_t n_inputs, size_t n, uint8_t *ok) { return guarded([&] {
  if (!vk_path || (!proofs_hex && n) || !ok) return ZKGPU_ERR_ARG;
  struct { std::shared_ptr<BatchVerifier> v; } slot{gpu_verifier_for_path(vk_path)};                                  // caller holds the device mutex (guarded)
  // strong IC: wrong input count rejects (r1cs_gg_ppzksnark.tcc:584-590)
  if (slot.v->num_inputs() != n_inputs) {
    for (size_t i = 0; i < n; i++) ok[i] = 0;
    return ZKGPU_OK;
  }
  std::vector<Proof> ps(n); std::vector<uint8_t> parsed(n);
  for (size_t i = 0; i < n; i++) {
    parsed[i] = strnlen(proofs_hex + 512 * i, 512) == 512 && proof_from_hex(proofs_hex + 512 * i, ps[i]);
    if (!parsed[i]) memset(&ps[i], 0, sizeof(Proof));
  }
  slot.v->verify(ps.data(), (const Fe32 *)inputs, n, ok);
  // 2: input accumulator at infinity, the host verifier decides (pairing.cuh)
  for (size_t i = 0; i < n; i++) {
    if (!parsed[i]) ok[i] = 0;
    else if (ok[i] == 2) ok[i] = verify_proof(*vk_for_path(vk_path), (const Fe32 *)inputs + i * n_inputs, n_inputs, ps[i]) ? 1 : 0;
  }
  return ZKGPU_OK; }); }
// ---- verifyBatch: the optional batch entry of include/zk_batch.h --------------------------------------------------- go-ethereum checks
// every ZK transaction twice, once in the pool and once in the block (core/tx_pool.go:612-645, core/state_processor.go:106-163), one cgo call and one key load
// per proof. A block's worth of proofs in ONE call is what the GPU verifier is for (kernel K9: one workgroup per proof interpreting the operation schedule of
// verify_sched.hpp on 29-bit limbs — 2.1 ms per launch up to 256 proofs, 31,000 proofs/s at 64, 124,000 at 512): the records are grouped by circuit kind and
// every group goes through verify_group above, exactly like the kind's verifyXproof symbol.
int verifyBatch(const zk_verify_item *items, int n, unsigned char *ok) {
  if (n < 0 || (n && (!items || !ok))) return -1;
  try {
    int accepted = 0; std::vector<int> idx[4];
    for (int i = 0; i < n; i++) { ok[i] = 0; if (items[i].kind >= 0 && items[i].kind <= 3) idx[items[i].kind].push_back(i); }
    for (int k = 0; k < 4; k++) { if (idx[k].empty()) continue; const CircuitKind kind = (CircuitKind)k; const size_t m = idx[k].size();
      std::vector<Proof> ps(m); std::vector<uint8_t> parsed(m), res(m, 0); std::vector<Fe32> inputs; size_t ni = 0;
      for (size_t j = 0; j < m; j++) {
        const zk_verify_item &it = items[idx[k][j]];
        parsed[j] = it.proof && strnlen(it.proof, 512) == 512 && proof_from_hex(it.proof, ps[j]);
        if (!parsed[j]) memset(&ps[j], 0, sizeof(Proof));
        std::vector<Fe32> in = pack_public_bits(public_bits(kind, it.args, it.value_s)); ni = in.size(); inputs.insert(inputs.end(), in.begin(), in.end()); }
      verify_group(kind, ps.data(), parsed.data(), inputs.data(), ni, m, res.data());
      for (size_t j = 0; j < m; j++) { ok[idx[k][j]] = res[j]; accepted += res[j]; } }
    return accepted;
  }
  catch (const std::exception &e) {
    zkgpu_set_error(e.what());
    fprintf(stderr, "libzkgpu: verifyBatch: %s\n", e.what());
    for (int i = 0; i < n; i++) ok[i] = 0;
    return -1;
  }
  catch (...) { for (int i = 0; i < n; i++) ok[i] = 0; return -1; }
}

// the reference's symbol names, exported by libzkgpu.so itself (the four libzk_*.so forward to the zkgpu_abi_* names above)
char *genCMT(uint64_t v, char *a, char *b) { return zkgpu_abi_genCMT(v, a, b); }
char *genCMTS(uint64_t v, char *a, char *b, char *c) { return zkgpu_abi_genCMTS(v, a, b, c); }
char *computePRF(char *a, char *b) { return zkgpu_abi_computePRF(a, b); }
char *computeCRH(char *a, char *b) { return zkgpu_abi_computeCRH(a, b); }
char *genRoot(char *a, int n) { return zkgpu_abi_genRoot(a, n); }
char *genMintproof(uint64_t a, uint64_t b, char *c, char *d, char *e, char *f, char *g, char *h, uint64_t i, char *j) {
  return zkgpu_abi_genMintproof(a, b, c, d, e, f, g, h, i, j);
}
bool verifyMintproof(char *a, char *b, char *c, char *d, uint64_t e) { return zkgpu_abi_verifyMintproof(a, b, c, d, e); }
char *genRedeemproof(uint64_t a, uint64_t b, char *c, char *d, char *e, char *f, char *g, char *h, uint64_t i, char *j) {
  return zkgpu_abi_genRedeemproof(a, b, c, d, e, f, g, h, i, j);
}
bool verifyRedeemproof(char *a, char *b, char *c, char *d, uint64_t e) { return zkgpu_abi_verifyRedeemproof(a, b, c, d, e); }
char *genSendproof(uint64_t a, char *b, char *c, char *d, char *e, char *f, uint64_t g, char *h, uint64_t i, char *j, char *k, char *l, char *m, char *n) {
  return zkgpu_abi_genSendproof(a, b, c, d, e, f, g, h, i, j, k, l, m, n);
}
bool verifySendproof(char *a, char *b, char *c, char *d, char *e) { return zkgpu_abi_verifySendproof(a, b, c, d, e); }
char *genDepositproof(uint64_t a, uint64_t b, char *c, char *d, char *e, char *f, char *g, char *h, char *i, char *j, uint64_t k, char *l, char *m, char *n,
    char *o, int p, char *q, char *r) {
  return zkgpu_abi_genDepositproof(a, b, c, d, e, f, g, h, i, j, k, l, m, n, o, p, q, r);
}
bool verifyDepositproof(char *a, char *b, char *c, char *d, char *e, char *f, char *g) { return zkgpu_abi_verifyDepositproof(a, b, c, d, e, f, g); }
}  // extern "C"
