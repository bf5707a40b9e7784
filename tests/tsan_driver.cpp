// TEST INFRASTRUCTURE — the multi-threaded host paths of libzkgpu under ThreadSanitizer on a box without a GPU (SURVEY.md §5 "race detection / sanitizers").
// Built by `make -C blockmaze_amd/csrc tsan` against libzkgpu_tsan.so (the host translation units instrumented, the HIP objects as they are, never called) and run by
// tests/test_sanitizers_cpu.py.  What runs on several threads on the host: the witness generators (one wave of SHA-256 compressions handed to the task pool's helper
// threads while the caller works on its share — circuit.cpp TaskPool, blockmaze_circuits.cpp), from several callers at once as cgo delivers them.  Every assignment is
// also compared with the one the first call made for the same statement: the generators are deterministic.  Prints "TSAN OK"; a report makes the exit status 66.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <atomic>
#include "../include/zkgpu.h"
#include "../include/zk_send.h"

static char *S(const char *s) { return const_cast<char *>(s); }
static std::string slurp(const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); if (!f) return ""; std::string s; char b[65536]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) s.append(b, n); fclose(f); return s; }
static std::atomic<int> fails{0};
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed at line %d: %s (%s)\n", __LINE__, #c, zkgpu_last_error()); fails++; } } while (0)

static int make(int kind, int variant, const std::string &path) {
  const uint64_t v = 100 + variant;
  std::string arr; for (int i = 0; i < 16; i++) { char b[80]; snprintf(b, sizeof b, "0x%064x", i + 1); arr += b; }   // (sixteen commitments back to back; cmtS = 0x0a is the tenth)
  switch (kind) {
    case 0: return zkgpu_witness_send(v, S("0x1"), S("0x22"), S("0x3"), S("0x4"), S("0x5"), 8, S("0x123"), v - 8, S("0x6"), S("0x12"), S("0x7"), S("0x1"), S("0x456"), path.c_str());
    case 1: return zkgpu_witness_mint_redeem(0, v + 7, v, S("0x1"), S("0x2"), S("0x3"), S("0x4"), S("0x5"), S("0x6"), 7, S("0x7"), path.c_str());
    case 2: return zkgpu_witness_mint_redeem(1, v - 9, v, S("0x1"), S("0x2"), S("0x3"), S("0x4"), S("0x5"), S("0x6"), 9, S("0x7"), path.c_str());
    default: return zkgpu_witness_deposit(255 + v, 255, S("0x1"), S("0x2"), S("0x3"), S("0x4"), S("0x5"), S("0x6"), S("0x7"), S("0x8"), v, S("0x9"), S("0xa"), S("0x0a"), S(arr.c_str()), 16, S("0xd"), 8, path.c_str());
  }
}

int main(int argc, char **argv) {
  if (argc < 2) { fprintf(stderr, "usage: host_tsan_driver <scratch dir> [threads] [rounds] [tests/golden dir] [gpu]\n"); return 2; }
  const std::string tmp = argv[1]; const int T = argc > 2 ? atoi(argv[2]) : 3, R = argc > 3 ? atoi(argv[3]) : 6;
  std::string want[4][2];
  for (int k = 0; k < 4; k++) for (int var = 0; var < 2; var++) { const std::string p = tmp + "/ref.bin"; CHECK(make(k, var, p) == ZKGPU_OK); want[k][var] = slurp(p); CHECK(want[k][var].size() > 1000000); }
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back([&, t] {
    const std::string p = tmp + "/w" + std::to_string(t) + ".bin";
    for (int r = 0; r < R; r++) for (int k = 0; k < 4; k++) { const int var = (r + t) & 1; CHECK(make((k + t) & 3, var, p) == ZKGPU_OK); CHECK(slurp(p) == want[(k + t) & 3][var]); }
  });
  for (auto &x : th) x.join();
  // the host verifier (prepared keys cached per file behind a mutex) and the hash symbols from several threads: what geth's txpool and block processor do on a host without a GPU
  if (argc > 4) {
    const std::string d = std::string(argv[4]) + "/groth16_small", meta = slurp(d + "/meta.json"), wit = slurp(d + "/wit.bin"); const size_t a = meta.find("\"proof\": \"") + 10; const std::string proof = meta.substr(a, 512);
    std::string bad = proof; bad[77] = bad[77] == '0' ? '1' : '0'; CHECK(wit.size() > 8 && proof.size() == 512);
    std::vector<std::thread> tv;
    for (int t = 0; t < T; t++) tv.emplace_back([&, t] {
      for (int r = 0; r < 4; r++) { CHECK(zkgpu_verify((d + "/vk.txt").c_str(), proof.c_str(), (const uint8_t *)wit.data() + 8, 3) == 1); CHECK(zkgpu_verify((d + "/vk.txt").c_str(), bad.c_str(), (const uint8_t *)wit.data() + 8, 3) == 0); }
    });
    for (auto &x : tv) x.join();
  }
  CHECK(zkgpu_test_scan_pool(4, 50) >= 0);                                               // the hand-over's scan pool from four callers (host only)
  // on a box WITH a GPU (argv[5] = "gpu"): the cgo symbols of the send circuit from T threads — provers of a pool, their submit threads, the hand-over of a circuit
  // board, the verifier's combiner of concurrent calls.  (The HIP runtime is not instrumented: its own threads are silenced in tests/tsan_gpu.supp.)
  if (argc > 5 && !strcmp(argv[5], "gpu") && zkgpu_device_count() > 0) {
    CHECK(zkgpu_keygen(1 /* send */, 8, 0xB10C4A2Eull, (tmp + "/sendpk.txt").c_str(), (tmp + "/sendvk.txt").c_str()) == ZKGPU_OK);
    setenv("ZK_PRFKEY_DIR", tmp.c_str(), 1);
    auto ox = [](char *h) { std::string r = std::string("0x") + h; free(h); return r; };
    std::vector<std::thread> tg; std::atomic<int> made{0}, accepted{0};
    for (int t = 0; t < T; t++) tg.emplace_back([&, t] {
      for (int r = 0; r < 12; r++) {
        char sk[80], r_old[80], rr[80]; snprintf(sk, sizeof sk, "0x%064x", 1 + t); snprintf(r_old, sizeof r_old, "0x%064x", 0x123456 + r); snprintf(rr, sizeof rr, "0x%064x", 0x12 + 16 * r + t);
        const char *pk_sender = "0x0000000000000000000000000000000000000456", *pk_recv = "0x0000000000000000000000000000000000000123"; const uint64_t value_old = 22 + r, value_s = 8, value = value_old - value_s;
        const std::string sn_old = ox(computePRF(sk, r_old)), cmtA_old = ox(genCMT(value_old, S(sn_old.c_str()), r_old)), r_s = ox(computeCRH(S(pk_sender), rr)), sn = ox(computePRF(sk, rr));
        const std::string cmtS = ox(genCMTS(value_s, S(pk_recv), S(r_s.c_str()), S(sn_old.c_str()))), cmtA = ox(genCMT(value, S(sn.c_str()), rr));
        char *p = genSendproof(value_old, S(r_s.c_str()), S(sn_old.c_str()), r_old, S(cmtS.c_str()), S(cmtA_old.c_str()), value_s, S(pk_recv), value, S(sn.c_str()), rr, S(cmtA.c_str()), sk, S(pk_sender));
        CHECK(p && strlen(p) == 512 && strncmp(p, "0000000000", 10) != 0); made++;
        if (p && verifySendproof(p, S(cmtA_old.c_str()), S(sn_old.c_str()), S(cmtS.c_str()), S(cmtA.c_str()))) accepted++;
        free(p);
      }
    });
    for (auto &x : tg) x.join();
    CHECK(made.load() == 12 * T && accepted.load() == 12 * T);
    printf("gpu: %d send proofs made and accepted through the cgo symbols from %d threads\n", accepted.load(), T);
    // the engine's own entry points side by side on two prover objects of one key: host buffers (the scan pool and the submit threads) and statements kept in HBM
    { const std::string wp = tmp + "/ws.bin";
      char sk[80] = "0x01", r_old[80] = "0x123456", rr[80] = "0x12"; const char *pk_sender = "0x456", *pk_recv = "0x123";
      const std::string sn_old = ox(computePRF(sk, r_old)), cmtA_old = ox(genCMT(22, S(sn_old.c_str()), r_old)), r_s = ox(computeCRH(S(pk_sender), rr)), sn = ox(computePRF(sk, rr));
      const std::string cmtS = ox(genCMTS(8, S(pk_recv), S(r_s.c_str()), S(sn_old.c_str()))), cmtA = ox(genCMT(14, S(sn.c_str()), rr));
      CHECK(zkgpu_witness_send(22, S(r_s.c_str()), S(sn_old.c_str()), r_old, S(cmtS.c_str()), S(cmtA_old.c_str()), 8, S(pk_recv), 14, S(sn.c_str()), rr, S(cmtA.c_str()), sk, S(pk_sender), wp.c_str()) == ZKGPU_OK);
      const std::string wit = slurp(wp); zkgpu_prover *h = zkgpu_prover_load((tmp + "/sendpk.txt").c_str()); CHECK(h != nullptr);
      if (h) { zkgpu_prover *h2 = zkgpu_prover_clone(h); CHECK(h2 != nullptr); const uint8_t *z = (const uint8_t *)wit.data() + 8;
        std::thread a([&] { char out[513]; for (int i = 0; i < 10; i++) CHECK(zkgpu_prover_prove(h, z, nullptr, nullptr, out) == ZKGPU_OK); });
        std::thread b([&] { char out[513]; uint32_t slot[12];                          // (bench.py's set-up and timed region: statements handed over and kept, then proved from HBM in turn)
          for (int i = 0; i < 12; i++) CHECK(zkgpu_prover_set_witness(h2, z) == ZKGPU_OK && zkgpu_prover_stash_witness(h2, &slot[i]) == ZKGPU_OK);
          for (int i = 0; i < 10; i++) CHECK(zkgpu_prover_prove_stashed(h2, slot[(5 * i) % 12], nullptr, nullptr, out) == ZKGPU_OK); });
        a.join(); b.join(); printf("gpu: 10 proofs from host buffers beside 10 from a stash on two provers of one key\n");
        // one call, eight statements (the pool's lanes hand them out among themselves), and the batch verifier on their proofs from two threads at once
        { const size_t nb = 8, zbytes = wit.size() - 8; std::vector<uint8_t> zs(nb * zbytes); for (size_t i = 0; i < nb; i++) memcpy(zs.data() + i * zbytes, z, zbytes);
          std::vector<char> proofs(513 * nb); CHECK(zkgpu_prover_prove_batch(h, zs.data(), nb, nullptr, proofs.data()) == ZKGPU_OK);
          size_t info[3] = {0, 0, 0}; CHECK(zkgpu_prover_info(h, info) == ZKGPU_OK); const size_t ni = info[1];
          std::string all; std::vector<uint8_t> inputs; for (size_t i = 0; i < nb; i++) { all.append(proofs.data() + 513 * i, 512); inputs.insert(inputs.end(), z, z + 32 * ni); }
          std::atomic<int> good{0};
          auto vb = [&] { std::vector<uint8_t> ok(nb, 0); if (zkgpu_verify_batch((tmp + "/sendvk.txt").c_str(), all.c_str(), inputs.data(), ni, nb, ok.data()) >= 0) for (uint8_t v : ok) good += v; };
          std::thread v1(vb), v2(vb); v1.join(); v2.join(); CHECK(good.load() == 2 * (int)nb);
          printf("gpu: a batch of %zu proofs made in one call, verified by two threads (%d accepted)\n", nb, good.load()); } } }
  }
  if (fails.load()) { fprintf(stderr, "%d checks failed\n", fails.load()); return 1; }
  printf("TSAN OK\n"); return 0;
}
