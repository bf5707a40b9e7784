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
  if (argc < 2) { fprintf(stderr, "usage: host_tsan_driver <scratch dir> [threads] [rounds] [tests/golden dir]\n"); return 2; }
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
  if (fails.load()) { fprintf(stderr, "%d checks failed\n", fails.load()); return 1; }
  printf("TSAN OK\n"); return 0;
}
