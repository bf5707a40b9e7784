// TEST INFRASTRUCTURE — host side of libzkgpu under AddressSanitizer + UndefinedBehaviorSanitizer on a box without a GPU (SURVEY.md §5 "race detection / sanitizers").
// Built by `make -C blockmaze_amd/csrc sanitize` against libzkgpu_san.so (the host translation units instrumented, the HIP objects as they are) and run by
// tests/test_sanitizers_cpu.py.  It goes through the C-ABI only, the way a caller would, with the malformed inputs the reference mishandles: convertFromAscii falls off
// its switch for a non-hex character and the proof parser reads 512 characters whatever the string holds (sendcgo.cpp:25-35, :388-448), genCMT writes past its
// buffer (sendcgo.cpp:249-251).  Prints "SANITIZE OK" when every call returned and behaved; any sanitizer report aborts the run.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/zk_mint.h"
#include "../include/zk_send.h"
#include "../include/zk_deposit.h"
#include "../include/zk_redeem.h"
#include "../include/zk_batch.h"
#include "../include/zkgpu.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #c); fails++; } } while (0)
static std::string slurp(const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); if (!f) return ""; std::string s; char b[65536]; size_t n; while ((n = fread(b, 1, sizeof b, f)) > 0) s.append(b, n); fclose(f); return s; }
static void spit(const std::string &p, const std::string &s) { FILE *f = fopen(p.c_str(), "wb"); fwrite(s.data(), 1, s.size(), f); fclose(f); }
static char *S(const char *s) { return const_cast<char *>(s); }

int main(int argc, char **argv) {
  if (argc < 3) { fprintf(stderr, "usage: host_san_driver <tests/golden dir> <scratch dir>\n"); return 2; }
  const std::string gold = argv[1], tmp = argv[2]; setenv("ZK_PRFKEY_DIR", tmp.c_str(), 1);
  // ---- hashes: every shape of string the hex reader can meet (uint256.h:222-248: blanks and 0x skipped, digits read from the end, zero-extension, stop at the first non-hex)
  const char *odd[] = {"", "0x", "0X", "   0x12", "zz", "0xzz12", "12zz", "0x1", std::string(10000, 'f').insert(0, "0x").c_str(), "0x\xff\xfe", "-1", "0x 12"};
  std::string longs = "0x" + std::string(10000, 'f'); odd[8] = longs.c_str();
  for (const char *a : odd) for (const char *b : odd) { char *h = computePRF(S(a), S(b)); CHECK(h && strlen(h) == 64); free(h); h = genCMT(7, S(a), S(b)); CHECK(h && strlen(h) == 64); free(h);
    h = computeCRH(S(a), S(b)); CHECK(h && strlen(h) == 64); free(h); h = genCMTS(~0ull, S(a), S(b), S(a)); CHECK(h && strlen(h) == 64); free(h); }
  { char *h = genRoot(S(""), 0); CHECK(h && !strcmp(h, "8eb3c27b218349e6b9b6037b8042f3751ee820e8a0319a1bda439b247456088c")); free(h);
    h = genRoot(S("0x01"), 300); CHECK(h && strlen(h) == 64); free(h); h = genRoot(nullptr, 5); CHECK(h && strlen(h) == 64); free(h); h = genRoot(S("junk"), -3); CHECK(h && strlen(h) == 64); free(h); }
  // ---- verify symbols: no key files in the scratch directory, and every kind of proof string
  std::string hex512(512, 'a'), nonhex(512, 'g'), shorty(100, '1');
  for (const std::string &p : {hex512, nonhex, shorty, std::string("")}) { CHECK(!verifySendproof(S(p.c_str()), S("0x1"), S("0x2"), S("0x3"), S("0x4"))); CHECK(!verifyMintproof(S(p.c_str()), S("zz"), S(""), S("0x3"), 5));
    CHECK(!verifyRedeemproof(S(p.c_str()), S("0x1"), S("0x2"), S("0x3"), 0)); CHECK(!verifyDepositproof(S(p.c_str()), S("0x1"), S("0x2"), S("0x3"), S("0x4"), S("0x5"), S("0x6"))); }
  CHECK(!verifySendproof(nullptr, S("0x1"), S("0x2"), S("0x3"), S("0x4")));
  // ---- proof generation without a device: the failure sentinel (default proof), never a crash
  { char *p = genSendproof(22, S("0x1"), S("0x2"), S("0x3"), S("0x4"), S("0x5"), 8, S("0x123"), 14, S("0x6"), S("0x12"), S("0x7"), S("0x1"), S("0x456")); CHECK(p && strlen(p) == 512 && !strncmp(p, "0000000000", 10)); free(p);
    p = genDepositproof(1, 2, S("0x1"), S("0x2"), S("0x3"), S("0x4"), S("0x5"), S("0x6"), S("0x7"), S("0x8"), 3, S("0x9"), S("0xa"), S("0xb"), S("not an array"), 400, S("0xc"), S("0xd")); CHECK(p && !strncmp(p, "0000000000", 10)); free(p); }
  // ---- verification keys: the golden file, and the same file cut short or damaged at many places
  for (const char *name : {"groth16_small", "groth16_step"}) { const std::string d = gold + "/" + name, vk = slurp(d + "/vk.txt"), meta = slurp(d + "/meta.json"); CHECK(!vk.empty());
    size_t a = meta.find("\"proof\": \"") + 10; std::string proof = meta.substr(a, 512); std::string wit = slurp(d + "/wit.bin"); size_t ni = name[8] == 's' && name[9] == 'm' ? 3 : 4;
    CHECK(zkgpu_verify((d + "/vk.txt").c_str(), proof.c_str(), (const uint8_t *)wit.data() + 8, ni) == 1);
    std::string bad = proof; bad[77] = bad[77] == '0' ? '1' : '0'; CHECK(zkgpu_verify((d + "/vk.txt").c_str(), bad.c_str(), (const uint8_t *)wit.data() + 8, ni) == 0); CHECK(zkgpu_verify((d + "/vk.txt").c_str(), proof.c_str(), (const uint8_t *)wit.data() + 8, ni - 1) == 0);
    const std::string t = tmp + "/vk_cut.txt";
    for (size_t cut = 0; cut < vk.size(); cut += 41) { spit(t, vk.substr(0, cut)); CHECK(zkgpu_verify(t.c_str(), proof.c_str(), (const uint8_t *)wit.data() + 8, ni) <= 0); }
    for (size_t pos = 3; pos < vk.size(); pos += 53) { std::string v = vk; v[pos] ^= 0x55; spit(t, v); (void)zkgpu_verify(t.c_str(), proof.c_str(), (const uint8_t *)wit.data() + 8, ni); }   // any answer but a crash
    // proving keys: the reader's error paths (a complete file reaches the point decompression, which needs the device and reports that)
    const std::string pk = slurp(d + "/pk.txt"), tp = tmp + "/pk_cut.txt";
    for (size_t cut = 0; cut <= pk.size(); cut += pk.size() / 37 + 1) { spit(tp, pk.substr(0, cut)); setenv("ZK_KEY_CACHE", "0", 1); zkgpu_prover *h = zkgpu_prover_load(tp.c_str()); CHECK(h == nullptr); }
    for (size_t pos = 200; pos < pk.size(); pos += pk.size() / 29 + 1) { std::string v = pk; v[pos] = '9'; spit(tp, v); zkgpu_prover *h = zkgpu_prover_load(tp.c_str()); CHECK(h == nullptr); } }
  // ---- circuits and witness generation (pure host): all four statement circuits on odd inputs
  { const std::string w = tmp + "/w.bin", r = tmp + "/r.bin"; CHECK(zkgpu_circuit_export(1, 8, r.c_str()) == ZKGPU_OK); CHECK(zkgpu_circuit_export(102, 8, r.c_str()) == ZKGPU_OK); CHECK(zkgpu_circuit_export(101, 3, r.c_str()) == ZKGPU_OK);
    CHECK(zkgpu_witness_send(22, S("0x1"), S("zz"), S(""), S("0x4"), S("0x5"), 8, S("0x123"), 14, S("0x6"), S("0x12"), S("0x7"), S("0x1"), S("0x456"), w.c_str()) == ZKGPU_OK);
    CHECK(zkgpu_witness_mint_redeem(0, ~0ull, 1, S("0x1"), S("0x2"), S("0x3"), S("0x4"), S("0x5"), S("0x6"), ~0ull, S("0x7"), w.c_str()) == ZKGPU_OK); CHECK(zkgpu_witness_mint_redeem(1, 5, 9, S("0x1"), S("0x2"), S("0x3"), S("0x4"), S("0x5"), S("0x6"), 20, S("0x7"), w.c_str()) == ZKGPU_OK);
    std::string arr; for (int i = 0; i < 16; i++) { char b[80]; snprintf(b, sizeof b, "0x%064x", i + 1); arr += b; }
    CHECK(zkgpu_witness_deposit(264, 255, S("0x1"), S("0x2"), S("0x3"), S("0x4"), S("0x5"), S("0x6"), S("0x7"), S("0x8"), 9, S("0x9"), S("0xa"), S("0x0a"), S(arr.c_str()), 16, S("0xd"), 8, w.c_str()) == ZKGPU_OK);          // cmtS = leaf 10
    CHECK(zkgpu_witness_deposit(264, 255, S("0x1"), S("0x2"), S("0x3"), S("0x4"), S("0x5"), S("0x6"), S("0x7"), S("0x8"), 9, S("0x9"), S("0xa"), S("0xff"), S(arr.c_str()), 16, S("0xd"), 8, w.c_str()) != ZKGPU_OK);          // cmtS not among the leaves
    CHECK(zkgpu_witness_lesscmp(5, 9, w.c_str()) == ZKGPU_OK); }
  // ---- container, device plan, batch entry
  CHECK(zkgpu_test_key_container((tmp + "/k.gpucache").c_str(), 300, 500, 512) == 0);
  CHECK(zkgpu_test_scan_pool(3, 200) >= 0);
  { int dev[64], ord[8]; CHECK(zkgpu_test_device_plan("0,1,,x", 4, 0, 2, dev, ord, 8) >= 1); CHECK(zkgpu_test_device_plan(nullptr, 0, 0, 2, dev, ord, 8) == 0); }
  { zk_verify_item it[3]; memset(it, 0, sizeof it); it[0].kind = 1; it[0].proof = hex512.c_str(); it[1].kind = 99; it[2].kind = 2; it[2].proof = nullptr; unsigned char ok[3] = {9, 9, 9}; CHECK(verifyBatch(it, 3, ok) == -1 && !ok[0] && !ok[1] && !ok[2]); CHECK(verifyBatch(nullptr, 0, nullptr) == 0); }
  if (fails) { fprintf(stderr, "%d check(s) failed\n", fails); return 1; }
  printf("SANITIZE OK\n"); return 0;
}
