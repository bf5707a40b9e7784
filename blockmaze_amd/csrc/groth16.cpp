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
#include "groth16.hpp"
#include "verify_sched.hpp"

namespace zk {
using namespace host;

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static HFq fq_of(const Fe32 &f) { HFq r; memcpy(r.l, &f, 32); return r; }
static HFr fr_of(const Fe32 &f) { HFr r; memcpy(r.l, &f, 32); return r; }
static Fe32 fe_of(const HFq &f) { Fe32 r; memcpy(&r, f.l, 32); return r; }
static Fe32 fe_of_r(const HFr &f) { Fe32 r; memcpy(&r, f.l, 32); return r; }
static HG1 g1_of(const G1AffineRaw &p) { return HG1::from_affine(fq_of(p.x), fq_of(p.y)); }
static HFq2 fq2_of(const Fe32 &a, const Fe32 &b) { return {fq_of(a), fq_of(b)}; }
static HG2 g2_of(const G2AffineRaw &p) { return HG2::from_affine(fq2_of(p.x0, p.x1), fq2_of(p.y0, p.y1)); }
static G1AffineRaw raw_of(const HG1 &p) { HFq x, y; p.to_affine(x, y); return {fe_of(x), fe_of(y)}; }
static G2AffineRaw raw_of(const HG2 &p) { HFq2 x, y; p.to_affine(x, y); return {fe_of(x.c0), fe_of(x.c1), fe_of(y.c0), fe_of(y.c1)}; }
static bool is_zero_raw(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; i++) if (b[i]) return false; return true; }

// ======================================================================================================================
// key files
// ======================================================================================================================
namespace {
struct Cursor {
  const uint8_t *p, *end; const char *what;
  void fail(const char *msg) const { throw std::runtime_error(std::string(what) + ": " + msg); }
  void skip_ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++; }
  void dec(uint32_t out[8]) { skip_ws(); memset(out, 0, 32); int nd = 0;
    while (p < end && *p >= '0' && *p <= '9') {
      uint64_t carry = *p - '0';
      for (int i = 0; i < 8; i++) {
        uint64_t v = (uint64_t)out[i] * 10 + carry;
        out[i] = (uint32_t)v;
        carry = v >> 32;
      }
      p++;
      nd++;
    }
    if (!nd) fail("expected a decimal number"); }
  size_t size() {
    uint32_t v[8];
    dec(v);
    for (int i = 2; i < 8; i++) if (v[i]) fail("count or index does not fit 64 bits");
    size_t r = (size_t)v[0] | ((size_t)v[1] << 32);
    if (r >> 40) fail("implausible count or index");
    return r;
  }
  void eat(char c) { if (p < end && *p == (uint8_t)c) p++; else fail("unexpected byte"); }
  // compressed points: ASCII is_zero, raw Montgomery X, ASCII lsb(Y)   (alt_bn128_g1.cpp:404-418, alt_bn128_g2.cpp:418-431)
  void g1(std::vector<Fe32> &xs, std::vector<uint8_t> &flags) {
    if (end - p < 34) fail("truncated G1");
    uint8_t z = *p++ - '0';
    Fe32 x;
    memcpy(&x, p, 32);
    p += 32;
    uint8_t lsb = *p++ - '0';
    if (z > 1 || lsb > 1) fail("bad G1 flag");
    xs.push_back(x);
    flags.push_back((uint8_t)(lsb | (z << 1)));
  }
  void g2(std::vector<Fe32> &xs, std::vector<uint8_t> &flags) {
    if (end - p < 66) fail("truncated G2");
    uint8_t z = *p++ - '0';
    Fe32 x[2];
    memcpy(x, p, 64);
    p += 64;
    uint8_t lsb = *p++ - '0';
    if (z > 1 || lsb > 1) fail("bad G2 flag");
    xs.push_back(x[0]);
    xs.push_back(x[1]);
    flags.push_back((uint8_t)(lsb | (z << 1)));
  }
};
std::vector<uint8_t> slurp(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  f.seekg(0, std::ios::end);
  size_t n = (size_t)f.tellg();
  f.seekg(0);
  std::vector<uint8_t> b(n);
  f.read((char *)b.data(), n);
  return b;
}

void put_dec(std::string &o, const uint32_t v[8]) {
  uint32_t t[8];
  memcpy(t, v, 32);
  char buf[80];
  int n = 0;
  bool zero = true;
  for (int i = 0; i < 8; i++) if (t[i]) zero = false;
  if (zero) { o.push_back('0'); return; }
  while (true) {
    uint64_t rem = 0;
    bool nz = false;
    for (int i = 7; i >= 0; i--) {
      uint64_t cur = (rem << 32) | t[i];
      t[i] = (uint32_t)(cur / 1000000000u);
      rem = cur % 1000000000u;
      if (t[i]) nz = true;
    }
    for (int k = 0; k < 9; k++) { buf[n++] = (char)('0' + rem % 10); rem /= 10; } if (!nz) break; }
  while (n > 1 && buf[n - 1] == '0') n--; while (n) o.push_back(buf[--n]); }
void put_size(std::string &o, size_t v) { o += std::to_string(v); }
void put_fq_mont(std::string &o, const Fe32 &x) { o.append((const char *)&x, 32); }
// zero = (0, 1, 0): lsb(Y) = 1
void put_g1(std::string &o, const G1AffineRaw &p) {
  bool z = is_zero_raw(&p, sizeof p);
  o.push_back(z ? '1' : '0');
  put_fq_mont(o, p.x);
  o.push_back(z ? '1' : (char)('0' + (fq_of(p.y).from_mont().l[0] & 1)));
}
void put_g2(std::string &o, const G2AffineRaw &p) {
  bool z = is_zero_raw(&p, sizeof p);
  o.push_back(z ? '1' : '0');
  put_fq_mont(o, p.x0);
  put_fq_mont(o, p.x1);
  o.push_back(z ? '1' : (char)('0' + (fq_of(p.y0).from_mont().l[0] & 1)));
}
// vector<G1> operator of libff.so: no per-element newline
void put_g1_vec(std::string &o, const std::vector<G1AffineRaw> &v) {
  put_size(o, v.size());
  o.push_back('\n');
  for (auto &p : v) put_g1(o, p);
}
}  // namespace

static size_t domain_size_for(size_t min_size);
ProvingKeyHost load_proving_key(const std::string &path) {
  std::vector<uint8_t> buf = slurp(path); Cursor c{buf.data(), buf.data() + buf.size(), "proving key"}; ProvingKeyHost pk;
  std::vector<Fe32> x1, x2; std::vector<uint8_t> f1, f2;     // every G1 / G2 of the file, decompressed in one batch each
  // alpha_g1 beta_g1 beta_g2 delta_g1 delta_g2 (r1cs_gg_ppzksnark.tcc:52-66)
  c.g1(x1, f1);
  c.eat('\n');
  c.g1(x1, f1);
  c.eat('\n');
  c.g2(x2, f2);
  c.eat('\n');
  c.g1(x1, f1);
  c.eat('\n');
  c.g2(x2, f2);
  c.eat('\n');
  size_t nA = c.size(); c.eat('\n'); for (size_t i = 0; i < nA; i++) c.g1(x1, f1);
  size_t dom = c.size(); size_t ni = c.size(); if (ni > nA) c.fail("B query has more entries than variables"); pk.B_idx.resize(ni);
  // sparse_vector.tcc:272-288; the device gathers z[B_idx[i]]
  for (size_t i = 0; i < ni; i++) {
    size_t idx = c.size();
    if (idx >= nA) c.fail("B query index out of range");
    if (i && idx <= pk.B_idx[i - 1]) c.fail("B query indices are not increasing");
    pk.B_idx[i] = (uint32_t)idx;
  }
  size_t nB = c.size(); c.eat('\n'); if (nB != ni || dom != nA) c.fail("inconsistent B query");
  // knowledge_commitment.tcc:121-125
  for (size_t i = 0; i < nB; i++) {
    c.g2(x2, f2);
    c.eat(' ');
    c.g1(x1, f1);
    c.eat('\n');
  }
  size_t nH = c.size(); c.eat('\n'); for (size_t i = 0; i < nH; i++) c.g1(x1, f1);
  size_t nL = c.size(); c.eat('\n'); for (size_t i = 0; i < nL; i++) c.g1(x1, f1);
  // r1cs.tcc:242-254
  R1csHost &cs = pk.cs;
  cs.n_inputs = c.size();
  cs.n_vars = cs.n_inputs + c.size();
  cs.n_cons = c.size();
  for (int m = 0; m < 3; m++) { cs.rowptr[m].reserve(cs.n_cons + 1); cs.rowptr[m].push_back(0); }
  for (size_t i = 0; i < cs.n_cons; i++) for (int m = 0; m < 3; m++) { size_t nt = c.size();
    for (size_t k = 0; k < nt; k++) {
      size_t idx = c.size();
      if (idx > cs.n_vars) c.fail("variable index out of range");
      Fe32 co;
      c.dec(co.l);
      cs.col[m].push_back((uint32_t)idx);
      cs.coeff[m].push_back(co);
    }
    cs.rowptr[m].push_back((uint32_t)cs.col[m].size()); }
  if (cs.n_inputs > cs.n_vars || nA != cs.n_vars + 1 || nL != cs.n_vars - cs.n_inputs) c.fail("query sizes do not match the constraint system");
  // r1cs_gg_ppzksnark.tcc:281: m - 1 powers
  if (nH + 1 != domain_size_for(cs.n_cons + cs.n_inputs + 1)) c.fail("H query size does not match the evaluation domain");
  std::vector<G1AffineRaw> p1(x1.size()); std::vector<G2AffineRaw> p2(f2.size());
  decompress_g1(x1.data(), f1.data(), x1.size(), p1.data()); decompress_g2(x2.data(), f2.data(), f2.size(), p2.data());
  size_t i1 = 0, i2 = 0; pk.alpha_g1 = p1[i1++]; pk.beta_g1 = p1[i1++]; pk.beta_g2 = p2[i2++]; pk.delta_g1 = p1[i1++]; pk.delta_g2 = p2[i2++];
  pk.A.assign(p1.begin() + i1, p1.begin() + i1 + nA);
  i1 += nA;
  pk.B_g1.assign(p1.begin() + i1, p1.begin() + i1 + nB);
  i1 += nB;
  pk.B_g2.assign(p2.begin() + i2, p2.begin() + i2 + nB);
  pk.H.assign(p1.begin() + i1, p1.begin() + i1 + nH); i1 += nH; pk.L.assign(p1.begin() + i1, p1.begin() + i1 + nL); return pk;
}

// host-only square roots for the handful of points in a verification key
static G1AffineRaw decompress_host_g1(const Fe32 &xm, uint8_t flags) {
  if (flags & 2) {
    G1AffineRaw z;
    memset(&z, 0, sizeof z);
    return z;
  }
  HFq x = fq_of(xm), y2 = x.sqr() * x + HFq::from_u64(3), y;
  if (!fq_sqrt(y2, y)) throw std::runtime_error("verification key: G1 point not on the curve");
  if ((y.from_mont().l[0] & 1) != (uint64_t)(flags & 1)) y = y.neg(); return {fe_of(x), fe_of(y)}; }
static bool fq2_sqrt_host(const HFq2 &a, HFq2 &out) {   // Adj & Rodriguez-Henriquez Alg. 9, q = 3 mod 4
  if (a.is_zero()) { out = a; return true; }
  uint64_t e34[4], e12[4];
  {
    uint64_t t[4];
    uint64_t br = 3;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)HFq::mod(i) - br;
      t[i] = (uint64_t)d;
      br = (uint64_t)(d >> 64) & 1;
    }
    for (int i = 0; i < 4; i++) e34[i] = (t[i] >> 2) | (i < 3 ? t[i + 1] << 62 : 0);
    br = 1;
    for (int i = 0; i < 4; i++) {
      u128 d = (u128)HFq::mod(i) - br;
      t[i] = (uint64_t)d;
      br = (uint64_t)(d >> 64) & 1;
    }
    for (int i = 0; i < 4; i++) e12[i] = (t[i] >> 1) | (i < 3 ? t[i + 1] << 63 : 0);
  }
  HFq2 a1 = a.pow(e34, 4), x0 = a1 * a, alpha = a1 * x0, a0 = alpha.frob(1) * alpha, m1 = HFq2::one().neg();
  if (a0 == m1) return false; if (alpha == m1) out = HFq2{x0.c1.neg(), x0.c0}; else out = (HFq2::one() + alpha).pow(e12, 4) * x0; return out.sqr() == a; }
static G2AffineRaw decompress_host_g2(const Fe32 &x0, const Fe32 &x1, uint8_t flags) { if (flags & 2) { G2AffineRaw z; memset(&z, 0, sizeof z); return z; }
  HFq2 x = fq2_of(x0, x1), tb = HFq2{HFq::from_u64(3), HFq::zero()} * HFq2{HFq::from_u64(9), HFq::one()}.inv(), y2 = x.sqr() * x + tb, y;
  if (!fq2_sqrt_host(y2, y)) throw std::runtime_error("verification key: G2 point not on the twist");
  if ((y.c0.from_mont().l[0] & 1) != (uint64_t)(flags & 1)) y = y.neg(); return {fe_of(x.c0), fe_of(x.c1), fe_of(y.c0), fe_of(y.c1)}; }

VerifyingKeyHost load_verifying_key(const std::string &path) {   // r1cs_gg_ppzksnark.tcc:100-108, accumulation_vector.tcc:63-69
  std::vector<uint8_t> buf = slurp(path); Cursor c{buf.data(), buf.data() + buf.size(), "verification key"}; VerifyingKeyHost vk;
  HFq *gt = reinterpret_cast<HFq *>(&vk.alpha_g1_beta_g2); for (int i = 0; i < 12; i++) { Fe32 v; c.dec(v.l); gt[i] = fq_of(v).to_mont(); } c.eat('\n');
  std::vector<Fe32> x;
  std::vector<uint8_t> f;
  c.g2(x, f);
  c.eat('\n');
  vk.gamma_g2 = decompress_host_g2(x[0], x[1], f[0]);
  x.clear();
  f.clear();
  c.g2(x, f);
  c.eat('\n');
  vk.delta_g2 = decompress_host_g2(x[0], x[1], f[0]);
  x.clear(); f.clear(); c.g1(x, f); c.eat('\n'); vk.IC.push_back(decompress_host_g1(x[0], f[0]));
  size_t dom = c.size(), ni = c.size();
  for (size_t i = 0; i < ni; i++) if (c.size() != i) c.fail("sparse IC vector");
  size_t nv = c.size();
  c.eat('\n');
  if (nv != ni || dom != ni) c.fail("inconsistent IC vector");
  for (size_t i = 0; i < nv; i++) { x.clear(); f.clear(); c.g1(x, f); c.eat('\n'); vk.IC.push_back(decompress_host_g1(x[0], f[0])); }
  return vk;
}

void save_proving_key(const std::string &path, const ProvingKeyHost &pk) {
  std::string o;
  o.reserve(64 * (pk.A.size() + pk.H.size() + pk.L.size()) + 200 * pk.B_idx.size() + 40 * (pk.cs.col[0].size() + pk.cs.col[1].size() + pk.cs.col[2].size()));
  put_g1(o, pk.alpha_g1);
  o.push_back('\n');
  put_g1(o, pk.beta_g1);
  o.push_back('\n');
  put_g2(o, pk.beta_g2);
  o.push_back('\n');
  put_g1(o, pk.delta_g1);
  o.push_back('\n');
  put_g2(o, pk.delta_g2);
  o.push_back('\n');
  put_g1_vec(o, pk.A);
  put_size(o, pk.A.size());
  o.push_back('\n');
  put_size(o, pk.B_idx.size());
  o.push_back('\n');
  for (uint32_t i : pk.B_idx) {
    put_size(o, i);
    o.push_back('\n');
  }
  put_size(o, pk.B_idx.size());
  o.push_back('\n');
  for (size_t i = 0; i < pk.B_idx.size(); i++) {
    put_g2(o, pk.B_g2[i]);
    o.push_back(' ');
    put_g1(o, pk.B_g1[i]);
    o.push_back('\n');
  }
  put_g1_vec(o, pk.H); put_g1_vec(o, pk.L);
  const R1csHost &cs = pk.cs;
  put_size(o, cs.n_inputs);
  o.push_back('\n');
  put_size(o, cs.n_vars - cs.n_inputs);
  o.push_back('\n');
  put_size(o, cs.n_cons);
  o.push_back('\n');
  for (size_t i = 0; i < cs.n_cons; i++) for (int m = 0; m < 3; m++) { put_size(o, cs.rowptr[m][i + 1] - cs.rowptr[m][i]); o.push_back('\n');
    for (uint32_t k = cs.rowptr[m][i]; k < cs.rowptr[m][i + 1]; k++) {
      put_size(o, cs.col[m][k]);
      o.push_back('\n');
      put_dec(o, cs.coeff[m][k].l);
      o.push_back('\n');
    }
  }
  std::ofstream f(path, std::ios::binary); if (!f) throw std::runtime_error("cannot write " + path); f.write(o.data(), (std::streamsize)o.size());
}
void save_verifying_key(const std::string &path, const VerifyingKeyHost &vk) {
  std::string o; const HFq *gt = reinterpret_cast<const HFq *>(&vk.alpha_g1_beta_g2);
  for (int i = 0; i < 12; i++) { HFq c = gt[i].from_mont(); Fe32 v = fe_of(c); put_dec(o, v.l); if (i < 11) o.push_back(' '); } o.push_back('\n');
  put_g2(o, vk.gamma_g2); o.push_back('\n'); put_g2(o, vk.delta_g2); o.push_back('\n'); put_g1(o, vk.IC[0]); o.push_back('\n');
  size_t n = vk.IC.size() - 1;
  put_size(o, n);
  o.push_back('\n');
  put_size(o, n);
  o.push_back('\n');
  for (size_t i = 0; i < n; i++) {
    put_size(o, i);
    o.push_back('\n');
  }
  put_size(o, n); o.push_back('\n'); for (size_t i = 0; i < n; i++) { put_g1(o, vk.IC[i + 1]); o.push_back('\n'); } o.push_back('\n'); o.push_back('\n');
  std::ofstream f(path, std::ios::binary); if (!f) throw std::runtime_error("cannot write " + path); f.write(o.data(), (std::streamsize)o.size());
}

// ======================================================================================================================
// fast key container
// ======================================================================================================================
namespace {
struct ContainerHeader { char magic[8]; uint32_t version, flags; int64_t src_size, src_mtime_s, src_mtime_ns;
    uint64_t n_inputs, n_vars, n_cons, m, nA, nB, nH, nL, nnz[3], payload_bytes, checksum; uint8_t pad[256 - 8 - 8 - 24 - 8 * 13]; };
static_assert(sizeof(ContainerHeader) == 256, "container header");
const char CONTAINER_MAGIC[8] = {'Z', 'K', 'G', 'P', 'U', 'K', 'C', '1'};
// four independent multiply-xor lanes over 8-byte words (about 10 GB/s): an integrity check against truncation and bit rot, not a MAC
uint64_t checksum64(const uint8_t *p, size_t n) {
  uint64_t h[4] = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull, 0xA4093822299F31D0ull, 0x082EFA98EC4E6C89ull}; size_t i = 0;
  for (; i + 32 <= n; i += 32) {
    uint64_t w[4];
    memcpy(w, p + i, 32);
    for (int k = 0; k < 4; k++) {
      h[k] = (h[k] ^ w[k]) * 0x9E3779B97F4A7C15ull;
      h[k] ^= h[k] >> 29;
    }
  }
  for (; i < n; i++) { h[0] = (h[0] ^ p[i]) * 0x100000001B3ull; }
  return (h[0] * 3) ^ (h[1] * 5) ^ (h[2] * 7) ^ (h[3] * 11) ^ n; }
size_t align64(size_t x) { return (x + 63) & ~(size_t)63; }
struct Section { const void *p; size_t bytes; };
std::vector<Section> sections_of(const ProvingKeyHost &pk) {
  std::vector<Section> s;
  s.push_back({&pk.alpha_g1, 64});
  s.push_back({&pk.beta_g1, 64});
  s.push_back({&pk.delta_g1, 64});
  s.push_back({&pk.beta_g2, 128});
  s.push_back({&pk.delta_g2, 128});
  s.push_back({pk.A.data(), pk.A.size() * 64});
  s.push_back({pk.B_idx.data(), pk.B_idx.size() * 4});
  s.push_back({pk.B_g1.data(), pk.B_g1.size() * 64});
  s.push_back({pk.B_g2.data(), pk.B_g2.size() * 128});
  s.push_back({pk.H_lagrange.data(), pk.H_lagrange.size() * 64}); s.push_back({pk.L_star.data(), pk.L_star.size() * 64});
  for (int m = 0; m < 3; m++) {
    s.push_back({pk.cs.rowptr[m].data(), pk.cs.rowptr[m].size() * 4});
    s.push_back({pk.cs.col[m].data(), pk.cs.col[m].size() * 4});
    s.push_back({pk.cs.coeff[m].data(), pk.cs.coeff[m].size() * 32});
  }
  return s; }
}  // namespace
static int env_int_early(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
bool key_stamp_of(const std::string &path, KeyStamp &out) {
  struct stat st;
  if (stat(path.c_str(), &st)) return false;
  out.size = st.st_size;
  out.mtime_s = st.st_mtim.tv_sec;
  out.mtime_ns = st.st_mtim.tv_nsec;
  return true;
}
std::string key_container_path(const std::string &pk_path) {
  const char *on = getenv("ZK_KEY_CACHE"); if (on && atoi(on) == 0) return "";
  // the container holds the TRANSFORMED queries: a run that switches a transform off works from the text key
  if (env_int_early("ZK_H_LAGRANGE", 1) == 0 || env_int_early("ZK_FOLD_C", 1) == 0) return "";
  const char *dir = getenv("ZK_KEY_CACHE_DIR"); if (!dir || !*dir) return pk_path + ".gpucache";
  // one file per ABSOLUTE key path: the name carries a 64-bit hash of it (flattening '/' to '_' let /a/b_c/k and /a/b/c_k share a container)
  char abs[PATH_MAX];
  std::string full = realpath(pk_path.c_str(), abs) ? std::string(abs) : pk_path;
  uint64_t hsh = 0xcbf29ce484222325ull;
  for (unsigned char ch : full) {
    hsh ^= ch;
    hsh *= 0x100000001b3ull;
  }
  std::string base = full.substr(full.find_last_of('/') == std::string::npos ? 0 : full.find_last_of('/') + 1);
  char hex[17];
  snprintf(hex, sizeof hex, "%016llx", (unsigned long long)hsh);
  return std::string(dir) + "/" + base + "." + hex + ".gpucache"; }
void save_key_container(const std::string &path, const ProvingKeyHost &pk, const KeyStamp &src) {
  if (pk.H_lagrange.empty() || pk.L_star.empty()) throw std::runtime_error("key container: the key has not been transformed yet");
  ContainerHeader h;
  memset(&h, 0, sizeof h);
  memcpy(h.magic, CONTAINER_MAGIC, 8);
  h.version = 1;
  h.flags = 3;
  h.src_size = src.size;
  h.src_mtime_s = src.mtime_s;
  h.src_mtime_ns = src.mtime_ns;
  h.n_inputs = pk.cs.n_inputs;
  h.n_vars = pk.cs.n_vars;
  h.n_cons = pk.cs.n_cons;
  h.m = pk.H_lagrange.size();
  h.nA = pk.A.size();
  h.nB = pk.B_idx.size();
  h.nH = pk.H_lagrange.size();
  h.nL = pk.L_star.size();
  for (int m = 0; m < 3; m++) h.nnz[m] = pk.cs.col[m].size();
  std::vector<Section> secs = sections_of(pk); size_t total = 0; for (auto &s : secs) total += align64(s.bytes);
  std::vector<uint8_t> buf(total, 0); size_t off = 0; for (auto &s : secs) { if (s.bytes) memcpy(buf.data() + off, s.p, s.bytes); off += align64(s.bytes); }
  h.payload_bytes = total; h.checksum = checksum64(buf.data(), total);
  // readable by the owner only: the payload is trusted as far as the checks of load_key_container go
  const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
  const int wfd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
  FILE *f = wfd < 0 ? nullptr : fdopen(wfd, "wb");
  if (!f) {
    if (wfd >= 0) close(wfd);
    throw std::runtime_error("key container: cannot write " + tmp);
  }
  bool ok = fwrite(&h, 1, sizeof h, f) == sizeof h && fwrite(buf.data(), 1, total, f) == total; ok = fclose(f) == 0 && ok;
  if (!ok || rename(tmp.c_str(), path.c_str())) { remove(tmp.c_str()); throw std::runtime_error("key container: cannot write " + path); } }
bool load_key_container(const std::string &path, const KeyStamp &src, ProvingKeyHost &pk) {
  int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) || (size_t)st.st_size < sizeof(ContainerHeader)) {
    close(fd);
    return false;
  }
  const size_t len = (size_t)st.st_size; void *map = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0); close(fd); if (map == MAP_FAILED) return false;
  struct Unmap { void *p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, len};
  const ContainerHeader &h = *(const ContainerHeader *)map; const uint8_t *pay = (const uint8_t *)map + sizeof(ContainerHeader);
  if (memcmp(h.magic, CONTAINER_MAGIC, 8) || h.version != 1 || h.flags != 3 || h.src_size != src.size || h.src_mtime_s != src.mtime_s ||
      h.src_mtime_ns != src.mtime_ns) return false;
  if (h.payload_bytes != len - sizeof(ContainerHeader) || checksum64(pay, h.payload_bytes) != h.checksum) return false;
  if (h.nA != h.n_vars + 1 || h.nL != h.n_vars + 1 || h.nH != h.m || h.n_inputs > h.n_vars || h.nB > h.nA) return false;
  ProvingKeyHost k;
  k.A.resize(h.nA);
  k.B_idx.resize(h.nB);
  k.B_g1.resize(h.nB);
  k.B_g2.resize(h.nB);
  k.H_lagrange.resize(h.nH);
  k.L_star.resize(h.nL);
  k.cs.n_inputs = h.n_inputs;
  k.cs.n_vars = h.n_vars;
  k.cs.n_cons = h.n_cons;
  for (int m = 0; m < 3; m++) { k.cs.rowptr[m].resize(h.n_cons + 1); k.cs.col[m].resize(h.nnz[m]); k.cs.coeff[m].resize(h.nnz[m]); }
  std::vector<Section> secs = sections_of(k); size_t total = 0; for (auto &s : secs) total += align64(s.bytes); if (total != h.payload_bytes) return false;
  size_t off = 0; for (auto &s : secs) { if (s.bytes) memcpy(const_cast<void *>(s.p), pay + off, s.bytes); off += align64(s.bytes); }
  // the same range and monotonicity checks as the text loader: a container with a valid checksum but indices out of range would make k_r1cs_rows_all / the
  // B-query gather read out of bounds on the device
  for (int m = 0; m < 3; m++) { if (k.cs.rowptr[m][0] != 0 || k.cs.rowptr[m][h.n_cons] != h.nnz[m]) return false;
    for (size_t r = 0; r < h.n_cons; r++) if (k.cs.rowptr[m][r] > k.cs.rowptr[m][r + 1]) return false;
    for (uint32_t cidx : k.cs.col[m]) if (cidx > h.n_vars) return false; }
  for (size_t i = 0; i < k.B_idx.size(); i++) if (k.B_idx[i] >= h.nA || (i && k.B_idx[i] <= k.B_idx[i - 1])) return false;
  pk = std::move(k); return true; }
ProvingKeyHost load_proving_key_fast(const std::string &pk_path, bool &from_container) {
  from_container = false; KeyStamp st; std::string cp = key_container_path(pk_path); ProvingKeyHost pk;
  if (!cp.empty() && key_stamp_of(pk_path, st) && load_key_container(cp, st, pk)) { from_container = true; return pk; }
  return load_proving_key(pk_path); }

// ======================================================================================================================
// generator
// ======================================================================================================================
static void urandom(void *p, size_t n) {   // the kernel's CSPRNG through getrandom(2): no file descriptor, no open() per proof
  uint8_t *b = (uint8_t *)p;
  while (n) {
    ssize_t k = getrandom(b, n, 0);
    if (k < 0) {
      if (errno == EINTR) continue;
      throw std::runtime_error("getrandom failed");
    }
    b += k;
    n -= (size_t)k;
  }
}
// uniform in [0, r) by rejection (bigint.tcc:167-179 / fp.tcc:695-721)
static HFr random_fr() {
  for (;;) {
    HFr v;
    urandom(v.l, 32);
    v.l[3] &= (1ull << 62) - 1;
    if (!HFr::geq_mod(v.l)) return v.to_mont();
  }
}
static uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
ToxicWaste ToxicWaste::random() { ToxicWaste t; HFr *f = &t.t; for (int i = 0; i < 7; i++) { do f[i] = random_fr(); while (f[i].is_zero()); } return t; }
ToxicWaste ToxicWaste::from_seed(uint64_t seed) {
  ToxicWaste t;
  HFr *f = &t.t;
  for (int i = 0; i < 7; i++) {
    HFr v;
    for (int k = 0; k < 4; k++) v.l[k] = splitmix(seed);
    v.l[3] &= (1ull << 61) - 1;
    f[i] = v.to_mont();
  }
  return t;
}

static size_t ceil_log2(size_t n) { size_t r = ((n & (n - 1)) == 0 ? 0 : 1); while (n > 1) { n >>= 1; r++; } return r; }
static HFr root_of_unity(size_t n) { HFr w; memcpy(w.l, FR_ROOT_OF_UNITY_2_28, 32); for (size_t i = 28; i > ceil_log2(n); --i) w = w.sqr(); return w; }
static void batch_inverse(std::vector<HFr> &v) {
  std::vector<HFr> pre(v.size());
  HFr acc = HFr::one();
  for (size_t i = 0; i < v.size(); i++) {
    pre[i] = acc;
    acc = acc * v[i];
  }
  HFr ai = acc.inv();
  for (size_t i = v.size(); i-- > 0;) {
    HFr t = ai * pre[i];
    ai = ai * v[i];
    v[i] = t;
  }
}
// L_i(t) on a power-of-two domain (basic_radix2_domain_aux.tcc:182-236)
static std::vector<HFr> radix2_lagrange(size_t m, const HFr &t) {
  std::vector<HFr> u(m, HFr::zero()); if (m == 1) { u[0] = HFr::one(); return u; } HFr w = root_of_unity(m), tm = t.pow_u64(m), one = HFr::one();
  if (tm == one) { HFr wi = one; for (size_t i = 0; i < m; i++) { if (wi == t) { u[i] = one; return u; } wi = wi * w; } }
  HFr Z = tm - one, l = Z * HFr::from_u64(m).inv(), r = one;
  std::vector<HFr> den(m);
  for (size_t i = 0; i < m; i++) {
    den[i] = t - r;
    r = r * w;
  }
  batch_inverse(den);
  for (size_t i = 0; i < m; i++) { u[i] = l * den[i]; l = l * w; } return u; }
struct DomainShape { size_t m; bool step; size_t B, S; };
static DomainShape domain_shape(size_t min_size) {
  DomainShape d{0, false, 0, 0};
  size_t lg = ceil_log2(min_size);
  if (min_size == ((size_t)1 << lg)) {
    d.m = min_size;
    return d;
  }
  size_t big = (size_t)1 << (lg - 1), small = min_size - big, rs = (size_t)1 << ceil_log2(small);
  d.m = small == rs ? min_size : big + rs;
  if (d.m != ((size_t)1 << ceil_log2(d.m))) {
    d.step = true;
    d.B = (size_t)1 << (ceil_log2(d.m) - 1);
    d.S = d.m - d.B;
  }
  return d;
}
static size_t domain_size_for(size_t min_size) { return domain_shape(min_size).m; }
// all Lagrange polynomials at t and Z(t)  (basic_radix2_domain.tcc:90-101; step_radix2_domain.tcc:169-215)
static std::vector<HFr> domain_lagrange(const DomainShape &d, const HFr &t, HFr &Zt) { HFr one = HFr::one();
  if (!d.step) { Zt = t.pow_u64(d.m) - one; return radix2_lagrange(d.m, t); }
  HFr w = root_of_unity((size_t)1 << ceil_log2(d.m)), wb = w.sqr(), wS = w.pow_u64(d.S);
  std::vector<HFr> ib = radix2_lagrange(d.B, t), is = radix2_lagrange(d.S, t * w.inv()), u(d.m);
  HFr L0 = t.pow_u64(d.S) - wS, bwS = wb.pow_u64(d.S), elt = one;
  std::vector<HFr> den(d.B);
  for (size_t i = 0; i < d.B; i++) {
    den[i] = elt - wS;
    elt = elt * bwS;
  }
  batch_inverse(den);
  for (size_t i = 0; i < d.B; i++) u[i] = ib[i] * L0 * den[i];
  HFr L1 = (t.pow_u64(d.B) - one) * (w.pow_u64(d.B) - one).inv(); for (size_t i = 0; i < d.S; i++) u[d.B + i] = L1 * is[i];
  Zt = (t.pow_u64(d.B) - one) * (t.pow_u64(d.S) - wS); return u; }

static HG2 default_g2_generator() {
  HFq v[4];
  for (int k = 0; k < 4; k++) memcpy(v[k].l, G2_GENERATOR[k], 32);
  return HG2{HFq2{v[0], v[1]}, HFq2{v[2], v[3]}, HFq2::one()};
}
static R1csHost swap_ab_if_beneficial(const R1csHost &in) {   // r1cs.tcc:182-231
  std::vector<uint8_t> ta(in.n_vars + 1, 0), tb(in.n_vars + 1, 0);
  for (uint32_t c : in.col[0]) ta[c] = 1;
  for (uint32_t c : in.col[1]) tb[c] = 1;
  size_t na = 0, nb = 0;
  for (size_t i = 0; i <= in.n_vars; i++) {
    na += ta[i];
    nb += tb[i];
  }
  R1csHost out = in;
  if (nb > na) {
    std::swap(out.rowptr[0], out.rowptr[1]);
    std::swap(out.col[0], out.col[1]);
    std::swap(out.coeff[0], out.coeff[1]);
  }
  return out;
}

void generate_keys(const R1csHost &cs_in, const ToxicWaste &tw, ProvingKeyHost &pk, VerifyingKeyHost &vk) {
  pk.cs = swap_ab_if_beneficial(cs_in);
  const R1csHost &cs = pk.cs;
  size_t nv = cs.n_vars, ni = cs.n_inputs, nc = cs.n_cons;
  DomainShape d = domain_shape(nc + ni + 1);
  size_t m = d.m;
  HFr Zt; std::vector<HFr> u = domain_lagrange(d, tw.t, Zt); std::vector<HFr> M[3]; for (int k = 0; k < 3; k++) M[k].assign(nv + 1, HFr::zero());
  // r1cs_to_qap.tcc:128-131
  for (size_t i = 0; i <= ni; i++) M[0][i] = u[nc + i];
  for (int k = 0; k < 3; k++) for (size_t i = 0; i < nc; i++) for (uint32_t e = cs.rowptr[k][i]; e < cs.rowptr[k][i + 1];
      e++) M[k][cs.col[k][e]] = M[k][cs.col[k][e]] + u[i] * fr_of(cs.coeff[k][e]).to_mont();
  HFr gi = tw.gamma.inv(), di = tw.delta.inv();
  auto canon = [](const HFr &x) { return fe_of_r(x.from_mont()); };
  std::vector<Fe32> sA(nv + 1), sB, sH(m - 1), sL(nv - ni), sIC(ni + 1);
  for (size_t i = 0; i <= nv; i++) sA[i] = canon(M[0][i]);
  // kc_multiexp.tcc:105-112
  pk.B_idx.clear();
  for (size_t i = 0; i <= nv; i++) if (!M[1][i].is_zero()) {
    pk.B_idx.push_back((uint32_t)i);
    sB.push_back(canon(M[1][i]));
  }
  // :330 batch_exp_with_coeff(Zt/delta, Ht), Ht truncated by 2 (:281)
  {
    HFr x = Zt * di;
    for (size_t i = 0; i + 1 < m; i++) {
      sH[i] = canon(x);
      x = x * tw.t;
    }
  }
  for (size_t i = 0; i < nv - ni; i++) { size_t j = ni + 1 + i; sL[i] = canon((tw.beta * M[0][j] + tw.alpha * M[1][j] + M[2][j]) * di); }          // :264-273
  for (size_t i = 0; i <= ni; i++) sIC[i] = canon((tw.beta * M[0][i] + tw.alpha * M[1][i] + M[2][i]) * gi);                                        // :253-260
  // G1 generator (1, 2)
  HG1 g1{HFq::from_u64(1), HFq::from_u64(2), HFq::one()};
  HG2 g2 = default_g2_generator();
  // random generators (:297,:307)
  {
    HFr k1 = tw.g1_scalar.from_mont(), k2 = tw.g2_scalar.from_mont();
    g1 = g1.mul(k1.l);
    g2 = g2.mul(k2.l);
  }
  auto mul1 = [&](const HFr &k) {
    HFr c = k.from_mont();
    return raw_of(g1.mul(c.l));
  };
  auto mul2 = [&](const HFr &k) {
    HFr c = k.from_mont();
    return raw_of(g2.mul(c.l));
  };
  pk.alpha_g1 = mul1(tw.alpha); pk.beta_g1 = mul1(tw.beta); pk.beta_g2 = mul2(tw.beta); pk.delta_g1 = mul1(tw.delta); pk.delta_g2 = mul2(tw.delta);
  pk.A.resize(nv + 1); fixed_base_mul_g1(g1, sA.data(), nv + 1, pk.A.data());
  pk.B_g1.resize(sB.size());
  pk.B_g2.resize(sB.size());
  fixed_base_mul_g1(g1, sB.data(), sB.size(), pk.B_g1.data());
  fixed_base_mul_g2(g2, sB.data(), sB.size(), pk.B_g2.data());
  pk.H.resize(m - 1); fixed_base_mul_g1(g1, sH.data(), m - 1, pk.H.data()); pk.L.resize(nv - ni); fixed_base_mul_g1(g1, sL.data(), nv - ni, pk.L.data());
  vk.IC.resize(ni + 1); fixed_base_mul_g1(g1, sIC.data(), ni + 1, vk.IC.data()); vk.gamma_g2 = mul2(tw.gamma); vk.delta_g2 = pk.delta_g2;
  // :355
  vk.alpha_g1_beta_g2 = reduced_pairing(fq_of(pk.alpha_g1.x), fq_of(pk.alpha_g1.y), fq2_of(pk.beta_g2.x0, pk.beta_g2.x1), fq2_of(pk.beta_g2.y0, pk.beta_g2.y1));
}

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
  if (!p.cs->check_result()) return false;
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
  Impl &p = *impl; LaneScope lane_scope(p.lane); run_device(p); for (int j = 0; j < 4; j++) p.settle(j); gpu_sync(); if (!p.cs->check_result()) return false;
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

// ======================================================================================================================
// verifier and proof encoding
// ======================================================================================================================
// one-off use; callers that verify more than once keep the prepared key
bool verify_proof(const VerifyingKeyHost &vk, const Fe32 *inputs, size_t n_inputs, const Proof &proof) {
  return verify_proof(*prepare_verifying_key(vk), inputs, n_inputs, proof);
}

std::shared_ptr<PreparedVerifyingKey> prepare_verifying_key(const VerifyingKeyHost &vk) {
  auto p = std::make_shared<PreparedVerifyingKey>(); p->vk = vk;
  p->gamma = precompute_g2(fq2_of(vk.gamma_g2.x0, vk.gamma_g2.x1), fq2_of(vk.gamma_g2.y0, vk.gamma_g2.y1));
  p->delta = precompute_g2(fq2_of(vk.delta_g2.x0, vk.delta_g2.x1), fq2_of(vk.delta_g2.y0, vk.delta_g2.y1));
  const size_t ni = vk.IC.size() ? vk.IC.size() - 1 : 0; p->ic_x.assign(ni * 32 * 255, HFq::zero()); p->ic_y.assign(ni * 32 * 255, HFq::zero());
  for (size_t j = 0; j < ni; j++) {
    HG1 wbase = is_zero_raw(&vk.IC[j + 1], sizeof(G1AffineRaw)) ? HG1::inf() : g1_of(vk.IC[j + 1]);
    std::vector<HG1> pts(32 * 255);
    for (int w = 0; w < 32; w++) { HG1 acc = wbase; for (int d = 1; d <= 255; d++) { pts[w * 255 + d - 1] = acc; acc = acc.add(wbase); } wbase = acc; }
    // one inversion for the whole table
    std::vector<HFq> pre(pts.size());
    HFq run = HFq::one();
    for (size_t k = 0; k < pts.size(); k++) {
      pre[k] = run;
      if (!pts[k].is_inf()) run = run * pts[k].Z;
    }
    HFq inv = run.inv();
    for (size_t k = pts.size(); k-- > 0;) {
      if (pts[k].is_inf()) continue;
      HFq zi = inv * pre[k];
      inv = inv * pts[k].Z;
      HFq z2 = zi.sqr();
      p->ic_x[j * 32 * 255 + k] = pts[k].X * z2;
      p->ic_y[j * 32 * 255 + k] = pts[k].Y * z2 * zi;
    }
  }
  return p;
}
bool verify_proof(const PreparedVerifyingKey &pvk, const Fe32 *inputs, size_t n_inputs, const Proof &proof) {
  const VerifyingKeyHost &vk = pvk.vk; if (vk.IC.size() != n_inputs + 1) return false;                                     // strong IC (:584-590)
  HG1 acc = g1_of(vk.IC[0]);
  for (size_t j = 0; j < n_inputs; j++) { const uint8_t *b = reinterpret_cast<const uint8_t *>(&inputs[j]);
    for (int w = 0; w < 32; w++) if (b[w]) {
      const size_t k = j * 32 * 255 + (size_t)w * 255 + b[w] - 1;
      if (!(pvk.ic_x[k].is_zero() && pvk.ic_y[k].is_zero())) acc = acc.add(HG1::from_affine(pvk.ic_x[k], pvk.ic_y[k]));
    }
  }
  HFq ax = fq_of(proof.A.x), ay = fq_of(proof.A.y), cx = fq_of(proof.C.x), cy = fq_of(proof.C.y);
  HFq2 bx = fq2_of(proof.B.x0, proof.B.x1), by = fq2_of(proof.B.y0, proof.B.y1);
  if (is_zero_raw(&proof.A, sizeof proof.A) || is_zero_raw(&proof.B, sizeof proof.B) || is_zero_raw(&proof.C, sizeof proof.C)) return false;
  if (!(g1_on_curve(ax, ay) && g2_on_curve(bx, by) && g1_on_curve(cx, cy))) return false;                                    // is_well_formed: on-curve only
  HFq accx, accy; acc.to_affine(accx, accy);
  HFq12 q1 = miller_loop(ax, ay, precompute_g2(bx, by)), q2 = acc.is_inf() ? HFq12::one() : miller_loop(accx, accy, pvk.gamma), q3 = miller_loop(cx, cy,
      pvk.delta);
  return final_exponentiation(q1 * (q2 * q3).conj()) == vk.alpha_g1_beta_g2;                                               // :556-560
}

// The decision of verify_proof() taken by the GPU verifier's SCHEDULE (verify_sched.hpp) interpreted on the host: what kernel K9 computes, without a GPU. Test
// entry (zkgpu_test_verify_schedule): the schedule is checked against the host verifier and the oracle on the CPU before any device runs it. stats: rounds,
// slots, products, linear operations, constants, then the WAVES of products / eight-lane sums / one-lane sums.
bool verify_by_schedule_on_host(const PreparedVerifyingKey &pvk, const Fe32 *inputs, size_t n_inputs, const Proof &proof, uint32_t stats[8]) {
  const VerifyingKeyHost &vk = pvk.vk; vsched::Schedule S = vsched::build(vk.alpha_g1_beta_g2, pvk.gamma, pvk.delta);
  if (stats) {
    stats[0] = S.n_rounds;
    stats[1] = S.n_slots;
    stats[2] = S.n_mul;
    stats[3] = S.n_lin;
    stats[4] = (uint32_t)S.consts.size();
    stats[5] = S.waves_of_kind[vsched::K_MUL];
    stats[6] = S.waves_of_kind[vsched::K_LIN8];
    stats[7] = S.waves_of_kind[vsched::K_LIN1];
  }
  if (vk.IC.size() != n_inputs + 1) return false;
  HG1 acc = g1_of(vk.IC[0]);
  for (size_t j = 0; j < n_inputs; j++) { const uint8_t *b = reinterpret_cast<const uint8_t *>(&inputs[j]);
    for (int w = 0; w < 32; w++) if (b[w]) {
      const size_t k = j * 32 * 255 + (size_t)w * 255 + b[w] - 1;
      if (!(pvk.ic_x[k].is_zero() && pvk.ic_y[k].is_zero())) acc = acc.add(HG1::from_affine(pvk.ic_x[k], pvk.ic_y[k]));
    }
  }
  if (is_zero_raw(&proof.A, sizeof proof.A) || is_zero_raw(&proof.B, sizeof proof.B) || is_zero_raw(&proof.C, sizeof proof.C)) return false;
  // (the kernel hands such a proof back to the host verifier: the gamma pairing is the identity then)
  if (acc.is_inf()) return verify_proof(pvk, inputs, n_inputs, proof);
  HFq accx, accy; acc.to_affine(accx, accy); HFq in[vsched::N_INPUTS];
  in[vsched::IN_AX] = fq_of(proof.A.x);
  in[vsched::IN_AY] = fq_of(proof.A.y);
  in[vsched::IN_BX0] = fq_of(proof.B.x0);
  in[vsched::IN_BX1] = fq_of(proof.B.x1);
  in[vsched::IN_BY0] = fq_of(proof.B.y0);
  in[vsched::IN_BY1] = fq_of(proof.B.y1);
  in[vsched::IN_CX] = fq_of(proof.C.x);
  in[vsched::IN_CY] = fq_of(proof.C.y);
  // (the accumulation kernel hands the point over as (x w, -y w, w) with w = ZZ ZZZ of its sum: a non-trivial w here too, so that the CPU tests cover the scaling)
  const HFq w = HFq::from_u64(0x9e3779b97f4a7c15ull ^ ((uint64_t)accx.l[0] | (uint64_t)accx.l[1] << 32)) + HFq::one();
  in[vsched::IN_NACCX] = accx * w;
  in[vsched::IN_NACCY] = (accy * w).neg();
  in[vsched::IN_NACCW] = w;
  std::vector<HFq> out = vsched::simulate(S, in); bool ok = true;
  // the GT comparison, then the on-curve residues: all zero; the norm of the Miller value: not zero
  for (int k = 0; k < vsched::N_OUT; k++) ok = ok && (k == vsched::OUT_NONZERO ? !out[k].is_zero() : out[k].is_zero());
  // ... and the same program on the kernel's own limb arithmetic (every bound asserted on the way): value by value the same verdicts
  uint32_t words[vsched::N_INPUTS][8]; for (int i = 0; i < vsched::N_INPUTS; i++) memcpy(words[i], in[i].l, 32);
  std::vector<bool> zero29 = vsched::simulate29(S, words);
  for (int k = 0; k < vsched::N_OUT;
      k++) if (zero29[k] != out[k].is_zero()) throw std::runtime_error("verify schedule: the 29-bit model and the field model disagree on output " +
      std::to_string(k));
  return ok;
}
// Kernel K9 against the host model of its own arithmetic, value by value: one proof runs through the device kernels with the LDS values written out after every
// `every`-th round; vsched::simulate29 is fed the same inputs (the accumulation kernel's record included) and must hold the same limbs in every slot that the
// schedule has written by then.  Returns -1 if every dump agrees, otherwise the first round whose dump differs (slot in *bad_slot); *device_ok = the kernel's verdict.
long verify_schedule_trace_on_device(BatchVerifier &bv, const PreparedVerifyingKey &pvk, const Fe32 *inputs, size_t n_inputs, const Proof &proof, uint32_t every,
    uint32_t *bad_slot, uint8_t *device_ok) {
  const VerifyingKeyHost &vk = pvk.vk; if (vk.IC.size() != n_inputs + 1 || bv.num_inputs() != n_inputs) throw std::runtime_error("verify trace: input count");
  vsched::Schedule S = vsched::build(vk.alpha_g1_beta_g2, pvk.gamma, pvk.delta);
  std::vector<uint32_t> values; uint8_t nacc[96]; const uint8_t ok = bv.trace(&proof, inputs, every, values, nacc); if (device_ok) *device_ok = ok;
  uint32_t words[vsched::N_INPUTS][8]; const Fe32 *pc = reinterpret_cast<const Fe32 *>(&proof);           // A.x A.y | B.x.c0 B.x.c1 B.y.c0 B.y.c1 | C.x C.y
  for (int k = 0; k < 8; k++) memcpy(words[k], &pc[k], 32);
  for (int k = 0; k < 3; k++) memcpy(words[vsched::IN_NACCX + k], nacc + 32 * k, 32);
  const size_t stride = (size_t)S.n_slots * l29::STRIDE; long first_bad = -1; uint32_t slot_bad = 0; std::vector<char> written(S.n_slots, 0);
  for (int k = 0; k < vsched::N_INPUTS; k++) written[k] = 1;
  vsched::simulate29(S, words, [&](uint32_t r, const std::vector<std::array<uint32_t, 9>> &slots) {
    vsched::for_each_op(S, r, [&](uint32_t, uint32_t, const uint32_t *w) { written[w[0] & 0x7fffu] = 1; });
    if (first_bad >= 0 || (r + 1) % every != 0 || r >= S.n_rounds) return; const uint32_t *dv = &values[(size_t)(r / every) * stride];
    for (uint32_t sl = 0; sl < S.n_slots && first_bad < 0; sl++) if (written[sl] && memcmp(dv + (size_t)sl * l29::STRIDE, slots[sl].data(), 36) != 0) { first_bad = (long)r; slot_bad = sl; } });
  if (bad_slot) *bad_slot = slot_bad; return first_bad;
}
std::unique_ptr<BatchVerifier> make_batch_verifier(const VerifyingKeyHost &vk) {
  return std::unique_ptr<BatchVerifier>(new BatchVerifier(vk.alpha_g1_beta_g2, vk.gamma_g2, vk.delta_g2, vk.IC.data(), vk.IC.size()));
}

static void put_hex_fq(std::string &o, const Fe32 &mont) {
  HFq c = fq_of(mont).from_mont();
  static const char *d = "0123456789abcdef";
  for (int i = 3; i >= 0; i--) for (int k = 15; k >= 0; k--) o.push_back(d[(c.l[i] >> (4 * k)) & 15]);
}
std::string proof_to_hex(const Proof &p) {
  std::string o;
  o.reserve(512);
  put_hex_fq(o, p.A.x);
  put_hex_fq(o, p.A.y);
  put_hex_fq(o, p.B.x1);
  put_hex_fq(o, p.B.x0);
  put_hex_fq(o, p.B.y1);
  put_hex_fq(o, p.B.y0);
  put_hex_fq(o, p.C.x);
  put_hex_fq(o, p.C.y);
  return o;
}
bool proof_from_hex(const char *hex, Proof &p) {
  static const bool strict = [] { const char *e = getenv("ZK_STRICT_PROOF_ENCODING"); return e && *e && *e != '0'; }();
  Fe32 v[8];
  for (int k = 0; k < 8; k++) {
    HFq c = HFq::zero();
    for (int i = 0; i < 64; i++) {
      char ch = hex[64 * k + i];
      int dgt = ch >= '0' && ch <= '9' ? ch - '0' : ch >= 'a' && ch <= 'f' ? ch - 'a' + 10 : -1;
      if (dgt < 0) return false;
      c.l[(63 - i) / 16] |= (uint64_t)dgt << (4 * ((63 - i) % 16));
    }
    // Any 256-bit value is a coordinate: the reference builds the field element with Fp_model(const bigint&) (sendcgo.cpp:422-446 -> fp.tcc:190-194), one
    // Montgomery product with R^2, which leaves value mod q.  to_mont() is that product (a < 2^256, R^2 < q: the sum stays below 2q, one subtraction), so a
    // coordinate c and c + kq are the same proof here as they are there — consensus needs the same accept set, not a stricter one (ZK_STRICT_PROOF_ENCODING=1
    // restores the rejection for deployments that want canonical encodings only; INTEGRATION.md "Not verbatim").
    if (strict && HFq::geq_mod(c.l)) return false;
    v[k] = fe_of(c.to_mont());
  }
  p.A = {v[0], v[1]}; p.B = {v[3], v[2], v[5], v[4]}; p.C = {v[6], v[7]}; return true;
}
Proof default_proof() {
  Proof p;
  HG1 g{HFq::from_u64(1), HFq::from_u64(2), HFq::one()};
  p.A = raw_of(g);
  p.C = p.A;
  p.B = raw_of(default_g2_generator());
  return p;
}

}  // namespace zk
