// Groth16 host code, part 1 of 4: key files (the reference's text format, the hybrid writer) and the fast key container.
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

}  // namespace zk
