// TEST INFRASTRUCTURE — not part of the product.
//
// Driver for the REAL reference (libff / libfqfft / libsnark as vendored by BlockMaze under
// /root/reference/libsnark-vnt/depends/libsnark).  It is compiled by oracle/Makefile (`make ref`) against the
// reference sources where they lie; only the resulting binary (oracle/_ref/ref_harness, git-ignored) travels.
// Every number it prints is computed by reference code; this file only feeds inputs and prints outputs.
//
// It takes the place the reference's own drivers (libsnark-vnt/src/X/main.cpp, getpvk.cpp, Xcgo.cpp) would have
// had: those need boost (absent from the image), so they are not built.  What they do around the libsnark calls is
// restated here where needed, with citations:
//   * proof -> 512 hex chars            : src/send/sendcgo.cpp:113-188 (string_proof_as_hex)
//   * 512 hex chars -> proof (Z stays 1): src/send/sendcgo.cpp:388-448
//   * prover tail with explicit r,s     : r1cs_gg_ppzksnark.tcc:417-419,487-495 (the stock prover draws r,s from
//                                         std::random_device, so byte parity needs them injected)
//   * key files                         : src/send/sendcgo.cpp:50-62 (writeToFile = operator<< into a file)
//
// Modes (all files little-endian; field elements = 32-byte LE canonical integers unless noted):
//   vectors <out.txt>                                   field / curve / domain / MSM / pairing known-answer vectors
//   sha256gadget <out_r1cs.bin> <out_wit.bin> <seed>    libsnark sha256_compression_function_gadget R1CS + witness
//   merklegadget <depth> <out_r1cs.bin> <out_wit.bin> <seed>
//   unpacker <nbits> <seed> <out_r1cs.bin> <out_wit.bin>  libsnark's multipacking_gadget as the four circuits build their public-input unpacker (e.g. send/circuit/gadget.tcc:87-106,198)
//   lesscmp <value_old> <value_s> <out_r1cs.bin> <out_wit.bin>   BlockMaze's less_comparison_gadget block (send/circuit/comparison.tcc) R1CS + witness
//   hashblock cmts|prf|crh <seed> <out_r1cs.bin> <out_wit.bin>   the CMTS / PRF / CRH blocks composed like commitment.tcc:100-320
//   cmta <seed> <out_r1cs.bin> <out_wit.bin>            two chained compression gadgets + hard-wired padding, composed like sha256_CMTA_gadget (commitment.tcc:12-110)
//   notehashes <seed> <count>                           Note::cm / NoteS::cm / Compute_PRF / Compute_CRH on seeded hex inputs (send/Note.h, util.h)
//   e2e <r1cs.bin> <wit.bin> <r_hex> <s_hex> <outdir>   is_satisfied, generator, write pk.txt/vk.txt, prove(r,s),
//                                                       verify; prints "proof <hex>"
//   prove <pk.txt> <wit.bin> <n_inputs> <r_hex> <s_hex> load pk with the reference operator>>, prove(r,s)
//   verify <vk.txt> <proof_hex> <n_inputs> <in0_dec> …  reference verifier_strong_IC -> "verify 0|1"
//   verifymany <vk.txt> <cases.txt>                     the same for every line "<512 hex> <n_inputs> <in0_dec> …" of a file (one key load): "v <line> 0|1"
//   hexblobs <strings.txt>                              every line = one C string, hex-escaped: uint256S / uint160S (send/uint256.h:222-248) and the hashes
//                                                       Compute_PRF(x, 0), Compute_CRH(x, 0) of the parsed blobs -> "blob <line> <u256> <u160> <prf> <crh>"
//   bench_prover <r1cs.bin> <wit.bin> [threads]         time the stock r1cs_gg_ppzksnark_prover on a key of the
//                                                       right shape (synthetic points), prints seconds per phase
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <iostream>
#include <string>
#include <vector>
#include <chrono>
#include <unistd.h>
#include <fcntl.h>
#include <sys/wait.h>
#ifdef MULTICORE
#include <omp.h>
#endif

#include "libsnark/zk_proof_systems/ppzksnark/r1cs_gg_ppzksnark/r1cs_gg_ppzksnark.hpp"
#include "libsnark/common/default_types/r1cs_gg_ppzksnark_pp.hpp"
#include <libsnark/gadgetlib1/gadgets/hashes/sha256/sha256_gadget.hpp>
#include <libsnark/gadgetlib1/gadgets/merkle_tree/merkle_tree_check_read_gadget.hpp>
#include <libfqfft/evaluation_domain/get_evaluation_domain.hpp>
#include <libff/algebra/scalar_multiplication/multiexp.hpp>
// BlockMaze's own sources that need no boost (the Makefile adds -I libsnark-vnt/src/send): the host note hashing (Note.h -> deps/sha256.h, uint256.h, util.h) and,
// further down, the comparison gadget (circuit/comparison.tcc; redeem/circuit/comparison.tcc is the same file).  circuit/utils.tcc — and with it note.tcc,
// commitment.tcc and gadget.tcc, which call into it — needs BOOST_FOREACH and stays unbuilt.
#include "Note.h"

using namespace libsnark;
using namespace libff;
#include "circuit/comparison.tcc"   // less_comparison_gadget (unqualified gadgetlib1 names: after the using-directives, as in send/main.cpp:17-20)
typedef alt_bn128_pp ppT;
typedef Fr<ppT> FrT;
typedef Fq<ppT> FqT;
typedef alt_bn128_Fq2 Fq2T;
typedef G1<ppT> G1T;
typedef G2<ppT> G2T;

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- SplitMix64: the PRNG shared with oracle/ and tests/ ------------------------------------------------------
struct SplitMix { uint64_t s; explicit SplitMix(uint64_t seed) : s(seed) {}
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); } };

// 253-bit value (< r and < q): limbs l0..l3 from four draws, top limb masked to 61 bits
template <typename F> static F rand_field(SplitMix &g) {
  bigint<4> b; for (int i = 0; i < 4; i++) b.data[i] = g.next(); b.data[3] &= ((1ull << 61) - 1); return F(b); }

template <typename F> static std::string hex_of(const F &x) {
  bigint<4> b = x.as_bigint(); char buf[65];
  snprintf(buf, sizeof buf, "%016lx%016lx%016lx%016lx", b.data[3], b.data[2], b.data[1], b.data[0]); return buf; }
static std::string hex2(const Fq2T &x) { return hex_of(x.c0) + "," + hex_of(x.c1); }
static std::string g1hex(G1T p) { if (p.is_zero()) return "inf"; p.to_affine_coordinates(); return hex_of(p.X) + "," + hex_of(p.Y); }
static std::string g2hex(G2T p) { if (p.is_zero()) return "inf"; p.to_affine_coordinates(); return hex2(p.X) + "," + hex2(p.Y); }

template <typename F> static F from_hex(const std::string &h) {
  bigint<4> b; std::string s = h; while (s.size() < 64) s = "0" + s;
  for (int i = 0; i < 4; i++) b.data[3 - i] = strtoull(s.substr(16 * i, 16).c_str(), nullptr, 16); return F(b); }
template <typename F> static F from_le32(const uint8_t *p) { bigint<4> b; memcpy(b.data, p, 32); return F(b); }
template <typename F> static void to_le32(const F &x, uint8_t *p) { bigint<4> b = x.as_bigint(); memcpy(p, b.data, 32); }

// ---- proof <-> hex, restating sendcgo.cpp:113-188 and :388-448 -------------------------------------------------
static std::string proof_hex(const r1cs_gg_ppzksnark_proof<ppT> &pr) {
  G1T a = pr.g_A, c = pr.g_C; G2T b = pr.g_B; a.to_affine_coordinates(); b.to_affine_coordinates(); c.to_affine_coordinates();
  return hex_of(a.X) + hex_of(a.Y) + hex_of(b.X.c1) + hex_of(b.X.c0) + hex_of(b.Y.c1) + hex_of(b.Y.c0) + hex_of(c.X) + hex_of(c.Y); }
static r1cs_gg_ppzksnark_proof<ppT> proof_from_hex(const std::string &h) {
  r1cs_gg_ppzksnark_proof<ppT> pr; // default ctor: (G1::one, G2::one, G1::one), i.e. Z = 1 everywhere
  auto f = [&](int k) { return from_hex<FqT>(h.substr(64 * k, 64)); };
  pr.g_A.X = f(0); pr.g_A.Y = f(1); pr.g_B.X.c1 = f(2); pr.g_B.X.c0 = f(3); pr.g_B.Y.c1 = f(4); pr.g_B.Y.c0 = f(5);
  pr.g_C.X = f(6); pr.g_C.Y = f(7); return pr; }

// ---- R1CS / witness exchange files ------------------------------------------------------------------------------
static const char MAGIC[8] = {'R', '1', 'C', 'S', 'B', 'M', '0', '1'};
static r1cs_constraint_system<FrT> load_r1cs(const char *path) {
  FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(2); }
  char mg[8]; uint64_t hdr[3]; if (fread(mg, 1, 8, f) != 8 || memcmp(mg, MAGIC, 8)) { fprintf(stderr, "bad magic\n"); exit(2); }
  if (fread(hdr, 8, 3, f) != 3) exit(2);
  r1cs_constraint_system<FrT> cs; cs.primary_input_size = hdr[0]; cs.auxiliary_input_size = hdr[1] - hdr[0];
  size_t nc = hdr[2]; std::vector<linear_combination<FrT>> lc[3];
  for (int m = 0; m < 3; m++) {
    uint64_t nnz; if (fread(&nnz, 8, 1, f) != 1) exit(2);
    std::vector<uint32_t> rp(nc + 1), col(nnz); std::vector<uint8_t> co(nnz * 32);
    if (fread(rp.data(), 4, nc + 1, f) != nc + 1 || fread(col.data(), 4, nnz, f) != nnz || fread(co.data(), 32, nnz, f) != nnz) exit(2);
    lc[m].resize(nc);
    for (size_t i = 0; i < nc; i++) for (uint32_t k = rp[i]; k < rp[i + 1]; k++)
      lc[m][i].add_term(variable<FrT>(col[k]), from_le32<FrT>(&co[32 * (size_t)k]));
  }
  fclose(f);
  for (size_t i = 0; i < nc; i++) cs.add_constraint(r1cs_constraint<FrT>(lc[0][i], lc[1][i], lc[2][i]));
  return cs; }
static void save_r1cs(const char *path, const r1cs_constraint_system<FrT> &cs) {
  FILE *f = fopen(path, "wb"); if (!f) { perror(path); exit(2); }
  uint64_t hdr[3] = {cs.num_inputs(), cs.num_variables(), cs.num_constraints()}; fwrite(MAGIC, 1, 8, f); fwrite(hdr, 8, 3, f);
  for (int m = 0; m < 3; m++) {
    std::vector<uint32_t> rp(1, 0), col; std::vector<uint8_t> co;
    for (auto &c : cs.constraints) { const linear_combination<FrT> &l = m == 0 ? c.a : m == 1 ? c.b : c.c;
      for (auto &t : l.terms) { col.push_back(t.index); co.resize(co.size() + 32); to_le32(t.coeff, &co[co.size() - 32]); }
      rp.push_back(col.size()); }
    uint64_t nnz = col.size(); fwrite(&nnz, 8, 1, f); fwrite(rp.data(), 4, rp.size(), f); fwrite(col.data(), 4, nnz, f); fwrite(co.data(), 32, nnz, f); }
  fclose(f); }
static std::vector<FrT> load_wit(const char *path) {
  FILE *f = fopen(path, "rb"); if (!f) { perror(path); exit(2); } uint64_t n; if (fread(&n, 8, 1, f) != 1) exit(2);
  std::vector<uint8_t> b(n * 32); if (fread(b.data(), 32, n, f) != n) exit(2); fclose(f);
  std::vector<FrT> w(n); for (size_t i = 0; i < n; i++) w[i] = from_le32<FrT>(&b[32 * i]); return w; }
static void save_wit(const char *path, const std::vector<FrT> &w) {
  FILE *f = fopen(path, "wb"); uint64_t n = w.size(); fwrite(&n, 8, 1, f); uint8_t b[32];
  for (auto &x : w) { to_le32(x, b); fwrite(b, 1, 32, f); } fclose(f); }

template <typename T> static void write_obj(const std::string &path, const T &obj) { // sendcgo.cpp:50-62
  std::stringstream ss; ss << obj; std::ofstream fh(path, std::ios::binary); fh << ss.rdbuf(); }
template <typename T> static T read_obj(const std::string &path) { // sendcgo.cpp:64-81
  std::stringstream ss; std::ifstream fh(path, std::ios::binary); if (!fh.is_open()) { perror(path.c_str()); exit(2); }
  ss << fh.rdbuf(); T obj; ss >> obj; return obj; }

// ---- prover with injected r, s: r1cs_gg_ppzksnark.tcc:391-506 with :418-419 replaced ---------------------------
static r1cs_gg_ppzksnark_proof<ppT> prove_fixed(const r1cs_gg_ppzksnark_proving_key<ppT> &pk, const std::vector<FrT> &primary,
                                                const std::vector<FrT> &aux, const FrT &r, const FrT &s, std::vector<FrT> *h_out = nullptr) {
  const qap_witness<FrT> qw = r1cs_to_qap_witness_map(pk.constraint_system, primary, aux, FrT::zero(), FrT::zero(), FrT::zero());
  if (h_out) *h_out = qw.coefficients_for_H;
  std::vector<FrT> cpa(1, FrT::one()); cpa.insert(cpa.end(), qw.coefficients_for_ABCs.begin(), qw.coefficients_for_ABCs.end());
  const size_t nv = qw.num_variables(), ni = qw.num_inputs();
  G1T eA = multi_exp_with_mixed_addition<G1T, FrT, multi_exp_method_BDLO12>(pk.A_query.begin(), pk.A_query.begin() + nv + 1, cpa.begin(), cpa.begin() + nv + 1, 1);
  knowledge_commitment<G2T, G1T> eB = kc_multi_exp_with_mixed_addition<G2T, G1T, FrT, multi_exp_method_BDLO12>(pk.B_query, 0, nv + 1, cpa.begin(), cpa.begin() + nv + 1, 1);
  G1T eH = multi_exp<G1T, FrT, multi_exp_method_BDLO12>(pk.H_query.begin(), pk.H_query.begin() + (qw.degree() - 1), qw.coefficients_for_H.begin(), qw.coefficients_for_H.begin() + (qw.degree() - 1), 1);
  G1T eL = multi_exp_with_mixed_addition<G1T, FrT, multi_exp_method_BDLO12>(pk.L_query.begin(), pk.L_query.end(), cpa.begin() + ni + 1, cpa.begin() + nv + 1, 1);
  G1T g1_A = pk.alpha_g1 + eA + r * pk.delta_g1;
  G1T g1_B = pk.beta_g1 + eB.h + s * pk.delta_g1;
  G2T g2_B = pk.beta_g2 + eB.g + s * pk.delta_g2;
  G1T g1_C = eH + eL + s * g1_A + r * g1_B - (r * s) * pk.delta_g1;
  return r1cs_gg_ppzksnark_proof<ppT>(std::move(g1_A), std::move(g2_B), std::move(g1_C)); }

// ================================================================================================================
static void emit_field_vectors(FILE *o) {
  SplitMix g(0xF1E1D5);
  for (int i = 0; i < 8; i++) {
    FrT a = rand_field<FrT>(g), b = rand_field<FrT>(g);
    fprintf(o, "fr %s %s mul=%s add=%s sub=%s inv=%s sqr=%s neg=%s\n", hex_of(a).c_str(), hex_of(b).c_str(), hex_of(a * b).c_str(),
            hex_of(a + b).c_str(), hex_of(a - b).c_str(), hex_of(a.inverse()).c_str(), hex_of(a.squared()).c_str(), hex_of(-a).c_str());
    FqT c = rand_field<FqT>(g), d = rand_field<FqT>(g); FqT c2 = c.squared();
    fprintf(o, "fq %s %s mul=%s add=%s sub=%s inv=%s sqr=%s neg=%s sqrt_of_sqr=%s\n", hex_of(c).c_str(), hex_of(d).c_str(), hex_of(c * d).c_str(),
            hex_of(c + d).c_str(), hex_of(c - d).c_str(), hex_of(c.inverse()).c_str(), hex_of(c2).c_str(), hex_of(-c).c_str(), hex_of(c2.sqrt()).c_str());
    Fq2T e(rand_field<FqT>(g), rand_field<FqT>(g)), f(rand_field<FqT>(g), rand_field<FqT>(g)); Fq2T e2 = e.squared();
    fprintf(o, "fq2 %s %s mul=%s sqr=%s inv=%s sqrt_of_sqr=%s frob=%s\n", hex2(e).c_str(), hex2(f).c_str(), hex2(e * f).c_str(), hex2(e2).c_str(),
            hex2(e.inverse()).c_str(), hex2(e2.sqrt()).c_str(), hex2(e.Frobenius_map(1)).c_str());
  }
  // Montgomery representation of 1 and of a sample value (pins R = 2^256 and the limb layout of key files)
  { FrT one = FrT::one(); FqT oneq = FqT::one();
    fprintf(o, "mont_one_fr %016lx%016lx%016lx%016lx\n", one.mont_repr.data[3], one.mont_repr.data[2], one.mont_repr.data[1], one.mont_repr.data[0]);
    fprintf(o, "mont_one_fq %016lx%016lx%016lx%016lx\n", oneq.mont_repr.data[3], oneq.mont_repr.data[2], oneq.mont_repr.data[1], oneq.mont_repr.data[0]); }
  fprintf(o, "fr_root_of_unity_2^28 %s\n", hex_of(FrT::root_of_unity).c_str());
  fprintf(o, "fr_mult_gen %s\n", hex_of(FrT::multiplicative_generator).c_str());
  fprintf(o, "twist_b %s\n", hex2(alt_bn128_twist_coeff_b).c_str());
}

static void emit_curve_vectors(FILE *o) {
  SplitMix g(0xC0FFEE);
  for (int i = 0; i < 6; i++) {
    FrT a = rand_field<FrT>(g), b = rand_field<FrT>(g), k = rand_field<FrT>(g);
    G1T P = a * G1T::one(), Q = b * G1T::one(); G1T Qa = Q; Qa.to_special();
    fprintf(o, "g1 a=%s b=%s k=%s P=%s Q=%s add=%s dbl=%s madd=%s kP=%s PminusQ=%s\n", hex_of(a).c_str(), hex_of(b).c_str(), hex_of(k).c_str(),
            g1hex(P).c_str(), g1hex(Q).c_str(), g1hex(P + Q).c_str(), g1hex(P.dbl()).c_str(), g1hex(P.mixed_add(Qa)).c_str(), g1hex(k * P).c_str(), g1hex(P - Q).c_str());
    G2T R = a * G2T::one(), S = b * G2T::one(); G2T Sa = S; Sa.to_special();
    fprintf(o, "g2 a=%s b=%s k=%s P=%s Q=%s add=%s dbl=%s madd=%s kP=%s\n", hex_of(a).c_str(), hex_of(b).c_str(), hex_of(k).c_str(),
            g2hex(R).c_str(), g2hex(S).c_str(), g2hex(R + S).c_str(), g2hex(R.dbl()).c_str(), g2hex(R.mixed_add(Sa)).c_str(), g2hex(k * R).c_str());
  }
  fprintf(o, "g1_one %s\n", g1hex(G1T::one()).c_str());
  fprintf(o, "g2_one %s\n", g2hex(G2T::one()).c_str());
  // compressed key-file encoding of a point (alt_bn128_g1.cpp:404-418 under BINARY_OUTPUT/MONTGOMERY_OUTPUT/pt compression)
  { G1T P = FrT(12345) * G1T::one(); std::stringstream ss; ss << P; std::string s = ss.str(); fprintf(o, "g1_ser 12345G %s ", g1hex(P).c_str());
    for (unsigned char ch : s) fprintf(o, "%02x", ch); fprintf(o, "\n");
    G2T Q = FrT(12345) * G2T::one(); std::stringstream s2; s2 << Q; s = s2.str(); fprintf(o, "g2_ser 12345G %s ", g2hex(Q).c_str());
    for (unsigned char ch : s) fprintf(o, "%02x", ch); fprintf(o, "\n");
    G1T Z = G1T::zero(); std::stringstream s3; s3 << Z; s = s3.str(); fprintf(o, "g1_ser zero inf "); for (unsigned char ch : s) fprintf(o, "%02x", ch); fprintf(o, "\n"); }
}

static void emit_domain_vectors(FILE *o) {
  // sizes: basic radix-2 and step radix-2 (2^k + 2^r); min_size handed to get_evaluation_domain exactly as r1cs_to_qap does
  const size_t sizes[] = {2, 4, 16, 24, 48, 64, 80, 1024, 1536, 4096, 5120};
  for (size_t m : sizes) {
    auto dom = libfqfft::get_evaluation_domain<FrT>(m);
    SplitMix g(0xD0D0 + m); std::vector<FrT> a(dom->m); for (auto &x : a) x = rand_field<FrT>(g);
    FrT t = rand_field<FrT>(g);
    auto dump = [&](const char *name, const std::vector<FrT> &v) { fprintf(o, "domain m=%zu dm=%zu %s", m, dom->m, name);
      if (v.size() <= 80) for (auto &x : v) fprintf(o, " %s", hex_of(x).c_str());
      else { FrT acc = FrT::zero(), w = FrT::one(); for (auto &x : v) { acc += w * x; w *= t; } fprintf(o, " polyeval_at_t=%s first=%s last=%s", hex_of(acc).c_str(), hex_of(v[0]).c_str(), hex_of(v.back()).c_str()); }
      fprintf(o, "\n"); };
    fprintf(o, "domain m=%zu dm=%zu t %s\n", m, dom->m, hex_of(t).c_str());
    dump("in", a);
    std::vector<FrT> b = a; dom->FFT(b); dump("fft", b);
    b = a; dom->iFFT(b); dump("ifft", b);
    b = a; dom->cosetFFT(b, FrT::multiplicative_generator); dump("cosetfft", b);
    b = a; dom->icosetFFT(b, FrT::multiplicative_generator); dump("icosetfft", b);
    b = a; dom->divide_by_Z_on_coset(b); dump("divZ", b);
    dump("lagrange", dom->evaluate_all_lagrange_polynomials(t));
    fprintf(o, "domain m=%zu dm=%zu Zt %s\n", m, dom->m, hex_of(dom->compute_vanishing_polynomial(t)).c_str());
    std::vector<FrT> h(dom->m + 1, FrT::zero()); dom->add_poly_Z(t, h); fprintf(o, "domain m=%zu dm=%zu addZ", m, dom->m);
    for (size_t i = 0; i <= dom->m; i++) if (!h[i].is_zero()) fprintf(o, " %zu:%s", i, hex_of(h[i]).c_str()); fprintf(o, "\n");
  }
  // domain selection for the circuit sizes in SURVEY.md §6
  for (size_t m : {(size_t)167270 + 5, (size_t)252286 + 6, (size_t)503863 + 7, (size_t)1177039 + 7}) {
    auto dom = libfqfft::get_evaluation_domain<FrT>(m); fprintf(o, "domain_select min=%zu m=%zu\n", m, dom->m); }
}

static void emit_msm_vectors(FILE *o) {
  for (size_t n : {(size_t)1, (size_t)2, (size_t)33, (size_t)1000, (size_t)4096}) {
    SplitMix g(0x3535 + n); std::vector<G1T> P(n); std::vector<G2T> Q(n); std::vector<FrT> k(n), z(n);
    // bases: P_i = (b0 + i)·G so that tests can rebuild them cheaply; scalars: full-width / witness-like mix
    FrT b0 = rand_field<FrT>(g); G1T p = b0 * G1T::one(); G2T q = b0 * G2T::one();
    for (size_t i = 0; i < n; i++) { P[i] = p; Q[i] = q; p = p + G1T::one(); q = q + G2T::one(); }
    batch_to_special<G1T>(P); batch_to_special<G2T>(Q);
    for (size_t i = 0; i < n; i++) { k[i] = rand_field<FrT>(g); uint64_t sel = g.next() % 100;
      z[i] = sel < 50 ? FrT::zero() : sel < 95 ? FrT::one() : sel < 98 ? FrT((long)(g.next() & 0xffffffff)) : rand_field<FrT>(g); }
    G1T r1 = multi_exp<G1T, FrT, multi_exp_method_BDLO12>(P.begin(), P.end(), k.begin(), k.end(), 1);
    G2T r2 = multi_exp<G2T, FrT, multi_exp_method_BDLO12>(Q.begin(), Q.end(), k.begin(), k.end(), 1);
    G1T r3 = multi_exp_with_mixed_addition<G1T, FrT, multi_exp_method_BDLO12>(P.begin(), P.end(), z.begin(), z.end(), 1);
    G2T r4 = multi_exp_with_mixed_addition<G2T, FrT, multi_exp_method_BDLO12>(Q.begin(), Q.end(), z.begin(), z.end(), 1);
    fprintf(o, "msm n=%zu b0=%s g1_full=%s g2_full=%s g1_mixed=%s g2_mixed=%s\n", n, hex_of(b0).c_str(), g1hex(r1).c_str(), g2hex(r2).c_str(), g1hex(r3).c_str(), g2hex(r4).c_str());
  }
}

static std::string gt_dec(const GT<ppT> &x) { std::stringstream ss; ss << x; return ss.str(); }
static void emit_pairing_vectors(FILE *o) {
  SplitMix g(0xAB12);
  for (int i = 0; i < 3; i++) { FrT a = i == 0 ? FrT::one() : rand_field<FrT>(g), b = i == 0 ? FrT::one() : rand_field<FrT>(g);
    GT<ppT> e = ppT::reduced_pairing(a * G1T::one(), b * G2T::one());
    fprintf(o, "pairing a=%s b=%s gt=%s\n", hex_of(a).c_str(), hex_of(b).c_str(), gt_dec(e).c_str()); }
}

static int cmd_vectors(const char *out) {
  FILE *o = fopen(out, "w"); if (!o) { perror(out); return 2; }
  emit_field_vectors(o); emit_curve_vectors(o); emit_domain_vectors(o); emit_msm_vectors(o); emit_pairing_vectors(o); fclose(o); return 0; }

// ---- libsnark gadgets: R1CS + witness dumps (pins the product's circuit building blocks) ------------------------
static int cmd_sha256gadget(const char *r1cs_out, const char *wit_out, uint64_t seed) {
  // same shape as gadgetlib1/gadgets/hashes/sha256/tests/test_sha256_gadget.cpp:20-41: two_to_one hash of a 512-bit block
  protoboard<FrT> pb; digest_variable<FrT> left(pb, SHA256_digest_size, "left"), right(pb, SHA256_digest_size, "right"), output(pb, SHA256_digest_size, "output");
  sha256_two_to_one_hash_gadget<FrT> f(pb, left, right, output, "f"); f.generate_r1cs_constraints();
  SplitMix g(seed); bit_vector lb(256), rb(256);
  if (seed == 0) { // the reference's own KAT inputs (test_sha256_gadget.cpp:29-31)
    lb = int_list_to_bits({0x426bc2d8, 0x4dc86782, 0x81e8957a, 0x409ec148, 0xe6cffbe8, 0xafe6ba4f, 0x9c6f1978, 0xdd7af7e9}, 32);
    rb = int_list_to_bits({0x038cce42, 0xabd366b8, 0x3ede7e00, 0x9130de53, 0x72cdf73d, 0xee825114, 0x8cb48d1b, 0x9af68ad0}, 32);
  } else for (int i = 0; i < 256; i++) { lb[i] = g.next() & 1; rb[i] = g.next() & 1; }
  left.generate_r1cs_witness(lb); right.generate_r1cs_witness(rb); f.generate_r1cs_witness();
  if (!pb.is_satisfied()) { fprintf(stderr, "unsatisfied\n"); return 1; }
  save_r1cs(r1cs_out, pb.get_constraint_system()); save_wit(wit_out, pb.full_variable_assignment());
  printf("sha256gadget constraints=%zu variables=%zu digest=", pb.num_constraints(), pb.num_variables());
  for (bool b : output.get_digest()) printf("%d", b ? 1 : 0); printf("\n"); return 0; }

static int cmd_merklegadget(size_t depth, const char *r1cs_out, const char *wit_out, uint64_t seed) {
  // mirrors merkle_tree_check_read_gadget.tcc:131-196 (test_merkle_tree_check_read_gadget) with seeded bits
  typedef sha256_two_to_one_hash_gadget<FrT> HashT; const size_t dl = HashT::get_digest_len(); SplitMix g(seed);
  std::vector<merkle_authentication_node> path(depth); bit_vector prev_hash(dl), leaf, address_bits; for (size_t i = 0; i < dl; i++) prev_hash[i] = g.next() & 1; leaf = prev_hash;
  size_t address = 0;
  for (long level = depth - 1; level >= 0; --level) { const bool computed_is_right = g.next() & 1; address |= (computed_is_right ? 1ul << (depth - 1 - level) : 0); address_bits.push_back(computed_is_right);
    bit_vector other(dl); for (size_t i = 0; i < dl; i++) other[i] = g.next() & 1; bit_vector block = prev_hash; block.insert(computed_is_right ? block.begin() : block.end(), other.begin(), other.end());
    bit_vector h = HashT::get_hash(block); path[level] = other; prev_hash = h; }
  bit_vector root = prev_hash;
  protoboard<FrT> pb; pb_variable_array<FrT> address_bits_va; address_bits_va.allocate(pb, depth, "address_bits");
  digest_variable<FrT> leaf_digest(pb, dl, "input_block"), root_digest(pb, dl, "output_digest"); merkle_authentication_path_variable<FrT, HashT> path_var(pb, depth, "path_var");
  merkle_tree_check_read_gadget<FrT, HashT> ml(pb, depth, address_bits_va, leaf_digest, root_digest, path_var, pb_variable<FrT>(0), "ml");
  path_var.generate_r1cs_constraints(); ml.generate_r1cs_constraints();
  address_bits_va.fill_with_bits(pb, address_bits); leaf_digest.generate_r1cs_witness(leaf); path_var.generate_r1cs_witness(address, path); ml.generate_r1cs_witness();
  address_bits_va.fill_with_bits(pb, address_bits); leaf_digest.generate_r1cs_witness(leaf); root_digest.generate_r1cs_witness(root);
  if (!pb.is_satisfied()) { fprintf(stderr, "unsatisfied\n"); return 1; }
  save_r1cs(r1cs_out, pb.get_constraint_system()); save_wit(wit_out, pb.full_variable_assignment());
  printf("merklegadget depth=%zu constraints=%zu variables=%zu address=%zu\n", depth, pb.num_constraints(), pb.num_variables(), address); return 0; }

// ---- BlockMaze's comparison gadget and bit-order helpers, compiled from libsnark-vnt/src/send ------------------------------
// The block of send's / redeem's note gadgets that proves value_s <= value_old, rebuilt from the reference's own pieces exactly as note.tcc:35-59,78-83 +
// less_cmp.tcc:23-34 compose them (those two files cannot be included: they call into utils.tcc): 64 + 64 bit variables, the two packed values,
// less_comparison_gadget(value_s_packed, value_old_packed); booleanity of the bits, the gadget's constraints; the bits filled in the order of
// uint64_to_bool_vector (utils.tcc:47-53 = convertIntToVectorLE + convertBytesVectorToVector of util.h, both called here), the packed values by BlockMaze's
// get_field_element_from_bits_by_order (pb_variable.tcc:119-133).
static int cmd_lesscmp(uint64_t v_old, uint64_t v_s, const char *r1cs_out, const char *wit_out) {
  protoboard<FrT> pb; pb_variable_array<FrT> value_old, value_s; value_old.allocate(pb, 64, "value_old"); value_s.allocate(pb, 64, "value_s");
  pb_variable<FrT> value_old_packed, value_s_packed; value_old_packed.allocate(pb, "value_old_packed"); value_s_packed.allocate(pb, "value_s_packed");
  less_comparison_gadget<FrT> less_cmp(pb, value_s_packed, value_old_packed, " less_cmp");
  for (size_t i = 0; i < 64; i++) generate_boolean_r1cs_constraint<FrT>(pb, value_old[i], "boolean_value_old");
  for (size_t i = 0; i < 64; i++) generate_boolean_r1cs_constraint<FrT>(pb, value_s[i], "boolean_value_s");
  less_cmp.generate_r1cs_constraints();
  auto bits_of = [](uint64_t v) { std::vector<unsigned char> le = convertIntToVectorLE(v); std::vector<bool> b(64, 0); convertBytesVectorToVector(le, b); return b; };
  value_old.fill_with_bits(pb, bits_of(v_old)); pb.lc_val(value_old_packed) = value_old.get_field_element_from_bits_by_order(pb);
  value_s.fill_with_bits(pb, bits_of(v_s)); pb.lc_val(value_s_packed) = value_s.get_field_element_from_bits_by_order(pb);
  less_cmp.generate_r1cs_witness();
  save_r1cs(r1cs_out, pb.get_constraint_system()); save_wit(wit_out, pb.full_variable_assignment());
  printf("lesscmp constraints=%zu variables=%zu satisfied=%d packed_old=%s packed_s=%s\n", pb.num_constraints(), pb.num_variables(), pb.is_satisfied() ? 1 : 0, hex_of(pb.val(value_old_packed)).c_str(), hex_of(pb.val(value_s_packed)).c_str()); return 0; }

// The public-input unpacker of the four circuits: packed inputs allocated first and declared the primary input (send/circuit/gadget.tcc:87-88), then the unpacked bits
// (:90-93: digest_variables / a 64-bit array, i.e. consecutive variables), multipacking_gadget(pb, bits, packed, FieldT::capacity()) (:99-105) and
// generate_r1cs_constraints(true) (:198); the witness goes from the bits to the packed values (generate_r1cs_witness_from_bits, :268).
static int cmd_unpacker(size_t nbits, uint64_t seed, const char *r1cs_out, const char *wit_out) {
  protoboard<FrT> pb; pb_variable_array<FrT> packed, bits; size_t np = (nbits + FrT::capacity() - 1) / FrT::capacity(); packed.allocate(pb, np, "packed"); pb.set_input_sizes(np); bits.allocate(pb, nbits, "bits");
  multipacking_gadget<FrT> unpacker(pb, bits, packed, FrT::capacity(), "unpacker"); unpacker.generate_r1cs_constraints(true);
  SplitMix g(seed); std::vector<bool> bv(nbits); for (size_t i = 0; i < nbits; i++) bv[i] = g.next() & 1; bits.fill_with_bits(pb, bv); unpacker.generate_r1cs_witness_from_bits();
  save_r1cs(r1cs_out, pb.get_constraint_system()); save_wit(wit_out, pb.full_variable_assignment());
  printf("unpacker bits=%zu constraints=%zu variables=%zu inputs=%zu satisfied=%d\n", nbits, pb.num_constraints(), pb.num_variables(), np, pb.is_satisfied() ? 1 : 0); return 0; }

// Two chained libsnark compression gadgets composed the way BlockMaze's sha256_CMTA_gadget composes them (send/circuit/commitment.tcc:12-110, which itself cannot be
// included: its padding comes from utils.tcc's from_bits): ZERO, value[64], sn[256], r[256], the output digest; then the intermediate digest, block1 = value | sn |
// r[0..192), block2 = r[192..256) | padding for a 576-bit message made of the constant ONE and the variable ZERO, hasher1 from SHA256_default_IV, hasher2 from the
// intermediate digest's bits.  Constraints: ZERO = 0, intermediate booleanity, hasher1, hasher2.  Pins the chained (non-IV) use of the compression gadget and the
// zero-coefficient bookkeeping of both kinds.
static int cmd_cmta(uint64_t seed, const char *r1cs_out, const char *wit_out) {
  protoboard<FrT> pb; pb_variable<FrT> ZERO; ZERO.allocate(pb, "zero"); pb_variable_array<FrT> v, sn, r; v.allocate(pb, 64, "v"); sn.allocate(pb, 256, "sn"); r.allocate(pb, 256, "r");
  digest_variable<FrT> cmtA(pb, 256, "cmtA"), inter(pb, 256, "");
  pb_variable_array<FrT> first_of_r(r.begin(), r.begin() + 192), last_of_r(r.begin() + 192, r.end()), padding;
  for (size_t i = 0; i < 448; i++) { bool bit = i == 0 || (i >= 384 && ((576ull >> (447 - i)) & 1)); padding.emplace_back(bit ? pb_variable<FrT>(0) : ZERO); }   // from_bits (utils.tcc:3-12): ONE is variable 0
  block_variable<FrT> block1(pb, {v, sn, first_of_r}, "b1"), block2(pb, {last_of_r, padding}, "b2");
  sha256_compression_function_gadget<FrT> h1(pb, SHA256_default_IV<FrT>(pb), block1.bits, inter, "h1"), h2(pb, pb_linear_combination_array<FrT>(inter.bits), block2.bits, cmtA, "h2");
  pb.add_r1cs_constraint(r1cs_constraint<FrT>(1, ZERO, 0), "zero"); inter.generate_r1cs_constraints(); h1.generate_r1cs_constraints(); h2.generate_r1cs_constraints();
  SplitMix g(seed); bit_vector bv(64), bsn(256), br(256); for (auto &&b : bv) b = g.next() & 1; for (auto &&b : bsn) b = g.next() & 1; for (auto &&b : br) b = g.next() & 1;
  pb.val(ZERO) = FrT::zero(); v.fill_with_bits(pb, bv); sn.fill_with_bits(pb, bsn); r.fill_with_bits(pb, br); h1.generate_r1cs_witness(); h2.generate_r1cs_witness();
  if (!pb.is_satisfied()) { fprintf(stderr, "unsatisfied\n"); return 1; }
  save_r1cs(r1cs_out, pb.get_constraint_system()); save_wit(wit_out, pb.full_variable_assignment());
  size_t terms[3] = {0, 0, 0}, zeros[3] = {0, 0, 0};
  for (auto &c : pb.get_constraint_system().constraints) { const linear_combination<FrT> *l[3] = {&c.a, &c.b, &c.c}; for (int m = 0; m < 3; m++) for (auto &t : l[m]->terms) { terms[m]++; if (t.coeff.is_zero()) zeros[m]++; } }
  printf("cmta constraints=%zu variables=%zu terms=%zu,%zu,%zu zero_terms=%zu,%zu,%zu digest=", pb.num_constraints(), pb.num_variables(), terms[0], terms[1], terms[2], zeros[0], zeros[1], zeros[2]);
  for (bool b : cmtA.get_digest()) printf("%d", b ? 1 : 0); printf("\n"); return 0; }

// The other three hash blocks of BlockMaze's circuits, composed from libsnark's own classes exactly as the reference composes them (send/circuit/commitment.tcc:
// sha256_CMTS_gadget :100-170 — 736-bit message, the serial number split 32 / 224 over the two blocks; sha256_PRF_gadget :172-258 — 512-bit message and a block of pure
// padding; sha256_CRH_gadget :260-320 — 416 bits and their padding in ONE block).  commitment.tcc itself includes circuit/utils.tcc (BOOST_FOREACH) and cannot be built here.
static int cmd_hashblock(const std::string &kind, uint64_t seed, const char *r1cs_out, const char *wit_out) {
  protoboard<FrT> pb; pb_variable<FrT> ZERO; ZERO.allocate(pb, "zero");
  std::vector<size_t> widths = kind == "cmts" ? std::vector<size_t>{64, 160, 256, 256} : kind == "prf" ? std::vector<size_t>{256, 256} : std::vector<size_t>{160, 256};
  std::vector<pb_variable_array<FrT>> in(widths.size()); size_t msg_bits = 0; for (size_t i = 0; i < widths.size(); i++) { in[i].allocate(pb, widths[i], "in"); msg_bits += widths[i]; }
  digest_variable<FrT> out(pb, 256, "out");
  auto padding = [&](size_t total) { pb_variable_array<FrT> p; size_t n = total - msg_bits; for (size_t i = 0; i < n; i++) { size_t from_end = n - 1 - i; bool bit = i == 0 || (from_end < 64 && ((uint64_t)msg_bits >> from_end) & 1); p.emplace_back(bit ? pb_variable<FrT>(0) : ZERO); } return p; };   // from_bits (utils.tcc:3-12): ONE is variable 0
  std::unique_ptr<digest_variable<FrT>> inter; std::unique_ptr<block_variable<FrT>> b1, b2; std::unique_ptr<sha256_compression_function_gadget<FrT>> h1, h2;
  if (kind == "crh") { b1.reset(new block_variable<FrT>(pb, {in[0], in[1], padding(512)}, "b1")); h1.reset(new sha256_compression_function_gadget<FrT>(pb, SHA256_default_IV<FrT>(pb), b1->bits, out, "h1")); }
  else { inter.reset(new digest_variable<FrT>(pb, 256, ""));
    if (kind == "cmts") { pb_variable_array<FrT> first(in[3].begin(), in[3].begin() + 32), last(in[3].begin() + 32, in[3].end()); b1.reset(new block_variable<FrT>(pb, {in[0], in[1], in[2], first}, "b1")); b2.reset(new block_variable<FrT>(pb, {last, padding(1024)}, "b2")); }
    else { b1.reset(new block_variable<FrT>(pb, {in[0], in[1]}, "b1")); b2.reset(new block_variable<FrT>(pb, {padding(1024)}, "b2")); }
    h1.reset(new sha256_compression_function_gadget<FrT>(pb, SHA256_default_IV<FrT>(pb), b1->bits, *inter, "h1")); h2.reset(new sha256_compression_function_gadget<FrT>(pb, pb_linear_combination_array<FrT>(inter->bits), b2->bits, out, "h2")); }
  pb.add_r1cs_constraint(r1cs_constraint<FrT>(1, ZERO, 0), "zero"); if (inter) inter->generate_r1cs_constraints(); h1->generate_r1cs_constraints(); if (h2) h2->generate_r1cs_constraints();
  SplitMix g(seed); pb.val(ZERO) = FrT::zero(); for (size_t i = 0; i < in.size(); i++) { bit_vector bv(widths[i]); for (auto &&b : bv) b = g.next() & 1; in[i].fill_with_bits(pb, bv); }
  h1->generate_r1cs_witness(); if (h2) h2->generate_r1cs_witness();
  if (!pb.is_satisfied()) { fprintf(stderr, "unsatisfied\n"); return 1; }
  save_r1cs(r1cs_out, pb.get_constraint_system()); save_wit(wit_out, pb.full_variable_assignment());
  size_t terms[3] = {0, 0, 0}, zeros[3] = {0, 0, 0};
  for (auto &c : pb.get_constraint_system().constraints) { const linear_combination<FrT> *l[3] = {&c.a, &c.b, &c.c}; for (int m = 0; m < 3; m++) for (auto &t : l[m]->terms) { terms[m]++; if (t.coeff.is_zero()) zeros[m]++; } }
  printf("%s constraints=%zu variables=%zu terms=%zu,%zu,%zu zero_terms=%zu,%zu,%zu digest=", kind.c_str(), pb.num_constraints(), pb.num_variables(), terms[0], terms[1], terms[2], zeros[0], zeros[1], zeros[2]);
  for (bool b : out.get_digest()) printf("%d", b ? 1 : 0); printf("\n"); return 0; }

// host note hashing through the reference's own classes (send/Note.h:14-80, util.h:233-258, uint256.h:222-248 via uint256S / uint160S)
static int cmd_notehashes(uint64_t seed, int count) {
  SplitMix g(seed); auto hex = [&](int nbytes) { std::string s = "0x"; char b[3]; for (int i = 0; i < nbytes; i++) { snprintf(b, 3, "%02x", (unsigned)(g.next() & 0xff)); s += b; } return s; };
  for (int i = 0; i < count; i++) { std::string sk = hex(32), r = hex(32), sn = hex(32), pk = hex(20); uint64_t v = g.next(); if (i == 0) v = 0; if (i == 1) { sk = "0x1"; r = "0x123456"; }   // short strings: zero-extension
    uint256 sk_ = uint256S(sk), r_ = uint256S(r), sn_ = uint256S(sn); uint160 pk_ = uint160S(pk);
    printf("notehash v=%llu sk=%s r=%s sn=%s pk=%s prf=%s crh=%s cm=%s cms=%s\n", (unsigned long long)v, sk.c_str(), r.c_str(), sn.c_str(), pk.c_str(), Compute_PRF(sk_, r_).GetHex().c_str(), Compute_CRH(pk_, r_).GetHex().c_str(),
           Note(v, sn_, r_).cm().GetHex().c_str(), NoteS(v, pk_, r_, sn_).cm().GetHex().c_str()); }
  return 0; }

// ---- Groth16 over an R1CS supplied by the caller ---------------------------------------------------------------
static int cmd_e2e(const char *r1cs_path, const char *wit_path, const char *r_hex, const char *s_hex, const std::string &outdir) {
  double t0 = now_s(); r1cs_constraint_system<FrT> cs = load_r1cs(r1cs_path); std::vector<FrT> w = load_wit(wit_path);
  std::vector<FrT> primary(w.begin(), w.begin() + cs.num_inputs()), aux(w.begin() + cs.num_inputs(), w.end());
  printf("r1cs constraints=%zu variables=%zu inputs=%zu load_s=%.3f\n", cs.num_constraints(), cs.num_variables(), cs.num_inputs(), now_s() - t0);
  bool sat = cs.is_satisfied(primary, aux); printf("satisfied %d\n", sat ? 1 : 0); if (!sat) return 1;
  t0 = now_s(); r1cs_gg_ppzksnark_keypair<ppT> kp = r1cs_gg_ppzksnark_generator<ppT>(cs); printf("generator_s %.3f\n", now_s() - t0);
  t0 = now_s(); write_obj(outdir + "/pk.txt", kp.pk); write_obj(outdir + "/vk.txt", kp.vk); printf("write_keys_s %.3f\n", now_s() - t0);
  FrT r = from_hex<FrT>(r_hex), s = from_hex<FrT>(s_hex); std::vector<FrT> h;
  t0 = now_s(); r1cs_gg_ppzksnark_proof<ppT> pr = prove_fixed(kp.pk, primary, aux, r, s, &h); printf("prove_s %.3f\n", now_s() - t0);
  save_wit((outdir + "/h_coeffs.bin").c_str(), h);
  printf("proof %s\n", proof_hex(pr).c_str());
  t0 = now_s(); bool ok = r1cs_gg_ppzksnark_verifier_strong_IC<ppT>(kp.vk, primary, pr); printf("verify %d verify_s %.4f\n", ok ? 1 : 0, now_s() - t0);
  // round trip through the hex form, as verifySendproof does (sendcgo.cpp:388-450)
  bool ok2 = r1cs_gg_ppzksnark_verifier_strong_IC<ppT>(kp.vk, primary, proof_from_hex(proof_hex(pr))); printf("verify_hex_roundtrip %d\n", ok2 ? 1 : 0);
  return ok && ok2 ? 0 : 1; }

static int cmd_prove(const char *pk_path, const char *wit_path, size_t n_inputs, const char *r_hex, const char *s_hex) {
  double t0 = now_s(); auto pk = read_obj<r1cs_gg_ppzksnark_proving_key<ppT>>(pk_path); printf("load_pk_s %.3f\n", now_s() - t0);
  std::vector<FrT> w = load_wit(wit_path); std::vector<FrT> primary(w.begin(), w.begin() + n_inputs), aux(w.begin() + n_inputs, w.end());
  bool sat = pk.constraint_system.is_satisfied(primary, aux); printf("satisfied %d\n", sat ? 1 : 0); if (!sat) return 1;
  t0 = now_s(); auto pr = prove_fixed(pk, primary, aux, from_hex<FrT>(r_hex), from_hex<FrT>(s_hex)); printf("prove_s %.3f\n", now_s() - t0);
  printf("proof %s\n", proof_hex(pr).c_str()); return 0; }

static int cmd_verify(int argc, char **argv) {
  auto vk = read_obj<r1cs_gg_ppzksnark_verification_key<ppT>>(argv[2]); std::string ph = argv[3]; size_t n = atoi(argv[4]);
  if (ph.size() < 512 || (size_t)argc < 5 + n) { fprintf(stderr, "usage\n"); return 2; }
  std::vector<FrT> primary; for (size_t i = 0; i < n; i++) primary.push_back(FrT(bigint<4>(argv[5 + i])));
  double t0 = now_s(); bool ok = r1cs_gg_ppzksnark_verifier_strong_IC<ppT>(vk, primary, proof_from_hex(ph)); printf("verify %d verify_s %.4f\n", ok ? 1 : 0, now_s() - t0); return ok ? 0 : 1; }

// one verdict per line of a file: what verifySendproof (sendcgo.cpp:388-448) decides for that character string and those public inputs.  The characters are
// the caller's business: a line that is not 512 lowercase hex digits is not fed to the reference (convertFromAscii, sendcgo.cpp:25-35, has no defined value there).
static int cmd_verifymany(const char *vk_path, const char *cases_path) {
  auto vk = read_obj<r1cs_gg_ppzksnark_verification_key<ppT>>(vk_path); std::ifstream in(cases_path); std::string line; size_t k = 0;
  while (std::getline(in, line)) {
    std::istringstream ls(line); std::string ph; size_t n = 0; ls >> ph >> n; if (ph.size() != 512) { fprintf(stderr, "line %zu: not 512 characters\n", k); return 2; }
    for (char c : ph) if (!((c >= '0' && c <= '9') || (c >= 'a' && c <= 'f'))) { fprintf(stderr, "line %zu: not lowercase hex\n", k); return 2; }
    std::vector<FrT> primary; for (size_t i = 0; i < n; i++) { std::string d; ls >> d; primary.push_back(FrT(bigint<4>(d.c_str()))); }
    // every case in a child process: the reference goes on to the pairing after is_well_formed() has failed (r1cs_gg_ppzksnark.tcc:528-560), and with B = (0, 0)
    // (Z = 1) the Miller loop's value is 0, whose inversion in the final exponentiation trips assert(!is_zero()) (fp.tcc:648) in a build without -DNDEBUG such as
    // this one: SIGABRT.  Reported as "v <line> A" — the verdict of an NDEBUG build for the same input is "reject" (result was already false).
    fflush(stdout); pid_t pid = fork(); if (pid < 0) { perror("fork"); return 2; }
    if (pid == 0) { int fd = open("/dev/null", O_WRONLY); if (fd >= 0) dup2(fd, 2); bool ok = r1cs_gg_ppzksnark_verifier_strong_IC<ppT>(vk, primary, proof_from_hex(ph)); _exit(ok ? 1 : 0); }
    int st = 0; if (waitpid(pid, &st, 0) != pid) { perror("waitpid"); return 2; }
    if (WIFEXITED(st)) printf("v %zu %d\n", k++, WEXITSTATUS(st)); else printf("v %zu A\n", k++); }
  return 0; }

static int cmd_hexblobs(const char *path) {
  std::ifstream in(path); std::string line; size_t k = 0;
  while (std::getline(in, line)) {
    std::string s; for (size_t i = 0; i + 1 < line.size(); i += 2) s.push_back((char)strtoul(line.substr(i, 2).c_str(), nullptr, 16));
    uint256 a = uint256S(s); uint160 b = uint160S(s); uint256 zero = uint256S("");
    printf("blob %zu %s %s %s %s\n", k++, a.GetHex().c_str(), b.GetHex().c_str(), Compute_PRF(a, zero).GetHex().c_str(), Compute_CRH(b, zero).GetHex().c_str()); }
  return 0; }

static int cmd_bench_prover(const char *r1cs_path, const char *wit_path) {
  // Times the STOCK prover (r1cs_gg_ppzksnark.tcc:391-506) on a proving key of the exact shape the generator
  // would emit for this R1CS, but with cheap synthetic points (i·G) — prover time does not depend on point values.
  r1cs_constraint_system<FrT> cs = load_r1cs(r1cs_path); std::vector<FrT> w = load_wit(wit_path); cs.swap_AB_if_beneficial();
  std::vector<FrT> primary(w.begin(), w.begin() + cs.num_inputs()), aux(w.begin() + cs.num_inputs(), w.end());
  auto dom = libfqfft::get_evaluation_domain<FrT>(cs.num_constraints() + cs.num_inputs() + 1); const size_t nv = cs.num_variables(), m = dom->m;
  size_t total = (nv + 1) + (m - 1) + (nv - cs.num_inputs()); std::vector<G1T> pts(total); G1T p = FrT(7) * G1T::one();
  for (auto &x : pts) { x = p; p = p + G1T::one(); } batch_to_special<G1T>(pts);
  std::vector<bool> b_used(nv + 1, false); for (auto &c : cs.constraints) for (auto &t : c.b.terms) b_used[t.index] = true;
  std::vector<size_t> bidx; for (size_t i = 0; i <= nv; i++) if (b_used[i]) bidx.push_back(i);
  std::vector<G2T> q2(bidx.size()); G2T q = FrT(7) * G2T::one(); for (auto &x : q2) { x = q; q = q + G2T::one(); } batch_to_special<G2T>(q2);
  G1_vector<ppT> A(pts.begin(), pts.begin() + nv + 1), H(pts.begin() + nv + 1, pts.begin() + nv + m), L(pts.begin() + nv + m, pts.end());
  knowledge_commitment_vector<G2T, G1T> B; B.domain_size_ = nv + 1; for (size_t j = 0; j < bidx.size(); j++) { B.indices.push_back(bidx[j]); B.values.emplace_back(knowledge_commitment<G2T, G1T>(q2[j], A[bidx[j]])); }
  G1T k_a = pts[0], k_b = pts[1], k_d = pts[2]; G2T k_b2 = q2[0], k_d2 = q2[1];
  r1cs_gg_ppzksnark_proving_key<ppT> pk(std::move(k_a), std::move(k_b), std::move(k_b2), std::move(k_d), std::move(k_d2), std::move(A), std::move(B), std::move(H), std::move(L), std::move(cs));
  printf("bench_prover constraints=%zu variables=%zu domain=%zu b_nonzero=%zu\n", pk.constraint_system.num_constraints(), nv, m, bidx.size());
  double t0 = now_s(); auto qw = r1cs_to_qap_witness_map(pk.constraint_system, primary, aux, FrT::zero(), FrT::zero(), FrT::zero()); double t_wm = now_s() - t0;
  t0 = now_s(); auto pr = r1cs_gg_ppzksnark_prover<ppT>(pk, primary, aux); double t_total = now_s() - t0;
  int threads = 1;
#ifdef MULTICORE
  threads = omp_get_max_threads();
#endif
  printf("witness_map_s %.3f prover_total_s %.3f proofs_per_s %.5f threads %d\n", t_wm, t_total, 1.0 / t_total, threads); (void)pr; (void)qw; return 0; }

int main(int argc, char **argv) {
  ppT::init_public_params(); libff::inhibit_profiling_info = true; libff::inhibit_profiling_counters = true;
  if (argc < 2) { fprintf(stderr, "usage: see header of oracle/ref_harness.cpp\n"); return 2; }
  std::string m = argv[1];
  if (m == "vectors" && argc == 3) return cmd_vectors(argv[2]);
  if (m == "sha256gadget" && argc == 5) return cmd_sha256gadget(argv[2], argv[3], strtoull(argv[4], 0, 0));
  if (m == "merklegadget" && argc == 6) return cmd_merklegadget(atoi(argv[2]), argv[3], argv[4], strtoull(argv[5], 0, 0));
  if (m == "unpacker" && argc == 6) return cmd_unpacker(strtoull(argv[2], 0, 0), strtoull(argv[3], 0, 0), argv[4], argv[5]);
  if (m == "lesscmp" && argc == 6) return cmd_lesscmp(strtoull(argv[2], 0, 0), strtoull(argv[3], 0, 0), argv[4], argv[5]);
  if (m == "cmta" && argc == 5) return cmd_cmta(strtoull(argv[2], 0, 0), argv[3], argv[4]);
  if (m == "hashblock" && argc == 6 && (!strcmp(argv[2], "cmts") || !strcmp(argv[2], "prf") || !strcmp(argv[2], "crh"))) return cmd_hashblock(argv[2], strtoull(argv[3], 0, 0), argv[4], argv[5]);
  if (m == "notehashes" && argc == 4) return cmd_notehashes(strtoull(argv[2], 0, 0), atoi(argv[3]));
  if (m == "e2e" && argc == 7) return cmd_e2e(argv[2], argv[3], argv[4], argv[5], argv[6]);
  if (m == "prove" && argc == 7) return cmd_prove(argv[2], argv[3], atoi(argv[4]), argv[5], argv[6]);
  if (m == "verify" && argc >= 5) return cmd_verify(argc, argv);
  if (m == "verifymany" && argc == 4) return cmd_verifymany(argv[2], argv[3]);
  if (m == "hexblobs" && argc == 3) return cmd_hexblobs(argv[2]);
  if (m == "bench_prover" && argc >= 4) return cmd_bench_prover(argv[2], argv[3]);
  fprintf(stderr, "bad arguments\n"); return 2; }
