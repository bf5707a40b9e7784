// Groth16 host code, part 2 of 4: randomness, evaluation domains, the key generator.
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
HFr random_fr() {
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
size_t domain_size_for(size_t min_size) { return domain_shape(min_size).m; }
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

HG2 default_g2_generator() {
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

}  // namespace zk
