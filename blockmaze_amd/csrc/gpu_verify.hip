// Host driver of the batched GPU verifier (kernel K9, pairing.cuh): assembles the bytecode and the per-key tables once, then checks any number of proofs per
// launch.
#include <atomic>
#include <condition_variable>
#include <exception>
#include <cstring>
#include <memory>
#include <mutex>
#include "gpu_internal.hpp"
#include "pairing.cuh"
#include "pairing_host.hpp"
#include "verify_sched.hpp"

namespace zk {
using host::HFq; using host::HFq2; using host::HG1;

static const uint64_t BN_Z = 4965661367192848881ull;                         // alt_bn128_init.cpp:327 (final_exponent_z)
static const uint64_t ATE_LOOP[2] = {0x9d797039be763ba8ull, 0x1ull};          // 6z+2 (alt_bn128_init.cpp:324)

// typed view of a byte allocation (DevBuf is instantiated for the raw interface types only)
template <class T> struct DevArr { DevBuf<uint8_t> b; DevArr() = default; explicit DevArr(size_t n) : b(n * sizeof(T)) {} T *get() const { return (T *)b.get();
    }
  void upload(const T *h, size_t n) { b.upload((const uint8_t *)h, n * sizeof(T)); } };
// acc_i = IC[0] + sum_j inputs[i][j] * IC[j+1] by ONE WORKGROUP of 64 quads per proof, the points SPREAD over the quads (htail29.cuh: lane k of a quad holds coordinate k of
// (X, Y, ZZ, ZZZ) on nine 29-bit limbs, an addition is four rounds of ONE product per lane).  The (input, window) pairs — 128 for a send statement — are dealt to the
// quads, the quads' sums meet in the workgroup's tree (four levels by shuffles, two through LDS), quad 0 adds IC[0] and hands the sum over as (x w, -y w, w), w = ZZ ZZZ:
// no inversion (verify_sched.hpp evaluates the gamma lines times w).  Round 6: the dependent chain is 2 + 6 + 1 additions of 4 products instead of the one-wave kernel's
// 2 + 6 + 1 of 10 to 14 in single lanes (0.074 ms of a verification's 0.85; now 0.025: `verifySendproof` 0.86 -> 0.81 ms a call, 8,400 -> 9,200 /s from 8 threads).
// The additions are the incomplete ones of the H tail: an operand equal to +-the other leaves ZZ = 0 (mod p), w = 0, and the schedule kernel hands the proof to the host
// verifier (verdict 2) — with canonical inputs the partial sums are multiples of ONE point by distinct integers below r or involve different points of the key, so this
// takes a discrete logarithm (or inputs that are not reduced: the host decides those).  tables261 / ic0_261: coordinates times 2^261, canonical 8-word form.
static __global__ void __launch_bounds__(256) k_verify_acc_quads(const Affine<Fq> *__restrict__ tables261, Affine<Fq> ic0_261, const Fr *__restrict__ inputs,
    uint32_t n_inputs, uint32_t n, NegAcc3 *__restrict__ acc_out) {
  __shared__ Point29Rec lds[4];
  const uint32_t i = blockIdx.x, q = threadIdx.x >> 2; const int k = threadIdx.x & 3; if (i >= n) return;
  __builtin_amdgcn_s_setprio(3);
  auto lift = [&](const Affine<Fq> &pt, bool absent) {                                     // an affine point as a spread (x, y, 1, 1); (0, 0) and a zero digit: the point at infinity
    QPoint29 r; uint32_t o = 0;
    const Fq &cw = k == 0 ? pt.x : pt.y;
#pragma unroll
    for (int t = 0; t < 8; t++) o |= cw.l[t];
    const uint32_t ox = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x00, 0xf, 0xf, false), oy = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o, 0x55, 0xf, 0xf, false);
    r.c = k < 2 ? Fq29::unpack(cw.l) : Fq29::one(); r.inf = absent || (ox | oy) == 0;
    return r;
  };
  QPoint29 acc = quad29_inf(); bool first = true;
#pragma unroll 1
  for (uint32_t pair = q; pair < n_inputs * 32; pair += 64) {
    const uint32_t j = pair >> 5, w = pair & 31; const Fr kj = inputs[(size_t)i * n_inputs + j]; const uint32_t d = (kj.l[w >> 2] >> ((w & 3) * 8)) & 0xffu;
    Affine<Fq> pt{Fq::zero(), Fq::zero()}; if (k < 2) { const Affine<Fq> *src = tables261 + ((size_t)j * 32 * 255 + w * 255 + (d ? d - 1 : 0)); if (k == 0) pt.x = src->x; else pt.y = src->y; }
    const QPoint29 cur = lift(pt, d == 0);
    if (first) { acc = cur; first = false; } else acc = quad29_add(acc, cur, k);
  }
  acc = block_quad29_tree(acc, lds, min(n_inputs * 32, 64u));
  if (q != 0) return;
  acc = quad29_add(acc, lift(ic0_261, false), k);
  // (x w, -y w, w) = (X ZZZ, -Y ZZ, ZZ ZZZ): one product per lane with the partner's coordinate [3, 2, 3, 3]
  Fq29 m = Fq29::mul(acc.c, quad29_perm<0xFB>(acc.c));
  if (k == 1) m = Fq29::neg_product(m);
  Fq out = Fq::zero();
  if (!acc.inf) { m.to_words(out.l); out = out.normalize(); }
  if (k < 3) reinterpret_cast<Fq *>(acc_out + i)[k] = out;
}

// One small verification in flight: its own stream, pinned staging and device buffers for up to CTX_CAP proofs — go-ethereum's verifyXproof calls arrive one
// proof at a time from many threads, and a proof occupies ONE compute unit for ~2 ms: several contexts let them overlap instead of queueing behind one stream.
struct VerifyCtx { std::mutex m; hipStream_t s = nullptr; uint8_t *h = nullptr /* pinned, mapped */, *hd = nullptr /* the device's view of h */; DevBuf<uint8_t> d; };
struct BatchVerifier::Impl {
  size_t n_inputs = 0;
  DevBuf<uint32_t> sched_prog, sched_consts;
  SchedInfo si{};
  DevBuf<uint32_t> prog;
  DevArr<EllCoeffsDev> gamma, delta;
  DevArr<FrobeniusDev> frob;
  DevArr<Fq12> alpha_beta;
  DevArr<Affine<Fq>> tables;
  Affine<Fq> ic0, ic0_261;                                                     // (261: coordinates times 2^261, the radix of the 29-bit limbs: k_verify_acc_quads)
  DevArr<Affine<Fq>> tables261;
  VerifyConsts K;
  size_t prog_len = 0;
  static constexpr size_t CTX_CAP = 64; std::vector<std::unique_ptr<VerifyCtx>> ctxs; std::atomic<unsigned> next_ctx{0}; std::mutex big; size_t lds = 0;
  // Small calls that meet are ONE launch (go-ethereum's verifyXproof calls arrive one proof at a time from many goroutines; a launch takes the same 0.8 ms for 1 or
  // 64 proofs): a caller either joins the batch that is waiting for a launch slot or opens one and leads it; a leader launches as soon as fewer than
  // ZK_VERIFY_STREAMS (16) launches of this key are under way — alone if nobody came —, so a lone caller never waits for company.
  // Independent launches overlap on the device (a proof occupies one CU), so sharing only sets in when callers outnumber the slots: measured from C threads through
  // verifySendproof (tools/verify_threads.sh): 1 / 8 / 16 threads 1,180 / 8,700 / 14,700 verifications/s (2 streams: 1,180 / 5,390 / 10,400).
  struct Pending { const void *proofs; const Fe32 *inputs; size_t n; uint8_t *ok; bool done = false; std::exception_ptr err; };
  int max_in_flight() const { return (int)ctxs.size(); }      // as many launches under way as there are streams (ZK_VERIFY_STREAMS, default 16)
  std::mutex cm; std::condition_variable ccv; std::vector<Pending *> open; size_t open_n = 0; bool open_led = false; int in_flight = 0;
  std::atomic<uint64_t> launches{0}, calls{0};
  // layout of a context's staging area (host and device alike): proofs | inputs | -acc | verdicts
  size_t off_in() const {
    return CTX_CAP * sizeof(VerifyItem);
  }
  size_t off_acc() const {
    return off_in() + CTX_CAP * (n_inputs + 1) * sizeof(Fe32);
  }
  size_t off_ok() const {
    return off_acc() + CTX_CAP * sizeof(NegAcc3);
  }
  size_t ctx_bytes() const {
    return off_ok() + CTX_CAP;
  }
  ~Impl() { for (auto &c : ctxs) { if (c->h) hipHostFree(c->h); if (c->s) hipStreamDestroy(c->s); } }
};

// registers of the program
enum { rF = 0, rT = 1, r2 = 2, r3 = 3, r4 = 4, r5 = 5, r6 = 6, r7 = 7, r8 = 8, r9 = 9, rONE = 10 };
// exp_by_neg_z (alt_bn128_pairing.cpp:84-96): conj(src^z) for src in the cyclotomic subgroup.  Width-3 NAF of z (18 non-zero digits in {+-1, +-3} instead of 28
// one bits; a negative digit costs nothing because the inverse in the cyclotomic subgroup is the conjugate) over cyclotomic squarings; any addition chain gives
// the same group element as libff's binary cyclotomic_exp (fp12_2over3over2.tcc:337-365).  Uses r9 (accumulator), r11 (src^3), rT (scratch).
enum { r11 = 11 };
static void emit_exp_neg_z(std::vector<uint32_t> &p, uint32_t dst, uint32_t src) {
  std::vector<int> dig;
  for (uint64_t k = BN_Z; k;) {
    int t = 0;
    if (k & 1) {
      t = (int)(k & 7);
      if (t >= 4) t -= 8;
      k -= (uint64_t)(int64_t)t;
    }
    dig.push_back(t);
    k >>= 1;
  }
  p.push_back(vm_ins(VM_CYCSQR, rT, src, 0)); p.push_back(vm_ins(VM_MUL, r11, rT, src)); bool started = false;
  for (size_t i = dig.size(); i-- > 0;) { int d = dig[i];
    if (started) p.push_back(vm_ins(VM_CYCSQR, r9, r9, 0));
    if (!d) continue; uint32_t reg = (d == 1 || d == -1) ? src : r11;
    if (!started) { if (d < 0) throw GpuError("verify: leading NAF digit"); p.push_back(vm_ins(VM_MUL, r9, rONE, reg)); started = true; continue; }
    if (d < 0) { p.push_back(vm_ins(VM_CONJ, rT, reg, 0)); reg = rT; }
    p.push_back(vm_ins(VM_MUL, r9, r9, reg)); }
  p.push_back(vm_ins(VM_CONJ, dst, r9, 0));
}
static std::vector<uint32_t> assemble_program(size_t n_lines) {
  std::vector<uint32_t> p; uint32_t idx = 0; bool found = false;
  auto lines = [&] {
    p.push_back(vm_ins(VM_MUL024, rF, rF, rT));
    p.push_back(vm_ins(VM_LINE, rT, 1, idx));
    p.push_back(vm_ins(VM_MUL024, rF, rF, rT));
    p.push_back(vm_ins(VM_LINE, rT, 2, idx));
    p.push_back(vm_ins(VM_MUL024, rF, rF, rT));
    idx++;
  };
  p.push_back(vm_ins(VM_ONE, rF, 0, 0)); p.push_back(vm_ins(VM_ONE, rONE, 0, 0));
  // miller_loop :368-418 / precomputation :305-366
  for (int i = 127; i >= 0; i--) {
    bool bit = (ATE_LOOP[i / 64] >> (i % 64)) & 1;
    if (!found) {
      found |= bit;
      continue;
    }
    p.push_back(vm_ins(VM_MUL, rF, rF, rF)); p.push_back(vm_ins(VM_DBL, rT, 0, 0)); lines();
    if (bit) { p.push_back(vm_ins(VM_ADD, rT, 0, 0)); lines(); } }
  p.push_back(vm_ins(VM_ADD, rT, 0, 1)); lines(); p.push_back(vm_ins(VM_ADD, rT, 0, 2)); lines();
  if (idx != n_lines || idx > 255) throw GpuError("verify: line count");
  // final_exponentiation :110-238 (first chunk :110-129, last chunk :131-238)
  // C0 = conj(f) * f^-1
  p.push_back(vm_ins(VM_CONJ, r3, rF, 0));
  p.push_back(vm_ins(VM_INV, r4, rF, 0));
  p.push_back(vm_ins(VM_MUL, r3, r3, r4));
  // first = C0^(q^2) * C0
  p.push_back(vm_ins(VM_FROB, r4, r3, 2));
  p.push_back(vm_ins(VM_MUL, r2, r4, r3));
  emit_exp_neg_z(p, r3, r2);                                                                                                                   // A
  // B = A^2, C = B^2, D = C*B
  p.push_back(vm_ins(VM_CYCSQR, r4, r3, 0));
  p.push_back(vm_ins(VM_CYCSQR, r5, r4, 0));
  p.push_back(vm_ins(VM_MUL, r6, r5, r4));
  emit_exp_neg_z(p, r7, r6);                                                                                                                   // E
  p.push_back(vm_ins(VM_CYCSQR, r5, r7, 0)); emit_exp_neg_z(p, r8, r5);                                                                          // F = E^2, G
  // H = conj(D), I = conj(G)
  p.push_back(vm_ins(VM_CONJ, r6, r6, 0));
  p.push_back(vm_ins(VM_CONJ, r8, r8, 0));
  // J = I*E, K = J*H
  p.push_back(vm_ins(VM_MUL, r8, r8, r7));
  p.push_back(vm_ins(VM_MUL, r8, r8, r6));
  // L = K*B, M = K*E, N = M*first
  p.push_back(vm_ins(VM_MUL, r6, r8, r4));
  p.push_back(vm_ins(VM_MUL, r7, r8, r7));
  p.push_back(vm_ins(VM_MUL, r7, r7, r2));
  // O = L^q, P = O*N
  p.push_back(vm_ins(VM_FROB, r4, r6, 1));
  p.push_back(vm_ins(VM_MUL, r4, r4, r7));
  // Q = K^(q^2), R = Q*P
  p.push_back(vm_ins(VM_FROB, r5, r8, 2));
  p.push_back(vm_ins(VM_MUL, r4, r5, r4));
  // S = conj(first), T = S*L, U = T^(q^3)
  p.push_back(vm_ins(VM_CONJ, r2, r2, 0));
  p.push_back(vm_ins(VM_MUL, r2, r2, r6));
  p.push_back(vm_ins(VM_FROB, r2, r2, 3));
  p.push_back(vm_ins(VM_MUL, rF, r2, r4)); p.push_back(vm_ins(VM_END, 0, 0, 0)); return p;                                                     // result = U*R
}
template <class T, class H> static T to_dev(const H &h) { static_assert(sizeof(T) == sizeof(H), "layout"); T t; memcpy(&t, &h, sizeof(T)); return t; }

BatchVerifier::BatchVerifier(const host::HFq12 &alpha_g1_beta_g2, const G2AffineRaw &gamma_g2, const G2AffineRaw &delta_g2, const G1AffineRaw *ic,
    size_t n_ic) : impl(new Impl) {
  Impl &d = *impl; if (n_ic < 1 || n_ic > 17) throw GpuError("verify: IC size"); d.n_inputs = n_ic - 1;
  auto fq2_of = [](const Fe32 &a, const Fe32 &b) { HFq2 r; memcpy(r.c0.l, &a, 32); memcpy(r.c1.l, &b, 32); return r; };
  host::G2Precomp pg = host::precompute_g2(fq2_of(gamma_g2.x0, gamma_g2.x1), fq2_of(gamma_g2.y0, gamma_g2.y1)), pd = host::precompute_g2(fq2_of(delta_g2.x0,
      delta_g2.x1), fq2_of(delta_g2.y0, delta_g2.y1));
  std::vector<EllCoeffsDev> lg(pg.size()), ld(pd.size());
  for (size_t i = 0; i < pg.size(); i++) {
    lg[i] = to_dev<EllCoeffsDev>(pg[i]);
    ld[i] = to_dev<EllCoeffsDev>(pd[i]);
  }
  d.gamma = DevArr<EllCoeffsDev>(lg.size());
  d.gamma.upload(lg.data(), lg.size());
  d.delta = DevArr<EllCoeffsDev>(ld.size());
  d.delta.upload(ld.data(), ld.size());
  { vsched::Schedule sc = vsched::build(alpha_g1_beta_g2, pg, pd);   // the workgroup-per-proof kernel's schedule (verify_sched.hpp)
    const std::vector<uint32_t> c29 = vsched::consts29(sc);
    d.sched_prog = DevBuf<uint32_t>(sc.prog.size());      // (padded by the builder: the kernel's ring reads PREFETCH_ROUNDS rounds ahead)
    d.sched_prog.upload(sc.prog.data(), sc.prog.size());
    d.sched_consts = DevBuf<uint32_t>(c29.size());
    d.sched_consts.upload(c29.data(), c29.size());
    d.si.n_rounds_padded = sc.n_rounds_padded;
    d.si.n_slots = sc.n_slots;
    d.si.n_consts = (uint32_t)sc.consts.size();
    for (int k = 0; k < vsched::N_OUT; k++) d.si.out_slot[k] = sc.out_slot[k];
    d.lds = ((size_t)sc.n_slots + sc.consts.size()) * l29::STRIDE * 4 + (size_t)vsched::PREFETCH_ROUNDS * 256 * 16;   // values, constants, the ring of instruction words
    if (d.lds > 160 * 1024) throw GpuError("verify: the schedule needs more LDS than a CU has");
    static std::atomic<uint64_t> attr_done{0};
    zk_raise_dynamic_lds((const void *)k_verify_sched29, 160 * 1024, attr_done);
  }
  { static const int n_ctx = [] { const char *e = getenv("ZK_VERIFY_STREAMS"); int v = e ? atoi(e) : 16; return v < 1 ? 1 : v > 32 ? 32 : v; }();
    for (int k = 0; k < n_ctx; k++) {
      std::unique_ptr<VerifyCtx> c(new VerifyCtx);
      HIP_CHECK(hipStreamCreateWithFlags(&c->s, hipStreamNonBlocking));
      HIP_CHECK(hipHostMalloc((void **)&c->h, d.ctx_bytes(), hipHostMallocMapped));
      HIP_CHECK(hipHostGetDevicePointer((void **)&c->hd, c->h, 0));
      c->d = DevBuf<uint8_t>(d.ctx_bytes());
      d.ctxs.push_back(std::move(c));
    }
  }
  std::vector<uint32_t> prog = assemble_program(pg.size());
  d.prog_len = prog.size();
  d.prog = DevBuf<uint32_t>(prog.size());
  d.prog.upload(prog.data(), prog.size());
  FrobeniusDev fr = to_dev<FrobeniusDev>(host::frobenius_tables()); d.frob = DevArr<FrobeniusDev>(1); d.frob.upload(&fr, 1);
  Fq12 ab = to_dev<Fq12>(alpha_g1_beta_g2); d.alpha_beta = DevArr<Fq12>(1); d.alpha_beta.upload(&ab, 1);
  HFq2 tb = HFq2{HFq::from_u64(3), HFq::zero()} * HFq2{HFq::from_u64(9), HFq::one()}.inv();
  d.K.twist_b = to_dev<Fq2>(tb);
  d.K.two_inv = to_dev<Fq>(HFq::from_u64(2).inv());
  memcpy(&d.ic0, &ic[0], sizeof(G1AffineRaw));
  // window tables of IC[1..]: table[j][w*255 + (dgt-1)] = dgt * 2^(8w) * IC[j+1], affine; one batch inversion per point
  std::vector<G1AffineRaw> tab(d.n_inputs * 32 * 255 + 1);
  for (size_t j = 0; j < d.n_inputs; j++) {
    HFq x, y;
    memcpy(x.l, &ic[j + 1].x, 32);
    memcpy(y.l, &ic[j + 1].y, 32);
    HG1 wbase = (x.is_zero() && y.is_zero()) ? HG1::inf() : HG1::from_affine(x, y);
    std::vector<HG1> pts(32 * 255);
    for (int w = 0; w < 32; w++) { HG1 acc = wbase; for (int dg = 1; dg <= 255; dg++) { pts[w * 255 + dg - 1] = acc; acc = acc.add(wbase); } wbase = acc; }
    std::vector<HFq> pre(pts.size());
    HFq run = HFq::one();
    for (size_t k = 0; k < pts.size(); k++) {
      pre[k] = run;
      if (!pts[k].is_inf()) run = run * pts[k].Z;
    }
    HFq inv = run.inv();
    for (size_t k = pts.size(); k-- > 0;) { G1AffineRaw &o = tab[j * 32 * 255 + k]; if (pts[k].is_inf()) { memset(&o, 0, sizeof o); continue; }
      HFq zi = inv * pre[k];
      inv = inv * pts[k].Z;
      HFq z2 = zi.sqr(), ax = pts[k].X * z2, ay = pts[k].Y * z2 * zi;
      memcpy(&o.x, ax.l, 32);
      memcpy(&o.y, ay.l, 32);
    }
  }
  d.tables = DevArr<Affine<Fq>>(tab.size()); d.tables.upload((const Affine<Fq> *)tab.data(), tab.size());
  { HFq c261; memcpy(c261.l, FQ_TWO261, 32);                                              // x 2^256 times the plain integer 2^261 mod q, Montgomery product: x 2^261
    auto to261 = [&](const G1AffineRaw &a) { G1AffineRaw o; HFq x, y; memcpy(x.l, &a.x, 32); memcpy(y.l, &a.y, 32); x = x * c261; y = y * c261; memcpy(&o.x, x.l, 32); memcpy(&o.y, y.l, 32); return o; };
    std::vector<G1AffineRaw> t261(tab.size()); for (size_t k = 0; k < tab.size(); k++) t261[k] = to261(tab[k]);
    d.tables261 = DevArr<Affine<Fq>>(t261.size()); d.tables261.upload((const Affine<Fq> *)t261.data(), t261.size());
    const G1AffineRaw i261 = to261(ic[0]); memcpy(&d.ic0_261, &i261, sizeof i261); }
}
BatchVerifier::~BatchVerifier() = default;
static void launch_sched(BatchVerifier::Impl &d, unsigned n, hipStream_t s, const VerifyItem *items, const NegAcc3 *acc, uint8_t *ok, uint32_t *trace = nullptr,
    uint32_t trace_every = 1) {
  hipLaunchKernelGGL(k_verify_sched29, dim3(n), dim3(256), d.lds, s, (const uint4 *)d.sched_prog.get(), (const uint4 *)d.sched_consts.get(), items, acc, (uint32_t)n, d.si, ok,
      trace, trace_every);
}
size_t BatchVerifier::num_inputs() const { return impl->n_inputs; }
size_t BatchVerifier::program_length() const { return impl->prog_len; }
void BatchVerifier::counters(uint64_t out[2]) const { out[0] = impl->calls.load(); out[1] = impl->launches.load(); }
// test entry: ONE proof through the kernels with the values of every `every`-th round written out; nacc_out: the 96 bytes the accumulation kernel handed to the
// schedule kernel (x w, -y w, w; Montgomery)
uint8_t BatchVerifier::trace(const void *proof_mont, const Fe32 *inputs_canonical, uint32_t every, std::vector<uint32_t> &values, uint8_t nacc_out[96]) {
  Impl &d = *impl; hipStream_t s = gpu().stream; if (!every) throw GpuError("verify trace: every");
  DevArr<VerifyItem> items(1); DevBuf<Fe32> in(d.n_inputs + 1); DevArr<NegAcc3> acc3(1); DevBuf<uint8_t> out(1);
  items.upload((const VerifyItem *)proof_mont, 1); if (d.n_inputs) in.upload(inputs_canonical, d.n_inputs);
  const size_t dumps = d.si.n_rounds_padded / every + 1, words = (size_t)d.si.n_slots * l29::STRIDE; DevBuf<uint32_t> tr(dumps * words);
  hipLaunchKernelGGL(k_verify_acc_quads, dim3(1), dim3(256), 0, s, (const Affine<Fq> *)d.tables261.get(), d.ic0_261, (const Fr *)in.get(), (uint32_t)d.n_inputs, 1u, acc3.get());
  launch_sched(d, 1, s, items.get(), acc3.get(), out.get(), tr.get(), every);
  HIP_CHECK(hipStreamSynchronize(s)); HIP_CHECK(hipGetLastError());
  values.resize(dumps * words); tr.download(values.data(), values.size()); uint8_t ok = 0; out.download(&ok, 1);
  HIP_CHECK(hipMemcpy(nacc_out, acc3.get(), 96, hipMemcpyDeviceToHost)); return ok;
}
void BatchVerifier::verify(const void *proofs_mont, const Fe32 *inputs_canonical, size_t n, uint8_t *ok) {
  // (gpu() also selects the device for this thread)
  if (!n) return;
  Impl &d = *impl;
  static_assert(sizeof(VerifyItem) == 256, "proof record");
  hipStream_t s = gpu().stream;
  // a few proofs: one of the small contexts, nothing allocated; calls that meet share a launch (Impl::Pending)
  if (n <= Impl::CTX_CAP) {
    Impl::Pending me{proofs_mont, inputs_canonical, n, ok}; d.calls.fetch_add(1, std::memory_order_relaxed);
    std::unique_lock<std::mutex> lk(d.cm);
    for (;;) {
      if (d.open_led && d.open_n + n <= Impl::CTX_CAP) {                       // join the batch that is waiting for its slot
        d.open.push_back(&me); d.open_n += n; d.ccv.wait(lk, [&] { return me.done; });
        if (me.err) std::rethrow_exception(me.err);
        return;
      }
      if (!d.open_led) break;
      d.ccv.wait(lk);                                                            // (the waiting batch is full: until it has left)
    }
    d.open.assign(1, &me); d.open_n = n; d.open_led = true;
    d.ccv.wait(lk, [&] { return d.in_flight < d.max_in_flight(); });
    std::vector<Impl::Pending *> batch; batch.swap(d.open); const size_t total = d.open_n; d.open_n = 0; d.open_led = false; d.in_flight++;
    lk.unlock(); d.ccv.notify_all();
    std::exception_ptr err;
    try {
      VerifyCtx *c = nullptr;
      for (size_t k = 0; k < d.ctxs.size() && !c; k++) {
        VerifyCtx *t = d.ctxs[(d.next_ctx.fetch_add(1) + k) % d.ctxs.size()].get();
        if (t->m.try_lock()) c = t;
      }
      if (!c) { c = d.ctxs[d.next_ctx.fetch_add(1) % d.ctxs.size()].get(); c->m.lock(); }
      std::lock_guard<std::mutex> ck(c->m, std::adopt_lock);
      // the records go into pinned memory the device reads in place (a few hundred bytes a proof: no copy command), the verdicts come back the same way
      size_t at = 0;
      for (Impl::Pending *q : batch) { memcpy(c->h + at * sizeof(VerifyItem), q->proofs, q->n * sizeof(VerifyItem));
        if (d.n_inputs) memcpy(c->h + d.off_in() + at * d.n_inputs * sizeof(Fe32), q->inputs, q->n * d.n_inputs * sizeof(Fe32));
        at += q->n; }
      uint8_t *dv = c->d.get(); volatile uint8_t *hok = c->h + d.off_ok(); for (size_t k = 0; k < total; k++) hok[k] = 0xff;
      { Stage st("verify.batch", c->s);
        hipLaunchKernelGGL(k_verify_acc_quads, dim3((unsigned)total), dim3(256), 0, c->s, (const Affine<Fq> *)d.tables261.get(), d.ic0_261, (const Fr *)(c->hd + d.off_in()),
            (uint32_t)d.n_inputs, (uint32_t)total, (NegAcc3 *)(dv + d.off_acc()));
        launch_sched(d, (unsigned)total, c->s, (const VerifyItem *)c->hd, (const NegAcc3 *)(dv + d.off_acc()), c->hd + d.off_ok());
      }
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(c->s));
      d.launches.fetch_add(1, std::memory_order_relaxed);
      at = 0;
      for (Impl::Pending *q : batch) { for (size_t k = 0; k < q->n; k++) { const uint8_t v = hok[at + k]; if (v > 2) throw GpuError("verify: a verdict did not arrive"); q->ok[k] = v; } at += q->n; }
    } catch (...) { err = std::current_exception(); }
    lk.lock(); d.in_flight--;
    for (Impl::Pending *q : batch) if (q != &me) { q->err = err; q->done = true; }
    lk.unlock(); d.ccv.notify_all();
    if (err) std::rethrow_exception(err);
    return;
  }
  std::lock_guard<std::mutex> lk(d.big);
  DevArr<VerifyItem> items(n); DevBuf<Fe32> in(n * d.n_inputs + 1); DevArr<Affine<Fq>> acc(n); DevBuf<uint8_t> out(n);
  items.upload((const VerifyItem *)proofs_mont, n); if (d.n_inputs) in.upload(inputs_canonical, n * d.n_inputs);
  Stage st("verify.batch");
  // up to WAVE_MAX proofs: one workgroup each (a proof's latency is the schedule's ~900 rounds; two workgroups share a CU: 0.9 ms for 256 proofs, 1.7 ms for 512,
  // 24.5 ms for 8,192 — 330 K proofs/s); beyond that the lane-per-proof kernel, whose 26 ms floor is then amortised over more (profiles/r06_verify_batch.txt)
  static const size_t wave_max = [] { const char *e = getenv("ZK_VERIFY_WAVE_MAX"); long v = e ? atol(e) : 8192; return (size_t)(v < 0 ? 0 : v); }();
  if (n <= wave_max) { DevArr<NegAcc3> acc3(n);
    hipLaunchKernelGGL(k_verify_acc_quads, dim3((unsigned)n), dim3(256), 0, s, (const Affine<Fq> *)d.tables261.get(), d.ic0_261, (const Fr *)in.get(),
        (uint32_t)d.n_inputs, (uint32_t)n, acc3.get());
    // (acc3 lives until the kernels are done)
    launch_sched(d, (unsigned)n, s, items.get(), acc3.get(), out.get());
    HIP_CHECK(hipStreamSynchronize(s));
  } else {
    hipLaunchKernelGGL(k_verify_acc, dim3(cdiv(n, 64)), dim3(64), 0, s, (const Affine<Fq> *)d.tables.get(), d.ic0, (const Fr *)in.get(), (uint32_t)d.n_inputs,
        (uint32_t)n, acc.get());
    hipLaunchKernelGGL(k_verify_batch, dim3(cdiv(n, 64)), dim3(64), 0, s, d.prog.get(), items.get(), acc.get(), d.gamma.get(), d.delta.get(), d.frob.get(),
        d.alpha_beta.get(), d.K, (uint32_t)n, out.get());
  }
  HIP_CHECK(hipGetLastError()); out.download(ok, n);
}

}  // namespace zk
