// Groth16 host code, part 4 of 4: the verifier (host pairing check, the GPU verifier's schedule interpreted on the host, the device trace), proof encoding.
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
