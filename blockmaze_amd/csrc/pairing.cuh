// Batched Groth16 verification on the GPU (kernel K9). Two kernels: k_verify_sched29 (round 3, at the end of this file) — one
// 256-thread workgroup per proof interpreting the operation schedule of verify_sched.hpp on 29-bit limbs, 2.1 ms per launch — for one proof up to a few
// thousand, and the first generation below, one LANE per proof, whose 25 ms floor only pays from several thousand proofs on.
//
// Restates r1cs_gg_ppzksnark_verifier_strong_IC (SNARK/.../r1cs_gg_ppzksnark.tcc:509-623) over libff's optimal-ate pairing
// (FF/algebra/curves/alt_bn128/alt_bn128_pairing.cpp: doubling / mixed-addition steps :242-293, G2 precomputation :305-366, miller_loop :368-418,
// final_exponentiation :110-238 — the exact chain, so that the GT value compares equal to the vk's alpha_g1_beta_g2) on the tower
// Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-(9+u)), Fq12 = Fq6[w]/(w^2-v) (fp6_3over2.tcc, fp12_2over3over2.tcc).
//
// GPU shape: a proof's check is ~30,000 dependent field products, far too long a chain to inline call by call, and hipcc's out-of-line device functions
// are not trustworthy with the inline-asm product (curve.cuh).  The whole check is therefore a small BYTECODE program (about 1,000 instructions: MUL,
// CONJ, FROB, INV on a file of Fq12 registers in per-lane memory, plus the G2 line steps) that the host assembles once per verifying key and every lane
// interprets in lock step: one inlined copy of each primitive, no divergence (the program does not depend on the proof), any batch size.
// The three Miller loops of the reference — e(A,B), and the double loop e(acc,gamma)·e(C,delta) that is conjugated before the final exponentiation —
// run as ONE accumulator with acc and C negated: f1·conj(f2·f3) and f1·f(-acc)·f(-C) have the same final exponentiation.
#pragma once
#include <hip/hip_runtime.h>
#include "curve.cuh"
#include "field29.cuh"
#include "verify_sched.hpp"

namespace zk {

__device__ __forceinline__ Fq shfl_down_fq(const Fq &v, int delta) { Fq r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = __shfl_down(v.l[i], delta, 64);
  return r; }
struct Fq6 {
  Fq2 c0, c1, c2;
  static __device__ __forceinline__ Fq6 zero() { return {Fq2::zero(), Fq2::zero(), Fq2::zero()}; }
  static __device__ __forceinline__ Fq6 one() { return {Fq2::one(), Fq2::zero(), Fq2::zero()}; }
  friend __device__ __forceinline__ Fq6 operator+(const Fq6 &a, const Fq6 &b) { return {a.c0 + b.c0, a.c1 + b.c1, a.c2 + b.c2}; }
  friend __device__ __forceinline__ Fq6 operator-(const Fq6 &a, const Fq6 &b) { return {a.c0 - b.c0, a.c1 - b.c1, a.c2 - b.c2}; }
  __device__ __forceinline__ Fq6 neg() const { return {c0.neg(), c1.neg(), c2.neg()}; }
  friend __device__ __forceinline__ Fq6 operator*(const Fq6 &a, const Fq6 &b) {   // Karatsuba, fp6_3over2.tcc:94-108
    Fq2 aA = a.c0 * b.c0, bB = a.c1 * b.c1, cC = a.c2 * b.c2;
    return {aA + ((a.c1 + a.c2) * (b.c1 + b.c2) - bB - cC).mul_xi(), (a.c0 + a.c1) * (b.c0 + b.c1) - aA - bB + cC.mul_xi(),
        (a.c0 + a.c2) * (b.c0 + b.c2) - aA + bB - cC};
  }
  __device__ __forceinline__ Fq6 mul_by_v() const { return {c2.mul_xi(), c0, c1}; }
  __device__ __forceinline__ bool operator==(const Fq6 &o) const { return c0 == o.c0 && c1 == o.c1 && c2 == o.c2; }
};
struct Fq12 { Fq6 c0, c1; __device__ __forceinline__ bool operator==(const Fq12 &o) const { return c0 == o.c0 && c1 == o.c1; } };

struct FrobeniusDev { Fq2 fq6_c1[6], fq6_c2[6], fq12_c1[12], twist_mul_by_q_x, twist_mul_by_q_y; };   // same layout as host::FrobeniusTables
struct EllCoeffsDev { Fq2 ell_0, ell_VW, ell_VV; };
// Montgomery form, as parsed from the 512 hex characters
struct VerifyItem { Affine<Fq> A; Affine<Fq2> B; Affine<Fq> C; };

enum VmOp : uint32_t { VM_MUL = 0, VM_CONJ, VM_FROB, VM_INV, VM_ONE, VM_DBL, VM_ADD, VM_LINE, VM_END, VM_MUL024, VM_CYCSQR };
__host__ __device__ inline uint32_t vm_ins(uint32_t op, uint32_t d, uint32_t a, uint32_t b) { return op | d << 8 | a << 16 | b << 24; }
constexpr int VM_REGS = 12;

struct VerifyConsts { Fq2 twist_b; Fq two_inv; };

// acc_i = IC[0] + sum_j inputs[i][j] * IC[j+1] from 8-bit window tables (table[j][w*255 + d-1] = d * 2^(8w) * IC[j+1]); output affine with y NEGATED
// (the pairing uses -acc), all-zero record if acc is the point at infinity.  inputs canonical.  r1cs_gg_ppzksnark.tcc:584-590 (strong IC).
__global__ void __launch_bounds__(64) k_verify_acc(const Affine<Fq> *__restrict__ tables, Affine<Fq> ic0, const Fr *__restrict__ inputs, uint32_t n_inputs,
    uint32_t n, Affine<Fq> *__restrict__ acc_out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; XYZZ<Fq> acc = XYZZ<Fq>::from_affine(ic0);
  for (uint32_t j = 0; j < n_inputs; j++) { Fr k = inputs[(size_t)i * n_inputs + j]; const Affine<Fq> *t = tables + (size_t)j * 32 * 255;
    for (int w = 0; w < 32; w++) { uint32_t d = (k.l[w >> 2] >> ((w & 3) * 8)) & 0xffu; if (d) acc.madd_inl(t[w * 255 + d - 1]); } }
  if (acc.is_inf()) { acc_out[i] = Affine<Fq>::inf(); return; }
  Fq zi = acc.ZZ.inv(), z3i = acc.ZZZ.inv(); acc_out[i] = {acc.X * zi, (acc.Y * z3i).neg()};
}

__device__ __forceinline__ Fq6 fq6_frob(const Fq6 &a, unsigned p, const FrobeniusDev &t) {
  return {a.c0.frob(p), t.fq6_c1[p % 6] * a.c1.frob(p), t.fq6_c2[p % 6] * a.c2.frob(p)};
}
__device__ __forceinline__ Fq6 fq6_mul_fq2(const Fq6 &a, const Fq2 &k) { return {a.c0 * k, a.c1 * k, a.c2 * k}; }
// the operand of mul_by_024
__device__ __forceinline__ Fq12 fq12_sparse(const Fq2 &ell_0, const Fq2 &ell_VW, const Fq2 &ell_VV) {
  return {{ell_0, Fq2::zero(), ell_VV}, {Fq2::zero(), ell_VW, Fq2::zero()}};
}

// prog: the bytecode.  lines[0] / lines[1]: precomputed G2 line coefficients of the vk's gamma / delta.  ok[i] = 1 if proof i is accepted.
__global__ void __launch_bounds__(64) k_verify_batch(const uint32_t *__restrict__ prog, const VerifyItem *__restrict__ items,
    const Affine<Fq> *__restrict__ neg_acc,
                                                    const EllCoeffsDev *__restrict__ gamma_lines, const EllCoeffsDev *__restrict__ delta_lines,
                                                        const FrobeniusDev *__restrict__ frob,
                                                    const Fq12 *__restrict__ alpha_beta, VerifyConsts K, uint32_t n, uint8_t *__restrict__ ok) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  const VerifyItem it = items[i]; const Affine<Fq> nacc = neg_acc[i]; const FrobeniusDev &T = *frob;
  // is_well_formed (on-curve only, alt_bn128_g1.cpp:92-117, alt_bn128_g2.cpp:98-127); a proof parsed from hex has Z = 1, so (0,0) is simply off-curve
  bool good = !it.A.is_inf() && !it.B.is_inf() && !it.C.is_inf() && it.A.y.sqr() == it.A.x.sqr() * it.A.x + Fq::from_u64(3) &&
      it.C.y.sqr() == it.C.x.sqr() * it.C.x + Fq::from_u64(3)
              && it.B.y.sqr() == it.B.x.sqr() * it.B.x + K.twist_b;
  const bool acc_inf = nacc.is_inf();
  Fq12 R[VM_REGS]; Fq2 X = it.B.x, Y = it.B.y, Z = Fq2::one();                                   // G2 point of the running loop, homogeneous projective
  // mul_by_q, alt_bn128_g2.cpp:367-372
  const Fq2 q1x = T.twist_mul_by_q_x * it.B.x.frob(1), q1y = T.twist_mul_by_q_y * it.B.y.frob(1), q2x = T.twist_mul_by_q_x * q1x.frob(1),
      q2y = (T.twist_mul_by_q_y * q1y.frob(1)).neg();
  const Fq cny = it.C.y.neg();
#pragma unroll 1
  for (uint32_t pc = 0;; pc++) {
    const uint32_t ins = prog[pc], op = ins & 0xff, d = (ins >> 8) & 0xff, a = (ins >> 16) & 0xff, b = ins >> 24;
    if (op == VM_END) break;
    // fp12_2over3over2.tcc:91-104
    if (op == VM_MUL) {
      const Fq12 x = R[a], y = R[b];
      Fq6 aA = x.c0 * y.c0, bB = x.c1 * y.c1;
      R[d] = {aA + bB.mul_by_v(), (x.c0 + x.c1) * (y.c0 + y.c1) - aA - bB};
    }
    // mul_by_024 (fp12_2over3over2.tcc:240-335): R[b] = (a, 0, c | 0, e, 0)
    else if (op == VM_MUL024) {
      const Fq12 x = R[a]; const Fq2 la = R[b].c0.c0, lc = R[b].c0.c2, le = R[b].c1.c1;
      // t0 = x.c0 * (a, 0, c): 5 products; t1 = x.c1 * (0, e, 0): 3 products; t2 = (x.c0 + x.c1) * (a, e, c): 6 products (Karatsuba) -> 14 Fq2 products instead
      // of 18
      Fq2 p0 = x.c0.c0 * la, p2 = x.c0.c2 * lc, p1c = x.c0.c1 * lc, p1a = x.c0.c1 * la, pm = (x.c0.c0 + x.c0.c2) * (la + lc);
      // (u0 a + xi u1 c, u1 a + xi u2 c, u2 a + u0 c)
      Fq6 t0 = {p0 + p1c.mul_xi(), p1a + p2.mul_xi(), pm - p0 - p2};
      Fq6 t1 = {(x.c1.c2 * le).mul_xi(), x.c1.c0 * le, x.c1.c1 * le};
      Fq6 sx = x.c0 + x.c1, sl = {la, le, lc}, t2 = sx * sl;
      R[d] = {t0 + t1.mul_by_v(), t2 - t0 - t1};
    }
    // cyclotomic_squared :173-238 (Granger-Scott: three Fq4 squarings)
    else if (op == VM_CYCSQR) {
      const Fq12 x = R[a]; Fq2 z0 = x.c0.c0, z4 = x.c0.c1, z3 = x.c0.c2, z2 = x.c1.c0, z1 = x.c1.c1, z5 = x.c1.c2, tmp, t0, t1, t2, t3, t4, t5;
      tmp = z0 * z1; t0 = (z0 + z1) * (z0 + z1.mul_xi()) - tmp - tmp.mul_xi(); t1 = tmp + tmp;
      tmp = z2 * z3; t2 = (z2 + z3) * (z2 + z3.mul_xi()) - tmp - tmp.mul_xi(); t3 = tmp + tmp;
      tmp = z4 * z5; t4 = (z4 + z5) * (z4 + z5.mul_xi()) - tmp - tmp.mul_xi(); t5 = tmp + tmp;
      z0 = t0 - z0; z0 = z0 + z0 + t0; z1 = t1 + z1; z1 = z1 + z1 + t1; tmp = t5.mul_xi(); z2 = tmp + z2; z2 = z2 + z2 + tmp;
      z3 = t4 - z3; z3 = z3 + z3 + t4; z4 = t2 - z4; z4 = z4 + z4 + t2; z5 = t3 + z5; z5 = z5 + z5 + t3;
      R[d] = {{z0, z4, z3}, {z2, z1, z5}};
    }
    // unitary_inverse
    else if (op == VM_CONJ) {
      const Fq12 x = R[a];
      R[d] = {x.c0, x.c1.neg()};
    }
    else if (op == VM_FROB) { const Fq12 x = R[a]; R[d] = {fq6_frob(x.c0, b, T), fq6_mul_fq2(fq6_frob(x.c1, b, T), T.fq12_c1[b % 12])}; }
    // fp12 :128-137 over fp6 :128-146
    else if (op == VM_INV) {
      const Fq12 x = R[a]; Fq6 s0 = x.c0 * x.c0, s1 = x.c1 * x.c1, t = s0 - s1.mul_by_v();
      Fq2 t0 = t.c0.sqr(), t1 = t.c1.sqr(), t2 = t.c2.sqr(), t3 = t.c0 * t.c1, t4 = t.c0 * t.c2, t5 = t.c1 * t.c2, d0 = t0 - t5.mul_xi(),
          d1 = t2.mul_xi() - t3, d2 = t1 - t4;
      Fq2 t6 = (t.c0 * d0 + (t.c2 * d1 + t.c1 * d2).mul_xi()).inv(); Fq6 ti = {t6 * d0, t6 * d1, t6 * d2}; R[d] = {x.c0 * ti, (x.c1 * ti).neg()};
    }
    else if (op == VM_ONE) R[d] = {Fq6::one(), Fq6::zero()};
    // doubling_step_for_flipped_miller_loop :242-268
    else if (op == VM_DBL) {
      Fq2 A = (X * Y).mul_fq(K.two_inv), B = Y.sqr(), C = Z.sqr(), D = C + C + C, E = K.twist_b * D, F = E + E + E, G = (B + F).mul_fq(K.two_inv),
          H = (Y + Z).sqr() - (B + C), I = E - B, J = X.sqr(), E2 = E.sqr();
      X = A * (B - F); Y = G.sqr() - (E2 + E2 + E2); Z = B * H;
      R[d] = fq12_sparse(I.mul_xi(), H.neg().mul_fq(it.A.y), (J + J + J).mul_fq(it.A.x));
    }
    // mixed_addition_step_for_flipped_miller_loop :270-293
    else if (op == VM_ADD) {
      const Fq2 x2 = b == 0 ? it.B.x : b == 1 ? q1x : q2x, y2 = b == 0 ? it.B.y : b == 1 ? q1y : q2y;
      Fq2 D = X - x2 * Z, E = Y - y2 * Z, F = D.sqr(), G = E.sqr(), H = D * F, I = X * F, J = H + Z * G - (I + I), Y1 = Y;
      X = D * J; Y = E * (I - J) - H * Y1; Z = Z * H;
      R[d] = fq12_sparse((E * x2 - D * y2).mul_xi(), D.mul_fq(it.A.y), E.neg().mul_fq(it.A.x));
    }
    // miller_loop :387-411 with a precomputed G2
    else if (op == VM_LINE) {
      const EllCoeffsDev c = a == 1 ? gamma_lines[b] : delta_lines[b];
      if (a == 1) R[d] = acc_inf ? Fq12{Fq6::one(), Fq6::zero()} : fq12_sparse(c.ell_0, c.ell_VW.mul_fq(nacc.y), c.ell_VV.mul_fq(nacc.x));
      else R[d] = fq12_sparse(c.ell_0, c.ell_VW.mul_fq(cny), c.ell_VV.mul_fq(it.C.x));
    }
  }
  // :556-560
  ok[i] = (good && R[0] == *alpha_beta) ? 1 : 0;
}


// ---- the schedule kernel: one 256-thread WORKGROUP per proof, interpreting the operation schedule of verify_sched.hpp
// ------------------------------------------- Every round of the schedule is one operation per lane on field elements held in LDS: a Montgomery product, or a
// linear combination with small integer coefficients (LIN8: eight lanes per value — three terms each on 64-bit limb accumulators, one carry step, a DPP tree
// over the eight lanes, one Barrett-like step; LIN1: one lane, up to three terms). Values are nine 29-bit limbs (Montgomery radix 2^261, field29_gfx950.inc /
// l29:: in verify_sched.hpp): 162 multiply-adds per product and no carry instruction, no modular correction per term of a sum. The schedule is the same for
// every proof, so the workgroups of a launch never diverge. History on MI355X (tools/verify_bench.py): one LANE per proof (round 1-2, the kernels above) 46 ->
// 25 ms per launch; one workgroup per proof on 8 x 32-bit limbs with NAF sums (round 3, first half): 4.0 ms, 2,944 rounds of ~1.36 us — a wave alone on a SIMD
// issues an instruction every ~7 cycles (tools/valu_probe.hip), so a round is as long as its longest lane's instruction count; this form: see
// profiles/r03_verify_batch.txt.
// ok[i]: 1 accept, 0 reject, 2 = the input accumulator was the point at infinity (the host verifier decides: the gamma pairing is the identity then, which a
// fixed schedule cannot express).
struct SchedInfo { uint32_t n_rounds_padded, n_slots, n_consts, out_slot[vsched::N_OUT]; };
constexpr uint32_t VS_CONST_FLAG = 0x8000u;
// LDS: [n_slots working values | n_consts constants of the key], 48 bytes each — the constants are copied in once per proof (coalesced), so that every operand
// of every round is LDS reads: with the constants in global memory a round's critical path held a dependent global load (measured: 1.4 us per round instead of
// ~0.6). The instruction words are a dense array (round, lane) -> uint4: the fetch of round r + D depends on r alone, and D = vsched::PREFETCH_ROUNDS rounds
// are in flight in a ring of registers (the loop is unrolled D times so that the ring's indices are static).
struct NegAcc3 { Fq xw, nyw, w; };     // -acc = (x, -y) as (x w, -y w, w), w = ZZ ZZZ of the accumulation's extended Jacobian sum; w = 0: the point at infinity
// barrier over the workgroup's LDS traffic only: __syncthreads() also drains the outstanding GLOBAL loads (s_waitcnt vmcnt(0)) — here the next round's
// instruction words, fetched a round ahead precisely so that nobody waits for them
__device__ __forceinline__ void vs_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void vs_load(const uint32_t *lds, uint32_t n_slots, uint32_t ref, uint32_t (&x)[9]) {
  const uint32_t idx = (ref & VS_CONST_FLAG) ? n_slots + (ref & (VS_CONST_FLAG - 1)) : (ref & 0xffffu); const uint32_t *p = lds + idx * l29::STRIDE;
  const uint4 q0 = *reinterpret_cast<const uint4 *>(p), q1 = *reinterpret_cast<const uint4 *>(p + 4);
  x[0] = q0.x;
  x[1] = q0.y;
  x[2] = q0.z;
  x[3] = q0.w;
  x[4] = q1.x;
  x[5] = q1.y;
  x[6] = q1.z;
  x[7] = q1.w;
  x[8] = p[8];
}
__device__ __forceinline__ void vs_store(uint32_t *lds, uint32_t slot, const uint32_t (&x)[9]) { uint32_t *p = lds + slot * l29::STRIDE;
  *reinterpret_cast<uint4 *>(p) = make_uint4(x[0], x[1], x[2], x[3]); *reinterpret_cast<uint4 *>(p + 4) = make_uint4(x[4], x[5], x[6], x[7]); p[8] = x[8]; }
// lane i += lane i - N within its row of 16 (DPP row_shr; lanes without a source add 0)
template <int N> __device__ __forceinline__ void vs_add_from_below(uint32_t (&l)[9]) {
#pragma unroll
  for (int i = 0; i < 9; i++) l[i] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)l[i], 0x110 + N, 0xf, 0xf, true); }
// one round of the schedule for this lane: wd = its four instruction words
__device__ __forceinline__ void vs_round(uint32_t *lds, uint32_t n_slots, const uint4 wd) {
  constexpr uint32_t LIVE = vsched::LIVE_BIT, STORE = vsched::STORE_BIT;
  const uint32_t kind = (uint32_t)__builtin_amdgcn_readfirstlane(wd.x) >> vsched::KIND_SHIFT;   // (the builtin returns int: a signed shift would smear the top bit)
    // the WAVE's kind (every lane of a wave carries it, idle lanes too)
  const bool live = wd.x & LIVE;
  uint32_t v[9];
  if (kind == vsched::K_MUL) {
    if (live) { Fq29 a, b; vs_load(lds, n_slots, wd.y, a.l); vs_load(lds, n_slots, wd.z, b.l); const Fq29 pr = Fq29::mul(a, b);
#pragma unroll
      for (int k = 0; k < 9; k++) v[k] = pr.l[k]; }
  } else if (kind != vsched::K_IDLE) {
    uint64_t acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (live) {
      uint32_t x[9];
      vs_load(lds, n_slots, wd.y & 0xffffu, x);
      l29::term(acc, x, wd.y >> 17, (wd.y >> 16) & 1);
      vs_load(lds, n_slots, wd.z & 0xffffu, x);
      l29::term(acc, x, wd.z >> 17, (wd.z >> 16) & 1);
      vs_load(lds, n_slots, wd.w & 0xffffu, x);
      l29::term(acc, x, wd.w >> 17, (wd.w >> 16) & 1);
    }
    l29::norm64(acc, v);
    // the sum of a group's eight lanes arrives in its last lane
    if (kind == vsched::K_LIN8) {
      vs_add_from_below<4>(v);
      vs_add_from_below<2>(v);
      l29::norm32(v);
      vs_add_from_below<1>(v);
    }
    l29::barrett(v);
  }
  // (no barrier between a round's reads and its writes: the builder never hands out a slot as destination in the round that reads it last)
  if (wd.x & STORE) vs_store(lds, wd.x & 0x7fffu, v);
}
// this lane's 16 bytes of round `src` straight from global memory into LDS (no register in between: nothing for the compiler to copy while the load is in flight).
// lds_wave_base: byte address in LDS of the wave's 64 x 16 bytes, wave-uniform. Completion is counted by vmcnt like any load's; the compiler does not know about
// this load, the kernel waits for it by hand.
__device__ __forceinline__ void vs_fetch_to_lds(const uint4 *src, uint32_t lds_wave_base) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_wave_base)) : "memory");
}
// A round's instruction words reach the lane through a ring of vsched::PREFETCH_ROUNDS rounds in LDS filled by LDS-direct loads.  (Measured against it, n = 1:
// a plain load at the head of every round — the round waits for it — 1.10 ms per launch against 0.95; a ring of registers as fast, but the compiler copies the ring at
// the loop's back edge behind a full wait, and 16 KB of LDS are there to be had.  profiles/r06_verify_batch.txt)
static __global__ void __launch_bounds__(256) k_verify_sched29(const uint4 *__restrict__ prog, const uint4 *__restrict__ consts,
    const VerifyItem *__restrict__ items, const NegAcc3 *__restrict__ neg_acc, uint32_t n, SchedInfo si, uint8_t *__restrict__ ok, uint32_t *__restrict__ trace,
    uint32_t trace_every) {
  extern __shared__ uint4 vs_lds4[]; uint32_t *lds = reinterpret_cast<uint32_t *>(vs_lds4);
  const uint32_t i = blockIdx.x, lane = threadIdx.x; if (i >= n) return;
  // a verification is one dependent chain on one CU: beside a prover's kernels its waves go first (the chain kernels of a proof run at 3 too, the accumulation at 0)
  __builtin_amdgcn_s_setprio(3);
  // (tests: proof 0's values after every trace_every-th round, n_slots x 12 words each — compared with the host model of the same arithmetic, vsched::simulate29)
  auto dump = [&](uint32_t r) { if (trace && i == 0 && (r + 1) % trace_every == 0) { const uint32_t words = si.n_slots * l29::STRIDE; uint32_t *dst = trace + (size_t)(r / trace_every) * words;
      for (uint32_t k = lane; k < words; k += 256) dst[k] = lds[k]; } };
  constexpr uint32_t DL = vsched::PREFETCH_ROUNDS; static_assert((DL & (DL - 1)) == 0 && DL <= 32, "ring size");
  // the LDS ring: after the values and the constants, DL rounds of 4 KB; a wave reads and refills only its own kilobyte of each
  uint4 *ring_lds = vs_lds4 + (size_t)(si.n_slots + si.n_consts) * (l29::STRIDE / 4);
  const uint32_t ring_wave_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)ring_lds + (lane >> 6) * 1024u;
  { uint4 *dst = vs_lds4 + si.n_slots * (l29::STRIDE / 4); for (uint32_t k = lane; k < si.n_consts * (l29::STRIDE / 4); k += 256) dst[k] = consts[k]; }
  const VerifyItem &it = items[i]; const NegAcc3 nacc = neg_acc[i];
  // vsched::IN_* order
  if (lane < 11) {
    const Fq v = lane == 0 ? it.A.x : lane == 1 ? it.A.y : lane == 2 ? it.B.x.c0 : lane == 3 ? it.B.x.c1 : lane == 4 ? it.B.y.c0 : lane == 5 ? it.B.y.c1 :
        lane == 6 ? it.C.x : lane == 7 ? it.C.y : lane == 8 ? nacc.xw : lane == 9 ? nacc.nyw : nacc.w;
    uint32_t w[8], l[9];
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = v.l[k];
    l29::lift(w, l); vs_store(lds, lane, l); }
  __syncthreads();
  const uint32_t n_rounds = si.n_rounds_padded - DL;        // (the builder's padding: a multiple of DL rounds, then DL idle ones that are fetched but never run)
  {
    // (every load the compiler knows of has been waited for at the barrier above; from here on this wave's only outstanding loads are the ring's, one per round)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (uint32_t d = 0; d < DL; d++) vs_fetch_to_lds(prog + (size_t)d * 256 + lane, ring_wave_base + d * 4096u);
    const uint4 *pf = prog + (size_t)DL * 256 + lane;
#pragma unroll 1
    for (uint32_t r = 0; r < n_rounds; r++) {
      const uint32_t slot = r & (DL - 1);
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DL - 1) : "memory");                     // the oldest of the DL fetches in flight — this round's — has landed
      const uint4 wd = ring_lds[slot * 256 + lane];
      vs_round(lds, si.n_slots, wd);
      vs_fetch_to_lds(pf + (size_t)r * 256, ring_wave_base + slot * 4096u);           // round r + DL into the kilobyte this round's words just left
      vs_lds_barrier();
      dump(r);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (lane == 0) { bool good = !it.A.is_inf() && !it.B.is_inf() && !it.C.is_inf();
    // the GT comparison and the on-curve residues: all zero; the norm of the Miller value: not zero
    for (int k = 0; k < vsched::N_OUT; k++) {
      uint32_t x[9];
      vs_load(lds, si.n_slots, si.out_slot[k], x);
      const bool z = l29::multiple_of_p(x);
      good = good && (k == vsched::OUT_NONZERO ? !z : z);
    }
    ok[i] = nacc.w.is_zero() ? 2 : good ? 1 : 0; }
}
}  // namespace zk
