// Issue cost of the VALU instructions the field arithmetic is made of, measured in place on gfx950: cycles per wave-instruction and SIMD at 1 / 2 / 4 waves per SIMD.
// Every kernel runs one asm block of 32 instructions per loop iteration; the clock is taken from s_memtime around the loop of wave 0, so the figures are cycles, not
// nanoseconds at an assumed frequency.   Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/valu_probe.hip && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define R4(x) x x x x
#define R8(x) R4(x) R4(x)
#define R16(x) R8(x) R8(x)
#define R32(x) R16(x) R16(x)

enum { MAD_IND = 0, MAD_DEP, ADDC_VCC, ADDC_E64, PAIR1, PAIR2, PAIR_SGPR, ADD_U32, MUL_LO, SUBB_CHAIN, MAD_S, PAIR_FRESH, AND_ADDC, ADD_E64, AND_LIT, LSHR_B64, ALIGNBIT, ADD_NOP, MAD_NOP, MAD_ADD, N_KINDS };
static const char *names[N_KINDS] = {"v_mad_u64_u32, 4 independent accumulators", "v_mad_u64_u32, one dependent accumulator", "v_addc_co_u32 e32 (vcc in, vcc out), 4 registers", "v_addc_co_u32 e64 with 0, 0 operands",
  "mad + addc pairs, ONE 96-bit accumulator (the product's inner pattern)", "mad + addc pairs, two accumulators alternating", "mad + addc pairs, carries in s[20:21] / s[22:23] instead of vcc", "v_add_u32, 4 independent", "v_mul_lo_u32, 4 independent",
  "v_subb_co_u32 chain (8 limbs)", "v_mad_u64_u32 with an SGPR factor, 4 accumulators", "pairs, every 4th carry written fresh (0 + 0 + vcc)", "v_and_b32 + v_addc_co_u32 (masked constant add)", "v_add_u32 forced to the 8-byte VOP3 encoding", "v_and_b32 with a 32-bit literal (8 bytes)", "v_lshrrev_b64 by 29, 4 independent", "v_alignbit_b32, 4 independent",
  "v_add_u32 e32 + s_nop 0 alternating (per pair)", "v_mad_u64_u32 + s_nop 0 alternating (per pair)", "v_mad_u64_u32 + v_add_u32 e32 alternating (per pair)"};

template <int KIND> __global__ void __launch_bounds__(256) k_probe(uint32_t *out, uint64_t *clk, int iters, uint32_t seed) {
  uint32_t x = threadIdx.x * 2654435761u + seed, y = blockIdx.x * 40503u + 7u + seed, sc = seed | 0x80000001u;
  uint64_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 1; uint32_t t0 = 1, t1 = 2, t2 = 3, t3 = 4, t4 = 5, t5 = 6, t6 = 7, t7 = 8;
  uint64_t c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (KIND == MAD_IND) asm volatile(R8("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %5, %4, %2\n v_mad_u64_u32 %3, vcc, %5, %4, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc");
    if (KIND == MAD_DEP) asm volatile(R32("v_mad_u64_u32 %0, vcc, %1, %2, %0\n") : "+v"(a0) : "v"(x), "v"(y) : "vcc");
    if (KIND == ADDC_VCC) asm volatile(R8("v_addc_co_u32 %0, vcc, 0, %0, vcc\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n") : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : : "vcc");
    if (KIND == ADDC_E64) asm volatile(R8("v_addc_co_u32 %0, vcc, 0, 0, vcc\n v_addc_co_u32 %1, vcc, 0, 0, vcc\n v_addc_co_u32 %2, vcc, 0, 0, vcc\n v_addc_co_u32 %3, vcc, 0, 0, vcc\n") : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : : "vcc");
    if (KIND == PAIR1) asm volatile(R16("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n") : "+v"(a0), "+v"(t0) : "v"(x), "v"(y) : "vcc");
    if (KIND == PAIR2) asm volatile(R8("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_mad_u64_u32 %2, vcc, %5, %4, %2\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n") : "+v"(a0), "+v"(t0), "+v"(a1), "+v"(t1) : "v"(x), "v"(y) : "vcc");
    if (KIND == PAIR_SGPR) asm volatile(R8("v_mad_u64_u32 %0, s[20:21], %4, %5, %0\n v_addc_co_u32 %1, s[20:21], 0, %1, s[20:21]\n v_mad_u64_u32 %2, s[22:23], %5, %4, %2\n v_addc_co_u32 %3, s[22:23], 0, %3, s[22:23]\n") : "+v"(a0), "+v"(t0), "+v"(a1), "+v"(t1) : "v"(x), "v"(y) : "s20", "s21", "s22", "s23");
    if (KIND == ADD_U32) asm volatile(R8("v_add_u32 %0, %4, %0\n v_add_u32 %1, %4, %1\n v_add_u32 %2, %4, %2\n v_add_u32 %3, %4, %3\n") : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x));
    if (KIND == MUL_LO) asm volatile(R8("v_mul_lo_u32 %0, %4, %0\n v_mul_lo_u32 %1, %4, %1\n v_mul_lo_u32 %2, %4, %2\n v_mul_lo_u32 %3, %4, %3\n") : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x));
    if (KIND == SUBB_CHAIN) asm volatile(R4("v_sub_co_u32 %0, vcc, %0, %8\n v_subb_co_u32 %1, vcc, %1, %9, vcc\n v_subb_co_u32 %2, vcc, %2, %8, vcc\n v_subb_co_u32 %3, vcc, %3, %9, vcc\n v_subb_co_u32 %4, vcc, %4, %8, vcc\n v_subb_co_u32 %5, vcc, %5, %9, vcc\n v_subb_co_u32 %6, vcc, %6, %8, vcc\n v_subb_co_u32 %7, vcc, %7, %9, vcc\n")
                                          : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7) : "v"(x), "v"(y) : "vcc");
    if (KIND == MAD_S) asm volatile(R8("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "s"(sc) : "vcc");
    if (KIND == PAIR_FRESH) asm volatile(R4("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, 0, vcc\n" "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n" "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n" "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n") : "+v"(a0), "+v"(t0) : "v"(x), "v"(y) : "vcc");
    if (KIND == ADD_E64) asm volatile(R8("v_add_u32_e64 %0, %4, %0\n v_add_u32_e64 %1, %4, %1\n v_add_u32_e64 %2, %4, %2\n v_add_u32_e64 %3, %4, %3\n") : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x));
    if (KIND == AND_LIT) asm volatile(R8("v_and_b32 %0, 0x1fffffff, %0\n v_and_b32 %1, 0x1ffffffe, %1\n v_and_b32 %2, 0x1ffffffd, %2\n v_and_b32 %3, 0x1ffffffb, %3\n") : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x));
    if (KIND == LSHR_B64) asm volatile(R8("v_lshrrev_b64 %0, 29, %0\n v_lshrrev_b64 %1, 29, %1\n v_lshrrev_b64 %2, 29, %2\n v_lshrrev_b64 %3, 29, %3\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x));
    if (KIND == ALIGNBIT) asm volatile(R8("v_alignbit_b32 %0, %4, %0, 29\n v_alignbit_b32 %1, %4, %1, 29\n v_alignbit_b32 %2, %4, %2, 29\n v_alignbit_b32 %3, %4, %3, 29\n") : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x));
    if (KIND == ADD_NOP) asm volatile(R16("v_add_u32 %0, %1, %0\n s_nop 0\n") : "+v"(t0) : "v"(x));
    if (KIND == MAD_NOP) asm volatile(R16("v_mad_u64_u32 %0, vcc, %1, %2, %0\n s_nop 0\n") : "+v"(a0) : "v"(x), "v"(y) : "vcc");
    if (KIND == MAD_ADD) asm volatile(R16("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_add_u32 %1, %2, %1\n") : "+v"(a0), "+v"(t0) : "v"(x), "v"(y) : "vcc");
    if (KIND == AND_ADDC) asm volatile(R16("v_and_b32 %1, %3, %2\n v_addc_co_u32 %0, vcc, %0, %1, vcc\n") : "+v"(t0), "+v"(t1) : "v"(x), "s"(sc) : "vcc");
  }
  uint64_t c1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3) ^ (uint32_t)((a0 + a1 + a2 + a3) >> 32) ^ t0 ^ t1 ^ t2 ^ t3 ^ t4 ^ t5 ^ t6 ^ t7;
  if (blockIdx.x == 0 && threadIdx.x == 0) *clk = c1 - c0;
}
template <int KIND> static int run(uint32_t *d, uint64_t *clk) {
  const int it = 4000;
  for (int wps = 1; wps <= 4; wps++) {   // waves per SIMD: 256 CUs x 4 SIMDs x wps waves
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); const int blocks = 256 * wps;
    hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, clk, it, 1u); CK(hipDeviceSynchronize());
    hipEventRecord(e0); hipLaunchKernelGGL(k_probe<KIND>, dim3(blocks), dim3(256), 0, 0, d, clk, it, 3u); hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); uint64_t c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const double instr_per_wave = 32.0 * it, cyc = (double)c / instr_per_wave / wps;   // s_memtime ticks of wave 0 per instruction, divided by the waves sharing its SIMD
    const double ns = ms * 1e6 / (instr_per_wave * wps);   // wall-clock nanoseconds per wave-instruction and SIMD
    printf("  %d wave(s)/SIMD: %6.3f ns per wave-instruction and SIMD = %5.2f cycles at 2.1 GHz, %7.1f G lane-ops/s\n", wps, ns, ns * 2.1, (double)blocks * 256 * instr_per_wave / ms / 1e6); (void)cyc;
  }
  return 0;
}
int main() {
  uint32_t *d; CK(hipMalloc(&d, 256 * 4 * 256 * 4 + 64)); uint64_t *clk; CK(hipMalloc(&clk, 8));
#define RUN(K) printf("%s\n", names[K]); if (run<K>(d, clk)) return 1;
  RUN(MAD_IND) RUN(MAD_DEP) RUN(MAD_S) RUN(ADDC_VCC) RUN(ADDC_E64) RUN(PAIR1) RUN(PAIR2) RUN(PAIR_SGPR) RUN(PAIR_FRESH) RUN(ADD_U32) RUN(MUL_LO) RUN(SUBB_CHAIN) RUN(AND_ADDC) RUN(ADD_E64) RUN(AND_LIT) RUN(LSHR_B64) RUN(ALIGNBIT) RUN(ADD_NOP) RUN(MAD_NOP) RUN(MAD_ADD)
  return 0;
}
