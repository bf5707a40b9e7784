// Microbenchmark: VALU integer throughput on gfx950 for the primitives the field multiplication is built from, and the
// throughput of Montgomery-multiplication variants.  Build: hipcc --offload-arch=gfx950 -O3 -o fmul_bench fmul_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../blockmaze_amd/csrc/field.cuh"
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_mad(uint32_t *out, int iters) {   // 8 independent v_mad_u64_u32 chains per lane
  uint32_t x = threadIdx.x * 2654435761u + 1, y = blockIdx.x * 40503u + 7; uint64_t a0 = x, a1 = y, a2 = x + 1, a3 = y + 1, a4 = x + 2, a5 = y + 2, a6 = x + 3, a7 = y + 3;
  for (int i = 0; i < iters; i++) { a0 = (uint64_t)(uint32_t)a0 * x + a0; a1 = (uint64_t)(uint32_t)a1 * y + a1; a2 = (uint64_t)(uint32_t)a2 * x + a2; a3 = (uint64_t)(uint32_t)a3 * y + a3;
    a4 = (uint64_t)(uint32_t)a4 * x + a4; a5 = (uint64_t)(uint32_t)a5 * y + a5; a6 = (uint64_t)(uint32_t)a6 * x + a6; a7 = (uint64_t)(uint32_t)a7 * y + a7; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ (uint32_t)((a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) >> 32);
}
__global__ void k_mad_dep(uint32_t *out, int iters) {   // one dependent chain: latency
  uint32_t x = threadIdx.x * 2654435761u + 1; uint64_t a0 = x;
  for (int i = 0; i < iters; i++) { a0 = (uint64_t)(uint32_t)a0 * x + a0; a0 = (uint64_t)(uint32_t)a0 * x + a0; a0 = (uint64_t)(uint32_t)a0 * x + a0; a0 = (uint64_t)(uint32_t)a0 * x + a0; a0 = (uint64_t)(uint32_t)a0 * x + a0; a0 = (uint64_t)(uint32_t)a0 * x + a0; a0 = (uint64_t)(uint32_t)a0 * x + a0; a0 = (uint64_t)(uint32_t)a0 * x + a0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)a0 ^ (uint32_t)(a0 >> 32);
}
__global__ void k_add(uint32_t *out, int iters) {   // 8 independent 32-bit adds
  uint32_t x = threadIdx.x + 1, a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  for (int i = 0; i < iters; i++) { a0 += a1 ^ x; a1 += a2 ^ x; a2 += a3 ^ x; a3 += a4 ^ x; a4 += a5 ^ x; a5 += a6 ^ x; a6 += a7 ^ x; a7 += a0 ^ x; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void k_mullo(uint32_t *out, int iters) {   // 8 independent v_mul_lo_u32
  uint32_t x = threadIdx.x * 2654435761u + 1, a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  for (int i = 0; i < iters; i++) { a0 *= x; a1 *= x; a2 *= x; a3 *= x; a4 *= x; a5 *= x; a6 *= x; a7 *= x; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
__global__ void k_mulhi(uint32_t *out, int iters) {
  uint32_t x = threadIdx.x * 2654435761u + 1, a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
  for (int i = 0; i < iters; i++) { a0 = __umulhi(a0, x) + 1; a1 = __umulhi(a1, x) + 1; a2 = __umulhi(a2, x) + 1; a3 = __umulhi(a3, x) + 1; a4 = __umulhi(a4, x) + 1; a5 = __umulhi(a5, x) + 1; a6 = __umulhi(a6, x) + 1; a7 = __umulhi(a7, x) + 1; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

// ---- Montgomery multiplication variants -------------------------------------------------------------------------------
// VA: per row, 8 independent 32x32->64 products, then two carry chains (low halves, high halves)
template <class P> __device__ __forceinline__ Fp<P> mul_va(const Fp<P> &a, const Fp<P> &b) {
  uint32_t t[10]; for (int j = 0; j < 10; j++) t[j] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint64_t p[8];
#pragma unroll
    for (int j = 0; j < 8; j++) p[j] = (uint64_t)a.l[j] * b.l[i];
    uint64_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += (uint64_t)t[j] + (uint32_t)p[j]; t[j] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[8] = (uint32_t)c; t[9] = (uint32_t)(c >> 32);
    c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += (uint64_t)t[j + 1] + (uint32_t)(p[j] >> 32); t[j + 1] = (uint32_t)c; c >>= 32; }
    t[9] += (uint32_t)c;
    uint32_t m = t[0] * P::INV;
#pragma unroll
    for (int j = 0; j < 8; j++) p[j] = (uint64_t)m * P::MOD[j];
    c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += (uint64_t)t[j] + (uint32_t)p[j]; t[j] = (uint32_t)c; c >>= 32; }
    c += t[8]; t[8] = (uint32_t)c; t[9] += (uint32_t)(c >> 32);
    c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { c += (uint64_t)t[j + 1] + (uint32_t)(p[j] >> 32); t[j + 1] = (uint32_t)c; c >>= 32; }
    t[9] += (uint32_t)c;
#pragma unroll
    for (int j = 0; j < 9; j++) t[j] = t[j + 1];
    t[9] = 0;
  }
  Fp<P> r; for (int j = 0; j < 8; j++) r.l[j] = t[j]; return Fp<P>::reduce_once(r);
}
// VB: separated operand scanning: full 16-limb product by rows with independent mads (64-bit addend = previous column value), then 8 reduction rows
template <class P> __device__ __forceinline__ Fp<P> mul_vb(const Fp<P> &a, const Fp<P> &b) {
  // product scanning with 64-bit lo/hi split sums: col[k] = sum of lo parts at k + hi parts at k-1, accumulated as 64-bit (at most 16 terms of 32 bits: fits)
  uint64_t col[17]; for (int k = 0; k < 17; k++) col[k] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) { uint64_t p = (uint64_t)a.l[j] * b.l[i]; col[i + j] += (uint32_t)p; col[i + j + 1] += p >> 32; }
  // Montgomery reduction, limb by limb; the running carry is folded into the next column
#pragma unroll
  for (int i = 0; i < 8; i++) {
    uint32_t m = (uint32_t)col[i] * P::INV;
#pragma unroll
    for (int j = 0; j < 8; j++) { uint64_t p = (uint64_t)m * P::MOD[j]; col[i + j] += (uint32_t)p; col[i + j + 1] += p >> 32; }
    col[i + 1] += col[i] >> 32;   // low word of col[i] is now zero mod 2^32
  }
  Fp<P> r; uint64_t c = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) { c += col[8 + j]; r.l[j] = (uint32_t)c; c >>= 32; }
  return Fp<P>::reduce_once(r);
}
// VC: product scanning (Comba) with a 96-bit column accumulator: every 32x32 MAC is one v_mad_u64_u32 whose carry-out feeds one v_addc into the third word
__device__ __forceinline__ void mac96(uint64_t &acc, uint32_t &top, uint32_t a, uint32_t b) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(top) : "v"(a), "v"(b) : "vcc");
}
__device__ __forceinline__ void mac96s(uint64_t &acc, uint32_t &top, uint32_t a, uint32_t b_const) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(top) : "v"(a), "s"(b_const) : "vcc");
}
template <class P> __device__ __forceinline__ Fp<P> mul_vc(const Fp<P> &a, const Fp<P> &b) {
  uint64_t acc = 0; uint32_t top = 0, m[8]; Fp<P> r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) mac96(acc, top, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) mac96s(acc, top, m[i], P::MOD[k - i]);
    m[k] = (uint32_t)acc * P::INV; mac96s(acc, top, m[k], P::MOD[0]);
    acc = (acc >> 32) | ((uint64_t)top << 32); top = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; k++) {
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac96(acc, top, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac96s(acc, top, m[i], P::MOD[k - i]);
    r.l[k - 8] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)top << 32); top = 0;
  }
  return Fp<P>::reduce_once(r);
}
// VD: the same schedule in plain C++ (what the compiler makes of a 96-bit accumulator)
__device__ __forceinline__ void mac96c(uint64_t &acc, uint32_t &top, uint32_t a, uint32_t b) { uint64_t p = (uint64_t)a * b, s = acc + p; top += s < p; acc = s; }
template <class P> __device__ __forceinline__ Fp<P> mul_vd(const Fp<P> &a, const Fp<P> &b) {
  uint64_t acc = 0; uint32_t top = 0, m[8]; Fp<P> r;
#pragma unroll
  for (int k = 0; k < 8; k++) {
#pragma unroll
    for (int i = 0; i <= k; i++) mac96c(acc, top, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = 0; i < k; i++) mac96c(acc, top, m[i], P::MOD[k - i]);
    m[k] = (uint32_t)acc * P::INV; mac96c(acc, top, m[k], P::MOD[0]);
    acc = (acc >> 32) | ((uint64_t)top << 32); top = 0;
  }
#pragma unroll
  for (int k = 8; k < 16; k++) {
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac96c(acc, top, a.l[i], b.l[k - i]);
#pragma unroll
    for (int i = k - 7; i < 8; i++) mac96c(acc, top, m[i], P::MOD[k - i]);
    r.l[k - 8] = (uint32_t)acc; acc = (acc >> 32) | ((uint64_t)top << 32); top = 0;
  }
  return Fp<P>::reduce_once(r);
}
template <int V> __device__ __forceinline__ Fq mulv(const Fq &a, const Fq &b) { if (V == 0) return a * b; if (V == 1) return mul_va(a, b); if (V == 2) return mul_vb(a, b); if (V == 3) return mul_vc(a, b); return mul_vd(a, b); }
template <int V> __global__ void k_mont(const Fq *in, Fq *out, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; Fq x = in[i], y = in[i + 1], z = in[i + 2], w = in[i + 3];
  for (int k = 0; k < iters; k++) { x = mulv<V>(x, y); z = mulv<V>(z, w); y = mulv<V>(y, x); w = mulv<V>(w, z); }   // two independent chains
  out[i] = x + y + z + w;
}
template <int V> __global__ void k_mont1(const Fq *in, Fq *out, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; Fq x = in[i], y = in[i + 1];
  for (int k = 0; k < iters; k++) { x = mulv<V>(x, y); y = mulv<V>(y, x); x = mulv<V>(x, y); y = mulv<V>(y, x); }   // one dependent chain
  out[i] = x + y;
}

template <class K, class... A> static double timeit(K kern, dim3 g, dim3 b, A... args) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipLaunchKernelGGL(kern, g, b, 0, 0, args...); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(kern, g, b, 0, 0, args...); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  const int blocks = 256 * 16, threads = 256, n = blocks * threads; uint32_t *d; CK(hipMalloc(&d, n * 4 + 64)); Fq *fin, *fout; CK(hipMalloc(&fin, (n + 8) * 32)); CK(hipMalloc(&fout, (n + 8) * 32));
  std::vector<uint32_t> h((n + 8) * 8); for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffffu : 0xffffffffu); CK(hipMemcpy(fin, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  int it = 2000; double ms;
  ms = timeit(k_mad, dim3(blocks), dim3(threads), d, it); printf("v_mad_u64_u32 independent: %.1f Gop/s (lane ops)\n", (double)n * it * 8 / ms / 1e6);
  ms = timeit(k_mad_dep, dim3(blocks), dim3(threads), d, it); printf("v_mad_u64_u32 dependent chain, 16 waves/CU... : %.1f Gop/s\n", (double)n * it * 8 / ms / 1e6);
  ms = timeit(k_mad_dep, dim3(256), dim3(64), d, it); printf("v_mad_u64_u32 dependent, 1 wave/CU: %.2f ns per op (latency)\n", ms * 1e6 / (it * 8.0));
  ms = timeit(k_add, dim3(blocks), dim3(threads), d, it); printf("v_add/xor pairs independent: %.1f Gop/s (pairs)\n", (double)n * it * 8 / ms / 1e6);
  ms = timeit(k_mullo, dim3(blocks), dim3(threads), d, it); printf("v_mul_lo_u32: %.1f Gop/s\n", (double)n * it * 8 / ms / 1e6);
  ms = timeit(k_mulhi, dim3(blocks), dim3(threads), d, it); printf("v_mul_hi_u32 (+add): %.1f Gop/s\n", (double)n * it * 8 / ms / 1e6);
  it = 200;
  ms = timeit(k_mont<0>, dim3(blocks), dim3(threads), fin, fout, it); printf("mont mul V0 (CIOS, current) 2 chains: %.2f Gmul/s\n", (double)n * it * 4 / ms / 1e6);
  ms = timeit(k_mont<1>, dim3(blocks), dim3(threads), fin, fout, it); printf("mont mul VA (row products + 2 carry chains) 2 chains: %.2f Gmul/s\n", (double)n * it * 4 / ms / 1e6);
  ms = timeit(k_mont<2>, dim3(blocks), dim3(threads), fin, fout, it); printf("mont mul VB (column sums 64-bit) 2 chains: %.2f Gmul/s\n", (double)n * it * 4 / ms / 1e6);
  ms = timeit(k_mont<3>, dim3(blocks), dim3(threads), fin, fout, it); printf("mont mul VC (Comba, 96-bit accumulator, asm carry) 2 chains: %.2f Gmul/s\n", (double)n * it * 4 / ms / 1e6);
  ms = timeit(k_mont<4>, dim3(blocks), dim3(threads), fin, fout, it); printf("mont mul VD (Comba, plain C++) 2 chains: %.2f Gmul/s\n", (double)n * it * 4 / ms / 1e6);
  ms = timeit(k_mont1<3>, dim3(blocks), dim3(threads), fin, fout, it); printf("mont mul VC 1 chain: %.2f Gmul/s\n", (double)n * it * 4 / ms / 1e6);
  ms = timeit(k_mont1<3>, dim3(256), dim3(64), fin, fout, it); printf("mont mul VC latency (1 wave/CU): %.1f ns per mul\n", ms * 1e6 / (it * 4.0));
  ms = timeit(k_mont1<0>, dim3(blocks), dim3(threads), fin, fout, it); printf("mont mul V0 1 chain: %.2f Gmul/s\n", (double)n * it * 4 / ms / 1e6);
  ms = timeit(k_mont1<1>, dim3(blocks), dim3(threads), fin, fout, it); printf("mont mul VA 1 chain: %.2f Gmul/s\n", (double)n * it * 4 / ms / 1e6);
  ms = timeit(k_mont1<2>, dim3(blocks), dim3(threads), fin, fout, it); printf("mont mul VB 1 chain: %.2f Gmul/s\n", (double)n * it * 4 / ms / 1e6);
  ms = timeit(k_mont1<0>, dim3(256), dim3(64), fin, fout, it); printf("mont mul V0 latency (1 wave/CU): %.1f ns per mul\n", ms * 1e6 / (it * 4.0));
  ms = timeit(k_mont1<1>, dim3(256), dim3(64), fin, fout, it); printf("mont mul VA latency (1 wave/CU): %.1f ns per mul\n", ms * 1e6 / (it * 4.0));
  ms = timeit(k_mont1<2>, dim3(256), dim3(64), fin, fout, it); printf("mont mul VB latency (1 wave/CU): %.1f ns per mul\n", ms * 1e6 / (it * 4.0));
  // correctness cross-check of the variants against V0
  std::vector<uint32_t> o0(n * 8), o1(n * 8), o2(n * 8); hipLaunchKernelGGL(k_mont<0>, dim3(blocks), dim3(threads), 0, 0, fin, fout, 3); CK(hipMemcpy(o0.data(), fout, n * 32, hipMemcpyDeviceToHost));
  hipLaunchKernelGGL(k_mont<1>, dim3(blocks), dim3(threads), 0, 0, fin, fout, 3); CK(hipMemcpy(o1.data(), fout, n * 32, hipMemcpyDeviceToHost)); hipLaunchKernelGGL(k_mont<2>, dim3(blocks), dim3(threads), 0, 0, fin, fout, 3); CK(hipMemcpy(o2.data(), fout, n * 32, hipMemcpyDeviceToHost));
  printf("VA == V0: %s, VB == V0: %s\n", o0 == o1 ? "yes" : "NO", o0 == o2 ? "yes" : "NO");
  hipLaunchKernelGGL(k_mont<3>, dim3(blocks), dim3(threads), 0, 0, fin, fout, 3); CK(hipMemcpy(o1.data(), fout, n * 32, hipMemcpyDeviceToHost)); hipLaunchKernelGGL(k_mont<4>, dim3(blocks), dim3(threads), 0, 0, fin, fout, 3); CK(hipMemcpy(o2.data(), fout, n * 32, hipMemcpyDeviceToHost));
  printf("VC == V0: %s, VD == V0: %s\n", o0 == o1 ? "yes" : "NO", o0 == o2 ? "yes" : "NO");
  return 0;
}
