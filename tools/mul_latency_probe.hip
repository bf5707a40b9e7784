// Latency against throughput of the 29-bit field product: a chain of DEPENDENT products per lane at 1, 2 and 4 waves per SIMD, for the generated product
// (field29_gfx950.inc: one accumulator, every multiply-add waits for the one before it) and for forms with independent column accumulators.
// hipcc --offload-arch=gfx950 -O3 -I blockmaze_amd/csrc -o /tmp/mul_latency_probe tools/mul_latency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
namespace zk {
#include "field29_gfx950.inc"
}
using namespace zk;
// the product with one accumulator per column, in plain C++: 17 independent sums, every m_k added to the eight columns above it as soon as it is known — the
// dependent sequence is carry -> m_k -> carry.  Measured (profiles/r05_mul_latency_probe.txt): NO faster than the generated one-accumulator form at any occupancy —
// a dependent v_mad_u64_u32 issues as soon as the one before it has, a wave alone on its SIMD already reaches 80 % of the SIMD's product rate.  Kept here only.
struct Ilp {
  static constexpr uint32_t MASK = Fq29::MASK, INV = Fq29::INV;
  static __device__ __forceinline__ Fq29 reduce_columns(uint64_t (&B)[17]) {
    Fq29 r; uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
      uint64_t t = B[k] + carry;
      const uint32_t m = ((uint32_t)t * INV) & MASK;
      t += (uint64_t)m * Fq29::P29[0]; carry = t >> 29;
#pragma unroll
      for (int j = 1; j < 9; j++) B[k + j] += (uint64_t)m * Fq29::P29[j];
    }
#pragma unroll
    for (int k = 9; k < 17; k++) { const uint64_t t = B[k] + carry; r.l[k - 9] = (uint32_t)t & MASK; carry = t >> 29; }
    r.l[8] = (uint32_t)carry; return r; }
  static __device__ __forceinline__ Fq29 mul_ilp(const Fq29 &a, const Fq29 &b) {
    uint64_t B[17];
#pragma unroll
    for (int k = 0; k < 17; k++) { uint64_t s = 0;
#pragma unroll
      for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) s += (uint64_t)a.l[i] * b.l[k - i];
      B[k] = s; }
    return reduce_columns(B); }
  static __device__ __forceinline__ Fq29 mul2_ilp(const Fq29 &a, const Fq29 &b, const Fq29 &c, const Fq29 &d) {
    uint64_t B[17];
#pragma unroll
    for (int k = 0; k < 17; k++) { uint64_t s = 0;
#pragma unroll
      for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) s += (uint64_t)a.l[i] * b.l[k - i] + (uint64_t)c.l[i] * d.l[k - i];
      B[k] = s; }
    return reduce_columns(B); }
};
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int V> __device__ __forceinline__ Fq29 prod(const Fq29 &a, const Fq29 &b) {
  if constexpr (V == 0) return Fq29::mul(a, b);
  else return Ilp::mul_ilp(a, b);
}
template <int V, int CHAINS> __global__ void __launch_bounds__(256) k_chain(const uint32_t *in, uint32_t *out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x; Fq29 a[CHAINS], b;
  for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 9; i++) a[c].l[i] = in[(t * 9 + i + 11 * c) % 4096] & (i < 8 ? 0x1fffffffu : 0x00ffffffu);
  for (int i = 0; i < 9; i++) b.l[i] = in[(t * 9 + i + 100) % 4096] & (i < 8 ? 0x1fffffffu : 0x00ffffffu);
#pragma unroll 1
  for (int k = 0; k < iters; k++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) a[c] = prod<V>(a[c], b);
  }
  uint32_t x = 0; for (int c = 0; c < CHAINS; c++) for (int i = 0; i < 9; i++) x = x * 31 + a[c].l[i];
  out[t] = x;
}
template <int V, int CHAINS> static int run(const char *name, const uint32_t *in, uint32_t *out, std::vector<uint32_t> &first) {
  for (int wps : {1, 2, 4}) { const int blocks = 256 * wps, it = 200; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_chain<V, CHAINS>), dim3(blocks), dim3(256), 0, 0, in, out, it); CK(hipDeviceSynchronize());
    hipEventRecord(e0); hipLaunchKernelGGL((k_chain<V, CHAINS>), dim3(blocks), dim3(256), 0, 0, in, out, it); hipEventRecord(e1); CK(hipEventSynchronize(e1)); float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint32_t> h(256 * 256); CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    if (CHAINS == 1) { if (first.empty()) first = h; else if (first != h) { printf("%s: RESULTS DIFFER from the generated product\n", name); return 1; } }
    printf("%-52s %d chain(s), %d waves/SIMD: %7.1f ns per product and wave, %7.1f G products/s chip-wide\n", name, CHAINS, wps, ms * 1e6 / (it * CHAINS), (double)blocks * 256 * it * CHAINS / ms / 1e6); }
  return 0;
}
int main() {
  uint32_t *in, *out; CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 1024 * 256 * 4));
  std::vector<uint32_t> h(4096); for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u + 12345u); CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  std::vector<uint32_t> first;
  if (run<0, 1>("generated product (one accumulator)", in, out, first)) return 1;
  if (run<1, 1>("independent column accumulators", in, out, first)) return 1;
  if (run<0, 2>("generated product (one accumulator)", in, out, first)) return 1;
  if (run<1, 2>("independent column accumulators", in, out, first)) return 1;
  return 0;
}
