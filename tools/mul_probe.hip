// Throughput of the field products the bucket accumulation is built from, chip-wide, at 3 and 4 waves per SIMD:  8 x 32-bit limbs with carries (field_mul_gfx950.inc)
// against 9 x 29-bit limbs without (field29_gfx950.inc).   hipcc --offload-arch=gfx950 -O3 -I blockmaze_amd/csrc -o /tmp/mul_probe tools/mul_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "field.cuh"
namespace zk {
#include "field29_gfx950.inc"
}
using namespace zk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int V> __global__ void __launch_bounds__(256) k_mul(const Fq *in, Fq *out, int iters) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; Fq x = in[i], y = in[i + 1], z = in[i + 2], w = in[i + 3];
  if (V == 0) { for (int k = 0; k < iters; k++) { x = Fq::mul_lazy(x, y); z = Fq::mul_lazy(z, w); y = Fq::mul_lazy(y, x); w = Fq::mul_lazy(w, z); } }
  if (V == 1) { for (int k = 0; k < iters; k++) { x = Fq::sqr_lazy(x); z = Fq::sqr_lazy(z); y = Fq::sqr_lazy(y); w = Fq::sqr_lazy(w); } }
  if (V == 2) { for (int k = 0; k < iters; k++) { x = Fq::sub_lazy(x, y); z = Fq::sub_lazy(z, w); y = Fq::sub_lazy(y, x); w = Fq::sub_lazy(w, z); } }
  if (V == 3 || V == 4 || V == 5) { Fq29 a = Fq29::unpack(x.l), b = Fq29::unpack(y.l), c = Fq29::unpack(z.l), d = Fq29::unpack(w.l);
    if (V == 3) for (int k = 0; k < iters; k++) { a = Fq29::mul(a, b); c = Fq29::mul(c, d); b = Fq29::mul(b, a); d = Fq29::mul(d, c); }
    if (V == 4) for (int k = 0; k < iters; k++) { a = Fq29::sqr(a); c = Fq29::sqr(c); b = Fq29::sqr(b); d = Fq29::sqr(d); }
    if (V == 5) for (int k = 0; k < iters; k++) { a = Fq29::sub<6>(a, b); c = Fq29::sub<6>(c, d); b = Fq29::sub<4>(b, a); d = Fq29::sub<4>(d, c); }
    for (int j = 0; j < 8; j++) { x.l[j] = a.l[j] ^ c.l[j + 1]; y.l[j] = b.l[j] ^ d.l[j + 1]; } z = x; w = y; }
  Fq r; for (int j = 0; j < 8; j++) r.l[j] = x.l[j] ^ y.l[j] ^ z.l[j] ^ w.l[j]; out[i] = r;
}
template <int V> static int run(const char *name, const Fq *in, Fq *out) {
  for (int wps = 3; wps <= 4; wps++) { const int blocks = 256 * wps, it = 300; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mul<V>, dim3(blocks), dim3(256), 0, 0, in, out, it); CK(hipDeviceSynchronize());
    hipEventRecord(e0); hipLaunchKernelGGL(k_mul<V>, dim3(blocks), dim3(256), 0, 0, in, out, it); hipEventRecord(e1); CK(hipEventSynchronize(e1)); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %d waves/SIMD: %7.1f G ops/s chip-wide\n", name, wps, (double)blocks * 256 * it * 4 / ms / 1e6); }
  return 0;
}
int main() {
  const int n = 256 * 4 * 256; Fq *in, *out; CK(hipMalloc(&in, (n + 8) * 32)); CK(hipMalloc(&out, (n + 8) * 32));
  std::vector<uint32_t> h((n + 8) * 8); for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffffu : 0xffffffffu); CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  if (run<0>("product, 8 x 32 (128 mad + 128 addc)", in, out)) return 1;
  if (run<3>("product, 9 x 29 (162 mad)", in, out)) return 1;
  if (run<1>("square, 8 x 32 (100 mad + 100 addc)", in, out)) return 1;
  if (run<4>("square, 9 x 29 (126 mad)", in, out)) return 1;
  if (run<2>("difference, 8 x 32 (sub + masked add: 25)", in, out)) return 1;
  if (run<5>("difference, 9 x 29 (9 + parallel carry)", in, out)) return 1;
  return 0;
}
