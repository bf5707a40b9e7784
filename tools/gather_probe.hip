// Calibration of rocprofv3's FETCH_SIZE on the access patterns of this repo (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern"):
// a streaming read of a 268 MB table (16 bytes per lane, coalesced), and N random gathers of 64-byte and of 128-byte records from the same table — the pattern of the
// H-query accumulation (4.19 M records of 64 bytes, each table entry touched about once).  Known byte counts: table size, N x 64, N x 128.
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe tools/gather_probe.hip;  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- ./gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_stream(const uint4 *t, size_t n16, uint32_t *out) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x; uint32_t a = 0;
  for (; i < n16; i += stride) { uint4 v = t[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; } out[blockIdx.x * blockDim.x + threadIdx.x] = a; }
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int REC> __global__ void k_gather(const uint4 *t, uint32_t n_rec, uint32_t per_lane, uint32_t *out) {   // every lane reads per_lane random records of REC bytes (a permutation-like hash of its counter)
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; uint32_t a = 0;
  for (uint32_t j = 0; j < per_lane; j++) { const uint32_t r = mix(g * per_lane + j + 0x9e3779b9u) % n_rec; const uint4 *p = t + (size_t)r * (REC / 16);
#pragma unroll
    for (int q = 0; q < REC / 16; q++) { uint4 v = p[q]; a ^= v.x ^ v.y ^ v.z ^ v.w; } }
  out[g] = a; }
int main() {
  const size_t bytes = (size_t)16 * 262144 * 64; uint4 *t; CK(hipMalloc(&t, bytes)); CK(hipMemset(t, 1, bytes)); const uint32_t lanes = 349504, per = 12; uint32_t *out; CK(hipMalloc(&out, (size_t)1 << 22));   // 4.19 M gathers, 12 per lane as in k_hacc_runs29
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, 0, t, bytes / 16, out);
    hipLaunchKernelGGL(k_gather<64>, dim3(lanes / 256), dim3(256), 0, 0, t, (uint32_t)(bytes / 64), per, out);
    hipLaunchKernelGGL(k_gather<128>, dim3(lanes / 256), dim3(256), 0, 0, t, (uint32_t)(bytes / 128), per, out);
    CK(hipDeviceSynchronize()); }
  printf("expected bytes: stream %zu, gather64 %zu, gather128 %zu\n", bytes, (size_t)lanes * per * 64, (size_t)lanes * per * 128); return 0;
}
