// probe: layout of global_load_lds_dwordx4 on gfx950 (does lane l of a wave land at M0 + 16 l?) and visibility after s_waitcnt vmcnt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const uint4 *src, uint4 *dst) {
  extern __shared__ uint4 lds[];
  const uint4 *p = src + threadIdx.x;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + (threadIdx.x / 64) * 1024;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p), "s"(__builtin_amdgcn_readfirstlane(base)) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  dst[threadIdx.x] = lds[threadIdx.x];     // own wave's region only: no barrier needed
}
int main() {
  const int n = 256; std::vector<uint4> h(n), o(n); for (int i = 0; i < n; i++) h[i] = make_uint4(i, i * 3 + 1, i ^ 0x55, 7 * i);
  uint4 *ds, *dd; hipMalloc(&ds, n * 16); hipMalloc(&dd, n * 16); hipMemcpy(ds, h.data(), n * 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(n), 8192, 0, ds, dd); hipMemcpy(o.data(), dd, n * 16, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < n; i++) if (o[i].x != h[i].x || o[i].y != h[i].y || o[i].z != h[i].z || o[i].w != h[i].w) { if (bad < 5) printf("lane %d: got %u %u %u %u\n", i, o[i].x, o[i].y, o[i].z, o[i].w); bad++; }
  printf("ldsdma probe: %d mismatches of %d\n", bad, n); return bad != 0;
}
