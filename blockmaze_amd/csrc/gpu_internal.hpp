// Shared by the HIP translation units of libzkgpu (gpu.hip, gpu_msm_g1.hip, gpu_msm_g2.hip, gpu_keyops.hip): device context, error macro, scan helper, stage
// timer.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>
#include "gpu.hpp"
#include "msm.cuh"

namespace zk {
// Wave priorities (round 5).  A proof is one VALU-saturating kernel (the H accumulation: four waves a SIMD, 52 % of the proof's instructions) and two dozen short
// dependent chains: the transforms and the other kernels of the critical chain, and the four witness MSMs beside them.  Whichever wave of a SIMD is oldest issues next,
// so the chains used to wait for each other and for the accumulation at random: the transforms took 222-244 us beside the witness MSMs against 148 us alone.  With
// s_setprio the order is explicit — chain kernels 3, witness MSMs 2, H accumulation 0: the chain runs at its stand-alone speed, the witness MSMs take the slots it
// leaves and still finish before the accumulation needs the whole chip (device side of a send proof 0.812 -> 0.765 ms; every other ranking tried is slower:
// profiles/r05_priorities.txt).  ZK_PRIO="family:level,..." overrides single families (ntt, rows, hsort, htail, step, expand, wit, wlanes, hacc; 0 = the hardware's
// default for everything: ZK_PRIO=off).  The level travels in bits 24.. of a small integer argument (field29.cuh: zk_take_prio).
inline uint32_t zk_prio_bits(const char *family) {
  static const std::string s = [] { const char *e = getenv("ZK_PRIO"); return std::string(e ? e : ""); }();
  static const char *defaults = "ntt:3,rows:3,hsort:3,htail:3,step:3,expand:3,wit:2,wlanes:2,hacc:0";
  if (s == "off") return 0;
  const std::string key = std::string(family) + ":";
  for (const std::string &src : {s, std::string(defaults)}) {
    const size_t at = src.find(key);
    if (at != std::string::npos && (at == 0 || src[at - 1] == ',')) { const int v = atoi(src.c_str() + at + key.size()); return (uint32_t)(v < 0 ? 0 : v > 3 ? 3 : v) << 24; }
  }
  return 0;
}
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw GpuError(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
class GpuContext {
 public:
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t fork_event = nullptr;
  hipEvent_t join_event[4] = {nullptr, nullptr, nullptr, nullptr};
  hipDeviceProp_t prop;
  explicit GpuContext(int device_index) {
    // (GPU_MAX_HW_QUEUES is raised by the library constructor in gpu.hip, before any HIP call of this process can have read it)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw GpuError("no HIP device visible: the prover's HIP path cannot run (there is no CPU fallback)");
    device = device_index % n;
    HIP_CHECK(hipSetDevice(device)); HIP_CHECK(hipGetDeviceProperties(&prop, device));
    // (measured and dropped: a high-priority main stream and low-priority auxiliary streams (1-2 % slower), and confining the auxiliary streams to 32-128 CUs
    // with hipExtStreamCreateWithCUMask — profiles/r03h_ab_cumask.txt: the transforms are not slowed by sharing SIMDs, the proof is bound by VALU issue as a
    // whole)
    HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (int i = 0; i < 4; i++) HIP_CHECK(hipStreamCreateWithFlags(&aux[i], hipStreamNonBlocking));
    HIP_CHECK(hipEventCreateWithFlags(&fork_event, hipEventDisableTiming));
    for (int i = 0; i < 4; i++) HIP_CHECK(hipEventCreateWithFlags(&join_event[i], hipEventDisableTiming));
  }
};
GpuContext &gpu();
static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
// optional HIP-event timing of a pipeline stage (bench.py's roofline leg)
struct Stage { size_t id; explicit Stage(const char *n, hipStream_t st = nullptr); ~Stage(); };

// exclusive scan of a uint32 array on the stream
struct Scanner {
  DevBuf<uint32_t> block_sums; size_t cap;
  explicit Scanner(size_t n) : block_sums(cdiv(n, SCAN_BLOCK * SCAN_ITEMS) + 1), cap(n) {}
  void run(const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
    unsigned nb = cdiv(n, SCAN_BLOCK * SCAN_ITEMS);
    hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, out, block_sums.get(), (uint32_t)n);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(SCAN_BLOCK), 0, s, block_sums.get(), nb);
    hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(SCAN_BLOCK), 0, s, out, block_sums.get(), (uint32_t)n);
  }
};


}  // namespace zk
