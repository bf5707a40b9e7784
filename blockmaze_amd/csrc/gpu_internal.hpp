// Shared by the HIP translation units of libzkgpu (gpu.hip, gpu_msm_g1.hip, gpu_msm_g2.hip, gpu_keyops.hip): device context, error macro, scan helper, stage
// timer.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
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
// (the nine families are resolved ONCE, at first use, into a table indexed by an enumerator: a launch pays one array read, not three std::strings)
enum ZkPrioFamily { ZKP_NTT, ZKP_ROWS, ZKP_HSORT, ZKP_HTAIL, ZKP_STEP, ZKP_EXPAND, ZKP_WIT, ZKP_WLANES, ZKP_HACC, ZKP_FAMILIES };
inline uint32_t zk_prio_bits(ZkPrioFamily family) {
  struct Table { uint32_t bits[ZKP_FAMILIES]; Table() {
    static const char *const names[ZKP_FAMILIES] = {"ntt", "rows", "hsort", "htail", "step", "expand", "wit", "wlanes", "hacc"};
    static const int dflt[ZKP_FAMILIES] = {3, 3, 3, 3, 3, 3, 2, 2, 0};
    const char *e = getenv("ZK_PRIO"); const std::string s = e ? e : "";
    for (int f = 0; f < ZKP_FAMILIES; f++) {
      int v = dflt[f];
      if (s == "off") v = 0;
      else { const std::string key = std::string(names[f]) + ":"; const size_t at = s.find(key);
        if (at != std::string::npos && (at == 0 || s[at - 1] == ',')) v = atoi(s.c_str() + at + key.size()); }
      bits[f] = (uint32_t)(v < 0 ? 0 : v > 3 ? 3 : v) << 24;
    } } };
  static const Table t;
  return t.bits[family];
}
// the priority rides in bits 24.. of a small integer argument: a value that does not leave them free travels without one (the kernel then runs at the hardware's default)
inline uint32_t zk_with_prio(size_t arg, ZkPrioFamily family) { return (uint32_t)arg | (arg < ((size_t)1 << 24) ? zk_prio_bits(family) : 0u); }
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw GpuError(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
// A kernel that asks for more than 64 KB of dynamic LDS has its limit raised first — per DEVICE (the attribute belongs to the function as loaded on the current device: a
// process that serves several GPUs, ZK_DEVICES, must raise it on each of them), once: `mask` is the caller's static word of devices already done.
inline void zk_raise_dynamic_lds(const void *kernel, int bytes, std::atomic<uint64_t> &mask) {
  int dev = 0; HIP_CHECK(hipGetDevice(&dev)); const uint64_t bit = 1ull << (dev & 63);
  if (mask.load(std::memory_order_acquire) & bit) return;
  HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  mask.fetch_or(bit, std::memory_order_release);
}
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
