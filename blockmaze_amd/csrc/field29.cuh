// Fq on nine 29-bit limbs (Montgomery radix 2^261): the generated device arithmetic (gen_field29.py -> field29_gfx950.inc), shared by the H accumulation
// (msm.cuh) and the verifier's schedule kernel (pairing.cuh). Device compilation only.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
namespace zk {
#include "field29_gfx950.inc"
// Wave priority of a kernel (gpu_internal.hpp: zk_prio_bits, switch ZK_PRIO): the host puts it into bits 24.. of a small integer argument; the kernel takes
// it out again and issues the matching s_setprio (an immediate operand).  Priority decides which of a SIMD's waves issues next: the short dependent chains of the
// witness MSMs and of the transforms compete with the H accumulation's four waves per SIMD.
__device__ __forceinline__ void zk_take_prio(uint32_t &arg) {
  const uint32_t prio = arg >> 24; arg &= 0xffffffu;
  if (prio == 1) __builtin_amdgcn_s_setprio(1); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else if (prio == 3) __builtin_amdgcn_s_setprio(3);
}

}
