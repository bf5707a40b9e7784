// Fq on nine 29-bit limbs (Montgomery radix 2^261): the generated device arithmetic (gen_field29.py -> field29_gfx950.inc), shared by the H accumulation
// (msm.cuh) and the verifier's schedule kernel (pairing.cuh). Device compilation only.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
namespace zk {
#include "field29_gfx950.inc"
}
