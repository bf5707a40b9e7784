/* TEST INFRASTRUCTURE (see oracle.h).  Multi-scalar multiplication — body in msm_body.inc. */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
static size_t ceil_log2(size_t n) { size_t r = ((n & (n - 1)) == 0 ? 0 : 1); while (n > 1) { n >>= 1; r++; } return r; } /* FF/common/utils.cpp:32-45 */
#define G(x) g1_##x
#define M(x) msm_g1_##x
#define GT g1
#include "msm_body.inc"
#undef G
#undef M
#undef GT
#define G(x) g2_##x
#define M(x) msm_g2_##x
#define GT g2
#include "msm_body.inc"
