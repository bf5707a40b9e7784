// Multi-scalar multiplication  sum_i k_i * P_i  on CDNA4 — the prover's K3/K4/K5 kernels.
//
// Replaces libff's multi_exp / multi_exp_with_mixed_addition (FF/algebra/scalar_multiplication/multiexp.tcc:165-282,
// :443-496) and libsnark's kc_multi_exp_with_mixed_addition (SNARK/knowledge_commitment/kc_multiexp.tcc:21-85).
// Like the reference it splits the scalars three ways — zero: skipped; one: plain point sum; anything else: Pippenger
// buckets — but the structure is a GPU one:
//   tables     (key load, optional) 2^(cw) P for every window w: all windows then share ONE array of 2^(c-1) buckets (k_msm_precompute)
//   classify   one thread per (point, scalar): leave Montgomery form, drop zeros / points at infinity, append "ones" to a compacted index list (one atomic per
//              wave), histogram the signed c-bit digits (in LDS first when the bucket array is small)
//   plan       bucket offsets (scan), buckets ranked by decreasing size, tasks of at most 16 entries; one single-workgroup launch for small bucket arrays
//   scatter    counting sort of (table index, sign) by bucket
//   H query (uniform scalars, fixed-base table) has a path of its own: one-pass sort by workgroup-local binning (k_hsort_bin / k_hsort_group), accumulation
//              over
//              fixed-length runs of the sorted entries on nine 29-bit limbs (k_hacc_runs29), pieces added up per bucket (k_hacc_combine29), weighted bucket sum
//              by weight
//              bits from two-level marginal sums (htail29.cuh: k_hmarg29 / k_hbits29); an overflow of the sort's regions or a degenerate key sends the MSM back
//              to the general path below
//   accumulate one lane per task walks its slice of the sorted list with mixed additions (XYZZ accumulator in VGPRs, next point's gather in flight)
//   combine    buckets cut into several tasks: a quad (or a workgroup for very full buckets) adds the partial sums
//   reduce     sum_b b*B_b by segments: running sums inside a segment, a small scalar multiple for the segment offset, then workgroup-level trees — all with
//              quad-cooperative additions (curve.cuh): four lanes share one addition, so the dependent chain is 4 products long instead of 14
//   ones       one quad per strided partial sum over the compacted list + the same trees
// The bucket-array sums (one with tables, W otherwise) and the ones-sum go back to the host, which does the remaining c*W Horner doublings if there are any.
// Signed digits halve the bucket count: digit d in [-2^(c-1), 2^(c-1)], bucket |d|, sign applied to y on load.
#pragma once
#include <hip/hip_runtime.h>
#include "curve.cuh"

#include "field29.cuh"
namespace zk { struct MsmCounters { uint32_t n_ones; uint32_t n_other; uint32_t pad[2]; }; }
#include "htail29.cuh"
#include "oct29.cuh"
namespace zk {

constexpr int MSM_MAX_WINDOWS = 64;

__host__ __device__ inline int msm_num_windows(int c) { return 254 / c + 1; }

// Signed c-bit digits of a canonical 254-bit scalar.  Returns via callback-free loop in callers: digit(w) needs the carry
// chain, so decompose once into a small array.
__device__ __forceinline__ void signed_digits(const uint32_t k[8], int c, int W, int *dig) {
  uint32_t carry = 0; const uint32_t half = 1u << (c - 1), full = 1u << c, mask = full - 1;
  for (int w = 0; w < W; w++) {
    int bit = w * c, word = bit >> 5, sh = bit & 31;
    uint64_t v = word < 8 ? k[word] : 0; if (word + 1 < 8) v |= (uint64_t)k[word + 1] << 32;
    uint32_t d = ((uint32_t)(v >> sh) & mask) + carry;
    if (d > half) { dig[w] = (int)d - (int)full; carry = 1; } else { dig[w] = (int)d; carry = 0; }
  }
}

// The same digits for a compile-time window size, handed to fn(w, digit) one by one: after unrolling every limb index is a constant and the scalar stays in
// registers. (With a runtime window size k[] is indexed dynamically and, like the digit array above, lives in scratch memory: 300-560 bytes per lane in the
// first version of the sort kernels.) C = 0 selects the runtime path in the kernels below.
template <int C, class Fn> __device__ __forceinline__ void for_each_digit(const uint32_t (&k)[8], Fn &&fn) {
  constexpr int W = 254 / (C ? C : 1) + 1; constexpr uint32_t half = 1u << ((C ? C : 1) - 1), full = 1u << (C ? C : 1), mask = full - 1; uint32_t carry = 0;
#pragma unroll
  for (int w = 0; w < W; w++) {
    const int bit = w * C, word = bit >> 5, sh = bit & 31;
    uint32_t v = word < 8 ? k[word] >> sh : 0u; if (sh + C > 32 && word + 1 < 8) v |= k[word + 1] << (32 - sh);
    const uint32_t d = (v & mask) + carry; int dg; if (d > half) { dg = (int)d - (int)full; carry = 1; } else { dg = (int)d; carry = 0; }
    fn(w, dg);
  }
}
// fn(w, digit) for every window: the compiled-in walk for the window sizes in use (C > 0; the host checks c == C before it launches such an instantiation),
// otherwise the digit array of signed_digits (runtime c, W)
template <int C, class Fn> __device__ __forceinline__ void msm_walk_digits(const uint32_t (&k)[8], int c, int W, Fn &&fn) {
  if constexpr (C > 0) { for_each_digit<C>(k, [&](int w, int d) { fn(w, d); }); }
  else { int dig[MSM_MAX_WINDOWS]; signed_digits(k, c, W, dig); for (int w = 0; w < W; w++) fn(w, dig[w]); }
}
constexpr uint32_t MSM_ENTRY_SIGN = 0x80000000u;   // entry = table index | sign

// (struct MsmCounters { n_ones, n_other, pad[2] }: declared at the top — htail29.cuh needs it)

// scalars: Fr in Montgomery form.  scalar_index (optional): scalar for point i is scalars[scalar_index[i]] (sparse
// B-query, kc_multiexp.tcc:52-56); otherwise scalars[i]. point_is_inf (optional): byte flags of key points at infinity. Bucket arrays of at most MSM_LDS_HIST
// counters (the witness MSMs: 128 buckets once all windows share one array) are histogrammed / ranked in LDS per workgroup and touch the global counters once
// per non-empty bucket and workgroup: with ~60,000 entries on 128 counters the global atomics would serialise.
constexpr uint32_t MSM_LDS_HIST = 4096;
template <int C>
__global__ void __launch_bounds__(256) k_msm_classify(const Fr *__restrict__ scalars, const uint32_t *__restrict__ scalar_index,
    const uint8_t *__restrict__ point_is_inf,
                               uint32_t n, int c, int W, int filter_ones, uint32_t hist_stride, uint32_t n_hist, uint32_t *__restrict__ hist,
                                   uint32_t *__restrict__ ones, MsmCounters *cnt, MsmCounters *cnt_next) {
  __shared__ uint32_t lh[MSM_LDS_HIST]; const bool use_lds = n_hist <= MSM_LDS_HIST;
  if (blockIdx.x == 0 && threadIdx.x == 0) *cnt_next = MsmCounters{0, 0, {0, 0}};   // the counters alternate between two slots: this run clears the next run's
  if (use_lds) { for (uint32_t b = threadIdx.x; b < n_hist; b += blockDim.x) lh[b] = 0; __syncthreads(); }
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; bool live = i < n && !(point_is_inf && point_is_inf[i]); Fr k = Fr::zero();
  if (live) { k = scalars[scalar_index ? scalar_index[i] : i].from_mont(); live = !k.is_zero(); }
  bool is_one = false; if (live && filter_ones) { uint32_t o = k.l[0] ^ 1u; for (int j = 1; j < 8; j++) o |= k.l[j]; is_one = o == 0; }
  { // the compacted list of scalar-one indices: one atomic per wave instead of one per lane (46 % of a witness are ones, all on the same counter)
    uint64_t m = __ballot(is_one);
    if (m) {
      uint32_t lane = threadIdx.x & 63, base = 0;
      if (lane == (uint32_t)__ffsll((long long)m) - 1) base = atomicAdd(&cnt->n_ones, (uint32_t)__popcll(m));
      base = __shfl(base, __ffsll((long long)m) - 1, 64); if (is_one) ones[base + __popcll(m & ((1ull << lane) - 1))] = i; } }
  if (live && !is_one)
    // hist_stride = 2^(c-1): buckets per window; 0: all windows share one bucket array (precomputed 2^(cw) P)
    msm_walk_digits<C>(k.l, c, W, [&](int w, int d) { if (d) { uint32_t key = (uint32_t)w * hist_stride + (uint32_t)(d < 0 ? -d : d) - 1;
        atomicAdd(use_lds ? &lh[key] : &hist[key], 1u); } });
  {
    uint64_t m = __ballot(live && !is_one);
    if (m && (threadIdx.x & 63) == (uint32_t)__ffsll((long long)m) - 1) atomicAdd(&cnt->n_other, (uint32_t)__popcll(m));
  }
  if (use_lds) { __syncthreads(); for (uint32_t b = threadIdx.x; b < n_hist; b += blockDim.x) if (lh[b]) atomicAdd(&hist[b], lh[b]); }
}

template <int C>
__global__ void __launch_bounds__(256) k_msm_scatter(const Fr *__restrict__ scalars, const uint32_t *__restrict__ scalar_index,
    const uint8_t *__restrict__ point_is_inf,
                              uint32_t n, int c, int W, int filter_ones, uint32_t hist_stride, uint32_t point_stride, uint32_t n_hist,
                                  const uint32_t *__restrict__ offsets, uint32_t *__restrict__ fill, uint32_t *__restrict__ entries) {
  __shared__ uint32_t lcnt[MSM_LDS_HIST], lbase[MSM_LDS_HIST]; const bool use_lds = n_hist <= MSM_LDS_HIST;
  if (use_lds) { for (uint32_t b = threadIdx.x; b < n_hist; b += blockDim.x) lcnt[b] = 0; __syncthreads(); }
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; bool live = i < n && !(point_is_inf && point_is_inf[i]); Fr k = Fr::zero();
  if (live) { k = scalars[scalar_index ? scalar_index[i] : i].from_mont(); live = !k.is_zero(); }
  if (live && filter_ones) { uint32_t o = k.l[0] ^ 1u; for (int j = 1; j < 8; j++) o |= k.l[j]; if (o == 0) live = false; }
  // point_stride = n: the entry addresses 2^(cw) P_i in the precomputed table
  auto entry_of = [&](int w, int d) {
    return (i + (uint32_t)w * point_stride) | (d < 0 ? MSM_ENTRY_SIGN : 0u);
  };
  if (!use_lds) {
    if (live) msm_walk_digits<C>(k.l, c, W, [&](int w, int d) { if (!d) return; uint32_t key = (uint32_t)w * hist_stride + (uint32_t)(d < 0 ? -d : d) - 1;
        entries[offsets[key] + atomicAdd(&fill[key], 1u)] = entry_of(w, d); });
    return;
  }
  // small bucket arrays: the workgroup counts its entries per bucket in LDS, reserves its share of every bucket with one global atomic, then walks the digits a
  // second time to place them (ranks come from the LDS counters; nothing per entry is kept in registers or scratch between the two walks)
  if (live) msm_walk_digits<C>(k.l, c, W, [&](int w, int d) { if (d) atomicAdd(&lcnt[(uint32_t)w * hist_stride + (uint32_t)(d < 0 ? -d : d) - 1], 1u); });
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < n_hist; b += blockDim.x) {
    const uint32_t m = lcnt[b];
    lbase[b] = m ? offsets[b] + atomicAdd(&fill[b], m) : 0;
    lcnt[b] = 0;
  }
  __syncthreads();
  if (live) msm_walk_digits<C>(k.l, c, W, [&](int w, int d) { if (!d) return; uint32_t key = (uint32_t)w * hist_stride + (uint32_t)(d < 0 ? -d : d) - 1;
      entries[lbase[key] + atomicAdd(&lcnt[key], 1u)] = entry_of(w, d); });
}

// ---- H query: one-pass sort by workgroup-local binning ------------------------------------------------
// A one-pass sort with fixed slots per bucket pays one device-scope atomic and one isolated 4-byte store per digit (4.2 M of each for the send circuit: 0.34
// ms, 13.6x the algorithmic traffic; round 1). Here the 2^(c-1) buckets are cut into HSORT_GROUPS groups by their high bits, and the sort runs in two short
// kernels whose atomics are all in LDS:
//   k_hsort_bin a workgroup takes HSORT_TILE scalars (forming a*b*z on the way), counts its digits per group in LDS, reserves its share of every group's region
//                  with ONE device-scope atomic per group (65 K per MSM instead of 4.2 M), then writes each entry next to its workgroup-mates of the same
//                  group:
//                  runs of ~128 bytes per group and workgroup instead of isolated words.  Entry = low bucket bits | sign | table index.
//   k_hsort_group one workgroup per group: the group's entries (16 K for send) are counted per bucket in LDS, the counts are scanned, every entry moves to its
//                  final place inside a copy of the group's region IN LDS (dynamic, 4 x region bytes: 86 KB for send — one workgroup a CU anyway), and the copy
//                  leaves in consecutive words.  (Round 6: 1,024 lanes instead of 512 and the staged copy-out instead of one scattered 4-byte store per entry:
//                  31.6 -> 17.3 us, +0.9 % proofs/s on one box, tools/ab_lib_value.sh.)  Emits counts[] / offsets[] and the number of entries of the group.  The entries keep their low bucket bits: k_hacc_runs29 finds the bucket boundaries by them.
// Uniform scalars fill every group to within a few per cent of n*W/G, so a region holds 1.25x that; a region that would overflow raises a flag and the host
// repeats the MSM on the two-pass path (any input stays correct).
constexpr uint32_t HSORT_GROUPS = 1024 /* at most; the shape says how many are used */, HSORT_BIN_THREADS = 256, HSORT_PER_THREAD = 2,
    HSORT_TILE = HSORT_BIN_THREADS * HSORT_PER_THREAD, HSORT_GROUP_THREADS = 1024, HSORT_MAX_PER_THREAD = 24, HSORT_SLICES = 8,
    HSORT_STAGE_W = 20 /* staged entries per scalar: at most 254 / c + 1 digits, c >= 13 */;
// groups * 2^low_bits = buckets; bucket = group << low_bits | low; entry = low << (idx_bits + 1) | sign << idx_bits | index; region: entry slots per group
struct HsortShape { uint32_t groups, low_bits, idx_bits, region; };
template <int C>
__global__ void __launch_bounds__(HSORT_BIN_THREADS) k_hsort_bin(const Fr *__restrict__ scalars, const Fr *__restrict__ mul_b, const Fr *__restrict__ mul_z,
    int z_is_table, const uint8_t *__restrict__ point_is_inf,
                                                                 uint32_t n, int c, int W, uint32_t point_stride, HsortShape sh,
                                                                     uint32_t *__restrict__ group_fill, uint32_t *__restrict__ mid, MsmCounters *cnt,
                                                                     MsmCounters *cnt_next) {
  zk_take_prio(sh.low_bits);
  __shared__ uint32_t lcnt[HSORT_GROUPS], lpos[HSORT_GROUPS], gbase[HSORT_GROUPS];
  __shared__ uint32_t stage[HSORT_TILE * HSORT_STAGE_W];
  __shared__ uint16_t stage_g[HSORT_TILE * HSORT_STAGE_W];
  __shared__ uint32_t wave_tot[HSORT_BIN_THREADS / 64];
  if (blockIdx.x == 0 && threadIdx.x == 0) *cnt_next = MsmCounters{0, 0, {0, 0}};
  for (uint32_t g = threadIdx.x; g < sh.groups; g += blockDim.x) lcnt[g] = 0;
  __syncthreads();
  Fr k[HSORT_PER_THREAD]; bool live[HSORT_PER_THREAD];
#pragma unroll
  for (int q = 0; q < (int)HSORT_PER_THREAD; q++) {
    const uint32_t i = blockIdx.x * HSORT_TILE + q * HSORT_BIN_THREADS + threadIdx.x;
    live[q] = i < n && !(point_is_inf && point_is_inf[i]);
    k[q] = Fr::zero();
    if (live[q]) { Fr v = scalars[i]; if (mul_b) v = v * mul_b[i] * mul_z[z_is_table ? i : 0]; k[q] = v.from_mont(); live[q] = !k[q].is_zero(); } }
  const uint32_t low_mask = (1u << sh.low_bits) - 1;
#pragma unroll
  for (int q = 0; q < (int)HSORT_PER_THREAD; q++) if (live[q]) msm_walk_digits<C>(k[q].l, c, W, [&](int, int d) {
      if (d) atomicAdd(&lcnt[((uint32_t)(d < 0 ? -d : d) - 1) >> sh.low_bits], 1u); });
  __syncthreads();
  // exclusive scan of the group counts (where each group's run starts in the staging tile) and the reservation of the runs in the groups' regions
  {
    uint32_t s = 0;
    const uint32_t per = (sh.groups + HSORT_BIN_THREADS - 1) / HSORT_BIN_THREADS, lo = threadIdx.x * per;
    for (uint32_t j = 0; j < per; j++) if (lo + j < sh.groups) s += lcnt[lo + j];
    uint32_t inc = s; for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(inc, d, 64); if ((int)(threadIdx.x & 63) >= d) inc += t; }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t ex = inc - s; for (uint32_t wv = 0; wv < (threadIdx.x >> 6); wv++) ex += wave_tot[wv];
    bool over = false;
    for (uint32_t j = 0; j < per; j++) if (lo + j < sh.groups) {
      const uint32_t g = lo + j, m = lcnt[g];
      lpos[g] = ex;
      ex += m;
      const uint32_t b = m ? atomicAdd(&group_fill[g], m) : 0;
      if (b + m > sh.region) over = true;
      gbase[g] = b;
    }
    if (over) atomicOr(&cnt->pad[0], 1u); }
  __syncthreads();
  for (uint32_t g = threadIdx.x; g < sh.groups; g += blockDim.x) lcnt[g] = lpos[g];       // running write position of every group inside the staging tile
  __syncthreads();
#pragma unroll
  for (int q = 0; q < (int)HSORT_PER_THREAD; q++) if (live[q]) { const uint32_t i = blockIdx.x * HSORT_TILE + q * HSORT_BIN_THREADS + threadIdx.x;
    msm_walk_digits<C>(k[q].l, c, W, [&](int w, int d) { if (!d) return; const uint32_t key = (uint32_t)(d < 0 ? -d : d) - 1, g = key >> sh.low_bits,
        p = atomicAdd(&lcnt[g], 1u);
      if (p < HSORT_TILE * HSORT_STAGE_W) { stage[p] = ((key & low_mask) << (sh.idx_bits + 1)) | ((d < 0 ? 1u : 0u) << sh.idx_bits) | (i +
          (uint32_t)w * point_stride); stage_g[p] = (uint16_t)g; } }); }
  __syncthreads();
  // copy-out: consecutive lanes write consecutive words of a group's run (about 128 bytes per group and workgroup) instead of one isolated word per digit
  const uint32_t total = min(lcnt[sh.groups - 1], HSORT_TILE * HSORT_STAGE_W);
  for (uint32_t p = threadIdx.x; p < total; p += blockDim.x) {
    const uint32_t g = stage_g[p], pos = gbase[g] + (p - lpos[g]);
    if (pos < sh.region) mid[(size_t)g * sh.region + pos] = stage[p];
  }
}
static __global__ void __launch_bounds__(HSORT_GROUP_THREADS) k_hsort_group(const uint32_t *__restrict__ mid, uint32_t *__restrict__ group_fill, HsortShape sh,
    uint32_t *__restrict__ entries, uint32_t *__restrict__ counts, uint32_t *__restrict__ offsets,
                                                                            uint32_t *__restrict__ group_n) {   // group_n[g] = entries of group g
  zk_take_prio(sh.low_bits);
  __shared__ uint32_t lcnt[1024], lpre[1024];                        // 2^low_bits <= 1024 buckets per group
  const uint32_t g = blockIdx.x, nb = 1u << sh.low_bits, n_g = min(group_fill[g], sh.region); const uint32_t *src = mid + (size_t)g * sh.region;
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) lcnt[b] = 0;
  __syncthreads();
  // (read above by every thread of this workgroup only, before the barrier) cleared for the next run
  if (threadIdx.x == 0) {
    group_fill[g] = 0;
    group_n[g] = n_g;
  }
  uint32_t e[HSORT_MAX_PER_THREAD]; int ne = 0;
#pragma unroll
  for (int j = 0; j < (int)HSORT_MAX_PER_THREAD; j++) {
    const uint32_t p = (uint32_t)j * HSORT_GROUP_THREADS + threadIdx.x;
    if (p < n_g) {
      e[j] = src[p];
      atomicAdd(&lcnt[e[j] >> (sh.idx_bits + 1)], 1u);
      ne = j + 1;
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) {                                           // exclusive scan of the bucket counts by one wave (nb / 64 consecutive buckets per lane)
    const uint32_t per = (nb + 63) / 64, lo = threadIdx.x * per; uint32_t s = 0; for (uint32_t j = 0; j < per; j++) if (lo + j < nb) s += lcnt[lo + j];
    uint32_t inc = s; for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(inc, d, 64); if ((int)threadIdx.x >= d) inc += t; }
    uint32_t ex = inc - s; for (uint32_t j = 0; j < per; j++) if (lo + j < nb) { lpre[lo + j] = ex; ex += lcnt[lo + j]; } }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < nb; b += blockDim.x) {
    const uint32_t bucket = (g << sh.low_bits) | b;
    counts[bucket] = lcnt[b];
    offsets[bucket] = g * sh.region + lpre[b];
    lcnt[b] = 0;
  }
  __syncthreads();
  uint32_t *dst = entries + (size_t)g * sh.region;
  extern __shared__ uint32_t hsg_stage[];                            // (dynamic: region words) the group in bucket order, then written out in consecutive words
#pragma unroll
  for (int j = 0; j < (int)HSORT_MAX_PER_THREAD; j++) if (j < ne) {
    const uint32_t v = e[j], b = v >> (sh.idx_bits + 1), r = atomicAdd(&lcnt[b], 1u);
    hsg_stage[lpre[b] + r] = v;
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < n_g; p += blockDim.x) dst[p] = hsg_stage[p];
}

static __global__ void k_fr_mul3(const Fr *__restrict__ a, const Fr *__restrict__ b, const Fr *__restrict__ z, int z_is_table, uint32_t n,
    Fr *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = a[i] * b[i] * z[z_is_table ? i : 0];
}

// ---- exclusive scan over uint32 (three small kernels; the arrays are <= 2^21 entries) ------------------------------
constexpr int SCAN_BLOCK = 1024, SCAN_ITEMS = 4;   // 4096 items per block
static __global__ void k_scan_local(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t *__restrict__ block_sums, uint32_t n) {
  __shared__ uint32_t sh[SCAN_BLOCK]; uint32_t base = blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + threadIdx.x * SCAN_ITEMS; uint32_t v[SCAN_ITEMS], s = 0;
  for (int j = 0; j < SCAN_ITEMS; j++) { v[j] = base + j < n ? in[base + j] : 0; s += v[j]; }
  sh[threadIdx.x] = s; __syncthreads();
  for (int d = 1; d < SCAN_BLOCK; d <<= 1) { uint32_t t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0; __syncthreads(); sh[threadIdx.x] += t; __syncthreads(); }
  uint32_t excl = sh[threadIdx.x] - s;
  for (int j = 0; j < SCAN_ITEMS; j++) { if (base + j < n) out[base + j] = excl; excl += v[j]; }
  if (threadIdx.x == SCAN_BLOCK - 1) block_sums[blockIdx.x] = sh[SCAN_BLOCK - 1];
}
static __global__ void k_scan_block_sums(uint32_t *block_sums, uint32_t nblocks) {   // single block, nblocks <= SCAN_BLOCK*? handled by loop
  __shared__ uint32_t sh[SCAN_BLOCK]; uint32_t carry = 0;
  for (uint32_t base = 0; base < nblocks; base += SCAN_BLOCK) {
    uint32_t i = base + threadIdx.x, v = i < nblocks ? block_sums[i] : 0; sh[threadIdx.x] = v; __syncthreads();
    for (int d = 1; d < SCAN_BLOCK; d <<= 1) {
      uint32_t t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) block_sums[i] = carry + sh[threadIdx.x] - v; uint32_t tot = sh[SCAN_BLOCK - 1]; __syncthreads(); carry += tot;
  }
}
static __global__ void k_scan_add(uint32_t *__restrict__ out, const uint32_t *__restrict__ block_sums, uint32_t n) {
  uint32_t base = blockIdx.x * SCAN_BLOCK * SCAN_ITEMS + threadIdx.x * SCAN_ITEMS, add = block_sums[blockIdx.x];
  for (int j = 0; j < SCAN_ITEMS; j++) if (base + j < n) out[base + j] += add;
}

// ---- wave-level helpers ------------------------------------------------------------------------------------------------
// (member by member: a reinterpret_cast of the struct's address would force the point into scratch memory)
template <class P> __device__ __forceinline__ Fp<P> shfl_down_struct(const Fp<P> &v, int delta) {
  Fp<P> r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = __shfl_down(v.l[i], delta, 64);
  return r;
}
__device__ __forceinline__ Fq2 shfl_down_struct(const Fq2 &v, int delta) { return {shfl_down_struct(v.c0, delta), shfl_down_struct(v.c1, delta)}; }
template <class F> __device__ __forceinline__ XYZZ<F> shfl_down_struct(const XYZZ<F> &v, int delta) {
  return {shfl_down_struct(v.X, delta), shfl_down_struct(v.Y, delta), shfl_down_struct(v.ZZ, delta), shfl_down_struct(v.ZZZ, delta)};
}

// ---- bucket accumulation, balanced by entries ---------------------------------------------------------------------------
// A bucket with `count` sorted entries is cut into tasks of at most TASK entries; one lane per task, whatever the digit
// distribution looks like (uniform H coefficients, the short top window, 0/1-heavy witnesses).  task_off is the exclusive
// scan of the per-bucket task counts (n_buckets + 1 entries, the last one = total).
constexpr uint32_t MSM_TASK = 16;
// Buckets are visited in order of decreasing entry count, so that the 64 lanes of a wave walk slices of (nearly) equal length: with Poisson-distributed
// bucket sizes the longest of 64 random buckets is about twice the mean, and every lane of the wave would wait for it.
constexpr uint32_t BSORT_CLASSES = 64, BSORT_BLOCK = 256;
__device__ __forceinline__ uint32_t bsort_class(uint32_t count) { return BSORT_CLASSES - 1 - min(count, BSORT_CLASSES - 1); }   // class 0 = the fullest buckets
static __global__ void __launch_bounds__(BSORT_BLOCK) k_bsort_hist(uint32_t *__restrict__ counts, uint32_t n_buckets, uint32_t n_blocks,
    uint32_t *__restrict__ block_hist, uint32_t clip) {
  __shared__ uint32_t h[BSORT_CLASSES]; if (threadIdx.x < BSORT_CLASSES) h[threadIdx.x] = 0; __syncthreads();
  // clip: slot capacity of the one-pass sort (an overflow is flagged there)
  uint32_t b = blockIdx.x * BSORT_BLOCK + threadIdx.x;
  if (b < n_buckets) {
    uint32_t cv = counts[b];
    if (clip && cv > clip) {
      cv = clip;
      counts[b] = cv;
    }
    atomicAdd(&h[bsort_class(cv)], 1u);
  }
  __syncthreads();
  // class-major, so one exclusive scan yields every (class, block) base
  if (threadIdx.x < BSORT_CLASSES) block_hist[threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}
static __global__ void __launch_bounds__(BSORT_BLOCK) k_bsort_scatter(const uint32_t *__restrict__ counts, uint32_t n_buckets, uint32_t n_blocks,
    const uint32_t *__restrict__ block_off,
                                                               uint32_t *__restrict__ order, uint32_t *__restrict__ rank_of, uint32_t *__restrict__ ntasks,
                                                                   uint32_t *__restrict__ cls_start, uint4 *__restrict__ bucket_mem, uint32_t bucket_u4) {
  __shared__ uint32_t h[BSORT_CLASSES]; if (threadIdx.x < BSORT_CLASSES) h[threadIdx.x] = 0; __syncthreads();
  uint32_t b = blockIdx.x * BSORT_BLOCK + threadIdx.x;
  if (b < n_buckets) {
    uint32_t cnt = counts[b], cls = bsort_class(cnt), pos = block_off[cls * n_blocks + blockIdx.x] + atomicAdd(&h[cls], 1u);
    order[pos] = b;
    rank_of[b] = pos;
    ntasks[pos] = (cnt + MSM_TASK - 1) / MSM_TASK;
    // empty bucket = point at infinity (all-zero record)
    if (cnt == 0) for (uint32_t j = 0; j < bucket_u4; j++) bucket_mem[(size_t)b * bucket_u4 + j] = make_uint4(0, 0, 0, 0);
  }
  if (b == 0) ntasks[n_buckets] = 0;
  // rank of the first bucket of each size class
  if (blockIdx.x == 0 && threadIdx.x < BSORT_CLASSES) cls_start[threadIdx.x] = block_off[threadIdx.x * n_blocks];
}

// The same planning for a small bucket array (witness MSMs: 32 windows x 128 buckets) in ONE launch of one workgroup instead of eleven launches:
// offsets = exclusive scan of the histogram; order / rank_of = buckets by decreasing size class; task_off = exclusive scan of the task counts in that order.
constexpr uint32_t PLAN_SMALL_MAX = 16384, PLAN_THREADS = 1024;
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *sh, uint32_t *total) {
  sh[threadIdx.x] = v; __syncthreads();
  for (int d = 1; d < (int)PLAN_THREADS; d <<= 1) {
    uint32_t t = threadIdx.x >= (uint32_t)d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t r = sh[threadIdx.x] - v; *total = sh[PLAN_THREADS - 1]; __syncthreads(); return r;
}
static __global__ void __launch_bounds__(PLAN_THREADS) k_msm_plan_small(const uint32_t *__restrict__ counts, uint32_t n_buckets,
    uint32_t *__restrict__ offsets, uint32_t *__restrict__ order, uint32_t *__restrict__ rank_of,
                                                                  uint32_t *__restrict__ task_off, uint32_t *__restrict__ cls_start,
                                                                      uint4 *__restrict__ bucket_mem, uint32_t bucket_u4) {
  __shared__ uint32_t sh[PLAN_THREADS]; __shared__ uint32_t cls_cnt[BSORT_CLASSES], cls_base[BSORT_CLASSES];
  const uint32_t per = (n_buckets + PLAN_THREADS - 1) / PLAN_THREADS, lo = threadIdx.x * per; uint32_t s = 0, total;
  for (uint32_t j = 0; j < per; j++) if (lo + j < n_buckets) s += counts[lo + j];
  uint32_t ex = block_exclusive_scan_1024(s, sh, &total);
  for (uint32_t j = 0; j < per; j++) if (lo + j < n_buckets) { offsets[lo + j] = ex; ex += counts[lo + j]; }
  if (threadIdx.x < BSORT_CLASSES) cls_cnt[threadIdx.x] = 0; __syncthreads();
  for (uint32_t j = 0; j < per; j++) if (lo + j < n_buckets) { uint32_t b = lo + j, cnt = counts[b]; rank_of[b] = atomicAdd(&cls_cnt[bsort_class(cnt)], 1u);
    if (cnt == 0) for (uint32_t q = 0; q < bucket_u4; q++) bucket_mem[(size_t)b * bucket_u4 + q] = make_uint4(0, 0, 0, 0); }
  __syncthreads();
  if (threadIdx.x == 0) { uint32_t a = 0; for (uint32_t c = 0; c < BSORT_CLASSES; c++) { cls_base[c] = a; cls_start[c] = a; a += cls_cnt[c]; } }
  __syncthreads();
  for (uint32_t j = 0; j < per; j++) if (lo + j < n_buckets) {
    uint32_t b = lo + j, pos = cls_base[bsort_class(counts[b])] + rank_of[b];
    rank_of[b] = pos;
    order[pos] = b;
  }
  __threadfence_block(); __syncthreads();
  s = 0; for (uint32_t j = 0; j < per; j++) if (lo + j < n_buckets) s += (counts[order[lo + j]] + MSM_TASK - 1) / MSM_TASK;
  ex = block_exclusive_scan_1024(s, sh, &total);
  for (uint32_t j = 0; j < per; j++) if (lo + j < n_buckets) { task_off[lo + j] = ex; ex += (counts[order[lo + j]] + MSM_TASK - 1) / MSM_TASK; }
  if (threadIdx.x == 0) task_off[n_buckets] = total;
}
// task_off: exclusive scan of ntasks over the SORTED bucket list (n_buckets + 1 entries, the last one = total)
// FROM261 (round 5): the table holds coordinates x 2^261 (the form the 29-bit kernels gather from; the x 2^256 copy of a query's fixed-base table is no longer kept
// beside it) — every loaded coordinate is multiplied by 2^251 (a Montgomery product: x 2^261 2^251 2^-256 = x 2^256).  (0, 0), the point at infinity, stays (0, 0).
template <class F> __device__ __forceinline__ Affine<F> affine_from_r261(Affine<F> p) {
  Fq c;
#pragma unroll
  for (int j = 0; j < 8; j++) c.l[j] = FQ_TWO251[j];
  if constexpr (sizeof(F) == 32) { p.x = p.x * c; p.y = p.y * c; }
  else { p.x = {p.x.c0 * c, p.x.c1 * c}; p.y = {p.y.c0 * c, p.y.c1 * c}; }
  return p;
}
template <class F, int FROM261>
__global__ void __launch_bounds__(256) k_msm_accumulate_tasks(const Affine<F> *__restrict__ points, const uint32_t *__restrict__ entries,
    const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ counts,
                                                              const uint32_t *__restrict__ order, const uint32_t *__restrict__ task_off, uint32_t n_buckets,
                                                                  uint32_t max_tasks, uint32_t task, XYZZ<F> *__restrict__ buckets,
                                                                  XYZZ<F> *__restrict__ partials) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; if (t >= max_tasks || t >= task_off[n_buckets]) return;
  uint32_t lo = 0, hi = n_buckets;                         // largest i with task_off[i] <= t
  while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (task_off[mid] <= t) lo = mid; else hi = mid; }
  uint32_t b = order[lo], j = t - task_off[lo], cnt = counts[b], beg = offsets[b] + j * task, end = offsets[b] + min(cnt, (j + 1) * task);
  XYZZ<F> acc = XYZZ<F>::inf();
  const uint32_t imask = ~MSM_ENTRY_SIGN;
  uint32_t v = entries[beg], vn = beg + 1 < end ? entries[beg + 1] : v; Affine<F> p = points[v & imask];   // (beg < end: tasks exist only for non-empty slices)
  // software pipeline: the next point and the entry after it are in flight while this point is added (random table gathers)
#pragma unroll 1
  for (uint32_t e = beg; e < end; e++) {
    Affine<F> pn = points[vn & imask]; uint32_t vnn = e + 2 < end ? entries[e + 2] : vn;
    if constexpr (FROM261 != 0) p = affine_from_r261(p);
    if (v >> 31) p.y = p.y.neg(); acc.madd_inl(p); v = vn; p = pn; vn = vnn; }
  if (cnt <= task) buckets[b] = acc; else partials[t] = acc;
}
// ---- H query accumulation over fixed-length runs, on nine 29-bit limbs (round 3)
// --------------------------------------------------------------------------------- Round 2 cut every bucket into the same NUMBER of slices; bucket sizes are
// Poisson (128 +- 11 for send), so the 64 lanes of a wave held slices of different lengths and the wave ran for the longest. Here a lane takes a RUN of `run`
// consecutive entries of its group's sorted list, wherever the bucket boundaries fall: every lane of the chip does the same number of additions. A lane sees a
// boundary as a change of the entries' low bucket bits: it stores the sum so far as piece (this run - the bucket's first run) of the old bucket, and starts
// over. Bucket b's pieces sit at partials[b * maxp + 0 .. np): no plan, no atomics; a bucket with more than maxp pieces (non-uniform scalars) raises the
// overflow flag of the one-pass sort and the host repeats the MSM on the two-pass path. v_addc_co_u32 / v_subb_co_u32 issue at the rate of v_mad_u64_u32 on
// gfx950 (tools/valu_probe.hip: 32 T lane-ops/s, v_add_u32 61 T), so the carry additions of a product on 8 x 32-bit limbs cost as much as its multiplications:
// the 32-bit form of this loop measured 12.3 k cycles per mixed addition (0.38 ms). On 29-bit limbs (field29_gfx950.inc: Montgomery radix 2^261) a product is
// 162 v_mad_u64_u32 and NO carry instruction, a difference nine 32-bit operations and a parallel carry step. The table this kernel gathers from holds the same
// points as the fixed-base table with coordinates x * 2^261 instead of x * 2^256 (k_table_to_r261 at key load: one product per coordinate; canonical, 8 words
// each, so the record size and the gather pattern do not change); limbs are unpacked after the load. A piece is stored as its 36 limbs; k_hacc_combine29 adds a
// bucket's pieces on 29-bit limbs too and converts the sum (one product per coordinate with 2^256 mod p) to the 8 x 32-bit form the weighted bucket sum reads.
// Operand = +-accumulator (P = 0) is not looked for in the loop: it leaves ZZ = 0 (mod p) for good, which the combine notices and reports like an overflow of
// the sort.
struct XYZZ29 { Fq29 X, Y, ZZ, ZZZ;
  // madd-2008-s in two steps, so that the caller can start the NEXT point's gather between them, into the registers this point's coordinates just left (its
  // words are dead after the first two products; loading the next point at the top of the loop instead cost 16 registers and with them the fourth wave per
  // SIMD). px normalized, py possibly K_2 - y (limbs below 2^31); bounds: gen_field29.py
  __device__ __forceinline__ void madd_head(const Fq29 &px, const Fq29 &py, Fq29 &Pv, Fq29 &Rv) const {
    Pv = Fq29::sub<6>(Fq29::mul(px, ZZ), X);
    Rv = Fq29::sub<4>(Fq29::mul(py, ZZZ), Y);
  }
  __device__ __forceinline__ void madd_tail(const Fq29 &Pv, const Fq29 &Rv) { madd_tail_pp(Pv, Rv, Fq29::sqr(Pv)); }
  __device__ __forceinline__ void madd_tail_pp(const Fq29 &Pv, const Fq29 &Rv, const Fq29 &PP) {
    const Fq29 PPP = Fq29::mul(Pv, PP), Q = Fq29::mul(X, PP);
    Fq29 s;
#pragma unroll
    for (int i = 0; i < 9; i++) s.l[i] = PPP.l[i] + 2u * Q.l[i];
    const Fq29 X3 = Fq29::sub<4>(Fq29::sqr(Rv), s);
    // Y3 = R (Q - X3) + (K_4 - Y) PPP as ONE dual product (round 5; Fq29::mul2: one Montgomery reduction for both halves — 81 multiply-adds and nine m_k fewer per
    // mixed addition, proof window -5 us; operands normalized, bounds: gen_field29.py, check_bounds_dual_y)
    ZZ = Fq29::mul(ZZ, PP); ZZZ = Fq29::mul(ZZZ, PPP);
    Fq29 zero;
#pragma unroll
    for (int i = 0; i < 9; i++) zero.l[i] = 0;
    Y = Fq29::mul2(Rv, Fq29::sub<6>(Q, X3), Fq29::sub<4>(zero, Y), PPP); X = X3;
  }
};
// a product's or a squaring's result (exact 29-bit limbs, value below 2 p) is 0 (mod p) iff it is 0 or p
__device__ __forceinline__ bool fq29_product_is_zero(const Fq29 &t) {
  uint32_t zero_or = 0, p_xor = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) { zero_or |= t.l[i]; p_xor |= t.l[i] ^ Fq29::P29[i]; }
  return zero_or == 0 || p_xor == 0;
}
// 2 (x, y) for an affine point (mdbl-2008-s-1, a = 0): x canonical, y canonical or K_2 - y with normalized limbs. The result keeps to the invariant of madd (X
// < 5.1 p, Y < 3.2 p, ZZ, ZZZ < 1.1 p; gen_field29.py: check_bounds_dbl). Used where a witness MSM meets the same point twice in a row (see k_wacc_lanes29).
__device__ __forceinline__ XYZZ29 xyzz29_dbl_affine(const Fq29 &x, const Fq29 &y) {
  Fq29 U, M3;
#pragma unroll
  for (int i = 0; i < 9; i++) U.l[i] = 2u * y.l[i];
  U = U.norm();
  const Fq29 V = Fq29::sqr(U), W = Fq29::mul(U, V), S = Fq29::mul(x, V), xx = Fq29::sqr(x);
#pragma unroll
  for (int i = 0; i < 9; i++) M3.l[i] = 3u * xx.l[i];
  const Fq29 M = M3.norm();
  Fq29 S2;
#pragma unroll
  for (int i = 0; i < 9; i++) S2.l[i] = 2u * S.l[i];
  XYZZ29 r;
  r.X = Fq29::sub<4>(Fq29::sqr(M), S2);
  r.Y = Fq29::sub<2>(Fq29::mul(M, Fq29::sub<6>(S, r.X)), Fq29::mul(W, y));
  r.ZZ = V; r.ZZZ = W;
  return r;
}
// add-2008-s on two accumulators within the invariant of madd (gen_field29.py: check_bounds_add); neither at infinity, and not +-each other (that leaves ZZ = 0
// mod p)
__device__ __forceinline__ XYZZ29 xyzz29_add(const XYZZ29 &a, const XYZZ29 &b) {
  const Fq29 U1 = Fq29::mul(a.X, b.ZZ), S1 = Fq29::mul(a.Y, b.ZZZ), Pv = Fq29::sub<2>(Fq29::mul(b.X, a.ZZ), U1), Rv = Fq29::sub<2>(Fq29::mul(b.Y, a.ZZZ), S1);
  const Fq29 PP = Fq29::sqr(Pv), PPP = Fq29::mul(Pv, PP), Q = Fq29::mul(U1, PP);
  Fq29 s;
#pragma unroll
  for (int i = 0; i < 9; i++) s.l[i] = PPP.l[i] + 2u * Q.l[i];
  XYZZ29 r; r.X = Fq29::sub<4>(Fq29::sqr(Rv), s); r.Y = Fq29::sub<2>(Fq29::mul(Rv, Fq29::sub<6>(Q, r.X)), Fq29::mul(S1, PPP));
  r.ZZ = Fq29::mul(Fq29::mul(a.ZZ, b.ZZ), PP); r.ZZZ = Fq29::mul(Fq29::mul(a.ZZZ, b.ZZZ), PPP); return r;
}
// A piece is stored as it is — 36 limbs; all-zero ZZ limbs mark the point at infinity — and converted by k_hacc_combine29 after the pieces of a bucket have
// been added up. (Converting here, four products per piece, looked cheap per lane and was not per wave: with 64 lanes and buckets of ~11 runs some lane reaches
// a bucket boundary in nearly every iteration, so the whole wave walked the conversion code every time: 0.366 ms, no faster than the 32-bit loop.)
struct Piece29 { uint32_t w[36]; };
__device__ __forceinline__ void hacc_flush29(const XYZZ29 &acc, bool inf, uint32_t bucket, uint32_t t, uint32_t g, const uint32_t *__restrict__ offsets,
    const HsortShape &sh, uint32_t run, uint32_t maxp,
                                             Piece29 *__restrict__ partials, MsmCounters *cnt) {
  // which of the bucket's pieces this lane holds: runs since the bucket's first entry
  const uint32_t piece = t - (offsets[bucket] - g * sh.region) / run;
  if (piece >= maxp) { atomicOr(&cnt->pad[0], 1u); return; }
  uint4 *dst = reinterpret_cast<uint4 *>(partials + (size_t)bucket * maxp + piece);
  const uint32_t z = inf ? 0u : ~0u;                                                     // all-zero ZZ limbs = the point at infinity
  const Fq29 &X = acc.X, &Y = acc.Y, &ZZ = acc.ZZ, &ZZZ = acc.ZZZ;
  dst[0] = make_uint4(X.l[0], X.l[1], X.l[2], X.l[3]);
  dst[1] = make_uint4(X.l[4], X.l[5], X.l[6], X.l[7]);
  dst[2] = make_uint4(X.l[8], Y.l[0], Y.l[1], Y.l[2]);
  dst[3] = make_uint4(Y.l[3], Y.l[4], Y.l[5], Y.l[6]);
  dst[4] = make_uint4(Y.l[7], Y.l[8], ZZ.l[0] & z, ZZ.l[1] & z);
  dst[5] = make_uint4(ZZ.l[2] & z, ZZ.l[3] & z, ZZ.l[4] & z, ZZ.l[5] & z);
  dst[6] = make_uint4(ZZ.l[6] & z, ZZ.l[7] & z, ZZ.l[8] & z, ZZZ.l[0]);
  dst[7] = make_uint4(ZZZ.l[1], ZZZ.l[2], ZZZ.l[3], ZZZ.l[4]);
  dst[8] = make_uint4(ZZZ.l[5], ZZZ.l[6], ZZZ.l[7], ZZZ.l[8]);
}
__device__ __forceinline__ XYZZ29 piece29_load(const Piece29 *p, bool &inf) {
  const uint4 *s = reinterpret_cast<const uint4 *>(p);
  uint32_t w[36];
#pragma unroll
  for (int i = 0; i < 9; i++) { const uint4 v = s[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
  XYZZ29 r;
  uint32_t zz_or = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) { r.X.l[i] = w[i]; r.Y.l[i] = w[9 + i]; r.ZZ.l[i] = w[18 + i]; r.ZZZ.l[i] = w[27 + i]; zz_or |= w[18 + i]; }
  inf = zz_or == 0;
  return r;
}
template <int ANY_INF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_hacc_runs29(const Affine<Fq> *__restrict__ points261, const uint32_t *__restrict__ entries, const uint32_t *__restrict__ group_n,
    const uint32_t *__restrict__ offsets, HsortShape sh, uint32_t run, uint32_t maxp,
              Piece29 *__restrict__ partials, MsmCounters *cnt) {
  zk_take_prio(maxp);
  // which run is this lane's?  base[g] = the number of runs in the groups before g: an exclusive scan of ceil(n_g / run), recomputed by every workgroup
  __shared__ uint32_t base[HSORT_GROUPS + 1], wave_tot[4];
  {
    const uint32_t per = (sh.groups + 255) / 256, lo = threadIdx.x * per;
    uint32_t s = 0;
    for (uint32_t j = 0; j < per; j++) if (lo + j < sh.groups) s += (min(group_n[lo + j], sh.region) + run - 1) / run;
    uint32_t inc = s;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d, 64); if ((int)(threadIdx.x & 63) >= d) inc += u; }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t ex = inc - s;
    for (uint32_t wv = 0; wv < (threadIdx.x >> 6); wv++) ex += wave_tot[wv];
    for (uint32_t j = 0; j < per; j++) if (lo + j < sh.groups) { base[lo + j] = ex; ex += (min(group_n[lo + j], sh.region) + run - 1) / run; }
    if (threadIdx.x == 255) base[sh.groups] = ex;
    __syncthreads();
  }
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= base[sh.groups]) return;
  uint32_t glo = 0, ghi = sh.groups;                                                     // the group g with base[g] <= r < base[g + 1]
  while (ghi - glo > 1) { const uint32_t mid = (glo + ghi) >> 1; if (base[mid] <= r) glo = mid; else ghi = mid; }
  const uint32_t g = glo, t = r - base[g], n_g = min(group_n[g], sh.region), beg = t * run, end = min(beg + run, n_g);
  const uint32_t idx_mask = (1u << sh.idx_bits) - 1, shift = sh.idx_bits + 1;            // entry = low bucket bits | sign | table index
  const uint32_t *e = entries + (size_t)g * sh.region;
  uint32_t v = e[beg], vn = beg + 1 < end ? e[beg + 1] : v, cur = v >> shift;
  Affine<Fq> p = points261[v & idx_mask];
  bool inf = true;
  XYZZ29 acc;
#pragma unroll
  for (int i = 0; i < 9; i++) { acc.X.l[i] = 0; acc.Y.l[i] = 0; acc.ZZ.l[i] = 0; acc.ZZZ.l[i] = 0; }
#pragma unroll 1
  for (uint32_t i = beg; i < end; i++) {
    // software pipeline: the next point's gather is in flight during this addition
    const Affine<Fq> pn = points261[vn & idx_mask];
    const uint32_t vnn = i + 2 < end ? e[i + 2] : vn, low = v >> shift;
    // a bucket boundary inside the run: the sum so far is a piece of the bucket left behind
    if (low != cur) {
      hacc_flush29(acc, inf, (g << sh.low_bits) | cur, t, g, offsets, sh, run, maxp, partials, cnt);
      cur = low; inf = true;
    }
    if (ANY_INF && p.is_inf()) {                                                         // a key point at infinity adds nothing
    } else {
      const Fq29 px = Fq29::unpack(p.x.l), py = Fq29::cond_neg(Fq29::unpack(p.y.l), (v >> sh.idx_bits) & 1u);
      if (inf) { acc.X = px; acc.Y = py.norm(); acc.ZZ = Fq29::one(); acc.ZZZ = Fq29::one(); inf = false; }
      else { Fq29 Pv, Rv; acc.madd_head(px, py, Pv, Rv); acc.madd_tail(Pv, Rv); }
    }
    v = vn; p = pn; vn = vnn;
  }
  hacc_flush29(acc, inf, (g << sh.low_bits) | cur, t, g, offsets, sh, run, maxp, partials, cnt);
}
// bucket b = the sum of its pieces, still on 29-bit limbs (14 products of 162 multiply-adds); 2^ll neighbouring lanes share the pieces. The sum leaves as a
// Point29Rec for the weighted bucket sum (htail29.cuh) — or, with buckets29 = null, converted to the lazy 8 x 32-bit form (one product per coordinate with
// 2^256 mod p). ZZ = 0 (mod p) in a sum — two pieces were +-each other somewhere, or an operand of the accumulation was +-its accumulator — raises the flag
// that sends the MSM to the general path.
static __global__ void __launch_bounds__(256) k_hacc_combine29(const Piece29 *__restrict__ partials, const uint32_t *__restrict__ offsets,
    const uint32_t *__restrict__ counts, HsortShape sh, uint32_t run, uint32_t maxp, uint32_t n_buckets,
                                                               uint32_t ll, XYZZ<Fq> *__restrict__ buckets, Point29Rec *__restrict__ buckets29,
                                                                   MsmCounters *cnt) {
  zk_take_prio(ll);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, b = t >> ll, sub = t & ((1u << ll) - 1), step = 1u << ll;
  const bool live = b < n_buckets;
  uint32_t np = 0;                                                                       // how many pieces the accumulation left for this bucket
  if (live) {
    const uint32_t c = counts[b], off = offsets[b] - (b >> sh.low_bits) * sh.region;
    if (c) np = min((off + c - 1) / run - off / run + 1, maxp);
  }
  const Piece29 *src = partials + (size_t)(live ? b : 0) * maxp;
  XYZZ29 acc;
  bool inf = true;
#pragma unroll
  for (int i = 0; i < 9; i++) { acc.X.l[i] = 0; acc.Y.l[i] = 0; acc.ZZ.l[i] = 0; acc.ZZZ.l[i] = 0; }
#pragma unroll 1
  for (uint32_t j = sub; j < np; j += step) {
    bool pinf;
    const XYZZ29 cur = piece29_load(src + j, pinf);
    if (pinf) continue;
    if (inf) { acc = cur; inf = false; } else acc = xyzz29_add(acc, cur);
  }
#pragma unroll 1
  for (uint32_t d = step >> 1; d >= 1; d >>= 1) {                                        // the lanes of a bucket meet by shuffles
    XYZZ29 o;
    const bool oinf = __shfl_down((int)inf, d, 64) != 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
      o.X.l[i] = __shfl_down(acc.X.l[i], d, 64); o.Y.l[i] = __shfl_down(acc.Y.l[i], d, 64);
      o.ZZ.l[i] = __shfl_down(acc.ZZ.l[i], d, 64); o.ZZZ.l[i] = __shfl_down(acc.ZZZ.l[i], d, 64);
    }
    if (sub + d < step && !oinf) { if (inf) { acc = o; inf = false; } else acc = xyzz29_add(acc, o); }
  }
  if (!live || sub != 0) return;
  // the weighted sum stays on 29-bit limbs (htail29.cuh): the sum as it is, four coordinate slots of twelve words; all-zero limbs = the point at infinity
  if (buckets29) {
    if (!inf && fq29_product_is_zero(acc.ZZ)) atomicOr(&cnt->pad[0], 1u);                // (ZZ is a product's result, or the lifted one: exact limbs)
    const uint32_t z = inf ? 0u : ~0u;
    uint4 *dst = reinterpret_cast<uint4 *>(buckets29 + b);
    const Fq29 *coord[4] = {&acc.X, &acc.Y, &acc.ZZ, &acc.ZZZ};
#pragma unroll
    for (int c4 = 0; c4 < 4; c4++) {
      const Fq29 &v = *coord[c4];
      dst[3 * c4] = make_uint4(v.l[0] & z, v.l[1] & z, v.l[2] & z, v.l[3] & z);
      dst[3 * c4 + 1] = make_uint4(v.l[4] & z, v.l[5] & z, v.l[6] & z, v.l[7] & z);
      dst[3 * c4 + 2] = make_uint4(v.l[8] & z, 0u, 0u, 0u);
    }
    return;
  }
  XYZZ<Fq> o = XYZZ<Fq>::inf();
  if (!inf) {
    acc.X.to_words(o.X.l); acc.Y.to_words(o.Y.l); acc.ZZ.to_words(o.ZZ.l); acc.ZZZ.to_words(o.ZZZ.l);
    if (o.ZZ.is_zero_lazy()) atomicOr(&cnt->pad[0], 1u);
    o = XYZZ<Fq>{o.X.normalize(), o.Y.normalize(), o.ZZ.normalize(), o.ZZZ.normalize()};
  }
  buckets[b] = o;
}
// table of the 29-bit kernel: coordinates x * 2^261 (mod p) from x * 2^256 — a Montgomery product with the plain integer 2^261 mod p; (0, 0) stays the point at
// infinity
static __global__ void k_table_to_r261(const Affine<Fq> *__restrict__ in, Affine<Fq> *__restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; Fq c;
#pragma unroll
  for (int j = 0; j < 8; j++) c.l[j] = FQ_TWO261[j];
  const Affine<Fq> p = in[i]; out[i] = {p.x * c, p.y * c};
}
// ---- sums by the 64 quads of a 256-thread workgroup ------------------------------------------------------------------------ quad q adds elements q, q+64,
// ...; then a tree over the 16 quads of each wave (shuffles) and over the 4 waves (LDS). The result is valid in lanes 0..3. tree over the 64 quads of a
// 256-thread workgroup: 16 quads per wave by shuffles, the 4 waves through LDS. Every quad brings `acc`; the sum is valid in lanes 0..3. `live` = how many of
// the quads (the first ones) hold something: levels whose partner quads are all empty are skipped (whole waves at a time), so a short list costs a short tree
template <class F> __device__ __forceinline__ XYZZ<F> block_quad_tree(XYZZ<F> acc, XYZZ<F> *lds, uint32_t live = 64) {
  const uint32_t q = threadIdx.x >> 2, wq = q & 15, wave = threadIdx.x >> 6; const int k = threadIdx.x & 3;
#pragma unroll 1
  for (int d = 8; d >= 1; d >>= 1) {
    if (wave * 16 + d >= live) continue;
    XYZZ<F> o = shfl_down_struct(acc, 4 * d);
    if (wq + d < 16) acc = quad_add(acc, o, k);
  }
  if ((threadIdx.x & 63) == 0) lds[wave] = acc;
  __syncthreads();
  if (wave == 0) { acc = q < 4 ? lds[q] : XYZZ<F>::inf();
#pragma unroll 1
    for (int d = 2; d >= 1; d >>= 1) {
      if ((uint32_t)d * 16 >= live) continue;
      XYZZ<F> o = shfl_down_struct(acc, 4 * d);
      if (q + d < 4) acc = quad_add(acc, o, k);
    }
  }
  return acc;
}
template <class F> __device__ __forceinline__ XYZZ<F> block_quad_sum(const XYZZ<F> *__restrict__ src, uint32_t len, XYZZ<F> *lds) {
  const uint32_t q = threadIdx.x >> 2; const int k = threadIdx.x & 3; XYZZ<F> acc = XYZZ<F>::inf();
  if (q < len) { XYZZ<F> nxt = src[q];
#pragma unroll 1
    for (uint32_t j = q; j < len; j += 64) { XYZZ<F> cur = nxt; if (j + 64 < len) nxt = src[j + 64]; acc = quad_add(acc, cur, k); } }
  return block_quad_tree(acc, lds, min(len, 64u));
}

constexpr uint32_t COMBINE_QUAD_MAX = 24;
template <class F>
__global__ void __launch_bounds__(256) k_msm_combine_tasks(const uint32_t *__restrict__ order, const uint32_t *__restrict__ task_off,
    const uint32_t *__restrict__ cls_start, uint32_t heavy_blocks,
                                                           const XYZZ<F> *__restrict__ partials, XYZZ<F> *__restrict__ buckets,
                                                               uint32_t *__restrict__ zero_words, uint32_t n_zero, int zero_empty) {
  __shared__ XYZZ<F> lds[4];
  // the histogram / slot counters are not needed any more: leave them cleared for the next run (saves a memset launch at the head of every MSM)
  {
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < n_zero) zero_words[gid] = 0;
  }
  // ranks below: count >= 63 (the only class that can hold more than COMBINE_QUAD_MAX tasks), count > 16
  const uint32_t n_big = cls_start[1], n_multi = cls_start[BSORT_CLASSES - 1 - MSM_TASK];
  if (blockIdx.x < heavy_blocks) {
    for (uint32_t r = blockIdx.x; r < n_big; r += heavy_blocks) { uint32_t beg = task_off[r], nt = task_off[r + 1] - beg; if (nt <= COMBINE_QUAD_MAX) continue;
      // (nt is the same for the whole workgroup)
      XYZZ<F> acc = block_quad_sum(partials + beg, nt, lds);
      if (threadIdx.x == 0) buckets[order[r]] = acc;
      __syncthreads();
    }
    return;
  }
  uint32_t r = (blockIdx.x - heavy_blocks) * 64 + (threadIdx.x >> 2); int k = threadIdx.x & 3; if (r >= n_multi) return;
  uint32_t beg = task_off[r], nt = task_off[r + 1] - beg; if (nt == 0 && zero_empty) { if (k == 0) buckets[order[r]] = XYZZ<F>::inf(); return; }
  if (nt < 2 || nt > COMBINE_QUAD_MAX) return; XYZZ<F> acc = partials[beg], nxt = partials[beg + 1];   // (one task: the accumulation wrote the bucket itself)
  // the next partial sum is in flight during the addition
#pragma unroll 1
  for (uint32_t j = 1; j < nt; j++) {
    XYZZ<F> cur = nxt;
    if (j + 1 < nt) nxt = partials[beg + j + 1];
    acc = quad_add(acc, cur, k);
  }
  if (k == 0) buckets[order[r]] = acc;
}

// ---- bucket reduction: sum_{b=1..NB} b * B_b per window, by segments of SEG buckets --------------------------------
// quad (w, s) handles buckets [s*SEG, (s+1)*SEG) of window w (bucket index j holds multiplier j+1):
//   run = sum B_j ; acc = sum (j - lo + 1) B_j  (running sums from the top) ; out = acc + lo * run
// Four lanes share each addition (curve.cuh, quad_add): the chain of 2*SEG additions and the ~log2(lo) doublings is what bounds this kernel.
template <class F>
__global__ void __launch_bounds__(64) k_msm_reduce_segments(const XYZZ<F> *__restrict__ buckets, uint32_t NB, uint32_t SEG, uint32_t n_seg_total,
    XYZZ<F> *__restrict__ seg_out) {
  uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; int k = threadIdx.x & 3; if (t >= n_seg_total) return;
  uint32_t segs_per_window = NB / SEG, w = t / segs_per_window, s = t % segs_per_window, lo = s * SEG;
  const XYZZ<F> *B = buckets + (size_t)w * NB + lo; XYZZ<F> run = XYZZ<F>::inf(), acc = XYZZ<F>::inf();
  XYZZ<F> nxt = B[SEG - 1];
  // next bucket in flight during the two additions
#pragma unroll 1
  for (int j = (int)SEG - 1; j >= 0; j--) {
    XYZZ<F> cur = nxt;
    if (j) nxt = B[j - 1];
    run = quad_add(run, cur, k);
    acc = quad_add(acc, run, k);
  }
  if (lo) acc = quad_add(acc, quad_mul_small(run, lo, k), k);
  if (k == 0) seg_out[t] = acc;
}

// ---- generic grouped sum: out[g] = sum_{j<len} in[g*len + j] (the last group may be short: n_in elements in total), one 256-thread workgroup per group.
// copy_src/copy_dst (optional): 16 bytes carried along by block 0 (the MSM counters travel to the host next to the result)
template <class F>
__global__ void __launch_bounds__(256) k_xyzz_group_sum(const XYZZ<F> *__restrict__ in, uint32_t len, uint32_t n_in, XYZZ<F> *__restrict__ out,
    uint4 *copy_src, uint4 *copy_dst) {
  __shared__ XYZZ<F> lds[4]; uint32_t g = blockIdx.x, beg = g * len, l = beg >= n_in ? 0 : min(len, n_in - beg);
  XYZZ<F> acc = block_quad_sum(in + beg, l, lds);
  if (threadIdx.x == 0) { out[g] = acc; if (copy_src && g == 0) *copy_dst = *copy_src; }
}

// ---- ones: strided partial sums over the compacted index list, one quad per partial -------------------------------------
template <class F>
__global__ void __launch_bounds__(256) k_msm_sum_ones(const Affine<F> *__restrict__ points, const uint32_t *__restrict__ ones, const MsmCounters *cnt,
    uint32_t n_quads, XYZZ<F> *__restrict__ partial) {
  uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; int k = threadIdx.x & 3; if (t >= n_quads) return;
  uint32_t n = cnt->n_ones; XYZZ<F> acc = XYZZ<F>::inf();
#pragma unroll 1
  for (uint32_t j = t; j < n; j += n_quads) acc = quad_madd(acc, points[ones[j]], k);
  if (k == 0) partial[t] = acc;
}

// ---- witness MSMs in four launches (sort, lanes, fold, tail) -----------------------------------------------------------------------------------------
// Measured with several proofs in flight (tools/inflight_probe.py): the four witness MSMs — 5 % of a proof's field products — took as much of the machine as
// the H query, because the general path above spends ten launches of tiny, dependent kernels on each of them (classify, plan, scatter, accumulate, combine,
// reduce, three tree levels, the ones sum). With fixed-base tables all windows share at most 128 buckets, and a workgroup is exactly the right size for one
// bucket:
//   sort k_wsort_tagged (the assignment arrived in compact form: a byte per variable says 0 / 1 / other) or k_wsort (plain vector): the ones go to a compacted
//             list —
//             four consecutive points at a time, as an index into the table of their subset sums —, every non-zero digit of the other scalars to its bucket's
//             REGION
//             (cap slots per bucket; a workgroup counts in LDS, reserves with one atomic per bucket, walks the digits again to place them). No histogram pass,
//             no
//             plan.  A bucket that would overflow raises the flag that sends the MSM back to the general path.
//   lanes     4,096 lanes share the buckets' entries in equal slices, 8,192 more stride over the list of ones: lane-serial mixed additions on nine 29-bit limbs
//             (k_wacc_lanes29 for G1 — complete: a repeated query point is doubled —, k_wacc_lanes_g2_29 over Fq2);
//   fold one workgroup per bucket adds its lanes' partial sums, 32 more fold the ones lanes to 32 partial sums: quad-cooperative trees (htail29.cuh: k_wfold29
//             on
//             29-bit limbs for G1; k_wacc_fold<Fq2> on 8 x 32-bit limbs for G2);
//   tail sum_b (b + 1) B_b as sum_k 2^k S_k, S_k = the sum of the buckets whose weight has bit k: one workgroup per weight bit, one for the ones' partial sums
//             (k_wtail29 / k_wtail<Fq2>); the host's Horner rule finishes with one-bit windows and adds the ones.
// The sort only depends on the scalars: MSMs over the same scalar vector (A and L*; B1 and B2) share one k_wsort (msm_impl.hpp: WsortBuffers).
constexpr uint32_t WFUSED_MAX_BUCKETS = 128;
// Key load: groups[g * 15 + v - 1] = the sum of P_(4g + j) over the bits j of v, v = 1 .. 15, affine (the all-zero record when the sum is the point at
// infinity; points beyond n count as infinity). 46 % of a BlockMaze assignment are ones — bits of SHA-256 states, as good as independent — so a group of four
// scalars holds two ones on average, which this table turns into one mixed addition (none for 1/16 of the groups): the ones are 80 % of a witness MSM's
// additions. 3.75x the memory of the points (54 MB for the A query, 65 MB for the G2 half of B).
template <class F>
__global__ void __launch_bounds__(64) k_ones_groups(const Affine<F> *__restrict__ points, uint32_t n, Affine<F> *__restrict__ groups) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; if (4 * g >= n) return;
  Affine<F> p0 = points[4 * g], p1 = 4 * g + 1 < n ? points[4 * g + 1] : Affine<F>::inf(), p2 = 4 * g + 2 < n ? points[4 * g + 2] : Affine<F>::inf(),
      p3 = 4 * g + 3 < n ? points[4 * g + 3] : Affine<F>::inf();
#pragma unroll 1
  for (uint32_t v = 1; v < 16; v++) { XYZZ<F> acc = XYZZ<F>::inf();
    if (v & 1) acc.madd_inl(p0); if (v & 2) acc.madd_inl(p1); if (v & 4) acc.madd_inl(p2); if (v & 8) acc.madd_inl(p3);
    // ZZ / ZZZ = 1 / Z
    Affine<F> o = Affine<F>::inf();
    if (!acc.is_inf()) {
      const F iz3 = acc.ZZZ.inv(), iz2 = (acc.ZZ * iz3).sqr();
      o = {acc.X * iz2, acc.Y * iz3};
    }
    groups[(size_t)g * 15 + v - 1] = o; }
}
template <int C>
__global__ void __launch_bounds__(256) k_wsort(const Fr *__restrict__ scalars, const uint32_t *__restrict__ scalar_index,
    const uint8_t *__restrict__ point_is_inf, uint32_t n, int c, int W, uint32_t point_stride, uint32_t NB, uint32_t cap,
                                               uint32_t *__restrict__ fill, uint32_t *__restrict__ fill_next, uint32_t *__restrict__ entries,
                                                   uint32_t *__restrict__ ones, MsmCounters *cnt, MsmCounters *cnt_next) {
  __shared__ uint32_t lcnt[WFUSED_MAX_BUCKETS], lbase[WFUSED_MAX_BUCKETS];
  // the two counter sets alternate: this run clears the next run's
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) *cnt_next = MsmCounters{0, 0, {0, 0}};
    if (threadIdx.x < NB) fill_next[threadIdx.x] = 0;
  }
  if (threadIdx.x < NB) lcnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63; bool live = i < n && !(point_is_inf && point_is_inf[i]); Fr k = Fr::zero();
  if (live) { k = scalars[scalar_index ? scalar_index[i] : i].from_mont(); live = !k.is_zero(); }
  bool is_one = false; if (live) { uint32_t o = k.l[0] ^ 1u; for (int j = 1; j < 8; j++) o |= k.l[j]; is_one = o == 0; }
  // the ones, four consecutive points at a time: lanes 4g .. 4g + 3 form the nibble v of their flags, and the quad's first lane appends ONE entry g * 15 + v -
  // 1 — the index of the precomputed sum of that subset in the groups table (k_ones_groups) — instead of up to four point indices: 15/16 of an addition per
  // group against 2 on average
  {
    const uint64_t m = __ballot(is_one);
    const uint32_t v = (uint32_t)(m >> (lane & ~3u)) & 15u;
    const bool lead = (lane & 3u) == 0 && v != 0;
    const uint64_t ml = __ballot(lead);
    __shared__ uint32_t wave_n[4], wg_at; const uint32_t wave = threadIdx.x >> 6;     // one atomic per workgroup on the list's counter (see k_wsort_tagged)
    if (lane == 0) wave_n[wave] = (uint32_t)__popcll(ml);
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t tot = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3]; wg_at = tot ? atomicAdd(&cnt->n_ones, tot) : 0u; }
    __syncthreads();
    if (lead) {
      uint32_t base = wg_at;
      for (uint32_t wv = 0; wv < wave; wv++) base += wave_n[wv];
      ones[base + __popcll(ml & ((1ull << lane) - 1))] = (i >> 2) * 15u + v - 1u;
    }
  }
  const bool other = live && !is_one;
  if (other) msm_walk_digits<C>(k.l, c, W, [&](int, int d) { if (d) atomicAdd(&lcnt[(uint32_t)(d < 0 ? -d : d) - 1], 1u); });
  __syncthreads();
  if (threadIdx.x < NB) {
    const uint32_t m = lcnt[threadIdx.x];
    uint32_t b = m ? atomicAdd(&fill[threadIdx.x], m) : 0;
    if (b + m > cap) atomicOr(&cnt->pad[0], 1u);
    lbase[threadIdx.x] = b;
    lcnt[threadIdx.x] = 0;
  }
  __syncthreads();
  if (other) msm_walk_digits<C>(k.l, c, W, [&](int w, int d) { if (!d) return; const uint32_t key = (uint32_t)(d < 0 ? -d : d) - 1,
      pos = lbase[key] + atomicAdd(&lcnt[key], 1u);
    if (pos < cap) entries[(size_t)key * cap + pos] = (i + (uint32_t)w * point_stride) | (d < 0 ? MSM_ENTRY_SIGN : 0u); });
}
// The same sort for an assignment that arrived in compact form (Prover::set_witness: 97 % of a BlockMaze assignment are the bits 0 and 1, so the device holds a
// byte per variable — 0 / 1 / 2 = anything else — and the list of the variables tagged 2, both written by k_expand_witness). k_wsort reads the 32-byte scalar
// of EVERY point and takes it out of Montgomery form (a field product per lane, 227 K + 136 K of them per proof) only to find a bit in 97 % of the cases; here
//   workgroups [0, ones_blocks) one lane per point: ONE byte decides "is one" (the nibble / subset-sum entry of k_wsort, unchanged); zeros and others do
//   nothing;
//   the workgroups after them one lane per listed variable (a few thousand): its position in this query (identity minus `base`, or the B query's inverse
//                                 index),
//                                 the scalar's digits, the same LDS counting and region reservation as k_wsort.
// WitnessTags travels by value; tags / z are the full vectors (variable 0 = ONE). A plain query (A, L): point i belongs to variable base + i. An indexed query
// (B): point i belongs to variable scalar_index[i], var_pos is the inverse of the query's whole index list (0xffffffff: no point for that variable) and `base`
// the first position of this slice (a shard) in it.
// (struct WitnessTags { tags, other_vars, n_other, var_pos, base }: gpu.hpp — the prover fills it)
template <int C>
__global__ void __launch_bounds__(256) k_wsort_tagged(const Fr *__restrict__ z, WitnessTags wt, const uint32_t *__restrict__ scalar_index,
    const uint8_t *__restrict__ point_is_inf, uint32_t n, int c, int W, uint32_t point_stride,
                                                      uint32_t NB, uint32_t cap, uint32_t ones_blocks, uint32_t *__restrict__ fill,
                                                          uint32_t *__restrict__ fill_next, uint32_t *__restrict__ entries, uint32_t *__restrict__ ones,
                                                      MsmCounters *cnt, MsmCounters *cnt_next) {
  __shared__ uint32_t lcnt[WFUSED_MAX_BUCKETS], lbase[WFUSED_MAX_BUCKETS];
  if (blockIdx.x == 0) {                                                                 // the two counter sets alternate: this run clears the next run's
    if (threadIdx.x == 0) *cnt_next = MsmCounters{0, 0, {0, 0}};
    if (threadIdx.x < NB) fill_next[threadIdx.x] = 0;
  }
  const uint32_t lane = threadIdx.x & 63;
  if (blockIdx.x < ones_blocks) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool is_one = false;
    if (i < n && !(point_is_inf && point_is_inf[i])) is_one = wt.tags[scalar_index ? scalar_index[i] : wt.base + i] == 1;
    // lanes 4g .. 4g + 3 form the nibble of their flags: ONE entry g * 15 + v - 1 per group (k_ones_groups)
    const uint64_t m = __ballot(is_one);
    const uint32_t v = (uint32_t)(m >> (lane & ~3u)) & 15u;
    const bool lead = (lane & 3u) == 0 && v != 0;
    const uint64_t ml = __ballot(lead);
    // ONE atomic per workgroup on the list's counter (3,500 waves adding to one address, one after the other, were what this kernel spent its 40 us on): the
    // waves' counts meet in LDS, the first lane reserves the workgroup's run
    __shared__ uint32_t wave_n[4], wg_at;
    const uint32_t wave = threadIdx.x >> 6;
    if (lane == 0) wave_n[wave] = (uint32_t)__popcll(ml);
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t tot = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3]; wg_at = tot ? atomicAdd(&cnt->n_ones, tot) : 0u; }
    __syncthreads();
    if (lead) {
      uint32_t at = wg_at;
      for (uint32_t wv = 0; wv < wave; wv++) at += wave_n[wv];
      ones[at + __popcll(ml & ((1ull << lane) - 1))] = (i >> 2) * 15u + v - 1u;
    }
    return;
  }
  // (a list made on the device: its length lies there too, the launch was sized by a bound — workgroups beyond the list leave at once)
  const uint32_t n_other = wt.n_other_dev ? min(*wt.n_other_dev, wt.n_other) : wt.n_other;
  if ((blockIdx.x - ones_blocks) * blockDim.x >= n_other) return;
  if (threadIdx.x < NB) lcnt[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t j = (blockIdx.x - ones_blocks) * blockDim.x + threadIdx.x;
  bool live = j < n_other;
  uint32_t pos = 0;
  Fr k = Fr::zero();
  if (live) {
    const uint32_t var = wt.other_vars[j];
    pos = (wt.var_pos ? wt.var_pos[var] : var) - wt.base;                                 // (a variable outside this slice wraps to a huge value)
    live = (wt.var_pos ? wt.var_pos[var] != 0xffffffffu : true) && pos < n && !(point_is_inf && point_is_inf[pos]);
    if (live) { k = z[var].from_mont(); live = !k.is_zero(); }
  }
  if (live) msm_walk_digits<C>(k.l, c, W, [&](int, int d) { if (d) atomicAdd(&lcnt[(uint32_t)(d < 0 ? -d : d) - 1], 1u); });
  __syncthreads();
  if (threadIdx.x < NB) {
    const uint32_t mcount = lcnt[threadIdx.x];
    const uint32_t b = mcount ? atomicAdd(&fill[threadIdx.x], mcount) : 0;
    if (b + mcount > cap) atomicOr(&cnt->pad[0], 1u);
    lbase[threadIdx.x] = b;
    lcnt[threadIdx.x] = 0;
  }
  __syncthreads();
  if (live) msm_walk_digits<C>(k.l, c, W, [&](int w, int d) {
    if (!d) return;
    const uint32_t key = (uint32_t)(d < 0 ? -d : d) - 1, slot = lbase[key] + atomicAdd(&lcnt[key], 1u);
    if (slot < cap) entries[(size_t)key * cap + slot] = (pos + (uint32_t)w * point_stride) | (d < 0 ? MSM_ENTRY_SIGN : 0u);
  });
}
// k_wacc_lanes + k_wacc_fold replace the first version of k_wacc (one workgroup of 64 quads per bucket, measured: 188 us, every SIMD of the chip busy with quad
// arithmetic and with tree levels in which most quads idle — 0.5 ms of the machine per proof for 5 % of its field products). Accumulation is lane-serial (10
// products per point, no exchange overhead, 160 waves in all); only the trees are cooperative, and they skip the levels a short list does not need.
// (lane counts swept again at the end of round 5, whole send proofs, device side: buckets 2048 / 4096 / 8192 / 12288 / 16384 with 8192 for the ones: 0.795 / 0.736 /
// 0.7325 / 0.737 / 0.759 ms; ones 4096 / 6144 / 8192 / 16384 with 8192 for the buckets: 0.778 / 0.743 / 0.7325 / 0.749 — profiles/r05_hacc_sweeps.txt)
constexpr uint32_t WFUSED_BUCKET_LANES = 8192, WFUSED_ONES_LANES = 8192, WFUSED_ONES_GROUPS = WFUSED_ONES_LANES / 256, WFUSED_MIN_SLICE = 8;
// Lanes are dealt to the buckets in proportion to their fill (a witness puts thousands of equal values into one bucket): slice length T = total / lanes, bucket
// b gets ceil(fill_b / T) lanes, lane_off[] (NB + 1 prefix sums, recomputed by every workgroup, written out by the first) tells the next kernel where each
// bucket's partial sums lie. The lane-serial accumulation of a G1 witness MSM on 29-bit limbs (round 4): k_wacc_lanes with the arithmetic of k_hacc_runs29 —
// the same madd-2008-s in two steps, 2,275 instructions instead of ~4,400 — gathering from the tables with coordinates x 2^261 (Bases::points261 / groups261)
// and leaving a Point29Rec per lane for k_wfold29 / k_wtail29 (htail29.cuh). COMPLETE in the lane (unlike the H accumulation): an operand equal to the
// accumulator is doubled, one equal to its negative leaves the point at infinity — repeated query points are a property of real keys, not an accident. The fold
// and the tail stay incomplete (two partial SUMS that coincide): ZZ = 0 (mod p) travels to a result slot, where k_wtail29 flags it and the MSM is repeated on
// the general path.
template <int UNIT>
__global__ void __launch_bounds__(256) k_wacc_lanes29(const Affine<Fq> *__restrict__ points261, const Affine<Fq> *__restrict__ groups261,
    const uint32_t *__restrict__ entries, const uint32_t *__restrict__ fill, uint32_t cap, uint32_t NB,
                                                      const uint32_t *__restrict__ ones, const MsmCounters *cnt, Point29Rec *__restrict__ partial,
                                                          uint32_t *__restrict__ lane_off) {
  zk_take_prio(NB);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ uint32_t m_of[WFUSED_MAX_BUCKETS], off[WFUSED_MAX_BUCKETS + 1], slice;
  const bool ones_lane = t >= WFUSED_BUCKET_LANES;                                       // (whole workgroups: the lane counts are multiples of 256)
  if (!ones_lane) {
    if (threadIdx.x < NB) m_of[threadIdx.x] = min(fill[threadIdx.x], cap);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      for (uint32_t b = 0; b < NB; b++) tot += m_of[b];
      const uint32_t T = max(WFUSED_MIN_SLICE, (tot + (WFUSED_BUCKET_LANES - NB) - 1) / (WFUSED_BUCKET_LANES - NB));
      uint32_t o = 0;
      for (uint32_t b = 0; b < NB; b++) { off[b] = o; o += (m_of[b] + T - 1) / T; }
      off[NB] = o; slice = T;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x <= NB) lane_off[threadIdx.x] = off[threadIdx.x];
  }
  // the lane's list: n_items table indices, `stride` apart
  const uint32_t *list = ones;
  const Affine<Fq> *table = groups261;
  uint32_t first = 0, n_items = 0, stride = 1;
  if (ones_lane) {
    const uint32_t u = t - WFUSED_BUCKET_LANES, n1 = cnt->n_ones;
    if (u >= WFUSED_ONES_LANES) return;
    first = u; stride = WFUSED_ONES_LANES; n_items = u < n1 ? (n1 - u + stride - 1) / stride : 0;
  } else {
    if (t >= off[NB]) return;
    uint32_t lo = 0, hi = NB;                                                             // the bucket b with off[b] <= t < off[b + 1]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= t) lo = mid; else hi = mid; }
    const uint32_t beg = (t - off[lo]) * slice, end = min(m_of[lo], beg + slice);
    list = entries + (size_t)lo * cap; table = points261; first = beg; n_items = end > beg ? end - beg : 0;
  }
  XYZZ29 acc; bool inf = true;
#pragma unroll
  for (int i = 0; i < 9; i++) { acc.X.l[i] = 0; acc.Y.l[i] = 0; acc.ZZ.l[i] = 0; acc.ZZZ.l[i] = 0; }
  if (n_items) {
    uint32_t v = list[first];
    Affine<Fq> p = table[v & ~MSM_ENTRY_SIGN];
#pragma unroll 1
    for (uint32_t i = 0; i < n_items; i++) {
      uint32_t vn = v; Affine<Fq> pn = p;
      // the next point's gather is in flight during this addition
      if (i + 1 < n_items) {
        vn = list[first + (size_t)(i + 1) * stride];
        pn = table[vn & ~MSM_ENTRY_SIGN];
      }
      if (!p.is_inf()) {
        const Fq29 px = Fq29::unpack(p.x.l), py = Fq29::cond_neg(Fq29::unpack(p.y.l), (v >> 31) != 0);
        if (inf) { acc.X = px; acc.Y = py.norm(); acc.ZZ = Fq29::one(); acc.ZZZ = Fq29::one(); inf = false; }
        else {
          Fq29 Pv, Rv; acc.madd_head(px, py, Pv, Rv);
          const Fq29 PP = Fq29::sqr(Pv);
          // the query points of a key repeat (two variables that enter the same constraint with the same coefficient and no other have the same A-query point),
          // and a lane meets such a pair back to back in most proofs: P = 0 means the operand is +-the accumulator — double it (R = 0 too) or leave the point
          // at infinity
          if (fq29_product_is_zero(PP)) { if (fq29_product_is_zero(Fq29::sqr(Rv))) acc = xyzz29_dbl_affine(px, py.norm()); else inf = true; }
          else acc.madd_tail_pp(Pv, Rv, PP);
        }
      }
      v = vn; p = pn;
    }
  }
  // the record of htail29.cuh: four coordinate slots of twelve words, all-zero = infinity
  const uint32_t z = inf ? 0u : ~0u;
  uint4 *dst = reinterpret_cast<uint4 *>(partial + t);
  const Fq29 *coord[4] = {&acc.X, &acc.Y, &acc.ZZ, &acc.ZZZ};
#pragma unroll
  for (int c4 = 0; c4 < 4; c4++) {
    const Fq29 &w = *coord[c4];
    dst[3 * c4] = make_uint4(w.l[0] & z, w.l[1] & z, w.l[2] & z, w.l[3] & z);
    dst[3 * c4 + 1] = make_uint4(w.l[4] & z, w.l[5] & z, w.l[6] & z, w.l[7] & z);
    dst[3 * c4 + 2] = make_uint4(w.l[8] & z, 0u, 0u, 0u);
  }
}
// ---- the G2 witness MSM (the G2 half of the B query, kc_multiexp.tcc:21-85) accumulated lane by lane on 29-bit limbs
// ----------------------------------------------------- Round 2 left this accumulation quad-cooperative (k_wacc_quads<Fq2>): a lane-serial mixed addition over
// Fq2 on 8 x 32-bit limbs keeps ~450 registers alive. On nine 29-bit limbs it fits (an accumulator is 72 registers) and needs a quarter of the instructions:
// four lanes no longer repeat each other's sums and selects, a product has no carry instruction, and a difference is nine additions and one carry step. Fq2 =
// Fq[u] / (u^2 + 1):
//   product Karatsuba, three Fq products: v0 = a0 b0, v1 = a1 b1, v2 = (a0 + a1)(b0 + b1); c0 = v0 - v1, c1 = v2 - v0 - v1 (the two sums go into their product
//             as
//             they are: limbs below 2^30 + 16 on both sides still fit the 64-bit column);
//   square    complex squaring, two Fq products:  c0 = (a0 + a1)(a0 - a1), c1 = 2 a0 a1;
//   X3, Y3    one Barrett step per component (Fq29::barrett: below 4.1 p) so that the next addition's differences take the constants K_6.
// Every constant of a difference and every value bound is checked by interval arithmetic in gen_field29.py (check_bounds_g2): X, Y < 4.1 p, ZZ / ZZZ < (3.2,
// 5.7) p per component. The formulas are incomplete like the H accumulation's: an operand equal to +-the accumulator leaves ZZ = 0 (mod p), which the lane
// finds when it stores its sum and reports like a sort overflow — the MSM is then repeated on the general path (complete formulas, msm_impl.hpp: finish_sync).
struct Fq2_29 { Fq29 c0, c1; };
__device__ __forceinline__ Fq2_29 fq2_29_mul(const Fq2_29 &a, const Fq2_29 &b) {
  const Fq29 v0 = Fq29::mul(a.c0, b.c0), v1 = Fq29::mul(a.c1, b.c1), v2 = Fq29::mul(Fq29::add_raw(a.c0, a.c1), Fq29::add_raw(b.c0, b.c1));
  return {Fq29::sub<2>(v0, v1), Fq29::sub<4>(v2, Fq29::add_raw(v0, v1))};
}
__device__ __forceinline__ Fq2_29 fq2_29_sqr(const Fq2_29 &a) {                            // components of a below 12 p
  const Fq29 m = Fq29::mul(a.c0, a.c1);
  Fq29 twice;
#pragma unroll
  for (int i = 0; i < 9; i++) twice.l[i] = 2u * m.l[i];
  return {Fq29::mul(Fq29::add_raw(a.c0, a.c1), Fq29::sub<12>(a.c0, a.c1)), twice.norm()};
}
template <int C0, int C1> __device__ __forceinline__ Fq2_29 fq2_29_sub(const Fq2_29 &a, const Fq2_29 &b) {
  return {Fq29::sub<C0>(a.c0, b.c0), Fq29::sub<C1>(a.c1, b.c1)};
}
struct XYZZ2_29 {
  Fq2_29 X, Y, ZZ, ZZZ;
  // madd-2008-s; px canonical, py canonical or K_2 - y, both with normalized limbs
  __device__ __forceinline__ void madd(const Fq2_29 &px, const Fq2_29 &py) {
    const Fq2_29 P = fq2_29_sub<6, 6>(fq2_29_mul(px, ZZ), X), R = fq2_29_sub<6, 6>(fq2_29_mul(py, ZZZ), Y);
    const Fq2_29 PP = fq2_29_sqr(P), PPP = fq2_29_mul(P, PP), Q = fq2_29_mul(X, PP), RR = fq2_29_sqr(R);
    Fq2_29 s;
#pragma unroll
    for (int i = 0; i < 9; i++) { s.c0.l[i] = PPP.c0.l[i] + 2u * Q.c0.l[i]; s.c1.l[i] = PPP.c1.l[i] + 2u * Q.c1.l[i]; }
    Fq2_29 X3 = fq2_29_sub<12, 18>(RR, s);
    X3 = {X3.c0.barrett(), X3.c1.barrett()};
    Fq2_29 Y3 = fq2_29_sub<4, 6>(fq2_29_mul(R, fq2_29_sub<6, 6>(Q, X3)), fq2_29_mul(Y, PPP));
    Y = {Y3.c0.barrett(), Y3.c1.barrett()};
    X = X3; ZZ = fq2_29_mul(ZZ, PP); ZZZ = fq2_29_mul(ZZZ, PPP);
  }
};
// a table point (coordinates x 2^261, canonical words) -> limbs; neg: the point's negative (y -> K_2 - y, normalized)
__device__ __forceinline__ void g2_29_unpack(const Affine<Fq2> &p, bool neg, Fq2_29 &px, Fq2_29 &py) {
  px = {Fq29::unpack(p.x.c0.l), Fq29::unpack(p.x.c1.l)};
  py = {Fq29::cond_neg(Fq29::unpack(p.y.c0.l), neg).norm(), Fq29::cond_neg(Fq29::unpack(p.y.c1.l), neg).norm()};
}
// Lane-serial accumulation of the G2 witness MSM (the layout of k_wacc_lanes: the first WFUSED_BUCKET_LANES lanes share the buckets' entries in slices
// proportional to the fill, the next WFUSED_ONES_LANES stride over the list of ones), from the tables with coordinates x 2^261 (points261 / groups261,
// k_table_to_r261_g2 at key load). A lane's sum leaves the 29-bit domain when it is stored: eight products with 2^256 mod p give the lazy 8 x 32-bit form that
// k_wacc_fold<Fq2> adds up.
// Round 5: the sum stays on 29-bit limbs — a Point29Rec2 (oct29.cuh: eight slots, component-major) for k_wfold_g2_29; no conversion, and the degenerate case
// (ZZ = 0 mod p) is left to the tail, where it arrives in at least one result: a product with ZZ = 0 stays 0.
template <int UNIT>   // (a template only so that the one translation unit that launches it instantiates it)
__global__ void __launch_bounds__(256) k_wacc_lanes_g2_29(const Affine<Fq2> *__restrict__ points261, const Affine<Fq2> *__restrict__ groups261,
    const uint32_t *__restrict__ entries, const uint32_t *__restrict__ fill, uint32_t cap,
                                                          uint32_t NB, const uint32_t *__restrict__ ones, MsmCounters *cnt, Point29Rec2 *__restrict__ partial,
                                                              uint32_t *__restrict__ lane_off) {
  zk_take_prio(NB);
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  __shared__ uint32_t m_of[WFUSED_MAX_BUCKETS], off[WFUSED_MAX_BUCKETS + 1], slice;
  const bool ones_lane = t >= WFUSED_BUCKET_LANES;                                       // (whole workgroups: the lane counts are multiples of 256)
  if (!ones_lane) {
    if (threadIdx.x < NB) m_of[threadIdx.x] = min(fill[threadIdx.x], cap);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      for (uint32_t b = 0; b < NB; b++) tot += m_of[b];
      const uint32_t T = max(WFUSED_MIN_SLICE, (tot + (WFUSED_BUCKET_LANES - NB) - 1) / (WFUSED_BUCKET_LANES - NB));
      uint32_t o = 0;
      for (uint32_t b = 0; b < NB; b++) { off[b] = o; o += (m_of[b] + T - 1) / T; }
      off[NB] = o; slice = T;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x <= NB) lane_off[threadIdx.x] = off[threadIdx.x];
  }
  // the lane's list: n_items table indices, `stride` apart in `list`, from `table`
  const uint32_t *list; const Affine<Fq2> *table; uint32_t first = 0, n_items = 0, stride = 1;
  if (ones_lane) {
    const uint32_t u = t - WFUSED_BUCKET_LANES, n1 = cnt->n_ones;
    if (u >= WFUSED_ONES_LANES) return;
    list = ones; table = groups261; first = u; stride = WFUSED_ONES_LANES; n_items = u < n1 ? (n1 - u + stride - 1) / stride : 0;
  } else {
    if (t >= off[NB]) return;
    uint32_t lo = 0, hi = NB;                                                             // the bucket b with off[b] <= t < off[b + 1]
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (off[mid] <= t) lo = mid; else hi = mid; }
    const uint32_t beg = (t - off[lo]) * slice, end = min(m_of[lo], beg + slice);
    list = entries + (size_t)lo * cap; table = points261; first = beg; n_items = end > beg ? end - beg : 0;
  }
  XYZZ2_29 acc; bool inf = true;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    acc.X.c0.l[i] = acc.X.c1.l[i] = acc.Y.c0.l[i] = acc.Y.c1.l[i] = acc.ZZ.c0.l[i] = acc.ZZ.c1.l[i] = acc.ZZZ.c0.l[i] = acc.ZZZ.c1.l[i] = 0;
  }
  if (n_items) {
    uint32_t v = list[first];
    Affine<Fq2> p = table[v & ~MSM_ENTRY_SIGN];
#pragma unroll 1
    for (uint32_t i = 0; i < n_items; i++) {
      uint32_t vn = v; Affine<Fq2> pn = p;
      // the next point's gather is in flight during this addition
      if (i + 1 < n_items) {
        vn = list[first + (size_t)(i + 1) * stride];
        pn = table[vn & ~MSM_ENTRY_SIGN];
      }
      if (!p.is_inf()) {
        Fq2_29 px, py; g2_29_unpack(p, (v >> 31) != 0, px, py);
        // lift: ZZ = ZZZ = 1 (their u-components stay zero)
        if (inf) {
          acc.X = px;
          acc.Y = py;
          acc.ZZ.c0 = Fq29::one();
          acc.ZZZ.c0 = Fq29::one();
          inf = false;
        }
        else acc.madd(px, py);
      }
      v = vn; p = pn;
    }
  }
  {
    uint4 *dst = reinterpret_cast<uint4 *>(partial + t);
    const uint32_t z = inf ? 0u : ~0u;
    const Fq29 *slot[8] = {&acc.X.c0, &acc.Y.c0, &acc.ZZ.c0, &acc.ZZZ.c0, &acc.X.c1, &acc.Y.c1, &acc.ZZ.c1, &acc.ZZZ.c1};
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const Fq29 &v = *slot[e];
      dst[3 * e] = make_uint4(v.l[0] & z, v.l[1] & z, v.l[2] & z, v.l[3] & z);
      dst[3 * e + 1] = make_uint4(v.l[4] & z, v.l[5] & z, v.l[6] & z, v.l[7] & z);
      dst[3 * e + 2] = make_uint4(v.l[8] & z, 0u, 0u, 0u);
    }
  }
}
// the G2 tables of the kernel above: coordinates x 2^261 (mod p) from x 2^256, component by component; (0, 0) stays the point at infinity
static __global__ void k_table_to_r261_g2(const Affine<Fq2> *__restrict__ in, Affine<Fq2> *__restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fq c;
#pragma unroll
  for (int j = 0; j < 8; j++) c.l[j] = FQ_TWO261[j];
  const Affine<Fq2> p = in[i];
  out[i] = {{p.x.c0 * c, p.x.c1 * c}, {p.y.c0 * c, p.y.c1 * c}};
}
constexpr int WTAIL_SLOTS = 8;   // the tails of the witness MSMs leave eight sums by weight bit (S_s, s = 0 .. 7) and the sum of the ones
// (the weighted bucket sum of the H query — sums by weight bit from two-level marginal sums, on 29-bit limbs — lives in htail29.cuh: k_hmarg29 / k_hbits29)

// ---- fixed-base precomputation: table[w*n + i] = 2^(c*w) * P_i, affine ------------------------------------------------
// The query points of a proving key never change, and 288 GB of HBM is plenty: with every window's multiple of every point stored, all windows of an MSM
// share ONE array of 2^(c-1) buckets — the bucket reduction shrinks by the number of windows and the host's c*W Horner doublings disappear.
// One lane per point: c doublings per window in XYZZ, the W-1 conversions to affine share one inversion (x = X/ZZ, y = Y/ZZZ, 1/ZZ = t*ZZZ, 1/ZZZ = t*ZZ with
// t = 1/(ZZ*ZZZ); Montgomery's trick over the windows).  tmp: (W-1)*n XYZZ, pref: (W-1)*n field elements, both window-major.
template <class F>
__global__ void __launch_bounds__(64) k_msm_precompute(Affine<F> *__restrict__ table, uint32_t n, int c, int W, XYZZ<F> *__restrict__ tmp,
    F *__restrict__ pref) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; const Affine<F> P = table[i];
  if (P.is_inf()) { for (int w = 1; w < W; w++) table[(size_t)w * n + i] = Affine<F>::inf(); return; }
  XYZZ<F> cur = XYZZ<F>::from_affine(P); F acc = F::one();
#pragma unroll 1
  for (int w = 1; w < W; w++) {
#pragma unroll 1
    for (int k = 0; k < c; k++) cur = cur.dbl_inl();
    tmp[(size_t)(w - 1) * n + i] = cur; pref[(size_t)(w - 1) * n + i] = acc; acc = acc * (cur.ZZ * cur.ZZZ);
  }
  F inv = acc.inv();
#pragma unroll 1
  for (int w = W - 1; w >= 1; w--) {
    const XYZZ<F> q = tmp[(size_t)(w - 1) * n + i];
    F t = inv * pref[(size_t)(w - 1) * n + i];
    inv = inv * (q.ZZ * q.ZZZ);
    table[(size_t)w * n + i] = {q.X * (t * q.ZZZ), q.Y * (t * q.ZZ)};
  }
}

}  // namespace zk
