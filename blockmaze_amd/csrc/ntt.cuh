// Radix-2 number-theoretic transform over Fr and the pointwise steps of the R1CS->QAP witness map (kernel K2,
// SURVEY.md §8a rows P2/P2').
//
// Replaces libfqfft's basic_radix2_domain::{FFT,iFFT,cosetFFT,icosetFFT,divide_by_Z_on_coset}
// (FQFFT/evaluation_domain/domains/basic_radix2_domain.tcc:48-112, basic_radix2_domain_aux.tcc:44-79,171-180) and, via
// the pre/post passes in step_domain kernels, step_radix2_domain (domains/step_radix2_domain.tcc:39-153,242-260).
// Field arithmetic is exact, so any butterfly schedule gives bit-identical vectors; the schedule here is the GPU one:
//   * n = n1 * n2: a column pass (k_ntt_cols) and a row pass (k_ntt_rows), each holding its tile in LDS — natural order in and out, no bit-reversal pass over HBM
//   * inside a tile: decimation-in-frequency stages three at a time with the eight values of a butterfly group in registers (one LDS round trip and one barrier
//     per three stages), stage twiddles staged in LDS, tile padded against bank conflicts
//   * `batch` independent vectors per launch (the witness map transforms A, B, C together)
//   * scaling by 1/n and the coset shift g^i are folded into the load of the column pass or the store of the row pass
//   * beyond 2^22 points: bit-reversal gather + one launch per stage (k_ntt_bitrev_scale, k_ntt_local, k_ntt_stage)
#pragma once
#include <hip/hip_runtime.h>
#include "field.cuh"

namespace zk {

constexpr int NTT_LOCAL_LOG = 10;            // 1024-point tiles: 32 KiB of LDS per workgroup
constexpr int NTT_LOCAL_THREADS = 512;

__device__ __forceinline__ uint32_t bitrev32(uint32_t x, int bits) { return __brev(x) >> (32 - bits); }

// out[i] = in[bitrev(i)] * (scale ? scale[bitrev(i)] : 1)   (out != in)
__global__ void k_ntt_bitrev_scale(const Fr *__restrict__ in, Fr *__restrict__ out, const Fr *__restrict__ scale, int logn, size_t stride_in, size_t stride_out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, n = 1u << logn; if (i >= n) return;
  const Fr *src = in + blockIdx.y * stride_in; Fr *dst = out + blockIdx.y * stride_out; uint32_t r = bitrev32(i, logn);
  Fr v = src[r]; if (scale) v = v * scale[r]; dst[i] = v;
}

// stages 1..min(logn, NTT_LOCAL_LOG) on contiguous 2^L tiles in LDS.  tw[j] = w^j, j < n/2, w the n-th root for this direction.
__global__ void __launch_bounds__(NTT_LOCAL_THREADS) k_ntt_local(Fr *__restrict__ data, const Fr *__restrict__ tw, int logn, int L, size_t stride) {
  extern __shared__ uint32_t lds_raw[]; Fr *tile = reinterpret_cast<Fr *>(lds_raw);
  const uint32_t T = 1u << L, half_n = 1u << (logn - 1); Fr *d = data + blockIdx.y * stride + (size_t)blockIdx.x * T;
  for (uint32_t i = threadIdx.x; i < T; i += blockDim.x) tile[i] = d[i];
  __syncthreads();
  for (int s = 1; s <= L; s++) {
    const uint32_t half = 1u << (s - 1);
    for (uint32_t b = threadIdx.x; b < T / 2; b += blockDim.x) {
      uint32_t j = b & (half - 1), k = (b >> (s - 1)) << s, i0 = k + j, i1 = i0 + half;
      Fr t = tile[i1]; if (j) t = t * tw[j * (half_n >> (s - 1))];
      Fr u = tile[i0]; tile[i0] = u + t; tile[i1] = u - t;
    }
    __syncthreads();
  }
  for (uint32_t i = threadIdx.x; i < T; i += blockDim.x) d[i] = tile[i];
}

// one global stage s (s > NTT_LOCAL_LOG): thread per butterfly
__global__ void k_ntt_stage(Fr *__restrict__ data, const Fr *__restrict__ tw, int logn, int s, size_t stride) {
  uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, half_n = 1u << (logn - 1); if (b >= half_n) return;
  Fr *d = data + blockIdx.y * stride; const uint32_t half = 1u << (s - 1); uint32_t j = b & (half - 1), i0 = ((b >> (s - 1)) << s) + j, i1 = i0 + half;
  Fr t = d[i1]; if (j) t = t * tw[j * (half_n >> (s - 1))];
  Fr u = d[i0]; d[i0] = u + t; d[i1] = u - t;
}

// ---- two-pass transform for n = n1 * n2 (both tiles fit LDS): natural order in, natural order out, no bit-reversal pass over HBM ----
//   X[k1 + n1*k2] = sum_{i2} w_n2^(i2 k2) * [ w_n^(i2 k1) * sum_{i1} x[i1*n2 + i2] * w_n1^(i1 k1) ]
// k_ntt_cols: the inner sums — for C adjacent columns i2 a workgroup loads the n1 x C tile (runs of C contiguous elements), runs log2(n1)
//   decimation-in-frequency stages in LDS, multiplies by the step twiddles and stores Y[k1*n2 + i2] to the same positions (safe in place).
// k_ntt_rows: the outer sums — C adjacent rows k1 (contiguous loads), log2(n2) stages in LDS, stores X[k1 + n1*k2] in runs of C.
// tw[j] = w_n^j for j < n/2.  Optional tables: `pre` multiplies the input (natural index), `post` the output (natural index).
constexpr int NTT_TILE_LOG = 11;             // at most 2048 elements per workgroup: 68 KiB of LDS for the padded tile + up to 32 KiB for its twiddles (the launch raises the dynamic LDS limit)
constexpr int NTT_TILE_THREADS = 256;

__device__ __forceinline__ Fr ntt_twiddle(const Fr *__restrict__ tw, uint32_t e, uint32_t half_n) { return e < half_n ? tw[e] : tw[e - half_n].neg(); }   // w^(n/2) = -1

// R decimation-in-frequency stages (s, s-1, ..., s-R+1) of the N-point transforms over the rows of tile[N][C] (element i of column c at tile[i*C + c]) with the
// 2^R values of a butterfly group held in registers: one LDS round trip and one barrier per R stages.  twl[j] = w_N^j (j < N/2) is an LDS copy of the twiddles.
// LDS layout of a tile: element e sits at e + (e >> 4) (one 32-byte pad per 16 elements).  Without it the last radix-8 pass (each lane owns 8 consecutive elements, so
// lanes are 16 elements = 512 bytes apart) and the bit-reversed read-out put all 64 lanes of an access on the same 8 banks.
__device__ __forceinline__ uint32_t ntt_pad(uint32_t e) { return e + (e >> 4); }
template <int R> __device__ __forceinline__ void ntt_lds_pass(Fr *tile, const Fr *twl, int logN, int logC, int s) {
  const uint32_t groups = (1u << (logN - R)) << logC, cmask = (1u << logC) - 1; const int sh = s - R;
  for (uint32_t w = threadIdx.x; w < groups; w += blockDim.x) {
    const uint32_t g = w >> logC, c = w & cmask, base = ((g >> sh) << s) | (g & ((1u << sh) - 1)); Fr x[1 << R];
#pragma unroll
    for (int q = 0; q < (1 << R); q++) x[q] = tile[ntt_pad(((base + ((uint32_t)q << sh)) << logC) + c)];
#pragma unroll
    for (int t = 0; t < R; t++) { const int st = s - t, hx = 1 << (R - 1 - t);
#pragma unroll
      for (int q = 0; q < (1 << R); q++) if (!(q & hx)) {
        uint32_t j = (base + ((uint32_t)q << sh)) & ((1u << (st - 1)) - 1); Fr u = x[q], v = x[q + hx]; x[q] = u + v; Fr d = u - v; if (j) d = d * twl[j << (logN - st)]; x[q + hx] = d; }
    }
#pragma unroll
    for (int q = 0; q < (1 << R); q++) tile[ntt_pad(((base + ((uint32_t)q << sh)) << logC) + c)] = x[q];
  }
  __syncthreads();
}
// log2(N) DIF stages; afterwards position p holds output bitrev(p).  tws = n / N: twl[j] = tw[j * tws]
__device__ __forceinline__ void ntt_lds_dif(Fr *tile, Fr *twl, int logN, int logC, const Fr *__restrict__ tw, uint32_t tws, int radix_log) {
  for (uint32_t j = threadIdx.x; j < (1u << logN) / 2; j += blockDim.x) twl[j] = tw[j * tws];
  __syncthreads();
  int s = logN;
  if (radix_log >= 3) for (; s >= 3; s -= 3) ntt_lds_pass<3>(tile, twl, logN, logC, s);
  if (radix_log >= 2) for (; s >= 2; s -= 2) ntt_lds_pass<2>(tile, twl, logN, logC, s);
  for (; s >= 1; s -= 1) ntt_lds_pass<1>(tile, twl, logN, logC, s);
}
__global__ void __launch_bounds__(NTT_TILE_THREADS) k_ntt_cols(const Fr *__restrict__ src, Fr *__restrict__ dst, const Fr *__restrict__ pre, const Fr *__restrict__ tw,
                                                               int logn, int log_n1, int logC, int radix_log, size_t stride_in, size_t stride_out) {
  extern __shared__ uint32_t lds_raw[]; Fr *tile = reinterpret_cast<Fr *>(lds_raw);
  // XCD-aware tile order.  A one-column tile reads 32 bytes of every 128-byte line it touches; the other three quarters belong to the next three columns.  Workgroup b
  // runs on XCD b % 8 (MI355X_MICROARCH.md, observed dispatch order), each XCD has its own L2, so with tile = blockIdx.x four different L2s fetched every line: 4.2x the
  // algorithmic traffic (PMC, profiles/r02e).  Giving XCD x the contiguous columns [x * n_tiles / 8, (x + 1) * n_tiles / 8) in dispatch order lets the four tiles of a
  // line share one L2 fetch.  (A different placement only costs the extra fetches again: results do not depend on it.)
  const uint32_t n_tiles = gridDim.x, tile_no = n_tiles >= 8 && n_tiles % 8 == 0 ? (blockIdx.x % 8) * (n_tiles / 8) + blockIdx.x / 8 : blockIdx.x;
  const int log_n2 = logn - log_n1; const uint32_t n2 = 1u << log_n2, C = 1u << logC, c0 = tile_no << logC, elems = (1u << log_n1) << logC, half_n = 1u << (logn - 1);
  const Fr *s = src + blockIdx.y * stride_in; Fr *d = dst + blockIdx.y * stride_out;
  for (uint32_t w = threadIdx.x; w < elems; w += blockDim.x) { uint32_t g = ((w >> logC) << log_n2) + c0 + (w & (C - 1)); Fr v = s[g]; if (pre) v = v * pre[g]; tile[ntt_pad(w)] = v; }
  __syncthreads();
  ntt_lds_dif(tile, tile + ntt_pad(elems), log_n1, logC, tw, n2, radix_log);
  for (uint32_t w = threadIdx.x; w < elems; w += blockDim.x) {
    uint32_t k1 = w >> logC, c = w & (C - 1), i2 = c0 + c, p = log_n1 ? bitrev32(k1, log_n1) : 0; Fr v = tile[ntt_pad((p << logC) + c)];
    uint32_t e = i2 * k1; if (e) v = v * ntt_twiddle(tw, e, half_n);
    d[(k1 << log_n2) + i2] = v;
  }
}
__global__ void __launch_bounds__(NTT_TILE_THREADS) k_ntt_rows(const Fr *__restrict__ src, Fr *__restrict__ dst, const Fr *__restrict__ post, const Fr *__restrict__ tw,
                                                               int logn, int log_n1, int logC, int radix_log, size_t stride_in, size_t stride_out) {
  extern __shared__ uint32_t lds_raw[]; Fr *tile = reinterpret_cast<Fr *>(lds_raw);
  const int log_n2 = logn - log_n1; const uint32_t n2 = 1u << log_n2, C = 1u << logC, r0 = blockIdx.x << logC, elems = n2 << logC;
  const Fr *s = src + blockIdx.y * stride_in; Fr *d = dst + blockIdx.y * stride_out;
  for (uint32_t w = threadIdx.x; w < elems; w += blockDim.x) { uint32_t c = w >> log_n2, i2 = w & (n2 - 1); tile[ntt_pad((i2 << logC) + c)] = s[((size_t)(r0 + c) << log_n2) + i2]; }
  __syncthreads();
  ntt_lds_dif(tile, tile + ntt_pad(elems), log_n2, logC, tw, 1u << log_n1, radix_log);
  for (uint32_t w = threadIdx.x; w < elems; w += blockDim.x) {
    uint32_t k2 = w >> logC, c = w & (C - 1), p = bitrev32(k2, log_n2), o = (k2 << log_n1) + r0 + c; Fr v = tile[ntt_pad((p << logC) + c)];
    if (post) v = v * post[o];
    d[o] = v;
  }
}

// a[i] *= table[i]
__global__ void k_fr_mul_table(Fr *__restrict__ a, const Fr *__restrict__ table, uint32_t n, size_t stride) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; Fr *d = a + blockIdx.y * stride; d[i] = d[i] * table[i];
}
// h[i] = (a[i]*b[i] - c[i]) * zinv[i or 0]   (r1cs_to_qap.tcc:281-310: H_tmp = A*B - C, then divide_by_Z_on_coset)
__global__ void k_qap_pointwise(Fr *__restrict__ a, const Fr *__restrict__ b, const Fr *__restrict__ c, const Fr *__restrict__ zinv, int zinv_is_table, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; Fr t = a[i] * b[i]; if (c) t = t - c[i]; a[i] = t * zinv[zinv_is_table ? i : 0];   // c == nullptr: the C polynomial is folded into the L query (ecntt.cuh)
}
__global__ void k_fr_to_mont(Fr *__restrict__ a, uint32_t n) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i].to_mont(); }
__global__ void k_fr_from_mont(Fr *__restrict__ a, uint32_t n) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i].from_mont(); }

constexpr uint32_t R1CS_LONG_ROW = 16;   // rows with more terms than this get a whole wave (k_r1cs_long_rows3)
// ---- R1CS rows times assignment (kernel K1; r1cs_to_qap.tcc:224-236,281-285; linear_combination::evaluate) ---------
// CSR with coefficient *indices* into a small table (the circuits use a few hundred distinct coefficients: +-1, +-2^k; table slots 0 / 1 are +1 / -1 and skip the
// multiply).  Rows with more than R1CS_LONG_ROW terms in any matrix (bit-packing constraints: 32 ... 253 terms) get one wave each (k_r1cs_long_rows3), the rest one lane.
// All three matrices in one launch, the evaluation vectors completed (input-consistency rows
// r1cs_to_qap.tcc:227-230, zero padding up to the domain size) and the satisfiability test a*b == c (protoboard::is_satisfied, sendcgo.cpp:209) done on the
// values while they are in registers.  A violated row stores `seq` (the number of this evaluation) to *fail, a word in mapped host memory: no reset, no copy.
struct R1csMatrices { const uint32_t *rowptr[3], *col[3], *cid[3]; };
__device__ __forceinline__ Fr r1cs_row_dot(const R1csMatrices &M, int mm, uint32_t r, const Fr *__restrict__ ctab, const Fr *__restrict__ z) {
  Fr acc = Fr::zero(); const uint32_t *col = M.col[mm], *cid = M.cid[mm];
  for (uint32_t k = M.rowptr[mm][r], e = M.rowptr[mm][r + 1]; k < e; k++) { uint32_t ci = cid[k]; Fr v = z[col[k]]; if (ci == 0) acc = acc + v; else if (ci == 1) acc = acc - v; else acc = acc + ctab[ci] * v; }
  return acc;
}
__global__ void __launch_bounds__(256) k_r1cs_rows3(R1csMatrices M, const Fr *__restrict__ ctab, const Fr *__restrict__ z, uint32_t n_rows, uint32_t n_inputs, uint32_t m, Fr *__restrict__ abc, uint32_t seq, uint32_t *fail) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; if (r >= m) return;
  if (r >= n_rows) { abc[r] = r <= n_rows + n_inputs ? z[r - n_rows] : Fr::zero(); abc[m + r] = Fr::zero(); abc[2 * (size_t)m + r] = Fr::zero(); return; }
  if (M.rowptr[0][r + 1] - M.rowptr[0][r] > R1CS_LONG_ROW || M.rowptr[1][r + 1] - M.rowptr[1][r] > R1CS_LONG_ROW || M.rowptr[2][r + 1] - M.rowptr[2][r] > R1CS_LONG_ROW) return;   // k_r1cs_long_rows3
  Fr a = r1cs_row_dot(M, 0, r, ctab, z), b = r1cs_row_dot(M, 1, r, ctab, z), c = r1cs_row_dot(M, 2, r, ctab, z);
  abc[r] = a; abc[m + r] = b; abc[2 * (size_t)m + r] = c;
  if (a * b != c) *fail = seq;
}
__global__ void __launch_bounds__(64) k_r1cs_long_rows3(const uint32_t *__restrict__ rows, R1csMatrices M, const Fr *__restrict__ ctab, const Fr *__restrict__ z, uint32_t m, Fr *__restrict__ abc, uint32_t seq, uint32_t *fail) {
  uint32_t r = rows[blockIdx.x], lane = threadIdx.x; Fr v[3];
#pragma unroll
  for (int mm = 0; mm < 3; mm++) { Fr acc = Fr::zero(); const uint32_t *col = M.col[mm], *cid = M.cid[mm];
    for (uint32_t k = M.rowptr[mm][r] + lane, e = M.rowptr[mm][r + 1]; k < e; k += 64) { uint32_t ci = cid[k]; Fr x = z[col[k]]; if (ci == 0) acc = acc + x; else if (ci == 1) acc = acc - x; else acc = acc + ctab[ci] * x; }
    const uint32_t len = M.rowptr[mm][r + 1] - M.rowptr[mm][r];      // (wave-uniform) the long matrix of a packing constraint has 32..35 terms, the other two have one: no tree for those
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) { if ((uint32_t)d >= len) continue; Fr o; for (int i = 0; i < 8; i++) o.l[i] = __shfl_down(acc.l[i], d, 64); acc = acc + o; }
    v[mm] = acc; }
  if (lane == 0) { abc[r] = v[0]; abc[m + r] = v[1]; abc[2 * (size_t)m + r] = v[2]; if (v[0] * v[1] != v[2]) *fail = seq; }
}
// Assignment upload in compact form: 97 % of a BlockMaze witness are the bits 0 and 1, so the host sends two bitmaps (value is `one` / value is something else), the
// running count of "something else" per 64 entries and only those values (0.3 MB instead of 7.3 MB over PCIe); this kernel rebuilds the vector.
__global__ void k_expand_witness(const uint64_t *__restrict__ ones_bm, const uint64_t *__restrict__ other_bm, const uint32_t *__restrict__ block_off, const Fr *__restrict__ values, Fr one_value, uint32_t n, Fr *__restrict__ out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; const uint32_t wd = i >> 6, bit = i & 63; const uint64_t ob = other_bm[wd];
  if ((ob >> bit) & 1) out[i] = values[block_off[wd] + (uint32_t)__popcll(ob & ((1ull << bit) - 1))];
  else out[i] = ((ones_bm[wd] >> bit) & 1) ? one_value : Fr::zero();
}
// both of the above in ONE launch (the two are independent and each too small to fill the chip for long: 27 + 25 us one after the other at the head of every proof's
// critical chain): the first `short_blocks` workgroups take the one-lane rows, the others four long rows each, one per wave
__global__ void __launch_bounds__(256) k_r1cs_rows_all(R1csMatrices M, const Fr *__restrict__ ctab, const Fr *__restrict__ z, uint32_t n_rows, uint32_t n_inputs, uint32_t m, const uint32_t *__restrict__ long_rows, uint32_t n_long, uint32_t short_blocks,
                                                       Fr *__restrict__ abc, uint32_t seq, uint32_t *fail) {
  if (blockIdx.x < short_blocks) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; if (r >= m) return;
    if (r >= n_rows) { abc[r] = r <= n_rows + n_inputs ? z[r - n_rows] : Fr::zero(); abc[m + r] = Fr::zero(); abc[2 * (size_t)m + r] = Fr::zero(); return; }
    if (M.rowptr[0][r + 1] - M.rowptr[0][r] > R1CS_LONG_ROW || M.rowptr[1][r + 1] - M.rowptr[1][r] > R1CS_LONG_ROW || M.rowptr[2][r + 1] - M.rowptr[2][r] > R1CS_LONG_ROW) return;
    Fr a = r1cs_row_dot(M, 0, r, ctab, z), b = r1cs_row_dot(M, 1, r, ctab, z), c = r1cs_row_dot(M, 2, r, ctab, z);
    abc[r] = a; abc[m + r] = b; abc[2 * (size_t)m + r] = c;
    if (a * b != c) *fail = seq;
    return; }
  const uint32_t w = (blockIdx.x - short_blocks) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63; if (w >= n_long) return;
  const uint32_t r = long_rows[w]; Fr v[3];
#pragma unroll
  for (int mm = 0; mm < 3; mm++) { Fr acc = Fr::zero(); const uint32_t *col = M.col[mm], *cid = M.cid[mm];
    for (uint32_t k = M.rowptr[mm][r] + lane, e = M.rowptr[mm][r + 1]; k < e; k += 64) { uint32_t ci = cid[k]; Fr x = z[col[k]]; if (ci == 0) acc = acc + x; else if (ci == 1) acc = acc - x; else acc = acc + ctab[ci] * x; }
    const uint32_t len = M.rowptr[mm][r + 1] - M.rowptr[mm][r];
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) { if ((uint32_t)d >= len) continue; Fr o; for (int i = 0; i < 8; i++) o.l[i] = __shfl_down(acc.l[i], d, 64); acc = acc + o; }
    v[mm] = acc; }
  if (lane == 0) { abc[r] = v[0]; abc[m + r] = v[1]; abc[2 * (size_t)m + r] = v[2]; if (v[0] * v[1] != v[2]) *fail = seq; }
}
// satisfiability: flag[0] |= (a[i]*b[i] != c[i]) over the constraint rows (protoboard::is_satisfied, sendcgo.cpp:209)
__global__ void k_r1cs_check(const Fr *__restrict__ a, const Fr *__restrict__ b, const Fr *__restrict__ c, uint32_t n_rows, uint32_t *flag) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; if (r >= n_rows) return; if (a[r] * b[r] != c[r]) atomicOr(flag, 1u);
}

}  // namespace zk
