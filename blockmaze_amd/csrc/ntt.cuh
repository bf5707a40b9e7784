// Radix-2 number-theoretic transform over Fr and the pointwise steps of the R1CS->QAP witness map (kernel K2,
// SURVEY.md §8a rows P2/P2').
//
// Replaces libfqfft's basic_radix2_domain::{FFT,iFFT,cosetFFT,icosetFFT,divide_by_Z_on_coset}
// (FQFFT/evaluation_domain/domains/basic_radix2_domain.tcc:48-112, basic_radix2_domain_aux.tcc:44-79,171-180) and, via
// the pre/post passes in step_domain kernels, step_radix2_domain (domains/step_radix2_domain.tcc:39-153,242-260).
// Field arithmetic is exact, so any butterfly schedule gives bit-identical vectors; the schedule here is the GPU one:
//   * n = n1 * n2: a column pass (k_ntt_cols) and a row pass (k_ntt_rows), each holding its tile in LDS — natural order in and out, no bit-reversal pass over
//   HBM
//   * inside a tile: elements on nine 29-bit limbs, Cooley-Tukey stages two (or three) at a time with the values of a butterfly group in registers (one LDS
//     round
//     trip and one barrier per pass), stage twiddles staged in LDS, tile padded against bank conflicts
//   * `batch` independent vectors per launch (the witness map transforms A, B, C together)
//   * scaling by 1/n and the coset shift g^i are folded into the load of the column pass or the store of the row pass
//   * beyond 2^22 points: bit-reversal gather + one launch per stage (k_ntt_bitrev_scale, k_ntt_local, k_ntt_stage)
#pragma once
#include <hip/hip_runtime.h>
#include "field.cuh"
#include "field29.cuh"

namespace zk {

constexpr int NTT_LOCAL_LOG = 10;            // 1024-point tiles: 32 KiB of LDS per workgroup
constexpr int NTT_LOCAL_THREADS = 512;

__device__ __forceinline__ uint32_t bitrev32(uint32_t x, int bits) { return __brev(x) >> (32 - bits); }

// out[i] = in[bitrev(i)] * (scale ? scale[bitrev(i)] : 1)   (out != in)
__global__ void k_ntt_bitrev_scale(const Fr *__restrict__ in, Fr *__restrict__ out, const Fr *__restrict__ scale, int logn, size_t stride_in,
    size_t stride_out) {
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
// at most 2048 elements per workgroup: 77 KiB of LDS for the padded tile + up to 36 KiB for its twiddles (the launch raises the dynamic LDS limit)
constexpr int NTT_TILE_LOG = 11;
constexpr int NTT_TILE_THREADS = 256;

// Round 3: INSIDE a tile the elements live on nine 29-bit limbs (Fr29, field29_gfx950.inc: Montgomery radix 2^261): a product is 162 multiply-adds and no carry
// instruction (the 8 x 32-bit product: 128 + 128 and a conditional subtraction), sums and differences are nine independent 32-bit operations and a parallel
// carry step. HBM keeps the 8 x 32-bit Montgomery form (radix 2^256), so nothing outside the two tile kernels changes. Conversions cost nothing: the eight
// words of x 2^256 ARE the integer of (x / 32) 2^261 — the transform is linear, the factor 1/32 rides through it —, and the multiplication every element gets
// on its way out (step twiddle in the column pass, the constant one in the row pass) uses the 2^261-form of its factor (tw261[j] = w^j 2^261 mod r, canonical,
// 8 words), which cancels the 2^261 and leaves y 2^256 below 2 r: packed to words and reduced once, the canonical form the next kernel expects.
// The butterflies are Cooley-Tukey's (u + c v, u - c v) in natural-in / bit-reversed-out order (the polynomial view: f mod (x^M - e) splits into f0 + c f1 and
// f0 - c f1 with c^2 = e, so all butterflies of a block share the twiddle c = w^(bitrev(block) * span)): with the product BEFORE the sum a value grows by at
// most 2.2 r per stage (u + t below u + 1.2 r, u + 2r - t below u + 2 r; t = c v below v / 169 + r) — 25 r after eleven stages, no reduction inside a tile;
// Gentleman-Sande's (u + v, (u - v) c) doubles the sum path every stage. Same positions in and out as the decimation-in-frequency passes this replaces:
// position p ends up holding output bitrev(p). LDS layout of a tile: element e sits at e + (e >> 4) (one element of padding per 16) as before.
__device__ __forceinline__ uint32_t ntt_pad(uint32_t e) { return e + (e >> 4); }
__device__ __forceinline__ Fr29 ntt29_from_words(const Fr &v) { uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = v.l[i];
  return Fr29::unpack(w); }
// value (any multiple of r added, exact limbs, below 2 r) -> canonical 8 x 32-bit words
__device__ __forceinline__ Fr ntt29_to_words(const Fr29 &t) { uint32_t w[8]; t.pack_words(w); Fr r;
#pragma unroll
  for (int i = 0; i < 8; i++) r.l[i] = w[i];
  return Fr::reduce_once(r); }
// R stages (s, s-1, ..., s-R+1; stage st pairs positions 2^(st-1) apart) of the N-point transforms over the rows of tile[N][C] (element i of column c at
// tile[i*C + c]) with the 2^R values of a butterfly group held in registers: one LDS round trip and one barrier per R stages. twl[j] = w_N^j 2^261 (j < N/2) is
// an LDS copy of the twiddles.
template <int R> __device__ __forceinline__ void ntt29_lds_pass(Fr29 *tile, const Fr29 *twl, int logN, int logC, int s) {
  const uint32_t groups = (1u << (logN - R)) << logC, cmask = (1u << logC) - 1; const int sh = s - R;
  for (uint32_t w = threadIdx.x; w < groups; w += blockDim.x) {
    const uint32_t g = w >> logC, c = w & cmask, base = ((g >> sh) << s) | (g & ((1u << sh) - 1)); Fr29 x[1 << R];
#pragma unroll
    for (int q = 0; q < (1 << R); q++) x[q] = tile[ntt_pad(((base + ((uint32_t)q << sh)) << logC) + c)];
    // bits: the width of the block index at this stage (0: the first stage, no twiddle)
#pragma unroll
    for (int t = 0; t < R; t++) {
      const int st = s - t, hx = 1 << (R - 1 - t), bits = logN - st;
#pragma unroll
      for (int q = 0; q < (1 << R); q++) if (!(q & hx)) {
        const uint32_t blk = (base + ((uint32_t)q << sh)) >> st; Fr29 u = x[q], v = x[q + hx];
        if (bits) v = Fr29::mul(v, twl[(__brev(blk) >> (32 - bits)) << (st - 1)]);   // (bits is the same for the whole workgroup)
        // limbs are brought back to 29 bits after every SECOND stage (and after the last one of the pass): a product takes one operand with limbs up to 2^31.4,
        // and two stages add at most 2 (2^30 + 64) to a normalized limb (gen_field29.py: check_bounds_ntt_stages) — a third would not fit
        const Fr29 sum = Fr29::add_raw(u, v), dif = Fr29::sub_product(u, v);
        const bool carry_now = (t & 1) == 1 || t == R - 1;
        x[q] = carry_now ? sum.norm() : sum;
        x[q + hx] = carry_now ? dif.norm() : dif;
      }
    }
#pragma unroll
    for (int q = 0; q < (1 << R); q++) tile[ntt_pad(((base + ((uint32_t)q << sh)) << logC) + c)] = x[q];
  }
  __syncthreads();
}
// log2(N) stages; afterwards position p holds output bitrev(p).  tws = n / N: twl[j] = tw261[j * tws]
__device__ __forceinline__ void ntt29_lds_transform(Fr29 *tile, Fr29 *twl, int logN, int logC, const Fr *__restrict__ tw261, uint32_t tws, int radix_log) {
  for (uint32_t j = threadIdx.x; j < (1u << logN) / 2; j += blockDim.x) twl[j] = ntt29_from_words(tw261[j * tws]);
  __syncthreads();
  int s = logN;
  if (radix_log >= 3) for (; s >= 3; s -= 3) ntt29_lds_pass<3>(tile, twl, logN, logC, s);
  if (radix_log >= 2) for (; s >= 2; s -= 2) ntt29_lds_pass<2>(tile, twl, logN, logC, s);
  for (; s >= 1; s -= 1) ntt29_lds_pass<1>(tile, twl, logN, logC, s);
}
// tw261[j] = w_n^j 2^261 mod r for j < n/2, canonical
// One transform's share of a tile launch. factor261: the column pass's optional factor per input element (f 2^261); post: the row pass's optional factor per
// output element (f 2^256, multiplied on the 8 x 32-bit side); scale261: one factor for the whole vector, applied by the pass that stores the final values (the
// row pass, or the column pass when it is the whole transform); tw261[j] = w_n^j 2^261 mod r for j < n/2 — all canonical. A launch carries up to TWO jobs
// (blockIdx.x < a.tiles: job a, else job b): the B-point and the S-point transform of a step-radix-2 domain (mint, redeem, deposit-32) run side by side in one
// launch instead of one after the other — each alone fills half the chip or less and is as long as its dependent passes. out261 (row pass only): a factor per
// OUTPUT element in place of the constant scale261 — the inverse transform of a radix-2 domain hands the coset factor g^i of the forward transform that follows
// to its own last product (1/m g^i 2^261), which costs that product a 32-byte load and saves the next column pass a product per element.
struct NttJob {
  const Fr *src;
  Fr *dst;
  const Fr *factor;
  const Fr *tw261;
  Fr scale261;
  int logn, log_n1, logC;
  uint32_t tiles;
  size_t stride_in, stride_out;
  const Fr *out261;
};
__global__ void __launch_bounds__(NTT_TILE_THREADS) k_ntt_cols(NttJob ja, NttJob jb, int radix_log) {
  // (bits 8.. of radix_log: the kernel's wave priority, gpu_internal.hpp: zk_prio_bits)
  { const int prio = radix_log >> 8; radix_log &= 0xff; if (prio == 1) __builtin_amdgcn_s_setprio(1); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else if (prio == 3) __builtin_amdgcn_s_setprio(3); }
  extern __shared__ uint32_t lds_raw[]; Fr29 *tile = reinterpret_cast<Fr29 *>(lds_raw);
  const bool second = blockIdx.x >= ja.tiles; const NttJob j = second ? jb : ja;   // by value: uniform selects, a reference would put both jobs on the stack
  const uint32_t bid = blockIdx.x - (second ? ja.tiles : 0u);
  const int logn = j.logn, log_n1 = j.log_n1, logC = j.logC; const Fr *__restrict__ tw261 = j.tw261; const Fr *__restrict__ pre261 = j.factor;
  // XCD-aware tile order. A narrow tile reads 32 or 64 bytes of every 128-byte line it touches; the rest belongs to the next columns. Workgroup b runs on XCD b
  // % 8 (MI355X_MICROARCH.md, observed dispatch order), each XCD has its own L2, so with tile = blockIdx.x four different L2s fetched every line: 4.2x the
  // algorithmic traffic (PMC, profiles/r02e). Giving XCD x the contiguous columns [x * n_tiles / 8, (x + 1) * n_tiles / 8) in dispatch order lets the tiles of
  // a line share one L2 fetch. (A different placement only costs the extra fetches again: results do not depend on it.)
  const uint32_t n_tiles = j.tiles, tile_no = n_tiles >= 8 && n_tiles % 8 == 0 ? (bid % 8) * (n_tiles / 8) + bid / 8 : bid;
  const int log_n2 = logn - log_n1;
  const uint32_t n2 = 1u << log_n2, C = 1u << logC, c0 = tile_no << logC, elems = (1u << log_n1) << logC, half_n = logn ? 1u << (logn - 1) : 1u;
  const Fr *s = j.src + blockIdx.y * j.stride_in; Fr *d = j.dst + blockIdx.y * j.stride_out;
  // (x 2^256)(f 2^261) / 2^261 = x f 2^256, exact limbs, below 2 r
  for (uint32_t w = threadIdx.x; w < elems; w += blockDim.x) {
    uint32_t g = ((w >> logC) << log_n2) + c0 + (w & (C - 1));
    Fr29 v = ntt29_from_words(s[g]);
    if (pre261) v = Fr29::mul(ntt29_from_words(pre261[g]), v);
    tile[ntt_pad(w)] = v;
  }
  __syncthreads();
  ntt29_lds_transform(tile, tile + ntt_pad(elems), log_n1, logC, tw261, n2, radix_log);
  for (uint32_t w = threadIdx.x; w < elems; w += blockDim.x) {
    uint32_t k1 = w >> logC, c = w & (C - 1), i2 = c0 + c, p = log_n1 ? bitrev32(k1, log_n1) : 0; const Fr29 v = tile[ntt_pad((p << logC) + c)];
    // w^(n/2) = -1; the factor also brings the element back below 2 r
    const uint32_t e = i2 * k1;
    Fr29 t = ntt29_from_words(log_n2 ? tw261[e < half_n ? e : e - half_n] : j.scale261);
    if (e >= half_n) t = Fr29::neg_product(t);
    d[(k1 << log_n2) + i2] = ntt29_to_words(Fr29::mul(t, v));
  }
}
__global__ void __launch_bounds__(NTT_TILE_THREADS) k_ntt_rows(NttJob ja, NttJob jb, int radix_log) {
  // (bits 8.. of radix_log: the kernel's wave priority, gpu_internal.hpp: zk_prio_bits)
  { const int prio = radix_log >> 8; radix_log &= 0xff; if (prio == 1) __builtin_amdgcn_s_setprio(1); else if (prio == 2) __builtin_amdgcn_s_setprio(2); else if (prio == 3) __builtin_amdgcn_s_setprio(3); }
  extern __shared__ uint32_t lds_raw[]; Fr29 *tile = reinterpret_cast<Fr29 *>(lds_raw);
  const bool second = blockIdx.x >= ja.tiles; const NttJob &j = second ? jb : ja; const uint32_t bid = blockIdx.x - (second ? ja.tiles : 0u);
  const int logn = j.logn, log_n1 = j.log_n1, logC = j.logC;
  const Fr *__restrict__ tw261 = j.tw261;
  const Fr *__restrict__ post = j.factor;
  const Fr *__restrict__ out261 = j.out261;
  const int log_n2 = logn - log_n1; const uint32_t n2 = 1u << log_n2, C = 1u << logC, r0 = bid << logC, elems = n2 << logC;
  const Fr *s = j.src + blockIdx.y * j.stride_in; Fr *d = j.dst + blockIdx.y * j.stride_out;
  for (uint32_t w = threadIdx.x; w < elems; w += blockDim.x) {
    uint32_t c = w >> log_n2, i2 = w & (n2 - 1);
    tile[ntt_pad((i2 << logC) + c)] = ntt29_from_words(s[((size_t)(r0 + c) << log_n2) + i2]);
  }
  __syncthreads();
  ntt29_lds_transform(tile, tile + ntt_pad(elems), log_n2, logC, tw261, 1u << log_n1, radix_log);
  const Fr29 one = ntt29_from_words(j.scale261);   // 2^261 mod r when nothing is to be scaled: the product then only brings the element back below 2 r
  for (uint32_t w = threadIdx.x; w < elems; w += blockDim.x) {
    uint32_t k2 = w >> logC, c = w & (C - 1), p = bitrev32(k2, log_n2), o = (k2 << log_n1) + r0 + c;
    const Fr29 f = out261 ? ntt29_from_words(out261[o]) : one;
    Fr v = ntt29_to_words(Fr29::mul(f, tile[ntt_pad((p << logC) + c)]));
    if (post) v = v * post[o];
    d[o] = v;
  }
}

// a[i] *= table[i]
__global__ void k_fr_mul_table(Fr *__restrict__ a, const Fr *__restrict__ table, uint32_t n, size_t stride) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; Fr *d = a + blockIdx.y * stride; d[i] = d[i] * table[i];
}
// h[i] = (a[i]*b[i] - c[i]) * zinv[i or 0]   (r1cs_to_qap.tcc:281-310: H_tmp = A*B - C, then divide_by_Z_on_coset)
__global__ void k_qap_pointwise(Fr *__restrict__ a, const Fr *__restrict__ b, const Fr *__restrict__ c, const Fr *__restrict__ zinv, int zinv_is_table,
    uint32_t n) {
  // c == nullptr: the C polynomial is folded into the L query (ecntt.cuh)
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fr t = a[i] * b[i];
  if (c) t = t - c[i];
  a[i] = t * zinv[zinv_is_table ? i : 0];
}
__global__ void k_fr_to_mont(Fr *__restrict__ a, uint32_t n) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i].to_mont(); }
__global__ void k_fr_from_mont(Fr *__restrict__ a, uint32_t n) { uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) a[i] = a[i].from_mont(); }

constexpr uint32_t R1CS_LONG_ROW = 16;   // rows with more terms than this get a whole wave (k_r1cs_long_rows3)
// ---- R1CS rows times assignment (kernel K1; r1cs_to_qap.tcc:224-236,281-285; linear_combination::evaluate) ---------
// CSR with coefficient *indices* into a small table (the circuits use a few hundred distinct coefficients: +-1, +-2^k; table slots 0 / 1 are +1 / -1 and skip
// the multiply). Rows with more than R1CS_LONG_ROW terms in any matrix (bit-packing constraints: 32 ... 253 terms) get one wave each (k_r1cs_long_rows3), the
// rest one lane. All three matrices in one launch, the evaluation vectors completed (input-consistency rows
// r1cs_to_qap.tcc:227-230, zero padding up to the domain size) and the satisfiability test a*b == c (protoboard::is_satisfied, sendcgo.cpp:209) done on the
// values while they are in registers.  A violated row stores `seq` (the number of this evaluation) to *fail, a word in mapped host memory: no reset, no copy.
struct R1csMatrices { const uint32_t *rowptr[3], *col[3], *cid[3]; };
__device__ __forceinline__ Fr r1cs_row_dot(const R1csMatrices &M, int mm, uint32_t r, const Fr *__restrict__ ctab, const Fr *__restrict__ z) {
  Fr acc = Fr::zero(); const uint32_t *col = M.col[mm], *cid = M.cid[mm];
  for (uint32_t k = M.rowptr[mm][r], e = M.rowptr[mm][r + 1]; k < e; k++) {
    uint32_t ci = cid[k];
    Fr v = z[col[k]];
    if (ci == 0) acc = acc + v;
    else if (ci == 1) acc = acc - v;
    else acc = acc + ctab[ci] * v;
  }
  return acc;
}
__global__ void __launch_bounds__(256) k_r1cs_rows3(R1csMatrices M, const Fr *__restrict__ ctab, const Fr *__restrict__ z, uint32_t n_rows, uint32_t n_inputs,
    uint32_t m, Fr *__restrict__ abc, uint32_t seq, uint32_t *fail) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; if (r >= m) return;
  if (r >= n_rows) { abc[r] = r <= n_rows + n_inputs ? z[r - n_rows] : Fr::zero(); abc[m + r] = Fr::zero(); abc[2 * (size_t)m + r] = Fr::zero(); return; }
  // k_r1cs_long_rows3
  if (M.rowptr[0][r + 1] - M.rowptr[0][r] > R1CS_LONG_ROW || M.rowptr[1][r + 1] - M.rowptr[1][r] > R1CS_LONG_ROW || M.rowptr[2][r + 1] -
      M.rowptr[2][r] > R1CS_LONG_ROW) return;
  Fr a = r1cs_row_dot(M, 0, r, ctab, z), b = r1cs_row_dot(M, 1, r, ctab, z), c = r1cs_row_dot(M, 2, r, ctab, z);
  abc[r] = a; abc[m + r] = b; abc[2 * (size_t)m + r] = c;
  if (a * b != c) { fail[1] = r; *fail = seq; }
}
__global__ void __launch_bounds__(64) k_r1cs_long_rows3(const uint32_t *__restrict__ rows, R1csMatrices M, const Fr *__restrict__ ctab,
    const Fr *__restrict__ z, uint32_t m, Fr *__restrict__ abc, uint32_t seq, uint32_t *fail) {
  uint32_t r = rows[blockIdx.x], lane = threadIdx.x; Fr v[3];
#pragma unroll
  for (int mm = 0; mm < 3; mm++) { Fr acc = Fr::zero(); const uint32_t *col = M.col[mm], *cid = M.cid[mm];
    for (uint32_t k = M.rowptr[mm][r] + lane, e = M.rowptr[mm][r + 1]; k < e; k += 64) {
      uint32_t ci = cid[k];
      Fr x = z[col[k]];
      if (ci == 0) acc = acc + x;
      else if (ci == 1) acc = acc - x;
      else acc = acc + ctab[ci] * x;
    }
    // (wave-uniform) the long matrix of a packing constraint has 32..35 terms, the other two have one: no tree for those
    const uint32_t len = M.rowptr[mm][r + 1] - M.rowptr[mm][r];
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
      if ((uint32_t)d >= len) continue;
      Fr o;
      for (int i = 0; i < 8; i++) o.l[i] = __shfl_down(acc.l[i], d, 64);
      acc = acc + o;
    }
    v[mm] = acc; }
  if (lane == 0) { abc[r] = v[0]; abc[m + r] = v[1]; abc[2 * (size_t)m + r] = v[2]; if (v[0] * v[1] != v[2]) { fail[1] = r; *fail = seq; } }
}
// Assignment upload in compact form: 97 % of a BlockMaze witness are the bits 0 and 1, so the host sends two bitmaps (value is `one` / value is something
// else), the running count of "something else" per 64 entries and only those values (0.3 MB instead of 7.3 MB over PCIe); this kernel rebuilds the vector. tags
// (optional): one byte per variable — 0 the value is zero, 1 it is one, 2 anything else — for the kernels that need not look at the 32-byte value of a bit
// (k_r1cs_rows_tagged below, k_wsort_tagged in msm.cuh); other_vars (optional): the list of the variables tagged 2, in the order of `values`.
constexpr uint8_t ZTAG_ZERO = 0, ZTAG_ONE = 1, ZTAG_OTHER = 2;
// canon_bm (optional): the variables whose value arrives canonical and is brought into Montgomery form here (a host-buffer assignment: all of them, i.e.
// other_bm itself; a circuit board: its small integers, circuit::Board::TAG_SMALL) — the device has the multipliers to spare, the calling thread does not.
__global__ void k_expand_witness(const uint64_t *__restrict__ ones_bm, const uint64_t *__restrict__ other_bm, const uint64_t *__restrict__ canon_bm,
    const uint32_t *__restrict__ block_off, const Fr *__restrict__ values, Fr one_value, uint32_t n,
                                 Fr *__restrict__ out, uint8_t *__restrict__ tags, uint32_t *__restrict__ other_vars) {
  zk_take_prio(n);
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; const uint32_t wd = i >> 6, bit = i & 63; const uint64_t ob = other_bm[wd];
  // (a canonical assignment: only these 3 % need the conversion, one_value is the Montgomery one)
  if ((ob >> bit) & 1) {
    const uint32_t at = block_off[wd] + (uint32_t)__popcll(ob & ((1ull << bit) - 1));
    Fr v = values[at];
    if (canon_bm && ((canon_bm[wd] >> bit) & 1)) v = v.to_mont();
    out[i] = v;
    if (tags) tags[i] = ZTAG_OTHER;
    if (other_vars) other_vars[at] = i;
  }
  else { const bool is_one = (ones_bm[wd] >> bit) & 1; out[i] = is_one ? one_value : Fr::zero(); if (tags) tags[i] = is_one ? ZTAG_ONE : ZTAG_ZERO; }
}
// The same tags and list for an assignment that is ALREADY in device memory as plain field elements (Prover::prove_stashed: a statement resident in HBM is the raw
// vector, n x 32 B, nothing derived from it): one lane per variable reads the value, compares it with 0 and with the Montgomery one — the classification libsnark's
// multi_exp_with_mixed_addition makes per call (multiexp.tcc:443-496) —, writes the tag byte and appends the variable to the list of other values (the waves'
// counts meet in LDS, ONE atomic per workgroup on *count; the list's order is therefore by workgroup arrival, as the host scan's chunks leave it).  The two
// counters alternate between calls: this one clears the next call's.  A 7.3 MB streaming pass for send.
constexpr uint32_t CLASSIFY_PER_LANE = 8;       // variables a lane classifies (strided by the workgroup's width): 2,048 a workgroup, ONE atomic on the list's counter each
__global__ void __launch_bounds__(256) k_classify_witness(const Fr *__restrict__ z, Fr one_value, uint32_t n, uint8_t *__restrict__ tags,
    uint32_t *__restrict__ other_vars, uint32_t *__restrict__ count, uint32_t *__restrict__ count_next) {
  zk_take_prio(n);
  // (887 workgroups of one variable a lane spent 9 of their 13.5 us queueing for the one counter: same-address atomics are served one after the other)
  __shared__ uint32_t wave_n[4], wg_at;
  if (blockIdx.x == 0 && threadIdx.x == 0) *count_next = 0;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, base = blockIdx.x * (256u * CLASSIFY_PER_LANE) + threadIdx.x;
  uint4 lo[CLASSIFY_PER_LANE], hi[CLASSIFY_PER_LANE];
#pragma unroll
  for (uint32_t k = 0; k < CLASSIFY_PER_LANE; k++) { const uint32_t i = base + k * 256u; if (i < n) { const uint4 *p = reinterpret_cast<const uint4 *>(z + i); lo[k] = p[0]; hi[k] = p[1]; } }
  uint32_t mine = 0, before[CLASSIFY_PER_LANE], wave_total = 0;          // mine: bit k = this lane's k-th variable is an "other" one; before[k]: such variables of the wave ahead of it
#pragma unroll
  for (uint32_t k = 0; k < CLASSIFY_PER_LANE; k++) { const uint32_t i = base + k * 256u; bool other = false;
    if (i < n) {
      const uint32_t any = lo[k].x | lo[k].y | lo[k].z | lo[k].w | hi[k].x | hi[k].y | hi[k].z | hi[k].w;
      const uint32_t d1 = (lo[k].x ^ one_value.l[0]) | (lo[k].y ^ one_value.l[1]) | (lo[k].z ^ one_value.l[2]) | (lo[k].w ^ one_value.l[3]) | (hi[k].x ^ one_value.l[4]) |
          (hi[k].y ^ one_value.l[5]) | (hi[k].z ^ one_value.l[6]) | (hi[k].w ^ one_value.l[7]);
      other = any != 0 && d1 != 0; tags[i] = any == 0 ? ZTAG_ZERO : d1 == 0 ? ZTAG_ONE : ZTAG_OTHER; }
    const uint64_t m = __ballot(other); before[k] = wave_total + (uint32_t)__popcll(m & ((1ull << lane) - 1)); wave_total += (uint32_t)__popcll(m); mine |= (uint32_t)other << k; }
  if (lane == 0) wave_n[wave] = wave_total;
  __syncthreads();
  if (threadIdx.x == 0) { const uint32_t tot = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3]; wg_at = tot ? atomicAdd(count, tot) : 0u; }
  __syncthreads();
  if (mine) { uint32_t at = wg_at; for (uint32_t wv = 0; wv < wave; wv++) at += wave_n[wv];
#pragma unroll
    for (uint32_t k = 0; k < CLASSIFY_PER_LANE; k++) if ((mine >> k) & 1) other_vars[at + before[k]] = base + k * 256u; }
}
// The hand-over of a circuit board (Prover::set_witness_board, the cgo path): the board's tag bytes as they are (0, 1, 2 = a Montgomery value in the board's wide
// array, 6 = a small integer kept canonical) and the values of the board's candidates — the variables that ever held something else than 0 / 1, a fixed list per
// circuit object — in the list's order.  Workgroups [0, tag_blocks): one lane per variable — zeros and ones from the byte; the others wait for their value.
// The workgroups after them: one lane per candidate — if its tag says "other" the value goes to its place (a small integer is brought into Montgomery form here)
// and the variable into the list of other values (one atomic per workgroup; the two counters alternate as in k_classify_witness).
__global__ void __launch_bounds__(256) k_expand_board(const uint8_t *__restrict__ board_tags, uint32_t n, uint32_t tag_blocks, const uint32_t *__restrict__ cand,
    const Fr *__restrict__ cand_vals, uint32_t n_cand, Fr one_value, Fr *__restrict__ z, uint8_t *__restrict__ tags, uint32_t *__restrict__ other_vars,
    uint32_t *__restrict__ count, uint32_t *__restrict__ count_next) {
  zk_take_prio(n);
  const bool by_var = n_cand >> 31; n_cand &= 0x7fffffffu;        // (cand_vals: the board's whole array of values, read in place, instead of the gathered list)
  if (blockIdx.x < tag_blocks) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *count_next = 0;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; const uint8_t t = board_tags[i];
    if (t & 2) tags[i] = ZTAG_OTHER; else { tags[i] = t ? ZTAG_ONE : ZTAG_ZERO; z[i] = t ? one_value : Fr::zero(); }
    return;
  }
  __shared__ uint32_t wave_n[4], wg_at;
  const uint32_t j = (blockIdx.x - tag_blocks) * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6; bool other = false; uint32_t var = 0;
  if (j < n_cand) { var = cand[j]; const uint8_t t = board_tags[var];
    if (t & 2) { other = true; Fr v = cand_vals[by_var ? var : j];
      if (t & 4) { for (int k = 2; k < 8; k++) v.l[k] = 0; v = v.to_mont(); }                 // (a small integer: only its low 64 bits are meaningful on the board)
      z[var] = v; } }
  const uint64_t m = __ballot(other);
  if (lane == 0) wave_n[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) { const uint32_t tot = wave_n[0] + wave_n[1] + wave_n[2] + wave_n[3]; wg_at = tot ? atomicAdd(count, tot) : 0u; }
  __syncthreads();
  if (other) { uint32_t at = wg_at; for (uint32_t wv = 0; wv < wave; wv++) at += wave_n[wv]; other_vars[at + (uint32_t)__popcll(m & ((1ull << lane) - 1))] = var; }
}
// Variables whose columns coincide in all three matrices of the constraint system (mint and deposit each hold such a pair) have EQUAL points in every query, and
// an incomplete addition that meets P + P leaves ZZ = 0: the MSM was then repeated on the general path (3 of 57,600 mixed proofs in round 5, whenever both values
// fell into one bucket on two lanes).  Equal columns make the assignment with z_a + z_b in one place and 0 in the other EQUIVALENT — the same A z, B z, C z, the same
// sums over every query — so the hand-over ends with this kernel: one lane per group of equal columns; if any member is tagged "other", the first such member takes
// the sum and keeps its tag (it is in the list of other values), the rest become zeros (a former "other" stays in the list, where its zero is skipped).  Groups of
// bits stay as they are: ones are summed by the lanes' complete accumulation, never as two equal partial sums.  Idempotent.  tags = null (dense hand-over): the
// first member takes the sum.
__global__ void k_merge_equal_columns(Fr *__restrict__ z, uint8_t *__restrict__ tags, const uint32_t *__restrict__ grp_ptr, const uint32_t *__restrict__ grp_mem,
    uint32_t n_groups) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; if (g >= n_groups) return;
  const uint32_t lo = grp_ptr[g], hi = grp_ptr[g + 1]; uint32_t target = lo;
  if (tags) { target = hi; for (uint32_t k = lo; k < hi; k++) if (tags[grp_mem[k]] == ZTAG_OTHER) { target = k; break; } if (target == hi) return; }
  Fr s = Fr::zero(); for (uint32_t k = lo; k < hi; k++) s = s + z[grp_mem[k]];
  for (uint32_t k = lo; k < hi; k++) { const uint32_t v = grp_mem[k]; if (k == target) z[v] = s; else { z[v] = Fr::zero(); if (tags) tags[v] = ZTAG_ZERO; } }
}
// both of the above in ONE launch (the two are independent and each too small to fill the chip for long: 27 + 25 us one after the other at the head of every
// proof's critical chain): the first `short_blocks` workgroups take the one-lane rows, the others four long rows each, one per wave
__global__ void __launch_bounds__(256) k_r1cs_rows_all(R1csMatrices M, const Fr *__restrict__ ctab, const Fr *__restrict__ z, uint32_t n_rows,
    uint32_t n_inputs, uint32_t m, const uint32_t *__restrict__ long_rows, uint32_t n_long, uint32_t short_blocks,
                                                       Fr *__restrict__ abc, uint32_t seq, uint32_t *fail) {
  if (blockIdx.x < short_blocks) {
    uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; if (r >= m) return;
    if (r >= n_rows) { abc[r] = r <= n_rows + n_inputs ? z[r - n_rows] : Fr::zero(); abc[m + r] = Fr::zero(); abc[2 * (size_t)m + r] = Fr::zero(); return; }
    if (M.rowptr[0][r + 1] - M.rowptr[0][r] > R1CS_LONG_ROW || M.rowptr[1][r + 1] - M.rowptr[1][r] > R1CS_LONG_ROW || M.rowptr[2][r + 1] -
        M.rowptr[2][r] > R1CS_LONG_ROW) return;
    Fr a = r1cs_row_dot(M, 0, r, ctab, z), b = r1cs_row_dot(M, 1, r, ctab, z), c = r1cs_row_dot(M, 2, r, ctab, z);
    abc[r] = a; abc[m + r] = b; abc[2 * (size_t)m + r] = c;
    if (a * b != c) { fail[1] = r; *fail = seq; }
    return; }
  const uint32_t w = (blockIdx.x - short_blocks) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63; if (w >= n_long) return;
  const uint32_t r = long_rows[w]; Fr v[3];
#pragma unroll
  for (int mm = 0; mm < 3; mm++) { Fr acc = Fr::zero(); const uint32_t *col = M.col[mm], *cid = M.cid[mm];
    for (uint32_t k = M.rowptr[mm][r] + lane, e = M.rowptr[mm][r + 1]; k < e; k += 64) {
      uint32_t ci = cid[k];
      Fr x = z[col[k]];
      if (ci == 0) acc = acc + x;
      else if (ci == 1) acc = acc - x;
      else acc = acc + ctab[ci] * x;
    }
    const uint32_t len = M.rowptr[mm][r + 1] - M.rowptr[mm][r];
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
      if ((uint32_t)d >= len) continue;
      Fr o;
      for (int i = 0; i < 8; i++) o.l[i] = __shfl_down(acc.l[i], d, 64);
      acc = acc + o;
    }
    v[mm] = acc; }
  if (lane == 0) { abc[r] = v[0]; abc[m + r] = v[1]; abc[2 * (size_t)m + r] = v[2]; if (v[0] * v[1] != v[2]) { fail[1] = r; *fail = seq; } }
}
// ---- the same evaluation for an assignment that arrived in compact form (k_expand_witness wrote a tag per variable)
// --------------------------------------------------- 97 % of a BlockMaze assignment are the bits 0 and 1. A term whose variable is 0 contributes nothing and
// is skipped after ONE byte load; a term whose variable is 1 contributes its coefficient — a 32-byte load from the small coefficient table and a modular
// addition, no product, whatever the coefficient (k_r1cs_rows_all runs the product path in every iteration in which ANY lane of the wave meets a coefficient
// other than +-1, i.e. nearly always: ~450 instructions per term); only a variable holding something else (packed words, field values: 3 %) loads its value and
// pays the product. Rows of one gadget are neighbours, so whole waves skip the product path. write_c = 0: the C polynomial is folded into the L query
// (ecntt.cuh) — its row value is still formed for the satisfiability test but not stored.
__device__ __forceinline__ Fr r1cs_term_tagged(const Fr &acc, uint32_t ci, uint32_t col, const Fr *__restrict__ ctab, const Fr *__restrict__ z,
    const uint8_t *__restrict__ tags) {
  const uint8_t t = tags[col];
  if (t == ZTAG_ZERO) return acc;
  const Fr coef = ctab[ci];
  if (t == ZTAG_ONE) return acc + coef;
  const Fr v = z[col];
  if (ci == 0) return acc + v;
  if (ci == 1) return acc - v;
  return acc + coef * v;
}
// One lane, one row, all three matrices. The walk is a chain of dependent loads (row pointer -> column index -> tag -> coefficient), and a lane that takes them
// one term at a time spends the kernel waiting for the L2: the first R1CS_HEAD terms of each matrix are therefore fetched together — 6 row pointers, then 24
// indices, then 12 tags in flight at once — and only rows with more terms than that (up to R1CS_LONG_ROW) go on term by term.
constexpr int R1CS_HEAD = 4;
__device__ __forceinline__ void r1cs_rows_tagged_lane(const R1csMatrices &M, uint32_t r, const Fr *__restrict__ ctab, const Fr *__restrict__ z,
    const uint8_t *__restrict__ tags, Fr (&out)[3]) {
  uint32_t beg[3], end[3];
#pragma unroll
  for (int mm = 0; mm < 3; mm++) { beg[mm] = M.rowptr[mm][r]; end[mm] = M.rowptr[mm][r + 1]; }
  uint32_t col[3][R1CS_HEAD], cid[3][R1CS_HEAD];
#pragma unroll
  for (int mm = 0; mm < 3; mm++) {
#pragma unroll
    for (int j = 0; j < R1CS_HEAD; j++) {
      const bool valid = beg[mm] + j < end[mm];
      col[mm][j] = valid ? M.col[mm][beg[mm] + j] : 0u;
      cid[mm][j] = valid ? M.cid[mm][beg[mm] + j] : 0u;
    }
  }
  uint8_t tag[3][R1CS_HEAD];
#pragma unroll
  for (int mm = 0; mm < 3; mm++) {
#pragma unroll
    for (int j = 0; j < R1CS_HEAD; j++) tag[mm][j] = beg[mm] + j < end[mm] ? tags[col[mm][j]] : ZTAG_ZERO;
  }
#pragma unroll
  for (int mm = 0; mm < 3; mm++) {
    Fr acc = Fr::zero();
#pragma unroll
    for (int j = 0; j < R1CS_HEAD; j++) {
      if (tag[mm][j] == ZTAG_ZERO) continue;
      const Fr coef = ctab[cid[mm][j]];
      if (tag[mm][j] == ZTAG_ONE) { acc = acc + coef; continue; }
      const Fr v = z[col[mm][j]];
      if (cid[mm][j] == 0) acc = acc + v;
      else if (cid[mm][j] == 1) acc = acc - v;
      else acc = acc + coef * v;
    }
    for (uint32_t k = beg[mm] + R1CS_HEAD; k < end[mm]; k++) acc = r1cs_term_tagged(acc, M.cid[mm][k], M.col[mm][k], ctab, z, tags);
    out[mm] = acc;
  }
}
__global__ void __launch_bounds__(256) k_r1cs_rows_tagged(R1csMatrices M, const Fr *__restrict__ ctab, const Fr *__restrict__ z,
    const uint8_t *__restrict__ tags, uint32_t n_rows, uint32_t n_inputs, uint32_t m,
                                                          const uint32_t *__restrict__ long_rows, uint32_t n_long, uint32_t short_blocks, int write_c,
                                                              Fr *__restrict__ abc, uint32_t seq, uint32_t *fail) {
  zk_take_prio(n_inputs);
  if (blockIdx.x < short_blocks) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m) return;
    if (r >= n_rows) {                                                                   // input-consistency rows and the zero padding up to the domain size
      abc[r] = r <= n_rows + n_inputs ? z[r - n_rows] : Fr::zero();
      abc[m + r] = Fr::zero();
      if (write_c) abc[2 * (size_t)m + r] = Fr::zero();
      return;
    }
    if (M.rowptr[0][r + 1] - M.rowptr[0][r] > R1CS_LONG_ROW || M.rowptr[1][r + 1] - M.rowptr[1][r] > R1CS_LONG_ROW || M.rowptr[2][r + 1] -
        M.rowptr[2][r] > R1CS_LONG_ROW) return;
    Fr v[3];
    r1cs_rows_tagged_lane(M, r, ctab, z, tags, v);
    abc[r] = v[0];
    abc[m + r] = v[1];
    if (write_c) abc[2 * (size_t)m + r] = v[2];
    if (v[0] * v[1] != v[2]) { fail[1] = r; *fail = seq; }
    return;
  }
  const uint32_t w = (blockIdx.x - short_blocks) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // a row of more than R1CS_LONG_ROW terms: one wave
  if (w >= n_long) return;
  const uint32_t r = long_rows[w];
  Fr v[3];
#pragma unroll
  for (int mm = 0; mm < 3; mm++) {
    Fr acc = Fr::zero();
    const uint32_t *col = M.col[mm], *cid = M.cid[mm];
    for (uint32_t k = M.rowptr[mm][r] + lane, e = M.rowptr[mm][r + 1]; k < e; k += 64) acc = r1cs_term_tagged(acc, cid[k], col[k], ctab, z, tags);
    // (wave-uniform) the long matrix of a packing constraint has 32..35 terms, the other two have one
    const uint32_t len = M.rowptr[mm][r + 1] - M.rowptr[mm][r];
#pragma unroll 1
    for (int d = 32; d >= 1; d >>= 1) {
      if ((uint32_t)d >= len) continue;
      Fr o;
      for (int i = 0; i < 8; i++) o.l[i] = __shfl_down(acc.l[i], d, 64);
      acc = acc + o;
    }
    v[mm] = acc;
  }
  if (lane == 0) {
    abc[r] = v[0];
    abc[m + r] = v[1];
    if (write_c) abc[2 * (size_t)m + r] = v[2];
    if (v[0] * v[1] != v[2]) { fail[1] = r; *fail = seq; }
  }
}
// satisfiability: flag[0] |= (a[i]*b[i] != c[i]) over the constraint rows (protoboard::is_satisfied, sendcgo.cpp:209)
__global__ void k_r1cs_check(const Fr *__restrict__ a, const Fr *__restrict__ b, const Fr *__restrict__ c, uint32_t n_rows, uint32_t *flag) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; if (r >= n_rows) return; if (a[r] * b[r] != c[r]) atomicOr(flag, 1u);
}

}  // namespace zk
