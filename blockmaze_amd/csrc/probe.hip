// Device-arithmetic probes: run the __device__ field / curve primitives on arrays so that parity tests can compare them
// with the CPU oracle element by element (zkgpu_test_* in include/zkgpu.h).  Inputs and outputs are canonical; the
// kernels convert to Montgomery form and back on the device.
#include <hip/hip_runtime.h>
#include "gpu.hpp"
#include "curve.cuh"

namespace zk {
extern hipStream_t gpu_stream();
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw GpuError(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

template <class F> __global__ void k_probe_field(int op, const F *a, const F *b, F *out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; F x = a[i].to_mont(), y = b ? b[i].to_mont() : F::zero(), r;
  switch (op) {
    case 0: r = x * y;
    break;
    case 1: r = x + y;
    break;
    case 2: r = x - y;
    break;
    case 3: r = x.inv();
    break;
    case 4: r = x.sqr();
    break;
    case 5: r = x.neg();
    break;
    // the lazy domain (field.cuh): operands pushed towards 2p first (x + p, y + p are the same residues), results normalized
    case 6: { F xl = x, yl = y, pm; for (int j = 0; j < 8; j++) pm.l[j] = F::modulus_limb(j); F::add_raw(xl, pm); F::add_raw(yl, pm);
        r = F::mul_lazy(xl, yl).normalize(); break; }
    case 7: { F xl = x, pm; for (int j = 0; j < 8; j++) pm.l[j] = F::modulus_limb(j); F::add_raw(xl, pm); r = F::sqr_lazy(xl).normalize(); break; }
    case 8: { F xl = x, yl = y, pm; for (int j = 0; j < 8; j++) pm.l[j] = F::modulus_limb(j); if (i & 1) F::add_raw(xl, pm); if (i & 2) F::add_raw(yl, pm);
        r = F::sub_lazy(xl, yl).normalize(); break; }
    default: { r = x; F::neg_masked(r, (i & 1) ? 0xffffffffu : 0u); r = (i & 1) ? r.normalize() : r.neg(); } }   // 9: -x both ways
  out[i] = r.from_mont();
}
__device__ __forceinline__ Fq2 fq2_in(const Fq2 &v) { return {v.c0.to_mont(), v.c1.to_mont()}; }
__device__ __forceinline__ Fq2 fq2_out(const Fq2 &v) { return {v.c0.from_mont(), v.c1.from_mont()}; }
__global__ void k_probe_fq2(int op, const Fq2 *a, const Fq2 *b, Fq2 *out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return; Fq2 x = fq2_in(a[i]), y = b ? fq2_in(b[i]) : Fq2::zero(), r;
  switch (op) { case 0: r = x * y; break; case 1: r = x.sqr(); break; default: r = x.inv(); }
  out[i] = fq2_out(r);
}
__device__ __forceinline__ Fq fin(const Fq &v) { return v.to_mont(); }
__device__ __forceinline__ Fq2 fin(const Fq2 &v) { return fq2_in(v); }
__device__ __forceinline__ Fq fout(const Fq &v) { return v.from_mont(); }
__device__ __forceinline__ Fq2 fout(const Fq2 &v) { return fq2_out(v); }
template <class F> __device__ __forceinline__ Affine<F> to_affine(const XYZZ<F> &p) {
  if (p.is_inf()) return Affine<F>::inf();
  F zi = p.ZZ.inv(), z3i = p.ZZZ.inv();
  return {p.X * zi, p.Y * z3i};
}
// a, b, out: affine canonical.  OP: 0 add of a point with a non-trivial ZZ, 1 dbl, 2 madd, 3 mul_small.
// One kernel per operation: with all of them inlined into one function hipcc 7.2's register allocator crashed (RAGreedy, splitSeparateComponents).
template <int OP, class F> __global__ void k_probe_group(const Affine<F> *a, const Affine<F> *b, Affine<F> *out, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  Affine<F> pa = {fin(a[i].x), fin(a[i].y)}; XYZZ<F> acc = XYZZ<F>::from_affine(pa);
  // o = 2b - b: a non-trivial ZZ, exercises the general formulas
  if constexpr (OP == 0) {
    Affine<F> pb = {fin(b[i].x), fin(b[i].y)};
    XYZZ<F> o = XYZZ<F>::from_affine(pb);
    o = o.dbl_inl();
    o.add_inl(XYZZ<F>::from_affine(pb).neg());
    acc.add_inl(o);
  }
  // 4a - 2a = 2a through dbl() of a non-affine point
  else if constexpr (OP == 1) {
    acc = acc.dbl_inl().dbl_inl();
    acc.add_inl(XYZZ<F>::from_affine(pa).dbl_inl().neg());
  }
  else if constexpr (OP == 2) { Affine<F> pb = {fin(b[i].x), fin(b[i].y)}; acc.madd_inl(pb); }
  else if constexpr (OP == 3) { uint32_t k = reinterpret_cast<const uint32_t *>(&b[i])[0]; acc = acc.mul_small(k); }
  Affine<F> r = to_affine(acc); out[i] = {fout(r.x), fout(r.y)};
}

template <class T, class K> static void run_probe(K launch, const T *a, const T *b, T *out, size_t n) {
  DevBuf<uint8_t> da(n * sizeof(T)), db(b ? n * sizeof(T) : 1), dout(n * sizeof(T));
  da.upload((const uint8_t *)a, n * sizeof(T)); if (b) db.upload((const uint8_t *)b, n * sizeof(T));
  launch((const T *)da.get(), b ? (const T *)db.get() : nullptr, (T *)dout.get());
  HIP_CHECK(hipGetLastError()); dout.download((uint8_t *)out, n * sizeof(T));
}
void probe_field(int field, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {
  unsigned g = (unsigned)((n + 255) / 256); hipStream_t s = gpu_stream();
  if (field == 0) run_probe<Fr>([&](const Fr *x, const Fr *y, Fr *o) { hipLaunchKernelGGL(k_probe_field<Fr>, dim3(g), dim3(256), 0, s, op, x, y, o,
      (uint32_t)n); }, (const Fr *)a, (const Fr *)b, (Fr *)out, n);
  else run_probe<Fq>([&](const Fq *x, const Fq *y, Fq *o) { hipLaunchKernelGGL(k_probe_field<Fq>, dim3(g), dim3(256), 0, s, op, x, y, o, (uint32_t)n);
      }, (const Fq *)a, (const Fq *)b, (Fq *)out, n);
}
void probe_fq2(int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {
  unsigned g = (unsigned)((n + 255) / 256); hipStream_t s = gpu_stream();
  run_probe<Fq2>([&](const Fq2 *x, const Fq2 *y, Fq2 *o) { hipLaunchKernelGGL(k_probe_fq2, dim3(g), dim3(256), 0, s, op, x, y, o, (uint32_t)n);
      }, (const Fq2 *)a, (const Fq2 *)b, (Fq2 *)out, n);
}
template <class F, class A> static void probe_group_t(int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {
  unsigned g = (unsigned)((n + 63) / 64); hipStream_t s = gpu_stream();
#define ZK_PG(OP) run_probe<A>([&](const A *x, const A *y, A *o) { hipLaunchKernelGGL((k_probe_group<OP, F>), dim3(g), dim3(64), 0, s, (const Affine<F> *)x, (const Affine<F> *)y, (Affine<F> *)o, (uint32_t)n); }, (const A *)a, (const A *)b, (A *)out, n)
  switch (op) {
    case 0: ZK_PG(0);
    break;
    case 1: ZK_PG(1);
    break;
    case 2: ZK_PG(2);
    break;
    case 3: ZK_PG(3);
    break;
    default: throw GpuError("probe_group: unknown operation");
  }
#undef ZK_PG
}
void probe_group(int group, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {
  if (group == 1) probe_group_t<Fq, G1Affine>(op, a, b, out, n); else probe_group_t<Fq2, G2Affine>(op, a, b, out, n);
}
}  // namespace zk
