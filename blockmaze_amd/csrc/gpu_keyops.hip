// Key-side device operations: batched point decompression and fixed-base batch exponentiation.
#include <cstring>
#include "gpu_internal.hpp"
#include "keyops.cuh"

namespace zk {
// ======================================================================================================================
// Key-side operations
// ======================================================================================================================
static FqPowConsts fq_pow_consts() {
  FqPowConsts pc; auto shr = [](const uint32_t *a, int k, uint32_t *o) { for (int i = 0; i < 8; i++) o[i] = (a[i] >> k) | (i < 7 ? a[i + 1] << (32 - k) : 0); };
  uint32_t t[8]; uint64_t c = 1; for (int i = 0; i < 8; i++) { c += FqParams::MOD[i]; t[i] = (uint32_t)c; c >>= 32; } shr(t, 2, pc.sqrt_exp);     // (q+1)/4
  // (q-3)/4
  uint64_t b = 3;
  for (int i = 0; i < 8; i++) {
    uint64_t d = (uint64_t)FqParams::MOD[i] - b;
    t[i] = (uint32_t)d;
    b = (d >> 32) & 1;
  }
  shr(t, 2, pc.qm3o4);
  // (q-1)/2
  b = 1;
  for (int i = 0; i < 8; i++) {
    uint64_t d = (uint64_t)FqParams::MOD[i] - b;
    t[i] = (uint32_t)d;
    b = (d >> 32) & 1;
  }
  shr(t, 1, pc.qm1o2);
  return pc;
}
static void check_bad(DevBuf<uint32_t> &bad, const char *what) {
  uint32_t h = 0;
  bad.download(&h, 1);
  if (h) throw GpuError(std::string(what) + ": " + std::to_string(h) + " x-coordinates are not on the curve");
}
void decompress_g1(const Fe32 *xs, const uint8_t *flags, size_t n, G1AffineRaw *out) {
  if (!n) return;
  DevBuf<Fe32> dx(n);
  DevBuf<uint8_t> df(n);
  DevBuf<G1AffineRaw> dout(n);
  DevBuf<uint32_t> bad(1);
  dx.upload(xs, n);
  df.upload(flags, n);
  bad.zero();
  hipLaunchKernelGGL(k_g1_decompress, dim3(cdiv(n, 128)), dim3(128), 0, gpu().stream, (const Fq *)dx.get(), df.get(), (Affine<Fq> *)dout.get(), (uint32_t)n,
      fq_pow_consts(), bad.get());
  HIP_CHECK(hipGetLastError()); dout.download(out, n); check_bad(bad, "G1 decompression");
}
void decompress_g2(const Fe32 *xs, const uint8_t *flags, size_t n, G2AffineRaw *out) {
  if (!n) return;
  DevBuf<Fe32> dx(2 * n);
  DevBuf<uint8_t> df(n);
  DevBuf<G2AffineRaw> dout(n);
  DevBuf<uint32_t> bad(1);
  dx.upload(xs, 2 * n);
  df.upload(flags, n);
  bad.zero();
  // 3 / (9 + u) (alt_bn128_init.cpp:193)
  host::HFq2 tb = host::HFq2{host::HFq::from_u64(3), host::HFq::zero()} * host::HFq2{host::HFq::from_u64(9), host::HFq::one()}.inv();
  Fq2 twist_b; memcpy(&twist_b.c0, tb.c0.l, 32); memcpy(&twist_b.c1, tb.c1.l, 32);
  hipLaunchKernelGGL(k_g2_decompress, dim3(cdiv(n, 64)), dim3(64), 0, gpu().stream, (const Fq2 *)dx.get(), df.get(), (Affine<Fq2> *)dout.get(), (uint32_t)n,
      fq_pow_consts(), twist_b, bad.get());
  HIP_CHECK(hipGetLastError()); dout.download(out, n); check_bad(bad, "G2 decompression");
}
template <class HF, class Raw> static void store_affine(const host::HPoint<HF> &p, Raw &o) {
  HF x, y;
  p.to_affine(x, y);
  memcpy(&o, &x, sizeof(HF));
  memcpy((char *)&o + sizeof(HF), &y, sizeof(HF));
}
template <class HF, class F, class Raw> static void fixed_base_mul(const host::HPoint<HF> &base, const Fe32 *scalars, size_t n, Raw *out) {
  if (!n) return; std::vector<Raw> table(32 * 255); host::HPoint<HF> wbase = base;
  // after 255 additions acc = 256 * wbase
  for (int w = 0; w < 32; w++) {
    host::HPoint<HF> acc = wbase;
    for (int d = 1; d <= 255; d++) {
      store_affine(acc, table[w * 255 + d - 1]);
      acc = acc.add(wbase);
    }
    wbase = acc;
  }
  DevBuf<Raw> dt(table.size()), dout(n); DevBuf<Fe32> ds(n); dt.upload(table.data(), table.size()); ds.upload(scalars, n);
  hipLaunchKernelGGL((k_fixed_base_mul<F>), dim3(cdiv(n, 128)), dim3(128), 0, gpu().stream, (const Affine<F> *)dt.get(), (const Fr *)ds.get(),
      (Affine<F> *)dout.get(), (uint32_t)n);
  HIP_CHECK(hipGetLastError()); dout.download(out, n);
}
void fixed_base_mul_g1(const host::HG1 &base, const Fe32 *scalars, size_t n, G1AffineRaw *out) {
  fixed_base_mul<host::HFq, Fq, G1AffineRaw>(base, scalars, n, out);
}
void fixed_base_mul_g2(const host::HG2 &base, const Fe32 *scalars, size_t n, G2AffineRaw *out) {
  fixed_base_mul<host::HFq2, Fq2, G2AffineRaw>(base, scalars, n, out);
}

}  // namespace zk
