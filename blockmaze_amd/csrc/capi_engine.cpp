// extern "C" surface of the engine layer (include/zkgpu.h).  Catches every C++ exception at the boundary: the library is
// loaded into a Go process through cgo and must never unwind into it, abort, or install signal handlers.
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/zkgpu.h"
#include "gpu.hpp"

namespace zk {
void probe_field(int field, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void probe_fq2(int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
void probe_group(int group, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n);
}
using namespace zk;
using namespace zk::host;

static thread_local std::string g_err;
void zkgpu_set_error(const std::string &s) { g_err = s; }
std::mutex g_gpu_mutex;   // one proof pipeline at a time on the device (cgo calls arrive on arbitrary OS threads)

template <class Fn> static int guarded(Fn fn) {
  try { if (!gpu_available()) { g_err = "no HIP device visible; libzkgpu has no CPU fallback"; return ZKGPU_ERR_NO_DEVICE; }
        std::lock_guard<std::mutex> lk(g_gpu_mutex); return fn(); }
  catch (const std::exception &e) { g_err = e.what(); return ZKGPU_ERR_RUNTIME; }
  catch (...) { g_err = "unknown error"; return ZKGPU_ERR_RUNTIME; }
}

// canonical bytes <-> Montgomery records
static void g1_to_raw(const uint8_t *p, G1AffineRaw &o) {
  bool z = true;
  for (int i = 0; i < 64; i++) if (p[i]) z = false;
  if (z) {
    memset(&o, 0, sizeof o);
    return;
  }
  HFq x, y; memcpy(x.l, p, 32); memcpy(y.l, p + 32, 32); x = x.to_mont(); y = y.to_mont(); memcpy(&o.x, x.l, 32); memcpy(&o.y, y.l, 32); }
static void g2_to_raw(const uint8_t *p, G2AffineRaw &o) {
  bool z = true;
  for (int i = 0; i < 128; i++) if (p[i]) z = false;
  if (z) {
    memset(&o, 0, sizeof o);
    return;
  }
  Fe32 *dst = &o.x0; for (int k = 0; k < 4; k++) { HFq v; memcpy(v.l, p + 32 * k, 32); v = v.to_mont(); memcpy(&dst[k], v.l, 32); } }
static void g1_out(const HG1 &p, uint8_t *o) { HFq x, y; p.to_affine(x, y); x = x.from_mont(); y = y.from_mont(); memcpy(o, x.l, 32); memcpy(o + 32, y.l, 32); }
static void g2_out(const HG2 &p, uint8_t *o) {
  HFq2 x, y;
  p.to_affine(x, y);
  HFq v[4] = {x.c0.from_mont(), x.c1.from_mont(), y.c0.from_mont(), y.c1.from_mont()};
  for (int k = 0; k < 4; k++) memcpy(o + 32 * k, v[k].l, 32);
}
static int auto_window(size_t n) { int lg = 0; while (((size_t)1 << lg) < n) lg++; int c = lg - 2; if (c < 7) c = 7; if (c > 16) c = 16; return c; }

struct zkgpu_msm { int group; size_t n; std::unique_ptr<MsmG1> g1; std::unique_ptr<MsmG2> g2; DevBuf<Fe32> scalars; };

extern "C" {
const char *zkgpu_last_error(void) { return g_err.c_str(); }
const char *zkgpu_version(void) { return "blockmaze_amd 0.1 (gfx950)"; }
int zkgpu_device_count(void) { return gpu_available() ? 1 : 0; }
int zkgpu_device_numa_node(int device) {
  int node = -1;
  guarded([&] { if (gpu_available()) node = gpu_device_numa_node(device); return ZKGPU_OK; });
  return node;
}
int zkgpu_init(void) { return guarded([] { gpu(); return ZKGPU_OK; }); }

int zkgpu_test_field_op(int field, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {
  return guarded([&] { probe_field(field, op, a, b, out, n); return ZKGPU_OK; });
}
int zkgpu_test_fq2_op(int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {
  return guarded([&] { probe_fq2(op, a, b, out, n); return ZKGPU_OK; });
}
int zkgpu_test_group_op(int group, int op, const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n) {
  return guarded([&] { probe_group(group, op, a, b, out, n); return ZKGPU_OK; });
}

zkgpu_msm *zkgpu_msm_create(int group, const uint8_t *points, size_t n, int window_bits, int filter_ones) {
  zkgpu_msm *h = nullptr;
  int rc = guarded([&] { if (group != 1 && group != 2) { g_err = "group must be 1 or 2"; return ZKGPU_ERR_ARG; }
    int c = window_bits ? window_bits : auto_window(n); std::unique_ptr<zkgpu_msm> m(new zkgpu_msm); m->group = group; m->n = n;
    if (group == 1) {
      std::vector<G1AffineRaw> raw(n ? n : 1);
      for (size_t i = 0; i < n; i++) g1_to_raw(points + 64 * i, raw[i]);
      m->g1.reset(new MsmG1(raw.data(), n, c, (filter_ones & 1) != 0, true, (filter_ones & 2) != 0));
    }
    else {
      std::vector<G2AffineRaw> raw(n ? n : 1);
      for (size_t i = 0; i < n; i++) g2_to_raw(points + 128 * i, raw[i]);
      m->g2.reset(new MsmG2(raw.data(), n, c, (filter_ones & 1) != 0, true, (filter_ones & 2) != 0));
    }
    m->scalars = DevBuf<Fe32>(n ? n : 1); h = m.release(); return ZKGPU_OK; });
  return rc == ZKGPU_OK ? h : nullptr;
}
int zkgpu_msm_set_scalars(zkgpu_msm *h, const uint8_t *scalars, size_t n) { return guarded([&] { if (!h || n != h->n) { g_err = "scalar count";
    return ZKGPU_ERR_ARG; }
  if (n) { h->scalars.upload((const Fe32 *)scalars, n); fr_to_mont_dev(h->scalars.get(), n); gpu_sync(); } return ZKGPU_OK; }); }
int zkgpu_msm_run(zkgpu_msm *h, uint8_t *out) { return guarded([&] { if (!h) return ZKGPU_ERR_ARG;
  if (h->group == 1) { h->g1->run(h->scalars.get(), nullptr); g1_out(h->g1->result(), out); } else { h->g2->run(h->scalars.get(), nullptr);
      g2_out(h->g2->result(), out); } return ZKGPU_OK; }); }
void zkgpu_msm_destroy(zkgpu_msm *h) { guarded([&] { delete h; return ZKGPU_OK; }); }
static int msm_oneshot(int group, const uint8_t *points, const uint8_t *scalars, size_t n, int c, int fo, uint8_t *out) {
  zkgpu_msm *h = zkgpu_msm_create(group, points, n, c, fo);
  if (!h) return ZKGPU_ERR_RUNTIME;
  int rc = zkgpu_msm_set_scalars(h, scalars, n);
  if (rc == ZKGPU_OK) rc = zkgpu_msm_run(h, out);
  zkgpu_msm_destroy(h);
  return rc;
}
int zkgpu_msm_g1(const uint8_t *points, const uint8_t *scalars, size_t n, int c, int fo, uint8_t out[64]) {
  return msm_oneshot(1, points, scalars, n, c, fo, out);
}
int zkgpu_msm_g2(const uint8_t *points, const uint8_t *scalars, size_t n, int c, int fo, uint8_t out[128]) {
  return msm_oneshot(2, points, scalars, n, c, fo, out);
}

size_t zkgpu_domain_size(size_t min_size) {   // pure host logic: get_evaluation_domain.tcc:33-52 restricted to basic / step radix-2
  if (min_size <= 1) return 0; size_t lg = 0; while (((size_t)1 << lg) < min_size) lg++; if (min_size == ((size_t)1 << lg)) return lg <= 28 ? min_size : 0;
  size_t big = (size_t)1 << (lg - 1), small = min_size - big, rs = 1;
  while (rs < small) rs <<= 1;
  size_t m = (small == rs) ? min_size : big + rs;
  return lg <= 28 ? m : 0;
}
int zkgpu_domain_transform(size_t min_size, int op, uint8_t *data) { return guarded([&] { Domain d(min_size); size_t m = d.m(); DevBuf<Fe32> buf(m);
    buf.upload((const Fe32 *)data, m); fr_to_mont_dev(buf.get(), m);
  switch (op) {
    case 0: d.fft(buf.get(), 1, m);
    break;
    case 1: d.ifft(buf.get(), 1, m);
    break;
    case 2: d.coset_fft(buf.get(), 1, m);
    break;
    case 3: d.icoset_fft(buf.get(), 1, m);
    break;
    default: g_err = "op";
    return ZKGPU_ERR_ARG;
  }
  fr_from_mont_dev(buf.get(), m); buf.download((Fe32 *)data, m); return ZKGPU_OK; }); }

struct zkgpu_r1cs { R1csHost host; std::unique_ptr<R1csDev> dev; std::unique_ptr<Domain> dom; };
zkgpu_r1cs *zkgpu_r1cs_create(size_t n_inputs, size_t n_vars, size_t n_cons, const uint32_t *const rowptr[3], const uint32_t *const col[3],
    const uint8_t *const coeff[3]) {
  zkgpu_r1cs *out = nullptr;
  guarded([&] { std::unique_ptr<zkgpu_r1cs> h(new zkgpu_r1cs); h->host.n_inputs = n_inputs; h->host.n_vars = n_vars; h->host.n_cons = n_cons;
    for (int m = 0; m < 3; m++) {
      size_t nnz = rowptr[m][n_cons];
      h->host.rowptr[m].assign(rowptr[m], rowptr[m] + n_cons + 1);
      h->host.col[m].assign(col[m], col[m] + nnz);
      h->host.coeff[m].resize(nnz);
      if (nnz) memcpy(h->host.coeff[m].data(), coeff[m], 32 * nnz);
    }
    h->dev.reset(new R1csDev(h->host)); h->dom.reset(new Domain(n_cons + n_inputs + 1)); out = h.release(); return ZKGPU_OK; });
  return out;
}
void zkgpu_r1cs_destroy(zkgpu_r1cs *cs) { guarded([&] { delete cs; return ZKGPU_OK; }); }
int zkgpu_witness_map(zkgpu_r1cs *cs, const uint8_t *z, uint8_t *h_out) { return guarded([&] { if (!cs) return ZKGPU_ERR_ARG;
  size_t m = cs->dom->m(), nv = cs->host.n_vars; std::vector<Fe32> zz(nv + 1); memset(&zz[0], 0, 32); zz[0].l[0] = 1; if (nv) memcpy(&zz[1], z, 32 * nv);
  DevBuf<Fe32> zd(nv + 1), abc(3 * m); zd.upload(zz.data(), nv + 1); fr_to_mont_dev(zd.get(), nv + 1);
  cs->dev->eval(zd.get(), abc.get(), m);
  if (!cs->dev->satisfied(abc.get(), m)) {
    g_err = "assignment does not satisfy the constraint system";
    return ZKGPU_ERR_UNSATISFIED;
  }
  // r1cs_to_qap.tcc:239-322 with d1 = d2 = d3 = 0
  cs->dom->ifft(abc.get(), 3, m);
  cs->dom->coset_fft(abc.get(), 3, m);
  cs->dom->qap_pointwise(abc.get(), abc.get() + m, abc.get() + 2 * m);
  cs->dom->icoset_fft(abc.get(), 1, m);
  fr_from_mont_dev(abc.get(), m); abc.download((Fe32 *)h_out, m); memset(h_out + 32 * m, 0, 32); return ZKGPU_OK; }); }
}  // extern "C"
