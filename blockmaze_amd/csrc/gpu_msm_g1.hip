// G1 instantiation of the MSM pipeline (kept in its own translation unit: the inlined field arithmetic makes these kernels slow to compile)
#include "msm_impl.hpp"
namespace zk {
struct MsmG1::Impl : MsmImpl<Fq, G1AffineRaw> { using MsmImpl::MsmImpl; };
MsmG1::MsmG1(const G1AffineRaw *p, size_t n, int c, bool fo, bool tables, bool uniform) : impl(new Impl(p, n, c, fo, tables, uniform)) {}
MsmG1::MsmG1(const MsmG1 &peer, bool fo, bool uniform) : impl(new Impl(peer.impl->bases, fo, uniform)) {}
MsmG1::~MsmG1() = default;
std::shared_ptr<WsortBuffers> MsmG1::sort_handle() const { return impl->wfused && impl->ws_leader ? impl->ws : nullptr; }
bool MsmG1::share_sort_with(const std::shared_ptr<WsortBuffers> &leader) {
  if (!impl->wfused || !leader || leader->NB != impl->NB || leader->n != impl->n) return false;
  impl->share_sort(leader);
  return true;
}
void MsmG1::run(const Fe32 *s, const uint32_t *idx) { impl->run(s, idx); }
void MsmG1::run_tagged(const Fe32 *z_all, const WitnessTags &wt, const uint32_t *idx) { impl->run_tagged(z_all, wt, idx); }
bool MsmG1::one_pass_sort() const { return impl->hsort; }
void MsmG1::run_product(const Fe32 *a, const Fe32 *b, const Fe32 *z, bool z_is_table) { impl->run_product(a, b, z, z_is_table); }
void MsmG1::set_label(const char *l) { impl->label = l; }
void MsmG1::set_crowded(bool c) { impl->crowded = c; }
void MsmG1::set_stream(int aux) { impl->stream_id = aux; }
void MsmG1::split_ones_path() { impl->enable_split_ones(); }
host::HG1 MsmG1::result() { impl->finish_sync(); return combine<host::HFq, Fq>(impl->host_sums(), impl->RS, impl->bitsum ? 1 : impl->c); }
size_t MsmG1::size() const { return impl->n; }
const G1AffineRaw *MsmG1::points_dev() const { return impl->points.get(); }
}  // namespace zk
