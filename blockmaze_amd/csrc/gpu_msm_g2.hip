// G2 (Fq2) instantiation of the MSM pipeline
#include "msm_impl.hpp"
namespace zk {
struct MsmG2::Impl : MsmImpl<Fq2, G2AffineRaw> { using MsmImpl::MsmImpl; };
void MsmG2::set_label(const char *l) { impl->label = l; }
void MsmG2::set_stream(int aux) { impl->stream_id = aux; }
void MsmG2::split_ones_path() { impl->enable_split_ones(); }
MsmG2::MsmG2(const G2AffineRaw *p, size_t n, int c, bool fo, bool tables, bool uniform) : impl(new Impl(p, n, c, fo, tables, uniform)) {}
MsmG2::MsmG2(const MsmG2 &peer, bool fo, bool uniform) : impl(new Impl(peer.impl->bases, fo, uniform)) {}
MsmG2::~MsmG2() = default;
std::shared_ptr<WsortBuffers> MsmG2::sort_handle() const { return impl->wfused && impl->ws_leader ? impl->ws : nullptr; }
bool MsmG2::share_sort_with(const std::shared_ptr<WsortBuffers> &leader) {
  if (!impl->wfused || !leader || leader->NB != impl->NB || leader->n != impl->n) return false;
  impl->share_sort(leader);
  return true;
}
void MsmG2::run(const Fe32 *s, const uint32_t *idx) { impl->run(s, idx); }
void MsmG2::run_tagged(const Fe32 *z_all, const WitnessTags &wt, const uint32_t *idx) { impl->run_tagged(z_all, wt, idx); }
host::HG2 MsmG2::result() { impl->finish_sync(); return combine<host::HFq2, Fq2>(impl->host_sums(), impl->RS, impl->bitsum ? 1 : impl->c); }
}  // namespace zk
