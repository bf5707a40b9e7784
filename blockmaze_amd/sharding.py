"""How the proving path is spread over the GPUs of one node (one process per GPU, `torch.distributed`).

Two independent axes:
  * proofs are independent units -> rank k proves instances k, k + N, k + 2N, ...; no data-path collective
    (`instances_for_rank`, `aggregate_throughput`);
  * one multi-exponentiation can be cut by contiguous point ranges -> rank k owns `msm_range(n, k, N)` of every query and
    the only exchange is an all-gather of the partial sums (5 points per rank), added up locally (`gather_partials`):
    RCCL has no elliptic-curve reduction operator, so the collective moves limbs and the group law runs after it.
Only plumbing lives here; the arithmetic is in libzkgpu.so.
"""
import numpy as np

def instances_for_rank(n_instances, rank, world):
    """round-robin assignment of independent proof instances"""
    return list(range(rank, n_instances, world))

def msm_range(n, rank, world):
    """contiguous slice [begin, end) of an n-point query owned by `rank`; slices tile [0, n) exactly"""
    base, rem = divmod(n, world); begin = rank * base + min(rank, rem); return begin, begin + base + (1 if rank < rem else 0)

def aggregate_throughput(units_per_rank, seconds, dist=None, device=None):
    """whole-job rate: all ranks' units over the slowest rank's time (max-reduce over the process group)"""
    if dist is not None:
        import torch
        t = torch.tensor([seconds], dtype=torch.float64, device=device); dist.all_reduce(t, op=dist.ReduceOp.MAX); seconds = float(t.item())
        u = torch.tensor([units_per_rank], dtype=torch.float64, device=device); dist.all_reduce(u, op=dist.ReduceOp.SUM); units = float(u.item())
    else:
        units = units_per_rank
    return units / seconds, seconds

def gather_partials(partial_bytes, dist, device=None):
    """all-gather one fixed-size byte record per rank (the partial MSM results); returns the list of records in rank order"""
    import torch
    mine = torch.frombuffer(bytearray(partial_bytes), dtype=torch.uint8).to(device) if device is not None else torch.frombuffer(bytearray(partial_bytes), dtype=torch.uint8)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]; dist.all_gather(out, mine)
    return [bytes(t.cpu().numpy().tobytes()) for t in out]
