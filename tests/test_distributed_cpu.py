"""The N > 1 plumbing on CPU: two processes over gloo (127.0.0.1) exercise the instance partition, the max-over-ranks
timing reduce and the all-gather of partial MSM records used when one multi-exponentiation is cut across GPUs."""
import os, socket, sys
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from blockmaze_amd import sharding

def test_partitions_tile_exactly():
    for n in (0, 1, 7, 64, 262143, 1179647):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world): seen += sharding.instances_for_rank(n if n < 100 else 64, r, world)
            assert sorted(seen) == list(range(n if n < 100 else 64))
            ranges = [sharding.msm_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and max(e - b for b, e in ranges) - min(e - b for b, e in ranges) <= 1

def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rate, slowest = sharding.aggregate_throughput(10, 1.0 + rank, dist)                       # rank r "took" 1 + r seconds for 10 units
    recs = sharding.gather_partials(bytes([rank + 1]) * 576, dist)                             # 4 G1 + 1 G2 partials = 576 B per rank
    q.put((rank, rate, slowest, [r[0] for r in recs], sharding.instances_for_rank(5, rank, world)))
    dist.barrier(); dist.destroy_process_group()

def test_two_ranks_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue(); procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs: p.join(timeout=60); assert p.exitcode == 0
    for rank, rate, slowest, firsts, inst in res:
        assert slowest == 2.0 and rate == 20 / 2.0 and firsts == [1, 2] and inst == ([0, 2, 4] if rank == 0 else [1, 3])

def test_group_calls_bind_to_torch_signatures(monkeypatch):
    """RCCL has never run under this repo (no multi-GPU box in its rounds).  Every torch.distributed call of the N > 1 path lives in sharding.Group; here the nccl branch is
    replayed against the REAL signatures of this torch build (inspect.signature(...).bind: a wrong keyword, a missing argument or a wrong argument order raises), with the
    collectives' effect faked for a world of 2 and device tensors created on the CPU instead — so the first run on 8 GPUs cannot die of an argument error."""
    import datetime, inspect
    calls = []
    def bound(real, impl):
        sig = inspect.signature(real)
        def f(*a, **k): ba = sig.bind(*a, **k); ba.apply_defaults(); calls.append((real.__name__, dict(ba.arguments))); return impl(ba.arguments)
        return f
    def fake_all_reduce(a): assert isinstance(a["tensor"], torch.Tensor) and isinstance(a["op"], dist.ReduceOp.RedOpType if hasattr(dist.ReduceOp, "RedOpType") else object)
    def fake_all_gather(a):
        assert all(t.shape == a["tensor"].shape and t.dtype == a["tensor"].dtype for t in a["tensor_list"])
        for i, t in enumerate(a["tensor_list"]): t.copy_(a["tensor"] + i)
    def fake_bcast(a): assert a["src"] == 0 and isinstance(a["object_list"], list) and len(a["object_list"]) == 1
    monkeypatch.setattr(dist, "init_process_group", bound(dist.init_process_group, lambda a: None))
    monkeypatch.setattr(dist, "all_reduce", bound(dist.all_reduce, fake_all_reduce)); monkeypatch.setattr(dist, "all_gather", bound(dist.all_gather, fake_all_gather))
    monkeypatch.setattr(dist, "barrier", bound(dist.barrier, lambda a: None)); monkeypatch.setattr(dist, "broadcast_object_list", bound(dist.broadcast_object_list, fake_bcast))
    monkeypatch.setattr(dist, "destroy_process_group", bound(dist.destroy_process_group, lambda a: None))
    HOST = object(); monkeypatch.setattr(dist, "new_group", bound(dist.new_group, lambda a: HOST))                  # (the host-side gloo group of an nccl run: the 384-byte records)
    set_dev = []; monkeypatch.setattr(torch.cuda, "set_device", lambda d: set_dev.append(d))
    wanted = []
    class CpuTensors(sharding.Group):                                                   # (no GPU here: the tensors a rank would create on its device are made on the CPU)
        def _tensor(self, values, dtype): wanted.append(self.device); return torch.tensor(values, dtype=dtype)
    g = CpuTensors("nccl", 0, 2, 0, timeout_s=77)
    init = calls[0][1]; assert calls[0][0] == "init_process_group" and init["backend"] == "nccl" and init["rank"] == 0 and init["world_size"] == 2
    assert init["device_id"] == torch.device("cuda", 0) and init["timeout"] == datetime.timedelta(seconds=77) and set_dev == [0] and g.device == torch.device("cuda", 0)
    g.barrier(); assert g.all_ok(True) is True and g.all_ok(False) is False
    assert g.share_from_rank0({"key_dir": "/tmp/x"}) == {"key_dir": "/tmp/x"}
    rate, slowest = g.aggregate_throughput(20, 2.0); assert slowest == 2.0 and rate == 10.0
    ng = [c[1] for c in calls if c[0] == "new_group"]; assert len(ng) == 1 and ng[0]["backend"] == "gloo" and ng[0]["timeout"] == datetime.timedelta(seconds=77) and g.host_group is HOST
    recs = g.gather_partials(bytes(range(1, 97)) * 4); assert len(recs) == 2 and len(recs[0]) == 384 and recs[0][0] == 1 and recs[1][0] == 2
    ag = [c[1] for c in calls if c[0] == "all_gather"][-1]; assert ag["group"] is HOST and ag["tensor"].device.type == "cpu"           # host tensors over the host group: nothing goes to the device and back
    g.close()
    names = [c[0] for c in calls]; assert names.count("all_reduce") == 4 and "all_gather" in names and "broadcast_object_list" in names and names[-1] == "destroy_process_group"
    assert all(d == torch.device("cuda", 0) for d in wanted)                             # every collective tensor of the nccl branch is asked for on this rank's GPU
    ops = [c[1]["op"] for c in calls if c[0] == "all_reduce"]; assert ops == [dist.ReduceOp.MIN, dist.ReduceOp.MIN, dist.ReduceOp.MAX, dist.ReduceOp.SUM]

def _group_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    g = sharding.Group("gloo", rank, world, rank, timeout_s=60)
    ok_all = g.all_ok(True); ok_one = g.all_ok(rank != 1); shared = g.share_from_rank0("dir-of-rank-0" if rank == 0 else None)
    rate, slowest = g.aggregate_throughput(10, 1.0 + rank); recs = g.gather_partials(bytes([rank + 1]) * 384); info = g.describe()
    q.put((rank, ok_all, ok_one, shared, rate, slowest, [r[0] for r in recs], [(r["rank"], r["world_size_reported"], r["backend"], r["barrier_us"] > 0, r["gather_384B_us"] > 0) for r in info])); g.barrier(); g.close()

def test_group_two_ranks_gloo():
    """the same Group object over a real (gloo) process group of two ranks: agreement on success, the hand-over from rank 0, the timing reduce, the record gather and the
    run's description of itself (Group.describe)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue(); procs = [ctx.Process(target=_group_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs: p.join(timeout=60); assert p.exitcode == 0
    for rank, ok_all, ok_one, shared, rate, slowest, firsts, info in res:
        assert ok_all is True and ok_one is False and shared == "dir-of-rank-0" and slowest == 2.0 and rate == 10.0 and firsts == [1, 2]
        assert info == [(0, 2, "gloo", True, True), (1, 2, "gloo", True, True)]          # Group.describe(): what an N > 1 run says about itself, on every rank, in rank order

def test_ranks_are_placed_on_their_gpus_socket():
    """sharding.host_cpus_for_rank: a rank runs on the CPUs of the NUMA node its GPU hangs off (bench.py binds itself before it allocates its assignments)"""
    from blockmaze_amd import sharding as s
    assert s.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and s.parse_cpulist("5") == [5] and s.parse_cpulist("") == []
    nodes = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}; everything = set(range(256)); gpus = [0, 0, 0, 0, 1, 1, 1, 1]
    assert s.host_cpus_for_rank(0, 1, [0], nodes, everything, 16) == list(range(0, 32))                      # one rank: a block of 32 neighbouring cores on its GPU's socket
    assert s.host_cpus_for_rank(0, 1, [1], nodes, everything, 16, near_cpu=100) == list(range(96, 128))      # ... the block it is running on, if that is on the socket
    assert s.host_cpus_for_rank(0, 1, [1], nodes, everything, 16, near_cpu=3) == list(range(64, 96))         # (running on the other socket: the socket's first block)
    assert s.host_cpus_for_rank(0, 1, [0], nodes, everything, 16, near_cpu=190) == list(range(160, 192))
    assert s.host_cpus_for_rank(0, 1, [-1], nodes, everything, 16, near_cpu=70) == list(range(64, 96))       # unknown node: a block of the allowed CPUs all the same
    assert s.host_cpus_for_rank(0, 1, [0], nodes, set(range(8)), 16) == list(range(8))                       # a small cpuset: all of it
    sib = {c: [c % 128, c % 128 + 128] for c in range(256)}; busy = {c: 0.0 for c in range(256)}; busy[130] = 0.9; busy[40] = 0.5   # a neighbour on the SMT sibling of core 2, another on core 40
    assert s.host_cpus_for_rank(0, 1, [0], nodes, everything, 16, near_cpu=5, busy=busy, siblings=sib) == list(range(32, 64))     # blocks 0-31 and 128-159 share cores with 130 (0.9); 32-63 and 160-191 with 40 (0.5)
    busy[40] = 0.95; assert s.host_cpus_for_rank(0, 1, [0], nodes, everything, 16, near_cpu=150, busy=busy, siblings=sib) == list(range(128, 160))   # now the first pair is the quieter one: the half it runs on
    assert s.host_cpus_for_rank(0, 1, [0], nodes, everything, 16, near_cpu=5, busy={c: 0.0 for c in range(256)}, siblings=sib) == list(range(0, 32))   # all quiet: where it runs
    assert s.quietest_block(list(range(64)), 32, {}, {}, prefer=40) == list(range(32, 64))
    assert isinstance(s.cpu_busy_fractions(0.01), dict) and isinstance(s.cpu_siblings(), dict)
    cuts = [s.host_cpus_for_rank(r, 8, gpus, nodes, everything, 16) for r in range(8)]
    assert all(len(c) == 2 for c in cuts) and len(set(sum(cuts, []))) == 16                                   # 16 usable cores over 8 ranks, no CPU twice
    assert all(set(cuts[r]) <= set(nodes[gpus[r]]) for r in range(8))                                         # each on its own GPU's socket
    assert s.host_cpus_for_rank(5, 8, gpus, nodes, set(range(0, 64)), 16) == [10, 11]                         # a cpuset without that socket: the old slices of the allowed CPUs
    assert s.host_cpus_for_rank(1, 2, [0], nodes, everything, 64) == list(range(32, 64))                      # two ranks sharing one GPU (gloo test runs): disjoint halves
    assert isinstance(s.host_node_cpus(), dict)
