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

def parse_cpulist(text):
    """'0-63,128-191' (sysfs) -> [0, ..., 63, 128, ..., 191]"""
    out = []
    for part in text.strip().split(","):
        if not part: continue
        a, _, b = part.partition("-"); out.extend(range(int(a), int(b or a) + 1))
    return out

def host_node_cpus():
    """{NUMA node: its CPUs} from sysfs; {} where there is none"""
    import glob, os, re
    nodes = {}
    for d in glob.glob("/sys/devices/system/node/node[0-9]*"):
        try: nodes[int(re.search(r"node(\d+)$", d).group(1))] = parse_cpulist(open(os.path.join(d, "cpulist")).read())
        except Exception: pass
    return nodes

def cpu_busy_fractions(sample_s=0.1):
    """{cpu: busy fraction over a short sample of /proc/stat}; {} where that cannot be read.  The GPU hosts are shared: a block of cores (or their SMT siblings) that another
    tenant is using is a bad place for a rank's helper threads."""
    import time
    def snap():
        out = {}
        for line in open("/proc/stat"):
            if line.startswith("cpu") and line[3].isdigit():
                f = line.split(); v = [int(x) for x in f[1:9]]; out[int(f[0][3:])] = (sum(v), v[3] + v[4])       # total, idle + iowait
        return out
    try:
        a = snap(); time.sleep(sample_s); b = snap()
        return {c: 1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]) for c in b if c in a}
    except Exception:
        return {}

def cpu_siblings():
    """{cpu: [its SMT siblings, itself included]} from sysfs; {} where there is none"""
    import glob, re
    out = {}
    for d in glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology/thread_siblings_list"):
        try: out[int(re.search(r"cpu(\d+)/topology", d).group(1))] = parse_cpulist(open(d).read())
        except Exception: pass
    return out

def quietest_block(cands, block, busy, siblings, prefer=None):
    """Of the aligned blocks of `block` CPUs in the list `cands`, the one whose cores — SMT siblings included — were least busy (pure function); ties and missing data go to
    the block that contains `prefer`, else to the first."""
    blocks = [cands[i:i + block] for i in range(0, len(cands), block)]; blocks = [b for b in blocks if len(b) >= max(1, block // 2)] or [cands]
    def load(b):
        cores = set(b)
        for c in b: cores.update(siblings.get(c, []))
        return round(sum(busy.get(c, 0.0) for c in cores), 2)
    best = min(blocks, key=lambda b: (load(b) if busy else 0.0, 0 if prefer in b else 1))
    return best

def host_cpus_for_rank(local_rank, world, gpu_nodes, node_cpus, allowed, cores, near_cpu=None, block=32, busy=None, siblings=None):
    """The CPUs a rank should run on (pure function; tests/test_distributed_cpu.py).  Two measured facts (profiles/r04w_host_placement.txt, r04y_affinity_sweep.txt): the
    socket matters less than COMPACTNESS — one and the same build takes 0.98-0.99 ms per proof with its twenty helper threads free to roam a 256-thread host (a helper that
    is woken lands on a cold, deeply idle core) and 0.90-0.93 ms confined to 16-32 neighbouring cores of either socket —, and ranks must not share cores.
      world == 1 -> a block of `block` neighbouring allowed CPUs on the GPU's node: the one whose cores and SMT siblings were least busy in a short sample (`busy`,
                    `siblings`: the hosts are shared), the block that contains `near_cpu` (where the process runs now) among equals;
      world  > 1 -> the ranks of one socket cut that socket's allowed CPUs into slices of cores // world.
    gpu_nodes[i]: NUMA node of local GPU i (-1 unknown); node_cpus: {node: [cpus]}; allowed: the process's affinity mask; cores: what the whole job may really use (cgroup
    quota).  Unknown node, or a node without an allowed CPU: the same rules over all allowed CPUs."""
    allowed = sorted(allowed); node = gpu_nodes[local_rank % len(gpu_nodes)] if gpu_nodes else -1
    mine = [c for c in node_cpus.get(node, []) if c in set(allowed)] if node is not None and node >= 0 else []
    if world <= 1:
        cand = mine or allowed
        if len(cand) <= block: return cand
        return quietest_block(cand, block, busy or {}, siblings or {}, prefer=near_cpu)
    per = max(1, cores // world)
    if not mine:
        cut = allowed[:cores][local_rank % world * per:(local_rank % world + 1) * per]; return cut or allowed
    peers = [r for r in range(world) if gpu_nodes[r % len(gpu_nodes)] == node]; k = peers.index(local_rank % world) if local_rank % world in peers else 0
    per = max(1, min(per, len(mine) // max(1, len(peers)))); cut = mine[k * per:(k + 1) * per]
    return cut or mine

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


class Group:
    """The ranks of one node as bench.py and the sharded prover use them: join, barrier, agree on success, hand one object from rank 0 to everybody, reduce the timing,
    gather partial records.  backend "nccl" is RCCL over xGMI (one rank per GPU, collectives on device tensors of that rank's GPU); "gloo" runs the same code path on a box
    with fewer GPUs than ranks (tests).  Every torch.distributed call of the N > 1 path is made HERE, so that a CPU test can replay them against the real signatures
    (tests/test_distributed_cpu.py::test_group_calls_bind_to_torch_signatures) before the first run on real multi-GPU hardware."""
    def __init__(self, backend, rank, world, local_rank, timeout_s=600):
        import datetime, torch, torch.distributed as dist
        self.dist, self.torch, self.backend, self.rank, self.world, self.local_rank = dist, torch, backend, rank, world, local_rank
        self.host_group = None
        if backend == "nccl":
            self.device = torch.device("cuda", local_rank); torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s), device_id=self.device)
            # the partial records of a sharded proof are 384 bytes a rank and ALREADY in host memory (the MSM kernels write their results into pinned host buffers): they
            # are gathered over a second, host-side group (gloo) — no copy to the device and back around a collective of 384 bytes.  RCCL keeps the barrier and the reductions.
            self.host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=timeout_s))
        else:
            self.device = torch.device("cpu"); dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    def _tensor(self, values, dtype):
        return self.torch.tensor(values, dtype=dtype, device=self.device)
    def barrier(self):
        self.dist.barrier()
    def all_ok(self, ok):
        """True iff every rank passed True: a rank whose set-up failed takes the whole group down instead of leaving the others at a barrier"""
        t = self._tensor([1 if ok else 0], self.torch.int32); self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN); return bool(int(t.item()))
    def share_from_rank0(self, obj):
        box = [obj if self.rank == 0 else None]; self.dist.broadcast_object_list(box, src=0); return box[0]
    def aggregate_throughput(self, units_per_rank, seconds):
        t = self._tensor([seconds], self.torch.float64); self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        u = self._tensor([units_per_rank], self.torch.float64); self.dist.all_reduce(u, op=self.dist.ReduceOp.SUM)
        return float(u.item()) / float(t.item()), float(t.item())
    def gather_partials(self, partial_bytes):
        """one fixed-size record per rank, in rank order, on every rank: host tensors over the host group (nccl runs) or over the only group there is (gloo runs)"""
        mine = self.torch.frombuffer(bytearray(partial_bytes), dtype=self.torch.uint8)
        out = [self.torch.empty_like(mine) for _ in range(self.world)]; self.dist.all_gather(out, mine, group=self.host_group)
        return [bytes(t.numpy().tobytes()) for t in out]
    def describe(self):
        """What an N > 1 run should say about itself before it reports a rate (no multi-GPU hardware was available while this was written: the first real run is to be
        read with this in hand).  After the first barrier, on every rank: the device it computes on (index, name, PCI bus id, total memory), the communicator's size as the
        backend reports it, and the round trip of a barrier and of one gather of 384-byte records (the sharded prover's only exchange), each timed over ten calls.
        Returns the list of all ranks' records in rank order (gathered as objects)."""
        import time
        self.barrier(); torch = self.torch; rec = {"rank": self.rank, "local_rank": self.local_rank, "backend": self.backend, "world_size_reported": self.dist.get_world_size(), "host": __import__("socket").gethostname()}
        if self.backend == "nccl":
            n_vis = torch.cuda.device_count()
            if n_vis < self.world: raise RuntimeError("nccl run with %d ranks on a node that shows %d devices: one rank per GPU is the contract" % (self.world, n_vis))
            pr = torch.cuda.get_device_properties(self.local_rank); rec.update({"device": self.local_rank, "name": pr.name, "total_memory_gb": round(pr.total_memory / 1e9, 1), "devices_visible": n_vis})
            try: rec["pci_bus_id"] = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            except Exception: rec["pci_bus_id"] = None
        def timed(fn, n=10):
            fn(); t0 = time.perf_counter()
            for _ in range(n): fn()
            if self.backend == "nccl": torch.cuda.synchronize()
            return round(1e6 * (time.perf_counter() - t0) / n, 1)
        rec["barrier_us"] = timed(self.barrier); rec["gather_384B_us"] = timed(lambda: self.gather_partials(bytes(384)))
        box = [None] * self.world; self.dist.all_gather_object(box, rec); return box
    def close(self):
        self.dist.destroy_process_group()
