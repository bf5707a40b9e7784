#!/usr/bin/env python3
"""Benchmark of the hot path: Groth16 proofs/sec for BlockMaze's send circuit on MI355X.

A step = one send proof per rank through libzkgpu.so's prover — the equivalent of one `r1cs_gg_ppzksnark_prover(pk, primary, auxiliary)` call (reference
r1cs_gg_ppzksnark.tcc:391-506) — on the next of the run's distinct statements, all of them RESIDENT IN HBM as raw assignments when the timed region starts
(`zkgpu_prover_prove_stashed`: everything the prover derives from an assignment, the classification of multiexp.tcc:443-496 included, happens inside the timed call), and the
serialized proof comes back.  The same call handed a fresh HOST buffer every step is timed right after it with the same steps, barriers and percentiles
(`value_from_host_buffers`).  N ranks prove independent seeded instances (proofs are independent units: no data-path collective), so value = N*K proofs / max-over-ranks wall
time ("weak" scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
        N > 1 without a launcher: this script starts the N ranks itself (child processes, one per GPU, created before
        anything in the parent touches the GPU) and exits with their status.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
        (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; --gpus must equal WORLD_SIZE)

Prints ONE JSON line on rank 0's stdout (everything else, including the reference-style chatter of the cgo symbols,
goes to stderr).  `roofline` times the dominant kernel (bucket accumulation of the H-query MSM) with HIP events on the
library's compute stream; `cpu_baseline` times the reference's own prover (oracle/_ref, kind "reference") on the host
cores, or the plain-C oracle (kind "port") where that binary is absent.
"""
import argparse, json, os, socket, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

H_PAIRS = 262143                       # H-query size of the send circuit (m - 1, m = 2^18)
BYTES_PER_G1_PAIR = 96                 # 64 B affine point + 32 B scalar, each read once
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s
MAX_DISTINCT_WITNESSES = 64            # host memory bound (7.3 MB each); longer runs cycle through them, consecutive steps still differ

def log(*a):
    print(*a, file=sys.stderr, flush=True)

def parse_args():
    ap = argparse.ArgumentParser(); ap.add_argument("--gpus", type=int, default=1); ap.add_argument("--steps", type=int, default=50); ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="only the timed region and the roofline leg (what the N > 1 scaling runs need)")
    ap.add_argument("--inflight", type=int, default=6, help="extra leg (not `value`): this many prover objects per GPU, one host thread each, proofs overlapping on the device; 0/1 = skip")
    ap.add_argument("--batch", type=int, default=64, help="extra leg (not `value`): zkgpu_prover_prove_batch with this many witnesses per call (BASELINE.json configs[2]); 0/1 = skip")
    ap.add_argument("--shard-msm", action="store_true", help="N > 1 only: all ranks prove ONE proof per step together, each holding 1/N of every query; one all-gather of 384-byte partial records per proof (strong scaling)")
    return ap.parse_args()

def launch(args):
    """--gpus N without a launcher: start the N ranks as child processes.  Nothing here imports torch or touches HIP, so no process that initialised the GPU is ever
    replaced or forked; a rank that fails takes the whole run down with a non-zero status."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ZK_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0; pending = set(range(args.gpus))
    while pending:
        for r in list(pending):
            c = procs[r].poll()
            if c is None: continue
            pending.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1; log("bench: rank %d exited with status %d, stopping the other ranks" % (r, c))
                for q in pending: procs[q].terminate()                       # exact child PIDs only
        time.sleep(0.05)
    return rc

def read_witness(path):
    """[u64 n | n * 32 bytes] as written by libzkgpu's witness generators -> (n, 4) uint64 canonical values"""
    import numpy as np
    b = open(path, "rb").read(); n = int(np.frombuffer(b, dtype=np.uint64, count=1)[0]); return np.frombuffer(b, dtype=np.uint64, count=4 * n, offset=8).reshape(n, 4).copy()

def usable_cores():
    """host cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a pod that sees 256 CPUs may own far fewer; oversubscribed OpenMP
    threads make the multi-core baseline slower than one thread)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max": n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0: n = max(1, min(n, q // per))
        except Exception: pass
    return min(n, 64)

def harness_kv(harness, *cmd, env=None, timeout=None):
    out = subprocess.run([harness, *cmd], capture_output=True, text=True, env=env, timeout=timeout).stdout; kv = {}
    for line in out.splitlines():
        tok = line.split()
        for a, b in zip(tok[0::2], tok[1::2]): kv[a] = b
    return kv

def run_rank(args):
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log("bench: --gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world)); return 2
    # the cgo symbols print the reference's progress lines on stdout (sendcgo.cpp:352,458): keep fd 1 for the JSON line only
    sys.stdout.flush(); real_stdout = os.fdopen(os.dup(1), "w"); os.dup2(2, 1)
    os.environ.setdefault("ZK_DEVICE", str(local_rank))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # torch initialises the HIP runtime before libzkgpu.so is loaded, so the library's own load-time default would come too late here
    # N = 1: the process is confined to a block of 32 neighbouring cores BEFORE the HIP runtime and torch start their threads (see the comment at `host_binding` below; for
    # N > 1 the ranks are placed after the devices are known).  The block is the quietest one of the socket this process is running on (a 0.1 s sample of /proc/stat, SMT siblings included: the hosts are shared): which socket hardly matters, compactness does.
    host_binding = "none"; bind = os.environ.get("ZK_BENCH_BIND", "1") != "0" and hasattr(os, "sched_setaffinity")
    if bind and world == 1:
        try:
            from blockmaze_amd import sharding as placement
            here = int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36]); nodes = placement.host_node_cpus()      # the CPU this thread last ran on
            node = next((k for k, v in nodes.items() if here in v), -1)
            mine = placement.host_cpus_for_rank(0, 1, [node], nodes, os.sched_getaffinity(0), usable_cores(), near_cpu=here, busy=placement.cpu_busy_fractions(0.1), siblings=placement.cpu_siblings(), block=int(os.environ.get("ZK_BENCH_BLOCK", "32")))
            if mine: os.sched_setaffinity(0, mine); host_binding = "rank confined to CPUs %d..%d (%d, NUMA node %d) before the runtime starts" % (mine[0], mine[-1], len(mine), node)
        except Exception as ex: log("bench: the kernel's placement stays (%s)" % ex)
    import torch
    backend = os.environ.get("ZK_BENCH_BACKEND", "nccl")                 # "gloo": lets the N > 1 code path run on a box with fewer GPUs than ranks (ranks share devices)
    grp = None; rank_info = None
    if world > 1:
        from blockmaze_amd import sharding
        grp = sharding.Group(backend, rank, world, local_rank, timeout_s=int(os.environ.get("ZK_BENCH_GROUP_TIMEOUT_S", "600")))   # every torch.distributed call of this script is made there
        log("bench: rank %d of %d joined the process group (backend %s)" % (grp.dist.get_rank(), grp.dist.get_world_size(), backend))
        assert grp.dist.get_world_size() == args.gpus
        # an N > 1 run describes itself (device per rank, PCI bus id, the communicator's size, barrier and gather round trips) before it measures anything: stderr
        # on rank 0, and `ranks` in the JSON line
        rank_info = grp.describe()
        if rank == 0:
            for r_ in rank_info: log("bench: rank %s" % json.dumps(r_))
            if len({(r_.get("host"), r_.get("pci_bus_id")) for r_ in rank_info}) < len(rank_info) and backend == "nccl": log("bench: WARNING: two ranks report the same device — the rates below are not a scaling measurement")
        if backend != "nccl" and torch.cuda.is_available(): torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count())); os.environ["ZK_DEVICE"] = str(local_rank % max(1, torch.cuda.device_count()))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    # host budget of a rank: with N ranks on one node every rank gets 1/N of the usable cores, is pinned to them, and sizes the prover's helper threads accordingly
    # (4 submit threads + a 3-thread witness pool per prover by default: 8 ranks x 8 threads on a 16-core pod would oversubscribe the host inside the timed region)
    cores = usable_cores(); per_rank = max(1, cores // world)
    if world > 1:
        if per_rank < 4: os.environ.setdefault("ZK_SUBMIT_THREADS", "0")        # too few cores for helper threads: this rank's one thread submits everything itself
        os.environ.setdefault("ZK_WITNESS_THREADS", str(max(0, min(3, per_rank - 1))))
    from blockmaze_amd import engine as e
    import workload as w
    hx = lambda a: [("0x" + x.hex()) if isinstance(x, bytes) else x for x in a]
    e.init()                                                                 # raises "no HIP device visible" on a box without a GPU: there is no CPU path to fall back to
    # A rank lives on a COMPACT set of CPUs near its GPU, chosen before it allocates a single assignment: a block of 32 neighbouring cores for N = 1, disjoint slices of the
    # GPU's socket for N > 1 (the old placement cut the first `cores` CPUs, all on socket 0, over the ranks of both sockets' GPUs).  On the 256-thread GPU hosts the same build
    # takes 0.98-0.99 ms per proof with its helper threads free to roam (a helper that is woken lands on a cold core) and 0.90-0.93 ms on 16-32 neighbouring cores
    # (profiles/r04y_affinity_sweep.txt); which socket hardly matters.  What any launcher does with numactl / taskset; ZK_BENCH_BIND=0 leaves the placement to the kernel.
    if bind and world > 1:
        try:
            from blockmaze_amd import sharding as placement
            n_dev = max(1, torch.cuda.device_count()); gpu_nodes = [e.device_numa_node(i) for i in range(n_dev)]
            here = int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36])              # the CPU this thread last ran on
            mine = placement.host_cpus_for_rank(local_rank, world, gpu_nodes, placement.host_node_cpus(), os.sched_getaffinity(0), cores, near_cpu=here)
            if mine:
                os.sched_setaffinity(0, mine); host_binding = "GPU on NUMA node %d: rank bound to CPUs %d..%d (%d)" % (gpu_nodes[local_rank % n_dev], mine[0], mine[-1], len(mine))
        except Exception as ex: log("bench: rank %d keeps the kernel's placement (%s)" % (rank, ex))

    # ---- untimed setup: test keys for the send circuit (seeded toxic waste), resident prover, one witness per step -----------------
    # ONE key for all ranks of the node: rank 0 makes a private directory (mkdtemp: no predictable path that a stale or foreign key could sit under), generates the key and,
    # by loading it first, leaves the fast container beside it; the path travels to the other ranks, which load that.  Every step of the set-up ends with the ranks agreeing
    # that all of them got through it: a rank that fails takes the group down with a non-zero status instead of leaving the others at a barrier.
    fail_at = os.environ.get("ZK_BENCH_TEST_FAIL", "")                    # tests only: "<stage>:<rank>" makes that rank raise there (stages: keygen, load)
    def stage(name, fn):
        err = None; out = None
        try:
            if fail_at == "%s:%d" % (name, rank): raise RuntimeError("ZK_BENCH_TEST_FAIL asked rank %d to fail at '%s'" % (rank, name))
            out = fn()
        except Exception as ex:
            err = ex; log("bench: rank %d failed at '%s': %s" % (rank, name, ex))
        ok = err is None if grp is None else grp.all_ok(err is None)
        if not ok:
            if grp is not None: log("bench: rank %d leaves: a rank failed at '%s'" % (rank, name)); grp.close()
            raise SystemExit(3)
        return out
    tmp = tempfile.mkdtemp(prefix="zkbench_%d_" % rank); shard = args.shard_msm and world > 1; t_keygen = 0.0
    def make_key():
        kd = tmp if world == 1 else tempfile.mkdtemp(prefix="zkbench_key_"); t0 = time.time(); e.keygen("send", os.path.join(kd, "sendpk.txt"), os.path.join(kd, "sendvk.txt"), seed=0xB10C4A2E); return kd, time.time() - t0
    made = stage("keygen", make_key if rank == 0 else (lambda: None))
    if rank == 0: key_dir, t_keygen = made
    if grp is not None: key_dir = grp.share_from_rank0(key_dir if rank == 0 else None)
    pk_path, vk_path = os.path.join(key_dir, "sendpk.txt"), os.path.join(key_dir, "sendvk.txt")
    hbm = {}
    def load():
        free0 = torch.cuda.mem_get_info()[0] if torch.cuda.is_available() else None
        t0 = time.time(); p = e.Prover(pk_path, rank, world) if shard else e.Prover(pk_path); dt_ = time.time() - t0
        if free0 is not None: hbm["key_and_first_prover_gb"] = round((free0 - torch.cuda.mem_get_info()[0]) / 1e9, 3)      # tables (one coordinate form since round 5) + twiddles + constraint system + one prover's workspaces
        return p, dt_
    first = stage("load", load if rank == 0 else (lambda: None))             # rank 0 first: its load from text leaves the container the others map
    rest = stage("load", load if rank != 0 else (lambda: None))
    prover, t_load = first if rank == 0 else rest
    n_inst = max(2, min(args.steps + args.warmup, MAX_DISTINCT_WITNESSES)); insts, zs = [], []; wp = os.path.join(tmp, "w.bin")
    for i in range(n_inst):
        d = w.send_instance(i if shard else rank + i * world); e.witness_send(*hx(w.send_args(d)), wp); insts.append(d); zs.append(read_witness(wp))
    e.witness_send(*hx(w.send_args(insts[0])), os.path.join(tmp, "w0.bin"))            # kept on disk for the CPU baseline

    def barrier():
        if grp is not None: grp.barrier()
        if torch.cuda.is_available(): torch.cuda.synchronize()

    remade = []                                                               # statements whose stash had to be made a second time (stash_failed below): none, normally
    if shard:
        def one_proof(i):                                                      # every rank runs the device pipeline on its slice; 384 B per rank are exchanged; rank 0 assembles
            prover.set_witness(zs[i % n_inst]); recs = grp.gather_partials(prover.prove_partial())
            return prover.finish(recs, 0x1234567 + i, 0x7654321 + i) if rank == 0 else None
    else:
        # host buffer in, fresh (r, s), serialized proof (512 hex characters) out; synchronous.  The C entry point is called with argument objects built once: what is timed
        # is the library, not numpy's contiguity checks and a Python string per step (20-25 us of a 1.1 ms step); the buffer of the last call is decoded for the verification below
        import ctypes
        _lib = e.lib(); _h = ctypes.c_void_p(prover.h); _zp = [z.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)) for z in zs]; _out = ctypes.create_string_buffer(513)
        def host_buffer_proof(i):
            rc = _lib.zkgpu_prover_prove(_h, _zp[i % n_inst], None, None, _out)
            if rc != 0: raise RuntimeError("zkgpu_prover_prove failed: %s" % _lib.zkgpu_last_error().decode())
            return _out
        # INPUTS RESIDENT IN HBM when the timed region starts (the contract's `value`): every distinct statement of the run is handed over once, before the clock starts, and
        # kept in device memory AS THE RAW ASSIGNMENT (zkgpu_prover_stash_witness: (n + 1) x 32 B, 7.3 MB each, nothing derived from it); a step proves the next one
        # (zkgpu_prover_prove_stashed): the tags and the list of values other than 0 / 1 — the classification of libsnark's multi_exp_with_mixed_addition — are derived by a
        # device kernel inside the timed call, then the whole pipeline runs on the stash in place.  The host-buffer-inclusive rate is `value_from_host_buffers`.
        slots = []
        for z in zs: prover.set_witness(z); slots.append(ctypes.c_uint32(prover.stash_witness()))
        import numpy as np
        def stash_failed(i):
            # (seen once, on one box, in three runs out of three, never again: a stashed statement whose host copy proves.  The step then says what it can about the
            # slot — which constraint, how the resident vector differs from the one handed over, whether the host buffer proves — makes the stash again ONCE beside the
            # old one and goes on; the line carries the count in config.stash_remade.  A statement that fails from its host buffer too ends the run.)
            k = i % n_inst; msg = _lib.zkgpu_last_error().decode(); back = prover.read_stash(slots[k].value); diff = np.nonzero((back != zs[k]).any(axis=1))[0]
            log("bench: prove_stashed(slot %d) failed: %s; the resident vector differs from the one handed over in %d variables%s" % (slots[k].value, msg, len(diff), (" (first: %s)" % diff[:8].tolist()) if len(diff) else ""))
            if len(remade) >= 3: raise RuntimeError("zkgpu_prover_prove_stashed failed again after %d stashes were made anew: %s" % (len(remade), msg))
            if _lib.zkgpu_prover_prove(_h, _zp[k], None, None, _out) != 0: raise RuntimeError("statement %d is not provable from its host buffer either: %s" % (k, _lib.zkgpu_last_error().decode()))
            prover.set_witness(zs[k]); slots[k] = ctypes.c_uint32(prover.stash_witness()); remade.append(k)
            if _lib.zkgpu_prover_prove_stashed(_h, slots[k], None, None, _out) != 0: raise RuntimeError("zkgpu_prover_prove_stashed failed on a stash made anew: %s" % _lib.zkgpu_last_error().decode())
            return _out
        if fail_at.startswith("stash:"):                                      # tests only: statement <k>'s stash holds one wrong value — the step that meets it must say so, make it anew and go on
            k = int(fail_at.split(":")[1]) % n_inst; wrong = zs[k].copy(); wrong[5000 % len(wrong)] = (3, 0, 0, 0); prover.set_witness(wrong); prover.drop_stash(slots[k].value); slots[k] = ctypes.c_uint32(prover.stash_witness())
        def one_proof(i):
            rc = _lib.zkgpu_prover_prove_stashed(_h, slots[i % n_inst], None, None, _out)
            return _out if rc == 0 else stash_failed(i)
    # (the interpreter's cycle collector stays out of the timed region: with torch and numpy loaded a full pass is ~10 ms — thirteen proofs — and its timing is a matter of
    # allocation counts; nothing in the loop makes cycles.  Collected once here, switched back on after the clock stops.)
    import gc; gc.collect(); gc.disable()
    # (the GPU's clocks take ~30 ms of load to rise after the idle set-up: the first proofs of a fresh prover take 0.80-0.85 ms, the 40th 0.75 — tools/first_steps.py,
    # profiles/r05_first_steps.txt.  The W warm-up steps of the contract follow a burst that brings the device to the state a prover under load is in; disclosed in config.clock_warmup.)
    clock_warmup = max(0, 40 - args.warmup)
    for i in range(clock_warmup): one_proof(i)
    for i in range(args.warmup): one_proof(i)
    step_t = [0.0] * (args.steps + 1); clock = time.perf_counter
    barrier(); t0 = clock()
    last = None; step_t[0] = t0
    for i in range(args.steps): last = one_proof(args.warmup + i); step_t[i + 1] = clock()        # (one clock read per step: the spread of the timed region goes into the line)
    barrier(); dt = clock() - t0; gc.enable()
    # the same prover call handed a fresh HOST buffer every step (scan of the 7.3 MB assignment into its compact form, one 0.3 MB copy over PCIe, the expansion on the
    # device): the same K steps, the same barriers, the same percentiles, in every mode — `value_from_host_buffers`, what rounds 1-4 reported as `value`
    hb_rate = hb_step_ms = None
    if not shard:
        gc.collect(); gc.disable()
        for i in range(args.warmup): host_buffer_proof(i)
        hb_t = [0.0] * (args.steps + 1); barrier(); hb_t[0] = clock()
        for i in range(args.steps): host_buffer_proof(args.warmup + i); hb_t[i + 1] = clock()
        barrier(); hb_dt = clock() - hb_t[0]; gc.enable()
        hb_rate, hb_dt = (args.steps / hb_dt, hb_dt) if grp is None else grp.aggregate_throughput(args.steps, hb_dt)
        hb_per = sorted(1e3 * (b - a) for a, b in zip(hb_t, hb_t[1:])); hpct = lambda q: round(hb_per[min(len(hb_per) - 1, int(q * len(hb_per)))], 4) if hb_per else None
        hb_step_ms = {"p10": hpct(0.10), "p50": hpct(0.50), "p90": hpct(0.90), "min": round(hb_per[0], 4) if hb_per else None, "max": round(hb_per[-1], 4) if hb_per else None, "ms_per_step": round(1e3 * hb_dt / args.steps, 4), "upload_ms_last": round(prover.timings()["upload_ms"], 4)}
    units = (args.steps if rank == 0 else 0) if shard else args.steps
    rate, dt = (units / dt, dt) if grp is None else grp.aggregate_throughput(units, dt)           # max over ranks, units summed
    per_step = sorted(1e3 * (b - a) for a, b in zip(step_t, step_t[1:])); pct = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 4) if per_step else None
    step_ms = {"p10": pct(0.10), "p50": pct(0.50), "p90": pct(0.90), "min": round(per_step[0], 4) if per_step else None, "max": round(per_step[-1], 4) if per_step else None, "rank": rank}
    d = insts[(args.warmup + args.steps - 1) % n_inst]
    if last is not None and not isinstance(last, str): last = last.value.decode()
    assert last is None or e.verify(vk_path, last, w.pack_public([d["cmtA_old"], d["sn_old"], d["cmtS"], d["cmtA"]])), "proof from the timed region does not verify"

    extra = {}; cpu_files = {}                                     # cpu_files: circuit legs whose libsnark time is reported beside them
    if world == 1 and not args.no_extra_legs:                      # (N > 1: only the timed region and the roofline leg — the driver's scaling runs pass no flags)
        nx = max(3, min(args.steps, 50))
        # through the drop-in cgo symbol genSendproof (adds witness generation on the host and hex marshalling), one caller and several at once as go-ethereum's goroutines do
        # (the first call builds the key's pool of provers; each of them starts its helper threads with its first proof: five untimed calls, then n_abi timed ones whose
        # argument tuples were built beforehand — what is timed is the symbol, one caller, a different instance every call)
        os.environ["ZK_PRFKEY_DIR"] = tmp; os.environ.setdefault("ZK_PROVERS_PER_KEY", str(max(2, args.inflight))); zk = e.Zk(); abi_args = [w.send_args(d) for d in insts]; n_abi = max(10, min(3 * args.steps, 60))
        for i in range(5): zk.GenSendProof(*abi_args[i % n_inst])
        t_abi = []
        for i in range(n_abi): t0 = time.perf_counter(); abi_proof = zk.GenSendProof(*abi_args[(5 + i) % n_inst]); t_abi.append(1e3 * (time.perf_counter() - t0))
        d_abi = insts[(5 + n_abi - 1) % n_inst]; assert zk.VerifySendProof(abi_proof, d_abi["cmtA_old"], d_abi["sn_old"], d_abi["cmtS"], d_abi["cmtA"]), "the last genSendproof proof does not verify"
        ms_abi = sum(t_abi) / n_abi; t_abi.sort(); extra["through_genSendproof"] = {"calls": n_abi, "ms_per_proof": round(ms_abi, 4), "proofs_per_s": round(1e3 / ms_abi, 2), "ms_p50": round(t_abi[n_abi // 2], 4), "ms_p90": round(t_abi[min(n_abi - 1, int(0.9 * n_abi))], 4)}
        import threading
        if args.inflight > 1:
            def caller(k):
                for i in range(nx): zk.GenSendProof(*abi_args[(i + k) % n_inst])
            ths = [threading.Thread(target=caller, args=(k,)) for k in range(args.inflight)]; t0 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            extra["through_genSendproof"]["concurrent_callers"] = args.inflight; extra["through_genSendproof"]["proofs_per_s_concurrent"] = round(nx * args.inflight / (time.perf_counter() - t0), 2)
            # K prover objects on their own stream sets, one host thread each, host-buffer witnesses
            provers = [prover] + [prover.clone() for _ in range(args.inflight - 1)]                                   # share the key's device tables
            for k, pv in enumerate(provers): pv.prove(zs[k % n_inst])
            per = max(4, 4 * args.steps)                                                                               # (long enough for the steady rate: the first ~50 ms of a leg are threads starting and clocks rising)
            def worker(k):
                for i in range(per): provers[k].prove(zs[(i + k) % n_inst])
            ths = [threading.Thread(target=worker, args=(k,)) for k in range(len(provers))]; t0 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            dti = time.perf_counter() - t0; extra["proofs_in_flight"] = {"provers": args.inflight, "proofs": per * args.inflight, "proofs_per_s": round(per * args.inflight / dti, 2), "ms_per_proof": round(1e3 * dti / (per * args.inflight), 4)}
            for pv in provers[1:]: pv.close()
        # B witnesses against one resident key in one call (BASELINE.json configs[2]: a batch of independent send proofs)
        if args.batch > 1 and hasattr(prover, "prove_batch"):
            import numpy as np
            B = args.batch; batch = np.ascontiguousarray(np.stack([zs[i % n_inst] for i in range(B)])); proofs = prover.prove_batch(batch); proofs = prover.prove_batch(batch); reps = max(4, min(12, 768 // B)); t0 = time.perf_counter()   # the B assignments back to back in one host buffer
            for _ in range(reps): proofs = prover.prove_batch(batch)
            dtb = time.perf_counter() - t0
            ok = all(e.verify(vk_path, proofs[k], w.pack_public([insts[k % n_inst][x] for x in ("cmtA_old", "sn_old", "cmtS", "cmtA")])) for k in (0, B - 1))
            extra["prove_batch"] = {"batch": B, "proofs_per_s": round(B * reps / dtb, 2), "ms_per_proof": round(1e3 * dtb / (B * reps), 4), "verified": ok}
        # BASELINE.json configs[0] beside it: one mint proof (step domain 196,608) on the GPU; its CPU time is in cpu_baseline
        mpk, mvk = os.path.join(tmp, "mintpk.txt"), os.path.join(tmp, "mintvk.txt"); e.keygen("mint", mpk, mvk, seed=0xB10C4A2F); mp = e.Prover(mpk); md = w.mint_instance(0); mw = os.path.join(tmp, "mint_w.bin")
        e.witness_mint_redeem(False, *hx(w.mint_args(md)), mw); mz = read_witness(mw); mp.prove(mz); t0 = time.perf_counter()
        for i in range(nx): mproof = mp.prove(mz)
        ms_mint = 1e3 * (time.perf_counter() - t0) / nx; mp.close()
        assert e.verify(mvk, mproof, w.pack_public([md["cmtA_old"], md["sn_old"], md["cmtA"]], md["value_s"]))
        extra["mint_single_proof"] = {"ms_per_proof": round(ms_mint, 4), "proofs_per_s": round(1e3 / ms_mint, 2)}
        # BASELINE.json configs[3]: deposit + redeem, a mixed batch on one GPU — two more resident keys beside send's and mint's (all four pk / vk pairs of the deployment have
        # then been exercised by this run); 16 + 16 proofs through the prover entry point, the two circuits alternating, host-buffer witnesses, one call at a time
        def circuit_leg(kind, depth=8):
            tag = kind if depth == 8 else "%s_depth%d" % (kind, depth); pk_, vk_, wf = (os.path.join(tmp, tag + x) for x in ("pk.txt", "vk.txt", "_w.bin"))
            t0 = time.time(); e.keygen(kind, pk_, vk_, seed=0xB10C4A30 + depth + len(kind), **({"tree_depth": depth} if kind == "deposit" else {})); t_gen = time.time() - t0
            t0 = time.time(); pv = e.Prover(pk_); t_ld = time.time() - t0
            if kind == "deposit":
                dd = w.deposit_instance(1); rt = dd["rt"] if depth == 8 else w.merkle_root_and_path(dd["leaves"], dd["index"], depth=depth)[0]
                e.witness_deposit(*hx(w.deposit_args(dd)), "".join("0x" + l.hex() for l in dd["leaves"]), len(dd["leaves"]), "0x" + dd["sk"].hex(), wf, **({"tree_depth": depth} if depth != 8 else {}))
                pub = w.pack_public([rt, dd["pk_recv"], dd["cmtB_old"], dd["sn_old"], dd["cmtB"], dd["sn_s"]])
            else:
                md_ = w.mint_instance(1, redeem=True); e.witness_mint_redeem(True, *hx(w.mint_args(md_)), wf); pub = w.pack_public([md_["cmtA_old"], md_["sn_old"], md_["cmtA"]], md_["value_s"])
            cpu_files[tag] = (kind, depth, wf); return pv, read_witness(wf), vk_, pub, {"keygen": round(t_gen, 2), "key_load": round(t_ld, 2)}
        dp, dz, dvk, dpub, dset = circuit_leg("deposit"); rp, rz, rvk, rpub, rset = circuit_leg("redeem")
        dp.prove(dz); rp.prove(rz); n_mix = 16; t0 = time.perf_counter()
        for i in range(n_mix): dproof = dp.prove(dz); rproof = rp.prove(rz)
        dt_mix = time.perf_counter() - t0; t0 = time.perf_counter()
        for i in range(4): dp.prove(dz)
        ms_dep = 1e3 * (time.perf_counter() - t0) / 4; t0 = time.perf_counter()
        for i in range(4): rp.prove(rz)
        ms_red = 1e3 * (time.perf_counter() - t0) / 4; dp.close(); rp.close()
        assert e.verify(dvk, dproof, dpub) and e.verify(rvk, rproof, rpub), "a proof of the mixed batch does not verify"
        extra["deposit_redeem_mixed_batch"] = {"config": "BASELINE.json configs[3]: %d deposit + %d redeem proofs alternating against two resident keys, 1 MI355X" % (n_mix, n_mix), "proofs": 2 * n_mix,
                                               "proofs_per_s": round(2 * n_mix / dt_mix, 2), "ms_per_proof": round(1e3 * dt_mix / (2 * n_mix), 4), "deposit_ms_per_proof": round(ms_dep, 4), "redeem_ms_per_proof": round(ms_red, 4),
                                               "verified": True, "setup_s": {"deposit": dset, "redeem": rset}}
        # BASELINE.json configs[4] on ONE GPU: the deposit circuit with the Merkle depth raised to 32 (1,070,591 variables, step domain 2^20 + 2^17, H query of 1,179,647 points)
        p32, z32, vk32, pub32, set32 = circuit_leg("deposit", 32); proof32 = p32.prove(z32); t0 = time.perf_counter()
        for i in range(3): proof32 = p32.prove(z32)
        ms32 = 1e3 * (time.perf_counter() - t0) / 3; p32.close(); assert e.verify(vk32, proof32, pub32), "the depth-32 deposit proof does not verify"
        extra["deposit_depth32_single_proof"] = {"config": "BASELINE.json configs[4] on one GPU: Merkle depth 32, 2^20-point MSMs", "ms_per_proof": round(ms32, 4), "proofs_per_s": round(1e3 / ms32, 2), "verified": True, "setup_s": set32}

    # ---- roofline leg: HIP-event time of the dominant kernel, same stream, after the timed region --------------------------
    e.profile_enable(True); nprof = max(3, min(args.steps, 10))
    for i in range(nprof): one_proof(i)
    stages = e.profile_report(); e.profile_enable(False)
    per_proof = {k: v["ms_total"] / nprof for k, v in stages.items()}
    dom = "msm_H.accumulate"; dom_ms = stages[dom]["ms_total"] / stages[dom]["count"]
    achieved = H_PAIRS * BYTES_PER_G1_PAIR / (dom_ms * 1e-3) / 1e9
    traffic = traffic_src = None
    pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc):
        try: j = json.load(open(pmc)); traffic = j.get("k_msm_accumulate_H", {}).get("hbm_bytes_per_launch"); traffic_src = "profiles/pmc_summary.json (%s)" % j.get("tag", "untagged")
        except Exception: traffic = None
    roofline = {"bound": "hbm", "kernel": "k_hacc_runs29 (bucket accumulation of the H-query MSM)", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": H_PAIRS * BYTES_PER_G1_PAIR}
    # The kernel shares the chip by design since round 5: the witness MSMs run at a higher wave priority and take their issue slots out of this kernel (gpu_internal.hpp:
    # zk_prio_bits), so its duration INSIDE a proof (avg_launch_ms, what `achieved` is computed from) includes their work.  Its duration ALONE on the chip is measured in this
    # run too: a second prover object on the same resident key with everything on ONE stream (ZK_MSM_ONE_STREAM read at construction), the same HIP-event stage timer
    alone_ms = None
    if not shard:
        try:
            os.environ["ZK_MSM_ONE_STREAM"] = "1"; solo = prover.clone(); del os.environ["ZK_MSM_ONE_STREAM"]
            for i in range(3): solo.prove(zs[i % n_inst])
            e.profile_enable(True)
            for i in range(nprof): solo.prove(zs[i % n_inst])
            st1 = e.profile_report(); e.profile_enable(False); solo.close(); alone_ms = st1[dom]["ms_total"] / st1[dom]["count"]
            roofline["avg_launch_ms_alone"] = round(alone_ms, 4); roofline["achieved_alone"] = round(H_PAIRS * BYTES_PER_G1_PAIR / (alone_ms * 1e-3) / 1e9, 3); roofline["alone_source"] = "measured in this run: a one-stream prover object on the same key, HIP events, %d launches" % st1[dom]["count"]; roofline["frac_alone"] = round(roofline["achieved_alone"] / HBM_PEAK_GBS, 6)
            roofline["note"] = "inside a proof the launch shares the chip with the witness MSMs, which the wave priorities deliberately place under it; alone it takes avg_launch_ms_alone"
        except Exception as ex:
            os.environ.pop("ZK_MSM_ONE_STREAM", None); log("bench: no stand-alone launch time (%s)" % ex); alone_ms = None
    # what actually bounds that kernel: 254-bit field arithmetic on the integer VALU.  A lane lifts the first point of every piece of its run of 11 sorted
    # entries and adds the others: one mixed addition = 8 products + 2 squarings on nine 29-bit limbs = 2,366 VALU instructions in the loop's ISA (profiles/r05_hacc_isa.txt):
    # 1,644 quarter-rate ones (v_mad_u64_u32, v_mul_lo_u32: 4.3 cycles per wave-instruction and SIMD, tools/valu_probe.hip) and 722 full-rate ones (2.25 cycles) — 3.67 cycles
    # per instruction for this mix.  Floors: (i) that mix at the nominal 2.4 GHz; (ii) at the clock the chip sustains under this kernel (GRBM_GUI_ACTIVE / wall time =
    # 2.15 GHz, profiles/r05_hacc_counters.json: the kernel is power-limited); (iii) the products alone at the rate of tools/mul_probe.hip
    entries = H_PAIRS * 16 * (1.0 - 2.0 ** -16); madds = entries - entries / 11.0 - 32768; cyc = 1644 * 4.3 + 722 * 2.25
    t_prod = madds * (8 / 165e9 + 2 / 207e9) * 1e3; t_mix24 = madds / 64 * cyc / 1024 / 2.4e9 * 1e3; t_mix215 = madds / 64 * cyc / 1024 / 2.15e9 * 1e3; ref_ms = alone_ms or dom_ms
    roofline_valu = {"bound": "valu-int (not part of the contract: the figures that track this kernel's quality)", "kernel": roofline["kernel"], "mixed_additions": int(madds), "valu_instructions_per_mixed_addition": 2366,
                     "achieved_ms_in_proof": round(dom_ms, 4), "achieved_ms_alone": round(alone_ms, 4) if alone_ms else None,
                     "floor_ms_instruction_mix_at_2.4GHz": round(t_mix24, 4), "frac_of_mix_floor_at_2.4GHz": round(t_mix24 / ref_ms, 4),
                     "floor_ms_instruction_mix_at_measured_clock_2.15GHz": round(t_mix215, 4), "frac_of_mix_floor_at_measured_clock": round(t_mix215 / ref_ms, 4),
                     "floor_ms_field_products_only": round(t_prod, 4), "frac_of_product_ceiling": round(t_prod / ref_ms, 4), "fractions_relative_to": "the kernel alone on the chip" if alone_ms else "the kernel inside a proof"}

    # ---- CPU baseline legs (rank 0, N = 1 only): the reference's own code on the host cores -------------------------------
    cpu = None; cpu_more = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness"); harness_mt = harness + "_mt"; r1cs_path = os.path.join(tmp, "send_r1cs.bin"); e.circuit_export("send", r1cs_path); w0 = os.path.join(tmp, "w0.bin")
        if os.path.exists(harness):
            kv = harness_kv(harness, "bench_prover", r1cs_path, w0)
            if "prover_total_s" in kv:
                cpu = {"value": round(1.0 / float(kv["prover_total_s"]), 5), "unit": "proofs/s", "cores": 1, "kind": "reference",
                       "sample": "1 send proof by libsnark's r1cs_gg_ppzksnark_prover (oracle/_ref), on a key of the send circuit's shape with synthetic points (timing only); %.2f s" % float(kv["prover_total_s"])}
            if os.path.exists(harness_mt):                                   # the reference's -DMULTICORE build (OpenMP over FFT butterflies and multi_exp chunks) on all host cores
                ncores = usable_cores(); kv = harness_kv(harness_mt, "bench_prover", r1cs_path, w0, env=dict(os.environ, OMP_NUM_THREADS=str(ncores)))
                if "prover_total_s" in kv: cpu_more["multicore"] = {"value": round(1.0 / float(kv["prover_total_s"]), 5), "unit": "proofs/s", "cores": int(kv.get("threads", ncores)), "kind": "reference", "sample": "same proof, libsnark built with -DMULTICORE -fopenmp; %.2f s" % float(kv["prover_total_s"])}
            if not args.no_extra_legs:
                # what one genSendproof call costs in the reference: the key file is parsed and its 943k points decompressed on EVERY call (sendcgo.cpp:345), then the prover runs
                kv = harness_kv(harness, "prove", pk_path, w0, "5", "1234", "5678")
                if "load_pk_s" in kv and "prove_s" in kv:
                    tot = float(kv["load_pk_s"]) + float(kv["prove_s"]); cpu_more["per_call_incl_key_load"] = {"value": round(1.0 / tot, 5), "unit": "proofs/s", "cores": 1, "kind": "reference", "key_load_s": float(kv["load_pk_s"]), "prove_s": float(kv["prove_s"]),
                                                              "sample": "reference operator>> on the 77 MB send key + prover, as every genSendproof call of the reference does; %.1f s" % tot}
                mr = os.path.join(tmp, "mint_r1cs.bin"); e.circuit_export("mint", mr); kv = harness_kv(harness, "bench_prover", mr, os.path.join(tmp, "mint_w.bin"))
                if "prover_total_s" in kv: cpu_more["mint_single_proof"] = {"value": round(1.0 / float(kv["prover_total_s"]), 5), "unit": "proofs/s", "cores": 1, "kind": "reference", "sample": "BASELINE.json configs[0]: 1 mint proof by libsnark's prover; %.2f s" % float(kv["prover_total_s"])}
                # configs[3] / [4]: libsnark's prover on the deposit, redeem and depth-32 deposit circuits (same bench_prover leg; the depth-32 one is given 90 s)
                for tag, (kind, depth, wf) in sorted(cpu_files.items()):
                    rp_ = os.path.join(tmp, tag + "_r1cs.bin"); L_ = e.lib(); rc_ = L_.zkgpu_circuit_export(e.KIND[kind], depth, rp_.encode()); t0 = time.time()
                    try:
                        kv = harness_kv(harness, "bench_prover", rp_, wf, timeout=90 if depth != 8 else 300) if rc_ == 0 else {}
                        if "prover_total_s" in kv: cpu_more[tag + "_single_proof"] = {"value": round(1.0 / float(kv["prover_total_s"]), 5), "unit": "proofs/s", "cores": 1, "kind": "reference", "sample": "1 %s proof by libsnark's prover (bench_prover leg); %.2f s" % (tag, float(kv["prover_total_s"]))}
                    except subprocess.TimeoutExpired:
                        cpu_more[tag + "_single_proof"] = {"value": None, "unit": "proofs/s", "cores": 1, "kind": "reference", "sample": "skipped: libsnark's prover on this circuit did not finish within %d s on this host" % int(time.time() - t0)}
                    try: os.remove(rp_)
                    except OSError: pass
        if cpu is None:
            from oracle import pyoracle as o      # checker code, used here ONLY as the timed CPU baseline
            cs = o.R1CS.load(r1cs_path); cs = cs.swapped() if cs.swap_ab_beneficial() else cs; pk, _ = o.parse_pk(pk_path); t0 = time.time(); o.prove(cs, zs[0], pk, 12345, 67890); t = time.time() - t0
            cpu = {"value": round(1.0 / t, 5), "unit": "proofs/s", "cores": 1, "kind": "port", "sample": "1 send proof by the plain-C oracle prover; %.2f s" % t}

    if rank == 0:
        line = {
            "metric": "Groth16 proofs/sec (send circuit, alt_bn128)", "value": round(rate, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "step_ms": step_ms, "value_p50": (round(1e3 / step_ms["p50"] * (1 if shard else world), 4) if step_ms.get("p50") else None), "value_from_host_buffers": (round(hb_rate, 4) if hb_rate else None), "host_buffers_step_ms": hb_step_ms, "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "u32 limbs (254-bit Fq/Fr Montgomery; all MSM accumulations, the H query's weighted bucket sum and the transforms' tiles on nine 29-bit limbs)", "data": "synthetic witnesses (seeded send instances, every constraint satisfied) on a proving key made by this repo's GPU generator for the real send circuit; only the libsnark CPU leg uses a key with synthetic points of the same shape",
            "config": {"workload": "send circuit single proof per step (252,286 constraints, domain 2^18; BASELINE.json configs[1])", "proofs_per_step": 1 if shard else world, "distinct_witnesses": n_inst,
                       "host_binding": host_binding, "stash_remade": len(remade), "clock_warmup": "%d untimed proofs ahead of the %d warm-up steps (the GPU's clocks need ~30 ms of load to rise after the idle set-up)" % (clock_warmup, args.warmup), "parallelism": ("one proof per step, every query cut into %d contiguous shards, one all-gather of 384 B per rank" % world) if shard else "independent proofs per GPU, no collective",
                       "includes": "one prover call per step on the next of the run's distinct statements, all of them RESIDENT IN HBM when the timed region starts (handed over before the clock starts, kept in device memory as the raw vector; a step classifies its assignment on the device and proves it in place): the classification of the values (0 / 1 / other), R1CS rows, the 7 NTTs of the witness map (4 run per proof; both transforms of C and the final inverse are folded into the L and H queries at key load), 5 MSM, host proof assembly, hex serialisation; excludes witness generation and key load.  `value_p50` = 1 / the median step (the mean carries the host's rare 2-5 ms steps); `value_from_host_buffers` = the same prover call handed a fresh host buffer every step (scan + PCIe + expansion included; N = 1 only) — what rounds 1-4 reported as `value`" if not shard else "one proof per step cut into shards; host-buffer hand-over included"},
            "ranks": rank_info, "roofline": roofline, "roofline_valu": roofline_valu, "cpu_baseline": cpu, "cpu_baseline_variants": cpu_more or None, "extra_legs": extra or None,
            "stage_ms_per_proof": {k: round(v, 4) for k, v in sorted(per_proof.items())}, "prover_timings_ms": prover.timings(), "setup_s": {"keygen": round(t_keygen, 2), "key_load": round(t_load, 2), "hbm": hbm or None}}
        real_stdout.write(json.dumps(line) + "\n"); real_stdout.flush()
    prover.close()
    if grp is not None: grp.close()
    return 0

def main():
    args = parse_args()
    if args.gpus < 1: log("bench: --gpus must be >= 1"); return 2
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ: return launch(args)
    return run_rank(args)

if __name__ == "__main__":
    sys.exit(main())
