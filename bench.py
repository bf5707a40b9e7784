#!/usr/bin/env python3
"""Benchmark of the hot path: Groth16 proofs/sec for BlockMaze's send circuit on MI355X.

A step = one send proof per rank through libzkgpu.so's resident prover (the r1cs_gg_ppzksnark_prover equivalent:
R1CS rows -> 7 NTTs (4 on the device per proof, 3 folded into the key) -> 5 MSMs -> proof assembly; reference r1cs_gg_ppzksnark.tcc:391-506), witness vector handed over as a
host buffer.  N ranks prove independent seeded instances (proofs are independent units: no data-path collective), so
value = N*K proofs / max-over-ranks wall time ("weak" scaling).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` times the dominant kernel (bucket accumulation of the H-query MSM) with HIP
events on the library's compute stream; `cpu_baseline` times the reference's own prover (oracle/_ref, kind "reference") or,
where that binary is absent, the plain-C oracle (kind "port") on one send proof on the host cores.
"""
import argparse, json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

H_PAIRS = 262143                       # H-query size of the send circuit (m - 1, m = 2^18)
BYTES_PER_G1_PAIR = 96                 # 64 B affine point + 32 B scalar, each read once
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--gpus", type=int, default=1); ap.add_argument("--steps", type=int, default=50); ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--inflight", type=int, default=4, help="extra leg (not `value`): this many prover objects per GPU, one host thread each, proofs overlapping on the device; 0/1 = skip")
    ap.add_argument("--shard-msm", action="store_true", help="N > 1 only: all ranks prove ONE proof per step together, each holding 1/N of every query; one all-gather of 384-byte partial records per proof (strong scaling)")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("ZK_DEVICE", str(local_rank))
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # the prover runs five streams; the HIP runtime (initialised by torch below) maps them onto 4 hardware queues by default
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("ZK_BENCH_BACKEND", "nccl")             # "gloo": lets the N > 1 code path run on a box with fewer GPUs than ranks (ranks share devices)
        if backend == "nccl": torch.cuda.set_device(local_rank); dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else: torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count())); dist.init_process_group(backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    from blockmaze_amd import engine as e
    from oracle import pyoracle as o      # used ONLY to read witness files back and for the cpu_baseline leg
    import workload as w

    # ---- untimed setup: test keys for the send circuit (seeded toxic waste), resident prover, witnesses -----------------
    tmp = tempfile.mkdtemp(prefix="zkbench_%d_" % rank); pk_path, vk_path = os.path.join(tmp, "sendpk.txt"), os.path.join(tmp, "sendvk.txt")
    t0 = time.time(); e.keygen("send", pk_path, vk_path, seed=0xB10C4A2E); t_keygen = time.time() - t0
    shard = args.shard_msm and world > 1
    t0 = time.time(); prover = e.Prover(pk_path, rank, world) if shard else e.Prover(pk_path); t_load = time.time() - t0
    n_inst = 4; insts, zs = [], []
    for i in range(n_inst):
        d = w.send_instance(i if shard else rank + i * world); wp = os.path.join(tmp, "w%d.bin" % i)
        e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp); insts.append(d); zs.append(o.load_witness(wp))

    def barrier():
        if dist is not None: dist.barrier()
        if torch.cuda.is_available(): torch.cuda.synchronize()

    from blockmaze_amd import sharding
    coll_dev = None if dist is None else ("cuda" if os.environ.get("ZK_BENCH_BACKEND", "nccl") == "nccl" else "cpu")
    prover.set_witness(zs[0])                                                  # the assignment is resident in HBM before the timed region starts
    if shard:
        def one_proof(i):                                                      # every rank runs the device pipeline on its slice; 384 B per rank are exchanged; rank 0 assembles
            recs = sharding.gather_partials(prover.prove_partial(), dist, coll_dev)
            return prover.finish(recs, 0x1234567 + i, 0x7654321 + i) if rank == 0 else None
    else:
        def one_proof(i): return prover.prove_resident()                        # synchronous: fresh (r, s), returns the serialized proof
    for i in range(args.warmup): one_proof(i)
    barrier(); t0 = time.perf_counter()
    last = None
    for i in range(args.steps): last = one_proof(i)
    barrier(); dt = time.perf_counter() - t0
    rate, dt = sharding.aggregate_throughput((args.steps if rank == 0 else 0) if shard else args.steps, dt, dist, coll_dev)      # max over ranks, units summed
    d = insts[0]
    assert last is None or e.verify(vk_path, last, w.pack_public([d["cmtA_old"], d["sn_old"], d["cmtS"], d["cmtA"]])), "proof from the timed region does not verify"

    # ---- not part of `value`: the same proofs with the witness handed over as a host buffer each time (PCIe inclusive), and through the
    # drop-in cgo symbol genSendproof (adds witness generation on the host and hex marshalling)
    nx = max(3, min(args.steps, 10)); ms_pcie = ms_abi = abi_conc = None
    if not shard:
        t0 = time.perf_counter()
        for i in range(nx): prover.prove(zs[i % n_inst])
        ms_pcie = 1e3 * (time.perf_counter() - t0) / nx
        os.environ["ZK_PRFKEY_DIR"] = tmp; os.environ.setdefault("ZK_PROVERS_PER_KEY", str(max(2, args.inflight))); zk = e.Zk(); zk.GenSendProof(*w.send_args(insts[0])); t0 = time.perf_counter()
        for i in range(nx): zk.GenSendProof(*w.send_args(insts[i % n_inst]))
        ms_abi = 1e3 * (time.perf_counter() - t0) / nx
        if args.inflight > 1:                                                  # the same symbol called from several threads at once, as go-ethereum's goroutines do
            import threading
            def caller(k):
                for i in range(nx): zk.GenSendProof(*w.send_args(insts[(i + k) % n_inst]))
            ths = [threading.Thread(target=caller, args=(k,)) for k in range(args.inflight)]; t0 = time.perf_counter()
            for t in ths: t.start()
            for t in ths: t.join()
            abi_conc = round(nx * args.inflight / (time.perf_counter() - t0), 2)

    # ---- not part of `value`: K proofs in flight per GPU (K prover objects on their own stream sets, one host thread each): what a batch of independent
    # proofs (BASELINE.json configs[2]) or concurrent cgo calls get
    inflight = None
    if not shard and args.inflight > 1:
        import threading
        provers = [prover] + [e.Prover(pk_path) for _ in range(args.inflight - 1)]
        for k, pv in enumerate(provers): pv.set_witness(zs[k % n_inst]); pv.prove_resident()
        per = max(4, args.steps)
        def worker(pv):
            for _ in range(per): pv.prove_resident()
        ths = [threading.Thread(target=worker, args=(pv,)) for pv in provers]; t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dti = time.perf_counter() - t0; inflight = {"provers": args.inflight, "proofs": per * args.inflight, "proofs_per_s": round(per * args.inflight / dti, 2), "ms_per_proof": round(1e3 * dti / (per * args.inflight), 4)}
        for pv in provers[1:]: pv.close()
        prover.set_witness(zs[0])

    # ---- roofline leg: HIP-event time of the dominant kernel, same stream, after the timed region --------------------------
    e.profile_enable(True); nprof = max(3, min(args.steps, 10))
    for i in range(nprof): one_proof(i)
    stages = e.profile_report(); e.profile_enable(False)
    per_proof = {k: v["ms_total"] / nprof for k, v in stages.items()}
    dom = "msm_H.accumulate"; dom_ms = stages[dom]["ms_total"] / stages[dom]["count"]
    achieved = H_PAIRS * BYTES_PER_G1_PAIR / (dom_ms * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc):
        try: traffic = json.load(open(pmc)).get("k_msm_accumulate_H", {}).get("hbm_bytes_per_launch")
        except Exception: traffic = None
    roofline = {"bound": "hbm", "kernel": "k_msm_accumulate_tasks<Fq> (H query)", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
                "traffic": traffic, "avg_launch_ms": round(dom_ms, 4), "algorithmic_bytes_per_launch": H_PAIRS * BYTES_PER_G1_PAIR}

    # what actually bounds that kernel: 254-bit field products on the integer VALU.  One mixed addition = 10 products; the H accumulation does one per
    # non-zero signed 16-bit digit (262,143 scalars x 16 windows, a digit is zero with probability 2^-16); ceiling = tools/fmul_bench.hip on the same chip
    madds = H_PAIRS * 16 * (1.0 - 2.0 ** -16); gprod = madds * 10 / (dom_ms * 1e-3) / 1e9
    roofline_valu = {"bound": "valu-int (not part of the contract: the number that tracks this kernel's quality)", "kernel": roofline["kernel"], "achieved": round(gprod, 2), "peak": 126.0, "unit": "G field products/s", "frac": round(gprod / 126.0, 4)}

    # ---- CPU baseline leg (rank 0, N = 1 only): the reference prover on one send proof ------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness"); r1cs_path = os.path.join(tmp, "send_r1cs.bin"); e.circuit_export("send", r1cs_path)
        if os.path.exists(harness):
            out = subprocess.run([harness, "bench_prover", r1cs_path, os.path.join(tmp, "w0.bin")], capture_output=True, text=True).stdout
            kv = dict(zip(out.split()[0::2], out.split()[1::2])) if False else {}
            for line in out.splitlines():
                tok = line.split()
                for a, b in zip(tok[0::2], tok[1::2]): kv[a] = b
            if "prover_total_s" in kv:
                cpu = {"value": round(1.0 / float(kv["prover_total_s"]), 5), "unit": "proofs/s", "cores": 1, "kind": "reference",
                       "sample": "1 send proof by libsnark's r1cs_gg_ppzksnark_prover (oracle/_ref), key of the send circuit's shape with synthetic points; %.2f s" % float(kv["prover_total_s"])}
        if cpu is None:
            cs = o.R1CS.load(r1cs_path); cs = cs.swapped() if cs.swap_ab_beneficial() else cs; pk, _ = o.parse_pk(pk_path); t0 = time.time(); o.prove(cs, zs[0], pk, 12345, 67890); t = time.time() - t0
            cpu = {"value": round(1.0 / t, 5), "unit": "proofs/s", "cores": 1, "kind": "port", "sample": "1 send proof by the plain-C oracle prover; %.2f s" % t}

    if rank == 0:
        total = args.steps * world
        print(json.dumps({
            "metric": "Groth16 proofs/sec (send circuit, alt_bn128)", "value": round(rate, 4), "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "u32 limbs (254-bit Fq/Fr Montgomery)", "data": "synthetic",
            "config": {"workload": "send circuit single proof per step (252,286 constraints, domain 2^18; BASELINE.json configs[1])", "proofs_per_step": 1 if shard else world, "parallelism": ("one proof per step, every query cut into %d contiguous shards, one all-gather of 384 B per rank" % world) if shard else "independent proofs per GPU, no collective",
                       "includes": "R1CS rows + the 7 NTTs of the witness map (4 run per proof; both transforms of C and the final inverse are folded into the L and H queries at key load) + 5 MSM + host proof assembly + hex serialisation, assignment resident in HBM; excludes witness generation and key load"},
            "proofs_in_flight": inflight, "ms_per_proof_host_buffer_in": ms_pcie and round(ms_pcie, 4), "ms_per_proof_through_genSendproof": ms_abi and round(ms_abi, 4), "proofs_per_s_through_genSendproof_concurrent_callers": abi_conc,
            "roofline": roofline, "roofline_valu": roofline_valu, "cpu_baseline": cpu,
            "stage_ms_per_proof": {k: round(v, 4) for k, v in sorted(per_proof.items())}, "prover_timings_ms": prover.timings(), "setup_s": {"keygen": round(t_keygen, 2), "key_load": round(t_load, 2)}}))
    prover.close()
    if dist is not None: dist.destroy_process_group()

if __name__ == "__main__":
    main()
