"""bench.py --gpus N must itself start N ranks (the driver invokes `python bench.py --gpus N` without a launcher) and report the process group's
world size.  Here, without a GPU, the two child ranks must rendezvous over gloo on 127.0.0.1 and then fail loudly at the first device call (no CPU
path exists) with a non-zero status; on a GPU box the same command runs to the JSON line with n_gpus = 2 (ranks sharing the device)."""
import json, os, subprocess, sys
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def run(args, env=None, timeout=900):
    e = dict(os.environ, ZK_BENCH_BACKEND="gloo"); e.pop("RANK", None); e.pop("WORLD_SIZE", None); e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)

def test_gpus_2_starts_two_ranks():
    import torch
    r = run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs"])
    assert "rank 0 of 2 joined the process group (backend gloo)" in r.stderr and "rank 1 of 2 joined the process group (backend gloo)" in r.stderr, r.stderr[-2000:]
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]; assert len(lines) == 1; j = json.loads(lines[0]); assert j["n_gpus"] == 2 and j["config"]["proofs_per_step"] == 2
    else:
        assert r.returncode != 0 and "no HIP device" in r.stderr and r.stdout.strip() == ""

def test_world_size_mismatch_is_an_error():
    r = run(["--gpus", "2"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode == 2 and "does not match WORLD_SIZE" in r.stderr
