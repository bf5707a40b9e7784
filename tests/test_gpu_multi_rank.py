"""The N > 1 paths of bench.py on ONE GPU box (ranks share the device over gloo; no multi-GPU node is available to this repo's rounds, so this is as close as the
data path gets to hardware): MSM sharding end to end, and the host side of a rank when 8 of them share the pod's cores."""
import json, os, subprocess, sys
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

def run(args, timeout=900):
    e = dict(os.environ, ZK_BENCH_BACKEND="gloo"); e.pop("RANK", None); e.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--no-cpu-baseline"], capture_output=True, text=True, env=e, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]; assert len(lines) == 1, r.stdout[-500:]; return json.loads(lines[0])

def test_shard_msm_two_ranks_end_to_end():
    """bench.py --gpus 2 --shard-msm: every rank holds half of each query, the five partial sums travel in one all-gather of 384 bytes per rank, rank 0 assembles the proof —
    and bench.py verifies the last proof of the timed region under the vk before it prints its line (an unverified proof is an assertion failure, rc != 0)"""
    j = run(["--gpus", "2", "--shard-msm", "--steps", "4", "--warmup", "1"])
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["proofs_per_step"] == 1 and "2 contiguous shards" in j["config"]["parallelism"] and j["value"] > 0

def test_eight_ranks_keep_their_host_time():
    """8 ranks on this pod (sharing the one GPU): every rank is pinned to 1/8 of the usable cores and sizes its helper threads for them (bench.py); the HOST time a rank
    spends per proof — hand-over of the assignment + kernel submission — must stay below the device time of a proof, so that on 8 real GPUs the host does not bound the rate.
    One shared key file: rank 0 generates it, the others load its container."""
    one = run(["--gpus", "1", "--steps", "20", "--warmup", "3", "--no-extra-legs"]); eight = run(["--gpus", "8", "--steps", "10", "--warmup", "2"], timeout=1500)
    h1 = one["prover_timings_ms"]["upload_ms"] + one["prover_timings_ms"]["enqueue_ms"]; h8 = eight["prover_timings_ms"]["upload_ms"] + eight["prover_timings_ms"]["enqueue_ms"]
    print("host ms per proof (upload + enqueue): 1 rank %.3f, 8 ranks sharing the pod %.3f; key load 1 rank %.2f s, rank 0 of 8 %.2f s" % (h1, h8, one["setup_s"]["key_load"], eight["setup_s"]["key_load"]))
    assert eight["n_gpus"] == 8 and eight["extra_legs"] is None and h8 < max(1.0, 3 * h1), (h1, h8)

def test_two_ranks_weak_scaling_end_to_end():
    """bench.py --gpus 2 as the driver's scaling run starts it (no --shard-msm): two ranks, each proving its own instances against the ONE key rank 0 generated (the path of
    its private directory travels through broadcast_object_list), max-over-ranks timing; every rank verifies the last proof of its timed region before the line is printed
    (an unverified proof is an assertion failure on that rank, hence rc != 0 for the run)"""
    j = run(["--gpus", "2", "--steps", "6", "--warmup", "2"])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["proofs_per_step"] == 2 and j["extra_legs"] is None and j["value"] > 0
    assert set(j["step_ms"]) >= {"p10", "p50", "p90"} and j["step_ms"]["p10"] <= j["step_ms"]["p50"] <= j["step_ms"]["p90"]

@pytest.mark.parametrize("where", ["keygen:0", "load:0", "load:1"])
def test_a_failing_rank_takes_the_group_down(where):
    """rank failure modes of the set-up (key generation on rank 0, key load on either rank): the other rank must not be left at a barrier — the ranks agree on success after
    every stage (sharding.Group.all_ok), everybody leaves with a non-zero status well inside the timeout, and nothing is printed on stdout"""
    import time
    env = dict(os.environ, ZK_BENCH_BACKEND="gloo", ZK_BENCH_TEST_FAIL=where); env.pop("RANK", None); env.pop("WORLD_SIZE", None); t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and time.time() - t0 < 240, (r.returncode, r.stdout[-300:], r.stderr[-1500:])
    assert "failed at '%s'" % where.split(":")[0] in r.stderr and "a rank failed at" in r.stderr, r.stderr[-1500:]

def test_a_stash_that_does_not_prove_is_reported_and_made_anew():
    """bench.py's timed region proves statements kept in HBM.  A resident vector that no longer is the statement handed over (here: one value replaced on purpose) makes the
    step fail with the violated constraint; the run says how the resident vector differs from the host's, proves the statement from its host buffer, makes the stash
    once more and carries the count in its line — it does not die without a line, and it does not hide the event"""
    env = dict(os.environ, ZK_BENCH_TEST_FAIL="stash:3"); env.pop("RANK", None); env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extra-legs"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.strip()][0]); assert j["config"]["stash_remade"] == 1 and j["value"] > 0
    assert "differs from the one handed over in 1 variables (first: [5000])" in r.stderr and "constraint " in r.stderr, r.stderr[-2000:]
    j = run(["--steps", "4", "--warmup", "1", "--no-extra-legs"]); assert j["config"]["stash_remade"] == 0
