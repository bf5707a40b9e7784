"""Multi-device pool planning (pure host logic of gpu.hip / capi_zk.cpp, no GPU needed): how ZK_DEVICES is read and in which order the members of a key's prover pool
are laid out — interleaved by device, so that concurrent cgo callers reach every GPU of the node before two of them share one."""
import ctypes, os
from blockmaze_amd import engine as e

def plan(spec, n_visible, fallback=0, per_device=2, n_order=16):
    L = e.lib(); dev = (ctypes.c_int * 64)(); order = (ctypes.c_int * n_order)(*([-1] * n_order))
    n = L.zkgpu_test_device_plan(spec.encode() if spec is not None else None, n_visible, fallback, per_device, dev, order, n_order); return [dev[i] for i in range(n)], [x for x in order if x >= 0]

def test_device_list_parsing():
    assert plan(None, 8)[0] == [0] and plan(None, 8, fallback=3)[0] == [3] and plan("", 8, fallback=11)[0] == [3]            # unset: the single device of ZK_DEVICE / LOCAL_RANK (mod the visible count)
    assert plan("all", 8)[0] == list(range(8)) and plan("all", 1)[0] == [0]
    assert plan("0,2,5", 8)[0] == [0, 2, 5] and plan("2,2,9,1", 8)[0] == [2, 1] and plan("junk", 8)[0] == [0]                 # duplicates and out-of-range entries dropped; nothing usable -> device 0
    assert plan("all", 0)[0] == []

def test_pool_is_interleaved_by_device():
    assert plan("all", 8, per_device=2)[1] == list(range(8)) * 2
    assert plan("all", 2, per_device=3, n_order=6)[1] == [0, 1, 0, 1, 0, 1]
    assert plan(None, 8, per_device=4, n_order=4)[1] == [0, 0, 0, 0]

def pool_plan(D, releases, spill=1):
    L = e.lib(); n = len(releases); rel = (ctypes.c_int * n)(*releases); out = (ctypes.c_int * n)(*([-1] * n))
    built = L.zkgpu_test_pool_plan(D, spill, rel, n, out); return built, list(out)

def test_concurrent_callers_spread_over_the_devices_before_two_share_one():
    """acquire_prover's policy (capi_zk.cpp: zk_pool_pick_device): nobody finishes -> every caller finds the loaded devices busy and opens the next one; once all
    devices hold a pool the least busy one is taken"""
    built, dev = pool_plan(8, [-1] * 20); assert built == 8 and dev[:8] == list(range(8)) and sorted(dev[8:16]) == list(range(8)) and max(dev.count(d) for d in range(8)) == 3
    built, dev = pool_plan(2, [-1] * 6); assert built == 2 and sorted(dev[:2]) == [0, 1] and dev.count(0) == dev.count(1) == 3

def test_one_caller_at_a_time_stays_on_one_device():
    """a process that never has two proofs in flight builds ONE pool (one copy of the key on one GPU), whatever ZK_DEVICES lists"""
    rel = [-1] + list(range(0, 11)); built, dev = pool_plan(8, rel); assert built == 1 and dev == [0] * 12
    rel = [-1, -1] + list(range(0, 10)); built, dev = pool_plan(8, rel); assert built == 2 and set(dev) == {0, 1}     # two in flight at any time: two devices
    built, dev = pool_plan(8, [-1] * 6, spill=3); assert built == 2 and dev == [0, 0, 0, 1, 1, 1]                      # ZK_SPILL_BUSY=3: three proofs per device first

def test_lane_planner_never_starves_a_device():
    """gpu.hip: lane_plan_pick.  The pools are built lazily, one device at a time, every member of every circuit kind taking a stream lane: with a quota per device slot the
    last device still gets its lanes (8 devices x 4 kinds x 6 members used to throw 'no stream lane left' on the later devices)"""
    L = e.lib(); per = (ctypes.c_int * 64)()
    assert L.zkgpu_test_lane_plan(8, 4, 6, per) == 1 and list(per[:8]) == [24] * 8                                   # 192 members, a lane each
    assert L.zkgpu_test_lane_plan(8, 4, 7, per) == 1 and list(per[:8]) == [28] * 8
    assert L.zkgpu_test_lane_plan(1, 4, 7, per) == 1 and per[0] == 28
    worst = L.zkgpu_test_lane_plan(16, 4, 6, per); assert worst == 2 and list(per[:16]) == [14] * 16                  # 16 devices: 14 lanes each, shared by 24 members
    worst = L.zkgpu_test_lane_plan(1, 8, 7, per); assert worst == 2 and per[0] == 31                                  # one device never binds more than 31 lanes

def test_scan_pool_counts_every_chunk_once_under_concurrent_callers():
    """groth16_prover.cpp: ScanPool — one caller at a time gets the pool (one broadcast wakes the helpers, the caller scans along, closes the round and waits only for helpers that are
    inside), the others run their job alone; whoever runs it, every chunk of every round is taken exactly once.  ZK_SCAN_THREADS=4: three helpers even on a small host."""
    import subprocess, sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from blockmaze_amd import engine as e; L = e.lib(); "
            "a = L.zkgpu_test_scan_pool(1, 300); b = L.zkgpu_test_scan_pool(4, 300); print('POOL', a, b)") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, ZK_SCAN_THREADS="4", ZK_SPIN_US="50"), timeout=300)
    tok = r.stdout.split(); assert "POOL" in tok, (r.stdout[-300:], r.stderr[-800:]); a, b = int(tok[tok.index("POOL") + 1]), int(tok[tok.index("POOL") + 2])
    assert a == 300 and 1 <= b <= 1200                      # a lone caller always gets the pool; four callers share it, nobody miscounts (-1)


def test_helper_pools_respect_a_cgroup_cpu_quota(tmp_path):
    """ADVICE (round 4): go-ethereum under a Kubernetes CPU limit without a cpuset sees every CPU of a 256-thread host in its affinity mask; sixteen polling scan helpers
    would get the process throttled on the proof's critical path.  usable_cpus() = min(affinity, quota): cpu.max of cgroup v2, cpu.cfs_quota_us / cpu.cfs_period_us of v1"""
    import ctypes
    from blockmaze_amd import engine as e
    L = e.lib(); q = lambda d: L.zkgpu_test_cgroup_quota(str(d).encode())
    v2 = tmp_path / "v2"; v2.mkdir(); (v2 / "cpu.max").write_text("max 100000\n"); assert q(v2) == 0
    (v2 / "cpu.max").write_text("250000 100000\n"); assert q(v2) == 3                                    # 2.5 CPUs: three threads may run
    (v2 / "cpu.max").write_text("50000 100000\n"); assert q(v2) == 1
    v1 = tmp_path / "v1"; (v1 / "cpu").mkdir(parents=True); (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n"); (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n"); assert q(v1) == 0
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("400000\n"); assert q(v1) == 4
    assert q(tmp_path / "nothing") == 0
