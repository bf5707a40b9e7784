"""Multi-device pool planning (pure host logic of gpu.hip / capi_zk.cpp, no GPU needed): how ZK_DEVICES is read and in which order the members of a key's prover pool
are laid out — interleaved by device, so that concurrent cgo callers reach every GPU of the node before two of them share one."""
import ctypes
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
