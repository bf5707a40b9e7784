"""The fast key container (groth16.hpp) without a GPU: a synthetic transformed key goes through the writer and the mapping reader unchanged; a stale
source stamp, a flipped payload bit and a truncated file are each refused (the caller then falls back to the text key).  The GPU side — same proof bytes from the
container as from the text key, and the load time — is in tests/test_gpu_groth16.py."""
import ctypes, os
from blockmaze_amd import engine as e

def test_container_round_trip_and_refusals(tmp_path):
    L = e.lib(); L.zkgpu_test_key_container.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]
    for shape in ((40, 60, 64), (5000, 7000, 8192), (3, 2, 4)):
        assert L.zkgpu_test_key_container(str(tmp_path / "k.gpucache").encode(), *shape) == 0, shape

def test_container_path_policy(tmp_path, monkeypatch):
    L = e.lib(); pk = str(tmp_path / "sendpk.txt"); open(pk, "wb").write(b"not a key")
    assert L.zkgpu_key_container_valid(pk.encode()) == 0                                    # nothing there yet
    open(pk + ".gpucache", "wb").write(b"ZKGPUKC1" + bytes(300)); assert L.zkgpu_key_container_valid(pk.encode()) == 0   # a header that does not match this key file
