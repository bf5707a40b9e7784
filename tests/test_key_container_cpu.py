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

def test_hand_over_block_classifiers_agree():
    """Prover::set_witness / set_witness_tagged classify an assignment 64 entries at a time, with 256-bit loads where the host has AVX2: those forms must give the scalar
    forms' masks on random blocks (tags 0 / 1 / 2 / 6; elements zero, one, small, full width, in canonical and in Montgomery 'one')."""
    import ctypes, random
    from blockmaze_amd import engine as e
    L = e.lib(); rnd = random.Random(64)
    for it in range(2000):
        tags = (ctypes.c_uint8 * 64)(*[rnd.choice((0, 1, 2, 6, 0, 1)) for _ in range(64)])
        one = [1, 0, 0, 0] if it % 2 else [rnd.getrandbits(64) for _ in range(4)]
        el = []
        for i in range(64):
            c = rnd.randrange(5); v = [0, 0, 0, 0] if c == 0 else list(one) if c == 1 else [rnd.getrandbits(32), 0, 0, 0] if c == 2 else [rnd.getrandbits(64) for _ in range(4)] if c == 3 else [one[0] ^ (1 << rnd.randrange(64)), one[1], one[2], one[3]]
            el += v
        elems = (ctypes.c_uint64 * 256)(*el); o = (ctypes.c_uint64 * 4)(*one); out = (ctypes.c_uint64 * 10)()
        assert L.zkgpu_test_scan_blocks(tags, elems, o, out) == 0
        assert list(out[0:3]) == list(out[3:6]) and list(out[6:8]) == list(out[8:10]), it
        assert out[0] == sum(1 << i for i in range(64) if tags[i] & 1) and out[1] == sum(1 << i for i in range(64) if tags[i] & 2) and out[2] == sum(1 << i for i in range(64) if tags[i] & 4)
        is1 = sum(1 << i for i in range(64) if el[4 * i:4 * i + 4] == one); nz = sum(1 << i for i in range(64) if any(el[4 * i:4 * i + 4])); assert out[6] == is1 and out[7] == nz & ~is1
