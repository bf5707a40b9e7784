"""ctypes binding of libzkgpu.so's engine layer (include/zkgpu.h).

This is plumbing for tests and the benchmark: numpy arrays in, numpy arrays out, every call goes through the C-ABI into
the HIP kernels.  There is no Python or CPU implementation behind it — if the shared library is missing or no MI355X is
visible the calls raise.
"""
import ctypes, os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libzkgpu.so")
if os.environ.get("ZKGPU_LIB"): LIB_PATH = os.environ["ZKGPU_LIB"]   # (A/B of two builds on one box: tools/ab_device.py "ZKGPU_LIB=tools/other_build.bin")

class ZkGpuError(RuntimeError):
    pass

_lib = None
def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZkGpuError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.zkgpu_last_error.restype = ctypes.c_char_p; L.zkgpu_version.restype = ctypes.c_char_p
        L.zkgpu_domain_size.restype = ctypes.c_size_t; L.zkgpu_domain_size.argtypes = [ctypes.c_size_t]
        L.zkgpu_msm_create.restype = ctypes.c_void_p; L.zkgpu_r1cs_create.restype = ctypes.c_void_p
        _lib = L
    return _lib

def _check(rc):
    if rc != 0:
        raise ZkGpuError("zkgpu error %d: %s" % (rc, lib().zkgpu_last_error().decode()))

def _bytes(a):
    a = np.ascontiguousarray(a); return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))

def device_count(): return int(lib().zkgpu_device_count())
def device_numa_node(device=0): return int(lib().zkgpu_device_numa_node(int(device)))   # -1 unknown
def init(): _check(lib().zkgpu_init())

# arrays are uint64 with 4 words per field element (canonical, little-endian), same convention as oracle/pyoracle.py
def field_op(field, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64); out = np.zeros_like(a); n = a.size // 4
    bb = np.ascontiguousarray(b, dtype=np.uint64) if b is not None else None
    _check(lib().zkgpu_test_field_op(field, {"mul": 0, "add": 1, "sub": 2, "inv": 3, "sqr": 4, "neg": 5, "mul_lazy": 6, "sqr_lazy": 7, "sub_lazy": 8, "neg_masked": 9}[op], _bytes(a), _bytes(bb) if bb is not None else None, _bytes(out), ctypes.c_size_t(n))); return out
def fq2_op(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64); out = np.zeros_like(a); n = a.size // 8
    bb = np.ascontiguousarray(b, dtype=np.uint64) if b is not None else None
    _check(lib().zkgpu_test_fq2_op({"mul": 0, "sqr": 1, "inv": 2}[op], _bytes(a), _bytes(bb) if bb is not None else None, _bytes(out), ctypes.c_size_t(n))); return out
def group_op(group, op, a, b=None):
    """group 1: (n,8) words, group 2: (n,16) words.  op: add / dbl / madd / mul_small (b = uint32 multipliers)"""
    a = np.ascontiguousarray(a, dtype=np.uint64); out = np.zeros_like(a); n = a.shape[0]
    if op == "mul_small":
        bb = np.zeros_like(a); bb[:, 0] = np.asarray(b, dtype=np.uint64)
    else:
        bb = np.ascontiguousarray(b, dtype=np.uint64) if b is not None else None
    _check(lib().zkgpu_test_group_op(group, {"add": 0, "dbl": 1, "madd": 2, "mul_small": 3}[op], _bytes(a), _bytes(bb) if bb is not None else None, _bytes(out), ctypes.c_size_t(n))); return out

def msm(group, points, scalars, window_bits=0, filter_ones=False):
    points = np.ascontiguousarray(points, dtype=np.uint64); scalars = np.ascontiguousarray(scalars, dtype=np.uint64); n = scalars.size // 4
    out = np.zeros(8 if group == 1 else 16, dtype=np.uint64)
    fn = lib().zkgpu_msm_g1 if group == 1 else lib().zkgpu_msm_g2
    _check(fn(_bytes(points), _bytes(scalars), ctypes.c_size_t(n), int(window_bits), int(filter_ones), _bytes(out))); return out

class ResidentMsm:
    """bases resident in HBM (one proving-key query); run() launches only the kernels + the host combine"""
    def __init__(self, group, points, window_bits=0, filter_ones=False):
        points = np.ascontiguousarray(points, dtype=np.uint64); self.group = group; self.n = points.shape[0]
        self.h = lib().zkgpu_msm_create(group, _bytes(points), ctypes.c_size_t(self.n), int(window_bits), int(filter_ones))
        if not self.h: raise ZkGpuError(lib().zkgpu_last_error().decode())
    def set_scalars(self, scalars):
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64); _check(lib().zkgpu_msm_set_scalars(ctypes.c_void_p(self.h), _bytes(scalars), ctypes.c_size_t(scalars.size // 4)))
    def run(self):
        out = np.zeros(8 if self.group == 1 else 16, dtype=np.uint64); _check(lib().zkgpu_msm_run(ctypes.c_void_p(self.h), _bytes(out))); return out
    def close(self):
        if self.h: lib().zkgpu_msm_destroy(ctypes.c_void_p(self.h)); self.h = None
    def __del__(self):
        try: self.close()
        except Exception: pass

def domain_size(min_size): return int(lib().zkgpu_domain_size(min_size))
def domain_transform(min_size, op, data):
    data = np.ascontiguousarray(data, dtype=np.uint64).copy()
    _check(lib().zkgpu_domain_transform(ctypes.c_size_t(min_size), {"fft": 0, "ifft": 1, "cosetfft": 2, "icosetfft": 3}[op], _bytes(data))); return data

class R1cs:
    def __init__(self, n_inputs, n_vars, n_cons, rowptr, col, coeff):
        self.n_inputs, self.n_vars, self.n_cons = n_inputs, n_vars, n_cons
        self._keep = [[np.ascontiguousarray(x, dtype=np.uint32) for x in rowptr], [np.ascontiguousarray(x, dtype=np.uint32) for x in col], [np.ascontiguousarray(x, dtype=np.uint64) for x in coeff]]
        P32 = ctypes.POINTER(ctypes.c_uint32); P8 = ctypes.POINTER(ctypes.c_uint8)
        rp = (P32 * 3)(*[x.ctypes.data_as(P32) for x in self._keep[0]]); cl = (P32 * 3)(*[x.ctypes.data_as(P32) for x in self._keep[1]]); co = (P8 * 3)(*[x.ctypes.data_as(P8) for x in self._keep[2]])
        self.h = lib().zkgpu_r1cs_create(ctypes.c_size_t(n_inputs), ctypes.c_size_t(n_vars), ctypes.c_size_t(n_cons), rp, cl, co)
        if not self.h: raise ZkGpuError(lib().zkgpu_last_error().decode())
        self.m = domain_size(n_cons + n_inputs + 1)
    def witness_map(self, z):
        z = np.ascontiguousarray(z, dtype=np.uint64); H = np.zeros((self.m + 1, 4), dtype=np.uint64)
        _check(lib().zkgpu_witness_map(ctypes.c_void_p(self.h), _bytes(z), _bytes(H))); return H
    def close(self):
        if self.h: lib().zkgpu_r1cs_destroy(ctypes.c_void_p(self.h)); self.h = None
    def __del__(self):
        try: self.close()
        except Exception: pass

# ---- circuits, keys, prover, verifier ---------------------------------------------------------------------------------
KIND = {"mint": 0, "send": 1, "deposit": 2, "redeem": 3, "sha256": 100, "merkle": 101, "lesscmp": 102, "cmta": 103, "cmts": 104, "prf": 105, "crh": 106, "unpacker": 107}
def circuit_export(kind, path, tree_depth=8): _check(lib().zkgpu_circuit_export(KIND[kind], tree_depth, path.encode()))
def keygen(kind, pk_path, vk_path, seed=0, tree_depth=8): _check(lib().zkgpu_keygen(KIND[kind], tree_depth, ctypes.c_uint64(seed), pk_path.encode(), vk_path.encode()))
def keygen_from_r1cs(r1cs_path, pk_path, vk_path, seed=0): _check(lib().zkgpu_keygen_from_r1cs(r1cs_path.encode(), ctypes.c_uint64(seed), pk_path.encode(), vk_path.encode()))
def witness_sha256(left32, right32, path): _check(lib().zkgpu_witness_sha256(bytes(left32), bytes(right32), path.encode()))
def witness_hashblock(kind, bits, path): _check(lib().zkgpu_witness_hashblock(KIND[kind] - 104, bytes(bits), path.encode()))
def witness_cmta(bits576, path): _check(lib().zkgpu_witness_cmta(bytes(bits576), path.encode()))
def witness_unpacker(bits, path): _check(lib().zkgpu_witness_unpacker(len(bits), bytes(bits), path.encode()))
def witness_lesscmp(value_old, value_s, path): _check(lib().zkgpu_witness_lesscmp(ctypes.c_uint64(value_old), ctypes.c_uint64(value_s), path.encode()))
def _s(x): return x if isinstance(x, bytes) else x.encode()
def witness_send(value_A, r_s, sn, r, cmt_s, cmtA, value_s, pk_recv, value_A_new, sn_A_new, r_A_new, cmt_A_new, sk, pk_sender, path):
    _check(lib().zkgpu_witness_send(ctypes.c_uint64(value_A), _s(r_s), _s(sn), _s(r), _s(cmt_s), _s(cmtA), ctypes.c_uint64(value_s), _s(pk_recv), ctypes.c_uint64(value_A_new), _s(sn_A_new), _s(r_A_new), _s(cmt_A_new), _s(sk), _s(pk_sender), path.encode()))
def witness_mint_redeem(redeem, value, value_old, sn_old, r_old, sn, r, cmtA_old, cmtA, value_s, sk, path):
    _check(lib().zkgpu_witness_mint_redeem(int(redeem), ctypes.c_uint64(value), ctypes.c_uint64(value_old), _s(sn_old), _s(r_old), _s(sn), _s(r), _s(cmtA_old), _s(cmtA), ctypes.c_uint64(value_s), _s(sk), path.encode()))

def witness_deposit(value, value_old, sn_old, r_old, sn, r, sns, rs, cmtB_old, cmtB, value_s, pk, sn_A_old, cmtS, cmtarray, n, sk, path, tree_depth=8):
    _check(lib().zkgpu_witness_deposit(ctypes.c_uint64(value), ctypes.c_uint64(value_old), _s(sn_old), _s(r_old), _s(sn), _s(r), _s(sns), _s(rs), _s(cmtB_old), _s(cmtB), ctypes.c_uint64(value_s), _s(pk), _s(sn_A_old), _s(cmtS), _s(cmtarray), int(n), _s(sk), int(tree_depth), path.encode()))
def witness_merkle(depth, leaf32, siblings, position, path): _check(lib().zkgpu_witness_merkle(int(depth), bytes(leaf32), b"".join(bytes(x) for x in siblings), ctypes.c_uint64(position), path.encode()))

class Prover:
    """a reference-format proving key resident in HBM"""
    def __init__(self, pk_path, shard_rank=0, shard_world=1):
        lib().zkgpu_prover_load_shard.restype = ctypes.c_void_p
        self.h = lib().zkgpu_prover_load_shard(pk_path.encode(), ctypes.c_size_t(shard_rank), ctypes.c_size_t(shard_world))
        if not self.h: raise ZkGpuError(lib().zkgpu_last_error().decode())
        info = (ctypes.c_size_t * 3)(); _check(lib().zkgpu_prover_info(ctypes.c_void_p(self.h), info)); self.n_vars, self.n_inputs, self.m = (int(x) for x in info)
    def prove(self, z, r=None, s=None):
        """z: (n_vars, 4) uint64 canonical.  r, s: ints or None.  Returns the 512-character proof hex."""
        z = np.ascontiguousarray(z, dtype=np.uint64); assert z.size == 4 * self.n_vars
        R = int(r).to_bytes(32, "little") if r is not None else None; S = int(s).to_bytes(32, "little") if s is not None else None
        out = ctypes.create_string_buffer(513); _check(lib().zkgpu_prover_prove(ctypes.c_void_p(self.h), _bytes(z), R, S, out)); return out.value.decode()
    def set_witness(self, z):
        z = np.ascontiguousarray(z, dtype=np.uint64); assert z.size == 4 * self.n_vars; _check(lib().zkgpu_prover_set_witness(ctypes.c_void_p(self.h), _bytes(z)))
    def prove_resident(self, r=None, s=None):
        R = int(r).to_bytes(32, "little") if r is not None else None; S = int(s).to_bytes(32, "little") if s is not None else None
        out = ctypes.create_string_buffer(513); _check(lib().zkgpu_prover_prove_resident(ctypes.c_void_p(self.h), R, S, out)); return out.value.decode()
    def stash_witness(self):
        """keep the assignment handed over last in HBM; returns its slot"""
        slot = ctypes.c_uint32(0); _check(lib().zkgpu_prover_stash_witness(ctypes.c_void_p(self.h), ctypes.byref(slot))); return int(slot.value)
    def drop_stash(self, slot=None):
        """free a kept assignment (None: all of them)"""
        _check(lib().zkgpu_prover_drop_stash(ctypes.c_void_p(self.h), ctypes.c_uint32(0xffffffff if slot is None else slot)))
    def equal_column_groups(self):
        k = ctypes.c_uint32(0); _check(lib().zkgpu_prover_equal_column_groups(ctypes.c_void_p(self.h), ctypes.byref(k))); return int(k.value)
    def read_stash(self, slot):
        """the kept assignment of a slot back on the host: (n_vars, 4) uint64 canonical"""
        z = np.zeros((self.n_vars, 4), dtype=np.uint64); _check(lib().zkgpu_prover_read_stash(ctypes.c_void_p(self.h), ctypes.c_uint32(slot), z.ctypes.data_as(ctypes.c_void_p))); return z
    def stash_count(self):
        k = ctypes.c_uint32(0); _check(lib().zkgpu_prover_stash_count(ctypes.c_void_p(self.h), ctypes.byref(k))); return int(k.value)
    def prove_stashed(self, slot, r=None, s=None):
        R = int(r).to_bytes(32, "little") if r is not None else None; S = int(s).to_bytes(32, "little") if s is not None else None
        out = ctypes.create_string_buffer(513); _check(lib().zkgpu_prover_prove_stashed(ctypes.c_void_p(self.h), ctypes.c_uint32(slot), R, S, out)); return out.value.decode()
    def prove_partial(self):
        """this shard's 384-byte record of partial sums (device pipeline on the resident witness)"""
        out = ctypes.create_string_buffer(384); _check(lib().zkgpu_prover_prove_partial(ctypes.c_void_p(self.h), out)); return out.raw
    def finish(self, records, r, s):
        """add the shard records (list of 384-byte strings, any order) and assemble the proof with the given randomness"""
        buf = b"".join(records); out = ctypes.create_string_buffer(513)
        _check(lib().zkgpu_prover_finish(ctypes.c_void_p(self.h), buf, ctypes.c_size_t(len(records)), int(r).to_bytes(32, "little"), int(s).to_bytes(32, "little"), out)); return out.value.decode()
    def clone(self):
        """another prover object on the same resident key (shares the device tables, owns its streams and workspaces)"""
        lib().zkgpu_prover_clone.restype = ctypes.c_void_p; h = lib().zkgpu_prover_clone(ctypes.c_void_p(self.h))
        if not h: raise ZkGpuError(lib().zkgpu_last_error().decode())
        c = object.__new__(Prover); c.h = h; c.n_vars, c.n_inputs, c.m = self.n_vars, self.n_inputs, self.m; return c
    def prove_batch(self, zs, rs=None):
        """zs: list of (n_vars, 4) uint64 assignments; rs: list of (r, s) int pairs or None.  One call, len(zs) proofs (512-character hex each)."""
        if isinstance(zs, np.ndarray) and zs.ndim == 3: Z = np.ascontiguousarray(zs, dtype=np.uint64); n = Z.shape[0]; assert Z.shape[1:] == (self.n_vars, 4)   # already back to back: no copy
        else: n = len(zs); Z = np.ascontiguousarray(np.stack([np.ascontiguousarray(z, dtype=np.uint64).reshape(self.n_vars, 4) for z in zs])) if n else np.zeros((0, self.n_vars, 4), dtype=np.uint64)
        R = b"".join(int(r).to_bytes(32, "little") + int(s).to_bytes(32, "little") for r, s in rs) if rs is not None else None
        out = ctypes.create_string_buffer(513 * max(1, n)); _check(lib().zkgpu_prover_prove_batch(ctypes.c_void_p(self.h), _bytes(Z), ctypes.c_size_t(n), R, out))
        return [out.raw[513 * i:513 * i + 512].decode() for i in range(n)]
    def timings(self):
        t = (ctypes.c_double * 5)(); _check(lib().zkgpu_prover_timings(ctypes.c_void_p(self.h), t)); return dict(zip(("upload_ms", "enqueue_ms", "device_ms", "finish_ms", "total_ms"), (float(x) for x in t)))
    def close(self):
        if self.h: lib().zkgpu_prover_destroy(ctypes.c_void_p(self.h)); self.h = None
    def __del__(self):
        try: self.close()
        except Exception: pass

def profile_enable(on=True): _check(lib().zkgpu_profile_enable(int(on)))
def profile_report():
    import json
    buf = ctypes.create_string_buffer(1 << 22); _check(lib().zkgpu_profile_report(buf, ctypes.c_size_t(len(buf)))); return json.loads(buf.value.decode())

def verify_batch(vk_path, proofs_hex, inputs):
    """proofs_hex: list of n 512-character strings; inputs: list of n lists of canonical ints -> list of n booleans (GPU, kernel K9)"""
    n = len(proofs_hex); ni = len(inputs[0]) if n else 0; blob = "".join(proofs_hex).encode(); assert len(blob) == 512 * n
    buf = (ctypes.c_uint8 * max(1, 32 * n * ni))(); 
    for i, row in enumerate(inputs):
        assert len(row) == ni
        for j, v in enumerate(row): buf[32 * (i * ni + j):32 * (i * ni + j + 1)] = list(int(v).to_bytes(32, "little"))
    ok = (ctypes.c_uint8 * max(1, n))(); _check(lib().zkgpu_verify_batch(vk_path.encode(), blob, buf, ctypes.c_size_t(ni), ctypes.c_size_t(n), ok)); return [bool(ok[i]) for i in range(n)]

def verify_schedule_on_host(vk_path, proof_hex, inputs):
    """the GPU verifier's operation schedule (csrc/verify_sched.hpp) interpreted on the host: (accept, {rounds, slots, products, linear_ops, constants, mul_waves, lin8_waves, lin1_waves}); needs no device"""
    buf = b"".join(int(x).to_bytes(32, "little") for x in inputs); st = (ctypes.c_uint32 * 8)()
    rc = lib().zkgpu_test_verify_schedule(vk_path.encode(), proof_hex.encode(), buf, ctypes.c_size_t(len(inputs)), st)
    if rc < 0: _check(rc)
    return rc == 1, dict(zip(("rounds", "slots", "products", "linear_ops", "constants", "mul_waves", "lin8_waves", "lin1_waves"), (int(x) for x in st)))
def equal_columns(r1cs_path):
    """groups (lists of variable numbers, 0 = ONE) of auxiliary variables whose columns coincide in A, B and C (host only)"""
    n = lib().zkgpu_test_equal_columns(r1cs_path.encode(), None, ctypes.c_size_t(0))
    if n < 0: _check(n)
    buf = (ctypes.c_uint32 * max(1, n))(); lib().zkgpu_test_equal_columns(r1cs_path.encode(), buf, ctypes.c_size_t(n)); out = []; i = 0
    while i < n: k = buf[i]; out.append([int(v) for v in buf[i + 1:i + 1 + k]]); i += 1 + k
    return out
def general_path_repeats():
    lib().zkgpu_general_path_repeats.restype = ctypes.c_uint64; return int(lib().zkgpu_general_path_repeats())
def verify_counters(vk_path):
    """(small verification calls taken by the key's GPU verifier, launches made for them)"""
    out = (ctypes.c_uint64 * 2)(); _check(lib().zkgpu_verify_counters(vk_path.encode(), out)); return int(out[0]), int(out[1])
def verify_trace(vk_path, proof_hex, inputs, every=1):
    """kernel K9 on one proof with its values written out after every `every`-th round, compared with the host model of the same arithmetic:
    (first differing round or -1, slot, the kernel's verdict)"""
    buf = b"".join(int(x).to_bytes(32, "little") for x in inputs); out = (ctypes.c_long * 3)()
    _check(lib().zkgpu_test_verify_trace(vk_path.encode(), proof_hex.encode(), buf, ctypes.c_size_t(len(inputs)), ctypes.c_uint32(every), out)); return int(out[0]), int(out[1]), int(out[2])
def verify(vk_path, proof_hex, inputs):
    """inputs: list of ints (packed public input).  True / False."""
    buf = b"".join(int(x).to_bytes(32, "little") for x in inputs)
    rc = lib().zkgpu_verify(vk_path.encode(), proof_hex.encode(), buf, ctypes.c_size_t(len(inputs)))
    if rc < 0: _check(rc)
    return bool(rc)

class Zk:
    """the drop-in symbols (what go-ethereum/zktx calls through cgo), bound the way zktx.go marshals them: "0x…" hex strings and uint64"""
    def __init__(self):
        L = lib()
        for f in ("genCMT", "genCMTS", "computePRF", "computeCRH", "genRoot", "genMintproof", "genSendproof", "genRedeemproof", "genDepositproof"): getattr(L, f).restype = ctypes.c_char_p
        for f in ("verifyMintproof", "verifySendproof", "verifyRedeemproof", "verifyDepositproof"): getattr(L, f).restype = ctypes.c_bool
        self.L = L
    @staticmethod
    def hx(b): return ("0x" + bytes(b).hex()).encode()          # common.ToHex
    def GenCMT(self, value, sn, r): return bytes.fromhex(self.L.genCMT(ctypes.c_uint64(value), self.hx(sn), self.hx(r)).decode())
    def GenCMTS(self, value, pk, rs, sn_old): return bytes.fromhex(self.L.genCMTS(ctypes.c_uint64(value), self.hx(pk), self.hx(rs), self.hx(sn_old)).decode())
    def ComputePRF(self, sk, r): return bytes.fromhex(self.L.computePRF(self.hx(sk), self.hx(r)).decode())
    def ComputeCRH(self, pk, r): return bytes.fromhex(self.L.computeCRH(self.hx(pk), self.hx(r)).decode())
    def GenRT(self, cmts): return bytes.fromhex(self.L.genRoot(b"".join(self.hx(c) for c in cmts), len(cmts)).decode())
    def GenSendProof(self, valueA, rS, snA, rA, cmtS, cmtA, valueS, pk_recv, valueANew, snAnew, rAnew, cmtAnew, sk, pk_sender):   # zktx.go:406-430
        return self.L.genSendproof(ctypes.c_uint64(valueA), self.hx(rS), self.hx(snA), self.hx(rA), self.hx(cmtS), self.hx(cmtA), ctypes.c_uint64(valueS), self.hx(pk_recv), ctypes.c_uint64(valueANew), self.hx(snAnew), self.hx(rAnew), self.hx(cmtAnew), self.hx(sk), self.hx(pk_sender)).decode()
    def VerifySendProof(self, proof, cmtA_old, sn_old, cmtS, cmtA_new): return bool(self.L.verifySendproof(proof.encode(), self.hx(cmtA_old), self.hx(sn_old), self.hx(cmtS), self.hx(cmtA_new)))
    def GenDepositProof(self, value, value_old, sn_old, r_old, sn, r, sns, rs, cmtB_old, cmtB, value_s, pk, sn_A_old, cmtS, cmts, RT, sk):
        return self.L.genDepositproof(ctypes.c_uint64(value), ctypes.c_uint64(value_old), self.hx(sn_old), self.hx(r_old), self.hx(sn), self.hx(r), self.hx(sns), self.hx(rs), self.hx(cmtB_old), self.hx(cmtB), ctypes.c_uint64(value_s),
                                      self.hx(pk), self.hx(sn_A_old), self.hx(cmtS), b"".join(self.hx(c) for c in cmts), len(cmts), self.hx(RT), self.hx(sk)).decode()
    def VerifyDepositProof(self, proof, RT, pk, cmtb_old, sn_old, cmtb, sns): return bool(self.L.verifyDepositproof(proof.encode(), self.hx(RT), self.hx(pk), self.hx(cmtb_old), self.hx(sn_old), self.hx(cmtb), self.hx(sns)))
    def GenMintProof(self, value, value_old, sn_old, r_old, sn, r, cmtA_old, cmtA, value_s, sk):
        return self.L.genMintproof(ctypes.c_uint64(value), ctypes.c_uint64(value_old), self.hx(sn_old), self.hx(r_old), self.hx(sn), self.hx(r), self.hx(cmtA_old), self.hx(cmtA), ctypes.c_uint64(value_s), self.hx(sk)).decode()
    def VerifyMintProof(self, proof, cmtA_old, sn_old, cmtA, value_s): return bool(self.L.verifyMintproof(proof.encode(), self.hx(cmtA_old), self.hx(sn_old), self.hx(cmtA), ctypes.c_uint64(value_s)))
    def VerifyBatch(self, items):
        """include/zk_batch.h: items = list of (kind, proof_hex, [big-endian byte strings in the order of the kind's verify symbol], value_s) -> (accepted, [bool])"""
        class Item(ctypes.Structure): _fields_ = [("kind", ctypes.c_int), ("proof", ctypes.c_char_p), ("args", ctypes.c_char_p * 6), ("value_s", ctypes.c_uint64)]
        arr = (Item * max(1, len(items)))(); keep = []
        for i, (kind, proof, args, value_s) in enumerate(items):
            arr[i].kind = KIND[kind] if isinstance(kind, str) else int(kind); pb = proof.encode() if isinstance(proof, str) else proof; keep.append(pb); arr[i].proof = pb; arr[i].value_s = int(value_s or 0)
            for j, a in enumerate(args): hb = self.hx(a); keep.append(hb); arr[i].args[j] = hb
        ok = (ctypes.c_ubyte * max(1, len(items)))(); self.L.verifyBatch.restype = ctypes.c_int; rc = self.L.verifyBatch(arr, len(items), ok); return rc, [bool(ok[i]) for i in range(len(items))]
    def GenRedeemProof(self, value, value_old, sn_old, r_old, sn, r, cmtA_old, cmtA, value_s, sk):
        return self.L.genRedeemproof(ctypes.c_uint64(value), ctypes.c_uint64(value_old), self.hx(sn_old), self.hx(r_old), self.hx(sn), self.hx(r), self.hx(cmtA_old), self.hx(cmtA), ctypes.c_uint64(value_s), self.hx(sk)).decode()
    def VerifyRedeemProof(self, proof, cmtA_old, sn_old, cmtA, value_s): return bool(self.L.verifyRedeemproof(proof.encode(), self.hx(cmtA_old), self.hx(sn_old), self.hx(cmtA), ctypes.c_uint64(value_s)))
