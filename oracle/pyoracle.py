"""TEST INFRASTRUCTURE — ctypes binding of oracle/liboracle.so (the CPU oracle; see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Integers <-> 4x64-bit LE limb arrays; points are tuples of Python ints, None = point at infinity.
"""
import ctypes, os, subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
FR, FQ = 0, 1

def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "oracle"])

def _load():
    so = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(so):
        build()
    return ctypes.CDLL(so)

lib = _load()
_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)
lib.o_domain_size.restype = ctypes.c_size_t
lib.o_witness_map.restype = ctypes.c_size_t
lib.o_setup.restype = ctypes.c_size_t

def limbs(x, n=4):
    return [(int(x) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]

def to_arr(vals, n=4):
    """list of ints -> (len, n) uint64 array"""
    a = np.zeros((len(vals), n), dtype=np.uint64)
    for i, v in enumerate(vals):
        a[i] = limbs(v, n)
    return a

def from_arr(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in a]

def _p(a):
    return a.ctypes.data_as(_u64p) if a is not None else None
def _p32(a):
    return a.ctypes.data_as(_u32p)

class SplitMix64:
    """PRNG shared with oracle/ref_harness.cpp and the product's test drivers."""
    def __init__(self, seed): self.s = seed & 0xFFFFFFFFFFFFFFFF
    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)
    def field(self):
        l = [self.next() for _ in range(4)]; l[3] &= (1 << 61) - 1
        return l[0] | l[1] << 64 | l[2] << 128 | l[3] << 192

# ---- fields -------------------------------------------------------------------------------------------------
_OPS = {"mul": 0, "add": 1, "sub": 2, "inv": 3, "sqr": 4, "neg": 5, "sqrt": 6}
def field_op(field, op, a, b=None):
    A = to_arr(a); B = to_arr(b) if b is not None else None; out = np.zeros_like(A)
    lib.o_field_op(field, _OPS[op], _p(A), _p(B), _p(out), ctypes.c_size_t(len(a)))
    return from_arr(out)
def to_mont(field, a):
    A = to_arr(a); out = np.zeros_like(A); lib.o_fe_to_mont(field, _p(A), _p(out), ctypes.c_size_t(len(a))); return from_arr(out)
def fq2_op(op, a, b=None):
    A = to_arr(list(a)).reshape(-1); B = to_arr(list(b)).reshape(-1) if b is not None else None; out = np.zeros(8, dtype=np.uint64)
    ok = lib.o_fq2_op({"mul": 0, "sqr": 1, "inv": 2, "sqrt": 3, "frob": 4}[op], _p(A), _p(B), _p(out))
    return tuple(from_arr(out)) if ok else None

# ---- groups -------------------------------------------------------------------------------------------------
def g1_arr(pts):
    a = np.zeros((len(pts), 8), dtype=np.uint64)
    for i, p in enumerate(pts):
        if p is not None: a[i, :4] = limbs(p[0]); a[i, 4:] = limbs(p[1])
    return a
def g2_arr(pts):
    a = np.zeros((len(pts), 16), dtype=np.uint64)
    for i, p in enumerate(pts):
        if p is not None:
            (x0, x1), (y0, y1) = p
            a[i, 0:4] = limbs(x0); a[i, 4:8] = limbs(x1); a[i, 8:12] = limbs(y0); a[i, 12:16] = limbs(y1)
    return a
def g1_from(a):
    out = []
    for r in np.asarray(a, dtype=np.uint64).reshape(-1, 8):
        v = from_arr(r); out.append(None if v == [0, 0] else (v[0], v[1]))
    return out
def g2_from(a):
    out = []
    for r in np.asarray(a, dtype=np.uint64).reshape(-1, 16):
        v = from_arr(r); out.append(None if v == [0, 0, 0, 0] else ((v[0], v[1]), (v[2], v[3])))
    return out
_GOPS = {"add": 0, "dbl": 1, "madd": 2, "mul": 3, "neg": 4}
def g1_op(op, a, b=None, k=None):
    A = g1_arr([a]); B = g1_arr([b]) if op in ("add", "madd") else None; K = to_arr([k]) if k is not None else None; out = np.zeros(8, dtype=np.uint64)
    lib.o_g1_op(_GOPS[op], _p(A), _p(B), _p(K), _p(out)); return g1_from(out)[0]
def g2_op(op, a, b=None, k=None):
    A = g2_arr([a]); B = g2_arr([b]) if op in ("add", "madd") else None; K = to_arr([k]) if k is not None else None; out = np.zeros(16, dtype=np.uint64)
    lib.o_g2_op(_GOPS[op], _p(A), _p(B), _p(K), _p(out)); return g2_from(out)[0]
def g1_gen():
    out = np.zeros(8, dtype=np.uint64); lib.o_g1_gen(_p(out)); return g1_from(out)[0]
def g2_gen():
    out = np.zeros(16, dtype=np.uint64); lib.o_g2_gen(_p(out)); return g2_from(out)[0]
def g1_consecutive(k0, n):
    out = np.zeros((n, 8), dtype=np.uint64); K = to_arr([k0]); lib.o_g1_consecutive(_p(K), ctypes.c_size_t(n), _p(out)); return out
def g2_consecutive(k0, n):
    out = np.zeros((n, 16), dtype=np.uint64); K = to_arr([k0]); lib.o_g2_consecutive(_p(K), ctypes.c_size_t(n), _p(out)); return out
def g1_on_curve(p): return bool(lib.o_g1_on_curve(_p(g1_arr([p]))))
def g2_on_curve(p): return bool(lib.o_g2_on_curve(_p(g2_arr([p]))))

def msm_g1(pts_arr, scalars_arr, mixed=False):
    pts_arr = np.ascontiguousarray(pts_arr, dtype=np.uint64); scalars_arr = np.ascontiguousarray(scalars_arr, dtype=np.uint64); out = np.zeros(8, dtype=np.uint64)
    lib.o_msm_g1(_p(pts_arr), _p(scalars_arr), ctypes.c_size_t(len(pts_arr)), int(mixed), _p(out)); return g1_from(out)[0]
def msm_g2(pts_arr, scalars_arr, mixed=False):
    pts_arr = np.ascontiguousarray(pts_arr, dtype=np.uint64); scalars_arr = np.ascontiguousarray(scalars_arr, dtype=np.uint64); out = np.zeros(16, dtype=np.uint64)
    lib.o_msm_g2(_p(pts_arr), _p(scalars_arr), ctypes.c_size_t(len(pts_arr)), int(mixed), _p(out)); return g2_from(out)[0]

# ---- domains ------------------------------------------------------------------------------------------------
def domain_size(min_size): return int(lib.o_domain_size(ctypes.c_size_t(min_size)))
_DOPS = {"fft": 0, "ifft": 1, "cosetfft": 2, "icosetfft": 3, "divZ": 4, "lagrange": 5, "Zt": 6, "addZ": 7}
def domain_op(op, min_size, data=None, t=None):
    """data: (m,4) uint64 array of canonical Fr values (or list of ints)."""
    m = domain_size(min_size)
    if data is not None and not isinstance(data, np.ndarray): data = to_arr(data)
    n = 1 if op == "Zt" else m + 1 if op == "addZ" else m
    out = np.zeros((n, 4), dtype=np.uint64); T = to_arr([t]) if t is not None else None
    rc = lib.o_domain_op(_DOPS[op], ctypes.c_size_t(min_size), _p(np.ascontiguousarray(data)) if data is not None else None, _p(T), _p(out))
    assert rc == 0
    return out

# ---- R1CS / Groth16 -------------------------------------------------------------------------------------------
class R1CS:
    """CSR triple.  rowptr[m]: uint32 (n_cons+1), col[m]: uint32 (nnz) with 0 = ONE, coeff[m]: uint64 (nnz,4) canonical."""
    def __init__(self, n_inputs, n_vars, n_cons, rowptr, col, coeff):
        self.n_inputs, self.n_vars, self.n_cons = n_inputs, n_vars, n_cons
        self.rowptr = [np.ascontiguousarray(x, dtype=np.uint32) for x in rowptr]
        self.col = [np.ascontiguousarray(x, dtype=np.uint32) for x in col]
        self.coeff = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in coeff]
    def _args(self):
        a = [ctypes.c_size_t(self.n_inputs), ctypes.c_size_t(self.n_vars), ctypes.c_size_t(self.n_cons)]
        for m in range(3): a += [_p32(self.rowptr[m]), _p32(self.col[m]), _p(self.coeff[m])]
        return a
    @property
    def domain_m(self): return domain_size(self.n_cons + self.n_inputs + 1)
    def swap_ab_beneficial(self):
        """r1cs.tcc:182-231: swap if B touches more distinct variables than A."""
        return len(np.unique(self.col[1])) > len(np.unique(self.col[0]))
    def swapped(self):
        return R1CS(self.n_inputs, self.n_vars, self.n_cons, [self.rowptr[1], self.rowptr[0], self.rowptr[2]],
                    [self.col[1], self.col[0], self.col[2]], [self.coeff[1], self.coeff[0], self.coeff[2]])
    MAGIC = b"R1CSBM01"
    def save(self, path):
        with open(path, "wb") as f:
            f.write(self.MAGIC); f.write(np.array([self.n_inputs, self.n_vars, self.n_cons], dtype=np.uint64).tobytes())
            for m in range(3):
                f.write(np.array([len(self.col[m])], dtype=np.uint64).tobytes()); f.write(self.rowptr[m].tobytes()); f.write(self.col[m].tobytes()); f.write(self.coeff[m].tobytes())
    @classmethod
    def load(cls, path):
        b = open(path, "rb").read(); assert b[:8] == cls.MAGIC
        ni, nv, nc = (int(x) for x in np.frombuffer(b, dtype=np.uint64, count=3, offset=8)); off = 32; rp, col, co = [], [], []
        for m in range(3):
            nnz = int(np.frombuffer(b, dtype=np.uint64, count=1, offset=off)[0]); off += 8
            rp.append(np.frombuffer(b, dtype=np.uint32, count=nc + 1, offset=off)); off += 4 * (nc + 1)
            col.append(np.frombuffer(b, dtype=np.uint32, count=nnz, offset=off)); off += 4 * nnz
            co.append(np.frombuffer(b, dtype=np.uint64, count=4 * nnz, offset=off)); off += 32 * nnz
        return cls(ni, nv, nc, rp, col, co)

def save_witness(path, z):
    z = np.ascontiguousarray(z, dtype=np.uint64).reshape(-1, 4)
    with open(path, "wb") as f: f.write(np.array([len(z)], dtype=np.uint64).tobytes()); f.write(z.tobytes())
def load_witness(path):
    b = open(path, "rb").read(); n = int(np.frombuffer(b, dtype=np.uint64, count=1)[0]); return np.frombuffer(b, dtype=np.uint64, count=4 * n, offset=8).reshape(n, 4).copy()

def r1cs_is_satisfied(cs, z):
    z = np.ascontiguousarray(z, dtype=np.uint64); return bool(lib.o_r1cs_is_satisfied(*cs._args(), _p(z)))
def witness_map(cs, z):
    z = np.ascontiguousarray(z, dtype=np.uint64); m = cs.domain_m; H = np.zeros((m + 1, 4), dtype=np.uint64); lib.o_witness_map(*cs._args(), _p(z), _p(H)); return H

class ProvingKey:
    """head: (56,) words alpha_g1 beta_g1 delta_g1 beta_g2 delta_g2; A (nA,8); B_idx (nB,), B_g2 (nB,16), B_g1 (nB,8); H (nH,8); L (nL,8)"""
    def __init__(self, head, A, B_idx, B_g2, B_g1, H, L):
        self.head = np.ascontiguousarray(head, dtype=np.uint64); self.A = np.ascontiguousarray(A, dtype=np.uint64)
        self.B_idx = np.ascontiguousarray(B_idx, dtype=np.uint32); self.B_g2 = np.ascontiguousarray(B_g2, dtype=np.uint64); self.B_g1 = np.ascontiguousarray(B_g1, dtype=np.uint64)
        self.H = np.ascontiguousarray(H, dtype=np.uint64); self.L = np.ascontiguousarray(L, dtype=np.uint64)
class VerifyingKey:
    def __init__(self, gt, gamma_g2, delta_g2, IC):
        self.gt = np.ascontiguousarray(gt, dtype=np.uint64); self.gamma_g2 = np.ascontiguousarray(gamma_g2, dtype=np.uint64)
        self.delta_g2 = np.ascontiguousarray(delta_g2, dtype=np.uint64); self.IC = np.ascontiguousarray(IC, dtype=np.uint64)

def prove(cs, z, pk, r, s):
    """cs must be the (possibly A/B-swapped) system stored in the proving key.  Returns 32 words: A(8) B(16) C(8)."""
    z = np.ascontiguousarray(z, dtype=np.uint64); out = np.zeros(32, dtype=np.uint64); R = to_arr([r]); S = to_arr([s])
    lib.o_prove(*cs._args(), _p(z), _p(pk.head), _p(pk.A), ctypes.c_size_t(len(pk.A)), _p32(pk.B_idx), _p(pk.B_g2), _p(pk.B_g1), ctypes.c_size_t(len(pk.B_idx)),
                _p(pk.H), ctypes.c_size_t(len(pk.H)), _p(pk.L), ctypes.c_size_t(len(pk.L)), _p(R), _p(S), _p(out))
    return out
def setup(cs, toxic):
    """toxic = (t, alpha, beta, gamma, delta, g1_gen_scalar, g2_gen_scalar).  cs is swapped here if beneficial (generator :218).  Returns (pk, vk, cs_used)."""
    if cs.swap_ab_beneficial(): cs = cs.swapped()
    nv, ni, m = cs.n_vars, cs.n_inputs, cs.domain_m; T = to_arr(list(toxic))
    head = np.zeros(56, dtype=np.uint64); A = np.zeros((nv + 1, 8), dtype=np.uint64); Bidx = np.zeros(nv + 1, dtype=np.uint32); B2 = np.zeros((nv + 1, 16), dtype=np.uint64); B1 = np.zeros((nv + 1, 8), dtype=np.uint64)
    H = np.zeros((m - 1, 8), dtype=np.uint64); L = np.zeros((nv - ni, 8), dtype=np.uint64); gt = np.zeros(48, dtype=np.uint64); gamma = np.zeros(16, dtype=np.uint64); IC = np.zeros((ni + 1, 8), dtype=np.uint64)
    nB = int(lib.o_setup(*cs._args(), _p(T), _p(head), _p(A), _p32(Bidx), _p(B2), _p(B1), _p(H), _p(L), _p(gt), _p(gamma), _p(IC)))
    pk = ProvingKey(head, A, Bidx[:nB], B2[:nB], B1[:nB], H, L); vk = VerifyingKey(gt, gamma, head[40:56].copy(), IC); return pk, vk, cs
def verify(vk, inputs, proof_words):
    IN = to_arr(list(inputs)) if not isinstance(inputs, np.ndarray) else np.ascontiguousarray(inputs, dtype=np.uint64); P = np.ascontiguousarray(proof_words, dtype=np.uint64)
    return bool(lib.o_verify(_p(vk.gt), _p(vk.gamma_g2), _p(vk.delta_g2), _p(vk.IC), ctypes.c_size_t(len(vk.IC)), _p(IN), ctypes.c_size_t(len(IN)), _p(P)))
def pairing(P, Q):
    out = np.zeros(48, dtype=np.uint64); lib.o_pairing(_p(g1_arr([P])), _p(g2_arr([Q])), _p(out)); return from_arr(out)

lib.o_pk_parse.restype = ctypes.c_void_p
lib.o_pk_ptr.restype = ctypes.c_void_p
lib.o_vk_parse.restype = ctypes.c_size_t
def parse_pk(path):
    """Reference-format proving key file -> (ProvingKey, R1CS as stored in the key)."""
    h = lib.o_pk_parse(path.encode())
    if not h: raise IOError(path)
    h = ctypes.c_void_p(h); sz = (ctypes.c_size_t * 12)(); lib.o_pk_sizes(h, sz)
    nA, nB, nH, nL, Bdom, ni, nv, nc, nza, nzb, nzc, err = (int(x) for x in sz)
    try:
        if err: raise ValueError("pk parse error %d" % err)
        def arr(which, dtype, shape):
            p = lib.o_pk_ptr(h, which); n = int(np.prod(shape))
            return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint64 if dtype == np.uint64 else ctypes.c_uint32)), shape=(n,)).reshape(shape).copy() if n else np.zeros(shape, dtype=dtype)
        pk = ProvingKey(arr(0, np.uint64, (56,)), arr(1, np.uint64, (nA, 8)), arr(6, np.uint32, (nB,)), arr(2, np.uint64, (nB, 16)), arr(3, np.uint64, (nB, 8)), arr(4, np.uint64, (nH, 8)), arr(5, np.uint64, (nL, 8)))
        nnz = (nza, nzb, nzc)
        cs = R1CS(ni, nv, nc, [arr(7 + 3 * m, np.uint32, (nc + 1,)) for m in range(3)], [arr(8 + 3 * m, np.uint32, (nnz[m],)) for m in range(3)], [arr(9 + 3 * m, np.uint64, (nnz[m], 4)) for m in range(3)])
        assert Bdom == nA == nv + 1
        return pk, cs
    finally:
        lib.o_pk_free(h)
def parse_vk(path, max_inputs=64):
    gt = np.zeros(48, dtype=np.uint64); gamma = np.zeros(16, dtype=np.uint64); delta = np.zeros(16, dtype=np.uint64); IC = np.zeros((max_inputs + 1, 8), dtype=np.uint64)
    n = int(lib.o_vk_parse(path.encode(), _p(gt), _p(gamma), _p(delta), _p(IC), ctypes.c_size_t(max_inputs + 1)))
    if not n: raise ValueError("vk parse error: " + path)
    return VerifyingKey(gt, gamma, delta, IC[:n].copy())

def proof_hex(proof_words):
    """A.x A.y B.x.c1 B.x.c0 B.y.c1 B.y.c0 C.x C.y, 64 lowercase hex each (sendcgo.cpp:113-188)."""
    v = from_arr(proof_words); order = [v[0], v[1], v[3], v[2], v[5], v[4], v[6], v[7]]
    return "".join("%064x" % x for x in order)

def proof_words_from_hex(h):
    """inverse of proof_hex: 512 hex characters -> the 32 words verify() takes (values are taken as they are, like sendcgo.cpp:388-448)"""
    v = [int(h[64 * k:64 * k + 64], 16) for k in range(8)]; return to_arr([v[0], v[1], v[3], v[2], v[5], v[4], v[6], v[7]]).reshape(-1)
