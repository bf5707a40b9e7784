"""Seeded BlockMaze instances: SplitMix64(seed = 0x5EED0000 + i) -> sk, r_old, r (32 B), pk_sender,
pk_recv (20 B), value_old in [1, 2^63), value_s in [0, value_old]; dependent hashes derived with plain hashlib SHA-256 in
the byte order of the reference (send/Note.h:30-78, util.h:233-258): blobs are hashed in their in-memory order, which is
the reverse of the big-endian bytes go-ethereum passes as hex."""
import hashlib, struct

class SplitMix64:
    """the PRNG shared by the tests, the oracle (oracle/pyoracle.py has the same class) and oracle/ref_harness.cpp; restated here so that this module
    (used by bench.py's setup) does not import the checker"""
    def __init__(self, seed): self.s = seed & 0xFFFFFFFFFFFFFFFF
    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF; z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF; z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)
    def field(self):
        l = [self.next() for _ in range(4)]; l[3] &= (1 << 61) - 1
        return l[0] | l[1] << 64 | l[2] << 128 | l[3] << 192

def rev(b): return bytes(b)[::-1]
def sha(b): return hashlib.sha256(b).digest()
# all values below are BIG-ENDIAN byte strings as zktx.go holds them (common.Hash / common.Address); blob = reversed
def prf(sk, r): return rev(sha(rev(sk) + rev(r)))
def crh(pk, r): return rev(sha(rev(pk) + rev(r)))
def cmt(value, sn, r): return rev(sha(struct.pack("<Q", value) + rev(sn) + rev(r)))
def cmts(value, pk, rs, sn_old): return rev(sha(struct.pack("<Q", value) + rev(pk) + rev(rs) + rev(sn_old)))

def send_instance(i):
    g = SplitMix64(0x5EED0000 + i)
    def rb(n): return b"".join(struct.pack(">Q", g.next()) for _ in range((n + 7) // 8))[:n]
    sk, r_old, r = rb(32), rb(32), rb(32); pk_sender, pk_recv = rb(20), rb(20)
    value_old = 1 + g.next() % ((1 << 63) - 1); value_s = g.next() % (value_old + 1); value = value_old - value_s
    sn_old = prf(sk, r_old); cmtA_old = cmt(value_old, sn_old, r_old); r_s = crh(pk_sender, r); sn = prf(sk, r); cmtS = cmts(value_s, pk_recv, r_s, sn_old); cmtA = cmt(value, sn, r)
    return dict(sk=sk, r_old=r_old, r=r, pk_sender=pk_sender, pk_recv=pk_recv, value_old=value_old, value_s=value_s, value=value, sn_old=sn_old, cmtA_old=cmtA_old, r_s=r_s, sn=sn, cmtS=cmtS, cmtA=cmtA)

def reference_send_fixture():
    """the fixed instance of libsnark-vnt/src/send/main.cpp:123-142,261-263"""
    def u(h, n=32): return int(h, 16).to_bytes(n, "big")
    sk, r_old, r = u("1"), u("123456"), u("12"); pk_sender, pk_recv = u("456", 20), u("123", 20); value_old, value_s, value = 22, 8, 14
    sn_old = prf(sk, r_old); cmtA_old = cmt(value_old, sn_old, r_old); r_s = crh(pk_sender, r); sn = prf(sk, r); cmtS = cmts(value_s, pk_recv, r_s, sn_old); cmtA = cmt(value, sn, r)
    return dict(sk=sk, r_old=r_old, r=r, pk_sender=pk_sender, pk_recv=pk_recv, value_old=value_old, value_s=value_s, value=value, sn_old=sn_old, cmtA_old=cmtA_old, r_s=r_s, sn=sn, cmtS=cmtS, cmtA=cmtA)

def send_args(d):
    """argument order of zktx.GenSendProof (zktx.go:426)"""
    return (d["value_old"], d["r_s"], d["sn_old"], d["r_old"], d["cmtS"], d["cmtA_old"], d["value_s"], d["pk_recv"], d["value"], d["sn"], d["r"], d["cmtA"], d["sk"], d["pk_sender"])

def mint_instance(i, redeem=False):
    g = SplitMix64(0x5EED1000 + i + (0x800 if redeem else 0))
    def rb(n): return b"".join(struct.pack(">Q", g.next()) for _ in range((n + 7) // 8))[:n]
    sk, r_old, r = rb(32), rb(32), rb(32); value_old = 1 + g.next() % ((1 << 62) - 1); value_s = g.next() % (value_old + 1)
    value = value_old - value_s if redeem else value_old + value_s
    sn_old = prf(sk, r_old); cmtA_old = cmt(value_old, sn_old, r_old); sn = prf(sk, r); cmtA = cmt(value, sn, r)
    return dict(sk=sk, r_old=r_old, r=r, value=value, value_old=value_old, value_s=value_s, sn_old=sn_old, sn=sn, cmtA_old=cmtA_old, cmtA=cmtA)
def mint_args(d): return (d["value"], d["value_old"], d["sn_old"], d["r_old"], d["sn"], d["r"], d["cmtA_old"], d["cmtA"], d["value_s"], d["sk"])

def pack_public(bits_be_blobs, extra_u64=None):
    """public input packing (X_gadget::witness_map): bits of each blob in memory order, MSB first per byte, 253-bit little-endian chunks"""
    bits = []
    for b in bits_be_blobs:
        for byte in rev(b): bits += [(byte >> (7 - j)) & 1 for j in range(8)]
    if extra_u64 is not None:
        for byte in struct.pack("<Q", extra_u64): bits += [(byte >> (7 - j)) & 1 for j in range(8)]
    return [sum(bit << j for j, bit in enumerate(bits[i:i + 253])) for i in range(0, len(bits), 253)]

def merkle_root_and_path(leaves_be, index, depth=8):
    """tree of 2^depth leaves padded with zero leaves, nodes = SHA-256 compression of left||right (no padding) in blob byte order; returns (root, siblings leaf level first), big-endian like everything else here"""
    def comp(l, r): return _sha256_compress(l + r)
    level = [rev(x) for x in leaves_be]; empty = bytes(32); sibs = []; pos = index
    for d in range(depth):
        sib = pos ^ 1; sibs.append(level[sib] if sib < len(level) else empty)
        nxt = [comp(level[i], level[i + 1] if i + 1 < len(level) else empty) for i in range(0, len(level), 2)]
        level = nxt if nxt else [comp(empty, empty)]; empty = comp(empty, empty); pos >>= 1
    return rev(level[0]), [rev(s) for s in sibs]

_K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
      0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
      0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
      0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
def _sha256_compress(block):
    """one SHA-256 compression of a 64-byte block from the standard IV, no padding (FIPS 180-4 §6.2.2)"""
    M = 0xFFFFFFFF; rotr = lambda x, n: ((x >> n) | (x << (32 - n))) & M
    w = list(struct.unpack(">16I", block))
    for i in range(16, 64):
        s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3); s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10); w.append((w[i - 16] + s0 + w[i - 7] + s1) & M)
    H = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]; a, b, c, d, e, f, g, h = H
    for i in range(64):
        t1 = (h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + _K[i] + w[i]) & M; t2 = ((rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))) & M
        h, g, f, e, d, c, b, a = g, f, e, (d + t1) & M, c, b, a, (t1 + t2) & M
    return struct.pack(">8I", *[(x + y) & M for x, y in zip(H, [a, b, c, d, e, f, g, h])])

def deposit_instance(i, n_leaves=16):
    g = SplitMix64(0x5EED2000 + i)
    def rb(n): return b"".join(struct.pack(">Q", g.next()) for _ in range((n + 7) // 8))[:n]
    sk, r_old, r, r_s, sn_A_old = rb(32), rb(32), rb(32), rb(32), rb(32); pk_recv = rb(20); value_old = 1 + g.next() % ((1 << 62) - 1); value_s = 1 + g.next() % ((1 << 61) - 1); value = value_old + value_s
    sn_old = prf(sk, r_old); cmtB_old = cmt(value_old, sn_old, r_old); sn_s = prf(sk, r_s); cmtS = cmts(value_s, pk_recv, r_s, sn_A_old); sn = prf(sk, r); cmtB = cmt(value, sn, r)
    index = g.next() % n_leaves; leaves = [rb(32) for _ in range(n_leaves)]; leaves[index] = cmtS; rt, _ = merkle_root_and_path(leaves, index)
    return dict(sk=sk, r_old=r_old, r=r, r_s=r_s, sn_A_old=sn_A_old, pk_recv=pk_recv, value=value, value_old=value_old, value_s=value_s, sn_old=sn_old, cmtB_old=cmtB_old, sn_s=sn_s, cmtS=cmtS, sn=sn, cmtB=cmtB, leaves=leaves, index=index, rt=rt)
def reference_deposit_fixture():
    """libsnark-vnt/src/deposit/main.cpp:131-167,331-333"""
    def u(h, n=32): return int(h, 16).to_bytes(n, "big")
    sk, r_old, r, r_s, sn_A_old = u("1"), u("123456"), u("12"), u("123"), u("123"); pk_recv = u("123", 20); value, value_old, value_s = 264, 255, 9
    sn_old = prf(sk, r_old); cmtB_old = cmt(value_old, sn_old, r_old); sn_s = prf(sk, r_s); cmtS = cmts(value_s, pk_recv, r_s, sn_A_old); sn = prf(sk, r); cmtB = cmt(value, sn, r)
    leaves = [u(str(k)) for k in range(1, 17)]; leaves[9] = cmtS; rt, _ = merkle_root_and_path(leaves, 9)
    return dict(sk=sk, r_old=r_old, r=r, r_s=r_s, sn_A_old=sn_A_old, pk_recv=pk_recv, value=value, value_old=value_old, value_s=value_s, sn_old=sn_old, cmtB_old=cmtB_old, sn_s=sn_s, cmtS=cmtS, sn=sn, cmtB=cmtB, leaves=leaves, index=9, rt=rt)
def deposit_args(d):
    """argument order of genDepositproof (depositcgo.hpp:13-31) without cmtarray/n/RT/sk, which callers append"""
    return (d["value"], d["value_old"], d["sn_old"], d["r_old"], d["sn"], d["r"], d["sn_s"], d["r_s"], d["cmtB_old"], d["cmtB"], d["value_s"], d["pk_recv"], d["sn_A_old"], d["cmtS"])
