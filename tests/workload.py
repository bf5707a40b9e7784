"""Seeded BlockMaze instances: SplitMix64(seed = 0x5EED0000 + i) -> sk, r_old, r (32 B), pk_sender,
pk_recv (20 B), value_old in [1, 2^63), value_s in [0, value_old]; dependent hashes derived with plain hashlib SHA-256 in
the byte order of the reference (send/Note.h:30-78, util.h:233-258): blobs are hashed in their in-memory order, which is
the reverse of the big-endian bytes go-ethereum passes as hex."""
import hashlib, struct
from oracle.pyoracle import SplitMix64

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
