"""Seeded mutations of a Groth16 proof's 512-character encoding, for comparing verifier VERDICTS with the reference's (TEST INFRASTRUCTURE).

The reference's verify*proof (sendcgo.cpp:388-448, same code in mintcgo / depositcgo / redeemcgo) takes the eight 64-digit coordinates as raw 256-bit integers and
builds field elements with Fp_model(const bigint&) (fp.tcc:190-194), i.e. modulo q, with Z = 1; r1cs_gg_ppzksnark_verifier_strong_IC then tests only that the three
points are on their curves (no subgroup test for B) before the pairing check.  Its accept set therefore contains every ALIAS c + kq < 2^256 of an accepted proof, the
(-A, -B) malleation and every re-randomisation (A/u, u(B + v delta), C + vA); everything else is rejected.  cases() produces all of these kinds and the invalid ones
around them; who decides what the right verdict is, is the reference (oracle/_ref/ref_harness verifymany; tests/golden/verify_mutations_*.txt hold its answers).

A case = (label, 512 lowercase hex characters, public inputs as integers)."""
from oracle import pyoracle as o

Q, R = o.Q_MOD, o.R_MOD
# third "verdict" of the reference: the process dies.  With B = (0, 0) the reference, which goes on to the pairing although is_well_formed() already failed
# (r1cs_gg_ppzksnark.tcc:528-560), inverts a zero Miller value: assert(!is_zero()) (fp.tcc:648) in a build without -DNDEBUG, "reject" with it.  This repo rejects.
ABORT = 2
def agrees(mine, ref): return int(mine) == (0 if ref == ABORT else ref)
ORDER = ["A.x", "A.y", "B.x.c1", "B.x.c0", "B.y.c1", "B.y.c0", "C.x", "C.y"]            # sendcgo.cpp:113-188

def coords(h): return [int(h[64 * k:64 * k + 64], 16) for k in range(8)]
def to_hex(c):
    assert all(0 <= x < 1 << 256 for x in c); return "".join("%064x" % x for x in c)
def points(c):
    """(A, B, C) as the oracle's affine tuples: G2 coordinates are (c0, c1)"""
    return (c[0], c[1]), ((c[3], c[2]), (c[5], c[4])), (c[6], c[7])
def from_points(A, B, C): return [A[0], A[1], B[0][1], B[0][0], B[1][1], B[1][0], C[0], C[1]]
def g1_neg(P): return (P[0], (Q - P[1]) % Q)
def g2_neg(P): return (P[0], ((Q - P[1][0]) % Q, (Q - P[1][1]) % Q))
def max_alias(x): return ((1 << 256) - 1 - x) // Q

def twist_b():
    """b' = 3 / (9 + u) (alt_bn128_init.cpp:191-197)"""
    inv = o.fq2_op("inv", (9, 1)); return o.fq2_op("mul", (3, 0), inv)
def twist_point(g):
    """a point of the twist curve E'(Fq2) with a random x: with overwhelming probability NOT in the order-r subgroup (the cofactor is ~2^254)"""
    b = twist_b()
    while True:
        big = lambda: g.next() | g.next() << 64 | g.next() << 128 | g.next() << 192
        x = (big() % Q, big() % Q)
        x3 = o.fq2_op("mul", o.fq2_op("sqr", x), x); rhs = ((x3[0] + b[0]) % Q, (x3[1] + b[1]) % Q); y = o.fq2_op("sqrt", rhs)
        if y is not None and o.g2_on_curve((x, y)): return (x, y)

def cases(vk_path, proof_hex, inputs, seed, light=False):
    """light: a subset of about sixty cases (one of each kind) for slow verifiers"""
    g = o.SplitMix64(seed); vk = o.parse_vk(vk_path); delta = o.g2_from(vk.delta_g2)[0]; c0 = coords(proof_hex); A, B, C = points(c0); inputs = list(inputs); out = []
    def rnd(n): return g.next() % n
    def big(): return g.next() | g.next() << 64 | g.next() << 128 | g.next() << 192
    def add(label, c, ins=None): out.append((label, to_hex(c), list(inputs if ins is None else ins)))
    def alias(c, ks): return [x + k * Q for x, k in zip(c, ks)]
    def random_alias(c):
        while True:
            ks = [rnd(max_alias(x) + 1) if rnd(3) else 0 for x in c]
            if any(ks): return alias(c, ks)
    add("canonical", c0)
    # -- aliases of the accepted proof: one coordinate at a time, every multiple that fits; then several at once
    for k in range(8):
        for j in range(1, max_alias(c0[k]) + 1):
            if light and j not in (1, max_alias(c0[k])): continue
            c = list(c0); c[k] += j * Q; add("alias %s +%dq" % (ORDER[k], j), c)
    for i in range(6 if light else 40): add("alias several #%d" % i, random_alias(c0))
    add("alias all +1q", alias(c0, [1] * 8)); add("alias all max", alias(c0, [max_alias(x) for x in c0]))
    # -- special values per coordinate
    for k in range(8):
        for name, v in (("0", 0), ("q", Q), ("q-1", Q - 1), ("2^256-1", (1 << 256) - 1), ("2q", 2 * Q), ("5q", 5 * Q), ("1", 1)):
            if light and name not in ("q", "2^256-1"): continue
            c = list(c0); c[k] = v; add("%s = %s" % (ORDER[k], name), c)
    # -- the all-zero point and its aliases (Z stays 1 in the reference: (0,0) is simply not on the curve)
    for name, idx in (("A", (0, 1)), ("B", (2, 3, 4, 5)), ("C", (6, 7)), ("all", tuple(range(8)))):
        for mul in (0, 1, 3):
            c = list(c0)
            for k in idx: c[k] = mul * Q
            add("%s = (0,0) as %dq" % (name, mul), c)
    # -- off the curve
    for k in range(8):
        c = list(c0); c[k] = (c[k] + 1) % Q; add("%s + 1" % ORDER[k], c)
        c = list(c0); c[k] = big() % Q; add("%s random" % ORDER[k], c)
        if not light: c = list(c0); c[k] = big(); add("%s random 256-bit" % ORDER[k], c)
    for i in range(8):
        k = rnd(8); c = list(c0); c[k] ^= 1 << rnd(254); add("bit flip in %s" % ORDER[k], c if c[k] < 1 << 256 else c0)
    # -- swapped coordinates
    for name, (i, j) in (("A.x<->A.y", (0, 1)), ("B.x.c1<->B.x.c0", (2, 3)), ("B.y.c1<->B.y.c0", (4, 5)), ("C.x<->C.y", (6, 7)), ("A.x<->C.x", (0, 6)), ("A.y<->C.y", (1, 7)), ("B.x.c1<->B.y.c1", (2, 4))):
        c = list(c0); c[i], c[j] = c[j], c[i]; add("swap " + name, c)
    c = list(c0); c[0], c[1], c[6], c[7] = c[6], c[7], c[0], c[1]; add("swap A<->C", c)
    c = list(c0); c[2], c[3], c[4], c[5] = c[4], c[5], c[2], c[3]; add("swap B.x<->B.y", c)
    # -- other points of the groups
    G1, G2 = o.g1_gen(), o.g2_gen()
    add("(-A, -B): accepted malleation", from_points(g1_neg(A), g2_neg(B), C)); add("(-A, -B) aliased", random_alias(from_points(g1_neg(A), g2_neg(B), C)))
    add("-A only", from_points(g1_neg(A), B, C)); add("-B only", from_points(A, g2_neg(B), C)); add("-C", from_points(A, B, g1_neg(C))); add("(-A, -B, -C)", from_points(g1_neg(A), g2_neg(B), g1_neg(C)))
    add("2A", from_points(o.g1_op("dbl", A), B, C)); add("2B", from_points(A, o.g2_op("dbl", B), C)); add("C + G", from_points(A, B, o.g1_op("add", C, G1)))
    add("A = G", from_points(G1, B, C)); add("B = G2", from_points(A, G2, C)); add("(G, G2, G)", from_points(G1, G2, G1)); add("A = C", from_points(C, B, C)); add("C = A", from_points(A, B, A))
    add("(2A, B/2)", from_points(o.g1_op("dbl", A), o.g2_op("mul", B, k=pow(2, -1, R)), C)); add("(2A, B/2) aliased", random_alias(from_points(o.g1_op("dbl", A), o.g2_op("mul", B, k=pow(2, -1, R)), C)))
    # -- re-randomised proofs: (A/u, u(B + v delta), C + vA) is accepted whenever (A, B, C) is
    for i in range(3 if light else 12):
        u, v = 1 + big() % (R - 1), big() % R; A2 = o.g1_op("mul", A, k=pow(u, -1, R)); B2 = o.g2_op("mul", o.g2_op("add", B, o.g2_op("mul", delta, k=v)), k=u); C2 = o.g1_op("add", C, o.g1_op("mul", A, k=v))
        c = from_points(A2, B2, C2); add("re-randomised #%d" % i, c); add("re-randomised #%d aliased" % i, random_alias(c))
        if i < 4: bad = from_points(A2, B2, C); add("re-randomised #%d with the old C" % i, bad)
    # -- B on the twist curve but outside the subgroup (the reference tests the curve equation only, and so decides by its Miller loop)
    for i in range(2 if light else 10):
        T = twist_point(g); c = from_points(A, T, C); add("B off-subgroup #%d" % i, c); add("B off-subgroup #%d aliased" % i, random_alias(c))
        if i < 3: add("B + off-subgroup point #%d" % i, from_points(A, o.g2_op("add", B, T), C))
    # -- aliased AND invalid
    for i in range(4 if light else 24):
        k = rnd(8); c = list(c0); c[k] = (c[k] + 1 + rnd(1000)) % Q; add("invalid and aliased #%d" % i, random_alias(c))
    # -- the public input side: another statement, a missing / an extra input
    for j in range(len(inputs)):
        bad = list(inputs); bad[j] = (bad[j] + 1) % R; add("input %d + 1" % j, c0, bad); add("input %d + 1, proof aliased" % j, random_alias(c0), bad)
    if inputs: add("one input less", c0, inputs[:-1]); add("inputs reversed", c0, inputs[::-1])
    add("one input more", c0, inputs + [0])
    return out

def write_cases(path, cs, verdicts=None):
    with open(path, "w") as f:
        for i, (label, h, ins) in enumerate(cs):
            f.write(("%d " % verdicts[i] if verdicts is not None else "") + h + " %d" % len(ins) + "".join(" %d" % x for x in ins) + ("  # " + label if verdicts is not None else "") + "\n")
def read_golden(path):
    """lines '<verdict> <512 hex> <n> <inputs…>  # label' -> [(label, hex, inputs, verdict)]"""
    out = []
    for line in open(path):
        body, _, label = line.rstrip("\n").partition("  # "); t = body.split(); n = int(t[2]); out.append((label, t[1], [int(x) for x in t[3:3 + n]], int(t[0])))   # verdict: 0 / 1 / 2 = ABORT
    return out
def reference_verdicts(harness, vk_path, cs, tmp):
    """the reference's decisions (ref_harness verifymany) for a list of cases"""
    import os, subprocess
    p = os.path.join(str(tmp), "cases_%d.txt" % os.getpid()); write_cases(p, cs); r = subprocess.run([harness, "verifymany", vk_path, p], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr; v = [ABORT if l.split()[2] == "A" else int(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("v ")]; assert len(v) == len(cs); return v

# ---- the public-input side: C strings as the cgo layer may receive them, for uint256S / uint160S (send/uint256.h:222-248) --------------------------------------
def blob_strings(seed, n_random=60):
    """byte strings (no NUL inside): well-formed, short, long, odd, mixed case, with blanks, with characters that are not hex digits"""
    g = o.SplitMix64(seed); hx = lambda n: "".join("0123456789abcdef"[g.next() % 16] for _ in range(n)); out = []
    out += [b"", b"0", b"0x", b"0X", b"x", b"0x0", b"1", b"0x1", b"0x123456", b"123456", b"0x12345", b"12345", b"0x" + b"f" * 64, b"f" * 64, b"0x" + b"00" * 31 + b"01", b"0x01" + b"00" * 31]
    out += [("0x" + hx(n)).encode() for n in (2, 3, 39, 40, 41, 42, 63, 64, 65, 66, 67, 70, 80, 128, 129)] + [hx(n).encode() for n in (40, 41, 64, 65, 100)]
    out += [("0x" + hx(64)).upper().replace("0X", "0x").encode(), ("0X" + hx(64)).encode(), ("0x" + hx(32) + hx(32).upper()).encode(), ("0x" + hx(40)).upper().encode()]
    out += [b" 0x12", b"  \t\n0x" + hx(64).encode(), b"\r\v\f 0x" + hx(40).encode(), b"0x 12", b"0x12 34", b"0x12\n", b"0x" + hx(64).encode() + b" ", b" " + hx(64).encode() + b"zz"]
    out += [b"0x12zz", b"0x12g34", b"0xg", b"zz", b"0x0x12", b"00x12", b"0xx12", b"-1", b"+1", b"0x-1", b"0x" + hx(30).encode() + b"." + hx(33).encode(), b"0x" + hx(64).encode() + b"#comment",
            b"\xff" + hx(10).encode(), b"0x\xff" + hx(10).encode(), b"0x" + hx(10).encode() + b"\x80" + hx(10).encode(), b"\xa0 0x12", b"0x" + hx(10).encode() + b"G" + hx(10).encode(), b"0x" + hx(20).encode() + b":" + hx(20).encode(),
            b"0x" + hx(20).encode() + b"@" + hx(3).encode(), b"0x" + hx(20).encode() + b"`" + hx(3).encode(), b"0x" + hx(20).encode() + b"/" + hx(3).encode(), b"0x" + hx(20).encode() + b"[" + hx(3).encode()]
    for i in range(n_random):
        n = g.next() % 90; body = hx(n); k = g.next() % 6
        if k == 0 and n: j = g.next() % n; body = body[:j] + "ghxz _-.,:;\t"[g.next() % 12] + body[j + 1:]
        if k == 1: body = body.upper()
        out.append(((" " * (g.next() % 3)) + ("0x" if g.next() % 4 else "") + body).encode())
    assert all(b"\0" not in x for x in out); return out
def write_blob_strings(path, strings):
    with open(path, "w") as f:
        for x in strings: f.write(x.hex() + "\n")
def reference_blobs(harness, strings, tmp):
    """[(uint256S hex, uint160S hex, Compute_PRF(x, 0), Compute_CRH(x, 0))] from the reference's own parser and hashes"""
    import os, subprocess
    p = os.path.join(str(tmp), "blobs_%d.txt" % os.getpid()); write_blob_strings(p, strings); r = subprocess.run([harness, "hexblobs", p], capture_output=True, text=True, timeout=120); assert r.returncode == 0, r.stderr
    v = [tuple(l.split()[2:6]) for l in r.stdout.splitlines() if l.startswith("blob ")]; assert len(v) == len(strings); return v
