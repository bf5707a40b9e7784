#!/usr/bin/env python3
"""Reads what a G1 witness MSM held when its fast path met a degenerate sum (ZK_DEBUG_DUMP_DEGENERATE=<prefix>, msm_impl.hpp: finish_sync) and says where: for the bucket
the flag word names, every lane's entries, the sum they should give (plain affine arithmetic on the key's points) against the lane's record, and lanes whose sums are
equal or opposite.   python tools/degenerate_dump.py <dump.bin> <pk.txt> [A|L]"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import pyoracle as o
import numpy as np
P = o.Q_MOD
def add(a, b):
    if a is None: return b
    if b is None: return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0: return None
        l = 3 * a[0] * a[0] * pow(2 * a[1], P - 2, P) % P
    else: l = (b[1] - a[1]) * pow(b[0] - a[0], P - 2, P) % P
    x = (l * l - a[0] - b[0]) % P; return (x, (l * (a[0] - x) - a[1]) % P)
def neg(a): return None if a is None else (a[0], (P - a[1]) % P)
def times_2k(a, k):
    for _ in range(k): a = add(a, a)
    return a
path, pkp = sys.argv[1], sys.argv[2]; which = sys.argv[3] if len(sys.argv) > 3 else ("L" if "msm_L" in path else "A")
b = open(path, "rb").read(); NB, cap, parity, why, n, LANES, RS, _ = struct.unpack_from("<8I", b, 0); at = 32
fill = np.frombuffer(b, dtype=np.uint32, count=2 * NB, offset=at); at += 8 * NB; lane_off = np.frombuffer(b, dtype=np.uint32, count=NB + 1, offset=at); at += 4 * (NB + 1)
cnt = np.minimum(np.maximum(fill[:NB], fill[NB:]), cap); ent = []
for k in range(NB): ent.append(np.frombuffer(b, dtype=np.uint32, count=int(cnt[k]), offset=at)); at += 4 * int(cnt[k])
p1 = np.frombuffer(b, dtype=np.uint32, count=LANES * RS // 4, offset=at).reshape(LANES, RS // 4); at += LANES * RS; p2 = np.frombuffer(b, dtype=np.uint32, count=NB * RS // 4, offset=at).reshape(NB, RS // 4)
INV261 = pow(pow(2, 261, P), P - 2, P)
def rec(r):
    c = [sum(int(r[12 * s + i]) << (29 * i) for i in range(9)) for s in range(4)]
    if not any(c): return None
    X, Y, ZZ, ZZZ = (v * INV261 % P for v in c)
    if ZZ == 0: return "ZZ=0"
    return (X * pow(ZZ, P - 2, P) % P, Y * pow(ZZZ, P - 2, P) % P)
slots = [s for s in range(24) if why >> (8 + s) & 1]; weight = sum(1 << s for s in slots); key = weight - 1
tot = int(cnt.sum()); T = max(8, (tot + (LANES - NB) - 1) // (LANES - NB))
print("NB %d, cap %d, n %d, flag 0x%x: slots %s -> the bucket of digit %d; %d entries in all, slices of %d; this bucket: %d entries on lanes %d..%d; its sum: %s" % (NB, cap, n, why, slots, weight, tot, T, cnt[key], lane_off[key], lane_off[key + 1] - 1, "ZZ=0" if rec(p2[key]) == "ZZ=0" else "a point"))
pk, cs = o.parse_pk(pkp); pts = o.g1_from(pk.A if which == "A" else pk.L); base = 0 if which == "A" else cs.n_inputs + 1
stride = n
def point_of(e):
    idx = int(e) & 0x7fffffff; w, pos = divmod(idx, stride); q = times_2k(pts[pos], 8 * w); return (neg(q) if int(e) >> 31 else q), pos, w
sums = []
for li, lane in enumerate(range(int(lane_off[key]), int(lane_off[key + 1]))):
    es = ent[key][li * T:(li + 1) * T]; s = None; desc = []
    for e_ in es: q, pos, w = point_of(e_); s = add(s, q); desc.append("%s%d@%d" % ("-" if int(e_) >> 31 else "+", pos + base, w))
    got = rec(p1[lane]); ok = got == s; sums.append((s, lane, desc))
    if not ok or s is None: print("  lane %d (%d entries: %s): expected %s, the lane's record %s" % (lane, len(es), " ".join(desc), "infinity" if s is None else "a point", "infinity" if got is None else got if got == "ZZ=0" else "another point"))
seen = {}
for s, lane, desc in sums:
    if s is None: continue
    if s[0] in seen: print("  lanes %d and %d hold %s sums: [%s] and [%s]" % (seen[s[0]][1], lane, "EQUAL" if seen[s[0]][0][1] == s[1] else "OPPOSITE", " ".join(seen[s[0]][2]), " ".join(desc)))
    else: seen[s[0]] = (s, lane, desc)
print("done: %d lanes looked at" % len(sums))
