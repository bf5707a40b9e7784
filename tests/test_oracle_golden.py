"""The CPU oracle (oracle/*.c) against the golden vectors produced by the REAL reference (tests/golden, made by
oracle/make_golden.py from oracle/_ref/ref_harness).  This is what pins the oracle; the GPU parity tests then compare
the HIP path with the oracle."""
import json, os
import numpy as np
import pytest
from oracle import pyoracle as o

def kv(parts): return {p.split("=", 1)[0]: p.split("=", 1)[1] for p in parts if "=" in p}
def H(s): return int(s, 16)
def P1(s): return None if s == "inf" else tuple(H(x) for x in s.split(","))
def P2(s):
    if s == "inf": return None
    v = [H(x) for x in s.split(",")]; return ((v[0], v[1]), (v[2], v[3]))

def test_field_vectors(ref_vectors):
    n = 0
    for l in ref_vectors:
        if l[0] in ("fr", "fq"):
            f = o.FR if l[0] == "fr" else o.FQ; a, b = H(l[1]), H(l[2]); d = kv(l[3:])
            for op in ("mul", "add", "sub"): assert o.field_op(f, op, [a], [b])[0] == H(d[op])
            for op in ("inv", "sqr", "neg"): assert o.field_op(f, op, [a])[0] == H(d[op])
            if l[0] == "fq": assert o.field_op(f, "sqrt", [H(d["sqr"])])[0] == H(d["sqrt_of_sqr"])
            n += 1
        if l[0] == "fq2":
            a, b = P1(l[1]), P1(l[2]); d = kv(l[3:])
            assert o.fq2_op("mul", a, b) == P1(d["mul"]) and o.fq2_op("sqr", a) == P1(d["sqr"]) and o.fq2_op("inv", a) == P1(d["inv"])
            assert o.fq2_op("sqrt", P1(d["sqr"])) == P1(d["sqrt_of_sqr"]) and o.fq2_op("frob", a) == P1(d["frob"])
            n += 1
        if l[0] == "mont_one_fr": assert o.to_mont(o.FR, [1])[0] == H(l[1])
        if l[0] == "mont_one_fq": assert o.to_mont(o.FQ, [1])[0] == H(l[1])
    assert n == 24

def test_curve_vectors(ref_vectors):
    n = 0
    for l in ref_vectors:
        if l[0] == "g1":
            d = kv(l[1:]); G = o.g1_gen(); p = o.g1_op("mul", G, k=H(d["a"])); q = o.g1_op("mul", G, k=H(d["b"]))
            assert p == P1(d["P"]) and q == P1(d["Q"]) and o.g1_op("add", p, q) == P1(d["add"]) and o.g1_op("dbl", p) == P1(d["dbl"])
            assert o.g1_op("madd", p, q) == P1(d["madd"]) and o.g1_op("mul", p, k=H(d["k"])) == P1(d["kP"]) and o.g1_op("add", p, o.g1_op("neg", q)) == P1(d["PminusQ"])
            assert o.g1_on_curve(p); n += 1
        if l[0] == "g2":
            d = kv(l[1:]); G = o.g2_gen(); p = o.g2_op("mul", G, k=H(d["a"])); q = o.g2_op("mul", G, k=H(d["b"]))
            assert p == P2(d["P"]) and q == P2(d["Q"]) and o.g2_op("add", p, q) == P2(d["add"]) and o.g2_op("dbl", p) == P2(d["dbl"])
            assert o.g2_op("madd", p, q) == P2(d["madd"]) and o.g2_op("mul", p, k=H(d["k"])) == P2(d["kP"]) and o.g2_on_curve(p); n += 1
        if l[0] == "g1_one": assert o.g1_gen() == P1(l[1])
        if l[0] == "g2_one": assert o.g2_gen() == P2(l[1])
    assert n == 12

def test_edge_cases_group_law():
    G = o.g1_gen()
    assert o.g1_op("add", G, None) == G and o.g1_op("add", None, G) == G and o.g1_op("madd", None, G) == G
    assert o.g1_op("add", G, o.g1_op("neg", G)) is None and o.g1_op("mul", G, k=0) is None and o.g1_op("mul", G, k=o.R_MOD) is None
    assert o.g1_op("add", G, G) == o.g1_op("dbl", G) == o.g1_op("madd", G, G)          # doubling fallback (alt_bn128_g1.cpp:167-171, :279-283)
    G2 = o.g2_gen()
    assert o.g2_op("add", G2, o.g2_op("neg", G2)) is None and o.g2_op("mul", G2, k=o.R_MOD) is None and o.g2_op("add", G2, G2) == o.g2_op("dbl", G2)

def test_domain_vectors(ref_vectors):
    R = o.R_MOD; dom = {}
    for l in ref_vectors:
        if l[0] == "domain": dom.setdefault(int(l[1].split("=")[1]), {"dm": int(l[2].split("=")[1])})[l[3]] = l[4:]
        if l[0] == "domain_select": assert o.domain_size(int(l[1].split("=")[1])) == int(l[2].split("=")[1])
    assert len(dom) == 11
    for m, d in dom.items():
        assert o.domain_size(m) == d["dm"]; t = H(d["t"][0]); g = o.SplitMix64(0xD0D0 + m); a = [g.field() for _ in range(d["dm"])]; assert g.field() == t
        for op in ("fft", "ifft", "cosetfft", "icosetfft", "divZ", "lagrange"):
            out = o.from_arr(o.domain_op(op, m, a if op != "lagrange" else None, t=t if op == "lagrange" else None))
            if d["dm"] <= 80: assert out == [H(x) for x in d[op]], (m, op)
            else:
                e = {p.split("=")[0]: H(p.split("=")[1]) for p in d[op]}; acc, w = 0, 1
                for x in out: acc = (acc + w * x) % R; w = w * t % R
                assert (acc, out[0], out[-1]) == (e["polyeval_at_t"], e["first"], e["last"]), (m, op)
        assert o.from_arr(o.domain_op("Zt", m, t=t))[0] == H(d["Zt"][0])
        h = o.from_arr(o.domain_op("addZ", m, t=t)); assert {i: x for i, x in enumerate(h) if x} == {int(p.split(":")[0]): H(p.split(":")[1]) for p in d["addZ"]}

def msm_inputs(n):
    g = o.SplitMix64(0x3535 + n); b0 = g.field(); k, z = [], []
    for _ in range(n):
        k.append(g.field()); sel = g.next() % 100
        z.append(0 if sel < 50 else 1 if sel < 95 else (g.next() & 0xFFFFFFFF) if sel < 98 else g.field())
    return b0, o.to_arr(k), o.to_arr(z)

def test_msm_vectors(ref_vectors):
    n_seen = 0
    for l in ref_vectors:
        if l[0] != "msm": continue
        d = kv(l[1:]); n = int(d["n"]); b0, K, Z = msm_inputs(n); assert b0 == H(d["b0"])
        P = o.g1_consecutive(b0, n); Q = o.g2_consecutive(b0, n)
        assert o.msm_g1(P, K) == P1(d["g1_full"]) and o.msm_g1(P, Z, True) == P1(d["g1_mixed"])
        assert o.msm_g2(Q, K) == P2(d["g2_full"]) and o.msm_g2(Q, Z, True) == P2(d["g2_mixed"]); n_seen += 1
    assert n_seen == 5
    assert o.msm_g1(np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64)) is None                      # empty input

def test_pairing_vectors(ref_vectors):
    n = 0
    for l in ref_vectors:
        if l[0] != "pairing": continue
        d = kv(l[1:3]); gt = [int(x) for x in " ".join(l[3:]).split("=")[1].split()]
        assert o.pairing(o.g1_op("mul", o.g1_gen(), k=H(d["a"])), o.g2_op("mul", o.g2_gen(), k=H(d["b"]))) == gt; n += 1
    assert n == 3

def test_point_serialisation_vectors(ref_vectors, tmp_path):
    """compressed Montgomery point encoding of the key files (alt_bn128_g1.cpp:404-465): parse a one-point vk-shaped stream"""
    for l in ref_vectors:
        if l[0] == "g1_ser" and l[1] == "12345G":
            raw = bytes.fromhex(l[3]); assert len(raw) == 34 and raw[0:1] == b"0"
            x = int.from_bytes(raw[1:33], "little"); exp = P1(l[2]); assert x == o.to_mont(o.FQ, [exp[0]])[0] and raw[33] - 48 == exp[1] & 1
        if l[0] == "g1_ser" and l[1] == "zero": assert bytes.fromhex(l[3]) == b"1" + bytes(32) + b"1"

@pytest.mark.parametrize("name", ["groth16_small", "groth16_step"])
def test_groth16_against_reference(golden_dir, name):
    d = os.path.join(golden_dir, name); meta = json.load(open(os.path.join(d, "meta.json")))
    pk, cs = o.parse_pk(os.path.join(d, "pk.txt")); vk = o.parse_vk(os.path.join(d, "vk.txt")); z = o.load_witness(os.path.join(d, "wit.bin"))
    orig = o.R1CS.load(os.path.join(d, "r1cs.bin"))
    if orig.swap_ab_beneficial(): orig = orig.swapped()                                                    # generator :218
    for m in range(3): assert np.array_equal(orig.rowptr[m], cs.rowptr[m]) and np.array_equal(orig.col[m], cs.col[m]) and np.array_equal(orig.coeff[m], cs.coeff[m])
    assert cs.domain_m == meta["domain_m"] and o.r1cs_is_satisfied(cs, z)
    assert np.array_equal(o.witness_map(cs, z), o.load_witness(os.path.join(d, "h_coeffs.bin")))
    pr = o.prove(cs, z, pk, H(meta["r"]), H(meta["s"]))
    assert o.proof_hex(pr) == meta["proof"]                                                                # byte parity with the reference prover
    assert o.verify(vk, z[:cs.n_inputs], pr)
    bad = pr.copy(); bad[24] ^= 1; assert not o.verify(vk, z[:cs.n_inputs], bad)                          # C.x perturbed: off-curve
    zin = z[:cs.n_inputs].copy(); zin[0, 0] ^= 1; assert not o.verify(vk, zin, pr)                         # wrong public input
    assert not o.verify(vk, z[:cs.n_inputs - 1], pr)                                                       # strong IC: wrong input count

def test_oracle_setup_is_consistent(golden_dir):
    """oracle generator (injected toxic waste) -> oracle prover -> oracle verifier; and unsatisfied witnesses fail"""
    d = os.path.join(golden_dir, "groth16_step"); cs = o.R1CS.load(os.path.join(d, "r1cs.bin")); z = o.load_witness(os.path.join(d, "wit.bin"))
    g = o.SplitMix64(4242); pk, vk, cs2 = o.setup(cs, [g.field() for _ in range(7)])
    pr = o.prove(cs2, z, pk, g.field(), g.field()); assert o.verify(vk, z[:cs.n_inputs], pr)
    z2 = z.copy(); z2[cs.n_inputs + 3, 0] ^= 1; assert not o.r1cs_is_satisfied(cs2, z2)
