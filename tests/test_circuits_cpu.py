"""CPU-side checks of the host logic: note hashing through the drop-in C-ABI against the golden vectors of SURVEY.md §8c
(captured from the reference's compiled libzk_send.so / libzk_deposit.so) and hashlib; circuit shape and witness of the
send / mint / redeem circuits against the reference's measured counts and golden primary inputs; the SHA-256 gadget
against libsnark's own gadget (canonical R1CS and witness hashes committed in tests/golden/sha256_gadget.json)."""
import ctypes, hashlib, json, os, subprocess
import numpy as np
import pytest
from oracle import pyoracle as o
from blockmaze_amd import engine as e
import workload as w

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")

@pytest.fixture(scope="module")
def zk(): return e.Zk()

def test_hash_golden_vectors(zk):
    sk = b"\x01" * 32; r = bytes(range(32)); pk = bytes.fromhex("00112233445566778899aabbccddeeff00112233"); z32 = bytes(32)
    assert zk.ComputePRF(sk, r).hex() == "98a493490d506d579a6af5bd1d179e471de2be83252014f9db7c2a4c61b1472c"
    assert zk.GenCMT(0, z32, z32).hex() == "0044f0b699cd2d866c8da0201dcc2a8b28bdf7d47f39e13ebe4e53a29b704a83"
    assert zk.ComputeCRH(pk, r).hex() == "53d843629c72b6d8fff202285172ec85389122fa7c71251bea73832d3c4a7029"
    assert zk.GenRT([]).hex() == "8eb3c27b218349e6b9b6037b8042f3751ee820e8a0319a1bda439b247456088c"
    assert zk.GenRT([bytes(31) + b"\x01"]).hex() == "a19a0d1fac447f65d273d5831827ccfa96c193a1b39618a23d11628d48e27a9e"
    # zero-extension of short inputs (uint256.h:222-248): 20 bytes of ff parse as the low 20 bytes
    L = zk.L; assert L.computePRF(b"0x" + b"ff" * 20, b"0").decode() == "bfb25c62b014a150278a49cf64b934d71f8a0caabb8d51b417fcfe00bed61fcb"
    # deposit/main.cpp:154-167: 16 leaves 0x1..0x9, cmtS, 0x11..0x16
    cmtS = bytes.fromhex("e4593e968e75e96fd5c51212cadd046547226e98c1715de9a10e6dfa3e9fdca5")
    leaves = [int(h, 16).to_bytes(32, "big") for h in "1 2 3 4 5 6 7 8 9".split()] + [cmtS] + [int(h, 16).to_bytes(32, "big") for h in "11 12 13 14 15 16".split()]
    assert zk.GenRT(leaves).hex() == "2630f036430a646118dbb95ba55e9e3803e35a680398d01f9942513ebbb7911e"
    assert zk.GenRT(leaves[:10]).hex() == "6b3ab57816ea4d7bb6410b4a81484d1b64d08e33cc29160a789eb7a94c75f267"

def test_hashes_match_hashlib(zk):
    for i in range(8):
        d = w.send_instance(i)
        assert zk.ComputePRF(d["sk"], d["r"]) == d["sn"] and zk.ComputeCRH(d["pk_sender"], d["r"]) == d["r_s"]
        assert zk.GenCMT(d["value"], d["sn"], d["r"]) == d["cmtA"] and zk.GenCMTS(d["value_s"], d["pk_recv"], d["r_s"], d["sn_old"]) == d["cmtS"]

def test_reference_send_fixture_hashes():
    d = w.reference_send_fixture()   # golden values captured from the reference
    assert d["sn_old"].hex() == "4a31770fe5354a1a9632ebe1481e108cd82ce514ac094c57b5ffdfaea8ac138a" and d["cmtA_old"].hex() == "036bdbabf553bd57e41289b8c13b80a9aef29c464a2cfca0c9e99c69fb9be4ff"
    assert d["sn"].hex() == "9b2d319b594f146c785aac11592a3e59f32cd6e6176a4d7a52d15a0b8eb6db06" and d["cmtA"].hex() == "589effbb69ee8401c4108ad1e9a3de34e0fc81e283c6da9e95e6b9555dcb2835"
    assert d["r_s"].hex() == "8fe3dac1d2c00b427c4406d3fdfe43df999e7c8cc70dd9bf57953c2a9622e1b9" and d["cmtS"].hex() == "e4b1743ea76c314992849f07fd9d8352d63e669773862df96305a4fd924cdbb4"

def canonical_hash(cs):
    """order-independent digest of an R1CS: per row the sorted (variable, coefficient) pairs with duplicates merged and zeros dropped"""
    h = hashlib.sha256()
    for m in range(3):
        vals = o.from_arr(cs.coeff[m]) if len(cs.coeff[m]) else []
        for i in range(cs.n_cons):
            d = {}
            for k in range(int(cs.rowptr[m][i]), int(cs.rowptr[m][i + 1])): d[int(cs.col[m][k])] = (d.get(int(cs.col[m][k]), 0) + vals[k]) % o.R_MOD
            h.update(repr(sorted((c, v) for c, v in d.items() if v)).encode())
    return h.hexdigest()

def test_sha256_gadget_matches_libsnark(tmp_path, golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "sha256_gadget.json")))
    p = str(tmp_path / "sha.bin"); e.circuit_export("sha256", p); cs = o.R1CS.load(p)
    assert (cs.n_cons, cs.n_vars) == (gold["seed0"]["constraints"], gold["seed0"]["variables"]) == (27280, 25560)
    assert canonical_hash(cs) == gold["canonical_r1cs_sha256"]
    for seed in (0, 1, 2):
        if seed == 0:   # libsnark's own KAT block (test_sha256_gadget.cpp:29-31)
            words = [0x426bc2d8, 0x4dc86782, 0x81e8957a, 0x409ec148, 0xe6cffbe8, 0xafe6ba4f, 0x9c6f1978, 0xdd7af7e9, 0x038cce42, 0xabd366b8, 0x3ede7e00, 0x9130de53, 0x72cdf73d, 0xee825114, 0x8cb48d1b, 0x9af68ad0]
            blk = b"".join(x.to_bytes(4, "big") for x in words)
        else:
            g = o.SplitMix64(seed); lb, rb = [], []
            for _ in range(256): lb.append(g.next() & 1); rb.append(g.next() & 1)
            pk = lambda bits: bytes(sum(bits[8 * i + j] << (7 - j) for j in range(8)) for i in range(32)); blk = pk(lb) + pk(rb)
        wp = str(tmp_path / ("w%d.bin" % seed)); e.witness_sha256(blk[:32], blk[32:], wp)
        assert hashlib.sha256(open(wp, "rb").read()).hexdigest() == gold["seed%d" % seed]["witness_sha256"]          # bit-identical to libsnark's assignment
        z = o.load_witness(wp); assert o.r1cs_is_satisfied(cs, z)
        digest_bits = "".join(str(int(v)) for v in o.from_arr(z[512:768])); assert digest_bits == gold["seed%d" % seed]["digest_bits"]

@pytest.fixture(scope="module")
def send_cs(tmp_path_factory):
    p = str(tmp_path_factory.mktemp("send") / "send.bin"); e.circuit_export("send", p); return o.R1CS.load(p)

def test_send_circuit_shape_and_fixture_witness(send_cs, tmp_path):
    cs = send_cs; assert (cs.n_inputs, cs.n_vars, cs.n_cons) == (5, 227046, 252286) and cs.domain_m == 262144
    assert cs.swap_ab_beneficial()                                                                                           # the reference's pk stores A/B swapped
    d = w.reference_send_fixture(); wp = str(tmp_path / "w.bin"); e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp)
    z = o.load_witness(wp); assert z.shape == (227046, 4) and o.r1cs_is_satisfied(cs, z)
    assert o.from_arr(z[:5]) == [379622515294215686324177489188214168996770001985522676859086851673897379839, 9229067119290276653789333223926290153177678652679541537576131534578011488910,
                                  1490550998170437254261127660234999754805136150295310288437901687192055892820, 9045781962496940631450103114965940036070457011305611729179959305608247203919, 423]
    assert o.from_arr(z[:5]) == w.pack_public([d["cmtA_old"], d["sn_old"], d["cmtS"], d["cmtA"]])
    vals = z[:, 0].astype(object); small = (z[:, 1:] == 0).all(axis=1)
    zeros = int(((z == 0).all(axis=1)).sum()); ones = int((small & (z[:, 0] == 1)).sum()); assert (zeros, ones + 1, len(z) - zeros - ones) == (115509, 103963, 7575)   # BASELINE.md §2.2 (ones incl. the constant)
    assert int(o.from_arr(z[1030:1031])[0]) == 0                                                                              # ZERO

def test_send_witness_seeded_and_negative(send_cs, tmp_path):
    for i in (0, 1):
        d = w.send_instance(i); wp = str(tmp_path / "w.bin"); e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(d)], wp)
        z = o.load_witness(wp); assert o.r1cs_is_satisfied(send_cs, z) and o.from_arr(z[:5]) == w.pack_public([d["cmtA_old"], d["sn_old"], d["cmtS"], d["cmtA"]])
    bad = dict(d); bad["value"] = d["value"] + 1; e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(bad)], wp); assert not o.r1cs_is_satisfied(send_cs, o.load_witness(wp))   # value != value_old - value_s
    bad = dict(d); bad["sk"] = bytes(31) + b"\x02"; e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(bad)], wp); assert not o.r1cs_is_satisfied(send_cs, o.load_witness(wp))   # wrong_sk (send/main.cpp:144-148)
    bad = dict(d); bad["value_s"] = d["value_old"] + 5; bad["value"] = 0; e.witness_send(*[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.send_args(bad)], wp); assert not o.r1cs_is_satisfied(send_cs, o.load_witness(wp))   # value_s > value_old

@pytest.mark.parametrize("kind,shape", [("mint", (4, 151512, 167270)), ("redeem", (4, 151579, 167853))])
def test_mint_redeem_circuits(kind, shape, tmp_path):
    p = str(tmp_path / "c.bin"); e.circuit_export(kind, p); cs = o.R1CS.load(p); assert (cs.n_inputs, cs.n_vars, cs.n_cons) == shape and cs.domain_m == 196608   # step domain 2^17 + 2^16
    d = w.mint_instance(3, redeem=(kind == "redeem")); wp = str(tmp_path / "w.bin"); e.witness_mint_redeem(kind == "redeem", *[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.mint_args(d)], wp)
    z = o.load_witness(wp); assert o.r1cs_is_satisfied(cs, z) and o.from_arr(z[:4]) == w.pack_public([d["cmtA_old"], d["sn_old"], d["cmtA"]], d["value_s"])
    bad = dict(d); bad["value_s"] += 1; e.witness_mint_redeem(kind == "redeem", *[("0x" + a.hex()) if isinstance(a, bytes) else a for a in w.mint_args(bad)], wp); assert not o.r1cs_is_satisfied(cs, o.load_witness(wp))

def test_merkle_gadget_matches_libsnark(tmp_path, golden_dir):
    gold = json.load(open(os.path.join(golden_dir, "merkle_gadget.json")))
    for depth in (2, 8):
        gd = gold["depth%d" % depth]; p = str(tmp_path / "m.bin"); e.circuit_export("merkle", p, tree_depth=depth); cs = o.R1CS.load(p)
        assert (cs.n_cons, cs.n_vars) == (gd["constraints"], gd["variables"])
        if depth == 2: assert canonical_hash(cs) == gd["canonical_r1cs_sha256"]
        # the harness' seeded instance (oracle/ref_harness.cpp cmd_merklegadget): leaf bits, then per level bottom-up: side bit, sibling bits
        g = o.SplitMix64(gd["seed"]); pk = lambda bits: bytes(sum(bits[8 * i + j] << (7 - j) for j in range(8)) for i in range(32))
        leaf = pk([g.next() & 1 for _ in range(256)]); sibs, pos = [], 0
        for d in range(depth):
            if g.next() & 1: pos |= 1 << d
            sibs.append(pk([g.next() & 1 for _ in range(256)]))
        assert pos == gd["address"]; wp = str(tmp_path / "mw.bin"); e.witness_merkle(depth, leaf, sibs, pos, wp)
        assert hashlib.sha256(open(wp, "rb").read()).hexdigest() == gd["witness_sha256"]                                      # bit-identical to libsnark's assignment
        assert o.r1cs_is_satisfied(cs, o.load_witness(wp))

def test_deposit_circuit(tmp_path):
    p = str(tmp_path / "d.bin"); e.circuit_export("deposit", p); cs = o.R1CS.load(p); assert (cs.n_inputs, cs.n_vars, cs.n_cons, cs.domain_m) == (6, 457127, 503863, 524288)
    H = lambda b: "0x" + b.hex()
    for d in (w.reference_deposit_fixture(), w.deposit_instance(1)):
        arr = "".join(H(x) for x in d["leaves"]); wp = str(tmp_path / "dw.bin")
        e.witness_deposit(*[H(a) if isinstance(a, bytes) else a for a in w.deposit_args(d)], arr, len(d["leaves"]), H(d["sk"]), wp)
        z = o.load_witness(wp); assert o.r1cs_is_satisfied(cs, z)
        assert o.from_arr(z[:6]) == w.pack_public([d["rt"], d["pk_recv"], d["cmtB_old"], d["sn_old"], d["cmtB"], d["sn_s"]])
    d = w.reference_deposit_fixture()
    assert d["cmtS"].hex() == "e4593e968e75e96fd5c51212cadd046547226e98c1715de9a10e6dfa3e9fdca5" and d["rt"].hex() == "2630f036430a646118dbb95ba55e9e3803e35a680398d01f9942513ebbb7911e"
    bad = dict(d); bad["leaves"] = list(d["leaves"]); bad["leaves"][3] = bytes(32)                                              # same leaf, different tree: root no longer matches the claimed rt? (rt is recomputed, so still valid)
    bad = dict(d); bad["sn_s"] = w.prf(d["sk"], (345).to_bytes(32, "big"))                                                       # wrong_sn_s (deposit/main.cpp:201-216)
    arr = "".join(H(x) for x in d["leaves"]); e.witness_deposit(*[H(a) if isinstance(a, bytes) else a for a in w.deposit_args(bad)], arr, 16, H(d["sk"]), wp); assert not o.r1cs_is_satisfied(cs, o.load_witness(wp))
