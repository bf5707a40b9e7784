"""Host verifier (r1cs_gg_ppzksnark_verifier_strong_IC on a prepared key; the path the verifyXproof symbols take) without a GPU:
  * the reference prover's proofs on the reference-made verification keys of tests/golden: accepted; every kind of tampering: rejected
  * the pairing itself against the reference's reduced_pairing VALUES (tests/golden/ref_vectors.txt): a verification key is crafted whose alpha_g1_beta_g2 is the
    reference's GT value for e(aG1, bG2) and whose other two pairings cancel (delta = -gamma, C = IC[0]); the proof (aG1, bG2, IC[0]) is accepted iff the
    engine's Miller loop + final exponentiation reproduce that value bit for bit
  * verifyBatch (include/zk_batch.h) is exported and fails cleanly without keys."""
import ctypes, json, os, subprocess, sys
import pytest
from oracle import pyoracle as o
from blockmaze_amd import engine as e
import verify_mutations as vm
from conftest import record_leg

HARNESS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_harness")

def H(x): return int(x, 16)

def g1_bytes(P):
    """compressed key-file encoding (alt_bn128_g1.cpp:404-418 under BINARY_OUTPUT / MONTGOMERY_OUTPUT): '0'|'1' is_zero, 32 bytes of Montgomery X (LE), '0'|'1' lsb of canonical Y"""
    if P is None: return b"1" + bytes(32) + b"1"
    return b"0" + o.to_mont(o.FQ, [P[0]])[0].to_bytes(32, "little") + (b"1" if P[1] & 1 else b"0")
def g2_bytes(Q):
    if Q is None: return b"1" + bytes(64) + b"1"
    (x0, x1), (y0, y1) = Q; m = o.to_mont(o.FQ, [x0, x1]); return b"0" + m[0].to_bytes(32, "little") + m[1].to_bytes(32, "little") + (b"1" if y0 & 1 else b"0")
def write_vk(path, gt12, gamma, delta, ic):
    """r1cs_gg_ppzksnark.tcc:100-108 + accumulation_vector.tcc:63-69"""
    n = len(ic) - 1; b = " ".join(str(c) for c in gt12).encode() + b"\n" + g2_bytes(gamma) + b"\n" + g2_bytes(delta) + b"\n" + g1_bytes(ic[0]) + b"\n"
    b += b"%d\n%d\n" % (n, n) + b"".join(b"%d\n" % i for i in range(n)) + b"%d\n" % n + b"".join(g1_bytes(p) + b"\n" for p in ic[1:]) + b"\n\n"
    open(path, "wb").write(b)
def proof_hex(A, B, C): return o.proof_hex(o.to_arr([A[0], A[1], B[0][0], B[0][1], B[1][0], B[1][1], C[0], C[1]]).reshape(-1))

@pytest.mark.parametrize("name", ["groth16_small", "groth16_step"])
def test_reference_proofs_on_reference_keys(golden_dir, name):
    d = os.path.join(golden_dir, name); meta = json.load(open(os.path.join(d, "meta.json"))); z = o.load_witness(os.path.join(d, "wit.bin")); vk = os.path.join(d, "vk.txt")
    inputs = o.from_arr(z[:meta["n_inputs"]]); proof = meta["proof"]; assert e.verify(vk, proof, inputs)
    for j in range(len(inputs)): bad = list(inputs); bad[j] = (bad[j] + 1) % o.R_MOD; assert not e.verify(vk, proof, bad)
    assert not e.verify(vk, proof, inputs[:-1]) and not e.verify(vk, proof, inputs + [0])                  # strong IC: the input count must match
    for k in range(8): pos = 64 * k + 21; assert not e.verify(vk, proof[:pos] + ("0" if proof[pos] != "0" else "1") + proof[pos + 1:], inputs)
    assert not e.verify(vk, "0" * 512, inputs) and not e.verify(vk, "zz" + proof[2:], inputs) and not e.verify(vk, "%064x" % o.Q_MOD + proof[64:], inputs)   # (0,0), non-hex, coordinate >= q
    ovk = o.parse_vk(vk); assert o.verify(ovk, inputs, o.proof_words_from_hex(proof))                      # and the oracle agrees

def test_pairing_values_of_the_reference(ref_vectors, tmp_path):
    G1, G2 = o.g1_gen(), o.g2_gen(); neg_g2 = (G2[0], ((o.Q_MOD - G2[1][0]) % o.Q_MOD, (o.Q_MOD - G2[1][1]) % o.Q_MOD)); n = 0; gts = []
    for l in ref_vectors:
        if l[0] != "pairing": continue
        a, b = H(l[1].split("=")[1]), H(l[2].split("=")[1]); gt = [int(l[3].split("=")[1])] + [int(x) for x in l[4:]]; assert len(gt) == 12; gts.append(gt)
        A, B = o.g1_op("mul", G1, k=a), o.g2_op("mul", G2, k=b); C = o.g1_op("mul", G1, k=77)
        vk = str(tmp_path / ("vk%d.txt" % n)); write_vk(vk, gt, G2, neg_g2, [C]); n += 1
        assert e.verify(vk, proof_hex(A, B, C), [])                                                        # e(A,B) * e(-C,gamma) * e(-C,-gamma) = e(A,B) == the reference's value
        assert not e.verify(vk, proof_hex(o.g1_op("dbl", A), B, C), [])
    assert n == 3
    vk = str(tmp_path / "vkx.txt"); write_vk(vk, gts[1], G2, neg_g2, [o.g1_op("mul", G1, k=77)])          # the GT value of another pair: rejected
    assert not e.verify(vk, proof_hex(G1, G2, o.g1_op("mul", G1, k=77)), [])

@pytest.mark.parametrize("name", ["groth16_small", "groth16_step"])
def test_gpu_verifier_schedule_on_the_host(golden_dir, name):
    """kernel K9's operation schedule (csrc/verify_sched.hpp: the whole pairing check as ~2,100 rounds of one field operation per lane) interpreted on the HOST, twice: on
    the host field type (what the program means) and on the kernel's own 29-bit limb arithmetic with every bound asserted (what the kernel does; a disagreement between
    the two raises).  The same decisions as the host verifier on the reference prover's proofs and on every kind of tampering — the schedule is proven right before a GPU
    ever runs it"""
    d = os.path.join(golden_dir, name); meta = json.load(open(os.path.join(d, "meta.json"))); z = o.load_witness(os.path.join(d, "wit.bin")); vk = os.path.join(d, "vk.txt")
    inputs = o.from_arr(z[:meta["n_inputs"]]); proof = meta["proof"]; ok, st = e.verify_schedule_on_host(vk, proof, inputs); assert ok and e.verify(vk, proof, inputs)
    assert 800 < st["rounds"] < 1000 and st["rounds"] <= st["mul_waves"] + st["lin8_waves"] + st["lin1_waves"] <= 4 * st["rounds"] and (st["slots"] + st["constants"]) * 48 <= 160 * 1024 and st["products"] > 20000, st      # four waves a round; fits the LDS of one CU
    for j in range(len(inputs)): bad = list(inputs); bad[j] = (bad[j] + 1) % o.R_MOD; assert not e.verify_schedule_on_host(vk, proof, bad)[0]
    assert not e.verify_schedule_on_host(vk, proof, inputs[:-1])[0]
    for k in range(8): pos = 64 * k + 21; assert not e.verify_schedule_on_host(vk, proof[:pos] + ("0" if proof[pos] != "0" else "1") + proof[pos + 1:], inputs)[0]      # every coordinate of A, B, C
    assert not e.verify_schedule_on_host(vk, "0" * 512, inputs)[0]

def test_gpu_verifier_schedule_reproduces_reference_pairing_values(ref_vectors, tmp_path):
    """the schedule's Miller loop and final exponentiation against the reference's reduced_pairing VALUES, as test_pairing_values_of_the_reference does for the host verifier"""
    G1, G2 = o.g1_gen(), o.g2_gen(); neg_g2 = (G2[0], ((o.Q_MOD - G2[1][0]) % o.Q_MOD, (o.Q_MOD - G2[1][1]) % o.Q_MOD)); n = 0
    for l in ref_vectors:
        if l[0] != "pairing": continue
        a, b = H(l[1].split("=")[1]), H(l[2].split("=")[1]); gt = [int(l[3].split("=")[1])] + [int(x) for x in l[4:]]
        A, B = o.g1_op("mul", G1, k=a), o.g2_op("mul", G2, k=b); C = o.g1_op("mul", G1, k=77); vk = str(tmp_path / ("vk%d.txt" % n)); write_vk(vk, gt, G2, neg_g2, [C]); n += 1
        assert e.verify_schedule_on_host(vk, proof_hex(A, B, C), [])[0] and not e.verify_schedule_on_host(vk, proof_hex(o.g1_op("dbl", A), B, C), [])[0]
    assert n == 3

# ---- the ACCEPT SET: the same verdict as the reference on every 512-character input (round 5) ----------------------------------------------------------------
@pytest.mark.parametrize("name", ["groth16_small", "groth16_step"])
def test_verdicts_are_the_reference_s_on_mutated_proofs(golden_dir, name):
    """tests/golden/verify_mutations_<fixture>.txt: ~290 mutations of the reference prover's proof with the verdict of the reference's own verifier behind
    sendcgo.cpp:388-448's parsing (oracle/make_golden.py, ref_harness verifymany).  The host verifier — what verify*proof runs when the device is busy or absent —, the
    oracle, and on every second case kernel K9's schedule (interpreted on the host on both arithmetics) decide every one of them the same way: aliases c + kq of
    accepted proofs ARE accepted (Fp_model(bigint) reduces, fp.tcc:190-194), as are (-A, -B) and re-randomisations; the six encodings on which an assert-enabled
    reference build aborts (B = (0, 0), verdict 2) are rejected, like the reference's NDEBUG build does"""
    d = os.path.join(golden_dir, name); vk = os.path.join(d, "vk.txt"); cases = vm.read_golden(os.path.join(golden_dir, "verify_mutations_%s.txt" % name)); ovk = o.parse_vk(vk)
    assert len(cases) >= 200 and sum(1 for c in cases if c[3] == 1) >= 100 and sum(1 for c in cases if c[3] == 0) >= 100 and sum(1 for c in cases if c[3] == vm.ABORT) == 6
    assert sum(1 for c in cases if c[3] == 1 and "alias" in c[0]) >= 70                                  # the point of the exercise
    for i, (label, h, ins, verdict) in enumerate(cases):
        assert vm.agrees(e.verify(vk, h, ins), verdict), (label, verdict)
        assert vm.agrees(o.verify(ovk, ins, o.proof_words_from_hex(h)), verdict), ("oracle", label, verdict)
        if i % 2 == 0: assert vm.agrees(e.verify_schedule_on_host(vk, h, ins)[0], verdict), ("K9 schedule", label, verdict)

@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not built (needs /root/reference at build time)")
@pytest.mark.parametrize("name", ["groth16_small", "groth16_step"])
def test_verdicts_against_the_live_reference(golden_dir, name, tmp_path):
    """the same comparison with FRESH mutations (another seed than the committed fixture's) decided by the reference binary on the spot"""
    import time
    d = os.path.join(golden_dir, name); meta = json.load(open(os.path.join(d, "meta.json"))); z = o.load_witness(os.path.join(d, "wit.bin")); vk = os.path.join(d, "vk.txt")
    cases = vm.cases(vk, meta["proof"], o.from_arr(z[:meta["n_inputs"]]), 0xA11A5 + len(name)); t0 = time.time(); ref = vm.reference_verdicts(HARNESS, vk, cases, tmp_path); record_leg("verifymany " + name, time.time() - t0)
    assert len(cases) >= 200 and ref.count(1) >= 100
    for (label, h, ins), verdict in zip(cases, ref): assert vm.agrees(e.verify(vk, h, ins), verdict), (label, verdict)

@pytest.mark.skipif(not os.path.exists(HARNESS + "_nd"), reason="oracle/_ref/ref_harness_nd not built")
def test_where_the_reference_aborts_its_ndebug_build_rejects(golden_dir, tmp_path):
    """verdict 2 of the fixtures: with B = (0, 0) the reference reaches assert(!is_zero()) in Fp::invert (fp.tcc:648) — it computes the pairing although
    is_well_formed() already failed — and the process dies; compiled with -DNDEBUG (libff's default build type) the very same call returns false, which is what this
    repo answers.  Every other case is decided alike by the two builds"""
    for name in ("groth16_small", "groth16_step"):
        cases = vm.read_golden(os.path.join(golden_dir, "verify_mutations_%s.txt" % name)); vk = os.path.join(golden_dir, name, "vk.txt")
        nd = vm.reference_verdicts(HARNESS + "_nd", vk, [c[:3] for c in cases], tmp_path)
        for c, v in zip(cases, nd): assert v == (0 if c[3] == vm.ABORT else c[3]), c[0]
        assert all("(0,0)" in c[0] and ("B" in c[0] or "all" in c[0]) for c in cases if c[3] == vm.ABORT)

def test_strict_encoding_is_an_opt_in_switch(golden_dir):
    """ZK_STRICT_PROOF_ENCODING=1 ("Not verbatim"): coordinates >= q are rejected instead of reduced; the default is the reference's behaviour"""
    d = os.path.join(golden_dir, "groth16_small"); cases = vm.read_golden(os.path.join(golden_dir, "verify_mutations_groth16_small.txt")); vk = os.path.join(d, "vk.txt")
    al = next(c for c in cases if c[0] == "alias all +1q"); can = cases[0]; assert al[3] == 1 and can[3] == 1
    code = "import sys; sys.path.insert(0, %r); from blockmaze_amd import engine as e; print(int(e.verify(%r, %r, %r)), int(e.verify(%r, %r, %r)))" % (os.path.dirname(os.path.dirname(golden_dir)), vk, can[1], can[2], vk, al[1], al[2])
    for env, want in (({}, "1 1"), ({"ZK_STRICT_PROOF_ENCODING": "1"}, "1 0"), ({"ZK_STRICT_PROOF_ENCODING": "0"}, "1 1")):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300); assert r.stdout.split("\n")[-2].strip() == want, (env, r.stdout, r.stderr)

def test_hex_arguments_parse_like_uint256S(golden_dir):
    """tests/golden/hex_blobs.txt: 130 C strings — short, long, odd length, mixed case, blanks, characters that are no hex digits in every position — with what the
    reference's uint256S / uint160S (send/uint256.h:222-248) make of them and the reference's Compute_PRF(x, 0) / Compute_CRH(x, 0) (send/util.h:233-258).  The cgo
    symbols computePRF / computeCRH / genCMT, fed the same strings, give the same hashes: every char* argument of the boundary goes through this parser"""
    from oracle import pyoracle  # noqa: F401
    import hashlib
    zk = e.Zk(); L = zk.L; n = 0
    for line in open(os.path.join(golden_dir, "hex_blobs.txt")):
        t = line.split(); s = b"" if t[0] == "-" else bytes.fromhex(t[0]); u256, u160, prf, crh = t[1:5]
        assert L.computePRF(s, b"").decode() == prf, s
        assert L.computeCRH(s, b"").decode() == crh, s
        blob = bytes.fromhex(u256)[::-1]; assert prf == hashlib.sha256(blob + bytes(32)).digest()[::-1].hex()            # (what the golden line itself says: the hash of the parsed blob)
        assert L.genCMT(ctypes.c_uint64(5), s, s).decode() == hashlib.sha256((5).to_bytes(8, "little") + blob + blob).digest()[::-1].hex(); n += 1
    assert n >= 120
@pytest.mark.skipif(not os.path.exists(HARNESS), reason="oracle/_ref/ref_harness not built")
def test_hex_arguments_against_the_live_reference(tmp_path):
    strings = vm.blob_strings(0xB10B5, n_random=300); ref = vm.reference_blobs(HARNESS, strings, tmp_path); L = e.Zk().L
    for s, (u256, u160, prf, crh) in zip(strings, ref): assert L.computePRF(s, b"").decode() == prf and L.computeCRH(s, b"").decode() == crh, s

def test_verify_batch_symbol(tmp_path, monkeypatch):
    zk = e.Zk(); assert hasattr(zk.L, "verifyBatch")
    monkeypatch.setenv("ZK_PRFKEY_DIR", str(tmp_path))                                                     # no key files there: no decision can be made
    rc, ok = zk.VerifyBatch([("send", "0" * 512, [bytes(32)] * 4, 0), (7, "0" * 512, [], 0)]); assert rc == -1 and ok == [False, False]
    assert zk.VerifyBatch([]) == (0, [])
