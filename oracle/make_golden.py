#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — regenerate tests/golden/* from the REAL reference.

Runs oracle/_ref/ref_harness (the reference's own libff/libfqfft/libsnark code compiled by `make -C oracle ref`; only
possible in the build container, where /root/reference exists) and stores its OUTPUTS as fixtures:

  ref_vectors.txt            field / curve / domain / MSM / pairing known answers
  groth16_small/             a 60-constraint R1CS + witness (inputs made by tests/r1cs_util.py, seed 7), the key pair
                             the reference generator produced for it (pk.txt / vk.txt in the reference's on-disk
                             format), fixed (r, s), the H coefficients and the proof bytes of the reference prover
  groth16_step/              same on a step-radix-2 domain (40 constraints + 4 inputs + 1 -> m = 48)
  sha256_gadget.json         constraint / variable counts, digest and SHA-256 of the R1CS / witness dumps of libsnark's
                             sha256_two_to_one_hash_gadget on the reference's own KAT input and on seeded inputs
  merkle_gadget.json         same for merkle_tree_check_read_gadget (depth 2 and 8)
  lesscmp_gadget.json        BlockMaze's less_comparison_gadget block (send/circuit/comparison.tcc compiled for real): canonical R1CS
                             hash and the witness digests for seven (value_old, value_s) pairs
  hash_blocks.json           the CMTS (736 bits), PRF (512 bits) and CRH (416 bits, one block) gadgets composed like commitment.tcc:100-320
  cmta_gadget.json           two chained compression gadgets with hard-wired padding, composed like sha256_CMTA_gadget (commitment.tcc:12-110)
  note_hashes.txt            Note::cm / NoteS::cm / Compute_PRF / Compute_CRH of send/Note.h and util.h on seeded hex strings

  unpacker_gadget.json       libsnark's multipacking_gadget as the circuits build their public-input unpacker (832 / 1024 / 1440 bits)
  verify_mutations_<fixture>.txt  the reference verifier's VERDICT (ref_harness verifymany = r1cs_gg_ppzksnark_verifier_strong_IC behind sendcgo.cpp:388-448's
                             hex parsing) on ~290 seeded mutations of each key fixture's proof (tests/verify_mutations.py): aliases c + kq, special values,
                             off-curve, off-subgroup, malleations, re-randomisations, other statements.  Verdict 2 = the reference process aborts
  hex_blobs.txt              uint256S / uint160S (send/uint256.h:222-248) and Compute_PRF / Compute_CRH of the parsed blob for ~150 odd C strings

Fixtures are data only; no reference source is copied.
"""
import hashlib, json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as o
from r1cs_util import random_r1cs
HARNESS = os.path.join(HERE, "_ref", "ref_harness"); GOLD = os.path.join(ROOT, "tests", "golden")

def run(*args):
    r = subprocess.run([HARNESS, *args], capture_output=True, text=True)
    if r.returncode != 0: raise RuntimeError(r.stdout + r.stderr)
    return r.stdout

def sha(path): return hashlib.sha256(open(path, "rb").read()).hexdigest()

def groth16_fixture(name, seed, ni, nv, nc, rs_seed):
    d = os.path.join(GOLD, name); os.makedirs(d, exist_ok=True)
    cs, z = random_r1cs(seed, ni, nv, nc); assert o.r1cs_is_satisfied(cs, z)
    cs.save(os.path.join(d, "r1cs.bin")); o.save_witness(os.path.join(d, "wit.bin"), z)
    g = o.SplitMix64(rs_seed); r, s = g.field(), g.field()
    out = run("e2e", os.path.join(d, "r1cs.bin"), os.path.join(d, "wit.bin"), "%x" % r, "%x" % s, d)
    proof = [l.split()[1] for l in out.splitlines() if l.startswith("proof ")][0]
    assert "verify 1" in out and "verify_hex_roundtrip 1" in out
    json.dump({"seed": seed, "n_inputs": ni, "n_vars": nv, "n_cons": nc, "domain_m": cs.domain_m, "r": "%x" % r, "s": "%x" % s, "proof": proof},
              open(os.path.join(d, "meta.json"), "w"), indent=1)

def gadget_fixture():
    res = {}
    with tempfile.TemporaryDirectory() as t:
        for seed in (0, 1, 2):
            out = run("sha256gadget", t + "/r.bin", t + "/w.bin", str(seed)); kv = dict(p.split("=") for p in out.split()[1:])
            if seed == 0:
                from test_circuits_cpu import canonical_hash
                res["canonical_r1cs_sha256"] = canonical_hash(o.R1CS.load(t + "/r.bin"))   # term-order independent digest, see tests/test_circuits_cpu.py
            res["seed%d" % seed] = {"constraints": int(kv["constraints"]), "variables": int(kv["variables"]), "digest_bits": kv["digest"], "r1cs_sha256": sha(t + "/r.bin"), "witness_sha256": sha(t + "/w.bin")}
        json.dump(res, open(os.path.join(GOLD, "sha256_gadget.json"), "w"), indent=1)
        res = {}
        for depth, seed in ((2, 5), (8, 6)):
            out = run("merklegadget", str(depth), t + "/r.bin", t + "/w.bin", str(seed)); kv = dict(p.split("=") for p in out.split()[1:])
            from test_circuits_cpu import canonical_hash
            res["depth%d" % depth] = {"canonical_r1cs_sha256": canonical_hash(o.R1CS.load(t + "/r.bin")), "seed": seed, "constraints": int(kv["constraints"]), "variables": int(kv["variables"]), "address": int(kv["address"]), "r1cs_sha256": sha(t + "/r.bin"), "witness_sha256": sha(t + "/w.bin")}
        json.dump(res, open(os.path.join(GOLD, "merkle_gadget.json"), "w"), indent=1)

LESSCMP_PAIRS = [(22, 8), (8, 8), (2 ** 64 - 1, 0), (5, 9), (0, 0), (1 << 63, (1 << 63) - 1), (0x0123456789abcdef, 0x00ffeeddccbbaa99)]
def blockmaze_fixture():
    """BlockMaze's own sources compiled for real (the ones that need no boost): comparison.tcc's gadget block and the host note hashes"""
    from test_circuits_cpu import canonical_hash
    res = {"pairs": []}
    with tempfile.TemporaryDirectory() as t:
        for vo, vs in LESSCMP_PAIRS:
            out = run("lesscmp", str(vo), str(vs), t + "/r.bin", t + "/w.bin"); kv = dict(p.split("=") for p in out.split()[1:])
            res["canonical_r1cs_sha256"] = canonical_hash(o.R1CS.load(t + "/r.bin")); res["constraints"] = int(kv["constraints"]); res["variables"] = int(kv["variables"])
            res["pairs"].append({"value_old": vo, "value_s": vs, "satisfied": int(kv["satisfied"]), "witness_sha256": sha(t + "/w.bin")})
    json.dump(res, open(os.path.join(GOLD, "lesscmp_gadget.json"), "w"), indent=1)
    open(os.path.join(GOLD, "note_hashes.txt"), "w").write(run("notehashes", "20241002", "12"))
    res = {}
    with tempfile.TemporaryDirectory() as t:
        for seed in (3, 4):
            out = run("cmta", str(seed), t + "/r.bin", t + "/w.bin"); kv = dict(p.split("=") for p in out.split()[1:])
            res["canonical_r1cs_sha256"] = canonical_hash(o.R1CS.load(t + "/r.bin")); res["constraints"] = int(kv["constraints"]); res["variables"] = int(kv["variables"])
            res["terms"] = [int(x) for x in kv["terms"].split(",")]; res["zero_coefficient_terms"] = [int(x) for x in kv["zero_terms"].split(",")]     # libsnark keeps terms with coefficient 0 (e.g. `c = 0`, IV bits that are 0)
            res["seed%d" % seed] = {"digest_bits": kv["digest"], "witness_sha256": sha(t + "/w.bin")}
        run("sha256gadget", t + "/r.bin", t + "/w.bin", "0"); cs = o.R1CS.load(t + "/r.bin")
        res["two_to_one_zero_coefficient_terms"] = [int((cs.coeff[m] == 0).all(axis=1).sum()) for m in range(3)]
    json.dump(res, open(os.path.join(GOLD, "cmta_gadget.json"), "w"), indent=1)
    res = {}                                                                           # the CMTS / PRF / CRH blocks composed like commitment.tcc:100-320 (ref_harness hashblock)
    with tempfile.TemporaryDirectory() as t:
        for kind in ("cmts", "prf", "crh"):
            r = {}
            for seed in (3, 4):
                out = run("hashblock", kind, str(seed), t + "/r.bin", t + "/w.bin"); kv = dict(p.split("=") for p in out.split()[1:])
                r["canonical_r1cs_sha256"] = canonical_hash(o.R1CS.load(t + "/r.bin")); r["constraints"] = int(kv["constraints"]); r["variables"] = int(kv["variables"])
                r["terms"] = [int(x) for x in kv["terms"].split(",")]; r["zero_coefficient_terms"] = [int(x) for x in kv["zero_terms"].split(",")]
                r["seed%d" % seed] = {"digest_bits": kv["digest"], "witness_sha256": sha(t + "/w.bin")}
            res[kind] = r
    json.dump(res, open(os.path.join(GOLD, "hash_blocks.json"), "w"), indent=1)

def unpacker_fixture():
    """libsnark's multipacking_gadget as the four circuits build their public-input unpacker: 832 bits (mint / redeem), 1024 (send), 1440 (deposit)"""
    from test_circuits_cpu import canonical_hash
    res = {}
    with tempfile.TemporaryDirectory() as t:
        for nbits in (832, 1024, 1440):
            r = {}
            for seed in (5, 6):
                out = run("unpacker", str(nbits), str(seed), t + "/r.bin", t + "/w.bin"); kv = dict(p.split("=") for p in out.split()[1:]); assert kv["satisfied"] == "1"
                r.update(canonical_r1cs_sha256=canonical_hash(o.R1CS.load(t + "/r.bin")), constraints=int(kv["constraints"]), variables=int(kv["variables"]), inputs=int(kv["inputs"])); r["seed%d" % seed] = {"witness_sha256": sha(t + "/w.bin")}
            res["bits%d" % nbits] = r
    json.dump(res, open(os.path.join(GOLD, "unpacker_gadget.json"), "w"), indent=1)

def verdict_fixture():
    import verify_mutations as vm
    with tempfile.TemporaryDirectory() as t:
        for name, seed in (("groth16_small", 0x5EED0001), ("groth16_step", 0x5EED0002)):
            d = os.path.join(GOLD, name); meta = json.load(open(os.path.join(d, "meta.json"))); z = o.load_witness(os.path.join(d, "wit.bin")); vk = os.path.join(d, "vk.txt")
            cs = vm.cases(vk, meta["proof"], o.from_arr(z[:meta["n_inputs"]]), seed); v = vm.reference_verdicts(HARNESS, vk, cs, t)
            assert v[0] == 1 and sum(1 for x in v if x == 1) > 100 and sum(1 for x in v if x == 0) > 100
            vm.write_cases(os.path.join(GOLD, "verify_mutations_%s.txt" % name), cs, v); print(name, len(cs), "cases:", v.count(1), "accepted,", v.count(0), "rejected,", v.count(2), "abort the reference")
        strings = vm.blob_strings(0x5EED0003); res = vm.reference_blobs(HARNESS, strings, t)
        with open(os.path.join(GOLD, "hex_blobs.txt"), "w") as f:
            for x, r in zip(strings, res): f.write((x.hex() or "-") + " " + " ".join(r) + "\n")
        print("hex_blobs:", len(strings), "strings")

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if "--verdicts-only" in sys.argv: verdict_fixture(); sys.exit(0)
    if "--unpacker-only" in sys.argv: unpacker_fixture(); sys.exit(0)
    if "--gadgets-only" not in sys.argv:      # the key fixtures come from the reference generator's std::random_device: regenerating them changes pk/vk/proof (consistently)
        run("vectors", os.path.join(GOLD, "ref_vectors.txt"))
        groth16_fixture("groth16_small", 7, 3, 40, 60, 99)
        groth16_fixture("groth16_step", 8, 4, 30, 40, 100)
    gadget_fixture(); blockmaze_fixture(); unpacker_fixture(); verdict_fixture()
    print("golden fixtures written to", GOLD)
