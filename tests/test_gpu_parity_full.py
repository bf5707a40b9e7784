"""GPU parity at the sizes BASELINE.json names, against the ORACLE (not only against properties or the product's own verifier):
  * one G1 multi-exponentiation of 2^18 full-width scalars and one witness-like (0 / 1 / small / full mix) MSM of 227,047 pairs vs liboracle's BDLO12
  * the witness map (R1CS rows + 7 transforms) of the exported send circuit on a real send witness vs liboracle's coefficient vector
  * every shipped switch (ZK_FOLD_C, ZK_H_LAGRANGE, ZK_MSM_PRECOMPUTE, ZK_MSM_H_TABLES, NTT radix / tile shape, submit / witness threads, one stream, the overflow hook): same proof bytes as the default
    configuration, on the golden fixtures (bytes of the reference prover) and on the full-size send key
  * BASELINE.json configs[2] / configs[4] shapes: the send circuit and the depth-32 deposit circuit cut into 8 shards (emulated on one GPU), and 64 send instances
    proved against one resident key and decided by the batched verifier."""
import json, os, subprocess, sys, time
import numpy as np
import pytest
from oracle import pyoracle as o
from blockmaze_amd import engine as e
import workload as w
from conftest import record_leg

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def hexargs(args): return [("0x" + a.hex()) if isinstance(a, bytes) else a for a in args]

@pytest.fixture(scope="module")
def send_keys(tmp_path_factory):
    d = tmp_path_factory.mktemp("prfKeyFull"); e.keygen("send", str(d / "sendpk.txt"), str(d / "sendvk.txt"), seed=0xB10C4A2E); return d

def test_msm_g1_at_h_query_size_matches_oracle():
    """n = 2^18 - 1 full-width scalars (the H query's shape) and n = 227,047 witness-like scalars (the A query's), both against the oracle's BDLO12 restatement"""
    n = (1 << 18) - 1; g = o.SplitMix64(0x4818); P = o.g1_consecutive(g.field(), n)
    K = np.zeros((n, 4), dtype=np.uint64); rng = np.random.default_rng(18); K[:] = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64); K[:, 3] &= np.uint64((1 << 61) - 1)
    t0 = time.time(); exp = o.msm_g1(P, K); t_o = time.time() - t0
    assert o.g1_from(e.msm(1, P, K))[0] == exp
    n2 = 227047; sel = rng.integers(0, 1000, size=n2); Z = K[:n2].copy(); Z[sel < 509] = 0; ones = (sel >= 509) & (sel < 967); Z[ones] = 0; Z[ones, 0] = 1
    small = (sel >= 967) & (sel < 995); Z[small, 1:] = 0; Z[small, 0] &= np.uint64(0xffffffff)                      # 50.9 % zero, 45.8 % one, 3.3 % other (mostly <= 32 bit)
    t0 = time.time(); exp2 = o.msm_g1(P[:n2], Z, mixed=True); t_o += time.time() - t0
    assert o.g1_from(e.msm(1, P[:n2], Z, filter_ones=True))[0] == exp2
    record_leg("oracle MSM at n = 2^18 and 227,047", t_o)

def test_msm_g2_at_b_query_size_matches_oracle():
    """the G2 half of the B query at its full size (136,316 pairs for send) with the witness mix, against the oracle's restatement of
    kc_multi_exp_with_mixed_addition (kc_multiexp.tcc:21-85) — at this size the fused witness path (k_wsort / k_wacc_quads / k_wtail over Fq2) runs on its fixed-base table"""
    n = 136316; g = o.SplitMix64(0xB2); P = o.g2_consecutive(g.field(), n); rng = np.random.default_rng(36)
    Z = np.zeros((n, 4), dtype=np.uint64); Z[:] = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n, 4), dtype=np.uint64); Z[:, 3] &= np.uint64((1 << 61) - 1)
    sel = rng.integers(0, 1000, size=n); Z[sel < 509] = 0; ones = (sel >= 509) & (sel < 967); Z[ones] = 0; Z[ones, 0] = 1; small = (sel >= 967) & (sel < 995); Z[small, 1:] = 0; Z[small, 0] &= np.uint64(0xffffffff)
    t0 = time.time(); exp = o.msm_g2(P, Z, mixed=True); t_o = time.time() - t0
    assert o.g2_from(e.msm(2, P, Z, filter_ones=True))[0] == exp
    record_leg("oracle G2 MSM at n = 136,316", t_o)

def test_witness_map_of_the_send_circuit_matches_oracle(tmp_path):
    """all m + 1 = 262,145 coefficients of H for a real send witness: device rows + transforms + pointwise step vs the oracle's r1cs_to_qap_witness_map restatement"""
    rp = str(tmp_path / "send.bin"); e.circuit_export("send", rp); cs = o.R1CS.load(rp); assert cs.domain_m == 262144
    d = w.send_instance(7); wp = str(tmp_path / "w.bin"); e.witness_send(*hexargs(w.send_args(d)), wp); z = o.load_witness(wp)
    dev = e.R1cs(cs.n_inputs, cs.n_vars, cs.n_cons, cs.rowptr, cs.col, cs.coeff); got = dev.witness_map(z); dev.close()
    t0 = time.time(); exp = o.witness_map(cs, z); record_leg("oracle witness map of the send circuit", time.time() - t0)
    assert got.shape == exp.shape == (262145, 4) and np.array_equal(got, exp)

SWITCHES = [{"ZK_MERGE_EQUAL_COLUMNS": "0"}, {"ZK_FOLD_C": "0"}, {"ZK_H_LAGRANGE": "0"}, {"ZK_MSM_PRECOMPUTE": "0"}, {"ZK_MSM_H_TABLES": "0"}, {"ZK_NTT_RADIX_LOG": "1"}, {"ZK_NTT_RADIX_LOG": "3"}, {"ZK_NTT_LOGC": "1"},
            {"ZK_NTT_LOGC": "2", "ZK_NTT_RADIX_LOG": "3"}, {"ZK_SUBMIT_THREADS": "0"}, {"ZK_MSM_ONE_STREAM": "1"}, {"ZK_WITNESS_THREADS": "0"}, {"ZK_WITNESS_DENSE": "1"}, {"ZK_MSM_H_RUN": "16"}, {"ZK_SCAN_THREADS": "1"}, {"ZK_MSM_PRECOMPUTE_MAX_MB": "300"}, {"ZK_PRIO": "off"}, {"ZK_PRIO": "hacc:3,wit:0,ntt:1"}, {"ZK_MSM_TABLES_FREE_SHARE": "0.000001"}, {"ZK_MSM_DIRECT_CAP": "1"}]   # (the last one: the test hook that forces every one-pass sort into its overflow fallback, i.e. the general MSM path)
PROVE_CODE = """
import json, os, sys
sys.path.insert(0, %r)
from blockmaze_amd import engine as e
import numpy as np
def rd(p):
    b = open(p, 'rb').read(); n = int(np.frombuffer(b, dtype=np.uint64, count=1)[0]); return np.frombuffer(b, dtype=np.uint64, count=4 * n, offset=8).reshape(n, 4).copy()
out = []
for pk, wit, r, s in json.loads(sys.argv[1]):
    p = e.Prover(pk); z = rd(wit); out.append(p.prove(z, int(r, 16), int(s, 16))); out.append(p.prove(z, int(r, 16), int(s, 16))); p.close()
print('PROOFS ' + json.dumps(out))
"""

@pytest.mark.parametrize("env", SWITCHES, ids=lambda d: ",".join("%s=%s" % kv for kv in d.items()))
def test_every_shipped_switch_gives_the_same_proof_bytes(env, golden_dir, send_keys, tmp_path):
    """a fresh process per configuration (the switches are read once): golden fixtures -> the reference prover's bytes; full-size send key -> the default configuration's bytes"""
    jobs = []; exp = []
    for name in ("groth16_small", "groth16_step"):
        d = os.path.join(golden_dir, name); meta = json.load(open(os.path.join(d, "meta.json"))); jobs.append([os.path.join(d, "pk.txt"), os.path.join(d, "wit.bin"), meta["r"], meta["s"]]); exp += [meta["proof"]] * 2
    wp = str(tmp_path / "w.bin"); dd = w.send_instance(11); e.witness_send(*hexargs(w.send_args(dd)), wp); pk_path = str(send_keys / "sendpk.txt")
    p = e.Prover(pk_path); ref = p.prove(o.load_witness(wp), 0x1234567, 0x89abcde); p.close(); jobs.append([pk_path, wp, "1234567", "89abcde"]); exp += [ref] * 2
    assert e.verify(str(send_keys / "sendvk.txt"), ref, w.pack_public([dd["cmtA_old"], dd["sn_old"], dd["cmtS"], dd["cmtA"]]))
    r = subprocess.run([sys.executable, "-c", PROVE_CODE % ROOT, json.dumps(jobs)], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("PROOFS ")]; assert line, r.stderr[-2000:]
    assert json.loads(line[0][7:]) == exp

def shard_proof(pk_path, z, world, r, s):
    recs = []
    for rank in range(world):
        p = e.Prover(pk_path, rank, world); p.set_witness(z); recs.append(p.prove_partial())
        if rank + 1 < world: p.close()
    out = p.finish(recs, r, s); p.close(); return out

def test_send_cut_into_8_shards(send_keys, tmp_path):
    """BASELINE.json configs[2]'s partition: every query of the send key in 8 contiguous ranges (ranks emulated one after the other on this GPU), 384-byte partial
    records added on the host: byte-identical to the unsharded proof"""
    pk_path = str(send_keys / "sendpk.txt"); d = w.send_instance(21); wp = str(tmp_path / "w.bin"); e.witness_send(*hexargs(w.send_args(d)), wp); z = o.load_witness(wp)
    full = e.Prover(pk_path); exp = full.prove(z, 77, 99); full.close()
    assert shard_proof(pk_path, z, 8, 77, 99) == exp and e.verify(str(send_keys / "sendvk.txt"), exp, w.pack_public([d["cmtA_old"], d["sn_old"], d["cmtS"], d["cmtA"]]))

def test_deposit_depth32_cut_into_8_shards(tmp_path):
    """BASELINE.json configs[4] in its sharded form: the depth-32 deposit key (H query of 1,179,648 Lagrange points) in 8 ranges"""
    pk_path, vk_path, wp = str(tmp_path / "pk.txt"), str(tmp_path / "vk.txt"), str(tmp_path / "w.bin"); e.keygen("deposit", pk_path, vk_path, seed=33, tree_depth=32)
    dd = w.deposit_instance(4); rt, _ = w.merkle_root_and_path(dd["leaves"], dd["index"], depth=32)
    e.witness_deposit(*hexargs(w.deposit_args(dd)), "".join("0x" + l.hex() for l in dd["leaves"]), len(dd["leaves"]), "0x" + dd["sk"].hex(), wp, tree_depth=32); z = o.load_witness(wp)
    full = e.Prover(pk_path); exp = full.prove(z, 5, 7); full.close()
    assert shard_proof(pk_path, z, 8, 5, 7) == exp
    assert e.verify(vk_path, exp, w.pack_public([rt, dd["pk_recv"], dd["cmtB_old"], dd["sn_old"], dd["cmtB"], dd["sn_s"]]))

def test_send_batch_of_64_instances(send_keys, tmp_path):
    """BASELINE.json configs[2]: 64 seeded send instances against one resident key — through the batch entry point (one call, B witnesses) and one at a time: identical
    bytes per proof for fixed (r_i, s_i); all 64 accepted by the batched GPU verifier and each rejected under its neighbour's public inputs"""
    pk_path, vk_path = str(send_keys / "sendpk.txt"), str(send_keys / "sendvk.txt"); p = e.Prover(pk_path); wp = str(tmp_path / "w.bin"); zs, ins = [], []
    for i in range(64):
        d = w.send_instance(300 + i); e.witness_send(*hexargs(w.send_args(d)), wp); zs.append(o.load_witness(wp)); ins.append(w.pack_public([d["cmtA_old"], d["sn_old"], d["cmtS"], d["cmtA"]]))
    rs = [(1000 + i, 2000 + i) for i in range(64)]
    one_by_one = [p.prove(zs[i], *rs[i]) for i in range(64)]
    if hasattr(p, "prove_batch"):
        assert p.prove_batch(zs, rs) == one_by_one
        assert p.prove_batch(zs[:5], rs[:5]) == one_by_one[:5] and p.prove_batch(zs[63:], rs[63:]) == one_by_one[63:]            # ragged batch sizes
        bad = list(zs[:4]); bad[2] = zs[2].copy(); bad[2][5000, 0] ^= 1
        with pytest.raises(e.ZkGpuError): p.prove_batch(bad, rs[:4])                                                              # one unsatisfying assignment fails the call, like prove()
    p.close()
    assert e.verify_batch(vk_path, one_by_one, ins) == [True] * 64
    assert e.verify_batch(vk_path, one_by_one, ins[1:] + ins[:1]) == [False] * 64
