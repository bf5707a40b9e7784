#!/usr/bin/env python3
"""A/B timing of prover configurations on ONE box in ONE run (boxes differ by 10-20 %, so numbers from different runs do not compare).
    python tools/ab_bench.py [--circuit send] [--proofs 40] "" "ZK_MSM_NO_HSORT=1" "ZK_NTT_RADIX_LOG=3,ZK_NTT_LOGC=2" ...
Each variant runs in a fresh process (switches are read once) on the same key file: ms per proof with the witness resident and as a host buffer, the per-stage HIP-event
times, and the proof bytes for fixed (r, s), which must be the same for every variant."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

def child(circuit, pk, wit, n):
    import numpy as np
    from blockmaze_amd import engine as e
    b = open(wit, "rb").read(); k = int(np.frombuffer(b, dtype=np.uint64, count=1)[0]); z = np.frombuffer(b, dtype=np.uint64, count=4 * k, offset=8).reshape(k, 4).copy()
    p = e.Prover(pk); proof = p.prove(z, 0x1234, 0x5678); p.set_witness(z)
    for _ in range(5): p.prove_resident()
    t0 = time.perf_counter()
    for _ in range(n): p.prove_resident()
    ms_res = 1e3 * (time.perf_counter() - t0) / n; t0 = time.perf_counter()
    for _ in range(n): p.prove(z)
    ms_host = 1e3 * (time.perf_counter() - t0) / n
    e.profile_enable(True)
    for _ in range(10): p.prove_resident()
    st = e.profile_report(); e.profile_enable(False); p.close()
    print("RESULT " + json.dumps({"proof": proof, "ms_resident": round(ms_res, 4), "ms_host_buffer": round(ms_host, 4), "stages": {k: round(v["ms_total"] / 10, 4) for k, v in sorted(st.items())}}))

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child": child(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])); sys.exit(0)
    import argparse
    ap = argparse.ArgumentParser(); ap.add_argument("--circuit", default="send"); ap.add_argument("--proofs", type=int, default=40); ap.add_argument("--rounds", type=int, default=2); ap.add_argument("variants", nargs="*", default=[""])
    a = ap.parse_args()
    from blockmaze_amd import engine as e
    import workload as w
    hx = lambda args: [("0x" + x.hex()) if isinstance(x, bytes) else x for x in args]
    tmp = tempfile.mkdtemp(); pk, vk, wit = os.path.join(tmp, "pk.txt"), os.path.join(tmp, "vk.txt"), os.path.join(tmp, "w.bin"); depth = 8
    kind = a.circuit
    if kind.startswith("deposit"): depth = int(kind[7:] or 8); kind = "deposit"
    e.keygen(kind, pk, vk, seed=7, tree_depth=depth)
    if kind == "send": e.witness_send(*hx(w.send_args(w.send_instance(1))), wit)
    elif kind in ("mint", "redeem"): e.witness_mint_redeem(kind == "redeem", *hx(w.mint_args(w.mint_instance(1, redeem=(kind == "redeem")))), wit)
    else: d = w.deposit_instance(1); e.witness_deposit(*hx(w.deposit_args(d)), "".join("0x" + l.hex() for l in d["leaves"]), len(d["leaves"]), "0x" + d["sk"].hex(), wit, tree_depth=depth)
    proofs = set(); rows = {}
    for rnd in range(a.rounds):                                                    # interleaved rounds: drift of the box shows up as a difference between rounds, not between variants
        for v in a.variants:
            env = dict(os.environ); env.update(dict(kv.split("=") for kv in v.split(",") if kv))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind, pk, wit, str(a.proofs)], capture_output=True, text=True, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line: print("variant %r FAILED\n%s" % (v, r.stderr[-1500:])); continue
            j = json.loads(line[0][7:]); proofs.add(j["proof"]); rows.setdefault(v, []).append(j)
    for v, js in rows.items():
        print("== %-40s resident %s ms | host buffer %s ms" % (v or "(default)", " / ".join("%.3f" % j["ms_resident"] for j in js), " / ".join("%.3f" % j["ms_host_buffer"] for j in js)))
        print("   " + "  ".join("%s %.3f" % (k, x) for k, x in js[-1]["stages"].items()))
    print("proof bytes identical across variants:", len(proofs) == 1)
