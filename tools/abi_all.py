"""The four gen*proof symbols from one caller (p50 / mean / p10 / p90 of 100 calls each) and the count of MSMs repeated on the general path.  python tools/abi_all.py  (ZK_TRACE_TIMES=1: the per-call breakdown on stderr)"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from blockmaze_amd import engine as e
import workload as w
tmp = tempfile.mkdtemp()
for kind in ("send", "mint", "redeem", "deposit"): e.keygen(kind, os.path.join(tmp, kind + "pk.txt"), os.path.join(tmp, kind + "vk.txt"), seed=0xB10C4A2E + len(kind))
os.environ["ZK_PRFKEY_DIR"] = tmp; zk = e.Zk()
def bench(name, fn, check):
    for i in range(30): p = fn(i)
    assert check(p, 29)
    ts = []
    for i in range(100): t0 = time.perf_counter(); fn(i); ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort(); print("%s one caller: mean %.4f p50 %.4f p10 %.4f p90 %.4f ms" % (name, sum(ts) / len(ts), ts[50], ts[10], ts[90]), file=sys.stderr)
ms = [w.mint_instance(i) for i in range(8)]; rs = [w.mint_instance(i, redeem=True) for i in range(8)]; ss = [w.send_instance(i) for i in range(8)]; ds = [w.deposit_instance(i) for i in range(4)]
bench("genSendproof", lambda i: zk.GenSendProof(*w.send_args(ss[i % 8])), lambda p, i: zk.VerifySendProof(p, ss[i % 8]["cmtA_old"], ss[i % 8]["sn_old"], ss[i % 8]["cmtS"], ss[i % 8]["cmtA"]))
bench("genMintproof", lambda i: zk.GenMintProof(*w.mint_args(ms[i % 8])), lambda p, i: zk.VerifyMintProof(p, ms[i % 8]["cmtA_old"], ms[i % 8]["sn_old"], ms[i % 8]["cmtA"], ms[i % 8]["value_s"]))
bench("genRedeemproof", lambda i: zk.GenRedeemProof(*w.mint_args(rs[i % 8])), lambda p, i: zk.VerifyRedeemProof(p, rs[i % 8]["cmtA_old"], rs[i % 8]["sn_old"], rs[i % 8]["cmtA"], rs[i % 8]["value_s"]))
bench("genDepositproof", lambda i: zk.GenDepositProof(*w.deposit_args(ds[i % 4]), ds[i % 4]["leaves"], ds[i % 4]["rt"], ds[i % 4]["sk"]), lambda p, i: zk.VerifyDepositProof(p, ds[i % 4]["rt"], ds[i % 4]["pk_recv"], ds[i % 4]["cmtB_old"], ds[i % 4]["sn_old"], ds[i % 4]["cmtB"], ds[i % 4]["sn_s"]))
print("MSMs repeated on the general path: %d" % e.general_path_repeats(), file=sys.stderr)
