"""The host threads of libzkgpu under ThreadSanitizer WITH the device at work (tests/tsan_driver.cpp, GPU section): the cgo symbols of the send circuit from three
threads (provers of a pool, their submit threads, the hand-over of a circuit board read in place, the verifier's combiner), then host-buffer proofs beside proofs from
a stash on two prover objects of one key, a batch of eight proofs in one call and the batch verifier from two threads.  (Run by hand with ZK_SHARD_DEVICES=2 as well: every
proof cut over two shard provers and their threads — no report either.)  The ROCm runtime is not instrumented (tests/tsan_gpu.supp); ASLR is switched off for the run (TSan's shadow does not fit this
kernel's randomised layout)."""
import os, shutil, subprocess
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

def test_cgo_symbols_and_engine_calls_under_tsan(tmp_path):
    csrc = os.path.join(ROOT, "blockmaze_amd", "csrc")
    if not os.path.exists(os.path.join(csrc, "build", "tsan", "host_tsan_driver")): subprocess.check_call(["make", "-s", "-C", csrc, "-j8", "tsan"], stdout=subprocess.DEVNULL)   # (normally prebuilt: __graft_entry__.build())
    if not shutil.which("setarch"): pytest.skip("no setarch on this box")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66 suppressions=" + os.path.join(ROOT, "tests", "tsan_gpu.supp")); env.pop("LD_PRELOAD", None)
    r = subprocess.run(["setarch", "x86_64", "-R", os.path.join(csrc, "build", "tsan", "host_tsan_driver"), str(tmp_path), "3", "1", os.path.join(ROOT, "tests", "golden"), "gpu"], capture_output=True, text=True, env=env, timeout=1200)
    if "unexpected memory mapping" in r.stderr: pytest.skip("ThreadSanitizer cannot map its shadow on this kernel")
    assert r.returncode == 0 and "TSAN OK" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.returncode, r.stdout[-600:], r.stderr[-3000:])
    assert "36 send proofs made and accepted" in r.stdout and "10 from a stash" in r.stdout and "(16 accepted)" in r.stdout
