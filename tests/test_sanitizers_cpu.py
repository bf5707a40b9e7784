"""The host side of libzkgpu (hex readers, key-file and container readers, circuits and witness generators, host verifier, batch entry) under AddressSanitizer +
UndefinedBehaviorSanitizer, driven through the C-ABI with the malformed inputs the reference mishandles (tests/sanitize_driver.cpp).  No GPU involved: the HIP
objects are linked as they are and every device path reports "no HIP device"."""
import os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def test_host_code_under_asan_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "blockmaze_amd", "csrc"); subprocess.check_call(["make", "-s", "-C", csrc, "-j8", "sanitize"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"); env.pop("LD_PRELOAD", None)   # (the drop-in surface leaks by contract: returned strings belong to the caller)
    r = subprocess.run([os.path.join(csrc, "build", "san", "host_san_driver"), os.path.join(ROOT, "tests", "golden"), str(tmp_path)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "SANITIZE OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr

def test_witness_generators_under_tsan(tmp_path):
    """the host code that runs on several threads — the witness generators' one wave of compressions on the task pool's helpers, from three callers at once, and the host verifier with its cache of prepared keys from three threads — under
    ThreadSanitizer (tests/tsan_driver.cpp; no report, and every assignment equal to the first one made for its statement).  The same driver built on the sources before
    round 6's fix of the doubly written compression outputs reports 56 races at Board::set_bit / eval_bit: it sees what the soak found"""
    csrc = os.path.join(ROOT, "blockmaze_amd", "csrc"); subprocess.check_call(["make", "-s", "-C", csrc, "-j8", "tsan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"); env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(csrc, "build", "tsan", "host_tsan_driver"), str(tmp_path), "3", "2", os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "TSAN OK" in r.stdout and "ThreadSanitizer" not in r.stderr, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
