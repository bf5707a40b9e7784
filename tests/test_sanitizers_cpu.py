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
