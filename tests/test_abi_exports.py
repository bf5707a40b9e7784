"""CPU-side check that the C-ABI library loads and exports every symbol the headers in include/ declare (no compute
calls: there is no GPU here), and that compute entry points fail loudly rather than fall back when no device exists."""
import ctypes, os, re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def declared_symbols(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S); src = re.sub(r"//[^\n]*", "", src)
    return sorted(set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", " ".join(l for l in src.splitlines() if not l.strip().startswith("#"))))
                  - {"defined", "sizeof"})

@pytest.fixture(scope="module")
def zkgpu():
    path = os.path.join(ROOT, "blockmaze_amd", "libzkgpu.so")
    if not os.path.exists(path):
        import __graft_entry__ as g; g.build()
    return ctypes.CDLL(path)

def test_engine_header_symbols_exported(zkgpu):
    syms = declared_symbols("zkgpu.h"); assert len(syms) >= 18
    for s in syms: assert hasattr(zkgpu, s), s

def test_batch_header_symbol_exported(zkgpu):
    assert declared_symbols("zk_batch.h") == ["verifyBatch"] and hasattr(zkgpu, "verifyBatch")

def test_no_cpu_fallback(zkgpu):
    import torch
    if torch.cuda.is_available(): pytest.skip("GPU present")
    zkgpu.zkgpu_last_error.restype = ctypes.c_char_p
    out = (ctypes.c_uint8 * 64)(); buf = (ctypes.c_uint8 * 64)()
    rc = zkgpu.zkgpu_msm_g1(buf, buf, ctypes.c_size_t(1), 0, 0, out)
    assert rc != 0 and b"no HIP device" in zkgpu.zkgpu_last_error()
    assert zkgpu.zkgpu_device_count() == 0

def test_domain_size_matches_oracle(zkgpu):
    from oracle import pyoracle as o
    zkgpu.zkgpu_domain_size.restype = ctypes.c_size_t; zkgpu.zkgpu_domain_size.argtypes = [ctypes.c_size_t]
    for m in [2, 3, 5, 16, 17, 24, 100, 167275, 252292, 503870, 1177046, 1 << 20]:
        assert zkgpu.zkgpu_domain_size(m) == o.domain_size(m), m

SYMS = {"zk_mint": ["genCMT", "computePRF", "genMintproof", "verifyMintproof"], "zk_redeem": ["genCMT", "computePRF", "genRedeemproof", "verifyRedeemproof"],
        "zk_send": ["genCMT", "genCMTS", "computePRF", "computeCRH", "genSendproof", "verifySendproof"], "zk_deposit": ["genCMT", "genCMTS", "computePRF", "genRoot", "genDepositproof", "verifyDepositproof"]}

def test_dropin_libraries_export_exactly_the_reference_symbols(zkgpu):
    """13 distinct symbols, 20 definitions (mint 4, send 6, deposit 6, redeem 4); headers in include/ declare the same sets"""
    import subprocess
    for lib, syms in SYMS.items():
        path = os.path.join(ROOT, "blockmaze_amd", "lib", "lib%s.so" % lib); out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
        defined = sorted(l.split()[-1] for l in out.splitlines() if " T " in l); assert defined == sorted(syms), lib
        assert sorted(set(declared_symbols(lib + ".h") + declared_symbols("zk_common.h"))) == sorted(syms), lib
        ctypes.CDLL(path)                                                    # loads (resolves libzkgpu.so through its rpath)
    for s in sorted(set(sum(SYMS.values(), []))): assert hasattr(zkgpu, s) and hasattr(zkgpu, "zkgpu_abi_" + s)
    for stub in ("libff.so", "libsnark.so"): assert os.path.exists(os.path.join(ROOT, "blockmaze_amd", "lib", stub))

@pytest.fixture(scope="module")
def driver(tmp_path_factory, zkgpu):
    import subprocess
    exe = str(tmp_path_factory.mktemp("drv") / "dropin_driver"); lib = os.path.join(ROOT, "blockmaze_amd", "lib")
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(ROOT, "tests", "dropin_driver.c"), "-L" + lib, "-lzk_mint", "-lzk_send", "-lzk_deposit", "-lzk_redeem", "-lff", "-lsnark", "-lpthread",
                           "-Wl,-rpath," + lib, "-Wl,-rpath-link," + os.path.join(ROOT, "blockmaze_amd")])     # the reference's cgo link line (zktx.go:4) minus gmp/stdc++
    return exe

def test_c_driver_merkle_roots_through_the_deposit_library(driver):
    """genRoot over 0, 1 and 16 leaves through the thin libzk_deposit.so exactly as the cgo side calls it (zktx.go:611 GenRT) — the SURVEY §8c goldens"""
    import subprocess
    out = dict(l.split() for l in subprocess.run([driver, "roots"], capture_output=True, text=True, check=True).stdout.splitlines())
    assert out["genRoot0"] == "8eb3c27b218349e6b9b6037b8042f3751ee820e8a0319a1bda439b247456088c"
    assert out["genRoot1"] == "a19a0d1fac447f65d273d5831827ccfa96c193a1b39618a23d11628d48e27a9e"
    assert out["genRoot16"] == "2630f036430a646118dbb95ba55e9e3803e35a680398d01f9942513ebbb7911e"

def test_c_driver_links_like_cgo_and_hashes_match(driver):
    import subprocess
    out = dict(l.split() for l in subprocess.run([driver, "hashes"], capture_output=True, text=True, check=True).stdout.splitlines())
    assert out["computePRF"] == "98a493490d506d579a6af5bd1d179e471de2be83252014f9db7c2a4c61b1472c" and out["genCMT0"] == "0044f0b699cd2d866c8da0201dcc2a8b28bdf7d47f39e13ebe4e53a29b704a83"
    assert out["computeCRH"] == "53d843629c72b6d8fff202285172ec85389122fa7c71251bea73832d3c4a7029" and out["genRoot0"] == "8eb3c27b218349e6b9b6037b8042f3751ee820e8a0319a1bda439b247456088c"
    import hashlib, struct
    sk = b"\x01" * 32; r = bytes(range(32)); pk = bytes.fromhex("00112233445566778899aabbccddeeff00112233")
    assert out["genCMTS"] == hashlib.sha256(struct.pack("<Q", 77) + pk[::-1] + r[::-1] + sk[::-1]).digest()[::-1].hex()   # = rev(SHA256(LE64(v) | rev(pk) | rev(r_s) | rev(sn))), SURVEY.md §8c
