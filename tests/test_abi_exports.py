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
