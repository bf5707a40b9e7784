"""blockmaze_amd — MI355X-native Groth16 prover behind BlockMaze's libzk_*.so C-ABI.

The product is the shared library blockmaze_amd/libzkgpu.so (HIP kernels for gfx950 + host engine + C-ABI, sources in
blockmaze_amd/csrc, headers in include/).  This Python package only binds it for tests and the benchmark.
"""
from . import engine  # noqa: F401
