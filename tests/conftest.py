import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")

def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")

@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

@pytest.fixture(scope="session")
def ref_vectors():
    """tests/golden/ref_vectors.txt (outputs of the real reference, see oracle/make_golden.py) as token lists."""
    return [l.split() for l in open(os.path.join(GOLDEN, "ref_vectors.txt"))]

# Which optional legs of the parity tests really ran (e.g. the real libsnark prover on the full-size send key): tests add to LEGS, the summary is printed at the very
# end of the run so that it shows in the tail of `pytest -q` that the driver records.
LEGS = {}
def record_leg(name, seconds=None): LEGS[name] = seconds
def pytest_terminal_summary(terminalreporter):
    if LEGS: terminalreporter.write_line("reference legs run: " + ", ".join("%s%s" % (k, "" if v is None else " (%.1f s)" % v) for k, v in sorted(LEGS.items())))
