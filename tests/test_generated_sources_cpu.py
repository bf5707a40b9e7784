"""The generated sources under blockmaze_amd/csrc are what their generators produce today: gen_field29.py (Fq29 / Fr29 on nine 29-bit limbs and the constants of the
verifier's linear-combination pipeline — the generator also runs its own checks on the way: the column schedules against big-number arithmetic, the interval bounds of the
mixed and the general addition, the Barrett-like step at its worst-case inputs), gen_field_mul.py (the 8 x 32-bit product, squaring and carry chains) and gen_params.py."""
import os, shutil, subprocess, sys
import pytest
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); CSRC = os.path.join(ROOT, "blockmaze_amd", "csrc")

@pytest.mark.parametrize("gen,outputs", [("gen_field29.py", ["field29_gfx950.inc", "field29_params.h"]), ("gen_field_mul.py", ["field_mul_gfx950.inc"]), ("gen_params.py", ["field_params.h"])])
def test_generator_reproduces_the_committed_file(gen, outputs, tmp_path):
    shutil.copy(os.path.join(CSRC, gen), tmp_path / gen)                       # the generators write next to themselves
    r = subprocess.run([sys.executable, str(tmp_path / gen)], capture_output=True, text=True, timeout=900); assert r.returncode == 0, r.stderr[-2000:]
    for o in outputs: assert open(tmp_path / o).read() == open(os.path.join(CSRC, o)).read(), "%s is not what %s generates" % (o, gen)
