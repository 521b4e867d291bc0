"""A slice of tools/fuzz_gpu.py under pytest: seeded random automata x kernel / chunk / slab / tile / compact / verify
options x both entry points, every result compared record for record with the oracle."""
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed0", [71000, 72000])
def test_gpu_fuzz_slice(seed0):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "60", str(seed0)], capture_output=True,
                         text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    last = [l for l in out.stdout.splitlines() if l.startswith("fuzz:")][-1]
    assert " 0 mismatches" in last, out.stdout[-3000:]
