"""A slice of tools/fuzz_gpu.py under pytest: seeded random automata x kernel / chunk / slab / tile / compact / verify
options x both entry points, every result compared record for record with the oracle."""
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def _fuzz(n, seed0, big=False):
    env = dict(os.environ)
    env.pop("FUZZ_BIG", None)
    if big:
        env["FUZZ_BIG"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), str(n), str(seed0)], capture_output=True,
                         text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    last = [l for l in out.stdout.splitlines() if l.startswith("fuzz:")][-1]
    assert " 0 mismatches" in last, out.stdout[-3000:]


@pytest.mark.parametrize("seed0", [71000, 72000])
def test_gpu_fuzz_slice(seed0):
    """2 x 160 cases (round 5 ran 2 x 60: a 600-case run of the builder's had found a record-placement bug that slice missed)."""
    _fuzz(160, seed0)


def test_gpu_fuzz_big_slice():
    """FUZZ_BIG: batches of a few MiB -- many tiles per wave, events by the thousand -- through the kernels that place records
    with running sums (kernel 6) and the automatic choice."""
    _fuzz(40, 73000, big=True)
