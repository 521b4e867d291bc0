"""One cell of tests/test_gpu_cliffs.py's matrix, repeated: what the automatic mode settles on and how it got there.

    ACMI_CHOICE_TRACE=1 python tools/exp_choice.py cfg3 8 256 [repeats]

Every repeat starts from a forgotten choice (set_option kernel 3), follows the batch before it in the matrix (density one step
lower) as the test does, gives the handle SETTLE batches and times three."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from php_aho_corasick_amd import api, batch, synth
import test_gpu_cliffs as T

kind, per, hay_len = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 10
base = synth.density_set(kind)
pats = [{"id": k, "value": p} for k, p in enumerate(base.patterns())]
res = api.ahocorasick_init(pats)
api.ahocorasick_finalize(res)
batch.bind_current_stream(res)


def cell(per):
    ds = synth.plant_density(base, per, hay_len)
    reps = T.BATCH_BYTES // len(ds.hay_blob)
    nh = reps * ds.n_haystacks
    return torch.from_numpy(ds.hay_blob).cuda().repeat(reps).contiguous(), torch.arange(nh + 1, dtype=torch.int64, device="cuda") * hay_len


prev = T.DENSITIES[max(0, T.DENSITIES.index(per) - 1)]
for r in range(repeats):
    b0, o0 = cell(prev)
    res.set_option("kernel", 3)
    T._timed(torch, batch, res, b0, o0, T.SETTLE)
    del b0, o0
    time.sleep(0.3 * (r % 3))          # the host-side oracle and generator work between two cells of the test
    b1, o1 = cell(per)
    res.set_option("kernel", 3)
    print(f"--- repeat {r}", file=sys.stderr, flush=True)
    ms, total = T._timed(torch, batch, res, b1, o1, T.SETTLE)
    print(f"repeat {r}: automatic {ms:.4f} ms, kernel {int(res.stat('kernel_used'))}, probes {int(res.stat('choice_probes'))}, switches {int(res.stat('choice_switches'))}", flush=True)
    del b1, o1
