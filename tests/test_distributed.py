"""Multi-GPU path on CPU: world_size-2 `gloo` run of the sharding + all-gatherv layer
(php_aho_corasick_amd.batch).  The per-rank scan itself needs a GPU, so here each rank's local
match array is produced by the oracle (test infrastructure standing in for the device scan); what
is under test is the contiguous sharding, the haystack-id rebasing and the variable-size gather,
which must reproduce the single-process result exactly and in order."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from helpers import ROOT
from php_aho_corasick_amd import batch


def test_shard_range_is_a_contiguous_partition():
    for n in (0, 1, 7, 8, 9, 1000, 524288):
        for world in (1, 2, 3, 8):
            spans = [batch.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    from oracle.pyoracle import Oracle
    from php_aho_corasick_amd import batch, synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gatherers = dict()
    for name in ("R1", "G5", "R2"):
        ds = synth.NAMED[name]()
        o = Oracle("port"); o.add_patterns((ds.pat_blob, ds.pat_off))
        for find_all in (True, False):
            gp, gr = o.match_batch(ds.hay_blob, ds.hay_off, find_all)          # single-process truth
            lo, hi = batch.shard_range(ds.n_haystacks, rank, world)
            off = ds.hay_off[lo:hi + 1] - ds.hay_off[lo]
            blob = ds.hay_blob[int(ds.hay_off[lo]):int(ds.hay_off[hi])]
            lp, lr = o.match_batch(blob, off, find_all)                         # this rank's shard, local ids
            counts = torch.from_numpy(np.diff(lp.astype(np.int64)))
            recs = torch.from_numpy(lr.astype(np.int32).reshape(-1, 4))
            for mode in ("padded", "padded-async", "gatherer-small", "gatherer"):
                if mode == "padded-async":
                    st = batch.allgatherv_start(counts, recs, lo)
                    prefix, records = batch.allgatherv_finish(st)
                elif mode == "padded":
                    prefix, records = batch.allgatherv_records(counts, recs, lo)
                else:  # fixed-capacity single collective; capacity 4 forces the grow-and-redo path
                    g = gatherers.setdefault(mode, batch.RecordGatherer(None, torch.device("cpu"), 4 if mode.endswith("small") else 1 << 15))
                    s1 = g.start(recs, hi - lo, lo)          # two batches in flight, as in bench.py
                    s2 = g.start(recs, hi - lo, lo)
                    prefix, records = g.finish(s1)
                    p2, r2 = g.finish(s2)
                    assert torch.equal(prefix, p2) and torch.equal(records, r2)
                assert np.array_equal(prefix.numpy().astype(np.uint64), gp), (name, find_all, rank, mode)
                assert np.array_equal(records.numpy().astype(np.uint32), gr), (name, find_all, rank, mode)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


@pytest.mark.parametrize("world", [2, 3])
def test_allgatherv_world(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out = "TIMEOUT"
        outs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all(f"rank {r} ok" in outs[r] for r in range(world))
