"""Cliff detector for the automatic kernel choice: a matrix of pattern set x match density x haystack size.  In
every cell the automatic mode (a measured choice: engine.cpp choose_strategy / learn_strategy) and every forced scan kernel
must produce the same record array -- checked against the oracle on a slice, and against each other in full -- and, once it
has settled, the automatic mode may cost at most 1.3 x what the best forced kernel costs.  The reference has no density
regime: one callback per event whatever their number (src/php_ahocorasick.c:542-589).  Needs an MI355X: -m gpu.

    python tests/test_gpu_cliffs.py [headline|words|cfg3 ...]      prints the matrix (ms per call, kernel chosen)"""
import os
import sys
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

DENSITIES = (1, 8, 32, 128, 512)          # planted occurrences per 8 KiB of text
HAY_SIZES = (256, 8192, 1 << 20)
BATCH_BYTES = 64 << 20
SETTLE = 10                                # batches the automatic mode gets before it is timed
KERNELS = (1, 2, 4, 6)


def _timed(torch, batch, res, d_blob, d_off, warm, n=3):
    for _ in range(warm):
        total = batch.match_device(res, d_blob, d_off, True)
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total = batch.match_device(res, d_blob, d_off, True)
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, total


def run_matrix(kind, densities=DENSITIES, hay_sizes=HAY_SIZES, verbose=False, oracle=True):
    import torch
    from oracle.pyoracle import Oracle
    from php_aho_corasick_amd import api, batch, synth
    base = synth.density_set(kind)
    pats = [{"id": k, "value": p} for k, p in enumerate(base.patterns())]
    # (oracle=False -- bench.py's leg for the 65,536-pattern set, whose CPU trie takes seconds to build: the kernels are compared
    # with each other only, four independent walks; this test always checks against the oracle)
    o = Oracle("port") if oracle else None
    if o:
        o.add_patterns((base.pat_blob, base.pat_off))
    handles = {}
    for kern in (3,) + KERNELS:
        res = api.ahocorasick_init(pats)
        res.set_option("kernel", kern)
        api.ahocorasick_finalize(res)
        if kern == 2 and not res.stat("anchor_q"):
            api.ahocorasick_deinit(res)
            continue
        batch.bind_current_stream(res)
        handles[kern] = res
    rows, bad = [], []
    for hay_len in hay_sizes:
        for per in densities:
            ds = synth.plant_density(base, per, hay_len)
            reps = BATCH_BYTES // len(ds.hay_blob)
            nh = reps * ds.n_haystacks
            d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(reps).contiguous()
            d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * hay_len
            n_chk = max(1, min(ds.n_haystacks, (128 << 10) // hay_len))
            pe, re_ = o.match_batch(ds.hay_blob[:int(ds.hay_off[n_chk])], ds.hay_off[:n_chk + 1], True) if o else (None, None)
            ref = None
            cell = {}
            for kern, res in handles.items():
                if kern == 3:
                    res.set_option("kernel", 3)   # (forgets what the cell before taught the handle: every cell starts from the seed)
                ms, total = _timed(torch, batch, res, d_blob, d_off, SETTLE if kern == 3 else 2)
                used = int(res.stat("kernel_used"))
                if kern in (2, 6) and used != kern:   # not eligible for this batch (kernel 6: patterns too long; ...): another kernel ran
                    continue
                rec = batch.result_view(res, total)
                if ref is None:
                    ref = rec.clone()
                    p, r = batch.result_numpy(res, nh, total)
                    ok = int(p[-1]) == total and (o is None or (np.array_equal(p[:n_chk + 1], pe) and np.array_equal(r[:int(pe[-1])], re_)))
                    if not ok:
                        bad.append(f"{kind} {per}/8KiB hay {hay_len}: kernel {kern} differs from the oracle")
                elif total != ref.shape[0] or not torch.equal(rec, ref):
                    bad.append(f"{kind} {per}/8KiB hay {hay_len}: kernel {kern} differs from kernel {list(handles)[0]}")
                cell[kern] = (ms, used)
            forced = {k: v[0] for k, v in cell.items() if k != 3}
            best_k = min(forced, key=forced.get)
            auto_ms, auto_used = cell[3]
            algo = BATCH_BYTES + 16 * ref.shape[0] + 8 * nh
            rows.append(dict(set=kind, per_8k=per, hay_len=hay_len, records=int(ref.shape[0]), auto_ms=round(auto_ms, 4), auto_kernel=auto_used,
                             best_forced_ms=round(forced[best_k], 4), best_forced_kernel=best_k, forced_ms={k: round(v, 4) for k, v in forced.items()},
                             frac_auto=round(algo / auto_ms / 1e6 / 8000.0, 4), probes=int(handles[3].stat("choice_probes"))))
            if auto_ms > 1.3 * forced[best_k] + 0.03:
                bad.append(f"{kind} {per}/8KiB hay {hay_len}: automatic {auto_ms:.3f} ms (kernel {auto_used}) > 1.3 x kernel {best_k} {forced[best_k]:.3f} ms")
            if verbose:
                print(rows[-1], flush=True)
            del d_blob, d_off, ref
    for res in handles.values():
        api.ahocorasick_deinit(res)
    if o:
        o.close()
    torch.cuda.empty_cache()
    return rows, bad


@pytest.mark.parametrize("kind", ["headline", "words", "cfg3"])
def test_no_cliff_in_the_automatic_choice(kind):
    rows, bad = run_matrix(kind)
    assert not bad, "\n".join(bad)
    assert len(rows) == len(DENSITIES) * len(HAY_SIZES)


if __name__ == "__main__":
    for kind in sys.argv[1:] or ["headline", "words", "cfg3"]:
        rows, bad = run_matrix(kind, verbose=True)
        print("\n".join(["CLIFFS:"] + bad) if bad else f"{kind}: no cliff", flush=True)
