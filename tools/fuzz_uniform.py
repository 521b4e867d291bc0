"""GPU fuzz of the uniform batch entry (acmi_match_batch_device_begin_uniform): random pattern sets, and per set a sequence of
batches -- shapes that repeat and change, blobs at shifting alignments, other text in the same shape, findAll on and off, two
batches in flight -- every result against the oracle.  What it is after: the item table a slot KEEPS between batches of one
shape.  Usage: python tools/fuzz_uniform.py [n_sequences] [seed0]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from gpu_util import resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, batch, synth

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
t0 = time.time()
bad = kept = 0
for q in range(n_seq):
    seed = seed0 + q
    r = np.random.default_rng(seed)
    alpha = [b"abcdef", b"ab", b"ACGT", None, bytes(range(97, 123))][int(r.integers(5))]
    lmin = int(r.choice([1, 4, 7, 8, 8, 12, 16, 16]))
    lmax = lmin + int(r.choice([0, 2, 9, 40]))
    n_pat = int(r.choice([3, 60, 700, 2048]))
    text_bytes = 1 << int(r.choice([16, 18, 20]))
    ds = synth._planted_ragged(f"U{seed}", seed, n_pat, lmin, lmax, 64, text_bytes // 64, text_bytes // 64, alpha, int(r.choice([0, 1, 4])))
    texts = [ds.hay_blob[:text_bytes].copy(), np.roll(ds.hay_blob[:text_bytes], int(r.integers(1, 5000))).copy()]
    texts.append(np.full(text_bytes, texts[0][0], np.uint8))
    d_texts = [torch.from_numpy(np.concatenate([np.zeros(64, np.uint8), t, np.zeros(64, np.uint8)])).cuda() for t in texts]
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    res = resource_for(ds, kernel=int(r.choice([0, 2, 2, 3])), chunk_bytes=int(r.choice([0, 0, 1024])))
    batch.bind_current_stream(res)
    shapes = [int(r.choice([64, 256, 1000, 4096, 8192, 16384])) for _ in range(2)]
    plan = []
    for _ in range(10):
        hl = shapes[0 if r.integers(4) else 1] if r.integers(8) else int(r.choice([128, 8192]))
        shift = int(r.choice([0, 0, 0, 0, 0, 0, 1, 16, 48]))
        nh = (text_bytes - 64) // hl
        if r.integers(10) == 0:
            nh = max(1, nh // 2)
        plan.append((int(r.integers(3)), hl, shift, nh, bool(r.integers(4))))
    want = []
    for t, hl, shift, nh, fa in plan:
        src = np.concatenate([np.zeros(64, np.uint8), texts[t], np.zeros(64, np.uint8)])[64 + shift: 64 + shift + hl * nh]
        want.append(o.match_batch(src, np.arange(nh + 1, dtype=np.uint64) * np.uint64(hl), fa))
    began = ended = 0
    while ended < len(plan):
        while began < len(plan) and began - ended < 2:
            t, hl, shift, nh, fa = plan[began]
            batch.match_device_begin_uniform(res, d_texts[t][64 + shift:], hl, nh, fa)
            began += 1
        total = batch.match_device_end(res)
        t, hl, shift, nh, fa = plan[ended]
        p, rr = batch.result_numpy(res, nh, total)
        if not (np.array_equal(p, want[ended][0]) and np.array_equal(rr, want[ended][1])):
            bad += 1
            print("MISMATCH", seed, ended, plan[ended], total, len(want[ended][1]), flush=True)
        ended += 1
    kept += int(res.stat("item_tables_kept"))
    api.ahocorasick_deinit(res)
    o.close()
    del d_texts
print(f"fuzz_uniform: {n_seq} sequences from seed {seed0}, {bad} mismatches, {kept} item tables kept, {time.time() - t0:.1f}s")
