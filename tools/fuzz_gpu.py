"""GPU fuzz: many seeded random automata / ragged batches through every kernel and entry point,
each compared record-for-record with the oracle.  Usage: python tools/fuzz_gpu.py [n_cases] [seed0]
FUZZ_BIG=1: batches of a few MiB (thousands of haystacks, or long ones) -- many tiles per wave, events by the thousand -- through
the kernels that place records with running sums (kernel 6: k_walk_chain / k_walk_tiles) and the automatic choice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from gpu_util import device_match, host_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
alphabets = [b"ab", b"abc", b"abcdef", b"ACGT", b"01", b"abcdefgh", b"abcdefghij", b"xyz\x00\x01\x02\x03\x04\x05", None,
             bytes(range(32, 127))]
t0 = time.time()
bad = 0
for case in range(n_cases):
    seed = seed0 + case
    r = np.random.default_rng(seed)
    alpha = alphabets[r.integers(len(alphabets))]
    lmin = int(r.choice([1, 1, 2, 4, 4, 5, 6, 7, 8, 11, 15, 19, 40]))   # 4..6: the dense anchor form when forced or selective
    lmax = int(min(1024, lmin + r.choice([0, 1, 3, 8, 20, 100, 900])))
    n_pat = int(r.choice([1, 3, 17, 60, 300, 1500, 4000]))   # 4000 x 4 offsets: the big-set anchor tables
    n_hay = int(r.choice([1, 2, 9, 64, 65, 200]))
    hmax = int(r.choice([0, 1, 15, 16, 17, 127, 128, 129, 1000, 5000, 20000]))
    plants = int(r.choice([0, 1, 3, 10]))
    if os.environ.get("FUZZ_BIG"):
        n_hay = int(r.choice([300, 2000, 6000]))
        hmax = int(r.choice([300, 3000, 9000, 70000]))
        while n_hay * hmax > (12 << 20):
            n_hay //= 2
        n_pat = int(r.choice([3, 17, 60, 300, 1500, 4000]))
        lmax = int(min(200, lmax))
    ds = synth._planted_ragged(f"F{seed}", seed, n_pat, lmin, lmax, n_hay, 0, hmax, alpha, plants) if plants else \
        synth.ragged(f"F{seed}", seed, n_pat, lmin, lmax, n_hay, 0, hmax, alpha)
    if lmin >= 7 and r.integers(3) == 0:   # a few short patterns among the long ones: the split set
        extra = [bytes(r.choice(np.frombuffer(alpha or bytes(range(256)), dtype=np.uint8), size=int(r.integers(1, 7))).tolist()) for _ in range(int(r.integers(1, 6)))]
        pats_all = ds.patterns()
        for e in extra:
            pats_all.insert(int(r.integers(0, len(pats_all) + 1)), e)
        pb, po = synth.pack(pats_all)
        ds = synth.Dataset(ds.name, pb, po, ds.hay_blob, ds.hay_off)
    o = Oracle("port")
    calls = int(r.choice([1, 1, 2, 5]))
    pats = ds.patterns()
    step = (len(pats) + calls - 1) // calls
    for lo in range(0, len(pats), step):
        o.add_patterns(pats[lo:lo + step])
    opts = dict(kernel=int(r.choice([6, 6, 6, 3, 3, 2, 4] if os.environ.get("FUZZ_BIG") else [0, 1, 2, 2, 3, 3, 4, 6, 6])), tile_w=int(r.choice([0, 0, 16, 32, 64, 128])), tile_p=int(r.choice([0, 0, 1, 3, 7, 16, 64])), walk_group=int(r.choice([1, 4, 8])), ilp=int(r.choice([1, 2])), group_quads=int(r.choice([4, 8])),
                chunk_bytes=int(r.choice([0, 0, 16, 48, 256, 1024, 4096])), slab_events=int(r.choice([1, 2, 8])),
                run_cap=int(r.choice([0, 1, 2, 8])), relist=int(r.choice([0, 1, 1])), persistent=int(r.choice([-1, 0, 1])), lds_bytes=int(r.choice([2048, 8192, 65536, 163000])),
                small_path=int(r.choice([0, 1])), tiny_rows=int(r.choice([0, 1, 1])), short_rows=int(r.choice([-1, 0, 1])), refill=int(r.choice([0, 1, 2, 2])), max_flat_outputs=int(r.choice([0, 1 << 25, 1 << 25])), compact=int(r.choice([-1, 0, 1, 1])), early_stop=int(r.choice([0, 1, 2, 2])), verify=int(r.choice([0, 1, 1])), dense_bytes=int(r.choice([0, 4096, 1 << 22])), anchor_dense=int(r.choice([-1, 1, 1, 0])), dense_hot=int(r.choice([-1, 0, 1])), split=int(r.choice([-1, -1, 0, 1])), anchor_slots2=int(r.choice([-1, 0, 1])),
                win_table=int(r.choice([-1, 0, 1, 1])), win_walk=int(r.choice([0, 1, 1])), wave_runs=int(r.choice([0, 1, 1])), chain_walk=int(r.choice([-1, -1, 0, 1, 1])), chain_bytes=int(r.choice([160 * 1024 - 768, 160 * 1024 - 768, 40000, 6000, 0])))
    t_case = time.time()
    res = resource_for(ds, calls=calls, **opts)
    for find_all in (True, False):
        pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
        for path in (host_match, device_match):
            p, rr = path(res, ds, find_all)
            if not (np.array_equal(p, pe) and np.array_equal(rr, re_)):
                bad += 1
                if os.environ.get("FUZZ_VERBOSE"):
                    k = 0
                    while k < min(len(rr), len(re_)) and (rr[k] == re_[k]).all():
                        k += 1
                    print("  first difference at record", k, "got", rr[max(0, k - 1):k + 4].tolist(), "expected", re_[max(0, k - 1):k + 4].tolist(),
                          "prefix equal", bool(np.array_equal(p, pe)), flush=True)
                print("MISMATCH", seed, opts, dict(alpha=alpha, lmin=lmin, lmax=lmax, n_pat=n_pat, n_hay=n_hay, hmax=hmax,
                                                   plants=plants, calls=calls), find_all, path.__name__, len(rr), len(re_), flush=True)
    api.ahocorasick_deinit(res)
    o.close()
    if os.environ.get("FUZZ_TIMES") and time.time() - t_case > float(os.environ["FUZZ_TIMES"]):   # cases slower than this many seconds
        print("SLOW", seed, round(time.time() - t_case, 1), opts, dict(lmin=lmin, lmax=lmax, n_pat=n_pat, n_hay=n_hay, hmax=hmax, plants=plants), flush=True)
print(f"fuzz: {n_cases} cases from seed {seed0}, {bad} mismatches, {time.time()-t0:.1f}s")
sys.exit(1 if bad else 0)
