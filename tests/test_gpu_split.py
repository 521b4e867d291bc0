"""Split pattern sets: a few patterns under 7 B among thousands of long ones (the reference takes any length from 1 to 1,024 B
in one trie, src/multifast/ahocorasick.c:101-105).  As a whole such a set is not eligible for the anchor filter; the library
matches it as two automata over the same batch -- the long patterns through the filter, the rest through the table walk -- and
merges the two match arrays per haystack (position ascending, longer pattern first: node.c:424-441).  Goldens made by the
compiled reference (tests/golden/golden_targeted.json: X1, X2)."""
import numpy as np
import pytest

from helpers import load_json, sha
from gpu_util import device_match, host_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def targeted():
    return {(c["dataset"], c["find_all"]): c for c in load_json("golden_targeted.json")["cases"]}


def _check(res, ds, golden, name, path):
    for find_all in (True, False):
        g = golden[(name, find_all)]
        prefix, rec = path(res, ds, find_all)
        assert len(rec) == g["n_records"], (name, find_all)
        assert sha(rec) == g["sha256"], (name, find_all)
        assert sha(prefix) == g["prefix_sha256"], (name, find_all)


@pytest.mark.parametrize("name", ["X1", "X2"])
@pytest.mark.parametrize("path", [host_match, device_match], ids=["host", "device"])
def test_split_set_goldens(name, path, targeted):
    ds = synth.TARGETED[name]()
    res = resource_for(ds)
    _check(res, ds, targeted, name, path)
    assert res.stat("split") == 1 and res.stat("kernel_used") == 2          # the long part ran through the anchor filter
    assert res.stat("split_short_patterns") in (7, 3) and res.stat("split_long_patterns") in (2048, 5000)
    assert res.stat("anchor_q") > 0                                            # (the filter's facts are the long part's)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("name", ["X1", "X2"])
@pytest.mark.parametrize("options", [dict(split=0), dict(split=1, chunk_bytes=256), dict(split=1, slab_events=1, run_cap=1), dict(split=-1, verify=0),
                                     dict(split=1, tiny_rows=0, short_rows=0), dict(kernel=1), dict(kernel=4),
                                     # the short part through the tile walk, which merges the long part's records as it writes its own
                                     dict(split=1, part_kernel=6), dict(split=1, part_kernel=6, tile_w=16, tile_p=5), dict(split=1, part_kernel=4)])
def test_split_set_options(name, options, targeted):
    """Not split (one automaton, the table walk), split with chunked items / overflowing slabs and run lists / walked candidates,
    and a forced kernel (which keeps the set whole): the same records."""
    ds = synth.TARGETED[name]()
    res = resource_for(ds, **options)
    _check(res, ds, targeted, name, device_match)
    assert res.stat("split") == (1 if options.get("split", -1) != 0 and "kernel" not in options else 0)
    api.ahocorasick_deinit(res)


def test_split_set_two_batches_in_flight_and_replace():
    import torch
    from php_aho_corasick_amd import batch
    ds = synth.x1()
    o = Oracle("port")
    for k, p in enumerate(ds.patterns()):
        o.set_replacement(k, synth.replacement(k, p))
    o.add_patterns((ds.pat_blob, ds.pat_off))
    res = resource_for(ds)
    for k, p in enumerate(ds.patterns()):
        api.set_replacement(res, k, synth.replacement(k, p))
    batch.bind_current_stream(res)
    halves = []
    for lo, hi in ((0, 40), (40, 90), (10, 70)):
        off = (ds.hay_off[lo:hi + 1] - ds.hay_off[lo])
        blob = ds.hay_blob[int(ds.hay_off[lo]):int(ds.hay_off[hi])]
        halves.append((torch.from_numpy(blob.copy()).cuda(), torch.from_numpy(off.astype(np.int64)).cuda(), hi - lo, o.match_batch(blob, off, True)))
    order = [0, 1, 2, 0]
    began = ended = 0
    while ended < len(order):
        while began < len(order) and began - ended < 2:
            batch.match_device_begin(res, halves[order[began]][0], halves[order[began]][1], True)
            began += 1
        total = batch.match_device_end(res)
        d = halves[order[ended]]
        p, r = batch.result_numpy(res, d[2], total)
        assert np.array_equal(p, d[3][0]) and np.array_equal(r, d[3][1]), ended
        ended += 1
    assert res.stat("split") == 1
    # search and replace over the merged match array
    for mode in (1, 2):
        out, off = api.replace_arrays(res, ds.hay_blob, ds.hay_off, mode)
        for h in range(0, ds.n_haystacks, 7):
            assert out[int(off[h]):int(off[h + 1])].tobytes() == o.replace(ds.haystack(h), mode), (h, mode)
    # single matches through the PHP surface (the one-launch small path is not taken by a split set)
    want = o.match_batch(ds.hay_blob[:int(ds.hay_off[3])], ds.hay_off[:4], True)[1]
    got = api.ahocorasick_match(ds.haystack(2), res)
    assert [(g["pos"], g["start_postion"]) for g in got] == [(int(r[1]), int(r[2])) for r in want[want[:, 0] == 2]]
    api.ahocorasick_deinit(res)


def test_merged_output_of_a_split_set_at_mid_density():
    """The headline set + short patterns over the text's own alphabet (~ 540 records per 8 KiB): the short part's tile walk writes
    the merged array -- the long part's records placed among its own -- in one pass; first call (the buffers grow: the part runs
    again), steady state, two batches in flight, and the automatic choice finding this path by itself."""
    import torch
    from php_aho_corasick_amd import batch
    base = synth.bench_batch(256, 8192, seed=1, planted=True)
    sp = synth.generate("short", 41, 8, 2, 6, 1, 16, b"abcdef")
    pb, po = synth.pack(base.patterns() + sp.patterns())
    ds = synth.Dataset("XM", pb, po, base.hay_blob, base.hay_off)
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, True)
    assert len(re_) > 100 * ds.n_haystacks
    for options in (dict(part_kernel=6), {}):
        res = resource_for(ds, **options)
        for _ in range(8 if not options else 3):
            p, r = device_match(res, ds)
            assert np.array_equal(p, pe) and np.array_equal(r, re_)
        assert res.stat("split") == 1 and res.stat("merged_batches") >= 1, options
        d_blob, d_off = torch.from_numpy(ds.hay_blob).cuda(), torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
        batch.bind_current_stream(res)
        batch.match_device_begin(res, d_blob, d_off, True)
        batch.match_device_begin(res, d_blob, d_off, True)
        for _ in range(2):
            total = batch.match_device_end(res)
            p, r = batch.result_numpy(res, ds.n_haystacks, total)
            assert np.array_equal(p, pe) and np.array_equal(r, re_)
        pf, rf = o.match_batch(ds.hay_blob, ds.hay_off, False)
        p, r = device_match(res, ds, False)
        assert np.array_equal(p, pf) and np.array_equal(r, rf)
        api.ahocorasick_deinit(res)
    o.close()


@pytest.mark.parametrize("name", ["X1", "X2"])
@pytest.mark.parametrize("options", [dict(split=1), dict(split=1, part_kernel=6), dict(split=1, part_kernel=4), dict(split=1, part_kernel=1)])
@pytest.mark.parametrize("find_all", [True, False])
def test_split_set_under_a_result_bound(name, options, find_all):
    """"max_result_bytes" on a split set (acmi.h: "never allocating more", ACMI_E_PARTIAL): each part is bounded like a whole
    handle, the merged result takes the haystacks both parts hold whose records together fit -- segment by segment the
    concatenation is the oracle's array, and no segment exceeds the bound."""
    import torch
    from php_aho_corasick_amd import batch, _native as N
    ds = synth.TARGETED[name]()
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
    per_hay = np.diff(pe)
    cap = max(4096, (int(per_hay.max()) * 16 + 15) & ~15, int(pe[-1]) * 16 // 5 & ~15)   # about five segments; the largest haystack fits
    res = resource_for(ds, max_result_bytes=cap, small_path=0, **options)
    d_blob = torch.from_numpy(ds.hay_blob).cuda()
    d_off = torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
    batch.bind_current_stream(res)
    got, segments = [], 0
    for first, k, total in batch.match_device_bounded(res, d_blob, d_off, find_all):
        p, r = batch.result_numpy(res, k, total)
        assert int(p[-1]) == total and total * 16 <= cap and k >= 1
        assert np.array_equal(p, pe[first:first + k + 1] - pe[first])
        r = r.copy()
        r[:, 0] += first
        got.append(r)
        segments += 1
    assert res.stat("split") == 1 and segments >= (3 if int(pe[-1]) * 16 >= 3 * cap else 1)
    assert np.array_equal(np.concatenate(got), re_)
    p, r = host_match(res, ds, find_all)                   # the host-buffer entry drains the same way inside the library
    assert np.array_equal(p, pe) and np.array_equal(r, re_)
    res.set_option("max_result_bytes", 4096)
    if int(per_hay.max()) * 16 > 4096:
        with pytest.raises(N.AcmiError) as e:
            batch.match_device(res, d_blob, d_off, find_all)
        assert e.value.status == N.E_TOO_LARGE
    res.set_option("max_result_bytes", 0)                  # ... and without the bound the merged path is back
    p, r = device_match(res, ds, find_all)
    assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)
    o.close()
