"""BASELINE.json-size checks through size-independent properties.  Needs an MI355X: -m gpu."""
import numpy as np
import pytest

from gpu_util import host_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, batch, synth

pytestmark = pytest.mark.gpu


def _device_batch(ds):
    import torch
    return torch.from_numpy(ds.hay_blob).cuda(), torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()


@pytest.mark.parametrize("ilp", [1, 2])
def test_config2_planted_at_scale(ilp):
    """2048x16 B patterns over 2^15 x 8 KiB haystacks (256 MiB): every haystack carries exactly one
    planted pattern at a known place; unplanted hits in random abcdef text are ~6e-6 per haystack.
    A 2,048-haystack slice is also checked record-for-record against the oracle."""
    nh = 1 << 15
    ds = synth.bench_batch(nh)
    res = resource_for(ds, ilp=ilp)
    d_blob, d_off = _device_batch(ds)
    batch.bind_current_stream(res)
    total = batch.match_device(res, d_blob, d_off)
    prefix, rec = batch.result_numpy(res, nh, total)
    h = np.arange(nh)
    at = (31 * h + 5) % (8192 - 16)
    planted = np.stack([h, at + 16, at, (7 * h + 3) % 2048], axis=1).astype(np.uint32)
    # every planted record is present, in order
    idx = prefix[:-1].astype(np.int64)
    assert total >= nh and np.all(np.diff(prefix.astype(np.int64)) >= 1)
    keys = rec[:, 0].astype(np.int64) * (1 << 32) + rec[:, 1]
    assert np.all(np.diff(keys) >= 0), "records must be ordered by (haystack, pos)"
    got = {tuple(int(x) for x in r) for r in rec}
    assert all(tuple(int(x) for x in p) in got for p in planted[:: 97])
    assert total - nh <= 4  # accidental extra hits are astronomically rare
    # oracle slice
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    lo, hi = 4096, 4096 + 2048
    sub_off = ds.hay_off[lo:hi + 1] - ds.hay_off[lo]
    pe, re_ = o.match_batch(ds.hay_blob[int(ds.hay_off[lo]):int(ds.hay_off[hi])], sub_off)
    re_[:, 0] += lo
    assert np.array_equal(rec[int(prefix[lo]):int(prefix[hi])], re_)
    _ = idx
    api.ahocorasick_deinit(res)


def test_config5_dense_count_formula():
    """Adversarial a^1..a^16 over a^8192: records = NH*(K*HL - K(K-1)/2); order and content exact."""
    nh, hl, k = 64, 8192, 16
    ds = synth.g3(n_hay=nh, hay_len=hl, k=k)
    res = resource_for(ds)
    d_blob, d_off = _device_batch(ds)
    batch.bind_current_stream(res)
    total = batch.match_device(res, d_blob, d_off)
    assert total == nh * (k * hl - k * (k - 1) // 2)
    prefix, rec = batch.result_numpy(res, nh, total)
    per = k * hl - k * (k - 1) // 2
    assert np.array_equal(prefix, np.arange(nh + 1, dtype=np.uint64) * np.uint64(per))
    # one haystack's stream, closed form: at pos p the patterns min(p,16)..1, longest first
    one = rec[:per]
    pos = np.concatenate([np.full(min(p, k), p) for p in range(1, hl + 1)])
    ln = np.concatenate([np.arange(min(p, k), 0, -1) for p in range(1, hl + 1)])
    assert np.array_equal(one[:, 1], pos) and np.array_equal(one[:, 3], ln - 1) and np.array_equal(one[:, 2], pos - ln)
    assert np.array_equal(rec[per:2 * per, 1:], one[:, 1:]) and np.all(rec[per:2 * per, 0] == 1)
    api.ahocorasick_deinit(res)


def test_idempotent_and_chunk_invariant():
    """Same batch, different work-item sizes / kernels -> byte-identical result arrays."""
    ds = synth.bench_batch(1 << 12)
    ref = None
    for opts in (dict(kernel=1, ilp=1), dict(kernel=1, ilp=2, chunk_bytes=1024), dict(kernel=0, chunk_bytes=2048),
                 dict(kernel=4), dict(kernel=4, walk_group=8, chunk_bytes=3000), dict(kernel=2)):
        res = resource_for(ds, **opts)
        d_blob, d_off = _device_batch(ds)
        batch.bind_current_stream(res)
        for _ in range(2):
            total = batch.match_device(res, d_blob, d_off)
            out = batch.result_numpy(res, ds.n_haystacks, total)
            if ref is None:
                ref = out
            assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1])
        api.ahocorasick_deinit(res)


def test_nine_million_tiny_haystacks():
    """More haystacks than the two-launch prefix sum takes (4,096 tiles of 2,048): the recursive
    three-kernel scan must give the same item table.  9.4 M haystacks of 0..3 bytes over {a,b}, patterns
    a / ab / ba / bb: per-haystack counts against a vectorised restatement, a slice against the oracle."""
    nh = 9_400_000
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 4, nh, dtype=np.int64)
    off = np.zeros(nh + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens, dtype=np.uint64)
    blob = rng.integers(0, 2, int(off[-1]), dtype=np.uint8) + ord("a")
    pat_blob, pat_off = synth.pack([b"a", b"ab", b"ba", b"bb"])
    ds = synth.Dataset("tiny", pat_blob, pat_off, blob, off)
    res = resource_for(ds, slab_events=4)
    d_blob, d_off = _device_batch(ds)
    batch.bind_current_stream(res)
    total = batch.match_device(res, d_blob, d_off)
    prefix, rec = batch.result_numpy(res, nh, total)
    # expected records per haystack: one per 'a', one per adjacent pair other than "aa"
    is_a = (blob == ord("a")).astype(np.int64)
    pair = np.zeros(len(blob), dtype=np.int64)
    pair[1:] = 1 - is_a[1:] * is_a[:-1]
    first = np.zeros(len(blob), dtype=bool)
    first[off[:-1][lens > 0].astype(np.int64)] = True     # a pair must not straddle haystacks
    pair[first] = 0
    per_byte = is_a + pair
    csum = np.concatenate([[0], np.cumsum(per_byte)])
    expect = csum[off[1:].astype(np.int64)] - csum[off[:-1].astype(np.int64)]
    assert total == int(expect.sum())
    assert np.array_equal(np.diff(prefix.astype(np.int64)), expect)
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    lo, hi = 9_000_000, 9_002_000
    sub_off = off[lo:hi + 1] - off[lo]
    pe, re_ = o.match_batch(blob[int(off[lo]):int(off[hi])], sub_off)
    re_[:, 0] += lo
    assert np.array_equal(rec[int(prefix[lo]):int(prefix[hi])], re_)
    api.ahocorasick_deinit(res)
    o.close()


def test_config3_full_size_planted_hits():
    """BASELINE configs[2] at its full size: 65,536 patterns of 8..64 B over 2^20 haystacks x 4 KiB (4 GiB resident).
    Size-independent properties: every haystack reports its planted pattern at its place, records are ordered by
    (haystack, position), the count repeats tile by tile, verify=1 (text compare) and verify=0 (automaton walk) agree;
    and the first 64 haystacks ARE the G6 golden (same generator, same plants), compared record for record."""
    import torch
    from helpers import load_json, sha
    tile_h, hl, nh = 4096, 4096, 1 << 20
    ds = synth.g6(n_hay=tile_h, hay_len=hl)
    reps = nh // tile_h
    d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(reps).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * hl
    results = []
    for verify in (1, 0):
        res = resource_for(ds, verify=verify)
        batch.bind_current_stream(res)
        total = batch.match_device(res, d_blob, d_off)
        assert res.stat("kernel_used") == 2 and res.stat("compact") == 1
        prefix, rec = batch.result_numpy(res, nh, total)
        results.append((total, sha(prefix), sha(rec)))
        if verify == 1:
            h = np.arange(tile_h, dtype=np.int64)
            pat = (1021 * h + 17) % ds.n_patterns
            start = (37 * h + 11) % (hl - 64)
            plen = np.diff(ds.pat_off.astype(np.int64))[pat]
            keys = rec[:, 0].astype(np.int64) * (1 << 32) + rec[:, 1]
            assert np.all(np.diff(keys) >= 0), "records must be ordered by (haystack, pos)"
            per_tile = int(prefix[tile_h])
            assert total == per_tile * reps and int(prefix[-1]) == total
            for t in (0, reps // 2, reps - 1):  # the same tile wherever it lies in the batch
                lo, hi = int(prefix[t * tile_h]), int(prefix[(t + 1) * tile_h])
                got = {(int(a) - t * tile_h, int(b), int(c), int(d)) for a, b, c, d in rec[lo:hi]}
                assert all((int(hh), int(s + l), int(s), int(p)) in got for hh, s, l, p in zip(h, start, plen, pat))
            g = {(c["dataset"], c["find_all"]): c for c in load_json("golden.json")["cases"]}[("G6", True)]
            n64 = int(prefix[64])
            assert n64 == g["n_records"] and sha(rec[:n64]) == g["sha256"]
        api.ahocorasick_deinit(res)
    assert results[0] == results[1]


@pytest.mark.parametrize("kernel", [3, 4, 6, 1])
def test_bounded_result_memory_drains_and_resumes(kernel):
    """BASELINE configs[4]'s "overflow -> drain -> resume" with a HARD bound on the device's result memory ("max_result_bytes"): G3
    (16 records per byte: 8.4 MB of records per 32 KiB of text) under a 1 MiB cap -- the device-resident entry returns
    ACMI_E_PARTIAL with the haystacks that fit and the index to resume from, the host entry drains segment by segment; either way
    the concatenation is the oracle's array bit for bit, and nothing above the cap was ever allocated for records."""
    import torch
    from php_aho_corasick_amd import batch, _native as N
    ds = synth.g3(n_hay=24, hay_len=4096)
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, True)
    per_hay = int(pe[1])
    cap = 3 * per_hay * 16 + 4096          # three haystacks' records and a bit
    res = resource_for(ds, kernel=kernel, max_result_bytes=cap, small_path=0)
    d_blob = torch.from_numpy(ds.hay_blob).cuda()
    d_off = torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
    batch.bind_current_stream(res)
    got, segments = [], []
    for first, k, total in batch.match_device_bounded(res, d_blob, d_off):
        p, r = batch.result_numpy(res, k, total)
        assert int(p[-1]) == total and 1 <= k <= 3
        r = r.copy()
        r[:, 0] += first
        got.append(r)
        segments.append(k)
    assert sum(segments) == ds.n_haystacks and len(segments) == 8
    assert np.array_equal(np.concatenate(got), re_)
    # the host-buffer entry (what the Zend shim calls) drains the same way inside the library
    p, r = host_match(res, ds)
    assert np.array_equal(p, pe) and np.array_equal(r, re_)
    # one haystack whose records alone exceed the cap: an error, not a hang
    res.set_option("max_result_bytes", 4096)
    with pytest.raises(N.AcmiError) as e:
        batch.match_device(res, d_blob, d_off, True)
    assert e.value.status == N.E_TOO_LARGE
    api.ahocorasick_deinit(res)
    o.close()
