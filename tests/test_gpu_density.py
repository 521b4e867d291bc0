"""Many matches per haystack: items whose run lists or event slabs overflow, the sizes a handle learns from its
data, the drains that write such items' records, and findAll = false on haystacks of many work items -- all record for record
against the oracle.  Needs an MI355X: -m gpu."""
import numpy as np
import pytest

from gpu_util import device_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, synth

pytestmark = pytest.mark.gpu


def _planted(n_hay, hay_len, per_hay, seed=3, n_pat=2048, plen=16, alpha=b"abcdef"):
    """Headline-like set over `n_hay` haystacks with `per_hay` occurrences planted in each."""
    ds = synth.generate("D", seed, n_pat, plen, plen, n_hay, hay_len, alpha)
    r = np.random.default_rng(seed)
    for h in range(n_hay):
        for k in range(per_hay):
            p = np.frombuffer(ds.pattern(int(r.integers(n_pat))), dtype=np.uint8)
            at = int(ds.hay_off[h]) + (hay_len // per_hay) * k + int(r.integers(0, max(1, hay_len // per_hay - plen)))
            ds.hay_blob[at:at + plen] = p
    return ds


def _oracle(ds, find_all=True):
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    out = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
    o.close()
    return out


@pytest.mark.parametrize("relist", [1, 0])
@pytest.mark.parametrize("per_hay", [6, 17, 40, 100])
def test_many_candidates_per_item_through_the_filter(per_hay, relist):
    """More candidate windows per 8-KiB item than the first window lists (16) and slabs hold.  relist = 1 (default): the batch is run
    again with the lists the scan's own count asks for ("relist_reruns"); relist = 0: the items are walked by their waves and
    drained, and the handle's lists and slabs grow for the calls after -- same records every time either way."""
    ds = _planted(512, 8192, per_hay)
    pe, re_ = _oracle(ds)
    res = resource_for(ds, relist=relist)
    overflow = []
    for _ in range(4):
        p, r = device_match(res, ds)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
        overflow.append(int(res.stat("overflow_items")))
    if per_hay <= 40:   # what the lists and slabs (<= 256) can learn to hold
        assert overflow[-1] == 0, overflow
    # (which kernel the later calls run on is a measured choice: tests/test_gpu_cliffs.py)
    if 17 <= per_hay <= 40:
        if relist:
            assert res.stat("relist_reruns") >= 1
        else:
            assert overflow[0] > 0 and res.stat("relist_reruns") == 0, overflow
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("kernel", [1, 4])
def test_event_slabs_learn_their_size(kernel):
    """Dense-ish text through the table-walk kernels: the first call sends most items to the drain, later calls none (the slabs
    grew), and a caller who fixes `slab_events` keeps the drain -- same records all along."""
    ds = _planted(256, 8192, 200, seed=5, n_pat=64, plen=12, alpha=b"ab")   # ~ 80 events per 2-KiB work item
    pe, re_ = _oracle(ds)
    res = resource_for(ds, kernel=kernel, chunk_bytes=2048)   # (slabs may take half the batch's bytes: 128 slots for such items)
    seen = []
    for _ in range(5):
        p, r = device_match(res, ds)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
        seen.append(int(res.stat("overflow_items")))
    assert seen[0] > 0 and seen[-1] < seen[0], seen
    api.ahocorasick_deinit(res)
    res = resource_for(ds, kernel=kernel, chunk_bytes=2048, slab_events=2)
    for _ in range(3):
        p, r = device_match(res, ds)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
        assert res.stat("overflow_items") > 0
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("kernel", [1, 2, 4])
def test_find_first_on_haystacks_of_many_items(kernel):
    """findAll = false: the first event of every haystack, also when the haystack is cut into hundreds of work items and its
    first match lies deep inside it (the filter that keeps a haystack's first item is item-parallel)."""
    lens = [200_000, 0, 5, 90_000, 17, 300_000, 64, 1]
    r = np.random.default_rng(9)
    pats = [bytes(r.choice(np.frombuffer(b"abcdef", dtype=np.uint8), size=16).tolist()) for _ in range(300)]
    hays = []
    for n in lens:
        t = r.choice(np.frombuffer(b"abcdef", dtype=np.uint8), size=n).astype(np.uint8)
        for at in range(int(n * 0.6), n - 16, 9000):   # first occurrence 60 % into the haystack, more behind it
            t[at:at + 16] = np.frombuffer(pats[int(r.integers(len(pats)))], dtype=np.uint8)
        hays.append(t.tobytes())
    pb, po = synth.pack(pats)
    hb, ho = synth.pack(hays)
    ds = synth.Dataset("first", pb, po, hb, ho)
    for find_all in (False, True):
        pe, re_ = _oracle(ds, find_all)
        res = resource_for(ds, kernel=kernel, chunk_bytes=1024)
        for _ in range(2):
            p, rr = device_match(res, ds, find_all)
            assert np.array_equal(p, pe) and np.array_equal(rr, re_)
        api.ahocorasick_deinit(res)


@pytest.mark.parametrize("kernel", [0, 1, 4, 6])
def test_events_of_more_than_255_records(kernel):
    """A nested family a^1 .. a^300: the deepest final state stands for 300 records, more than the one-word head of a state's
    output (`out_head`: the first pattern and min(count, 255)) can say -- the kernels fall back to the full count and list.
    Two haystacks: long enough for every state, and a short one."""
    pats = [b"a" * k for k in range(1, 301)] + [b"ba", b"ab"]
    hays = [b"b" + b"a" * 700 + b"b" + b"a" * 40, b"aaab", b""]
    pb, po = synth.pack(pats)
    hb, ho = synth.pack(hays)
    ds = synth.Dataset("nested300", pb, po, hb, ho)
    res = resource_for(ds, kernel=kernel)
    for find_all in (True, False):
        pe, re_ = _oracle(ds, find_all)
        p, r = device_match(res, ds, find_all)
        assert np.array_equal(p, pe) and np.array_equal(r, re_), (kernel, find_all, len(r), len(re_))
    assert int(pe[-1]) >= 1 and len(_oracle(ds)[1]) > 150000
    api.ahocorasick_deinit(res)


_CFG3 = {}


@pytest.mark.parametrize("per_8k", [32, 128, 512])
@pytest.mark.parametrize("hay_len", [8192, 65536])
def test_cfg3_dense_candidates_against_the_oracle(per_8k, hay_len):
    """The 65,536-pattern set (8..64 B, Sigma = 256) over text with 32 / 128 / 512 planted occurrences per 8 KiB -- tens to hundreds
    of candidate windows per work item of the big-set filter: window lists that grow, k_runs_wave (a wave per item, a lane per
    window, the occurrences ordered in LDS), k_expand_dense, items that shrink at 512 -- through the filter (kernel 2), the
    automatic choice, and with the wave kernels switched off (a lane per item); every call record for record against the oracle
    (the reference orders a position's records longest first: tests/test1.phpt:60-157; ahocorasick.c:214-231 reports in text order)."""
    base = synth.density_set("cfg3")
    ds = synth.plant_density(base, per_8k, hay_len)
    n = min(ds.n_haystacks, (512 << 10) // hay_len)    # 512 KiB of the tile: 64 items of 8 KiB
    sub = synth.Dataset(ds.name, ds.pat_blob, ds.pat_off, ds.hay_blob[:int(ds.hay_off[n])], ds.hay_off[:n + 1])
    if "o" not in _CFG3:   # (the CPU trie of 65,536 patterns takes seconds to build: once for the six cells)
        _CFG3["o"] = Oracle("port")
        _CFG3["o"].add_patterns((base.pat_blob, base.pat_off))
    pe, re_ = _CFG3["o"].match_batch(sub.hay_blob, sub.hay_off, True)
    assert len(re_) >= per_8k * 60
    for opts in (dict(kernel=2), dict(kernel=3), dict(kernel=2, wave_runs=0)):
        res = resource_for(sub, **opts)
        seen = []
        for _ in range(5):   # (the hand-over learns: list and slab sizes, item size, which kernels are enqueued)
            p, r = device_match(res, sub)
            assert np.array_equal(p, pe) and np.array_equal(r, re_), (opts, len(r), len(re_))
            seen.append((int(res.stat("kernel_used")), int(res.stat("wave_runs_used")), int(res.stat("max_windows"))))
        if opts.get("kernel") == 2 and opts.get("wave_runs", 1):
            assert seen[-1][1] == 1, seen   # k_runs_wave took the busy items
        for find_all in (False,):
            pe1, re1 = _CFG3["o"].match_batch(sub.hay_blob, sub.hay_off, find_all)
            p, r = device_match(res, sub, find_all)
            assert np.array_equal(p, pe1) and np.array_equal(r, re1)
        api.ahocorasick_deinit(res)


def test_bounded_result_memory_with_dense_candidates():
    """"max_result_bytes" (config 5's overflow -> drain -> resume) on text with many matches behind the filter: the segments the
    device-resident entry hands out -- through window lists that grow, k_runs_wave and k_expand_dense, which must not write past
    the cap -- concatenate to the oracle's array."""
    import torch
    from php_aho_corasick_amd import batch
    base = synth.density_set("headline")
    ds = synth.plant_density(base, 64, 8192)
    n = 96
    ds = synth.Dataset(ds.name, ds.pat_blob, ds.pat_off, ds.hay_blob[:int(ds.hay_off[n])], ds.hay_off[:n + 1])
    pe, re_ = _oracle(ds)
    per_hay = int(pe[-1]) // n + 8
    res = resource_for(ds, kernel=2, small_path=0, max_result_bytes=max(4096, 10 * per_hay * 16))
    d_blob = torch.from_numpy(ds.hay_blob).cuda()
    d_off = torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
    batch.bind_current_stream(res)
    for _ in range(3):   # (the hand-over learns over the calls)
        got, segs = [], 0
        for first, k, total in batch.match_device_bounded(res, d_blob, d_off):
            p, r = batch.result_numpy(res, k, total)
            assert int(p[-1]) == total and k >= 1
            r = r.copy()
            r[:, 0] += first
            got.append(r)
            segs += 1
        assert segs >= 8 and np.array_equal(np.concatenate(got), re_)
    assert res.stat("kernel_used") == 2
    api.ahocorasick_deinit(res)
