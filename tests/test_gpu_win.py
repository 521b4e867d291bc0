"""k_walk_win (kernel 4 over the two-byte window table, csrc/k_win.hip): trie depth <= 2 decided in LDS from the last two text
bytes -- one child exactly, two by fingerprints, three or more by their list behind the cells, more than 62 "always maybe" --
a "maybe" settled by one probe of the depth-3 hash, deeper states walked through the compact records; every cell form, the
walk's hand-overs (deep -> shallow through a failure link, a maybe on the very byte that ended a descent, items cut in chunks,
ragged and unaligned haystacks), against the oracle and against k_walk_items on the same handle options.  The reference's hop
loop is src/multifast/ahocorasick.c:199-212.  Needs an MI355X: -m gpu."""
import numpy as np
import pytest

from gpu_util import device_match, host_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, synth

pytestmark = pytest.mark.gpu


def _oracle(ds, find_all=True):
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    out = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
    o.close()
    return out


def _families(seed=3):
    """Patterns whose first two bytes are shared by 1, 2, 3, 9, 40 and 200 third bytes (every cell form of the table), some of
    them prefixes / suffixes of others (failure links that lead back to depth 2 and into another descent), lengths 3..40."""
    r = np.random.default_rng(seed)
    pats = []
    for a, b, kids in ((1, 2, 1), (3, 4, 2), (5, 6, 3), (7, 8, 9), (9, 10, 40), (11, 12, 200), (0, 0, 5), (255, 255, 3)):
        for c in r.choice(256, size=kids, replace=False):
            tail = bytes(r.integers(0, 256, size=int(r.integers(0, 38))).tolist())
            pats.append(bytes([a, b, int(c)]) + tail)
    # nested: a pattern inside another, a pattern that is the suffix of another and begins with a listed digram
    pats += [bytes([7, 8, 1, 2, 5, 6, 9]), bytes([1, 2, 5, 6]), bytes([2, 5, 6]), bytes([5, 6, 7, 8]), bytes([1, 2, 5]), bytes([0, 0, 0]), bytes([0, 0, 0, 0, 0])]
    pats = list(dict.fromkeys(pats))
    hays = []
    for h in range(40):
        n = int(r.choice([0, 1, 2, 3, 5, 17, 64, 65, 300, 1500, 5000, 20000]))
        t = r.integers(0, 256, size=n).astype(np.uint8)
        for _ in range(n // 40):   # plants, cut plants, digrams of the families with a random third byte
            p = pats[int(r.integers(len(pats)))]
            cut = p[:int(r.integers(2, len(p) + 1))] if r.random() < 0.4 else p
            at = int(r.integers(0, max(1, n - len(cut))))
            if at + len(cut) <= n:
                t[at:at + len(cut)] = np.frombuffer(cut, dtype=np.uint8)
        hays.append(t.tobytes())
    hays.append(bytes([0]) * 3000)
    hays.append(bytes([1, 2, 5, 6, 7, 8] * 500))
    pb, po = synth.pack(pats)
    hb, ho = synth.pack(hays)
    return synth.Dataset("win-families", pb, po, hb, ho)


@pytest.mark.parametrize("opts", [dict(), dict(compact=1, dense_bytes=0), dict(compact=1, lds_bytes=4096), dict(chunk_bytes=256), dict(chunk_bytes=48, slab_events=1),
                                  dict(compact=1, dense_bytes=0, chunk_bytes=1024, refill=1)])
def test_window_walk_every_cell_form(opts):
    ds = _families()
    res = resource_for(ds, kernel=4, win_table=1, **opts)
    api.ahocorasick_finalize(res)
    assert res.stat("win_table") == 1
    for find_all in (True, False):
        pe, re_ = _oracle(ds, find_all)
        for path in (device_match, host_match):
            p, r = path(res, ds, find_all)
            assert np.array_equal(p, pe) and np.array_equal(r, re_), (opts, find_all, path.__name__, len(r), len(re_))
    assert len(_oracle(ds)[1]) > 300
    res.set_option("win_walk", 0)   # the same handle through k_walk_items
    p, r = device_match(res, ds)
    pe, re_ = _oracle(ds)
    assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("name", ["G1", "G2", "G6", "R3", "R7"])
def test_window_walk_on_the_goldens(name):
    """Golden datasets whose shortest pattern is >= 3 B through kernel 4 with the window table forced (small sets: dense rows
    below depth 2; G6: 2.27 M states, compact records)."""
    ds = synth.NAMED[name]()
    if min(len(p) for p in ds.patterns() if p) < 3:
        pytest.skip("a pattern shorter than 3 B: no window table")
    pe, re_ = _oracle(ds)
    res = resource_for(ds, kernel=4, win_table=1)
    api.ahocorasick_finalize(res)
    assert res.stat("win_table") == 1
    p, r = device_match(res, ds)
    assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)


def test_no_window_table_for_short_patterns():
    ds = synth.g4()   # patterns of 1..4 B: a state of depth <= 2 is final
    res = resource_for(ds, kernel=4, win_table=1)
    api.ahocorasick_finalize(res)
    assert res.stat("win_table") == 0
    pe, re_ = _oracle(ds)
    p, r = device_match(res, ds)
    assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)
