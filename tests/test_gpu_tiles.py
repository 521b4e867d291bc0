"""k_walk_tiles (kernel 6): the exact walk with exact-sized ordered output in one pass -- no slabs, no drain.
Every result record for record against the oracle (reference order: position ascending, longest pattern first,
src/multifast/node.c:424-441; one callback per event, src/php_ahocorasick.c:542-589).  Needs an MI355X: -m gpu."""
import numpy as np
import pytest

from gpu_util import device_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, batch, synth

pytestmark = pytest.mark.gpu


def _oracle(ds, find_all=True):
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    out = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
    o.close()
    return out


def _short_over_own_alphabet(n_hay=256, hay_len=8192, seed=41):
    """The headline text with eight patterns of 2..6 B over its own six letters: ~ 540 records per 8 KiB."""
    ds = synth.generate("T", 1, 8, 2, 6, n_hay, hay_len, b"abcdef")
    sp = synth.generate("short", seed, 8, 2, 6, 1, 16, b"abcdef")
    return synth.Dataset("T", sp.pat_blob, sp.pat_off, ds.hay_blob, ds.hay_off)


@pytest.mark.parametrize("geometry", [{}, dict(tile_w=16, tile_p=64), dict(tile_w=32, tile_p=5), dict(tile_w=64, tile_p=64), dict(tile_w=64, tile_p=1)])
def test_mid_density_in_one_pass(geometry):
    ds = _short_over_own_alphabet()
    pe, re_ = _oracle(ds)
    assert 100 * ds.n_haystacks < len(re_) < 2000 * ds.n_haystacks
    res = resource_for(ds, kernel=6, **geometry)
    for _ in range(3):   # the first call grows the record buffer and runs the walk twice
        p, r = device_match(res, ds)
        assert res.stat("kernel_used") == 6 and res.stat("overflow_items") == 0
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)


def test_automatic_mode_takes_the_tile_walk_for_keyword_like_text():
    """What used to go to the whole-item walk and its drains: a dense-ish set over its own alphabet."""
    ds = _short_over_own_alphabet(n_hay=512)
    pe, re_ = _oracle(ds)
    res = resource_for(ds)
    seen = []
    for _ in range(4):
        p, r = device_match(res, ds)
        seen.append(int(res.stat("kernel_used")))
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
    assert seen[-1] == 6, seen
    # findAll = false keeps the slab path (one event per haystack)
    pf, rf = _oracle(ds, False)
    p, r = device_match(res, ds, False)
    assert res.stat("kernel_used") != 6 and np.array_equal(p, pf) and np.array_equal(r, rf)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("name", ["G3", "G4"])
def test_dense_output_overflows_the_queue_and_is_written_by_the_second_walk(name):
    """A record (and more) per byte: more events than a wave's queue holds -- the lanes write their records in place."""
    ds = synth.NAMED[name]()
    pe, re_ = _oracle(ds)
    res = resource_for(ds, kernel=6, small_path=0)
    for _ in range(2):
        p, r = device_match(res, ds)
        assert res.stat("kernel_used") == 6
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("hmax,lmax", [(100, 4), (700, 17), (5000, 33), (20000, 129), (3000, 1)])
def test_ragged_haystacks_and_pattern_lengths(hmax, lmax):
    """Unaligned starts, haystacks shorter than a piece and longer than an item, warm-ups of 0 .. 8 blocks."""
    ds = synth._planted_ragged(f"TR{hmax}", 900 + hmax, 60, 1, lmax, 300, 0, hmax, b"abc", 4)
    pe, re_ = _oracle(ds)
    for geometry in ({}, dict(tile_w=16, tile_p=9), dict(tile_w=64, tile_p=64)):
        res = resource_for(ds, kernel=6, small_path=0, **geometry)
        p, r = device_match(res, ds)
        assert res.stat("kernel_used") == 6
        assert np.array_equal(p, pe) and np.array_equal(r, re_), geometry
        api.ahocorasick_deinit(res)


def test_patterns_too_long_for_lane_warmups_take_the_whole_item_walk():
    ds = synth._planted_ragged("TL", 77, 20, 100, 400, 50, 0, 3000, b"ab", 3)
    pe, re_ = _oracle(ds)
    res = resource_for(ds, kernel=6, small_path=0)
    p, r = device_match(res, ds)
    assert res.stat("kernel_used") == 4
    assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)


def test_two_batches_in_flight_and_a_buffer_that_is_too_small():
    """The first batch's records outgrow the optimistic capacity after the second begin has reused the look-back words and the
    queues: it runs again as a whole."""
    import torch
    ds = _short_over_own_alphabet(n_hay=300, hay_len=5000)
    pe, re_ = _oracle(ds)
    res = resource_for(ds, kernel=6)
    d_blob = torch.from_numpy(ds.hay_blob).cuda()
    d_off = torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
    batch.bind_current_stream(res)
    batch.match_device_begin(res, d_blob, d_off, True)
    batch.match_device_begin(res, d_blob, d_off, True)
    for _ in range(2):
        total = batch.match_device_end(res)
        p, r = batch.result_numpy(res, ds.n_haystacks, total)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)
