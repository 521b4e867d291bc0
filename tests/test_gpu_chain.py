"""k_walk_chain (kernel 6 over the chain-compressed LDS image of the WHOLE automaton, csrc/k_chain.hip): dense rows for the branching
states, goto edge + failure pointer -- the reference's node record, src/multifast/node.h:37-67 -- for the insides of the patterns, the
hop loop of ahocorasick.c:199-212 ended after one hop by construction.  Every result record for record against the oracle
(position ascending, longest pattern first: node.c:424-441).  Needs an MI355X: -m gpu."""
import numpy as np
import pytest

from gpu_util import device_match, host_match, resource_for
from helpers import load_json, sha
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, batch, synth

pytestmark = pytest.mark.gpu


def _oracle(ds, find_all=True):
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    out = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
    o.close()
    return out


def _check(res, ds, pe, re_, calls=2):
    for _ in range(calls):   # the first call grows the record buffer and runs the walk twice
        p, r = device_match(res, ds)
        assert res.stat("kernel_used") == 6 and res.stat("chain_walk_used") == 1
        assert np.array_equal(p, pe) and np.array_equal(r, re_)


@pytest.mark.parametrize("name", ["G1", "G2", "R1", "R2", "K1", "T256", "TRAG", "S3K", "B512", "BRAG"])
def test_goldens_through_the_chain_walk(name):
    """The golden datasets whose automaton does not fit a 2 KiB tile as dense rows: hashes made by the compiled reference."""
    golden = {(c["dataset"], c["find_all"]): c for c in load_json("golden.json")["cases"] + load_json("golden_targeted.json")["cases"]}
    ds = (synth.NAMED.get(name) or synth.TARGETED[name])()
    res = resource_for(ds, kernel=6, lds_bytes=2048, small_path=0, split=0, chain_walk=1)
    p, r = device_match(res, ds)
    if res.stat("chain_image_bytes") == 0:
        api.ahocorasick_deinit(res)
        pytest.skip("no chain image for this set")
    assert res.stat("kernel_used") == 6 and res.stat("chain_walk_used") == 1
    g = golden[(name, True)]
    assert len(r) == g["n_records"] and sha(r) == g["sha256"] and sha(p) == g["prefix_sha256"]
    # ... and the same records as k_walk_tiles gives for the same set
    res.set_option("chain_walk", 0)
    p2, r2 = device_match(res, ds)
    assert res.stat("chain_walk_used") == 0 and np.array_equal(p, p2) and np.array_equal(r, r2)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("kind", ["headline", "words"])
@pytest.mark.parametrize("per,hay_len", [(1, 8192), (64, 8192), (512, 8192), (128, 256), (32, 1 << 20), (256, 5000)])
def test_density_sets_at_their_real_size(kind, per, hay_len):
    """The two sets the bench's density legs use (2,048 x 16 B over six letters: 25,580 states; 2,048 words: 12,846), whole in LDS."""
    base = synth.density_set(kind)
    ds = synth.plant_density(base, per, hay_len)
    n = max(1, min(ds.n_haystacks, (2 << 20) // hay_len))
    ds = synth.Dataset(ds.name, ds.pat_blob, ds.pat_off, ds.hay_blob[:int(ds.hay_off[n])], ds.hay_off[:n + 1])
    pe, re_ = _oracle(ds)
    res = resource_for(ds, kernel=6, small_path=0)
    _check(res, ds, pe, re_)
    assert res.stat("chain_image_bytes") > 100_000
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("hmax,lmax", [(100, 4), (700, 17), (5000, 33), (20000, 129), (3000, 1)])
def test_ragged_haystacks_and_pattern_lengths(hmax, lmax):
    """Unaligned starts, haystacks shorter than a piece and longer than an item, warm-ups of 0 .. 8 blocks."""
    ds = synth._planted_ragged(f"CR{hmax}", 900 + hmax, 400, 1, lmax, 300, 0, hmax, b"abc", 4)
    pe, re_ = _oracle(ds)
    for geometry in ({}, dict(tile_w=16, tile_p=9), dict(tile_w=64, tile_p=64), dict(tile_w=32, tile_p=1), dict(tile_w=128, tile_p=5),
                     dict(tile_w=128, tile_p=64)):   # (128: round 6, k_walk_chain's pieces over big batches)
        res = resource_for(ds, kernel=6, small_path=0, lds_bytes=2048, chain_walk=1, **geometry)
        p, r = device_match(res, ds)
        if lmax > 4:
            assert res.stat("kernel_used") == 6 and res.stat("chain_walk_used") == 1
        assert np.array_equal(p, pe) and np.array_equal(r, re_), geometry
        api.ahocorasick_deinit(res)


@pytest.mark.parametrize("name", ["G3", "G4", "G5", "R1"])
def test_small_automata_all_rows_in_lds(name):
    """A small automaton's image is all of its rows (one read per step): kernel 6 keeps k_walk_tiles for those unless "chain_walk" 1
    asks for k_walk_chain -- the same records, dense output (a^1..a^16: sixteen records per byte) included."""
    ds = synth.NAMED[name]()
    pe, re_ = _oracle(ds)
    res = resource_for(ds, kernel=6, small_path=0, chain_walk=1)
    _check(res, ds, pe, re_)
    assert res.stat("chain_dense_states") == res.stat("states") - 1
    res.set_option("chain_walk", -1)
    p, r = device_match(res, ds)
    assert res.stat("kernel_used") == 6 and res.stat("chain_walk_used") == 0 and np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)


def _nested(n_hay=64, hay_len=4096):
    """a^1 .. a^48 (an event of up to 48 records: more than the five bits of a record's count) among 80 random patterns over {a, b},
    over text that is mostly 'a': a record and more per byte, more events per lane than its LDS list holds."""
    rng = np.random.default_rng(5)
    pats = [b"a" * k for k in range(1, 49)] + [bytes(rng.choice(list(b"ab"), size=int(rng.integers(12, 30))).tolist()) for _ in range(80)]
    text = rng.choice(np.frombuffer(b"aaaaaaab", np.uint8), size=n_hay * hay_len).astype(np.uint8)
    pat_off = np.concatenate([[0], np.cumsum([len(p) for p in pats])]).astype(np.uint64)
    return synth.Dataset("NEST", np.frombuffer(b"".join(pats), np.uint8).copy(), pat_off, text, (np.arange(n_hay + 1) * hay_len).astype(np.uint64))


def test_saturated_counts_and_lane_lists_that_overflow():
    ds = _nested()
    pe, re_ = _oracle(ds)
    assert len(re_) > 4 * ds.hay_blob.size
    res = resource_for(ds, kernel=6, small_path=0, lds_bytes=2048, chain_walk=1)
    _check(res, ds, pe, re_, calls=3)
    api.ahocorasick_deinit(res)


def test_event_logs_that_overflow_are_grown_and_the_batch_runs_again():
    """The waves' event logs are sized for an event per 32 bytes of text (learned like the record buffer): 96 MiB with an event at
    EVERY byte -- more tiles than the grid has waves, so that a wave's log is not simply its one tile's worst case -- overflows
    them, the batch is run again with logs for eight times the events (twice: 32 -> 4 -> 1 bytes per event), and the result is
    the oracle's on a slice, the count formula's in full."""
    import torch
    rng = np.random.default_rng(6)
    pats = [b"a"] + [bytes(rng.choice(list(b"ab"), size=int(rng.integers(12, 30))).tolist()) for _ in range(80)]
    pats = [p for p in pats if p == b"a" or b"b" in p]
    n_hay, hay_len = 24576, 4096
    text = np.full(4 * hay_len, ord("a"), np.uint8)
    pat_off = np.concatenate([[0], np.cumsum([len(p) for p in pats])]).astype(np.uint64)
    small = synth.Dataset("LOG", np.frombuffer(b"".join(pats), np.uint8).copy(), pat_off, text, (np.arange(5) * hay_len).astype(np.uint64))
    pe, re_ = _oracle(small)
    assert len(re_) == 4 * hay_len                                   # "a" at every byte, nothing else in text without a 'b'
    res = resource_for(small, kernel=6, small_path=0, lds_bytes=2048, chain_walk=1)
    d_blob = torch.full((n_hay * hay_len,), ord("a"), dtype=torch.uint8, device="cuda")
    d_off = torch.arange(n_hay + 1, dtype=torch.int64, device="cuda") * hay_len
    batch.bind_current_stream(res)
    for _ in range(2):
        total = batch.match_device(res, d_blob, d_off, True)
        assert res.stat("kernel_used") == 6 and res.stat("chain_walk_used") == 1 and total == n_hay * hay_len
    assert res.stat("chain_log_reruns") >= 1
    p, r = batch.result_numpy(res, n_hay, total)
    assert np.array_equal(p, np.arange(n_hay + 1, dtype=np.uint64) * np.uint64(hay_len))
    assert np.array_equal(r[:len(re_)], re_)
    tail = r[-hay_len:]
    assert np.all(tail[:, 0] == n_hay - 1) and np.array_equal(tail[:, 1], np.arange(1, hay_len + 1)) and np.all(tail[:, 3] == 0)
    api.ahocorasick_deinit(res)


def test_two_batches_in_flight_and_a_buffer_that_is_too_small():
    import torch
    base = synth.density_set("words")
    ds = synth.plant_density(base, 256, 5000)
    n = min(300, ds.n_haystacks)
    ds = synth.Dataset(ds.name, ds.pat_blob, ds.pat_off, ds.hay_blob[:int(ds.hay_off[n])], ds.hay_off[:n + 1])
    pe, re_ = _oracle(ds)
    res = resource_for(ds, kernel=6, small_path=0)
    d_blob = torch.from_numpy(ds.hay_blob).cuda()
    d_off = torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
    batch.bind_current_stream(res)
    batch.match_device_begin(res, d_blob, d_off, True)
    batch.match_device_begin(res, d_blob, d_off, True)
    for _ in range(2):
        total = batch.match_device_end(res)
        assert res.stat("chain_walk_used") == 1
        p, r = batch.result_numpy(res, ds.n_haystacks, total)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)


def test_bounded_result_memory_drains_and_resumes():
    """"max_result_bytes" through the chain walk: what fits is written, whole events in order; the rest is matched again."""
    import torch
    ds = _nested(n_hay=24, hay_len=2048)
    pe, re_ = _oracle(ds)
    per_hay = int(np.diff(pe).max())
    cap = 3 * per_hay * 16 + 4096
    res = resource_for(ds, kernel=6, small_path=0, lds_bytes=2048, max_result_bytes=cap, chain_walk=1)
    d_blob = torch.from_numpy(ds.hay_blob).cuda()
    d_off = torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
    batch.bind_current_stream(res)
    got = []
    for first, k, total in batch.match_device_bounded(res, d_blob, d_off):
        assert res.stat("chain_walk_used") == 1
        p, r = batch.result_numpy(res, k, total)
        assert int(p[-1]) == total and total * 16 <= cap
        r = r.copy()
        r[:, 0] += first
        got.append(r)
    assert len(got) >= 6 and np.array_equal(np.concatenate(got), re_)
    p, r = host_match(res, ds)
    assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("reserve", [1, 8, 100])
def test_reserved_compute_units(reserve):
    """"reserve_cus" (acmi.h): the walk runs on all but N compute units -- the logs and overflow regions are per wave of the grid that
    runs, so the walk, the emit and the host's sizing have to agree on it."""
    base = synth.density_set("headline")
    ds = synth.plant_density(base, 64, 8192)
    n = min(128, ds.n_haystacks)
    ds = synth.Dataset(ds.name, ds.pat_blob, ds.pat_off, ds.hay_blob[:int(ds.hay_off[n])], ds.hay_off[:n + 1])
    pe, re_ = _oracle(ds)
    res = resource_for(ds, kernel=6, small_path=0, reserve_cus=reserve)
    _check(res, ds, pe, re_, calls=3)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("pairs,tail,expect_chain", [(280, 5, 1), (430, 2, 0)])
def test_wide_rows_take_the_halved_offsets_or_leave_the_chain_walk(pairs, tail, expect_chain):
    """Round 6's LDS layout (k_chain.hip chain_layout): a record names its failure state's row by a 16-bit offset -- whole while the
    rows end below 64 KiB, halved up to 128 KiB -- and a D state has a record of its own.  120 byte values in use (121 classes: rows of
    242 B, TABLE mode): 280 branching depth-2 states put the rows' end at ~ 97 KB (halved offsets), 430 at ~ 133 KB -- the builder
    still makes the image (it fits its budget), the kernel's layout does not hold it and kernel 6 is k_walk_tiles."""
    r = np.random.default_rng(4242 + pairs)
    alpha = np.arange(8, 128, dtype=np.uint8)
    pats = []
    for x in alpha:   # every depth-1 state branches
        pats.append(bytes([int(x), int(alpha[0])]) + bytes(r.choice(alpha, size=tail - 1).tolist()))
    for _ in range(pairs):   # depth-2 states with two children
        head = bytes(r.choice(alpha, size=2).tolist())
        a, b = r.choice(alpha, size=2, replace=False)
        pats.append(head + bytes([int(a)]) + bytes(r.choice(alpha, size=tail).tolist()))
        pats.append(head + bytes([int(b)]) + bytes(r.choice(alpha, size=tail).tolist()))
    pats = list(dict.fromkeys(pats))
    n_hay, hay_len = 96, 3000
    text = r.choice(alpha, size=n_hay * hay_len).astype(np.uint8)
    for k in range(0, len(text) - 16, 97):   # occurrences and near misses
        p = np.frombuffer(pats[int(r.integers(len(pats)))], dtype=np.uint8)
        text[k:k + len(p) - (k // 97) % 2] = p[:len(p) - (k // 97) % 2]
    pb, po = synth.pack(pats)
    ds = synth.Dataset(f"WIDE{pairs}", pb, po, text, (np.arange(n_hay + 1, dtype=np.uint64) * hay_len))
    pe, re_ = _oracle(ds)
    assert len(re_) > 500
    res = resource_for(ds, kernel=6, small_path=0, split=0, lds_bytes=2048)
    for _ in range(2):
        p, rr = device_match(res, ds)
        assert np.array_equal(p, pe) and np.array_equal(rr, re_)
    assert res.stat("chain_image_bytes") > 60_000
    assert res.stat("kernel_used") == 6 and res.stat("chain_walk_used") == expect_chain
    rows_end = res.stat("chain_dense_states") * res.stat("columns") * 2
    assert (65_536 < rows_end <= 131_072) if expect_chain else rows_end > 131_072
    api.ahocorasick_deinit(res)
