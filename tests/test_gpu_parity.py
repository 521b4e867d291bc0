"""Parity of the HIP path with the oracle / reference goldens.  Needs an MI355X: -m gpu.

Everything goes through the C ABI of libacmi.so (acmi_match / acmi_match_batch /
acmi_match_batch_device).  The oracle is only the checker.
"""
import itertools

import numpy as np
import pytest

from helpers import as_pairs, sha
from gpu_util import device_match, host_match, resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, synth

pytestmark = pytest.mark.gpu

SMALL = ["G1", "G2", "G3", "G4", "G5", "R1", "R2", "R3", "R4", "R5", "R6", "R7", "R8"]


def _check_golden(res, ds, golden, name, path=host_match):
    for find_all in (True, False):
        g = golden[(name, find_all)]
        prefix, rec = path(res, ds, find_all)
        assert len(rec) == g["n_records"], (name, find_all)
        assert sha(rec) == g["sha256"], (name, find_all)
        assert sha(prefix) == g["prefix_sha256"], (name, find_all)


@pytest.mark.parametrize("name", SMALL)
def test_golden_default(name, golden):
    """Committed goldens (made by the compiled reference), default configuration, host-buffer entry."""
    ds = synth.NAMED[name]()
    res = resource_for(ds)
    _check_golden(res, ds, golden, name)
    assert api.ahocorasick_deinit(res)


@pytest.mark.parametrize("name", SMALL)
def test_golden_device_entry(name, golden):
    """Same through acmi_match_batch_device (haystacks resident in HBM, torch-allocated)."""
    ds = synth.NAMED[name]()
    res = resource_for(ds)
    _check_golden(res, ds, golden, name, device_match)
    api.ahocorasick_deinit(res)


def test_golden_g6_64k_patterns(golden):
    """Config 3 automaton: 65,536 patterns, 2.27 M states, uint32 entries, 256 columns."""
    ds = synth.g6()
    res = resource_for(ds)
    _check_golden(res, ds, golden, "G6")
    assert res.stat("nodes") == golden[("G6", True)]["nodes"]
    assert res.stat("entry_bytes") == 4
    assert res.stat("compact") == 1 and res.stat("automaton_bytes") < 64 << 20   # 2.3 GB as dense rows
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("kernel", [0, 1, 4])
def test_golden_g6_compact_other_kernels(golden, kernel):
    """The same automaton through the table-walk kernels: every cold step is a goto / failure record."""
    ds = synth.g6()
    res = resource_for(ds, kernel=kernel)
    _check_golden(res, ds, golden, "G6")
    api.ahocorasick_deinit(res)


def test_g6_dense_equals_compact(golden):
    """Dense rows for all 2.27 M states (2.3 GB, the round-1 layout) give the same records."""
    ds = synth.g6()
    res = resource_for(ds, compact=0)
    _check_golden(res, ds, golden, "G6")
    assert res.stat("compact") == 0 and res.stat("table_bytes") > 2e9
    api.ahocorasick_deinit(res)


VARIANTS = [dict(kernel=1, persistent=1, chunk_bytes=64), dict(kernel=1, persistent=0, ilp=2), dict(kernel=2), dict(kernel=2, chunk_bytes=256), dict(kernel=2, run_cap=1, slab_events=1), dict(kernel=2, chunk_bytes=2048, run_cap=2),
            dict(kernel=1, ilp=1), dict(kernel=1, ilp=1, group_quads=4), dict(kernel=1, ilp=2, run_cap=1), dict(kernel=1, ilp=1, run_cap=2, chunk_bytes=512), dict(kernel=0), dict(kernel=1, ilp=1), dict(kernel=1, ilp=2),
            dict(kernel=1, ilp=1, lds_bytes=4096), dict(kernel=1, ilp=2, lds_bytes=2048),
            dict(kernel=1, ilp=1, chunk_bytes=64), dict(kernel=1, ilp=2, chunk_bytes=256),
            dict(kernel=0, chunk_bytes=16), dict(kernel=1, ilp=1, chunk_bytes=4096, slab_events=1),
            dict(kernel=1, ilp=2, slab_events=1), dict(kernel=1, ilp=1, slab_events=3, chunk_bytes=128),
            # both drains of overflowed items on every dataset: one thread per item / one wave per item
            dict(kernel=1, slab_events=1, refill=2, chunk_bytes=512), dict(kernel=2, slab_events=1, refill=1),
            dict(kernel=0, slab_events=1, refill=2, chunk_bytes=48), dict(kernel=2, slab_events=2, refill=2, chunk_bytes=1000),
            dict(kernel=1, slab_events=1, refill=2, max_flat_outputs=0),
            # compact automaton (goto / failure records outside the first trie levels) forced on small sets; automatic
            # only when the dense table would exceed 64 MiB.  dense_bytes=0: dense rows for the tile and depth <= 1 only
            dict(kernel=1, lds_bytes=2048, compact=1, dense_bytes=0), dict(kernel=2, lds_bytes=4096, compact=1, chunk_bytes=256, dense_bytes=0),
            dict(kernel=0, compact=1, dense_bytes=0), dict(kernel=3, compact=1, lds_bytes=8192, run_cap=1, dense_bytes=4096),
            dict(kernel=4, compact=1, dense_bytes=0, lds_bytes=2048, chunk_bytes=128), dict(kernel=1, compact=1, dense_bytes=0, slab_events=1, refill=2, lds_bytes=2048),
            dict(kernel=1, compact=1, dense_bytes=0, slab_events=1, refill=1, lds_bytes=4096, small_path=0),
            # exact walk of whole items, 1 / 2 / 4 items per thread
            dict(kernel=4, walk_group=1), dict(kernel=4, walk_group=4, chunk_bytes=48, slab_events=1), dict(kernel=4, walk_group=8, chunk_bytes=256, lds_bytes=4096),
            dict(kernel=4, walk_group=8, slab_events=2, refill=2),
            # the tile walk (k_walk_tiles): automatic geometry, every piece size, odd lane counts per item, tiny LDS tile + compact
            # records, no flat output lists
            dict(kernel=6, small_path=0), dict(kernel=6, small_path=0, tile_w=16, tile_p=3), dict(kernel=6, small_path=0, tile_w=32, tile_p=64),
            dict(kernel=6, small_path=0, tile_w=64, tile_p=1), dict(kernel=6, small_path=0, tile_w=16, tile_p=64, compact=1, dense_bytes=0, lds_bytes=2048),
            dict(kernel=6, small_path=0, tile_w=64, tile_p=7, max_flat_outputs=0),
            # early stop of single-window anchor walks: forced on / off
            dict(kernel=2, early_stop=2), dict(kernel=2, early_stop=2, chunk_bytes=256, compact=1, dense_bytes=0, lds_bytes=2048), dict(kernel=2, early_stop=0)]


@pytest.mark.parametrize("name,variant", list(itertools.product(["G2", "G4", "G5", "R1", "R2", "R3", "R4", "R5", "R6", "R7", "R8"], range(len(VARIANTS)))))
def test_golden_variants(name, variant, golden):
    """Every kernel / ILP / LDS-tile size (forces cold states + trap) / chunk size (forces warm-up
    overlap, R3 has 1 KiB patterns) / slab size (forces the overflow -> refill path)."""
    ds = synth.NAMED[name]()
    res = resource_for(ds, **VARIANTS[variant])
    _check_golden(res, ds, golden, name)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("refill,flat", [(0, 1), (1, 1), (2, 1), (2, 0)])
def test_overflow_path_is_taken(refill, flat):
    ds = synth.g3(n_hay=3, hay_len=4096)
    res = resource_for(ds, slab_events=2, refill=refill, **({} if flat else {"max_flat_outputs": 0}))
    prefix, rec = host_match(res, ds)
    assert (res.stat("flat_outputs") > 0) == bool(flat)
    assert res.stat("overflow_items") > 0
    assert len(rec) == 3 * (16 * 4096 - 120)  # NH * (K*HL - K(K-1)/2), SURVEY.md 8d config 5
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    _, exp = o.match_batch(ds.hay_blob, ds.hay_off)
    assert np.array_equal(rec, exp)
    api.ahocorasick_deinit(res)


def test_automatic_choice_of_exact_walk(golden):
    """Automatic kernel choice.  Dense sets (all strings <= 4 over {a,b}) start on the exact whole-item walk:
    the builder's estimate says the LDS-tile kernel's optimistic pass would flag every work item.  A sparse set
    starts on the optimistic kernel, leaves it when a batch of >= 1 MiB shows that most bytes were walked exactly
    anyway, and probes it again later; results never change."""
    ds = synth.g4()
    res = resource_for(ds)
    api.ahocorasick_finalize(res)
    assert res.stat("est_flag_rate") > 1
    first = device_match(res, ds)
    assert res.stat("kernel_used") == 4
    _check_golden(res, ds, golden, "G4")
    api.ahocorasick_deinit(res)

    # 8 patterns of 6 letters: sparse on uniform text (8 / 26^6 events per byte), but this text is ALL pattern
    ds = synth.generate("sw", 77, 8, 6, 6, 1, 64, bytes(range(97, 123)))
    res = resource_for(ds, anchor_dense=0)   # (with the dense anchor form these patterns take kernel 2: second half of the test)
    api.ahocorasick_finalize(res)
    assert res.stat("est_flag_rate") * 8192 < 1 and res.stat("anchor_q") == 0
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    hay = np.frombuffer(b"".join(ds.pattern(k % 8) for k in range(8192 // 6 * 256 + 300)), dtype=np.uint8)[:256 * 8192].copy()
    off = (np.arange(257, dtype=np.uint64) * 8192)
    pe, re_ = o.match_batch(hay, off, True)
    import torch
    from php_aho_corasick_amd import batch
    d_blob, d_off = torch.from_numpy(hay).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
    batch.bind_current_stream(res)
    used = []
    for _ in range(6):
        total = batch.match_device(res, d_blob, d_off, True)
        used.append(int(res.stat("kernel_used")))
        p, r = batch.result_numpy(res, 256, total)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
    # measured choice: the optimistic kernel first (what the pattern set suggests), then -- it cost several times what an exact
    # walk is expected to -- the exact walks are probed and one of them kept
    assert used[0] == 1 and used[-1] in (4, 6) and res.stat("choice_switches") >= 1, used
    api.ahocorasick_deinit(res)
    # the same set with the dense anchor form (shortest pattern 6 B: windows at every byte offset): this text has a hit window
    # every six bytes, the window lists overflow, whole items are walked -- the handle leaves the filter after the first batch
    res = resource_for(ds)
    api.ahocorasick_finalize(res)
    assert res.stat("anchor_q") == 4 and res.stat("anchor_dense") == 1
    batch.bind_current_stream(res)
    used = []
    for _ in range(6):
        total = batch.match_device(res, d_blob, d_off, True)
        used.append(int(res.stat("kernel_used")))
        p, r = batch.result_numpy(res, 256, total)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
    assert used[0] == 2 and used[-1] != 2, used
    api.ahocorasick_deinit(res)


def test_incremental_adds_equal_single_init(golden):
    """tests/test3.phpt's property on a big set: N add_patterns calls vs one init differ only by the
    duplicate rule, which the oracle reproduces call by call."""
    ds = synth.r1()
    res = resource_for(ds, calls=4)
    o = Oracle("port")
    pats = ds.patterns()
    step = (len(pats) + 3) // 4
    for lo in range(0, len(pats), step):
        o.add_patterns(pats[lo:lo + step])
    for find_all in (True, False):
        p, r = host_match(res, ds, find_all)
        pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)


def test_kat_through_php_surface(kat):
    """The reference's .phpt cases through the mirrored PHP function surface."""
    for case in kat:
        res = api.ahocorasick_init(case["init"])
        assert res is not False
        for call in case["adds"]:
            assert api.ahocorasick_add_patterns(res, call) is True
        reps = 3 if case["name"] == "KAT-6" else 1
        for _ in range(reps):
            for q in case["queries"]:
                if "findAll" in q:
                    got = api.ahocorasick_match(q["haystack"], res, q["findAll"])
                else:
                    got = api.ahocorasick_match(q["haystack"], res)
                assert [as_pairs(r) for r in got] == q["expect"], (case["name"], q["haystack"])
        if "isValid_before_deinit" in case:
            assert api.ahocorasick_isValid(res) is case["isValid_before_deinit"]
        if case.get("deinit"):
            assert api.ahocorasick_deinit(res) is True
            if "isValid_after_deinit" in case:
                assert api.ahocorasick_isValid(res) is False
                assert api.ahocorasick_deinit(res) is case["second_deinit"]


def test_kat6_init_match_deinit_cycles(kat):
    """tests/test4.phpt shape (scaled down): repeated init / many matches / deinit stays exact."""
    case = next(c for c in kat if c["name"] == "KAT-6")
    q = case["queries"][0]
    for _ in range(4):
        res = api.ahocorasick_init(case["init"])
        for _ in range(50):
            d = api.ahocorasick_match(q["haystack"], res)
            assert d and len(d) == 4
        assert api.ahocorasick_deinit(res)


def test_batch_equals_foreach(kat):
    """ahocorasick_match_batch (new) == foreach ahocorasick_match (examples/benchmark.php:65-67)."""
    case = next(c for c in kat if c["name"] == "KAT-3/4")
    res = api.ahocorasick_init(case["init"])
    hays = [q["haystack"] for q in case["queries"]] + ["", "aux", "éé"]
    each = [api.ahocorasick_match(h, res) for h in hays]
    assert api.ahocorasick_match_batch(hays, res) == each
    assert api.ahocorasick_match_batch(hays, res, False) == [api.ahocorasick_match(h, res, False) for h in hays]
    api.ahocorasick_deinit(res)


def test_empty_inputs():
    res = api.ahocorasick_init([])
    assert api.ahocorasick_match("anything", res) == []          # empty automaton
    assert api.ahocorasick_match_batch([], res) == []
    api.ahocorasick_deinit(res)
    res = api.ahocorasick_init([{"value": "a"}])
    assert api.ahocorasick_match("", res) == []                  # empty haystack
    assert api.ahocorasick_match_batch(["", "", "a", ""], res) == [[], [], [{"pos": 1, "start_postion": 0, "value": "a"}], []]
    api.ahocorasick_deinit(res)


def test_bytes_nul_and_high():
    """Opaque bytes: NUL and >= 0x80 inside patterns and haystacks."""
    pats = [{"value": b"\x00\xff"}, {"value": b"\xff"}, {"id": 7, "value": b"a\x00b"}]
    res = api.ahocorasick_init(pats)
    o = Oracle("port")
    o.add_patterns([p["value"] for p in pats])
    text = b"xa\x00b\x00\xff\xff\x00"
    got = api.ahocorasick_match(text, res)
    exp = o.match(text)
    assert [(r["pos"], r["start_postion"]) for r in got] == [(int(r[1]), int(r[2])) for r in exp]
    assert got[0] == {"pos": 4, "keyIdx": 7, "start_postion": 1, "value": b"a\x00b"}
    api.ahocorasick_deinit(res)


def test_unaligned_device_blob():
    """Haystack blob that starts at an odd device address (head/tail byte paths)."""
    import torch
    from php_aho_corasick_amd import batch
    ds = synth.r2()
    res = resource_for(ds)
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off)
    for shift in (1, 7, 13):
        raw = torch.zeros(ds.hay_blob.size + 64, dtype=torch.uint8, device="cuda")
        view = raw[shift:shift + ds.hay_blob.size]
        view.copy_(torch.from_numpy(ds.hay_blob))
        d_off = torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
        batch.bind_current_stream(res)
        total = batch.match_device(res, view, d_off)
        p, r = batch.result_numpy(res, ds.n_haystacks, total)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)


def test_random_against_oracle():
    """Seeded random automata/haystacks at sizes the oracle finishes instantly, all entry points."""
    for seed in range(60, 72):
        alpha = [b"ab", b"abcdefgh", b"ACGT", None, b"xyz\x00\x01\x02\x03\x04\x05"][seed % 5]
        ds = synth.ragged(f"S{seed}", seed, 20 + 13 * (seed % 7), 1, 3 + seed % 9, 33 + seed, 0, 700 + 97 * (seed % 4), alpha)
        o = Oracle("port")
        o.add_patterns((ds.pat_blob, ds.pat_off))
        res = resource_for(ds, kernel=[1, 2, 0, 3][seed % 4], ilp=1 + seed % 2, chunk_bytes=[0, 48, 512][seed % 3], slab_events=[8, 1, 2][seed % 3])
        for find_all in (True, False):
            pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
            for path in (host_match, device_match):
                p, r = path(res, ds, find_all)
                assert np.array_equal(p, pe), (seed, find_all, path.__name__)
                assert np.array_equal(r, re_), (seed, find_all, path.__name__)
        api.ahocorasick_deinit(res)
        o.close()


def test_device_entry_rejects_bad_offsets():
    """Device-resident offsets cannot be checked on the host: the pipeline flags them itself."""
    import torch
    from php_aho_corasick_amd import batch
    from php_aho_corasick_amd._native import AcmiError, E_ARG
    res = api.ahocorasick_init([{"value": "abc"}])
    d_blob = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    batch.bind_current_stream(res)
    good = torch.tensor([0, 100, 100, 4096], dtype=torch.int64, device="cuda")
    assert batch.match_device(res, d_blob, good) == 0
    bad = torch.tensor([0, 200, 100, 4096], dtype=torch.int64, device="cuda")
    with pytest.raises(AcmiError) as e:
        batch.match_device(res, d_blob, bad)
    assert e.value.status == E_ARG
    assert batch.match_device(res, d_blob, good) == 0      # the handle stays usable
    api.ahocorasick_deinit(res)


def test_automatic_kernel_choice_and_fallback():
    """kernel=3 (default): anchor filter for long-pattern sets, LDS-tile walk otherwise, and a handle
    leaves the filter for good once a batch defeats it (text made of pattern fragments)."""
    ds = synth.g2()
    res = resource_for(ds)
    host_match(res, ds)
    assert res.stat("kernel_used") == 2
    api.ahocorasick_deinit(res)
    res = resource_for(synth.r1())
    host_match(res, synth.r1())
    assert res.stat("kernel_used") in (4, 6)             # short patterns: not eligible, and dense by the builder's estimate: an exact walk
    api.ahocorasick_deinit(res)
    # adversarial text: every haystack is a pattern's first 15 bytes repeated -> anchors everywhere
    # (2 MiB: batches under 1 MiB decide nothing)
    ds = synth.g2(256)
    frag = np.frombuffer(ds.pattern(5)[:15] * 600, dtype=np.uint8)[:8192]
    ds.hay_blob[:] = np.tile(frag, 256)
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off)
    res = resource_for(ds)
    # measured choice: the filter costs this text many times what the other strategies are expected to -- they are probed, the
    # cheapest is kept; the filter is probed again on its back-off (16, 32, ... batches), loses again, and the results never change
    used = []
    for _ in range(60):
        p, r = host_match(res, ds)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
        used.append(int(res.stat("kernel_used")))
    # (round 6: the filter's hand-over first learns this text -- lists, item size, the wave kernels: two batches that decide nothing --
    # and then costs 0.09 ms per call instead of 5.7; the whole-automaton walk still wins at 0.05)
    assert used[0] == 2 and used[12] != 2 and used[-1] != 2, used
    assert used[8:].count(2) <= 3 and res.stat("choice_switches") >= 1 and res.stat("choice_probes") >= 2, used
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("chunk", [0, 256])
def test_anchor_overlapping_occurrences_of_unequal_length(chunk):
    """The anchor path starts a walk only m_min-1 bytes before a run where it can prove that enough, and
    merges or clamps runs of neighbouring windows.  Long and short patterns that overlap, nest and share
    suffixes, in periodic text where nearly every window hits, must still give the oracle's records."""
    pats = [b"a" * 11, b"a" * 40 + b"b", b"ab" * 30, b"b" + b"a" * 25, b"abba" * 6, b"a" * 13 + b"ba" * 20,
            b"bbbbbbbbbbbb", b"ab" * 8 + b"c" + b"ab" * 8, b"c" + b"a" * 60 + b"c", b"a" * 12 + b"c"]
    rng = np.random.default_rng(7)
    hays = []
    for h in range(40):
        parts = []
        for _ in range(int(rng.integers(1, 30))):
            k = int(rng.integers(0, 6))
            parts.append([b"a" * int(rng.integers(1, 90)), b"b", b"ab" * int(rng.integers(1, 40)), b"c",
                          pats[int(rng.integers(len(pats)))], b"ba" * int(rng.integers(1, 25))][k])
        hays.append(b"".join(parts)[: int(rng.integers(0, 3000))])
    pat_blob, pat_off = synth.pack(pats)
    hay_blob, hay_off = synth.pack(hays)
    ds = synth.Dataset("overlap", pat_blob, pat_off, hay_blob, hay_off)
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    for run_cap in (8, 2):
        res = resource_for(ds, kernel=2, chunk_bytes=chunk, run_cap=run_cap, slab_events=2)
        for find_all in (True, False):
            pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
            for path in (host_match, device_match):
                p, r = path(res, ds, find_all)
                assert np.array_equal(p, pe) and np.array_equal(r, re_), (chunk, run_cap, find_all, path.__name__)
        assert res.stat("kernel_used") == 2 and res.stat("anchor_q") == 8
        api.ahocorasick_deinit(res)
    o.close()


@pytest.mark.parametrize("name", ["R1", "R4", "R5", "G5"])
def test_stream_feed_matches_keep_mode(name):
    """acmi_stream_feed (carry of Lmax-1 bytes + scan from the root) against the oracle's keep=1 mode:
    pieces shorter than the longest pattern, empty pieces, matches straddling several pieces."""
    from php_aho_corasick_amd.api import AhoStream
    ds = synth.NAMED[name]()
    res = resource_for(ds)
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    rng = np.random.default_rng(11)
    checked = 0
    for h in range(min(ds.n_haystacks, 12)):
        text = ds.haystack(h)
        for sizes in ((1, 4), (1, 40), (200, 3000)):
            s = AhoStream(res)
            at, first = 0, True
            while at < len(text) or first:
                n = int(rng.integers(sizes[0], sizes[1] + 1)) if rng.integers(0, 8) else 0
                piece = text[at:at + n]
                want = o.stream_feed(piece, first)
                got = s.feed(piece)
                assert np.array_equal(got, want), (name, h, sizes, at, n)
                checked += len(want)
                at += n
                first = False
            assert s.position == len(text)
            s.close()
    assert checked > 0
    api.ahocorasick_deinit(res)
    o.close()


def _slice(ds, max_hay, max_bytes):
    """First haystacks of a dataset that fit the one-launch path's limits."""
    n = 0
    while n < min(ds.n_haystacks, max_hay) and int(ds.hay_off[n + 1]) <= max_bytes:
        n += 1
    end = int(ds.hay_off[n])
    return synth.Dataset(ds.name + "-slice", ds.pat_blob, ds.pat_off, ds.hay_blob[:end].copy(), ds.hay_off[:n + 1].copy())


@pytest.mark.parametrize("name", ["G2", "G3", "G4", "G5", "R1", "R2", "R3", "R4", "R5", "R6", "R7", "R8"])
def test_small_batch_one_launch_path(name):
    """Host batches of <= 64 KiB take k_match_small (one launch on pinned memory); dense ones whose records
    exceed its buffer fall back to the pipeline.  Both must give the oracle's records, and each other's."""
    full = synth.NAMED[name]()
    o = Oracle("port")
    o.add_patterns((full.pat_blob, full.pat_off))
    res = resource_for(full)
    for max_hay, max_bytes in ((1024, 65536), (3, 65536), (1, 9000)):
        ds = _slice(full, max_hay, max_bytes)
        if ds.n_haystacks == 0:
            continue
        for find_all in (True, False):
            pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
            res.set_option("small_path", 1)
            p1, r1 = host_match(res, ds, find_all)
            used = res.stat("kernel_used")
            res.set_option("small_path", 0)
            p0, r0 = host_match(res, ds, find_all)
            assert res.stat("kernel_used") != 5
            assert np.array_equal(p1, pe) and np.array_equal(r1, re_), (name, max_hay, find_all, "small", used)
            assert np.array_equal(p0, pe) and np.array_equal(r0, re_), (name, max_hay, find_all, "pipeline")
            assert used == (5 if len(re_) <= 8192 else used)
    api.ahocorasick_deinit(res)
    o.close()


def test_255_used_byte_values_are_not_identity_classes():
    """255 distinct byte values in the patterns give 256 table columns with column 0 = "every other byte":
    the byte -> column map is NOT the identity there (found by tools/fuzz_gpu.py seed 50771)."""
    missing = 0x41
    vals = [b for b in range(256) if b != missing]
    pats = [bytes(vals[i:i + 16]) for i in range(0, 255 - 15, 7)] + [bytes(vals[-16:])]
    hay = bytes([missing]) * 5 + pats[3] + bytes([missing, 0, 255]) + pats[-1] + pats[0][:9] + bytes([missing]) + pats[10]
    pat_blob, pat_off = synth.pack(pats)
    hay_blob, hay_off = synth.pack([hay, hay[::-1], hay * 3])
    ds = synth.Dataset("c255", pat_blob, pat_off, hay_blob, hay_off)
    o = Oracle("port")
    o.add_patterns(pats)
    for opts in (dict(), dict(kernel=1), dict(kernel=4), dict(kernel=0), dict(small_path=0)):
        res = resource_for(ds, **opts)
        for find_all in (True, False):
            pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
            assert len(re_) > 0
            for path in (host_match, device_match):
                p, r = path(res, ds, find_all)
                assert np.array_equal(p, pe) and np.array_equal(r, re_), (opts, find_all, path.__name__)
        assert res.stat("columns") == 256
        api.ahocorasick_deinit(res)
    o.close()


@pytest.mark.parametrize("kernel,items_ahead", [(1, 0), (2, 0), (3, 0), (2, 1), (3, 1), (4, 1)])
def test_two_batches_in_flight(kernel, items_ahead):
    """acmi_match_batch_device_begin twice before the first _end: the batches end in the order they were begun, each
    with its own records (== oracle), a third begin is refused, and a batch whose records outgrow the capacity the
    handle expected -- after a later begin has reused the pipeline's intermediates -- still comes back whole."""
    import torch
    from php_aho_corasick_amd import batch
    from php_aho_corasick_amd._native import AcmiError, E_ARG
    sets = [synth.ragged(f"F{seed}", seed, 40, 12, 20, 300 + 50 * seed, 100, 3000, b"abcd") for seed in (1, 2, 3)]
    # one pattern set for all batches: batch k is scanned with the patterns of set 0
    base = sets[0]
    o = Oracle("port")
    o.add_patterns((base.pat_blob, base.pat_off))
    # items_ahead: every batch's work-item table is built on the handle's side stream, next to the batch begun before it
    res = resource_for(base, kernel=kernel, items_ahead=items_ahead)
    batch.bind_current_stream(res)
    dev, want = [], []
    for ds in sets:
        dev.append((torch.from_numpy(ds.hay_blob.copy()).cuda(), torch.from_numpy(ds.hay_off.astype(np.int64)).cuda(), len(ds.hay_off) - 1))
        want.append(o.match_batch(ds.hay_blob, ds.hay_off, True))
    # a dense batch: far more records than the first batches made the handle expect (capacity grows at its end)
    nh_d = 64
    dense_blob = np.frombuffer(bytes(base.pattern(0)) * (nh_d * 4096 // 12 + 1), dtype=np.uint8)[:nh_d * 4096].copy()
    dense_off = (np.arange(nh_d + 1, dtype=np.uint64) * 4096)
    dev.append((torch.from_numpy(dense_blob).cuda(), torch.from_numpy(dense_off.astype(np.int64)).cuda(), nh_d))
    want.append(o.match_batch(dense_blob, dense_off, True))

    def check(k, total):
        p, r = batch.result_numpy(res, dev[k][2], total)
        assert total == len(want[k][1]), k
        assert np.array_equal(p, want[k][0]), k
        assert np.array_equal(r, want[k][1]), k

    order = [0, 1, 2, 3, 0, 3, 3, 1]
    began = ended = 0
    while ended < len(order):
        while began < len(order) and began - ended < 2:
            batch.match_device_begin(res, dev[order[began]][0], dev[order[began]][1], True)
            began += 1
        if began - ended == 2:
            with pytest.raises(AcmiError) as e:   # a third batch is refused, the two in flight are not disturbed
                batch.match_device_begin(res, dev[0][0], dev[0][1], True)
            assert e.value.status == E_ARG
        check(order[ended], batch.match_device_end(res))
        ended += 1
    with pytest.raises(AcmiError):
        batch.match_device_end(res)               # nothing left in flight
    # a synchronous call drops what was begun and never ended
    batch.match_device_begin(res, dev[1][0], dev[1][1], True)
    check(2, batch.match_device(res, dev[2][0], dev[2][1], True))
    with pytest.raises(AcmiError):
        batch.match_device_end(res)
    api.ahocorasick_deinit(res)
    o.close()


def test_automatic_choice_small_haystacks():
    """Automatic mode takes the LDS-tile walk instead of the anchor filter when haystacks average under 192 B (the anchor
    kernel spends a row of 64 lanes per work item at least): same records, other kernel; big haystacks keep the filter."""
    import torch
    from php_aho_corasick_amd import batch
    ds = synth.generate("SH", 5, 256, 16, 16, 16384, 128, b"abcdef")       # 2 MiB of 128-byte haystacks
    big = synth.generate("SB", 5, 256, 16, 16, 256, 8192, b"abcdef")       # 2 MiB of 8-KiB haystacks, same patterns
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    res = resource_for(ds)
    batch.bind_current_stream(res)
    for data, kernel in ((ds, 1), (big, 2), (ds, 1)):
        d_blob = torch.from_numpy(data.hay_blob.copy()).cuda()
        d_off = torch.from_numpy(data.hay_off.astype(np.int64)).cuda()
        total = batch.match_device(res, d_blob, d_off, True)
        assert int(res.stat("kernel_used")) == kernel
        p, r = batch.result_numpy(res, len(data.hay_off) - 1, total)
        pe, re_ = o.match_batch(data.hay_blob, data.hay_off, True)
        assert np.array_equal(p, pe) and np.array_equal(r, re_)
    api.ahocorasick_deinit(res)
    o.close()


@pytest.mark.parametrize("short_rows", [0, 1])
@pytest.mark.parametrize("plen", [7, 11, 16, 24])
def test_anchor_occurrences_at_and_across_haystack_borders(plen, short_rows):
    """The anchor path scans from the aligned address BELOW an unaligned haystack (bytes of the haystack before it) and
    through the aligned block behind its end: a pattern that straddles two haystacks must not be reported, one that
    begins with a haystack's first byte or ends with its last must be."""
    rng = np.random.default_rng(plen)
    pats = [bytes(rng.integers(97, 103, plen, dtype=np.uint8)) for _ in range(40)]
    hays = []
    for k in range(300):
        n = int(rng.integers(0, 700))
        body = bytearray(rng.integers(97, 103, n, dtype=np.uint8).tobytes())
        p = pats[k % len(pats)]
        mode = k % 6
        if mode == 0 and n >= plen:
            body[:plen] = p                      # begins with the haystack
        elif mode == 1 and n >= plen:
            body[n - plen:] = p                  # ends with the haystack
        elif mode == 2 and n >= plen // 2:
            body[n - plen // 2:] = p[:plen // 2]  # first half at the end: the next haystack gets the rest
        if hays and (k - 1) % 6 == 2 and n >= plen - plen // 2:
            q = pats[(k - 1) % len(pats)]
            body[:plen - plen // 2] = q[plen // 2:]
        hays.append(bytes(body))
    blob = np.frombuffer(b"".join(hays), dtype=np.uint8).copy()
    off = np.zeros(len(hays) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(h) for h in hays])
    pat_blob, pat_off = synth.pack(pats)
    ds = synth.Dataset("borders", pat_blob, pat_off, blob, off)
    o = Oracle("port")
    o.add_patterns((pat_blob, pat_off))
    res = resource_for(ds, kernel=2, short_rows=short_rows)
    api.ahocorasick_finalize(res)
    assert int(res.stat("anchor_q")) > 0
    for find_all in (True, False):
        pe, re_ = o.match_batch(blob, off, find_all)
        assert len(re_) > 50
        for path in (host_match, device_match):
            p, r = path(res, ds, find_all)
            assert np.array_equal(p, pe) and np.array_equal(r, re_), (find_all, path.__name__)
    api.ahocorasick_deinit(res)
    o.close()


def test_uniform_batch_entry_equals_the_offsets_entry():
    """acmi_match_batch_device_begin_uniform: haystacks of one length, back to back, no offsets array -- the same records as the
    general entry for one item per haystack, chunked haystacks, tiny ones, a geometry that changes between two batches in flight,
    and findAll = false."""
    import torch
    from php_aho_corasick_amd import batch
    ds = synth.g2()
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    blob = ds.hay_blob
    d_blob = torch.from_numpy(blob).cuda()
    for options in ({}, dict(kernel=1), dict(kernel=6), dict(chunk_bytes=1024), dict(kernel=4, chunk_bytes=256)):
        res = resource_for(ds, **options)
        batch.bind_current_stream(res)
        shapes = [(8192, 256), (4096, 512), (256, 8192), (8192, 100), (65536, 32)]
        want = {}
        for hl, nh in shapes:
            off = np.arange(nh + 1, dtype=np.uint64) * np.uint64(hl)
            want[(hl, nh)] = o.match_batch(blob[:hl * nh], off, True)
        # two in flight, the geometry changing from batch to batch
        began = ended = 0
        while ended < len(shapes):
            while began < len(shapes) and began - ended < 2:
                batch.match_device_begin_uniform(res, d_blob, shapes[began][0], shapes[began][1], True)
                began += 1
            total = batch.match_device_end(res)
            hl, nh = shapes[ended]
            p, r = batch.result_numpy(res, nh, total)
            assert np.array_equal(p, want[(hl, nh)][0]) and np.array_equal(r, want[(hl, nh)][1]), (options, hl, nh)
            ended += 1
        batch.match_device_begin_uniform(res, d_blob, 8192, 256, False)
        total = batch.match_device_end(res)
        pf, rf = o.match_batch(blob, ds.hay_off, False)
        p, r = batch.result_numpy(res, 256, total)
        assert np.array_equal(p, pf) and np.array_equal(r, rf)
        api.ahocorasick_deinit(res)
    o.close()


def test_uniform_batches_keep_their_item_table():
    """A slot that holds the item table of the same uniform shape keeps it (no k_items_uniform: k_walk_runs writes every item's
    counters): the same shape again with OTHER text -- events in other haystacks, in none, findAll on and off --, other shapes in
    between, an unaligned blob, two batches in flight: always the records of the general entry."""
    import torch
    from php_aho_corasick_amd import batch
    ds = synth.g2()
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    texts = [ds.hay_blob.copy() for _ in range(3)]
    texts[1][:] = np.roll(ds.hay_blob, 4099)          # the plants elsewhere
    texts[2][:] = ord("x")                            # nothing to find
    n_bytes = len(ds.hay_blob)
    d_texts = [torch.from_numpy(np.concatenate([np.zeros(48, np.uint8), t])).cuda() for t in texts]   # + room for an unaligned view
    res = resource_for(ds, kernel=2)
    batch.bind_current_stream(res)
    plan = [(0, 8192, 0, True), (1, 8192, 0, True), (2, 8192, 0, True), (0, 8192, 0, False), (1, 4096, 0, True), (1, 8192, 0, True),
            (0, 8192, 16, True), (1, 8192, 16, True), (2, 8192, 16, False), (0, 8192, 0, True), (0, 8192, 0, True), (1, 8192, 0, True)]
    want = []
    for t, hl, shift, fa in plan:
        nh = (n_bytes - 64) // hl
        src = np.concatenate([np.zeros(48, np.uint8), texts[t]])[48 + shift: 48 + shift + hl * nh]
        want.append(o.match_batch(src, np.arange(nh + 1, dtype=np.uint64) * np.uint64(hl), fa))
    began = ended = 0
    while ended < len(plan):
        while began < len(plan) and began - ended < 2:
            t, hl, shift, fa = plan[began]
            nh = (n_bytes - 64) // hl
            batch.match_device_begin_uniform(res, d_texts[t][48 + shift:], hl, nh, fa)
            began += 1
        total = batch.match_device_end(res)
        t, hl, shift, fa = plan[ended]
        nh = (n_bytes - 64) // hl
        p, r = batch.result_numpy(res, nh, total)
        assert np.array_equal(p, want[ended][0]) and np.array_equal(r, want[ended][1]), (ended, plan[ended], total, len(want[ended][1]))
        ended += 1
    assert res.stat("item_tables_kept") >= 3
    api.ahocorasick_deinit(res)
    o.close()
