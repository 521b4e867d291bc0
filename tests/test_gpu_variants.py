"""Targeted parity for the item-shape variants of k_scan_anchor -- TINY (items of one row, eight to a wave step), SHORT (the
rows an item lacks are skipped), BIG (LDS bitmap + L2 fingerprint table) and BIG x TINY -- on haystacks of exactly those
shapes (synth.TARGETED: regular 256 B / 512 B / 1 KiB, ragged <= 1 KiB, 3 KiB and 2,500 B; plants that begin at a row start,
end at a row / item end and straddle a row border), against goldens made by the compiled reference
(tests/golden/golden_targeted.json, tests/golden/make_golden.py).  `anchor_variant` and `tiny_waves` (acmi_get_stat) prove
which template and which wave path ran.  K1..K3: pattern sets whose shortest pattern is 4..6 B -- the filter's dense form
(bit 64)."""
import numpy as np
import pytest

from helpers import load_json, sha
from gpu_util import device_match, host_match, resource_for
from php_aho_corasick_amd import api, synth

pytestmark = pytest.mark.gpu

BIG, SHORT, TINY, DENSE = 8, 16, 32, 64


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


# dataset -> (bits the automatic choice must set, bits it must not set, tiny waves expected)
EXPECT = {"T256": (TINY | SHORT, BIG, True), "T512": (TINY | SHORT, BIG, True), "T1K": (TINY | SHORT, BIG, True), "TRAG": (TINY | SHORT, BIG, True),
          "S3K": (SHORT, BIG | TINY, False), "S2K5": (SHORT, BIG | TINY, False),
          "B512": (BIG | TINY, 0, True), "BRAG": (BIG | TINY, 0, True), "B3K": (BIG, TINY, False),
          # shortest pattern 4..6 B: the dense form (windows at every byte offset), on long, ragged and tiny haystacks
          "K1": (DENSE | SHORT | 1, BIG | TINY, False), "K2": (DENSE | SHORT | 1, BIG | TINY, False), "K3": (DENSE | SHORT | TINY | 1, BIG, True)}


@pytest.mark.parametrize("name", list(EXPECT))
@pytest.mark.parametrize("path", [host_match, device_match], ids=["host", "device"])
def test_targeted_goldens_automatic_variant(name, path, targeted):
    ds = synth.TARGETED[name]()
    res = resource_for(ds)
    _check(res, ds, targeted, name, path)
    must, must_not, tiny = EXPECT[name]
    v = int(res.stat("anchor_variant"))
    assert res.stat("kernel_used") == 2 and v & must == must and v & must_not == 0, (name, v)
    assert (res.stat("tiny_waves") > 0) == tiny, (name, res.stat("tiny_waves"))
    api.ahocorasick_deinit(res)


OPTIONS = [dict(tiny_rows=0), dict(tiny_rows=0, short_rows=0), dict(short_rows=1), dict(wave_items=1), dict(wave_items=7, persistent=1),
           dict(wave_items=64, persistent=0), dict(chunk_bytes=512), dict(chunk_bytes=1024, run_cap=1), dict(slab_events=1, refill=2),
           dict(verify=0), dict(kernel=1), dict(kernel=4), dict(kernel=0), dict(run_cap=3), dict(run_cap=200, chunk_bytes=256), dict(anchor_dense=0),
           dict(short_rows=0, tiny_rows=0, persistent=1), dict(dense_hot=1), dict(dense_hot=0), dict(anchor_slots2=0), dict(anchor_slots2=0, dense_hot=1), dict(anchor_slots2=1, dense_hot=0, verify=0), dict(dense_hot=1, tiny_rows=0, chunk_bytes=2048), dict(early_stop=2, compact=1, dense_bytes=0, lds_bytes=4096)]


@pytest.mark.parametrize("name", list(EXPECT))
@pytest.mark.parametrize("opt", range(len(OPTIONS)))
def test_targeted_goldens_every_option(name, opt, targeted):
    """The same goldens with the variants switched off / forced, other wave-batch sizes, chunking below the haystack size,
    overflowing run lists and slabs, walked instead of verified candidates, and through the other scan kernels."""
    ds = synth.TARGETED[name]()
    options = dict(OPTIONS[opt])
    options.setdefault("kernel", 2)
    res = resource_for(ds, **options)
    _check(res, ds, targeted, name, device_match)
    v = int(res.stat("anchor_variant"))
    if options["kernel"] != 2:
        assert v == 0
    elif options.get("tiny_rows") == 0:
        assert v & TINY == 0 and res.stat("tiny_waves") == 0
    api.ahocorasick_deinit(res)


def test_tiny_waves_fall_back_when_an_item_is_longer_than_a_row():
    """A batch that averages under 1 KiB takes the TINY template, but a wave whose 64 items include a longer one takes the usual
    path: 63 haystacks of 200 B and one of 5,000 B per wave batch.  Same records as the oracle either way."""
    from oracle.pyoracle import Oracle
    base = synth.t256()
    rng = np.random.default_rng(9)
    hays = []
    for h in range(640):
        n = 5000 if h % 64 == 17 else 200
        t = rng.choice(np.frombuffer(b"abcdef", dtype=np.uint8), size=n)
        p = np.frombuffer(base.pattern((13 * h + 1) % base.n_patterns), dtype=np.uint8)
        at = (n - 16) if h % 3 == 0 else (0 if h % 3 == 1 else n // 2)
        t[at:at + 16] = p
        hays.append(t.tobytes())
    hay_blob, hay_off = synth.pack(hays)
    ds = synth.Dataset("mix", base.pat_blob, base.pat_off, hay_blob, hay_off)
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    for find_all in (True, False):
        pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
        for wave_items in (0, 64, 16):
            res = resource_for(ds, kernel=2, wave_items=wave_items)
            prefix, rec = device_match(res, ds, find_all)
            assert np.array_equal(prefix, pe) and np.array_equal(rec, re_), (find_all, wave_items)
            assert int(res.stat("anchor_variant")) & TINY
            if wave_items == 16:   # batches of 16 items: three in four have no long item and take the tiny path
                assert res.stat("tiny_waves") > 0
            api.ahocorasick_deinit(res)


def test_dense_form_items_finished_by_the_deciding_wave():
    """k_dense_decide finishes an item itself when every hit window is decided there (k_walk.hip): occurrences whose ends do not
    ascend with their starts (a long pattern, shorter ones inside it), several that end at the same byte (one event, the longest's
    state: node.c:424-441 lists the others behind it), an occurrence that ends in the warm-up of the next work item, one at the
    blob's last byte -- against the oracle, with the lists' entries left to k_dense_emit where a window holds two patterns."""
    from oracle.pyoracle import Oracle
    pats = [b"abcdefghijkl", b"bcdefg", b"cdef", b"fghijkl", b"hijkl", b"mnop", b"mnopq", b"wxyz", b"ghij", b"efghi"]
    r = np.random.default_rng(5)
    unit = b"abcdefghijkl....mnopq..wxyz.abcdefg.hijkl..efghij.mnop."
    hays = []
    for k in range(400):
        body = bytearray()
        while len(body) < 9000:
            body += unit if r.integers(3) else bytes(r.integers(97, 123, size=int(r.integers(1, 40)), dtype=np.uint8))
        hays.append(bytes(body[:int(r.integers(4090, 9000))]) + b"abcdefghijkl")
    # (with "mnopq" next to "mnop" almost every item has a window of two patterns and goes to k_dense_emit; without it none has)
    for pset in (pats, [q for q in pats if q != b"mnopq"]):
        ds = synth.Dataset("dense_waves", *synth.pack(pset), *synth.pack(hays))
        o = Oracle("port")
        o.add_patterns(pset)
        for opts in (dict(kernel=2, anchor_dense=1, small_path=0), dict(kernel=2, anchor_dense=1, small_path=0, chunk_bytes=1024),
                     dict(kernel=2, anchor_dense=1, small_path=0, dense_hot=0, slab_events=2)):
            res = resource_for(ds, **opts)
            for find_all in (True, False):
                pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
                for path in (host_match, device_match):
                    p, rec = path(res, ds, find_all)
                    assert np.array_equal(p, pe) and np.array_equal(rec, re_), (len(pset), opts, find_all, path.__name__)
            assert int(res.stat("anchor_variant")) & DENSE
            api.ahocorasick_deinit(res)
        o.close()
