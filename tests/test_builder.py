"""Host automaton builder (csrc/automaton.cpp) checked on the CPU: its flat tables are walked in
numpy exactly as the device's exact path walks them and compared with the oracle."""
import numpy as np
import pytest

from helpers import FlatTables
from gpu_util import resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import api, synth


def _compare(ds, res, max_hay=6, calls=1):
    ft = FlatTables(res)
    o = Oracle("port")
    pats = ds.patterns()
    step = (len(pats) + calls - 1) // calls if pats else 1
    for lo in range(0, max(len(pats), 1), step):
        o.add_patterns(pats[lo:lo + step])
    o.finalize()
    assert ft.n_states == o.node_count() + 1
    assert res.stat("accepted") == o.pattern_count()
    plen = np.diff(ds.pat_off.astype(np.int64))
    for h in range(min(ds.n_haystacks, max_hay)):
        for find_all in (True, False):
            exp = [tuple(int(x) for x in r) for r in o.match(ds.haystack(h), find_all, h)]
            assert ft.walk(ds.haystack(h), plen, h, find_all) == exp, (ds.name, h, find_all)
    return ft


@pytest.mark.parametrize("name", ["G2", "G3", "G4", "G5", "R1", "R2", "R3"])
@pytest.mark.parametrize("lds,chains", [(None, -1), (2048, 0), (2048, 1)])
def test_tables_match_oracle(name, lds, chains):
    ds = synth.NAMED[name]()
    opts = {"lds_bytes": lds} if lds else {}
    res = resource_for(ds, cold_chains=chains, **opts)
    api.ahocorasick_finalize(res)
    ft = _compare(ds, res, max_hay=3 if name in ("G2", "R3") else 6)
    assert ft.out_off is not None
    if chains == 1 and name in ("G2", "R2", "R3"):
        # long patterns outside a 2 KiB tile: chain states exist, are numbered along their pattern ...
        assert ft.chain is not None and ft.trap < ft.chain_lo < ft.f2
        nxt = ft.chain >> 8
        inside = (nxt >= ft.chain_lo) & (nxt < ft.f2)
        assert np.all(nxt[inside] == np.arange(ft.chain_lo, ft.f2)[inside] + 1)  # ... child of a chain state = next id
    if chains == 0:
        assert ft.chain is None and ft.chain_lo == ft.f2
    # numbering invariants the kernels rely on
    assert 0 < ft.f1 <= ft.trap < ft.n_states and ft.trap < ft.f2 <= ft.n_states
    assert np.all(ft.lds <= ft.trap)                      # tile never points outside itself
    assert np.all(ft.table[ft.trap] == ft.trap)           # trap row is absorbing
    assert np.all(ft.lds[:ft.trap][ft.table[:ft.trap] <= ft.trap] == ft.table[:ft.trap][ft.table[:ft.trap] <= ft.trap])
    hot_final = np.arange(ft.f1, ft.trap)
    assert np.all(ft.cnt[hot_final] > 0) and np.all(ft.cnt[:ft.f1] == 0)
    assert np.all(ft.cnt[ft.f2:] > 0) and np.all(ft.cnt[ft.trap:ft.f2] == 0)
    if lds:
        assert (ft.trap + 1) * ft.n_columns * ft.entry_bytes <= lds
    api.ahocorasick_deinit(res)


def test_incremental_calls_and_duplicates():
    ds = synth.r1()
    res = resource_for(ds, calls=5)
    api.ahocorasick_finalize(res)
    _compare(ds, res, calls=5)
    api.ahocorasick_deinit(res)


def test_class_modes():
    res = resource_for(synth.g1(8))
    api.ahocorasick_finalize(res)
    ft = FlatTables(res)
    assert ft.class_mode == 1 and ft.swar_shift == 0 and ft.n_columns == 7    # abcdef -> buckets 1..6, other = 0
    assert [int(ft.cls[b]) for b in b"abcdef"] == [1, 2, 3, 4, 5, 6] and ft.cls[ord("A")] == 0 and ft.cls[0x69] == 0
    assert ft.n_states == 25581 and res.stat("nodes") == 25580                # 25,580 nodes
    api.ahocorasick_deinit(res)
    res = resource_for(synth.r2())
    api.ahocorasick_finalize(res)
    ft = FlatTables(res)
    assert ft.class_mode == 0 and ft.n_columns == res.stat("used_bytes") + 1
    api.ahocorasick_deinit(res)
    pats = [{"value": bytes([b])} for b in range(256)]                        # all 256 bytes used: identity columns
    res = api.ahocorasick_init(pats)
    api.ahocorasick_finalize(res)
    ft = FlatTables(res)
    assert ft.n_columns == 256 and list(ft.cls) == list(range(256))
    api.ahocorasick_deinit(res)
    res = api.ahocorasick_init([{"value": "ACGT"}, {"value": "TTGACA"}])       # ACGT: 3-bit window exists
    api.ahocorasick_finalize(res)
    assert FlatTables(res).class_mode == 1
    api.ahocorasick_deinit(res)


def test_wide_entries_for_big_automata():
    ds = synth.generate("W", 21, 12000, 8, 8, 1, 64, None)      # > 65535 states -> uint32 entries
    res = resource_for(ds)
    api.ahocorasick_finalize(res)
    assert res.stat("states") > 65536 and res.stat("entry_bytes") == 4
    _compare(ds, res, max_hay=1)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("name,q,mode", [("G2", 12, 1), ("R3", 16, 1), ("R4", 8, 1), ("R5", 12, 1), ("R6", 16, 1),
                                         ("R7", 4, 1), ("R8", 4, 2), ("G6", 4, 2)])
def test_anchor_table_has_no_false_negatives(name, q, mode):
    """Every pattern's q-grams at offsets 0..3 must probe positive (a miss would lose matches); random
    windows almost never do."""
    ds = synth.NAMED[name]()
    res = resource_for(ds)
    api.ahocorasick_finalize(res)
    ft = FlatTables(res)
    assert ft.anchor_q == q and ft.anchor_mode == mode and ft.min_pattern_length >= q + 3
    for k in range(0, ds.n_patterns, 1 if ds.n_patterns <= 5000 else 37):
        p = ds.pattern(k)
        for o in range(4):
            assert ft.anchor_probe(p[o:o + q]), (name, k, o)
    rng = np.random.default_rng(5)
    fp = sum(ft.anchor_probe(rng.integers(0, 256, q, dtype=np.uint8).tobytes()) for _ in range(4000))
    assert fp <= 8   # q = 4 over G6's 262,144 q-grams: 6e-5 per random window
    api.ahocorasick_deinit(res)


def test_anchor_not_offered_for_short_patterns():
    for name in ("G4", "R1", "R2"):
        res = resource_for(synth.NAMED[name]())
        api.ahocorasick_finalize(res)
        assert FlatTables(res).anchor_q == 0
        api.ahocorasick_deinit(res)
