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
@pytest.mark.parametrize("lds,compact", [(None, -1), (2048, 0), (2048, 1), (None, 1)])
def test_tables_match_oracle(name, lds, compact):
    ds = synth.NAMED[name]()
    opts = {"lds_bytes": lds} if lds else {}
    if compact == 1:
        opts["dense_bytes"] = 0   # dense rows for the LDS tile and trie depth <= 1 only
    res = resource_for(ds, compact=compact, **opts)
    api.ahocorasick_finalize(res)
    ft = _compare(ds, res, max_hay=3 if name in ("G2", "R3") else 6)
    assert ft.out_off is not None
    assert ft.compact == (1 if compact == 1 else 0)
    if compact == 1:
        assert ft.n_dense_rows == ft.dense_nf_end + (ft.dense_f_end - ft.f2) and ft.trap < ft.dense_nf_end <= ft.f2
        code = ft.cold_rec[:, 1] >> 24
        cold = np.ones(ft.n_states, bool)
        cold[:ft.dense_nf_end] = False
        cold[ft.f2:ft.dense_f_end] = False
        assert np.all(code[~cold] == 0)
        if name in ("G2", "R3") and lds:
            # long patterns outside a 2 KiB tile: the insides of patterns are numbered along the pattern, so the
            # child of most single-child states is the next id (one 128-byte line per 16 steps of a walk)
            single = np.nonzero(cold & (code == 1))[0]
            assert single.size > 0
            nxt = ft.cold_rec[single, 0] >> 8
            assert np.mean(nxt == single + 1) > 0.5
        if ft.chain_byte is not None:
            # the bulk-compare arrays restate the records: edge byte, and the run of "child == id + 1" states
            ids = np.arange(ft.chain_lo, ft.f2)
            assert np.all(code[ids] == 1) and np.all((ft.cold_rec[ids, 0] & 0xFF) == ft.chain_byte[:ids.size])
            consecutive = ((ft.cold_rec[ids, 0] >> 8) == ids + 1) & (ids + 1 < ft.f2)
            run = np.zeros(ids.size + 1, np.int64)
            for k in range(ids.size - 1, -1, -1):
                run[k] = min(255, 1 + run[k + 1]) if consecutive[k] else 0
            assert np.array_equal(run[:-1], ft.chain_run[:ids.size])
            assert np.all(ft.chain_byte[ids.size:] == 0) and np.all(ft.chain_run[ids.size:] == 0)
    if compact != 1 and ft.chain_byte is not None:
        # dense rows (round 4): the single-child states below the tile are numbered along their patterns too -- the arrays of the
        # bulk compare restate the TABLE: the edge byte's column of a chain state leads to its trie child, which is the next id
        # for as long as the run says
        ids = np.arange(ft.chain_lo, ft.f2)
        assert ids.size > 0 and ft.chain_lo > ft.trap
        child = ft.table[ids, ft.cls[ft.chain_byte[:ids.size]]]
        consecutive = (child == ids + 1) & (ids + 1 < ft.f2)
        run = np.zeros(ids.size + 1, np.int64)
        for k in range(ids.size - 1, -1, -1):
            run[k] = min(255, 1 + run[k + 1]) if consecutive[k] else 0
        assert np.array_equal(run[:-1], ft.chain_run[:ids.size])
        if name in ("G2", "R3") and lds:
            assert np.mean(consecutive) > 0.5
    # numbering invariants the kernels rely on
    assert 0 < ft.f1 <= ft.trap < ft.n_states and ft.trap < ft.f2 <= ft.n_states
    assert np.all(ft.lds <= ft.trap)                      # tile never points outside itself
    assert np.all(ft.table[ft.trap] == ft.trap)           # trap row is absorbing (its row index is its id: trap < dense_nf_end)
    assert np.all(ft.lds[:ft.trap][ft.table[:ft.trap] <= ft.trap] == ft.table[:ft.trap][ft.table[:ft.trap] <= ft.trap])
    hot_final = np.arange(ft.f1, ft.trap)
    assert np.all(ft.cnt[hot_final] > 0) and np.all(ft.cnt[:ft.f1] == 0)
    assert np.all(ft.cnt[ft.f2:] > 0) and np.all(ft.cnt[ft.trap:ft.f2] == 0)
    if lds:
        assert (ft.trap + 1) * ft.n_columns * ft.entry_bytes <= lds
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("name,lds,fits", [("G2", 2048, True), ("G2", None, True), ("G4", 2048, True), ("G5", 2048, True), ("R1", 2048, True), ("R2", 2048, False),   # 254 byte classes: a record's key byte holds 127
                                          
                                           ("R3", 2048, False),    # 38,357 states: more than an index has bits for
                                           ("K1", 8192, True), ("K2", 8192, False), ("X1", 2048, False)])   # rows of 512 B / 26 B: too big
def test_chain_image_walks_like_the_oracle(name, lds, fits):
    """The chain-compressed image of the whole automaton (automaton.hpp ChainImage: dense rows for the branching states, goto edge +
    failure pointer for the insides of the patterns -- node.h:37-67's record) walked in numpy the way k_walk_chain does."""
    ds = (synth.NAMED.get(name) or synth.TARGETED[name])()
    opts = {"lds_bytes": lds} if lds else {}
    res = resource_for(ds, split=0, **opts)
    api.ahocorasick_finalize(res)
    ft = FlatTables(res)
    assert (ft.chain is not None) == fits, (name, ft.n_states)
    if not fits:
        api.ahocorasick_deinit(res)
        return
    ch = ft.chain
    assert ch["n_index"] == ft.n_states - 1 and 0 < ch["n_dense"] <= ch["n_index"] and ch["bytes"] <= 160 * 1024 - 768
    if (ft.n_states - 1) * (2 * ft.n_columns + 1) + 4 <= 160 * 1024 - 768:
        assert ch["n_dense"] == ch["n_index"] and ch["recs"].size == 0          # a small automaton: all of its rows, no records
    assert sorted(ch["orig"].tolist()) == sorted(set(range(ft.n_states)) - {ft.trap})      # every state once
    nd = ch["n_dense"]
    key = (ch["recs"] >> 16) & 0xFF
    assert np.all((ch["recs"] & 0xFFFF) < nd) and np.all(((key < 2 * ft.n_columns) & (key % 2 == 0)) | (key == 0xFF))
    assert np.all(key[-1:] == 0xFF)                                                        # (the last chain ends: vacuous without records)
    o = Oracle("port")
    o.add_patterns(ds.patterns())
    plen = np.diff(ds.pat_off.astype(np.int64))
    for h in range(min(ds.n_haystacks, 4)):
        for find_all in (True, False):
            exp = [tuple(int(x) for x in r) for r in o.match(ds.haystack(h), find_all, h)]
            assert ft.walk_chain(ds.haystack(h), plen, h, find_all) == exp, (name, h, find_all)
    o.close()
    res2 = resource_for(ds, split=0, chain_bytes=0, **opts)
    api.ahocorasick_finalize(res2)
    assert FlatTables(res2).chain is None
    api.ahocorasick_deinit(res)
    api.ahocorasick_deinit(res2)


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
        for a in range(4):
            o = ft.anchor_offset(k, a)   # the anchor window of an occurrence whose first aligned dword lies a bytes in
            assert o % 4 == a and o + q <= len(p) and (o == a or ft.anchor_choice == 1)
            assert ft.anchor_probe(p[o:o + q]), (name, k, a, o)
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


@pytest.mark.parametrize("name", ["K1", "K2", "K3"])
def test_dense_anchor_form_has_no_false_negatives(name):
    """Shortest pattern 4..6 B: the filter's dense form -- one key per pattern, its first four bytes, hashed raw -- must probe
    positive for every accepted pattern; random windows almost never do."""
    ds = synth.TARGETED[name]()
    res = resource_for(ds, keep_host_tables=1)
    api.ahocorasick_finalize(res)
    ft = FlatTables(res)
    assert ft.anchor_dense == 1 and ft.anchor_q == 4 and ft.anchor_mode == 1 and 4 <= ft.min_pattern_length <= 6
    assert res.stat("anchor_dense_rate") * 8192 <= 160
    for k in range(ds.n_patterns):
        assert ft.anchor_probe(ds.pattern(k)[:4]), (name, k)
    rng = np.random.default_rng(6)
    fp = sum(ft.anchor_probe(rng.integers(0, 256, 4, dtype=np.uint8).tobytes()) for _ in range(4000))
    assert fp <= 4
    api.ahocorasick_deinit(res)


def test_dense_anchor_form_only_when_selective():
    """2,048 patterns over six letters plus one of five bytes: every 4-letter window is some pattern's prefix -- the estimate
    says so and the dense form is not chosen (the table walk is); it can be forced."""
    base = synth.g1(1)
    pats = base.patterns() + [b"abcde"]
    res = api.ahocorasick_init([{"value": p} for p in pats])
    res.set_option("split", 0)   # (as ONE set; by default it is split in two: the long patterns keep the aligned filter)
    api.ahocorasick_finalize(res)
    assert res.stat("anchor_q") == 0 and res.stat("anchor_dense") == 0 and res.stat("anchor_dense_rate") * 8192 > 160
    api.ahocorasick_deinit(res)
    res = api.ahocorasick_init([{"value": p} for p in pats])
    api.ahocorasick_finalize(res)
    assert res.stat("split") == 1 and res.stat("split_long_patterns") == 2048 and res.stat("split_short_patterns") == 1 and res.stat("anchor_q") == 12
    api.ahocorasick_deinit(res)
    res = api.ahocorasick_init([{"value": p} for p in pats])
    res.set_option("anchor_dense", 1)
    api.ahocorasick_finalize(res)
    assert res.stat("anchor_q") == 4 and res.stat("anchor_dense") == 1
    api.ahocorasick_deinit(res)
    res = api.ahocorasick_init([{"value": p} for p in synth.k1().patterns()])
    res.set_option("anchor_dense", 0)
    api.ahocorasick_finalize(res)
    assert res.stat("anchor_q") == 0
    api.ahocorasick_deinit(res)


def test_window_hash_places_families_that_share_three_bytes():
    """Word lists are full of families that share their first three letters (conc, cond, conf, ...): the 4-byte window hash must
    spread them over buckets (a product alone moves a byte's influence upward only: with the fourth byte reaching the
    fingerprint bits only, such a family met in two buckets of four slots and the filter was refused)."""
    import itertools
    family = [("con" + c + "xyz").encode() for c in "cdefgjnqstvabhik"] + [b"inter", b"internal", b"integer", b"house"]
    aligned = [("abc" + c + "mno").encode() for c in "abcdefghijklmnopqrstuvwxyz"]
    suffix = [(c + "abcmnop").encode() for c in "abcdefghijklmnopqrstuvwxyz"]
    every = [("".join(t) + "zz").encode() for t in itertools.product("abcdefgh", repeat=4)]
    for pats, dense in ((family, 1), (aligned, 0), (suffix, 0), (every, 1)):
        res = api.ahocorasick_init([{"value": w} for w in pats])
        res.set_option("keep_host_tables", 1)
        if dense:
            res.set_option("anchor_dense", 1)
        api.ahocorasick_finalize(res)
        ft = FlatTables(res)
        assert ft.anchor_q == 4 and ft.anchor_mode == 1 and ft.anchor_dense == dense
        for p in pats:
            for o in ((0,) if dense else range(4)):
                assert ft.anchor_probe(p[o:o + 4]), (p, o)
        api.ahocorasick_deinit(res)


@pytest.mark.parametrize("name", ["G2", "R3", "R7", "W1"])
@pytest.mark.parametrize("compact", [0, 1])
def test_window_table_walks_like_the_oracle(name, compact):
    """The two-byte window table (automaton.hpp win_table; k_win.hip k_walk_win) walked in numpy as the kernel walks it, against the
    oracle: every cell form, the depth-3 hash, the hand-overs between the table and the records (helpers.FlatTables.walk_win)."""
    if name == "W1":   # families of 1 / 2 / 3 / 9 / 70 third bytes behind one digram, nested patterns
        r = np.random.default_rng(5)
        pats = []
        for a, b, kids in ((1, 2, 1), (3, 4, 2), (5, 6, 3), (7, 8, 9), (9, 10, 70)):
            for c in r.choice(256, size=kids, replace=False):
                pats.append(bytes([a, b, int(c)]) + bytes(r.integers(0, 256, size=int(r.integers(0, 9))).tolist()))
        pats += [bytes([7, 8, 1, 2, 5, 6, 9]), bytes([1, 2, 5, 6]), bytes([2, 5, 6]), bytes([5, 6, 7, 8]), bytes([1, 2, 5])]
        pats = list(dict.fromkeys(pats))
        hays = []
        for _ in range(6):
            t = r.integers(0, 12, size=600).astype(np.uint8)
            for _k in range(40):
                p = pats[int(r.integers(len(pats)))]
                at = int(r.integers(0, 600 - len(p)))
                t[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
            hays.append(t.tobytes())
        pb, po = synth.pack(pats)
        hb, ho = synth.pack(hays)
        ds = synth.Dataset("W1", pb, po, hb, ho)
    else:
        ds = synth.NAMED[name]()
    opts = dict(dense_bytes=0, lds_bytes=4096) if compact else {}
    res = resource_for(ds, win_table=1, compact=compact, **opts)
    api.ahocorasick_finalize(res)
    ft = FlatTables(res)
    assert ft.win is not None and res.stat("win_table") == 1
    o = Oracle("port")
    o.add_patterns(ds.patterns())
    o.finalize()
    n_rec = 0
    for h in range(min(ds.n_haystacks, 3 if name in ("G1", "G2", "R3") else 6)):
        for find_all in (True, False):
            exp = [tuple(int(x) for x in r) for r in o.match(ds.haystack(h), find_all, h)]
            assert ft.walk_win(ds.haystack(h), None, h, find_all) == exp, (ds.name, h, find_all)
            n_rec += len(exp)
    assert n_rec > 0
    o.close()
    api.ahocorasick_deinit(res)
