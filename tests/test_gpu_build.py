"""The level-synchronous part of the automaton build on the device (failure links + per-state outputs, one launch per
trie depth: k_misc.hip k_build_level) must leave byte-identical tables (reference: ac_trie_set_failure,
ahocorasick.c:344-368; node_collect_matches, node.c:424-441)."""
import time

import numpy as np
import pytest

from helpers import FlatTables
from gpu_util import host_match, resource_for
from php_aho_corasick_amd import api, synth

pytestmark = pytest.mark.gpu

FIELDS = ("table", "lds", "cls", "own", "dict", "cnt", "cold_rec", "cold_edges", "chain_byte", "chain_run", "out_off", "out_list",
          "anchor", "anchor_bitmap")
SCALARS = ("n_states", "n_columns", "entry_bytes", "f1", "trap", "f2", "compact", "dense_nf_end", "dense_f_end", "chain_lo",
           "anchor_q", "anchor_mul", "anchor_mode", "anchor_keys", "anchor_bucket_bits", "anchor_bucket_shift", "anchor_choice")


@pytest.mark.parametrize("name", ["G2", "G4", "R1", "R3", "R8", "G6"])
def test_device_build_equals_host_build(name, golden):
    ds = synth.NAMED[name]()
    tabs, secs = [], []
    for dev in (1, 0):
        res = resource_for(ds, device_build=dev, **({"compact": 1, "dense_bytes": 0, "lds_bytes": 4096} if name in ("R1", "R3") else {}))
        t0 = time.perf_counter()
        api.ahocorasick_finalize(res)
        secs.append(time.perf_counter() - t0)
        assert res.stat("built_on_device") == dev
        tabs.append((FlatTables(res), res))
    a, b = tabs[0][0], tabs[1][0]
    for f in SCALARS:
        assert getattr(a, f) == getattr(b, f), f
    for f in FIELDS:
        x, y = getattr(a, f), getattr(b, f)
        assert (x is None) == (y is None), f
        if x is not None:
            assert np.array_equal(x, y), f
    # and the device-built automaton matches like the host-built one
    g = golden[(name, True)]
    prefix, rec = host_match(tabs[0][1], ds, True)
    assert len(rec) == g["n_records"]
    print(f"{name}: finalize {secs[0] * 1e3:.0f} ms with the device, {secs[1] * 1e3:.0f} ms on the host, {a.n_states} states")
    for _, res in tabs:
        api.ahocorasick_deinit(res)
