"""Search and replace on the device (acmi_replace_batch) against the reference's goldens and the oracle:
multifast_replace + multifast_rep_flush(keep=0), src/multifast/replace.c:469-568, NORMAL and LAZY modes."""
import hashlib

import numpy as np
import pytest

from helpers import load_json
from gpu_util import resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import _native as N, api, synth

pytestmark = pytest.mark.gpu


def _resource(ds, **options):
    res = resource_for(ds, **options)
    for k, p in enumerate(ds.patterns()):
        api.set_replacement(res, k, synth.replacement(k, p))
    return res


@pytest.mark.parametrize("name", ["G2", "G3", "G4", "G5", "R1", "R2", "R3", "R4", "R5", "R6", "R7", "R8"])
def test_replace_goldens(name):
    cases = {(c["dataset"], c["mode"]): c for c in load_json("replace_golden.json")["cases"]}
    ds = synth.NAMED[name]()
    res = _resource(ds)
    for mode in (1, 2):
        out, off = api.replace_arrays(res, ds.hay_blob, ds.hay_off, mode)
        g = cases[(name, mode)]
        assert int(off[-1]) == g["out_bytes"]
        assert hashlib.sha256(off.astype(np.uint64).tobytes() + out.tobytes()).hexdigest() == g["sha256"], (name, mode)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("options", [dict(kernel=1, chunk_bytes=64), dict(kernel=0), dict(kernel=4, slab_events=1), dict(kernel=2, chunk_bytes=256)])
def test_replace_through_every_scan_kernel(options):
    ds = synth.r1() if options.get("kernel") != 2 else synth.r4()
    res = _resource(ds, **options)
    o = Oracle("port")
    for k, p in enumerate(ds.patterns()):
        o.set_replacement(k, synth.replacement(k, p))
    o.add_patterns((ds.pat_blob, ds.pat_off))
    for mode in (0, 1, 2):
        out, off = api.replace_arrays(res, ds.hay_blob, ds.hay_off, mode)
        for h in range(ds.n_haystacks):
            assert out[int(off[h]):int(off[h + 1])].tobytes() == o.replace(ds.haystack(h), mode), (h, mode)
    api.ahocorasick_deinit(res)


def test_replace_quirks_and_errors():
    # a replacement text with a NUL byte is kept up to the NUL and zero-filled (mpool_strndup, mpool.c:165-178);
    # a duplicate pattern's replacement never counts (the first inserted pattern owns the node, ahocorasick.c:123-124);
    # a pattern without replacement inside one with: the longer one is the nominee (node.c:337-362)
    pats = [b"abc", b"bc", b"abc", b"zz"]
    res = api.ahocorasick_init([{"value": p} for p in pats])
    api.set_replacement(res, 0, b"FIRST")      # loses: the LAST duplicate in caller order is inserted first
    api.set_replacement(res, 2, b"X\x00Y")
    api.set_replacement(res, 3, b"")
    o = Oracle("port")
    for k, r in ((0, b"FIRST"), (2, b"X\x00Y"), (3, b"")):
        o.set_replacement(k, r)
    o.add_patterns(pats)
    blob = np.frombuffer(b"..abc..bc..zz.." + b"abcabc", dtype=np.uint8)
    off = np.array([0, 15, 21], dtype=np.uint64)
    for mode in (1, 2):
        out, ooff = api.replace_arrays(res, blob, off, mode)
        for h in range(2):
            assert out[int(ooff[h]):int(ooff[h + 1])].tobytes() == o.replace(blob[int(off[h]):int(off[h + 1])].tobytes(), mode)
    assert out[:int(ooff[1])].tobytes() == b"..X\x00\x00..bc...."
    api.ahocorasick_deinit(res)
    res = api.ahocorasick_init([{"value": "abc"}])
    with pytest.raises(N.AcmiError) as e:
        api.replace_arrays(res, blob, off, 1)
    assert e.value.status == N.E_NO_REPLACEMENT
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("name", ["G2", "R1", "R2", "R5"])
def test_replace_device_entry_goldens(name):
    """acmi_replace_batch_device: haystacks resident in HBM, output left there (device prefix sum of the output lengths,
    copy cut by output bytes) -- same bytes and offsets as the reference's replace.c (goldens)."""
    import torch
    from php_aho_corasick_amd import batch
    cases = {(c["dataset"], c["mode"]): c for c in load_json("replace_golden.json")["cases"]}
    ds = synth.NAMED[name]()
    res = _resource(ds)
    d_blob = torch.from_numpy(ds.hay_blob if ds.hay_blob.size else np.zeros(16, np.uint8)).cuda()
    d_off = torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
    batch.bind_current_stream(res)
    for mode in (1, 2, 1):   # (the third call reuses buffers sized by the first two)
        out, off = batch.replace_device(res, d_blob, d_off, mode)
        out, off = out.cpu().numpy(), off.cpu().numpy().astype(np.uint64)
        g = cases[(name, mode)]
        assert int(off[-1]) == g["out_bytes"] == len(out)
        assert hashlib.sha256(off.tobytes() + out.tobytes()).hexdigest() == g["sha256"], (name, mode)
    api.ahocorasick_deinit(res)


@pytest.mark.parametrize("n_plants", [0, 3, 5000])
def test_replace_one_big_haystack(n_plants):
    """The README's one-big-file case: ONE haystack of 6 MiB (its output is cut into 32-KiB ranges over the whole chip),
    with no, a few and thousands of replacements, output longer and shorter than the input; empty haystacks around it."""
    import torch
    from php_aho_corasick_amd import batch
    rng = np.random.default_rng(5)
    pats = [b"needle-%03d" % k for k in range(40)] + [b"ndl"]
    big = rng.integers(97, 123, size=6 << 20, dtype=np.uint8)
    for k in range(n_plants):
        p = np.frombuffer(pats[k % len(pats)], dtype=np.uint8)
        at = int(rng.integers(0, len(big) - 16))
        big[at:at + len(p)] = p
    small = np.frombuffer(b"xx needle-007 yy ndl", dtype=np.uint8)
    blob = np.concatenate([small, big, small])
    off = np.array([0, 0, len(small), len(small), len(small) + len(big), len(blob), len(blob)], dtype=np.uint64)
    res = api.ahocorasick_init([{"value": p} for p in pats])
    o = Oracle("port")
    for k, p in enumerate(pats):
        r = None if k % 9 == 4 else (b"" if k % 5 == 0 else (p * 3 if k % 2 else b"#"))
        api.set_replacement(res, k, r)
        o.set_replacement(k, r)
    o.add_patterns(pats)
    batch.bind_current_stream(res)
    d_blob, d_off = torch.from_numpy(blob).cuda(), torch.from_numpy(off.astype(np.int64)).cuda()
    for mode in (1, 2):
        out, ooff = batch.replace_device(res, d_blob, d_off, mode)
        out, ooff = out.cpu().numpy(), ooff.cpu().numpy()
        for h in range(len(off) - 1):
            want = o.replace(blob[int(off[h]):int(off[h + 1])].tobytes(), mode)
            assert out[int(ooff[h]):int(ooff[h + 1])].tobytes() == want, (h, mode, n_plants)
    api.ahocorasick_deinit(res)
