#!/usr/bin/env python3
"""Throughput bench of the ac_trie_search() replacement on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--haystacks NH]

One "step" = one pass of the hot path over one batch that is already resident in HBM:
acmi_match_batch_device() = work-item table, LDS-tile scan kernel, record-base scan, expansion of
events into the final {hay,pos,start,pattern} array (left on the device).  K steps = K batches begun
and ended inside the timed region, up to two of them in flight (acmi_match_batch_device_begin/_end:
the second batch's kernels queue behind the first one's, the host's turn-around is hidden;
`roofline.one_at_a_time_ms_per_step` is the same step without that).  With N > 1 (launched by
torch.distributed.run, one rank per GPU, backend "nccl" == RCCL) every rank scans its own shard of
the same size (weak scaling) and the per-rank match arrays are all-gathered; the collective of step
i runs on RCCL's stream while step i+1 scans.

Workload (config.workload): BASELINE.json configs[1] shape -- 2048 patterns x 16 B over 8 KiB
haystacks, alphabet abcdef, one planted hit per haystack -- scaled to 2^19 haystacks (4 GiB) per
GPU, because the literal 256-haystack / 2 MiB batch is 0.3 us of HBM time, below launch latency.

The LAST stdout line (rank 0) is the result: one JSON object under 4 KB with the contract keys, `roofline`, `cpu_baseline`
and one {frac, ok} pair per secondary leg (`legs`).  The legs in full go to an earlier line prefixed "BENCH_DETAIL " and to
gpurun_out/bench_legs.json.  `roofline.achieved` = algorithmic bytes of one scan launch
(haystack bytes + 16 B per record + 8 B per haystack) / mean duration of the scan
kernel measured with HIP events on the stream it runs on.  `cpu_baseline` = the reference's own
MultiFast C (oracle/_ref, kind "reference") or our restatement (kind "port") timed on this box's
host cores over a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
TILE_HAYSTACKS = 8192  # 64 MiB generated on the host, replicated on the device
HAY_LEN = 8192


def _cgroup_cpu_max():
    """The CPU time this process may use per period, if its cgroup says so ("max 100000": no limit): an all-cores figure
    far under cores x the one-core rate is this quota, or the memory system, not the trie."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            return open(path).read().strip()
        except Exception:
            pass
    return None


def cpu_baseline_start(ds, seconds_target: float = 10.0):
    """The one-core half of cpu_baseline() on a thread of its own (the C loop releases the GIL): it runs next to the untimed
    secondary legs -- one busy core of the box's sixteen -- never next to the timed steps.  cpu_baseline_finish() joins it and
    measures all cores."""
    box = {}

    def work():
        try:
            box["out"] = _cpu_one_core(ds, seconds_target)
        except Exception as e:   # reported by cpu_baseline_finish
            box["error"] = e
    th = threading.Thread(target=work, daemon=True)
    th.start()
    return th, box, seconds_target


def cpu_baseline_finish(ds, started):
    th, box, seconds_target = started
    th.join()
    if "error" in box:
        raise box["error"]
    return _cpu_all_cores(ds, seconds_target, *box["out"])


def cpu_baseline(ds, seconds_target: float = 10.0):
    """Reference CPU path over a bounded sample: 1 thread (the reference as shipped) and all cores
    (one trie per thread -- AC_TRIE_t is not re-entrant, ahocorasick.h:49-65)."""
    return _cpu_all_cores(ds, seconds_target, *_cpu_one_core(ds, seconds_target))


def _cpu_make(ds, kind):
    from oracle import pyoracle
    o = pyoracle.Oracle(kind)
    t0 = time.perf_counter()
    o.add_patterns((ds.pat_blob, ds.pat_off))
    o.finalize()
    return o, time.perf_counter() - t0


def _cpu_one_core(ds, seconds_target):
    from oracle import pyoracle
    kind = "reference" if pyoracle.have_reference() else "port"
    o, build_s = _cpu_make(ds, kind)
    nh = 2048  # 16 MiB probe to size the sample
    sub_off = ds.hay_off[:nh + 1]
    sub_blob = ds.hay_blob[:int(sub_off[-1])]
    t0 = time.perf_counter()
    o.scan_checksum(sub_blob, sub_off)
    probe = time.perf_counter() - t0
    reps = max(1, int(seconds_target / max(probe, 1e-3)))
    reps = min(reps, 64)
    t0 = time.perf_counter()
    n_rec = 0
    for _ in range(reps):
        n_rec, _ck = o.scan_checksum(sub_blob, sub_off)
    dt = time.perf_counter() - t0
    one = {"value": round(reps * len(sub_blob) / dt / 1e9, 5), "unit": "GB/s", "cores": 1, "kind": kind,
           "sample": f"{reps} x {nh} haystacks x {HAY_LEN} B ({reps * len(sub_blob) >> 20} MiB) of the bench "
                     f"workload, callback = count + checksum, {n_rec} records per pass",
           "build_ms": round(build_s * 1e3, 2)}
    return one, kind, nh, sub_blob, sub_off, probe


def _cpu_all_cores(ds, seconds_target, one, kind, nh, sub_blob, sub_off, probe):
    def make():
        return _cpu_make(ds, kind)

    # All cores: one trie replica and ONE call per thread, the repetitions looped in C (*_php_scan_checksum_reps) over the whole
    # sample -- >= 0.5 s of trie work per thread.  (Round 3 gave each thread 8 haystacks x 14 Python-level calls: 22 ms of work
    # behind 3,584 GIL hand-overs, 13.7 x one core on 256 cores -- that measured the interpreter.)
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = visible
    quota = _cgroup_cpu_max()
    try:   # "1600000 100000": this container may use 16 CPUs' worth of time however many it sees -- more threads only take turns
        q, period = quota.split()
        if q != "max":
            cores = max(1, min(visible, int(round(int(q) / int(period)))))
    except Exception:
        pass
    tries = [make()[0] for _ in range(cores)]
    per_pass = probe   # seconds one thread needs for one pass over the sample
    reps_mt = max(1, int(round(max(0.8, seconds_target / 8) / max(per_pass, 1e-3))))

    def work(i):
        tries[i].scan_checksum_reps(sub_blob, sub_off, reps_mt)   # ctypes releases the GIL for the whole call

    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    dt = time.perf_counter() - t0
    allc = {"value": round(cores * reps_mt * len(sub_blob) / dt / 1e9, 5), "unit": "GB/s", "cores": cores,
            "kind": kind, "speedup_over_one_core": round(cores * reps_mt * len(sub_blob) / dt / 1e9 / one["value"], 1),
            "cores_visible": visible, "cgroup_cpu_max": quota,
            "sample": f"{cores} threads x {reps_mt} x {nh} haystacks x {HAY_LEN} B ({reps_mt * len(sub_blob) >> 20} MiB each), one trie replica and one C call per thread"}
    return one, allc


def dense_leg(torch, api, batch, synth, dev, nh=4096, k=16):
    """Secondary figure, outside the timed region: BASELINE configs[4], the adversarial high-hit-rate set
    (patterns a^1..a^16 over haystacks of 'a': 16 records = 256 B written per byte read), whole calls of
    acmi_match_batch_device on a resident batch.  Write-bound: the rate is set against the same 8 TB/s."""
    ds = synth.g3(n_hay=1, hay_len=HAY_LEN, k=k)
    res = api.ahocorasick_init([{"id": i, "value": ds.pattern(i)} for i in range(ds.n_patterns)])
    res.set_option("profile", 2)
    d_blob = torch.from_numpy(ds.hay_blob).to(dev).repeat(nh).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device=dev) * HAY_LEN
    batch.bind_current_stream(res)
    ms, total = [], 0
    for i in range(13):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total = batch.match_device(res, d_blob, d_off, True)  # returns after the call's one host synchronisation
        if i >= 6:  # the first call grows the record buffer and runs the expansion twice; automatic mode may probe a kernel early on
            ms.append(time.perf_counter() - t0)
    t = float(np.median(ms))
    expect = nh * (k * HAY_LEN - k * (k - 1) // 2)
    # content at the bench's own size, not only the count: the record blocks of three haystacks (first, middle, last) against the
    # closed form -- at position p the patterns a^min(p,k) .. a^1, longest first (node.c:424-441), pattern a^l being index l - 1
    per_hay = expect // nh
    pos = np.repeat(np.arange(1, HAY_LEN + 1, dtype=np.int64), np.minimum(np.arange(1, HAY_LEN + 1), k))
    first = np.concatenate(([0], np.cumsum(np.minimum(np.arange(1, HAY_LEN + 1), k))))[:-1]
    ln = np.minimum(pos, k) - (np.arange(len(pos)) - np.repeat(first, np.minimum(np.arange(1, HAY_LEN + 1), k)))
    want = np.stack([np.zeros_like(pos), pos, pos - ln, ln - 1], axis=1).astype(np.int32)
    content_ok = total == expect and len(want) == per_hay
    if content_ok:
        view = batch.result_view(res, total)
        for hh in (0, nh // 2, nh - 1):
            got = view[hh * per_hay:(hh + 1) * per_hay].cpu().numpy()
            want[:, 0] = hh
            content_ok = content_ok and bool(np.array_equal(got, want))
    import hashlib
    want[:, 0] = nh - 1
    out = {"workload": f"patterns a^1..a^{k} over {nh} haystacks x {HAY_LEN} B of 'a' ({k} records per byte)",
           "records": total, "count_ok": bool(total == expect), "content_ok": bool(content_ok),
           "last_haystack_sha256": hashlib.sha256(want.tobytes()).hexdigest()[:16] if content_ok else None,
           "ms_per_call": round(t * 1e3, 3),
           "matches_per_s": round(total / t, 1), "written_GB_per_s": round(total * 16 / t / 1e9, 1),
           "frac_of_hbm_peak": round(total * 16 / t / 1e9 / HBM_PEAK_GBS, 4), "overflow_items": int(res.stat("overflow_items")),
           "kernel": _kernel_name(res), "choice_probes": int(res.stat("choice_probes"))}
    api.ahocorasick_deinit(res)
    return out


def _timed_calls(torch, batch, res, d_blob, d_off, calls=6, skip=2):
    """Whole acmi_match_batch_device calls on a resident batch: (mean host ms per call, mean scan-kernel ms, mean
    walk-kernel ms, records).  Kernel times come from calls at profile level 2 (five events on the stream, each one a
    barrier between two kernels), the whole-call time from further calls at level 0; the first `skip` calls grow
    workspaces and probe kernels and are not counted."""
    ms, scan, walk, total = [], [], [], 0
    for _ in range(6):   # automatic mode is a measured choice: the probes of the alternatives happen here, not in the timed calls
        total = batch.match_device(res, d_blob, d_off, True)
    res.set_option("profile", 2)
    for i in range(calls):
        total = batch.match_device(res, d_blob, d_off, True)
        if i >= skip:
            scan.append(res.stat("scan_ms"))
            walk.append(res.stat("walk_ms"))
    res.set_option("profile", 0)
    for i in range(calls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        total = batch.match_device(res, d_blob, d_off, True)
        ms.append((time.perf_counter() - t0) * 1e3)
    _two_in_flight(torch, batch, res, d_blob, d_off)
    res.set_option("profile", 2)
    return float(np.median(ms)), float(np.median(scan)), float(np.median(walk)), total


def _two_in_flight(torch, batch, res, d_blob, d_off, n=6):
    """The same batch with two calls in flight (begin, begin, end, begin, end, ...): the rate a caller of the split
    entry points gets.  Leaves the figure where _leg_summary picks it up."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    began = ended = 0
    while ended < n:
        while began < n and began - ended < 2:
            batch.match_device_begin(res, d_blob, d_off, True)
            began += 1
        batch.match_device_end(res)
        ended += 1
    torch.cuda.synchronize()
    _timed_calls.two_in_flight_ms = (time.perf_counter() - t0) / n * 1e3


KERNEL_NAMES = {0: "k_scan_simple", 1: "k_scan_lds", 2: "k_scan_anchor", 4: "k_walk_items", 5: "k_match_small", 6: "k_walk_tiles"}


def _kernel_name(res):
    """The scan kernel of the handle's last batch (kernel 6 is k_walk_chain when the whole automaton was walked in LDS)."""
    k = int(res.stat("kernel_used"))
    return "k_walk_chain" if k == 6 and int(res.stat("chain_walk_used")) else KERNEL_NAMES.get(k, "?")


def _leg_summary(workload, blob_bytes, nh, total, ms, scan, walk, res, ok):
    algo = blob_bytes + 16 * total + 8 * nh  # 1 B per haystack byte + 16 B per record + 8 B per haystack
    return {"workload": workload, "kernel": _kernel_name(res),
            "ms_per_call": round(ms, 4), "scan_kernel_ms": round(scan, 4), "walk_kernel_ms": round(walk, 4),
            "haystack_GB_per_s": round(blob_bytes / ms / 1e6, 1), "records": int(total), "check_ok": bool(ok),
            "ms_per_call_two_in_flight": round(_timed_calls.two_in_flight_ms, 4),
            "haystack_GB_per_s_two_in_flight": round(blob_bytes / _timed_calls.two_in_flight_ms / 1e6, 1),
            "algorithmic_bytes": int(algo), "frac_scan_kernel": round(algo / (scan * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frac_scan_plus_walk": round(algo / ((scan + walk) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frac_whole_call": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frac_whole_call_two_in_flight": round(algo / (_timed_calls.two_in_flight_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}


def config3_cpu_start(synth, nh=1 << 20):
    """The config-3 leg's CPU baseline (the reference's trie of 65,536 patterns takes seconds to build, on one core) on a thread of
    its own, begun behind the timed region: it runs -- in C, the GIL released -- next to the GPU legs and is joined by config3_leg."""
    box = {}

    def work():
        try:
            ds = synth.g6(n_hay=4096, hay_len=4096)
            box["out"] = config3_cpu_baseline(ds, 4096, 4096, max(1, nh // 4096) * 4096)
        except Exception as e:   # reported with the leg, not fatal
            box["out"] = {"error": str(e)[:200]}
    th = threading.Thread(target=work, daemon=True)
    th.start()
    return th, box


def config3_leg(torch, api, batch, synth, dev, nh=1 << 20, cpu=None):
    """BASELINE configs[2]: 65,536 patterns of 8..64 B over Sigma=256 (2.27 M states, seed 7: the G6 automaton),
    2^20 haystacks x 4 KiB = 4 GiB resident, one planted pattern per haystack.  Check: every haystack's plant is
    reported at its place (pattern, start and end), and the tile's record count repeats."""
    tile_h, hl = 4096, 4096
    ds = synth.g6(n_hay=tile_h, hay_len=hl)
    pats = [{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)]
    t0 = time.perf_counter()
    res = api.ahocorasick_init(pats)
    t1 = time.perf_counter()
    api.ahocorasick_finalize(res)   # (the HIP runtime is up since the headline run: its start is not in this figure)
    add_ms, finalize_ms = (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3
    build_ms = add_ms + finalize_ms
    res.set_option("profile", 2)
    reps = max(1, nh // tile_h)
    nh = reps * tile_h
    d_blob = torch.from_numpy(ds.hay_blob).to(dev).repeat(reps).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device=dev) * hl
    batch.bind_current_stream(res)
    ms, scan, walk, total = _timed_calls(torch, batch, res, d_blob, d_off)
    prefix, rec = batch.result_numpy(res, nh, total)
    # plants of synth.g6: hay[h][(37 h + 11) mod (hl - 64) ...) = pattern (1021 h + 17) mod 65536, h within the tile
    h = np.arange(tile_h, dtype=np.int64)
    pat = (1021 * h + 17) % ds.n_patterns
    start = (37 * h + 11) % (hl - 64)
    plen = np.diff(ds.pat_off.astype(np.int64))[pat]
    want = np.stack([h, start + plen, start, pat], axis=1).astype(np.uint32)
    first_tile = rec[:int(prefix[tile_h])]
    have = set(map(tuple, first_tile.tolist()))
    ok = all(tuple(w) in have for w in want.tolist()) and total == int(prefix[tile_h]) * reps and int(prefix[-1]) == total
    out = _leg_summary(f"65,536 patterns of 8..64 B (Sigma=256, seed 7: 2,269,830 states) over {nh} haystacks x {hl} B "
                       f"({nh * hl >> 20} MiB), one planted pattern per haystack; BASELINE configs[2] shape",
                       nh * hl, nh, total, ms, scan, walk, res, ok)
    out.update({"automaton_build_ms": round(build_ms, 1), "add_patterns_ms": round(add_ms, 1), "finalize_ms": round(finalize_ms, 1),
                "built_on_device": int(res.stat("built_on_device")), "automaton_bytes": int(res.stat("automaton_bytes")),
                "compact": int(res.stat("compact")), "anchor_keys": int(res.stat("anchor_keys"))})
    api.ahocorasick_deinit(res)
    del d_blob
    if cpu:
        cpu[0].join()
        out["cpu_baseline"] = cpu[1].get("out")
    return out


def config3_cpu_baseline(ds, tile_h, hl, full_nh):
    """BASELINE.md section 3 for configs[2]: the reference's own C (or the port) on ONE core over a 4,096-haystack subsample of the
    leg's tile, the time for the whole batch extrapolated from it (the scan is linear in the text; the build is paid once)."""
    from oracle import pyoracle
    kind = "reference" if pyoracle.have_reference() else "port"
    o = pyoracle.Oracle(kind)
    t0 = time.perf_counter()
    o.add_patterns((ds.pat_blob, ds.pat_off))
    o.finalize()
    build_s = time.perf_counter() - t0
    nh = min(4096, tile_h)
    off = ds.hay_off[:nh + 1]
    blob = ds.hay_blob[:int(off[-1])]
    t0 = time.perf_counter()
    n_rec, _ck = o.scan_checksum(blob, off)
    dt = time.perf_counter() - t0
    o.close()
    rate = len(blob) / dt / 1e9
    return {"value": round(rate, 5), "unit": "GB/s", "cores": 1, "kind": kind, "build_ms": round(build_s * 1e3, 1),
            "sample": f"{nh} haystacks x {hl} B ({len(blob) >> 20} MiB) of the leg's workload, callback = count + checksum, {n_rec} records",
            "extrapolated_s_for_the_whole_batch": round(full_nh * hl / (rate * 1e9), 1),
            "extrapolation": "scan time is linear in the text: subsample rate x the batch's bytes; the build is not in it"}


def benchmark_php_leg(api, synth, samples=10):
    """The reference's only published number (README.md:182-187) is a SAMPLE of examples/benchmark.php:55-72: init (2,048 needles of
    16 chars) + 256 x ahocorasick_match (8,192 chars each) + deinit, averaged over 10 samples.  The same loop through this library's
    mirror of the PHP functions (one launch per ahocorasick_match call: k_match_small), the same with ONE ahocorasick_match_batch
    call for the 256 haystacks, and the reference's own C (oracle/_ref, no Zend layer: no result arrays are built) doing the same
    loop on this box's CPU.  Build and scan are reported separately."""
    from oracle import pyoracle
    ds = synth.g2()   # the literal BASELINE configs[1] shape with a planted occurrence per haystack (G1 has no match at all)
    pats = [{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)]
    hays = [ds.haystack(h) for h in range(ds.n_haystacks)]
    kind = "reference" if pyoracle.have_reference() else "port"
    o = pyoracle.Oracle(kind)
    n_ref, b_s, s_s, r_s = o.benchmark_sample(ds.pat_blob, ds.pat_off, ds.hay_blob, ds.hay_off, samples)
    o.close()
    out = {"workload": f"{samples} samples x (ahocorasick_init of 2,048 x 16 B + 256 x ahocorasick_match of 8,192 B + ahocorasick_deinit): examples/benchmark.php:55-72",
           "reference_published_s_per_sample": 0.174326, "reference_published_on": "unspecified hardware, PHP + Zend arrays included (README.md:183)",
           "cpu": {"kind": kind, "cores": 1, "build_ms_per_sample": round(b_s / samples * 1e3, 3), "scan_ms_per_sample": round(s_s / samples * 1e3, 3),
                   "release_ms_per_sample": round(r_s / samples * 1e3, 3), "s_per_sample": round((b_s + s_s + r_s) / samples, 5), "records_per_sample": n_ref,
                   "note": "the reference's MultiFast C without the Zend layer (callback = count + checksum)"}}
    for name, batched in (("per_call", False), ("one_match_batch", True)):
        build = scan = deinit = 0.0
        n_rec = 0
        for i in range(samples + 1):   # (the first sample pays the runtime's first use of the kernels: not counted)
            t0 = time.perf_counter()
            res = api.ahocorasick_init(pats)
            api.ahocorasick_finalize(res)
            t1 = time.perf_counter()
            if batched:
                got = api.ahocorasick_match_batch(hays, res)
                n_rec = sum(len(g) for g in got)
            else:
                n_rec = 0
                for hay in hays:
                    n_rec += len(api.ahocorasick_match(hay, res))
            t2 = time.perf_counter()
            api.ahocorasick_deinit(res)
            t3 = time.perf_counter()
            if i:
                build += t1 - t0
                scan += t2 - t1
                deinit += t3 - t2
        out[name] = {"build_ms_per_sample": round(build / samples * 1e3, 3), "scan_ms_per_sample": round(scan / samples * 1e3, 3),
                     "deinit_ms_per_sample": round(deinit / samples * 1e3, 3), "s_per_sample": round((build + scan + deinit) / samples, 5),
                     "records_per_sample": n_rec, "check_ok": bool(n_rec == n_ref),
                     "note": "through the Python mirror of the PHP functions: result arrays (pos / keyIdx / start_postion / value) are built for every record"}
    return out


def density_leg(torch, api, batch, synth, dev, nh=1 << 17):
    """How a whole call degrades with the density of matches: the headline patterns with 1 / 16 / 64 / 256
    planted occurrences per 8 KiB haystack, and the 2,048-word list over prose of which a share is listed words (53 / 200 / 910
    records per 8 KiB), and the 65,536-pattern set of config 3 with 32 / 128 / 512 planted occurrences per 8 KiB (round 6: tens to
    hundreds of candidate windows per work item behind the big-set filter), 1 GiB each, automatic kernel choice (measured:
    engine.cpp choose_strategy); the first 32 (cfg3: 8) haystacks of every cell against the oracle.  ms per algorithmic GB is what
    neighbouring cells are compared by."""
    from oracle import pyoracle
    cells = []
    sets = {"headline": synth.density_set("headline"), "words": synth.density_set("words"), "cfg3": synth.density_set("cfg3")}
    plan = [("headline", "plants", p) for p in (1, 16, 64, 256)] + [("words", "prose", sh) for sh in (0.05, 0.2, 1.0)] + \
           [("cfg3", "plants", p) for p in (32, 128, 512)]   # (the 65,536-pattern set: round 5's weakest cells, 0.007-0.015)
    oracles, handles = {}, {}
    for kind, base in sets.items():
        # (the 65,536-pattern trie takes the port seconds to build and the compiled reference half of that: whichever is there)
        o = pyoracle.Oracle("reference" if kind == "cfg3" and pyoracle.have_reference() else "port")
        o.add_patterns((base.pat_blob, base.pat_off))
        oracles[kind] = o
    for kind, how, level in plan:
        base = sets[kind]
        ds = synth.plant_density(base, level, HAY_LEN) if how == "plants" else synth.prose(base, level)
        res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(base.patterns())])
        reps = nh // ds.n_haystacks
        d_blob = torch.from_numpy(ds.hay_blob).to(dev).repeat(reps).contiguous()
        d_off = torch.arange(nh + 1, dtype=torch.int64, device=dev) * HAY_LEN
        batch.bind_current_stream(res)
        for _ in range(10):   # the measured choice settles
            total = batch.match_device(res, d_blob, d_off, True)
        ms = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            total = batch.match_device(res, d_blob, d_off, True)
            ms.append((time.perf_counter() - t0) * 1e3)
        t = float(np.median(ms))
        prefix, rec = batch.result_numpy(res, nh, total)
        n_chk = 8 if kind == "cfg3" else 32
        pe, re_ = oracles[kind].match_batch(ds.hay_blob[:int(ds.hay_off[n_chk])], ds.hay_off[:n_chk + 1], True)
        ok = np.array_equal(prefix[:n_chk + 1], pe) and np.array_equal(rec[:int(pe[-1])], re_) and total == int(prefix[ds.n_haystacks]) * reps
        algo = d_blob.numel() + 16 * total + 8 * nh
        cells.append({"set": kind, "text": f"{level} planted occurrences per 8 KiB" if how == "plants" else f"prose, {int(level * 100)} % of it listed words",
                      "records_per_8KiB": round(total / nh, 1), "ms_per_call": round(t, 4), "kernel": _kernel_name(res),
                      "algorithmic_bytes": int(algo), "ms_per_algorithmic_GB": round(t / (algo / 1e9), 4),
                      "frac_whole_call": round(algo / (t * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "check_ok": bool(ok),
                      "choice_probes": int(res.stat("choice_probes")), "choice_switches": int(res.stat("choice_switches"))})
        api.ahocorasick_deinit(res)
        del d_blob, d_off
        torch.cuda.empty_cache()
    for o in oracles.values():
        o.close()
    worst = 0.0
    for a, b in zip(cells[:-1], cells[1:]):
        if a["set"] == b["set"]:
            r = max(a["ms_per_algorithmic_GB"], b["ms_per_algorithmic_GB"]) / min(a["ms_per_algorithmic_GB"], b["ms_per_algorithmic_GB"])
            worst = max(worst, r)
    return {"cells": cells, "min_frac_whole_call": min(c["frac_whole_call"] for c in cells), "max_ratio_between_neighbours": round(worst, 2),
            "all_check_ok": bool(all(c["check_ok"] for c in cells))}


def choice_matrix_leg(full=False):
    """The cliff detector of tests/test_gpu_cliffs.py as a driver-visible figure: {2,048 x 16 B; 2,048 words; 65,536 x 8..64 B} x
    {1, 32, 512 occurrences per 8 KiB} x {256 B, 8 KiB, 1 MiB haystacks} (--choice-matrix 2: the test's own five densities), 64 MiB
    batches; in every cell the automatic mode (10 batches to settle) against every forced scan kernel, all record arrays equal
    (the oracle on a slice; the 65,536-pattern set, whose CPU trie takes seconds to build: the four kernels among themselves).  The test itself runs the whole matrix; profiles/r05/cliffs_*.txt are five runs of it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_cliffs import run_matrix, DENSITIES
    out = {"cells": [], "violations": []}
    for kind in ("headline", "words", "cfg3"):
        rows, bad = run_matrix(kind, densities=DENSITIES if full else (1, 32, 512), oracle=full or kind != "cfg3")
        out["cells"] += [{"set": r["set"], "per_8KiB": r["per_8k"], "haystack_bytes": r["hay_len"], "records": r["records"], "automatic_ms": r["auto_ms"],
                          "automatic_kernel": r["auto_kernel"], "best_forced_ms": r["best_forced_ms"], "best_forced_kernel": r["best_forced_kernel"],
                          "forced_ms": r["forced_ms"]} for r in rows]
        out["violations"] += bad
    out["worst_ratio_automatic_over_best_forced"] = round(max(c["automatic_ms"] / c["best_forced_ms"] for c in out["cells"]), 3)
    out["rule"] = "automatic <= 1.3 x best forced + 0.03 ms, every result equal to the oracle's"
    return out


def general_leg(torch, api, batch, res_patterns, d_blob, d_off, nh, expect_records):
    """The headline batch through the general path north_star describes (LDS-tile table walk, k_scan_lds):
    what a pattern set that is not eligible for the anchor filter gets."""
    res = api.ahocorasick_init(res_patterns)
    res.set_option("kernel", 1)
    res.set_option("profile", 2)
    batch.bind_current_stream(res)
    ms, scan, walk, total = _timed_calls(torch, batch, res, d_blob, d_off, calls=5, skip=2)
    out = _leg_summary("the headline batch with kernel=1 (LDS-tile table walk)", d_blob.numel(), nh, total, ms, scan, walk, res,
                       total == expect_records)
    api.ahocorasick_deinit(res)
    return out


def keywords_leg(torch, api, batch, synth, dev, nh=1 << 17, words_only=False):
    """Keyword-like set (the use the reference's README sells): 2,048 patterns of 4..12 bytes over text of letters and spaces,
    2^17 haystacks x 8 KiB = 1 GiB; automatic kernel choice.  `words_only` False (the leg of rounds 1 and 2, kept for
    comparison): the patterns are drawn from the text's own alphabet, spaces included (5 of 31 symbols) -- 115 windows per 8 KiB
    are some pattern's first four bytes; True: the patterns are WORDS, letters only -- 18 per 8 KiB.  Check against the CPU
    oracle on the first 64 haystacks of the tile."""
    alpha = bytes(range(ord("a"), ord("z") + 1)) + b"     "
    ds = synth.generate("words", 33, 2048, 4, 12, 8192, HAY_LEN, alpha)
    if words_only:
        pats = synth.generate("words", 34, 2048, 4, 12, 1, 16, bytes(range(ord("a"), ord("z") + 1)))
        ds = synth.Dataset("words", pats.pat_blob, pats.pat_off, ds.hay_blob, ds.hay_off)
        for h in range(0, 8192, 2):   # a planted word in every second haystack (random text holds the 4-letter ones only)
            p = np.frombuffer(ds.pattern((13 * h + 5) % 2048), dtype=np.uint8)
            at = int(ds.hay_off[h]) + (131 * h + 7) % (HAY_LEN - 16)
            ds.hay_blob[at:at + len(p)] = p
    res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(ds.patterns())])
    for kv in filter(None, os.environ.get("ACMI_KW_OPTS", "").split(",")):   # analysis: e.g. ACMI_KW_OPTS=anchor_dense=0
        res.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    res.set_option("profile", 2)
    d_blob = torch.from_numpy(ds.hay_blob).to(dev).repeat(nh // 8192).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device=dev) * HAY_LEN
    batch.bind_current_stream(res)
    first_kernel = None
    total = batch.match_device(res, d_blob, d_off, True)
    first_kernel = int(res.stat("kernel_used"))
    ms, scan, walk, total = _timed_calls(torch, batch, res, d_blob, d_off, calls=4, skip=1)
    prefix, rec = batch.result_numpy(res, nh, total)
    from oracle import pyoracle
    o = pyoracle.Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    n_chk = 64
    pe, re_ = o.match_batch(ds.hay_blob[:int(ds.hay_off[n_chk])], ds.hay_off[:n_chk + 1], True)
    ok = np.array_equal(prefix[:n_chk + 1], pe) and np.array_equal(rec[:int(pe[-1])], re_) and total == int(prefix[8192]) * (nh // 8192)
    what = "2,048 words of 4..12 letters" if words_only else "2,048 patterns of 4..12 letters and spaces (the text's alphabet)"
    out = _leg_summary(f"{what} over letters + spaces, {nh} haystacks x {HAY_LEN} B ({nh * HAY_LEN >> 20} MiB)",
                       nh * HAY_LEN, nh, total, ms, scan, walk, res, ok)
    out["first_batch_kernel"] = KERNEL_NAMES.get(first_kernel, "?")
    out["records_per_haystack"] = round(total / nh, 2)
    out.update({"anchor_variant": int(res.stat("anchor_variant")), "hit_windows_expected_per_8KiB": round(res.stat("anchor_dense_rate") * 8192, 1),
                "fallback_bytes": int(res.stat("fallback_bytes")), "walked_bytes": int(res.stat("walked_bytes")), "overflow_items": int(res.stat("overflow_items"))})
    api.ahocorasick_deinit(res)
    return out


L2_LINES_PER_S = 34.5e12 / 128   # MI355X_MICROARCH.md: the L2s' 34.5 TB/s in 128-byte lines: what a table walk's rows outside LDS cost


def dictionary_leg(torch, api, batch, synth, dev, nh=1 << 17):
    """Big dictionaries of short words (the reference's README: "especially effective if there is a large database of needles"):
    16,384 and 65,536 words of 4..12 letters over 1 GiB of letters and spaces, automatic kernel choice, the first 16 haystacks
    against the oracle.  No filter applies (14 % of all byte offsets of such text start a 4-gram of the 65,536-word list, and
    nothing shorter than the shortest word can be hashed): it is a table walk, nearly every step of which leaves the LDS tile --
    so next to the HBM fraction the leg states the floor the L2's request rate sets (one 128-byte line per text byte)."""
    from oracle import pyoracle
    letters = bytes(range(ord("a"), ord("z") + 1))
    text = synth.generate("words", 33, 1, 4, 12, 2048, HAY_LEN, letters + b"     ")
    out = []
    for n_words in (16384, 65536):
        pats = synth.generate("words", 34, n_words, 4, 12, 1, 16, letters)
        t0 = time.perf_counter()
        res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(pats.patterns())])
        api.ahocorasick_finalize(res)
        build_ms = (time.perf_counter() - t0) * 1e3
        d_blob = torch.from_numpy(text.hay_blob).to(dev).repeat(nh // 2048).contiguous()
        d_off = torch.arange(nh + 1, dtype=torch.int64, device=dev) * HAY_LEN
        batch.bind_current_stream(res)
        ms, scan, walk, total = _timed_calls(torch, batch, res, d_blob, d_off, calls=4, skip=1)
        prefix, rec = batch.result_numpy(res, nh, total)
        o = pyoracle.Oracle("port")
        o.add_patterns((pats.pat_blob, pats.pat_off))
        n_chk = 16
        pe, re_ = o.match_batch(text.hay_blob[:int(text.hay_off[n_chk])], text.hay_off[:n_chk + 1], True)
        o.close()
        ok = np.array_equal(prefix[:n_chk + 1], pe) and np.array_equal(rec[:int(pe[-1])], re_) and total == int(prefix[2048]) * (nh // 2048)
        leg = _leg_summary(f"{n_words} words of 4..12 letters over letters + spaces, {nh} haystacks x {HAY_LEN} B ({nh * HAY_LEN >> 20} MiB)",
                           nh * HAY_LEN, nh, total, ms, scan, walk, res, ok)
        floor_ms = nh * HAY_LEN / L2_LINES_PER_S * 1e3
        leg.update({"words": n_words, "states": int(res.stat("states")), "automaton_bytes": int(res.stat("automaton_bytes")), "build_ms": round(build_ms, 1),
                    "records_per_haystack": round(total / nh, 2),
                    "l2_request_floor_ms": round(floor_ms, 3), "l2_request_floor": "one 128-byte line per text byte (every step outside the LDS tile: the 65,536-word list; the 16,384-word list leaves it at ~ 0.6 of its steps) at the L2s' 34.5 TB/s (DESIGN 4.16)",
                    "floor_over_measured": round(floor_ms / ms, 3)})
        out.append(leg)
        api.ahocorasick_deinit(res)
        del d_blob, d_off
        torch.cuda.empty_cache()
    return out


def short_patterns_leg(torch, api, batch, synth, dev, ds, nh=1 << 17):
    """The headline pattern set (2,048 x 16 B over abcdef) PLUS eight patterns of 2..6 B -- the reference takes any length from
    1 to 1,024 B in one trie (src/multifast/ahocorasick.c:101-105); here a pattern under 7 B takes the whole set off the
    aligned-anchor filter (and, over this six-letter alphabet, off the dense form too: every 4-letter window is some pattern's
    prefix).  Two cases over 1 GiB of the headline text: the short patterns over letters the text does not hold (no match: what
    the set costs), and over the text's own alphabet (a 2-byte pattern then matches every 36 bytes: dense output)."""
    from oracle import pyoracle
    out = {}
    reps = nh // TILE_HAYSTACKS
    d_blob = torch.from_numpy(ds.hay_blob).to(dev).repeat(reps).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device=dev) * HAY_LEN
    for name, alpha in (("absent_letters", b"uvwxyz"), ("same_alphabet", b"abcdef")):
        sp = synth.generate("short", 41, 8, 2, 6, 1, 16, alpha)
        pats = ds.patterns() + sp.patterns()
        res = api.ahocorasick_init([{"id": k, "value": p} for k, p in enumerate(pats)])
        batch.bind_current_stream(res)
        batch.match_device(res, d_blob, d_off, True)
        ms, scan, walk, total = _timed_calls(torch, batch, res, d_blob, d_off, calls=4, skip=1)
        prefix, rec = batch.result_numpy(res, nh, total)
        o = pyoracle.Oracle("port")
        o.add_patterns(pats)
        n_chk = 32
        pe, re_ = o.match_batch(ds.hay_blob[:int(ds.hay_off[n_chk])], ds.hay_off[:n_chk + 1], True)
        ok = np.array_equal(prefix[:n_chk + 1], pe) and np.array_equal(rec[:int(pe[-1])], re_) and total == int(prefix[TILE_HAYSTACKS]) * reps
        leg = _leg_summary(f"headline patterns + {[p.decode() for p in sp.patterns()]} over {nh} haystacks x {HAY_LEN} B", nh * HAY_LEN, nh, total, ms, scan, walk, res, ok)
        leg["records_per_haystack"] = round(total / nh, 2)
        out[name] = leg
        api.ahocorasick_deinit(res)
    del d_blob
    return out


def host_leg(api, synth, ds, expect_per_tile):
    """The host-buffer entry the Zend shim calls (acmi_match_batch: borrowed host memory in, host records out):
    PCIe-inclusive rate on a 512 MiB batch, the literal 256 x 8 KiB batch of BASELINE configs[1] as ONE call, and
    single ahocorasick_match calls of 8 KiB (examples/benchmark.php:65-67 loops over these)."""
    res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
    nh = 1 << 16
    blob = np.tile(ds.hay_blob, nh // TILE_HAYSTACKS)
    off = np.arange(nh + 1, dtype=np.uint64) * HAY_LEN
    api.match_arrays(res, blob, off)
    best = 1e9
    n_rec = 0
    for _ in range(3):
        t0 = time.perf_counter()
        prefix, rec = api.match_arrays(res, blob, off)
        best = min(best, time.perf_counter() - t0)
        n_rec = len(rec)
    # what the copy alone costs: the same 512 MiB, pageable host memory -> HBM (the entry is bound by this, not by the scan)
    import torch
    dst = torch.empty(blob.nbytes, dtype=torch.uint8, device="cuda")
    src = torch.from_numpy(blob)
    h2d = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(src)
        torch.cuda.synchronize()
        h2d = min(h2d, time.perf_counter() - t0)
    del dst
    lit_blob, lit_off = ds.hay_blob[:256 * HAY_LEN], off[:257]
    api.match_arrays(res, lit_blob, lit_off)
    lit = []
    for _ in range(20):
        t0 = time.perf_counter()
        p2, r2 = api.match_arrays(res, lit_blob, lit_off)
        lit.append(time.perf_counter() - t0)
    one = ds.haystack(3)
    api.ahocorasick_match(one, res)
    t0 = time.perf_counter()
    for _ in range(200):
        api.ahocorasick_match(one, res)
    single = (time.perf_counter() - t0) / 200
    # the same call through the C ABI alone (what a Zend shim pays before it builds its result arrays)
    import ctypes as C
    from php_aho_corasick_amd import _native as N
    buf = np.frombuffer(one, dtype=np.uint8)
    recs, n_out = C.POINTER(N.Record)(), C.c_size_t(0)
    c_args = (res._h, buf.ctypes.data, len(one), 1, C.byref(recs), C.byref(n_out))
    for _ in range(5):
        N.lib.acmi_match(*c_args)
    t0 = time.perf_counter()
    for _ in range(200):
        N.lib.acmi_match(*c_args)
    single_c = (time.perf_counter() - t0) / 200
    res.set_option("profile", 1)
    api.ahocorasick_match(one, res)
    small_kernel_us = res.stat("scan_ms") * 1e3
    res.set_option("profile", 0)
    out = {"batch_512MiB": {"GB_per_s_pcie_inclusive": round(blob.nbytes / best / 1e9, 2), "ms": round(best * 1e3, 2), "records": n_rec,
                            "check_ok": bool(n_rec == expect_per_tile * (nh // TILE_HAYSTACKS)),
                            "h2d_copy_alone_GB_per_s": round(blob.nbytes / h2d / 1e9, 2), "pcie_gen5_x16_spec_GB_per_s": 63.0},
           "literal_256x8KiB_one_call_ms": round(float(np.median(lit)) * 1e3, 4), "literal_records": len(r2),
           "single_match_8KiB_us": round(single * 1e6, 1), "single_match_8KiB_us_c_abi_only": round(single_c * 1e6, 1),
           "single_match_8KiB_kernel_us": round(small_kernel_us, 1)}
    api.ahocorasick_deinit(res)
    return out



def _sclk_mhz(pci_bus=None):
    """Shader clock right now (sysfs pp_dpm_sclk: the level marked '*') of the GPU on PCI bus `pci_bus` -- the one this rank
    runs on -- or of every GPU the kernel driver exposes when the bus is unknown, so that a reader can tell which clock
    level a run sat on; {} when the files are not there."""
    import glob
    out = {}
    for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            card = path.split("/")[4]
            bus = int(os.path.realpath(os.path.dirname(path)).split("/")[-1].split(":")[1], 16)
            if pci_bus is not None and bus != pci_bus:
                continue
            for line in open(path).read().splitlines():
                if line.rstrip().endswith("*"):
                    out[card] = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
        except Exception:
            pass
    return out


def _stats(xs):
    xs = [float(x) for x in xs]
    if not xs:
        return None
    return {"min": round(min(xs), 4), "median": round(float(np.median(xs)), 4), "max": round(max(xs), 4), "n": len(xs)}


class Workload:
    """What a step scans: a pattern set and a 64-MiB-ish tile of haystacks that is replicated on the device."""

    def __init__(self, name, synth, planted=True):
        self.name = name
        if name == "config3":
            # BASELINE configs[2] / [3]: 65,536 patterns of 8..64 B over Sigma=256 (seed 7: the G6 automaton), 4 KiB haystacks
            self.tile_h, self.hay_len, self.default_nh = 4096, 4096, 1 << 20
            self.ds = synth.g6(n_hay=self.tile_h, hay_len=self.hay_len)
            self.label = "65,536 patterns of 8..64 B (Sigma=256, seed 7: 2,269,830 states)"
            self.shape = "BASELINE configs[2] shape (configs[3] when sharded)"
        else:
            self.tile_h, self.hay_len, self.default_nh = TILE_HAYSTACKS, HAY_LEN, 1 << 19
            self.ds = synth.bench_batch(self.tile_h, self.hay_len, seed=1, planted=planted)
            self.label = "2048 patterns x 16 B (abcdef, seed 1)"
            self.shape = "BASELINE configs[1] shape"
        self.planted = planted

    def patterns(self):
        return [{"id": k, "value": self.ds.pattern(k)} for k in range(self.ds.n_patterns)]

    def check(self, prefix, rec, nh, total):
        """Every haystack carries its planted record; the tile's record count repeats over the replicas."""
        reps = nh // self.tile_h
        tile_count = int(prefix[self.tile_h])
        ok = int(prefix[-1]) == total and total == tile_count * reps
        if not self.planted:
            return ok, tile_count
        ok = ok and total >= nh
        if self.name == "config3":
            ds = self.ds
            h = np.arange(self.tile_h, dtype=np.int64)
            pat = (1021 * h + 17) % ds.n_patterns
            start = (37 * h + 11) % (self.hay_len - 64)
            plen = np.diff(ds.pat_off.astype(np.int64))[pat]
            want = np.stack([h, start + plen, start, pat], axis=1).astype(np.uint32)
            have = set(map(tuple, rec[:tile_count].tolist()))
            ok = ok and all(tuple(w) in have for w in want.tolist())
        return ok, tile_count


def short_haystacks_leg(torch, api, batch, synth, dev):
    """The headline pattern set over 1 GiB of SHORT haystacks (work items shorter than the anchor kernel's 8-KiB wave step):
    256 B and 1 KiB regular, 4,001 B back to back (every start unaligned).  Which template ran is in `anchor_variant`
    (window dwords | 8 big set | 16 short rows | 32 tiny items) and `tiny_waves`; the first 64 haystacks are checked
    against the oracle."""
    from oracle import pyoracle
    out = []
    for hl in (256, 1024, 4001):
        n_tile = max(1, (32 << 20) // hl)
        ds = synth.generate("L", 1, 2048, 16, 16, n_tile, hl, b"abcdef")
        every = max(1, HAY_LEN // hl)   # the headline's hit density: one planted pattern per 8 KiB of text
        for h in range(0, n_tile, every):
            p = np.frombuffer(ds.pattern((7 * h + 3) % ds.n_patterns), dtype=np.uint8)
            at = int(ds.hay_off[h]) + (31 * h + 5) % (hl - 16)
            ds.hay_blob[at:at + 16] = p
        res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
        reps = max(1, (1 << 30) // (n_tile * hl))
        nh = reps * n_tile
        d_blob = torch.from_numpy(ds.hay_blob).to(dev).repeat(reps).contiguous()
        d_off = torch.arange(nh + 1, dtype=torch.int64, device=dev) * hl
        batch.bind_current_stream(res)
        ms, scan, walk, total = _timed_calls(torch, batch, res, d_blob, d_off, calls=5, skip=2)
        prefix, rec = batch.result_numpy(res, nh, total)
        o = pyoracle.Oracle("port")
        o.add_patterns((ds.pat_blob, ds.pat_off))
        n_chk = 64
        pe, re_ = o.match_batch(ds.hay_blob[:int(ds.hay_off[n_chk])], ds.hay_off[:n_chk + 1], True)
        ok = np.array_equal(prefix[:n_chk + 1], pe) and np.array_equal(rec[:int(pe[-1])], re_) and total == int(prefix[n_tile]) * reps and total >= nh // every
        leg = _leg_summary(f"headline patterns over {nh} haystacks x {hl} B ({nh * hl >> 20} MiB){', every start unaligned' if hl % 16 else ''}, one planted pattern per 8 KiB of text",
                           nh * hl, nh, total, ms, scan, walk, res, ok)
        leg.update({"haystack_bytes": hl, "anchor_variant": int(res.stat("anchor_variant")), "tiny_waves": int(res.stat("tiny_waves")),
                    "work_item_bytes": int(res.stat("chunk_bytes"))})
        if hl % 16 == 0:
            # the entry the headline uses (acmi_match_batch_device_begin_uniform: no offsets array, the item table in closed form):
            # what a caller with a regular batch gets; the figures above are the offsets entry's, as in earlier rounds
            res.set_option("profile", 0)
            uni = []
            for i in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                batch.match_device_begin_uniform(res, d_blob, hl, nh, True)
                t_uni = batch.match_device_end(res)
                if i >= 2:
                    uni.append((time.perf_counter() - t0) * 1e3)
            p_uni, r_uni = batch.result_numpy(res, nh, t_uni)
            leg["uniform_entry"] = {"ms_per_call": round(float(np.median(uni)), 4),
                                    "frac_whole_call": round(leg["algorithmic_bytes"] / (float(np.median(uni)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                    "check_ok": bool(t_uni == total and np.array_equal(p_uni, prefix) and np.array_equal(r_uni, rec))}
        out.append(leg)
        api.ahocorasick_deinit(res)
        del d_blob, d_off
        torch.cuda.empty_cache()
    return out


def replace_leg(torch, api, batch, synth, dev, nh=1 << 17):
    """multifast_replace on the device (reference src/multifast/replace.c:469-568): the headline batch shape
    (one planted pattern per 8 KiB haystack, every pattern with a replacement text "<k>"), input and output resident in HBM:
    acmi_replace_batch_device = the match pipeline + nominee plan + device prefix sum of the output lengths + the copy.
    Rate = (bytes read + bytes written) per second against the 8 TB/s; the first 32 haystacks against the oracle."""
    from oracle import pyoracle
    ds = synth.bench_batch(TILE_HAYSTACKS, HAY_LEN, seed=1, planted=True)
    res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
    o = pyoracle.Oracle("port")
    for k in range(ds.n_patterns):
        r = b"<%d>" % k
        api.set_replacement(res, k, r)
        o.set_replacement(k, r)
    o.add_patterns((ds.pat_blob, ds.pat_off))
    reps = nh // TILE_HAYSTACKS
    d_blob = torch.from_numpy(ds.hay_blob).to(dev).repeat(reps).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device=dev) * HAY_LEN
    batch.bind_current_stream(res)
    ms = []
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, off = batch.replace_device(res, d_blob, d_off, 1)
        if i >= 2:
            ms.append((time.perf_counter() - t0) * 1e3)
    t = float(np.mean(ms))
    off_h = off[:33].cpu().numpy()
    out_h = out[:int(off_h[-1])].cpu().numpy()
    ok = int(off[-1].item()) == out.numel()
    for h in range(32):
        ok = ok and out_h[int(off_h[h]):int(off_h[h + 1])].tobytes() == o.replace(ds.haystack(h), 1)
    moved = d_blob.numel() + out.numel()
    leg = {"workload": f"headline patterns, each with a replacement text, over {nh} haystacks x {HAY_LEN} B ({nh * HAY_LEN >> 20} MiB), "
                       f"one replacement per haystack; input and output resident",
           "ms_per_call": round(t, 4), "bytes_in": int(d_blob.numel()), "bytes_out": int(out.numel()),
           "in_plus_out_GB_per_s": round(moved / t / 1e6, 1), "frac_of_hbm_peak": round(moved / t / 1e6 / HBM_PEAK_GBS, 4), "check_ok": bool(ok)}
    api.ahocorasick_deinit(res)
    return leg


def _leg_scalar(name, leg):
    """One leg of the bench reduced to what the result line carries: the smallest roofline fraction any of its cells reached
    (whole calls; a leg that reports no fraction carries its own headline figure) and whether every check in it passed."""
    fracs, oks = [], []

    def walk(x):
        if isinstance(x, dict):
            for k, v in x.items():
                if k in ("frac_whole_call", "frac_of_hbm_peak") and isinstance(v, (int, float)):
                    fracs.append(float(v))
                elif k in ("check_ok", "count_ok", "content_ok", "all_check_ok", "ok") and isinstance(v, bool):
                    oks.append(v)
                else:
                    walk(v)
        elif isinstance(x, list):
            for v in x:
                walk(v)
    walk(leg)
    s = {"ok": bool(all(oks)) if oks else None}
    if fracs:
        s["frac"] = round(min(fracs), 4)
    if name == "choice_matrix":
        s = {"ok": not leg["violations"], "violations": len(leg["violations"]), "worst_ratio": leg["worst_ratio_automatic_over_best_forced"]}
    elif name == "host_entry":
        s["pcie_inclusive_GB_per_s"] = leg["batch_512MiB"]["GB_per_s_pcie_inclusive"]
        s["single_match_8KiB_us"] = leg.get("single_match_8KiB_us")
    elif name == "benchmark_php_sample":
        for k in ("s_per_sample", "gpu_s_per_sample"):
            if k in leg:
                s[k] = leg[k]
    return s


RESULT_LINE_LIMIT = 4096   # bytes; the driver reads the LAST stdout line, and a 26 KB one (round 4) was more than it takes


def _result_line(out, detail, leg_s):
    """The bench's result: the contract keys, `roofline`, `cpu_baseline` and one scalar pair per secondary leg -- short enough
    for the driver to parse.  Everything else (`detail`) goes to an earlier, prefixed line and to gpurun_out/bench_legs.json."""
    line = dict(out)
    line["legs"] = {name: _leg_scalar(name, leg) for name, leg in detail.items()}
    line["legs_seconds"] = round(sum(leg_s.values()), 1)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= RESULT_LINE_LIMIT - 256:   # never expected; what goes first is commentary, not a figure
        for key in ("note",):
            line.get("ranks", {}).pop(key, None)
        for key in ("kernel_ms_per_step", "host_ms_between_step_ends", "kernels_of_timed_steps", "sclk_mhz_start", "sclk_mhz_end"):
            line.get("roofline", {}).pop(key, None)
        for c in (line.get("cpu_baseline"), line.get("cpu_baseline_all_cores")):
            if c:
                c["sample"] = c["sample"][:80]
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < RESULT_LINE_LIMIT, f"result line of {len(text)} bytes"
    return text


def _emit(out, detail, leg_s):
    full = dict(out)
    full.update(detail)
    full["leg_seconds"] = leg_s
    blob = json.dumps(full)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_legs.json"), "w") as f:
            f.write(blob + "\n")
    except OSError:
        pass
    if detail:
        print("BENCH_DETAIL " + blob, flush=True)   # (prefixed: not a JSON line, cannot be taken for the result)
    print(_result_line(out, detail, leg_s), flush=True)


def _spawn(args):
    """`python bench.py --gpus N` without a launcher: become N ranks under torch.distributed.run (one per GPU, RCCL).
    Fewer than N GPUs is an error, not a silent single-GPU run (ACMI_BENCH_BACKEND=gloo, ranks sharing GPU 0, exists to
    exercise this very path on a one-GPU box)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("ACMI_BENCH_BACKEND", "nccl") == "nccl":
        print(f"bench.py: --gpus {args.gpus} but this box has {have} GPU(s); refusing to run a smaller job under that name", file=sys.stderr)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _c_abi_steps(torch, dist, api, batch, res, rank, world, dev, d_blob, hay_len, nh, expect_total, steps):
    """K steps whose exchange is the C ABI's own (acmi_comm_allgather_result: RCCL bound by libacmi.so itself -- what a non-Python
    host would call; the id travels over the process group): scan, then the all-gatherv of the match arrays, one after the other
    (the C call is synchronous), timed like the torch.distributed steps: barrier + synchronize on both sides, MAX over ranks."""
    import ctypes as C
    from php_aho_corasick_amd import _native as N
    ident = torch.zeros(128, dtype=torch.uint8, device=dev if dist.get_backend() == "nccl" else "cpu")
    if rank == 0:
        buf = (C.c_char * 128)()
        N.check(N.lib.acmi_comm_get_id(buf))
        ident.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
    dist.broadcast(ident, 0)
    raw = (C.c_char * 128).from_buffer_copy(ident.cpu().numpy().tobytes())
    comm = C.c_void_p()
    N.check(N.lib.acmi_comm_create(res._h, rank, world, raw, C.byref(comm)), res._h)
    dp, dr, nr, nhg = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()

    def step():
        batch.match_device_begin_uniform(res, d_blob, hay_len, nh, True)
        total = batch.match_device_end(res)
        N.check(N.lib.acmi_comm_allgather_result(comm, res._h, nh, C.byref(dp), C.byref(dr), C.byref(nr), C.byref(nhg)), res._h)
        return total
    total = step()   # (warm-up: the communicator's buffers)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        total = step()
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    v = C.c_double()
    stat = {}
    for key in ("backend_ranks", "transport", "rounds", "bytes_sent", "bytes_received"):
        N.lib.acmi_comm_get_stat(comm, key.encode(), C.byref(v))
        stat[key] = int(v.value)
    ok = nr.value == expect_total * world and nhg.value == nh * world and total == expect_total
    N.lib.acmi_comm_destroy(comm)
    return {"ok": bool(ok), "steps": steps, "ms_per_step": round(float(dt.item()) / steps * 1e3, 4), "records": int(nr.value), "haystacks": int(nhg.value), **stat}


def _config3_sharded(torch, dist, api, batch, synth, rank, world, dev, backend, steps=3):
    """BASELINE configs[3] next to the headline line of an N > 1 run: the 65,536-pattern set over 2^16 haystacks x 4 KiB per rank
    (256 MiB), sharded like the headline (contiguous haystack ranges, replicated automaton, all-gatherv of the match arrays);
    every plant of the tile checked on every rank, the gathered arrays against the shards' counts."""
    wl = Workload("config3", synth)
    res = api.ahocorasick_init(wl.patterns())
    api.ahocorasick_finalize(res)
    nh = 1 << 16
    reps = nh // wl.tile_h
    d_blob = torch.from_numpy(wl.ds.hay_blob).to(dev).repeat(reps).contiguous()
    batch.bind_current_stream(res)
    gatherer = batch.RecordGatherer(None, dev, capacity=nh + nh // 8, depth=2)

    def step():
        batch.match_device_begin_uniform(res, d_blob, wl.hay_len, nh, True)
        total = batch.match_device_end(res)
        return total, gatherer.finish(gatherer.start(batch.result_view(res, total), nh, rank * nh))
    total, got = step()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        total, got = step()
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    prefix, rec = batch.result_numpy(res, nh, total)
    ok, _tile = wl.check(prefix, rec, nh, total)
    ok = ok and int(got[1].shape[0]) == total * world and int(got[0].shape[0]) - 1 == nh * world
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    api.ahocorasick_deinit(res)
    t = float(dt.item()) / steps
    return {"workload": f"{wl.label} over {nh} haystacks x {wl.hay_len} B per rank, {world} ranks ({backend})", "check_ok": bool(flag.item()),
            "ms_per_step": round(t * 1e3, 4), "haystack_GB_per_s": round(world * nh * wl.hay_len / t / 1e9, 1), "records_per_step": int(got[1].shape[0])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("headline", "config3"), default="headline",
                    help="headline: 2048 x 16 B over 8 KiB haystacks (BASELINE configs[1], the metric's config); config3: 65,536 patterns over "
                         "4 KiB haystacks (configs[2]; with --gpus N > 1 that is configs[3])")
    ap.add_argument("--haystacks", type=int, default=0, help="haystacks per GPU (default: 2^19 x 8 KiB headline, 2^20 x 4 KiB config3)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, what the driver's scaling runs use): --haystacks per GPU; strong: --haystacks in total, split into contiguous shards")
    ap.add_argument("--ilp", type=int, default=int(os.environ.get("ACMI_ILP", "0")))
    ap.add_argument("--group-quads", type=int, default=0)
    ap.add_argument("--kernel", type=int, default=-1, help="scan kernel: 1 LDS-tile table walk, 2 anchor filter, default automatic")
    ap.add_argument("--no-plant", action="store_true", help="analysis only: no planted hits (the reference's own benchmark has ~0 matches)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-dense", action="store_true", help="skip the secondary BASELINE configs[4] (dense-match) measurement")
    ap.add_argument("--no-legs", action="store_true", help="skip the other secondary legs (config 3, general path, keyword set, host-buffer entry, ...)")
    ap.add_argument("--choice-matrix", type=int, default=1, help="0: leave the kernel-choice matrix (tests/test_gpu_cliffs.py's, the longest leg) out of the default legs; 2: all five densities of the test instead of three")
    ap.add_argument("--legs", default="", help="comma-separated subset of the secondary legs to run (default: all)")
    ap.add_argument("--events", type=int, default=1, help="analysis: 0 = no HIP events around the scan kernel in the timed region (roofline fields are then meaningless)")
    ap.add_argument("--in-flight", type=int, default=2, help="batches begun before the oldest is ended (1 or 2)")
    ap.add_argument("--phase-times", action="store_true", help="analysis: add the host time per step phase to the JSON")
    ap.add_argument("--items-ahead", action="store_true", help="analysis: every batch's item table on the handle's side stream (acmi.h \"items_ahead\" 1; measured: no gain)")
    ap.add_argument("--offsets-entry", action="store_true", help="time acmi_match_batch_device_begin (explicit offsets array) instead of _begin_uniform")
    ap.add_argument("--reserve-cus", type=int, default=0, help="analysis: acmi.h \"reserve_cus\" -- the scan leaves N compute units to a collective in flight")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="analysis: acmi_set_option on the headline handle (repeatable), e.g. --option scan_verify=0")
    ap.add_argument("--no-c-abi-gather", action="store_true", help="N > 1: skip the untimed exchange through acmi_comm_* after the timed steps")
    args = ap.parse_args()

    # `python bench.py --gpus N` on its own: N ranks under torch.distributed.run (the driver launches that itself for N > 1)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(_spawn(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    from php_aho_corasick_amd import synth
    wl = Workload(args.workload, synth, planted=not args.no_plant)
    ds = wl.ds

    # (the CPU baseline's one-core pass runs on a thread of its own, begun BEHIND the timed region and the one-at-a-time steps: it
    # has the box's cores to itself but for the host thread that drives the untimed secondary legs; joined, and all cores
    # measured, when those legs are through: cpu_baseline_finish below)
    cpu1 = cpu_all = cpu_started = None
    want_cpu = rank == 0 and args.gpus == 1 and not args.no_cpu and args.workload == "headline"

    import torch
    import torch.distributed as dist
    from php_aho_corasick_amd import api, batch

    # one rank per GPU; ACMI_BENCH_BACKEND=gloo (+ ranks sharing a device) exists only to exercise the
    # N > 1 code path on a single-GPU box -- the driver's multi-GPU runs use nccl (= RCCL over xGMI)
    backend = os.environ.get("ACMI_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if world > n_dev and backend == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} GPU(s): RCCL wants one GPU per rank")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    gather = world > 1 or os.environ.get("ACMI_BENCH_FORCE_GATHER") == "1"  # analysis: gather path at N=1
    if gather:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev} if backend == "nccl" else {}))
        elif backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    try:
        pci_bus = int(torch.cuda.get_device_properties(dev_index).pci_bus_id)
    except Exception:
        pci_bus = None

    # ---- automaton (replicated on every rank) ----
    t0 = time.perf_counter()
    res = api.ahocorasick_init(wl.patterns())
    api.ahocorasick_finalize(res)
    build_ms = (time.perf_counter() - t0) * 1e3
    res.set_option("profile", 1 if args.events else 0)
    if args.ilp:
        res.set_option("ilp", args.ilp)
    if args.group_quads:
        res.set_option("group_quads", args.group_quads)
    if args.kernel >= 0:
        res.set_option("kernel", args.kernel)
    if args.reserve_cus:
        res.set_option("reserve_cus", args.reserve_cus)
    if args.items_ahead:   # analysis: every batch's work-item table built on the handle's side stream (acmi.h "items_ahead")
        res.set_option("items_ahead", 1)
    for kv in args.option:
        k_, v_ = kv.split("=")
        res.set_option(k_, int(v_))

    # ---- batch resident in HBM: the tile replicated to the requested size ----
    want = args.haystacks or wl.default_nh
    per_gpu = want // world if args.scaling == "strong" else want
    nh = max(wl.tile_h, (per_gpu // wl.tile_h) * wl.tile_h)
    reps = nh // wl.tile_h
    tile = torch.from_numpy(ds.hay_blob).to(dev)
    d_blob = tile.repeat(reps).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device=dev) * wl.hay_len
    del tile
    batch.bind_current_stream(res)
    hay_base = rank * nh
    bytes_per_step = nh * wl.hay_len

    # what a plain streaming read of the same buffer reaches on this chip (context for the roofline)
    import ctypes as C
    from php_aho_corasick_amd import _native as N
    batch.match_device(res, d_blob[:16], d_off[:1], True)  # empty batch: initialises the device, launches nothing
    ms = C.c_double(0)
    N.check(N.lib.acmi_debug_stream_read(res._h, C.c_void_p(d_blob.data_ptr()), d_blob.numel(), 3, C.byref(ms)), res._h)
    stream_read_gbs = d_blob.numel() / (ms.value * 1e-3) / 1e9

    pending = []
    scan_ms = []
    walk_ms = []
    variants_seen = []  # k_scan_anchor template of every timed step (0: another kernel ran)
    kernels_seen = []   # kernel of every timed step (automatic mode is a measured choice: a step may be a probe of another kernel)
    end_stamps = []
    records_per_step = 0

    phase = {"begin": 0.0, "gather": 0.0, "end": 0.0}   # host seconds, --phase-times

    # A step is one batch through the whole pipeline.  Up to `in_flight` (default 2) batches are begun before the oldest
    # is ended: the second one's kernels are queued behind the first one's, so the device does not idle while the host
    # turns around (acmi.h, acmi_match_batch_device_begin).  K steps = K batches begun and ended inside the timed region.
    depth = max(1, min(2, args.in_flight))

    def begin_one():
        t_a = time.perf_counter()
        # (the batch is uniform -- every haystack wl.hay_len bytes, back to back, like examples/benchmark.php's buffers: the entry
        # for that shape needs no offsets array and writes the work-item table in closed form; --offsets-entry: the general one)
        if args.offsets_entry:
            batch.match_device_begin(res, d_blob, d_off, True)
        else:
            batch.match_device_begin_uniform(res, d_blob, wl.hay_len, nh, True)
        phase["begin"] += time.perf_counter() - t_a

    def end_one():
        nonlocal records_per_step
        t_a = time.perf_counter()
        total = batch.match_device_end(res)
        t_b = time.perf_counter()
        end_stamps.append(t_b)
        records_per_step = total
        scan_ms.append(res.stat("scan_ms"))
        kernels_seen.append(int(res.stat("kernel_used")))
        variants_seen.append(int(res.stat("anchor_variant")))
        if gather:
            # all-gatherv of the match arrays: one asynchronous fixed-capacity collective per step, in flight on RCCL's
            # stream while later steps scan.  The payload copy is enqueued here, on the scan's stream, BEFORE the next
            # begin: that begin reuses this result's buffers (valid until the second begin after its own).
            while len(pending) > 2:  # three collectives in flight: the oldest has long finished, finish() does not block
                gatherer.finish(pending.pop(0))
            pending.append(gatherer.start(batch.result_view(res, total), nh, hay_base))
        phase["end"] += t_b - t_a
        phase["gather"] += time.perf_counter() - t_b

    def run_steps(k, in_flight):
        began = ended = 0
        while ended < k:
            while began < k and began - ended < in_flight:
                begin_one()
                began += 1
            end_one()
            ended += 1

    # capacity of the fixed-size collective: the planted workload yields one record per haystack (+ slack;
    # a batch that exceeds it costs one more, larger collective and the capacity stays grown)
    gatherer = batch.RecordGatherer(None, dev, capacity=max(1 << 16, nh + nh // 8), depth=3) if gather else None

    def drain():
        out = None
        while pending:
            out = gatherer.finish(pending.pop(0))
        return out

    run_steps(args.warmup, depth)
    drain()
    scan_ms.clear()
    walk_ms.clear()
    kernels_seen.clear()
    variants_seen.clear()
    end_stamps.clear()
    for k_ in phase:
        phase[k_] = 0.0
    if gather:
        dist.barrier()
    torch.cuda.synchronize()
    clocks_start = _sclk_mhz(pci_bus)   # (right behind the warm-up steps: the level the timed region starts on)
    t0 = time.perf_counter()
    run_steps(args.steps, depth)
    gathered = drain()
    torch.cuda.synchronize()
    if gather:
        dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    ranks_in_collective = 1
    if gather:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)                          # how many ranks the backend's collective really joined
        ranks_in_collective = int(one.item())
    dt = float(t.item())
    clocks_end = _sclk_mhz(pci_bus)
    step_gaps = [(b - a) * 1e3 for a, b in zip(end_stamps[:-1], end_stamps[1:])]   # host time between consecutive ends

    # the same steps one at a time (begin, end, begin, ...): what a caller that cannot keep two batches in flight sees
    n_sync = min(args.steps, 20)
    scan_keep, kernels_keep = list(scan_ms), list(kernels_seen)
    timed_variants = [v for v, k in zip(variants_seen, kernels_seen) if k == 2]
    timed_kernel = max(set(kernels_seen), key=kernels_seen.count) if kernels_seen else int(res.stat("kernel_used"))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run_steps(n_sync, 1)
    drain()
    torch.cuda.synchronize()
    sync_ms_per_step = (time.perf_counter() - t1) / n_sync * 1e3
    scan_ms[:] = scan_keep
    kernels_seen[:] = kernels_keep

    # the candidate walk (second kernel of a step) timed with its own events on a few extra, untimed steps: every
    # event is a barrier on the stream, the timed region carries only the two around the scan kernel
    res.set_option("profile", 2)
    for _ in range(5):
        batch.match_device(res, d_blob, d_off, True)
        walk_ms.append(res.stat("walk_ms"))
    res.set_option("profile", 1)

    if want_cpu:   # (nothing of the headline line is timed from here on)
        cpu_started = cpu_baseline_start(ds)

    # sanity: every haystack carries its planted record; the tile's count repeats
    prefix, rec = batch.result_numpy(res, nh, records_per_step)
    ok, tile_count = wl.check(prefix, rec, nh, records_per_step)
    gathered_records = gathered_haystacks = None
    if gathered is not None:
        gathered_records, gathered_haystacks = int(gathered[1].shape[0]), int(gathered[0].shape[0]) - 1
        # the gathered GLOBAL arrays: every rank's shard in rank order (weak scaling: the same shard `world` times)
        ok = ok and int(gathered[0][-1].item()) == gathered_records == records_per_step * world and gathered_haystacks == nh * world
        ok = ok and bool((gathered[0][::nh][:world + 1].cpu() == torch.arange(world + 1) * records_per_step).all())

    if rank == 0:
        # (the roofline figure is the timed kernel's own mean: a step that automatic mode spent probing another kernel is not in it)
        own = [m for m, k in zip(scan_ms, kernels_seen) if k == timed_kernel] or scan_ms
        kernel_ms = float(np.mean(own)) if own else float("nan")
        if not args.events:
            kernel_ms = 1e9  # (analysis run without events: no kernel time, the roofline fields read ~ 0)
        walk_kernel_ms = float(np.mean(walk_ms)) if walk_ms else 0.0
        algo_bytes = bytes_per_step + 16 * records_per_step + 8 * nh
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.workload == "headline":
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "haystack_GB_per_s", "value": round(world * bytes_per_step * args.steps / dt / 1e9, 2),
            "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{wl.label} over {nh} haystacks x {wl.hay_len} B per GPU "
                                   f"({bytes_per_step >> 20} MiB), one planted hit per haystack; {wl.shape}",
                       "patterns": ds.n_patterns, "haystacks_per_gpu": nh, "haystack_bytes": wl.hay_len,
                       "states": int(res.stat("states")), "hot_states": int(res.stat("hot_states")),
                       "lds_tile_bytes": int(res.stat("lds_image_bytes")), "work_item_bytes": int(res.stat("chunk_bytes")),
                       "ilp": args.ilp or 1, "items_ahead": int(args.items_ahead), "reserve_cus": args.reserve_cus,
                       "entry": "acmi_match_batch_device_begin" if args.offsets_entry else "acmi_match_batch_device_begin_uniform", "sharding": f"{world} contiguous haystack ranges, {backend} all-gatherv of match arrays" if world > 1 else "none"},
            "ranks": {"launched": world, "process_group": (dist.get_world_size() if gather else 1), "joined_a_collective": ranks_in_collective,
                      "backend": backend if gather else None, "gpus_on_box": n_dev,
                      "gathered_records": gathered_records, "gathered_haystacks": gathered_haystacks, "c_abi_exchange": None,
                      "note": "one line per N: no scaling curve is measured by a single run (the driver's SCALE run over N = 1, 2, 4, 8 computes efficiency)"},
            "matches_per_s": round(world * records_per_step * args.steps / dt, 1),
            "records_per_step_per_gpu": records_per_step, "check_ok": bool(ok),
            "automaton_build_ms": round(build_ms, 2),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": KERNEL_NAMES.get(timed_kernel, "?"), "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes": algo_bytes,
                         "kernels_of_timed_steps": {KERNEL_NAMES.get(k, str(k)): kernels_seen.count(k) for k in sorted(set(kernels_seen))},
                         "kernel_ms_per_step": _stats(scan_ms), "host_ms_between_step_ends": _stats(step_gaps),
                         "anchor_variant": max(set(timed_variants), key=timed_variants.count) if timed_variants else 0,
                         "walk_kernel_ms": round(walk_kernel_ms, 4),
                         "frac_scan_plus_walk": round(algo_bytes / ((kernel_ms + walk_kernel_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "frac_whole_step": round(algo_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                         "step_minus_kernels_us": round((dt / args.steps * 1e3 - kernel_ms - walk_kernel_ms) * 1e3, 1),
                         "batches_in_flight": depth, "one_at_a_time_ms_per_step": round(sync_ms_per_step, 4),
                         "one_at_a_time_step_minus_kernels_us": round((sync_ms_per_step - kernel_ms - walk_kernel_ms) * 1e3, 1),
                         "measured_stream_read_GBs": round(stream_read_gbs, 1),
                         "frac_of_measured_stream_read": round(achieved / stream_read_gbs, 4),
                         "sclk_mhz_start": clocks_start, "sclk_mhz_end": clocks_end},
            "cpu_baseline": cpu1, "cpu_baseline_all_cores": cpu_all,
        }
    # N > 1: behind the timed steps (whose exchange is torch.distributed's), the same steps with the C ABI's own exchange
    # (acmi_comm_*: RCCL bound by libacmi.so itself -- what a PHP host would call), timed, at every N; then BASELINE configs[3] for
    # a few steps.  Both behind a watchdog: rank 0 prints the measured line if either does not return (ACMI_BENCH_C_ABI=0: neither).
    if world > 1 and os.environ.get("ACMI_BENCH_C_ABI", "1") != "0":
        done = threading.Event()

        def watchdog():
            if not done.wait(180):
                if rank == 0:
                    out["ranks"].setdefault("c_abi_exchange", {"ok": False, "error": "did not return within 180 s"})
                    print(_result_line(out, {}, {}), flush=True)
                os._exit(0 if rank == 0 else 3)
        threading.Thread(target=watchdog, daemon=True).start()
        if backend == "nccl" and not args.no_c_abi_gather:
            try:
                c_abi = _c_abi_steps(torch, dist, api, batch, res, rank, world, dev, d_blob, wl.hay_len, nh, records_per_step, max(1, min(args.steps, 10)))
            except Exception as e:  # reported, not fatal: the timed figure stands on the torch.distributed exchange
                c_abi = {"ok": False, "error": str(e)[:300]}
        else:
            c_abi = {"ok": None, "skipped": f"backend {backend}: the C ABI's communicator is RCCL, one GPU per rank"}
        if rank == 0:
            out["ranks"]["c_abi_exchange"] = c_abi
        if args.workload == "headline" and not args.no_legs:
            del d_blob
            torch.cuda.empty_cache()
            try:
                cfg3 = _config3_sharded(torch, dist, api, batch, synth, rank, world, dev, backend)
            except Exception as e:
                cfg3 = {"check_ok": False, "error": str(e)[:300]}
            if rank == 0:
                out["config3_sharded"] = cfg3
            d_blob = None
        done.set()
    if rank == 0:
        legs = set(x for x in args.legs.split(",") if x)
        want_leg = lambda name: world == 1 and not args.no_legs and (not legs or name in legs)  # noqa: E731
        detail, leg_s = {}, {}
        cfg3_cpu = config3_cpu_start(synth) if (want_leg("config3") and args.workload == "headline" and not args.no_cpu) else None

        def run_leg(name, fn, *a, **kw):
            t_leg = time.perf_counter()
            detail[name] = fn(*a, **kw)
            leg_s[name] = round(time.perf_counter() - t_leg, 2)

        if world == 1 and not args.no_dense and (not legs or "config5_dense" in legs):
            run_leg("config5_dense", dense_leg, torch, api, batch, synth, dev)
        # secondary figures, measured outside the timed region on the same GPU, same process
        if want_leg("general_path") and args.workload == "headline":
            run_leg("general_path", general_leg, torch, api, batch, wl.patterns(), d_blob, d_off, nh, records_per_step)
        d_blob = None
        torch.cuda.empty_cache()
        if want_leg("short_haystacks"):
            run_leg("short_haystacks", short_haystacks_leg, torch, api, batch, synth, dev)
        if want_leg("replace"):
            run_leg("replace", replace_leg, torch, api, batch, synth, dev)
        if want_leg("keyword_set"):
            run_leg("keyword_set", keywords_leg, torch, api, batch, synth, dev)
        if want_leg("keyword_words"):
            run_leg("keyword_words", keywords_leg, torch, api, batch, synth, dev, words_only=True)
        if want_leg("short_patterns") and args.workload == "headline":
            run_leg("short_patterns", short_patterns_leg, torch, api, batch, synth, dev, ds)
        if want_leg("config3") and args.workload == "headline":
            run_leg("config3", config3_leg, torch, api, batch, synth, dev, cpu=cfg3_cpu)
        if want_leg("big_dictionary"):
            run_leg("big_dictionary", dictionary_leg, torch, api, batch, synth, dev)
        if want_leg("density_sweep"):
            run_leg("density_sweep", density_leg, torch, api, batch, synth, dev)
        if want_leg("choice_matrix") and (args.choice_matrix or "choice_matrix" in legs):
            run_leg("choice_matrix", choice_matrix_leg, full=args.choice_matrix > 1)
        if cpu_started is not None:
            t_leg = time.perf_counter()
            out["cpu_baseline"], out["cpu_baseline_all_cores"] = cpu_baseline_finish(ds, cpu_started)
            out["cpu_baseline"]["ran"] = "behind the timed steps, next to the untimed legs"
            leg_s["cpu_baseline_wait"] = round(time.perf_counter() - t_leg, 2)
        # (the legs that time the HOST side run with no CPU baseline thread next to them)
        if want_leg("host_entry") and args.workload == "headline":
            run_leg("host_entry", host_leg, api, synth, ds, tile_count)
        if want_leg("benchmark_php_sample") and args.workload == "headline":
            run_leg("benchmark_php_sample", benchmark_php_leg, api, synth)
        if args.phase_times:
            out["host_us_per_step"] = {k: round(v / args.steps * 1e6, 1) for k, v in phase.items()}
        _emit(out, detail, leg_s)
    api.ahocorasick_deinit(res)
    if gather:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
