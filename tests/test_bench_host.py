"""bench.py's host-side helpers, no GPU: the result line stays under the 4 KB the driver parses however much the legs report, and
the one-core CPU baseline started on a thread (cpu_baseline_start / cpu_baseline_finish) returns what the blocking call returns."""
import json
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from php_aho_corasick_amd import synth  # noqa: E402


def test_result_line_stays_under_the_limit_whatever_the_legs_carry():
    out = {"metric": "haystack_GB_per_s", "value": 1.0, "unit": "GB/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 1.0,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "w" * 300}, "ranks": {"note": "n" * 300},
           "roofline": {"bound": "hbm", "achieved": 1.0, "peak": 8000.0, "unit": "GB/s", "frac": 0.1, "traffic": 1,
                        "kernel_ms_per_step": {"min": 1, "median": 1, "max": 1, "n": 20}, "sclk_mhz_start": {"card0": 1}},
           "cpu_baseline": {"value": 0.04, "unit": "GB/s", "cores": 1, "kind": "reference", "sample": "s" * 200},
           "cpu_baseline_all_cores": {"value": 0.6, "unit": "GB/s", "cores": 16, "kind": "reference", "sample": "s" * 200}}
    big_leg = {"cells": [{"frac_whole_call": 0.1 + 0.01 * k, "check_ok": True, "text": "x" * 500} for k in range(60)]}
    detail = {f"leg{k}": big_leg for k in range(14)}
    detail["choice_matrix"] = {"violations": [], "worst_ratio_automatic_over_best_forced": 1.2, "cells": [big_leg] * 30}
    detail["host_entry"] = {"batch_512MiB": {"GB_per_s_pcie_inclusive": 54.0, "check_ok": True}, "single_match_8KiB_us": 25.0}
    text = bench._result_line(out, detail, {k: 1.0 for k in detail})
    assert len(text) < bench.RESULT_LINE_LIMIT
    line = json.loads(text)
    assert line["legs"]["leg3"] == {"ok": True, "frac": 0.1}
    assert line["legs"]["choice_matrix"] == {"ok": True, "violations": 0, "worst_ratio": 1.2}
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line


def test_cpu_baseline_on_a_thread_is_the_blocking_call():
    ds = synth.bench_batch(2048, bench.HAY_LEN, seed=1, planted=True)
    started = bench.cpu_baseline_start(ds, seconds_target=0.4)
    one, allc = bench.cpu_baseline_finish(ds, started)
    assert one["cores"] == 1 and one["unit"] == "GB/s" and one["value"] > 0 and one["kind"] in ("reference", "port")
    assert "2048 records per pass" in one["sample"]
    assert allc["cores"] >= 1 and allc["value"] > 0
    one2, _ = bench.cpu_baseline(ds, seconds_target=0.2)
    assert one2["kind"] == one["kind"] and 0.3 < one2["value"] / one["value"] < 3.0
