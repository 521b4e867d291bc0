"""bench.py's own multi-rank path (BASELINE configs[3]; the driver's SCALE runs use exactly this code with backend nccl):
`python bench.py --gpus 2` spawns its two ranks itself; here they share this box's one GPU and the all-gatherv of the match
arrays travels over gloo (ACMI_BENCH_BACKEND=gloo -- RCCL refuses two ranks on one device).  Weak and strong scaling, the
headline workload and the 65,536-pattern one."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT


def _bench(*argv, backend="gloo", timeout=1500):
    env = dict(os.environ, ACMI_BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, (json.loads(lines[-1]) if lines else None)


def test_more_ranks_than_gpus_is_refused_not_shrunk():
    """`--gpus 64` on a box without 64 GPUs must fail loudly (backend nccl), never run a smaller job under that name."""
    out, line = _bench("--gpus", "64", "--no-cpu", "--no-legs", "--no-dense", backend="nccl", timeout=300)
    assert out.returncode != 0 and line is None
    assert "refusing" in out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_headline(scaling):
    out, line = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--haystacks", "16384", "--scaling", scaling, "--no-cpu", "--no-legs", "--no-dense")
    assert out.returncode == 0 and line is not None, out.stdout[-2000:] + out.stderr[-3000:]
    per_rank = 16384 if scaling == "weak" else 8192
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["check_ok"] is True
    assert line["ranks"]["launched"] == 2 and line["ranks"]["process_group"] == 2 and line["ranks"]["joined_a_collective"] == 2
    assert line["config"]["haystacks_per_gpu"] == per_rank
    assert line["records_per_step_per_gpu"] >= per_rank                       # one planted hit per haystack
    assert line["ranks"]["gathered_haystacks"] == 2 * per_rank               # the global prefix array covers both shards
    assert line["ranks"]["gathered_records"] == 2 * line["records_per_step_per_gpu"]
    assert line["roofline"]["kernel"] == "k_scan_anchor" and line["roofline"]["kernel_ms_per_step"]["n"] == 3


@pytest.mark.gpu
def test_bench_two_ranks_config3():
    """BASELINE configs[3]: the 65,536-pattern workload sharded over two ranks, every plant of the tile checked."""
    out, line = _bench("--gpus", "2", "--workload", "config3", "--steps", "2", "--warmup", "1", "--haystacks", "8192", "--no-cpu", "--no-legs", "--no-dense")
    assert out.returncode == 0 and line is not None, out.stdout[-2000:] + out.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["check_ok"] is True and line["config"]["patterns"] == 65536
    assert line["ranks"]["joined_a_collective"] == 2 and line["ranks"]["gathered_haystacks"] == 2 * 8192
    assert line["roofline"]["anchor_variant"] & 8                               # the big-set filter ran


@pytest.mark.gpu
def test_bench_two_ranks_c_abi_steps_and_config3_leg():
    """N > 1 readiness (BASELINE configs[3]): behind the timed steps the line carries the steps whose exchange is the C ABI's own
    communicator (timed under RCCL; with two gloo ranks on one GPU: named as skipped, RCCL wants a GPU per rank) and a short run of
    the 65,536-pattern workload sharded over the ranks, checked; the result line stays under 4 KB."""
    out, line = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--haystacks", "16384", "--no-cpu", "--no-dense")
    assert out.returncode == 0 and line is not None, out.stdout[-2000:] + out.stderr[-3000:]
    last = out.stdout.rstrip("\n").split("\n")[-1]
    assert last.startswith("{") and len(last) < 4096
    assert line["n_gpus"] == 2 and line["check_ok"] is True and line["ranks"]["joined_a_collective"] == 2
    x = line["ranks"]["c_abi_exchange"]
    assert x["ok"] is None and "RCCL" in x["skipped"]
    c3 = line["config3_sharded"]
    assert c3["check_ok"] is True and c3["records_per_step"] >= 2 * 65536 and c3["ms_per_step"] > 0 and "65,536" in c3["workload"]


@pytest.mark.gpu
def test_bench_one_rank_has_the_contract_fields():
    out, line = _bench("--steps", "3", "--warmup", "1", "--haystacks", "16384", "--no-legs", "--no-dense", backend="nccl")
    assert out.returncode == 0 and line is not None, out.stdout[-2000:] + out.stderr[-3000:]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line
    assert line["n_gpus"] == 1 and line["check_ok"] is True and line["cpu_baseline"]["kind"] in ("reference", "port")
    r = line["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["kernel_ms_per_step"]["min"] <= r["kernel_ms_per_step"]["median"] <= r["kernel_ms_per_step"]["max"]


@pytest.mark.gpu
def test_default_command_prints_a_result_line_the_driver_can_read():
    """`python bench.py` exactly as the driver runs it (every leg): the LAST stdout line is the result, one JSON object under
    4 KB carrying the contract keys with `roofline` and `cpu_baseline`; the legs in full are on an earlier, prefixed line and
    in gpurun_out/bench_legs.json.  (Round 4's 26 KB last line was not parsed by the driver.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = out.stdout.rstrip("\n").split("\n")
    last = lines[-1]
    assert len(last) < 4096 and last.startswith("{"), (len(last), last[:200])
    line = json.loads(last)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "check_ok", "matches_per_s", "legs"):
        assert key in line, key
    assert line["steps"] == 20 and line["warmup"] == 5 and line["check_ok"] is True
    assert 0 < line["roofline"]["frac"] < 1 and line["roofline"]["bound"] == "hbm" and line["cpu_baseline"]["value"] > 0
    assert line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline_all_cores"]["value"] > 0
    want = {"config5_dense", "general_path", "host_entry", "short_haystacks", "replace", "keyword_set", "keyword_words", "short_patterns",
            "config3", "big_dictionary", "density_sweep", "choice_matrix", "benchmark_php_sample"}
    assert want <= set(line["legs"]), want - set(line["legs"])
    assert all(v["ok"] is not False for k, v in line["legs"].items() if k != "choice_matrix"), line["legs"]
    assert all(set(v) <= {"ok", "frac", "violations", "worst_ratio", "pcie_inclusive_GB_per_s", "single_match_8KiB_us", "s_per_sample", "gpu_s_per_sample"} for v in line["legs"].values())
    detail = [l for l in lines[:-1] if l.startswith("BENCH_DETAIL ")]
    assert len(detail) == 1 and not [l for l in lines[:-1] if l.startswith("{")]
    full = json.loads(detail[0][len("BENCH_DETAIL "):])
    assert want <= set(full) and full["value"] == line["value"]
    assert json.load(open(os.path.join(ROOT, "gpurun_out", "bench_legs.json")))["value"] == line["value"]
