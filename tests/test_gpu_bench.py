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
def test_bench_one_rank_has_the_contract_fields():
    out, line = _bench("--steps", "3", "--warmup", "1", "--haystacks", "16384", "--no-legs", "--no-dense", backend="nccl")
    assert out.returncode == 0 and line is not None, out.stdout[-2000:] + out.stderr[-3000:]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line
    assert line["n_gpus"] == 1 and line["check_ok"] is True and line["cpu_baseline"]["kind"] in ("reference", "port")
    r = line["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] < 1 and r["kernel_ms_per_step"]["min"] <= r["kernel_ms_per_step"]["median"] <= r["kernel_ms_per_step"]["max"]
