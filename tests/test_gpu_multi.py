"""The N > 1 path on the HIP scan (BASELINE configs[3]): contiguous haystack shards, one exchange step.

* two ranks sharing this box's one GPU, every rank scanning its shard on the device, all-gatherv over gloo
  (RCCL refuses two ranks on one device) -- R1, G5, G2 and the 65,536-pattern G6 set;
* the C ABI's own communicator (acmi_comm_*, RCCL bound at run time): a world of one on any box, two ranks when
  the box has two GPUs, and the unpacking half on three hand-made payloads."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT
from gpu_util import resource_for
from oracle.pyoracle import Oracle
from php_aho_corasick_amd import _native as N, api, batch, synth

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, world):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "mp_worker.py"), mode]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)


def test_two_ranks_share_one_gpu_device_scan_gloo_gather():
    out = _launch("gloo", 2)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("gloo ")]
    assert len(lines) == 8 and all(l.split(":")[1].strip().startswith("ok") for l in lines), out.stdout


def test_two_ranks_rccl_through_the_c_abi():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL wants one GPU per rank; this box has one")
    out = _launch("rccl", 2)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]


def _device_array(ptr, shape, typestr):
    import torch
    return torch.as_tensor(batch._DevView(ptr, shape, typestr), device="cuda").cpu().numpy()


def test_comm_world_of_one_and_unpacking_of_three_payloads():
    import torch
    ds = synth.r2()
    res = resource_for(ds)
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, True)
    batch.bind_current_stream(res)
    d_blob = torch.from_numpy(ds.hay_blob).cuda()
    d_off = torch.from_numpy(ds.hay_off.astype(np.int64)).cuda()
    total = batch.match_device(res, d_blob, d_off, True)
    comm = C.c_void_p()
    N.check(N.lib.acmi_comm_create(res._h, 0, 1, None, C.byref(comm)), res._h)
    dp, dr, nr, nh = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
    N.check(N.lib.acmi_comm_allgather_result(comm, res._h, ds.n_haystacks, C.byref(dp), C.byref(dr), C.byref(nr), C.byref(nh)), res._h)
    assert nr.value == total == len(re_) and nh.value == ds.n_haystacks
    assert np.array_equal(_device_array(dp.value, (nh.value + 1,), "<i8").astype(np.uint64), pe)
    assert np.array_equal(_device_array(dr.value, (nr.value, 4), "<i4").astype(np.uint32), re_)

    # three shards scanned one after the other on this GPU, their payloads laid out as the exchange would leave them
    world = 3
    shards = []
    for r in range(world):
        lo, hi = batch.shard_range(ds.n_haystacks, r, world)
        off = (ds.hay_off[lo:hi + 1] - ds.hay_off[lo]).astype(np.int64)
        blob = ds.hay_blob[int(ds.hay_off[lo]):int(ds.hay_off[hi])]
        t = batch.match_device(res, torch.from_numpy(blob).cuda(), torch.from_numpy(off).cuda(), True)
        shards.append(batch.result_numpy(res, hi - lo, t))
    nh_cap = max(len(p) - 1 for p, _ in shards)
    rec_cap = max(len(r) for _, r in shards)
    stride = (nh_cap + 1) * 8 + rec_cap * 16
    pay = np.zeros((world, stride), np.uint8)
    sizes = np.zeros(2 * world, np.uint64)
    for r, (p, rec) in enumerate(shards):
        pay[r, :len(p) * 8] = p.astype(np.uint64).view(np.uint8)
        pay[r, (nh_cap + 1) * 8:(nh_cap + 1) * 8 + rec.size * 4] = rec.astype(np.uint32).view(np.uint8).reshape(-1)
        sizes[2 * r], sizes[2 * r + 1] = len(rec), len(p) - 1
    d_pay = torch.from_numpy(pay.reshape(-1)).cuda()
    rc = N.lib.acmi_comm_debug_compact(comm, res._h, world, C.c_void_p(d_pay.data_ptr()), stride, nh_cap, sizes.ctypes.data,
                                       C.byref(dp), C.byref(dr), C.byref(nr), C.byref(nh))
    N.check(rc, res._h)
    assert nr.value == len(re_) and nh.value == ds.n_haystacks
    assert np.array_equal(_device_array(dp.value, (nh.value + 1,), "<i8").astype(np.uint64), pe)
    assert np.array_equal(_device_array(dr.value, (nr.value, 4), "<i4").astype(np.uint32), re_)
    N.check(N.lib.acmi_comm_destroy(comm))
    api.ahocorasick_deinit(res)
