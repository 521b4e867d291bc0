"""The N > 1 path on the HIP scan (BASELINE configs[3]): contiguous haystack shards, one exchange step.

* two ranks sharing this box's one GPU, every rank scanning its shard on the device, all-gatherv over gloo
  (RCCL refuses two ranks on one device) -- R1, G5, G2 and the 65,536-pattern G6 set;
* the C ABI's own communicator (acmi_comm_*): a world of one on any box, two RCCL ranks when the box has two GPUs, and
  2..5 ranks of ONE process on this GPU through the process-local transport -- the size round, the packing kernel and the
  per-rank broadcast group of acmi_comm_allgather_result itself, its growth path and its error path."""
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


def _comm_stat(comm, key):
    v = C.c_double()
    assert N.lib.acmi_comm_get_stat(comm, key.encode(), C.byref(v)) == 0
    return v.value


def test_comm_world_of_one():
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
    assert _comm_stat(comm, "world") == 1 and _comm_stat(comm, "transport") == 0 and _comm_stat(comm, "bytes_sent") == 0
    N.check(N.lib.acmi_comm_destroy(comm))
    api.ahocorasick_deinit(res)


def _local_world(ds, world, find_all=True, wrong_rank=None, rounds=1, shards=None):
    """`world` ranks of ONE process on this GPU, each with its own handle and thread, meeting through the C ABI's
    process-local transport (acmi_comm_create_local): the same size round, packing kernel and per-rank broadcast
    group as over RCCL, with device-to-device copies in place of the collectives.  Returns per rank
    (status, prefix, records, stats) of the last round."""
    import threading
    import torch
    ress = [resource_for(ds) for _ in range(world)]
    streams = [torch.cuda.Stream() for _ in range(world)]
    handles = (N.Handle * world)(*[r._h for r in ress])
    for r in ress:  # the communicators live on the handles' device: touch it
        N.check(N.lib.acmi_finalize(r._h), r._h)
    d0 = torch.zeros(16, dtype=torch.uint8, device="cuda")
    o0 = torch.zeros(1, dtype=torch.int64, device="cuda")
    for r in ress:
        batch.match_device(r, d0, o0, True)
    comms = (C.c_void_p * world)()
    assert N.lib.acmi_comm_create_local(handles, world, comms) == 0
    out = [None] * world

    def rank_main(r):
        with torch.cuda.stream(streams[r]):
            batch.bind_current_stream(ress[r])
            for it in range(rounds):
                lo, hi = shards[it][r] if shards else batch.shard_range(ds.n_haystacks, r, world)
                off = (ds.hay_off[lo:hi + 1] - ds.hay_off[lo]).astype(np.int64)
                blob = ds.hay_blob[int(ds.hay_off[lo]):int(ds.hay_off[hi])]
                d_blob = torch.from_numpy(blob if blob.size else np.zeros(16, np.uint8)).cuda()
                d_off = torch.from_numpy(off).cuda()
                batch.match_device(ress[r], d_blob, d_off, find_all)
                dp, dr, nr, nh = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
                claim = hi - lo + (1 if wrong_rank == r else 0)
                rc = N.lib.acmi_comm_allgather_result(comms[r], ress[r]._h, claim, C.byref(dp), C.byref(dr), C.byref(nr), C.byref(nh))
                if rc < 0:
                    out[r] = (rc, None, None, N.lib.acmi_comm_last_error(comms[r]).decode())
                    continue
                prefix = _device_array(dp.value, (nh.value + 1,), "<i8").astype(np.uint64)
                rec = _device_array(dr.value, (nr.value, 4), "<i4").astype(np.uint32) if nr.value else np.zeros((0, 4), np.uint32)
                out[r] = (rc, prefix, rec, {k: _comm_stat(comms[r], k) for k in ("rounds", "bytes_sent", "bytes_received", "transport", "world")})

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not any(t.is_alive() for t in th), "a rank is stuck in the exchange"
    for c in comms:
        N.check(N.lib.acmi_comm_destroy(c))
    for r in ress:
        api.ahocorasick_deinit(r)
    return out


@pytest.mark.parametrize("name,world,find_all", [("R2", 3, True), ("R2", 2, False), ("G5", 4, True), ("R1", 5, True), ("G3", 2, True)])
def test_allgatherv_layout_with_several_ranks_on_one_gpu(name, world, find_all):
    """W > 1 through acmi_comm_allgather_result itself: every rank ends with the oracle's single-process arrays; the
    bytes a rank receives are exactly the other ranks' shards (size-proportional: nothing is padded to the largest)."""
    ds = synth.NAMED[name]()
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
    out = _local_world(ds, world, find_all)
    sizes = []
    for r in range(world):
        lo, hi = batch.shard_range(ds.n_haystacks, r, world)
        sizes.append((int(pe[hi] - pe[lo]) * 8 + (hi - lo) * 8))   # 8 bytes per record on the wire {pos, pattern}, 8 per haystack
    for r, (rc, prefix, rec, st) in enumerate(out):
        assert rc == 0, out[r]
        assert np.array_equal(prefix, pe) and np.array_equal(rec.reshape(-1, 4), re_), (name, r)
        assert st["transport"] == 2 and st["world"] == world
        assert st["bytes_sent"] == sizes[r] and st["bytes_received"] == sum(sizes) - sizes[r]
        assert st["rounds"] in (2, 3)   # sizes [+ allocation status when the global arrays grew] + the payload group


def test_allgatherv_grows_and_repeats():
    """Three exchanges on one communicator, the second with far more records than the first (the global arrays grow:
    one extra 32-byte status round, taken by every rank), the third smaller again (no extra round)."""
    ds = synth.r1()
    n = ds.n_haystacks
    shards = [[(0, 1), (1, 2)], [(0, n // 2), (n // 2, n)], [(0, 3), (3, 5)]]
    o = Oracle("port")
    o.add_patterns((ds.pat_blob, ds.pat_off))
    out = _local_world(ds, 2, True, rounds=3, shards=shards)
    off5 = ds.hay_off[:6]
    pe, re_ = o.match_batch(ds.hay_blob[:int(off5[-1])], off5, True)
    for rc, prefix, rec, st in out:
        assert rc == 0 and np.array_equal(prefix, pe) and np.array_equal(rec.reshape(-1, 4), re_)
        assert st["rounds"] == 2


def test_allgatherv_error_reaches_every_rank_instead_of_hanging():
    """One rank claims a haystack count it has no result for: it reports that in the size round, which it still joins, and
    ALL ranks return an error (the rank at fault ACMI_E_ARG, its peers ACMI_E_DEVICE) -- nobody waits inside a collective."""
    ds = synth.r2()
    out = _local_world(ds, 3, True, wrong_rank=1)
    assert [o[0] for o in out] == [N.E_DEVICE, N.E_ARG, N.E_DEVICE], out
    assert all("rank 1" in o[3] for o in out)
