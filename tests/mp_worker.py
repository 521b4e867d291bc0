"""Worker of tests/test_gpu_multi.py: one rank of a sharded batch (launched by torch.distributed.run).

    mode "gloo":  ranks share cuda:0, torch.distributed gloo carries the all-gatherv (batch.sharded_match)
    mode "rccl":  one GPU per rank, the C ABI's own communicator (acmi_comm_*, RCCL bound by libacmi.so);
                  the id travels over a gloo broadcast
Each rank scans its contiguous haystack range ON THE DEVICE; the gathered global result must equal the oracle's
single-process result, record for record."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

from oracle.pyoracle import Oracle
from php_aho_corasick_amd import _native as N, api, batch, synth

mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev_index = 0 if mode == "gloo" else rank % torch.cuda.device_count()
torch.cuda.set_device(dev_index)
dev = torch.device("cuda", dev_index)

cases = [("R1", synth.r1()), ("G5", synth.g5()), ("G2", synth.g2()), ("G6", synth.g6())]
bad = 0
for name, ds in cases:
    res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
    batch.bind_current_stream(res)
    lo, hi = batch.shard_range(ds.n_haystacks, rank, world)
    off = (ds.hay_off[lo:hi + 1] - ds.hay_off[lo]).astype(np.int64)
    blob = ds.hay_blob[int(ds.hay_off[lo]):int(ds.hay_off[hi])]
    d_blob = torch.from_numpy(blob if blob.size else np.zeros(16, np.uint8)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    comm = C.c_void_p()
    if mode == "rccl":
        ident = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_char * 128)()
            N.check(N.lib.acmi_comm_get_id(buf))
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        dist.broadcast(ident, 0)
        raw = (C.c_char * 128).from_buffer_copy(ident.numpy().tobytes())
        batch.match_device(res, d_blob[:0] if False else d_blob, d_off[:1], True)  # touches the device: the communicator lives there
        N.check(N.lib.acmi_comm_create(res._h, rank, world, raw, C.byref(comm)), res._h)
    for find_all in (True, False):
        if mode == "gloo":
            prefix, records = batch.sharded_match(res, d_blob, d_off, lo, find_all)
            prefix, records = prefix.cpu().numpy().astype(np.uint64), records.cpu().numpy().astype(np.uint32)
        else:
            batch.match_device(res, d_blob, d_off, find_all)
            dp, dr, nr, nhg = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
            N.check(N.lib.acmi_comm_allgather_result(comm, res._h, hi - lo, C.byref(dp), C.byref(dr), C.byref(nr), C.byref(nhg)), res._h)
            prefix = torch.as_tensor(batch._DevView(dp.value, (nhg.value + 1,), "<i8"), device=dev).cpu().numpy().astype(np.uint64)
            records = (torch.as_tensor(batch._DevView(dr.value, (nr.value, 4), "<i4"), device=dev).cpu().numpy().astype(np.uint32)
                       if nr.value else np.zeros((0, 4), np.uint32))
        if rank == 0:
            o = Oracle("port")
            o.add_patterns((ds.pat_blob, ds.pat_off))
            pe, re_ = o.match_batch(ds.hay_blob, ds.hay_off, find_all)
            ok = np.array_equal(prefix, pe) and np.array_equal(records.reshape(-1, 4), re_)
            print(f"{mode} {name} find_all={find_all}: {'ok' if ok else 'MISMATCH'} ({len(re_)} records, kernel {int(res.stat('kernel_used'))})", flush=True)
            bad += 0 if ok else 1
    if comm:
        N.lib.acmi_comm_destroy(comm)
    api.ahocorasick_deinit(res)
dist.barrier()
dist.destroy_process_group()
sys.exit(1 if bad else 0)
