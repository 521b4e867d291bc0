"""Device-resident batch entry and multi-GPU sharding.

torch is plumbing here: device memory for the haystack blob, the current HIP stream, and
``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm; "gloo" in CPU tests).

Sharding: haystacks are independent (keep=0, php_ahocorasick.c:745;
tests/test6.phpt), so a batch is cut into contiguous haystack ranges, one per rank; the
automaton is replicated (every rank builds it from the same pattern list); the only exchange
step is the all-gatherv that concatenates the per-rank match arrays in rank order, which is
the reference's global order because the shards are contiguous.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N
from .api import AhoResource


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous balanced split of n haystacks: ranks [0, n % world) get one extra."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def bind_current_stream(res: AhoResource) -> None:
    """Run the handle's launches on torch's current HIP stream."""
    import torch
    N.check(N.lib.acmi_set_stream(res._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)), res._h)


def match_device(res: AhoResource, d_blob, d_off, find_all: bool = True) -> int:
    """Scan a batch that already lives in HBM.  ``d_blob``: uint8 CUDA tensor, ``d_off``: int64 CUDA
    tensor of n+1 byte offsets.  Results stay on the device (see ``result_tensors``).  Returns the
    number of records."""
    assert d_blob.is_cuda and d_off.is_cuda and d_blob.is_contiguous() and d_off.is_contiguous()
    nh = d_off.numel() - 1
    total = C.c_uint64(0)
    N.check(N.lib.acmi_match_batch_device(res._h, C.c_void_p(d_blob.data_ptr()), d_blob.numel(),
                                          C.c_void_p(d_off.data_ptr()), nh, 1 if find_all else 0,
                                          C.byref(total)), res._h)
    return int(total.value)


def match_device_bounded(res: AhoResource, d_blob, d_off, find_all: bool = True):
    """A resident batch under "max_result_bytes" (acmi.h): yields (first haystack, haystacks, records) per segment -- after every
    yield the segment's result is the handle's current one (``result_view`` / ``result_numpy``: haystack ids count from the
    segment's first haystack) -- resuming behind the last complete haystack until the batch is done."""
    nh = d_off.numel() - 1
    first = 0
    while first < nh or nh == 0:
        total = C.c_uint64(0)
        rc = N.lib.acmi_match_batch_device(res._h, C.c_void_p(d_blob.data_ptr()), d_blob.numel(), C.c_void_p(d_off.data_ptr() + 8 * first),
                                           nh - first, 1 if find_all else 0, C.byref(total))
        if rc == N.E_PARTIAL:
            k = int(res.stat("resume_haystack"))
        else:
            N.check(rc, res._h)
            k = nh - first
        yield first, k, int(total.value)
        first += k
        if nh == 0:
            break


def match_device_begin(res: AhoResource, d_blob, d_off, find_all: bool = True) -> None:
    """Enqueue the scan of a resident batch and return immediately (see ``match_device_end``).  Up to two batches may
    be in flight per resource: the second one's kernels queue behind the first one's (acmi.h)."""
    assert d_blob.is_cuda and d_off.is_cuda and d_blob.is_contiguous() and d_off.is_contiguous()
    N.check(N.lib.acmi_match_batch_device_begin(res._h, C.c_void_p(d_blob.data_ptr()), d_blob.numel(),
                                                C.c_void_p(d_off.data_ptr()), d_off.numel() - 1,
                                                1 if find_all else 0), res._h)


def match_device_begin_uniform(res: AhoResource, d_blob, haystack_bytes: int, n_haystacks: int, find_all: bool = True) -> None:
    """``match_device_begin`` for a batch of ``n_haystacks`` haystacks of ``haystack_bytes`` each, back to back in ``d_blob``: no
    offsets tensor; the library writes the work-item table in closed form (acmi.h)."""
    assert d_blob.is_cuda and d_blob.is_contiguous() and d_blob.numel() >= haystack_bytes * n_haystacks
    N.check(N.lib.acmi_match_batch_device_begin_uniform(res._h, C.c_void_p(d_blob.data_ptr()), haystack_bytes, n_haystacks,
                                                        1 if find_all else 0), res._h)


def match_device_end(res: AhoResource) -> int:
    """Wait for the OLDEST batch enqueued by ``match_device_begin``; returns its record count.  Its result
    (``result_view`` / ``result_numpy`` / ...) stays valid until the second begin after its own."""
    total = C.c_uint64(0)
    N.check(N.lib.acmi_match_batch_device_end(res._h, C.byref(total)), res._h)
    return int(total.value)


def result_tensors(res: AhoResource, n_haystacks: int, total: int, device=None):
    """Copy the last result into torch-owned CUDA tensors: (prefix int64[nh+1], records int32[n,4])."""
    import torch
    device = device or torch.device("cuda", torch.cuda.current_device())
    prefix = torch.empty(n_haystacks + 1, dtype=torch.int64, device=device)
    records = torch.empty((max(total, 1), 4), dtype=torch.int32, device=device)
    N.check(N.lib.acmi_result_to_device(res._h, C.c_void_p(prefix.data_ptr()), C.c_void_p(records.data_ptr())),
            res._h)
    return prefix, records[:total]


def result_numpy(res: AhoResource, n_haystacks: int, total: int):
    prefix = np.zeros(n_haystacks + 1, dtype=np.uint64)
    records = np.zeros((max(total, 1), 4), dtype=np.uint32)
    N.check(N.lib.acmi_result_to_host(res._h, prefix.ctypes.data, records.ctypes.data), res._h)
    return prefix, records[:total]


class _DevView:
    """Zero-copy handle on library-owned device memory for ``torch.as_tensor``."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False), "version": 2}


def result_view(res: AhoResource, total: int):
    """The last result's record array as an int32 [total, 4] CUDA tensor WITHOUT a copy.  It aliases
    the handle's buffer: valid until the next match call on ``res``."""
    import torch
    d_prefix, d_rec, n = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
    N.check(N.lib.acmi_result_device(res._h, C.byref(d_prefix), C.byref(d_rec), C.byref(n)), res._h)
    if total == 0:
        return torch.zeros((0, 4), dtype=torch.int32, device="cuda")
    return torch.as_tensor(_DevView(d_rec.value, (total, 4), "<i4"), device="cuda")


def replace_device(res: AhoResource, d_blob, d_off, mode: int = 1):
    """Search and replace over a batch resident in HBM (acmi_replace_batch_device: multifast_replace + flush,
    reference src/multifast/replace.c:469-568).  Returns (out_blob uint8[total], out_off int64[n+1]) as CUDA tensors
    that ALIAS the handle's buffers: valid until the next call on ``res``."""
    import torch
    assert d_blob.is_cuda and d_off.is_cuda and d_blob.is_contiguous() and d_off.is_contiguous()
    nh = d_off.numel() - 1
    d_out, d_oo, total = C.c_void_p(), C.c_void_p(), C.c_uint64(0)
    N.check(N.lib.acmi_replace_batch_device(res._h, C.c_void_p(d_blob.data_ptr()), d_blob.numel(), C.c_void_p(d_off.data_ptr()), nh,
                                            mode, C.byref(d_out), C.byref(d_oo), C.byref(total)), res._h)
    off = torch.as_tensor(_DevView(d_oo.value, (nh + 1,), "<i8"), device="cuda")
    if total.value == 0:
        return torch.zeros(0, dtype=torch.uint8, device="cuda"), off
    return torch.as_tensor(_DevView(d_out.value, (total.value,), "|u1"), device="cuda"), off


class RecordGatherer:
    """All-gatherv of match arrays with ONE collective per batch and no blocking size exchange.

    RCCL has no v-variant of all-gather.  Every rank sends a fixed-capacity buffer
    ``[n_records, n_haystacks | records ...]``; the capacity is agreed implicitly (all ranks start
    equal and grow it identically, from the headers everybody receives) and only a batch that
    exceeds it costs a second, larger collective.  Per-haystack counts are not shipped: records
    carry global haystack ids in ascending order, so the prefix array is a local ``bincount``.

    ``start`` only enqueues (payload copy on the current stream, then the collective on the
    backend's stream), so the caller can scan its next batch while this one travels over xGMI;
    ``finish`` waits, reads the headers and returns ``(prefix int64[n_global+1], records int32[n,4])``.
    Buffers rotate over ``depth`` slots (default 2): at most ``depth`` batches between start and finish.
    With the scan kernel holding every CU's LDS, a collective started during step t only gets CUs when that
    step's scan lets go, so a caller that must not block in ``finish`` keeps three in flight.
    """

    def __init__(self, group=None, device=None, capacity: int = 1 << 16, depth: int = 2):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.device = device
        self.cap = int(capacity)
        self._bufs = {}
        self._hdr, self._base, self._base_val = {}, {}, {}
        self._parity = 0
        self.depth = max(2, int(depth))

    def _buffers(self, parity: int, cap: int):
        import torch
        key = (parity, cap)
        if key not in self._bufs:
            for k in [k for k in self._bufs if k[0] == parity]:
                del self._bufs[k]
            n = 2 + 2 * cap  # int64 words: header + 16 B per record
            self._bufs[key] = (torch.zeros(n, dtype=torch.int64, device=self.device),
                               torch.zeros(self.world * n, dtype=torch.int64, device=self.device))
        return self._bufs[key]

    def _launch(self, send, recv, async_op):
        import torch.distributed as dist
        # one flat output buffer: the list form makes the backend gather into a temporary and copy out
        return dist.all_gather_into_tensor(recv, send, group=self.group, async_op=async_op)

    def start(self, records_local, n_haystacks_local: int, hay_base: int):
        import torch
        if self.device is None:
            self.device = records_local.device
        n = int(records_local.shape[0])
        cap = self.cap
        parity, self._parity = self._parity, (self._parity + 1) % self.depth
        send, recv = self._buffers(parity, cap)
        # header from a per-parity pinned tensor (one async copy), payload with the haystack base added in
        # the same kernel that copies it: start() runs between two scans, so its host time is exposed
        cuda = torch.device(self.device).type == "cuda"
        if parity not in self._hdr:
            self._hdr[parity] = torch.zeros(2, dtype=torch.int64, pin_memory=cuda)
            self._base[parity] = torch.zeros((1, 4), dtype=torch.int32, device=self.device)
        hdr = self._hdr[parity]
        hdr[0], hdr[1] = n, n_haystacks_local
        send[:2].copy_(hdr, non_blocking=True)
        m = min(n, cap)
        if m:
            base = self._base[parity]
            if self._base_val.get(parity) != hay_base:
                base[0, 0] = hay_base
                self._base_val[parity] = hay_base
            pay = send[2:2 + 2 * m].view(torch.int32).view(m, 4)
            torch.add(records_local[:m], base, out=pay)
        spill = None
        if n > cap:  # rare: keep everything for the second, larger collective
            spill = records_local.clone()
            spill[:, 0] += hay_base
        work = self._launch(send, recv, True)
        return {"work": work, "cap": cap, "parity": parity, "n": n, "nh": n_haystacks_local, "spill": spill}

    def finish(self, st):
        """Unpack on a side stream (CUDA) so that a scan already running on the main stream is not
        delayed; the main stream is made to wait for the side stream before returning."""
        import torch
        if self.device is not None and torch.device(self.device).type == "cuda":
            if getattr(self, "_side", None) is None:
                # high priority: its own hardware queue (HIP maps streams onto a few; behind the scan kernel's
                # queue the unpack would wait for the whole scan)
                self._side = torch.cuda.Stream(device=self.device, priority=-1)
            main = torch.cuda.current_stream(self.device)
            # no side.wait_stream(main): the collective already ordered itself after the payload copy,
            # and waiting for `main` here would serialise behind the scan that is running on it
            with torch.cuda.stream(self._side):
                out = self._finish(st)
            main.wait_stream(self._side)
            return out
        return self._finish(st)

    def _finish(self, st):
        import torch
        st["work"].wait()
        send, recv = self._buffers(st["parity"], st["cap"])
        rows = recv.view(self.world, -1)
        hdr = rows[:, :2].cpu()
        need = int(hdr[:, 0].max())
        if need > st["cap"]:  # every rank sees the same headers and takes the same branch
            cap = 1 << max(need - 1, 1).bit_length()
            self.cap = max(self.cap, cap)
            send2 = torch.zeros(2 + 2 * cap, dtype=torch.int64, device=self.device)
            recv2 = torch.zeros(self.world * (2 + 2 * cap), dtype=torch.int64, device=self.device)
            send2[:2] = send[:2]
            n = st["n"]
            if n:
                src = st["spill"] if st["spill"] is not None else send[2:2 + 2 * n].view(torch.int32).view(n, 4)
                send2[2:2 + 2 * n].view(torch.int32).view(n, 4).copy_(src)
            self._launch(send2, recv2, False)
            rows = recv2.view(self.world, -1)
        parts = [rows[r, 2:2 + 2 * int(hdr[r, 0])].view(torch.int32).view(-1, 4) for r in range(self.world)]
        records = torch.cat(parts) if parts else torch.zeros((0, 4), dtype=torch.int32, device=self.device)
        nh_global = int(hdr[:, 1].sum())
        # records are sorted by global haystack id, so prefix[h] = first record with id >= h: one binary-search
        # kernel.  (bincount + cumsum use LDS, which the scan kernel holds on every CU: queued on the side
        # stream they only ran once a scan let go, and the next finish() blocked behind them.)
        if records.shape[0]:
            prefix = torch.searchsorted(records[:, 0].contiguous(),
                                        torch.arange(nh_global + 1, dtype=torch.int32, device=self.device))
        else:
            prefix = torch.zeros(nh_global + 1, dtype=torch.int64, device=self.device)
        return prefix, records


def allgatherv_start(counts_local, records_local, hay_base: int, group=None):
    """First half of the all-gatherv of match arrays (see ``allgatherv_records``): exchanges sizes
    (blocking, 16 bytes per rank) and LAUNCHES the padded payload all_gathers asynchronously, so the
    caller can start scanning its next batch while RCCL moves this one over xGMI."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = records_local.device
    rec = records_local.clone()
    if rec.numel():
        rec[:, 0] += hay_base
    sizes = torch.tensor([rec.shape[0], counts_local.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = torch.stack(all_sizes).cpu()
    max_rec, max_cnt = int(all_sizes[:, 0].max()), int(all_sizes[:, 1].max())
    rec_pad = torch.zeros((max(max_rec, 1), 4), dtype=torch.int32, device=dev)
    rec_pad[:rec.shape[0]] = rec
    cnt_pad = torch.zeros(max(max_cnt, 1), dtype=torch.int64, device=dev)
    cnt_pad[:counts_local.numel()] = counts_local
    rec_all = [torch.empty_like(rec_pad) for _ in range(world)]
    cnt_all = [torch.empty_like(cnt_pad) for _ in range(world)]
    w1 = dist.all_gather(rec_all, rec_pad, group=group, async_op=True)
    w2 = dist.all_gather(cnt_all, cnt_pad, group=group, async_op=True)
    return all_sizes, rec_all, cnt_all, (w1, w2), (rec_pad, cnt_pad)


def allgatherv_finish(state):
    """Second half: wait for the payloads, drop the padding, concatenate in rank order."""
    import torch
    all_sizes, rec_all, cnt_all, works, _keep = state
    for w in works:
        w.wait()
    world = len(rec_all)
    records = torch.cat([rec_all[r][:int(all_sizes[r, 0])] for r in range(world)])
    counts = torch.cat([cnt_all[r][:int(all_sizes[r, 1])] for r in range(world)])
    prefix = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=records.device)
    prefix[1:] = torch.cumsum(counts, 0)
    return prefix, records


def allgatherv_records(counts_local, records_local, hay_base: int, group=None):
    """All-gatherv of match arrays: every rank ends with the concatenation, in rank order, of all
    ranks' per-haystack counts and records (haystack ids made global with ``hay_base``).

    RCCL has no native v-variant; sizes are exchanged first (fixed-size all_gather), then the
    payloads are padded to the largest shard and exchanged with one all_gather each.  Match arrays
    are tiny next to the haystacks on sparse workloads (16 B per record), so padding costs nothing
    that matters; the dense adversarial case would rather gather to the caller's rank only.

    ``counts_local``: int64[n_local] records per local haystack; ``records_local``: int32[n,4].
    Returns (prefix int64[n_global+1], records int32[total,4]) on the same device as the inputs.
    """
    return allgatherv_finish(allgatherv_start(counts_local, records_local, hay_base, group))


def sharded_match(res: AhoResource, d_blob_local, d_off_local, hay_base: int, find_all: bool = True, group=None,
                  gatherer: RecordGatherer | None = None):
    """One rank's part of a sharded batch: scan the local contiguous haystack range on this GPU,
    then all-gatherv.  Returns the GLOBAL (prefix, records) on every rank."""
    nh = d_off_local.numel() - 1
    total = match_device(res, d_blob_local, d_off_local, find_all)
    g = gatherer or RecordGatherer(group, d_blob_local.device)
    return g.finish(g.start(result_view(res, total), nh, hay_base))
