// engine_comm.cpp -- multi-GPU: contiguous haystack shards, one exchange step (acmi_comm_*)
#include "engine.hpp"

// ---- multi-GPU: contiguous haystack shards, one exchange step ------------------------------------------
// RCCL is bound at run time (dlopen): libacmi.so has no link-time dependency on it, a host without RCCL simply gets
// ACMI_E_NO_DEVICE from acmi_comm_*.  torch.distributed's "nccl" backend is the same library: a process that has
// torch loaded reuses its copy.
//
// The exchange is SIZE-PROPORTIONAL: after a 32-byte header per rank (records, haystacks, status) every rank knows every
// shard's size, writes its own shard -- haystack ids and prefix words already global -- into ITS slice of the global
// arrays, and one group of per-rank ncclBroadcast calls (root r, exactly r's bytes, in place for the root) fills in the
// other slices: no padding to the largest shard, no staging copy, no unpacking kernel.  A rank that cannot take part (no
// result, wrong haystack count, out of memory) says so IN THE HEADER and still joins the header round, so that every rank
// returns an error together instead of leaving its peers inside a collective.
//
// The collectives sit behind a small transport interface with two implementations: RCCL (one process or thread per GPU),
// and a process-local fabric (acmi_comm_create_local: W communicators of one process, each used by its own thread, copies
// instead of collectives) -- the same packing / offset logic runs on a box with ONE GPU, which is how the W > 1 layout is
// tested where RCCL refuses several ranks per device.
namespace {

struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, acmi_comm_id, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = std::getenv("ACMI_RCCL_LIBRARY");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {  // a copy that is already in the process first (torch's)
            if (n && (r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) break;
        }
        for (const char *n : names) {
            if (r.lib) break;
            if (n) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!r.lib) { r.error = "librccl.so not found (set ACMI_RCCL_LIBRARY)"; return; }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.lib, "ncclAllGather"));
        r.Broadcast = reinterpret_cast<decltype(r.Broadcast)>(dlsym(r.lib, "ncclBroadcast"));
        r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.lib, "ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.lib, "ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(r.lib, "ncclCommCount"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.Broadcast || !r.GroupStart || !r.GroupEnd || !r.CommDestroy)
            r.error = "librccl.so lacks a needed symbol";
    });
    return r;
}

constexpr int kNcclUint8 = 1;  // ncclDataType_t (rccl.h:460)

// One broadcast of a group: `bytes` from rank `root`'s `send` into every rank's `recv` (the root passes send == recv).
struct BcastOp {
    const void *send;
    void *recv;
    size_t bytes;
    int root;
};

// The process-local fabric: W communicators of one process meet here.  Every collective is a rendezvous (all ranks
// post their buffers, wait for each other, copy what they receive on their own stream, synchronise it, meet again):
// as blocking as can be, which is all a test transport and a single-process multi-GPU host need.
struct LocalFabric {
    std::mutex m;
    std::condition_variable cv;
    int world = 0, refs = 0;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void *> send;                 // all-gather: per rank
    std::vector<std::vector<BcastOp>> ops;          // broadcast group: per rank
    std::vector<hipEvent_t> ready;                  // per rank: its send data is complete on its stream
    void barrier() {
        std::unique_lock<std::mutex> g(m);
        const uint64_t gen = generation;
        if (++arrived == world) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(g, [&] { return generation != gen; });
    }
};

}  // namespace

struct acmi_comm {
    void *comm = nullptr;           // ncclComm_t (RCCL transport)
    LocalFabric *fabric = nullptr;  // process-local transport
    int rank = 0, world = 1, dev = 0;
    int backend_ranks = 1;          // ranks the backend itself reports (ncclCommCount)
    DevBuf d_hdr, d_prefix, d_records, d_wire;   // (d_wire: every rank's records as they travel, 8 bytes each {pos, pattern})
    uint64_t *h_hdr = nullptr;      // pinned: [mine 4 | all world x 4] words
    // capacities every rank holds, kept identical on all ranks (they grow by the same rule from the same headers)
    uint64_t cap_rec = 0, cap_hay = 0;
    bool cap_ok = true;             // this rank's buffers really have that capacity
    int last_root = -2;             // root of the last exchange (-1: all-gather; -2: none yet)
    uint64_t total_rec = 0, total_hay = 0;
    uint64_t bytes_sent = 0, bytes_received = 0, rounds = 0;  // of the last exchange (statistics)
    std::string last_error;
};

namespace {

constexpr uint64_t kHdrWords = 4;  // {records, haystacks, status, reserved}
enum : uint64_t { kCommOk = 0, kCommNoResult = 1, kCommNoMemory = 2, kCommDevice = 3 };

int comm_fail(acmi_handle *h, acmi_comm *c, int code, const std::string &text) {
    c->last_error = text;
    if (h && live(h)) h->last_error = text;
    return code;
}

// all-gather of `bytes` per rank (device buffers), enqueued on `st`
int comm_allgather(acmi_comm *c, const void *send, void *recv, size_t bytes, hipStream_t st) {
    if (c->world == 1) {
        if (send != recv && hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return ACMI_E_DEVICE;
        return ACMI_OK;
    }
    if (c->fabric) {
        LocalFabric &F = *c->fabric;
        // (an error on this rank travels THROUGH both barriers: a rank that left early would leave its peers waiting for ever)
        int rc = hipEventRecord(F.ready[size_t(c->rank)], st) == hipSuccess ? ACMI_OK : ACMI_E_DEVICE;
        F.send[size_t(c->rank)] = send;
        F.barrier();
        for (int r = 0; r < c->world && rc == ACMI_OK; ++r) {
            if (hipStreamWaitEvent(st, F.ready[size_t(r)], 0) != hipSuccess ||
                hipMemcpyAsync(static_cast<uint8_t *>(recv) + size_t(r) * bytes, F.send[size_t(r)], bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                rc = ACMI_E_DEVICE;
        }
        if (hipStreamSynchronize(st) != hipSuccess) rc = ACMI_E_DEVICE;
        F.barrier();  // nobody reuses its send buffer before everybody has copied it
        return rc;
    }
    Rccl &R = rccl();
    const int e = R.AllGather(send, recv, bytes, kNcclUint8, c->comm, st);
    if (e != 0) { c->last_error = std::string("ncclAllGather: ") + (R.GetErrorString ? R.GetErrorString(e) : "error"); return ACMI_E_DEVICE; }
    return ACMI_OK;
}

// one group of broadcasts; every rank passes the same list (same order, sizes and roots).  dest >= 0: only rank `dest` receives
// (a gather: a send from every root to dest, ncclSend / ncclRecv in one group)
int comm_bcast_group(acmi_comm *c, const std::vector<BcastOp> &ops, hipStream_t st, int dest = -1) {
    if (c->world == 1 || ops.empty()) return ACMI_OK;
    if (c->fabric) {
        LocalFabric &F = *c->fabric;
        int rc = hipEventRecord(F.ready[size_t(c->rank)], st) == hipSuccess ? ACMI_OK : ACMI_E_DEVICE;
        F.ops[size_t(c->rank)] = ops;
        F.barrier();
        for (size_t i = 0; i < ops.size() && rc == ACMI_OK; ++i) {
            const BcastOp &mine = ops[i];
            if (mine.root == c->rank || mine.bytes == 0 || (dest >= 0 && dest != c->rank)) continue;
            const BcastOp &theirs = F.ops[size_t(mine.root)][i];
            if (hipStreamWaitEvent(st, F.ready[size_t(mine.root)], 0) != hipSuccess ||
                hipMemcpyAsync(mine.recv, theirs.send, mine.bytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
                rc = ACMI_E_DEVICE;
        }
        if (hipStreamSynchronize(st) != hipSuccess) rc = ACMI_E_DEVICE;
        F.barrier();
        return rc;
    }
    Rccl &R = rccl();
    if (dest >= 0 && (!R.Send || !R.Recv)) { c->last_error = "librccl.so lacks ncclSend / ncclRecv"; return ACMI_E_NO_DEVICE; }
    int e = R.GroupStart();
    for (size_t i = 0; i < ops.size() && e == 0; ++i) {
        if (!ops[i].bytes) continue;
        if (dest < 0) e = R.Broadcast(ops[i].send, ops[i].recv, ops[i].bytes, kNcclUint8, ops[i].root, c->comm, st);
        else if (ops[i].root == dest) continue;
        else if (c->rank == ops[i].root) e = R.Send(ops[i].send, ops[i].bytes, kNcclUint8, dest, c->comm, st);
        else if (c->rank == dest) e = R.Recv(ops[i].recv, ops[i].bytes, kNcclUint8, ops[i].root, c->comm, st);
    }
    const int e2 = R.GroupEnd();
    if (e == 0) e = e2;
    if (e != 0) { c->last_error = std::string("ncclBroadcast: ") + (R.GetErrorString ? R.GetErrorString(e) : "error"); return ACMI_E_DEVICE; }
    return ACMI_OK;
}

// header round: every rank's {records, haystacks, status} -> c->h_hdr[4 ..]; the exchange's one host wait
int comm_headers(acmi_comm *c, uint64_t n_rec, uint64_t n_hay, uint64_t status, hipStream_t st) {
    const int W = c->world;
    uint64_t *mine = c->h_hdr, *all = c->h_hdr + kHdrWords;
    mine[0] = n_rec; mine[1] = n_hay; mine[2] = status; mine[3] = 0;
    uint64_t *d_mine = c->d_hdr.as<uint64_t>(), *d_all = d_mine + kHdrWords;
    if (hipMemcpyAsync(d_mine, mine, kHdrWords * 8, hipMemcpyHostToDevice, st) != hipSuccess) return ACMI_E_DEVICE;
    int rc = comm_allgather(c, d_mine, d_all, kHdrWords * 8, st);
    if (rc < 0) return rc;
    if (hipMemcpyAsync(all, d_all, size_t(W) * kHdrWords * 8, hipMemcpyDeviceToHost, st) != hipSuccess) return ACMI_E_DEVICE;
    if (hipStreamSynchronize(st) != hipSuccess) return ACMI_E_DEVICE;
    ++c->rounds;
    return ACMI_OK;
}

acmi_comm *comm_new(acmi_handle *h, int rank, int world) {
    acmi_comm *c = new acmi_comm();
    c->rank = rank;
    c->world = world;
    c->dev = h->dev_id;
    c->backend_ranks = world;
    // the header buffers exist from the start: the header round itself must not be able to fail on an allocation
    if (hipHostMalloc(reinterpret_cast<void **>(&c->h_hdr), size_t(world + 1) * kHdrWords * 8, hipHostMallocDefault) != hipSuccess ||
        c->d_hdr.ensure(size_t(world + 1) * kHdrWords * 8) != hipSuccess) {
        (void)hipGetLastError();
        if (c->h_hdr) (void)hipHostFree(c->h_hdr);
        c->d_hdr.release();
        delete c;
        return nullptr;
    }
    return c;
}

}  // namespace

extern "C" {

int acmi_comm_get_id(acmi_comm_id *id) {
    return guarded(nullptr, [&]() -> int {
        if (!id) return ACMI_E_ARG;
        Rccl &R = rccl();
        if (!R.error.empty()) return ACMI_E_NO_DEVICE;
        return R.GetUniqueId(id) == 0 ? ACMI_OK : ACMI_E_DEVICE;
    });
}

int acmi_comm_create(acmi_handle *h, int rank, int world, const acmi_comm_id *id, acmi_comm **out) {
    return guarded(h, [&]() -> int {
        int rc = check_ready(h);  // the communicator lives on the handle's device
        if (rc < 0) return rc;
        if (!out || world < 1 || world > int(kMaxRanks) || rank < 0 || rank >= world || (world > 1 && !id)) return ACMI_E_ARG;
        acmi_comm *c = comm_new(h, rank, world);
        if (!c) return ACMI_E_NOMEM;
        if (world > 1) {
            Rccl &R = rccl();
            if (!R.error.empty()) { h->last_error = R.error; acmi_comm_destroy(c); return ACMI_E_NO_DEVICE; }
            const int e = R.CommInitRank(&c->comm, world, *id, rank);
            if (e != 0) {
                h->last_error = std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(e) : "error");
                c->comm = nullptr;
                acmi_comm_destroy(c);
                return ACMI_E_DEVICE;
            }
            int n = 0;
            if (R.CommCount && R.CommCount(c->comm, &n) == 0) c->backend_ranks = n;
        }
        *out = c;
        return ACMI_OK;
    });
}

int acmi_comm_create_local(acmi_handle *const *handles, int world, acmi_comm **out) {
    return guarded(nullptr, [&]() -> int {
        if (!handles || !out || world < 1 || world > int(kMaxRanks)) return ACMI_E_ARG;
        for (int r = 0; r < world; ++r) {
            const int rc = check_ready(handles[r]);
            if (rc < 0) return rc;
        }
        LocalFabric *F = new LocalFabric();
        F->world = world;
        F->send.assign(size_t(world), nullptr);
        F->ops.resize(size_t(world));
        F->ready.assign(size_t(world), nullptr);
        int made = 0;
        for (int r = 0; r < world; ++r) {
            acmi_comm *c = nullptr;
            if (hipSetDevice(handles[r]->dev_id) == hipSuccess && hipEventCreateWithFlags(&F->ready[size_t(r)], hipEventDisableTiming) == hipSuccess)
                c = comm_new(handles[r], r, world);
            if (!c) {
                for (int k = 0; k < made; ++k) { out[k]->fabric = nullptr; acmi_comm_destroy(out[k]); out[k] = nullptr; }
                for (hipEvent_t e : F->ready) if (e) (void)hipEventDestroy(e);
                delete F;
                return ACMI_E_NOMEM;
            }
            c->fabric = F;
            ++F->refs;
            out[r] = c;
            ++made;
        }
        return ACMI_OK;
    });
}

int acmi_comm_destroy(acmi_comm *c) {
    return guarded(nullptr, [&]() -> int {
        if (!c) return ACMI_E_INVALID;
        (void)hipSetDevice(c->dev);
        if (c->comm) rccl().CommDestroy(c->comm);
        if (c->fabric) {
            bool last;
            {
                std::lock_guard<std::mutex> g(c->fabric->m);
                last = --c->fabric->refs == 0;
            }
            if (last) {
                for (hipEvent_t e : c->fabric->ready) if (e) (void)hipEventDestroy(e);
                delete c->fabric;
            }
        }
        for (DevBuf *b : {&c->d_hdr, &c->d_prefix, &c->d_records, &c->d_wire}) b->release();
        if (c->h_hdr) (void)hipHostFree(c->h_hdr);
        delete c;
        return ACMI_OK;
    });
}

// root < 0: all-gatherv (every rank ends with the global arrays); else a gather to `root`
static int comm_exchange(acmi_comm *c, acmi_handle *h, uint64_t n_haystacks_local, int root, const void **d_prefix,
                         const void **d_records, uint64_t *n_records, uint64_t *n_haystacks) {
    return guarded(h, [&]() -> int {
        if (!c) return ACMI_E_ARG;
        if (!live(h)) return ACMI_E_INVALID;   // (no stream to join a collective on: the caller broke the contract)
        if (root >= c->world) return ACMI_E_ARG;
        HIPCHK(h, hipSetDevice(h->dev_id));
        hipStream_t st = h->stream;
        const int W = c->world;
        const bool receiver = root < 0 || root == c->rank;
        c->rounds = c->bytes_sent = c->bytes_received = 0;
        // what is wrong on THIS rank travels in the header; the rank joins the header round all the same
        uint64_t status = kCommOk;
        if (!h->have_result || n_haystacks_local != h->last_nh) status = kCommNoResult;
        else if (!c->cap_ok) status = kCommNoMemory;
        const uint64_t my_rec = status == kCommOk ? h->last_total : 0, my_hay = status == kCommOk ? h->last_nh : 0;
        int rc = comm_headers(c, my_rec, my_hay, status, st);
        if (rc < 0) return comm_fail(h, c, rc, c->last_error.empty() ? "header exchange failed" : c->last_error);
        const uint64_t *all = c->h_hdr + kHdrWords;
        uint64_t rec_base[kMaxRanks + 1] = {0}, hay_base[kMaxRanks + 1] = {0};
        int bad = -1;
        for (int r = 0; r < W; ++r) {
            rec_base[r + 1] = rec_base[r] + all[size_t(r) * kHdrWords];
            hay_base[r + 1] = hay_base[r] + all[size_t(r) * kHdrWords + 1];
            if (all[size_t(r) * kHdrWords + 2] != kCommOk && bad < 0) bad = r;
        }
        if (bad >= 0) {  // every rank sees the same headers and leaves here together
            const uint64_t why = all[size_t(bad) * kHdrWords + 2];
            const std::string text = "rank " + std::to_string(bad) + (why == kCommNoResult ? " has no result for the haystack count it gave"
                                                                      : why == kCommNoMemory ? " is out of device memory" : " reported a device error");
            return comm_fail(h, c, bad == c->rank ? (why == kCommNoMemory ? ACMI_E_NOMEM : ACMI_E_ARG) : ACMI_E_DEVICE, text);
        }
        const uint64_t total_rec = rec_base[W], total_hay = hay_base[W];
        // Global arrays: grown by the same rule on every rank (the totals are the same everywhere), and a rank whose
        // allocation fails tells the others in a second, tiny round -- taken by all ranks or by none.  (A rank that only
        // sends -- a gather's non-root -- holds its own slice of the prefix words and of the wire, no more.)
        // (... and when the ROOT is another than last time -- every rank knows that: the rank that becomes the receiver allocates
        // its record array inside the agreed round, not behind it where a failure would leave its peers waiting in the collective)
        const bool root_changed = root != c->last_root;
        c->last_root = root;
        if (total_rec > c->cap_rec || total_hay > c->cap_hay || !c->d_prefix.p || root_changed) {
            c->cap_rec = std::max(c->cap_rec, total_rec + total_rec / 4);
            c->cap_hay = std::max(c->cap_hay, total_hay + total_hay / 4);
            c->cap_ok = c->d_prefix.ensure((c->cap_hay + 1) * 8) == hipSuccess && c->d_wire.ensure(std::max<uint64_t>(16, c->cap_rec * 8)) == hipSuccess &&
                        (!receiver || c->d_records.ensure(std::max<uint64_t>(16, c->cap_rec * 16)) == hipSuccess);
            if (!c->cap_ok) (void)hipGetLastError();
            rc = comm_headers(c, 0, 0, c->cap_ok ? kCommOk : kCommNoMemory, st);
            if (rc < 0) return comm_fail(h, c, rc, c->last_error.empty() ? "status exchange failed" : c->last_error);
            for (int r = 0; r < W; ++r)
                if (all[size_t(r) * kHdrWords + 2] != kCommOk) {
                    // (every rank: the next call tries again from scratch -- also the rank whose allocation failed, which
                    // would otherwise report "out of memory" in every later header round without ever trying again)
                    c->cap_rec = c->cap_hay = 0;
                    c->cap_ok = true;
                    c->d_prefix.release();
                    c->d_records.release();
                    c->d_wire.release();
                    return comm_fail(h, c, r == c->rank ? ACMI_E_NOMEM : ACMI_E_DEVICE, "rank " + std::to_string(r) + " is out of device memory");
                }
        }
        if (receiver && c->d_records.cap < std::max<uint64_t>(16, total_rec * 16)) {   // (a rank that was a gather's sender so far)
            if (c->d_records.ensure(std::max<uint64_t>(16, c->cap_rec * 16)) != hipSuccess) { (void)hipGetLastError(); return comm_fail(h, c, ACMI_E_NOMEM, "out of device memory for the gathered records"); }
        }
        // this rank's slice, already global: haystack ids + hay_base, prefix words + rec_base (and the closing word); its
        // records a second time in the 8-byte form that travels
        uint64_t *g_prefix = c->d_prefix.as<uint64_t>();
        uint4 *g_records = receiver ? c->d_records.as<uint4>() : nullptr;
        uint2 *g_wire = c->d_wire.as<uint2>();
        launch_gather_pack(h->d_prefix().as<uint64_t>(), h->d_out().as<uint4>(), my_hay, my_rec, hay_base[c->rank], rec_base[c->rank],
                           g_prefix + hay_base[c->rank], g_records ? g_records + rec_base[c->rank] : nullptr, g_wire + rec_base[c->rank],
                           g_prefix + total_hay, total_rec, st);
        HIPCHK(h, hipGetLastError());
        // one group: rank r's slices from rank r, exactly their size
        std::vector<BcastOp> ops;
        for (int r = 0; r < W; ++r) {
            const uint64_t nr = rec_base[r + 1] - rec_base[r], nh = hay_base[r + 1] - hay_base[r];
            ops.push_back({g_wire + rec_base[r], g_wire + rec_base[r], size_t(nr) * 8, r});
            ops.push_back({g_prefix + hay_base[r], g_prefix + hay_base[r], size_t(nh) * 8, r});
            if (r == c->rank) c->bytes_sent = (W > 1 && root != c->rank ? nr * 8 + nh * 8 : 0);
            else if (receiver) c->bytes_received += nr * 8 + nh * 8;
        }
        rc = comm_bcast_group(c, ops, st, root);
        if (rc < 0) return comm_fail(h, c, rc, c->last_error.empty() ? "payload exchange failed" : c->last_error);
        ++c->rounds;
        // the other ranks' records, expanded where they belong: haystack from the prefix array, start = pos - pattern length
        if (receiver && W > 1)
            launch_gather_unpack(g_prefix, total_hay, g_wire, total_rec, rec_base[c->rank], rec_base[c->rank + 1], h->dA.pattern_length, g_records, st);
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipStreamSynchronize(st));
        c->total_rec = receiver ? total_rec : 0;
        c->total_hay = receiver ? total_hay : 0;
        if (d_prefix) *d_prefix = receiver ? c->d_prefix.p : nullptr;
        if (d_records) *d_records = receiver ? c->d_records.p : nullptr;
        if (n_records) *n_records = total_rec;
        if (n_haystacks) *n_haystacks = total_hay;
        return ACMI_OK;
    });
}

int acmi_comm_allgather_result(acmi_comm *c, acmi_handle *h, uint64_t n_haystacks_local, const void **d_prefix,
                               const void **d_records, uint64_t *n_records, uint64_t *n_haystacks) {
    return comm_exchange(c, h, n_haystacks_local, -1, d_prefix, d_records, n_records, n_haystacks);
}

int acmi_comm_gather_result(acmi_comm *c, acmi_handle *h, uint64_t n_haystacks_local, int root, const void **d_prefix,
                            const void **d_records, uint64_t *n_records, uint64_t *n_haystacks) {
    if (root < 0) return ACMI_E_ARG;
    return comm_exchange(c, h, n_haystacks_local, root, d_prefix, d_records, n_records, n_haystacks);
}

int acmi_comm_result_to_host(acmi_comm *c, uint64_t *prefix, acmi_record *records) {
    return guarded(nullptr, [&]() -> int {
        if (!c) return ACMI_E_ARG;
        if (!c->d_prefix.p || (c->total_rec && !c->d_records.p)) return ACMI_E_ARG;   // (no gathered result on this rank)
        if (hipSetDevice(c->dev) != hipSuccess) return ACMI_E_DEVICE;
        if (prefix && hipMemcpy(prefix, c->d_prefix.p, (c->total_hay + 1) * 8, hipMemcpyDeviceToHost) != hipSuccess) return ACMI_E_DEVICE;
        if (records && c->total_rec && hipMemcpy(records, c->d_records.p, c->total_rec * 16, hipMemcpyDeviceToHost) != hipSuccess) return ACMI_E_DEVICE;
        return ACMI_OK;
    });
}

int acmi_comm_get_stat(const acmi_comm *c, const char *key, double *value) {
    return guarded(nullptr, [&]() -> int {
        if (!c || !key || !value) return ACMI_E_ARG;
        const std::string k(key);
        if (k == "world") *value = c->world;
        else if (k == "rank") *value = c->rank;
        else if (k == "backend_ranks") *value = c->backend_ranks;          // ncclCommCount (RCCL transport)
        else if (k == "transport") *value = c->world == 1 ? 0 : (c->fabric ? 2 : 1);  // 0 none, 1 RCCL, 2 process-local
        else if (k == "rounds") *value = double(c->rounds);                // collectives of the last exchange
        else if (k == "bytes_sent") *value = double(c->bytes_sent);
        else if (k == "bytes_received") *value = double(c->bytes_received);
        else if (k == "records") *value = double(c->total_rec);
        else if (k == "haystacks") *value = double(c->total_hay);
        else return ACMI_E_ARG;
        return ACMI_OK;
    });
}

const char *acmi_comm_last_error(const acmi_comm *c) { return c ? c->last_error.c_str() : "no communicator"; }

}  // extern "C"

