"""BASELINE configs[4] (adversarial high-hit-rate sets): G3 = patterns a^1..a^16 over runs of 'a'
(16 records per byte, 256 B written per byte read) and G4 = all {a,b} strings of length 1..4 over random
{a,b} text (4 records per byte).  Reports records/s and the HBM write rate of the whole call, checks the
count formula of SURVEY 8(d) and a sampled haystack against the closed form.

    python tools/bench_dense.py [n_haystacks=4096] [hay_len=8192]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from php_aho_corasick_amd import api, batch, synth

nh = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hl = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
light = len(sys.argv) > 3 and sys.argv[3] == "light"   # counter passes: default drain only, 3 calls, no result copy


def run(name, ds, expect_total, refill=0):
    res = api.ahocorasick_init([{"id": k, "value": ds.pattern(k)} for k in range(ds.n_patterns)])
    api.ahocorasick_finalize(res)
    res.set_option("profile", 2)
    res.set_option("refill", refill)
    reps = nh // ds.n_haystacks
    d_blob = torch.from_numpy(ds.hay_blob).cuda().repeat(reps).contiguous()
    d_off = torch.arange(nh + 1, dtype=torch.int64, device="cuda") * hl
    batch.bind_current_stream(res)
    ms, total = [], 0
    for _ in range(3 if light else 6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        total = batch.match_device(res, d_blob, d_off)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    t = float(np.mean(ms[2:]))
    if light:
        print(f"{name}: {total} records, {t:.3f} ms")
        api.ahocorasick_deinit(res)
        return expect_total is None or total == expect_total
    ok = expect_total is None or total == expect_total
    print(f"{name} refill={refill}: {nh} x {hl} B, {total} records ({total / (nh * hl):.2f}/byte) in {t:.3f} ms = "
          f"{total / t / 1e6:.2f} G records/s, {total * 16 / t / 1e6:.0f} GB/s written, {nh * hl / t / 1e6:.1f} GB/s read; "
          f"scan kernel {res.stat('scan_ms'):.3f} ms, overflow items {int(res.stat('overflow_items'))}, "
          f"items {int(res.stat('items'))}, count {'ok' if ok else 'MISMATCH'}", flush=True)
    # sampled haystack against the closed form / the first copy
    prefix, rec = batch.result_tensors(res, nh, total)
    h = nh - 1
    a, b = int(prefix[h]), int(prefix[h + 1])
    a0, b0 = int(prefix[0]), int(prefix[1])
    last, first = rec[a:b], rec[a0:b0]
    same = bool((last[:, 1:] == first[:, 1:]).all()) if reps > 1 and (h % ds.n_haystacks) == 0 else None
    print(f"   haystack {h}: {b - a} records, hay id ok {bool((last[:, 0] == h).all())}, equal to haystack 0's: {same}")
    api.ahocorasick_deinit(res)
    del d_blob
    torch.cuda.empty_cache()
    return ok


k = 16
ok = True
for refill in ((0,) if light else (1, 2)):
    ok &= run("G3 (a^1..a^16 over a-runs)", synth.g3(n_hay=1, hay_len=hl, k=k), nh * (k * hl - k * (k - 1) // 2), refill)
    ok &= run("G4 ({a,b} strings <= 4)", synth.g4(n_hay=min(nh, 64), hay_len=hl), None, refill)
if light:
    # calibration for WRITE_SIZE (uncalibrated on gfx950, MI355X_MICROARCH.md): a fill of exactly 2 GiB in the same pass
    cal = torch.empty(1 << 31, dtype=torch.uint8, device="cuda")
    cal.fill_(1)
    torch.cuda.synchronize()
sys.exit(0 if ok else 1)
