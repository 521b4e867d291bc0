"""HBM write rate reference: hipMemsetAsync / torch fill of 8 GiB, for the dense-match (write-bound) roofline."""
import torch, time
x = torch.empty(8 << 30, dtype=torch.uint8, device="cuda")
y = torch.empty((8 << 30) // 16, 4, dtype=torch.int32, device="cuda")
for name, fn in (("fill_ u8", lambda: x.fill_(1)), ("zero_ (memset)", lambda: x.zero_()), ("fill_ int32x4", lambda: y.fill_(7))):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    print(f"{name}: {(8 << 30) / t / 1e9:.0f} GB/s")
