import os, sys, time
import torch as pt
x = pt.empty(64 << 20, dtype=pt.uint8, device="cuda").zero_()
pt.cuda.synchronize()
for _ in range(2): y = x.cpu()
t0 = time.perf_counter()
for _ in range(5): y = x.cpu()
dt = (time.perf_counter() - t0) / 5
h = pt.empty(64 << 20, dtype=pt.uint8)
for _ in range(2): z = h.cuda()
pt.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): z = h.cuda()
pt.cuda.synchronize(); du = (time.perf_counter() - t0) / 5
print(f"GPU_PINNED_MIN_XFER_SIZE={os.environ.get('GPU_PINNED_MIN_XFER_SIZE')}: 64 MiB pageable D2H {64/1024/dt:.1f} GiB/s, H2D {64/1024/du:.1f} GiB/s")
