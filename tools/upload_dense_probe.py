"""dev helper: a host batch of short ragged rows (25 fp32 snapshots = 100 bytes) up to the device -- laid out with the device
pitch of 128 bytes on the host (one contiguous copy, what ExportData._upload does) against dense over PCIe (100-byte pitch,
22 % fewer bytes) and re-pitched by a device kernel"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
from sparsespatialsampling_amd import hipops

n, t = 4_991_774, int(sys.argv[1]) if len(sys.argv) > 1 else 25
rng = np.random.default_rng(0)
used = np.sort(rng.choice(n, 2_430_607, replace=False)).astype(np.int32)
order = rng.permutation(len(used))                       # (ExportData uploads the rows in ascending order; see below)
host = pt.empty((n, t), dtype=pt.float32).normal_()
dev = hipops.device()
pitched = hipops.padded_rows(len(used), t, pt.float32, dev)
dense = pt.empty((len(used), t), dtype=pt.float32, device=dev)


def timed(label, fn, reps=5):
    fn(); pt.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    pt.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:60s} {dt * 1e3:7.2f} ms  {len(used) * t * 4 / dt / 1e9:6.1f} GB/s of payload", flush=True)


timed("pitched on the host, one copy (current)", lambda: hipops.upload_rows_indexed(host, used, pitched))
timed("dense over PCIe", lambda: hipops.upload_rows_indexed(host, used, dense))
timed("dense over PCIe + re-pitch on the device", lambda: (hipops.upload_rows_indexed(host, used, dense), hipops.gather_rows(dense, None, pitched)))
chk = pitched.clone()
hipops.upload_rows_indexed(host, used, pitched); pt.cuda.synchronize()
print("same rows:", bool(pt.equal(chk, pitched)))
# where the time goes: the same number of bytes as one contiguous block (no gather on the host), and with pinned source memory
block = host[:len(used)].contiguous()
timed("contiguous block, dense (no gather on the host)", lambda: hipops.upload_rows(block, dense))
pinned = pt.empty((len(used), t), dtype=pt.float32, pin_memory=True)
pinned.copy_(block)
timed("page-locked source, one hipMemcpyAsync (PCIe alone)", lambda: dense.copy_(pinned, non_blocking=True))
