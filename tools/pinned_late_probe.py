"""dev helper: does host-memory churn before a page-locked allocation slow the device-to-host copy into it?
    python tools/pinned_late_probe.py [churn_gb]"""
import sys, time, ctypes as C
import numpy as np, torch as pt
churn = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
def thp():
    out = {}
    for line in open("/proc/meminfo"):
        if line.startswith(("AnonHugePages", "MemFree", "HugePages_Total")): out[line.split(":")[0]] = line.split()[1]
    return out
print("THP enabled:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
n = 2 << 30
src = pt.empty(n, dtype=pt.uint8, device="cuda")
if churn > 0:
    t0 = time.perf_counter()
    keep = []
    for i in range(int(churn * 16)):                    # 64-MB pieces, touched, every eighth kept for a while
        a = np.ones(64 << 20, dtype=np.uint8)
        if i % 8 == 0: keep.append(a[::4096].copy())
    del keep
    print(f"churned {churn} GB in {time.perf_counter() - t0:.1f} s")
def rate(dst):
    dst.copy_(src); pt.cuda.synchronize()
    e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
    e0.record(); dst.copy_(src, non_blocking=True); e1.record(); pt.cuda.synchronize()
    return n / (e0.elapsed_time(e1) * 1e-3) / 1e9
before = thp()
dst = pt.empty(n, dtype=pt.uint8, pin_memory=True)
print(f"churn {churn} GB: D2H into a fresh page-locked buffer {rate(dst):.1f} GB/s; meminfo before {before} after {thp()}")
