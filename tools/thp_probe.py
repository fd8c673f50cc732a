"""dev helper: first-touch cost of a fresh 1-GB numpy array with and without MADV_HUGEPAGE (the finished grid of box5e7 is
1 GB of downloads into fresh pageable memory)"""
import ctypes, time, numpy as np
libc = ctypes.CDLL("libc.so.6", use_errno=True)
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
n = 1 << 30
for advise in (False, True, False, True):
    a = np.empty(n, dtype=np.uint8)
    if advise:
        addr = a.ctypes.data
        lo = (addr + (1 << 21) - 1) & ~((1 << 21) - 1)
        hi = (addr + n) & ~((1 << 21) - 1)
        rc = libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(hi - lo), 14)     # MADV_HUGEPAGE
    t0 = time.perf_counter()
    a[::4096] = 1
    t1 = time.perf_counter()
    a[:] = 2
    t2 = time.perf_counter()
    print(f"madvise={advise}: first touch {1e3 * (t1 - t0):.1f} ms, full write {1e3 * (t2 - t1):.1f} ms", (rc if advise else None))
    del a
