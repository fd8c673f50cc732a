"""dev helper: where the time of compute_svd goes at the bench's shape (461 130 x 1000 f64 on the device): mean, Gram kernel,
eigen-solve (library), mode GEMM -- steady state (third call)
    python tools/svd_breakdown.py [n_rows] [t] [rank]"""
import sys, time
import numpy as np, torch as pt
sys.path.insert(0, ".")
from sparsespatialsampling_amd import svd, metrics, hipops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 461130
t = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 50
gen = pt.Generator(device="cuda").manual_seed(5)
x = pt.empty((n, t), dtype=pt.float64, device="cuda").normal_(generator=gen)
modes = pt.empty((n, 40), dtype=pt.float64, device="cuda").normal_(generator=gen)
x += modes @ pt.empty((40, t), dtype=pt.float64, device="cuda").normal_(generator=gen) * 5
area = pt.rand(n, dtype=pt.float64, device="cuda", generator=gen) + 0.1
def clock(fn, reps=3):
    out = None
    for _ in range(reps):
        pt.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); pt.cuda.synchronize(); dt = time.perf_counter() - t0
    return out, dt * 1e3
mean, t_mean = clock(lambda: metrics.temporal_mean(x))
gram, t_gram = clock(lambda: svd.weighted_gram(x, mean, area))
(lam, vec), t_eigh = clock(lambda: svd._eigh(gram))
b = (vec[:, -rank:] ).contiguous().cuda()
u, t_gemm = clock(lambda: svd.centered_gemm(x, mean, b))
_, t_all = clock(lambda: svd.compute_svd(x, area, rank))
_, t_none = clock(lambda: svd.compute_svd(x, area, None), reps=2)
print(f"n={n} t={t} rank={rank}: mean {t_mean:.1f} ms, Gram {t_gram:.1f}, eigen-solve {t_eigh:.1f}, mode GEMM {t_gemm:.1f}; "
      f"compute_svd(rank={rank}) {t_all:.1f} ms, compute_svd(rank=None) {t_none:.1f} ms")
g = gram / gram.diagonal().max()
_, t_dev = clock(lambda: pt.linalg.eigh(g))
_, t_vals = clock(lambda: pt.linalg.eigvalsh(g))
gc = g.cpu()
_, t_cpu = clock(lambda: pt.linalg.eigh(gc), reps=2)
print(f"eigh on the device {t_dev:.1f} ms, eigvalsh {t_vals:.1f} ms, eigh on the host ({pt.get_num_threads()} threads) {t_cpu:.1f} ms")
