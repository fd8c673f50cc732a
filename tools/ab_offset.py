"""dev helper: does the base address of the snapshot matrix / of the output move the planned kernel's time?  One big allocation, the
same batch placed at different byte offsets inside it (and the output likewise), interleaved timing in one process."""
import os, sys, logging, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
T = 1000
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric, geos, kw = bench.build_case("cylinder3D_Re3900", cfg, geometry)
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, **kw); tree.refine()
centers = tree.all_centers.numpy(); tree.close()
k = 26
knn = hipops.KnnIndex(x, hipops.knn_occupancy(k, 3)); idx, dist = knn.query(centers, k); knn.close()
w = hipops.idw_weights(dist)
used, remap = hipops.referenced_rows([idx], len(x), coords=x)
hipops.remap_indices(idx, remap)
n, nc = int(used.numel()), len(centers)
plan = hipops.InterpPlan(idx, n, centers); plan.set_weights(w)
pitch = hipops.padded_rows(4, T, pt.float32, "cuda").stride(0)
slack = 64 << 20
big = pt.empty(n * pitch + slack // 4, dtype=pt.float32, device="cuda")
big.normal_()
obig = pt.empty(nc * T + slack // 8, dtype=pt.float64, device="cuda")
print("base addresses: data %#x (mod 2 MiB %#x), out %#x" % (big.data_ptr(), big.data_ptr() % (2 << 20), obig.data_ptr()))
offsets = [0, 128, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 20, (16 << 20) + 128 * 7]
res = {}
for rnd in range(4):
    for off in offsets:
        data = big[off // 4: off // 4 + n * pitch].view(n, pitch)[:, :T]
        out = obig[off // 8: off // 8 + nc * T].view(nc, T)
        plan.interp(w, data, out=out); pt.cuda.synchronize()
        e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(8):
            plan.interp(w, data, out=out)
        e1.record(); pt.cuda.synchronize()
        if rnd: res.setdefault(off, []).append(e0.elapsed_time(e1) / 8)
for off, t in res.items():
    print(f"offset {off:>9d} B: median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}")
