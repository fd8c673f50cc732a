"""dev helper: neighbour-sharing statistics of consecutive-cell tiles (creation order vs Morton order)"""
import sys, time, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")  # run from the repo root
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
cfg = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cylinder3D_Re3900"])
x, metric = bench.synthetic_cylinder3d(cfg)
geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
        geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
t0 = time.perf_counter()
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
tree.refine(); pt.cuda.synchronize()
print("refine+init %.3f s" % (time.perf_counter() - t0), tree.data_final_mesh["t_uniform"], tree.data_final_mesh["t_adaptive"])
centers = tree.all_centers.numpy()
knn = hipops.KnnIndex(x)
idx, _ = knn.query(centers, 26)
idx = idx.long()
nc, k = idx.shape
def morton(c):
    lo, hi = c.min(0), c.max(0)
    q = ((c - lo) / (hi - lo).max() * 1023).astype(np.int64)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
orders = {"creation": pt.arange(nc, device="cuda"), "morton": pt.from_numpy(np.argsort(morton(centers), kind="stable")).cuda()}
for name, perm in orders.items():
    for tc in (8, 16, 32, 64, 128):
        ii = idx[perm]
        nt = (nc + tc - 1) // tc
        tile = (pt.arange(nc, device="cuda") // tc)[:, None].expand(nc, k)
        key = tile.reshape(-1) * (1 << 32) + ii.reshape(-1)
        u = pt.unique(key)
        ut = pt.bincount((u >> 32), minlength=nt).double()
        print(f"{name:9s} TC={tc:4d}: mean U_t={ut.mean().item():8.1f}  max={ut.max().item():6.0f}  p99={ut.quantile(0.99).item():7.0f} sharing={(nc*k)/u.numel():.2f}")
print("global unique", pt.unique(idx).numel(), "of", nc * k)
