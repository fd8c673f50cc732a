"""dev helper: per-tile row multiplicity histogram (how many cells of a 64-cell Morton tile use a staged row)"""
import sys, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric = bench.synthetic_cylinder3d(cfg)
geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
        geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=5, min_metric=0.75)
tree.refine()
centers = tree.all_centers.numpy(); levels = tree.all_levels.numpy().ravel()
knn = hipops.KnnIndex(x); idx, _ = knn.query(centers, 26); idx = idx.long()
nc, k = idx.shape
c = centers; lo = c.min(0); q = ((c - lo) / (c.max(0) - lo).max() * 2097151).astype(np.int64)
def spread(v):
    v &= 0x1fffff; v = (v | v << 32) & 0x1f00000000ffff; v = (v | v << 16) & 0x1f0000ff0000ff
    v = (v | v << 8) & 0x100f00f00f00f00f; v = (v | v << 4) & 0x10c30c30c30c30c3; v = (v | v << 2) & 0x1249249249249249
    return v
perm = pt.from_numpy(np.argsort(spread(q[:, 0]) | spread(q[:, 1]) << 1 | spread(q[:, 2]) << 2, kind="stable")).cuda()
print("levels:", {int(l): int((levels == l).sum()) for l in np.unique(levels)})
for tc in (64, 128, 256):
    tile = (pt.arange(nc, device="cuda") // tc)[:, None].expand(nc, k)
    key = tile.reshape(-1) * (1 << 32) + idx[perm].reshape(-1)
    u, cnt = pt.unique(key, return_counts=True)
    tot = u.numel()
    h = pt.bincount(cnt.clamp(max=9))
    print(f"TC={tc}: staged rows {tot}, dedup {nc*k/tot:.2f}; multiplicity share of staged rows: " +
          " ".join(f"{m}:{h[m].item()/tot:.2f}" for m in range(1, len(h))) +
          f" | share of gathers from single-use rows: {h[1].item()/(nc*k):.2f}")
