"""dev helper (experiment): does chunk-phase alignment of the resident tiles raise the L2 hit rate?
mode 'full':   one plan / one launch over all cells (steady state: resident tiles are at staggered chunk phases)
mode 'waves':  cells cut along a space-filling curve into groups of 512 tiles; one launch per group, so every group
               starts all its tiles together (aligned phases), at the price of launch tails
Run each mode under `rocprofv3 --pmc FETCH_SIZE` and compare the bytes fetched per staged row."""
import sys, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
mode = sys.argv[1]
group_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = dict(bench.WORKLOADS["cylinder3D_Re3900"])
x, metric = bench.synthetic_cylinder3d(cfg)
geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
        geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
tree.refine()
centers = tree.all_centers.numpy()
levels = tree.all_levels.numpy().reshape(-1)
if len(sys.argv) > 3 and sys.argv[3] == "fine":          # only the cells of the dominant fine level (homogeneous tiles)
    centers = np.ascontiguousarray(centers[levels == 8])
q = ((centers - centers.min(0)) / (centers.max(0) - centers.min(0)).max() * 1023).astype(np.int64)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    return (v | (v << 2)) & 0x09249249
order = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind="stable")
centers = np.ascontiguousarray(centers[order])
k, T = 26, 1000
idx, dist = hipops.KnnIndex(x).query(centers, k)
w = hipops.idw_weights(dist)
nc = len(centers)
data = hipops.padded_rows(len(x), T, pt.float32, "cuda"); data.normal_()
out = pt.empty((nc, T), dtype=pt.float64, device="cuda")
if mode == "full":
    groups = [(0, nc)]
else:
    per = group_tiles * 40                               # ~40 cells per tile on this grid
    groups = [(a, min(nc, a + per)) for a in range(0, nc, per)]
plans = [(a, b, hipops.InterpPlan(idx[a:b].contiguous(), len(x), centers[a:b])) for a, b in groups]
print(mode, "groups", len(plans), "tiles", sum(p.n_tiles for _, _, p in plans), "staged rows", sum(p.total_rows for _, _, p in plans), flush=True)
for rep in range(3):
    e0, e1 = pt.cuda.Event(enable_timing=True), pt.cuda.Event(enable_timing=True)
    e0.record()
    for a, b, p in plans:
        p.interp(w[a:b], data, out=out[a:b])
    e1.record(); pt.cuda.synchronize()
    print(f"rep {rep}: {e0.elapsed_time(e1):.3f} ms", flush=True)
