"""dev helper: host-side profile of SamplingTree.refine() on the bench workload (not part of the product)"""
import cProfile, pstats, sys, time, logging
import numpy as np, torch as pt
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sparsespatialsampling_amd import geometry
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
cfg = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cylinder3D_Re3900"])
x, metric = bench.synthetic_cylinder3d(cfg)
geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
        geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
warm = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
warm.refine()                     # (the first refine of a process also pays for allocator pools and lazy library loading)
warm.close()
pt.cuda.synchronize()
t0 = time.perf_counter()
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
pt.cuda.synchronize(); t1 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
tree.refine()
pt.cuda.synchronize()
pr.disable(); t2 = time.perf_counter()
print("init %.3f s  refine %.3f s" % (t1 - t0, t2 - t1))
print({k: v for k, v in tree.data_final_mesh.items() if k.startswith("t_") or k in ("iterations", "n_cells")})
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
