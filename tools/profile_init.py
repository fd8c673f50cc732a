"""dev helper: host-side profile of the SamplingTree constructor on the bench workload"""
import cProfile, pstats, sys, time, logging
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
hipops.device(); pt.zeros(1, device="cuda"); pt.cuda.synchronize()
for rep in range(2):
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
    pt.cuda.synchronize()
    pr.disable()
    print("init %.3f s" % (time.perf_counter() - t0))
    tree._backend.close()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
