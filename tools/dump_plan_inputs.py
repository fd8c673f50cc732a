"""dev helper: write the neighbour table + cell centres of a bench workload to gpurun_out/ so that tile-packing
strategies of the plan builder can be studied offline (tools/plan_experiment.cpp)"""
import sys, logging
import numpy as np, torch as pt
sys.path.insert(0, ".")  # run from the repo root
import bench
from sparsespatialsampling_amd import geometry, hipops
from sparsespatialsampling_amd.s_cube import SamplingTree
logging.getLogger().setLevel(logging.WARNING)
name = sys.argv[1] if len(sys.argv) > 1 else "cylinder3D_Re3900"
cfg = dict(bench.WORKLOADS[name])
x, metric = bench.synthetic_cylinder3d(cfg)
geos = [geometry.CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
        geometry.CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"], min_metric=cfg["min_metric"])
tree.refine()
centers = tree.all_centers.numpy()
levels = tree.all_levels.numpy().reshape(-1).astype(np.int8)
idx, _ = hipops.KnnIndex(x).query(centers, cfg.get("k", 26))
idx = np.sort(idx.cpu().numpy().astype(np.int32), axis=1)
print(name, "cells", centers.shape, "idx", idx.shape, "n_src", len(x))
np.savez_compressed(f"gpurun_out/plan_inputs_{name}.npz", centers=centers, idx=idx, n_src=len(x), levels=levels)
