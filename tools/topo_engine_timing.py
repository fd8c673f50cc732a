"""dev helper: refine() wall-clock of a bench workload with the device-resident topology engine and with the host engine
(S3_TOPOLOGY), interleaved in one process; prints the renumbering share as well"""
import os, sys, time, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch as pt
import bench
from sparsespatialsampling_amd import geometry, s_cube
logging.getLogger().setLevel(logging.WARNING)
name = sys.argv[1] if len(sys.argv) > 1 else "cylinder3D_Re3900"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x, metric, geos, tree_kw = bench.build_case(name, dict(bench.WORKLOADS[name]), geometry)
xs, ms = pt.from_numpy(x), pt.from_numpy(metric)
for rep in range(reps):
    for mode in ("device", "host"):
        os.environ["S3_TOPOLOGY"] = mode
        pt.cuda.synchronize()
        t0 = time.perf_counter()
        tree = s_cube.SamplingTree(xs, ms, geos, **tree_kw)
        t1 = time.perf_counter()
        tree.refine()
        pt.cuda.synchronize()
        t2 = time.perf_counter()
        info = tree.data_final_mesh
        print(f"{name} rep {rep} {mode:6s}: constructor {t1 - t0:.3f} s, refine {t2 - t1:.3f} s (uniform {info['t_uniform']:.3f}, adaptive "
              f"{info['t_adaptive']:.3f}, geometry {info['t_geometry'] or 0:.3f}, renumbering {info['t_renumbering']:.3f}), "
              f"{len(tree.all_centers)} leaves, {tree._topo_engine.n_created} cells", flush=True)
        tree.close()
        del tree
