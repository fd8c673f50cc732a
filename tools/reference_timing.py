"""dev-container helper: wall-clock of the REAL reference (import shims of tests/golden/ref_stubs.py) on the reduced bench
workload `cylinder3D_small` -- refine() and interpolate_data().  Never runs on the GPU box (no /root/reference there)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import ref_stubs  # noqa
import numpy as np, torch as pt

if __name__ == "__main__":
    import bench
    from sparseSpatialSampling.s_cube import SamplingTree
    from sparseSpatialSampling.export import interpolate_data
    from sparseSpatialSampling.geometry import CubeGeometry, CylinderGeometry3D
    from sklearn.neighbors import NearestNeighbors
    cfg = dict(bench.WORKLOADS["cylinder3D_small"])
    x, metric = bench.synthetic_cylinder3d(cfg)
    geos = [CubeGeometry("domain", True, cfg["lo"], [float(v) for v in cfg["hi"]]),
            CylinderGeometry3D("cylinder", False, [(0.8, 1.0, -1.0), (0.8, 1.0, 1.0)], 0.05, refine=True)]
    n_jobs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    t0 = time.perf_counter()
    tree = SamplingTree(pt.from_numpy(x), pt.from_numpy(metric), geos, uniform_level=cfg["uniform_levels"],
                        min_metric=cfg["min_metric"], n_jobs=n_jobs)
    tree.refine()
    t1 = time.perf_counter()
    centers = tree.all_centers.numpy()
    print(f"REFERENCE refine: N={len(x)} -> {len(centers)} leaves, {tree.data_final_mesh['iterations']} iterations, "
          f"{t1 - t0:.2f} s (n_jobs={n_jobs})", flush=True)
    nn = NearestNeighbors(n_neighbors=26, n_jobs=n_jobs).fit(x)
    t2 = time.perf_counter(); dist, idx = nn.kneighbors(centers); t3 = time.perf_counter()
    w = 1.0 / pt.clamp(pt.from_numpy(dist), min=1e-12); w /= w.sum(1, keepdim=True)
    data = pt.randn((len(x), 1, 256), dtype=pt.float32)
    t4 = time.perf_counter(); out = interpolate_data(w, pt.from_numpy(idx), data); t5 = time.perf_counter()
    print(f"REFERENCE knn cache {t3 - t2:.2f} s; interpolate_data {len(centers)} cells x 256 snapshots: {t5 - t4:.2f} s = "
          f"{len(centers) * 256 / (t5 - t4) / 1e6:.1f} Mcells*snapshots/s ({pt.get_num_threads()} torch threads)", flush=True)
    np.save("/tmp/ref_small_centers.npy", centers)
