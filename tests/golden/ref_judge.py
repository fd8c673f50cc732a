"""
The reference's own loader / XDMF writer as the judge of files THIS build wrote (development container only: the
reference never travels to the GPU box).  Run as a script in its own process, so that the import shims of
``ref_stubs.py`` (numba, flowtorch, shapely, h5py -> h5py_standin.py) and the name ``sparseSpatialSampling`` = the
reference never leak into the test process:

    python ref_judge.py load <dir> <file.h5> <out.npz>        # reference Dataloader (data.py:22-300) -> everything it exposes
    python ref_judge.py xdmf <dir> <file.h5> <mixed 0|1>      # reference XDMFWriter (data.py:504-777) writes <file>.xdmf
    python ref_judge.py write <dir> <src.h5> <dst.h5>         # reference Datawriter (data.py:303-501) through inputs.datawriter_script
    python ref_judge.py fuzz_export <dir> <seed0> <n>         # n random exports (inputs.random_export_case) by the REAL reference,
                                                              # grid generation included, one sub-directory per seed

Contains no reference code: it calls the reference's public classes.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402,F401

import numpy as np  # noqa: E402
import torch as pt  # noqa: E402


def load(directory, file_name, out):
    from sparseSpatialSampling.data import Dataloader
    ld = Dataloader(directory, file_name, dtype=pt.float64)
    res = {"vertices": ld.vertices.numpy(), "nodes": ld.nodes.numpy(), "faces": ld.faces.numpy(), "levels": ld.levels.numpy(),
           "metric": ld.metric.numpy(), "weights": ld.weights.numpy(),
           "write_times": np.array(json.dumps(ld.write_times)), "field_names": np.array(json.dumps(ld.field_names))}
    fields = sorted({f for v in ld.field_names.values() for f in v})
    for f in fields:
        times = [t for t in ld.write_times if f in ld.field_names[t]]
        res[f"snap_{f}"] = ld.load_snapshot(f, times).numpy()
        res[f"times_{f}"] = np.array(json.dumps(times))
    if len(fields) > 1:
        both = [t for t in ld.write_times if all(f in ld.field_names[t] for f in fields[:2])]
        pair = ld.load_snapshot(fields[:2], both)
        res["pair_shapes"] = np.array(json.dumps([list(p.shape) for p in pair]))
    np.savez(out, **res)


def xdmf(directory, file_name, mixed):
    from sparseSpatialSampling.data import XDMFWriter
    XDMFWriter(directory, file_name, mixed=bool(int(mixed))).write_xdmf()


def write(directory, src_name, dst_name):
    """the calls of inputs.datawriter_script on the REFERENCE's Datawriter / Dataloader"""
    from inputs import datawriter_script
    from sparseSpatialSampling.data import Dataloader, Datawriter
    datawriter_script(Dataloader, Datawriter, directory, src_name, dst_name)


def fuzz_export(directory, seed0, n):
    import sparseSpatialSampling.geometry as ref_geometry
    from inputs import random_export_case, random_export_cloud, run_random_export
    from sparseSpatialSampling.export import ExportData
    from sparseSpatialSampling.sparse_spatial_sampling import SparseSpatialSampling
    import logging
    logging.getLogger().setLevel(logging.ERROR)
    for seed in range(int(seed0), int(seed0) + int(n)):
        case = random_export_case(seed)
        x, y, geos, kw = random_export_cloud(case, ref_geometry)
        out = os.path.join(directory, f"seed{seed}")
        os.makedirs(out)
        s3 = SparseSpatialSampling(pt.from_numpy(x), pt.from_numpy(y), geos, out, "case", n_jobs=1, **kw)
        s3.execute_grid_generation()
        run_random_export(case, s3, ExportData, pt.from_numpy, x)


def facade(directory, case, out_json):
    """reference SparseSpatialSampling on refine case `case` (inputs.REFINE_CASES): what execute_grid_generation leaves behind"""
    import sparseSpatialSampling.geometry as ref_geometry
    from inputs import describe_facade, refine_inputs
    from sparseSpatialSampling.sparse_spatial_sampling import SparseSpatialSampling
    x, y, geos, kw = refine_inputs(case, ref_geometry)
    kw = {{"uniform_level": "uniform_levels", "n_cells": "n_cells_max"}.get(k, k): v for k, v in kw.items()}
    s3 = SparseSpatialSampling(pt.from_numpy(x), pt.from_numpy(y), geos, directory, "case", n_jobs=1, **kw)
    s3.execute_grid_generation()
    json.dump(describe_facade(s3, directory), open(out_json, "w"))


def errors(out_json):
    """outcome (exception type or "ok") of every call of inputs.invalid_calls on the REFERENCE's classes"""
    import sparseSpatialSampling.geometry as ref_geometry
    from inputs import invalid_calls, outcomes_of
    from sparseSpatialSampling.s_cube import SamplingTree
    from sparseSpatialSampling.sparse_spatial_sampling import SparseSpatialSampling
    json.dump(outcomes_of(invalid_calls(ref_geometry, SparseSpatialSampling, SamplingTree)), open(out_json, "w"))


def logs(directory, out_json):
    """every logging record of inputs.logged_run on the REFERENCE's classes"""
    import sparseSpatialSampling.geometry as ref_geometry
    from inputs import logged_run
    from sparseSpatialSampling.export import ExportData
    from sparseSpatialSampling.sparse_spatial_sampling import SparseSpatialSampling
    json.dump(logged_run(ref_geometry, SparseSpatialSampling, ExportData, directory), open(out_json, "w"))


if __name__ == "__main__":
    {"load": load, "xdmf": xdmf, "write": write, "fuzz_export": fuzz_export, "facade": facade, "errors": errors,
     "logs": logs}[sys.argv[1]](*sys.argv[2:])
