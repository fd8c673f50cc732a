"""
Host logic of ExportData (batching state machine, HDF5 layout, XDMF text) on CPU, writing REAL HDF5 files through the
package's native sink (libs3h5.so on the HDF5 C library; h5py if that is what is installed).  The GPU entry points are
replaced by oracle-backed stand-ins (test-only monkeypatch); tests/test_gpu_refine.py runs the same class with the real
kernels.  ``test_reference_fixture_file`` reads the reference's own test file (tests/golden/s_cube_test_dataset.h5, a data
file of the reference's tests) with this package's loader and checks what the reference's
tests/test_s_cube_dataloader.py:40-57 checks.
"""
import os
import types

import numpy as np
import pytest
import torch as pt

from oracle import s3_oracle as orc
from sparsespatialsampling_amd import h5io

pytestmark = pytest.mark.skipif(h5io.native_lib() is None and __import__("importlib").util.find_spec("h5py") is None,
                                reason="neither libs3h5.so nor h5py available")


def dump(path):
    """{dataset path: array} of a whole HDF5 file"""
    out = {}
    with h5io.open_h5(path, "r") as f:
        def walk(group):
            for k in f.keys(group):
                p = k if group == "/" else f"{group}/{k}"
                try:
                    out[p] = f.read(p)
                except (FileNotFoundError, h5io.H5Error):
                    walk(p)
        walk("/")
    return out


class _CpuKnn:
    def __init__(self, pts, target_occupancy=0.0):
        self.pts = pts.numpy() if isinstance(pts, pt.Tensor) else np.asarray(pts)
        self.n, self.dim = self.pts.shape

    def query(self, q, k):
        q = q.numpy() if isinstance(q, pt.Tensor) else np.asarray(q)
        idx, dist = orc.knn(self.pts, q, k)
        return pt.from_numpy(idx.astype(np.int32)), pt.from_numpy(dist)

    def close(self):
        pass


def _cpu_ops():
    ops = types.SimpleNamespace()
    ops.KnnIndex = _CpuKnn
    ops.device = lambda: pt.device("cpu")
    ops.synchronize = lambda: None
    ops.to_device = lambda x, dtype=None: (pt.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x).to(
        dtype if dtype is not None else x.dtype).contiguous()
    ops.idw_weights = lambda dist: pt.from_numpy(orc.idw_weights(dist.numpy()))
    ops.interp = lambda w, idx, data, out=None: pt.from_numpy(orc.interp(w.numpy(), idx.numpy().astype(np.int64),
                                                                          data.numpy()))
    ops.InterpPlan = type("InterpPlan", (), {"__init__": lambda self, *a, **k: None, "n_table": None,
                                            "set_weights": lambda self, w: None,
                                            "set_source_ids": lambda self, ids, n: setattr(self, "n_table", n),
                                            "supports": staticmethod(lambda k, d: False)})
    ops.padded_rows = lambda n_rows, row_len, dtype, dev, extra_lines=0: pt.empty((n_rows, row_len + 3), dtype=dtype)[:, :row_len]

    def referenced_rows(tables, n_src, coords=None):
        used = np.unique(np.concatenate([t.numpy().ravel() for t in tables])).astype(np.int32)
        remap = np.full(n_src, -1, dtype=np.int32)
        remap[used] = np.arange(len(used), dtype=np.int32)
        return pt.from_numpy(used), pt.from_numpy(remap)

    def remap_indices(idx, remap):
        idx.copy_(remap[idx.long()])
        return idx

    def gather_rows(src, ids, dst):
        dst.copy_(src if ids is None else src[ids.long()])
        return dst
    ops.referenced_rows, ops.remap_indices, ops.gather_rows = referenced_rows, remap_indices, gather_rows
    ops.knn_occupancy = lambda k, dim: 0.0
    ops.upload_rows = lambda host, rows: rows.copy_(host)
    ops.upload_rows_indexed = lambda host, ids, rows: rows.copy_(host[pt.from_numpy(np.asarray(ids)).long()])
    ops.snapshot_major = lambda v, n_comp, n_snap: v.reshape(v.shape[0], n_comp, n_snap).permute(2, 0, 1).contiguous()
    return ops


@pytest.fixture
def export_mod(monkeypatch):
    import sparsespatialsampling_amd.export as export
    monkeypatch.setattr(export, "hipops", _cpu_ops())
    yield export


def _scube(tmp_path, d=2, nc=40):
    rng = np.random.default_rng(d)
    centers = rng.random((nc, d))
    return types.SimpleNamespace(n_dimensions=d, faces=pt.arange(nc * 2 ** d, dtype=pt.int32).reshape(nc, 2 ** d),
                                 centers=pt.from_numpy(centers), vertices=pt.from_numpy(rng.random((nc * 2 ** d, d))),
                                 levels=pt.ones((nc, 1), dtype=pt.int64), metric=None, size_initial_cell=2.5,
                                 save_path=str(tmp_path), save_name="case", grid_name="grid_s_cube")


def test_export_batches_layout_and_xdmf(export_mod, tmp_path):
    rng = np.random.default_rng(0)
    n, t_total = 500, 5
    coords = rng.random((n, 2))
    s = _scube(tmp_path)
    metric0 = rng.random(n)
    s.metric = pt.from_numpy(metric0)
    ex = export_mod.ExportData(s, write_times=[str(0.1 * i) for i in range(t_total)])
    p = rng.standard_normal((n, 1, t_total)).astype(np.float32)
    u = rng.standard_normal((n, 2, t_total)).astype(np.float32)
    # scalar field in two batches (3 + 2 snapshots), then a vector field at once
    ex.export(pt.from_numpy(coords), pt.from_numpy(p[:, :, :3]), "p", n_snapshots_total=t_total)
    assert ex._snapshot_counter == 3
    ex.export(pt.from_numpy(coords), pt.from_numpy(p[:, :, 3:]), "p", n_snapshots_total=t_total)
    assert ex._snapshot_counter == 0                       # finished -> state reset (reference export.py:302-319)
    ex.export(pt.from_numpy(coords), pt.from_numpy(u), "U")

    h5 = dump(os.path.join(str(tmp_path), "case.h5"))
    assert h5["grid/faces"].dtype == np.int32 and h5["constant/levels"].shape == (40, 1) and h5["constant/size_initial_cell"].shape == ()
    idx, dist = orc.knn(coords, s.centers.numpy(), 8)
    w = orc.idw_weights(dist)
    assert set(k for k in h5 if k.startswith("grid/")) == {"grid/faces", "grid/vertices", "grid/centers"}
    assert set(k for k in h5 if k.startswith("constant/")) == {"constant/levels", "constant/metric",
                                                               "constant/size_initial_cell"}
    np.testing.assert_allclose(h5["constant/metric"], orc.interp(w, idx, metric0), rtol=1e-13)
    ref_p, ref_u = orc.interp(w, idx, p), orc.interp(w, idx, u)
    for i in range(t_total):
        t = str(0.1 * i)
        assert h5[f"data/{t}/p_center"].shape == (40,)                       # scalars are squeezed (export.py:285-287)
        assert h5[f"data/{t}/U_center"].shape == (40, 2)
        np.testing.assert_allclose(h5[f"data/{t}/p_center"], ref_p[:, 0, i], rtol=1e-13)
        np.testing.assert_allclose(h5[f"data/{t}/U_center"], ref_u[:, :, i], rtol=1e-13)

    xdmf = open(os.path.join(str(tmp_path), "case.xdmf")).read()
    assert xdmf.startswith('<?xml version="1.0"?>\n<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>\n<Xdmf Version="2.0">\n<Domain>\n'
                           '<Grid Name="grid_s_cube" GridType="Collection" CollectionType="temporal">\n')
    assert xdmf.count('<Time Value=') == t_total and xdmf.endswith('</Grid>\n</Domain>\n</Xdmf>')
    assert '<Topology TopologyType="Quadrilateral" NumberOfElements="40">\n<DataItem Format="HDF" DataType="Int" ' \
           'Dimensions="40 4">\ncase.h5:/grid/faces\n' in xdmf
    assert xdmf.count('<Attribute Name="levels" AttributeType="Vector" Center="Cell">') == 1     # only in the first step
    assert '<Attribute Name="U" AttributeType="Vector" Center="Cell">\n<DataItem NumberType="Float" Precision="8" ' \
           'Format="HDF" Dimensions="40 2">\ncase.h5:/data/0.0/U_center\n</DataItem>\n</Attribute>\n' in xdmf


def test_export_errors(export_mod, tmp_path):
    s = _scube(tmp_path)
    s.metric = pt.zeros(10)
    ex = export_mod.ExportData(s)                         # write_times=None -> warning now, ValueError on export
    with pytest.raises(ValueError):
        ex.export(pt.zeros((10, 2)), pt.zeros((10, 1, 2)), "p")
    ex.write_times = "0.5"
    assert ex.write_times == ["0.5"]
    with pytest.raises(ValueError):
        ex.export(pt.rand((10, 2)), pt.zeros(10), "p")     # rank-1 data


def test_dataloader_roundtrip(export_mod, tmp_path):
    from sparsespatialsampling_amd.data import Dataloader, Datawriter
    rng = np.random.default_rng(1)
    wr = Datawriter(str(tmp_path), "g.h5")
    wr.write_data("centers", group="grid", data=rng.random((7, 3)))
    wr.write_data("vertices", group="grid", data=rng.random((20, 3)))
    wr.write_data("faces", group="grid", data=np.arange(56).reshape(7, 8))
    wr.write_data("levels", group="constant", data=np.array([[1], [2], [2], [3], [1], [1], [2]]))
    wr.write_data("size_initial_cell", group="constant", data=2.0)
    wr.n_cells = 7
    for t in ("0.1", "0.2"):
        wr.write_data("p", group="data", time_step=t, data=rng.random(7))          # suffix added: p_center
        wr.write_data("U", group="data", time_step=t, data=rng.random((7, 3)))
    wr.write_data("p", group="data", time_step="0.1", data=rng.random(7))          # duplicate -> warning, no crash
    wr.write_xdmf_file()
    ld = Dataloader(str(tmp_path), "g.h5")
    assert ld.write_times == ["0.1", "0.2"] and ld.field_names["0.1"] == ["U", "p"]      # HDF5 lists members in name order
    assert ld.load_snapshot("p").shape == (7, 2) and ld.load_snapshot("U").shape == (7, 3, 2)
    assert pt.allclose(ld.weights, (2.0 / 2.0 ** ld.levels.double()) ** 3)
    assert 'TopologyType="Hexahedron"' in open(os.path.join(str(tmp_path), "g.xdmf")).read()


def test_reference_fixture_file():
    """the reference's own test file through this package's loader: the expectations of the reference's
    tests/test_s_cube_dataloader.py:40-57 (209 cells, 247 nodes, 2-D, one write time with the field p)"""
    from sparsespatialsampling_amd.data import Dataloader, XDMFWriter
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    ld = Dataloader(here, "s_cube_test_dataset.h5")
    assert len(ld.write_times) == 1 and ld.write_times == ["0.4"] and ld.field_names == {"0.4": ["p"]}
    assert ld.vertices.shape == (209, 2) and ld.weights.shape == ld.levels.shape and ld.faces.shape == (209, 4)
    assert ld.nodes.shape == (247, 2) and ld.load_snapshot("p", "0.4").shape == (209, 1)
    assert ld.load_snapshot("p", "0.4").dtype == pt.float32 and ld.metric.shape == (209,)
    # cells tile their nodes: centre = mean of the four corner nodes of its face
    corners = ld.nodes[ld.faces.long()]
    assert pt.allclose(corners.mean(1), ld.vertices, atol=1e-12)
    # the areas follow from the levels and the initial cell size
    edge = corners.max(1).values - corners.min(1).values
    assert pt.allclose(edge.prod(1), ld.weights.to(edge.dtype), rtol=1e-6)


def test_reference_fixture_roundtrip(tmp_path):
    """copy the reference's fixture through Dataloader -> Datawriter (write_grid + constants + data) and read it back:
    same arrays, and an XDMF file that names every dataset"""
    from sparsespatialsampling_amd.data import Dataloader, Datawriter
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    src = Dataloader(here, "s_cube_test_dataset.h5", dtype=pt.float64)
    wr = Datawriter(str(tmp_path), "copy.h5")
    wr.write_grid(src)
    wr.write_data("levels", group="constant", data=src.levels.unsqueeze(-1))
    wr.write_data("metric", group="constant", data=src.metric)
    wr.write_data("size_initial_cell", group="constant", data=float(src._size_initial_cell))
    wr.write_data("p", group="data", time_step="0.4", data=src.load_snapshot("p", "0.4").squeeze(-1))   # -> p_center
    with pytest.raises(ValueError):
        wr.write_data("x", group="nonsense", data=np.zeros(3))
    wr.write_xdmf_file()
    a, b = dump(os.path.join(here, "s_cube_test_dataset.h5")), dump(os.path.join(str(tmp_path), "copy.h5"))
    for key in ("grid/centers", "grid/vertices", "grid/faces", "constant/levels", "constant/metric", "data/0.4/p_center"):
        assert np.array_equal(np.squeeze(a[key]), np.squeeze(b[key])), key
    xdmf = open(os.path.join(str(tmp_path), "copy.xdmf")).read()
    assert "copy.h5:/data/0.4/p_center" in xdmf and 'NumberOfElements="209"' in xdmf and 'Dimensions="247 2"' in xdmf


def test_async_batches_and_duplicates(tmp_path):
    """the background writer: batches queued from snapshot-major buffers, a buffer reused only after wait_buffer, datasets
    that exist already skipped and counted"""
    from sparsespatialsampling_amd.data import Datawriter
    rng = np.random.default_rng(0)
    wr = Datawriter(str(tmp_path), "b.h5")
    bufs = [pt.empty((6, 50, 3), dtype=pt.float64), pt.empty((6, 50, 3), dtype=pt.float64)]
    want = {}
    for batch in range(5):
        buf = bufs[batch % 2]
        wr.wait_buffer(buf)                                        # the writer may still be storing batch - 2 from it
        buf.copy_(pt.from_numpy(rng.random((6, 50, 3))))
        times = [f"{batch}.{i}" for i in range(6)]
        wr.write_snapshots("U_center", times, buf)
        for i, t in enumerate(times):
            want[t] = buf[i].clone().numpy()
    wr.write_snapshots("U_center", ["0.0", "9.9"], bufs[0][:2].contiguous())     # the first exists already
    assert wr._file.flush() == 1
    wr.close()
    got = dump(os.path.join(str(tmp_path), "b.h5"))
    assert len(got) == 31
    for t, v in want.items():
        assert np.array_equal(got[f"data/{t}/U_center"], v)


@pytest.mark.parametrize("big", [False, True])
def test_writer_waits_for_the_batch_to_land(tmp_path, monkeypatch, big):
    """``write_snapshots(ready=(flag, value))`` (s3h5_write_snapshots_async_when): the batch is handed to the background writer
    while its values are still on their way into the buffer -- ExportData queues the device-to-host copy and hands over at once
    (reference export.py:233-319 writes after the interpolation returned) -- and nothing is read before the producer's word
    says so.  Here a thread plays the copy: it fills the buffer late and raises the word; the file must hold the final values.
    A word that never comes ends in an error at flush, not in a hang."""
    import threading
    import time
    if h5io.native_lib() is None:
        pytest.skip("needs the native sink")
    t, n = 4, (300_000 if big else 500)                    # big: datasets of a megabyte or more take the raw-write path
    path = str(tmp_path / "late.h5")
    buf = np.full((t, n), -1.0)
    flag = pt.zeros(1, dtype=pt.int32)
    final = np.arange(t * n, dtype=np.float64).reshape(t, n)

    def copy_arrives():
        time.sleep(0.3)
        buf[:] = final
        flag[0] = 7

    with h5io.open_h5(path, "w") as f:
        worker = threading.Thread(target=copy_arrives)
        worker.start()
        f.write_snapshots([str(i) for i in range(t)], "p_center", buf, ready=(flag, 7))
        assert f.flush() == 0
        worker.join()
    with h5io.open_h5(path, "r") as f:
        for i in range(t):
            assert np.array_equal(f.read(f"data/{i}/p_center"), final[i])
    # the word never comes
    monkeypatch.setenv("S3H5_READY_TIMEOUT_S", "0.5")
    code = ("import sys, numpy as np, torch as pt; sys.path.insert(0, %r); from sparsespatialsampling_amd import h5io\n"
            "f = h5io.open_h5(%r, 'w'); f.write_snapshots(['0'], 'p', np.zeros((1, 10)), ready=(pt.zeros(1, dtype=pt.int32), 1))\n"
            "try:\n    f.flush(); print('no error')\nexcept h5io.H5Error as e:\n    print('error:', e)\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "never.h5")))
    import subprocess, sys
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, S3H5_READY_TIMEOUT_S="0.5"))
    assert "error:" in out.stdout and "never arrived" in out.stdout, out.stdout + out.stderr


def test_snapshot_pieces_follow_chunk_size(export_mod, tmp_path, monkeypatch):
    """``chunk_size`` (which bounds the reference's [chunk, k, n_comp, T] temporary, export.py:463-467) sets the length of the
    snapshot pieces a batch is pipelined in: about chunk_size * 256 output values each, at least 32 snapshots, multiples of 8,
    covering the batch without gaps; host batches are cut on request only (S3_EXPORT_PIPELINE=host)"""
    s = _scube(tmp_path, nc=400)
    ex = export_mod.ExportData(s, write_times=["0"])
    d = pt.zeros((10, 1, 200), dtype=pt.float32)
    assert ex._snapshot_pieces(d, 1, 200) == [(0, 200)]                       # a host batch, default: one piece
    monkeypatch.setenv("S3_EXPORT_PIPELINE", "host")
    for chunk, t, ncomp in ((100, 200, 1), (100, 1000, 1), (50, 130, 3), (1000, 257, 1), (5, 64, 1), (100, 63, 1)):
        ex._chunk_size = chunk
        pieces = ex._snapshot_pieces(pt.zeros((10, ncomp, t), dtype=pt.float32), ncomp, t)
        assert pieces[0][0] == 0 and pieces[-1][1] == t and all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
        want = max(32, chunk * 256 // (400 * ncomp) // 8 * 8)
        if t < 64 or want >= t:
            assert pieces == [(0, t)]
        else:
            lengths = [b - a for a, b in pieces]
            assert all(n % 8 == 0 for n in lengths[:-1]) and len(pieces) in (-(-t // want), -(-t // want) - 1)
            assert max(lengths) <= want + 8 + 15 and min(lengths) >= 16           # (a tail of < 16 snapshots joins the piece before it)
    # the tails the rounding to multiples of 8 used to leave: 97 = 32 + 32 + 32 + 1, 129 = 4 x 32 + 1
    ex._chunk_size = 50
    for t in (97, 98, 99, 129, 130, 161, 1001):
        pieces = ex._snapshot_pieces(pt.zeros((10, 1, t), dtype=pt.float32), 1, t)
        assert pieces[0][0] == 0 and pieces[-1][1] == t and all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
        assert min(b - a for a, b in pieces) >= 16, (t, pieces)
    monkeypatch.setenv("S3_EXPORT_PIPELINE", "0")
    assert ex._snapshot_pieces(pt.zeros((10, 1, 1000), dtype=pt.float32), 1, 1000) == [(0, 1000)]


def test_append_and_file_per_field(export_mod, tmp_path):
    """the export state machine's other modes on real files: ``append_existing`` adds a field to a finished file (grid and
    constants are not rewritten, reference export.py:86-92), ``write_new_file_for_each_field`` gives one file per field"""
    rng = np.random.default_rng(3)
    n, t_total = 300, 4
    coords = rng.random((n, 2))
    times = [str(0.5 * i) for i in range(t_total)]
    p = rng.standard_normal((n, 1, t_total)).astype(np.float32)
    u = rng.standard_normal((n, 2, t_total)).astype(np.float32)
    idx, dist = orc.knn(coords, _scube(tmp_path).centers.numpy(), 8)
    w = orc.idw_weights(dist)

    s = _scube(tmp_path)
    s.metric = pt.from_numpy(rng.random(n))
    export_mod.ExportData(s, write_times=times).export(pt.from_numpy(coords), pt.from_numpy(p), "p")
    s2 = _scube(tmp_path)
    s2.metric = pt.from_numpy(rng.random(n))
    ex = export_mod.ExportData(s2, write_times=times, append_existing=True)
    ex.export(pt.from_numpy(coords), pt.from_numpy(u[:, :, :2]), "U", n_snapshots_total=t_total)
    ex.export(pt.from_numpy(coords), pt.from_numpy(u[:, :, 2:]), "U", n_snapshots_total=t_total)
    h5 = dump(os.path.join(str(tmp_path), "case.h5"))
    ref_p, ref_u = orc.interp(w, idx, p), orc.interp(w, idx, u)
    for i, t in enumerate(times):
        np.testing.assert_allclose(h5[f"data/{t}/p_center"], ref_p[:, 0, i], rtol=1e-13)
        np.testing.assert_allclose(h5[f"data/{t}/U_center"], ref_u[:, :, i], rtol=1e-13)
    assert "case.h5:/data/1.5/U_center" in open(os.path.join(str(tmp_path), "case.xdmf")).read()

    s3 = _scube(tmp_path)
    s3.save_name, s3.metric = "split", pt.from_numpy(rng.random(n))
    ex = export_mod.ExportData(s3, write_times=times, write_new_file_for_each_field=True)
    ex.export(pt.from_numpy(coords), pt.from_numpy(p), "p")
    ex.export(pt.from_numpy(coords), pt.from_numpy(u), "U")
    for field in ("p", "U"):
        part = dump(os.path.join(str(tmp_path), f"split_{field}.h5"))
        assert "grid/faces" in part and f"data/0.0/{field}_center" in part and len([k for k in part if k.startswith("data/")]) == t_total
        assert os.path.exists(os.path.join(str(tmp_path), f"split_{field}.xdmf"))


@pytest.mark.parametrize("raw", ["1", "0"])
def test_large_snapshots_take_the_raw_write_path_and_read_back(tmp_path, monkeypatch, raw):
    """(raw = "0": S3_H5_RAW_WRITES=0 sends everything through H5Dwrite -- the switch for files that are not plain POSIX files)
    datasets of a megabyte or more are created by HDF5 (space allocated at creation) and their values written by several
    threads straight into the file at the offsets HDF5 reports: the file must read back -- through the library -- exactly
    as written, also mixed with small datasets, a second batch, an existing dataset in the batch and a re-opened file"""
    from sparsespatialsampling_amd import h5io
    if h5io.native_lib() is None:
        pytest.skip("native HDF5 sink not built")
    monkeypatch.setenv("S3_H5_RAW_WRITES", raw)
    rng = np.random.default_rng(11)
    path = os.path.join(str(tmp_path), "raw.h5")
    n = 300_001                                               # 2.4 MB per scalar snapshot, not a multiple of anything
    a = rng.standard_normal((5, n))
    b = rng.standard_normal((3, n, 2)).astype(np.float32)     # 2.4 MB per snapshot, two components
    small = rng.standard_normal((4, 100))
    with h5io.open_h5(path, "w") as f:
        f.write("grid/centers", rng.random((n, 2)))
        f.write_snapshots([f"{0.1 * i:.1f}" for i in range(5)], "p_center", a)
        f.write_snapshots([f"{0.1 * i:.1f}" for i in range(3)], "U_center", b)
        f.write_snapshots([f"{0.1 * i:.1f}" for i in range(4)], "q_center", small)
        assert f.flush() == 0
        f.write_snapshots(["0.4", "0.5"], "p_center", a[:2].copy())            # "0.4" exists: skipped, "0.5" is new
        assert f.flush() == 1
        assert np.array_equal(f.read("data/0.3/p_center"), a[3])               # readable while the file is still open
    with h5io.open_h5(path, "r") as f:
        for i in range(5):
            assert np.array_equal(f.read(f"data/{0.1 * i:.1f}/p_center"), a[i])
        for i in range(3):
            got = f.read(f"data/{0.1 * i:.1f}/U_center")
            assert got.dtype == np.float32 and np.array_equal(got, b[i])
        for i in range(4):
            assert np.array_equal(f.read(f"data/{0.1 * i:.1f}/q_center"), small[i])
        assert np.array_equal(f.read("data/0.5/p_center"), a[1])
    with h5io.open_h5(path, "a") as f:                                         # append to the finished file
        f.write_snapshots(["0.6"], "p_center", a[4:5].copy())
    with h5io.open_h5(path, "r") as f:
        assert np.array_equal(f.read("data/0.6/p_center"), a[4]) and np.array_equal(f.read("data/0.0/p_center"), a[0])


def test_export_openfoam_fields_batches_like_the_reference(export_mod, tmp_path):
    """the batching wrapper (reference utils.py:155-226): the reader is asked once for the field names / the write times that
    were not given, every field goes through ``export()`` in batches of ``batch_size`` snapshots with the total announced,
    a field the reader does not find is skipped, and the file equals the one a single call per field produces"""
    from sparsespatialsampling_amd import utils
    from sparsespatialsampling_amd.data import Dataloader
    rng = np.random.default_rng(11)
    n, t_all = 300, 7
    times = [f"{0.1 * (i + 1):.1f}" for i in range(t_all)]
    x = rng.random((n, 2))
    store = {"p": rng.standard_normal((n, 1, t_all)).astype(np.float32), "U": rng.standard_normal((n, 3, t_all)).astype(np.float32)}
    calls = []

    def reader(path, n_dims, bounds, field_names=None, write_times=None, get_field_names_and_times=False):
        calls.append((field_names, None if write_times is None else tuple(write_times), get_field_names_and_times))
        assert path == "case" and n_dims == 2 and bounds == [[0, 0], [1, 1]]
        if get_field_names_and_times:
            return list(times), ["p", "U", "ghost"]
        if field_names not in store:
            return None, None
        cols = [times.index(t) for t in write_times]
        return pt.from_numpy(x), pt.from_numpy(store[field_names][:, :, cols])

    def run(name, batch_size, fields):
        sc = _scube(tmp_path / name, d=2, nc=40)
        sc.metric = pt.from_numpy(np.linspace(0.0, 1.0, n))
        os.makedirs(sc.save_path, exist_ok=True)
        ex = export_mod.ExportData(sc, write_times=None)
        utils.export_openfoam_fields(ex, "case", [[0, 0], [1, 1]], batch_size=batch_size, fields=fields, loader=reader)
        return sc

    calls.clear()
    sc = run("batched", 3, None)
    meta = [c for c in calls if c[2]]
    assert len(meta) == 2                                       # once for the field names, once for the write times
    loads = [c for c in calls if not c[2]]
    assert [c[1] for c in loads if c[0] == "p"] == [tuple(times[0:3]), tuple(times[3:6]), tuple(times[6:7])]
    assert [c[0] for c in loads].count("ghost") == 3            # asked for, not found, skipped
    one = run("single", None, ["p", "U"])
    a = dump(os.path.join(sc.save_path, sc.save_name + ".h5"))
    b = dump(os.path.join(one.save_path, one.save_name + ".h5"))
    assert a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)
    loaded = Dataloader(sc.save_path, sc.save_name + ".h5")
    assert sorted(loaded.write_times, key=float) == times and loaded.load_snapshot("U", times).shape[-1] == t_all
    with pytest.raises(ValueError):
        utils.export_openfoam_fields(export_mod.ExportData(_scube(tmp_path / "bad"), write_times=times), "case", [[0, 0], [1, 1]],
                                     batch_size=0, fields="p", loader=reader)
    with pytest.raises(ImportError, match="flowtorch"):
        utils.export_openfoam_fields(export_mod.ExportData(_scube(tmp_path / "nofoam"), write_times=times), "case",
                                     [[0, 0], [1, 1]], fields="p")


def test_tke_metric_formula():
    """examples/s3_for_cylinder3D_Re3900.py:104"""
    from sparsespatialsampling_amd import metrics
    t = pt.from_numpy(np.random.default_rng(2).random((50, 6)))
    assert pt.equal(metrics.tke_from_uprime2mean(t), 0.5 * t[:, [0, 3, 5]].sum(-1))
    with pytest.raises(ValueError):
        metrics.tke_from_uprime2mean(t[:, :5])
