"""The C-ABI library loads without a GPU and exports every symbol include/s3hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "s3hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(s3_[a-z0-9_]+)\s*\(", text)))


def header_abi_version():
    m = re.search(r"#define\s+S3_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "s3hip.h")).read())
    return int(m.group(1))


def test_stale_library_is_refused_with_a_rebuild_hint(monkeypatch):
    """a libs3hip.so built from another revision of the header must not be used silently"""
    from sparsespatialsampling_amd import _lib
    _lib.hip_lib()
    monkeypatch.setattr(_lib, "_hip", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.HipUnavailableError, match="rebuild"):
        _lib.hip_lib()
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION - 1)
    monkeypatch.setattr(_lib, "_hip", None)
    sigs = dict(_lib.HIP_SIGNATURES)
    sigs["s3_no_such_symbol"] = (None, [])
    monkeypatch.setattr(_lib, "HIP_SIGNATURES", sigs)
    with pytest.raises(_lib.HipUnavailableError, match="s3_no_such_symbol"):
        _lib.hip_lib()


def test_header_symbols_exported_and_bound():
    from sparsespatialsampling_amd import _lib
    lib = _lib.hip_lib()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libs3hip.so does not export {n}"
        assert n in _lib.HIP_SIGNATURES, f"{n} has no ctypes prototype in _lib.py"
    assert set(_lib.HIP_SIGNATURES) == set(names)
    assert lib.s3_abi_version() == _lib.ABI_VERSION == header_abi_version()


def test_h5_sink_symbols_exported_and_bound():
    """libs3h5.so (HDF5 sink) exports what include/s3h5.h declares and h5io binds all of it"""
    from sparsespatialsampling_amd import h5io
    lib = h5io.native_lib()
    if lib is None:
        pytest.skip("HDF5 C library not available on this machine: libs3h5.so not built")
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "s3h5.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(s3h5_[a-z0-9_]+)\s*\(", text)))
    assert len(names) >= 12 and set(names) == set(h5io.H5_SIGNATURES)
    for n in names:
        assert hasattr(lib, n), f"libs3h5.so does not export {n}"
    major = ctypes.c_uint(0)
    assert lib.s3h5_version(ctypes.byref(major), ctypes.byref(ctypes.c_uint(0)), ctypes.byref(ctypes.c_uint(0))) == 0 and major.value == 1


def test_no_torch_types_in_abi():
    text = open(os.path.join(ROOT, "include", "s3hip.h")).read() + open(os.path.join(ROOT, "include", "s3h5.h")).read()
    code = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                # declarations only, comments stripped
    assert "torch" not in code.lower() and "at::" not in code and "std::" not in code and "Tensor" not in code


def test_fails_loudly_without_device():
    """no GPU here: every path into the hot path must raise, never fall back to a CPU computation"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sparsespatialsampling_amd import _lib, geometry, hipops
    from sparsespatialsampling_amd.s_cube import SamplingTree
    with pytest.raises(_lib.HipUnavailableError):
        hipops.device()
    x = torch.rand(100, 2)
    with pytest.raises(_lib.HipUnavailableError):
        SamplingTree(x, torch.ones(100), [geometry.CubeGeometry("d", True, [0, 0], [1, 1])], uniform_level=1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sparsespatialsampling_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle" not in src.lower(), f


def test_topology_tables_selfcheck():
    from sparsespatialsampling_amd import _lib
    assert _lib.topo_lib().s3t_selfcheck(2) == 0 and _lib.topo_lib().s3t_selfcheck(3) == 0


def test_header_is_plain_c():
    """include/s3hip.h and the torch-free host in tests/native compile as C11"""
    import subprocess
    src = os.path.join(ROOT, "tests", "native", "c_host.c")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src],
                   check=True)


def build_c_host(tmp_path):
    import subprocess
    pkg = os.path.join(ROOT, "sparsespatialsampling_amd")
    exe = str(tmp_path / "c_host")
    subprocess.run(["gcc", "-std=c11", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "c_host.c"),
                    "-o", exe, "-L", pkg, "-ls3hip", "-lm", f"-Wl,-rpath,{pkg}"], check=True)
    return exe


@pytest.mark.gpu
def test_c_host_without_torch(tmp_path):
    """a plain-C program drives KNN -> weights -> direct and planned interpolation through the C ABI (own HIP runtime, no
    torch in the process; its host arrays are pageable: s3_memcpy_* stage them) and checks the results against a scalar loop.
    (The weighted-SVD chain of the same program: tests/test_gpu_kernels.py::test_c_host_svd_chain_without_torch.)"""
    import subprocess
    run = subprocess.run([build_c_host(tmp_path), "nosvd"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "mismatches 0" in run.stdout and "SVD chain not asked for" in run.stdout
