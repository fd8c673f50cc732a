"""CPU-side sanitizer job (no GPU sanitizer exists on the pool): the host-side native code -- the transfer lanes of libs3hip.so
(csrc/host_lanes.h), the topology engine + CPython-set restatement (libs3topo.so) and the HDF5 sink (libs3h5.so) -- built with
-fsanitize=thread / address,undefined and driven by their own tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sparsespatialsampling_amd", "csrc")


def _runtime(name):
    path = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        pytest.skip(f"{name} is not installed with this gcc")
    return path


@pytest.mark.parametrize("flags,env", [(["-fsanitize=thread", "-DLANES_NO_FORK"], {"TSAN_OPTIONS": "halt_on_error=1"}),
                                       (["-fsanitize=address,undefined", "-fno-sanitize-recover=all"],
                                        {"ASAN_OPTIONS": "detect_leaks=0", "UBSAN_OPTIONS": "print_stacktrace=1"})])
def test_transfer_lanes_under_sanitizers(tmp_path, flags, env):
    """LanePool (jobs of changing width, shutdown + reuse, a forked child) and StreamPacker under ThreadSanitizer and under
    AddressSanitizer + UBSan"""
    _runtime("libtsan.so" if "thread" in flags[0] else "libasan.so")
    exe = str(tmp_path / "lanes_test")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g"] + flags + ["-I", CSRC, os.path.join(ROOT, "tests", "native", "lanes_test.cpp"), "-o", exe,
                    "-lpthread"], check=True)
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert run.returncode == 0 and "lanes_test ok" in run.stdout, run.stdout[-2000:] + run.stderr[-4000:]


def test_topology_engine_and_hdf5_sink_under_asan_ubsan(tmp_path):
    """libs3topo.so (topology.cpp + pyset.cpp) and libs3h5.so (h5sink.cpp) rebuilt with -fsanitize=address,undefined and their own
    test files run against those builds (S3_TOPO_SO / S3_H5_SO; the ASan runtime preloaded into the interpreter)"""
    asan = _runtime("libasan.so")
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    topo = str(tmp_path / "libs3topo_asan.so")
    subprocess.run(["g++"] + flags + ["-ffp-contract=off", "-o", topo, os.path.join(CSRC, "topology.cpp"), os.path.join(CSRC, "pyset.cpp")], check=True)
    env = dict(os.environ, S3_TOPO_SO=topo, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:handle_segv=0",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    files = ["tests/test_pyset.py", "tests/test_topology_parallel.py"]
    prefix = os.environ.get("S3_HDF5_PREFIX", "/opt/conda")
    if os.path.exists(os.path.join(prefix, "include", "hdf5.h")):
        h5 = str(tmp_path / "libs3h5_asan.so")
        # (libhdf5 by its path, not through -L: the HDF5 prefix of the image carries an older libasan / libubsan of its own, and
        # a second sanitizer runtime in the process aborts at start-up)
        subprocess.run(["g++"] + flags + ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(prefix, "include"), "-o", h5,
                        os.path.join(CSRC, "h5sink.cpp"), os.path.join(prefix, "lib", "libhdf5.so"),
                        f"-Wl,-rpath,{os.path.join(prefix, 'lib')}", "-lpthread"], check=True)
        env["S3_H5_SO"] = h5
        files.append("tests/test_export_host_logic.py")
    # (the three long random traces stay out: 13-15 s each natively, several times that under the sanitizer; the shorter traces and
    # the refine-like ones walk the same code)
    skip = [a for seed in (0, 9, 11) for a in ("--deselect", f"tests/test_pyset.py::test_random_traces_slot_for_slot[{seed}]")]
    run = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + skip + files, cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=1500)
    tail = run.stdout[-3000:] + run.stderr[-3000:]
    assert run.returncode == 0, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error:" not in run.stderr, tail
