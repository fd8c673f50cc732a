"""dev helper (development container: needs /root/reference): randomly drawn exports run by the REAL reference (own process; h5py = the
stand-in of tests/golden/h5py_standin.py) and by this package's ExportData host logic, compared file by file -- same datasets in the
same order, shapes, dtypes, grid bit-identical, values 1e-12, XDMF byte-identical (tests/test_export_vs_reference.py has the machinery)
    python tools/fuzz_export_vs_reference.py [first seed] [number of seeds]"""
import os, sys, tempfile, shutil, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import __graft_entry__ as entry
entry.build_oracle(); entry.build_h5()
from tests.test_export_vs_reference import fuzz_export_against_reference
seed0, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
t0, done = time.time(), 0
for s in range(seed0, seed0 + n, 5):
    d = tempfile.mkdtemp(prefix="s3_fuzz_export_")
    try:
        done += fuzz_export_against_reference(d, s, min(5, seed0 + n - s))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    print(f"{done} cases equal the reference ({time.time() - t0:.0f} s)", flush=True)
