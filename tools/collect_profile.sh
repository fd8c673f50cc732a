#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: the rocprofv3 passes behind profiles/<tag>/<name>/.
#   gpurun --timeout 900 -- 'bash tools/collect_profile.sh r02 cylinder3D'                        (default bench command)
#   gpurun --timeout 900 -- 'bash tools/collect_profile.sh r02 box5e7 --workload box5e7'          (any bench arguments)
# 1) bench.py as the driver runs it (JSON line), 2) --kernel-trace --stats of the same command, 3) one --pmc pass per
# counter (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as gpurun requires).  Raw output -> gpurun_out/prof_<tag>/<name>/;
# tools/summarize_profile.py turns it into profiles/<tag>/<name>/ afterwards (in the dev container).
set -o pipefail
tag=${1:-r02}; name=${2:-cylinder3D}; shift 2
root=$(pwd)
# one gpurun call = one lease of a box: S3_LEASE=<n> keeps the collections of several leases apart (.../<name>/lease<n>/),
# tools/summarize_profile.py lists all of them in summary.json
out=$root/gpurun_out/prof_$tag/$name${S3_LEASE:+/lease$S3_LEASE}
mkdir -p "$out"
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 ${S3_BENCH_FAST:+--no-cpu-baseline --no-batches --no-traffic} "$@" > "$out/bench.json" 2> "$out/bench.err" || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-batches --no-pitched-copy --no-traffic "$@" > "$out/stats.log" 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_$c" -- python "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-batches --no-pitched-copy --no-traffic "$@" > "$out/pmc_$c.log" 2>&1 || exit 1
done
cd "$root"
# keep what travels back small: per-kernel stats + the counter rows of the interpolation kernels
find "$out" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
for c in FETCH_SIZE WRITE_SIZE; do
    f=$(find "$out/pmc_$c" -name "*counter_collection.csv" | head -n 1)
    python - "$f" "$out/pmc_$c.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
with open(sys.argv[2], "w") as f:
    f.write("Kernel_Name,Counter_Name,Counter_Value\n")
    for r in rows:
        if "interp" in r["Kernel_Name"]:
            f.write('"%s",%s,%f\n' % (r["Kernel_Name"][:80], r["Counter_Name"], float(r["Counter_Value"])))
PY
done
rm -rf "$out/stats" "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE"
ls -la "$out"
