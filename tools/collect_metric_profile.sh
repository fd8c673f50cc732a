#!/bin/bash
# Runs ON THE GPU BOX from the repo root: rocprofv3 evidence for the temporal-moments kernel (s3_row_moments) on the
# matrices of tools/metric_probe.py -> gpurun_out/prof_<tag>/metric_kernel_stats.csv, metric_fetch.csv
set -o pipefail
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/m_stats" -- python "$root/tools/metric_probe.py" > "$out/metric_probe.log" 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/m_pmc" -- python "$root/tools/metric_probe.py" > "$out/metric_pmc.log" 2>&1 || exit 1
cd "$root"
f=$(find "$out/m_stats" -name "*kernel_stats.csv" | head -n 1)
head -n 1 "$f" > "$out/metric_kernel_stats.csv"
grep "row_moments" "$f" >> "$out/metric_kernel_stats.csv"
g=$(find "$out/m_pmc" -name "*counter_collection.csv" | head -n 1)
python - "$g" "$out/metric_fetch.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "row_moments" in r["Kernel_Name"]]
acc = {}
for r in rows:
    acc.setdefault(r["Kernel_Name"][:80], []).append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    f.write("Kernel_Name,launches,FETCH_SIZE_KB_mean_per_launch\n")
    for k, v in acc.items():
        f.write('"%s",%d,%f\n' % (k, len(v), sum(v) / len(v)))
PY
rm -rf "$out/m_stats" "$out/m_pmc"
grep -v "^/opt" "$out/metric_probe.log" | grep "^N="
cat "$out/metric_kernel_stats.csv" | cut -c1-220
cat "$out/metric_fetch.csv"
