#!/bin/bash
# GPU box: duration (µs) of every child_metric* launch, in launch order, of bench.py's refine of one workload
# (rocprofv3 --kernel-trace) -> gpurun_out/knn_batches_<workload>.txt; and the refine wall-clock with the wavefront kernels
# on (default) and off (S3_KNN_COOP=0), no profiler -> gpurun_out/knn_wall_<workload>.txt
set -o pipefail
w=${1:-cylinder3D_Re3900}
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/kb_$w" -- python3 "$root/bench.py" --workload "$w" --no-cpu-baseline --no-batches --steps 2 --warmup 1 > "$out/kb_$w.log" 2>&1 || exit 1
f=$(find "$out/kb_$w" -name "*kernel_trace.csv" | head -n 1)
python3 - "$f" > "$out/knn_batches_$w.txt" <<'PY'
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "child_metric" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
line = []
for r in rows:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void s3::", "").replace("child_metric_", "")
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if name.startswith("coop") or name.startswith("kernel"):
        if line: print("  ".join(line))
        line = []
    line.append("%s %.0f" % (name, us))
if line: print("  ".join(line))
PY
rm -rf "$out/kb_$w"
cd "$root"
for c in "" 0; do
    S3_KNN_COOP=$c python3 bench.py --workload "$w" --no-cpu-baseline --no-batches --steps 2 --warmup 1 2>/dev/null \
        | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S3_KNN_COOP=%r' % '$c', d['refine_wall_s'], d['refine_runs_s'])" || exit 1
done > "$out/knn_wall_$w.txt"
cat "$out/knn_wall_$w.txt"
