#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats of one bench run (4 refines + a few interpolation steps) of a workload
# -> gpurun_out/refine_<workload>_kernel_stats.csv
set -o pipefail
w=${1:-cylinder3D_Re3900}
root=$(pwd); out=$root/gpurun_out; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/rks_$w" -- python3 "$root/bench.py" --workload "$w" --no-cpu-baseline --no-batches --steps 2 --warmup 1 > "$out/rks_$w.log" 2>&1 || exit 1
find "$out/rks_$w" -name "*kernel_stats.csv" -exec cp {} "$out/refine_${w}_kernel_stats.csv" \; ; rm -rf "$out/rks_$w"
python3 - "$out/refine_${w}_kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms" % (tot / 1e6))
for r in rows[:22]:
    print("%8.2f ms %6s calls %9.1f us avg  %s" % (int(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, re.sub(r"\(.*", "", r["Name"])[:90]))
PY
