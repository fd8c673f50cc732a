#!/bin/bash
# Runs ON THE GPU BOX from the repo root: wave-state / instruction counters of the KNN kernels (tools/knn_probe.py), one
# rocprofv3 --pmc run per counter pair (kernel-trace only).  Output: gpurun_out/knn_counters.csv
set -o pipefail
root=$(pwd); out=$root/gpurun_out/knn_pmc; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
echo "Kernel_Name,Counter_Name,mean_per_launch,launches" > "$root/gpurun_out/knn_counters.csv"
for group in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVES" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM"; do
    gname=$(echo $group | tr ' ' '_')
    if rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$out/x_$gname" -- python "$root/tools/knn_probe.py" > "$out/x_$gname.log" 2>&1; then
        f=$(find "$out/x_$gname" -name "*counter_collection.csv" | head -n 1)
        python - "$f" "$root/gpurun_out/knn_counters.csv" <<'PY'
import csv, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "idw_predict" in r["Kernel_Name"] or "knn_query" in r["Kernel_Name"]:
        acc.setdefault((r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(sys.argv[2], "a") as f:
    for (k, c), v in acc.items():
        f.write('"%s",%s,%f,%d\n' % (k, c, sum(v) / len(v), len(v)))
PY
    else
        echo "group $group failed"; tail -3 "$out/x_$gname.log"
    fi
    rm -rf "$out/x_$gname"
done
cd "$root"; cat gpurun_out/knn_counters.csv
