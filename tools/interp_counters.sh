#!/bin/bash
# Runs ON THE GPU BOX from the repo root: wave-state / instruction counters of the interpolation kernel of a bench workload
# (rocprofv3 --pmc, one pass per counter pair, kernel-trace only).  Output: gpurun_out/interp_counters_<name>.csv
#   gpurun -- 'bash tools/interp_counters.sh cylinder3D'      gpurun -- 'bash tools/interp_counters.sh box5e7 --workload box5e7'
set -o pipefail
name=${1:-cylinder3D}; shift
root=$(pwd); out=$root/gpurun_out/interp_pmc; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
csv="$root/gpurun_out/interp_counters_$name.csv"
echo "Kernel_Name,Counter_Name,mean_per_launch,launches" > "$csv"
for group in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVES" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
    gname=$(echo $group | tr ' ' '_')
    if rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$out/x_$gname" -- python "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$out/x_$gname.log" 2>&1; then
        f=$(find "$out/x_$gname" -name "*counter_collection.csv" | head -n 1)
        python - "$f" "$csv" <<'PY'
import csv, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "interp_planned" in r["Kernel_Name"]:
        acc.setdefault((r["Kernel_Name"].split("(")[0][:48], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(sys.argv[2], "a") as f:
    for (k, c), v in acc.items():
        f.write('"%s",%s,%f,%d\n' % (k, c, sum(v) / len(v), len(v)))
PY
    else
        echo "group $group failed"; tail -3 "$out/x_$gname.log"
    fi
    rm -rf "$out/x_$gname"
done
cd "$root"; cat "$csv"
