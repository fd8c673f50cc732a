#!/bin/bash
# Runs ON THE GPU BOX from the repo root: extra PMC passes for the dominant kernel (cache and LDS behaviour), one
# rocprofv3 --pmc run per counter group (kernel-trace only).  Output: gpurun_out/prof_<tag>/pmc_extra.csv
#   gpurun --timeout 900 -- 'bash tools/collect_counters.sh r02 cylinder3D [bench args]'
set -o pipefail
tag=${1:-r02}; name=${2:-cylinder3D}; shift 2
root=$(pwd)
out=$root/gpurun_out/prof_$tag/$name
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
echo "Kernel_Name,Counter_Name,Counter_Value" > "$out/pmc_extra.csv"
for group in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VALU"; do
    gname=$(echo $group | tr ' ' '_')
    if rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$out/x_$gname" -- python "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$out/x_$gname.log" 2>&1; then
        f=$(find "$out/x_$gname" -name "*counter_collection.csv" | head -n 1)
        python - "$f" "$out/pmc_extra.csv" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "interp_planned" in r["Kernel_Name"]]
acc = {}
for r in rows:
    acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open(sys.argv[2], "a") as f:
    for k, v in acc.items():
        f.write('"%s (mean of %d launches)",%s,%f\n' % (rows[0]["Kernel_Name"].split("(")[0][:60], len(v), k, sum(v) / len(v)))
PY
    else
        echo "group '$group' not collectable on this box" >> "$out/pmc_extra.notes"
        tail -3 "$out/x_$gname.log" >> "$out/pmc_extra.notes"
    fi
    rm -rf "$out/x_$gname" "$out/x_$gname.log"
done
cd "$root"
cat "$out/pmc_extra.csv"
