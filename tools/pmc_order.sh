#!/bin/bash
# Runs ON THE GPU BOX from the repo root: FETCH_SIZE / WRITE_SIZE / L2 hit counters of the planned interpolation (headline
# shape, read in place) under launch-time switches, one rocprofv3 --pmc pass per counter group and variant.
#   gpurun -- 'bash tools/pmc_order.sh r05 "base:" "split4:S3_PLAN_SPLIT=4" ...'      -> gpurun_out/pmc_order_<tag>.csv
set -o pipefail
tag=$1; shift
root=$(pwd); out=$root/gpurun_out/pmc_order_$tag; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
echo "variant,kernel,counter,mean_per_launch,launches" > "$out.csv"
for v in "$@"; do
    name=${v%%:*}; envs=${v#*:}
    for group in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
        g=$(echo $group | tr ' ' '_')
        ( for e in $(echo "$envs" | tr ',' ' '); do export "$e"; done
          rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$out/${name}_$g" -- python "$root/tools/ab_order.py" "v:" ${AB_T:-1000} 0 3 > "$out/$name.$g.log" 2>&1 )
        f=$(find "$out/${name}_$g" -name "*counter_collection.csv" | head -n 1)
        python - "$f" "$out.csv" "$name" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "interp_planned" in r["Kernel_Name"]]
acc = {}
for r in rows:
    acc.setdefault((r["Kernel_Name"].split("(")[0][:50], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(sys.argv[2], "a") as f:
    for (kn, c), v in acc.items():
        f.write('%s,"%s",%s,%f,%d\n' % (sys.argv[3], kn, c, sum(v) / len(v), len(v)))
PY
        rm -rf "$out/${name}_$g"
    done
done
cd "$root"
cat "$out.csv"
