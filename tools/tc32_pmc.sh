#!/bin/bash
# Runs ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE of tools/tc32_probe.py's launches (64-cell against 32-cell tiles), one pass per counter
root=$(pwd); out=$root/gpurun_out/tc32_pmc; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python $root/tools/tc32_probe.py 1 > $out/$c.log 2>&1
  f=$(find $out/$c -name "*counter_collection.csv" | head -n 1)
  python - "$f" <<'PY'
import csv, sys, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "interp_planned_kernel" in r["Kernel_Name"]:
        by[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(by.items()):
    print(k, "mean KiB", round(sum(v) / len(v)), len(v))
PY
  rm -rf $out/$c
done
