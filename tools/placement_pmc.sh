#!/bin/bash
# Runs ON THE GPU BOX: UTCL1 translation hits / misses of the in-place headline launch per TABLE of tools/placement_probe.py (the
# launches go round the tables in a fixed order: 4 tables x 2 outputs x 9 launches per round, 6 rounds)
root=$(pwd); out=$root/gpurun_out/placement_pmc; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"; do
  g=$(echo $grp | tr ' ' '_')
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/$g -- python $root/tools/placement_probe.py 4 > $out/$g.log 2>&1
  f=$(find $out/$g -name "*counter_collection.csv" | head -n 1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "interp_planned_shift" in r["Kernel_Name"]]
by = collections.defaultdict(list)
for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in by.items():
    n = len(v); per = 9; assert n == 4 * 2 * per * 6, n
    acc = collections.defaultdict(list)
    for i, val in enumerate(v):
        table = (i // (2 * per)) % 4
        acc[table].append(val)
    print(c, {t: round(sum(a) / len(a)) for t, a in sorted(acc.items())})
PY
  grep "median" $out/$g.log
  rm -rf $out/$g
done
