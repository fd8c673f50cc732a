#!/bin/bash
# Runs ON THE GPU BOX: per-kernel UTCL1 translation hits / misses and fabric bytes of tools/yard_probe.py's launches (one pass per
# counter group, kernel-trace only): the headline kernel against the load-only yardstick with one / two lines of a row per visit
root=$(pwd); out=$root/gpurun_out/yard_pmc; mkdir -p $out; export TMPDIR=/tmp
cd /tmp
# (one pass per group; FETCH_SIZE and WRITE_SIZE do NOT fit one pass -- asked for together rocprofv3 aborts and then hangs)
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  g=$(echo $grp | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/$g -- python $root/tools/yard_probe.py 1 plan > $out/$g.log 2>&1 &
  pid=$!
  while kill -0 $pid 2>/dev/null; do sleep 30; echo "[yard_pmc] $g still running"; done
  wait $pid
  f=$(find $out/$g -name "*counter_collection.csv" | head -n 1)
  python - "$f" <<'PY' | tee $out/$g.summary
import csv, sys, collections
by = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if any(s in name for s in ("plan_loads_kernel", "interp_planned_shift", "interp_planned_kernel", "stream_kernel")):
        by[(name.split("(")[0][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (name, c), v in sorted(by.items()):
    print(f"{name:72s} {c:34s} mean {sum(v) / len(v):16.0f}  ({len(v)} launches)")
PY
  rm -rf $out/$g
done
