#!/bin/bash
# GPU box: wave-state / instruction counters of the child-metric kernels (tools/knn_coop_probe.py, 40 000 cells per level), one
# rocprofv3 --pmc run per counter pair (kernel-trace only) -> gpurun_out/coop_pmc.csv.  Lane utilisation of the vector ALU =
# SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
root=$(pwd); out=$root/gpurun_out/coop_pmc; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
echo "Kernel,Counter,mean" > $root/gpurun_out/coop_pmc.csv
for group in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"; do
  gname=$(echo $group | tr ' ' '_')
  rocprofv3 --pmc $group --kernel-trace --output-format csv -d $out/$gname -- python3 $root/tools/knn_coop_probe.py 40000 > $out/$gname.log 2>&1
  f=$(find $out/$gname -name "*counter_collection.csv" | head -n 1)
  python - "$f" "$root/gpurun_out/coop_pmc.csv" <<'PY'
import csv, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "child_metric" in r["Kernel_Name"]:
        acc.setdefault((r["Kernel_Name"].split("(")[0][:44], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(sys.argv[2], "a") as f:
    for (k, c), v in acc.items():
        f.write('"%s",%s,%f,%d\n' % (k, c, sum(v) / len(v), len(v)))
PY
  rm -rf $out/$gname
done
