root=$(pwd); out=$root/gpurun_out/fs; mkdir -p $out; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  S3_STREAM_MAX_CHUNKS=64 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/$c -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-batches > $out/$c.log 2>&1
  f=$(find $out/$c -name "*counter_collection.csv" | head -n 1)
  python3 - "$f" <<'PY'
import csv, sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "interp_planned" in r["Kernel_Name"] and "permute" not in r["Kernel_Name"]]
names={r["Kernel_Name"][:60] for r in csv.DictReader(open(sys.argv[1])) if "interp_planned" in r["Kernel_Name"]}
print(names, sum(v)/len(v), len(v))
PY
  rm -rf $out/$c
done
