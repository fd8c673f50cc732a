#!/bin/bash
# GPU box: refine wall-clock (bench.py, 4 refines: first + 3 steady) of a workload under each given environment setting
#   tools/refine_ab.sh cylinder3D_Re3900 S3_LEAF_SET_THREAD=1 S3_LEAF_SET_THREAD=0   ("-" = no setting)
# -> gpurun_out/refine_ab_<workload>.txt
w=$1; shift
out=gpurun_out/refine_ab_$w.txt; : > $out
for rep in 1 2; do for kv in "$@"; do
    if [ "$kv" = "-" ]; then pre=""; else pre="$kv"; fi
    env $pre python3 bench.py --workload "$w" --no-cpu-baseline --no-batches --steps 2 --warmup 1 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kv', d['refine_wall_s'], [round(t, 4) for t in d['refine_runs_s']])" >> $out || exit 1
done; done
cat $out
