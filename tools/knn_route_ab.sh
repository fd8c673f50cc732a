#!/bin/bash
# GPU box: refine wall-clock of a bench workload against S3_KNN_COOP_MIN (smallest batch that takes the wavefront kernels)
# and with them switched off -> gpurun_out/knn_route_<workload>.txt
w=${1:-cylinder3D_Re3900}
out=gpurun_out/knn_route_$w.txt; : > $out
for min in off 1 512 4096; do
    if [ $min = off ]; then export S3_KNN_COOP=0; else unset S3_KNN_COOP; export S3_KNN_COOP_MIN=$min; fi
    python3 bench.py --workload "$w" --no-cpu-baseline --no-batches --steps 2 --warmup 1 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('S3_KNN_COOP_MIN $min', d['refine_wall_s'], d['refine_runs_s'])" >> $out || exit 1
done
cat $out
