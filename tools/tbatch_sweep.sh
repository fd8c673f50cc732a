#!/bin/bash
# GPU box: kernel time and fraction of peak of the interpolation for a list of batch lengths (bench.py --t-batch T)
#   tools/tbatch_sweep.sh 24 25 26 27 28 32   -> gpurun_out/tbatch_sweep.txt
out=gpurun_out/tbatch_sweep.txt; : > $out
for t in "$@"; do
    python3 bench.py --t-batch $t --no-cpu-baseline --no-batches --steps 20 --warmup 3 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('T=$t', r['kernel'], 'ms median %.4f' % r['kernel_ms_median'], 'frac %.3f' % r['frac'], 'alg MB %.0f' % (r['algorithmic_bytes']/1e6))" >> $out || exit 1
done
cat $out
