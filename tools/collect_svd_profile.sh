#!/bin/bash
# GPU box: rocprofv3 kernel stats of the weighted Gram kernel (tools/svd_probe.py at the bench grid's size)
set -o pipefail
tag=${1:-r02}; root=$(pwd); out=$root/gpurun_out/prof_$tag/svd; mkdir -p "$out"; export TMPDIR=/tmp
python tools/svd_probe.py > "$out/svd_probe.txt" 2>&1 || exit 1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python "$root/tools/svd_probe.py" > "$out/stats.log" 2>&1 || exit 1
cd "$root"; find "$out" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \; ; rm -rf "$out/stats"
cat "$out/svd_probe.txt"; head -5 "$out/kernel_stats.csv"
