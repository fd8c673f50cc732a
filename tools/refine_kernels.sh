#!/bin/bash
# GPU box: per-kernel time of one SamplingTree.refine() of the bench workload (rocprofv3 kernel trace of tools/profile_refine.py)
set -o pipefail
root=$(pwd); out=$root/gpurun_out/prof_refine; mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python "$root/tools/profile_refine.py" ${1:-cylinder3D_Re3900} > "$out/run.log" 2>&1 || exit 1
cd "$root"; find "$out" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
head -25 "$out/kernel_stats.csv"
