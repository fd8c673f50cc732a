#!/bin/bash
# Runs ON THE GPU BOX: one lease of the round-3 evidence (S3_LEASE=<n>): the headline bench + the batch shapes of roofline_batches
# (rocprofv3 stats + FETCH / WRITE each); lease 1 also collects box5e7, the SVD line and the refine kernels' stats.
lease=${S3_LEASE:-1}
export S3_LEASE=$lease
bash tools/collect_profile.sh r03 cylinder3D || exit 1
S3_BENCH_FAST=1 bash tools/collect_profile.sh r03 cylinder3D_T25 --t-batch 25 || exit 1
S3_BENCH_FAST=1 bash tools/collect_profile.sh r03 cylinder3D_T25x3 --t-batch 25 --n-comp 3 || exit 1
S3_BENCH_FAST=1 bash tools/collect_profile.sh r03 cylinder3D_T100 --t-batch 100 || exit 1
if [ "$lease" = "1" ]; then
    S3_BENCH_FAST=1 bash tools/collect_profile.sh r03 box5e7 --workload box5e7 || exit 1
    root=$(pwd); out=$root/gpurun_out/prof_r03/extra; mkdir -p $out; export TMPDIR=/tmp
    python bench.py --workload svd > $out/bench_svd.json 2> $out/bench_svd.err
    cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/svd -- python $root/bench.py --workload svd --steps 5 --warmup 2 > $out/svd.log 2>&1
    find $out/svd -name "*kernel_stats.csv" -exec cp {} $out/svd_kernel_stats.csv \; ; rm -rf $out/svd
    for w in cylinder3D_Re3900 box5e7; do
        rocprofv3 --kernel-trace --stats --output-format csv -d $out/refine_$w -- python $root/bench.py --workload $w --no-cpu-baseline --no-batches --steps 2 --warmup 1 > $out/refine_$w.log 2>&1
        find $out/refine_$w -name "*kernel_stats.csv" -exec cp {} $out/refine_${w}_kernel_stats.csv \; ; rm -rf $out/refine_$w
    done
    cd $root
    for w in cylinder3D_Re3900 box5e7; do bash tools/knn_batches.sh $w > /dev/null || exit 1; cp gpurun_out/knn_batches_$w.txt gpurun_out/knn_wall_$w.txt $out/; done
    python3 tools/knn_coop_probe.py > $out/knn_coop_probe.txt 2>&1
fi
