#!/bin/bash
# Runs ON THE GPU BOX: one lease of the round-4 evidence (S3_LEASE=<n>): the headline bench (dense batch read in place,
# interp_planned_shift_kernel) + the batch shapes of roofline_batches (rocprofv3 stats + FETCH / WRITE each); lease 1 also
# collects box5e7, the SVD line (+ kernel stats) and a two-rank rehearsal of bench.py --gpus 2 on the one GPU (export_sharded)
lease=${S3_LEASE:-1}
export S3_LEASE=$lease
bash tools/collect_profile.sh r04 cylinder3D || exit 1
S3_BENCH_FAST=1 bash tools/collect_profile.sh r04 cylinder3D_T25 --t-batch 25 || exit 1
S3_BENCH_FAST=1 bash tools/collect_profile.sh r04 cylinder3D_T25x3 --t-batch 25 --n-comp 3 || exit 1
S3_BENCH_FAST=1 bash tools/collect_profile.sh r04 cylinder3D_T100 --t-batch 100 || exit 1
if [ "$lease" = "1" ]; then
    S3_BENCH_FAST=1 bash tools/collect_profile.sh r04 box5e7 --workload box5e7 || exit 1
    root=$(pwd); out=$root/gpurun_out/prof_r04/extra; mkdir -p $out; export TMPDIR=/tmp
    python bench.py --workload svd > $out/bench_svd.json 2> $out/bench_svd.err
    cd /tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/svd -- python $root/bench.py --workload svd --steps 5 --warmup 2 > $out/svd.log 2>&1
    find $out/svd -name "*kernel_stats.csv" -exec cp {} $out/svd_kernel_stats.csv \; ; rm -rf $out/svd
    cd $root
    S3_BENCH_SHARE_GPU=1 S3_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29655 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_2ranks_one_gpu.json 2> $out/bench_2ranks_one_gpu.err
    python tools/ab_inplace.py 1000 200 > $out/ab_inplace.txt 2>&1
    python tools/e2e_probe.py 200 > $out/e2e_probe_T200.txt 2>&1
    python tools/e2e_probe.py 25 12 > $out/e2e_probe_T25.txt 2>&1
    python tools/transport_probe.py 1.0 > $out/transport_probe.txt 2>&1
    python tools/svd_breakdown.py > $out/svd_breakdown.txt 2>&1
    python tools/file_probe.py 25 16 > $out/file_probe.txt 2>&1
    python tools/ab_plan.py "tc64:;tc128:S3_TILE_CELLS=128" 5 10 > $out/ab_tile_cells.txt 2>&1
    hipcc --offload-arch=gfx950 -O3 -o /tmp/s3_gather_probe tools/gather_probe.hip > /dev/null 2>&1 && /tmp/s3_gather_probe > $out/gather_probe.txt 2>&1
fi
