#!/bin/bash
# Runs ON THE GPU BOX: the round-5 evidence, ONE collection after the last csrc/ commit (S3_LEASE=<n> for a second box): the
# headline bench as the driver runs it (traffic measured inside the run) + rocprofv3 stats + FETCH / WRITE passes, the batch shapes
# of roofline_batches, box5e7, the SVD line, the self-launched two-rank runs (bench.py --gpus 2 without a launcher, one GPU shared)
lease=${S3_LEASE:-1}
export S3_LEASE=$lease
part=${1:-all}
if [ "$part" = "all" ] || [ "$part" = "headline" ]; then
    bash tools/collect_profile.sh r05 cylinder3D || exit 1
fi
if [ "$part" = "all" ] || [ "$part" = "shapes" ]; then
    S3_BENCH_FAST=1 bash tools/collect_profile.sh r05 cylinder3D_T25 --t-batch 25 || exit 1
    S3_BENCH_FAST=1 bash tools/collect_profile.sh r05 cylinder3D_T25x3 --t-batch 25 --n-comp 3 || exit 1
    S3_BENCH_FAST=1 bash tools/collect_profile.sh r05 cylinder3D_T100 --t-batch 100 || exit 1
    S3_BENCH_FAST=1 bash tools/collect_profile.sh r05 box5e7 --workload box5e7 || exit 1
fi
if [ "$part" = "all" ] || [ "$part" = "extra" ]; then
    root=$(pwd); out=$root/gpurun_out/prof_r05/extra; mkdir -p $out; export TMPDIR=/tmp
    python bench.py --workload svd > $out/bench_svd.json 2> $out/bench_svd.err
    S3_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_2ranks_self_launched.json 2> $out/bench_2ranks_self_launched.err
    S3_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --workload box5e7_small --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_2ranks_box_small.json 2> $out/bench_2ranks_box_small.err
    S3_BENCH_SHARE_GPU=1 S3_BENCH_HANG="1:alive:rccl" S3_BENCH_BOOT_TIMEOUT_S=20 python bench.py --gpus 2 --workload cylinder3D_small --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_2ranks_wedged_then_gloo.json 2> $out/bench_2ranks_wedged_then_gloo.err
    python tools/ab_order.py "default:;tail_off:S3_PLAN_TAIL=0;hold_off:S3_OUT_HOLD=0" 1000 5 10 > $out/ab_tail_hold.txt 2>&1
    python tools/ab_inplace.py 1000 > $out/ab_inplace.txt 2>&1
    python examples/s3_for_synthetic_OAT15.py /tmp/s3_oat15 500 > $out/example_oat15.txt 2>&1; ls -la /tmp/s3_oat15 >> $out/example_oat15.txt 2>&1; rm -rf /tmp/s3_oat15
fi
