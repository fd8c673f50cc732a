#!/bin/bash
# Runs ON THE GPU BOX: the round-6 evidence, ONE collection after the last csrc/ commit (S3_LEASE=<n> for a second box): the headline
# bench as the driver runs it (traffic of the headline AND of the batch shapes measured inside the run, yardsticks in the line) +
# rocprofv3 stats + FETCH / WRITE passes, the batch shapes of roofline_batches, box5e7, the SVD line (eigenproblem through s3_sym_eig),
# self-launched multi-rank runs on the one GPU (2 and 5 ranks), the writer A/B
lease=${S3_LEASE:-1}
export S3_LEASE=$lease
part=${1:-all}
if [ "$part" = "all" ] || [ "$part" = "headline" ]; then
    bash tools/collect_profile.sh r06 cylinder3D || exit 1
fi
if [ "$part" = "all" ] || [ "$part" = "shapes" ]; then
    S3_BENCH_FAST=1 bash tools/collect_profile.sh r06 cylinder3D_T25 --t-batch 25 || exit 1
    S3_BENCH_FAST=1 bash tools/collect_profile.sh r06 cylinder3D_T25x3 --t-batch 25 --n-comp 3 || exit 1
    S3_BENCH_FAST=1 bash tools/collect_profile.sh r06 cylinder3D_T100 --t-batch 100 || exit 1
    S3_BENCH_FAST=1 bash tools/collect_profile.sh r06 box5e7 --workload box5e7 || exit 1
fi
if [ "$part" = "all" ] || [ "$part" = "extra" ]; then
    root=$(pwd); out=$root/gpurun_out/prof_r06/extra; mkdir -p $out; export TMPDIR=/tmp
    python bench.py --workload svd > $out/bench_svd.json 2> $out/bench_svd.err
    S3_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_2ranks_self_launched.json 2> $out/bench_2ranks_self_launched.err
    S3_BENCH_SHARE_GPU=1 python bench.py --gpus 5 --workload cylinder3D_small --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_5ranks_small.json 2> $out/bench_5ranks_small.err
    python tools/ab_inplace.py 1000 > $out/ab_inplace.txt 2>&1
    python examples/s3_for_synthetic_OAT15.py /tmp/s3_oat15 500 > $out/example_oat15.txt 2>&1; ls -la /tmp/s3_oat15 >> $out/example_oat15.txt 2>&1; rm -rf /tmp/s3_oat15
fi
