#!/bin/bash
# Runs ON THE GPU BOX: one lease of the round-4 evidence (S3_LEASE=<n>): the headline bench (dense batch read in place,
# interp_planned_shift_kernel) + the batch shapes of roofline_batches (rocprofv3 stats + FETCH / WRITE each)
lease=${S3_LEASE:-1}
export S3_LEASE=$lease
bash tools/collect_profile.sh r04 cylinder3D || exit 1
S3_BENCH_FAST=1 bash tools/collect_profile.sh r04 cylinder3D_T25 --t-batch 25 || exit 1
S3_BENCH_FAST=1 bash tools/collect_profile.sh r04 cylinder3D_T25x3 --t-batch 25 --n-comp 3 || exit 1
S3_BENCH_FAST=1 bash tools/collect_profile.sh r04 cylinder3D_T100 --t-batch 100 || exit 1
