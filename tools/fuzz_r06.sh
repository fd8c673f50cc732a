#!/bin/bash
# Runs ON THE GPU BOX: the randomised differential tests on the final native sources of round 6 (switches now parsed once: set in the environment of each process)
out=gpurun_out/fuzz_r06.txt; : > $out
run() { echo "== $*" >> $out; ( "$@" 2>&1 | tail -2 ) >> $out; }
run python tools/fuzz_interp.py 601 300
run python tools/fuzz_stream.py 602 300
run python tools/fuzz_inplace.py 603 300
S3_PLAN_MIN_BLOCKS=1 S3_PLAN_TAIL=5x3 run python tools/fuzz_inplace.py 604 200
S3_PLAN_MIN_BLOCKS=1 S3_PLAN_TAIL=9x4 S3_OUT_HOLD=1 run python tools/fuzz_inplace.py 605 200
S3_PLAN_SPLIT=3 S3_PLAN_BRICK=4 run python tools/fuzz_interp.py 606 150
S3_PLAN_MIN_BLOCKS=1 S3_PLAN_TAIL=3x2 run python tools/fuzz_interp.py 607 150
run python tools/fuzz_export.py 608 100
run python tools/fuzz_knn.py 609 200
run python tools/fuzz_refine_gpu.py 610 60
cat $out
