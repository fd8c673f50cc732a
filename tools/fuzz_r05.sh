#!/bin/bash
# Runs ON THE GPU BOX: the randomised differential tests on the round's final native sources, incl. the optional launch forms of r5
out=gpurun_out/fuzz_r05.txt; : > $out
run() { echo "== $*" >> $out; ( "$@" 2>&1 | tail -2 ) >> $out; }
run python tools/fuzz_interp.py 501 300
run python tools/fuzz_stream.py 502 300
run python tools/fuzz_inplace.py 503 300
S3_PLAN_MIN_BLOCKS=1 S3_PLAN_TAIL=5x3 run python tools/fuzz_inplace.py 504 200
S3_PLAN_MIN_BLOCKS=1 S3_PLAN_TAIL=9x4 S3_OUT_HOLD=1 run python tools/fuzz_inplace.py 505 200
S3_PLAN_SPLIT=3 S3_PLAN_BRICK=4 run python tools/fuzz_interp.py 506 150
S3_PLAN_MIN_BLOCKS=1 S3_PLAN_TAIL=3x2 run python tools/fuzz_interp.py 507 150
run python tools/fuzz_export.py 508 100
run python tools/fuzz_knn.py 509 200
run python tools/fuzz_refine_gpu.py 510 60
cat $out
