# dev probe: which tests have to run in one process for the silent SIGABRT to appear?
a=0; b=0
for i in $(seq 1 15); do
  python -X faulthandler -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "upload_rows" > gpurun_out/p2_a_$i.log 2>&1 || { a=$((a+1)); echo "A (upload_rows only) run $i: status non-zero"; grep -n "File \"/root/repo" gpurun_out/p2_a_$i.log | head -2; }
  python -X faulthandler -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "registered_host_memory or upload_rows" > gpurun_out/p2_b_$i.log 2>&1 || { b=$((b+1)); echo "B (register + upload_rows) run $i: status non-zero"; grep -n "File \"/root/repo" gpurun_out/p2_b_$i.log | head -2; }
done
echo "A: $a of 15 aborted; B: $b of 15 aborted"
