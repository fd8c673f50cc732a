# dev probe: the GPU kernel tests N times in fresh processes with the native SIGABRT backtrace on; stops at the first non-zero status
n=${1:-10}
export S3_ABORT_BACKTRACE=$(pwd)/gpurun_out/abort_backtrace.txt
for i in $(seq 1 $n); do
  python -X faulthandler -m pytest tests/test_gpu_kernels.py -x -q -s -m gpu > gpurun_out/exit_probe_$i.log 2>&1
  rc=$?
  echo "run $i: exit status $rc"
  if [ $rc -ne 0 ]; then grep -n "File \"/root/repo" gpurun_out/exit_probe_$i.log | head -3; grep -v "^  File\|^\.*$\|Extension modules" gpurun_out/exit_probe_$i.log | tail -25 | cut -c1-300; break; fi
done
