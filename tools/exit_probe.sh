# dev probe: the GPU kernel tests N times in fresh processes; prints every exit status, stops at the first non-zero one and shows its log
n=${1:-10}
for i in $(seq 1 $n); do
  python -X faulthandler -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/exit_probe_$i.log 2>&1
  rc=$?
  echo "run $i: exit status $rc"
  if [ $rc -ne 0 ]; then grep -v "^  File" gpurun_out/exit_probe_$i.log | tail -20 | cut -c1-200; break; fi
done
