# dev probe: does the interpreter exit cleanly after the GPU kernel tests?  (up to 4 runs; stops at the first non-zero status)
for i in 1 2 3 4; do
  python -X faulthandler -m pytest tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/exit_probe_$i.log 2>&1
  rc=$?
  echo "run $i: exit status $rc"
  if [ $rc -ne 0 ]; then tail -80 gpurun_out/exit_probe_$i.log | cut -c1-200; break; fi
done
