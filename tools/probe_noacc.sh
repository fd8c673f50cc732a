# dev helper (GPU box): the persistent kernel with parts of its step compiled out (S3_PROBE_NO_ACC = 1: no accumulate phase, 2: its LDS
# reads only, 3: its conversions + FMAs only, 4: everything but the output stores) -- where a step's time goes.  Builds into the
# box's scratch copy of the tree only.
set -e
root=$(pwd); out=$root/gpurun_out/noacc; mkdir -p $out
python tools/rowlen_probe.py 25 32 75 100 > $out/mode0.txt 2>&1
for mode in ${S3_PROBE_MODES:-1 2 3 4}; do
    cd sparsespatialsampling_amd/csrc
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DS3_PROBE_NO_ACC=$mode -I $root/include -I /opt/rocm/include -c interp_plan.hip -o _obj/interp_plan.hip.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../libs3hip.so _obj/*.o -ldl
    cd $root
    python tools/rowlen_probe.py 25 32 75 100 > $out/mode$mode.txt 2>&1
done
for mode in 0 ${S3_PROBE_MODES:-1 2 3 4}; do echo "mode $mode"; grep "T=" $out/mode$mode.txt | cut -c1-75; done
