# dev helper (GPU box): builds the probe variant of interp_plan.hip (phase timers in the persistent kernel) into the box's scratch copy
# and prints the phases of a step
set -e
root=$(pwd)
cd sparsespatialsampling_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DS3_PROBE_STAMPS -I $root/include -I /opt/rocm/include -c interp_plan.hip -o _obj/interp_plan.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o ../libs3hip.so _obj/*.o -ldl
cd $root
python tools/stream_phases.py "$@"
