#!/bin/bash
# Diagnosis: build libdfl_hip.so with -DDFL_CONV_TRACE into a scratch dir and print where the first wave of each
# workgroup of one 3x3 layer spends its shader-clock time (run on the GPU box: gpurun -- bash docs/experiments/conv_phase_trace.sh).
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/deepfluorolabeling-ipcai2020_amd/csrc
out=/tmp/dfl_trace_build
mkdir -p $out
for f in api conv_gemm wgrad_gemm direct_small bn_elem head loss prep; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDFL_CONV_TRACE -c $src/$f.hip -o $out/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdfl_hip.so $out/*.o
DFL_LIB_OVERRIDE=$out/libdfl_hip.so python $root/docs/experiments/conv_phase_trace.py "$@"
