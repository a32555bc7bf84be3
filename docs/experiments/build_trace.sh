#!/bin/bash
# Diagnosis build: the library with phase clocks in the patch-resident convolution kernel -> docs/experiments/bin/libdfl_trace.so
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
src="$root/deepfluorolabeling-ipcai2020_amd/csrc"; lib="$root/deepfluorolabeling-ipcai2020_amd/lib"
bash "$src/build.sh" >/dev/null
mkdir -p "$root/docs/experiments/bin"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp -DDFL_CONVP_TRACE -c "$src/convp_bf16.hip" -o "$root/docs/experiments/bin/convp_trace.o" &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp -DDFL_WGP_TRACE -c "$src/wgradp_bf16.hip" -o "$root/docs/experiments/bin/wgradp_trace.o" &
wait
objs=$(ls "$lib"/*.o | grep -v convp_bf16.o | grep -v wgradp_bf16.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/docs/experiments/bin/libdfl_trace.so" $objs "$root/docs/experiments/bin/convp_trace.o" "$root/docs/experiments/bin/wgradp_trace.o"
echo built docs/experiments/bin/libdfl_trace.so
