#!/bin/bash
# Diagnosis build: the library with phase clocks in the LDS-DMA weight-gradient kernel -> docs/experiments/bin/libdfl_wgqtrace.so
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
src="$root/deepfluorolabeling-ipcai2020_amd/csrc"; lib="$root/deepfluorolabeling-ipcai2020_amd/lib"
bash "$src/build.sh" >/dev/null
mkdir -p "$root/docs/experiments/bin"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp -DDFL_WGQ_TRACE "$@" -c "$src/wgradq_bf16.hip" -o "$root/docs/experiments/bin/wgradq_trace.o"
objs=$(ls "$lib"/*.o | grep -v wgradq_bf16.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/docs/experiments/bin/libdfl_wgqtrace.so" $objs "$root/docs/experiments/bin/wgradq_trace.o"
echo built docs/experiments/bin/libdfl_wgqtrace.so
