#!/bin/bash
# Diagnosis build: the library with phase clocks in the unrolled 3x3 convolution -> docs/experiments/bin/libdfl_qtrace.so
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
src="$root/deepfluorolabeling-ipcai2020_amd/csrc"; lib="$root/deepfluorolabeling-ipcai2020_amd/lib"
bash "$src/build.sh" >/dev/null
mkdir -p "$root/docs/experiments/bin"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-sched-strategy=max-ilp -DDFL_CONVQ_TRACE -c "$src/convq_bf16.hip" -o "$root/docs/experiments/bin/convq_trace.o"
objs=$(ls "$lib"/*.o | grep -v convq_bf16.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/docs/experiments/bin/libdfl_qtrace.so" $objs "$root/docs/experiments/bin/convq_trace.o"
echo built docs/experiments/bin/libdfl_qtrace.so
