#!/bin/bash
# Variant build of ONE kernel file with extra flags -> docs/experiments/bin/libdfl_<name>.so   (build_variant.sh name file.hip -DFLAG ...)
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
src="$root/deepfluorolabeling-ipcai2020_amd/csrc"; lib="$root/deepfluorolabeling-ipcai2020_amd/lib"
name="$1"; file="$2"; shift 2
mkdir -p "$root/docs/experiments/bin"
base="${file%.hip}"
extra=""; case "$base" in convp_bf16|convq_bf16|convn_bf16|wgradp_bf16|bn_elem) extra="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $extra "$@" -c "$src/$file" -o "$root/docs/experiments/bin/${base}_$name.o"
objs=$(ls "$lib"/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/docs/experiments/bin/libdfl_$name.so" $objs "$root/docs/experiments/bin/${base}_$name.o"
echo built docs/experiments/bin/libdfl_$name.so
