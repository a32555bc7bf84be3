#!/bin/bash
# Builds libdfl_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
srcs=(api.hip conv_gemm.hip conv_rows.hip convp_bf16.hip convq_bf16.hip convn_bf16.hip convs_bf16.hip convs_f32.hip wgradp_bf16.hip wgrad_gemm.hip direct_small.hip bn_elem.hip head.hip loss.hip prep.hip upsample.hip)
objs=()
pids=()
for s in "${srcs[@]}"; do
  o="$out/${s%.hip}.o"
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$here/$s" -nt "$o" ] || [ "$here/common.h" -nt "$o" ] || [ "$here/direct_small.h" -nt "$o" ] || [ "$here/conv_epilogue.h" -nt "$o" ] || [ "$here/conv_rows.h" -nt "$o" ] || [ "$here/convp.h" -nt "$o" ] || [ "$here/head_caps.inc" -nt "$o" ] || [ "$here/head_mfma.inc" -nt "$o" ] || [ "$here/../../include/dfl_hip.h" -nt "$o" ]; then
    # the two patch-resident kernels live at 1-3 waves per SIMD: schedule them for instruction-level parallelism instead of
    # register pressure (same instructions, same results; measured 4.47 -> 4.45 ms per step), and so are the streaming kernels of
    # bn_elem.hip (the batched sums issue their loads earlier: 0.23 -> 0.21 ms per step)
    extra=""
    case "$s" in convp_bf16.hip|convq_bf16.hip|convn_bf16.hip|convs_bf16.hip|convs_f32.hip|wgradp_bf16.hip|bn_elem.hip) extra="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $extra "$@" -c "$here/$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out/libdfl_hip.so" "${objs[@]}"
echo "built $out/libdfl_hip.so"
