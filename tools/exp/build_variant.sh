#!/bin/bash
# Build a variant of libdfl_hip.so with extra -D flags into /tmp/dfl_variant (experiments): tools/exp/build_variant.sh -DFOO
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/deepfluorolabeling-ipcai2020_amd/csrc
out=/tmp/dfl_variant
mkdir -p $out
for f in api conv_gemm wgrad_gemm direct_small bn_elem head loss prep; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $src/$f.hip -o $out/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdfl_hip.so $out/*.o
echo $out/libdfl_hip.so
