#!/bin/bash
# Build a variant of libdfl_hip.so with extra -D flags (experiments).  The output lives in-tree (git-ignored *.so) so that
# it travels to the GPU box:  tools/exp/build_variant.sh NAME -DFOO ...  ->  tools/exp/bin/NAME/libdfl_hip.so
# Use with DFL_LIB_OVERRIDE=$GRAFT_REPO_ROOT/tools/exp/bin/NAME/libdfl_hip.so
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/deepfluorolabeling-ipcai2020_amd/csrc
out=$root/tools/exp/bin/$name
mkdir -p $out
for f in api conv_gemm conv_rows wgrad_gemm direct_small bn_elem head loss prep; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $src/$f.hip -o $out/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdfl_hip.so $out/*.o
rm -f $out/*.o
echo $out/libdfl_hip.so
