#!/bin/bash
# Build a variant of libdfl_hip.so: the named sources recompiled with extra -D flags, every other object taken from the
# regular build (csrc/build.sh must have run).  The output lives in-tree (git-ignored) so that it travels to the GPU box:
#   docs/experiments/build_variant.sh NAME "convp_bf16 wgradp_bf16" -DFOO=1 ...  ->  docs/experiments/bin/NAME/libdfl_hip.so
# Use with DFL_LIB_OVERRIDE=$GRAFT_REPO_ROOT/docs/experiments/bin/NAME/libdfl_hip.so
set -e
name=$1; files=$2; shift 2
root=$(cd "$(dirname "$0")/../.." && pwd)
src=$root/deepfluorolabeling-ipcai2020_amd/csrc; lib=$root/deepfluorolabeling-ipcai2020_amd/lib
out=$root/docs/experiments/bin/$name
mkdir -p $out
objs=""
for o in $lib/*.o; do
  b=$(basename $o .o)
  case " $files " in *" $b "*) ;; *) objs="$objs $o";; esac
done
for f in $files; do
  extra=""
  case "$f" in convp_bf16|wgradp_bf16|bn_elem) extra="-mllvm -amdgpu-sched-strategy=max-ilp";; esac      # as csrc/build.sh
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $extra "$@" -c $src/$f.hip -o $out/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libdfl_hip.so $objs $out/*.o
rm -f $out/*.o
echo $out/libdfl_hip.so
