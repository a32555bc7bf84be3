#!/bin/bash
# Build libdfl_hip.so of a git revision into docs/experiments/bin/libdfl_<name>.so (A/B runs in ONE gpurun call:
# DFL_LIB_OVERRIDE=docs/experiments/bin/libdfl_<name>.so python tools/kbench_bf16.py).   usage: build_rev.sh <rev> <name>
set -euo pipefail
rev="$1"; name="$2"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
tmp="$(mktemp -d)"
git -C "$root" archive "$rev" deepfluorolabeling-ipcai2020_amd/csrc include | tar -x -C "$tmp"
bash "$tmp/deepfluorolabeling-ipcai2020_amd/csrc/build.sh" > /dev/null
mkdir -p "$root/docs/experiments/bin"
cp "$tmp/deepfluorolabeling-ipcai2020_amd/lib/libdfl_hip.so" "$root/docs/experiments/bin/libdfl_$name.so"
rm -rf "$tmp"
echo "built docs/experiments/bin/libdfl_$name.so from $rev"
