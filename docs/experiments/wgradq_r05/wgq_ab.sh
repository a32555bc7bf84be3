cd $GRAFT_REPO_ROOT
for i in 1 2; do
TAG=wgq0 DFL_WGQ=0 python tools/kbench_bf16.py bf16s 3 2>&1 | grep step | cut -c1-200
TAG=wgq1 DFL_WGQ=1 python tools/kbench_bf16.py bf16s 3 2>&1 | grep step | cut -c1-200
done
