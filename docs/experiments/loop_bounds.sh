# What does the bf16x3 GEMM loop wait for?  Times one level-2 layer with diagnosis builds that remove one resource each
# (results are wrong, only the time matters).  The DFL_EXP_* branches left csrc/conv_gemm.hip in round 5: apply
# docs/experiments/conv_gemm_loop_bounds_r02.diff (patch -p0 -R style: it is product -> diagnosis) first.  Build (CPU box):
#   for v in NOLOAD NOMFMA NOBAR; do docs/experiments/build_variant.sh $v -DDFL_EXP_$v; done
#   docs/experiments/build_variant.sh NOLOAD_NOBAR -DDFL_EXP_NOLOAD -DDFL_EXP_NOBAR ; ... (any combination)
# Run (GPU box): bash docs/experiments/loop_bounds.sh
root=$GRAFT_REPO_ROOT
export DFL_MATH=bf16x3
for flags in WX sabrW; do
  for shape in "16 48 48 128 128 3" "16 48 48 256 128 3" "16 12 12 512 512 3"; do
    echo "== $shape $flags"
    printf "%-14s" base; python $root/tools/kbench.py conv $shape 1 50 $flags | sed 's/.*: //'
    for v in $(ls $root/docs/experiments/bin); do
      [ -f $root/docs/experiments/bin/$v/libdfl_hip.so ] || continue
      printf "%-14s" $v; DFL_LIB_OVERRIDE=$root/docs/experiments/bin/$v/libdfl_hip.so python $root/tools/kbench.py conv $shape 1 50 $flags | sed 's/.*: //'
    done
  done
done
