#!/bin/bash
# on the GPU box: rocprofv3 --kernel-trace --stats of 100 batch-1 192x192 forwards in the parity arithmetics (fp32 tensors)
root=$(pwd); out=$root/gpurun_out/infer192_modes; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in fp32 bf16x3; do
  rocprofv3 --kernel-trace --stats -d $out/$m -o t --output-format csv -- python $root/docs/experiments/infer192_r05/infer_loop.py 100 $m > $out/$m.txt 2> $out/$m.err
  cp "$(find $out/$m -name '*kernel_stats.csv' | head -1)" $out/r05_infer192_${m}_kernel_stats.csv
  find $out/$m -name '*kernel_trace.csv' -delete; find $out/$m -name '*.db' -delete
done
cd $root
grep "per forward" $out/*.txt
