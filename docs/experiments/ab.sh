#!/bin/bash
# A/B of environment switches on the default bench workload, inside ONE gpurun call (boxes differ by a few percent).
# Usage: docs/experiments/ab.sh "VAR=1 VAR2=x" "VAR=0" ...   -> gpurun_out/ab.txt
out=gpurun_out/ab.txt
: > $out
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting python bench.py --no-cpu-baseline --no-fwd --no-fp32-reference --steps 30 > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err
  python - "$setting" gpurun_out/ab_$i.json >> $out <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); sys.exit(0)
k = d.get('kernels', {})
def fam(prefix):
    return sum(v['ms'] for n, v in k.items() if n.startswith(prefix))
print('%-40s %8.1f img/s %6.3f ms/step | gpu %6.3f host %5.2f idle %.3f | convp %.3f wgradp %.3f reduce %.3f bnrelu %.3f bnfin %.3f head %.3f direct %.3f' % (
    sys.argv[1], d['value'], d['ms_per_step'], d.get('gpu_time_ms_per_step', 0), d.get('host_enqueue_ms_per_step', 0), d.get('gpu_idle_frac', 0),
    fam('convp_kernel'), fam('wgradp_kernel'), fam('ReduceBatch'), fam('BnReluBwd'), fam('BnFinalize') + fam('BnBwdFinalize'), fam('Head'), fam('direct_')) + ' partial MB %s' % d.get('partial_sum_mb_per_step'))
PY
done
cat $out
