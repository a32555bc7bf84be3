#!/bin/bash
# LDS / issue / wait counters of the bf16 kernels inside the default bench workload, three rocprofv3 --pmc passes (counters only).
# Usage (on the GPU box): docs/experiments/pmc_bf16.sh   -> gpurun_out/pmc_bf16.txt  (per kernel: average counter value per launch)
root=$(pwd); cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/pmc_bf16.txt; : > $out
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1)); d=/tmp/pmc_bf16_$i; rm -rf $d
  rocprofv3 --kernel-trace --pmc $set -d $d -o p --output-format csv -- python $root/bench.py --steps 4 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --no-fp32-reference --no-fwd --no-profile > /dev/null 2> /tmp/pmc_bf16_$i.err
  python - "$d" >> $out <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + '/**/p_counter_collection.csv', recursive=True)
d = collections.defaultdict(lambda: collections.defaultdict(list))
def short(n):
    n = n.replace('void ', '').replace('dfl::', '')
    m = re.match(r'(convp_kernel<\d+, \d+, \d+, \d+, \d+), (true|false)', n)
    if m: return m.group(1) + (', GA>' if m.group(2) == 'true' else '>')
    m = re.match(r'(wgradp_kernel<\d+, \d+)', n)
    if m: return m.group(1) + '>'
    return n.split('(')[0][:40]
for r in csv.DictReader(open(f[0])):
    d[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
import os
pref = os.environ.get('PMC_KERNELS', 'convp_kernel<1, 4, 3, 1;convp_kernel<4, 1, 3, 1;wgradp_kernel<3, 3;convp_kernel<2, 2, 3, 1').split(';')
keep = [k for k in d if any(k.startswith(q) for q in pref)]
for k in sorted(keep):
    print(k, len(next(iter(d[k].values()))), 'launches')
    for c, v in d[k].items():
        print('   %-28s %14.0f' % (c, sum(v) / len(v)))
PY
done
cat $out
