#!/bin/bash
# Memory-path counters of the default training step, one rocprofv3 pass per group (run ON the GPU box) -> gpurun_out/pmc_mem.txt
set -u
root=$(pwd); out=$root/gpurun_out/pmc_mem; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bench="python $root/bench.py --steps 5 --warmup 2 --prewarm-seconds 0 --no-cpu-baseline --no-fp32-reference --no-fwd --no-configs3 --no-profile"
i=0
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr" \
           "SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d "$out/g$i" -o p --output-format csv -- $bench > /dev/null 2> "$out/g$i.err"
done
cd "$root"
python - "$out" > gpurun_out/pmc_mem.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0]
        k = k.replace('void dfl::', '').replace('dfl::', '')
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
        cnt[k][row['Counter_Name']] += 1
names = sorted({c for k in acc for c in acc[k]})
want = [k for k in acc if k.startswith(('wgradp_kernel<3, 3', 'wgradp_kernel<3,3', 'convp_kernel<1, 4, 3, 1', 'convp_kernel<1,4,3,1', 'reduce_batch', 'wgradp_kernel<1', 'convp_kernel<4, 1, 3'))]
for k in sorted(want):
    print(k[:70])
    for c in names:
        if c in acc[k]:
            print('   %-36s %14.1f per launch (%d)' % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
PY
find "$out" -name '*.csv' -delete; find "$out" -name '*.db' -delete
