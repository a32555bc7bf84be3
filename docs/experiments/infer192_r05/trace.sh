#!/bin/bash
# on the GPU box: kernel trace of 100 replays of the batch-1 192x192 forward; per-kernel durations and the gaps between them
root=$(pwd); out=$root/gpurun_out/infer192; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $root/docs/experiments/infer192_r05/infer_loop.py 200 > $out/plain.txt 2>&1
rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python $root/docs/experiments/infer192_r05/infer_loop.py 100 > $out/under_prof.txt 2> $out/trace.err
cd $root
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/infer192/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 20 forwards: find period by the head kernel
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'head_mfma_fwd' in n or 'head_fwd' in n]
lo, hi = idx[-3] + 1, idx[-2] + 1
with open('gpurun_out/infer192/one_forward.txt', 'w') as o:
    prev_end = None
    tot_k = 0; tot_gap = 0
    for r in rows[lo:hi]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        gap = (s - prev_end) if prev_end else 0
        tot_k += e - s; tot_gap += gap
        o.write('%-70s %7.2f us  gap %6.2f  grid %s wg %s lds %s\n' % (r['Kernel_Name'][:70], (e - s) / 1e3, gap / 1e3, r.get('Grid_Size_X', ''), r.get('Workgroup_Size_X', ''), r.get('LDS_Block_Size', '')))
        prev_end = e
    o.write('kernels %d  kernel time %.1f us  gaps %.1f us\n' % (hi - lo, tot_k / 1e3, tot_gap / 1e3))
PY
find $out -name '*kernel_trace.csv' -delete; find $out -name '*.db' -delete
cat $out/plain.txt | tail -1; tail -1 $out/one_forward.txt
