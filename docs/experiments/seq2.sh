#!/bin/bash
root=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --hip-runtime-trace -d /tmp/seq2 -o s --output-format csv -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-fwd --no-profile --math bf16s > /dev/null 2>/tmp/seq2.err
ls /tmp/seq2/*/ 2>/dev/null | head
python - <<'PY'
import csv, glob
kf = glob.glob('/tmp/seq2/**/*kernel_trace.csv', recursive=True)[0]
af = glob.glob('/tmp/seq2/**/*hip_api_trace.csv', recursive=True)[0]
K = list(csv.DictReader(open(kf)))
A = list(csv.DictReader(open(af)))
K.sort(key=lambda r: int(r['Start_Timestamp']))
# the last sgd kernel pair
idx = [i for i, r in enumerate(K) if 'sgd_kernel' in r['Kernel_Name']]
i0 = idx[-2]
t_ref = int(K[i0]['Start_Timestamp'])
ev = []
for r in K[i0 - 4:i0 + 6]:
    ev.append((int(r['Start_Timestamp']), 'K start %s (corr %s, queue %s)' % (r['Kernel_Name'][:40], r.get('Correlation_Id'), r.get('Queue_Id'))))
    ev.append((int(r['End_Timestamp']), 'K end   %s' % r['Kernel_Name'][:40]))
lo, hi = t_ref - 400000, t_ref + 100000
for r in A:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if lo <= s <= hi:
        ev.append((s, 'API %s (%.1f us, corr %s)' % (r['Function'], (e - s) / 1e3, r.get('Correlation_Id'))))
ev.sort()
out = ['%9.1f us  %s' % ((t - t_ref) / 1e3, s) for t, s in ev]
open('/root/repo/gpurun_out/seq2.txt', 'w').write('\n'.join(out))
# where was the sgd launch call issued?
c = K[i0].get('Correlation_Id')
for r in A:
    if r.get('Correlation_Id') == c:
        print('sgd launch API call at %.1f us relative to its kernel start' % ((int(r['Start_Timestamp']) - t_ref) / 1e3))
PY
