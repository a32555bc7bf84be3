#!/bin/bash
# kernel sequence of one training step (bf16s) -> gpurun_out/seq.txt (start offset, duration, gap to the previous kernel's end, queue, name)
root=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/seq -o s --output-format csv -- python $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-fp32-reference --no-fwd --no-profile --math bf16s > /dev/null 2>/tmp/seq.err
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/seq/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last two steps: from the third-last sgd kernel on
sg = [i for i, r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
first = sg[-5] if len(sg) >= 5 else 0
t0 = int(rows[first]['Start_Timestamp'])
out, prev_end = [], None
for r in rows[first:]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev_end) / 1e3 if prev_end else 0
    name = r['Kernel_Name'].replace('void ', '').replace('dfl::', '')
    out.append('%9.1f  %7.1f us  gap %7.1f  q%s  %s' % ((st - t0) / 1e3, (en - st) / 1e3, gap, r.get('Queue_Id', '?'), name[:80]))
    prev_end = max(prev_end or 0, en)
open('/root/repo/gpurun_out/seq.txt', 'w').write('\n'.join(out))
PY
