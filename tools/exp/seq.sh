#!/bin/bash
# kernel sequence of one training step (bf16s) -> gpurun_out/seq.txt
root=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/seq -o s --output-format csv -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp32-reference --no-fwd --no-profile --math bf16s > /dev/null 2>/tmp/seq.err
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/seq/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'][:60] for r in rows]
# last step: from the last 'prep'/'direct_conv' start; print the final 330 kernels with durations and gaps
out = []
prev_end = None
for r in rows[-330:]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - prev_end) / 1e3 if prev_end else 0
    out.append('%7.1f us  gap %6.1f  %s' % ((en - st) / 1e3, gap, r['Kernel_Name'][:70]))
    prev_end = en
open('/root/repo/gpurun_out/seq.txt', 'w').write('\n'.join(out))
PY
