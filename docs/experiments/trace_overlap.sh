cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $root/gpurun_out/trace_ov -o t --output-format csv -- python $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile > /dev/null 2>&1
cd $root
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/trace_ov/**/t_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: take last 400 kernels
rows=rows[-330:]
t0=int(rows[0]['Start_Timestamp'])
prev_end=t0
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'].replace('void dfl::','')[:34]
    print('%8.1f %7.1f q%s %s %s'%((s-t0)/1e3,(e-s)/1e3,r['Queue_Id'],'OVL' if s<prev_end-500 else '   ',name))
    prev_end=max(prev_end,e)
PY
