cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $root/gpurun_out/trace_gap -o t --output-format csv -- python $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile > /dev/null 2>&1
cd $root
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/trace_gap/**/t_kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find step boundaries via sgd_kernel occurrences
idx=[i for i,r in enumerate(rows) if 'sgd_kernel' in r['Kernel_Name']]
# steps end at every second sgd kernel (2 launches per step)
ends=idx[1::2]
a,b=ends[-2]+1,ends[-1]+1
step=rows[a:b]
t0=int(step[0]['Start_Timestamp']); t1=int(step[-1]['End_Timestamp'])
busy=0; prev_end=t0; gaps=[]; 
for r in step:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if s>prev_end: gaps.append((s-prev_end, r['Kernel_Name'][:50]))
    prev_end=max(prev_end,e)
tot_gap=sum(g for g,_ in gaps)
print('step span %.1f us, kernels %d, idle gaps total %.1f us (%d gaps), mean gap %.2f us'%((t1-t0)/1e3,len(step),tot_gap/1e3,len(gaps),tot_gap/1e3/max(len(gaps),1)))
big=sorted(gaps,reverse=True)[:12]
for g,n in big: print('  gap %.1f us before %s'%(g/1e3,n))
# time from end of last kernel of previous step to first kernel of this step
print('inter-step gap %.1f us'%((int(rows[a]['Start_Timestamp'])-int(rows[a-1]['End_Timestamp']))/1e3))
PY
