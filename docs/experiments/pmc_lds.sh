# LDS / issue counters of one conv layer (level 2 of the paper net) per kernel variant.  Usage (on the GPU box):
#   DFL_MATH=bf16x3 bash docs/experiments/pmc_lds.sh "sbrW sabrW WX"        (KIND=wgrad SHAPE="16 48 48 128 128 3" for the weight gradient)
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for flags in ${1:-sabr}; do
echo "== flags $flags (DFL_MATH=${DFL_MATH:-0})"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  d=$root/gpurun_out/pmc_lds_${flags}_$(echo $set | cut -c4-12)
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $set -d $d -o p --output-format csv -- python $root/tools/kbench.py ${KIND:-conv} ${SHAPE:-16 48 48 128 128 3} 1 5 $flags > /dev/null 2>&1
  python - "$d" <<'PY'
import csv,glob,sys,collections,os
KN='wgrad_kernel' if os.environ.get('KIND')=='wgrad' else 'conv_gemm'
f=glob.glob(sys.argv[1]+'/**/p_counter_collection.csv',recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if KN in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items(): print('%-28s %14.0f'%(k,sum(v)/len(v)))
t=glob.glob(sys.argv[1]+'/**/p_kernel_trace.csv',recursive=True)
du=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in csv.DictReader(open(t[0])) if KN in r['Kernel_Name']]
print('%-28s %14.0f ns (%s)'%('duration',sum(du)/len(du),[r['Kernel_Name'][:70] for r in csv.DictReader(open(t[0])) if KN in r['Kernel_Name']][0]))
PY
done
done
