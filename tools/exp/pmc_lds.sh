cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  d=$root/gpurun_out/pmc_lds_$(echo $set | cut -c4-12)
  rocprofv3 --kernel-trace --pmc $set -d $d -o p --output-format csv -- python $root/tools/kbench.py conv 16 48 48 128 128 3 1 5 x > /dev/null 2>&1
  python - "$d" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/p_counter_collection.csv',recursive=True)
d=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'conv_gemm' in r['Kernel_Name']: d[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in d.items(): print('%-28s %14.0f'%(k,sum(v)/len(v)))
PY
done
