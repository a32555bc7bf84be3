cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
d=$root/gpurun_out/pmc_clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $d -o p --output-format csv -- python $root/tools/kbench.py conv 16 48 48 128 128 3 1 20 x > /dev/null 2>&1
python - "$d" <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/p_counter_collection.csv',recursive=True)
for r in csv.DictReader(open(f[0])):
    if 'conv_gemm' in r['Kernel_Name']:
        dur=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
        print(r['Counter_Name'], r['Counter_Value'], 'dur_ns', dur, 'cycles/ns %.3f'%(float(r['Counter_Value'])/dur))
PY
rocm-smi --showclocks 2>/dev/null | head -20
