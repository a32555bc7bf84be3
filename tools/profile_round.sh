#!/bin/bash
# Run ON the MI355X box (gpurun): rocprofv3 evidence for one round.  Usage: tools/profile_round.sh r02 [extra bench.py arguments]
#  1. kernel trace + stats of the default bench workload
#  2. HBM traffic counters, one pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters only, no API traces)
#  3. SQ counters (wave / wait / MFMA-busy cycles)
set -u
tag=${1:-r01}
shift || true
extra="$*"
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
bench="python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-reference --no-fwd $extra"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o t --output-format csv -- $bench > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_$c" -o p --output-format csv -- $bench --no-profile > /dev/null 2> "$out/pmc_$c.err"
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
  -d "$out/pmc_SQ" -o p --output-format csv -- $bench --no-profile > /dev/null 2> "$out/pmc_SQ.err"
cd "$root"
python tools/summarize_profile.py "$out" "$tag"
