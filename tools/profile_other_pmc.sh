#!/bin/bash
# Run ON the MI355X box (gpurun), behind tools/profile_round.sh: HBM traffic counters of BASELINE configs[3] / configs[4]
# (tools/bench_other.py 3 | 4), one counter per pass (kernel trace + counters only), summarised like the default workload's.
# Usage: tools/profile_other_pmc.sh r06
set -u
tag=${1:-r01}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
for c in 3 4; do
  mkdir -p "$out/cfg$c"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    DFL_MATH=bf16s rocprofv3 --kernel-trace --pmc $ctr -d "$out/cfg$c/pmc_$ctr" -o p --output-format csv -- python $root/tools/bench_other.py $c > /dev/null 2> "$out/cfg$c/pmc_$ctr.err"
  done
done
cd "$root"
for c in 3 4; do
  python tools/summarize_profile.py "$out/cfg$c" "${tag}_cfg$c"
  cp "$out/cfg$c/${tag}_cfg${c}_traffic.json" "$out/cfg$c/${tag}_cfg${c}_pmc_summary.csv" "$out/cfg$c/${tag}_cfg${c}_kernel_stats.csv" "$out/" 2>/dev/null
done
find "$out" -name '*kernel_trace.csv' -delete 2>/dev/null || true
find "$out" -name '*counter_collection.csv' -delete 2>/dev/null || true
find "$out" -name '*.db' -delete 2>/dev/null || true
