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
bench="python $root/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-reference --no-fwd --no-configs3 $extra"
rocprofv3 --kernel-trace --stats -d "$out/trace" -o t --output-format csv -- $bench > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d "$out/pmc_$c" -o p --output-format csv -- $bench --no-profile > /dev/null 2> "$out/pmc_$c.err"
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU \
  -d "$out/pmc_SQ" -o p --output-format csv -- $bench --no-profile > /dev/null 2> "$out/pmc_SQ.err"
# 4. the other BASELINE shapes (configs[3] 768x768 training, configs[4] 1440x1440 ensemble inference): kernel trace + stats
for c in 3 4; do
  mkdir -p "$out/cfg$c"
  DFL_MATH=${DFL_MATH_OTHER:-bf16s} rocprofv3 --kernel-trace --stats -d "$out/cfg$c/trace" -o t --output-format csv -- python $root/tools/bench_other.py $c > "$out/cfg$c/bench_other.json" 2> "$out/cfg$c/trace.err"
done
# 5. the parity-holding arithmetics (fp32 MFMA = the 1e-4 gate, bf16x3): kernel trace + stats of the same workload
for m in fp32 bf16x3; do
  mkdir -p "$out/$m"
  rocprofv3 --kernel-trace --stats -d "$out/$m/trace" -o t --output-format csv -- $bench --math $m > "$out/$m/bench_under_rocprof.json" 2> "$out/$m/trace.err"
done
cd "$root"
python tools/summarize_profile.py "$out" "$tag"
for m in fp32 bf16x3; do
  python tools/summarize_profile.py "$out/$m" "${tag}_$m"
  cp "$out/$m/${tag}_${m}_kernel_stats.csv" "$out/"
  cp "$out/$m/bench_under_rocprof.json" "$out/${tag}_${m}_bench_under_rocprof.json"
done
for c in 3 4; do
  python tools/summarize_profile.py "$out/cfg$c" "${tag}_cfg$c"
  cp "$out/cfg$c/bench_other.json" "$out/${tag}_cfg${c}_bench.json"
  cp "$out/cfg$c/${tag}_cfg${c}_kernel_stats.csv" "$out/"
done
# gpurun copies at most 64 MiB back: keep the summaries and the raw --stats tables, drop the per-dispatch traces and counter dumps
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "$out/${tag}_rocprofv3_kernel_stats_raw.csv" 2>/dev/null || true
find "$out" -name '*kernel_trace.csv' -delete 2>/dev/null || true
find "$out" -name '*counter_collection.csv' -delete 2>/dev/null || true
find "$out" -name '*.db' -delete 2>/dev/null || true
# 6. HBM counters of the other BASELINE shapes (separate passes again)
bash tools/profile_other_pmc.sh "$tag" > "$out/other_pmc.log" 2>&1 || true
du -sh "$out" || true
