export DFL_MATH=bf16x3
for w in 1024 512 256 128; do export DFL_EXP_WANT=$w; echo "WANT $w"; for s in "16 48 48 128 128 3" "16 96 96 64 64 3" "16 24 24 256 256 3" "16 12 12 512 512 3" "16 48 48 256 128 3" "16 96 96 128 64 3"; do python tools/kbench.py conv $s 1 50 sabr; done; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile | cut -c150-260; done
