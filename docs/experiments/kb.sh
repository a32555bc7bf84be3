for w in 4096 3072 2048 1536; do export DFL_EXP_WAVES=$w; echo "WAVES $w"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile | cut -c150-260; done
