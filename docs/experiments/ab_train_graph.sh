for g in 1 0 1 0; do DFL_TRAIN_GRAPH=$g python bench.py --no-cpu-baseline --no-fp32-reference --no-fwd --no-configs3 --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('graph $g', d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"; done
