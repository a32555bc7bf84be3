"""Print the full step-by-step report of tests/test_gpu_bf16_stepwise.py for some problems (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
torch.set_num_threads(min(64, os.cpu_count() or 32))
import problems as PR
import test_gpu_bf16_stepwise as T
for key in sys.argv[1:] or ['ragged__37x41__mp0', 'ragged__50x70__mp1', 'paper__paper_sc_l14__b2', 'paper__paper_mp_l0__b2', 'paper__paper_sc_l14__b16']:
    rep, res = T.stepwise(PR.REGISTRY[key]())
    print(key, T.summarize(rep), flush=True)
    for k, v in rep.items():
        if v['kind'] == 'bf16':
            if v['max_ulps'] > 1.01 or v['frac'] > 5e-3 or v['rel_l2'] > 1e-4:
                print('   bf16 %-50s ulps %.2f frac %.2e rel %.2e excess %.2e' % (k, v['max_ulps'], v['frac'], v['rel_l2'], v['excess']))
        elif v['rel_l2'] > 2e-6:
            print('   fp32 %-50s rel %.2e maxrel %.2e' % (k, v['rel_l2'], v['max_rel']))
