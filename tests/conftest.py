import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))


@pytest.fixture(scope='session')
def golden():
    return load_golden


# kwargs of every tiny fixture written by tools/gen_golden.py
TINY_BASE = dict(n_classes=7, depth=3, wf=2, batch_norm=True, padding=True, do_res=True, block_depth=2)
TINY_CFGS = {
    'tiny_sc_l14': dict(TINY_BASE, max_pool=False, num_lands=14),
    'tiny_mp_l0': dict(TINY_BASE, max_pool=True, num_lands=0),
    'tiny_mp_l14': dict(TINY_BASE, max_pool=True, num_lands=14),
    'tiny_valid_nores': dict(n_classes=7, depth=2, wf=2, batch_norm=True, padding=False, do_res=False,
                             block_depth=2, max_pool=True, num_lands=0),
    'tiny_nobn_d1': dict(n_classes=3, depth=2, wf=3, batch_norm=False, padding=True, do_res=True,
                         block_depth=1, max_pool=False, num_lands=14),
    'tiny_nobn_nores': dict(n_classes=3, depth=2, wf=3, batch_norm=False, padding=True, do_res=False,
                            block_depth=2, max_pool=False, num_lands=0),
    'tiny_bd3_nosm': dict(n_classes=5, depth=2, wf=3, batch_norm=True, padding=True, do_res=True,
                          block_depth=3, max_pool=True, num_lands=0, do_soft_max=False),
}
PAPER_CFGS = {
    'paper_sc_l14': (1234, dict(n_classes=7, depth=6, wf=5, batch_norm=True, padding=True, max_pool=False,
                                num_lands=14, do_res=True, block_depth=2)),
    'paper_mp_l0': (1235, dict(n_classes=7, depth=6, wf=5, batch_norm=True, padding=True, max_pool=True,
                               num_lands=0, do_res=True, block_depth=2)),
    # BASELINE configs[0]'s program (train.py:326-327: strided convolutions, segmentation head only, DiceLoss2D alone) at its batch 4
    'paper_sc_l0': (1236, dict(n_classes=7, depth=6, wf=5, batch_norm=True, padding=True, max_pool=False,
                               num_lands=0, do_res=True, block_depth=2)),
}
PAPER_BATCH = {'paper_sc_l0': 4}          # batch of each paper fixture (default 2)


def paper_key(name):
    return 'paper__%s__b%d' % (name, PAPER_BATCH.get(name, 2))


# ---- product arithmetic of the GEMM kernels (include/dfl_hip.h: dfl_set_math_mode) -------------------------------------
# 'fp32' is the parity gate: the tolerances written in the tests are for it.  'bf16x3' (split-bf16 products, 2^-16 per
# product) must keep every FORWARD bar (1e-4) and gets documented, looser bars where fp32 rounding noise is amplified
# (gradients through BatchNorm cancellations, arg-max at margins below 1e-4, chaotic 30-step trajectories).
MATH_MODES = ['fp32', 'bf16x3']


@pytest.fixture(params=MATH_MODES)
def math_mode(request):
    from dfl_amd import _native as nat
    lib = nat.lib()
    nat.check(lib.dfl_set_math_mode(MATH_MODES.index(request.param)), 'dfl_set_math_mode')
    yield request.param
    nat.check(lib.dfl_set_math_mode(0), 'dfl_set_math_mode')


def by_mode(mode, fp32, bf16x3):
    return fp32 if mode == 'fp32' else bf16x3


@pytest.fixture(autouse=True, scope='module')
def _free_between_modules():
    """The GPU suite runs in one process: the fp64 oracles and plans a module cached (300 MB per paper-preset oracle, GBs of plan
    buffers at 768 x 768 / 1440 x 1440) are dropped when it is done."""
    yield
    import gc
    try:
        import noise_floor as NF
        NF._CHECKS.clear()
    except Exception:
        pass
    for name in ('test_gpu_fullsize',):
        mod = sys.modules.get(name)
        if mod is not None and hasattr(mod, '_C4'):
            mod._C4.clear()
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    except Exception:
        pass
