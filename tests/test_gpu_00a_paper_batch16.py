"""North-star: BASELINE configs[1] itself -- the paper preset with both heads at batch 16, the step bench.py times -- forward, loss and
every gradient against the fp64 oracle in the two parity arithmetics and in the bf16 STORAGE arithmetic of the headline (there also
step by step against oracle/bf16_emu.py).  Bars: tests/paper_gradient.py.  pytest -m gpu."""
import pytest

from paper_gradient import paper_gradient

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode', ['fp32', 'bf16x3', 'bf16s'])
def test_paper_batch16_gradient(mode):
    paper_gradient(mode, 16)
