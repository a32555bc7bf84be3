"""North-star: the paper's literal batch (train_test_code/Readme.md:16: --batch-size 5): 5 x 6 x 6 = 180 pixels at level 5 -- not a
multiple of 16 -- in every patch, K-slice and tile decision.  Same bars as the batch-16 step (tests/paper_gradient.py).  pytest -m gpu."""
import pytest

from paper_gradient import paper_gradient

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode', ['fp32', 'bf16s'])
def test_paper_batch5_gradient(mode):
    paper_gradient(mode, 5)
