"""-m gpu: two ranks sharing the box's single GPU run the sharded search end to end (gloo exchange, device merge)."""
import pytest

from test_dist_cpu import _launch

pytestmark = pytest.mark.gpu


def test_sharded_search_world2_on_one_gpu():
    _launch("gpu")
