"""-m gpu: two ranks sharing the box's single GPU run the sharded search end to end (gloo exchange, device merge)."""
import pytest

from test_dist_cpu import _launch

pytestmark = pytest.mark.gpu


def test_sharded_search_world2_on_one_gpu():
    _launch("gpu")


def test_rccl_exchange_one_rank():
    """The nccl branch executed for real: one rank, RCCL all-gather / all-reduce inside the library, Python and C++ hosts."""
    import os
    import subprocess
    import sys
    from pkg import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
