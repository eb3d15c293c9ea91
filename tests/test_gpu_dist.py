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
    # (MGX_SEED_EXCHANGE_MIN_DOCS=0: the seed-key all-gather of mgx_batch_execute_sharded — normally entered on shards of
    # a million docs and more — runs on this small table too, through RCCL)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MGX_SEED_EXCHANGE_MIN_DOCS="0",
               MGX_SEED_EXCHANGE="1", MGX_VERBOSE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "seed-bound exchange" in r.stderr, r.stderr[-2000:]


def test_seed_bound_exchange_world2():
    """mgx_batch_execute_gather with two real ranks: 1.2M-doc shards (seed items exist), the seeds' keys of both ranks
    all-gathered (gloo) and applied as every query's pruning bound; results identical to the unsharded index."""
    import os
    import subprocess
    import sys
    from pkg import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "tests", "dist_seed_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2, r.stdout
    assert "seed-bound exchange" in r.stderr and "world 2" in r.stderr, r.stderr[-2000:]
