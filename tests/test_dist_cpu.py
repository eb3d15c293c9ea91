"""world_size-2 gloo rehearsal of the doc-range sharding (CPU only; see tests/dist_worker.py)."""
import os
import subprocess
import sys

from pkg import ROOT


def _launch(mode, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "tests", "dist_worker.py"), mode]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2, r.stdout


def test_sharding_design_world2_gloo():
    _launch("cpu")
