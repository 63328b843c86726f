"""World-size-2 `gloo` test of the ensemble path (CPU).  The replica sharding and the
diagnostics gather of quflow_amd.ensemble are exercised end to end; the per-replica
stepper is injected from the CPU oracle because no GPU exists here (tests only -- the
product default is the HIP DeviceTrajectory)."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import REPO

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["QF_REPO"])
import torch
import torch.distributed as dist
from quflow_amd import ensemble
from oracle import isomp_oracle as oracle

class CpuTrajectory:
    """Oracle-backed stand-in with the DeviceTrajectory interface (tests only)."""
    def __init__(self, W0):
        self.W = W0.copy()
    def advance(self, dt, steps, **kw):
        stats = {"iterations": 0.0}
        self.W = oracle.isomp(self.W, dt, steps=steps, stats=stats, **kw)
        return stats
    def diagnostics(self):
        return oracle.energy_euler(self.W), oracle.enstrophy(self.W)

dist.init_process_group("gloo")
N = 16
dt = 0.25 * oracle.hbar(N)
hist, trajs = ensemble.run_ensemble(N, seeds=[0, 1, 2], dt=dt, steps=6, steps_out=3, dist=dist,
                                    trajectory_factory=CpuTrajectory)
np.save(os.path.join(os.environ["QF_OUT"], "hist_rank%d.npy" % dist.get_rank()), np.stack(hist))
np.save(os.path.join(os.environ["QF_OUT"], "seeds_rank%d.npy" % dist.get_rank()),
        np.array([s for s, _ in trajs]))
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_ensemble_world2_gloo(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, QF_REPO=REPO, QF_OUT=str(tmp_path), OMP_NUM_THREADS="1",
               OPENBLAS_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    h0 = np.load(tmp_path / "hist_rank0.npy")
    h1 = np.load(tmp_path / "hist_rank1.npy")
    np.testing.assert_array_equal(h0, h1)                 # every rank holds the full table
    assert h0.shape == (2, 3, 4)                          # 2 chunks x 3 replicas x 4 columns
    assert list(np.load(tmp_path / "seeds_rank0.npy")) == [0, 2]
    assert list(np.load(tmp_path / "seeds_rank1.npy")) == [1]
    # rows sorted by seed, values equal to a serial run of each replica
    N = 16
    dt = 0.25 * oracle.hbar(N)
    for seed in range(3):
        W = oracle.make_W0(N, seed)
        for chunk in range(2):
            stats = {"iterations": 0.0}
            W = oracle.isomp(W, dt, steps=3, stats=stats)
            row = h0[chunk, seed]
            assert row[0] == seed
            np.testing.assert_allclose(row[1], oracle.energy_euler(W), rtol=1e-12)
            np.testing.assert_allclose(row[2], oracle.enstrophy(W), rtol=1e-12)
            assert row[3] == stats["iterations"]
