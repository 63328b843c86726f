"""bench.py's own launch path (`python bench.py --gpus N` without torch.distributed.run) and its
gather report, on CPU: two ranks, gloo instead of RCCL, and a trajectory injected from the CPU oracle
(tests only -- a measurement never sets QUFLOW_BENCH_TRAJECTORY; the line then says so in `data`)."""
import json
import os
import subprocess
import sys

from conftest import REPO

TRAJ = r'''
import os, sys
sys.path.insert(0, os.environ["QF_REPO"])
from oracle import isomp_oracle as oracle


class CpuTrajectory:
    """Oracle-backed stand-in with the DeviceTrajectory interface (tests only)."""
    def __init__(self, W0):
        self.W = W0.copy()

    def advance(self, dt, steps, **kw):
        stats = {"iterations": 0.0}
        self.W = oracle.isomp(self.W, dt, steps=steps, stats=stats, **kw)
        stats["total_iterations"] = int(round(stats["iterations"] * steps))
        return stats

    def diagnostics(self):
        return oracle.energy_euler(self.W), oracle.enstrophy(self.W)

    def sync(self):
        pass
'''


def run_bench(tmp_path, argv, **env_extra):
    traj = tmp_path / "cpu_traj.py"
    traj.write_text(TRAJ)
    env = dict(os.environ, QF_REPO=REPO, QUFLOW_BENCH_BACKEND="gloo",
               QUFLOW_BENCH_TRAJECTORY="%s:CpuTrajectory" % traj, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + argv, env=env, capture_output=True,
                          text=True, timeout=600)


def test_self_launch_two_ranks_gloo(tmp_path, oracle):
    res = run_bench(tmp_path, ["--gpus", "2", "--steps", "3", "--warmup", "1", "--N", "16", "--cpu-seconds", "0"],
                    QUFLOW_BENCH_FAKE_GPUS="2")
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                 # rank 0 prints ONE line, the launcher passes it through
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["unit"] == "timesteps/s"
    g = d["config"]["gather"]
    assert g["backend"] == "gloo" and g["rccl_world_size"] == 2
    assert g["gathered_rows_ok"] is True and g["seeds_gathered"] == [0, 1]
    assert d["config"]["gathered_rows"] == 2
    rates = d["config"]["per_rank_timesteps_per_s"]
    assert len(rates) == 2 and all(r > 0 for r in rates)
    # whole-job value = ranks x steps / slowest rank's time
    assert abs(d["value"] - 2 * min(rates)) <= 1e-9 * d["value"]
    assert d["data"].startswith("injected")
    # the same form as the single-process line
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d


def test_single_process_line_has_the_same_form(tmp_path, oracle):
    res = run_bench(tmp_path, ["--steps", "2", "--warmup", "1", "--N", "16", "--cpu-seconds", "0"])
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["gather"]["rccl_world_size"] == 1
    assert d["config"]["gather"]["gathered_rows_ok"] is True


def test_launcher_refuses_when_gpus_are_missing(tmp_path):
    res = run_bench(tmp_path, ["--gpus", "2", "--steps", "2", "--warmup", "0", "--N", "16", "--cpu-seconds", "0"],
                    QUFLOW_BENCH_FAKE_GPUS="1")
    assert res.returncode != 0
    assert "only 1 GPU" in res.stderr
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]     # never a line that claims fewer GPUs


def test_rank_refuses_a_world_size_mismatch(tmp_path):
    res = run_bench(tmp_path, ["--gpus", "2", "--steps", "2", "--warmup", "0", "--N", "16", "--cpu-seconds", "0"],
                    WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
