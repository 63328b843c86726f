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


def test_self_launch_eight_ranks_gloo(tmp_path, oracle):
    """The N = 8 form the driver's scaling run uses (bench.py starts its own ranks): eight seeds, eight rows
    gathered, eight per-rank rates; every rank pinned to its own slice of the cpus when there are enough."""
    res = run_bench(tmp_path, ["--gpus", "8", "--steps", "2", "--warmup", "1", "--N", "16", "--cpu-seconds", "0"],
                    QUFLOW_BENCH_FAKE_GPUS="8")
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    g = d["config"]["gather"]
    assert g["rccl_world_size"] == 8 and g["gathered_rows_ok"] is True
    assert g["seeds_gathered"] == list(range(8)) and d["config"]["gathered_rows"] == 8
    rates = d["config"]["per_rank_timesteps_per_s"]
    assert len(rates) == 8 and all(r > 0 for r in rates)
    assert abs(d["value"] - 8 * min(rates)) <= 1e-9 * d["value"]
    # round 5: first contact with an 8-GPU node diagnoses itself -- one row per rank in the line (device bound, PCI bus id,
    # rate, outcome of the clock warm-up, pid) and one line per rank on stderr saying where it landed
    rows = d["config"]["ranks"]
    assert [r["rank"] for r in rows] == list(range(8)) and len({r["pid"] for r in rows}) == 8
    assert all(set(r) >= {"hip_device", "pci_bus_id", "timesteps_per_s", "prewarm_last_chunk_timesteps_per_s"} for r in rows)
    assert all(abs(r["timesteps_per_s"] - x) <= 1e-9 * x for r, x in zip(rows, rates))
    for r in range(8):
        assert ("bench.py: rank %d/8 pid" % r) in res.stderr
    ncpu = len(os.sched_getaffinity(0))
    if ncpu >= 16:
        assert d["config"]["rank0_cpus_pinned"] == ncpu // 8
    else:
        assert d["config"]["rank0_cpus_pinned"] is None        # too few cpus to give every rank two: left alone


def test_pin_rank_cpus_splits_the_affinity_mask(tmp_path):
    code = r'''
import json, os, sys
sys.path.insert(0, os.environ["QF_REPO"])
import bench
cpus = sorted(os.sched_getaffinity(0))
out = {"cpus": cpus}
os.environ.pop("QUFLOW_BENCH_CPUS", None)
out["one_rank"] = bench.pin_rank_cpus(0, 1)
got = bench.pin_rank_cpus(1, 2)
out["rank1_of_2"] = got
out["now"] = sorted(os.sched_getaffinity(0))
print(json.dumps(out))
'''
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, QF_REPO=REPO), capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    cpus = out["cpus"]
    assert out["one_rank"] is None
    if len(cpus) >= 4:
        per = len(cpus) // 2
        assert out["rank1_of_2"] == cpus[per:2 * per] and out["now"] == cpus[per:2 * per]
    else:
        assert out["rank1_of_2"] is None and out["now"] == cpus


def test_launcher_ends_all_ranks_when_one_fails(tmp_path):
    """A rank that dies early (here: rank 1's trajectory class raises at construction) must not leave the others
    in the rendezvous: the launcher terminates them and returns non-zero within its supervision loop."""
    bad = tmp_path / "bad_traj.py"
    bad.write_text(TRAJ + r'''

class FailsOnRankOne(CpuTrajectory):
    def __init__(self, W0):
        if os.environ.get("RANK") == "1":
            raise SystemExit(7)
        super().__init__(W0)

    def advance(self, dt, steps, **kw):
        import time
        time.sleep(0.2)
        return super().advance(dt, steps, **kw)
''')
    import time as _time
    t0 = _time.monotonic()
    res = run_bench(tmp_path, ["--gpus", "2", "--steps", "2", "--warmup", "1", "--N", "16", "--cpu-seconds", "0"],
                    QUFLOW_BENCH_FAKE_GPUS="2", QUFLOW_BENCH_TRAJECTORY="%s:FailsOnRankOne" % bad)
    assert res.returncode != 0
    assert "rank(s) failed" in res.stderr and "(1, 7)" in res.stderr
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert _time.monotonic() - t0 < 120


def test_rank_names_itself_when_the_rendezvous_fails(tmp_path):
    """A rank whose rendezvous cannot complete (here: a one-rank launch told to expect two, with a short time-out) says
    which rank failed at which stage and exits non-zero -- no JSON line, nothing left running."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    res = run_bench(tmp_path, ["--gpus", "2", "--steps", "2", "--warmup", "0", "--N", "16", "--cpu-seconds", "0"],
                    WORLD_SIZE="2", RANK="1", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                    QUFLOW_BENCH_RDZV_TIMEOUT="5")
    assert res.returncode == 3, res.stderr[-3000:]
    assert "RANK 1 of 2 FAILED at: rendezvous" in res.stderr
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")]


def test_pci_bus_id_round_trip():
    sys.path.insert(0, REPO)
    import importlib
    bench = importlib.import_module("bench")
    for bus in ("0000:05:00.0", "0002:c3:1f.7", "ffff:ff:00.1"):
        assert bench.unpack_pci(float(bench.pack_pci(bus))) == bus
    assert bench.pack_pci(None) == -1 and bench.unpack_pci(-1.0) is None


def test_rehearsal_mode_marks_its_line(tmp_path, oracle):
    """QUFLOW_BENCH_REHEARSAL=share-gpu (the N-rank flow on a box with fewer GPUs than ranks): the launcher starts more ranks
    than there are GPUs, the gather goes over gloo whatever backend was asked for, and the line says it is a rehearsal -- while
    a normal launch carries config.rehearsal = None and still refuses to run short-handed."""
    res = run_bench(tmp_path, ["--gpus", "3", "--steps", "2", "--warmup", "1", "--N", "16", "--cpu-seconds", "0"],
                    QUFLOW_BENCH_FAKE_GPUS="1", QUFLOW_BENCH_REHEARSAL="share-gpu", QUFLOW_BENCH_BACKEND="nccl")
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 3 and d["config"]["gather"]["backend"] == "gloo" and d["config"]["gather"]["seeds_gathered"] == [0, 1, 2]
    assert "NOT an N-GPU measurement" in d["config"]["rehearsal"] and d["config"]["rehearsal"].startswith("3 ranks")
    plain = run_bench(tmp_path, ["--gpus", "2", "--steps", "2", "--warmup", "1", "--N", "16", "--cpu-seconds", "0"], QUFLOW_BENCH_FAKE_GPUS="2")
    assert plain.returncode == 0
    assert json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][0])["config"]["rehearsal"] is None
    short = run_bench(tmp_path, ["--gpus", "3", "--steps", "2", "--warmup", "1", "--N", "16", "--cpu-seconds", "0"], QUFLOW_BENCH_FAKE_GPUS="1")
    assert short.returncode == 2 and "only 1 GPU(s) visible" in short.stderr
