"""Torch-free ensemble communicator (quflow_amd/comm.py, qf_comm_* of include/quflow_hip.h).

CPU part: the id hand-out between ranks (plain sockets, a fake id), the ragged gather arithmetic of
gather_diagnostics on a communicator stand-in, and the C entry points' argument / no-device errors.
GPU part (one rank on the box's one card): a real ncclCommInitRank + all-gather + barrier through the
C ABI, and bench.py's whole distributed path over it with no torch in the process.
"""
import ctypes
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_proc(rank, world, port, q):
    sys.path.insert(0, REPO)
    from quflow_amd.comm import exchange_id
    blob = exchange_id(rank, world, "127.0.0.1", port, lambda: bytes(range(128)), timeout=30.0)
    q.put((rank, blob))


@pytest.mark.parametrize("world", [2, 4])
def test_id_handout_reaches_every_rank(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    # peers first: they must keep retrying until rank 0 listens
    procs = [ctx.Process(target=_rank_proc, args=(r, world, port, q)) for r in list(range(1, world)) + [0]]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert sorted(got) == list(range(world))
    assert all(v == bytes(range(128)) for v in got.values())


def _rank_proc_nonce(rank, world, port, q):
    sys.path.insert(0, REPO)
    from quflow_amd.comm import exchange_id
    blob = exchange_id(rank, world, "127.0.0.1", port, lambda: bytes(range(128)), timeout=30.0, nonce="c0ffee")
    q.put((rank, blob))


def test_id_handout_survives_stray_connections():
    """A local process that connects and says nothing, one that sends garbage, and one that claims a rank without
    the launcher's nonce: none of them aborts rank 0 or takes a peer's place."""
    import time
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p0 = ctx.Process(target=_rank_proc_nonce, args=(0, 2, port, q))
    p0.start()
    strays = []
    deadline = time.monotonic() + 20
    while time.monotonic() < deadline:
        try:
            s1 = socket.create_connection(("127.0.0.1", port), timeout=1.0)     # silent: rank 0's recv times out
            strays.append(s1)
            break
        except OSError:
            time.sleep(0.05)
    assert strays, "rank 0 never listened"
    s2 = socket.create_connection(("127.0.0.1", port), timeout=1.0)
    s2.sendall(b"\xff\xfe garbage\n")
    s2.close()
    s3 = socket.create_connection(("127.0.0.1", port), timeout=15.0)  # (served after the silent one's 2 s)
    s3.sendall(b"1 wrongnonce\n")
    assert s3.recv(128) == b""                                                  # refused: no id for an impostor
    s3.close()
    strays[0].close()
    p1 = ctx.Process(target=_rank_proc_nonce, args=(1, 2, port, q))
    p1.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in (p0, p1):
        p.join(timeout=30)
        assert p.exitcode == 0
    assert got[0] == got[1] == bytes(range(128))


def test_id_handout_times_out_without_rank0():
    from quflow_amd.comm import exchange_id
    with pytest.raises(TimeoutError):
        exchange_id(1, 2, "127.0.0.1", _free_port(), lambda: b"", timeout=0.5)


class _TwoRankStandIn:
    """What NativeComm.allgather_f64 returns on rank 0 of a 2-rank world whose peer holds `peer_rows`."""

    def __init__(self, peer_rows):
        self.peer = np.asarray(peer_rows, dtype=np.float64).reshape(-1, 4)

    def is_initialized(self):
        return True

    def get_world_size(self):
        return 2

    def allgather_f64(self, values):
        mine = np.asarray(values, dtype=np.float64).ravel()
        if mine.size == 1:                       # the counts round
            return np.array([[mine[0]], [float(self.peer.shape[0])]])
        other = np.zeros_like(mine)
        other[:self.peer.size] = self.peer.ravel()
        return np.stack([mine, other])


def test_ragged_gather_over_a_native_communicator():
    from quflow_amd.ensemble import gather_diagnostics
    mine = [[0.0, 1.0, 2.0, 3.0]]
    peer = [[1.0, 4.0, 5.0, 6.0], [3.0, 7.0, 8.0, 9.0]]
    out = gather_diagnostics(mine, dist=_TwoRankStandIn(peer))
    np.testing.assert_array_equal(out, np.array(mine + peer))
    # the rank that owns nothing still takes part
    out = gather_diagnostics(np.zeros((0, 4)), dist=_TwoRankStandIn(peer))
    np.testing.assert_array_equal(out, np.array(peer))


def test_even_gather_is_one_collective():
    """rows_per_rank = n: every rank owns n rows (config 4, bench.py) -- ONE all-gather per chunk, no counts round; a rank
    whose row count disagrees says so instead of corrupting the table."""
    from quflow_amd.ensemble import gather_diagnostics

    class Counting(_TwoRankStandIn):
        calls = 0

        def allgather_f64(self, values):
            Counting.calls += 1
            return super().allgather_f64(values)
    mine = [[0.0, 1.0, 2.0, 3.0], [2.0, 1.5, 2.5, 3.5]]
    peer = [[1.0, 4.0, 5.0, 6.0], [3.0, 7.0, 8.0, 9.0]]
    out = gather_diagnostics(mine, dist=Counting(peer), rows_per_rank=2)
    np.testing.assert_array_equal(out, np.array(mine + peer))
    assert Counting.calls == 1
    with pytest.raises(ValueError):
        gather_diagnostics(mine[:1], dist=Counting(peer), rows_per_rank=2)


def test_comm_entry_points_validate_before_touching_a_device():
    from quflow_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    # 1 = QF_ERR_INVALID, 2 = QF_ERR_NO_DEVICE (include/quflow_hip.h)
    blob = ctypes.create_string_buffer(128)
    assert lib.qf_comm_create(None, 0, 1, 0, blob) == 1
    assert lib.qf_comm_create(ctypes.byref(h), 0, 2, 2, blob) == 1      # rank outside the world
    assert lib.qf_comm_create(ctypes.byref(h), 0, 0, 0, blob) == 1
    assert lib.qf_comm_unique_id(None) == 1
    assert lib.qf_comm_barrier(None) == 1
    assert lib.qf_comm_allgather_f64(None, None, 1, None) == 1
    assert lib.qf_comm_destroy(None) == _lib.QF_OK
    if _lib.device_count() == 0:
        assert lib.qf_comm_create(ctypes.byref(h), 0, 1, 0, blob) == 2
        assert not h.value


@pytest.mark.gpu
def test_native_comm_one_rank_on_the_card():
    from quflow_amd.comm import NativeComm
    from quflow_amd.ensemble import gather_diagnostics
    comm = NativeComm(rank=0, world=1, device=0)
    try:
        comm.barrier()
        vals = np.arange(300, dtype=np.float64) * 0.5      # beyond the first staging size: exercises the regrow
        np.testing.assert_array_equal(comm.allgather_f64(vals), vals[None, :])
        np.testing.assert_array_equal(comm.allgather_f64([3.0]), [[3.0]])
        rows = [[0.0, 1.0, 2.0, 3.0], [1.0, 4.0, 5.0, 6.0]]
        np.testing.assert_array_equal(gather_diagnostics(rows, dist=comm), np.array(rows))
        comm.barrier()
    finally:
        comm.destroy_process_group()


@pytest.mark.gpu
def test_bench_distributed_path_without_torch(tmp_path):
    """bench.py as one launched rank with QUFLOW_BENCH_GATHER=native: barrier, gather and the max-over-ranks
    time all go through qf_comm_* and torch is never imported."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), TORCHELASTIC_RUN_ID="native-comm-test", QUFLOW_BENCH_GATHER="native",
               QUFLOW_BENCH_ASSERT_NO_TORCH="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                        "--N", "256", "--no-side-runs", "--no-config3", "--cpu-seconds", "0"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    g = line["config"]["gather"]
    assert g["backend"].startswith("rccl (native") and g["gathered_rows_ok"] and g["rccl_world_size"] == 1
    assert line["n_gpus"] == 1 and line["value"] > 0


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_one_rank(tmp_path):
    """The driver's multi-GPU launch form with one rank on the box's one card: torch is imported first (its
    bundled HIP runtime is the one libquflow_hip.so then binds to), the process group is RCCL, barrier and
    gather go through torch.distributed, and the line reports what the collective saw."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
                        "QUFLOW_BENCH_GATHER", "QUFLOW_BENCH_BACKEND")}
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--N", "256",
           "--no-side-runs", "--no-config3", "--cpu-seconds", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    g = line["config"]["gather"]
    assert g["backend"] == "nccl" and g["rccl_world_size"] == 1 and g["gathered_rows_ok"]
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["scaling"] == "weak"


@pytest.mark.gpu
def test_both_gather_routes_print_the_same_line_form():
    """The torch.distributed (RCCL) route and the torch-free route (qf_comm_*) on the box's one card: the same
    keys on the line, in `config` and in `config.gather`, the same seeds gathered -- a consumer of the line does
    not have to know which route a run took."""
    base = {k: v for k, v in os.environ.items()
            if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID",
                         "QUFLOW_BENCH_GATHER", "QUFLOW_BENCH_BACKEND")}
    args = ["--gpus", "1", "--steps", "4", "--warmup", "1", "--N", "256", "--no-side-runs", "--no-config3",
            "--cpu-seconds", "0"]
    lines = {}
    for route in ("torch", "native"):
        env = dict(base, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
                   TORCHELASTIC_RUN_ID="route-test", QUFLOW_BENCH_GATHER=route)
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True, text=True,
                           timeout=600, cwd=REPO)
        assert r.returncode == 0, (route, r.stderr[-2000:])
        lines[route] = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    a, b = lines["torch"], lines["native"]
    assert sorted(a) == sorted(b)
    assert sorted(a["config"]) == sorted(b["config"])
    assert sorted(a["config"]["gather"]) == sorted(b["config"]["gather"])
    for key in ("rccl_world_size", "gathered_rows_ok", "seeds_gathered"):
        assert a["config"]["gather"][key] == b["config"]["gather"][key], key
    assert a["config"]["gathered_rows"] == b["config"]["gathered_rows"] == 1
    assert a["config"]["iterations_per_step"] == b["config"]["iterations_per_step"]
    assert len(a["config"]["per_rank_timesteps_per_s"]) == len(b["config"]["per_rank_timesteps_per_s"]) == 1
