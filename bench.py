#!/usr/bin/env python3
"""Headline benchmark: isospectral time steps per second (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W

A "step" is ONE isospectral-midpoint time step (quflow's `isomp`, adaptive fixed-point
iteration with the reference defaults tol='auto', maxit=10, minit=1) of the su(N)
vorticity flow at N=1024 (BASELINE.json configs[2]/[3] size), dt = 0.25*hbar, synthetic
random skew-Hermitian trace-free W0 (make_W0(N, seed), seed = rank).  With N GPUs each
rank advances its own independent initial condition (weak scaling, replicas only --
SURVEY.md 8e) and the ranks all_gather (energy, enstrophy, iterations) over RCCL at
the end of the chunk; `value` = n_gpus * K / max-over-ranks wall time.

Launch: under `python -m torch.distributed.run` the ranks are given (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* from the environment).  Started plainly with --gpus N > 1, this process
starts the N ranks itself as child processes (one per GPU, 127.0.0.1 rendezvous) BEFORE it touches
the GPU, and fails if fewer than N devices are visible -- it never reports a 1-GPU run as N.

The state is resident in HBM before the timed region starts; the timed region is
bracketed by a barrier + device synchronisation on both sides.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (first product of the complex GEMM pair, fp64 MFMA):
                  algorithmic flops per launch (8 N^3) / mean launch duration measured
                  with HIP events on the launch stream during the timed region (`frac`), next to the
                  flops the 3M kernel EXECUTES (6 N^3: `executed_frac`), the other two kernels of an
                  iteration (instrumented pass after the timed region), the whole step against its
                  bound (`whole_step`) and the fixed-iteration protocol of the reference's profiler.
  cpu_baseline -- the CPU oracle (oracle/isomp_oracle.py: numpy zgemm + OpenMP Thomas),
                  timed on this host's cores on a bounded sample of the same workload
                  (rank 0, --gpus 1 only).  The oracle is the checker/baseline, never the
                  thing measured as `value`.
"""
import argparse
import ctypes
import importlib.util
import json
import os
import socket
import subprocess
import sys
import time

# multi-process GPU work on this pool (RCCL, sharing device memory across ranks) needs the dmabuf IPC mode: the host driver does
# not support the legacy one (hipIpcGetMemHandle: invalid argument).  The image exports it; a launcher that scrubs the
# environment must not take it away from the ranks.  Set before anything initialises the HIP runtime.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _limit_host_thread_pools():
    """numpy's OpenBLAS (and OpenMP) size their pools by the VISIBLE cpus (256 on the GPU host)
    and their idle workers spin.  Inside a CPU-quota cgroup (cpu.max: 16 cores on the GPU box)
    that burns the whole quota within a scheduler period and the kernel then freezes EVERY
    thread of the process -- the stepper's launch/poll thread included -- for the rest of the
    100 ms period: measured 70-80 ms stalls in the timed region in ~half of the runs.  Keep
    the pools within the quota (must happen before numpy is imported)."""
    limit = 16
    # ranks of one node share the cgroup: split the quota, keep one core per rank for the
    # stepper's launch/poll thread
    ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            limit = max(1, min(limit, int(quota) // int(period) // ranks - (1 if ranks > 1 else 0)))
    except Exception:
        pass
    if hasattr(os, "sched_getaffinity"):
        limit = max(1, min(limit, len(os.sched_getaffinity(0))))
    if os.environ.get("QUFLOW_BENCH_CPUS"):      # this rank's slice (pin_rank_cpus): one cpu stays with the poll thread
        limit = max(1, min(limit, len(os.environ["QUFLOW_BENCH_CPUS"].split(",")) - 1))
    for var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(var, str(limit))
    return limit


HOST_THREADS = _limit_host_thread_pools()

METRIC = "isospectral timesteps/sec at N=1024 (1 GPU) + ensemble steps/sec at 1/2/4/8 GPUs"
# MI355X fp64 matrix peak (datasheet, dense): 256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz.
# MI355X_MICROARCH.md lists no f64 MFMA row; tools/mfma_clock.hip measures the issue rate (64 clk) and
# tools/launch_probe.hip the clock the chip holds under this load (~2.3 GHz: 75 TFLOP/s attainable).
PEAK_FP64_MFMA_TFLOPS = 78.6
PEAK_FP32_MFMA_TFLOPS = 157.3    # v_mfma_f32_32x32x2_f32: 64 flop/clk/SIMD (MI355X_MICROARCH.md, chip-level parameters)
PEAK_HBM_GBS = 8000.0
EVENT_STRIDE = 8      # per-launch HIP events of the timed region: one first-product launch in 8 is bracketed
# int8 matrix peak, dense: v_mfma_i32_32x32x32_i8 = 65,536 ops / 32 clk / SIMD = 2 x the bf16 rate
# (MI355X_MICROARCH.md, MFMA table, I8 row): 256 x 4 x 2048 x 2.4 GHz
PEAK_I8_MFMA_TOPS = 5033.0
I8_OPS_PER_PRODUCT = 45 * 2.0      # per N^3: 15 digit pairs x 3 real products (ozaki.hip), 2 ops per MAC


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--N", type=int, default=1024, help="matrix size (headline: 1024)")
    ap.add_argument("--ic", choices=["A", "B"], default="A", help="A: white-noise make_W0; B: smoothed")
    ap.add_argument("--stepsize", type=float, default=0.25, help="dt = stepsize * hbar(N)")
    ap.add_argument("--fixed-iters", type=int, default=0, help="minit=maxit=K (roofline mode); 0 = adaptive")
    ap.add_argument("--compsum", action="store_true")
    ap.add_argument("--stepper", choices=["isomp", "euler", "heun", "rk4", "isomp_simple", "isomp_quasinewton"], default="isomp",
                    help="isomp = the headline metric; the explicit steppers (SURVEY.md 8f) are extra lines")
    ap.add_argument("--products", choices=["f64", "i8", "i8x6", "i8h", "i8hx6", "i8x6f", "i8x65"], default="f64",
                    help="f64: both commutator products on the fp64 matrix cores (headline, full parity); "
                         "i8x6: BASELINE.json config 3 -- digit-split products (6 base-128 digits) on the int8 matrix "
                         "cores + fp64 Laplacian, fp64-fixture parity; i8: the 5-digit variant (faster, drift above the "
                         "fp64 run's: a demonstration, not config 3's acceptance line); i8h / i8hx6: hybrid -- first "
                         "product on the fp64 matrix cores, only the second one digit-split (5 / 6 digits)")
    ap.add_argument("--dtype", choices=["c128", "c64"], default="c128",
                    help="c128: the headline (complex128 state, fp64 arithmetic).  c64: a complex64 state advanced in "
                         "single precision as the reference does with complex64 input (float32 Poisson solve, complex64 "
                         "products on the fp32 matrix cores, roofline against the 157.3 TFLOP/s fp32 MFMA peak)")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the short int8-products side measurement the default single-GPU run appends")
    ap.add_argument("--no-per-call", action="store_true",
                    help="skip the per-call timings of the public host-array API (solve_poisson / laplace / commutator)")
    ap.add_argument("--no-side-runs", action="store_true",
                    help="skip the instrumented pass, the fixed-iteration protocol and the other sizes")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="GPU clock warm-up before the W warm-up steps: a scratch trajectory of the same workload "
                         "is advanced for this long (a fresh process finds the GPU idle; its clock takes ~100 ms of "
                         "load to come up).  0 disables")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget; 0 disables")
    ap.add_argument("--cpu-cores", type=int, default=16, help="threads for the CPU baseline (BLAS + OpenMP)")
    ap.add_argument("--no-kernel-events", action="store_true", help="no per-launch HIP events in the timed region")
    ap.add_argument("--kernel-table", action="store_true",
                    help="HIP events around every hot-path launch; per-kernel table on stderr (diagnostic)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------
# launch: N ranks of one node
# ------------------------------------------------------------------------------------------------

def visible_gpu_count():
    """GPUs this process may use, WITHOUT initialising the HIP runtime (the launcher must not touch
    the GPU before it starts its children): KFD topology nodes with SIMDs, narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES."""
    forced = os.environ.get("QUFLOW_BENCH_FAKE_GPUS")       # tests of the launcher on a CPU-only machine
    if forced is not None:
        return int(forced)
    n = 0
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(root):
            try:
                props = open(os.path.join(root, node, "properties")).read().split()
            except OSError:
                continue
            kv = dict(zip(props[0::2], props[1::2]))
            if int(kv.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        n = 0
    if n == 0:
        # no KFD topology in this sandbox: count the render nodes instead (-1 = cannot tell)
        try:
            n = len([d for d in os.listdir("/dev/dri") if d.startswith("renderD")])
        except OSError:
            n = 0
        if n == 0:
            return -1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def pin_rank_cpus(local_rank, local_world):
    """Per-rank CPU pinning: each rank's launch / poll thread busy-spins and the ranks of a node share one cgroup
    quota, so every rank gets its own slice of the cpus it may use -- QUFLOW_BENCH_CPUS from bench.py's own
    launcher, else (under torch.distributed.run) an equal split of the affinity mask by LOCAL_RANK.  Returns the
    cpus the rank ended up on (None: unchanged -- one rank, too few cpus, QUFLOW_BENCH_PIN=0, or not Linux)."""
    if not hasattr(os, "sched_setaffinity") or os.environ.get("QUFLOW_BENCH_PIN", "1") == "0":
        return None
    try:
        spec = os.environ.get("QUFLOW_BENCH_CPUS")
        if spec:
            mine = [int(c) for c in spec.split(",") if c.strip() != ""]
        else:
            if local_world <= 1:
                return None
            cpus = sorted(os.sched_getaffinity(0))
            if len(cpus) < 2 * local_world:
                return None
            per = len(cpus) // local_world
            mine = cpus[local_rank * per:(local_rank + 1) * per]
        if not mine:
            return None
        os.sched_setaffinity(0, mine)
        return mine
    except (OSError, ValueError):
        return None


def rank_fail(rank, world, stage, exc):
    """A rank that cannot take part names itself and the stage on stderr and ends with a status of its own: bench.py's
    launcher (and torch.distributed.run) then ends the other ranks instead of leaving them in a barrier.  Fresh child
    processes only -- nothing is re-executed."""
    print("bench.py: RANK %d of %d FAILED at: %s -- %s: %s" % (rank, world, stage, type(exc).__name__, exc), file=sys.stderr, flush=True)
    sys.exit(3)


def pack_pci(bus_id):
    """'dddd:bb:dd.f' -> one integer (exact in a double), -1 if it cannot be read."""
    try:
        dom, bus, rest = str(bus_id).split(":")
        dev, fn = rest.split(".")
        return (int(dom, 16) << 16) | (int(bus, 16) << 8) | (int(dev, 16) << 3) | int(fn, 16)
    except Exception:
        return -1


def unpack_pci(x):
    x = int(x)
    if x < 0:
        return None
    return "%04x:%02x:%02x.%x" % (x >> 16, (x >> 8) & 0xff, (x >> 3) & 0x1f, x & 7)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` started without a launcher: start the N ranks as children (fresh
    processes, one LOCAL_RANK each), wait for them, pass rank 0's JSON line through.  Nothing here
    touches the GPU; a rank never exec's after it has."""
    have = visible_gpu_count()
    if os.environ.get("QUFLOW_BENCH_REHEARSAL", "") == "share-gpu":
        have = max(have, args.gpus)         # the ranks share what is there (see main)
    if 0 <= have < args.gpus:
        print("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, have), file=sys.stderr)
        return 2
    if have < 0:
        # neither /sys/class/kfd nor /dev/dri can be read here: let the ranks find out (each one refuses
        # to run without a device of its own, so a short-handed launch still fails instead of under-reporting)
        print("bench.py: cannot count GPUs without touching them; starting %d ranks" % args.gpus, file=sys.stderr)
    port = _free_port()
    comm_port = _free_port()            # the torch-free gather's id hand-out (quflow_amd/comm.py) gets a port of its own
    nonce = "%016x" % int.from_bytes(os.urandom(8), "little")
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   QUFLOW_COMM_PORT=str(comm_port), QUFLOW_COMM_NONCE=nonce, QUFLOW_BENCH_CHILD="1")
        if len(cpus) >= 2 * args.gpus and os.environ.get("QUFLOW_BENCH_PIN", "1") != "0":
            # a rank's launch / poll thread busy-spins: give every rank its own slice of the cpus this process may
            # use (the first one of the slice is where the stepper's thread ends up; BLAS / OpenMP get the rest)
            per = len(cpus) // args.gpus
            env["QUFLOW_BENCH_CPUS"] = ",".join(str(c) for c in cpus[r * per:(r + 1) * per])
        for var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
            env.pop(var, None)          # each rank sizes its pools for its share of the quota
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # Supervise ALL children: the first non-zero exit (no device, import error, ...) or the deadline ends the
    # launch -- the remaining ranks would otherwise sit in the rendezvous / a barrier holding their GPUs.
    # (fresh child processes only: nothing is re-exec'ed.)
    import threading
    out_chunks = []
    reader = threading.Thread(target=lambda: out_chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get("QUFLOW_BENCH_LAUNCH_TIMEOUT", "3000"))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [(r, rc) for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = "rank(s) failed: %s" % bad
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            failed = "ranks still running at the deadline"
            break
        time.sleep(0.05)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_kill = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10.0)
    out0 = b"".join(c for c in out_chunks if c)
    if out0 and not failed:
        sys.stdout.write(out0.decode("utf-8", "replace"))
        sys.stdout.flush()
    if failed:
        print("bench.py: %s" % failed, file=sys.stderr)
        return 1
    return 0


# ------------------------------------------------------------------------------------------------
# side measurements
# ------------------------------------------------------------------------------------------------

def cpu_baseline(args, dt, N=None, seconds=None, min_steps=2, ic=None):
    """Oracle (CPU restatement of the reference path) on a bounded sample of the workload (`N`, `seconds`: the same
    workload at another target size with its own, smaller budget -- the entries of `other_sizes`; `min_steps`: never
    fewer timed steps than that, whatever the budget says; `ic`: another initial condition)."""
    import copy
    import numpy as np
    if N is not None:
        args = copy.copy(args)
        args.N = N
        args.cpu_seconds = seconds
    if ic is not None:
        args = copy.copy(args)
        args.ic = ic
    from oracle import isomp_oracle as oracle
    oracle.build()
    # the GPU box's CPU share for one GPU is 16 cores; never oversubscribe past the affinity mask
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(args.cpu_cores, avail, HOST_THREADS))
    oracle.set_threads(cores)
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=cores, user_api="blas")
    except Exception:
        pass
    W = oracle.make_W0(args.N, 0) if args.ic == "A" else oracle.make_W0_smooth(args.N, 0)
    if getattr(args, "dtype", "c128") == "c64":
        W = W.astype("complex64")          # the oracle's float32 restatement (numpy cgemm + float32 Thomas)
    kw = {}
    if args.fixed_iters:
        kw = dict(minit=args.fixed_iters, maxit=args.fixed_iters)
    if args.compsum:
        kw["compsum"] = True
    t0 = time.perf_counter()
    oracle.isomp(W, dt, steps=1, **kw)            # warm-up (BLAS threads, caches, table build)
    t1 = time.perf_counter() - t0
    steps = int(max(min_steps, min(200, args.cpu_seconds / max(t1, 1e-3))))
    stats = {"iterations": 0.0}
    t0 = time.perf_counter()
    oracle.isomp(W, dt, steps=steps, stats=stats, **kw)
    el = time.perf_counter() - t0
    return {"value": steps / el, "unit": "timesteps/s", "cores": int(cores), "kind": "port",
            "sample": "%d steps of the same N=%d workload (IC-%s, dt=%.2f*hbar, %.2f its/step), %.1f s, "
                      "numpy/OpenBLAS zgemm + OpenMP Thomas (oracle/)" %
                      (steps, args.N, args.ic, args.stepsize, stats["iterations"], el)}


def _read_kernel_times(lib, h, _lib, names, executed):
    """mean duration (s) per EXECUTED launch of the named kernels since the last profile reset
    (sampled launches scaled to all launches seen; tagged no-op launches stay in the numerator)."""
    n = ctypes.c_longlong()
    ms = ctypes.c_double()
    seen = ctypes.c_longlong()
    out = {}
    for name in names:
        _lib.check(lib.qf_profile_read(h, _lib.KERNEL_IDS[name], ctypes.byref(n), ctypes.byref(ms)))
        _lib.check(lib.qf_profile_seen(h, _lib.KERNEL_IDS[name], ctypes.byref(seen)))
        tot = ms.value * (seen.value / n.value if n.value else 0.0)
        out[name] = {"avg_s": 1e-3 * tot / max(executed, 1), "timed": int(n.value), "total_ms": tot}
    return out


def instrumented_pass(qfa, _lib, W0, dt, steps, kw, device, warmup=5):
    """An extra, UN-timed pass of the same workload on its own trajectory with HIP events around every
    launch of the three kernels of an iteration: the second product and the Laplacian inverse for the
    whole-step roofline (their events are kept out of the timed region, where each pair costs time)."""
    tr = qfa.DeviceTrajectory(W0, device=device)
    lib, h = tr.ctx._lib, tr.ctx.handle
    tr.advance(dt, warmup, **kw)
    _lib.check(lib.qf_profile_reset(h))
    _lib.check(lib.qf_profile_stride(h, 1))
    _lib.check(lib.qf_profile_enable(h, sum(1 << _lib.KERNEL_IDS[k] for k in ("poisson", "gemm1", "gemm2"))))
    st = tr.advance(dt, steps, **kw)
    tr.sync()
    _lib.check(lib.qf_profile_enable(h, 0))
    times = _read_kernel_times(lib, h, _lib, ("poisson", "gemm1", "gemm2"), int(st["total_iterations"]))
    plan = plan_of(tr)
    tr.ctx.close()
    return times, st, plan


def fixed_iteration_run(qfa, _lib, W0, N, device, iters=10, steps=40, warmup=5):
    """The reference profiler's protocol (profiling/run_profiling.py:124-127):
    dt = 0.01*hbar, minit = maxit = 10 -- deterministic work, no data-dependent exit."""
    dt = 0.01 * qfa.hbar(N)
    tr = qfa.DeviceTrajectory(W0, device=device)
    kw = dict(minit=iters, maxit=iters)
    tr.advance(dt, warmup, **kw)
    tr.sync()
    t0 = time.perf_counter()
    st = tr.advance(dt, steps, **kw)
    tr.sync()
    el = time.perf_counter() - t0
    tr.ctx.close()
    return {"protocol": "dt=0.01*hbar, minit=maxit=%d (quflow profiling/run_profiling.py:124-127)" % iters,
            "value": steps / el, "unit": "timesteps/s", "steps": steps, "ms_per_step": 1e3 * el / steps,
            "iterations_per_step": st["iterations"], "us_per_iteration": 1e6 * el / max(int(st["total_iterations"]), 1)}


def plan_of(tr):
    """What the trajectory's context launched for each role of the hot path (qf_plan_describe: recorded by the
    launchers themselves -- this file holds no copy of the library's kernel-selection rules)."""
    try:
        return tr.ctx.plan()
    except Exception as e:      # (an injected test trajectory has no device context)
        return {"error": str(e)}


def tile_share(plan):
    """Share of the full tile grid the second product multiplies, as its launcher recorded it (1.0: a full product)."""
    sp = (plan or {}).get("second_product") or {}
    return float(sp.get("tile_share", 1.0))


def kernel_label(entry):
    if not entry:
        return None
    return "%s (%s; %s x %s tiles, %s workgroups)" % (entry.get("kernel"), entry.get("arithmetic", ""),
                                                      (entry.get("tile") or ["?", "?"])[0], (entry.get("tile") or ["?", "?"])[1],
                                                      entry.get("workgroups"))


def step_bound_c64(N, iterations_per_step, share2):
    """The same bound for a complex64 state: executed flops at the fp32 MFMA peak, half the bytes per entry."""
    flops_it = 6.0 * N ** 3 * (1.0 + share2)
    bytes_it = (20.0 + 120.0) * N * N
    t_it = flops_it / (PEAK_FP32_MFMA_TFLOPS * 1e12) + bytes_it / (PEAK_HBM_GBS * 1e9)
    t_step = iterations_per_step * t_it + 3 * 8.0 * N * N / (PEAK_HBM_GBS * 1e9)
    return {"executed_flops_per_iteration": flops_it, "bytes_per_iteration": bytes_it, "bound_ms_per_step": 1e3 * t_step,
            "second_product_tile_share": share2}


def step_bound(N, iterations_per_step, share2):
    """Lower bound of one time step (seconds) from the work the kernels EXECUTE and the bytes of the
    minimal fused schedule: per iteration the 3M products (6 N^3 for the first, the
    upper-triangle tile share of 6 N^3 for the second) at the fp64 MFMA peak + (40 + 240) N^2 bytes at
    the HBM peak; per step the W update, 3 x 16 N^2 bytes."""
    flops_it = 6.0 * N ** 3 * (1.0 + share2)
    bytes_it = (40.0 + 240.0) * N * N
    t_it = flops_it / (PEAK_FP64_MFMA_TFLOPS * 1e12) + bytes_it / (PEAK_HBM_GBS * 1e9)
    t_step = iterations_per_step * t_it + 3 * 16.0 * N * N / (PEAK_HBM_GBS * 1e9)
    return {"executed_flops_per_iteration": flops_it, "bytes_per_iteration": bytes_it, "bound_ms_per_step": 1e3 * t_step,
            "second_product_tile_share": share2}


def static_traffic(key):
    """Fabric-side bytes per launch from profiles/pmc_traffic.json (separate rocprofv3 --pmc passes; not collected by this run)."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(key)
    except Exception:
        return None


def other_size_run(args, qfa, N, steps, warmup, device):
    """The same workload at another target size of BASELINE.json (N = 512, 2048; fp64 products),
    measured in the same process: rate and the first product's fraction of the fp64 MFMA roofline
    (HIP events around its launches, as for the headline)."""
    from quflow_amd import _lib
    W0 = qfa.ensemble.make_W0(N, 0)
    dt = args.stepsize * qfa.hbar(N)
    tr = qfa.DeviceTrajectory(W0, device=device)
    lib, h = tr.ctx._lib, tr.ctx.handle
    t_end = time.perf_counter() + 1e-3 * args.prewarm_ms      # (the GPU idled while numpy worked on the side run before)
    while time.perf_counter() < t_end:
        tr.advance(dt, 10)
    tr.advance(dt, warmup)
    tr.sync()
    # the rate without events (an event pair around a launch costs ~5 us of stream time: 4-8 % of a step at
    # N = 512), then the two products' durations in a second pass with them
    t0 = time.perf_counter()
    st = tr.advance(dt, steps)
    tr.sync()
    el = time.perf_counter() - t0
    _lib.check(lib.qf_profile_reset(h))
    _lib.check(lib.qf_profile_stride(h, EVENT_STRIDE))
    _lib.check(lib.qf_profile_enable(h, sum(1 << _lib.KERNEL_IDS[k] for k in ("gemm1", "gemm2", "poisson"))))
    st_ev = tr.advance(dt, max(steps // 2, 10))
    tr.sync()
    _lib.check(lib.qf_profile_enable(h, 0))
    executed = max(int(st_ev["total_iterations"]), 1)
    times = _read_kernel_times(lib, h, _lib, ("gemm1", "gemm2", "poisson"), executed)
    flops = 8.0 * N ** 3
    e1, s1 = tr.diagnostics()
    plan = plan_of(tr)
    tr.ctx.close()
    a1, a2, a0 = times["gemm1"]["avg_s"], times["gemm2"]["avg_s"], times["poisson"]["avg_s"]
    share2 = tile_share(plan)
    b = step_bound(N, st["iterations"], share2)
    return {"value": steps / el, "unit": "timesteps/s", "steps": steps, "ms_per_step": 1e3 * el / steps,
            "iterations_per_step": st["iterations"], "first_product_us": 1e6 * a1,
            "second_product_us": 1e6 * a2,
            # executed flops (6 N^3: 3M) / launch time / fp64 MFMA peak; the 8 N^3 figure is the algorithmic one
            "roofline_frac_first_product": 0.75 * flops / a1 / 1e12 / PEAK_FP64_MFMA_TFLOPS,
            "roofline_frac_second_product": 0.75 * flops * share2 / a2 / 1e12 / PEAK_FP64_MFMA_TFLOPS,
            "algorithmic_frac_first_product": flops / a1 / 1e12 / PEAK_FP64_MFMA_TFLOPS,
            "second_product_tile_share": share2,
            "kernels": {k: (plan.get(k) or {}).get("kernel") for k in ("laplacian_inverse", "first_product", "second_product")},
            # the HBM-bound kernel of an iteration (BASELINE config 5: "HBM-bandwidth roofline report"): 40 N^2 algorithmic
            # bytes per launch over its mean launch duration (events, one launch in EVENT_STRIDE)
            "laplacian_inverse": {"kernel": (plan.get("laplacian_inverse") or {}).get("kernel"), "bound": "hbm",
                                  "avg_launch_us": 1e6 * a0, "algorithmic_bytes_per_launch": 40.0 * N * N,
                                  "achieved_GBs": 40.0 * N * N / a0 / 1e9, "peak_GBs": PEAK_HBM_GBS,
                                  "frac": 40.0 * N * N / a0 / 1e9 / PEAK_HBM_GBS,
                                  "traffic": static_traffic("k_solve_bytes_per_launch_N%d" % N),
                                  "traffic_source": "profiles/pmc_traffic.json (static, rocprofv3 --pmc)"},
            "kernel_us_per_iteration": {"k_solve": 1e6 * a0, "first_product": 1e6 * a1, "second_product": 1e6 * a2,
                                        "sum": 1e6 * (a0 + a1 + a2),
                                        "wall_per_iteration_in_the_timed_region": 1e6 * el / max(int(st["total_iterations"]), 1)},
            "whole_step_bound_ms": b["bound_ms_per_step"],
            "whole_step_frac": b["bound_ms_per_step"] / (1e3 * el / steps),
            "enstrophy": s1}


def config3_other_size_run(args, qfa, N, steps, warmup, device, fp64_value, products="i8x65"):
    """Config 3's products at another target size, beside that size's fp64 line (same protocol as other_size_run's rate)."""
    old = os.environ.get("QUFLOW_HIP_GEMM")
    os.environ["QUFLOW_HIP_GEMM"] = products
    try:
        W0 = qfa.ensemble.make_W0(N, 0)
        dt = args.stepsize * qfa.hbar(N)
        tr = qfa.DeviceTrajectory(W0, device=device)
        t_end = time.perf_counter() + 1e-3 * args.prewarm_ms
        while time.perf_counter() < t_end:
            tr.advance(dt, 10)
        tr.advance(dt, warmup)
        tr.sync()
        t0 = time.perf_counter()
        st = tr.advance(dt, steps)
        tr.sync()
        el = time.perf_counter() - t0
        W = tr.download()
        tr.ctx.close()
    finally:
        if old is None:
            os.environ.pop("QUFLOW_HIP_GEMM", None)
        else:
            os.environ["QUFLOW_HIP_GEMM"] = old
    import numpy as np
    return {"products": products, "N": N, "value": steps / el, "unit": "timesteps/s", "steps": steps,
            "iterations_per_step": st["iterations"], "vs_fp64_same_size": (steps / el) / fp64_value,
            "skew_hermitian_exact": bool(np.array_equal(W, -W.conj().T))}


def complex64_side_run(args, qfa, N, steps, warmup, device):
    """The same workload on a complex64 state (the reference computes it in single precision throughout; here: float32
    Poisson solve, products on the fp32 matrix cores -- DESIGN.md 3.7), same process: rate and the first product's
    fraction of the fp32 MFMA roofline.  `python bench.py --dtype c64` gives the full line with its CPU baseline."""
    from quflow_amd import _lib
    import numpy as np
    W0 = qfa.ensemble.make_W0(N, 0).astype(np.complex64)
    dt = args.stepsize * qfa.hbar(N)
    tr = qfa.DeviceTrajectory(W0, device=device)
    lib, h = tr.ctx._lib, tr.ctx.handle
    t_end = time.perf_counter() + 1e-3 * args.prewarm_ms
    while time.perf_counter() < t_end:
        tr.advance(dt, 10)
    tr.advance(dt, warmup)
    tr.sync()
    t0 = time.perf_counter()
    st = tr.advance(dt, steps)
    tr.sync()
    el = time.perf_counter() - t0
    _lib.check(lib.qf_profile_reset(h))
    _lib.check(lib.qf_profile_stride(h, EVENT_STRIDE))
    _lib.check(lib.qf_profile_enable(h, (1 << _lib.KERNEL_IDS["gemm1"]) | (1 << _lib.KERNEL_IDS["gemm2"])))
    st_ev = tr.advance(dt, max(steps // 2, 10))
    tr.sync()
    _lib.check(lib.qf_profile_enable(h, 0))
    times = _read_kernel_times(lib, h, _lib, ("gemm1", "gemm2"), max(int(st_ev["total_iterations"]), 1))
    e1, s1 = tr.diagnostics()
    tr.ctx.close()
    a1, a2 = times["gemm1"]["avg_s"], times["gemm2"]["avg_s"]
    return {"value": steps / el, "unit": "timesteps/s", "dtype": "f32", "steps": steps, "ms_per_step": 1e3 * el / steps,
            "iterations_per_step": st["iterations"], "first_product_us": 1e6 * a1, "second_product_us": 1e6 * a2,
            "roofline_frac_first_product": 6.0 * N ** 3 / a1 / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "mfma_peak_TFLOPs": PEAK_FP32_MFMA_TFLOPS, "enstrophy": s1}


def replicas_per_gpu_run(args, qfa, N, k, steps, device, warmup=20):
    """k independent replicas advanced together on ONE GPU (DeviceEnsemble / qf_isomp_multi):
    sum of their timesteps/s against one trajectory alone, same size, same process."""
    dt = args.stepsize * qfa.hbar(N)
    def rate(kk):
        ens = qfa.DeviceEnsemble([qfa.ensemble.make_W0(N, s) for s in range(kk)], device=device)
        # (the side runs before this one left the GPU idle while numpy worked: bring its clock back up, as
        # the headline does, and give the BLAS workers numpy woke up time to go back to sleep -- k replicas
        # at N=512 need ~180,000 launches per second from ONE host thread)
        t_end = time.perf_counter() + 1e-3 * max(args.prewarm_ms, 150.0)
        while time.perf_counter() < t_end:
            ens.advance(dt, 10)
        ens.advance(dt, warmup)
        ens.sync()
        t0 = time.perf_counter()
        st = ens.advance(dt, steps)
        ens.sync()
        el = time.perf_counter() - t0
        ens.close()
        return kk * steps / el, sum(x["iterations"] for x in st) / kk
    single, its1 = rate(1)
    together, itsk = rate(k)
    return {"N": N, "replicas": k, "steps": steps, "sum_timesteps_per_s": together, "single_trajectory_timesteps_per_s": single,
            "ratio": together / single, "iterations_per_step": itsk,
            "how": "each replica bit-identical to its own single-trajectory run (tests/test_hip_parity.py::"
                   "test_device_ensemble_members_are_bit_identical_to_single_runs)"}


def config3_side_run(args, qfa, tr_f64, W0, dt, kw, device, products="i8x6"):
    """BASELINE.json config 3 beside the headline: the same trajectory (same W0, same number of
    steps) with both commutator products on the int8 matrix cores by digit splitting (ozaki.hip),
    Laplacian inverse in fp64.  Reports its rate and how far its state, spectrum and Casimirs are
    from the fp64 run's on the same steps (the stepper's own fixed-point tolerance is ~sqrt(eps))."""
    import numpy as np
    W_f64 = tr_f64.download()
    old = os.environ.get("QUFLOW_HIP_GEMM"), os.environ.get("QUFLOW_HIP_I8_MIN_N")
    os.environ["QUFLOW_HIP_GEMM"] = products
    os.environ["QUFLOW_HIP_I8_MIN_N"] = "64"
    try:
        # the headline's protocol: W warm-up steps and K timed steps from a cold clock (`value_without_prewarm`), then
        # the clock warm-up the headline gets (prewarm_ms of load on a scratch trajectory) and the SAME K steps on a
        # fresh trajectory -- so that state and drifts are those of warmup + K steps, as for the fp64 run
        value_without_prewarm = None
        scratch = qfa.DeviceTrajectory(W0, device=device)
        if args.prewarm_ms > 0:
            if args.warmup > 0:
                scratch.advance(dt, args.warmup, **kw)
            scratch.sync()
            tc = time.perf_counter()
            scratch.advance(dt, args.steps, **kw)
            scratch.sync()
            value_without_prewarm = args.steps / (time.perf_counter() - tc)
        tr = qfa.DeviceTrajectory(W0, device=device)
        t_end = time.perf_counter() + 1e-3 * args.prewarm_ms
        while time.perf_counter() < t_end:
            scratch.advance(dt, 10, **kw)
        scratch.sync()
        if args.warmup > 0:
            tr.advance(dt, args.warmup, **kw)
        tr.sync()
        t0 = time.perf_counter()
        st = tr.advance(dt, args.steps, **kw)
        tr.sync()
        el = time.perf_counter() - t0
        W_i8 = tr.download()
        tr.ctx.close()
        scratch.ctx.close()
    finally:
        for k, v in zip(("QUFLOW_HIP_GEMM", "QUFLOW_HIP_I8_MIN_N"), old):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def casimirs(W):
        A = 1j * W
        A2 = A @ A
        return np.array([np.trace(A2).real, np.trace(A2 @ A).real, np.trace(A2 @ A2).real]) / W.shape[0]
    c0 = casimirs(W0)
    res = {"products": products, "value": args.steps / el, "unit": "timesteps/s", "ms_per_step": 1e3 * el / args.steps,
           "value_without_prewarm": value_without_prewarm,
           "iterations_per_step": st["iterations"],
           "max_abs_state_diff_vs_f64_run": float(np.abs(W_i8 - W_f64).max()),
           "casimir_drift": float(np.abs(casimirs(W_i8) - c0).max()),
           "casimir_drift_f64_run": float(np.abs(casimirs(W_f64) - c0).max()),
           "skew_hermitian_exact": bool(np.array_equal(W_i8, -W_i8.conj().T)),
           "how": "QUFLOW_HIP_GEMM=%s: same W0, warmup and steps as the headline run; "
                  "python bench.py --products %s gives its own roofline line" % (products, products)}
    if W0.shape[0] <= 1024:
        ev0 = np.linalg.eigvalsh(1j * W0)
        res["spectrum_drift"] = float(np.abs(np.linalg.eigvalsh(1j * W_i8) - ev0).max())
        res["spectrum_drift_f64_run"] = float(np.abs(np.linalg.eigvalsh(1j * W_f64) - ev0).max())
    return res


def smooth_data_side_run(args, qfa, N, steps, warmup, device):
    """IC-B: W0 = normalised solve_poisson(make_W0) -- smooth data, the regime the reference's notebook
    works in (more fixed-point iterations per step than white noise).  Rate and iterations per step at the headline size;
    `oracle_check` (filled in beside the CPU baseline) is the CPU oracle's iteration count on the first steps of the same
    trajectory."""
    import numpy as np
    W0 = qfa.solve_poisson(qfa.ensemble.make_W0(N, 0)).copy()
    W0 /= np.linalg.norm(W0, "fro") / np.sqrt(N)
    dt = args.stepsize * qfa.hbar(N)
    check_steps = 3
    trc = qfa.DeviceTrajectory(W0, device=device)
    stc = trc.advance(dt, check_steps)
    trc.ctx.close()
    tr = qfa.DeviceTrajectory(W0, device=device)
    t_end = time.perf_counter() + 1e-3 * args.prewarm_ms
    while time.perf_counter() < t_end:
        tr.advance(dt, 10)
    tr.advance(dt, warmup)
    tr.sync()
    t0 = time.perf_counter()
    st = tr.advance(dt, steps, diagnostics=True)
    tr.sync()
    el = time.perf_counter() - t0
    plan = plan_of(tr)
    tr.ctx.close()
    share2 = tile_share(plan)
    b = step_bound(N, st["iterations"], share2)
    return {"workload": "IC-B: W0 = solve_poisson(make_W0(N, 0)) normalised to enstrophy 1/2 (smooth data), N=%d, dt=%.2f*hbar, "
                        "adaptive defaults" % (N, args.stepsize),
            "value": steps / el, "unit": "timesteps/s", "steps": steps, "ms_per_step": 1e3 * el / steps,
            "iterations_per_step": st["iterations"], "number_of_maxit": st["number_of_maxit"],
            "us_per_iteration": 1e6 * el / max(int(st["total_iterations"]), 1),
            "whole_step_bound_ms": b["bound_ms_per_step"], "whole_step_frac": b["bound_ms_per_step"] / (1e3 * el / steps),
            "energy": st["energy"], "enstrophy": st["enstrophy"],
            "oracle_check": {"steps": check_steps, "device_total_iterations": int(stc["total_iterations"])}}


def smooth_data_oracle_check(args, entry, N):
    """The CPU oracle on the first steps of the IC-B trajectory: its iteration count beside the device's (equal)."""
    from oracle import isomp_oracle as oracle
    oracle.build()
    W = oracle.make_W0_smooth(N, 0)
    n = entry["oracle_check"]["steps"]
    stats = {"iterations": 0.0}
    t0 = time.perf_counter()
    oracle.isomp(W, args.stepsize * oracle.hbar(N), steps=n, stats=stats)
    el = time.perf_counter() - t0
    total = int(round(stats["iterations"] * n))
    entry["oracle_check"].update({"oracle_total_iterations": total, "equal": total == entry["oracle_check"]["device_total_iterations"],
                                  "oracle_timesteps_per_s": n / el})


def per_call_run(qfa, sizes=(512, 1024, 2048), budget_s=0.6):
    """Seconds per call of the PUBLIC host-array API, the way the reference's harness times it
    (profiling/run_profiling.py:48-94: matmul / commutator / solve_poisson / laplace, one warm-up call, then the mean of
    `repeats` calls): qfa.solve_poisson(W), qfa.laplace(P), qfa.commutator(W, P) with host ndarrays in and out -- PCIe
    both ways included; what a notebook user who holds no DeviceTrajectory sees.  The oracle's seconds per call
    (`oracle_*`: OpenMP Thomas / stencil, numpy zgemm) on this host's cores beside them; bounded by `budget_s` per entry."""
    import numpy as np
    from oracle import isomp_oracle as oracle
    oracle.build()

    def comm_cpu(W, P):                       # commutator_skewherm, isospectral.py:40-57: W@P - (W@P)^H
        X = W @ P
        return X - X.conj().T

    def timeit(fn, *a):
        fn(*a)                                # warm-up (contexts, tables, BLAS threads)
        t0 = time.perf_counter()
        fn(*a)
        t1 = max(time.perf_counter() - t0, 1e-6)
        reps = int(max(2, min(200, budget_s / t1)))
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(*a)
        return (time.perf_counter() - t0) / reps, reps

    out = {"protocol": "profiling/run_profiling.py:48-94: one warm-up call, mean of `repeats` calls; host ndarray in, host ndarray "
                       "out (PCIe both ways; laplace / commutator hand back a FRESH array per call as the reference's do, solve_poisson "
                       "its persistent one, cpu.py:726); oracle_* = the CPU oracle's call on this host (%d threads), timed in a second "
                       "pass AFTER every device timing (its idle BLAS / OpenMP workers spin and would eat the host thread's share "
                       "of a CPU quota)" % HOST_THREADS}
    data = {}
    time.sleep(0.3)                           # (the CPU baseline ran just before: let its workers go back to sleep)
    for N in sizes:                           # pass 1: the device path, before any oracle call of this function
        W = qfa.ensemble.make_W0(N, 0)
        P = qfa.solve_poisson(W).copy()
        data[N] = (W, P)
        row = {}
        for name, fn, a in (("solve_poisson", qfa.solve_poisson, (W,)), ("laplace", qfa.laplace, (P,)),
                            ("commutator", qfa.commutator, (W, P))):
            t, reps = timeit(fn, *a)
            row[name] = {"seconds_per_call": t, "repeats": reps, "host_bytes_moved": (16.0 * N * N) * (len(a) + 1)}
        out["N%d" % N] = row
    for N in sizes:                           # pass 2: the oracle's calls
        W, P = data[N]
        for name, cpu, a in (("solve_poisson", oracle.solve_poisson, (W,)), ("laplace", oracle.laplace, (P,)),
                             ("commutator", comm_cpu, (W, P))):
            tc, repc = timeit(cpu, *a)
            out["N%d" % N][name].update({"oracle_seconds_per_call": tc, "oracle_repeats": repc})
    from quflow_amd.context import release_contexts
    release_contexts()
    return out


def _load_injected_trajectory():
    """Tests of the launch / gather plumbing on a machine without a GPU inject a trajectory class
    (QUFLOW_BENCH_TRAJECTORY=/path/file.py:Class; interface of DeviceTrajectory: advance, diagnostics,
    sync).  Never set in a measurement: the line then says data = "injected trajectory (test)"."""
    spec = os.environ.get("QUFLOW_BENCH_TRAJECTORY")
    if not spec:
        return None
    path, _, cls = spec.rpartition(":")
    m = importlib.util.spec_from_file_location("qf_bench_injected", path)
    mod = importlib.util.module_from_spec(m)
    m.loader.exec_module(mod)
    return getattr(mod, cls)


def main():
    # (read by the HIP runtime when it initialises -- under torch.distributed.run that is torch's doing,
    # before quflow_amd is imported: see quflow_amd/__init__.py and DESIGN.md 4d)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pinned = pin_rank_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    if args.gpus != world:
        # the launcher decides how many ranks exist; a mismatch is a mis-launch, not a 1-GPU measurement
        print("bench.py: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    injected = _load_injected_trajectory()
    backend = os.environ.get("QUFLOW_BENCH_BACKEND", "nccl")      # "nccl" = RCCL on ROCm; "gloo" in the CPU tests
    # REHEARSAL of the N-rank flow on a box with fewer GPUs than ranks (the build's box has one): the ranks share the
    # visible GPU(s) round-robin and gather over gloo (RCCL refuses two ranks on one device).  Launcher, per-rank contexts,
    # pinning, timed region, gather and max-over-ranks are the real ones; the line says `rehearsal` and is NOT an N-GPU
    # measurement (tools/gpu/r6_rehearse_ranks.sh).
    rehearsal = os.environ.get("QUFLOW_BENCH_REHEARSAL", "") == "share-gpu"
    if rehearsal:
        backend = "gloo"
    device_index = local_rank
    dist = None
    torch = None
    native_gather = os.environ.get("QUFLOW_BENCH_GATHER", "torch") == "native"
    if native_gather and (world > 1 or "TORCHELASTIC_RUN_ID" in os.environ):
        # torch-free: RCCL behind the C ABI (quflow_amd.comm.NativeComm); created after the package import below
        pass
    elif world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        # launched by torch.distributed.run or by self_launch (also with one rank: exercises the RCCL path).
        # torch first: its bundled HIP runtime must be the one libquflow_hip.so binds to
        import datetime
        import torch
        import torch.distributed as dist
        rdzv = datetime.timedelta(seconds=float(os.environ.get("QUFLOW_BENCH_RDZV_TIMEOUT", "300")))
        try:
            if backend == "nccl":
                if torch.cuda.device_count() <= local_rank:
                    print("bench.py: rank %d has no GPU (%d visible)" % (rank, torch.cuda.device_count()), file=sys.stderr)
                    sys.exit(2)
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=rdzv)
            else:
                dist.init_process_group(backend, timeout=rdzv)
        except Exception as e:      # a rendezvous that does not complete names its rank and ends the launch
            rank_fail(rank, world, "rendezvous (init_process_group, backend %s, MASTER %s:%s)" % (
                backend, os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")), e)
    gather_device = (torch.device("cuda", local_rank) if (dist is not None and backend == "nccl") else None)

    if args.products != "f64":
        os.environ["QUFLOW_HIP_GEMM"] = args.products  # read when the device context is created
    import numpy as np
    import quflow_amd as qfa
    from quflow_amd import _lib
    if native_gather and injected is None and (world > 1 or "TORCHELASTIC_RUN_ID" in os.environ):
        from quflow_amd.comm import NativeComm
        try:
            dist = NativeComm(rank=rank, world=world, device=device_index)
        except Exception as e:
            rank_fail(rank, world, "communicator (RCCL behind the C ABI, id hand-out on port %s)" % os.environ.get("QUFLOW_COMM_PORT"), e)
    dev_info = {"ordinal": None, "pci_bus_id": None, "name": "injected trajectory (no device)"}
    if injected is None:
        if rehearsal:
            device_index = local_rank % max(1, qfa.device_count())
        qfa.set_device(device_index)
        if qfa.device_count() <= device_index:
            raise SystemExit("bench.py: no HIP device for rank %d; the benchmark has no CPU path" % rank)
        dev_info = qfa.device_info(device_index)
    # every rank says where it landed (stderr): the first thing to read when an N-GPU launch misbehaves
    print("bench.py: rank %d/%d pid %d LOCAL_RANK %d -> HIP device %s, PCI %s (%s, %s CUs); cpus %s; gather %s"
          % (rank, world, os.getpid(), local_rank, dev_info.get("ordinal"), dev_info.get("pci_bus_id"), dev_info.get("name"),
             dev_info.get("compute_units"), ("%d pinned" % len(pinned)) if pinned else "not pinned",
             "none" if dist is None else ("native RCCL" if native_gather else backend)), file=sys.stderr, flush=True)

    N = args.N
    dt = args.stepsize * qfa.hbar(N)
    seed = rank
    W0 = qfa.ensemble.make_W0(N, seed)
    if args.ic == "B":
        W0 = qfa.solve_poisson(W0).copy()
        W0 /= np.linalg.norm(W0, "fro") / np.sqrt(N)
    kw = {}
    if args.fixed_iters:
        kw = dict(minit=args.fixed_iters, maxit=args.fixed_iters)
    if args.compsum:
        kw["compsum"] = True
    c64 = args.dtype == "c64"
    if c64:
        if args.products != "f64" or args.stepper != "isomp":
            raise SystemExit("bench.py: --dtype c64 goes with the isomp stepper and its own products")
        W0 = W0.astype(np.complex64)
        args.no_config3 = True

    if injected is not None:
        tr = injected(W0)
        lib = h = None
    else:
        tr = qfa.DeviceTrajectory(W0, device=device_index)        # state resident in HBM
        lib, h = tr.ctx._lib, tr.ctx.handle

    def barrier():
        tr.sync()
        if dist is not None:
            if torch is not None and backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    # the collectives of the timed region once, long before it: their first use pays communicator set-up, channel
    # connects and buffer registration (tens of milliseconds the first time, ~1 ms the second under torch's NCCL backend)
    # -- here, ahead of the clock warm-up, not inside a 20-step timed region and not as an idle gap in front of it
    if dist is not None:
        for _ in range(2):
            qfa.ensemble.gather_diagnostics([[float(seed), 0.0, 0.0, 0.0]], dist=dist, device=gather_device, rows_per_rank=1)
            barrier()

    if args.stepper in ("isomp_simple", "isomp_quasinewton"):
        def advance(n):
            return tr.advance_lu(args.stepper, dt, n)
    elif args.stepper != "isomp":
        def advance(n):
            return tr.advance_erk(args.stepper, dt, n)
    else:
        def advance(n):
            return tr.advance(dt, n, **kw)

    # let the BLAS/OpenMP workers that make_W0 woke up go back to sleep (they spin for some
    # milliseconds after their last job and would eat into the cgroup's CPU quota) BEFORE the warm-up,
    # so that nothing idles the GPU between the W warm-up steps and the timed region: its clock is
    # up when the timed region starts (an idle gap of 0.2 s costs ~2 % of a 200-step run)
    time.sleep(0.2)
    scratch = None
    value_without_prewarm = None
    prewarm_rate = 0.0
    if args.prewarm_ms > 0 and args.stepper == "isomp" and injected is None:
        # clock warm-up on a scratch trajectory (not the measured state, not counted in W or K)
        scratch = qfa.DeviceTrajectory(W0, device=device_index)
        if world == 1 and not c64:
            # the same measurement WITHOUT the clock warm-up, taken first (a fresh process, an idle GPU): W warm-up
            # steps, then K timed steps with the chunk's diagnostics, on the scratch trajectory.  Reported beside
            # `value` as config.value_without_prewarm; it also is the first part of the warm-up of what follows.
            if args.warmup > 0:
                scratch.advance(dt, args.warmup, **kw)
            scratch.sync()
            tc = time.perf_counter()
            scratch.advance(dt, args.steps, diagnostics=True, **kw)
            scratch.sync()
            value_without_prewarm = args.steps / (time.perf_counter() - tc)
        t_end = time.perf_counter() + 1e-3 * args.prewarm_ms
        while time.perf_counter() < t_end:
            tp = time.perf_counter()
            scratch.advance(dt, 10, **kw)
            scratch.sync()
            prewarm_rate = 10.0 / (time.perf_counter() - tp)       # the last chunk's rate: is this GPU's clock up?
        scratch.sync()
        # the scratch context is released AFTER the timed region: freeing its buffers idles the GPU for
        # ~6 ms (rocprof timeline, tools/trace_timeline.py), and its clock then needs ~10 ms of load to come
        # back -- longer than a 20-step timed region
    if args.warmup > 0:
        advance(args.warmup)
    e0, s0 = (0.0, 0.0) if os.environ.get("BENCH_SKIP_DIAG0") else tr.diagnostics()

    events = (lib is not None) and not args.no_kernel_events
    if lib is not None:
        _lib.check(lib.qf_profile_reset(h))
    if events:
        # HIP events in the timed region around launches of the DOMINANT kernel only (the first
        # product): every launch for the kernel table, otherwise one launch in 4 (short runs) to 8 --
        # an event pair around a launch costs ~5 us of stream time, ~6 % of the rate when every product
        # launch carries one.  The other kernels are timed in a separate pass after the timed region.
        if args.kernel_table:
            mask, stride = (1 << len(_lib.KERNEL_IDS)) - 1, 1
        else:
            mask = 1 << _lib.KERNEL_IDS["gemm1"]
            stride = 4 if args.steps < 100 else EVENT_STRIDE
        _lib.check(lib.qf_profile_stride(h, stride))
        _lib.check(lib.qf_profile_enable(h, mask))

    barrier()
    t0 = time.perf_counter()
    if lib is not None:
        _lib.check(lib.qf_timer_start(h))
    if args.stepper == "isomp" and injected is None:
        # an output chunk: advance, then the diagnostics the ranks gather -- queued behind the last step,
        # one synchronisation (qf_isomp_diag)
        st = tr.advance(dt, args.steps, diagnostics=True, **kw)
        e1, s1 = st["energy"], st["enstrophy"]
    else:
        st = advance(args.steps)
        e1, s1 = tr.diagnostics()
    t_adv = time.perf_counter()
    table = qfa.ensemble.gather_diagnostics([[float(seed), e1, s1, st["iterations"]]], dist=dist, device=gather_device,
                                             rows_per_rank=1)       # one replica per rank: ONE collective per chunk
    t_gat = time.perf_counter()
    ev_ms = ctypes.c_double()
    if lib is not None:
        _lib.check(lib.qf_timer_stop(h, ctypes.byref(ev_ms)))
    # this rank's K steps are done and synchronised (advance returned the diagnostics, the gather its rows): its elapsed
    # time stops HERE; the closing barrier brackets the region, and the job's time is the MAX of the ranks' elapsed times
    # (all-gathered below) -- the same instant the barrier releases everyone, without the barrier's own latency on top
    tr.sync()
    elapsed_rank = time.perf_counter() - t0
    barrier()
    t_bar = time.perf_counter()
    region_ms = {"advance_and_diagnostics": 1e3 * (t_adv - t0), "gather": 1e3 * (t_gat - t_adv),
                 "closing_barrier_not_counted": 1e3 * (t_bar - t0) - 1e3 * elapsed_rank}
    if scratch is not None:
        scratch.ctx.close()
        scratch = None
    if events:
        _lib.check(lib.qf_profile_enable(h, 0))

    elapsed = elapsed_rank
    rank_rates = [args.steps / elapsed_rank]
    # one row per rank: elapsed time, the device it bound (ordinal, PCI domain:bus:device.function packed into one
    # exactly representable number), the rate of its last clock warm-up chunk
    my_row = [elapsed_rank, float(dev_info["ordinal"] if dev_info.get("ordinal") is not None else -1),
              float(pack_pci(dev_info.get("pci_bus_id"))), float(prewarm_rate), float(os.getpid())]
    rank_rows = [my_row]
    try:
        if dist is not None and hasattr(dist, "allgather_f64"):
            rank_rows = [[float(x) for x in r] for r in dist.allgather_f64(my_row)]
        elif dist is not None:
            t = torch.tensor(my_row, dtype=torch.float64, device=gather_device or "cpu")
            ts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(ts, t)
            rank_rows = [[float(v) for v in x.tolist()] for x in ts]
    except Exception as e:
        rank_fail(rank, world, "all-gather of the per-rank rows", e)
    if dist is not None:
        per_rank = [r[0] for r in rank_rows]
        elapsed = max(per_rank)
        rank_rates = [args.steps / x for x in per_rank]

    executed = int(st.get("total_iterations", 0))
    per = _read_kernel_times(lib, h, _lib, ("gemm1",), executed) if events else {}

    if args.kernel_table and rank == 0 and lib is not None:
        n = ctypes.c_longlong()
        ms = ctypes.c_double()
        tot = 0.0
        for name, kid in _lib.KERNEL_IDS.items():
            _lib.check(lib.qf_profile_read(h, kid, ctypes.byref(n), ctypes.byref(ms)))
            tot += ms.value
            print("kernel-table %-8s launches %6d  total %9.3f ms  avg %8.2f us" %
                  (name, n.value, ms.value, 1e3 * ms.value / max(1, n.value)), file=sys.stderr)
        print("kernel-table sum %.3f ms of %.3f ms elapsed" % (tot, 1e3 * elapsed), file=sys.stderr)

    if rank == 0:
        value = world * args.steps / elapsed
        out = {
            "metric": METRIC, "value": value, "unit": "timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if c64 else "f64" if args.products == "f64" else
                     "i8 digits (%d x 7 bit, int32 accumulate) for the %s, f64 elsewhere" % (
                         6 if "x6" in args.products else 5, "second product" if "h" in args.products else "products"),
            "data": "synthetic" if injected is None else "injected trajectory (test)",
            "config": {"workload": "%s on random skew-Hermitian "
                                   "trace-free W0, N=%d complex128, dt=%.2f*hbar, IC-%s, one independent "
                                   "trajectory per GPU" % (
                                       "isomp (adaptive fixed-point, tol=auto, maxit=10)" if args.stepper == "isomp"
                                       else args.stepper + " (explicit, quflow/integrators/erk.py)",
                                       N, args.stepsize, args.ic)
                                   + (" -- complex64 state, single-precision arithmetic as the reference's complex64 path"
                                      if c64 else ""),
                       "state_dtype": "complex64" if c64 else "complex128",
                       "stepper": args.stepper, "products": args.products,
                       "N": N, "stepsize": args.stepsize, "ic": args.ic,
                       "iterations_per_step": st["iterations"], "fixed_iters": args.fixed_iters,
                       "compsum": bool(args.compsum), "gpu_clock_prewarm_ms": args.prewarm_ms,
                       "value_without_prewarm": value_without_prewarm, "replicas": world, "parallelism": "replicas x%d" % world,
                       "device_ms_per_step_rank0": ev_ms.value / args.steps,
                       "energy_drift": e1 - e0, "enstrophy_drift": s1 - s0,
                       "gathered_rows": int(table.shape[0]),
                       "gather": {"backend": (dist.get_backend() if dist is not None else "none (one process)"),
                                  "rccl_world_size": (dist.get_world_size() if dist is not None else 1),
                                  "gathered_rows_ok": bool(int(table.shape[0]) == world),
                                  "seeds_gathered": sorted(int(x) for x in table[:, 0])},
                       "per_rank_timesteps_per_s": rank_rates,
                       "ranks": [{"rank": r, "hip_device": int(row[1]), "pci_bus_id": unpack_pci(row[2]),
                                  "timesteps_per_s": args.steps / row[0], "prewarm_last_chunk_timesteps_per_s": row[3],
                                  "pid": int(row[4])} for r, row in enumerate(rank_rows)],
                       "distinct_devices_bound": len({(int(row[1]), int(row[2])) for row in rank_rows}),
                       "rehearsal": ("%d ranks sharing %d GPU(s), gather over gloo -- the N-rank flow rehearsed on a smaller "
                                     "box, NOT an N-GPU measurement" % (world, len({(int(row[1]), int(row[2])) for row in rank_rows}))
                                     if rehearsal else None),
                       "timed_region_ms_rank0": region_ms,
                       "rank0_cpus_pinned": (len(pinned) if pinned else None)},
        }
        avg1 = per["gemm1"]["avg_s"] if per.get("gemm1", {}).get("timed") else None
        if avg1:
            flops = 8.0 * N ** 3                      # algorithmic: one complex N^3 GEMM
            # Tagged launches that were not due are no-ops whose (tiny) time stays in the numerator:
            # the averages are per EXECUTED launch.  isomp: one first product (k_zgemm, the dominant
            # kernel) and one second product per executed iteration; the second product is the
            # upper-triangle stream-K kernel k_zgemm_tri when W is skew-Hermitian and N >= 768.
            traffic = None
            traffic2 = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if c64:
                        cj = tj.get("complex64_N%d" % N) or (tj.get("round3") or {}).get("complex64_N%d" % N) or {}
                        traffic = cj.get("cgemm_ks_bytes_per_launch")
                        traffic2 = cj.get("cgemm_tri_bytes_per_launch")
                    elif args.products == "f64":
                        traffic = tj.get("zgemm_plain_bytes_per_launch_N%d" % N)
                        traffic2 = tj.get("zgemm_tri_bytes_per_launch_N%d" % N)
                    else:
                        traffic = tj.get("oz_gemm_%s_plain_bytes_per_launch_N%d" % (args.products, N))
                        traffic2 = tj.get("oz_gemm_%s_fused_bytes_per_launch_N%d" % (args.products, N))
                except Exception:
                    traffic = None
            ach = flops / avg1 / 1e12
            peak, unit = PEAK_FP64_MFMA_TFLOPS, "TFLOP/s"
            plan = plan_of(tr)                        # what the library launched in the timed region, in its own words
            kname = "first product Phalf@Whalf: " + str(kernel_label(plan.get("first_product")))
            exec_flops = 6.0 * N ** 3                 # what the 3M kernel issues: 3 real MFMA products
            if c64:
                peak = PEAK_FP32_MFMA_TFLOPS
            exec_flops2 = None
            if args.products in ("i8", "i8x6", "i8x6f", "i8x65") and args.stepper == "isomp":
                # the int8 kernel is priced in the int8 operations it issues: digit pairs x 3 real products x 2 ops per
                # MAC per N^3 -- 90 N^3 (15 pairs, five digits) or 126 N^3 (21 pairs, six digits); the pair count of EACH
                # product is the one its launcher recorded (i8x65: 21 for the first product, 15 for the second)
                def pairs_of(entry, fallback):
                    return float((entry or {}).get("digit_pairs") or fallback)
                fallback = 15 if args.products == "i8" else 21
                flops = 6.0 * pairs_of(plan.get("first_product"), fallback) * N ** 3
                exec_flops = flops
                if (plan.get("second_product") or {}).get("digit_pairs"):
                    exec_flops2 = 6.0 * pairs_of(plan.get("second_product"), fallback) * N ** 3
                ach, peak, unit = flops / avg1 / 1e12, PEAK_I8_MFMA_TOPS, "TOP/s"
            # `frac` (and `achieved`) price what the kernel EXECUTES: a 3M complex product issues 6 N^3 real
            # flops for the 8 N^3 of SURVEY.md 8d's algorithmic count, so the algorithmic figure over the
            # matrix peak can exceed 1 and is not a roofline fraction -- it is kept as `algorithmic_frac`.
            ach_exec = exec_flops / avg1 / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach_exec, "peak": peak, "unit": unit,
                               "frac": ach_exec / peak, "traffic": traffic,
                               "traffic_source": ("profiles/pmc_traffic.json (static, rocprofv3 --pmc; not collected by this run)"
                                                  if traffic is not None else None),
                               "kernel": kname,
                               "launched": {k: plan.get(k) for k in ("laplacian_inverse", "first_product", "second_product", "slicing")
                                            if isinstance(plan, dict)},
                               "launches": executed, "launches_timed_with_events": int(per["gemm1"]["timed"]),
                               "avg_launch_us": 1e6 * avg1, "flops_per_launch": flops,
                               "executed_flops_per_launch": exec_flops,
                               "executed_frac": ach_exec / peak,
                               "algorithmic_achieved": ach, "algorithmic_frac": ach / peak,
                               "note": "frac = flops the kernel executes (6 N^3: 3M complex product) / launch time / peak: the share "
                                       "of the matrix pipe it keeps busy; algorithmic_frac prices the 8 N^3 of a complex product "
                                       "(SURVEY.md 8d) and may exceed 1"}
            if (world == 1 and args.stepper == "isomp" and not args.no_side_runs and not args.kernel_table):
                # second product and Laplacian inverse: events around every launch, outside the timed region
                times, st2, plan2 = instrumented_pass(qfa, _lib, W0, dt, min(args.steps, 50), kw, device_index)
                a1, a2, a0 = times["gemm1"]["avg_s"], times["gemm2"]["avg_s"], times["poisson"]["avg_s"]
                share2 = tile_share(plan2)
                ef2 = exec_flops
                peak2 = peak
                if exec_flops2 is not None and str((plan2.get("second_product") or {}).get("kernel", "")).startswith("k_oz_gemm"):
                    ef2 = exec_flops2             # the second product's own digit-pair count
                elif exec_flops2 is None and args.products not in ("f64",) and not c64:
                    ef2, peak2 = 6.0 * N ** 3, PEAK_FP64_MFMA_TFLOPS      # i8x6f: the second product runs on the fp64 matrix cores
                out["roofline"]["second_product"] = {
                    "kernel": kernel_label(plan2.get("second_product")),
                    "tile_share": share2,
                    "avg_launch_us": 1e6 * a2,
                    "executed_flops_per_launch": ef2 * share2,
                    "frac": ef2 * share2 / a2 / 1e12 / peak2,
                    "executed_frac": ef2 * share2 / a2 / 1e12 / peak2,
                    "algorithmic_TFLOPs": flops / a2 / 1e12, "algorithmic_frac": flops / a2 / 1e12 / peak,
                    "traffic": traffic2,
                    "traffic_source": ("profiles/pmc_traffic.json (static, rocprofv3 --pmc)" if traffic2 is not None else None),
                    "measured": "instrumented pass after the timed region (events around every launch)"}
                out["roofline"]["laplacian_inverse"] = {
                    "kernel": (plan2.get("laplacian_inverse") or {}).get("kernel"), "bound": "hbm", "avg_launch_us": 1e6 * a0,
                    "algorithmic_bytes_per_launch": (20.0 if c64 else 40.0) * N * N,
                    "achieved_GBs": (20.0 if c64 else 40.0) * N * N / a0 / 1e9,
                    "peak_GBs": PEAK_HBM_GBS, "frac": (20.0 if c64 else 40.0) * N * N / a0 / 1e9 / PEAK_HBM_GBS}
                if c64:
                    b = step_bound_c64(N, st["iterations"], share2)
                    b["measured_ms_per_step"] = 1e3 * elapsed / args.steps
                    b["frac"] = b["bound_ms_per_step"] / b["measured_ms_per_step"]
                    b["kernel_us_per_iteration"] = {"k_solve": 1e6 * a0, "first_product": 1e6 * a1, "second_product": 1e6 * a2,
                                                    "sum": 1e6 * (a0 + a1 + a2)}
                    b["how"] = ("(executed flops / %.1f TFLOP/s + (20 + 120) N^2 B / 8 TB/s) x iterations + 24 N^2 B / 8 TB/s, "
                                "over the measured step" % PEAK_FP32_MFMA_TFLOPS)
                    out["roofline"]["whole_step"] = b
                if args.products == "f64" and not c64:
                    b = step_bound(N, st["iterations"], share2)
                    b["measured_ms_per_step"] = 1e3 * elapsed / args.steps
                    b["frac"] = b["bound_ms_per_step"] / b["measured_ms_per_step"]
                    b["kernel_us_per_iteration"] = {"k_solve": 1e6 * a0, "first_product": 1e6 * a1, "second_product": 1e6 * a2,
                                                    "sum": 1e6 * (a0 + a1 + a2),
                                                    "wall_per_iteration_in_the_timed_region": 1e6 * elapsed / max(executed, 1)}
                    b["how"] = ("(executed flops / %.1f TFLOP/s + (40 + 240) N^2 B / 8 TB/s) x iterations + 48 N^2 B / 8 TB/s, "
                                "over the measured step" % PEAK_FP64_MFMA_TFLOPS)
                    out["roofline"]["whole_step"] = b
                    out["roofline"]["fixed_iterations_10"] = fixed_iteration_run(qfa, _lib, W0, N, device_index)
        else:
            out["roofline"] = None
        if (world == 1 and args.products == "f64" and args.stepper == "isomp" and not args.no_config3
                and injected is None and N % 64 == 0 and N >= 256):
            # BASELINE.json config 3: the low-precision-MFMA commutator = six int8 digits
            # (round 4: six digits for the first product, five for the second -- `i8x65`; the six-and-six form of
            # rounds 1-3 beside it)
            out["config3_lowprecision_products"] = config3_side_run(args, qfa, tr, W0, dt, kw, device_index, "i8x65")
            out["config3_lowprecision_products"]["vs_fp64_headline"] = out["config3_lowprecision_products"]["value"] / out["value"]
            if not args.no_side_runs:
                alt = config3_side_run(args, qfa, tr, W0, dt, kw, device_index, "i8x6")
                out["config3_lowprecision_products"]["six_digits_both_products"] = {
                    k: alt[k] for k in ("products", "value", "value_without_prewarm", "max_abs_state_diff_vs_f64_run",
                                        "casimir_drift", "spectrum_drift") if k in alt}
            if N == 1024 and args.ic == "A" and not kw and not args.no_side_runs:
                # the other two target sizes of BASELINE.json's north_star, same process, fp64 products
                out["other_sizes"] = {"N512": other_size_run(args, qfa, 512, 200, 20, device_index),
                                      "N2048": other_size_run(args, qfa, 2048, 60, 6, device_index)}
                # config 3's products at N = 2048 beside the fp64 line of that size
                out["config3_lowprecision_products"]["N2048"] = config3_other_size_run(
                    args, qfa, 2048, 60, 6, device_index, out["other_sizes"]["N2048"]["value"])
                # ensembles with more replicas than GPUs: several trajectories per GPU, advanced together
                # (N = 1024: declined -- one fp64 workgroup owns a CU, two replicas can only fill each other's idle CUs:
                # 1.07-1.08 x measured, ceiling 1.10; the row left the line in round 5)
                out["replicas_per_gpu"] = {"N512_x4": replicas_per_gpu_run(args, qfa, 512, 4, 300, device_index)}
                # complex64 input: single precision throughout, as the reference computes it
                out["complex64_state"] = {"N1024": complex64_side_run(args, qfa, 1024, 200, 20, device_index),
                                          "N512": complex64_side_run(args, qfa, 512, 400, 20, device_index)}
                # smooth initial data (IC-B: ~7 iterations per step instead of ~2), headline size
                out["smooth_data"] = {"N1024": smooth_data_side_run(args, qfa, 1024, 100, 10, device_index)}
        if world == 1 and args.cpu_seconds > 0 and injected is None:
            out["cpu_baseline"] = cpu_baseline(args, dt)
            for key, n_side in (("N512", 512), ("N2048", 2048)):
                if key in (out.get("other_sizes") or {}):      # bounded: <= 3 s of CPU work each (+ one warm-up step)
                    import quflow_amd as _q
                    out["other_sizes"][key]["cpu_baseline"] = cpu_baseline(args, args.stepsize * _q.hbar(n_side), N=n_side, seconds=3.0,
                                                                          min_steps=10)
            if "N1024" in (out.get("smooth_data") or {}):
                smooth_data_oracle_check(args, out["smooth_data"]["N1024"], 1024)
            if "other_sizes" in out and not args.no_per_call:
                # the public per-call API on host arrays, beside the oracle's (the reference harness's other rows)
                out["per_call"] = per_call_run(qfa)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if os.environ.get("QUFLOW_BENCH_ASSERT_NO_TORCH") and "torch" in sys.modules:
        print("bench.py: torch was imported on the torch-free path", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
