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

The state is resident in HBM before the timed region starts; the timed region is
bracketed by a barrier + device synchronisation on both sides.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (the complex GEMM pair, fp64 MFMA): algorithmic
                  flops per launch (8 N^3) / mean launch duration measured with HIP events
                  on the launch stream during the timed region.
  cpu_baseline -- the CPU oracle (oracle/isomp_oracle.py: numpy zgemm + OpenMP Thomas),
                  timed on this host's cores on a bounded sample of the same workload
                  (rank 0, --gpus 1 only).  The oracle is the checker/baseline, never the
                  thing measured as `value`.
"""
import argparse
import ctypes
import json
import os
import sys
import time

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
    for var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(var, str(limit))
    return limit


HOST_THREADS = _limit_host_thread_pools()

METRIC = "isospectral timesteps/sec at N=1024 (1 GPU) + ensemble steps/sec at 1/2/4/8 GPUs"
# MI355X fp64 matrix peak (datasheet, dense): 256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz.
# MI355X_MICROARCH.md lists no f64 MFMA row; bench.py --mfma-probe measures the issue rate.
PEAK_FP64_MFMA_TFLOPS = 78.6
EVENT_STRIDE = 8      # per-launch HIP events of the timed region: one product launch in 8 is bracketed
# int8 matrix peak, dense: v_mfma_i32_32x32x32_i8 = 65,536 ops / 32 clk / SIMD = 2 x the bf16 rate
# (MI355X_MICROARCH.md, MFMA table, I8 row): 256 x 4 x 2048 x 2.4 GHz
PEAK_I8_MFMA_TOPS = 5033.0
I8_OPS_PER_PRODUCT = 45 * 2.0      # per N^3: 15 digit pairs x 3 real products (ozaki.hip), 2 ops per MAC


def parse():
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
    ap.add_argument("--products", choices=["f64", "i8", "i8x6"], default="f64",
                    help="f64: both commutator products on the fp64 matrix cores (headline, full parity); "
                         "i8 / i8x6: BASELINE.json config 3, digit-split products (5 / 6 base-128 digits) on the int8 "
                         "matrix cores + fp64 Laplacian")
    ap.add_argument("--no-config3", action="store_true",
                    help="skip the short int8-products side measurement the default single-GPU run appends")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="GPU clock warm-up before the W warm-up steps: a scratch trajectory of the same workload "
                         "is advanced for this long (a fresh process finds the GPU idle; its clock takes ~100 ms of "
                         "load to come up).  0 disables")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget; 0 disables")
    ap.add_argument("--cpu-cores", type=int, default=16, help="threads for the CPU baseline (BLAS + OpenMP)")
    ap.add_argument("--no-kernel-events", action="store_true", help="no per-launch HIP events in the timed region")
    ap.add_argument("--kernel-table", action="store_true",
                    help="HIP events around every hot-path launch; per-kernel table on stderr (diagnostic)")
    return ap.parse_args()


def cpu_baseline(args, dt):
    """Oracle (CPU restatement of the reference path) on a bounded sample of the workload."""
    import numpy as np
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
    kw = {}
    if args.fixed_iters:
        kw = dict(minit=args.fixed_iters, maxit=args.fixed_iters)
    if args.compsum:
        kw["compsum"] = True
    t0 = time.perf_counter()
    oracle.isomp(W, dt, steps=1, **kw)            # warm-up (BLAS threads, caches, table build)
    t1 = time.perf_counter() - t0
    steps = int(max(2, min(200, args.cpu_seconds / max(t1, 1e-3))))
    stats = {"iterations": 0.0}
    t0 = time.perf_counter()
    oracle.isomp(W, dt, steps=steps, stats=stats, **kw)
    el = time.perf_counter() - t0
    return {"value": steps / el, "unit": "timesteps/s", "cores": int(cores), "kind": "port",
            "sample": "%d steps of the same N=%d workload (IC-%s, dt=%.2f*hbar, %.2f its/step), %.1f s, "
                      "numpy/OpenBLAS zgemm + OpenMP Thomas (oracle/)" %
                      (steps, args.N, args.ic, args.stepsize, stats["iterations"], el)}


def other_size_run(args, qfa, N, steps, warmup, device):
    """The same workload at another target size of BASELINE.json (N = 512, 2048; fp64 products),
    measured in the same process: rate and the first product's fraction of the fp64 MFMA roofline
    (HIP events around its launches, as for the headline)."""
    from quflow_amd import _lib
    W0 = qfa.ensemble.make_W0(N, 0)
    dt = args.stepsize * qfa.hbar(N)
    tr = qfa.DeviceTrajectory(W0, device=device)
    lib, h = tr.ctx._lib, tr.ctx.handle
    tr.advance(dt, warmup)
    _lib.check(lib.qf_profile_reset(h))
    _lib.check(lib.qf_profile_stride(h, EVENT_STRIDE))
    _lib.check(lib.qf_profile_enable(h, (1 << _lib.KERNEL_IDS["gemm1"]) | (1 << _lib.KERNEL_IDS["gemm2"])))
    tr.sync()
    time.sleep(0.05)
    t0 = time.perf_counter()
    st = tr.advance(dt, steps)
    tr.sync()
    el = time.perf_counter() - t0
    _lib.check(lib.qf_profile_enable(h, 0))
    n = ctypes.c_longlong()
    ms = ctypes.c_double()
    executed = max(int(st["total_iterations"]), 1)
    avg = {}
    seen = ctypes.c_longlong()
    for name in ("gemm1", "gemm2"):
        _lib.check(lib.qf_profile_read(h, _lib.KERNEL_IDS[name], ctypes.byref(n), ctypes.byref(ms)))
        _lib.check(lib.qf_profile_seen(h, _lib.KERNEL_IDS[name], ctypes.byref(seen)))
        avg[name] = 1e-3 * ms.value * (seen.value / max(n.value, 1)) / executed
    flops = 8.0 * N ** 3
    e1, s1 = tr.diagnostics()
    tr.ctx.close()
    return {"value": steps / el, "unit": "timesteps/s", "steps": steps, "ms_per_step": 1e3 * el / steps,
            "iterations_per_step": st["iterations"], "first_product_us": 1e6 * avg["gemm1"],
            "second_product_us": 1e6 * avg["gemm2"],
            "roofline_frac_first_product": flops / avg["gemm1"] / 1e12 / PEAK_FP64_MFMA_TFLOPS,
            "enstrophy": s1}


def config3_side_run(args, qfa, tr_f64, W0, dt, kw, device, products="i8"):
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
        tr = qfa.DeviceTrajectory(W0, device=device)
        if args.warmup > 0:
            tr.advance(dt, args.warmup, **kw)
        tr.sync()
        time.sleep(0.1)
        t0 = time.perf_counter()
        st = tr.advance(dt, args.steps, **kw)
        tr.sync()
        el = time.perf_counter() - t0
        W_i8 = tr.download()
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
    res = {"value": args.steps / el, "unit": "timesteps/s", "ms_per_step": 1e3 * el / args.steps,
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


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    torch = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        # launched by torch.distributed.run (also with one rank: exercises the RCCL path).
        # torch first: its bundled HIP runtime must be the one libquflow_hip.so binds to
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    if args.products != "f64":
        os.environ["QUFLOW_HIP_GEMM"] = args.products  # read when the device context is created
    import numpy as np
    import quflow_amd as qfa
    from quflow_amd import _lib
    qfa.set_device(local_rank)
    if qfa.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible; the benchmark has no CPU path")

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

    tr = qfa.DeviceTrajectory(W0, device=local_rank)        # state resident in HBM
    lib, h = tr.ctx._lib, tr.ctx.handle

    def barrier():
        tr.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()

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
    if args.prewarm_ms > 0 and args.stepper == "isomp":
        # clock warm-up on a scratch trajectory (not the measured state, not counted in W or K)
        scratch = qfa.DeviceTrajectory(W0, device=local_rank)
        t_end = time.perf_counter() + 1e-3 * args.prewarm_ms
        while time.perf_counter() < t_end:
            scratch.advance(dt, 10, **kw)
        scratch.sync()
        scratch.ctx.close()
    if args.warmup > 0:
        advance(args.warmup)
    e0, s0 = (0.0, 0.0) if os.environ.get("BENCH_SKIP_DIAG0") else tr.diagnostics()

    gemm_mask = (1 << _lib.KERNEL_IDS["gemm1"]) | (1 << _lib.KERNEL_IDS["gemm2"])
    _lib.check(lib.qf_profile_reset(h))
    if args.kernel_table:
        gemm_mask = (1 << len(_lib.KERNEL_IDS)) - 1
    if not args.no_kernel_events:
        # HIP events around the product launches of the timed region: every launch for the kernel
        # table, otherwise one launch in EVENT_STRIDE (bracketing every launch costs ~6 % of the rate)
        # (short runs keep every launch: a handful of samples would be noise)
        stride = 1 if (args.kernel_table or args.steps < 25) else (2 if args.steps < 100 else EVENT_STRIDE)
        _lib.check(lib.qf_profile_stride(h, stride))
        _lib.check(lib.qf_profile_enable(h, gemm_mask))

    barrier()
    t0 = time.perf_counter()
    _lib.check(lib.qf_timer_start(h))
    st = advance(args.steps)
    e1, s1 = tr.diagnostics()
    table = qfa.ensemble.gather_diagnostics([[float(seed), e1, s1, st["iterations"]]], dist=dist,
                                            device=(torch.device("cuda", local_rank) if dist is not None else None))
    ev_ms = ctypes.c_double()
    _lib.check(lib.qf_timer_stop(h, ctypes.byref(ev_ms)))
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.check(lib.qf_profile_enable(h, 0))

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-launch durations of the two products (HIP events on the launch stream)
    n = ctypes.c_longlong()
    ms = ctypes.c_double()
    per = {}
    seen = ctypes.c_longlong()
    for name in ("gemm1", "gemm2"):
        _lib.check(lib.qf_profile_read(h, _lib.KERNEL_IDS[name], ctypes.byref(n), ctypes.byref(ms)))
        _lib.check(lib.qf_profile_seen(h, _lib.KERNEL_IDS[name], ctypes.byref(seen)))
        # (measured launches, their time scaled to all launches of the kernel in the timed region)
        per[name] = (n.value, ms.value * (seen.value / n.value if n.value else 0.0))
    launches = per["gemm1"][0] + per["gemm2"][0]
    gemm_ms = per["gemm1"][1] + per["gemm2"][1]

    if args.kernel_table and rank == 0:
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
            "dtype": "f64" if args.products == "f64" else
                     "i8 digits (%d x 7 bit, int32 accumulate) for the products, f64 elsewhere" % (6 if args.products == "i8x6" else 5),
            "data": "synthetic",
            "config": {"workload": "%s on random skew-Hermitian "
                                   "trace-free W0, N=%d complex128, dt=%.2f*hbar, IC-%s, one independent "
                                   "trajectory per GPU" % (
                                       "isomp (adaptive fixed-point, tol=auto, maxit=10)" if args.stepper == "isomp"
                                       else args.stepper + " (explicit, quflow/integrators/erk.py)",
                                       N, args.stepsize, args.ic),
                       "stepper": args.stepper, "products": args.products,
                       "N": N, "stepsize": args.stepsize, "ic": args.ic,
                       "iterations_per_step": st["iterations"], "fixed_iters": args.fixed_iters,
                       "compsum": bool(args.compsum), "gpu_clock_prewarm_ms": args.prewarm_ms, "replicas": world, "parallelism": "replicas x%d" % world,
                       "device_ms_per_step_rank0": ev_ms.value / args.steps,
                       "energy_drift": e1 - e0, "enstrophy_drift": s1 - s0,
                       "gathered_rows": int(table.shape[0])},
        }
        if launches:
            flops = 8.0 * N ** 3                      # algorithmic: one complex N^3 GEMM
            # Tagged launches that were not due are no-ops whose (tiny) time stays in the numerator:
            # the averages are per EXECUTED launch.  isomp: one first product (k_zgemm, the dominant
            # kernel) and one second product per executed iteration; the second product is the
            # upper-triangle stream-K kernel k_zgemm_tri when W is skew-Hermitian and N >= 768.
            executed = int(st["total_iterations"])
            traffic = None
            traffic2 = None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if args.products == "f64":
                        traffic = tj.get("zgemm_plain_bytes_per_launch_N%d" % N)
                        traffic2 = tj.get("zgemm_tri_bytes_per_launch_N%d" % N)
                    else:
                        traffic = tj.get("oz_gemm_%s_plain_bytes_per_launch_N%d" % (args.products, N))
                        traffic2 = tj.get("oz_gemm_%s_fused_bytes_per_launch_N%d" % (args.products, N))
                except Exception:
                    traffic = None
            if args.stepper == "isomp":
                avg1 = 1e-3 * per["gemm1"][1] / max(executed, 1)
                avg2 = 1e-3 * per["gemm2"][1] / max(executed, 1)
            else:   # explicit steppers on skew-Hermitian data: one product per right-hand side
                avg1 = 1e-3 * per["gemm1"][1] / max(executed, 1)
                avg2 = None
            ach = flops / avg1 / 1e12
            peak, unit = PEAK_FP64_MFMA_TFLOPS, "TFLOP/s"
            kname = "k_zgemm (first product Phalf@Whalf, v_mfma_f64_16x16x4_f64, 3M)"
            if args.products != "f64" and args.stepper == "isomp":
                # the int8 kernel is priced in the int8 operations it issues: 90 (126) N^3 per product
                flops = (I8_OPS_PER_PRODUCT if args.products == "i8" else 63 * 2.0) * N ** 3
                ach, peak, unit = flops / avg1 / 1e12, PEAK_I8_MFMA_TOPS, "TOP/s"
                kname = "k_oz_gemm (first product Phalf@Whalf, v_mfma_i32_32x32x32_i8, %d digit pairs x 3M)" % (
                    15 if args.products == "i8" else 21)
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": unit,
                               "frac": ach / peak, "traffic": traffic,
                               "kernel": kname,
                               "launches": executed, "launches_timed_with_events": int(per["gemm1"][0]),
                               "avg_launch_us": 1e6 * avg1, "flops_per_launch": flops,
                               "gemm_share_of_step": gemm_ms / (1e3 * elapsed) if world == 1 else None}
            if avg2:
                out["roofline"]["second_product"] = {
                    "kernel": ("k_oz_gemm<fused epilogue> (DESIGN.md 3.6)" if args.products != "f64" else
                               "k_zgemm_tri (upper triangle, stream-K) or k_zgemm+epilogue (see DESIGN.md 3.1b)"),
                    "avg_launch_us": 1e6 * avg2, "algorithmic_TFLOPs": flops / avg2 / 1e12,
                    "frac_of_peak_algorithmic": flops / avg2 / 1e12 / peak,
                    "traffic": traffic2}
        else:
            out["roofline"] = None
        if (world == 1 and args.products == "f64" and args.stepper == "isomp" and not args.no_config3
                and N % 64 == 0 and N >= 256):
            out["config3_int8_products"] = config3_side_run(args, qfa, tr, W0, dt, kw, local_rank, "i8")
            out["config3_int8_products_6_digits"] = config3_side_run(args, qfa, tr, W0, dt, kw, local_rank, "i8x6")
            if N == 1024 and args.ic == "A" and not kw:
                # the other two target sizes of BASELINE.json's north_star, same process, fp64 products
                out["other_sizes"] = {"N512": other_size_run(args, qfa, 512, 200, 20, local_rank),
                                      "N2048": other_size_run(args, qfa, 2048, 60, 6, local_rank)}
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, dt)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
