"""Independent-initial-condition ensembles: one replica (or more) per GPU.

This is a NEW capability relative to the reference: quflow has
no distributed code, and a single trajectory does not shard (time steps and fixed-point
iterations are sequential; its batched (k,N,N) input is not an ensemble).  Replicas are
embarrassingly parallel: rank r owns seeds r, r+world, ...; there is no data-path
collective.  The only communication is one all_gather of a few diagnostic scalars per
output chunk through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
node, "gloo" in the CPU tests) or, torch-free, through quflow_amd.comm.NativeComm (RCCL behind
the C ABI, qf_comm_*).
"""
import numpy as np


def shard(items, rank, world):
    """Round-robin replica -> rank partition."""
    return list(items)[rank::world]


def make_W0(N, seed):
    """Deterministic synthetic initial condition IC-A: PCG64(seed),
    A = randn + i randn, W = A - A^H, trace removed, ||W||_F = sqrt(N) (enstrophy 1/2)."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
    W = A - A.conj().T
    W -= np.eye(N) * (np.trace(W) / N)
    W /= np.linalg.norm(W, "fro") / np.sqrt(N)
    return W


def _has_flat_all_gather(dist):
    """Whether this process group has `all_gather_into_tensor` -- decided from what the backend IS (the same answer on
    every rank, before anyone communicates), never by catching an error of the collective itself: a rank-local
    RuntimeError (a timeout, an asynchronous RCCL error) would otherwise send that one rank into a different
    collective than its peers.  gloo has no flat form (torch 2.x raises "no support for _allgather_base")."""
    if not hasattr(dist, "all_gather_into_tensor"):
        return False
    try:
        return str(dist.get_backend()).lower() in ("nccl", "rccl")
    except Exception:
        return False


def gather_diagnostics(local_rows, dist=None, device=None, rows_per_rank=None):
    """all_gather of per-replica rows [seed, energy, enstrophy, iterations] -> (n_total, 4)
    float64 array on every rank.  `dist` is torch.distributed (initialised) or None for a
    single process.  Ranks may own different numbers of replicas; `rows_per_rank` = n says every rank owns exactly n
    (the seeds divide evenly, as in BASELINE config 4 and in bench.py): ONE collective per chunk instead of two, and one
    device-to-host copy instead of one per rank."""
    local = np.asarray(local_rows, dtype=np.float64).reshape(-1, 4)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if rows_per_rank is not None and local.shape[0] != rows_per_rank:
        # (a caller error: rows_per_rank is derived from the shard sizes, which every rank computes alike -- run_ensemble
        # passes it only when the seeds divide evenly -- so a mismatch is the same mismatch on every rank)
        raise ValueError("gather_diagnostics: %d local rows where every rank was said to own %d" % (local.shape[0], rows_per_rank))
    if hasattr(dist, "allgather_f64"):
        # quflow_amd.comm.NativeComm: RCCL through the C ABI, no torch in the process
        if rows_per_rank is not None:
            return dist.allgather_f64(local.ravel()).reshape(world * rows_per_rank, 4)
        counts = dist.allgather_f64([float(local.shape[0])])[:, 0].astype(int)
        nmax = int(counts.max())
        pad = np.zeros((nmax, 4))
        pad[:local.shape[0]] = local
        blocks = dist.allgather_f64(pad.ravel()).reshape(world, nmax, 4)
        return np.concatenate([blocks[r, :counts[r]] for r in range(world)], axis=0)
    import torch
    dev = device if device is not None else "cpu"
    if rows_per_rank is not None:
        buf = torch.from_numpy(local).to(dev)
        if _has_flat_all_gather(dist):
            out = torch.empty((world * rows_per_rank, 4), dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(out, buf)          # (an error here is a real one: it propagates on this rank)
            return out.cpu().numpy()
        bufs = [torch.zeros_like(buf) for _ in range(world)]
        dist.all_gather(bufs, buf)
        return torch.cat(bufs, dim=0).cpu().numpy()
    count = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    nmax = int(max(int(c.item()) for c in counts))
    buf = torch.zeros((nmax, 4), dtype=torch.float64, device=dev)
    if local.shape[0]:
        buf[:local.shape[0]] = torch.from_numpy(local).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    rows = [b[:int(c.item())].cpu().numpy() for b, c in zip(bufs, counts)]
    return np.concatenate(rows, axis=0) if rows else local


def run_ensemble(N, seeds, dt, steps, steps_out=None, dist=None, device=None, trajectory_factory=None,
                 stepper_kwargs=None, rank=None, world=None):
    """Advance the replicas owned by this rank and gather diagnostics every `steps_out` steps.

    trajectory_factory(W0) must return an object with advance(dt, steps, **kw) -> stats dict,
    diagnostics() -> (energy, enstrophy); the default is the device-resident
    quflow_amd.integrators.DeviceTrajectory (HIP) -- a rank that owns several seeds advances them
    together as a DeviceEnsemble (k trajectories on one GPU, overlapped).  Returns a list with one (n_total, 4)
    array per output chunk, rows sorted by seed.
    """
    device_ensemble = trajectory_factory is None
    if trajectory_factory is None:
        from .integrators import DeviceEnsemble, DeviceTrajectory
        trajectory_factory = DeviceTrajectory
    if rank is None:
        rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    if world is None:
        world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    steps_out = steps if steps_out is None else min(steps_out, steps)
    kw = dict(stepper_kwargs or {})
    mine = shard(seeds, rank, world)
    group = None
    if device_ensemble and len(mine) > 1 and not (kw.get("compsum") or kw.get("reinitialize")):
        # several replicas on this rank's GPU: one host loop feeds all their streams (qf_isomp_multi: default
        # stepper options only -- falsy compsum / reinitialize keys say just that and are not forwarded)
        group = DeviceEnsemble([make_W0(N, seed) for seed in mine])
        trajs = list(zip(mine, group.members))
        group_kw = {k: v for k, v in kw.items() if k not in ("compsum", "reinitialize")}
    else:
        trajs = [(seed, trajectory_factory(make_W0(N, seed))) for seed in mine]
    # every rank owns the same number of seeds (config 4: 8 seeds on 8 ranks): one collective per chunk
    even_rows = len(mine) if (world > 1 and len(seeds) % world == 0) else None
    history = []
    done = 0
    while done < steps:
        n = min(steps_out, steps - done)
        rows = []
        if group is not None:
            sts = group.advance(dt, n, **group_kw)
            for (seed, tr), st in zip(trajs, sts):
                e, s = tr.diagnostics()
                rows.append([float(seed), e, s, st["iterations"]])
        else:
            for seed, tr in trajs:
                if device_ensemble:
                    st = tr.advance(dt, n, diagnostics=True, **kw)      # one synchronisation per chunk
                    e, s = st["energy"], st["enstrophy"]
                else:
                    st = tr.advance(dt, n, **kw)
                    e, s = tr.diagnostics()
                rows.append([float(seed), e, s, st["iterations"]])
        allrows = gather_diagnostics(rows, dist=dist, device=device, rows_per_rank=even_rows)
        history.append(allrows[np.argsort(allrows[:, 0], kind="stable")])
        done += n
    return history, trajs
