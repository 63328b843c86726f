"""Soak of the threading rule of include/quflow_hip.h (round 6): T host threads, one DeviceTrajectory each, sizes mixed, advancing
`chunks` chunks concurrently on ONE GPU; every chunk's state digest / statistics / diagnostics against the same chain run alone.
Usage: python tools/gpu/r6_soak_threads.py [chunks]"""
import hashlib
import json
import sys
import threading
import time

sys.path.insert(0, ".")
import numpy as np
import quflow_amd as qfa

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
cases = [(512, 1, 5), (1024, 2, 3), (256, 3, 8), (768, 4, 3), (1000, 5, 2), (64, 6, 20)]     # (N, seed, steps per chunk)


def chain(N, seed, steps, barrier=None):
    tr = qfa.DeviceTrajectory(qfa.ensemble.make_W0(N, seed))
    dt = 0.25 * qfa.hbar(N)
    rows = []
    if barrier is not None:
        barrier.wait()
    for c in range(chunks):
        st = tr.advance(dt, steps, diagnostics=(c % 10 == 0))
        if c % 10 == 0:
            rows.append((hashlib.sha256(tr.download().tobytes()).hexdigest(), st["total_iterations"], st["tol"], st["last_resnorm"], st["energy"], st["enstrophy"]))
        else:
            rows.append((st["total_iterations"], st["tol"], st["last_resnorm"]))
    tr.ctx.close()
    return rows


t0 = time.time()
alone = [chain(*c) for c in cases]
t_seq = time.time() - t0
barrier = threading.Barrier(len(cases))
got, errs = [None] * len(cases), []


def work(i):
    try:
        got[i] = chain(*cases[i], barrier=barrier)
    except BaseException as e:
        errs.append((i, repr(e)))
        barrier.abort()


t0 = time.time()
ths = [threading.Thread(target=work, args=(i,)) for i in range(len(cases))]
[t.start() for t in ths]
[t.join() for t in ths]
t_par = time.time() - t0
ok = not errs and all(g == a for g, a in zip(got, alone))
print(json.dumps({"threads": len(cases), "cases_N_seed_steps_per_chunk": cases, "chunks_each": chunks, "errors": errs,
                  "bit_identical_to_the_runs_alone": ok, "steps_total": sum(c[2] for c in cases) * chunks,
                  "seconds_one_after_the_other": round(t_seq, 2), "seconds_concurrently": round(t_par, 2)}))
sys.exit(0 if ok else 1)
