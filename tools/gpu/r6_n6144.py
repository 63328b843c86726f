"""One-off (round 6): the stepper at N = 6144 read 1.1 iterations per step in the size sweep where 4096 and 8192 read 2.0.
Three steps of the device against the oracle on the same W0: iteration counts per step and the state."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import quflow_amd as qfa
from oracle import isomp_oracle as oracle
oracle.build()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
W0 = oracle.make_W0(N, 0)
dt = 0.25 * qfa.hbar(N)
tr = qfa.DeviceTrajectory(W0)
its_g = []
for s in range(3):
    st = tr.advance(dt, 1) if s == 0 else tr.advance(dt, 1)
    its_g.append(int(st["total_iterations"]))
# (chunked one step at a time: dW restarts at every call; the single-call form below is the warm-started one)
tr2 = qfa.DeviceTrajectory(W0)
st2 = tr2.advance(dt, 3)
Wg = tr2.download()
t0 = time.time()
sc = {"iterations": 0.0}
Wc = oracle.isomp(W0.copy(), dt, steps=3, stats=sc)
print({"N": N, "device_iterations_one_call_of_3": int(st2["total_iterations"]), "oracle_iterations_one_call_of_3": int(round(sc["iterations"] * 3)),
       "device_iterations_three_calls_of_1": its_g, "max_abs_state_diff": float(np.abs(Wg - Wc).max()), "tol_device": st2["tol"], "tol_oracle": sc.get("tol_auto"),
       "last_resnorm_device": st2["last_resnorm"], "oracle_seconds": round(time.time() - t0, 1)})
