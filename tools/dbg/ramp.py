import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import quflow_amd as qfa
N = 1024
W0 = qfa.ensemble.make_W0(N, 0)
dt = 0.25 * qfa.hbar(N)
tr = qfa.DeviceTrajectory(W0)
tr.advance(dt, 20); tr.sync()
def timed(n):
    t0 = time.perf_counter(); tr.advance(dt, n); tr.sync(); return n / (time.perf_counter() - t0)
for rep in range(2):
    time.sleep(0.2); a = timed(200)
    time.sleep(0.2); tr.advance(dt, 5); tr.sync(); b = timed(200)
    c = timed(200)          # back to back, no sleep
    d = timed(2000)
    time.sleep(0.2); e = timed(50)
    print("after sleep: %.0f | sleep+5 warm steps: %.0f | back-to-back: %.0f | 2000 steps: %.0f | 50 steps after sleep: %.0f" % (a, b, c, d, e))
