"""Where the wall time of one bench step goes outside dotmi_step (host harness costs)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
sc, ep, n = load_workload(sys.argv[1] if len(sys.argv) > 1 else "bar17K_twist")
ts = DOTTimeStepper(sc, ep, n)
T = {k: 0.0 for k in ("getResult", "script", "setDirichlet", "step_wall", "step_ms_total", "loop", "hess", "fact")}
N = 20
for k in range(N + 3):
    t0 = time.perf_counter(); x = ts.getResult()
    t1 = time.perf_counter(); idx, pos = sc.scripter.step(x, sc.cfg.dt)
    t2 = time.perf_counter(); ts.setDirichlet(idx, pos)
    t3 = time.perf_counter(); st = ts.step()
    t4 = time.perf_counter()
    if k >= 3:
        for key, v in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, st.ms_total / 1e3, st.ms_loop / 1e3, st.ms_hessian / 1e3, st.ms_factor / 1e3)):
            T[key] += v
print({k: round(1e3 * v / N, 3) for k, v in T.items()})
