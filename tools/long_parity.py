"""One-off validation: many steps of a workload on the GPU and in the oracle, iteration counts and positions.
(env RUN_CASE_FLAGS = dotmi flag bits for the GPU side, e.g. 260 = FORCE_DIST | OWNER_EXCHANGE with DOTMI_SHARD_ELEMS=1)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
from tests import oracle_py as O
name = sys.argv[1]; nsteps = int(sys.argv[2])
sc, ep, n = load_workload(name); cfg = sc.cfg
ts = DOTTimeStepper(sc, ep, n, flags=int(os.environ.get("RUN_CASE_FLAGS", "0")))
orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n, cfg.with_gravity)
O.lib().dor_set_threads(16)
same, worst, first_diff = 0, 0.0, None
for k in range(nsteps):
    x = ts.getResult()
    idx, pos = sc.scripter.step(x, cfg.dt)
    ts.setDirichlet(idx, pos); orc.move(idx, pos)
    st, so = ts.step(), orc.step()
    dx = np.abs(ts.getResult() - orc.state()[0]).max()
    worst = max(worst, dx)
    ok = (st.iters, st.ls_halvings) == (so.iters, so.ls_halvings)
    same += ok
    if not ok and first_diff is None:
        first_diff = (k, st.iters, so.iters, st.ls_halvings, so.ls_halvings, dx)
print(f"{name}: {same}/{nsteps} steps with identical (iterations, halvings); max|dx| {worst:.3e}; first difference {first_diff}")
