"""Step by step: the default device loop (early back-solve), the q-based loop (DOTMI_EARLY_BACKSOLVE=0) and the CPU oracle on
one workload -- iterations / back-tracking per step and the largest position difference between the three.
python tools/early_divergence.py <workload> <steps> [oracle=1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
name = sys.argv[1]; steps = int(sys.argv[2]); with_orc = len(sys.argv) < 4 or sys.argv[3] != "0"
sc, ep, n = load_workload(name)
a = DOTTimeStepper(sc, ep, n)
os.environ["DOTMI_EARLY_BACKSOLVE"] = "0"
sc2, _, _ = load_workload(name)
b = DOTTimeStepper(sc2, ep, n)
del os.environ["DOTMI_EARLY_BACKSOLVE"]
orc = None
if with_orc:
    from tests import oracle_py as O
    cfg = sc.cfg
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n, cfg.with_gravity)
for k in range(steps):
    for ts, s_ in ((a, sc), (b, sc2)):
        idx, pos = s_.scripter.step(ts.getResult(), s_.cfg.dt)
        ts.setDirichlet(idx, pos)
    sa, sb = a.step(), b.step()
    line = f"{k:3d} early {sa.iters:3d}/{sa.ls_halvings:3d}  q-based {sb.iters:3d}/{sb.ls_halvings:3d}  |x_e - x_q| {np.abs(a.getResult() - b.getResult()).max():.2e}"
    if orc is not None:
        orc.move(idx, pos)   # the scripted handles do not depend on the free vertices
        so = orc.step()
        xo = orc.state()[0]
        line += f"  oracle {so.iters:3d}/{so.ls_halvings:3d}  |x_e - x_o| {np.abs(a.getResult() - xo).max():.2e}  |x_q - x_o| {np.abs(b.getResult() - xo).max():.2e}"
    print(line, flush=True)
