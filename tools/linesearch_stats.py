"""Per-iteration line-search record of a workload from the CPU oracle (host only): alpha_0, the energy of the first trial, the
energy it has to beat, the accepted step -- the data a forecast of "the first trial will be rejected" (DevLoop::holdNext) can
be judged on.   usage: python tools/linesearch_stats.py <workload> <steps> <out.npz>"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import oracle_py as O
from tests.workloads import load_workload

name, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
sc, ep, n = load_workload(name)
cfg = sc.cfg
orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n, cfg.with_gravity)
if O.ref_solver_available():
    O.use_reference_cholmod(orc, True)
rows = []
for s in range(steps):
    x, v, xt = orc.state()
    idx, pos = sc.scripter.step(x, cfg.dt)
    orc.move(idx, pos)
    t0 = time.time()
    orc.step_begin()
    it = 0
    while True:
        x, g, S, Y, lastE = orc.lbfgs_state()
        pr = orc.probe_direction(x, S, Y)
        p = pr["p"]
        pg = float((p * g).sum())
        pHp = float((p * orc.spmv(p)).sum())
        rc = orc.step_iterate()
        a, e, g2 = orc.iter_log()
        rows.append((s, it, pr["alpha0"], pr["E"], lastE, a[-1], e[-1], g2[-1], pg, len(S), pHp))
        it += 1
        if rc != 0:
            break
    st = orc.step_end()
    print(f"step {s}: {it} iterations, {st.ls_halvings if hasattr(st,'ls_halvings') else '?'} halvings, {time.time()-t0:.1f} s", flush=True)
np.savez(out, rows=np.array(rows))
