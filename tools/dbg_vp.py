import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
keep = []
for en in ("FCR", "SNH"):
    sc0, ep0, n0 = load_workload("bunny5K_LTSS"); sc0.cfg.energy = en; t0 = DOTTimeStepper(sc0, ep0, n0); t0.features(); keep.append(t0)
    rng = np.random.default_rng(1); x = sc0.x0 + 0.05 * rng.standard_normal(sc0.x0.shape)
    t0.computeEnergyVal(x); t0.computeGradient(x); t0.computeElemHessians(x)
sc, ep, n = load_workload("monkey18K_stiff")
ts = DOTTimeStepper(sc, ep, n)
x = ts.getResult(); idx, pos = sc.scripter.step(x, sc.cfg.dt); ts.setDirichlet(idx, pos)
try:
    st = ts.step()
    print("ok iters", st.iters, "halv", st.ls_halvings, "E", st.E, "g2", st.g2, "E0", st.E0, "g20", st.g2_0)
except Exception as e:
    print("FAILED", e)
a, e, g2 = ts.iterLog()
print("log n", len(e), "alpha", np.array(a)[:8], "E", np.array(e)[:8], "g2", np.array(g2)[:6])
print("x finite", np.isfinite(ts.getResult()).all())
