"""First-contact GPU check: kernel-level entry points vs the oracle, then a few steps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
from tests import oracle_py as O

name = sys.argv[1] if len(sys.argv) > 1 else "bunny5K_LTSS"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
sc, ep, nparts = load_workload(name)
cfg = sc.cfg
t0 = time.time()
ts = DOTTimeStepper(sc, ep, nparts)
print("create %.2fs tol %.9g" % (time.time() - t0, ts.targetGRes), flush=True)
orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, nparts, cfg.with_gravity)
print("oracle tol %.9g" % orc.target_gres)
def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
A, vol, mass = ts.features(); Ao, volo, masso, _, _ = orc.features()
print("features", rel(A, Ao), rel(vol, volo), rel(mass, masso))
rng = np.random.default_rng(0)
x = sc.x0 + 0.01 * rng.standard_normal(sc.x0.shape)
print("E    ", ts.computeEnergyVal(x), orc.energy(x))
g = ts.computeGradient(x); go = orc.gradient(x)
print("grad ", rel(g, go))
H = ts.computeElemHessians(x); Ho = orc.elem_hessians(x)
print("Helem", rel(H, Ho), "bitwise equal frac", (H == Ho).mean())
H0 = ts.computeElemHessians(sc.x0); Ho0 = orc.elem_hessians(sc.x0)
print("Helem@rest", rel(H0, Ho0), "per-elem max", np.abs(H0 - Ho0).reshape(len(H0), -1).max(axis=1).max())
ts.updatePrecondMtrAndFactorize(x); orc.refactor(x)
p = rng.standard_normal(x.shape); p[sc.fixed.astype(bool)] = 0
print("spmv ", rel(ts.multiply(p), orc.spmv(p)))
print("prec ", rel(ts.applyPrecond(p), orc.apply_precond(p)))
M, l2g = ts.partMatrix(0, False); Mo = orc.part_dense(0)
print("Hs   ", rel(M, Mo), (l2g == orc.part_verts(0)).all())
Mi, _ = ts.partMatrix(0, True)
print("Hs^-1", np.abs((Mi.T @ Mi) @ Mo - np.eye(len(Mo))).max(), "upper-zero", np.abs(np.triu(Mi, 1)).max())
ts.updatePrecondMtrAndFactorize(sc.x0); orc.refactor(sc.x0)
for step in range(nsteps):
    xs = ts.getResult()
    idx, pos = sc.scripter.step(xs, cfg.dt)
    # scripter state mutates (velocity flips): use one scripter call for both
    ts.setDirichlet(idx, pos); orc.move(idx, pos)
    t0 = time.time(); st = ts.step(); t1 = time.time(); so = orc.step(); t2 = time.time()
    xg = ts.getResult(); xo = orc.state()[0]
    print("step %d gpu it %d ls %d E %.10g g2 %.3e | orc it %d ls %d E %.10g g2 %.3e | dx %.3e | gpu %.1f ms (loop %.1f hess %.2f fact %.2f) orc %.0f ms"
          % (step, st.iters, st.ls_halvings, st.E, st.g2, so.iters, so.ls_halvings, so.E, so.g2, np.abs(xg - xo).max(),
             st.ms_total, st.ms_loop, st.ms_hessian, st.ms_factor, so.ms_total), flush=True)
ms, nb = ts.benchPrecond(50)
print("precond kernel %.3f ms/launch, %.1f MB -> %.1f GB/s" % (ms, nb / 1e6, nb / ms / 1e6))
ms, nb = ts.benchEnergy(50)
print("energy kernel %.4f ms/launch, %.2f MB -> %.1f GB/s" % (ms, nb / 1e6, nb / ms / 1e6))
