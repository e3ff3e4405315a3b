"""GPU-only run of a workload (no oracle): per-step stats."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd import lib as dl
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
name = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t0 = time.time(); sc, ep, n = load_workload(name); t1 = time.time()
ts = DOTTimeStepper(sc, ep, n, flags=dl.FLAG_TIME_BACKSOLVE); t2 = time.time()
print(f"factor storage {dl.load().dotmi_factor_storage_bytes(ts._h)/1e9:.3f} GB", flush=True)
print(f"{name}: nV {sc.V_rest.shape[0]} nT {sc.T.shape[0]} parts {n} | load {t1-t0:.1f}s create {t2-t1:.1f}s tol {ts.targetGRes:.4e}", flush=True)
for k in range(nsteps):
    t = time.time(); rc = ts.solve(1); st = ts.last_stats
    bw = st.precond_bytes * st.precond_launches / max(st.ms_precond, 1e-9) / 1e6
    print(f"step {k} rc {rc} it {st.iters} ls {st.ls_halvings} E {st.E:.8g} g2 {st.g2:.3e} | {1e3*(time.time()-t):.1f} ms (loop {st.ms_loop:.1f} hess {st.ms_hessian:.2f} fact {st.ms_factor:.2f}) back-solve {st.ms_precond/max(st.precond_launches,1)*1e3:.1f} us {bw:.0f} GB/s", flush=True)
