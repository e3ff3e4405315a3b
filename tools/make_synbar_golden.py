"""Full-size oracle fixture for BASELINE.json configs[4]: runs oracle/dot_oracle.c -- with the reference's
own CHOLMODSolver doing the subdomain factorisations and solves (oracle/_ref/librefsolver.so), which is what makes 256
subdomains of ~3.2 k dofs feasible on 8 host cores -- on the 1 M-tet synthetic bar (synbar:140x35x35:256) for the first
time steps in the BUILD container and stores, per step: iterations, halvings, energy evaluations, (E0, |g|^2_0), the
per-iteration (alpha, E, |g|^2) log and the positions of a fixed 4 096-vertex sample.

    OMP_NUM_THREADS=8 python tools/make_synbar_golden.py [steps=1] [workload=synbar:140x35x35:256]
        -> tests/golden/synbar_1M_oracle.npz   (read by tests/test_gpu_round5.py at full size on the GPU)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dot_amd.workloads import load_workload
from tests import oracle_py as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
name = sys.argv[2] if len(sys.argv) > 2 else "synbar:140x35x35:256"
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                          "synbar_1M_oracle.npz")
O.lib().dor_set_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
t0 = time.time()
sc, ep, n = load_workload(name)
cfg = sc.cfg
nV = sc.V_rest.shape[0]
orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n, cfg.with_gravity)
print("oracle created in %.1f s (%d vertices, %d tets, %d subdomains)" % (time.time() - t0, nV, sc.T.shape[0], n), flush=True)
ref = O.ref_solver_available()
if ref:
    assert O.use_reference_cholmod(orc) == 0
print("linear algebra:", "reference CHOLMODSolver" if ref else "oracle's envelope Cholesky", flush=True)
sample = np.sort(np.random.default_rng(20260930).choice(nV, size=min(4096, nV), replace=False)).astype(np.int32)
rec = {"workload": np.array(name), "sample": sample, "nV": nV, "nT": sc.T.shape[0], "nparts": n,
       "target_gres": orc.target_gres, "reference_cholmod": int(ref)}
for k in range(steps):
    t1 = time.time()
    x = orc.state()[0]
    idx, pos = sc.scripter.step(x, cfg.dt)
    orc.move(idx, pos)
    so = orc.step()
    a, e, g2 = orc.iter_log()
    xs = orc.state()[0]
    assert not orc.factor_failed()
    rec.update({f"iters{k}": so.iters, f"halvings{k}": so.ls_halvings, f"evals{k}": so.energy_evals, f"status{k}": so.status,
                f"E0_{k}": so.E0, f"g20_{k}": so.g2_0, f"E_{k}": so.E, f"g2_{k}": so.g2,
                f"alpha{k}": np.asarray(a), f"Elog{k}": np.asarray(e), f"g2log{k}": np.asarray(g2),
                f"x{k}": xs[sample].copy()})
    print("step %d: %d iterations, %d halvings, E %.12g, |g|^2 %.3e, %.1f s" % (k, so.iters, so.ls_halvings, so.E, so.g2,
                                                                                   time.time() - t1), flush=True)
rec["steps"] = steps
np.savez_compressed(out, **rec)
print("wrote", out, os.path.getsize(out), "bytes")
