"""Tightest position pin this image allows on BASELINE.json configs[1]: the CPU oracle with BOTH compiled
reference pieces plugged in -- every simulation-level SVD through the reference's own AVX kernel (oracle/_ref/librefpin.so:
ref_svd, Utils/SVD_EFTYCHIOS compiled where it lies) and every subdomain factorisation / solve through the reference's own
CHOLMODSolver (oracle/_ref/librefsolver.so) -- stepped on bar17K_twist (StableNH, 32 METIS subdomains) in the BUILD container.
Stored per step: iterations, halvings, energy evaluations, (E0, |g|^2_0), the per-iteration (alpha, E, |g|^2) log and the
positions of a fixed 4 096-vertex sample.

    OMP_NUM_THREADS=8 python tools/make_refpin_golden.py [steps=10] [workload=bar17K_twist]
        -> tests/golden/bar17K_refpin_oracle.npz   (read by tests/test_gpu_round6.py on the GPU)
    REFPIN_SVD=0 ... -> tests/golden/bar17K_refcholmod_oracle.npz   (the reference's CHOLMODSolver only, exact SVD)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from dot_amd.workloads import load_workload
from tests import oracle_py as O

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
name = sys.argv[2] if len(sys.argv) > 2 else "bar17K_twist"
# REFPIN_SVD=0: the second fixture -- the reference's CHOLMODSolver, the oracle's own exact Jacobi SVD (what the device uses too):
# the pair says how much of a difference is the SVD kernel's rounding and how much anything else
ref_svd = os.environ.get("REFPIN_SVD", "1") != "0"
out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                                          "bar17K_refpin_oracle.npz" if ref_svd else "bar17K_refcholmod_oracle.npz")
O.lib().dor_set_threads(int(os.environ.get("OMP_NUM_THREADS", "8")))
sc, ep, n = load_workload(name)
cfg = sc.cfg
nV = sc.V_rest.shape[0]
assert O.ref_solver_available(), "oracle/_ref is not built (make -C oracle ref; needs /root/reference)"
O.use_reference_svd(ref_svd)
orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n, cfg.with_gravity)
assert O.use_reference_cholmod(orc) == 0
sample = np.sort(np.random.default_rng(20260930).choice(nV, size=min(4096, nV), replace=False)).astype(np.int32)
rec = {"workload": np.array(name), "sample": sample, "nV": nV, "nT": sc.T.shape[0], "nparts": n,
       "target_gres": orc.target_gres, "reference_cholmod": 1, "reference_svd": int(ref_svd)}
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
orc.close()
O.use_reference_svd(False)
rec["steps"] = steps
np.savez_compressed(out, **rec)
print("wrote", out, os.path.getsize(out), "bytes")
