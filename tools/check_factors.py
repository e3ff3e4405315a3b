"""X^T X H_s = I for every subdomain of a workload under the current environment (layout switches): python tools/check_factors.py <workload>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
sc, ep, n = load_workload(sys.argv[1] if len(sys.argv) > 1 else "bunny5K_LTSS")
ts = DOTTimeStepper(sc, ep, n)
worst = 0.0
for p in range(min(n, int(sys.argv[2]) if len(sys.argv) > 2 else n)):
    H, l2g = ts.partMatrix(p, inverse=False)
    X, _ = ts.partMatrix(p, inverse=True)
    live = np.abs(np.diag(H)) > 0
    E = (X.T @ X @ H)[np.ix_(live, live)] - np.eye(int(live.sum()))
    err = float(np.abs(E).max())
    worst = max(worst, err)
    if not np.isfinite(err) or err > 1e-6:
        print("part", p, "size", H.shape[0], "live", int(live.sum()), "max |X^T X H - I|", err, "finite X", bool(np.isfinite(X).all()))
print("worst", worst)
r = np.random.default_rng(0).standard_normal(sc.x0.shape) * (1 - sc.fixed[:, None])
z = ts.applyPrecond(r)
print("apply_precond finite", bool(np.isfinite(z).all()), "norm", float(np.abs(z).max()))
