"""Does a part-major vertex numbering pay?  The same workload with its mesh renumbered so that the vertices of a subdomain
are consecutive (by first owning part, then by old id) fed to the unchanged library: loop / factor ms per step, iterations.
python tools/renumber_probe.py <workload> [steps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.scene import build_scene
from dot_amd.timestepper import DOTTimeStepper
name = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def run(sc, ep, n, label):
    ts = DOTTimeStepper(sc, ep, n)
    rows = []
    for k in range(steps):
        idx, pos = sc.scripter.step(ts.getResult(), sc.cfg.dt); ts.setDirichlet(idx, pos)
        st = ts.step(); rows.append((st.iters, st.ms_total, st.ms_loop, st.ms_factor))
    a = np.array(rows[2:], dtype=float)
    print(f"{label}: iters {a[:,0].mean():.2f} ms {a[:,1].mean():.3f} loop {a[:,2].mean():.3f} factor {a[:,3].mean():.3f}")
    ts.close()


sc, ep, n = load_workload(name)
run(sc, ep, n, "as numbered ")
nV = sc.V_rest.shape[0]
first = np.full(nV, n, dtype=np.int64)
for k in range(4):
    np.minimum.at(first, sc.T[:, k], ep)
order = np.lexsort((np.arange(nV), first))          # new -> old
inv = np.empty(nV, dtype=np.int64); inv[order] = np.arange(nV)
V2 = sc.V_rest[order]; T2 = inv[sc.T].astype(np.int32)
sc2 = build_scene(sc.cfg, V2, T2)                    # (no rotation in these workloads: normalising again changes nothing)
assert np.abs(sc2.V_rest - V2).max() < 1e-12
run(sc2, ep, n, "part-major  ")
