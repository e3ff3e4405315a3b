"""two-level back-solve against the one-pass form: M r on random right-hand sides, then a few time steps
    python tools/tl_check.py <workload> [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper

name = sys.argv[1] if len(sys.argv) > 1 else "bar17K_twist"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
res = {}
extra = dict(kv.split("=") for kv in os.environ.get("TL_ENV", "").split())   # e.g. TL_ENV="DOTMI_ND_LEVELS=5 DOTMI_ND_MIN=128"
runs = (("one-pass", {"DOTMI_TWO_LEVEL": "0"}), ("two-level", dict({"DOTMI_TWO_LEVEL": "1"}, **extra)))
if os.environ.get("TL_ONLY"):
    runs = runs[1:] * 2   # (the two-level form alone, twice: profiles)
    runs = (("one-pass", runs[0][1]), runs[1])
for tag, env in runs:
    for k in extra:
        os.environ.pop(k, None)
    os.environ.update(env)
    sc, ep, n = load_workload(name)
    ts = DOTTimeStepper(sc, ep, n)
    rng = np.random.default_rng(5)
    r = rng.standard_normal((sc.V_rest.shape[0], 3)) * (1 - sc.fixed[:, None])
    p = ts.applyPrecond(r)
    its, ms = [], []
    for s in range(steps):
        t0 = time.perf_counter()
        st = ts.step()
        ms.append((time.perf_counter() - t0) * 1e3)
        its.append(st.iters)
    res[tag] = (p, ts.getResult().copy(), its, ms, (st.ms_total, st.ms_loop, st.ms_factor, st.precond_bytes))
    ts.close()
a, b = res["one-pass"], res["two-level"]
print(name, "M r: max|diff|", np.abs(a[0] - b[0]).max(), "of max", np.abs(a[0]).max(), "nan", np.isnan(b[0]).sum())
print("iterations", a[2], b[2])
print("positions max|diff|", np.abs(a[1] - b[1]).max())
print("ms per step (wall)", [round(x, 2) for x in a[3]], [round(x, 2) for x in b[3]])
print("last step: ms total / loop / factor, bytes per application", a[4], b[4])
