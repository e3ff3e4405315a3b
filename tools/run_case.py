"""per-step stats of a workload on the GPU:  python tools/run_case.py <workload> [nparts] [steps] [energy]
(env RUN_CASE_FLAGS = dotmi flag bits, e.g. 4 = DOTMI_FLAG_FORCE_DIST: the sharded sequencing on a 1-rank communicator)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
name = sys.argv[1]; nparts = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] != "-" else None
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
sc, ep, n = load_workload(name, nparts)
if len(sys.argv) > 4: sc.cfg.energy = sys.argv[4]
t = time.time(); ts = DOTTimeStepper(sc, ep, n, flags=int(os.environ.get('RUN_CASE_FLAGS', '0'))); print("create %.2f s" % (time.time() - t))
for k in range(steps):
    x = ts.getResult(); idx, pos = sc.scripter.step(x, sc.cfg.dt); ts.setDirichlet(idx, pos)
    st = ts.step()
    print(k, "iters", st.iters, "halv", st.ls_halvings, "ms %.2f loop %.2f hess %.2f fact %.2f" % (st.ms_total, st.ms_loop, st.ms_hessian, st.ms_factor), "status", st.status)
ms, nb = ts.benchPrecond(20); print("backsolve %.1f us, %.1f MB, %.2f TB/s" % (1e3 * ms, nb / 1e6, nb / ms / 1e9))
