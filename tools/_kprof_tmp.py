import ctypes, os, sys
sys.path.insert(0, "/root/repo")
from tests.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
from dot_amd import lib as dl
sc, ep, n = load_workload("bar17K_twist")
ts = DOTTimeStepper(sc, ep, n)
L = dl.load()
for k in range(6):
    idx, pos = sc.scripter.step(ts.getResult(), sc.cfg.dt); ts.setDirichlet(idx, pos); ts.step()
buf = (ctypes.c_ulonglong * 64)()
L.dotmi_debug_kprof(buf)
for i, nm in [(0,"elem"),(1,"gather"),(2,"spmv"),(8,"spmv: until block sums"),(9,"spmv: start of 2nd trip")]:
    s, mx, c = buf[4 * i], buf[4 * i + 1], buf[4 * i + 2]
    print(nm, "WG avg %.2f us max %.2f us count %d" % (s / max(c, 1) / 100.0, mx / 100.0, c))
