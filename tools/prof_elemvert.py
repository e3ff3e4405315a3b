"""Stage times inside the one-launch element pass + gather (library built with -DK_PROFILE: tools/prof_elemvert.sh), from the
stamps of thread 0 of every workgroup, on the kernel as the running loop launches it (the last trial of a step)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
from dot_amd import lib as dl
sc, ep, n = load_workload(sys.argv[1] if len(sys.argv) > 1 else "bar17K_twist")
ts = DOTTimeStepper(sc, ep, n)
for _ in range(3):
    x = ts.getResult(); idx, pos = sc.scripter.step(x, sc.cfg.dt); ts.setDirichlet(idx, pos); st = ts.step()
L = dl.load()
buf = (ctypes.c_longlong * (512 * 8))()
L.dotmi_debug_evprof.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
assert L.dotmi_debug_evprof(buf) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(512, 8)
a = a[a[:, 0] > 0]
t0 = a[:, 7].min()
T = (a[:, :7] - t0) / 100.0
print("   %-70s p50 %.2f  p95 %.2f us" % ("kernel start -> loop state there", np.median((a[:, 0] - a[:, 7]) / 100.0), np.percentile((a[:, 0] - a[:, 7]) / 100.0, 95)))
print("%d workgroups, first start -> last end %.2f us; start p50 %.2f max %.2f" % (len(a), T[:, 6].max(), np.median(T[:, 0]), T[:, 0].max()))
for k, nm in enumerate(["loop state there -> every load requested", "alpha (SpMV partials) + barrier", "positions there, trial points in LDS + barrier",
                        "2 x 256 elements, owned corners' entries in LDS + barrier", "run sums, gradient / pair / right-hand side stores, statistics",
                        "block sums -> end"]):
    d = T[:, k + 1] - T[:, k]
    print("   %-70s p50 %.2f  p95 %.2f us" % (nm, np.median(d), np.percentile(d, 95)))
d = T[:, 6] - T[:, 0]
print("   %-70s p50 %.2f  p95 %.2f us" % ("workgroup lifetime", np.median(d), np.percentile(d, 95)))
