"""Per-workgroup phase times of one element-pass launch (library built with -DEP_PROFILE: tools/prof_elem.sh)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
from dot_amd import lib as dl
sc, ep, n = load_workload(sys.argv[1] if len(sys.argv) > 1 else "synbar:140x35x35:256")
ts = DOTTimeStepper(sc, ep, n)
L = dl.load()
ms, nb = ctypes.c_double(), ctypes.c_int64()
L.dotmi_bench_kernel(ts._h, 0, 5, ctypes.byref(ms), ctypes.byref(nb))     # elem_energy_grad, stand-alone
N = 8192
buf = (ctypes.c_longlong * (6 * N))()
L.dotmi_debug_ep_prof.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
assert L.dotmi_debug_ep_prof(buf, N) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(N, 6)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
T = (a[:, :5] - t0) / 100.0
print("launch %.1f us by events; workgroups %d, span %.1f us" % (1e3 * ms.value, len(a), T[:, 4].max()))
names = ["start -> operands + positions in LDS", "element work + runs in LDS", "run sums + partial stores issued", "energy reduce -> end"]
for k, nm in enumerate(names):
    d = T[:, k + 1] - T[:, k]
    print("  %-40s mean %.2f  p50 %.2f  p95 %.2f us" % (nm, d.mean(), np.median(d), np.percentile(d, 95)))
d = T[:, 4] - T[:, 0]
print("  %-40s mean %.2f  p50 %.2f  p95 %.2f us" % ("workgroup lifetime", d.mean(), np.median(d), np.percentile(d, 95)))
grid = np.arange(0, T[:, 4].max() + 1, 4.0)
print("busy workgroups every 4 us:", [int(((T[:, 0] <= t) & (T[:, 4] > t)).sum()) for t in grid])
