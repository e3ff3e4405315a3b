"""Phase times inside the diagonal / row tile tasks of the factorisation (library built with -DDIAG_PROFILE: tools/prof_diag.sh)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
from dot_amd import lib as dl
sc, ep, n = load_workload(sys.argv[1] if len(sys.argv) > 1 else "bunny5K_LTSS")
ts = DOTTimeStepper(sc, ep, n)          # one factorisation in dotmi_create
L = dl.load()
N = 4096
buf = (ctypes.c_longlong * (8 * N))()
L.dotmi_debug_diag_prof.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
assert L.dotmi_debug_diag_prof(buf, N) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(N, 8)
a = a[a[:, 0] > 0]
for post, name in ((1, "DIAG"), (2, "ROW")):
    m = a[:, 6] == post
    if not m.any():
        continue
    b = a[m]
    d = np.diff(b[:, :5], axis=1) / 100.0      # 100 MHz -> us
    print(f"{name}: {m.sum()} tasks, products mean {b[:,7].mean():.1f} max {b[:,7].max()}")
    print("   first loads + init %.2f | product loop %.2f | %s %.2f | store + drain %.2f | total %.2f us" % (
        d[:, 0].mean(), d[:, 1].mean(), "Cholesky + inverse of the 64 x 64 tile in LDS" if post == 1 else "Q_kk^T G", d[:, 2].mean(),
        d[:, 3].mean(), (b[:, 4] - b[:, 0]).mean() / 100.0))
    z = b[b[:, 7] == 0]
    if len(z):
        dz = np.diff(z[:, :5], axis=1) / 100.0
        print("   tasks without products (%d): %s total %.2f" % (len(z), np.round(dz.mean(0), 2).tolist(), (z[:, 4] - z[:, 0]).mean() / 100.0))
