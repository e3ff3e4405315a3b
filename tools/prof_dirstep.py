"""Per-workgroup timeline of one speculative unit-step launch (library built with -DDS_PROFILE: tools/prof_dirstep.sh)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
from dot_amd import lib as dl
sc, ep, n = load_workload(sys.argv[1] if len(sys.argv) > 1 else "bar17K_twist")
ts = DOTTimeStepper(sc, ep, n)
for _ in range(3):
    x = ts.getResult(); idx, pos = sc.scripter.step(x, sc.cfg.dt); ts.setDirichlet(idx, pos); ts.step()
L = dl.load()
ms, nb = ctypes.c_double(), ctypes.c_int64()
for kind, name in ((11, "spmv_zp"), (13, "elem_step"), (15, "dirstep")):
    rc = L.dotmi_bench_kernel(ts._h, kind, 20, ctypes.byref(ms), ctypes.byref(nb))
    print("%-10s rc %d  %.2f us per launch back to back" % (name, rc, 1e3 * ms.value))
N = 4096
buf = (ctypes.c_longlong * (3 * N))()
L.dotmi_debug_ds_prof.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
assert L.dotmi_debug_ds_prof(buf, N) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(N, 3)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
S, E, P = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, a[:, 2]
print("workgroups %d, span %.1f us" % (len(a), E.max()))
for k, nm in enumerate(["direction rows (spmv_zp_body)", "patches (elem_patch_body, SPEC)", "trial point + inertia"]):
    m = P == k
    if m.any():
        print("  %-34s n %4d  start p50 %.2f max %.2f | end p50 %.2f p95 %.2f max %.2f | lifetime p50 %.2f p95 %.2f us" % (
            nm, m.sum(), np.median(S[m]), S[m].max(), np.median(E[m]), np.percentile(E[m], 95), E[m].max(),
            np.median((E - S)[m]), np.percentile((E - S)[m], 95)))
grid = np.arange(0, E.max() + 1, 2.0)
print("busy workgroups every 2 us:", [int(((S <= t) & (E > t)).sum()) for t in grid])

if hasattr(L, "dotmi_debug_ds_ep_prof"):
    N = 8192
    buf = (ctypes.c_longlong * (6 * N))()
    L.dotmi_debug_ds_ep_prof.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    assert L.dotmi_debug_ds_ep_prof(buf, N) == 0
    b = np.frombuffer(buf, dtype=np.int64).reshape(N, 6)
    b = b[(b[:, 0] > 0) & (b[:, 4] > b[:, 0])]
    T = (b[:, :5] - t0) / 100.0
    print("phases of the patches' workgroups (%d):" % len(b))
    for k, nm in enumerate(["start -> operands + positions in LDS", "element work + runs in LDS", "run sums + partial stores issued", "energy reduce -> end"]):
        d = T[:, k + 1] - T[:, k]
        print("  %-40s mean %.2f  p50 %.2f  p95 %.2f us" % (nm, d.mean(), np.median(d), np.percentile(d, 95)))

if hasattr(L, "dotmi_debug_ds_sp_prof"):
    N = 4096
    buf = (ctypes.c_longlong * (4 * N))()
    L.dotmi_debug_ds_sp_prof.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    assert L.dotmi_debug_ds_sp_prof(buf, N) == 0
    c = np.frombuffer(buf, dtype=np.int64).reshape(N, 4)
    c = c[c[:, 0] > 0]
    T = (c[:, :3] - t0) / 100.0
    print("speculative prologue: requests issued at p50 %.2f us; ids there + positions requested + short lists in LDS at %.2f; "
          "delta + barrier done at %.2f" % tuple(np.median(T[:, k]) for k in range(3)))

if hasattr(L, "dotmi_debug_ds_sv_prof"):
    N = 256
    buf = (ctypes.c_longlong * (6 * N))()
    L.dotmi_debug_ds_sv_prof.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    assert L.dotmi_debug_ds_sv_prof(buf, N) == 0
    c = np.frombuffer(buf, dtype=np.int64).reshape(N, 6)
    c = c[c[:, 0] > 0]
    T = (c[:, :5] - t0) / 100.0
    print("direction rows, wave 0 (p50, us since the launch's first workgroup): prologue requested %.2f | row ranges + own operands "
          "there, column loop starts %.2f | column loop done %.2f | delta + barrier %.2f | rows stored %.2f" % tuple(np.median(T[:, k]) for k in range(5)))
