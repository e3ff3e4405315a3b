"""Per-workgroup timeline of one back-solve launch (library built with -DBS_PROFILE: tools/prof_backsolve.sh)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
from dot_amd import lib as dl
sc, ep, n = load_workload(sys.argv[1] if len(sys.argv) > 1 else "bar17K_twist")
ts = DOTTimeStepper(sc, ep, n)
r = np.random.default_rng(0).standard_normal(sc.x0.shape) * (1 - sc.fixed[:, None])
for _ in range(5):
    ts.applyPrecond(r)
L = dl.load()
N = 8192
buf = (ctypes.c_longlong * (5 * N))()
L.dotmi_debug_bs_prof.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
assert L.dotmi_debug_bs_prof(buf, N) == 0
a = np.frombuffer(buf, dtype=np.int64).reshape(N, 5)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0   # 100 MHz -> us
print("workgroups", len(a), "span %.1f us" % en.max())
for lo, hi in [(0, 256), (256, 512), (512, 1024), (1024, 1536), (1536, 2560), (2560, 5120)]:
    m = (a[:, 2] > lo) & (a[:, 2] <= hi) & (a[:, 3] > 0)
    if m.any():
        print("len (%4d,%4d]: %4d tiles  start %.1f..%.1f  duration mean %.1f max %.1f  end max %.1f" % (
            lo, hi, m.sum(), st[m].min(), st[m].max(), (en - st)[m].mean(), (en - st)[m].max(), en[m].max()))
# busy workgroups over time
grid = np.arange(0, en.max() + 1, 2.0)
busy = [(int(((st <= t) & (en > t)).sum())) for t in grid]
print("busy workgroups every 2 us:", busy)
xcc = (a[:, 4] >> 32) & 15
print("tiles per XCC:", np.bincount(xcc.astype(int), minlength=8).tolist())
byx = [en[xcc == x].max() for x in range(8) if (xcc == x).any()]
print("last end per XCC:", [round(float(v), 1) for v in byx])
order = np.argsort(en)[-10:]
pk = a[:, 3] < 0
if pk.any():
    print("packs of four small tiles: %d  start %.1f..%.1f  duration (wavefront 0) mean %.1f max %.1f  end max %.1f" % (pk.sum(), st[pk].min(), st[pk].max(), (en - st)[pk].mean(), (en - st)[pk].max(), en[pk].max()))
print("last 10 to finish: (block, len, rows, start, end)", [(int(i), int(a[i, 2]), int(a[i, 3]), round(float(st[i]), 1), round(float(en[i]), 1)) for i in order])
