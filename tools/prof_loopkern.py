"""Stage times inside the loop's vector kernels (library built with -DK_PROFILE: tools/prof_loopkern.sh): the gather and
merge_early as the device loop launches them, from wall-clock stamps of thread 0 of every workgroup."""
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
L.dotmi_debug_kprof.argtypes = [ctypes.POINTER(ctypes.c_longlong)]
names = {14: ("gather_early", 0, ["start -> loop state read, index + operand loads requested", "partial ranges there (1st round trip)",
                                  "partials summed (2nd round trip)", "stores + statistics", "block reduce -> end"]),
         12: ("merge_early", 1, ["start -> loop state read, list head + operands requested", "list walked, partials summed",
                                 "history terms + stores", "block reduce -> end"])}
for kind, (name, slot, stages) in names.items():
    rc = L.dotmi_bench_kernel(ts._h, kind, 20, ctypes.byref(ms), ctypes.byref(nb))
    buf = (ctypes.c_longlong * (4 * 256 * 8))()
    assert L.dotmi_debug_kprof(buf) == 0
    a = np.frombuffer(buf, dtype=np.int64).reshape(4, 256, 8)[slot]
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    ns = len(stages) + 1
    T = (a[:, :ns] - t0) / 100.0
    print("%s: rc %d, %.2f us per launch back to back; %d workgroups, last one ends at %.2f us after the first starts" % (
        name, rc, 1e3 * ms.value, len(a), T[:, ns - 1].max()))
    for k, nm in enumerate(stages):
        d = T[:, k + 1] - T[:, k]
        print("   %-62s p50 %.2f  p95 %.2f us" % (nm, np.median(d), np.percentile(d, 95)))
