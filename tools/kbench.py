"""the loop's kernel forms launched back to back on the resident state of a workload after a few steps (dotmi_bench_kernel):
   python tools/kbench.py [workload] [kinds ...]      (default kinds: spmv_zp merge_early elem_step gather_early dirstep)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
from dot_amd import lib as dl
wl = sys.argv[1] if len(sys.argv) > 1 else "bar17K_twist"
kinds = sys.argv[2:] or ["spmv_zp", "merge_early", "elem_step", "gather_early", "dirstep"]
sc, ep, n = load_workload(wl)
ts = DOTTimeStepper(sc, ep, n)
for _ in range(3):
    x = ts.getResult(); idx, pos = sc.scripter.step(x, sc.cfg.dt); ts.setDirichlet(idx, pos); st = ts.step()
L = dl.load()
ms, nb = ctypes.c_double(), ctypes.c_int64()
out = []
for k in kinds:
    rc = L.dotmi_bench_kernel(ts._h, dl.BENCH_KERNELS.index(k), 50, ctypes.byref(ms), ctypes.byref(nb))
    out.append("%s %.2f us" % (k, 1e3 * ms.value) if rc == 0 else "%s rc %d" % (k, rc))
print(wl, "iters", st.iters, "|", " | ".join(out))
