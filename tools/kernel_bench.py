"""Per-kernel-class launch time / algorithmic GB/s of a workload (dotmi_bench_kernel): python tools/kernel_bench.py <workload> [steps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd import lib as dl
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
name = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sc, ep, n = load_workload(name)
ts = DOTTimeStepper(sc, ep, n)
for _ in range(nsteps):
    ts.solve(1)
L = dl.load()
print(f"{name}: nV {sc.V_rest.shape[0]} nT {sc.T.shape[0]} parts {n} PATCH_ELEMS={os.environ.get('DOTMI_PATCH_ELEMS','')}")
for kind, kn in enumerate(dl.BENCH_KERNELS):
    ms, nb = C.c_double(), C.c_int64()
    rc = L.dotmi_bench_kernel(ts._h, kind, 30, C.byref(ms), C.byref(nb))
    if rc: print(kn, "rc", rc); continue
    print(f"  {kn:18s} {1e3*ms.value:9.2f} us  {nb.value/1e6:10.2f} MB  {nb.value/ms.value/1e6:8.1f} GB/s  frac {nb.value/ms.value/1e6/8000:.3f}")
