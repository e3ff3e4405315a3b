import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
sc, ep, n = load_workload(sys.argv[1] if len(sys.argv) > 1 else "bar17K_twist")
ts = DOTTimeStepper(sc, ep, n)
ms, nb = ts.benchPrecond(200 if sc.T.shape[0] < 500000 else 40)
print("back-solve (right-hand-side gather + kernel) %.4f ms  %.1f GB/s algorithmic bytes %d per launch" % (ms, nb / ms / 1e6, nb))
