"""Print the nested-dissection layout (host-only, no GPU) for a workload."""
import sys
import numpy as np
sys.path.insert(0, ".")
from dot_amd.workloads import load_workload
from dot_amd.sharding import plan_layout

name = sys.argv[1] if len(sys.argv) > 1 else "bar17K_twist"
levels = int(sys.argv[2]) if len(sys.argv) > 2 else -1
msplit = int(sys.argv[3]) if len(sys.argv) > 3 else -1
sc, ep, nparts = load_workload(name)
nodes, nmax, pos, verts = plan_layout(sc.V_rest, sc.T, ep, nparts, levels=levels, min_split=msplit)
print("nmax", nmax, "max part dofs", max(3 * v.size for v in verts))
for i, n in enumerate(nodes):
    print(i, dict(zip("off size a c offS sizeS".split(), [int(x) for x in n])))
for i, n in enumerate(nodes):
    ro, sz = (n[0], n[1]) if n[2] < 0 else (n[4], n[5])
    used = [int(((p >= ro) & (p < ro + sz)).sum() * 3) for p in pos]
    print("region", i, "rows", int(ro), int(sz), "used min/mean/max", min(used), int(np.mean(used)), max(used))
