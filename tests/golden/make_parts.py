"""Element partitions from the reference's vendored METIS 5.1.0 (oracle/_ref/metis_part, built by
`make -C oracle ref` from /root/reference/SuiteSparse/metis-5.1.0).  Run HERE:
    python tests/golden/make_parts.py
Fixtures are data: one small integer per tet."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from dot_amd.scene import load_mesh_npz  # noqa: E402
from tests import oracle_py as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
os.makedirs(os.path.join(HERE, "parts"), exist_ok=True)
for mesh, nparts in (("bunny5K", 8), ("bunny5K", 6), ("bar17K", 32), ("bar17K", 6), ("horse7K", 8),
                     ("monkey18K", 64), ("kingkong18K", 18), ("monkey18K", 18)):   # 18 = nV / 1024 + 1 (`DOT -1 1024`)
    V, T = load_mesh_npz(os.path.join(HERE, "meshes", mesh + ".npz"))
    ep = O.metis_partition(T, V.shape[0], nparts)
    assert ep.min() == 0 and ep.max() == nparts - 1
    np.save(os.path.join(HERE, "parts", f"{mesh}_{nparts}.npy"), ep.astype(np.uint8))
    print(mesh, nparts, np.bincount(ep))

# vertex partitions (METIS::partMesh_nodes, METIS.hpp:161-193) for LBFGS-JH
for mesh, nparts in (("bunny5K", 8),):
    V, T = load_mesh_npz(os.path.join(HERE, "meshes", mesh + ".npz"))
    vp = O.metis_partition(T, V.shape[0], nparts, nodal=True)
    assert vp.size == V.shape[0] and vp.min() == 0 and vp.max() == nparts - 1
    np.save(os.path.join(HERE, "parts", f"{mesh}_{nparts}_nodes.npy"), vp.astype(np.uint8))
    print(mesh, nparts, "nodes", np.bincount(vp))
