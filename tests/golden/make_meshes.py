"""Convert the reference's tet-mesh data files to compact .npz fixtures.

Run HERE (needs /root/reference):  python tests/golden/make_meshes.py
Fixtures are DATA (vertex coordinates and tet indices exactly as parsed from the .msh files the
reference ships under input/tetMeshes); nothing from the reference's source is stored.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from dot_amd.scene import read_tet_msh  # noqa: E402

REF = os.environ.get("DOT_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(__file__), "meshes")
os.makedirs(OUT, exist_ok=True)
for name in ("bunny5K", "bar17K", "horse7K", "monkey18K", "kingkong18K"):
    V, T = read_tet_msh(os.path.join(REF, "input", "tetMeshes", name + ".msh"))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), V=V, T=T)
    print(name, V.shape, T.shape)
