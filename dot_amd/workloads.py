"""Named workloads (BASELINE.json `configs`) for bench.py, the tools and the tests.  Each mirrors
one of the reference's input scripts with the overrides BASELINE.json names; the script *values* are restated here so
nothing needs /root/reference at run time.  The meshes and the reference's METIS partitions are data fixtures under
tests/golden/ (DOT_MESH_DIR / DOT_PART_DIR point elsewhere); the hot path itself (libdotmi, timestepper.py) reads no
fixture -- a caller with its own mesh never imports this module."""
from __future__ import annotations

import os

import numpy as np

from dot_amd.scene import Config, Scene, build_scene, load_mesh_npz, partition_rcb, synthetic_bar

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # dot_amd/ -> repo root
MESH_DIR = os.environ.get("DOT_MESH_DIR") or os.path.join(_ROOT, "tests", "golden", "meshes")
PART_DIR = os.environ.get("DOT_PART_DIR") or os.path.join(_ROOT, "tests", "golden", "parts")

# name -> (mesh, Config kwargs, nParts)
WORKLOADS = {
    # input/bunny5K_LTSS_DOT.txt, FixedCoRot, 8 subdomains (BASELINE.json configs[0])
    "bunny5K_LTSS": ("bunny5K", dict(energy="FCR", size=1.0, duration=5.0, dt=0.025, rho=1000.0, YM=1e5,
                                     PR=0.4, script="twistnsns"), 8),
    # input/bar17K_twist_DOT.txt, StableNH, 32 subdomains (configs[1], the bench workload)
    "bar17K_twist": ("bar17K", dict(energy="SNH", size=1.0, duration=5.0, dt=0.025, rho=1000.0, YM=1e5,
                                    PR=0.4, script="twist"), 32),
    # input/tb1_horse_scalab/horse7K_stretch_DOT.txt (stand-in for the missing horse136K), FCR, 8
    "horse7K_stretch": ("horse7K", dict(energy="FCR", size=1.0, duration=10.0, dt=0.025, rho=1000.0, YM=1e5,
                                        PR=0.4, script="stretch"), 8),
    # input/tb2_monkey_mat_dt/monkey18K_TSS_DOT_E4e5.txt + time 10 0.04, StableNH, 64 subdomains
    "monkey18K_stiff": ("monkey18K", dict(energy="SNH", size=1.0, duration=10.0, dt=0.04, rho=1000.0, YM=4e5,
                                          PR=0.4, script="twistnsns_old", rot_deg=40.0, rot_axis=(0.0, 1.0, 0.0),
                                          handle_ratio=0.02), 64),
    # input/tb5_ablation/kingkong18K_SS_DOT-1K.txt: `timeStepper DOT -1 1024` -> nV / 1024 + 1 = 18 subdomains
    # (main.cpp:792-798)
    "kingkong18K_SS_1K": ("kingkong18K", dict(energy="FCR", size=1.0, duration=10.0, dt=0.025, rho=1000.0, YM=1e5,
                                              PR=0.4, script="stretchnsquash", block_size=1024), -1),
    # input/tb5_ablation/monkey18K_TSS_DOT-1K.txt
    "monkey18K_TSS_1K": ("monkey18K", dict(energy="FCR", size=1.0, duration=10.0, dt=0.025, rho=1000.0, YM=1e5,
                                           PR=0.4, script="twistnsns_old", rot_deg=40.0, rot_axis=(0.0, 1.0, 0.0),
                                           handle_ratio=0.02, block_size=1024), -1),
}


def load_workload(name: str, nparts: int | None = None):
    """-> (Scene, epart, nparts).  Partition = committed METIS fixture when it exists for this (mesh, nparts), else the
    library's own partitioner (dotmi_partition); the synthetic bars keep the seedless coordinate bisection SURVEY.md
    section 8(d) M5 specifies for them.  Block-size scripts (`DOT -1 <b>`) get nV / b + 1 subdomains."""
    if name.startswith("synbar"):
        # synbar:<nx>x<ny>x<nz>:<nparts>  e.g. the 1M-tet bar = synbar:140x35x35:256
        _, dims, npart_s = name.split(":")
        nx, ny, nz = (int(t) for t in dims.split("x"))
        V, T = synthetic_bar(nx, ny, nz)
        cfg = Config(energy="SNH", size=1.0, duration=5.0, dt=0.025, rho=1000.0, YM=1e5, PR=0.4, script="twist")
        sc = build_scene(cfg, V, T)
        np_ = int(npart_s) if nparts is None else nparts
        return sc, partition_rcb(sc.V_rest, sc.T, np_), np_
    # <workload>@r<k>:<nparts>  = the workload's mesh red-refined k times (x8 tets per level), e.g. horse7K_stretch@r1:64,
    # the scale stand-in for the horse meshes the reference checkout lacks
    refine = 0
    if "@r" in name:
        name, tail = name.split("@r")
        lev, _, np_s = tail.partition(":")
        refine = int(lev)
        if nparts is None and np_s:
            nparts = int(np_s)
    mesh, kw, np_default = WORKLOADS[name]
    V, T = load_mesh_npz(os.path.join(MESH_DIR, mesh + ".npz"))
    for _ in range(refine):
        from dot_amd.scene import refine_red
        V, T = refine_red(V, T)
    if refine:
        mesh = f"{mesh}_r{refine}"
    cfg = Config(**kw)
    if cfg.block_size > 0 and nparts is None:
        np_default = V.shape[0] // cfg.block_size + 1      # main.cpp:792-798
    np_ = np_default if nparts is None else nparts
    cfg.partition_amt = np_
    sc = build_scene(cfg, V, T)
    f = os.path.join(PART_DIR, f"{mesh}_{np_}.npy")
    if os.path.exists(f):
        epart = np.load(f).astype(np.int32)
    else:
        from dot_amd.scene import partition_dual
        epart = partition_dual(sc.V_rest, sc.T, np_)
    return sc, epart, np_
