"""Host-side logic around the hot path: script parsing, scene construction, scripted Dirichlet
motion, partitioners, workload table.  No GPU, no oracle."""
import math
import os
import subprocess

import numpy as np
import pytest

from dot_amd import scene
from tests.workloads import WORKLOADS, load_workload
from dot_amd.sharding import owned_elements, part_scalar_sizes, plan_shards, vertex_slice

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parse_script_tokens(tmp_path):
    # same token set as the reference's input/bunny5K_LTSS_DOT.txt (values restated)
    p = tmp_path / "s.txt"
    p.write_text("energy FCR\ntimeStepper DOT 6\ninexactSolve 0\nwarmStart 2\nresolution 1000\nsize 1\n"
                 "time 5 0.025\ndensity 1000\nstiffness 100000 0.4\nscript twistnsns\n"
                 "shape input input/tetMeshes/bunny5K.msh\nview orthographic\nzoom 1\n"
                 "tol 2\n1e-4\n1e-5\nunknownToken 3\nhandleRatio 0.02\nturnOffGravity\n")
    cfg = scene.parse_script(str(p))
    assert cfg.energy == "FCR" and cfg.partition_amt == 6 and cfg.dt == 0.025 and cfg.duration == 5
    assert cfg.rho == 1000 and cfg.YM == 1e5 and cfg.PR == 0.4 and cfg.script == "twistnsns"
    assert cfg.shape_path.endswith("bunny5K.msh") and cfg.tol == [1e-4, 1e-5]
    assert cfg.handle_ratio == 0.02 and not cfg.with_gravity
    p.write_text("timeStepper DOT 1\n")
    assert scene.parse_script(str(p)).partition_amt == 4          # "<2 -> 4"
    p.write_text("timeStepper DOT -1 1024\n")
    assert scene.parse_script(str(p)).block_size == 1024


def test_msh_reader_roundtrip(tmp_path):
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]], dtype=float)
    T = np.array([[0, 1, 2, 3], [1, 2, 3, 4]])
    f = tmp_path / "m.msh"
    with open(f, "w") as fh:
        fh.write("$MeshFormat\n4 0 8\n$EndMeshFormat\n$Entities\n0 0 0 1\n$EndEntities\n$Nodes\n1 5\n0 1 0 5\n")
        for i, v in enumerate(V):
            fh.write(f"{i+1} {float(v[0])!r} {float(v[1])!r} {float(v[2])!r}\n")
        fh.write("$EndNodes\n$Elements\n1 2\n0 1 4 2\n")
        for i, t in enumerate(T):
            fh.write(f"{i+1} {t[0]+1} {t[1]+1} {t[2]+1} {t[3]+1}\n")
        fh.write("$EndElements\n")
    V2, T2 = scene.read_tet_msh(str(f))
    assert np.array_equal(V, V2) and np.array_equal(T, T2)


def test_normalize_and_handles():
    rng = np.random.default_rng(0)
    V = rng.random((200, 3)) * [4, 1, 2] + 3
    Vn = scene.normalize(V, 1.0)
    assert np.allclose(Vn.min(axis=0), 0) and np.isclose((Vn.max(axis=0)).max(), 1.0)
    g0, g1 = scene.find_border_verts(Vn, 0.05)
    assert (Vn[g0, 0] < 0.05).all() and (Vn[g1, 0] > 0.95).all() and not set(g0) & set(g1)
    R = scene.angle_axis_matrix(0.3, (1, 0, 0))
    assert np.allclose(R @ R.T, np.eye(3)) and R[0, 0] == 1.0
    assert np.allclose(R[1:, 1:], [[math.cos(0.3), -math.sin(0.3)], [math.sin(0.3), math.cos(0.3)]])


def test_twist_script_rotates_handles_rigidly():
    V, T = scene.synthetic_bar(8, 2, 2)
    cfg = scene.Config(script="twist", dt=0.025, handle_ratio=0.01)
    sc = scene.build_scene(cfg, V, T)
    fixed = sc.fixed.astype(bool)
    assert fixed.sum() == 2 * 9
    x = sc.x0.copy()
    idx, pos = sc.scripter.step(x, cfg.dt)
    assert set(idx) == set(np.nonzero(fixed)[0])
    c = sc.scripter.rot_center
    # rigid rotation about the x axis through the bbox centre: distances to the axis preserved
    r0 = np.linalg.norm((x[idx] - c)[:, 1:], axis=1)
    r1 = np.linalg.norm((pos - c)[:, 1:], axis=1)
    assert np.allclose(r0, r1, atol=1e-14) and np.allclose(pos[:, 0], x[idx, 0])
    # the two ends turn in opposite directions by 0.1*pi*dt (AnimScripter.cpp:138-155)
    ang = np.arctan2((pos - c)[:, 2], (pos - c)[:, 1]) - np.arctan2((x[idx] - c)[:, 2], (x[idx] - c)[:, 1])
    ang = (ang + math.pi) % (2 * math.pi) - math.pi
    off = r0 > 1e-9                                         # vertices on the axis do not move
    left = x[idx, 0] < 0.5
    assert np.allclose(np.abs(ang[off]), 0.1 * math.pi * cfg.dt, atol=1e-12)
    assert (np.sign(ang[off & left]) == -np.sign(ang[off & ~left][0])).all()


def test_twistnsns_velocity_flips_at_turning_points():
    V, T = scene.synthetic_bar(8, 2, 2)
    cfg = scene.Config(script="twistnsns", dt=0.025)
    sc = scene.build_scene(cfg, V, T)
    x = sc.x0.copy()
    tv = sc.scripter.turn_vert
    xs = []
    for _ in range(80):
        idx, pos = sc.scripter.step(x, cfg.dt)
        x[idx] = pos
        xs.append(x[tv, 0])
    xs = np.array(xs)
    d = np.diff(xs)
    assert (np.sign(d[:-1]) != np.sign(d[1:])).sum() >= 1            # it turned around at least once
    assert xs.max() <= sc.x0[tv, 0] + 0.4 + 1.2 * cfg.dt + 1e-12
    assert np.allclose(np.abs(d), 1.2 * cfg.dt)                        # |v_x| = 1.2 (AnimScripter.cpp:179-217)


def test_unsupported_script_is_rejected():
    V, T = scene.synthetic_bar(2, 1, 1)
    with pytest.raises(ValueError):
        scene.build_scene(scene.Config(script="bend"), V, T)          # 2-D only script


def test_rubber_band_pull_releases_the_waist():
    V, T = scene.synthetic_bar(2, 40, 2, lx=0.2, ly=4.0, lz=0.2)
    sc = scene.build_scene(scene.Config(script="rubberBandPull", dt=0.025), V, T)
    nfix0 = int(sc.fixed.sum())
    x = sc.x0.copy()
    released_at = None
    for k in range(100):
        idx, pos = sc.scripter.step(x, 0.025)
        x[idx] = pos
        if sc.scripter.changed:
            released_at = k
            break
    # the waist travels 5 units at 2.5 units/s: released at t = 2 s = step 80 (AnimScripter.cpp:245-250, :404)
    assert released_at == 80 and 0 < int(sc.fixed.sum()) < nfix0
    idx, pos = sc.scripter.step(x, 0.025)
    assert np.array_equal(pos, x[idx])                                    # everything stopped


def test_synthetic_bar_is_positively_oriented_and_conforming():
    V, T = scene.synthetic_bar(5, 3, 2)
    assert T.shape == (5 * 3 * 2 * 6, 4) and V.shape == (6 * 4 * 3, 3)
    d = V[T[:, 1:]] - V[T[:, :1]]
    vol = np.einsum("ij,ij->i", d[:, 0], np.cross(d[:, 1], d[:, 2])) / 6
    assert (vol > 0).all() and np.isclose(vol.sum(), 4.0)
    # every interior face is shared by exactly two tets
    faces = np.sort(np.concatenate([T[:, [0, 1, 2]], T[:, [0, 1, 3]], T[:, [0, 2, 3]], T[:, [1, 2, 3]]]), axis=1)
    _, cnt = np.unique(faces, axis=0, return_counts=True)
    assert set(cnt) <= {1, 2}
    Vj, _ = scene.synthetic_bar(5, 3, 2, jitter=0.1)
    assert np.array_equal(Vj, scene.synthetic_bar(5, 3, 2, jitter=0.1)[0])   # seeded


def test_rcb_partition_is_balanced_and_complete():
    V, T = scene.synthetic_bar(16, 4, 4)
    for nparts in (2, 5, 8):
        ep = scene.partition_rcb(V, T, nparts)
        cnt = np.bincount(ep, minlength=nparts)
        assert cnt.min() > 0 and cnt.max() - cnt.min() <= max(2, 0.02 * len(T))


def test_workload_table_and_fixtures():
    for name, (mesh, kw, nparts) in WORKLOADS.items():
        sc, ep, n = load_workload(name)
        if nparts < 0:                                   # block-size mode: nV / block + 1 (main.cpp:792-798)
            nparts = sc.V_rest.shape[0] // sc.cfg.block_size + 1
        assert n == nparts and ep.shape == (sc.T.shape[0],) and ep.min() == 0 and ep.max() == nparts - 1
        assert sc.fixed.sum() > 0 and sc.V_rest.max() <= 1.0 + 1e-12
        d = sc.V_rest[sc.T[:, 1:]] - sc.V_rest[sc.T[:, :1]]
        assert (np.einsum("ij,ij->i", d[:, 0], np.cross(d[:, 1], d[:, 2])) > 0).all()
    sc, ep, n = load_workload("bunny5K_LTSS")
    assert sc.V_rest.shape == (4670, 3) and sc.T.shape == (19379, 4) and sc.fixed.sum() == 23
    sc, ep, n = load_workload("synbar:4x2x2:3")
    assert n == 3 and ep.max() == 2


def test_shard_plan_properties():
    sc, ep, nparts = load_workload("bar17K_twist")
    ps = part_scalar_sizes(sc.T, ep, nparts)
    assert ps.max() == 3 * 716                                   # "max 716 local verts/part" 
    for world in (1, 2, 4, 8, 32, 40):
        first = plan_shards(ps, world)
        assert first[0] == 0 and first[-1] == nparts and all(a <= b for a, b in zip(first, first[1:]))
        owned = [owned_elements(ep, first, r) for r in range(world)]
        assert sum(len(o) for o in owned) == len(ep)
        if world <= 8:
            cost = [float((ps[first[r]:first[r + 1]].astype(float) ** 2).sum()) for r in range(world)]
            assert max(cost) <= 1.25 * (sum(cost) / world)
    sl = [vertex_slice(101, r, 4) for r in range(4)]
    assert sl[0][0] == 0 and sl[-1][1] == 101 and all(a[1] == b[0] for a, b in zip(sl, sl[1:]))


def _write_msh(path, V, T):
    with open(path, "w") as fh:
        fh.write("$MeshFormat\n4 0 8\n$EndMeshFormat\n$Entities\n0 0 0 1\n$EndEntities\n$Nodes\n1 %d\n0 1 0 %d\n" % (len(V), len(V)))
        for i, v in enumerate(V):
            fh.write("%d %.17g %.17g %.17g\n" % (i + 1, v[0], v[1], v[2]))
        fh.write("$EndNodes\n$Elements\n1 %d\n0 1 4 %d\n" % (len(T), len(T)))
        for i, t in enumerate(T):
            fh.write("%d %d %d %d %d\n" % (i + 1, t[0] + 1, t[1] + 1, t[2] + 1, t[3] + 1))
        fh.write("$EndElements\n")


@pytest.mark.parametrize("script", ["twist", "twistnsns", "stretch", "twistnstretch", "squash", "stretchnsquash"])
def test_cpp_scene_layer_matches_python(tmp_path, script):
    """dot_amd/host/Scene.hpp (C++ Config parser, .msh reader, normalise, handles, AnimScripter) against
    dot_amd/scene.py on the same script + mesh: identical handle sets and scripted positions."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "dot_amd", "dot_hip")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "dot_amd", "host")])
    V, T = scene.synthetic_bar(10, 3, 2, jitter=0.05)
    V = V * 1.7 + np.array([3.0, -1.0, 0.5])
    _write_msh(tmp_path / "bar.msh", V, T)
    (tmp_path / "s.txt").write_text(f"energy SNH\ntimeStepper DOT 3\nsize 1\ntime 1 0.02\ndensity 1000\n"
                                    f"stiffness 100000 0.4\nscript {script}\nshape input bar.msh\nhandleRatio 0.05\n")
    nmove = 40
    out = subprocess.check_output([exe, "100", str(tmp_path / "s.txt"), "--mesh-root", str(tmp_path),
                                   "--dump-scene", str(nmove)]).decode().splitlines()
    sc = scene.load_scene(str(tmp_path / "s.txt"), mesh_dir=str(tmp_path))   # no npz there -> reads the .msh
    head = out[0].split()
    assert int(head[2]) == sc.V_rest.shape[0] and int(head[4]) == sc.T.shape[0] and int(head[6]) == int(sc.fixed.sum())
    x = sc.x0.copy()
    for k in range(nmove):
        idx, pos = sc.scripter.step(x, sc.cfg.dt)
        x[idx] = pos
        tok = out[1 + k].split()
        assert int(tok[3]) == len(idx)
        assert np.allclose([float(tok[5]), float(tok[6]), float(tok[7])], pos.sum(axis=0), rtol=1e-14, atol=1e-13)


@pytest.mark.parametrize("workload,levels", [("bunny5K_LTSS", 2), ("bar17K_twist", 2), ("bar17K_twist", 1),
                                             ("horse7K_stretch", 3)])
def test_nested_dissection_layout_is_a_valid_separator_ordering(workload, levels):
    """dotmi_plan_layout (host only): the order the dense subdomain blocks are laid out in.  What the factor
    kernels and the back-solve rely on: every local vertex has its own 3 padded slots inside its region; the
    tree's ranges nest; and no mesh edge joins the A and C sides of any dissection node -- i.e. for every
    coupled pair the earlier position is not left of the first column the later one's rows start at."""
    from dot_amd.sharding import plan_layout
    sc, ep, nparts = load_workload(workload)
    nodes, nmax, pos, verts = plan_layout(sc.V_rest, sc.T, ep, nparts, levels=levels, min_split=256)
    assert nmax % 64 == 0 and nmax == nodes[0][1] and nodes[0][0] == 0
    first_col = np.zeros(nmax, dtype=np.int64)      # first column a row at this position can be non-zero in
    region_of = -np.ones(nmax, dtype=np.int64)
    for i, (off, size, a, c, offS, sizeS) in enumerate(nodes):
        if a < 0:
            assert c < 0 and size % 64 == 0
            first_col[off:off + size] = off
            region_of[off:off + size] = i
        else:
            A, C = nodes[a], nodes[c]
            assert A[0] == off and C[0] == off + A[1] and offS == C[0] + C[1] and size == A[1] + C[1] + sizeS
            first_col[offS:offS + sizeS] = off
            region_of[offS:offS + sizeS] = i
    assert (region_of >= 0).all()
    # edges of the mesh = couplings of the Hessian
    E = np.unique(np.sort(np.concatenate([sc.T[:, [i, j]] for i in range(4) for j in range(i + 1, 4)]), axis=1), axis=0)
    for p, (ps, vs) in enumerate(zip(pos, verts)):
        assert ps.size == vs.size and ps.min() >= 0 and ps.max() + 3 <= nmax
        assert (np.diff(np.sort(ps)) >= 3).all()            # three slots per vertex, no overlap
        assert (region_of[ps] == region_of[ps + 2]).all()   # ... inside one region
        where = -np.ones(sc.V_rest.shape[0], dtype=np.int64)
        where[vs] = ps
        pu, pv = where[E[:, 0]], where[E[:, 1]]
        both = (pu >= 0) & (pv >= 0)
        lo, hi = np.minimum(pu, pv)[both], np.maximum(pu, pv)[both]
        assert (lo >= first_col[hi]).all(), f"part {p}: an edge crosses a dissection"
    # the layout pays off: strictly fewer structural non-zeros than the dense triangle on the big workload
    if workload == "bar17K_twist":
        ps, vs = pos[0], verts[0]
        nnz = sum(int(((ps >= first_col[q]) & (ps <= q)).sum()) for q in ps)
        assert nnz < 0.7 * ps.size * (ps.size + 1) / 2


def test_node_ele_restart_and_partition_files_cpp_vs_python(tmp_path):
    """The remaining pieces of the reference's file layer, C++ (dot_amd/host) against Python (dot_amd/scene.py):
    the TetGen .node/.ele reader (IglUtils.cpp:751-793), the `restart <status>` token with the status<n> text format
    (Optimizer.cpp:126-177, :1096-1132), and the partition files label.obj / wire.poly of
    ADMMDDTimeStepper.cpp:375-442."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "dot_amd", "dot_hip")
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "dot_amd", "host")])
    V, T = scene.synthetic_bar(6, 2, 2, jitter=0.05)
    nV, nT = V.shape[0], T.shape[0]
    with open(tmp_path / "bar.node", "w") as f:                       # zero-based ids, as the reference expects
        f.write(f"{nV} 3 0 0\n" + "".join(f"{i} {p[0]!r} {p[1]!r} {p[2]!r}\n" for i, p in enumerate(V.tolist())))
    with open(tmp_path / "bar.ele", "w") as f:
        f.write(f"{nT} 4 0\n" + "".join(f"{i} {t[0]} {t[1]} {t[2]} {t[3]}\n" for i, t in enumerate(T.tolist())))
    V2, T2 = scene.load_tet_mesh(str(tmp_path / "bar"))
    assert np.array_equal(V2, V) and np.array_equal(T2, T)
    # a status file in the reference's format, referenced by the script
    rng = np.random.default_rng(5)
    xs, vs = rng.standard_normal((nV, 3)), rng.standard_normal((nV, 3))
    scene.write_status(str(tmp_path / "status7"), 7, xs, vs, xs - 0.1)
    t, xr, vr = scene.read_status(str(tmp_path / "status7"), nV)
    assert t == 7 and np.allclose(xr, xs, rtol=1e-6) and np.allclose(vr, vs, rtol=1e-6)   # %le keeps 7 digits
    (tmp_path / "s.txt").write_text("energy FCR\ntimeStepper DOT 3\nsize 1\ntime 1 0.02\ndensity 1000\n"
                                    "stiffness 100000 0.4\nscript stretch\nshape input bar\nhandleRatio 0.1\n"
                                    f"restart {tmp_path / 'status7'}\n")
    cfg = scene.parse_script(str(tmp_path / "s.txt"))
    assert cfg.restart and cfg.status_path.endswith("status7")
    ep = scene.partition_rcb(scene.normalize(V, 1.0, 0.0, (0, 1, 0)), T, 3)
    ep.astype(np.int32).tofile(tmp_path / "ep.i32")
    out = subprocess.check_output([exe, "100", str(tmp_path / "s.txt"), "--mesh-root", str(tmp_path), "--epart",
                                   str(tmp_path / "ep.i32"), "--out", str(tmp_path / "o"), "--dump-scene", "0"]
                                  ).decode().splitlines()
    head = out[0].split()
    assert int(head[2]) == nV and int(head[4]) == nT
    rs = [l for l in out if l.startswith("restart")][0].split()
    assert int(rs[2]) == 7 and np.isclose(float(rs[4]), xr.sum(), rtol=1e-12) and np.isclose(float(rs[6]), vr.sum(), rtol=1e-12)
    # label.obj: one line per surface triangle with the subdomain of its tet -- same multiset as Python's
    tris, tet = scene.surface_triangles(T)
    labels = sorted(int(l.split()[1]) for l in (tmp_path / "o" / "label.obj").read_text().splitlines())
    assert labels == sorted(ep[tet].tolist())
    wire = (tmp_path / "o" / "wire.poly").read_text().splitlines()
    npts = wire.index("POLYS") - 1
    assert wire[0] == "POINTS" and wire[-1] == "END" and npts == np.unique(tris).size
    assert len(wire) == 1 + npts + 1 + 3 * tris.shape[0] + 1


@pytest.mark.parametrize("workload", ["bar17K_twist", "bunny5K_LTSS", "horse7K_stretch"])
def test_scripter_tracking_equals_reading_positions_back(workload):
    """bench.py lets the scripter remember the Dirichlet positions it set (step(None, dt)) instead of reading all
    positions back each step: must give bit-identical moves, and is refused where it would be wrong."""
    sc, _, _ = load_workload(workload)
    sc2, _, _ = load_workload(workload)
    with pytest.raises(ValueError):
        sc2.scripter.step(None, sc2.cfg.dt)          # not tracking yet
    sc2.scripter.track(sc2.x0)
    x = sc.x0.copy()
    for _ in range(80):
        i1, p1 = sc.scripter.step(x, sc.cfg.dt)
        x[i1] = p1
        i2, p2 = sc2.scripter.step(None, sc2.cfg.dt)
        assert np.array_equal(i1, i2) and np.array_equal(p1, p2)


# ---- f2: the reference's output formats, read back by a restated reader of those formats ----------------------------
def _read_timer_tables(lines):
    """Timer::print blocks (Utils/Timer.hpp:58-68): '<n> activities:' then n lines '%10g s: <name>' and the Total"""
    import re
    tables, i = [], 0
    while i < len(lines):
        m = re.fullmatch(r"(\d+) activities:", lines[i])
        if not m:
            break
        n = int(m.group(1))
        rows = []
        for l in lines[i + 1:i + 2 + n]:
            mm = re.fullmatch(r"\s*(\S+) s: (\S+)", l)
            assert mm and len(l.split(" s: ")[0]) >= 10, l       # os.width(10), right aligned
            rows.append((mm.group(2), float(mm.group(1))))
        assert rows[-1][0] == "Total" and abs(rows[-1][1] - sum(v for _, v in rows[:-1])) <= 1e-5 * max(1.0, rows[-1][1])
        tables.append(rows[:-1])
        i += n + 2
    return tables, lines[i:]


def read_info_txt(path):
    """saveInfoForPresent (main.cpp:338-358)"""
    lines = open(path).read().splitlines()
    nV, nT = (int(t) for t in lines[0].split())
    f = lines[1].split()
    assert len(f) == 5 and f[2:] == ["0", "0", "0"]
    tables, rest = _read_timer_tables(lines[2:])
    assert rest == ["0 0"]
    return dict(nV=nV, nT=nT, iterNum=int(f[0]), innerIterAmt=int(f[1]), timer=tables[0], timer_step=tables[1],
                timer_temp3=tables[2])


def read_obj(path):
    """igl::writeOBJ(V, F) (libigl writeOBJ.cpp:100-120): 'v x y z' rows then 'f a b c' rows, 1-based"""
    V, F = [], []
    for l in open(path).read().splitlines():
        t = l.split(" ")
        if t[0] == "v":
            assert not F and len(t) == 4
            V.append([float(u) for u in t[1:]])
        else:
            assert t[0] == "f" and len(t) == 4
            F.append([int(u) - 1 for u in t[1:]])
    return np.array(V), np.array(F)


TIMER_STEP = ["matrixComputation", "matrixAssembly", "symbolicFactorization", "numericalFactorization", "backSolve",
              "lineSearch_other", "modifyGrad", "modifySearchDir", "updateHistory", "lineSearch_eVal",
              "fullyImplicit_eComp", "solve_extraComp", "compGrad", "CCD"]                      # main.cpp:867-880
TIMER_TEMP3 = ["init", "initPrimal", "initDual", "initWeights", "initCons", "subdSolve", "consSolve"]   # :882-888


@pytest.mark.parametrize("with_surface_section", [False, True])
def test_info_txt_and_surface_obj_formats(tmp_path, with_surface_section):
    """info.txt = two header lines + the three Timer tables with the reference's 14 + 7 slot names;
    <n>.obj = the re-indexed SURFACE mesh (not all tet vertices), 15 significant digits.  `dot_hip --dump-formats`
    writes both from the initial scene without a GPU."""
    exe = os.path.join(ROOT, "dot_amd", "dot_hip")
    if not os.path.exists(exe):
        pytest.skip("dot_hip not built")
    V, T = scene.synthetic_bar(5, 2, 2)
    rng = np.random.default_rng(9)
    V = V + 0.01 * rng.standard_normal(V.shape)
    nV, nT = V.shape[0], T.shape[0]
    SF = None
    msh = "$MeshFormat\n4 0 8\n$EndMeshFormat\n$Nodes\n1 %d\n0 3 0 %d\n" % (nV, nV)
    msh += "".join("%d %r %r %r\n" % (i + 1, p[0], p[1], p[2]) for i, p in enumerate(V.tolist()))
    msh += "$EndNodes\n$Elements\n1 %d\n0 3 4 %d\n" % (nT, nT)
    msh += "".join("%d %d %d %d %d\n" % (i + 1, t[0] + 1, t[1] + 1, t[2] + 1, t[3] + 1) for i, t in enumerate(T.tolist()))
    msh += "$EndElements\n"
    if with_surface_section:
        SF = scene.surface_triangles(T)[0][::-1].copy()          # any order the file chooses must be kept
        msh += "$Surface\n%d\n" % SF.shape[0] + "".join("%d %d %d\n" % (a + 1, b + 1, c + 1) for a, b, c in SF.tolist())
        msh += "$EndSurface\n"
    (tmp_path / "m.msh").write_text(msh)
    (tmp_path / "s.txt").write_text("energy FCR\ntimeStepper DOT 2\nsize 1\ntime 1 0.02\ndensity 1000\n"
                                    "stiffness 100000 0.4\nscript twist\nshape input m.msh\nhandleRatio 0.1\n")
    subprocess.check_call([exe, "100", str(tmp_path / "s.txt"), "--mesh-root", str(tmp_path), "--dump-formats",
                           str(tmp_path / "o")])
    info = read_info_txt(tmp_path / "o" / "info.txt")
    assert (info["nV"], info["nT"], info["iterNum"], info["innerIterAmt"]) == (nV, nT, 7, 123)
    assert info["timer"] == [("descent", 12.5)]
    assert [n for n, _ in info["timer_step"]] == TIMER_STEP and [n for n, _ in info["timer_temp3"]] == TIMER_TEMP3
    assert [v for _, v in info["timer_step"]] == [round(0.001 * (k + 1), 3) for k in range(14)]
    assert all(v == 0.0 for _, v in info["timer_temp3"])
    # surface mesh: the scene normalises the positions, the Python mirror does the same
    sc = scene.load_scene(str(tmp_path / "s.txt"), mesh_dir=str(tmp_path))
    s2t, Fs = scene.surface_mesh(T, SF)
    Vo, Fo = read_obj(tmp_path / "o" / "0.obj")
    assert Vo.shape == (s2t.size, 3) and s2t.size < nV             # interior vertices are not written
    assert np.array_equal(Fo, Fs)                                   # same triangles, same order, re-indexed
    assert np.abs(Vo - sc.x0[s2t]).max() <= 1e-14 * np.abs(sc.x0).max()
    # label.obj follows the same triangle order (one subdomain label per surface triangle)
    labels = (tmp_path / "o" / "label.obj").read_text().splitlines()
    assert len(labels) == Fs.shape[0]


# ---- the built-in partitioner (dotmi_partition, host only) ----------------------------------------------------------
@pytest.mark.parametrize("workload,nparts,metis_iface", [("bunny5K_LTSS", 8, 1031), ("bar17K_twist", 32, 8099),
                                                         ("horse7K_stretch", 8, 1197)])
def test_builtin_partitioner_quality_balance_and_determinism(workload, nparts, metis_iface):
    """dotmi_partition (recursive dual-graph bisection + FM refinement) stands in for METIS::partMesh on meshes without
    a fixture: every part non-empty and within 6 % of the mean size, deterministic, element ids untouched, and an
    interface size (sum over subdomains of vertices shared with another subdomain) within 20 % of the reference's
    METIS partition of the same mesh -- coordinate bisection is 20-70 % above it."""
    sc, ep_metis, n = load_workload(workload)
    assert n == nparts
    a = scene.partition_dual(sc.V_rest, sc.T, nparts)
    b = scene.partition_dual(sc.V_rest, sc.T, nparts)
    assert np.array_equal(a, b) and a.shape == (sc.T.shape[0],) and a.min() == 0 and a.max() == nparts - 1
    cnt = np.bincount(a, minlength=nparts)
    assert cnt.min() > 0 and np.abs(cnt - cnt.mean()).max() <= 0.06 * cnt.mean()

    def interface(e):
        vs = [np.unique(sc.T[e == p]) for p in range(nparts)]
        dup = np.zeros(sc.V_rest.shape[0], dtype=int)
        for v in vs:
            dup[v] += 1
        return int(sum((dup[v] > 1).sum() for v in vs))

    assert interface(ep_metis) == metis_iface                       # C2 sum of interface = 8 099
    own, rcb = interface(a), interface(scene.partition_rcb(sc.V_rest, sc.T, nparts))
    assert own <= 1.20 * metis_iface and own < 0.9 * rcb, (own, metis_iface, rcb)
    # degenerate requests
    one = scene.partition_dual(sc.V_rest, sc.T, 1)
    assert not one.any()
    many = scene.partition_dual(sc.V_rest[:], sc.T[:50], 50)
    assert sorted(many.tolist()) == list(range(50))


# ---- f2 pinned against files the REFERENCE's own writers produced --------------------------------
FORMATS = os.path.join(ROOT, "tests", "golden", "ref_formats.json")


def test_config_echo_is_byte_identical_to_the_references_saveToFile(tmp_path):
    """output/<name>/config.txt (main.cpp:786): `dot_hip --echo-config` on each of the 62 shipped scripts (+5 texts with
    the tokens none of them uses: tuning, appendStr, disableCout, restart, ADMM's iteration count, unknown view/shape
    names) against the bytes Config::saveToFile of the reference itself wrote for the same text
    (tests/golden/ref_formats.json, made by oracle/_ref/ref_formats = src/Config.cpp compiled in place).  The golden
    texts carry no `script` line (oracle/ref_formats.cpp) and show the marker @SCRIPT@ there; ours prints the
    parser's default name."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "dot_amd", "dot_hip")
    with open(FORMATS) as f:
        G = json.load(f)["config"]
    assert len(G) >= 65
    from tests.test_oracle_pin import reference_script_text
    nshipped = 0
    for rel, rec in G.items():
        src, dst = tmp_path / "s.txt", tmp_path / "config.txt"
        # the fixture stores the reference writer's OUTPUT; the shipped scripts' texts are read from the reference checkout
        text = rec["text"] if "text" in rec else reference_script_text(rel)
        if text is None:
            continue
        nshipped += "text" not in rec
        src.write_text(text)
        subprocess.check_call([exe, "100", str(src), "--echo-config", str(dst)])
        ours = dst.read_bytes()
        assert b"\nscript null\n" in ours, rel
        assert ours.replace(b"\nscript null\n", b"\nscript @SCRIPT@\n") == rec["echo"].encode(), rel
    assert nshipped >= 60 or not os.path.isdir("/root/reference/input")
    # and the script token itself round-trips
    src.write_text("energy FCR\ntimeStepper DOT 6\nscript twistnsns\nshape input a.msh\n")
    subprocess.check_call([exe, "100", str(src), "--echo-config", str(dst)])
    assert b"\nscript twistnsns\n" in dst.read_bytes()


def test_info_txt_is_byte_identical_to_the_references_timer_print(tmp_path):
    """info.txt (saveInfoForPresent, main.cpp:338-358): `dot_hip --write-info` against the file the reference's own
    Timer::print (Utils/Timer.hpp:58-68, compiled in place) wrote for the same numbers -- widths, %g-style values,
    totals, the three tables and the two header / one trailer lines."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "dot_amd", "dot_hip")
    with open(FORMATS) as f:
        G = json.load(f)["info"]
    assert len(G) >= 2
    for rec in G:
        dst = tmp_path / "info.txt"
        args = [str(rec["nV"]), str(rec["nT"]), str(rec["iterNum"]), str(rec["inner"])] + [repr(t) for t in rec["t"]]
        subprocess.check_call([exe, "--write-info", str(dst)] + args)
        assert dst.read_bytes() == rec["text"].encode()


# ---- bench.py: who starts the ranks -------------------------------------------------------------
def test_bench_launch_plan_never_reports_ranks_it_did_not_start():
    """`python bench.py --gpus N` without a launcher must start N ranks itself (or refuse), never run one rank and print N:
    plan_launch is the whole decision, a pure function of (--gpus, environment, visible GPUs, argv)."""
    import bench
    argv = ["bench.py", "--gpus", "8", "--steps", "5"]
    # one GPU asked for, no launcher: run in this process
    assert bench.plan_launch(1, {}, 1, argv) == ("run", None)
    assert bench.plan_launch(1, {}, 8, argv) == ("run", None)
    # N > 1 without a launcher: re-exec under torch.distributed.run with N ranks on 127.0.0.1 and the same arguments
    act, cmd = bench.plan_launch(8, {}, 8, argv)
    assert act == "reexec"
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-len(argv):] == argv
    # fewer GPUs than ranks: refuse loudly, launcher or not
    act, msg = bench.plan_launch(8, {}, 1, argv)
    assert act == "fail" and "8" in msg and "1 GPU" in msg
    act, msg = bench.plan_launch(2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, 1, argv)
    assert act == "fail"
    # started by the driver's launcher: go on as one of the ranks; a mismatch between --gpus and WORLD_SIZE is an error
    assert bench.plan_launch(4, {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3"}, 8, argv) == ("run", None)
    assert bench.plan_launch(8, {"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, 8, argv)[0] == "fail"
    assert bench.plan_launch(1, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, 8, argv)[0] == "fail"
    assert bench.plan_launch(0, {}, 8, argv)[0] == "fail"
    # a stale WORLD_SIZE in the environment without RANK is not a launcher
    assert bench.plan_launch(1, {"WORLD_SIZE": "8"}, 8, argv) == ("run", None)


def test_bench_reexec_starts_the_ranks_and_passes_their_exit_code(tmp_path):
    """the re-exec command really starts N processes with RANK / WORLD_SIZE set (a stand-in script instead of bench.py:
    no GPU here)"""
    import subprocess
    import bench
    script = tmp_path / "fake_bench.py"
    script.write_text("import os, sys\n"
                      "open(os.path.join(os.path.dirname(__file__), 'rank%s' % os.environ['RANK']), 'w')"
                      ".write(os.environ['WORLD_SIZE'] + ' ' + ' '.join(sys.argv[1:]))\n"
                      "sys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    act, cmd = bench.plan_launch(2, {}, 2, [str(script), "--gpus", "2"])
    assert act == "reexec"
    rc = subprocess.call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    assert rc != 0                                            # rank 1's failure is the job's failure
    assert (tmp_path / "rank0").read_text() == "2 --gpus 2"
    assert (tmp_path / "rank1").read_text() == "2 --gpus 2"


def test_bench_source_id_is_stable_and_pmc_files_carry_it():
    import bench
    a, b = bench.source_id(), bench.source_id()
    assert a == b and len(a) == 12 and int(a, 16) >= 0


# ---- the job table of the back-solve launches (dot_amd/csrc/bs_tiles.hpp; round 5) ---------------------------------------
def _plan_bs_tiles(name, env=None):
    import ctypes as C
    from dot_amd import lib as dl
    from dot_amd.sharding import plan_layout
    from dot_amd.workloads import load_workload
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        sc, ep, n = load_workload(name)
        L = dl.load()
        T = np.ascontiguousarray(sc.T, dtype=np.int32)
        X = np.ascontiguousarray(sc.V_rest, dtype=np.float64)
        epa = np.ascontiguousarray(ep, dtype=np.int32)
        counts = np.zeros(8, dtype=np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        dp = X.ctypes.data_as(C.POINTER(C.c_double))
        assert L.dotmi_plan_backsolve_tiles(X.shape[0], T.shape[0], ip(T), dp, ip(epa), n, 0, n, 0, None, ip(counts)) == 0
        tiles = np.zeros((int(counts[0]), 6), dtype=np.int32)
        assert L.dotmi_plan_backsolve_tiles(X.shape[0], T.shape[0], ip(T), dp, ip(epa), n, 0, n, tiles.shape[0], ip(tiles), ip(counts)) == 0
        nodes, nmax, pos, verts = plan_layout(sc.V_rest, sc.T, ep, n)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    return tiles, dict(zip("tiles n_wide n_narrow n_packs n_long nmax shallow few".split(), (int(c) for c in counts))), nodes, nmax, pos, n


@pytest.mark.parametrize("name", ["bar17K_twist", "monkey18K_stiff", "horse7K_stretch", "synbar:40x10x10:256", "kingkong18K_SS_1K"])
def test_backsolve_job_table_covers_every_live_row_once_and_sorts_tiles_into_their_kernels(name):
    """The product's own planner (dotmi_plan_backsolve_tiles = nd_choose_depth + nd_plan + plan_backsolve_tiles, host only):
    every live row of every subdomain lies in exactly one tile, a tile stays inside one 64-row block of the factor storage and
    starts at or right of its region's first column, tile indices of a part are 0 .. k-1, and every tile sits in the launch /
    kernel form its row length asks for: packs of four (one wavefront each) up to 256 columns, the 256-thread kernel up to 3072,
    the 512-thread kernel up to 5120, the two-phase kernel beyond; one-tile jobs heavy first, pack members at most four per job."""
    tiles, c, nodes, nmax, pos, n = _plan_bs_tiles(name)
    assert c["nmax"] == nmax and c["tiles"] == len(tiles)
    part, r0, rows, cb, idx, job = tiles.T
    ln = r0 + rows - cb
    assert (rows >= 1).all() and (rows <= 64).all() and ((r0 % 64) + rows <= 64).all() and (cb >= 0).all() and (cb <= r0).all()
    for p in range(n):
        m = part == p
        live = np.zeros(nmax, dtype=np.int32)
        for q in pos[p]:
            live[q:q + 3] += 1
        cover = np.zeros(nmax, dtype=np.int32)
        for a, k in zip(r0[m], rows[m]):
            cover[a:a + k] += 1
        assert np.array_equal(cover, live), (name, p)                     # every live row once, no padding row at all
        assert sorted(idx[m]) == list(range(int(m.sum())))
    nW, nN, nP = c["n_wide"], c["n_narrow"], c["n_packs"]
    long_ = job < 0
    assert long_.sum() == c["n_long"] and (ln[long_] > 5120).all() and (ln[~long_] <= 5120).all()
    wide = (~long_) & (ln > 3072)
    assert wide.sum() == nW and sorted(job[wide]) == list(range(nW))
    packed = (~long_) & (~wide) & (job >= nN)
    single = (~long_) & (~wide) & (job < nN)
    assert sorted(job[single]) == list(range(nN))
    if nP:
        assert (ln[packed] <= 256).all() and (ln[single] > 256).all()
        assert np.bincount(job[packed] - nN, minlength=nP).max() <= 4 and len(np.unique(job[packed])) == nP
    else:
        assert packed.sum() == 0
    work = (rows * (r0 + 64 - cb)).astype(np.int64)
    order = np.argsort(job[single], kind="stable")
    assert (np.diff(work[single][order]) <= 0).all()                      # heavy first: the hardware starts jobs in index order


def test_backsolve_job_table_reproduces_the_launches_measured_on_the_gpu():
    """Counts the -DBS_PROFILE build printed on the device (profiles/r05_backsolve_tiles.txt) from the host-only planner: bar17K on
    its three-level layout = 469 one-tile jobs (212 root tiles of 32 rows: the 4-pass rule of a shallow launch) + 199 packs = 668
    workgroups; without packs and with 64-row tiles 1163; the stiff monkey 394 + 231 = 625; horse7K / 8 keeps the few-subdomains
    rule and two levels; the switches select what they say."""
    _, c, _, nmax, _, _ = _plan_bs_tiles("bar17K_twist")
    assert (nmax, c["n_wide"], c["n_narrow"], c["n_packs"], c["shallow"], c["few"]) == (2944, 0, 469, 199, 1, 0)
    t, c, *_ = _plan_bs_tiles("bar17K_twist", {"DOTMI_WAVE_PACKS": "0", "DOTMI_TILE_PASSES": "8"})
    assert (c["n_narrow"], c["n_packs"], c["shallow"]) == (1163, 0, 0)
    assert int(((t[:, 1] + t[:, 2] - t[:, 3]) > 2560).sum()) == 110          # the root-separator tiles that ran the whole launch
    _, c, *_ = _plan_bs_tiles("bar17K_twist", {"DOTMI_TILE_PASSES": "8"})
    assert (c["n_narrow"], c["n_packs"]) == (367, 199)                         # 566 workgroups: packs alone
    _, c, *_ = _plan_bs_tiles("bar17K_twist", {"DOTMI_ND_LEVELS": "2"})
    assert c["nmax"] == 2368                                                    # round 4's two-level layout
    _, c, _, nmax, _, _ = _plan_bs_tiles("monkey18K_stiff")
    assert (nmax, c["n_narrow"] + c["n_packs"], c["n_packs"], c["shallow"]) == (1664, 625, 231, 1)
    _, c, _, nmax, _, _ = _plan_bs_tiles("horse7K_stretch")
    assert (nmax, c["few"], c["shallow"]) == (3456, 1, 0)


# ---- leaves-first layout of the two-level back-solve (dot_amd/csrc/nd_layout.hpp nd_relayout_leaves_first; round 6) ----------
@pytest.mark.parametrize("name,levels,min_split", [("bar17K_twist", -1, -1), ("synbar:40x10x10:8", 4, 256), ("horse7K_stretch", 3, 384)])
def test_leaves_first_layout_moves_regions_not_vertices(name, levels, min_split):
    """dotmi_plan_layout under DOTMI_TWO_LEVEL=1 (host only): the same tree, the same vertices in the same order inside every region,
    the regions re-placed -- every leaf in front (tree order), every separator behind them in post-order.  Checked: same padded size
    and node sizes; per subdomain the positions are distinct and a vertex keeps its offset INSIDE its region; every leaf row lies in
    front of every separator row; an internal node's `off` is the first separator column of its sub-tree, its separators a
    contiguous range that ends with its own."""
    from dot_amd.sharding import plan_layout
    from dot_amd.workloads import load_workload
    sc, ep, n = load_workload(name)
    old = os.environ.pop("DOTMI_TWO_LEVEL", None)
    try:
        nodes0, nmax0, pos0, verts0 = plan_layout(sc.V_rest, sc.T, ep, n, levels=levels, min_split=min_split)
        os.environ["DOTMI_TWO_LEVEL"] = "1"
        nodes1, nmax1, pos1, verts1 = plan_layout(sc.V_rest, sc.T, ep, n, levels=levels, min_split=min_split)
    finally:
        os.environ.pop("DOTMI_TWO_LEVEL", None)
        if old is not None:
            os.environ["DOTMI_TWO_LEVEL"] = old
    assert nmax0 == nmax1 and nodes0.shape == nodes1.shape and nodes0[0, 2] >= 0
    leaf = nodes0[:, 2] < 0
    assert np.array_equal(nodes0[:, 2:4], nodes1[:, 2:4]) and np.array_equal(nodes0[leaf, 1], nodes1[leaf, 1])
    assert np.array_equal(nodes0[~leaf, 5], nodes1[~leaf, 5])

    def region_of(nodes, q):   # node id and offset of padded position q inside that node's own region
        for k, (off, size, a, c, offS, sizeS) in enumerate(nodes):
            if a < 0 and off <= q < off + size:
                return k, q - off
            if a >= 0 and offS <= q < offS + sizeS:
                return k, q - offS
        raise AssertionError(q)

    first_sep = min(int(r[4]) for r in nodes1 if r[2] >= 0)
    assert all(int(r[0]) + int(r[1]) <= first_sep for r in nodes1 if r[2] < 0)          # every leaf in front of every separator
    for p in range(0, n, max(1, n // 6)):
        assert np.array_equal(verts0[p], verts1[p]) and len(set(pos1[p].tolist())) == len(pos1[p])
        for q0, q1 in list(zip(pos0[p], pos1[p]))[:: max(1, len(pos0[p]) // 50)]:
            assert region_of(nodes0, int(q0)) == region_of(nodes1, int(q1))

    def seps(k):   # separator ranges of the sub-tree of node k, in post-order
        off, size, a, c, offS, sizeS = (int(v) for v in nodes1[k])
        return [] if a < 0 else seps(a) + seps(c) + [(offS, sizeS)]

    for k in np.nonzero(~leaf)[0]:
        rng_ = seps(int(k))
        assert int(nodes1[k, 0]) == rng_[0][0]
        assert all(rng_[i][0] + rng_[i][1] == rng_[i + 1][0] for i in range(len(rng_) - 1))


# ---- which form of the block solve a mesh takes by default (dotmi_plan_backsolve_form, host only; round 6) -------------------------
@pytest.mark.parametrize("name,form,mb", [("bunny5K_LTSS", 0, 41.4), ("horse7K_stretch", 0, 107.3), ("monkey18K_stiff", 0, 123.6),
                                          ("bar17K_twist", 0, 196.6), ("kingkong18K_SS_1K", 1, 244.6),
                                          ("horse7K_stretch@r1:64", 1, 681.3)])
def test_default_form_of_the_block_solve_follows_the_one_pass_bytes(name, form, mb):
    """dotmi_create takes the two-level form where the explicit inverse would stream 240 MB or more per application on its own
    layout (counted over all subdomains; profiles/r06_two_level.txt H: the measured crossover).  Pinned here for the BASELINE
    workloads and the two that switched, with the byte counts the device handles report as precond_bytes in the one-pass form."""
    import ctypes as C
    from dot_amd import lib as dl
    from dot_amd.workloads import load_workload
    old = os.environ.pop("DOTMI_TWO_LEVEL", None)
    try:
        sc, ep, n = load_workload(name)
        L = dl.load()
        T = np.ascontiguousarray(sc.T, dtype=np.int32)
        X = np.ascontiguousarray(sc.V_rest, dtype=np.float64)
        epa = np.ascontiguousarray(ep, dtype=np.int32)
        f, b = C.c_int32(-1), C.c_int64(0)
        L.dotmi_plan_backsolve_form.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                                C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        rc = L.dotmi_plan_backsolve_form(X.shape[0], T.shape[0], T.ctypes.data_as(C.POINTER(C.c_int32)),
                                         X.ctypes.data_as(C.POINTER(C.c_double)), epa.ctypes.data_as(C.POINTER(C.c_int32)), n,
                                         C.byref(f), C.byref(b))
    finally:
        if old is not None:
            os.environ["DOTMI_TWO_LEVEL"] = old
    assert rc == 0 and f.value == form
    assert abs(b.value / 1e6 - mb) < 0.06, b.value
