"""Scene layer: script (Config) parsing, tet-mesh IO, normalisation, handle detection and the
scripted Dirichlet motion.  Host-side logic that sits *before* the hot path: it produces the plain
arrays the C ABI (include/dotmi.h) takes.

Mirrors, in numpy, what the reference does in
  src/Config.cpp:43-208          (script tokens)
  src/Utils/IglUtils.cpp:680-749 (.msh reader), :909-927 (findBorderVerts)
  src/main.cpp:692-712           (rotate, scale to `size`, move min corner to origin)
  src/AnimScripter.cpp:29-289    (initAnimScript), :291-470 (stepAnimScript)
  src/Mesh.cpp:741-744           (Lame parameters)
"""
from __future__ import annotations

import dataclasses
import math
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

ENERGY_FCR = 0
ENERGY_SNH = 1


@dataclasses.dataclass
class Config:
    """Subset of DOT::Config the DOT path reads (Config.hpp; defaults Config.cpp:33-37)."""
    energy: str = "FCR"
    time_stepper: str = "DOT"
    partition_amt: int = -1
    block_size: int = -1
    size: float = 1.0
    duration: float = 10.0
    dt: float = 0.025
    rho: float = 1.0
    YM: float = 100.0
    PR: float = 0.4
    with_gravity: bool = True
    script: str = "null"
    shape_path: str = ""
    rot_deg: float = 0.0
    rot_axis: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    handle_ratio: float = 0.01
    warm_start: int = 2
    tol: Optional[List[float]] = None
    restart: bool = False          # `restart <status file>` (Config.cpp:164-167)
    status_path: str = ""

    @property
    def energy_id(self) -> int:
        return ENERGY_SNH if self.energy == "SNH" else ENERGY_FCR


ENERGY_NAMES = ("FCR", "SNH")                               # Config.cpp:14-16
STEPPER_NAMES = ("Newton", "ADMM", "ADMMDD", "LBFGS", "LBFGSH", "LBFGSHI", "LBFGSJH", "DOT", "GSDD")   # :23-28


def parse_script(path: str) -> Config:
    """One token per line, order-free, unknown tokens ignored (Config.cpp:43-208)."""
    cfg = Config()
    with open(path) as f:
        lines = [l.strip() for l in f.readlines()]
    i = 0
    while i < len(lines):
        tok = lines[i].split()
        i += 1
        if not tok:
            continue
        key = tok[0]
        if key == "energy":
            # unknown names fall back to the default type, Config::getEnergyTypeByStr (Config.cpp:348-357)
            cfg.energy = tok[1] if tok[1] in ENERGY_NAMES else "SNH"
        elif key == "timeStepper":
            # unknown names are Newton, Config::getTimeStepperTypeByStr (Config.cpp:378-387)
            tok[1] = tok[1] if tok[1] in STEPPER_NAMES else "Newton"
            cfg.time_stepper = tok[1]
            # Config.cpp:62-80: only the domain-decomposed steppers read a partition count; a negative count is
            # followed by the nodes per block ("DOT -1 1024", main.cpp:792-798), 0 / 1 mean the default 4
            if tok[1] in ("ADMMDD", "DOT", "LBFGSJH", "GSDD") and len(tok) > 2:
                n = int(tok[2])
                cfg.partition_amt = n
                if n < 0:
                    if len(tok) > 3:
                        cfg.block_size = int(tok[3])
                elif n < 2:
                    cfg.partition_amt = 4
        elif key == "size":
            cfg.size = float(tok[1])
        elif key == "time":
            cfg.duration, cfg.dt = float(tok[1]), float(tok[2])
        elif key == "density":
            cfg.rho = float(tok[1])
        elif key == "stiffness":
            cfg.YM, cfg.PR = float(tok[1]), float(tok[2])
        elif key == "turnOffGravity":
            cfg.with_gravity = False
        elif key == "script":
            cfg.script = tok[1]
        elif key == "shape" and len(tok) > 2 and tok[1] == "input":
            cfg.shape_path = tok[2]
        elif key == "rotateModel":
            cfg.rot_axis = (float(tok[1]), float(tok[2]), float(tok[3]))   # axis first, Config.cpp:173-176
            cfg.rot_deg = float(tok[4])
        elif key == "handleRatio":
            cfg.handle_ratio = float(tok[1])
        elif key == "warmStart":
            cfg.warm_start = int(tok[1])
        elif key == "restart":
            cfg.restart = True
            cfg.status_path = tok[1] if len(tok) > 1 else ""
        elif key == "tol":
            n = int(tok[1])
            cfg.tol = [float(lines[i + k]) for k in range(n)]
            i += n
    return cfg


def read_tet_msh(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Custom MSH-4-like ASCII reader (IglUtils.cpp:680-749). Returns V (nV,3) f64, T (nT,4) i32."""
    with open(path) as f:
        lines = f.readlines()
    i = 0
    while not lines[i].startswith("$Nodes"):
        i += 1
    nV = int(lines[i + 1].split()[1])
    i += 3
    V = np.array([[float(t) for t in lines[i + k].split()[1:4]] for k in range(nV)], dtype=np.float64)
    i += nV
    while not lines[i].startswith("$Elements"):
        i += 1
    nT = int(lines[i + 1].split()[1])
    i += 3
    T = np.array([[int(t) for t in lines[i + k].split()[1:5]] for k in range(nT)], dtype=np.int32) - 1
    return V, T


def read_node_ele(prefix: str) -> Tuple[np.ndarray, np.ndarray]:
    """TetGen <prefix>.node / <prefix>.ele the way the reference reads them (IglUtils.cpp:751-793): the element
    indices are used as they are (zero-based files)."""
    with open(prefix + ".node") as f:
        tok = f.read().split()
    nN, nDim = int(tok[0]), int(tok[1])
    if nN < 4 or nDim != 3:
        raise ValueError(f"malformed {prefix}.node")
    V = np.array(tok[4:4 + 4 * nN], dtype=np.float64).reshape(nN, 4)[:, 1:4].copy()
    with open(prefix + ".ele") as f:
        tok = f.read().split()
    nE, nD1 = int(tok[0]), int(tok[1])
    if nD1 != 4:
        raise ValueError(f"malformed {prefix}.ele")
    T = np.array(tok[3:3 + 5 * nE], dtype=np.int64).reshape(nE, 5)[:, 1:5].astype(np.int32)
    return V, T


def load_tet_mesh(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """main.cpp:678-691: no suffix -> .node/.ele pair, '.msh' -> the MSH reader."""
    ext = os.path.splitext(path)[1]
    if ext == "":
        return read_node_ele(path)
    if ext == ".msh":
        return read_tet_msh(path)
    raise ValueError(f"unsupported tet mesh file format: {path}")


def write_status(path: str, timestep: int, x: np.ndarray, v: np.ndarray, xtilde: Optional[np.ndarray] = None) -> None:
    """status<n> in the reference's text format (Optimizer::saveStatus, Optimizer.cpp:1096-1132)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, 3)
    v = np.asarray(v, dtype=np.float64).reshape(-1)
    dxe = x - np.asarray(xtilde, dtype=np.float64).reshape(-1, 3) if xtilde is not None else np.zeros_like(x)
    with open(path, "w") as f:
        f.write(f"timestep {timestep}\n\nposition {x.shape[0]} 3\n")
        for r in x:
            f.write("%e %e %e\n" % tuple(r))
        f.write(f"\nvelocity {v.size}\n")
        for c in v:
            f.write("%e\n" % c)
        f.write(f"\ndx_Elastic {x.shape[0]} 3\n")
        for r in dxe:
            f.write("%e %e %e\n" % tuple(r))


def read_status(path: str, nV: int) -> Tuple[int, np.ndarray, np.ndarray]:
    """(timestep, x (nV,3), v (nV,3)) from a status file (the restart path of Optimizer's ctor, Optimizer.cpp:126-177)."""
    with open(path) as f:
        tok = f.read().split()
    i, timestep, x, v = 0, 0, None, None
    while i < len(tok):
        if tok[i] == "timestep":
            timestep = int(tok[i + 1]); i += 2
        elif tok[i] == "position":
            rows, cols = int(tok[i + 1]), int(tok[i + 2])
            if rows != nV or cols != 3:
                raise ValueError("status file does not match the mesh")
            x = np.array(tok[i + 3:i + 3 + 3 * nV], dtype=np.float64).reshape(nV, 3); i += 3 + 3 * nV
        elif tok[i] == "velocity":
            n = int(tok[i + 1])
            if n != 3 * nV:
                raise ValueError("status file does not match the mesh")
            v = np.array(tok[i + 2:i + 2 + n], dtype=np.float64).reshape(nV, 3); i += 2 + n
        elif tok[i] == "dx_Elastic":
            i += 3 + 3 * int(tok[i + 1])
        else:
            i += 1
    if x is None or v is None:
        raise ValueError("malformed status file")
    return timestep, x, v


def surface_triangles(T: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Surface triangles in the ORDER of IglUtils::findSurfaceTris (IglUtils.cpp:558-590) and the tet each belongs to
    (buildSTri2Tet, :591-625): the four outward faces of every tet go into a map keyed by the oriented vertex triple
    (lexicographic, Triplet.h:29-45); a face is on the surface when none of the three rotations of its reversal is a
    key; output in key order."""
    FACE = ((0, 2, 1), (0, 3, 2), (0, 1, 3), (1, 2, 3))
    tri = {}
    for e, t in enumerate(np.asarray(T).tolist()):
        for f in FACE:
            tri[(t[f[0]], t[f[1]], t[f[2]])] = e
    out, tet = [], []
    for k in sorted(tri):
        if (k[2], k[1], k[0]) in tri or (k[1], k[0], k[2]) in tri or (k[0], k[2], k[1]) in tri:
            continue
        out.append(k)
        tet.append(tri[k])
    return np.array(out, dtype=np.int32).reshape(-1, 3), np.array(tet, dtype=np.int32)


def surface_mesh(T: np.ndarray, SF: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
    """(surfIndToTet, F_surf): surface vertices in ascending tet-vertex order and the surface triangles re-indexed to
    them -- V_surf / F_surf of main.cpp:800-830, written as <n>.obj by Optimizer::saveStatus (Optimizer.cpp:1137-1150).
    SF: the `$Surface` rows of the mesh file when it has them, else findSurfaceTris."""
    if SF is None:
        SF, _ = surface_triangles(T)
    s2t = np.unique(SF)
    t2s = np.full(int(np.asarray(T).max()) + 1, -1, dtype=np.int64)
    t2s[s2t] = np.arange(s2t.size)
    return s2t.astype(np.int32), t2s[SF].astype(np.int32)


def load_mesh_npz(path: str) -> Tuple[np.ndarray, np.ndarray]:
    d = np.load(path)
    return np.ascontiguousarray(d["V"], dtype=np.float64), np.ascontiguousarray(d["T"], dtype=np.int32)


def angle_axis_matrix(angle: float, axis) -> np.ndarray:
    """Eigen::AngleAxis::toRotationMatrix (Rodrigues)."""
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / np.linalg.norm(ax)
    s, c = math.sin(angle), math.cos(angle)
    sin_axis = s * ax
    cos1_axis = (1.0 - c) * ax
    R = np.empty((3, 3))
    tmp = cos1_axis[0] * ax[1]
    R[0, 1] = tmp - sin_axis[2]
    R[1, 0] = tmp + sin_axis[2]
    tmp = cos1_axis[0] * ax[2]
    R[0, 2] = tmp + sin_axis[1]
    R[2, 0] = tmp - sin_axis[1]
    tmp = cos1_axis[1] * ax[2]
    R[1, 2] = tmp - sin_axis[0]
    R[2, 1] = tmp + sin_axis[0]
    R[0, 0] = cos1_axis[0] * ax[0] + c
    R[1, 1] = cos1_axis[1] * ax[1] + c
    R[2, 2] = cos1_axis[2] * ax[2] + c
    return R


def normalize(V: np.ndarray, size: float = 1.0, rot_deg: float = 0.0, rot_axis=(0, 1, 0)) -> np.ndarray:
    """main.cpp:692-712: optional rotation, longest bbox side -> `size`, min corner -> origin."""
    V = np.array(V, dtype=np.float64)
    if rot_deg != 0.0:
        R = angle_axis_matrix(rot_deg / 180.0 * math.pi, rot_axis)
        V = (R @ V.T).T
    V = V * (size / (V.max(axis=0) - V.min(axis=0)).max())
    V = V - V.min(axis=0)
    return np.ascontiguousarray(V)


def find_border_verts(V: np.ndarray, ratio: float) -> List[np.ndarray]:
    """IglUtils.cpp:909-927: two x-extreme slabs."""
    lo, hi = V.min(axis=0), V.max(axis=0)
    rng = hi - lo
    g0 = np.nonzero(V[:, 0] < lo[0] + rng[0] * ratio)[0]
    g1 = np.nonzero((V[:, 0] > hi[0] - rng[0] * ratio) & ~(V[:, 0] < lo[0] + rng[0] * ratio))[0]
    return [g0.astype(np.int32), g1.astype(np.int32)]


def lame(YM: float, PR: float) -> Tuple[float, float]:
    return YM / 2.0 / (1.0 + PR), YM * PR / (1.0 + PR) / (1.0 - 2.0 * PR)


class AnimScripter:
    """3-D Dirichlet scripts of the reference (AnimScripter.cpp). The fixed set is constant for every
    script except rubberBandPull, whose release step returns changed=True (-> dotmi_refix)."""

    SUPPORTED = ("null", "fall", "hang", "stretch", "squash", "stretchnsquash", "twist",
                 "twistnstretch", "twistnsns", "twistnsns_old", "rubberBandPull")

    def __init__(self, script: str, V_rest: np.ndarray, border: List[np.ndarray]):
        if script not in self.SUPPORTED:
            raise ValueError(f"unsupported script {script!r}")
        self.script = script
        nV = V_rest.shape[0]
        self.fixed = np.zeros(nV, dtype=np.uint8)
        self.ang_vel: Dict[int, float] = {}
        self.vel: Dict[int, np.ndarray] = {}
        bbox = np.stack([V_rest.min(axis=0), V_rest.max(axis=0)])
        self.rot_center = bbox.mean(axis=0)
        self.turn_vert = -1
        self._last = None   # see track()
        self._rot_cache = None
        self.turn_lo = -math.inf
        self.turn_hi = math.inf
        x0 = V_rest  # result.V == V_rest at init (main.cpp:712 UV = V)
        sgn = lambda b: (-1.0) ** b
        if script in ("null", "fall"):
            pass
        elif script == "rubberBandPull":
            # AnimScripter.cpp:220-258: top/bottom 2% slabs pulled apart in y, the waist pulled in -x
            lo, hi = V_rest.min(axis=0), V_rest.max(axis=0)
            rng = hi - lo
            self._rb_waist, self._rb_ends = [], []
            for v in range(nV):
                y = x0[v, 1]
                if y < lo[1] + rng[1] * 0.02:
                    self.fixed[v] = 1; self.vel[v] = np.array([0.0, -0.2, 0.0]); self._rb_ends.append(v)
                elif y > hi[1] - rng[1] * 0.02:
                    self.fixed[v] = 1; self.vel[v] = np.array([0.0, 0.2, 0.0]); self._rb_ends.append(v)
                elif (y < hi[1] - rng[1] * 0.48) and (y > lo[1] + rng[1] * 0.48):
                    self.fixed[v] = 1; self.vel[v] = np.array([-2.5, 0.0, 0.0]); self._rb_waist.append(v)
                    if self.turn_vert < 0:
                        self.turn_vert = v
                        self.turn_lo = x0[v, 0] - 5.0
        elif script == "hang":
            # AnimScripter.cpp:60-65 AST_HANG: fix the last vertex of each border group
            for grp in border:
                if len(grp):
                    self.fixed[grp[-1]] = 1
        else:
            for bI, grp in enumerate(border):
                self.fixed[grp] = 1
                for v in grp:
                    v = int(v)
                    if script == "stretch":
                        self.vel[v] = np.array([sgn(bI) * -0.1, 0, 0])
                    elif script == "squash":
                        self.vel[v] = np.array([sgn(bI) * 0.03, 0, 0])
                    elif script == "stretchnsquash":
                        self.vel[v] = np.array([sgn(bI) * -0.9, 0, 0])
                    elif script == "twist":
                        self.ang_vel[v] = sgn(bI) * -0.1 * math.pi
                    elif script == "twistnstretch":
                        self.ang_vel[v] = sgn(bI) * -0.1 * math.pi
                        self.vel[v] = np.array([sgn(bI) * -0.1, 0, 0])
                    elif script in ("twistnsns", "twistnsns_old"):
                        self.ang_vel[v] = sgn(bI) * -0.4 * math.pi
                        self.vel[v] = np.array([sgn(bI) * (-1.2 if script == "twistnsns" else -0.9), 0, 0])
            if script in ("twistnsns", "twistnsns_old", "stretchnsquash"):
                self.turn_vert = int(border[0][0])
                xv = x0[self.turn_vert, 0]
                if script == "twistnsns":
                    self.turn_lo, self.turn_hi = xv - 1.2, xv + 0.4
                elif script == "twistnsns_old":
                    self.turn_lo, self.turn_hi = xv - 0.8, xv + 0.4
                else:
                    self.turn_lo, self.turn_hi = xv - 0.8, xv + 0.4
        self.changed = False   # set by step() when the fixed set changed (rubberBandPull release)
        self.handle_idx = np.nonzero(self.fixed)[0].astype(np.int32)
        # dense per-handle arrays (same order as handle_idx) for the vectorised step
        self._w = np.array([self.ang_vel.get(int(v), 0.0) for v in self.handle_idx])
        self._vel = np.array([self.vel.get(int(v), np.zeros(3)) for v in self.handle_idx]).reshape(-1, 3)
        if script == "fall":
            self.init_offset_y = 0.5 * float(np.linalg.norm(V_rest.max(axis=0) - V_rest.min(axis=0)))
        else:
            self.init_offset_y = 0.0

    def initial_positions(self, V_rest: np.ndarray) -> np.ndarray:
        x = np.array(V_rest, dtype=np.float64)
        if self.init_offset_y:
            x[:, 1] += self.init_offset_y
        return x

    def step(self, x: np.ndarray, dt: float) -> Tuple[np.ndarray, np.ndarray]:
        """Returns (idx, new positions) of the scripted vertices for this step, from the current
        positions x (nV,3).  AnimScripter.cpp:291-470."""
        idx = self.handle_idx
        self.changed = False
        if idx.size == 0 or self.script in ("hang", "null", "fall"):
            return idx[:0], np.zeros((0, 3))
        if x is None:
            # Scripted vertices are Dirichlet nodes: the solver never moves them, so their current positions are
            # the ones this scripter set last (the reference reads them from the host mesh for free).  Not valid
            # for rubberBandPull, whose release frees scripted vertices.
            if self.script == "rubberBandPull" or self._last is None:
                raise ValueError("step(None, dt) needs track(x0) first and a script that keeps its fixed set")
            x = self._last
        if self.script == "rubberBandPull":
            if self.turn_vert >= 0 and x[self.turn_vert, 0] <= self.turn_lo:   # AnimScripter.cpp:404-417
                self.turn_lo = -math.inf
                self.fixed[self._rb_waist] = 0
                self._vel[:] = 0.0            # waist released, ends stop
                self.changed = True
            return idx, np.array(x[idx], dtype=np.float64) + self._vel * dt
        flip = False
        if self.turn_vert >= 0:
            xv = x[self.turn_vert, 0]
            flip = (xv <= self.turn_lo) or (xv >= self.turn_hi)
        xh = np.array(x[idx], dtype=np.float64)
        disp = np.zeros_like(xh)
        if self.ang_vel:
            if self._rot_cache is None or self._rot_cache[0] != dt:     # groups and matrices are the same every step
                groups = [(np.nonzero(self._w == w)[0], angle_axis_matrix(w * dt, (1.0, 0.0, 0.0)))
                          for w in np.unique(self._w)]
                self._rot_cache = (dt, groups)
            for sel, R in self._rot_cache[1]:
                rel = xh[sel] - self.rot_center
                disp[sel] = (rel @ R.T + self.rot_center) - xh[sel]
        if self.vel:
            if flip:
                self._vel[:, 0] *= -1.0
            disp += self._vel * dt
        pos = xh + disp
        if self._last is not None:
            self._last[idx] = pos
        return idx, pos

    def track(self, x0: np.ndarray) -> None:
        """Start remembering the scripted positions, so that step(None, dt) works without reading x back."""
        self._last = np.array(x0, dtype=np.float64, copy=True)


@dataclasses.dataclass
class Scene:
    cfg: Config
    V_rest: np.ndarray      # (nV,3) normalised rest positions
    T: np.ndarray           # (nT,4) int32
    scripter: AnimScripter
    x0: np.ndarray          # initial positions

    @property
    def fixed(self) -> np.ndarray:
        return self.scripter.fixed


def build_scene(cfg: Config, V_raw: np.ndarray, T: np.ndarray) -> Scene:
    V = normalize(V_raw, cfg.size, cfg.rot_deg, cfg.rot_axis if cfg.rot_deg else (0, 1, 0))
    border = find_border_verts(V, cfg.handle_ratio)
    scr = AnimScripter(cfg.script, V, border)
    return Scene(cfg=cfg, V_rest=V, T=np.ascontiguousarray(T, dtype=np.int32), scripter=scr,
                 x0=scr.initial_positions(V))


def load_scene(script_path: str, mesh_dir: Optional[str] = None) -> Scene:
    """Load a reference-format script. Mesh paths inside scripts are relative to the reference
    root; `mesh_dir` (default: tests/golden/meshes next to this package) supplies <name>.npz."""
    cfg = parse_script(script_path)
    name = os.path.splitext(os.path.basename(cfg.shape_path))[0]
    if mesh_dir is None:
        mesh_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "tests", "golden", "meshes")
    npz = os.path.join(mesh_dir, name + ".npz")
    if os.path.exists(npz):
        V, T = load_mesh_npz(npz)
    else:
        path = cfg.shape_path
        if not os.path.isabs(path) and not os.path.exists(path):
            path = os.path.join(mesh_dir, path)
        V, T = load_tet_mesh(path)
    return build_scene(cfg, V, T)


def synthetic_bar(nx: int, ny: int, nz: int, lx: float = 4.0, ly: float = 1.0, lz: float = 1.0,
                  jitter: float = 0.0, seed: int = 12345) -> Tuple[np.ndarray, np.ndarray]:
    """Deterministic box of nx*ny*nz cubes, 6 Kuhn tets each."""
    xs = np.linspace(0, lx, nx + 1)
    ys = np.linspace(0, ly, ny + 1)
    zs = np.linspace(0, lz, nz + 1)
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    V = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    if jitter > 0:
        rng = np.random.default_rng(seed)
        h = min(lx / nx, ly / ny, lz / nz)
        interior = ((X > 0) & (X < lx) & (Y > 0) & (Y < ly) & (Z > 0) & (Z < lz)).ravel()
        V[interior] += (rng.random((int(interior.sum()), 3)) - 0.5) * 2 * jitter * h

    def vid(i, j, k):
        return (i * (ny + 1) + j) * (nz + 1) + k

    I, J, K = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    I, J, K = I.ravel(), J.ravel(), K.ravel()
    c = [vid(I + a, J + b, K + d) for a in (0, 1) for b in (0, 1) for d in (0, 1)]
    # corners indexed by bits (a,b,d): c[4a+2b+d]; Kuhn: paths 000 -> 111
    perms = [(4, 2, 1), (4, 1, 2), (2, 4, 1), (2, 1, 4), (1, 4, 2), (1, 2, 4)]
    tets = []
    for p in perms:
        v0 = c[0]
        v1 = c[p[0]]
        v2 = c[p[0] + p[1]]
        v3 = c[7]
        tets.append(np.stack([v0, v1, v2, v3], axis=1))
    T = np.concatenate(tets, axis=0).astype(np.int32)
    # positive orientation
    d = V[T[:, 1:]] - V[T[:, :1]]
    det = np.einsum("ij,ij->i", d[:, 0], np.cross(d[:, 1], d[:, 2]))
    neg = det < 0
    T[neg, 2], T[neg, 3] = T[neg, 3].copy(), T[neg, 2].copy()
    return V, T


def refine_red(V: np.ndarray, T: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Regular (red) refinement: every tet -> 8 (four corner tets + the inner octahedron cut along the m02-m13
    diagonal, Bey's rule), one new vertex per edge.  Deterministic; used for the scale stand-in of the meshes the
    reference checkout lacks (horse136K, .MISSING_LARGE_BLOBS:11)."""
    nV = V.shape[0]
    pairs = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    E = np.concatenate([np.sort(T[:, list(pq)], axis=1) for pq in pairs], axis=0).astype(np.int64)
    key = E[:, 0] * nV + E[:, 1]
    uniq, inv = np.unique(key, return_inverse=True)
    mid = 0.5 * (V[uniq // nV] + V[uniq % nV])
    V2 = np.concatenate([V, mid], axis=0)
    nT = T.shape[0]
    m = {pq: (nV + inv[k * nT:(k + 1) * nT]).astype(np.int32) for k, pq in enumerate(pairs)}
    v = [T[:, k].astype(np.int32) for k in range(4)]
    m01, m02, m03, m12, m13, m23 = (m[pq] for pq in pairs)
    kids = [(v[0], m01, m02, m03), (m01, v[1], m12, m13), (m02, m12, v[2], m23), (m03, m13, m23, v[3]),
            (m01, m02, m03, m13), (m01, m02, m12, m13), (m02, m03, m13, m23), (m02, m12, m13, m23)]
    T2 = np.stack([np.stack(k, axis=1) for k in kids], axis=1).reshape(-1, 4).astype(np.int32)   # children of a tet adjacent
    d = V2[T2[:, 1:]] - V2[T2[:, :1]]
    det = np.einsum("ij,ij->i", d[:, 0], np.cross(d[:, 1], d[:, 2]))
    neg = det < 0
    T2[neg, 2], T2[neg, 3] = T2[neg, 3].copy(), T2[neg, 2].copy()
    return V2, T2


def partition_dual(V: np.ndarray, T: np.ndarray, nparts: int) -> np.ndarray:
    """The library's built-in partitioner (dotmi_partition, host only: recursive bisection of the tets' face-adjacency
    graph with Fiduccia-Mattheyses refinement) -- what stands in for METIS::partMesh on meshes without a fixture."""
    from . import lib as _lib
    L = _lib.load()
    V = np.ascontiguousarray(V, dtype=np.float64)
    T = np.ascontiguousarray(T, dtype=np.int32)
    ep = np.zeros(T.shape[0], dtype=np.int32)
    rc = L.dotmi_partition(V.shape[0], T.shape[0], _lib.ip(T), _lib.dp(V), int(nparts), _lib.ip(ep))
    if rc != 0:
        raise ValueError(f"dotmi_partition failed ({rc})")
    return ep


def partition_rcb(V: np.ndarray, T: np.ndarray, nparts: int) -> np.ndarray:
    """Seedless recursive coordinate bisection of element centroids (own partitioner for meshes
    without a METIS fixture; results then differ from the reference only within solver tolerance)."""
    cent = V[T].mean(axis=1)
    epart = np.zeros(T.shape[0], dtype=np.int32)

    def rec(ids, lo, n):
        if n == 1:
            epart[ids] = lo
            return
        c = cent[ids]
        ax = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
        order = ids[np.argsort(c[:, ax], kind="stable")]
        nl = n // 2
        cut = int(round(len(order) * nl / n))
        rec(order[:cut], lo, nl)
        rec(order[cut:], lo + nl, n - nl)

    rec(np.arange(T.shape[0]), 0, nparts)
    return epart
