"""Python-side mirror of the reference's stepper surface over the C ABI.

`DOTTimeStepper` keeps the method names main.cpp calls on `DOT::Optimizer<3>`
(src/TimeStepper/Optimizer.hpp:83-112): setRelGL2Tol, solve, getResult, getIterNum,
getInnerIterAmt, plus the kernel-level hooks.  All numerics run in libdotmi.so on the GPU.
(The C++ adapter with the same surface is dot_amd/host/DotHipTimeStepper.hpp.)
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import lib as _lib
from .lib import DotmiError, Mesh, Params, StepStats, dp, ip, up
from .scene import Scene, lame


class DOTTimeStepper:
    def __init__(self, scene: Scene, epart: np.ndarray, nparts: int, energy: Optional[int] = None,
                 device: int = 0, rank: int = 0, world: int = 1, comm_id: Optional[bytes] = None,
                 history: int = 5, rel_tol: float = 1e-5, iter_cap: int = 10000, flags: int = 0, allreduce=None,
                 alpha_min: float = 0.1, vpart: Optional[np.ndarray] = None):
        """allreduce: optional callable(np.ndarray) that sums the array over the ranks IN PLACE (world > 1): the
        library then stages its collectives through host memory and calls it instead of RCCL
        (dotmi_params::allreduce) -- e.g. a torch.distributed gloo all_reduce."""
        L = _lib.load()
        cfg = scene.cfg
        self.scene = scene
        self.nV, self.nT = scene.V_rest.shape[0], scene.T.shape[0]
        self.dt = cfg.dt
        self.frameAmt = int(cfg.duration / cfg.dt)  # Optimizer::setTime, Optimizer.cpp:249-257
        self.globalIterNum = 0
        self.innerIterAmt = 0
        mu, lam = lame(cfg.YM, cfg.PR)
        # keep every array alive for the lifetime of the handle
        self._X = np.ascontiguousarray(scene.V_rest, dtype=np.float64)
        self._T = np.ascontiguousarray(scene.T, dtype=np.int32)
        self._mu = np.full(self.nT, mu, dtype=np.float64)
        self._lam = np.full(self.nT, lam, dtype=np.float64)
        self._fixed = np.ascontiguousarray(scene.fixed, dtype=np.uint8)
        self._epart = np.ascontiguousarray(epart, dtype=np.int32)
        self.nparts = int(nparts)
        self._vpart = np.ascontiguousarray(vpart, dtype=np.int32) if vpart is not None else None
        m = Mesh(self.nV, self.nT, dp(self._X), ip(self._T), dp(self._mu), dp(self._lam), cfg.rho,
                 up(self._fixed), ip(self._epart), self.nparts, ip(self._vpart) if vpart is not None else None)
        p = Params()
        p.energy = cfg.energy_id if energy is None else energy
        p.dt = cfg.dt
        p.gravity[0], p.gravity[1], p.gravity[2] = 0.0, (-9.80665 if cfg.with_gravity else 0.0), 0.0
        p.relTol = rel_tol
        p.history = history
        p.iterCap = iter_cap
        p.alphaMin = alpha_min   # 0.1 = DOT's clamp (Optimizer.cpp:1085); 1.0 = the unit first step of LBFGS-H (:1088)
        p.device = device
        p.rank, p.world = rank, world
        self._comm = C.create_string_buffer(comm_id, 128) if comm_id is not None else None
        p.comm_id = C.cast(self._comm, C.c_void_p) if self._comm is not None else None
        p.flags = flags
        self._arcb = None
        self._ar_error = None
        if allreduce is not None:
            def _cb(ctx, buf, n, _f=allreduce):
                # an exception cannot cross the C frame: ctypes would print and swallow it and the library would carry
                # on with this rank's un-reduced partial sums.  Record it, poison the payload (NaN spreads through
                # every later sum, so the step fails its convergence test on this rank) and re-raise from _check.
                a = np.ctypeslib.as_array(buf, shape=(n,))
                try:
                    if self._ar_error is None:
                        _f(a)
                    else:
                        a[:] = np.nan
                except BaseException as e:   # noqa: BLE001 - must not propagate into the C caller
                    self._ar_error = e
                    a[:] = np.nan
            self._arcb = _lib.ALLREDUCE_CB(_cb)          # kept alive with the handle
            p.allreduce = C.cast(self._arcb, C.c_void_p)
        x0 = np.ascontiguousarray(scene.x0, dtype=np.float64)
        h = C.c_void_p()
        rc = L.dotmi_create(C.byref(m), C.byref(p), dp(x0), C.byref(h))
        if rc != 0:
            raise DotmiError(f"dotmi_create failed ({rc}): {L.dotmi_last_error(None).decode()}")
        self._h = h
        self._L = L
        self.last_stats: Optional[StepStats] = None

    # ---- lifetime -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.dotmi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if getattr(self, "_ar_error", None) is not None:
            e, self._ar_error = self._ar_error, None
            raise DotmiError(f"{what}: the all-reduce hook raised {type(e).__name__}: {e}") from e
        if rc < 0:
            raise DotmiError(f"{what} failed ({rc}): {self._L.dotmi_last_error(self._h).decode()}")
        return rc

    # ---- Optimizer surface ----------------------------------------------------------------------
    @property
    def targetGRes(self) -> float:
        return self._L.dotmi_target_gres(self._h)

    def getIterNum(self) -> int:
        return self.globalIterNum

    def getInnerIterAmt(self) -> int:
        return self.innerIterAmt

    def getResult(self) -> np.ndarray:
        """result.V (nV,3)"""
        x = np.empty((self.nV, 3))
        self._check(self._L.dotmi_get_state(self._h, dp(x), None, None), "get_state")
        return x

    def getState(self):
        x, v, xt = (np.empty((self.nV, 3)) for _ in range(3))
        self._check(self._L.dotmi_get_state(self._h, dp(x), dp(v), dp(xt)), "get_state")
        return x, v, xt

    def setState(self, x, v, xn=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        xn_ = np.ascontiguousarray(xn, dtype=np.float64) if xn is not None else None
        self._check(self._L.dotmi_set_state(self._h, dp(x), dp(v), dp(xn_) if xn_ is not None else None),
                    "set_state")

    def setDirichlet(self, idx, pos):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        self._check(self._L.dotmi_set_dirichlet(self._h, idx.size, ip(idx), dp(pos)), "set_dirichlet")

    def refix(self, fixed):
        """fixed set changed: DOTTimeStepper::updatePrecondMtrAndFactorize (DOTTimeStepper.cpp:185-270)"""
        self._fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        self._check(self._L.dotmi_refix(self._h, up(self._fixed)), "refix")

    def solve(self, maxIter: int = 1) -> int:
        """Optimizer::solve (Optimizer.cpp:327-368): script move, one BE step. 0 stepped, 1 all frames
        done, 2 stepped but hit the iteration cap / line-search failure."""
        flag = 0
        for _ in range(maxIter):
            x = self.getResult()
            idx, pos = self.scene.scripter.step(x, self.dt)
            if idx.size:
                self.setDirichlet(idx, pos)
            if getattr(self.scene.scripter, "changed", False):
                self.refix(self.scene.scripter.fixed)      # updatePrecondMtrAndFactorize
            if self.globalIterNum >= self.frameAmt:
                self.globalIterNum += 1
                return 1
            st = self.step()
            if st.status == 2:
                flag = 2
            self.globalIterNum += 1
        return flag

    def step(self) -> StepStats:
        """fullyImplicit + BE update without the script move."""
        st = StepStats()
        self._check(self._L.dotmi_step(self._h, C.byref(st)), "step")
        self.innerIterAmt += st.iters
        self.last_stats = st
        return st

    def iterLog(self):
        cap = 10001
        a, e, g = (np.zeros(cap) for _ in range(3))
        n = self._L.dotmi_last_iter_log(self._h, cap, dp(a), dp(e), dp(g))
        return a[:n], e[:n], g[:n]

    # ---- kernel-level hooks ---------------------------------------------------------------------
    def computeEnergyVal(self, x) -> float:
        x = np.ascontiguousarray(x, dtype=np.float64)
        E = C.c_double()
        self._check(self._L.dotmi_eval_energy(self._h, dp(x), C.cast(C.byref(E), _lib.c_dp)), "eval_energy")
        return E.value

    def computeGradient(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.empty((self.nV, 3))
        self._check(self._L.dotmi_eval_gradient(self._h, dp(x), dp(g)), "eval_gradient")
        return g

    def computeElemHessians(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        H = np.empty((self.nT, 12, 12))
        self._check(self._L.dotmi_eval_elem_hessians(self._h, dp(x), dp(H)), "eval_elem_hessians")
        return H

    def updatePrecondMtrAndFactorize(self, x=None):
        xp = dp(np.ascontiguousarray(x, dtype=np.float64)) if x is not None else None
        self._check(self._L.dotmi_refactor(self._h, xp), "refactor")

    def applyPrecond(self, r) -> np.ndarray:
        r = np.ascontiguousarray(r, dtype=np.float64)
        p = np.empty((self.nV, 3))
        self._check(self._L.dotmi_apply_precond(self._h, dp(r), dp(p)), "apply_precond")
        return p

    def probeDirection(self, x, S=None, Y=None):
        """one solve_oneStep up to its first trial from a given iterate + history -> dict(g, q, z, p, alpha0, E)"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        m = 0 if S is None else len(S)
        Sa = np.ascontiguousarray(S, dtype=np.float64).reshape(m, -1) if m else None
        Ya = np.ascontiguousarray(Y, dtype=np.float64).reshape(m, -1) if m else None
        out = {k: np.empty((self.nV, 3)) for k in ("g", "q", "z", "p")}
        a0, E = C.c_double(), C.c_double()
        self._check(self._L.dotmi_probe_direction(
            self._h, dp(x), m, dp(Sa) if m else None, dp(Ya) if m else None, dp(out["g"]), dp(out["q"]),
            dp(out["z"]), dp(out["p"]), C.cast(C.byref(a0), _lib.c_dp), C.cast(C.byref(E), _lib.c_dp)),
            "probe_direction")
        out["alpha0"], out["E"] = a0.value, E.value
        return out

    def multiply(self, p) -> np.ndarray:
        p = np.ascontiguousarray(p, dtype=np.float64)
        out = np.empty((self.nV, 3))
        self._check(self._L.dotmi_spmv(self._h, dp(p), dp(out)), "spmv")
        return out

    def features(self):
        A = np.empty((self.nT, 9)); vol = np.empty(self.nT); mass = np.empty(self.nV)
        self._check(self._L.dotmi_get_features(self._h, dp(A), dp(vol), dp(mass)), "get_features")
        return A, vol, mass

    def partMatrix(self, part: int, inverse: bool = False):
        n = self._L.dotmi_part_size(self._h, part)
        M = np.empty((n, n))
        l2g = np.empty(n // 3, dtype=np.int32)
        self._check(self._L.dotmi_part_matrix(self._h, part, int(inverse), dp(M), ip(l2g)), "part_matrix")
        return M, l2g

    def backsolveForm(self) -> int:
        """0: explicit inverse in one pass; 1: two-level form (dotmi_backsolve_form)"""
        return int(self._L.dotmi_backsolve_form(self._h))

    def benchPrecond(self, reps: int = 50):
        ms = C.c_double(); nb = C.c_int64()
        self._check(self._L.dotmi_bench_precond(self._h, reps, C.cast(C.byref(ms), _lib.c_dp), C.byref(nb)),
                    "bench_precond")
        return ms.value, nb.value

    def benchEnergy(self, reps: int = 50):
        ms = C.c_double(); nb = C.c_int64()
        self._check(self._L.dotmi_bench_energy(self._h, reps, C.cast(C.byref(ms), _lib.c_dp), C.byref(nb)),
                    "bench_energy")
        return ms.value, nb.value


def comm_unique_id() -> bytes:
    L = _lib.load()
    buf = C.create_string_buffer(128)
    rc = L.dotmi_comm_unique_id(buf)
    if rc != 0:
        raise DotmiError("dotmi_comm_unique_id failed")
    return buf.raw
