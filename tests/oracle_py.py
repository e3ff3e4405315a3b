"""ctypes binding of oracle/liboracle.so (the CPU restatement) and oracle/_ref/librefpin.so (pieces
of the reference itself).  TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg -- never by the dot_amd package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_up = C.POINTER(C.c_ubyte)


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _ip(a):
    return a.ctypes.data_as(c_ip)


class StepStats(C.Structure):
    _fields_ = [("iters", C.c_int), ("ls_halvings", C.c_int), ("energy_evals", C.c_int),
                ("status", C.c_int), ("E0", C.c_double), ("g2_0", C.c_double), ("E", C.c_double),
                ("g2", C.c_double), ("ms_total", C.c_double), ("ms_energy", C.c_double),
                ("ms_gradient", C.c_double), ("ms_backsolve", C.c_double),
                ("ms_hessian", C.c_double), ("ms_factor", C.c_double)]


_lib = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ORACLE_DIR, "liboracle.so")
        if not os.path.exists(so):
            build_oracle()
        L = C.CDLL(so)
        L.dor_create.restype = C.c_void_p
        L.dor_create.argtypes = [C.c_int, C.c_int, c_dp, c_ip, C.c_double, C.c_double, C.c_double,
                                 C.c_int, C.c_double, C.c_int, c_up, c_dp, c_ip, C.c_int, C.c_double]
        L.dor_create_v.restype = C.c_void_p
        L.dor_create_v.argtypes = [C.c_int, C.c_int, c_dp, c_ip, C.c_double, C.c_double, C.c_double,
                                   C.c_int, C.c_double, C.c_int, c_up, c_dp, c_ip, c_ip, C.c_int, C.c_double]
        L.dor_destroy.argtypes = [C.c_void_p]
        L.dor_move.argtypes = [C.c_void_p, C.c_int, c_ip, c_dp]
        L.dor_step.argtypes = [C.c_void_p, C.POINTER(StepStats)]
        L.dor_step.restype = C.c_int
        L.dor_step_gsdd.argtypes = [C.c_void_p, C.POINTER(StepStats)]
        L.dor_step_gsdd.restype = C.c_int
        L.dor_step_newton.argtypes = [C.c_void_p, C.POINTER(StepStats)]
        L.dor_step_newton.restype = C.c_int
        L.dor_step_begin.argtypes = [C.c_void_p]
        L.dor_step_iterate.argtypes = [C.c_void_p]
        L.dor_step_iterate.restype = C.c_int
        L.dor_step_end.argtypes = [C.c_void_p, C.POINTER(StepStats), C.c_double]
        L.dor_step_end.restype = C.c_int
        L.dor_get_lbfgs.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp, c_dp]
        L.dor_get_lbfgs.restype = C.c_int
        L.dor_probe_direction.argtypes = [C.c_void_p, c_dp, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
        L.dor_last_iter_log.argtypes = [C.c_void_p, C.c_int, c_dp, c_dp, c_dp]
        L.dor_last_iter_log.restype = C.c_int
        L.dor_get_state.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
        L.dor_set_state.argtypes = [C.c_void_p, c_dp, c_dp, c_dp]
        L.dor_target_gres.argtypes = [C.c_void_p]
        L.dor_target_gres.restype = C.c_double
        L.dor_set_alpha_min.argtypes = [C.c_void_p, C.c_double]
        L.dor_set_fixed.argtypes = [C.c_void_p, c_up]
        L.dor_get_features.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp, c_dp]
        L.dor_get_dup.argtypes = [C.c_void_p, c_ip]
        L.dor_part_size.argtypes = [C.c_void_p, C.c_int]
        L.dor_part_size.restype = C.c_int
        L.dor_part_verts.argtypes = [C.c_void_p, C.c_int, c_ip]
        L.dor_eval_energy.argtypes = [C.c_void_p, c_dp]
        L.dor_eval_energy.restype = C.c_double
        L.dor_eval_gradient.argtypes = [C.c_void_p, c_dp, c_dp]
        L.dor_eval_elem_hessians.argtypes = [C.c_void_p, c_dp, c_dp]
        L.dor_refactor.argtypes = [C.c_void_p, c_dp]
        L.dor_apply_precond.argtypes = [C.c_void_p, c_dp, c_dp]
        L.dor_spmv.argtypes = [C.c_void_p, c_dp, c_dp]
        L.dor_part_dense.argtypes = [C.c_void_p, C.c_int, c_dp]
        L.dor_set_threads.argtypes = [C.c_int]
        # element-level
        L.dor_svd3.argtypes = [c_dp, c_dp, c_dp, c_dp]
        L.dor_psi.argtypes = [C.c_int, c_dp, C.c_double, C.c_double]
        L.dor_psi.restype = C.c_double
        L.dor_dpsi.argtypes = [C.c_int, c_dp, C.c_double, C.c_double, c_dp]
        L.dor_d2psi.argtypes = [C.c_int, c_dp, C.c_double, C.c_double, c_dp]
        L.dor_bleft.argtypes = [C.c_int, c_dp, C.c_double, C.c_double, c_dp]
        L.dor_make_pd3.argtypes = [c_dp]
        L.dor_make_pd2.argtypes = [c_dp]
        L.dor_dPdF.argtypes = [C.c_int, c_dp, c_dp, c_dp, C.c_double, C.c_double, C.c_double, C.c_int, c_dp]
        L.dor_elem_hessian_x.argtypes = [C.c_int, c_dp, c_dp, C.c_double, C.c_double, C.c_double, C.c_int, c_dp]
        L.dor_elem_energy_grad_x.argtypes = [C.c_int, c_dp, c_dp, C.c_double, C.c_double, C.c_double, c_dp, c_dp]
        _lib = L
    return _lib


class OracleSim:
    """Thin object wrapper over dor_sim."""

    def __init__(self, V_rest, T, YM, PR, rho, material, dt, fixed, x_init, epart, nparts,
                 with_gravity=True, rel_tol=1e-5, vpart=None):
        L = lib()
        self.nV, self.nT = V_rest.shape[0], T.shape[0]
        self.nparts = int(nparts)
        V_rest = np.ascontiguousarray(V_rest, dtype=np.float64)
        T = np.ascontiguousarray(T, dtype=np.int32)
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        x_init = np.ascontiguousarray(x_init, dtype=np.float64)
        epart = np.ascontiguousarray(epart, dtype=np.int32)
        if vpart is not None:
            vpart = np.ascontiguousarray(vpart, dtype=np.int32)
            self.h = L.dor_create_v(self.nV, self.nT, _dp(V_rest), _ip(T), YM, PR, rho, material, dt,
                                    int(with_gravity), fixed.ctypes.data_as(c_up), _dp(x_init),
                                    _ip(epart), _ip(vpart), self.nparts, rel_tol)
        else:
            self.h = L.dor_create(self.nV, self.nT, _dp(V_rest), _ip(T), YM, PR, rho, material, dt,
                                  int(with_gravity), fixed.ctypes.data_as(c_up), _dp(x_init),
                                  _ip(epart), self.nparts, rel_tol)

    def close(self):
        if self.h:
            lib().dor_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def move(self, idx, pos):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        lib().dor_move(self.h, idx.size, _ip(idx), _dp(pos))

    def step(self):
        st = StepStats()
        lib().dor_step(self.h, C.byref(st))
        return st

    def factor_failed(self) -> bool:
        """a refresh since creation (or since use_reference_cholmod) met a subdomain that could not be factorised"""
        L = lib()
        L.dor_factor_failed.argtypes = [C.c_void_p]
        L.dor_factor_failed.restype = C.c_int
        return bool(L.dor_factor_failed(self.h))

    def step_gsdd(self):
        st = StepStats()
        lib().dor_step_gsdd(self.h, C.byref(st))
        return st

    def step_newton(self):
        st = StepStats()
        lib().dor_step_newton(self.h, C.byref(st))
        return st

    # ---- a step in pieces (teacher forcing) ----
    def step_begin(self):
        lib().dor_step_begin(self.h)

    def step_iterate(self) -> int:
        """0 go on, 1 converged, 2 iteration cap, 3 line search failed"""
        return lib().dor_step_iterate(self.h)

    def step_end(self):
        st = StepStats()
        lib().dor_step_end(self.h, C.byref(st), 0.0)
        return st

    def lbfgs_state(self):
        """(x, g, S[m,nV,3], Y[m,nV,3], lastE) between two iterations of a running step"""
        n = 3 * self.nV
        x, g = np.zeros((self.nV, 3)), np.zeros((self.nV, 3))
        S, Y = np.zeros((5, n)), np.zeros((5, n))
        E = C.c_double()
        m = lib().dor_get_lbfgs(self.h, _dp(x), _dp(g), _dp(S), _dp(Y), C.cast(C.byref(E), c_dp))
        return x, g, S[:m].copy(), Y[:m].copy(), E.value

    def probe_direction(self, x, S=None, Y=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        m = 0 if S is None else len(S)
        Sa = np.ascontiguousarray(S, dtype=np.float64).reshape(m, -1) if m else None
        Ya = np.ascontiguousarray(Y, dtype=np.float64).reshape(m, -1) if m else None
        out = {k: np.zeros((self.nV, 3)) for k in ("g", "q", "z", "p")}
        a0, E = C.c_double(), C.c_double()
        lib().dor_probe_direction(self.h, _dp(x), m, _dp(Sa) if m else None, _dp(Ya) if m else None, _dp(out["g"]),
                                  _dp(out["q"]), _dp(out["z"]), _dp(out["p"]), C.cast(C.byref(a0), c_dp),
                                  C.cast(C.byref(E), c_dp))
        out["alpha0"], out["E"] = a0.value, E.value
        return out

    def iter_log(self):
        cap = 10001
        a, e, g = (np.zeros(cap) for _ in range(3))
        n = lib().dor_last_iter_log(self.h, cap, _dp(a), _dp(e), _dp(g))
        return a[:n], e[:n], g[:n]

    def state(self):
        x, v, xt = (np.zeros((self.nV, 3)) for _ in range(3))
        lib().dor_get_state(self.h, _dp(x), _dp(v), _dp(xt))
        return x, v, xt

    def set_state(self, x, v, xn=None):
        x = np.ascontiguousarray(x, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        xn_p = _dp(np.ascontiguousarray(xn, dtype=np.float64)) if xn is not None else None
        lib().dor_set_state(self.h, _dp(x), _dp(v), xn_p)

    def set_alpha_min(self, a):
        lib().dor_set_alpha_min(self.h, float(a))

    def set_fixed(self, fixed):
        fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
        lib().dor_set_fixed(self.h, fixed.ctypes.data_as(c_up))

    @property
    def target_gres(self):
        return lib().dor_target_gres(self.h)

    def features(self):
        A = np.zeros((self.nT, 9)); vol = np.zeros(self.nT); mass = np.zeros(self.nV)
        mu = np.zeros(self.nT); lam = np.zeros(self.nT)
        lib().dor_get_features(self.h, _dp(A), _dp(vol), _dp(mass), _dp(mu), _dp(lam))
        return A, vol, mass, mu, lam

    def dup(self):
        d = np.zeros(self.nV, dtype=np.int32)
        lib().dor_get_dup(self.h, _ip(d))
        return d

    def part_verts(self, p):
        n = lib().dor_part_size(self.h, p)
        a = np.zeros(n, dtype=np.int32)
        lib().dor_part_verts(self.h, p, _ip(a))
        return a

    def energy(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return lib().dor_eval_energy(self.h, _dp(x))

    def gradient(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros((self.nV, 3))
        lib().dor_eval_gradient(self.h, _dp(x), _dp(g))
        return g

    def elem_hessians(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        H = np.zeros((self.nT, 12, 12))
        lib().dor_eval_elem_hessians(self.h, _dp(x), _dp(H))
        return H

    def refactor(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        lib().dor_refactor(self.h, _dp(x))

    def apply_precond(self, r):
        r = np.ascontiguousarray(r, dtype=np.float64)
        p = np.zeros((self.nV, 3))
        lib().dor_apply_precond(self.h, _dp(r), _dp(p))
        return p

    def spmv(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        out = np.zeros((self.nV, 3))
        lib().dor_spmv(self.h, _dp(p), _dp(out))
        return out

    def part_dense(self, part):
        n = 3 * lib().dor_part_size(self.h, part)
        Hs = np.zeros((n, n))
        lib().dor_part_dense(self.h, part, _dp(Hs))
        return Hs


# ---- reference pieces (oracle/_ref/librefpin.so) -------------------------------------------------
_ref = None


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "librefpin.so"))


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "librefpin.so"))
        R.ref_svd.argtypes = [C.c_int, c_dp, c_dp, c_dp, c_dp]
        R.ref_svd.restype = C.c_int
        R.ref_autoflip_svd.argtypes = [c_dp, c_dp, c_dp, c_dp]
        R.ref_make_pd3.argtypes = [c_dp]
        R.ref_make_pd2.argtypes = [c_dp]
        R.ref_hessian_from_dPdF.argtypes = [c_dp, c_dp, c_dp]
        R.ref_energy_phat.argtypes = [C.c_int, C.c_int, c_dp, c_dp, c_dp, c_dp, c_dp]
        R.ref_energy_phat.restype = C.c_int
        _ref = R
    return _ref


_ref_nofma = None


def ref_nofma_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "librefpin_nofma.so"))


def ref_nofma():
    """the same reference sources compiled with -ffp-contract=off (oracle/Makefile)"""
    global _ref_nofma
    if _ref_nofma is None:
        R = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "librefpin_nofma.so"))
        R.ref_svd.argtypes = [C.c_int, c_dp, c_dp, c_dp, c_dp]
        R.ref_svd.restype = C.c_int
        _ref_nofma = R
    return _ref_nofma


def use_reference_svd(on: bool, contracted: bool = True):
    """Route every simulation-level SVD of the oracle through the reference's own AVX kernel (librefpin.so:ref_svd,
    Utils/SVD_EFTYCHIOS compiled in place; contracted=False: the build without FMA contraction) -- or back to dor_svd3."""
    L = lib()
    L.dor_set_svd_batch.argtypes = [C.c_void_p]
    R = ref() if contracted else ref_nofma()
    L.dor_set_svd_batch(C.cast(R.ref_svd, C.c_void_p) if on else None)


_refsolver = None


def ref_solver_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "librefsolver.so")) and os.path.exists(
        "/opt/conda/lib/libmkl_rt.so.1")


def _load_refsolver():
    global _refsolver
    if _refsolver is None:
        os.environ.setdefault("MKL_THREADING_LAYER", "SEQUENTIAL")
        # the BLAS is named by its full path: an rpath to /opt/conda/lib would also pull in that directory's libstdc++
        C.CDLL("/opt/conda/lib/libmkl_rt.so.1", mode=C.RTLD_GLOBAL)
        _refsolver = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "librefsolver.so"))
        _refsolver.ref_linsys_run.argtypes = [C.c_int, C.c_int, c_ip, c_up, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, c_dp]
        _refsolver.ref_linsys_run.restype = C.c_int
        _refsolver.ref_linsys_time.argtypes = [C.c_int, C.c_int, c_ip, c_up, c_dp, c_dp, C.c_int, C.c_int, c_dp]
        _refsolver.ref_linsys_time.restype = C.c_int
    return _refsolver


class ExtSolverAPI(C.Structure):
    """dor_ext_solver (oracle/dot_oracle.h)"""
    _fields_ = [("create", C.c_void_p), ("factor", C.c_void_p), ("solve", C.c_void_p), ("destroy", C.c_void_p)]


def use_reference_cholmod(sim, on=True):
    """Route the subdomain factorisations and solves of an OracleSim through the reference's own CHOLMODSolver, one
    object per subdomain kept across the steps (oracle/_ref/librefsolver.so: ref_sub_*, oracle/ref_linsys.cpp) -- or back
    to the built-in envelope Cholesky."""
    L = lib()
    L.dor_use_ext_solver.argtypes = [C.c_void_p, C.c_void_p]
    L.dor_use_ext_solver.restype = C.c_int
    if not on:
        return L.dor_use_ext_solver(sim.h, None)
    R = _load_refsolver()
    api = ExtSolverAPI(*(C.cast(getattr(R, n), C.c_void_p) for n in ("ref_sub_create", "ref_sub_factor", "ref_sub_solve",
                                                                      "ref_sub_destroy")))
    return L.dor_use_ext_solver(sim.h, C.byref(api))


def ref_linsys_time(T, fixed, He, mass, nfact=3, nsolve=10):
    """(ms per numeric factorisation, ms per solve, non-zeros of L) of the reference's CHOLMODSolver on the matrix of this
    (sub-)mesh"""
    R = _load_refsolver()
    T = np.ascontiguousarray(T, dtype=np.int32)
    fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
    He = np.ascontiguousarray(He, dtype=np.float64)
    mass = np.ascontiguousarray(mass, dtype=np.float64)
    out = np.zeros(3)
    if R.ref_linsys_time(mass.size, T.shape[0], _ip(T), fixed.ctypes.data_as(c_up), _dp(He), _dp(mass), nfact, nsolve, _dp(out)):
        raise RuntimeError("reference CHOLMODSolver::factorize failed")
    return float(out[0]), float(out[1]), int(out[2])


def ref_linsys(T, fixed, He, mass, rhs, v, want_dense=False):
    """The reference's LinSysSolver + CHOLMODSolver (oracle/ref_linsys.cpp): assemble the global matrix from the
    element Hessians as DOTTimeStepper::computeHElemAndFillIn does, factorize, solve(rhs), multiply(v)."""
    _refsolver = _load_refsolver()
    T = np.ascontiguousarray(T, dtype=np.int32)
    fixed = np.ascontiguousarray(fixed, dtype=np.uint8)
    He = np.ascontiguousarray(He, dtype=np.float64)
    mass = np.ascontiguousarray(mass, dtype=np.float64)
    rhs = np.ascontiguousarray(rhs, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    nV, nT = mass.size, T.shape[0]
    n = 3 * nV
    sol, Av = np.zeros(n), np.zeros(n)
    dense = np.zeros((n, n)) if want_dense else None
    rc = _refsolver.ref_linsys_run(nV, nT, _ip(T), fixed.ctypes.data_as(c_up), _dp(He), _dp(mass), _dp(rhs), _dp(v),
                                   _dp(sol), _dp(Av), _dp(dense) if want_dense else None)
    if rc != 0:
        raise RuntimeError("reference CHOLMODSolver::factorize failed")
    return sol, Av, dense


def ref_config_parse(path):
    """DOT::Config::loadFromFile of the reference (oracle/_ref/ref_config, oracle/ref_config.cpp) -> dict of strings"""
    txt = subprocess.check_output([os.path.join(ORACLE_DIR, "_ref", "ref_config"), path]).decode()
    out = {}
    for line in txt.splitlines():
        k, _, v = line.partition(" ")
        out[k] = v
    return out


def metis_partition(T, nV, nparts, tmpdir="/tmp", nodal=False):
    """Run the reference's vendored METIS (oracle/_ref/metis_part) on a tet list: the element partition of
    METIS::partMesh, or (nodal) the vertex partition of METIS::partMesh_nodes."""
    exe = os.path.join(ORACLE_DIR, "_ref", "metis_part")
    tin = os.path.join(tmpdir, f"_tets_{os.getpid()}.i32")
    tout = os.path.join(tmpdir, f"_epart_{os.getpid()}.i32")
    np.ascontiguousarray(T, dtype=np.int32).tofile(tin)
    subprocess.check_call([exe, tin, str(nV), str(nparts), tout] + (["nodal"] if nodal else []),
                          stderr=subprocess.DEVNULL)
    ep = np.fromfile(tout, dtype=np.int32)
    os.remove(tin)
    os.remove(tout)
    return ep
