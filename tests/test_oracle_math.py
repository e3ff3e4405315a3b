"""The reference's own notion of a kernel test (Energy::unitTest_*, Energy.cpp:1279-1521;
Energy::checkHessian :221-291; Optimizer::checkGradient, Optimizer.cpp:1331-1372): symbolic
derivatives against finite differences at random and degenerate inputs -- applied to the oracle."""
import ctypes as C

import numpy as np
import pytest

from tests.workloads import load_workload
from tests import oracle_py as O

dp = O._dp


def _tet(rng):
    Xr = rng.standard_normal((4, 3))
    Ds = (Xr[1:] - Xr[0]).T
    if np.linalg.det(Ds) < 0:
        Xr[[2, 3]] = Xr[[3, 2]]
        Ds = (Xr[1:] - Xr[0]).T
    return Xr, np.ascontiguousarray(np.linalg.inv(Ds))


@pytest.mark.parametrize("mat", [0, 1])
@pytest.mark.parametrize("amp", [0.0, 1e-8, 1e-2, 0.3])
def test_element_gradient_and_hessian_fd(mat, amp):
    L = O.lib()
    rng = np.random.default_rng(7 + mat)
    mu, lam, w = 100 / 2 / 1.4, 100 * 0.4 / 1.4 / 0.2, 0.7  # E=100, nu=0.4 (Energy.cpp:1283)
    for _ in range(8):
        Xr, A = _tet(rng)
        x = np.ascontiguousarray(Xr + amp * rng.standard_normal((4, 3)))
        g = np.zeros(12)
        L.dor_elem_energy_grad_x(mat, dp(x), dp(A), mu, lam, w, None, dp(g))
        H = np.zeros((12, 12))
        L.dor_elem_hessian_x(mat, dp(x), dp(A), mu, lam, w, 0, dp(H))
        eps = 1e-6
        fg = np.zeros(12); fH = np.zeros((12, 12))
        for k in range(12):
            xp = x.ravel().copy(); xp[k] += eps
            xm = x.ravel().copy(); xm[k] -= eps
            ep, em = C.c_double(), C.c_double()
            gp = np.zeros(12); gm = np.zeros(12)
            L.dor_elem_energy_grad_x(mat, dp(xp), dp(A), mu, lam, w, C.byref(ep), dp(gp))
            L.dor_elem_energy_grad_x(mat, dp(xm), dp(A), mu, lam, w, C.byref(em), dp(gm))
            fg[k] = (ep.value - em.value) / (2 * eps)
            fH[:, k] = (gp - gm) / (2 * eps)
        scale = max(np.abs(fH).max(), 1e-30)
        assert np.abs(g - fg).max() < 1e-6 * max(np.abs(fg).max(), mu * w)
        assert np.abs(H - fH).max() < 1e-6 * scale
        assert np.abs(H - H.T).max() < 1e-12 * scale


@pytest.mark.parametrize("mat", [0, 1])
def test_projected_hessian_is_psd_and_below_true_hessian(mat):
    L = O.lib()
    rng = np.random.default_rng(3)
    mu, lam, w = 3.0, 5.0, 1.0
    for _ in range(32):
        Xr, A = _tet(rng)
        x = np.ascontiguousarray(Xr + 0.6 * rng.standard_normal((4, 3)))  # includes inverted tets
        H = np.zeros((12, 12))
        L.dor_elem_hessian_x(mat, dp(x), dp(A), mu, lam, w, 1, dp(H))
        ev = np.linalg.eigvalsh(0.5 * (H + H.T))
        assert ev[0] > -1e-9 * max(ev[-1], 1.0)


def test_degenerate_inputs_do_not_produce_nans():
    L = O.lib()
    for F in (np.zeros((3, 3)), np.eye(3), np.diag([1.0, 1.0, 0.0]), np.ones((3, 3))):
        U = np.zeros((3, 3)); S = np.zeros(3); V = np.zeros((3, 3))
        L.dor_svd3(dp(np.ascontiguousarray(F)), dp(U), dp(S), dp(V))
        assert np.isfinite(U).all() and np.isfinite(V).all() and np.isfinite(S).all()
        assert np.abs(U @ np.diag(S) @ V.T - F).max() < 1e-14
        for mat in (0, 1):
            M = np.zeros((9, 9))
            L.dor_dPdF(mat, dp(U), dp(S), dp(V), 3.0, 5.0, 1.0, 1, dp(M))
            assert np.isfinite(M).all()


@pytest.fixture(scope="module")
def bunny_sim():
    sc, ep, nparts = load_workload("bunny5K_LTSS")
    cfg = sc.cfg
    sim = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep,
                      nparts, cfg.with_gravity)
    return sc, ep, nparts, sim


def test_global_gradient_fd_and_spmv(bunny_sim):
    sc, ep, nparts, sim = bunny_sim
    rng = np.random.default_rng(0)
    fx = sc.fixed.astype(bool)
    x = sc.x0 + 0.02 * rng.standard_normal(sc.x0.shape)
    p = rng.standard_normal(x.shape); p[fx] = 0
    eps = 1e-6
    dE = (sim.energy(x + eps * p) - sim.energy(x - eps * p)) / (2 * eps)
    g = sim.gradient(x)
    assert abs(dE - (g * p).sum()) < 1e-6 * abs(dE)
    assert np.abs(g[fx]).max() == 0.0
    # assembled H equals the element sum (+ mass, identity on fixed rows): DOTTimeStepper.cpp:588-613
    sim.refactor(x)
    He = sim.elem_hessians(x)
    _, _, mass, _, _ = sim.features()
    pe = np.where(fx[sc.T][:, :, None], 0.0, p[sc.T]).reshape(-1, 12)
    he = np.einsum("eij,ej->ei", He, pe).reshape(-1, 4, 3)
    ref = np.zeros_like(p)
    np.add.at(ref, sc.T, he)
    ref += mass[:, None] * p
    ref[fx] = p[fx]
    Hp = sim.spmv(p)
    assert np.abs(Hp - ref).max() < 1e-12 * np.abs(ref).max()
    sim.refactor(sc.x0)


def test_subdomain_matrix_is_principal_submatrix_and_precond_is_exact(bunny_sim):
    """H_s = R_s H R_s^T on free DOFs; p = D^-1 sum R^T H_s^-1 R q."""
    sc, ep, nparts, sim = bunny_sim
    rng = np.random.default_rng(1)
    fx = sc.fixed.astype(bool)
    r = rng.standard_normal(sc.x0.shape); r[fx] = 0
    ref = np.zeros_like(r)
    dup = sim.dup()
    assert dup.min() >= 1 and dup.sum() == sum(len(sim.part_verts(s)) for s in range(nparts))
    for s in range(nparts):
        l2g = sim.part_verts(s)
        # vertices of the part = vertices of its elements (ADMMDDTimeStepper.cpp:161-193)
        assert np.array_equal(l2g, np.unique(sc.T[ep == s]))
        Hs = sim.part_dense(s)
        assert np.abs(Hs - Hs.T).max() < 1e-12 * np.abs(Hs).max()
        # entries = H p restricted: check one probe vector through the global SpMV
        e = np.zeros_like(r)
        e[l2g] = rng.standard_normal((len(l2g), 3)); e[fx] = 0
        assert np.abs(sim.spmv(e)[l2g].ravel() - Hs @ e[l2g].ravel()).max() < 1e-10 * np.abs(Hs).max()
        ref[l2g] += np.linalg.solve(Hs, r[l2g].ravel()).reshape(-1, 3)
    ref /= np.maximum(dup, 1)[:, None]
    p = sim.apply_precond(r)
    assert np.abs(p - ref).max() < 1e-10 * np.abs(ref).max()
