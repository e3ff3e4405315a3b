"""GPU parity tests added in round 2 (through the C ABI, against the CPU oracle): the BASELINE.json configurations
that round 1 left without a `-m gpu` test, ABI calls the bundled scripters never make, and the failure paths."""
import numpy as np
import pytest

from dot_amd import lib as dl
from dot_amd import scene
from tests.workloads import load_workload
from dot_amd.timestepper import DOTTimeStepper
from tests import oracle_py as O

pytestmark = pytest.mark.gpu


def make_pair(name, energy=None, nparts=None, **kw):
    sc, ep, n = load_workload(name, nparts)
    if energy is not None:
        sc.cfg.energy = energy
    cfg = sc.cfg
    ts = DOTTimeStepper(sc, ep, n, **kw)
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n,
                      cfg.with_gravity)
    return sc, ep, n, ts, orc


def test_set_dirichlet_with_changing_index_sets_then_step():
    """a caller that passes a different index list on every call (e.g. only the still-fixed vertices
    after a release) went through a destroyed HIP event.  Two different sets, then a step, against the oracle."""
    sc, ep, n, ts, orc = make_pair("synbar:8x3x3:4")
    fixed = np.nonzero(sc.fixed)[0].astype(np.int32)
    assert fixed.size >= 8
    x = ts.getResult()
    for k, sel in enumerate([fixed, fixed[::2], fixed[1::2], fixed[:3]]):
        pos = x[sel] + 1e-3 * (k + 1)
        ts.setDirichlet(sel, pos)
        orc.move(sel, pos)
        x[sel] = pos
        st, so = ts.step(), orc.step()
        assert st.status == 0 and st.iters == so.iters
        assert np.abs(ts.getResult() - orc.state()[0]).max() < 1e-9
        x = ts.getResult()
    ts.close(); orc.close()


def test_failed_factorisation_poisons_the_handle():
    """after DOTMI_E_NOTSPD the handle must not keep stepping on garbage factors."""
    sc, ep, n = load_workload("synbar:4x2x2:2")
    ts = DOTTimeStepper(sc, ep, n)
    # an inside-out state: every tet inverted and scaled by 1e8 -> the projected Hessian stays PSD, so instead
    # poison through the refactor entry with NaN positions (every pivot test `piv > 0` fails on NaN)
    bad = np.full_like(sc.x0, np.nan)
    with pytest.raises(dl.DotmiError) as e:
        ts.updatePrecondMtrAndFactorize(bad)
    assert "-3" in str(e.value)
    with pytest.raises(dl.DotmiError) as e:
        ts.step()
    assert "-3" in str(e.value) and "invalid" in str(e.value)
    with pytest.raises(dl.DotmiError):
        ts.applyPrecond(np.ones_like(sc.x0))
    ts.updatePrecondMtrAndFactorize(sc.x0)   # a good factorisation heals it
    assert ts.step().status == 0
    ts.close()


def run_both(sc, ts, orc, nsteps):
    out = []
    for _ in range(nsteps):
        x = ts.getResult()
        idx, pos = sc.scripter.step(x, sc.cfg.dt)
        ts.setDirichlet(idx, pos)
        orc.move(idx, pos)
        out.append((ts.step(), orc.step()))
    return out


# ---- BASELINE.json configs[2]: horse, FixedCoRot, 8 subdomains (horse7K is the stand-in for the missing 136K mesh) --
def test_horse7K_stretch_fcr_8_steps_match_oracle():
    """Steps 0-4 of input/tb1_horse_scalab/horse7K_stretch_DOT.txt (FCR, 8 METIS parts): identical L-BFGS iterations
    and back-tracking halvings on all five; positions to 1e-9 while the line search does not amplify rounding
    (steps 0-3: measured 2e-16, 3e-16, 1e-14, 7e-12).  Step 4 back-tracks six times and the free runs drift to 4e-6
    with the same decisions, step 5 takes 8 vs 11 halvings: the teacher-forced tests below show that each single
    iteration of those steps still agrees to rounding."""
    sc, ep, n, ts, orc = make_pair("horse7K_stretch")
    assert sc.cfg.energy == "FCR" and n == 8
    # BASELINE.md section 2, the reference's own run of this script with 8 parts: 9 11 14 16 20 21 28 27 29 32.  The
    # HIP path reproduces steps 0-1 exactly and 2-4 within one (13 17 19): this script back-tracks from step 2 on
    # and the counts then depend on the SVD kernel's rounding (the oracle with the reference's own AVX SVD plugged
    # in takes 9 12 13 17 19 -- also +-1, on other steps; see test_oracle_pin.py)
    published = [9, 11, 14, 16, 20]
    for k in range(5):
        (st, so), = run_both(sc, ts, orc, 1)
        assert (st.status, st.iters, st.ls_halvings) == (so.status, so.iters, so.ls_halvings), k
        assert st.iters == published[k] if k < 2 else abs(st.iters - published[k]) <= 1, (k, st.iters)
        assert st.g2 <= ts.targetGRes
        dx = np.abs(ts.getResult() - orc.state()[0]).max()
        if k < 4:
            assert dx < 1e-9 and abs(st.E - so.E) <= 1e-10 * abs(so.E), (k, dx)
        else:
            assert dx < 1e-4 and abs(st.E - so.E) <= 1e-5 * abs(so.E), (k, dx)   # trajectory band
        if k == 3:
            x, v, xt = ts.getState()
            xo, vo, xto = orc.state()
            assert np.abs(v - vo).max() < 1e-7 and np.abs(xt - xto).max() < 1e-9
    ts.close(); orc.close()


# ---- BASELINE.json configs[4]: synthetic 1M-tet bar twist, StableNH, 256 subdomains -------------------------------
def test_synbar_256_parts_reduced_size_matches_oracle():
    """The same generator, script, material and 256-part recursive-coordinate-bisection partition as the 1M-tet
    configuration on a mesh the oracle finishes in seconds (40x10x10 cubes = 24 000 tets)."""
    sc, ep, n, ts, orc = make_pair("synbar:40x10x10:256")
    assert n == 256 and sc.T.shape[0] == 24000 and sc.cfg.energy == "SNH"
    for k, (st, so) in enumerate(run_both(sc, ts, orc, 3)):
        assert (st.status, st.iters, st.ls_halvings) == (so.status, so.iters, so.ls_halvings), k
    assert np.abs(ts.getResult() - orc.state()[0]).max() < 1e-9
    r = np.random.default_rng(3).standard_normal((sc.V_rest.shape[0], 3))
    r[sc.fixed.astype(bool)] = 0
    po = orc.apply_precond(r)
    assert np.abs(ts.applyPrecond(r) - po).max() <= 1e-9 * np.abs(po).max()
    ts.close(); orc.close()


def test_synbar_1M_tets_256_parts_full_size_properties():
    """configs[4] at full size (140x35x35 cubes = 1 029 000 tets, 182 736 vertices, 256 parts) through the
    size-independent properties: the line search never accepts an energy increase, the step ends below the
    tolerance, the block solve is H_s^-1 on right-hand sides one subdomain owns (two subdomains), H_s is the principal sub-matrix
    of the global H (checked through the SpMV), the block solve is linear, and a second handle reproduces the step bit for bit."""
    sc, ep, n = load_workload("synbar:140x35x35:256")
    assert sc.T.shape[0] == 1029000 and sc.V_rest.shape[0] == 182736 and n == 256
    cfg = sc.cfg

    def one(ts):
        x = ts.getResult()
        idx, pos = sc.scripter.step(x, cfg.dt)
        ts.setDirichlet(idx, pos)
        st = ts.step()
        return st, ts.getResult(), ts.iterLog()

    ts = DOTTimeStepper(sc, ep, n)
    st, x1, (alpha, E, g2) = one(ts)
    assert st.status == 0 and st.iters == len(E) and st.g2 <= ts.targetGRes
    assert E[0] <= st.E0 and np.all(np.diff(E) <= 0.0)          # E(x + alpha p) > E(x) is never accepted
    assert np.all((alpha > 0) & (alpha <= 1.0))
    assert g2[-1] == st.g2 and np.all(g2[:-1] > ts.targetGRes)
    free = ~sc.fixed.astype(bool)
    rng = np.random.default_rng(11)
    # (round 6: at this size the factors are in the two-level form -- there is no explicit inverse X_s to read back, and the
    # checks below go through H_s itself: M r = H_s^-1 r on right-hand sides a single subdomain owns)
    two_level = ts.backsolveForm() == 1
    assert two_level
    for part in (0, 137):
        Hs, l2g = ts.partMatrix(part)
        ns = Hs.shape[0]
        assert np.abs(Hs - Hs.T).max() == 0.0 and np.linalg.eigvalsh(Hs).min() > 0.0
        # rows of the global SpMV restricted to the part's vertices == H_s (free dofs; fixed rows are identity)
        p = np.zeros((sc.V_rest.shape[0], 3))
        p[l2g] = rng.standard_normal((l2g.size, 3))
        Hp = ts.multiply(p)[l2g].reshape(-1)
        assert np.abs(Hp - Hs @ p[l2g].reshape(-1)).max() <= 1e-10 * np.abs(Hp).max()
    # the block solve is linear, and on right-hand sides supported on vertices owned by ONE subdomain it is that
    # subdomain's X^T X (no averaging there: dup = 1)
    nV = sc.V_rest.shape[0]
    r = rng.standard_normal((nV, 3)) * free[:, None]
    r2 = rng.standard_normal((nV, 3)) * free[:, None]
    Mr, Mr2, Mc = ts.applyPrecond(r), ts.applyPrecond(r2), ts.applyPrecond(0.5 * r - 2.0 * r2)
    assert np.abs(Mc - (0.5 * Mr - 2.0 * Mr2)).max() <= 1e-11 * np.abs(Mr).max()
    dup = np.zeros(nV, dtype=np.int32)
    for p_ in range(n):
        dup[np.unique(sc.T[ep == p_])] += 1
    for part in (0, 137):
        Hs, l2g = ts.partMatrix(part)
        own = (dup[l2g] == 1) & free[l2g]
        assert own.sum() > 10
        rs = np.zeros((l2g.size, 3))
        rs[own] = rng.standard_normal((int(own.sum()), 3))
        rr = np.zeros((nV, 3))
        rr[l2g] = rs
        zs = np.linalg.solve(Hs, rs.reshape(-1)).reshape(-1, 3)
        z = ts.applyPrecond(rr)
        assert np.abs(z[l2g][own] - zs[own]).max() <= 1e-9 * np.abs(zs).max()
    ts.close()
    sc2, ep2, _ = load_workload("synbar:140x35x35:256")     # fresh scripter state
    ts2 = DOTTimeStepper(sc2, ep2, n)
    x = ts2.getResult()
    idx, pos = sc2.scripter.step(x, cfg.dt)
    ts2.setDirichlet(idx, pos)
    st2 = ts2.step()
    assert (st2.iters, st2.ls_halvings, st2.E, st2.g2) == (st.iters, st.ls_halvings, st.E, st.g2)
    assert np.array_equal(ts2.getResult(), x1)
    ts2.close()


# ---- teacher-forced per-iteration parity (reference rows DOTTimeStepper.cpp:299,304,329) -------
def teacher_forced(sc, ts, orc, max_iters):
    """Walk the ORACLE through one time step; before every iteration hand its (x, history) to both probes and compare
    what the HIP path and the oracle make of the same input.  Returns the per-iteration relative errors."""
    err = {k: [] for k in ("g", "q", "z", "p", "alpha0", "E")}
    decisions = 0
    orc.step_begin()
    it = 0
    while it < max_iters:
        xk, gk, S, Y, lastE = orc.lbfgs_state()
        po = orc.probe_direction(xk, S, Y)
        pd = ts.probeDirection(xk, S, Y)
        for k in ("g", "q", "z", "p"):
            err[k].append(np.abs(pd[k] - po[k]).max() / np.abs(po[k]).max())
        err["alpha0"].append(abs(pd["alpha0"] - po["alpha0"]) / po["alpha0"])
        err["E"].append(abs(pd["E"] - po["E"]) / abs(po["E"]))
        # the accept / halve decision of the first trial, where it is not a rounding coin flip
        if abs(po["E"] - lastE) > 1e-9 * abs(lastE):
            assert (pd["E"] > lastE) == (po["E"] > lastE), it
            decisions += 1
        it += 1
        if orc.step_iterate() != 0:
            break
    return {k: np.array(v) for k, v in err.items()}, it, decisions


def sync_to_oracle(ts, orc):
    """start-of-step state of the oracle -> the device: (x, v, x_n), x~ and the factors at x_n"""
    x, v, _ = orc.state()
    ts.setState(x, v, x)
    ts.updatePrecondMtrAndFactorize(x)


def test_teacher_forced_stiff_monkey_first_100_iterations():
    """monkey18K, StableNH, E = 4e5, dt = 0.04, 64 parts (BASELINE.json configs[3]): the regime with > 100 halvings
    per step, where free-running trajectories part ways inside step 0.  With the oracle's own (x, history) forced in,
    every iteration's q, block solve z, direction p, alpha_0 and trial energy agree to the bounds below -- so the
    early divergence of free runs is rounding amplified by the line search, not a different preconditioner."""
    sc, ep, n, ts, orc = make_pair("monkey18K_stiff")
    x = ts.getResult()
    idx, pos = sc.scripter.step(x, sc.cfg.dt)
    ts.setDirichlet(idx, pos)
    orc.move(idx, pos)
    # same start-of-step x (scripted handles moved); x~ and the factors are those of the rest state on both sides
    err, iters, decisions = teacher_forced(sc, ts, orc, 101)
    assert iters >= 100
    print("teacher-forced monkey18K_stiff: max rel err over", iters, "iterations:",
          {k: float(v.max()) for k, v in err.items()}, "decisions checked:", decisions)
    assert err["g"].max() < 1e-10 and err["q"].max() < 1e-10
    assert err["z"].max() < 1e-8, err["z"].max()       # the block solve with the explicit inverse factor
    assert err["p"].max() < 1e-8
    assert err["alpha0"].max() < 1e-8 and err["E"].max() < 1e-12
    ts.close(); orc.close()


@pytest.mark.parametrize("step", [4, 5])
def test_teacher_forced_horse7K_back_tracking_steps(step):
    """horse7K_stretch free runs drift at step 4 (same decisions, 4e-6 in x) and differ at step 5 (8 vs 11 halvings).
    Each of those steps teacher-forced from the oracle's start-of-step state: every iteration agrees to rounding."""
    sc, ep, n, ts, orc = make_pair("horse7K_stretch")
    for _ in range(step):
        x = orc.state()[0]
        idx, pos = sc.scripter.step(x, sc.cfg.dt)
        orc.move(idx, pos)
        orc.step()
    sync_to_oracle(ts, orc)
    x = orc.state()[0]
    idx, pos = sc.scripter.step(x, sc.cfg.dt)
    orc.move(idx, pos)
    ts.setDirichlet(idx, pos)
    err, iters, decisions = teacher_forced(sc, ts, orc, 200)
    print(f"teacher-forced horse7K step {step}: max rel err over", iters, "iterations:",
          {k: float(v.max()) for k, v in err.items()}, "decisions checked:", decisions)
    assert iters >= 10 and decisions >= iters // 2
    assert err["g"].max() < 1e-10 and err["q"].max() < 1e-10
    assert err["z"].max() < 1e-9 and err["p"].max() < 1e-9
    assert err["alpha0"].max() < 1e-9 and err["E"].max() < 1e-12
    ts.close(); orc.close()


# ---- the reference's own LinSysSolver + CHOLMODSolver as golden (tests/golden/ref_linsys.npz) ------------------------
@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_device_assembly_spmv_and_solve_match_reference_cholmod_solver(k):
    """HIP path vs vectors produced by the reference's compiled LinSysSolver.hpp + CHOLMODSolver.cpp (vendored CHOLMOD):
    element Hessians -> global block-CSR -> SpMV (a9, CHOLMODSolver::multiply) and, with the whole mesh as one
    subdomain, dense fill -> inverse-Cholesky factor -> back-solve (a10, a11: factorize + solve)."""
    import json
    import os
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_linsys.npz"))
    meta = json.loads(str(G["meta"]))[k]
    sc, ep, n = load_workload(meta["workload"])
    sc.cfg.energy = meta["energy"]
    ts = DOTTimeStepper(sc, ep, n)
    ts.updatePrecondMtrAndFactorize(G[f"c{k}_x"])
    Av = ts.multiply(G[f"c{k}_v"])
    assert np.abs(Av - G[f"c{k}_Av"]).max() <= 1e-12 * np.abs(G[f"c{k}_Av"]).max()
    sol = ts.applyPrecond(G[f"c{k}_rhs"])
    assert np.abs(sol - G[f"c{k}_sol"]).max() <= 1e-9 * np.abs(G[f"c{k}_sol"]).max()
    if f"c{k}_dense" in G:
        Hs, l2g = ts.partMatrix(0)
        assert np.array_equal(l2g, np.arange(sc.V_rest.shape[0]))
        D = G[f"c{k}_dense"]
        assert np.abs(Hs - D).max() <= 1e-13 * np.abs(D).max()
    ts.close()


# ---- the reference's shipped scripts use `timeStepper DOT 6`: subdomains of ~3000 vertices ---------------------------
# iterations per step of the CPU oracle on this configuration (10 steps, ~2.5 s each on 8 threads, recorded from
# oracle/dot_oracle.c; the first two are re-run live below).  BASELINE.md section 2 lists 9 10 12 14 14 15 15 16 15 16
# for the reference's own run of the script as shipped: +-1 on four steps.  MEASURED CAUSE (round 3,
# tests/test_oracle_pin.py::test_published_bar17K_as_shipped_list_needs_the_references_svd_rounding): the SVD kernel's
# rounding.  With the reference's own compiled AVX SVD plugged into the oracle and nothing else changed, the oracle
# takes exactly the published 9 10 12 14 14 15 15 16 15 16.  The step-0 preconditioner is the projected Hessian of the
# rest state, where makePD2d halves each B block on a rounding coin flip (88 % of the element Hessians differ by up
# to 23 % between the two SVDs), and step 0 stops at |g|^2 / tol = 0.976 after 8 iterations with exact Jacobi
# rotations (device and oracle) but needs a 9th with the reference kernel.  Partition and solver are not involved.
BAR6_ITERS = [8, 10, 12, 14, 14, 14, 15, 15, 16, 15]
BAR6_PUBLISHED = [9, 10, 12, 14, 14, 15, 15, 16, 15, 16]


def test_bar17K_as_shipped_fcr_6_subdomains():
    """input/bar17K_twist_DOT.txt:2 `timeStepper DOT 6` with the reference's METIS partition (fixture bar17K_6): the
    largest subdomain has 3271 vertices (n_s = 9813), beyond the 4096-column register tile of the single-pass
    back-solve -- its separator rows go through the two-phase long-row kernel and the dissection tree is 4 deep.
    The device path takes the oracle's iterations on all ten steps (within one of the reference's published run) and
    agrees with the oracle run live on the first two (its envelope Cholesky needs seconds per step at this size)."""
    sc, ep, n = load_workload("bar17K_twist", 6)
    sc.cfg.energy = "FCR"
    cfg = sc.cfg
    assert n == 6 and max(np.unique(sc.T[ep == p]).size for p in range(6)) == 3271
    ts = DOTTimeStepper(sc, ep, n)
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n,
                      cfg.with_gravity)
    r = np.random.default_rng(5).standard_normal((sc.V_rest.shape[0], 3)) * (1 - sc.fixed[:, None])
    po = orc.apply_precond(r)
    assert np.abs(ts.applyPrecond(r) - po).max() <= 1e-9 * np.abs(po).max()
    iters = []
    for k in range(10):
        x = ts.getResult()
        idx, pos = sc.scripter.step(x, cfg.dt)
        ts.setDirichlet(idx, pos)
        st = ts.step()
        assert st.status == 0 and st.g2 <= ts.targetGRes
        iters.append(st.iters)
        if k < 2:
            orc.move(idx, pos)
            so = orc.step()
            assert (st.iters, st.ls_halvings) == (so.iters, so.ls_halvings)
            assert np.abs(ts.getResult() - orc.state()[0]).max() < 1e-9
    assert iters == BAR6_ITERS, iters
    assert all(abs(a - b) <= 1 for a, b in zip(iters, BAR6_PUBLISHED))
    ts.close(); orc.close()


# ---- f3: block-size mode `timeStepper DOT -1 1024` (main.cpp:792-798, input/tb5_ablation/*-1K.txt) ------------------
@pytest.mark.parametrize("name", ["kingkong18K_SS_1K", "monkey18K_TSS_1K"])
def test_block_size_mode_matches_oracle(name):
    """nV / 1024 + 1 = 18 subdomains (the reference's METIS partition for 18 parts as fixture), FCR, the ablation
    scripts' stretchnsquash / twistnsns_old: three steps against the oracle."""
    sc, ep, n, ts, orc = make_pair(name)
    assert n == sc.V_rest.shape[0] // 1024 + 1 == 18 and sc.cfg.block_size == 1024
    halvings = 0
    for k in range(3):
        (st, so), = run_both(sc, ts, orc, 1)
        assert (st.status, st.iters, st.ls_halvings) == (so.status, so.iters, so.ls_halvings), k
        halvings += st.ls_halvings
        dx = np.abs(ts.getResult() - orc.state()[0]).max()
        print(name, "step", k, "iters", st.iters, "halvings", st.ls_halvings, "max|dx|", dx)
        # a back-tracking line search amplifies rounding differences (see the teacher-forced tests): rounding-level
        # agreement is asserted until the first halving, the trajectory band of SURVEY 8(c) F5 after it
        assert dx < (1e-9 if halvings == 0 else 1e-5), (k, dx)
    ts.close(); orc.close()


def test_builtin_partitioner_through_the_abi():
    """dotmi_mesh.epart == NULL: the library partitions the elements itself (dotmi_partition, what stands in for
    METIS::partMesh on meshes without a fixture).  Same result as passing that partition explicitly; the oracle with
    the same partition agrees; and the partition is close to METIS in interface size where RCB is not."""
    import ctypes as C
    sc, ep_metis, n = load_workload("kingkong18K_SS_1K")
    cfg = sc.cfg
    ep = scene.partition_dual(sc.V_rest, sc.T, n)
    assert ep.min() == 0 and ep.max() == n - 1 and np.bincount(ep).min() > 0.9 * ep.size / n

    def interface(e):
        vs = [np.unique(sc.T[e == p]) for p in range(n)]
        dup = np.zeros(sc.V_rest.shape[0], dtype=int)
        for v in vs:
            dup[v] += 1
        return int(sum((dup[v] > 1).sum() for v in vs))

    i_metis, i_own, i_rcb = interface(ep_metis), interface(ep), interface(scene.partition_rcb(sc.V_rest, sc.T, n))
    print("interface vertices kingkong18K/18: METIS", i_metis, "dotmi_partition", i_own, "RCB", i_rcb)
    assert i_own < 1.35 * i_metis and i_own < 0.85 * i_rcb
    ts0 = DOTTimeStepper(sc, ep, n)
    ts1 = DOTTimeStepper(sc, ep, n)
    # rebuild ts1 with a NULL partition through the raw ABI
    L = dl.load()
    ts1.close()
    m = dl.Mesh(ts0.nV, ts0.nT, dl.dp(ts0._X), dl.ip(ts0._T), dl.dp(ts0._mu), dl.dp(ts0._lam), cfg.rho,
                dl.up(ts0._fixed), None, n)
    p = dl.Params()
    p.energy, p.dt, p.relTol, p.history, p.iterCap, p.alphaMin = cfg.energy_id, cfg.dt, 1e-5, 5, 10000, 0.1
    p.gravity[1] = -9.80665
    p.world = 1
    h = C.c_void_p()
    x0 = np.ascontiguousarray(sc.x0)
    assert L.dotmi_create(C.byref(m), C.byref(p), dl.dp(x0), C.byref(h)) == 0
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n,
                      cfg.with_gravity)
    x = ts0.getResult()
    idx, pos = sc.scripter.step(x, cfg.dt)
    ts0.setDirichlet(idx, pos)
    orc.move(idx, pos)
    idx32, posc = np.ascontiguousarray(idx, dtype=np.int32), np.ascontiguousarray(pos)
    assert L.dotmi_set_dirichlet(h, idx32.size, dl.ip(idx32), dl.dp(posc)) == 0
    st0, so = ts0.step(), orc.step()
    st1 = dl.StepStats()
    assert L.dotmi_step(h, C.byref(st1)) == 0
    assert (st0.iters, st0.E) == (st1.iters, st1.E) and st0.iters == so.iters
    x1 = np.empty_like(x0)
    L.dotmi_get_state(h, dl.dp(x1), None, None)
    assert np.array_equal(x1, ts0.getResult()) and np.abs(x1 - orc.state()[0]).max() < 1e-9
    L.dotmi_destroy(h)
    ts0.close(); orc.close()


# ---- f4, first slice: the reference's GSDD iteration on the same factors and kernels --------------------------------
@pytest.mark.parametrize("name,energy,steps", [("bunny5K_LTSS", "FCR", 2), ("synbar:12x4x4:6", "SNH", 3)])
def test_gsdd_steps_match_oracle(name, energy, steps):
    """`timeStepper GSDD <n>` (DOTTimeStepper::solve_oneStep_GSDD, DOTTimeStepper.cpp:507-565): Gauss-Seidel sweeps over
    the subdomains, each with its own solve + line search from step 1.  DOTMI_FLAG_GSDD vs dor_step_gsdd: identical
    sweep counts and halvings, positions to 1e-9; and it converges to the same tolerance as L-BFGS-H."""
    sc, ep, n, ts, orc = make_pair(name, energy=energy, flags=dl.FLAG_GSDD)
    for k in range(steps):
        x = ts.getResult()
        idx, pos = sc.scripter.step(x, sc.cfg.dt)
        ts.setDirichlet(idx, pos)
        orc.move(idx, pos)
        st, so = ts.step(), orc.step_gsdd()
        print(name, "GSDD step", k, "sweeps", st.iters, so.iters, "halvings", st.ls_halvings, so.ls_halvings)
        assert (st.status, st.iters, st.ls_halvings) == (so.status, so.iters, so.ls_halvings), k
        assert st.g2 <= ts.targetGRes and abs(st.E - so.E) <= 1e-10 * abs(so.E)
        assert st.energy_evals == so.energy_evals
    assert np.abs(ts.getResult() - orc.state()[0]).max() < 1e-9
    ts.close(); orc.close()


def test_gsdd_with_long_row_subdomains():
    """`timeStepper GSDD 6` is what the reference ships (input/otherMethods/monkey18K_TSS_GSDD_E2.5e4.txt:2): six
    subdomains of ~10 k DOFs, whose separator rows are longer than the register tile of the single-pass back-solve.  The
    per-subdomain solve of the GSDD sweep then runs the part's long-row work items through the two-phase kernel.  On
    bar17K / 6 (METIS fixture bar17K_6, n_s up to 9813): one step against the oracle."""
    sc, ep, n, ts, orc = make_pair("bar17K_twist", energy="FCR", nparts=6, flags=dl.FLAG_GSDD)
    x = ts.getResult()
    idx, pos = sc.scripter.step(x, sc.cfg.dt)
    ts.setDirichlet(idx, pos)
    orc.move(idx, pos)
    st, so = ts.step(), orc.step_gsdd()
    print("bar17K/6 GSDD sweeps", st.iters, so.iters, "halvings", st.ls_halvings, so.ls_halvings)
    assert (st.status, st.iters, st.ls_halvings, st.energy_evals) == (so.status, so.iters, so.ls_halvings, so.energy_evals)
    assert st.g2 <= ts.targetGRes and abs(st.E - so.E) <= 1e-10 * abs(so.E)
    assert np.abs(ts.getResult() - orc.state()[0]).max() < 1e-9
    ts.close(); orc.close()


# ---- f4: LBFGS-H (`timeStepper LBFGSH`) = this path with the whole mesh as ONE subdomain and a unit first step --------
def test_lbfgs_h_is_the_one_subdomain_case_with_unit_first_step():
    """LBFGSTimeStepper with D0T_H (LBFGSTimeStepper.cpp:196-262 precompute, :338-420 solve_oneStep, :300-307 refresh):
    the same two-loop recursion with the factored GLOBAL projected Hessian as initial inverse Hessian, refreshed at the
    end of every step, and a line search that starts from step 1 (Optimizer::initStepSize only estimates alpha_0 for
    TST_DOT, Optimizer.cpp:1076-1093).  Here: nParts = 1 (no interface, dup = 1) and alphaMin = 1, which makes
    clamp(-p.g / p.Hp, alphaMin, 1) the constant 1.  Against the oracle configured the same way; and the first
    direction of a step is the Newton direction of the projected Hessian: H p = -g."""
    sc, _, _ = load_workload("synbar:16x5x5:1")
    cfg = sc.cfg
    ep = np.zeros(sc.T.shape[0], dtype=np.int32)
    ts = DOTTimeStepper(sc, ep, 1, alpha_min=1.0)
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, 1,
                      cfg.with_gravity)
    orc.set_alpha_min(1.0)
    for k in range(4):
        (st, so), = run_both(sc, ts, orc, 1)
        assert (st.status, st.iters, st.ls_halvings) == (so.status, so.iters, so.ls_halvings), k
        a, _, _ = ts.iterLog()
        assert np.all(a == 2.0 ** -np.round(-np.log2(a)))        # every accepted step is 1, 1/2, 1/4, ...
        assert a[0] == 1.0 or st.ls_halvings > 0
    assert np.abs(ts.getResult() - orc.state()[0]).max() < 1e-9
    x = ts.getResult()
    pr = ts.probeDirection(x)
    Hp = ts.multiply(pr["p"])
    assert np.abs(Hp + pr["g"]).max() <= 1e-9 * np.abs(pr["g"]).max()
    assert pr["alpha0"] == 1.0
    ts.close(); orc.close()


def test_projected_newton_matches_oracle():
    """`timeStepper Newton` (base Optimizer::fullyImplicit + solve_oneStep, Optimizer.cpp:654-749): per iteration a
    fresh projected Hessian, factorisation, H p = -g, line search from 1.  DOTMI_FLAG_NEWTON with the mesh as one
    subdomain vs dor_step_newton; Newton converges in a handful of iterations where L-BFGS-H takes more."""
    sc, _, _ = load_workload("synbar:16x5x5:1")
    cfg = sc.cfg
    ep = np.zeros(sc.T.shape[0], dtype=np.int32)
    ts = DOTTimeStepper(sc, ep, 1, flags=dl.FLAG_NEWTON)
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, 1,
                      cfg.with_gravity)
    for k in range(3):
        x = ts.getResult()
        idx, pos = sc.scripter.step(x, cfg.dt)
        ts.setDirichlet(idx, pos)
        orc.move(idx, pos)
        st, so = ts.step(), orc.step_newton()
        print("Newton step", k, "iterations", st.iters, so.iters, "halvings", st.ls_halvings, so.ls_halvings)
        assert (st.status, st.iters, st.ls_halvings) == (so.status, so.iters, so.ls_halvings), k
        assert st.g2 <= ts.targetGRes and st.iters <= 8
    assert np.abs(ts.getResult() - orc.state()[0]).max() < 1e-9
    ts.close(); orc.close()


def test_lbfgs_jh_block_jacobi_on_a_vertex_partition():
    """LBFGS-JH (`timeStepper LBFGSJH <n>`, LBFGSTimeStepper with D0T_JH: node lists from METIS::partMesh_nodes
    LBFGSTimeStepper.cpp:70-90, per-block factors of the global Hessian's principal sub-matrices :240-262 / :309-331,
    block-Jacobi solve :381-393, unit first step): dotmi_mesh.vpart = the reference's METIS vertex partition (fixture
    bunny5K_8_nodes) + alphaMin = 1.  Subdomains are disjoint (dup = 1 everywhere), so the block solve of a vector
    supported on one block is that block's solve and zero elsewhere.  Against the oracle configured the same way."""
    import os
    from tests.workloads import PART_DIR
    sc, ep, n = load_workload("bunny5K_LTSS")
    cfg = sc.cfg
    vp = np.load(os.path.join(PART_DIR, "bunny5K_8_nodes.npy")).astype(np.int32)
    assert vp.size == sc.V_rest.shape[0] and np.bincount(vp).min() > 500
    ts = DOTTimeStepper(sc, ep, n, alpha_min=1.0, vpart=vp)
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n,
                      cfg.with_gravity, vpart=vp)
    orc.set_alpha_min(1.0)
    assert np.all(orc.dup() == 1)
    for s_ in range(n):
        Hs, l2g = ts.partMatrix(s_)
        assert np.array_equal(l2g, np.nonzero(vp == s_)[0])
    r = np.zeros_like(sc.x0)
    blk = np.nonzero((vp == 3) & ~sc.fixed.astype(bool))[0]
    r[blk] = np.random.default_rng(2).standard_normal((blk.size, 3))
    z = ts.applyPrecond(r)
    assert not z[vp != 3].any()
    zo = orc.apply_precond(r)
    assert np.abs(z - zo).max() <= 1e-9 * np.abs(zo).max()
    for k in range(3):
        (st, so), = run_both(sc, ts, orc, 1)
        print("LBFGS-JH step", k, "iterations", st.iters, so.iters, "halvings", st.ls_halvings, so.ls_halvings)
        assert (st.status, st.iters, st.ls_halvings) == (so.status, so.iters, so.ls_halvings), k
    assert np.abs(ts.getResult() - orc.state()[0]).max() < 1e-9
    ts.close(); orc.close()


def test_random_layout_sweep_matches_oracle():
    """tools/fuzz_layouts.py in the suite: random jittered bars x partitions (some ragged) x dissection depths x materials x
    scripts, two steps each: identical iterations and halvings, positions to 1e-9."""
    import os
    rng = np.random.default_rng(20260929)
    keep = {k: os.environ.get(k) for k in ("DOTMI_ND_LEVELS", "DOTMI_ND_MIN")}
    try:
        for trial in range(8):
            nx, ny, nz = int(rng.integers(4, 14)), int(rng.integers(2, 6)), int(rng.integers(2, 6))
            nparts = int(rng.integers(2, 9))
            levels, ndmin = int(rng.integers(0, 4)), int(rng.choice([128, 192, 256, 512, 768]))
            energy = str(rng.choice(["FCR", "SNH"]))
            script = str(rng.choice(["stretch", "twist", "squash", "hang"]))
            os.environ["DOTMI_ND_LEVELS"], os.environ["DOTMI_ND_MIN"] = str(levels), str(ndmin)
            V, T = scene.synthetic_bar(nx, ny, nz, jitter=0.08)
            cfg = scene.Config(energy=energy, script=script, dt=0.02, rho=1000.0, YM=1e5, PR=0.4, handle_ratio=0.05)
            sc = scene.build_scene(cfg, V, T)
            ep = scene.partition_dual(sc.V_rest, sc.T, nparts) if trial % 2 else scene.partition_rcb(sc.V_rest, sc.T, nparts)
            if rng.random() < 0.3:                      # ragged: move a random tenth of the elements into part 0
                ep = ep.copy(); ep[rng.random(ep.size) < 0.1] = 0
            ts = DOTTimeStepper(sc, ep, nparts)
            orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, nparts)
            for k in range(2):
                x = ts.getResult()
                idx, pos = sc.scripter.step(x, cfg.dt)
                if idx.size:
                    ts.setDirichlet(idx, pos); orc.move(idx, pos)
                st, so = ts.step(), orc.step()
                what = (trial, nx, ny, nz, nparts, levels, ndmin, energy, script, k)
                assert (st.iters, st.ls_halvings) == (so.iters, so.ls_halvings), what
                assert np.abs(ts.getResult() - orc.state()[0]).max() < 1e-9, what
            ts.close(); orc.close()
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
