"""Pins the oracle (oracle/dot_oracle.c) against the REFERENCE: golden vectors produced by the
reference's own code (tests/golden/ref_vectors.npz, see make_ref_vectors.py) and, when
oracle/_ref/librefpin.so is present, the reference pieces called live."""
import os

import numpy as np
import pytest

from tests import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

dp = O._dp


def _svd(F):
    L = O.lib()
    U = np.zeros((3, 3)); S = np.zeros(3); V = np.zeros((3, 3))
    L.dor_svd3(dp(np.ascontiguousarray(F)), dp(U), dp(S), dp(V))
    return U, S, V


def test_svd_against_reference_vectors(golden):
    F, Ur, Sr, Vr = golden["svd_F"], golden["svd_U"], golden["svd_S"], golden["svd_V"]
    worst_s = worst_rec = 0.0
    for k in range(F.shape[0]):
        U, S, V = _svd(F[k])
        # conventions of the reference SVD: rotations, ordered, sign on the last value
        assert abs(np.linalg.det(U) - 1) < 1e-12 and abs(np.linalg.det(V) - 1) < 1e-12
        assert S[0] >= S[1] >= abs(S[2]) - 1e-15
        scale = max(np.abs(F[k]).max(), 1e-300)
        worst_rec = max(worst_rec, np.abs(U @ np.diag(S) @ V.T - F[k]).max() / scale)
        # the reference's own reconstruction error on the same input: its 10 approximate Jacobi
        # sweeps (Main_Kernel_Body.hpp:91) leave U, V accurate to ~1e-10 only (measured max 1.5e-10)
        ref_rec = np.abs(Ur[k] @ np.diag(Sr[k]) @ Vr[k].T - F[k]).max() / scale
        assert ref_rec < 1e-9
        # singular values are unique: compare directly (U, V are not unique for repeated values)
        worst_s = max(worst_s, np.abs(S - Sr[k]).max() / max(np.abs(Sr[k]).max(), 1.0))
    assert worst_rec < 1e-14
    # reference singular values are accurate to ~3e-12 (vs LAPACK); the oracle's to 1e-15
    assert worst_s < 1e-11, worst_s


@pytest.mark.parametrize("mat,name", [(0, "fcr"), (1, "snh")])
def test_energy_and_phat_against_reference_macros(golden, mat, name):
    L = O.lib()
    sig, mu, lam = golden["mat_sigma"], golden["mat_mu"], golden["mat_lam"]
    for k in range(sig.shape[0]):
        s = np.ascontiguousarray(sig[k])
        psi = L.dor_psi(mat, dp(s), mu[k], lam[k])
        d = np.zeros(3)
        L.dor_dpsi(mat, dp(s), mu[k], lam[k], dp(d))
        scale = max(abs(golden[f"psi_{name}"][k]), mu[k])
        assert abs(psi - golden[f"psi_{name}"][k]) <= 1e-13 * scale
        assert np.abs(d - golden[f"phat_{name}"][k]).max() <= 1e-13 * max(np.abs(golden[f"phat_{name}"][k]).max(), mu[k])


def test_psd_clamps_against_reference(golden):
    L = O.lib()
    for A, ref in zip(golden["pd3_in"], golden["pd3_out"]):
        a = np.ascontiguousarray(A.copy())
        L.dor_make_pd3(dp(a))
        assert np.abs(a - ref).max() < 1e-12 * max(np.abs(A).max(), 1.0)
    for B, ref in zip(golden["pd2_in"], golden["pd2_out"]):
        b = np.ascontiguousarray(B.copy())
        L.dor_make_pd2(dp(b))
        # closed form: must agree to rounding, including the reference's non-orthogonal projection
        assert np.abs(b - ref).max() < 1e-14 * max(np.abs(B).max(), 1.0)


def test_rest_state_b_block_quirk(golden):
    """IglUtils.hpp:271-309 does not renormalise the eigenvector: the rest-state block [[1,1],[1,1]]
    is halved as soon as rounding makes its zero eigenvalue negative.  The oracle must reproduce it."""
    L = O.lib()
    b = np.array([[1.0 - 1e-17, 1.0], [1.0, 1.0 - 1e-17]])
    L.dor_make_pd2(dp(b))
    assert np.allclose(b, golden["pd2_out"][1], rtol=0, atol=1e-15)


def test_hessian_contraction_against_reference(golden):
    """H = G M G^T: the oracle's elem_hessian path vs two reference dF_div_dx_mult passes."""
    import ctypes as C
    L = O.lib()
    # reach the static helper through dor_elem_hessian_x is not possible for arbitrary M, so check
    # the identity H = G M G^T with the G the oracle's gradient contraction defines
    for M, A, Href in zip(golden["hess_M"], golden["hess_A"], golden["hess_H"]):
        G = np.zeros((12, 9))
        for k in range(9):
            P = np.zeros(9); P[k] = 1.0
            g = np.zeros(12)
            L.dor_dFdx_mult_vec(dp(P), dp(np.ascontiguousarray(A)), dp(g))
            G[:, k] = g
        H = G @ M @ G.T
        assert np.abs(H - Href).max() < 1e-12 * np.abs(Href).max()


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_live_reference_pieces_match_golden(golden):
    """The committed vectors really are what the reference code produces (regenerates a sample)."""
    R = O.ref()
    F = np.ascontiguousarray(golden["svd_F"][:64])
    U = np.zeros_like(F); S = np.zeros((64, 3)); V = np.zeros_like(F)
    assert R.ref_svd(64, dp(F), dp(U), dp(S), dp(V)) == 0
    assert np.array_equal(S, golden["svd_S"][:64])
    b = np.ascontiguousarray(golden["pd2_in"][1].copy())
    R.ref_make_pd2(dp(b))
    assert np.array_equal(b, golden["pd2_out"][1])


@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_autoflip_svd_identity():
    """Optimizer.cpp:617-626 feeds F = I through AutoFlipSVD for the tolerance constant."""
    R = O.ref()
    F = np.eye(3)
    U = np.zeros((3, 3)); S = np.zeros(3); V = np.zeros((3, 3))
    R.ref_autoflip_svd(dp(F), dp(U), dp(S), dp(V))
    assert np.allclose(S, 1.0) and np.allclose(U @ V.T, np.eye(3))


# ---- round 2: the reference's LinSysSolver + CHOLMODSolver (a9 global assembly, a11 factorize / solve / multiply) ----
LINSYS = os.path.join(ROOT, "tests", "golden", "ref_linsys.npz")
CONFIG = os.path.join(ROOT, "tests", "golden", "ref_config.json")
REFERENCE = os.environ.get("DOT_REFERENCE", "/root/reference")


def reference_script_text(rel):
    """The text of one of the reference's input scripts without its `script` line (oracle/ref_config.cpp), read where it
    lies; the fixtures keep only what the reference's parser / writer made of it.  None without the reference checkout."""
    path = os.path.join(REFERENCE, rel)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return "".join(l for l in f.readlines() if l.split()[:1] != ["script"])


def _linsys_cases():
    import json
    G = np.load(LINSYS)
    meta = json.loads(str(G["meta"]))
    return G, meta


def _oracle_for(meta_k):
    from tests.workloads import load_workload
    sc, ep, _ = load_workload(meta_k["workload"])
    sc.cfg.energy = meta_k["energy"]
    cfg = sc.cfg
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, 1,
                      cfg.with_gravity)
    return sc, orc


@pytest.mark.parametrize("k", [0, 1, 2, 3])
def test_oracle_assembly_solve_and_spmv_match_reference_cholmod_solver(k):
    """Golden vectors made by the reference's own LinSysSolver.hpp + CHOLMODSolver.cpp on the vendored CHOLMOD
    (tests/golden/make_ref_vectors2.py): the oracle's global assembly (fixed rows, mass on the free diagonal, block
    indexing), its sparse product and -- with the whole mesh as ONE subdomain -- its factor + solve must agree."""
    G, meta = _linsys_cases()
    sc, orc = _oracle_for(meta[k])
    x, rhs, v = G[f"c{k}_x"], G[f"c{k}_rhs"], G[f"c{k}_v"]
    orc.refactor(x)
    Av = orc.spmv(v)
    assert np.abs(Av - G[f"c{k}_Av"]).max() <= 1e-13 * np.abs(G[f"c{k}_Av"]).max()
    sol = orc.apply_precond(rhs)
    assert np.abs(sol - G[f"c{k}_sol"]).max() <= 1e-10 * np.abs(G[f"c{k}_sol"]).max()
    if f"c{k}_dense" in G:
        D = G[f"c{k}_dense"]
        assert np.array_equal(D, D.T)
        Hs = orc.part_dense(0)
        assert np.abs(Hs - D).max() <= 1e-14 * np.abs(D).max()
        fixed = np.repeat(sc.fixed.astype(bool), 3)
        assert np.array_equal(D[fixed][:, fixed], np.eye(int(fixed.sum())))       # IglUtils.hpp:148-157
        assert not D[fixed][:, ~fixed].any()
    orc.close()


@pytest.mark.skipif(not O.ref_solver_available(), reason="oracle/_ref/librefsolver.so or the image's MKL not present")
def test_reference_cholmod_solver_live_against_oracle():
    """the same comparison with the compiled reference code called live on a fresh random state"""
    G, meta = _linsys_cases()
    sc, orc = _oracle_for(dict(workload="synbar:7x3x2:1", energy="SNH"))
    rng = np.random.default_rng(77)
    nV = sc.V_rest.shape[0]
    x = sc.x0 + 0.05 * rng.standard_normal(sc.x0.shape)
    orc.refactor(x)
    rhs = rng.standard_normal((nV, 3)) * (1 - sc.fixed[:, None])
    v = rng.standard_normal((nV, 3))
    sol, Av, _ = O.ref_linsys(sc.T, sc.fixed, orc.elem_hessians(x), orc.features()[2], rhs, v)
    assert np.abs(orc.spmv(v).reshape(-1) - Av).max() <= 1e-13 * np.abs(Av).max()
    assert np.abs(orc.apply_precond(rhs).reshape(-1) - sol).max() <= 1e-10 * np.abs(sol).max()
    orc.close()


# ---- round 2: the reference's Config::loadFromFile (f1 parser) -------------------------------------------------------
def _cfg_fields(cfg):
    """dot_amd.scene.Config -> the key/value strings of oracle/ref_config.cpp"""
    out = dict(energy=cfg.energy, timeStepper=cfg.time_stepper, partitionAmt=str(cfg.partition_amt),
               blockSize=str(cfg.block_size), size=cfg.size, duration=cfg.duration, dt=cfg.dt, rho=cfg.rho, YM=cfg.YM,
               PR=cfg.PR, withGravity=str(int(cfg.with_gravity)), inputShapePath=cfg.shape_path,
               warmStart=str(cfg.warm_start), handleRatio=cfg.handle_ratio, rotDeg=cfg.rot_deg,
               restart=str(int(cfg.restart)), statusPath=cfg.status_path)
    if cfg.rot_deg != 0.0:
        out["rotAxis"] = cfg.rot_axis
    return out


def _same(ours, ref):
    if isinstance(ours, float):
        return float(ref) == ours
    if isinstance(ours, tuple):
        return tuple(float(t) for t in ref.split()) == ours
    return ours == ref


def test_script_parsers_match_reference_config_on_every_input_script(tmp_path):
    """Every input/**/*.txt of the reference (62 scripts; `script` line removed, see oracle/ref_config.cpp) parsed by
    the reference's own Config::loadFromFile (golden) vs dot_amd.scene.parse_script and the C++ parse_script of
    dot_amd/host/Scene.hpp (through `dot_hip 100 <script> --dump-config`)."""
    import json
    import subprocess
    from dot_amd import scene
    with open(CONFIG) as f:
        G = json.load(f)
    assert len(G) >= 60
    exe = os.path.join(ROOT, "dot_amd", "dot_hip")
    if not os.path.isdir(os.path.join(REFERENCE, "input")):
        pytest.skip("the script texts are read from the reference checkout (not stored in the fixture)")
    nblock = nrot = 0
    for rel, rec in G.items():
        p = tmp_path / "s.txt"
        p.write_text(reference_script_text(rel))
        ref = rec["parsed"]
        assert ref["rc"] == "0"
        ours = _cfg_fields(scene.parse_script(str(p)))
        for key, val in ours.items():
            assert _same(val, ref[key]), (rel, key, val, ref[key])
        nblock += ref["blockSize"] != "-1"
        nrot += "rotAxis" in ref
        if os.path.exists(exe):
            txt = subprocess.check_output([exe, "100", str(p), "--dump-config"]).decode()
            cpp = dict(l.partition(" ")[::2] for l in txt.splitlines())
            for key in ours:
                a, b = cpp[key], ref[key]
                if key in ("size", "duration", "dt", "rho", "YM", "PR", "handleRatio", "rotDeg"):
                    assert float(a) == float(b), (rel, key, a, b)
                elif key == "rotAxis":
                    assert [float(t) for t in a.split()] == [float(t) for t in b.split()], (rel, key)
                else:
                    assert a == b, (rel, key, a, b)
    assert nblock >= 3 and nrot >= 5     # the block-size mode and rotateModel are exercised by the fixture


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "ref_config")),
                    reason="oracle/_ref/ref_config not built")
def test_reference_config_live_tokens(tmp_path):
    """tokens no shipped script uses, parsed live by the reference and by us: tol list, restart, turnOffGravity,
    partition counts 0 / 1 (-> 4), a negative count with a block size, an unknown energy name (-> SNH); every text
    names energy and timeStepper because the reference's constructor leaves those two uninitialised"""
    from dot_amd import scene
    texts = ["energy SNH\ntimeStepper DOT 1\ntime 3 0.01\nturnOffGravity\ntol 2\n1e-3\n2e-4\n",
             "energy FCR\ntimeStepper GSDD -7 300\nrestart some/status12\nwarmStart 0\nhandleRatio 0.25\n",
             "energy NH\ntimeStepper LBFGSH 9\nstiffness 123 0.3\ndensity 7\nsize 2.5\nrotateModel 1 0 0 -30\n"]
    for t in texts:
        p = tmp_path / "s.txt"
        p.write_text(t)
        ref = O.ref_config_parse(str(p))
        cfg = scene.parse_script(str(p))
        for key, val in _cfg_fields(cfg).items():
            assert _same(val, ref[key]), (t, key, val, ref[key])
        tol = [float(x) for x in ref["tol"].split()[1:]]
        assert (cfg.tol or []) == tol


# ---- why the shipped bar17K script (FCR, `timeStepper DOT 6`) differs by +-1 from the published run --------------------
@pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_published_bar17K_as_shipped_list_needs_the_references_svd_rounding():
    """BASELINE.md section 2 lists 9 10 12 14 14 15 15 16 15 16 for the reference's own run of input/bar17K_twist_DOT.txt
    as shipped; oracle and HIP path take 8 10 12 14 14 14 15 15 16 15.  Measured cause: the SVD kernel's rounding.
    With the oracle's SVD calls routed through the reference's own compiled AVX kernel (dor_set_svd_batch ->
    librefpin.so:ref_svd) and NOTHING else changed, the published list is reproduced on all ten steps (first three
    asserted here, the other seven recorded in DESIGN.md section 7).  Mechanism: the first step's preconditioner is
    the projected Hessian of the REST state, where every B block of compute_dP_div_dF (Energy.cpp:1151-1172) has a
    zero eigenvalue and makePD2d (IglUtils.hpp:271-309) halves the block iff rounding made it negative -- three coin
    flips per tet decided by the last bits of sigma.  The two SVDs disagree on 88 % (= 1 - 2^-3) of the rest-state
    element Hessians by up to 23 %; step 0 then stops at |g|^2 / tol = 0.976 after 8 iterations (exact Jacobi) or
    needs a 9th (reference kernel)."""
    from tests.workloads import load_workload
    sc, ep, n = load_workload("bar17K_twist", 6)
    sc.cfg.energy = "FCR"
    cfg = sc.cfg

    def run(nsteps):
        sim = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n,
                          cfg.with_gravity)
        He = np.zeros((sc.T.shape[0], 144))
        O.lib().dor_eval_elem_hessians(sim.h, dp(np.ascontiguousarray(sc.x0)), dp(He))
        its, margin = [], []
        for _ in range(nsteps):
            idx, pos = sc.scripter.step(sim.state()[0], cfg.dt)
            sim.move(idx, pos)
            st = sim.step()
            its.append(st.iters); margin.append(st.g2 / sim.target_gres)
        sim.close()
        return its, margin, He

    own, m_own, He_own = run(1)
    assert own == [8] and 0.95 < m_own[0] < 1.0          # stops with 2.4 % to spare
    O.use_reference_svd(True)
    try:
        sc, ep, n = load_workload("bar17K_twist", 6)     # fresh scripter state
        sc.cfg.energy = "FCR"
        ref, m_ref, He_ref = run(3)
    finally:
        O.use_reference_svd(False)
    assert ref == [9, 10, 12]                            # BASELINE.md section 2, "as shipped"
    d = np.abs(He_own - He_ref).max(1) / np.abs(He_own).max(1)
    frac = (d > 1e-6).mean()
    assert 0.85 < frac < 0.90 and d.max() < 0.3


@pytest.mark.skipif(not (O.ref_available() and O.ref_nofma_available()), reason="oracle/_ref not built (needs /root/reference)")
def test_the_published_list_belongs_to_the_fma_contraction_of_the_reference_build():
    """Round 4: why no device mode "follows the reference's SVD sequence".  The reference's SVD kernel is
    written as separate _mm256_mul_pd / _mm256_add_pd; its CMake flags (-O3 -mavx2 -mfma) let GCC contract them into 1087
    vfmadd instructions.  The SAME kernel compiled with -ffp-contract=off -- the operation sequence as written, which is what a
    device restatement of the source would compute -- agrees with the contracted build on the bits of sigma for only about
    half of near-identity inputs, moves a fifth of the rest-state element Hessians, and takes 8 iterations on step 0 of the
    shipped bar17K script where the contracted build takes the published 9.  The published lists are therefore a property
    of one compiler's contraction choices; reproducing them on the device means mirroring those 1087 placements."""
    rng = np.random.default_rng(0)
    n = 8192
    F = np.tile(np.eye(3).ravel(), (n, 1)) + rng.standard_normal((n, 9)) * 1.1e-16
    S = {}
    for name, R in (("fma", O.ref()), ("nofma", O.ref_nofma())):
        U, Sg, V = np.zeros((n, 9)), np.zeros((n, 3)), np.zeros((n, 9))
        assert R.ref_svd(n, dp(F), dp(U), dp(Sg), dp(V)) == 0
        S[name] = Sg
        A = np.einsum("nij,nj,nkj->nik", U.reshape(n, 3, 3), Sg, V.reshape(n, 3, 3))
        assert np.abs(A - F.reshape(n, 3, 3)).max() < 1e-9          # both are SVDs of F
    same = np.all(S["fma"] == S["nofma"], axis=1).mean()
    assert 0.3 < same < 0.8, same                                    # measured 0.56
    from tests.workloads import load_workload

    def first_step(contracted):
        sc, ep, npart = load_workload("bar17K_twist", 6)
        sc.cfg.energy = "FCR"
        cfg = sc.cfg
        O.use_reference_svd(True, contracted=contracted)
        try:
            sim = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, npart,
                              cfg.with_gravity)
            He = np.zeros((sc.T.shape[0], 144))
            O.lib().dor_eval_elem_hessians(sim.h, dp(np.ascontiguousarray(sc.x0)), dp(He))
            idx, pos = sc.scripter.step(sim.state()[0], cfg.dt)
            sim.move(idx, pos)
            it = sim.step().iters
            sim.close()
        finally:
            O.use_reference_svd(False)
        return it, He

    it_fma, He_fma = first_step(True)
    it_nofma, He_nofma = first_step(False)
    assert it_fma == 9 and it_nofma == 8       # BASELINE.md section 2 publishes 9
    d = np.abs(He_fma - He_nofma).max(1) / np.abs(He_fma).max(1)
    assert 0.1 < (d > 1e-6).mean() < 0.35      # measured 0.204


# ---- round 4: the oracle stepping on the reference's own subdomain solver (bench.py's second CPU leg) -------------------
@pytest.mark.skipif(not O.ref_solver_available(), reason="oracle/_ref/librefsolver.so or the image's MKL not present")
@pytest.mark.parametrize("workload,steps", [("bunny5K_LTSS", 3), ("synbar:12x4x4:6", 3)])
def test_oracle_on_reference_cholmod_takes_the_same_steps(workload, steps):
    """dor_use_ext_solver bound to the reference's CHOLMODSolver (one persistent object per subdomain: pattern + analyze
    once, setCoeff + factorize per refresh, solve per L-BFGS iteration -- DOTTimeStepper.cpp:363-377, :406-431): the same
    iterations and halvings as the oracle's own envelope Cholesky, positions to 1e-9, and the block solve itself to 1e-10."""
    from tests.workloads import load_workload
    sc, ep, n = load_workload(workload)
    cfg = sc.cfg
    mk = lambda: O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n,
                             cfg.with_gravity)
    a, b = mk(), mk()
    assert O.use_reference_cholmod(b) == 0
    r = np.random.default_rng(5).standard_normal(sc.x0.shape) * (1 - sc.fixed[:, None])
    za, zb = a.apply_precond(r), b.apply_precond(r)
    assert np.abs(za - zb).max() <= 1e-10 * np.abs(za).max()
    for k in range(steps):
        idx, pos = sc.scripter.step(a.state()[0], cfg.dt)
        a.move(idx, pos); b.move(idx, pos)
        sa, sb = a.step(), b.step()
        assert (sa.status, sa.iters, sa.ls_halvings) == (sb.status, sb.iters, sb.ls_halvings), k
        assert np.abs(a.state()[0] - b.state()[0]).max() < 1e-9
    O.use_reference_cholmod(b, False)                 # and back
    zb = b.apply_precond(r)
    za = a.apply_precond(r)
    assert np.abs(za - zb).max() <= 1e-10 * np.abs(za).max()
    a.close(); b.close()
