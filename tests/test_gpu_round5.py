"""Round 5: the 1 M-tet configuration against a full-size oracle fixture, the fused loop launches against the unfused ones,
the macro-tile factorisation against the 64-tile schedule -- through the C ABI on the GPU."""
import os

import numpy as np
import pytest

from dot_amd import lib as dl
from dot_amd.timestepper import DOTTimeStepper
from tests.workloads import load_workload

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_synbar_1M_tets_full_size_matches_the_oracle_fixture():
    """BASELINE.json configs[4] at FULL size (140 x 35 x 35 cubes = 1 029 000 tets, 182 736 vertices, 256 subdomains) against
    the CPU oracle itself: tests/golden/synbar_1M_oracle.npz holds what oracle/dot_oracle.c (subdomain linear algebra in the
    reference's CHOLMODSolver) produced for the first time steps in the build container (tools/make_synbar_golden.py) --
    iterations, halvings, energy evaluations, (E0, |g|^2_0), the per-iteration (alpha, E, |g|^2) log and the positions of a
    fixed 4 096-vertex sample.  The HIP path must take the same iterations and land on the same positions to 1e-9
    (configs[4] moves from property checks to oracle parity)."""
    G = np.load(os.path.join(GOLD, "synbar_1M_oracle.npz"))
    name = str(G["workload"])
    sc, ep, n = load_workload(name)
    assert (sc.V_rest.shape[0], sc.T.shape[0], n) == (int(G["nV"]), int(G["nT"]), int(G["nparts"])) == (182736, 1029000, 256)
    cfg = sc.cfg
    ts = DOTTimeStepper(sc, ep, n)
    try:
        assert abs(ts.targetGRes - float(G["target_gres"])) <= 1e-15 * float(G["target_gres"])
        sample = G["sample"]
        for k in range(int(G["steps"])):
            x = ts.getResult()
            idx, pos = sc.scripter.step(x, cfg.dt)
            ts.setDirichlet(idx, pos)
            st = ts.step()
            assert st.status == int(G[f"status{k}"]) == 0
            assert (st.iters, st.ls_halvings, st.energy_evals) == (int(G[f"iters{k}"]), int(G[f"halvings{k}"]), int(G[f"evals{k}"])), k
            # (positions agree to 1e-9 on a unit box, so the energies built on them agree to ~1e-9 relative from step 1 on; the
            # residual norm near convergence is the most sensitive number of the log)
            a, e, g2 = ts.iterLog()
            dx = np.abs(ts.getResult()[sample] - G[f"x{k}"]).max()
            print(f"synbar 1M step {k}: {st.iters} iterations, max|dx| on the sample vs the oracle fixture {dx:.2e}, "
                  f"dE0 {abs(st.E0 - float(G[f'E0_{k}'])) / abs(float(G[f'E0_{k}'])):.1e}, "
                  f"dE {np.abs(e / G[f'Elog{k}'] - 1).max():.1e}, dalpha {np.abs(a / G[f'alpha{k}'] - 1).max():.1e}, "
                  f"dg2 {np.abs(g2 / G[f'g2log{k}'] - 1).max():.1e}")
            assert abs(st.E0 - float(G[f"E0_{k}"])) <= 1e-9 * abs(float(G[f"E0_{k}"]))
            assert abs(st.g2_0 - float(G[f"g20_{k}"])) <= 1e-6 * float(G[f"g20_{k}"])
            assert len(e) == st.iters
            assert np.allclose(a, G[f"alpha{k}"], rtol=1e-5, atol=0)
            assert np.allclose(e, G[f"Elog{k}"], rtol=1e-9, atol=0)
            assert np.allclose(g2, G[f"g2log{k}"], rtol=1e-3, atol=0)
            assert dx < 1e-9, (k, dx)
    finally:
        ts.close()


@pytest.mark.parametrize("workload", ["bar17K_twist", "monkey18K_stiff", "synbar:40x10x10:256"])
def test_tile_packs_and_short_long_row_tiles_apply_the_same_preconditioner(workload):
    """Round 5's back-solve tiling -- tiles of at most 256 columns four to a workgroup, one wavefront each
    (backsolve_wave_tile, <1, 32> up to 128 columns, <2, 16> above), long rows of a shallow launch in 4-pass tiles -- against
    the one-tile-per-workgroup, 64-row tiling of round 4 on the same factors: p = M r agrees to rounding for random right-hand
    sides (the two sum every row's dot product and every column's updates in another order, nothing else differs), X^T X H_s = I
    still holds, and both agree with the CPU oracle's block solve.  synbar:40x10x10:256 has ONLY small tiles (165 dofs per
    subdomain): its narrow launch consists of packs alone."""
    from tests import oracle_py as O
    sc, ep, n = load_workload(workload)
    cfg = sc.cfg
    rng = np.random.default_rng(5)
    r = rng.standard_normal((sc.V_rest.shape[0], 3)) * (1 - sc.fixed[:, None])
    out = {}
    for name, env in (("packs", {}), ("plain", {"DOTMI_WAVE_PACKS": "0", "DOTMI_TILE_PASSES": "8"})):
        os.environ.update(env)
        try:
            ts = DOTTimeStepper(sc, ep, n)
        finally:
            for k in env:
                del os.environ[k]
        out[name] = ts.applyPrecond(r)
        if name == "packs":
            Hs, _ = ts.partMatrix(0)
            X, _ = ts.partMatrix(0, inverse=True)
            assert np.abs(X.T @ X @ Hs - np.eye(Hs.shape[0])).max() < 1e-9
        ts.close()
    scale = np.abs(out["plain"]).max()
    assert np.abs(out["packs"] - out["plain"]).max() <= 1e-12 * scale
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n, cfg.with_gravity)
    po = orc.apply_precond(r)
    orc.close()
    assert np.abs(out["packs"] - po).max() <= 1e-9 * np.abs(po).max()


# ---- paired line-search trials (DOTMI_PAIR_TRIALS) ---------------------------------------------------------
@pytest.mark.parametrize("workload,steps", [("monkey18K_stiff", 2), ("bunny5K_LTSS", 14), ("horse7K_stretch", 8)])
def test_paired_trials_take_the_same_steps_bit_for_bit(workload, steps, monkeypatch):
    """A paired slot evaluates the half step in full and the ENERGY of the full step in one launch when alpha_0 < 1; the controller
    then decides as the reference's line search would have (Optimizer.cpp:806-833: the full step first).  Whatever the rule pairs
    or redoes, the trajectory is the unpaired one bit for bit: iterations, halvings, energy evaluations, energies, positions."""
    import numpy as np
    from tests.workloads import load_workload
    from dot_amd.timestepper import DOTTimeStepper

    def run(mode):
        monkeypatch.setenv("DOTMI_PAIR_TRIALS", mode)
        # (on the element patches, all three: by default a step of the stiff monkey -- Stable Neo-Hookean -- that does not pair takes
        # the vertex patches and a paired one the element patches, so the form is pinned; pairing ON vertex patches against never
        # paired there: tests/test_gpu_round6.py)
        monkeypatch.setenv("DOTMI_VERTEX_PATCHES", "0")
        sc, ep, n = load_workload(workload)
        ts = DOTTimeStepper(sc, ep, n)
        rec, paired, redone, stopped = [], 0, 0, []
        for _ in range(steps):
            x = ts.getResult()
            idx, pos = sc.scripter.step(x, sc.cfg.dt)
            ts.setDirichlet(idx, pos)
            st = ts.step()
            rec.append((st.iters, st.ls_halvings, st.energy_evals, st.E))
            paired += st.paired_slots
            redone += st.paired_redone
            stopped.append((st.backsolve_stopped, st.backsolve_launches))
        x = ts.getResult().copy()
        ts.close()
        return rec, x, paired, redone, stopped

    rec0, x0, p0, r0, s0 = run("0")
    rec1, x1, p1, r1, s1 = run("1")
    assert p0 == 0 and r0 == 0
    assert rec1 == rec0
    assert np.array_equal(x1, x0)
    halvings = sum(r[1] for r in rec0)
    print(f"{workload}: {sum(r[0] for r in rec0)} iterations, {halvings} halvings; paired slots {p1}, redone {r1}")
    if workload == "monkey18K_stiff":
        assert p1 >= 20 and r1 <= p1 // 3       # the rule finds its trials there, and is right about most of them
    # a stopped launch per rejected trial that had a slot of its own, one per redone slot, one at the end of the step
    for (st1, ln1), (it, hv, _, _) in zip(s1, rec1):
        assert ln1 == it and 1 <= st1 <= hv + 1 + r1
