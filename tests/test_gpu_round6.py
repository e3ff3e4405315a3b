"""Round 6: the speculative unit-step launch (k_dirstep.hip) against the unspeculated loop -- through the C ABI on the GPU."""
import os

import numpy as np
import pytest

from dot_amd import lib as dl
from dot_amd.timestepper import DOTTimeStepper
from tests.workloads import load_workload

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _run(workload, steps, env, monkeypatch, log=False):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sc, ep, n = load_workload(workload)
    ts = DOTTimeStepper(sc, ep, n)
    rec, spec, redone, stopped, logs = [], 0, 0, [], []
    for _ in range(steps):
        x = ts.getResult()
        idx, pos = sc.scripter.step(x, sc.cfg.dt)
        ts.setDirichlet(idx, pos)
        st = ts.step()
        rec.append((st.status, st.iters, st.ls_halvings, st.energy_evals, st.E, st.g2, st.E0, st.g2_0))
        spec += st.spec_slots
        redone += st.spec_redone
        stopped.append((st.backsolve_stopped, st.backsolve_launches, st.spec_redone))
        if log:
            logs.append(tuple(np.array(v).copy() for v in ts.iterLog()))
    x = ts.getResult().copy()
    v = ts.getState()[1].copy()
    ts.close()
    for k in env:
        monkeypatch.delenv(k)
    return rec, x, v, spec, redone, stopped, logs


# bar17K: the estimate alpha_0 = clamp(-p.g / p.Hp, 0.1, 1) is the unit step in every iteration (tools/linesearch_stats.py: 192 of
# 192) -- every speculation holds; bunny5K: most; the stiff monkey: alpha_0 < 1 in three iterations of ten and more than a
# halving per iteration -- redone slots, retries of redone slots, held back-solves; horse7K back-tracks from step 4 on
@pytest.mark.parametrize("workload,steps", [("bar17K_twist", 4), ("bunny5K_LTSS", 14), ("monkey18K_stiff", 2), ("horse7K_stretch", 8)])
def test_speculative_unit_step_takes_the_same_steps_bit_for_bit(workload, steps, monkeypatch):
    """A speculating slot (DOTMI_SPEC_STEP, k_dirstep.hip) evaluates x + 1 p in the launch that computes p -- the element
    workgroups form p_v = z_v + sum_j delta_j s_j[v] themselves -- and the controller checks alpha_0 afterwards: when it is not 1
    the slot is redone at alpha_0 as the first trial it is (Optimizer.cpp:1076-1093, :806-833).  Whatever is speculated or redone,
    the trajectory is the unspeculated loop's bit for bit: status, iterations, halvings, energy evaluations, energies, residuals,
    the per-iteration log and the positions."""
    # (a speculating step works on the patch set that keeps its launch resident at once -- 512-element patches on bar17K and the
    # monkey, where the default 256-element ones are too many: the comparison fixes that size for both runs, so that the energy
    # partials are grouped alike)
    pe = {"DOTMI_PATCH_ELEMS": "512"} if workload in ("bar17K_twist", "monkey18K_stiff") else {}
    pe["DOTMI_VERTEX_PATCHES"] = "0"   # (the unspeculated run on the element patches as well, not on vertex patches)
    rec0, x0, v0, s0, r0, st0, log0 = _run(workload, steps, {"DOTMI_SPEC_STEP": "0", "DOTMI_PAIR_TRIALS": "0", **pe}, monkeypatch, log=True)
    rec1, x1, v1, s1, r1, st1, log1 = _run(workload, steps, {"DOTMI_SPEC_STEP": "1", "DOTMI_PAIR_TRIALS": "0", **pe}, monkeypatch, log=True)
    assert s0 == 0 and r0 == 0
    iters = sum(r[1] for r in rec0)
    print(f"{workload}: {iters} iterations, {sum(r[2] for r in rec0)} halvings; speculating slots {s1}, redone {r1}")
    assert rec1 == rec0
    assert np.array_equal(x1, x0) and np.array_equal(v1, v0)
    for (a0, e0, g0), (a1, e1, g1) in zip(log0, log1):
        assert np.array_equal(a0, a1) and np.array_equal(e0, e1) and np.array_equal(g0, g1)
    # every new direction was speculated on: one slot per iteration (+ the rejected last trials of steps that end in a failure)
    assert s1 >= iters
    if workload == "bar17K_twist":
        assert r1 == 0                       # the unit estimate in every iteration
    if workload == "monkey18K_stiff":
        assert r1 >= iters // 10             # ... and far from it there: the redo path is what this case exercises
    # a stopped launch per rejected trial, one per redone slot, one at the end of the step
    for (stp, ln, rd), r in zip(st1, rec1):
        assert ln == r[1]
        if r[0] == 0:
            assert stp == r[2] + 1 + rd, (stp, r, rd)


def test_speculation_is_gated_by_the_previous_steps_unit_estimates(monkeypatch):
    """The per-step rule (DOTMI_SPEC_STEP=-1): a step speculates when at least nine in ten first trials of the step before took the
    unit estimate.  bar17K: every step after the first; the stiff monkey: none (its steps pair their trials instead).  Unset
    (the default since the launch was measured to pay nothing, profiles/r06_spec_step.txt): never."""
    rec, x, v, spec, redone, stopped, _ = _run("bar17K_twist", 3, {"DOTMI_SPEC_STEP": "-1"}, monkeypatch)
    iters = [r[1] for r in rec]
    assert spec == sum(iters[1:]) and redone == 0, (spec, redone, iters)
    rec, x, v, spec, redone, stopped, _ = _run("monkey18K_stiff", 2, {"DOTMI_SPEC_STEP": "-1"}, monkeypatch)
    assert spec == 0 and redone == 0
    rec, x, v, spec, redone, stopped, _ = _run("bar17K_twist", 2, {}, monkeypatch)
    assert spec == 0 and redone == 0


def test_the_speculative_launch_is_a_bench_kind_on_handles_that_can_speculate(monkeypatch):
    import ctypes as C
    monkeypatch.setenv("DOTMI_SPEC_STEP", "1")
    sc, ep, n = load_workload("bunny5K_LTSS")
    ts = DOTTimeStepper(sc, ep, n)
    ts.solve(1)
    ms, nb = C.c_double(), C.c_int64()
    assert dl.load().dotmi_bench_kernel(ts._h, dl.BENCH_KERNELS.index("dirstep"), 3, C.byref(ms), C.byref(nb)) == 0
    assert ms.value > 0 and nb.value > 0
    ts.close()


def test_speculative_steps_match_the_oracle(monkeypatch):
    """The speculating loop against the CPU oracle: bunny5K, 6 steps, identical iterations and halvings, positions to 1e-9
    (the bar of tests/test_gpu_parity.py::test_time_steps_match_oracle, with every step speculating)."""
    from tests import oracle_py as O
    monkeypatch.setenv("DOTMI_SPEC_STEP", "1")
    sc, ep, n = load_workload("bunny5K_LTSS")
    cfg = sc.cfg
    ts = DOTTimeStepper(sc, ep, n)
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n, cfg.with_gravity)
    try:
        spec = 0
        for k in range(6):
            x = ts.getResult()
            idx, pos = sc.scripter.step(x, cfg.dt)
            ts.setDirichlet(idx, pos)
            orc.move(idx, pos)
            st, so = ts.step(), orc.step()
            spec += st.spec_slots
            assert st.status == 0 and (st.iters, st.ls_halvings) == (so.iters, so.ls_halvings), k
            dx = np.abs(ts.getResult() - orc.state()[0]).max()
            assert dx < 1e-9, (k, dx)
        assert spec > 0
    finally:
        ts.close()
        orc.close()


# ---- tightest position pin on BASELINE.json configs[1] --------------------------------------------------
def _walk_fixture(fname):
    G = np.load(os.path.join(GOLD, fname))
    sc, ep, n = load_workload(str(G["workload"]))
    assert (sc.V_rest.shape[0], sc.T.shape[0], n) == (int(G["nV"]), int(G["nT"]), int(G["nparts"]))
    ts = DOTTimeStepper(sc, ep, n)
    out = []
    try:
        assert abs(ts.targetGRes - float(G["target_gres"])) <= 1e-15 * float(G["target_gres"])
        for k in range(int(G["steps"])):
            x = ts.getResult()
            idx, pos = sc.scripter.step(x, sc.cfg.dt)
            ts.setDirichlet(idx, pos)
            st = ts.step()
            a, e, g2 = ts.iterLog()
            out.append((st, np.array(a), np.array(e), np.array(g2), ts.getResult()[G["sample"]].copy()))
    finally:
        ts.close()
    return G, out


def test_bar17K_matches_the_oracle_on_the_references_cholmod_solver_to_1e_9():
    """configs[1] (bar17K_twist, StableNH, 32 subdomains), ten steps against tests/golden/bar17K_refcholmod_oracle.npz: the CPU
    oracle with every subdomain factorisation and solve done by the REFERENCE's own CHOLMODSolver (compiled where it lies,
    oracle/_ref/librefsolver.so; fixture written by tools/make_refpin_golden.py with REFPIN_SVD=0 in the build container).
    Identical iterations, halvings and energy evaluations in all ten steps, every accepted step length equal, energies to 1e-9,
    the sampled positions to 1e-9: the linear algebra under the north star's position tolerance is the reference's."""
    G, out = _walk_fixture("bar17K_refcholmod_oracle.npz")
    assert int(G["reference_cholmod"]) == 1 and int(G["reference_svd"]) == 0
    for k, (st, a, e, g2, xs) in enumerate(out):
        assert st.status == int(G[f"status{k}"]) == 0
        assert (st.iters, st.ls_halvings, st.energy_evals) == (int(G[f"iters{k}"]), int(G[f"halvings{k}"]), int(G[f"evals{k}"])), k
        assert np.allclose(a, G[f"alpha{k}"], rtol=1e-6, atol=0)
        assert np.allclose(e, G[f"Elog{k}"], rtol=1e-9, atol=0)
        dx = np.abs(xs - G[f"x{k}"]).max()
        print(f"bar17K step {k}: {st.iters} iterations, max|dx| on the sample vs the reference-CHOLMOD oracle {dx:.2e}")
        assert dx < 1e-9, (k, dx)


def test_bar17K_stays_within_the_stated_band_of_the_oracle_on_the_references_svd_and_cholmod():
    """The same ten steps against tests/golden/bar17K_refpin_oracle.npz: the oracle with BOTH compiled reference pieces -- every
    SVD through the reference's AVX kernel (Utils/SVD_EFTYCHIOS, librefpin.so:ref_svd) AND the reference's CHOLMODSolver.  The
    reference's SVD kernel (approximate Jacobi sweeps, reconstruction 1.5e-10, FMA-contracted by its build) decides the
    `if (L2 < 0)` branches of makePD2d at the rest state differently from any exact SVD (88 % of the rest-state element Hessians
    move), so the first step runs on another preconditioner and the runs meet again only at the minimiser, to
    the solver tolerance.  Stated FP64 tolerance of the device path against this reference arithmetic, asserted here: step 0
    identical iterations; every step within +-1 iteration (the CPU oracle with an exact SVD differs from this fixture in the same
    two steps, 20 / 25 against 19 / 26); no back-tracking on either side; positions of the sampled vertices within 2.5e-4 of a
    unit-size mesh (measured 1.2e-4 at most; the reference's own 6-against-8-subdomain discrepancy is 5e-4 at step 0), start
    energies within 2.5e-4 relative."""
    G, out = _walk_fixture("bar17K_refpin_oracle.npz")
    assert int(G["reference_cholmod"]) == 1 and int(G["reference_svd"]) == 1
    worst = 0.0
    for k, (st, a, e, g2, xs) in enumerate(out):
        assert st.status == 0 and st.ls_halvings == int(G[f"halvings{k}"]) == 0
        assert abs(st.iters - int(G[f"iters{k}"])) <= 1, (k, st.iters, int(G[f"iters{k}"]))
        if k == 0:
            assert st.iters == int(G["iters0"])
            assert abs(st.E0 - float(G["E0_0"])) <= 1e-12 * abs(float(G["E0_0"]))    # (no SVD in the energy of the first iterate)
        assert abs(st.E0 - float(G[f"E0_{k}"])) <= 2.5e-4 * abs(float(G[f"E0_{k}"]))
        dx = np.abs(xs - G[f"x{k}"]).max()
        worst = max(worst, dx)
        assert dx < 2.5e-4, (k, dx)
    print(f"bar17K vs the oracle on the reference's SVD + CHOLMOD: iterations {[o[0].iters for o in out]} against "
          f"{[int(G[f'iters{k}']) for k in range(len(out))]}, max|dx| over ten steps {worst:.2e}")


# ---- vertex patches: the trial's element pass + vertex gather in one launch (k_elemvert.hip) ------------------------------------
@pytest.mark.parametrize("workload,steps", [("bar17K_twist", 4), ("bunny5K_LTSS", 8), ("monkey18K_stiff", 1), ("horse7K_stretch", 4)])
def test_vertex_patches_take_the_element_patches_steps(workload, steps, monkeypatch):
    """Default since round 6 on one rank for Stable Neo-Hookean meshes where every patch is a workgroup of its own (with the
    fixed-corotational SVD the doubled element work costs what the saved launch gives: DOTMI_VERTEX_PATCHES=1 forces it there): a vertex patch owns its vertices and carries
    every element incident to them, so ONE launch does what elem_patch_kernel + vertex_gather_kernel did (energy, gradient summed in
    ascending element order over all incident elements, inertia, trial point, pair, statistics, -g into the right-hand sides).
    Against the two-launch path (DOTMI_VERTEX_PATCHES=0): the sums are grouped differently (no per-patch partials any more), nothing
    else -- identical iterations, halvings and energy evaluations, energies to 1e-9, positions to 1e-9 on the steps that do not
    back-track (the stiff monkey, chaotic: its identical prefix of decisions)."""
    base = {"DOTMI_PAIR_TRIALS": "0", "DOTMI_SPEC_STEP": "0"}
    rec0, x0, v0, _, _, _, log0 = _run(workload, steps, {**base, "DOTMI_VERTEX_PATCHES": "0"}, monkeypatch, log=True)
    rec1, x1, v1, _, _, _, log1 = _run(workload, steps, {**base, "DOTMI_VERTEX_PATCHES": "1"}, monkeypatch, log=True)   # (1: FCR too)
    if workload == "monkey18K_stiff":
        (a0, e0, _), (a1, e1, _) = log0[0], log1[0]
        m = min(len(a0), len(a1))
        same = np.isclose(a0[:m], a1[:m], rtol=1e-6, atol=0) & np.isclose(e0[:m], e1[:m], rtol=1e-9, atol=0)
        prefix = m if same.all() else int(np.argmin(same))
        print("monkey: identical prefix", prefix, "of", m, "iterations;", rec0, rec1)
        assert prefix >= 25 and rec0[0][0] == rec1[0][0] == 0
        return
    for k, (r0, r1) in enumerate(zip(rec0, rec1)):
        assert r0[:4] == r1[:4], (k, r0, r1)                       # status, iterations, halvings, energy evaluations
        assert abs(r0[4] - r1[4]) <= 1e-9 * abs(r0[4]), (k, r0[4], r1[4])
    quiet = all(r[2] == 0 for r in rec0)
    dx = np.abs(x1 - x0).max()
    print(f"{workload}: vertex patches vs element patches, max|dx| after {steps} steps {dx:.2e}")
    assert dx < (1e-9 if quiet else 1e-6), dx


def test_paired_trials_on_vertex_patches_take_the_same_steps_bit_for_bit(monkeypatch):
    """elem_vertex_kernel<MAT, PAIR> + loop_control_body<CTL_PAIR_VP>: the stiff monkey's steps pair their trials on vertex patches
    (the second half of a launch twice as wide leaves the energy of the full step).  Paired in every step against never paired:
    status, iterations, halvings, energy evaluations, energies and positions bit for bit (as on the element patches,
    tests/test_gpu_round5.py), and the rule does pair there."""
    def run(mode):
        monkeypatch.setenv("DOTMI_PAIR_TRIALS", mode)
        monkeypatch.setenv("DOTMI_VERTEX_PATCHES", "1")   # (forced: by default a paired step keeps the element patches)
        sc, ep, n = load_workload("monkey18K_stiff")
        ts = DOTTimeStepper(sc, ep, n)
        rec, paired, redone = [], 0, 0
        for _ in range(2):
            x = ts.getResult()
            idx, pos = sc.scripter.step(x, sc.cfg.dt)
            ts.setDirichlet(idx, pos)
            st = ts.step()
            rec.append((st.status, st.iters, st.ls_halvings, st.energy_evals, st.E, st.g2))
            paired += st.paired_slots
            redone += st.paired_redone
        x = ts.getResult().copy()
        ts.close()
        return rec, x, paired, redone
    rec0, x0, p0, r0 = run("0")
    rec1, x1, p1, r1 = run("1")
    assert p0 == 0 and p1 >= 20 and r1 <= p1 // 3
    assert rec1 == rec0
    assert np.array_equal(x1, x0)
