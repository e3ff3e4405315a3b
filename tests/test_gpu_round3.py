"""Round 3: the level-scheduled tile factorisation, the compact factor storage, the patch element pass and the
per-kernel bench entry -- through the C ABI on the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

from dot_amd import lib as dl
from dot_amd.timestepper import DOTTimeStepper
from tests import oracle_py as O
from tests.workloads import load_workload

pytestmark = pytest.mark.gpu


def _pair(name):
    sc, ep, n = load_workload(name)
    cfg = sc.cfg
    ts = DOTTimeStepper(sc, ep, n)
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, n,
                      cfg.with_gravity)
    return sc, ep, n, ts, orc


def test_compact_factor_storage_is_close_to_the_structural_nonzeros():
    """the factors were nParts x nmax^2 doubles (1.44 GB for 229 MB of non-zeros on bar17K).  With the
    64-row blocks of the compact layout (dotmi_internal.hpp RowTile) the storage stays within 2 x the bytes one back-solve
    streams (round 3: 1.39 x on two dissection levels; round 5's third level streams 14 % fewer bytes -- 229 -> 197 MB -- out
    of a layout whose padded regions store 13 % more -- 318 -> 360 MB: 1.83 x), and X^T X H_s = I still holds on the factors
    read back through dotmi_part_matrix."""
    sc, ep, n = load_workload("bar17K_twist")
    ts = DOTTimeStepper(sc, ep, n)
    st = ts.step()
    stored = dl.load().dotmi_factor_storage_bytes(ts._h)
    assert 0.9 * st.precond_bytes < stored < 2.0 * st.precond_bytes, (stored, st.precond_bytes)
    assert stored < 0.3 * 8 * n * dl.load().dotmi_padded_size(ts._h) ** 2      # (dense blocks would be nParts x nmax^2)
    for p in (0, 17):
        Hs, _ = ts.partMatrix(p)
        X, _ = ts.partMatrix(p, inverse=True)
        err = np.abs(X.T @ X @ Hs - np.eye(Hs.shape[0])).max()
        assert err < 1e-9, err
    ts.close()


def test_synthetic_bar_of_4M_tets_steps_on_one_gpu():
    """a synthetic bar of more than 4 M tets (224 x 56 x 56 cubes, 1024 subdomains) on ONE MI355X.
    Size-independent properties: the step converges to the reference's tolerance, the energy of the iterates never
    increases, the factors fit in a fraction of the HBM (they would have been > 110 GB as dense blocks)."""
    sc, ep, n = load_workload("synbar:224x56x56:1024")
    assert sc.T.shape[0] > 4_000_000
    ts = DOTTimeStepper(sc, ep, n)
    stored = dl.load().dotmi_factor_storage_bytes(ts._h)
    # (round 6: this mesh takes the two-level form of the block solve; its factor buffer also holds, per separator row block, the
    # leaf columns of the sub-tree as full rows -- of which only the panel rows next to a leaf are non-zero and streamed (packed))
    two_level = ts.backsolveForm() == 1
    assert stored < (24e9 if two_level else 20e9)
    for _ in range(2):
        idx, pos = sc.scripter.step(ts.getResult(), sc.cfg.dt)
        ts.setDirichlet(idx, pos)
        st = ts.step()
        assert st.status == 0 and st.g2 <= ts.targetGRes
        alpha, E, g2 = ts.iterLog()
        assert np.all(np.diff(E) <= 1e-12 * np.abs(E[:-1]))
        assert stored < (2.5 if two_level else 1.5) * st.precond_bytes
    ts.close()


def test_bench_kernel_entry_covers_every_kernel_class():
    """dotmi_bench_kernel: every kind launches, reports a positive time and the section-8(d) byte count."""
    sc, ep, n = load_workload("bunny5K_LTSS")
    ts = DOTTimeStepper(sc, ep, n)
    ts.solve(1)
    L = dl.load()
    nT, nV = sc.T.shape[0], sc.V_rest.shape[0]
    seen = {}
    for kind, name in enumerate(dl.BENCH_KERNELS):
        ms, nb = C.c_double(), C.c_int64()
        rc = L.dotmi_bench_kernel(ts._h, kind, 3, C.byref(ms), C.byref(nb))
        if name in ("dirstep", "elem_vertex") and rc < 0:
            continue    # (the speculative unit-step launch exists on handles created with DOTMI_SPEC_STEP != 0: tests/test_gpu_round6.py)
        assert rc == 0, name
        assert ms.value > 0 and nb.value > 0
        seen[name] = nb.value
    assert seen["elem_energy"] == 112 * nT + 56 * nV and seen["elem_hessian"] == (112 + 1152) * nT
    assert L.dotmi_bench_kernel(ts._h, len(dl.BENCH_KERNELS), 3, None, None) < 0
    # the state of the handle is still good for stepping
    assert ts.solve(1) == 0
    ts.close()


@pytest.mark.parametrize("env", [{"DOTMI_PATCH_ELEMS": "512"}, {"DOTMI_TILE_EAGER_MIN_RMUL": "-1"}, {"DOTMI_TILE_EAGER_MIN_RMUL": "0"},
                                 {"DOTMI_TILE_EAGER_MIN": "1000"}, {"DOTMI_TILE_EAGER_MIN": "1"}, {"DOTMI_PAIR_TRIALS": "1"},
                                 {"DOTMI_TILE_ROWS_LONG": "16"}, {"DOTMI_TILE_SPLIT": "1"}, {"DOTMI_EARLY_BACKSOLVE": "0"},
                                 {"DOTMI_EARLY_BACKSOLVE": "2"}, {"DOTMI_EARLY_BACKSOLVE": "2", "DOTMI_FUSE_STEP": "0"},
                                 {"DOTMI_EARLY_BACKSOLVE": "2", "DOTMI_FUSE_DIR": "0"},
                                 {"DOTMI_EARLY_BACKSOLVE": "2", "DOTMI_EARLY_ABORT": "0"},
                                 {"DOTMI_ND_LEVELS": "3"}, {"DOTMI_WAVE_PACKS": "0"}, {"DOTMI_TILE_PASSES": "8"},
                                 {"DOTMI_WAVE_PACKS": "0", "DOTMI_TILE_PASSES": "8", "DOTMI_ND_LEVELS": "2"}])
def test_tuning_switches_do_not_change_results(env):
    """Every tuning variable of DESIGN.md section 10 that selects another variant of a round-3 kernel / schedule: same
    iterations as the oracle, positions to 1e-9 (horse7K: FCR with SVD in the element pass, 8 subdomains, back-tracking)."""
    os.environ.update(env)
    try:
        sc, ep, n, ts, orc = _pair("horse7K_stretch")
    finally:
        for k in env:
            del os.environ[k]
    for _ in range(3):
        idx, pos = sc.scripter.step(ts.getResult(), sc.cfg.dt)
        ts.setDirichlet(idx, pos)
        orc.move(idx, pos)
        st, so = ts.step(), orc.step()
        assert (st.status, st.iters, st.ls_halvings) == (so.status, so.iters, so.ls_halvings)
    assert np.abs(ts.getResult() - orc.state()[0]).max() < 1e-9
    ts.close(); orc.close()


@pytest.mark.parametrize("workload,nparts,steps,history", [("bar17K_twist", None, 6, 5), ("bunny5K_LTSS", None, 10, 5),
                                                           ("horse7K_stretch", None, 4, 5), ("bar17K_twist", 6, 2, 5),
                                                           ("kingkong18K_SS_1K", None, 2, 5), ("bunny5K_LTSS", None, 5, 2)])
def test_early_backsolve_matches_the_q_based_loop(workload, nparts, steps, history):
    """The default device loop issues the back-solve of the next direction on the trial gradient, with the controller as
    one workgroup of that launch, and forms z = u - sum_j xi_j (M y_j) from cached M y_j (M is linear and fixed during a
    step; M y_new = u_old - u).  DOTMI_EARLY_BACKSOLVE=0 is the order of the reference (DOTTimeStepper.cpp:406-494):
    controller, q = -g - sum_j xi_j y_j, z = M q.  Same decisions in every step (iterations, back-tracking, energy
    evaluations), the per-iteration log (step and energy to 1e-9 relative, |g|^2 to 1e-6), positions to 1e-9 -- rounding only.
    Covers retries (bunny step 0 halves once), the long-row back-solve (bar17K in 6 subdomains, `DOT 6`: the launch that
    hosts the controller is not the only one) and a history shorter than the step (pairs dropped)."""
    sc, ep, n = load_workload(workload, nparts)
    try:
        os.environ["DOTMI_EARLY_BACKSOLVE"] = "2"     # in every step (the default picks per step, by the last step's counts)
        a = DOTTimeStepper(sc, ep, n, history=history)
        os.environ["DOTMI_EARLY_BACKSOLVE"] = "0"
        sc2, _, _ = load_workload(workload, nparts)
        b = DOTTimeStepper(sc2, ep, n, history=history)
    finally:
        del os.environ["DOTMI_EARLY_BACKSOLVE"]
    for k in range(steps):
        for ts, s_ in ((a, sc), (b, sc2)):
            idx, pos = s_.scripter.step(ts.getResult(), s_.cfg.dt)
            ts.setDirichlet(idx, pos)
        sa, sb = a.step(), b.step()
        assert (sa.status, sa.iters, sa.ls_halvings, sa.energy_evals) == (sb.status, sb.iters, sb.ls_halvings, sb.energy_evals), k
        # |g|^2 near convergence is a small difference of large element forces: it follows the positions' rounding
        # amplified by the stiffness, so its bar is wider
        for (u, w), rtol in zip(zip(a.iterLog(), b.iterLog()), (1e-9, 1e-9, 1e-6)):
            np.testing.assert_allclose(u, w, rtol=rtol, atol=0)
        assert np.abs(a.getResult() - b.getResult()).max() < 1e-9
    a.close(); b.close()


@pytest.mark.parametrize("workload,steps", [("bar17K_twist", 5), ("horse7K_stretch", 4)])
def test_asynchronous_refresh_changes_only_when_things_are_reported(workload, steps):
    """DOTMI_FLAG_ASYNC_REFRESH: dotmi_step returns with the end-of-step refresh (element Hessians, assembly, factorisation,
    DOTTimeStepper.cpp:349-380) still queued, so the caller's work between two steps overlaps it.  Same kernels in the same
    order on the same stream: positions bit-identical to the synchronous handle after every step, same iterations; the
    refresh's device times are reported one step later (step 0 reports none, the others the refresh that prepared their
    preconditioner); a call that reads the state in between (getResult) sees the finished step."""
    sc, ep, n = load_workload(workload)
    a = DOTTimeStepper(sc, ep, n, flags=dl.FLAG_ASYNC_REFRESH)
    sc2, _, _ = load_workload(workload)
    b = DOTTimeStepper(sc2, ep, n)
    for k in range(steps):
        # the scripted handles do not depend on the free vertices: both get the same targets without reading `a` back,
        # so `a` really runs its steps back to back behind the queued refreshes
        idx, pos = sc2.scripter.step(b.getResult(), sc2.cfg.dt)
        a.setDirichlet(idx, pos); b.setDirichlet(idx, pos)
        sa, sb = a.step(), b.step()
        assert (sa.status, sa.iters, sa.ls_halvings, sa.energy_evals) == (sb.status, sb.iters, sb.ls_halvings, sb.energy_evals)
        assert sa.E == sb.E and sa.g2 == sb.g2
        assert (sa.ms_factor == 0.0) == (k == 0) and sb.ms_factor > 0.0
        assert sa.ms_loop > 0.0
        if k == steps - 2:
            assert np.array_equal(a.getResult(), b.getResult())       # resolves the pending refresh on the way
    assert np.array_equal(a.getResult(), b.getResult())
    a.close(); b.close()


@pytest.mark.parametrize("shard_elems", ["0", "1", "owner"])
@pytest.mark.parametrize("workload,steps", [("bunny5K_LTSS", 6), ("bar17K_twist", 3)])
def test_early_order_on_the_sharded_path_matches_the_single_rank_run(workload, steps, shard_elems):
    """The early back-solve with sharded subdomains: partial merge of this rank's subdomains into a staging buffer, the
    all-reduce of that buffer, then the division by the multiplicity and the history terms (merge_tiles_early_kernel, zsum).
    shard_elems 1 (round 4): the element pass, the rows of H and the refresh are sharded too -- p.g / p.Hp go through a
    two-scalar all-reduce into the element pass that takes the step, the partial gradient through [g ; 0 ; E], and
    pair_stats writes -g into the right-hand sides and H s_new from the sum.  With the 1-rank communicator of
    DOTMI_FLAG_FORCE_DIST the sums are the single-rank ones taken in two steps: same iterations and halvings, positions to
    1e-9 (the multi-rank runs are in tests/test_gpu_two_ranks.py)."""
    # "owner" (round 4): DOTMI_FLAG_OWNER_EXCHANGE on the 1-rank RCCL communicator -- packed (here: empty) interface vectors,
    # the scalar all-reduces of 2 / 5 / 21 doubles, the partial operator of alpha_0, the positions made whole per step
    os.environ["DOTMI_SHARD_ELEMS"] = "1" if shard_elems == "owner" else shard_elems
    try:
        sc, ep, n = load_workload(workload)
        a = DOTTimeStepper(sc, ep, n, flags=dl.FLAG_FORCE_DIST | (dl.FLAG_OWNER_EXCHANGE if shard_elems == "owner" else 0))
    finally:
        del os.environ["DOTMI_SHARD_ELEMS"]
    sc2, _, _ = load_workload(workload)
    b = DOTTimeStepper(sc2, ep, n)
    for k in range(steps):
        for ts, s_ in ((a, sc), (b, sc2)):
            idx, pos = s_.scripter.step(ts.getResult(), s_.cfg.dt)
            ts.setDirichlet(idx, pos)
        sa, sb = a.step(), b.step()
        assert (sa.status, sa.iters, sa.ls_halvings) == (sb.status, sb.iters, sb.ls_halvings), k
        # both ran the early order (the single-rank run may take the unit step speculatively: a redone slot stops one more launch)
        assert sa.backsolve_stopped == sb.backsolve_stopped - sb.spec_redone == sa.ls_halvings + 1
        assert np.abs(a.getResult() - b.getResult()).max() < 1e-9
    a.close(); b.close()

