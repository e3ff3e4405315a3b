"""Round 6: the two-level form of the block solve (leaves of the dissection against the separator complement, DevTwoLevel in
dot_amd/csrc/dotmi_internal.hpp) against the one-pass explicit inverse and against the oracle.

Role in the reference: DOTTimeStepper.cpp:406-450 (the block solve of solve_oneStep), CHOLMODSolver.cpp:149-163 (solve)."""
import os

import numpy as np
import pytest

from dot_amd.timestepper import DOTTimeStepper
from dot_amd.lib import DotmiError
from dot_amd.workloads import load_workload
from tests.test_gpu_parity import make_pair

pytestmark = pytest.mark.gpu


def _with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


@pytest.mark.parametrize("workload,steps,chaotic", [("bar17K_twist", 3, False), ("bunny5K_LTSS", 4, False),
                                                    ("monkey18K_stiff", 1, True), ("kingkong18K_SS_1K", 2, False)])
def test_two_level_form_applies_the_same_preconditioner_and_takes_the_same_steps(workload, steps, chaotic):
    """DOTMI_TWO_LEVEL=1 against =0 on the same mesh: M r on a random right-hand side agrees to rounding (1e-11 of its size) and
    with the oracle's block solve to 1e-9; the time steps take the same iterations and halvings and end at the same positions
    (1e-9; Stable Neo-Hookean and fixed-corotational meshes, deep and shallow trees, few and many subdomains).  The stiff monkey's
    107 back-tracking iterations follow the factors' last bits (tests/test_gpu_parity.py: only a prefix is comparable): there both
    forms must converge, to the same energy."""
    out = {}
    for tag, v in (("one", "0"), ("two", "1")):
        def run():
            sc, ep, n = load_workload(workload)
            ts = DOTTimeStepper(sc, ep, n)
            assert ts.backsolveForm() == int(v)
            rng = np.random.default_rng(5)
            r = rng.standard_normal((sc.V_rest.shape[0], 3)) * (1 - sc.fixed[:, None])
            p = ts.applyPrecond(r)
            log = []
            for _ in range(steps):
                idx, pos = sc.scripter.step(ts.getResult(), sc.cfg.dt)
                ts.setDirichlet(idx, pos)
                st = ts.step()
                log.append((st.status, st.iters, st.ls_halvings))
            x = ts.getResult().copy()
            nbytes, E = st.precond_bytes, st.E
            ts.close()
            return r, p, log, x, nbytes, E
        out[tag] = _with_env({"DOTMI_TWO_LEVEL": v}, run)
    (r, p1, log1, x1, b1, E1), (_, p2, log2, x2, b2, E2) = out["one"], out["two"]
    assert np.abs(p1 - p2).max() <= 1e-11 * np.abs(p1).max()
    assert all(s[0] == 0 for s in log1 + log2)
    if chaotic:
        assert abs(E1 - E2) <= 1e-4 * abs(E1)
    else:
        assert log1 == log2
        assert np.abs(x1 - x2).max() < 1e-9
    assert b2 != b1      # (another count of bytes: leaves + separator complement once, panels twice)
    sc, ep, n, ts, orc = _with_env({"DOTMI_TWO_LEVEL": "1"}, lambda: make_pair(workload))
    try:
        po = orc.apply_precond(r)
        assert np.abs(ts.applyPrecond(r) - po).max() <= 1e-9 * np.abs(po).max()
    finally:
        ts.close(); orc.close()


def test_two_level_form_has_no_explicit_inverse_to_return_but_the_subdomain_matrix():
    """dotmi_part_matrix in the two-level form: H_s is returned as ever (the separators' leaf columns live in a second storage
    range), the inverse is an error with a message."""
    def run():
        sc, ep, n = load_workload("bunny5K_LTSS")
        ts = DOTTimeStepper(sc, ep, n)
        H2, l2g2 = ts.partMatrix(1)
        with pytest.raises(DotmiError, match="two-level"):
            ts.partMatrix(1, inverse=True)
        ts.close()
        return H2, l2g2
    H2, l2g2 = _with_env({"DOTMI_TWO_LEVEL": "1"}, run)
    sc, ep, n = load_workload("bunny5K_LTSS")
    ts = _with_env({"DOTMI_TWO_LEVEL": "0"}, lambda: DOTTimeStepper(sc, ep, n))
    H1, l2g1 = ts.partMatrix(1)
    ts.close()
    # (the same matrix to rounding, not to the bit: the assembled global Hessian is symmetric only to an ulp -- its (i, j) block and
    # the transpose of its (j, i) block are separate sums -- and which of the two lands in the stored triangle follows the order of
    # the two vertices in the layout; the forms' default trees differ)
    assert np.array_equal(l2g1, l2g2) and np.abs(H1 - H2).max() <= 1e-14 * np.abs(H1).max()


def test_two_level_form_is_the_default_at_1M_tets_and_reads_0p7_of_the_one_pass_bytes():
    """configs[4]: where one pass would stream 240 MB or more per application (here 3.83 GB) the factors are built in the two-level form on a four-level tree; one application
    streams 2.70 GB against the 3.83 GB of the explicit inverse on the round-5 layout (profiles/r06_two_level.txt).  Parity at
    this size: tests/test_gpu_round5.py::test_synbar_1M_tets_full_size_matches_the_oracle_fixture runs on this default."""
    sc, ep, n = load_workload("synbar:140x35x35:256")
    ts = DOTTimeStepper(sc, ep, n)
    try:
        assert ts.backsolveForm() == 1
        idx, pos = sc.scripter.step(ts.getResult(), sc.cfg.dt)
        ts.setDirichlet(idx, pos)
        st = ts.step()
        assert st.status == 0
        assert 2.5e9 < st.precond_bytes < 0.72 * 3833832864
    finally:
        ts.close()


@pytest.mark.parametrize("workload,steps,shard_elems,world", [("horse7K_stretch", 4, "0", 2), ("bar17K_twist", 2, "owner", 4),
                                                              ("bunny5K_LTSS", 3, "1", 2)])
def test_sharded_subdomains_take_the_two_level_form_too(workload, steps, shard_elems, world):
    """The form is per subdomain, so a rank's share of them can be in it: ranks as processes on one GPU (gloo, real partial
    ownership; tests/test_gpu_two_ranks.py's harness) with the form forced on -- bit-identical ranks, the single-GPU run's
    iterations and halvings, positions to 1e-9 (all-reduce, sharded element pass, owner exchange)."""
    from tests.test_gpu_two_ranks import test_ranks_on_one_gpu_reproduce_the_single_gpu_run as harness
    _with_env({"DOTMI_TWO_LEVEL": "1"}, lambda: harness(workload, steps, shard_elems, world))
