"""GPU parity tests added in round 2 (through the C ABI, against the CPU oracle): the BASELINE.json configurations
that round 1 left without a `-m gpu` test, ABI calls the bundled scripters never make, and the failure paths."""
import numpy as np
import pytest

from dot_amd import lib as dl
from dot_amd import scene
from dot_amd.configs import load_workload
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
