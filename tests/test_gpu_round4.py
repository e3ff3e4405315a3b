"""Round 4: failure handling of the asynchronous refresh, parity at the bench's stand-in sizes, the kernels priced in the
form the loop runs them -- through the C ABI on the GPU."""
import os
import sys

import numpy as np
import pytest

from dot_amd import lib as dl
from dot_amd.timestepper import DOTTimeStepper
from tests import oracle_py as O
from tests.workloads import load_workload

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _async_failure_worker(q):
    """runs in a process of its own: the fault-injection build is chosen at import time (DOTMI_LIBRARY)"""
    os.environ["DOTMI_LIBRARY"] = os.path.join(ROOT, "dot_amd", "libdotmi_testhooks.so")
    os.environ["DOTMI_TEST_FAIL_REFRESH"] = "3"     # 1 = the one in dotmi_create, 2 = end of step 0, 3 = end of step 1
    sys.path.insert(0, ROOT)
    from dot_amd import lib as dl_
    from dot_amd.timestepper import DOTTimeStepper as TS
    from tests.workloads import load_workload as lw
    out = {}
    try:
        sc, ep, n = lw("bunny5K_LTSS")
        ts = TS(sc, ep, n, flags=dl_.FLAG_ASYNC_REFRESH)
        sc.scripter.track(sc.x0)                       # the scripter keeps the handle positions itself: nothing reads back
        for k in range(2):
            idx, pos = sc.scripter.step(None, sc.cfg.dt)
            ts.setDirichlet(idx, pos)
            out[f"status{k}"] = ts.step().status       # step 1 returns with its (failing) refresh still queued
        idx, pos = sc.scripter.step(None, sc.cfg.dt)
        ts.setDirichlet(idx, pos)                      # does not wait for the refresh
        for name, call in (("step", ts.step), ("precond", lambda: ts.applyPrecond(np.ones_like(sc.x0)))):
            try:
                call()
                out[name] = "no error"
            except dl_.DotmiError as e:
                out[name] = str(e)
        x, v, _ = ts.getState()
        out["finite"] = bool(np.isfinite(x).all() and np.isfinite(v).all())
        ts.updatePrecondMtrAndFactorize(ts.getResult())     # a good factorisation heals the handle
        out["healed"] = ts.step().status
        ts.close()
    except Exception as e:   # noqa: BLE001
        out["exception"] = repr(e)
    q.put(out)


def test_a_failed_asynchronous_refresh_stops_the_next_call():
    """with DOTMI_FLAG_ASYNC_REFRESH a step returns before its refresh has been judged.  The next dotmi_step (and
    dotmi_apply_precond) must wait for that verdict BEFORE enqueuing anything: no L-BFGS loop on garbage factors, the state
    stays that of the completed step, and the error is the synchronous path's DOTMI_E_NOTSPD (-3)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_async_failure_worker, args=(q,))
    p.start()
    out = q.get(timeout=300)
    p.join(timeout=60)
    assert "exception" not in out, out
    assert out["status0"] == 0 and out["status1"] == 0
    assert "-3" in out["step"] and "-3" in out["precond"], out
    assert out["finite"] and out["healed"] == 0
