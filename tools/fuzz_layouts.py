"""One-off robustness sweep: random small meshes / partitions / dissection settings, two steps against the oracle."""
import os, sys, itertools
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dot_amd import scene
from dot_amd.timestepper import DOTTimeStepper
from tests import oracle_py as O
rng = np.random.default_rng(7)
bad = 0
cases = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    nx, ny, nz = int(rng.integers(4, 14)), int(rng.integers(2, 6)), int(rng.integers(2, 6))
    nparts = int(rng.integers(2, 9))
    levels, ndmin = int(rng.integers(0, 4)), int(rng.choice([128, 192, 256, 512, 768]))
    energy = str(rng.choice(["FCR", "SNH"]))
    script = str(rng.choice(["stretch", "twist", "squash", "hang"]))
    os.environ["DOTMI_ND_LEVELS"], os.environ["DOTMI_ND_MIN"] = str(levels), str(ndmin)
    V, T = scene.synthetic_bar(nx, ny, nz, jitter=0.08)
    cfg = scene.Config(energy=energy, script=script, dt=0.02, rho=1000.0, YM=1e5, PR=0.4, handle_ratio=0.05)
    sc = scene.build_scene(cfg, V, T)
    ep = scene.partition_rcb(sc.V_rest, sc.T, nparts)
    if rng.random() < 0.3:                      # ragged: move a random tenth of the elements into part 0
        ep = ep.copy(); ep[rng.random(ep.size) < 0.1] = 0
    ts = DOTTimeStepper(sc, ep, nparts)
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, nparts)
    ok = True
    for k in range(2):
        x = ts.getResult()
        idx, pos = sc.scripter.step(x, cfg.dt)
        if idx.size:
            ts.setDirichlet(idx, pos); orc.move(idx, pos)
        st, so = ts.step(), orc.step()
        dx = np.abs(ts.getResult() - orc.state()[0]).max()
        ok &= (st.iters, st.ls_halvings) == (so.iters, so.ls_halvings) and dx < 1e-9
    cases += 1
    bad += not ok
    print(f"{'ok ' if ok else 'BAD'} bar {nx}x{ny}x{nz} parts {nparts} nd {levels}/{ndmin} {energy} {script}: iters {st.iters}/{so.iters} dx {dx:.2e}", flush=True)
    ts.close(); orc.close()
print(f"{cases - bad}/{cases} cases agree")
