"""Round-2 golden vectors produced by the REFERENCE's own code compiled in place (make -C oracle ref):

  ref_linsys.npz   oracle/_ref/librefsolver.so = src/LinSysSolver/{LinSysSolver.hpp,CHOLMODSolver.cpp} on the vendored
                   CHOLMOD 3.0.12 (BLAS: /opt/conda/lib/libmkl_rt.so) + IglUtils::addBlockToMatrix<3>, driven by
                   oracle/ref_linsys.cpp with the statements of DOTTimeStepper::computeHElemAndFillIn
                   (DOTTimeStepper.cpp:588-613): set_pattern -> addCoeff/setCoeff -> factorize -> solve -> multiply.
                   Inputs: mesh (generator arguments), positions x, right-hand side, vector; the projected element
                   Hessians fed to the reference come from the oracle at x (their pieces are pinned by ref_vectors.npz).
                   Outputs: A^-1 rhs, A v, and for the small case the whole matrix the reference solver holds.
  ref_config.json  oracle/_ref/librefconfig.so = src/Config.cpp: Config::loadFromFile on every input/**/*.txt of the
                   reference with the `script <name>` line removed (its name table lives in AnimScripter.cpp, which
                   needs TBB).  Stored: the script text that was parsed (the reference's input data files) and the
                   fields the reference parsed out of it.

Run HERE (needs /root/reference and oracle/_ref):  python tests/golden/make_ref_vectors2.py
"""
import ctypes as C
import glob
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from tests.workloads import load_workload  # noqa: E402
from tests import oracle_py as O  # noqa: E402

REF = os.environ.get("DOT_REFERENCE", "/root/reference")


def linsys_case(name, energy, amp, seed, want_dense):
    sc, ep, _ = load_workload(name)
    sc.cfg.energy = energy
    cfg = sc.cfg
    orc = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep, 1,
                      cfg.with_gravity)
    rng = np.random.default_rng(seed)
    nV = sc.V_rest.shape[0]
    x = sc.x0 + amp * rng.standard_normal(sc.x0.shape)
    rhs = rng.standard_normal((nV, 3))
    rhs[sc.fixed.astype(bool)] = 0.0
    v = rng.standard_normal((nV, 3))
    He = orc.elem_hessians(x)
    _, _, mass, _, _ = orc.features()
    sol, Av, dense = O.ref_linsys(sc.T, sc.fixed, He, mass, rhs, v, want_dense)
    orc.close()
    out = dict(x=x, rhs=rhs, v=v, sol=sol.reshape(nV, 3), Av=Av.reshape(nV, 3))
    if want_dense:
        out["dense"] = dense
    return out


def main():
    out = {}
    # (workload, energy, perturbation amplitude, seed, store the dense matrix)
    cases = [("synbar:4x2x2:1", "FCR", 0.0, 1, True), ("synbar:4x2x2:1", "SNH", 0.08, 2, True),
             ("synbar:10x4x4:1", "SNH", 0.03, 3, False), ("synbar:10x4x4:1", "FCR", 0.002, 4, False)]
    meta = []
    for k, (name, energy, amp, seed, dense) in enumerate(cases):
        r = linsys_case(name, energy, amp, seed, dense)
        for key, val in r.items():
            out[f"c{k}_{key}"] = val
        meta.append(dict(workload=name, energy=energy, amp=amp, seed=seed))
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "ref_linsys.npz"), **out)
    print("ref_linsys.npz:", {k: getattr(v, "shape", None) for k, v in out.items()})

    cfgs = {}
    for path in sorted(glob.glob(os.path.join(REF, "input", "**", "*.txt"), recursive=True)):
        with open(path) as f:
            text = "".join(l for l in f.readlines() if l.split()[:1] != ["script"])
        with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as tf:
            tf.write(text)
        parsed = O.ref_config_parse(tf.name)
        os.remove(tf.name)
        cfgs[os.path.relpath(path, REF)] = dict(parsed=parsed)   # the text stays in the reference checkout
    with open(os.path.join(HERE, "ref_config.json"), "w") as f:
        json.dump(cfgs, f, indent=0, sort_keys=True)
    print("ref_config.json:", len(cfgs), "scripts")


if __name__ == "__main__":
    main()
