"""N>1 path on CPU: two gloo ranks shard the work exactly as dotmi_create does (contiguous groups of
subdomains, their elements, a vertex slice), compute their partial contributions with the oracle's
element-level functions, exchange with sum-all-reduces and must reproduce the single-process
result.  This covers the sharding logic and the collectives' shapes; the kernels themselves are the
-m gpu tests."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, initfile, outdir):
    sys.path.insert(0, ROOT)
    os.environ["OMP_NUM_THREADS"] = "2"
    from tests.workloads import load_workload
    from dot_amd.sharding import owned_elements, part_scalar_sizes, plan_shards, vertex_slice
    from tests import oracle_py as O

    dist.init_process_group("gloo", init_method=f"file://{initfile}", rank=rank, world_size=world)
    sc, ep, nparts = load_workload("bunny5K_LTSS")
    cfg = sc.cfg
    sim = O.OracleSim(sc.V_rest, sc.T, cfg.YM, cfg.PR, cfg.rho, cfg.energy_id, cfg.dt, sc.fixed, sc.x0, ep,
                      nparts, cfg.with_gravity)
    L = O.lib(); dp = O._dp
    A, vol, mass, mu, lam = sim.features()
    rng = np.random.default_rng(5)                        # same stream on both ranks
    x = np.ascontiguousarray(sc.x0 + 0.01 * rng.standard_normal(sc.x0.shape))
    xt = sim.state()[2]
    fx = sc.fixed.astype(bool)
    # what this rank owns comes from the LIBRARY's own plan (dotmi_plan_rank, host only: the code dotmi_create
    # runs), not from a re-derivation; the Python mirror in dot_amd/sharding.py must agree with it
    import ctypes as C
    from dot_amd import lib as dl
    Ld = dl.load()
    T32 = np.ascontiguousarray(sc.T, dtype=np.int32)
    ep32 = np.ascontiguousarray(ep, dtype=np.int32)
    own = np.zeros(T32.shape[0], dtype=np.int32)
    psz = np.zeros(nparts, dtype=np.int32)
    p0, p1, ne, v0c, v1c = (C.c_int32() for _ in range(5))
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    assert Ld.dotmi_plan_rank(sc.V_rest.shape[0], T32.shape[0], ip(T32), ip(ep32), nparts, rank, world, C.byref(p0),
                              C.byref(p1), ip(own), C.byref(ne), C.byref(v0c), C.byref(v1c), ip(psz)) == 0
    own = own[:ne.value]
    v0, v1 = v0c.value, v1c.value
    first = plan_shards(part_scalar_sizes(sc.T, ep, nparts), world)
    assert (first[rank], first[rank + 1]) == (p0.value, p1.value)
    assert np.array_equal(own, owned_elements(ep, first, rank))
    assert (v0, v1) == vertex_slice(sc.V_rest.shape[0], rank, world)
    assert np.array_equal(psz, part_scalar_sizes(sc.T, ep, nparts))
    for s_ in range(nparts):
        assert psz[s_] == 3 * sim.part_verts(s_).size

    # ---- [g ; E]: element contributions of the owned parts + inertia of the vertex slice ---------
    buf = np.zeros(3 * len(x) + 1)
    g = buf[:-1].reshape(-1, 3)
    for e in own:
        ge = np.zeros(12); pe = C.c_double()
        x4 = np.ascontiguousarray(x[sc.T[e]])
        L.dor_elem_energy_grad_x(cfg.energy_id, dp(x4), dp(np.ascontiguousarray(A[e])), mu[e], lam[e],
                                 cfg.dt ** 2 * vol[e], C.byref(pe), dp(ge))
        g[sc.T[e]] += ge.reshape(4, 3)
        buf[-1] += pe.value
    d = x[v0:v1] - xt[v0:v1]
    g[v0:v1] += mass[v0:v1, None] * d
    buf[-1] += 0.5 * (mass[v0:v1] * (d * d).sum(axis=1)).sum()
    g[fx] = 0
    t = torch.from_numpy(buf)
    dist.all_reduce(t)                                    # ONE collective carries g and E
    # ---- p: back-solves of the owned parts, summed, then averaged by dup --------------------------
    r = rng.standard_normal(x.shape); r[fx] = 0
    p = np.zeros_like(r)
    for s in range(first[rank], first[rank + 1]):
        l2g = sim.part_verts(s)
        p[l2g] += np.linalg.solve(sim.part_dense(s), r[l2g].ravel()).reshape(-1, 3)
    tp = torch.from_numpy(p)
    dist.all_reduce(tp)
    p /= np.maximum(sim.dup(), 1)[:, None]
    # ---- alpha_0 scalars: rows of the vertex slice ---------------------------------------------------
    Hp = sim.spmv(p)
    sc2 = torch.tensor([(p[v0:v1] * g[v0:v1]).sum(), (p[v0:v1] * Hp[v0:v1]).sum()], dtype=torch.float64)
    dist.all_reduce(sc2)
    if rank == 0:
        gref = sim.gradient(x); Eref = sim.energy(x); pref = sim.apply_precond(r)
        np.savez(os.path.join(outdir, "out.npz"),
                 g_err=np.abs(g - gref).max() / np.abs(gref).max(), E_err=abs(buf[-1] - Eref) / abs(Eref),
                 p_err=np.abs(p - pref).max() / np.abs(pref).max(),
                 pg_err=abs(sc2[0].item() - (p * gref).sum()) / abs((p * gref).sum()),
                 pHp_err=abs(sc2[1].item() - (p * Hp).sum()) / abs((p * Hp).sum()),
                 nown=len(own), first=np.array(first))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo_sharding_reproduces_single_process():
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.start_processes(_worker, args=(2, initfile, d), nprocs=2, join=True, start_method="spawn")
        out = np.load(os.path.join(d, "out.npz"))
        assert out["g_err"] < 1e-12 and out["E_err"] < 1e-13
        assert out["p_err"] < 1e-10
        assert out["pg_err"] < 1e-10 and out["pHp_err"] < 1e-10
        assert 0 < out["nown"] < 19379 and list(out["first"]) == [0, 4, 8]
