"""The element patches of the element pass (dot_amd/csrc/patches.hpp) on the CPU: the host-only entry dotmi_plan_patches
returns the lists the kernel reads; the invariants it relies on are checked and its two-stage sum -- element corner
entries -> per-(patch, vertex) partials in run order -> per-vertex sum of the partials -- is replayed in numpy against a
plain scatter-add over Mesh::vFLoc (Energy.cpp:543-563)."""
import ctypes as C

import numpy as np
import pytest

from dot_amd import lib as dl
from dot_amd.scene import synthetic_bar
from dot_amd.workloads import load_workload


def plan(V, T, PE):
    L = C.CDLL(dl.LIB_PATH)
    f = L.dotmi_plan_patches
    i32, u16, dp = C.POINTER(C.c_int32), C.POINTER(C.c_uint16), C.POINTER(C.c_double)
    f.argtypes = [C.c_int32, C.c_int32, i32, dp, C.c_int32, i32, i32, i32, i32, u16, u16, i32, i32, i32, u16, i32]
    V = np.ascontiguousarray(V, dtype=np.float64); T = np.ascontiguousarray(T, dtype=np.int32)
    nP, PV, nS = C.c_int32(), C.c_int32(), C.c_int32()
    none = lambda t: C.cast(None, t)
    assert f(V.shape[0], T.shape[0], T.ctypes.data_as(i32), V.ctypes.data_as(dp), PE, C.byref(nP), C.byref(PV), C.byref(nS),
             none(i32), none(u16), none(u16), none(i32), none(i32), none(i32), none(u16), none(i32)) == 0
    nP, PV, nS = nP.value, PV.value, nS.value
    a = dict(elem=np.zeros(nP * PE, np.int32), tl=np.zeros(4 * nP * PE, np.uint16), epos=np.zeros(4 * nP * PE, np.uint16),
             pv_gid=np.zeros(nP * PV, np.int32), pv_slot=np.zeros(nP * PV, np.int32), pv_cnt=np.zeros(nP, np.int32),
             c_ptr=np.zeros(nP * (PV + 1), np.uint16), pp_rng=np.zeros(2 * V.shape[0], np.int32))
    x, y, z = C.c_int32(), C.c_int32(), C.c_int32()
    assert f(V.shape[0], T.shape[0], T.ctypes.data_as(i32), V.ctypes.data_as(dp), PE, C.byref(x), C.byref(y), C.byref(z),
             a["elem"].ctypes.data_as(i32), a["tl"].ctypes.data_as(u16), a["epos"].ctypes.data_as(u16),
             a["pv_gid"].ctypes.data_as(i32), a["pv_slot"].ctypes.data_as(i32), a["pv_cnt"].ctypes.data_as(i32),
             a["c_ptr"].ctypes.data_as(u16), a["pp_rng"].ctypes.data_as(i32)) == 0
    return nP, PV, nS, a


@pytest.mark.parametrize("mesh,PE", [("synbar", 256), ("bunny", 256), ("bunny", 512)])
def test_patch_lists_and_two_stage_sum(mesh, PE):
    if mesh == "synbar":
        V, T = synthetic_bar(13, 5, 4)
    else:
        sc, _, _ = load_workload("bunny5K_LTSS")
        V, T = sc.V_rest, sc.T
    nV, nT = V.shape[0], T.shape[0]
    nP, PV, nS, a = plan(V, T, PE)
    elem = a["elem"].reshape(nP, PE)
    tl, epos = a["tl"].reshape(nP, PE, 4), a["epos"].reshape(nP, PE, 4)
    gid, slot = a["pv_gid"].reshape(nP, PV), a["pv_slot"].reshape(nP, PV)
    cptr, rng_ = a["c_ptr"].reshape(nP, PV + 1), a["pp_rng"].reshape(nV, 2)
    # every element sits in exactly one slot; padding slots only at the end of the last patch
    live = elem >= 0
    assert np.array_equal(np.sort(elem[live]), np.arange(nT)) and nP == -(-nT // PE)
    assert live[:-1].all() and np.all(np.diff(live[-1].astype(int)) <= 0)
    for p in range(nP):
        nv = a["pv_cnt"][p]
        assert nv <= PV and np.all(np.diff(gid[p, :nv]) > 0) and np.all(gid[p, nv:] == -1)
        # local corner indices name the right vertices; slots ascend in element id
        e = elem[p][live[p]]
        assert np.all(np.diff(e) > 0)
        assert np.array_equal(gid[p][tl[p][live[p]].astype(int)], T[e])
        assert np.all(tl[p][~live[p]] == 0xFFFF)
        # corner runs: ranges of c_ptr, every position used exactly once, runs ascending in element id
        assert cptr[p, 0] == 0 and cptr[p, nv] == 4 * len(e) and np.all(np.diff(cptr[p, :nv + 1].astype(int)) >= 1)
        pos = epos[p][live[p]].astype(int)
        assert np.array_equal(np.sort(pos.ravel()), np.arange(4 * len(e)))
        lv = tl[p][live[p]].astype(int)
        assert np.all(pos >= cptr[p][lv]) and np.all(pos < cptr[p][lv + 1])
        owner = np.empty(4 * len(e), dtype=np.int64); owner[pos.ravel()] = np.repeat(e, 4)
        for v in range(nv):
            assert np.all(np.diff(owner[cptr[p, v]:cptr[p, v + 1]]) >= 0)
    # a vertex's partial slots are a contiguous range, one per patch that touches it, in ascending patch order
    cnt = np.zeros(nV, dtype=np.int64)
    seen = {}
    for p in range(nP):
        nv = a["pv_cnt"][p]
        for v, s in zip(gid[p, :nv], slot[p, :nv]):
            assert rng_[v, 0] <= s < rng_[v, 1] and s == rng_[v, 0] + cnt[v]
            cnt[v] += 1
            assert s not in seen
            seen[s] = p
    assert np.array_equal(cnt, rng_[:, 1] - rng_[:, 0]) and len(seen) == nS == int(cnt.sum())
    assert cnt.mean() < 4.0        # a handful of partials per vertex instead of ~22 incident corners
    # replay: random 12-vectors per element
    g = np.random.default_rng(0).standard_normal((nT, 4, 3))
    ref = np.zeros((nV, 3))
    np.add.at(ref, T.ravel(), g.reshape(-1, 3))
    gpart = np.zeros((nS, 3))
    for p in range(nP):
        nv = a["pv_cnt"][p]
        e = elem[p][live[p]]
        runs = np.zeros((4 * len(e), 3))
        runs[epos[p][live[p]].astype(int).ravel()] = g[e].reshape(-1, 3)
        for v in range(nv):
            s = np.zeros(3)
            for k in range(cptr[p, v], cptr[p, v + 1]):
                s += runs[k]
            gpart[slot[p, v]] = s
    out = np.array([gpart[rng_[v, 0]:rng_[v, 1]].sum(0) for v in range(nV)])
    assert np.abs(out - ref).max() < 1e-12


def test_vertex_patch_plan_owns_every_vertex_once_and_counts_every_element_once():
    """dotmi_plan_vpatches (host only): every vertex is owned by exactly one patch, a patch carries every element incident to its
    vertices (so its runs are complete: as many entries as incident corners), every element's energy is counted exactly once,
    element slots ascend in element id inside a patch."""
    import ctypes as C
    sc, ep, n = load_workload("bunny5K_LTSS")
    L = dl.load()
    T = np.ascontiguousarray(sc.T, dtype=np.int32)
    X = np.ascontiguousarray(sc.V_rest, dtype=np.float64)
    nV, nT = X.shape[0], T.shape[0]
    hdr = np.zeros(5, dtype=np.int32)
    L.dotmi_plan_vpatches.argtypes = [C.c_int32, C.c_int32, dl.c_ip, dl.c_dp, C.c_int32, C.c_int32, dl.c_ip, dl.c_ip, dl.c_ip, dl.c_ip,
                                      dl.c_ip]
    assert L.dotmi_plan_vpatches(nV, nT, dl.ip(T), dl.dp(X), 512, 85, dl.ip(hdr), None, None, None, None) == 0
    nP, PE, PV, PO, RUN = (int(v) for v in hdr)
    assert 0 < nP <= 512 and PE == 512 and PO <= 85
    owner = np.zeros(nV, dtype=np.int32)
    elem = np.zeros(nP * PE, dtype=np.int32)
    eown = np.zeros(nP * PE, dtype=np.int32)
    runlen = np.zeros(nV, dtype=np.int32)
    assert L.dotmi_plan_vpatches(nV, nT, dl.ip(T), dl.dp(X), 512, 85, dl.ip(hdr), dl.ip(owner), dl.ip(elem), dl.ip(eown),
                                 dl.ip(runlen)) == 0
    assert owner.min() >= 0 and owner.max() == nP - 1
    inc = np.bincount(T.ravel(), minlength=nV)
    assert np.array_equal(runlen, inc)                                   # complete runs
    counted = np.zeros(nT, dtype=np.int64)
    for p in range(nP):
        el = elem[p * PE:(p + 1) * PE]
        live = el[el >= 0]
        assert np.all(np.diff(live) > 0)
        need = np.unique(np.nonzero(np.isin(T, np.nonzero(owner == p)[0]).any(axis=1))[0])
        assert np.array_equal(np.sort(live), need)                       # exactly the elements incident to the owned vertices
        counted[live[eown[p * PE:(p + 1) * PE][el >= 0] != 0]] += 1
    assert np.all(counted == 1)
