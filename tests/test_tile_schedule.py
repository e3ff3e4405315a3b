"""The level schedule of the tile factorisation (dot_amd/csrc/tile_factor.hpp) executed in numpy on the CPU: the host-only
entry dotmi_plan_tile_schedule returns the tasks of one block with their levels; running them level after level --
every task of a level reading the state left by the levels before it -- must give the inverse Cholesky factor, no two
tasks of a level may write the same tile, and no task may read a tile that another task of its level writes.
Round 5: two buffers of one layout -- offsets [0, storage) are the FACTOR buffer (only tiles of Q are ever written there),
[storage, 2 storage) the WORK buffer the fill writes H into and R overwrites; a task reads its `c` tile and writes its `o`
tile (different only for the diagonal tasks), and the last task of an off-diagonal Q tile multiplies with -Q_jj itself."""
import ctypes as C

import numpy as np
import pytest

from dot_amd import lib as dl

TF_FACT, TF_INV = 0, 1
TP_STORE, TP_DIAG, TP_ROW, TP_NEG, TP_RMUL = 0, 1, 2, 3, 4


def plan(nt, live, pat, c0, eager_min, eager_chunk):
    L = C.CDLL(dl.LIB_PATH)
    f = L.dotmi_plan_tile_schedule
    u8, i32, i64 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    f.argtypes = [C.c_int32, u8, u8, i32, C.c_int32, C.c_int32, i64, i64, i64, i64, i64, i64, i64, i64, i32]
    live = np.ascontiguousarray(live, dtype=np.uint8)
    pat = np.ascontiguousarray(pat, dtype=np.uint8)
    c0 = np.ascontiguousarray(c0, dtype=np.int32)
    n = [C.c_int64() for _ in range(5)]
    roff, rld = np.zeros(nt, dtype=np.int64), np.zeros(nt, dtype=np.int32)
    args = lambda t, p: (nt, live.ctypes.data_as(u8), pat.ctypes.data_as(u8), c0.ctypes.data_as(i32), eager_min, eager_chunk, t,
                         p, C.byref(n[0]), C.byref(n[1]), C.byref(n[2]), C.byref(n[3]), C.byref(n[4]),
                         roff.ctypes.data_as(i64), rld.ctypes.data_as(i32))
    assert f(*args(None, None)) == 0
    tasks = np.zeros((n[0].value, 11), dtype=np.int64)
    prods = np.zeros((max(n[1].value, 1), 4), dtype=np.int64)
    assert f(*args(tasks.ctypes.data_as(i64), prods.ctypes.data_as(i64))) == 0
    return tasks, prods, n[2].value, n[3].value, roff, rld


def nd_pattern(nt, leaves, seps, rng, drop=0.0):
    """upper tile pattern of a nested-dissection ordered matrix: `leaves` diagonal bands, separators coupled to their leaves"""
    pat = np.zeros((nt, nt), dtype=np.uint8)
    c0 = np.arange(nt, dtype=np.int32)
    for (a, b) in leaves:
        for i in range(a, b):
            for j in range(i, b):
                if j - i <= 2 and rng.random() >= drop:
                    pat[i, j] = 1
            pat[i, i] = 1
        c0[a:b] = a
    for (a, b, first) in seps:          # separator rows a..b couple to every tile column from `first`
        for j in range(a, b):
            for i in range(first, j + 1):
                if i >= a or rng.random() < 0.5:
                    pat[i, j] = 1
            pat[j, j] = 1
        c0[a:b] = first
    return pat, c0


def run_schedule(nt, live, pat, c0, eager_min, eager_chunk, seed):
    rng = np.random.default_rng(seed)
    tasks, prods, nlev, storage, roff, rld = plan(nt, live, pat, c0, eager_min, eager_chunk)
    n = 64 * nt
    # an SPD matrix with exactly this tile pattern (symmetric), identity on the rows of non-live tiles
    H = np.zeros((n, n))
    for i in range(nt):
        for j in range(i, nt):
            if pat[i, j] and live[i] and live[j]:
                B = rng.standard_normal((64, 64)) * 0.05
                H[64 * i:64 * i + 64, 64 * j:64 * j + 64] = B
    H = np.triu(H) + np.triu(H, 1).T
    H += np.diag(np.abs(H).sum(1) + 1.0)
    for j in range(nt):
        if not live[j]:
            H[64 * j:64 * j + 64, :] = 0
            H[:, 64 * j:64 * j + 64] = 0
            H[64 * j:64 * j + 64, 64 * j:64 * j + 64] = np.eye(64)
    # everything starts as NaN: only the tiles the fill writes into (the pattern of H, in the work buffer) are cleared and
    # filled on the device; pure fill-in tiles and every tile of the factor buffer must be written before they are read
    M = np.full(2 * storage + 4096, np.nan)

    def tile(off, ld):      # column-major 64 x 64 view
        return np.lib.stride_tricks.as_strided(M[off:], shape=(64, 64), strides=(8, 8 * ld))

    # fill: column-major element (r, c), r <= c in tile terms, lives in row block c // 64 of the WORK buffer
    for j in range(nt):
        if not live[j]:
            continue
        for i in range(int(c0[j]), j + 1):
            if pat[i, j] or i == j:
                tile(storage + roff[j] + 64 * i - 64 * c0[j], rld[j])[:, :] = H[64 * i:64 * i + 64, 64 * j:64 * j + 64]
    order = np.argsort(tasks[:, 0], kind="stable")
    lv = 0
    k = 0
    while k < len(order):
        lv = tasks[order[k], 0]
        group = []
        while k < len(order) and tasks[order[k], 0] == lv:
            group.append(order[k]); k += 1
        snap = M.copy()
        written, read = set(), set()

        def stile(off, ld):
            return np.lib.stride_tricks.as_strided(snap[off:], shape=(64, 64), strides=(8, 8 * ld)).copy()
        for t in group:
            _, form, init, post, nprod, first, coff, qoff, ldc, ldq, ooff = tasks[t]
            assert ooff < storage or post != TP_RMUL          # Q tiles go to the factor buffer ...
            assert (ooff < storage) == (post in (TP_DIAG, TP_RMUL) or form == TF_INV)   # ... and nothing else does
            acc = stile(coff, ldc) if init else np.zeros((64, 64))
            if init:
                read.add(coff)
            for p in range(first, first + nprod):
                a, b, lda, ldb = prods[p]
                A, Bm = stile(a, lda), stile(b, ldb)
                read.update((a, b))
                acc = acc - A.T @ Bm if form == TF_FACT else acc + A @ Bm
            if post == TP_DIAG:
                R = np.linalg.cholesky(acc).T          # acc = R^T R
                out = np.linalg.inv(R)                  # Q_jj (upper)
            elif post == TP_ROW:
                read.add(qoff)
                out = stile(qoff, ldq).T @ acc
            elif post == TP_RMUL:
                read.add(qoff)
                out = -acc @ np.triu(stile(qoff, ldq))
            elif post == TP_NEG:
                out = -acc
            else:
                out = acc
            assert ooff not in written, "two tasks of one level write the same tile"
            written.add(ooff)
            tile(ooff, ldc)[:, :] = out
        # a tile written in this level may only be read by the task that writes it (its own init)
        for t in group:
            _, form, init, post, nprod, first, coff, qoff, ldc, ldq, ooff = tasks[t]
            others = written - {ooff}
            ops = {prods[p][0] for p in range(first, first + nprod)} | {prods[p][1] for p in range(first, first + nprod)}
            if post in (TP_ROW, TP_RMUL):
                ops.add(qoff)
            if init:
                ops.add(coff)
            assert not (ops & others) and ooff not in (ops - {coff}), "a task reads a tile that is written in its own level"
    # compare with the dense inverse factor: H = R^T R, Q = R^-1 (upper), stored tile (i, j) = Q[64 i.., 64 j..]
    Q = np.linalg.inv(np.linalg.cholesky(H).T)
    worst = 0.0
    for j in range(nt):
        if not live[j]:
            continue
        for i in range(int(c0[j]), j + 1):
            got = tile(roff[j] + 64 * i - 64 * c0[j], rld[j])
            ref = Q[64 * i:64 * i + 64, 64 * j:64 * j + 64]
            if i == j:
                assert np.isfinite(got).all()
                worst = max(worst, np.abs(np.triu(got) - ref).max())
            elif np.abs(ref).max() > 0:
                assert np.isfinite(got).all(), "a tile of Q was never written"
                worst = max(worst, np.abs(got - ref).max())
            else:
                assert np.all(np.isnan(got)) or np.abs(got).max() < 1e-13   # outside the pattern of Q: untouched or zero
    # what lies outside the stored columns must be structurally zero in Q
    for j in range(nt):
        assert np.abs(Q[:64 * int(c0[j]), 64 * j:64 * j + 64]).max() < 1e-13 if c0[j] > 0 else True
    return worst, nlev, len(tasks)


@pytest.mark.parametrize("eager", [(1000, 1), (2, 1), (4, 4)])
def test_schedule_of_a_two_level_dissection(eager):
    """four leaves, two level-1 separators, a root separator; some padding-only tile rows"""
    rng = np.random.default_rng(1)
    nt = 14
    leaves = [(0, 3), (3, 5), (6, 8), (8, 11)]
    seps = [(5, 6, 0), (11, 12, 6), (12, 14, 0)]
    pat, c0 = nd_pattern(nt, leaves, seps, rng)
    live = np.ones(nt, dtype=np.uint8)
    live[3] = 0            # a tile row of pure identity padding inside a leaf region
    pat[3, :] = 0; pat[:, 3] = 0
    worst, nlev, ntask = run_schedule(nt, live, pat, c0, eager[0], eager[1], seed=2)
    assert worst < 1e-10, worst
    assert nlev <= 2 * nt + 2          # ~two launches per tile column on the critical path


def test_schedule_of_a_dense_block_and_of_random_patterns():
    rng = np.random.default_rng(5)
    nt = 6
    pat = np.triu(np.ones((nt, nt), dtype=np.uint8))
    worst, nlev, _ = run_schedule(nt, np.ones(nt, dtype=np.uint8), pat, np.zeros(nt, dtype=np.int32), 2, 1, seed=3)
    assert worst < 1e-10 and nlev == 2 * nt   # 2 per tile column (the last column of the inverse no longer costs a level of its own)
    for seed in range(4):
        nt = int(rng.integers(4, 10))
        pat = np.triu((rng.random((nt, nt)) < 0.35).astype(np.uint8))
        np.fill_diagonal(pat, 1)
        worst, _, _ = run_schedule(nt, np.ones(nt, dtype=np.uint8), pat, np.zeros(nt, dtype=np.int32), 3, 2, seed=10 + seed)
        assert worst < 1e-9, (seed, worst)


def plan_deps(nt, live, pat, c0, eager_min, eager_chunk, ntasks):
    L = C.CDLL(dl.LIB_PATH)
    f = L.dotmi_plan_tile_deps
    u8, i32, i64 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    f.argtypes = [C.c_int32, u8, u8, i32, C.c_int32, C.c_int32, i64, i64, i64]
    live = np.ascontiguousarray(live, dtype=np.uint8)
    pat = np.ascontiguousarray(pat, dtype=np.uint8)
    c0 = np.ascontiguousarray(c0, dtype=np.int32)
    nd = C.c_int64()
    assert f(nt, live.ctypes.data_as(u8), pat.ctypes.data_as(u8), c0.ctypes.data_as(i32), eager_min, eager_chunk, None, None,
             C.byref(nd)) == 0
    ptr = np.zeros(ntasks + 1, dtype=np.int64)
    idx = np.zeros(max(nd.value, 1), dtype=np.int64)
    assert f(nt, live.ctypes.data_as(u8), pat.ctypes.data_as(u8), c0.ctypes.data_as(i32), eager_min, eager_chunk,
             ptr.ctypes.data_as(i64), idx.ctypes.data_as(i64), C.byref(nd)) == 0
    return ptr, idx[:nd.value]


def run_dataflow(nt, live, pat, c0, eager_min, eager_chunk, seed, orders=3):
    """The tasks executed ONE AT A TIME, IN PLACE, in random orders that respect only the edges of dotmi_plan_tile_deps
    (what the persistent workgroups of tile_flow_kernel wait for): every such order must give the inverse factor -- a
    missing read-after-write, write-after-read or write-after-write edge shows up as a wrong or NaN tile."""
    tasks, prods, nlev, storage, roff, rld = plan(nt, live, pat, c0, eager_min, eager_chunk)
    order0 = np.argsort(tasks[:, 0], kind="stable")          # the order of dotmi_plan_tile_deps
    tasks = tasks[order0]
    ptr, idx = plan_deps(nt, live, pat, c0, eager_min, eager_chunk, len(tasks))
    assert ptr[-1] == len(idx)
    for v in range(len(tasks)):
        assert all(u < v for u in idx[ptr[v]:ptr[v + 1]])   # topological: a task only waits for earlier tickets
    rng = np.random.default_rng(seed)
    n = 64 * nt
    H = np.zeros((n, n))
    for i in range(nt):
        for j in range(i, nt):
            if pat[i, j] and live[i] and live[j]:
                H[64 * i:64 * i + 64, 64 * j:64 * j + 64] = rng.standard_normal((64, 64)) * 0.05
    H = np.triu(H) + np.triu(H, 1).T
    H += np.diag(np.abs(H).sum(1) + 1.0)
    for j in range(nt):
        if not live[j]:
            H[64 * j:64 * j + 64, :] = 0
            H[:, 64 * j:64 * j + 64] = 0
            H[64 * j:64 * j + 64, 64 * j:64 * j + 64] = np.eye(64)
    Q = np.linalg.inv(np.linalg.cholesky(H).T)
    succ = [[] for _ in tasks]
    for v in range(len(tasks)):
        for u in idx[ptr[v]:ptr[v + 1]]:
            succ[u].append(v)
    worst = 0.0
    for rep in range(orders):
        M = np.full(2 * storage + 4096, np.nan)

        def tile(off, ld):
            return np.lib.stride_tricks.as_strided(M[off:], shape=(64, 64), strides=(8, 8 * ld))
        for j in range(nt):
            if live[j]:
                for i in range(int(c0[j]), j + 1):
                    if pat[i, j] or i == j:
                        tile(storage + roff[j] + 64 * i - 64 * c0[j], rld[j])[:, :] = H[64 * i:64 * i + 64, 64 * j:64 * j + 64]
        left = np.array([ptr[v + 1] - ptr[v] for v in range(len(tasks))])
        ready = [v for v in range(len(tasks)) if left[v] == 0]
        done = 0
        while ready:
            # rep 0: latest ready task first (the most adversarial simple order), then random picks
            k = len(ready) - 1 if rep == 0 else int(rng.integers(len(ready)))
            t = ready.pop(k)
            _, form, init, post, nprod, first, coff, qoff, ldc, ldq, ooff = tasks[t]
            acc = tile(coff, ldc).copy() if init else np.zeros((64, 64))
            for p in range(first, first + nprod):
                a, b, lda, ldb = prods[p]
                acc = acc - tile(a, lda).T @ tile(b, ldb) if form == TF_FACT else acc + tile(a, lda) @ tile(b, ldb)
            if post == TP_DIAG:
                out = np.linalg.inv(np.linalg.cholesky(acc).T)
            elif post == TP_ROW:
                out = tile(qoff, ldq).T @ acc
            elif post == TP_RMUL:
                out = -acc @ np.triu(tile(qoff, ldq))
            elif post == TP_NEG:
                out = -acc
            else:
                out = acc
            tile(ooff, ldc)[:, :] = out
            done += 1
            for w in succ[t]:
                left[w] -= 1
                if left[w] == 0:
                    ready.append(w)
        assert done == len(tasks)
        for j in range(nt):
            if not live[j]:
                continue
            for i in range(int(c0[j]), j + 1):
                got = tile(roff[j] + 64 * i - 64 * c0[j], rld[j])
                ref = Q[64 * i:64 * i + 64, 64 * j:64 * j + 64]
                if i == j:
                    worst = max(worst, np.abs(np.triu(got) - ref).max())
                elif np.abs(ref).max() > 0:
                    assert np.isfinite(got).all()
                    worst = max(worst, np.abs(got - ref).max())
    return worst, len(idx), len(tasks)


@pytest.mark.parametrize("eager", [(1000, 1), (2, 1), (4, 4)])
def test_dataflow_dependencies_of_a_two_level_dissection(eager):
    rng = np.random.default_rng(1)
    nt = 14
    pat, c0 = nd_pattern(nt, [(0, 3), (3, 5), (6, 8), (8, 11)], [(5, 6, 0), (11, 12, 6), (12, 14, 0)], rng)
    live = np.ones(nt, dtype=np.uint8)
    live[3] = 0
    pat[3, :] = 0; pat[:, 3] = 0
    worst, ndeps, ntask = run_dataflow(nt, live, pat, c0, eager[0], eager[1], seed=4)
    assert worst < 1e-10, worst
    assert ndeps < 12 * ntask          # the transitive edges are dropped: a handful of flags per task


def test_dataflow_dependencies_of_dense_and_random_patterns():
    rng = np.random.default_rng(6)
    pat = np.triu(np.ones((6, 6), dtype=np.uint8))
    worst, _, _ = run_dataflow(6, np.ones(6, dtype=np.uint8), pat, np.zeros(6, dtype=np.int32), 2, 1, seed=7)
    assert worst < 1e-10
    for seed in range(4):
        nt = int(rng.integers(4, 10))
        pat = np.triu((rng.random((nt, nt)) < 0.35).astype(np.uint8))
        np.fill_diagonal(pat, 1)
        worst, _, _ = run_dataflow(nt, np.ones(nt, dtype=np.uint8), pat, np.zeros(nt, dtype=np.int32), 3, 2, seed=20 + seed)
        assert worst < 1e-9, (seed, worst)
