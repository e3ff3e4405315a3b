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


# ---- round 6: the schedule in the two-level form (leaves against the separator complement; tile_factor.hpp twoLevel) -------------
def plan_two_level(nt, live, pat, c0, leaf, c0m, ntm, eager_min, eager_chunk):
    L = C.CDLL(dl.LIB_PATH)
    f = L.dotmi_plan_tile_schedule_two_level
    u8, i32, i64 = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    f.argtypes = [C.c_int32, u8, u8, i32, u8, i32, i32, C.c_int32, C.c_int32, i64, i64, i64, i64, i64, i64, i64, i32, i64, i32]
    arrs = [np.ascontiguousarray(a, dtype=t) for a, t in ((live, np.uint8), (pat, np.uint8), (c0, np.int32), (leaf, np.uint8),
                                                          (c0m, np.int32), (ntm, np.int32))]
    n = [C.c_int64() for _ in range(4)]
    roff, rld, roffm, rldm = np.zeros(nt, np.int64), np.zeros(nt, np.int32), np.zeros(nt, np.int64), np.zeros(nt, np.int32)
    p = lambda a, t: a.ctypes.data_as(t)
    args = lambda t, pr: (nt, p(arrs[0], u8), p(arrs[1], u8), p(arrs[2], i32), p(arrs[3], u8), p(arrs[4], i32), p(arrs[5], i32),
                          eager_min, eager_chunk, t, pr, C.byref(n[0]), C.byref(n[1]), C.byref(n[2]), C.byref(n[3]),
                          p(roff, i64), p(rld, i32), p(roffm, i64), p(rldm, i32))
    assert f(*args(None, None)) == 0
    tasks = np.zeros((n[0].value, 11), dtype=np.int64)
    prods = np.zeros((max(n[1].value, 1), 4), dtype=np.int64)
    assert f(*args(p(tasks, i64), p(prods, i64))) == 0
    return tasks, prods, n[2].value, n[3].value, roff, rld, roffm, rldm


@pytest.mark.parametrize("eager", [(1000, 1), (2, 1), (4, 4)])
def test_schedule_in_the_two_level_form_gives_the_leaves_and_the_separator_complements_inverse_factors_and_the_panels(eager):
    """A leaves-first block: four leaves (tile rows 0-9), then the separators in post-order (10: of leaves 1, 2; 11: of leaves 3, 4;
    12-13: the root).  The tasks of dotmi_plan_tile_schedule_two_level run level after level in numpy (every task of a level on the
    state the levels before it left, no tile written twice in a level, none read that its level writes) must leave, with
    H = R^T R and Q = R^-1:  Q_ij for i, j in one leaf and for i, j both separators (the inverse of the separator complement's own
    factor = that block of Q), and in the tiles (leaf i, separator j)  T_ij = sum over m in i's leaf of Q_im R_mj  -- the transposed
    panels L_GD X_DD; nothing else of Q is computed.  The two-level solve assembled from exactly these tiles solves H p = r."""
    rng = np.random.default_rng(3)
    nt = 14
    leaves = [(0, 3), (3, 5), (5, 7), (7, 10)]
    leaf_of = np.full(nt, -1)
    for k, (a, b) in enumerate(leaves):
        leaf_of[a:b] = k
    leaf = (leaf_of >= 0).astype(np.uint8)
    sub = {10: (0, 5), 11: (5, 10), 12: (0, 10), 13: (0, 10)}          # leaf tile range of the separator's sub-tree
    c0 = np.arange(nt, dtype=np.int32)
    c0m, ntm = np.zeros(nt, np.int32), np.zeros(nt, np.int32)
    for (a, b) in leaves:
        c0[a:b] = a
    c0[10], c0[11], c0[12], c0[13] = 10, 11, 10, 10                    # first separator column of the sub-tree
    for j, (a, b) in sub.items():
        c0m[j], ntm[j] = a, b - a
    pat = np.zeros((nt, nt), dtype=np.uint8)
    for (a, b) in leaves:
        for i in range(a, b):
            for j in range(i, b):
                pat[i, j] = 1 if j - i <= 1 else 0
    for j, (a, b) in sub.items():
        for i in range(a, b):
            pat[i, j] = rng.random() < 0.6                             # the separator touches some tiles of its leaves
        pat[a, j] = 1
    pat[10, 12] = pat[11, 12] = pat[10, 13] = pat[12, 13] = 1          # separators below the root couple to it
    np.fill_diagonal(pat, 1)
    live = np.ones(nt, dtype=np.uint8)
    tasks, prods, nlev, storage, roff, rld, roffm, rldm = plan_two_level(nt, live, pat, c0, leaf, c0m, ntm, eager[0], eager[1])
    n = 64 * nt
    H = np.zeros((n, n))
    for i in range(nt):
        for j in range(i, nt):
            if pat[i, j]:
                H[64 * i:64 * i + 64, 64 * j:64 * j + 64] = rng.standard_normal((64, 64)) * 0.05
    H = np.triu(H) + np.triu(H, 1).T
    H += np.diag(np.abs(H).sum(1) + 1.0)
    M = np.full(2 * storage + 4096, np.nan)

    def addr(i, j):     # offset and leading dimension of tile (i, j) of the column-major matrix in a buffer
        if leaf[i] and not leaf[j]:
            assert c0m[j] <= i < c0m[j] + ntm[j]
            return roffm[j] + 64 * (i - c0m[j]), rldm[j]
        assert i >= c0[j]
        return roff[j] + 64 * (i - c0[j]), rld[j]

    def tile(off, ld, buf=None):
        return np.lib.stride_tricks.as_strided((M if buf is None else buf)[off:], shape=(64, 64), strides=(8, 8 * ld))

    for i in range(nt):
        for j in range(i, nt):
            if pat[i, j]:
                o, ld = addr(i, j)
                tile(storage + o, ld)[:, :] = H[64 * i:64 * i + 64, 64 * j:64 * j + 64]
    order = np.argsort(tasks[:, 0], kind="stable")
    k = 0
    while k < len(order):
        lv = tasks[order[k], 0]
        group = []
        while k < len(order) and tasks[order[k], 0] == lv:
            group.append(order[k]); k += 1
        snap = M.copy()
        written = set()
        for t in group:
            _, form, init, post, nprod, first, coff, qoff, ldc, ldq, ooff = tasks[t]
            st = lambda off, ld: tile(off, ld, snap).copy()
            acc = st(coff, ldc) if init else np.zeros((64, 64))
            ops = {coff} if init else set()
            for p_ in range(first, first + nprod):
                a, b, lda, ldb = prods[p_]
                acc = acc - st(a, lda).T @ st(b, ldb) if form == TF_FACT else acc + st(a, lda) @ st(b, ldb)
                ops.update((a, b))
            if post == TP_DIAG:
                out = np.linalg.inv(np.linalg.cholesky(acc).T)
            elif post == TP_ROW:
                out = st(qoff, ldq).T @ acc
                ops.add(qoff)
            elif post == TP_RMUL:
                out = -acc @ np.triu(st(qoff, ldq))
                ops.add(qoff)
            else:
                assert post == TP_STORE
                out = acc
            assert np.isfinite(out).all(), "a task read a tile nobody had written"
            assert ooff not in written
            written.add(ooff)
            tile(ooff, ldc)[:, :] = out
        for t in group:     # a tile written in this level is read only by its own writer
            _, form, init, post, nprod, first, coff, qoff, ldc, ldq, ooff = tasks[t]
            ops = {prods[p_][0] for p_ in range(first, first + nprod)} | {prods[p_][1] for p_ in range(first, first + nprod)}
            if post in (TP_ROW, TP_RMUL):
                ops.add(qoff)
            assert not (ops & (written - {ooff}))
    R = np.linalg.cholesky(H).T
    Q = np.linalg.inv(R)
    blk = lambda A, i, j: A[64 * i:64 * i + 64, 64 * j:64 * j + 64]
    nD = 64 * 10
    QD, QG, T = np.zeros((nD, nD)), np.zeros((n - nD, n - nD)), np.zeros((nD, n - nD))
    for j in range(nt):
        for i in range(j + 1):
            same_leaf = leaf[i] and leaf[j] and leaf_of[i] == leaf_of[j]
            both_sep = not leaf[i] and not leaf[j]
            if same_leaf or (both_sep and i >= c0[j]):
                o, ld = addr(i, j)
                got = tile(o, ld)
                got = np.triu(got) if i == j else got
                if np.abs(blk(Q, i, j)).max() > 0:
                    assert np.abs(got - blk(Q, i, j)).max() < 1e-10
                    (QD if same_leaf else QG)[64 * (i - (0 if same_leaf else 10)):64 * (i - (0 if same_leaf else 10)) + 64,
                                              64 * (j - (0 if same_leaf else 10)):64 * (j - (0 if same_leaf else 10)) + 64] = got
            elif leaf[i] and not leaf[j] and c0m[j] <= i < c0m[j] + ntm[j]:
                a, b = leaves[leaf_of[i]]
                ref = sum(blk(Q, i, m) @ blk(R, m, j) for m in range(i, b))
                o, ld = addr(i, j)
                got = tile(o, ld)
                if np.abs(ref).max() > 0:
                    assert np.abs(got - ref).max() < 1e-10
                    T[64 * i:64 * i + 64, 64 * (j - 10):64 * (j - 10) + 64] = got
                else:
                    assert np.all(np.isnan(got)) or np.abs(got).max() < 1e-13
    # the two-level solve from these pieces: t_G = r_G - T^T r_D ; p_G = Q_GG Q_GG^T t_G ; p_D = Q_DD Q_DD^T r_D - T p_G
    r = rng.standard_normal(n)
    tG = r[nD:] - T.T @ r[:nD]
    pG = QG @ (QG.T @ tG)
    pD = QD @ (QD.T @ r[:nD]) - T @ pG
    p = np.concatenate([pD, pG])
    assert np.abs(H @ p - r).max() < 1e-9 * np.abs(r).max()
    # and it computes less than the explicit inverse of the same block
    tasks1, prods1, *_ = plan(nt, live, pat, np.where(leaf, c0, 0).astype(np.int32), eager[0], eager[1])
    assert len(prods) < len(prods1)
