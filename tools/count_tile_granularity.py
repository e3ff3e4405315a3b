"""Tile factorisation at 64 vs 128 granularity, counted on the real subdomain patterns (host only): tasks, products, levels,
flop and tile traffic of the level schedule (dotmi_plan_tile_schedule on each subdomain's upper tile pattern in the library's
dissection layout; the 128-pattern is the 64-pattern coarsened 2 x 2).   python tools/count_tile_granularity.py bar17K_twist"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from dot_amd import lib as dl
from dot_amd.sharding import plan_layout
from dot_amd.workloads import load_workload


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
    return tasks, n[1].value, n[2].value


name = sys.argv[1] if len(sys.argv) > 1 else "bar17K_twist"
eager = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc, ep, nparts = load_workload(name)
nodes, nmax, pos, verts = plan_layout(sc.V_rest, sc.T, ep, nparts)
nV = sc.V_rest.shape[0]
T = sc.T
import scipy.sparse as sp
i = T[:, [0, 0, 0, 1, 1, 2]].ravel(); j = T[:, [1, 2, 3, 2, 3, 3]].ravel()
A = sp.coo_matrix((np.ones(i.size, dtype=np.int8), (i, j)), shape=(nV, nV)).tocsr()
A = ((A + A.T + sp.identity(nV, dtype=np.int8)) > 0).tocsr()
nodeC0 = np.zeros(nmax, dtype=np.int32)
for (off, size, a, c, offS, sizeS) in nodes:
    if a < 0:
        nodeC0[off:off + size] = off
    else:
        nodeC0[offS:offS + sizeS] = off
tot = {64: np.zeros(5), 128: np.zeros(5)}
for p in range(nparts):
    v, ps = verts[p], pos[p]
    g2p = -np.ones(nV, dtype=np.int64)
    g2p[v] = ps
    nt = nmax // 64
    pat = np.zeros((nt, nt), dtype=np.uint8)
    live = np.zeros(nt, dtype=np.uint8)
    for q in ps:
        live[q // 64] = 1
        live[(q + 2) // 64] = 1
    sub = A[v][:, v].tocoo()
    r0 = ps[sub.row]; c0_ = ps[sub.col]
    for da in (0, 2):
        for db in (0, 2):
            I = (c0_ + db) // 64; J = (r0 + da) // 64
            m = I <= J
            pat[I[m], J[m]] = 1
    for G in (64, 128):
        if G == 64:
            P, Lv, c0t, n_t = pat, live, nodeC0[::64] // 64, nt
        else:
            n_t = (nt + 1) // 2
            P = np.zeros((n_t, n_t), dtype=np.uint8)
            Lv = np.zeros(n_t, dtype=np.uint8)
            for I in range(nt):
                Lv[I // 2] |= live[I]
                for J in range(I, nt):
                    if pat[I, J]:
                        P[I // 2, J // 2] = 1
            c0t = np.array([min(nodeC0[min(128 * J, nmax - 1)], nodeC0[min(128 * J + 64, nmax - 1)]) // 128 for J in range(n_t)], dtype=np.int32)
        tasks, nprods, nlev = plan(n_t, Lv, P, c0t, eager, eager)
        ndiag = int((tasks[:, 3] == 1).sum()); nrow = int((tasks[:, 3] == 2).sum())
        tot[G] += np.array([len(tasks), nprods, nlev, ndiag, nrow])
for G in (64, 128):
    t = tot[G]
    ntask, nprod, nlev, ndiag, nrow = t
    tile_b = G * G * 8
    flop = 2.0 * G ** 3 * (nprod + nrow) + ndiag * (2.0 / 3.0) * G ** 3
    traffic = (2 * ntask + 2 * nprod + nrow) * tile_b      # C read + write per task, two operands per product, Q_kk per row task
    print(f"{name} tiles of {G}: tasks {int(ntask)}, products {int(nprod)}, levels per subdomain {nlev / nparts:.1f}, diag {int(ndiag)}, row {int(nrow)}: "
          f"{flop / 1e9:.1f} GF, tile traffic {traffic / 1e9:.2f} GB, {flop / traffic:.1f} flop/B")

# ---- where the products are (64 granularity, last subdomain planned above is representative; summed over all) -------------
print("\nproducts by task kind at 64 granularity, one subdomain (the last):")
tasks, nprods, nlev = plan(nt, live, pat, nodeC0[::64] // 64, eager, eager)
kinds = {(0, 0): "FACT eager (TP_STORE)", (0, 1): "FACT DIAG", (0, 2): "FACT ROW", (1, 0): "INV eager (TP_STORE)", (1, 3): "INV QFIN (TP_NEG)", (1, 4): "INV final (TP_RMUL)"}
for key, nm in kinds.items():
    m = (tasks[:, 1] == key[0]) & (tasks[:, 3] == key[1])
    print(f"   {nm:24s} tasks {int(m.sum()):5d} products {int(tasks[m, 4].sum()):6d}")
# 2 x 2 population: for INV tasks, pair rows (i, i+1) and columns (j, j+1): how many of the 4 targets exist
