"""what would the back-solve READ per application in three forms, counted from the symbolic pattern of the
subdomain matrices (host only, no GPU):

  (a) today's explicit inverse X_s = chol(H_s)^-1 in the library's nested-dissection layout, one streaming pass
      (p = X^T (X r)): 8 x the structural non-zeros of X (rows of a tree region run from the region's first column);
  (b) a PARTITIONED inverse on a dissection tree of depth d: the Cholesky factor L kept as it is in the off-diagonal blocks
      between tree nodes (L_{S,A} ...), only the DIAGONAL blocks inverted (dense triangles), applied level by level up the
      tree (forward) and down again (backward) -- two passes over every stored entry:
      16 x (sum over nodes of n_node (n_node + 1) / 2  +  nnz of the off-diagonal blocks of L);
      off-diagonal blocks counted (b1) exactly (sparse storage, the lower bound) and (b2) as dense panels over the rows of
      the ancestor node that are non-zero anywhere in the block (what a tile kernel would stream);
  (c) the factor itself, forward + backward substitution: 16 x nnz(L) -- with the library's ordering, and CHOLMOD's own
      nnz_L (bench.py's cpu_baseline.reference_cholmod) quoted beside it when given;
  (h_k) the HYBRID: only the separators of the top k tree levels kept as L blocks (dense panels as in b2
      + their inverted diagonal triangles), every subtree below them as an explicit inverse in the layout (rows from the
      region's first column).  2 k joins instead of 2 x depth -- but the subtrees' inverses are then read TWICE as well
      (y_R = X_R r_R before the top levels, p_R = X_R^T (y_R - sum_S L_SR^T p_S) after them):
      16 x (nnz(X) of the rows below level k + panels and triangles of the top k levels).

L is the symbolic Cholesky factor of the subdomain's vertex graph (3 x 3 blocks: x 9 scalars per vertex pair, the
diagonal blocks' triangles x 6) in the order the layout gives (leaves, then separators, children before parents).

    python tools/count_backsolve_forms.py bar17K_twist [split=<smallest region split, scalars>] [levels ...]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from dot_amd.sharding import plan_layout
from dot_amd.workloads import load_workload


def vertex_adjacency(T, nV):
    import scipy.sparse as sp
    i = T[:, [0, 0, 0, 1, 1, 2]].ravel()
    j = T[:, [1, 2, 3, 2, 3, 3]].ravel()
    A = sp.coo_matrix((np.ones(i.size, dtype=np.int8), (i, j)), shape=(nV, nV)).tocsr()
    A = ((A + A.T) > 0).astype(np.int8).tocsr()
    return A


def symbolic_cholesky(adj_lists, order):
    """row structures of L (as sets of column positions < row position) for the graph in elimination order `order`
    (quotient-free up-looking version through the elimination tree: struct(L_i*) = reach of adj(i) below i)"""
    n = len(order)
    posof = {v: k for k, v in enumerate(order)}
    parent = [-1] * n
    rows = [None] * n
    for i, v in enumerate(order):
        mark = {i}
        row = set()
        for w in adj_lists[v]:
            k = posof.get(w, -1)
            if k < 0 or k >= i:
                continue
            # walk up the elimination tree from k until a marked node
            while k not in mark and k < i:
                row.add(k)
                mark.add(k)
                if parent[k] < 0:
                    parent[k] = i
                k = parent[k]
        rows[i] = row
    return rows


def count(name, levels_list, min_split=-1):
    sc, ep, nparts = load_workload(name)
    nV = sc.V_rest.shape[0]
    A = vertex_adjacency(sc.T, nV)
    fixed = sc.fixed.astype(bool)
    print(f"{name}: {nV} vertices, {sc.T.shape[0]} tets, {nparts} subdomains")
    print(f"{'levels':>6} {'nmax':>6} | {'(a) X one pass':>15} | {'(b1) part. inv, sparse L blocks':>32} | "
          f"{'(b2) dense panels':>18} | {'(c) 2 x nnz(L)':>15} | (h_k) hybrid, top k = 1, 2, 3 levels as L panels   [MB per back-solve]")
    for levels in levels_list:
        nodes, nmax, pos, verts = plan_layout(sc.V_rest, sc.T, ep, nparts, levels=levels, min_split=min_split)
        # node of every padded row: leaves own [off, off+size), separators [offS, offS+sizeS); depth-first ids, children > parent
        nn = len(nodes)
        first = np.zeros(nmax, dtype=np.int32)    # first column of the row's region (explicit inverse)
        node_of = -np.ones(nmax, dtype=np.int32)
        for idx, (off, size, a, c, offS, sizeS) in enumerate(nodes):
            if a < 0:
                node_of[off:off + size] = idx
                first[off:off + size] = off
            else:
                node_of[offS:offS + sizeS] = idx
                first[offS:offS + sizeS] = off
        a_bytes = b1 = b2 = c_bytes = b3 = b4 = 0
        b5 = [0, 0, 0]
        b6 = 0
        depth = np.zeros(nn, dtype=np.int32)      # children follow their parent in the node list
        for idx, (off, size, a, c, offS, sizeS) in enumerate(nodes):
            if a >= 0:
                depth[a] = depth[c] = depth[idx] + 1
        KS = (1, 2, 3)
        hyb = {k: 0 for k in KS}
        for p in range(nparts):
            v, ps = verts[p], pos[p]
            free = ~fixed[v]
            # (fixed vertices are identity rows / columns: they stay in the layout, couple to nothing)
            order = v[np.argsort(ps)]
            psort = np.sort(ps)
            # (a): rows from the region's first column to the diagonal, live columns only
            live = np.zeros(nmax + 1, dtype=np.int64)
            for q in ps:
                live[q + 1:q + 4] = 1
            cum = np.cumsum(live)
            for q in ps:
                for d in range(3):
                    r = q + d
                    a_bytes += 8 * (cum[r + 1] - cum[first[r]])
                    for k in KS:     # rows of a separator in the top k levels are L panels (below); every other row as in (a), twice
                        nr = node_of[r]
                        if not (nodes[nr][2] >= 0 and depth[nr] < k):
                            hyb[k] += 16 * (cum[r + 1] - cum[first[r]])
            inpart = set(int(x) for x in v)
            adj = {int(x): [int(w) for w in A.indices[A.indptr[x]:A.indptr[x + 1]] if int(w) in inpart and not fixed[w] and not fixed[x]]
                   for x in v}
            rows = symbolic_cholesky(adj, [int(x) for x in order])
            nd_v = node_of[psort]      # tree node of every vertex in elimination order
            nnzL = 0
            blk = {}                   # (row node, col node) -> [nnz (vertex pairs), set of rows]
            nsize = np.bincount(nd_v, minlength=nn)
            tpairs = set()             # (64-row tile, 64-column tile) pairs of the padded layout that hold an off-diagonal-block entry
            for i, row in enumerate(rows):
                nnzL += 9 * len(row) + 6
                for k in row:
                    key = (nd_v[i], nd_v[k])
                    if key[0] != key[1]:
                        e = blk.setdefault(key, [0, set()])
                        e[0] += 1
                        e[1].add(i)
                        for ti in {int(psort[i]) // 64, (int(psort[i]) + 2) // 64}:
                            for tk in {int(psort[k]) // 64, (int(psort[k]) + 2) // 64}:
                                tpairs.add((ti, tk))
            diag = sum(int(3 * s) * (int(3 * s) + 1) // 2 for s in nsize)
            b1 += 16 * (diag + sum(9 * e[0] for e in blk.values()))
            b2 += 16 * (diag + sum(9 * len(e[1]) * int(nsize[key[1]]) for key, e in blk.items()))
            c_bytes += 16 * nnzL
            # (b3) the column form M_SN = L_SN X_NN: panels read twice (up and down the tree), the nodes' triangles ONCE (y_N = X_NN r'_N
            # and X_NN^T y_N in the same pass, as today); (b4) the same with the panels at the 64 x 64 tiles of the padded layout
            pan = sum(9 * len(e[1]) * int(nsize[key[1]]) for key, e in blk.items())
            b3 += 8 * (diag + 2 * pan)
            b4 += 8 * (diag + 2 * 4096 * len(tpairs))
            # (b6) TWO levels only: the leaves against the separator complement G (all separators of the subdomain together).  Leaves'
            # triangles once, X_GG = today's inverse restricted to separator rows x separator columns once, and the leaf panels
            # M_GD = L_GD X_DD (rows: the separator vertices next to leaf D, packed) twice.
            isleaf = np.array([nodes[nidx][2] < 0 for nidx in range(nn)])
            leaf_tri = sum(int(3 * nsize[nidx]) * (int(3 * nsize[nidx]) + 1) // 2 for nidx in range(nn) if isleaf[nidx])
            sepcol = np.zeros(nmax + 1, dtype=np.int64)
            for i, q in enumerate(psort):
                if not isleaf[nd_v[i]]:
                    sepcol[q + 1:q + 4] = 1
            csep = np.cumsum(sepcol)
            xgg = 0
            for i, q in enumerate(psort):
                if not isleaf[nd_v[i]]:
                    for d in range(3):
                        r = q + d
                        xgg += csep[r + 1] - csep[first[r]]
            lpan = {}
            for (rn, cn), e in blk.items():
                if isleaf[cn]:
                    lpan.setdefault(int(cn), set()).update(e[1])
            pan6 = sum(9 * len(rs) * int(nsize[cn]) for cn, rs in lpan.items())
            b6 += 8 * (leaf_tri + xgg + 2 * pan6)
            # (b5) = (b4) after re-ordering the vertices INSIDE every separator by the set of descendant nodes their row is non-zero in
            # (rows with the same set become contiguous; the order inside a node is free).  Panels counted at (TR-row tile of the
            # separator) x (whole descendant node), TR = 64 / 32 / 16: a tile kernel that skips a node's columns when none of its
            # rows holds an entry there.
            sig = {}
            for (rn, cn), e in blk.items():
                for i in e[1]:
                    sig.setdefault(i, set()).add(int(cn))
            newrank = {}
            for nidx in range(nn):
                if nodes[nidx][2] < 0:
                    continue
                mem = [i for i in range(len(rows)) if nd_v[i] == nidx]
                mem.sort(key=lambda i: tuple(sorted(sig.get(i, ()))))
                for r, i in enumerate(mem):
                    newrank[i] = r
            for TR, acc in ((64, 0), (32, 1), (16, 2)):
                tot = 0
                for (rn, cn), e in blk.items():
                    tiles_r = {(3 * newrank[i]) // TR for i in e[1]} | {(3 * newrank[i] + 2) // TR for i in e[1]}
                    tot += len(tiles_r) * TR * 3 * int(nsize[cn])
                b5[acc] += 8 * (diag + 2 * tot)
            for k in KS:
                for nidx in range(nn):
                    if nodes[nidx][2] >= 0 and depth[nidx] < k:
                        hyb[k] += 16 * (int(3 * nsize[nidx]) * (int(3 * nsize[nidx]) + 1) // 2)
                for key, e in blk.items():
                    if nodes[key[0]][2] >= 0 and depth[key[0]] < k:
                        hyb[k] += 16 * 9 * len(e[1]) * int(nsize[key[1]])
        print(f"{levels:>6} {nmax:>6} | {a_bytes / 1e6:>15.1f} | {b1 / 1e6:>32.1f} | {b2 / 1e6:>18.1f} | {c_bytes / 1e6:>15.1f} | "
              + " ".join(f"{hyb[k] / 1e6:>9.1f}" for k in KS)
              + f" | (b3) {b3 / 1e6:.1f}  (b4, 64-tiles) {b4 / 1e6:.1f}  (b5: separators re-ordered, row tiles of 64 / 32 / 16) "
              + " / ".join(f"{x / 1e6:.1f}" for x in b5) + f"  (b6: leaves | separator complement) {b6 / 1e6:.1f}")


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "bar17K_twist"
    ms = -1
    args = sys.argv[2:]
    if args and args[0].startswith("split="):
        ms = int(args[0].split("=")[1])
        args = args[1:]
    lv = [int(x) for x in args] or [-1, 3, 4, 5]
    count(name, lv, ms)
