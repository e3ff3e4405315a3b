// nd_layout.hpp -- nested-dissection layout of the subdomain graphs (host only, plain C++).
// Used by build_device_mesh() and by the host-only ABI entry dotmi_plan_layout().
#pragma once
#include <algorithm>
#include <vector>

namespace dotmi {

// Nested-dissection layout shared by all owned subdomains.  The local vertices of every subdomain are
// ordered [A | C | S] recursively, S being a vertex separator of the subdomain's own graph, so that the
// sub-matrix, its Cholesky factor AND the inverse factor X all have a zero (C,A) block.  A leaf is a
// dense diagonal block; region sizes are padded to the maximum over the owned subdomains (identity
// padding) so that one strided-batched GEMM serves every subdomain.
struct NdNode {
    int off = 0, size = 0;   // padded scalar range [off, off+size) of this node
    int a = -1, c = -1;      // children (both or none)
    int offS = 0, sizeS = 0; // separator block (internal nodes)
};

// vertex adjacency incl. self, ascending (the block pattern of the global Hessian)
inline void build_adjacency(int nV, int nT, const int *T, std::vector<int> &adj_ptr, std::vector<int> &adj_idx)
{
    std::vector<std::vector<int>> nb(nV);
    for (int e = 0; e < nT; ++e)
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) nb[T[4 * e + a]].push_back(T[4 * e + b]);
    adj_ptr.assign(nV + 1, 0);
    for (int v = 0; v < nV; ++v) {
        auto &l = nb[v];
        l.push_back(v);
        std::sort(l.begin(), l.end());
        l.erase(std::unique(l.begin(), l.end()), l.end());
        adj_ptr[v + 1] = adj_ptr[v] + (int)l.size();
    }
    adj_idx.resize(adj_ptr[nV]);
    for (int v = 0; v < nV; ++v) std::copy(nb[v].begin(), nb[v].end(), adj_idx.begin() + adj_ptr[v]);
}

// ---- nested dissection of the subdomain graphs (host, once per mesh) -------------------------------
// The reference factors each subdomain with CHOLMOD's own fill-reducing ordering (CHOLMODSolver.cpp:
// 103-141, analyze_pattern); the ordering is free as far as the result goes.  Here it is chosen so that
// the explicit inverse factor this path streams stays block-sparse: a geometric vertex separator per
// level, the same tree for every owned subdomain.
struct NdBuilder {
    std::vector<NdNode> &tree;
    std::vector<std::vector<std::vector<int>>> &region;
    const std::vector<int> &adj_ptr, &adj_idx;
    const double *X;
    std::vector<int> mark;  // nV, -1
    int maxDepth, minSplit;

    // flop model of the node's factorisation; every part is padded to the largest A, C of the batch, so an
    // unbalanced split costs as much as its bigger half twice
    static long long cost(long long a, long long c, long long s)
    {
        const long long m = std::max(a, c);
        return 2 * m * m * m + 8 * s * m * m + 8 * s * s * m + s * s * s;
    }

    // split `vs` into A | C | S with no edge between A and C
    void split(const std::vector<int> &vs, std::vector<int> &A, std::vector<int> &C, std::vector<int> &S)
    {
        const int n = (int)vs.size();
        A.clear(); C.clear(); S.clear();
        if (n < 24) {
            A = vs;
            return;
        }
        long long best = -1;
        int bestAxis = 0, bestT = n / 2, bestSide = 0;
        std::vector<int> ord(vs);
        for (int axis = 0; axis < 3; ++axis) {
            std::sort(ord.begin(), ord.end(), [&](int u, int v) {
                const double xu = X[3 * u + axis], xv = X[3 * v + axis];
                return xu < xv || (xu == xv && u < v);
            });
            for (int k = -8; k <= 8; ++k) {
                const int t = std::min(n - 1, std::max(1, n / 2 + k * n / 48));
                for (int i = 0; i < n; ++i) mark[ord[i]] = i < t ? 0 : 1;
                int sL = 0, sR = 0;
                for (int i = 0; i < n; ++i) {
                    const int v = ord[i], side = mark[v];
                    bool cut = false;
                    for (int e = adj_ptr[v]; e < adj_ptr[v + 1] && !cut; ++e) {
                        const int mu = mark[adj_idx[e]];
                        cut = mu >= 0 && mu != side;
                    }
                    if (cut) (side == 0 ? sL : sR)++;
                }
                const long long c0 = cost(t - sL, n - t, sL), c1 = cost(t, n - t - sR, sR);
                if (best < 0 || c0 < best) { best = c0; bestAxis = axis; bestT = t; bestSide = 0; }
                if (c1 < best) { best = c1; bestAxis = axis; bestT = t; bestSide = 1; }
            }
            for (int v : vs) mark[v] = -1;
        }
        std::sort(ord.begin(), ord.end(), [&](int u, int v) {
            const double xu = X[3 * u + bestAxis], xv = X[3 * v + bestAxis];
            return xu < xv || (xu == xv && u < v);
        });
        for (int i = 0; i < n; ++i) mark[ord[i]] = i < bestT ? 0 : 1;
        for (int i = 0; i < n; ++i) {
            const int v = ord[i], side = mark[v];
            bool cut = false;
            if (side == bestSide)
                for (int e = adj_ptr[v]; e < adj_ptr[v + 1] && !cut; ++e) {
                    const int mu = mark[adj_idx[e]];
                    cut = mu >= 0 && mu != side;
                }
            (cut ? S : side == 0 ? A : C).push_back(v);
        }
        for (int v : vs) mark[v] = -1;
        std::sort(A.begin(), A.end());
        std::sort(C.begin(), C.end());
        std::sort(S.begin(), S.end());
    }

    int build(std::vector<std::vector<int>> &sets, int depth)
    {
        const int id = (int)tree.size();
        tree.emplace_back();
        region.emplace_back();
        const int np = (int)sets.size();
        int mx = 0;
        for (auto &v : sets) mx = std::max(mx, 3 * (int)v.size());
        auto make_leaf = [&]() {
            tree[id].size = std::max(64, (mx + 63) / 64 * 64);
            region[id] = sets;
            return id;
        };
        if (depth >= maxDepth || mx < minSplit) return make_leaf();
        std::vector<std::vector<int>> As(np), Cs(np), Ss(np);
        int mc = 0, ms = 0;
        for (int p = 0; p < np; ++p) {
            split(sets[p], As[p], Cs[p], Ss[p]);
            mc = std::max(mc, (int)Cs[p].size());
            ms = std::max(ms, 3 * (int)Ss[p].size());
        }
        if (mc == 0) return make_leaf();
        const int a = build(As, depth + 1);
        const int c = build(Cs, depth + 1);
        tree[id].a = a;
        tree[id].c = c;
        tree[id].sizeS = (ms + 63) / 64 * 64;
        tree[id].size = tree[a].size + tree[c].size + tree[id].sizeS;
        region[id] = Ss;
        return id;
    }

    void layout(int id, int off)
    {
        NdNode &N = tree[id];
        N.off = off;
        if (N.a < 0) return;
        layout(N.a, off);
        layout(N.c, off + tree[N.a].size);
        N.offS = off + tree[N.a].size + tree[N.c].size;
    }
};


// layout of the given vertex sets (one per owned subdomain): tree[0] is the root, region[node][part] the
// vertices of the node's leaf block / separator, ascending; returns the padded size (lda, multiple of 64)
inline int nd_plan(const std::vector<std::vector<int>> &partVerts, int nV, const std::vector<int> &adj_ptr,
                   const std::vector<int> &adj_idx, const double *Xrest, int levels, int minSplit,
                   std::vector<NdNode> &tree, std::vector<std::vector<std::vector<int>>> &region)
{
    tree.clear();
    region.clear();
    NdBuilder nb{tree, region, adj_ptr, adj_idx, Xrest, std::vector<int>(nV, -1), levels, minSplit};
    std::vector<std::vector<int>> sets(partVerts);
    const int root = nb.build(sets, 0);
    if (tree[root].size < 128) tree[root].size = 128;  // only a leaf root can be that small
    nb.layout(root, 0);
    return tree[root].size;
}

}  // namespace dotmi
