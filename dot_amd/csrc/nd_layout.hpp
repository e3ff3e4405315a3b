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
    int tail = 0;            // leaf: only its last `tail` rows can couple to the root separator
    int crows = 0;           // rows of this sub-tree that can couple to the root separator (leaf tails + separators)
    int tail1 = 0;           // leaf: only its last `tail1` rows can couple to its PARENT's separator (tail1 >= tail)
    // leaves-first layouts (nd_relayout_leaves_first, the two-level back-solve): the leaf columns [offL, endL) of this sub-tree;
    // an internal node's `off` is then the first SEPARATOR column of its sub-tree (endL == offL: the [A | C | S] layout)
    int offL = 0, endL = 0;
};

// first padded row of a node's own region (leaf block / separator) in a subdomain that has `used` live
// scalars there: leaves are right-aligned (padding in front) so that the rows coupled to the root separator
// end exactly at the end of the leaf, separators are left-aligned
inline int nd_region_first_row(const NdNode &N, int used) { return N.a < 0 ? N.off + N.size - used : N.offS; }

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
    std::vector<std::vector<int>> rootS;  // per part: the root separator (set once the root is split)

    // flop model of the node's factorisation; every part is padded to the largest A, C of the batch, so an
    // unbalanced split costs as much as its bigger half twice
    static long long cost(long long a, long long c, long long s)
    {
        // in padded scalar sizes (blocks of 64), raw sizes as the tie-break
        auto model = [](long long a_, long long c_, long long s_) {
            const long long m = std::max(a_, c_);
            return 2 * m * m * m + 8 * s_ * m * m + 8 * s_ * s_ * m + s_ * s_ * s_;
        };
        auto r64 = [](long long v) { return (3 * v + 63) / 64 * 64; };
        return model(r64(a), r64(c), r64(s)) + model(3 * a, 3 * c, 3 * s) / 64;
    }

    // Smallest vertex separator that an ordered bisection (first t of `ord` | rest) admits: a minimum vertex
    // cover of the cut edges, from a maximum bipartite matching (Koenig's theorem).  mark[] must hold the
    // side (0 / 1) of every vertex of `ord`.  Returns the cover; nl / nr = how many of it lie on each side.
    std::vector<int> lid, bl, br, matchL, matchR, seenR, eptr, eidx;
    bool augment(int l, int stamp)
    {
        for (int e = eptr[l]; e < eptr[l + 1]; ++e) {
            const int r = eidx[e];
            if (seenR[r] == stamp) continue;
            seenR[r] = stamp;
            if (matchR[r] < 0 || augment(matchR[r], stamp)) {
                matchR[r] = l;
                matchL[l] = r;
                return true;
            }
        }
        return false;
    }
    void min_cover(const std::vector<int> &ord, int t, std::vector<int> &cover, int &nl, int &nr)
    {
        if (lid.size() != mark.size()) lid.assign(mark.size(), -1);
        bl.clear(); br.clear(); eptr.assign(1, 0); eidx.clear();
        for (int i = 0; i < t; ++i) {
            const int v = ord[i];
            const size_t e0 = eidx.size();
            for (int e = adj_ptr[v]; e < adj_ptr[v + 1]; ++e) {
                const int u = adj_idx[e];
                if (mark[u] != 1) continue;
                if (lid[u] < 0) {
                    lid[u] = (int)br.size();
                    br.push_back(u);
                }
                eidx.push_back(lid[u]);
            }
            if (eidx.size() > e0) {
                bl.push_back(v);
                eptr.push_back((int)eidx.size());
            }
        }
        for (int u : br) lid[u] = -1;
        const int nL = (int)bl.size(), nR = (int)br.size();
        matchL.assign(nL, -1);
        matchR.assign(nR, -1);
        seenR.assign(nR, -1);
        for (int l = 0; l < nL; ++l) augment(l, l);
        // Z = reachable from the unmatched left vertices by alternating paths
        std::vector<char> zl(nL, 0), zr(nR, 0);
        std::vector<int> stack;
        for (int l = 0; l < nL; ++l)
            if (matchL[l] < 0) {
                zl[l] = 1;
                stack.push_back(l);
            }
        while (!stack.empty()) {
            const int l = stack.back();
            stack.pop_back();
            for (int e = eptr[l]; e < eptr[l + 1]; ++e) {
                const int r = eidx[e];
                if (zr[r]) continue;
                zr[r] = 1;
                const int l2 = matchR[r];
                if (l2 >= 0 && !zl[l2]) {
                    zl[l2] = 1;
                    stack.push_back(l2);
                }
            }
        }
        cover.clear();
        nl = nr = 0;
        for (int l = 0; l < nL; ++l)
            if (!zl[l]) {
                cover.push_back(bl[l]);
                ++nl;
            }
        for (int r = 0; r < nR; ++r)
            if (zr[r]) {
                cover.push_back(br[r]);
                ++nr;
            }
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
        int bestAxis = 0, bestT = n / 2;
        std::vector<int> ord(vs), cov;
        auto by_axis = [&](int axis) {
            std::sort(ord.begin(), ord.end(), [&](int u, int v) {
                const double xu = X[3 * u + axis], xv = X[3 * v + axis];
                return xu < xv || (xu == xv && u < v);
            });
        };
        for (int axis = 0; axis < 3; ++axis) {
            by_axis(axis);
            for (int k = -16; k <= 16; ++k) {
                const int t = std::min(n - 1, std::max(1, n / 2 + k * n / 96));
                for (int i = 0; i < n; ++i) mark[ord[i]] = i < t ? 0 : 1;
                int nl = 0, nr = 0;
                min_cover(ord, t, cov, nl, nr);
                const long long c = cost(t - nl, n - t - nr, nl + nr);
                if (best < 0 || c < best) { best = c; bestAxis = axis; bestT = t; }
            }
        }
        by_axis(bestAxis);
        for (int i = 0; i < n; ++i) mark[ord[i]] = i < bestT ? 0 : 1;
        int nl = 0, nr = 0;
        min_cover(ord, bestT, cov, nl, nr);
        for (int v : cov) mark[v] = 2;
        for (int v : vs) (mark[v] == 2 ? S : mark[v] == 0 ? A : C).push_back(v);
        for (int v : vs) mark[v] = -1;
        std::sort(A.begin(), A.end());
        std::sort(C.begin(), C.end());
        std::sort(S.begin(), S.end());
    }

    int build(std::vector<std::vector<int>> &sets, int depth, const std::vector<std::vector<int>> *parentS = nullptr)
    {
        const int id = (int)tree.size();
        tree.emplace_back();
        region.emplace_back();
        const int np = (int)sets.size();
        int mx = 0;
        for (auto &v : sets) mx = std::max(mx, 3 * (int)v.size());
        auto make_leaf = [&]() {
            tree[id].size = std::max(64, (mx + 63) / 64 * 64);
            tree[id].tail = tree[id].tail1 = tree[id].size;
            region[id] = sets;
            if (!rootS.empty()) {
                // order: interior | next to the parent's separator only | next to the root separator.
                // H(leaf, S_root) is zero above the last class and H(leaf, S_parent) above the last two, which the
                // triangular products of those two nodes exploit (TriMult)
                int mt = 0, mt1 = 0;
                for (int p = 0; p < np; ++p) {
                    if (parentS)
                        for (int v : (*parentS)[p]) mark[v] = 4;
                    for (int v : rootS[p]) mark[v] = 3;
                    std::vector<int> in, bp, br;
                    for (int v : sets[p]) {
                        bool ar = false, ap = false;
                        for (int e = adj_ptr[v]; e < adj_ptr[v + 1]; ++e) {
                            ar |= mark[adj_idx[e]] == 3;
                            ap |= mark[adj_idx[e]] == 4;
                        }
                        (ar ? br : ap ? bp : in).push_back(v);
                    }
                    if (parentS)
                        for (int v : (*parentS)[p]) mark[v] = -1;
                    for (int v : rootS[p]) mark[v] = -1;
                    mt = std::max(mt, 3 * (int)br.size());
                    mt1 = std::max(mt1, 3 * (int)(br.size() + bp.size()));
                    in.insert(in.end(), bp.begin(), bp.end());
                    in.insert(in.end(), br.begin(), br.end());
                    region[id][p] = in;
                }
                tree[id].tail = std::min(tree[id].size, (mt + 63) / 64 * 64);
                tree[id].tail1 = std::min(tree[id].size, (mt1 + 63) / 64 * 64);
            }
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
        if (depth == 0) rootS = Ss;
        const int a = build(As, depth + 1, &Ss);
        const int c = build(Cs, depth + 1, &Ss);
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
        N.crows = N.tail;
        if (N.a < 0) return;
        layout(N.a, off);
        layout(N.c, off + tree[N.a].size);
        N.offS = off + tree[N.a].size + tree[N.c].size;
        N.crows = tree[N.a].crows + tree[N.c].crows + N.sizeS;
    }
};


// Leaves-first layout (round 6, the two-level form of the back-solve): every leaf of the tree in front, in tree order, then every
// separator in post-order (children's separators before the parent's).  The vertices keep their regions and their order inside a
// region; only the regions' places change, and the order stays a valid elimination order (a leaf is decoupled from every other
// leaf; a separator still comes behind everything it separates).  What it buys: the separator complement is ONE contiguous range
// at the end, inside which the separators of a sub-tree are contiguous -- so "the rows of a separator from its sub-tree's first
// SEPARATOR column to the diagonal" is a contiguous row range again, the inverse of the separator complement's own factor.
inline void nd_relayout_leaves_first(std::vector<NdNode> &tree)
{
    if (tree.empty() || tree[0].a < 0) return;
    int at = 0;
    // leaves, depth first
    struct L {
        static void leaves(std::vector<NdNode> &t, int id, int &at)
        {
            NdNode &N = t[id];
            N.offL = at;
            if (N.a < 0) {
                N.off = at;
                at += N.size;
            } else {
                leaves(t, N.a, at);
                leaves(t, N.c, at);
            }
            N.endL = at;
        }
        static int seps(std::vector<NdNode> &t, int id, int &at)   // returns the first separator column of the sub-tree (-1: none)
        {
            NdNode &N = t[id];
            if (N.a < 0) return -1;
            const int fa = seps(t, N.a, at), fc = seps(t, N.c, at);
            N.offS = at;
            at += N.sizeS;
            N.off = fa >= 0 ? fa : fc >= 0 ? fc : N.offS;
            return N.off;
        }
    };
    const int total = tree[0].size;
    L::leaves(tree, 0, at);
    L::seps(tree, 0, at);
    tree[0].size = total;   // (== at: the same regions)
}

// default depth of the dissection: two levels for the ~2000-dof subdomains of the headline configurations; big
// subdomains (`timeStepper DOT 6` on a 17k-vertex mesh: ~9800 dofs) go deeper until the leaves are ~1200 dofs.
// Round 5: a THIRD level already from ~2900 dofs per subdomain when the mesh has at least 16 subdomains (nsmax, nParts over
// ALL subdomains of the mesh, so that every rank of a sharded run builds the tree a single GPU builds).  The explicit inverse
// gets ~16 % smaller (1 M tets / 256 subdomains: 4544 -> 3834 MB per back-solve, the factorisation 14.5 -> 14.0 ms, the step
// 68.5 -> 66 ms; kingkong18K / 18 subdomains: factor 1.67 -> 1.35 ms) while the padded layout stays within the single-pass
// back-solve kernel's BS_LONG columns; with few subdomains the launch is bound by its longest tile and the longer padded
// rows cost more than the bytes save (horse7K / 8 subdomains: loop 2.48 -> 3.38 ms) -- there the two levels stay.
inline int nd_default_levels(int nsmax, int nPartsMesh)
{
    int levels = 2;
    for (int sz = nsmax; sz > 4800 && levels < 6; sz /= 2) ++levels;
    if (levels == 2 && nsmax > 2900 && nPartsMesh >= 16) levels = 3;
    return levels;
}
inline int nd_default_levels(const std::vector<std::vector<int>> &partVerts)
{
    int nsmax = 0;
    for (auto &v : partVerts) nsmax = std::max(nsmax, 3 * (int)v.size());
    return nd_default_levels(nsmax, (int)partVerts.size());
}
// Depth and split threshold for a mesh (round 5).  When the size rule above leaves two levels, a THIRD level with regions split
// down to 384 scalars is tried on the layout of ALL subdomains of the mesh (so that every rank of a sharded run decides alike)
// and kept when its padded size still fits the 256-thread back-solve kernel (narrowLimit = BS_NARROW columns, two workgroups
// per CU): bar17K / 32 subdomains 2368 -> 2944 columns, X 229 -> 197 MB, factorisation 1.08 -> 0.98 ms; bunny5K / 8
// 2368 -> 2688, factorisation 0.35 -> 0.31 ms.  Beyond that limit the longer padded rows cost the back-solve more than the
// bytes save (horse7K / 8: 4096 columns) and the two levels stay.
inline void nd_choose_depth(const std::vector<std::vector<int>> &allParts, int nV, const std::vector<int> &adj_ptr,
                            const std::vector<int> &adj_idx, const double *Xrest, int narrowLimit, int defaultMinSplit,
                            int &levels, int &minSplit);
constexpr int ND_MIN_SPLIT = 512;   // smallest region (scalars) that is still split (round 5: 768 -> 512: the 1200-dof subdomains
                                    // of the stiff monkey get their second level -- X 165 -> 124 MB, factor 0.78 -> 0.55 ms)

// layout of the given vertex sets (one per owned subdomain): tree[0] is the root, region[node][part] the
// vertices of the node's leaf block / separator in layout order; returns the padded size (lda, multiple of 64)
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

inline void nd_choose_depth(const std::vector<std::vector<int>> &allParts, int nV, const std::vector<int> &adj_ptr,
                            const std::vector<int> &adj_idx, const double *Xrest, int narrowLimit, int defaultMinSplit,
                            int &levels, int &minSplit)
{
    levels = nd_default_levels(allParts);
    minSplit = defaultMinSplit;
    if (levels != 2) return;
    std::vector<NdNode> tree2, tree3;
    std::vector<std::vector<std::vector<int>>> region;
    const int n2 = nd_plan(allParts, nV, adj_ptr, adj_idx, Xrest, 2, defaultMinSplit, tree2, region);
    const int n3 = nd_plan(allParts, nV, adj_ptr, adj_idx, Xrest, 3, 384, tree3, region);
    // (a third level that really splits something -- a bigger tree -- and still fits the narrow kernel.  ADVICE r05: the earlier
    // extra condition n3 > n2 rejected a three-level layout whose padded size came out equal or smaller, the better case)
    if (tree3.size() > tree2.size() && n3 <= narrowLimit) {
        levels = 3;
        minSplit = 384;
    }
}

}  // namespace dotmi
