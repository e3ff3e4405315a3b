// partition.hpp -- element partitioner for meshes that come without the reference's METIS partition (host only, plain
// C++, seedless and deterministic).  Role in the reference: METIS::partMesh (src/Utils/METIS.hpp:109-140), i.e.
// METIS_PartMeshDual on the tet list; METIS itself is third-party and stays an input (dotmi_mesh::epart).  This is
// the fallback the ABI applies when epart == NULL, and what `DOT -1 <nodes/block>` runs use for fixture-less meshes.
//
// Method: recursive bisection of the DUAL graph (tets adjacent through a shared face).  Every bisection starts from a
// coordinate split along the longest extent of the element centroids and is then refined by Fiduccia-Mattheyses style
// passes: boundary elements move to the other side in order of decreasing gain (cut faces removed - cut faces added)
// while the sizes stay within a tolerance of the target; a pass may take zero- and negative-gain moves and is rolled
// back to its best prefix, so it climbs out of the jagged local minima a plain coordinate cut leaves.  The quantity
// that matters downstream is the number of interface vertices (vertices shared by several subdomains: they enlarge
// every subdomain matrix and the averaging error of the preconditioner); the face cut is its proxy, as in METIS.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <numeric>
#include <vector>

namespace dotmi {

struct DualGraph {
    std::vector<std::array<int, 4>> nb;  // per tet: the tet across each face, -1 = boundary
};

inline DualGraph build_dual_graph(int nT, const int32_t *T)
{
    static const int FACE[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
    struct Key {
        int a, b, c, e, f;
    };
    std::vector<Key> keys((size_t)4 * nT);
    for (int e = 0; e < nT; ++e)
        for (int f = 0; f < 4; ++f) {
            int v[3] = {T[4 * e + FACE[f][0]], T[4 * e + FACE[f][1]], T[4 * e + FACE[f][2]]};
            std::sort(v, v + 3);
            keys[(size_t)4 * e + f] = {v[0], v[1], v[2], e, f};
        }
    std::sort(keys.begin(), keys.end(), [](const Key &x, const Key &y) {
        if (x.a != y.a) return x.a < y.a;
        if (x.b != y.b) return x.b < y.b;
        if (x.c != y.c) return x.c < y.c;
        return x.e < y.e;
    });
    DualGraph G;
    G.nb.assign(nT, {-1, -1, -1, -1});
    for (size_t i = 0; i + 1 < keys.size(); ++i) {
        const Key &x = keys[i], &y = keys[i + 1];
        if (x.a == y.a && x.b == y.b && x.c == y.c) {
            G.nb[x.e][x.f] = y.e;
            G.nb[y.e][y.f] = x.e;
            ++i;
        }
    }
    return G;
}

struct Bisector {
    const DualGraph &G;
    const std::vector<std::array<double, 3>> &cent;
    std::vector<int> side;   // per tet: -1 = not in the current subset, 0 / 1 = side of the running bisection
    std::vector<int32_t> &epart;
    std::vector<char> locked = {};   // per tet: moved in the running refinement pass (all zero between passes)

    // gain of moving e to the other side: cut faces removed - cut faces added (neighbours inside the subset only)
    int gain(int e) const
    {
        int g = 0;
        for (int k = 0; k < 4; ++k) {
            const int u = G.nb[e][k];
            if (u < 0 || side[u] < 0) continue;
            g += (side[u] != side[e]) ? 1 : -1;
        }
        return g;
    }
    bool on_boundary(int e) const
    {
        for (int k = 0; k < 4; ++k) {
            const int u = G.nb[e][k];
            if (u >= 0 && side[u] >= 0 && side[u] != side[e]) return true;
        }
        return false;
    }

    // minLeft / minRight: every side keeps at least as many elements as it will be split into parts
    void refine(const std::vector<int> &ids, int targetLeft, int minLeft, int minRight)
    {
        const int n = (int)ids.size();
        const int tol = std::max(2, n / 64);   // sizes within ~1.5 % of the target
        int left = 0;
        for (int e : ids) left += side[e] == 0;
        if (locked.size() != G.nb.size()) locked.assign(G.nb.size(), 0);   // once per Bisector; every pass leaves it all zero
        for (int pass = 0; pass < 12; ++pass) {
            // buckets of boundary elements by gain (-4..4), FIFO inside a bucket, lazily validated
            std::vector<std::vector<int>> bucket(9);
            for (int e : ids)
                if (on_boundary(e)) bucket[gain(e) + 4].push_back(e);
            std::vector<size_t> head(9, 0);
            std::vector<int> moved;
            int cur = 0, best = 0, bestAt = 0;
            const int maxMoves = std::max(64, n / 8);
            while ((int)moved.size() < maxMoves) {
                int pick = -1;
                for (int b = 8; b >= 0 && pick < 0; --b)
                    while (head[b] < bucket[b].size()) {
                        const int e = bucket[b][head[b]++];
                        if (locked[e] || gain(e) + 4 != b || !on_boundary(e)) continue;   // stale entry
                        const int nl = left + (side[e] == 0 ? -1 : 1);
                        if (nl < minLeft || n - nl < minRight) continue;
                        if (std::abs(nl - targetLeft) > tol && std::abs(nl - targetLeft) >= std::abs(left - targetLeft)) continue;
                        pick = e;
                        break;
                    }
                if (pick < 0) break;
                cur += gain(pick);
                left += side[pick] == 0 ? -1 : 1;
                side[pick] ^= 1;
                locked[pick] = 1;
                moved.push_back(pick);
                for (int k = 0; k < 4; ++k) {
                    const int u = G.nb[pick][k];
                    if (u >= 0 && side[u] >= 0 && !locked[u] && on_boundary(u)) bucket[gain(u) + 4].push_back(u);
                }
                if (cur > best) {
                    best = cur;
                    bestAt = (int)moved.size();
                }
                if ((int)moved.size() - bestAt > 200) break;   // long way below the best prefix: stop climbing
            }
            // roll back to the best prefix
            for (int i = (int)moved.size() - 1; i >= bestAt; --i) {
                const int e = moved[i];
                left += side[e] == 0 ? -1 : 1;
                side[e] ^= 1;
            }
            for (int e : moved) locked[e] = 0;
            if (best <= 0) break;
        }
    }

    void run(std::vector<int> &ids, int lo, int nparts)
    {
        if (nparts == 1) {
            for (int e : ids) epart[e] = lo;
            return;
        }
        // candidates: a coordinate split along each axis and a graph-growing split (breadth-first from the extreme
        // element of the longest axis: follows the shape where a plane would cut through several limbs), each
        // refined; the one with the fewest cut faces wins (ties: first)
        const int nl = nparts / 2;
        const int cut = (int)((long long)ids.size() * nl / nparts);
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
        for (int e : ids)
            for (int d = 0; d < 3; ++d) {
                mn[d] = std::min(mn[d], cent[e][d]);
                mx[d] = std::max(mx[d], cent[e][d]);
            }
        int longest = 0;
        for (int d = 1; d < 3; ++d)
            if (mx[d] - mn[d] > mx[longest] - mn[longest]) longest = d;
        auto cut_faces = [&]() {
            long c = 0;
            for (int e : ids)
                if (side[e] == 0)
                    for (int k = 0; k < 4; ++k) {
                        const int u = G.nb[e][k];
                        c += (u >= 0 && side[u] == 1);
                    }
            return c;
        };
        std::vector<int> bestSide;
        long bestCut = -1;
        std::vector<int> ord(ids);
        for (int cand = 0; cand < 4; ++cand) {
            if (cand < 3) {
                std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return cent[a][cand] < cent[b][cand]; });
            } else {
                // breadth-first order of the subset's dual graph from the lowest element along the longest axis
                // (disconnected remainders are appended in that axis' order)
                std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return cent[a][longest] < cent[b][longest]; });
                std::vector<int> bfs;
                bfs.reserve(ord.size());
                for (int e : ids) side[e] = 2;   // 2 = not visited yet
                for (int seed : ord) {
                    if (side[seed] != 2) continue;
                    side[seed] = 3;
                    bfs.push_back(seed);
                    for (size_t h = bfs.size() - 1; h < bfs.size(); ++h)
                        for (int k = 0; k < 4; ++k) {
                            const int u = G.nb[bfs[h]][k];
                            if (u >= 0 && side[u] == 2) {
                                side[u] = 3;
                                bfs.push_back(u);
                            }
                        }
                }
                ord.swap(bfs);
            }
            for (size_t i = 0; i < ord.size(); ++i) side[ord[i]] = (int)i < cut ? 0 : 1;
            refine(ids, cut, nl, nparts - nl);
            const long c = cut_faces();
            if (bestCut < 0 || c < bestCut) {
                bestCut = c;
                bestSide.resize(ids.size());
                for (size_t i = 0; i < ids.size(); ++i) bestSide[i] = side[ids[i]];
            }
        }
        for (size_t i = 0; i < ids.size(); ++i) side[ids[i]] = bestSide[i];
        std::vector<int> L, R;
        for (int e : ids) (side[e] == 0 ? L : R).push_back(e);
        for (int e : ids) side[e] = -1;
        std::sort(L.begin(), L.end());
        std::sort(R.begin(), R.end());
        run(L, lo, nl);
        run(R, lo + nl, nparts - nl);
    }
};

// epart[e] in [0, nParts): every part non-empty when nT >= nParts
inline void partition_elements(int nV, int nT, const int32_t *T, const double *X, int nParts, int32_t *epart_out)
{
    (void)nV;
    std::vector<int32_t> epart(nT, 0);
    if (nParts > 1 && nT > 0) {
        const DualGraph G = build_dual_graph(nT, T);
        std::vector<std::array<double, 3>> cent(nT);
        for (int e = 0; e < nT; ++e)
            for (int d = 0; d < 3; ++d)
                cent[e][d] = 0.25 * (X[3 * T[4 * e] + d] + X[3 * T[4 * e + 1] + d] + X[3 * T[4 * e + 2] + d] + X[3 * T[4 * e + 3] + d]);
        Bisector B{G, cent, std::vector<int>(nT, -1), epart};
        std::vector<int> ids(nT);
        std::iota(ids.begin(), ids.end(), 0);
        B.run(ids, 0, std::min(nParts, nT));
    }
    std::copy(epart.begin(), epart.end(), epart_out);
}

}  // namespace dotmi
