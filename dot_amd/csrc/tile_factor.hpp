// tile_factor.hpp -- static schedule of the tile-level inverse-Cholesky of the subdomain blocks (host only).
//
// Role in the reference: CHOLMODSolver::factorize (CHOLMODSolver.cpp:143, called from DOTTimeStepper.cpp:363-377), with
// the symbolic analysis of CHOLMODSolver::analyze_pattern (:103-141) done here at tile granularity.
//
// Every subdomain block (nmax x nmax, column-major, nested-dissection order) is cut into 64 x 64 tiles.  With H = R^T R
// (R upper) and Q = R^-1:
//   DIAG(j)    G = H_jj - sum_m R_mj^T R_mj ;  Q_jj = chol(G)^-1
//   ROW(k,j)   G = H_kj - sum_m R_mk^T R_mj ;  R_kj = Q_kk^T G                       (k < j)
//   INV(i,j)   Q_ij = -(sum_{i <= m < j} Q_im R_mj) Q_jj                              (i < j)
// Round 5: TWO buffers of the same compact layout.  H is filled into the WORK buffer, whose tiles R overwrites in place; the
// tiles of Q go to the FACTOR buffer the back-solve streams and are never anything else.  Until round 4 Q overwrote R in
// one buffer: the sum T_ij of INV went to a scratch tile and a task of its own (QFIN) multiplied it with -Q_jj once every
// reader of R_ij had finished -- a third of all tasks, a fifth of the tile traffic (scratch written and read back) and two
// launches at the end of every factorisation for one product each.  With R and Q apart nothing waits for a reader: the
// last task of T_ij multiplies with -Q_jj itself (TP_RMUL) and stores Q_ij; its eager parts keep their partial sum in the
// Q tile itself.
// Only tiles that can be non-zero exist: the tile pattern of H (from the fill list) is closed under the symbolic
// factorisation, the pattern of Q under the symbolic inversion; tiles that consist of identity padding only are
// skipped altogether -- so the flop count follows each subdomain's own size, not the padded shared layout.
// A task's LEVEL is one more than the highest level among the tasks it reads from (and, for QFIN, among the tasks that
// still read the R tile it overwrites); all tasks of one level, over all subdomains and all tree nodes, run in ONE
// kernel launch (one workgroup per task), the launches of a factorisation being the levels in order.  The four
// leaves of a subdomain advance in the same launches, the separators follow, the inversion of a column overlaps the
// factorisation of the later ones: ~2 launches per tile column on the critical path instead of the ~15 dependent
// GEMM / diagonal-block launches per 128 columns of the recursive formulation.
// Products whose operands exist early are applied eagerly (see `emit`), so a launch is as long as a few products, not as
// the longest product list.  Every sum has a fixed order, so results are bit-identical run to run.
#pragma once
#include <algorithm>
#include <cstdint>
#include <unordered_map>
#include <vector>

namespace dotmi {

constexpr int TILE = 64;

// a task:  acc = (init ? tile c : 0);  acc (-)+= products;  post
enum TileForm { TF_FACT = 0 /* acc -= A^T B */, TF_INV = 1 /* acc += A B */ };
enum TilePost {
    TP_STORE = 0,   // c = acc                       (eager partial update of an H tile / of a scratch T tile)
    TP_DIAG = 1,    // c = (chol(acc)^-1)^T          (Q_jj; strictly lower part zero)
    TP_ROW = 2,     // c = Q_kk^T acc                (R_kj; q = tile of Q_kk)
    TP_NEG = 3,     // c = -acc                      (unused since round 5: was Q_ij = -T_ij Q_jj as a task of its own)
    TP_RMUL = 4     // o = -acc Q_jj                 (Q_ij; q = tile of Q_jj)
};

struct TileProd {
    double *a, *b;   // origins of the two tiles (column-major)
    int lda, ldb;    // their leading dimensions (a tile's rows live in one row block of the factor storage; scratch: 64)
};
struct TileTask {
    int form, init, post, nprod;   // TileForm, read c first?, TilePost, number of products
    int first, sub;                // products [first, first + nprod) of the level-ordered product array; owned subdomain
    int ldc, ldq;                  // leading dimensions of the c tile and of the q tile
    int pivotBase, pad;            // TP_DIAG: scalar offset of the tile's first row (for the non-SPD report)
    double *c;                     // the tile read (init)
    double *q;                     // TP_ROW: tile of Q_kk;  TP_RMUL: tile of Q_jj
    double *o;                     // the tile written (== c except for TP_DIAG: reads the work buffer's H_jj, writes the factor buffer's Q_jj)
    TileProd p0;                   // copy of the first product: its tiles are requested straight from the descriptor, one
                                   // dependent round trip earlier than through the product array
};

struct TileSchedule {
    std::vector<TileTask> tasks;      // level after level
    std::vector<TileProd> prods;      // in task order
    std::vector<int> levelStart;      // tasks of level l are [levelStart[l], levelStart[l+1])
    std::vector<int> levelDiag;       // the first levelDiag[l] of them are the TP_DIAG tasks (run by the diagonal-block kernel)
    std::vector<double *> clearTiles; // origins of the tiles the fill writes into (cleared before the refill; fill-in tiles are
                                      // written before they are read)
    std::vector<int> clearLd;         // and their leading dimensions
    size_t scratchTiles = 0;          // 64 x 64 scratch tiles needed (none since round 5)
    double flops = 0;                 // FP64 flop of one factorisation as executed
    long long liveTiles = 0, qTiles = 0;
};

// One subdomain: nt tiles per side; live[t] = tile row/column t holds at least one live scalar; hpat = upper tile
// pattern of H (hpat[i * nt + j], i <= j).  Appends the subdomain's tasks (with levels) to the lists.
struct TileTaskL {
    TileTask t;
    int level;
    int row = 0;   // tile row of the target: tasks of one subdomain and row share their A operands
    std::vector<TileProd> prods;
};

// rtOff / rtLd / rtC0: the subdomain's row blocks of the factor storage (dotmi_internal.hpp RowTile), W its base, W2 the base
// of the work buffer of the same layout (H, then R).  Tile (i, j) of the column-major matrix = memory rows of block j, memory
// columns 64 i ...: origin base + off_j + 64 i - c0_j, leading dimension ld_j.
inline void plan_subdomain_tiles(int sub, int nt, double *W, const long long *rtOff, const int *rtLd, const int *rtC0,
                                 const std::vector<uint8_t> &live, std::vector<uint8_t> pat /* by value: gets the fill */,
                                 double *W2, size_t &scratchNext, std::vector<TileTaskL> &out,
                                 std::vector<double *> &clearTiles, std::vector<int> &clearLd, double &flops, long long &qTiles,
                                 int eagerMin = 2, int eagerChunk = 1, int eagerMinDiag = 0, bool balance = true,
                                 int eagerMinRmul = -1, const long long *rtOffM = nullptr, const int *rtLdM = nullptr,
                                 const int *rtC0M = nullptr, const uint8_t *leafTile = nullptr)
{
    // Two-level form (leafTile != nullptr; leaves-first layout, nd_layout.hpp): tile rows of the LEAVES against the separator
    // complement.  A separator's row block j stores its separator columns in the main table (rtC0[j] = the first separator column
    // of its sub-tree) and the leaf columns of its sub-tree in a second one (rtOffM / rtLdM / rtC0M).  The factorisation proper is
    // unchanged (R_ij of a leaf row i and a separator column j lives in that second range); of the inversion only the diagonal
    // blocks remain -- Q_ij for i, j in one leaf or both in the separator complement -- and a tile (leaf i, separator j) receives
    //     T_ij = sum over the tiles m >= i of i's leaf of Q_im R_mj          (= the tile of (L_GD X_DD)^T, no product with Q_jj,
    // no terms through the separators in between): what the two-level back-solve streams instead of the inverse's dense
    // (separator, leaf) blocks (k_backsolve.hip, twolevel_*).
    const bool twoLevel = leafTile != nullptr;
    const size_t o0 = out.size();   // this subdomain's tasks are out[o0 ...)
    auto inM = [&](int i, int j) { return twoLevel && (long long)i * TILE < rtC0[j]; };
    auto toff = [&](int i, int j) {
        return inM(i, j) ? rtOffM[j] + (long long)i * TILE - rtC0M[j] : rtOff[j] + (long long)i * TILE - rtC0[j];
    };
    auto tile = [&](int i, int j) { return W2 + toff(i, j); };    // H, then R (work buffer)
    auto qtile = [&](int i, int j) { return W + toff(i, j); };    // Q (factor buffer)
    (void)scratchNext;
    auto tld = [&](int i, int j) { return inM(i, j) ? rtLdM[j] : rtLd[j]; };
    auto P = [&](int i, int j) -> uint8_t & { return pat[(size_t)i * nt + j]; };
    for (int j = 0; j < nt; ++j)
        if (live[j]) P(j, j) = 1;
    const std::vector<uint8_t> hpat = pat;   // tiles the fill writes into (before the symbolic fill-in)
    auto Hp = [&](int i, int j) { return hpat[(size_t)i * nt + j] != 0; };
    // symbolic factorisation: struct(R)
    for (int k = 0; k < nt; ++k) {
        if (!live[k]) continue;
        std::vector<int> row;
        for (int j = k + 1; j < nt; ++j)
            if (P(k, j)) row.push_back(j);
        for (size_t a = 0; a < row.size(); ++a)
            for (size_t b = a; b < row.size(); ++b) P(row[a], row[b]) = 1;
    }
    std::vector<uint8_t> rpat = pat;   // struct(R), diagonal included
    auto Rp = [&](int i, int j) { return rpat[(size_t)i * nt + j] != 0; };
    // symbolic inversion: struct(Q col j) = {j} + union of struct(Q col m) over m < j with R_mj != 0
    std::vector<std::vector<int>> qcol(nt);
    std::vector<uint8_t> qp((size_t)nt * nt, 0);
    for (int j = 0; j < nt; ++j) {
        if (!live[j]) continue;
        std::vector<uint8_t> in(nt, 0);
        in[j] = 1;
        for (int m = 0; m < j; ++m)
            if (Rp(m, j))
                for (int i : qcol[m]) {
                    // two-level: a separator column takes a leaf's rows from that leaf's own columns only (T_ij), not through
                    // the separators in between
                    if (twoLevel && !leafTile[m] && leafTile[i]) continue;
                    in[i] = 1;
                }
        for (int i = 0; i <= j; ++i)
            if (in[i]) {
                qcol[j].push_back(i);
                qp[(size_t)i * nt + j] = 1;
            }
    }
    auto Qp = [&](int i, int j) { return qp[(size_t)i * nt + j] != 0; };
    // A target tile with its products and the level after which each product's operands exist.  Products that are ready
    // early are applied EAGERLY, in tasks of their own at the first possible level (where the GPU has room anyway), so
    // that the task on the critical path -- the one that must wait for the last operand -- only carries the last few
    // products: the level's duration is that of its longest product loop.  A tile is touched by at most one task per
    // level (its eager tasks have distinct levels, all below the final one), so there is no race and the order of the
    // additions is fixed.
    struct PA {
        TileProd p;
        int avail;
    };
    const int EAGER_MIN = eagerMin;   // early products a task on the critical path may keep (besides the last level's)
    const size_t CHUNK = (size_t)std::max(1, eagerChunk);   // early products per eager task (ties in availability stay together)
    auto emit = [&](int row, int form, int post, double *c, int ldc, double *q, int ldq, int pivotBase, std::vector<PA> &pa,
                    int minFinal, bool initFromC, double *dst = nullptr) -> int {
        if (!dst) dst = c;
        std::stable_sort(pa.begin(), pa.end(), [](const PA &x, const PA &y) { return x.avail < y.avail; });
        int amax = 0;
        for (auto &x : pa) amax = std::max(amax, x.avail);
        const int lf = std::max(amax, minFinal) + 1;
        // eager part: everything available before level lf - 1, grouped by availability, as long as more than EAGER_MIN
        // products would otherwise wait for the final task
        size_t nEarly = 0;
        while (nEarly < pa.size() && pa[nEarly].avail + 1 < lf) ++nEarly;
        // a diagonal task is the longest of its level (~20 us of factor + invert on one workgroup) and every later task of
        // the column waits for it: it keeps no early product at all
        // (TP_RMUL: eagerMinRmul >= 0 overrides -- 0 peels every product that is ready before Q_jj into a task that runs beside
        // DIAG(j), where the launch lasts ~26 us anyway, and leaves the multiplication with -Q_jj for the level after it)
        const int keep = post == TP_DIAG ? eagerMinDiag : (post == TP_RMUL && eagerMinRmul >= 0) ? eagerMinRmul : EAGER_MIN;
        if (nEarly <= (size_t)keep && !(post == TP_RMUL && eagerMinRmul == 0 && nEarly > 0)) nEarly = 0;
        bool have = initFromC;
        size_t k0 = 0;
        while (k0 < nEarly) {
            size_t k1 = (post == TP_RMUL && eagerMinRmul == 0 && nEarly <= (size_t)EAGER_MIN) ? nEarly : std::min(nEarly, k0 + CHUNK);
            while (k1 < nEarly && pa[k1].avail == pa[k1 - 1].avail) ++k1;
            TileTaskL E;
            E.t = TileTask{form, have ? 1 : 0, TP_STORE, 0, 0, sub, ldc, 0, 0, 0, c, nullptr, c};
            for (size_t k = k0; k < k1; ++k) E.prods.push_back(pa[k].p);
            E.level = pa[k1 - 1].avail + 1;
            E.row = row;
            flops += 2.0 * TILE * TILE * TILE * E.prods.size();
            out.push_back(std::move(E));
            have = true;
            k0 = k1;
        }
        TileTaskL F;
        F.t = TileTask{form, have ? 1 : 0, post, 0, 0, sub, ldc, ldq, pivotBase, 0, c, q, dst};
        for (size_t k = nEarly; k < pa.size(); ++k) F.prods.push_back(pa[k].p);
        F.level = lf;
        F.row = row;
        flops += 2.0 * TILE * TILE * TILE * (F.prods.size() + ((post == TP_ROW || post == TP_RMUL) ? 1 : 0)) +
                 (post == TP_DIAG ? 2.0 / 3.0 * TILE * TILE * TILE : 0.0);
        out.push_back(std::move(F));
        return lf;
    };
    // levels of the factorisation
    std::vector<int> lvR((size_t)nt * nt, 0), lvD(nt, 0);
    auto LR = [&](int i, int j) -> int & { return lvR[(size_t)i * nt + j]; };
    std::vector<PA> pa;
    for (int j = 0; j < nt; ++j) {
        if (!live[j]) continue;
        for (int k = 0; k < j; ++k) {
            if (!Rp(k, j)) continue;
            pa.clear();
            for (int m = 0; m < k; ++m)
                if (Rp(m, k) && Rp(m, j))
                    pa.push_back({{tile(m, k), tile(m, j), tld(m, k), tld(m, j)}, std::max(LR(m, k), LR(m, j))});
            // a pure fill-in tile holds nothing to start from: its first task starts from zero, and nobody has to clear it
            LR(k, j) = emit(k, TF_FACT, TP_ROW, tile(k, j), tld(k, j), qtile(k, k), tld(k, k), 0, pa, lvD[k], Hp(k, j));
        }
        pa.clear();
        for (int m = 0; m < j; ++m)
            if (Rp(m, j)) pa.push_back({{tile(m, j), tile(m, j), tld(m, j), tld(m, j)}, LR(m, j)});
        lvD[j] = emit(j, TF_FACT, TP_DIAG, tile(j, j), tld(j, j), nullptr, 0, j * TILE, pa, 0, true, qtile(j, j));
    }
    // levels of the inversion; lvQ(i,j) = level after which tile (i,j) of the factor buffer holds Q_ij
    std::vector<int> lvQ((size_t)nt * nt, 0);
    auto LQ = [&](int i, int j) -> int & { return lvQ[(size_t)i * nt + j]; };
    for (int j = 0; j < nt; ++j)
        if (live[j]) LQ(j, j) = lvD[j];
    for (int j = 0; j < nt; ++j) {
        if (!live[j]) continue;
        clearTiles.push_back(tile(j, j));
        clearLd.push_back(tld(j, j));
        ++qTiles;
        for (int i : qcol[j]) {
            if (i == j) continue;
            if (Hp(i, j)) {   // only the tiles the fill writes into are read before they are written
                clearTiles.push_back(tile(i, j));
                clearLd.push_back(tld(i, j));
            }
            ++qTiles;
            pa.clear();
            const bool panel = twoLevel && leafTile[i] && !leafTile[j];   // T_ij: the leaf's own columns only, stored as it is
            for (int m = i; m < j; ++m)
                if (Qp(i, m) && Rp(m, j) && !(panel && !leafTile[m]))
                    pa.push_back({{qtile(i, m), tile(m, j), tld(i, m), tld(m, j)}, std::max(LQ(i, m), LR(m, j))});
            // Q_ij = -(sum) Q_jj: the last task multiplies with Q_jj (ready after DIAG(j)); nothing else is waited for -- R_ij
            // lives in the work buffer and stays there
            if (panel) LQ(i, j) = emit(i, TF_INV, TP_STORE, qtile(i, j), tld(i, j), nullptr, 0, 0, pa, 0, false);
            else LQ(i, j) = emit(i, TF_INV, TP_RMUL, qtile(i, j), tld(i, j), qtile(j, j), tld(j, j), 0, pa, lvD[j], false);
        }
    }
    if (!balance) return;
    // ---- second pass: move the tasks that have slack out of the launches the critical path runs through ---------------
    // The levels above are as-soon-as-possible.  The chain DIAG(j-1) -> ROW(j-1, j) -> DIAG(j) paces the factorisation:
    // the launch that holds a diagonal task lasts >= ~25 us whatever else is in it, while the launch between two of them
    // is as long as everything that was scheduled into it -- mostly eager updates and tasks of the inversion, which
    // nobody needs for several levels.  Those go one level later, next to the diagonal tasks, whenever their
    // latest possible level (as-late-as-possible, same total length) allows.
    // Dependencies are re-derived from the tiles the tasks touch, in the order of the valid ASAP schedule: per tile, a
    // writer comes after the previous writer and after every reader since then, a reader after the last writer.
    const size_t n = out.size() - o0;
    std::vector<size_t> ord(n);
    for (size_t k = 0; k < n; ++k) ord[k] = o0 + k;
    std::stable_sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return out[a].level < out[b].level; });
    struct Acc {
        int writer = -1;
        std::vector<int> readers;
    };
    std::unordered_map<const double *, Acc> acc;
    std::vector<std::vector<int>> pred(n), succ(n);
    auto edge = [&](int u, int v) {
        if (u < 0 || u == v) return;
        pred[v].push_back(u);
        succ[u].push_back(v);
    };
    for (size_t oi = 0; oi < n; ++oi) {
        const int v = (int)(ord[oi] - o0);
        const TileTaskL &T = out[ord[oi]];
        auto rd = [&](const double *x) {
            Acc &a = acc[x];
            edge(a.writer, v);
            a.readers.push_back(v);
        };
        for (auto &pr : T.prods) {
            rd(pr.a);
            if (pr.b != pr.a) rd(pr.b);
        }
        if (T.t.q) rd(T.t.q);
        if (T.t.init && T.t.c != T.t.o) rd(T.t.c);
        Acc &c = acc[T.t.o];
        edge(c.writer, v);
        for (int r : c.readers) edge(r, v);
        c.writer = v;
        c.readers.clear();
    }
    int lmax = 0;
    std::vector<char> diagLevel;
    for (size_t k = 0; k < n; ++k) lmax = std::max(lmax, out[o0 + k].level);
    diagLevel.assign(lmax + 2, 0);
    for (size_t k = 0; k < n; ++k)
        if (out[o0 + k].t.post == TP_DIAG) diagLevel[out[o0 + k].level] = 1;
    std::vector<int> latest(n, lmax), fin(n, 0);
    for (size_t oi = n; oi-- > 0;) {
        const int v = (int)(ord[oi] - o0);
        for (int w : succ[v]) latest[v] = std::min(latest[v], latest[w] - 1);
    }
    for (size_t oi = 0; oi < n; ++oi) {
        const int v = (int)(ord[oi] - o0);
        TileTaskL &T = out[ord[oi]];
        int lo = T.level;
        for (int u : pred[v]) lo = std::max(lo, fin[u] + 1);
        const bool bulk = T.t.post == TP_STORE || T.t.form == TF_INV;
        if (bulk && !diagLevel[lo] && lo + 1 <= latest[v] && diagLevel[lo + 1]) ++lo;
        fin[v] = lo;
    }
    for (size_t k = 0; k < n; ++k) out[o0 + k].level = fin[k];
}

// merge the per-subdomain task lists into level order
inline void finish_tile_schedule(std::vector<TileTaskL> &all, TileSchedule &S, bool xcdGroups = true)
{
    int maxLevel = 0;
    for (auto &t : all) maxLevel = std::max(maxLevel, t.level);
    std::vector<std::vector<size_t>> byLevel(maxLevel + 1);
    for (size_t k = 0; k < all.size(); ++k) byLevel[all[k].level].push_back(k);
    S.levelStart.assign(1, 0);
    S.levelDiag.clear();
    for (int l = 1; l <= maxLevel; ++l) {
        // Order inside a level: tasks of the same subdomain and target row read the same A tiles (R_mk / Q_im), so
        // they are made neighbours ON ONE XCD -- workgroup b runs on XCD b % 8, each XCD has its own 4 MB L2 -- and their
        // shared operands come from that L2 instead of HBM.  Groups are dealt to the eight XCD lanes, heaviest first.
        auto &v = byLevel[l];
        if (!xcdGroups) {   // plain: long tasks first
            std::stable_sort(v.begin(), v.end(), [&](size_t a, size_t b) {
                auto cost = [&](const TileTaskL &t) { return (int)t.prods.size() + (t.t.post == TP_DIAG ? 4 : 0); };
                return cost(all[a]) > cost(all[b]);
            });
        } else {
        std::stable_sort(v.begin(), v.end(), [&](size_t a, size_t b) {
            if (all[a].t.sub != all[b].t.sub) return all[a].t.sub < all[b].t.sub;
            if (all[a].row != all[b].row) return all[a].row < all[b].row;
            return all[a].prods.size() > all[b].prods.size();
        });
        {
            std::vector<std::pair<size_t, size_t>> groups;   // [first, end) in v
            for (size_t i = 0; i < v.size();) {
                size_t j = i;
                while (j < v.size() && all[v[j]].t.sub == all[v[i]].t.sub && all[v[j]].row == all[v[i]].row) ++j;
                groups.push_back({i, j});
                i = j;
            }
            auto weight = [&](const std::pair<size_t, size_t> &g) {
                size_t wsum = 0;
                for (size_t i = g.first; i < g.second; ++i) wsum += all[v[i]].prods.size() + 2 + (all[v[i]].t.post == TP_DIAG ? 4 : 0);
                return wsum;
            };
            std::stable_sort(groups.begin(), groups.end(),
                             [&](const std::pair<size_t, size_t> &a, const std::pair<size_t, size_t> &b) { return weight(a) > weight(b); });
            constexpr int NX = 8;
            std::vector<size_t> lane[NX];
            size_t load[NX] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (auto &g : groups) {
                int x = 0;
                for (int y = 1; y < NX; ++y)
                    if (load[y] < load[x]) x = y;
                for (size_t i = g.first; i < g.second; ++i) lane[x].push_back(v[i]);
                load[x] += weight(g);
            }
            std::vector<size_t> order;
            size_t longest = 0;
            for (int x = 0; x < NX; ++x) longest = std::max(longest, lane[x].size());
            // position 8 n + x belongs to lane x; a lane that has run out leaves its positions to the others
            std::vector<size_t> pos(NX, 0);
            while (order.size() < v.size())
                for (int x = 0; x < NX; ++x)
                    if (pos[x] < lane[x].size()) order.push_back(lane[x][pos[x]++]);
            v.swap(order);
        }
        }
        // the diagonal-block tasks first: they go to a kernel of their own (tile_diag / tile_gemm, k_tilefactor.hip)
        std::stable_partition(v.begin(), v.end(), [&](size_t k) { return all[k].t.post == TP_DIAG; });
        int nd = 0;
        for (size_t k : v) nd += all[k].t.post == TP_DIAG;
        S.levelDiag.push_back(nd);
        for (size_t k : v) {
            TileTask t = all[k].t;
            t.first = (int)S.prods.size();
            t.nprod = (int)all[k].prods.size();
            t.p0 = all[k].prods.empty() ? TileProd{nullptr, nullptr, 0, 0} : all[k].prods[0];
            for (auto &p : all[k].prods) S.prods.push_back(p);
            S.tasks.push_back(t);
        }
        S.levelStart.push_back((int)S.tasks.size());
    }
}

// ---- dependencies of the ordered task list, for the dataflow kernel (tile_flow_kernel, k_tilefactor.hip) ------------------------
// The level schedule above synchronises with launch boundaries: every task of level l waits for ALL tasks of level l - 1.
// The dataflow kernel hands the same tasks, in the same (topological) order, to persistent workgroups and lets each wait
// only for the tasks whose tiles it touches: per tile, a reader comes after the last writer, a writer after the previous
// writer and after every reader since then (the rule of the second scheduling pass above).  dep[depPtr[v] .. depPtr[v+1])
// = the tasks task v waits for, all with smaller indices; an edge u -> v is dropped when u is already a predecessor of
// another predecessor of v (it has finished by then).
inline void build_tile_deps(const std::vector<TileTask> &tasks, const std::vector<TileProd> &prods, std::vector<int> &depPtr,
                            std::vector<int> &depIdx)
{
    struct Acc {
        int writer = -1;
        std::vector<int> readers;
    };
    std::unordered_map<const double *, Acc> acc;
    acc.reserve(tasks.size() * 2);
    const int n = (int)tasks.size();
    std::vector<std::vector<int>> pred(n);
    for (int v = 0; v < n; ++v) {
        const TileTask &T = tasks[v];
        auto rd = [&](const double *x) {
            Acc &a = acc[x];
            if (a.writer >= 0) pred[v].push_back(a.writer);
            a.readers.push_back(v);
        };
        for (int p = T.first; p < T.first + T.nprod; ++p) {
            rd(prods[p].a);
            if (prods[p].b != prods[p].a) rd(prods[p].b);
        }
        if (T.q) rd(T.q);
        if (T.init && T.c != T.o) rd(T.c);
        Acc &c = acc[T.o];
        if (c.writer >= 0) pred[v].push_back(c.writer);
        for (int r : c.readers)
            if (r != v) pred[v].push_back(r);
        c.writer = v;
        c.readers.clear();
        std::sort(pred[v].begin(), pred[v].end());
        pred[v].erase(std::unique(pred[v].begin(), pred[v].end()), pred[v].end());
    }
    depPtr.assign(1, 0);
    depIdx.clear();
    std::vector<int> mark(n, -1);
    for (int v = 0; v < n; ++v) {
        for (int w : pred[v])
            for (int u : pred[w]) mark[u] = v;
        for (int u : pred[v])
            if (mark[u] != v) depIdx.push_back(u);
        depPtr.push_back((int)depIdx.size());
    }
}

}  // namespace dotmi
