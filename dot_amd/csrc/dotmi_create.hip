// dotmi_create.hip -- dotmi_create and what it builds: mesh features, tolerance, partition maps, dissection layout, back-solve tiles, factor storage, tile schedule, patches; the host-only planning entry points; dotmi_destroy
#include "dotmi_handle.hpp"
#include "bs_tiles.hpp"

namespace dotmi {
std::string g_create_error;
}

namespace dotmi {

// Mesh::computeFeatures (Mesh.cpp:589-700), computeMassMatrix tets (:552-585)
void host_features(dotmi_handle *h)
{
    const int nV = h->nV, nT = h->nT;
    h->A.assign((size_t)9 * nT, 0.0);
    h->vol.assign(nT, 0.0);
    h->mass.assign(nV, 0.0);
    for (int e = 0; e < nT; ++e) {
        const int *t = &h->T[4 * e];
        const double *p0 = &h->Xrest[3 * t[0]], *p1 = &h->Xrest[3 * t[1]], *p2 = &h->Xrest[3 * t[2]],
                     *p3 = &h->Xrest[3 * t[3]];
        Mat3 X0;
        for (int i = 0; i < 3; ++i) {
            X0.m[i][0] = p1[i] - p0[i];
            X0.m[i][1] = p2[i] - p0[i];
            X0.m[i][2] = p3[i] - p0[i];
        }
        const double d = det3(X0), id = 1.0 / d;
        const double(*m)[3] = X0.m;
        double *R = &h->A[(size_t)9 * e];
        R[0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) * id;
        R[1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id;
        R[2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
        R[3] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) * id;
        R[4] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id;
        R[5] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
        R[6] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) * id;
        R[7] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id;
        R[8] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
        h->vol[e] = d / 3.0 / 2.0;  // signed triArea, Mesh.cpp:639
        double a[3], b[3], c[3];
        for (int i = 0; i < 3; ++i) {
            a[i] = p0[i] - p3[i];
            b[i] = p1[i] - p3[i];
            c[i] = p2[i] - p3[i];
        }
        const double vv = std::fabs(a[0] * (b[1] * c[2] - b[2] * c[1]) + a[1] * (b[2] * c[0] - b[0] * c[2]) +
                                    a[2] * (b[0] * c[1] - b[1] * c[0])) / 6.0;
        for (int k = 0; k < 4; ++k) h->mass[t[k]] += vv / 4.0;
    }
    for (int v = 0; v < nV; ++v) h->mass[v] *= h->density;
}

// Optimizer::computeCharNormSq (Optimizer.cpp:613-651)
double host_target_gres(const dotmi_handle *h)
{
    Mat3 Aw;
    double Bw[3][4];
    const double S1[3] = {1, 1, 1};
    if (h->mat == 0) spectral_blocks<0>(S1, h->mu[0], h->lam[0], 1.0, false, Aw, Bw);
    else spectral_blocks<1>(S1, h->mu[0], h->lam[0], 1.0, false, Aw, Bw);
    // with U = V = I the 9x9 matrix is exactly the 21 spectral entries (Energy.cpp:1183-1207)
    double sqH = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) sqH += Aw.m[i][j] * Aw.m[i][j];
    for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 4; ++k) sqH += Bw[c][k] * Bw[c][k];
    std::vector<double> ls(h->nV, 0.0);
    for (int e = 0; e < h->nT; ++e) {
        const int *t = &h->T[4 * e];
        for (int i = 0; i < 4; ++i) {
            const double *a = &h->Xrest[3 * t[(i + 1) % 4]], *b = &h->Xrest[3 * t[(i + 2) % 4]],
                         *c = &h->Xrest[3 * t[(i + 3) % 4]];
            const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
            const double w[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            const double cx = u[1] * w[2] - u[2] * w[1], cy = u[2] * w[0] - u[0] * w[2],
                         cz = u[0] * w[1] - u[1] * w[0];
            ls[t[i]] += 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
        }
    }
    double sql = 0;
    for (int v = 0; v < h->nV; ++v) sql += ls[v] * ls[v];
    // data0 carries exactly one fixed vertex (Mesh.cpp:592-598)
    const double cn = h->relTol * h->relTol * sqH * sql * (double)(h->nV - 1) / (double)h->nV;
    return cn * h->dtSq * h->dtSq;
}

// vertex patches (vpatches.hpp) -> device
int upload_vpatches(dotmi_handle *h, const HostVPatches &H, DevVPatches &D)
{
    D.nPatches = H.nPatches;
    D.PE = H.PE;
    D.PV = H.PV;
    D.PO = H.PO;
    D.RUN = H.RUN;
    const size_t ns = (size_t)H.nPatches * H.PE;
    std::vector<ushort4> tl(ns), ep(ns);
    std::vector<double> A(9 * ns, 0.0), mu(ns, 1.0), lam(ns, 1.0), vol(ns, 0.0), volE(ns, 0.0);
    D.nSlotsUsed = 0;
    for (size_t s = 0; s < ns; ++s) {
        tl[s] = make_ushort4(H.tl[4 * s], H.tl[4 * s + 1], H.tl[4 * s + 2], H.tl[4 * s + 3]);
        ep[s] = make_ushort4(H.epos[4 * s], H.epos[4 * s + 1], H.epos[4 * s + 2], H.epos[4 * s + 3]);
        const int e = H.elem[s];
        if (e < 0) continue;
        D.nSlotsUsed++;
        for (int k = 0; k < 9; ++k) A[(size_t)k * ns + s] = h->A[(size_t)9 * e + k];
        mu[s] = h->mu[e];
        lam[s] = h->lam[e];
        vol[s] = h->vol[e];
        volE[s] = H.eown[s] ? h->vol[e] : 0.0;
    }
    if (int rc = upload(h, &D.tl, tl)) return rc;
    if (int rc = upload(h, &D.epos, ep)) return rc;
    if (int rc = upload(h, &D.A, A)) return rc;
    bool uniform = !h->mu.empty();
    for (size_t e = 1; e < h->mu.size() && uniform; ++e) uniform = h->mu[e] == h->mu[0] && h->lam[e] == h->lam[0];
    D.mu = D.lam = nullptr;
    if (uniform) {
        D.mu0 = h->mu[0];
        D.lam0 = h->lam[0];
    } else {
        if (int rc = upload(h, &D.mu, mu)) return rc;
        if (int rc = upload(h, &D.lam, lam)) return rc;
    }
    if (int rc = upload(h, &D.vol, vol)) return rc;
    if (int rc = upload(h, &D.volE, volE)) return rc;
    if (int rc = upload(h, &D.pv_gid, H.pv_gid)) return rc;
    if (int rc = upload(h, &D.pv_cnt, H.pv_cnt)) return rc;
    if (int rc = upload(h, &D.po_cnt, H.po_cnt)) return rc;
    if (int rc = upload(h, &D.c_ptr, H.c_ptr)) return rc;
    return 0;
}

// patch lists + the element operands in patch order -> device
int upload_patches(dotmi_handle *h, const HostPatches &H, DevPatches &D)
{
    D.nPatches = H.nPatches;
    D.PE = H.PE;
    D.PV = H.PV;
    D.nSlots = H.nSlots;
    const size_t ns = (size_t)H.nPatches * H.PE;
    std::vector<ushort4> tl(ns);
    std::vector<double> A(9 * ns, 0.0), mu(ns, 1.0), lam(ns, 1.0), vol(ns, 0.0);
    D.nElem = 0;
    for (size_t s = 0; s < ns; ++s) {
        tl[s] = make_ushort4(H.tl[4 * s], H.tl[4 * s + 1], H.tl[4 * s + 2], H.tl[4 * s + 3]);
        const int e = H.elem[s];
        if (e < 0) continue;
        D.nElem++;
        for (int k = 0; k < 9; ++k) A[(size_t)k * ns + s] = h->A[(size_t)9 * e + k];
        mu[s] = h->mu[e];
        lam[s] = h->lam[e];
        vol[s] = h->vol[e];
    }
    if (int rc = upload(h, &D.tl, tl)) return rc;
    if (int rc = upload(h, &D.A, A)) return rc;
    // one material (every input deck of the reference: Mesh.cpp:741-744 fills u / lambda from one Young's modulus and
    // Poisson ratio): the element pass takes the two numbers as kernel arguments instead of 16 bytes per tet
    bool uniform = !h->mu.empty();
    for (size_t e = 1; e < h->mu.size() && uniform; ++e) uniform = h->mu[e] == h->mu[0] && h->lam[e] == h->lam[0];
    D.mu = D.lam = nullptr;
    if (uniform) {
        D.mu0 = h->mu[0];
        D.lam0 = h->lam[0];
    } else {
        if (int rc = upload(h, &D.mu, mu)) return rc;
        if (int rc = upload(h, &D.lam, lam)) return rc;
    }
    if (int rc = upload(h, &D.vol, vol)) return rc;
    if (int rc = upload(h, &D.pv_gid, H.pv_gid)) return rc;
    if (int rc = upload(h, &D.pv_slot, H.pv_slot)) return rc;
    if (int rc = upload(h, &D.pv_cnt, H.pv_cnt)) return rc;
    if (int rc = upload(h, &D.c_ptr, H.c_ptr)) return rc;
    {
        std::vector<ushort4> ep(ns);
        for (size_t s2 = 0; s2 < ns; ++s2) ep[s2] = make_ushort4(H.epos[4 * s2], H.epos[4 * s2 + 1], H.epos[4 * s2 + 2], H.epos[4 * s2 + 3]);
        if (int rc = upload(h, &D.epos, ep)) return rc;
    }
    {
        std::vector<int2> rng(H.pp_rng.size() / 2);
        for (size_t v = 0; v < rng.size(); ++v) rng[v] = make_int2(H.pp_rng[2 * v], H.pp_rng[2 * v + 1]);
        if (int rc = upload(h, &D.pp_rng, rng)) return rc;
    }
    if (int rc = dalloc(h, &D.gpart, (size_t)3 * std::max(H.nSlots, 1))) return rc;
    HIPCHECK(h, hipMemset(D.gpart, 0, sizeof(double) * 3 * (size_t)std::max(H.nSlots, 1)));
    return 0;
}

// structural non-zeros of the one-pass form (explicit inverse) on a planned layout: every row of a region from the region's first
// column to its diagonal, live columns only -- x 8 = the bytes one application streams (dotmi_step_stats.precond_bytes)
static long long one_pass_nnz(const std::vector<NdNode> &nd, const std::vector<std::vector<std::vector<int>>> &reg, int nParts)
{
    long long nnz = 0;
    const int nmax0 = nd[0].size;
    std::vector<int> usedBefore(nmax0 + 1);
    for (int ls = 0; ls < nParts; ++ls) {
        std::vector<uint8_t> live(nmax0, 0);
        for (size_t k = 0; k < nd.size(); ++k) {
            const int used = 3 * (int)reg[k][ls].size(), ro = nd_region_first_row(nd[k], used);
            std::fill(live.begin() + ro, live.begin() + ro + used, 1);
        }
        usedBefore[0] = 0;
        for (int c = 0; c < nmax0; ++c) usedBefore[c + 1] = usedBefore[c] + live[c];
        for (size_t k = 0; k < nd.size(); ++k) {
            const NdNode &N = nd[k];
            const int used = 3 * (int)reg[k][ls].size(), ro = nd_region_first_row(N, used);
            const int cb = N.a < 0 ? (ro & ~15) : N.off;
            for (int r = ro; r < ro + used; ++r) nnz += usedBefore[r + 1] - usedBefore[cb];
        }
    }
    return nnz;
}
constexpr long long TWO_LEVEL_FROM_BYTES = 240000000ll;   // one-pass bytes per application from which the two-level form is the default

int build_device_mesh(dotmi_handle *h)
{
    const int nV = h->nV, nT = h->nT;
    DevMesh &M = h->M;
    M.nV = nV;
    M.nT = nT;
    M.nTp = (nT + 63) / 64 * 64;
    // elements
    {
        std::vector<int4> T4(nT);
        for (int e = 0; e < nT; ++e) T4[e] = make_int4(h->T[4 * e], h->T[4 * e + 1], h->T[4 * e + 2], h->T[4 * e + 3]);
        if (int rc = upload(h, &M.T, T4)) return rc;
        std::vector<double> Asoa((size_t)9 * M.nTp, 0.0);
        for (int e = 0; e < nT; ++e)
            for (int k = 0; k < 9; ++k) Asoa[(size_t)k * M.nTp + e] = h->A[(size_t)9 * e + k];
        if (int rc = upload(h, &M.A, Asoa)) return rc;
        if (int rc = upload(h, &M.mu, h->mu)) return rc;
        if (int rc = upload(h, &M.lam, h->lam)) return rc;
        if (int rc = upload(h, &M.vol, h->vol)) return rc;
        if (int rc = upload(h, &M.mass, h->mass)) return rc;
        if (int rc = upload(h, &M.fixed, h->fixed)) return rc;
    }
    // adjacency incl. self
    std::vector<int> adj_ptr, adj_idx;
    build_adjacency(nV, nT, h->T.data(), adj_ptr, adj_idx);
    M.nnzb = adj_ptr[nV];
    std::vector<int> blk_row(M.nnzb);
    for (int v = 0; v < nV; ++v)
        for (int k = adj_ptr[v]; k < adj_ptr[v + 1]; ++k) blk_row[k] = v;
    auto find_block = [&](int v, int u) {
        const int *b = &adj_idx[adj_ptr[v]], *e = &adj_idx[adj_ptr[v + 1]];
        return (int)(std::lower_bound(b, e, u) - adj_idx.data());
    };
    // per-block contributions, ascending element
    std::vector<int> blk_ptr(M.nnzb + 1, 0), blk_ent((size_t)16 * nT), eblk((size_t)16 * nT);
    for (int e = 0; e < nT; ++e)
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                const int k = find_block(h->T[4 * e + a], h->T[4 * e + b]);
                eblk[(size_t)16 * e + 4 * a + b] = k;
                blk_ptr[k + 1]++;
            }
    for (int k = 0; k < M.nnzb; ++k) blk_ptr[k + 1] += blk_ptr[k];
    {
        std::vector<int> cur(blk_ptr.begin(), blk_ptr.end() - 1);
        for (int e = 0; e < nT; ++e)
            for (int ab = 0; ab < 16; ++ab) blk_ent[cur[eblk[(size_t)16 * e + ab]]++] = 16 * e + ab;
    }
    if (int rc = upload(h, &M.adj_ptr, adj_ptr)) return rc;
    if (int rc = upload(h, &M.adj_idx, adj_idx)) return rc;
    if (int rc = upload(h, &M.blk_ptr, blk_ptr)) return rc;
    if (int rc = upload(h, &M.blk_ent, blk_ent)) return rc;
    if (int rc = upload(h, &M.blk_row, blk_row)) return rc;

    // ---- subdomains (ADMMDDTimeStepper.cpp:88-262) ------------------------------------------------
    const int nP = h->nPartsAll;
    h->partVerts.assign(nP, {});
    {
        std::vector<int> mark(nV, -1);
        if (!h->vpart.empty()) {   // vertex partition given: disjoint vertex sets (block-Jacobi, LBFGS-JH)
            for (int v = 0; v < nV; ++v) h->partVerts[h->vpart[v]].push_back(v);
        } else
        for (int pI = 0; pI < nP; ++pI) {
            for (int e = 0; e < nT; ++e)
                if (h->epart[e] == pI)
                    for (int k = 0; k < 4; ++k) {
                        const int v = h->T[4 * e + k];
                        if (mark[v] != pI) {
                            mark[v] = pI;
                            h->partVerts[pI].push_back(v);
                        }
                    }
            std::sort(h->partVerts[pI].begin(), h->partVerts[pI].end());
        }
    }
    h->dup.assign(nV, 0);
    int nsmax = 0;
    for (int pI = 0; pI < nP; ++pI) {
        for (int v : h->partVerts[pI]) h->dup[v]++;
        nsmax = std::max(nsmax, 3 * (int)h->partVerts[pI].size());
    }
    // ownership: contiguous groups of parts balanced by sum n_s^2 (the back-solve cost)
    {
        std::vector<int32_t> ps(nP), first(h->world + 1);
        for (int pI = 0; pI < nP; ++pI) ps[pI] = 3 * (int32_t)h->partVerts[pI].size();
        dotmi_plan_shards(nP, ps.data(), h->world, first.data());
        h->p0 = first[h->rank];
        h->p1 = first[h->rank + 1];
        h->firstPart = first;
    }
    DevParts &P = h->P;
    P.nParts = h->p1 - h->p0;
    // ---- nested-dissection layout of the owned subdomains ---------------------------------------
    int ndLevels = h->tune.ndLevels;
    int ndMin = h->tune.ndMin;
    std::vector<std::vector<std::vector<int>>> region;  // [node][owned part] -> vertices of the leaf / separator
    {
        std::vector<std::vector<int>> sets(P.nParts);
        for (int ls = 0; ls < P.nParts; ++ls) sets[ls] = h->partVerts[h->p0 + ls];
        // two-level form of the back-solve (DevTwoLevel): the block solve as a whole (not GSDD's subdomain at
        // a time); per subdomain, so sharded subdomains take it as well.  By default where the one-pass form would stream at least 240 MB per application on its own
        // layout: the three launches around the tile kernels and the split merge cost 25-40 us per iteration, the form saves bytes
        // and a third of the factorisation -- measured (profiles/r06_two_level.txt H): bar17K (197 MB) +21 % per step, monkey
        // (124 MB) +42 %, horse7K / 8 (107 MB) +8 %; kingkong18K / 18 (245 MB) -10 %, horse7K@r1:64 (681 MB) -12 %, 1 M tets
        // (3.8 GB) -16 %.  The form wants a tree of at least four levels with regions split down to ~200 scalars (the horse at
        // three levels / 384: +4 % instead of -12 %).
        const bool eligible = !(h->flags & DOTMI_FLAG_GSDD);
        const bool userDepth = ndLevels >= 0 || getenv("DOTMI_ND_MIN") != nullptr;
        bool wantTwoLevel = eligible && h->tune.twoLevel > 0;
        int lvUsed = 0, mnUsed = 0;
        auto plan = [&](bool twoLevelDepth) {
            int lv = ndLevels, mn = ndMin;
            if (twoLevelDepth && !userDepth) {
                // sixteen leaves per subdomain where its size allows: four levels, regions split down to a fifteenth of the biggest
                // subdomain (between 128 and 256 scalars) -- 1 M tets / 256 (3.2 k dofs): 213, the same tree as with 256; 4.2 M tets /
                // 1024 (2.9 k dofs): 192, 16 leaves instead of 8: 13.8 -> 9.8 GB per application, the step -7 %; 192 k tets / 64:
                // 433 -> 325 MB, -7.5 % (profiles/r06_two_level.txt J)
                lv = std::max(4, nd_default_levels(h->partVerts));
                int nsmax = 0;
                for (auto &v : h->partVerts) nsmax = std::max(nsmax, 3 * (int)v.size());
                mn = std::min(256, std::max(128, nsmax / 15));
            } else if (lv < 0 && !getenv("DOTMI_ND_MIN")) {
                // depth and split threshold from ALL subdomains of the mesh: the same tree on every rank (nd_layout.hpp)
                nd_choose_depth(h->partVerts, nV, adj_ptr, adj_idx, h->Xrest.data(), BS_NARROW, mn, lv, mn);
            } else if (lv < 0) {
                lv = nd_default_levels(h->partVerts);
            }
            if (h->tune.fuseLog) fprintf(stderr, "dotmi: dissection: %d levels, regions split down to %d scalars\n", lv, mn);
            lvUsed = lv;
            mnUsed = mn;
            nd_plan(sets, nV, adj_ptr, adj_idx, h->Xrest.data(), lv, mn, h->nd, region);
        };
        plan(wantTwoLevel);
        if (!wantTwoLevel && eligible && h->tune.twoLevel < 0 && !h->nd.empty() && h->nd[0].a >= 0) {
            // counted over ALL subdomains of the mesh, so that every rank of a sharded run -- and the single-GPU run it is
            // compared with -- takes the same form (a rank's own factors are private, but the forms differ in rounding)
            long long nnz = 0;
            if (P.nParts == (int)h->partVerts.size()) {
                nnz = one_pass_nnz(h->nd, region, P.nParts);
            } else {
                std::vector<NdNode> treeAll;
                std::vector<std::vector<std::vector<int>>> regionAll;
                nd_plan(h->partVerts, nV, adj_ptr, adj_idx, h->Xrest.data(), lvUsed, mnUsed, treeAll, regionAll);
                if (!treeAll.empty()) nnz = one_pass_nnz(treeAll, regionAll, (int)h->partVerts.size());
            }
            if (8 * nnz >= TWO_LEVEL_FROM_BYTES) {
                wantTwoLevel = true;
                if (!userDepth) plan(true);
            }
        }
        h->twoLevel = wantTwoLevel && !h->nd.empty() && h->nd[0].a >= 0;   // (a tree that has separators at all)
        if (h->twoLevel) nd_relayout_leaves_first(h->nd);
    }
    P.nmax = h->nd[0].size;
    h->tileMode = P.nParts > 0;   // (a rank without subdomains plans nothing)
    // per part: padded position of every local vertex, tiles of the back-solve, structural non-zeros
    h->partPos.assign(P.nParts, {});
    std::vector<int> dofmap((size_t)P.nParts * P.nmax, -1);
    // the job table of the back-solve launches (bs_tiles.hpp): tiles per region, wide / narrow / packs, heavy first
    BsTilePlan TP;
    plan_backsolve_tiles(h->nd, [&](int k, int ls) { return 3 * (int)region[k][ls].size(); }, P.nParts, P.nmax,
                         BsTileRules{h->tune.tileRows, h->tune.tileRowsLong, h->tune.tilePasses, h->tune.wavePacks}, TP);
    std::vector<int4> &tiles = TP.tiles;
    std::vector<std::vector<int2>> &ranges = TP.ranges;
    h->precond_bytes = 0;
    int64_t nnzX = 0;
    for (int ls = 0; ls < P.nParts; ++ls) {
        const auto &pv = h->partVerts[h->p0 + ls];
        std::unordered_map<int, int> posOf;
        posOf.reserve(pv.size() * 2);
        std::vector<int> usedBefore(P.nmax + 1, 0);  // number of live columns before a padded position
        std::vector<uint8_t> live(P.nmax, 0);
        for (size_t nd = 0; nd < h->nd.size(); ++nd) {
            const NdNode &N = h->nd[nd];
            const auto &rv = region[nd][ls];
            const int ro = nd_region_first_row(N, 3 * (int)rv.size());
            for (size_t k = 0; k < rv.size(); ++k) {
                posOf[rv[k]] = ro + 3 * (int)k;
                for (int d = 0; d < 3; ++d) {
                    dofmap[(size_t)ls * P.nmax + ro + 3 * k + d] = 3 * rv[k] + d;
                    live[ro + 3 * k + d] = 1;
                }
            }
        }
        for (int c = 0; c < P.nmax; ++c) usedBefore[c + 1] = usedBefore[c] + live[c];
        h->partPos[ls].resize(pv.size());
        for (size_t i = 0; i < pv.size(); ++i) h->partPos[ls][i] = posOf.at(pv[i]);
        // structural non-zeros: every row of a region from the region's first column to the diagonal, live columns only
        for (size_t nd = 0; nd < h->nd.size(); ++nd) {
            const NdNode &N = h->nd[nd];
            const int used = 3 * (int)region[nd][ls].size();
            const int ro = nd_region_first_row(N, used);
            const int cb = N.a < 0 ? (ro & ~15) : N.off;
            for (int r = ro; r < ro + used; ++r) nnzX += usedBefore[r + 1] - usedBefore[cb];
        }
    }
    // every structural non-zero of the inverse factors is streamed once per back-solve
    h->precond_bytes = nnzX * 8;
    P.nbmax = 1;
    for (auto &r : ranges) P.nbmax = std::max(P.nbmax, (int)r.size());
    std::vector<int2> trange((size_t)std::max(P.nParts, 1) * P.nbmax, make_int2(0, 0));
    for (int ls = 0; ls < P.nParts; ++ls) std::copy(ranges[ls].begin(), ranges[ls].end(), trange.begin() + (size_t)ls * P.nbmax);
    std::vector<int4> &tilesByPart = TP.tilesByPart, &ltilesByPart = TP.ltilesByPart, &ltiles = TP.ltiles;
    std::vector<int2> &lworkByPart = TP.lworkByPart, &lwork = TP.lwork;
    h->partTilePtr = TP.partTilePtr;
    h->partLworkPtr = TP.partLworkPtr;
    P.maxTileLen = TP.maxTileLen;
    P.maxChunks = TP.maxChunks;
    P.ntiles = TP.ntiles;
    P.ntilesWide = TP.ntilesWide;
    P.nquad = TP.nquad;
    if (h->tune.fuseLog)
        fprintf(stderr, "dotmi: back-solve: %d one-tile jobs (%d wide), %d packs of four small tiles%s\n", P.ntiles, P.ntilesWide,
                P.nquad, TP.shallow ? ", shallow launch: long rows in 4-pass tiles" : "");
    P.nltiles = (int)ltiles.size();
    P.nlwork = (int)lwork.size();
    // merge lists (owned parts only)
    std::vector<int> vp_ptr(nV + 1, 0), vp_off;
    {
        for (int ls = 0; ls < P.nParts; ++ls)
            for (int v : h->partVerts[h->p0 + ls]) vp_ptr[v + 1]++;
        for (int v = 0; v < nV; ++v) vp_ptr[v + 1] += vp_ptr[v];
        vp_off.resize(vp_ptr[nV]);
        std::vector<int> cur(vp_ptr.begin(), vp_ptr.end() - 1);
        for (int ls = 0; ls < P.nParts; ++ls) {
            const auto &pv = h->partVerts[h->p0 + ls];
            for (int i = 0; i < (int)pv.size(); ++i) vp_off[cur[pv[i]]++] = ls * P.nmax + h->partPos[ls][i];
        }
    }
    // ---- factor storage: 64-row blocks (RowTile) ---------------------------------------------------------------
    const int ntl = P.nmax / 64;
    std::vector<RowTile> rtab((size_t)std::max(P.nParts, 1) * ntl, RowTile{-1, 0, 0});
    std::vector<long long> rtOff(rtab.size(), -1);
    std::vector<int> rtLd(rtab.size(), 0), rtC0(rtab.size(), 0);
    size_t wTotal = 0;
    {
        // first column a row of the layout can have non-zero: that of its tree node
        std::vector<int> nodeC0(P.nmax, 0);
        for (const NdNode &N : h->nd) {
            if (N.a < 0)
                for (int r = N.off; r < N.off + N.size; ++r) nodeC0[r] = N.off;
            else
                for (int r = N.offS; r < N.offS + N.sizeS; ++r) nodeC0[r] = N.off;
        }
        for (int ls = 0; ls < P.nParts; ++ls)
            for (int J = 0; J < ntl; ++J) {
                RowTile &R = rtab[(size_t)ls * ntl + J];
                {
                    bool live = false;
                    for (int r = 64 * J; r < 64 * J + 64 && !live; ++r) live = dofmap[(size_t)ls * P.nmax + r] >= 0;
                    if (!live) continue;   // identity padding only: nothing stored, nothing read
                    const int c0 = nodeC0[64 * J];
                    R = RowTile{(long long)wTotal, 64 * (J + 1) - c0, c0};
                    wTotal += (size_t)64 * R.ld;
                }
                rtOff[(size_t)ls * ntl + J] = R.off;
                rtLd[(size_t)ls * ntl + J] = R.ld;
                rtC0[(size_t)ls * ntl + J] = R.c0;
            }
        // the factors are the one allocation that grows with the square of the subdomain size: refuse what cannot fit
        // instead of failing somewhere inside hipMalloc
        size_t freeB = 0, totalB = 0;
        HIPCHECK(h, hipMemGetInfo(&freeB, &totalB));
        const double need = 8.0 * (double)wTotal * 2.1;   // + the work buffer of the factorisation
        if (need > 0.9 * (double)freeB) {
            h->err = "the subdomain factors need " + std::to_string((long long)(need / 1e9)) + " GB (" + std::to_string(P.nParts) +
                     " subdomains, padded size " + std::to_string(P.nmax) + "), more than the free HBM: use more subdomains";
            return DOTMI_E_INVALID;
        }
    }
    // two-level form: a separator's row blocks store the LEAF columns of their sub-tree in a second range (their main range starts
    // at the sub-tree's first separator column)
    std::vector<long long> rtOffM(rtab.size(), -1);
    std::vector<int> rtLdM(rtab.size(), 0), rtC0M(rtab.size(), 0);
    std::vector<uint8_t> leafTile(ntl, 0);
    if (h->twoLevel) {
        std::vector<int> nodeOfRow(P.nmax, -1);
        for (size_t k = 0; k < h->nd.size(); ++k) {
            const NdNode &N = h->nd[k];
            if (N.a < 0)
                for (int r = N.off; r < N.off + N.size; ++r) leafTile[r / 64] = 1;
            else
                for (int r = N.offS; r < N.offS + N.sizeS; ++r) nodeOfRow[r] = (int)k;
        }
        for (int ls = 0; ls < P.nParts; ++ls)
            for (int J = 0; J < ntl; ++J) {
                const size_t at = (size_t)ls * ntl + J;
                if (rtab[at].off < 0 || nodeOfRow[64 * J] < 0) continue;
                const NdNode &N = h->nd[nodeOfRow[64 * J]];
                rtOffM[at] = (long long)wTotal;
                // (+ 16: a leaf range is a multiple of 64 columns, often a power of two -- consecutive rows of a panel would then
                // start on the same HBM channels)
                rtLdM[at] = N.endL - N.offL + 16;
                rtC0M[at] = N.offL;
                wTotal += (size_t)64 * rtLdM[at];
            }
        size_t freeB = 0, totalB = 0;
        HIPCHECK(h, hipMemGetInfo(&freeB, &totalB));
        if (8.0 * (double)wTotal * 2.1 > 0.9 * (double)freeB) {
            h->err = "the two-level factors do not fit the free HBM";
            return DOTMI_E_INVALID;
        }
    }
    h->rtOff = rtOff;
    h->rtLd = rtLd;
    h->rtC0 = rtC0;
    h->rtOffM = rtOffM;
    h->rtLdM = rtLdM;
    h->rtC0M = rtC0M;
    h->wTotal = wTotal;
    // offset in W of (memory row r, column c) of owned subdomain ls, or -1 when that place is not stored
    auto waddr = [&](int ls, int r, int c) -> long long {
        const size_t at = (size_t)ls * ntl + (r >> 6);
        const RowTile &R = rtab[at];
        if (R.off < 0) return -1;
        if (c < R.c0) {
            if (rtOffM[at] < 0 || c < rtC0M[at] || c >= rtC0M[at] + rtLdM[at]) return -1;
            return rtOffM[at] + (long long)(r & 63) * rtLdM[at] + (c - rtC0M[at]);
        }
        if (c >= R.c0 + R.ld) return -1;
        return R.off + (long long)(r & 63) * R.ld + (c - R.c0);
    };
    // dense fill list: per scalar of every 3x3 block of the principal sub-matrix
    std::vector<long long> fill_dst, pad_dst;
    std::vector<int> fill_src;
    std::vector<int4> fillBlk;   // (owned subdomain, memory row, memory column) of the blocks' corners, for the tile pattern
    {
        std::vector<int> g2p(nV, -1);
        for (int ls = 0; ls < P.nParts; ++ls) {
            const auto &pv = h->partVerts[h->p0 + ls];
            for (int i = 0; i < (int)pv.size(); ++i) g2p[pv[i]] = h->partPos[ls][i];
            for (int i = 0; i < (int)pv.size(); ++i) {
                const int v = pv[i];
                for (int k = adj_ptr[v]; k < adj_ptr[v + 1]; ++k) {
                    const int j = g2p[adj_idx[k]];
                    if (j < 0) continue;
                    const int r0 = h->partPos[ls][i];
                    for (int rc = 0; rc < 9; ++rc) fill_dst.push_back(waddr(ls, r0 + rc / 3, j + rc % 3));
                    fill_src.push_back(k);
                    fillBlk.push_back(make_int4(ls, r0, j, 0));
                }
            }
            for (int r = 0; r < P.nmax; ++r)
                if (dofmap[(size_t)ls * P.nmax + r] < 0) {
                    const long long a = waddr(ls, r, r);
                    if (a >= 0) pad_dst.push_back(a);
                }
            for (int v : pv) g2p[v] = -1;
        }
    }
    P.nfill = (int)fill_src.size();
    P.npad = (int)pad_dst.size();
    if (int rc = upload(h, &P.dofmap, dofmap)) return rc;
    if (int rc = upload(h, &P.tile, tiles)) return rc;
    if (int rc = upload(h, &P.tileByPart, tilesByPart)) return rc;
    if (int rc = upload(h, &P.ltileByPart, ltilesByPart)) return rc;
    if (int rc = upload(h, &P.lworkByPart, lworkByPart)) return rc;
    if (int rc = upload(h, &P.ltile, ltiles)) return rc;
    if (int rc = upload(h, &P.lwork, lwork)) return rc;
    if (int rc = dalloc(h, &P.tdots, (size_t)std::max(P.nltiles, 1) * P.maxChunks * 64)) return rc;
    HIPCHECK(h, hipMemset(P.tdots, 0, sizeof(double) * (size_t)std::max(P.nltiles, 1) * P.maxChunks * 64));
    if (int rc = upload(h, &P.trange, trange)) return rc;
    {
        // reduce_partial_p: the tiles of a part that hold a group of 16 columns, ascending (a tile's range starts on a multiple of 16)
        const int ng = P.nmax / 16;
        std::vector<int> rptr((size_t)std::max(P.nParts, 1) * (ng + 1) , 0), ridx;
        for (int ls = 0; ls < P.nParts; ++ls) {
            std::vector<std::vector<int>> lists(ng);
            for (size_t b = 0; b < ranges[ls].size(); ++b)
                for (int g = ranges[ls][b].x / 16; g <= (ranges[ls][b].y - 1) / 16 && g < ng; ++g)
                    lists[g].push_back((int)b | (std::min(16, ranges[ls][b].y - 16 * g) << 24));   // tile | columns of the group it holds
            for (int g = 0; g < ng; ++g) {
                rptr[(size_t)ls * (ng + 1) + g] = (int)ridx.size();
                ridx.insert(ridx.end(), lists[g].begin(), lists[g].end());
            }
            rptr[(size_t)ls * (ng + 1) + ng] = (int)ridx.size();
        }
        if (ridx.empty()) ridx.push_back(0);
        if (int rc = upload(h, &P.rp_ptr, rptr)) return rc;
        if (int rc = upload(h, &P.rp_idx, ridx)) return rc;
    }
    if (int rc = upload(h, &P.vp_ptr, vp_ptr)) return rc;
    if (int rc = upload(h, &P.vp_off, vp_off)) return rc;
    {
        // merge straight from the tile partials (merge_tiles_kernel): per global scalar dof the ppart entries that make
        // up its value -- subdomain after subdomain (vp order), inside a subdomain the tiles that hold the column in
        // tile order; the first entry of a subdomain is stored complemented.  Same sums, same order as
        // reduce_partial_p + merge.
        P.mt_ptr = nullptr;
        P.mt_ent = nullptr;
        P.mt_wave = nullptr;
        P.mt_il = nullptr;
        const long long ppartN = (long long)P.nParts * P.nbmax * P.nmax;
        // Big meshes: the walk over a dof's ~20 tile partials is a walk over scattered 8-byte words and 4-byte list entries
        // (1 M tets: 75 us per iteration at 0.26 of the HBM peak); the two-launch form reads the partials coalesced in the
        // subdomains' own order and gathers one 24-byte triple per (vertex, subdomain).  Small meshes keep the one launch.
        P.splitMerge = h->tune.splitMerge >= 0 ? (h->tune.splitMerge != 0) : (3ll * nV >= 400000 || ppartN >= (1ll << 31));
        if (h->twoLevel) P.splitMerge = 1;   // (the leaves' results are finished on the per-subdomain sums psub)
        const bool lists = !P.splitMerge && ppartN < (1ll << 31) && !(h->flags & DOTMI_FLAG_GSDD);
        if (!(h->flags & DOTMI_FLAG_GSDD)) {
            std::vector<int> mp(lists ? (size_t)3 * nV + 1 : 0, 0), ment;
            long long count = 0;
            for (int v = 0; v < nV; ++v)
                for (int d = 0; d < 3; ++d) {
                    for (int k = vp_ptr[v]; k < vp_ptr[v + 1]; ++k) {
                        const int ls = vp_off[k] / P.nmax, col = vp_off[k] % P.nmax + d;
                        bool first = true;
                        for (size_t b = 0; b < ranges[ls].size(); ++b)
                            if (col >= ranges[ls][b].x && col < ranges[ls][b].y) {
                                ++count;
                                if (lists) {
                                    const int off = (int)(((long long)ls * P.nbmax + (long long)b) * P.nmax + col);
                                    ment.push_back(first ? ~off : off);
                                }
                                first = false;
                            }
                    }
                    if (lists) mp[(size_t)3 * v + d + 1] = (int)ment.size();
                }
            // Few subdomains with long rows in short tiles (bunny5K: 8-16 rows per tile): a column is covered by dozens of tiles,
            // ~30 scattered partials per dof against ~10 on bar17K -- there too the coalesced within-subdomain sum first is the
            // shorter way (bunny5K 1.395 -> 1.367 ms per step; the stiff monkey, 8 per dof, loses 3 % with it)
            const bool longLists = h->tune.splitMerge < 0 && count >= 24ll * 3 * nV;
            if (lists && longLists) {
                P.splitMerge = 1;
            } else if (lists) {
                if (int rc = upload(h, &P.mt_ptr, mp)) return rc;
                if (int rc = upload(h, &P.mt_ent, ment)) return rc;
                // the lists once more, interleaved by wavefront (DevParts::mt_il): what the merge kernels walk when they visit
                // every dof (the owner exchange's vertex lists keep the CSR walk)
                const int n3 = 3 * nV, nw = (n3 + 63) / 64;
                std::vector<int2> mw(nw);
                std::vector<int> il;
                for (int w = 0; w < nw; ++w) {
                    int L = 0;
                    for (int l = 0; l < 64 && 64 * w + l < n3; ++l) L = std::max(L, mp[64 * w + l + 1] - mp[64 * w + l]);
                    mw[w] = make_int2((int)il.size(), L);
                    il.resize(il.size() + (size_t)64 * L, MT_PAD);
                    for (int l = 0; l < 64 && 64 * w + l < n3; ++l)
                        for (int q = 0, e = mp[64 * w + l]; e < mp[64 * w + l + 1]; ++q, ++e) il[(size_t)mw[w].x + 64 * q + l] = ment[e];
                }
                if (il.empty()) il.push_back(MT_PAD);
                if (il.size() < (size_t)1 << 31) {
                    if (int rc = upload(h, &P.mt_wave, mw)) return rc;
                    if (int rc = upload(h, &P.mt_il, il)) return rc;
                }
            }
            if (h->tune.fuseLog) {
                int lmax = 0;
                long long over24 = 0;
                for (size_t k = 0; k + 1 < mp.size(); ++k) {
                    lmax = std::max(lmax, mp[k + 1] - mp[k]);
                    over24 += mp[k + 1] - mp[k] > 24;
                }
                fprintf(stderr, "dotmi: merge: %.1f tile partials per dof (longest list %d, %lld lists beyond 24) -> %s\n",
                        (double)count / std::max(1, 3 * nV), lmax, over24,
                        P.splitMerge ? "sum per subdomain, then gather (split)" : "one walk over the list");
            }
            h->mergeEntries = count;   // tile partials one merge reads (either form)
        } else {
            P.splitMerge = 0;
        }
    }
    if (int rc = upload(h, &P.dup, h->dup)) return rc;
    if (int rc = upload(h, &P.fill_dst, fill_dst)) return rc;
    if (int rc = upload(h, &P.fill_src, fill_src)) return rc;
    if (int rc = upload(h, &P.pad_dst, pad_dst)) return rc;
    if (int rc = dalloc(h, &P.W, std::max<size_t>(wTotal, 64))) return rc;
    if (int rc = upload(h, &P.rt, rtab)) return rc;
    // ---- tile schedule of the factorisation (tile_factor.hpp) ------------------------------------------------
    if (h->tileMode) {
        const int nt = P.nmax / TILE;
        std::vector<std::vector<uint8_t>> live(P.nParts, std::vector<uint8_t>(nt, 0)), pat(P.nParts);
        for (int ls = 0; ls < P.nParts; ++ls) {
            for (int r = 0; r < P.nmax; ++r)
                if (dofmap[(size_t)ls * P.nmax + r] >= 0) live[ls][r / TILE] = 1;
            pat[ls].assign((size_t)nt * nt, 0);
        }
        for (const int4 &fb : fillBlk) {
            const int ls = fb.x, r0 = fb.y, c0 = fb.z;   // memory row / column of the 3x3 block's corner
            for (int a = 0; a < 3; a += 2)
                for (int b = 0; b < 3; b += 2) {
                    const int I = (c0 + b) / TILE, J = (r0 + a) / TILE;   // column-major element (c0+b, r0+a)
                    if (I <= J) pat[ls][(size_t)I * nt + J] = 1;
                }
        }
        // eager partial updates shorten the launches of a latency-bound factorisation (few subdomains) and cost tile
        // traffic in a throughput-bound one (measured: profiles/r03_factor_tiles.txt)
        // (round 4: with very few tile columns in total -- bunny5K: 8 x 32 -- the chain of dependent tasks is all there is, and
        // the dataflow launch runs finer eager tasks at no barrier cost: 2 / 2 there, factor 0.43 -> 0.39 ms; horse7K, 8 x 47,
        // keeps 4 / 4)
        const bool tiny = (long long)P.nParts * nt <= 320;
        const int eagerMin = h->tune.tileEagerMin > 0 ? h->tune.tileEagerMin : (tiny ? 2 : P.nParts <= 64 ? 4 : 8);
        const int eagerChunk = tiny ? 2 : P.nParts <= 64 ? 4 : 8;   // early products per eager task
        // the last task of a Q tile (sum, then the multiplication with -Q_jj): up to 64 subdomains it keeps ONE early product and
        // hands the others to a task that runs beside DIAG(j) -- the launch between two diagonal launches is then as short as
        // before round 5 (bar17K 1.125 -> 1.077 ms); above, where every launch is several rounds of workgroups, it keeps them
        // like any other task and saves the partial sum's round trip (1 M tets 15.5 -> 14.5 ms)
        const int eagerMinRmul = h->tune.tileEagerMinRmul >= -1 && getenv("DOTMI_TILE_EAGER_MIN_RMUL") ? h->tune.tileEagerMinRmul
                                                                                                          : (P.nParts <= 64 ? 1 : -1);
        // the work buffer (H filled in, R in place of it): same layout as the factor buffer W, which only ever holds Q
        if (int rc = dalloc(h, &h->W2, std::max<size_t>(wTotal, 64))) return rc;
        TileSchedule S;
        {
            std::vector<TileTaskL> all;
            size_t sn = 0;
            for (int ls = 0; ls < P.nParts; ++ls)
                plan_subdomain_tiles(ls, nt, P.W, &rtOff[(size_t)ls * nt], &rtLd[(size_t)ls * nt], &rtC0[(size_t)ls * nt],
                                     live[ls], pat[ls], h->W2, sn, all, S.clearTiles, S.clearLd, S.flops, S.qTiles,
                                     eagerMin, eagerChunk, 0, true, eagerMinRmul,
                                     h->twoLevel ? &rtOffM[(size_t)ls * nt] : nullptr, h->twoLevel ? &rtLdM[(size_t)ls * nt] : nullptr,
                                     h->twoLevel ? &rtC0M[(size_t)ls * nt] : nullptr, h->twoLevel ? leafTile.data() : nullptr);
            finish_tile_schedule(all, S);
        }
        if (int rc = upload(h, &h->ttasks, S.tasks)) return rc;
        if (int rc = upload(h, &h->tprods, S.prods)) return rc;
        if (int rc = upload(h, &h->tclear, S.clearTiles)) return rc;
        if (int rc = upload(h, &h->tclearLd, S.clearLd)) return rc;
        h->nTclear = (int)S.clearTiles.size();
        h->tlevelStart = S.levelStart;
        h->tlevelDiag = S.levelDiag;
        h->tileSplit = h->tune.tileSplit >= 0 ? h->tune.tileSplit != 0 : P.nParts > 64;
        h->nTtasks = (int)S.tasks.size();
        // Dataflow or levels (profiles/r04_factor_flow.txt): per task the dataflow launch pays a ticket, a look at its
        // dependencies' flags and write-through stores, and it runs the level kernel's 77 KB workgroups -- it wins where the
        // levels are launches of less than one round of workgroups, i.e. the chain of dependent tasks paces the phase
        // (bunny5K / 8 subdomains: 217 tasks per level, 0.57 -> 0.41 ms), and loses where the levels are several rounds
        // (bar17K / 32: 1000 per level, 1.11 -> 1.21 ms; 1 M tets: 15 -> 23 ms).
        const size_t nLevels = std::max<size_t>(S.levelStart.size() - 1, 1);
        h->tileFlow = !S.tasks.empty() &&
                      (h->tune.tileFlow > 0 || (h->tune.tileFlow < 0 && S.tasks.size() / nLevels <= 512));
        // the diagonal tasks' per-lane bottom steps (k_tilefactor.hip, block_chol_inv<N, FAST>): every layout (DOTMI_FAST_DIAG=0: the
        // one-row-per-lane base of round 3); the 256-thread level kernel keeps the old base, and then so does the dataflow launch
        h->fastDiag = h->tune.fastDiag != 0;
        if (h->tileFlow) {
            std::vector<int> depPtr, depIdx;
            build_tile_deps(S.tasks, S.prods, depPtr, depIdx);
            if (depIdx.empty()) depIdx.push_back(0);
            if (int rc = upload(h, &h->tdepPtr, depPtr)) return rc;
            if (int rc = upload(h, &h->tdepIdx, depIdx)) return rc;
            if (int rc = dalloc(h, &h->tdone, S.tasks.size())) return rc;
            if (int rc = dalloc(h, &h->tnext, 2)) return rc;
            HIPCHECK(h, hipMemset(h->tdone, 0, sizeof(int) * S.tasks.size()));
            HIPCHECK(h, hipMemset(h->tnext, 0, sizeof(int) * 2));
            hipDeviceProp_t prop;
            HIPCHECK(h, hipGetDeviceProperties(&prop, h->device));
            h->tileFlowWg = 2 * prop.multiProcessorCount;
            h->tileSplit = false;
            if (h->tune.fuseLog)
                fprintf(stderr, "dotmi: tile dataflow: %zu tasks, %zu dependencies, %d workgroups\n", S.tasks.size(), depIdx.size(),
                        h->tileFlowWg);
        }
        if (h->tileSplit) {
            HIPCHECK(h, hipStreamCreateWithFlags(&h->stDiag, hipStreamNonBlocking));
            h->tFork.resize(S.levelDiag.size());
            h->tJoin.resize(S.levelDiag.size());
            for (auto &e : h->tFork) HIPCHECK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            for (auto &e : h->tJoin) HIPCHECK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        h->tileFlops = S.flops;
        if (h->tune.fuseLog)
            fprintf(stderr, "dotmi: tile schedule: %zu tasks, %zu products, %zu levels, %lld Q tiles, %.1f GF\n", S.tasks.size(),
                    S.prods.size(), S.levelStart.size() - 1, S.qTiles, S.flops / 1e9);
    }
    if (int rc = dalloc(h, &P.ppart, (size_t)P.nParts * P.nbmax * P.nmax)) return rc;
    if (int rc = dalloc(h, &P.psub, (size_t)P.nParts * P.nmax)) return rc;
    if (int rc = dalloc(h, &P.rpad, (size_t)P.nParts * P.nmax + 8)) return rc;
    HIPCHECK(h, hipMemset(P.rpad, 0, sizeof(double) * ((size_t)P.nParts * P.nmax + 8)));
    P.tl = DevTwoLevel{};
    if (h->twoLevel) {
        // panels: per (subdomain, leaf) the separator vertices next to the leaf (a mesh edge into it; no fill path leaves a leaf),
        // rows ascending in the layout
        std::vector<int4> panel;
        std::vector<long long> rowBase, rowDst;
        std::vector<int> rowPos, gIdx;
        long long packedN = 0;
        std::vector<std::vector<int>> sub((size_t)P.nParts * P.nmax);
        std::vector<int> leafOf(nV, -1), sepPos(nV, -1);
        long long panelBytes = 0;
        int maxRows = 0, maxCols = 0;
        for (int ls = 0; ls < P.nParts; ++ls) {
            for (size_t k = 0; k < h->nd.size(); ++k) {
                const NdNode &N = h->nd[k];
                const auto &rv = region[k][ls];
                const int ro = nd_region_first_row(N, 3 * (int)rv.size());
                for (size_t q = 0; q < rv.size(); ++q) {
                    if (N.a < 0) leafOf[rv[q]] = (int)k;
                    else sepPos[rv[q]] = ro + 3 * (int)q;
                }
            }
            for (size_t k = 0; k < h->nd.size(); ++k) {
                const NdNode &N = h->nd[k];
                if (N.a >= 0 || region[k][ls].empty()) continue;
                const auto &rv = region[k][ls];
                std::vector<int> rows;   // padded positions of the coupled separator vertices
                for (int v : rv)
                    for (int e = adj_ptr[v]; e < adj_ptr[v + 1]; ++e)
                        if (sepPos[adj_idx[e]] >= 0) rows.push_back(sepPos[adj_idx[e]]);
                std::sort(rows.begin(), rows.end());
                rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
                if (rows.empty()) continue;
                // (the kernels take two columns per lane: the panel starts on an even column -- one column of the leaf's identity
                // padding in front when its first live column is odd: zeros in M and in the right-hand side -- and ends at the
                // leaf's end, a multiple of 64)
                const int c0 = nd_region_first_row(N, 3 * (int)rv.size()) & ~1, used = N.off + N.size - c0;
                panel.push_back(make_int4((int)rowBase.size(), 3 * (int)rows.size(), ls * P.nmax + c0, used));
                maxRows = std::max(maxRows, 3 * (int)rows.size());
                maxCols = std::max(maxCols, used);
                panelBytes += 8ll * 3 * (long long)rows.size() * used;
                for (int rp : rows)
                    for (int d = 0; d < 3; ++d) {
                        const long long a = waddr(ls, rp + d, c0);
                        if (a < 0) {
                            h->err = "two-level layout: a panel row has no storage";
                            return DOTMI_E_INVALID;
                        }
                        sub[(size_t)ls * P.nmax + rp + d].push_back((int)rowBase.size());
                        rowBase.push_back(a);
                        rowDst.push_back(packedN);
                        packedN += used;
                        rowPos.push_back(ls * P.nmax + rp + d);
                    }
            }
            for (int v : h->partVerts[h->p0 + ls]) leafOf[v] = sepPos[v] = -1;
        }
        std::vector<int> gPtr((size_t)P.nParts * P.nmax + 1, 0);
        for (size_t i = 0; i < sub.size(); ++i) {
            gPtr[i] = (int)gIdx.size();
            gIdx.insert(gIdx.end(), sub[i].begin(), sub[i].end());
        }
        gPtr[sub.size()] = (int)gIdx.size();
        if (gIdx.empty()) gIdx.push_back(0);
        if (panel.empty()) panel.push_back(make_int4(0, 0, 0, 0));
        if (rowBase.empty()) {
            rowBase.push_back(0);
            rowDst.push_back(0);
            rowPos.push_back(0);
        }
        if (maxRows > 7000) {   // (twolevel_backward_kernel keeps a panel's p_G entries in LDS)
            h->err = "two-level form: a leaf couples to " + std::to_string(maxRows) + " separator rows (limit 7000): split the regions "
                     "further (DOTMI_ND_LEVELS / DOTMI_ND_MIN) or use DOTMI_TWO_LEVEL=0";
            return DOTMI_E_INVALID;
        }
        std::vector<int2> items;
        for (size_t q = 0; q < panel.size(); ++q)
            for (int k0 = 0; k0 < panel[q].y; k0 += 8) items.push_back(make_int2((int)q, k0));
        if (items.empty()) items.push_back(make_int2(0, 0));
        int2 *dItem = nullptr;
        if (int rc = upload(h, &dItem, items)) return rc;
        P.tl.item = dItem;
        P.tl.nItems = maxRows > 0 ? (int)items.size() : 0;
        int4 *dPanel = nullptr;
        long long *dBase = nullptr, *dDst = nullptr;
        if (int rc = upload(h, &dDst, rowDst)) return rc;
        if (int rc = dalloc(h, &P.tl.packed, (size_t)std::max<long long>(packedN, 2))) return rc;
        int *dPos = nullptr, *dPtr = nullptr, *dIdx = nullptr;
        P.tl.nPanels = maxRows > 0 ? (int)panel.size() : 0;
        if (int rc = upload(h, &dPanel, panel)) return rc;
        if (int rc = upload(h, &dBase, rowBase)) return rc;
        if (int rc = upload(h, &dPos, rowPos)) return rc;
        if (int rc = upload(h, &dPtr, gPtr)) return rc;
        if (int rc = upload(h, &dIdx, gIdx)) return rc;
        if (int rc = dalloc(h, &P.tl.cbuf, rowBase.size())) return rc;
        if (int rc = dalloc(h, &P.tl.rpad2, (size_t)P.nParts * P.nmax + 8)) return rc;
        HIPCHECK(h, hipMemset(P.tl.rpad2, 0, sizeof(double) * ((size_t)P.nParts * P.nmax + 8)));
        P.tl.panel = dPanel;
        P.tl.rowSrc = dBase;
        P.tl.rowBase = dDst;
        P.tl.rowPos = dPos;
        P.tl.gPtr = dPtr;
        P.tl.gIdx = dIdx;
        P.tl.maxRows = maxRows;
        P.tl.maxCols = maxCols;
        P.tl.on = 1;
        h->precond_bytes += 2 * panelBytes;   // (the panels are streamed twice per application)
        if (h->tune.fuseLog)
            fprintf(stderr, "dotmi: two-level back-solve: %d panels (at most %d rows x %d columns), %.1f MB of panels (x 2), %.1f MB in all\n",
                    P.tl.nPanels, maxRows, maxCols, panelBytes / 1e6, h->precond_bytes / 1e6);
    }
    if (int rc = dalloc(h, &h->info_dev, (size_t)std::max(P.nParts, 1))) return rc;
    HIPCHECK(h, hipHostMalloc((void **)&h->h_info, sizeof(int) * std::max(P.nParts, 1)));
    memset(h->h_info, 0, sizeof(int) * std::max(P.nParts, 1));

    // element ownership + inertia vertex slice
    if (h->shardElems) {
        std::vector<int> el;
        for (int e = 0; e < nT; ++e)
            if (h->epart[e] >= h->p0 && h->epart[e] < h->p1) el.push_back(e);
        h->nOwnElem = (int)el.size();
        if (int rc = upload(h, &h->elist, el)) return rc;
        h->v0 = (int)((long long)nV * h->rank / h->world);
        h->v1 = (int)((long long)nV * (h->rank + 1) / h->world);
    } else {
        h->elist = nullptr;
        h->nOwnElem = nT;
        h->v0 = 0;
        h->v1 = nV;
    }
    // ---- sharded refresh lists ------------------------------------------------------------------------------------
    // The block rows this rank reads: the rows of its subdomains' vertices (dense fill of H_s = R_s H R_s^T) and its
    // slice [v0, v1) of the SpMV.  Their blocks are sums over the elements incident to the row vertex, so the elements
    // needed are the rank's own plus the halo that touches its interface vertices -- recomputed locally instead of
    // exchanging 1152 bytes per element (DOTTimeStepper.cpp:349-380, :574-616 run on every rank's share).
    h->shardHess = h->shardElems && (h->owner || (h->tune.shardHess >= 0 ? h->tune.shardHess != 0 : true));
    h->nHessElems = nT;
    if (h->shardHess) {
        std::vector<uint8_t> needV(nV, 0);
        for (int pI = h->p0; pI < h->p1; ++pI)
            for (int v : h->partVerts[pI]) needV[v] = 1;
        if (!h->owner)
            for (int v = h->v0; v < h->v1; ++v) needV[v] = 1;
        std::vector<int> el, e2c(nT, -1);
        for (int e = 0; e < nT; ++e)
            if (needV[h->T[4 * e]] || needV[h->T[4 * e + 1]] || needV[h->T[4 * e + 2]] || needV[h->T[4 * e + 3]]) {
                e2c[e] = (int)el.size();
                el.push_back(e);
            }
        std::vector<int> bl, bptr(1, 0), bent, optr(1, 0), oent;
        for (int v = 0; v < nV; ++v) {
            if (!needV[v]) continue;
            for (int k = adj_ptr[v]; k < adj_ptr[v + 1]; ++k) {
                bl.push_back(k);
                for (int i = blk_ptr[k]; i < blk_ptr[k + 1]; ++i) {
                    const int e = blk_ent[i] >> 4;
                    bent.push_back((e2c[e] << 4) | (blk_ent[i] & 15));   // every contributor touches v: listed
                    if (h->owner && h->epart[e] >= h->p0 && h->epart[e] < h->p1) oent.push_back(bent.back());
                }
                bptr.push_back((int)bent.size());
                optr.push_back((int)oent.size());
            }
        }
        if (h->owner) {
            if (oent.empty()) oent.push_back(-1);
            if (int rc = upload(h, &h->ownBlkPtr, optr)) return rc;
            if (int rc = upload(h, &h->ownBlkEnt, oent)) return rc;
        }
        h->nHessElems = (int)el.size();
        h->nHessBlk = (int)bl.size();
        if (int rc = upload(h, &h->hessElems, el)) return rc;
        if (int rc = upload(h, &h->hessBlk, bl)) return rc;
        if (int rc = upload(h, &h->hessBlkPtr, bptr)) return rc;
        if (int rc = upload(h, &h->hessBlkEnt, bent)) return rc;
    }
    // element patches (patches.hpp): PTall covers every element (the kernel-level entry points evaluate the whole mesh on
    // every rank), PT this rank's own elements -- the same object unless the element pass is sharded
    {
        int PE = h->tune.patchElems > 0 ? (h->tune.patchElems <= 256 ? 256 : 512) : 256;
        std::vector<int> all(nT);
        for (int e = 0; e < nT; ++e) all[e] = e;
        if (int rc = upload_patches(h, build_patches(nV, h->T.data(), h->Xrest.data(), all, PE), h->PTall)) return rc;
        if (h->shardElems) {
            std::vector<int> own;
            for (int e = 0; e < nT; ++e)
                if (h->epart[e] >= h->p0 && h->epart[e] < h->p1) own.push_back(e);
            if (int rc = upload_patches(h, build_patches(nV, h->T.data(), h->Xrest.data(), own, PE), h->PT)) return rc;
        } else {
            h->PT = h->PTall;
        }
        // the patches of a speculating step (one rank): the set that keeps direction rows + patches + trial-point workgroups
        // resident at once -- the default set where it does, 512-element patches where only those do, none otherwise
        h->specFits = false;
        if (!h->dist && h->world == 1 && h->tune.specStep != 0) {
            const int nbv = (nV + 255) / 256, room = 512 - NB_RED - nbv;
            if (std::max(h->PT.nPatches, nbv) <= room) {
                h->PTspec = h->PT;
                h->specFits = true;
            } else if (h->tune.patchElems == 0 && PE == 256) {
                const HostPatches H2 = build_patches(nV, h->T.data(), h->Xrest.data(), all, 512);
                if (std::max(H2.nPatches, nbv) <= room) {
                    if (int rc = upload_patches(h, H2, h->PTspec)) return rc;
                    h->specFits = true;
                }
            }
        }
    }
    if (h->owner) {
        // who holds / owns a vertex: a rank HOLDS the vertices of its subdomains (= of its elements); the lowest rank that
        // holds a vertex OWNS it (its inertia term, its share of every dot product).  Vertices held by two or more ranks
        // are the only ones whose entries travel inside the loop.
        std::vector<int> holders(nV, 0), last(nV, -1), ownerR(nV, -1);
        for (int r = 0; r < h->world; ++r)
            for (int pI = h->firstPart[r]; pI < h->firstPart[r + 1]; ++pI)
                for (int v : h->partVerts[pI])
                    if (last[v] != r) {
                        last[v] = r;
                        holders[v]++;
                        if (ownerR[v] < 0) ownerR[v] = r;
                    }
        std::vector<uint8_t> own(nV, 0), held(nV, 0);
        std::vector<int> iface;
        std::vector<double> mo(nV, 0.0);
        for (int pI = h->p0; pI < h->p1; ++pI)
            for (int v : h->partVerts[pI]) held[v] = 1;
        for (int v = 0; v < nV; ++v) {
            own[v] = ownerR[v] == h->rank || (ownerR[v] < 0 && h->rank == 0);
            if (own[v]) mo[v] = h->mass[v];
            if (holders[v] >= 2) iface.push_back(v);
        }
        h->nIface = (int)iface.size();
        if (iface.empty()) iface.push_back(0);
        std::vector<int> hl;
        for (int v = 0; v < nV; ++v)
            if (held[v]) hl.push_back(v);
        h->nHeld = (int)hl.size();
        if (hl.empty()) hl.push_back(0);
        if (int rc = upload(h, &h->heldList, hl)) return rc;
        if (int rc = upload(h, &h->ownMask, own)) return rc;
        if (int rc = upload(h, &h->heldMask, held)) return rc;
        {
            std::vector<uint8_t> kind(nV);
            for (int v = 0; v < nV; ++v) kind[v] = (uint8_t)((own[v] ? 1 : 0) | (holders[v] >= 2 ? 2 : 0));
            if (int rc = upload(h, &h->vkind, kind)) return rc;
            std::vector<int> sh;
            for (int v = 0; v < nV; ++v)
                if (held[v] && holders[v] >= 2) sh.push_back(v);
            h->nShared = (int)sh.size();
            if (sh.empty()) sh.push_back(0);
            if (int rc = upload(h, &h->sharedList, sh)) return rc;
        }
        if (int rc = upload(h, &h->ifaceIdx, iface)) return rc;
        if (int rc = upload(h, &h->massOwn, mo)) return rc;
        if (int rc = dalloc(h, &h->xpack, (size_t)3 * h->nIface + 8 + RED_K)) return rc;
        // the element pass' inertia term 1/2 m |x - x~|^2 by ownership too: the same kernel over every vertex with the
        // owner's share of the mass (positions outside the held vertices stay where the warm start put them)
        h->Mown = h->M;
        h->Mown.mass = h->massOwn;
        if (h->tune.fuseLog)
            fprintf(stderr, "dotmi: owner exchange: rank %d holds %d of %d vertices, %d are held by more than one rank\n", h->rank,
                    (int)std::count(held.begin(), held.end(), 1), nV, h->nIface);
    }
    return 0;
}

LbfgsArgs lbfgs_args(const dotmi_handle *h)
{
    LbfgsArgs L;
    memset(&L, 0, sizeof(L));
    L.m = h->m;
    for (int i = 0; i < h->m; ++i) {
        L.s[i] = h->S[h->order[i]];
        L.y[i] = h->Y[h->order[i]];
        L.ys[i] = h->ys[i];
        for (int j = 0; j < h->m; ++j) L.sy[i][j] = h->sy[i][j];
    }
    return L;
}

int free_slot(const dotmi_handle *h)
{
    for (int s = 0; s <= h->hist; ++s) {
        bool used = false;
        for (int i = 0; i < h->m; ++i) used |= (h->order[i] == s);
        if (!used) return s;
    }
    return 0;
}

}  // namespace dotmi

extern "C" {

// host-only: no device is touched
int dotmi_plan_shards(int32_t nParts, const int32_t *part_scalar_size, int32_t world, int32_t *first_part)
{
    if (nParts < 0 || world < 1 || !first_part || (nParts > 0 && !part_scalar_size)) return DOTMI_E_INVALID;
    std::vector<double> cost(nParts + 1, 0.0);
    for (int p = 0; p < nParts; ++p) cost[p + 1] = cost[p] + (double)part_scalar_size[p] * part_scalar_size[p];
    first_part[0] = 0;
    first_part[world] = nParts;
    for (int r = 1; r < world; ++r) {
        const double target = cost[nParts] * r / world;
        int c = (int)(std::lower_bound(cost.begin(), cost.end(), target) - cost.begin());
        if (c > 0 && target - cost[c - 1] < cost[c] - target) --c;
        c = std::min(std::max(c, first_part[r - 1]), nParts);
        first_part[r] = c;
    }
    return 0;
}

const char *dotmi_last_error(const dotmi_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

// host-only: the element patches of patches.hpp for all elements of a mesh.  First call with elem == NULL for the sizes
// (n_patches, pv, n_slots), then with arrays of nPatches*PE (elem), 4*nPatches*PE (tl, epos: uint16), nPatches*PV (pv_gid,
// pv_slot), nPatches (pv_cnt), nPatches*(PV+1) (c_ptr: uint16) and 2*nV (pp_rng).  tests/test_patches.py checks the
// invariants the element pass relies on and replays the two-stage gradient sum in numpy.
int dotmi_plan_patches(int32_t nV, int32_t nT, const int32_t *T, const double *X, int32_t PE, int32_t *n_patches, int32_t *pv,
                       int32_t *n_slots, int32_t *elem, uint16_t *tl, uint16_t *epos, int32_t *pv_gid, int32_t *pv_slot,
                       int32_t *pv_cnt, uint16_t *c_ptr, int32_t *pp_rng)
{
    if (nV < 1 || nT < 1 || !T || !X || (PE != 256 && PE != 512) || !n_patches || !pv || !n_slots) return DOTMI_E_INVALID;
    std::vector<int> all(nT);
    for (int e = 0; e < nT; ++e) all[e] = e;
    const HostPatches H = build_patches(nV, T, X, all, PE);
    *n_patches = H.nPatches;
    *pv = H.PV;
    *n_slots = H.nSlots;
    if (!elem) return 0;
    std::copy(H.elem.begin(), H.elem.end(), elem);
    std::copy(H.tl.begin(), H.tl.end(), tl);
    std::copy(H.epos.begin(), H.epos.end(), epos);
    std::copy(H.pv_gid.begin(), H.pv_gid.end(), pv_gid);
    std::copy(H.pv_slot.begin(), H.pv_slot.end(), pv_slot);
    std::copy(H.pv_cnt.begin(), H.pv_cnt.end(), pv_cnt);
    std::copy(H.c_ptr.begin(), H.c_ptr.end(), c_ptr);
    std::copy(H.pp_rng.begin(), H.pp_rng.end(), pp_rng);
    return 0;
}

// host-only: the vertex patches of vpatches.hpp (see include/dotmi.h)
int dotmi_plan_vpatches(int32_t nV, int32_t nT, const int32_t *T, const double *X, int32_t PE, int32_t max_own, int32_t *hdr,
                        int32_t *owner, int32_t *elem, int32_t *eown, int32_t *runlen)
{
    if (nV < 1 || nT < 1 || !T || !X || PE < 1 || max_own < 1 || !hdr) return DOTMI_E_INVALID;
    const HostVPatches H = build_vpatches(nV, nT, T, X, PE, max_own);
    hdr[0] = H.nPatches;
    hdr[1] = H.PE;
    hdr[2] = H.PV;
    hdr[3] = H.PO;
    hdr[4] = H.RUN;
    if (!owner || H.nPatches == 0) return 0;
    for (int p = 0; p < H.nPatches; ++p) {
        const uint16_t *cp = &H.c_ptr[(size_t)p * (H.PO + 1)];
        for (int lv = 0; lv < H.po_cnt[p]; ++lv) {
            const int v = H.pv_gid[(size_t)p * H.PV + lv];
            owner[v] = p;
            if (runlen) runlen[v] = cp[lv + 1] - cp[lv];
        }
    }
    if (elem) std::copy(H.elem.begin(), H.elem.end(), elem);
    if (eown)
        for (size_t s = 0; s < H.eown.size(); ++s) eown[s] = H.eown[s];
    return 0;
}

// host-only: the level schedule of tile_factor.hpp for ONE block of nt x nt tiles with the given upper tile pattern, in the
// compact row-block layout (c0[j] = first tile column stored for tile row j, c0[j] <= every pattern entry of column j).
// Offsets are in doubles into one array: the factor storage first, the scratch tiles after it (*scratch_base).
//   tasks: 10 int64 per task  {level, form, init, post, nprod, first product, c offset, q offset (-1), ldc, ldq}
//   prods:  4 int64 per product {a offset, b offset, lda, ldb}
// With tasks == NULL only the counts are returned.  The tests execute the schedule in numpy, level after level, and
// compare with a dense inverse Cholesky factor (tests/test_tile_schedule.py).
static int plan_tile_schedule_impl(int32_t nt, const uint8_t *live, const uint8_t *pattern, const int32_t *c0, int32_t eager_min,
                                   int32_t eager_chunk, int64_t *tasks, int64_t *prods, int64_t *n_tasks, int64_t *n_prods,
                                   int64_t *n_levels, int64_t *storage, int64_t *scratch_base, int64_t *row_off, int32_t *row_ld,
                                   const uint8_t *leaf_tile, const int32_t *c0m, const int32_t *ntm, int64_t *row_off_m,
                                   int32_t *row_ld_m)
{
    if (nt < 1 || !live || !pattern || !c0 || !n_tasks || !n_prods) return DOTMI_E_INVALID;
    std::vector<long long> rtOff(nt, -1);
    std::vector<int> rtLd(nt, 0), rtC0(nt, 0);
    long long tot = 0;
    for (int j = 0; j < nt; ++j) {
        if (!live[j]) continue;
        rtC0[j] = 64 * c0[j];
        rtLd[j] = 64 * (j + 1) - rtC0[j];
        rtOff[j] = tot;
        tot += 64ll * rtLd[j];
    }
    // two-level form: a second range per separator row block, tile columns [c0m[j], c0m[j] + ntm[j])
    std::vector<long long> rtOffM(nt, -1);
    std::vector<int> rtLdM(nt, 0), rtC0M(nt, 0);
    if (leaf_tile)
        for (int j = 0; j < nt; ++j) {
            if (!live[j] || leaf_tile[j] || !c0m || !ntm || ntm[j] <= 0) continue;
            rtC0M[j] = 64 * c0m[j];
            rtLdM[j] = 64 * ntm[j] + 16;
            rtOffM[j] = tot;
            tot += 64ll * rtLdM[j];
        }
    double *const W = reinterpret_cast<double *>(1ull << 40);   // never dereferenced: only offsets leave this function
    double *const scratch = W + tot;
    std::vector<uint8_t> lv(live, live + nt), pat(pattern, pattern + (size_t)nt * nt);
    std::vector<TileTaskL> all;
    TileSchedule S;
    size_t sn = 0;
    plan_subdomain_tiles(0, nt, W, rtOff.data(), rtLd.data(), rtC0.data(), lv, pat, scratch, sn, all, S.clearTiles, S.clearLd,
                         S.flops, S.qTiles, std::max(1, eager_min), std::max(1, eager_chunk), 0, true, -1,
                         leaf_tile ? rtOffM.data() : nullptr, leaf_tile ? rtLdM.data() : nullptr, leaf_tile ? rtC0M.data() : nullptr,
                         leaf_tile);
    std::vector<int> levelOf;
    {
        // finish_tile_schedule reorders inside levels; keep the level of every task
        std::stable_sort(all.begin(), all.end(), [](const TileTaskL &a, const TileTaskL &b) { return a.level < b.level; });
        for (auto &t : all) levelOf.push_back(t.level);
    }
    size_t np = 0;
    for (auto &t : all) np += t.prods.size();
    *n_tasks = (int64_t)all.size();
    *n_prods = (int64_t)np;
    if (n_levels) *n_levels = all.empty() ? 0 : all.back().level;
    if (storage) *storage = tot;
    if (scratch_base) *scratch_base = tot;
    if (row_off)
        for (int j = 0; j < nt; ++j) row_off[j] = rtOff[j];
    if (row_ld)
        for (int j = 0; j < nt; ++j) row_ld[j] = rtLd[j];
    if (row_off_m)
        for (int j = 0; j < nt; ++j) row_off_m[j] = rtOffM[j];
    if (row_ld_m)
        for (int j = 0; j < nt; ++j) row_ld_m[j] = rtLdM[j];
    if (!tasks || !prods) return 0;
    size_t pi = 0;
    for (size_t k = 0; k < all.size(); ++k) {
        const TileTask &t = all[k].t;
        int64_t *o = tasks + 11 * k;
        o[10] = t.o - W;
        o[0] = all[k].level;
        o[1] = t.form;
        o[2] = t.init;
        o[3] = t.post;
        o[4] = (int64_t)all[k].prods.size();
        o[5] = (int64_t)pi;
        o[6] = t.c - W;
        o[7] = t.q ? t.q - W : -1;
        o[8] = t.ldc;
        o[9] = t.ldq;
        for (auto &pr : all[k].prods) {
            int64_t *q = prods + 4 * pi++;
            q[0] = pr.a - W;
            q[1] = pr.b - W;
            q[2] = pr.lda;
            q[3] = pr.ldb;
        }
    }
    return 0;
}

int dotmi_plan_tile_schedule(int32_t nt, const uint8_t *live, const uint8_t *pattern, const int32_t *c0, int32_t eager_min,
                             int32_t eager_chunk, int64_t *tasks, int64_t *prods, int64_t *n_tasks, int64_t *n_prods,
                             int64_t *n_levels, int64_t *storage, int64_t *scratch_base, int64_t *row_off, int32_t *row_ld)
{
    return plan_tile_schedule_impl(nt, live, pattern, c0, eager_min, eager_chunk, tasks, prods, n_tasks, n_prods, n_levels, storage,
                                   scratch_base, row_off, row_ld, nullptr, nullptr, nullptr, nullptr, nullptr);
}
// host-only: the same schedule in the TWO-LEVEL form (tile_factor.hpp, twoLevel; a leaves-first block): leaf_tile[t] = tile row t
// belongs to a leaf; a separator's row block j stores the leaf tile columns [c0m[j], c0m[j] + ntm[j]) of its sub-tree in a second
// range (row_off_m / row_ld_m) and its main range starts at its sub-tree's first separator column c0[j]
int dotmi_plan_tile_schedule_two_level(int32_t nt, const uint8_t *live, const uint8_t *pattern, const int32_t *c0,
                                       const uint8_t *leaf_tile, const int32_t *c0m, const int32_t *ntm, int32_t eager_min,
                                       int32_t eager_chunk, int64_t *tasks, int64_t *prods, int64_t *n_tasks, int64_t *n_prods,
                                       int64_t *n_levels, int64_t *storage, int64_t *row_off, int32_t *row_ld, int64_t *row_off_m,
                                       int32_t *row_ld_m)
{
    if (!leaf_tile || !c0m || !ntm) return DOTMI_E_INVALID;
    return plan_tile_schedule_impl(nt, live, pattern, c0, eager_min, eager_chunk, tasks, prods, n_tasks, n_prods, n_levels, storage,
                                   nullptr, row_off, row_ld, leaf_tile, c0m, ntm, row_off_m, row_ld_m);
}

// host-only: the dependencies the dataflow kernel (tile_flow_kernel) waits on, for the task list dotmi_plan_tile_schedule
// returns (same arguments, same task order): task v may run once the tasks dep_idx[dep_ptr[v] .. dep_ptr[v+1]) have finished.
// With dep_idx == NULL only *n_deps is returned.  tests/test_tile_schedule.py executes the tasks in random orders that
// respect exactly these edges.
int dotmi_plan_tile_deps(int32_t nt, const uint8_t *live, const uint8_t *pattern, const int32_t *c0, int32_t eager_min,
                         int32_t eager_chunk, int64_t *dep_ptr, int64_t *dep_idx, int64_t *n_deps)
{
    if (nt < 1 || !live || !pattern || !c0 || !n_deps) return DOTMI_E_INVALID;
    std::vector<long long> rtOff(nt, -1);
    std::vector<int> rtLd(nt, 0), rtC0(nt, 0);
    long long tot = 0;
    for (int j = 0; j < nt; ++j) {
        if (!live[j]) continue;
        rtC0[j] = 64 * c0[j];
        rtLd[j] = 64 * (j + 1) - rtC0[j];
        rtOff[j] = tot;
        tot += 64ll * rtLd[j];
    }
    double *const W = reinterpret_cast<double *>(1ull << 40);
    double *const scratch = W + tot;
    std::vector<uint8_t> lv(live, live + nt), pat(pattern, pattern + (size_t)nt * nt);
    std::vector<TileTaskL> all;
    TileSchedule S;
    size_t sn = 0;
    plan_subdomain_tiles(0, nt, W, rtOff.data(), rtLd.data(), rtC0.data(), lv, pat, scratch, sn, all, S.clearTiles, S.clearLd,
                         S.flops, S.qTiles, std::max(1, eager_min), std::max(1, eager_chunk));
    std::stable_sort(all.begin(), all.end(), [](const TileTaskL &a, const TileTaskL &b) { return a.level < b.level; });
    for (auto &t : all) {
        TileTask k = t.t;
        k.first = (int)S.prods.size();
        k.nprod = (int)t.prods.size();
        for (auto &p : t.prods) S.prods.push_back(p);
        S.tasks.push_back(k);
    }
    std::vector<int> depPtr, depIdx;
    build_tile_deps(S.tasks, S.prods, depPtr, depIdx);
    *n_deps = (int64_t)depIdx.size();
    if (!dep_ptr || !dep_idx) return 0;
    for (size_t k = 0; k < depPtr.size(); ++k) dep_ptr[k] = depPtr[k];
    for (size_t k = 0; k < depIdx.size(); ++k) dep_idx[k] = depIdx[k];
    return 0;
}

// host-only: the nested-dissection layout build_device_mesh() would use for parts [p0,p1)
int dotmi_plan_layout(int32_t nV, int32_t nT, const int32_t *T, const double *Xrest, const int32_t *epart,
                      int32_t nParts, int32_t p0, int32_t p1, int32_t levels, int32_t min_split, int32_t node_cap,
                      int32_t *nodes, int32_t *n_nodes, int32_t *nmax, int32_t *pos)
{
    if (nV < 1 || nT < 1 || !T || !Xrest || !epart || nParts < 1 || p0 < 0 || p1 > nParts || p0 > p1 || !n_nodes ||
        !nmax)
        return DOTMI_E_INVALID;
    for (int e = 0; e < nT; ++e) {
        if (epart[e] < 0 || epart[e] >= nParts) return DOTMI_E_INVALID;
        for (int k = 0; k < 4; ++k)
            if (T[4 * e + k] < 0 || T[4 * e + k] >= nV) return DOTMI_E_INVALID;
    }
    std::vector<int> adj_ptr, adj_idx;
    build_adjacency(nV, nT, T, adj_ptr, adj_idx);
    std::vector<std::vector<int>> sets(p1 - p0);
    {
        std::vector<int> mark(nV, -1);
        for (int e = 0; e < nT; ++e) {
            const int pI = epart[e];
            if (pI < p0 || pI >= p1) continue;
            for (int k = 0; k < 4; ++k) sets[pI - p0].push_back(T[4 * e + k]);
        }
        for (auto &v : sets) {
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
        }
    }
    std::vector<NdNode> tree;
    std::vector<std::vector<std::vector<int>>> region;
    // dotmi_create's rule: depth (and, with it, the split threshold) from ALL subdomains of the mesh
    int levelsUse = levels, minUse = min_split < 128 ? ND_MIN_SPLIT : min_split;
    if (levels < 0) {
        std::vector<std::vector<int>> allSets(nParts);
        for (int e = 0; e < nT; ++e)
            for (int k = 0; k < 4; ++k) allSets[epart[e]].push_back(T[4 * e + k]);
        for (auto &v : allSets) {
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
        }
        if (min_split < 128) nd_choose_depth(allSets, nV, adj_ptr, adj_idx, Xrest, BS_NARROW, ND_MIN_SPLIT, levelsUse, minUse);
        else levelsUse = nd_default_levels(allSets);
    }
    *nmax = nd_plan(sets, nV, adj_ptr, adj_idx, Xrest, levelsUse, minUse, tree, region);
    // (the leaves-first layout of the two-level back-solve when the environment forces that form: DOTMI_TWO_LEVEL=1; the automatic
    // choice of dotmi_create -- where the one-pass form would stream 240 MB or more, on at least four levels -- is the caller's to mirror)
    if (getenv("DOTMI_TWO_LEVEL") && atoi(getenv("DOTMI_TWO_LEVEL")) > 0) nd_relayout_leaves_first(tree);
    *n_nodes = (int32_t)tree.size();
    if (nodes) {
        if ((int)tree.size() > node_cap) return DOTMI_E_INVALID;
        for (size_t i = 0; i < tree.size(); ++i) {
            const NdNode &N = tree[i];
            const int32_t row[6] = {N.off, N.size, N.a, N.c, N.offS, N.sizeS};
            std::copy(row, row + 6, nodes + 6 * i);
        }
    }
    if (pos) {
        size_t base = 0;
        for (size_t ls = 0; ls < sets.size(); ++ls) {
            std::unordered_map<int, int> posOf;
            for (size_t nd = 0; nd < tree.size(); ++nd) {
                const auto &rv = region[nd][ls];
                const int ro = nd_region_first_row(tree[nd], 3 * (int)rv.size());
                for (size_t k = 0; k < rv.size(); ++k) posOf[rv[k]] = ro + 3 * (int)k;
            }
            for (size_t i = 0; i < sets[ls].size(); ++i) pos[base + i] = posOf.at(sets[ls][i]);
            base += sets[ls].size();
        }
    }
    return 0;
}

// host-only: the job table of the back-solve launches dotmi_create builds for parts [p0, p1) of this mesh (bs_tiles.hpp) -- the
// product's own planners (nd_choose_depth / nd_plan, plan_backsolve_tiles) on the caller's mesh, no device touched.
// tiles: up to `cap` rows of 6 int32 {part (local), first row, rows, first column, tile index in the part, job}: job = index of
// the launch's workgroup that runs the tile -- 0 .. n_wide-1 in the 512-thread launch, then 0 .. n_narrow-1 one-tile jobs of the
// 256-thread launch, then n_narrow + k for the four tiles of pack k; tiles of the two-phase kernel (rows beyond 5120 columns)
// have job = -1.  counts: {tiles, n_wide, n_narrow, n_packs, n_long, nmax, shallow launch (0 / 1), few-subdomains rule (0 / 1)}.
int dotmi_plan_backsolve_tiles(int32_t nV, int32_t nT, const int32_t *T, const double *Xrest, const int32_t *epart, int32_t nParts,
                               int32_t p0, int32_t p1, int32_t cap, int32_t *tiles, int32_t *counts)
{
    if (nV < 1 || nT < 1 || !T || !Xrest || !epart || nParts < 1 || p0 < 0 || p1 > nParts || p0 > p1 || !counts) return DOTMI_E_INVALID;
    for (int e = 0; e < nT; ++e) {
        if (epart[e] < 0 || epart[e] >= nParts) return DOTMI_E_INVALID;
        for (int k = 0; k < 4; ++k)
            if (T[4 * e + k] < 0 || T[4 * e + k] >= nV) return DOTMI_E_INVALID;
    }
    std::vector<int> adj_ptr, adj_idx;
    build_adjacency(nV, nT, T, adj_ptr, adj_idx);
    std::vector<std::vector<int>> allSets(nParts);
    for (int e = 0; e < nT; ++e)
        for (int k = 0; k < 4; ++k) allSets[epart[e]].push_back(T[4 * e + k]);
    for (auto &v : allSets) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
    }
    const Tuning tune = Tuning::from_env();
    int levels = tune.ndLevels, minSplit = tune.ndMin;
    if (levels < 0 && !getenv("DOTMI_ND_MIN")) nd_choose_depth(allSets, nV, adj_ptr, adj_idx, Xrest, BS_NARROW, minSplit, levels, minSplit);
    else if (levels < 0) levels = nd_default_levels(allSets);
    std::vector<std::vector<int>> sets(allSets.begin() + p0, allSets.begin() + p1);
    std::vector<NdNode> tree;
    std::vector<std::vector<std::vector<int>>> region;
    const int nmax = nd_plan(sets, nV, adj_ptr, adj_idx, Xrest, levels, minSplit, tree, region);
    BsTilePlan TP;
    plan_backsolve_tiles(tree, [&](int k, int ls) { return 3 * (int)region[k][ls].size(); }, p1 - p0, nmax,
                         BsTileRules{tune.tileRows, tune.tileRowsLong, tune.tilePasses, tune.wavePacks}, TP);
    const int nN = TP.ntiles - TP.ntilesWide;
    int n = 0;
    auto put = [&](const int4 &t, int job) {
        if (tiles && n < cap) {
            int32_t *o = tiles + 6 * (size_t)n;
            o[0] = t.x; o[1] = t.y; o[2] = t.z >> 16; o[3] = t.w; o[4] = t.z & 0xffff; o[5] = job;
        }
        ++n;
    };
    for (int j = 0; j < TP.ntiles; ++j) put(TP.tiles[j], j < TP.ntilesWide ? j : j - TP.ntilesWide);
    for (int k = 0; k < 4 * TP.nquad; ++k)
        if ((TP.tiles[TP.ntiles + k].z >> 16) > 0) put(TP.tiles[TP.ntiles + k], nN + k / 4);
    for (const int4 &t : TP.ltiles) put(t, -1);
    counts[0] = n; counts[1] = TP.ntilesWide; counts[2] = nN; counts[3] = TP.nquad; counts[4] = (int)TP.ltiles.size();
    counts[5] = nmax; counts[6] = TP.shallow ? 1 : 0; counts[7] = TP.fewTiles ? 1 : 0;
    return (tiles && n > cap) ? DOTMI_E_INVALID : 0;
}

// host-only: what one rank owns under dotmi_create's plan
int dotmi_plan_rank(int32_t nV, int32_t nT, const int32_t *T, const int32_t *epart, int32_t nParts, int32_t rank,
                    int32_t world, int32_t *p0, int32_t *p1, int32_t *elems, int32_t *n_elems, int32_t *v0, int32_t *v1,
                    int32_t *part_size)
{
    if (nV < 1 || nT < 1 || !T || !epart || nParts < 1 || world < 1 || rank < 0 || rank >= world) return DOTMI_E_INVALID;
    std::vector<int32_t> ps(nParts, 0), first(world + 1);
    {
        std::vector<int> mark(nV, -1);
        for (int pI = 0; pI < nParts; ++pI)
            for (int e = 0; e < nT; ++e)
                if (epart[e] == pI)
                    for (int k = 0; k < 4; ++k) {
                        const int v = T[4 * e + k];
                        if (v < 0 || v >= nV) return DOTMI_E_INVALID;
                        if (mark[v] != pI) {
                            mark[v] = pI;
                            ps[pI] += 3;
                        }
                    }
    }
    dotmi_plan_shards(nParts, ps.data(), world, first.data());
    if (p0) *p0 = first[rank];
    if (p1) *p1 = first[rank + 1];
    int ne = 0;
    for (int e = 0; e < nT; ++e)
        if (epart[e] >= first[rank] && epart[e] < first[rank + 1]) {
            if (elems) elems[ne] = e;
            ++ne;
        }
    if (n_elems) *n_elems = ne;
    if (v0) *v0 = (int32_t)((long long)nV * rank / world);
    if (v1) *v1 = (int32_t)((long long)nV * (rank + 1) / world);
    if (part_size) std::copy(ps.begin(), ps.end(), part_size);
    return 0;
}

// host-only: the built-in element partitioner (partition.hpp)
int dotmi_partition(int32_t nV, int32_t nT, const int32_t *T, const double *X, int32_t nParts, int32_t *epart)
{
    if (nV < 1 || nT < 1 || !T || !X || nParts < 1 || !epart) return DOTMI_E_INVALID;
    for (int i = 0; i < 4 * nT; ++i)
        if (T[i] < 0 || T[i] >= nV) return DOTMI_E_INVALID;
    partition_elements(nV, nT, T, X, nParts, epart);
    return 0;
}

int dotmi_comm_unique_id(void *out128)
{
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return DOTMI_E_DEVICE;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    memcpy(out128, &id, 128);
    return 0;
}

int32_t dotmi_factor_kind(const dotmi_handle *h)
{
    if (!h) return DOTMI_E_INVALID;
    return h->tileFlow ? 2 : h->tileSplit ? 3 : 1;
}

// host-only: the form of the block solve dotmi_create chooses for this mesh and partition when DOTMI_TWO_LEVEL is unset (0 = explicit
// inverse in one pass, 1 = two-level), and the bytes per application of the one-pass form on its own layout, counted over all
// subdomains, that decide it (>= 240 MB and a tree with separators: two-level)
int dotmi_plan_backsolve_form(int32_t nV, int32_t nT, const int32_t *T, const double *Xrest, const int32_t *epart, int32_t nParts,
                              int32_t *form, int64_t *one_pass_bytes)
{
    if (nV < 1 || nT < 1 || !T || !Xrest || !epart || nParts < 1 || !form) return DOTMI_E_INVALID;
    for (int e = 0; e < nT; ++e) {
        if (epart[e] < 0 || epart[e] >= nParts) return DOTMI_E_INVALID;
        for (int k = 0; k < 4; ++k)
            if (T[4 * e + k] < 0 || T[4 * e + k] >= nV) return DOTMI_E_INVALID;
    }
    std::vector<int> adj_ptr, adj_idx;
    build_adjacency(nV, nT, T, adj_ptr, adj_idx);
    std::vector<std::vector<int>> allSets(nParts);
    for (int e = 0; e < nT; ++e)
        for (int k = 0; k < 4; ++k) allSets[epart[e]].push_back(T[4 * e + k]);
    for (auto &v : allSets) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
    }
    int levels = -1, minSplit = ND_MIN_SPLIT;
    nd_choose_depth(allSets, nV, adj_ptr, adj_idx, Xrest, BS_NARROW, ND_MIN_SPLIT, levels, minSplit);
    std::vector<NdNode> tree;
    std::vector<std::vector<std::vector<int>>> region;
    nd_plan(allSets, nV, adj_ptr, adj_idx, Xrest, levels, minSplit, tree, region);
    const long long bytes = tree.empty() ? 0 : 8 * one_pass_nnz(tree, region, nParts);
    *form = (!tree.empty() && tree[0].a >= 0 && bytes >= TWO_LEVEL_FROM_BYTES) ? 1 : 0;
    if (one_pass_bytes) *one_pass_bytes = bytes;
    return 0;
}

int32_t dotmi_backsolve_form(const dotmi_handle *h)
{
    if (!h) return DOTMI_E_INVALID;
    return h->twoLevel ? 1 : 0;
}

int32_t dotmi_comm_ranks(const dotmi_handle *h)
{
    if (!h) return DOTMI_E_INVALID;
    if (h->comm) {
        int n = 0;
        if (ncclCommCount(h->comm, &n) != ncclSuccess) return DOTMI_E_DEVICE;
        return n;
    }
    return h->arCb ? -h->world : 1;
}

void dotmi_destroy(dotmi_handle *h)
{
    if (!h) return;
    hipSetDevice(h->device);
    if (h->st) hipStreamSynchronize(h->st);
    if (h->comm) ncclCommDestroy(h->comm);
    for (void *p : h->allocs) hipFree(p);
    if (h->arStage) hipHostFree(h->arStage);
    if (h->h_partE) hipHostFree(h->h_partE);
    if (h->h_partR) hipHostFree(h->h_partR);
    if (h->h_alpha) hipHostFree(h->h_alpha);
    if (h->h_ctl) hipHostFree(h->h_ctl);
    if (h->dposPinned) hipHostFree(h->dposPinned);
    if (h->evDir) hipEventDestroy(h->evDir);
    if (h->h_info) hipHostFree(h->h_info);
    if (h->h_flags) hipHostFree(h->h_flags);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->ev2) hipEventDestroy(h->ev2);
    if (h->evA) hipEventDestroy(h->evA);
    for (hipEvent_t e : h->evP)
        if (e) hipEventDestroy(e);
    for (hipEvent_t e : h->tFork) hipEventDestroy(e);
    for (hipEvent_t e : h->tJoin) hipEventDestroy(e);
    if (h->stDiag) hipStreamDestroy(h->stDiag);
    for (hipEvent_t e : h->evPre) hipEventDestroy(e);
    for (hipEvent_t e : h->evAr) hipEventDestroy(e);
    if (h->factorGraph) hipGraphExecDestroy(h->factorGraph);
    if (h->st) hipStreamDestroy(h->st);
    delete h;
}

static int create_impl(dotmi_handle *h, const dotmi_mesh *mesh, const dotmi_params *prm, const double *x_init)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        h->err = "no HIP device available: libdotmi has no CPU fallback";
        return DOTMI_E_NOGPU;
    }
    if (!mesh || !prm || !x_init || mesh->nV <= 0 || mesh->nT <= 0 || !mesh->X_rest || !mesh->T || !mesh->mu ||
        !mesh->lambda || !mesh->fixed || mesh->nParts < 1 || prm->dt <= 0 ||
        prm->history < 1 || prm->history > HIST_MAX || prm->world < 1 || prm->rank < 0 ||
        prm->rank >= prm->world || (prm->world > 1 && !prm->comm_id && !prm->allreduce) ||
        (prm->energy != DOTMI_ENERGY_FCR && prm->energy != DOTMI_ENERGY_SNH)) {
        h->err = "invalid argument";
        return DOTMI_E_INVALID;
    }
    for (int e = 0; e < mesh->nT; ++e) {
        if (mesh->epart && !mesh->vpart && (mesh->epart[e] < 0 || mesh->epart[e] >= mesh->nParts)) {
            h->err = "epart out of range";
            return DOTMI_E_INVALID;
        }
        for (int k = 0; k < 4; ++k)
            if (mesh->T[4 * e + k] < 0 || mesh->T[4 * e + k] >= mesh->nV) {
                h->err = "tet index out of range";
                return DOTMI_E_INVALID;
            }
    }
    h->nV = mesh->nV;
    h->nT = mesh->nT;
    h->n = 3 * mesh->nV;
    h->mat = prm->energy;
    h->hist = prm->history;
    h->iterCap = prm->iterCap > 0 ? prm->iterCap : 10000;
    h->dt = prm->dt;
    h->dtSq = prm->dt * prm->dt;
    for (int d = 0; d < 3; ++d) {
        h->grav[d] = prm->gravity[d];
        h->gdtsq[d] = h->dtSq * prm->gravity[d];
    }
    h->relTol = prm->relTol;
    h->alphaMin = prm->alphaMin;
    h->device = prm->device;
    h->rank = prm->rank;
    h->world = prm->world;
    h->flags = prm->flags;
    h->tune = Tuning::from_env();
#ifdef DOTMI_TEST_HOOKS
    h->testIterDelta = Tuning::geti("DOTMI_TEST_ITER_DELTA", 0);
    h->testFailRefresh = Tuning::geti("DOTMI_TEST_FAIL_REFRESH", 0);
#endif
    h->density = mesh->density;
    h->nPartsAll = mesh->nParts;
    h->T.assign(mesh->T, mesh->T + 4 * (size_t)h->nT);
    if (mesh->vpart) {
        if (prm->world > 1 || (prm->flags & DOTMI_FLAG_FORCE_DIST)) {
            h->err = "a vertex partition (vpart) is single-GPU only";
            return DOTMI_E_INVALID;
        }
        for (int v = 0; v < mesh->nV; ++v)
            if (mesh->vpart[v] < 0 || mesh->vpart[v] >= mesh->nParts) {
                h->err = "vpart out of range";
                return DOTMI_E_INVALID;
            }
        h->vpart.assign(mesh->vpart, mesh->vpart + h->nV);
        h->epart.assign(h->nT, 0);   // unused: the subdomains are vertex sets
    } else if (mesh->epart) {
        h->epart.assign(mesh->epart, mesh->epart + h->nT);
    } else {   // no partition given: the built-in partitioner (the reference calls METIS here, METIS.hpp:109-140)
        h->epart.resize(h->nT);
        partition_elements(mesh->nV, mesh->nT, mesh->T, mesh->X_rest, mesh->nParts, h->epart.data());
    }
    h->fixed.assign(mesh->fixed, mesh->fixed + h->nV);
    h->Xrest.assign(mesh->X_rest, mesh->X_rest + h->n);
    h->mu.assign(mesh->mu, mesh->mu + h->nT);
    h->lam.assign(mesh->lambda, mesh->lambda + h->nT);

    HIPCHECK(h, hipSetDevice(h->device));
    HIPCHECK(h, hipStreamCreate(&h->st));
    HIPCHECK(h, hipEventCreate(&h->ev0));
    HIPCHECK(h, hipEventCreate(&h->ev1));
    HIPCHECK(h, hipEventCreate(&h->ev2));
    HIPCHECK(h, hipEventCreate(&h->evA));
    h->timePhases = (h->flags & DOTMI_FLAG_TIME_PHASES) != 0;
    if (h->timePhases)
        for (auto &e : h->evP) HIPCHECK(h, hipEventCreate(&e));
    h->dist = h->world > 1 || (h->flags & DOTMI_FLAG_FORCE_DIST);
    h->shardElems = h->dist && (h->tune.shardElems >= 0 ? h->tune.shardElems != 0 : h->nT >= 400000);
    h->owner = h->dist && (h->flags & DOTMI_FLAG_OWNER_EXCHANGE);
    if (h->owner) {
        if (h->flags & (DOTMI_FLAG_HOST_LOOP | DOTMI_FLAG_TIME_PHASES | DOTMI_FLAG_GSDD | DOTMI_FLAG_NEWTON)) {
            h->err = "DOTMI_FLAG_OWNER_EXCHANGE: device loop only";
            return DOTMI_E_INVALID;
        }
        h->shardElems = true;   // own elements, own rows, own share of the refresh
    }
    h->arCb = prm->allreduce;
    h->arCtx = prm->allreduce_ctx;
    if (h->dist && !h->arCb) {
        ncclUniqueId id;
        if (h->world > 1) memcpy(&id, prm->comm_id, 128);
        else NCCLCHECK(h, ncclGetUniqueId(&id));
        NCCLCHECK(h, ncclCommInitRank(&h->comm, h->world, id, h->rank));
    }
    {
        void *cd = nullptr;
        HIPCHECK(h, hipMalloc(&cd, sizeof(double) * (RED_K + 8)));
        h->allocs.push_back(cd);
        h->ctrlDev = (double *)cd;
    }

    h->timeStride = h->tune.timeStride;
    if (h->flags & DOTMI_FLAG_TIME_BACKSOLVE) {
        h->evPre.resize(2 * 512);
        for (auto &e : h->evPre) HIPCHECK(h, hipEventCreate(&e));
        if (h->dist) {
            h->evAr.resize(2 * 512);
            for (auto &e : h->evAr) HIPCHECK(h, hipEventCreate(&e));
        }
    }
    host_features(h);
    h->targetGRes = host_target_gres(h);
    if (int rc = build_device_mesh(h)) return rc;
    const int n = h->n;
    double **vecs[] = {&h->x, &h->x_trial, &h->xn, &h->v, &h->xt, &h->g, &h->g_trial, &h->p, &h->q, &h->z,
                       &h->Hp, &h->tmpn};
    for (double **pp : vecs) {
        if (int rc = dalloc(h, pp, (size_t)n + 8)) return rc;
        HIPCHECK(h, hipMemsetAsync(*pp, 0, sizeof(double) * (n + 8), h->st));
    }
    for (int s = 0; s <= h->hist; ++s) {
        if (int rc = dalloc(h, &h->S[s], (size_t)n)) return rc;
        if (int rc = dalloc(h, &h->Y[s], (size_t)n)) return rc;
    }
    if (int rc = dalloc(h, &h->He, (size_t)144 * std::max(h->nHessElems, 1))) return rc;
    if (int rc = dalloc(h, &h->Hval, hval_size(h->M.nnzb))) return rc;
    if (h->owner) {
        if (int rc = dalloc(h, &h->HvalOwn, hval_size(h->M.nnzb))) return rc;
        HIPCHECK(h, hipMemsetAsync(h->HvalOwn, 0, sizeof(double) * hval_size(h->M.nnzb), h->st));
    }
    if (int rc = dalloc(h, &h->partE, (size_t)4 * ELEM_NB_MAX)) return rc;   // (second half: a paired trial's full-step partials)
    double **parts[] = {&h->partR, &h->partC, &h->partS, &h->partG, &h->partGR, &h->partGC, &h->partCT, &h->partST};
    for (double **pp : parts) {
        // (partR: a step on vertex patches leaves one row of statistics per PATCH, up to 512 of them -- k_elemvert.hip)
        const size_t rows = pp == &h->partR ? 512 : NB_RED;
        if (int rc = dalloc(h, pp, rows * RED_K)) return rc;
        HIPCHECK(h, hipMemsetAsync(*pp, 0, sizeof(double) * rows * RED_K, h->st));
    }
    if (int rc = dalloc(h, &h->alpha_dev, 8)) return rc;   // (dalloc counts doubles: [0] the trial's step, [1] a paired trial's full step)
    if (int rc = dalloc(h, &h->gstage, (size_t)n + 2)) return rc;
    HIPCHECK(h, hipMemsetAsync(h->gstage, 0, sizeof(double) * ((size_t)n + 2), h->st));
    HIPCHECK(h, hipHostMalloc((void **)&h->h_partE, sizeof(double) * 2 * ELEM_NB_MAX));
    HIPCHECK(h, hipHostMalloc((void **)&h->h_partR, sizeof(double) * NB_RED * RED_K));
    HIPCHECK(h, hipHostMalloc((void **)&h->h_alpha, sizeof(double) * 8));
    {
        h->gsdd = (h->flags & DOTMI_FLAG_GSDD) != 0;
        h->newton = (h->flags & DOTMI_FLAG_NEWTON) != 0;
        if (h->newton && (h->dist || h->gsdd)) {
            h->err = "DOTMI_FLAG_NEWTON: single GPU, not together with DOTMI_FLAG_GSDD";
            return DOTMI_E_INVALID;
        }
        if (h->gsdd && h->dist) {
            h->err = "DOTMI_FLAG_GSDD: single GPU only";
            return DOTMI_E_INVALID;
        }
        h->devLoop = !h->gsdd && !h->newton && !(h->flags & (DOTMI_FLAG_HOST_LOOP | DOTMI_FLAG_TIME_PHASES));
        // replicated element pass, merged tile partials: the back-solve of the next direction is issued on the trial
        // gradient, beside the controller (enqueue_loop_slot); sharded subdomains keep their one collective per iteration
        // (round 4: also with the sharded element pass -- the scatter of -g and H s_new then happen in pair_stats, behind the
        // gradient's all-reduce; DOTMI_EARLY_SHARDED=0 keeps the q-based order there)
        h->earlyBs = h->devLoop && h->tune.earlyBs != 0 &&
                     (h->P.mt_ptr != nullptr || h->P.splitMerge);
        if (h->owner && !(h->earlyBs && h->tune.fuseDir)) {
            h->err = "DOTMI_FLAG_OWNER_EXCHANGE needs the device loop's early order with the fused direction kernel";
            return DOTMI_E_INVALID;
        }
        // the trials' grouping of the energy partials, everywhere: the start-of-step evaluation and the fused-step trials must
        // sum E in the same grouping or an `E > E_cur` verdict can flip at rounding level (the owner exchange runs the fused
        // step on its sharded element pass too)
        if (h->earlyBs && h->tune.fuseStep && (!h->shardElems || h->owner)) h->PT.wgCap = h->PTspec.wgCap = 512;
        // vertex patches: one rank, the early order with both fused kernels (their statements are what k_elemvert.hip keeps), a
        // 256-thread back-solve launch to host the controller, every patch a workgroup of its own, its LDS within 64 KB
        h->vpFits = false;
        // (Stable Neo-Hookean only: its element work is ~1 us of a workgroup's ~12, so carrying every element 1.9 times is free; with
        // the fixed-corotational SVD the doubled element work costs what the saved launch gives -- bunny5K 1.656 -> 1.696 ms, measured)
        if (h->earlyBs && !h->dist && h->world == 1 && h->tune.fuseStep && h->tune.fuseDir && h->tune.vertexPatches != 0 &&
            (h->mat == DOTMI_ENERGY_SNH || h->tune.vertexPatches > 0) &&
            h->P.ntiles - h->P.ntilesWide + h->P.nquad > 0 && !(h->flags & (DOTMI_FLAG_GSDD | DOTMI_FLAG_NEWTON | DOTMI_FLAG_HOST_LOOP))) {
            const HostVPatches HV = build_vpatches(h->nV, h->nT, h->T.data(), h->Xrest.data(), 512, 85);
            const size_t shm = sizeof(double) * ((size_t)3 * HV.PV + (size_t)3 * HV.RUN) + 2 * (size_t)((HV.PO + 1 + 3) & ~3);
            if (HV.nPatches > 0 && HV.nPatches <= 512 && HV.nPatches <= ELEM_NB_MAX && shm <= 64 * 1024 && HV.PO + 1 <= 256) {
                if (int rc = upload_vpatches(h, HV, h->VP)) return rc;
                h->vpFits = true;
                if (h->tune.fuseLog)
                    fprintf(stderr, "dotmi: vertex patches: %d patches, %.2f x the elements, up to %d owned / %d touched vertices, runs %d, %zu B LDS\n",
                            HV.nPatches, (double)h->VP.nSlotsUsed / std::max(1, h->nT), HV.PO, HV.PV, HV.RUN, shm);
            }
        }
        if (h->dist) {
            if (int rc = dalloc(h, &h->zstage, (size_t)h->n)) return rc;
            HIPCHECK(h, hipMemsetAsync(h->zstage, 0, sizeof(double) * h->n, h->st));   // (owner exchange: stays zero off the held set)
        }
        if (h->earlyBs) {
            if (int rc = dalloc(h, &h->u_old, (size_t)h->n)) return rc;
            for (int sl = 0; sl <= h->hist; ++sl) {
                if (int rc = dalloc(h, &h->MY[sl], (size_t)h->n)) return rc;
                if (int rc = dalloc(h, &h->HS[sl], (size_t)h->n)) return rc;
            }
        }
        h->logCap = std::min(h->iterCap, 10001) + 1;
        h->kindCap = 4096;
        HIPCHECK(h, hipHostMalloc((void **)&h->h_ctl, sizeof(DevLoop)));
        HIPCHECK(h, hipHostMalloc((void **)&h->h_flags, sizeof(int) * 4));
        HIPCHECK(h, hipMalloc((void **)&h->ctl, sizeof(DevLoop)));
        h->allocs.push_back(h->ctl);
        if (int rc = dalloc(h, &h->dlog, (size_t)3 * h->logCap)) return rc;
        HIPCHECK(h, hipMalloc((void **)&h->dkind, sizeof(int) * h->kindCap));
        h->allocs.push_back(h->dkind);
    }

    // Optimizer.cpp:124-184: result = data0 (+script init), v = 0, x_n = x, x~
    HIPCHECK(h, hipMemcpyAsync(h->x, x_init, sizeof(double) * n, hipMemcpyHostToDevice, h->st));
    HIPCHECK(h, hipMemcpyAsync(h->xn, h->x, sizeof(double) * n, hipMemcpyDeviceToDevice, h->st));
    launch_be_update(h->nV, h->M.fixed, h->x, h->xn, h->v, h->xt, h->dt, h->gdtsq, h->st);
    HIPCHECK(h, hipStreamSynchronize(h->st));
    // DOTTimeStepper::precompute (DOTTimeStepper.cpp:150-178)
    return refactor(h, h->x, nullptr, nullptr);
}

int dotmi_create(const dotmi_mesh *mesh, const dotmi_params *prm, const double *x_init, dotmi_handle **out)
{
    if (!out) return DOTMI_E_INVALID;
    *out = nullptr;
    dotmi_handle *h = new dotmi_handle();
    int rc = create_impl(h, mesh, prm, x_init);
    if (rc != 0) {
        g_create_error = h->err;
        dotmi_destroy(h);
        return rc;
    }
    *out = h;
    return 0;
}

}  // extern "C"
