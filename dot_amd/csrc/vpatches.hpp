// vpatches.hpp -- vertex patches: the element pass and the vertex gather in ONE launch (host only, plain C++; round 6).
//
// The element patches of patches.hpp partition the ELEMENTS; a vertex then collects one partial per patch that touches it, which is
// a launch of its own (vertex_gather_kernel) behind a grid-wide join.  On meshes small enough that every kernel of the loop is one
// latency chain per workgroup (<= 512 patches: bar17K, the monkey, the bunny) that join costs more than the work behind it: ~3 us of
// launch, ~2.3 us until the loop state and the first operands are there, ~1.2 us until the last store has drained -- per kernel,
// whatever it does (tools/prof_loopkern.sh).  Vertex patches partition the VERTICES instead: a patch owns up to 85 vertices that are
// close together (Morton order of the rest positions) and carries EVERY element incident to one of them -- its own and a halo that
// neighbouring patches carry as well (each element ~1.9 times on the reference's meshes).  The workgroup then has every contribution
// to its vertices' gradient on chip: it sums them in ascending element order (the reference's vFLoc order, Energy.cpp:543-563, now
// over ALL incident elements -- no per-patch partial sums), adds the inertia term (Optimizer.cpp:1239-1252), forms the new L-BFGS
// pair with its statistics (DOTTimeStepper.cpp:474-494) and writes -g into the padded right-hand sides: what the gather did.  An
// element's ENERGY is counted by the patch that owns its first corner.  Every order is fixed at build time.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

#include "patches.hpp"

namespace dotmi {

struct HostVPatches {
    int nPatches = 0, PE = 0, PV = 0, PO = 0, RUN = 0;   // slots per patch / most touched vertices / most owned vertices / longest sum of runs
    std::vector<int> elem;              // nPatches*PE: global element id of every slot (ascending inside a patch), -1 = padding slot
    std::vector<uint16_t> tl;           // nPatches*PE*4: patch-local index of the slot's corners in the touched list (0xFFFF: padding slot)
    std::vector<uint16_t> epos;         // nPatches*PE*4: position of the corner in its vertex's run; 0xFFFF: the vertex is not owned here
    std::vector<uint8_t> eown;          // nPatches*PE: this patch counts the element's energy
    std::vector<int> pv_gid;            // nPatches*PV: touched vertices, the OWNED ones first (local index < po_cnt), -1 pad
    std::vector<int> pv_cnt, po_cnt;    // nPatches
    std::vector<uint16_t> c_ptr;        // per patch (PO+1): offsets of the owned vertices' runs
};

inline HostVPatches build_vpatches(int nV, int nT, const int32_t *T, const double *X, int PE, int maxOwn)
{
    HostVPatches H;
    H.PE = PE;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v = 0; v < nV; ++v)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], X[3 * v + d]);
            hi[d] = std::max(hi[d], X[3 * v + d]);
        }
    double ext = 0;
    for (int d = 0; d < 3; ++d) ext = std::max(ext, hi[d] - lo[d]);
    if (!(ext > 0)) ext = 1.0;
    std::vector<std::pair<uint64_t, int>> key(nV);
    for (int v = 0; v < nV; ++v) {
        uint32_t q[3];
        for (int d = 0; d < 3; ++d) q[d] = (uint32_t)std::min(2097151.0, std::max(0.0, (X[3 * v + d] - lo[d]) / ext * 2097151.0));
        key[v] = {morton3(q[0], q[1], q[2]), v};
    }
    std::sort(key.begin(), key.end());
    // vertex -> incident elements (CSR, ascending element id)
    std::vector<int> ip(nV + 1, 0), ie((size_t)4 * nT);
    for (int e = 0; e < nT; ++e)
        for (int k = 0; k < 4; ++k) ip[T[4 * e + k] + 1]++;
    for (int v = 0; v < nV; ++v) ip[v + 1] += ip[v];
    {
        std::vector<int> at(ip.begin(), ip.end() - 1);
        for (int e = 0; e < nT; ++e)
            for (int k = 0; k < 4; ++k) ie[at[T[4 * e + k]]++] = e;
    }
    // greedy: vertices in Morton order into the running patch while its element set stays within PE and it owns at most maxOwn
    std::vector<int> owner(nV, -1), mark(nT, -1);
    std::vector<std::vector<int>> own, els;
    {
        std::vector<int> curOwn, curEl;
        auto flush = [&]() {
            if (curOwn.empty()) return;
            own.push_back(curOwn);
            els.push_back(curEl);
            curOwn.clear();
            curEl.clear();
        };
        for (int i = 0; i < nV; ++i) {
            const int v = key[i].second, p = (int)own.size();
            int add = 0;
            for (int q = ip[v]; q < ip[v + 1]; ++q) add += mark[ie[q]] != p;
            if (!curOwn.empty() && ((int)curEl.size() + add > PE || (int)curOwn.size() >= maxOwn)) flush();
            const int pp = (int)own.size();
            for (int q = ip[v]; q < ip[v + 1]; ++q)
                if (mark[ie[q]] != pp) {
                    mark[ie[q]] = pp;
                    curEl.push_back(ie[q]);
                }
            curOwn.push_back(v);
            owner[v] = pp;
        }
        flush();
    }
    H.nPatches = (int)own.size();
    int pvmax = 0, pomax = 0, runmax = 0, emax = 0;
    std::vector<std::vector<int>> pverts(H.nPatches);
    for (int p = 0; p < H.nPatches; ++p) {
        std::sort(els[p].begin(), els[p].end());
        std::sort(own[p].begin(), own[p].end());
        std::vector<int> others;
        for (int e : els[p])
            for (int k = 0; k < 4; ++k)
                if (owner[T[4 * e + k]] != p) others.push_back(T[4 * e + k]);
        std::sort(others.begin(), others.end());
        others.erase(std::unique(others.begin(), others.end()), others.end());
        pverts[p] = own[p];
        pverts[p].insert(pverts[p].end(), others.begin(), others.end());
        pvmax = std::max(pvmax, (int)pverts[p].size());
        pomax = std::max(pomax, (int)own[p].size());
        emax = std::max(emax, (int)els[p].size());
        int run = 0;
        for (int v : own[p]) run += ip[v + 1] - ip[v];
        runmax = std::max(runmax, run);
    }
    if (emax > PE) {   // (one vertex with more than PE incident elements: no such patch -- the caller keeps the element patches)
        H.nPatches = 0;
        return H;
    }
    H.PV = std::max(8, (pvmax + 7) / 8 * 8);
    H.PO = pomax;
    H.RUN = (runmax + 7) / 8 * 8;
    H.elem.assign((size_t)H.nPatches * PE, -1);
    H.tl.assign((size_t)H.nPatches * PE * 4, 0xFFFF);
    H.epos.assign((size_t)H.nPatches * PE * 4, 0xFFFF);
    H.eown.assign((size_t)H.nPatches * PE, 0);
    H.pv_gid.assign((size_t)H.nPatches * H.PV, -1);
    H.pv_cnt.assign(H.nPatches, 0);
    H.po_cnt.assign(H.nPatches, 0);
    H.c_ptr.assign((size_t)H.nPatches * (H.PO + 1), 0);
    std::vector<int> loc(nV, -1);
    for (int p = 0; p < H.nPatches; ++p) {
        const auto &pv = pverts[p];
        const int no = (int)own[p].size();
        H.pv_cnt[p] = (int)pv.size();
        H.po_cnt[p] = no;
        for (size_t lv = 0; lv < pv.size(); ++lv) {
            H.pv_gid[(size_t)p * H.PV + lv] = pv[lv];
            loc[pv[lv]] = (int)lv;
        }
        uint16_t *cp = &H.c_ptr[(size_t)p * (H.PO + 1)];
        int off = 0;
        for (int lv = 0; lv < no; ++lv) {
            cp[lv] = (uint16_t)off;
            off += ip[pv[lv] + 1] - ip[pv[lv]];
        }
        for (int lv = no; lv <= H.PO; ++lv) cp[lv] = (uint16_t)off;
        std::vector<int> at(no);
        for (int lv = 0; lv < no; ++lv) at[lv] = cp[lv];
        const size_t eb = (size_t)p * PE;
        for (size_t i = 0; i < els[p].size(); ++i) {   // slots ascending == element ids ascending; corners ascending inside
            const int e = els[p][i];
            H.elem[eb + i] = e;
            H.eown[eb + i] = owner[T[4 * e]] == p ? 1 : 0;
            for (int k = 0; k < 4; ++k) {
                const int lv = loc[T[4 * e + k]];
                H.tl[(eb + i) * 4 + k] = (uint16_t)lv;
                if (lv < no) H.epos[(eb + i) * 4 + k] = (uint16_t)at[lv]++;
            }
        }
        for (int v : pv) loc[v] = -1;
    }
    return H;
}

}  // namespace dotmi
