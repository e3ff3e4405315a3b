// patches.hpp -- element patches of the element pass (host only, plain C++).
//
// The reference sums element gradients into vertices through Mesh::vFLoc (Energy.cpp:543-563).  Done literally on a
// GPU that is 96 bytes per tet written to HBM and read back by the vertex pass.  Here the elements are grouped into
// PATCHES of up to PE elements that are close together in space; one workgroup stages the positions of a patch's
// vertices in LDS, drops the 12 gradient entries of its elements into per-vertex runs in LDS and sums the runs there,
// so only ONE 24-byte partial per (patch, vertex) goes through HBM -- about two per vertex instead of the ~22 incident
// (element, slot) contributions.  A vertex's partials sit next to each other in `gpart` (ascending patch), which is
// what the vertex pass reads.  Every order is fixed at build time: results are bit-identical run to run.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace dotmi {

struct HostPatches {
    int nPatches = 0, PE = 0, PV = 0, nSlots = 0;
    std::vector<int> elem;              // nPatches*PE: global element id of every slot, -1 = padding slot
    std::vector<uint16_t> tl;           // nPatches*PE*4: patch-local vertex index of the slot's four corners (0xFFFF pad)
    std::vector<int> pv_gid, pv_slot;   // nPatches*PV: global vertex id (-1 pad) / slot in gpart of a patch vertex
    std::vector<int> pv_cnt;            // nPatches
    std::vector<uint16_t> c_ptr;        // per patch: (PV+1) offsets of the vertices' corner runs
    std::vector<uint16_t> epos;         // nPatches*PE*4: position of a slot's corner in its vertex's run (runs ascending in element id)
    std::vector<int> pp_rng;            // 2*nV: a vertex's partials are gpart[3*pp_rng[2v] .. 3*pp_rng[2v+1]); vertices are laid
                                        // out in the order of their first patch, so a patch's stores are mostly close together
};

inline uint64_t morton3(uint32_t x, uint32_t y, uint32_t z)
{
    auto spread = [](uint64_t v) {   // 21 bits -> every third bit
        v &= 0x1fffff;
        v = (v | v << 32) & 0x1f00000000ffffull;
        v = (v | v << 16) & 0x1f0000ff0000ffull;
        v = (v | v << 8) & 0x100f00f00f00f00full;
        v = (v | v << 4) & 0x10c30c30c30c30c3ull;
        v = (v | v << 2) & 0x1249249249249249ull;
        return v;
    };
    return spread(x) | spread(y) << 1 | spread(z) << 2;
}

// elems: the elements this pass covers (all, or a rank's own), T: nT*4, X: rest positions nV*3
inline HostPatches build_patches(int nV, const int32_t *T, const double *X, const std::vector<int> &elems, int PE)
{
    HostPatches H;
    H.PE = PE;
    const size_t ne = elems.size();
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v = 0; v < nV; ++v)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], X[3 * v + d]);
            hi[d] = std::max(hi[d], X[3 * v + d]);
        }
    double ext = 0;
    for (int d = 0; d < 3; ++d) ext = std::max(ext, hi[d] - lo[d]);
    if (!(ext > 0)) ext = 1.0;
    std::vector<std::pair<uint64_t, int>> key(ne);
    for (size_t i = 0; i < ne; ++i) {
        const int e = elems[i];
        uint32_t q[3];
        for (int d = 0; d < 3; ++d) {
            double c = 0;
            for (int k = 0; k < 4; ++k) c += X[3 * T[4 * e + k] + d];
            const double u = (0.25 * c - lo[d]) / ext;
            q[d] = (uint32_t)std::min(2097151.0, std::max(0.0, u * 2097151.0));
        }
        key[i] = {morton3(q[0], q[1], q[2]), e};
    }
    std::sort(key.begin(), key.end());
    H.nPatches = (int)((ne + PE - 1) / PE);
    H.elem.assign((size_t)H.nPatches * PE, -1);
    H.pv_cnt.assign(H.nPatches, 0);
    std::vector<std::vector<int>> pverts(H.nPatches);
    int pvmax = 0;
    for (int p = 0; p < H.nPatches; ++p) {
        const size_t b = (size_t)p * PE, e = std::min(ne, b + PE);
        std::vector<int> el;
        for (size_t i = b; i < e; ++i) el.push_back(key[i].second);
        std::sort(el.begin(), el.end());   // slots ascending in element id: the CSR below is then in vFLoc order
        auto &pv = pverts[p];
        for (size_t i = 0; i < el.size(); ++i) {
            H.elem[b + i] = el[i];
            for (int k = 0; k < 4; ++k) pv.push_back(T[4 * el[i] + k]);
        }
        std::sort(pv.begin(), pv.end());
        pv.erase(std::unique(pv.begin(), pv.end()), pv.end());
        H.pv_cnt[p] = (int)pv.size();
        pvmax = std::max(pvmax, (int)pv.size());
    }
    H.PV = std::max(8, (pvmax + 7) / 8 * 8);
    H.pv_gid.assign((size_t)H.nPatches * H.PV, -1);
    H.pv_slot.assign((size_t)H.nPatches * H.PV, 0);
    H.tl.assign((size_t)H.nPatches * PE * 4, 0xFFFF);
    H.c_ptr.assign((size_t)H.nPatches * (H.PV + 1), 0);
    H.epos.assign((size_t)H.nPatches * PE * 4, 0);
    H.pp_rng.assign((size_t)2 * nV, 0);
    std::vector<int> kv(nV, 0), cur(nV, -1);
    for (int p = 0; p < H.nPatches; ++p)
        for (int v : pverts[p]) kv[v]++;
    H.nSlots = 0;
    for (int p = 0; p < H.nPatches; ++p)
        for (int v : pverts[p])
            if (cur[v] < 0) {   // first patch that touches v: its run of kv[v] slots starts here
                cur[v] = H.nSlots;
                H.pp_rng[2 * v] = H.nSlots;
                H.nSlots += kv[v];
                H.pp_rng[2 * v + 1] = H.nSlots;
            }
    std::vector<int> cnt;
    for (int p = 0; p < H.nPatches; ++p) {
        const auto &pv = pverts[p];
        const size_t vb = (size_t)p * H.PV, eb = (size_t)p * PE;
        for (size_t lv = 0; lv < pv.size(); ++lv) {
            H.pv_gid[vb + lv] = pv[lv];
            H.pv_slot[vb + lv] = cur[pv[lv]]++;   // patches ascending: a vertex's partials are in patch order
        }
        cnt.assign(pv.size() + 1, 0);
        for (int i = 0; i < PE; ++i) {
            const int e = H.elem[eb + i];
            if (e < 0) continue;
            for (int k = 0; k < 4; ++k) {
                const int lv = (int)(std::lower_bound(pv.begin(), pv.end(), T[4 * e + k]) - pv.begin());
                H.tl[(eb + i) * 4 + k] = (uint16_t)lv;
                cnt[lv + 1]++;
            }
        }
        for (size_t lv = 0; lv < pv.size(); ++lv) cnt[lv + 1] += cnt[lv];
        uint16_t *cp = &H.c_ptr[(size_t)p * (H.PV + 1)];
        for (size_t lv = 0; lv <= pv.size(); ++lv) cp[lv] = (uint16_t)cnt[lv];
        for (int lv = (int)pv.size() + 1; lv <= H.PV; ++lv) cp[lv] = cp[pv.size()];
        std::vector<int> at(cnt.begin(), cnt.end() - 1);
        for (int i = 0; i < PE; ++i) {   // slots ascending == element ids ascending; corners ascending inside
            if (H.elem[eb + i] < 0) continue;
            for (int k = 0; k < 4; ++k) H.epos[(eb + i) * 4 + k] = (uint16_t)at[H.tl[(eb + i) * 4 + k]]++;
        }
    }
    return H;
}

}  // namespace dotmi
