// bs_tiles.hpp -- the job table of the subdomain back-solve launch (host only): how the rows of every tree region of every owned
// subdomain are cut into tiles, which tiles go to which launch / kernel form, and in which order the hardware starts them.
// Pure function of the dissection tree and the number of live rows per (region, subdomain); dotmi_create uses it, and the
// host-only entry dotmi_plan_backsolve_tiles returns it for tests (tests/test_host_logic.py) and tools.
//
// Role in the reference: none of its own -- CHOLMODSolver::solve (CHOLMODSolver.cpp:149-163) walks CHOLMOD's supernodes; here
// the solve is p_s = X_s^T (X_s r_s) streamed once, and this table is its schedule (k_backsolve.hip, backsolve_*).
#pragma once
#include <algorithm>
#include <vector>

#include "dotmi_internal.hpp"

namespace dotmi {

struct BsTileRules {
    int tileRows = 0;        // DOTMI_TILE_ROWS       (0: 64, or 32 for few subdomains)
    int tileRowsLong = 0;    // DOTMI_TILE_ROWS_LONG  (0: automatic)
    int tilePasses = 4;      // DOTMI_TILE_PASSES
    bool wavePacks = true;   // DOTMI_WAVE_PACKS
};

struct BsTilePlan {
    // job table of the launches: [tiles of more than BS_NARROW columns (512-thread launch) | one-tile jobs of the 256-thread
    // launch, heavy first | packs: 4 entries each, rows = 0 padding]; entry = (part, first row, tile index in the part | rows << 16,
    // first column)
    std::vector<int4> tiles;
    int ntiles = 0, ntilesWide = 0, nquad = 0, maxTileLen = 0;
    // rows beyond BS_LONG columns: two-phase kernel, (tile, column chunk) work items
    std::vector<int4> ltiles;
    std::vector<int2> lwork;
    int maxChunks = 1;
    // the same tiles grouped by part (GSDD solves one subdomain at a time)
    std::vector<int4> tilesByPart, ltilesByPart;
    std::vector<int2> lworkByPart;
    std::vector<int> partTilePtr, partLworkPtr;
    // per part: columns [first, end) each of its tiles contributes to, in tile-index order
    std::vector<std::vector<int2>> ranges;
    bool fewTiles = false, shallow = false;
};

inline int bs_tile_len(const int4 &t) { return t.y + (t.z >> 16) - t.w; }

// usedRows(nd, ls) = live scalar rows of tree node nd (its leaf block / separator) in owned subdomain ls
template <class UsedRows>
inline void plan_backsolve_tiles(const std::vector<NdNode> &nd, UsedRows usedRows, int nParts, int nmax, const BsTileRules &R,
                                 BsTilePlan &out)
{
    out = BsTilePlan();
    // rows per back-solve tile: 64, or 32 when 64-row tiles would not give every CU two workgroups (few subdomains:
    // the launch is then bound by the pass chain of a workgroup, which halves)
    const bool fewTiles = (long long)nParts * nmax / 64 < 2 * 256;
    int tileRows = fewTiles ? 32 : 64;
    if (R.tileRows > 0) tileRows = R.tileRows;
    out.fewTiles = fewTiles;
    auto region_geom = [&](size_t k, int ls, int &ro, int &used, int &cb) {
        const NdNode &N = nd[k];
        used = usedRows((int)k, ls);
        ro = nd_region_first_row(N, used);   // first live row of the region
        // the rows of a region start at their node's first column (a leaf's padding sits in front of its
        // live rows and is skipped; 16-column granularity keeps the 128-byte lines whole)
        cb = N.a < 0 ? (ro & ~15) : N.off;
    };
    // Round 5: is the back-solve launch SHALLOW -- its workgroups (one-tile jobs + packs of four small tiles, at 64 rows per tile)
    // resident at once, or nearly (<= 1.5 x the 512 slots of two 256-thread workgroups per CU)?  Then the launch lasts as long as
    // its longest tile, and the long rows' tiles are cut to DOTMI_TILE_PASSES passes (below); in a deep launch (horse7K@r1:64:
    // 1646 workgroups, 1 M tets: 8640) the queue sets the length and more, smaller tiles cost (+8 % per iteration on the horse).
    bool shallowLaunch = false;
    if (R.wavePacks && !fewTiles) {
        long long big = 0, small = 0;
        for (int ls = 0; ls < nParts; ++ls)
            for (size_t k = 0; k < nd.size(); ++k) {
                int ro, used, cb;
                region_geom(k, ls, ro, used, cb);
                for (int r0 = ro, rows = 0; r0 < ro + used; r0 += rows) {
                    rows = std::min(std::min(tileRows, ro + used - r0), 64 - (r0 & 63));
                    (r0 + rows - cb <= BS_WAVE ? small : big)++;
                }
            }
        shallowLaunch = big + (small + 3) / 4 <= 768;
    }
    out.shallow = shallowLaunch;
    std::vector<int4> tiles;
    out.ranges.assign(nParts, {});
    for (int ls = 0; ls < nParts; ++ls) {
        int b = 0;
        for (size_t k = 0; k < nd.size(); ++k) {
            int ro, used, cb;
            region_geom(k, ls, ro, used, cb);
            // a tile stays inside one 64-row block of the factor storage (RowTile): the first tile of a region ends
            // at the next multiple of 64
            // rows of more than 1536 columns (the separators of the upper tree levels) can take fewer rows per tile
            // (DOTMI_TILE_ROWS_LONG).  With few subdomains the launch lasts as long as its longest tile
            // (bunny5K / 8: a 32-row tile of the root separator is 512 KB at ~30 GB/s per workgroup), so those rows get
            // tiles of ~256 KB: 16 rows at 2000 columns, 8 at 3000 (round 4: bunny5K 23.0 -> 16.8 us, horse7K 46.5 -> 31.8)
            // Round 5: a tile is a CHAIN of passes (rows in registers -> dot products -> butterfly -> exchange -> update), ~5.5 us
            // each, and the launch lasts at least as long as its longest chain.  Rows of more than 1024 columns go 8 to a pass,
            // so their 64-row tiles were 8 passes = 42-50 us -- the whole launch on bar17K (tools/prof_backsolve.sh: the 110
            // root tiles start at t = 0 and end last, whatever the other slots do).  In a shallow launch (above) the rows beyond
            // 1536 columns are cut to DOTMI_TILE_PASSES = 4 passes (32 rows).  That pays only together with the packs of small
            // tiles: alone either change leaves the launch at 50 us (the shorter root tiles queue behind ~800 small workgroups
            // for the slots), together 50.0 -> 42.1; cutting the rows beyond 1024 columns too puts the queue back (47.9)
            // (profiles/r05_backsolve_tiles.txt).
            const int len = ro + used - cb;
            int trows = tileRows;
            if (len > 1536 && R.tileRowsLong > 0) trows = std::min(tileRows, R.tileRowsLong);
            else if (len > 1536 && fewTiles) trows = std::min(tileRows, std::max(8, (32768 / len) / 8 * 8));
            else if (len > 1536 && shallowLaunch) trows = std::min(tileRows, 8 * R.tilePasses);
            for (int r0 = ro, rows = 0; r0 < ro + used; r0 += rows) {
                rows = std::min(std::min(trows, ro + used - r0), 64 - (r0 & 63));
                tiles.push_back(make_int4(ls, r0, b | (rows << 16), cb));
                out.ranges[ls].push_back(make_int2(cb, r0 + rows));
                ++b;
            }
        }
    }
    // the same tiles grouped by part (GSDD solves one subdomain at a time): register-kernel tiles, and the long-row tiles
    // with their (tile, column chunk) work items
    out.partTilePtr.assign(nParts + 1, 0);
    out.partLworkPtr.assign(nParts + 1, 0);
    for (const int4 &t : tiles) {   // generated part after part
        if (bs_tile_len(t) > BS_LONG) {
            const int nch = (((bs_tile_len(t) + 15) & ~15) + BS_LONG - 1) / BS_LONG;
            for (int c = 0; c < nch; ++c) out.lworkByPart.push_back(make_int2((int)out.ltilesByPart.size(), c));
            out.ltilesByPart.push_back(t);
            out.partLworkPtr[t.x + 1] += nch;
        } else {
            out.tilesByPart.push_back(t);
            out.partTilePtr[t.x + 1]++;
        }
    }
    for (int ls = 0; ls < nParts; ++ls) {
        out.partTilePtr[ls + 1] += out.partTilePtr[ls];
        out.partLworkPtr[ls + 1] += out.partLworkPtr[ls];
    }
    // heavy tiles first: work ~ rows * row length
    auto tile_work = [](const int4 &t) { return (long long)(t.z >> 16) * (t.y + 64 - t.w); };
    std::stable_sort(tiles.begin(), tiles.end(), [&](const int4 &a, const int4 &b) { return tile_work(a) > tile_work(b); });
    // rows longer than the register tile of the single-pass kernel go through the two-phase kernel, cut into
    // column chunks of BS_LONG
    {
        std::vector<int4> shortTiles;
        for (const int4 &t : tiles) {
            const int len = bs_tile_len(t);
            if (len > BS_LONG) {
                const int nch = (((len + 15) & ~15) + BS_LONG - 1) / BS_LONG;
                for (int c = 0; c < nch; ++c) out.lwork.push_back(make_int2((int)out.ltiles.size(), c));
                out.maxChunks = std::max(out.maxChunks, nch);
                out.ltiles.push_back(t);
            } else {
                out.maxTileLen = std::max(out.maxTileLen, len);
                shortTiles.push_back(t);
            }
        }
        tiles.swap(shortTiles);
    }
    // tiles whose rows need the 512-thread variant (more than BS_NARROW columns) first: when both kinds exist they are
    // launched separately, so that the short ones run on the 256-thread kernel (two workgroups per CU instead of one)
    std::stable_partition(tiles.begin(), tiles.end(), [](const int4 &t) { return bs_tile_len(t) > BS_NARROW; });
    for (const int4 &t : tiles) out.ntilesWide += (bs_tile_len(t) > BS_NARROW);
    // small tiles (rows of at most BS_WAVE columns) leave the one-tile jobs: four of them share a workgroup, one wavefront each
    // (k_backsolve.hip, backsolve_wave_tile); heavy first, so the four of a pack are about equally long.  They go BEHIND the
    // one-tile jobs (spread evenly among them: no gain, 1 M tets +3 %: profiles/r05_backsolve_tiles.txt H)
    if (R.wavePacks) {
        std::vector<int4> big, small;
        for (const int4 &t : tiles) (bs_tile_len(t) <= BS_WAVE ? small : big).push_back(t);
        if (small.size() >= 8) {
            while (small.size() % 4) small.push_back(make_int4(0, 0, 0, 0));   // rows = 0: the wavefront leaves at once
            out.nquad = (int)small.size() / 4;
            tiles = big;
            out.ntiles = (int)tiles.size();
            tiles.insert(tiles.end(), small.begin(), small.end());
        }
    }
    if (out.nquad == 0) out.ntiles = (int)tiles.size();
    out.tiles.swap(tiles);
}

}  // namespace dotmi
