// dotmi_internal.hpp -- device-side data layout and kernel launch prototypes of libdotmi.
#pragma once
#include <hip/hip_runtime.h>

#include "nd_layout.hpp"
#include "tile_factor.hpp"
#include <stdint.h>

namespace dotmi {

constexpr int HIST_MAX = 6;     // L-BFGS pairs kept (reference uses 5, DOTTimeStepper.cpp:45)
// workgroups of the element pass when the mesh has more patches than that: what is resident at once with the prefetching
// instantiation's registers (Stable Neo-Hookean 158: three per CU; fixed-corotational with its SVD 190: two) x 256 CUs
inline int elem_wg_cap(int mat) { return mat == 1 ? 768 : 512; }
constexpr int NB_RED = 256;     // blocks of every reducing kernel (fixed => run-to-run bit-identical sums)
constexpr int RED_K = 3 * HIST_MAX + 3;  // partial values per block
constexpr int BS_WAVE = 256;    // rows of at most that many columns: four tiles per workgroup, one wavefront each
constexpr int BS_NARROW = 3072; // longest row the 256-thread form of that kernel takes (two workgroups per CU; round 5: 2560 -> 3072 with the
                                // right-hand side of the sixth chunk's rows in LDS); longer rows go to the 512-thread form in a launch of their own
constexpr int BS_LONG = 5120;   // longest row (columns) the single-pass back-solve kernel holds in registers (512 threads x 5 x 2)
constexpr int ELEM_NB_MAX = 8192; // most workgroups (= partial energy rows) of the element pass; beyond, workgroups loop over patches
constexpr int CHOL_NB = 64;     // tile of the inverse-Cholesky factorisation (LDS resident)

// ---- mesh + topology resident in HBM ------------------------------------------------------------
// Layout of the global Hessian's 3 x 3 blocks (round 6): blocks in groups of eight, a group's 72 doubles entry-major --
// entry e of block b at (b >> 3) * 72 + 8 e + (b & 7).  The eight lanes of a row's group in the SpMV kernels read the same entry
// of eight consecutive blocks with one instruction: 64 contiguous bytes instead of eight words 72 bytes apart.  With block-major
// storage (9 b + e, until round 5) every one of the 27 block loads of a trip touched ~80 lines per wave and the direction kernel
// spent ~7 of its ~12 us in the address path of the vector memory unit (tools/prof_dirstep.sh), not in waiting.
__host__ __device__ inline size_t hval_idx(int blk, int e) { return (size_t)(blk >> 3) * 72 + (size_t)(8 * e + (blk & 7)); }
inline size_t hval_size(int nnzb) { return (size_t)72 * (((size_t)nnzb + 7) / 8); }

struct DevMesh {
    int nV, nT, nTp;        // nTp = nT padded to 64 (SoA stride)
    int4 *T;                // nT   vertex ids
    double *A;              // [9][nTp] rest-shape inverse, SoA (coalesced per-lane loads)
    double *mu, *lam, *vol; // nT
    double *mass;           // nV
    uint8_t *fixed;         // nV
    // block-CSR of the global Hessian: vertex adjacency incl. self, ascending
    int *adj_ptr, *adj_idx;
    int nnzb;
    // per block: contributing (elem*16 + a*4 + b), ascending elem  -> deterministic gather assembly
    int *blk_ptr, *blk_ent;
    int *blk_row;           // nnzb: row vertex of each block
};

// ---- element patches of the element pass (patches.hpp) ---------------------------------------------
struct DevPatches {
    int nPatches = 0, PE = 0, PV = 0, nSlots = 0, nElem = 0;
    int wgCap = 0;                // > 0: every instantiation of the element pass uses at most this many workgroups on this handle
                                  // (one grouping of the energy partials for the start-of-step evaluation and the trials)
    ushort4 *tl = nullptr;        // nPatches*PE: patch-local vertex indices of a slot's corners (x == 0xFFFF: padding slot)
    double *A = nullptr;          // [9][nPatches*PE] rest-shape inverse in patch order, SoA
    double *mu = nullptr, *lam = nullptr, *vol = nullptr;   // nPatches*PE in patch order (0 on padding)
    double mu0 = 0.0, lam0 = 0.0;   // mu == nullptr: every element has these Lame parameters (one material: 16 B per tet not read)
    int *pv_gid = nullptr, *pv_slot = nullptr, *pv_cnt = nullptr;   // nPatches*PV, nPatches*PV, nPatches
    unsigned short *c_ptr = nullptr;   // per patch (PV+1): offsets of the vertices' corner runs
    ushort4 *epos = nullptr;          // nPatches*PE: position of a slot's four corners in their vertices' runs
    int2 *pp_rng = nullptr;       // nV: vertex v owns gpart[3*pp_rng[v].x .. 3*pp_rng[v].y)
    double *gpart = nullptr;      // 3*nSlots per-(patch, vertex) partial gradients, vertex-major
};

// ---- vertex patches (vpatches.hpp): element pass + vertex gather in one launch (k_elemvert.hip) -------------------------------------
struct DevVPatches {
    int nPatches = 0, PE = 0, PV = 0, PO = 0, RUN = 0, nSlotsUsed = 0;
    ushort4 *tl = nullptr, *epos = nullptr;   // nPatches*PE: corners' patch-local vertex indices / positions in their vertices' runs (0xFFFF: not owned)
    double *A = nullptr;                      // [9][nPatches*PE] rest-shape inverse in patch order, SoA
    double *vol = nullptr, *volE = nullptr;   // nPatches*PE: volume (0 on padding) / the same where this patch counts the element's energy, else 0
    double *mu = nullptr, *lam = nullptr;     // nPatches*PE, or nullptr: every element has mu0 / lam0
    double mu0 = 0.0, lam0 = 0.0;
    int *pv_gid = nullptr, *pv_cnt = nullptr, *po_cnt = nullptr;   // touched vertices (owned first) / their number / the owned ones
    unsigned short *c_ptr = nullptr;          // per patch (PO+1): offsets of the owned vertices' runs
};
struct ElemVertArgs {
    const double *mass, *xt, *p, *hp;
    const double *spmv_partials;   // p.g / p.Hp partials, COLUMN-major [2][NB_RED] (launch_spmv_zp's partialsT)
    const uint8_t *fixed;
    const int *vp_ptr, *vp_off;   // the copies of a vertex in the padded right-hand sides (DevParts)
    double *rpad;
    double *partE, *partR, *alpha_out;
    double dtSq, alpha_min;
    const double *x0;             // ctl == nullptr (start of a step): evaluate here, no step, no pair ...
    double *g0;                   // ... the gradient goes here
};

// ---- factor storage -------------------------------------------------------------------------------
// A subdomain's X_s is kept by 64-row blocks (and H_s, then R, in a work buffer of the same layout: tile_factor.hpp).  Memory
// row i of block J = i / 64, column c (c0 <= c < 64 (J + 1)) is at W[off + (i - 64 J) * ld + (c - c0)].  A block only holds
// what its rows can have non-zero -- from the first column of the rows' tree node to the end of the block's diagonal tile
// (ld = 64 (J + 1) - c0) --, blocks of pure padding rows hold nothing (off = -1): the storage is ~1.2x the structural
// non-zeros the back-solve streams instead of nmax^2 per subdomain.
struct RowTile {
    long long off;
    int ld, c0;
};

// ---- two-level form of the back-solve (round 6; leaves-first layout, nd_layout.hpp) -----------------------------------------
// The factor buffer holds X_DD = chol(H_DD)^-1 of every leaf D, X_GG = the inverse of the separator complement's own factor, and in
// the separator rows' leaf columns M_GD = L_GD X_DD -- non-zero only in the rows of separator vertices NEXT TO leaf D.  One
// application:  c = M_GD r_D (panel rows, twolevel_forward_kernel) -> t_G = r_G - sum c (twolevel_rhs_kernel) -> the one-pass tile
// kernels on the leaves' and the separators' own rows (q_D = X_DD^T X_DD r_D, p_G = X_GG^T X_GG t_G) -> p_D = q_D - M_GD^T p_G
// (twolevel_backward_kernel on the per-subdomain sums psub).  The panels are read twice, everything else once.
struct DevTwoLevel {
    int on;
    int nPanels;                 // (owned subdomain, leaf) pairs with at least one coupled separator row
    int maxRows, maxCols;        // over the panels
    int nItems;                  // twolevel_forward_kernel's work items: (panel, first of eight rows)
    const int2 *item;
    const int4 *panel;           // x = first panel row, y = rows, z = ls * nmax + the leaf's first live column, w = live columns
    const long long *rowSrc;     // per panel row: offset in W of (that separator row, the panel's first column): where the
                                 // factorisation leaves it
    const long long *rowBase;    // per panel row: offset in `packed` of the row's copy (the rows of a panel one behind the other)
    double *packed;              // the panels, packed after every factorisation (twolevel_pack_kernel): what the two kernels stream
    const int *rowPos;           // per panel row: ls * nmax + the separator row's padded position
    double *cbuf;                // per panel row: c = M[row, leaf columns] . r_leaf
    const int *gPtr, *gIdx;      // per padded position (nParts * nmax + 1, CSR): the panel rows whose c is subtracted there
    double *rpad2;               // the right-hand sides with t_G in the separator positions (what the tile kernels read)
};

// ---- subdomains owned by this rank ---------------------------------------------------------------
struct DevParts {
    DevTwoLevel tl;
    int nParts;             // owned
    int nmax;               // padded scalar size of every owned dense block (multiple of 64) = its lda
    int *dofmap;            // owned * nmax: padded local position -> global scalar dof, -1 = padding
    double *W;              // factor storage of the owned subdomains (RowTile): H_s, then X_s = chol(H_s)^-1 with memory
                            // row i = row i of X_s (column-major upper factor Q = R^-1 of H_s = R^T R)
    RowTile *rt;            // owned * (nmax / 64) row blocks
    int ntiles;             // back-solve jobs, heavy first:
    int4 *tile;             //   (part, first row, tile index within the part | rows << 16, first column)
    int nquad;              // behind the ntiles one-tile jobs: nquad packs of four SMALL tiles (rows of at most BS_WAVE columns; one
                            //   wavefront each, backsolve_wave_tile), padded with empty entries (rows = 0): tile[ntiles + 4 k ..]
    int ntilesWide;         // the first ntilesWide tiles have rows of more than BS_NARROW columns (512-thread kernel)
    int maxTileLen;         // longest row of a tile in `tile` (<= BS_LONG: longer tiles are in `ltile`)
    // tiles whose rows exceed BS_LONG columns: two-phase back-solve over column chunks (k_backsolve.hip)
    int nltiles, nlwork;    // long tiles; (long tile, chunk) work items
    int4 *ltile;            // same encoding as `tile`
    int2 *lwork;            // (index into ltile, chunk)
    int maxChunks;          // chunks of the longest long tile
    double *tdots;          // nltiles * maxChunks * 64 chunk partials of the row dot products
    int4 *tileByPart;       // the register-kernel tiles grouped by part (GSDD solves one subdomain at a time)
    int4 *ltileByPart;      // the long-row tiles grouped by part, and their work items (x = index into ltileByPart)
    int2 *lworkByPart;
    int nbmax;              // max row tiles per part
    int2 *trange;           // owned * nbmax: columns [first, end) each tile of a part contributes to
    int *rp_ptr, *rp_idx;   // reduce_partial_p: per part and group of 16 columns (CSR, owned * (nmax / 16 + 1)) the tiles that hold it, ascending
    double *ppart;          // owned * nbmax * nmax partial results of the back-solve tiles
    double *psub;           // owned * nmax per-part results (padded positions)
    double *rpad;           // owned * nmax right-hand sides in padded order (zeros on the padding): what the back-solve
                            // tiles read, contiguous -- filled by build_qpad (loop) or gathered from q (launch_gemv)
    // merge: per vertex list of positions in psub (all parts on this rank), CSR over vertices
    int *vp_ptr, *vp_off;
    // the same merge straight from the tile partials: per global scalar dof (CSR mt_ptr over 3 nV) the offsets into ppart
    // that make up its value, subdomain after subdomain; the first entry of a subdomain is stored complemented (~off)
    int *mt_ptr, *mt_ent;
    // round 6: the same lists INTERLEAVED by wavefront -- entry q of the 64 consecutive dofs [64 w, 64 w + 64) is contiguous at
    // mt_il[mt_wave[w].x + 64 q + lane], q < mt_wave[w].y (the longest list of the 64; shorter ones end in MT_PAD) -- so that a
    // list load of a wave touches two 128-byte lines instead of ~34 (a lane's own list is ~17 x 4 contiguous bytes, the lanes'
    // lists lie one behind the other: every load instruction of the CSR walk strides through 64 x 68 bytes)
    int2 *mt_wave;
    int *mt_il;
    int splitMerge;         // big meshes: no such lists -- reduce_partial_p (coalesced, in the subdomains' own order) leaves psub
                            // and the merge kernels gather from it through vp_ptr / vp_off (round 4: 1 M tets, loop -2 ms)
    int *dup;               // nV (global multiplicity, DOTTimeStepper.cpp:47-56)
    // dense fill list
    int nfill;
    long long *fill_dst;    // per scalar of every 3x3 block (9 nfill): its offset in W, or -1 (not stored)
    int *fill_src;          // block index in Hval
    int npad;               // identity padding entries
    long long *pad_dst;
};

constexpr int MT_PAD = -2147483647 - 1;   // end of a dof's interleaved list (no offset, plain or complemented, has this value)

struct LbfgsArgs {
    int m;
    const double *s[HIST_MAX];  // chronological, oldest first
    const double *y[HIST_MAX];
    double ys[HIST_MAX];        // y_i . s_i
    double sy[HIST_MAX][HIST_MAX];  // sy[i][j] = s_i . y_j
};

// Canonical order of every host- or controller-side sum over per-block partials: SUM_CHUNKS contiguous
// chunks summed left to right, then the chunk sums left to right.  Host loop and device controller use
// the same order, so they agree bit for bit, and the dependent add chain is short enough for one wave.
constexpr int SUM_CHUNKS = 16;
template <class Get>
__host__ __device__ inline double chunked_sum(int n, Get get)
{
    const int L = (n + SUM_CHUNKS - 1) / SUM_CHUNKS;
    double tot = 0.0;
    for (int c = 0; c < SUM_CHUNKS; ++c) {
        double acc = 0.0;
        const int e = (c + 1) * L < n ? (c + 1) * L : n;
        for (int k = c * L; k < e; ++k) acc += get(k);
        tot += acc;
    }
    return tot;
}

__host__ __device__ inline int pair_band(double a0) { return a0 < 0.7 ? 0 : (a0 < 0.9 ? 1 : 2); }

struct XiArgs {
    double xi[HIST_MAX];
};

// L-BFGS loop state resident in HBM (single-GPU path).  loop_control_kernel advances it after every
// line-search trial exactly as the host loop of dotmi_step would, and the loop kernels take their
// operands from it, so the host can enqueue iterations ahead of the device instead of synchronising
// once per trial (that round trip was ~30 us of every ~180 us iteration on bar17K).
struct DevLoop {
    int status;  // 0 running, 1 converged, 2 iteration cap, 3 line search collapsed (alpha == 0)
    int phase;   // 0: the next slot computes a new direction; 1: it retries the current one with `alpha`
    int iter, iterCap, hist, halvings, evals, slots;
    int notifyFrom;  // the controller posts {status, slots} to pinned host memory from this slot on (and at the end)
    int pad0;
    double tol, dtSq;
    double alpha;          // step of the next retry
    double E_cur, g2_cur;  // at x_cur
    double E0, g2_0;       // after initX (start of the step)
    double *x_cur, *x_trial, *g_cur, *g_trial;
    int slot;              // free history slot that receives the pair of the running trial
    int order[HIST_MAX + 1];
    double b[HIST_MAX];    // s_i . g_cur
    LbfgsArgs L;           // chronological view of the stored pairs (pointers, ys, sy)
    XiArgs X;              // first half of the two-loop for the next direction
    double *S[HIST_MAX + 1], *Y[HIST_MAX + 1];
    double *log_alpha, *log_E, *log_g2;
    int *slot_kind;        // per slot: 1 new direction, 2 retry (slots after the end are not logged)
    int logCap, kindCap;
    // early back-solve (enqueue_loop_slot): u = -M g of the accepted iterate and M y_i of the stored pairs (same slots as Y)
    int pairNew;           // the controller's last accept stored a pair (its M y goes to MY[order[m - 1]])
    int abortEpoch;        // `slots` value of the last slot whose trial was rejected or that ended the loop: the speculative
                           // back-solve of that slot (which knows its epoch) stops when it sees it
    // Held back-solves (round 4): the tiles of a slot whose trial is EXPECTED to be rejected do not start streaming beside
    // the controller; they wait for its verdict (holdVerdict = 2 * slot + (rejected or last iterate)) and leave at once
    // when it is a rejection.  The expectation is what happened to the last trial of the same kind (first trial of an
    // iteration / retry after a halving), kept by the controller; holdNext is its forecast for the slot that follows.
    int holdEnable, holdNext, holdVerdict;
    // Paired trial: the slot evaluated alpha_0 / 2 in full and the energy at alpha_0.  alpha_0 rejected (as expected): the
    // controller goes on with the half step as if a retry slot had run.  alpha_0 acceptable: redo = 1, the next slot evaluates
    // alpha_0 in full as a plain trial (phase 1, alpha = alpha_0) and keeps the statistics of a first trial.
    int redo;
    int heldSlots, heldRejected;   // slots whose tiles were told to wait / of those, rejected (statistics)
    int pairSlots, pairRedo;       // paired slots / of those, the ones whose full step was acceptable (statistics)
    // is "alpha_0 < 1" a sign of a rejection on THIS workload?  Learned from every first trial with alpha_0 < 1, paired or not:
    // one saturating counter (0 .. 3) per band of alpha_0 (< 0.7, < 0.9, < 1); the element pass pairs when the band's counter
    // is 3 (stiff monkey: nine in ten such trials are rejected; refined horse, early steps: one in three -- there a redone
    // slot costs more than a saved retry, so the rule has to be sure)
    int pairCtr[3], pairPad;
    // two-level forecast per kind of trial (0: first trial of an iteration, 1: retry): the last two outcomes of the kind
    // select one of four saturating counters (0 .. 3, >= 2 forecasts a rejection) -- a plain "same as last time" is wrong
    // every time on the alternating pattern stiff steps show (measured: profiles/r04_hold.txt)
    int predHist[2], predCtr[2][4];
    // Speculative unit step (round 6, k_dirstep.hip): the SpMV partial rows the controller of such a step sums for alpha_0, the
    // lower bound of the clamp; statistics: new-direction slots that speculated / of those, redone with alpha_0 < 1; first trials
    // of the step / of those, the ones whose estimate was the unit step (every controller counts these two: the next step's gate)
    const double *specPartials;
    double alphaMin;
    int specSlots, specRedo, firstTrials, unitFirst;
    double *u_old, *MY[HIST_MAX + 1];
    // H s_i of the stored pairs (same slots as S): H p = H z + sum_j delta_j (H s_j), H s_new = alpha H p (spmv_zp_kernel)
    double *HS[HIST_MAX + 1];
    // Round 6: what the loop's kernels used to find by index -- S[slot], HS[order[j]], MY[order[i]] -- RESOLVED by whoever changes
    // slot / order (devloop_resolve: the host at the start of a step, the controller after an accepted trial).  Every kernel of
    // the loop starts with a read of this struct from memory another XCD wrote (~1 us per dependent scalar load): with the
    // pointers resolved and all of a kernel's loop-state loads issued in front of its first branch that is ONE round trip
    // instead of three (tools/prof_loopkern.sh: 2.9 us of the gather's 6.2 and of merge_early's 9.6 were this prologue).
    double *s_new, *y_new, *hs_new;          // S[slot], Y[slot], HS[slot]: the pair of the running trial
    double *my_new;                          // MY[order[m - 1]]: where M y of the newest stored pair goes (merge_early)
    const double *Lmy[HIST_MAX], *Lhs[HIST_MAX];   // chronological views like L.s / L.y: M y_i, H s_i of the stored pairs
};
// the resolved views from slot / order / L.m (host: before the state is uploaded; controller: after it has changed them)
template <class DL>
__host__ __device__ inline void devloop_resolve(DL &C)
{
    C.s_new = C.S[C.slot];
    C.y_new = C.Y[C.slot];
    C.hs_new = C.HS[C.slot];
    const int m = C.L.m;
    for (int i = 0; i < HIST_MAX; ++i) {
        C.Lmy[i] = i < m ? C.MY[C.order[i]] : nullptr;
        C.Lhs[i] = i < m ? C.HS[C.order[i]] : nullptr;
    }
    C.my_new = m > 0 ? C.MY[C.order[m - 1]] : nullptr;
}
// owner exchange: the vertices a rank holds -- the vector kernels of the loop then visit only these (everything else is zero
// and stays zero); v == nullptr: every vertex
struct VList {
    const int *v = nullptr;
    int n = 0;
};
// the controller's operands when it runs as one workgroup of another launch (launch_gemv)
struct CtlArgs {
    DevLoop *ctl;
    const double *partE, *partR, *alpha_dev;
    int *flags_host;
    int nbE;
    int init;   // bit 0: the evaluation at the start of the step (nothing to decide yet); bit 1: a step with paired trials -- the
                // full step's energy partials follow partE at + 2 ELEM_NB_MAX, alpha_dev[1] > 0 marks a paired slot
};

// ---- kernel launchers (k_*.hip) ------------------------------------------------------------------
// Launchers with a trailing `ctl` run in device-loop mode when it is non-null: operands that change from
// iteration to iteration come from *ctl and the kernel returns at once when the loop has ended (or, for
// the direction kernels, when the slot is a line-search retry).
// x = x0 + alpha * p; alpha = alpha_scale * clamp(-pg/pHp) from SpMV partials when use_partials
void launch_step_forward(int n, const double *x0, const double *p, double *x, const double *spmv_partials,
                         double alpha_host, int use_partials, double alpha_min, double *alpha_out,
                         double *alpha_out_host, hipStream_t st, const DevLoop *ctl = nullptr, VList vl = VList());
// element pass: partial energy sums (+ inertia) and, optionally, element gradients
// grad != 0: the per-(patch, vertex) partial gradients go to PT.gpart (read by launch_vertex_gather)
// step (device loop only): the line-search step x_trial = x_cur + alpha p is taken inside the element pass
// (alpha from the SpMV partials as in launch_step_forward) instead of by a launch of its own
struct StepArgs {
    const double *p, *spmv_partials;
    double *alpha_out;       // [0] the step of the trial the gather and the controller work on, [1] paired trial: the full step (else 0)
    // alpha_min < 0: PAIRED launch (DOTMI_PAIR_TRIALS, elem_patch_kernel) with the lower bound -alpha_min -- twice as many
    // workgroups; when the step estimate alpha_0 is below 1 -- the first trial is then rejected nine times out of ten on
    // back-tracking workloads -- the first half evaluates alpha_0 / 2 in full and the second half the ENERGY at alpha_0, into the
    // second half of the partials array (+ 2 ELEM_NB_MAX).  (Packed into the sign: every byte added to the kernel arguments of the
    // loop's kernels showed up as time -- profiles/r05_paired_trials.txt F.)
    double alpha_min;
};
void launch_elem_energy_grad(const DevMesh &M, const DevPatches &PT, int mat, double dtSq, const double *x,
                             const double *xt, int v0, int v1, int grad, double *partials, int *nblocks_out,
                             hipStream_t st, const DevLoop *ctl = nullptr, const StepArgs *step = nullptr);
// the PAIR instantiation of the same launch (k_element.hip): StepArgs::alpha_min < 0 makes it a paired launch
void launch_elem_energy_grad_pair(const DevMesh &M, const DevPatches &PT, int mat, double dtSq, const double *x,
                                  const double *xt, int v0, int v1, int grad, double *partials, int *nblocks_out,
                                  hipStream_t st, const DevLoop *ctl = nullptr, const StepArgs *step = nullptr);
// vertex gather of element gradients + inertia; optional L-BFGS pair + stats partials
struct GatherArgs {
    const double *x, *xt, *g_old, *p, *alpha_dev;
    double *g_new, *s_new, *y_new;
    int make_pair;
    int iv0, iv1;  // vertex range whose inertia term m_v (x_v - x~_v) this rank adds
    int stage;     // device loop, sharded element pass: g_new is a staging buffer (kept as passed), no pair
    // early back-solve: -g is also written straight into the padded right-hand sides of every subdomain that holds the
    // vertex (DevParts::vp_ptr / vp_off), so no build_qpad launch stands between the gather and the back-solve
    const int *vp_ptr, *vp_off;
    double *rpad;
    const double *hp;   // early order with the fused direction kernel: H p of the running trial; H s_new = alpha H p goes to
                        // DevLoop::HS[slot] beside s_new
    // owner exchange (DOTMI_FLAG_OWNER_EXCHANGE): this rank adds the inertia term of the vertices it OWNS (instead of a slice
    // [iv0, iv1)) and pair_stats sums its statistics over them only (every vertex counted once over the ranks)
    const uint8_t *ownMask;
    const int *vlist;   // owner exchange: the held vertices (the gather and pair_stats visit only these); nullptr: all
    int nlist;
    // owner exchange with the statistics in the gradient's packet (pre = 1): on the vertices only this rank holds the gather does
    // its whole work (the gradient is complete there); on the shared ones (kind bit 1) this rank's part of the gradient goes to
    // `gshare` and its share of the sums that are linear in the gradient to the partials; pair_stats visits them after the exchange
    const uint8_t *kind;
    int pre;
    double *gshare;
};
void launch_vertex_gather(const DevMesh &M, const DevPatches &PT, const GatherArgs &a, const LbfgsArgs &L,
                          double *partials, hipStream_t st, const DevLoop *ctl = nullptr);
// pair + statistics from already summed gradients (multi-GPU: after the all-reduce of g)
void launch_pair_stats(int n, const GatherArgs &a, const LbfgsArgs &L, double *partials, hipStream_t st,
                       const double *gsrc = nullptr, const DevLoop *ctl = nullptr);
// two-loop, first half:  b_i = s_i . g  partials
void launch_multidot(int n, const double *v, const double *const *vecs, int m, double *partials,
                     hipStream_t st);
// q = -g - sum_j xi_j y_j with xi from partials (b) and SY
void launch_build_q(int n, const double *g, const LbfgsArgs &L, const double *xi_host, double *q,
                    hipStream_t st, const DevLoop *ctl = nullptr);
// subdomain back-solve: psub_s = X_s^T (X_s q[dofmap_s])
// q == nullptr: P.rpad already holds the right-hand sides (launch_build_qpad); else they are gathered from q first
// ca: the loop controller runs as workgroup 0 of the (first) launch, beside the tiles, which then ignore the retry phase
// (the back-solve is speculative: issued on the trial gradient before the controller has accepted the trial)
void launch_gemv(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl = nullptr,
                 hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, const CtlArgs *ca = nullptr, int spec = 0);
// ... the PAIR instantiation: the controller that understands paired slots (CtlArgs::init bit 1)
void launch_gemv_pair(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl = nullptr,
                      hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, const CtlArgs *ca = nullptr, int spec = 0);
// ... the controller of a step that takes the unit step speculatively (checks alpha_0 afterwards; DevLoop::specPartials)
void launch_gemv_spec(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl = nullptr,
                      hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, const CtlArgs *ca = nullptr, int spec = 0);
// (a.alpha_min < 0: a PAIRED launch -- twice as many workgroups, the second half the energy of the full step: StepArgs::alpha_min)
void launch_elem_vertex(const DevVPatches &VP, int mat, const ElemVertArgs &a, hipStream_t st, const DevLoop *ctl);
void launch_gemv_pair_vp(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl = nullptr,
                         hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, const CtlArgs *ca = nullptr, int spec = 0);
// ... the controller of a step on vertex patches (k_elemvert.hip): it sums as many statistic rows as there are patches (CtlArgs::nbE)
void launch_gemv_vp(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl = nullptr,
                    hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, const CtlArgs *ca = nullptr, int spec = 0);
// the direction kernel and the first trial's element pass at the unit step in one launch of two workgroup populations
// (k_dirstep.hip): replaces launch_spmv_zp + launch_elem_energy_grad in the slots of a speculating step
bool dirstep_fits(const DevPatches &PT);   // (meshes of at most 512 patches: every patch a workgroup)
void launch_dirstep(const DevMesh &M, const DevPatches &PT, int mat, double dtSq, const double *xt, double *partE, int *nblocks_out,
                    const double *Hval, const double *z, const double *c_partials, double *p, double *Hp, double *partS,
                    hipStream_t st, const DevLoop *ctl, const StepArgs &sa);
// rpad_s[k] = q[dofmap_s[k]] with q = -g - sum_j xi_j y_j formed on the fly (same operations as build_q), 0 on padding
// spec (device loop, early back-solve): 1: rpad = -g_cur, 2: rpad = -g_trial whatever the phase; no history terms
void launch_build_qpad(const DevParts &P, const double *g, const LbfgsArgs &L, const double *xi_host, hipStream_t st,
                       const DevLoop *ctl = nullptr, int spec = 0);
// early back-solve: u = merge(tile partials) / dup = -M g;  M y of the newest pair = u_old - u;  z = u - sum_j xi_j M y_j
// (+ partial dots y_i . z);  first: start of the step (no history, u_old is only set)
void launch_reduce_partial(const DevParts &P, hipStream_t st, const DevLoop *ctl = nullptr);
void launch_twolevel_pack(const DevParts &P, hipStream_t st);   // after a factorisation (two-level form): the panels' rows -> P.tl.packed
// zsum: the all-reduced sum (over all ranks' subdomains) of the undivided partial merges, in a staging buffer
// ownMask (owner exchange): the y_i . z partials over the vertices this rank owns only
// kind, pre, zshare (owner exchange with the y_i . z in the packet; zsum = nullptr: the kernel merges this rank's tiles itself):
// before the exchange the whole work on the vertices only this rank holds; on the shared ones (kind bit 1) this rank's part of the
// sum goes to zshare and its share of y_i . z to the partials; afterwards a launch over the shared vertices (zsum = the unpacked
// sums, partials = nullptr)
void launch_merge_early(const DevMesh &M, const DevParts &P, double *z, double *partials, int first, hipStream_t st,
                        const DevLoop *ctl, const double *zsum = nullptr, const uint8_t *ownMask = nullptr, VList vl = VList(),
                        const uint8_t *kind = nullptr, int pre = 0, double *zshare = nullptr, double *partialsT = nullptr);
// owner exchange: the entries of the vertices held by more than one rank, packed / unpacked (idx: their vertex ids);
// tail: `ntail` further scalars copied from / to tailp behind the packed entries
// red0 / red1: partial arrays whose rows workgroup 0 of the pack sums into the packet's tail (pack[dst ...]): `cols` columns of
// `rows` rows, or (combine) s0 * column 0 + s1 * column 1 into one entry
struct PackRed {
    const double *part;
    int rows, stride, cols, combine, dst;
    double s0, s1;
};
void launch_pack_iface(int nI, const int *idx, const double *src, double *pack, const double *tailp, int ntail, hipStream_t st,
                       const PackRed *red0 = nullptr, const PackRed *red1 = nullptr);
// dst2 / ntail2: the tail behind the first one goes to dst2, with the squares of the packet's summed vector entries added to
// dst2[0] (|g|^2 over the shared vertices)
void launch_unpack_iface(int nI, const int *idx, const double *pack, const uint8_t *heldMask, double *dst, double *tailp, int ntail,
                         hipStream_t st, double *dst2 = nullptr, int ntail2 = 0);
// |v|^2 over the owned vertices -> column 0 of the partial rows;  v := own ? v : 0 (in place)
void launch_masked_norm2(int n, const double *v, const uint8_t *ownMask, double *partials, hipStream_t st, int exact = 0);
void launch_mask_owned(int n, double *v, const uint8_t *ownMask, hipStream_t st);
// one subdomain only (GSDD, DOTTimeStepper.cpp:507-565): psub_s = X_s^T (X_s q[dofmap_s]) for owned part `ls`, whose
// tiles are job[0..njobs); then p = 0 except p[dofs of part ls] = psub_s  (ADMMDDTimeStepper::fill, :1646-1665)
void launch_gemv_part(const DevParts &P, int ls, const int4 *job, int njobs, const int2 *lwork, int nlwork, const double *q,
                      int n, double *p, hipStream_t st);
// z = merge(psub) / dup  (+ partial dots y_i . z)
void launch_merge(const DevMesh &M, const DevParts &P, const LbfgsArgs &L, double *z, double *partials,
                  int with_dots, hipStream_t st, const DevLoop *ctl = nullptr, VList vl = VList());
// partial dots y_i . z only (multi-GPU path after all-reduce)
// p = z + sum_j delta_j s_j, delta from c partials, xi and SY
void launch_build_p(int n, const double *z, const LbfgsArgs &L, const double *c_partials,
                    const double *xi_host, double *p, hipStream_t st, const DevLoop *ctl = nullptr);
// early order, one launch for build_p + spmv_dots: p = z + sum_j delta_j s_j (build_p's statements), H p = H z + sum_j
// delta_j (H s_j) from the cached H s_j, and the partial sums of p.g and p.Hp -- the only sparse product is H z, which
// does not wait for delta
// rows [v0, v1) only for the product and the dots (sharded rows); v1 < 0: every row
// rowMask / ownMask (owner exchange): the product on the rows of the vertices this rank holds (with ITS elements' part of H),
// p.Hp over them, p.g over the vertices it owns
void launch_spmv_zp(const DevMesh &M, const double *Hval, const double *z, const double *c_partials, double *p, double *Hp,
                    double *partials, hipStream_t st, const DevLoop *ctl, int v0 = 0, int v1 = -1,
                    const uint8_t *rowMask = nullptr, const uint8_t *ownMask = nullptr, VList vl = VList(), bool ctrans = false,
                    double *partialsT = nullptr);   // (partialsT: p.g / p.Hp once more, column-major [2][NB_RED])
// Hp = H p on rows [v0,v1), partial sums of p.g and p.Hp
void launch_spmv_dots(const DevMesh &M, const double *Hval, const double *p, const double *g, double *Hp,
                      int v0, int v1, double *partials, hipStream_t st, const DevLoop *ctl = nullptr);
// one wavefront: sums the partials of the finished trial and advances *ctl (accept / halve / stop);
// flags_host (pinned, 2 ints) receives {status, slots done}.  init != 0: the partials are those of the evaluation at the
// start of the step (after initX); the controller only records E and |g|^2
void launch_loop_control(DevLoop *ctl, const double *partE, int nbE, const double *partR,
                         const double *alpha_dev, int *flags_host, hipStream_t st, int init = 0);
// element Hessians (12x12 projected), one wavefront per element in the expansion phase
// elist != null: only the listed elements (row i of He = element elist[i]);  blist != null: only the listed blocks, whose
// contribution lists blk_ptr / blk_ent then index by list position and name rows of that compact He
void launch_elem_hessians(const DevMesh &M, int mat, double dtSq, const double *x, double *He,
                          hipStream_t st, const int *elist = nullptr, int nList = 0);
// mass != null: the diagonal term comes from this array instead of the lumped masses (owner exchange: the owner's share)
void launch_assemble(const DevMesh &M, const double *He, double *Hval, hipStream_t st, const int *blist = nullptr,
                     int nList = 0, const int *blk_ptr = nullptr, const int *blk_ent = nullptr, const double *mass = nullptr);
void launch_dense_fill(const DevParts &P, const double *Hval, hipStream_t st);
// one level of the tile schedule (tile_factor.hpp): one workgroup per task
// fastDiag: the diagonal tasks' 16 x 16 bottom steps in the per-lane 8 x 8 form (k_tilefactor.hip, block_chol_inv<N, FAST>; 512 threads)
void launch_tile_level(const TileTask *tasks, int ntasks, const TileProd *prods, int *info, hipStream_t st, bool fastDiag = true);
// the non-diagonal tasks of a level on half tiles, four workgroups per CU (tile_gemm_kernel)
void launch_tile_gemm(const TileTask *tasks, int ntasks, const TileProd *prods, hipStream_t st);
void launch_tile_flow(const TileTask *tasks, int ntasks, const TileProd *prods, const int *depPtr, const int *depIdx, int *done,
                      int *next, int epoch, int *info, int nwg, hipStream_t st, double waitMs, bool fastDiag = true);
void launch_clear_tiles(double *const *tiles, const int *lds_, int ntiles, hipStream_t st);
// small helpers
void launch_init_x(int nV, const uint8_t *fixed, const double *v, double dt, const double *gdtsq,
                   double *x, hipStream_t st);
void launch_be_update(int nV, const uint8_t *fixed, double *x, double *xn, double *v, double *xt,
                      double dt, const double *gdtsq, hipStream_t st);
void launch_scatter_rows(int n, const int *idx, const double *pos, double *x, hipStream_t st);
void launch_copy(int n, const double *src, double *dst, hipStream_t st);
// sharded subdomains, after the all-reduce of the merged sums: z_v /= dup_v and the y_i . z partials
void launch_zfinish(int nV, const int *dup, const LbfgsArgs &L, double *z, double *partials, hipStream_t st,
                    const DevLoop *ctl = nullptr);

}  // namespace dotmi
