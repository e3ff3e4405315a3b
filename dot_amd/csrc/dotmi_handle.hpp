// dotmi_handle.hpp -- the handle behind the C ABI and what the translation units of libdotmi share (private).
//   dotmi_create.hip      setup: mesh features, partition maps, dissection layout, tile schedule, buffers; host-only planners
//   dotmi_refresh.hip     Hessian refresh + subdomain factorisation (issue / finish / asynchronous verdict)
//   dotmi_collectives.hip all-reduce (RCCL or host hook) and the owner exchange's packets
//   dotmi_loop.hip        the L-BFGS-H loop: slots of the device loop, host loop, GSDD, Newton, dotmi_step
//   dotmi_api.hip         the remaining ABI entry points (state, kernel-level calls, probes, measurement)
//
// Control flow mirrors (paths relative to /root/reference/src)
//   dotmi_create       Optimizer ctor (TimeStepper/Optimizer.cpp:52-196), ADMMDDTimeStepper ctor
//                      (ADMMDDTimeStepper.cpp:44-443: partition -> local maps), DOTTimeStepper ctor +
//                      precompute (DOTTimeStepper.cpp:38-178), Mesh::computeFeatures (Mesh.cpp:589-700)
//   dotmi_step         Optimizer::solve (Optimizer.cpp:327-368) -> DOTTimeStepper::fullyImplicit
//                      (DOTTimeStepper.cpp:273-346) -> solve_oneStep (:384-504) -> Optimizer::lineSearch
//                      (Optimizer.cpp:752-881) ; updateHessianAndFactor (DOTTimeStepper.cpp:349-380)
// The data path is entirely on the device; the host only sequences launches and, in the host-driven loop, evaluates the
// m x m scalar recurrences of the two-loop recursion and takes the accept / halve / converged decisions from one small
// read-back per line-search trial.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/dotmi.h"
#include "dotmi_internal.hpp"
#include "elem_math.hpp"
#include "partition.hpp"
#include "patches.hpp"
#include "vpatches.hpp"


namespace dotmi {

extern std::string g_create_error;

inline double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// single wave: out[j] = sum_b partials[b*stride+j]; combine: out[0] = s0*sum0 + s1*sum1   (dotmi_collectives.hip)
__global__ void reduce_rows_kernel(const double *__restrict__ partials, int nblocks, int stride, int K, double s0, double s1,
                                   int combine, double *__restrict__ out);

}  // namespace dotmi

using namespace dotmi;

// ---- tuning / ablation switches -----------------------------------------------------------------------------------
// Read ONCE per dotmi_create from the environment; every one is optional and the defaults are the product path.  They
// exist for the A/B measurements logged under profiles/ (tools/ab.sh) and for tests that force a code path; none changes
// results beyond rounding.  Listed in include/dotmi.h ("Environment") and DESIGN.md section 10.
struct Tuning {
    int ndLevels = -1;        // DOTMI_ND_LEVELS      depth of the nested dissection (-1: nd_default_levels)
    int ndMin = ND_MIN_SPLIT; // DOTMI_ND_MIN         smallest region (scalars) that is still split
    bool wavePacks = true;    // DOTMI_WAVE_PACKS=0   small back-solve tiles as one-tile jobs like the others instead of four per workgroup
    int tilePasses = 4;       // DOTMI_TILE_PASSES    most passes (of 8 rows) a back-solve tile of rows beyond 1024 columns takes (8: 64-row tiles)
    int tileRows = 0;         // DOTMI_TILE_ROWS      rows per back-solve tile (0: 64, or 32 for few subdomains)
    int tileRowsLong = 0;     // DOTMI_TILE_ROWS_LONG rows per back-solve tile when the rows have more than 1536 columns (0: as the
                              //                      other rows, or ~256 KB tiles where few subdomains leave the launch bound by
                              //                      its longest tile)
    int twoLevel = -1;        // DOTMI_TWO_LEVEL      1: the back-solve in its two-level form (leaves against the separator complement,
                              //                      DevTwoLevel); 0: the explicit inverse in one pass; -1: two-level where one pass would stream >= 240 MB
    int splitMerge = -1;      // DOTMI_SPLIT_MERGE    1 / 0: the merge as reduce_partial_p + a gather from psub (the early order included) /
                              //                      as one walk over the tile partials; default: split from 400 k scalar dofs
    bool fuseLog = false;     // DOTMI_FUSE_LOG       print the fused-leaf units
    bool factorGraph = true;  // DOTMI_FACTOR_GRAPH=0 the level launches of the factorisation issued directly instead of replayed as a hipGraph
    int shardElems = -1;      // DOTMI_SHARD_ELEMS    0 / 1: force the replicated / sharded element pass (-1: by size)
    int shardHess = -1;       // DOTMI_SHARD_HESS     0 / 1: force the replicated / sharded once-per-step phase
    int timeStride = 8;       // DOTMI_TIME_STRIDE    DOTMI_FLAG_TIME_BACKSOLVE brackets every n-th back-solve
    int patchElems = 0;       // DOTMI_PATCH_ELEMS    elements per patch of the element pass (0: default)
    int tileSplit = -1;       // DOTMI_TILE_SPLIT     0 / 1: one task kernel per level / diagonal and half-tile kernels side by side
                              //                      (-1: the latter above 64 subdomains, where the factorisation is throughput-bound)
    int tileEagerMin = 0;     // DOTMI_TILE_EAGER_MIN early products a critical-path tile task may keep
    int fastDiag = 1;         // DOTMI_FAST_DIAG      1 / 0: the diagonal tile tasks' 16 x 16 bottom steps on 4 x 4 blocks every lane factors for
                              //                      itself (12.0 us per 64 x 64 step) / one row per lane with v_readlane operands (15.3 us)
    int tileFlow = -1;        // DOTMI_TILE_FLOW      1: the factorisation as ONE launch of persistent workgroups with per-task
                              //                         dependencies (tile_flow_kernel) instead of one launch per level; 0: never;
                              //                         default: where a level holds fewer tasks than the GPU holds workgroups
    int tileFlowWaitMs = 2000;   // DOTMI_TILE_FLOW_WAIT_MS  a task that waits longer for one of its dependencies gives up (error)
    int tileEagerMinRmul = -1; // DOTMI_TILE_EAGER_MIN_RMUL early products the last task of a Q tile may keep (-1: as the others; 0: none)
    bool fuseDir = true;      // DOTMI_FUSE_DIR=0     (early order) build_p and spmv_dots as two launches instead of one on cached H s_j
    bool fuseStep = true;     // DOTMI_FUSE_STEP=0    (early order) step_forward as a launch of its own instead of inside the element pass
    bool earlyAbort = true;   // DOTMI_EARLY_ABORT=0  (ablation) speculative back-solves run to their end even when the trial is rejected
    int pairTrials = -1;      // DOTMI_PAIR_TRIALS    -1 (default): paired line-search trials (StepArgs::pairBlocks) in a step whose predecessor
                              //                      halved in at least a quarter of its iterations; 1: in every step; 0: never
    int vertexPatches = -1;   // DOTMI_VERTEX_PATCHES 0: the element pass and the vertex gather of a trial as two launches on element patches
                              //                      (until round 5); -1 (default): as ONE launch on vertex patches (k_elemvert.hip) where every
                              //                      patch is a workgroup of its own (<= 512 patches, one rank); 1: the same rule (reserved)
    int specStep = 0;         // DOTMI_SPEC_STEP      the unit step taken speculatively beside the direction kernel (k_dirstep.hip): 0 (default)
                              //                      never -- measured: the fused launch is as long as its two parts, profiles/r06_spec_step.txt;
                              //                      -1: in a step whose predecessor's first trials took the unit estimate at least nine times
                              //                      in ten; 1: in every step
    bool earlyHold = true;    // DOTMI_EARLY_HOLD=0   the back-solve of a trial that is expected to be rejected still starts speculatively
    int earlyBs = 2;          // DOTMI_EARLY_BACKSOLVE 0: the back-solve after the controller, on q; 1: speculatively on the trial
                              //                      gradient with the controller inside its launch, in the steps where
                              //                      the last step's counts say it pays (run_device_loop); 2 (default): in
                              //                      every step
    static int geti(const char *name, int dflt)
    {
        const char *ev = getenv(name);
        return ev ? atoi(ev) : dflt;
    }
    static Tuning from_env()
    {
        Tuning t;
        t.wavePacks = geti("DOTMI_WAVE_PACKS", 1) != 0;
        t.tilePasses = std::min(8, std::max(1, geti("DOTMI_TILE_PASSES", 4)));
        t.ndLevels = geti("DOTMI_ND_LEVELS", -1);
        if (t.ndLevels < -1) t.ndLevels = 0;
        t.ndMin = std::max(128, geti("DOTMI_ND_MIN", ND_MIN_SPLIT));
        if (const char *ev = getenv("DOTMI_TILE_ROWS")) t.tileRows = std::min(64, std::max(8, atoi(ev) / 8 * 8));
        t.tileRowsLong = geti("DOTMI_TILE_ROWS_LONG", 0);
        if (t.tileRowsLong > 0) t.tileRowsLong = std::min(64, std::max(8, t.tileRowsLong / 8 * 8));
        t.splitMerge = geti("DOTMI_SPLIT_MERGE", -1);
        t.twoLevel = geti("DOTMI_TWO_LEVEL", -1);
        t.fuseLog = getenv("DOTMI_FUSE_LOG") != nullptr;
        t.factorGraph = geti("DOTMI_FACTOR_GRAPH", 1) != 0;
        t.shardElems = geti("DOTMI_SHARD_ELEMS", -1);
        t.shardHess = geti("DOTMI_SHARD_HESS", -1);
        t.timeStride = std::max(1, geti("DOTMI_TIME_STRIDE", 8));
        t.patchElems = std::max(0, geti("DOTMI_PATCH_ELEMS", 0));
        t.tileSplit = geti("DOTMI_TILE_SPLIT", -1);
        t.tileEagerMin = std::max(0, geti("DOTMI_TILE_EAGER_MIN", 0));
        t.tileFlow = geti("DOTMI_TILE_FLOW", -1);
        t.fastDiag = geti("DOTMI_FAST_DIAG", 1);
        t.tileFlowWaitMs = std::max(1, geti("DOTMI_TILE_FLOW_WAIT_MS", 2000));
        t.tileEagerMinRmul = geti("DOTMI_TILE_EAGER_MIN_RMUL", -1);
        t.earlyBs = geti("DOTMI_EARLY_BACKSOLVE", 2) != 0 ? 2 : 0;   // (1, round 3's per-step rule, now means "on")
        t.earlyAbort = geti("DOTMI_EARLY_ABORT", 1) != 0;
        t.earlyHold = geti("DOTMI_EARLY_HOLD", 1) != 0;
        t.pairTrials = geti("DOTMI_PAIR_TRIALS", -1);
        t.specStep = geti("DOTMI_SPEC_STEP", 0);
        t.vertexPatches = geti("DOTMI_VERTEX_PATCHES", -1);
        t.fuseStep = geti("DOTMI_FUSE_STEP", 1) != 0;
        t.fuseDir = geti("DOTMI_FUSE_DIR", 1) != 0;
        return t;
    }
};

struct dotmi_handle {
    Tuning tune;
#ifdef DOTMI_TEST_HOOKS
    int testIterDelta = 0;   // fault injection for tests/test_gpu_two_ranks.py (libdotmi_testhooks.so only)
    int testFailRefresh = 0, testRefreshCount = 0;   // DOTMI_TEST_FAIL_REFRESH=k: the k-th factorisation reports a bad pivot
#endif
    // configuration
    int nV = 0, nT = 0, n = 0, mat = 0, hist = 5, iterCap = 10000;
    double dt = 0, dtSq = 0, grav[3] = {0, 0, 0}, gdtsq[3] = {0, 0, 0}, relTol = 1e-5, alphaMin = 0.1;
    double targetGRes = 0, density = 0;
    int device = 0, rank = 0, world = 1, flags = 0;
    bool dist = false;  // world > 1, or DOTMI_FLAG_FORCE_DIST: subdomains (factor + back-solve) are sharded
    // element pass + SpMV rows sharded too (one more all-reduce per trial): only pays on big meshes -- for a
    // 86k-tet mesh the whole element pass is 18 us, cheaper than any collective
    bool shardElems = false;
    std::string err;

    // host copies
    std::vector<int> T, epart, vpart;
    std::vector<uint8_t> fixed;
    std::vector<double> Xrest, A, vol, mass, mu, lam;
    int nPartsAll = 0, p0 = 0, p1 = 0;  // owned global parts [p0,p1)
    std::vector<std::vector<int>> partVerts;  // all parts: ascending global vertex ids
    std::vector<int> dup;
    std::vector<NdNode> nd;                 // nested-dissection layout shared by the owned parts (root = 0)
    std::vector<std::vector<int>> partPos;  // owned parts: padded scalar position of partVerts[p][i]
    std::vector<int> partTilePtr;           // owned parts: range of each part's tiles in DevParts::tileByPart
    std::vector<int> partLworkPtr;          //   and of its long-row work items in DevParts::lworkByPart

    // device
    hipStream_t st = nullptr;
    ncclComm_t comm = nullptr;
    void (*arCb)(void *, double *, int64_t) = nullptr;   // host all-reduce hook (dotmi_params::allreduce) instead of RCCL
    void *arCtx = nullptr;
    double *arStage = nullptr;                          // pinned staging of the hook's payload
    size_t arCap = 0;
    double *ctrlDev = nullptr;                          // RED_K + 2 doubles: control scalars of a trial (rank 0's are used)
    double ctrl[RED_K + 2] = {0};                       // world > 1: [E, alpha, R[0..RED_K)] of the last trial as adopted from rank 0
    DevMesh M{};
    DevParts P{};
    int *elist = nullptr;
    // sharded once-per-step refresh (N > 1 with a sharded element pass): this rank computes the element Hessians of the
    // elements that touch its vertices only and assembles the block rows it reads only
    bool shardHess = false;
    int *hessElems = nullptr, *hessBlk = nullptr, *hessBlkPtr = nullptr, *hessBlkEnt = nullptr;
    int nHessElems = 0, nHessBlk = 0;
    // level-scheduled tile factorisation (tile_factor.hpp)
    bool tileMode = false;
    bool twoLevel = false;                       // the factors are in the two-level form (DevTwoLevel)
    TileTask *ttasks = nullptr;
    TileProd *tprods = nullptr;
    double **tclear = nullptr;
    int *tclearLd = nullptr;
    std::vector<long long> rtOff;   // host copy of the RowTile table (dotmi_part_matrix)
    std::vector<int> rtLd, rtC0;
    std::vector<long long> rtOffM;  // two-level form: the separators' row blocks' second range (their sub-tree's leaf columns)
    std::vector<int> rtLdM, rtC0M;
    size_t wTotal = 0;
    double *W2 = nullptr;             // tile factorisation: the work buffer (H, then R), laid out like P.W (which holds Q only)
    int nTclear = 0;
    std::vector<int> tlevelStart, tlevelDiag;
    bool tileSplit = false;
    int predState[10] = {0, 0, 1, 1, 1, 1, 1, 1, 1, 1};   // DevLoop::predHist / predCtr between the steps
    int heldSlots = 0, heldRejected = 0;                  // held back-solves of the last step (DevLoop::holdNext)
    bool tileFlow = false;            // dataflow factorisation (tile_flow_kernel)
    bool fastDiag = false;            // diagonal tasks with the per-lane 8 x 8 bottom steps (block_chol_inv<N, true>)
    int *tdepPtr = nullptr, *tdepIdx = nullptr, *tdone = nullptr, *tnext = nullptr;
    int tileEpoch = 0, nTtasks = 0, tileFlowWg = 0;
    hipStream_t stDiag = nullptr;              // side stream of the diagonal-block tasks
    std::vector<hipEvent_t> tFork, tJoin;      // per level
    double tileFlops = 0;
    DevPatches PT, PTall;   // element patches: this rank's own elements / all elements (same unless shardElems)
    // the patches a SPECULATING step works on (k_dirstep.hip): the launch pays when its three workgroup populations are resident
    // together (512 workgroups at the direction rows' register count), so where the 256-element patches are too many for that
    // (bar17K: 256 + 337 + 68) such a step takes patches of 512 elements (256 + 169 + 68).  One patch set per step -- the
    // start-of-step evaluation, the trials and the gathers of a step agree on the grouping of the energy partials.
    DevPatches PTspec;
    bool specFits = false;
    // vertex patches (vpatches.hpp): the trial's element pass + gather in one launch (k_elemvert.hip); vpFits: this handle's loop may
    // use them (one rank, early order with the fused kernels, every patch a workgroup); vpNow: the running step does (a step that
    // pairs or speculates keeps the element patches -- one patch set per step, so its energies are grouped alike)
    DevVPatches VP;
    bool vpFits = false, vpNow = false;
    int nOwnElem = 0, v0 = 0, v1 = 0;
    double *x = nullptr, *x_trial = nullptr, *xn = nullptr, *v = nullptr, *xt = nullptr;
    double *g = nullptr, *g_trial = nullptr, *p = nullptr, *q = nullptr, *z = nullptr, *Hp = nullptr;
    double *He = nullptr, *Hval = nullptr, *tmpn = nullptr;
    double *S[HIST_MAX + 1] = {nullptr}, *Y[HIST_MAX + 1] = {nullptr};
    // early back-solve (enqueue_loop_slot): u = -M g of the current iterate, M y_i of the stored pairs (slots as Y)
    bool refreshPending = false;   // DOTMI_FLAG_ASYNC_REFRESH: the last step's refresh is enqueued, not yet judged / timed
    double carryHess = 0, carryFact = 0;   // device times of a refresh resolved outside dotmi_step (reported by the next step)
    bool earlyBs = false;     // possible on this handle (buffers exist)
    bool earlyNow = false;    // chosen for the running step
    int prevIters = -1, prevHalv = 0;   // last step's iterations / line-search halvings (-1: no step yet)
    double *u_old = nullptr, *MY[HIST_MAX + 1] = {nullptr};
    double *HS[HIST_MAX + 1] = {nullptr};   // H s_i of the stored pairs (fused direction kernel of the early order)
    double *partE = nullptr, *partR = nullptr, *partC = nullptr, *partS = nullptr, *partG = nullptr;
    double *partST = nullptr;   // p.g / p.Hp partials, column-major [2][NB_RED] (spmv_zp_body's partialsT): what elem_vertex_kernel reads
    double *partCT = nullptr;   // the y_i . z partials once more, column-major (write_partials' partialsT): what the one-rank loop's direction kernels read
    bool pairNow = false;       // this step's slots launch the element pass twice as wide (enqueue_loop_slot_early)
    int pairSlots = 0, pairRedo = 0;
    bool specNow = false;       // this step's new-direction slots take the unit step speculatively (launch_dirstep)
    const DevPatches &stepPT() const { return specNow ? PTspec : PT; }   // the element patches of the running step
    int specSlots = 0, specRedo = 0;
    int prevFirst = 0, prevUnit = 0;   // last step's first trials / of those, the ones whose estimate alpha_0 was the unit step
    int pairState[3] = {1, 1, 1};   // DevLoop::pairCtr, carried from step to step
    double *gstage = nullptr;   // sharded element pass, device loop: [g (n) ; 0 ; E] staging buffer of the gradient all-reduce
    double *zstage = nullptr;   // sharded subdomains, early order: this rank's undivided partial merge, all-reduced in place
    // owner exchange (DOTMI_FLAG_OWNER_EXCHANGE)
    bool owner = false;
    std::vector<int32_t> firstPart;        // parts of rank r: [firstPart[r], firstPart[r + 1])
    uint8_t *ownMask = nullptr, *heldMask = nullptr;   // nV: this rank owns the vertex / holds it in one of its subdomains
    uint8_t *vkind = nullptr;              // nV: bit 0 = owned by this rank, bit 1 = held by more than one rank
    int *sharedList = nullptr;             // the vertices this rank holds together with other ranks, ascending
    int nShared = 0;
    VList shared() const { return VList{sharedList, nShared}; }
    int *ifaceIdx = nullptr;               // the vertices held by more than one rank (the same list on every rank), ascending
    int nIface = 0;
    int *heldList = nullptr;               // the held vertices, ascending: the loop's vector kernels visit only these
    int nHeld = 0;
    VList held() const { return owner ? VList{heldList, nHeld} : VList(); }
    double *xpack = nullptr;               // 3 nIface + 8 + RED_K doubles: the packed entries (+ E, + the statistics) that travel
    double *massOwn = nullptr;             // nV: lumped mass on the owned vertices, 0 elsewhere
    double *HvalOwn = nullptr;             // block-CSR values of this rank's OWN elements' part of H (+ massOwn): alpha_0's p.Hp
    int *ownBlkPtr = nullptr, *ownBlkEnt = nullptr;   // contribution lists of that assembly (over hessBlk)
    double *partGR = nullptr, *partGC = nullptr;      // all-reduced statistics / y_i.z in row 0 of a zeroed partial array
    DevMesh Mown;                          // the mesh with massOwn for mass (element pass of the owner exchange)
    double *alpha_dev = nullptr;
    int *info_dev = nullptr, *h_info = nullptr;  // per owned part: failing pivot (device / pinned copy)
    bool wDirty = false;                         // W has been through a factorisation (targeted clearing applies)
    bool poisoned = false;                       // the last factorisation failed: the factors in W are garbage
    int *didx = nullptr;
    double *dpos = nullptr;
    size_t dcap = 0;
    std::vector<int32_t> didxHost;  // last scripted index set (uploaded only when it changes)
    double *dposPinned = nullptr;
    hipEvent_t evDir = nullptr;
    std::vector<void *> allocs;
    // pinned host
    double *h_partE = nullptr, *h_partR = nullptr, *h_alpha = nullptr;
    // device-resident loop control (single-GPU path)
    bool gsdd = false;   // DOTMI_FLAG_GSDD
    bool newton = false; // DOTMI_FLAG_NEWTON
    bool devLoop = false;
    DevLoop *ctl = nullptr, *h_ctl = nullptr;  // device / pinned staging
    int *h_flags = nullptr;                    // pinned: {status, slots done}, written by the controller
    double *dlog = nullptr;                    // 3 * logCap doubles
    int *dkind = nullptr;
    std::vector<int> slotTimed;                // per enqueued slot: index of its event pair in evPre, or -1
    std::vector<int> slotKind;                 // last step: kind of every enqueued slot (1 = ran a back-solve)
    int logCap = 0, kindCap = 0;
    int logPending = 0;                        // device-loop log entries not fetched yet
    int prevSlots = 0;                         // slots the previous step's loop took (enqueue-ahead horizon)
    int timeStride = 8;                        // DOTMI_FLAG_TIME_BACKSOLVE brackets every timeStride-th back-solve
    int timeCount = 0;
    int nbE = 0;
    long long mergeEntries = 0;   // tile partials the merge sums (for the byte count of dotmi_bench_kernel)
    int M_nbR() const { return NB_RED; }   // rows of the statistics partials

    // L-BFGS host state (chronological)
    int m = 0;
    int order[HIST_MAX + 1] = {0};
    double ys[HIST_MAX] = {0}, sy[HIST_MAX][HIST_MAX] = {{0}}, b[HIST_MAX] = {0};

    // logs / stats
    std::vector<double> log_alpha, log_E, log_g2;
    long long numLineSearch = 0;
    int energy_evals = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, evA = nullptr;
    // DOTMI_FLAG_TIME_PHASES: boundaries of the phases of one line-search trial (host loop: every trial ends in a
    // stream synchronisation, after which the brackets recorded since the last one are read and the events reused)
    bool timePhases = false;
    hipEvent_t evP[8] = {nullptr};
    int evPn = 0;              // boundaries recorded since the last synchronisation
    int evPslot[8] = {0};      // ms_phase slot of the interval that ENDS at boundary k (k >= 1)
    double phaseMs[DOTMI_T_COUNT] = {0};
    // the level launches of the tile factorisation are a fixed sequence on fixed pointers: captured once into a hipGraph and
    // replayed every step (removes the host launch cost between them)
    hipGraphExec_t factorGraph = nullptr;
    int graphState = 0;  // 0 = not tried, 1 = ready, -1 = capture unavailable -> direct launches
    std::vector<hipEvent_t> evPre;  // DOTMI_FLAG_TIME_BACKSOLVE: (start, stop) pairs around each back-solve
    int evUsed = 0;
    // the same flag samples the collectives of the sharded path (every timeStride-th one): (start, stop) pairs + payloads
    std::vector<hipEvent_t> evAr;
    std::vector<size_t> arTimedBytes;
    int evArUsed = 0;
    long long arCount = 0, arCallsStep = 0;
    double arBytesStep = 0;
    int64_t precond_bytes = 0;
    double flopCount = 0, factorFlops = 0;  // running counter of the recursion; FP64 flop of one factorisation
};

#define HIPCHECK(h, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return DOTMI_E_DEVICE;                                                                \
        }                                                                                         \
    } while (0)
#define NCCLCHECK(h, call)                                                                        \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess) {                                                                  \
            (h)->err = std::string(#call) + ": " + ncclGetErrorString(r_);                        \
            return DOTMI_E_DEVICE;                                                                \
        }                                                                                         \
    } while (0)

namespace dotmi {

template <class T>
int dalloc(dotmi_handle *h, T **ptr, size_t count)
{
    void *p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIPCHECK(h, hipMalloc(&p, bytes));
    h->allocs.push_back(p);
    *ptr = (T *)p;
    return 0;
}

template <class T>
int upload(dotmi_handle *h, T **ptr, const std::vector<T> &v)
{
    int rc = dalloc(h, ptr, v.size());
    if (rc) return rc;
    if (!v.empty()) HIPCHECK(h, hipMemcpy(*ptr, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// ---- shared between the translation units --------------------------------------------------------------------------
// dotmi_create.hip
int build_device_mesh(dotmi_handle *h);
LbfgsArgs lbfgs_args(const dotmi_handle *h);
int free_slot(const dotmi_handle *h);
// dotmi_refresh.hip
int refactor_issue(dotmi_handle *h, const double *x);
int refactor_finish(dotmi_handle *h, double *ms_hess, double *ms_fact);
int refactor(dotmi_handle *h, const double *x, double *ms_hess, double *ms_fact);
int resolve_refresh(dotmi_handle *h, double *ms_hess = nullptr, double *ms_fact = nullptr);
int enter_with_factors(dotmi_handle *h);
// dotmi_collectives.hip
int allreduce_sum(dotmi_handle *h, double *dev, size_t n);
int adopt_rank0(dotmi_handle *h, double *vals, int n);
int exchange_iface(dotmi_handle *h, double *vec, double *tailp, int ntail);
int exchange_gradient_packed(dotmi_handle *h, int n, int nbE, const double *partials, int ncols);
int exchange_solve_packed(dotmi_handle *h);
// dotmi_loop.hip
int apply_precond(dotmi_handle *h, const double *q, double *z, const LbfgsArgs &L);
int enqueue_loop_slot(dotmi_handle *h);

}  // namespace dotmi

