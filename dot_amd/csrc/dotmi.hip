// dotmi.hip -- host side of libdotmi.so: setup, the L-BFGS-H time-step loop, and the C ABI.
//
// Control flow mirrors (paths relative to /root/reference/src)
//   dotmi_create       Optimizer ctor (TimeStepper/Optimizer.cpp:52-196), ADMMDDTimeStepper ctor
//                      (ADMMDDTimeStepper.cpp:44-443: partition -> local maps), DOTTimeStepper ctor +
//                      precompute (DOTTimeStepper.cpp:38-178), Mesh::computeFeatures (Mesh.cpp:589-700)
//   dotmi_step         Optimizer::solve (Optimizer.cpp:327-368) -> DOTTimeStepper::fullyImplicit
//                      (DOTTimeStepper.cpp:273-346) -> solve_oneStep (:384-504) -> Optimizer::lineSearch
//                      (Optimizer.cpp:752-881) ; updateHessianAndFactor (DOTTimeStepper.cpp:349-380)
// The data path is entirely on the device; the host only sequences launches, evaluates the m x m
// scalar recurrences of the two-loop recursion and takes the accept / halve / converged decisions
// from one small read-back per line-search trial.
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

using namespace dotmi;

namespace {

std::string g_create_error;

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch())
        .count();
}

__global__ void reduce_rows_kernel(const double *__restrict__ partials, int nblocks, int stride, int K,
                                   double s0, double s1, int combine, double *__restrict__ out)
{
    // single wave; out[j] = sum_b partials[b*stride+j]; combine: out[0] = s0*sum0 + s1*sum1
    const int lane = threadIdx.x;
    double first = 0.0;
    for (int j = 0; j < K; ++j) {
        double acc = 0.0;
        for (int b = lane; b < nblocks; b += 64) acc += partials[(size_t)b * stride + j];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) {
            if (combine) {
                if (j == 0) first = s0 * acc;
                else if (j == 1) out[0] = first + s1 * acc;
            } else {
                out[j] = acc;
            }
        }
    }
}

template <class T>
struct DBuf {
    T *p = nullptr;
    size_t n = 0;
};

}  // namespace

// ---- tuning / ablation switches -----------------------------------------------------------------------------------
// Read ONCE per dotmi_create from the environment; every one is optional and the defaults are the product path.  They
// exist for the A/B measurements logged under profiles/ (tools/ab.sh) and for tests that force a code path; none changes
// results beyond rounding.  Listed in include/dotmi.h ("Environment") and DESIGN.md section 10.
struct Tuning {
    int ndLevels = -1;        // DOTMI_ND_LEVELS      depth of the nested dissection (-1: nd_default_levels)
    int ndMin = ND_MIN_SPLIT; // DOTMI_ND_MIN         smallest region (scalars) that is still split
    int tileRows = 0;         // DOTMI_TILE_ROWS      rows per back-solve tile (0: 64, or 32 for few subdomains)
    int tileRowsLong = 0;     // DOTMI_TILE_ROWS_LONG rows per back-solve tile when the rows have more than 1536 columns (0: as the
                              //                      other rows, or ~256 KB tiles where few subdomains leave the launch bound by
                              //                      its longest tile)
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
    int tileEagerChunk = 0;   // DOTMI_TILE_EAGER_CHUNK early products per eager tile task
    bool fuseDir = true;      // DOTMI_FUSE_DIR=0     (early order) build_p and spmv_dots as two launches instead of one on cached H s_j
    bool fuseStep = true;     // DOTMI_FUSE_STEP=0    (early order) step_forward as a launch of its own instead of inside the element pass
    bool earlyAbort = true;   // DOTMI_EARLY_ABORT=0  (ablation) speculative back-solves run to their end even when the trial is rejected
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
        t.ndLevels = geti("DOTMI_ND_LEVELS", -1);
        if (t.ndLevels < -1) t.ndLevels = 0;
        t.ndMin = std::max(128, geti("DOTMI_ND_MIN", ND_MIN_SPLIT));
        if (const char *ev = getenv("DOTMI_TILE_ROWS")) t.tileRows = std::min(64, std::max(8, atoi(ev) / 8 * 8));
        t.tileRowsLong = geti("DOTMI_TILE_ROWS_LONG", 0);
        if (t.tileRowsLong > 0) t.tileRowsLong = std::min(64, std::max(8, t.tileRowsLong / 8 * 8));
        t.splitMerge = geti("DOTMI_SPLIT_MERGE", -1);
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
        t.tileEagerChunk = std::max(0, geti("DOTMI_TILE_EAGER_CHUNK", 0));
        t.earlyBs = geti("DOTMI_EARLY_BACKSOLVE", 2) != 0 ? 2 : 0;   // (1, round 3's per-step rule, now means "on")
        t.earlyAbort = geti("DOTMI_EARLY_ABORT", 1) != 0;
        t.earlyHold = geti("DOTMI_EARLY_HOLD", 1) != 0;
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
    TileTask *ttasks = nullptr;
    TileProd *tprods = nullptr;
    double **tclear = nullptr;
    int *tclearLd = nullptr;
    std::vector<long long> rtOff;   // host copy of the RowTile table (dotmi_part_matrix)
    std::vector<int> rtLd, rtC0;
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

namespace {

int allreduce_sum(dotmi_handle *h, double *dev, size_t n);
int adopt_rank0(dotmi_handle *h, double *vals, int n);

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
    const int ndMin = h->tune.ndMin;
    std::vector<std::vector<std::vector<int>>> region;  // [node][owned part] -> vertices of the leaf / separator
    {
        std::vector<std::vector<int>> sets(P.nParts);
        for (int ls = 0; ls < P.nParts; ++ls) sets[ls] = h->partVerts[h->p0 + ls];
        if (ndLevels < 0) {   // from the sizes of ALL subdomains of the mesh: the same tree on every rank
            int nsAll = 0;
            for (const auto &pv : h->partVerts) nsAll = std::max(nsAll, 3 * (int)pv.size());
            ndLevels = nd_default_levels(nsAll, (int)h->partVerts.size());
        }
        nd_plan(sets, nV, adj_ptr, adj_idx, h->Xrest.data(), ndLevels, ndMin, h->nd, region);
    }
    P.nmax = h->nd[0].size;
    h->tileMode = P.nParts > 0;   // (a rank without subdomains plans nothing)
    // per part: padded position of every local vertex, tiles of the back-solve, structural non-zeros
    h->partPos.assign(P.nParts, {});
    std::vector<int> dofmap((size_t)P.nParts * P.nmax, -1);
    // rows per back-solve tile: 64, or 32 when 64-row tiles would not give every CU two workgroups (few subdomains:
    // the launch is then bound by the pass chain of a workgroup, which halves)
    const bool fewTiles = (long long)P.nParts * P.nmax / 64 < 2 * 256;
    int tileRows = fewTiles ? 32 : 64;
    if (h->tune.tileRows > 0) tileRows = h->tune.tileRows;
    std::vector<int4> tiles;
    std::vector<std::vector<int2>> ranges(P.nParts);
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
        int b = 0;
        for (size_t nd = 0; nd < h->nd.size(); ++nd) {
            const NdNode &N = h->nd[nd];
            const int used = 3 * (int)region[nd][ls].size();
            const int ro = nd_region_first_row(N, used);  // first live row of the region
            // the rows of a region start at their node's first column (a leaf's padding sits in front of its
            // live rows and is skipped; 16-column granularity keeps the 128-byte lines whole)
            const int cb = N.a < 0 ? (ro & ~15) : N.off;
            // a tile stays inside one 64-row block of the factor storage (RowTile): the first tile of a region ends
            // at the next multiple of 64
            // rows of more than 1536 columns (the separators of the upper tree levels) can take fewer rows per tile
            // (DOTMI_TILE_ROWS_LONG).  Where every CU has its two workgroups anyway (bar17K: 1116 tiles) that buys nothing
            // (profiles/r03_factor_tiles.txt section E); with few subdomains the launch lasts as long as its longest tile
            // (bunny5K / 8: a 32-row tile of the root separator is 512 KB at ~30 GB/s per workgroup), so those rows get
            // tiles of ~256 KB: 16 rows at 2000 columns, 8 at 3000 (round 4: bunny5K 23.0 -> 16.8 us, horse7K 46.5 -> 31.8)
            const int len = ro + used - cb;
            int trows = tileRows;
            if (len > 1536) {
                if (h->tune.tileRowsLong > 0) trows = std::min(tileRows, h->tune.tileRowsLong);
                else if (fewTiles) trows = std::min(tileRows, std::max(8, (32768 / len) / 8 * 8));
            }
            for (int r0 = ro, rows = 0; r0 < ro + used; r0 += rows) {
                rows = std::min(std::min(trows, ro + used - r0), 64 - (r0 & 63));
                tiles.push_back(make_int4(ls, r0, b | (rows << 16), cb));
                ranges[ls].push_back(make_int2(cb, r0 + rows));
                ++b;
            }
            for (int r = ro; r < ro + used; ++r) nnzX += usedBefore[r + 1] - usedBefore[cb];
        }
    }
    // every structural non-zero of the inverse factors is streamed once per back-solve
    h->precond_bytes = nnzX * 8;
    P.nbmax = 1;
    for (auto &r : ranges) P.nbmax = std::max(P.nbmax, (int)r.size());
    std::vector<int2> trange((size_t)std::max(P.nParts, 1) * P.nbmax, make_int2(0, 0));
    for (int ls = 0; ls < P.nParts; ++ls) std::copy(ranges[ls].begin(), ranges[ls].end(), trange.begin() + (size_t)ls * P.nbmax);
    // the same tiles grouped by part (GSDD solves one subdomain at a time): register-kernel tiles, and the long-row tiles
    // with their (tile, column chunk) work items
    auto tile_len = [](const int4 &t) { return t.y + (t.z >> 16) - t.w; };
    std::vector<int4> tilesByPart, ltilesByPart;
    std::vector<int2> lworkByPart;
    h->partTilePtr.assign(P.nParts + 1, 0);
    h->partLworkPtr.assign(P.nParts + 1, 0);
    for (const int4 &t : tiles) {   // generated part after part
        if (tile_len(t) > BS_LONG) {
            const int nch = (((tile_len(t) + 15) & ~15) + BS_LONG - 1) / BS_LONG;
            for (int c = 0; c < nch; ++c) lworkByPart.push_back(make_int2((int)ltilesByPart.size(), c));
            ltilesByPart.push_back(t);
            h->partLworkPtr[t.x + 1] += nch;
        } else {
            tilesByPart.push_back(t);
            h->partTilePtr[t.x + 1]++;
        }
    }
    for (int ls = 0; ls < P.nParts; ++ls) {
        h->partTilePtr[ls + 1] += h->partTilePtr[ls];
        h->partLworkPtr[ls + 1] += h->partLworkPtr[ls];
    }
    // heavy tiles first: work ~ rows * row length
    auto tile_work = [](const int4 &t) { return (long long)(t.z >> 16) * (t.y + 64 - t.w); };
    std::stable_sort(tiles.begin(), tiles.end(), [&](const int4 &a, const int4 &b) { return tile_work(a) > tile_work(b); });
    // rows longer than the register tile of the single-pass kernel go through the two-phase kernel, cut into
    // column chunks of BS_LONG
    std::vector<int4> ltiles;
    std::vector<int2> lwork;
    {
        std::vector<int4> shortTiles;
        P.maxTileLen = 0;
        P.maxChunks = 1;
        for (const int4 &t : tiles) {
            const int len = t.y + (t.z >> 16) - t.w;
            if (len > BS_LONG) {
                const int nch = (((len + 15) & ~15) + BS_LONG - 1) / BS_LONG;
                for (int c = 0; c < nch; ++c) lwork.push_back(make_int2((int)ltiles.size(), c));
                P.maxChunks = std::max(P.maxChunks, nch);
                ltiles.push_back(t);
            } else {
                P.maxTileLen = std::max(P.maxTileLen, len);
                shortTiles.push_back(t);
            }
        }
        tiles.swap(shortTiles);
    }
    // tiles whose rows need the 512-thread variant (more than 2560 columns) first: when both kinds exist they are
    // launched separately, so that the short ones run on the 256-thread kernel (two workgroups per CU instead of one)
    std::stable_partition(tiles.begin(), tiles.end(), [](const int4 &t) { return t.y + (t.z >> 16) - t.w > 2560; });
    P.ntilesWide = 0;
    for (const int4 &t : tiles) P.ntilesWide += (t.y + (t.z >> 16) - t.w > 2560);
    P.ntiles = (int)tiles.size();
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
    h->rtOff = rtOff;
    h->rtLd = rtLd;
    h->rtC0 = rtC0;
    h->wTotal = wTotal;
    // offset in W of (memory row r, column c) of owned subdomain ls, or -1 when that place is not stored
    auto waddr = [&](int ls, int r, int c) -> long long {
        const RowTile &R = rtab[(size_t)ls * ntl + (r >> 6)];
        if (R.off < 0 || c < R.c0 || c >= R.c0 + R.ld) return -1;
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
    if (int rc = upload(h, &P.vp_ptr, vp_ptr)) return rc;
    if (int rc = upload(h, &P.vp_off, vp_off)) return rc;
    {
        // merge straight from the tile partials (merge_tiles_kernel): per global scalar dof the ppart entries that make
        // up its value -- subdomain after subdomain (vp order), inside a subdomain the tiles that hold the column in
        // tile order; the first entry of a subdomain is stored complemented.  Same sums, same order as
        // reduce_partial_p + merge.
        P.mt_ptr = nullptr;
        P.mt_ent = nullptr;
        const long long ppartN = (long long)P.nParts * P.nbmax * P.nmax;
        // Big meshes: the walk over a dof's ~20 tile partials is a walk over scattered 8-byte words and 4-byte list entries
        // (1 M tets: 75 us per iteration at 0.26 of the HBM peak); the two-launch form reads the partials coalesced in the
        // subdomains' own order and gathers one 24-byte triple per (vertex, subdomain).  Small meshes keep the one launch.
        P.splitMerge = h->tune.splitMerge >= 0 ? (h->tune.splitMerge != 0) : (3ll * nV >= 400000 || ppartN >= (1ll << 31));
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
            }
            if (h->tune.fuseLog)
                fprintf(stderr, "dotmi: merge: %.1f tile partials per dof -> %s\n", (double)count / std::max(1, 3 * nV),
                        P.splitMerge ? "sum per subdomain, then gather (split)" : "one walk over the list");
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
        const int eagerChunk = h->tune.tileEagerChunk > 0 ? h->tune.tileEagerChunk : (tiny ? 2 : P.nParts <= 64 ? 4 : 8);
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
                                     eagerMin, eagerChunk, 0, true, eagerMinRmul);
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
        // the diagonal tasks' per-lane bottom steps (kernels.hip, block_chol_inv<N, FAST>): every layout (DOTMI_FAST_DIAG=0: the
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

// issue the tile factorisation of every owned subdomain on h->st (tile_factor.hpp): one dataflow launch, or one launch per level
int issue_factor(dotmi_handle *h)
{
    if (h->tileFlow) {
        launch_tile_flow(h->ttasks, h->nTtasks, h->tprods, h->tdepPtr, h->tdepIdx, h->tdone, h->tnext, ++h->tileEpoch, h->info_dev,
                         h->tileFlowWg, h->st, (double)h->tune.tileFlowWaitMs, h->fastDiag);
        h->flopCount = h->tileFlops;
        return 0;
    }
    {
        // one launch per level of the static tile schedule; a launch boundary is the only synchronisation
        for (size_t l = 0; l + 1 < h->tlevelStart.size(); ++l) {
            const int n = h->tlevelStart[l + 1] - h->tlevelStart[l];
            if (!h->tileSplit) {
                launch_tile_level(h->ttasks + h->tlevelStart[l], n, h->tprods, h->info_dev, h->st, h->fastDiag);
                continue;
            }
            // the level's diagonal-block tasks (77 KB of LDS, ~20 us each) on the side stream, its product / row / inverse
            // tasks (half tiles, four workgroups per CU) on the main one, side by side; the next level waits for both
            const int nd = h->tlevelDiag[l], ng = n - nd;
            const TileTask *t0 = h->ttasks + h->tlevelStart[l];
            if (nd > 0 && ng > 0) {
                HIPCHECK(h, hipEventRecord(h->tFork[l], h->st));
                HIPCHECK(h, hipStreamWaitEvent(h->stDiag, h->tFork[l], 0));
                launch_tile_level(t0, nd, h->tprods, h->info_dev, h->stDiag, h->fastDiag);
                launch_tile_gemm(t0 + nd, ng, h->tprods, h->st);
                HIPCHECK(h, hipEventRecord(h->tJoin[l], h->stDiag));
                HIPCHECK(h, hipStreamWaitEvent(h->st, h->tJoin[l], 0));
            } else if (nd > 0) {
                launch_tile_level(t0, nd, h->tprods, h->info_dev, h->st, h->fastDiag);
            } else {
                launch_tile_gemm(t0, ng, h->tprods, h->st);
            }
        }
    }
    h->flopCount = h->tileFlops;
    return 0;
}

int run_factor(dotmi_handle *h)
{
    if (h->graphState == 0) {
        h->graphState = -1;
        if (h->tune.factorGraph && !h->tileFlow) {   // (the dataflow launch carries its epoch as an argument: not replayed)
            // one pass outside of capture (lazy code-object loads), then capture the same sequence
            h->flopCount = 0;
            if (int rc = issue_factor(h)) return rc;
            h->factorFlops = h->flopCount;
            HIPCHECK(h, hipStreamSynchronize(h->st));
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(h->st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int rc = issue_factor(h);
                const hipError_t e = hipStreamEndCapture(h->st, &graph);
                if (rc == 0 && e == hipSuccess && graph &&
                    hipGraphInstantiate(&h->factorGraph, graph, nullptr, nullptr, 0) == hipSuccess)
                    h->graphState = 1;
                if (graph) hipGraphDestroy(graph);
            }
            (void)hipGetLastError();
            if (h->graphState == 1) return 0;  // the warm-up pass already factored this H
            // capture failed: the warm-up pass overwrote W in place, which is what this call wanted anyway
            return 0;
        }
    }
    if (h->graphState == 1) {
        HIPCHECK(h, hipGraphLaunch(h->factorGraph, h->st));
        return 0;
    }
    h->flopCount = 0;
    const int rc = issue_factor(h);
    h->factorFlops = h->flopCount;
    return rc;
}

int refactor_issue(dotmi_handle *h, const double *x)
{
    HIPCHECK(h, hipEventRecord(h->ev0, h->st));
    if (h->shardHess) {
        launch_elem_hessians(h->M, h->mat, h->dtSq, x, h->He, h->st, h->hessElems, h->nHessElems);
        launch_assemble(h->M, h->He, h->Hval, h->st, h->hessBlk, h->nHessBlk, h->hessBlkPtr, h->hessBlkEnt);
        // owner exchange: this rank's own elements' part of the same rows (+ the mass of the vertices it owns): the
        // operator behind alpha_0's p.Hp -- the parts of all ranks add up to H, and no row needs a vertex the rank does not hold
        if (h->owner)
            launch_assemble(h->M, h->He, h->HvalOwn, h->st, h->hessBlk, h->nHessBlk, h->ownBlkPtr, h->ownBlkEnt, h->massOwn);
    } else {
        launch_elem_hessians(h->M, h->mat, h->dtSq, x, h->He, h->st);
        launch_assemble(h->M, h->He, h->Hval, h->st);
    }
    HIPCHECK(h, hipEventRecord(h->evA, h->st));
    // only the blocks the factorisation leaves non-zero are cleared before the refill: the leaf squares and
    // the separator panels; the (A,C) blocks and the cleared mirror panels stay zero for the handle's life
    // tile factorisation: H goes into the WORK buffer (tile_factor.hpp); the factor buffer W was zeroed once and only ever
    // receives tiles of Q
    DevParts Pf = h->P;
    if (h->tileMode) Pf.W = h->W2;
    if (h->wDirty) {
        launch_clear_tiles(h->tclear, h->tclearLd, h->nTclear, h->st);
    } else if (h->P.nParts > 0) {
        HIPCHECK(h, hipMemsetAsync(h->P.W, 0, h->wTotal * sizeof(double), h->st));
        if (h->tileMode) HIPCHECK(h, hipMemsetAsync(h->W2, 0, h->wTotal * sizeof(double), h->st));
        h->wDirty = true;
    }
    launch_dense_fill(Pf, h->Hval, h->st);
    HIPCHECK(h, hipEventRecord(h->ev1, h->st));
    if (h->P.nParts > 0) {
        HIPCHECK(h, hipMemsetAsync(h->info_dev, 0, sizeof(int) * h->P.nParts, h->st));
        if (int rc = run_factor(h)) return rc;
        HIPCHECK(h, hipMemcpyAsync(h->h_info, h->info_dev, sizeof(int) * h->P.nParts, hipMemcpyDeviceToHost, h->st));
    }
    HIPCHECK(h, hipEventRecord(h->ev2, h->st));
    return 0;
}

// after the stream has been synchronised: SPD check of every owned subdomain and the two timings
int refactor_finish(dotmi_handle *h, double *ms_hess, double *ms_fact)
{
    int bad = -1;
#ifdef DOTMI_TEST_HOOKS
    if (h->testFailRefresh > 0 && ++h->testRefreshCount == h->testFailRefresh && h->P.nParts > 0) h->h_info[0] = 7;
#endif
    for (int i = 0; i < h->P.nParts && bad < 0; ++i)
        if (h->h_info[i] != 0) bad = i;
    if (h->world > 1) {
        // all ranks fail together: a rank that returned alone would leave the others blocked in the next collective
        double f = bad >= 0 ? 1.0 : 0.0;
        HIPCHECK(h, hipMemcpyAsync(h->ctrlDev, &f, sizeof(double), hipMemcpyHostToDevice, h->st));
        if (int rc = allreduce_sum(h, h->ctrlDev, 1)) return rc;
        HIPCHECK(h, hipMemcpyAsync(&f, h->ctrlDev, sizeof(double), hipMemcpyDeviceToHost, h->st));
        HIPCHECK(h, hipStreamSynchronize(h->st));
        if (f > 0.0 && bad < 0) {
            h->err = "a subdomain Hessian on another rank is not positive definite";
            h->poisoned = true;
            return DOTMI_E_NOTSPD;
        }
    }
    // a dataflow wait that timed out anywhere is a DEVICE failure, whatever pivot report stands in front of it
    int stuck = -1;
    for (int i = 0; i < h->P.nParts && stuck < 0; ++i)
        if (h->h_info[i] >= (1 << 30)) stuck = i;
    if (stuck >= 0) {
        h->err = "the tile factorisation's dataflow scheduler waited for a task that never finished (subdomain " +
                 std::to_string(h->p0 + stuck) + ")";
        h->poisoned = true;
        return DOTMI_E_DEVICE;
    }
    if (bad >= 0) {
        h->err = "subdomain " + std::to_string(h->p0 + bad) + " Hessian not positive definite (pivot " +
                 std::to_string(h->h_info[bad]) + ")";
        h->poisoned = true;  // every later step / back-solve fails until a factorisation succeeds
        return DOTMI_E_NOTSPD;
    }
    h->poisoned = false;
    float a = 0, b = 0, c = 0;
    hipEventElapsedTime(&a, h->ev0, h->ev1);
    hipEventElapsedTime(&b, h->ev1, h->ev2);
    hipEventElapsedTime(&c, h->ev0, h->evA);
    if (ms_hess) *ms_hess += a;
    if (ms_fact) *ms_fact += b;
    h->phaseMs[DOTMI_T_MATRIX_COMPUTATION] += c;       // element Hessians + global assembly
    h->phaseMs[DOTMI_T_MATRIX_ASSEMBLY] += a - c;      // clear + dense sub-matrix fill
    h->phaseMs[DOTMI_T_NUMERICAL_FACTORIZATION] += b;
    HIPCHECK(h, hipGetLastError());
    return 0;
}

// element Hessians -> global H -> dense sub-matrices -> inverse Cholesky factors
// (DOTTimeStepper::updateHessianAndFactor, DOTTimeStepper.cpp:349-380)
int refactor(dotmi_handle *h, const double *x, double *ms_hess, double *ms_fact)
{
    if (int rc = refactor_issue(h, x)) return rc;
    HIPCHECK(h, hipStreamSynchronize(h->st));
    return refactor_finish(h, ms_hess, ms_fact);
}

// DOTMI_FLAG_ASYNC_REFRESH: wait for the refresh the last step left running, take its verdict and its device times
// (into *ms_hess / *ms_fact, or carried to the next step's statistics)
int resolve_refresh(dotmi_handle *h, double *ms_hess = nullptr, double *ms_fact = nullptr)
{
    if (!h->refreshPending) return 0;
    h->refreshPending = false;
    HIPCHECK(h, hipStreamSynchronize(h->st));
    double a = 0, b = 0;
    const int rc = refactor_finish(h, &a, &b);
    if (ms_hess) *ms_hess += a;
    else h->carryHess += a;
    if (ms_fact) *ms_fact += b;
    else h->carryFact += b;
    return rc;
}

// Every entry point that reads the factors starts here: the refresh a step left running (DOTMI_FLAG_ASYNC_REFRESH) is waited
// for and judged FIRST, then the handle's verdict is tested -- so a non-SPD subdomain found by an asynchronous refresh stops
// the next call exactly like one found by the synchronous path
int enter_with_factors(dotmi_handle *h)
{
    const int rc = resolve_refresh(h);
    if (rc == DOTMI_E_DEVICE) return rc;
    if (h->poisoned) {
        if (rc != DOTMI_E_NOTSPD) h->err = "the subdomain factors are invalid (the last factorisation failed): " + h->err;
        return DOTMI_E_NOTSPD;
    }
    return 0;
}

// sum over the ranks of n doubles at `dev`, in place, ordered on the handle's stream: RCCL, or the host hook
int allreduce_sum(dotmi_handle *h, double *dev, size_t n)
{
    h->arCallsStep++;
    h->arBytesStep += 8.0 * (double)n;
    if (h->comm) {
        const bool timed = !h->evAr.empty() && h->evArUsed + 2 <= (int)h->evAr.size() && (h->arCount++ % h->timeStride) == 0;
        if (timed) HIPCHECK(h, hipEventRecord(h->evAr[h->evArUsed], h->st));
        NCCLCHECK(h, ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, h->comm, h->st));
        if (timed) {
            HIPCHECK(h, hipEventRecord(h->evAr[h->evArUsed + 1], h->st));
            h->arTimedBytes.push_back(8 * n);
            h->evArUsed += 2;
        }
        return 0;
    }
    if (!h->arCb) return 0;   // single rank without a communicator (cannot happen on the sharded path)
    if (n > h->arCap) {
        if (h->arStage) hipHostFree(h->arStage);
        h->arStage = nullptr;
        HIPCHECK(h, hipHostMalloc((void **)&h->arStage, sizeof(double) * n));
        h->arCap = n;
    }
    HIPCHECK(h, hipMemcpyAsync(h->arStage, dev, sizeof(double) * n, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    h->arCb(h->arCtx, h->arStage, (int64_t)n);
    HIPCHECK(h, hipMemcpyAsync(dev, h->arStage, sizeof(double) * n, hipMemcpyHostToDevice, h->st));
    return 0;
}

// Every rank takes its accept / halve / converged decisions from RANK 0's control scalars (the ranks
// compute them redundantly on replicated data, but a single differing bit would make them branch apart and dead-lock
// in the next collective).  vals: host array, replaced by rank 0's on every rank.  One small collective per trial.
int adopt_rank0(dotmi_handle *h, double *vals, int n)
{
    if (h->world <= 1) return 0;
    if (h->rank != 0)
        for (int i = 0; i < n; ++i) vals[i] = 0.0;
    HIPCHECK(h, hipMemcpyAsync(h->ctrlDev, vals, sizeof(double) * n, hipMemcpyHostToDevice, h->st));
    if (int rc = allreduce_sum(h, h->ctrlDev, n)) return rc;   // x + 0 + ... + 0 is exact: a broadcast
    HIPCHECK(h, hipMemcpyAsync(vals, h->ctrlDev, sizeof(double) * n, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    return 0;
}

// DOTMI_FLAG_TIME_PHASES: a phase boundary on the stream; the interval that ends here is booked under `slot`
// (slot < 0: the boundary only starts an interval)
inline void phase_mark(dotmi_handle *h, int slot)
{
    if (!h->timePhases || h->evPn >= 8) return;
    hipEventRecord(h->evP[h->evPn], h->st);
    h->evPslot[h->evPn] = slot;
    h->evPn++;
}
// after a stream synchronisation: read the recorded brackets
inline void phase_collect(dotmi_handle *h)
{
    for (int k = 1; k < h->evPn; ++k) {
        float ms = 0;
        if (h->evPslot[k] >= 0 && hipEventElapsedTime(&ms, h->evP[k - 1], h->evP[k]) == hipSuccess)
            h->phaseMs[h->evPslot[k]] += ms;
    }
    h->evPn = 0;
}

// p = D^-1 sum_s R_s^T W_s R_s q   (DOTTimeStepper.cpp:406-450); leaves y_i.z partials in partC
int apply_precond(dotmi_handle *h, const double *q, double *z, const LbfgsArgs &L)
{
    const bool timed = (h->flags & DOTMI_FLAG_TIME_BACKSOLVE) && h->evUsed + 2 <= (int)h->evPre.size() &&
                       (h->timeCount++ % h->timeStride) == 0;
    launch_gemv(h->P, q, h->st, nullptr, timed ? h->evPre[h->evUsed] : nullptr, timed ? h->evPre[h->evUsed + 1] : nullptr);
    if (timed) h->evUsed += 2;
    if (!h->dist) {
        launch_merge(h->M, h->P, L, z, h->partC, 1 | 2, h->st);
    } else {
        launch_merge(h->M, h->P, L, z, h->partC, 0, h->st);
        if (int rc = allreduce_sum(h, z, h->n)) return rc;
        launch_zfinish(h->nV, h->P.dup, L, z, h->partC, h->st);
    }
    return 0;
}

// energy + element gradients + vertex gather (+ pair) at `xeval`; results: *E, stats in h_partR
int trial(dotmi_handle *h, const double *xeval, double *gout, int make_pair, const LbfgsArgs &L, int slot,
          double *E, int evalSlot = DOTMI_T_LINESEARCH_EVAL, int gradSlot = DOTMI_T_UPDATE_HISTORY)
{
    int nb = 0;
    // single-GPU: the reduction partials go straight to pinned host memory (zero-copy), so one stream
    // synchronisation is the only host<->device interaction of a line-search trial
    double *partE = h->shardElems ? h->partE : h->h_partE;
    double *partR = h->shardElems ? h->partR : h->h_partR;
    launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, xeval, h->xt, h->v0, h->v1, 1, partE, &nb, h->st);
    h->nbE = nb;
    phase_mark(h, evalSlot);
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.x = xeval;
    a.xt = h->xt;
    a.g_old = h->g;
    a.p = h->p;
    a.alpha_dev = h->alpha_dev;
    a.g_new = gout;
    a.s_new = h->S[slot];
    a.y_new = h->Y[slot];
    a.iv0 = h->v0;
    a.iv1 = h->v1;
    a.stage = 0;
    if (!h->shardElems) {
        a.make_pair = make_pair;
        launch_vertex_gather(h->M, h->PT, a, L, partR, h->st);
    } else {
        a.make_pair = 0;
        launch_vertex_gather(h->M, h->PT, a, L, h->partR, h->st);
        // pack E_local behind the gradient and reduce both in one collective
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, h->st, h->partE, nb, 2, 2, h->dtSq, 1.0, 1,
                           gout + h->n);
        if (int rc = allreduce_sum(h, gout, (size_t)h->n + 1)) return rc;
        if (make_pair) launch_pair_stats(h->n, a, L, h->partR, h->st);
        else {
            // |g|^2 only
            const double *vecs[1] = {gout};
            launch_multidot(h->n, gout, vecs, 1, h->partR, h->st);
        }
        HIPCHECK(h, hipMemcpyAsync(h->h_partE, gout + h->n, sizeof(double), hipMemcpyDeviceToHost, h->st));
        HIPCHECK(h, hipMemcpyAsync(h->h_partR, h->partR, sizeof(double) * NB_RED * RED_K, hipMemcpyDeviceToHost,
                                   h->st));
    }
    phase_mark(h, gradSlot);
    HIPCHECK(h, hipStreamSynchronize(h->st));
    phase_collect(h);
    if (!h->shardElems) {
        const double se = chunked_sum(nb, [&](int b) { return h->h_partE[2 * b]; });
        const double si = chunked_sum(nb, [&](int b) { return h->h_partE[2 * b + 1]; });
        *E = h->dtSq * se + si;
    } else {
        *E = h->h_partE[0];
    }
    if (h->world > 1) {
        // one set of control scalars for all ranks: rank 0's (energy, step length, every column of the statistics)
        h->ctrl[0] = *E;
        h->ctrl[1] = h->h_alpha[0];
        for (int j = 0; j < RED_K; ++j)
            h->ctrl[2 + j] = chunked_sum(h->M_nbR(), [&](int b) { return h->h_partR[(size_t)b * RED_K + j]; });
        if (int rc = adopt_rank0(h, h->ctrl, RED_K + 2)) return rc;
        *E = h->ctrl[0];
        h->h_alpha[0] = h->ctrl[1];
    }
    h->energy_evals++;
    return 0;
}

void sum_stats(const dotmi_handle *h, int nvals, double *R)
{
    if (h->world > 1) {   // what trial() adopted from rank 0
        for (int j = 0; j < nvals; ++j) R[j] = h->ctrl[2 + j];
        return;
    }
    for (int j = 0; j < nvals; ++j) R[j] = chunked_sum(NB_RED, [&](int b) { return h->h_partR[(size_t)b * RED_K + j]; });
}

// owner exchange: the entries of the vertices held by more than one rank (and `ntail` scalars behind them) summed over the ranks
int exchange_iface(dotmi_handle *h, double *vec, double *tailp, int ntail)
{
    if (3 * h->nIface + ntail == 0) return 0;   // (no vertex is shared -- one rank --: the same on every rank, nothing to send)
    launch_pack_iface(h->nIface, h->ifaceIdx, vec, h->xpack, tailp, ntail, h->st);
    if (int rc = allreduce_sum(h, h->xpack, (size_t)3 * h->nIface + ntail)) return rc;
    launch_unpack_iface(h->nIface, h->ifaceIdx, h->xpack, h->heldMask, vec, tailp, ntail, h->st);
    return 0;
}

// owner exchange, packed form.  The gradient's packet: [3 nIface entries | E | ncols statistics]; `partials` holds the sums this
// rank took BEFORE the exchange (pair_stats with pre = 1, or the |g|^2 of the vertices only it holds at the start of a step:
// ncols = 1) and partE the element pass' energy partials -- both are summed by the pack's workgroup 0 straight into the tail.
// The summed statistics land in row 0 of partGR with the shared entries' squares added to |g|^2, E in gstage[n + 1].
int exchange_gradient_packed(dotmi_handle *h, int n, int nbE, const double *partials, int ncols)
{
    const int n3 = 3 * h->nIface;
    PackRed rE{h->partE, nbE, 2, 2, 1, n3, h->dtSq, 1.0};
    PackRed rS{partials, NB_RED, RED_K, ncols, 0, n3 + 1, 0.0, 0.0};
    launch_pack_iface(h->nIface, h->ifaceIdx, h->gstage, h->xpack, nullptr, 0, h->st, &rE, &rS);
    if (int rc = allreduce_sum(h, h->xpack, (size_t)n3 + 1 + ncols)) return rc;
    launch_unpack_iface(h->nIface, h->ifaceIdx, h->xpack, h->heldMask, h->gstage, h->gstage + n + 1, 1, h->st, h->partGR, ncols);
    return 0;
}
// The merged back-solve's packet: [3 nIface entries | HIST_MAX sums y_i . z] (merge_early with pre = 1 left this rank's share
// in partC); the sums land in row 0 of partGC.
int exchange_solve_packed(dotmi_handle *h)
{
    const int n3 = 3 * h->nIface;
    PackRed rC{h->partC, NB_RED, RED_K, HIST_MAX, 0, n3, 0.0, 0.0};
    launch_pack_iface(h->nIface, h->ifaceIdx, h->zstage, h->xpack, nullptr, 0, h->st, &rC);
    if (int rc = allreduce_sum(h, h->xpack, (size_t)n3 + HIST_MAX)) return rc;
    launch_unpack_iface(h->nIface, h->ifaceIdx, h->xpack, h->heldMask, h->zstage, h->partGC, HIST_MAX, h->st);
    return 0;
}

// One slot of the device-resident loop: the nine kernels of an L-BFGS iteration (or, when the controller
// asked for a retry, only the three of a line-search trial -- the others return at once) and the controller.
// Early back-solve (one rank, h->earlyBs): the preconditioner M is fixed during a step and linear, so the solve for the
// next direction does not have to wait for the controller's verdict and for q.  The slot starts at build_p; after the
// trial's gradient is gathered the back-solve runs on -g_trial with the CONTROLLER AS ONE WORKGROUP OF ITS LAUNCH, and
// merge_early forms z = u - sum_j xi_j (M y_j) from the cached M y_j (the newest: u_old - u).  The controller's ~7 us
// and its launch boundary leave the critical path of every iteration; a rejected trial (line-search halving) and the
// last iteration of a step stream the factors once for nothing.  z differs from the q-based value by rounding only
// (tests/test_gpu_round3.py: same iteration counts, positions to 1e-9).
int enqueue_loop_slot_early(dotmi_handle *h)
{
    const int n = h->n;
    LbfgsArgs L0;
    memset(&L0, 0, sizeof(L0));
    const bool fuseDir = h->tune.fuseDir;
    const bool se = h->shardElems;   // sharded element pass: this rank's rows of H, its elements; three collectives per slot
    const bool ow = h->owner;        // owner exchange: only the entries of shared vertices travel, dots are owner-summed scalars
    if (ow) {
        launch_spmv_zp(h->M, h->HvalOwn, h->z, h->partGC, h->p, h->Hp, h->partS, h->st, h->ctl, 0, -1, h->heldMask, h->ownMask,
                       h->held());
    } else if (fuseDir) {   // build_p + spmv_dots in one launch, H p from the cached H s_j
        launch_spmv_zp(h->M, h->Hval, h->z, h->partC, h->p, h->Hp, h->partS, h->st, h->ctl, se ? h->v0 : 0, se ? h->v1 : -1);
    } else {
        launch_build_p(n, h->z, L0, h->partC, nullptr, h->p, h->st, h->ctl);
        launch_spmv_dots(h->M, h->Hval, h->p, h->g, nullptr, h->v0, h->v1, h->partS, h->st, h->ctl);
    }
    const double *spart = h->partS;
    if (se) {
        // this rank's rows of p.g and p.Hp -> two scalars -> summed over the ranks (row 0 of partG; rows >= 1 stay zero).
        // It cannot ride on the z all-reduce in front of it: p.Hp is quadratic in the reduced vector
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, h->st, h->partS, NB_RED, RED_K, 2, 0.0, 0.0, 0, h->partG);
        if (int rc = allreduce_sum(h, h->partG, 2)) return rc;
        spart = h->partG;
    }
    int nb = 0;
    // (not on meshes whose workgroups walk several patches: the prefetched operands leave no registers for it)
    // (sharded element pass: the fused form writes the trial point only on this rank's vertex slice -- its inertia loop --,
    // so the step stays a launch of its own there)
    // (owner exchange: the inertia loop runs over every vertex with the owner's share of the mass, so the fused form writes the
    // whole trial point there too -- x + alpha 0 off the held vertices)
    if (h->tune.fuseStep && (!se || ow)) {   // the step x_trial = x_cur + alpha p inside the element pass
        StepArgs sa{h->p, spart, h->alpha_dev, h->alphaMin};
        launch_elem_energy_grad(ow ? h->Mown : h->M, h->PT, h->mat, h->dtSq, h->x_trial, h->xt, ow ? 0 : h->v0,
                                ow ? h->nV : h->v1, 1, h->partE, &nb, h->st, h->ctl, &sa);
    } else {
        launch_step_forward(n, h->x, h->p, h->x_trial, spart, 0.0, 1, h->alphaMin, h->alpha_dev, nullptr, h->st, h->ctl, h->held());
        launch_elem_energy_grad(ow ? h->Mown : h->M, h->PT, h->mat, h->dtSq, h->x_trial, h->xt, ow ? 0 : h->v0, ow ? h->nV : h->v1,
                                1, h->partE, &nb, h->st, h->ctl);
    }
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.xt = h->xt;
    a.p = h->p;
    a.alpha_dev = h->alpha_dev;
    a.iv0 = h->v0;
    a.iv1 = h->v1;
    a.make_pair = 1;
    a.hp = (fuseDir || ow) ? h->Hp : nullptr;   // H s_new = alpha H p beside s_new
    // -g_trial goes straight into the padded right-hand sides, whatever the controller will say about the trial
    a.vp_ptr = h->P.vp_ptr;
    a.vp_off = h->P.vp_off;
    a.rpad = h->P.rpad;
    const double *ctlE = h->partE;
    const bool packed = ow;   // owner exchange: the statistics ride in the gradient's packet
    if (!se) {
        launch_vertex_gather(h->M, h->PT, a, L0, h->partR, h->st, h->ctl);
    } else if (packed) {
        // Owner exchange, the statistics in the gradient's packet.  The gradient is complete on the vertices only this rank
        // holds: the gather forms the pair, the right-hand sides and the statistics there as on one GPU; on the shared vertices
        // it leaves this rank's part of the gradient in the staging buffer and its share of the sums -- those are linear in the
        // gradient -- in the partials, which ride in the packet's tail (summed by the pack's workgroup 0, like E).  After the
        // exchange only the shared vertices are left (|g|^2 over them: from the summed packet, in the unpack)
        GatherArgs ag = a;
        ag.ownMask = h->ownMask;
        ag.vlist = h->heldList;
        ag.nlist = h->nHeld;
        ag.kind = h->vkind;
        ag.pre = 1;
        ag.gshare = h->gstage;
        launch_vertex_gather(h->M, h->PT, ag, L0, h->partR, h->st, h->ctl);
        if (int rc = exchange_gradient_packed(h, n, nb, h->partR, RED_K)) return rc;
        ag.vlist = h->sharedList;
        ag.nlist = h->nShared;
        ag.pre = 0;
        if (h->nShared > 0) launch_pair_stats(n, ag, L0, nullptr, h->st, h->gstage, h->ctl);
        ctlE = h->gstage + n;   // the controller reads the energy as one block (0, E)
        nb = 1;
    } else {
        // this rank's partial gradient and energy to the staging buffer [g (n) ; 0 ; E_local], one all-reduce, then the pair,
        // its statistics, -g into the right-hand sides and H s_new from the SUM (pair_stats)
        GatherArgs ag = a;
        ag.make_pair = 0;
        ag.stage = 1;
        ag.g_new = h->gstage;
        ag.hp = nullptr;
        ag.vp_ptr = ag.vp_off = nullptr;
        ag.rpad = nullptr;
        launch_vertex_gather(h->M, h->PT, ag, L0, h->partR, h->st, h->ctl);
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, h->st, h->partE, nb, 2, 2, h->dtSq, 1.0, 1,
                           h->gstage + n + 1);
        if (int rc = allreduce_sum(h, h->gstage, (size_t)n + 2)) return rc;
        launch_pair_stats(n, a, L0, h->partR, h->st, h->gstage, h->ctl);
        ctlE = h->gstage + n;   // the controller reads the energy as one block (0, E)
        nb = 1;
    }
    const double *ctlR = ow ? h->partGR : h->partR;
    const bool timed = (h->flags & DOTMI_FLAG_TIME_BACKSOLVE) && h->evUsed + 2 <= (int)h->evPre.size() &&
                       (h->timeCount++ % h->timeStride) == 0;
    h->slotTimed.push_back(timed ? h->evUsed : -1);
    CtlArgs ca{h->ctl, ctlE, ctlR, h->alpha_dev, h->h_flags, nb, 0};
    launch_gemv(h->P, nullptr, h->st, h->ctl, timed ? h->evPre[h->evUsed] : nullptr, timed ? h->evPre[h->evUsed + 1] : nullptr, &ca,
                h->tune.earlyAbort ? (int)h->slotTimed.size() /* the slot's epoch, 1-based */ : (1 << 30) /* never stopped */);
    if (timed) h->evUsed += 2;
    if (!h->dist) {
        launch_merge_early(h->M, h->P, h->z, h->partC, 0, h->st, h->ctl);
    } else {
        // sharded subdomains: this rank's part of the sum, the one collective of the iteration (issued in every slot,
        // whatever the controller decided: every rank enqueues the same sequence), then the division and the history terms
        // (the sum travels in a staging buffer: in a slot whose merge is gated off -- retry, past the end -- the collective
        // still runs, on stale scratch, and z is left alone)
        if (!ow)   // (owner exchange: merge_early merges the tiles itself)
            launch_merge(h->M, h->P, L0, h->zstage, h->partC, 0, h->st, h->ctl, h->held());
        if (!ow) {
            if (int rc = allreduce_sum(h, h->zstage, n)) return rc;
            launch_merge_early(h->M, h->P, h->z, h->partC, 0, h->st, h->ctl, h->zstage);
        } else {
            // zstage: this rank's subdomains' sum, zero on the vertices it does not hold; only the shared vertices' entries
            // are summed over the ranks.  z is then whole on the held vertices and zero elsewhere -- and so is everything
            // the loop forms from it.  The five y_i . z travel inside the packet: merge_early merges this rank's tiles itself -- z,
            // u_old, M y_new and the y_i . z on the vertices only this rank holds, its part of the sum (to zstage) and its share
            // of the y_i . z on the shared ones --, then the exchange, then the shared vertices
            launch_merge_early(h->M, h->P, h->z, h->partC, 0, h->st, h->ctl, nullptr, h->ownMask, h->held(), h->vkind, 1,
                               h->zstage);
            if (int rc = exchange_solve_packed(h)) return rc;
            if (h->nShared > 0)
                launch_merge_early(h->M, h->P, h->z, nullptr, 0, h->st, h->ctl, h->zstage, h->ownMask, h->shared());
        }
    }
    return 0;
}

int enqueue_loop_slot(dotmi_handle *h)
{
    if (h->earlyNow) return enqueue_loop_slot_early(h);
    const int n = h->n;
    LbfgsArgs L0;
    memset(&L0, 0, sizeof(L0));
    launch_build_qpad(h->P, h->g, L0, nullptr, h->st, h->ctl);   // q, straight into the padded right-hand sides
    // an event record costs ~6 us of stream time: sample, do not bracket every launch
    const bool timed = (h->flags & DOTMI_FLAG_TIME_BACKSOLVE) && h->evUsed + 2 <= (int)h->evPre.size() &&
                       (h->timeCount++ % h->timeStride) == 0;
    h->slotTimed.push_back(timed ? h->evUsed : -1);
    launch_gemv(h->P, nullptr, h->st, h->ctl, timed ? h->evPre[h->evUsed] : nullptr,
                timed ? h->evPre[h->evUsed + 1] : nullptr);
    if (timed) h->evUsed += 2;
    if (!h->dist) {
        launch_merge(h->M, h->P, L0, h->z, h->partC, 1 | 2, h->st, h->ctl);
    } else {
        // sharded subdomains: the one collective of an iteration, enqueued like a kernel.  It runs in every slot (also
        // in retries and past the end, where the kernels around it return at once), so every rank issues the same
        // sequence of collectives whatever the controller decides
        launch_merge(h->M, h->P, L0, h->z, h->partC, 0, h->st, h->ctl);
        if (int rc = allreduce_sum(h, h->z, n)) return rc;
        launch_zfinish(h->nV, h->P.dup, L0, h->z, h->partC, h->st, h->ctl);
    }
    launch_build_p(n, h->z, L0, h->partC, nullptr, h->p, h->st, h->ctl);
    launch_spmv_dots(h->M, h->Hval, h->p, h->g, nullptr, h->v0, h->v1, h->partS, h->st, h->ctl);
    const double *spart = h->partS;
    if (h->shardElems) {
        // this rank's rows of p.g and p.Hp -> two scalars -> summed over the ranks (row 0 of partG; rows >= 1 stay zero)
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, h->st, h->partS, NB_RED, RED_K, 2, 0.0, 0.0, 0,
                           h->partG);
        if (int rc = allreduce_sum(h, h->partG, 2)) return rc;
        spart = h->partG;
    }
    launch_step_forward(n, h->x, h->p, h->x_trial, spart, 0.0, 1, h->alphaMin, h->alpha_dev, nullptr, h->st, h->ctl);
    int nb = 0;
    launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x_trial, h->xt, h->v0, h->v1, 1, h->partE, &nb, h->st, h->ctl);
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.xt = h->xt;
    a.p = h->p;
    a.alpha_dev = h->alpha_dev;
    a.iv0 = h->v0;
    a.iv1 = h->v1;
    if (!h->shardElems) {
        a.make_pair = 1;
        launch_vertex_gather(h->M, h->PT, a, L0, h->partR, h->st, h->ctl);
        launch_loop_control(h->ctl, h->partE, nb, h->partR, h->alpha_dev, h->h_flags, h->st);
        return 0;
    }
    // sharded element pass: the partial gradient of this rank's elements and its energy go to the staging buffer
    // [g (n) ; 0 ; E_local], one all-reduce, then the pair + statistics from the summed gradient (which is copied to
    // the trial gradient, whose address only the controller knows).  The controller reads the energy as one "block"
    // (0, E): dtSq * 0 + E.
    a.make_pair = 0;
    a.stage = 1;
    a.g_new = h->gstage;
    launch_vertex_gather(h->M, h->PT, a, L0, h->partR, h->st, h->ctl);
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, h->st, h->partE, nb, 2, 2, h->dtSq, 1.0, 1,
                       h->gstage + n + 1);
    if (int rc = allreduce_sum(h, h->gstage, (size_t)n + 2)) return rc;
    a.make_pair = 1;
    launch_pair_stats(n, a, L0, h->partR, h->st, h->gstage, h->ctl);
    launch_loop_control(h->ctl, h->gstage + n, 1, h->partR, h->alpha_dev, h->h_flags, h->st);
    return 0;
}

// The L-BFGS loop of one time step with the control flow on the device (DevLoop).  The host only keeps
// the queue a few slots ahead of the controller's progress, which it reads from pinned memory.
int run_device_loop(dotmi_handle *h, double *lastE, double *g2, int *it, bool *failed, double *E0, double *g20)
{
    DevLoop &C = *h->h_ctl;
    memset(&C, 0, sizeof(C));
    // The early order (back-solve issued on the trial gradient, beside the controller) in every step of a handle that has its
    // buffers: it takes the controller (~9 us with its launch boundary) off every iteration and starts a back-solve for
    // nothing per rejected trial and once at the end of the step, which the controller tells to stop (DevLoop::abortEpoch).
    // (Round 3's per-step rule from the previous step's counts, DOTMI_EARLY_BACKSOLVE=1, is gone: with the stop and the held
    // launches the early order is at least as fast on every workload, and the owner exchange has no other order -- ADVICE r04.)
    h->earlyNow = h->earlyBs;
    C.iterCap = h->iterCap;
    C.hist = h->hist;
    C.tol = h->targetGRes;
    C.dtSq = h->dtSq;
    C.x_cur = h->x;
    C.x_trial = h->x_trial;
    C.g_cur = h->g;
    C.g_trial = h->g_trial;
    for (int s = 0; s <= h->hist; ++s) {
        C.S[s] = h->S[s];
        C.Y[s] = h->Y[s];
        C.MY[s] = h->MY[s];
        C.HS[s] = h->HS[s];
    }
    C.u_old = h->u_old;
    C.holdEnable = h->tune.earlyHold && h->tune.earlyAbort ? 1 : 0;
    // the forecast carries over from the last step (a function of the handle's own history)
    memcpy(C.predHist, h->predState, sizeof(int) * 2);
    memcpy(&C.predCtr[0][0], h->predState + 2, sizeof(int) * 8);
    C.log_alpha = h->dlog;
    C.log_E = h->dlog + h->logCap;
    C.log_g2 = h->dlog + 2 * (size_t)h->logCap;
    C.slot_kind = h->dkind;
    C.logCap = h->logCap;
    C.kindCap = h->kindCap;
    volatile int *flags = h->h_flags;
    const int n_ = h->n;
    flags[0] = 0;
    flags[1] = 0;
    // blind up to a little before last step's slot count, then two slots ahead of the posted progress
    const int AHEAD = 2;
    C.notifyFrom = std::max(0, std::min(h->prevSlots - 3, h->prevSlots * 3 / 4));  // a shorter step wastes few slots
    if (h->dist) C.notifyFrom = 1 << 30;   // deterministic batches: nobody reads the progress (a host store costs ~18 us)
    const int notifyFrom = C.notifyFrom;
    HIPCHECK(h, hipMemcpyAsync(h->ctl, h->h_ctl, sizeof(DevLoop), hipMemcpyHostToDevice, h->st));
    {
        // energy and gradient at the start of the step, reduced by the controller (no host round trip)
        int nb = 0;
        launch_elem_energy_grad(h->owner ? h->Mown : h->M, h->PT, h->mat, h->dtSq, h->x, h->xt, h->owner ? 0 : h->v0,
                                h->owner ? h->nV : h->v1, 1, h->partE, &nb, h->st);
        GatherArgs a;
        memset(&a, 0, sizeof(a));
        a.x = h->x;
        a.xt = h->xt;
        a.g_new = h->shardElems ? h->gstage : h->g;
        a.make_pair = 0;
        a.iv0 = h->v0;
        a.iv1 = h->v1;
        a.ownMask = h->owner ? h->ownMask : nullptr;
        a.vlist = h->owner ? h->heldList : nullptr;
        a.nlist = h->owner ? h->nHeld : 0;
        // owner exchange: the loop only touches the held vertices' entries; the trial buffer starts as a copy of the iterate
        // (the two swap roles on every accepted trial and must agree off the held set)
        if (h->owner) HIPCHECK(h, hipMemcpyAsync(h->x_trial, h->x, sizeof(double) * h->n, hipMemcpyDeviceToDevice, h->st));
        LbfgsArgs L0;
        memset(&L0, 0, sizeof(L0));
        if (h->earlyNow && !h->shardElems) {   // -g_0 straight into the padded right-hand sides (their padding entries stay zero)
            a.vp_ptr = h->P.vp_ptr;
            a.vp_off = h->P.vp_off;
            a.rpad = h->P.rpad;
        }
        launch_vertex_gather(h->M, h->PT, a, L0, h->partR, h->st);
        if (!h->shardElems && h->earlyNow) {
            // the first direction's solve, u = -M g_0 and z = u, with the start-of-step controller inside its launch
            CtlArgs ca{h->ctl, h->partE, h->partR, h->alpha_dev, h->h_flags, nb, 1};
            launch_gemv(h->P, nullptr, h->st, h->ctl, nullptr, nullptr, &ca, 1 << 30);
            if (!h->dist) {
                launch_merge_early(h->M, h->P, h->z, h->partC, 1, h->st, h->ctl);
            } else {
                launch_merge(h->M, h->P, L0, h->zstage, h->partC, 0, h->st, h->ctl);
                if (int rc = allreduce_sum(h, h->zstage, n_)) return rc;
                launch_merge_early(h->M, h->P, h->z, h->partC, 1, h->st, h->ctl, h->zstage);
            }
        } else if (!h->shardElems) {
            launch_loop_control(h->ctl, h->partE, nb, h->partR, h->alpha_dev, h->h_flags, h->st, 1);
        } else {
            if (!h->owner)
                hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, h->st, h->partE, nb, 2, 2, h->dtSq, 1.0, 1,
                                   h->gstage + n_ + 1);
            const double *ctlR = h->partR;
            if (!h->owner) {
                if (int rc = allreduce_sum(h, h->gstage, (size_t)n_ + 2)) return rc;
            } else {
                // |g|^2 over the owned vertices no other rank holds rides with the packet, the shared entries' squares are
                // added from the summed packet
                launch_masked_norm2(n_, h->gstage, h->vkind, h->partR, h->st, 1);
                if (int rc = exchange_gradient_packed(h, n_, nb, h->partR, 1)) return rc;
                ctlR = h->partGR;
            }
            HIPCHECK(h, hipMemcpyAsync(h->g, h->gstage, sizeof(double) * n_, hipMemcpyDeviceToDevice, h->st));
            if (!h->owner) {
                const double *vecs[1] = {h->g};
                launch_multidot(n_, h->g, vecs, 1, h->partR, h->st);   // |g|^2
            }
            if (!h->earlyNow) {
                launch_loop_control(h->ctl, h->gstage + n_, 1, ctlR, h->alpha_dev, h->h_flags, h->st, 1);
            } else {
                // early order on the sharded element pass: -g_0 (the summed gradient) into this rank's right-hand sides, the
                // first direction's solve with the start-of-step controller inside its launch, the sum over the ranks, z = u
                launch_build_qpad(h->P, h->g, L0, nullptr, h->st, nullptr);
                CtlArgs ca{h->ctl, h->gstage + n_, ctlR, h->alpha_dev, h->h_flags, 1, 1};
                launch_gemv(h->P, nullptr, h->st, h->ctl, nullptr, nullptr, &ca, 1 << 30);
                launch_merge(h->M, h->P, L0, h->zstage, h->partC, 0, h->st, h->ctl, h->held());
                if (!h->owner) {
                    if (int rc = allreduce_sum(h, h->zstage, n_)) return rc;
                    launch_merge_early(h->M, h->P, h->z, h->partC, 1, h->st, h->ctl, h->zstage);
                } else {
                    if (int rc = exchange_iface(h, h->zstage, nullptr, 0)) return rc;
                    // (no pair yet: no y_i . z to sum)
                    launch_merge_early(h->M, h->P, h->z, nullptr, 1, h->st, h->ctl, h->zstage, h->ownMask, h->held());
                    HIPCHECK(h, hipMemsetAsync(h->partGC, 0, sizeof(double) * HIST_MAX, h->st));
                }
            }
        }
    }
    int enq = 0;
    const double tStart = now_ms();
    long spins = 0;
    if (h->dist) {
        // Sharded subdomains: every slot carries a collective, so every rank must enqueue the SAME number of slots.
        // Deterministic batches instead of following the posted progress: one slot more than the last step used, then
        // (rarely) short batches; after a batch the ranks check against rank 0 that they stopped in the same state.
        int target = std::max(h->prevSlots + 1, 4);
        for (;;) {
            while (enq < target) {
                if (int rc = enqueue_loop_slot(h)) return rc;
                ++enq;
            }
            HIPCHECK(h, hipMemcpyAsync(h->h_ctl, h->ctl, sizeof(DevLoop), hipMemcpyDeviceToHost, h->st));
            HIPCHECK(h, hipStreamSynchronize(h->st));
            HIPCHECK(h, hipGetLastError());
            // Every rank contributes v = (status, slots, iterations, halvings), its squares and a local error flag to ONE
            // sum all-reduce.  All ranks see the same sums, so the verdict is a function of reduced data only and they
            // fail together instead of one of them waiting in the next collective: the ranks agree exactly when the
            // variance vanishes, world * sum(v^2) == (sum v)^2 (small integers, exact in FP64) -- the earlier test
            // `sum == world * mine` could pass on one rank and fail on the others for world >= 3.
            double mine[4] = {(double)C.status, (double)C.slots, (double)C.iter, (double)C.halvings}, red[9];
#ifdef DOTMI_TEST_HOOKS
            mine[2] += (double)h->testIterDelta;   // this process reports a different iteration count
#endif
            for (int i = 0; i < 4; ++i) {
                red[i] = mine[i];
                red[4 + i] = mine[i] * mine[i];
            }
            red[8] = 0.0;   // local error flag (set by a rank that cannot go on; summed like the rest)
            if (h->world > 1) {
                HIPCHECK(h, hipMemcpyAsync(h->ctrlDev, red, sizeof(red), hipMemcpyHostToDevice, h->st));
                if (int rc = allreduce_sum(h, h->ctrlDev, 9)) return rc;
                HIPCHECK(h, hipMemcpyAsync(red, h->ctrlDev, sizeof(red), hipMemcpyDeviceToHost, h->st));
                HIPCHECK(h, hipStreamSynchronize(h->st));
            }
            const double w = h->world > 1 ? (double)h->world : 1.0;
            bool agree = red[8] == 0.0;
            for (int i = 0; i < 4; ++i) agree = agree && w * red[4 + i] == red[i] * red[i];
            if (!agree) {
                h->err = "the ranks left the L-BFGS loop in different states (this rank: status " + std::to_string(C.status) +
                         " after " + std::to_string(C.slots) + " slots, " + std::to_string(C.iter) + " iterations; sum over " +
                         std::to_string(h->world) + " ranks: " + std::to_string((long long)red[0]) + " / " +
                         std::to_string((long long)red[1]) + " / " + std::to_string((long long)red[2]) + ")";
                h->poisoned = true;
                return DOTMI_E_DEVICE;
            }
            if (C.status != 0) break;
            target = enq + std::max(2, std::min(8, enq / 4));
            if (now_ms() - tStart > 600000.0) {
                h->err = "device loop timed out";
                return DOTMI_E_DEVICE;
            }
        }
    } else
    while (flags[0] == 0) {
        if (enq < std::max(notifyFrom, (int)flags[1]) + AHEAD) {
            if (int rc = enqueue_loop_slot(h)) return rc;
            ++enq;
        } else if ((++spins & 0xfffff) == 0) {
            if (hipStreamQuery(h->st) != hipErrorNotReady && flags[0] == 0) {
                // the queue drained without the controller reporting progress: a kernel failed
                HIPCHECK(h, hipStreamSynchronize(h->st));
                HIPCHECK(h, hipGetLastError());
                if (flags[0] == 0 && enq >= std::max(notifyFrom, (int)flags[1]) + AHEAD) {
                    h->err = "device loop made no progress";
                    return DOTMI_E_DEVICE;
                }
            }
            if (now_ms() - tStart > 600000.0) {
                h->err = "device loop timed out";
                return DOTMI_E_DEVICE;
            }
        }
    }
    HIPCHECK(h, hipMemcpyAsync(h->h_ctl, h->ctl, sizeof(DevLoop), hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    *it = C.iter;
    h->prevSlots = C.slots;
    h->prevIters = C.iter;
    h->prevHalv = C.halvings;
    memcpy(h->predState, C.predHist, sizeof(int) * 2);
    memcpy(h->predState + 2, &C.predCtr[0][0], sizeof(int) * 8);
    h->heldSlots = C.heldSlots;
    h->heldRejected = C.heldRejected;
    *failed = C.status == 3;
    *lastE = C.E_cur;
    *g2 = C.g2_cur;
    *E0 = C.E0;
    *g20 = C.g2_0;
    h->x = C.x_cur;
    h->x_trial = C.x_trial;
    h->g = C.g_cur;
    h->g_trial = C.g_trial;
    h->numLineSearch += C.halvings;
    h->energy_evals += C.evals;
    // the per-iteration log stays on the device until somebody asks for it (dotmi_last_iter_log)
    h->logPending = std::min(C.iter, h->logCap);
    // which of the enqueued slots really ran a back-solve (for DOTMI_FLAG_TIME_BACKSOLVE)
    h->slotKind.assign(enq, 0);
    const int nk = std::min(std::min(C.slots, enq), h->kindCap);
    if (nk > 0 && (h->flags & DOTMI_FLAG_TIME_BACKSOLVE))
        HIPCHECK(h, hipMemcpy(h->slotKind.data(), h->dkind, sizeof(int) * nk, hipMemcpyDeviceToHost));
    if (h->earlyNow && (h->flags & DOTMI_FLAG_TIME_BACKSOLVE)) {
        // early order: slot sl's back-solve ran to its end iff its trial was accepted and the loop went on, i.e. iff slot
        // sl + 1 computed a new direction (kind 1); the others were told to stop and do not count as timed launches
        for (int sl = 0; sl < nk; ++sl) h->slotKind[sl] = (sl + 1 < nk && h->slotKind[sl + 1] == 1) ? 1 : 0;
    }
    return 0;
}

// The reference's Gauss-Seidel domain-decomposition iteration (`timeStepper GSDD`, DOTTimeStepper::solve_oneStep_GSDD,
// DOTTimeStepper.cpp:507-565, driven by fullyImplicit :299-337) on this path's factors and kernels: one sweep over the
// subdomains per iteration; for subdomain s  p_s = H_s^-1 (-g restricted to s)  (:521-527), the search direction is p_s
// on the subdomain's vertices and zero elsewhere (:529-532), the line search starts from step 1 (initStepSize,
// Optimizer.cpp:1076-1093: only TST_DOT estimates it) and halves while the energy increases, and the gradient is
// brought up to date before the next subdomain (:541-551).  Host-driven: every trial needs its energy on the host.
int run_gsdd_loop(dotmi_handle *h, double *lastE, double *g2, int *it, bool *failed)
{
    const int n = h->n;
    LbfgsArgs L0;
    memset(&L0, 0, sizeof(L0));
    double R[RED_K];
    do {
        for (int ls = 0; ls < h->P.nParts && !*failed; ++ls) {
            launch_build_q(n, h->g, L0, nullptr, h->q, h->st);                     // q = -g
            launch_gemv_part(h->P, ls, h->P.tileByPart + h->partTilePtr[ls], h->partTilePtr[ls + 1] - h->partTilePtr[ls],
                             h->P.lworkByPart + h->partLworkPtr[ls], h->partLworkPtr[ls + 1] - h->partLworkPtr[ls],
                             h->q, n, h->p, h->st);
            double alpha = 1.0, E = 0;
            launch_step_forward(n, h->x, h->p, h->x_trial, nullptr, alpha, 0, h->alphaMin, h->alpha_dev, h->h_alpha, h->st);
            if (int rc = trial(h, h->x_trial, h->g_trial, 0, L0, 0, &E)) return rc;
            while (E > *lastE && alpha > 0.0) {   // Optimizer.cpp:806-833
                alpha /= 2.0;
                h->numLineSearch++;
                if (alpha == 0.0) {
                    *failed = true;
                    break;
                }
                launch_step_forward(n, h->x, h->p, h->x_trial, nullptr, alpha, 0, h->alphaMin, h->alpha_dev, h->h_alpha,
                                    h->st);
                if (int rc = trial(h, h->x_trial, h->g_trial, 0, L0, 0, &E)) return rc;
            }
            std::swap(h->x, h->x_trial);   // also on failure: the reference stays at the last trial point
            std::swap(h->g, h->g_trial);   // the gradient of the accepted point came with its energy
            *lastE = E;
            h->log_alpha.push_back(alpha);
            h->log_E.push_back(E);
            sum_stats(h, 1, R);
            h->log_g2.push_back(R[0]);
        }
        if (*failed) break;
        sum_stats(h, 1, R);
        *g2 = R[0];
        if (++*it >= h->iterCap) break;
    } while (*g2 > h->targetGRes);
    return 0;
}

// The reference's projected Newton (`timeStepper Newton`: the base Optimizer::fullyImplicit, Optimizer.cpp:654-700, with
// Optimizer::solve_oneStep :703-749 and needRefactorize set): per iteration the projected Hessian at the current iterate
// is assembled and factorised (:705-729), p = H^-1 (-g) (:735-737), the line search starts from step 1 (initStepSize
// :1088) and the gradient is refreshed (:745).  Uses the same refresh / back-solve kernels as the DOT path.
int run_newton_loop(dotmi_handle *h, double *lastE, double *g2, int *it, bool *failed, double *ms_hess, double *ms_fact)
{
    const int n = h->n;
    LbfgsArgs L0;
    memset(&L0, 0, sizeof(L0));
    double R[RED_K];
    do {
        if (int rc = refactor(h, h->x, ms_hess, ms_fact)) return rc;
        launch_build_q(n, h->g, L0, nullptr, h->q, h->st);                     // q = -g
        if (int rc = apply_precond(h, h->q, h->p, L0)) return rc;               // p = H^-1 q (one subdomain: no averaging)
        double alpha = 1.0, E = 0;
        launch_step_forward(n, h->x, h->p, h->x_trial, nullptr, alpha, 0, h->alphaMin, h->alpha_dev, h->h_alpha, h->st);
        if (int rc = trial(h, h->x_trial, h->g_trial, 0, L0, 0, &E)) return rc;
        while (E > *lastE && alpha > 0.0) {   // Optimizer.cpp:806-833
            alpha /= 2.0;
            h->numLineSearch++;
            if (alpha == 0.0) {
                *failed = true;
                break;
            }
            launch_step_forward(n, h->x, h->p, h->x_trial, nullptr, alpha, 0, h->alphaMin, h->alpha_dev, h->h_alpha, h->st);
            if (int rc = trial(h, h->x_trial, h->g_trial, 0, L0, 0, &E)) return rc;
        }
        std::swap(h->x, h->x_trial);
        std::swap(h->g, h->g_trial);
        *lastE = E;
        if (*failed) break;
        sum_stats(h, 1, R);
        *g2 = R[0];
        h->log_alpha.push_back(alpha);
        h->log_E.push_back(E);
        h->log_g2.push_back(*g2);
        if (++*it >= h->iterCap) break;
    } while (*g2 > h->targetGRes);
    return 0;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
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

// host-only: the level schedule of tile_factor.hpp for ONE block of nt x nt tiles with the given upper tile pattern, in the
// compact row-block layout (c0[j] = first tile column stored for tile row j, c0[j] <= every pattern entry of column j).
// Offsets are in doubles into one array: the factor storage first, the scratch tiles after it (*scratch_base).
//   tasks: 10 int64 per task  {level, form, init, post, nprod, first product, c offset, q offset (-1), ldc, ldq}
//   prods:  4 int64 per product {a offset, b offset, lda, ldb}
// With tasks == NULL only the counts are returned.  The tests execute the schedule in numpy, level after level, and
// compare with a dense inverse Cholesky factor (tests/test_tile_schedule.py).
int dotmi_plan_tile_schedule(int32_t nt, const uint8_t *live, const uint8_t *pattern, const int32_t *c0, int32_t eager_min,
                             int32_t eager_chunk, int64_t *tasks, int64_t *prods, int64_t *n_tasks, int64_t *n_prods,
                             int64_t *n_levels, int64_t *storage, int64_t *scratch_base, int64_t *row_off, int32_t *row_ld)
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
    double *const W = reinterpret_cast<double *>(1ull << 40);   // never dereferenced: only offsets leave this function
    double *const scratch = W + tot;
    std::vector<uint8_t> lv(live, live + nt), pat(pattern, pattern + (size_t)nt * nt);
    std::vector<TileTaskL> all;
    TileSchedule S;
    size_t sn = 0;
    plan_subdomain_tiles(0, nt, W, rtOff.data(), rtLd.data(), rtC0.data(), lv, pat, scratch, sn, all, S.clearTiles, S.clearLd,
                         S.flops, S.qTiles, std::max(1, eager_min), std::max(1, eager_chunk));
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
    int levelsDefault = 2;
    if (levels < 0) {   // dotmi_create's rule: from the sizes of ALL subdomains of the mesh
        std::vector<int> cnt(nParts, 0), mark(nV, -1);
        for (int pI = 0; pI < nParts; ++pI)
            for (int e = 0; e < nT; ++e)
                if (epart[e] == pI)
                    for (int k = 0; k < 4; ++k)
                        if (mark[T[4 * e + k]] != pI) {
                            mark[T[4 * e + k]] = pI;
                            cnt[pI] += 3;
                        }
        levelsDefault = nd_default_levels(*std::max_element(cnt.begin(), cnt.end()), nParts);
    }
    *nmax = nd_plan(sets, nV, adj_ptr, adj_idx, Xrest, levels < 0 ? levelsDefault : levels,
                    min_split < 128 ? ND_MIN_SPLIT : min_split, tree, region);
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
    return h->tileFlow ? 2 : 1;
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
    if (int rc = dalloc(h, &h->Hval, (size_t)9 * h->M.nnzb)) return rc;
    if (h->owner) {
        if (int rc = dalloc(h, &h->HvalOwn, (size_t)9 * h->M.nnzb)) return rc;
        HIPCHECK(h, hipMemsetAsync(h->HvalOwn, 0, sizeof(double) * 9 * h->M.nnzb, h->st));
    }
    if (int rc = dalloc(h, &h->partE, (size_t)2 * ELEM_NB_MAX)) return rc;
    double **parts[] = {&h->partR, &h->partC, &h->partS, &h->partG, &h->partGR, &h->partGC};
    for (double **pp : parts) {
        if (int rc = dalloc(h, pp, (size_t)NB_RED * RED_K)) return rc;
        HIPCHECK(h, hipMemsetAsync(*pp, 0, sizeof(double) * NB_RED * RED_K, h->st));
    }
    if (int rc = dalloc(h, &h->alpha_dev, 8)) return rc;
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
        if (h->earlyBs && h->tune.fuseStep && (!h->shardElems || h->owner)) h->PT.wgCap = 512;
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

int dotmi_set_state(dotmi_handle *h, const double *x, const double *v, const double *x_n)
{
    if (!h || !x || !v) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    const size_t bytes = sizeof(double) * h->n;
    HIPCHECK(h, hipMemcpyAsync(h->x, x, bytes, hipMemcpyHostToDevice, h->st));
    HIPCHECK(h, hipMemcpyAsync(h->v, v, bytes, hipMemcpyHostToDevice, h->st));
    HIPCHECK(h, hipMemcpyAsync(h->xn, x_n ? x_n : x, bytes, hipMemcpyHostToDevice, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    // x~ = x_n + dt v + dt^2 g on free vertices, x_n on fixed ones (Optimizer.cpp:585-610)
    std::vector<double> xt(h->n);
    const double *xn_h = x_n ? x_n : x;
    for (int i = 0; i < h->nV; ++i)
        for (int d = 0; d < 3; ++d) {
            const int k = 3 * i + d;
            xt[k] = h->fixed[i] ? xn_h[k] : xn_h[k] + (v[k] * h->dt + h->gdtsq[d]);
        }
    HIPCHECK(h, hipMemcpy(h->xt, xt.data(), bytes, hipMemcpyHostToDevice));
    return 0;
}

int dotmi_get_state(dotmi_handle *h, double *x, double *v, double *x_tilde)
{
    if (!h) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    const size_t bytes = sizeof(double) * h->n;
    HIPCHECK(h, hipStreamSynchronize(h->st));
    if (x) HIPCHECK(h, hipMemcpy(x, h->x, bytes, hipMemcpyDeviceToHost));
    if (v) HIPCHECK(h, hipMemcpy(v, h->v, bytes, hipMemcpyDeviceToHost));
    if (x_tilde) HIPCHECK(h, hipMemcpy(x_tilde, h->xt, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int dotmi_set_dirichlet(dotmi_handle *h, int32_t n, const int32_t *idx, const double *pos)
{
    if (!h || n < 0 || (n > 0 && (!idx || !pos))) return DOTMI_E_INVALID;
    if (n == 0) return 0;
    HIPCHECK(h, hipSetDevice(h->device));
    for (int i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= h->nV) {
            h->err = "dirichlet index out of range";
            return DOTMI_E_INVALID;
        }
    if ((size_t)n > h->dcap) {
        if (int rc = dalloc(h, &h->didx, (size_t)n)) return rc;
        if (int rc = dalloc(h, &h->dpos, (size_t)3 * n)) return rc;
        h->dcap = n;
    }
    // the scripted set is the same every step: the indices go up only when they change, the positions through a
    // pinned staging buffer, and nothing waits here (the stream orders the scatter before the step's kernels)
    if (h->didxHost.size() != (size_t)n || memcmp(h->didxHost.data(), idx, sizeof(int32_t) * n) != 0) {
        HIPCHECK(h, hipStreamSynchronize(h->st));
        h->didxHost.assign(idx, idx + n);
        if (h->dposPinned) hipHostFree(h->dposPinned);   // the stream is idle: nothing reads the staging buffer
        h->dposPinned = nullptr;
        HIPCHECK(h, hipHostMalloc((void **)&h->dposPinned, sizeof(double) * 3 * n));
        HIPCHECK(h, hipMemcpyAsync(h->didx, h->didxHost.data(), sizeof(int) * n, hipMemcpyHostToDevice, h->st));
    }
    if (!h->evDir) HIPCHECK(h, hipEventCreateWithFlags(&h->evDir, hipEventDisableTiming));
    else HIPCHECK(h, hipEventSynchronize(h->evDir));  // the previous upload out of the staging buffer (normally long done)
    memcpy(h->dposPinned, pos, sizeof(double) * 3 * n);
    HIPCHECK(h, hipMemcpyAsync(h->dpos, h->dposPinned, sizeof(double) * 3 * n, hipMemcpyHostToDevice, h->st));
    HIPCHECK(h, hipEventRecord(h->evDir, h->st));
    launch_scatter_rows(n, h->didx, h->dpos, h->x, h->st);
    return 0;
}

int dotmi_refix(dotmi_handle *h, const uint8_t *fixed)
{
    if (!h || !fixed) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    h->fixed.assign(fixed, fixed + h->nV);
    HIPCHECK(h, hipMemcpy(h->M.fixed, fixed, h->nV, hipMemcpyHostToDevice));
    return refactor(h, h->x, nullptr, nullptr);
}

double dotmi_target_gres(const dotmi_handle *h) { return h ? h->targetGRes : 0.0; }

int dotmi_last_iter_log(const dotmi_handle *h, int32_t cap, double *alpha, double *E, double *g2)
{
    if (!h) return DOTMI_E_INVALID;
    if (h->logPending > 0) {
        dotmi_handle *hm = const_cast<dotmi_handle *>(h);
        const int nlog = h->logPending;
        hm->logPending = 0;
        hm->log_alpha.resize(nlog);
        hm->log_E.resize(nlog);
        hm->log_g2.resize(nlog);
        if (hipSetDevice(h->device) != hipSuccess ||
            hipMemcpy(hm->log_alpha.data(), h->dlog, sizeof(double) * nlog, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hm->log_E.data(), h->dlog + h->logCap, sizeof(double) * nlog, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hm->log_g2.data(), h->dlog + 2 * (size_t)h->logCap, sizeof(double) * nlog, hipMemcpyDeviceToHost) !=
                hipSuccess)
            return DOTMI_E_DEVICE;
    }
    const int n = std::min<int>(cap, (int)h->log_alpha.size());
    for (int i = 0; i < n; ++i) {
        if (alpha) alpha[i] = h->log_alpha[i];
        if (E) E[i] = h->log_E[i];
        if (g2) g2[i] = h->log_g2[i];
    }
    return (int)h->log_alpha.size();
}

int dotmi_step(dotmi_handle *h, dotmi_step_stats *st)
{
    if (!h) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    // a refresh still running from the last step is judged BEFORE anything of this step is enqueued: a step never runs on
    // factors whose factorisation failed (what the caller did between the two steps has already overlapped the refresh)
    if (int rc = enter_with_factors(h)) return rc;
    const double T0 = now_ms();
    const int n = h->n;
    h->m = 0;
    h->energy_evals = 0;
    h->evUsed = 0;
    h->evArUsed = 0;
    h->arTimedBytes.clear();
    h->arCallsStep = 0;
    h->arBytesStep = 0;
    h->log_alpha.clear();
    h->log_E.clear();
    h->log_g2.clear();
    h->logPending = 0;
    const long long ls0 = h->numLineSearch;
    double ms_hess = 0, ms_fact = 0;

    for (double &v : h->phaseMs) v = 0.0;
    h->evPn = 0;
    phase_mark(h, -1);
    // initX(2): x += dt v + dt^2 g on free vertices (Optimizer.cpp:442-582)
    launch_init_x(h->nV, h->M.fixed, h->v, h->dt, h->gdtsq, h->x, h->st);
    LbfgsArgs L = lbfgs_args(h);
    double lastE = 0, R[RED_K], g2 = 0, E0 = 0, g20 = 0;
    if (!h->devLoop) {
        if (int rc = trial(h, h->x, h->g, 0, L, 0, &lastE, DOTMI_T_FULLYIMPLICIT_ECOMP, DOTMI_T_FULLYIMPLICIT_ECOMP))
            return rc;
        sum_stats(h, 1, R);
        g2 = R[0];
        E0 = lastE;
        g20 = g2;
    }

    int it = 0, status = 0;
    bool failed = false;
    const double Tloop = now_ms();
    h->slotKind.clear();
    h->slotTimed.clear();
    if (h->newton) {
        if (int rc = run_newton_loop(h, &lastE, &g2, &it, &failed, &ms_hess, &ms_fact)) return rc;
    } else if (h->gsdd) {
        if (int rc = run_gsdd_loop(h, &lastE, &g2, &it, &failed)) return rc;
    } else if (h->devLoop) {
        if (int rc = run_device_loop(h, &lastE, &g2, &it, &failed, &E0, &g20)) return rc;
    } else
    do {
        // ---- two-loop, first half (host scalars) + q ------------------------------------------------
        double xi[HIST_MAX] = {0};
        for (int i = h->m - 1; i >= 0; --i) {
            double sq = -h->b[i];
            for (int j = h->m - 1; j > i; --j) sq -= xi[j] * h->sy[i][j];
            xi[i] = sq / h->ys[i];
        }
        L = lbfgs_args(h);
        phase_mark(h, -1);
        launch_build_qpad(h->P, h->g, L, xi, h->st);   // q, straight into the padded right-hand sides
        phase_mark(h, DOTMI_T_MODIFY_GRAD);
        // ---- subdomain back-solve, merge, second half ------------------------------------------------
        if (int rc = apply_precond(h, nullptr, h->z, L)) return rc;
        phase_mark(h, DOTMI_T_BACKSOLVE);
        launch_build_p(n, h->z, L, h->partC, xi, h->p, h->st);
        phase_mark(h, DOTMI_T_MODIFY_SEARCHDIR);
        // ---- alpha_0 and the first trial ---------------------------------------------------------------
        launch_spmv_dots(h->M, h->Hval, h->p, h->g, nullptr, h->v0, h->v1, h->partS, h->st);
        const double *spart = h->partS;
        if (h->shardElems) {
            hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(64), 0, h->st, h->partS, NB_RED, RED_K, 2, 0.0,
                               0.0, 0, h->partG);
            if (int rc = allreduce_sum(h, h->partG, 2)) return rc;
            spart = h->partG;  // rows >= 1 stay zero
        }
        launch_step_forward(n, h->x, h->p, h->x_trial, spart, 0.0, 1, h->alphaMin, h->alpha_dev, h->h_alpha, h->st);
        phase_mark(h, DOTMI_T_LINESEARCH_OTHER);
        const int slot = free_slot(h);
        double E = 0;
        if (int rc = trial(h, h->x_trial, h->g_trial, 1, L, slot, &E)) return rc;
        double alpha = h->h_alpha[0];
        // ---- back-tracking (Optimizer.cpp:806-833; c1 = 0, lower bound 0) ----------------------------
        while (E > lastE && alpha > 0.0) {
            alpha /= 2.0;
            h->numLineSearch++;
            if (alpha == 0.0) {
                failed = true;
                break;
            }
            phase_mark(h, -1);
            launch_step_forward(n, h->x, h->p, h->x_trial, nullptr, alpha, 0, h->alphaMin, h->alpha_dev, h->h_alpha, h->st);
            phase_mark(h, DOTMI_T_LINESEARCH_OTHER);
            if (int rc = trial(h, h->x_trial, h->g_trial, 1, L, slot, &E)) return rc;
        }
        if (failed) {
            // the reference leaves result.V at the last trial point and lastEnergyVal at its energy
            // (Optimizer.cpp:819-861); the iteration is not counted (DOTTimeStepper.cpp:311-316)
            std::swap(h->x, h->x_trial);
            lastE = E;
            break;
        }
        std::swap(h->x, h->x_trial);
        std::swap(h->g, h->g_trial);
        lastE = E;
        // ---- history update (DOTTimeStepper.cpp:474-494) --------------------------------------------
        sum_stats(h, RED_K, R);
        g2 = R[0];
        const double ys_new = R[1], sg_new = R[2];
        double *siy = R + 3, *snyj = R + 3 + HIST_MAX, *sig = R + 3 + 2 * HIST_MAX;
        if (ys_new > 0.0) {
            int m = h->m;
            int off = 0;
            if (m == h->hist) {  // drop the oldest pair
                off = 1;
                for (int i = 0; i + 1 < m; ++i) {
                    h->order[i] = h->order[i + 1];
                    h->ys[i] = h->ys[i + 1];
                    for (int j = 0; j + 1 < m; ++j) h->sy[i][j] = h->sy[i + 1][j + 1];
                }
                m -= 1;
            }
            for (int i = 0; i < m; ++i) {
                h->sy[i][m] = siy[i + off];
                h->sy[m][i] = snyj[i + off];
                h->b[i] = sig[i + off];
            }
            h->order[m] = slot;
            h->ys[m] = ys_new;
            h->sy[m][m] = ys_new;
            h->b[m] = sg_new;
            h->m = m + 1;
        } else {
            for (int i = 0; i < h->m; ++i) h->b[i] = sig[i];
        }
        h->log_alpha.push_back(alpha);
        h->log_E.push_back(lastE);
        h->log_g2.push_back(g2);
        if (++it >= h->iterCap) break;
    } while (g2 > h->targetGRes);
    if (h->owner) {
        // owner exchange: the loop kept the positions of the vertices this rank holds; every rank's positions are made whole
        // again here, once per step (the owners' entries, zeros elsewhere, summed) -- the refresh reads the halo elements'
        // vertices, dotmi_get_state everything
        launch_mask_owned(h->n, h->x, h->ownMask, h->st);
        if (int rc = allreduce_sum(h, h->x, (size_t)h->n)) return rc;
    }
    double Tloop1 = now_ms();
    ms_hess += h->carryHess;
    ms_fact += h->carryFact;
    h->carryHess = h->carryFact = 0.0;

    const bool refreshAtEnd = !failed && !h->newton;   // Newton refreshes at the START of every iteration instead
    if (failed) status = 2;
    else {
        if (it >= h->iterCap) status = 2;
        if (refreshAtEnd)
            if (int rc = refactor_issue(h, h->x)) return rc;
    }
    // BE update (Optimizer.cpp:354-361)
    phase_mark(h, -1);
    launch_be_update(h->nV, h->M.fixed, h->x, h->xn, h->v, h->xt, h->dt, h->gdtsq, h->st);
    phase_mark(h, DOTMI_T_SOLVE_EXTRACOMP);
    int rcFactor = 0;
    const bool asyncRefresh = refreshAtEnd && (h->flags & DOTMI_FLAG_ASYNC_REFRESH) && h->devLoop && !h->dist;
    if (asyncRefresh) {
        // the refresh and the BE update stay queued; whoever needs their result next waits for them (resolve_refresh)
        h->refreshPending = true;
    } else {
        HIPCHECK(h, hipStreamSynchronize(h->st));
        phase_collect(h);
        HIPCHECK(h, hipGetLastError());
        if (refreshAtEnd) rcFactor = refactor_finish(h, &ms_hess, &ms_fact);
        if (rcFactor == DOTMI_E_DEVICE) return rcFactor;
    }
    if (st) {
        memset(st, 0, sizeof(*st));
        st->iters = it;
        st->ls_halvings = (int)(h->numLineSearch - ls0);
        st->energy_evals = h->energy_evals;
        st->status = status;
        st->E0 = E0;
        st->g2_0 = g20;
        st->E = lastE;
        st->g2 = g2;
        st->ms_total = now_ms() - T0;
        st->ms_loop = Tloop1 - Tloop;
        st->ms_hessian = ms_hess;
        st->ms_factor = ms_fact;
        std::vector<char> ran(h->evUsed / 2 + 1, h->devLoop ? 0 : 1);
        // device loop: slots enqueued past the end, and line-search retries, ran no back-solve
        for (size_t sl = 0; sl < h->slotTimed.size(); ++sl)
            if (h->slotTimed[sl] >= 0 && sl < h->slotKind.size() && h->slotKind[sl] == 1) ran[h->slotTimed[sl] / 2] = 1;
        for (int k = 0; k + 1 < h->evUsed; k += 2) {
            if (!ran[k / 2]) continue;
            float ms = 0;
            hipEventElapsedTime(&ms, h->evPre[k], h->evPre[k + 1]);
            st->ms_precond += ms;
            st->precond_launches++;
        }
        st->precond_bytes = h->precond_bytes;
        st->factor_flops = h->factorFlops;
        st->backsolve_launches = it;
        st->backsolve_stopped = (h->devLoop && h->earlyNow) ? (h->numLineSearch - ls0) + 1 : 0;
        st->backsolve_held = (h->devLoop && h->earlyNow) ? h->heldSlots : 0;
        st->backsolve_held_rejected = (h->devLoop && h->earlyNow) ? h->heldRejected : 0;
        for (int k = 0; k + 1 < h->evArUsed; k += 2) {
            float ms = 0;
            hipEventElapsedTime(&ms, h->evAr[k], h->evAr[k + 1]);
            st->ms_collective += ms;
            st->collective_timed++;
            st->collective_timed_bytes += (int64_t)h->arTimedBytes[k / 2];
        }
        st->collective_calls = h->arCallsStep;
        st->collective_bytes = (int64_t)h->arBytesStep;
        for (int k = 0; k < DOTMI_T_COUNT; ++k) st->ms_phase[k] = h->phaseMs[k];
    }
    // a non-SPD subdomain: the step itself is complete (x, v advanced as the reference would have before it
    // exit(-1)s in the factorisation, Optimizer.cpp:301-313), the handle is poisoned until a refactor succeeds
    if (rcFactor) return rcFactor;
    return status;
}

// ---- kernel-level entry points ------------------------------------------------------------------
static int upload_tmp(dotmi_handle *h, const double *x, double *dst)
{
    HIPCHECK(h, hipMemcpyAsync(dst, x, sizeof(double) * h->n, hipMemcpyHostToDevice, h->st));
    return 0;
}

int dotmi_eval_energy(dotmi_handle *h, const double *x, double *E)
{
    if (!h || !x || !E) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    if (int rc = upload_tmp(h, x, h->x_trial)) return rc;
    int nb = 0;
    launch_elem_energy_grad(h->M, h->PTall, h->mat, h->dtSq, h->x_trial, h->xt, 0, h->nV, 0, h->partE,
                            &nb, h->st);
    HIPCHECK(h, hipMemcpyAsync(h->h_partE, h->partE, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    double se = 0, si = 0;
    for (int b = 0; b < nb; ++b) {
        se += h->h_partE[2 * b];
        si += h->h_partE[2 * b + 1];
    }
    *E = h->dtSq * se + si;
    return 0;
}

int dotmi_eval_gradient(dotmi_handle *h, const double *x, double *g)
{
    if (!h || !x || !g) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    if (int rc = upload_tmp(h, x, h->x_trial)) return rc;
    int nb = 0;
    launch_elem_energy_grad(h->M, h->PTall, h->mat, h->dtSq, h->x_trial, h->xt, 0, h->nV, 1, h->partE,
                            &nb, h->st);
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.x = h->x_trial;
    a.xt = h->xt;
    a.g_new = h->g_trial;
    a.make_pair = 0;
    a.iv0 = 0;
    a.iv1 = h->nV;
    LbfgsArgs L;
    memset(&L, 0, sizeof(L));
    launch_vertex_gather(h->M, h->PTall, a, L, h->partR, h->st);
    HIPCHECK(h, hipMemcpyAsync(g, h->g_trial, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    return 0;
}

int dotmi_eval_elem_hessians(dotmi_handle *h, const double *x, double *H)
{
    if (!h || !x || !H) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    if (int rc = upload_tmp(h, x, h->x_trial)) return rc;
    // scratch copy so the resident He (state of the current preconditioner) is not disturbed
    double *tmp = nullptr;
    HIPCHECK(h, hipMalloc((void **)&tmp, sizeof(double) * 144 * (size_t)h->nT));
    launch_elem_hessians(h->M, h->mat, h->dtSq, h->x_trial, tmp, h->st);
    hipError_t e = hipMemcpyAsync(H, tmp, sizeof(double) * 144 * (size_t)h->nT, hipMemcpyDeviceToHost, h->st);
    hipStreamSynchronize(h->st);
    hipFree(tmp);
    HIPCHECK(h, e);
    return 0;
}

int dotmi_refactor(dotmi_handle *h, const double *x)
{
    if (!h) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    const double *xd = h->x;
    if (x) {
        if (int rc = upload_tmp(h, x, h->x_trial)) return rc;
        xd = h->x_trial;
    }
    return refactor(h, xd, nullptr, nullptr);
}

int dotmi_apply_precond(dotmi_handle *h, const double *r, double *p)
{
    if (!h || !r || !p) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (int rc = enter_with_factors(h)) return rc;
    if (int rc = upload_tmp(h, r, h->q)) return rc;
    LbfgsArgs L;
    memset(&L, 0, sizeof(L));
    if (int rc = apply_precond(h, h->q, h->z, L)) return rc;
    HIPCHECK(h, hipMemcpyAsync(p, h->z, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    return 0;
}

// One L-BFGS-H direction and its first line-search trial from a caller-supplied iterate and history, with the
// kernels of the host-driven loop (DOTTimeStepper.cpp:386-467, Optimizer.cpp:1076-1093, :791).  Teacher forcing
// (SURVEY.md 8(c) F4): a test feeds the oracle's (x, history) of iteration k and compares q, z, p, alpha_0 and
// E(x + alpha_0 p).  Uses the current factors and x~; the L-BFGS slots it overwrites are reset by the next step.
int dotmi_probe_direction(dotmi_handle *h, const double *x, int32_t m, const double *S, const double *Y, double *g_out,
                          double *q_out, double *z_out, double *p_out, double *alpha0, double *E_trial)
{
    if (!h || !x || m < 0 || m > h->hist || (m > 0 && (!S || !Y))) return DOTMI_E_INVALID;
    if (h->dist) {
        h->err = "dotmi_probe_direction: single-GPU handles only";
        return DOTMI_E_INVALID;
    }
    HIPCHECK(h, hipSetDevice(h->device));
    if (int rc = enter_with_factors(h)) return rc;
    const int n = h->n;
    const size_t bytes = sizeof(double) * n;
    // the probe works on tmpn (iterate), g_trial (gradient), x_trial (trial point): the handle's own x, g stay
    HIPCHECK(h, hipMemcpyAsync(h->tmpn, x, bytes, hipMemcpyHostToDevice, h->st));
    for (int i = 0; i < m; ++i) {
        HIPCHECK(h, hipMemcpyAsync(h->S[i], S + (size_t)i * n, bytes, hipMemcpyHostToDevice, h->st));
        HIPCHECK(h, hipMemcpyAsync(h->Y[i], Y + (size_t)i * n, bytes, hipMemcpyHostToDevice, h->st));
    }
    int nb = 0;
    launch_elem_energy_grad(h->M, h->PTall, h->mat, h->dtSq, h->tmpn, h->xt, 0, h->nV, 1, h->partE, &nb,
                            h->st);
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.x = h->tmpn;
    a.xt = h->xt;
    a.g_new = h->g_trial;
    a.iv0 = 0;
    a.iv1 = h->nV;
    LbfgsArgs L;
    memset(&L, 0, sizeof(L));
    launch_vertex_gather(h->M, h->PTall, a, L, h->partR, h->st);
    std::vector<double> g(n);
    HIPCHECK(h, hipMemcpyAsync(g.data(), h->g_trial, bytes, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    // Gram matrix and the first half of the two-loop on the host (the running loop gets the same numbers from the
    // gather kernel's partial sums)
    L.m = m;
    double b[HIST_MAX] = {0}, xi[HIST_MAX] = {0};
    auto dot = [&](const double *u, const double *v) {
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc += u[k] * v[k];
        return acc;
    };
    for (int i = 0; i < m; ++i) {
        L.s[i] = h->S[i];
        L.y[i] = h->Y[i];
        b[i] = dot(S + (size_t)i * n, g.data());
        for (int j = 0; j < m; ++j) L.sy[i][j] = dot(S + (size_t)i * n, Y + (size_t)j * n);
        L.ys[i] = L.sy[i][i];
    }
    for (int i = m - 1; i >= 0; --i) {
        double sq = -b[i];
        for (int j = m - 1; j > i; --j) sq -= xi[j] * L.sy[i][j];
        xi[i] = sq / L.ys[i];
    }
    launch_build_q(n, h->g_trial, L, xi, h->q, h->st);
    if (int rc = apply_precond(h, h->q, h->z, L)) return rc;
    launch_build_p(n, h->z, L, h->partC, xi, h->p, h->st);
    launch_spmv_dots(h->M, h->Hval, h->p, h->g_trial, nullptr, 0, h->nV, h->partS, h->st);
    launch_step_forward(n, h->tmpn, h->p, h->x_trial, h->partS, 0.0, 1, h->alphaMin, h->alpha_dev, h->h_alpha, h->st);
    launch_elem_energy_grad(h->M, h->PTall, h->mat, h->dtSq, h->x_trial, h->xt, 0, h->nV, 0, h->partE, &nb,
                            h->st);
    HIPCHECK(h, hipMemcpyAsync(h->h_partE, h->partE, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, h->st));
    if (g_out) memcpy(g_out, g.data(), bytes);
    if (q_out) HIPCHECK(h, hipMemcpyAsync(q_out, h->q, bytes, hipMemcpyDeviceToHost, h->st));
    if (z_out) HIPCHECK(h, hipMemcpyAsync(z_out, h->z, bytes, hipMemcpyDeviceToHost, h->st));
    if (p_out) HIPCHECK(h, hipMemcpyAsync(p_out, h->p, bytes, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    if (alpha0) *alpha0 = h->h_alpha[0];
    if (E_trial) {
        const double se = chunked_sum(nb, [&](int k) { return h->h_partE[2 * k]; });
        const double si = chunked_sum(nb, [&](int k) { return h->h_partE[2 * k + 1]; });
        *E_trial = h->dtSq * se + si;
    }
    return 0;
}

int dotmi_spmv(dotmi_handle *h, const double *p, double *Hp)
{
    if (!h || !p || !Hp) return DOTMI_E_INVALID;
    if (h->shardHess) {
        h->err = "dotmi_spmv: the rows of the global Hessian are sharded over the ranks on this handle (DOTMI_SHARD_HESS=0 keeps them replicated)";
        return DOTMI_E_INVALID;
    }
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    if (int rc = upload_tmp(h, p, h->tmpn)) return rc;
    launch_spmv_dots(h->M, h->Hval, h->tmpn, nullptr, h->Hp, 0, h->nV, h->partS, h->st);
    HIPCHECK(h, hipMemcpyAsync(Hp, h->Hp, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    return 0;
}

int dotmi_get_features(dotmi_handle *h, double *A, double *vol, double *mass)
{
    if (!h) return DOTMI_E_INVALID;
    if (A) memcpy(A, h->A.data(), sizeof(double) * h->A.size());
    if (vol) memcpy(vol, h->vol.data(), sizeof(double) * h->vol.size());
    if (mass) memcpy(mass, h->mass.data(), sizeof(double) * h->mass.size());
    return 0;
}

int32_t dotmi_part_size(const dotmi_handle *h, int32_t part)
{
    if (!h || part < 0 || part >= h->nPartsAll) return DOTMI_E_INVALID;
    return 3 * (int32_t)h->partVerts[part].size();
}

int32_t dotmi_padded_size(const dotmi_handle *h) { return h ? h->P.nmax : DOTMI_E_INVALID; }
int64_t dotmi_factor_storage_bytes(const dotmi_handle *h) { return h ? (int64_t)(8 * h->wTotal) : DOTMI_E_INVALID; }

int dotmi_part_matrix(dotmi_handle *h, int32_t part, int inverse, double *Mout, int32_t *l2g)
{
    if (!h || part < h->p0 || part >= h->p1 || !Mout) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (inverse) {
        if (int rc = enter_with_factors(h)) return rc;
    } else if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    const int ls = part - h->p0;
    const int ns = 3 * (int)h->partVerts[part].size();
    const int nmax = h->P.nmax, ntl = nmax / 64;
    double *W = h->P.W;
    double *tmp = nullptr;
    if (!inverse) {
        // rebuild H_s from the resident block-CSR into a scratch copy of the factor storage
        HIPCHECK(h, hipMalloc((void **)&tmp, sizeof(double) * std::max<size_t>(h->wTotal, 64)));
        DevParts Pt = h->P;
        Pt.W = tmp;
        HIPCHECK(h, hipMemsetAsync(tmp, 0, sizeof(double) * h->wTotal, h->st));
        launch_dense_fill(Pt, h->Hval, h->st);
        W = tmp;
    }
    // the subdomain's row blocks (RowTile) lie one after the other in W: copy their span, then read (row, column) through
    // the table -- back into ascending vertex order
    long long lo = -1, hi = -1;
    for (int J = 0; J < ntl; ++J) {
        const long long o = h->rtOff[(size_t)ls * ntl + J];
        if (o < 0) continue;
        if (lo < 0) lo = o;
        hi = o + 64ll * h->rtLd[(size_t)ls * ntl + J];
    }
    std::vector<double> span((size_t)std::max<long long>(hi - lo, 1));
    hipError_t e = lo >= 0 ? hipMemcpyAsync(span.data(), W + lo, sizeof(double) * (size_t)(hi - lo), hipMemcpyDeviceToHost, h->st)
                           : hipSuccess;
    hipStreamSynchronize(h->st);
    if (tmp) hipFree(tmp);
    HIPCHECK(h, e);
    auto at = [&](int r, int c) -> double {   // memory row r, column c; what is not stored is zero
        const size_t k = (size_t)ls * ntl + (r >> 6);
        const long long o = h->rtOff[k];
        const int c0 = h->rtC0[k], ld = h->rtLd[k];
        if (o < 0 || c < c0 || c >= c0 + ld) return 0.0;
        return span[(size_t)(o - lo + (long long)(r & 63) * ld + (c - c0))];
    };
    const auto &pos = h->partPos[ls];
    for (int i = 0; i < ns; ++i)
        for (int j = 0; j < ns; ++j) {
            // memory row r holds row r of X up to the diagonal; the other triangle is not part of X (the tile
            // factorisation leaves the mirror copy of H there, the compact layout does not even store it) -- and H_s
            // itself is read symmetrically from the stored triangle
            const int r = pos[i / 3] + i % 3, c = pos[j / 3] + j % 3;
            Mout[(size_t)i * ns + j] = inverse ? (c <= r ? at(r, c) : 0.0) : at(std::max(r, c), std::min(r, c));
        }
    if (l2g)
        for (size_t i = 0; i < h->partVerts[part].size(); ++i) l2g[i] = h->partVerts[part][i];
    return 0;
}

int dotmi_bench_precond(dotmi_handle *h, int32_t reps, double *ms_per_launch, int64_t *bytes_per_launch)
{
    if (!h || reps < 1) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (int rc = enter_with_factors(h)) return rc;
    launch_gemv(h->P, h->q, h->st);  // warm
    HIPCHECK(h, hipEventRecord(h->ev0, h->st));
    for (int i = 0; i < reps; ++i) launch_gemv(h->P, h->q, h->st);
    HIPCHECK(h, hipEventRecord(h->ev1, h->st));
    HIPCHECK(h, hipEventSynchronize(h->ev1));
    float ms = 0;
    HIPCHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (ms_per_launch) *ms_per_launch = ms / reps;
    if (bytes_per_launch) *bytes_per_launch = h->precond_bytes;
    return 0;
}

// One kernel class of the hot path launched `reps` times back to back on the handle's resident data (warm-up launch
// first), HIP events on the library's stream around them.  bytes = the algorithmic bytes of ONE launch by the formulas
// of SURVEY.md section 8(d) (spelled out per kind below and in DESIGN.md section 4).  The state of the handle is used as
// it is (call between steps); kinds that would disturb it (Hessian refresh) write to their usual buffers, which the next
// refresh overwrites anyway.  Kinds: enum dotmi_bench_kind.
int dotmi_bench_kernel(dotmi_handle *h, int32_t kind, int32_t reps, double *ms_per_launch, int64_t *bytes_per_launch)
{
    if (!h || reps < 1) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (int rc = enter_with_factors(h)) return rc;
    const int n = h->n, nV = h->nV;
    const int64_t nTo = h->PT.nElem, nVo = h->v1 - h->v0, m = h->m > 0 ? h->m : h->hist;
    LbfgsArgs L = lbfgs_args(h);
    L.m = (int)std::min<int64_t>(m, h->hist);   // as in a running step with a full history
    for (int i = 0; i < L.m; ++i) {
        L.s[i] = h->S[i];
        L.y[i] = h->Y[i];
        if (L.ys[i] == 0.0) L.ys[i] = 1.0;
    }
    int nb = 0;
    int64_t bytes = 0;
    bool live = false;   // the kernel form reads the device loop's state
    std::function<void()> run;
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.x = h->x;
    a.xt = h->xt;
    a.g_old = h->g;
    a.p = h->p;
    a.alpha_dev = h->alpha_dev;
    a.g_new = h->g_trial;
    a.s_new = h->S[h->hist];
    a.y_new = h->Y[h->hist];
    a.iv0 = h->v0;
    a.iv1 = h->v1;
    a.make_pair = 1;
    double xi[HIST_MAX] = {0, 0, 0, 0, 0, 0};
    switch (kind) {
    case DOTMI_BENCH_ELEM_ENERGY_GRAD:   // 112 nT + 56 nV: energy evaluation incl. inertia (the gradient entries stay on chip)
        bytes = 112 * nTo + 56 * nVo;
        run = [&] { launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x, h->xt, h->v0, h->v1, 1, h->partE, &nb, h->st); };
        break;
    case DOTMI_BENCH_ELEM_ENERGY:        // 112 nT + 56 nV
        bytes = 112 * nTo + 56 * nVo;
        run = [&] { launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x, h->xt, h->v0, h->v1, 0, h->partE, &nb, h->st); };
        break;
    case DOTMI_BENCH_VERTEX_GATHER:      // g write + x, x~, m read (80 nV) + pair: g_old, p read, s, y write + 2m history vectors
        bytes = 80 * (int64_t)nV + (int64_t)(4 + 2 * L.m) * 8 * n;
        run = [&] { launch_vertex_gather(h->M, h->PT, a, L, h->partR, h->st); };
        break;
    case DOTMI_BENCH_SPMV_DOTS:          // 72 nnzb (full symmetric block rows) + p, g read
        bytes = 72 * (int64_t)h->M.nnzb + 2 * 8 * (int64_t)n;
        run = [&] { launch_spmv_dots(h->M, h->Hval, h->p, h->g, nullptr, h->v0, h->v1, h->partS, h->st); };
        break;
    case DOTMI_BENCH_BACKSOLVE:          // 8 x structural non-zeros of the block-sparse inverse factors
        bytes = h->precond_bytes;
        run = [&] { launch_gemv(h->P, h->q, h->st); };
        break;
    case DOTMI_BENCH_MERGE:              // z write + the tile partials that make it up + m history vectors (y_i . z)
        bytes = 8 * (int64_t)n * (2 + L.m) + 8 * (int64_t)h->mergeEntries;
        // (split form, big meshes: the coalesced reduce of the tile partials is the first half of the merge)
        run = [&] {
            if (!h->P.mt_ptr) launch_reduce_partial(h->P, h->st);
            launch_merge(h->M, h->P, L, h->z, h->partC, 1 | 2, h->st);
        };
        break;
    case DOTMI_BENCH_BUILD_QPAD:         // g + m history vectors read, padded right-hand sides written
        bytes = 8 * (int64_t)n * (1 + L.m) + 8 * (int64_t)h->P.nParts * h->P.nmax;
        run = [&] { launch_build_qpad(h->P, h->g, L, xi, h->st); };
        break;
    case DOTMI_BENCH_BUILD_P:            // z + m history vectors read, p written
        bytes = 8 * (int64_t)n * (2 + L.m);
        run = [&] { launch_build_p(n, h->z, L, h->partC, xi, h->p, h->st); };
        break;
    case DOTMI_BENCH_STEP_FORWARD:       // x, p read, x_trial written
        bytes = 8 * (int64_t)n * 3;
        run = [&] { launch_step_forward(n, h->x, h->p, h->x_trial, h->partS, 0.0, 1, h->alphaMin, h->alpha_dev, nullptr, h->st); };
        break;
    case DOTMI_BENCH_ELEM_HESSIAN:       // 112 nT in, 1152 nT out
        if (h->world > 1) return DOTMI_E_INVALID;   // (the refresh re-issued below ends in a collective: not from one rank alone)
        bytes = (int64_t)(112 + 1152) * h->nHessElems;
        run = [&] {
            if (h->shardHess) launch_elem_hessians(h->M, h->mat, h->dtSq, h->x, h->He, h->st, h->hessElems, h->nHessElems);
            else launch_elem_hessians(h->M, h->mat, h->dtSq, h->x, h->He, h->st);
        };
        break;
    case DOTMI_BENCH_ASSEMBLE:           // 1152 nT in, 72 nnzb out
        if (h->world > 1) return DOTMI_E_INVALID;
        bytes = (int64_t)1152 * h->nHessElems + 72 * (int64_t)(h->shardHess ? h->nHessBlk : h->M.nnzb);
        run = [&] {
            if (h->shardHess) launch_assemble(h->M, h->He, h->Hval, h->st, h->hessBlk, h->nHessBlk, h->hessBlkPtr, h->hessBlkEnt);
            else launch_assemble(h->M, h->He, h->Hval, h->st);
        };
        break;
    // ---- the forms the device loop's early order really launches.  They read the loop state on the
    // device, so the state the last step left there (history full, buffer roles, xi / delta) is switched back to "running,
    // new direction" for the duration of the measurement and restored afterwards; none of them advances it (only the
    // controller does), so every repetition does the same work.  They overwrite loop-internal vectors (p, z, the trial
    // point and gradient, the free pair slot, the padded right-hand sides), all of which the next step rewrites.
    case DOTMI_BENCH_SPMV_ZP:            // 72 nnzb + z, g read, p, Hp written, m pairs of (s_j, H s_j) read
        bytes = 72 * (int64_t)h->M.nnzb + 8 * (int64_t)n * (4 + 2 * L.m);
        live = true;
        run = [&] { launch_spmv_zp(h->M, h->Hval, h->z, h->partC, h->p, h->Hp, h->partS, h->st, h->ctl); };
        break;
    case DOTMI_BENCH_MERGE_EARLY:        // tile partials + u_old read / written, z written, M y_new written, m x (y_j, M y_j) read
        bytes = 8 * (int64_t)h->mergeEntries + 8 * (int64_t)n * (4 + 2 * L.m);
        live = true;
        run = [&] {
            if (!h->P.mt_ptr) launch_reduce_partial(h->P, h->st, h->ctl);
            launch_merge_early(h->M, h->P, h->z, h->partC, 0, h->st, h->ctl);
        };
        break;
    case DOTMI_BENCH_ELEM_STEP: {        // the element pass with the line-search step inside: + p read, trial point written
        bytes = 112 * nTo + 56 * nVo + 48 * (int64_t)nV;
        live = true;
        run = [&] {
            StepArgs sa{h->p, h->partS, h->alpha_dev, h->alphaMin};
            launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x_trial, h->xt, h->v0, h->v1, 1, h->partE, &nb, h->st, h->ctl,
                                    h->tune.fuseStep ? &sa : nullptr);
        };
        break;
    }
    case DOTMI_BENCH_GATHER_EARLY: {     // + H p read, H s_new written, -g into the padded right-hand sides of every holder
        long long held = 0;
        for (int v = 0; v < nV; ++v) held += h->dup[v];
        bytes = 80 * (int64_t)nV + (int64_t)(6 + 2 * L.m) * 8 * n + 24 * held;
        live = true;
        a.x = nullptr;
        a.g_old = nullptr;
        a.g_new = nullptr;
        a.s_new = nullptr;
        a.y_new = nullptr;
        a.hp = h->tune.fuseDir ? h->Hp : nullptr;
        a.vp_ptr = h->P.vp_ptr;
        a.vp_off = h->P.vp_off;
        a.rpad = h->P.rpad;
        run = [&] { launch_vertex_gather(h->M, h->PT, a, L, h->partR, h->st, h->ctl); };
        break;
    }
    default:
        return DOTMI_E_INVALID;
    }
    DevLoop saved;
    if (live) {
        if (!h->earlyBs || h->prevSlots < 0 || !h->P.vp_ptr) {
            h->err = "the in-loop kernel forms need a handle that has run a step of the device loop's early order";
            return DOTMI_E_INVALID;
        }
        HIPCHECK(h, hipMemcpy(&saved, h->ctl, sizeof(DevLoop), hipMemcpyDeviceToHost));
        DevLoop live_ctl = saved;
        live_ctl.status = 0;
        live_ctl.phase = 0;
        HIPCHECK(h, hipMemcpy(h->ctl, &live_ctl, sizeof(DevLoop), hipMemcpyHostToDevice));
    }
    run();   // warm
    HIPCHECK(h, hipEventRecord(h->ev0, h->st));
    for (int i = 0; i < reps; ++i) run();
    HIPCHECK(h, hipEventRecord(h->ev1, h->st));
    HIPCHECK(h, hipEventSynchronize(h->ev1));
    HIPCHECK(h, hipGetLastError());
    if (live) HIPCHECK(h, hipMemcpy(h->ctl, &saved, sizeof(DevLoop), hipMemcpyHostToDevice));
    float ms = 0;
    HIPCHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (kind == DOTMI_BENCH_ELEM_HESSIAN || kind == DOTMI_BENCH_ASSEMBLE) {
        // these two rewrote the element / global Hessians at the CURRENT positions; the factors (and the alpha_0 of the next
        // step) belong to the positions of the last refresh -- bring everything back in line
        if (int rc = refactor(h, h->x, nullptr, nullptr)) return rc;
    }
    if (ms_per_launch) *ms_per_launch = ms / reps;
    if (bytes_per_launch) *bytes_per_launch = bytes;
    return 0;
}

int dotmi_bench_energy(dotmi_handle *h, int32_t reps, double *ms_per_launch, int64_t *bytes_per_launch)
{
    if (!h || reps < 1) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    int nb = 0;
    launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x, h->xt, h->v0, h->v1, 0, h->partE, &nb, h->st);
    HIPCHECK(h, hipEventRecord(h->ev0, h->st));
    for (int i = 0; i < reps; ++i)
        launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x, h->xt, h->v0, h->v1, 0, h->partE, &nb, h->st);
    HIPCHECK(h, hipEventRecord(h->ev1, h->st));
    HIPCHECK(h, hipEventSynchronize(h->ev1));
    float ms = 0;
    HIPCHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (ms_per_launch) *ms_per_launch = ms / reps;
    // 112 B per tet + 56 B per vertex
    if (bytes_per_launch) *bytes_per_launch = (int64_t)112 * h->nOwnElem + (int64_t)56 * (h->v1 - h->v0);
    return 0;
}

}  // extern "C"
