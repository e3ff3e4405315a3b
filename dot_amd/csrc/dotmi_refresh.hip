// dotmi_refresh.hip -- the refresh at the end of a step (DOTTimeStepper::updateHessianAndFactor, DOTTimeStepper.cpp:349-380): element Hessians, assembly, fill of the work buffer, the tile factorisation; its asynchronous form
#include "dotmi_handle.hpp"

namespace dotmi {

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
        launch_twolevel_pack(h->P, h->st);
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
    // a dataflow wait that timed out anywhere is a DEVICE failure, whatever pivot report stands in front of it
    int stuck = -1;
    for (int i = 0; i < h->P.nParts && stuck < 0; ++i)
        if (h->h_info[i] >= (1 << 30)) stuck = i;
    if (h->world > 1) {
        // all ranks fail together -- a rank that returned alone would leave the others blocked in the next collective -- and with
        // the SAME error class: both flags are reduced, every rank classifies from the sums
        double f[2] = {bad >= 0 ? 1.0 : 0.0, stuck >= 0 ? 1.0 : 0.0};
        HIPCHECK(h, hipMemcpyAsync(h->ctrlDev, f, sizeof(f), hipMemcpyHostToDevice, h->st));
        if (int rc = allreduce_sum(h, h->ctrlDev, 2)) return rc;
        HIPCHECK(h, hipMemcpyAsync(f, h->ctrlDev, sizeof(f), hipMemcpyDeviceToHost, h->st));
        HIPCHECK(h, hipStreamSynchronize(h->st));
        if (f[1] > 0.0 && stuck < 0) {
            h->err = "the tile factorisation's dataflow scheduler timed out on another rank";
            h->poisoned = true;
            return DOTMI_E_DEVICE;
        }
        if (f[0] > 0.0 && bad < 0) {
            h->err = "a subdomain Hessian on another rank is not positive definite";
            h->poisoned = true;
            return DOTMI_E_NOTSPD;
        }
    }
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
int resolve_refresh(dotmi_handle *h, double *ms_hess, double *ms_fact)
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

}  // namespace dotmi

