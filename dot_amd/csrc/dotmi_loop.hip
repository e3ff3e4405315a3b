// dotmi_loop.hip -- the L-BFGS-H loop of one time step (DOTTimeStepper::fullyImplicit / solve_oneStep, DOTTimeStepper.cpp:273-504; Optimizer::lineSearch, Optimizer.cpp:752-881): device-resident loop, host loop, GSDD and Newton siblings, dotmi_step
#include "dotmi_handle.hpp"

namespace dotmi {

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

// the operands of the one-launch element pass + gather on vertex patches (k_elemvert.hip)
static ElemVertArgs elem_vertex_args(const dotmi_handle *h)
{
    ElemVertArgs a;
    memset(&a, 0, sizeof(a));
    a.mass = h->M.mass;
    a.xt = h->xt;
    a.p = h->p;
    a.hp = h->Hp;
    a.spmv_partials = h->partST;
    a.fixed = h->M.fixed;
    a.vp_ptr = h->P.vp_ptr;
    a.vp_off = h->P.vp_off;
    a.rpad = h->P.rpad;
    a.partE = h->partE;
    a.partR = h->partR;
    a.alpha_out = h->alpha_dev;
    a.dtSq = h->dtSq;
    a.alpha_min = h->alphaMin;
    return a;
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
    int nb = 0;
    if (h->specNow) {
        // the direction kernel and the first trial's element pass at the unit step in ONE launch (k_dirstep.hip); a retry slot:
        // the element pass alone, on the stored p with the controller's alpha
        StepArgs sa{h->p, h->partS, h->alpha_dev, h->alphaMin};
        launch_dirstep(h->M, h->PTspec, h->mat, h->dtSq, h->xt, h->partE, &nb, h->Hval, h->z, h->partCT, h->p, h->Hp, h->partS, h->st,
                       h->ctl, sa);
    } else if (ow) {
        launch_spmv_zp(h->M, h->HvalOwn, h->z, h->partGC, h->p, h->Hp, h->partS, h->st, h->ctl, 0, -1, h->heldMask, h->ownMask,
                       h->held());
    } else if (fuseDir) {   // build_p + spmv_dots in one launch, H p from the cached H s_j
        // (one rank: the y_i . z partials from their column-major twin, which merge_early writes beside the rows)
        launch_spmv_zp(h->M, h->Hval, h->z, h->dist ? h->partC : h->partCT, h->p, h->Hp, h->partS, h->st, h->ctl, se ? h->v0 : 0,
                       se ? h->v1 : -1, nullptr, nullptr, VList(), !h->dist, h->partST);
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
    // (not on meshes whose workgroups walk several patches: the prefetched operands leave no registers for it)
    // (sharded element pass: the fused form writes the trial point only on this rank's vertex slice -- its inertia loop --,
    // so the step stays a launch of its own there)
    // (owner exchange: the inertia loop runs over every vertex with the owner's share of the mass, so the fused form writes the
    // whole trial point there too -- x + alpha 0 off the held vertices)
    if (h->specNow) {
    } else if (h->vpNow) {
        // vertex patches: the element pass, the step, the gather, the pair and its statistics, -g into the right-hand sides: one launch
        ElemVertArgs ea = elem_vertex_args(h);
        if (h->pairNow) ea.alpha_min = -h->alphaMin;   // (negative: a paired launch, as StepArgs::alpha_min)
        launch_elem_vertex(h->VP, h->mat, ea, h->st, h->ctl);
        nb = h->VP.nPatches;
    } else if (h->tune.fuseStep && (!se || ow)) {   // the step x_trial = x_cur + alpha p inside the element pass
        StepArgs sa{h->p, spart, h->alpha_dev, h->pairNow ? -h->alphaMin : h->alphaMin};   // (negative: a paired launch)
        // (a step with paired trials takes the PAIR instantiations of these two launches)
        (h->pairNow ? launch_elem_energy_grad_pair : launch_elem_energy_grad)(
            ow ? h->Mown : h->M, h->PT, h->mat, h->dtSq, h->x_trial, h->xt, ow ? 0 : h->v0, ow ? h->nV : h->v1, 1, h->partE, &nb,
            h->st, h->ctl, &sa);
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
    if (h->vpNow) {
    } else if (!se) {
        launch_vertex_gather(h->M, h->stepPT(), a, L0, h->partR, h->st, h->ctl);
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
    CtlArgs ca{h->ctl, ctlE, ctlR, h->alpha_dev, h->h_flags, nb, h->pairNow ? 2 : 0};
    (h->pairNow ? (h->vpNow ? launch_gemv_pair_vp : launch_gemv_pair) : h->specNow ? launch_gemv_spec : h->vpNow ? launch_gemv_vp : launch_gemv)(
        h->P, nullptr, h->st, h->ctl, timed ? h->evPre[h->evUsed] : nullptr, timed ? h->evPre[h->evUsed + 1] : nullptr, &ca,
        h->tune.earlyAbort ? (int)h->slotTimed.size() /* the slot's epoch, 1-based */ : (1 << 30) /* never stopped */);
    if (timed) h->evUsed += 2;
    if (!h->dist) {
        launch_merge_early(h->M, h->P, h->z, h->partC, 0, h->st, h->ctl, nullptr, nullptr, VList(), nullptr, 0, nullptr, h->partCT);
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
    // paired trials: one rank, the fused step inside the element pass, the early order with held launches (the tiles of a paired
    // slot wait for the verdict); -1: only in steps that follow a step with halvings in at least a quarter of its iterations
    h->pairNow = h->earlyNow && !h->dist && h->tune.fuseStep && h->tune.fuseDir && h->tune.earlyAbort && h->tune.earlyHold &&
                 (h->tune.pairTrials > 0 || (h->tune.pairTrials < 0 && h->prevIters > 0 && 4 * h->prevHalv >= h->prevIters));
    // the unit step speculatively beside the direction kernel: one rank, the fused kernels, every patch a workgroup, no pairing
    // in this step (a step whose predecessor halved often); -1: only after a step whose first trials took the unit estimate at
    // least nine times in ten (a slot whose estimate is below 1 is redone: ~45 us lost against ~10 saved)
    h->specNow = h->earlyNow && !h->dist && !h->pairNow && h->tune.fuseStep && h->tune.fuseDir && h->tune.earlyAbort &&
                 h->specFits &&
                 (h->tune.specStep > 0 || (h->tune.specStep < 0 && h->prevFirst > 0 && 10 * h->prevUnit >= 9 * h->prevFirst));
    // the trial's element pass + gather as one launch on vertex patches (paired steps too: elem_vertex_kernel<MAT, true>); a step that
    // speculates keeps the element patches (one patch set per step: the start-of-step evaluation and the trials group their
    // energy partials alike)
    // A step that pairs its trials (its predecessor halved in a quarter of its iterations) keeps the element patches unless the
    // vertex patches are forced: a rejected trial costs the whole fused launch there (18 us against the element pass' 10), and the
    // second half of a paired launch is the element pass' energy only -- stiff monkey, ten steps, same box: 15.12 -> 14.17 ms per
    // step (111.8 -> 107.6 us per iteration)
    h->vpNow = h->vpFits && h->earlyNow && !h->dist && !h->specNow && !(h->pairNow && h->tune.vertexPatches < 0);
    C.specPartials = h->partS;
    C.alphaMin = h->alphaMin;
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
    devloop_resolve(C);   // (slot 0, no pair yet)
    C.holdEnable = h->tune.earlyHold && h->tune.earlyAbort ? 1 : 0;
    // the forecast carries over from the last step (a function of the handle's own history)
    memcpy(C.predHist, h->predState, sizeof(int) * 2);
    memcpy(&C.predCtr[0][0], h->predState + 2, sizeof(int) * 8);
    memcpy(C.pairCtr, h->pairState, sizeof(int) * 3);
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
        if (h->vpNow) {
            // energy, gradient, |g|^2 and -g_0 in the padded right-hand sides from the one launch (no step, no pair: ctl == nullptr)
            ElemVertArgs ea = elem_vertex_args(h);
            ea.x0 = h->x;
            ea.g0 = h->g;
            launch_elem_vertex(h->VP, h->mat, ea, h->st, nullptr);
            nb = h->VP.nPatches;
        } else
        launch_elem_energy_grad(h->owner ? h->Mown : h->M, h->stepPT(), h->mat, h->dtSq, h->x, h->xt, h->owner ? 0 : h->v0,
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
        if (!h->vpNow) launch_vertex_gather(h->M, h->stepPT(), a, L0, h->partR, h->st);
        if (!h->shardElems && h->earlyNow) {
            // the first direction's solve, u = -M g_0 and z = u, with the start-of-step controller inside its launch
            CtlArgs ca{h->ctl, h->partE, h->partR, h->alpha_dev, h->h_flags, nb, 1};
            (h->vpNow ? launch_gemv_vp : launch_gemv)(h->P, nullptr, h->st, h->ctl, nullptr, nullptr, &ca, 1 << 30);
            if (!h->dist) {
                launch_merge_early(h->M, h->P, h->z, h->partC, 1, h->st, h->ctl, nullptr, nullptr, VList(), nullptr, 0, nullptr, h->partCT);
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
    memcpy(h->pairState, C.pairCtr, sizeof(int) * 3);
    h->heldSlots = C.heldSlots;
    h->heldRejected = C.heldRejected;
    h->pairSlots = C.pairSlots;
    h->pairRedo = C.pairRedo;
    h->specSlots = C.specSlots;
    h->specRedo = C.specRedo;
    h->prevFirst = C.firstTrials;
    h->prevUnit = C.unitFirst;
    if (h->tune.fuseLog && C.pairSlots)
        fprintf(stderr, "dotmi: paired trials: %d slots, %d of them redone (the full step was acceptable)\n", C.pairSlots, C.pairRedo);
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

}  // namespace dotmi

extern "C" {

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
        // (a paired slot whose full step was rejected takes a halving without a stopped launch; one that is redone stops without one)
        if (h->devLoop && h->pairNow) st->backsolve_stopped += 2 * h->pairRedo - h->pairSlots;
        if (h->devLoop && h->specNow) st->backsolve_stopped += h->specRedo;   // (a redone slot's back-solve stops without a halving)
        st->spec_slots = (h->devLoop && h->specNow) ? h->specSlots : 0;
        st->spec_redone = (h->devLoop && h->specNow) ? h->specRedo : 0;
        st->backsolve_held = (h->devLoop && h->earlyNow) ? h->heldSlots : 0;
        st->backsolve_held_rejected = (h->devLoop && h->earlyNow) ? h->heldRejected : 0;
        st->paired_slots = (h->devLoop && h->pairNow) ? h->pairSlots : 0;
        st->paired_redone = (h->devLoop && h->pairNow) ? h->pairRedo : 0;
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

}  // extern "C"
