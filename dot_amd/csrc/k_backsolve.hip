// k_backsolve.hip -- subdomain back-solve (DOTTimeStepper.cpp:406-431) and the loop controller (Optimizer.cpp:806-833, DOTTimeStepper.cpp:474-494)
// (one translation unit per kernel family since round 6: an edit to one family no longer moves the register allocation and
// scalar loads of the others; every unit is compiled once.  Conventions and the reference map: k_device.hpp)
#include "k_device.hpp"

namespace dotmi {

// ------------------------------------------------------------------------------------------------
// loop controller: what the host loop of dotmi_step does between a trial and the next launch
// (line search Optimizer.cpp:806-833, history update DOTTimeStepper.cpp:474-494, stopping test
// Optimizer.cpp:317-330), on the device.  One wavefront; the partial sums are added in block order,
// the same order the host path uses, so both paths produce the same bits.
// ------------------------------------------------------------------------------------------------
// (a device function: the controller is a launch of its own, or workgroup 0 of the back-solve launch -- 256 threads)
// CTL_PAIR: the controller of a step with paired line-search trials (it understands paired slots: the full step's energy partials
// in partE2, alpha_dev[1] > 0 marks a paired slot).  CTL_SPEC: the controller of a step whose new-direction slots take the unit
// step speculatively (k_dirstep.hip): it sums the SpMV partials itself -- the element pass's prologue, the same additions -- and
// when alpha_0 = clamp(-p.g / p.Hp, alphaMin, 1) is not 1 the slot did not evaluate the reference's first trial: its back-solve is
// stopped and the next slot evaluates x + alpha_0 p as the first trial it is (phase 1, redo: nothing counted twice).  Template
// parameters: the plain instantiation compiles those branches away (the pairing until round 5 through a second compilation of the
// file under -DDOTMI_PAIR_TU).
// CTL_VP: the controller of a step on vertex patches (k_elemvert.hip): the statistics come in one row per PATCH (nbE rows, like the
// energy partials) instead of NB_RED rows; summed in chunked_sum's order over that many rows.
// CTL_PAIR_VP: both -- a step with paired trials on vertex patches.
constexpr int CTL_PLAIN = 0, CTL_PAIR = 1, CTL_SPEC = 2, CTL_VP = 3, CTL_PAIR_VP = 4;
template <int CTL>
__device__ __forceinline__ void loop_control_body(DevLoop *__restrict__ ctl, const double *__restrict__ partE, int nbE,
                                  const double *__restrict__ partR, const double *__restrict__ alpha_dev,
                                  int *__restrict__ flags_host, int init, const double *__restrict__ partE2 = nullptr)
{
    constexpr bool PAIR = CTL == CTL_PAIR || CTL == CTL_PAIR_VP, SPEC = CTL == CTL_SPEC;
    constexpr bool VPROWS = CTL == CTL_VP || CTL == CTL_PAIR_VP;   // one row of statistics per patch (nbE rows)
    static_assert(sizeof(DevLoop) % 8 == 0, "DevLoop is copied as 8-byte words");
    static_assert(RED_K <= 32, "two passes of 16 columns");
    __shared__ double chunk[RED_K + 2][SUM_CHUNKS];
    __shared__ double R[RED_K + 2];
    __shared__ DevLoop C;  // the state is worked on in LDS: one wide load, one wide store
    const int t = threadIdx.x;
    constexpr int NW8 = (int)(sizeof(DevLoop) / 8);
    {
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(ctl);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&C);
        for (int i = t; i < NW8; i += 256) dst[i] = src[i];
    }
    const double alpha_in = *alpha_dev;
    // SPEC: the partials of p.g and p.Hp, requested with everything else (wave 0; summed below as elem_patch_body's prologue does)
    double pgv[NB_RED / 64], pHpv[NB_RED / 64];
    if constexpr (SPEC) {
        if (t < 64 && !init) {
            const double *__restrict__ sp = ctl->specPartials;
#pragma unroll
            for (int u = 0; u < NB_RED / 64; ++u) {
                pgv[u] = sp[(size_t)(t + 64 * u) * RED_K];
                pHpv[u] = sp[(size_t)(t + 64 * u) * RED_K + 1];
            }
        }
    }
    const double alpha_full = (PAIR && partE2 && !init) ? alpha_dev[1] : 0.0;   // > 0: a paired slot (elem_patch_kernel)
    __shared__ double chunk2[PAIR ? 2 : 1][SUM_CHUNKS];
    // chunked_sum() order (dotmi_internal.hpp), one thread per (column, chunk): every load of the kernel is in flight at
    // once and the dependent add chains are 16 long instead of NB_RED long.  The column index runs fastest over the
    // lanes, so a load instruction touches a few 168-byte partial rows instead of 64 different ones.
    {
        constexpr int LR = (NB_RED + SUM_CHUNKS - 1) / SUM_CHUNKS;
        static_assert(LR * SUM_CHUNKS == NB_RED, "chunks of equal length");
        constexpr int NPAIR = RED_K * SUM_CHUNKS;            // (statistic column, chunk) pairs
        static_assert(NPAIR + 2 * SUM_CHUNKS <= 512, "two passes of 256 threads");
        const int qa = t, qb = t + 256;
        const int colA = qa % RED_K, chA = qa / RED_K;
        const int colB = qb % RED_K, chB = qb / RED_K;
        const bool hasA = qa < NPAIR, hasB = qb < NPAIR;
        double va[LR], vb[LR];
        if constexpr (!VPROWS) {
            if (hasA) {
#pragma unroll
                for (int k = 0; k < LR; ++k) va[k] = partR[(size_t)(chA * LR + k) * RED_K + colA];
            }
            if (hasB) {
#pragma unroll
                for (int k = 0; k < LR; ++k) vb[k] = partR[(size_t)(chB * LR + k) * RED_K + colB];
            }
        }
        const int te = qb - NPAIR;  // the next 2 * SUM_CHUNKS slots: the two energy columns
        if (te >= 0 && te < 2 * SUM_CHUNKS) {
            const int LE = (nbE + SUM_CHUNKS - 1) / SUM_CHUNKS, c = te / SUM_CHUNKS, ch = te % SUM_CHUNKS;
            const int k0 = ch * LE, k1 = min(nbE, (ch + 1) * LE);
            double e = 0.0;
            int k = k0;
            for (; k + 8 <= k1; k += 8) {
                double ve[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) ve[u] = partE[2 * (k + u) + c];
#pragma unroll
                for (int u = 0; u < 8; ++u) e += ve[u];
            }
            for (; k < k1; ++k) e += partE[2 * k + c];
            chunk[RED_K + c][ch] = e;
            if (PAIR && alpha_full > 0.0) {   // the same chunks of the full step's partials (same order: the same bits as a plain slot)
                double e2 = 0.0;
                int k2 = k0;
                for (; k2 + 8 <= k1; k2 += 8) {
                    double ve[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) ve[u] = partE2[2 * (k2 + u) + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) e2 += ve[u];
                }
                for (; k2 < k1; ++k2) e2 += partE2[2 * k2 + c];
                chunk2[PAIR ? c : 0][ch] = e2;
            }
        }
        if constexpr (VPROWS) {
            // nbE rows of statistics: chunk ch of column col = rows [ch LE, (ch + 1) LE), eight loads in flight
            auto chunk_rows = [&](int col, int ch) {
                const int LE = (nbE + SUM_CHUNKS - 1) / SUM_CHUNKS;
                const int k0 = ch * LE, k1 = min(nbE, (ch + 1) * LE);
                double s = 0.0;
                int k = k0;
                for (; k + 8 <= k1; k += 8) {
                    double v8[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v8[u] = partR[(size_t)(k + u) * RED_K + col];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v8[u];
                }
                for (; k < k1; ++k) s += partR[(size_t)k * RED_K + col];
                return s;
            };
            if (hasA) chunk[colA][chA] = chunk_rows(colA, chA);
            if (hasB) chunk[colB][chB] = chunk_rows(colB, chB);
        } else {
        if (hasA) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < LR; ++k) a += va[k];
            chunk[colA][chA] = a;
        }
        if (hasB) {
            double b = 0.0;
#pragma unroll
            for (int k = 0; k < LR; ++k) b += vb[k];
            chunk[colB][chB] = b;
        }
        }
    }
    double a0spec = 1.0;   // SPEC: alpha_0 of the direction the slot computed (valid in thread 0)
    if constexpr (SPEC) {
        if (t < 64 && !init) {
            double pg = 0.0, pHp = 0.0;
#pragma unroll
            for (int u = 0; u < NB_RED / 64; ++u) {
                pg += pgv[u];
                pHp += pHpv[u];
            }
            pg = wave_sum(pg);
            pHp = wave_sum(pHp);
            a0spec = fmax(ctl->alphaMin, fmin(1.0, -pg / pHp));  // Optimizer.cpp:1085
        }
    }
    __syncthreads();
    if (C.status != 0) return;
    if (t < RED_K + 2) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < SUM_CHUNKS; ++c) acc += chunk[t][c];
        R[t] = acc;
    }
    __syncthreads();
    if (t == 0 && init) {
        // evaluation at the start of the step (DOTTimeStepper.cpp:299): nothing to decide yet
        const double E = C.dtSq * R[RED_K] + R[RED_K + 1];
        C.evals++;
        C.E_cur = C.E0 = E;
        C.g2_cur = C.g2_0 = R[0];
    } else if (t == 0) {
        double alpha = alpha_in;
        const double E = C.dtSq * R[RED_K] + R[RED_K + 1];
        int kind = 0;
        bool decided = false;
        if constexpr (PAIR) {
            if (C.slots < C.kindCap) C.slot_kind[C.slots] = C.phase == 0 ? 1 : 2;
            C.slots++;
            kind = (C.phase == 0 || C.redo) ? 0 : 1;   // first trial of an iteration / retry after a halving
            // what a first trial with alpha_0 < 1 teaches the pairing rule (a redone trial has taught it already)
            const double a0 = alpha_full > 0.0 ? alpha_full : alpha_in;
            const bool learns = C.phase == 0 && a0 < 1.0;
            if (C.phase == 0) {
                C.firstTrials++;
                C.unitFirst += a0 == 1.0 ? 1 : 0;
            }
            C.redo = 0;
            if (alpha_full > 0.0) {
                // Paired slot: the energy of the FULL step first, as the reference's line search would see it
                double ea = 0.0, ei = 0.0;
#pragma unroll
                for (int c = 0; c < SUM_CHUNKS; ++c) {
                    ea += chunk2[0][c];
                    ei += chunk2[PAIR ? 1 : 0][c];
                }
                const double EA = C.dtSq * ea + ei;
                C.pairSlots++;
                {
                    int &pc = C.pairCtr[pair_band(alpha_full)];
                    pc = (EA > C.E_cur) ? min(3, pc + 1) : max(0, pc - 1);
                }
                if (!(EA > C.E_cur)) {
                    // the full step is acceptable: the slot's gradient belongs to the half step and is of no use.  The next slot
                    // evaluates the full step as a plain trial (its energy is counted there); nothing else has happened
                    C.pairRedo++;
                    C.abortEpoch = C.slots;
                    __hip_atomic_store(&ctl->abortEpoch, C.slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    C.holdVerdict = 2 * C.slots + 1;
                    __hip_atomic_store(&ctl->holdVerdict, C.holdVerdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    C.phase = 1;
                    C.alpha = alpha_full;
                    C.redo = 1;
                    decided = true;
                } else {
                    // rejected: one halving; the trial in front of the controller is the retry with the half step
                    C.evals++;
                    C.halvings++;
                    int &ctr = C.predCtr[kind][C.predHist[kind] & 3];
                    ctr = min(3, ctr + 1);
                    C.predHist[kind] = ((C.predHist[kind] << 1) | 1) & 3;
                    kind = 1;
                }
            }
            if (!decided) C.evals++;
            C.heldSlots += (C.holdNext || alpha_full > 0.0) ? 1 : 0;
            if (learns && alpha_full == 0.0) {
                int &pc = C.pairCtr[pair_band(a0)];
                pc = (E > C.E_cur && alpha > 0.0) ? min(3, pc + 1) : max(0, pc - 1);
            }
        } else if constexpr (SPEC) {
            if (C.slots < C.kindCap) C.slot_kind[C.slots] = C.phase == 0 ? 1 : 2;
            C.slots++;
            kind = (C.phase == 0 || C.redo) ? 0 : 1;   // first trial of an iteration (a redone slot is one) / retry after a halving
            C.redo = 0;
            if (C.phase == 0) {
                // the slot computed a new direction and evaluated x + 1 p beside it
                C.firstTrials++;
                C.unitFirst += a0spec == 1.0 ? 1 : 0;
                C.specSlots++;
                if (a0spec != 1.0) {
                    // the estimate lies inside (alphaMin, 1): the unit step is not the line search's first trial.  Nothing has
                    // happened yet: the slot's back-solve stops, the next slot evaluates x + alpha_0 p (its energy is counted there)
                    C.specRedo++;
                    C.abortEpoch = C.slots;
                    __hip_atomic_store(&ctl->abortEpoch, C.slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    C.holdVerdict = 2 * C.slots + 1;
                    __hip_atomic_store(&ctl->holdVerdict, C.holdVerdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    C.phase = 1;
                    C.alpha = a0spec;
                    C.redo = 1;
                    decided = true;
                }
            }
            if (!decided) C.evals++;
            C.heldSlots += C.holdNext;
        } else {
            C.evals++;
            if (C.slots < C.kindCap) C.slot_kind[C.slots] = C.phase == 0 ? 1 : 2;
            C.slots++;
            kind = C.phase == 0 ? 0 : 1;   // first trial of an iteration / retry after a halving
            if (C.phase == 0) {   // (the gate of the next step's speculation: how often the estimate is the unit step)
                C.firstTrials++;
                C.unitFirst += alpha_in == 1.0 ? 1 : 0;
            }
            C.heldSlots += C.holdNext;
        }
        if (decided) {
        } else if (E > C.E_cur && alpha > 0.0) {
            // back-tracking (c1 = 0, lower bound 0)
            // a speculative back-solve on this trial's gradient may be running beside this workgroup: tell it to stop
            C.abortEpoch = C.slots;
            __hip_atomic_store(&ctl->abortEpoch, C.slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            C.holdVerdict = 2 * C.slots + 1;   // ... and one that has been waiting for the verdict to leave
            __hip_atomic_store(&ctl->holdVerdict, C.holdVerdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            {
                int &ctr = C.predCtr[kind][C.predHist[kind] & 3];
                ctr = min(3, ctr + 1);
                C.predHist[kind] = ((C.predHist[kind] << 1) | 1) & 3;
            }
            C.heldRejected += (C.holdNext || alpha_full > 0.0) ? 1 : 0;
            alpha /= 2.0;
            C.halvings++;
            if (alpha == 0.0) {
                // the step length underflowed: the reference stops at the last trial point (Optimizer.cpp:819-861)
                C.status = 3;
                double *tmp = C.x_cur;
                C.x_cur = C.x_trial;
                C.x_trial = tmp;
                C.E_cur = E;
            } else {
                C.phase = 1;
                C.alpha = alpha;
            }
        } else {
            double *tmp = C.x_cur;
            C.x_cur = C.x_trial;
            C.x_trial = tmp;
            tmp = C.g_cur;
            C.g_cur = C.g_trial;
            C.g_trial = tmp;
            C.E_cur = E;
            const double g2 = R[0];
            C.g2_cur = g2;
            const bool last = C.iter + 1 >= C.iterCap || !(g2 > C.tol);
            if (last) {
                // the loop ends with this iterate: a speculative back-solve for the next direction (running beside this
                // workgroup) may stop -- said before the history update below
                C.abortEpoch = C.slots;
                __hip_atomic_store(&ctl->abortEpoch, C.slots, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            C.holdVerdict = 2 * C.slots + (last ? 1 : 0);   // a held back-solve may start now (or leave)
            __hip_atomic_store(&ctl->holdVerdict, C.holdVerdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            {
                int &ctr = C.predCtr[kind][C.predHist[kind] & 3];
                ctr = max(0, ctr - 1);
                C.predHist[kind] = (C.predHist[kind] << 1) & 3;
            }
            // The history update and the first half of the two-loop below are the host loop's statements
            // (dotmi_step) with every array held in registers: all loops are unrolled to HIST_MAX with guards, so
            // nothing is indexed dynamically and no LDS round trip sits in the dependent chain.  The operations
            // and their order are the host's, so the two loops stay bit-identical.
            constexpr int H = HIST_MAX;
            const double ys_new = R[1], sg_new = R[2];
            double siy[H], snyj[H], sig[H], ys[H], b[H], sy[H][H], xi[H];
            int order[H];
#pragma unroll
            for (int i = 0; i < H; ++i) {
                siy[i] = R[3 + i];
                snyj[i] = R[3 + H + i];
                sig[i] = R[3 + 2 * H + i];
                ys[i] = C.L.ys[i];
                b[i] = C.b[i];
                order[i] = C.order[i];
#pragma unroll
                for (int jj = 0; jj < H; ++jj) sy[i][jj] = C.L.sy[i][jj];
            }
            int m = C.L.m;
            const int hist = C.hist, newslot = C.slot;
            C.pairNew = ys_new > 0.0 ? 1 : 0;
            if (ys_new > 0.0) {
                int off = 0;
                if (m == hist) {  // drop the oldest pair
                    off = 1;
#pragma unroll
                    for (int i = 0; i + 1 < H; ++i) {
                        order[i] = order[i + 1];
                        ys[i] = ys[i + 1];
#pragma unroll
                        for (int jj = 0; jj + 1 < H; ++jj) sy[i][jj] = sy[i + 1][jj + 1];
                    }
                    m -= 1;
                }
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    // value i + off of the three statistic rows
                    const double a_siy = (off && i + 1 < H) ? siy[i + 1 < H ? i + 1 : i] : siy[i];
                    const double a_sny = (off && i + 1 < H) ? snyj[i + 1 < H ? i + 1 : i] : snyj[i];
                    const double a_sig = (off && i + 1 < H) ? sig[i + 1 < H ? i + 1 : i] : sig[i];
                    if (i < m) {
#pragma unroll
                        for (int jj = 0; jj < H; ++jj)
                            if (jj == m) {
                                sy[i][jj] = a_siy;   // sy[i][m]
                            }
#pragma unroll
                        for (int ii = 0; ii < H; ++ii)
                            if (ii == m) sy[ii][i] = a_sny;   // sy[m][i]
                        b[i] = a_sig;
                    }
                }
#pragma unroll
                for (int i = 0; i < H; ++i)
                    if (i == m) {
                        order[i] = newslot;
                        ys[i] = ys_new;
                        sy[i][i] = ys_new;
                        b[i] = sg_new;
                    }
                m += 1;
            } else {
#pragma unroll
                for (int i = 0; i < H; ++i)
                    if (i < m) b[i] = sig[i];
            }
            if (C.iter < C.logCap) {
                C.log_alpha[C.iter] = alpha;
                C.log_E[C.iter] = E;
                C.log_g2[C.iter] = g2;
            }
            C.iter++;
#pragma unroll
            for (int i = 0; i < H; ++i) xi[i] = 0.0;
            int fs = C.slot;
            if (C.iter >= C.iterCap) C.status = 2;
            else if (!(g2 > C.tol)) C.status = 1;
            if (C.status == 0) {
                // next direction: first half of the two-loop, free slot
#pragma unroll
                for (int i = H - 1; i >= 0; --i)
                    if (i < m) {
                        double sq = -b[i];
#pragma unroll
                        for (int jj = H - 1; jj > i; --jj)
                            if (jj < m) sq -= xi[jj] * sy[i][jj];
                        xi[i] = sq / ys[i];
                    }
                fs = 0;
                bool found = false;
#pragma unroll
                for (int sl = 0; sl <= H; ++sl) {
                    bool used = false;
#pragma unroll
                    for (int i = 0; i < H; ++i) used |= (i < m && order[i] == sl);
                    if (!found && sl <= hist && !used) {
                        fs = sl;
                        found = true;
                    }
                }
                C.phase = 0;
            }
            // back to the shared copy (stores only; the operand views take their pointers from the slot table)
            C.L.m = m;
            C.slot = fs;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                C.L.ys[i] = ys[i];
                C.b[i] = b[i];
                C.order[i] = order[i];
                C.X.xi[i] = xi[i];
#pragma unroll
                for (int jj = 0; jj < H; ++jj) C.L.sy[i][jj] = sy[i][jj];
                if (i < m) {
                    C.L.s[i] = C.S[order[i]];
                    C.L.y[i] = C.Y[order[i]];
                }
            }
            devloop_resolve(C);   // (S[slot], MY[order[i]], HS[order[i]]: the kernels of the next slot read them resolved)
        }
    }
    // forecast for the slot that follows: hold its back-solve if its kind's counter for the current pattern says "rejected"
    if (t == 0 && !init) {
        const int nk = C.phase == 0 ? 0 : 1;
        C.holdNext = (C.holdEnable && C.status == 0 && !((PAIR || SPEC) && C.redo) && C.predCtr[nk][C.predHist[nk] & 3] >= 2) ? 1 : 0;
    }
    __syncthreads();
    {
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(ctl);
        const unsigned long long *src = reinterpret_cast<const unsigned long long *>(&C);
        for (int i = t; i < NW8; i += 256) dst[i] = src[i];
    }
    // a store to host memory holds the kernel's end back by several microseconds: only near the expected
    // end of the loop, where the host needs the progress to stop enqueueing
    if (t == 0 && !init && (C.status != 0 || C.slots >= C.notifyFrom)) {
        __threadfence_system();
        flags_host[1] = C.slots;
        flags_host[0] = C.status;
    }
}

__global__ __launch_bounds__(256) void loop_control_kernel(DevLoop *__restrict__ ctl,
                                                           const double *__restrict__ partE, int nbE,
                                                           const double *__restrict__ partR,
                                                           const double *__restrict__ alpha_dev,
                                                           int *__restrict__ flags_host, int init)
{
    loop_control_body<CTL_PLAIN>(ctl, partE, nbE, partR, alpha_dev, flags_host, init);
}

void launch_loop_control(DevLoop *ctl, const double *partE, int nbE, const double *partR,
                         const double *alpha_dev, int *flags_host, hipStream_t st, int init)
{
    hipLaunchKernelGGL(loop_control_kernel, dim3(1), dim3(256), 0, st, ctl, partE, nbE, partR, alpha_dev, flags_host,
                       init);
}

__global__ __launch_bounds__(256) void gather_pad_kernel(int total, const int *__restrict__ dofmap,
                                                         const double *__restrict__ q, double *__restrict__ rpad)
{
    const int stride = gridDim.x * blockDim.x;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        const int d = dofmap[k];
        rpad[k] = d >= 0 ? q[d] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// subdomain back-solve  p_s = H_s^-1 r_s = X^T (X r_s),  X = chol(H_s)^-1  (lower triangular)
//   -- THE HBM-bound kernel of the L-BFGS loop.
// Storage: memory row i holds row i of X, X(i,k) for k <= i, contiguously (zeros for k > i); that is
//   the column-major upper factor Q = R^-1 of H = R^T R that chol_inv_tree() produces.  In the
//   nested-dissection order of the subdomain (nd_layout.hpp) row i is non-zero only from the first
//   column of its tree node on, so X is block-sparse.
// One pass: for every memory row i   t_i = row_i . r   and then   p += t_i * row_i
//   so each stored entry is read from HBM exactly ONCE per back-solve:
//   algorithmic bytes per launch = 8 x the structural non-zeros of all X_s (dotmi_step_stats.precond_bytes).
// A workgroup owns up to BS_ROWS consecutive rows of one tree region of one subdomain and walks them a
// few at a time: the rows sit in VGPRs (16 B per lane per row chunk), their dot products are combined
// with a transposed butterfly (10 shuffles instead of 48) + one LDS exchange, and the rank-k update of
// p is applied from the same registers.  Loads stop at the 128-byte line of each row's diagonal and
// skip the identity-padding columns.  The workgroup's partial p goes to ppart[s][tile][.]; the tiles
// of a subdomain are summed in fixed order by reduce_partial_p_kernel (no atomics, deterministic).
// ------------------------------------------------------------------------------------------------
constexpr int BS_ROWS = 64;   // memory rows per workgroup
typedef double nt_double2 __attribute__((ext_vector_type(2)));

// One tile with rows of at most 2*THREADS*MAXCH columns, SUB rows in registers at a time.  Short rows
// (the leaves of the dissection) take many rows per pass, long rows few, so that every pass has about
// the same number of bytes in flight: the pass count of a tile -- a chain of HBM latency, butterfly
// and LDS exchange -- is what bounds a tile, not its byte count.
// RLDS (round 5, rows of 2561 .. 3072 columns on the 256-thread kernel): the thread's right-hand side entries live in LDS
// (rs: each thread reads back only what it wrote -- a manual spill of 24 registers) so that six chunks of rows fit the
// register tile at two workgroups per CU; the dot products keep their order of additions (chunk after chunk).
template <int THREADS, int MAXCH, int SUB, bool RLDS = false>
__device__ __forceinline__ void backsolve_tile(const int4 jb, const int *__restrict__ dofmap,
                                               const double *__restrict__ W, int nmax, const RowTile *__restrict__ rt,
                                               const double *__restrict__ q, double *__restrict__ ppart,
                                               int nbmax, double (*sm)[THREADS / 64][32],
                                               const int *__restrict__ abortp = nullptr, int epoch = 0,
                                               int *s_abort = nullptr, double2 *__restrict__ rs = nullptr)
{
    constexpr int NW = THREADS / 64;
    const int s = jb.x, i0 = jb.y, tileIdx = jb.z & 0xffff, cb = jb.w;
    const int ns = i0 + (jb.z >> 16);          // one past the last live row of this tile
    // columns cb <= k < ns can be non-zero in these rows (nested dissection: everything left of the
    // tile's node is structurally zero); whole 128-byte lines are loaded
    const int ncol = min((ns + 15) & ~15, nmax);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // the tile's rows lie in ONE 64-row block of the factor storage (dotmi_internal.hpp RowTile): row i, column c is at
    // W[rt.off + (i - first row of the block) * rt.ld + (c - rt.c0)]
    const RowTile rb = rt[(size_t)s * (nmax >> 6) + (i0 >> 6)];
    const int ldw = rb.ld;
    const double *Ws = W + (rb.off - (long long)(i0 & ~63) * ldw - rb.c0);
    const int *dm = dofmap + (size_t)s * nmax;
    const double *rp = q + (size_t)s * nmax;   // right-hand side in padded order (zeros on the padding)
    double2 r[RLDS ? 1 : MAXCH], pacc[RLDS ? 1 : MAXCH];   // (RLDS: both in LDS, rs[.] and rs[THREADS * MAXCH + .])
    int cend[MAXCH];  // first column this thread's pair of chunk m is NOT loaded for: 0 for identity-padding columns
#pragma unroll
    for (int m = 0; m < MAXCH; ++m) {
        const int c = cb + 2 * tid + 2 * THREADS * m;
        const int2 dd = (c < ncol) ? *reinterpret_cast<const int2 *>(dm + c) : make_int2(-1, -1);
        const int d0 = dd.x, d1 = dd.y;
        const double2 rv = (c < ncol) ? *reinterpret_cast<const double2 *>(rp + c) : make_double2(0.0, 0.0);
        if constexpr (RLDS) {
            rs[tid + THREADS * m] = rv;
            rs[THREADS * MAXCH + tid + THREADS * m] = make_double2(0.0, 0.0);
        } else {
            r[m] = rv;
            pacc[m] = make_double2(0.0, 0.0);
        }
        // padding columns of live rows hold zeros (separator rows span the padding of every region): not read
        cend[m] = (d0 >= 0 || d1 >= 0) ? c : 0x7fffffff;
    }
#pragma unroll 1
    for (int sb = 0; sb < BS_ROWS / SUB; ++sb) {
        const int ib = i0 + sb * SUB;
        if (ib >= ns) break;
        // speculative launch: has the controller (workgroup 0 of the launch) rejected the trial meanwhile?  One thread asks,
        // the answer is shared through LDS behind this pass' barrier (double-buffered like sm), so the whole workgroup
        // leaves together
        // (requested here, in front of the pass' row loads; stored to LDS only next to the dot products, so that the
        // asking wave does not wait for the answer before it issues its rows)
        int abortSeen = 0;
        if (abortp && tid == 0) abortSeen = __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        double2 y[SUB][MAXCH];
#pragma unroll
        for (int rr = 0; rr < SUB; ++rr) {
            const double *row = Ws + (long long)min(ib + rr, ns - 1) * ldw;
            // a row is zero right of its diagonal: stop at the end of its own 128-byte line, not of the tile
            const int rend = ((ib + rr) < ns) ? min(ncol, (ib + rr + 16) & ~15) : 0;
#pragma unroll
            for (int m = 0; m < MAXCH; ++m) {
                const int c = cb + 2 * tid + 2 * THREADS * m;
                if (cend[m] < rend) {
                    // streamed once per launch by exactly one workgroup: non-temporal, so the 229 MB of factors do not
                    // push the small hot arrays of the other loop kernels out of the 256 MB Infinity Cache
                    const nt_double2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_double2 *>(row + c));
                    y[rr][m] = make_double2(v.x, v.y);
                } else {
                    y[rr][m] = make_double2(0.0, 0.0);
                }
            }
        }
        const int buf = sb & 1;
#pragma unroll
        for (int g = 0; g < SUB / 8; ++g) {
            double d[8];
            if constexpr (RLDS) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) d[rr] = 0.0;
#pragma unroll
                for (int m = 0; m < MAXCH; ++m) {
                    const double2 rv = rs[tid + THREADS * m];
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) d[rr] += y[8 * g + rr][m].x * rv.x + y[8 * g + rr][m].y * rv.y;
                }
            } else {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < MAXCH; ++m) acc += y[8 * g + rr][m].x * r[m].x + y[8 * g + rr][m].y * r[m].y;
                d[rr] = acc;
            }
            }
            // transposed butterfly: 8 values over 64 lanes -> lane group lane>>3 holds one row's wave sum
            double e4[4], e2[2], e1;
            {
                const bool hi = lane & 32;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double keep = hi ? d[k + 4] : d[k], send = hi ? d[k] : d[k + 4];
                    e4[k] = keep + __shfl_xor(send, 32, 64);
                }
            }
            {
                const bool hi = lane & 16;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double keep = hi ? e4[k + 2] : e4[k], send = hi ? e4[k] : e4[k + 2];
                    e2[k] = keep + __shfl_xor(send, 16, 64);
                }
            }
            {
                const bool hi = lane & 8;
                const double keep = hi ? e2[1] : e2[0], send = hi ? e2[0] : e2[1];
                e1 = keep + __shfl_xor(send, 8, 64);
            }
            e1 += __shfl_xor(e1, 4, 64);
            e1 += __shfl_xor(e1, 2, 64);
            e1 += __shfl_xor(e1, 1, 64);
            // lane bits (5,4,3) = (b2,b1,b0): row index = 4*b2 + 2*b1 + b0
            if ((lane & 7) == 0) sm[buf][wv][8 * g + (lane >> 3)] = e1;
        }
        if (abortp && tid == 0) s_abort[sb & 1] = abortSeen;
        __syncthreads();
        if (abortp && s_abort[sb & 1] == epoch) return;   // the result would not be used
        if constexpr (RLDS) {
            // the same additions in the same order (row after row into each accumulator), the accumulators through LDS
            double t[SUB];
#pragma unroll
            for (int rr = 0; rr < SUB; ++rr) {
                double a = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) a += sm[buf][w][rr];
                t[rr] = a;
            }
#pragma unroll
            for (int m = 0; m < MAXCH; ++m) {
                double2 pa = rs[THREADS * MAXCH + tid + THREADS * m];
#pragma unroll
                for (int rr = 0; rr < SUB; ++rr) {
                    pa.x += t[rr] * y[rr][m].x;
                    pa.y += t[rr] * y[rr][m].y;
                }
                rs[THREADS * MAXCH + tid + THREADS * m] = pa;
            }
        } else {
#pragma unroll
        for (int rr = 0; rr < SUB; ++rr) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += sm[buf][w][rr];
#pragma unroll
            for (int m = 0; m < MAXCH; ++m) {
                pacc[m].x += t * y[rr][m].x;
                pacc[m].y += t * y[rr][m].y;
            }
        }
        }
    }
    double *out = ppart + ((size_t)s * nbmax + tileIdx) * nmax;
#pragma unroll
    for (int m = 0; m < MAXCH; ++m) {
        const int c = cb + 2 * tid + 2 * THREADS * m;
        if (c < ncol) {
            if constexpr (RLDS) *reinterpret_cast<double2 *>(out + c) = rs[THREADS * MAXCH + tid + THREADS * m];
            else *reinterpret_cast<double2 *>(out + c) = pacc[m];
        }
    }
}


#ifdef BS_PROFILE
__device__ long long g_bs_prof[8192][5];
#endif

// ---- small tiles, one WAVEFRONT each (round 5) -----------------------------------------------------------------------------
// On a deep dissection most tiles are small: bar17K on three levels has 1163 tiles of which 796 have rows of at most 256
// columns -- 68 % of the workgroups for 17 % of the bytes (512 of them average 24 KB), each holding a 252-register slot of the
// launch for ~10 us of latency (descriptor -> right-hand side -> rows -> butterfly -> LDS exchange -> barrier; tools/
// prof_backsolve.sh: every slot of the GPU busy for the whole launch, 3.9 TB/s).  Four such tiles share a workgroup now,
// one wavefront each, with nothing in common: no LDS, no barrier -- lane l holds columns cb + 2 l (+ 128), 16 rows per pass
// in registers, the rows' dot products through the transposed butterfly and v_readlane broadcasts, the rank-16 update from the
// same registers.  The tile's partial result goes where the block form puts it (ppart[s][tile][.]).
__device__ __forceinline__ double bs_readlane(double v, int srclane)
{
    union {
        double d;
        int i[2];
    } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], srclane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], srclane);
    return u.d;
}
// WCH chunks of 128 columns, WSUB rows per pass: <2, 16> for rows of 129 .. 256 columns, <1, 32> for rows of at most 128 (most
// small tiles: bar17K's 512 tiles of that kind average 24 KB) -- the same registers hold twice the rows, so a 64-row tile is
// a chain of two passes instead of four (the packs were the last finishers of the launch: monkey18K 34.4 us, its last ten
// workgroups packs of 48 .. 80-column tiles that started at 17 us and took 15; profiles/r05_backsolve_tiles.txt G)
template <int WCH, int WSUB>
__device__ __forceinline__ void backsolve_wave_tile_t(const int4 jb, const int *__restrict__ dofmap, const double *__restrict__ W,
                                                      int nmax, const RowTile *__restrict__ rt, const double *__restrict__ q,
                                                      double *__restrict__ ppart, int nbmax, const int *__restrict__ abortp,
                                                      int epoch)
{
    const int rows = jb.z >> 16;
    const int s = jb.x, i0 = jb.y, tileIdx = jb.z & 0xffff, cb = jb.w;
    const int ns = i0 + rows;
    const int ncol = min((ns + 15) & ~15, nmax);
    const int lane = threadIdx.x & 63;
    const RowTile rb = rt[(size_t)s * (nmax >> 6) + (i0 >> 6)];
    const int ldw = rb.ld;
    const double *Ws = W + (rb.off - (long long)(i0 & ~63) * ldw - rb.c0);
    const int *dm = dofmap + (size_t)s * nmax;
    const double *rp = q + (size_t)s * nmax;
    double2 r[WCH], pacc[WCH];
    int cend[WCH];
#pragma unroll
    for (int m = 0; m < WCH; ++m) {
        const int c = cb + 2 * lane + 128 * m;
        const int2 dd = (c < ncol) ? *reinterpret_cast<const int2 *>(dm + c) : make_int2(-1, -1);
        r[m] = (c < ncol) ? *reinterpret_cast<const double2 *>(rp + c) : make_double2(0.0, 0.0);
        pacc[m] = make_double2(0.0, 0.0);
        cend[m] = (dd.x >= 0 || dd.y >= 0) ? c : 0x7fffffff;
    }
#pragma unroll 1
    for (int ib = i0; ib < ns; ib += WSUB) {
        int ab = 0;
        if (abortp) ab = __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (answer used behind the loads)
        double2 y[WSUB][WCH];
#pragma unroll
        for (int rr = 0; rr < WSUB; ++rr) {
            const double *row = Ws + (long long)min(ib + rr, ns - 1) * ldw;
            const int rend = ((ib + rr) < ns) ? min(ncol, (ib + rr + 16) & ~15) : 0;
#pragma unroll
            for (int m = 0; m < WCH; ++m) {
                const int c = cb + 2 * lane + 128 * m;
                if (cend[m] < rend) {
                    const nt_double2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_double2 *>(row + c));
                    y[rr][m] = make_double2(v.x, v.y);
                } else {
                    y[rr][m] = make_double2(0.0, 0.0);
                }
            }
        }
        if (abortp && __builtin_amdgcn_readfirstlane(ab) == epoch) return;   // the result would not be used
        // eight rows at a time: their dot products, the wave sums, and at once their rank-8 update (rows ascending into every
        // accumulator, as in the block form) -- only eight row sums are alive
#pragma unroll
        for (int g = 0; g < WSUB / 8; ++g) {
            double d[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < WCH; ++m) acc += y[8 * g + rr][m].x * r[m].x + y[8 * g + rr][m].y * r[m].y;
                d[rr] = acc;
            }
            const double e1 = wave_sum8_transposed(d, lane);   // lane 8 k holds the wave sum of row k of the group
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const double t = bs_readlane(e1, 8 * rr);
#pragma unroll
                for (int m = 0; m < WCH; ++m) {
                    pacc[m].x += t * y[8 * g + rr][m].x;
                    pacc[m].y += t * y[8 * g + rr][m].y;
                }
            }
        }
    }
    double *out = ppart + ((size_t)s * nbmax + tileIdx) * nmax;
#pragma unroll
    for (int m = 0; m < WCH; ++m) {
        const int c = cb + 2 * lane + 128 * m;
        if (c < ncol) *reinterpret_cast<double2 *>(out + c) = pacc[m];
    }
}
__device__ __forceinline__ void backsolve_wave_tile(const int4 jb, const int *__restrict__ dofmap, const double *__restrict__ W,
                                                    int nmax, const RowTile *__restrict__ rt, const double *__restrict__ q,
                                                    double *__restrict__ ppart, int nbmax, const int *__restrict__ abortp,
                                                    int epoch)
{
    const int rows = jb.z >> 16;
    if (rows == 0) return;              // padding of the last pack
    if (abortp && __builtin_amdgcn_readfirstlane(__hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == epoch)
        return;
    // (wave-uniform: a tile is one wavefront's)
    if (jb.y + rows - jb.w <= 128) backsolve_wave_tile_t<1, 32>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, abortp, epoch);
    else backsolve_wave_tile_t<2, 16>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, abortp, epoch);
}


// the tile of job[jobIdx] by the calling workgroup
template <int THREADS>
__device__ __forceinline__ void backsolve_block(int jobIdx, const int4 *__restrict__ job, const int *__restrict__ dofmap,
                                                const double *__restrict__ W, int nmax, const RowTile *__restrict__ rt,
                                                const double *__restrict__ q, double *__restrict__ ppart, int nbmax,
                                                double (*sm)[THREADS / 64][32], const int *__restrict__ abortp = nullptr,
                                                int epoch = 0, int *s_abort = nullptr, double2 *__restrict__ rs = nullptr)
{
    if (abortp) {   // workgroups that start after the verdict leave at once (one thread asks: a uniform answer)
        if (threadIdx.x == 0) s_abort[2] = __hip_atomic_load(abortp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (s_abort[2] == epoch) return;
    }
    const int4 jb = job[jobIdx];
    const int len = jb.y + (jb.z >> 16) - jb.w;   // longest row of the tile
#ifdef BS_PROFILE
    if (threadIdx.x == 0 && jobIdx < 8192) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_bs_prof[jobIdx][0] = wall_clock64();
        g_bs_prof[jobIdx][2] = len;
        g_bs_prof[jobIdx][3] = jb.z >> 16;
        g_bs_prof[jobIdx][4] = (long long)hw | ((long long)(xcc & 15) << 32);
    }
#endif
    if constexpr (THREADS == 256) {
        if (len <= 512) backsolve_tile<256, 1, 32>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, sm, abortp, epoch, s_abort);
        else if (len <= 1024) backsolve_tile<256, 2, 16>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, sm, abortp, epoch, s_abort);
        else if (len <= 1536) backsolve_tile<256, 3, 8>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, sm, abortp, epoch, s_abort);
        else if (len <= 2560) backsolve_tile<256, 5, 8>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, sm, abortp, epoch, s_abort);
        else backsolve_tile<256, 6, 8, true>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, sm, abortp, epoch, s_abort, rs);   // <= BS_NARROW
    } else {
        if (len <= 1024) backsolve_tile<512, 1, 32>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, sm, abortp, epoch, s_abort);
        else if (len <= 2048) backsolve_tile<512, 2, 16>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, sm, abortp, epoch, s_abort);
        else if (len <= 4096) backsolve_tile<512, 4, 8>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, sm, abortp, epoch, s_abort);
        else backsolve_tile<512, 5, 8>(jb, dofmap, W, nmax, rt, q, ppart, nbmax, sm, abortp, epoch, s_abort);   // <= BS_LONG = 5120
    }
#ifdef BS_PROFILE
    __syncthreads();
    if (threadIdx.x == 0 && jobIdx < 8192) g_bs_prof[jobIdx][1] = wall_clock64();
#endif
}

// spec: the launch is speculative (early back-solve, enqueue_loop_slot): it runs on the trial gradient before the
// controller has decided about the trial, so the retry phase does not gate it
template <int THREADS>
__global__ __launch_bounds__(THREADS, 2) void backsolve_kernel(const int4 *__restrict__ job,
                                                            const int *__restrict__ dofmap,
                                                            const double *__restrict__ W, int nmax,
                                                            const RowTile *__restrict__ rt,
                                                            const double *__restrict__ q,
                                                            double *__restrict__ ppart, int nbmax,
                                                            const DevLoop *__restrict__ ctl, int spec, int nBig)
{
    __shared__ double sm[2][THREADS / 64][32];
    __shared__ int s_abort[3];
    __shared__ double2 rs[THREADS == 256 ? 2 * 256 * 6 : 1];   // (256 threads: right-hand side + accumulators of rows beyond 2560 columns)
    if (ctl && (ctl->status != 0 || (ctl->phase != 0 && !spec))) return;
    // spec > 0: the slot's epoch (its 1-based index in the step); the controller, which runs meanwhile, publishes the epoch
    // of a slot whose trial it rejects or that ends the loop (DevLoop::abortEpoch)
    const int *abortp = (ctl && spec > 0 && spec < (1 << 30)) ? &ctl->abortEpoch : nullptr;
    if constexpr (THREADS == 256) {
        if ((int)blockIdx.x >= nBig) {   // a pack of four small tiles, one wavefront each (backsolve_wave_tile)
            const int4 jq = job[nBig + 4 * ((int)blockIdx.x - nBig) + (threadIdx.x >> 6)];
#ifdef BS_PROFILE
            if (threadIdx.x == 0 && blockIdx.x < 8192) {
                g_bs_prof[blockIdx.x][0] = wall_clock64();
                g_bs_prof[blockIdx.x][2] = jq.y + (jq.z >> 16) - jq.w;
                g_bs_prof[blockIdx.x][3] = -(jq.z >> 16);
                g_bs_prof[blockIdx.x][4] = 0;
            }
#endif
            backsolve_wave_tile(jq, dofmap, W, nmax, rt, q, ppart, nbmax, abortp, spec);
#ifdef BS_PROFILE
            if (threadIdx.x == 0 && blockIdx.x < 8192) g_bs_prof[blockIdx.x][1] = wall_clock64();
#endif
            return;
        }
    }
    backsolve_block<THREADS>(blockIdx.x, job, dofmap, W, nmax, rt, q, ppart, nbmax, sm, abortp, spec, s_abort, rs);
}

// The same tiles with the loop controller as workgroup 0 of the launch: the controller's ~7 us (partial sums, the
// decision about the trial, the history update) run beside the ~45 us of streaming instead of in front of them.  The
// tiles read the loop state while workgroup 0 may be rewriting it: whichever value of `status` they see, the result is
// only used (merge_early, after the launch) if the final state says so.
template <int CTL>
__global__ __launch_bounds__(256, 2) void backsolve_ctl_kernel(const int4 *__restrict__ job, const int *__restrict__ dofmap,
                                                            const double *__restrict__ W, int nmax,
                                                            const RowTile *__restrict__ rt, const double *__restrict__ q,
                                                            double *__restrict__ ppart, int nbmax, CtlArgs ca,
                                                            int epoch, int nBig)
{
    __shared__ double sm[2][4][32];
    __shared__ int s_abort[3];
    __shared__ double2 rs[2 * 256 * 6];
    // (which workgroup hosts the controller makes no difference: index 0 / 256 / 520 / last measured 47.0-47.5 us)
    if (blockIdx.x == 0) {
        if constexpr (CTL == CTL_PAIR || CTL == CTL_PAIR_VP)
            loop_control_body<CTL>(ca.ctl, ca.partE, ca.nbE, ca.partR, ca.alpha_dev, ca.flags_host, ca.init & 1,
                                   (ca.init & 2) ? ca.partE + 2 * ELEM_NB_MAX : nullptr);
        else
            loop_control_body<CTL>(ca.ctl, ca.partE, ca.nbE, ca.partR, ca.alpha_dev, ca.flags_host, ca.init);
        return;
    }
    if (ca.ctl->status != 0) return;
    // (PAIR: a paired slot -- alpha_dev[1] > 0, written by the element pass of this slot -- waits as well: its gather worked on the
    // half step, which only counts if the controller finds the full step's energy too high)
    if (epoch < (1 << 30) && (ca.ctl->holdNext || ((CTL == CTL_PAIR || CTL == CTL_PAIR_VP) && (ca.init & 2) && ca.alpha_dev[1] > 0.0))) {
        // the trial is expected to be rejected (DevLoop::holdNext): wait for the controller's verdict instead of streaming
        // the factors beside it -- a rejection then costs the controller's ~7 us, not a stopped back-solve's ~20.  (A
        // workgroup that starts after the controller has stored its forecast for the NEXT slot reads that one: the verdict
        // is out by then, so it neither waits nor decides anything else than the abort test would.)
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            int v;
            while (((v = __hip_atomic_load(&ca.ctl->holdVerdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 1) < epoch) {
                __builtin_amdgcn_s_sleep(16);
                if (wall_clock64() - t0 > 200000000ll) break;   // 2 s: go on speculatively (the result is only used if valid)
            }
            s_abort[2] = ((v >> 1) >= epoch && (v & 1)) ? 1 : 0;
        }
        __syncthreads();
        if (s_abort[2]) return;
        __syncthreads();
    }
    const int jobIdx = (int)blockIdx.x - 1;
    if (jobIdx >= nBig) {   // a pack of four small tiles, one wavefront each
        const int4 jq = job[nBig + 4 * (jobIdx - nBig) + (threadIdx.x >> 6)];
#ifdef BS_PROFILE
        if (threadIdx.x == 0 && jobIdx < 8192) {
            g_bs_prof[jobIdx][0] = wall_clock64();
            g_bs_prof[jobIdx][2] = jq.y + (jq.z >> 16) - jq.w;
            g_bs_prof[jobIdx][3] = -(jq.z >> 16);   // (negative: a pack; rows of its first tile)
            g_bs_prof[jobIdx][4] = 0;
        }
#endif
        backsolve_wave_tile(jq, dofmap, W, nmax, rt, q, ppart, nbmax, epoch < (1 << 30) ? &ca.ctl->abortEpoch : nullptr, epoch);
#ifdef BS_PROFILE
        if (threadIdx.x == 0 && jobIdx < 8192) g_bs_prof[jobIdx][1] = wall_clock64();   // (wavefront 0 of the four)
#endif
        return;
    }
    backsolve_block<256>(jobIdx, job, dofmap, W, nmax, rt, q, ppart, nbmax, sm,
                         epoch < (1 << 30) ? &ca.ctl->abortEpoch : nullptr, epoch, s_abort, rs);
}
#ifdef BS_PROFILE
extern "C" int dotmi_debug_bs_prof(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bs_prof), sizeof(long long) * 5 * (size_t)n);
}
#endif


// psub_s[k] = sum over the row tiles b of the part whose column range holds k of ppart[s][b][k]
//   (fixed order b = 0, 1, ..., coalesced in k)
// Round 5: the tiles that can hold a column are LISTED per group of 16 columns (rp_ptr / rp_idx, ascending b; a tile's range
// starts on a multiple of 16, so a listed tile holds the first entry >> 24 columns of the group) instead of testing all
// nbmax tiles of the part for every column -- with three dissection levels a part has ~80 tiles of which ~17 hold a given
// column (1 M tets: 44.4 -> 17.7 us per launch, bunny5K 8 -> 6.6; profiles/r05_factor.txt E).  Same additions, same order.
__global__ __launch_bounds__(256) void reduce_partial_p_kernel(const int *__restrict__ rp_ptr, const int *__restrict__ rp_idx,
                                                               const double *__restrict__ ppart, int nmax,
                                                               int nbmax, double *__restrict__ psub,
                                                               const DevLoop *__restrict__ ctl, int s0)
{
    if (ctl && (ctl->status != 0 || ctl->phase != 0)) return;
    const int s = blockIdx.y + s0;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nmax) return;
    const double *base = ppart + (size_t)s * nbmax * nmax + k;
    const int ng = nmax >> 4;
    const int *pp = rp_ptr + (size_t)s * (ng + 1) + (k >> 4);
    const int e0 = pp[0], e1 = pp[1];
    // a listed tile that ends in front of column k reads a zero instead (select on the address, not a branch around the
    // load), so the loads of a batch are all in flight together
    double acc = 0.0;
    int e = e0;
    for (; e + 8 <= e1; e += 8) {
        int bb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bb[u] = rp_idx[e + u];
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // entry = tile | (columns of the group the tile holds, 1 .. 16) << 24
            const double *src = (k & 15) < (bb[u] >> 24) ? base + (size_t)(bb[u] & 0xffffff) * nmax : &g_zero_slot;
            v[u] = *src;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    {
        int bb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) bb[u] = (e + u < e1) ? rp_idx[e + u] : 0;
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double *src = (k & 15) < (bb[u] >> 24) ? base + (size_t)(bb[u] & 0xffffff) * nmax : &g_zero_slot;
            v[u] = *src;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e + u < e1) acc += v[u];
    }
    psub[(size_t)s * nmax + k] = acc;
}

// ---- rows longer than one workgroup's register tile (subdomains beyond ~1300 vertices: `timeStepper DOT 6` on a
// 17k-vertex mesh gives n_s ~ 9800) -----------------------------------------------------------------------------
// A long tile (<= 64 rows of one tree region) is cut into column chunks of BSL_CW; the single pass becomes two:
//   phase 0  tdots[tile][chunk][row] = row[chunk] . r[chunk]                       (streams the tile once)
//   phase 1  t_row = sum over chunks (fixed order);  ppart[tile][chunk columns] = sum_rows t_row * row[chunk]
//                                                                                  (streams it a second time)
// so these rows cost 2x their bytes.  Only the separator rows of the upper tree levels of big subdomains are that
// long; everything else stays on the single-pass kernel above.
constexpr int BSL_THREADS = 512, BSL_CH = 4, BSL_CW = 2 * BSL_THREADS * BSL_CH;   // 4096 columns per chunk

template <int PHASE>
__global__ __launch_bounds__(BSL_THREADS) void backsolve_long_kernel(const int4 *__restrict__ ljob,
                                                                     const int2 *__restrict__ lwork,
                                                                     const int *__restrict__ dofmap,
                                                                     const double *__restrict__ W, int nmax,
                                                                     const RowTile *__restrict__ rt,
                                                                     const double *__restrict__ q,
                                                                     double *__restrict__ tdots, int maxChunks,
                                                                     double *__restrict__ ppart, int nbmax,
                                                                     const DevLoop *__restrict__ ctl, int spec)
{
    constexpr int NW = BSL_THREADS / 64;
    __shared__ double sm[2][NW][8];
    __shared__ double tsh[BS_ROWS];
    if (ctl && (ctl->status != 0 || (ctl->phase != 0 && !spec))) return;
    const int2 wk = lwork[blockIdx.x];          // (long-tile index, chunk)
    const int4 jb = ljob[wk.x];
    const int s = jb.x, i0 = jb.y, tileIdx = jb.z & 0xffff, cb = jb.w;
    const int ns = i0 + (jb.z >> 16);
    const int ncol = min((ns + 15) & ~15, nmax);
    const int c0 = cb + wk.y * BSL_CW;           // this workgroup's columns [c0, c0 + BSL_CW) ∩ [cb, ncol)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const RowTile rb = rt[(size_t)s * (nmax >> 6) + (i0 >> 6)];
    const int ldw = rb.ld;
    const double *Ws = W + (rb.off - (long long)(i0 & ~63) * ldw - rb.c0);
    const int *dm = dofmap + (size_t)s * nmax;
    int cend[BSL_CH];
    double2 r[BSL_CH], pacc[BSL_CH];
#pragma unroll
    for (int m = 0; m < BSL_CH; ++m) {
        const int c = c0 + 2 * tid + 2 * BSL_THREADS * m;
        const int d0 = (c < ncol) ? dm[c] : -1, d1 = (c < ncol) ? dm[c + 1] : -1;
        if (PHASE == 0) r[m] = (c < ncol) ? *reinterpret_cast<const double2 *>(q + (size_t)s * nmax + c) : make_double2(0.0, 0.0);
        pacc[m] = make_double2(0.0, 0.0);
        cend[m] = (d0 >= 0 || d1 >= 0) ? c : 0x7fffffff;
    }
    if (PHASE == 1) {
        // t_row: the chunk partials of phase 0 in chunk order
        const int nch = (ncol - cb + BSL_CW - 1) / BSL_CW;
        if (tid < BS_ROWS) {
            double t = 0.0;
            for (int c = 0; c < nch; ++c) t += tdots[((size_t)wk.x * maxChunks + c) * BS_ROWS + tid];
            tsh[tid] = t;
        }
        __syncthreads();
    }
#pragma unroll 1
    for (int sb = 0; sb < BS_ROWS / 8; ++sb) {
        const int ib = i0 + sb * 8;
        if (ib >= ns) break;
        double2 y[8][BSL_CH];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const double *row = Ws + (long long)min(ib + rr, ns - 1) * ldw;
            const int rend = ((ib + rr) < ns) ? min(ncol, (ib + rr + 16) & ~15) : 0;
#pragma unroll
            for (int m = 0; m < BSL_CH; ++m) {
                const int c = c0 + 2 * tid + 2 * BSL_THREADS * m;
                y[rr][m] = (cend[m] < rend) ? *reinterpret_cast<const double2 *>(row + c) : make_double2(0.0, 0.0);
            }
        }
        if (PHASE == 0) {
            const int buf = sb & 1;
            double d[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < BSL_CH; ++m) acc += y[rr][m].x * r[m].x + y[rr][m].y * r[m].y;
                d[rr] = wave_sum(acc);
            }
            if (lane == 0) {
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) sm[buf][wv][rr] = d[rr];
            }
            __syncthreads();
            if (tid < 8) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += sm[buf][w][tid];
                tdots[((size_t)wk.x * maxChunks + wk.y) * BS_ROWS + 8 * sb + tid] = t;
            }
        } else {
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const double t = tsh[8 * sb + rr];   // rows past the tile's end were loaded as zeros
#pragma unroll
                for (int m = 0; m < BSL_CH; ++m) {
                    pacc[m].x += t * y[rr][m].x;
                    pacc[m].y += t * y[rr][m].y;
                }
            }
        }
    }
    if (PHASE == 1) {
        double *out = ppart + ((size_t)s * nbmax + tileIdx) * nmax;
#pragma unroll
        for (int m = 0; m < BSL_CH; ++m) {
            const int c = c0 + 2 * tid + 2 * BSL_THREADS * m;
            if (c < ncol) *reinterpret_cast<double2 *>(out + c) = pacc[m];
        }
    }
}

// ---- two-level form (DevTwoLevel, dotmi_internal.hpp): the three kernels around the tile kernels -------------------------------
// Reference role: the forward and backward substitution of CHOLMODSolver::solve (CHOLMODSolver.cpp:149-163) across the boundary
// between a subdomain's leaves and its separators; here with the leaves' inverse factors multiplied in (M_GD = L_GD X_DD).
// c[k] = M[row k, leaf columns] . r_leaf : a WAVEFRONT = eight consecutive rows of a panel (item = (panel, first row): no LDS, no
// barrier, the waves of a workgroup are independent -- a workgroup per panel was a chain of ~4 us steps per wave behind a ~3 us
// start, four rounds of them per CU: 181 us for 662 MB at 1 M tets), two columns per lane (a panel starts on an even column and
// has an even number of them: dotmi_create), the packed rows of a panel one behind the other
__global__ __launch_bounds__(256) void twolevel_forward_kernel(int nItems, const int2 *__restrict__ item, const int4 *__restrict__ panel,
                                                               const long long *__restrict__ rowBase, const double *__restrict__ Mp,
                                                               const double *__restrict__ rpad, double *__restrict__ cbuf,
                                                               const DevLoop *__restrict__ ctl)
{
    if (ctl && ctl->status != 0) return;
    const int lane = threadIdx.x & 63, it = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= nItems) return;
    const int2 im = item[it];
    const int4 pn = panel[im.x];
    constexpr int RW = 8;
    const int k0 = im.y, n2 = pn.w >> 1, nr = min(RW, pn.y - k0);
    const double2 *row0 = reinterpret_cast<const double2 *>(Mp + rowBase[pn.x + k0]);
    const double2 *r2 = reinterpret_cast<const double2 *>(rpad + pn.z);
    double acc[RW];
#pragma unroll
    for (int u = 0; u < RW; ++u) acc[u] = 0.0;
    for (int c = lane; c < n2; c += 64) {
        const double2 rc = r2[c];
        double2 w[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) w[u] = row0[(size_t)min(u, nr - 1) * n2 + c];
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            acc[u] += w[u].x * rc.x;
            acc[u] += w[u].y * rc.y;
        }
    }
#pragma unroll
    for (int u = 0; u < RW; ++u) {
        const double t = wave_sum(acc[u]);
        if (lane == 0 && u < nr) cbuf[pn.x + k0 + u] = t;
    }
}
// the panels' rows from where the factorisation leaves them (rows of the separators' leaf ranges, a leaf range apart) into one
// packed array, panel after panel, row after row: once per factorisation; workgroup = panel
__global__ __launch_bounds__(256) void twolevel_pack_kernel(const int4 *__restrict__ panel, const long long *__restrict__ rowSrc,
                                                            const long long *__restrict__ rowDst, const double *__restrict__ W,
                                                            double *__restrict__ packed)
{
    const int4 pn = panel[blockIdx.x];
    const int n2 = pn.w >> 1, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = wv; k < pn.y; k += 4) {
        const double2 *src = reinterpret_cast<const double2 *>(W + rowSrc[pn.x + k]);
        double2 *dst = reinterpret_cast<double2 *>(packed + rowDst[pn.x + k]);
        for (int c = lane; c < n2; c += 64) dst[c] = src[c];
    }
}
void launch_twolevel_pack(const DevParts &P, hipStream_t st)
{
    if (P.tl.on && P.tl.nPanels > 0)
        hipLaunchKernelGGL(twolevel_pack_kernel, dim3(P.tl.nPanels), dim3(256), 0, st, P.tl.panel, P.tl.rowSrc, P.tl.rowBase,
                           (const double *)P.W, P.tl.packed);
}
// t = r - sum of the panel rows' c at the separator positions (list order), r itself everywhere else
__global__ __launch_bounds__(256) void twolevel_rhs_kernel(int total, const int *__restrict__ gPtr, const int *__restrict__ gIdx,
                                                           const double *__restrict__ rpad, const double *__restrict__ cbuf,
                                                           double *__restrict__ rpad2, const DevLoop *__restrict__ ctl)
{
    if (ctl && ctl->status != 0) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        double v = rpad[i];
        const int e1 = gPtr[i + 1];
        for (int e = gPtr[i]; e < e1; ++e) v -= cbuf[gIdx[e]];
        rpad2[i] = v;
    }
}
// p_leaf -= M[rows, leaf columns]^T p_G[rows] on the per-subdomain sums: workgroup = panel; thread = (column pair, row group):
// CW column pairs across, 256 / CW row groups that take the rows in turns of eight (ascending inside a group, four interleaved
// partial sums each); the groups' sums are added in group order
template <int CW>
__global__ __launch_bounds__(256) void twolevel_backward_kernel(const int4 *__restrict__ panel, const long long *__restrict__ rowBase,
                                                                const int *__restrict__ rowPos, const double *__restrict__ W,
                                                                double *__restrict__ psub, const DevLoop *__restrict__ ctl)
{
    extern __shared__ __attribute__((aligned(16))) double tl_lds[];
    if (ctl && (ctl->status != 0 || ctl->phase != 0)) return;   // (as reduce_partial_p_kernel, whose sums this finishes)
    constexpr int RG = 256 / CW, RB = 8;
    const int4 pn = panel[blockIdx.x];
    const int tid = threadIdx.x, cg = tid % CW, g = tid / CW;
    const int nr = (pn.y + RB - 1) / RB * RB;
    double *pg = tl_lds;
    double2 *part = reinterpret_cast<double2 *>(tl_lds + nr);   // [RG][CW]
    for (int k = tid; k < nr; k += 256) pg[k] = k < pn.y ? psub[rowPos[pn.x + k]] : 0.0;
    __syncthreads();
    const int n2 = pn.w >> 1;
    // (the packed rows of a panel lie one behind the other; the rows past the panel's end -- multiplied by zeros -- re-read its last)
    const double2 *rows = reinterpret_cast<const double2 *>(W + rowBase[pn.x]);
    const int last = pn.y - 1;
    for (int c0 = 0; c0 < n2; c0 += CW) {
        const int c = c0 + cg;
        double2 a[4] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
        if (c < n2)
            for (int k = RB * g; k < nr; k += RB * RG) {
                double2 w[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u) w[u] = rows[(size_t)min(k + u, last) * n2 + c];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    a[u & 3].x += w[u].x * pg[k + u];
                    a[u & 3].y += w[u].y * pg[k + u];
                }
            }
        part[g * CW + cg] = make_double2((a[0].x + a[1].x) + (a[2].x + a[3].x), (a[0].y + a[1].y) + (a[2].y + a[3].y));
        __syncthreads();
        if (g == 0 && c < n2) {
            double2 t = part[cg];
#pragma unroll
            for (int q = 1; q < RG; ++q) {
                t.x += part[q * CW + cg].x;
                t.y += part[q * CW + cg].y;
            }
            double2 *dst = reinterpret_cast<double2 *>(psub + pn.z) + c;
            double2 v = *dst;
            v.x -= t.x;
            v.y -= t.y;
            *dst = v;
        }
        __syncthreads();
    }
}

template <int CTL>
static void launch_gemv_impl(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl, hipEvent_t ev0, hipEvent_t ev1,
                             const CtlArgs *ca, int spec)
{
    if (P.ntiles == 0 && P.nquad == 0 && P.nltiles == 0) return;
    if (q) {   // right-hand sides not in padded order yet
        const int total = P.nParts * P.nmax;
        int nb = (total + 255) / 256;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(gather_pad_kernel, dim3(nb), dim3(256), 0, st, total, P.dofmap, q, P.rpad);
    }
    const double *rhs = P.rpad;
    hipEvent_t evLast = nullptr;   // two-level form: the timed region runs from the forward kernel to the backward kernel
    if (P.tl.on) {   // two-level form: t_G = r_G - M_GD r_D in front of the tile kernels
        if (P.tl.nPanels > 0) {
            if (ev0 && ev1) {
                hipExtLaunchKernelGGL(twolevel_forward_kernel, dim3((P.tl.nItems + 3) / 4), dim3(256), 0, st, ev0, (hipEvent_t) nullptr, 0,
                                      P.tl.nItems, P.tl.item, P.tl.panel, P.tl.rowBase, (const double *)P.tl.packed,
                                      (const double *)P.rpad, P.tl.cbuf, ctl);
                evLast = ev1;
                ev0 = ev1 = nullptr;
            } else
                hipLaunchKernelGGL(twolevel_forward_kernel, dim3((P.tl.nItems + 3) / 4), dim3(256), 0, st, P.tl.nItems, P.tl.item,
                                   P.tl.panel, P.tl.rowBase, (const double *)P.tl.packed, (const double *)P.rpad, P.tl.cbuf, ctl);
        }
        const int total = P.nParts * P.nmax;
        hipLaunchKernelGGL(twolevel_rhs_kernel, dim3(std::min((total + 255) / 256, 4096)), dim3(256), 0, st, total, P.tl.gPtr, P.tl.gIdx,
                           (const double *)P.rpad, (const double *)P.tl.cbuf, P.tl.rpad2, ctl);
        rhs = P.tl.rpad2;
    }
    // optional events time the streaming kernel alone (the roofline entry of bench.py is about that kernel): they are
    // attached to the dispatch itself (hipExtLaunchKernelGGL: the packet's own begin / end time stamps, what rocprofv3
    // reports as the kernel's duration) -- two hipEventRecord calls around the launch add ~5 us of barrier packets
    // wide tiles (rows of 2561..4096 columns) on the 512-thread kernel, the rest on the 256-thread one (two workgroups per
    // CU instead of one); the events (if any) span both launches: start of the first, stop of the last
    // P.tile = [wide tiles | narrow tiles of more than 256 columns | packs of four small tiles]: job k < nN of the narrow launch
    // is one tile, job nN + k the four tiles P.tile[ntiles + 4 k ..] (one wavefront each, backsolve_wave_tile)
    const int nW = P.ntilesWide, nN = P.ntiles - P.ntilesWide, nG = nN + P.nquad;
    const bool timed = (ev0 && ev1) || evLast;
    if (ca && spec <= 0) spec = 1;   // (callers pass the slot's epoch: > 0)
    if (ca && nG == 0)   // no launch of the 256-thread kernel to host it: the controller on its own, in front
        launch_loop_control(ca->ctl, ca->partE, ca->nbE, ca->partR, ca->alpha_dev, ca->flags_host, st, ca->init & 1);
    if (nW > 0) {
        if (timed)
            hipExtLaunchKernelGGL((backsolve_kernel<512>), dim3(nW), dim3(512), 0, st, ev0, nG > 0 ? (hipEvent_t) nullptr : ev1, 0,
                                  P.tile, P.dofmap, P.W, P.nmax, P.rt, rhs, P.ppart, P.nbmax, ctl, spec, nW);
        else
            hipLaunchKernelGGL((backsolve_kernel<512>), dim3(nW), dim3(512), 0, st, P.tile, P.dofmap, P.W, P.nmax, P.rt, rhs,
                               P.ppart, P.nbmax, ctl, spec, nW);
    }
    if (nG > 0 && ca) {
        // one workgroup more: the controller (backsolve_ctl_kernel)
        if (timed)
            hipExtLaunchKernelGGL(backsolve_ctl_kernel<CTL>, dim3(nG + 1), dim3(256), 0, st, nW > 0 ? (hipEvent_t) nullptr : ev0, ev1, 0,
                                  P.tile + nW, P.dofmap, P.W, P.nmax, P.rt, rhs, P.ppart, P.nbmax, *ca, spec, nN);
        else
            hipLaunchKernelGGL(backsolve_ctl_kernel<CTL>, dim3(nG + 1), dim3(256), 0, st, P.tile + nW, P.dofmap, P.W, P.nmax, P.rt,
                               rhs, P.ppart, P.nbmax, *ca, spec, nN);
    } else if (nG > 0) {
        if (timed)
            hipExtLaunchKernelGGL((backsolve_kernel<256>), dim3(nG), dim3(256), 0, st, nW > 0 ? (hipEvent_t) nullptr : ev0, ev1, 0,
                                  P.tile + nW, P.dofmap, P.W, P.nmax, P.rt, rhs, P.ppart, P.nbmax, ctl, spec, nN);
        else
            hipLaunchKernelGGL((backsolve_kernel<256>), dim3(nG), dim3(256), 0, st, P.tile + nW, P.dofmap, P.W, P.nmax, P.rt,
                               rhs, P.ppart, P.nbmax, ctl, spec, nN);
    }
    if (P.nltiles > 0) {
        hipLaunchKernelGGL((backsolve_long_kernel<0>), dim3(P.nlwork), dim3(BSL_THREADS), 0, st, P.ltile, P.lwork, P.dofmap,
                           P.W, P.nmax, P.rt, rhs, P.tdots, P.maxChunks, P.ppart, P.nbmax, ctl, spec);
        hipLaunchKernelGGL((backsolve_long_kernel<1>), dim3(P.nlwork), dim3(BSL_THREADS), 0, st, P.ltile, P.lwork, P.dofmap,
                           P.W, P.nmax, P.rt, rhs, P.tdots, P.maxChunks, P.ppart, P.nbmax, ctl, spec);
    }
    if (!P.mt_ptr) launch_reduce_partial(P, st, ctl);   // (merge_tiles_kernel sums the tile partials itself)
    if (P.tl.on && P.tl.nPanels > 0)   // p_D = q_D - M_GD^T p_G on the sums just formed
    {
        const size_t shm = 8 * (size_t)((P.tl.maxRows + 7) & ~7) + 16 * 256;   // (dotmi_create refuses panels beyond 7000 rows)
        const int n2 = P.tl.maxCols / 2;
        if (n2 <= 32)
            hipExtLaunchKernelGGL(twolevel_backward_kernel<32>, dim3(P.tl.nPanels), dim3(256), shm, st, (hipEvent_t) nullptr, evLast, 0,
                                  P.tl.panel, P.tl.rowBase, P.tl.rowPos, (const double *)P.tl.packed, P.psub, ctl);
        else if (n2 <= 64)
            hipExtLaunchKernelGGL(twolevel_backward_kernel<64>, dim3(P.tl.nPanels), dim3(256), shm, st, (hipEvent_t) nullptr, evLast, 0,
                                  P.tl.panel, P.tl.rowBase, P.tl.rowPos, (const double *)P.tl.packed, P.psub, ctl);
        else
            hipExtLaunchKernelGGL(twolevel_backward_kernel<128>, dim3(P.tl.nPanels), dim3(256), shm, st, (hipEvent_t) nullptr, evLast, 0,
                                  P.tl.panel, P.tl.rowBase, P.tl.rowPos, (const double *)P.tl.packed, P.psub, ctl);
    }
}
void launch_gemv(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl, hipEvent_t ev0, hipEvent_t ev1,
                 const CtlArgs *ca, int spec)
{
    launch_gemv_impl<CTL_PLAIN>(P, q, st, ctl, ev0, ev1, ca, spec);
}
// ... with the controller that understands paired slots (CtlArgs::init bit 1)
void launch_gemv_pair(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl, hipEvent_t ev0, hipEvent_t ev1,
                      const CtlArgs *ca, int spec)
{
    launch_gemv_impl<CTL_PAIR>(P, q, st, ctl, ev0, ev1, ca, spec);
}
// ... paired trials on vertex patches
void launch_gemv_pair_vp(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl, hipEvent_t ev0, hipEvent_t ev1,
                         const CtlArgs *ca, int spec)
{
    launch_gemv_impl<CTL_PAIR_VP>(P, q, st, ctl, ev0, ev1, ca, spec);
}
// ... with the controller of a step on vertex patches (k_elemvert.hip)
void launch_gemv_vp(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl, hipEvent_t ev0, hipEvent_t ev1,
                    const CtlArgs *ca, int spec)
{
    launch_gemv_impl<CTL_VP>(P, q, st, ctl, ev0, ev1, ca, spec);
}
// ... with the controller of a step that takes the unit step speculatively (k_dirstep.hip)
void launch_gemv_spec(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl, hipEvent_t ev0, hipEvent_t ev1,
                      const CtlArgs *ca, int spec)
{
    launch_gemv_impl<CTL_SPEC>(P, q, st, ctl, ev0, ev1, ca, spec);
}
// the tile partials of every owned subdomain summed in the subdomains' own order (coalesced) -> psub
void launch_reduce_partial(const DevParts &P, hipStream_t st, const DevLoop *ctl)
{
    if (P.nParts > 0)
        hipLaunchKernelGGL(reduce_partial_p_kernel, dim3((P.nmax + 255) / 256, P.nParts), dim3(256), 0, st, P.rp_ptr, P.rp_idx,
                           P.ppart, P.nmax, P.nbmax, P.psub, ctl, 0);
}

// p[dofmap_s[k]] = psub_s[k] on the live positions of part s (p was cleared by the caller)
__global__ __launch_bounds__(256) void fill_part_kernel(const int *__restrict__ dofmap, const double *__restrict__ psub,
                                                        int nmax, int s, double *__restrict__ p)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nmax) return;
    const int d = dofmap[(size_t)s * nmax + k];
    if (d >= 0) p[d] = psub[(size_t)s * nmax + k];
}

void launch_gemv_part(const DevParts &P, int ls, const int4 *job, int njobs, const int2 *lwork, int nlwork, const double *q,
                      int n, double *p, hipStream_t st)
{
    hipMemsetAsync(p, 0, sizeof(double) * n, st);
    if (njobs <= 0 && nlwork <= 0) return;
    hipLaunchKernelGGL(gather_pad_kernel, dim3((P.nmax + 255) / 256), dim3(256), 0, st, P.nmax, P.dofmap + (size_t)ls * P.nmax,
                       q, P.rpad + (size_t)ls * P.nmax);
    if (njobs > 0) {
        if (P.maxTileLen <= BS_NARROW)
            hipLaunchKernelGGL((backsolve_kernel<256>), dim3(njobs), dim3(256), 0, st, job, P.dofmap, P.W, P.nmax, P.rt, P.rpad,
                               P.ppart, P.nbmax, (const DevLoop *)nullptr, 0, njobs);
        else
            hipLaunchKernelGGL((backsolve_kernel<512>), dim3(njobs), dim3(512), 0, st, job, P.dofmap, P.W, P.nmax, P.rt, P.rpad,
                               P.ppart, P.nbmax, (const DevLoop *)nullptr, 0, njobs);
    }
    if (nlwork > 0) {   // rows beyond the register tile: the two-phase kernel on this part's work items
        hipLaunchKernelGGL((backsolve_long_kernel<0>), dim3(nlwork), dim3(BSL_THREADS), 0, st, P.ltileByPart, lwork, P.dofmap,
                           P.W, P.nmax, P.rt, P.rpad, P.tdots, P.maxChunks, P.ppart, P.nbmax, (const DevLoop *)nullptr, 0);
        hipLaunchKernelGGL((backsolve_long_kernel<1>), dim3(nlwork), dim3(BSL_THREADS), 0, st, P.ltileByPart, lwork, P.dofmap,
                           P.W, P.nmax, P.rt, P.rpad, P.tdots, P.maxChunks, P.ppart, P.nbmax, (const DevLoop *)nullptr, 0);
    }
    hipLaunchKernelGGL(reduce_partial_p_kernel, dim3((P.nmax + 255) / 256, 1), dim3(256), 0, st, P.rp_ptr, P.rp_idx, P.ppart,
                       P.nmax, P.nbmax, P.psub, (const DevLoop *)nullptr, ls);
    hipLaunchKernelGGL(fill_part_kernel, dim3((P.nmax + 255) / 256), dim3(256), 0, st, P.dofmap, P.psub, P.nmax, ls, p);
}

}  // namespace dotmi
