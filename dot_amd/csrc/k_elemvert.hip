// k_elemvert.hip -- element pass + vertex gather of a line-search trial in ONE launch, on vertex patches (round 6; vpatches.hpp).
//
// Reference roles: Energy.cpp:294-423 / :910-972 (element energy and gradients), Energy.cpp:543-563 (the vFLoc sum into the
// vertices), Optimizer.cpp:1202-1215 / :1239-1252 (inertia energy and gradient, fixed rows zero), Optimizer.cpp:1023-1042 (the trial
// point x + alpha p, alpha_0 from :1076-1093), DOTTimeStepper.cpp:474-494 (the new pair and the dot products of the history update).
//
// Until round 6 these were two launches with a grid-wide join between them: elem_patch_kernel left one partial gradient per
// (element patch, vertex) in HBM and vertex_gather_kernel summed them, added the inertia term, formed the pair and its 21
// statistics and scattered -g into the padded right-hand sides of the back-solve.  Where every patch is a workgroup of its own
// (<= 512 patches) both launches are latency chains, and a launch costs ~6.5 us before and after its work whatever that work is
// (launch, kernel arguments -> loop state -> first operands, store drain; tools/prof_loopkern.sh).  A vertex patch owns its
// vertices: every element incident to them is in the patch (a halo that the neighbours carry too: ~1.9 x the element work, which
// at one workgroup per CU is idle time anyway), so the workgroup finishes its vertices itself -- no partials through HBM, no join,
// one launch less per trial.
//
// Workgroup = patch, 256 threads.  Phase 1, thread = touched vertex: positions (x_cur + alpha p, alpha from the SpMV partials as in
// elem_patch_body's prologue) -> LDS.  Phase 2, thread = 2 element slots: F, Psi, P, the 12 gradient entries; the entries of corners
// whose vertex this patch owns go to that vertex's run in LDS (ascending element id), the energy counts where this patch owns the
// element's first corner.  Phase 3, thread = (owned vertex, component): run sum -> + m (x - x~) -> g; trial point, g, -g into the
// right-hand sides, s = alpha p, y = g - g_old, H s = alpha H p, the statistics; lane 0 of a vertex its inertia energy.  Block sums:
// two energy columns (partE row = patch) and 21 statistics (partR row = patch; the controller sums as many rows as there are
// patches: loop_control_body<CTL_VP>).
// ctl == nullptr: the evaluation at the start of a step (no step, no pair: x0 -> g0, |g|^2, -g into the right-hand sides).
#include "k_device.hpp"

namespace dotmi {

constexpr int EV_EPT = 2;   // element slots per thread (vertex patches of 512 slots)

#ifdef K_PROFILE
// stage stamps of thread 0 of every workgroup (tools/prof_elemvert.sh)
__device__ long long g_evprof[512][8];
extern "C" int dotmi_debug_evprof(long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_evprof), sizeof(long long) * 512 * 8);
}
#define EVSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_evprof[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define EVSTAMP(i) do { } while (0)
#endif

// PAIR: the instantiation of a step with paired line-search trials (elem_patch_body's: when alpha_0 < 1 the first half of a launch
// twice as wide evaluates alpha_0 / 2 in full and the second half -- workgroup nP + b mirrors patch b -- the ENERGY at alpha_0,
// into the second half of the energy partials; otherwise the second half leaves at once)
template <int MAT, bool PAIR>
__global__ __launch_bounds__(256) void elem_vertex_kernel(DevVPatches VP, ElemVertArgs a, const DevLoop *__restrict__ ctl)
{
    extern __shared__ double lds[];
    __shared__ double sm[4 * RED_K];
    __shared__ double sme[8];
    __shared__ double sh_alpha;
    const bool pairedLaunch = PAIR && a.alpha_min < 0.0;
    const double alphaMin = pairedLaunch ? -a.alpha_min : a.alpha_min;
    const bool second = pairedLaunch && (int)blockIdx.x >= VP.nPatches;
    const int tid = threadIdx.x, p = second ? (int)blockIdx.x - VP.nPatches : (int)blockIdx.x;
    long long t_start = 0;
#ifdef K_PROFILE
    t_start = wall_clock64();
#endif
    // ---- loop state (one batch of scalar loads in front of the first branch) ---------------------------------------------------
    const double *__restrict__ x = a.x0;
    double *__restrict__ x_out = nullptr, *__restrict__ g_out = a.g0;
    const double *__restrict__ g_old = nullptr;
    double *__restrict__ s_new = nullptr, *__restrict__ y_new = nullptr, *__restrict__ hs_new = nullptr;
    const double *hs[HIST_MAX], *hy[HIST_MAX];
    int hm = 0, lphase = 1;
    double lalpha = 0.0;
    const bool loop = ctl != nullptr;
    if (loop) {
        const int status = ctl->status;
        lphase = ctl->phase;
        lalpha = ctl->alpha;
        x = ctl->x_cur;
        x_out = ctl->x_trial;
        g_out = ctl->g_trial;
        g_old = ctl->g_cur;
        s_new = ctl->s_new;
        y_new = ctl->y_new;
        hs_new = ctl->hs_new;
        hm = ctl->L.m;
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i) {
            hs[i] = ctl->L.s[i];
            hy[i] = ctl->L.y[i];
        }
        if (status != 0) return;
    }
#ifdef K_PROFILE
    if (threadIdx.x == 0 && blockIdx.x < 512) g_evprof[blockIdx.x][7] = t_start;   // (only launches that do work leave stamps)
#endif
    EVSTAMP(0);
    // alpha_0 = clamp(-p.g / p.Hp, alphaMin, 1) from the SpMV partials (a new direction), else the controller's halved step
    double pgv[NB_RED / 64], pHpv[NB_RED / 64];
    const bool usePart = loop && lphase == 0;
    if (usePart && tid < 64) {
#pragma unroll
        for (int u = 0; u < NB_RED / 64; ++u) {
            pgv[u] = a.spmv_partials[tid + 64 * u];             // (column-major: 512 contiguous bytes per load and wave)
            pHpv[u] = a.spmv_partials[NB_RED + tid + 64 * u];
        }
    }
    // ---- LDS: xs[3 PV] | gs[3][RUN] | cptr[PO + 1 .. padded to 4] (u16) ---------------------------------------------------------
    constexpr int PE = 256 * EV_EPT;
    double *xs = lds, *gs = lds + 3 * VP.PV;
    unsigned short *cptr = reinterpret_cast<unsigned short *>(gs + 3 * VP.RUN);
    // (the vertex ids do not wait for the patch's counts: the touched list is padded with -1, and a lane of phase 3 beyond the owned
    // vertices reads a touched vertex it then ignores -- one dependent round trip less in front of the positions)
    const int nv = VP.pv_cnt[p], no = VP.po_cnt[p];
    const size_t vb = (size_t)p * VP.PV;
    const int gid = tid < VP.PV ? VP.pv_gid[vb + tid] : -1;
    // (the run offsets go to LDS further down: a store here would wait for its load -- and, the loads returning in order, hold
    // back the request for phase 3's vertex ids behind a whole round trip)
    const unsigned short cp_reg = tid <= VP.PO ? VP.c_ptr[(size_t)p * (VP.PO + 1) + tid] : (unsigned short)0;
    // phase 3's vertex: lane (v, d) of the first 3 no lanes
    const int ov = tid / 3, od = tid - 3 * ov;
    const int ogid_raw = ov < VP.PV ? VP.pv_gid[vb + ov] : -1;
    const bool vlane = tid < 3 * no;
    const int ogid = vlane ? ogid_raw : -1;
    // ---- element operands (patch order: every load of a wave is one contiguous run) ---------------------------------------------
    ushort4 tl[EV_EPT], ep[EV_EPT];
    double Ai[EV_EPT][9], mu_[EV_EPT], la_[EV_EPT], vo[EV_EPT], voE[EV_EPT];
    const size_t strideA = (size_t)VP.nPatches * PE;
#pragma unroll
    for (int u = 0; u < EV_EPT; ++u) {
        const size_t s = (size_t)p * PE + u * 256 + tid;
        tl[u] = VP.tl[s];
        ep[u] = VP.epos[s];
#pragma unroll
        for (int k = 0; k < 9; ++k) Ai[u][k] = VP.A[(size_t)k * strideA + s];
        mu_[u] = VP.mu ? VP.mu[s] : VP.mu0;
        la_[u] = VP.mu ? VP.lam[s] : VP.lam0;
        vo[u] = VP.vol[s];
        voE[u] = VP.volE[s];
    }
    // ---- positions of the touched vertices, the step's direction there ----------------------------------------------------------
    double xv[3] = {0, 0, 0}, pv[3] = {0, 0, 0};
    if (gid >= 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            xv[d] = x[3 * gid + d];
            if (loop) pv[d] = a.p[3 * gid + d];
        }
    }
    // ---- phase 3's operands: requested now, used at the end ---------------------------------------------------------------------
    const int ok = 3 * ogid + od;   // the lane's scalar dof
    bool fx = true;
    double gold = 0.0, pk = 0.0, hpk = 0.0, ms = 0.0, xt_own = 0.0, xtv[3] = {0, 0, 0}, si[HIST_MAX], yi[HIST_MAX];
    int cb = 0, ce = 0;
#pragma unroll
    for (int i = 0; i < HIST_MAX; ++i) si[i] = yi[i] = 0.0;
    if (vlane) {
        fx = a.fixed[ogid] != 0;
        ms = a.mass[ogid];
        xt_own = a.xt[ok];
        if (od == 0) {   // (the vertex' inertia energy is this lane's)
#pragma unroll
            for (int d = 0; d < 3; ++d) xtv[d] = a.xt[3 * ogid + d];
        }
        cb = a.vp_ptr[ogid];
        ce = a.vp_ptr[ogid + 1];
        if (loop) {
            gold = g_old[ok];
            pk = a.p[ok];
            hpk = a.hp[ok];
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i) {
                si[i] = (i < hm) ? hs[i][ok] : 0.0;
                yi[i] = (i < hm) ? hy[i][ok] : 0.0;
            }
        }
    }
    if (tid <= VP.PO) cptr[tid] = cp_reg;
    EVSTAMP(1);
    // ---- alpha ------------------------------------------------------------------------------------------------------------------
    if (tid < 64) {
        double al = loop ? lalpha : 0.0;
        if (usePart) {
            double pg = 0.0, pHp = 0.0;
#pragma unroll
            for (int u = 0; u < NB_RED / 64; ++u) {
                pg += pgv[u];
                pHp += pHpv[u];
            }
            pg = __shfl(wave_sum(pg), 0, 64);
            pHp = __shfl(wave_sum(pHp), 0, 64);
            al = fmax(alphaMin, fmin(1.0, -pg / pHp));  // Optimizer.cpp:1085
        }
        if (tid == 0) {
            if constexpr (PAIR) {
                const bool pair = pairedLaunch && usePart && al < 1.0 && al / 2.0 > 0.0 && ctl->pairCtr[pair_band(al)] >= 3;
                sh_alpha = pair ? (second ? al : al / 2.0) : (second ? -1.0 : al);
                if (blockIdx.x == 0) {
                    a.alpha_out[0] = pair ? al / 2.0 : al;
                    if (pairedLaunch) a.alpha_out[1] = pair ? al : 0.0;
                }
            } else {
                sh_alpha = al;
                if (p == 0 && loop) *a.alpha_out = al;
            }
        }
    }
    __syncthreads();
    const double alpha = sh_alpha;
    if (second && alpha < 0.0) return;   // (the whole workgroup: not a paired slot)
    EVSTAMP(2);
    // ---- phase 1: trial positions of the touched vertices -> LDS ----------------------------------------------------------------
    if (gid >= 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) xs[3 * tid + d] = loop ? xv[d] + alpha * pv[d] : xv[d];
    }
    for (int lv = tid + 256; lv < nv; lv += 256) {   // (a patch that touches more than 256 vertices)
        const int g2 = VP.pv_gid[vb + lv];
#pragma unroll
        for (int d = 0; d < 3; ++d) xs[3 * lv + d] = loop ? x[3 * g2 + d] + alpha * a.p[3 * g2 + d] : x[3 * g2 + d];
    }
    __syncthreads();
    EVSTAMP(3);
    // the padded positions of the vertex' copies (a second dependent load behind the vertex id: requested here, where the first has
    // long arrived, not in front of the barriers above; used at the end)
    constexpr int VC = 4;   // copies of a vertex (subdomains that hold it) whose positions are requested ahead of the stores
    int vof[VC];
#pragma unroll
    for (int c = 0; c < VC; ++c) vof[c] = (cb + c < ce) ? a.vp_off[cb + c] : 0;
    // ---- phase 2: elements ------------------------------------------------------------------------------------------------------
    double acc = 0.0;   // sum vol * Psi over the elements whose energy this patch counts
#pragma unroll
    for (int u = 0; u < EV_EPT; ++u) {
        if (tl[u].x == 0xFFFF) continue;   // padding slot
        const double *p0 = xs + 3 * tl[u].x, *p1 = xs + 3 * tl[u].y, *p2 = xs + 3 * tl[u].z, *p3 = xs + 3 * tl[u].w;
        Mat3 F;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double d0 = p1[r] - p0[r], d1 = p2[r] - p0[r], d2 = p3[r] - p0[r];
#pragma unroll
            for (int c = 0; c < 3; ++c) F.m[r][c] = d0 * Ai[u][c] + d1 * Ai[u][3 + c] + d2 * Ai[u][6 + c];
        }
        const double w = a.dtSq * vo[u];
        double P[3][3];
        if constexpr (MAT == 1) {
            // Stable Neo-Hookean in |F|^2 and det F (elem_patch_body: the same statements)
            const double J = det3(F);
            const double ic = F.m[0][0] * F.m[0][0] + F.m[0][1] * F.m[0][1] + F.m[0][2] * F.m[0][2] +
                              F.m[1][0] * F.m[1][0] + F.m[1][1] * F.m[1][1] + F.m[1][2] * F.m[1][2] +
                              F.m[2][0] * F.m[2][0] + F.m[2][1] * F.m[2][1] + F.m[2][2] * F.m[2][2];
            const double JmA = J - (1.0 + mu_[u] / la_[u]);
            acc += (mu_[u] * (ic - 3.0) + la_[u] * JmA * JmA) / 2.0 * voE[u];
            const double t = la_[u] * JmA;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int r1 = (r + 1) % 3, r2 = (r + 2) % 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                    const double cof = F.m[r1][c1] * F.m[r2][c2] - F.m[r1][c2] * F.m[r2][c1];
                    P[r][c] = w * (mu_[u] * F.m[r][c] + t * cof);
                }
            }
        } else {
            Mat3 U, V;
            double S[3];
            svd3(F, U, S, V);
            acc += psi<MAT>(S, mu_[u], la_[u]) * voE[u];
            double dd[3];
            dpsi<MAT>(S, mu_[u], la_[u], dd);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    P[r][c] = w * (U.m[r][0] * dd[0] * V.m[c][0] + U.m[r][1] * dd[1] * V.m[c][1] + U.m[r][2] * dd[2] * V.m[c][2]);
        }
        double g[12];
#pragma unroll
        for (int aa = 0; aa < 3; ++aa)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                g[3 + 3 * aa + c] = Ai[u][3 * aa] * P[c][0] + Ai[u][3 * aa + 1] * P[c][1] + Ai[u][3 * aa + 2] * P[c][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) g[c] = -g[3 + c] - g[6 + c] - g[9 + c];
        const int pkk[4] = {ep[u].x, ep[u].y, ep[u].z, ep[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (pkk[k] != 0xFFFF && !second) {   // the corner's vertex is this patch's
                gs[pkk[k]] = g[3 * k];
                gs[VP.RUN + pkk[k]] = g[3 * k + 1];
                gs[2 * VP.RUN + pkk[k]] = g[3 * k + 2];
            }
    }
    __syncthreads();
    EVSTAMP(4);
    // ---- phase 3: the owned vertices --------------------------------------------------------------------------------------------
    double st[RED_K];
#pragma unroll
    for (int j = 0; j < RED_K; ++j) st[j] = 0.0;
    double ine = 0.0;
    if (vlane && second) {
        if (od == 0) {   // (the energy of the full step: the inertia term of the owned vertices)
            const double dx = xs[3 * ov] - xtv[0], dy = xs[3 * ov + 1] - xtv[1], dz = xs[3 * ov + 2] - xtv[2];
            ine = (dx * dx + dy * dy + dz * dz) * ms / 2.0;
        }
    } else if (vlane) {
        const int kb = cptr[ov], ke = cptr[ov + 1];
        const double *run = gs + od * VP.RUN;
        double sum = 0.0;
        for (int k = kb; k < ke; k += 4) {   // four entries in flight, added in run order (ascending element id)
            const double w0 = run[k], w1 = k + 1 < ke ? run[k + 1] : 0.0, w2 = k + 2 < ke ? run[k + 2] : 0.0,
                         w3 = k + 3 < ke ? run[k + 3] : 0.0;
            sum += w0;
            if (k + 1 < ke) sum += w1;
            if (k + 2 < ke) sum += w2;
            if (k + 3 < ke) sum += w3;
        }
        const double xt_k = xs[3 * ov + od];   // the trial point at this dof (owned vertices are the first of the touched list)
        // fixed rows of the gradient are zero (Optimizer.cpp:1239-1252); the inertia term m (x - x~)
        const double g = fx ? 0.0 : sum + ms * (xt_k - xt_own);
        if (x_out) x_out[ok] = xt_k;
        g_out[ok] = g;
#pragma unroll
        for (int c = 0; c < VC; ++c)
            if (cb + c < ce) a.rpad[vof[c] + od] = -g;
        for (int c = cb + VC; c < ce; ++c) a.rpad[a.vp_off[c] + od] = -g;
        st[0] = g * g;
        if (loop) {
            const double sn = alpha * pk;
            const double yn = g - gold;
            s_new[ok] = sn;
            y_new[ok] = yn;
            hs_new[ok] = alpha * hpk;   // H s_new = alpha H p
            st[1] = yn * sn;
            st[2] = sn * g;
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i)
                if (i < hm) {
                    st[3 + i] = si[i] * yn;
                    st[3 + HIST_MAX + i] = sn * yi[i];
                    st[3 + 2 * HIST_MAX + i] = si[i] * g;
                }
        }
        if (od == 0) {   // the vertex' inertia energy 1/2 m |x - x~|^2 (Optimizer.cpp:1202-1215)
            const double dx = xs[3 * ov] - xtv[0], dy = xs[3 * ov + 1] - xtv[1], dz = xs[3 * ov + 2] - xtv[2];
            ine = (dx * dx + dy * dy + dz * dz) * ms / 2.0;
        }
    }
    EVSTAMP(5);
    // ---- block sums: the two energy columns, the statistics ---------------------------------------------------------------------
    const double we = wave_sum(acc), wi = wave_sum(ine);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) {
        sme[wv] = we;
        sme[4 + wv] = wi;
    }
    if (!second) write_partials(st, loop ? RED_K : 1, a.partR, sm);   // (its first barrier also publishes sme)
    else __syncthreads();
    if (tid == 0) {
        double *pe = second ? a.partE + 2 * ELEM_NB_MAX : a.partE;   // (second half: the full step's partials)
        pe[2 * p] = (sme[0] + sme[1]) + (sme[2] + sme[3]);      // to be scaled by dtSq by the consumer
        pe[2 * p + 1] = (sme[4] + sme[5]) + (sme[6] + sme[7]);
    }
    EVSTAMP(6);
}

void launch_elem_vertex(const DevVPatches &VP, int mat, const ElemVertArgs &a, hipStream_t st, const DevLoop *ctl)
{
    const size_t shm = sizeof(double) * ((size_t)3 * VP.PV + (size_t)3 * VP.RUN) + 2 * (size_t)((VP.PO + 1 + 3) & ~3);
    const bool paired = ctl && a.alpha_min < 0.0;
    if (paired) {
        if (mat == 0) hipLaunchKernelGGL((elem_vertex_kernel<0, true>), dim3(2 * VP.nPatches), dim3(256), shm, st, VP, a, ctl);
        else hipLaunchKernelGGL((elem_vertex_kernel<1, true>), dim3(2 * VP.nPatches), dim3(256), shm, st, VP, a, ctl);
    } else {
        if (mat == 0) hipLaunchKernelGGL((elem_vertex_kernel<0, false>), dim3(VP.nPatches), dim3(256), shm, st, VP, a, ctl);
        else hipLaunchKernelGGL((elem_vertex_kernel<1, false>), dim3(VP.nPatches), dim3(256), shm, st, VP, a, ctl);
    }
}

}  // namespace dotmi
