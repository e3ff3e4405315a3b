// k_elembody.hpp -- the element pass as a device function (private): shared by k_element.hip (a launch of its own) and
// k_dirstep.hip (one population of the speculative unit-step launch).
#pragma once
#include "k_device.hpp"

namespace dotmi {

// ------------------------------------------------------------------------------------------------
// element pass: energy (+ inertia) partials and per-(patch, vertex) partial gradients
//
// One workgroup per PATCH of <= PE = 256 * EPT elements (patches.hpp): the positions of the patch's vertices are staged
// in LDS (each read from HBM once per patch instead of once per incident corner), a lane computes F, Psi and the
// 12 entries of the element gradient of its EPT elements -- all of its operands (corner indices, rest-shape inverse,
// material, volume) are stored in patch order, so every load of a wave is one contiguous run --, drops the 12 entries
// into LDS at the corner's position in its vertex's run, and the runs are then summed in ascending element order (the
// reference's vFLoc order, Energy.cpp:543-563), one lane per (vertex, component).  HBM sees one 24-byte partial per (patch, vertex): ~2 per
// vertex instead of the ~22 incident contributions of 24 bytes each that a global scatter / gather moves twice.
// LDS: xs[3 * PV] | gs[3][4 PE] (component-major corner runs) | cptr | slot.
// ------------------------------------------------------------------------------------------------
#ifdef EP_PROFILE
// per-workgroup wall-clock stamps of the element pass (tools/prof_elem.sh): start, operands + positions in LDS, element
// work + gradient runs in LDS, run sums + stores issued, end
static __device__ long long g_ep_prof[8192][6];   // (read back by dotmi_debug_ep_prof, k_element.hip: that unit's copy)
#define EP_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_ep_prof[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define EP_STAMP(i) do { } while (0)
#endif
#ifdef DS_PROFILE
static __device__ long long g_sp_prof[4096][4];   // (tools/prof_dirstep.sh: stamps inside the speculative prologue)
#define SP_STAMP(i) do { if (SPEC && threadIdx.x == 0 && bIdx0 < 4096) g_sp_prof[bIdx0][i] = wall_clock64(); } while (0)
#else
#define SP_STAMP(i) do { } while (0)
#endif
// PAIR: the instantiation of a step with paired line-search trials -- a template parameter, so the plain
// instantiation compiles every paired branch away (until round 5 the file was compiled twice, with and without -DDOTMI_PAIR_TU)
// SPEC (k_dirstep.hip, round 6): the unit step is taken SPECULATIVELY beside the direction kernel instead of behind it.  The
// launch has two workgroup populations with no dependency between them: spmv_zp_body's rows (p, H p, the partials of alpha_0 =
// clamp(-p.g / p.Hp)) and this body's patches, which evaluate x + 1 p with p_v = z_v + sum_j delta_j s_j[v] formed HERE for the
// vertices they read (delta from the same y_i . z partials, the additions in the order spmv_zp_body's lane groups make them, so
// the positions are the bits the plain slot forms from the stored p with alpha = 1).  alpha_0 clamps to 1 in most iterations
// (bar17K: all of them); the controller checks it afterwards and has the slot redone with the true alpha_0 otherwise
// (loop_control_body).  A retry slot (phase != 0) of such a step runs this body in its plain form on the stored p.
// The inertia term and the trial point (one vertex per thread, element workgroup b those of vertices 256 b ...) are the work of a
// third population of that launch (inertia_step_body below: nbIne = ceil(nV / 256) small workgroups that write the inertia
// column of the energy partials -- the same vertices, the same sums as the element workgroup b makes in the plain form), so the
// patches' workgroups do not carry the operands of p for a second vertex (42 registers).
struct SpecArgs {
    const double *z, *c_partials;   // the preconditioned vector of the two-loop, the y_i . z partials in COLUMN-major form (HIST_MAX x NB_RED)
    int nbIne;                      // workgroups of the inertia population (element workgroups beyond them store a zero inertia partial)
};
// the operands of p at scalar dof k: z and the stored s_j, requested together, combined once delta is known
struct DirOps {
    double w[1 + HIST_MAX];
};
// delta (second half of the two-loop) and p = z + sum_j delta_j s_j formed outside spmv_zp_body, with ITS statements: the y_i . z
// partial columns requested by wave 0 (request), reduced and run through the recurrence (finish: delta -> LDS; the caller's
// barrier follows), p_k combined in the order the row's lane group of spmv_zp_body adds its eight terms (form)
struct SpecDir {
    double cyz[8];
    TwoLoopCoef coef;
    int hm;
    const double *__restrict__ hs[HIST_MAX];
    const double *__restrict__ z;
    __device__ __forceinline__ void request(const DevLoop *__restrict__ ctl, const SpecArgs &sx, bool spec)
    {
        hm = ctl->L.m;
        z = sx.z;
#pragma unroll
        for (int j = 0; j < HIST_MAX; ++j) hs[j] = ctl->L.s[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) cyz[i] = 0.0;
        if (spec && threadIdx.x < 64) {
            coef.request(ctl);
            load_yz_partials(sx.c_partials, cyz, true);   // (the column-major twin: SpecArgs)
        }
    }
    // wave 0 (threadIdx.x < 64): delta[0 .. HIST_MAX) -> LDS
    __device__ __forceinline__ void finish(const DevLoop *__restrict__, double *delta) { coef.delta(cyz, hm, delta); }
    __device__ __forceinline__ void load(int k, DirOps &o) const
    {
        o.w[0] = z[k];
#pragma unroll
        for (int j = 0; j < HIST_MAX; ++j) o.w[1 + j] = (j < hm) ? hs[j][k] : 0.0;
    }
    // lane j of the row's group of eight holds t_j = (j == 0 ? z_k : 0) + s_j[k] delta_j (zero beyond the stored pairs) and the
    // group's butterfly adds them as ((t0 + t4) + (t2 + t6)) + ((t1 + t5) + (t3 + t7))
    __device__ __forceinline__ double form(const DirOps &o, const double *delta) const
    {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double zz = j == 0 ? o.w[0] : 0.0;
            const double sj = j < HIST_MAX ? o.w[1 + (j < HIST_MAX ? j : 0)] : 0.0;
            const double dj = j < HIST_MAX ? delta[j < HIST_MAX ? j : 0] : 0.0;
            t[j] = zz + sj * dj;
        }
        return ((t[0] + t[4]) + (t[2] + t[6])) + ((t[1] + t[5]) + (t[3] + t[7]));
    }
};
// bIdx0 / nbAll: this workgroup's index in the population of element workgroups and their number (the whole grid unless SPEC);
// lds: the dynamic LDS of the launch; sm: 8 doubles; sh: 1 + HIST_MAX doubles (alpha, delta)
template <int MAT, bool GRAD, int EPT, bool FUSE, bool PIPE, bool PAIR, bool SPEC = false>
__device__ __forceinline__ void elem_patch_body(const DevPatches &PT, const double *__restrict__ mass,
                                                const double *__restrict__ x, const double *__restrict__ xt, int v0, int v1,
                                                double dtSq, double *__restrict__ partials, const DevLoop *__restrict__ ctl,
                                                const StepArgs &sa, const SpecArgs &sx, int bIdx0, int nbAll, double *lds,
                                                double *sm, double *sh)
{
    static_assert(!(PAIR && SPEC), "a step pairs its trials or speculates on the unit step");
    static_assert(!SPEC || (FUSE && GRAD), "the speculative form is the fused step with gradients");
    double &sh_alpha = sh[0];
    // sa.p != nullptr (device loop): the line-search step x_trial = x_cur + alpha p (step_forward_kernel's statements) is
    // taken HERE: every position this kernel reads is formed on the fly, the trial point is written by the inertia loop
    // (which visits every vertex exactly once), and alpha comes from the SpMV partials in wave 0's prologue -- one launch
    // and one pass over x, p less per line-search trial
    constexpr bool fuse = FUSE;   // (a template parameter: the plain instantiation keeps its register count)
    double *__restrict__ x_out = nullptr;
    // (device loop: the loop state this body needs, every load of it in front of the first branch -- one round trip)
    int lphase = 0;
    double lalpha = 0.0;
    if (ctl) {
        const int status = ctl->status;
        lphase = ctl->phase;
        lalpha = ctl->alpha;
        const double *xc = ctl->x_cur;
        double *xtr = ctl->x_trial;
        if (status != 0) return;
        x = fuse ? xc : xtr;
        x_out = xtr;
    }
    // paired trial (PAIR, StepArgs::alpha_min < 0): the launch is twice as wide; workgroup nbP + b mirrors workgroup b on the FULL step
    const bool pairedLaunch = PAIR && fuse && sa.alpha_min < 0.0;
    const double alphaMin = pairedLaunch ? -sa.alpha_min : sa.alpha_min;
    const int nbP = pairedLaunch ? nbAll / 2 : nbAll;
    const bool second = pairedLaunch && bIdx0 >= nbP;
    const int bIdx = second ? bIdx0 - nbP : bIdx0;
    const bool withGrad = GRAD && !second;
    double pgv[NB_RED / 64], pHpv[NB_RED / 64];
    // SPEC: the slot computes a new direction (phase 0) -> the unit step on a direction formed here; a retry: the plain form
    const bool spec = SPEC && lphase == 0;
    const bool usePart = fuse && !spec && lphase == 0;   // a retry steps with the halved alpha the controller left
    // SPEC: wave 0 requests the y_i . z partial columns now (spmv_zp_body's prologue, the same statements: the same delta)
    SpecDir sd;
    if constexpr (SPEC) sd.request(ctl, sx, spec);
    auto dir_load = [&](int k, DirOps &o) { sd.load(k, o); };
    auto dir_form = [&](const DirOps &o) -> double { return sd.form(o, sh + 1); };
    if (usePart && threadIdx.x < 64) {              // requested here, summed after this thread's other loads are out
#pragma unroll
        for (int u = 0; u < NB_RED / 64; ++u) {
            pgv[u] = sa.spmv_partials[(size_t)(threadIdx.x + 64 * u) * RED_K];
            pHpv[u] = sa.spmv_partials[(size_t)(threadIdx.x + 64 * u) * RED_K + 1];
        }
    }
    bool haveAlpha = false;
    double alpha = 0.0;
    auto finish_alpha = [&]() {   // (all threads; one barrier)
        if (spec) {
            // delta (second half of the two-loop, build_p_kernel's prologue) by wave 0 -> LDS; the step is 1
            if (threadIdx.x < 64) {
                sd.finish(ctl, sh + 1);
                if (threadIdx.x == 0) {
                    sh_alpha = 1.0;
                    if (bIdx0 == 0) *sa.alpha_out = 1.0;   // what the gather scales the pair with; the controller checks alpha_0
                }
            }
            __syncthreads();
            alpha = 1.0;
            haveAlpha = true;
            return;
        }
        if (threadIdx.x < 64) {
            double a = lalpha;
            if (usePart) {
                double pg = 0.0, pHp = 0.0;
#pragma unroll
                for (int u = 0; u < NB_RED / 64; ++u) {
                    pg += pgv[u];
                    pHp += pHpv[u];
                }
                pg = __shfl(wave_sum(pg), 0, 64);
                pHp = __shfl(wave_sum(pHp), 0, 64);
                a = fmax(alphaMin, fmin(1.0, -pg / pHp));  // Optimizer.cpp:1085
            }
            if (threadIdx.x == 0) {
                if constexpr (PAIR) {
                    // paired: alpha_0 < 1 (the quadratic model's minimum lies inside the unit step) -- the full step's energy
                    // from the second half of the launch, the half step in full from the first
                    const bool pair = pairedLaunch && usePart && a < 1.0 && a / 2.0 > 0.0 && ctl->pairCtr[pair_band(a)] >= 3;
                    sh_alpha = pair ? (second ? a : a / 2.0) : (second ? -1.0 : a);
                    if (bIdx0 == 0) {
                        sa.alpha_out[0] = pair ? a / 2.0 : a;
                        if (pairedLaunch) sa.alpha_out[1] = pair ? a : 0.0;
                    }
                } else {
                    sh_alpha = a;
                    if (bIdx0 == 0) *sa.alpha_out = a;
                }
            }
        }
        __syncthreads();
        alpha = sh_alpha;
        haveAlpha = true;
    };
    EP_STAMP(0);
    constexpr int PE = 256 * EPT;
    // LDS: xs[3 PV] | gs[3][4 PE] | cptr[PV + 1 .. padded] (u16) | slot[PV] (i32)
    double *xs = lds, *gs = lds + 3 * PT.PV;
    unsigned short *cptr = reinterpret_cast<unsigned short *>(gs + (GRAD ? 12 * PE : 0));
    int *vslot = reinterpret_cast<int *>(cptr + ((PT.PV + 1 + 3) & ~3));
    const int tid = threadIdx.x;
    const size_t strideA = (size_t)PT.nPatches * PE;
    double acc = 0.0;  // sum vol * Psi
    // inertia operands of this thread's first vertex: independent of the element work, requested ahead of it
    const int gstride = nbP * blockDim.x;
    const int vfirst = v0 + bIdx * blockDim.x + tid;
    // (SPEC: the inertia term and the trial point are another population's work, inertia_step_body)
    double ix[3] = {0, 0, 0}, ixt[3] = {0, 0, 0}, ip[3] = {0, 0, 0}, im = 0.0;
    if (!SPEC && vfirst < v1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ix[d] = x[3 * vfirst + d];
            ixt[d] = xt[3 * vfirst + d];
            if (fuse) ip[d] = sa.p[3 * vfirst + d];
        }
        im = mass[vfirst];
    }
    // The operands of a patch in registers.  A workgroup that owns several patches (meshes beyond elem_wg_cap() patches) requests
    // the NEXT patch's operands and positions while it works on the current one: the ~5 us a patch waits for its two
    // dependent round trips (vertex id -> position) are then covered by the previous patch's arithmetic and run sums.
    struct PatchOps {
        int nv, gid0, slot0;
        unsigned short cp0;
        ushort4 tl[EPT], ep[EPT];
        double Ai[EPT][9], m[EPT], l[EPT], vo[EPT];
        double xv[3], pv[3];
        DirOps po[SPEC ? 3 : 1];   // SPEC: the operands of p at the patch vertex (combined when delta is there)
    };
    // three stages, each one dependent round trip: vertex ids (and the short lists), positions, element operands
    auto issue_ids = [&](int p, PatchOps &o) {
        o.nv = PT.pv_cnt[p];
        const size_t vb = (size_t)p * PT.PV;
        o.gid0 = tid < o.nv ? PT.pv_gid[vb + tid] : -1;
        o.cp0 = 0;
        o.slot0 = 0;
        if (GRAD) {
            if (tid <= o.nv) o.cp0 = PT.c_ptr[(size_t)p * (PT.PV + 1) + tid];
            if (tid < o.nv) o.slot0 = PT.pv_slot[vb + tid];
        }
    };
    auto issue_ops = [&](int p, PatchOps &o) {
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            const size_t s = (size_t)p * PE + u * 256 + tid;
            o.tl[u] = PT.tl[s];
            if (GRAD) o.ep[u] = PT.epos[s];
#pragma unroll
            for (int k = 0; k < 9; ++k) o.Ai[u][k] = PT.A[(size_t)k * strideA + s];
            o.m[u] = PT.mu ? PT.mu[s] : PT.mu0;
            o.l[u] = PT.mu ? PT.lam[s] : PT.lam0;
            o.vo[u] = PT.vol[s];
        }
    };
    auto issue_pos = [&](PatchOps &o) {
#pragma unroll
        for (int d = 0; d < 3; ++d) o.xv[d] = o.pv[d] = 0.0;
        if (o.gid0 >= 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                o.xv[d] = x[3 * o.gid0 + d];
                if constexpr (SPEC) {
                    if (spec) dir_load(3 * o.gid0 + d, o.po[d]);
                    else o.pv[d] = sa.p[3 * o.gid0 + d];
                } else if (fuse) o.pv[d] = sa.p[3 * o.gid0 + d];
            }
        }
    };
    PatchOps cur, nxt;
    int nvN = 0, gidN = -1, slotN = 0;   // ids of the patch after next
    unsigned short cpN = 0;
    SP_STAMP(0);
    if ((int)bIdx < PT.nPatches) {
        issue_ids(bIdx, cur);
        issue_ops(bIdx, cur);
        issue_pos(cur);
        if (PIPE && (int)(bIdx + nbP) < PT.nPatches) issue_ids(bIdx + nbP, nxt);
    }
    for (int p = bIdx; p < PT.nPatches; p += nbP) {
        // PIPE: the instantiation for meshes whose workgroups walk several patches (the prefetched set costs ~50 registers,
        // which the one-patch-per-workgroup meshes keep for occupancy)
        const int pn = p + nbP, pn2 = pn + nbP;
        const bool more = PIPE && pn < PT.nPatches, more2 = PIPE && pn2 < PT.nPatches;
        if (!PIPE && p != (int)bIdx) {
            issue_ids(p, cur);
            issue_ops(p, cur);
            issue_pos(cur);
        }
        const int nv = cur.nv;
        const size_t vb = (size_t)p * PT.PV;
        auto &tl = cur.tl;
        auto &ep = cur.ep;
        auto &Ai = cur.Ai;
        auto &m = cur.m;
        auto &l = cur.l;
        auto &vo = cur.vo;
        if (GRAD) {
            if (tid <= nv) cptr[tid] = cur.cp0;
            if (tid < nv) vslot[tid] = cur.slot0;
            const unsigned short *cp = PT.c_ptr + (size_t)p * (PT.PV + 1);
            for (int lv = tid + 256; lv <= nv; lv += 256) cptr[lv] = cp[lv];
            for (int lv = tid + 256; lv < nv; lv += 256) vslot[lv] = PT.pv_slot[vb + lv];
        }
        SP_STAMP(1);
        if (fuse && !haveAlpha) {
            finish_alpha();
            if (second && alpha < 0.0) return;   // (the whole workgroup: not a paired slot)
        }
        SP_STAMP(2);
        if (cur.gid0 >= 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if constexpr (SPEC) {
                    if (spec) cur.pv[d] = dir_form(cur.po[d]);
                }
                xs[3 * tid + d] = fuse ? cur.xv[d] + alpha * cur.pv[d] : cur.xv[d];
            }
        }
        for (int lv = tid + 256; lv < nv; lv += 256) {
            const int gid = PT.pv_gid[vb + lv];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if constexpr (SPEC) {
                    if (spec) {
                        DirOps o;
                        dir_load(3 * gid + d, o);
                        xs[3 * lv + d] = x[3 * gid + d] + alpha * dir_form(o);
                        continue;
                    }
                }
                xs[3 * lv + d] = fuse ? x[3 * gid + d] + alpha * sa.p[3 * gid + d] : x[3 * gid + d];
            }
        }
        // the next patch: its ids came in during the previous patch -> positions and operands now, in flight during this
        // patch's work; the ids of the patch after it as well
        if (more) {
            issue_pos(nxt);
            issue_ops(pn, nxt);
        }
        if (more2) {
            PatchOps t;
            issue_ids(pn2, t);
            nvN = t.nv;
            gidN = t.gid0;
            slotN = t.slot0;
            cpN = t.cp0;
        }
        __syncthreads();
        EP_STAMP(1);
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (tl[u].x == 0xFFFF) continue;   // padding slot of the last patch
            const double *p0 = xs + 3 * tl[u].x, *p1 = xs + 3 * tl[u].y, *p2 = xs + 3 * tl[u].z, *p3 = xs + 3 * tl[u].w;
            Mat3 F;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double d0 = p1[r] - p0[r], d1 = p2[r] - p0[r], d2 = p3[r] - p0[r];
#pragma unroll
                for (int c = 0; c < 3; ++c) F.m[r][c] = d0 * Ai[u][c] + d1 * Ai[u][3 + c] + d2 * Ai[u][6 + c];
            }
            const double w = dtSq * vo[u];
            double P[3][3];
            if constexpr (MAT == 1) {
                // Stable Neo-Hookean: Psi(sigma) = (mu (|sigma|^2 - 3) + lam (J - a)^2) / 2, a = 1 + mu / lam
                // (StableNHEnergy.cpp:91-130), is a function of |F|_F^2 = |sigma|^2 and det F = J only (the reference's SVD
                // has U, V in SO(3) and the sign of det F on sigma_3), and U diag(dPsi/dsigma) V^T = mu F + lam (J - a) cof F.
                // Energy and first Piola stress therefore need no SVD here; the Hessian (once per step) keeps the SVD.
                const double J = det3(F);
                const double ic = F.m[0][0] * F.m[0][0] + F.m[0][1] * F.m[0][1] + F.m[0][2] * F.m[0][2] +
                                  F.m[1][0] * F.m[1][0] + F.m[1][1] * F.m[1][1] + F.m[1][2] * F.m[1][2] +
                                  F.m[2][0] * F.m[2][0] + F.m[2][1] * F.m[2][1] + F.m[2][2] * F.m[2][2];
                const double JmA = J - (1.0 + m[u] / l[u]);
                acc += (m[u] * (ic - 3.0) + l[u] * JmA * JmA) / 2.0 * vo[u];
                if (withGrad) {
                    const double t = l[u] * JmA;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int r1 = (r + 1) % 3, r2 = (r + 2) % 3;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
                            const double cof = F.m[r1][c1] * F.m[r2][c2] - F.m[r1][c2] * F.m[r2][c1];
                            P[r][c] = w * (m[u] * F.m[r][c] + t * cof);
                        }
                    }
                }
            } else {
                Mat3 U, V;
                double S[3];
                svd3(F, U, S, V);
                acc += psi<MAT>(S, m[u], l[u]) * vo[u];
                if (withGrad) {
                    double d[3];
                    dpsi<MAT>(S, m[u], l[u], d);
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c)
                            P[r][c] = w * (U.m[r][0] * d[0] * V.m[c][0] + U.m[r][1] * d[1] * V.m[c][1] +
                                           U.m[r][2] * d[2] * V.m[c][2]);
                }
            }
            if (withGrad) {
                double g[12];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        g[3 + 3 * a + c] = Ai[u][3 * a] * P[c][0] + Ai[u][3 * a + 1] * P[c][1] + Ai[u][3 * a + 2] * P[c][2];
#pragma unroll
                for (int c = 0; c < 3; ++c) g[c] = -g[3 + c] - g[6 + c] - g[9 + c];
                const int pk[4] = {ep[u].x, ep[u].y, ep[u].z, ep[u].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    gs[pk[k]] = g[3 * k];
                    gs[4 * PE + pk[k]] = g[3 * k + 1];
                    gs[8 * PE + pk[k]] = g[3 * k + 2];
                }
            }
        }
        if (withGrad) {
            __syncthreads();
            EP_STAMP(2);
            // one lane per (vertex, component): a contiguous run of LDS, four entries in flight, added in run order
            // (vertex-major: three consecutive lanes write the 24 contiguous bytes of one partial)
            for (int item = tid; item < 3 * nv; item += 256) {
                const int lv = item / 3, d = item - 3 * lv;
                const int kb = cptr[lv], ke = cptr[lv + 1];
                const double *run = gs + d * 4 * PE;
                double sum = 0.0;
                for (int k = kb; k < ke; k += 4) {
                    const double w0 = run[k], w1 = k + 1 < ke ? run[k + 1] : 0.0, w2 = k + 2 < ke ? run[k + 2] : 0.0,
                                 w3 = k + 3 < ke ? run[k + 3] : 0.0;
                    sum += w0;
                    if (k + 1 < ke) sum += w1;
                    if (k + 2 < ke) sum += w2;
                    if (k + 3 < ke) sum += w3;
                }
                PT.gpart[(size_t)3 * vslot[lv] + d] = sum;
            }
        }
        EP_STAMP(3);
        __syncthreads();   // the next patch of this workgroup reuses xs / gs
        if (more) {
            cur = nxt;
            nxt.nv = nvN;
            nxt.gid0 = gidN;
            nxt.slot0 = slotN;
            nxt.cp0 = cpN;
        }
    }
    // inertia: sum_v 1/2 m_v |x_v - x~_v|^2 over this rank's vertex slice
    double ine = 0.0;
    if (fuse && !haveAlpha) {   // (a workgroup without a patch)
        finish_alpha();
        if (second && alpha < 0.0) return;   // (the whole workgroup: not a paired slot)
    }
    if (!SPEC && vfirst < v1) {
        if (fuse) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                ix[d] = ix[d] + alpha * ip[d];
                if (!second) x_out[3 * vfirst + d] = ix[d];
            }
        }
        const double dx = ix[0] - ixt[0], dy = ix[1] - ixt[1], dz = ix[2] - ixt[2];
        ine += (dx * dx + dy * dy + dz * dz) * im / 2.0;
    }
    for (int v = vfirst + gstride; !SPEC && v < v1; v += gstride) {
        double xv[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            xv[d] = x[3 * v + d];
            if (fuse) {
                xv[d] = xv[d] + alpha * sa.p[3 * v + d];
                if (!second) x_out[3 * v + d] = xv[d];
            }
        }
        const double dx = xv[0] - xt[3 * v], dy = xv[1] - xt[3 * v + 1], dz = xv[2] - xt[3 * v + 2];
        ine += (dx * dx + dy * dy + dz * dz) * mass[v] / 2.0;
    }
    // both block sums through one exchange
    const double we = wave_sum(acc), wi = wave_sum(ine);
    const int lane = tid & 63, w = tid >> 6;
    if (lane == 0) {
        sm[w] = we;
        sm[4 + w] = wi;
    }
    __syncthreads();
    if (tid == 0) {
        if (second) partials += 2 * ELEM_NB_MAX;
        partials[2 * bIdx] = (sm[0] + sm[1]) + (sm[2] + sm[3]);      // to be scaled by dtSq by the consumer
        if constexpr (SPEC) {
            if (bIdx >= sx.nbIne) partials[2 * bIdx + 1] = 0.0;   // (below: inertia_step_body's)
        } else {
            partials[2 * bIdx + 1] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
        }
    }
    EP_STAMP(4);
}

// SPEC: the inertia slice of the element pass -- trial point x_trial = x_cur + alpha p and sum_v 1/2 m_v |x_v - x~_v|^2
// (Optimizer.cpp:1023-1042, :1202-1215) -- for the vertices element workgroup j takes in the plain form (256 j + thread, then a
// grid stride of 256 nbElem), as a workgroup of its own in the speculative launch; alpha and p as there: phase 0 the unit step on
// p formed here (SpecDir), a retry slot the controller's alpha on the stored p.  sm: 8 doubles, sh: 1 + HIST_MAX doubles.
__device__ __forceinline__ void inertia_step_body(const double *__restrict__ mass, const double *__restrict__ xt, int nV,
                                                  double *__restrict__ partials, const DevLoop *__restrict__ ctl,
                                                  const StepArgs &sa, const SpecArgs &sx, int j, int nbElem, double *sm, double *sh)
{
    const int status = ctl->status, lphase = ctl->phase;
    const double lalpha = ctl->alpha;
    const double *__restrict__ x = ctl->x_cur;
    double *__restrict__ x_out = ctl->x_trial;
    if (status != 0) return;
    const bool spec = lphase == 0;
    SpecDir sd;
    sd.request(ctl, sx, spec);
    const int tid = threadIdx.x;
    const int gstride = nbElem * 256;
    const int vfirst = j * 256 + tid;
    double ix[3] = {0, 0, 0}, ixt[3] = {0, 0, 0}, ip[3] = {0, 0, 0}, im = 0.0;
    DirOps ipo[3];
    if (vfirst < nV) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            ix[d] = x[3 * vfirst + d];
            ixt[d] = xt[3 * vfirst + d];
            if (spec) sd.load(3 * vfirst + d, ipo[d]);
            else ip[d] = sa.p[3 * vfirst + d];
        }
        im = mass[vfirst];
    }
    double alpha = 1.0;
    if (spec) {
        if (tid < 64) sd.finish(ctl, sh + 1);
        __syncthreads();
    } else {
        alpha = lalpha;
    }
    double ine = 0.0;
    if (vfirst < nV) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (spec) ip[d] = sd.form(ipo[d], sh + 1);
            ix[d] = ix[d] + alpha * ip[d];
            x_out[3 * vfirst + d] = ix[d];
        }
        const double dx = ix[0] - ixt[0], dy = ix[1] - ixt[1], dz = ix[2] - ixt[2];
        ine += (dx * dx + dy * dy + dz * dz) * im / 2.0;
    }
    for (int v = vfirst + gstride; v < nV; v += gstride) {   // (no trip while nbElem >= ceil(nV / 256): the meshes this form is used on)
        double xv[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double pd;
            if (spec) {
                DirOps o;
                sd.load(3 * v + d, o);
                pd = sd.form(o, sh + 1);
            } else {
                pd = sa.p[3 * v + d];
            }
            xv[d] = x[3 * v + d] + alpha * pd;
            x_out[3 * v + d] = xv[d];
        }
        const double dx = xv[0] - xt[3 * v], dy = xv[1] - xt[3 * v + 1], dz = xv[2] - xt[3 * v + 2];
        ine += (dx * dx + dy * dy + dz * dz) * mass[v] / 2.0;
    }
    const double wi = wave_sum(ine);
    const int lane = tid & 63, w = tid >> 6;
    if (lane == 0) sm[4 + w] = wi;
    __syncthreads();
    if (tid == 0) partials[2 * j + 1] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
}


}  // namespace dotmi
