// kernels.hip -- hand-written CDNA4 (gfx950) kernels of the DOT time-step hot path.
//
// Conventions
//   * 64-lane wavefronts; workgroups of 256 threads (4 waves) unless noted.
//   * every floating-point reduction has a fixed shape (fixed block count, fixed tree), so results
//     are bit-identical run to run -- the reference is bit-deterministic.
//   * no FP atomics anywhere: scatter steps are written in gather form over precomputed CSR lists
//     (the reference's own vFLoc form, Energy.cpp:543-563).
//   * reductions leave per-block partials; the consumer (next kernel's prologue, or the host) sums
//     them in index order.  That removes every "final reduce" launch from the L-BFGS loop.
//
// Reference map (paths relative to /root/reference/src):
//   elem_energy_grad_kernel   Energy.cpp:294-423 (F, SVD, Psi), :910-972 (P-hat, P, element gradient),
//                             Optimizer.cpp:1202-1215 (inertia energy)
//   vertex_gather_kernel      Energy.cpp:543-563, Optimizer.cpp:1239-1252, DOTTimeStepper.cpp:474-494
//   build_q / build_p         DOTTimeStepper.cpp:386-400, :455-467 (two-loop recursion, compact form)
//   backsolve_kernel + reduce_partial_p + merge
//                             DOTTimeStepper.cpp:406-450 (subdomain back-solve, average by dup)
//   loop_control_kernel       Optimizer.cpp:806-833 (line search), DOTTimeStepper.cpp:474-494 (history),
//                             Optimizer.cpp:317-330 (stopping test) -- the host loop's control flow, on device
//   tile_task / tile_flow / tile_gemm (schedule: tile_factor.hpp)
//                             CHOLMODSolver.cpp:143 factorize, as a block-sparse inverse-Cholesky on 64 x 64 tiles
//   spmv_dots / step_forward  Optimizer.cpp:1076-1093 (alpha_0), :1023-1042 (x = x0 + alpha p)
//   elem_hessian_kernel       Energy.cpp:738-777, :1129-1270, IglUtils.hpp:466-479
//   assemble_kernel           DOTTimeStepper.cpp:588-613, IglUtils.hpp:143-220
//   dense_fill_kernel         DOTTimeStepper.cpp:619-797 (== principal sub-matrix of the global H)
#include "dotmi_internal.hpp"
#include <hip/hip_ext.h>
#include "elem_math.hpp"

namespace dotmi {

// This file is compiled TWICE (Makefile): as it stands into kernels.o, and with -DDOTMI_PAIR_TU into kernels_pair.o, whose only
// exported functions are launch_elem_energy_grad_pair / launch_gemv_pair -- the element pass and the back-solve + controller
// launch of a step with PAIRED line-search trials.  The plain unit compiles every paired branch away, so
// its kernels are the ones it had before the pairing existed: the loop's kernels reacted to ANY change of their code or
// argument layout by 1-2 % (profiles/r05_paired_trials.txt F), and the steps that never pair should not pay for those that do.
#ifdef DOTMI_PAIR_TU
#define launch_elem_energy_grad launch_elem_energy_grad_pair
#define launch_gemv launch_gemv_pair
#define PAIR_ARG_DECL0 , const double *__restrict__ partE2 = nullptr
#define PAIR_ARG_DECL , const double *__restrict__ partE2
#define PAIR_INIT_MASK &1
#else
#define PAIR_ARG_DECL0
#define PAIR_ARG_DECL
#define PAIR_INIT_MASK
#endif

// ------------------------------------------------------------------------------------------------
// reductions
// ------------------------------------------------------------------------------------------------
// held-vertex lists (dotmi_internal.hpp VList): logical index -> vertex / scalar dof
__device__ __forceinline__ int vl_count3(const VList &L, int n3) { return L.v ? 3 * L.n : n3; }
__device__ __forceinline__ int vl_dof(const VList &L, int i) { return L.v ? 3 * L.v[i / 3] + i % 3 : i; }
__device__ __forceinline__ int vl_vtx(const VList &L, int j) { return L.v ? L.v[j] : j; }
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

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
__device__ long long g_ep_prof[8192][6];
extern "C" int dotmi_debug_ep_prof(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ep_prof), sizeof(long long) * 6 * (size_t)n);
}
#define EP_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_ep_prof[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define EP_STAMP(i) do { } while (0)
#endif
template <int MAT, bool GRAD, int EPT, bool FUSE, bool PIPE>
__global__ __launch_bounds__(256) void elem_patch_kernel(DevPatches PT, const double *__restrict__ mass,
                                                         const double *__restrict__ x, const double *__restrict__ xt,
                                                         int v0, int v1, double dtSq, double *__restrict__ partials,
                                                         const DevLoop *__restrict__ ctl, StepArgs sa)
{
    extern __shared__ double lds[];
    __shared__ double sm[8];
    __shared__ double sh_alpha;
    // sa.p != nullptr (device loop): the line-search step x_trial = x_cur + alpha p (step_forward_kernel's statements) is
    // taken HERE: every position this kernel reads is formed on the fly, the trial point is written by the inertia loop
    // (which visits every vertex exactly once), and alpha comes from the SpMV partials in wave 0's prologue -- one launch
    // and one pass over x, p less per line-search trial
    constexpr bool fuse = FUSE;   // (a template parameter: the plain instantiation keeps its register count)
    double *__restrict__ x_out = nullptr;
    if (ctl) {
        if (ctl->status != 0) return;
        x = fuse ? ctl->x_cur : ctl->x_trial;
        x_out = ctl->x_trial;
    }
#ifdef DOTMI_PAIR_TU
    // paired trial (StepArgs::alpha_min < 0): the launch is twice as wide; workgroup nbP + b mirrors workgroup b on the FULL step
    const bool pairedLaunch = fuse && sa.alpha_min < 0.0;
    const double alphaMin = pairedLaunch ? -sa.alpha_min : sa.alpha_min;
    const int nbP = pairedLaunch ? (int)gridDim.x / 2 : (int)gridDim.x;
    const bool second = pairedLaunch && (int)blockIdx.x >= nbP;
    const int bIdx = second ? (int)blockIdx.x - nbP : (int)blockIdx.x;
#define EP_BIDX bIdx
#define EP_NBP nbP
#define EP_AMIN alphaMin
#define EP_GRAD (GRAD && !second)
#define EP_LEAVE_IF_NOT_PAIRED() do { if (second && alpha < 0.0) return; } while (0)   /* (the whole workgroup: not a paired slot) */
#define EP_FIRST_HALF(stmt) do { if (!second) { stmt; } } while (0)
#else
#define EP_BIDX blockIdx.x
#define EP_NBP gridDim.x
#define EP_AMIN sa.alpha_min
#define EP_GRAD GRAD
#define EP_LEAVE_IF_NOT_PAIRED() do { } while (0)
#define EP_FIRST_HALF(stmt) stmt
#endif
    double pgv[NB_RED / 64], pHpv[NB_RED / 64];
    const bool usePart = fuse && ctl->phase == 0;   // a retry steps with the halved alpha the controller left
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
        if (threadIdx.x < 64) {
            double a = ctl->alpha;
            if (usePart) {
                double pg = 0.0, pHp = 0.0;
#pragma unroll
                for (int u = 0; u < NB_RED / 64; ++u) {
                    pg += pgv[u];
                    pHp += pHpv[u];
                }
                pg = __shfl(wave_sum(pg), 0, 64);
                pHp = __shfl(wave_sum(pHp), 0, 64);
                a = fmax(EP_AMIN, fmin(1.0, -pg / pHp));  // Optimizer.cpp:1085
            }
            if (threadIdx.x == 0) {
#ifdef DOTMI_PAIR_TU
                // paired: alpha_0 < 1 (the quadratic model's minimum lies inside the unit step) -- the full step's energy
                // from the second half of the launch, the half step in full from the first
                const bool pair = pairedLaunch && usePart && a < 1.0 && a / 2.0 > 0.0 && ctl->pairCtr[pair_band(a)] >= 3;
                sh_alpha = pair ? (second ? a : a / 2.0) : (second ? -1.0 : a);
                if (blockIdx.x == 0) {
                    sa.alpha_out[0] = pair ? a / 2.0 : a;
                    if (pairedLaunch) sa.alpha_out[1] = pair ? a : 0.0;
                }
#else
                sh_alpha = a;
                if (blockIdx.x == 0) *sa.alpha_out = a;
#endif
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
    const int gstride = EP_NBP * blockDim.x;
    const int vfirst = v0 + EP_BIDX * blockDim.x + tid;
    double ix[3] = {0, 0, 0}, ixt[3] = {0, 0, 0}, ip[3] = {0, 0, 0}, im = 0.0;
    if (vfirst < v1) {
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
                if (fuse) o.pv[d] = sa.p[3 * o.gid0 + d];
            }
        }
    };
    PatchOps cur, nxt;
    int nvN = 0, gidN = -1, slotN = 0;   // ids of the patch after next
    unsigned short cpN = 0;
    if ((int)EP_BIDX < PT.nPatches) {
        issue_ids(EP_BIDX, cur);
        issue_ops(EP_BIDX, cur);
        issue_pos(cur);
        if (PIPE && (int)(EP_BIDX + EP_NBP) < PT.nPatches) issue_ids(EP_BIDX + EP_NBP, nxt);
    }
    for (int p = EP_BIDX; p < PT.nPatches; p += EP_NBP) {
        // PIPE: the instantiation for meshes whose workgroups walk several patches (the prefetched set costs ~50 registers,
        // which the one-patch-per-workgroup meshes keep for occupancy)
        const int pn = p + EP_NBP, pn2 = pn + EP_NBP;
        const bool more = PIPE && pn < PT.nPatches, more2 = PIPE && pn2 < PT.nPatches;
        if (!PIPE && p != (int)EP_BIDX) {
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
        if (fuse && !haveAlpha) {
            finish_alpha();
            EP_LEAVE_IF_NOT_PAIRED();
        }
        if (cur.gid0 >= 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) xs[3 * tid + d] = fuse ? cur.xv[d] + alpha * cur.pv[d] : cur.xv[d];
        }
        for (int lv = tid + 256; lv < nv; lv += 256) {
            const int gid = PT.pv_gid[vb + lv];
#pragma unroll
            for (int d = 0; d < 3; ++d) xs[3 * lv + d] = fuse ? x[3 * gid + d] + alpha * sa.p[3 * gid + d] : x[3 * gid + d];
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
                if (EP_GRAD) {
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
                if (EP_GRAD) {
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
            if (EP_GRAD) {
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
        if (EP_GRAD) {
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
        EP_LEAVE_IF_NOT_PAIRED();
    }
    if (vfirst < v1) {
        if (fuse) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                ix[d] = ix[d] + alpha * ip[d];
                EP_FIRST_HALF(x_out[3 * vfirst + d] = ix[d]);
            }
        }
        const double dx = ix[0] - ixt[0], dy = ix[1] - ixt[1], dz = ix[2] - ixt[2];
        ine += (dx * dx + dy * dy + dz * dz) * im / 2.0;
    }
    for (int v = vfirst + gstride; v < v1; v += gstride) {
        double xv[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            xv[d] = x[3 * v + d];
            if (fuse) {
                xv[d] = xv[d] + alpha * sa.p[3 * v + d];
                EP_FIRST_HALF(x_out[3 * v + d] = xv[d]);
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
#ifdef DOTMI_PAIR_TU
        if (second) partials += 2 * ELEM_NB_MAX;
#endif
        partials[2 * EP_BIDX] = (sm[0] + sm[1]) + (sm[2] + sm[3]);      // to be scaled by dtSq by the consumer
        partials[2 * EP_BIDX + 1] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
    }
    EP_STAMP(4);
}

void launch_elem_energy_grad(const DevMesh &M, const DevPatches &PT, int mat, double dtSq, const double *x,
                             const double *xt, int v0, int v1, int grad, double *partials, int *nblocks_out,
                             hipStream_t st, const DevLoop *ctl, const StepArgs *step)
{
    StepArgs sa{nullptr, nullptr, nullptr, 0.0};
    if (step && ctl) sa = *step;
#ifdef DOTMI_PAIR_TU
    const bool paired = sa.p && sa.alpha_min < 0.0 && grad;
#endif
    // at most elem_wg_cap() workgroups take the patches (as many as are resident at once): beyond that a workgroup walks several
    // patches and prefetches the next one's operands (elem_patch_kernel)
    // (the instantiation with the step inside: two per CU; a handle whose loop uses it fixes 512 for all of them, PT.wgCap)
    const int cap = PT.wgCap > 0 ? PT.wgCap : ((step && ctl) ? 512 : elem_wg_cap(mat));
    const bool pipe = PT.nPatches > cap;
    int nb = pipe ? cap : PT.nPatches;
    const int nbv = (v1 - v0 + 255) / 256;
    if (nb < nbv && !pipe) nb = nbv;    // the inertia loop likes one vertex per thread on small meshes
    if (nb > ELEM_NB_MAX) nb = ELEM_NB_MAX;
    if (nb < 1) nb = 1;
    *nblocks_out = nb;
#ifdef DOTMI_PAIR_TU
    const int nbLaunch = paired ? 2 * nb : nb;
#else
#define nbLaunch nb
#endif
    const int ept = PT.PE / 256;
    const size_t shm = sizeof(double) * ((size_t)3 * PT.PV + (grad ? (size_t)12 * PT.PE : 0)) +
                       (grad ? 2 * (size_t)((PT.PV + 1 + 3) & ~3) + 4 * (size_t)PT.PV : 0);
#define DM_LAUNCH(MATV, GRADV, EPTV)                                                                            \
    do {                                                                                                            \
        if (sa.p && pipe)                                                                                           \
            hipLaunchKernelGGL((elem_patch_kernel<MATV, GRADV, EPTV, true, true>), dim3(nbLaunch), dim3(256), shm, st, PT, M.mass, \
                               x, xt, v0, v1, dtSq, partials, ctl, sa);                                             \
        else if (sa.p)                                                                                              \
            hipLaunchKernelGGL((elem_patch_kernel<MATV, GRADV, EPTV, true, false>), dim3(nbLaunch), dim3(256), shm, st, PT, M.mass, \
                               x, xt, v0, v1, dtSq, partials, ctl, sa);                                             \
        else if (pipe)                                                                                              \
            hipLaunchKernelGGL((elem_patch_kernel<MATV, GRADV, EPTV, false, true>), dim3(nbLaunch), dim3(256), shm, st, PT, M.mass, \
                               x, xt, v0, v1, dtSq, partials, ctl, sa);                                             \
        else                                                                                                        \
            hipLaunchKernelGGL((elem_patch_kernel<MATV, GRADV, EPTV, false, false>), dim3(nbLaunch), dim3(256), shm, st, PT, M.mass, \
                               x, xt, v0, v1, dtSq, partials, ctl, sa);                                             \
    } while (0)
#define DM_LAUNCH_E(MATV, GRADV)      \
    do {                              \
        if (ept == 1) DM_LAUNCH(MATV, GRADV, 1); \
        else DM_LAUNCH(MATV, GRADV, 2);          \
    } while (0)
    if (mat == 0) {
        if (grad) DM_LAUNCH_E(0, true);
        else DM_LAUNCH_E(0, false);
    } else {
        if (grad) DM_LAUNCH_E(1, true);
        else DM_LAUNCH_E(1, false);
    }
#undef DM_LAUNCH_E
#undef DM_LAUNCH
#ifndef DOTMI_PAIR_TU
#undef nbLaunch
#endif
}

// ------------------------------------------------------------------------------------------------
// vertex gather of element gradients (+ inertia), new L-BFGS pair and its statistics
// partial layout per block (m = L.m):
//   [0] |g_new|^2   [1] y_new.s_new   [2] s_new.g_new
//   [3+i] s_i.y_new   [3+HIST_MAX+j] s_new.y_j   [3+2*HIST_MAX+i] s_i.g_new
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pair_stats_accum(int k, double gn, double sn, double yn, const LbfgsArgs &L,
                                                 double (&acc)[RED_K])
{
    acc[0] += gn * gn;
    acc[1] += yn * sn;
    acc[2] += sn * gn;
    const int m = L.m;
#pragma unroll
    for (int i = 0; i < HIST_MAX; ++i) {
        if (i < m) {
            const double si = L.s[i][k], yi = L.y[i][k];
            acc[3 + i] += si * yn;
            acc[3 + HIST_MAX + i] += sn * yi;
            acc[3 + 2 * HIST_MAX + i] += si * gn;
        }
    }
}

// Wave totals of 8 values per lane with a transposed butterfly (10 cross-lane steps instead of 48): afterwards every
// lane holds the total of value number lane >> 3 (lane bits 5,4,3 select the value, bits 2,1,0 were summed last).
__device__ __forceinline__ double wave_sum8_transposed(const double (&d)[8], int lane)
{
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
    return e1;
}

// block-sum the first nvals accumulators (nvals uniform over the block) and store them as this block's partial row.
// sm: 4*RED_K doubles.
__device__ __forceinline__ void write_partials(double (&acc)[RED_K], int nvals, double *partials, double *sm)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    constexpr int NG = (RED_K + 7) / 8;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (8 * g >= nvals) break;   // groups of 8 values; the ones nobody asked for are not reduced
        double d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = (8 * g + k < RED_K) ? acc[8 * g + k < RED_K ? 8 * g + k : 0] : 0.0;
        const double s = wave_sum8_transposed(d, lane);
        const int j = 8 * g + (lane >> 3);
        if ((lane & 7) == 0 && j < RED_K) sm[w * RED_K + j] = s;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < nvals)
        partials[(size_t)blockIdx.x * RED_K + t] = (sm[t] + sm[RED_K + t]) + (sm[2 * RED_K + t] + sm[3 * RED_K + t]);
}

// sum over the 8 lanes of an aligned lane group (fixed butterfly => deterministic); all 8 get the total
__device__ __forceinline__ double group8_sum(double v)
{
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// One lane per scalar degree of freedom k = 3 v + d: the vertex's per-patch partial gradients (usually 1-4 of them,
// contiguous in gpart, ascending patch) are added in that order, then the inertia term; the new L-BFGS pair and its
// statistics follow from the same registers.  GATHER_R dofs per lane and trip, one grid stride apart, so that the two
// dependent round trips of a trip (partial range -> partials) are paid once for all of them.
constexpr int GATHER_R = 4;
constexpr int GATHER_P = 4;   // partials requested together; a vertex with more takes further rounds
template <bool DEV>
__global__ __launch_bounds__(256) void vertex_gather_kernel(
    int nV, const int2 *__restrict__ pp_rng, const double *__restrict__ gpart,
    const uint8_t *__restrict__ fixed, const double *__restrict__ mass, GatherArgs a, LbfgsArgs L,
    double *__restrict__ partials, const DevLoop *__restrict__ ctl)
{
    __shared__ double sm[4 * RED_K];
    if constexpr (DEV) {
        if (ctl->status != 0) return;
        a.x = ctl->x_trial;
        a.g_old = ctl->g_cur;
        if (!a.stage) a.g_new = ctl->g_trial;   // stage: this rank's partial gradient goes to the buffer the host named
        a.s_new = ctl->S[ctl->slot];
        a.y_new = ctl->Y[ctl->slot];
    }
    double *__restrict__ hs_new = nullptr;
    if constexpr (DEV) {
        if (a.hp) hs_new = ctl->HS[ctl->slot];
    }
    const LbfgsArgs &Lr = [&]() -> const LbfgsArgs & {
        if constexpr (DEV) return ctl->L;
        else return L;
    }();
    double acc[RED_K];
#pragma unroll
    for (int j = 0; j < RED_K; ++j) acc[j] = 0.0;
    const double alpha = a.make_pair ? *a.alpha_dev : 0.0;
    const VList vl{a.vlist, a.nlist};   // owner exchange: only the held vertices are visited
    const int n = vl_count3(vl, 3 * nV), G = gridDim.x * blockDim.x;
    constexpr int R = GATHER_R;
    for (int kbase = blockIdx.x * blockDim.x + threadIdx.x; kbase < n; kbase += R * G) {
        double gn[R], ine[R], gold[R], pk[R], si[R][HIST_MAX], yi[R][HIST_MAX];
        int kb[R], ke[R], dd[R], cb[R], ce[R], kk[R], kd[R];
        bool live[R];
#pragma unroll
        for (int u = 0; u < R; ++u) {
            live[u] = kbase + u * G < n;
            const int k = live[u] ? vl_dof(vl, kbase + u * G) : 0;
            kk[u] = k;
            gn[u] = ine[u] = gold[u] = pk[u] = 0.0;
            kb[u] = ke[u] = dd[u] = cb[u] = ce[u] = kd[u] = 0;
            if (live[u]) {
                const int v = k / 3;
                dd[u] = k - 3 * v;
                if (a.pre) kd[u] = a.kind[v];
                if (a.rpad && !(kd[u] & 2)) {
                    cb[u] = a.vp_ptr[v];
                    ce[u] = a.vp_ptr[v + 1];
                }
                const bool fx = fixed[v];
                if (!fx) {   // fixed rows of the gradient are zero (Optimizer.cpp:1239-1252)
                    const int2 r = pp_rng[v];
                    kb[u] = r.x;
                    ke[u] = r.y;
                    if (a.ownMask ? a.ownMask[v] != 0 : (v >= a.iv0 && v < a.iv1)) ine[u] = mass[v] * (a.x[k] - a.xt[k]);
                }
                if (a.make_pair) {
                    gold[u] = a.g_old[k];
                    pk[u] = a.p[k];
#pragma unroll
                    for (int i = 0; i < HIST_MAX; ++i) {
                        si[u][i] = (i < Lr.m) ? Lr.s[i][k] : 0.0;
                        yi[u][i] = (i < Lr.m) ? Lr.y[i][k] : 0.0;
                    }
                }
            }
        }
        // the first copies' padded positions are requested before the partial sums (off the stores' dependent chain)
        constexpr int VC = 4;
        int vo[R][VC];
#pragma unroll
        for (int u = 0; u < R; ++u)
#pragma unroll
            for (int c = 0; c < VC; ++c) vo[u][c] = (cb[u] + c < ce[u]) ? a.vp_off[cb[u] + c] : 0;
        int nkmax = 0;
#pragma unroll
        for (int u = 0; u < R; ++u) nkmax = max(nkmax, ke[u] - kb[u]);
        for (int t = 0; t < nkmax; t += GATHER_P) {
            double w[R][GATHER_P];
#pragma unroll
            for (int u = 0; u < R; ++u)
#pragma unroll
                for (int j = 0; j < GATHER_P; ++j)
                    if (kb[u] + t + j < ke[u]) w[u][j] = gpart[(size_t)3 * (kb[u] + t + j) + dd[u]];
#pragma unroll
            for (int u = 0; u < R; ++u)
#pragma unroll
                for (int j = 0; j < GATHER_P; ++j)
                    if (kb[u] + t + j < ke[u]) gn[u] += w[u][j];
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (!live[u]) continue;
            const int k = kk[u];
            const double g = gn[u] + ine[u];
            if (kd[u] & 2) {
                // Owner exchange with the statistics in the gradient's packet (a.pre), at a vertex other ranks hold too: g is
                // this rank's PART of the gradient there -- it goes to the buffer the packet is filled from, and this rank's
                // share of the sums that are linear in the gradient to the partials (all but |g|^2: the sum over the ranks of
                // (part) . v is the whole product; the terms without the new gradient are the owner's).  The vertex' pair,
                // right-hand side entries and H s are formed after the exchange (pair_stats over the shared vertices).
                a.gshare[k] = g;
                const double w = (kd[u] & 1) ? 1.0 : 0.0;
                const double sn = alpha * pk[u];
                const double yp = g - w * gold[u];
                acc[1] += yp * sn;
                acc[2] += sn * g;
#pragma unroll
                for (int i = 0; i < HIST_MAX; ++i)
                    if (i < Lr.m) {
                        acc[3 + i] += si[u][i] * yp;
                        acc[3 + HIST_MAX + i] += w * (sn * yi[u][i]);
                        acc[3 + 2 * HIST_MAX + i] += si[u][i] * g;
                    }
                continue;
            }
            a.g_new[k] = g;
#pragma unroll
            for (int c = 0; c < VC; ++c)
                if (cb[u] + c < ce[u]) a.rpad[vo[u][c] + dd[u]] = -g;
            for (int c = cb[u] + VC; c < ce[u]; ++c) a.rpad[a.vp_off[c] + dd[u]] = -g;
            if (a.make_pair) {
                const double sn = alpha * pk[u];
                const double yn = g - gold[u];
                a.s_new[k] = sn;
                a.y_new[k] = yn;
                if (hs_new) hs_new[k] = alpha * a.hp[k];   // H s_new = alpha H p
                acc[0] += g * g;
                acc[1] += yn * sn;
                acc[2] += sn * g;
#pragma unroll
                for (int i = 0; i < HIST_MAX; ++i)
                    if (i < Lr.m) {
                        acc[3 + i] += si[u][i] * yn;
                        acc[3 + HIST_MAX + i] += sn * yi[u][i];
                        acc[3 + 2 * HIST_MAX + i] += si[u][i] * g;
                    }
            } else {
                acc[0] += g * g;
            }
        }
    }
    write_partials(acc, a.make_pair ? RED_K : 1, partials, sm);
}

void launch_vertex_gather(const DevMesh &M, const DevPatches &PT, const GatherArgs &a, const LbfgsArgs &L,
                          double *partials, hipStream_t st, const DevLoop *ctl)
{
    if (ctl)
        hipLaunchKernelGGL(vertex_gather_kernel<true>, dim3(NB_RED), dim3(256), 0, st, M.nV, PT.pp_rng, PT.gpart,
                           M.fixed, M.mass, a, L, partials, ctl);
    else
        hipLaunchKernelGGL(vertex_gather_kernel<false>, dim3(NB_RED), dim3(256), 0, st, M.nV, PT.pp_rng, PT.gpart,
                           M.fixed, M.mass, a, L, partials, ctl);
}

// gsrc != nullptr: the summed gradient is read from gsrc and copied to g_new on the way (device loop: the all-reduce
// runs on a staging buffer because the trial gradient's address is only known on the device)
template <bool DEV>
__global__ __launch_bounds__(256) void pair_stats_kernel(int n, GatherArgs a, LbfgsArgs L, const double *__restrict__ gsrc,
                                                         double *__restrict__ partials, const DevLoop *__restrict__ ctl)
{
    __shared__ double sm[4 * RED_K];
    if constexpr (DEV) {
        if (ctl->status != 0) return;
        a.g_old = ctl->g_cur;
        a.g_new = ctl->g_trial;
        a.s_new = ctl->S[ctl->slot];
        a.y_new = ctl->Y[ctl->slot];
    }
    const LbfgsArgs &Lr = [&]() -> const LbfgsArgs & {
        if constexpr (DEV) return ctl->L;
        else return L;
    }();
    double acc[RED_K];
#pragma unroll
    for (int j = 0; j < RED_K; ++j) acc[j] = 0.0;
    const double alpha = *a.alpha_dev;
    const int stride = gridDim.x * blockDim.x;
    double *__restrict__ hs_new = nullptr;
    if constexpr (DEV) {
        if (a.hp) hs_new = ctl->HS[ctl->slot];
    }
    const VList vl{a.vlist, a.nlist};
    const int cnt = vl_count3(vl, n);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        const int k = vl_dof(vl, i);
        const double gn = gsrc ? gsrc[k] : a.g_new[k];
        if (gsrc) a.g_new[k] = gn;
        const double sn = alpha * a.p[k];
        const double yn = gn - a.g_old[k];
        a.s_new[k] = sn;
        a.y_new[k] = yn;
        // early order on the sharded element pass: what vertex_gather does on one rank happens here, on the SUMMED gradient --
        // -g into the padded right-hand sides of this rank's subdomains that hold the vertex, H s_new = alpha H p
        if (a.rpad) {
            const int v = k / 3, dd = k - 3 * v;
            for (int c = a.vp_ptr[v]; c < a.vp_ptr[v + 1]; ++c) a.rpad[a.vp_off[c] + dd] = -gn;
        }
        if (hs_new) hs_new[k] = alpha * a.hp[k];
        if (partials && (!a.ownMask || a.ownMask[k / 3])) pair_stats_accum(k, gn, sn, yn, Lr, acc);
    }
    if (partials) write_partials(acc, RED_K, partials, sm);   // (nullptr: the statistics came with the packet, stats_pre_kernel)
}

void launch_pair_stats(int n, const GatherArgs &a, const LbfgsArgs &L, double *partials, hipStream_t st,
                       const double *gsrc, const DevLoop *ctl)
{
    if (ctl) hipLaunchKernelGGL(pair_stats_kernel<true>, dim3(NB_RED), dim3(256), 0, st, n, a, L, gsrc, partials, ctl);
    else hipLaunchKernelGGL(pair_stats_kernel<false>, dim3(NB_RED), dim3(256), 0, st, n, a, L, gsrc, partials, ctl);
}

// ------------------------------------------------------------------------------------------------
// two-loop recursion in compact form
//   loop 1:  xi_i = (s_i . q_i)/ys_i with s_i.q_i = -b_i - sum_{j>i} xi_j (s_i.y_j)   (host, FP64)
//            q = -g - sum_j xi_j y_j
//   loop 2:  beta_i = (y_i . p_i)/ys_i with y_i.p_i = c_i + sum_{j<i} delta_j (s_j.y_i), c_i = y_i.z
//            delta_i = xi_i - beta_i ;  p = z + sum_j delta_j s_j
// identical in exact arithmetic to DOTTimeStepper.cpp:386-400 / :455-467
// ------------------------------------------------------------------------------------------------
template <bool DEV>
__global__ __launch_bounds__(256) void build_q_kernel(int n, const double *__restrict__ g, LbfgsArgs L,
                                                      XiArgs X, double *__restrict__ q,
                                                      const DevLoop *__restrict__ ctl)
{
    if constexpr (DEV) {
        if (ctl->status != 0 || ctl->phase != 0) return;
        g = ctl->g_cur;
    }
    const LbfgsArgs &Lr = [&]() -> const LbfgsArgs & {
        if constexpr (DEV) return ctl->L;
        else return L;
    }();
    const XiArgs &Xr = [&]() -> const XiArgs & {
        if constexpr (DEV) return ctl->X;
        else return X;
    }();
    const int stride = gridDim.x * blockDim.x;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        double v = -g[k];
        // newest to oldest, as the reference subtracts them
#pragma unroll
        for (int j = HIST_MAX - 1; j >= 0; --j)
            if (j < Lr.m) v -= Xr.xi[j] * Lr.y[j][k];
        q[k] = v;
    }
}

void launch_build_q(int n, const double *g, const LbfgsArgs &L, const double *xi_host, double *q,
                    hipStream_t st, const DevLoop *ctl)
{
    XiArgs X;
    for (int i = 0; i < HIST_MAX; ++i) X.xi[i] = (xi_host && i < L.m) ? xi_host[i] : 0.0;
    int nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (ctl) hipLaunchKernelGGL(build_q_kernel<true>, dim3(nb), dim3(256), 0, st, n, g, L, X, q, ctl);
    else hipLaunchKernelGGL(build_q_kernel<false>, dim3(nb), dim3(256), 0, st, n, g, L, X, q, ctl);
}

// the same q, written straight into the padded per-subdomain right-hand sides the back-solve tiles read (a vertex
// shared by k subdomains is written k times): the tiles then start from ONE contiguous load instead of an index load
// followed by scattered gathers, repeated by every tile of the subdomain
template <bool DEV>
__global__ __launch_bounds__(256) void build_qpad_kernel(int total, const int *__restrict__ dofmap,
                                                         const double *__restrict__ g, LbfgsArgs L, XiArgs X,
                                                         double *__restrict__ rpad, const DevLoop *__restrict__ ctl,
                                                         int spec)
{
    if constexpr (DEV) {
        if (ctl->status != 0 || (ctl->phase != 0 && spec != 2)) return;
        g = spec == 2 ? ctl->g_trial : ctl->g_cur;
    }
    const LbfgsArgs &Lr = [&]() -> const LbfgsArgs & {
        if constexpr (DEV) return ctl->L;
        else return L;
    }();
    const XiArgs &Xr = [&]() -> const XiArgs & {
        if constexpr (DEV) return ctl->X;
        else return X;
    }();
    const int m = spec ? 0 : Lr.m;   // early back-solve: the history terms are applied after the solve (merge_early)
    const int stride = gridDim.x * blockDim.x;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
        const int d = dofmap[k];
        double v = 0.0;
        if (d >= 0) {
            v = -g[d];
#pragma unroll
            for (int j = HIST_MAX - 1; j >= 0; --j)
                if (j < m) v -= Xr.xi[j] * Lr.y[j][d];
        }
        rpad[k] = v;
    }
}

void launch_build_qpad(const DevParts &P, const double *g, const LbfgsArgs &L, const double *xi_host, hipStream_t st,
                       const DevLoop *ctl, int spec)
{
    XiArgs X;
    for (int i = 0; i < HIST_MAX; ++i) X.xi[i] = (xi_host && i < L.m) ? xi_host[i] : 0.0;
    const int total = P.nParts * P.nmax;
    if (total <= 0) return;
    int nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (ctl) hipLaunchKernelGGL(build_qpad_kernel<true>, dim3(nb), dim3(256), 0, st, total, P.dofmap, g, L, X, P.rpad, ctl, spec);
    else hipLaunchKernelGGL(build_qpad_kernel<false>, dim3(nb), dim3(256), 0, st, total, P.dofmap, g, L, X, P.rpad, ctl, 0);
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

template <bool DEV>
__global__ __launch_bounds__(256) void build_p_kernel(int n, const double *__restrict__ z, LbfgsArgs L,
                                                      XiArgs X, const double *__restrict__ c_partials,
                                                      int c_blocks, double *__restrict__ p,
                                                      const DevLoop *__restrict__ ctl)
{
    __shared__ double delta[HIST_MAX];
    if constexpr (DEV) {
        if (ctl->status != 0 || ctl->phase != 0) return;
    }
    const LbfgsArgs &Lr = [&]() -> const LbfgsArgs & {
        if constexpr (DEV) return ctl->L;
        else return L;
    }();
    const XiArgs &Xr = [&]() -> const XiArgs & {
        if constexpr (DEV) return ctl->X;
        else return X;
    }();
    // the body's operands do not depend on the coefficients: they are requested before the prologue below (partial sums
    // and the short recurrence), so their latency is hidden behind it.  The grid covers n in one trip (launch_build_p).
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = Lr.m;
    double zv = 0.0, sv[HIST_MAX];
    if (k < n) {
        zv = z[k];
#pragma unroll
        for (int j = 0; j < HIST_MAX; ++j) sv[j] = (j < m) ? Lr.s[j][k] : 0.0;
    }
    if (threadIdx.x < 64) {
        // all partial columns first (independent loads in flight together), then the short recurrence
        double c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = 0.0;
        static_assert(HIST_MAX <= 8, "one transposed butterfly");
        for (int b = threadIdx.x; b < c_blocks; b += 64) {
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i) c[i] += c_partials[(size_t)b * RED_K + i];  // columns >= m: unused
        }
        // wave totals of the (up to 8) columns in 10 cross-lane steps; lane 8 i holds column i
        const double tot = wave_sum8_transposed(c, threadIdx.x);
        double ct[HIST_MAX], rys[HIST_MAX];
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i) {
            ct[i] = __shfl(tot, 8 * i, 64);
            rys[i] = (i < m) ? 1.0 / Lr.ys[i] : 0.0;   // independent divisions, off the recurrence's dependent chain
        }
        double d[HIST_MAX];
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i) {
            d[i] = 0.0;
            if (i < m) {
                double yp = ct[i];
#pragma unroll
                for (int j = 0; j < HIST_MAX; ++j)
                    if (j < i) yp += d[j] * Lr.sy[j][i];
                d[i] = Xr.xi[i] - yp * rys[i];
            }
            if (threadIdx.x == 0) delta[i] = d[i];
        }
    }
    __syncthreads();
    if (k < n) {
        double v = zv;
#pragma unroll
        for (int j = 0; j < HIST_MAX; ++j)
            if (j < m) v += sv[j] * delta[j];
        p[k] = v;
    }
}

void launch_build_p(int n, const double *z, const LbfgsArgs &L, const double *c_partials,
                    const double *xi_host, double *p, hipStream_t st, const DevLoop *ctl)
{
    XiArgs X;
    for (int i = 0; i < HIST_MAX; ++i) X.xi[i] = (xi_host && i < L.m) ? xi_host[i] : 0.0;
    const int nb = (n + 255) / 256;   // one element per thread (the kernel has no grid-stride loop)
    if (ctl) hipLaunchKernelGGL(build_p_kernel<true>, dim3(nb), dim3(256), 0, st, n, z, L, X, c_partials, NB_RED, p, ctl);
    else hipLaunchKernelGGL(build_p_kernel<false>, dim3(nb), dim3(256), 0, st, n, z, L, X, c_partials, NB_RED, p, ctl);
}

// generic multi-dot: partials[b][i] = sum_k v[k]*vecs_i[k]   (used on the multi-GPU path)
struct VecList {
    const double *v[HIST_MAX];
};
__global__ __launch_bounds__(256) void multidot_kernel(int n, const double *__restrict__ v, VecList W, int m,
                                                       double *__restrict__ partials)
{
    __shared__ double sm[4 * RED_K];
    double acc[RED_K];
#pragma unroll
    for (int j = 0; j < RED_K; ++j) acc[j] = 0.0;
    const int stride = gridDim.x * blockDim.x;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) {
        const double vk = v[k];
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i)
            if (i < m) acc[i] += vk * W.v[i][k];
    }
    write_partials(acc, m, partials, sm);
}

void launch_multidot(int n, const double *v, const double *const *vecs, int m, double *partials,
                     hipStream_t st)
{
    VecList W;
    for (int i = 0; i < HIST_MAX; ++i) W.v[i] = i < m ? vecs[i] : nullptr;
    hipLaunchKernelGGL(multidot_kernel, dim3(NB_RED), dim3(256), 0, st, n, v, W, m, partials);
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

__device__ __forceinline__ void loop_control_body(DevLoop *__restrict__ ctl, const double *__restrict__ partE, int nbE,
                                  const double *__restrict__ partR, const double *__restrict__ alpha_dev,
                                  int *__restrict__ flags_host, int init PAIR_ARG_DECL0);

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
#ifdef DOTMI_PAIR_TU
        loop_control_body(ca.ctl, ca.partE, ca.nbE, ca.partR, ca.alpha_dev, ca.flags_host, ca.init & 1,
                          (ca.init & 2) ? ca.partE + 2 * ELEM_NB_MAX : nullptr);
#else
        loop_control_body(ca.ctl, ca.partE, ca.nbE, ca.partR, ca.alpha_dev, ca.flags_host, ca.init);
#endif
        return;
    }
    if (ca.ctl->status != 0) return;
#ifdef DOTMI_PAIR_TU
    // (a paired slot -- alpha_dev[1] > 0, written by the element pass of this slot -- waits as well: its gather worked on the half
    // step, which only counts if the controller finds the full step's energy too high)
    if (epoch < (1 << 30) && (ca.ctl->holdNext || ((ca.init & 2) && ca.alpha_dev[1] > 0.0))) {
#else
    if (epoch < (1 << 30) && ca.ctl->holdNext) {
#endif
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

__device__ const double g_zero_slot = 0.0;

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

void launch_gemv(const DevParts &P, const double *q, hipStream_t st, const DevLoop *ctl, hipEvent_t ev0, hipEvent_t ev1,
                 const CtlArgs *ca, int spec)
{
    if (P.ntiles == 0 && P.nquad == 0 && P.nltiles == 0) return;
    if (q) {   // right-hand sides not in padded order yet
        const int total = P.nParts * P.nmax;
        int nb = (total + 255) / 256;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(gather_pad_kernel, dim3(nb), dim3(256), 0, st, total, P.dofmap, q, P.rpad);
    }
    // optional events time the streaming kernel alone (the roofline entry of bench.py is about that kernel): they are
    // attached to the dispatch itself (hipExtLaunchKernelGGL: the packet's own begin / end time stamps, what rocprofv3
    // reports as the kernel's duration) -- two hipEventRecord calls around the launch add ~5 us of barrier packets
    // wide tiles (rows of 2561..4096 columns) on the 512-thread kernel, the rest on the 256-thread one (two workgroups per
    // CU instead of one); the events (if any) span both launches: start of the first, stop of the last
    // P.tile = [wide tiles | narrow tiles of more than 256 columns | packs of four small tiles]: job k < nN of the narrow launch
    // is one tile, job nN + k the four tiles P.tile[ntiles + 4 k ..] (one wavefront each, backsolve_wave_tile)
    const int nW = P.ntilesWide, nN = P.ntiles - P.ntilesWide, nG = nN + P.nquad;
    const bool timed = ev0 && ev1;
    if (ca && spec <= 0) spec = 1;   // (callers pass the slot's epoch: > 0)
    if (ca && nG == 0)   // no launch of the 256-thread kernel to host it: the controller on its own, in front
        launch_loop_control(ca->ctl, ca->partE, ca->nbE, ca->partR, ca->alpha_dev, ca->flags_host, st, ca->init PAIR_INIT_MASK);
    if (nW > 0) {
        if (timed)
            hipExtLaunchKernelGGL((backsolve_kernel<512>), dim3(nW), dim3(512), 0, st, ev0, nG > 0 ? (hipEvent_t) nullptr : ev1, 0,
                                  P.tile, P.dofmap, P.W, P.nmax, P.rt, (const double *)P.rpad, P.ppart, P.nbmax, ctl, spec, nW);
        else
            hipLaunchKernelGGL((backsolve_kernel<512>), dim3(nW), dim3(512), 0, st, P.tile, P.dofmap, P.W, P.nmax, P.rt, P.rpad,
                               P.ppart, P.nbmax, ctl, spec, nW);
    }
    if (nG > 0 && ca) {
        // one workgroup more: the controller (backsolve_ctl_kernel)
        if (timed)
            hipExtLaunchKernelGGL(backsolve_ctl_kernel, dim3(nG + 1), dim3(256), 0, st, nW > 0 ? (hipEvent_t) nullptr : ev0, ev1, 0,
                                  P.tile + nW, P.dofmap, P.W, P.nmax, P.rt, (const double *)P.rpad, P.ppart, P.nbmax, *ca, spec, nN);
        else
            hipLaunchKernelGGL(backsolve_ctl_kernel, dim3(nG + 1), dim3(256), 0, st, P.tile + nW, P.dofmap, P.W, P.nmax, P.rt,
                               (const double *)P.rpad, P.ppart, P.nbmax, *ca, spec, nN);
    } else if (nG > 0) {
        if (timed)
            hipExtLaunchKernelGGL((backsolve_kernel<256>), dim3(nG), dim3(256), 0, st, nW > 0 ? (hipEvent_t) nullptr : ev0, ev1, 0,
                                  P.tile + nW, P.dofmap, P.W, P.nmax, P.rt, (const double *)P.rpad, P.ppart, P.nbmax, ctl, spec, nN);
        else
            hipLaunchKernelGGL((backsolve_kernel<256>), dim3(nG), dim3(256), 0, st, P.tile + nW, P.dofmap, P.W, P.nmax, P.rt,
                               P.rpad, P.ppart, P.nbmax, ctl, spec, nN);
    }
    if (P.nltiles > 0) {
        hipLaunchKernelGGL((backsolve_long_kernel<0>), dim3(P.nlwork), dim3(BSL_THREADS), 0, st, P.ltile, P.lwork, P.dofmap,
                           P.W, P.nmax, P.rt, P.rpad, P.tdots, P.maxChunks, P.ppart, P.nbmax, ctl, spec);
        hipLaunchKernelGGL((backsolve_long_kernel<1>), dim3(P.nlwork), dim3(BSL_THREADS), 0, st, P.ltile, P.lwork, P.dofmap,
                           P.W, P.nmax, P.rt, P.rpad, P.tdots, P.maxChunks, P.ppart, P.nbmax, ctl, spec);
    }
    if (!P.mt_ptr) launch_reduce_partial(P, st, ctl);   // (merge_tiles_kernel sums the tile partials itself)
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

// ------------------------------------------------------------------------------------------------
// inverse-Cholesky of a diagonal tile, base case: for one NB x NB diagonal block per wavefront compute
// L = chol(A_kk), X = L^-1 and store Q_kk = X^T (the inverse of the upper factor R_kk = L^T).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int srclane)
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

// One wavefront, everything in VGPRs: on entry lane i holds row i of the symmetric block, a[j] = A(i,j);
// on exit lane c holds column c of X = chol(A)^-1, x[i] = X(i,c) (zero for i < c).  Cross-lane operands
// are wave-uniform broadcasts (v_readlane), so there is no LDS traffic and no barrier in the O(NB^3)
// part; all loops are fully unrolled so the register arrays are statically indexed.
// Returns 0, or 1 + the index of the first non-positive pivot.
template <int NB>
__device__ __forceinline__ int wave_chol_inv(double (&a)[NB], double (&x)[NB], int lane)
{
    static_assert(NB <= 64, "one lane per row");
    int bad = 0;
    double mypiv = 1.0;  // lane k keeps pivot k, so the square roots / reciprocals are done in one go
    // right-looking Cholesky with deferred column scaling: A(i,j) -= A(i,k) A(j,k) / A(k,k)
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const double piv = readlane_f64(a[k], k);
        if (!(piv > 0.0) && bad == 0) bad = k + 1;
        if (lane == k) mypiv = piv;
        // 1/piv: hardware estimate + two Newton steps (full double accuracy, no IEEE-division fix-up code)
        double rp = __builtin_amdgcn_rcp(piv);
        rp = __builtin_fma(__builtin_fma(-piv, rp, 1.0), rp, rp);
        rp = __builtin_fma(__builtin_fma(-piv, rp, 1.0), rp, rp);
        const double t = a[k] * rp;
#pragma unroll
        for (int j = k + 1; j < NB; ++j) a[j] = __builtin_fma(-t, readlane_f64(a[k], j), a[j]);
    }
    // L(i,k) = A(i,k)/sqrt(A(k,k));  dinv = 1/L(lane,lane): one sqrt and one division per LANE
    const double dmine = sqrt(mypiv);
    const double dinv = 1.0 / dmine;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const double r = readlane_f64(dinv, k);
        a[k] = (lane == k) ? dmine : a[k] * r;
    }
    // X = L^-1 column by column: lane c solves L x = e_c;  L(i,k) is broadcast from lane i
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        double sacc = (lane == i) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) sacc = __builtin_fma(-readlane_f64(a[k], i), x[k], sacc);
        x[i] = (lane <= i) ? sacc * readlane_f64(dinv, i) : 0.0;
    }
    return bad;
}

typedef double mfma_v4d __attribute__((ext_vector_type(4)));

// M x M x M product (M = 16 or 32) on LDS operands with v_mfma_f64_16x16x4_f64, one 16 x 16 tile per wavefront
// (M = 16: wave 0 only): store(i, j, sum_k A(i,k) B(k,j)).  Operand / result layout as in mfma_gemm64 below.
template <int M, class FA, class FB, class FS>
__device__ __forceinline__ void mfma_gemm_small(FA A, FB B, FS store, int tid)
{
    static_assert(M == 16 || M == 32, "one tile per wave");
    const int lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int ti = (M == 32) ? (w >> 1) : 0, tj = (M == 32) ? (w & 1) : 0;
    const bool active = (M == 32 && w < 4) || w == 0;   // workgroups of more than four waves: the others only keep the barriers
    mfma_v4d acc = (mfma_v4d){0.0, 0.0, 0.0, 0.0};
    if (active) {
#pragma unroll
        for (int kk = 0; kk < M / 4; ++kk) {
            const double a = A(16 * ti + lr, 4 * kk + lk);
            const double b = B(4 * kk + lk, 16 * tj + lr);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        }
    }
    __syncthreads();
    if (active) {
#pragma unroll
        for (int r = 0; r < 4; ++r) store(16 * ti + lk + 4 * r, 16 * tj + lr, acc[r]);
    }
    __syncthreads();
}

constexpr int LD64 = CHOL_NB + 1;
#ifndef DOTMI_LANE_Q
#define DOTMI_LANE_Q 4   // side of the blocks a lane factors for itself (8: the 64 x 64 step 10.7 instead of 12.0 us, but ~100
                         // VGPRs for the lane's triangle -- the 512-thread tile kernel 168 instead of 118 -> one workgroup per CU)
#endif
// ---- 16 x 16 base case without cross-lane traffic -------------------------------------------------------------------------------
// wave_chol_inv<16> above keeps one row per lane and pays two v_readlane per multiply-add: 2.9 us per block, four of them in a
// row on the factorisation's dependent chain (tools/bench_diag.hip: 11.8 of the 15.3 us of block_chol_inv<64>).  Here the block
// is split further, down to Q x Q blocks (Q = 4) that EVERY lane factors for itself: the Q (Q + 1) / 2 entries of the lower
// triangle sit in the lane's registers (broadcast LDS reads), the Cholesky is straight-line code with static indices, and
// lane c then solves L x = e_c for column c of the inverse.  The products between the halves of the 2 x 2 recursion (4^3, 8^3)
// take one result entry per lane (R12 rests in the X12 block, which is zero in the end).  One wavefront, LDS as the only
// exchange, wave-level fences, no workgroup barrier: 2.0 us per 16 x 16 block.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// sqrt(d) and 1/sqrt(d) to double accuracy from the hardware estimate: two coupled Newton (Goldschmidt) steps and a residual
// correction of the root
__device__ __forceinline__ void sqrt_rsqrt(double d, double &root, double &rroot)
{
    const double r0 = __builtin_amdgcn_rsq(d);
    double g = d * r0, h = 0.5 * r0;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    g = __builtin_fma(__builtin_fma(-g, g, d), h, g);
    root = g;
    rroot = h + h;
}
// the Q x Q block at [o, o+Q): X = chol(G)^-1 into X (zero above the diagonal); returns 0 or 1 + the first bad pivot
template <int Q>
__device__ __forceinline__ int lane_chol_inv(double (*G)[LD64], double (*X)[LD64], int o, int lane)
{
    double a[Q * (Q + 1) / 2];   // a[i (i+1)/2 + j] = G(i, j), j <= i
#pragma unroll
    for (int i = 0; i < Q; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[i * (i + 1) / 2 + j] = G[o + i][o + j];
    int bad = 0;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        const double piv = a[k * (k + 1) / 2 + k];
        if (!(piv > 0.0) && bad == 0) bad = k + 1;
        double root, rr;
        sqrt_rsqrt(piv, root, rr);
        a[k * (k + 1) / 2 + k] = rr;   // (the diagonal slot keeps 1 / L_kk: the root itself is not needed again)
#pragma unroll
        for (int i = k + 1; i < Q; ++i) a[i * (i + 1) / 2 + k] *= rr;
#pragma unroll
        for (int j = k + 1; j < Q; ++j)
#pragma unroll
            for (int i = j; i < Q; ++i)
                a[i * (i + 1) / 2 + j] = __builtin_fma(-a[i * (i + 1) / 2 + k], a[j * (j + 1) / 2 + k], a[i * (i + 1) / 2 + j]);
    }
    // lane c solves L x = e_c (every lane runs the same straight-line code; x_i = 0 for i < c comes out by itself)
    // (column by column: x_k is final after one multiplication, the updates of the rows below it are independent of each other)
    const int c = lane & (Q - 1);
    double x[Q];
#pragma unroll
    for (int i = 0; i < Q; ++i) x[i] = (i == c) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < Q; ++k) {
        x[k] *= a[k * (k + 1) / 2 + k];
#pragma unroll
        for (int i = k + 1; i < Q; ++i) x[i] = __builtin_fma(-a[i * (i + 1) / 2 + k], x[k], x[i]);
    }
    if (lane < Q) {
#pragma unroll
        for (int i = 0; i < Q; ++i) X[o + i][o + lane] = (i >= lane) ? x[i] : 0.0;
    }
    return bad;
}
// the N x N block (N = Q, 2 Q, ... <= 16) at [b0, b0+N) by ONE wavefront: the 2 x 2 recursion of block_chol_inv down to Q x Q
// blocks that every lane factors for itself; the products between the halves take one result entry per lane
template <int N, int Q>
__device__ __forceinline__ int wave_chol_inv_lds(double (*G)[LD64], double (*X)[LD64], int b0, int lane)
{
    if constexpr (N == Q) {
        return lane_chol_inv<Q>(G, X, b0, lane);
    } else {
        constexpr int H = N / 2;
        const int i = (lane / H) % H, j = lane % H;
        const bool on = lane < H * H;
        auto T = [&](int r, int c) -> double & { return X[b0 + r][b0 + H + c]; };   // R12 rests in the X12 block (zero in the end)
        int bad = wave_chol_inv_lds<H, Q>(G, X, b0, lane);
        wave_lds_sync();
        if (on) {   // R12(i,j) = sum_k X11(i,k) A12(k,j)
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < H; ++k) s = __builtin_fma(X[b0 + i][b0 + k], G[b0 + k][b0 + H + j], s);
            T(i, j) = s;
        }
        wave_lds_sync();
        if (on) {   // A22(c,d) -= sum_k R12(k,c) R12(k,d)
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < H; ++k) s = __builtin_fma(T(k, i), T(k, j), s);
            G[b0 + H + i][b0 + H + j] -= s;
        }
        wave_lds_sync();
        const int b2 = wave_chol_inv_lds<H, Q>(G, X, b0 + H, lane);
        if (bad == 0 && b2) bad = H + b2;
        wave_lds_sync();
        if (on) {   // V(c,j) = sum_k R12(k,c) X11(k,j)  -> the A11 area (free by now)
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < H; ++k) s = __builtin_fma(T(k, i), X[b0 + k][b0 + j], s);
            G[b0 + i][b0 + j] = s;
        }
        wave_lds_sync();
        if (on) {   // X21(i,j) = -sum_c X22(i,c) V(c,j) ;  X12 = 0
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < H; ++k) s = __builtin_fma(X[b0 + H + i][b0 + H + k], G[b0 + k][b0 + j], s);
            X[b0 + H + i][b0 + j] = -s;
            X[b0 + i][b0 + H + j] = 0.0;
        }
        wave_lds_sync();
        return bad;
    }
}
__device__ __forceinline__ int wave_chol_inv16_lds(double (*G)[LD64], double (*X)[LD64], int b0, int lane)
{
    return wave_chol_inv_lds<16, DOTMI_LANE_Q>(G, X, b0, lane);
}

// X = chol(A)^-1 of the N x N diagonal block at [b0, b0+N) of a 64 x 64 matrix held in LDS, by one workgroup
// of 256 threads by a 2 x 2 recursion inside LDS -- the 16 x 16 bottom
// steps run in the registers of wave 0 (wave_chol_inv), the products on the FP64 matrix cores (mfma_gemm_small).
//   G: A on entry (row-major, symmetric), destroyed.   X: X(i,k) on exit, zero above the diagonal.
//   T32 / T16: 32x33 and 16x17 scratch.   Returns 0 or 1 + index (relative to b0) of the first non-positive
//   pivot (valid in wave 0).
// FAST: the 16 x 16 bottom steps by wave_chol_inv16_lds (blocks factored per lane: 2.0 instead of 2.9 us, the whole 64 x 64
// step 12.0 instead of 15.3 us, same register budget) -- the 512-thread tile kernels' form (DOTMI_FAST_DIAG=0 and the
// 256-thread form keep the one-row-per-lane base)
template <int N, bool FAST = false>
__device__ __forceinline__ int block_chol_inv(double (*G)[LD64], double (*X)[LD64], int b0, double (*T32)[33],
                                              double (*T16)[17], int tid)
{
    if constexpr (N == 16) {
        int bad = 0;
        if ((tid >> 6) == 0) {
            if constexpr (FAST) {
                bad = wave_chol_inv16_lds(G, X, b0, tid & 63);
            } else {
                const int lane = tid & 63;
                double a[16], x[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) a[j] = G[b0 + (lane & 15)][b0 + j];
                bad = wave_chol_inv<16>(a, x, lane);
                if (lane < 16) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) X[b0 + i][b0 + lane] = x[i];
                }
            }
        }
        __syncthreads();
        return bad;
    } else {
        constexpr int H = N / 2;
        auto T = [&](int i, int j) -> double & {
            if constexpr (H == 32) return T32[i][j];
            else return T16[i][j];
        };
        int bad = block_chol_inv<H, FAST>(G, X, b0, T32, T16, tid);
        // R12(i,j) = sum_k X11(i,k) A12(k,j)
        mfma_gemm_small<H>([&](int i, int k) { return X[b0 + i][b0 + k]; }, [&](int k, int j) { return G[b0 + k][b0 + H + j]; },
                    [&](int i, int j, double v) { T(i, j) = v; }, tid);
        // A22(c,d) -= sum_k R12(k,c) R12(k,d)
        mfma_gemm_small<H>([&](int c, int k) { return T(k, c); }, [&](int k, int d) { return T(k, d); },
                    [&](int c, int d, double v) { G[b0 + H + c][b0 + H + d] -= v; }, tid);
        const int b2 = block_chol_inv<H, FAST>(G, X, b0 + H, T32, T16, tid);
        if (bad == 0 && b2) bad = H + b2;
        // V(c,j) = sum_k R12(k,c) X11(k,j)  -> the A11 area (free by now)
        mfma_gemm_small<H>([&](int c, int k) { return T(k, c); }, [&](int k, int j) { return X[b0 + k][b0 + j]; },
                    [&](int c, int j, double v) { G[b0 + c][b0 + j] = v; }, tid);
        // X21(i,j) = -sum_c X22(i,c) V(c,j) ;  X12 = 0
        mfma_gemm_small<H>([&](int i, int c) { return X[b0 + H + i][b0 + H + c]; }, [&](int c, int j) { return G[b0 + c][b0 + j]; },
                    [&](int i, int j, double v) {
                        X[b0 + H + i][b0 + j] = -v;
                        X[b0 + i][b0 + H + j] = 0.0;
                    },
                    tid);
        return bad;
    }
}

// ------------------------------------------------------------------------------------------------
// Tile-level inverse-Cholesky (tile_factor.hpp): one workgroup = one tile task, one launch = one level of the
// static schedule.  64 x 64 tiles, products on v_mfma_f64_16x16x4_f64 from LDS, the next
// product's two tiles in flight (global -> registers) while the current one is multiplied; the accumulator tile
// stays in registers over the whole product list.  Product forms: TF_FACT  C -= A^T B ;  TF_INV  C += A B.
// ------------------------------------------------------------------------------------------------
// THREADS = 256: wave w owns rows [16 w, 16 w + 16) and all four 16-column tiles;  THREADS = 512: wave w owns rows
// [16 (w & 3), ...) and the two column tiles 2 (w >> 2), 2 (w >> 2) + 1 -- twice the waves per workgroup on the same
// LDS, so one wave's LDS reads and tile loads overlap the other's matrix-core time.
template <int THREADS>
struct TileGeom {
    static constexpr int NT = THREADS == 256 ? 4 : 2;   // 16-column tiles per wave
    static constexpr int NL = 2048 / THREADS;           // 16-byte pieces of a 64 x 64 tile per thread
};
template <int THREADS>
struct BlkT {
    double2 v[TileGeom<THREADS>::NL];
};
template <int THREADS>
__device__ __forceinline__ BlkT<THREADS> tile_load(const double *__restrict__ src, size_t ld, int tid)
{
    BlkT<THREADS> b;
#pragma unroll
    for (int u = 0; u < TileGeom<THREADS>::NL; ++u) {
        const int idx2 = tid + THREADS * u;
        b.v[u] = *reinterpret_cast<const double2 *>(src + (size_t)(idx2 >> 5) * ld + 2 * (idx2 & 31));
    }
    return b;
}
template <int THREADS>
__device__ __forceinline__ void tile_to_lds(const BlkT<THREADS> &b, double (*G)[CHOL_NB + 1], int tid)
{
#pragma unroll
    for (int u = 0; u < TileGeom<THREADS>::NL; ++u) {
        const int idx2 = tid + THREADS * u, j = idx2 >> 5, k = 2 * (idx2 & 31);
        G[k][j] = b.v[u].x;
        G[k + 1][j] = b.v[u].y;
    }
}
// LDS tile G[k][j] (element (row k, column j)) -> column-major global tile, 16-byte stores along the columns
template <int THREADS>
__device__ __forceinline__ void tile_store(double (*G)[CHOL_NB + 1], double *__restrict__ dst, size_t ld, int tid)
{
#pragma unroll
    for (int u = 0; u < TileGeom<THREADS>::NL; ++u) {
        const int idx2 = tid + THREADS * u, j = idx2 >> 5, k = 2 * (idx2 & 31);
        *reinterpret_cast<double2 *>(dst + (size_t)j * ld + k) = make_double2(G[k][j], G[k + 1][j]);
    }
}
// The same store WRITE-THROUGH (sc1) for the dataflow kernel: the tile leaves the XCD's L2 as it is written, so publishing
// it needs no release fence (a buffer_wbl2 behind 32 KB of fresh lines is ~6 us), only the storing waves' vmcnt drain in
// front of the flag.  16-byte raw buffer stores through a descriptor on the tile's (wave-uniform) origin, aux 16 = sc1.
typedef unsigned int tile_v4u __attribute__((ext_vector_type(4)));
template <int THREADS, bool TRANSPOSED>
__device__ __forceinline__ void tile_store_wt(double (*G)[CHOL_NB + 1], double *__restrict__ dst, int ld, int tid)
{
    const unsigned long long a = reinterpret_cast<unsigned long long>(dst);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const int ldu = __builtin_amdgcn_readfirstlane(ld);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0, (63 * ldu + 64) * 8, 0x00020000);
#pragma unroll
    for (int u = 0; u < TileGeom<THREADS>::NL; ++u) {
        const int idx2 = tid + THREADS * u, j = idx2 >> 5, k = 2 * (idx2 & 31);
        union {
            double d[2];
            tile_v4u v;
        } w;
        w.d[0] = TRANSPOSED ? G[j][k] : G[k][j];
        w.d[1] = TRANSPOSED ? G[j][k + 1] : G[k + 1][j];
        __builtin_amdgcn_raw_buffer_store_b128(w.v, rsrc, (j * ldu + k) * 8, 0, 16);
    }
}
template <int THREADS, bool TRANS_A>
__device__ __forceinline__ void mfma_acc_tile(mfma_v4d (&acc)[TileGeom<THREADS>::NT], double (*La)[CHOL_NB + 1],
                                              double (*Lb)[CHOL_NB + 1], double sign, int tid)
{
    constexpr int NT = TileGeom<THREADS>::NT;
    const int lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
    const int rb = 16 * (w & 3), cb = 16 * NT * (w >> 2);
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
        const double a = sign * (TRANS_A ? La[4 * kk + lk][rb + lr] : La[rb + lr][4 * kk + lk]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const double b = Lb[4 * kk + lk][cb + 16 * t + lr];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
    }
}

#ifdef DIAG_PROFILE
__device__ long long g_diag_prof[4096][8];
__device__ int g_diag_prof_n;
extern "C" int dotmi_debug_diag_prof(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_prof), sizeof(long long) * 8 * (size_t)n);
}
#define DPROF(k) do { if (slot >= 0 && threadIdx.x == 0) g_diag_prof[slot][k] = wall_clock64(); } while (0)
#else
#define DPROF(k) do { } while (0)
#endif
// one tile task on the workgroup's LDS tiles (the body of both the level kernel and the dataflow kernel below)
template <int THREADS, bool COH = false, bool FAST = false>
__device__ __forceinline__ void tile_task_body(const TileTask &t, const TileProd *__restrict__ prods, int *__restrict__ info,
                                               double (*La)[CHOL_NB + 1], double (*Lb)[CHOL_NB + 1], double (*T32)[33],
                                               double (*T16)[17])
{
    constexpr int NB = CHOL_NB, LD = NB + 1, NT = TileGeom<THREADS>::NT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4;
    auto TL = [](const double *src, int ld, int tid_) { return tile_load<THREADS>(src, (size_t)ld, tid_); };
#ifdef DIAG_PROFILE
    __shared__ int s_slot;
    if (threadIdx.x == 0) s_slot = (t.post == TP_DIAG || t.post == TP_ROW) ? atomicAdd(&g_diag_prof_n, 1) : -1;
    __syncthreads();
    const int slot = (s_slot >= 0 && s_slot < 4096) ? s_slot : -1;
    if (slot >= 0 && threadIdx.x == 0) { g_diag_prof[slot][6] = t.post; g_diag_prof[slot][7] = t.nprod; }
    DPROF(0);
#endif
    const int rb = 16 * (w & 3), cb = 16 * NT * (w >> 2);   // this wave's rows / first column of the accumulator tiles
    const bool fact = t.form == TF_FACT;   // C -= A^T B on H tiles;  else C += A B
    const TileProd *pl = prods + t.first;
    mfma_v4d acc[NT];
#pragma unroll
    for (int q = 0; q < NT; ++q) acc[q] = (mfma_v4d){0.0, 0.0, 0.0, 0.0};
    BlkT<THREADS> ra, rbk;
    if (t.nprod > 0) {
        ra = TL(t.p0.a, t.p0.lda, tid);
        if (t.p0.b != t.p0.a) rbk = TL(t.p0.b, t.p0.ldb, tid);
    } else if (t.post == TP_ROW) {
        ra = TL(t.q, t.ldq, tid);
    } else if (t.post == TP_RMUL) {
        rbk = TL(t.q, t.ldq, tid);
    }
    if (t.init) {
        tile_to_lds<THREADS>(TL(t.c, t.ldc, tid), La, tid);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][r] = La[rb + lk + 4 * r][cb + 16 * q + lr];
        __syncthreads();
    }
    DPROF(1);
    for (int p = 0; p < t.nprod; ++p) {
        const bool same = pl[p].b == pl[p].a;
        tile_to_lds<THREADS>(ra, La, tid);
        if (!same) tile_to_lds<THREADS>(rbk, Lb, tid);
        __syncthreads();
        if (p + 1 < t.nprod) {
            ra = TL(pl[p + 1].a, pl[p + 1].lda, tid);
            if (pl[p + 1].b != pl[p + 1].a) rbk = TL(pl[p + 1].b, pl[p + 1].ldb, tid);
        } else if (t.post == TP_ROW) {
            ra = TL(t.q, t.ldq, tid);   // Q_kk for the final multiplication
        } else if (t.post == TP_RMUL) {
            rbk = TL(t.q, t.ldq, tid);  // Q_jj for the final multiplication
        }
        if (fact) mfma_acc_tile<THREADS, true>(acc, La, same ? La : Lb, -1.0, tid);
        else mfma_acc_tile<THREADS, false>(acc, La, Lb, 1.0, tid);
        __syncthreads();
    }
    if (t.post == TP_STORE || t.post == TP_NEG) {
        const double sg = t.post == TP_NEG ? -1.0 : 1.0;
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) La[rb + lk + 4 * r][cb + 16 * q + lr] = sg * acc[q][r];
        __syncthreads();
        if constexpr (COH) tile_store_wt<THREADS, false>(La, t.o, t.ldc, tid);
        else tile_store<THREADS>(La, t.o, t.ldc, tid);
        return;
    }
    if (t.post == TP_RMUL) {
        // Q_ij = -T Q_jj:  the sum T (accumulators) becomes the A operand in LDS, element (i, k) at La[i][k]; the stored
        // Q_jj tile (upper triangular) is the B operand, element (k, j) at Lb[k][j]
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) La[rb + lk + 4 * r][cb + 16 * q + lr] = acc[q][r];
        tile_to_lds<THREADS>(rbk, Lb, tid);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[q] = (mfma_v4d){0.0, 0.0, 0.0, 0.0};
        mfma_acc_tile<THREADS, false>(acc, La, Lb, -1.0, tid);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) La[rb + lk + 4 * r][cb + 16 * q + lr] = acc[q][r];
        __syncthreads();
        if constexpr (COH) tile_store_wt<THREADS, false>(La, t.o, t.ldc, tid);
        else tile_store<THREADS>(La, t.o, t.ldc, tid);
        return;
    }
    DPROF(2);
    // G = updated H tile -> LDS
    double (*G)[LD] = t.post == TP_ROW ? Lb : La;
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) G[rb + lk + 4 * r][cb + 16 * q + lr] = acc[q][r];
    if (t.post == TP_ROW) {
        // R_kj = Q_kk^T G:  R(i,j) = sum_k X(i,k) G(k,j),  X(i,k) = element (k,i) of the stored Q_kk tile
        tile_to_lds<THREADS>(ra, La, tid);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[q] = (mfma_v4d){0.0, 0.0, 0.0, 0.0};
        mfma_acc_tile<THREADS, true>(acc, La, Lb, 1.0, tid);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) Lb[rb + lk + 4 * r][cb + 16 * q + lr] = acc[q][r];
        __syncthreads();
        DPROF(3);
        if constexpr (COH) tile_store_wt<THREADS, false>(Lb, t.o, t.ldc, tid);
        else tile_store<THREADS>(Lb, t.o, t.ldc, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DPROF(4);
        return;
    }
    __syncthreads();
    const int bad = block_chol_inv<64, FAST>(La, Lb, 0, T32, T16, tid);
    DPROF(3);
    // Q_jj = X^T: column i of the stored tile, row k <- X(i,k) (zero for k > i: the strictly lower part is cleared)
    if constexpr (COH) {
        tile_store_wt<THREADS, true>(Lb, t.o, t.ldc, tid);
    } else {
        for (int idx = tid; idx < NB * NB; idx += THREADS) {
            const int i = idx / NB, k = idx % NB;
            t.o[(size_t)i * t.ldc + k] = Lb[i][k];
        }
    }
    if (tid == 0 && bad) atomicMax(info + t.sub, t.pivotBase + bad);
#ifdef DIAG_PROFILE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DPROF(4);
#endif
}

template <int THREADS, bool FAST = false>
__global__ __launch_bounds__(THREADS, 2) void tile_task_kernel(const TileTask *__restrict__ tasks,
                                                               const TileProd *__restrict__ prods, int *__restrict__ info)
{
    constexpr int NB = CHOL_NB, LD = NB + 1;
    __shared__ double La[NB][LD], Lb[NB][LD], T32[32][33], T16[16][17];
    const TileTask t = tasks[blockIdx.x];
    tile_task_body<THREADS, false, FAST>(t, prods, info, La, Lb, T32, T16);
}

// Dataflow form of the same factorisation (DOTMI_TILE_FLOW): ONE launch of persistent workgroups that
// pull the tasks of the whole schedule, in its (topological) order, from a counter and wait -- per task -- only for the
// tasks whose tiles it touches (build_tile_deps, tile_factor.hpp), so the levels overlap: a workgroup that has finished a
// task of level l goes on with the next unissued task whatever the other workgroups of level l are doing, and a diagonal
// task starts the moment its own row tiles are there.  done[v] == epoch <=> task v of this factorisation has finished
// (the epoch grows by one per factorisation, so nothing is cleared); next[epoch & 1] is the ticket counter, the other one
// is reset for the next launch by whoever draws ticket 0.  Tickets are drawn in order and a task only waits for tasks with
// smaller tickets, all of which are held by workgroups that are running: no deadlock whatever the grid size.  Sums keep
// their fixed order (a tile is still written by one task at a time): results equal to the level kernel's bit for bit.
// A wait that exceeds ~2 s (never, unless a kernel before it failed) flags the subdomain and goes on, so the launch ends.
// (The second launch bound is WAVES PER SIMD, not workgroups per CU: with 2 the compiler takes 211-256 VGPRs here and ONE
// persistent 512-thread workgroup is resident per CU.  Round 5 tried 4 -- two per CU, <= 128 VGPRs: 80 spilled registers with
// the body inlined; with the body as a noinline call 120 VGPRs and no spill, bar17K 1.128 -> 1.110 ms, monkey 0.793 -> 0.740,
// but bunny5K 0.360 -> 0.457 (the chain of dependent tasks pays the call) and the FAST = false form failed its parity test:
// not kept.)
template <int THREADS, bool FAST = false>
__global__ __launch_bounds__(THREADS, 2) void tile_flow_kernel(const TileTask *__restrict__ tasks,
                                                               const TileProd *__restrict__ prods, int ntasks,
                                                               const int *__restrict__ depPtr, const int *__restrict__ depIdx,
                                                               int *__restrict__ done, int *__restrict__ next, int epoch,
                                                               long long waitTicks, int *__restrict__ info)
{
    // the 256-thread form (256 VGPRs + 84 bytes of scratch) gave a wrong factor on horse7K -- the same non-SPD pivot in every
    // run -- and was not pursued: it cannot be instantiated
    static_assert(THREADS == 512, "tile_flow_kernel: only the 512-thread form is validated");
    constexpr int NB = CHOL_NB, LD = NB + 1;
    __shared__ double La[NB][LD], Lb[NB][LD], T32[32][33], T16[16][17];
    __shared__ int s_ticket;
    const int tid = threadIdx.x;
    int *const ctr = next + (epoch & 1);
    // (the ticket for the next task is drawn by the thread that publishes the finished one, in ONE divergent region that a
    // barrier follows: two regions `if (tid == 0)` on either side of the loop's back edge get threaded into one path by the
    // compiler, after which the other lanes of wave 0 spin through the loop without lane 0 -- seen in the ISA, hangs)
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        __syncthreads();
        const int ti = s_ticket;
        if (ti >= ntasks) return;
        const TileTask t = tasks[ti];
        const int d0 = depPtr[ti], d1 = depPtr[ti + 1];
        if (tid < 64) {   // ONE wave polls (one flag per lane), relaxed, with a sleep between the looks
            for (int d = d0 + tid; d < d1; d += 64) {
                const int *flag = done + depIdx[d];
                const long long tStart = wall_clock64();
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - tStart > waitTicks) {   // 100 MHz counter
                        atomicMax(info + t.sub, 1 << 30);
                        break;
                    }
                }
            }
            // ONE acquire after the last flag has been seen (the barrier hands it on).  (sc1 tile loads and no fence
            // measured 3 % faster on bunny5K; the fence is the form the guide's hand-off recipe validates, kept.)
            if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        tile_task_body<THREADS, true, FAST>(t, prods, info, La, Lb, T32, T16);   // (result tile stored write-through)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave: its part of the result tile has left
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(done + ti, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ti == 0) __hip_atomic_store(next + ((epoch + 1) & 1), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ticket = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The product / row / inverse tasks (everything but TP_DIAG) on HALF tiles: LDS holds 32 x 64 of A and of B at a time
// (33 KB instead of the 77 KB of the task kernel above), registers the accumulator and one prefetched half pair, so
// four workgroups are resident per CU instead of two -- a launch of ~600-1700 short tasks runs in half the rounds, and
// the diagonal-block tasks of the same level run next to it from their own launch (launch_tile_level).
//   TF_FACT  acc -= sum_k A(k, i) B(k, j):  K = the tiles' rows;  TF_INV  acc += sum_k A(i, k) B(k, j):  K = A's columns.
// In both cases the A half is stored K-major, La[k][i], so one inner loop serves both.
__global__ __launch_bounds__(256, 4) void tile_gemm_kernel(const TileTask *__restrict__ tasks,
                                                           const TileProd *__restrict__ prods)
{
    constexpr int KH = 32, LD = CHOL_NB + 1;
    __shared__ double Ls[2 * KH * LD];                     // La | Lb, or one whole 64 x 65 tile (4160 doubles either way)
    double (*La)[LD] = reinterpret_cast<double (*)[LD]>(Ls);
    double (*Lb)[LD] = reinterpret_cast<double (*)[LD]>(Ls + KH * LD);
    double (*Lf)[LD] = reinterpret_cast<double (*)[LD]>(Ls);
    const TileTask t = tasks[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lk = lane >> 4, rb = 16 * w;
    const bool fact = t.form == TF_FACT;
    const TileProd *pl = prods + t.first;
    const int nsteps = 2 * t.nprod;
    // half tiles: `rows` = rows [32 h, 32 h + 32) of all 64 columns (K = rows), `cols` = columns [32 h, ...) (K = columns)
    auto load_rows = [&](const double *p, int ld, int h, double2 (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx2 = tid + 256 * u;
            v[u] = *reinterpret_cast<const double2 *>(p + (size_t)(idx2 >> 4) * ld + 32 * h + 2 * (idx2 & 15));
        }
    };
    auto store_rows = [&](const double2 (&v)[4], double (*Lx)[LD]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx2 = tid + 256 * u, j = idx2 >> 4, k = 2 * (idx2 & 15);
            Lx[k][j] = v[u].x;
            Lx[k + 1][j] = v[u].y;
        }
    };
    auto load_cols = [&](const double *p, int ld, int h, double2 (&v)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx2 = tid + 256 * u;
            v[u] = *reinterpret_cast<const double2 *>(p + (size_t)(32 * h + (idx2 >> 5)) * ld + 2 * (idx2 & 31));
        }
    };
    auto store_cols = [&](const double2 (&v)[4], double (*Lx)[LD]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx2 = tid + 256 * u, c = idx2 >> 5, r = 2 * (idx2 & 31);
            Lx[c][r] = v[u].x;
            Lx[c][r + 1] = v[u].y;
        }
    };
    double2 ra[4], rbk[4];
    auto fetch = [&](int s) {   // half step s of the product list
        const TileProd pr = s < 2 ? t.p0 : pl[s >> 1];
        if (fact) load_rows(pr.a, pr.lda, s & 1, ra);
        else load_cols(pr.a, pr.lda, s & 1, ra);
        load_rows(pr.b, pr.ldb, s & 1, rbk);
    };
    auto mfma_half = [&](mfma_v4d (&acc)[4], double sign) {
#pragma unroll
        for (int kk = 0; kk < KH / 4; ++kk) {
            const double a = sign * La[4 * kk + lk][rb + lr];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double b = Lb[4 * kk + lk][16 * q + lr];
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
            }
        }
    };
    mfma_v4d acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = (mfma_v4d){0.0, 0.0, 0.0, 0.0};
    if (nsteps > 0) fetch(0);
    else if (t.post == TP_ROW) load_rows(t.q, t.ldq, 0, ra);
    else if (t.post == TP_RMUL) load_rows(t.q, t.ldq, 0, rbk);
    if (t.init) {
        // the c tile through LDS into the accumulator layout (element (i, j) of acc[q][r]: i = rb + lk + 4 r, j = 16 q + lr)
        double2 c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx2 = tid + 256 * u;
            c[u] = *reinterpret_cast<const double2 *>(t.c + (size_t)(idx2 >> 5) * t.ldc + 2 * (idx2 & 31));
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx2 = tid + 256 * u, j = idx2 >> 5, r = 2 * (idx2 & 31);
            Lf[r][j] = c[u].x;
            Lf[r + 1][j] = c[u].y;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][r] = Lf[rb + lk + 4 * r][16 * q + lr];
        __syncthreads();
    }
    for (int s = 0; s < nsteps; ++s) {
        if (fact) store_rows(ra, La);
        else store_cols(ra, La);
        store_rows(rbk, Lb);
        __syncthreads();
        if (s + 1 < nsteps) fetch(s + 1);
        else if (t.post == TP_ROW) load_rows(t.q, t.ldq, 0, ra);   // first half of Q_kk for the final multiplication
        else if (t.post == TP_RMUL) load_rows(t.q, t.ldq, 0, rbk);  // first K half (rows 0..31) of Q_jj
        mfma_half(acc, fact ? -1.0 : 1.0);
        __syncthreads();
    }
    if (t.post == TP_RMUL) {
        // Q_ij = -T Q_jj:  C(i, j) = -sum_k T(i, k) Q(k, j), K in two halves: the T half (columns 32 h .. of the accumulators, which
        // every wave holds for its 16 rows) K-major into La[k][i], the Q half (rows 32 h .., all 64 columns) from HBM into Lb[k][j]
        mfma_v4d acc2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc2[q] = (mfma_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            store_rows(rbk, Lb);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) La[16 * q + lr][rb + lk + 4 * r] = acc[2 * h + q][r];
            __syncthreads();
            if (h == 0) load_rows(t.q, t.ldq, 1, rbk);
            mfma_half(acc2, -1.0);
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = acc2[q];
    }
    if (t.post == TP_ROW) {
        // R_kj = Q_kk^T G:  R(i, j) = sum_k Q(k, i) G(k, j), K in two halves: the Q half from HBM, the G half from the
        // accumulators of the two waves that own those rows
        mfma_v4d acc2[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc2[q] = (mfma_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            store_rows(ra, La);
            if ((w >> 1) == h) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Lb[rb - 32 * h + lk + 4 * r][16 * q + lr] = acc[q][r];
            }
            __syncthreads();
            if (h == 0) load_rows(t.q, t.ldq, 1, ra);
            mfma_half(acc2, 1.0);
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = acc2[q];
    }
    const double sg = t.post == TP_NEG ? -1.0 : 1.0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) Lf[rb + lk + 4 * r][16 * q + lr] = sg * acc[q][r];
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int idx2 = tid + 256 * u, j = idx2 >> 5, r = 2 * (idx2 & 31);
        *reinterpret_cast<double2 *>(t.o + (size_t)j * t.ldc + r) = make_double2(Lf[r][j], Lf[r + 1][j]);
    }
}

void launch_tile_level(const TileTask *tasks, int ntasks, const TileProd *prods, int *info, hipStream_t st, bool fastDiag)
{
    if (ntasks <= 0) return;
    if (fastDiag) hipLaunchKernelGGL((tile_task_kernel<512, true>), dim3(ntasks), dim3(512), 0, st, tasks, prods, info);
    else hipLaunchKernelGGL((tile_task_kernel<512, false>), dim3(ntasks), dim3(512), 0, st, tasks, prods, info);
}
void launch_tile_flow(const TileTask *tasks, int ntasks, const TileProd *prods, const int *depPtr, const int *depIdx, int *done,
                      int *next, int epoch, int *info, int nwg, hipStream_t st, double waitMs, bool fastDiag)
{
    if (ntasks <= 0) return;
    const int grid = std::min(ntasks, nwg);
    const long long waitTicks = (long long)(waitMs * 1e5);
    if (fastDiag)
        hipLaunchKernelGGL((tile_flow_kernel<512, true>), dim3(grid), dim3(512), 0, st, tasks, prods, ntasks, depPtr, depIdx, done,
                           next, epoch, waitTicks, info);
    else
        hipLaunchKernelGGL((tile_flow_kernel<512, false>), dim3(grid), dim3(512), 0, st, tasks, prods, ntasks, depPtr, depIdx, done,
                           next, epoch, waitTicks, info);
}
void launch_tile_gemm(const TileTask *tasks, int ntasks, const TileProd *prods, hipStream_t st)
{
    if (ntasks > 0) hipLaunchKernelGGL(tile_gemm_kernel, dim3(ntasks), dim3(256), 0, st, tasks, prods);
}

// zero a list of 64 x 64 tiles (the tiles a factorisation leaves non-zero, before the refill)
__global__ __launch_bounds__(256) void clear_tiles_kernel(double *const *__restrict__ tiles, const int *__restrict__ lds_)
{
    double *tp = tiles[blockIdx.x];
    const int lda = lds_[blockIdx.x];
    for (int idx = threadIdx.x; idx < 64 * 32; idx += 256) {
        const int c = idx >> 5, r2 = idx & 31;
        *reinterpret_cast<double2 *>(tp + (size_t)c * lda + 2 * r2) = make_double2(0.0, 0.0);
    }
}
void launch_clear_tiles(double *const *tiles, const int *lds_, int ntiles, hipStream_t st)
{
    if (ntiles > 0) hipLaunchKernelGGL(clear_tiles_kernel, dim3(ntiles), dim3(256), 0, st, tiles, lds_);
}

// z_v = (sum over parts containing v of p_s[local v]) / dup_v ; partial dots c_i = y_i . z
template <bool DEV>
__global__ __launch_bounds__(256) void merge_kernel(int nV, const int *__restrict__ vp_ptr,
                                                    const int *__restrict__ vp_off,
                                                    const int *__restrict__ dup,
                                                    const double *__restrict__ psub, LbfgsArgs L,
                                                    int with_dots, int divide, double *__restrict__ z,
                                                    double *__restrict__ partials,
                                                    const DevLoop *__restrict__ ctl, VList vl)
{
    __shared__ double sm[4 * RED_K];
    if constexpr (DEV) {
        if (ctl->status != 0 || ctl->phase != 0) return;
    }
    const LbfgsArgs &Lr = [&]() -> const LbfgsArgs & {
        if constexpr (DEV) return ctl->L;
        else return L;
    }();
    double acc[RED_K];
#pragma unroll
    for (int j = 0; j < RED_K; ++j) acc[j] = 0.0;
    const int stride = gridDim.x * blockDim.x;
    const int nvis = vl.v ? vl.n : nV;
    for (int jv = blockIdx.x * blockDim.x + threadIdx.x; jv < nvis; jv += stride) {
        const int v = vl_vtx(vl, jv);
        double z0 = 0, z1 = 0, z2 = 0;
        const int k0 = vp_ptr[v], k1 = vp_ptr[v + 1];
        // everything that does not depend on the slot list is requested before it is walked
        const int d = divide ? dup[v] : 1;
        double yv[HIST_MAX][3];
        if (with_dots) {
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i)
                if (i < Lr.m) {
                    const double *yi = Lr.y[i] + 3 * v;
                    yv[i][0] = yi[0];
                    yv[i][1] = yi[1];
                    yv[i][2] = yi[2];
                }
        }
        // slots four at a time (a vertex is in 1-3 subdomains, rarely more): offsets first, then the values
        for (int k = k0; k < k1; k += 4) {
            int off[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) off[u] = (k + u < k1) ? vp_off[k + u] : -1;
            double w[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (off[u] >= 0) {
                    const double *ps = psub + off[u];
                    w[u][0] = ps[0];
                    w[u][1] = ps[1];
                    w[u][2] = ps[2];
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (off[u] >= 0) {
                    z0 += w[u][0];
                    z1 += w[u][1];
                    z2 += w[u][2];
                }
        }
        if (d > 1) {
            z0 /= d;
            z1 /= d;
            z2 /= d;
        }
        z[3 * v] = z0;
        z[3 * v + 1] = z1;
        z[3 * v + 2] = z2;
        if (with_dots) {
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i)
                if (i < Lr.m) acc[i] += yv[i][0] * z0 + yv[i][1] * z1 + yv[i][2] * z2;
        }
    }
    // device-loop mode always stores all HIST_MAX columns: the consumer's m is only known on the device
    if (with_dots) write_partials(acc, DEV ? HIST_MAX : Lr.m, partials, sm);
}

// The partial-sum reduce and the merge in one launch: thread = global scalar dof; its value is the sum over the
// subdomains holding the vertex of (the sum over that subdomain's tiles holding the column of ppart[tile][column]) --
// the additions of reduce_partial_p_kernel followed by merge_kernel, in their order, without the psub round trip and
// without a launch in between.
constexpr int MT_CH = 24;   // list entries in flight per thread (a dof has ~15: two subdomains x ~8 tiles)
template <bool DEV>
__global__ __launch_bounds__(256) void merge_tiles_kernel(int n3, const int *__restrict__ mt_ptr,
                                                          const int *__restrict__ mt_ent, const int *__restrict__ dup,
                                                          const double *__restrict__ ppart, LbfgsArgs L, int with_dots,
                                                          int divide, double *__restrict__ z,
                                                          double *__restrict__ partials, const DevLoop *__restrict__ ctl, VList vl)
{
    __shared__ double sm[4 * RED_K];
    if constexpr (DEV) {
        if (ctl->status != 0 || ctl->phase != 0) return;
    }
    const LbfgsArgs &Lr = [&]() -> const LbfgsArgs & {
        if constexpr (DEV) return ctl->L;
        else return L;
    }();
    double acc[RED_K];
#pragma unroll
    for (int j = 0; j < RED_K; ++j) acc[j] = 0.0;
    const int stride = gridDim.x * blockDim.x;
    const int cnt = vl_count3(vl, n3);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        const int k = vl_dof(vl, i);
        const int e0 = mt_ptr[k], e1 = mt_ptr[k + 1];
        const int d = divide ? dup[k / 3] : 1;
        double yk[HIST_MAX];
        if (with_dots) {
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i)
                if (i < Lr.m) yk[i] = Lr.y[i][k];
        }
        double zk = 0.0, ps = 0.0;
        // MT_CH entries at a time: offsets first, then the values, then the adds in list order
        for (int e = e0; e < e1; e += MT_CH) {
            int off[MT_CH];
#pragma unroll
            for (int u = 0; u < MT_CH; ++u) off[u] = (e + u < e1) ? mt_ent[e + u] : 0;
            double w[MT_CH];
#pragma unroll
            for (int u = 0; u < MT_CH; ++u) {
                const int o = off[u] < 0 ? ~off[u] : off[u];
                w[u] = (e + u < e1) ? ppart[o] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < MT_CH; ++u)
                if (e + u < e1) {
                    if (off[u] < 0 && e + u > e0) {   // a new subdomain starts: close the previous one
                        zk += ps;
                        ps = 0.0;
                    }
                    ps += w[u];
                }
        }
        zk += ps;
        if (d > 1) zk /= d;
        z[k] = zk;
        if (with_dots) {
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i)
                if (i < Lr.m) acc[i] += yk[i] * zk;
        }
    }
    if (with_dots) write_partials(acc, DEV ? HIST_MAX : Lr.m, partials, sm);
}

// Early back-solve (enqueue_loop_slot): the tiles hold the partials of u = -M g for the gradient of the iterate the
// controller has just accepted.  M is fixed during a step and linear, so with the M y_i of the stored pairs kept beside
// the y_i the preconditioned vector of the two-loop is  z = M (-g - sum_j xi_j y_j) = u - sum_j xi_j (M y_j), and the
// newest pair's M y = M (g - g_old) = u_old - u costs no solve of its own.  Same sums per dof as merge_tiles_kernel
// (tiles of a subdomain, then subdomains, then the division by the multiplicity), then the history terms newest first
// like build_qpad's.  first: start of the step (no pair yet; u_old is only set).
__global__ __launch_bounds__(256) void merge_tiles_early_kernel(int n3, const int *__restrict__ mt_ptr,
                                                                const int *__restrict__ mt_ent, const int *__restrict__ dup,
                                                                const double *__restrict__ ppart, int first,
                                                                const double *__restrict__ zsum,
                                                                const int *__restrict__ vp_ptr, const int *__restrict__ vp_off,
                                                                const double *__restrict__ psub,
                                                                const uint8_t *__restrict__ ownMask, VList vl,
                                                                const uint8_t *__restrict__ kind, int pre,
                                                                double *__restrict__ zshare,
                                                                double *__restrict__ z, double *__restrict__ partials,
                                                                const DevLoop *__restrict__ ctl)
{
    __shared__ double sm[4 * RED_K];
    if (ctl->status != 0 || ctl->phase != 0) return;
    const LbfgsArgs &Lr = ctl->L;
    const int m = first ? 0 : Lr.m;
    const bool pairNew = !first && ctl->pairNew != 0 && m > 0;
    double *__restrict__ u_old = ctl->u_old;
    const double *my[HIST_MAX];
    double xi[HIST_MAX];
    double *my_new = nullptr;
#pragma unroll
    for (int i = 0; i < HIST_MAX; ++i) {
        my[i] = (i < m) ? ctl->MY[ctl->order[i]] : nullptr;
        xi[i] = (i < m) ? ctl->X.xi[i] : 0.0;
        if (pairNew && i == m - 1) my_new = ctl->MY[ctl->order[i]];
    }
    double acc[RED_K];
#pragma unroll
    for (int j = 0; j < RED_K; ++j) acc[j] = 0.0;
    const int stride = gridDim.x * blockDim.x;
    const int cnt = vl_count3(vl, n3);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        const int k = vl_dof(vl, i);
        const int vtx = k / 3;
        int e0 = 0, e1 = 0, c0 = 0, c1 = 0;
        if (psub) {
            c0 = vp_ptr[vtx];
            c1 = vp_ptr[vtx + 1];
        } else if (!zsum) {
            e0 = mt_ptr[k];
            e1 = mt_ptr[k + 1];
        }
        const int d = dup[vtx];
        double yk[HIST_MAX], mk[HIST_MAX];
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i) {
            yk[i] = (i < m) ? Lr.y[i][k] : 0.0;
            mk[i] = (i < m && !(pairNew && i == m - 1)) ? my[i][k] : 0.0;
        }
        const double uo = first ? 0.0 : u_old[k];
        // zsum (sharded subdomains): the all-reduced sum over every rank's subdomains (merge_tiles_kernel without the division
        // into a staging buffer, then the collective -- on the staging buffer, so that a slot whose merge is gated off leaves z
        // alone); only the division and the history terms are left
        double u = zsum ? zsum[k] : 0.0, ps = 0.0;
        if (psub && !zsum) {
            // split form (big meshes): the subdomains' own sums are in psub (reduce_partial_p_kernel); same additions in the
            // same order as the list walk below -- tiles of a subdomain first, then the subdomains
            const int dd = k - 3 * vtx;
            for (int c = c0; c < c1; c += 4) {
                double w[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = (c + q < c1) ? psub[vp_off[c + q] + dd] : 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c + q < c1) u += w[q];
            }
        }
        for (int e = e0; e < e1; e += MT_CH) {
            int off[MT_CH];
#pragma unroll
            for (int q = 0; q < MT_CH; ++q) off[q] = (e + q < e1) ? mt_ent[e + q] : 0;
            double w[MT_CH];
#pragma unroll
            for (int q = 0; q < MT_CH; ++q) {
                const int o = off[q] < 0 ? ~off[q] : off[q];
                w[q] = (e + q < e1) ? ppart[o] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < MT_CH; ++q)
                if (e + q < e1) {
                    if (off[q] < 0 && e + q > e0) {   // a new subdomain starts: close the previous one
                        u += ps;
                        ps = 0.0;
                    }
                    ps += w[q];
                }
        }
        u += ps;
        if (pre && (kind[vtx] & 2)) {
            // Owner exchange with the y_i . z in the packet, BEFORE it travels, at a vertex other ranks hold too: u is this
            // rank's subdomains' PART of the sum -- it goes to the buffer the packet is filled from.  With U = (sum over the
            // ranks)/multiplicity the lines below form  z = (1 + xi_new) U - xi_new u_old - sum_{j stored before} xi_j M y_j,
            // so  y_i . z = (1 + xi_new) sum_ranks y_i . (part/multiplicity) - [owner] y_i . (xi_new u_old + sum_j xi_j M y_j):
            // this rank's share goes to the partials, the vertex' z / u_old / M y_new are formed after the exchange (a second
            // launch over the shared vertices, partials == nullptr)
            zshare[k] = u;
            const double xin = pairNew ? xi[m - 1] : 0.0;
            double t = (1.0 + xin) * (d > 1 ? u / d : u);
            if (kind[vtx] & 1) {
                double r = pairNew ? xin * uo : 0.0;
#pragma unroll
                for (int j = HIST_MAX - 1; j >= 0; --j)
                    if (j < m && !(pairNew && j == m - 1)) r += xi[j] * mk[j];
                t -= r;
            }
#pragma unroll
            for (int j = 0; j < HIST_MAX; ++j)
                if (j < m) acc[j] += yk[j] * t;
            continue;
        }
        if (d > 1) u /= d;
        u_old[k] = u;
        const double myn = uo - u;   // M y of the pair the controller has just stored
        if (pairNew) my_new[k] = myn;
        double zk = u;
#pragma unroll
        for (int j = HIST_MAX - 1; j >= 0; --j)
            if (j < m) zk -= xi[j] * ((pairNew && j == m - 1) ? myn : mk[j]);
        z[k] = zk;
        if (partials && (!ownMask || ownMask[vtx])) {
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i)
                if (i < m) acc[i] += yk[i] * zk;
        }
    }
    if (partials) write_partials(acc, HIST_MAX, partials, sm);   // (nullptr: the y_i . z came with the packet, yz_pre_kernel)
}

// ---- owner exchange (DOTMI_FLAG_OWNER_EXCHANGE): only the entries of vertices held by more than one rank travel ----------------
// red0 / red1: up to two partial arrays whose rows workgroup 0 sums into the packet's tail on the way (the energy's two columns
// combined, the statistics' columns) -- the same single-wave sums as dotmi_collectives.hip's reduce_rows_kernel, without launches of their own
__device__ __forceinline__ void pack_reduce_rows(const PackRed &r, double *__restrict__ pack)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (!r.part) return;
    if (r.combine) {
        if (w != 0) return;
        double a0 = 0.0, a1 = 0.0;
        for (int b = lane; b < r.rows; b += 64) {
            a0 += r.part[(size_t)b * r.stride];
            a1 += r.part[(size_t)b * r.stride + 1];
        }
        for (int o = 32; o > 0; o >>= 1) {
            a0 += __shfl_down(a0, o, 64);
            a1 += __shfl_down(a1, o, 64);
        }
        if (lane == 0) pack[r.dst] = r.s0 * a0 + r.s1 * a1;
        return;
    }
    for (int j = w; j < r.cols; j += 4) {
        double acc = 0.0;
        for (int b = lane; b < r.rows; b += 64) acc += r.part[(size_t)b * r.stride + j];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) pack[r.dst + j] = acc;
    }
}
__global__ __launch_bounds__(256) void pack_iface_kernel(int nI, const int *__restrict__ idx, const double *__restrict__ src,
                                                         double *__restrict__ pack, const double *__restrict__ tailp, int ntail,
                                                         PackRed red0, PackRed red1)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 3 * nI) pack[t] = src[3 * idx[t / 3] + t % 3];
    else if (t < 3 * nI + ntail) pack[t] = tailp[t - 3 * nI];
    if (blockIdx.x == 0) {
        pack_reduce_rows(red0, pack);
        pack_reduce_rows(red1, pack);
    }
}
// (only the vertices THIS rank holds take the sum: a vertex shared by two other ranks stays zero here)
// tail2 (gradient's packet): dst2[0 .. ntail2) = the summed tail behind the first one, dst2[0] += the squares of the packet's
// summed vector entries -- |g|^2 over the shared vertices, the same bits on every rank (workgroup 0)
__global__ __launch_bounds__(256) void unpack_iface_kernel(int nI, const int *__restrict__ idx, const double *__restrict__ pack,
                                                           const uint8_t *__restrict__ heldMask, double *__restrict__ dst,
                                                           double *__restrict__ tailp, int ntail, double *__restrict__ dst2,
                                                           int ntail2)
{
    __shared__ double sm[4];
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < 3 * nI) {
        const int v = idx[t / 3];
        if (heldMask[v]) dst[3 * v + t % 3] = pack[t];
    } else if (t < 3 * nI + ntail) {
        tailp[t - 3 * nI] = pack[t];
    }
    if (blockIdx.x == 0 && ntail2 > 0) {
        double a = 0.0;
        for (int q = threadIdx.x; q < 3 * nI; q += 256) a += pack[q] * pack[q];
        for (int o = 32; o > 0; o >>= 1) a += __shfl_down(a, o, 64);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x < ntail2)
            dst2[threadIdx.x] =
                pack[3 * nI + ntail + threadIdx.x] + (threadIdx.x == 0 ? (sm[0] + sm[1]) + (sm[2] + sm[3]) : 0.0);
    }
}
void launch_pack_iface(int nI, const int *idx, const double *src, double *pack, const double *tailp, int ntail, hipStream_t st,
                       const PackRed *red0, const PackRed *red1)
{
    const int tot = 3 * nI + ntail;
    const PackRed none{nullptr, 0, 0, 0, 0, 0, 0.0, 0.0};
    if (tot > 0 || red0 || red1)
        hipLaunchKernelGGL(pack_iface_kernel, dim3(std::max(1, (tot + 255) / 256)), dim3(256), 0, st, nI, idx, src, pack, tailp,
                           ntail, red0 ? *red0 : none, red1 ? *red1 : none);
}
void launch_unpack_iface(int nI, const int *idx, const double *pack, const uint8_t *heldMask, double *dst, double *tailp, int ntail,
                         hipStream_t st, double *dst2, int ntail2)
{
    const int tot = 3 * nI + ntail;
    if (tot > 0 || ntail2 > 0)
        hipLaunchKernelGGL(unpack_iface_kernel, dim3(std::max(1, (tot + 255) / 256)), dim3(256), 0, st, nI, idx, pack, heldMask, dst,
                           tailp, ntail, dst2, ntail2);
}
__global__ __launch_bounds__(256) void masked_norm2_kernel(int n, const double *__restrict__ v, const uint8_t *__restrict__ ownMask,
                                                           int exact, double *__restrict__ partials)
{
    __shared__ double sm[4 * RED_K];
    double acc[RED_K];
#pragma unroll
    for (int j = 0; j < RED_K; ++j) acc[j] = 0.0;
    const int stride = gridDim.x * blockDim.x;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride)
        if (exact ? ownMask[k / 3] == exact : ownMask[k / 3] != 0) acc[0] += v[k] * v[k];
    write_partials(acc, 1, partials, sm);
}
void launch_masked_norm2(int n, const double *v, const uint8_t *ownMask, double *partials, hipStream_t st, int exact)
{
    hipLaunchKernelGGL(masked_norm2_kernel, dim3(NB_RED), dim3(256), 0, st, n, v, ownMask, exact, partials);
}
__global__ __launch_bounds__(256) void mask_owned_kernel(int n, double *__restrict__ v, const uint8_t *__restrict__ ownMask)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n && !ownMask[k / 3]) v[k] = 0.0;
}
void launch_mask_owned(int n, double *v, const uint8_t *ownMask, hipStream_t st)
{
    hipLaunchKernelGGL(mask_owned_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, v, ownMask);
}

void launch_merge_early(const DevMesh &M, const DevParts &P, double *z, double *partials, int first, hipStream_t st,
                        const DevLoop *ctl, const double *zsum, const uint8_t *ownMask, VList vl, const uint8_t *kind, int pre,
                        double *zshare)
{
    const bool split = !P.mt_ptr;
    hipLaunchKernelGGL(merge_tiles_early_kernel, dim3(NB_RED), dim3(256), 0, st, 3 * M.nV, P.mt_ptr, P.mt_ent, P.dup, P.ppart,
                       first, zsum, split ? P.vp_ptr : nullptr, split ? P.vp_off : nullptr,
                       split ? (const double *)P.psub : nullptr, ownMask, vl, kind, pre, zshare, z, partials, ctl);
}

void launch_merge(const DevMesh &M, const DevParts &P, const LbfgsArgs &L, double *z, double *partials,
                  int with_dots, hipStream_t st, const DevLoop *ctl, VList vl)
{
    if (P.mt_ptr) {   // (launch_gemv left the tile partials in ppart and skipped the reduce)
        if (ctl)
            hipLaunchKernelGGL(merge_tiles_kernel<true>, dim3(NB_RED), dim3(256), 0, st, 3 * M.nV, P.mt_ptr, P.mt_ent, P.dup,
                               P.ppart, L, with_dots & 1, (with_dots >> 1) & 1, z, partials, ctl, vl);
        else
            hipLaunchKernelGGL(merge_tiles_kernel<false>, dim3(NB_RED), dim3(256), 0, st, 3 * M.nV, P.mt_ptr, P.mt_ent, P.dup,
                               P.ppart, L, with_dots & 1, (with_dots >> 1) & 1, z, partials, ctl, vl);
        return;
    }
    // with_dots: bit0 = accumulate y_i.z partials, bit1 = divide by dup
    if (ctl)
        hipLaunchKernelGGL(merge_kernel<true>, dim3(NB_RED), dim3(256), 0, st, M.nV, P.vp_ptr, P.vp_off, P.dup, P.psub,
                           L, with_dots & 1, (with_dots >> 1) & 1, z, partials, ctl, vl);
    else
        hipLaunchKernelGGL(merge_kernel<false>, dim3(NB_RED), dim3(256), 0, st, M.nV, P.vp_ptr, P.vp_off, P.dup, P.psub,
                           L, with_dots & 1, (with_dots >> 1) & 1, z, partials, ctl, vl);
}

// Sharded subdomains: z holds the all-reduced SUM over every subdomain; z_v /= dup_v and the partial dots c_i = y_i . z,
// i.e. the second half of merge_kernel after the collective (same vertex loop and partial layout).
template <bool DEV>
__global__ __launch_bounds__(256) void zfinish_kernel(int nV, const int *__restrict__ dup, LbfgsArgs L,
                                                      double *__restrict__ z, double *__restrict__ partials,
                                                      const DevLoop *__restrict__ ctl)
{
    __shared__ double sm[4 * RED_K];
    if constexpr (DEV) {
        if (ctl->status != 0 || ctl->phase != 0) return;
    }
    const LbfgsArgs &Lr = [&]() -> const LbfgsArgs & {
        if constexpr (DEV) return ctl->L;
        else return L;
    }();
    double acc[RED_K];
#pragma unroll
    for (int j = 0; j < RED_K; ++j) acc[j] = 0.0;
    const int stride = gridDim.x * blockDim.x;
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < nV; v += stride) {
        const int d = dup[v];
        double z0 = z[3 * v], z1 = z[3 * v + 1], z2 = z[3 * v + 2];
        if (d > 1) {
            z0 /= d;
            z1 /= d;
            z2 /= d;
            z[3 * v] = z0;
            z[3 * v + 1] = z1;
            z[3 * v + 2] = z2;
        }
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i)
            if (i < Lr.m) {
                const double *yi = Lr.y[i] + 3 * v;
                acc[i] += yi[0] * z0 + yi[1] * z1 + yi[2] * z2;
            }
    }
    write_partials(acc, DEV ? HIST_MAX : Lr.m, partials, sm);
}

void launch_zfinish(int nV, const int *dup, const LbfgsArgs &L, double *z, double *partials, hipStream_t st,
                    const DevLoop *ctl)
{
    if (ctl) hipLaunchKernelGGL(zfinish_kernel<true>, dim3(NB_RED), dim3(256), 0, st, nV, dup, L, z, partials, ctl);
    else hipLaunchKernelGGL(zfinish_kernel<false>, dim3(NB_RED), dim3(256), 0, st, nV, dup, L, z, partials, ctl);
}

// ------------------------------------------------------------------------------------------------
// alpha_0 = clamp(-p.g / p.Hp, alphaMin, 1): block-CSR SpMV fused with the two dot products
// ------------------------------------------------------------------------------------------------
constexpr int SPMV_R = 3;   // block rows a lane group works on at a time
__global__ __launch_bounds__(256) void spmv_dots_kernel(int v0, int v1, const int *__restrict__ adj_ptr,
                                                        const int *__restrict__ adj_idx,
                                                        const double *__restrict__ Hval,
                                                        const double *__restrict__ p,
                                                        const double *__restrict__ g,
                                                        double *__restrict__ Hp,
                                                        double *__restrict__ partials,
                                                        const DevLoop *__restrict__ ctl)
{
    __shared__ double sm[8];
    if (ctl) {
        if (ctl->status != 0 || ctl->phase != 0) return;
        g = ctl->g_cur;
    }
    double pg = 0, pHp = 0;
    const int sub = threadIdx.x & 7;
    const int ngroups = gridDim.x * 32;
    // SPMV_R block rows per lane group and trip, their column loops interleaved: a trip is a chain of three dependent
    // memory round trips (row range -> column indices -> entries of p, ~1 us each) whatever the number of rows in it, so
    // a 17 k-vertex mesh takes one trip instead of three.  The sums keep a fixed order (row by row, as before).
    constexpr int R = SPMV_R;
    for (int vbase = v0 + blockIdx.x * 32 + (threadIdx.x >> 3); vbase < v1; vbase += R * ngroups) {
        double a[R][3], q[R][3], gg[R][3];
        int kb[R], nk[R], nkmax = 0;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int v = vbase + u * ngroups;
            kb[u] = nk[u] = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) a[u][d] = q[u][d] = gg[u][d] = 0.0;
            if (v < v1) {
                kb[u] = adj_ptr[v];
                nk[u] = adj_ptr[v + 1] - kb[u];
                // the row's own p and g do not depend on the column loop: request them first
                if (sub == 0) {
                    q[u][0] = p[3 * v]; q[u][1] = p[3 * v + 1]; q[u][2] = p[3 * v + 2];
                    if (g) { gg[u][0] = g[3 * v]; gg[u][1] = g[3 * v + 1]; gg[u][2] = g[3 * v + 2]; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) nkmax = max(nkmax, nk[u]);
        for (int t = sub; t < nkmax; t += 8) {
            int col[R];
            double h[R][9], pc[R][3];
#pragma unroll
            for (int u = 0; u < R; ++u) col[u] = (t < nk[u]) ? adj_idx[kb[u] + t] : -1;
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (t < nk[u]) {
                    const double *b = Hval + (size_t)9 * (kb[u] + t);
#pragma unroll
                    for (int i = 0; i < 9; ++i) h[u][i] = b[i];
                }
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (col[u] >= 0) {
                    const double *pu = p + 3 * col[u];
                    pc[u][0] = pu[0]; pc[u][1] = pu[1]; pc[u][2] = pu[2];
                }
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (col[u] >= 0) {
                    a[u][0] += h[u][0] * pc[u][0] + h[u][1] * pc[u][1] + h[u][2] * pc[u][2];
                    a[u][1] += h[u][3] * pc[u][0] + h[u][4] * pc[u][1] + h[u][5] * pc[u][2];
                    a[u][2] += h[u][6] * pc[u][0] + h[u][7] * pc[u][1] + h[u][8] * pc[u][2];
                }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int v = vbase + u * ngroups;
            const double a0 = group8_sum(a[u][0]), a1 = group8_sum(a[u][1]), a2 = group8_sum(a[u][2]);
            if (sub == 0 && v < v1) {
                if (Hp) {
                    Hp[3 * v] = a0;
                    Hp[3 * v + 1] = a1;
                    Hp[3 * v + 2] = a2;
                }
                pHp += q[u][0] * a0 + q[u][1] * a1 + q[u][2] * a2;
                if (g) pg += q[u][0] * gg[u][0] + q[u][1] * gg[u][1] + q[u][2] * gg[u][2];
            }
        }
    }
    // both block sums through one exchange
    const double w0 = wave_sum(pg), w1 = wave_sum(pHp);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        sm[w] = w0;
        sm[4 + w] = w1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.x * RED_K] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
        partials[(size_t)blockIdx.x * RED_K + 1] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
    }
}

// Early order: build_p and spmv_dots in one launch.  The sparse product runs on z (ready when the launch starts); the
// history terms of the direction and of H p are local to a row: p_v = z_v + sum_j delta_j s_j[v] (summed over the row's
// lane group, pair j on lane j), (H p)_v = (H z)_v + sum_j delta_j (H s_j)[v] with the H s_j cached beside the s_j (H is fixed
// during a step; H s_new = alpha H p is written by the vertex gather).  delta comes from the y_i . z partials in wave 0's
// prologue (requested first, finished behind the column loop).  Device loop only.
__global__ __launch_bounds__(256) void spmv_zp_kernel(int nV, int v0, int v1, const uint8_t *__restrict__ rowMask,
                                                      const uint8_t *__restrict__ ownMask, const int *__restrict__ adj_ptr,
                                                      const int *__restrict__ adj_idx,
                                                      const double *__restrict__ Hval, const double *__restrict__ z,
                                                      const double *__restrict__ c_partials, int c_blocks,
                                                      double *__restrict__ p, double *__restrict__ Hp,
                                                      double *__restrict__ partials, const DevLoop *__restrict__ ctl, VList vl)
{
    __shared__ double sm[8];
    __shared__ double delta[HIST_MAX];
    if (ctl->status != 0 || ctl->phase != 0) return;
    const double *__restrict__ g = ctl->g_cur;
    const LbfgsArgs &Lr = ctl->L;
    const int m = Lr.m;
    // wave 0: the y_i . z partial columns are requested now ...
    double c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = 0.0;
    if (threadIdx.x < 64) {
        for (int b = threadIdx.x; b < c_blocks; b += 64) {
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i) c[i] += c_partials[(size_t)b * RED_K + i];
        }
    }
    bool haveDelta = false;
    const int sub = threadIdx.x & 7;
    double mydelta = 0.0;   // delta of the pair this lane carries (lane j of a row's group: pair j)
    auto finish_delta = [&]() {   // ... and reduced here (build_p_kernel's prologue); one barrier, all threads
        if (threadIdx.x < 64) {
            const double tot = wave_sum8_transposed(c, threadIdx.x);
            double ct[HIST_MAX], rys[HIST_MAX];
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i) {
                ct[i] = __shfl(tot, 8 * i, 64);
                rys[i] = (i < m) ? 1.0 / Lr.ys[i] : 0.0;
            }
            double d[HIST_MAX];
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i) {
                d[i] = 0.0;
                if (i < m) {
                    double yp = ct[i];
#pragma unroll
                    for (int j = 0; j < HIST_MAX; ++j)
                        if (j < i) yp += d[j] * Lr.sy[j][i];
                    d[i] = ctl->X.xi[i] - yp * rys[i];
                }
                if (threadIdx.x == 0) delta[i] = d[i];
            }
        }
        __syncthreads();
        mydelta = (sub < m) ? delta[sub < HIST_MAX ? sub : 0] : 0.0;
        haveDelta = true;
    };
    double pg = 0, pHp = 0;
    const int ngroups = gridDim.x * 32;
    constexpr int R = SPMV_R;
    static_assert(HIST_MAX <= 8, "one lane of a row's group per stored pair");
    const double *__restrict__ hist_s = nullptr, *__restrict__ hist_hs = nullptr;
#pragma unroll
    for (int j = 0; j < HIST_MAX; ++j)
        if (j == sub && j < m) {
            hist_s = Lr.s[j];
            hist_hs = ctl->HS[ctl->order[j]];
        }
    // (the trip count is the same for every thread of a workgroup: finish_delta's barrier sits inside the first trip)
    const int nrows = vl.v ? vl.n : nV;   // owner exchange: the rows of the held vertices only (p is zero elsewhere)
    for (int base = blockIdx.x * 32; base < nrows; base += R * ngroups) {
        const int vbase = base + (threadIdx.x >> 3);
        double a[R][3], zv[R][3], gg[R][3], sv[R][3], hv[R][3];   // sv / hv: pair number `sub` of the history (lanes 0 .. m-1)
        int kb[R], nk[R], vv[R], nkmax = 0;
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int jv = vbase + u * ngroups;
            const int v = jv < nrows ? vl_vtx(vl, jv) : nV;
            vv[u] = v;
            kb[u] = nk[u] = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) a[u][d] = zv[u][d] = gg[u][d] = sv[u][d] = hv[u][d] = 0.0;
            if (v < nV) {
                // sharded rows (N > 1 with the sharded element pass): p for every vertex, the product and the two dots only on
                // this rank's rows [v0, v1) -- the others' H p (and cached H s_j) are never read
                if (rowMask ? rowMask[v] != 0 : (v >= v0 && v < v1)) {
                    kb[u] = adj_ptr[v];
                    nk[u] = adj_ptr[v + 1] - kb[u];
                }
                // the row's own operands do not depend on the column loop: requested first.  Lane j of the row's group
                // takes pair j of the history (HIST_MAX <= 8 lanes)
                if (sub == 0) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        zv[u][d] = z[3 * v + d];
                        gg[u][d] = g[3 * v + d];
                    }
                }
                if (sub < m) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        sv[u][d] = hist_s[3 * v + d];
                        hv[u][d] = hist_hs[3 * v + d];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < R; ++u) nkmax = max(nkmax, nk[u]);
        for (int t = sub; t < nkmax; t += 8) {
            int col[R];
            double h[R][9], pc[R][3];
#pragma unroll
            for (int u = 0; u < R; ++u) col[u] = (t < nk[u]) ? adj_idx[kb[u] + t] : -1;
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (t < nk[u]) {
                    const double *b = Hval + (size_t)9 * (kb[u] + t);
#pragma unroll
                    for (int i = 0; i < 9; ++i) h[u][i] = b[i];
                }
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (col[u] >= 0) {
                    const double *pu = z + 3 * col[u];
                    pc[u][0] = pu[0]; pc[u][1] = pu[1]; pc[u][2] = pu[2];
                }
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (col[u] >= 0) {
                    a[u][0] += h[u][0] * pc[u][0] + h[u][1] * pc[u][1] + h[u][2] * pc[u][2];
                    a[u][1] += h[u][3] * pc[u][0] + h[u][4] * pc[u][1] + h[u][5] * pc[u][2];
                    a[u][2] += h[u][6] * pc[u][0] + h[u][7] * pc[u][1] + h[u][8] * pc[u][2];
                }
        }
        if (!haveDelta) finish_delta();
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int v = vv[u];
            // (H p)_v = sum over the group of [its columns' part of (H z)_v + delta_j (H s_j)_v];  p_v = z_v + sum_j delta_j s_j[v]
            double pv[3], hp[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                hp[d] = group8_sum(a[u][d] + hv[u][d] * mydelta);
                pv[d] = group8_sum(zv[u][d] + sv[u][d] * mydelta);
            }
            if (sub == 0 && v < nV) {
#pragma unroll
                for (int d = 0; d < 3; ++d) p[3 * v + d] = pv[d];
                if (rowMask ? rowMask[v] != 0 : (v >= v0 && v < v1)) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) Hp[3 * v + d] = hp[d];
                    pHp += pv[0] * hp[0] + pv[1] * hp[1] + pv[2] * hp[2];
                    if (!ownMask || ownMask[v]) pg += pv[0] * gg[u][0] + pv[1] * gg[u][1] + pv[2] * gg[u][2];
                }
            }
        }
    }
    if (!haveDelta) finish_delta();
    const double w0 = wave_sum(pg), w1 = wave_sum(pHp);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        sm[w] = w0;
        sm[4 + w] = w1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[(size_t)blockIdx.x * RED_K] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
        partials[(size_t)blockIdx.x * RED_K + 1] = (sm[4] + sm[5]) + (sm[6] + sm[7]);
    }
}

void launch_spmv_zp(const DevMesh &M, const double *Hval, const double *z, const double *c_partials, double *p, double *Hp,
                    double *partials, hipStream_t st, const DevLoop *ctl, int v0, int v1, const uint8_t *rowMask,
                    const uint8_t *ownMask, VList vl)
{
    if (v1 < 0) v1 = M.nV;
    hipLaunchKernelGGL(spmv_zp_kernel, dim3(NB_RED), dim3(256), 0, st, M.nV, v0, v1, rowMask, ownMask, M.adj_ptr, M.adj_idx, Hval,
                       z, c_partials, NB_RED, p, Hp, partials, ctl, vl);
}

void launch_spmv_dots(const DevMesh &M, const double *Hval, const double *p, const double *g, double *Hp,
                      int v0, int v1, double *partials, hipStream_t st, const DevLoop *ctl)
{
    hipLaunchKernelGGL(spmv_dots_kernel, dim3(NB_RED), dim3(256), 0, st, v0, v1, M.adj_ptr, M.adj_idx, Hval,
                       p, g, Hp, partials, ctl);
}

__global__ __launch_bounds__(256) void step_forward_kernel(int n, const double *__restrict__ x0,
                                                           const double *__restrict__ p,
                                                           double *__restrict__ x,
                                                           const double *__restrict__ spmv_partials,
                                                           double alpha_host, int use_partials,
                                                           double alpha_min, double *__restrict__ alpha_out,
                                                           double *__restrict__ alpha_out_host,
                                                           const DevLoop *__restrict__ ctl, VList vl)
{
    __shared__ double sh_alpha;
    if (ctl) {
        if (ctl->status != 0) return;
        x0 = ctl->x_cur;
        x = ctl->x_trial;
        use_partials = ctl->phase == 0;   // a retry steps with the halved alpha the controller left
        alpha_host = ctl->alpha;
    }
    if (threadIdx.x < 64) {
        double alpha = alpha_host;
        if (use_partials) {
            double pg = 0.0, pHp = 0.0;  // both columns in flight together
            for (int b = threadIdx.x; b < NB_RED; b += 64) {
                pg += spmv_partials[(size_t)b * RED_K];
                pHp += spmv_partials[(size_t)b * RED_K + 1];
            }
            pg = __shfl(wave_sum(pg), 0, 64);
            pHp = __shfl(wave_sum(pHp), 0, 64);
            alpha = fmax(alpha_min, fmin(1.0, -pg / pHp));  // Optimizer.cpp:1085
        }
        if (threadIdx.x == 0) {
            sh_alpha = alpha;
            if (blockIdx.x == 0) {
                *alpha_out = alpha;
                if (alpha_out_host) *alpha_out_host = alpha;  // pinned host copy: no D2H memcpy on the hot path
            }
        }
    }
    __syncthreads();
    const double alpha = sh_alpha;
    const int stride = gridDim.x * blockDim.x;
    const int cnt = vl_count3(vl, n);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += stride) {
        const int k = vl_dof(vl, i);
        x[k] = x0[k] + alpha * p[k];
    }
}

void launch_step_forward(int n, const double *x0, const double *p, double *x, const double *spmv_partials,
                         double alpha_host, int use_partials, double alpha_min, double *alpha_out,
                         double *alpha_out_host, hipStream_t st, const DevLoop *ctl, VList vl)
{
    int nb = ((vl.v ? 3 * vl.n : n) + 255) / 256;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(step_forward_kernel, dim3(nb), dim3(256), 0, st, n, x0, p, x, spmv_partials,
                       alpha_host, use_partials, alpha_min, alpha_out, alpha_out_host, ctl, vl);
}

// ------------------------------------------------------------------------------------------------
// loop controller: what the host loop of dotmi_step does between a trial and the next launch
// (line search Optimizer.cpp:806-833, history update DOTTimeStepper.cpp:474-494, stopping test
// Optimizer.cpp:317-330), on the device.  One wavefront; the partial sums are added in block order,
// the same order the host path uses, so both paths produce the same bits.
// ------------------------------------------------------------------------------------------------
// (a device function: the controller is a launch of its own, or workgroup 0 of the back-solve launch -- 256 threads)
__device__ __forceinline__ void loop_control_body(DevLoop *__restrict__ ctl, const double *__restrict__ partE, int nbE,
                                  const double *__restrict__ partR, const double *__restrict__ alpha_dev,
                                  int *__restrict__ flags_host, int init PAIR_ARG_DECL)
{
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
#ifdef DOTMI_PAIR_TU
    const double alpha_full = (partE2 && !init) ? alpha_dev[1] : 0.0;   // > 0: a paired slot (elem_patch_kernel)
    __shared__ double chunk2[2][SUM_CHUNKS];
#endif
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
        if (hasA) {
#pragma unroll
            for (int k = 0; k < LR; ++k) va[k] = partR[(size_t)(chA * LR + k) * RED_K + colA];
        }
        if (hasB) {
#pragma unroll
            for (int k = 0; k < LR; ++k) vb[k] = partR[(size_t)(chB * LR + k) * RED_K + colB];
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
#ifdef DOTMI_PAIR_TU
            if (alpha_full > 0.0) {   // the same chunks of the full step's partials (same order: the same bits as a plain slot)
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
                chunk2[c][ch] = e2;
            }
#endif
        }
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
#ifdef DOTMI_PAIR_TU
        if (C.slots < C.kindCap) C.slot_kind[C.slots] = C.phase == 0 ? 1 : 2;
        C.slots++;
        int kind = (C.phase == 0 || C.redo) ? 0 : 1;   // first trial of an iteration / retry after a halving
        // what a first trial with alpha_0 < 1 teaches the pairing rule (a redone trial has taught it already)
        const double a0 = alpha_full > 0.0 ? alpha_full : alpha_in;
        const bool learns = C.phase == 0 && a0 < 1.0;
        C.redo = 0;
        bool decided = false;
        if (alpha_full > 0.0) {
            // Paired slot: the energy of the FULL step first, as the reference's line search would see it
            double ea = 0.0, ei = 0.0;
#pragma unroll
            for (int c = 0; c < SUM_CHUNKS; ++c) {
                ea += chunk2[0][c];
                ei += chunk2[1][c];
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
        if (decided) {
        } else if (E > C.E_cur && alpha > 0.0) {
#else
        C.evals++;
        if (C.slots < C.kindCap) C.slot_kind[C.slots] = C.phase == 0 ? 1 : 2;
        C.slots++;
        const int kind = C.phase == 0 ? 0 : 1;   // first trial of an iteration / retry after a halving
        C.heldSlots += C.holdNext;
        if (E > C.E_cur && alpha > 0.0) {
#endif
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
#ifdef DOTMI_PAIR_TU
            C.heldRejected += (C.holdNext || alpha_full > 0.0) ? 1 : 0;
#else
            C.heldRejected += C.holdNext;
#endif
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
        }
    }
    // forecast for the slot that follows: hold its back-solve if its kind's counter for the current pattern says "rejected"
    if (t == 0 && !init) {
        const int nk = C.phase == 0 ? 0 : 1;
#ifdef DOTMI_PAIR_TU
        C.holdNext = (C.holdEnable && C.status == 0 && !C.redo && C.predCtr[nk][C.predHist[nk] & 3] >= 2) ? 1 : 0;
#else
        C.holdNext = (C.holdEnable && C.status == 0 && C.predCtr[nk][C.predHist[nk] & 3] >= 2) ? 1 : 0;
#endif
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
    loop_control_body(ctl, partE, nbE, partR, alpha_dev, flags_host, init);
}

void launch_loop_control(DevLoop *ctl, const double *partE, int nbE, const double *partR,
                         const double *alpha_dev, int *flags_host, hipStream_t st, int init)
{
    hipLaunchKernelGGL(loop_control_kernel, dim3(1), dim3(256), 0, st, ctl, partE, nbE, partR, alpha_dev, flags_host,
                       init);
}

// ------------------------------------------------------------------------------------------------
// element Hessians  H_e = (dF/dx)^T [w U^(A (+) B)U^^T]_PSD (dF/dx)   (Energy.cpp:738-777, :1129-1270, IglUtils.hpp:466-479)
// One wavefront = 64 tets.
//   phase 1 (lane = tet): F, SVD, projected spectral blocks A_w (3x3), B_w (three 2x2) -> LDS, together with U and,
//           for each of the 4 vertices, y_v = V^T c_v  (c_v = that vertex' row of dF/dx: c_0 = -(sum of the rest-inverse
//           rows), c_k = row k-1)
//   phase 2 (16 lanes per tet, lane = vertex pair (v, v')): the 3x3 block
//               H_vv' = U N_vv' U^T,    N_vv'[a][c] = sum_{b,d} Mh[(a,b),(c,d)] y_v[b] y_v'[d]
//           with Mh the 21 spectral entries (A_w on ((a,a),(c,c)); B_w of the pair (p,q) on ((p,q),(q,p)) x itself), i.e.
//               N[a][c]  = A_w[a][c] y_v[a] y_v'[c]                                   for all a, c
//               N[p][p] += b00 y_v[q] y_v'[q] ;  N[p][q] += b01 y_v[q] y_v'[p]
//               N[q][p] += b10 y_v[p] y_v'[q] ;  N[q][q] += b11 y_v[p] y_v'[p]       for (p,q) = (0,1),(1,2),(2,0)
//           For (2,0) the pair order (p,q),(q,p) is index 6 then 2: the reference's "transposed" fill M(6,6)=B(0,0),
//           M(6,2)=B(0,1), M(2,6)=B(1,0), M(2,2)=B(1,1)  (Energy.cpp:1203-1207).
//   Same contraction as expanding the 9x9 dP/dF and applying dF/dx twice, in ~2.4 kflop per tet instead of ~11 kflop
//   (round 1 expanded all 81 + 144 entries with 64 lanes per tet and was instruction-bound: 131 us on 86k tets).
// ------------------------------------------------------------------------------------------------
constexpr int EH_FIELDS = 9 + 12 + 9 + 12;  // U, y_0..y_3, Aw, Bw

template <int MAT>
__global__ __launch_bounds__(64) void elem_hessian_kernel(const int4 *__restrict__ T,
                                                          const double *__restrict__ A, int nTp, int nT,
                                                          const double *__restrict__ mu,
                                                          const double *__restrict__ lam,
                                                          const double *__restrict__ vol,
                                                          const double *__restrict__ x, double dtSq,
                                                          const int *__restrict__ elist, double *__restrict__ He)
{
    // tet-major, odd row length: the 16 lanes of a tet read different fields of one row (distinct banks)
    __shared__ double pack[64][EH_FIELDS + 1];
    const int lane = threadIdx.x;
    // elist: the elements this rank needs (sharded refresh); row i of He then belongs to element elist[i]
    const int ei = blockIdx.x * 64 + lane;
    const int e = (elist && ei < nT) ? elist[ei] : ei;
    if (ei < nT) {
        const int4 t = T[e];
        double xs[4][3];
        const int vid[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int d = 0; d < 3; ++d) xs[k][d] = x[3 * vid[k] + d];
        double Ai[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Ai[r][c] = A[(size_t)(3 * r + c) * nTp + e];
        Mat3 F;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double d0 = xs[1][r] - xs[0][r], d1 = xs[2][r] - xs[0][r], d2 = xs[3][r] - xs[0][r];
#pragma unroll
            for (int c = 0; c < 3; ++c) F.m[r][c] = d0 * Ai[0][c] + d1 * Ai[1][c] + d2 * Ai[2][c];
        }
        Mat3 U, V, Aw;
        double S[3], Bw[3][4];
        svd3(F, U, S, V);
        spectral_blocks<MAT>(S, mu[e], lam[e], dtSq * vol[e], true, Aw, Bw);
        double *row = pack[lane];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                row[3 * r + c] = U.m[r][c];
                row[21 + 3 * r + c] = Aw.m[r][c];
            }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) row[30 + 4 * c + k] = Bw[c][k];
        // y_v[b] = sum_j V[j][b] c_v[j]
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            double cv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) cv[j] = (v == 0) ? (-Ai[0][j] - Ai[1][j] - Ai[2][j]) : Ai[v - 1][j];
#pragma unroll
            for (int b = 0; b < 3; ++b) row[9 + 3 * v + b] = V.m[0][b] * cv[0] + V.m[1][b] * cv[1] + V.m[2][b] * cv[2];
        }
    }
    __syncthreads();
    const int nloc = min(64, nT - blockIdx.x * 64);
    const int pr = lane & 15, v = pr >> 2, w = pr & 3;
#pragma unroll 2
    for (int trip = 0; trip < 16; ++trip) {
        const int le = 4 * trip + (lane >> 4);
        if (le >= nloc) continue;
        const double *row = pack[le];
        double Um[3][3], Aw[3][3], yv[3], yw[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            yv[r] = row[9 + 3 * v + r];
            yw[r] = row[9 + 3 * w + r];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Um[r][c] = row[3 * r + c];
                Aw[r][c] = row[21 + 3 * r + c];
            }
        }
        double N[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) N[a][c] = Aw[a][c] * yv[a] * yw[c];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int p_ = c, q_ = (c + 1) % 3;
            const double b00 = row[30 + 4 * c], b01 = row[30 + 4 * c + 1], b10 = row[30 + 4 * c + 2],
                         b11 = row[30 + 4 * c + 3];
            N[p_][p_] += b00 * yv[q_] * yw[q_];
            N[p_][q_] += b01 * yv[q_] * yw[p_];
            N[q_][p_] += b10 * yv[p_] * yw[q_];
            N[q_][q_] += b11 * yv[p_] * yw[p_];
        }
        // H_vw = U N U^T
        double UN[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) UN[r][c] = Um[r][0] * N[0][c] + Um[r][1] * N[1][c] + Um[r][2] * N[2][c];
        double *out = He + (size_t)144 * (blockIdx.x * 64 + le) + 36 * v + 3 * w;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                out[12 * r + c] = UN[r][0] * Um[c][0] + UN[r][1] * Um[c][1] + UN[r][2] * Um[c][2];
    }
}

void launch_elem_hessians(const DevMesh &M, int mat, double dtSq, const double *x, double *He,
                          hipStream_t st, const int *elist, int nList)
{
    const int n = elist ? nList : M.nT;
    const int nb = (n + 63) / 64;
    if (nb <= 0) return;
    if (mat == 0)
        hipLaunchKernelGGL((elem_hessian_kernel<0>), dim3(nb), dim3(64), 0, st, M.T, M.A, M.nTp, n, M.mu, M.lam, M.vol,
                           x, dtSq, elist, He);
    else
        hipLaunchKernelGGL((elem_hessian_kernel<1>), dim3(nb), dim3(64), 0, st, M.T, M.A, M.nTp, n, M.mu, M.lam, M.vol,
                           x, dtSq, elist, He);
}

// global block-CSR assembly in gather form: thread = (block k, entry rc)
__global__ __launch_bounds__(256) void assemble_kernel(int nnzb, const int *__restrict__ blk_ptr,
                                                       const int *__restrict__ blk_ent,
                                                       const int *__restrict__ blk_row,
                                                       const int *__restrict__ adj_idx,
                                                       const uint8_t *__restrict__ fixed,
                                                       const double *__restrict__ mass,
                                                       const double *__restrict__ He,
                                                       const int *__restrict__ blist, double *__restrict__ Hval)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)nnzb * 9) return;
    // blist: the blocks this rank needs (sharded refresh): blk_ptr / blk_ent are then indexed by the position in the
    // list and name rows of the rank's compact He
    const int ki = (int)(t / 9), rc = (int)(t % 9);
    const int k = blist ? blist[ki] : ki;
    const int r = rc / 3, c = rc % 3;
    const int vr = blk_row[k], vc = adj_idx[k];
    double acc = 0.0;
    if (fixed[vr]) {
        acc = (vr == vc && r == c) ? 1.0 : 0.0;  // IglUtils.hpp:148-157
    } else if (!fixed[vc]) {
        // contributions four at a time: index loads first, then the four value loads, then the adds in list order
        const int b1 = blk_ptr[ki + 1];
        for (int i = blk_ptr[ki]; i < b1; i += 4) {
            int ent[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ent[u] = (i + u < b1) ? blk_ent[i + u] : -1;
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = ent[u] >> 4, a = (ent[u] >> 2) & 3, b = ent[u] & 3;
                const double *src = ent[u] >= 0 ? He + (size_t)144 * e + 12 * (3 * a + r) + 3 * b + c : &g_zero_slot;
                v[u] = *src;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (ent[u] >= 0) acc += v[u];
        }
        if (vr == vc && r == c) acc += mass[vr];  // DOTTimeStepper.cpp:598-607
    }
    Hval[(size_t)9 * k + rc] = acc;
}

void launch_assemble(const DevMesh &M, const double *He, double *Hval, hipStream_t st, const int *blist, int nList,
                     const int *blk_ptr, const int *blk_ent, const double *mass)
{
    const long long tot = (long long)(blist ? nList : M.nnzb) * 9;
    if (tot <= 0) return;
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, (int)(tot / 9),
                       blist ? blk_ptr : M.blk_ptr, blist ? blk_ent : M.blk_ent, M.blk_row, M.adj_idx, M.fixed,
                       mass ? mass : M.mass, He, blist, Hval);
}

// dense principal sub-matrices: W_s[(3i+r)*lda + 3j+c] = H[l2g_i, l2g_j][r][c]
__global__ __launch_bounds__(256) void dense_fill_kernel(long long nfill9, const long long *__restrict__ dst,
                                                         const int *__restrict__ src,
                                                         const double *__restrict__ Hval,
                                                         double *__restrict__ W)
{
    // one thread per scalar of a 3x3 block: dst[t] = its place in the factor storage, or -1 when that place is not
    // stored (the mirror copy right of a row block's diagonal tile in the compact layout)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nfill9) return;
    const long long d = dst[t];
    if (d >= 0) W[d] = Hval[(size_t)9 * src[t / 9] + t % 9];
}
__global__ void pad_identity_kernel(int npad, const long long *__restrict__ dst, double *__restrict__ W)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < npad) W[dst[t]] = 1.0;
}

void launch_dense_fill(const DevParts &P, const double *Hval, hipStream_t st)
{
    // the caller has cleared W (or the blocks of it a factorisation dirtied)
    if (P.nfill) {
        const long long tot = (long long)P.nfill * 9;
        hipLaunchKernelGGL(dense_fill_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, tot,
                           P.fill_dst, P.fill_src, Hval, P.W);
    }
    if (P.npad)
        hipLaunchKernelGGL(pad_identity_kernel, dim3((P.npad + 255) / 256), dim3(256), 0, st, P.npad,
                           P.pad_dst, P.W);
}

// ------------------------------------------------------------------------------------------------
// small state kernels
// ------------------------------------------------------------------------------------------------
struct Vec3Arg {
    double v[3];
};

// Optimizer::initX(2) (Optimizer.cpp:472-493, :580-581): x += dt v + dt^2 g on free vertices
__global__ void init_x_kernel(int nV, const uint8_t *__restrict__ fixed, const double *__restrict__ v,
                              double dt, Vec3Arg gdtsq, double *__restrict__ x)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nV) return;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double pd = fixed[i] ? 0.0 : dt * v[3 * i + d] + gdtsq.v[d];
        x[3 * i + d] = x[3 * i + d] + 1.0 * pd;
    }
}
void launch_init_x(int nV, const uint8_t *fixed, const double *v, double dt, const double *gdtsq,
                   double *x, hipStream_t st)
{
    Vec3Arg g = {{gdtsq[0], gdtsq[1], gdtsq[2]}};
    hipLaunchKernelGGL(init_x_kernel, dim3((nV + 255) / 256), dim3(256), 0, st, nV, fixed, v, dt, g, x);
}

// BE update (Optimizer.cpp:354-361) + computeXTilta (:585-610)
__global__ void be_update_kernel(int nV, const uint8_t *__restrict__ fixed, const double *__restrict__ x,
                                 double *__restrict__ xn, double *__restrict__ v,
                                 double *__restrict__ xt, double dt, Vec3Arg gdtsq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nV) return;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int k = 3 * i + d;
        const double xv = x[k];
        const double vel = (xv - xn[k]) / dt;
        v[k] = vel;
        xn[k] = xv;
        xt[k] = fixed[i] ? xv : xv + (vel * dt + gdtsq.v[d]);
    }
}
void launch_be_update(int nV, const uint8_t *fixed, double *x, double *xn, double *v, double *xt,
                      double dt, const double *gdtsq, hipStream_t st)
{
    Vec3Arg g = {{gdtsq[0], gdtsq[1], gdtsq[2]}};
    hipLaunchKernelGGL(be_update_kernel, dim3((nV + 255) / 256), dim3(256), 0, st, nV, fixed, x, xn, v, xt,
                       dt, g);
}

__global__ void scatter_rows_kernel(int n, const int *__restrict__ idx, const double *__restrict__ pos,
                                    double *__restrict__ x)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int v = idx[i];
    x[3 * v] = pos[3 * i];
    x[3 * v + 1] = pos[3 * i + 1];
    x[3 * v + 2] = pos[3 * i + 2];
}
void launch_scatter_rows(int n, const int *idx, const double *pos, double *x, hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(scatter_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, idx, pos, x);
}

#ifdef DOTMI_PAIR_TU
// the code object of this unit is loaded at the first use of one of its kernels (a few milliseconds): dotmi_create asks for it
// up front instead of leaving it to the first step that pairs
void warm_pair_unit()
{
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&backsolve_ctl_kernel));
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&elem_patch_kernel<1, true, 1, true, false>));
}
#endif

void launch_copy(int n, const double *src, double *dst, hipStream_t st)
{
    hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
}

}  // namespace dotmi
