// k_dirbody.hpp -- the direction kernel of the early order as a device function (private): p = z + sum_j delta_j s_j, H p, the
// partial sums of p.g and p.Hp (DOTTimeStepper.cpp:455-467 + Optimizer.cpp:1076-1093).  Shared by k_loopvec.hip (a launch of its
// own) and k_dirstep.hip (one population of the speculative unit-step launch).
#pragma once
#include "k_device.hpp"

namespace dotmi {

constexpr int SPMV_R = 3;   // block rows a lane group works on at a time
#ifdef DS_PROFILE
static __device__ long long g_sv_prof[256][6];   // (tools/prof_dirstep.sh: stamps inside the direction rows' workgroups)
#define SV_STAMP(i) do { if (threadIdx.x == 0 && bidx < 256) g_sv_prof[bidx][i] = wall_clock64(); } while (0)
#else
#define SV_STAMP(i) do { } while (0)
#endif

// Early order: build_p and spmv_dots in one launch.  The sparse product runs on z (ready when the launch starts); the
// history terms of the direction and of H p are local to a row: p_v = z_v + sum_j delta_j s_j[v] (summed over the row's
// lane group, pair j on lane j), (H p)_v = (H z)_v + sum_j delta_j (H s_j)[v] with the H s_j cached beside the s_j (H is fixed
// during a step; H s_new = alpha H p is written by the vertex gather).  delta comes from the y_i . z partials in wave 0's
// prologue (requested first, finished behind the column loop).  Device loop only.
// (a device function: the kernel of its own -- spmv_zp_kernel, k_loopvec.hip -- or one of the two workgroup populations of the
// speculative unit-step launch -- dirstep_kernel, k_dirstep.hip; bidx / nblocks: this workgroup's index and the number of
// workgroups of the population; sm: 8 doubles, delta: HIST_MAX doubles of LDS)
// NT threads per workgroup, R block rows per lane group and trip (256 x 3: one wave per SIMD with three rows' loads interleaved;
// 1024 x 1: four waves per SIMD, a row each -- the same rows per workgroup and trip, the latencies overlapped by the hardware's
// wave scheduling instead of by interleaving inside one wave; sm: 2 NT / 64 doubles)
template <int NT = 256, int R = SPMV_R>
__device__ __forceinline__ void spmv_zp_body(int nV, int v0, int v1, const uint8_t *__restrict__ rowMask,
                                             const uint8_t *__restrict__ ownMask, const int *__restrict__ adj_ptr,
                                             const int *__restrict__ adj_idx, const double *__restrict__ Hval,
                                             const double *__restrict__ z, const double *__restrict__ c_partials, int c_blocks,
                                             double *__restrict__ p, double *__restrict__ Hp, double *__restrict__ partials,
                                             const DevLoop *__restrict__ ctl, VList vl, double *sm, double *delta, int bidx,
                                             int nblocks, double *__restrict__ partialsT = nullptr)
{
    // the loop state this body needs, every load of it in front of the first branch (one round trip, not one per index)
    const int status = ctl->status, phase = ctl->phase, m = ctl->L.m;
    const double *__restrict__ g = ctl->g_cur;
    const double *hs_[HIST_MAX], *hhs_[HIST_MAX];
#pragma unroll
    for (int j = 0; j < HIST_MAX; ++j) {
        hs_[j] = ctl->L.s[j];
        hhs_[j] = ctl->Lhs[j];
    }
    // wave 0: the y_i . z partial columns and the recurrence's coefficients are requested now ...
    double c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) c[i] = 0.0;
    TwoLoopCoef coef;
    if (threadIdx.x < 64) {
        coef.request(ctl);
        load_yz_partials(c_partials, c, c_blocks < 0);   // (|c_blocks| == NB_RED: every caller's; negative: the column-major twin)
    }
    if (status != 0 || phase != 0) return;
    bool haveDelta = false;
    const int sub = threadIdx.x & 7;
    double mydelta = 0.0;   // delta of the pair this lane carries (lane j of a row's group: pair j)
    auto finish_delta = [&]() {   // ... and reduced here (build_p_kernel's prologue); one barrier, all threads
        if (threadIdx.x < 64) coef.delta(c, m, delta);
        __syncthreads();
        mydelta = (sub < m) ? delta[sub < HIST_MAX ? sub : 0] : 0.0;
        haveDelta = true;
    };
    SV_STAMP(0);
    double pg = 0, pHp = 0;
    const int ngroups = nblocks * (NT / 8);
    static_assert(HIST_MAX <= 8, "one lane of a row's group per stored pair");
    const double *__restrict__ hist_s = nullptr, *__restrict__ hist_hs = nullptr;
#pragma unroll
    for (int j = 0; j < HIST_MAX; ++j)
        if (j == sub && j < m) {
            hist_s = hs_[j];
            hist_hs = hhs_[j];
        }
    // (the trip count is the same for every thread of a workgroup: finish_delta's barrier sits inside the first trip)
    const int nrows = vl.v ? vl.n : nV;   // owner exchange: the rows of the held vertices only (p is zero elsewhere)
    for (int base = bidx * (NT / 8); base < nrows; base += R * ngroups) {
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
        SV_STAMP(1);
        for (int t = sub; t < nkmax; t += 8) {
            int col[R];
            double h[R][9], pc[R][3];
#pragma unroll
            for (int u = 0; u < R; ++u) col[u] = (t < nk[u]) ? adj_idx[kb[u] + t] : -1;
#pragma unroll
            for (int u = 0; u < R; ++u)
                if (t < nk[u]) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) h[u][i] = Hval[hval_idx(kb[u] + t, i)];
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
        SV_STAMP(2);
        if (!haveDelta) finish_delta();
        SV_STAMP(3);
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
    SV_STAMP(4);
    const double w0 = wave_sum(pg), w1 = wave_sum(pHp);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    if (lane == 0) {
        sm[w] = w0;
        sm[NW + w] = w1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // (a fixed pairwise tree over the waves' sums: (s0 + s1) + (s2 + s3) for four waves, as ever)
        double t0[NW], t1[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            t0[i] = sm[i];
            t1[i] = sm[NW + i];
        }
#pragma unroll
        for (int span = 1; span < NW; span *= 2)
#pragma unroll
            for (int i = 0; i + span < NW; i += 2 * span) {
                t0[i] = t0[i] + t0[i + span];
                t1[i] = t1[i] + t1[i + span];
            }
        partials[(size_t)bidx * RED_K] = t0[0];
        partials[(size_t)bidx * RED_K + 1] = t1[0];
        if (partialsT) {   // the two columns once more, column-major: what EVERY workgroup of the element pass reads (k_device.hpp, write_partials)
            partialsT[bidx] = t0[0];
            partialsT[NB_RED + bidx] = t1[0];
        }
    }
}


}  // namespace dotmi
