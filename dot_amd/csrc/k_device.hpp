// k_device.hpp -- what the kernel translation units of libdotmi share on the device side (private).
//
// Units (round 6; one 4 100-line kernels.hip before):
//   k_element.hip     element pass: energy (+ inertia) partials, per-(patch, vertex) partial gradients, the fused line-search step,
//                     the paired / speculative instantiations
//   k_loopvec.hip     the loop's vector kernels: vertex gather, pair statistics, two-loop kernels, merges, SpMV + dots, owner-exchange
//                     packets, small state kernels
//   k_backsolve.hip   subdomain back-solve tiles, the loop controller (a launch of its own or workgroup 0 of the back-solve launch)
//   k_tilefactor.hip  block-sparse inverse-Cholesky on 64 x 64 tiles (FP64 MFMA), level and dataflow schedules
//   k_refresh.hip     once per step: element Hessians, global assembly, subdomain matrix fill
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
//   elem_patch_kernel         Energy.cpp:294-423 (F, SVD, Psi), :910-972 (P-hat, P, element gradient),
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
#pragma once
#include "dotmi_internal.hpp"
#include <hip/hip_ext.h>
#include "elem_math.hpp"

namespace dotmi {

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
// partialsT (round 6): the same values once more in COLUMN-major form, partialsT[t * NB_RED + block] -- the layout for partial
// arrays that EVERY workgroup of the next kernel reads (the y_i . z columns in front of the direction kernels): a wave's load of
// one column of 64 consecutive rows is then 4 lines of 128 bytes instead of 64 (rows are 168 bytes apart), which with ~600
// workgroups reading the same 43 KB was ~1 M line requests on a few L2 channels -- microseconds in front of every workgroup's
// first real load (tools/prof_dirstep.sh)
// NW: waves of the workgroup (4; 16 for the 1024-thread forms big meshes launch -- sm then holds NW * RED_K doubles and the
// waves' sums are added in a fixed pairwise tree, which for four waves is (s0 + s1) + (s2 + s3) as ever)
template <int NW = 4>
__device__ __forceinline__ void write_partials(double (&acc)[RED_K], int nvals, double *partials, double *sm,
                                               double *partialsT = nullptr)
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
    if (t < nvals) {
        double tw[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) tw[i] = sm[i * RED_K + t];
#pragma unroll
        for (int span = 1; span < NW; span *= 2)
#pragma unroll
            for (int i = 0; i + span < NW; i += 2 * span) tw[i] = tw[i] + tw[i + span];
        const double v = tw[0];
        partials[(size_t)blockIdx.x * RED_K + t] = v;
        if (partialsT) partialsT[(size_t)t * NB_RED + blockIdx.x] = v;
    }
}

// sum over the 8 lanes of an aligned lane group (fixed butterfly => deterministic); all 8 get the total
__device__ __forceinline__ double group8_sum(double v)
{
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

// the y_i . z partial columns of the NB_RED rows, summed per lane of wave 0 (lane l: rows l, l + 64, ...): every load of the lane
// requested before the first addition (a loop over the rows with a running sum was four dependent round trips -- ~6 us on the
// critical path of every kernel that starts with the two-loop's coefficients), the additions in row order as before
// transposed: c_partials is the column-major twin (write_partials' partialsT)
__device__ __forceinline__ void load_yz_partials(const double *__restrict__ c_partials, double (&c)[8], bool transposed = false)
{
    static_assert(NB_RED % 64 == 0 && HIST_MAX <= 8, "rows per lane, one transposed butterfly");
    constexpr int U = NB_RED / 64;
    double v[U][HIST_MAX];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i)
            v[u][i] = transposed ? c_partials[(size_t)i * NB_RED + threadIdx.x + 64 * u] : c_partials[(size_t)(threadIdx.x + 64 * u) * RED_K + i];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i) c[i] += v[u][i];
}

__device__ __forceinline__ double lane_bcast(double v, int srclane)   // (srclane: a constant)
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
// The coefficients of the second half of the two-loop -- ys_i, s_j . y_i, xi_i: 48 doubles that lie one behind the other in the
// loop state (DevLoop::L.ys, L.sy, X.xi) -- requested by wave 0 as ONE vector load, a coefficient per lane, together with the
// y_i . z partial columns at the kernel's start; until round 6 they were scalar loads inside the recurrence, a round trip to the
// memory the controller's XCD wrote on the critical path of every direction kernel.  delta: the recurrence of build_p_kernel
// (DOTTimeStepper.cpp:455-467), the same operations in the same order.
struct TwoLoopCoef {
    double v;
    static constexpr int NCOEF = HIST_MAX + HIST_MAX * HIST_MAX + HIST_MAX;
    __device__ __forceinline__ void request(const DevLoop *__restrict__ ctl)   // (wave 0)
    {
        static_assert(offsetof(DevLoop, X) == offsetof(DevLoop, L) + offsetof(LbfgsArgs, ys) + sizeof(double) * (HIST_MAX + HIST_MAX * HIST_MAX),
                      "ys, sy, xi contiguous");
        static_assert(NCOEF <= 64, "a coefficient per lane");
        const double *__restrict__ base = ctl->L.ys;
        v = (int)threadIdx.x < NCOEF ? base[threadIdx.x] : 0.0;
    }
    // wave 0, all lanes: c = this lane's sums of the partial columns (load_yz_partials); lane 0 stores delta[0 .. HIST_MAX)
    __device__ __forceinline__ void delta(const double (&c)[8], int m, double *__restrict__ out) const
    {
        const double tot = wave_sum8_transposed(c, threadIdx.x);
        double ct[HIST_MAX], rys[HIST_MAX];
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i) {
            ct[i] = __shfl(tot, 8 * i, 64);
            rys[i] = (i < m) ? 1.0 / lane_bcast(v, i) : 0.0;   // independent divisions, off the recurrence's dependent chain
        }
        double d[HIST_MAX];
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i) {
            d[i] = 0.0;
            if (i < m) {
                double yp = ct[i];
#pragma unroll
                for (int j = 0; j < HIST_MAX; ++j)
                    if (j < i) yp += d[j] * lane_bcast(v, HIST_MAX + HIST_MAX * j + i);   // s_j . y_i
                d[i] = lane_bcast(v, HIST_MAX + HIST_MAX * HIST_MAX + i) - yp * rys[i];
            }
            if (threadIdx.x == 0) out[i] = d[i];
        }
    }
};

// a zero to select instead of branching around a load (one copy per unit)
static __device__ const double g_zero_slot = 0.0;

}  // namespace dotmi
