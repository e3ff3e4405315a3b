// k_refresh.hip -- once per step: element Hessians, global assembly, subdomain matrix fill (Energy.cpp:738-777, DOTTimeStepper.cpp:588-797)
// (one translation unit per kernel family since round 6: an edit to one family no longer moves the register allocation and
// scalar loads of the others; every unit is compiled once.  Conventions and the reference map: k_device.hpp)
#include "k_device.hpp"

namespace dotmi {

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
    Hval[hval_idx(k, rc)] = acc;
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
    if (d >= 0) W[d] = Hval[hval_idx(src[t / 9], (int)(t % 9))];
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

}  // namespace dotmi
