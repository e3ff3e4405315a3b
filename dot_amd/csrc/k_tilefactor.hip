// k_tilefactor.hip -- subdomain factorisation: block-sparse inverse-Cholesky on 64 x 64 tiles (CHOLMODSolver.cpp:143 via DOTTimeStepper.cpp:363-377)
// (one translation unit per kernel family since round 6: an edit to one family no longer moves the register allocation and
// scalar loads of the others; every unit is compiled once.  Conventions and the reference map: k_device.hpp)
#include "k_device.hpp"

namespace dotmi {

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

}  // namespace dotmi
