// k_loopvec.hip -- the loop's vector kernels (gather, pair statistics, two-loop, merges, SpMV + dots, packets, state)
// (one translation unit per kernel family since round 6: an edit to one family no longer moves the register allocation and
// scalar loads of the others; every unit is compiled once.  Conventions and the reference map: k_device.hpp)
#include "k_device.hpp"
#include "k_dirbody.hpp"

namespace dotmi {

#ifdef K_PROFILE
// stage stamps of the loop's vector kernels (tools/prof_loopkern.sh): [kernel][workgroup][stage], thread 0 of the workgroup
__device__ long long g_kprof[4][256][8];
extern "C" int dotmi_debug_kprof(long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kprof), sizeof(long long) * 4 * 256 * 8);
}
#define KSTAMP(kern, i) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_kprof[kern][blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define KSTAMP(kern, i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// vertex gather of element gradients (+ inertia), new L-BFGS pair and its statistics
// partial layout per block (m = L.m):
//   [0] |g_new|^2   [1] y_new.s_new   [2] s_new.g_new
//   [3+i] s_i.y_new   [3+HIST_MAX+j] s_new.y_j   [3+2*HIST_MAX+i] s_i.g_new
// ------------------------------------------------------------------------------------------------
// the stored pairs' vectors as a kernel holds them in registers (copied from the loop state in ONE batch of scalar loads)
struct HistView {
    int m;
    const double *s[HIST_MAX], *y[HIST_MAX];
    __device__ __forceinline__ void load(const LbfgsArgs &L)
    {
        m = L.m;
#pragma unroll
        for (int i = 0; i < HIST_MAX; ++i) {
            s[i] = L.s[i];
            y[i] = L.y[i];
        }
    }
};
template <class LV>
__device__ __forceinline__ void pair_stats_accum(int k, double gn, double sn, double yn, const LV &L,
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

// One lane per scalar degree of freedom k = 3 v + d: the vertex's per-patch partial gradients (usually 1-4 of them,
// contiguous in gpart, ascending patch) are added in that order, then the inertia term; the new L-BFGS pair and its
// statistics follow from the same registers.  GATHER_R dofs per lane and trip, one grid stride apart, so that the two
// dependent round trips of a trip (partial range -> partials) are paid once for all of them.
constexpr int GATHER_R = 4;
constexpr int GATHER_P = 4;   // partials requested together; a vertex with more takes further rounds
// NT: threads per workgroup (256; 1024 for meshes whose dofs take the 256-thread form more than two trips: the same dofs per
// workgroup with a quarter of them per lane -- four waves per SIMD instead of four dofs interleaved in one)
template <bool DEV, int NT = 256, int GR = GATHER_R>
__global__ __launch_bounds__(NT) void vertex_gather_kernel(
    int nV, const int2 *__restrict__ pp_rng, const double *__restrict__ gpart,
    const uint8_t *__restrict__ fixed, const double *__restrict__ mass, GatherArgs a, LbfgsArgs L,
    double *__restrict__ partials, const DevLoop *__restrict__ ctl)
{
    __shared__ double sm[(NT / 64) * RED_K];
    KSTAMP(0, 0);
    // the loop state this kernel needs, every load of it in front of the first branch (one round trip to the memory the
    // controller's XCD wrote, not one per dependent index: DevLoop, "resolved")
    double *__restrict__ hs_new = nullptr;
    HistView Lr;
    const double alpha = a.make_pair ? *a.alpha_dev : 0.0;
    if constexpr (DEV) {
        const int status = ctl->status;
        a.x = ctl->x_trial;
        a.g_old = ctl->g_cur;
        if (!a.stage) a.g_new = ctl->g_trial;   // stage: this rank's partial gradient goes to the buffer the host named
        a.s_new = ctl->s_new;
        a.y_new = ctl->y_new;
        if (a.hp) hs_new = ctl->hs_new;
        Lr.load(ctl->L);
        if (status != 0) return;
    } else {
        Lr.load(L);
    }
    double acc[RED_K];
#pragma unroll
    for (int j = 0; j < RED_K; ++j) acc[j] = 0.0;
    const VList vl{a.vlist, a.nlist};   // owner exchange: only the held vertices are visited
    const int n = vl_count3(vl, 3 * nV), G = gridDim.x * blockDim.x;
    constexpr int R = GR;
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
        KSTAMP(0, 1);
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
        KSTAMP(0, 2);
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
        KSTAMP(0, 3);
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
    KSTAMP(0, 4);
    write_partials<NT / 64>(acc, a.make_pair ? RED_K : 1, partials, sm);
    KSTAMP(0, 5);
}

void launch_vertex_gather(const DevMesh &M, const DevPatches &PT, const GatherArgs &a, const LbfgsArgs &L,
                          double *partials, hipStream_t st, const DevLoop *ctl)
{
    const long long ndof = a.vlist ? 3ll * a.nlist : 3ll * M.nV;
    if (ctl && ndof > 2ll * GATHER_R * NB_RED * 256)   // (more than two trips of the 256-thread form: 1 M tets)
        hipLaunchKernelGGL((vertex_gather_kernel<true, 512, 2>), dim3(NB_RED), dim3(512), 0, st, M.nV, PT.pp_rng, PT.gpart,
                           M.fixed, M.mass, a, L, partials, ctl);
    else if (ctl)
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
        load_yz_partials(c_partials, c);   // (c_blocks == NB_RED; columns >= m: unused)
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
// the walk over dof k's list in its wave-interleaved form (DevParts::mt_il): u = sum over the subdomains of (sum over the
// subdomain's tiles of their partial) -- the additions of the CSR walk in its order; k >> 6 is uniform over the wave
__device__ __forceinline__ double merge_walk_interleaved(int k, const int2 *__restrict__ mt_wave, const int *__restrict__ mt_il,
                                                         const double *__restrict__ ppart, double u)
{
    const int2 wb = mt_wave[k >> 6];
    const int *__restrict__ lst = mt_il + wb.x + (k & 63);
    double ps = 0.0;
    for (int q0 = 0; q0 < wb.y; q0 += MT_CH) {
        int off[MT_CH];
#pragma unroll
        for (int q = 0; q < MT_CH; ++q) off[q] = (q0 + q < wb.y) ? lst[64 * (q0 + q)] : MT_PAD;
        double w[MT_CH];
#pragma unroll
        for (int q = 0; q < MT_CH; ++q) {
            const int o = off[q] < 0 ? ~off[q] : off[q];
            w[q] = (off[q] != MT_PAD) ? ppart[o] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < MT_CH; ++q)
            if (off[q] != MT_PAD) {
                if (off[q] < 0 && q0 + q > 0) {   // a new subdomain starts: close the previous one
                    u += ps;
                    ps = 0.0;
                }
                ps += w[q];
            }
    }
    return u + ps;
}
template <bool DEV>
__global__ __launch_bounds__(256) void merge_tiles_kernel(int n3, const int *__restrict__ mt_ptr,
                                                          const int *__restrict__ mt_ent, const int *__restrict__ dup,
                                                          const double *__restrict__ ppart, LbfgsArgs L, int with_dots,
                                                          int divide, double *__restrict__ z,
                                                          double *__restrict__ partials, const DevLoop *__restrict__ ctl, VList vl,
                                                          const int2 *__restrict__ mt_wave, const int *__restrict__ mt_il)
{
    __shared__ double sm[4 * RED_K];
    if constexpr (DEV) {
        if (ctl->status != 0 || ctl->phase != 0) return;
    }
    const bool il = mt_il != nullptr && vl.v == nullptr;
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
        const int e0 = il ? 0 : mt_ptr[k], e1 = il ? 0 : mt_ptr[k + 1];
        const int d = divide ? dup[k / 3] : 1;
        double yk[HIST_MAX];
        if (with_dots) {
#pragma unroll
            for (int i = 0; i < HIST_MAX; ++i)
                if (i < Lr.m) yk[i] = Lr.y[i][k];
        }
        double zk = 0.0, ps = 0.0;
        if (il) zk = merge_walk_interleaved(k, mt_wave, mt_il, ppart, 0.0);
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
template <int NT = 256>
__global__ __launch_bounds__(NT) void merge_tiles_early_kernel(int n3, const int *__restrict__ mt_ptr,
                                                                const int *__restrict__ mt_ent, const int *__restrict__ dup,
                                                                const double *__restrict__ ppart, int first,
                                                                const double *__restrict__ zsum,
                                                                const int *__restrict__ vp_ptr, const int *__restrict__ vp_off,
                                                                const double *__restrict__ psub,
                                                                const uint8_t *__restrict__ ownMask, VList vl,
                                                                const uint8_t *__restrict__ kind, int pre,
                                                                double *__restrict__ zshare,
                                                                double *__restrict__ z, double *__restrict__ partials,
                                                                const DevLoop *__restrict__ ctl, const int2 *__restrict__ mt_wave,
                                                                const int *__restrict__ mt_il, double *__restrict__ partialsT)
{
    __shared__ double sm[(NT / 64) * RED_K];
    KSTAMP(1, 0);
    // (every load of the loop state in front of the first branch, the pointers resolved by the controller: one round trip)
    const int status = ctl->status, phase = ctl->phase, lm = ctl->L.m, pnew = ctl->pairNew;
    double *__restrict__ u_old = ctl->u_old;
    double *my_new_r = ctl->my_new;
    HistView Lr;
    Lr.load(ctl->L);
    const double *my[HIST_MAX];
    double xi[HIST_MAX];
#pragma unroll
    for (int i = 0; i < HIST_MAX; ++i) {
        my[i] = ctl->Lmy[i];
        xi[i] = ctl->X.xi[i];
    }
    if (status != 0 || phase != 0) return;
    const bool il = mt_il != nullptr && vl.v == nullptr && !psub && !zsum;
    const int m = first ? 0 : lm;
    const bool pairNew = !first && pnew != 0 && m > 0;
    double *my_new = pairNew ? my_new_r : nullptr;
#pragma unroll
    for (int i = 0; i < HIST_MAX; ++i) {
        if (!(i < m)) {
            my[i] = nullptr;
            xi[i] = 0.0;
        }
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
        } else if (!zsum && !il) {
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
        KSTAMP(1, 1);
        // zsum (sharded subdomains): the all-reduced sum over every rank's subdomains (merge_tiles_kernel without the division
        // into a staging buffer, then the collective -- on the staging buffer, so that a slot whose merge is gated off leaves z
        // alone); only the division and the history terms are left
        double u = zsum ? zsum[k] : 0.0, ps = 0.0;
        if (il) u = merge_walk_interleaved(k, mt_wave, mt_il, ppart, 0.0);
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
        KSTAMP(1, 2);
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
    KSTAMP(1, 3);
    if (partials) write_partials<NT / 64>(acc, HIST_MAX, partials, sm, partialsT);   // (nullptr: the y_i . z came with the packet, yz_pre_kernel)
    KSTAMP(1, 4);
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
                        double *zshare, double *partialsT)
{
    const bool split = !P.mt_ptr;
    // (a thread of the 256-thread form takes a dof per trip: beyond two trips -- 1 M tets: eight -- workgroups of 1024 threads)
    const long long ndof = vl.v ? 3ll * vl.n : 3ll * M.nV;
    if (ndof > 2ll * NB_RED * 256)
        hipLaunchKernelGGL(merge_tiles_early_kernel<512>, dim3(NB_RED), dim3(512), 0, st, 3 * M.nV, P.mt_ptr, P.mt_ent, P.dup, P.ppart,
                           first, zsum, split ? P.vp_ptr : nullptr, split ? P.vp_off : nullptr,
                           split ? (const double *)P.psub : nullptr, ownMask, vl, kind, pre, zshare, z, partials, ctl, P.mt_wave, P.mt_il,
                           partialsT);
    else
    hipLaunchKernelGGL(merge_tiles_early_kernel<256>, dim3(NB_RED), dim3(256), 0, st, 3 * M.nV, P.mt_ptr, P.mt_ent, P.dup, P.ppart,
                       first, zsum, split ? P.vp_ptr : nullptr, split ? P.vp_off : nullptr,
                       split ? (const double *)P.psub : nullptr, ownMask, vl, kind, pre, zshare, z, partials, ctl, P.mt_wave, P.mt_il, partialsT);
}

void launch_merge(const DevMesh &M, const DevParts &P, const LbfgsArgs &L, double *z, double *partials,
                  int with_dots, hipStream_t st, const DevLoop *ctl, VList vl)
{
    if (P.mt_ptr) {   // (launch_gemv left the tile partials in ppart and skipped the reduce)
        if (ctl)
            hipLaunchKernelGGL(merge_tiles_kernel<true>, dim3(NB_RED), dim3(256), 0, st, 3 * M.nV, P.mt_ptr, P.mt_ent, P.dup,
                               P.ppart, L, with_dots & 1, (with_dots >> 1) & 1, z, partials, ctl, vl, P.mt_wave, P.mt_il);
        else
            hipLaunchKernelGGL(merge_tiles_kernel<false>, dim3(NB_RED), dim3(256), 0, st, 3 * M.nV, P.mt_ptr, P.mt_ent, P.dup,
                               P.ppart, L, with_dots & 1, (with_dots >> 1) & 1, z, partials, ctl, vl, P.mt_wave, P.mt_il);
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
#pragma unroll
                    for (int i = 0; i < 9; ++i) h[u][i] = Hval[hval_idx(kb[u] + t, i)];
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

// Early order: build_p and spmv_dots in one launch (the body: k_dirbody.hpp, spmv_zp_body).  Device loop only.
__global__ __launch_bounds__(256) void spmv_zp_kernel(int nV, int v0, int v1, const uint8_t *__restrict__ rowMask,
                                                      const uint8_t *__restrict__ ownMask, const int *__restrict__ adj_ptr,
                                                      const int *__restrict__ adj_idx,
                                                      const double *__restrict__ Hval, const double *__restrict__ z,
                                                      const double *__restrict__ c_partials, int c_blocks,
                                                      double *__restrict__ p, double *__restrict__ Hp,
                                                      double *__restrict__ partials, const DevLoop *__restrict__ ctl, VList vl,
                                                      double *__restrict__ partialsT)
{
    __shared__ double sm[8];
    __shared__ double delta[HIST_MAX];
    spmv_zp_body(nV, v0, v1, rowMask, ownMask, adj_ptr, adj_idx, Hval, z, c_partials, c_blocks, p, Hp, partials, ctl, vl, sm, delta,
                 (int)blockIdx.x, (int)gridDim.x, partialsT);
}
// the same rows by workgroups of 1024 threads, one row per lane group (four waves per SIMD)
__global__ __launch_bounds__(1024) void spmv_zp_wide_kernel(int nV, int v0, int v1, const uint8_t *__restrict__ rowMask,
                                                            const uint8_t *__restrict__ ownMask, const int *__restrict__ adj_ptr,
                                                            const int *__restrict__ adj_idx, const double *__restrict__ Hval,
                                                            const double *__restrict__ z, const double *__restrict__ c_partials,
                                                            int c_blocks, double *__restrict__ p, double *__restrict__ Hp,
                                                            double *__restrict__ partials, const DevLoop *__restrict__ ctl, VList vl,
                                                            double *__restrict__ partialsT)
{
    __shared__ double sm[32];
    __shared__ double delta[HIST_MAX];
    spmv_zp_body<1024, 1>(nV, v0, v1, rowMask, ownMask, adj_ptr, adj_idx, Hval, z, c_partials, c_blocks, p, Hp, partials, ctl, vl, sm,
                          delta, (int)blockIdx.x, (int)gridDim.x, partialsT);
}

void launch_spmv_zp(const DevMesh &M, const double *Hval, const double *z, const double *c_partials, double *p, double *Hp,
                    double *partials, hipStream_t st, const DevLoop *ctl, int v0, int v1, const uint8_t *rowMask,
                    const uint8_t *ownMask, VList vl, bool ctrans, double *partialsT)
{
    if (v1 < 0) v1 = M.nV;
    // Rows beyond one trip of the 256-thread form (256 workgroups x 32 lane groups x 3 rows): workgroups of 1024 threads, a
    // row per lane group -- four waves per SIMD hide what the interleaving of three rows inside one wave cannot once the kernel
    // is bound by throughput (1 M tets: 76.9 -> 56.9 us; bar17K, one trip: 11.8 -> 12.9, stays on 256 threads).
    // DOTMI_SPMV_WIDE=0 / 1 forces the form.
    static const int wideEnv = getenv("DOTMI_SPMV_WIDE") ? atoi(getenv("DOTMI_SPMV_WIDE")) : -1;
    const int nrows = vl.v ? vl.n : M.nV;
    const bool wide = wideEnv >= 0 ? wideEnv != 0 : nrows > NB_RED * 32 * SPMV_R;
    if (wide)
        hipLaunchKernelGGL(spmv_zp_wide_kernel, dim3(NB_RED), dim3(1024), 0, st, M.nV, v0, v1, rowMask, ownMask, M.adj_ptr, M.adj_idx,
                           Hval, z, c_partials, ctrans ? -NB_RED : NB_RED, p, Hp, partials, ctl, vl, partialsT);
    else
    hipLaunchKernelGGL(spmv_zp_kernel, dim3(NB_RED), dim3(256), 0, st, M.nV, v0, v1, rowMask, ownMask, M.adj_ptr, M.adj_idx, Hval,
                       z, c_partials, ctrans ? -NB_RED : NB_RED, p, Hp, partials, ctl, vl, partialsT);
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

void launch_copy(int n, const double *src, double *dst, hipStream_t st)
{
    hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
}

}  // namespace dotmi
