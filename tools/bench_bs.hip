// micro-benchmark harness for back-solve kernel variants (timing only; data is synthetic)
#include "../dot_amd/csrc/k_backsolve.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace dotmi;
#define CK(x) do{ auto e_=(x); if((int)e_!=0){printf("fail %s -> %d\n",#x,(int)e_); exit(1);} }while(0)

// variant A: stream only (each lane sums what it loads), same tiling / same loads as backsolve_kernel<256,5>
template <int THREADS, int MAXCH, int MODE, int SUBV = 8, bool FULL = false>
__global__ __launch_bounds__(THREADS) void variant_kernel(const int4 *__restrict__ job, const int *__restrict__ psize,
                                                          const double *__restrict__ W, int nmax, double *__restrict__ out)
{
    constexpr int NW = THREADS / 64;
    __shared__ double sm[2][NW][8];
    const int4 jb = job[blockIdx.x];
    const int s = jb.x, i0 = jb.y;
    const int ns = psize[s];
    const int len = min(i0 + BS_ROWS, ns);
    const int ncol = FULL ? nmax : ((len + 127) & ~127);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double *Ws = W + (size_t)s * nmax * nmax;
    double2 pacc[MAXCH];
#pragma unroll
    for (int m = 0; m < MAXCH; ++m) pacc[m] = make_double2(0.0, 0.0);
#pragma unroll 1
    for (int sb = 0; sb < BS_ROWS / SUBV; ++sb) {
        const int ib = i0 + sb * SUBV;
        if (ib >= ns) break;
        double2 y[SUBV][MAXCH];
#pragma unroll
        for (int rr = 0; rr < SUBV; ++rr) {
            const double *row = Ws + (size_t)min(ib + rr, ns - 1) * nmax;
#pragma unroll
            for (int m = 0; m < MAXCH; ++m) {
                const int c = 2 * tid + 2 * THREADS * m;
                y[rr][m] = (c < ncol) ? *reinterpret_cast<const double2 *>(row + c) : make_double2(0.0, 0.0);
            }
        }
        if (MODE == 0) {  // stream only
#pragma unroll
            for (int m = 0; m < MAXCH; ++m)
#pragma unroll
                for (int rr = 0; rr < SUBV; ++rr) { pacc[m].x += y[rr][m].x; pacc[m].y += y[rr][m].y; }
        } else {          // dots + per-wave reduce only (no LDS exchange, no barrier)
            double d[BS_SUB];
#pragma unroll
            for (int rr = 0; rr < BS_SUB; ++rr) {
                double acc = 0.0;
#pragma unroll
                for (int m = 0; m < MAXCH; ++m) acc += y[rr][m].x * 1.5 + y[rr][m].y * 0.5;
                d[rr] = acc;
            }
            if (MODE >= 2) {
#pragma unroll
                for (int rr = 0; rr < BS_SUB; ++rr)
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) d[rr] += __shfl_xor(d[rr], o, 64);
            }
            if (MODE >= 3) {
                if (lane < BS_SUB) sm[sb & 1][wv][lane] = d[lane & (BS_SUB - 1)];
                __syncthreads();
#pragma unroll
                for (int rr = 0; rr < BS_SUB; ++rr) { double a = 0; for (int w = 0; w < NW; ++w) a += sm[sb & 1][w][rr]; d[rr] = a; }
            }
#pragma unroll
            for (int m = 0; m < MAXCH; ++m)
#pragma unroll
                for (int rr = 0; rr < BS_SUB; ++rr) { pacc[m].x += d[rr] * y[rr][m].x; pacc[m].y += d[rr] * y[rr][m].y; }
        }
    }
    double acc = 0;
#pragma unroll
    for (int m = 0; m < MAXCH; ++m) acc += pacc[m].x + pacc[m].y;
    if (acc == 1.2345678) out[blockIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const int nParts = 32, ns = 2148, nmax = 2176;
    std::vector<int> psize(nParts, ns), dof_ptr(nParts + 1), dofmap((size_t)nParts * ns);
    for (int s = 0; s <= nParts; ++s) dof_ptr[s] = s * ns;
    for (size_t i = 0; i < dofmap.size(); ++i) dofmap[i] = (int)i;
    std::vector<int4> tiles;
    for (int s = 0; s < nParts; ++s) for (int i = 0; i < (ns + 63) / 64; ++i) tiles.push_back(make_int4(s, i * 64, i, 0));
    std::stable_sort(tiles.begin(), tiles.end(), [](const int4 &a, const int4 &b) { return a.y > b.y; });
    DevParts P{}; P.nParts = nParts; P.nmax = nmax; P.ntiles = (int)tiles.size(); P.nbmax = nmax / 64;
    CK(hipMalloc(&P.psize, 4 * nParts)); CK(hipMemcpy(P.psize, psize.data(), 4 * nParts, hipMemcpyHostToDevice));
    CK(hipMalloc(&P.dof_ptr, 4 * (nParts + 1))); CK(hipMemcpy(P.dof_ptr, dof_ptr.data(), 4 * (nParts + 1), hipMemcpyHostToDevice));
    CK(hipMalloc(&P.dofmap, 4 * dofmap.size())); CK(hipMemcpy(P.dofmap, dofmap.data(), 4 * dofmap.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&P.tile, 16 * tiles.size())); CK(hipMemcpy(P.tile, tiles.data(), 16 * tiles.size(), hipMemcpyHostToDevice));
    size_t wbytes = (size_t)nParts * nmax * nmax * 8;
    CK(hipMalloc(&P.W, wbytes)); CK(hipMemset(P.W, 0, wbytes));
    CK(hipMalloc(&P.ppart, (size_t)nParts * P.nbmax * nmax * 8)); CK(hipMalloc(&P.psub, (size_t)nParts * ns * 8));
    double *q; CK(hipMalloc(&q, (size_t)nParts * ns * 8)); CK(hipMemset(q, 0, (size_t)nParts * ns * 8));
    double *out; CK(hipMalloc(&out, 8 * tiles.size()));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double bytes = (double)nParts * ns * (ns + 1) / 2 * 8;
    auto timeit = [&](const char *name, auto fn) {
        fn(); CK(hipEventRecord(e0)); for (int r = 0; r < 50; ++r) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
        printf("%-46s %8.2f us  %7.1f GB/s\n", name, ms * 1e3, bytes / ms / 1e6);
    };
    timeit("library back-solve (kernel + reduce)", [&] { launch_gemv(P, q, 0); });
    timeit("kernel only", [&] { hipLaunchKernelGGL((backsolve_kernel<256, 5>), dim3(P.ntiles), dim3(256), 0, 0, P.tile, P.psize, P.dof_ptr, P.dofmap, P.W, P.nmax, q, P.ppart, P.nbmax); });
    timeit("variant: stream only (same loads)", [&] { hipLaunchKernelGGL((variant_kernel<256, 5, 0>), dim3(P.ntiles), dim3(256), 0, 0, P.tile, P.psize, P.W, nmax, out); });
    {
        const double b0 = bytes;
        auto t2 = [&](const char *name, double by, auto fn) {
            fn(); CK(hipEventRecord(e0)); for (int r = 0; r < 50; ++r) fn(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 50;
            printf("%-46s %8.2f us  %7.1f GB/s\n", name, ms * 1e3, by / ms / 1e6);
        };
        const double full = (double)nParts * ns * nmax * 8;
        t2("stream only, SUB=4 (triangle)", b0, [&] { hipLaunchKernelGGL((variant_kernel<256, 5, 0, 4>), dim3(P.ntiles), dim3(256), 0, 0, P.tile, P.psize, P.W, nmax, out); });
        t2("stream only, SUB=2 (triangle)", b0, [&] { hipLaunchKernelGGL((variant_kernel<256, 5, 0, 2>), dim3(P.ntiles), dim3(256), 0, 0, P.tile, P.psize, P.W, nmax, out); });
        t2("stream only, SUB=8 FULL square", full, [&] { hipLaunchKernelGGL((variant_kernel<256, 5, 0, 8, true>), dim3(P.ntiles), dim3(256), 0, 0, P.tile, P.psize, P.W, nmax, out); });
        t2("stream only, SUB=2 FULL square", full, [&] { hipLaunchKernelGGL((variant_kernel<256, 5, 0, 2, true>), dim3(P.ntiles), dim3(256), 0, 0, P.tile, P.psize, P.W, nmax, out); });
    }
    timeit("variant: + dots + axpy, no reduce", [&] { hipLaunchKernelGGL((variant_kernel<256, 5, 1>), dim3(P.ntiles), dim3(256), 0, 0, P.tile, P.psize, P.W, nmax, out); });
    timeit("variant: + full wave butterflies", [&] { hipLaunchKernelGGL((variant_kernel<256, 5, 2>), dim3(P.ntiles), dim3(256), 0, 0, P.tile, P.psize, P.W, nmax, out); });
    timeit("variant: + LDS exchange + barrier", [&] { hipLaunchKernelGGL((variant_kernel<256, 5, 3>), dim3(P.ntiles), dim3(256), 0, 0, P.tile, P.psize, P.W, nmax, out); });
    return 0;
}
