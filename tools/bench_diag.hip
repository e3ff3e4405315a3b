// Where the time of a diagonal tile task goes: the 64 x 64 inverse-Cholesky step (block_chol_inv<64>, k_tilefactor.hip) and its
// parts -- the 16 x 16 register base case and the small matrix-core products -- timed with the 100 MHz wall clock inside one
// 512-thread workgroup (and 32 of them side by side), results checked against a host Cholesky.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I dot_amd/csrc -o tools/bench_diag tools/bench_diag.hip
#include "k_tilefactor.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do{ auto e_=(x); if((int)e_!=0){printf("fail %s -> %d\n",#x,(int)e_); exit(1);} }while(0)
namespace dotmi {
template <int WHAT>
__global__ __launch_bounds__(512, 2) void diag_bench(const double *__restrict__ A, double *__restrict__ Xout,
                                                     long long *__restrict__ stamps, int reps)
{
    constexpr int NB = CHOL_NB, LD = NB + 1;
    __shared__ double La[NB][LD], Lb[NB][LD], T32[32][33], T16[16][17];
    const int tid = threadIdx.x;
    long long tot = 0;
    for (int r = 0; r < reps; ++r) {
        for (int idx = tid; idx < NB * NB; idx += 512) La[idx / NB][idx % NB] = A[idx];
        __syncthreads();
        const long long t0 = wall_clock64();
        if constexpr (WHAT == 0) block_chol_inv<64>(La, Lb, 0, T32, T16, tid);
        if constexpr (WHAT == 1) block_chol_inv<16>(La, Lb, 0, T32, T16, tid);
        if constexpr (WHAT == 2)
            mfma_gemm_small<16>([&](int i, int k) { return La[i][k]; }, [&](int k, int j) { return La[k][16 + j]; },
                                [&](int i, int j, double v) { T16[i][j] = v; }, tid);
        if constexpr (WHAT == 3)
            mfma_gemm_small<32>([&](int i, int k) { return La[i][k]; }, [&](int k, int j) { return La[k][32 + j]; },
                                [&](int i, int j, double v) { T32[i][j] = v; }, tid);
        if constexpr (WHAT == 4) block_chol_inv<32>(La, Lb, 0, T32, T16, tid);
        if constexpr (WHAT == 5) block_chol_inv<64, true>(La, Lb, 0, T32, T16, tid);
        if constexpr (WHAT == 6) block_chol_inv<16, true>(La, Lb, 0, T32, T16, tid);
        __syncthreads();
        tot += wall_clock64() - t0;
    }
    if (tid == 0) stamps[blockIdx.x] = tot;
    if (WHAT == 0 || WHAT == 5)
        for (int idx = tid; idx < NB * NB; idx += 512) Xout[(size_t)blockIdx.x * NB * NB + idx] = Lb[idx / NB][idx % NB];
}
}  // namespace dotmi
int main()
{
    const int N = 64;
    std::vector<double> B(N * N), A(N * N, 0.0);
    srand(7);
    for (auto &v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 0;
            for (int k = 0; k < N; ++k) s += B[i * N + k] * B[j * N + k];
            A[i * N + j] = s + (i == j ? 4.0 : 0.0);
        }
    // host reference: L = chol(A), X = L^-1
    std::vector<double> L(A), X(N * N, 0.0);
    for (int k = 0; k < N; ++k) {
        L[k * N + k] = std::sqrt(L[k * N + k]);
        for (int i = k + 1; i < N; ++i) L[i * N + k] /= L[k * N + k];
        for (int j = k + 1; j < N; ++j)
            for (int i = j; i < N; ++i) L[i * N + j] -= L[i * N + k] * L[j * N + k];
    }
    for (int c = 0; c < N; ++c)
        for (int i = c; i < N; ++i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s -= L[i * N + k] * X[k * N + c];
            X[i * N + c] = s / L[i * N + i];
        }
    double *dA, *dX; long long *dS;
    const int G = 32, reps = 50;
    CK(hipMalloc(&dA, sizeof(double) * N * N)); CK(hipMalloc(&dX, sizeof(double) * N * N * G)); CK(hipMalloc(&dS, 8 * G));
    CK(hipMemcpy(dA, A.data(), sizeof(double) * N * N, hipMemcpyHostToDevice));
    auto run = [&](const char *name, auto kern, int grid, bool check) {
        for (int w = 0; w < 2; ++w) {
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, dA, dX, dS, reps);
            CK(hipDeviceSynchronize());
        }
        std::vector<long long> s(grid);
        CK(hipMemcpy(s.data(), dS, 8 * grid, hipMemcpyDeviceToHost));
        double mean = 0;
        for (auto v : s) mean += v;
        mean /= grid * (double)reps * 100.0;
        printf("%-44s %2d workgroup(s): %7.2f us", name, grid, mean);
        if (check) {
            std::vector<double> h(N * N);
            CK(hipMemcpy(h.data(), dX, sizeof(double) * N * N, hipMemcpyDeviceToHost));
            double err = 0;
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) err = std::fmax(err, std::fabs(h[i * N + j] - X[i * N + j]));
            printf("   max |X - X_host| %.2e", err);
        }
        printf("\n");
    };
    for (int grid : {1, 32}) {
        run("block_chol_inv<64>", dotmi::diag_bench<0>, grid, true);
        run("block_chol_inv<32>", dotmi::diag_bench<4>, grid, false);
        run("block_chol_inv<16> (one row per lane)", dotmi::diag_bench<1>, grid, false);
        run("mfma_gemm_small<16>", dotmi::diag_bench<2>, grid, false);
        run("mfma_gemm_small<32>", dotmi::diag_bench<3>, grid, false);
        run("block_chol_inv<64, FAST> (blocks factored per lane)", dotmi::diag_bench<5>, grid, true);
        run("block_chol_inv<16, FAST>", dotmi::diag_bench<6>, grid, false);
    }
    return 0;
}
