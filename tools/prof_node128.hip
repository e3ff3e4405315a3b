// where do the ~60 us of chol_inv_node128_kernel go?  Phase time stamps (wall_clock64, 100 MHz) of workgroup 0.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DNODE128_PROFILE -o tools/prof_node128 tools/prof_node128.hip
#include "../dot_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
using namespace dotmi;
int main()
{
    const int nmax = 2368, batch = 32, o = 1024;
    std::vector<double> W((size_t)nmax * nmax, 0.0);
    srand(1);
    const int n = 128;
    std::vector<double> M(n * n);
    for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double s = (i == j) ? n * 0.1 : 0.0;
            for (int k = 0; k < n; ++k) s += M[i * n + k] * M[j * n + k];
            W[(size_t)(o + j) * nmax + o + i] = s;
        }
    double *dW; int *info;
    hipMalloc(&dW, (size_t)batch * W.size() * 8); hipMalloc(&info, 4 * batch); hipMemset(info, 0, 4 * batch);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        for (int b = 0; b < batch; ++b) hipMemcpy(dW + (size_t)b * W.size(), W.data(), W.size() * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        launch_chol_inv_node128(dW, nmax, batch, o, info, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long t[32];
        hipMemcpyFromSymbol(t, HIP_SYMBOL(g_node128_prof), sizeof(t));
        printf("launch %.1f us |", ms * 1e3);
        const char *names[] = {"load H11,H12", "chol_inv 64 #1", "load H22", "3 gemm64", "store X11", "chol_inv 64 #2", "gemm64", "store"};
        for (int i = 0; i < 8; ++i) printf(" %s %.2f |", names[i], (t[i + 1] - t[i]) / 100.0);
        printf(" total %.2f us\n", (t[8] - t[0]) / 100.0);
    }
    return 0;
}
