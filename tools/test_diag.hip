// standalone check of chol_inv_base_kernel / chol_inv_node128_kernel against a host Cholesky
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../dot_amd/csrc/dotmi_internal.hpp"
using namespace dotmi;
int main()
{
    const int nmax = 256, n = 128, batch = 3, o = 64;
    std::vector<double> W((size_t)batch * nmax * nmax, 0.0), A((size_t)batch * n * n);
    srand(1);
    for (int b = 0; b < batch; ++b) {
        std::vector<double> M(n * n);
        for (auto &v : M) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = (i == j) ? n * 0.1 : 0.0;
                for (int k = 0; k < n; ++k) s += M[i * n + k] * M[j * n + k];
                A[(size_t)b * n * n + i * n + j] = s;
                W[(size_t)b * nmax * nmax + (size_t)(o + j) * nmax + o + i] = s;
            }
    }
    double *dW; int *info;
    hipMalloc(&dW, W.size() * 8); hipMalloc(&info, 4 * batch); hipMemset(info, 0, 4 * batch);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice);
        const int nn = mode == 0 ? 64 : 128;
        if (mode == 0) launch_chol_inv_base(dW, nmax, batch, o, info, 0);
        else launch_chol_inv_node128(dW, nmax, batch, o, info, 0);
        std::vector<double> R(W.size());
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(R.data(), dW, W.size() * 8, hipMemcpyDeviceToHost);
        int inf[3]; hipMemcpy(inf, info, 12, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int b = 0; b < batch; ++b) {
            // X(i,k) = memory row o+i, col o+k;  check X A X^T = I on the leading nn x nn
            for (int i = 0; i < nn; ++i)
                for (int j = 0; j < nn; ++j) {
                    double s = 0;
                    for (int k = 0; k < nn; ++k)
                        for (int l = 0; l < nn; ++l)
                            s += R[(size_t)b * nmax * nmax + (size_t)(o + i) * nmax + o + k] * A[(size_t)b * n * n + k * n + l] *
                                 R[(size_t)b * nmax * nmax + (size_t)(o + j) * nmax + o + l];
                    worst = std::fmax(worst, std::fabs(s - (i == j)));
                }
            double up = 0;
            for (int i = 0; i < nn; ++i)
                for (int k = i + 1; k < nn; ++k) up = std::fmax(up, std::fabs(R[(size_t)b * nmax * nmax + (size_t)(o + i) * nmax + o + k]));
            printf("mode %d batch %d info %d upper %.3g\n", mode, b, inf[b], up);
        }
        printf("mode %d (%s) err %s worst |X A X^T - I| = %.3e\n", mode, mode ? "node128" : "base64", hipGetErrorString(e), worst);
    }
    return 0;
}
