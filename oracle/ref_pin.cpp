// ref_pin.cpp -- driver that exposes pieces of the REFERENCE ITSELF through a C ABI, so the
// restatement in dot_oracle.c can be pinned against them.  TEST INFRASTRUCTURE ONLY.
//
// This file is ours; every function below calls into reference code that is compiled from the
// sources where they lie under /root/reference (see oracle/Makefile, target _ref/librefpin.so):
//   * src/Utils/SVD_EFTYCHIOS/Singular_Value_Decomposition_Helper.cpp  (the AVX SVD, K2)
//   * src/Utils/IglUtils.hpp   header-only templates makePD / makePD2d / dF_div_dx_mult<N>
//   * src/Utils/SIMD_DOUBLE_MACROS.hpp   ENERGY_* and PHAT_* macros (K3, K4)
//   * src/Utils/AutoFlipSVD.hpp          scalar implicit-QR SVD (used for F = I in the tolerance)
// The reference's Energy.cpp / Optimizer.cpp / DOTTimeStepper.cpp are NOT built: they include
// <tbb/tbb.h>, which this image does not have, and the round rules forbid stand-in headers.
#include <immintrin.h>
#include <cstdlib>
#include <cstring>

#include "IglUtils.hpp"
#include "AutoFlipSVD.hpp"
#include "SIMD_DOUBLE_MACROS.hpp"
#include "Singular_Value_Decomposition_Helper.h"

using namespace Singular_Value_Decomposition;

namespace {
const int CAP = 65536;  // an instantiated size (Singular_Value_Decomposition_Helper.cpp tail)
double *buf[30];
bool inited = false;
void init()
{
    if (inited) return;
    for (int i = 0; i < 30; ++i) buf[i] = (double *)aligned_alloc(64, sizeof(double) * CAP);
    inited = true;
}
}  // namespace

extern "C" {

// n 3x3 matrices, row-major; returns U, S, V row-major.  Mirrors the AoS->SoA staging of
// IglUtils.cpp:929-1085 (a11..a33 column-major naming: aRC = F(R-1,C-1)).
int ref_svd(int n, const double *F, double *U, double *S, double *V)
{
    init();
    if (n > CAP) return -1;
    int np = (n + 3) / 4 * 4;
    double **a = buf, **u = buf + 9, **v = buf + 18, **s = buf + 27;
    // order: 11,21,31,12,22,32,13,23,33
    for (int i = 0; i < np; ++i) {
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r)
                a[3 * c + r][i] = (i < n) ? F[9 * i + 3 * r + c] : (r == c ? 1.0 : 0.0);
    }
    Singular_Value_Decomposition_Size_Specific_Helper<double, 65536> h(
        a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], u[0], u[1], u[2], u[3], u[4], u[5],
        u[6], u[7], u[8], v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], s[0], s[1], s[2]);
    h.Run_Index_Range(0, np);
    for (int i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c)
            for (int r = 0; r < 3; ++r) {
                U[9 * i + 3 * r + c] = u[3 * c + r][i];
                V[9 * i + 3 * r + c] = v[3 * c + r][i];
            }
        S[3 * i] = s[0][i];
        S[3 * i + 1] = s[1][i];
        S[3 * i + 2] = s[2][i];
    }
    return 0;
}

void ref_autoflip_svd(const double *F, double *U, double *S, double *V)
{
    Eigen::Matrix3d Fm;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Fm(r, c) = F[3 * r + c];
    DOT::AutoFlipSVD<Eigen::Matrix3d> svd(Fm, Eigen::ComputeFullU | Eigen::ComputeFullV);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            U[3 * r + c] = svd.matrixU()(r, c);
            V[3 * r + c] = svd.matrixV()(r, c);
        }
    for (int i = 0; i < 3; ++i) S[i] = svd.singularValues()[i];
}

void ref_make_pd3(double *A)
{
    Eigen::Matrix3d M;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M(r, c) = A[3 * r + c];
    DOT::IglUtils::makePD(M);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[3 * r + c] = M(r, c);
}

void ref_make_pd2(double *B)
{
    Eigen::Matrix2d M;
    M << B[0], B[1], B[2], B[3];
    DOT::IglUtils::makePD2d(M);
    B[0] = M(0, 0); B[1] = M(0, 1); B[2] = M(1, 0); B[3] = M(1, 1);
}

// H (12x12 row-major) = two dF_div_dx_mult passes around a 9x9 M (Energy.cpp:767-769)
void ref_hessian_from_dPdF(const double *M, const double *A, double *H)
{
    Eigen::Matrix<double, 9, 9> Mm;
    Eigen::Matrix3d Am;
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) Mm(r, c) = M[9 * r + c];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Am(r, c) = A[3 * r + c];
    Eigen::Matrix<double, 12, 9> wdP_div_dx;
    Eigen::Matrix<double, 12, 12> hessian;
    Eigen::Matrix<double, 9, 9> Mt = Mm.transpose();
    DOT::IglUtils::dF_div_dx_mult<9>(Mt, Am, wdP_div_dx, false);
    Eigen::Matrix<double, 9, 12> Wt = wdP_div_dx.transpose();
    DOT::IglUtils::dF_div_dx_mult<12>(Wt, Am, hessian, true);
    for (int r = 0; r < 12; ++r)
        for (int c = 0; c < 12; ++c) H[12 * r + c] = hessian(r, c);
}

// mat: 0 FCR, 1 SNH.  n elements (any n; padded internally like Energy.cpp:852-907)
int ref_energy_phat(int mat, int n, const double *mu, const double *lam, const double *sig,
                    double *psi, double *phat)
{
    init();
    if (n > CAP) return -1;
    int np = (n + 3) / 4 * 4;
    double *Gmu = buf[0], *Glambda = buf[1], *Gs0 = buf[2], *Gs1 = buf[3], *Gs2 = buf[4];
    double *o0 = buf[5], *o1 = buf[6], *o2 = buf[7], *oe = buf[8];
    for (int i = 0; i < np; ++i) {
        Gmu[i] = i < n ? mu[i] : 0.0;
        Glambda[i] = i < n ? lam[i] : 1.0;
        Gs0[i] = i < n ? sig[3 * i] : 1.0;
        Gs1[i] = i < n ? sig[3 * i + 1] : 1.0;
        Gs2[i] = i < n ? sig[3 * i + 2] : 1.0;
    }
    __m256d vOne = _mm256_set1_pd(1.0), vOneHalf = _mm256_set1_pd(0.5), vTwo = _mm256_set1_pd(2.0),
            vThree = _mm256_set1_pd(3.0);
    for (int e = 0; e < np / 4; ++e) {
        __m256d vE, r0, r1, r2;
        if (mat == 0) {
            ENERGY_FIXED_COROTATED(e, vOne, vOneHalf, Gmu, Glambda, Gs0, Gs1, Gs2, vE);
            PHAT_FIXED_COROTATED(e, vOne, vTwo, Gmu, Glambda, Gs0, Gs1, Gs2, r0, r1, r2);
        } else {
            ENERGY_Stable_NeoHookean(e, vOne, vOneHalf, vThree, Gmu, Glambda, Gs0, Gs1, Gs2, vE);
            PHAT_Stable_NeoHookean(e, vOne, vTwo, Gmu, Glambda, Gs0, Gs1, Gs2, r0, r1, r2);
        }
        _mm256_store_pd(oe + 4 * e, vE);
        _mm256_store_pd(o0 + 4 * e, r0);
        _mm256_store_pd(o1 + 4 * e, r1);
        _mm256_store_pd(o2 + 4 * e, r2);
    }
    for (int i = 0; i < n; ++i) {
        psi[i] = oe[i];
        phat[3 * i] = o0[i];
        phat[3 * i + 1] = o1[i];
        phat[3 * i + 2] = o2[i];
    }
    return 0;
}

}  // extern "C"
