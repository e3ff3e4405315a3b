/*
 * dot_oracle.c -- CPU restatement of the reference DOT time-step.  TEST INFRASTRUCTURE ONLY:
 * see dot_oracle.h for who may load this and for the pinning status of each part.
 *
 * Reference paths are relative to /root/reference/src.  Nothing here is copied from the
 * reference: it is the same algorithm re-expressed in plain C on flat row-major arrays.
 */
#include "dot_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

void dor_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * small dense helpers
 * ---------------------------------------------------------------------------------------- */
static double det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

static void inv3(const double M[9], double R[9])
{
    double d = det3(M);
    double id = 1.0 / d;
    R[0] = (M[4] * M[8] - M[5] * M[7]) * id;
    R[1] = (M[2] * M[7] - M[1] * M[8]) * id;
    R[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    R[3] = (M[5] * M[6] - M[3] * M[8]) * id;
    R[4] = (M[0] * M[8] - M[2] * M[6]) * id;
    R[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    R[6] = (M[3] * M[7] - M[4] * M[6]) * id;
    R[7] = (M[1] * M[6] - M[0] * M[7]) * id;
    R[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}

/* cyclic Jacobi eigen-decomposition of a symmetric 3x3 (row-major).  Q columns = eigenvectors,
 * w ascending.  Used for the PSD clamp (IglUtils.hpp:253 uses Eigen::SelfAdjointEigenSolver)
 * and for V of the SVD (Main_Kernel_Body.hpp:48-91 does Jacobi on F^T F). */
static void sym_eig3(const double Ain[9], double w[3], double Q[9])
{
    double a[3][3] = {{Ain[0], Ain[1], Ain[2]}, {Ain[1], Ain[4], Ain[5]}, {Ain[2], Ain[5], Ain[8]}};
    double q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    static const int PQ[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        double dia = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-20 * dia) break;
        for (int k = 0; k < 3; ++k) {
            int p = PQ[k][0], r = PQ[k][1];
            double apq = a[p][r];
            if (apq == 0.0) continue;
            if (fabs(apq) < 1e-19 * (fabs(a[p][p]) + fabs(a[r][r]))) {
                a[p][r] = a[r][p] = 0.0;
                continue;
            }
            double theta = (a[r][r] - a[p][p]) / (2.0 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            a[p][p] -= t * apq;
            a[r][r] += t * apq;
            a[p][r] = a[r][p] = 0.0;
            int o = 3 - p - r;
            double aop = a[o][p], aor = a[o][r];
            a[o][p] = a[p][o] = c * aop - s * aor;
            a[o][r] = a[r][o] = s * aop + c * aor;
            for (int i = 0; i < 3; ++i) {
                double qp = q[i][p], qr = q[i][r];
                q[i][p] = c * qp - s * qr;
                q[i][r] = s * qp + c * qr;
            }
        }
    }
    int idx[3] = {0, 1, 2};
    double d[3] = {a[0][0], a[1][1], a[2][2]};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (d[idx[j]] > d[idx[j + 1]]) {
                int t = idx[j];
                idx[j] = idx[j + 1];
                idx[j + 1] = t;
            }
    for (int k = 0; k < 3; ++k) {
        w[k] = d[idx[k]];
        for (int i = 0; i < 3; ++i) Q[3 * i + k] = q[i][idx[k]];
    }
}

/* Rotation-variant SVD with the reference's conventions (Utils/SVD_EFTYCHIOS/Main_Kernel_Body.hpp,
 * driver IglUtils.cpp:929-1085): F = U diag(S) V^T, U,V in SO(3), S[0] >= S[1] >= |S[2]|,
 * sign(S[2]) = sign(det F).  Same structure (symmetric eigen-analysis of F^T F for V, then an
 * orthogonal factorisation of F V for U and S) with exact Jacobi rotations instead of the
 * reference's fixed 10 approximate sweeps. */
void dor_svd3(const double F[9], double U[9], double S[3], double V[9])
{
    double C[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = F[i] * F[j] + F[3 + i] * F[3 + j] + F[6 + i] * F[6 + j];
    double w[3], Q[9];
    sym_eig3(C, w, Q);
    /* descending order: column k of V = eigenvector of w[2-k] */
    for (int i = 0; i < 3; ++i) {
        V[3 * i + 0] = Q[3 * i + 2];
        V[3 * i + 1] = Q[3 * i + 1];
        V[3 * i + 2] = Q[3 * i + 0];
    }
    if (det3(V) < 0)
        for (int i = 0; i < 3; ++i) V[3 * i + 2] = -V[3 * i + 2];
    double B[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            B[3 * i + j] = F[3 * i] * V[j] + F[3 * i + 1] * V[3 + j] + F[3 * i + 2] * V[6 + j];
    double b0[3] = {B[0], B[3], B[6]}, b1[3] = {B[1], B[4], B[7]}, b2[3] = {B[2], B[5], B[8]};
    double u0[3], u1[3], u2[3];
    double n0 = sqrt(b0[0] * b0[0] + b0[1] * b0[1] + b0[2] * b0[2]);
    if (n0 > 0) {
        u0[0] = b0[0] / n0; u0[1] = b0[1] / n0; u0[2] = b0[2] / n0;
    } else {
        u0[0] = 1; u0[1] = 0; u0[2] = 0;
    }
    double d01 = u0[0] * b1[0] + u0[1] * b1[1] + u0[2] * b1[2];
    double r1[3] = {b1[0] - d01 * u0[0], b1[1] - d01 * u0[1], b1[2] - d01 * u0[2]};
    double n1 = sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
    if (n1 > 1e-14 * n0 && n1 > 0) {
        u1[0] = r1[0] / n1; u1[1] = r1[1] / n1; u1[2] = r1[2] / n1;
    } else {
        /* rank <= 1: any unit vector orthogonal to u0 */
        int k = (fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2])) ? 0
                : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
        double e[3] = {0, 0, 0};
        e[k] = 1;
        double d = u0[k];
        double t[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
        double nt = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        u1[0] = t[0] / nt; u1[1] = t[1] / nt; u1[2] = t[2] / nt;
    }
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
    u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
    u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    S[0] = n0;
    S[1] = u1[0] * b1[0] + u1[1] * b1[1] + u1[2] * b1[2];
    S[2] = u2[0] * b2[0] + u2[1] * b2[1] + u2[2] * b2[2];
    for (int i = 0; i < 3; ++i) {
        U[3 * i + 0] = u0[i];
        U[3 * i + 1] = u1[i];
        U[3 * i + 2] = u2[i];
    }
}

/* Optional replacement of the 3x3 SVD by a caller-supplied BATCH routine (n row-major matrices in, U, S, V out).
 * tests/ bind this to oracle/_ref/librefpin.so:ref_svd, i.e. to the reference's own AVX kernel
 * (Utils/SVD_EFTYCHIOS, staged as IglUtils.cpp:929-1085 does), so that the restated step can be run with exactly
 * the singular values the reference feeds its energies and -- through the cached `svd` (redoSVD = false,
 * DOTTimeStepper.cpp:580) -- its Hessian projection.  NULL (default) = dor_svd3. */
static dor_svd_batch_fn g_svd_batch = NULL;
void dor_set_svd_batch(dor_svd_batch_fn fn) { g_svd_batch = fn; }

/* ------------------------------------------------------------------------------------------
 * materials in singular-value space
 * ---------------------------------------------------------------------------------------- */
/* FixedCoRotEnergy.cpp:83-92 / SIMD_DOUBLE_MACROS.hpp:74-99;  StableNHEnergy.cpp:91-94 /
 * SIMD_DOUBLE_MACROS.hpp:133-155 */
double dor_psi(int mat, const double s[3], double mu, double lam)
{
    double J = s[0] * s[1] * s[2];
    if (mat == DOR_FCR) {
        double a = s[0] - 1, b = s[1] - 1, c = s[2] - 1;
        return mu * (a * a + b * b + c * c) + lam / 2.0 * (J - 1.0) * (J - 1.0);
    }
    double JmA = J - (1.0 + mu / lam);
    return (mu * (s[0] * s[0] + s[1] * s[1] + s[2] * s[2] - 3.0) + lam * JmA * JmA) / 2.0;
}

/* FixedCoRotEnergy.cpp:94-119, StableNHEnergy.cpp:116-126 */
void dor_dpsi(int mat, const double s[3], double mu, double lam, double d[3])
{
    double J = s[0] * s[1] * s[2];
    double pn[3] = {s[1] * s[2], s[2] * s[0], s[0] * s[1]};
    if (mat == DOR_FCR) {
        double t = lam * (J - 1.0);
        for (int i = 0; i < 3; ++i) d[i] = 2.0 * mu * (s[i] - 1.0) + pn[i] * t;
    } else {
        double t = lam * (J - (1.0 + mu / lam));
        for (int i = 0; i < 3; ++i) d[i] = s[i] * mu + t * pn[i];
    }
}

/* FixedCoRotEnergy.cpp:121-156, StableNHEnergy.cpp:173-196 */
void dor_d2psi(int mat, const double s[3], double mu, double lam, double A[9])
{
    double J = s[0] * s[1] * s[2];
    double pn[3] = {s[1] * s[2], s[2] * s[0], s[0] * s[1]};
    if (mat == DOR_FCR) {
        for (int i = 0; i < 3; ++i) A[4 * i] = 2.0 * mu + lam * pn[i] * pn[i];
        A[1] = A[3] = lam * (s[2] * (J - 1.0) + pn[0] * pn[1]);
        A[2] = A[6] = lam * (s[1] * (J - 1.0) + pn[0] * pn[2]);
        A[5] = A[7] = lam * (s[0] * (J - 1.0) + pn[2] * pn[1]);
    } else {
        double l2 = lam * (2.0 * J - (1.0 + mu / lam));
        for (int i = 0; i < 3; ++i) A[4 * i] = mu + lam * pn[i] * pn[i];
        A[1] = A[3] = s[2] * l2;
        A[2] = A[6] = s[1] * l2;
        A[5] = A[7] = s[0] * l2;
    }
}

/* FixedCoRotEnergy.cpp:158-172, StableNHEnergy.cpp:219-228; b[c] pairs (0,1),(1,2),(2,0) */
void dor_bleft(int mat, const double s[3], double mu, double lam, double b[3])
{
    double J = s[0] * s[1] * s[2];
    if (mat == DOR_FCR) {
        double h = lam / 2.0;
        b[0] = mu - h * s[2] * (J - 1.0);
        b[1] = mu - h * s[0] * (J - 1.0);
        b[2] = mu - h * s[1] * (J - 1.0);
    } else {
        double t = lam * (J - (1.0 + mu / lam));
        b[0] = (mu - t * s[2]) / 2.0;
        b[1] = (mu - t * s[0]) / 2.0;
        b[2] = (mu - t * s[1]) / 2.0;
    }
}

/* IglUtils.hpp:253-269: clamp negative eigenvalues to 0, only if the smallest is negative */
void dor_make_pd3(double A[9])
{
    double w[3], Q[9];
    sym_eig3(A, w, Q);
    if (w[0] >= 0.0) return;
    for (int i = 0; i < 3; ++i)
        if (w[i] < 0.0) w[i] = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            A[3 * i + j] = Q[3 * i] * w[0] * Q[3 * j] + Q[3 * i + 1] * w[1] * Q[3 * j + 1] +
                           Q[3 * i + 2] * w[2] * Q[3 * j + 2];
}

/* IglUtils.hpp:271-309 closed-form 2x2 PSD clamp */
void dor_make_pd2(double B[4])
{
    const double a = B[0], b = (B[1] + B[2]) / 2.0, d = B[3];
    double b2 = b * b;
    const double D = a * d - b2;
    const double T2 = (a + d) / 2.0;
    const double sq = sqrt(T2 * T2 - D);
    const double L2 = T2 - sq;
    if (L2 < 0.0) {
        const double L1 = T2 + sq;
        if (L1 <= 0.0) {
            B[0] = B[1] = B[2] = B[3] = 0.0;
        } else if (b2 == 0.0) {
            B[0] = L1;
            B[1] = B[2] = B[3] = 0.0;
        } else {
            const double L1md = L1 - d;
            const double r = L1md / L1;
            B[0] = r * L1md;
            B[1] = B[2] = b * r;
            B[3] = b2 / L1;
        }
    }
}

/* Energy.cpp:1129-1270: w * dP/dF (9x9, F vectorised row-major: ij = 3i+j) */
void dor_dPdF(int mat, const double U[9], const double S[3], const double V[9], double mu,
              double lam, double w, int project, double M[81])
{
    double dE[3], A[9], bl[3];
    dor_dpsi(mat, S, mu, lam, dE);
    dor_d2psi(mat, S, mu, lam, A);
    if (project) dor_make_pd3(A);
    dor_bleft(mat, S, mu, lam, bl);
    double B[3][4];
    for (int c = 0; c < 3; ++c) {
        int cp = (c + 1) % 3;
        double right = dE[c] + dE[cp];
        double sum = S[c] + S[cp];
        const double eps = 1.0e-6;
        if (sum < eps) right /= 2.0 * eps;
        else right /= 2.0 * sum;
        B[c][0] = B[c][3] = bl[c] + right;
        B[c][1] = B[c][2] = bl[c] - right;
        if (project) dor_make_pd2(B[c]);
    }
    /* Mh in the (a,b) singular basis, index 3a+b (Energy.cpp:1183-1207) */
    double Mh[81];
    memset(Mh, 0, sizeof(Mh));
#define MH(r, c) Mh[9 * (r) + (c)]
    MH(0, 0) = w * A[0]; MH(0, 4) = w * A[1]; MH(0, 8) = w * A[2];
    MH(4, 0) = w * A[3]; MH(4, 4) = w * A[4]; MH(4, 8) = w * A[5];
    MH(8, 0) = w * A[6]; MH(8, 4) = w * A[7]; MH(8, 8) = w * A[8];
    MH(1, 1) = w * B[0][0]; MH(1, 3) = w * B[0][1]; MH(3, 1) = w * B[0][2]; MH(3, 3) = w * B[0][3];
    MH(5, 5) = w * B[1][0]; MH(5, 7) = w * B[1][1]; MH(7, 5) = w * B[1][2]; MH(7, 7) = w * B[1][3];
    MH(2, 2) = w * B[2][3]; MH(2, 6) = w * B[2][2]; MH(6, 2) = w * B[2][1]; MH(6, 6) = w * B[2][0];
    /* dPdF(ij,rs) = sum Mh(ab,cd) U(i,a)V(j,b)U(r,c)V(s,d): two-sided transform with
     * K(ij,ab) = U(i,a) V(j,b) */
    double K[81];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) K[9 * (3 * i + j) + 3 * a + b] = U[3 * i + a] * V[3 * j + b];
    double T[81];
    for (int r = 0; r < 9; ++r)
        for (int c = 0; c < 9; ++c) {
            double acc = 0;
            for (int k = 0; k < 9; ++k) acc += K[9 * r + k] * MH(k, c);
            T[9 * r + c] = acc;
        }
    for (int r = 0; r < 9; ++r)
        for (int c = r; c < 9; ++c) {
            double acc = 0;
            for (int k = 0; k < 9; ++k) acc += T[9 * r + k] * K[9 * c + k];
            M[9 * r + c] = M[9 * c + r] = acc;
        }
#undef MH
}

/* IglUtils.cpp:836-870 (3-D branch): g = (dF/dx)^T : P */
void dor_dFdx_mult_vec(const double P[9], const double A[9], double g[12])
{
    for (int a = 0; a < 3; ++a)
        for (int c = 0; c < 3; ++c)
            g[3 + 3 * a + c] = A[3 * a] * P[3 * c] + A[3 * a + 1] * P[3 * c + 1] + A[3 * a + 2] * P[3 * c + 2];
    for (int c = 0; c < 3; ++c) g[c] = -g[3 + c] - g[6 + c] - g[9 + c];
}

static void deformation_gradient(const double x4[12], const double A[9], double F[9])
{
    /* Energy.cpp:309-321: Xt.col(k) = x_{k+1} - x_0 ; F = Xt * A */
    double Ds[9];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) Ds[3 * i + k] = x4[3 * (k + 1) + i] - x4[i];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            F[3 * i + j] = Ds[3 * i] * A[j] + Ds[3 * i + 1] * A[3 + j] + Ds[3 * i + 2] * A[6 + j];
}

/* Energy.cpp:738-777: H_e = G (w dP/dF) G^T via two dF_div_dx_mult passes (IglUtils.hpp:466-479) */
static void elem_hessian_from_M(const double M[81], const double A[9], double H[144])
{
    /* G row r (12), col k=3c+b (9): G[3+3a+c][3c+b] = A[a][b]; G[c][3c+b] = -sum_a A[a][b] */
    double GM[12 * 9];
    for (int col = 0; col < 9; ++col) {
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c)
                GM[9 * (3 + 3 * a + c) + col] = A[3 * a] * M[9 * (3 * c) + col] +
                                                A[3 * a + 1] * M[9 * (3 * c + 1) + col] +
                                                A[3 * a + 2] * M[9 * (3 * c + 2) + col];
        for (int c = 0; c < 3; ++c)
            GM[9 * c + col] = -GM[9 * (3 + c) + col] - GM[9 * (6 + c) + col] - GM[9 * (9 + c) + col];
    }
    for (int col = 0; col < 12; ++col) {
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 3; ++c)
                H[12 * (3 + 3 * a + c) + col] = A[3 * a] * GM[9 * col + 3 * c] +
                                                 A[3 * a + 1] * GM[9 * col + 3 * c + 1] +
                                                 A[3 * a + 2] * GM[9 * col + 3 * c + 2];
        for (int c = 0; c < 3; ++c)
            H[12 * c + col] = -H[12 * (3 + c) + col] - H[12 * (6 + c) + col] - H[12 * (9 + c) + col];
    }
}

static void elem_hessian_usv(int mat, const double U[9], const double S[3], const double V[9], const double A[9],
                             double mu, double lam, double w, int project, double H[144])
{
    double M[81];
    dor_dPdF(mat, U, S, V, mu, lam, w, project, M);
    elem_hessian_from_M(M, A, H);
}

void dor_elem_hessian_x(int mat, const double x4[12], const double A[9], double mu, double lam,
                        double w, int project, double H[144])
{
    double F[9], U[9], S[3], V[9];
    deformation_gradient(x4, A, F);
    dor_svd3(F, U, S, V);
    elem_hessian_usv(mat, U, S, V, A, mu, lam, w, project, H);
}

/* Energy.cpp:910-972 (SIMD path): P = U diag(dPsi/dsigma) V^T ; g_e = dF/dx^T : (w P) */
static void elem_energy_grad_usv(int mat, const double U[9], const double S[3], const double V[9], const double A[9],
                                 double mu, double lam, double w, double *psi_w, double g[12]);

void dor_elem_energy_grad_x(int mat, const double x4[12], const double A[9], double mu, double lam,
                            double w, double *psi_w, double g[12])
{
    double F[9], U[9], S[3], V[9];
    deformation_gradient(x4, A, F);
    dor_svd3(F, U, S, V);
    elem_energy_grad_usv(mat, U, S, V, A, mu, lam, w, psi_w, g);
}

static void elem_energy_grad_usv(int mat, const double U[9], const double S[3], const double V[9], const double A[9],
                                 double mu, double lam, double w, double *psi_w, double g[12])
{
    double d[3], P[9];
    if (psi_w) *psi_w = w * dor_psi(mat, S, mu, lam);
    if (g) {
        dor_dpsi(mat, S, mu, lam, d);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j)
                P[3 * i + j] = w * (U[3 * i] * d[0] * V[3 * j] + U[3 * i + 1] * d[1] * V[3 * j + 1] +
                                    U[3 * i + 2] * d[2] * V[3 * j + 2]);
        dor_dFdx_mult_vec(P, A, g);
    }
}

/* ------------------------------------------------------------------------------------------
 * simulation object
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int nv;        /* local vertex count */
    int *l2g;      /* ascending global ids */
    int *pos;      /* local -> position in the RCM order */
    int *ord;      /* position -> local */
    long *rowptr;  /* scalar row start offsets in L (n+1) */
    int *first;    /* scalar row first column */
    double *L;     /* envelope storage */
    /* external solver (dor_use_ext_solver): its object, and the local block pattern handed to it */
    void *ext;
    int *nbr_ptr, *nbr_idx;
    size_t *nbr_src; /* per pattern entry: block index in Hval */
} dor_part;

#define HIST 5 /* DOTTimeStepper.cpp:45 historySize */

struct dor_sim {
    int nV, nT, mat, nParts;
    double dt, dtSq, gravity[3], relTol, targetGRes;
    double alphaMin; /* lower clamp of the initial step length: 0.1 for DOT (Optimizer.cpp:1085); 1 = the fixed unit step of
                        the other steppers (initStepSize's else branch, :1088) */
    int *T;
    double *Xrest, *A, *vol, *mu, *lam, *mass;
    unsigned char *fixed;
    int *vf_ptr, *vf_elem, *vf_slot;       /* vFLoc (Mesh.cpp:609-614), sorted (elem,slot) */
    int *adj_ptr, *adj_idx;                /* vNeighbor + self, ascending */
    int *eblk;                             /* nT*16: block index of (T[e][a], T[e][b]) */
    double *Hval;                          /* nnzb*9 */
    double *He;                            /* nT*144 */
    int *epart, *dup;
    int *vpart; /* optional vertex partition: subdomains = vertex sets (block Jacobi) */
    dor_part *parts;
    dor_ext_solver ext; /* ext.create != NULL: the subdomain systems go through it */
    /* state */
    double *x, *xn, *v, *xt, *g, *p;
    /* L-BFGS */
    double *hs[HIST], *hy[HIST], hys[HIST];
    int nh;
    /* scratch */
    double *ework, *gcont, *x0, *q, *gold, *Hp, *tmp_s, *tmp_y;
    double *usv; /* per element U[9] S[3] V[9] of the last evaluation (svd_all) */
    /* logs */
    int log_n, log_cap;
    double *log_alpha, *log_E, *log_g2;
    long numLineSearch, ls0;
    /* running step (dor_step_begin / _iterate / _end) */
    int it, failed;
    double lastE, g2, E0, g2_0;
    double t_energy, t_grad, t_solve, t_hess, t_factor;
    int factor_failed;  /* sticky: some dor_refactor since creation (or since dor_use_ext_solver) met a non-SPD subdomain */
    int energy_evals;
};

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

/* Mesh.cpp:589-700 computeFeatures + :552-585 lumped mass + :741-744 Lame parameters */
static void build_features(dor_sim *s, double YM, double PR, double rho)
{
    int nV = s->nV, nT = s->nT;
    for (int v = 0; v < nV; ++v) s->mass[v] = 0;
    for (int e = 0; e < nT; ++e) {
        const int *t = s->T + 4 * e;
        const double *p0 = s->Xrest + 3 * t[0], *p1 = s->Xrest + 3 * t[1], *p2 = s->Xrest + 3 * t[2],
                     *p3 = s->Xrest + 3 * t[3];
        double X0[9];
        for (int i = 0; i < 3; ++i) {
            X0[3 * i + 0] = p1[i] - p0[i];
            X0[3 * i + 1] = p2[i] - p0[i];
            X0[3 * i + 2] = p3[i] - p0[i];
        }
        inv3(X0, s->A + 9 * e);
        s->vol[e] = det3(X0) / 3.0 / 2.0; /* signed, Mesh.cpp:639 */
        /* mass: |det[v0-v3, v1-v3, v2-v3]|/6/4 per corner (Mesh.cpp:565-577) */
        double a[3], b[3], c[3];
        for (int i = 0; i < 3; ++i) {
            a[i] = p0[i] - p3[i];
            b[i] = p1[i] - p3[i];
            c[i] = p2[i] - p3[i];
        }
        double vv = fabs(a[0] * (b[1] * c[2] - b[2] * c[1]) + a[1] * (b[2] * c[0] - b[0] * c[2]) +
                         a[2] * (b[0] * c[1] - b[1] * c[0])) / 6.0;
        for (int k = 0; k < 4; ++k) s->mass[t[k]] += vv / 4.0;
        s->mu[e] = YM / 2.0 / (1.0 + PR);
        s->lam[e] = YM * PR / (1.0 + PR) / (1.0 - 2.0 * PR);
    }
    for (int v = 0; v < nV; ++v) s->mass[v] *= rho;
}

static void build_topology(dor_sim *s)
{
    int nV = s->nV, nT = s->nT;
    /* vFLoc */
    s->vf_ptr = calloc(nV + 1, sizeof(int));
    for (int e = 0; e < nT; ++e)
        for (int k = 0; k < 4; ++k) s->vf_ptr[s->T[4 * e + k] + 1]++;
    for (int v = 0; v < nV; ++v) s->vf_ptr[v + 1] += s->vf_ptr[v];
    s->vf_elem = malloc(sizeof(int) * 4 * nT);
    s->vf_slot = malloc(sizeof(int) * 4 * nT);
    int *cur = malloc(sizeof(int) * nV);
    memcpy(cur, s->vf_ptr, sizeof(int) * nV);
    for (int e = 0; e < nT; ++e)
        for (int k = 0; k < 4; ++k) {
            int v = s->T[4 * e + k];
            s->vf_elem[cur[v]] = e;
            s->vf_slot[cur[v]] = k;
            cur[v]++;
        }
    /* adjacency with self */
    int *cnt = calloc(nV + 1, sizeof(int));
    int *tmp = malloc(sizeof(int) * (16 * (size_t)nT + nV));
    int *tptr = calloc(nV + 1, sizeof(int));
    for (int e = 0; e < nT; ++e)
        for (int a = 0; a < 4; ++a) tptr[s->T[4 * e + a] + 1] += 4;
    for (int v = 0; v < nV; ++v) tptr[v + 1] += tptr[v];
    memset(cur, 0, sizeof(int) * nV);
    for (int e = 0; e < nT; ++e)
        for (int a = 0; a < 4; ++a) {
            int v = s->T[4 * e + a];
            for (int b = 0; b < 4; ++b) tmp[tptr[v] + cur[v]++] = s->T[4 * e + b];
        }
    s->adj_ptr = calloc(nV + 1, sizeof(int));
    for (int v = 0; v < nV; ++v) {
        int n = tptr[v + 1] - tptr[v];
        int *l = tmp + tptr[v];
        qsort(l, n, sizeof(int), cmp_int);
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (i == 0 || l[i] != l[i - 1]) l[m++] = l[i];
        cnt[v] = m;
        s->adj_ptr[v + 1] = s->adj_ptr[v] + m;
    }
    s->adj_idx = malloc(sizeof(int) * s->adj_ptr[nV]);
    for (int v = 0; v < nV; ++v) memcpy(s->adj_idx + s->adj_ptr[v], tmp + tptr[v], sizeof(int) * cnt[v]);
    free(tmp);
    free(tptr);
    free(cnt);
    free(cur);
    /* element block slots */
    s->eblk = malloc(sizeof(int) * 16 * (size_t)nT);
    for (int e = 0; e < nT; ++e)
        for (int a = 0; a < 4; ++a) {
            int v = s->T[4 * e + a];
            for (int b = 0; b < 4; ++b) {
                int u = s->T[4 * e + b];
                int lo = s->adj_ptr[v], hi = s->adj_ptr[v + 1] - 1;
                while (lo < hi) {
                    int mid = (lo + hi) / 2;
                    if (s->adj_idx[mid] < u) lo = mid + 1;
                    else hi = mid;
                }
                s->eblk[16 * e + 4 * a + b] = lo;
            }
        }
    s->Hval = calloc((size_t)s->adj_ptr[nV] * 9, sizeof(double));
}

static int find_block(const dor_sim *s, int v, int u)
{
    int lo = s->adj_ptr[v], hi = s->adj_ptr[v + 1] - 1;
    while (lo < hi) {
        int mid = (lo + hi) / 2;
        if (s->adj_idx[mid] < u) lo = mid + 1;
        else hi = mid;
    }
    return (s->adj_idx[lo] == u) ? lo : -1;
}

/* ADMMDDTimeStepper.cpp:88-262 (partition -> local index maps), DOTTimeStepper.cpp:47-56 (dup).
 * Local factor: envelope Cholesky after a reverse Cuthill-McKee ordering (stands in for CHOLMOD's
 * supernodal LL^T, CHOLMODSolver.cpp:136-163: any exact SPD solve gives the same p_s up to
 * rounding). */
static void build_parts(dor_sim *s)
{
    int nV = s->nV, nT = s->nT, nP = s->nParts;
    s->parts = calloc(nP, sizeof(dor_part));
    s->dup = calloc(nV, sizeof(int));
    int *mark = malloc(sizeof(int) * nV);
    int *g2l = malloc(sizeof(int) * nV);
    for (int pI = 0; pI < nP; ++pI) {
        dor_part *P = &s->parts[pI];
        memset(mark, 0, sizeof(int) * nV);
        int n = 0;
        if (s->vpart) {   /* vertex partition (LBFGS-JH, LBFGSTimeStepper.cpp:70-90): disjoint vertex sets */
            for (int v = 0; v < nV; ++v)
                if (s->vpart[v] == pI) {
                    mark[v] = 1;
                    n++;
                }
        } else
        for (int e = 0; e < nT; ++e)
            if (s->epart[e] == pI)
                for (int k = 0; k < 4; ++k) {
                    int v = s->T[4 * e + k];
                    if (!mark[v]) {
                        mark[v] = 1;
                        n++;
                    }
                }
        P->nv = n;
        P->l2g = malloc(sizeof(int) * (n > 0 ? n : 1));
        int m = 0;
        for (int v = 0; v < nV; ++v) {
            g2l[v] = -1;
            if (mark[v]) {
                g2l[v] = m;
                P->l2g[m++] = v;
                s->dup[v]++;
            }
        }
        /* local graph degrees */
        int *deg = calloc(n > 0 ? n : 1, sizeof(int));
        for (int i = 0; i < n; ++i) {
            int v = P->l2g[i];
            for (int k = s->adj_ptr[v]; k < s->adj_ptr[v + 1]; ++k) {
                int u = s->adj_idx[k];
                if (u != v && g2l[u] >= 0) deg[i]++;
            }
        }
        /* RCM */
        P->ord = malloc(sizeof(int) * (n > 0 ? n : 1));
        P->pos = malloc(sizeof(int) * (n > 0 ? n : 1));
        int *visited = calloc(n > 0 ? n : 1, sizeof(int));
        int *queue = malloc(sizeof(int) * (n > 0 ? n : 1));
        int *nb = malloc(sizeof(int) * 512);
        int count = 0;
        while (count < n) {
            /* start: unvisited node of minimum degree, refined by two BFS passes */
            int start = -1;
            for (int i = 0; i < n; ++i)
                if (!visited[i] && (start < 0 || deg[i] < deg[start])) start = i;
            for (int pass = 0; pass < 2; ++pass) {
                int *lvl = calloc(n, sizeof(int));
                int qh = 0, qt = 0;
                queue[qt++] = start;
                lvl[start] = 1;
                int last = start;
                while (qh < qt) {
                    int i = queue[qh++];
                    last = i;
                    int v = P->l2g[i];
                    for (int k = s->adj_ptr[v]; k < s->adj_ptr[v + 1]; ++k) {
                        int j = g2l[s->adj_idx[k]];
                        if (j >= 0 && !visited[j] && !lvl[j]) {
                            lvl[j] = lvl[i] + 1;
                            queue[qt++] = j;
                        }
                    }
                }
                /* among the last level pick the min degree */
                int best = last;
                for (int k2 = 0; k2 < qt; ++k2)
                    if (lvl[queue[k2]] == lvl[last] && deg[queue[k2]] < deg[best]) best = queue[k2];
                start = best;
                free(lvl);
            }
            int qh = count, qt = count;
            P->ord[qt++] = start;
            visited[start] = 1;
            while (qh < qt) {
                int i = P->ord[qh++];
                int v = P->l2g[i];
                int nn = 0;
                for (int k = s->adj_ptr[v]; k < s->adj_ptr[v + 1]; ++k) {
                    int j = g2l[s->adj_idx[k]];
                    if (j >= 0 && !visited[j]) {
                        visited[j] = 1;
                        if (nn < 512) nb[nn++] = j;
                    }
                }
                /* ascending degree (insertion sort) */
                for (int a = 1; a < nn; ++a) {
                    int t = nb[a], b = a - 1;
                    while (b >= 0 && deg[nb[b]] > deg[t]) {
                        nb[b + 1] = nb[b];
                        b--;
                    }
                    nb[b + 1] = t;
                }
                for (int a = 0; a < nn; ++a) P->ord[qt++] = nb[a];
            }
            count = qt;
        }
        /* reverse */
        for (int i = 0; i < n / 2; ++i) {
            int t = P->ord[i];
            P->ord[i] = P->ord[n - 1 - i];
            P->ord[n - 1 - i] = t;
        }
        for (int i = 0; i < n; ++i) P->pos[P->ord[i]] = i;
        /* envelope */
        int N = 3 * n;
        P->first = malloc(sizeof(int) * (N > 0 ? N : 1));
        P->rowptr = malloc(sizeof(long) * (N + 1));
        P->rowptr[0] = 0;
        for (int pi = 0; pi < n; ++pi) {
            int i = P->ord[pi];
            int v = P->l2g[i];
            int f = pi;
            for (int k = s->adj_ptr[v]; k < s->adj_ptr[v + 1]; ++k) {
                int j = g2l[s->adj_idx[k]];
                if (j >= 0 && P->pos[j] < f) f = P->pos[j];
            }
            for (int r = 0; r < 3; ++r) {
                int row = 3 * pi + r;
                P->first[row] = 3 * f;
                P->rowptr[row + 1] = P->rowptr[row] + (row - 3 * f + 1);
            }
        }
        P->L = malloc(sizeof(double) * (size_t)(P->rowptr[N] > 0 ? P->rowptr[N] : 1));
        free(deg);
        free(visited);
        free(queue);
        free(nb);
    }
    free(mark);
    free(g2l);
}

/* fill the envelope of part P with R_s H R_s^T (DOTTimeStepper.cpp:619-797 yields exactly the
 * principal sub-matrix of the global projected Hessian on free DOFs -- SURVEY.md section 0 fact 1)
 * and factor in place. returns 0 on success */
static int factor_part(const dor_sim *s, dor_part *P)
{
    int n = P->nv, N = 3 * n;
    memset(P->L, 0, sizeof(double) * (size_t)P->rowptr[N]);
    for (int pi = 0; pi < n; ++pi) {
        int i = P->ord[pi];
        int v = P->l2g[i];
        /* neighbours inside the part: walk both sorted lists */
        int a = 0;
        for (int k = s->adj_ptr[v]; k < s->adj_ptr[v + 1]; ++k) {
            int u = s->adj_idx[k];
            while (a < n && P->l2g[a] < u) a++;
            if (a >= n) break;
            if (P->l2g[a] != u) continue;
            int pj = P->pos[a];
            if (pj > pi) continue;
            const double *blk = s->Hval + 9 * (size_t)k;
            for (int r = 0; r < 3; ++r) {
                int row = 3 * pi + r;
                double *Lr = P->L + P->rowptr[row] - P->first[row];
                for (int c = 0; c < 3; ++c) {
                    int col = 3 * pj + c;
                    if (col <= row) Lr[col] = blk[3 * r + c];
                }
            }
        }
    }
    for (int i = 0; i < N; ++i) {
        double *Li = P->L + P->rowptr[i] - P->first[i];
        int fi = P->first[i];
        for (int j = fi; j < i; ++j) {
            const double *Lj = P->L + P->rowptr[j] - P->first[j];
            int k0 = fi > P->first[j] ? fi : P->first[j];
            double acc = Li[j];
            for (int k = k0; k < j; ++k) acc -= Li[k] * Lj[k];
            Li[j] = acc / Lj[j];
        }
        double d = Li[i];
        for (int k = fi; k < i; ++k) d -= Li[k] * Li[k];
        if (!(d > 0.0)) return -1;
        Li[i] = sqrt(d);
    }
    return 0;
}

static void solve_part(const dor_part *P, double *b /* in RCM scalar order, in/out */)
{
    int N = 3 * P->nv;
    for (int i = 0; i < N; ++i) {
        const double *Li = P->L + P->rowptr[i] - P->first[i];
        double acc = b[i];
        for (int k = P->first[i]; k < i; ++k) acc -= Li[k] * b[k];
        b[i] = acc / Li[i];
    }
    for (int i = N - 1; i >= 0; --i) {
        const double *Li = P->L + P->rowptr[i] - P->first[i];
        double xi = b[i] / Li[i];
        b[i] = xi;
        for (int k = P->first[i]; k < i; ++k) b[k] -= Li[k] * xi;
    }
}

/* ---- external subdomain solver (dor_use_ext_solver) ---- */
static void ext_release(dor_sim *s)
{
    for (int pI = 0; pI < s->nParts; ++pI) {
        dor_part *P = &s->parts[pI];
        if (P->ext && s->ext.destroy) s->ext.destroy(P->ext);
        P->ext = NULL;
        free(P->nbr_ptr); free(P->nbr_idx); free(P->nbr_src);
        P->nbr_ptr = P->nbr_idx = NULL;
        P->nbr_src = NULL;
    }
}

/* the block pattern of H_s = R_s H R_s^T in local ids (both sorted lists walked as in factor_part) + one object per part */
static void ext_build(dor_sim *s)
{
    for (int pI = 0; pI < s->nParts; ++pI) {
        dor_part *P = &s->parts[pI];
        int n = P->nv, cnt = 0;
        P->nbr_ptr = (int *)malloc(sizeof(int) * (n + 1));
        for (int pass = 0; pass < 2; ++pass) {
            cnt = 0;
            for (int i = 0; i < n; ++i) {
                int v = P->l2g[i], a = 0;
                if (pass == 0) P->nbr_ptr[i] = cnt;
                for (int k = s->adj_ptr[v]; k < s->adj_ptr[v + 1]; ++k) {
                    int u = s->adj_idx[k];
                    while (a < n && P->l2g[a] < u) a++;
                    if (a >= n) break;
                    if (P->l2g[a] != u) continue;
                    if (pass == 1) {
                        P->nbr_idx[cnt] = a;
                        P->nbr_src[cnt] = (size_t)k;
                    }
                    cnt++;
                }
            }
            if (pass == 0) {
                P->nbr_ptr[n] = cnt;
                P->nbr_idx = (int *)malloc(sizeof(int) * (cnt > 0 ? cnt : 1));
                P->nbr_src = (size_t *)malloc(sizeof(size_t) * (cnt > 0 ? cnt : 1));
            }
        }
        unsigned char *fx = (unsigned char *)malloc(n > 0 ? n : 1);
        for (int i = 0; i < n; ++i) fx[i] = s->fixed[P->l2g[i]];
        P->ext = s->ext.create(n, P->nbr_ptr, P->nbr_idx, fx);
        free(fx);
    }
}

static int ext_factor_part(const dor_sim *s, dor_part *P)
{
    int cnt = P->nbr_ptr[P->nv];
    double *blk = (double *)malloc(sizeof(double) * 9 * (size_t)(cnt > 0 ? cnt : 1));
    for (int e = 0; e < cnt; ++e) memcpy(blk + 9 * (size_t)e, s->Hval + 9 * P->nbr_src[e], sizeof(double) * 9);
    int rc = s->ext.factor(P->ext, blk);
    free(blk);
    return rc;
}

/* b in RCM scalar order, in / out, whichever solver holds the factors */
static void solve_part_any(const dor_sim *s, int pI, double *b)
{
    const dor_part *P = &s->parts[pI];
    if (!s->ext.create) {
        solve_part(P, b);
        return;
    }
    double *loc = (double *)malloc(sizeof(double) * 3 * (P->nv > 0 ? P->nv : 1));
    for (int i = 0; i < P->nv; ++i)
        for (int d = 0; d < 3; ++d) loc[3 * i + d] = b[3 * P->pos[i] + d];
    s->ext.solve(P->ext, loc);
    for (int i = 0; i < P->nv; ++i)
        for (int d = 0; d < 3; ++d) b[3 * P->pos[i] + d] = loc[3 * i + d];
    free(loc);
}

/* Optimizer.cpp:613-651 computeCharNormSq, :1045 updateTargetGRes, :222-228 setRelGL2Tol */
static double char_norm_sq(const dor_sim *s, double epsSq)
{
    double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, S1[3] = {1, 1, 1}, M[81];
    dor_dPdF(s->mat, I, S1, I, s->mu[0], s->lam[0], 1.0, 0, M);
    double sqH = 0;
    for (int i = 0; i < 81; ++i) sqH += M[i] * M[i];
    double *ls = calloc(s->nV, sizeof(double));
    for (int e = 0; e < s->nT; ++e) {
        const int *t = s->T + 4 * e;
        for (int i = 0; i < 4; ++i) {
            /* area of the face opposite vertex i (igl::face_areas, face_areas.cpp:41-62) */
            const double *a = s->Xrest + 3 * t[(i + 1) % 4], *b = s->Xrest + 3 * t[(i + 2) % 4],
                         *c = s->Xrest + 3 * t[(i + 3) % 4];
            double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
            double w[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
            double cx = u[1] * w[2] - u[2] * w[1], cy = u[2] * w[0] - u[0] * w[2],
                   cz = u[0] * w[1] - u[1] * w[0];
            ls[t[i]] += 0.5 * sqrt(cx * cx + cy * cy + cz * cz);
        }
    }
    double sql = 0;
    for (int v = 0; v < s->nV; ++v) sql += ls[v] * ls[v];
    free(ls);
    /* data0 has exactly one fixed vertex (Mesh.cpp:592-598) */
    double cn = epsSq * sqH * sql * (double)(s->nV - 1) / (double)s->nV;
    return cn * s->dtSq * s->dtSq;
}

static void compute_xtilde(dor_sim *s)
{
    /* Optimizer.cpp:585-610 */
    for (int v = 0; v < s->nV; ++v)
        for (int d = 0; d < 3; ++d) {
            if (s->fixed[v]) s->xt[3 * v + d] = s->xn[3 * v + d];
            else s->xt[3 * v + d] = s->xn[3 * v + d] + (s->v[3 * v + d] * s->dt + s->dtSq * s->gravity[d]);
        }
}

/* F, U, S, V of every element at x (Energy.cpp:294-423: F = Ds A, then the batch SVD).  21 doubles per element in
 * s->usv: U[9] S[3] V[9].  With the batch hook the matrices go through it in slices of at most 65536 (the size the
 * reference instantiates its helper for); without it every element runs dor_svd3. */
static void svd_all(dor_sim *s, const double *x)
{
    int nT = s->nT;
    if (!s->usv) s->usv = (double *)malloc(sizeof(double) * 21 * (size_t)nT);
    if (!g_svd_batch) {
#pragma omp parallel for schedule(static)
        for (int e = 0; e < nT; ++e) {
            const int *t = s->T + 4 * e;
            double x4[12], F[9];
            for (int k = 0; k < 4; ++k)
                for (int d = 0; d < 3; ++d) x4[3 * k + d] = x[3 * t[k] + d];
            deformation_gradient(x4, s->A + 9 * e, F);
            double *o = s->usv + 21 * (size_t)e;
            dor_svd3(F, o, o + 9, o + 12);
        }
        return;
    }
    const int CH = 65536;
    double *F = (double *)malloc(sizeof(double) * 9 * CH), *U = (double *)malloc(sizeof(double) * 9 * CH),
           *S = (double *)malloc(sizeof(double) * 3 * CH), *V = (double *)malloc(sizeof(double) * 9 * CH);
    for (int e0 = 0; e0 < nT; e0 += CH) {
        int n = nT - e0 < CH ? nT - e0 : CH;
#pragma omp parallel for schedule(static)
        for (int i = 0; i < n; ++i) {
            const int *t = s->T + 4 * (e0 + i);
            double x4[12];
            for (int k = 0; k < 4; ++k)
                for (int d = 0; d < 3; ++d) x4[3 * k + d] = x[3 * t[k] + d];
            deformation_gradient(x4, s->A + 9 * (size_t)(e0 + i), F + 9 * (size_t)i);
        }
        g_svd_batch(n, F, U, S, V);
        for (int i = 0; i < n; ++i) {
            double *o = s->usv + 21 * (size_t)(e0 + i);
            memcpy(o, U + 9 * (size_t)i, 72);
            memcpy(o + 9, S + 3 * (size_t)i, 24);
            memcpy(o + 12, V + 9 * (size_t)i, 72);
        }
    }
    free(F); free(U); free(S); free(V);
}

/* Optimizer.cpp:1183-1218 + Energy.cpp:426-437, :852-907 */
double dor_eval_energy(dor_sim *s, const double *x)
{
    double t0 = now_ms();
    int nT = s->nT, nV = s->nV;
    svd_all(s, x);
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nT; ++e)
        s->ework[e] = dor_psi(s->mat, s->usv + 21 * (size_t)e + 9, s->mu[e], s->lam[e]) * s->vol[e];
    double sum = 0;
    for (int e = 0; e < nT; ++e) sum += s->ework[e];
    double E = s->dtSq * sum;
    double si = 0;
    for (int v = 0; v < nV; ++v) {
        double dx = x[3 * v] - s->xt[3 * v], dy = x[3 * v + 1] - s->xt[3 * v + 1],
               dz = x[3 * v + 2] - s->xt[3 * v + 2];
        si += (dx * dx + dy * dy + dz * dz) * s->mass[v] / 2.0;
    }
    s->t_energy += now_ms() - t0;
    s->energy_evals++;
    return E + si;
}

/* Optimizer.cpp:1220-1255 + Energy.cpp:441-564, :910-972 */
void dor_eval_gradient(dor_sim *s, const double *x, double *g)
{
    double t0 = now_ms();
    int nT = s->nT, nV = s->nV;
    svd_all(s, x);
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nT; ++e) {
        const double *o = s->usv + 21 * (size_t)e;
        elem_energy_grad_usv(s->mat, o, o + 9, o + 12, s->A + 9 * e, s->mu[e], s->lam[e], s->dtSq * s->vol[e], NULL,
                             s->gcont + 12 * (size_t)e);
    }
#pragma omp parallel for schedule(static)
    for (int v = 0; v < nV; ++v) {
        double a[3] = {0, 0, 0};
        if (!s->fixed[v]) {
            for (int k = s->vf_ptr[v]; k < s->vf_ptr[v + 1]; ++k) {
                const double *ge = s->gcont + 12 * (size_t)s->vf_elem[k] + 3 * s->vf_slot[k];
                a[0] += ge[0];
                a[1] += ge[1];
                a[2] += ge[2];
            }
            for (int d = 0; d < 3; ++d) a[d] += s->mass[v] * (x[3 * v + d] - s->xt[3 * v + d]);
        }
        g[3 * v] = a[0];
        g[3 * v + 1] = a[1];
        g[3 * v + 2] = a[2];
    }
    s->t_grad += now_ms() - t0;
}

/* Energy.cpp:673-701 */
void dor_eval_elem_hessians(dor_sim *s, const double *x, double *H)
{
    int nT = s->nT;
    svd_all(s, x);
#pragma omp parallel for schedule(static)
    for (int e = 0; e < nT; ++e) {
        const double *o = s->usv + 21 * (size_t)e;
        elem_hessian_usv(s->mat, o, o + 9, o + 12, s->A + 9 * e, s->mu[e], s->lam[e], s->dtSq * s->vol[e], 1,
                         H + 144 * (size_t)e);
    }
}

/* DOTTimeStepper.cpp:574-616 (global assembly, IglUtils.hpp:143-220) + :349-380 (factor) */
void dor_refactor(dor_sim *s, const double *x)
{
    double t0 = now_ms();
    int nV = s->nV;
    dor_eval_elem_hessians(s, x, s->He);
#pragma omp parallel for schedule(static)
    for (int v = 0; v < nV; ++v) {
        for (int k = s->adj_ptr[v]; k < s->adj_ptr[v + 1]; ++k)
            for (int i = 0; i < 9; ++i) s->Hval[9 * (size_t)k + i] = 0.0;
        if (s->fixed[v]) {
            int k = find_block(s, v, v);
            s->Hval[9 * (size_t)k] = s->Hval[9 * (size_t)k + 4] = s->Hval[9 * (size_t)k + 8] = 1.0;
            continue;
        }
        for (int k = s->vf_ptr[v]; k < s->vf_ptr[v + 1]; ++k) {
            int e = s->vf_elem[k], a = s->vf_slot[k];
            const double *He = s->He + 144 * (size_t)e;
            for (int b = 0; b < 4; ++b) {
                if (s->fixed[s->T[4 * e + b]]) continue;
                double *blk = s->Hval + 9 * (size_t)s->eblk[16 * e + 4 * a + b];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) blk[3 * r + c] += He[12 * (3 * a + r) + 3 * b + c];
            }
        }
        int kd = find_block(s, v, v);
        s->Hval[9 * (size_t)kd] += s->mass[v];
        s->Hval[9 * (size_t)kd + 4] += s->mass[v];
        s->Hval[9 * (size_t)kd + 8] += s->mass[v];
    }
    double t1 = now_ms();
    s->t_hess += t1 - t0;
    int fail = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int pI = 0; pI < s->nParts; ++pI)
        if (s->ext.create ? ext_factor_part(s, &s->parts[pI]) : factor_part(s, &s->parts[pI])) fail = 1;
    if (fail) {
        fprintf(stderr, "dot_oracle: subdomain factorisation failed (non-SPD)\n");
        s->factor_failed = 1;
    }
    s->t_factor += now_ms() - t1;
}

/* DOTTimeStepper.cpp:406-450: p = D^-1 sum_s R_s^T (R_s H R_s^T)^-1 R_s r */
void dor_apply_precond(dor_sim *s, const double *r, double *p)
{
    double t0 = now_ms();
    int nP = s->nParts;
    double **ps = malloc(sizeof(double *) * nP);
#pragma omp parallel for schedule(dynamic, 1)
    for (int pI = 0; pI < nP; ++pI) {
        const dor_part *P = &s->parts[pI];
        double *b = malloc(sizeof(double) * 3 * (P->nv > 0 ? P->nv : 1));
        for (int pi = 0; pi < P->nv; ++pi) {
            int v = P->l2g[P->ord[pi]];
            b[3 * pi] = r[3 * v];
            b[3 * pi + 1] = r[3 * v + 1];
            b[3 * pi + 2] = r[3 * v + 2];
        }
        solve_part_any(s, pI, b);
        ps[pI] = b;
    }
    memset(p, 0, sizeof(double) * 3 * s->nV);
    for (int pI = 0; pI < nP; ++pI) {
        const dor_part *P = &s->parts[pI];
        for (int i = 0; i < P->nv; ++i) {
            int v = P->l2g[i], pi = P->pos[i];
            p[3 * v] += ps[pI][3 * pi];
            p[3 * v + 1] += ps[pI][3 * pi + 1];
            p[3 * v + 2] += ps[pI][3 * pi + 2];
        }
        free(ps[pI]);
    }
    free(ps);
    for (int v = 0; v < s->nV; ++v)
        if (s->dup[v] > 1) {
            p[3 * v] /= s->dup[v];
            p[3 * v + 1] /= s->dup[v];
            p[3 * v + 2] /= s->dup[v];
        }
    s->t_solve += now_ms() - t0;
}

/* CHOLMODSolver.cpp:185-208 (cholmod_sdmult on the symmetric global matrix) */
void dor_spmv(dor_sim *s, const double *p, double *Hp)
{
    int nV = s->nV;
#pragma omp parallel for schedule(static)
    for (int v = 0; v < nV; ++v) {
        double a[3] = {0, 0, 0};
        for (int k = s->adj_ptr[v]; k < s->adj_ptr[v + 1]; ++k) {
            const double *b = s->Hval + 9 * (size_t)k;
            const double *pu = p + 3 * s->adj_idx[k];
            a[0] += b[0] * pu[0] + b[1] * pu[1] + b[2] * pu[2];
            a[1] += b[3] * pu[0] + b[4] * pu[1] + b[5] * pu[2];
            a[2] += b[6] * pu[0] + b[7] * pu[1] + b[8] * pu[2];
        }
        Hp[3 * v] = a[0];
        Hp[3 * v + 1] = a[1];
        Hp[3 * v + 2] = a[2];
    }
}

void dor_part_dense(const dor_sim *s, int part, double *Hs)
{
    const dor_part *P = &s->parts[part];
    int n = P->nv, N = 3 * n;
    memset(Hs, 0, sizeof(double) * (size_t)N * N);
    for (int i = 0; i < n; ++i) {
        int v = P->l2g[i];
        int a = 0;
        for (int k = s->adj_ptr[v]; k < s->adj_ptr[v + 1]; ++k) {
            int u = s->adj_idx[k];
            while (a < n && P->l2g[a] < u) a++;
            if (a >= n) break;
            if (P->l2g[a] != u) continue;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c)
                    Hs[(size_t)(3 * i + r) * N + 3 * a + c] = s->Hval[9 * (size_t)k + 3 * r + c];
        }
    }
}

static double dotn(const double *a, const double *b, int n)
{
    double acc = 0;
    for (int i = 0; i < n; ++i) acc += a[i] * b[i];
    return acc;
}

dor_sim *dor_create(int nV, int nT, const double *Xrest, const int *T, double YM, double PR,
                    double rho, int material, double dt, int withGravity,
                    const unsigned char *fixed, const double *x_init, const int *epart,
                    int nParts, double relTol)
{
    return dor_create_v(nV, nT, Xrest, T, YM, PR, rho, material, dt, withGravity, fixed, x_init, epart, NULL, nParts,
                        relTol);
}

/* vpart != NULL: the subdomains are the vertex sets of a VERTEX partition (LBFGS-JH), epart is ignored */
dor_sim *dor_create_v(int nV, int nT, const double *Xrest, const int *T, double YM, double PR,
                      double rho, int material, double dt, int withGravity,
                      const unsigned char *fixed, const double *x_init, const int *epart, const int *vpart,
                      int nParts, double relTol)
{
    dor_sim *s = calloc(1, sizeof(dor_sim));
    s->nV = nV;
    s->nT = nT;
    s->mat = material;
    s->nParts = nParts;
    s->dt = dt;
    s->dtSq = dt * dt;
    s->gravity[1] = withGravity ? -9.80665 : 0.0; /* Optimizer.cpp:107-110 */
    s->relTol = relTol;
    s->alphaMin = 0.1;
    int n = 3 * nV;
    s->T = malloc(sizeof(int) * 4 * (size_t)nT);
    memcpy(s->T, T, sizeof(int) * 4 * (size_t)nT);
    s->Xrest = malloc(sizeof(double) * n);
    memcpy(s->Xrest, Xrest, sizeof(double) * n);
    s->A = malloc(sizeof(double) * 9 * (size_t)nT);
    s->vol = malloc(sizeof(double) * nT);
    s->mu = malloc(sizeof(double) * nT);
    s->lam = malloc(sizeof(double) * nT);
    s->mass = malloc(sizeof(double) * nV);
    s->fixed = malloc(nV);
    memcpy(s->fixed, fixed, nV);
    s->epart = calloc(nT, sizeof(int));
    if (epart) memcpy(s->epart, epart, sizeof(int) * nT);
    if (vpart) {
        s->vpart = malloc(sizeof(int) * nV);
        memcpy(s->vpart, vpart, sizeof(int) * nV);
    }
    build_features(s, YM, PR, rho);
    build_topology(s);
    build_parts(s);
    s->He = malloc(sizeof(double) * 144 * (size_t)nT);
    s->x = malloc(sizeof(double) * n);
    s->xn = malloc(sizeof(double) * n);
    s->v = calloc(n, sizeof(double));
    s->xt = malloc(sizeof(double) * n);
    s->g = calloc(n, sizeof(double));
    s->p = calloc(n, sizeof(double));
    for (int i = 0; i < HIST; ++i) {
        s->hs[i] = malloc(sizeof(double) * n);
        s->hy[i] = malloc(sizeof(double) * n);
    }
    s->ework = malloc(sizeof(double) * nT);
    s->gcont = malloc(sizeof(double) * 12 * (size_t)nT);
    s->x0 = malloc(sizeof(double) * n);
    s->q = malloc(sizeof(double) * n);
    s->gold = malloc(sizeof(double) * n);
    s->Hp = malloc(sizeof(double) * n);
    s->tmp_s = malloc(sizeof(double) * n);
    s->tmp_y = malloc(sizeof(double) * n);
    s->log_cap = 10001;
    s->log_alpha = malloc(sizeof(double) * s->log_cap);
    s->log_E = malloc(sizeof(double) * s->log_cap);
    s->log_g2 = malloc(sizeof(double) * s->log_cap);
    /* Optimizer.cpp:124-184: result = data0 (+script init), v = 0, x_n = x, x~ */
    memcpy(s->x, x_init, sizeof(double) * n);
    memcpy(s->xn, x_init, sizeof(double) * n);
    compute_xtilde(s);
    s->targetGRes = char_norm_sq(s, relTol * relTol);
    /* DOTTimeStepper.cpp:150-178 precompute: H and factors at the initial configuration */
    dor_refactor(s, s->x);
    return s;
}

void dor_destroy(dor_sim *s)
{
    if (!s) return;
    ext_release(s);
    for (int pI = 0; pI < s->nParts; ++pI) {
        dor_part *P = &s->parts[pI];
        free(P->l2g); free(P->pos); free(P->ord); free(P->rowptr); free(P->first); free(P->L);
    }
    free(s->parts);
    for (int i = 0; i < HIST; ++i) { free(s->hs[i]); free(s->hy[i]); }
    free(s->T); free(s->Xrest); free(s->A); free(s->vol); free(s->mu); free(s->lam); free(s->mass);
    free(s->fixed); free(s->vf_ptr); free(s->vf_elem); free(s->vf_slot); free(s->adj_ptr);
    free(s->adj_idx); free(s->eblk); free(s->Hval); free(s->He); free(s->epart); free(s->vpart); free(s->dup);
    free(s->x); free(s->xn); free(s->v); free(s->xt); free(s->g); free(s->p);
    free(s->ework); free(s->gcont); free(s->x0); free(s->q); free(s->gold); free(s->Hp); free(s->tmp_s); free(s->tmp_y);
    free(s->log_alpha); free(s->log_E); free(s->log_g2); free(s->usv);
    free(s);
}

void dor_move(dor_sim *s, int n, const int *idx, const double *pos)
{
    for (int k = 0; k < n; ++k)
        for (int d = 0; d < 3; ++d) s->x[3 * idx[k] + d] = pos[3 * k + d];
}

/* Optimizer.cpp:752-881 lineSearch (armijoParam = 0, lowerBound = 0; allowEDecRelTol off,
 * main.cpp:942) with :1076-1093 initStepSize.  returns 1 if the step underflowed to 0. */
static int line_search(dor_sim *s, double *alpha_out, double *lastE)
{
    int n = 3 * s->nV;
    dor_spmv(s, s->p, s->Hp);
    double pg = dotn(s->p, s->g, n), pHp = dotn(s->p, s->Hp, n);
    double alpha = fmax(s->alphaMin, fmin(1.0, -pg / pHp));
    memcpy(s->x0, s->x, sizeof(double) * n);
    for (int i = 0; i < n; ++i) s->x[i] = s->x0[i] + alpha * s->p[i];
    double E = dor_eval_energy(s, s->x);
    int stopped = 0;
    while (E > *lastE && alpha > 0.0) {
        alpha /= 2.0;
        s->numLineSearch++;
        if (alpha == 0.0) {
            stopped = 1;
            break;
        }
        for (int i = 0; i < n; ++i) s->x[i] = s->x0[i] + alpha * s->p[i];
        E = dor_eval_energy(s, s->x);
    }
    *lastE = E;
    *alpha_out = alpha;
    return stopped;
}

/* DOTTimeStepper.cpp:384-504 */
static int solve_one_step(dor_sim *s, double *lastE, double *alpha)
{
    int n = 3 * s->nV;
    double ksi[HIST];
    for (int i = 0; i < n; ++i) s->q[i] = -s->g[i];
    for (int h = s->nh - 1; h >= 0; --h) {
        ksi[h] = dotn(s->hs[h], s->q, n) / s->hys[h];
        for (int i = 0; i < n; ++i) s->q[i] -= ksi[h] * s->hy[h][i];
    }
    dor_apply_precond(s, s->q, s->p);
    for (int h = 0; h < s->nh; ++h) {
        double c = ksi[h] - dotn(s->hy[h], s->p, n) / s->hys[h];
        for (int i = 0; i < n; ++i) s->p[i] += s->hs[h][i] * c;
    }
    int stopped = line_search(s, alpha, lastE);
    /* history update (DOTTimeStepper.cpp:474-494) */
    memcpy(s->gold, s->g, sizeof(double) * n);
    for (int i = 0; i < n; ++i) s->tmp_s[i] = *alpha * s->p[i];
    dor_eval_gradient(s, s->x, s->g);
    for (int i = 0; i < n; ++i) s->tmp_y[i] = s->g[i] - s->gold[i];
    double ys = dotn(s->tmp_y, s->tmp_s, n);
    if (ys > 0.0) {
        int slot;
        if (s->nh == HIST) {
            double *s0 = s->hs[0], *y0 = s->hy[0];
            for (int h = 0; h < HIST - 1; ++h) {
                s->hs[h] = s->hs[h + 1];
                s->hy[h] = s->hy[h + 1];
                s->hys[h] = s->hys[h + 1];
            }
            s->hs[HIST - 1] = s0;
            s->hy[HIST - 1] = y0;
            slot = HIST - 1;
        } else {
            slot = s->nh++;
        }
        memcpy(s->hs[slot], s->tmp_s, sizeof(double) * n);
        memcpy(s->hy[slot], s->tmp_y, sizeof(double) * n);
        s->hys[slot] = ys;
    }
    return stopped;
}

/* Optimizer.cpp:327-368 solve(1) without the script move (dor_move) +
 * DOTTimeStepper.cpp:273-346 fullyImplicit, split into begin / iterate / end so that a test can stop
 * between two L-BFGS iterations and look at (x, g, history): teacher forcing, SURVEY.md section 8(c) F4. */
void dor_step_begin(dor_sim *s)
{
    int n = 3 * s->nV;
    s->t_energy = s->t_grad = s->t_solve = s->t_hess = s->t_factor = 0;
    s->energy_evals = 0;
    s->ls0 = s->numLineSearch;
    s->nh = 0;
    s->log_n = 0;
    s->it = 0;
    s->failed = 0;
    /* initX(2), Optimizer.cpp:442-582 */
    for (int v = 0; v < s->nV; ++v)
        for (int d = 0; d < 3; ++d) {
            double pd = s->fixed[v] ? 0.0 : s->dt * s->v[3 * v + d] + s->dtSq * s->gravity[d];
            s->x[3 * v + d] = s->x[3 * v + d] + 1.0 * pd;
        }
    s->lastE = dor_eval_energy(s, s->x);
    dor_eval_gradient(s, s->x, s->g);
    s->g2 = dotn(s->g, s->g, n);
    s->E0 = s->lastE;
    s->g2_0 = s->g2;
}

/* one pass of the do-while body of fullyImplicit (DOTTimeStepper.cpp:303-337).
 * returns 0 = go on, 1 = converged, 2 = iteration cap, 3 = line search failed */
int dor_step_iterate(dor_sim *s)
{
    int n = 3 * s->nV;
    const int iterCap = 10000;
    double alpha;
    if (solve_one_step(s, &s->lastE, &alpha)) {
        s->failed = 1;
        return 3;
    }
    s->g2 = dotn(s->g, s->g, n);
    if (s->log_n < s->log_cap) {
        s->log_alpha[s->log_n] = alpha;
        s->log_E[s->log_n] = s->lastE;
        s->log_g2[s->log_n] = s->g2;
        s->log_n++;
    }
    if (++s->it >= iterCap) return 2;
    return s->g2 > s->targetGRes ? 0 : 1;
}

int dor_step_end(dor_sim *s, dor_step_stats *st, double T0)
{
    int n = 3 * s->nV, status = 0;
    if (s->failed) status = 2;
    else {
        if (s->it >= 10000) status = 2;
        dor_refactor(s, s->x);
    }
    /* BE update, Optimizer.cpp:354-361 */
    for (int i = 0; i < n; ++i) {
        s->v[i] = (s->x[i] - s->xn[i]) / s->dt;
        s->xn[i] = s->x[i];
    }
    compute_xtilde(s);
    if (st) {
        st->E0 = s->E0;
        st->g2_0 = s->g2_0;
        st->iters = s->it;
        st->ls_halvings = (int)(s->numLineSearch - s->ls0);
        st->energy_evals = s->energy_evals;
        st->status = status;
        st->E = s->lastE;
        st->g2 = s->g2;
        st->ms_total = T0 > 0 ? now_ms() - T0 : 0.0;
        st->ms_energy = s->t_energy;
        st->ms_gradient = s->t_grad;
        st->ms_backsolve = s->t_solve;
        st->ms_hessian = s->t_hess;
        st->ms_factor = s->t_factor;
    }
    return status;
}

/* `timeStepper GSDD`: DOTTimeStepper::solve_oneStep_GSDD (DOTTimeStepper.cpp:507-565) inside fullyImplicit's loop
 * (:299-337).  One iteration = one Gauss-Seidel sweep over the subdomains: for subdomain sI the right-hand side is minus
 * the CURRENT gradient on its vertices (:509-511 for the first, computeGradient_extract :546-550 afterwards), the
 * search direction is the subdomain solve filled into a zero vector (:521-532, ADMMDDTimeStepper::fill :1646-1665), the
 * line search starts at step 1 (Optimizer::initStepSize :1076-1093 -- only TST_DOT estimates the step) and halves while
 * the energy increases (:806-833). */
int dor_step_gsdd(dor_sim *s, dor_step_stats *st)
{
    double T0 = now_ms();
    int n = 3 * s->nV;
    dor_step_begin(s);
    do {
        for (int sI = 0; sI < s->nParts && !s->failed; ++sI) {
            const dor_part *P = &s->parts[sI];
            double *b = malloc(sizeof(double) * 3 * (P->nv > 0 ? P->nv : 1));
            for (int pi = 0; pi < P->nv; ++pi) {
                int v = P->l2g[P->ord[pi]];
                for (int d = 0; d < 3; ++d) b[3 * pi + d] = -s->g[3 * v + d];
            }
            solve_part_any(s, sI, b);
            memset(s->p, 0, sizeof(double) * n);
            for (int i = 0; i < P->nv; ++i) {
                int v = P->l2g[i], pi = P->pos[i];
                for (int d = 0; d < 3; ++d) s->p[3 * v + d] = b[3 * pi + d];
            }
            free(b);
            double alpha = 1.0;
            memcpy(s->x0, s->x, sizeof(double) * n);
            for (int i = 0; i < n; ++i) s->x[i] = s->x0[i] + alpha * s->p[i];
            double E = dor_eval_energy(s, s->x);
            while (E > s->lastE && alpha > 0.0) {
                alpha /= 2.0;
                s->numLineSearch++;
                if (alpha == 0.0) {
                    s->failed = 1;
                    break;
                }
                for (int i = 0; i < n; ++i) s->x[i] = s->x0[i] + alpha * s->p[i];
                E = dor_eval_energy(s, s->x);
            }
            s->lastE = E;
            dor_eval_gradient(s, s->x, s->g);
        }
        if (s->failed) break;
        s->g2 = dotn(s->g, s->g, n);
        if (s->log_n < s->log_cap) {
            s->log_alpha[s->log_n] = 0.0;
            s->log_E[s->log_n] = s->lastE;
            s->log_g2[s->log_n] = s->g2;
            s->log_n++;
        }
        if (++s->it >= 10000) break;
    } while (s->g2 > s->targetGRes);
    return dor_step_end(s, st, T0);
}

/* `timeStepper Newton`: the base Optimizer::fullyImplicit (Optimizer.cpp:654-700) with Optimizer::solve_oneStep
 * (:703-749, needRefactorize): per iteration assemble + factorise the projected Hessian at the current iterate, solve
 * H p = -g, line search from step 1 (:1088), refresh the gradient.  The reference's method for nParts = 1 (the block
 * solve is then the global solve). */
int dor_step_newton(dor_sim *s, dor_step_stats *st)
{
    double T0 = now_ms();
    int n = 3 * s->nV;
    dor_step_begin(s);
    do {
        dor_refactor(s, s->x);
        for (int i = 0; i < n; ++i) s->q[i] = -s->g[i];
        dor_apply_precond(s, s->q, s->p);
        double alpha = 1.0;
        memcpy(s->x0, s->x, sizeof(double) * n);
        for (int i = 0; i < n; ++i) s->x[i] = s->x0[i] + alpha * s->p[i];
        double E = dor_eval_energy(s, s->x);
        while (E > s->lastE && alpha > 0.0) {
            alpha /= 2.0;
            s->numLineSearch++;
            if (alpha == 0.0) {
                s->failed = 1;
                break;
            }
            for (int i = 0; i < n; ++i) s->x[i] = s->x0[i] + alpha * s->p[i];
            E = dor_eval_energy(s, s->x);
        }
        s->lastE = E;
        dor_eval_gradient(s, s->x, s->g);
        if (s->failed) break;
        s->g2 = dotn(s->g, s->g, n);
        if (s->log_n < s->log_cap) {
            s->log_alpha[s->log_n] = alpha;
            s->log_E[s->log_n] = s->lastE;
            s->log_g2[s->log_n] = s->g2;
            s->log_n++;
        }
        if (++s->it >= 10000) break;
    } while (s->g2 > s->targetGRes);
    /* dor_step_end refreshes the factors once more; harmless: the next step's first iteration does it again */
    return dor_step_end(s, st, T0);
}

int dor_step(dor_sim *s, dor_step_stats *st)
{
    double T0 = now_ms();
    dor_step_begin(s);
    while (dor_step_iterate(s) == 0) {
    }
    return dor_step_end(s, st, T0);
}

/* state between two iterations of a running step: iterate, gradient, stored pairs (oldest first) */
int dor_get_lbfgs(const dor_sim *s, double *x, double *g, double *S, double *Y, double *lastE)
{
    int n = 3 * s->nV;
    if (x) memcpy(x, s->x, sizeof(double) * n);
    if (g) memcpy(g, s->g, sizeof(double) * n);
    for (int h = 0; h < s->nh; ++h) {
        if (S) memcpy(S + (size_t)h * n, s->hs[h], sizeof(double) * n);
        if (Y) memcpy(Y + (size_t)h * n, s->hy[h], sizeof(double) * n);
    }
    if (lastE) *lastE = s->lastE;
    return s->nh;
}

/* One L-BFGS-H direction + first line-search trial from a GIVEN iterate and history, nothing of the running
 * state is used or changed except the current factors and x~ (DOTTimeStepper.cpp:386-467, Optimizer.cpp:1076-1093,
 * :791).  Outputs (any may be NULL): g = gradient at x, q = after the first loop, z = M^-1 q, p = direction,
 * alpha0 = clamp(-p.g / p.Hp, 0.1, 1), Etrial = E(x + alpha0 p). */
void dor_probe_direction(dor_sim *s, const double *x, int m, const double *S, const double *Y, double *g_out,
                         double *q_out, double *z_out, double *p_out, double *alpha0, double *Etrial)
{
    int n = 3 * s->nV;
    double *g = malloc(sizeof(double) * n), *q = malloc(sizeof(double) * n), *p = malloc(sizeof(double) * n);
    double *Hp = malloc(sizeof(double) * n), *xt = malloc(sizeof(double) * n);
    double ksi[64], ys[64];
    dor_eval_gradient(s, x, g);
    for (int i = 0; i < n; ++i) q[i] = -g[i];
    for (int h = 0; h < m; ++h) ys[h] = dotn(Y + (size_t)h * n, S + (size_t)h * n, n);
    for (int h = m - 1; h >= 0; --h) {
        ksi[h] = dotn(S + (size_t)h * n, q, n) / ys[h];
        for (int i = 0; i < n; ++i) q[i] -= ksi[h] * Y[(size_t)h * n + i];
    }
    dor_apply_precond(s, q, p);
    if (z_out) memcpy(z_out, p, sizeof(double) * n);
    for (int h = 0; h < m; ++h) {
        double c = ksi[h] - dotn(Y + (size_t)h * n, p, n) / ys[h];
        for (int i = 0; i < n; ++i) p[i] += S[(size_t)h * n + i] * c;
    }
    dor_spmv(s, p, Hp);
    double pg = dotn(p, g, n), pHp = dotn(p, Hp, n);
    double a = fmax(s->alphaMin, fmin(1.0, -pg / pHp));
    for (int i = 0; i < n; ++i) xt[i] = x[i] + a * p[i];
    if (Etrial) *Etrial = dor_eval_energy(s, xt);
    if (alpha0) *alpha0 = a;
    if (g_out) memcpy(g_out, g, sizeof(double) * n);
    if (q_out) memcpy(q_out, q, sizeof(double) * n);
    if (p_out) memcpy(p_out, p, sizeof(double) * n);
    free(g); free(q); free(p); free(Hp); free(xt);
}

int dor_last_iter_log(const dor_sim *s, int cap, double *alpha, double *E, double *g2)
{
    int n = s->log_n < cap ? s->log_n : cap;
    for (int i = 0; i < n; ++i) {
        alpha[i] = s->log_alpha[i];
        E[i] = s->log_E[i];
        g2[i] = s->log_g2[i];
    }
    return s->log_n;
}

void dor_get_state(const dor_sim *s, double *x, double *v, double *xtilde)
{
    int n = 3 * s->nV;
    if (x) memcpy(x, s->x, sizeof(double) * n);
    if (v) memcpy(v, s->v, sizeof(double) * n);
    if (xtilde) memcpy(xtilde, s->xt, sizeof(double) * n);
}

void dor_set_state(dor_sim *s, const double *x, const double *v, const double *xn)
{
    int n = 3 * s->nV;
    memcpy(s->x, x, sizeof(double) * n);
    memcpy(s->v, v, sizeof(double) * n);
    memcpy(s->xn, xn ? xn : x, sizeof(double) * n);
    compute_xtilde(s);
}

double dor_target_gres(const dor_sim *s) { return s->targetGRes; }
void dor_set_alpha_min(dor_sim *s, double a) { s->alphaMin = a; }

/* fixed set changed by the script (rubberBandPull release, AnimScripter.cpp:404-417):
 * Optimizer::solve -> updatePrecondMtrAndFactorize (DOTTimeStepper.cpp:185-270) re-patterns and
 * refactors at the current configuration; x~ is NOT recomputed until the end of the step */
void dor_set_fixed(dor_sim *s, const unsigned char *fixed)
{
    memcpy(s->fixed, fixed, s->nV);
    if (s->ext.create) { /* the pattern of the fixed rows changes with the set */
        ext_release(s);
        ext_build(s);
    }
    dor_refactor(s, s->x);
}

int dor_use_ext_solver(dor_sim *s, const dor_ext_solver *api)
{
    ext_release(s);
    memset(&s->ext, 0, sizeof(s->ext));
    if (api && api->create) {
        s->ext = *api;
        ext_build(s);
    }
    double t = s->t_factor;
    s->factor_failed = 0;
    dor_refactor(s, s->xn);
    s->t_factor = t; /* (the timing of this extra refresh is not a step's) */
    return s->factor_failed;
}

int dor_factor_failed(const dor_sim *s) { return s->factor_failed; }

void dor_get_features(const dor_sim *s, double *A, double *vol, double *mass, double *mu, double *lam)
{
    if (A) memcpy(A, s->A, sizeof(double) * 9 * (size_t)s->nT);
    if (vol) memcpy(vol, s->vol, sizeof(double) * s->nT);
    if (mass) memcpy(mass, s->mass, sizeof(double) * s->nV);
    if (mu) memcpy(mu, s->mu, sizeof(double) * s->nT);
    if (lam) memcpy(lam, s->lam, sizeof(double) * s->nT);
}

void dor_get_dup(const dor_sim *s, int *dup) { memcpy(dup, s->dup, sizeof(int) * s->nV); }
int dor_part_size(const dor_sim *s, int part) { return s->parts[part].nv; }
void dor_part_verts(const dor_sim *s, int part, int *l2g)
{
    memcpy(l2g, s->parts[part].l2g, sizeof(int) * s->parts[part].nv);
}
