// elem_math.hpp -- per-tet FP64 math of the DOT hot path as inlinable device functions.
//
// Everything here works on scalars / small fixed arrays with compile-time indices only, so that
// the arrays live in VGPRs (no scratch).  The file is compiled with -ffp-contract=off: the PSD
// projection of the reference takes data-dependent branches on quantities that are exactly zero at
// the rest state (IglUtils.hpp:271-309 `if(L2 < 0.0)`), so the spectral part must round the same
// way on every build for results to be reproducible.
//
// Reference (paths relative to /root/reference/src):
//   F = Ds * A                       Energy/Energy.cpp:309-321
//   3x3 SVD conventions              Utils/IglUtils.cpp:929-1085, Utils/SVD_EFTYCHIOS/*
//   Psi, dPsi/dsigma, d2Psi, B-left  Energy/Physics_Elasticity/FixedCoRotEnergy.cpp:83-172,
//                                    StableNHEnergy.cpp:91-228
//   PSD clamps                       Utils/IglUtils.hpp:253-309
//   dP/dF in the singular basis      Energy/Energy.cpp:1129-1270
#pragma once
#include <hip/hip_runtime.h>

#define DM_HD __host__ __device__ __forceinline__

namespace dotmi {

struct Mat3 {
    double m[3][3];
};

DM_HD double det3(const Mat3 &M)
{
    return M.m[0][0] * (M.m[1][1] * M.m[2][2] - M.m[1][2] * M.m[2][1]) -
           M.m[0][1] * (M.m[1][0] * M.m[2][2] - M.m[1][2] * M.m[2][0]) +
           M.m[0][2] * (M.m[1][0] * M.m[2][1] - M.m[1][1] * M.m[2][0]);
}

// one Jacobi rotation annihilating a[P][R]; O is the third index
template <int P, int R, int O>
DM_HD void jacobi_rot(double (&a)[3][3], double (&q)[3][3])
{
    double apq = a[P][R];
    if (apq == 0.0) return;
    if (fabs(apq) < 1e-19 * (fabs(a[P][P]) + fabs(a[R][R]))) {
        a[P][R] = a[R][P] = 0.0;
        return;
    }
    double theta = (a[R][R] - a[P][P]) / (2.0 * apq);
    double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    a[P][P] -= t * apq;
    a[R][R] += t * apq;
    a[P][R] = a[R][P] = 0.0;
    double aop = a[O][P], aor = a[O][R];
    a[O][P] = a[P][O] = c * aop - s * aor;
    a[O][R] = a[R][O] = s * aop + c * aor;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double qp = q[i][P], qr = q[i][R];
        q[i][P] = c * qp - s * qr;
        q[i][R] = s * qp + c * qr;
    }
}

// eigen-decomposition of a symmetric 3x3: Q columns = eigenvectors, w ascending
DM_HD void sym_eig3(const Mat3 &Ain, double (&w)[3], Mat3 &Q)
{
    double a[3][3] = {{Ain.m[0][0], Ain.m[0][1], Ain.m[0][2]},
                      {Ain.m[0][1], Ain.m[1][1], Ain.m[1][2]},
                      {Ain.m[0][2], Ain.m[1][2], Ain.m[2][2]}};
    double q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        double dia = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off <= 1e-300 || off <= 1e-20 * dia) break;
        jacobi_rot<0, 1, 2>(a, q);
        jacobi_rot<0, 2, 1>(a, q);
        jacobi_rot<1, 2, 0>(a, q);
    }
    // stable 3-element sort (ascending) of (d_k, column k)
    double d0 = a[0][0], d1 = a[1][1], d2 = a[2][2];
    double c0[3] = {q[0][0], q[1][0], q[2][0]};
    double c1[3] = {q[0][1], q[1][1], q[2][1]};
    double c2[3] = {q[0][2], q[1][2], q[2][2]};
#define DM_CSWAP(da, ca, db, cb)                 \
    if (da > db) {                               \
        double t_ = da; da = db; db = t_;        \
        for (int i_ = 0; i_ < 3; ++i_) {         \
            double u_ = ca[i_]; ca[i_] = cb[i_]; cb[i_] = u_; \
        }                                        \
    }
    DM_CSWAP(d0, c0, d1, c1)
    DM_CSWAP(d1, c1, d2, c2)
    DM_CSWAP(d0, c0, d1, c1)
#undef DM_CSWAP
    w[0] = d0; w[1] = d1; w[2] = d2;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Q.m[i][0] = c0[i];
        Q.m[i][1] = c1[i];
        Q.m[i][2] = c2[i];
    }
}

// F = U diag(S) V^T, U,V in SO(3), S0 >= S1 >= |S2|, sign(S2) = sign(det F)
DM_HD void svd3(const Mat3 &F, Mat3 &U, double (&S)[3], Mat3 &V)
{
    Mat3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C.m[i][j] = F.m[0][i] * F.m[0][j] + F.m[1][i] * F.m[1][j] + F.m[2][i] * F.m[2][j];
    double w[3];
    Mat3 Q;
    sym_eig3(C, w, Q);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        V.m[i][0] = Q.m[i][2];
        V.m[i][1] = Q.m[i][1];
        V.m[i][2] = Q.m[i][0];
    }
    if (det3(V) < 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) V.m[i][2] = -V.m[i][2];
    }
    double b0[3], b1[3], b2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        b0[i] = F.m[i][0] * V.m[0][0] + F.m[i][1] * V.m[1][0] + F.m[i][2] * V.m[2][0];
        b1[i] = F.m[i][0] * V.m[0][1] + F.m[i][1] * V.m[1][1] + F.m[i][2] * V.m[2][1];
        b2[i] = F.m[i][0] * V.m[0][2] + F.m[i][1] * V.m[1][2] + F.m[i][2] * V.m[2][2];
    }
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
        // rank <= 1: unit vector orthogonal to u0 built from the smallest-|component| axis
        double ax = fabs(u0[0]), ay = fabs(u0[1]), az = fabs(u0[2]);
        double e0 = 0, e1 = 0, e2 = 0, d;
        if (ax <= ay && ax <= az) { e0 = 1; d = u0[0]; }
        else if (ay <= az) { e1 = 1; d = u0[1]; }
        else { e2 = 1; d = u0[2]; }
        double t0 = e0 - d * u0[0], t1 = e1 - d * u0[1], t2 = e2 - d * u0[2];
        double nt = sqrt(t0 * t0 + t1 * t1 + t2 * t2);
        u1[0] = t0 / nt; u1[1] = t1 / nt; u1[2] = t2 / nt;
    }
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
    u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
    u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    S[0] = n0;
    S[1] = u1[0] * b1[0] + u1[1] * b1[1] + u1[2] * b1[2];
    S[2] = u2[0] * b2[0] + u2[1] * b2[1] + u2[2] * b2[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        U.m[i][0] = u0[i];
        U.m[i][1] = u1[i];
        U.m[i][2] = u2[i];
    }
}

template <int MAT>
DM_HD double psi(const double (&s)[3], double mu, double lam)
{
    double J = s[0] * s[1] * s[2];
    if (MAT == 0) {
        double a = s[0] - 1, b = s[1] - 1, c = s[2] - 1;
        return mu * (a * a + b * b + c * c) + lam / 2.0 * (J - 1.0) * (J - 1.0);
    }
    double JmA = J - (1.0 + mu / lam);
    return (mu * (s[0] * s[0] + s[1] * s[1] + s[2] * s[2] - 3.0) + lam * JmA * JmA) / 2.0;
}

template <int MAT>
DM_HD void dpsi(const double (&s)[3], double mu, double lam, double (&d)[3])
{
    double J = s[0] * s[1] * s[2];
    double pn[3] = {s[1] * s[2], s[2] * s[0], s[0] * s[1]};
    if (MAT == 0) {
        double t = lam * (J - 1.0);
#pragma unroll
        for (int i = 0; i < 3; ++i) d[i] = 2.0 * mu * (s[i] - 1.0) + pn[i] * t;
    } else {
        double t = lam * (J - (1.0 + mu / lam));
#pragma unroll
        for (int i = 0; i < 3; ++i) d[i] = s[i] * mu + t * pn[i];
    }
}

template <int MAT>
DM_HD void d2psi(const double (&s)[3], double mu, double lam, Mat3 &A)
{
    double J = s[0] * s[1] * s[2];
    double pn[3] = {s[1] * s[2], s[2] * s[0], s[0] * s[1]};
    if (MAT == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) A.m[i][i] = 2.0 * mu + lam * pn[i] * pn[i];
        A.m[0][1] = A.m[1][0] = lam * (s[2] * (J - 1.0) + pn[0] * pn[1]);
        A.m[0][2] = A.m[2][0] = lam * (s[1] * (J - 1.0) + pn[0] * pn[2]);
        A.m[1][2] = A.m[2][1] = lam * (s[0] * (J - 1.0) + pn[2] * pn[1]);
    } else {
        double l2 = lam * (2.0 * J - (1.0 + mu / lam));
#pragma unroll
        for (int i = 0; i < 3; ++i) A.m[i][i] = mu + lam * pn[i] * pn[i];
        A.m[0][1] = A.m[1][0] = s[2] * l2;
        A.m[0][2] = A.m[2][0] = s[1] * l2;
        A.m[1][2] = A.m[2][1] = s[0] * l2;
    }
}

template <int MAT>
DM_HD void bleft(const double (&s)[3], double mu, double lam, double (&b)[3])
{
    double J = s[0] * s[1] * s[2];
    if (MAT == 0) {
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

DM_HD void make_pd3(Mat3 &A)
{
    double w[3];
    Mat3 Q;
    sym_eig3(A, w, Q);
    if (w[0] >= 0.0) return;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (w[i] < 0.0) w[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            A.m[i][j] = Q.m[i][0] * w[0] * Q.m[j][0] + Q.m[i][1] * w[1] * Q.m[j][1] +
                        Q.m[i][2] * w[2] * Q.m[j][2];
}

// closed-form 2x2 clamp exactly as the reference codes it (IglUtils.hpp:271-309)
DM_HD void make_pd2(double &B00, double &B01, double &B10, double &B11)
{
    const double a = B00, b = (B01 + B10) / 2.0, d = B11;
    double b2 = b * b;
    const double D = a * d - b2;
    const double T2 = (a + d) / 2.0;
    const double sq = sqrt(T2 * T2 - D);
    const double L2 = T2 - sq;
    if (L2 < 0.0) {
        const double L1 = T2 + sq;
        if (L1 <= 0.0) {
            B00 = B01 = B10 = B11 = 0.0;
        } else if (b2 == 0.0) {
            B00 = L1;
            B01 = B10 = B11 = 0.0;
        } else {
            const double L1md = L1 - d;
            const double r = L1md / L1;
            B00 = r * L1md;
            B01 = B10 = b * r;
            B11 = b2 / L1;
        }
    }
}

// Spectral coefficients of w * dP/dF in the singular basis: Aw (3x3) and three 2x2 blocks
// Bw[c] = {b00,b01,b10,b11} for pairs (0,1),(1,2),(2,0).  Energy.cpp:1129-1207.
template <int MAT>
DM_HD void spectral_blocks(const double (&S)[3], double mu, double lam, double w, bool project,
                           Mat3 &Aw, double (&Bw)[3][4])
{
    double dE[3], bl[3];
    dpsi<MAT>(S, mu, lam, dE);
    d2psi<MAT>(S, mu, lam, Aw);
    if (project) make_pd3(Aw);
    bleft<MAT>(S, mu, lam, bl);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int cp = (c + 1) % 3;
        double right = dE[c] + dE[cp];
        double sum = S[c] + S[cp];
        const double eps = 1.0e-6;
        if (sum < eps) right /= 2.0 * eps;
        else right /= 2.0 * sum;
        double b00 = bl[c] + right, b11 = b00, b01 = bl[c] - right, b10 = b01;
        if (project) make_pd2(b00, b01, b10, b11);
        Bw[c][0] = w * b00; Bw[c][1] = w * b01; Bw[c][2] = w * b10; Bw[c][3] = w * b11;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Aw.m[i][j] = w * Aw.m[i][j];
}

}  // namespace dotmi
