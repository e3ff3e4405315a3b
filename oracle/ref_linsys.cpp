// ref_linsys.cpp -- TEST INFRASTRUCTURE ONLY.  A driver around the REFERENCE's own linear-system code, compiled where
// it lies (oracle/Makefile target _ref/librefsolver.so): src/LinSysSolver/LinSysSolver.hpp (pattern + coefficient
// indexing), src/LinSysSolver/CHOLMODSolver.cpp (analyze / factorize / solve / multiply) on the vendored CHOLMOD
// 3.0.12 + AMD + COLAMD (plain gcc on their .c files; BLAS/LAPACK = /opt/conda/lib/libmkl_rt.so, a real BLAS that
// is part of the image), and IglUtils::addBlockToMatrix<3> (src/Utils/IglUtils.hpp:143-220).  Nothing is copied.
//
// What it pins (VERDICT r01 "weak 1"): the global assembly a9 (set_pattern index maps, the fixed-vertex unit rows,
// the mass on the free diagonal -- the statements of DOTTimeStepper::computeHElemAndFillIn, DOTTimeStepper.cpp:588-613,
// are re-issued here against the reference's solver object because DOTTimeStepper.cpp itself needs <tbb/tbb.h>) and
// a11 (CHOLMODSolver::factorize / solve / multiply).
#include "CHOLMODSolver.hpp"
#include "IglUtils.hpp"

#include <chrono>
#include <algorithm>
#include <cstring>
#include <set>
#include <vector>

using Solver = DOT::CHOLMODSolver<Eigen::VectorXi, Eigen::VectorXd>;

// a view into the reference solver's protected members (this driver's code; the class itself is the reference's)
struct SolverView : Solver {
    bool has(int r, int c) const { return r < (int)IJ2aI.size() && IJ2aI[r].find(c) != IJ2aI[r].end(); }
    double nnzL() const { return cm.lnz; }   // cholmod_common::lnz of the last analyze
};

// one subdomain system kept alive across time steps (DOTTimeStepper keeps linSysSolver_subdomain[s] the same way:
// set_pattern + analyze_pattern once in the constructor, setCoeff + factorize per refresh, solve per L-BFGS iteration)
struct SubSystem {
    SolverView solver;
    int nv = 0;
    std::vector<int> nbr_ptr, nbr_idx;
    Eigen::VectorXd rhs, res;
};

namespace {
// pattern + coefficients of a (sub-)mesh into the reference's solver object
void assemble(Solver &solver, int nV, int nT, const int *T, const unsigned char *fixed, const double *He, const double *mass)
{
    // Mesh::computeFeatures builds vNeighbor / vFLoc like this (Mesh.cpp:600-614)
    std::vector<std::set<int>> vNeighbor(nV);
    std::vector<std::set<std::pair<int, int>>> vFLoc(nV);
    for (int e = 0; e < nT; ++e)
        for (int a = 0; a < 4; ++a) {
            vFLoc[T[4 * e + a]].insert(std::pair<int, int>(e, a));
            for (int b = 0; b < 4; ++b)
                if (a != b) vNeighbor[T[4 * e + a]].insert(T[4 * e + b]);
        }
    std::set<int> fixedVert;
    for (int v = 0; v < nV; ++v)
        if (fixed[v]) fixedVert.insert(v);
    solver.set_type(1, 2);
    solver.set_pattern(vNeighbor, fixedVert);
    solver.analyze_pattern();
    // DOTTimeStepper::computeHElemAndFillIn, DOTTimeStepper.cpp:588-613 (vInds: Energy.cpp:771-775)
    solver.setZero();
    for (int vI = 0; vI < nV; ++vI) {
        for (const auto &FLocI : vFLoc[vI]) {
            const int e = FLocI.first;
            Eigen::Matrix<double, 12, 12> H;
            for (int r = 0; r < 12; ++r)
                for (int c = 0; c < 12; ++c) H(r, c) = He[(size_t)144 * e + 12 * r + c];
            Eigen::Matrix<int, 1, 4> vInd;
            for (int k = 0; k < 4; ++k) vInd[k] = fixed[T[4 * e + k]] ? (-T[4 * e + k] - 1) : T[4 * e + k];
            DOT::IglUtils::addBlockToMatrix<3>(H.block(FLocI.second * 3, 0, 3, 12), vInd, FLocI.second, &solver);
        }
        if (!fixed[vI]) {
            const int ind0 = vI * 3;
            solver.addCoeff(ind0, ind0, mass[vI]);
            solver.addCoeff(ind0 + 1, ind0 + 1, mass[vI]);
            solver.addCoeff(ind0 + 2, ind0 + 2, mass[vI]);
        }
    }
}
double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" {

// Timing leg of bench.py's cpu_baseline: the reference's CHOLMODSolver on one (sub-)mesh matrix -- analyze_pattern once
// (as the reference does at set-up), then `nfact` numeric factorisations and `nsolve` solves; out = {ms per
// factorisation, ms per solve, non-zeros of L}.  returns 1 when a factorisation failed
int ref_linsys_time(int nV, int nT, const int *T, const unsigned char *fixed, const double *He, const double *mass,
                    int nfact, int nsolve, double *out)
{
    Solver solver;
    assemble(solver, nV, nT, T, fixed, He, mass);
    const int n = 3 * nV;
    // best of the repetitions: the first call pays the BLAS' own start-up, and the GPU boxes' host cores are shared
    double best = 1e300;
    for (int k = 0; k < nfact; ++k) {
        const double t0 = now_ms();
        if (solver.factorize()) return 1;
        best = std::min(best, now_ms() - t0);
    }
    out[0] = nfact > 0 ? best : 0.0;
    Eigen::VectorXd b = Eigen::VectorXd::Ones(n), r;
    best = 1e300;
    for (int k = 0; k < nsolve; ++k) {
        const double t0 = now_ms();
        solver.solve(b, r);
        best = std::min(best, now_ms() - t0);
    }
    out[1] = nsolve > 0 ? best : 0.0;
    out[2] = static_cast<const SolverView &>(solver).nnzL();
    return 0;
}

// ---- persistent subdomain systems for the oracle's external-solver hook (dor_use_ext_solver, oracle/dot_oracle.h) ----------
// nbr: CSR over the local vertices of their neighbours INSIDE the subdomain, itself included, ascending local ids -- the block
// pattern of H_s = R_s H R_s^T.  The reference builds the same pattern from the subdomain mesh's vNeighbor
// (ADMMDDTimeStepper.cpp:88-262 -> LinSysSolver::set_pattern, LinSysSolver.hpp:37-135).
void *ref_sub_create(int nv, const int *nbr_ptr, const int *nbr_idx, const unsigned char *fixed)
{
    SubSystem *S = new SubSystem;
    S->nv = nv;
    S->nbr_ptr.assign(nbr_ptr, nbr_ptr + nv + 1);
    S->nbr_idx.assign(nbr_idx, nbr_idx + nbr_ptr[nv]);
    std::vector<std::set<int>> vNeighbor(nv);
    std::set<int> fixedVert;
    for (int i = 0; i < nv; ++i) {
        for (int k = nbr_ptr[i]; k < nbr_ptr[i + 1]; ++k)
            if (nbr_idx[k] != i) vNeighbor[i].insert(nbr_idx[k]);
        if (fixed[i]) fixedVert.insert(i);
    }
    S->solver.set_type(1, 2);
    S->solver.set_pattern(vNeighbor, fixedVert);
    S->solver.analyze_pattern();
    S->rhs.resize(3 * nv);
    return S;
}

// blocks: 9 doubles (row-major 3 x 3) per pattern entry, in the CSR order of ref_sub_create.  The solver stores the upper
// triangle of the free rows / columns and a unit diagonal for the fixed ones (set_pattern); entries outside that are skipped
// (they are zeros / the unit diagonal in the assembled global matrix the blocks come from).  returns 1 when factorize fails
int ref_sub_factor(void *h, const double *blocks)
{
    SubSystem *S = static_cast<SubSystem *>(h);
    S->solver.setZero();
    for (int i = 0; i < S->nv; ++i)
        for (int k = S->nbr_ptr[i]; k < S->nbr_ptr[i + 1]; ++k) {
            const int j = S->nbr_idx[k];
            if (j < i) continue;
            const double *b = blocks + 9 * (size_t)k;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) {
                    const int row = 3 * i + r, col = 3 * j + c;
                    if (row <= col && S->solver.has(row, col)) S->solver.setCoeff(row, col, b[3 * r + c]);
                }
        }
    return S->solver.factorize() ? 1 : 0;
}

void ref_sub_solve(void *h, double *b)
{
    SubSystem *S = static_cast<SubSystem *>(h);
    std::memcpy(S->rhs.data(), b, sizeof(double) * 3 * S->nv);
    S->solver.solve(S->rhs, S->res);
    std::memcpy(b, S->res.data(), sizeof(double) * 3 * S->nv);
}

double ref_sub_nnzL(void *h) { return static_cast<SubSystem *>(h)->solver.nnzL(); }

void ref_sub_destroy(void *h) { delete static_cast<SubSystem *>(h); }

// T: nT*4, fixed: nV, He: nT*144 row-major element Hessians (already projected, dt^2 vol included), mass: nV
// rhs, x: n = 3 nV.  Outputs: sol = A^-1 rhs, Ax = A x, and (if dense != NULL) the n*n matrix the solver holds.
// returns 0, or 1 when the factorisation failed
int ref_linsys_run(int nV, int nT, const int *T, const unsigned char *fixed, const double *He, const double *mass,
                   const double *rhs, const double *x, double *sol, double *Ax, double *dense)
{
    Solver solver;
    assemble(solver, nV, nT, T, fixed, He, mass);
    const int n = 3 * nV;
    if (dense) {
        Eigen::SparseMatrix<double> M;
        solver.getCoeffMtr(M);
        std::memset(dense, 0, sizeof(double) * (size_t)n * n);
        for (int k = 0; k < M.outerSize(); ++k)
            for (Eigen::SparseMatrix<double>::InnerIterator it(M, k); it; ++it) dense[(size_t)it.row() * n + it.col()] = it.value();
    }
    if (solver.factorize()) return 1;   // CHOLMODSolver::factorize returns !cholmod_factorize(...)
    Eigen::VectorXd b = Eigen::Map<const Eigen::VectorXd>(rhs, n), r;
    solver.solve(b, r);
    std::memcpy(sol, r.data(), sizeof(double) * n);
    Eigen::VectorXd xv = Eigen::Map<const Eigen::VectorXd>(x, n), y;
    solver.multiply(xv, y);
    std::memcpy(Ax, y.data(), sizeof(double) * n);
    return 0;
}

}  // extern "C"
