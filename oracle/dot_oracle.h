/*
 * dot_oracle.h -- CPU restatement of the reference DOT time-step (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle for dot_amd: a plain-C (C99 + optional OpenMP) restatement of the
 * algorithm of penn-graphics-research/DOT's DOTTimeStepper hot path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product path
 * (dot_amd/csrc, libdotmi.so) never links, includes or calls anything in this directory.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference/src).
 *
 * Pinning status (see oracle/README.md and DESIGN.md section "Oracle"):
 *   - 3x3 SVD, Psi(sigma), dPsi/dsigma, makePD, makePD2d, dF_div_dx_mult: checked against the
 *     reference's own sources compiled unmodified into oracle/_ref (ref_pin).
 *   - METIS partition: produced by the reference's vendored METIS 5.1.0 compiled into oracle/_ref.
 *   - step level: tolerance constant and L-BFGS iteration counts per step published in
 *     BASELINE.md section 2 (bunny5K FCR/8 parts, bar17K SNH/32 parts).
 *   - global Hessian scatter (LinSysSolver::set_pattern indexing, addBlockToMatrix, fixed rows, mass), factor, solve,
 *     SpMV: checked against the reference's LinSysSolver.hpp + CHOLMODSolver.cpp compiled unmodified on the vendored
 *     CHOLMOD (oracle/_ref/librefsolver.so, tests/golden/ref_linsys.npz) -- round 2.
 *   - d2Psi/dsigma2, B-coefficients, dP/dF assembly, subdomain-matrix fix-up, two-loop, line search: restated from
 *     source, validated by finite differences / algebraic identities (the reference's own test strategy,
 *     Energy.cpp:1279-1521); the reference's .cpp files for these need TBB headers, which this image lacks, so they
 *     are NOT compiled here.
 */
#ifndef DOT_ORACLE_H
#define DOT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define DOR_FCR 0 /* FixedCoRotEnergy  */
#define DOR_SNH 1 /* StableNHEnergy (non-log variant, Types.hpp:37 SNH_WITHLOG off) */

/* ---- element-level math (row-major 3x3) ---- */
void dor_svd3(const double F[9], double U[9], double S[3], double V[9]);
double dor_psi(int mat, const double s[3], double mu, double lam);
void dor_dpsi(int mat, const double s[3], double mu, double lam, double d[3]);
void dor_d2psi(int mat, const double s[3], double mu, double lam, double A[9]);
void dor_bleft(int mat, const double s[3], double mu, double lam, double b[3]);
void dor_make_pd3(double A[9]);
void dor_make_pd2(double B[4]);
void dor_dPdF(int mat, const double U[9], const double S[3], const double V[9], double mu,
              double lam, double w, int project, double M[81]);
void dor_dFdx_mult_vec(const double P[9], const double A[9], double g[12]);
/* element Hessian from 4 vertex positions; H is 12x12 row-major */
void dor_elem_hessian_x(int mat, const double x4[12], const double A[9], double mu, double lam,
                        double w, int project, double H[144]);
void dor_elem_energy_grad_x(int mat, const double x4[12], const double A[9], double mu, double lam,
                            double w, double *psi_w, double g[12]);

/* optional batch SVD (n row-major 3x3 in; U, S, V out) used by every simulation-level evaluation instead of dor_svd3;
 * tests bind it to the reference's own AVX kernel (oracle/_ref/librefpin.so:ref_svd).  NULL restores dor_svd3. */
typedef int (*dor_svd_batch_fn)(int n, const double *F, double *U, double *S, double *V);
void dor_set_svd_batch(dor_svd_batch_fn fn);

/* ---- simulation object ---- */
typedef struct dor_sim dor_sim;

/* optional external solver for the subdomain systems (DOTTimeStepper.cpp:363-377 factorize, :406-431 solve): bench.py's
 * second CPU leg binds it to the reference's own CHOLMODSolver (oracle/_ref/librefsolver.so: ref_sub_*), so that the CPU
 * baseline's linear algebra is the reference's.  One object per subdomain: create(local vertex count, CSR of the local
 * neighbours of every local vertex incl. itself -- ascending local ids --, fixed flags) once; factor(blocks) with the
 * 3 x 3 row-major blocks of H_s in that CSR order after every refresh; solve(b) in place, local vertex order. */
typedef struct {
    void *(*create)(int nv, const int *nbr_ptr, const int *nbr_idx, const unsigned char *fixed);
    int (*factor)(void *h, const double *blocks);
    void (*solve)(void *h, double *b);
    void (*destroy)(void *h);
} dor_ext_solver;
/* api == NULL: back to the built-in envelope Cholesky.  Re-factors at the current x.  returns 0, or 1 if a factorisation failed */
int dor_use_ext_solver(dor_sim *s, const dor_ext_solver *api);
/* 1 if any refresh since creation / since the last dor_use_ext_solver met a subdomain matrix that could not be factorised
 * (the steps after it ran on a stale or partial factor: a timing or parity leg must not be trusted) */
int dor_factor_failed(const dor_sim *s);

typedef struct {
    int iters;         /* L-BFGS iterations this step (innerIterAmt delta) */
    int ls_halvings;   /* numOfLineSearch delta */
    int energy_evals;
    int status;        /* 0 ok, 2 = iteration cap or line-search failure */
    double E0, g2_0;   /* after initX */
    double E, g2;      /* at exit */
    double ms_total, ms_energy, ms_gradient, ms_backsolve, ms_hessian, ms_factor;
} dor_step_stats;

dor_sim *dor_create(int nV, int nT, const double *Xrest, const int *T, double YM, double PR,
                    double rho, int material, double dt, int withGravity,
                    const unsigned char *fixed, const double *x_init, const int *epart,
                    int nParts, double relTol);
/* the same with an optional VERTEX partition vpart[nV] (METIS::partMesh_nodes): subdomains = disjoint vertex sets, the
 * block-Jacobi initialiser of LBFGS-JH (LBFGSTimeStepper.cpp:70-90, :240-262, :381-393); epart may then be NULL */
dor_sim *dor_create_v(int nV, int nT, const double *Xrest, const int *T, double YM, double PR,
                      double rho, int material, double dt, int withGravity,
                      const unsigned char *fixed, const double *x_init, const int *epart, const int *vpart,
                      int nParts, double relTol);
void dor_destroy(dor_sim *s);

/* scripted Dirichlet motion: x[idx[k]] = pos[3k..] (AnimScripter.cpp:456-466) */
void dor_move(dor_sim *s, int n, const int *idx, const double *pos);
int dor_step(dor_sim *s, dor_step_stats *st);
/* the same step with the reference's GSDD iteration (DOTTimeStepper::solve_oneStep_GSDD, DOTTimeStepper.cpp:507-565) */
int dor_step_gsdd(dor_sim *s, dor_step_stats *st);
/* ... with the reference's projected Newton (base Optimizer::solve_oneStep, Optimizer.cpp:703-749) */
int dor_step_newton(dor_sim *s, dor_step_stats *st);
/* the same step in pieces (teacher forcing: stop between two L-BFGS iterations):
 * begin = initX + first evaluation; iterate = one solve_oneStep, returns 0 go on / 1 converged / 2 cap /
 * 3 line search failed; end = refactor + BE update (T0 = 0: no wall time) */
void dor_step_begin(dor_sim *s);
int dor_step_iterate(dor_sim *s);
int dor_step_end(dor_sim *s, dor_step_stats *st, double T0);
/* iterate, gradient and the stored pairs (oldest first, m*n each) of a running step; returns m */
int dor_get_lbfgs(const dor_sim *s, double *x, double *g, double *S, double *Y, double *lastE);
/* one direction + first trial from a given iterate and history (current factors and x~) */
void dor_probe_direction(dor_sim *s, const double *x, int m, const double *S, const double *Y, double *g_out,
                         double *q_out, double *z_out, double *p_out, double *alpha0, double *Etrial);
/* per-iteration log of the last step: alpha, E, g2 (iterStats.txt columns) */
int dor_last_iter_log(const dor_sim *s, int cap, double *alpha, double *E, double *g2);

void dor_get_state(const dor_sim *s, double *x, double *v, double *xtilde);
void dor_set_state(dor_sim *s, const double *x, const double *v, const double *xn);
double dor_target_gres(const dor_sim *s);
/* lower clamp of the initial step length: 0.1 = DOT (Optimizer.cpp:1085), 1.0 = unit step of the other steppers (:1088) */
void dor_set_alpha_min(dor_sim *s, double a);
void dor_set_fixed(dor_sim *s, const unsigned char *fixed);

/* features / structure getters */
void dor_get_features(const dor_sim *s, double *A /*nT*9*/, double *vol /*nT*/, double *mass /*nV*/,
                      double *mu, double *lam);
void dor_get_dup(const dor_sim *s, int *dup);
int dor_part_size(const dor_sim *s, int part); /* local vertex count */
void dor_part_verts(const dor_sim *s, int part, int *l2g);

/* kernel-level entry points (mirror of the C ABI's parity entry points) */
double dor_eval_energy(dor_sim *s, const double *x);
void dor_eval_gradient(dor_sim *s, const double *x, double *g);
void dor_eval_elem_hessians(dor_sim *s, const double *x, double *H /*nT*144*/);
void dor_refactor(dor_sim *s, const double *x);              /* H(x), H_s, factor */
void dor_apply_precond(dor_sim *s, const double *r, double *p);
void dor_spmv(dor_sim *s, const double *p, double *Hp);
/* dense principal sub-matrix of the assembled global H on a part (n_s*3)^2 row-major */
void dor_part_dense(const dor_sim *s, int part, double *Hs);
void dor_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
