/*
 * dotmi.h -- C ABI of the MI355X-native DOT time-step (libdotmi.so).
 *
 * Drop-in boundary for the reference's DOT stepper: a host adapter with the reference's
 * `DOT::Optimizer<3>` surface (src/TimeStepper/Optimizer.hpp:83-112; picked by the factory in
 * src/main.cpp:905-938) forwards to these entry points.  Plain pointers and sizes only; caller
 * owns every host array; the handle owns all device memory, the rocBLAS handle and (N>1) the
 * RCCL communicator.  One host thread per handle.  No exceptions cross this boundary.
 *
 * Return convention (mirrors Optimizer::solve, Optimizer.cpp:327-368):
 *   0  ok / stepped
 *   2  stepped but hit the iteration cap or the line search underflowed (solve() == 2)
 *  <0  error (DOTMI_E_*); dotmi_last_error() has the text
 *
 * All floating point is FP64; indices are int32; positions / velocities / gradients are
 * nV x 3 row-major (xyz interleaved), exactly the layout of the reference's `velocity` vector
 * (Optimizer.cpp:357) and of `result.V` rows.
 */
#ifndef DOTMI_H
#define DOTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DOTMI_ENERGY_FCR 0 /* FixedCoRotEnergy   (src/Energy/Physics_Elasticity/FixedCoRotEnergy.cpp) */
#define DOTMI_ENERGY_SNH 1 /* StableNHEnergy     (src/Energy/Physics_Elasticity/StableNHEnergy.cpp)  */

#define DOTMI_E_INVALID -1  /* bad argument */
#define DOTMI_E_DEVICE -2   /* HIP / rocBLAS / RCCL runtime error */
#define DOTMI_E_NOTSPD -3   /* a subdomain Hessian was not positive definite (non-positive pivot);
                               the reference dumps the matrix and exit(-1)s, Optimizer.cpp:301-313 */
#define DOTMI_E_NOGPU -4    /* no usable HIP device: the product path has no CPU fallback */

typedef struct dotmi_handle dotmi_handle;

/* Mesh<3> fields the path consumes (src/Mesh.hpp:38-60): rest positions V_rest, tets F,
 * per-element Lame parameters u / lambda (Mesh.cpp:741-744), density (lumped mass is derived as in
 * Mesh.cpp:552-585), the fixed set isFixedVert (after AnimScripter::initAnimScript,
 * AnimScripter.cpp:29) and the element partition that METIS::partMesh returns
 * (src/Utils/METIS.hpp:109-140, consumed in ADMMDDTimeStepper.cpp:88-262). */
typedef struct {
    int32_t nV, nT;
    const double *X_rest;  /* nV*3 */
    const int32_t *T;      /* nT*4, positive orientation */
    const double *mu;      /* nT */
    const double *lambda;  /* nT */
    double density;
    const uint8_t *fixed;  /* nV, 1 = Dirichlet */
    const int32_t *epart;  /* nT, values in [0,nParts); NULL = partition with dotmi_partition() */
    int32_t nParts;
    const int32_t *vpart;  /* optional, nV values in [0,nParts): a VERTEX partition (METIS::partMesh_nodes,
                              METIS.hpp:161-193).  When given, subdomain s is the vertex set {v : vpart[v] == s} --
                              disjoint blocks of the global Hessian, no interface, no averaging: the block-Jacobi
                              initialiser of LBFGS-JH (LBFGSTimeStepper.cpp:70-90, :240-262, :381-393); epart is then
                              ignored.  Single GPU. */
} dotmi_mesh;

/* Config / Optimizer constants (src/Config.hpp, Optimizer.cpp:98-111, DOTTimeStepper.cpp:45) */
typedef struct {
    int32_t energy;    /* DOTMI_ENERGY_* */
    double dt;         /* Optimizer::setTime */
    double gravity[3]; /* (0,-9.80665,0) unless turnOffGravity */
    double relTol;     /* setRelGL2Tol argument, 1e-5 */
    int32_t history;   /* L-BFGS pairs, 5 */
    int32_t iterCap;   /* 10000 */
    double alphaMin;   /* lower clamp of the initial step length, 0.1 (Optimizer.cpp:1085) */
    int32_t device;    /* HIP device ordinal */
    int32_t rank;      /* this process' rank among `world` cooperating handles */
    int32_t world;     /* 1 = single GPU */
    const void *comm_id; /* world>1: the 128-byte id from dotmi_comm_unique_id (same on all ranks) */
    int32_t flags;     /* DOTMI_FLAG_* */
    /* world>1, optional: a host sum-all-reduce the library calls INSTEAD of RCCL (comm_id may then be NULL): buf holds
     * n doubles in host memory, to be replaced on every rank by the sum over the ranks; every rank calls it with the
     * same n in the same order.  The library stages the payload through pinned host memory.  This is how a
     * deployment without RCCL (or a test: two processes on one GPU over gloo) runs the sharded path. */
    void (*allreduce)(void *ctx, double *buf, int64_t n);
    void *allreduce_ctx;
} dotmi_params;

#define DOTMI_FLAG_FORCE_DIST 4     /* take the sharded code path (element lists, partial sums, RCCL
                                       all-reduces) even with world == 1: lets a 1-GPU box test it */
#define DOTMI_FLAG_TIME_BACKSOLVE 2 /* bracket every 8th launch of the streaming back-solve kernel inside dotmi_step
                                       with HIP events on the handle's stream (an event record costs ~6 us of
                                       stream time); totals land in dotmi_step_stats */
#define DOTMI_FLAG_HOST_LOOP 8      /* drive the L-BFGS loop from the host (one stream synchronisation per
                                     * line-search trial) instead of the device-resident loop control; the
                                     * two produce identical iterates -- kept for A/B tests.  (The sharded
                                     * multi-GPU path runs the device-resident loop too, with deterministic
                                     * slot batches; this flag selects the host loop there as well.) */

#define DOTMI_FLAG_TIME_PHASES 16    /* drive the loop from the host and bracket every phase of an iteration with HIP
                                     * events on the handle's stream; the device times land in
                                     * dotmi_step_stats.ms_phase under the reference's timer_step slots.  Costs
                                     * ~8 event records per iteration: for info.txt, not for benchmarks */

/* indices of dotmi_step_stats.ms_phase = the reference's timer_step activities (src/main.cpp:867-880) */
enum {
    DOTMI_T_MATRIX_COMPUTATION = 0,  /* element Hessians + global assembly (slot started at DOTTimeStepper.cpp:576) */
    DOTMI_T_MATRIX_ASSEMBLY = 1,     /* subdomain matrices (fillInDecomposedHessians, :623) */
    DOTMI_T_SYMBOLIC_FACTORIZATION = 2,
    DOTMI_T_NUMERICAL_FACTORIZATION = 3,
    DOTMI_T_BACKSOLVE = 4,           /* subdomain solves + merge (:404-452) */
    DOTMI_T_LINESEARCH_OTHER = 5,    /* alpha_0 SpMV, stepForward (Optimizer.cpp:756-881) */
    DOTMI_T_MODIFY_GRAD = 6,         /* two-loop, first half (:386-400) */
    DOTMI_T_MODIFY_SEARCHDIR = 7,    /* two-loop, second half (:455-467) */
    DOTMI_T_UPDATE_HISTORY = 8,      /* gradient + (s, y) update (:475-494) */
    DOTMI_T_LINESEARCH_EVAL = 9,     /* energy evaluations of the line search (Optimizer.cpp:791,830) */
    DOTMI_T_FULLYIMPLICIT_ECOMP = 10, /* initX + first energy / gradient (DOTTimeStepper.cpp:288-293) */
    DOTMI_T_SOLVE_EXTRACOMP = 11,    /* BE update, x~ (Optimizer.cpp:353-365) */
    DOTMI_T_COMPGRAD = 12,
    DOTMI_T_CCD = 13,
    DOTMI_T_COUNT = 14
};

#define DOTMI_FLAG_GSDD 32           /* dotmi_step runs the reference's Gauss-Seidel domain-decomposition iteration
                                     * (`timeStepper GSDD <n>`, DOTTimeStepper::solve_oneStep_GSDD,
                                     * DOTTimeStepper.cpp:507-565) on the same factors and kernels instead of
                                     * L-BFGS-H: per iteration one sweep over the subdomains -- solve H_s p_s = -g_s,
                                     * line search from step 1 along p_s, refresh the gradient.  stats.iters = sweeps.
                                     * Single GPU. */

#define DOTMI_FLAG_ASYNC_REFRESH 128 /* dotmi_step returns as soon as the refresh at the end of the step (element Hessians,
                                     * assembly, factorisation: DOTTimeStepper.cpp:349-380) is ENQUEUED instead of waiting
                                     * for it: the caller's work between two steps (moving the handles, writing output)
                                     * overlaps it.  Its device times (ms_hessian, ms_factor) and its verdict
                                     * (DOTMI_E_NOTSPD) are then reported by the NEXT dotmi_step -- whose loop has by then
                                     * run on that factorisation -- or by the next call that needs the factors.  Single
                                     * rank, device loop; ignored otherwise. */
#define DOTMI_FLAG_NEWTON 64         /* dotmi_step runs the reference's projected Newton (`timeStepper Newton`, the base
                                     * Optimizer::fullyImplicit / solve_oneStep, Optimizer.cpp:654-749): every iteration
                                     * re-evaluates the projected Hessian at the current iterate, factorises, solves
                                     * H p = -g and line-searches from step 1; no refresh at the end of the step.  It is
                                     * the reference's method when the mesh is ONE subdomain (nParts = 1); with more
                                     * subdomains the solve is the domain-decomposed block solve instead of H^-1.
                                     * stats.iters = Newton iterations.  Single GPU. */
#define DOTMI_FLAG_OWNER_EXCHANGE 256 /* N > 1 (or DOTMI_FLAG_FORCE_DIST), device loop: owner-computes exchange.  Every rank keeps the
                                      * loop's vectors valid on the vertices of ITS subdomains only (zero elsewhere); the two
                                      * vector all-reduces of an iteration carry just the entries of vertices held by more than one
                                      * rank (packed; interior entries never travel), the dot products ride in the tails of those
                                      * packets (each vertex counted once over the ranks), alpha_0's p.Hp uses each rank's own
                                      * elements' part of H, and the positions are made whole again once per step.  Implies the
                                      * sharded element pass and refresh.  Same iterations as one GPU; tests/test_gpu_two_ranks.py */

typedef struct {
    int32_t iters;        /* L-BFGS iterations (innerIterAmt delta, DOTTimeStepper.cpp:338) */
    int32_t ls_halvings;  /* numOfLineSearch delta (Optimizer.cpp:816) */
    int32_t energy_evals;
    int32_t status;       /* 0 / 2 */
    double E0, g2_0;      /* after initX (iterStats.txt header line, DOTTimeStepper.cpp:299) */
    double E, g2;         /* at exit */
    double ms_total;      /* host wall time of the step */
    double ms_loop;       /* L-BFGS loop */
    double ms_hessian;    /* element Hessians + global assembly + dense gather (device time) */
    double ms_factor;     /* inverse-Cholesky factors of all owned subdomains (device time) */
    double ms_precond;    /* DOTMI_FLAG_TIME_BACKSOLVE: summed device time of the timed backsolve_kernel launches */
    int64_t precond_launches; /* how many launches were timed */
    int64_t precond_bytes; /* algorithmic bytes per back-solve launch: 8 x the structural non-zeros of the
                              block-sparse inverse factors X_s of this rank (each is streamed once) */
    double factor_flops;  /* FP64 flop of one factorisation of this rank's subdomains as executed (GEMMs incl. the
                             identity padding + diagonal blocks): ms_factor's MFMA roofline */
    double ms_phase[DOTMI_T_COUNT]; /* device milliseconds of this step under the reference's timer_step slots.  Always
                             filled: MATRIX_COMPUTATION, MATRIX_ASSEMBLY, NUMERICAL_FACTORIZATION (HIP events of the
                             refresh at the end of the step).  The loop slots (BACKSOLVE ... FULLYIMPLICIT_ECOMP,
                             SOLVE_EXTRACOMP) only with DOTMI_FLAG_TIME_PHASES. */
    /* sharded (N > 1) path: the all-reduces this rank issued during the step, end-of-step refresh included */
    int64_t collective_calls;       /* number of all-reduces */
    int64_t collective_bytes;       /* their summed payload (8 x doubles) */
    double ms_collective;           /* DOTMI_FLAG_TIME_BACKSOLVE + RCCL: summed device time between HIP events put
                                       around every 8th ncclAllReduce on the handle's stream (queueing behind the
                                       slower rank included) */
    int64_t collective_timed;       /* how many were bracketed */
    int64_t collective_timed_bytes; /* and their payload */
    int64_t backsolve_launches;     /* back-solves of the step's loop that ran to their end (= iters) */
    int64_t backsolve_stopped;      /* speculative back-solves the controller stopped: ls_halvings + 1 in a step whose device
                                       loop issued them on the trial gradient (DESIGN.md section 5,
                                       DOTMI_EARLY_BACKSOLVE), else 0 */
    int64_t backsolve_held;         /* slots whose back-solve waited for the controller's verdict instead of streaming beside it
                                       (the trial was expected to be rejected: a two-level forecast from the outcomes of the earlier trials of its kind) */
    int64_t backsolve_held_rejected; /* ... and were rejected: these left without reading a factor (they are part of
                                       backsolve_stopped) */
    int64_t paired_slots;           /* DOTMI_PAIR_TRIALS: slots that evaluated the half step in full and the energy of the full step
                                       (the line search of Optimizer.cpp:806-833 with its first two trials in one launch) */
    int64_t paired_redone;          /* ... whose full step was acceptable after all: evaluated again, in full, by the next slot */
    int64_t spec_slots;             /* DOTMI_SPEC_STEP: new-direction slots that evaluated the unit step beside the direction kernel ... */
    int64_t spec_redone;            /* ... whose step estimate alpha_0 turned out below 1: evaluated again, at alpha_0, by the next slot */
} dotmi_step_stats;

/* ---- lifetime -------------------------------------------------------------------------------- */
/* Builds all device state, computes restTriInv / volumes / lumped mass / tolerance, and performs
 * DOTTimeStepper::precompute (DOTTimeStepper.cpp:150-178): Hessian + subdomain factors at x_init.
 * x_init: nV*3 initial positions (result.V after initAnimScript; usually == X_rest). */
int dotmi_create(const dotmi_mesh *mesh, const dotmi_params *params, const double *x_init,
                 dotmi_handle **out);
void dotmi_destroy(dotmi_handle *h);
const char *dotmi_last_error(const dotmi_handle *h); /* h may be NULL: last create() error */

/* Host-only planning helper (touches no device): splits parts 0..nParts-1 into `world` contiguous
 * groups balanced by sum n_s^2 -- the bytes the back-solve streams per L-BFGS iteration.
 * first_part has world+1 entries; rank r owns parts [first_part[r], first_part[r+1]).  This is the
 * split dotmi_create applies; elements follow their part (ADMMDDTimeStepper.cpp:161). */
int dotmi_plan_shards(int32_t nParts, const int32_t *part_scalar_size, int32_t world, int32_t *first_part);

/* Host-only planning helper (touches no device): everything rank `rank` of `world` owns under dotmi_create's plan --
 * parts [*p0, *p1) (dotmi_plan_shards on the parts' scalar sizes), the elements of those parts (elems: up to nT
 * ids ascending, *n_elems of them; elemList_subdomain of the owned parts, ADMMDDTimeStepper.cpp:161), the vertex slice
 * [*v0, *v1) whose inertia terms / SpMV rows it adds, and part_size[nParts] = 3 x vertices of every part.  Any output
 * pointer may be NULL. */
int dotmi_plan_rank(int32_t nV, int32_t nT, const int32_t *T, const int32_t *epart, int32_t nParts, int32_t rank,
                    int32_t world, int32_t *p0, int32_t *p1, int32_t *elems, int32_t *n_elems, int32_t *v0, int32_t *v1,
                    int32_t *part_size);

/* Host-only planning helper (touches no device): the nested-dissection layout dotmi_create gives the
 * dense blocks of parts [p0,p1).  The reference leaves the ordering of each subdomain matrix to CHOLMOD's
 * analyze step (CHOLMODSolver.cpp:103-141); here every owned subdomain is ordered [A | C | S] recursively
 * (S a vertex separator, no mesh edge between A and C) on one tree shared by the owned parts, region sizes
 * padded to the maximum over the parts.  nodes: n_nodes rows of 6 int32 {off, size, childA, childC, offS,
 * sizeS} (children -1 for a dense leaf, root = row 0); nmax: padded scalar size (multiple of 64);
 * pos: for every part in [p0,p1), for every local vertex in ascending global id, the padded scalar
 * position of its first dof.  levels < 0 / min_split < 128 select the defaults.  nodes, pos may be NULL. */
/* (host only) the element patches of the element pass for all elements of a mesh (dot_amd/csrc/patches.hpp); call with
 * elem == NULL for the sizes first.  tests/test_patches.py */
int dotmi_plan_patches(int32_t nV, int32_t nT, const int32_t *T, const double *X, int32_t PE, int32_t *n_patches, int32_t *pv,
                       int32_t *n_slots, int32_t *elem, uint16_t *tl, uint16_t *epos, int32_t *pv_gid, int32_t *pv_slot,
                       int32_t *pv_cnt, uint16_t *c_ptr, int32_t *pp_rng);
/* (host only, round 6) the VERTEX patches of the one-launch element pass + gather (dot_amd/csrc/vpatches.hpp, k_elemvert.hip): a
 * patch owns up to max_own vertices and carries every element incident to them.  hdr[5] = {patches, PE, most touched vertices, most
 * owned vertices, longest sum of runs}; with owner != NULL also: owner[nV] (the patch that owns a vertex), elem[patches * PE] (global
 * element of every slot, -1 padding), eown[patches * PE] (1: this patch counts the element's energy), runlen[nV] (entries of the
 * vertex' gradient run in its patch).  tests/test_gpu_round6.py / test_host_logic.py */
int dotmi_plan_vpatches(int32_t nV, int32_t nT, const int32_t *T, const double *X, int32_t PE, int32_t max_own, int32_t *hdr,
                        int32_t *owner, int32_t *elem, int32_t *eown, int32_t *runlen);
/* (host only) the level schedule of the tile factorisation for one nt x nt tile block with the given upper tile pattern in
 * the compact row-block layout; see dot_amd/csrc/tile_factor.hpp and tests/test_tile_schedule.py */
int dotmi_plan_tile_schedule(int32_t nt, const uint8_t *live, const uint8_t *pattern, const int32_t *c0, int32_t eager_min,
                             int32_t eager_chunk, int64_t *tasks, int64_t *prods, int64_t *n_tasks, int64_t *n_prods,
                             int64_t *n_levels, int64_t *storage, int64_t *scratch_base, int64_t *row_off, int32_t *row_ld);
/* (host only, round 6) that schedule in the TWO-LEVEL form of the block solve (dotmi_backsolve_form = 1) for a leaves-first block:
 * leaf_tile[t] = tile row t belongs to a leaf of the dissection; a separator's row block j keeps the leaf tile columns
 * [c0m[j], c0m[j] + ntm[j]) of its sub-tree in a second storage range (row_off_m / row_ld_m), its main range starts at its
 * sub-tree's first separator column c0[j].  Tiles (leaf i, separator j) receive T_ij = sum over the tiles m >= i of i's leaf of
 * Q_im R_mj (post = store); the inverse's other cross terms do not exist.  tests/test_tile_schedule.py runs it in numpy */
int dotmi_plan_tile_schedule_two_level(int32_t nt, const uint8_t *live, const uint8_t *pattern, const int32_t *c0,
                                       const uint8_t *leaf_tile, const int32_t *c0m, const int32_t *ntm, int32_t eager_min,
                                       int32_t eager_chunk, int64_t *tasks, int64_t *prods, int64_t *n_tasks, int64_t *n_prods,
                                       int64_t *n_levels, int64_t *storage, int64_t *row_off, int32_t *row_ld, int64_t *row_off_m,
                                       int32_t *row_ld_m);
/* (host only) the dependencies of that task list for the dataflow form of the factorisation (DOTMI_TILE_FLOW, one launch of
 * persistent workgroups; tile_flow_kernel): task v waits for dep_idx[dep_ptr[v] .. dep_ptr[v+1]).  Same arguments and task
 * order as dotmi_plan_tile_schedule; dep_idx == NULL returns the count only. */
int dotmi_plan_tile_deps(int32_t nt, const uint8_t *live, const uint8_t *pattern, const int32_t *c0, int32_t eager_min,
                         int32_t eager_chunk, int64_t *dep_ptr, int64_t *dep_idx, int64_t *n_deps);
/* (host only) the job table of the back-solve launches dotmi_create builds for parts [p0, p1) of this mesh: how the rows of every
 * tree region are cut into tiles, which launch / kernel form takes them and which workgroup runs them (dot_amd/csrc/bs_tiles.hpp;
 * the reference has no counterpart: CHOLMODSolver::solve, CHOLMODSolver.cpp:149-163, walks CHOLMOD's supernodes).  tiles: up to cap
 * rows of 6 int32 {local part, first row, rows, first column, tile index in the part, job}; job = workgroup of its launch (wide
 * launch: 0 .. n_wide-1; narrow launch: one-tile jobs 0 .. n_narrow-1, then n_narrow + k for the four tiles of pack k; -1: the
 * two-phase kernel of rows beyond 5120 columns).  counts[8] = {tiles, n_wide, n_narrow, n_packs, n_long, nmax, shallow, few}.
 * tiles may be NULL (counts only).  tests/test_host_logic.py */
int dotmi_plan_backsolve_tiles(int32_t nV, int32_t nT, const int32_t *T, const double *Xrest, const int32_t *epart, int32_t nParts,
                               int32_t p0, int32_t p1, int32_t cap, int32_t *tiles, int32_t *counts);
int dotmi_plan_layout(int32_t nV, int32_t nT, const int32_t *T, const double *Xrest, const int32_t *epart,
                      int32_t nParts, int32_t p0, int32_t p1, int32_t levels, int32_t min_split,
                      int32_t node_cap, int32_t *nodes, int32_t *n_nodes, int32_t *nmax, int32_t *pos);

/* Host-only (touches no device): the built-in element partitioner, for meshes that come without the partition METIS
 * gives the reference (METIS::partMesh, src/Utils/METIS.hpp:109-140; block-size runs `DOT -1 <nodes/block>`,
 * src/main.cpp:792-798).  Recursive bisection of the face-adjacency graph of the tets, coordinate split + Fiduccia-
 * Mattheyses refinement of the face cut; seedless and deterministic.  epart: nT values in [0,nParts). */
int dotmi_partition(int32_t nV, int32_t nT, const int32_t *T, const double *X, int32_t nParts, int32_t *epart);

/* world>1: rank 0 calls this and ships the 128 bytes to every rank (e.g. torch.distributed
 * broadcast); all ranks pass it as params.comm_id.  */
int dotmi_comm_unique_id(void *out128);
/* The number of ranks in the handle's RCCL communicator as RCCL itself reports it (ncclCommCount): what a benchmark prints
 * beside the number of processes it believes it started.  1: no communicator (single GPU); -world: the ranks cooperate
 * through the host all-reduce hook of dotmi_params instead of RCCL. */
int32_t dotmi_comm_ranks(const dotmi_handle *h);

/* ---- state ------------------------------------------------------------------------------------ */
/* (x, v[, x_n]) round-trip = what Optimizer::saveStatus / restart carry (Optimizer.cpp:1096-1177) */
int dotmi_set_state(dotmi_handle *h, const double *x, const double *v, const double *x_n /*or NULL*/);
int dotmi_get_state(dotmi_handle *h, double *x, double *v, double *x_tilde /*any may be NULL*/);
/* scripted handle motion for this step: x[idx[k]] = pos[k] (AnimScripter::stepAnimScript,
 * AnimScripter.cpp:456-466) */
int dotmi_set_dirichlet(dotmi_handle *h, int32_t n, const int32_t *idx, const double *pos);
/* fixed set changed (rubberBandPull release): re-pattern + refactor
 * (DOTTimeStepper::updatePrecondMtrAndFactorize, DOTTimeStepper.cpp:185-270) */
int dotmi_refix(dotmi_handle *h, const uint8_t *fixed);

/* ---- the hot path ----------------------------------------------------------------------------- */
/* One backward-Euler step = Optimizer::solve(1) minus the script move:
 * DOTTimeStepper::fullyImplicit (DOTTimeStepper.cpp:273-346) + BE update (Optimizer.cpp:354-361). */
int dotmi_step(dotmi_handle *h, dotmi_step_stats *stats);
/* per-iteration (alpha, E, ||g||^2) of the last step = iterStats.txt rows (DOTTimeStepper.cpp:304,329) */
int dotmi_last_iter_log(const dotmi_handle *h, int32_t cap, double *alpha, double *E, double *g2);
double dotmi_target_gres(const dotmi_handle *h); /* Optimizer::targetGRes (Optimizer.cpp:613-651) */

/* ---- kernel-level entry points (parity tests, adapters that override single hooks) ------------- */
int dotmi_eval_energy(dotmi_handle *h, const double *x, double *E);        /* Optimizer::computeEnergyVal, :1183 */
int dotmi_eval_gradient(dotmi_handle *h, const double *x, double *g);      /* Optimizer::computeGradient, :1220 */
int dotmi_eval_elem_hessians(dotmi_handle *h, const double *x, double *H); /* Energy::computeElemHessianByPK, Energy.cpp:673; nT*144 */
int dotmi_refactor(dotmi_handle *h, const double *x /*NULL = current*/);   /* DOTTimeStepper::updateHessianAndFactor, :349 */
int dotmi_apply_precond(dotmi_handle *h, const double *r, double *p);      /* DOTTimeStepper.cpp:406-450 */
int dotmi_spmv(dotmi_handle *h, const double *p, double *Hp);              /* LinSysSolver::multiply, CHOLMODSolver.cpp:185 */
/* One iteration of DOTTimeStepper::solve_oneStep up to its first line-search trial (DOTTimeStepper.cpp:386-467,
 * Optimizer::initStepSize :1076-1093, the first computeEnergyVal of lineSearch :791) from a caller-supplied
 * iterate x and m <= history stored pairs (S, Y: m*n each, oldest first): g = gradient at x, q = after the first
 * loop, z = the domain-decomposed block solve of q, p = search direction, alpha0, E(x + alpha0 p).  Any output may
 * be NULL.  Uses the handle's current factors and x~ and leaves its state alone (teacher-forced parity). */
int dotmi_probe_direction(dotmi_handle *h, const double *x, int32_t m, const double *S, const double *Y,
                          double *g, double *q, double *z, double *p, double *alpha0, double *E_trial);
/* derived mesh features, any pointer may be NULL: restTriInv nT*9 row-major, triArea nT, mass nV */
int dotmi_get_features(dotmi_handle *h, double *A, double *vol, double *mass);
/* dense principal sub-matrix R_s H R_s^T currently on the device (n_s = 3*local verts), row-major.
 * `inverse` != 0 returns the stored factor instead: X with H_s^-1 = X^T X (n_s x n_s, row-major).  On the
 * device X is the inverse Cholesky factor in the subdomain's nested-dissection order (lower triangular,
 * block-sparse); both matrices are returned in ascending-vertex order (l2g), i.e. X comes back as
 * P^T X_nd P.  l2g (local vertex -> global) may be NULL. */
int32_t dotmi_part_size(const dotmi_handle *h, int32_t part);
/* padded scalar size of the dense block every subdomain of this rank is stored in (the shared dissection layout: node
 * sizes are the maximum over the subdomains, rounded up to 64) -- what the factorisation's flop count is executed on */
int32_t dotmi_padded_size(const dotmi_handle *h);
/* bytes of HBM that hold the factors X_s of this rank's subdomains (compact 64-row blocks: ~1.2x the structural
 * non-zeros dotmi_step_stats.precond_bytes counts; DOTMI_TILE_FACTOR=0: nParts x padded_size^2 x 8) */
int64_t dotmi_factor_storage_bytes(const dotmi_handle *h);
/* which kernel family factorises this handle's subdomains (for measurement records): 1 = tile tasks, one launch per level
 * (tile_task_kernel), 2 = tile tasks as one dataflow launch (tile_flow_kernel), 3 = one launch pair per level: the diagonal
 * tasks in tile_task_kernel beside the product / row / inverse tasks on half tiles in tile_gemm_kernel (above 64 subdomains) */
int32_t dotmi_factor_kind(const dotmi_handle *h);
/* which form of the block solve this handle's factors are in: 0 = the explicit inverse X_s = chol(H_s)^-1 of every subdomain,
 * streamed in one pass (p_s = X_s^T X_s r_s); 1 = the two-level form (round 6; default where form 0 would stream 240 MB or more per application over all subdomains of the mesh,
 * DOTMI_TWO_LEVEL): the inverse factors of the dissection's leaves and of the separator complement, and between them the panels
 * L_GD X_DD of the factor itself -- the forward / backward substitution of CHOLMODSolver::solve (CHOLMODSolver.cpp:149-163) across
 * that one boundary.  dotmi_part_matrix(inverse = 1) is an error in form 1 (there is no explicit inverse of a whole subdomain) */
int32_t dotmi_backsolve_form(const dotmi_handle *h);
/* (host only) the form dotmi_create chooses on its own for this mesh and partition, and the bytes per application of form 0 on its own
 * layout (over all subdomains) that decide it; tests/test_host_logic.py pins the choice for the BASELINE workloads */
int dotmi_plan_backsolve_form(int32_t nV, int32_t nT, const int32_t *T, const double *Xrest, const int32_t *epart, int32_t nParts,
                              int32_t *form, int64_t *one_pass_bytes);
int dotmi_part_matrix(dotmi_handle *h, int32_t part, int inverse, double *M, int32_t *l2g);

/* ---- measurement ------------------------------------------------------------------------------ */
/* Launch the subdomain back-solve kernel `reps` times on the handle's stream between two HIP events
 * and return the average milliseconds per launch and the algorithmic bytes per launch. */
int dotmi_bench_precond(dotmi_handle *h, int32_t reps, double *ms_per_launch, int64_t *bytes_per_launch);
int dotmi_bench_energy(dotmi_handle *h, int32_t reps, double *ms_per_launch, int64_t *bytes_per_launch);
/* The same for every kernel class of the path (SURVEY.md section 8d "per kernel class"): `reps` back-to-back launches on
 * the handle's resident state between two HIP events on the library's stream; bytes = algorithmic bytes of one launch
 * (formulas: DESIGN.md section 4).  Call between steps. */
enum dotmi_bench_kind {
    DOTMI_BENCH_ELEM_ENERGY_GRAD = 0, /* element pass: energy + per-(patch, vertex) partial gradients */
    DOTMI_BENCH_ELEM_ENERGY = 1,      /* element pass, energy only (line-search retries) */
    DOTMI_BENCH_VERTEX_GATHER = 2,    /* vertex pass: gradient + new L-BFGS pair + its 21 statistics */
    DOTMI_BENCH_SPMV_DOTS = 3,        /* H p and the two dots of alpha_0 */
    DOTMI_BENCH_BACKSOLVE = 4,        /* subdomain back-solve (all kernels of one application) */
    DOTMI_BENCH_MERGE = 5,            /* sum over subdomains / dup (+ y_i . z) */
    DOTMI_BENCH_BUILD_QPAD = 6,       /* two-loop first half into the padded right-hand sides */
    DOTMI_BENCH_BUILD_P = 7,          /* two-loop second half */
    DOTMI_BENCH_STEP_FORWARD = 8,     /* x_trial = x + alpha p */
    DOTMI_BENCH_ELEM_HESSIAN = 9,     /* projected 12x12 element Hessians (once per step) */
    DOTMI_BENCH_ASSEMBLE = 10,        /* global block-CSR assembly (once per step) */
    /* the forms the device loop's early order launches (they read the loop state the last step left on the device; the
     * handle must have run a step): */
    DOTMI_BENCH_SPMV_ZP = 11,         /* build_p + SpMV + dots in one launch on cached H s_j */
    DOTMI_BENCH_MERGE_EARLY = 12,     /* tile partials -> u, M y_new, z, y_i . z */
    DOTMI_BENCH_ELEM_STEP = 13,       /* element pass that takes the line-search step itself */
    DOTMI_BENCH_GATHER_EARLY = 14,    /* vertex pass that also writes -g into the padded right-hand sides and H s_new */
    DOTMI_BENCH_DIRSTEP = 15,         /* (round 6) the speculative unit-step launch: direction kernel + element pass + trial point in one launch */
    DOTMI_BENCH_ELEM_VERTEX = 16,     /* (round 6) element pass + step + vertex gather of a trial in one launch on vertex patches */
    DOTMI_BENCH_COUNT = 17
};
int dotmi_bench_kernel(dotmi_handle *h, int32_t kind, int32_t reps, double *ms_per_launch, int64_t *bytes_per_launch);

#ifdef __cplusplus
}
#endif
#endif /* DOTMI_H */
