// k_dirstep.hip -- the speculative unit-step launch of the early order (round 6): the direction kernel and the element pass of the
// first line-search trial in ONE launch of two independent workgroup populations.
//
// Reference roles: DOTTimeStepper.cpp:455-467 (second half of the two-loop), Optimizer.cpp:1076-1093 (initStepSize) and the first
// evaluation of Optimizer::lineSearch (:791), Energy.cpp:294-423 / :910-972 (element energy and gradients).
//
// Three populations (workgroup index order = dispatch order): [0, NB_RED) the direction rows, [NB_RED, NB_RED + nbE) the patches,
// then ceil(nV / 256) small workgroups for the trial point and the inertia term of one vertex per thread (inertia_step_body: what
// element workgroup b does for vertices 256 b ... in the plain form -- kept apart so that the patches' workgroups do not hold the
// operands of p for a second vertex).  Meshes of at most 512 patches only (every patch a workgroup): beyond that the element
// pass is bound by throughput, its workgroups walk several patches with the next one's operands prefetched, and a second
// population on the same CUs takes away what the fusion gives (run_device_loop does not speculate there).
//
// Until round 5 a slot of the loop ran  spmv_zp -> element pass -> gather -> back-solve + controller -> merge:  the element pass
// waited for the direction kernel although it needs nothing of it but alpha_0 = clamp(-p.g / p.Hp, alphaMin, 1) -- and that clamp
// binds at 1 in most iterations (bar17K: 192 of 192; refined horse 82 %; stiff monkey 69 %).  Here workgroups [0, NB_RED) run
// spmv_zp_body (p, H p, the partials of p.g and p.Hp) and workgroups [NB_RED, NB_RED + nbE) run elem_patch_body in its SPEC form
// on x + 1 p, forming p_v = z_v + sum_j delta_j s_j[v] themselves for the vertices they read (k_elembody.hpp).  No population
// waits for the other; the gather scales the new pair with 1; the controller (loop_control_body<CTL_SPEC>, k_backsolve.hip) sums
// the SpMV partials as the element pass's prologue would have, and when alpha_0 is NOT 1 it stops the slot's back-solve and has
// the slot redone as the first trial with the true alpha_0 (phase 1, DevLoop::redo) -- so the sequence of evaluated trials, their
// energies, the accepted steps and every vector are the unspeculated loop's, bit for bit.  A step speculates when at least nine
// in ten first trials of the step before took the unit estimate (run_device_loop).
#include "k_device.hpp"
#include "k_dirbody.hpp"
#include "k_elembody.hpp"

namespace dotmi {

#ifdef DS_PROFILE
// per-workgroup wall-clock stamps of the launch (tools/prof_dirstep.sh): start, end, population
__device__ long long g_ds_prof[4096][3];
extern "C" int dotmi_debug_ds_prof(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ds_prof), sizeof(long long) * 3 * (size_t)n);
}
#define DS_STAMP(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_ds_prof[blockIdx.x][i] = (v); } while (0)
extern "C" int dotmi_debug_ds_sp_prof(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sp_prof), sizeof(long long) * 4 * (size_t)n);
}
extern "C" int dotmi_debug_ds_sv_prof(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sv_prof), sizeof(long long) * 6 * (size_t)n);
}
#ifdef EP_PROFILE
// (with -DEP_PROFILE as well: the phase stamps of the patches' workgroups, this unit's copy of g_ep_prof)
extern "C" int dotmi_debug_ds_ep_prof(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ep_prof), sizeof(long long) * 6 * (size_t)n);
}
#endif
#else
#define DS_STAMP(i, v) do { } while (0)
#endif

template <int MAT, int EPT>
__global__ __launch_bounds__(256, 2) void dirstep_kernel(DevPatches PT, const double *__restrict__ mass, const double *__restrict__ xt,
                                                      double dtSq, double *__restrict__ partE, const DevLoop *__restrict__ ctl,
                                                      StepArgs sa, int nV, const int *__restrict__ adj_ptr,
                                                      const int *__restrict__ adj_idx, const double *__restrict__ Hval,
                                                      const double *__restrict__ z, const double *__restrict__ c_partials,
                                                      double *__restrict__ p, double *__restrict__ Hp, double *__restrict__ partS, int nbE)
{
    extern __shared__ double lds[];
    __shared__ double sm[8];
    __shared__ double sh[1 + HIST_MAX];
    DS_STAMP(0, wall_clock64());
    if ((int)blockIdx.x < NB_RED) {
        // the direction rows (they return at once in a retry slot: ctl->phase != 0)
        DS_STAMP(2, 0);
        spmv_zp_body(nV, 0, nV, nullptr, nullptr, adj_ptr, adj_idx, Hval, z, c_partials, -NB_RED, p, Hp, partS, ctl, VList(), sm, sh + 1,
                     (int)blockIdx.x, NB_RED);
        DS_STAMP(1, wall_clock64());
        return;
    }
    const SpecArgs sx{z, c_partials, (int)gridDim.x - NB_RED - nbE};
    if ((int)blockIdx.x < NB_RED + nbE) {
        DS_STAMP(2, 1);
        elem_patch_body<MAT, true, EPT, true, false, false, true>(PT, mass, nullptr, xt, 0, nV, dtSq, partE, ctl, sa, sx,
                                                                   (int)blockIdx.x - NB_RED, nbE, lds, sm, sh);
        DS_STAMP(1, wall_clock64());
        return;
    }
    DS_STAMP(2, 2);
    inertia_step_body(mass, xt, nV, partE, ctl, sa, sx, (int)blockIdx.x - NB_RED - nbE, nbE, sm, sh);
    DS_STAMP(1, wall_clock64());
}

// can a step of this handle speculate?  (every patch a workgroup of the fused step: launch_elem_energy_grad's rule)
bool dirstep_fits(const DevPatches &PT)
{
    const int cap = PT.wgCap > 0 ? PT.wgCap : 512;
    return PT.nPatches <= cap && PT.nPatches <= ELEM_NB_MAX;
}

// the launch of a speculating step's slot in place of launch_spmv_zp + launch_elem_energy_grad (one rank, fused step)
void launch_dirstep(const DevMesh &M, const DevPatches &PT, int mat, double dtSq, const double *xt, double *partE, int *nblocks_out,
                    const double *Hval, const double *z, const double *c_partials, double *p, double *Hp, double *partS,
                    hipStream_t st, const DevLoop *ctl, const StepArgs &sa)
{
    // the element population: as many workgroups as launch_elem_energy_grad gives the fused step (the same grouping of the energy
    // partials, so the energy is the plain slot's bit for bit); every patch a workgroup (the caller has checked)
    int nb = PT.nPatches;
    const int nbv = (M.nV + 255) / 256;
    if (nb < nbv) nb = nbv;
    if (nb < 1) nb = 1;
    *nblocks_out = nb;
    const int ept = PT.PE / 256;
    const size_t shm = sizeof(double) * ((size_t)3 * PT.PV + (size_t)12 * PT.PE) + 2 * (size_t)((PT.PV + 1 + 3) & ~3) + 4 * (size_t)PT.PV;
#define DS_LAUNCH(MATV, EPTV)                                                                                                     \
    hipLaunchKernelGGL((dirstep_kernel<MATV, EPTV>), dim3(NB_RED + nb + nbv), dim3(256), shm, st, PT, M.mass, xt, dtSq, partE, ctl, sa, \
                       M.nV, M.adj_ptr, M.adj_idx, Hval, z, c_partials, p, Hp, partS, nb)
    if (mat == 0) {
        if (ept == 1) DS_LAUNCH(0, 1);
        else DS_LAUNCH(0, 2);
    } else {
        if (ept == 1) DS_LAUNCH(1, 1);
        else DS_LAUNCH(1, 2);
    }
#undef DS_LAUNCH
}

}  // namespace dotmi
