// k_element.hip -- element pass of the L-BFGS loop (Energy.cpp:294-423, :910-972; Optimizer.cpp:1202-1215)
// (one translation unit per kernel family since round 6: an edit to one family no longer moves the register allocation and
// scalar loads of the others; every unit is compiled once.  Conventions and the reference map: k_device.hpp)
#include "k_device.hpp"
#include "k_elembody.hpp"

namespace dotmi {

#ifdef EP_PROFILE
extern "C" int dotmi_debug_ep_prof(long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ep_prof), sizeof(long long) * 6 * (size_t)n);
}
#endif

// the element pass as a launch of its own (the body: k_elembody.hpp)
template <int MAT, bool GRAD, int EPT, bool FUSE, bool PIPE, bool PAIR>
__global__ __launch_bounds__(256) void elem_patch_kernel(DevPatches PT, const double *__restrict__ mass,
                                                         const double *__restrict__ x, const double *__restrict__ xt,
                                                         int v0, int v1, double dtSq, double *__restrict__ partials,
                                                         const DevLoop *__restrict__ ctl, StepArgs sa)
{
    extern __shared__ double lds[];
    __shared__ double sm[8];
    __shared__ double sh[1];
    elem_patch_body<MAT, GRAD, EPT, FUSE, PIPE, PAIR, false>(PT, mass, x, xt, v0, v1, dtSq, partials, ctl, sa, SpecArgs{nullptr, nullptr},
                                                             (int)blockIdx.x, (int)gridDim.x, lds, sm, sh);
}

template <bool PAIR>
static void launch_elem_impl(const DevMesh &M, const DevPatches &PT, int mat, double dtSq, const double *x, const double *xt,
                             int v0, int v1, int grad, double *partials, int *nblocks_out, hipStream_t st, const DevLoop *ctl,
                             const StepArgs *step)
{
    StepArgs sa{nullptr, nullptr, nullptr, 0.0};
    if (step && ctl) sa = *step;
    const bool paired = PAIR && sa.p && sa.alpha_min < 0.0 && grad;
    // at most elem_wg_cap() workgroups take the patches (as many as are resident at once): beyond that a workgroup walks several
    // patches and prefetches the next one's operands (elem_patch_kernel)
    // (the instantiation with the step inside: two per CU; a handle whose loop uses it fixes 512 for all of them, PT.wgCap)
    const int cap = PT.wgCap > 0 ? PT.wgCap : ((step && ctl) ? 512 : elem_wg_cap(mat));
    const bool pipe = PT.nPatches > cap;
    int nb = pipe ? cap : PT.nPatches;
    const int nbv = (v1 - v0 + 255) / 256;
    if (nb < nbv && !pipe) nb = nbv;    // the inertia loop likes one vertex per thread on small meshes
    if (nb > ELEM_NB_MAX) nb = ELEM_NB_MAX;
    if (nb < 1) nb = 1;
    *nblocks_out = nb;
    const int nbLaunch = paired ? 2 * nb : nb;
    const int ept = PT.PE / 256;
    const size_t shm = sizeof(double) * ((size_t)3 * PT.PV + (grad ? (size_t)12 * PT.PE : 0)) +
                       (grad ? 2 * (size_t)((PT.PV + 1 + 3) & ~3) + 4 * (size_t)PT.PV : 0);
    // (the paired instantiations exist for the fused step with gradients only: what a paired step launches)
#define DM_LAUNCH(MATV, GRADV, EPTV)                                                                            \
    do {                                                                                                            \
        if (sa.p && pipe)                                                                                           \
            hipLaunchKernelGGL((elem_patch_kernel<MATV, GRADV, EPTV, true, true, PAIR && GRADV>), dim3(nbLaunch), dim3(256), shm, st, \
                               PT, M.mass, x, xt, v0, v1, dtSq, partials, ctl, sa);                                 \
        else if (sa.p)                                                                                              \
            hipLaunchKernelGGL((elem_patch_kernel<MATV, GRADV, EPTV, true, false, PAIR && GRADV>), dim3(nbLaunch), dim3(256), shm, st, \
                               PT, M.mass, x, xt, v0, v1, dtSq, partials, ctl, sa);                                 \
        else if (pipe)                                                                                              \
            hipLaunchKernelGGL((elem_patch_kernel<MATV, GRADV, EPTV, false, true, false>), dim3(nbLaunch), dim3(256), shm, st, PT, \
                               M.mass, x, xt, v0, v1, dtSq, partials, ctl, sa);                                     \
        else                                                                                                        \
            hipLaunchKernelGGL((elem_patch_kernel<MATV, GRADV, EPTV, false, false, false>), dim3(nbLaunch), dim3(256), shm, st, PT, \
                               M.mass, x, xt, v0, v1, dtSq, partials, ctl, sa);                                     \
    } while (0)
#define DM_LAUNCH_E(MATV, GRADV)      \
    do {                              \
        if (ept == 1) DM_LAUNCH(MATV, GRADV, 1); \
        else DM_LAUNCH(MATV, GRADV, 2);          \
    } while (0)
    if (mat == 0) {
        if (grad) DM_LAUNCH_E(0, true);
        else DM_LAUNCH_E(0, false);
    } else {
        if (grad) DM_LAUNCH_E(1, true);
        else DM_LAUNCH_E(1, false);
    }
#undef DM_LAUNCH_E
#undef DM_LAUNCH
}

void launch_elem_energy_grad(const DevMesh &M, const DevPatches &PT, int mat, double dtSq, const double *x,
                             const double *xt, int v0, int v1, int grad, double *partials, int *nblocks_out,
                             hipStream_t st, const DevLoop *ctl, const StepArgs *step)
{
    launch_elem_impl<false>(M, PT, mat, dtSq, x, xt, v0, v1, grad, partials, nblocks_out, st, ctl, step);
}
// a step with paired trials (StepArgs::alpha_min < 0 makes it a paired launch)
void launch_elem_energy_grad_pair(const DevMesh &M, const DevPatches &PT, int mat, double dtSq, const double *x,
                                  const double *xt, int v0, int v1, int grad, double *partials, int *nblocks_out,
                                  hipStream_t st, const DevLoop *ctl, const StepArgs *step)
{
    launch_elem_impl<true>(M, PT, mat, dtSq, x, xt, v0, v1, grad, partials, nblocks_out, st, ctl, step);
}

}  // namespace dotmi
