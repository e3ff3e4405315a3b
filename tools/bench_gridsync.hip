// bench_gridsync.hip -- what does a device-wide barrier cost on MI355X at the grid sizes of the L-BFGS loop kernels
// (256 / 512 workgroups of 256 threads), against the dependent kernel boundary it would replace?
// (VERDICT r01 "weak 4": DESIGN section 8a rejected a persistent loop kernel on an estimated barrier cost.)
//
//   boundary      K dependent launches of a kernel that touches one 24-byte record per thread
//   flat          one monotonic counter, lane-0 release fence -> arrive -> relaxed sc1 poll + s_sleep -> acquire fence
//   xcd           hierarchical: per-XCD counter, the last arriver of an XCD arrives at the top counter, every workgroup
//                 polls its XCD's generation word
// Every spin is bounded (a stuck barrier sets a flag and falls through) so a bug cannot hang the GPU.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bench_gridsync tools/bench_gridsync.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#define CK(x)                                                     \
    do {                                                          \
        auto e_ = (x);                                            \
        if ((int)e_ != 0) {                                       \
            printf("fail %s -> %d\n", #x, (int)e_);               \
            exit(1);                                              \
        }                                                         \
    } while (0)

constexpr long SPIN_LIMIT = 2000000;

struct Bar {
    unsigned top;          // monotonic arrivals (flat: workgroups, xcd: XCDs)
    unsigned pad0[31];
    unsigned xcnt[8][32];  // per-XCD arrivals (one 128-byte line each)
    unsigned xgen[8][32];  // per-XCD generation
    unsigned stuck;
    unsigned pad1[31];
    unsigned census_top;   // the census barrier of the xcd variant counts here
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void barrier_flat(Bar *b, unsigned *ctr, unsigned nwg, unsigned &epoch)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = epoch * nwg;
        long spins = 0;
        while ((int)(ld_relaxed(ctr) - target) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > SPIN_LIMIT) {
                b->stuck = 1;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}

// nx[x] = workgroups resident on XCD x (counted by a census pass at kernel start)
__device__ __forceinline__ void barrier_xcd(Bar *b, unsigned x, unsigned nx, unsigned nxcd, unsigned &epoch)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        ++epoch;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned prev = __hip_atomic_fetch_add(&b->xcnt[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        if (prev + 1 == epoch * nx) {
            // last of this XCD: arrive at the top, wait for every XCD, then open this XCD's generation
            __hip_atomic_fetch_add(&b->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * nxcd;
            while ((int)(ld_relaxed(&b->top) - target) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) {
                    b->stuck = 1;
                    break;
                }
            }
            __hip_atomic_store(&b->xgen[x][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while ((int)(ld_relaxed(&b->xgen[x][0]) - epoch) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > SPIN_LIMIT) {
                    b->stuck = 1;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// phase body: every thread rewrites one record and reads a record another workgroup wrote in the previous phase
__device__ __forceinline__ double body(double *buf, int n, int phase, double carry)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (i + 4099 * (phase + 1)) % n;   // some other workgroup's record
    const double v = buf[((phase + 1) & 1) * (size_t)n + j];
    buf[(phase & 1) * (size_t)n + i] = v + 1.0 + carry * 1e-30;
    return v;
}

__global__ __launch_bounds__(256) void k_boundary(double *buf, int n, int phase)
{
    body(buf, n, phase, 0.0);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_persistent(double *buf, int n, int phases, Bar *b, unsigned *census)
{
    __shared__ unsigned s_x, s_nx, s_nxcd;
    unsigned epoch = 0;
    if (MODE == 1) {
        if (threadIdx.x == 0) {
            s_x = xcc_id();
            atomicAdd(&census[s_x], 1u);
        }
        barrier_flat(b, &b->census_top, gridDim.x, epoch);   // census complete
        if (threadIdx.x == 0) {
            unsigned nx = 0, nxcd = 0;
            for (int x = 0; x < 8; ++x) {
                const unsigned c = ld_relaxed(&census[x]);
                if (x == (int)s_x) nx = c;
                nxcd += c > 0;
            }
            s_nx = nx;
            s_nxcd = nxcd;
        }
        __syncthreads();
        epoch = 0;
    }
    double carry = 0.0;
    for (int p = 0; p < phases; ++p) {
        carry = body(buf, n, p, carry);
        if (MODE == 0) barrier_flat(b, &b->top, gridDim.x, epoch);
        else barrier_xcd(b, s_x, s_nx, s_nxcd, epoch);
    }
}

int main()
{
    const int phases = 200;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int nwg : {256, 512, 1024}) {
        const int n = nwg * 256;
        double *buf;
        Bar *bar;
        unsigned *census;
        CK(hipMalloc(&buf, sizeof(double) * 2 * n));
        CK(hipMemset(buf, 0, sizeof(double) * 2 * n));
        CK(hipMalloc(&bar, sizeof(Bar)));
        CK(hipMalloc(&census, 64));
        float ms;
        // dependent launches
        for (int p = 0; p < 20; ++p) k_boundary<<<nwg, 256>>>(buf, n, p);
        CK(hipEventRecord(e0));
        for (int p = 0; p < phases; ++p) k_boundary<<<nwg, 256>>>(buf, n, p);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us_launch = 1e3 * ms / phases;
        double us[2] = {0, 0};
        unsigned stuck[2] = {0, 0};
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(bar, 0, sizeof(Bar)));
                CK(hipMemset(census, 0, 64));
                // the flat census barrier of mode 1 uses `top` as well: give it its own epoch space
                CK(hipEventRecord(e0));
                if (mode == 0) k_persistent<0><<<nwg, 256>>>(buf, n, phases, bar, census);
                else {
                    // census barrier counts on bar->top too; a second Bar keeps the two apart
                    k_persistent<1><<<nwg, 256>>>(buf, n, phases, bar, census);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                us[mode] = 1e3 * ms / phases;
                Bar hb;
                CK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
                stuck[mode] |= hb.stuck;
            }
        }
        // verify: after `phases` phases every record of the last written buffer equals `phases` (+ the 20 warm-ups)
        printf("workgroups %4d x 256 threads: dependent launch %.2f us/phase | flat barrier %.2f us/phase%s | "
               "xcd barrier %.2f us/phase%s\n",
               nwg, us_launch, us[0], stuck[0] ? " (STUCK)" : "", us[1], stuck[1] ? " (STUCK)" : "");
        CK(hipFree(buf));
        CK(hipFree(bar));
        CK(hipFree(census));
    }
    return 0;
}
