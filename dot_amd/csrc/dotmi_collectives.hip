// dotmi_collectives.hip -- sum over the ranks (RCCL, or the host hook of dotmi_params) and the owner exchange's packets
#include "dotmi_handle.hpp"

namespace dotmi {

__global__ void reduce_rows_kernel(const double *__restrict__ partials, int nblocks, int stride, int K,
                                   double s0, double s1, int combine, double *__restrict__ out)
{
    // single wave; out[j] = sum_b partials[b*stride+j]; combine: out[0] = s0*sum0 + s1*sum1
    const int lane = threadIdx.x;
    double first = 0.0;
    for (int j = 0; j < K; ++j) {
        double acc = 0.0;
        for (int b = lane; b < nblocks; b += 64) acc += partials[(size_t)b * stride + j];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) {
            if (combine) {
                if (j == 0) first = s0 * acc;
                else if (j == 1) out[0] = first + s1 * acc;
            } else {
                out[j] = acc;
            }
        }
    }
}

}  // namespace dotmi

namespace dotmi {

// sum over the ranks of n doubles at `dev`, in place, ordered on the handle's stream: RCCL, or the host hook
int allreduce_sum(dotmi_handle *h, double *dev, size_t n)
{
    h->arCallsStep++;
    h->arBytesStep += 8.0 * (double)n;
    if (h->comm) {
        const bool timed = !h->evAr.empty() && h->evArUsed + 2 <= (int)h->evAr.size() && (h->arCount++ % h->timeStride) == 0;
        if (timed) HIPCHECK(h, hipEventRecord(h->evAr[h->evArUsed], h->st));
        NCCLCHECK(h, ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, h->comm, h->st));
        if (timed) {
            HIPCHECK(h, hipEventRecord(h->evAr[h->evArUsed + 1], h->st));
            h->arTimedBytes.push_back(8 * n);
            h->evArUsed += 2;
        }
        return 0;
    }
    if (!h->arCb) return 0;   // single rank without a communicator (cannot happen on the sharded path)
    if (n > h->arCap) {
        if (h->arStage) hipHostFree(h->arStage);
        h->arStage = nullptr;
        HIPCHECK(h, hipHostMalloc((void **)&h->arStage, sizeof(double) * n));
        h->arCap = n;
    }
    HIPCHECK(h, hipMemcpyAsync(h->arStage, dev, sizeof(double) * n, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    h->arCb(h->arCtx, h->arStage, (int64_t)n);
    HIPCHECK(h, hipMemcpyAsync(dev, h->arStage, sizeof(double) * n, hipMemcpyHostToDevice, h->st));
    return 0;
}

// Every rank takes its accept / halve / converged decisions from RANK 0's control scalars (the ranks
// compute them redundantly on replicated data, but a single differing bit would make them branch apart and dead-lock
// in the next collective).  vals: host array, replaced by rank 0's on every rank.  One small collective per trial.
int adopt_rank0(dotmi_handle *h, double *vals, int n)
{
    if (h->world <= 1) return 0;
    if (h->rank != 0)
        for (int i = 0; i < n; ++i) vals[i] = 0.0;
    HIPCHECK(h, hipMemcpyAsync(h->ctrlDev, vals, sizeof(double) * n, hipMemcpyHostToDevice, h->st));
    if (int rc = allreduce_sum(h, h->ctrlDev, n)) return rc;   // x + 0 + ... + 0 is exact: a broadcast
    HIPCHECK(h, hipMemcpyAsync(vals, h->ctrlDev, sizeof(double) * n, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    return 0;
}

// owner exchange: the entries of the vertices held by more than one rank (and `ntail` scalars behind them) summed over the ranks
int exchange_iface(dotmi_handle *h, double *vec, double *tailp, int ntail)
{
    if (3 * h->nIface + ntail == 0) return 0;   // (no vertex is shared -- one rank --: the same on every rank, nothing to send)
    launch_pack_iface(h->nIface, h->ifaceIdx, vec, h->xpack, tailp, ntail, h->st);
    if (int rc = allreduce_sum(h, h->xpack, (size_t)3 * h->nIface + ntail)) return rc;
    launch_unpack_iface(h->nIface, h->ifaceIdx, h->xpack, h->heldMask, vec, tailp, ntail, h->st);
    return 0;
}

// owner exchange, packed form.  The gradient's packet: [3 nIface entries | E | ncols statistics]; `partials` holds the sums this
// rank took BEFORE the exchange (pair_stats with pre = 1, or the |g|^2 of the vertices only it holds at the start of a step:
// ncols = 1) and partE the element pass' energy partials -- both are summed by the pack's workgroup 0 straight into the tail.
// The summed statistics land in row 0 of partGR with the shared entries' squares added to |g|^2, E in gstage[n + 1].
int exchange_gradient_packed(dotmi_handle *h, int n, int nbE, const double *partials, int ncols)
{
    const int n3 = 3 * h->nIface;
    PackRed rE{h->partE, nbE, 2, 2, 1, n3, h->dtSq, 1.0};
    PackRed rS{partials, NB_RED, RED_K, ncols, 0, n3 + 1, 0.0, 0.0};
    launch_pack_iface(h->nIface, h->ifaceIdx, h->gstage, h->xpack, nullptr, 0, h->st, &rE, &rS);
    if (int rc = allreduce_sum(h, h->xpack, (size_t)n3 + 1 + ncols)) return rc;
    launch_unpack_iface(h->nIface, h->ifaceIdx, h->xpack, h->heldMask, h->gstage, h->gstage + n + 1, 1, h->st, h->partGR, ncols);
    return 0;
}

// The merged back-solve's packet: [3 nIface entries | HIST_MAX sums y_i . z] (merge_early with pre = 1 left this rank's share
// in partC); the sums land in row 0 of partGC.
int exchange_solve_packed(dotmi_handle *h)
{
    const int n3 = 3 * h->nIface;
    PackRed rC{h->partC, NB_RED, RED_K, HIST_MAX, 0, n3, 0.0, 0.0};
    launch_pack_iface(h->nIface, h->ifaceIdx, h->zstage, h->xpack, nullptr, 0, h->st, &rC);
    if (int rc = allreduce_sum(h, h->xpack, (size_t)n3 + HIST_MAX)) return rc;
    launch_unpack_iface(h->nIface, h->ifaceIdx, h->xpack, h->heldMask, h->zstage, h->partGC, HIST_MAX, h->st);
    return 0;
}

}  // namespace dotmi

