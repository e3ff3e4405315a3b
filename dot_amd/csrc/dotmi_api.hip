// dotmi_api.hip -- the remaining entry points of include/dotmi.h: state, scripted handles, kernel-level calls for parity tests, probes, measurement
#include "dotmi_handle.hpp"

extern "C" {

int dotmi_set_state(dotmi_handle *h, const double *x, const double *v, const double *x_n)
{
    if (!h || !x || !v) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    const size_t bytes = sizeof(double) * h->n;
    HIPCHECK(h, hipMemcpyAsync(h->x, x, bytes, hipMemcpyHostToDevice, h->st));
    HIPCHECK(h, hipMemcpyAsync(h->v, v, bytes, hipMemcpyHostToDevice, h->st));
    HIPCHECK(h, hipMemcpyAsync(h->xn, x_n ? x_n : x, bytes, hipMemcpyHostToDevice, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    // x~ = x_n + dt v + dt^2 g on free vertices, x_n on fixed ones (Optimizer.cpp:585-610)
    std::vector<double> xt(h->n);
    const double *xn_h = x_n ? x_n : x;
    for (int i = 0; i < h->nV; ++i)
        for (int d = 0; d < 3; ++d) {
            const int k = 3 * i + d;
            xt[k] = h->fixed[i] ? xn_h[k] : xn_h[k] + (v[k] * h->dt + h->gdtsq[d]);
        }
    HIPCHECK(h, hipMemcpy(h->xt, xt.data(), bytes, hipMemcpyHostToDevice));
    return 0;
}

int dotmi_get_state(dotmi_handle *h, double *x, double *v, double *x_tilde)
{
    if (!h) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    const size_t bytes = sizeof(double) * h->n;
    HIPCHECK(h, hipStreamSynchronize(h->st));
    if (x) HIPCHECK(h, hipMemcpy(x, h->x, bytes, hipMemcpyDeviceToHost));
    if (v) HIPCHECK(h, hipMemcpy(v, h->v, bytes, hipMemcpyDeviceToHost));
    if (x_tilde) HIPCHECK(h, hipMemcpy(x_tilde, h->xt, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int dotmi_set_dirichlet(dotmi_handle *h, int32_t n, const int32_t *idx, const double *pos)
{
    if (!h || n < 0 || (n > 0 && (!idx || !pos))) return DOTMI_E_INVALID;
    if (n == 0) return 0;
    HIPCHECK(h, hipSetDevice(h->device));
    for (int i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= h->nV) {
            h->err = "dirichlet index out of range";
            return DOTMI_E_INVALID;
        }
    if ((size_t)n > h->dcap) {
        if (int rc = dalloc(h, &h->didx, (size_t)n)) return rc;
        if (int rc = dalloc(h, &h->dpos, (size_t)3 * n)) return rc;
        h->dcap = n;
    }
    // the scripted set is the same every step: the indices go up only when they change, the positions through a
    // pinned staging buffer, and nothing waits here (the stream orders the scatter before the step's kernels)
    if (h->didxHost.size() != (size_t)n || memcmp(h->didxHost.data(), idx, sizeof(int32_t) * n) != 0) {
        HIPCHECK(h, hipStreamSynchronize(h->st));
        h->didxHost.assign(idx, idx + n);
        if (h->dposPinned) hipHostFree(h->dposPinned);   // the stream is idle: nothing reads the staging buffer
        h->dposPinned = nullptr;
        HIPCHECK(h, hipHostMalloc((void **)&h->dposPinned, sizeof(double) * 3 * n));
        HIPCHECK(h, hipMemcpyAsync(h->didx, h->didxHost.data(), sizeof(int) * n, hipMemcpyHostToDevice, h->st));
    }
    if (!h->evDir) HIPCHECK(h, hipEventCreateWithFlags(&h->evDir, hipEventDisableTiming));
    else HIPCHECK(h, hipEventSynchronize(h->evDir));  // the previous upload out of the staging buffer (normally long done)
    memcpy(h->dposPinned, pos, sizeof(double) * 3 * n);
    HIPCHECK(h, hipMemcpyAsync(h->dpos, h->dposPinned, sizeof(double) * 3 * n, hipMemcpyHostToDevice, h->st));
    HIPCHECK(h, hipEventRecord(h->evDir, h->st));
    launch_scatter_rows(n, h->didx, h->dpos, h->x, h->st);
    return 0;
}

int dotmi_refix(dotmi_handle *h, const uint8_t *fixed)
{
    if (!h || !fixed) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    h->fixed.assign(fixed, fixed + h->nV);
    HIPCHECK(h, hipMemcpy(h->M.fixed, fixed, h->nV, hipMemcpyHostToDevice));
    return refactor(h, h->x, nullptr, nullptr);
}

double dotmi_target_gres(const dotmi_handle *h) { return h ? h->targetGRes : 0.0; }

int dotmi_last_iter_log(const dotmi_handle *h, int32_t cap, double *alpha, double *E, double *g2)
{
    if (!h) return DOTMI_E_INVALID;
    if (h->logPending > 0) {
        dotmi_handle *hm = const_cast<dotmi_handle *>(h);
        const int nlog = h->logPending;
        hm->logPending = 0;
        hm->log_alpha.resize(nlog);
        hm->log_E.resize(nlog);
        hm->log_g2.resize(nlog);
        if (hipSetDevice(h->device) != hipSuccess ||
            hipMemcpy(hm->log_alpha.data(), h->dlog, sizeof(double) * nlog, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hm->log_E.data(), h->dlog + h->logCap, sizeof(double) * nlog, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(hm->log_g2.data(), h->dlog + 2 * (size_t)h->logCap, sizeof(double) * nlog, hipMemcpyDeviceToHost) !=
                hipSuccess)
            return DOTMI_E_DEVICE;
    }
    const int n = std::min<int>(cap, (int)h->log_alpha.size());
    for (int i = 0; i < n; ++i) {
        if (alpha) alpha[i] = h->log_alpha[i];
        if (E) E[i] = h->log_E[i];
        if (g2) g2[i] = h->log_g2[i];
    }
    return (int)h->log_alpha.size();
}

// ---- kernel-level entry points ------------------------------------------------------------------
static int upload_tmp(dotmi_handle *h, const double *x, double *dst)
{
    HIPCHECK(h, hipMemcpyAsync(dst, x, sizeof(double) * h->n, hipMemcpyHostToDevice, h->st));
    return 0;
}

int dotmi_eval_energy(dotmi_handle *h, const double *x, double *E)
{
    if (!h || !x || !E) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    if (int rc = upload_tmp(h, x, h->x_trial)) return rc;
    int nb = 0;
    launch_elem_energy_grad(h->M, h->PTall, h->mat, h->dtSq, h->x_trial, h->xt, 0, h->nV, 0, h->partE,
                            &nb, h->st);
    HIPCHECK(h, hipMemcpyAsync(h->h_partE, h->partE, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    double se = 0, si = 0;
    for (int b = 0; b < nb; ++b) {
        se += h->h_partE[2 * b];
        si += h->h_partE[2 * b + 1];
    }
    *E = h->dtSq * se + si;
    return 0;
}

int dotmi_eval_gradient(dotmi_handle *h, const double *x, double *g)
{
    if (!h || !x || !g) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    if (int rc = upload_tmp(h, x, h->x_trial)) return rc;
    int nb = 0;
    launch_elem_energy_grad(h->M, h->PTall, h->mat, h->dtSq, h->x_trial, h->xt, 0, h->nV, 1, h->partE,
                            &nb, h->st);
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.x = h->x_trial;
    a.xt = h->xt;
    a.g_new = h->g_trial;
    a.make_pair = 0;
    a.iv0 = 0;
    a.iv1 = h->nV;
    LbfgsArgs L;
    memset(&L, 0, sizeof(L));
    launch_vertex_gather(h->M, h->PTall, a, L, h->partR, h->st);
    HIPCHECK(h, hipMemcpyAsync(g, h->g_trial, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    return 0;
}

int dotmi_eval_elem_hessians(dotmi_handle *h, const double *x, double *H)
{
    if (!h || !x || !H) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    if (int rc = upload_tmp(h, x, h->x_trial)) return rc;
    // scratch copy so the resident He (state of the current preconditioner) is not disturbed
    double *tmp = nullptr;
    HIPCHECK(h, hipMalloc((void **)&tmp, sizeof(double) * 144 * (size_t)h->nT));
    launch_elem_hessians(h->M, h->mat, h->dtSq, h->x_trial, tmp, h->st);
    hipError_t e = hipMemcpyAsync(H, tmp, sizeof(double) * 144 * (size_t)h->nT, hipMemcpyDeviceToHost, h->st);
    hipStreamSynchronize(h->st);
    hipFree(tmp);
    HIPCHECK(h, e);
    return 0;
}

int dotmi_refactor(dotmi_handle *h, const double *x)
{
    if (!h) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    const double *xd = h->x;
    if (x) {
        if (int rc = upload_tmp(h, x, h->x_trial)) return rc;
        xd = h->x_trial;
    }
    return refactor(h, xd, nullptr, nullptr);
}

int dotmi_apply_precond(dotmi_handle *h, const double *r, double *p)
{
    if (!h || !r || !p) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (int rc = enter_with_factors(h)) return rc;
    if (int rc = upload_tmp(h, r, h->q)) return rc;
    LbfgsArgs L;
    memset(&L, 0, sizeof(L));
    if (int rc = apply_precond(h, h->q, h->z, L)) return rc;
    HIPCHECK(h, hipMemcpyAsync(p, h->z, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    return 0;
}

// One L-BFGS-H direction and its first line-search trial from a caller-supplied iterate and history, with the
// kernels of the host-driven loop (DOTTimeStepper.cpp:386-467, Optimizer.cpp:1076-1093, :791).  Teacher forcing
// (SURVEY.md 8(c) F4): a test feeds the oracle's (x, history) of iteration k and compares q, z, p, alpha_0 and
// E(x + alpha_0 p).  Uses the current factors and x~; the L-BFGS slots it overwrites are reset by the next step.
int dotmi_probe_direction(dotmi_handle *h, const double *x, int32_t m, const double *S, const double *Y, double *g_out,
                          double *q_out, double *z_out, double *p_out, double *alpha0, double *E_trial)
{
    if (!h || !x || m < 0 || m > h->hist || (m > 0 && (!S || !Y))) return DOTMI_E_INVALID;
    if (h->dist) {
        h->err = "dotmi_probe_direction: single-GPU handles only";
        return DOTMI_E_INVALID;
    }
    HIPCHECK(h, hipSetDevice(h->device));
    if (int rc = enter_with_factors(h)) return rc;
    const int n = h->n;
    const size_t bytes = sizeof(double) * n;
    // the probe works on tmpn (iterate), g_trial (gradient), x_trial (trial point): the handle's own x, g stay
    HIPCHECK(h, hipMemcpyAsync(h->tmpn, x, bytes, hipMemcpyHostToDevice, h->st));
    for (int i = 0; i < m; ++i) {
        HIPCHECK(h, hipMemcpyAsync(h->S[i], S + (size_t)i * n, bytes, hipMemcpyHostToDevice, h->st));
        HIPCHECK(h, hipMemcpyAsync(h->Y[i], Y + (size_t)i * n, bytes, hipMemcpyHostToDevice, h->st));
    }
    int nb = 0;
    launch_elem_energy_grad(h->M, h->PTall, h->mat, h->dtSq, h->tmpn, h->xt, 0, h->nV, 1, h->partE, &nb,
                            h->st);
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.x = h->tmpn;
    a.xt = h->xt;
    a.g_new = h->g_trial;
    a.iv0 = 0;
    a.iv1 = h->nV;
    LbfgsArgs L;
    memset(&L, 0, sizeof(L));
    launch_vertex_gather(h->M, h->PTall, a, L, h->partR, h->st);
    std::vector<double> g(n);
    HIPCHECK(h, hipMemcpyAsync(g.data(), h->g_trial, bytes, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    // Gram matrix and the first half of the two-loop on the host (the running loop gets the same numbers from the
    // gather kernel's partial sums)
    L.m = m;
    double b[HIST_MAX] = {0}, xi[HIST_MAX] = {0};
    auto dot = [&](const double *u, const double *v) {
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc += u[k] * v[k];
        return acc;
    };
    for (int i = 0; i < m; ++i) {
        L.s[i] = h->S[i];
        L.y[i] = h->Y[i];
        b[i] = dot(S + (size_t)i * n, g.data());
        for (int j = 0; j < m; ++j) L.sy[i][j] = dot(S + (size_t)i * n, Y + (size_t)j * n);
        L.ys[i] = L.sy[i][i];
    }
    for (int i = m - 1; i >= 0; --i) {
        double sq = -b[i];
        for (int j = m - 1; j > i; --j) sq -= xi[j] * L.sy[i][j];
        xi[i] = sq / L.ys[i];
    }
    launch_build_q(n, h->g_trial, L, xi, h->q, h->st);
    if (int rc = apply_precond(h, h->q, h->z, L)) return rc;
    launch_build_p(n, h->z, L, h->partC, xi, h->p, h->st);
    launch_spmv_dots(h->M, h->Hval, h->p, h->g_trial, nullptr, 0, h->nV, h->partS, h->st);
    launch_step_forward(n, h->tmpn, h->p, h->x_trial, h->partS, 0.0, 1, h->alphaMin, h->alpha_dev, h->h_alpha, h->st);
    launch_elem_energy_grad(h->M, h->PTall, h->mat, h->dtSq, h->x_trial, h->xt, 0, h->nV, 0, h->partE, &nb,
                            h->st);
    HIPCHECK(h, hipMemcpyAsync(h->h_partE, h->partE, sizeof(double) * 2 * nb, hipMemcpyDeviceToHost, h->st));
    if (g_out) memcpy(g_out, g.data(), bytes);
    if (q_out) HIPCHECK(h, hipMemcpyAsync(q_out, h->q, bytes, hipMemcpyDeviceToHost, h->st));
    if (z_out) HIPCHECK(h, hipMemcpyAsync(z_out, h->z, bytes, hipMemcpyDeviceToHost, h->st));
    if (p_out) HIPCHECK(h, hipMemcpyAsync(p_out, h->p, bytes, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    if (alpha0) *alpha0 = h->h_alpha[0];
    if (E_trial) {
        const double se = chunked_sum(nb, [&](int k) { return h->h_partE[2 * k]; });
        const double si = chunked_sum(nb, [&](int k) { return h->h_partE[2 * k + 1]; });
        *E_trial = h->dtSq * se + si;
    }
    return 0;
}

int dotmi_spmv(dotmi_handle *h, const double *p, double *Hp)
{
    if (!h || !p || !Hp) return DOTMI_E_INVALID;
    if (h->shardHess) {
        h->err = "dotmi_spmv: the rows of the global Hessian are sharded over the ranks on this handle (DOTMI_SHARD_HESS=0 keeps them replicated)";
        return DOTMI_E_INVALID;
    }
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    if (int rc = upload_tmp(h, p, h->tmpn)) return rc;
    launch_spmv_dots(h->M, h->Hval, h->tmpn, nullptr, h->Hp, 0, h->nV, h->partS, h->st);
    HIPCHECK(h, hipMemcpyAsync(Hp, h->Hp, sizeof(double) * h->n, hipMemcpyDeviceToHost, h->st));
    HIPCHECK(h, hipStreamSynchronize(h->st));
    return 0;
}

int dotmi_get_features(dotmi_handle *h, double *A, double *vol, double *mass)
{
    if (!h) return DOTMI_E_INVALID;
    if (A) memcpy(A, h->A.data(), sizeof(double) * h->A.size());
    if (vol) memcpy(vol, h->vol.data(), sizeof(double) * h->vol.size());
    if (mass) memcpy(mass, h->mass.data(), sizeof(double) * h->mass.size());
    return 0;
}

int32_t dotmi_part_size(const dotmi_handle *h, int32_t part)
{
    if (!h || part < 0 || part >= h->nPartsAll) return DOTMI_E_INVALID;
    return 3 * (int32_t)h->partVerts[part].size();
}

int32_t dotmi_padded_size(const dotmi_handle *h) { return h ? h->P.nmax : DOTMI_E_INVALID; }

int64_t dotmi_factor_storage_bytes(const dotmi_handle *h) { return h ? (int64_t)(8 * h->wTotal) : DOTMI_E_INVALID; }

int dotmi_part_matrix(dotmi_handle *h, int32_t part, int inverse, double *Mout, int32_t *l2g)
{
    if (!h || part < h->p0 || part >= h->p1 || !Mout) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (inverse && h->twoLevel) {
        h->err = "the factors are in the two-level form (DOTMI_TWO_LEVEL): there is no explicit inverse of a whole subdomain to return";
        return DOTMI_E_INVALID;
    }
    if (inverse) {
        if (int rc = enter_with_factors(h)) return rc;
    } else if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    const int ls = part - h->p0;
    const int ns = 3 * (int)h->partVerts[part].size();
    const int nmax = h->P.nmax, ntl = nmax / 64;
    double *W = h->P.W;
    double *tmp = nullptr;
    if (!inverse) {
        // rebuild H_s from the resident block-CSR into a scratch copy of the factor storage
        HIPCHECK(h, hipMalloc((void **)&tmp, sizeof(double) * std::max<size_t>(h->wTotal, 64)));
        DevParts Pt = h->P;
        Pt.W = tmp;
        HIPCHECK(h, hipMemsetAsync(tmp, 0, sizeof(double) * h->wTotal, h->st));
        launch_dense_fill(Pt, h->Hval, h->st);
        W = tmp;
    }
    // the subdomain's row blocks (RowTile) lie one after the other in W: copy their span, then read (row, column) through
    // the table -- back into ascending vertex order
    long long lo = -1, hi = -1;
    for (int J = 0; J < ntl; ++J) {
        const long long o = h->rtOff[(size_t)ls * ntl + J];
        if (o < 0) continue;
        if (lo < 0) lo = o;
        hi = o + 64ll * h->rtLd[(size_t)ls * ntl + J];
    }
    // (two-level form: the separators' row blocks hold their sub-tree's leaf columns in a second range)
    long long lo2 = -1, hi2 = -1;
    if (h->twoLevel)
        for (int J = 0; J < ntl; ++J) {
            const long long o = h->rtOffM[(size_t)ls * ntl + J];
            if (o < 0) continue;
            if (lo2 < 0) lo2 = o;
            hi2 = o + 64ll * h->rtLdM[(size_t)ls * ntl + J];
        }
    std::vector<double> span((size_t)std::max<long long>(hi - lo, 1)), span2((size_t)std::max<long long>(hi2 - lo2, 1));
    hipError_t e = lo >= 0 ? hipMemcpyAsync(span.data(), W + lo, sizeof(double) * (size_t)(hi - lo), hipMemcpyDeviceToHost, h->st)
                           : hipSuccess;
    if (e == hipSuccess && lo2 >= 0)
        e = hipMemcpyAsync(span2.data(), W + lo2, sizeof(double) * (size_t)(hi2 - lo2), hipMemcpyDeviceToHost, h->st);
    hipStreamSynchronize(h->st);
    if (tmp) hipFree(tmp);
    HIPCHECK(h, e);
    auto at = [&](int r, int c) -> double {   // memory row r, column c; what is not stored is zero
        const size_t k = (size_t)ls * ntl + (r >> 6);
        const long long o = h->rtOff[k];
        const int c0 = h->rtC0[k], ld = h->rtLd[k];
        if (o >= 0 && c < c0 && lo2 >= 0 && h->rtOffM[k] >= 0 && c >= h->rtC0M[k] && c < h->rtC0M[k] + h->rtLdM[k])
            return span2[(size_t)(h->rtOffM[k] - lo2 + (long long)(r & 63) * h->rtLdM[k] + (c - h->rtC0M[k]))];
        if (o < 0 || c < c0 || c >= c0 + ld) return 0.0;
        return span[(size_t)(o - lo + (long long)(r & 63) * ld + (c - c0))];
    };
    const auto &pos = h->partPos[ls];
    for (int i = 0; i < ns; ++i)
        for (int j = 0; j < ns; ++j) {
            // memory row r holds row r of X up to the diagonal; the other triangle is not part of X (the tile
            // factorisation leaves the mirror copy of H there, the compact layout does not even store it) -- and H_s
            // itself is read symmetrically from the stored triangle
            const int r = pos[i / 3] + i % 3, c = pos[j / 3] + j % 3;
            Mout[(size_t)i * ns + j] = inverse ? (c <= r ? at(r, c) : 0.0) : at(std::max(r, c), std::min(r, c));
        }
    if (l2g)
        for (size_t i = 0; i < h->partVerts[part].size(); ++i) l2g[i] = h->partVerts[part][i];
    return 0;
}

int dotmi_bench_precond(dotmi_handle *h, int32_t reps, double *ms_per_launch, int64_t *bytes_per_launch)
{
    if (!h || reps < 1) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (int rc = enter_with_factors(h)) return rc;
    launch_gemv(h->P, h->q, h->st);  // warm
    HIPCHECK(h, hipEventRecord(h->ev0, h->st));
    for (int i = 0; i < reps; ++i) launch_gemv(h->P, h->q, h->st);
    HIPCHECK(h, hipEventRecord(h->ev1, h->st));
    HIPCHECK(h, hipEventSynchronize(h->ev1));
    float ms = 0;
    HIPCHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (ms_per_launch) *ms_per_launch = ms / reps;
    if (bytes_per_launch) *bytes_per_launch = h->precond_bytes;
    return 0;
}

// One kernel class of the hot path launched `reps` times back to back on the handle's resident data (warm-up launch
// first), HIP events on the library's stream around them.  bytes = the algorithmic bytes of ONE launch by the formulas
// of SURVEY.md section 8(d) (spelled out per kind below and in DESIGN.md section 4).  The state of the handle is used as
// it is (call between steps); kinds that would disturb it (Hessian refresh) write to their usual buffers, which the next
// refresh overwrites anyway.  Kinds: enum dotmi_bench_kind.
int dotmi_bench_kernel(dotmi_handle *h, int32_t kind, int32_t reps, double *ms_per_launch, int64_t *bytes_per_launch)
{
    if (!h || reps < 1) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (int rc = enter_with_factors(h)) return rc;
    const int n = h->n, nV = h->nV;
    const int64_t nTo = h->PT.nElem, nVo = h->v1 - h->v0, m = h->m > 0 ? h->m : h->hist;
    LbfgsArgs L = lbfgs_args(h);
    L.m = (int)std::min<int64_t>(m, h->hist);   // as in a running step with a full history
    for (int i = 0; i < L.m; ++i) {
        L.s[i] = h->S[i];
        L.y[i] = h->Y[i];
        if (L.ys[i] == 0.0) L.ys[i] = 1.0;
    }
    int nb = 0;
    int64_t bytes = 0;
    bool live = false;   // the kernel form reads the device loop's state
    std::function<void()> run;
    GatherArgs a;
    memset(&a, 0, sizeof(a));
    a.x = h->x;
    a.xt = h->xt;
    a.g_old = h->g;
    a.p = h->p;
    a.alpha_dev = h->alpha_dev;
    a.g_new = h->g_trial;
    a.s_new = h->S[h->hist];
    a.y_new = h->Y[h->hist];
    a.iv0 = h->v0;
    a.iv1 = h->v1;
    a.make_pair = 1;
    double xi[HIST_MAX] = {0, 0, 0, 0, 0, 0};
    switch (kind) {
    case DOTMI_BENCH_ELEM_ENERGY_GRAD:   // 112 nT + 56 nV: energy evaluation incl. inertia (the gradient entries stay on chip)
        bytes = 112 * nTo + 56 * nVo;
        run = [&] { launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x, h->xt, h->v0, h->v1, 1, h->partE, &nb, h->st); };
        break;
    case DOTMI_BENCH_ELEM_ENERGY:        // 112 nT + 56 nV
        bytes = 112 * nTo + 56 * nVo;
        run = [&] { launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x, h->xt, h->v0, h->v1, 0, h->partE, &nb, h->st); };
        break;
    case DOTMI_BENCH_VERTEX_GATHER:      // g write + x, x~, m read (80 nV) + pair: g_old, p read, s, y write + 2m history vectors
        bytes = 80 * (int64_t)nV + (int64_t)(4 + 2 * L.m) * 8 * n;
        run = [&] { launch_vertex_gather(h->M, h->PT, a, L, h->partR, h->st); };
        break;
    case DOTMI_BENCH_SPMV_DOTS:          // 72 nnzb (full symmetric block rows) + p, g read
        bytes = 72 * (int64_t)h->M.nnzb + 2 * 8 * (int64_t)n;
        run = [&] { launch_spmv_dots(h->M, h->Hval, h->p, h->g, nullptr, h->v0, h->v1, h->partS, h->st); };
        break;
    case DOTMI_BENCH_BACKSOLVE:          // 8 x structural non-zeros of the block-sparse inverse factors
        bytes = h->precond_bytes;
        run = [&] { launch_gemv(h->P, h->q, h->st); };
        break;
    case DOTMI_BENCH_MERGE:              // z write + the tile partials that make it up + m history vectors (y_i . z)
        bytes = 8 * (int64_t)n * (2 + L.m) + 8 * (int64_t)h->mergeEntries;
        // (split form, big meshes: the coalesced reduce of the tile partials is the first half of the merge)
        run = [&] {
            if (!h->P.mt_ptr) launch_reduce_partial(h->P, h->st);
            launch_merge(h->M, h->P, L, h->z, h->partC, 1 | 2, h->st);
        };
        break;
    case DOTMI_BENCH_BUILD_QPAD:         // g + m history vectors read, padded right-hand sides written
        bytes = 8 * (int64_t)n * (1 + L.m) + 8 * (int64_t)h->P.nParts * h->P.nmax;
        run = [&] { launch_build_qpad(h->P, h->g, L, xi, h->st); };
        break;
    case DOTMI_BENCH_BUILD_P:            // z + m history vectors read, p written
        bytes = 8 * (int64_t)n * (2 + L.m);
        run = [&] { launch_build_p(n, h->z, L, h->partC, xi, h->p, h->st); };
        break;
    case DOTMI_BENCH_STEP_FORWARD:       // x, p read, x_trial written
        bytes = 8 * (int64_t)n * 3;
        run = [&] { launch_step_forward(n, h->x, h->p, h->x_trial, h->partS, 0.0, 1, h->alphaMin, h->alpha_dev, nullptr, h->st); };
        break;
    case DOTMI_BENCH_ELEM_HESSIAN:       // 112 nT in, 1152 nT out
        if (h->world > 1) return DOTMI_E_INVALID;   // (the refresh re-issued below ends in a collective: not from one rank alone)
        bytes = (int64_t)(112 + 1152) * h->nHessElems;
        run = [&] {
            if (h->shardHess) launch_elem_hessians(h->M, h->mat, h->dtSq, h->x, h->He, h->st, h->hessElems, h->nHessElems);
            else launch_elem_hessians(h->M, h->mat, h->dtSq, h->x, h->He, h->st);
        };
        break;
    case DOTMI_BENCH_ASSEMBLE:           // 1152 nT in, 72 nnzb out
        if (h->world > 1) return DOTMI_E_INVALID;
        bytes = (int64_t)1152 * h->nHessElems + 72 * (int64_t)(h->shardHess ? h->nHessBlk : h->M.nnzb);
        run = [&] {
            if (h->shardHess) launch_assemble(h->M, h->He, h->Hval, h->st, h->hessBlk, h->nHessBlk, h->hessBlkPtr, h->hessBlkEnt);
            else launch_assemble(h->M, h->He, h->Hval, h->st);
        };
        break;
    // ---- the forms the device loop's early order really launches.  They read the loop state on the
    // device, so the state the last step left there (history full, buffer roles, xi / delta) is switched back to "running,
    // new direction" for the duration of the measurement and restored afterwards; none of them advances it (only the
    // controller does), so every repetition does the same work.  They overwrite loop-internal vectors (p, z, the trial
    // point and gradient, the free pair slot, the padded right-hand sides), all of which the next step rewrites.
    case DOTMI_BENCH_SPMV_ZP:            // 72 nnzb + z, g read, p, Hp written, m pairs of (s_j, H s_j) read
        bytes = 72 * (int64_t)h->M.nnzb + 8 * (int64_t)n * (4 + 2 * L.m);
        live = true;
        run = [&] {
            launch_spmv_zp(h->M, h->Hval, h->z, h->dist ? h->partC : h->partCT, h->p, h->Hp, h->partS, h->st, h->ctl, 0, -1, nullptr,
                           nullptr, VList(), !h->dist);
        };
        break;
    case DOTMI_BENCH_MERGE_EARLY:        // tile partials + u_old read / written, z written, M y_new written, m x (y_j, M y_j) read
        bytes = 8 * (int64_t)h->mergeEntries + 8 * (int64_t)n * (4 + 2 * L.m);
        live = true;
        run = [&] {
            if (!h->P.mt_ptr) launch_reduce_partial(h->P, h->st, h->ctl);
            launch_merge_early(h->M, h->P, h->z, h->partC, 0, h->st, h->ctl, nullptr, nullptr, VList(), nullptr, 0, nullptr,
                               h->dist ? nullptr : h->partCT);
        };
        break;
    case DOTMI_BENCH_ELEM_STEP: {        // the element pass with the line-search step inside: + p read, trial point written
        bytes = 112 * nTo + 56 * nVo + 48 * (int64_t)nV;
        live = true;
        run = [&] {
            StepArgs sa{h->p, h->partS, h->alpha_dev, h->alphaMin};
            launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x_trial, h->xt, h->v0, h->v1, 1, h->partE, &nb, h->st, h->ctl,
                                    h->tune.fuseStep ? &sa : nullptr);
        };
        break;
    }
    case DOTMI_BENCH_GATHER_EARLY: {     // + H p read, H s_new written, -g into the padded right-hand sides of every holder
        long long held = 0;
        for (int v = 0; v < nV; ++v) held += h->dup[v];
        bytes = 80 * (int64_t)nV + (int64_t)(6 + 2 * L.m) * 8 * n + 24 * held;
        live = true;
        a.x = nullptr;
        a.g_old = nullptr;
        a.g_new = nullptr;
        a.s_new = nullptr;
        a.y_new = nullptr;
        a.hp = h->tune.fuseDir ? h->Hp : nullptr;
        a.vp_ptr = h->P.vp_ptr;
        a.vp_off = h->P.vp_off;
        a.rpad = h->P.rpad;
        run = [&] { launch_vertex_gather(h->M, h->PT, a, L, h->partR, h->st, h->ctl); };
        break;
    }
    case DOTMI_BENCH_ELEM_VERTEX: {     // what the stepping element pass and the early gather read and write together (the halo's
                                        // re-reads are traffic, not algorithmic bytes)
        if (!h->vpFits) return DOTMI_E_INVALID;
        long long held = 0;
        for (int v = 0; v < nV; ++v) held += h->dup[v];
        bytes = 112 * nTo + 56 * nVo + 48 * (int64_t)nV + 80 * (int64_t)nV + (int64_t)(6 + 2 * L.m) * 8 * n + 24 * held;
        live = true;
        run = [&] {
            ElemVertArgs ea;
            memset(&ea, 0, sizeof(ea));
            ea.mass = h->M.mass;
            ea.xt = h->xt;
            ea.p = h->p;
            ea.hp = h->Hp;
            ea.spmv_partials = h->partST;
            ea.fixed = h->M.fixed;
            ea.vp_ptr = h->P.vp_ptr;
            ea.vp_off = h->P.vp_off;
            ea.rpad = h->P.rpad;
            ea.partE = h->partE;
            ea.partR = h->partR;
            ea.alpha_out = h->alpha_dev;
            ea.dtSq = h->dtSq;
            ea.alpha_min = h->alphaMin;
            launch_elem_vertex(h->VP, h->mat, ea, h->st, h->ctl);
        };
        break;
    }
    case DOTMI_BENCH_DIRSTEP: {         // what spmv_zp and the stepping element pass read and write together (p is not re-read)
        if (h->dist || !h->specFits) return DOTMI_E_INVALID;
        bytes = 72 * (int64_t)h->M.nnzb + 8 * (int64_t)n * (4 + 2 * L.m) + 112 * nTo + 56 * nVo + 24 * (int64_t)nV;
        live = true;
        run = [&] {
            StepArgs sa{h->p, h->partS, h->alpha_dev, h->alphaMin};
            launch_dirstep(h->M, h->PTspec, h->mat, h->dtSq, h->xt, h->partE, &nb, h->Hval, h->z, h->partCT, h->p, h->Hp, h->partS, h->st,
                           h->ctl, sa);
        };
        break;
    }
    default:
        return DOTMI_E_INVALID;
    }
    DevLoop saved;
    if (live) {
        if (!h->earlyBs || h->prevSlots < 0 || !h->P.vp_ptr) {
            h->err = "the in-loop kernel forms need a handle that has run a step of the device loop's early order";
            return DOTMI_E_INVALID;
        }
        HIPCHECK(h, hipMemcpy(&saved, h->ctl, sizeof(DevLoop), hipMemcpyDeviceToHost));
        DevLoop live_ctl = saved;
        live_ctl.status = 0;
        live_ctl.phase = 0;
        HIPCHECK(h, hipMemcpy(h->ctl, &live_ctl, sizeof(DevLoop), hipMemcpyHostToDevice));
    }
    run();   // warm
    HIPCHECK(h, hipEventRecord(h->ev0, h->st));
    for (int i = 0; i < reps; ++i) run();
    HIPCHECK(h, hipEventRecord(h->ev1, h->st));
    HIPCHECK(h, hipEventSynchronize(h->ev1));
    HIPCHECK(h, hipGetLastError());
    if (live) HIPCHECK(h, hipMemcpy(h->ctl, &saved, sizeof(DevLoop), hipMemcpyHostToDevice));
    float ms = 0;
    HIPCHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (kind == DOTMI_BENCH_ELEM_HESSIAN || kind == DOTMI_BENCH_ASSEMBLE) {
        // these two rewrote the element / global Hessians at the CURRENT positions; the factors (and the alpha_0 of the next
        // step) belong to the positions of the last refresh -- bring everything back in line
        if (int rc = refactor(h, h->x, nullptr, nullptr)) return rc;
    }
    if (ms_per_launch) *ms_per_launch = ms / reps;
    if (bytes_per_launch) *bytes_per_launch = bytes;
    return 0;
}

int dotmi_bench_energy(dotmi_handle *h, int32_t reps, double *ms_per_launch, int64_t *bytes_per_launch)
{
    if (!h || reps < 1) return DOTMI_E_INVALID;
    HIPCHECK(h, hipSetDevice(h->device));
    if (resolve_refresh(h) == DOTMI_E_DEVICE) return DOTMI_E_DEVICE;   // (asynchronous refresh of the last step)
    int nb = 0;
    launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x, h->xt, h->v0, h->v1, 0, h->partE, &nb, h->st);
    HIPCHECK(h, hipEventRecord(h->ev0, h->st));
    for (int i = 0; i < reps; ++i)
        launch_elem_energy_grad(h->M, h->PT, h->mat, h->dtSq, h->x, h->xt, h->v0, h->v1, 0, h->partE, &nb, h->st);
    HIPCHECK(h, hipEventRecord(h->ev1, h->st));
    HIPCHECK(h, hipEventSynchronize(h->ev1));
    float ms = 0;
    HIPCHECK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (ms_per_launch) *ms_per_launch = ms / reps;
    // 112 B per tet + 56 B per vertex
    if (bytes_per_launch) *bytes_per_launch = (int64_t)112 * h->nOwnElem + (int64_t)56 * (h->v1 - h->v0);
    return 0;
}

}  // extern "C"
