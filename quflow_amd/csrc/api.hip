// C ABI of libquflow_hip.so (see include/quflow_hip.h) and the host-side control flow
// of the isospectral midpoint stepper (quflow/integrators/isospectral.py:338-613).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <limits>

#include "qf_internal.h"

static thread_local char g_err[512] = "";

void qf_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

namespace {
int drain_events(qf_ctx *ctx);

struct prof_scope {
    qf_ctx *ctx;
    qf_event_pair ev;
    bool active;
    prof_scope(qf_ctx *c, int id) : ctx(c), active(((c->profile_mask >> id) & 1) != 0)
    {
        if (!active) return;
        if (ctx->events_free.empty()) {
            if (hipEventCreate(&ev.start) != hipSuccess || hipEventCreate(&ev.stop) != hipSuccess) {
                active = false;
                return;
            }
        } else {
            ev = ctx->events_free.back();
            ctx->events_free.pop_back();
        }
        ev.kernel_id = id;
        (void)hipEventRecord(ev.start, ctx->stream);
    }
    ~prof_scope()
    {
        if (!active) return;
        (void)hipEventRecord(ev.stop, ctx->stream);
        ctx->events_busy.push_back(ev);
        if (ctx->events_busy.size() >= 8192) (void)drain_events(ctx);
    }
};

int drain_events(qf_ctx *ctx)
{
    if (ctx->events_busy.empty()) return QF_OK;
    QF_HIP(hipStreamSynchronize(ctx->stream));
    for (auto &ev : ctx->events_busy) {
        float ms = 0.f;
        QF_HIP(hipEventElapsedTime(&ms, ev.start, ev.stop));
        ctx->prof_launches[ev.kernel_id] += 1;
        ctx->prof_ms[ev.kernel_id] += (double)ms;
        ctx->events_free.push_back(ev);
    }
    ctx->events_busy.clear();
    return QF_OK;
}

int alloc_factors(qf_ctx *ctx, qf_factors *f)
{
    const size_t NN = (size_t)ctx->N * ctx->N;
    QF_HIP(hipMalloc((void **)&f->wtab, NN * sizeof(double)));
    QF_HIP(hipMalloc((void **)&f->invtab, NN * sizeof(double)));
    return QF_OK;
}

int check_ctx(const qf_ctx *ctx)
{
    if (!ctx) {
        qf_set_error("null qf_ctx");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipSetDevice(ctx->device));
    return QF_OK;
}

int read_scalar(qf_ctx *ctx, const double *dev, double *out)
{
    QF_HIP(hipMemcpyAsync(ctx->host_scalars, dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    *out = ctx->host_scalars[0];
    return QF_OK;
}

}  // namespace

extern "C" {

int qf_version(void) { return QF_VERSION; }

const char *qf_last_error(void) { return g_err; }

int qf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

double qf_hbar(int N) { return 2.0 / std::sqrt((double)N * (double)N - 1.0); }

int qf_ctx_create(int N, int device, qf_ctx **out)
{
    if (!out) {
        qf_set_error("qf_ctx_create: out is null");
        return QF_ERR_INVALID;
    }
    *out = nullptr;
    if (N < 2 || N > 8192) {
        qf_set_error("qf_ctx_create: N=%d out of range [2, 8192]", N);
        return QF_ERR_INVALID;
    }
    int ndev = qf_device_count();
    if (ndev <= 0) {
        qf_set_error("qf_ctx_create: no HIP device visible (this library has no CPU fallback)");
        return QF_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        qf_set_error("qf_ctx_create: device %d out of range (0..%d)", device, ndev - 1);
        return QF_ERR_NO_DEVICE;
    }
    QF_HIP(hipSetDevice(device));
    qf_ctx *ctx = new qf_ctx();
    ctx->N = N;
    ctx->device = device;
    const size_t NN = (size_t)N * N;
    const size_t mbytes = NN * sizeof(cplx);
    int rc = QF_OK;
    auto fail = [&](int code) {
        qf_ctx_destroy(ctx);
        return code;
    };
#define QF_CREATE_HIP(call)                                                                   \
    do {                                                                                      \
        hipError_t _e = (call);                                                               \
        if (_e != hipSuccess) {                                                               \
            qf_set_error("%s failed: %s", #call, hipGetErrorString(_e));                     \
            return fail(QF_ERR_HIP);                                                          \
        }                                                                                     \
    } while (0)
    QF_CREATE_HIP(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    cplx **mats[] = {&ctx->W, &ctx->dW[0], &ctx->dW[1], &ctx->Whalf, &ctx->Phalf, &ctx->PW, &ctx->stage};
    for (cplx **m : mats) {
        QF_CREATE_HIP(hipMalloc((void **)m, mbytes));
        QF_CREATE_HIP(hipMemsetAsync(*m, 0, mbytes, ctx->stream));
    }
    QF_CREATE_HIP(hipMalloc((void **)&ctx->lap, 2 * NN * sizeof(double)));
    ctx->rowpart_tiles = qf_gemm_tiles_n(N);
    QF_CREATE_HIP(hipMalloc((void **)&ctx->rowpart, (size_t)ctx->rowpart_tiles * N * sizeof(double)));
    QF_CREATE_HIP(hipMalloc((void **)&ctx->rowsum, (size_t)N * sizeof(double)));
    QF_CREATE_HIP(hipMalloc((void **)&ctx->scalars, 4096 * sizeof(double)));
    QF_CREATE_HIP(hipHostMalloc((void **)&ctx->host_scalars, 64 * sizeof(double), hipHostMallocDefault));
    QF_CREATE_HIP(hipEventCreate(&ctx->timer_start));
    QF_CREATE_HIP(hipEventCreate(&ctx->timer_stop));
#undef QF_CREATE_HIP
    // coefficient table of Delta_N with the bc of cpu.py:90, and its factorisation (once per N)
    if ((rc = alloc_factors(ctx, &ctx->poisson)) != QF_OK) return fail(rc);
    if ((rc = qf_launch_lap_table(ctx, 1, ctx->lap)) != QF_OK) return fail(rc);
    if ((rc = qf_launch_build_factors(ctx, ctx->lap, ctx->poisson)) != QF_OK) return fail(rc);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
        qf_set_error("qf_ctx_create: table construction failed");
        return fail(QF_ERR_HIP);
    }
    *out = ctx;
    return QF_OK;
}

int qf_ctx_destroy(qf_ctx *ctx)
{
    if (!ctx) return QF_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    void *ptrs[] = {ctx->W, ctx->dW[0], ctx->dW[1], ctx->Whalf, ctx->Phalf, ctx->PW, ctx->kahan_c, ctx->stage,
                    ctx->lap, ctx->lap_user, ctx->poisson.wtab, ctx->poisson.invtab, ctx->rowpart, ctx->rowsum,
                    ctx->scalars};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &kv : ctx->user_factors) {
        if (kv.second.wtab) (void)hipFree(kv.second.wtab);
        if (kv.second.invtab) (void)hipFree(kv.second.invtab);
    }
    if (ctx->host_scalars) (void)hipHostFree(ctx->host_scalars);
    for (auto &ev : ctx->events_busy) {
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    for (auto &ev : ctx->events_free) {
        (void)hipEventDestroy(ev.start);
        (void)hipEventDestroy(ev.stop);
    }
    if (ctx->timer_start) (void)hipEventDestroy(ctx->timer_start);
    if (ctx->timer_stop) (void)hipEventDestroy(ctx->timer_stop);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return QF_OK;
}

int qf_ctx_size(const qf_ctx *ctx) { return ctx ? ctx->N : -1; }

int qf_sync(qf_ctx *ctx)
{
    QF_TRY(check_ctx(ctx));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_laplacian_table(qf_ctx *ctx, int bc, double *lap_host)
{
    QF_TRY(check_ctx(ctx));
    if (!lap_host) {
        qf_set_error("qf_laplacian_table: null output");
        return QF_ERR_INVALID;
    }
    const size_t bytes = 2 * (size_t)ctx->N * ctx->N * sizeof(double);
    double *dst = ctx->lap;
    if (!bc) {
        if (!ctx->lap_user) QF_HIP(hipMalloc((void **)&ctx->lap_user, bytes));
        dst = ctx->lap_user;
        QF_TRY(qf_launch_lap_table(ctx, 0, dst));
    }
    QF_HIP(hipMemcpyAsync(lap_host, dst, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_solve_poisson(qf_ctx *ctx, const void *W_host, void *P_host, int skewh)
{
    QF_TRY(check_ctx(ctx));
    if (!W_host || !P_host) {
        qf_set_error("qf_solve_poisson: null buffer");
        return QF_ERR_INVALID;
    }
    const size_t bytes = (size_t)ctx->N * ctx->N * sizeof(cplx);
    QF_HIP(hipMemcpyAsync(ctx->stage, W_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_solve(ctx, ctx->poisson, ctx->stage, ctx->Phalf, 1.0, skewh));
    QF_HIP(hipMemcpyAsync(P_host, ctx->Phalf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_laplace(qf_ctx *ctx, const void *P_host, void *W_host)
{
    QF_TRY(check_ctx(ctx));
    if (!W_host || !P_host) {
        qf_set_error("qf_laplace: null buffer");
        return QF_ERR_INVALID;
    }
    const size_t bytes = (size_t)ctx->N * ctx->N * sizeof(cplx);
    QF_HIP(hipMemcpyAsync(ctx->stage, P_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_laplace(ctx, ctx->stage, ctx->Phalf));
    QF_HIP(hipMemcpyAsync(W_host, ctx->Phalf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_solve_tridiagonal(qf_ctx *ctx, const double *lap_host, unsigned long long table_key,
                         const void *W_host, void *P_host, int skewh)
{
    QF_TRY(check_ctx(ctx));
    if (!lap_host || !W_host || !P_host) {
        qf_set_error("qf_solve_tridiagonal: null buffer");
        return QF_ERR_INVALID;
    }
    const size_t NN = (size_t)ctx->N * ctx->N;
    qf_factors f;
    auto it = table_key ? ctx->user_factors.find(table_key) : ctx->user_factors.end();
    if (it != ctx->user_factors.end()) {
        f = it->second;
    } else {
        if (!ctx->lap_user) QF_HIP(hipMalloc((void **)&ctx->lap_user, 2 * NN * sizeof(double)));
        auto slot = ctx->user_factors.find(0);
        if (table_key == 0 && slot != ctx->user_factors.end()) {
            f = slot->second;  // reuse the anonymous slot
        } else {
            QF_TRY(alloc_factors(ctx, &f));
            ctx->user_factors[table_key] = f;
        }
        QF_HIP(hipMemcpyAsync(ctx->lap_user, lap_host, 2 * NN * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        QF_TRY(qf_launch_build_factors(ctx, ctx->lap_user, f));
    }
    QF_HIP(hipMemcpyAsync(ctx->stage, W_host, NN * sizeof(cplx), hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_solve(ctx, f, ctx->stage, ctx->Phalf, 1.0, skewh));
    QF_HIP(hipMemcpyAsync(P_host, ctx->Phalf, NN * sizeof(cplx), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_upload_W(qf_ctx *ctx, const void *W_host)
{
    QF_TRY(check_ctx(ctx));
    if (!W_host) {
        qf_set_error("qf_upload_W: null buffer");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipMemcpyAsync(ctx->W, W_host, (size_t)ctx->N * ctx->N * sizeof(cplx), hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_download_W(qf_ctx *ctx, void *W_host)
{
    QF_TRY(check_ctx(ctx));
    if (!W_host) {
        qf_set_error("qf_download_W: null buffer");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipMemcpyAsync(W_host, ctx->W, (size_t)ctx->N * ctx->N * sizeof(cplx), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_norm_inf_W(qf_ctx *ctx, double *out)
{
    QF_TRY(check_ctx(ctx));
    QF_TRY(qf_launch_norm_inf(ctx, ctx->W, ctx->scalars));
    return read_scalar(ctx, ctx->scalars, out);
}

// isomp_fixedpoint, quflow/integrators/isospectral.py:338-613 (autonomous, built-in Hamiltonian).
int qf_isomp(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum,
             int reinitialize, qf_isomp_stats *stats_out)
{
    QF_TRY(check_ctx(ctx));
    if (minit < 1) {  // isospectral.py:400
        qf_set_error("minit must be at least 1.");
        return QF_ERR_INVALID;
    }
    if (maxit < minit) {  // isospectral.py:401
        qf_set_error("maxit must be at minit.");
        return QF_ERR_INVALID;
    }
    if (steps < 0) {
        qf_set_error("qf_isomp: steps must be >= 0");
        return QF_ERR_INVALID;
    }
    const int N = ctx->N;
    const size_t mbytes = (size_t)N * N * sizeof(cplx);
    const double hb = qf_hbar(N);          // isospectral.py:436
    const double vareps = dt / (2 * hb);   // isospectral.py:437

    // tolerance, isospectral.py:440-452
    if (tol < 0) {
        double mach_eps = std::numeric_limits<double>::epsilon();
        if (!compsum) mach_eps = std::sqrt(mach_eps);
        double nrm = 0.0;
        QF_TRY(qf_norm_inf_W(ctx, &nrm));
        tol = (mach_eps * dt / hb) * nrm;
    }

    // dW = 0 at every entry (isospectral.py:430) => Whalf = W
    ctx->dw_cur = 0;
    QF_HIP(hipMemsetAsync(ctx->dW[0], 0, mbytes, ctx->stream));
    QF_HIP(hipMemcpyAsync(ctx->Whalf, ctx->W, mbytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (compsum) {
        if (!ctx->kahan_c) QF_HIP(hipMalloc((void **)&ctx->kahan_c, mbytes));
        QF_HIP(hipMemsetAsync(ctx->kahan_c, 0, mbytes, ctx->stream));  // isospectral.py:457
    }

    long long total_iterations = 0, number_of_maxit = 0;
    double resnorm = std::numeric_limits<double>::infinity();
    double *d_res = ctx->scalars + 1;

    for (int k = 0; k < steps; ++k) {
        resnorm = std::numeric_limits<double>::infinity();  // isospectral.py:470
        if (reinitialize && k > 0) {
            // dW.fill(0): Whalf was already set to W by the previous update (reinitialize path)
            QF_HIP(hipMemsetAsync(ctx->dW[ctx->dw_cur], 0, mbytes, ctx->stream));
        }
        bool broke = false;
        for (int i = 0; i < maxit; ++i) {
            total_iterations += 1;
            cplx *dW_old = ctx->dW[ctx->dw_cur];
            cplx *dW_new = ctx->dW[ctx->dw_cur ^ 1];
            {   // Phalf = vareps * solve_poisson(Whalf)          isospectral.py:488-492
                prof_scope p(ctx, QF_KERNEL_POISSON);
                QF_TRY(qf_launch_solve(ctx, ctx->poisson, ctx->Whalf, ctx->Phalf, vareps, 1));
            }
            {   // PW = Phalf @ Whalf                              isospectral.py:496
                prof_scope p(ctx, QF_KERNEL_GEMM1);
                QF_TRY(qf_launch_zgemm(ctx, ctx->Phalf, ctx->Whalf, ctx->PW, nullptr));
            }
            {   // dW = PW @ Phalf + (PW - PW^H); Whalf = W + dW; row sums of |dW_old - dW|
                prof_scope p(ctx, QF_KERNEL_GEMM2);
                qf_epilogue ep;
                ep.PW = ctx->PW;
                ep.W = ctx->W;
                ep.dW_old = dW_old;
                ep.dW_new = dW_new;
                ep.Whalf = ctx->Whalf;
                ep.rowpart = ctx->rowpart;
                QF_TRY(qf_launch_zgemm(ctx, ctx->PW, ctx->Phalf, nullptr, &ep));
            }
            ctx->dw_cur ^= 1;
            if (i + 1 >= minit) {  // isospectral.py:523-536
                double resnorm_old = resnorm;
                {
                    prof_scope p(ctx, QF_KERNEL_NORM);
                    QF_TRY(qf_launch_norm_from_rowpart(ctx, ctx->rowpart, ctx->rowpart_tiles, d_res));
                }
                QF_TRY(read_scalar(ctx, d_res, &resnorm));
                if (resnorm <= tol || resnorm >= resnorm_old) {
                    broke = true;
                    break;
                }
            }
        }
        if (!broke) number_of_maxit += 1;  // for-else, isospectral.py:538-540
        {   // W += 2*(PW - PW^H) (Kahan if compsum); Whalf = W + dW     isospectral.py:547-592
            prof_scope p(ctx, QF_KERNEL_UPDATE);
            QF_TRY(qf_launch_update(ctx, ctx->PW, ctx->W, ctx->dW[ctx->dw_cur], ctx->Whalf,
                                    compsum ? ctx->kahan_c : nullptr, reinitialize));
        }
    }
    QF_HIP(hipStreamSynchronize(ctx->stream));
    if (stats_out) {
        stats_out->total_iterations = total_iterations;
        stats_out->number_of_maxit = number_of_maxit;
        stats_out->tol_used = tol;
        stats_out->last_resnorm = resnorm;
    }
    return QF_OK;
}

int qf_diagnostics(qf_ctx *ctx, double *energy_euler, double *enstrophy)
{
    QF_TRY(check_ctx(ctx));
    const int N = ctx->N;
    // P = solve_poisson(W); energy = -inner_L2(W, P)/2; enstrophy = inner_L2(W, W)/2
    QF_TRY(qf_launch_solve(ctx, ctx->poisson, ctx->W, ctx->stage, 1.0, 1));
    QF_TRY(qf_launch_inner(ctx, ctx->W, ctx->stage, ctx->scalars + 2));
    double wp = 0.0, ww = 0.0;
    QF_TRY(read_scalar(ctx, ctx->scalars + 2, &wp));
    QF_TRY(qf_launch_inner(ctx, ctx->W, ctx->W, ctx->scalars + 3));
    QF_TRY(read_scalar(ctx, ctx->scalars + 3, &ww));
    if (energy_euler) *energy_euler = -(wp / N) / 2.0;
    if (enstrophy) *enstrophy = (ww / N) / 2.0;
    return QF_OK;
}

int qf_profile_enable(qf_ctx *ctx, int mask)
{
    QF_TRY(check_ctx(ctx));
    if (!mask) QF_TRY(drain_events(ctx));
    ctx->profile_mask = mask;
    return QF_OK;
}

int qf_profile_reset(qf_ctx *ctx)
{
    QF_TRY(check_ctx(ctx));
    QF_TRY(drain_events(ctx));
    for (int i = 0; i < QF_KERNEL_COUNT; ++i) {
        ctx->prof_launches[i] = 0;
        ctx->prof_ms[i] = 0.0;
    }
    return QF_OK;
}

int qf_profile_read(qf_ctx *ctx, int kernel_id, long long *launches, double *total_ms)
{
    QF_TRY(check_ctx(ctx));
    if (kernel_id < 0 || kernel_id >= QF_KERNEL_COUNT) {
        qf_set_error("qf_profile_read: bad kernel id %d", kernel_id);
        return QF_ERR_INVALID;
    }
    QF_TRY(drain_events(ctx));
    if (launches) *launches = ctx->prof_launches[kernel_id];
    if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
    return QF_OK;
}

int qf_timer_start(qf_ctx *ctx)
{
    QF_TRY(check_ctx(ctx));
    QF_HIP(hipEventRecord(ctx->timer_start, ctx->stream));
    return QF_OK;
}

int qf_timer_stop(qf_ctx *ctx, double *elapsed_ms)
{
    QF_TRY(check_ctx(ctx));
    QF_HIP(hipEventRecord(ctx->timer_stop, ctx->stream));
    QF_HIP(hipEventSynchronize(ctx->timer_stop));
    float ms = 0.f;
    QF_HIP(hipEventElapsedTime(&ms, ctx->timer_start, ctx->timer_stop));
    if (elapsed_ms) *elapsed_ms = (double)ms;
    return QF_OK;
}

int qf_download_buffer(qf_ctx *ctx, int which, void *host)
{
    QF_TRY(check_ctx(ctx));
    const cplx *src = nullptr;
    switch (which) {
        case QF_BUF_W: src = ctx->W; break;
        case QF_BUF_DW: src = ctx->dW[ctx->dw_cur]; break;
        case QF_BUF_WHALF: src = ctx->Whalf; break;
        case QF_BUF_PHALF: src = ctx->Phalf; break;
        case QF_BUF_PW: src = ctx->PW; break;
        default:
            qf_set_error("qf_download_buffer: unknown buffer %d", which);
            return QF_ERR_INVALID;
    }
    if (!host) {
        qf_set_error("qf_download_buffer: null host pointer");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipMemcpyAsync(host, src, (size_t)ctx->N * ctx->N * sizeof(cplx), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_zgemm(qf_ctx *ctx, const void *A_host, const void *B_host, void *C_host)
{
    QF_TRY(check_ctx(ctx));
    if (!A_host || !B_host || !C_host) {
        qf_set_error("qf_zgemm: null buffer");
        return QF_ERR_INVALID;
    }
    const size_t bytes = (size_t)ctx->N * ctx->N * sizeof(cplx);
    QF_HIP(hipMemcpyAsync(ctx->stage, A_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(ctx->Phalf, B_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_zgemm(ctx, ctx->stage, ctx->Phalf, ctx->PW, nullptr));
    QF_HIP(hipMemcpyAsync(C_host, ctx->PW, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

}  // extern "C"
