// C ABI of libquflow_hip.so, part 2 of 5: the LAPLACIAN backend -- coefficient tables, solve_poisson / laplace, the
// cached factorisations of caller-supplied tridiagonal tables (solve_heat / helmholtz / viscdamp / globalqg), the
// diagnostics that need a solve, and the same entry points for complex64 data (float32 arithmetic).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>

#include "qf_api.h"

extern "C" {

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
    const bool resident = !W_host && !P_host;     // on the context's state, in place
    if (!lap_host || (!resident && (!W_host || !P_host))) {
        qf_set_error("qf_solve_tridiagonal: null buffer");
        return QF_ERR_INVALID;
    }
    const size_t NN = (size_t)ctx->N * ctx->N;
    // fingerprint of the caller's table: 4096 entries spread over it (FNV-1a over their bit patterns).
    // A key is a caller-side hash; the fingerprint catches a key that returns with different content.
    unsigned long long fp = 1469598103934665603ull;
    {
        const size_t n = 2 * NN, stride = n / 4096 ? n / 4096 : 1;
        for (size_t i = 0; i < n; i += stride) {
            unsigned long long bits;
            memcpy(&bits, lap_host + i, sizeof(bits));
            fp = (fp ^ bits) * 1099511628211ull;
        }
        unsigned long long bits;
        memcpy(&bits, lap_host + (n - 1), sizeof(bits));
        fp = (fp ^ bits) * 1099511628211ull;
    }
    qf_factors f;
    auto it = ctx->user_factors.find(table_key);
    const bool hit = table_key != 0 && it != ctx->user_factors.end() && it->second.fingerprint == fp;
    if (hit) {
        f = it->second.f;
        it->second.last_used = ++ctx->factor_clock;
    } else {
        if (!ctx->lap_user) QF_HIP(hipMalloc((void **)&ctx->lap_user, 2 * NN * sizeof(double)));
        if (it != ctx->user_factors.end()) {
            f = it->second.f;                    // same key (or the anonymous slot 0), other table: refactor in place
        } else {
            // new key: a fresh pair while the budget lasts, else the least recently used entry's buffers
            // (no hipFree: work queued on the stream may still read them, and the stream orders the reuse)
            const size_t entry_bytes = 2 * NN * sizeof(double);
            if ((ctx->user_factors.size() + 1) * entry_bytes > ctx->factor_budget_bytes && !ctx->user_factors.empty()) {
                auto lru = ctx->user_factors.begin();
                for (auto jt = ctx->user_factors.begin(); jt != ctx->user_factors.end(); ++jt)
                    if (jt->second.last_used < lru->second.last_used) lru = jt;
                f = lru->second.f;
                ctx->user_factors.erase(lru);
            } else {
                QF_TRY(alloc_factors(ctx, &f));
            }
        }
        qf_ctx::factor_entry &e = ctx->user_factors[table_key];
        e.f = f;
        e.fingerprint = fp;
        e.last_used = ++ctx->factor_clock;
        QF_HIP(hipMemcpyAsync(ctx->lap_user, lap_host, 2 * NN * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        QF_TRY(qf_launch_build_factors(ctx, ctx->lap_user, f));
        // (lap_host must stay valid until the copy has been queued from pageable memory: hipMemcpyAsync
        // from pageable memory returns after staging, so the caller's buffer is free on return)
    }
    if (resident) {     // W <- T^-1 W (a Strang half step of a viscous / damped run between device steps)
        if (!skewh) ctx->w_skew_known = false;   // (the skew-Hermitian solve mirrors exactly: the property survives)
        QF_HIP(hipMemcpyAsync(ctx->stage, ctx->W, NN * sizeof(cplx), hipMemcpyDeviceToDevice, ctx->stream));
        QF_TRY(qf_launch_solve(ctx, f, ctx->stage, ctx->W, 1.0, skewh));
        return QF_OK;
    }
    QF_HIP(hipMemcpyAsync(ctx->stage, W_host, NN * sizeof(cplx), hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_solve(ctx, f, ctx->stage, ctx->Phalf, 1.0, skewh));
    QF_HIP(hipMemcpyAsync(P_host, ctx->Phalf, NN * sizeof(cplx), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_factor_cache_stats(qf_ctx *ctx, int *entries, unsigned long long *device_bytes)
{
    QF_TRY(check_ctx(ctx));
    if (entries) *entries = (int)ctx->user_factors.size();
    if (device_bytes) *device_bytes = (unsigned long long)ctx->user_factors.size() * 2ull * ctx->N * ctx->N * sizeof(double);
    return QF_OK;
}


int qf_diagnostics(qf_ctx *ctx, double *energy_euler, double *enstrophy)
{
    QF_TRY(check_ctx(ctx));
    const int N = ctx->N;
    // P = solve_poisson(W); energy = -inner_L2(W, P)/2; enstrophy = inner_L2(W, W)/2
    QF_TRY(qf_enqueue_diagnostics(ctx));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    const double wp = ctx->host_scalars[0], ww = ctx->host_scalars[1];
    if (energy_euler) *energy_euler = -(wp / N) / 2.0;
    if (enstrophy) *enstrophy = (ww / N) / 2.0;
    return QF_OK;
}


// ---- complex64 data: float32 arithmetic, as the reference computes it (cpu.py:725, isospectral.py:440-448) ----

int qf_need_c64(qf_ctx *ctx)
{
    QF_TRY(check_ctx(ctx));
    return qf_c64_alloc(ctx);
}

int qf_c64_laplacian_table(qf_ctx *ctx, int bc, float *lap_host)
{
    QF_TRY(qf_need_c64(ctx));
    if (!lap_host) {
        qf_set_error("qf_c64_laplacian_table: null buffer");
        return QF_ERR_INVALID;
    }
    qf_c64 *f = ctx->c64;
    const size_t bytes = 2 * (size_t)ctx->N * ctx->N * sizeof(float);
    // (the resident table is the bc = True one: any other goes through the staging matrix, which has the same size)
    float *dst = bc ? f->lap : reinterpret_cast<float *>(f->stage);
    if (!bc) QF_TRY(qf_launch_lap_table_f32(ctx, 0, dst));
    QF_HIP(hipMemcpyAsync(lap_host, dst, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_c64_solve_poisson(qf_ctx *ctx, const void *W_host, void *P_host, int skewh)
{
    QF_TRY(qf_need_c64(ctx));
    if (!W_host || !P_host) {
        qf_set_error("qf_c64_solve_poisson: null buffer");
        return QF_ERR_INVALID;
    }
    qf_c64 *f = ctx->c64;
    const size_t bytes = (size_t)ctx->N * ctx->N * sizeof(float2);
    QF_HIP(hipMemcpyAsync(f->stage, W_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_solve_f32(ctx, f->tab, f->stage, f->Phalf, 1.0f, skewh));
    QF_HIP(hipMemcpyAsync(P_host, f->Phalf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_c64_solve_tridiagonal(qf_ctx *ctx, const float *lap_host, const void *W_host, void *P_host, int skewh)
{
    QF_TRY(qf_need_c64(ctx));
    if (!lap_host || !W_host || !P_host) {
        qf_set_error("qf_c64_solve_tridiagonal: null buffer");
        return QF_ERR_INVALID;
    }
    qf_c64 *f = ctx->c64;
    const size_t NN = (size_t)ctx->N * ctx->N, bytes = NN * sizeof(float2);
    // the caller's float32 table and its factorisation live in two scratch matrices of the float32 working set
    // (no cache: this is the secondary, host-in / host-out route; a factorisation is one sequential sweep per walk)
    float *lap_dev = reinterpret_cast<float *>(f->PW);
    float2 *tab_dev = f->dW[1];
    QF_HIP(hipMemcpyAsync(lap_dev, lap_host, 2 * NN * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_build_factors_f32(ctx, lap_dev, tab_dev));
    QF_HIP(hipMemcpyAsync(f->stage, W_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_solve_f32(ctx, tab_dev, f->stage, f->Phalf, 1.0f, skewh));
    QF_HIP(hipMemcpyAsync(P_host, f->Phalf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    f->increment_valid = false;      // (dW[1] was scratch)
    return QF_OK;
}

int qf_c64_laplace(qf_ctx *ctx, const void *P_host, void *W_host)
{
    QF_TRY(qf_need_c64(ctx));
    if (!W_host || !P_host) {
        qf_set_error("qf_c64_laplace: null buffer");
        return QF_ERR_INVALID;
    }
    qf_c64 *f = ctx->c64;
    const size_t bytes = (size_t)ctx->N * ctx->N * sizeof(float2);
    QF_HIP(hipMemcpyAsync(f->stage, P_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_laplace_f32(ctx, f->stage, f->Phalf));
    QF_HIP(hipMemcpyAsync(W_host, f->Phalf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}


int qf_c64_diagnostics(qf_ctx *ctx, double *energy_euler, double *enstrophy)
{
    QF_TRY(qf_need_c64(ctx));
    qf_c64 *f = ctx->c64;
    const int N = ctx->N;
    // P = solve_poisson(W); energy = -inner_L2(W, P)/2; enstrophy = inner_L2(W, W)/2  (physics.py:26-38)
    QF_TRY(qf_launch_solve_f32(ctx, f->tab, f->W, f->stage, 1.0f, 1));
    QF_TRY(qf_launch_inner2_f32(ctx, f->W, f->stage, ctx->scalars + 2));
    QF_HIP(hipMemcpyAsync(ctx->host_scalars, ctx->scalars + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    const double wp = ctx->host_scalars[0], ww = ctx->host_scalars[1];
    if (energy_euler) *energy_euler = -(wp / N) / 2.0;
    if (enstrophy) *enstrophy = (ww / N) / 2.0;
    return QF_OK;
}


}  // extern "C"
