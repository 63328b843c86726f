// C ABI of libquflow_hip.so, part 5 of 5: spherical-harmonics <-> matrix TRANSFORMS (quflow/quantization.py) and the
// plain matrix-product entry points (qf_zgemm, qf_zgemm_i8, qf_cgemm).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>

#include "qf_api.h"

extern "C" {

// ---- spherical-harmonics transforms (quflow/quantization.py) -------------------------------
static int need_basis(qf_ctx *ctx, const char *who)
{
    if (!ctx->basis) {
        qf_set_error("%s: no quantization basis on this context (call qf_basis_upload first)", who);
        return QF_ERR_STATE;
    }
    return QF_OK;
}

// band limit of a coefficient array with n entries: quantization.py:204-208,294-298 (parallel form)
static int band_limit(int N, long long n)
{
    if (n >= (long long)N * N) return N;
    return (int)std::sqrt((double)n);
}

static int alloc_sh(qf_ctx *ctx);

int qf_basis_upload(qf_ctx *ctx, const double *basis_host, long long count)
{
    QF_TRY(check_ctx(ctx));
    const long long N = ctx->N;
    const long long want = N * (N + 1) * (2 * N + 1) / 6;
    if (!basis_host || count != want) {
        qf_set_error("qf_basis_upload: the basis for N=%d has %lld entries (got %lld)", ctx->N, want, count);
        return QF_ERR_INVALID;
    }
    QF_TRY(alloc_sh(ctx));
    QF_HIP(hipMemcpyAsync(ctx->basis, basis_host, (size_t)want * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

static int alloc_sh(qf_ctx *ctx)
{
    const long long N = ctx->N;
    const long long want = N * (N + 1) * (2 * N + 1) / 6;
    if (!ctx->basis) QF_HIP(hipMalloc((void **)&ctx->basis, (size_t)want * sizeof(double)));
    if (!ctx->sh_stage) QF_HIP(hipMalloc((void **)&ctx->sh_stage, (size_t)4 * (N * (N + 1) / 2) * sizeof(cplx)));
    if (!ctx->sh_omega) QF_HIP(hipMalloc((void **)&ctx->sh_omega, (size_t)2 * N * N * sizeof(double)));
    return QF_OK;
}

int qf_basis_compute(qf_ctx *ctx)
{
    QF_TRY(check_ctx(ctx));
    const bool fresh = (ctx->basis == nullptr);
    QF_TRY(alloc_sh(ctx));
    const int rc = qf_launch_basis(ctx, ctx->basis);
    if (rc != QF_OK || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        if (fresh) {   // never leave a half-built basis behind
            (void)hipFree(ctx->basis);
            ctx->basis = nullptr;
        }
        if (rc == QF_OK) qf_set_error("qf_basis_compute: kernel failed");
        return rc == QF_OK ? QF_ERR_HIP : rc;
    }
    return QF_OK;
}

int qf_basis_download(qf_ctx *ctx, double *basis_host, long long count)
{
    QF_TRY(check_ctx(ctx));
    QF_TRY(need_basis(ctx, "qf_basis_download"));
    const long long N = ctx->N;
    const long long want = N * (N + 1) * (2 * N + 1) / 6;
    if (!basis_host || count != want) {
        qf_set_error("qf_basis_download: the basis for N=%d has %lld entries (got %lld)", ctx->N, want, count);
        return QF_ERR_INVALID;
    }
    QF_HIP(hipMemcpyAsync(basis_host, ctx->basis, (size_t)want * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_shr2mat(qf_ctx *ctx, const double *omega_host, long long n_omega, void *W_host)
{
    QF_TRY(check_ctx(ctx));
    QF_TRY(need_basis(ctx, "qf_shr2mat"));
    if (n_omega < 1) {
        qf_set_error("qf_shr2mat: empty coefficient array");
        return QF_ERR_INVALID;
    }
    const long long NN = (long long)ctx->N * ctx->N;
    const int Nmax = band_limit(ctx->N, n_omega);
    const long long ncopy = n_omega < NN ? n_omega : NN;
    // omega_host == NULL: the coefficients the last qf_mat2shr left on the device
    if (omega_host)
        QF_HIP(hipMemcpyAsync(ctx->sh_omega, omega_host, (size_t)ncopy * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    cplx *dst = W_host ? ctx->stage : ctx->W;
    if (!W_host) ctx->w_skew_known = false;
    QF_TRY(qf_launch_shr2mat(ctx, Nmax, ctx->sh_omega, dst));
    if (W_host) QF_HIP(hipMemcpyAsync(W_host, dst, (size_t)NN * sizeof(cplx), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_mat2shr(qf_ctx *ctx, const void *W_host, double *omega_host, long long n_omega)
{
    QF_TRY(check_ctx(ctx));
    QF_TRY(need_basis(ctx, "qf_mat2shr"));
    if (n_omega < 1) {
        qf_set_error("qf_mat2shr: empty coefficient array");
        return QF_ERR_INVALID;
    }
    const long long NN = (long long)ctx->N * ctx->N;
    const int Nmax = band_limit(ctx->N, n_omega);
    const long long ncopy = n_omega < NN ? n_omega : NN;
    const cplx *src = ctx->W;
    if (W_host) {
        QF_HIP(hipMemcpyAsync(ctx->stage, W_host, (size_t)NN * sizeof(cplx), hipMemcpyHostToDevice, ctx->stream));
        src = ctx->stage;
    }
    QF_HIP(hipMemsetAsync(ctx->sh_omega, 0, (size_t)ncopy * sizeof(double), ctx->stream));   // np.zeros, quantization.py:516
    QF_TRY(qf_launch_mat2shr(ctx, Nmax, src, ctx->sh_omega));
    // omega_host == NULL: leave the coefficients on the device (a following qf_shr2mat(NULL) uses them)
    if (omega_host)
        QF_HIP(hipMemcpyAsync(omega_host, ctx->sh_omega, (size_t)ncopy * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    if (omega_host)
        for (long long i = ncopy; i < n_omega; ++i) omega_host[i] = 0.0;
    return QF_OK;
}

int qf_shc2mat(qf_ctx *ctx, const void *omega_host, void *W_host)
{
    QF_TRY(check_ctx(ctx));
    QF_TRY(need_basis(ctx, "qf_shc2mat"));
    if (!omega_host) {
        qf_set_error("qf_shc2mat: null coefficient array");
        return QF_ERR_INVALID;
    }
    const size_t NN = (size_t)ctx->N * ctx->N;
    QF_HIP(hipMemcpyAsync(ctx->sh_omega, omega_host, NN * sizeof(cplx), hipMemcpyHostToDevice, ctx->stream));
    cplx *dst = W_host ? ctx->stage : ctx->W;
    if (!W_host) ctx->w_skew_known = false;
    QF_TRY(qf_launch_shc2mat(ctx, ctx->sh_omega, dst));
    if (W_host) QF_HIP(hipMemcpyAsync(W_host, dst, NN * sizeof(cplx), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_mat2shc(qf_ctx *ctx, const void *W_host, void *omega_host)
{
    QF_TRY(check_ctx(ctx));
    QF_TRY(need_basis(ctx, "qf_mat2shc"));
    if (!omega_host) {
        qf_set_error("qf_mat2shc: null coefficient array");
        return QF_ERR_INVALID;
    }
    const size_t NN = (size_t)ctx->N * ctx->N;
    const cplx *src = ctx->W;
    if (W_host) {
        QF_HIP(hipMemcpyAsync(ctx->stage, W_host, NN * sizeof(cplx), hipMemcpyHostToDevice, ctx->stream));
        src = ctx->stage;
    }
    QF_TRY(qf_launch_mat2shc(ctx, src, ctx->sh_omega));
    QF_HIP(hipMemcpyAsync(omega_host, ctx->sh_omega, NN * sizeof(cplx), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}


int qf_zgemm_i8(qf_ctx *ctx, const void *A_host, const void *B_host, void *C_host)
{
    QF_TRY(check_ctx(ctx));
    if (!A_host || !B_host || !C_host) {
        qf_set_error("qf_zgemm_i8: null buffer");
        return QF_ERR_INVALID;
    }
    if (ctx->N % 64 != 0) {
        qf_set_error("qf_zgemm_i8: N=%d is not a multiple of 64", ctx->N);
        return QF_ERR_INVALID;
    }
    QF_TRY(qf_oz_alloc(ctx));
    const size_t bytes = (size_t)ctx->N * ctx->N * sizeof(cplx);
    QF_HIP(hipMemcpyAsync(ctx->stage, A_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(ctx->Phalf, B_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    qf_oz_jobs jobs;
    jobs.n = 2;
    jobs.j[0].X = ctx->stage;
    jobs.j[0].planes = ctx->oz_planes[0];
    jobs.j[0].scale = ctx->oz_scale[0];
    jobs.j[1].X = ctx->Phalf;                  // B skew-Hermitian, sliced by rows like A (ozaki.hip)
    jobs.j[1].planes = ctx->oz_planes[1];
    jobs.j[1].scale = ctx->oz_scale[1];
    QF_TRY(qf_launch_oz_slice(ctx, jobs));
    QF_TRY(qf_launch_oz_gemm(ctx, ctx->oz_planes[0], ctx->oz_scale[0], ctx->oz_planes[1], ctx->oz_scale[1], ctx->PW));
    QF_HIP(hipMemcpyAsync(C_host, ctx->PW, bytes, hipMemcpyDeviceToHost, ctx->stream));
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

// commutator_skewherm / commutator_generic (isospectral.py:22-57) with the combination on the device: the staging
// matrices of the context (stage, Phalf, PW, Whalf) are per-iteration temporaries, free between stepper calls
int qf_commutator(qf_ctx *ctx, const void *W_host, const void *P_host, void *C_host, int skewherm)
{
    QF_TRY(check_ctx(ctx));
    if (!W_host || !P_host || !C_host) {
        qf_set_error("qf_commutator: null buffer");
        return QF_ERR_INVALID;
    }
    const size_t bytes = (size_t)ctx->N * ctx->N * sizeof(cplx);
    QF_HIP(hipMemcpyAsync(ctx->stage, W_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(ctx->Phalf, P_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_zgemm(ctx, ctx->stage, ctx->Phalf, ctx->PW, nullptr));                       // X = W @ P
    if (skewherm) {
        QF_TRY(qf_launch_neg_conj_transpose(ctx, ctx->PW, ctx->Whalf));                            // -X^H
        QF_TRY(qf_launch_lincomb(ctx, 1.0, ctx->PW, 1.0, ctx->Whalf, 0.0, ctx->PW));               // X - X^H      (:52)
    } else {
        QF_TRY(qf_launch_zgemm(ctx, ctx->Phalf, ctx->stage, ctx->Whalf, nullptr));                 // P @ W
        QF_TRY(qf_launch_lincomb(ctx, 1.0, ctx->PW, -1.0, ctx->Whalf, 0.0, ctx->PW));              // W @ P - P @ W (:33-34)
    }
    QF_HIP(hipMemcpyAsync(C_host, ctx->PW, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}


int qf_cgemm(qf_ctx *ctx, const void *A_host, const void *B_host, void *C_host)
{
    QF_TRY(qf_need_c64(ctx));
    if (!A_host || !B_host || !C_host) {
        qf_set_error("qf_cgemm: null buffer");
        return QF_ERR_INVALID;
    }
    qf_c64 *f = ctx->c64;
    const size_t bytes = (size_t)ctx->N * ctx->N * sizeof(float2);
    QF_HIP(hipMemcpyAsync(f->stage, A_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(f->Phalf, B_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_cgemm(ctx, f->stage, f->Phalf, f->PW, nullptr));
    QF_HIP(hipMemcpyAsync(C_host, f->PW, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}


}  // extern "C"
