// C ABI of libquflow_hip.so (include/quflow_hip.h), part 1 of 5: the library and the CONTEXT -- error text, device
// information, what-was-launched descriptions, context creation / destruction, state upload / download, the
// per-launch event profile and the call timer.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>

#include "qf_api.h"

static thread_local char g_err[512] = "";

void qf_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int qf_version(void) { return QF_VERSION; }

int qf_device_info(int device, char *buf, int n)
{
    if ((n > 0 && !buf) || n < 0) {
        qf_set_error("qf_device_info: bad arguments");
        return -QF_ERR_INVALID;
    }
    const int ndev = qf_device_count();
    if (device < 0 || device >= ndev) {
        qf_set_error("qf_device_info: device %d out of range (%d visible)", device, ndev);
        return -QF_ERR_NO_DEVICE;
    }
    hipDeviceProp_t prop;
    char pci[64] = "unknown";
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) {
        qf_set_error("qf_device_info: hipGetDeviceProperties(%d) failed", device);
        return -QF_ERR_HIP;
    }
    (void)hipDeviceGetPCIBusId(pci, (int)sizeof(pci), device);
    char text[512];
    int len = snprintf(text, sizeof(text),
                       "{\"ordinal\": %d, \"pci_bus_id\": \"%s\", \"name\": \"%.120s\", \"gcn_arch\": \"%.120s\", \"compute_units\": %d, "
                       "\"memory_bytes\": %zu}",
                       device, pci, prop.name, prop.gcnArchName, prop.multiProcessorCount, (size_t)prop.totalGlobalMem);
    if (len >= (int)sizeof(text)) len = (int)sizeof(text) - 1;     // (cannot happen with the bounded fields; never copy past text[])
    if (n > 0) {
        const int m = len < n - 1 ? len : n - 1;
        memcpy(buf, text, (size_t)m);
        buf[m] = 0;
    }
    return len;
}

int qf_plan_describe(qf_ctx *ctx, char *buf, int n)
{
    if (!ctx || (n > 0 && !buf) || n < 0) {
        qf_set_error("qf_plan_describe: bad arguments");
        return -QF_ERR_INVALID;
    }
    static const char *const role[QF_KERNEL_COUNT] = {"laplacian_inverse", "first_product", "second_product", "residual_norm",
                                                      "step_update", "slicing"};
    std::string out = "{\"N\": " + std::to_string(ctx->N);
    for (int r = 0; r < QF_KERNEL_COUNT; ++r) {
        out += ", \"";
        out += role[r];
        out += "\": ";
        out += ctx->plan[r].text[0] ? ctx->plan[r].text : "null";
    }
    out += "}";
    if (n > 0) {
        const size_t m = out.size() < (size_t)(n - 1) ? out.size() : (size_t)(n - 1);
        memcpy(buf, out.data(), m);
        buf[m] = 0;
    }
    return (int)out.size();
}

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
    if (const char *g = getenv("QUFLOW_HIP_GEMM")) {   // the products' arithmetic: fp64 3M (default), int8 digit splits, "auto"
        ctx->gemm_i8_allowed = (g[0] == 'i');       // "i8" / "i8x6": both products on the int8 matrix cores (ozaki.hip)
        if (g[0] == 'i' && strstr(g, "x6")) ctx->oz_digits = 6;
        // "i8x65": six digits for the first product (the commutator is read from it), FIVE for the second (T = PW @ Phalf
        // is O(|Phalf|) smaller than the commutator it is added to): 45 instead of 63 int8 GEMMs
        if (g[0] == 'i' && strstr(g, "x65")) ctx->oz_digits2 = 5;
        // "i8h" / "i8hx6": hybrid -- the first product stays on the fp64 matrix cores, only the second one
        // (T = PW @ Phalf, O(|Phalf|) smaller than the commutator term it is added to) is digit-split
        if (g[0] == 'i' && strchr(g, 'h')) ctx->gemm_i8_hybrid = true;
        // "i8x6f": the other hybrid -- the FIRST product (PW = Phalf @ Whalf: a full product, and the one the
        // commutator is read from: six digits) is digit-split, the second stays the fp64 upper-triangle kernel,
        // which needs no sliced PW: one slicing launch per iteration instead of two
        if (g[0] == 'i' && strchr(g, 'f') && !ctx->gemm_i8_hybrid) ctx->gemm_i8_first = true;
        if (g[0] == 'a') {      // "auto": the fastest products under which the whole GPU suite is green -- int8 digits from N = 1024
            ctx->gemm_i8_allowed = true;          // (below that the fp64 kernels win: DESIGN.md 3.6)
            ctx->oz_digits = 6;
            ctx->oz_digits2 = 5;                  // (round 4: the second product on the leading five -- "i8x65")
            ctx->gemm_i8_min_n = 1024;
        }
    }
    if (const char *g = getenv("QUFLOW_HIP_GEMM2")) ctx->gemm_tri_allowed = !(g[0] == 'f');   // "full" | "tri" (default)
    if (const char *g = getenv("QUFLOW_HIP_FUSED")) ctx->fused_allowed = !(g[0] == '0');
    if (const char *g = getenv("QUFLOW_HIP_I8_MIN_N")) ctx->gemm_i8_min_n = atoi(g);
    if (const char *g = getenv("QUFLOW_HIP_TRI_MIN_N")) ctx->gemm_tri_min_n = atoi(g);
    if (const char *g = getenv("QUFLOW_HIP_DEFER")) ctx->defer_allowed = !(g[0] == '0');
    if (getenv("QUFLOW_HIP_DEBUG"))
        if (const char *g = getenv("QUFLOW_HIP_DEBUG_DROP_FLAG")) ctx->debug_drop = atoi(g);   // fault injection (tests)
    if (const char *g = getenv("QUFLOW_HIP_FACTOR_CACHE_MB")) ctx->factor_budget_bytes = (size_t)(atoi(g) > 0 ? atoi(g) : 1) << 20;
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
    // (sized for the narrowest column tiles any second product uses: k_zgemm_tri32's 32 -- it can be selected above
    // N = 768 too, by QUFLOW_HIP_TRI_MIN_N or qf_fixedpoint_products, where the default kernels' tiles are 64 wide)
    {
        const int slots32 = (N + 31) / 32;
        QF_CREATE_HIP(hipMalloc((void **)&ctx->rowpart, (size_t)(ctx->rowpart_tiles > slots32 ? ctx->rowpart_tiles : slots32) * N * sizeof(double)));
    }
    QF_CREATE_HIP(hipMalloc((void **)&ctx->rowsum, (size_t)N * sizeof(double)));
    QF_CREATE_HIP(hipMalloc((void **)&ctx->scalars, 4096 * sizeof(double)));
    QF_CREATE_HIP(hipHostMalloc((void **)&ctx->host_scalars, 64 * sizeof(double), hipHostMallocDefault));
    QF_CREATE_HIP(hipMalloc((void **)&ctx->state, sizeof(qf_dev_state)));
    QF_CREATE_HIP(hipMemsetAsync(ctx->state, 0, sizeof(qf_dev_state), ctx->stream));
    // [0] rows done, [1 + y] blocks of row y (k_update); [600..632] k_call_begin, [640..672] k_inner2 (group counters)
    QF_CREATE_HIP(hipMalloc((void **)&ctx->ticket, 704 * sizeof(unsigned)));
    QF_CREATE_HIP(hipMemsetAsync(ctx->ticket, 0, 704 * sizeof(unsigned), ctx->stream));
    // coherent (fine-grained) pinned memory: device stores become visible to the polling host
    QF_CREATE_HIP(hipHostMalloc((void **)&ctx->host_rec, sizeof(qf_host_record), hipHostMallocCoherent));
    memset(ctx->host_rec, 0, sizeof(qf_host_record));
    {   // stream-K exchange area of the upper-triangle second product (exact 64x64 tilings, 3M kernel)
        hipDeviceProp_t prop;
        QF_CREATE_HIP(hipGetDeviceProperties(&prop, device));
        ctx->num_cus = prop.multiProcessorCount;
        if (N % 64 == 0 && ctx->num_cus > 0) {
            ctx->sk_slots = ctx->num_cus;          // one 64 KiB slot per workgroup of the contiguous partition
            QF_CREATE_HIP(hipMalloc((void **)&ctx->sk_partial, (size_t)ctx->sk_slots * 64 * 64 * sizeof(cplx)));
            // [num_cus] piece flags + 1 epilogue ticket (fused step end)
            QF_CREATE_HIP(hipMalloc((void **)&ctx->sk_flags, (size_t)(ctx->sk_slots + 16) * sizeof(unsigned)));
            QF_CREATE_HIP(hipMemsetAsync(ctx->sk_flags, 0, (size_t)(ctx->sk_slots + 16) * sizeof(unsigned), ctx->stream));
        }
    }
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
                    ctx->lap, ctx->lap_user, ctx->poisson.tab, ctx->rowpart, ctx->rowsum,
                    ctx->t32_partial, ctx->t32_arrive, ctx->W2, ctx->Whalf2, ctx->ns_inv, ctx->ns_tmp, ctx->multi_rowpart, ctx->scalars, ctx->sk_partial, ctx->sk_flags, ctx->basis, ctx->sh_stage, ctx->sh_omega};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (cplx *p : ctx->multi)
        if (p) (void)hipFree(p);
    for (int q = 0; q < 4; ++q) {
        if (ctx->oz_planes[q]) (void)hipFree(ctx->oz_planes[q]);
        if (ctx->oz_scale[q]) (void)hipFree(ctx->oz_scale[q]);
        if (q == 0 && ctx->oz_tbuf) (void)hipFree(ctx->oz_tbuf);
        if (q == 0 && ctx->oz_tflags) (void)hipFree(ctx->oz_tflags);
        if (q == 0 && ctx->oz_diag) (void)hipFree(ctx->oz_diag);
    }
    for (auto &kv : ctx->user_factors) {
        if (kv.second.f.tab) (void)hipFree(kv.second.f.tab);
    }
    for (int q = 0; q < 3; ++q)
        if (ctx->hook_host[q]) (void)hipHostFree(ctx->hook_host[q]);
    qf_c64_free(ctx->c64);
    ctx->c64 = nullptr;
    if (ctx->host_scalars) (void)hipHostFree(ctx->host_scalars);
    if (ctx->host_rec) (void)hipHostFree(ctx->host_rec);
    if (ctx->state) (void)hipFree(ctx->state);
    if (ctx->ticket) (void)hipFree(ctx->ticket);
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


int qf_upload_W(qf_ctx *ctx, const void *W_host)
{
    QF_TRY(check_ctx(ctx));
    if (!W_host) {
        qf_set_error("qf_upload_W: null buffer");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipMemcpyAsync(ctx->W, W_host, (size_t)ctx->N * ctx->N * sizeof(cplx), hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->w_skew_known = false;
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


int qf_c64_upload_W(qf_ctx *ctx, const void *W_host)
{
    QF_TRY(qf_need_c64(ctx));
    if (!W_host) {
        qf_set_error("qf_c64_upload_W: null buffer");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipMemcpyAsync(ctx->c64->W, W_host, (size_t)ctx->N * ctx->N * sizeof(float2), hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    ctx->c64->increment_valid = false;
    ctx->c64->w_skew_known = false;
    return QF_OK;
}

int qf_c64_download_W(qf_ctx *ctx, void *W_host)
{
    QF_TRY(qf_need_c64(ctx));
    if (!W_host) {
        qf_set_error("qf_c64_download_W: null buffer");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipMemcpyAsync(W_host, ctx->c64->W, (size_t)ctx->N * ctx->N * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
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
        ctx->prof_seen[i] = 0;
        ctx->prof_ms[i] = 0.0;
    }
    return QF_OK;
}

int qf_profile_stride(qf_ctx *ctx, int stride)
{
    QF_TRY(check_ctx(ctx));
    if (stride < 1) {
        qf_set_error("qf_profile_stride: stride must be >= 1");
        return QF_ERR_INVALID;
    }
    ctx->profile_stride = stride;
    return QF_OK;
}

int qf_profile_seen(qf_ctx *ctx, int kernel_id, long long *seen)
{
    QF_TRY(check_ctx(ctx));
    if (kernel_id < 0 || kernel_id >= QF_KERNEL_COUNT || !seen) {
        qf_set_error("qf_profile_seen: bad argument");
        return QF_ERR_INVALID;
    }
    *seen = ctx->prof_seen[kernel_id];
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

int qf_debug_modulus(qf_ctx *ctx, int n, const double *er_host, const double *ei_host, double *out_modulus_host,
                     double *out_sqrt_host)
{
    QF_TRY(check_ctx(ctx));
    if (n < 1 || (size_t)n > (size_t)ctx->N * ctx->N || !er_host || !ei_host || !out_modulus_host || !out_sqrt_host) {
        qf_set_error("qf_debug_modulus: bad arguments (n=%d)", n);
        return QF_ERR_INVALID;
    }
    // staging: four real vectors of n <= N^2 doubles in the two staging matrices (2 N^2 doubles each)
    double *d = reinterpret_cast<double *>(ctx->stage), *o = reinterpret_cast<double *>(ctx->PW);
    QF_HIP(hipMemcpyAsync(d, er_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(d + n, ei_host, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_debug_modulus(ctx, n, d, d + n, o, o + n));
    QF_HIP(hipMemcpyAsync(out_modulus_host, o, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipMemcpyAsync(out_sqrt_host, o + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}


}  // extern "C"
