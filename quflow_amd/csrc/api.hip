// C ABI of libquflow_hip.so (see include/quflow_hip.h) and the host-side control flow
// of the isospectral midpoint stepper (quflow/integrators/isospectral.py:338-613).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>

#include "qf_internal.h"

#include <sched.h>

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
        ctx->plan_role = id;       // (qf_plan_note: the launchers inside this scope describe what they launch for it)
        if (!active) return;
        // sampling: one launch in profile_stride carries the event pair (the events themselves cost
        // host time and stream slots: ~6 % of the step rate when every product launch is bracketed)
        if ((ctx->prof_seen[id]++ % ctx->profile_stride) != 0) {
            active = false;
            return;
        }
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
        ctx->plan_role = -1;
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
    QF_HIP(hipMalloc((void **)&f->tab, NN * sizeof(double2)));
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

// isomp_fixedpoint, quflow/integrators/isospectral.py:338-613 (autonomous, built-in Hamiltonian).
//
// Control flow.  The reference decides after every iteration on the host whether to break
// (isospectral.py:535).  Here that decision is taken on the device (k_norm_decide) and every
// hot-path launch is tagged (step, iteration): a launch whose tag is not due is a no-op.  The
// host therefore never waits for a residual: it enqueues `pred` iterations per step (the
// count recent steps needed) and the step-end update, whose last block publishes progress
// to pinned host memory, and runs up to QF_RUN_AHEAD steps ahead of what it has seen finish.
//   * a step that converges earlier: its surplus iteration launches are no-ops;
//   * a step that needs more: its update/advance and everything enqueued behind it are
//     no-ops (the state is untouched); the host notices (advance executed, step counter did
//     not move), enqueues the remaining iterations of that step and re-enqueues what followed.
// Either way the arithmetic performed is exactly the reference's iteration sequence.
#ifndef QF_RUN_AHEAD
#define QF_RUN_AHEAD 3
#endif

// The second product of an iteration is skew-Hermitian when W is (Phalf always is: k_solve
// mirrors, cpu.py:334,340): then only its upper triangle is multiplied (k_zgemm_tri), which also
// mirrors Whalf and the residual sums and therefore wants W[j,i] == -conj(W[i,j]) EXACTLY -- what
// A - A^H, the reference's own initial data and every isomp update (conj_subtract_) produce.  Any
// other W takes the full product, as the reference's np.matmul does.
static int oz_alloc(qf_ctx *ctx);

// exchange area of k_zgemm_tri32 (allocated when first chosen): two parked half-K partial tiles and one
// arrival counter per upper-triangle tile; K split in two where that keeps the grid within the CUs
static int tri32_alloc(qf_ctx *ctx)
{
    const int nt = (ctx->N + 31) / 32;
    const int n_tiles = nt * (nt + 1) / 2;
    if (!ctx->t32_partial) {
        QF_HIP(hipMalloc((void **)&ctx->t32_partial, (size_t)n_tiles * 4 * 32 * 32 * sizeof(cplx)));
        QF_HIP(hipMalloc((void **)&ctx->t32_arrive, (size_t)n_tiles * sizeof(unsigned)));
        QF_HIP(hipMemsetAsync(ctx->t32_arrive, 0, (size_t)n_tiles * sizeof(unsigned), ctx->stream));
        int so = 2, sd = 1;
        {
            // one workgroup per CU at most.  Measured (tools/gemm_time.hip, fused step end): N=512 26.8 us with (2,1) = 256
            // workgroups against 27.4 with (2,2) = 272 and 28.5 for the full product; N=256 17.0 with (2,2) = 72
            // workgroups against 18.2 with (2,1) and 17.8 for the full product.
            // Round 3, four pieces per tile ((4,2) = 512 workgroups at N = 512, two per CU -- they
            // drift apart, and one's exchange and epilogue run under the other's K loop): a single trajectory gains 2 %
            // (9,427 against 9,217 timesteps/s; (4,4) = 544 workgroups: 8,618), but k replicas per GPU lose what the
            // extra exchange costs once the replicas fill the CUs anyway (k = 4: sum 15,074 against 17,875; k = 2: 12,599
            // against 14,656) -- and an ensemble member must run its single-trajectory launches to stay bit-identical
            // to its own run.  The default stays (qf_fixedpoint_products takes the split as an argument for the parity tests).
            const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
            if (nt * (nt - 1) + 2 * nt <= cus) sd = 2;
            else if (nt * (nt - 1) + nt > cus) so = 1;
        }
        ctx->tri32_split = so;
        ctx->tri32_split_diag = sd;
    }
    return QF_OK;
}

static int select_second_product(qf_ctx *ctx)
{
    ctx->gemm_tri = false;
    ctx->gemm_tri32 = false;
    ctx->gemm_i8 = false;
    const bool want_tri = ctx->gemm_tri_allowed && ctx->sk_partial && ctx->N >= ctx->gemm_tri_min_n;
    // below that size: the upper triangle of 32x32 tiles, K split over two workgroups (k_zgemm_tri32)
    // (and wherever the stream-K form is not available: N a multiple of 32 but not of 64, at any size)
    // (any N: edge tiles are guarded when N is no multiple of 32)
    // (its exchange area is addressed through ONE buffer resource with a 32-bit offset: 4 slots of 16 KiB per tile must
    // stay below 2 GiB -- nt <= 255, N <= 8160; past that the full product, rather than stores the hardware would drop)
    const size_t nt32 = (size_t)(ctx->N + 31) / 32;
    const bool tri32_fits = nt32 * (nt32 + 1) / 2 * 4 * 32 * 32 * sizeof(cplx) <= (size_t)0x7fffffff;
    const bool want_tri32 = ctx->gemm_tri_allowed && !want_tri && ctx->N >= 64 && tri32_fits;
    const bool want_i8 = ctx->gemm_i8_allowed && ctx->N % 64 == 0 && ctx->N >= ctx->gemm_i8_min_n && ctx->N <= 4096;   // k_oz_slice: one lane per 4 entries of a row
    if (!want_tri && !want_i8 && !want_tri32) return QF_OK;
    // (a state this stepper produced from a skew-Hermitian one is skew-Hermitian: W += 2 (PW - PW^H)
    // keeps the property exactly, so only the first call on an uploaded state pays for the check)
    if (!ctx->w_skew_known) {
        QF_TRY(qf_launch_skew_defect(ctx, ctx->W, ctx->scalars + 4));
        QF_HIP(hipMemcpyAsync(ctx->host_scalars, ctx->scalars + 4, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        QF_HIP(hipStreamSynchronize(ctx->stream));
        ctx->w_skew_known = (ctx->host_scalars[0] == 0.0);
    }
    const bool skew = ctx->w_skew_known;
    ctx->gemm_tri = want_tri && skew;
    ctx->gemm_i8 = want_i8 && skew;      // the sliced right operands are built from rows: B^T = -conj(B)
    ctx->gemm_tri32 = want_tri32 && skew && (!ctx->gemm_i8 || ctx->gemm_i8_first);
    if (ctx->gemm_tri32) QF_TRY(tri32_alloc(ctx));
    return QF_OK;
}

// column-tile slots of the partial row sums the second product writes (its tile width differs
// between the full kernel's size classes and the 64-wide upper-triangle form)
static int rowpart_slots(const qf_ctx *ctx) { return ctx->gemm_tri ? ctx->N / 64 : ctx->gemm_tri32 ? (ctx->N + 31) / 32 : ctx->rowpart_tiles; }

static int enqueue_iterations(qf_ctx *ctx, int step, int first, int count, double vareps)
{
    for (int i = first; i < first + count; ++i) {
        qf_guard g;
        g.state = ctx->state;
        g.step = step;
        g.iter = i;
        {   // Phalf = vareps * solve_poisson(Whalf)          isospectral.py:488-492
            prof_scope p(ctx, QF_KERNEL_POISSON);
            QF_TRY(qf_launch_solve(ctx, ctx->poisson, ctx->Whalf, ctx->Phalf, vareps, 1, g));
        }
        {   // PW = Phalf @ Whalf                              isospectral.py:496
            prof_scope p(ctx, QF_KERNEL_GEMM1);
            QF_TRY(qf_launch_zgemm(ctx, ctx->Phalf, ctx->Whalf, ctx->PW, nullptr, g));
        }
        {   // dW = PW @ Phalf + (PW - PW^H); Whalf = W + dW; row sums of |dW_old - dW|
            prof_scope p(ctx, QF_KERNEL_GEMM2);
            qf_epilogue ep;
            ep.PW = ctx->PW;
            ep.W = ctx->W;
            ep.dW[0] = ctx->dW[0];
            ep.dW[1] = ctx->dW[1];
            ep.Whalf = ctx->Whalf;
            ep.rowpart = ctx->rowpart;
            QF_TRY(qf_launch_zgemm(ctx, ctx->PW, ctx->Phalf, nullptr, &ep, g));
        }
        {   // residual norm + break decision                  isospectral.py:523-536
            prof_scope p(ctx, QF_KERNEL_NORM);
            QF_TRY(qf_launch_norm_decide(ctx, ctx->rowpart, rowpart_slots(ctx), g));
        }
    }
    return QF_OK;
}

// the same iteration on complex64 data: float32 solve, complex64 products on the fp32 matrix cores (single.hip),
// the exit decision on the double row sums the second product's epilogue leaves
static int enqueue_iterations_c64(qf_ctx *ctx, int step, int first, int count, double vareps)
{
    qf_c64 *f = ctx->c64;
    for (int i = first; i < first + count; ++i) {
        qf_guard g;
        g.state = ctx->state;
        g.step = step;
        g.iter = i;
        {   // Phalf = vareps * solve_poisson(Whalf): the scale is applied in float32, as `Phalf *= vareps` on a
            // complex64 array is (isospectral.py:488-492)
            prof_scope p(ctx, QF_KERNEL_POISSON);
            QF_TRY(qf_launch_solve_f32(ctx, f->tab, f->Whalf, f->Phalf, (float)vareps, 1, g));
        }
        {
            prof_scope p(ctx, QF_KERNEL_GEMM1);
            QF_TRY(qf_launch_cgemm(ctx, f->Phalf, f->Whalf, f->PW, nullptr, g));
        }
        {
            prof_scope p(ctx, QF_KERNEL_GEMM2);
            qf_epilogue_f ep;
            ep.PW = f->PW;
            ep.W = f->W;
            ep.dW[0] = f->dW[0];
            ep.dW[1] = f->dW[1];
            ep.Whalf = f->Whalf;
            ep.rowpart = f->rowpart;
            QF_TRY(qf_launch_cgemm(ctx, f->PW, f->Phalf, nullptr, &ep, g));
        }
        {
            prof_scope p(ctx, QF_KERNEL_NORM);
            QF_TRY(qf_launch_norm_decide(ctx, f->rowpart, f->rowpart_tiles, g));
        }
    }
    return QF_OK;
}

// complex64 data with the fused step end: three launches per iteration, the step's W update and
// the exit decision in the second product's epilogue / last tile
static int enqueue_iterations_fused_c64(qf_ctx *ctx, int step, int first, int count, double vareps, bool last_step = false)
{
    qf_c64 *f = ctx->c64;
    for (int i = first; i < first + count; ++i) {
        qf_guard g;
        g.state = ctx->state;
        g.step = step;
        g.iter = i;
        g.alt = f->Whalf2;       // read instead of Whalf when the previous iteration closed a step
        {
            prof_scope p(ctx, QF_KERNEL_POISSON);
            QF_TRY(qf_launch_solve_f32(ctx, f->tab, f->Whalf, f->Phalf, (float)vareps, 1, g));
        }
        {
            prof_scope p(ctx, QF_KERNEL_GEMM1);
            QF_TRY(qf_launch_cgemm(ctx, f->Phalf, f->Whalf, f->PW, nullptr, g));
        }
        {
            prof_scope p(ctx, QF_KERNEL_GEMM2);
            qf_epilogue_f ep;
            ep.PW = f->PW;
            ep.W = f->W;
            ep.dW[0] = f->dW[0];
            ep.dW[1] = f->dW[1];
            ep.Whalf = f->Whalf;
            ep.rowpart = f->rowpart;
            ep.fused = 1;
            ep.Wpair[0] = f->W;
            ep.Wpair[1] = f->W2;
            ep.Whalf_step = f->Whalf2;
            g.alt = nullptr;
            if (f->tri) QF_TRY(qf_launch_cgemm_tri(ctx, f->PW, f->Phalf, &ep, g));     // skew-Hermitian W: upper triangle only
            else QF_TRY(qf_launch_cgemm(ctx, f->PW, f->Phalf, nullptr, &ep, g));
        }
    }
    (void)last_step;
    return QF_OK;
}

// Fused step end: with the upper-triangle second product the step's
// W update and the exit decision live in that product's epilogue / last finisher, so an
// iteration is three launches and a step has no launches of its own.
static qf_decide deferred_decision(qf_ctx *ctx)
{
    qf_decide d;
    d.rowpart = ctx->rowpart;
    d.slots = ctx->gemm_tri32 ? (ctx->N + 31) / 32 : ctx->N / 64;       // column tiles of k_zgemm_tri32 / k_zgemm_tri
    d.state_rw = ctx->state;
    d.rec = ctx->host_rec;
    d.ticket = ctx->ticket + 402;        // (400: k_zgemm<.., FUSED>, 401: k_zgemm_tri32's own step end)
    return d;
}

static int enqueue_iterations_fused(qf_ctx *ctx, int step, int first, int count, double vareps, bool last_step = false)
{
    const qf_decide dec = deferred_decision(ctx);
    for (int i = first; i < first + count; ++i) {
        qf_guard g;
        g.state = ctx->state;
        g.step = step;
        g.iter = i;
        g.alt = ctx->Whalf2;     // read instead of Whalf when the previous iteration closed a step
        {
            prof_scope p(ctx, QF_KERNEL_POISSON);
            QF_TRY(qf_launch_solve(ctx, ctx->poisson, ctx->Whalf, ctx->Phalf, vareps, 1, g, ctx->defer ? &dec : nullptr));
        }
        {
            prof_scope p(ctx, QF_KERNEL_GEMM1);
            QF_TRY(qf_launch_zgemm(ctx, ctx->Phalf, ctx->Whalf, ctx->PW, nullptr, g));
        }
        {
            prof_scope p(ctx, QF_KERNEL_GEMM2);
            qf_epilogue ep;
            ep.PW = ctx->PW;
            ep.W = ctx->W;
            ep.dW[0] = ctx->dW[0];
            ep.dW[1] = ctx->dW[1];
            ep.Whalf = ctx->Whalf;
            ep.rowpart = ctx->rowpart;
            ep.fused = 1;
            ep.Wpair[0] = ctx->W;
            ep.Wpair[1] = ctx->W2;
            ep.Whalf_step = ctx->Whalf2;
            g.alt = nullptr;
            // upper-triangle stream-K kernel for skew-Hermitian W (N >= 768), else the full product
            QF_TRY(qf_launch_zgemm(ctx, ctx->PW, ctx->Phalf, nullptr, &ep, g));
        }
    }
    // deferred step end: the decision of an iteration is taken by the next solve; behind the last step's
    // iterations there is none, so a one-workgroup launch takes it (a no-op when nothing is pending)
    if (ctx->defer && last_step && count > 0) QF_TRY(qf_launch_decide(ctx, dec));
    return QF_OK;
}

// the same iteration with both products on the int8 matrix cores (ozaki.hip): the operands are
// cut into digit planes first (Phalf in both forms and Whalf in one launch, PW in another)
static int enqueue_iterations_fused_i8(qf_ctx *ctx, int step, int first, int count, double vareps)
{
    for (int i = first; i < first + count; ++i) {
        qf_guard g;
        g.state = ctx->state;
        g.step = step;
        g.iter = i;
        g.alt = ctx->Whalf2;
        {
            prof_scope p(ctx, QF_KERNEL_POISSON);
            QF_TRY(qf_launch_solve(ctx, ctx->poisson, ctx->Whalf, ctx->Phalf, vareps, 1, g));
        }
        if (ctx->gemm_i8_hybrid) {
            // hybrid: PW = Phalf @ Whalf in fp64 (k_zgemm), then PW and Phalf are sliced in one launch for the
            // digit-split second product
            {
                prof_scope p(ctx, QF_KERNEL_GEMM1);
                QF_TRY(qf_launch_zgemm(ctx, ctx->Phalf, ctx->Whalf, ctx->PW, nullptr, g));
            }
            g.alt = nullptr;
            {
                prof_scope p(ctx, QF_KERNEL_SLICE);
                qf_oz_jobs jobs;
                jobs.n = 2;
                jobs.j[0].X = ctx->PW;
                jobs.j[0].planes = ctx->oz_planes[3];
                jobs.j[0].scale = ctx->oz_scale[3];
                jobs.j[1].X = ctx->Phalf;
                jobs.j[1].planes = ctx->oz_planes[0];
                jobs.j[1].scale = ctx->oz_scale[0];
                QF_TRY(qf_launch_oz_slice(ctx, jobs, g));
            }
            {
                prof_scope p(ctx, QF_KERNEL_GEMM2);
                qf_epilogue ep;
                ep.PW = ctx->PW;
                ep.W = ctx->W;
                ep.dW[0] = ctx->dW[0];
                ep.dW[1] = ctx->dW[1];
                ep.Whalf = ctx->Whalf;
                ep.rowpart = ctx->rowpart;
                ep.fused = 1;
                ep.Wpair[0] = ctx->W;
                ep.Wpair[1] = ctx->W2;
                ep.Whalf_step = ctx->Whalf2;
                QF_TRY(qf_launch_oz_gemm(ctx, ctx->oz_planes[3], ctx->oz_scale[3], ctx->oz_planes[0], ctx->oz_scale[0], nullptr,
                                         &ep, g));
            }
            continue;
        }
        g.alt = nullptr;
        {
            prof_scope p(ctx, QF_KERNEL_SLICE);
            qf_oz_jobs jobs;
            jobs.n = 2;
            jobs.j[0].X = ctx->Phalf;                 // left operand of the first product, right one of the second
            jobs.j[0].planes = ctx->oz_planes[0];
            jobs.j[0].scale = ctx->oz_scale[0];
            jobs.j[1].X = ctx->Whalf;                 // right operand of the first product
            jobs.j[1].X_alt = ctx->Whalf2;
            jobs.j[1].planes = ctx->oz_planes[2];
            jobs.j[1].scale = ctx->oz_scale[2];
            jobs.diag = ctx->oz_diag;                 // Im (Phalf @ Whalf)_ii in fp64: tr (PW - PW^H) on the fp64 products' line
            QF_TRY(qf_launch_oz_slice(ctx, jobs, g));
        }
        {
            prof_scope p(ctx, QF_KERNEL_GEMM1);
            QF_TRY(qf_launch_oz_gemm(ctx, ctx->oz_planes[0], ctx->oz_scale[0], ctx->oz_planes[2], ctx->oz_scale[2],
                                     ctx->PW, nullptr, g, 0, 0, ctx->oz_diag));
        }
        if (ctx->gemm_i8_first) {
            // the second product on the fp64 matrix cores: the upper-triangle kernels (k_zgemm_tri / k_zgemm_tri32) read
            // PW and Phalf as they are
            prof_scope p(ctx, QF_KERNEL_GEMM2);
            qf_epilogue ep;
            ep.PW = ctx->PW;
            ep.W = ctx->W;
            ep.dW[0] = ctx->dW[0];
            ep.dW[1] = ctx->dW[1];
            ep.Whalf = ctx->Whalf;
            ep.rowpart = ctx->rowpart;
            ep.fused = 1;
            ep.Wpair[0] = ctx->W;
            ep.Wpair[1] = ctx->W2;
            ep.Whalf_step = ctx->Whalf2;
            QF_TRY(qf_launch_zgemm(ctx, ctx->PW, ctx->Phalf, nullptr, &ep, g));
            continue;
        }
        {
            prof_scope p(ctx, QF_KERNEL_SLICE);
            qf_oz_jobs jobs;
            jobs.n = 1;
            jobs.j[0].X = ctx->PW;                    // left operand of the second product
            jobs.j[0].planes = ctx->oz_planes[3];
            jobs.j[0].scale = ctx->oz_scale[3];
            QF_TRY(qf_launch_oz_slice(ctx, jobs, g, ctx->oz_digits2));
        }
        {
            prof_scope p(ctx, QF_KERNEL_GEMM2);
            qf_epilogue ep;
            ep.PW = ctx->PW;
            ep.W = ctx->W;
            ep.dW[0] = ctx->dW[0];
            ep.dW[1] = ctx->dW[1];
            ep.Whalf = ctx->Whalf;
            ep.rowpart = ctx->rowpart;
            ep.fused = 1;
            ep.Wpair[0] = ctx->W;
            ep.Wpair[1] = ctx->W2;
            ep.Whalf_step = ctx->Whalf2;
            QF_TRY(qf_launch_oz_gemm(ctx, ctx->oz_planes[3], ctx->oz_scale[3], ctx->oz_planes[0], ctx->oz_scale[0], nullptr,
                                     &ep, g, ctx->oz_digits2, ctx->oz_digits2 ? ctx->oz_digits : 0));
        }
    }
    return QF_OK;
}

// host side of the fused protocol: enqueue `pred` iterations per step up to QF_RUN_AHEAD steps
// ahead, poll the 8-byte progress word (steps << 32 | iterations of the current step).
// A state machine (begin / pump) so that one host thread can drive several independent
// trajectories -- one context and stream each -- at the same time (qf_isomp_multi).
struct fused_run {
    qf_ctx *ctx = nullptr;
    int steps = 0, minit = 1, maxit = 1;
    double vareps = 0.0;
    int pred = 1, pred0 = 1;
    std::vector<int> enq_iters;
    int known = 0, enq = 0;
    bool first_seen = false;
    bool aborted = false;
    unsigned long long idle_polls = 0;

    bool c64 = false;
    int enqueue(int step, int first, int count)
    {
        if (c64) return enqueue_iterations_fused_c64(ctx, step, first, count, vareps, step == steps - 1);
        return ctx->gemm_i8 ? enqueue_iterations_fused_i8(ctx, step, first, count, vareps)
                            : enqueue_iterations_fused(ctx, step, first, count, vareps, step == steps - 1);
    }
    bool cold_start = true;
    void begin(qf_ctx *c, int steps_, int minit_, int maxit_, double vareps_, bool c64_ = false)
    {
        ctx = c;
        c64 = c64_;
        cold_start = c64_ ? c->c64_increment_is_zero : c->increment_is_zero;
        steps = steps_;
        minit = minit_;
        maxit = maxit_;
        vareps = vareps_;
        pred = ctx->pred_iters;
        if (pred < minit) pred = minit;
        if (pred > maxit) pred = maxit;
        // The first step of a call starts from dW = 0 (isospectral.py:430) and typically needs one
        // iteration more than the warm-started ones: it gets its own prediction (learned from the
        // previous call's first step).  A surplus iteration is three no-op launches; a missing one
        // drains the pipeline (~0.6 ms at N=1024: everything enqueued behind it was a no-op).
        pred0 = ctx->pred_first_iters > 0 ? ctx->pred_first_iters : pred + 1;
        if (pred0 < pred) pred0 = pred;
        if (pred0 > maxit) pred0 = maxit;
        enq_iters.assign((size_t)steps + 1, 0);
        known = enq = 0;
        first_seen = false;
        aborted = false;
        idle_polls = 0;
    }
    bool done() const { return known >= steps; }
    // enqueue what may be enqueued, look at the progress word once; never blocks
    int pump()
    {
        if (done()) return QF_OK;
        while (enq < steps && enq - known < QF_RUN_AHEAD) {
            const int n = (enq == 0 && cold_start) ? pred0 : pred;
            QF_TRY(enqueue(enq, 0, n));
            enq_iters[enq] = n;
            ++enq;
        }
        volatile qf_host_record *rec = ctx->host_rec;
        unsigned long long p = __atomic_load_n(&rec->progress, __ATOMIC_ACQUIRE);
        int ps = (int)(p >> 32), pi = (int)(p & 0xffffffffull);
        if (ps >= QF_STEP_ABORTED) {
            // the device closed the call on a non-finite residual (qf_step_end.h): what is queued are no-ops, nothing more
            // is enqueued; fused_leave reports it with W as the last completed step left it
            aborted = true;
            known = steps;
            return QF_OK;
        }
        if (!(ps > known || (ps == known && pi >= enq_iters[known]))) {
            // step `known` is neither over nor out of enqueued iterations yet
            if (++idle_polls > (1ull << 22)) {
                QF_HIP(hipStreamSynchronize(ctx->stream));   // also surfaces faults
                p = __atomic_load_n(&rec->progress, __ATOMIC_ACQUIRE);
                ps = (int)(p >> 32);
                pi = (int)(p & 0xffffffffull);
                if (ps >= QF_STEP_ABORTED) {
                    aborted = true;
                    known = steps;
                    return QF_OK;
                }
                if (!(ps > known || (ps == known && pi >= enq_iters[known]))) {
                    qf_set_error("qf_isomp: device progress stuck at step %d iteration %d (waiting for step %d)", ps, pi, known);
                    return QF_ERR_STATE;
                }
            } else {
                return QF_OK;
            }
        }
        idle_polls = 0;
        if (ps > known) {
            const int it = rec->last_step_iters;
            if (known == 0 && ps == 1 && !first_seen && cold_start) {
                first_seen = true;               // that was the cold first step: remember it separately
                if (it >= minit && it <= maxit) ctx->pred_first_iters = it;
            } else if (it >= minit && it <= maxit) {
                pred = it;
            }
            known = ps < enq ? ps : enq;
            if (done()) ctx->pred_iters = pred;
            return QF_OK;
        }
        // the step needs more iterations than were enqueued: everything behind them was a no-op
        const int have = enq_iters[known];
        if (have >= maxit) {
            qf_set_error("qf_isomp: step %d did not close after maxit=%d iterations (internal error)", known, maxit);
            return QF_ERR_STATE;
        }
        QF_TRY(enqueue(known, have, maxit - have));
        enq_iters[known] = maxit;
        enq = known + 1;
        if (pred < maxit) pred += 1;
        return QF_OK;
    }
};

// Between two looks at the progress record the host thread executes `pause` (a core per rank is the
// normal deployment: one process per GPU).  QUFLOW_HIP_POLL=yield gives the core away instead
// (sched_yield) for hosts where the ranks outnumber the cores they may use.
static bool poll_yields()
{
    static const int mode = [] {
        const char *e = getenv("QUFLOW_HIP_POLL");
        return (e && strcmp(e, "yield") == 0) ? 1 : 0;
    }();
    return mode == 1;
}

static inline void poll_relax()
{
    if (poll_yields()) {
        sched_yield();
        return;
    }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}

static int run_fused(qf_ctx *ctx, int steps, int minit, int maxit, double vareps, bool c64 = false)
{
    fused_run run;
    run.begin(ctx, steps, minit, maxit, vareps, c64);
    while (!run.done()) {
        QF_TRY(run.pump());
        poll_relax();
    }
    ctx->pred_iters = run.pred;
    return QF_OK;
}

static int enqueue_step_end(qf_ctx *ctx, int step, int compsum, int reinitialize, bool c64 = false)
{
    qf_guard g;
    g.state = ctx->state;
    g.step = step;
    if (c64) {
        qf_c64 *f = ctx->c64;
        prof_scope p(ctx, QF_KERNEL_UPDATE);
        QF_TRY(qf_launch_update_f32(ctx, f->PW, f->W, f->dW[0], f->dW[1], f->Whalf, compsum ? f->kahan_c : nullptr,
                                    reinitialize, g));
        return QF_OK;
    }
    {   // W += 2*(PW - PW^H) (Kahan if compsum); Whalf = W + dW     isospectral.py:547-592
        prof_scope p(ctx, QF_KERNEL_UPDATE);
        QF_TRY(qf_launch_update(ctx, ctx->PW, ctx->W, ctx->dW[0], ctx->dW[1], ctx->Whalf,
                                compsum ? ctx->kahan_c : nullptr, reinitialize, g));
    }
    return QF_OK;
}

// spin on the pinned record until `seq` advances have executed (the GPU is busy: no sleep)
static int wait_for_advance(qf_ctx *ctx, unsigned long long seq)
{
    volatile qf_host_record *rec = ctx->host_rec;
    unsigned long long spins = 0;
    while (__atomic_load_n(&rec->seq, __ATOMIC_ACQUIRE) < seq) {
        if (++spins > (1ull << 22)) {
            // nothing for a long time: fall back to a real synchronisation (also surfaces faults)
            const unsigned long long before = __atomic_load_n(&rec->seq, __ATOMIC_ACQUIRE);
            const hipError_t q = hipStreamQuery(ctx->stream);
            QF_HIP(hipStreamSynchronize(ctx->stream));
            if (getenv("QUFLOW_HIP_DEBUG"))
                fprintf(stderr, "[quflow_hip] wait_for_advance: spin limit; want seq %llu, saw %llu before sync (stream %s), %llu after; step_index %d incomplete %d\n",
                        seq, before, q == hipSuccess ? "idle" : "busy", (unsigned long long)rec->seq, rec->step_index, rec->incomplete);
            if (__atomic_load_n(&rec->seq, __ATOMIC_ACQUIRE) < seq) {
                qf_set_error("qf_isomp: device progress record stuck at %llu (< %llu)",
                             (unsigned long long)rec->seq, seq);
                return QF_ERR_STATE;
            }
            break;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return QF_OK;
}

static int isomp_impl(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum,
                      int reinitialize, qf_isomp_stats *stats_out, bool carry_increment, bool c64 = false);

// A call that ended in an error leaves its counters wherever the cut found them.  The flag is ONE per context, the
// counters are per working set (complex128 and complex64): whichever precision enters next rebuilds ALL of them --
// the monotone arrival counters of a triangle product are tested as `old % split`, so one left out of phase would
// make a later product of the other precision combine pieces before all are parked.
static int reset_after_abort(qf_ctx *ctx)
{
    if (!ctx->needs_reset) return QF_OK;
    QF_HIP(hipMemsetAsync(ctx->ticket, 0, 704 * sizeof(unsigned), ctx->stream));   // (a finished launch leaves them at 0)
    if (ctx->sk_flags) QF_HIP(hipMemsetAsync(ctx->sk_flags + (ctx->sk_slots > 0 ? ctx->sk_slots : ctx->num_cus), 0, 16 * sizeof(unsigned), ctx->stream));
    if (ctx->t32_arrive) {
        const size_t nt = (size_t)(ctx->N + 31) / 32;
        QF_HIP(hipMemsetAsync(ctx->t32_arrive, 0, nt * (nt + 1) / 2 * sizeof(unsigned), ctx->stream));
    }
    if (ctx->c64 && ctx->c64->tri_arrive)
        QF_HIP(hipMemsetAsync(ctx->c64->tri_arrive, 0, ctx->c64->tri_arrive_count * sizeof(unsigned), ctx->stream));
    ctx->needs_reset = false;
    return QF_OK;
}

// ---- entry and exit of a call in the fused protocol, shared by qf_isomp and qf_isomp_multi ----
// everything up to the first iteration launch: tolerance (formed on the device when automatic),
// choice of the product kernels, dW = 0 / Whalf = W (or the carried increment), control state
static int fused_enter(qf_ctx *ctx, double dt, double tol, int minit, int maxit, bool carry)
{
    const int N = ctx->N;
    const size_t mbytes = (size_t)N * N * sizeof(cplx);
    const double hb = qf_hbar(N);
    const bool tol_on_device = tol < 0;
    const double tol_factor = tol_on_device ? std::sqrt(std::numeric_limits<double>::epsilon()) * dt / hb : 0.0;   // isospectral.py:440-448 (no compsum here)
    QF_TRY(reset_after_abort(ctx));
    QF_TRY(select_second_product(ctx));
    // deferred step end: with k_zgemm_tri32 up to N = 512 (every workgroup of the deciding launch re-reads the
    // N x N/32 row sums: 64 KiB at N = 512)
    // (built for the stream-K product too, up to N = 1024: 2,575 against 2,587 timesteps/s; removed in round 5)
    ctx->defer = ctx->defer_allowed && !ctx->gemm_i8 && ctx->gemm_tri32 && ctx->N <= 512;
    ctx->increment_is_zero = !carry;
    ctx->increment_valid = true;
    // The host polls the pinned record: it resets the word it polls itself (nothing is in flight on
    // this stream that writes it: every call ends with a synchronisation), the device resets the rest
    // in stream order -- no wait between the two.
    volatile qf_host_record *rec = ctx->host_rec;
    rec->progress = 0ull;
    rec->step_index = 0;
    rec->fault = 0;
    rec->nonfinite = 0;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    if (carry) {
        if (tol_on_device) QF_TRY(qf_launch_norm_inf(ctx, ctx->W, ctx->scalars));
        if (ctx->dw_cur != 0)
            QF_HIP(hipMemcpyAsync(ctx->dW[0], ctx->dW[ctx->dw_cur], mbytes, hipMemcpyDeviceToDevice, ctx->stream));
        QF_TRY(qf_launch_lincomb(ctx, 1.0, ctx->W, 1.0, ctx->dW[0], 0.0, ctx->Whalf));
        QF_TRY(qf_launch_state_init(ctx, tol, minit, maxit, tol_on_device ? ctx->scalars : nullptr, tol_factor));
    } else {
        // dW = 0, Whalf = W, the norm for the tolerance and the control state: one launch
        QF_TRY(qf_launch_call_begin(ctx, tol, minit, maxit, tol_on_device ? 1 : 0, tol_factor));
    }
    if (!ctx->W2) QF_HIP(hipMalloc((void **)&ctx->W2, mbytes));
    if (!ctx->Whalf2) QF_HIP(hipMalloc((void **)&ctx->Whalf2, mbytes));
    if (ctx->gemm_i8) QF_TRY(oz_alloc(ctx));
    return QF_OK;
}

// after the last step has been seen complete: adopt the buffers the device ended in, restore the
// triangles the upper-triangle product skipped, synchronise, report
// P = solve_poisson(W); <W, P> and <W, W> in one pass; the two sums on their way to the pinned scalars
static int enqueue_diagnostics(qf_ctx *ctx)
{
    QF_TRY(qf_launch_solve(ctx, ctx->poisson, ctx->W, ctx->stage, 1.0, 1));
    QF_TRY(qf_launch_inner2(ctx, ctx->W, ctx->stage, ctx->scalars + 2));
    QF_HIP(hipMemcpyAsync(ctx->host_scalars, ctx->scalars + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    return QF_OK;
}

static int fused_leave(qf_ctx *ctx, int steps, qf_isomp_stats *stats_out)
{
    volatile qf_host_record *rec = ctx->host_rec;
    // everything the host needs came with the progress word (qf_fused_step_end publishes the
    // parities before it); steps == 0: nothing ran, the init kernel's values stand
    const int w_parity = steps > 0 ? rec->w_parity : 0, wh_sel = steps > 0 ? rec->wh_sel : 0;
    ctx->dw_cur = steps > 0 ? rec->dw_parity : 0;
    if (w_parity) {              // the state ended in the second buffer of the pair
        cplx *t = ctx->W;
        ctx->W = ctx->W2;
        ctx->W2 = t;
    }
    if (wh_sel) {                // keep "Whalf" = what the next iteration would read
        cplx *t = ctx->Whalf;
        ctx->Whalf = ctx->Whalf2;
        ctx->Whalf2 = t;
    }
    if ((ctx->gemm_tri || ctx->gemm_tri32) && (!ctx->gemm_i8 || ctx->gemm_i8_first) && steps > 0) {
        // the upper-triangle product leaves W and dW on and above the diagonal tiles only (zgemm.hip)
        QF_TRY(qf_launch_mirror_lower(ctx, ctx->W));
        QF_TRY(qf_launch_mirror_lower(ctx, ctx->dW[ctx->dw_cur]));
    }
    if (ctx->diag_at_exit) {
        // (the state is complete: both triangles of W are in place behind the mirror launches above)
        QF_TRY(enqueue_diagnostics(ctx));
        ctx->diag_valid = true;
    }
    QF_HIP(hipStreamSynchronize(ctx->stream));      // (also surfaces asynchronous faults)
    if (rec->nonfinite) {       // what scipy.linalg.norm raises in the reference's exit test (isospectral.py:534)
        // the device closed the call there (QF_STEP_ABORTED): W is the state after the last completed step (adopted and
        // mirrored above), the iteration vector of the broken step is not one to carry
        qf_set_error("array must not contain infs or NaNs");
        ctx->needs_reset = true;
        ctx->increment_valid = false;
        ctx->diag_valid = false;
        rec->nonfinite = 0;
        return QF_ERR_NONFINITE;
    }
    if (steps > 0 && rec->step_index != steps) {
        qf_set_error("qf_isomp: device completed %d of %d steps (internal error)", rec->step_index, steps);
        return QF_ERR_STATE;
    }
    if (rec->fault) {
        qf_set_error("qf_isomp: a device-side wait of the second product ran out (a parked partial tile or a mirrored result tile was never published)");
        return QF_ERR_STATE;
    }
    if (stats_out) {
        stats_out->total_iterations = steps > 0 ? rec->total_iterations : 0;
        stats_out->number_of_maxit = steps > 0 ? rec->number_of_maxit : 0;
        stats_out->tol_used = rec->tol;
        stats_out->last_resnorm = rec->resnorm;
    }
    return QF_OK;
}

// the same two for complex64 data (buffers of qf_c64; the control state, the record and the host protocol are shared)
static int fused_enter_c64(qf_ctx *ctx, double dt, double tol, int minit, int maxit, bool carry)
{
    qf_c64 *f = ctx->c64;
    const int N = ctx->N;
    const size_t fbytes = (size_t)N * N * sizeof(float2);
    const bool tol_on_device = tol < 0;
    // np.finfo(complex64).eps, its square root taken in float32 (isospectral.py:440-448, no compsum here)
    const double tol_factor = tol_on_device ? (double)std::sqrt(std::numeric_limits<float>::epsilon()) * dt / qf_hbar(N) : 0.0;
    QF_TRY(reset_after_abort(ctx));
    // the upper-triangle second product for an exactly skew-Hermitian state (checked once per uploaded state, as
    // select_second_product does for complex128 data)
    f->tri = false;
    if (f->tri_allowed && ctx->gemm_tri_allowed && (qf_c64_tile(ctx) == 32 || N % 64 == 0) && N >= 64) {      // (QUFLOW_HIP_GEMM2=full: A/B)
        if (!f->w_skew_known) {
            QF_TRY(qf_launch_skew_defect_f32(ctx, f->W, ctx->scalars + 4));
            QF_HIP(hipMemcpyAsync(ctx->host_scalars, ctx->scalars + 4, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            QF_HIP(hipStreamSynchronize(ctx->stream));
            f->w_skew_known = (ctx->host_scalars[0] == 0.0);
        }
        if (f->w_skew_known) {
            QF_TRY(qf_c64_tri_alloc(ctx));
            f->tri = true;
        }
    }
    // (the deferred exit decision of DESIGN.md 4f was built for complex64 too -- bit-identical, no gain: N = 512 23,780 against
    // 23,940 timesteps/s, the triangle product sheds 2.1 us, the solve takes 2.8 -- and removed in round 5)
    ctx->c64_increment_is_zero = !carry;
    volatile qf_host_record *rec = ctx->host_rec;
    rec->progress = 0ull;
    rec->step_index = 0;
    rec->fault = 0;
    rec->nonfinite = 0;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    if (tol_on_device) QF_TRY(qf_launch_norm_inf_f32(ctx, f->W, ctx->scalars));
    if (carry) {
        if (f->dw_cur != 0) QF_HIP(hipMemcpyAsync(f->dW[0], f->dW[f->dw_cur], fbytes, hipMemcpyDeviceToDevice, ctx->stream));
        QF_TRY(qf_launch_lincomb_f32(ctx, 1.0f, f->W, 1.0f, f->dW[0], f->Whalf));
    } else {
        QF_HIP(hipMemsetAsync(f->dW[0], 0, fbytes, ctx->stream));
        QF_HIP(hipMemcpyAsync(f->Whalf, f->W, fbytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    f->increment_valid = true;
    QF_TRY(qf_launch_state_init(ctx, tol, minit, maxit, tol_on_device ? ctx->scalars : nullptr, tol_factor));
    if (!f->W2) QF_HIP(hipMalloc((void **)&f->W2, fbytes));
    if (!f->Whalf2) QF_HIP(hipMalloc((void **)&f->Whalf2, fbytes));
    return QF_OK;
}

static int fused_leave_c64(qf_ctx *ctx, int steps, qf_isomp_stats *stats_out)
{
    qf_c64 *f = ctx->c64;
    volatile qf_host_record *rec = ctx->host_rec;
    const int w_parity = steps > 0 ? rec->w_parity : 0, wh_sel = steps > 0 ? rec->wh_sel : 0;
    f->dw_cur = steps > 0 ? rec->dw_parity : 0;
    if (w_parity) std::swap(f->W, f->W2);
    if (wh_sel) std::swap(f->Whalf, f->Whalf2);
    if (f->tri && steps > 0) {
        // the upper-triangle product leaves W and dW on and above the diagonal tiles only
        QF_TRY(qf_launch_mirror_lower_f32(ctx, f->W));
        QF_TRY(qf_launch_mirror_lower_f32(ctx, f->dW[f->dw_cur]));
    }
    QF_HIP(hipStreamSynchronize(ctx->stream));
    if (rec->nonfinite) {       // what scipy.linalg.norm raises in the reference's exit test (isospectral.py:534); see fused_leave
        qf_set_error("array must not contain infs or NaNs");
        ctx->needs_reset = true;
        f->increment_valid = false;
        rec->nonfinite = 0;
        return QF_ERR_NONFINITE;
    }
    if (steps > 0 && rec->step_index != steps) {
        qf_set_error("qf_isomp: device completed %d of %d steps (internal error)", rec->step_index, steps);
        return QF_ERR_STATE;
    }
    if (stats_out) {
        stats_out->total_iterations = steps > 0 ? rec->total_iterations : 0;
        stats_out->number_of_maxit = steps > 0 ? rec->number_of_maxit : 0;
        stats_out->tol_used = rec->tol;
        stats_out->last_resnorm = rec->resnorm;
    }
    return QF_OK;
}

// A call that ended in an error (a device-side wait ran out, the progress watchdog fired): drain the stream --
// what is still queued are tagged launches that are not due -- and mark the context for a rebuild of its
// counters at the next entry.  The state W is undefined after such a call (upload it again); the context
// itself stays usable and destroyable.  Never re-executes anything: an error return only.
static void fused_abort(qf_ctx *ctx)
{
    (void)hipStreamSynchronize(ctx->stream);
    ctx->needs_reset = true;
    ctx->increment_valid = false;
    ctx->w_skew_known = false;
    if (ctx->c64) {
        ctx->c64->w_skew_known = false;
        ctx->c64->increment_valid = false;
    }
    ctx->host_rec->fault = 0;
    ctx->host_rec->nonfinite = 0;
}

// k independent trajectories (one context -- buffers, control state, stream -- each) advanced by ONE
// host thread: every context runs exactly the launches qf_isomp would issue for it, so each result is
// bit-identical to its own qf_isomp call; the streams let the GPU overlap the replicas -- a kernel of
// one fills the dependent-launch gaps of another, and where a kernel leaves LDS and registers free
// (N < 768: one 50 KB workgroup per CU) two replicas' workgroups share the CUs and their matrix pipes.
int qf_isomp_multi(qf_ctx **ctxs, int k, double dt, int steps, double tol, int minit, int maxit, qf_isomp_stats *stats_out)
{
    if (!ctxs || k < 1) {
        qf_set_error("qf_isomp_multi: bad arguments (k=%d)", k);
        return QF_ERR_INVALID;
    }
    if (minit < 1) {
        qf_set_error("minit must be at least 1.");
        return QF_ERR_INVALID;
    }
    if (maxit < minit) {
        qf_set_error("maxit must be at minit.");
        return QF_ERR_INVALID;
    }
    if (steps < 0) {
        qf_set_error("qf_isomp_multi: steps must be >= 0");
        return QF_ERR_INVALID;
    }
    for (int r = 0; r < k; ++r) {
        QF_TRY(check_ctx(ctxs[r]));
        for (int q = 0; q < r; ++q)
            if (ctxs[q] == ctxs[r]) {
                qf_set_error("qf_isomp_multi: context %d is listed twice", r);
                return QF_ERR_INVALID;
            }
        if (ctxs[r]->device != ctxs[0]->device) {
            qf_set_error("qf_isomp_multi: the contexts live on different devices");
            return QF_ERR_INVALID;
        }
        if (!ctxs[r]->fused_allowed) {
            // (QUFLOW_HIP_FUSED=0 / QUFLOW_HIP_GEMM=4m A/B switches): one after the other
            for (int q = 0; q < k; ++q)
                QF_TRY(isomp_impl(ctxs[q], dt, steps, tol, minit, maxit, 0, 0, stats_out ? stats_out + q : nullptr, false));
            return QF_OK;
        }
    }
    std::vector<fused_run> runs((size_t)k);
    for (int r = 0; r < k; ++r) {
        QF_TRY(fused_enter(ctxs[r], dt, tol, minit, maxit, false));
        runs[r].begin(ctxs[r], steps, minit, maxit, dt / (2 * qf_hbar(ctxs[r]->N)));
    }
    for (;;) {
        bool all = true;
        for (int r = 0; r < k; ++r) {
            if (runs[r].done()) continue;
            const int rc = runs[r].pump();
            if (rc != QF_OK) {
                for (int q = 0; q < k; ++q) fused_abort(ctxs[q]);
                return rc;
            }
            all = all && runs[r].done();
        }
        if (all) break;
        poll_relax();
    }
    int first_rc = QF_OK;
    for (int r = 0; r < k; ++r) {
        const int rc = fused_leave(ctxs[r], steps, stats_out ? stats_out + r : nullptr);
        if (rc != QF_OK && first_rc == QF_OK) first_rc = rc;
    }
    return first_rc;
}

// the same for complex64 trajectories (qf_c64_upload_W states): the float32 launches qf_c64_isomp issues for each
int qf_c64_isomp_multi(qf_ctx **ctxs, int k, double dt, int steps, double tol, int minit, int maxit, qf_isomp_stats *stats_out)
{
    if (!ctxs || k < 1) {
        qf_set_error("qf_c64_isomp_multi: bad arguments (k=%d)", k);
        return QF_ERR_INVALID;
    }
    if (minit < 1) {
        qf_set_error("minit must be at least 1.");
        return QF_ERR_INVALID;
    }
    if (maxit < minit) {
        qf_set_error("maxit must be at minit.");
        return QF_ERR_INVALID;
    }
    if (steps < 0) {
        qf_set_error("qf_c64_isomp_multi: steps must be >= 0");
        return QF_ERR_INVALID;
    }
    bool together = true;
    for (int r = 0; r < k; ++r) {
        QF_TRY(check_ctx(ctxs[r]));
        if (!ctxs[r]->c64) {
            qf_set_error("qf_c64_isomp_multi: context %d holds no complex64 state (qf_c64_upload_W)", r);
            return QF_ERR_STATE;
        }
        for (int q = 0; q < r; ++q)
            if (ctxs[q] == ctxs[r]) {
                qf_set_error("qf_c64_isomp_multi: context %d is listed twice", r);
                return QF_ERR_INVALID;
            }
        if (ctxs[r]->device != ctxs[0]->device) {
            qf_set_error("qf_c64_isomp_multi: the contexts live on different devices");
            return QF_ERR_INVALID;
        }
        together = together && ctxs[r]->fused_allowed;
    }
    QF_HIP(hipSetDevice(ctxs[0]->device));
    if (!together) {
        for (int q = 0; q < k; ++q)
            QF_TRY(isomp_impl(ctxs[q], dt, steps, tol, minit, maxit, 0, 0, stats_out ? stats_out + q : nullptr, false, true));
        return QF_OK;
    }
    auto abort_all = [&] {
        for (int q = 0; q < k; ++q) {
            (void)hipStreamSynchronize(ctxs[q]->stream);
            ctxs[q]->needs_reset = true;
            ctxs[q]->c64->increment_valid = false;
        }
    };
    std::vector<fused_run> runs((size_t)k);
    for (int r = 0; r < k; ++r) {
        const int rc = fused_enter_c64(ctxs[r], dt, tol, minit, maxit, false);
        if (rc != QF_OK) {
            abort_all();
            return rc;
        }
        runs[r].begin(ctxs[r], steps, minit, maxit, dt / (2 * qf_hbar(ctxs[r]->N)), true);
    }
    for (;;) {
        bool all = true;
        for (int r = 0; r < k; ++r) {
            if (runs[r].done()) continue;
            const int rc = runs[r].pump();
            if (rc != QF_OK) {
                abort_all();
                return rc;
            }
            all = all && runs[r].done();
        }
        if (all) break;
        poll_relax();
    }
    int first_rc = QF_OK;
    for (int r = 0; r < k; ++r) {
        ctxs[r]->pred_iters = runs[r].pred;
        const int rc = fused_leave_c64(ctxs[r], steps, stats_out ? stats_out + r : nullptr);
        if (rc != QF_OK && first_rc == QF_OK) first_rc = rc;
    }
    if (first_rc != QF_OK) abort_all();
    return first_rc;
}

int qf_isomp(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum,
             int reinitialize, qf_isomp_stats *stats_out)
{
    return isomp_impl(ctx, dt, steps, tol, minit, maxit, compsum, reinitialize, stats_out, false);
}

int qf_isomp_continue(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum,
                      int reinitialize, qf_isomp_stats *stats_out)
{
    return isomp_impl(ctx, dt, steps, tol, minit, maxit, compsum, reinitialize, stats_out, true);
}

static int isomp_impl(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum,
                      int reinitialize, qf_isomp_stats *stats_out, bool carry_increment, bool c64)
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
    const bool dbg = getenv("QUFLOW_HIP_DEBUG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(now() - t).count();
    };
    const auto t_entry = now();
    double t_tol = 0, t_sel = 0, t_init = 0, t_loop = 0, t_waitmax = 0, t_waitat = 0, t_enqmax = 0, t_enqat = 0;
    int waitstep = -1, enqstep = -1;
    const int N = ctx->N;
    const size_t mbytes = (size_t)N * N * sizeof(cplx);
    const double hb = qf_hbar(N);          // isospectral.py:436
    const double vareps = dt / (2 * hb);   // isospectral.py:437

    // fused step end (either second-product kernel): plain W update, warm-started dW.  The norm for an
    // automatic tolerance stays on the device and the tolerance is formed there (k_state_init), no host
    // round trip; it comes back with the record.
    // (complex64 data: the two-kernel protocol on the float32 kernels)
    const bool fused = ctx->fused_allowed && !compsum && !reinitialize && !c64;
    if (c64 && ctx->fused_allowed && !compsum && !reinitialize) {
        QF_TRY(fused_enter_c64(ctx, dt, tol, minit, maxit, carry_increment && ctx->c64->increment_valid));
        t_init = ms_since(t_entry);
        int rc = run_fused(ctx, steps, minit, maxit, vareps, true);
        if (rc == QF_OK) rc = fused_leave_c64(ctx, steps, stats_out);
        if (rc != QF_OK) {
            (void)hipStreamSynchronize(ctx->stream);
            ctx->needs_reset = true;
            ctx->c64->increment_valid = false;
        }
        if (dbg)
            fprintf(stderr, "[quflow_hip] qf_c64_isomp %d steps (fused step end): init %.3f end %.3f ms (cumulative); %lld iterations\n",
                    steps, t_init, ms_since(t_entry), (long long)ctx->host_rec->total_iterations);
        return rc;
    }
    if (fused) {
        QF_TRY(fused_enter(ctx, dt, tol, minit, maxit, carry_increment && ctx->increment_valid));
        t_init = ms_since(t_entry);
        int rc = run_fused(ctx, steps, minit, maxit, vareps);
        if (rc != QF_OK) {
            fused_abort(ctx);
            return rc;
        }
        const double t_run = ms_since(t_entry);
        rc = fused_leave(ctx, steps, stats_out);
        if (rc != QF_OK) fused_abort(ctx);
        if (dbg)
            fprintf(stderr, "[quflow_hip] qf_isomp %d steps (fused step end): init %.3f run %.3f end %.3f ms (cumulative); %lld iterations\n",
                    steps, t_init, t_run, ms_since(t_entry), (long long)ctx->host_rec->total_iterations);
        return rc;
    }

    // tolerance, isospectral.py:440-452
    qf_c64 *f32 = c64 ? ctx->c64 : nullptr;
    if (tol < 0 && c64) {
        // the machine epsilon of the data's type (np.finfo(W.dtype).eps, :441), its square root taken in float32
        float mach_eps = std::numeric_limits<float>::epsilon();
        if (!compsum) mach_eps = std::sqrt(mach_eps);
        double nrm = 0.0;
        QF_TRY(qf_launch_norm_inf_f32(ctx, f32->W, ctx->scalars));
        QF_TRY(read_scalar(ctx, ctx->scalars, &nrm));
        tol = ((double)mach_eps * dt / hb) * nrm;
    } else if (tol < 0) {
        double mach_eps = std::numeric_limits<double>::epsilon();
        if (!compsum) mach_eps = std::sqrt(mach_eps);
        double nrm = 0.0;
        QF_TRY(qf_norm_inf_W(ctx, &nrm));
        tol = (mach_eps * dt / hb) * nrm;
    }

    t_tol = ms_since(t_entry);
    if (!c64) QF_TRY(select_second_product(ctx));
    t_sel = ms_since(t_entry);

    // dW = 0 at every entry (isospectral.py:430) => Whalf = W.  qf_isomp_continue: this call goes on
    // inside one call of the reference (host hooks between the steps): the increment of the
    // previous call on this context and the Kahan term carry over, Whalf = W + dW.
    if (c64) {
        const size_t fbytes = (size_t)N * N * sizeof(float2);
        const bool carry32 = carry_increment && f32->increment_valid && !reinitialize;
        if (carry32) {
            if (f32->dw_cur != 0)
                QF_HIP(hipMemcpyAsync(f32->dW[0], f32->dW[f32->dw_cur], fbytes, hipMemcpyDeviceToDevice, ctx->stream));
            QF_TRY(qf_launch_lincomb_f32(ctx, 1.0f, f32->W, 1.0f, f32->dW[0], f32->Whalf));
        } else {
            QF_HIP(hipMemsetAsync(f32->dW[0], 0, fbytes, ctx->stream));
            QF_HIP(hipMemcpyAsync(f32->Whalf, f32->W, fbytes, hipMemcpyDeviceToDevice, ctx->stream));
        }
        if (compsum) {
            const bool had = f32->kahan_c != nullptr;
            if (!f32->kahan_c) QF_HIP(hipMalloc((void **)&f32->kahan_c, fbytes));
            if (!(carry_increment && f32->increment_valid && had)) QF_HIP(hipMemsetAsync(f32->kahan_c, 0, fbytes, ctx->stream));
        }
        f32->increment_valid = true;
    }
    const bool carry = !c64 && carry_increment && ctx->increment_valid && !reinitialize;
    if (!c64) ctx->increment_is_zero = !carry;
    if (c64) {
        // (buffers prepared above)
    } else if (carry) {
        if (ctx->dw_cur != 0)
            QF_HIP(hipMemcpyAsync(ctx->dW[0], ctx->dW[ctx->dw_cur], mbytes, hipMemcpyDeviceToDevice, ctx->stream));
        QF_TRY(qf_launch_lincomb(ctx, 1.0, ctx->W, 1.0, ctx->dW[0], 0.0, ctx->Whalf));
    } else {
        QF_HIP(hipMemsetAsync(ctx->dW[0], 0, mbytes, ctx->stream));
        QF_HIP(hipMemcpyAsync(ctx->Whalf, ctx->W, mbytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (compsum && !c64) {
        const bool had = ctx->kahan_c != nullptr;
        if (!ctx->kahan_c) QF_HIP(hipMalloc((void **)&ctx->kahan_c, mbytes));
        // the compensation term lives for the whole reference call (isospectral.py:455-459): a continued call
        // keeps it whether or not `reinitialize` restarts the iteration vector every step (:471-472)
        if (!(carry_increment && ctx->increment_valid && had)) QF_HIP(hipMemsetAsync(ctx->kahan_c, 0, mbytes, ctx->stream));
    }
    if (!c64) ctx->increment_valid = true;
    ctx->gemm_i8 = false;        // the int8 products exist in the fused protocol only
    QF_TRY(qf_launch_state_init(ctx, tol, minit, maxit));
    // the init kernel must have reset the record before the host starts polling it
    QF_HIP(hipStreamSynchronize(ctx->stream));
    t_init = ms_since(t_entry);
    int pred = ctx->pred_iters;
    if (pred < minit) pred = minit;
    if (pred > maxit) pred = maxit;
    int hist[4] = {pred, pred, pred, pred};
    int hist_pos = 0;

    std::vector<unsigned long long> mark((size_t)steps + 1, 0);  // advance count after step s
    unsigned long long advances = 0;
    int known = 0;      // steps the host has seen complete
    int enq = 0;        // next step to enqueue
    int enq_iters_of_known = pred;  // iterations enqueued so far for step `known`
    std::vector<int> enq_iters((size_t)steps + 1, 0);
    volatile qf_host_record *rec = ctx->host_rec;

    while (known < steps) {
        while (enq < steps && enq - known < QF_RUN_AHEAD) {
            const auto te = now();
            QF_TRY(c64 ? enqueue_iterations_c64(ctx, enq, 0, pred, vareps) : enqueue_iterations(ctx, enq, 0, pred, vareps));
            QF_TRY(enqueue_step_end(ctx, enq, compsum, reinitialize, c64));
            if (dbg) {
                const double w = ms_since(te);
                if (w > t_enqmax) { t_enqmax = w; t_enqat = ms_since(t_entry); enqstep = enq; }
            }
            enq_iters[enq] = pred;
            mark[enq] = ++advances;
            ++enq;
        }
        {
            const auto tw = now();
            QF_TRY(wait_for_advance(ctx, mark[known]));
            const double w = ms_since(tw);
            if (w > t_waitmax) { t_waitmax = w; t_waitat = ms_since(t_entry); waitstep = known; }
        }
        if (rec->nonfinite) break;                // the device closed the call (k_norm_decide): reported below
        const int done_steps = rec->step_index;   // monotone; may already be ahead of `known`
        if (done_steps > known) {
            // learn from what the finished steps needed
            // predict the most recent count: an under-prediction costs one pipeline refill, an
            // over-prediction one no-op iteration (4 empty launches); outliers are rare
            const int it = rec->last_step_iters;
            if (it >= minit && it <= maxit) {
                hist[hist_pos++ & 3] = it;
                pred = it;
            }
            known = done_steps < enq ? done_steps : enq;
            continue;
        }
        // advance(known) ran but the step did not finish: it needs more iterations than were
        // enqueued.  Everything enqueued behind it was a no-op; supply the rest of this step
        // (guarded: surplus launches are no-ops) and re-enqueue the steps that followed.
        (void)enq_iters_of_known;
        const int have = enq_iters[known];
        if (have >= maxit) {
            qf_set_error("qf_isomp: step %d did not complete after maxit=%d iterations (internal error)", known, maxit);
            return QF_ERR_STATE;
        }
        // wait until the no-op tail has drained so that marks stay ordered
        QF_TRY(wait_for_advance(ctx, advances));
        QF_TRY(c64 ? enqueue_iterations_c64(ctx, known, have, maxit - have, vareps) : enqueue_iterations(ctx, known, have, maxit - have, vareps));
        QF_TRY(enqueue_step_end(ctx, known, compsum, reinitialize, c64));
        enq_iters[known] = maxit;
        mark[known] = ++advances;
        enq = known + 1;
        if (pred < maxit) pred += 1;
        for (int h = 0; h < 4; ++h) hist[h] = hist[h] < pred ? pred : hist[h];
    }
    ctx->pred_iters = pred;
    t_loop = ms_since(t_entry);

    QF_HIP(hipStreamSynchronize(ctx->stream));
    const double t_sync = ms_since(t_entry);
    qf_dev_state st;
    QF_HIP(hipMemcpy(&st, ctx->state, sizeof(st), hipMemcpyDeviceToHost));
    if (dbg)
        fprintf(stderr, "[quflow_hip] qf_isomp %d steps: tol %.3f sel %.3f init %.3f loop %.3f sync %.3f copy %.3f ms (cumulative); longest wait %.3f ms (step %d, ended at %.3f), longest enqueue %.3f ms (step %d, ended at %.3f)\n",
                steps, t_tol, t_sel, t_init, t_loop, t_sync, ms_since(t_entry), t_waitmax, waitstep, t_waitat, t_enqmax, enqstep, t_enqat);
    if (st.fault == QF_FAULT_NONFINITE || rec->nonfinite) {     // (k_norm_decide: isospectral.py:534)
        // W is the state after the last completed step (the broken step's update never ran); its iteration vector is not one to carry
        qf_set_error("array must not contain infs or NaNs");
        ctx->needs_reset = true;
        if (c64) f32->increment_valid = false;
        else ctx->increment_valid = false;
        rec->nonfinite = 0;
        return QF_ERR_NONFINITE;
    }
    if (st.step_index != steps) {
        qf_set_error("qf_isomp: device completed %d of %d steps (internal error)", st.step_index, steps);
        return QF_ERR_STATE;
    }
    if (c64) f32->dw_cur = st.dw_parity;
    else ctx->dw_cur = st.dw_parity;
    if (rec->fault) {
        qf_set_error("qf_isomp: a device-side wait of the second product ran out (a parked partial tile or a mirrored result tile was never published)");
        return QF_ERR_STATE;
    }
    if (stats_out) {
        stats_out->total_iterations = st.total_iterations;
        stats_out->number_of_maxit = st.number_of_maxit;
        stats_out->tol_used = tol;
        stats_out->last_resnorm = rec->resnorm;
    }
    return QF_OK;
}

// euler / heun / rk4 with the built-in Hamiltonian (quflow/integrators/erk.py:19-160).
// One right-hand side: P = solve_poisson(X); K = bracket(P, X) = (P@X - X@P)/hbar
// (quflow/geometry.py:41-49).  For skew-Hermitian data X@P = (P@X)^H: one product per stage.
int qf_erk(qf_ctx *ctx, int method, double dt, int steps, int skewh)
{
    QF_TRY(check_ctx(ctx));
    if (method < QF_ERK_EULER || method > QF_ERK_RK4) {
        qf_set_error("qf_erk: unknown method %d", method);
        return QF_ERR_INVALID;
    }
    if (steps < 0) {
        qf_set_error("qf_erk: steps must be >= 0");
        return QF_ERR_INVALID;
    }
    const double inv_hb = 1.0 / qf_hbar(ctx->N);
    ctx->w_skew_known = false;
    bool one_product = false;
    if (skewh) {
        QF_TRY(qf_launch_skew_defect(ctx, ctx->W, ctx->scalars + 4));
        QF_HIP(hipMemcpyAsync(ctx->host_scalars, ctx->scalars + 4, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        QF_HIP(hipStreamSynchronize(ctx->stream));
        one_product = (ctx->host_scalars[0] == 0.0);
    }
    cplx *W = ctx->W, *Wp = ctx->Whalf, *P = ctx->Phalf, *A = ctx->PW, *B = ctx->stage, *acc = ctx->dW[0];
    // K(X) into the stage kernel: products of P = Delta^-1 X with X
    auto products = [&](const cplx *X) -> int {
        {
            prof_scope p(ctx, QF_KERNEL_POISSON);
            QF_TRY(qf_launch_solve(ctx, ctx->poisson, X, P, 1.0, skewh ? 1 : 0));
        }
        {
            prof_scope p(ctx, QF_KERNEL_GEMM1);
            QF_TRY(qf_launch_zgemm(ctx, P, X, A, nullptr));
        }
        if (!one_product) {
            prof_scope p(ctx, QF_KERNEL_GEMM2);
            QF_TRY(qf_launch_zgemm(ctx, X, P, B, nullptr));
        }
        return QF_OK;
    };
    const cplx *Bk = one_product ? nullptr : B;
    auto stage = [&](cplx *acc_, double c_acc, cplx *Wp_, double c_wp, cplx *Wout_, double c_fin) -> int {
        prof_scope p(ctx, QF_KERNEL_UPDATE);
        return qf_launch_erk_stage(ctx, A, Bk, inv_hb, W, acc_, c_acc, Wp_, c_wp, Wout_, c_fin);
    };
    for (int k = 0; k < steps; ++k) {
        if (method == QF_ERK_EULER) {           // erk.py:53-56
            QF_TRY(products(W));
            QF_TRY(stage(nullptr, 0.0, nullptr, 0.0, W, dt));
        } else if (method == QF_ERK_HEUN) {     // erk.py:91-110
            QF_TRY(products(W));
            QF_TRY(stage(acc, 0.0, Wp, dt, nullptr, 0.0));            // F0; Wprime = W + dt*F0
            QF_TRY(products(Wp));
            QF_TRY(stage(acc, 1.0, nullptr, 0.0, W, dt / 2.0));       // F += F0; F *= dt/2; W += F
        } else {                                // erk.py:139-156
            QF_TRY(products(W));
            QF_TRY(stage(acc, 0.0, Wp, dt / 2.0, nullptr, 0.0));      // K1
            QF_TRY(products(Wp));
            QF_TRY(stage(acc, 2.0, Wp, dt / 2.0, nullptr, 0.0));      // K1 + 2 K2
            QF_TRY(products(Wp));
            QF_TRY(stage(acc, 2.0, Wp, dt, nullptr, 0.0));            // ... + 2 K3
            QF_TRY(products(Wp));
            QF_TRY(stage(acc, 1.0, nullptr, 0.0, W, dt / 6.0));       // ... + K4; W += (dt/6) * (...)
        }
    }
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

// euler / heun / rk4 on a stack of k states (erk.py:19-160 with W.shape = (k,N,N)): the Hamiltonian reads state 0
// (solve_poisson reduces a stack to its first state, cpu.py:672-674,696-697) and bracket(P, W) broadcasts the one
// stream matrix over the stack (geometry.py:41-49: P@W - W@P with numpy's batched matmul) -- every state is
// advected by state 0's flow, stage by stage.  Host in / host out.
int qf_erk_states(qf_ctx *ctx, void *states_host, int k, int method, double dt, int steps, int skewh)
{
    QF_TRY(check_ctx(ctx));
    if (method < QF_ERK_EULER || method > QF_ERK_RK4 || steps < 0 || k < 1 || !states_host) {
        qf_set_error("qf_erk_states: bad arguments (method %d, steps %d, k %d)", method, steps, k);
        return QF_ERR_INVALID;
    }
    const int N = ctx->N;
    const size_t mbytes = (size_t)N * N * sizeof(cplx);
    const double inv_hb = 1.0 / qf_hbar(N);
    while (ctx->multi.size() < (size_t)3 * k) {          // per state: X (state), Xp (stage argument), acc
        cplx *p = nullptr;
        QF_HIP(hipMalloc((void **)&p, mbytes));
        ctx->multi.push_back(p);
    }
    struct st { cplx *X, *Xp, *acc; };
    std::vector<st> S((size_t)k);
    bool one_product = skewh != 0;
    for (int j = 0; j < k; ++j) {
        S[j].X = ctx->multi[3 * j];
        S[j].Xp = ctx->multi[3 * j + 1];
        S[j].acc = ctx->multi[3 * j + 2];
        QF_HIP(hipMemcpyAsync(S[j].X, (const char *)states_host + (size_t)j * mbytes, mbytes, hipMemcpyHostToDevice, ctx->stream));
        if (one_product) {       // X@P = (P@X)^H needs every state exactly skew-Hermitian
            QF_TRY(qf_launch_skew_defect(ctx, S[j].X, ctx->scalars + 4));
            QF_HIP(hipMemcpyAsync(ctx->host_scalars, ctx->scalars + 4, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            QF_HIP(hipStreamSynchronize(ctx->stream));
            one_product = (ctx->host_scalars[0] == 0.0);
        }
    }
    cplx *P = ctx->Phalf, *A = ctx->PW, *B = ctx->stage;
    // one stage for the whole stack: P from the stage argument of state 0, then every state's slope
    auto stage_all = [&](bool from_state, double c_acc, bool want_wp, double c_wp, bool fin, double c_fin) -> int {
        QF_TRY(qf_launch_solve(ctx, ctx->poisson, from_state ? S[0].X : S[0].Xp, P, 1.0, skewh ? 1 : 0));
        // (P is complete before state 0's stage overwrites its stage argument; the other states need only P)
        for (int j = 0; j < k; ++j) {
            const cplx *Xarg = from_state ? S[j].X : S[j].Xp;
            QF_TRY(qf_launch_zgemm(ctx, P, Xarg, A, nullptr));
            if (!one_product) QF_TRY(qf_launch_zgemm(ctx, Xarg, P, B, nullptr));
            QF_TRY(qf_launch_erk_stage(ctx, A, one_product ? nullptr : B, inv_hb, S[j].X, c_acc == 0.0 && !want_wp && fin ? nullptr : S[j].acc,
                                       c_acc, want_wp ? S[j].Xp : nullptr, c_wp, fin ? S[j].X : nullptr, c_fin));
        }
        return QF_OK;
    };
    for (int s = 0; s < steps; ++s) {
        if (method == QF_ERK_EULER) {
            QF_TRY(stage_all(true, 0.0, false, 0.0, true, dt));
        } else if (method == QF_ERK_HEUN) {
            QF_TRY(stage_all(true, 0.0, true, dt, false, 0.0));
            QF_TRY(stage_all(false, 1.0, false, 0.0, true, dt / 2.0));
        } else {
            QF_TRY(stage_all(true, 0.0, true, dt / 2.0, false, 0.0));
            QF_TRY(stage_all(false, 2.0, true, dt / 2.0, false, 0.0));
            QF_TRY(stage_all(false, 2.0, true, dt, false, 0.0));
            QF_TRY(stage_all(false, 1.0, false, 0.0, true, dt / 6.0));
        }
    }
    for (int j = 0; j < k; ++j)
        QF_HIP(hipMemcpyAsync((char *)states_host + (size_t)j * mbytes, S[j].X, mbytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

// ---- isomp_simple / isomp_quasinewton (quflow/integrators/isospectral.py:155-335) -----------
// Both need X = A^-1 W and Wtilde = A^-1 (-X^H) with A = I - E, E = (stepsize/2) Ptilde
// skew-Hermitian.  The reference factors A with LAPACK (lu_factor / lu_solve).  On this machine
// the inverse is formed on the fp64 matrix cores instead: A^H A = I + E^H E, so A is always well
// conditioned (singular values in [1, sqrt(1 + |E|^2)]) and Newton-Schulz
//      Y <- Y + Y (I - A Y),     Y0 = A^H / (1 + |E|_inf^2)  (or the previous inverse, warm)
// converges quadratically from a residual <= |E|^2/(1+|E|^2) < 1: 3-5 iterations of two N^3
// products, no pivoting, no triangular solves.  Same stepper, same result to rounding
// (cond(A) ~ 1); only the linear-solve method differs from the reference.
struct ns_work {
    cplx *E, *Y, *R, *T;     // E = (stepsize/2) P;  Y ~ A^-1;  R, T scratch
    bool warm = false;
    bool general = false;    // E is not known to be skew-Hermitian (foreign Hamiltonian): conservative start
    int iterations = 0;      // Newton-Schulz iterations performed (diagnostic)
    double floor = 0.0;      // |I - A Y|_inf behind the last update: the rounding noise of this inverse
};

static int ns_invert(qf_ctx *ctx, ns_work &w)
{
    // R = I - A Y = I - Y + E Y
    auto residual = [&](double *r_out) -> int {
        QF_TRY(qf_launch_zgemm(ctx, w.E, w.Y, w.T, nullptr));
        QF_TRY(qf_launch_lincomb(ctx, -1.0, w.Y, 1.0, w.T, 1.0, w.R));
        QF_TRY(qf_launch_norm_inf(ctx, w.R, ctx->scalars + 6));
        return read_scalar(ctx, ctx->scalars + 6, r_out);
    };
    double r = 2.0;
    if (w.warm) {
        QF_TRY(residual(&r));
        // E moved by no more than the rounding noise of the inverse: Y stays as it is.  An update from a residual at the noise
        // level only reshuffles Y's last bits, and with them the two solves' -- the quasi-Newton iteration's exit test asks for a
        // bit-level fixed point (|Wt - Wt_new|_inf < eps * stepsize * |W|, isospectral.py:190-191,227), which a Y that keeps
        // moving reaches one to three passes late or not before maxit (the reference's LU is a fixed function of A)
        if (w.floor > 0.0 && r <= 2.0 * w.floor) return QF_OK;
    }
    if (!(r < 0.5)) {
        // cold start: Y0 = A^H / (1 + |E|_inf^2) = (I + E) / (1 + c)
        double en = 0.0;
        QF_TRY(qf_launch_norm_inf(ctx, w.E, ctx->scalars + 6));
        QF_TRY(read_scalar(ctx, ctx->scalars + 6, &en));
        if (!(en == en) || en > 1e150) {       // (scipy.linalg.lu_factor checks its argument: isospectral.py:211, 290)
            qf_set_error("array must not contain infs or NaNs");
            return QF_ERR_NONFINITE;
        }
        // skew-Hermitian E: A A^H = I + E E^H, spectrum in [1, 1 + |E|^2].  A Hamiltonian that is not
        // skew-Hermitian (foreign hook): A A^H has its spectrum in [(1 - |E|)^2, (1 + |E|)^2]
        const double s = w.general ? 1.0 / ((1.0 + en) * (1.0 + en)) : 1.0 / (1.0 + en * en);
        if (w.general) {
            // Y0 = s A^H = s (I - E^H): -E^H through the transpose kernel
            QF_TRY(qf_launch_neg_conj_transpose(ctx, w.E, w.T));
            QF_TRY(qf_launch_lincomb(ctx, s, w.T, 0.0, nullptr, s, w.Y));
        } else {
            QF_TRY(qf_launch_lincomb(ctx, s, w.E, 0.0, nullptr, s, w.Y));
        }
        QF_TRY(residual(&r));
    }
    for (int it = 0; it < 200; ++it) {
        // Y <- Y + Y R
        QF_TRY(qf_launch_zgemm(ctx, w.Y, w.R, w.T, nullptr));
        QF_TRY(qf_launch_lincomb(ctx, 1.0, w.Y, 1.0, w.T, 0.0, w.Y));
        w.iterations += 1;
        if (r < 1e-8) {          // the update just applied leaves a residual ~ r^2 < eps: what is measured now is the noise floor
            w.warm = true;
            QF_TRY(residual(&w.floor));
            return QF_OK;
        }
        const double r_prev = r;
        QF_TRY(residual(&r));
        if (!(r == r) || (it > 8 && r > r_prev)) break;
    }
    qf_set_error("isomp linear solve: Newton-Schulz did not converge (residual %.3e)", r);
    return QF_ERR_STATE;
}

// one pass of the two solves: X = A^-1 Wrhs;  Wt_out = A^-1 (-X^H)    (isospectral.py:214-218, 293-297)
static int ns_two_solves(qf_ctx *ctx, ns_work &w, const cplx *Wrhs, cplx *X, cplx *Wt_out)
{
    QF_TRY(qf_launch_zgemm(ctx, w.Y, Wrhs, X, nullptr));
    QF_TRY(qf_launch_neg_conj_transpose(ctx, X, w.T));
    QF_TRY(qf_launch_zgemm(ctx, w.Y, w.T, Wt_out, nullptr));
    return QF_OK;
}

// W <- A^H Wt A = (I + E) Wt (I - E)    (isospectral.py:232, 300; A^H = I + E for the skew-Hermitian E of the
// built-in Hamiltonian).  EH != nullptr: a scratch matrix -- E need not be skew-Hermitian (foreign Hamiltonian, the
// general Poisson branch): A^H = I - E^H is formed explicitly.
static int ns_update_W(qf_ctx *ctx, ns_work &w, const cplx *Wt, cplx *Wout, cplx *EH = nullptr)
{
    if (EH) {
        QF_TRY(qf_launch_neg_conj_transpose(ctx, w.E, EH));                 // -E^H
        QF_TRY(qf_launch_zgemm(ctx, EH, Wt, w.T, nullptr));                 // -E^H Wt
        QF_TRY(qf_launch_lincomb(ctx, 1.0, Wt, 1.0, w.T, 0.0, w.R));        // V = (I - E^H) Wt
        QF_TRY(qf_launch_zgemm(ctx, w.R, w.E, w.T, nullptr));               // V E
        QF_TRY(qf_launch_lincomb(ctx, 1.0, w.R, -1.0, w.T, 0.0, Wout));     // V - V E
        return QF_OK;
    }
    QF_TRY(qf_launch_zgemm(ctx, w.E, Wt, w.T, nullptr));                    // E Wt
    QF_TRY(qf_launch_lincomb(ctx, 1.0, Wt, 1.0, w.T, 0.0, w.R));            // V = Wt + E Wt
    QF_TRY(qf_launch_zgemm(ctx, w.R, w.E, w.T, nullptr));                   // V E
    QF_TRY(qf_launch_lincomb(ctx, 1.0, w.R, -1.0, w.T, 0.0, Wout));         // V - V E
    return QF_OK;
}

static int ns_setup(qf_ctx *ctx, ns_work &w)
{
    const size_t mbytes = (size_t)ctx->N * ctx->N * sizeof(cplx);
    if (!ctx->ns_inv) QF_HIP(hipMalloc((void **)&ctx->ns_inv, mbytes));
    if (!ctx->ns_tmp) QF_HIP(hipMalloc((void **)&ctx->ns_tmp, mbytes));
    w.E = ctx->Phalf;
    w.Y = ctx->ns_inv;
    w.R = ctx->ns_tmp;
    w.T = ctx->PW;
    return QF_OK;
}

// E = (stepsize/2) Ptilde with Ptilde = hamiltonian(Wtilde): the built-in Delta^-1 on the device, or the
// caller's hook on pinned host copies (isospectral.py:207, 286: `Ptilde = hamiltonian(Wtilde)`)
static int lu_hamiltonian(qf_ctx *ctx, ns_work &w, const cplx *Wt, double half_stepsize, const qf_isomp_hooks *hooks)
{
    // (the Laplacian backend's select_skewherm flag picks the solve's branch, cpu.py:563-591)
    if (!hooks || !hooks->hamiltonian) return qf_launch_solve(ctx, ctx->poisson, Wt, w.E, half_stepsize, (!hooks || hooks->solve_skewh) ? 1 : 0);
    const size_t mbytes = (size_t)ctx->N * ctx->N * sizeof(cplx);
    if (ctx->hook_host_bytes < mbytes) {
        for (int q = 0; q < 3; ++q) {
            if (ctx->hook_host[q]) (void)hipHostFree(ctx->hook_host[q]);
            ctx->hook_host[q] = nullptr;
        }
        ctx->hook_host_bytes = 0;
        for (int q = 0; q < 3; ++q) QF_HIP(hipHostMalloc((void **)&ctx->hook_host[q], mbytes, hipHostMallocDefault));
        ctx->hook_host_bytes = mbytes;
    }
    cplx *hW = ctx->hook_host[0], *hP = ctx->hook_host[1];
    QF_HIP(hipMemcpyAsync(hW, Wt, mbytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    const int rc = hooks->hamiltonian(hooks->user, hW, hP, 0.0);
    if (rc != 0) {
        qf_set_error("hamiltonian hook returned %d", rc);
        return QF_ERR_CALLBACK;
    }
    QF_HIP(hipMemcpyAsync(w.T, hP, mbytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_lincomb(ctx, half_stepsize, w.T, 0.0, nullptr, 0.0, w.E));
    return QF_OK;
}

static int isomp_simple_impl(qf_ctx *ctx, double dt, int steps, const qf_isomp_hooks *hooks)
{
    QF_TRY(check_ctx(ctx));
    if (steps < 0) {
        qf_set_error("qf_isomp_simple: steps must be >= 0");
        return QF_ERR_INVALID;
    }
    const size_t mbytes = (size_t)ctx->N * ctx->N * sizeof(cplx);
    const double stepsize = dt / qf_hbar(ctx->N);           // isospectral.py:281
    ctx->w_skew_known = false;
    ns_work w;
    QF_TRY(ns_setup(ctx, w));
    const bool general_branch = hooks && !hooks->skewh;      // select_skewherm(False): isospectral.py:303-314
    w.general = (hooks && (hooks->hamiltonian || !hooks->solve_skewh)) || general_branch;
    cplx *Wt = ctx->Whalf, *X = ctx->stage;
    QF_HIP(hipMemcpyAsync(Wt, ctx->W, mbytes, hipMemcpyDeviceToDevice, ctx->stream));   // Wtilde = W.copy()
    ns_work w2;                                               // general branch: the inverse of Aalt = I + E
    if (general_branch) {
        while (ctx->multi.size() < 2) {
            cplx *p = nullptr;
            QF_HIP(hipMalloc((void **)&p, mbytes));
            ctx->multi.push_back(p);
        }
        w2.E = ctx->multi[0];        // -E
        w2.Y = ctx->multi[1];
        w2.R = w.R;
        w2.T = w.T;
        w2.general = true;
    }
    for (int k = 0; k < steps; ++k) {
        QF_TRY(lu_hamiltonian(ctx, w, Wt, stepsize / 2.0, hooks));                     // E = (stepsize/2) Ptilde
        QF_TRY(ns_invert(ctx, w));
        if (general_branch) {
            // X = A^-1 W;  Wtilde = (Aalt^-H X^H)^H = X Aalt^-1, Aalt = I + E;  W = Aalt Wtilde A   (:305-314)
            QF_TRY(qf_launch_lincomb(ctx, -1.0, w.E, 0.0, nullptr, 0.0, w2.E));
            QF_TRY(ns_invert(ctx, w2));
            QF_TRY(qf_launch_zgemm(ctx, w.Y, ctx->W, X, nullptr));
            QF_TRY(qf_launch_zgemm(ctx, X, w2.Y, Wt, nullptr));
            QF_TRY(ns_update_W(ctx, w, Wt, ctx->W));
            continue;
        }
        QF_TRY(ns_two_solves(ctx, w, ctx->W, X, Wt));
        QF_TRY(ns_update_W(ctx, w, Wt, ctx->W, w.general ? X : nullptr));
    }
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_isomp_simple(qf_ctx *ctx, double dt, int steps) { return isomp_simple_impl(ctx, dt, steps, nullptr); }

// ... with a foreign `hamiltonian(Wtilde)` (the `hamiltonian` and `user` members of the hook table): the state, the
// Newton-Schulz inverse and the products stay on the device, Wtilde goes down and Ptilde comes up once per pass
int qf_isomp_simple_hooked(qf_ctx *ctx, double dt, int steps, const qf_isomp_hooks *hooks)
{
    return isomp_simple_impl(ctx, dt, steps, hooks);
}

static int isomp_quasinewton_impl(qf_ctx *ctx, double dt, int steps, double tol, int maxit, qf_isomp_stats *stats_out,
                                  const qf_isomp_hooks *hooks);

int qf_isomp_quasinewton(qf_ctx *ctx, double dt, int steps, double tol, int maxit, qf_isomp_stats *stats_out)
{
    return isomp_quasinewton_impl(ctx, dt, steps, tol, maxit, stats_out, nullptr);
}

int qf_isomp_quasinewton_hooked(qf_ctx *ctx, double dt, int steps, double tol, int maxit, qf_isomp_stats *stats_out,
                                const qf_isomp_hooks *hooks)
{
    return isomp_quasinewton_impl(ctx, dt, steps, tol, maxit, stats_out, hooks);
}

static int isomp_quasinewton_impl(qf_ctx *ctx, double dt, int steps, double tol, int maxit, qf_isomp_stats *stats_out,
                                  const qf_isomp_hooks *hooks)
{
    QF_TRY(check_ctx(ctx));
    if (steps < 0 || maxit < 1) {
        qf_set_error("qf_isomp_quasinewton: steps must be >= 0 and maxit >= 1");
        return QF_ERR_INVALID;
    }
    const size_t mbytes = (size_t)ctx->N * ctx->N * sizeof(cplx);
    const double stepsize = dt / qf_hbar(ctx->N);           // isospectral.py:187
    ctx->w_skew_known = false;
    if (tol < 0) {                                          // isospectral.py:190-191
        double nrm = 0.0;
        QF_TRY(qf_norm_inf_W(ctx, &nrm));
        tol = std::numeric_limits<double>::epsilon() * stepsize * nrm;
    }
    ns_work w;
    QF_TRY(ns_setup(ctx, w));
    w.general = hooks && (hooks->hamiltonian || !hooks->solve_skewh);
    cplx *Wt = ctx->Whalf, *Wt_new = ctx->dW[0], *X = ctx->stage, *D = ctx->dW[1];
    QF_HIP(hipMemcpyAsync(Wt, ctx->W, mbytes, hipMemcpyDeviceToDevice, ctx->stream));   // Wtilde = W.copy()
    long long total_iterations = 0, number_of_maxit = 0;
    double resnorm = 0.0;
    for (int k = 0; k < steps; ++k) {
        bool converged = false;
        for (int i = 0; i < maxit; ++i) {
            total_iterations += 1;
            QF_TRY(lu_hamiltonian(ctx, w, Wt, stepsize / 2.0, hooks));                 // A = Id - (stepsize/2) Ptilde
            QF_TRY(ns_invert(ctx, w));
            QF_TRY(ns_two_solves(ctx, w, ctx->W, X, Wt_new));
            // resnorm = |Wtilde - Wtilde_new|_inf    (isospectral.py:221)
            QF_TRY(qf_launch_lincomb(ctx, 1.0, Wt, -1.0, Wt_new, 0.0, D));
            QF_TRY(qf_launch_norm_inf(ctx, D, ctx->scalars + 7));
            QF_TRY(read_scalar(ctx, ctx->scalars + 7, &resnorm));
            if (!QF_FINITE(resnorm)) {       // scipy.linalg.norm raises here (isospectral.py:221)
                qf_set_error("array must not contain infs or NaNs");
                return QF_ERR_NONFINITE;
            }
            cplx *t = Wt; Wt = Wt_new; Wt_new = t;                                     // Wtilde = Wtilde_new
            if (resnorm < tol) {                                                      // isospectral.py:227
                converged = true;
                break;
            }
        }
        if (!converged) number_of_maxit += 1;
        QF_TRY(ns_update_W(ctx, w, Wt, ctx->W, w.general ? X : nullptr));
    }
    // leave Wtilde where later calls expect scratch only; nothing to restore
    QF_HIP(hipStreamSynchronize(ctx->stream));
    if (stats_out) {
        stats_out->total_iterations = total_iterations;
        stats_out->number_of_maxit = number_of_maxit;
        stats_out->tol_used = tol;
        stats_out->last_resnorm = resnorm;
    }
    return QF_OK;
}

// ---- isomp on a stack of states / magmp -------------------------------------------------------
// isomp_fixedpoint with W.shape = (k,N,N) (quflow/integrators/isospectral.py:463-611, 3-D
// branches): P comes from state 0 only (cpu.py:696-697), every state runs the same products, the
// exit test uses state 0's residual (isospectral.py:527-532).  magnetic != 0 (k == 2): magmp,
// quflow/integrators/mhd.py:235-456 with hamiltonian = solve_mhd (mhd.py:10-18): B = Delta Theta
// and the vorticity state gets [B, Theta] on top.  Host in / host out; the iteration control is
// host-side (one scalar read-back per iteration, as the reference does): these are the secondary
// steppers, their products (>= 4 per iteration) dwarf the read-back.
int qf_isomp_states(qf_ctx *ctx, void *states_host, int k, double dt, int steps, double tol, int minit, int maxit,
                    int reinitialize, int magnetic, qf_isomp_stats *stats_out)
{
    QF_TRY(check_ctx(ctx));
    if (minit < 1) {
        qf_set_error("minit must be at least 1.");
        return QF_ERR_INVALID;
    }
    if (maxit < minit) {
        qf_set_error("maxit must be at minit.");
        return QF_ERR_INVALID;
    }
    if (!states_host || k < 1 || steps < 0 || (magnetic && k != 2)) {
        qf_set_error("qf_isomp_states: bad arguments (k=%d, steps=%d, magnetic=%d)", k, steps, magnetic);
        return QF_ERR_INVALID;
    }
    const int N = ctx->N;
    const size_t NN = (size_t)N * N, mbytes = NN * sizeof(cplx);
    const double hb = qf_hbar(N);
    const double vareps = dt / (2 * hb);
    // per state: X, dX[2], Xhalf, PXc;  magnetic: Bhalf, BT, BTP
    const size_t need = (size_t)5 * k + (magnetic ? 3 : 0);
    while (ctx->multi.size() < need) {
        cplx *p = nullptr;
        QF_HIP(hipMalloc((void **)&p, mbytes));
        ctx->multi.push_back(p);
    }
    const int slots32 = (N + 31) / 32;
    if (!ctx->multi_rowpart) QF_HIP(hipMalloc((void **)&ctx->multi_rowpart, (size_t)slots32 * N * sizeof(double)));
    struct st { cplx *X, *dX[2], *Xhalf, *PXc; int cur; };
    std::vector<st> S((size_t)k);
    for (int j = 0; j < k; ++j) {
        S[j].X = ctx->multi[5 * j];
        S[j].dX[0] = ctx->multi[5 * j + 1];
        S[j].dX[1] = ctx->multi[5 * j + 2];
        S[j].Xhalf = ctx->multi[5 * j + 3];
        S[j].PXc = ctx->multi[5 * j + 4];
        S[j].cur = 0;
        QF_HIP(hipMemcpyAsync(S[j].X, (const char *)states_host + (size_t)j * mbytes, mbytes, hipMemcpyHostToDevice, ctx->stream));
        QF_HIP(hipMemsetAsync(S[j].dX[0], 0, mbytes, ctx->stream));                          // dW = zeros_like(W)
        QF_HIP(hipMemcpyAsync(S[j].Xhalf, S[j].X, mbytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    cplx *Bhalf = magnetic ? ctx->multi[5 * k] : nullptr;
    cplx *BT = magnetic ? ctx->multi[5 * k + 1] : nullptr;
    cplx *BTP = magnetic ? ctx->multi[5 * k + 2] : nullptr;

    // the upper-triangle second product wants every state exactly skew-Hermitian
    bool tri = ctx->gemm_tri_allowed && ctx->sk_partial && N >= ctx->gemm_tri_min_n;
    for (int j = 0; j < k && tri; ++j) {
        QF_TRY(qf_launch_skew_defect(ctx, S[j].X, ctx->scalars + 4));
        QF_HIP(hipMemcpyAsync(ctx->host_scalars, ctx->scalars + 4, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        QF_HIP(hipStreamSynchronize(ctx->stream));
        tri = (ctx->host_scalars[0] == 0.0);
    }
    const bool tri_saved = ctx->gemm_tri, tri32_saved = ctx->gemm_tri32;
    ctx->gemm_tri = tri;
    ctx->gemm_tri32 = false;       // (stacks below N = 768 keep the full product)
    auto restore = [&](int rc) { ctx->gemm_tri = tri_saved; ctx->gemm_tri32 = tri32_saved; return rc; };
#define QF_TRY_R(call)                  \
    do {                                \
        int _r = (call);                \
        if (_r != QF_OK) return restore(_r); \
    } while (0)

    // tolerance from state 0 (isospectral.py:440-452; magmp uses sqrt(eps) always, mhd.py:331-341)
    if (tol < 0) {
        double nrm = 0.0;
        QF_TRY_R(qf_launch_norm_inf(ctx, S[0].X, ctx->scalars));
        QF_TRY_R(read_scalar(ctx, ctx->scalars, &nrm));
        tol = (std::sqrt(std::numeric_limits<double>::epsilon()) * dt / hb) * nrm;
    }

    long long total_iterations = 0, number_of_maxit = 0;
    double resnorm = 0.0;
    for (int step = 0; step < steps; ++step) {
        resnorm = std::numeric_limits<double>::infinity();
        bool broke = false;
        for (int i = 0; i < maxit; ++i) {
            total_iterations += 1;
            // Phalf = vareps * Delta^-1 Whalf[0]   (+ Bhalf = vareps * Delta Thetahalf)
            QF_TRY_R(qf_launch_solve(ctx, ctx->poisson, S[0].Xhalf, ctx->Phalf, vareps, 1));
            if (magnetic) {
                QF_TRY_R(qf_launch_laplace(ctx, S[1].Xhalf, Bhalf));
                QF_TRY_R(qf_launch_lincomb(ctx, vareps, Bhalf, 0.0, nullptr, 0.0, Bhalf));
            }
            for (int j = 0; j < k; ++j)                                   // Pstatecomm = Phalf @ statehalf
                QF_TRY_R(qf_launch_zgemm(ctx, ctx->Phalf, S[j].Xhalf, S[j].PXc, nullptr));
            if (magnetic) {
                QF_TRY_R(qf_launch_zgemm(ctx, Bhalf, S[1].Xhalf, BT, nullptr));       // BThetacomm
                QF_TRY_R(qf_launch_zgemm(ctx, BT, ctx->Phalf, BTP, nullptr));         // BThetaPhalf
            }
            for (int j = 0; j < k; ++j) {
                // dX = PXc @ Phalf + (PXc - PXc^H);  Xhalf = X + dX;  row sums of |dX_old - dX|
                qf_epilogue ep;
                ep.PW = S[j].PXc;
                ep.W = S[j].X;
                ep.dW[0] = S[j].dX[S[j].cur];
                ep.dW[1] = S[j].dX[S[j].cur ^ 1];
                ep.Whalf = S[j].Xhalf;
                ep.rowpart = ctx->rowpart;
                QF_TRY_R(qf_launch_zgemm(ctx, S[j].PXc, ctx->Phalf, nullptr, &ep));
                if (j == 0) {
                    if (magnetic) {
                        QF_TRY_R(qf_launch_magnetic_fix(ctx, BTP, BT, S[0].dX[S[0].cur ^ 1], S[0].dX[S[0].cur], S[0].X,
                                                        S[0].Xhalf, ctx->multi_rowpart));
                        QF_TRY_R(qf_launch_norm_from_rowpart(ctx, ctx->multi_rowpart, slots32, ctx->scalars + 1));
                    } else {
                        QF_TRY_R(qf_launch_norm_from_rowpart(ctx, ctx->rowpart, rowpart_slots(ctx), ctx->scalars + 1));
                    }
                }
                S[j].cur ^= 1;
            }
            if (i + 1 >= minit) {
                const double resnorm_old = resnorm;
                QF_TRY_R(read_scalar(ctx, ctx->scalars + 1, &resnorm));
                if (!QF_FINITE(resnorm)) {       // scipy.linalg.norm raises here (isospectral.py:534, mhd.py: same test)
                    qf_set_error("array must not contain infs or NaNs");
                    return restore(QF_ERR_NONFINITE);
                }
                if (resnorm <= tol || resnorm >= resnorm_old) {
                    broke = true;
                    break;
                }
            }
        }
        if (!broke) number_of_maxit += 1;
        // W += 2 (PXc - PXc^H) for every state; Xhalf = X + dX for the next step
        for (int j = 0; j < k; ++j)
            QF_TRY_R(qf_launch_update(ctx, S[j].PXc, S[j].X, S[j].dX[S[j].cur], S[j].dX[S[j].cur], S[j].Xhalf, nullptr,
                                      reinitialize));
        if (magnetic)
            QF_TRY_R(qf_launch_magnetic_update(ctx, BT, S[0].X, reinitialize ? nullptr : S[0].dX[S[0].cur], S[0].Xhalf));
    }
    for (int j = 0; j < k; ++j)
        QF_HIP(hipMemcpyAsync((char *)states_host + (size_t)j * mbytes, S[j].X, mbytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
#undef QF_TRY_R
    ctx->gemm_tri = tri_saved;
    ctx->gemm_tri32 = tri32_saved;
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
    QF_TRY(enqueue_diagnostics(ctx));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    const double wp = ctx->host_scalars[0], ww = ctx->host_scalars[1];
    if (energy_euler) *energy_euler = -(wp / N) / 2.0;
    if (enstrophy) *enstrophy = (ww / N) / 2.0;
    return QF_OK;
}

int qf_isomp_diag(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum, int reinitialize,
                  qf_isomp_stats *stats_out, double *energy_euler, double *enstrophy)
{
    QF_TRY(check_ctx(ctx));
    ctx->diag_at_exit = true;
    ctx->diag_valid = false;
    const int rc = isomp_impl(ctx, dt, steps, tol, minit, maxit, compsum, reinitialize, stats_out, false);
    ctx->diag_at_exit = false;
    if (rc != QF_OK) return rc;
    if (!ctx->diag_valid) return qf_diagnostics(ctx, energy_euler, enstrophy);     // (a path without the fused exit)
    ctx->diag_valid = false;
    const double wp = ctx->host_scalars[0], ww = ctx->host_scalars[1];
    if (energy_euler) *energy_euler = -(wp / ctx->N) / 2.0;
    if (enstrophy) *enstrophy = (ww / ctx->N) / 2.0;
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

int qf_fixedpoint_products(qf_ctx *ctx, const void *Phalf_host, const void *Whalf_host, const void *W_host,
                           const void *dW_old_host, int variant, void *dW_new_host, void *Whalf_new_host,
                           double *rowsum_host)
{
    QF_TRY(check_ctx(ctx));
    if (!Phalf_host || !Whalf_host || !W_host || !dW_old_host || !dW_new_host || !Whalf_new_host || !rowsum_host) {
        qf_set_error("qf_fixedpoint_products: null buffer");
        return QF_ERR_INVALID;
    }
    // variant: low 4 bits = the second product's kernel; the parity tests also choose its partition here (the stepper's own
    // partition follows rules, not switches): kind 1: bits 8-15 = least K-tiles per workgroup (0: the rule), bits 16-23 =
    // 64 + the epilogue weight E (0: the rule); kind 2: bits 8-11 / 12-15 = K pieces per off-diagonal / diagonal tile
    const int variant_arg = variant;
    variant &= 15;
    if (variant == 1 && !ctx->sk_partial) {
        qf_set_error("qf_fixedpoint_products: the upper-triangle product needs N %% 64 == 0 (N=%d)", ctx->N);
        return QF_ERR_INVALID;
    }
    if (variant == 2) {
        if (ctx->N < 64) {
            qf_set_error("qf_fixedpoint_products: the 32x32 upper-triangle product needs N >= 64 (N=%d)", ctx->N);
            return QF_ERR_INVALID;
        }
        QF_TRY(tri32_alloc(ctx));
    }
    const int N = ctx->N;
    const size_t bytes = (size_t)N * N * sizeof(cplx);
    QF_HIP(hipMemcpyAsync(ctx->Phalf, Phalf_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(ctx->Whalf, Whalf_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(ctx->stage, W_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(ctx->dW[0], dW_old_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_zgemm(ctx, ctx->Phalf, ctx->Whalf, ctx->PW, nullptr));
    qf_epilogue ep;
    ep.PW = ctx->PW;
    ep.W = ctx->stage;
    ep.dW[0] = ctx->dW[0];
    ep.dW[1] = ctx->dW[1];
    ep.Whalf = ctx->Whalf;
    ep.rowpart = ctx->rowpart;
    const bool saved = ctx->gemm_tri, saved32 = ctx->gemm_tri32;
    const int s_min = ctx->sk_min_units, s_epi = ctx->sk_epi_units, s_so = ctx->tri32_split, s_sd = ctx->tri32_split_diag;
    ctx->gemm_tri = (variant == 1);
    ctx->gemm_tri32 = (variant == 2);
    if (variant == 1) {
        if ((variant_arg >> 8) & 0xff) ctx->sk_min_units = (variant_arg >> 8) & 0xff;
        if ((variant_arg >> 16) & 0xff) ctx->sk_epi_units = ((variant_arg >> 16) & 0xff) - 64;
    } else if (variant == 2) {
        const int so = (variant_arg >> 8) & 15, sd = (variant_arg >> 12) & 15;
        if (so == 1 || so == 2 || so == 4) ctx->tri32_split = so;
        if (sd == 1 || sd == 2 || sd == 4) ctx->tri32_split_diag = sd;
    }
    int rc = qf_launch_zgemm(ctx, ctx->PW, ctx->Phalf, nullptr, &ep);   // unguarded: parity 0, writes dW[1]
    ctx->gemm_tri = saved;
    ctx->gemm_tri32 = saved32;
    ctx->sk_min_units = s_min;
    ctx->sk_epi_units = s_epi;
    ctx->tri32_split = s_so;
    ctx->tri32_split_diag = s_sd;
    QF_TRY(rc);
    // row sums of |dW_old - dW_new| in the fixed slot order k_norm_decide uses
    QF_TRY(qf_launch_sum_rowpart(ctx, ctx->rowpart, variant == 1 ? ctx->N / 64 : variant == 2 ? (ctx->N + 31) / 32 : ctx->rowpart_tiles, ctx->rowsum));
    QF_HIP(hipMemcpyAsync(dW_new_host, ctx->dW[1], bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipMemcpyAsync(Whalf_new_host, ctx->Whalf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipMemcpyAsync(rowsum_host, ctx->rowsum, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

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

static int oz_alloc(qf_ctx *ctx)
{
    for (int q = 0; q < 4; ++q) {
        if (!ctx->oz_planes[q]) QF_HIP(hipMalloc((void **)&ctx->oz_planes[q], qf_oz_operand_bytes(ctx->N, ctx->oz_digits)));
        if (!ctx->oz_scale[q])      // row record: N scales, then the int32 digit sums (ozaki.hip)
            QF_HIP(hipMalloc((void **)&ctx->oz_scale[q], qf_oz_record_bytes(ctx->N, ctx->oz_digits)));
    }
    if (!ctx->oz_diag) QF_HIP(hipMalloc((void **)&ctx->oz_diag, (size_t)ctx->N * sizeof(double)));
    if (!ctx->oz_tbuf) {     // result tiles + epoch flags of the upper-triangle second product
        const size_t t = (size_t)(ctx->N / 64), nup = t * (t + 1) / 2;
        QF_HIP(hipMalloc((void **)&ctx->oz_tbuf, nup * 64 * 64 * sizeof(cplx)));
        QF_HIP(hipMalloc((void **)&ctx->oz_tflags, nup * sizeof(unsigned)));
        QF_HIP(hipMemsetAsync(ctx->oz_tflags, 0, nup * sizeof(unsigned), ctx->stream));
    }
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
    QF_TRY(oz_alloc(ctx));
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

// ---- complex64 data: float32 arithmetic, as the reference computes it (cpu.py:725, isospectral.py:440-448) ----

static int need_c64(qf_ctx *ctx)
{
    QF_TRY(check_ctx(ctx));
    return qf_c64_alloc(ctx);
}

int qf_c64_laplacian_table(qf_ctx *ctx, int bc, float *lap_host)
{
    QF_TRY(need_c64(ctx));
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
    QF_TRY(need_c64(ctx));
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
    QF_TRY(need_c64(ctx));
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
    QF_TRY(need_c64(ctx));
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

int qf_c64_upload_W(qf_ctx *ctx, const void *W_host)
{
    QF_TRY(need_c64(ctx));
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
    QF_TRY(need_c64(ctx));
    if (!W_host) {
        qf_set_error("qf_c64_download_W: null buffer");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipMemcpyAsync(W_host, ctx->c64->W, (size_t)ctx->N * ctx->N * sizeof(float2), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

int qf_c64_isomp(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum, int reinitialize,
                 qf_isomp_stats *stats_out)
{
    QF_TRY(need_c64(ctx));
    return isomp_impl(ctx, dt, steps, tol, minit, maxit, compsum, reinitialize, stats_out, false, true);
}

int qf_c64_isomp_continue(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum, int reinitialize,
                          qf_isomp_stats *stats_out)
{
    QF_TRY(need_c64(ctx));
    return isomp_impl(ctx, dt, steps, tol, minit, maxit, compsum, reinitialize, stats_out, true, true);
}

int qf_c64_diagnostics(qf_ctx *ctx, double *energy_euler, double *enstrophy)
{
    QF_TRY(need_c64(ctx));
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

int qf_cgemm(qf_ctx *ctx, const void *A_host, const void *B_host, void *C_host)
{
    QF_TRY(need_c64(ctx));
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

int qf_c64_fixedpoint_products(qf_ctx *ctx, const void *Phalf_host, const void *Whalf_host, const void *W_host,
                               const void *dW_old_host, void *dW_new_host, void *Whalf_new_host, double *rowsum_host)
{
    QF_TRY(need_c64(ctx));
    if (!Phalf_host || !Whalf_host || !W_host || !dW_old_host || !dW_new_host || !Whalf_new_host || !rowsum_host) {
        qf_set_error("qf_c64_fixedpoint_products: null buffer");
        return QF_ERR_INVALID;
    }
    qf_c64 *f = ctx->c64;
    const int N = ctx->N;
    const size_t bytes = (size_t)N * N * sizeof(float2);
    QF_HIP(hipMemcpyAsync(f->Phalf, Phalf_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(f->Whalf, Whalf_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(f->stage, W_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(f->dW[0], dW_old_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_cgemm(ctx, f->Phalf, f->Whalf, f->PW, nullptr));
    qf_epilogue_f ep;
    ep.PW = f->PW;
    ep.W = f->stage;
    ep.dW[0] = f->dW[0];
    ep.dW[1] = f->dW[1];
    ep.Whalf = f->Whalf;
    ep.rowpart = f->rowpart;
    QF_TRY(qf_launch_cgemm(ctx, f->PW, f->Phalf, nullptr, &ep));    // unguarded: parity 0, writes dW[1]
    QF_TRY(qf_launch_sum_rowpart(ctx, f->rowpart, f->rowpart_tiles, ctx->rowsum));
    QF_HIP(hipMemcpyAsync(dW_new_host, f->dW[1], bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipMemcpyAsync(Whalf_new_host, f->Whalf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipMemcpyAsync(rowsum_host, ctx->rowsum, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

// the same through the upper-triangle second product (k_cgemm_tri; N % 64 == 0, skew-Hermitian operands)
int qf_c64_fixedpoint_products_tri(qf_ctx *ctx, const void *Phalf_host, const void *Whalf_host, const void *W_host,
                                   const void *dW_old_host, void *dW_new_host, void *Whalf_new_host, double *rowsum_host)
{
    QF_TRY(need_c64(ctx));
    if (!Phalf_host || !Whalf_host || !W_host || !dW_old_host || !dW_new_host || !Whalf_new_host || !rowsum_host) {
        qf_set_error("qf_c64_fixedpoint_products_tri: null buffer");
        return QF_ERR_INVALID;
    }
    QF_TRY(qf_c64_tri_alloc(ctx));
    qf_c64 *f = ctx->c64;
    const int N = ctx->N;
    const size_t bytes = (size_t)N * N * sizeof(float2);
    QF_HIP(hipMemcpyAsync(f->Phalf, Phalf_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(f->Whalf, Whalf_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(f->stage, W_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_HIP(hipMemcpyAsync(f->dW[0], dW_old_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    QF_TRY(qf_launch_cgemm(ctx, f->Phalf, f->Whalf, f->PW, nullptr));
    qf_epilogue_f ep;
    ep.PW = f->PW;
    ep.W = f->stage;
    ep.dW[0] = f->dW[0];
    ep.dW[1] = f->dW[1];
    ep.Whalf = f->Whalf;
    ep.rowpart = f->rowpart;
    QF_TRY(qf_launch_cgemm_tri(ctx, f->PW, f->Phalf, &ep));    // unguarded: parity 0, writes dW[1] on and above the diagonal tiles
    QF_TRY(qf_launch_mirror_lower_f32(ctx, f->dW[1]));
    QF_TRY(qf_launch_sum_rowpart(ctx, f->rowpart, (N + qf_c64_tile(ctx) - 1) / qf_c64_tile(ctx), ctx->rowsum));
    QF_HIP(hipMemcpyAsync(dW_new_host, f->dW[1], bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipMemcpyAsync(Whalf_new_host, f->Whalf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipMemcpyAsync(rowsum_host, ctx->rowsum, (size_t)N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

}  // extern "C"
