// Shared by the five api_*.hip translation units (the C ABI of include/quflow_hip.h): the per-launch event scope,
// small host helpers, and the few functions one unit defines and another calls.  Host code only.
#pragma once

#include "qf_internal.h"

#include <sched.h>

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

// ---- defined in one api_*.hip, used by another (plain C++ linkage, not part of the exported C ABI) ----
extern "C" {
int qf_need_c64(qf_ctx *ctx);                 // api_laplacian.hip: the complex64 working set exists (qf_c64_upload_W or a c64 entry point made it)
int qf_oz_alloc(qf_ctx *ctx);                 // api_isomp.hip: digit planes and scales of the int8 products
int qf_rowpart_slots(const qf_ctx *ctx);      // api_isomp.hip: column-tile slots of the residual row sums for the selected second product
int qf_enqueue_diagnostics(qf_ctx *ctx);      // api_isomp.hip: P = solve(W); <W, P>, <W, W> on their way to the pinned scalars
}
