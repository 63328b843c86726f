// C ABI of libquflow_hip.so, part 3 of 5: the host-side control flow of the ISOSPECTRAL MIDPOINT stepper
// (quflow/integrators/isospectral.py:338-613): kernel selection, tagged launches, the fused / deferred / two-kernel
// step-end protocols, qf_isomp, qf_isomp_multi, their complex64 forms and the parity entries of the products.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>

#include "qf_api.h"

extern "C" {

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
int qf_oz_alloc(qf_ctx *ctx);

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
int qf_rowpart_slots(const qf_ctx *ctx) { return ctx->gemm_tri ? ctx->N / 64 : ctx->gemm_tri32 ? (ctx->N + 31) / 32 : ctx->rowpart_tiles; }

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
            QF_TRY(qf_launch_norm_decide(ctx, ctx->rowpart, qf_rowpart_slots(ctx), g));
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
    if (ctx->gemm_i8) QF_TRY(qf_oz_alloc(ctx));
    return QF_OK;
}

// after the last step has been seen complete: adopt the buffers the device ended in, restore the
// triangles the upper-triangle product skipped, synchronise, report
// P = solve_poisson(W); <W, P> and <W, W> in one pass; the two sums on their way to the pinned scalars
int qf_enqueue_diagnostics(qf_ctx *ctx)
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
        QF_TRY(qf_enqueue_diagnostics(ctx));
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
    // (on the context's own stream: the library never uses the NULL stream -- its hardware queue, created at first use, would
    // shift the pipes of every stream created after it: profiles/r06_x4_hardware_queues.txt)
    QF_HIP(hipMemcpyAsync(&st, ctx->state, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
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


int qf_oz_alloc(qf_ctx *ctx)
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


int qf_c64_isomp(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum, int reinitialize,
                 qf_isomp_stats *stats_out)
{
    QF_TRY(qf_need_c64(ctx));
    return isomp_impl(ctx, dt, steps, tol, minit, maxit, compsum, reinitialize, stats_out, false, true);
}

int qf_c64_isomp_continue(qf_ctx *ctx, double dt, int steps, double tol, int minit, int maxit, int compsum, int reinitialize,
                          qf_isomp_stats *stats_out)
{
    QF_TRY(qf_need_c64(ctx));
    return isomp_impl(ctx, dt, steps, tol, minit, maxit, compsum, reinitialize, stats_out, true, true);
}


int qf_c64_fixedpoint_products(qf_ctx *ctx, const void *Phalf_host, const void *Whalf_host, const void *W_host,
                               const void *dW_old_host, void *dW_new_host, void *Whalf_new_host, double *rowsum_host)
{
    QF_TRY(qf_need_c64(ctx));
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
    QF_TRY(qf_need_c64(ctx));
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
