// C ABI of libquflow_hip.so, part 4 of 5: the OTHER steppers on the same kernels -- euler / heun / rk4
// (quflow/integrators/erk.py), isomp_simple / isomp_quasinewton (isospectral.py:155-335), isomp / magmp on stacks of
// states (host loop per iteration).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>

#include "qf_api.h"

extern "C" {

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
        if (!QF_FINITE(en)) {       // (scipy.linalg.lu_factor checks its argument: isospectral.py:211, 290)
            qf_set_error("array must not contain infs or NaNs");
            return QF_ERR_NONFINITE;
        }
        if (en > 1e150) {
            // finite, but 1 + |E|^2 overflows: the Newton-Schulz start has no scale.  The reference's LU would go on (and
            // produce nothing usable): an error of this implementation, named as one -- not the reference's ValueError
            qf_set_error("isomp_simple / isomp_quasinewton: |dt/(2 hbar) P|_inf = %.3g is too large for the Newton-Schulz inverse (limit 1e150)", en);
            return QF_ERR_STATE;
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
    if (!ctx->multi_rowpart) QF_HIP(hipMalloc((void **)&ctx->multi_rowpart, (size_t)2 * slots32 * N * sizeof(double)));   // (sized as hooks.hip sizes it)
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
                        QF_TRY_R(qf_launch_norm_from_rowpart(ctx, ctx->rowpart, qf_rowpart_slots(ctx), ctx->scalars + 1));
                    }
                } else if (j < 48 && i + 1 >= minit) {
                    // the exit test looks at state 0's residual, but scipy.linalg.norm checks the WHOLE stack for infs / NaNs
                    // before it reduces (check_finite, isospectral.py:528): the other states' norms are formed for that check
                    QF_TRY_R(qf_launch_norm_from_rowpart(ctx, ctx->rowpart, qf_rowpart_slots(ctx), ctx->scalars + 16 + j));
                }
                S[j].cur ^= 1;
            }
            if (i + 1 >= minit) {
                const double resnorm_old = resnorm;
                const int kk = k < 48 ? k : 48;
                if (kk > 1 && hipMemcpyAsync(ctx->host_scalars + 1, ctx->scalars + 17, (size_t)(kk - 1) * sizeof(double), hipMemcpyDeviceToHost,
                                             ctx->stream) != hipSuccess) {
                    qf_set_error("qf_isomp_states: copy of the states' residual norms failed");
                    return restore(QF_ERR_HIP);
                }
                QF_TRY_R(read_scalar(ctx, ctx->scalars + 1, &resnorm));
                bool finite = QF_FINITE(resnorm);
                for (int j = 1; j < kk; ++j) finite = finite && QF_FINITE(ctx->host_scalars[j]);
                if (!finite) {       // scipy.linalg.norm raises here (isospectral.py:534, mhd.py: same test)
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


}  // extern "C"
