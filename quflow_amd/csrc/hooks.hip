// isomp_fixedpoint and euler / heun / rk4 with HOST HOOKS (quflow/integrators/isospectral.py:338-613,
// quflow/integrators/erk.py:19-160): `forcing(P, W[, time])`, a foreign `hamiltonian(W[, time])`,
// `strang_splitting(h, W)`, `callback(W, dW)`, the general (not skew-Hermitian) branch of
// select_skewherm(False), and all of these -- plus compsum -- on (k,N,N) stacks of states.
//
// The state, the iteration vector dW, Whalf, the products and the Kahan term stay on the device for the
// whole call.  A hook is a C function pointer; what it needs crosses PCIe through pinned staging
// buffers, and nothing else does:
//     foreign hamiltonian : Whalf down, P up                       (2 matrices per iteration)
//     forcing             : P and Whalf down (Whalf only once if the Hamiltonian took it), F up
//     strang_splitting    : W down and up, twice per step          (or a resident tridiagonal solve: none)
//     callback            : W and the commutator down, once per step
// The two products run on the fp64 matrix cores (k_zgemm, plain stores), one pass (k_hook_assemble)
// forms the commutator, dW, Whalf and the residual row sums, one pass (k_hook_update) the step's
// update; the exit test reads one scalar back per iteration (the hooks synchronise the host anyway).
#include <cmath>
#include <limits>
#include <vector>

#include "qf_internal.h"

#pragma clang fp contract(off)   // the reference's passes are separate numpy operations: no fused multiply-adds

namespace {

constexpr int TH = 32;   // tile of the assemble / update passes

// One pass after the two products of an iteration (isospectral.py:499-534):
//     comm = SKEW ? PW - PW^H (conj_subtract_, :66-81, :503) : PW as it comes (the caller has already
//            subtracted Whalf @ Phalf, :505)
//     dW   = T + comm [+ F]          T = PW @ Phalf came into the dW buffer (:499), F = forcing * dt/2 (:519-520)
//     Whalf = W + dW                 (:481-482 of the next iteration)
//     rowpart[slot][row] = sum over the slot's columns of |dW_old - dW|      (:526-534)
// comm replaces PW in place (the update and the callback want it, :547-551).  Block (bi, bj), bi <= bj,
// handles the tile pair (bi,bj), (bj,bi): the in-place transposed difference needs both before either is
// overwritten.
template <bool SKEW>
__global__ __launch_bounds__(256) void k_hook_assemble(int N, cplx *__restrict__ PW, cplx *__restrict__ dW, const cplx *__restrict__ F,
                                                        const cplx *__restrict__ W, cplx *__restrict__ Whalf,
                                                        const cplx *__restrict__ dW_old, double *__restrict__ rowpart)
{
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (SKEW && bj < bi) return;
    __shared__ cplx Ta[TH][TH + 1], Tb[TH][TH + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int i0 = bi * TH, j0 = bj * TH;
    if (SKEW) {
        for (int r = ty; r < TH; r += 8) {
            const int gi = i0 + r, gj = j0 + tx;              // tile (bi,bj), row-coalesced
            Ta[r][tx] = (gi < N && gj < N) ? PW[(size_t)gi * N + gj] : make_double2(0.0, 0.0);
            const int hi = j0 + r, hj = i0 + tx;              // tile (bj,bi)
            Tb[r][tx] = (hi < N && hj < N) ? PW[(size_t)hi * N + hj] : make_double2(0.0, 0.0);
        }
        __syncthreads();
    }
    // the two tiles of the pair (one when bi == bj, or in the general branch)
    const int ntiles = (SKEW && bi != bj) ? 2 : 1;
    for (int which = 0; which < ntiles; ++which) {
        const int r0 = which ? j0 : i0, c0 = which ? i0 : j0, slot = which ? bi : bj;
        for (int r = ty; r < TH; r += 8) {
            const int gi = r0 + r, gj = c0 + tx;
            double a = 0.0;
            if (gi < N && gj < N) {
                const size_t e = (size_t)gi * N + gj;
                cplx c;
                if (SKEW) {
                    const cplx p = which ? Tb[r][tx] : Ta[r][tx];
                    const cplx q = which ? Ta[tx][r] : Tb[tx][r];     // PW[gj, gi]
                    c = make_double2(p.x - q.x, p.y + q.y);           // a - conj(a^T)
                } else {
                    c = PW[e];            // general branch: PW - Whalf @ Phalf was formed by the caller (:505)
                }
                if (SKEW) PW[e] = c;
                cplx d = dW[e];
                d.x += c.x;
                d.y += c.y;
                if (F) {
                    const cplx f = F[e];
                    d.x += f.x;
                    d.y += f.y;
                }
                dW[e] = d;
                const cplx w = W[e];
                Whalf[e] = make_double2(w.x + d.x, w.y + d.y);
                const cplx o = dW_old[e];
                const double er = o.x - d.x, ei = o.y - d.y;
                a = qf_modulus(er, ei);
            }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);   // the 32 columns of this tile row
            if (rowpart && tx == 0 && gi < N) rowpart[(size_t)slot * N + gi] = a;
        }
    }
}

// End of a step (isospectral.py:547-596): W += 2 comm (`PWcomm *= 2; W += PWcomm`, Kahan-compensated
// with the persistent term kc when KAHAN, :553-586), then W += 2 F (`FW *= 2; W += FW`, :594-596);
// Whalf = W + dW for the next step (dW zeroed first when `reinitialize`, :471-472).
template <bool KAHAN>
__global__ __launch_bounds__(256) void k_hook_update(size_t n, const cplx *__restrict__ comm, const cplx *__restrict__ F,
                                                      cplx *__restrict__ W, cplx *__restrict__ kc, cplx *__restrict__ dW,
                                                      int reinitialize, cplx *__restrict__ Whalf)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const cplx c = comm[e];
        const double dr = 2.0 * c.x, di = 2.0 * c.y;
        cplx w = W[e];
        if (KAHAN) {
            cplx k = kc[e];
            const double yr = dr - k.x, yi = di - k.y;
            const double tr = w.x + yr, ti = w.y + yi;
            k.x = (tr - w.x) - yr;
            k.y = (ti - w.y) - yi;
            kc[e] = k;
            w.x = tr;
            w.y = ti;
        } else {
            w.x += dr;
            w.y += di;
        }
        if (F) {
            const cplx f = F[e];
            w.x += 2.0 * f.x;
            w.y += 2.0 * f.y;
        }
        W[e] = w;
        if (reinitialize) {
            dW[e] = make_double2(0.0, 0.0);
            Whalf[e] = w;
        } else {
            const cplx d = dW[e];
            Whalf[e] = make_double2(w.x + d.x, w.y + d.y);
        }
    }
}

// magmp with forcing: the force term joins the iteration vector AFTER the magnetic terms (mhd.py:389-402:
// dW += PWcomm; the three magnetic updates of dW[0]; then dW += FW): dW += F, Whalf = W + dW, and -- state 0 --
// the row sums of |dW_old - dW| again (one slot per 32-column tile, as k_magnetic_fix leaves them)
__global__ __launch_bounds__(256) void k_hook_add_forcing(int N, const cplx *__restrict__ F, cplx *__restrict__ dW,
                                                          const cplx *__restrict__ dW_old, const cplx *__restrict__ W,
                                                          cplx *__restrict__ Whalf, double *__restrict__ rowpart)
{
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int i0 = blockIdx.y * TH, j0 = blockIdx.x * TH;
    for (int r = ty; r < TH; r += 8) {
        const int gi = i0 + r, gj = j0 + tx;
        double a = 0.0;
        if (gi < N && gj < N) {
            const size_t e = (size_t)gi * N + gj;
            const cplx f = F[e];
            cplx d = dW[e];
            d.x += f.x;
            d.y += f.y;
            dW[e] = d;
            const cplx w = W[e];
            Whalf[e] = make_double2(w.x + d.x, w.y + d.y);
            if (rowpart) {
                const cplx o = dW_old[e];
                const double er = o.x - d.x, ei = o.y - d.y;
                a = qf_modulus(er, ei);
            }
        }
        if (rowpart) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
            if (tx == 0 && gi < N) rowpart[(size_t)blockIdx.x * N + gi] = a;
        }
    }
}

// K = (A - B) * (1/hbar) + F (bracket(P, X) + forcing(P, X), erk.py:47-49) and the running combinations
// of qf_launch_erk_stage (elementwise.hip), for the hooked explicit steppers
__global__ __launch_bounds__(256) void k_erk_stage_forced(size_t n, const cplx *__restrict__ A, const cplx *__restrict__ B,
                                                           const cplx *__restrict__ F, double inv_hb, const cplx *W,
                                                           cplx *acc, double c_acc, cplx *Wp, double c_wp, cplx *Wout,
                                                           double c_fin)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const cplx a = A[e], b = B[e];
        double kr = (a.x - b.x) * inv_hb, ki = (a.y - b.y) * inv_hb;
        if (F) {
            const cplx f = F[e];
            kr += f.x;
            ki += f.y;
        }
        double ar = kr, ai = ki;
        if (c_acc != 0.0) {
            const cplx o = acc[e];
            ar = o.x + c_acc * kr;
            ai = o.y + c_acc * ki;
        }
        if (acc) acc[e] = make_double2(ar, ai);
        const cplx w = W[e];
        if (Wp) Wp[e] = make_double2(w.x + c_wp * kr, w.y + c_wp * ki);
        if (Wout) Wout[e] = make_double2(w.x + c_fin * ar, w.y + c_fin * ai);
    }
}

int launch_assemble(qf_ctx *ctx, bool skew, cplx *PW, cplx *dW, const cplx *F, const cplx *W, cplx *Whalf, const cplx *dW_old,
                    double *rowpart)
{
    const int tiles = (ctx->N + TH - 1) / TH;
    if (skew)
        hipLaunchKernelGGL(k_hook_assemble<true>, dim3(tiles, tiles), dim3(256), 0, ctx->stream, ctx->N, PW, dW, F, W, Whalf, dW_old,
                           rowpart);
    else
        hipLaunchKernelGGL(k_hook_assemble<false>, dim3(tiles, tiles), dim3(256), 0, ctx->stream, ctx->N, PW, dW, F, W, Whalf, dW_old,
                           rowpart);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int launch_update(qf_ctx *ctx, const cplx *comm, const cplx *F, cplx *W, cplx *kc, cplx *dW, int reinitialize, cplx *Whalf)
{
    const size_t n = (size_t)ctx->N * ctx->N;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    if (kc) hipLaunchKernelGGL(k_hook_update<true>, dim3(blocks), dim3(256), 0, ctx->stream, n, comm, F, W, kc, dW, reinitialize, Whalf);
    else hipLaunchKernelGGL(k_hook_update<false>, dim3(blocks), dim3(256), 0, ctx->stream, n, comm, F, W, kc, dW, reinitialize, Whalf);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int need_device(qf_ctx *ctx, size_t count)
{
    const size_t mbytes = (size_t)ctx->N * ctx->N * sizeof(cplx);
    while (ctx->multi.size() < count) {
        cplx *p = nullptr;
        QF_HIP(hipMalloc((void **)&p, mbytes));
        ctx->multi.push_back(p);
    }
    return QF_OK;
}

int need_host(qf_ctx *ctx, int k)
{
    const size_t bytes = (size_t)k * ctx->N * ctx->N * sizeof(cplx);
    if (ctx->hook_host_bytes < bytes) {
        for (int q = 0; q < 3; ++q) {
            if (ctx->hook_host[q]) (void)hipHostFree(ctx->hook_host[q]);
            ctx->hook_host[q] = nullptr;
        }
        ctx->hook_host_bytes = 0;
        for (int q = 0; q < 3; ++q) QF_HIP(hipHostMalloc((void **)&ctx->hook_host[q], bytes, hipHostMallocDefault));
        ctx->hook_host_bytes = bytes;
    }
    return QF_OK;
}

int hook_failed(const char *which, int rc)
{
    qf_set_error("%s hook returned %d", which, rc);
    return QF_ERR_CALLBACK;
}

int read_scalar_sync(qf_ctx *ctx, const double *dev, double *out)
{
    QF_HIP(hipMemcpyAsync(ctx->host_scalars, dev, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    *out = ctx->host_scalars[0];
    return QF_OK;
}

}  // namespace

extern "C" {

int qf_isomp_hooked(qf_ctx *ctx, void *states_host, int k, double dt, int steps, double tol, int minit, int maxit,
                    int compsum, int reinitialize, const qf_isomp_hooks *hooks, qf_isomp_stats *stats_out)
{
    if (!ctx) {
        qf_set_error("null qf_ctx");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipSetDevice(ctx->device));
    if (minit < 1) {  // isospectral.py:400
        qf_set_error("minit must be at least 1.");
        return QF_ERR_INVALID;
    }
    if (maxit < minit) {  // isospectral.py:401
        qf_set_error("maxit must be at minit.");
        return QF_ERR_INVALID;
    }
    if (!states_host || !hooks || k < 1 || steps < 0) {
        qf_set_error("qf_isomp_hooked: bad arguments (k=%d, steps=%d)", k, steps);
        return QF_ERR_INVALID;
    }
    if (compsum && hooks->forcing && steps > 0) {   // isospectral.py:588-589
        qf_set_error("Compensated sum with forcing is not yet implemented.");
        return QF_ERR_UNSUPPORTED;
    }
    const int N = ctx->N;
    const size_t NN = (size_t)N * N, mbytes = NN * sizeof(cplx);
    const double hb = qf_hbar(N);                 // isospectral.py:436
    const double vareps = dt / (2 * hb);          // isospectral.py:437
    const bool magnetic = hooks->magnetic != 0;
    if (magnetic && (k != 2 || compsum || !hooks->skewh || hooks->strang || hooks->strang_table)) {
        qf_set_error("qf_isomp_hooked: magmp (mhd.py:235-456) takes a (2,N,N) state, skew-Hermitian matrices, no compsum, no strang_splitting");
        return QF_ERR_INVALID;
    }
    const bool skew = hooks->skewh != 0;
    const bool forced = hooks->forcing != nullptr;
    const bool foreign = hooks->hamiltonian != nullptr;
    ctx->w_skew_known = false;
    ctx->increment_valid = false;

    // per state: W, dW[2], Whalf, PW (-> comm), F, Kahan term; shared: C3 (general branch), magmp: Bhalf, BT, BTP
    // states_p: the foreign Hamiltonian returns one stream matrix PER STATE ((k,N,N): np.matmul batches the products)
    // states_p < 0: the caller does not know yet -- the hook stores 0 or 1 into the field during its FIRST call and the
    // answer is read back behind that call (no entry probe: the user's function is evaluated as often as the reference
    // evaluates it); the k stream matrices are reserved up front while that is open
    bool per_state = foreign && hooks->states_p > 0 && !magnetic && k > 1;
    bool states_p_open = foreign && hooks->states_p < 0 && !magnetic && k > 1;
    if (per_state && k > 48) {
        qf_set_error("qf_isomp_hooked: a Hamiltonian with one stream matrix per state takes at most 48 states (k=%d)", k);
        return QF_ERR_UNSUPPORTED;
    }
    const size_t per = 7;
    QF_TRY(need_device(ctx, per * k + 4 + ((per_state || (states_p_open && k <= 48)) ? (size_t)k : 0)));
    QF_TRY(need_host(ctx, k));
    struct st { cplx *W, *dW[2], *Whalf, *PW, *F, *kc; int cur; };
    std::vector<st> S((size_t)k);
    for (int j = 0; j < k; ++j) {
        cplx **b = &ctx->multi[per * j];
        S[j] = {b[0], {b[1], b[2]}, b[3], b[4], b[5], b[6], 0};
        QF_HIP(hipMemcpyAsync(S[j].W, (const char *)states_host + (size_t)j * mbytes, mbytes, hipMemcpyHostToDevice, ctx->stream));
        QF_HIP(hipMemsetAsync(S[j].dW[0], 0, mbytes, ctx->stream));                      // dW = zeros_like(W), :430
        QF_HIP(hipMemcpyAsync(S[j].Whalf, S[j].W, mbytes, hipMemcpyDeviceToDevice, ctx->stream));
        if (compsum) QF_HIP(hipMemsetAsync(S[j].kc, 0, mbytes, ctx->stream));             // :457
    }
    auto Pj = [&](int j) { return per_state ? ctx->multi[per * k + 4 + j] : ctx->Phalf; };     // state j's Phalf
    cplx *C3 = ctx->multi[per * k];
    cplx *Bhalf = ctx->multi[per * k + 1], *BT = ctx->multi[per * k + 2], *BTP = ctx->multi[per * k + 3];
    cplx *hW = ctx->hook_host[0], *hP = ctx->hook_host[1], *hF = ctx->hook_host[2];
    const int slots = (N + TH - 1) / TH;
    if (!ctx->multi_rowpart) QF_HIP(hipMalloc((void **)&ctx->multi_rowpart, (size_t)2 * slots * N * sizeof(double)));   // (second half: magmp's second state, finite check)

    // tolerance from state 0 (isospectral.py:440-452)
    if (tol < 0) {
        double mach_eps = std::numeric_limits<double>::epsilon();
        if (!compsum) mach_eps = std::sqrt(mach_eps);
        double nrm = 0.0;
        QF_TRY(qf_launch_norm_inf(ctx, S[0].W, ctx->scalars));
        QF_TRY(read_scalar_sync(ctx, ctx->scalars, &nrm));
        tol = (mach_eps * dt / hb) * nrm;
    }

    auto download_stack = [&](cplx *dst, int which) -> int {     // which: 0 W, 1 Whalf, 2 comm (in PW)
        for (int j = 0; j < k; ++j) {
            const cplx *src = which == 0 ? S[j].W : which == 1 ? S[j].Whalf : S[j].PW;
            QF_HIP(hipMemcpyAsync(dst + (size_t)j * NN, src, mbytes, hipMemcpyDeviceToHost, ctx->stream));
        }
        QF_HIP(hipStreamSynchronize(ctx->stream));
        return QF_OK;
    };
    // half a Strang step on the state (isospectral.py:466-467, 598-599); Whalf = W + dW afterwards
    auto strang_half = [&]() -> int {
        if (hooks->strang_table) {
            for (int j = 0; j < k; ++j) {
                cplx *saveW = ctx->W;
                ctx->W = S[j].W;                 // the resident form of qf_solve_tridiagonal works on ctx->W
                const int rc = qf_solve_tridiagonal(ctx, hooks->strang_table, hooks->strang_key, nullptr, nullptr, hooks->solve_skewh);
                ctx->W = saveW;
                QF_TRY(rc);
            }
        } else if (hooks->strang) {
            QF_TRY(download_stack(hW, 0));
            const int rc = hooks->strang(hooks->user, dt / 2, hW);
            if (rc) return hook_failed("strang_splitting", rc);
            for (int j = 0; j < k; ++j)
                QF_HIP(hipMemcpyAsync(S[j].W, hW + (size_t)j * NN, mbytes, hipMemcpyHostToDevice, ctx->stream));
        } else {
            return QF_OK;
        }
        for (int j = 0; j < k; ++j) QF_TRY(qf_launch_lincomb(ctx, 1.0, S[j].W, 1.0, S[j].dW[S[j].cur], 0.0, S[j].Whalf));
        return QF_OK;
    };

    double time = hooks->time;
    long long total_iterations = 0, number_of_maxit = 0;
    double resnorm = 0.0;
    for (int step = 0; step < steps; ++step) {
        QF_TRY(strang_half());
        resnorm = std::numeric_limits<double>::infinity();          // :470
        // (`reinitialize`: dW was zeroed and Whalf = W set by the previous step's update, or is so at entry)
        bool broke = false;
        for (int i = 0; i < maxit; ++i) {
            total_iterations += 1;                                  // :478
            bool have_whalf_host = false;
            // ---- Phalf = vareps * hamiltonian(Whalf)             :488-492
            if (foreign) {
                QF_TRY(download_stack(hW, 1));
                have_whalf_host = true;
                const int rc = hooks->hamiltonian(hooks->user, hW, hP, hooks->hamiltonian_takes_time ? time + dt / 2 : 0.0);
                if (rc) return hook_failed("hamiltonian", rc);
                if (states_p_open) {
                    states_p_open = false;
                    per_state = hooks->states_p > 0;
                    if (per_state && k > 48) {
                        qf_set_error("qf_isomp_hooked: a Hamiltonian with one stream matrix per state takes at most 48 states (k=%d)", k);
                        return QF_ERR_UNSUPPORTED;
                    }
                }
                for (size_t e = 0; e < (magnetic ? 2 * NN : per_state ? (size_t)k * NN : NN); ++e) {   // Phalf *= vareps (magmp: Bhalf *= vareps too, mhd.py:375-376)
                    hP[e].x *= vareps;
                    hP[e].y *= vareps;
                }
                if (per_state) {
                    for (int j = 0; j < k; ++j) QF_HIP(hipMemcpyAsync(Pj(j), hP + (size_t)j * NN, mbytes, hipMemcpyHostToDevice, ctx->stream));
                } else {
                    QF_HIP(hipMemcpyAsync(ctx->Phalf, hP, mbytes, hipMemcpyHostToDevice, ctx->stream));
                }
                if (magnetic) QF_HIP(hipMemcpyAsync(Bhalf, hP + NN, mbytes, hipMemcpyHostToDevice, ctx->stream));
            } else {
                QF_TRY(qf_launch_solve(ctx, ctx->poisson, S[0].Whalf, ctx->Phalf, vareps, hooks->solve_skewh ? 1 : 0));
                if (magnetic) {      // solve_mhd (mhd.py:10-18): B = laplace(Theta)
                    QF_TRY(qf_launch_laplace(ctx, S[1].Whalf, Bhalf));
                    QF_TRY(qf_launch_lincomb(ctx, vareps, Bhalf, 0.0, nullptr, 0.0, Bhalf));
                }
            }
            // ---- the products                                    :496-505
            for (int j = 0; j < k; ++j) {
                QF_TRY(qf_launch_zgemm(ctx, Pj(j), S[j].Whalf, S[j].PW, nullptr));                       // PWcomm = Phalf @ Whalf
                QF_TRY(qf_launch_zgemm(ctx, S[j].PW, Pj(j), S[j].dW[S[j].cur ^ 1], nullptr));            // dW = PWcomm @ Phalf
                if (!skew) {                                                                             // PWcomm -= Whalf @ Phalf
                    QF_TRY(qf_launch_zgemm(ctx, S[j].Whalf, Pj(j), C3, nullptr));
                    QF_TRY(qf_launch_lincomb(ctx, 1.0, S[j].PW, -1.0, C3, 0.0, S[j].PW));
                }
            }
            if (magnetic) {                                         // mhd.py:381,385
                QF_TRY(qf_launch_zgemm(ctx, Bhalf, S[1].Whalf, BT, nullptr));       // BThetacomm = Bhalf @ Thetahalf
                QF_TRY(qf_launch_zgemm(ctx, BT, ctx->Phalf, BTP, nullptr));         // BThetaPhalf = BThetacomm @ Phalf
            }
            // ---- forcing(Phalf / vareps, Whalf[, time + dt/2]) * dt/2     :512-520  (before Whalf is rewritten)
            if (forced) {
                if (!foreign) QF_HIP(hipMemcpyAsync(hP, ctx->Phalf, mbytes, hipMemcpyDeviceToHost, ctx->stream));
                if (!have_whalf_host) QF_TRY(download_stack(hW, 1));
                else QF_HIP(hipStreamSynchronize(ctx->stream));
                const double inv = 1.0 / vareps;                    // `Phalf /= vareps`: numpy multiplies by the reciprocal
                for (size_t e = 0; e < (per_state ? (size_t)k * NN : NN); ++e) {
                    hP[e].x *= inv;
                    hP[e].y *= inv;
                }
                const int rc = hooks->forcing(hooks->user, hP, hW, hF, hooks->forcing_takes_time ? time + dt / 2 : 0.0);
                if (rc) return hook_failed("forcing", rc);
                const double half = dt / 2;
                for (size_t e = 0; e < (size_t)k * NN; ++e) {       // FW *= dt/2
                    hF[e].x *= half;
                    hF[e].y *= half;
                }
                for (int j = 0; j < k; ++j)
                    QF_HIP(hipMemcpyAsync(S[j].F, hF + (size_t)j * NN, mbytes, hipMemcpyHostToDevice, ctx->stream));
            }
            // ---- comm, dW += comm [+ F], Whalf = W + dW, residual row sums of state 0      :500-534
            // every state's residual norm is formed: with one stream matrix per state the exit test looks at all of them
            // (`resnormvec.max()`, :527-532); with a shared one it looks at state 0's, but scipy.linalg.norm checks the WHOLE
            // stack for infs / NaNs before it reduces (check_finite, :528) -- a NaN in a passively advected state raises
            // there, so it is an error return here too.  (magmp: state 0's row sums are completed by the magnetic terms
            // below; the finite check of its second state is the one of qf_isomp_states.)
            const bool norms_all = (per_state || !magnetic) && k <= 48;
            for (int j = 0; j < k; ++j) {
                double *rp = (j == 0 || norms_all) ? ctx->multi_rowpart : (magnetic && j == 1) ? ctx->multi_rowpart + (size_t)slots * N : nullptr;
                QF_TRY(launch_assemble(ctx, skew, S[j].PW, S[j].dW[S[j].cur ^ 1], (forced && !magnetic) ? S[j].F : nullptr, S[j].W,
                                       S[j].Whalf, S[j].dW[S[j].cur], rp));
                if (norms_all && i + 1 >= minit) QF_TRY(qf_launch_norm_from_rowpart(ctx, ctx->multi_rowpart, slots, ctx->scalars + 16 + j));
                // magmp: the second state's residual norm, for the finite check alone (its force term has not joined yet: a
                // non-finite force shows in the next iteration's Whalf)
                if (magnetic && j == 1 && i + 1 >= minit) QF_TRY(qf_launch_norm_from_rowpart(ctx, rp, slots, ctx->scalars + 17));
            }
            if (magnetic) {
                // the three magnetic updates of dW[0] (mhd.py:389-392), then the force term (:395-402), in that order
                QF_TRY(qf_launch_magnetic_fix(ctx, BTP, BT, S[0].dW[S[0].cur ^ 1], S[0].dW[S[0].cur], S[0].W, S[0].Whalf,
                                              ctx->multi_rowpart));
                if (forced) {
                    for (int j = 0; j < k; ++j) {
                        hipLaunchKernelGGL(k_hook_add_forcing, dim3(slots, slots), dim3(256), 0, ctx->stream, N, S[j].F,
                                           S[j].dW[S[j].cur ^ 1], S[j].dW[S[j].cur], S[j].W, S[j].Whalf,
                                           j == 0 ? ctx->multi_rowpart : nullptr);
                        QF_HIP(hipGetLastError());
                    }
                }
            }
            for (int j = 0; j < k; ++j) S[j].cur ^= 1;
            // ---- exit test on state 0                            :523-536
            if (i + 1 >= minit) {
                const double resnorm_old = resnorm;
                if (norms_all) {
                    const int kk = k < 48 ? k : 48;
                    QF_HIP(hipMemcpyAsync(ctx->host_scalars, ctx->scalars + 16, (size_t)kk * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                    QF_HIP(hipStreamSynchronize(ctx->stream));
                    resnorm = ctx->host_scalars[0];
                    for (int j = 1; j < kk; ++j) {
                        const double r = ctx->host_scalars[j];
                        if (per_state) {                 // numpy's max: a NaN wins
                            if (r != r || resnorm != resnorm) resnorm = std::numeric_limits<double>::quiet_NaN();
                            else if (r > resnorm) resnorm = r;
                        } else if (!QF_FINITE(r)) {      // shared stream matrix: state 0's residual decides, check_finite sees every state
                            qf_set_error("array must not contain infs or NaNs");
                            return QF_ERR_NONFINITE;
                        }
                    }
                } else {
                    QF_TRY(qf_launch_norm_from_rowpart(ctx, ctx->multi_rowpart, slots, ctx->scalars + 1));
                    QF_TRY(read_scalar_sync(ctx, ctx->scalars + 1, &resnorm));
                    if (magnetic && k > 1) {
                        double r1 = 0.0;
                        QF_TRY(read_scalar_sync(ctx, ctx->scalars + 17, &r1));
                        if (!QF_FINITE(r1)) resnorm = std::numeric_limits<double>::quiet_NaN();
                    }
                }
                if (!QF_FINITE(resnorm)) {       // scipy.linalg.norm raises here (isospectral.py:534)
                    qf_set_error("array must not contain infs or NaNs");
                    return QF_ERR_NONFINITE;
                }
                if (resnorm <= tol || resnorm >= resnorm_old) {
                    broke = true;
                    break;
                }
            }
        }
        if (!broke) number_of_maxit += 1;                           // :538-540
        // ---- callback(W, 2 comm) before the update               :547-551
        if (hooks->callback) {
            QF_TRY(download_stack(hW, 0));
            QF_TRY(download_stack(hF, 2));
            for (size_t e = 0; e < (size_t)k * NN; ++e) {
                hF[e].x *= 2.0;
                hF[e].y *= 2.0;
            }
            const int rc = hooks->callback(hooks->user, hW, hF);
            if (rc) return hook_failed("callback", rc);
        }
        // ---- W += 2 comm [+ 2 F];  Whalf = W + dW                :553-596
        for (int j = 0; j < k; ++j)
            QF_TRY(launch_update(ctx, S[j].PW, forced ? S[j].F : nullptr, S[j].W, compsum ? S[j].kc : nullptr, S[j].dW[S[j].cur],
                                 reinitialize, S[j].Whalf));
        if (magnetic)                                               // W[0] += 2 BThetacomm (mhd.py:431,438)
            QF_TRY(qf_launch_magnetic_update(ctx, BT, S[0].W, reinitialize ? nullptr : S[0].dW[S[0].cur], S[0].Whalf));
        if (hooks->has_time) time += dt;                            // :598-599
        QF_TRY(strang_half());
    }
    for (int j = 0; j < k; ++j)
        QF_HIP(hipMemcpyAsync((char *)states_host + (size_t)j * mbytes, S[j].W, mbytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    if (stats_out) {
        stats_out->total_iterations = total_iterations;
        stats_out->number_of_maxit = number_of_maxit;
        stats_out->tol_used = tol;
        stats_out->last_resnorm = resnorm;
    }
    return QF_OK;
}

// euler / heun / rk4 (quflow/integrators/erk.py:19-160) with `forcing(P, W)` and / or a foreign
// `hamiltonian(W)`: the state and the stage combinations stay on the device; per right-hand side a
// foreign Hamiltonian moves X down and P up, forcing moves P and X down (X only once) and F up.
int qf_erk_hooked(qf_ctx *ctx, void *W_host, int method, double dt, int steps, const qf_isomp_hooks *hooks)
{
    if (!ctx) {
        qf_set_error("null qf_ctx");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipSetDevice(ctx->device));
    if (!W_host || !hooks || method < QF_ERK_EULER || method > QF_ERK_RK4 || steps < 0) {
        qf_set_error("qf_erk_hooked: bad arguments (method=%d, steps=%d)", method, steps);
        return QF_ERR_INVALID;
    }
    const int N = ctx->N;
    const size_t NN = (size_t)N * N, mbytes = NN * sizeof(cplx);
    const double inv_hb = 1.0 / qf_hbar(N);
    QF_TRY(need_host(ctx, 1));
    QF_TRY(need_device(ctx, 1));
    ctx->w_skew_known = false;
    cplx *W = ctx->W, *Wp = ctx->Whalf, *P = ctx->Phalf, *A = ctx->PW, *B = ctx->stage, *acc = ctx->dW[0], *F = ctx->multi[0];
    cplx *hX = ctx->hook_host[0], *hP = ctx->hook_host[1], *hF = ctx->hook_host[2];
    QF_HIP(hipMemcpyAsync(W, W_host, mbytes, hipMemcpyHostToDevice, ctx->stream));
    const unsigned blocks = (unsigned)((NN + 255) / 256 < 4096 ? (NN + 255) / 256 : 4096);
    // products and forcing of one right-hand side rhs(P, X) at X (a device matrix)
    auto rhs = [&](const cplx *X) -> int {
        bool have_x = false;
        if (hooks->hamiltonian) {
            QF_HIP(hipMemcpyAsync(hX, X, mbytes, hipMemcpyDeviceToHost, ctx->stream));
            QF_HIP(hipStreamSynchronize(ctx->stream));
            have_x = true;
            const int rc = hooks->hamiltonian(hooks->user, hX, hP, 0.0);
            if (rc) return hook_failed("hamiltonian", rc);
            QF_HIP(hipMemcpyAsync(P, hP, mbytes, hipMemcpyHostToDevice, ctx->stream));
        } else {
            QF_TRY(qf_launch_solve(ctx, ctx->poisson, X, P, 1.0, hooks->solve_skewh ? 1 : 0));
        }
        QF_TRY(qf_launch_zgemm(ctx, P, X, A, nullptr));      // bracket(P, X) = (P@X - X@P)/hbar, geometry.py:41-49
        QF_TRY(qf_launch_zgemm(ctx, X, P, B, nullptr));
        if (hooks->forcing) {
            if (!hooks->hamiltonian) QF_HIP(hipMemcpyAsync(hP, P, mbytes, hipMemcpyDeviceToHost, ctx->stream));
            if (!have_x) QF_HIP(hipMemcpyAsync(hX, X, mbytes, hipMemcpyDeviceToHost, ctx->stream));
            QF_HIP(hipStreamSynchronize(ctx->stream));
            const int rc = hooks->forcing(hooks->user, hP, hX, hF, 0.0);
            if (rc) return hook_failed("forcing", rc);
            QF_HIP(hipMemcpyAsync(F, hF, mbytes, hipMemcpyHostToDevice, ctx->stream));
        }
        return QF_OK;
    };
    const cplx *Fk = hooks->forcing ? F : nullptr;
    auto stage = [&](cplx *acc_, double c_acc, cplx *Wp_, double c_wp, cplx *Wout_, double c_fin) -> int {
        hipLaunchKernelGGL(k_erk_stage_forced, dim3(blocks), dim3(256), 0, ctx->stream, NN, A, B, Fk, inv_hb, W, acc_, c_acc, Wp_,
                           c_wp, Wout_, c_fin);
        QF_HIP(hipGetLastError());
        return QF_OK;
    };
    for (int s = 0; s < steps; ++s) {
        if (method == QF_ERK_EULER) {           // erk.py:53-56
            QF_TRY(rhs(W));
            QF_TRY(stage(nullptr, 0.0, nullptr, 0.0, W, dt));
        } else if (method == QF_ERK_HEUN) {     // erk.py:101-110
            QF_TRY(rhs(W));
            QF_TRY(stage(acc, 0.0, Wp, dt, nullptr, 0.0));
            QF_TRY(rhs(Wp));
            QF_TRY(stage(acc, 1.0, nullptr, 0.0, W, dt / 2.0));
        } else {                                // erk.py:146-156
            QF_TRY(rhs(W));
            QF_TRY(stage(acc, 0.0, Wp, dt / 2.0, nullptr, 0.0));
            QF_TRY(rhs(Wp));
            QF_TRY(stage(acc, 2.0, Wp, dt / 2.0, nullptr, 0.0));
            QF_TRY(rhs(Wp));
            QF_TRY(stage(acc, 2.0, Wp, dt, nullptr, 0.0));
            QF_TRY(rhs(Wp));
            QF_TRY(stage(acc, 1.0, nullptr, 0.0, W, dt / 6.0));
        }
    }
    QF_HIP(hipMemcpyAsync(W_host, W, mbytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

// The same on a (k,N,N) stack of states (erk.py with batched input): `hamiltonian(stack)` returns ONE (N,N) stream
// matrix (the built-in one solves for state 0, cpu.py:696-697), bracket(P, X_j) for every state, `forcing(P, stack)`
// returns a stack.  States, stage arguments and accumulators stay on the device; per right-hand side a foreign
// Hamiltonian moves the stage arguments down and P up, forcing moves P and the stage arguments down (once) and F up.
int qf_erk_states_hooked(qf_ctx *ctx, void *states_host, int k, int method, double dt, int steps, const qf_isomp_hooks *hooks)
{
    if (!ctx) {
        qf_set_error("null qf_ctx");
        return QF_ERR_INVALID;
    }
    QF_HIP(hipSetDevice(ctx->device));
    if (!states_host || !hooks || k < 1 || method < QF_ERK_EULER || method > QF_ERK_RK4 || steps < 0) {
        qf_set_error("qf_erk_states_hooked: bad arguments (k=%d, method=%d, steps=%d)", k, method, steps);
        return QF_ERR_INVALID;
    }
    const int N = ctx->N;
    const size_t NN = (size_t)N * N, mbytes = NN * sizeof(cplx);
    const double inv_hb = 1.0 / qf_hbar(N);
    QF_TRY(need_host(ctx, k));
    QF_TRY(need_device(ctx, (size_t)5 * k));           // per state: X, stage argument, accumulator, forcing term, stream matrix
    ctx->w_skew_known = false;
    cplx *P = ctx->Phalf, *A = ctx->PW, *B = ctx->stage;
    cplx *hX = ctx->hook_host[0], *hP = ctx->hook_host[1], *hF = ctx->hook_host[2];
    auto X = [&](int j) { return ctx->multi[(size_t)5 * j]; };
    auto Xp = [&](int j) { return ctx->multi[(size_t)5 * j + 1]; };
    auto Acc = [&](int j) { return ctx->multi[(size_t)5 * j + 2]; };
    auto F = [&](int j) { return ctx->multi[(size_t)5 * j + 3]; };
    auto Pj = [&](int j) { return ctx->multi[(size_t)5 * j + 4]; };
    // hooks->states_p: the foreign Hamiltonian fills one stream matrix PER STATE (bracket(P, W) batched, erk.py with a
    // (k,N,N) P); otherwise one for all states
    bool per_state = hooks->hamiltonian && hooks->states_p > 0;      // (< 0: settled by the hook's first call, as in qf_isomp_hooked)
    bool states_p_open = hooks->hamiltonian && hooks->states_p < 0;
    for (int j = 0; j < k; ++j)
        QF_HIP(hipMemcpyAsync(X(j), (const char *)states_host + (size_t)j * mbytes, mbytes, hipMemcpyHostToDevice, ctx->stream));
    const unsigned blocks = (unsigned)((NN + 255) / 256 < 4096 ? (NN + 255) / 256 : 4096);
    // one stage for the whole stack at the stage arguments (the states themselves when from_state)
    auto stage_all = [&](bool from_state, double c_acc, bool want_wp, double c_wp, bool fin, double c_fin) -> int {
        auto arg = [&](int j) { return from_state ? X(j) : Xp(j); };
        bool have_x = false;
        auto stack_down = [&]() -> int {
            if (have_x) return QF_OK;
            for (int j = 0; j < k; ++j) QF_HIP(hipMemcpyAsync(hX + (size_t)j * NN, arg(j), mbytes, hipMemcpyDeviceToHost, ctx->stream));
            have_x = true;
            return QF_OK;
        };
        if (hooks->hamiltonian) {
            QF_TRY(stack_down());
            QF_HIP(hipStreamSynchronize(ctx->stream));
            const int rc = hooks->hamiltonian(hooks->user, hX, hP, 0.0);
            if (rc) return hook_failed("hamiltonian", rc);
            if (states_p_open) {
                states_p_open = false;
                per_state = hooks->states_p > 0;
            }
            if (per_state) {
                for (int j = 0; j < k; ++j) QF_HIP(hipMemcpyAsync(Pj(j), hP + (size_t)j * NN, mbytes, hipMemcpyHostToDevice, ctx->stream));
            } else {
                QF_HIP(hipMemcpyAsync(P, hP, mbytes, hipMemcpyHostToDevice, ctx->stream));
            }
        } else {
            QF_TRY(qf_launch_solve(ctx, ctx->poisson, arg(0), P, 1.0, hooks->solve_skewh ? 1 : 0));
        }
        if (hooks->forcing) {
            if (!hooks->hamiltonian) QF_HIP(hipMemcpyAsync(hP, P, mbytes, hipMemcpyDeviceToHost, ctx->stream));
            QF_TRY(stack_down());
            QF_HIP(hipStreamSynchronize(ctx->stream));
            const int rc = hooks->forcing(hooks->user, hP, hX, hF, 0.0);
            if (rc) return hook_failed("forcing", rc);
            for (int j = 0; j < k; ++j) QF_HIP(hipMemcpyAsync(F(j), hF + (size_t)j * NN, mbytes, hipMemcpyHostToDevice, ctx->stream));
        }
        for (int j = 0; j < k; ++j) {
            const cplx *Pm = per_state ? Pj(j) : P;
            QF_TRY(qf_launch_zgemm(ctx, Pm, arg(j), A, nullptr));      // bracket(P, X_j) = (P@X_j - X_j@P)/hbar, geometry.py:41-49
            QF_TRY(qf_launch_zgemm(ctx, arg(j), Pm, B, nullptr));
            hipLaunchKernelGGL(k_erk_stage_forced, dim3(blocks), dim3(256), 0, ctx->stream, NN, A, B, hooks->forcing ? F(j) : nullptr, inv_hb,
                               X(j), (c_acc == 0.0 && !want_wp && fin) ? nullptr : Acc(j), c_acc, want_wp ? Xp(j) : nullptr, c_wp,
                               fin ? X(j) : nullptr, c_fin);
            QF_HIP(hipGetLastError());
        }
        return QF_OK;
    };
    for (int s = 0; s < steps; ++s) {
        if (method == QF_ERK_EULER) {           // erk.py:53-56
            QF_TRY(stage_all(true, 0.0, false, 0.0, true, dt));
        } else if (method == QF_ERK_HEUN) {     // erk.py:101-110
            QF_TRY(stage_all(true, 0.0, true, dt, false, 0.0));
            QF_TRY(stage_all(false, 1.0, false, 0.0, true, dt / 2.0));
        } else {                                // erk.py:146-156
            QF_TRY(stage_all(true, 0.0, true, dt / 2.0, false, 0.0));
            QF_TRY(stage_all(false, 2.0, true, dt / 2.0, false, 0.0));
            QF_TRY(stage_all(false, 2.0, true, dt, false, 0.0));
            QF_TRY(stage_all(false, 1.0, false, 0.0, true, dt / 6.0));
        }
    }
    for (int j = 0; j < k; ++j)
        QF_HIP(hipMemcpyAsync((char *)states_host + (size_t)j * mbytes, X(j), mbytes, hipMemcpyDeviceToHost, ctx->stream));
    QF_HIP(hipStreamSynchronize(ctx->stream));
    return QF_OK;
}

}  // extern "C"
