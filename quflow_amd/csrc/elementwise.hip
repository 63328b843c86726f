// HBM-bound passes of the isospectral stepper that are not fused into a GEMM epilogue:
// the end-of-step update W += 2 (PW - PW^H) (isospectral.py:547-592, with the Kahan
// variant :553-586), the residual / tolerance norms (isospectral.py:448,534) and the
// diagnostics reductions (quflow/physics.py:26-38, quflow/geometry.py:72-76).
#include "qf_internal.h"

#pragma clang fp contract(off)  // Kahan summation must not be re-associated or fused

namespace {

constexpr int TU = 32;  // update tile

// End-of-step bookkeeping, executed by the LAST block of k_update to finish (ticket counter), i.e.
// after every block has read the control state: advance the step counter, reset the per-step
// fields, publish progress to the pinned host record.  If the step is NOT finished (the host
// enqueued fewer iterations than it needed) nothing changes and the host sees that the
// advance ran without the step counter moving.
__device__ void qf_step_advance(qf_dev_state *state, qf_host_record *rec, const qf_guard &guard)
{
    const bool mine = state->step_index == guard.step;
    const bool complete = mine && (state->step_done != 0 || state->iters_this_step >= state->maxit);
    int incomplete = 0;
    if (complete) {
        if (!state->step_done) state->number_of_maxit += 1;   // for-else, isospectral.py:538-540
        rec->last_step_iters = state->iters_this_step;
        state->step_index += 1;
        state->iters_this_step = 0;
        state->step_done = 0;
        rec->resnorm = state->resnorm;
        state->resnorm = __builtin_inf();                     // isospectral.py:470
    } else if (mine) {
        incomplete = 1;
    }
    rec->total_iterations = state->total_iterations;
    rec->number_of_maxit = state->number_of_maxit;
    rec->step_index = state->step_index;
    rec->incomplete = incomplete;
    if (state->fault == QF_FAULT_NONFINITE) rec->nonfinite = 1;      // k_norm_decide closed the call (QF_STEP_ABORTED)
    __hip_atomic_store(&rec->seq, rec->seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}


// W += 2 * (PW - PW^H);  Whalf = W + dW  (dW == nullptr: Whalf = W).
// Tile (bi,bj) reads PW tiles (bi,bj) and (bj,bi); the mirrored one goes through LDS so that
// both global reads are row-coalesced.
template <bool KAHAN>
__global__ __launch_bounds__(256) void k_update(int N, const cplx *__restrict__ PW, cplx *__restrict__ W,
                                                 cplx *dW_a, cplx *dW_b, cplx *__restrict__ Whalf,
                                                 cplx *__restrict__ kc, int reinitialize, qf_guard guard,
                                                 qf_dev_state *state, qf_host_record *rec, unsigned *ticket)
{
    // the update runs once the iteration of step `guard.step` has finished (break taken or
    // maxit reached); otherwise the launch only takes part in the end-of-step ticket
    const bool due = qf_guard_step_end(guard);
    if (due) {
    // current iteration vector: the device knows how many iterations were executed
    cplx *dWc = (guard.state && guard.state->dw_parity) ? dW_b : dW_a;
    const cplx *dW = reinitialize ? nullptr : dWc;
    __shared__ cplx Ts[TU][TU + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int i0 = blockIdx.y * TU, j0 = blockIdx.x * TU;
    for (int r = ty; r < TU; r += 8) {
        int gj = j0 + r, gi = i0 + tx;  // row gj of the mirrored tile, column gi
        cplx tv = make_double2(0.0, 0.0);
        if (gj < N && gi < N) tv = PW[(size_t)gj * N + gi];
        Ts[r][tx] = tv;
    }
    __syncthreads();
    for (int r = ty; r < TU; r += 8) {
        int gi = i0 + r, gj = j0 + tx;
        if (gi < N && gj < N) {
            const size_t e = (size_t)gi * N + gj;
            const cplx pw = PW[e];
            const cplx pwt = Ts[tx][r];
            // 2 * (PW[i,j] - conj(PW[j,i]))   (conj_subtract_ then `PWcomm *= 2`, isospectral.py:503,547)
            const double dr = 2.0 * (pw.x - pwt.x);
            const double di = 2.0 * (pw.y + pwt.y);
            cplx w = W[e];
            if (KAHAN) {
                // isospectral.py:568-586:  y = d - c;  t = W + y;  c = (t - W) - y;  W = t
                cplx c = kc[e];
                const double yr = dr - c.x, yi = di - c.y;
                const double tr = w.x + yr, ti = w.y + yi;
                c.x = (tr - w.x) - yr;
                c.y = (ti - w.y) - yi;
                kc[e] = c;
                w.x = tr;
                w.y = ti;
            } else {
                w.x += dr;  // isospectral.py:592
                w.y += di;
            }
            W[e] = w;
            if (dW) {
                const cplx d = dW[e];
                Whalf[e] = make_double2(w.x + d.x, w.y + d.y);  // isospectral.py:481-482 of the next step
            } else {
                Whalf[e] = w;
                dWc[e] = make_double2(0.0, 0.0);                // dW.fill(0), isospectral.py:471-472
            }
        }
    }
    }   // due
    if (state) {
        // last block to finish performs the step bookkeeping (it knows that every other block
        // has already read the control state)
        // two-level ticket (a contended device-scope atomic costs ~12 ns: one counter per
        // grid row, then one for the rows): ticket[1 + y] counts the blocks of row y, ticket[0]
        // the finished rows
        __syncthreads();
        if (threadIdx.x == 0) {
            if (atomicAdd(&ticket[1 + blockIdx.y], 1u) == gridDim.x - 1) {
                ticket[1 + blockIdx.y] = 0;
                if (atomicAdd(&ticket[0], 1u) == gridDim.y - 1) {
                    ticket[0] = 0;
                    qf_step_advance(state, rec, guard);
                }
            }
        }
    }
}


// One stage of the explicit Runge-Kutta steppers (quflow/integrators/erk.py:19-160) on the
// products A = P@X (and B = X@P):
//   K = (A - B) / hbar                      bracket(), quflow/geometry.py:41-49; numpy divides a
//                                           complex array by a real scalar by MULTIPLYING with
//                                           1/hbar (its complex-division loop with a zero
//                                           imaginary divisor), hence inv_hb
//   acc = c_acc == 0 ? K : acc + c_acc*K    running combination of the stage slopes
//   Wp   = W + c_wp*K                       next stage's argument (erk.py:142,146,150)
//   Wout = W + c_fin*acc                    the update (erk.py:56,110,156)
// SKEW: P and X skew-Hermitian => X@P = (P@X)^H, so B is the mirrored A tile (through LDS so
// that both global reads are row-coalesced) and only one product is needed per stage.
template <bool SKEW>
__global__ __launch_bounds__(256) void k_erk_stage(int N, const cplx *__restrict__ A, const cplx *__restrict__ B,
                                                    double inv_hb, const cplx *W, cplx *acc, double c_acc,
                                                    cplx *Wp, double c_wp, cplx *Wout, double c_fin)
{
    __shared__ cplx Ts[TU][TU + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int i0 = blockIdx.y * TU, j0 = blockIdx.x * TU;
    if (SKEW) {
        for (int r = ty; r < TU; r += 8) {
            const int gj = j0 + r, gi = i0 + tx;  // row gj of the mirrored tile, column gi
            cplx tv = make_double2(0.0, 0.0);
            if (gj < N && gi < N) tv = A[(size_t)gj * N + gi];
            Ts[r][tx] = tv;
        }
        __syncthreads();
    }
    for (int r = ty; r < TU; r += 8) {
        const int gi = i0 + r, gj = j0 + tx;
        if (gi < N && gj < N) {
            const size_t e = (size_t)gi * N + gj;
            const cplx a = A[e];
            cplx b;
            if (SKEW) {
                const cplx t = Ts[tx][r];
                b = make_double2(t.x, -t.y);      // conj(A[j,i])
            } else {
                b = B[e];
            }
            const double kr = (a.x - b.x) * inv_hb, ki = (a.y - b.y) * inv_hb;
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
}

// out = a*X + b*Y + c*I  (Y may be nullptr; out may alias X or Y): the O(N^2) glue of the
// Newton-Schulz linear solves of isomp_simple / isomp_quasinewton (api_steppers.hip)
__global__ __launch_bounds__(256) void k_lincomb(int N, double a, const cplx *X, double b, const cplx *Y, double c,
                                                  cplx *out)
{
    const size_t n = (size_t)N * N;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const cplx x = X[e];
        double re = a * x.x, im = a * x.y;
        if (Y) {
            const cplx y = Y[e];
            re += b * y.x;
            im += b * y.y;
        }
        if (c != 0.0 && e % ((size_t)N + 1) == 0) re += c;
        out[e] = make_double2(re, im);
    }
}

// out = -X^H (tile transpose through LDS: both global accesses row-coalesced)
__global__ __launch_bounds__(256) void k_neg_conj_transpose(int N, const cplx *__restrict__ X, cplx *__restrict__ out)
{
    __shared__ cplx Ts[TU][TU + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.y * TU, j0 = blockIdx.x * TU;
    for (int r = ty; r < TU; r += 8) {
        const int gj = j0 + r, gi = i0 + tx;
        cplx tv = make_double2(0.0, 0.0);
        if (gj < N && gi < N) tv = X[(size_t)gj * N + gi];
        Ts[r][tx] = tv;
    }
    __syncthreads();
    for (int r = ty; r < TU; r += 8) {
        const int gi = i0 + r, gj = j0 + tx;
        if (gi < N && gj < N) {
            const cplx t = Ts[tx][r];
            out[(size_t)gi * N + gj] = make_double2(-t.x, t.y);
        }
    }
}

// X[j,i] = -conj(X[i,j]) for i < j: restores the lower triangle of a skew-Hermitian matrix from its
// upper one (the fused upper-triangle second product leaves W and dW above the diagonal only,
// zgemm.hip).  One block per 32 x 32 tile strictly below the diagonal, both accesses row-coalesced.
__global__ __launch_bounds__(256) void k_mirror_lower(int N, cplx *__restrict__ X)
{
    __shared__ cplx Ts[TU][TU + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int bi = blockIdx.y, bj = blockIdx.x;    // target tile (rows bi, columns bj)
    if (bj > bi) return;
    const int i0 = bi * TU, j0 = bj * TU;
    for (int r = ty; r < TU; r += 8) {
        const int gj = j0 + r, gi = i0 + tx;       // source: row gj, column gi (above the diagonal when gj < gi)
        cplx tv = make_double2(0.0, 0.0);
        if (gj < N && gi < N) tv = X[(size_t)gj * N + gi];
        Ts[r][tx] = tv;
    }
    __syncthreads();
    for (int r = ty; r < TU; r += 8) {
        const int gi = i0 + r, gj = j0 + tx;
        if (gi < N && gj < N && gj < gi) {
            const cplx t = Ts[tx][r];
            X[(size_t)gi * N + gj] = make_double2(-t.x, t.y);
        }
    }
}

// magmp (quflow/integrators/mhd.py:235-456), vorticity state only: after the fused second product
// has produced dW' = PWcomm@Phalf + (PWcomm - PWcomm^H), add the magnetic terms in the
// reference's order (mhd.py:389-392)
//     dW = ((dW' + BTP[i,j]) - conj(BTP[j,i])) + (BT[i,j] - conj(BT[j,i]))
// (BTP = BThetacomm@Phalf, BT = Bhalf@Thetahalf before conj_subtract_), rebuild
// Whalf = W + dW and the row sums of |dW_old - dW| (one slot per column tile).
__global__ __launch_bounds__(256) void k_magnetic_fix(int N, const cplx *__restrict__ BTP, const cplx *__restrict__ BT,
                                                       cplx *__restrict__ dW, const cplx *__restrict__ dW_old,
                                                       const cplx *__restrict__ W, cplx *__restrict__ Whalf,
                                                       double *__restrict__ rowpart)
{
    __shared__ cplx Tp[TU][TU + 1], Tb[TU][TU + 1];
    __shared__ double rs[8][TU];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int i0 = blockIdx.y * TU, j0 = blockIdx.x * TU;
    for (int r = ty; r < TU; r += 8) {
        const int gj = j0 + r, gi = i0 + tx;  // row gj of the mirrored tile, column gi
        cplx a = make_double2(0.0, 0.0), b = a;
        if (gj < N && gi < N) {
            a = BTP[(size_t)gj * N + gi];
            b = BT[(size_t)gj * N + gi];
        }
        Tp[r][tx] = a;
        Tb[r][tx] = b;
    }
    __syncthreads();
    for (int r = ty; r < TU; r += 8) {
        const int gi = i0 + r, gj = j0 + tx;
        double a = 0.0;
        if (gi < N && gj < N) {
            const size_t e = (size_t)gi * N + gj;
            const cplx bp = BTP[e], bpt = Tp[tx][r], bt = BT[e], btt = Tb[tx][r];
            cplx d = dW[e];
            d.x += bp.x;  d.y += bp.y;              // dW += BThetaPhalf
            d.x -= bpt.x; d.y += bpt.y;             // dW -= BThetaPhalf^H
            d.x += bt.x - btt.x;                    // dW += conj_subtract_(BThetacomm)
            d.y += bt.y + btt.y;
            dW[e] = d;
            const cplx w = W[e];
            Whalf[e] = make_double2(w.x + d.x, w.y + d.y);
            const cplx o = dW_old[e];
            const double er = o.x - d.x, ei = o.y - d.y;
            a = qf_modulus(er, ei);
        }
        // sum over the 32 columns of this tile row (lanes tx of one half-wave)
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if (tx == 0 && gi < N) rowpart[(size_t)blockIdx.x * N + gi] = a;
    }
    (void)rs;
}

// end of a magmp step, vorticity state: W += 2 (BT - BT^H) after the common update, and the next
// step's Whalf = W + dW (dW == nullptr: reinitialize, Whalf = W)     (mhd.py:436-441)
__global__ __launch_bounds__(256) void k_magnetic_update(int N, const cplx *__restrict__ BT, cplx *__restrict__ W,
                                                          const cplx *__restrict__ dW, cplx *__restrict__ Whalf)
{
    __shared__ cplx Tb[TU][TU + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int i0 = blockIdx.y * TU, j0 = blockIdx.x * TU;
    for (int r = ty; r < TU; r += 8) {
        const int gj = j0 + r, gi = i0 + tx;
        cplx b = make_double2(0.0, 0.0);
        if (gj < N && gi < N) b = BT[(size_t)gj * N + gi];
        Tb[r][tx] = b;
    }
    __syncthreads();
    for (int r = ty; r < TU; r += 8) {
        const int gi = i0 + r, gj = j0 + tx;
        if (gi < N && gj < N) {
            const size_t e = (size_t)gi * N + gj;
            const cplx bt = BT[e], btt = Tb[tx][r];
            cplx w = W[e];
            w.x += 2.0 * (bt.x - btt.x);
            w.y += 2.0 * (bt.y + btt.y);
            W[e] = w;
            if (dW) {
                const cplx d = dW[e];
                Whalf[e] = make_double2(w.x + d.x, w.y + d.y);
            } else {
                Whalf[e] = w;
            }
        }
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// rowsum[i] = sum_j |A[i,j]|, one 256-thread block per row (fixed reduction tree)
__global__ __launch_bounds__(256) void k_row_abs_sum(int N, const cplx *__restrict__ A, double *__restrict__ rowsum)
{
    __shared__ double part[4];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int j = threadIdx.x; j < N; j += 256) {
        cplx z = A[(size_t)i * N + j];
        s += hypot(z.x, z.y);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) rowsum[i] = (part[0] + part[1]) + (part[2] + part[3]);
}

// out = max_i sum_t parts[t*N + i]   (tiles == 1: plain max over rowsum); single block.
__global__ __launch_bounds__(1024) void k_max_rows(int N, int tiles, const double *__restrict__ parts,
                                                    double *__restrict__ out)
{
    __shared__ double part[16];
    double m = 0.0;
    bool nan = false;
    for (int i = threadIdx.x; i < N; i += 1024) {
        double s = 0.0;
        for (int t = 0; t < tiles; ++t) s += parts[(size_t)t * N + i];
        if (s != s) nan = true;
        m = fmax(m, s);
    }
    if (nan) m = __builtin_nan("");
    // fmax drops NaNs: carry them explicitly so that a diverged iteration is visible
    double isn = wave_max(nan ? 1.0 : 0.0);
    m = wave_max(nan ? 0.0 : m);
    if (isn > 0.0) m = __builtin_nan("");
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        bool anynan = false;
        for (int w = 0; w < 16; ++w) {
            if (part[w] != part[w]) anynan = true;
            else r = fmax(r, part[w]);
        }
        out[0] = anynan ? __builtin_nan("") : r;
    }
}

// max |A[i,j] + conj(A[j,i])| and max |A[i,j]| over the rows of one block (NaN-propagating:
// a NaN anywhere makes the defect NaN, which fails the host's `<=` test).
__global__ __launch_bounds__(256) void k_skew_defect_partial(int N, const cplx *__restrict__ A,
                                                              double *__restrict__ defect, double *__restrict__ amax)
{
    __shared__ double sd[4], sa[4];
    double d = 0.0, a = 0.0;
    bool nan = false;
    for (int i = blockIdx.x; i < N; i += gridDim.x)
        for (int j = threadIdx.x; j < N; j += 256) {
            const cplx x = A[(size_t)i * N + j], y = A[(size_t)j * N + i];
            const double dr = x.x + y.x, di = x.y - y.y;
            const double dd = fmax(fabs(dr), fabs(di));
            if (dd != dd) nan = true;
            d = fmax(d, dd);
            a = fmax(a, fmax(fabs(x.x), fabs(x.y)));
        }
    if (nan) d = __builtin_inf();
    d = wave_max(d);
    a = wave_max(a);
    if ((threadIdx.x & 63) == 0) {
        sd[threadIdx.x >> 6] = d;
        sa[threadIdx.x >> 6] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        defect[blockIdx.x] = fmax(fmax(sd[0], sd[1]), fmax(sd[2], sd[3]));
        amax[blockIdx.x] = fmax(fmax(sa[0], sa[1]), fmax(sa[2], sa[3]));
    }
}

// rowsum[i] = sum over the column-tile slots, in the order k_norm_decide uses
__global__ void k_sum_rowpart(int N, int tiles, const double *__restrict__ parts, double *__restrict__ rowsum)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double s = 0.0;
    for (int t = 0; t < tiles; ++t) s += parts[(size_t)t * N + i];
    rowsum[i] = s;
}

__global__ void k_max_partials2(int n, const double *__restrict__ p0, const double *__restrict__ p1,
                                double *__restrict__ out)
{
    double d = 0.0, a = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) {
        d = fmax(d, p0[i]);
        a = fmax(a, p1[i]);
    }
    d = wave_max(d);
    a = wave_max(a);
    if (threadIdx.x == 0) {
        out[0] = d;
        out[1] = a;
    }
}

// Residual norm of iteration guard.iter and the exit decision of isospectral.py:523-536.
// Single block; rows are summed over the column tiles in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void k_norm_decide(int N, int tiles, const double *__restrict__ parts,
                                                       qf_dev_state *state, qf_guard guard)
{
    if (!qf_guard_iter(guard)) return;
    __shared__ double part[16];
    __shared__ int nanflag[16];
    const bool check = (guard.iter + 1 >= state->minit);   // uniform
    double m = 0.0;
    int nan = 0;
    if (check) {
        for (int i = threadIdx.x; i < N; i += 1024) {
            double s = 0.0;
            for (int t = 0; t < tiles; ++t) s += parts[(size_t)t * N + i];
            if (s != s) nan = 1; else m = fmax(m, s);
        }
        m = wave_max(m);
        double nn = wave_max((double)nan);
        if ((threadIdx.x & 63) == 0) {
            part[threadIdx.x >> 6] = m;
            nanflag[threadIdx.x >> 6] = nn > 0.0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        state->iters_this_step = guard.iter + 1;
        state->total_iterations += 1;                       // isospectral.py:478
        state->dw_parity ^= 1;                              // the product just wrote the other buffer
        if (check) {
            double r = 0.0;
            bool anynan = false;
            for (int w = 0; w < 16; ++w) {
                if (nanflag[w]) anynan = true;
                r = fmax(r, part[w]);
            }
            if (anynan) r = __builtin_nan("");
            const double resnorm_old = state->resnorm;      // isospectral.py:525
            state->resnorm = r;
            if (!QF_FINITE(r)) {
                // scipy.linalg.norm raises there (isospectral.py:534) with W as the last completed step left it: close the
                // call -- no tag matches the parked counter, so neither this step's update nor anything queued behind it runs
                state->fault = QF_FAULT_NONFINITE;
                state->step_index = QF_STEP_ABORTED;
                state->iters_this_step = 0;
            } else if (r <= state->tol || r >= resnorm_old) state->step_done = 1;   // isospectral.py:535-536
        }
    }
}

__device__ void qf_state_reset(qf_dev_state *state, qf_host_record *rec, double tol, int minit, int maxit);

__global__ void k_state_init(qf_dev_state *state, qf_host_record *rec, double tol, int minit, int maxit,
                             const double *norm_dev, double tol_factor)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (norm_dev) tol = tol_factor * norm_dev[0];          // isospectral.py:446-448
    qf_state_reset(state, rec, tol, minit, maxit);
}

__device__ void qf_state_reset(qf_dev_state *state, qf_host_record *rec, double tol, int minit, int maxit)
{
    state->resnorm = __builtin_inf();
    state->tol = tol;
    rec->tol = tol;
    rec->w_parity = 0;
    rec->wh_sel = 0;
    rec->dw_parity = 0;
    rec->fault = 0;
    rec->nonfinite = 0;
    state->total_iterations = 0;
    state->number_of_maxit = 0;
    state->step_index = 0;
    state->iters_this_step = 0;
    state->step_done = 0;
    state->minit = minit;
    state->maxit = maxit;
    state->dw_parity = 0;
    state->fault = 0;
    state->w_parity = 0;
    state->wh_sel = 0;
    state->pending = 0;
    state->pending_iter = 0;
    rec->total_iterations = 0;
    rec->number_of_maxit = 0;
    rec->resnorm = __builtin_inf();
    rec->step_index = 0;
    rec->last_step_iters = 0;
    rec->incomplete = 0;
    rec->progress = 0ull;
    __hip_atomic_store(&rec->seq, 0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// "Am I the last block of this launch?" for a launch of nb blocks, with at most 32 blocks meeting at any
// one counter: t[1 + g] counts the blocks of group g (32 consecutive block ids), t[0] the finished groups.
// (One device atomic on a contended address costs ~12-20 ns; 1024 blocks on one counter were 20 us.)
// The caller (one thread) has stored its partial result with relaxed agent-scope stores -- write-through --
// before the call, and the last block reads the partials with relaxed agent-scope loads: draining the
// stores (vmcnt) orders them before the counter, the counter form of the hand-off (guide section 6 G16) as in
// zgemm.hip.  No __threadfence(): an agent-scope release writes the whole L2 of the XCD back, once per
// block -- with 32 MiB of fresh dW / Whalf lines in it that was most of k_call_begin's time.
// The counters are left at zero for the next launch.
__device__ __forceinline__ bool last_block_of_launch(unsigned *t, int bid, int nb)
{
    const int g = bid >> 5, ng = (nb + 31) >> 5;
    const int gsize = (g == ng - 1) ? nb - (g << 5) : 32;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__hip_atomic_fetch_add(t + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(gsize - 1)) return false;
    __hip_atomic_store(t + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_fetch_add(t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(ng - 1)) return false;
    __hip_atomic_store(t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

// Entry of a stepper call in ONE launch (fused protocol, dW restarting from zero): row i of
//   dW = 0 (isospectral.py:430),  Whalf = W (:481-482 with dW = 0),  rowsum[i] = sum_j |W[i,j]|
// by block i (the reduction order of k_row_abs_sum), and the block that finishes last forms the
// automatic tolerance tol_factor * max_i rowsum[i] (:440-448; k_max_rows' NaN rule) and resets the
// control state -- what took a norm launch pair, a fill, a copy and k_state_init before.
__global__ __launch_bounds__(256) void k_call_begin(int N, const cplx *__restrict__ W, cplx *__restrict__ dW0,
                                                     cplx *__restrict__ Whalf, double *__restrict__ rowsum, unsigned *ticket,
                                                     qf_dev_state *state, qf_host_record *rec, double tol, int minit,
                                                     int maxit, int auto_tol, double tol_factor)
{
    __shared__ double part[4];
    __shared__ int last;
    // a block takes rows blockIdx.x, blockIdx.x + gridDim.x, ... (at most 1024 blocks: 32 ticket groups) and
    // keeps four of a thread's loads in flight; the sum of a row is formed in the order of k_row_abs_sum
    // (thread t adds columns t, t+256, ...; wave sums; (0+1)+(2+3))
    for (int i = blockIdx.x; i < N; i += gridDim.x) {
        double s = 0.0;
        for (int j0 = threadIdx.x; j0 < N; j0 += 4 * 256) {
            cplx z[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 256;
                z[u] = make_double2(0.0, 0.0);
                if (j < N) z[u] = W[(size_t)i * N + j];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u * 256;
                if (j < N) {
                    const size_t e = (size_t)i * N + j;
                    s += hypot(z[u].x, z[u].y);
                    dW0[e] = make_double2(0.0, 0.0);
                    Whalf[e] = z[u];
                }
            }
        }
        s = wave_sum(s);
        __syncthreads();      // (part[] of the previous row has been read)
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_store(rowsum + i, (part[0] + part[1]) + (part[2] + part[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) last = last_block_of_launch(ticket, blockIdx.x, gridDim.x) ? 1 : 0;
    __syncthreads();
    if (!last) return;
    double m = 0.0;
    bool nan = false;
    if (auto_tol) {
        for (int r = threadIdx.x; r < N; r += 256) {
            const double v = __hip_atomic_load(rowsum + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v != v) nan = true;
            else m = fmax(m, v);
        }
        const double isn = wave_max(nan ? 1.0 : 0.0);
        m = wave_max(m);
        if (isn > 0.0) m = __builtin_nan("");
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (auto_tol) {
            double r = 0.0;
            bool anynan = false;
            for (int w = 0; w < 4; ++w) {
                if (part[w] != part[w]) anynan = true;
                else r = fmax(r, part[w]);
            }
            tol = tol_factor * (anynan ? __builtin_nan("") : r);     // isospectral.py:446-448
        }
        qf_state_reset(state, rec, tol, minit, maxit);
    }
}

// partial[b] = sum over a fixed slice of Re(A conj(B)); then k_sum_partials folds them in order
__global__ __launch_bounds__(256) void k_inner_partial(size_t n, const cplx *__restrict__ A,
                                                        const cplx *__restrict__ B, double *__restrict__ partial)
{
    __shared__ double part[4];
    double s = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        cplx a = A[e], b = B[e];
        s += a.x * b.x + a.y * b.y;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

__global__ __launch_bounds__(64) void k_sum_partials(int n, const double *__restrict__ partial, double *__restrict__ out)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) s += partial[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = s;
}

// both inner products of the diagnostics in one pass (quflow/physics.py:26-38): partial sums exactly as
// k_inner_partial forms them, folded by the block that finishes last exactly as k_sum_partials does
__global__ __launch_bounds__(256) void k_inner2(size_t n, const cplx *__restrict__ A, const cplx *__restrict__ B,
                                                 double *__restrict__ partial, unsigned *ticket, double *__restrict__ out)
{
    __shared__ double pab[4], paa[4];
    __shared__ int last;
    double sab = 0.0, saa = 0.0;
    // four strides' loads in flight per thread, added in the order of the plain loop (same bits)
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t e0 = (size_t)blockIdx.x * 256 + threadIdx.x; e0 < n; e0 += 4 * stride) {
        cplx a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t e = e0 + u * stride;
            a[u] = b[u] = make_double2(0.0, 0.0);
            if (e < n) {
                a[u] = A[e];
                b[u] = B[e];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (e0 + u * stride < n) {
                sab += a[u].x * b[u].x + a[u].y * b[u].y;
                saa += a[u].x * a[u].x + a[u].y * a[u].y;
            }
        }
    }
    sab = wave_sum(sab);
    saa = wave_sum(saa);
    if ((threadIdx.x & 63) == 0) {
        pab[threadIdx.x >> 6] = sab;
        paa[threadIdx.x >> 6] = saa;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(partial + blockIdx.x, (pab[0] + pab[1]) + (pab[2] + pab[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(partial + 1024 + blockIdx.x, (paa[0] + paa[1]) + (paa[2] + paa[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = last_block_of_launch(ticket, blockIdx.x, gridDim.x) ? 1 : 0;
    }
    __syncthreads();
    if (!last || threadIdx.x >= 64) return;
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 64) {
        s0 += __hip_atomic_load(partial + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s1 += __hip_atomic_load(partial + 1024 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (threadIdx.x == 0) {
        out[0] = s0;
        out[1] = s1;
    }
}

}  // namespace

int qf_launch_call_begin(qf_ctx *ctx, double tol, int minit, int maxit, int auto_tol, double tol_factor)
{
    hipLaunchKernelGGL(k_call_begin, dim3(ctx->N < 1024 ? ctx->N : 1024), dim3(256), 0, ctx->stream, ctx->N, ctx->W, ctx->dW[0], ctx->Whalf, ctx->rowsum,
                       ctx->ticket + 600, ctx->state, ctx->host_rec, tol, minit, maxit, auto_tol, tol_factor);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_inner2(qf_ctx *ctx, const cplx *A, const cplx *B, double *out_dev)
{
    const size_t n = (size_t)ctx->N * ctx->N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_inner2, dim3(blocks), dim3(256), 0, ctx->stream, n, A, B, ctx->scalars + 64, ctx->ticket + 640, out_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_update(qf_ctx *ctx, const cplx *PW, cplx *W, const cplx *dW_a, const cplx *dW_b, cplx *Whalf,
                     cplx *kahan_c, int reinitialize, qf_guard guard)
{
    const int N = ctx->N;
    dim3 grid((N + TU - 1) / TU, (N + TU - 1) / TU), block(256);
    cplx *a = const_cast<cplx *>(dW_a), *b = const_cast<cplx *>(dW_b);
    if (kahan_c)
        hipLaunchKernelGGL(k_update<true>, grid, block, 0, ctx->stream, N, PW, W, a, b, Whalf, kahan_c, reinitialize, guard,
                           guard.state ? ctx->state : nullptr, ctx->host_rec, ctx->ticket);
    else
        hipLaunchKernelGGL(k_update<false>, grid, block, 0, ctx->stream, N, PW, W, a, b, Whalf, kahan_c, reinitialize, guard,
                           guard.state ? ctx->state : nullptr, ctx->host_rec, ctx->ticket);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_norm_decide(qf_ctx *ctx, const double *rowpart, int tiles, qf_guard guard)
{
    hipLaunchKernelGGL(k_norm_decide, dim3(1), dim3(1024), 0, ctx->stream, ctx->N, tiles, rowpart, ctx->state, guard);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_state_init(qf_ctx *ctx, double tol, int minit, int maxit, const double *norm_dev, double tol_factor)
{
    hipLaunchKernelGGL(k_state_init, dim3(1), dim3(64), 0, ctx->stream, ctx->state, ctx->host_rec, tol, minit, maxit,
                       norm_dev, tol_factor);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_norm_from_rowpart(qf_ctx *ctx, const double *rowpart, int tiles, double *out_dev)
{
    hipLaunchKernelGGL(k_max_rows, dim3(1), dim3(1024), 0, ctx->stream, ctx->N, tiles, rowpart, out_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_norm_inf(qf_ctx *ctx, const cplx *A, double *out_dev)
{
    hipLaunchKernelGGL(k_row_abs_sum, dim3(ctx->N), dim3(256), 0, ctx->stream, ctx->N, A, ctx->rowsum);
    QF_HIP(hipGetLastError());
    return qf_launch_norm_from_rowpart(ctx, ctx->rowsum, 1, out_dev);
}

int qf_launch_erk_stage(qf_ctx *ctx, const cplx *A, const cplx *B, double inv_hb, const cplx *W, cplx *acc,
                        double c_acc, cplx *Wp, double c_wp, cplx *Wout, double c_fin)
{
    const int N = ctx->N;
    dim3 grid((N + TU - 1) / TU, (N + TU - 1) / TU), block(256);
    if (B) hipLaunchKernelGGL(k_erk_stage<false>, grid, block, 0, ctx->stream, N, A, B, inv_hb, W, acc, c_acc, Wp, c_wp, Wout, c_fin);
    else hipLaunchKernelGGL(k_erk_stage<true>, grid, block, 0, ctx->stream, N, A, B, inv_hb, W, acc, c_acc, Wp, c_wp, Wout, c_fin);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_magnetic_fix(qf_ctx *ctx, const cplx *BTP, const cplx *BT, cplx *dW, const cplx *dW_old, const cplx *W,
                           cplx *Whalf, double *rowpart)
{
    const int N = ctx->N;
    dim3 grid((N + TU - 1) / TU, (N + TU - 1) / TU), block(256);
    hipLaunchKernelGGL(k_magnetic_fix, grid, block, 0, ctx->stream, N, BTP, BT, dW, dW_old, W, Whalf, rowpart);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_magnetic_update(qf_ctx *ctx, const cplx *BT, cplx *W, const cplx *dW, cplx *Whalf)
{
    const int N = ctx->N;
    dim3 grid((N + TU - 1) / TU, (N + TU - 1) / TU), block(256);
    hipLaunchKernelGGL(k_magnetic_update, grid, block, 0, ctx->stream, N, BT, W, dW, Whalf);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_lincomb(qf_ctx *ctx, double a, const cplx *X, double b, const cplx *Y, double c, cplx *out)
{
    const size_t n = (size_t)ctx->N * ctx->N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_lincomb, dim3(blocks), dim3(256), 0, ctx->stream, ctx->N, a, X, b, Y, c, out);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_mirror_lower(qf_ctx *ctx, cplx *X)
{
    const int tiles = (ctx->N + TU - 1) / TU;
    hipLaunchKernelGGL(k_mirror_lower, dim3(tiles, tiles), dim3(256), 0, ctx->stream, ctx->N, X);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_neg_conj_transpose(qf_ctx *ctx, const cplx *X, cplx *out)
{
    const int N = ctx->N;
    dim3 grid((N + TU - 1) / TU, (N + TU - 1) / TU), block(256);
    hipLaunchKernelGGL(k_neg_conj_transpose, grid, block, 0, ctx->stream, N, X, out);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_sum_rowpart(qf_ctx *ctx, const double *rowpart, int tiles, double *rowsum_dev)
{
    hipLaunchKernelGGL(k_sum_rowpart, dim3((ctx->N + 255) / 256), dim3(256), 0, ctx->stream, ctx->N, tiles, rowpart,
                       rowsum_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_skew_defect(qf_ctx *ctx, const cplx *A, double *out_dev)
{
    const int N = ctx->N;
    int blocks = N < 1024 ? N : 1024;
    double *partial = ctx->scalars + 64;   // [2][blocks]
    hipLaunchKernelGGL(k_skew_defect_partial, dim3(blocks), dim3(256), 0, ctx->stream, N, A, partial, partial + 1024);
    QF_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_max_partials2, dim3(1), dim3(64), 0, ctx->stream, blocks, partial, partial + 1024, out_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_inner(qf_ctx *ctx, const cplx *A, const cplx *B, double *out_dev)
{
    const size_t n = (size_t)ctx->N * ctx->N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    double *partial = ctx->scalars + 64;  // scalars[0..63] are result slots
    hipLaunchKernelGGL(k_inner_partial, dim3(blocks), dim3(256), 0, ctx->stream, n, A, B, partial);
    QF_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, ctx->stream, blocks, partial, out_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

// tests only (qf_debug_modulus): qf_modulus beside the compiler's sqrt on the same arguments
__global__ void k_debug_modulus(int n, const double *__restrict__ er, const double *__restrict__ ei, double *__restrict__ out_mod,
                                double *__restrict__ out_sqrt)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out_mod[i] = qf_modulus(er[i], ei[i]);
    out_sqrt[i] = sqrt(er[i] * er[i] + ei[i] * ei[i]);
}

int qf_launch_debug_modulus(qf_ctx *ctx, int n, const double *er, const double *ei, double *out_mod, double *out_sqrt)
{
    hipLaunchKernelGGL(k_debug_modulus, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, n, er, ei, out_mod, out_sqrt);
    QF_HIP(hipGetLastError());
    return QF_OK;
}
