// Device-side end of an iteration of the fused protocol, shared by the
// second-product kernels of zgemm.hip (fp64 matrix cores) and ozaki.hip (int8 matrix cores).
#pragma once

#include "qf_internal.h"

// Fused step end, executed by the last finishing workgroup of the second product (k_zgemm_tri or
// k_zgemm<.., FUSED>; 256 threads): the
// residual norm of this iteration from the per-tile row sums (isospectral.py:526-534), the exit
// test (isospectral.py:535-536), and -- if the step is over -- the step advance that the separate
// update kernel used to do: flip the W pair, select the prepared Whalf, count, publish.
// rowpart was stored write-through by the finishers and is read with sc1 loads (never through
// this CU's L1); sums run over the column tiles in a fixed order (deterministic).
// (scratch: 8 doubles of the kernel's DYNAMIC LDS -- a static __shared__ here would shift the
// dynamic base off its 16-byte alignment and slow every ds_read_b128 of the K loop, guide G17)
// (ROWS: rows per lane and round trip -- 2 * 16 * ROWS registers hold the loads in flight: 4 where the product's own
// accumulators already need that many registers, 1 in the complex64 kernels, whose occupancy the finale must not cut)
template <int ROWS = 4>
__device__ inline void qf_fused_step_end(int N, int slots, const double *rowpart, unsigned *ticket, qf_dev_state *state,
                                  qf_host_record *rec, int g_iter, int tid, double *scratch)
{
    double *part = scratch;
    int *nanflag = reinterpret_cast<int *>(scratch + 4);
    const bool check = (g_iter + 1 >= state->minit);
    double mx = 0.0;
    int nan = 0;
    if (check) {
        // sc1 loads, 16 * ROWS in flight per lane (ROWS rows x 16 column tiles per round): a relaxed atomic
        // load per element would be waited for one by one (measured: +21 us at N=1024, +105 us at
        // N=2048).  This runs after the segment loop, when nothing else is live.
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const __amdgpu_buffer_rsrc_t rs_rp = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(rowpart), 0, (int)((size_t)slots * N * sizeof(double)), 0x00020000);
        const unsigned slot_bytes = (unsigned)((size_t)N * sizeof(double));
        for (int ib = tid; ib < N; ib += ROWS * 256) {
            double sum[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) sum[r] = 0.0;
            for (int t0 = 0; t0 < slots; t0 += 16) {
                v2u v[ROWS][16];
#pragma unroll
                for (int r = 0; r < ROWS; ++r) {
                    // rows past the end re-read row `ib` and are dropped below; same for slots
                    const unsigned vo = (unsigned)(((ib + 256 * r < N) ? ib + 256 * r : ib) * sizeof(double));
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const unsigned so = (t0 + t < slots) ? (unsigned)(t0 + t) * slot_bytes : 0u;
                        v[r][t] = __builtin_amdgcn_raw_buffer_load_b64(rs_rp, vo, so, 16);
                    }
                }
#pragma unroll
                for (int r = 0; r < ROWS; ++r)
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (t0 + t < slots) sum[r] += *reinterpret_cast<const double *>(&v[r][t]);
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r)
                if (ib + 256 * r < N) {
                    if (sum[r] != sum[r]) nan = 1; else mx = fmax(mx, sum[r]);
                }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mx = fmax(mx, __shfl_xor(mx, off, 64));
            nan |= __shfl_xor(nan, off, 64);
        }
        if ((tid & 63) == 0) {
            part[tid >> 6] = mx;
            nanflag[tid >> 6] = nan;
        }
    }
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        const int iters = g_iter + 1;
        state->total_iterations += 1;                       // isospectral.py:478
        state->dw_parity ^= 1;                              // this product wrote the other dW buffer
        bool done = false, aborted = false;
        if (check) {
            double r = fmax(fmax(part[0], part[1]), fmax(part[2], part[3]));
            if (nanflag[0] | nanflag[1] | nanflag[2] | nanflag[3]) r = __builtin_nan("");
            const double resnorm_old = state->resnorm;      // isospectral.py:525
            state->resnorm = r;
            // scipy.linalg.norm(dW_old, ord=inf) checks its argument: the reference raises ValueError here (isospectral.py:534)
            if (!QF_FINITE(r)) aborted = true;
            if (r <= state->tol || r >= resnorm_old) done = true;   // isospectral.py:535-536
        }
        if (aborted) {
            // close the CALL: W stays what the last completed step left (no flip), nothing queued behind this launch is due
            rec->nonfinite = 1;
            rec->last_step_iters = iters;
            rec->resnorm = state->resnorm;
            state->step_index = QF_STEP_ABORTED;
            state->iters_this_step = 0;
            state->wh_sel = 0;
        } else if (done || iters >= state->maxit) {
            if (!done) state->number_of_maxit += 1;         // for-else, isospectral.py:538-540
            rec->last_step_iters = iters;
            rec->resnorm = state->resnorm;
            state->step_index += 1;
            state->iters_this_step = 0;
            state->resnorm = __builtin_inf();               // isospectral.py:470
            state->w_parity ^= 1;                           // W += 2 (PW - PW^H): the candidate becomes the state
            state->wh_sel = 1;                              // next iteration: Whalf = W_next + dW
        } else {
            state->iters_this_step = iters;
            state->wh_sel = 0;
        }
        rec->total_iterations = state->total_iterations;
        rec->number_of_maxit = state->number_of_maxit;
        rec->step_index = state->step_index;
        rec->w_parity = state->w_parity;
        rec->wh_sel = state->wh_sel;
        rec->dw_parity = state->dw_parity;
        const unsigned long long prog = ((unsigned long long)(unsigned)state->step_index << 32) |
                                        (unsigned long long)(unsigned)state->iters_this_step;
        __hip_atomic_store(&rec->progress, prog, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}



// ---- deferred step end.  The second product leaves only its row sums and qf_dev_state::pending; the
// next launch's workgroups EACH form the residual norm and the new control state from them (the same fixed summation
// order as qf_fused_step_end: identical bits, identical decisions), act on it at once, and the workgroup whose ticket
// came last -- every other one has read the old state by then -- writes it back and publishes the progress.
struct qf_new_state {
    int step_index, iters_this_step, wh_sel, w_parity, dw_parity;
    int closed, last_step_iters, hit_maxit;
    int nonfinite;          // the residual this decision looked at is inf / NaN (the reference raises there)
    double resnorm;         // the control state's residual after the decision (inf at a step's start)
    double last_resnorm;    // the residual this decision looked at
};

// every thread of the block (blockDim.x a multiple of 64, <= 1024); scratch: >= 40 doubles of LDS, free on return
__device__ inline qf_new_state qf_decide_compute(int N, int slots, const double *__restrict__ rowpart,
                                                 const qf_dev_state *st, double *scratch)
{
    const int tid = threadIdx.x, nth = blockDim.x, nw = nth >> 6;
    double *part = scratch;
    int *nanflag = reinterpret_cast<int *>(scratch + 20);
    const int iters = st->pending_iter + 1;
    const int minit = st->minit, maxit = st->maxit;
    const double tol = st->tol, resnorm_old = st->resnorm;
    const int step_index = st->step_index, w_parity = st->w_parity, dw_parity = st->dw_parity;
    const bool check = iters >= minit;
    double r = 0.0;
    // (the row sums are read whether or not this iteration is checked -- below minit their maximum is not used:
    // their addresses do not depend on the control state, so the two round trips overlap instead of following
    // each other)
    {
        double mx = 0.0;
        int nan = 0;
        // two rows x sixteen column slots per trip, all 32 loads in flight together (a loop of dependent
        // load-and-add round trips here cost the deciding launch ~8 us at N = 512); the sums run over the slots in
        // order, as in qf_fused_step_end.  (slots <= 16: the deferral stops at N = 512)
        for (int i0 = tid; i0 < N; i0 += 2 * nth) {
            double v[2][16];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = (i0 + rr * nth < N) ? i0 + rr * nth : i0;
#pragma unroll
                for (int t = 0; t < 16; ++t) v[rr][t] = rowpart[(size_t)(t < slots ? t : 0) * N + row];
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                double sum = 0.0;
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    if (t < slots) sum += v[rr][t];
                if (i0 + rr * nth < N) {
                    if (sum != sum) nan = 1; else mx = fmax(mx, sum);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mx = fmax(mx, __shfl_xor(mx, off, 64));
            nan |= __shfl_xor(nan, off, 64);
        }
        if ((tid & 63) == 0) {
            part[tid >> 6] = mx;
            nanflag[tid >> 6] = nan;
        }
        __syncthreads();
        int anynan = 0;
        for (int w = 0; w < nw; ++w) {
            r = fmax(r, part[w]);
            anynan |= nanflag[w];
        }
        if (anynan) r = __builtin_nan("");
        __syncthreads();
    }
    qf_new_state ns;
    ns.dw_parity = dw_parity ^ 1;                       // the product wrote the other dW buffer
    ns.last_resnorm = check ? r : resnorm_old;
    ns.nonfinite = (check && !QF_FINITE(r)) ? 1 : 0;
    bool done = false;
    if (check && (r <= tol || r >= resnorm_old)) done = true;       // isospectral.py:535-536
    ns.hit_maxit = 0;
    if (ns.nonfinite) {
        // the reference raises here (isospectral.py:534): close the CALL -- no flip of W, no launch behind this one is due
        ns.closed = 0;
        ns.last_step_iters = iters;
        ns.step_index = QF_STEP_ABORTED;
        ns.iters_this_step = 0;
        ns.resnorm = r;
        ns.w_parity = w_parity;
        ns.wh_sel = 0;
    } else if (done || iters >= maxit) {
        ns.hit_maxit = done ? 0 : 1;                    // for-else, isospectral.py:538-540
        ns.closed = 1;
        ns.last_step_iters = iters;
        ns.step_index = step_index + 1;
        ns.iters_this_step = 0;
        ns.resnorm = __builtin_inf();                   // isospectral.py:470
        ns.w_parity = w_parity ^ 1;                     // W += 2 (PW - PW^H): the candidate becomes the state
        ns.wh_sel = 1;
    } else {
        ns.closed = 0;
        ns.last_step_iters = 0;
        ns.step_index = step_index;
        ns.iters_this_step = iters;
        ns.resnorm = check ? r : resnorm_old;
        ns.w_parity = w_parity;
        ns.wh_sel = 0;
    }
    return ns;
}

// one thread of the last-arriving workgroup
__device__ inline void qf_decide_apply(qf_dev_state *state, qf_host_record *rec, unsigned *ticket, const qf_new_state &ns)
{
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next deciding launch
    state->total_iterations += 1;                       // isospectral.py:478
    state->dw_parity = ns.dw_parity;
    if (ns.closed || ns.nonfinite) {
        state->number_of_maxit += ns.hit_maxit;
        rec->last_step_iters = ns.last_step_iters;
        rec->resnorm = ns.last_resnorm;
    }
    state->step_index = ns.step_index;
    state->iters_this_step = ns.iters_this_step;
    state->resnorm = ns.resnorm;
    state->w_parity = ns.w_parity;
    state->wh_sel = ns.wh_sel;
    state->pending = 0;
    if (ns.nonfinite) rec->nonfinite = 1;
    rec->total_iterations = state->total_iterations;
    rec->number_of_maxit = state->number_of_maxit;
    rec->step_index = state->step_index;
    rec->w_parity = state->w_parity;
    rec->wh_sel = state->wh_sel;
    rec->dw_parity = state->dw_parity;
    const unsigned long long prog = ((unsigned long long)(unsigned)state->step_index << 32) |
                                    (unsigned long long)(unsigned)state->iters_this_step;
    __hip_atomic_store(&rec->progress, prog, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
