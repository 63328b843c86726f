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
__device__ inline void qf_fused_step_end(int N, int slots, const double *rowpart, unsigned *ticket, qf_dev_state *state,
                                  qf_host_record *rec, int g_iter, int tid, double *scratch)
{
    double *part = scratch;
    int *nanflag = reinterpret_cast<int *>(scratch + 4);
    const bool check = (g_iter + 1 >= state->minit);
    double mx = 0.0;
    int nan = 0;
    if (check) {
        // sc1 loads, 64 in flight per lane (4 rows x 16 column tiles per round): a relaxed atomic
        // load per element would be waited for one by one (measured: +21 us at N=1024, +105 us at
        // N=2048).  This runs after the segment loop, when nothing else is live.
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const __amdgpu_buffer_rsrc_t rs_rp = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<double *>(rowpart), 0, (int)((size_t)slots * N * sizeof(double)), 0x00020000);
        const unsigned slot_bytes = (unsigned)((size_t)N * sizeof(double));
        for (int ib = tid; ib < N; ib += 4 * 256) {
            double sum[4] = {0.0, 0.0, 0.0, 0.0};
            for (int t0 = 0; t0 < slots; t0 += 16) {
                v2u v[4][16];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // rows past the end re-read row `ib` and are dropped below; same for slots
                    const unsigned vo = (unsigned)(((ib + 256 * r < N) ? ib + 256 * r : ib) * sizeof(double));
#pragma unroll
                    for (int t = 0; t < 16; ++t) {
                        const unsigned so = (t0 + t < slots) ? (unsigned)(t0 + t) * slot_bytes : 0u;
                        v[r][t] = __builtin_amdgcn_raw_buffer_load_b64(rs_rp, vo, so, 16);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (t0 + t < slots) sum[r] += *reinterpret_cast<const double *>(&v[r][t]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
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
        bool done = false;
        if (check) {
            double r = fmax(fmax(part[0], part[1]), fmax(part[2], part[3]));
            if (nanflag[0] | nanflag[1] | nanflag[2] | nanflag[3]) r = __builtin_nan("");
            const double resnorm_old = state->resnorm;      // isospectral.py:525
            state->resnorm = r;
            if (r <= state->tol || r >= resnorm_old) done = true;   // isospectral.py:535-536
        }
        if (done || iters >= state->maxit) {
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

