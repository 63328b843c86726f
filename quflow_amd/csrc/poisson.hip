// Per-diagonal tridiagonal Laplacian kernels for gfx950.
//
// Reference semantics: quflow/laplacian/cpu.py
//   _compute_cpu_laplacian  :55-95    coefficient table lap[i,j,{0,1}]
//   _solve_cpu_skewh        :281-362  Thomas solve of the N upper diagonals + mirror
//   _solve_cpu_nonskewh     :200-278  Thomas solve of all 2N-1 diagonals
//   _dot_cpu_generic        :98-108   3-point stencil along diagonals
//
// Design (MI355X-first, not a translation of the numba loops)
// ---------------------------------------------------------
// * Flat walk.  Entry (i,j) has flat index e = i*N + j; stepping along a diagonal is
//   e += N+1.  Starting from e = t (t = 0..N) the walk covers upper diagonal t and then
//   wraps into lower diagonal -(N+1-t); the coupling coefficient a_0 at the head of every
//   diagonal is 0, so chained diagonals decouple.  At a fixed step k the entries of
//   consecutive walks t, t+1, ... are CONSECUTIVE in memory: every load/store of the
//   sweeps is a unit-stride, fully coalesced access with no gather.
// * The matrix T_m is data independent, so its LU factors are computed once per (N, table)
//   by k_build_factors (multipliers w_k = a_k/b'_{k-1}, reciprocal pivots 1/b'_k).  A solve
//   is then two first-order linear recurrences
//        forward   y_k = f_k - w_k y_{k-1}
//        backward  p_k = y_k/b'_k - w_{k+1} p_{k+1}
//   i.e. one dependent FMA per step instead of the reference's divide chain.
// * Depth reduction.  Each walk is cut into chunks of L steps; a thread owns one chunk in
//   registers.  Pass 1 sweeps every chunk with a zero carry-in (all chunks of all diagonals
//   in parallel), pass 2 propagates the chunk carries through LDS (N/L sequential steps),
//   pass 3 adds carry * prod(-w) to every entry.  Sequential depth drops from 2N to about
//   2(2L + N/L) dependent FMAs; HBM/L2 traffic is the algorithmic minimum (W once, the two
//   factor tables once, P once).
// * m = 0: tr(W)/N is removed from the right-hand side and tr(P)/N from the solution
//   (cpu.py:311-317,342-352) with deterministic block reductions.
#include "qf_internal.h"

#pragma clang fp contract(off)  // table arithmetic mirrors the reference's op order; FMAs are explicit

namespace {

__device__ __forceinline__ double lap_b(int N, int i, int j)
{
    // cpu.py:82  -((N-1)(2k+1+|m|) - 2k(k+|m|)),  k = min(i,j), |m| = |j-i|
    long long k = i < j ? i : j;
    long long am = i < j ? j - i : i - j;
    long long NN = N;
    return -(double)((NN - 1) * (2 * k + 1 + am) - 2 * k * (k + am));
}

__device__ __forceinline__ double lap_a(int N, int i, int j)
{
    // cpu.py:83  sqrt(((k+|m|)(N-k-|m|)) (k(N-k)))  (exact integer under the root for N <= 8192)
    long long k = i < j ? i : j;
    long long am = i < j ? j - i : i - j;
    long long NN = N;
    return sqrt((double)(((k + am) * (NN - k - am)) * (k * (NN - k))));
}

__global__ void k_lap_table(int N, int bc, double *__restrict__ lap)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)N * N) return;
    int i = (int)(e / N), j = (int)(e % N);
    double b = lap_b(N, i, j);
    if (bc && e == 0) b -= 0.5;  // cpu.py:90
    lap[2 * e] = b;
    lap[2 * e + 1] = lap_a(N, i, j);
}

// One thread per flat walk t = 0..N; sequential (runs once per table).
// Same operation order as cpu.py:309,324-325: w = a/b'_{k-1}; b'_k = b_k - w a_k.
__global__ void k_build_factors(int N, const double *__restrict__ lap, double *__restrict__ wtab,
                                double *__restrict__ invtab)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > N) return;
    size_t NN = (size_t)N * N;
    double bp_prev = 1.0;
    for (size_t e = t; e < NN; e += (size_t)N + 1) {
        int i = (int)(e / N), j = (int)(e % N);
        double b = lap[2 * e], a = lap[2 * e + 1];
        double w, bp;
        if (i == 0 || j == 0) {  // head of a diagonal: the reference starts its sweep at k = 1
            w = 0.0;
            bp = b;
        } else {
            w = a / bp_prev;
            bp = b - w * a;
        }
        wtab[e] = w;
        invtab[e] = 1.0 / bp;
        bp_prev = bp;
    }
}

// _dot_cpu_generic, cpu.py:98-108 (coefficients recomputed on the fly: no table traffic)
__global__ void k_laplace(int N, const cplx *__restrict__ P, cplx *__restrict__ W)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= N) return;
    size_t e = (size_t)i * N + j;
    double b = lap_b(N, i, j);
    cplx p = P[e];
    double wr = b * p.x, wi = b * p.y;
    if (i < N - 1 && j < N - 1) {
        double a = lap_a(N, i + 1, j + 1);
        cplx q = P[e + N + 1];
        wr += a * q.x;
        wi += a * q.y;
    }
    if (i > 0 && j > 0) {
        double a = lap_a(N, i, j);
        cplx q = P[e - N - 1];
        wr += a * q.x;
        wi += a * q.y;
    }
    W[e] = make_double2(wr, wi);
}

// deterministic block-wide sum of a complex value (fixed tree in LDS)
__device__ __forceinline__ cplx block_sum(cplx v, cplx *red, int tid, int nthreads)
{
    red[tid] = v;
    __syncthreads();
    // nthreads is a multiple of 64 but not necessarily a power of two
    int n = nthreads;
    while (n > 1) {
        int half = (n + 1) >> 1;
        if (tid < n - half) {
            red[tid].x += red[tid + half].x;
            red[tid].y += red[tid + half].y;
        }
        __syncthreads();
        n = half;
    }
    cplx r = red[0];
    __syncthreads();
    return r;
}

// Chunked two-level Thomas solve.  Block = G walks x C chunks (G*C threads, lane-fastest in g).
//   SKEWH = 1: walks t = 0..N-1 restricted to the upper triangle (length N-t), result
//              mirrored as P[j,i] = -conj(P[i,j])           (cpu.py:281-362)
//   SKEWH = 0: walks t = 0..N over the whole matrix          (cpu.py:200-278)
template <int L, int SKEWH>
__global__ __launch_bounds__(L == 16 ? 512 : 256) void k_solve(int N, int G, int C, const cplx *__restrict__ W, cplx *__restrict__ P,
                        const double *__restrict__ wtab, const double *__restrict__ invtab, double scale,
                        qf_guard guard)
{
    if (!qf_guard_iter(guard)) return;   // tagged stepper launch that is not due: no-op
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;  // = G*C rounded up to a multiple of 64
    const int g = tid % G;
    const int jc = tid / G;           // chunk index (>= C for padding threads)
    const int t = blockIdx.x * G + g;
    const int T = SKEWH ? N : N + 1;
    const size_t NN = (size_t)N * N;
    const size_t stride = (size_t)N + 1;

    // LDS carve-up: endv[C*G] complex, endc[C*G] real, carry[C*G] complex, red[nthreads] complex
    cplx *endv = reinterpret_cast<cplx *>(smem_raw);
    cplx *carry = endv + (size_t)C * G;
    cplx *red = carry + (size_t)C * G;
    double *endc = reinterpret_cast<double *>(red + nthreads);

    int len = 0;
    if (t < T && jc < C) len = SKEWH ? (N - t) : (int)((NN - 1 - (size_t)t) / stride) + 1;
    const bool has_trace = (blockIdx.x == 0);  // the block that owns walk t = 0 (m = 0)
    const bool on_diag = (t == 0 && jc < C);

    // ---- m = 0: circulation tr(W)/N, cpu.py:311-317
    cplx trW = make_double2(0.0, 0.0);
    if (has_trace) {
        cplx s = make_double2(0.0, 0.0);
        for (int k = tid; k < N; k += nthreads) {
            cplx d = W[(size_t)k * stride];
            s.x += d.x;
            s.y += d.y;
        }
        s = block_sum(s, red, tid, nthreads);
        double invN = 1.0 / (double)N;
        trW = make_double2(s.x * invN, s.y * invN);
    }

    const int k0 = jc * L;
    const size_t e0 = (size_t)t + (size_t)k0 * stride;

    cplx v[L];
    double w[L + 1];

    // ---- pass 1: local forward sweep with zero carry-in
    {
        cplx yprev = make_double2(0.0, 0.0);
        double cprod = 1.0;
#pragma unroll
        for (int s = 0; s < L; ++s) {
            const bool valid = (k0 + s) < len;
            const size_t e = e0 + (size_t)s * stride;
            cplx f = make_double2(0.0, 0.0);
            double ws = 0.0;
            if (valid) {
                f = W[e];
                ws = wtab[e];
                if (on_diag) {
                    f.x -= trW.x;
                    f.y -= trW.y;
                }
            }
            cplx y;
            y.x = __fma_rn(-ws, yprev.x, f.x);
            y.y = __fma_rn(-ws, yprev.y, f.y);
            v[s] = y;
            w[s] = ws;
            cprod *= -ws;
            yprev = y;
        }
        // multiplier that links this chunk's last entry to the next chunk (backward sweep)
        {
            const bool valid = (k0 + L) < len;
            w[L] = valid ? wtab[e0 + (size_t)L * stride] : 0.0;
        }
        if (jc < C) {
            endv[jc * G + g] = yprev;
            endc[jc * G + g] = cprod;
        }
    }
    __syncthreads();

    // ---- pass 2: chunk carries of the forward recurrence (sequential over C, per walk)
    if (tid < G) {
        cplx c = make_double2(0.0, 0.0);
        for (int q = 0; q < C; ++q) {
            carry[q * G + tid] = c;
            cplx ev = endv[q * G + tid];
            double ec = endc[q * G + tid];
            c.x = __fma_rn(ec, c.x, ev.x);
            c.y = __fma_rn(ec, c.y, ev.y);
        }
    }
    __syncthreads();

    // ---- pass 3: apply the carry, normalise by the pivot:  c_k = y_k / b'_k
    {
        cplx corr = (jc < C) ? carry[jc * G + g] : make_double2(0.0, 0.0);
#pragma unroll
        for (int s = 0; s < L; ++s) {
            const bool valid = (k0 + s) < len;
            const size_t e = e0 + (size_t)s * stride;
            double inv = valid ? invtab[e] : 0.0;
            corr.x *= -w[s];
            corr.y *= -w[s];
            v[s].x = (v[s].x + corr.x) * inv;
            v[s].y = (v[s].y + corr.y) * inv;
        }
    }
    __syncthreads();  // carry[] / endv[] are reused below

    // ---- pass 4: local backward sweep with zero carry-in:  p_k = c_k - w_{k+1} p_{k+1}
    {
        cplx pnext = make_double2(0.0, 0.0);
        double dprod = 1.0;
#pragma unroll
        for (int s = L - 1; s >= 0; --s) {
            cplx p;
            p.x = __fma_rn(-w[s + 1], pnext.x, v[s].x);
            p.y = __fma_rn(-w[s + 1], pnext.y, v[s].y);
            v[s] = p;
            dprod *= -w[s + 1];
            pnext = p;
        }
        if (jc < C) {
            endv[jc * G + g] = pnext;
            endc[jc * G + g] = dprod;
        }
    }
    __syncthreads();

    // ---- pass 5: chunk carries of the backward recurrence
    if (tid < G) {
        cplx c = make_double2(0.0, 0.0);
        for (int q = C - 1; q >= 0; --q) {
            carry[q * G + tid] = c;
            cplx ev = endv[q * G + tid];
            double ec = endc[q * G + tid];
            c.x = __fma_rn(ec, c.x, ev.x);
            c.y = __fma_rn(ec, c.y, ev.y);
        }
    }
    __syncthreads();

    // ---- pass 6: apply the carry
    {
        cplx corr = (jc < C) ? carry[jc * G + g] : make_double2(0.0, 0.0);
#pragma unroll
        for (int s = L - 1; s >= 0; --s) {
            corr.x *= -w[s + 1];
            corr.y *= -w[s + 1];
            v[s].x += corr.x;
            v[s].y += corr.y;
        }
    }

    // ---- m = 0: remove tr(P)/N, cpu.py:342-352
    if (has_trace) {
        cplx s = make_double2(0.0, 0.0);
        if (on_diag) {
#pragma unroll
            for (int q = 0; q < L; ++q) {
                if ((k0 + q) < len) {
                    s.x += v[q].x;
                    s.y += v[q].y;
                }
            }
        }
        s = block_sum(s, red, tid, nthreads);
        if (on_diag) {
            double invN = 1.0 / (double)N;
            double tx = s.x * invN, ty = s.y * invN;
#pragma unroll
            for (int q = 0; q < L; ++q) {
                v[q].x -= tx;
                v[q].y -= ty;
            }
        }
    }

    // ---- store (scaled), and mirror for the skew-Hermitian solve
#pragma unroll
    for (int s = 0; s < L; ++s) {
        const int k = k0 + s;
        if (k < len) {
            const size_t e = e0 + (size_t)s * stride;
            cplx p = make_double2(v[s].x * scale, v[s].y * scale);
            P[e] = p;
            if (SKEWH && t != 0) {
                // (i,j) = (k, k+t)  ->  P[j,i] = -conj(P[i,j]), cpu.py:334,340
                P[(size_t)(k + t) * N + k] = make_double2(-p.x, p.y);
            }
        }
    }
}

struct solve_cfg {
    int L, G, C, threads;
    size_t smem;
};

solve_cfg pick_cfg(int N)
{
    solve_cfg c;
    c.L = 16;
    c.C = (N + c.L - 1) / c.L;
    if (c.C > 128) {  // N > 2048: longer chunks keep the block within its thread budget
        c.L = 32;
        c.C = (N + c.L - 1) / c.L;
    }
    const int max_threads = c.L == 16 ? 512 : 256;  // register budget of k_solve<L>
    int G = 64;
    while (G > 1 && G * c.C > max_threads) G >>= 1;
    // prefer enough blocks to spread over the CUs
    while (G > 8 && (N + G - 1) / G < 64) G >>= 1;
    c.G = G;
    c.threads = ((G * c.C + 63) / 64) * 64;
    c.smem = (size_t)c.C * G * (16 + 16 + 8) + (size_t)c.threads * 16;
    return c;
}

}  // namespace

int qf_launch_lap_table(qf_ctx *ctx, int bc, double *lap_dev)
{
    size_t NN = (size_t)ctx->N * ctx->N;
    int threads = 256;
    unsigned blocks = (unsigned)((NN + threads - 1) / threads);
    hipLaunchKernelGGL(k_lap_table, dim3(blocks), dim3(threads), 0, ctx->stream, ctx->N, bc, lap_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_build_factors(qf_ctx *ctx, const double *lap_dev, qf_factors f)
{
    int threads = 64;
    unsigned blocks = (unsigned)((ctx->N + 1 + threads - 1) / threads);
    hipLaunchKernelGGL(k_build_factors, dim3(blocks), dim3(threads), 0, ctx->stream, ctx->N, lap_dev,
                       f.wtab, f.invtab);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_solve(qf_ctx *ctx, const qf_factors &f, const cplx *W, cplx *P, double scale, int skewh,
                    qf_guard guard)
{
    const int N = ctx->N;
    solve_cfg c = pick_cfg(N);
    if (c.G * c.C > (c.L == 16 ? 512 : 256)) {
        qf_set_error("qf_launch_solve: N=%d too large for the chunked solver", N);
        return QF_ERR_INVALID;
    }
    const int T = skewh ? N : N + 1;
    unsigned blocks = (unsigned)((T + c.G - 1) / c.G);
    dim3 grid(blocks), block(c.threads);
#define QF_SOLVE(LL, SK)                                                                            \
    hipLaunchKernelGGL((k_solve<LL, SK>), grid, block, c.smem, ctx->stream, N, c.G, c.C, W, P, f.wtab, \
                       f.invtab, scale, guard)
    if (c.L == 16) {
        if (skewh) QF_SOLVE(16, 1); else QF_SOLVE(16, 0);
    } else {
        if (skewh) QF_SOLVE(32, 1); else QF_SOLVE(32, 0);
    }
#undef QF_SOLVE
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_laplace(qf_ctx *ctx, const cplx *P, cplx *W)
{
    const int N = ctx->N;
    dim3 block(256), grid((N + 255) / 256, N);
    hipLaunchKernelGGL(k_laplace, grid, block, 0, ctx->stream, N, P, W);
    QF_HIP(hipGetLastError());
    return QF_OK;
}
