// Complex128 N x N x N product on the INT8 matrix cores (v_mfma_i32_32x32x32_i8) by digit
// splitting ("Ozaki scheme"), for the two products of the isospectral iteration
// (quflow/integrators/isospectral.py:496,499) -- BASELINE.json config 3's "low-precision-MFMA
// commutator with fp64 Laplacian", built on the int8 rather than the bf16 pipe:
//   * the i8 MFMA runs at twice the bf16 rate (32 cycles for 32x32x32) and accumulates in int32,
//     so every digit product is EXACT for any N here (a bf16 digit product is exact in fp32 only
//     while N 64 64 < 2^24);
//   * an operand costs 15 bytes per complex entry (5 digits x {re, im, re+im}) -- less than the
//     16 bytes of the complex128 itself, where bf16 digits would cost 30.
//
// Numerics.  Row i of A (column j of B) is scaled by a power of two s >= 4 max(|re|,|im|) and cut
// into K_DIG = 5 balanced base-128 digits, |d| <= 64: x/s = sum_t d_t 128^-(t+1) + r, |r| <= 2^-36.
// With the 3M form (T1 = ar br, T2 = ai bi, T3 = (ar+ai)(br+bi)) a complex product is
// sum over digit pairs (a,b), a+b < 5, of three exact int8 GEMMs; pairs of equal a+b share an
// int32 accumulator (|sum| <= 5 N 2^12 << 2^31).  The only error is the truncation of the digit
// series: relative 2^-35 of (row scale x column scale), i.e. the accuracy the stepper needs
// (tools/bf16_split_study.py: at 5 digits the trajectory equals the fp64 one to ~1e-12 and the
// Casimir drift is the reference's; the fixed-point tolerance is sqrt(eps)).
//
// Layout.  A sliced operand is stored [row][N/16 k-groups][15 planes][16 bytes]: the 480 bytes a
// workgroup needs of one row for one K-step (32 k) are contiguous, and one ds_read_b128 hands a
// lane its whole MFMA fragment (lane l: row l&31, k = 16 (l>>5) .. +15; probed with exact integer
// data, tools/i8probe).  The B operand is stored TRANSPOSED the same way ([column][k]); for the
// skew-Hermitian matrices of this path B[k][j] = -conj(B[j][k]), so both forms are produced by
// one row-wise pass (k_oz_slice, `conjneg`).
//
// Kernel.  64x64 output tile per workgroup, four waves of one 32x32 MFMA tile each; per K-step:
// 30 fragment reads and 45 MFMAs (1440 matrix-pipe cycles) per wave; accumulators = 15 groups x
// 16 registers (AGPRs: this file is compiled WITHOUT -amdgpu-mfma-vgpr-form); LDS double buffered
// (2 x 62 KiB); global -> register -> LDS staging of the next K-step under the MFMAs.
#include "qf_internal.h"
#include "qf_step_end.h"

// timing-only ablation knobs of the diagnostic build (tools/oz_probe.hip); results are wrong when set
#ifndef OZ_ABL_NOLOAD
#define OZ_ABL_NOLOAD 0     // no global loads in the K loop
#endif
#ifndef OZ_ABL_NOSTORE
#define OZ_ABL_NOSTORE 0    // no LDS staging writes in the K loop
#endif
#ifndef OZ_ABL_NOFRAG
#define OZ_ABL_NOFRAG 0     // no LDS fragment reads in the K loop
#endif
#ifndef OZ_ABL_NOBARRIER
#define OZ_ABL_NOBARRIER 0  // no per-K-step barrier
#endif
#ifndef OZ_ABL_NOMFMA
#define OZ_ABL_NOMFMA 0     // no MFMAs
#endif

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int K_DIG = 5;                  // digits per real value
constexpr int PLANES = 3 * K_DIG;         // {re, im, re+im} x digits
constexpr int GROUP_BYTES = PLANES * 16;  // one row, one k-group of 16: 240 bytes
constexpr int OZ_BK = 32;                 // K per step = one i8 MFMA
constexpr int OZ_T = 64;                  // tile edge
constexpr int ROW_LDS = 2 * GROUP_BYTES + 16;   // 496: padded row of an LDS stage (conflict-free b128 reads)
constexpr int STAGE_BYTES = 2 * OZ_T * ROW_LDS; // A rows then B rows: 63,488
constexpr size_t OZ_SMEM = 2 * (size_t)STAGE_BYTES;

// ---- slicing: one workgroup per (job, row); up to three operands per launch.  The row is read
// once, coalesced, into LDS (padded by one entry per 16: the 16-entry groups a lane then reads are
// bank-conflict free), its maximum gives the power-of-two scale s >= 4 max(|re|,|im|) (so that
// |re|, |im|, |re+im| <= s/2 and the leading balanced digit is <= 64), and every lane cuts one
// k-group of 16 entries into its 15 x 16 digit bytes.  conjneg: the digits of -conj(x), i.e. the
// transposed operand of a skew-Hermitian matrix.
__global__ __launch_bounds__(256) void k_oz_slice(int N, qf_oz_jobs jobs, qf_guard guard)
{
    if (!qf_guard_iter(guard)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    cplx *rowbuf = reinterpret_cast<cplx *>(smem);                 // [N + N/16]
    double *red = reinterpret_cast<double *>(rowbuf + N + N / 16);   // [4]
    const int job = blockIdx.x / N, row = blockIdx.x % N;
    const qf_oz_job jb = jobs.j[job];
    const cplx *X = jb.X;
    if (jb.X_alt && guard.state && guard.state->wh_sel) X = jb.X_alt;   // fused protocol: next step's Whalf
    const cplx *x = X + (size_t)row * N;
    const int groups = N / 16;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    double m = 0.0;
    for (int k = threadIdx.x; k < N; k += nthreads) {
        const cplx v = x[k];
        m = fmax(m, fmax(fabs(v.x), fabs(v.y)));
        rowbuf[k + (k >> 4)] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
    for (int w = 1; w < nwaves; ++w) m = fmax(m, red[w]);
    int e = 0;
    if (m > 0.0 && m < 1e300) {
        (void)frexp(m, &e);      // m = f 2^e, f in [0.5, 1)
        e += 2;
    }
    const double s = ldexp(1.0, e), inv_s = ldexp(1.0, -e);
    if (threadIdx.x == 0) jb.scale[row] = s;
    signed char *out = jb.planes + (size_t)row * groups * GROUP_BYTES;
    for (int g = threadIdx.x; g < groups; g += nthreads) {
        unsigned w[PLANES][4];          // 16 digits of each plane, packed (register resident: all loops unrolled)
#pragma unroll
        for (int p = 0; p < PLANES; ++p) w[p][0] = w[p][1] = w[p][2] = w[p][3] = 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            cplx v = rowbuf[g * 17 + j];
            if (jb.conjneg) v.x = -v.x;          // -conj(re + i im) = -re + i im
            const double r3[3] = {v.x * inv_s, v.y * inv_s, (v.x + v.y) * inv_s};
#pragma unroll
            for (int tau = 0; tau < 3; ++tau) {
                double r = r3[tau];
#pragma unroll
                for (int t = 0; t < K_DIG; ++t) {
                    const double xx = r * 128.0;
                    const double d = rint(xx);
                    r = xx - d;                  // exact: |r| <= 1/2
                    w[tau * K_DIG + t][j >> 2] |= ((unsigned)(int)d & 0xffu) << (8 * (j & 3));
                }
            }
        }
        v4u *o = reinterpret_cast<v4u *>(out + (size_t)g * GROUP_BYTES);
#pragma unroll
        for (int p = 0; p < PLANES; ++p) {
            v4u t = {w[p][0], w[p][1], w[p][2], w[p][3]};
            o[p] = t;
        }
    }
}

// ---- the product.  C = A @ B from sliced operands (pa / pb: planes, sa / sb: row / column scales).
// FUSEDEPI: the second product of an iteration with the fused epilogue and step end of
// k_zgemm<.., FUSED> (zgemm.hip): dW = C + (PW - PW^H), Whalf = W + dW, the
// speculative next state / next-step Whalf, the residual row sums, the tile ticket and the decision.
template <bool FUSEDEPI>
__global__ __launch_bounds__(256) void k_oz_gemm(int N, const signed char *__restrict__ pa, const double *__restrict__ sa,
                                                  const signed char *__restrict__ pb, const double *__restrict__ sb,
                                                  cplx *__restrict__ C, qf_epilogue ep, qf_guard guard)
{
    if (!qf_guard_iter(guard)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int tiles = N / OZ_T;
    const int tm = blockIdx.x / tiles, tn = blockIdx.x % tiles;
    const int i0 = tm * OZ_T, j0 = tn * OZ_T;
    const int row_bytes = (N / 16) * GROUP_BYTES;      // one row of a sliced operand
    const int KT = N / OZ_BK;

    // staging map: a stage holds 64 A rows and 64 B rows of 480 bytes (30 pieces of 16)
    // piece idx = tid + 256 q, q < 8: idx < 1920 -> (row = idx / 30, piece = idx % 30)
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<signed char *>(pa), 0, (int)((size_t)N * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<signed char *>(pb), 0, (int)((size_t)N * row_bytes), 0x00020000);
    unsigned voffA[8], voffB[8], ldsoff[8];
    bool live[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int idx = tid + 256 * q;
        live[q] = idx < OZ_T * 30;
        const int row = live[q] ? idx / 30 : 0, piece = live[q] ? idx % 30 : 0;
        voffA[q] = (unsigned)((size_t)(i0 + row) * row_bytes + piece * 16);
        voffB[q] = (unsigned)((size_t)(j0 + row) * row_bytes + piece * 16);
        ldsoff[q] = (unsigned)(row * ROW_LDS + piece * 16);
    }
    // (a second register set, i.e. a prefetch distance of two K-steps, was tried: 24-32 VGPRs spill)
    v4u stA[1][8], stB[1][8];
#define OZ_LOAD(kt_, SET_)                                                             \
    {                                                                                  \
        const unsigned so = (unsigned)(kt_) * (2 * GROUP_BYTES);                       \
        _Pragma("unroll") for (int q = 0; q < 8; ++q)                                  \
        {                                                                              \
            stA[SET_][q] = __builtin_amdgcn_raw_buffer_load_b128(ra, voffA[q], so, 0); \
            stB[SET_][q] = __builtin_amdgcn_raw_buffer_load_b128(rb, voffB[q], so, 0); \
        }                                                                              \
    }
#define OZ_STORE(buf_, SET_)                                                           \
    {                                                                                  \
        unsigned char *base = smem + (buf_) * STAGE_BYTES;                             \
        _Pragma("unroll") for (int q = 0; q < 8; ++q)                                  \
            if (live[q]) {                                                             \
                *reinterpret_cast<v4u *>(base + ldsoff[q]) = stA[SET_][q];             \
                *reinterpret_cast<v4u *>(base + OZ_T * ROW_LDS + ldsoff[q]) = stB[SET_][q]; \
            }                                                                          \
    }

    v16i acc[3][K_DIG];
#pragma unroll
    for (int tau = 0; tau < 3; ++tau)
#pragma unroll
        for (int s = 0; s < K_DIG; ++s)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[tau][s][q] = 0;

    const unsigned fragA = (unsigned)((wm * 32 + r) * ROW_LDS + h * GROUP_BYTES);
    const unsigned fragB = (unsigned)(OZ_T * ROW_LDS + (wn * 32 + r) * ROW_LDS + h * GROUP_BYTES);

    // Software pipeline of one K-step (fragments double buffered in registers, FA_/FB_):
    //   tau 0 fragments are already in FA_ (read at the end of the previous K-step);
    //   read tau 1 -> FB_, start the global loads of K-step kt+1, multiply tau 0;
    //   read tau 2 -> FA_, multiply tau 1;
    //   write K-step kt+1 into the other LDS stage, barrier (every read of this stage has been
    //   issued by then), read tau 0 of K-step kt+1 -> FB_, multiply tau 2.
    struct frag_t { v4i a[K_DIG], b[K_DIG]; };
    frag_t F0, F1;
#define OZ_FRAGS(F_, base_, tau_)                                                      \
    if (!OZ_ABL_NOFRAG || (tau_) == 99) {                                              \
        _Pragma("unroll") for (int d = 0; d < K_DIG; ++d)                              \
        {                                                                              \
            F_.a[d] = *reinterpret_cast<const v4i *>((base_) + fragA + ((tau_) * K_DIG + d) * 16); \
            F_.b[d] = *reinterpret_cast<const v4i *>((base_) + fragB + ((tau_) * K_DIG + d) * 16); \
        }                                                                              \
    }
#define OZ_MFMA(F_, tau_)                                                              \
    {                                                                                  \
        _Pragma("unroll") for (int a = 0; a < K_DIG; ++a)                              \
            _Pragma("unroll") for (int b = 0; b < K_DIG - a; ++b)                      \
                if (!OZ_ABL_NOMFMA)                                                    \
                    acc[tau_][a + b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F_.a[a], F_.b[b], acc[tau_][a + b], 0, 0, 0); \
                else acc[tau_][a + b][0] += F_.a[a][0] ^ F_.b[b][0];                   \
    }
#define OZ_KSTEP(FA_, FB_, kt_, BUF_)                                                  \
    {                                                                                  \
        const unsigned char *base = smem + (BUF_) * STAGE_BYTES;                       \
        const unsigned char *nbase = smem + ((BUF_) ^ 1) * STAGE_BYTES;                \
        const bool more = (kt_) + 1 < KT;                                              \
        OZ_FRAGS(FB_, base, 1)                                                         \
        if (more && !OZ_ABL_NOLOAD) OZ_LOAD((kt_) + 1, 0)                              \
        OZ_MFMA(FA_, 0)                                                                \
        OZ_FRAGS(FA_, base, 2)                                                         \
        OZ_MFMA(FB_, 1)                                                                \
        if (more && !OZ_ABL_NOSTORE) OZ_STORE((BUF_) ^ 1, 0)                           \
        if (!OZ_ABL_NOBARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        if (more) OZ_FRAGS(FB_, nbase, 0)                                              \
        OZ_MFMA(FA_, 2)                                                                \
    }
    OZ_LOAD(0, 0)
    OZ_STORE(0, 0)
    __syncthreads();
    OZ_FRAGS(F0, smem, 0)
    if (OZ_ABL_NOFRAG) {     // diagnostic: some fragments once, so that the MFMAs have defined inputs
        _Pragma("unroll") for (int d = 0; d < K_DIG; ++d)
        {
            F0.a[d] = F0.b[d] = F1.a[d] = F1.b[d] = *reinterpret_cast<const v4i *>(smem + fragA + d * 16);
        }
    }
    for (int kt = 0; kt < KT; kt += 2) {     // KT = N/32 is even (N % 64 == 0)
        OZ_KSTEP(F0, F1, kt, 0)
        OZ_KSTEP(F1, F0, kt + 1, 1)
    }
    __syncthreads();
#undef OZ_KSTEP
#undef OZ_MFMA
#undef OZ_FRAGS
#undef OZ_LOAD
#undef OZ_STORE

    // T_tau = s_a s_b sum_s G_s 128^-(s+2);  Re = T1 - T2, Im = T3 - T1 - T2
    const int gj = j0 + wn * 32 + r;
    const double sbj = sb[gj];
#define OZ_RESULT(reg_, gi_, tre_, tim_)                                               \
    {                                                                                  \
        const double sc_ = sa[gi_] * sbj;                                              \
        double T_[3];                                                                  \
        _Pragma("unroll") for (int tau = 0; tau < 3; ++tau)                            \
        {                                                                              \
            double t_ = 0.0;                                                           \
            _Pragma("unroll") for (int s_ = K_DIG - 1; s_ >= 0; --s_) /* small terms first */ \
                t_ += (double)acc[tau][s_][reg_] * (1.0 / (double)(1ull << (7 * (s_ + 2)))); \
            T_[tau] = t_ * sc_;                                                        \
        }                                                                              \
        tre_ = T_[0] - T_[1];                                                          \
        tim_ = (T_[2] - T_[0]) - T_[1];                                                \
    }
    if constexpr (!FUSEDEPI) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int gi = i0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            double tre, tim;
            OZ_RESULT(reg, gi, tre, tim)
            C[(size_t)gi * N + gj] = make_double2(tre, tim);
        }
    } else {
        const int parity = guard.state ? guard.state->dw_parity : 0;
        const cplx *__restrict__ dW_old = ep.dW[parity];
        cplx *__restrict__ dW_new = ep.dW[parity ^ 1];
        const int wpar = guard.state ? guard.state->w_parity : 0;
        const cplx *__restrict__ Wcur = wpar ? ep.Wpair[1] : ep.Wpair[0];
        cplx *__restrict__ Wnext = wpar ? ep.Wpair[0] : ep.Wpair[1];
        double *rs = reinterpret_cast<double *>(smem);     // [2][64] row sums; the K loop is done with LDS
        // four rounds of four rows: 16 operand loads in flight per lane
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            cplx pw[4], pwt[4], wv[4], dold[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int reg = 4 * q4 + u;
                const int gi = i0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const size_t e = (size_t)gi * N + gj;
                pw[u] = ep.PW[e];
                pwt[u] = ep.PW[(size_t)gj * N + gi];
                wv[u] = Wcur[e];
                dold[u] = dW_old[e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int reg = 4 * q4 + u;
                const int li = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const int gi = i0 + li;
                const size_t e = (size_t)gi * N + gj;
                double tre, tim;
                OZ_RESULT(reg, gi, tre, tim)
                // comm = PW - PW^H (conj_subtract_, isospectral.py:66-81);  dW = PW@Phalf + comm (:499,509)
                const double cr = pw[u].x - pwt[u].x, ci = pw[u].y + pwt[u].y;
                const double dr = tre + cr, di = tim + ci;
                dW_new[e] = make_double2(dr, di);
                ep.Whalf[e] = make_double2(wv[u].x + dr, wv[u].y + di);                 // isospectral.py:481-482
                const double wr = wv[u].x + 2.0 * cr, wi = wv[u].y + 2.0 * ci;         // isospectral.py:547,592
                Wnext[e] = make_double2(wr, wi);
                ep.Whalf_step[e] = make_double2(wr + dr, wi + di);
                const double er = dold[u].x - dr, ei = dold[u].y - di;                 // isospectral.py:526,534
                double a = sqrt(er * er + ei * ei);
#pragma unroll
                for (int off = 1; off < 32; off <<= 1) a += __shfl_xor(a, off, 64);    // the 32 lanes of this row
                if (r == 0) rs[wn * 64 + li] = a;
            }
        }
        __syncthreads();
        if (tid < 64)
            __hip_atomic_store(ep.rowpart + (size_t)tn * N + i0 + tid, rs[tid] + rs[64 + tid], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // the last tile to get here closes the iteration (qf_step_end.h)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned *last_flag = reinterpret_cast<unsigned *>(rs + 128);
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(ep.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *last_flag = (old == (unsigned)(ep.n_tiles - 1)) ? 1u : 0u;
        }
        __syncthreads();
        if (*last_flag != 0u)
            qf_fused_step_end(N, tiles, ep.rowpart, ep.ticket, ep.state_rw, ep.rec, guard.iter, tid, rs + 130);
    }
#undef OZ_RESULT
}

}  // namespace

size_t qf_oz_operand_bytes(int N) { return (size_t)N * (N / 16) * GROUP_BYTES; }

int qf_launch_oz_slice(qf_ctx *ctx, const qf_oz_jobs &jobs, qf_guard guard)
{
    const int N = ctx->N;
    int threads = ((N / 16 + 63) / 64) * 64;
    if (threads > 256) threads = 256;
    if (threads < 64) threads = 64;
    const size_t smem = (size_t)(N + N / 16) * sizeof(cplx) + 4 * sizeof(double);
    static size_t attr_bytes = 0;
    if (smem > 64 * 1024 && smem > attr_bytes) {
        QF_HIP(hipFuncSetAttribute((const void *)k_oz_slice, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_bytes = smem;
    }
    hipLaunchKernelGGL(k_oz_slice, dim3(jobs.n * N), dim3(threads), smem, ctx->stream, N, jobs, guard);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_oz_gemm(qf_ctx *ctx, const signed char *pa, const double *sa, const signed char *pb, const double *sb,
                      cplx *C, const qf_epilogue *ep, qf_guard guard)
{
    const int N = ctx->N;
    if (N % 64 != 0) {
        qf_set_error("qf_launch_oz_gemm: N=%d is not a multiple of 64", N);
        return QF_ERR_INVALID;
    }
    static bool attr_set = false;
    if (!attr_set) {
        QF_HIP(hipFuncSetAttribute((const void *)k_oz_gemm<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OZ_SMEM));
        QF_HIP(hipFuncSetAttribute((const void *)k_oz_gemm<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OZ_SMEM));
        attr_set = true;
    }
    const int tiles = N / 64;
    qf_epilogue e;
    if (ep) {
        e = *ep;
        e.ticket = ctx->ticket + 400;      // the tile-ticket word of the fused step end (cf. zgemm.hip launch4)
        e.n_tiles = tiles * tiles;
        e.state_rw = ctx->state;
        e.rec = ctx->host_rec;
        hipLaunchKernelGGL(k_oz_gemm<true>, dim3(tiles * tiles), dim3(256), OZ_SMEM, ctx->stream, N, pa, sa, pb, sb, C, e,
                           guard);
    } else {
        hipLaunchKernelGGL(k_oz_gemm<false>, dim3(tiles * tiles), dim3(256), OZ_SMEM, ctx->stream, N, pa, sa, pb, sb, C, e,
                           guard);
    }
    QF_HIP(hipGetLastError());
    return QF_OK;
}
