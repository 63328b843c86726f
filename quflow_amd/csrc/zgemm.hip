// Complex128 N x N x N GEMM on the gfx950 fp64 matrix cores (v_mfma_f64_16x16x4_f64),
// with the fused epilogue of the isospectral fixed-point iteration.
//
// Reference: the two np.matmul calls of quflow/integrators/isospectral.py:496,499
//      PWcomm = Phalf @ Whalf          (plain store)
//      dW     = PWcomm @ Phalf         (fused epilogue, isospectral.py:500-509,526-534)
// and, for the second product, conj_subtract_ (isospectral.py:66-81), `dW += PWcomm`,
// `Whalf = W + dW` of the NEXT iteration (isospectral.py:481-482) and the row sums of
// |dW_old - dW| that feed the residual norm (isospectral.py:526-534).
//
// Design
//   * operands stay interleaved (re,im): one ds_read_b128 gives a lane the complex entry
//     whose real and imaginary parts are the two f64 MFMA operands it needs.
//   * one complex MAC tile = 4 real MFMAs (ar*br, -ai*bi -> Re;  ar*bi, ai*br -> Im).
//   * block tile BM x BN (complex), BK = 16, register-prefetched double-buffered LDS, one
//     barrier per K-tile.  A is staged k-major (transposed) so that both fragment reads are
//     the conflict-free "16 consecutive complex per k-row" pattern of ds_read_b128.
//   * MFMA f64 16x16x4 lane maps (cdna_hip_programming.md section 3):
//        A[i = lane&15][k = lane>>4],  B[k = lane>>4][j = lane&15],
//        C[row = (lane>>4) + 4*reg][col = lane&15].
#include "qf_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK = 16;

template <int BM, int BN>
struct tile_smem {
    // A is staged k-major (transposed) and rotated by its k-row, column (i + k) mod BM: the
    // transposing ds_write_b128 and the fragment ds_read_b128 are both conflict-free without
    // padding, and two K-tile buffers of a 64x64 tile are exactly 64 KiB.
    static constexpr int A_STRIDE = BM;
    static constexpr int B_STRIDE = BN;
    static constexpr size_t main_bytes = (size_t)2 * BK * (A_STRIDE + B_STRIDE) * sizeof(cplx);
    static constexpr size_t epi_bytes = (size_t)4 * BM * sizeof(double);
    static constexpr size_t bytes = main_bytes > epi_bytes ? main_bytes : epi_bytes;
};

// Bijective XCD-aware remap (cdna_hip_programming.md, "XCD swizzle must be bijective"):
// hardware deals consecutive block ids round-robin over the 8 XCDs; give each XCD a
// contiguous range of logical tile ids so that tiles sharing operand panels share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

// Software-pipelined main loop (one wave per SIMD keeps the f64 MFMA pipe busy by itself:
// v_mfma_f64_16x16x4_f64 issues every 64 cycles even on a dependent accumulator chain --
// tools/mfma_clock.hip -- so the loop only has to make sure that no LDS/HBM wait or barrier
// ever sits between two MFMAs):
//   * fragments are double-buffered in registers: phase k4 issues the ds_reads of phase k4+1
//     before its own 4*MT*NT MFMAs;
//   * the next K-tile travels HBM/L2 -> registers one whole K-tile ahead and is written to
//     the other LDS buffer in phase 1;
//   * the single barrier per K-tile sits between phases 2 and 3, after this wave has
//     fetched its last fragments of the current buffer; phase 3 already prefetches the
//     first fragments of the next buffer.  The barrier is a raw s_barrier with lgkmcnt(0)
//     only, so the global loads in flight are not drained.
template <int BM, int BN, int WM, int WN, bool EPI, bool EXACT>
__global__ __launch_bounds__(WM *WN * 64) void k_zgemm(int N, int tiles_m, int tiles_n,
                                                        const cplx *__restrict__ A,
                                                        const cplx *__restrict__ B, cplx *__restrict__ C,
                                                        qf_epilogue ep)
{
    constexpr int T = WM * WN * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
    constexpr int MT = WTM / 16, NT = WTN / 16;  // MFMA tiles per wave
    constexpr int A_STRIDE = tile_smem<BM, BN>::A_STRIDE;
    constexpr int B_STRIDE = tile_smem<BM, BN>::B_STRIDE;
    constexpr int A_PER = (BM * BK) / T;  // complex entries each thread stages per K-tile
    constexpr int B_PER = (BN * BK) / T;
    static_assert((BM * BK) % T == 0 && (BN * BK) % T == 0, "tile/threads mismatch");
    static_assert((BM & (BM - 1)) == 0, "BM must be a power of two (rotation mask)");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cplx *As = reinterpret_cast<cplx *>(smem_raw);           // [2][BK][A_STRIDE]
    cplx *Bs = As + (size_t)2 * BK * A_STRIDE;               // [2][BK][B_STRIDE]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, q4 = lane >> 4;

    const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int i0 = tm * BM, j0 = tn * BN;
    const cplx zero = make_double2(0.0, 0.0);

    // ---- epilogue operands of the second product are fetched FIRST and held in registers
    // (one wave per SIMD owns all 512 VGPRs): their latency hides under the whole main loop
    // and the epilogue itself is register arithmetic plus two streaming stores.
    cplx e_pw[EPI ? MT : 1][EPI ? NT : 1][4], e_pwt[EPI ? MT : 1][EPI ? NT : 1][4];
    cplx e_w[EPI ? MT : 1][EPI ? NT : 1][4], e_old[EPI ? MT : 1][EPI ? NT : 1][4];
    if constexpr (EPI) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int gi = i0 + wm * WTM + mi * 16 + q4 + 4 * reg;
                    const int gj = j0 + wn * WTN + ni * 16 + r16;
                    e_pw[mi][ni][reg] = zero;
                    e_pwt[mi][ni][reg] = zero;
                    e_w[mi][ni][reg] = zero;
                    e_old[mi][ni][reg] = zero;
                    if (EXACT || (gi < N && gj < N)) {
                        const size_t e = (size_t)gi * N + gj;
                        e_pw[mi][ni][reg] = ep.PW[e];
                        e_pwt[mi][ni][reg] = ep.PW[(size_t)gj * N + gi];  // mirrored entry, 64-B row segments
                        e_w[mi][ni][reg] = ep.W[e];
                        e_old[mi][ni][reg] = ep.dW_old[e];
                    }
                }
    }

    v4d accR[MT][NT], accI[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            accR[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            accI[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        }

    cplx ra[A_PER], rb[B_PER];      // K-tile in flight HBM -> registers -> LDS
    cplx fa[2][MT], fb[2][NT];      // double-buffered MFMA fragments

    // staging helpers are macros on purpose: lambdas capturing the register arrays by
    // reference made hipcc keep them in scratch memory
#define QF_LOAD_TILE(k0_)                                                              \
    {                                                                                  \
        const int k0v = (k0_);                                                         \
        _Pragma("unroll") for (int r = 0; r < A_PER; ++r)                              \
        {                                                                              \
            const int idx = tid + r * T;                                               \
            const int i = idx / BK, kk = idx % BK; /* lanes run along k: 256-B rows */ \
            const int gi = i0 + i, gk = k0v + kk;                                      \
            ra[r] = zero;                                                              \
            if (EXACT || (gi < N && gk < N)) ra[r] = A[(size_t)gi * N + gk];           \
        }                                                                              \
        _Pragma("unroll") for (int r = 0; r < B_PER; ++r)                              \
        {                                                                              \
            const int idx = tid + r * T;                                               \
            const int kk = idx / BN, jj = idx % BN; /* lanes run along j: full rows */ \
            const int gk = k0v + kk, gj = j0 + jj;                                     \
            rb[r] = zero;                                                              \
            if (EXACT || (gk < N && gj < N)) rb[r] = B[(size_t)gk * N + gj];           \
        }                                                                              \
    }
#define QF_STORE_TILE(buf_)                                                            \
    {                                                                                  \
        cplx *as_w = As + (size_t)(buf_) * BK * A_STRIDE;                              \
        cplx *bs_w = Bs + (size_t)(buf_) * BK * B_STRIDE;                              \
        _Pragma("unroll") for (int r = 0; r < A_PER; ++r)                              \
        {                                                                              \
            const int idx = tid + r * T;                                               \
            const int i = idx / BK, kk = idx % BK;                                     \
            as_w[kk * A_STRIDE + ((i + kk) & (BM - 1))] = ra[r];                       \
        }                                                                              \
        _Pragma("unroll") for (int r = 0; r < B_PER; ++r)                              \
        {                                                                              \
            const int idx = tid + r * T;                                               \
            const int kk = idx / BN, jj = idx % BN;                                    \
            bs_w[kk * B_STRIDE + jj] = rb[r];                                          \
        }                                                                              \
    }
#define QF_READ_FRAGS(set_, buf_, k4_)                                                 \
    {                                                                                  \
        const int krow = (k4_) * 4 + q4;                                               \
        const cplx *as_r = As + ((size_t)(buf_) * BK + krow) * A_STRIDE;               \
        const cplx *bs_r = Bs + ((size_t)(buf_) * BK + krow) * B_STRIDE + wn * WTN + r16; \
        _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                              \
            fa[set_][mi] = as_r[(wm * WTM + mi * 16 + r16 + krow) & (BM - 1)];         \
        _Pragma("unroll") for (int ni = 0; ni < NT; ++ni) fb[set_][ni] = bs_r[ni * 16]; \
    }
    // 4*MT*NT MFMAs; first products on every accumulator, then the second ones
#define QF_MFMA(set_)                                                                  \
    {                                                                                  \
        _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                              \
            _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                          \
        {                                                                              \
            accR[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set_][mi].x, fb[set_][ni].x, accR[mi][ni], 0, 0, 0); \
            accI[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set_][mi].x, fb[set_][ni].y, accI[mi][ni], 0, 0, 0); \
        }                                                                              \
        _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                              \
        {                                                                              \
            const double nai = -fa[set_][mi].y;                                        \
            _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                          \
            {                                                                          \
                accR[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, fb[set_][ni].y, accR[mi][ni], 0, 0, 0); \
                accI[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[set_][mi].y, fb[set_][ni].x, accI[mi][ni], 0, 0, 0); \
            }                                                                          \
        }                                                                              \
    }

    const int KT = (N + BK - 1) / BK;
    QF_LOAD_TILE(0)
    QF_STORE_TILE(0)
    __syncthreads();
    if (KT > 1) QF_LOAD_TILE(BK)
    QF_READ_FRAGS(0, 0, 0)

    // instruction-class masks of __builtin_amdgcn_sched_group_barrier
    constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_WR = 0x200;
    // One K-tile.  STORE_/LOAD_/NEXT_ are literal 0/1 so that the steady-state body is a
    // single basic block (the loop is peeled below): only then can the scheduler place one
    // staging instruction into each MFMA gap of phase 1.
#define QF_KTILE(kt_, STORE_, LOAD_, NEXT_)                                            \
    {                                                                                  \
        const int cur = (kt_) & 1, nxt = cur ^ 1;                                      \
        /* phase 0: fetch the fragments of phase 1, then multiply */                   \
        QF_READ_FRAGS(1, cur, 1)                                                       \
        __builtin_amdgcn_sched_barrier(0);                                             \
        QF_MFMA(0)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* phase 1: the other LDS buffer is free (every wave passed the barrier of    \
           K-tile kt-1): write K-tile kt+1 into it, start fetching K-tile kt+2 */      \
        QF_READ_FRAGS(0, cur, 2)                                                       \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if (STORE_) QF_STORE_TILE(nxt)                                                 \
        if (LOAD_) QF_LOAD_TILE(((kt_) + 2) * BK)                                      \
        QF_MFMA(1)                                                                     \
        if (EXACT && (STORE_)) {                                                       \
            _Pragma("unroll") for (int g = 0; g < A_PER + B_PER; ++g)                  \
            {                                                                          \
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);                   \
                __builtin_amdgcn_sched_group_barrier(SG_DS_WR, 1, 0);                  \
            }                                                                          \
        }                                                                              \
        if (EXACT && (LOAD_)) {                                                        \
            _Pragma("unroll") for (int g = 0; g < A_PER + B_PER; ++g)                  \
            {                                                                          \
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);                   \
                __builtin_amdgcn_sched_group_barrier(SG_VMEM_RD, 1, 0);                \
            }                                                                          \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* phase 2 */                                                                  \
        QF_READ_FRAGS(1, cur, 3)                                                       \
        __builtin_amdgcn_sched_barrier(0);                                             \
        QF_MFMA(0)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* my LDS reads of `cur` have landed and my writes to `nxt` are done */        \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* phase 3: first fragments of the next K-tile */                              \
        if (NEXT_) QF_READ_FRAGS(0, nxt, 0)                                            \
        __builtin_amdgcn_sched_barrier(0);                                             \
        QF_MFMA(1)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* lgkmcnt(0) only (0xC07F): by now the prefetch has landed; stating it keeps  \
           hipcc from waiting conservatively at the loop head, across the back edge */ \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                            \
        __builtin_amdgcn_sched_barrier(0);                                             \
    }

    __builtin_amdgcn_s_waitcnt(0xC07F);
    int kt = 0;
    for (; kt + 2 < KT; ++kt) QF_KTILE(kt, 1, 1, 1)
    if (kt + 1 < KT) {
        QF_KTILE(kt, 1, 0, 1)
        ++kt;
    }
    QF_KTILE(kt, 0, 0, 0)
#undef QF_KTILE
#undef QF_LOAD_TILE
#undef QF_STORE_TILE
#undef QF_READ_FRAGS
#undef QF_MFMA

    if constexpr (!EPI) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    int gi = i0 + wm * WTM + mi * 16 + q4 + 4 * reg;
                    int gj = j0 + wn * WTN + ni * 16 + r16;
                    if (EXACT || (gi < N && gj < N)) C[(size_t)gi * N + gj] = make_double2(accR[mi][ni][reg], accI[mi][ni][reg]);
                }
    } else {
        // ---- fused epilogue of the second product (operands already in registers)
        double *rs = reinterpret_cast<double *>(smem_raw);  // [WN][BM]; the K-loop is done with LDS
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int li = wm * WTM + mi * 16 + q4 + 4 * reg;
                const int gi = i0 + li;
                double rsum = 0.0;
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    const int gj = j0 + wn * WTN + ni * 16 + r16;
                    if (EXACT || (gi < N && gj < N)) {
                        const size_t e = (size_t)gi * N + gj;
                        const cplx pw = e_pw[mi][ni][reg];
                        const cplx pwt = e_pwt[mi][ni][reg];
                        // conj_subtract_: PW[i,j] - conj(PW[j,i])   (isospectral.py:71-74)
                        const double cr = pw.x - pwt.x;
                        const double ci = pw.y + pwt.y;
                        // dW = (PW @ Phalf) + comm                  (isospectral.py:499,509)
                        const double dr = accR[mi][ni][reg] + cr;
                        const double di = accI[mi][ni][reg] + ci;
                        ep.dW_new[e] = make_double2(dr, di);
                        // Whalf = W + dW for the next iteration      (isospectral.py:481-482)
                        const cplx w = e_w[mi][ni][reg];
                        ep.Whalf[e] = make_double2(w.x + dr, w.y + di);
                        // |dW_old - dW|                             (isospectral.py:526,534)
                        const cplx o = e_old[mi][ni][reg];
                        const double er = o.x - dr, ei = o.y - di;
                        rsum += sqrt(er * er + ei * ei);
                    }
                }
                // sum over the 16 lanes that share this row (fixed butterfly: deterministic)
                rsum += __shfl_xor(rsum, 1, 64);
                rsum += __shfl_xor(rsum, 2, 64);
                rsum += __shfl_xor(rsum, 4, 64);
                rsum += __shfl_xor(rsum, 8, 64);
                if (r16 == 0) rs[wn * BM + li] = rsum;
            }
        }
        __syncthreads();
        for (int li = tid; li < BM; li += T) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < WN; ++c) s += rs[c * BM + li];
            if (EXACT || i0 + li < N) ep.rowpart[(size_t)tn * N + i0 + li] = s;
        }
    }
}

struct gemm_cfg {
    int BM, BN;
};

gemm_cfg pick_gemm(int N)
{
    // fill the 256 CUs: 64x64 tiles from N = 1024 up (>= 256 tiles), 32x32 below
    gemm_cfg c;
    if (N >= 768) { c.BM = 64; c.BN = 64; }
    else { c.BM = 32; c.BN = 32; }
    return c;
}

template <int BM, int BN, int WM, int WN, bool EPI, bool EXACT>
int launch2(qf_ctx *ctx, const cplx *A, const cplx *B, cplx *C, const qf_epilogue &ep)
{
    const int N = ctx->N;
    const int tiles_m = (N + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const size_t smem = tile_smem<BM, BN>::bytes;
    static_assert(tile_smem<BM, BN>::bytes <= 64 * 1024, "dynamic LDS above 64 KiB needs hipFuncSetAttribute");
    dim3 grid(tiles_m * tiles_n), block(WM * WN * 64);
    hipLaunchKernelGGL((k_zgemm<BM, BN, WM, WN, EPI, EXACT>), grid, block, smem, ctx->stream, N, tiles_m, tiles_n,
                       A, B, C, ep);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

template <int BM, int BN, int WM, int WN>
int launch(qf_ctx *ctx, const cplx *A, const cplx *B, cplx *C, const qf_epilogue *ep)
{
    const int N = ctx->N;
    const bool exact = (N % BM == 0) && (N % BN == 0) && (N % BK == 0);
    qf_epilogue none;
    if (ep) {
        if (exact) return launch2<BM, BN, WM, WN, true, true>(ctx, A, B, C, *ep);
        return launch2<BM, BN, WM, WN, true, false>(ctx, A, B, C, *ep);
    }
    if (exact) return launch2<BM, BN, WM, WN, false, true>(ctx, A, B, C, none);
    return launch2<BM, BN, WM, WN, false, false>(ctx, A, B, C, none);
}

}  // namespace

int qf_gemm_tiles_n(int N)
{
    gemm_cfg c = pick_gemm(N);
    return (N + c.BN - 1) / c.BN;
}

int qf_launch_zgemm(qf_ctx *ctx, const cplx *A, const cplx *B, cplx *C, const qf_epilogue *ep)
{
    gemm_cfg c = pick_gemm(ctx->N);
    if (c.BM == 64) return launch<64, 64, 2, 2>(ctx, A, B, C, ep);
    return launch<32, 32, 2, 2>(ctx, A, B, C, ep);
}
