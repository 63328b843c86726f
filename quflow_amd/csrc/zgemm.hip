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
    static constexpr int A_STRIDE = BM + 1;  // complex entries per k-row of the transposed A tile
    static constexpr int B_STRIDE = BN;
    static constexpr size_t main_bytes = (size_t)2 * BK * (A_STRIDE + B_STRIDE) * sizeof(cplx);
    static constexpr size_t epi_bytes = (size_t)BN * (BM + 1) * sizeof(cplx) + (size_t)4 * BM * sizeof(double);
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

template <int BM, int BN, int WM, int WN, bool EPI>
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

    v4d accR[MT][NT], accI[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            accR[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            accI[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        }

    cplx ra[A_PER], rb[B_PER];
    const cplx zero = make_double2(0.0, 0.0);

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
            if (gi < N && gk < N) ra[r] = A[(size_t)gi * N + gk];                      \
        }                                                                              \
        _Pragma("unroll") for (int r = 0; r < B_PER; ++r)                              \
        {                                                                              \
            const int idx = tid + r * T;                                               \
            const int kk = idx / BN, jj = idx % BN; /* lanes run along j: full rows */ \
            const int gk = k0v + kk, gj = j0 + jj;                                     \
            rb[r] = zero;                                                              \
            if (gk < N && gj < N) rb[r] = B[(size_t)gk * N + gj];                      \
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
            as_w[kk * A_STRIDE + i] = ra[r];                                           \
        }                                                                              \
        _Pragma("unroll") for (int r = 0; r < B_PER; ++r)                              \
        {                                                                              \
            const int idx = tid + r * T;                                               \
            const int kk = idx / BN, jj = idx % BN;                                    \
            bs_w[kk * B_STRIDE + jj] = rb[r];                                          \
        }                                                                              \
    }

    const int KT = (N + BK - 1) / BK;
    QF_LOAD_TILE(0)
    QF_STORE_TILE(0)
    __syncthreads();

    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) QF_LOAD_TILE((kt + 1) * BK)

        const cplx *as = As + (size_t)cur * BK * A_STRIDE + wm * WTM + r16;
        const cplx *bs = Bs + (size_t)cur * BK * B_STRIDE + wn * WTN + r16;
#pragma unroll
        for (int k4 = 0; k4 < BK / 4; ++k4) {
            cplx a[MT], b[NT];
            const int krow = k4 * 4 + q4;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) a[mi] = as[krow * A_STRIDE + mi * 16];
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) b[ni] = bs[krow * B_STRIDE + ni * 16];
            // first products on every accumulator, then the second ones: consecutive
            // MFMAs never depend on each other
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    accR[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi].x, b[ni].x, accR[mi][ni], 0, 0, 0);
                    accI[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi].x, b[ni].y, accI[mi][ni], 0, 0, 0);
                }
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const double nai = -a[mi].y;
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    accR[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(nai, b[ni].y, accR[mi][ni], 0, 0, 0);
                    accI[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi].y, b[ni].x, accI[mi][ni], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < KT) QF_STORE_TILE(cur ^ 1)
        __syncthreads();
    }

    if constexpr (!EPI) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    int gi = i0 + wm * WTM + mi * 16 + q4 + 4 * reg;
                    int gj = j0 + wn * WTN + ni * 16 + r16;
                    if (gi < N && gj < N) C[(size_t)gi * N + gj] = make_double2(accR[mi][ni][reg], accI[mi][ni][reg]);
                }
    } else {
        // ---- fused epilogue of the second product.
        // Stage the mirrored tile PW[j0.., i0..] in LDS (coalesced rows), read it transposed.
        constexpr int TS = BM + 1;
        cplx *Ts = reinterpret_cast<cplx *>(smem_raw);                 // [BN][TS]
        double *rs = reinterpret_cast<double *>(Ts + (size_t)BN * TS);  // [WN][BM]
        for (int idx = tid; idx < BN * BM; idx += T) {
            int jj = idx / BM, ii = idx % BM;
            int gj = j0 + jj, gi = i0 + ii;
            cplx tv = zero;
            if (gj < N && gi < N) tv = ep.PW[(size_t)gj * N + gi];
            Ts[jj * TS + ii] = tv;
        }
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
                    const int lj = wn * WTN + ni * 16 + r16;
                    const int gj = j0 + lj;
                    if (gi < N && gj < N) {
                        const size_t e = (size_t)gi * N + gj;
                        const cplx pw = ep.PW[e];
                        const cplx pwt = Ts[lj * TS + li];
                        // conj_subtract_: PW[i,j] - conj(PW[j,i])   (isospectral.py:71-74)
                        const double cr = pw.x - pwt.x;
                        const double ci = pw.y + pwt.y;
                        // dW = (PW @ Phalf) + comm                  (isospectral.py:499,509)
                        const double dr = accR[mi][ni][reg] + cr;
                        const double di = accI[mi][ni][reg] + ci;
                        ep.dW_new[e] = make_double2(dr, di);
                        // Whalf = W + dW for the next iteration      (isospectral.py:481-482)
                        const cplx w = ep.W[e];
                        ep.Whalf[e] = make_double2(w.x + dr, w.y + di);
                        // |dW_old - dW|                             (isospectral.py:526,534)
                        const cplx o = ep.dW_old[e];
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
            if (i0 + li < N) ep.rowpart[(size_t)tn * N + i0 + li] = s;
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

template <int BM, int BN, int WM, int WN>
int launch(qf_ctx *ctx, const cplx *A, const cplx *B, cplx *C, const qf_epilogue *ep)
{
    const int N = ctx->N;
    const int tiles_m = (N + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const size_t smem = tile_smem<BM, BN>::bytes;
    dim3 grid(tiles_m * tiles_n), block(WM * WN * 64);
    if (ep) {
        static bool attr_set = false;
        if (!attr_set && smem > 64 * 1024) {
            QF_HIP(hipFuncSetAttribute((const void *)k_zgemm<BM, BN, WM, WN, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_set = true;
        }
        hipLaunchKernelGGL((k_zgemm<BM, BN, WM, WN, true>), grid, block, smem, ctx->stream, N, tiles_m, tiles_n,
                           A, B, C, *ep);
    } else {
        static bool attr_set = false;
        if (!attr_set && smem > 64 * 1024) {
            QF_HIP(hipFuncSetAttribute((const void *)k_zgemm<BM, BN, WM, WN, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            attr_set = true;
        }
        qf_epilogue none;
        hipLaunchKernelGGL((k_zgemm<BM, BN, WM, WN, false>), grid, block, smem, ctx->stream, N, tiles_m, tiles_n,
                           A, B, C, none);
    }
    QF_HIP(hipGetLastError());
    return QF_OK;
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
