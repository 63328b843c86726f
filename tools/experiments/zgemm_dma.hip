// EXPERIMENT (round 2), not part of the library: measured slower than k_zgemm (101.0 against 97.0 us per launch at
// N = 1024) -- see DESIGN.md section 3.1 "both operands by LDS DMA".  Kept with its probe (pw_dma_probe.hip) because the
// ablation numbers in DESIGN.md come from it.  Bit-identical results to k_zgemm (checked through the stepper).
// First product of a fixed-point iteration, PW = Phalf @ Whalf (isospectral.py:496), for the case the
// stepper itself produces: Phalf comes out of the skew-Hermitian Poisson solve, so it is EXACTLY
// skew-Hermitian (k_solve writes P[j,i] = -conj(P[i,j]) from the same value).
//
// That symmetry removes the one thing the general kernel (zgemm.hip: k_zgemm) needs registers and LDS
// store instructions for -- transposing the left operand's tile to k-major:
//   A[i,k] = -conj(P[k,i]),
// so "column k of the A tile" is 64 CONSECUTIVE entries of ROW k of P, exactly like a k-row of the right
// operand's tile.  Both operands therefore travel global memory -> LDS by DMA (buffer_load ... lds, 1 KiB
// per wave instruction: one k-row of one tile), with no staging registers, no ds_write and no transposition;
// the sign and the conjugation are folded into the 3M combination after the K loop:
//   loaded s = P[k,i] = sr + i si   =>   Ar = -sr, Ai = si
//   accR += sr * br          T1 = sum Ar Br = -accR
//   accI += si * bi          T2 = sum Ai Bi =  accI
//   accS += (si - sr)(br+bi) T3 = sum (Ar+Ai)(Br+Bi) = accS
//   Re = T1 - T2 = (-accR) - accI,   Im = (T3 - T1) - T2 = (accS + accR) - accI
// Every product and every partial sum is the negative of (or equal to) the one k_zgemm forms, in the same
// order, so the result is bit-identical to k_zgemm's (negation is exact) -- the parity tests of the stepper
// do not know which kernel ran.
//
// 64 x 64 tiles, 4 wavefronts (2 x 2, 32 x 32 each: 2 x 2 MFMA 16x16x4 f64 tiles x 3 accumulators), K-tiles
// of 16, three LDS stages per operand (96 KiB): the DMA of K-tile kt+3 is issued into the stage K-tile kt
// has just been read from, two K-tiles before anybody needs it.  One barrier per K-tile, ahead of its
// fourth MFMA group.  N % 64 == 0 only; the host picks this kernel where it would pick k_zgemm's
// 64 x 64 form (N >= 768).
#include "../../quflow_amd/csrc/qf_internal.h"

// timing-only ablation knobs of tools/pw_dma_probe.hip (results wrong when set; never set in the library)
#ifndef PW_ABL_NODMA
#define PW_ABL_NODMA 0
#endif
#ifndef PW_ABL_NOBARRIER
#define PW_ABL_NOBARRIER 0
#endif
#ifndef PW_ABL_NOREAD
#define PW_ABL_NOREAD 0
#endif
#ifndef PW_ABL_NOSUMS
#define PW_ABL_NOSUMS 0
#endif

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int PT = 64;                   // tile edge
constexpr int PK = 16;                   // K-tile depth
constexpr int P_ROW = PT * (int)sizeof(cplx);       // one k-row of a staged tile: 1 KiB
constexpr int P_TILE = PK * P_ROW;                  // 16 KiB
#ifndef PW_STAGES
#define PW_STAGES 3
#endif
constexpr int P_STAGES = PW_STAGES;
constexpr size_t PW_SMEM_BYTES = (size_t)2 * P_STAGES * P_TILE;   // 96 KiB
constexpr int P_DMA_PER_WAVE = 8;        // 4 k-rows of each operand per wave and K-tile

// same bijective XCD-aware tile order as k_zgemm (zgemm.hip: xcd_remap)
__device__ __forceinline__ int pw_xcd_remap(int bid, int nwg)
{
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

__device__ __forceinline__ void pw_dma16(__amdgpu_buffer_rsrc_t rsrc, lds_void *dst, unsigned voffset, unsigned soffset)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, soffset, 0, 0);
}

struct pw_frag {
    cplx a[2], b[2];
    double as[2], bs[2];
};

__global__ __launch_bounds__(256) void k_zgemm_pw_dma(int N, const cplx *__restrict__ P, const cplx *__restrict__ B,
                                                       cplx *__restrict__ C, qf_guard guard)
{
    // (the tag is checked first here: a launch that is not due must not leave DMA writes in flight
    // into an LDS allocation it has already given back)
    if (!qf_guard_iter(guard)) return;
    // fused step end: the first product of a step's first iteration takes the Whalf prepared for it
    if (guard.alt && guard.state->wh_sel) B = static_cast<const cplx *>(guard.alt);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int nt = N / PT;
    const int lid = pw_xcd_remap(blockIdx.x, nt * nt);
    const int tm = lid / nt, tn = lid % nt;
    const int i0 = tm * PT, j0 = tn * PT;
    const int KT = N / PK;

    const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(P), 0, (int)((size_t)N * N * sizeof(cplx)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(B), 0, (int)((size_t)N * N * sizeof(cplx)), 0x00020000);
    const unsigned voff = (unsigned)lane * (unsigned)sizeof(cplx);
    const unsigned row_bytes = (unsigned)N * (unsigned)sizeof(cplx);
    const unsigned a_col = (unsigned)i0 * (unsigned)sizeof(cplx), b_col = (unsigned)j0 * (unsigned)sizeof(cplx);

    // K-tile kt_ into stage st_: this wave's four k-rows of both tiles (past the last K-tile: the last one
    // again, into a stage nobody reads any more -- the waits below stay uniform)
#define PW_DMA(kt_, st_)                                                               \
    if (!PW_ABL_NODMA) {                                                               \
        const int kk_ = (kt_) < KT ? (kt_) : KT - 1;                                   \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                  \
        {                                                                              \
            const int k_ = wave * 4 + r;                                               \
            const unsigned grow_ = (unsigned)(kk_ * PK + k_) * row_bytes;              \
            pw_dma16(rsrcP, (lds_void *)(smem + (st_) * P_TILE + k_ * P_ROW), voff, grow_ + a_col); \
            pw_dma16(rsrcB, (lds_void *)(smem + (P_STAGES + (st_)) * P_TILE + k_ * P_ROW), voff, grow_ + b_col); \
        }                                                                              \
    }

    // one of this wave's eight DMA instructions of a K-tile: q_ = 0..3 rows of A, 4..7 rows of B
#define PW_DMA1(kt_, st_, q_)                                                          \
    if (!PW_ABL_NODMA) {                                                               \
        const int kk_ = (kt_) < KT ? (kt_) : KT - 1;                                   \
        const int k_ = wave * 4 + ((q_) & 3);                                          \
        const unsigned grow_ = (unsigned)(kk_ * PK + k_) * row_bytes;                  \
        if ((q_) < 4) pw_dma16(rsrcP, (lds_void *)(smem + (st_) * P_TILE + k_ * P_ROW), voff, grow_ + a_col); \
        else pw_dma16(rsrcB, (lds_void *)(smem + (P_STAGES + (st_)) * P_TILE + k_ * P_ROW), voff, grow_ + b_col); \
    }

    v4d accR[2][2], accI[2][2], accS[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            accR[mi][ni] = (v4d){0.0, 0.0, 0.0, 0.0};
            accI[mi][ni] = (v4d){0.0, 0.0, 0.0, 0.0};
            accS[mi][ni] = (v4d){0.0, 0.0, 0.0, 0.0};
        }

    // fragment addresses: k = 4 K4 + q4 of the stage, 16 consecutive entries per group of 16 lanes
    // (256 B = every bank once: conflict-free ds_read_b128)
    const unsigned lds_base = (unsigned)(size_t)smem;      // (low half of the flat address of an LDS object = its LDS byte address)
    const unsigned lds_a = lds_base + (unsigned)(q4 * P_ROW + (wm * 32 + r16) * (int)sizeof(cplx));
    const unsigned lds_b = lds_base + (unsigned)(P_STAGES * P_TILE + q4 * P_ROW + (wn * 32 + r16) * (int)sizeof(cplx));

// Fragment reads and their waits are written as asm: a ds_read the compiler can see makes it wait for
// EVERY LDS DMA in flight first (s_waitcnt vmcnt(0): it cannot tell the stages apart), which would expose
// the latency of the K-tile requested a moment ago in every K-tile.  The wait carries the fragment
// registers as operands so that no MFMA that consumes them can be scheduled above it.
#define PW_READ(F_, st_, K4_)                                                          \
    {                                                                                  \
        const unsigned pa_ = lds_a + (unsigned)((st_) * P_TILE);                       \
        const unsigned pb_ = lds_b + (unsigned)((st_) * P_TILE);                       \
        v4u t0_ = {1, 2, 3, 4}, t1_ = t0_, t2_ = t0_, t3_ = t0_;                       \
        if (!PW_ABL_NOREAD)                                                            \
        asm volatile("ds_read_b128 %0, %4 offset:%6\n\t"                               \
                     "ds_read_b128 %1, %4 offset:%7\n\t"                               \
                     "ds_read_b128 %2, %5 offset:%6\n\t"                               \
                     "ds_read_b128 %3, %5 offset:%7"                                    \
                     : "+&v"(t0_), "+&v"(t1_), "+&v"(t2_), "+&v"(t3_)                   \
                     : "v"(pa_), "v"(pb_), "n"((K4_) * 4 * P_ROW), "n"((K4_) * 4 * P_ROW + 16 * (int)sizeof(cplx)) \
                     : "memory");                                                      \
        F_.a[0] = *reinterpret_cast<const cplx *>(&t0_);                               \
        F_.a[1] = *reinterpret_cast<const cplx *>(&t1_);                               \
        F_.b[0] = *reinterpret_cast<const cplx *>(&t2_);                               \
        F_.b[1] = *reinterpret_cast<const cplx *>(&t3_);                               \
    }
#define PW_LANDED(F_)                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                \
                 : "+v"(F_.a[0].x), "+v"(F_.a[0].y), "+v"(F_.a[1].x), "+v"(F_.a[1].y),  \
                   "+v"(F_.b[0].x), "+v"(F_.b[0].y), "+v"(F_.b[1].x), "+v"(F_.b[1].y)   \
                 :: "memory");
#define PW_SUMS(F_)                                                                    \
    if (PW_ABL_NOSUMS) {                                                               \
        F_.as[0] = F_.a[0].y; F_.as[1] = F_.a[1].y; F_.bs[0] = F_.b[0].x; F_.bs[1] = F_.b[1].x; \
    } else {                                                                           \
        F_.as[0] = F_.a[0].y - F_.a[0].x;                                              \
        F_.as[1] = F_.a[1].y - F_.a[1].x;                                              \
        F_.bs[0] = F_.b[0].x + F_.b[0].y;                                              \
        F_.bs[1] = F_.b[1].x + F_.b[1].y;                                              \
    }
#define PW_MFMA1(F_, mi_, ni_)                                                         \
    {                                                                                  \
        accR[mi_][ni_] = __builtin_amdgcn_mfma_f64_16x16x4f64(F_.a[mi_].x, F_.b[ni_].x, accR[mi_][ni_], 0, 0, 0); \
        accI[mi_][ni_] = __builtin_amdgcn_mfma_f64_16x16x4f64(F_.a[mi_].y, F_.b[ni_].y, accI[mi_][ni_], 0, 0, 0); \
        accS[mi_][ni_] = __builtin_amdgcn_mfma_f64_16x16x4f64(F_.as[mi_], F_.bs[ni_], accS[mi_][ni_], 0, 0, 0);   \
    }
// one MFMA, then an (empty) asm that takes its accumulator and the NEXT MFMA's: the memory operation that
// follows in the source (a DMA instruction) stays between the two MFMAs -- issued in the first one's shadow
// instead of in a burst that holds the matrix pipe up (8 back-to-back DMA issues cost 11 us per launch)
#define PW_M_PIN(acc_, a_, b_, next_)                                                  \
    {                                                                                  \
        acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, acc_, 0, 0, 0);            \
        asm volatile("" : "+v"(acc_), "+v"(next_) :: "memory");                        \
    }
// A slot boundary: sched_barrier alone does not keep a later pass from sinking half of a slot's MFMAs below
// the following slots (seen in the ISA: 6 of 12 stayed); an empty asm that takes every accumulator in and
// out pins the order -- each MFMA of the slot before it, each MFMA of the next slot behind it.
#define PW_FENCE()                                                                     \
    {                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("" : "+v"(accR[0][0]), "+v"(accR[0][1]), "+v"(accR[1][0]), "+v"(accR[1][1]), \
                          "+v"(accI[0][0]), "+v"(accI[0][1]), "+v"(accI[1][0]), "+v"(accI[1][1]), \
                          "+v"(accS[0][0]), "+v"(accS[0][1]), "+v"(accS[1][0]), "+v"(accS[1][1])); \
        __builtin_amdgcn_sched_barrier(0);                                             \
    }

    pw_frag F0, F1;

#pragma unroll
    for (int q = 0; q < P_STAGES; ++q) PW_DMA(q, q)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((P_STAGES - 1) * P_DMA_PER_WAVE) : "memory");   // K-tile 0 is in, everywhere
    PW_READ(F0, 0, 0)

    int st = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const int st_next = (st == P_STAGES - 1) ? 0 : st + 1;
        // ---- K4 = 0
        PW_LANDED(F0)
        PW_READ(F1, st, 1)
        PW_FENCE()      // (the reads go out before this slot's MFMAs, not behind them)
        PW_SUMS(F0)
        PW_MFMA1(F0, 0, 0) PW_MFMA1(F0, 0, 1) PW_MFMA1(F0, 1, 0) PW_MFMA1(F0, 1, 1)
        PW_FENCE()
        // ---- K4 = 1
        PW_LANDED(F1)
        PW_READ(F0, st, 2)
        PW_FENCE()
        PW_SUMS(F1)
        PW_MFMA1(F1, 0, 0) PW_MFMA1(F1, 0, 1) PW_MFMA1(F1, 1, 0) PW_MFMA1(F1, 1, 1)
        PW_FENCE()
        // ---- K4 = 2: the last fragments of this stage
        PW_LANDED(F0)
        PW_READ(F1, st, 3)
        PW_FENCE()
        PW_SUMS(F0)
        PW_MFMA1(F0, 0, 0) PW_MFMA1(F0, 0, 1) PW_MFMA1(F0, 1, 0) PW_MFMA1(F0, 1, 1)
        PW_FENCE()
        // ---- K4 = 3.  The hand-over first: this wave is done with stage st, K-tile kt+1 has landed; after
        // the barrier both hold for everybody -- the next K-tile's first fragments can be read and stage st
        // takes K-tile kt + P_STAGES, one DMA instruction behind each of the first eight MFMAs
        PW_LANDED(F1)
        if (PW_ABL_NOBARRIER) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((P_STAGES - 2) * P_DMA_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((P_STAGES - 2) * P_DMA_PER_WAVE) : "memory");
        PW_READ(F0, st_next, 0)
        PW_FENCE()
        PW_SUMS(F1)
        PW_M_PIN(accR[0][0], F1.a[0].x, F1.b[0].x, accI[0][0]) PW_DMA1(kt + P_STAGES, st, 0)
        PW_M_PIN(accI[0][0], F1.a[0].y, F1.b[0].y, accS[0][0]) PW_DMA1(kt + P_STAGES, st, 4)
        PW_M_PIN(accS[0][0], F1.as[0], F1.bs[0], accR[0][1])   PW_DMA1(kt + P_STAGES, st, 1)
        PW_M_PIN(accR[0][1], F1.a[0].x, F1.b[1].x, accI[0][1]) PW_DMA1(kt + P_STAGES, st, 5)
        PW_M_PIN(accI[0][1], F1.a[0].y, F1.b[1].y, accS[0][1]) PW_DMA1(kt + P_STAGES, st, 2)
        PW_M_PIN(accS[0][1], F1.as[0], F1.bs[1], accR[1][0])   PW_DMA1(kt + P_STAGES, st, 6)
        PW_M_PIN(accR[1][0], F1.a[1].x, F1.b[0].x, accI[1][0]) PW_DMA1(kt + P_STAGES, st, 3)
        PW_M_PIN(accI[1][0], F1.a[1].y, F1.b[0].y, accS[1][0]) PW_DMA1(kt + P_STAGES, st, 7)
        accS[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(F1.as[1], F1.bs[0], accS[1][0], 0, 0, 0);
        PW_MFMA1(F1, 1, 1)
        PW_FENCE()
        st = st_next;
    }
    // nothing may still be on its way into this workgroup's LDS when it ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- combine (see the header) and store write-through: the second product reads PW from every XCD
    const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int gi = i0 + wm * 32 + mi * 16 + q4 + 4 * reg;
                const int gj = j0 + wn * 32 + ni * 16 + r16;
                const double cre = (-accR[mi][ni][reg]) - accI[mi][ni][reg];
                const double cim = (accS[mi][ni][reg] + accR[mi][ni][reg]) - accI[mi][ni][reg];
                const cplx v = make_double2(cre, cim);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4u *>(&v), rsrcC,
                                                       (unsigned)(((size_t)gi * N + gj) * sizeof(cplx)), 0, 16);
            }
#undef PW_DMA
#undef PW_READ
#undef PW_LANDED
#undef PW_SUMS
#undef PW_MFMA1
#undef PW_M_PIN
#undef PW_DMA1
#undef PW_FENCE
}

}  // namespace

// PW = P @ B for an exactly skew-Hermitian P (the stepper's Phalf); falls back to the general kernel where
// the 64 x 64 DMA form does not apply (N % 64 != 0, N < 768) or when QUFLOW_HIP_GEMM1_DMA=0.
int qf_launch_zgemm_skew_left(qf_ctx *ctx, const cplx *P, const cplx *B, cplx *C, qf_guard guard)
{
    static const int enabled = [] {
        const char *e = getenv("QUFLOW_HIP_GEMM1_DMA");
        return (e && e[0] == '0') ? 0 : 1;
    }();
    const int N = ctx->N;
    if (!enabled || N % PT != 0 || N < 768 || !ctx->gemm_3m) return qf_launch_zgemm(ctx, P, B, C, nullptr, guard);
    static bool attr_set = false;
    if (!attr_set) {
        QF_HIP(hipFuncSetAttribute((const void *)k_zgemm_pw_dma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PW_SMEM_BYTES));
        attr_set = true;
    }
    const int nt = N / PT;
    hipLaunchKernelGGL(k_zgemm_pw_dma, dim3(nt * nt), dim3(256), PW_SMEM_BYTES, ctx->stream, N, P, B, C, guard);
    QF_HIP(hipGetLastError());
    return QF_OK;
}
