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
// What the hardware dictates (measured with tools/mfma_clock.hip on MI355X):
//   * v_mfma_f64_16x16x4_f64 issues every 64 cycles per SIMD (77.5 TFLOP/s chip-wide at
//     2.39 GHz), also on a dependent accumulator chain and with C/D in VGPRs;
//   * it runs on the fp64 VALU datapath: VALU instructions of the same wave do NOT overlap
//     with it (4 v_fma_f64 between MFMAs: 64 -> 96 cycles).  Every VALU instruction in the
//     K loop is pure loss, so the loop below has none: all LDS addresses are
//     per-thread bases + immediates, global addresses are SGPR bases + one fixed VGPR
//     offset, and -Im(a) comes from the MFMA's neg modifier (blgp bit 0), not from v_xor.
//
// Design
//   * operands stay interleaved (re,im): one ds_read_b128 gives a lane the complex entry
//     whose halves are the two f64 MFMA operands it needs.
//   * one complex MAC tile = 3 real MFMAs (Karatsuba/"3M": T1 = ar*br, T2 = ai*bi,
//     T3 = (ar+ai)(br+bi);  Re = T1 - T2, Im = T3 - T1 - T2): 6 N^3 executed flops for the
//     8 N^3 of a complex product.  The sums (ar+ai), (br+bi) are formed once per staged entry
//     and kept in a second LDS plane.  (A 4-MFMA form existed until round 5: slower, removed -- DESIGN.md 3.1.)
//   * block tile BM x BN (complex), BK = 16, K-tiles double-buffered in LDS; A is staged
//     k-major (transposed) with one complex of row padding so that the transposing
//     ds_write_b128 and both fragment ds_read_b128 patterns are conflict-free.
//   * software pipeline (one wave per SIMD keeps the matrix pipe busy by itself):
//       - MFMA fragments are double-buffered in registers: phase k4 issues the ds_reads of
//         phase k4+1 before its own 4*MT*NT MFMAs;
//       - K-tiles kt+2 and kt+3 travel L2 -> registers (two register sets: more than two
//         K-tiles of latency budget, enough for Infinity-Cache misses) while K-tile kt+1 is
//         written to the other LDS buffer, one memory instruction per MFMA gap of phase 1;
//       - the single barrier per K-tile sits between phases 2 and 3 (raw s_barrier with
//         lgkmcnt(0) only: global loads in flight are not drained) and phase 3 already
//         prefetches the first fragments of the next buffer.
//   * MFMA f64 16x16x4 lane maps (cdna_hip_programming.md section 3):
//        A[i = lane&15][k = lane>>4],  B[k = lane>>4][j = lane&15],
//        C[row = (lane>>4) + 4*reg][col = lane&15].
#include "qf_internal.h"
#include "qf_step_end.h"

#include <cstring>

typedef double v4d __attribute__((ext_vector_type(4)));

// Diagnostic builds only (tools/zgemm_probe.hip defines QF_STAMP): per-wave s_memtime stamps
// after every K-tile go to a side buffer that nothing else reads.  No stamp executes in the
// shipped library.
#ifdef QF_STAMP
__device__ unsigned long long *qf_stamp_buf = nullptr;   // [blocks*waves][QF_STAMP_SLOTS]
#define QF_STAMP_SLOTS 80
#ifdef QF_STAMP_LIGHT
// segment-level stamps of k_zgemm_tri only (tools/tri_probe.hip -DQF_STAMP_LIGHT): nothing inside the K loop,
// so the timeline is the shipped kernel's to within a few store instructions per segment
#define QF_STAMP_AT(slot_)
#define QF_STAMP_PH(kt_, ph_)
__device__ unsigned long long *qf_phase_buf = nullptr;
#else
#define QF_STAMP_AT(slot_)                                                              \
    if (qf_stamp_buf && lane == 0 && (slot_) < QF_STAMP_SLOTS)                          \
        qf_stamp_buf[((size_t)blockIdx.x * (T / 64) + wave) * QF_STAMP_SLOTS + (slot_)] = __builtin_amdgcn_s_memtime();
// per-phase stamps of K-tiles 20..27 (wave 0 of every block)
__device__ unsigned long long *qf_phase_buf = nullptr;   // [blocks][8 tiles][5]
#define QF_STAMP_PH(kt_, ph_)                                                           \
    if (qf_phase_buf && tid == 0 && (kt_) >= 20 && (kt_) < 28)                          \
        qf_phase_buf[((size_t)blockIdx.x * 8 + ((kt_) - 20)) * 5 + (ph_)] = __builtin_amdgcn_s_memtime();
#endif
// k_zgemm_tri: per workgroup and segment, stamps of (start, K loop done, published / pieces
// gathered, epilogue done) + the segment's (k0, KT)
__device__ unsigned long long *qf_tri_buf = nullptr;     // [blocks][4 segments][8]
#define QF_TRI_STAMP(seg_, k_)                                                          \
    if (qf_tri_buf && tid == 0 && (seg_) < 4)                                           \
        qf_tri_buf[((size_t)blockIdx.x * 4 + (seg_)) * 8 + (k_)] = __builtin_amdgcn_s_memtime();
#define QF_TRI_NOTE(seg_, k_, v_)                                                       \
    if (qf_tri_buf && tid == 0 && (seg_) < 4) qf_tri_buf[((size_t)blockIdx.x * 4 + (seg_)) * 8 + (k_)] = (unsigned long long)(v_);
#else
#define QF_STAMP_AT(slot_)
#define QF_STAMP_PH(kt_, ph_)
#define QF_TRI_STAMP(seg_, k_)
#define QF_TRI_NOTE(seg_, k_, v_)
#endif
// timing-only ablation knobs of the diagnostic build (results are wrong when set)
#ifndef QF_ABL_NOSTORE
#define QF_ABL_NOSTORE 0    // skip the LDS staging writes
#endif
#ifndef QF_ABL_NOSUMS
#define QF_ABL_NOSUMS 0     // skip the re+im plane writes (3M)
#endif
#ifndef QF_ABL_NOADD
#define QF_ABL_NOADD 0      // the re+im planes get re alone: the K loop without its v_add_f64 (timing only)
#endif
#define QF_SUM3M(c_) (QF_ABL_NOADD ? (c_).x : (c_).x + (c_).y)
#ifndef QF_ABL_NOGLOAD
#define QF_ABL_NOGLOAD 0    // skip the global loads of the K loop
#endif
#ifndef QF_ABL_NOBARRIER
#define QF_ABL_NOBARRIER 0  // drop the per-K-tile barrier
#endif
// cache policy of the stream-K exchange (timing experiments only: 16 = sc1, through to / from memory -- what pieces
// that cross XCDs need)
#ifndef QF_SK_AUX_ST
#define QF_SK_AUX_ST 16
#endif
#ifndef QF_SK_AUX_LD
#define QF_SK_AUX_LD 16
#endif
#ifndef QF_XCD_BLOCK
#define QF_XCD_BLOCK 1       // full-product kernels on square grids of a multiple of 16 tiles: 4 x 8 tile blocks per XCD instead of
                             // contiguous row-major ranges (round 4, same box: N = 2048 first product 754 -> 747 us, 401.2 -> 404.3
                             // timesteps/s; N = 1024 98.8 us either way)
#endif
#ifndef QF_POLL_SLEEP
#define QF_POLL_SLEEP 8      // s_sleep argument between two looks at a piece's flag (x 64 cycles)
#endif
#ifndef QF_TAIL_LITERAL
#define QF_TAIL_LITERAL 1    // the steady K-tile to the end of a K range (0: run-time flags in the last four K-tiles)
#endif
#ifndef QF_TAIL_LITERAL32
#define QF_TAIL_LITERAL32 1  // ... also in the exact 32 x 32 tilings (0: only the 64 x 64 buffer-load kernels)
#endif
#ifndef QF_STAGE_SPREAD
#define QF_STAGE_SPREAD 1    // LDS staging stores spread over phases 0-1, global loads in phase 2 (0: all in phase 1)
#endif

// QF_WT_PW = 1: the first product stores PW write-through (`sc1`).  The second product reads PW from every
// XCD, so its lines have to reach memory anyway; written through they leave while the kernel still runs
// instead of as 16 MiB of dirty L2 lines at the release of the kernel boundary: 99.6 -> 98.0 us per launch at
// N = 1024 (the same change to the second product's and the solve's stores measured nothing).
#ifndef QF_WT_PW
#define QF_WT_PW 1
#endif

namespace {

constexpr int BK = 16;

// LDS image of the A tile: k-major (transposed), row stride BM complex with column c of k-row k
// stored at slot c ^ (k & 7).  ds_read_b128 is served in four groups of 16 lanes
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ... -- MI355X_MICROARCH.md, LDS) over 64 banks, i.e.
// a group holds two k-rows with 8 columns each: with a stride of 0 mod 16 slots and the XOR
// the 16 slots are distinct; the transposing ds_write_b128 (8 contiguous lanes = 8 consecutive
// k, 32 banks = 8 slots) is conflict-free because k & 7 is.  (The first version padded the rows
// by one entry instead: fine for the writes, but every fragment read had one two-way conflict
// per lane group -- 20 % of all LDS cycles, SQ_LDS_BANK_CONFLICT in profiles/r01_pmc_summary.txt.)
template <int BM, int BN, bool PAIR = false>
struct tile_smem {
    static constexpr int A_STRIDE = BM;      // complex entries per k-row of the transposed A tile
    static constexpr int B_STRIDE = BN;
    static constexpr int A_BUF_BYTES = BK * A_STRIDE * (int)sizeof(cplx);
    static constexpr int B_BUF_BYTES = BK * B_STRIDE * (int)sizeof(cplx);
    static constexpr int B_OFFSET = 2 * A_BUF_BYTES;
    // planes of re+im (3M), same [buffer][k][column] shape, one double per entry
    // doubles per k-row of the A sum plane.  Pair layout of the exact 64x64 tilings (entries i and
    // i+16 adjacent, one 16-byte slot): unpadded and swizzled like the A tile; generic [k][column]
    // layout: padded by two doubles
    static constexpr int A3_STRIDE = PAIR ? BM : BM + 2;
    static constexpr int B3_STRIDE = BN;
    static constexpr int A3_BUF_BYTES = BK * A3_STRIDE * (int)sizeof(double);
    static constexpr int B3_BUF_BYTES = BK * B3_STRIDE * (int)sizeof(double);
    static constexpr int A3_OFFSET = 2 * (A_BUF_BYTES + B_BUF_BYTES);
    static constexpr int B3_OFFSET = A3_OFFSET + 2 * A3_BUF_BYTES;
    static constexpr size_t main_bytes = (size_t)2 * (A_BUF_BYTES + B_BUF_BYTES) + (size_t)2 * (A3_BUF_BYTES + B3_BUF_BYTES);
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

template <int BM, int BN, int WM, int WN, bool EPI, bool EXACT, bool FUSED = false>
__global__ __launch_bounds__(WM *WN * 64) void k_zgemm(int N, int tiles_m, int tiles_n,
                                                        const cplx *__restrict__ A,
                                                        const cplx *__restrict__ B, cplx *__restrict__ C,
                                                        qf_epilogue ep, qf_guard guard)
{
    // stepper launches are tagged (step, iteration): no-op unless the device state says this iteration is
    // due (uniform scalar loads; see qf_internal.h).  The tag is looked at BELOW, behind the requests for A's
    // first two K-tiles: the state lives in memory another XCD wrote last, its scalar load is a full memory
    // round trip, and the first operand tiles can travel during it (a launch that is not due drops them).
    constexpr int T = WM * WN * 64;
    // FAST: exact 64x64 tilings of the 3M kernel get a K loop without address VALU at all --
    // buffer loads (descriptor + fixed VGPR offset + SGPR offset advanced by SALU), and sum
    // planes laid out in (i, i+16) pairs so that the two sums a wave needs (and the two a
    // staging thread produces) are one 16-byte LDS access with an immediate offset.
    constexpr bool FAST = EXACT && BM == 64 && BN == 64 && T == 256;
    using SM = tile_smem<BM, BN, FAST>;
    constexpr int WTM = BM / WM, WTN = BN / WN;  // wave tile
    constexpr int MT = WTM / 16, NT = WTN / 16;  // MFMA tiles per wave
    constexpr int A_STRIDE = SM::A_STRIDE, B_STRIDE = SM::B_STRIDE;
    constexpr int A_PER = (BM * BK) / T;  // complex entries each thread stages per K-tile
    constexpr int B_PER = (BN * BK) / T;
    constexpr int A_ROWS_PER = T / BK;    // A rows covered by one staging pass of the block
    constexpr int B_ROWS_PER = T / BN;    // B k-rows covered by one staging pass
    static_assert((BM * BK) % T == 0 && (BN * BK) % T == 0 && T % BK == 0 && T % BN == 0, "tile/threads mismatch");
    constexpr int A3_STRIDE = SM::A3_STRIDE, B3_STRIDE = SM::B3_STRIDE;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, q4 = lane >> 4;

    // workgroup -> tile.  Default: each XCD a contiguous row-major range of tiles (two tile rows at N = 1024: 2 A panels
    // and all 16 B panels per L2).  QF_XCD_BLOCK (tile rows a multiple of 4, columns of 2): XCD x works on a compact
    // (tiles/4) x (tiles/2) part of the grid, in 4 x 8 blocks where that divides -- 4 A panels and 8 B panels per L2 at a time.
    int tm, tn;
    if (QF_XCD_BLOCK && tiles_m % 4 == 0 && tiles_n % 2 == 0) {
        // XCD x = (x >> 1, x & 1) of a 4 x 2 arrangement owns the tile rows [(x>>1) R, +R) and columns [(x&1) C, +C): equal
        // shares, so the round-robin deal of workgroup ids gives every XCD exactly its region; inside it 4 x 8 blocks where
        // the region's edges allow, row by row otherwise
        const int x = blockIdx.x & 7, l = blockIdx.x >> 3;
        const int R = tiles_m / 4, C = tiles_n / 2;
        int lr, lc;
        if (R % 4 == 0 && C % 8 == 0) {
            const int blk = l >> 5, in = l & 31;
            const int bpr = C / 8;
            lr = (blk / bpr) * 4 + (in >> 3);
            lc = (blk % bpr) * 8 + (in & 7);
        } else {
            lr = l / C;
            lc = l % C;
        }
        tm = (x >> 1) * R + lr;
        tn = (x & 1) * C + lc;
    } else {
        const int lid0 = xcd_remap(blockIdx.x, tiles_m * tiles_n);
        tm = lid0 / tiles_n;
        tn = lid0 % tiles_n;
    }
    const int lid = tm * tiles_n + tn;
    const int i0 = tm * BM, j0 = tn * BN;
    const cplx zero = make_double2(0.0, 0.0);
    const int parity = (EPI && guard.state) ? guard.state->dw_parity : 0;
    const cplx *__restrict__ ep_dW_old = ep.dW[parity];
    cplx *__restrict__ ep_dW_new = ep.dW[parity ^ 1];
    // fused step end: the state is Wpair[w_parity]; the candidate next state
    // W + 2 (PW - PW^H) goes to the other buffer of the pair
    const int wpar = (FUSED && guard.state) ? guard.state->w_parity : 0;
    const cplx *__restrict__ ep_W = FUSED ? (wpar ? ep.Wpair[1] : ep.Wpair[0]) : ep.W;
    cplx *__restrict__ ep_Wnext = FUSED ? (wpar ? ep.Wpair[0] : ep.Wpair[1]) : nullptr;

    // ---- per-thread LDS bases; everything else in the K loop is an immediate offset
    // A tile: column c of k-row k sits at slot c ^ (k & 7) (tile_smem).  A fragment lane reads
    // k = 4*K4 + q4, so its swizzle is q4 ^ 4*(K4 & 1): one base for even K4, one for odd
    // (lds_fa[K4 & 1]); the staging thread writes k = tid % BK.
    const unsigned char *lds_fa[2] = {
        smem_raw + (size_t)(q4 * A_STRIDE + wm * WTM + (r16 ^ q4)) * sizeof(cplx),
        smem_raw + (size_t)(q4 * A_STRIDE + wm * WTM + (r16 ^ q4 ^ 4)) * sizeof(cplx)};
    const unsigned char *lds_fb = smem_raw + SM::B_OFFSET + (size_t)(q4 * B_STRIDE + wn * WTN + r16) * sizeof(cplx);
    unsigned char *lds_sa = smem_raw + (size_t)((tid % BK) * A_STRIDE + ((tid / BK) ^ (tid % BK & 7))) * sizeof(cplx);
    unsigned char *lds_sb = smem_raw + SM::B_OFFSET + (size_t)tid * sizeof(cplx);
    // sum planes: generic layout [k][column]; FAST layout [k][pair]: entries i and i+16 adjacent in
    // one 16-byte slot, slots swizzled like the A tile's
    const unsigned char *lds_fa3[2] = {
        smem_raw + SM::A3_OFFSET + (size_t)(q4 * A3_STRIDE + (FAST ? wm * WTM + 2 * (r16 ^ q4) : wm * WTM + r16)) * sizeof(double),
        smem_raw + SM::A3_OFFSET + (size_t)(q4 * A3_STRIDE + (FAST ? wm * WTM + 2 * (r16 ^ q4 ^ 4) : wm * WTM + r16)) * sizeof(double)};
    const unsigned char *lds_fb3 = smem_raw + SM::B3_OFFSET +
        (size_t)(q4 * B3_STRIDE + (FAST ? wn * WTN + 2 * r16 : wn * WTN + r16)) * sizeof(double);
    unsigned char *lds_sa3 = smem_raw + SM::A3_OFFSET +
        (size_t)((tid % BK) * A3_STRIDE + (FAST ? 2 * ((tid / BK) ^ (tid % BK & 7)) : tid / BK)) * sizeof(double);
    // FAST B staging map: thread -> k-rows tid/32 and tid/32 + 8, columns jA and jA + 16
    const int b_jA = ((tid % 32) / 16) * 32 + (tid % 16);
    unsigned char *lds_sb3 = smem_raw + SM::B3_OFFSET +
        (size_t)(FAST ? (tid / 32) * B3_STRIDE + 32 * ((tid % 32) / 16) + 2 * (tid % 16) : tid) * sizeof(double);
    if (FAST) lds_sb = smem_raw + SM::B_OFFSET + (size_t)((tid / 32) * B_STRIDE + b_jA) * sizeof(cplx);

    // ---- global staging addresses: uniform (SGPR) row bases + one fixed per-thread offset
    // A entry (i0 + tid/BK + r*A_ROWS_PER, k0 + tid%BK);  B entry (k0 + tid/BN + r*B_ROWS_PER, j0 + tid%BN)
    const unsigned a_voff = (unsigned)(((size_t)(tid / BK) * N + (tid % BK)) * sizeof(cplx));
    const unsigned b_voff = (unsigned)(((size_t)(tid / BN) * N + (tid % BN)) * sizeof(cplx));
    const int kt_off = 0;                                          // (first K-tile of this workgroup's range, for the generic arms' bounds)
    const unsigned char *a_row = reinterpret_cast<const unsigned char *>(A) + (size_t)i0 * N * sizeof(cplx);
    const size_t a_pass = (size_t)A_ROWS_PER * N * sizeof(cplx);   // bytes between staging passes of A
    const size_t b_pass = (size_t)B_ROWS_PER * N * sizeof(cplx);
    const size_t b_ktile = (size_t)BK * N * sizeof(cplx);          // B advances BK rows per K-tile
    // FAST: buffer descriptors over the whole operands; voffset fixed per thread, soffset uniform
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cplx *>(A), 0, FAST ? (int)((size_t)N * N * sizeof(cplx)) : 0, 0x00020000);
    const unsigned fa_voff = (unsigned)(((size_t)(tid / BK) * N + (tid % BK)) * sizeof(cplx));
    const unsigned fb_voff = (unsigned)(((size_t)(tid / 32) * N + b_jA) * sizeof(cplx));
    const unsigned fa_soff0 = (unsigned)((size_t)i0 * N * sizeof(cplx));   // + 16 rows per pass, + BK cols per K-tile
    const unsigned fb_soff0 = (unsigned)((size_t)j0 * sizeof(cplx));       // + 8 rows per pass, + BK rows per K-tile
    const unsigned f_rows16 = (unsigned)((size_t)16 * N * sizeof(cplx));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));

    // ---- epilogue operands of the second product are prefetched into registers DURING the
    // main loop (one wave per SIMD owns all 512 VGPRs): K-tiles 1..4 each fetch one operand
    // tile, so the loads neither delay the first K-tiles (vmcnt retires in order) nor show up
    // in the epilogue, which becomes register arithmetic + two streaming stores.
    // e_c: PW[i,j], then the commutator PW[i,j] - conj(PW[j,i]) once the mirrored entry e_t
    // has arrived; e_w: W[i,j]; e_old: dW_old[i,j]  (at most three tiles live: 192 VGPRs).
    cplx e_c[EPI ? MT : 1][EPI ? NT : 1][4], e_t[EPI ? MT : 1][EPI ? NT : 1][4];
    cplx e_w[EPI ? MT : 1][EPI ? NT : 1][4], e_old[EPI ? MT : 1][EPI ? NT : 1][4];
    // conj_subtract_: PW[i,j] - conj(PW[j,i])   (isospectral.py:71-74)
#define QF_EPI_COMM                                                                    \
    {                                                                                  \
        _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                              \
            _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                          \
                _Pragma("unroll") for (int reg = 0; reg < 4; ++reg)                    \
        {                                                                              \
            e_c[mi][ni][reg].x = e_c[mi][ni][reg].x - e_t[mi][ni][reg].x;              \
            e_c[mi][ni][reg].y = e_c[mi][ni][reg].y + e_t[mi][ni][reg].y;              \
        }                                                                              \
    }
#define QF_EPI_FETCH(dst_, src_, TRANSPOSED_)                                          \
    {                                                                                  \
        _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                              \
            _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                          \
                _Pragma("unroll") for (int reg = 0; reg < 4; ++reg)                    \
        {                                                                              \
            const int gi = i0 + wm * WTM + mi * 16 + q4 + 4 * reg;                     \
            const int gj = j0 + wn * WTN + ni * 16 + r16;                              \
            dst_[mi][ni][reg] = zero;                                                  \
            if (EXACT || (gi < N && gj < N))                                           \
                dst_[mi][ni][reg] = (TRANSPOSED_) ? (src_)[(size_t)gj * N + gi] : (src_)[(size_t)gi * N + gj]; \
        }                                                                              \
    }

    // accR = T1 = sum ar*br, accI = T2 = sum ai*bi, accS = T3 = sum (ar+ai)(br+bi); combined after the K loop.
    v4d accR[MT][NT], accI[MT][NT], accS[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            accR[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            accI[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            accS[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        }

    cplx ra[2][A_PER], rb[2][B_PER];   // two K-tiles in flight L2 -> registers -> LDS (set = K-tile parity)
    cplx fa[2][MT], fb[2][NT];      // double-buffered MFMA fragments
    double fas[2][MT], fbs[2][NT];   // re+im of the fragments

    // All helpers are macros on purpose: lambdas capturing the register arrays by reference
    // made hipcc keep them in scratch memory.
    // Load K-tile number kt_ (k0 = kt_*BK) from global memory into ra/rb.
    // (exact tilings: a K-tile index past the K range is clamped to its last K-tile -- see QF_KLOOP_TAIL)
#define QF_TAIL_STEADY_HERE (QF_TAIL_LITERAL && EXACT && (FAST || QF_TAIL_LITERAL32))
#define QF_KT_CLAMP(kt_) ((QF_TAIL_STEADY_HERE && (kt_) >= KT) ? KT - 1 : (kt_))
#define QF_LOAD_TILE_A(kt_, SET_)                                                      \
    if (FAST) {                                                                        \
        const unsigned sa = fa_soff0 + (unsigned)QF_KT_CLAMP(kt_) * (unsigned)(BK * sizeof(cplx)); \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                  \
        {                                                                              \
            const v4u t = __builtin_amdgcn_raw_buffer_load_b128(rsrcA, fa_voff, sa + r * f_rows16, 0); \
            ra[SET_][r] = *reinterpret_cast<const cplx *>(&t);                         \
        }                                                                              \
    } else {                                                                           \
        const unsigned char *ap = a_row + (size_t)QF_KT_CLAMP(kt_) * (BK * sizeof(cplx)); \
        _Pragma("unroll") for (int r = 0; r < A_PER; ++r)                              \
        {                                                                              \
            ra[SET_][r] = zero;                                                        \
            if (EXACT || (i0 + tid / BK + r * A_ROWS_PER < N && ((kt_) + kt_off) * BK + tid % BK < N)) \
                ra[SET_][r] = *reinterpret_cast<const cplx *>((ap + r * a_pass) + a_voff); \
        }                                                                              \
    }
#define QF_LOAD_TILE_B(kt_, SET_)                                                      \
    if (FAST) {                                                                        \
        const unsigned sb = fb_soff0 + (unsigned)QF_KT_CLAMP(kt_) * f_rows16;          \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                  \
        {                                                                              \
            const v4u t = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, fb_voff + (r & 1) * 256u, sb + (r >> 1) * (f_rows16 / 2), 0); \
            rb[SET_][r] = *reinterpret_cast<const cplx *>(&t);                         \
        }                                                                              \
    } else {                                                                           \
        const unsigned char *bp = b_col + (size_t)QF_KT_CLAMP(kt_) * b_ktile; \
        _Pragma("unroll") for (int r = 0; r < B_PER; ++r)                              \
        {                                                                              \
            rb[SET_][r] = zero;                                                        \
            if (EXACT || (((kt_) + kt_off) * BK + tid / BN + r * B_ROWS_PER < N && j0 + tid % BN < N)) \
                rb[SET_][r] = *reinterpret_cast<const cplx *>((bp + r * b_pass) + b_voff); \
        }                                                                              \
    }
#define QF_LOAD_TILE(kt_, SET_)                                                        \
    {                                                                                  \
        QF_LOAD_TILE_A(kt_, SET_)                                                      \
        QF_LOAD_TILE_B(kt_, SET_)                                                      \
    }
    // Write ra / rb into LDS buffer BUF_ (literal 0/1): A transposed to k-major.
#define QF_STORE_A(BUF_, SET_)                                                         \
    if (FAST) {                                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                  \
            *reinterpret_cast<cplx *>(lds_sa + (BUF_) * SM::A_BUF_BYTES + r * 16 * (int)sizeof(cplx)) = ra[SET_][r]; \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                  \
            *reinterpret_cast<double2 *>(lds_sa3 + (BUF_) * SM::A3_BUF_BYTES + h * 32 * (int)sizeof(double)) =             \
                make_double2(QF_SUM3M(ra[SET_][2 * h]), QF_SUM3M(ra[SET_][2 * h + 1]));       \
    } else {                                                                           \
        _Pragma("unroll") for (int r = 0; r < A_PER; ++r)                              \
            *reinterpret_cast<cplx *>(lds_sa + (BUF_) * SM::A_BUF_BYTES + r * A_ROWS_PER * (int)sizeof(cplx)) = ra[SET_][r]; \
        if (!QF_ABL_NOSUMS) {                                                    \
            _Pragma("unroll") for (int r = 0; r < A_PER; ++r)                          \
                *reinterpret_cast<double *>(lds_sa3 + (BUF_) * SM::A3_BUF_BYTES + r * A_ROWS_PER * (int)sizeof(double)) = ra[SET_][r].x + ra[SET_][r].y; \
        }                                                                              \
    }
#define QF_STORE_B(BUF_, SET_)                                                         \
    if (FAST) {                                                                        \
        _Pragma("unroll") for (int r = 0; r < 4; ++r)                                  \
            *reinterpret_cast<cplx *>(lds_sb + (BUF_) * SM::B_BUF_BYTES + ((r & 1) * 16 + (r >> 1) * 8 * B_STRIDE) * (int)sizeof(cplx)) = rb[SET_][r]; \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                  \
            *reinterpret_cast<double2 *>(lds_sb3 + (BUF_) * SM::B3_BUF_BYTES + h * 8 * B3_STRIDE * (int)sizeof(double)) =  \
                make_double2(QF_SUM3M(rb[SET_][2 * h]), QF_SUM3M(rb[SET_][2 * h + 1]));       \
    } else {                                                                           \
        _Pragma("unroll") for (int r = 0; r < B_PER; ++r)                              \
            *reinterpret_cast<cplx *>(lds_sb + (BUF_) * SM::B_BUF_BYTES + r * B_ROWS_PER * B_STRIDE * (int)sizeof(cplx)) = rb[SET_][r]; \
        if (!QF_ABL_NOSUMS) {                                                    \
            _Pragma("unroll") for (int r = 0; r < B_PER; ++r)                          \
                *reinterpret_cast<double *>(lds_sb3 + (BUF_) * SM::B3_BUF_BYTES + r * B_ROWS_PER * B_STRIDE * (int)sizeof(double)) = rb[SET_][r].x + rb[SET_][r].y; \
        }                                                                              \
    }
#define QF_STORE_TILE(BUF_, SET_)                                                      \
    {                                                                                  \
        QF_STORE_A(BUF_, SET_)                                                         \
        QF_STORE_B(BUF_, SET_)                                                         \
    }
#define QF_READ_FRAGS(SET_, BUF_, K4_)                                                 \
    {                                                                                  \
        _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                              \
            fa[SET_][mi] = *reinterpret_cast<const cplx *>(                            \
                lds_fa[(K4_) & 1] + (BUF_) * SM::A_BUF_BYTES + ((K4_) * 4 * A_STRIDE + mi * 16) * (int)sizeof(cplx)); \
        _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                              \
            fb[SET_][ni] = *reinterpret_cast<const cplx *>(                            \
                lds_fb + (BUF_) * SM::B_BUF_BYTES + ((K4_) * 4 * B_STRIDE + ni * 16) * (int)sizeof(cplx)); \
        if (FAST) {                                                                    \
            const double2 sa2 = *reinterpret_cast<const double2 *>(                    \
                lds_fa3[(K4_) & 1] + (BUF_) * SM::A3_BUF_BYTES + (K4_) * 4 * A3_STRIDE * (int)sizeof(double)); \
            const double2 sb2 = *reinterpret_cast<const double2 *>(                    \
                lds_fb3 + (BUF_) * SM::B3_BUF_BYTES + (K4_) * 4 * B3_STRIDE * (int)sizeof(double)); \
            fas[SET_][0] = sa2.x;                                                      \
            fas[SET_][MT - 1] = sa2.y;                                                 \
            fbs[SET_][0] = sb2.x;                                                      \
            fbs[SET_][NT - 1] = sb2.y;                                                 \
        } else {                                                                       \
            _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                          \
                fas[SET_][mi] = *reinterpret_cast<const double *>(                     \
                    lds_fa3[(K4_) & 1] + (BUF_) * SM::A3_BUF_BYTES + ((K4_) * 4 * A3_STRIDE + mi * 16) * (int)sizeof(double)); \
            _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                          \
                fbs[SET_][ni] = *reinterpret_cast<const double *>(                     \
                    lds_fb3 + (BUF_) * SM::B3_BUF_BYTES + ((K4_) * 4 * B3_STRIDE + ni * 16) * (int)sizeof(double)); \
        }                                                                              \
    }
    // 3*MT*NT MFMAs into T1, T2, T3
#define QF_MFMA(SET_)                                                                  \
    {                                                                                  \
        _Pragma("unroll") for (int mi = 0; mi < MT; ++mi)                              \
            _Pragma("unroll") for (int ni = 0; ni < NT; ++ni)                          \
        {                                                                              \
            accR[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[SET_][mi].x, fb[SET_][ni].x, accR[mi][ni], 0, 0, 0); \
            accI[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[SET_][mi].y, fb[SET_][ni].y, accI[mi][ni], 0, 0, 0); \
            accS[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fas[SET_][mi], fbs[SET_][ni], accS[mi][ni], 0, 0, 0); \
        }                                                                              \
    }

    // instruction-class masks of __builtin_amdgcn_sched_group_barrier
    constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_WR = 0x200;
    // One K-tile in LDS buffer BUF_ (literal).  STORE_/LOAD_/NEXT_ are literal 1 in the
    // steady state, so that the body is one basic block and the scheduler can put one staging
    // instruction into each MFMA gap of phase 1; the tail uses run-time conditions.
    // Staging spread over the K-tile.  The LDS store path of a CU takes ~79 B/clk (ds_write_b128,
    // MI355X_MICROARCH.md, LDS) and all four waves stage in the same phase: twelve 1 KiB stores per
    // wave squeezed into one phase ask for 128 B/clk and back up into the wave's instruction issue --
    // the MFMAs behind them wait (ablation: staging stores off -> 18 % faster K loop).  The other LDS
    // buffer is free from the barrier of K-tile kt-1 on, so the A half (4 + 2 stores) goes into
    // phase 0, the B half into phase 1, one store per two MFMA gaps, and the global loads of K-tile
    // kt+3 into phase 2, one per gap.
#define QF_KTILE_SPREAD(kt_, BUF_, STORE_, LOAD_, NEXT_, STEADY_, PREF_)                      \
    {                                                                                  \
        /* phase 0: fetch the fragments of phase 1; stage the A half of K-tile kt+1 */ \
        QF_READ_FRAGS(1, BUF_, 1)                                                      \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if ((STORE_) && !QF_ABL_NOSTORE) { QF_STORE_A((BUF_) ^ 1, (BUF_) ^ 1) }        \
        QF_MFMA(0)                                                                     \
        if (EXACT && (STEADY_)) {                                                \
            _Pragma("unroll") for (int g = 0; g < (A_PER * 3) / 2; ++g)                \
            {                                                                          \
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);                   \
                __builtin_amdgcn_sched_group_barrier(SG_DS_WR, 1, 0);                  \
            }                                                                          \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* phase 1: the B half */                                                      \
        QF_READ_FRAGS(0, BUF_, 2)                                                      \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if ((STORE_) && !QF_ABL_NOSTORE) { QF_STORE_B((BUF_) ^ 1, (BUF_) ^ 1) }        \
        if (EPI && (PREF_) == 1) QF_EPI_FETCH(e_c, ep.PW, false)                       \
        if (EPI && (PREF_) == 2) QF_EPI_FETCH(e_t, ep.PW, true)                        \
        if (EPI && (PREF_) == 3) QF_EPI_COMM                                           \
        if (EPI && (PREF_) == 5) {                                                     \
            QF_EPI_FETCH(e_w, ep_W, false)                                             \
            QF_EPI_FETCH(e_old, ep_dW_old, false)                                      \
        }                                                                              \
        QF_MFMA(1)                                                                     \
        if (EXACT && (STEADY_)) {                                                \
            _Pragma("unroll") for (int g = 0; g < (B_PER * 3) / 2; ++g)                \
            {                                                                          \
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 2, 0);                   \
                __builtin_amdgcn_sched_group_barrier(SG_DS_WR, 1, 0);                  \
            }                                                                          \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* phase 2: the register set is free again: K-tile kt+3 starts its way L2 -> registers */ \
        QF_READ_FRAGS(1, BUF_, 3)                                                      \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if (QF_STAGE_SPREAD != 2 && (LOAD_) && !QF_ABL_NOGLOAD) { QF_LOAD_TILE((kt_) + 3, (BUF_) ^ 1) } \
        QF_MFMA(0)                                                                     \
        if (QF_STAGE_SPREAD != 2 && EXACT && (STEADY_)) {                        \
            _Pragma("unroll") for (int g = 0; g < A_PER + B_PER; ++g)                  \
            {                                                                          \
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);                   \
                __builtin_amdgcn_sched_group_barrier(SG_VMEM_RD, 1, 0);                \
            }                                                                          \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* my LDS reads of this buffer have landed and my writes to the other are done */ \
        if (QF_ABL_NOBARRIER) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* phase 3: first fragments of the next K-tile */                              \
        if (NEXT_) QF_READ_FRAGS(0, (BUF_) ^ 1, 0)                                     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if (QF_STAGE_SPREAD == 2 && (LOAD_) && !QF_ABL_NOGLOAD) { QF_LOAD_TILE((kt_) + 3, (BUF_) ^ 1) } \
        QF_MFMA(1)                                                                     \
        if (QF_STAGE_SPREAD == 2 && EXACT && (STEADY_)) {                        \
            _Pragma("unroll") for (int g = 0; g < A_PER + B_PER; ++g)                  \
            {                                                                          \
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);                   \
                __builtin_amdgcn_sched_group_barrier(SG_VMEM_RD, 1, 0);                \
            }                                                                          \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* lgkmcnt(0) only (0xC07F): by now the prefetch has landed; stating it keeps  \
           hipcc from waiting conservatively at the loop head, across the back edge */ \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                            \
        __builtin_amdgcn_sched_barrier(0);                                             \
        QF_STAMP_AT((kt_) + 2)                                                         \
    }
#define QF_KTILE_PHASE1(kt_, BUF_, STORE_, LOAD_, NEXT_, STEADY_, PREF_)                      \
    {                                                                                  \
        /* phase 0: fetch the fragments of phase 1, then multiply */                   \
        QF_STAMP_PH(kt_, 0)                                                            \
        QF_READ_FRAGS(1, BUF_, 1)                                                      \
        __builtin_amdgcn_sched_barrier(0);                                             \
        QF_MFMA(0)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* phase 1: the other LDS buffer is free (every wave passed the barrier of    \
           K-tile kt-1): write K-tile kt+1 into it (it arrived in registers two       \
           K-tiles ago) and start fetching K-tile kt+3 into the freed register set */  \
        QF_STAMP_PH(kt_, 1)                                                            \
        QF_READ_FRAGS(0, BUF_, 2)                                                      \
        __builtin_amdgcn_sched_barrier(0);                                             \
        if ((STORE_) && !QF_ABL_NOSTORE) QF_STORE_TILE((BUF_) ^ 1, (BUF_) ^ 1)         \
        if ((LOAD_) && !QF_ABL_NOGLOAD) QF_LOAD_TILE((kt_) + 3, (BUF_) ^ 1)            \
        if (EPI && (PREF_) == 1) QF_EPI_FETCH(e_c, ep.PW, false)                       \
        if (EPI && (PREF_) == 2) QF_EPI_FETCH(e_t, ep.PW, true)                        \
        if (EPI && (PREF_) == 3) QF_EPI_COMM                                           \
        if (EPI && (PREF_) == 5) {                                                     \
            QF_EPI_FETCH(e_w, ep_W, false)                                             \
            QF_EPI_FETCH(e_old, ep_dW_old, false)                                      \
        }                                                                              \
        QF_MFMA(1)                                                                     \
        if (EXACT && (STEADY_)) {                                                \
            /* 3*MT*NT MFMAs; 2*(A_PER+B_PER) LDS writes, A_PER+B_PER global loads */  \
            _Pragma("unroll") for (int g = 0; g < (A_PER + B_PER); ++g)                \
            {                                                                          \
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);                   \
                __builtin_amdgcn_sched_group_barrier(SG_DS_WR, 2, 0);                  \
            }                                                                          \
            _Pragma("unroll") for (int g = 0; g < (A_PER + B_PER) / 2; ++g)            \
            {                                                                          \
                __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);                   \
                __builtin_amdgcn_sched_group_barrier(SG_VMEM_RD, 2, 0);                \
            }                                                                          \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* phase 2 */                                                                  \
        QF_STAMP_PH(kt_, 2)                                                            \
        QF_READ_FRAGS(1, BUF_, 3)                                                      \
        __builtin_amdgcn_sched_barrier(0);                                             \
        QF_MFMA(0)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* my LDS reads of this buffer have landed and my writes to the other are done */ \
        if (QF_ABL_NOBARRIER) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");           \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* phase 3: first fragments of the next K-tile */                              \
        QF_STAMP_PH(kt_, 3)                                                            \
        if (NEXT_) QF_READ_FRAGS(0, (BUF_) ^ 1, 0)                                     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        QF_MFMA(1)                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        /* lgkmcnt(0) only (0xC07F): by now the prefetch has landed; stating it keeps  \
           hipcc from waiting conservatively at the loop head, across the back edge */ \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                            \
        __builtin_amdgcn_sched_barrier(0);                                             \
        QF_STAMP_PH(kt_, 4)                                                            \
        QF_STAMP_AT((kt_) + 2)                                                         \
    }
    // exact 64x64 tilings (FAST) stage spread over the K-tile; the small-tile kernels (N < 768: two
    // staging stores per operand and thread) keep everything in phase 1 (measured: N=512 first product
    // 17.5 us against 19.1 us spread; N=1024 97.4 against 102.6 us the other way round)
#define QF_KTILE(kt_, BUF_, STORE_, LOAD_, NEXT_, STEADY_, PREF_)                      \
    {                                                                                  \
        if (FAST && QF_STAGE_SPREAD) QF_KTILE_SPREAD(kt_, BUF_, STORE_, LOAD_, NEXT_, STEADY_, PREF_) \
        else QF_KTILE_PHASE1(kt_, BUF_, STORE_, LOAD_, NEXT_, STEADY_, PREF_)          \
    }
#define QF_KTILE_STEADY(kt_, BUF_, PREF_) QF_KTILE(kt_, BUF_, 1, 1, 1, 1, PREF_)
#define QF_KTILE_TAIL(kt_, BUF_) QF_KTILE(kt_, BUF_, ((kt_) + 1 < KT), ((kt_) + 3 < KT), ((kt_) + 1 < KT), 0, 0)
#define QF_KTILE_LAST(kt_, BUF_) QF_KTILE(kt_, BUF_, 0, 0, 0, 0, 5)
    // The last K-tiles of a K range (round 4).  QF_KTILE_TAIL's run-time flags split a K-tile into basic blocks, so
    // its staging instructions are not placed in the MFMA gaps -- four such K-tiles per K range are 6 % of the first
    // product's K loop but 25 % of a stream-K workgroup's (two ranges of ~17 K-tiles).  The exact tilings instead run
    // the STEADY K-tile to the very end: the fetch of K-tile kt+3 is CLAMPED to the range's last K-tile (a re-read that
    // hits in the L2; a scalar min), the staging of a K-tile kt+1 that does not exist writes stale registers into the
    // LDS buffer nobody multiplies from any more, and the last K-tile's fragment prefetch reads that buffer into
    // registers nothing uses (k_zgemm_tri zero-fills its staging registers once, so that a one-K-tile first segment
    // stages defined values) -- no new code, one basic block per K-tile throughout.  (Literal NOLOAD / END variants of the K-tile were built first: six more
    // copies of the K-tile body cost k_zgemm_tri 32 bytes of scratch.)
#define QF_KLOOP_TAIL(kt_)                                                             \
    {                                                                                  \
        if (QF_TAIL_STEADY_HERE && (kt_) < KT) {    /* (the steady loop ran in pairs up to here: at most one left) */ \
            QF_KTILE_STEADY(kt_, 0, 0)                                                 \
            ++(kt_);                                                                   \
        }                                                                              \
        for (; (kt_) < KT; ++(kt_)) {                                                  \
            if ((kt_) & 1) { QF_KTILE_TAIL(kt_, 1) } else { QF_KTILE_TAIL(kt_, 0) }    \
        }                                                                              \
    }

    const int KT = (N + BK - 1) / BK;
    QF_STAMP_AT(0)
    QF_LOAD_TILE_A(0, 0)
    if (KT > 1) { QF_LOAD_TILE_A(1, 1) }
    if (!qf_guard_iter(guard)) return;
    // fused step end: the first product of a step's first iteration takes the Whalf prepared for it
    if (!EPI && guard.alt && guard.state->wh_sel) B = static_cast<const cplx *>(guard.alt);
    const unsigned char *b_col = reinterpret_cast<const unsigned char *>(B) + (size_t)j0 * sizeof(cplx);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<cplx *>(B), 0, FAST ? (int)((size_t)N * N * sizeof(cplx)) : 0, 0x00020000);
    QF_LOAD_TILE_B(0, 0)
    if (KT > 1) { QF_LOAD_TILE_B(1, 1) }   // (K-tile 1 on its way before K-tile 0 is staged: phase 0 of K-tile 0 writes it)
    QF_STORE_TILE(0, 0)
    __syncthreads();
    if (KT > 2) QF_LOAD_TILE(2, 0)
    QF_READ_FRAGS(0, 0, 0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    QF_STAMP_AT(1)

    int kt = 0;
    const bool spread = EPI && KT >= 10;  // enough K-tiles to hide the epilogue operand fetch
    if (spread) {
        // e_c <- PW (tile 1), e_t <- PW^T (tile 2), commutator formed in tile 4; W and dW_old
        // are fetched during the LAST K-tile, when the staging registers are free again
        QF_KTILE_STEADY(0, 0, 0)
        QF_KTILE_STEADY(1, 1, 1)
        QF_KTILE_STEADY(2, 0, 2)
        QF_KTILE_STEADY(3, 1, 0)
        QF_KTILE_STEADY(4, 0, 3)
        QF_KTILE_STEADY(5, 1, 0)
        kt = 6;
    }
    // steady state: two K-tiles per trip so that the LDS buffer / register-set index is a literal
    for (; kt + ((QF_TAIL_STEADY_HERE) ? 1 : 4) < KT; kt += 2) {
        QF_KTILE_STEADY(kt, 0, 0)
        QF_KTILE_STEADY(kt + 1, 1, 0)
    }
    // tail (kt is even here; exact tilings: at most one K-tile, else at most 4)
    QF_KLOOP_TAIL(kt)
    if (EPI) {
        if (!spread) {
            QF_EPI_FETCH(e_c, ep.PW, false)
            QF_EPI_FETCH(e_t, ep.PW, true)
            QF_EPI_COMM
        }
        // W and dW_old are plain row-coalesced reads: fetched here, when the staging and
        // fragment registers are dead (prefetching them inside the K loop made hipcc spill)
        QF_EPI_FETCH(e_w, ep_W, false)
        QF_EPI_FETCH(e_old, ep_dW_old, false)
    }
    // (the K-loop macros stay defined: k_zgemm_tri below is built from the same pieces)

    if constexpr (!EPI) {
#if QF_WT_PW
        const __amdgpu_buffer_rsrc_t rsrcC = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0x7fffffff, 0x00020000);
#endif
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    int gi = i0 + wm * WTM + mi * 16 + q4 + 4 * reg;
                    int gj = j0 + wn * WTN + ni * 16 + r16;
                    const double cre = accR[mi][ni][reg] - accI[mi][ni][reg];
                    const double cim = (accS[mi][ni][reg] - accR[mi][ni][reg]) - accI[mi][ni][reg];
                    if (EXACT || (gi < N && gj < N)) {
#if QF_WT_PW
                        const cplx v = make_double2(cre, cim);
                        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4u *>(&v), rsrcC,
                                                               (unsigned)(((size_t)gi * N + gj) * sizeof(cplx)), 0, 16);
#else
                        C[(size_t)gi * N + gj] = make_double2(cre, cim);
#endif
                    }
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
                        const double cr = e_c[mi][ni][reg].x;   // commutator, formed in the K loop
                        const double ci = e_c[mi][ni][reg].y;
                        // dW = (PW @ Phalf) + comm                  (isospectral.py:499,509)
                        const double tre = accR[mi][ni][reg] - accI[mi][ni][reg];
                        const double tim = (accS[mi][ni][reg] - accR[mi][ni][reg]) - accI[mi][ni][reg];
                        const double dr = tre + cr;
                        const double di = tim + ci;
                        ep_dW_new[e] = make_double2(dr, di);
                        // Whalf = W + dW for the next iteration      (isospectral.py:481-482)
                        const cplx w = e_w[mi][ni][reg];
                        ep.Whalf[e] = make_double2(w.x + dr, w.y + di);
                        if (FUSED) {
                            // should this be the step's last iteration: W_next = W + 2 comm
                            // (isospectral.py:547,592) and the next step's first Whalf = W_next + dW
                            const double wr = w.x + 2.0 * cr, wi = w.y + 2.0 * ci;
                            ep_Wnext[e] = make_double2(wr, wi);
                            ep.Whalf_step[e] = make_double2(wr + dr, wi + di);
                        }
                        // |dW_old - dW|                             (isospectral.py:526,534)
                        const cplx o = e_old[mi][ni][reg];
                        const double er = o.x - dr, ei = o.y - di;
                        rsum += qf_modulus(er, ei);
                    }
                }
                // sum over the 16 lanes that share this row (fixed butterfly: deterministic)
                rsum = qf_row16_sum(rsum);      // (xor butterfly 1, 2, 4, 8 on DPP: same tree, same bits as four __shfl_xor steps)
                if (r16 == 0) rs[wn * BM + li] = rsum;
            }
        }
        __syncthreads();
        for (int li = tid; li < BM; li += T) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < WN; ++c) s += rs[c * BM + li];
            if (EXACT || i0 + li < N) {
                if (FUSED) __hip_atomic_store(ep.rowpart + (size_t)tn * N + i0 + li, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else ep.rowpart[(size_t)tn * N + i0 + li] = s;
            }
        }
        if (FUSED) {
            // the last tile to get here closes the iteration (ticket: guide section 6 G16, counter form)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned *last_flag = reinterpret_cast<unsigned *>(rs + WN * BM);
            if (tid == 0) {
                unsigned old = 0u;
                if (!((ep.debug_drop & 2) && lid == 0)) old = __hip_atomic_fetch_add(ep.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *last_flag = (old == (unsigned)(ep.n_tiles - 1)) ? 1u : 0u;
            }
            __syncthreads();
            if (*last_flag != 0u)
                qf_fused_step_end(N, tiles_n, ep.rowpart, ep.ticket, ep.state_rw, ep.rec, guard.iter, tid, rs + WN * BM + 2);
        }
    }
    QF_STAMP_AT(KT + 2)
}

// ===========================================================================================
// Second product of a skew-Hermitian iteration on the upper triangle only, stream-K.
//
// With Phalf and Whalf skew-Hermitian, T = (Phalf Whalf) Phalf is skew-Hermitian and so is
// dW = T + (PW - PW^H) (isospectral.py:499-509): only the nt(nt+1)/2 tiles on and above the
// diagonal are multiplied, and the workgroup that finishes a tile also writes its mirror image
// dW[j,i] = -conj(dW[i,j]).  That is 53 % of the MFMA work of the full product, but 136 (N=1024)
// or 528 (N=2048) tiles quantise badly on 256 CUs, so the work is not dealt out by tiles: the
// nt(nt+1)/2 x N/16 (tile, K-tile) units are cut into gridDim.x equal contiguous ranges, one per
// CU (stream-K).  A workgroup whose range starts inside a tile multiplies that piece first,
// parks the partial tile in global memory and raises its flag; the workgroup whose range holds
// the tile's first K-tile ("head") adds the parked pieces in a fixed order (bit-reproducible
// runs) and runs the fused epilogue for the tile and its mirror.
//   * pieces are produced at the START of a workgroup's life and consumed at the END of
//     another's, so a consumer practically never waits;
//   * the grid never exceeds the CU count and a workgroup needs > half a CU's LDS, so all
//     workgroups are resident: the waits cannot deadlock.  They are bounded all the same and
//     report through qf_dev_state::fault (checked by qf_isomp).
//   * flags carry the launch's epoch, so nothing is reset between launches and a guarded
//     (no-op) launch leaves nothing behind.
// Only for exact 64x64 tilings (N % 64 == 0); the host checks that W is skew-Hermitian.
// dynamic LDS of k_zgemm_tri: the K-loop buffers (97 KiB) or the epilogue's two transposition
// tiles + sum scratch (132 KiB), whichever is larger; one workgroup per CU either way
constexpr size_t TRI_SMEM_BYTES = 2 * 64 * 65 * sizeof(cplx) + 4 * 64 * sizeof(double) + 64;   // + the 'cannot close' flag word

// (Round 4 also built a heads-and-contributors schedule -- workgroup t < n_tiles multiplies only tile t's first K-tiles,
// the others share the tails -- and blocked tile orders: correct, measured slower / neutral, removed in round 5; the
// measurements are in DESIGN.md 3.1b.)
__global__ __launch_bounds__(256) void k_zgemm_tri(int N, int nt, int U, int E, const cplx *__restrict__ A,
                                                    const cplx *__restrict__ B, qf_epilogue ep, qf_guard guard,
                                                    qf_streamk sk)
{
    // (the launch's tag is looked at BELOW, behind the requests for the first segment's first two K-tiles: the control
    // state was last written by another XCD, its scalar loads are a memory round trip, and the operand tiles -- whose
    // addresses follow from the partition alone -- travel during it; a launch that is not due drops them)
    constexpr int BM = 64, BN = 64, WM = 2, WN = 2;
    constexpr bool EPI = true, EXACT = true, FAST = true;
    using SM = tile_smem<BM, BN, true>;
    constexpr int T = WM * WN * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MT = WTM / 16, NT = WTN / 16;
    constexpr int A_STRIDE = SM::A_STRIDE, B_STRIDE = SM::B_STRIDE;
    constexpr int A_PER = (BM * BK) / T, B_PER = (BN * BK) / T;
    constexpr int A_ROWS_PER = T / BK, B_ROWS_PER = T / BN;
    constexpr int A3_STRIDE = SM::A3_STRIDE, B3_STRIDE = SM::B3_STRIDE;
    constexpr int TS = BN + 1;   // row stride (complex) of the tile parked in LDS for the mirror pass
    constexpr int TT_BYTES = BM * TS * (int)sizeof(cplx);
    static_assert((size_t)2 * TT_BYTES + (WN * BM + WM * BN) * sizeof(double) + 16 <= TRI_SMEM_BYTES, "epilogue scratch exceeds the LDS request");
    static_assert(TRI_SMEM_BYTES >= SM::main_bytes, "K-loop buffers exceed the LDS request");
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_WR = 0x200;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16_c = lane & 15, q4_c = lane >> 4;
    const int G = gridDim.x;
    const int c = xcd_remap(blockIdx.x, G);
    const int KTN = N / BK;                        // K-tiles of one output tile
    // Cost space of the partition: tile t occupies [t S, (t+1) S), S = KTN + E.  Position 0 is its
    // K-tile 0, positions 1..E stand for the finisher's extra work (gathering the parked pieces,
    // the epilogue, the fused step end: E K-tiles' worth, measured), positions E+1..S-1 are
    // K-tiles 1..KTN-1.  Equal cost ranges per workgroup: a workgroup that finishes a tile gets
    // correspondingly fewer K-tiles than one that only multiplies.
    const int S = KTN + E;
    // (cost positions fit 32 bits -- U < 2^31 is checked by the host; only the products c U need 64: the partition's
    // running values live in SGPRs, and this kernel has none to spare)
    int u = (int)((long long)c * U / G);
    const int u_end = (int)((long long)(c + 1) * U / G);
    const cplx zero = make_double2(0.0, 0.0);

    // per-thread LDS bases (FAST layout of k_zgemm)
    const unsigned char *lds_fa[2] = {
        smem_raw + (size_t)(q4_c * A_STRIDE + wm * WTM + (r16_c ^ q4_c)) * sizeof(cplx),
        smem_raw + (size_t)(q4_c * A_STRIDE + wm * WTM + (r16_c ^ q4_c ^ 4)) * sizeof(cplx)};
    const unsigned char *lds_fb = smem_raw + SM::B_OFFSET + (size_t)(q4_c * B_STRIDE + wn * WTN + r16_c) * sizeof(cplx);
    unsigned char *lds_sa = smem_raw + (size_t)((tid % BK) * A_STRIDE + ((tid / BK) ^ (tid % BK & 7))) * sizeof(cplx);
    const unsigned char *lds_fa3[2] = {
        smem_raw + SM::A3_OFFSET + (size_t)(q4_c * A3_STRIDE + wm * WTM + 2 * (r16_c ^ q4_c)) * sizeof(double),
        smem_raw + SM::A3_OFFSET + (size_t)(q4_c * A3_STRIDE + wm * WTM + 2 * (r16_c ^ q4_c ^ 4)) * sizeof(double)};
    const unsigned char *lds_fb3 = smem_raw + SM::B3_OFFSET + (size_t)(q4_c * B3_STRIDE + wn * WTN + 2 * r16_c) * sizeof(double);
    unsigned char *lds_sa3 = smem_raw + SM::A3_OFFSET + (size_t)((tid % BK) * A3_STRIDE + 2 * ((tid / BK) ^ (tid % BK & 7))) * sizeof(double);
    const int b_jA = ((tid % 32) / 16) * 32 + (tid % 16);
    unsigned char *lds_sb3 = smem_raw + SM::B3_OFFSET + (size_t)((tid / 32) * B3_STRIDE + 32 * ((tid % 32) / 16) + 2 * (tid % 16)) * sizeof(double);
    unsigned char *lds_sb = smem_raw + SM::B_OFFSET + (size_t)((tid / 32) * B_STRIDE + b_jA) * sizeof(cplx);
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(A), 0, (int)((size_t)N * N * sizeof(cplx)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(B), 0, (int)((size_t)N * N * sizeof(cplx)), 0x00020000);
    const unsigned fa_voff = (unsigned)(((size_t)(tid / BK) * N + (tid % BK)) * sizeof(cplx));
    const unsigned fb_voff = (unsigned)(((size_t)(tid / 32) * N + b_jA) * sizeof(cplx));
    const unsigned f_rows16 = (unsigned)((size_t)16 * N * sizeof(cplx));
    // exchange area: one 64 KiB slot per workgroup, element q of thread tid at [q][tid]
    const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(sk.partial, 0, (int)((size_t)sk.slots * (BM * BN) * sizeof(cplx)), 0x00020000);
    const unsigned p_voff = (unsigned)(tid * sizeof(cplx));
    // names that only the generic (non-FAST) arms of the shared macros mention; never executed here
    const unsigned char *a_row = nullptr, *b_col = nullptr;
    const size_t a_pass = 0, b_pass = 0, b_ktile = 0;
    const unsigned a_voff = 0, b_voff = 0;
    const int kt_off = 0;
    const int i0 = 0, j0 = 0;   // (shadowed by the current tile's origin inside the segment loop)

    cplx e_c[MT][NT][4], e_t[MT][NT][4], e_w[MT][NT][4], e_old[MT][NT][4];
    v4d accR[MT][NT], accI[MT][NT], accS[MT][NT];
    cplx ra[2][A_PER], rb[2][B_PER];
    cplx fa[2][MT], fb[2][NT];
    double fas[2][MT], fbs[2][NT];
    // (the second staging set is loaded only by segments of two K-tiles or more, but staged by the steady K-tile of any)
#pragma unroll
    for (int r = 0; r < A_PER; ++r) ra[1][r] = rb[1][r] = zero;

    // first K-tile whose cost position is >= p (p relative to the tile's origin)
#define QF_TRI_KOF(p_) ((p_) <= 0 ? 0 : ((p_) <= E + 1 ? 1 : ((p_) - E > KTN ? KTN : (int)((p_) - E))))
    // next non-empty segment (= the part of one tile's K range that falls into this workgroup's
    // cost range) at or after cost position u_; on return u_ is the position behind it
#define QF_TRI_NEXT(u_, found_, t_, k0_, KT_, tm_, tn_)                                \
    {                                                                                  \
        found_ = false;                                                                \
        while (!(found_) && (u_) < u_end) {                                            \
            t_ = (int)((u_) / S);                                                      \
            const int pa_ = (u_) - (t_) * S;                                           \
            int pb_ = u_end - (t_) * S;                                                \
            if (pb_ > S) pb_ = S;                                                      \
            const int klo_ = QF_TRI_KOF(pa_), khi_ = QF_TRI_KOF(pb_);                  \
            u_ = (t_) * S + pb_;                                                       \
            if (klo_ < khi_) {                                                         \
                found_ = true;                                                         \
                k0_ = klo_;                                                            \
                KT_ = khi_ - klo_;                                                     \
                int rem_ = (t_);                                                       \
                tm_ = 0;                                                               \
                while (rem_ >= nt - (tm_)) { rem_ -= nt - (tm_); ++(tm_); }            \
                tn_ = (tm_) + rem_;                                                    \
            }                                                                          \
        }                                                                              \
    }
    // K-tiles 0 and 1 of a segment start their way L2 -> registers before the previous segment's
    // publish / epilogue, so that a segment's prologue does not pay two exposed memory latencies
#define QF_TRI_START_LOADS(k0_, KT_, tm_, tn_)                                         \
    {                                                                                  \
        const int kt_of_segment_ = (KT_);                                              \
        fa_soff0 = (unsigned)(((size_t)(tm_) * BM * N + (size_t)(k0_) * BK) * sizeof(cplx)); \
        fb_soff0 = (unsigned)(((size_t)(k0_) * BK * N + (size_t)(tn_) * BN) * sizeof(cplx)); \
        {                                                                              \
            const int KT = kt_of_segment_;       /* (QF_KT_CLAMP reads `KT`: the segment these loads belong to) */ \
            QF_LOAD_TILE(0, 0)                                                         \
            if (KT > 1) { QF_LOAD_TILE(1, 1) }                                         \
        }                                                                              \
    }
    int seg = 0;       // (counted for the stamps of the diagnostic build)
    (void)seg;
    bool run_finale = false;     // this workgroup's epilogue was the last of all: it closes the iteration
    int t = 0, k0 = 0, KT = 0, tm = 0, tn = 0;
    unsigned fa_soff0 = 0, fb_soff0 = 0;
    bool have = false;
    QF_TRI_NEXT(u, have, t, k0, KT, tm, tn)
    if (have) QF_TRI_START_LOADS(k0, KT, tm, tn)
    if (!qf_guard_iter(guard)) have = false;
    const int parity = guard.state ? guard.state->dw_parity : 0;
    const cplx *__restrict__ ep_dW_old = ep.dW[parity];
    cplx *__restrict__ ep_dW_new = ep.dW[parity ^ 1];
    // fused step end: current state = Wpair[w_parity]; the candidate next state goes to the other
    const int wpar = (ep.fused && guard.state) ? guard.state->w_parity : 0;
    const cplx *__restrict__ ep_W = ep.fused ? ep.Wpair[wpar] : ep.W;
    cplx *__restrict__ ep_Wnext = ep.fused ? ep.Wpair[wpar ^ 1] : nullptr;
    // Fused step end: the candidate next state and the next step's first Whalf are written "in case this
    // iteration closes the step".  A tile can often tell that it will not: the step stays open if the
    // iteration is below minit, and in a step's FIRST iteration (previous residual = inf, so only
    // residual <= tol can close it, isospectral.py:535-536) as soon as one of this tile's partial row sums
    // of |dW_old - dW| alone exceeds tol -- the full row sum, hence the norm, is at least that.  Never when
    // this is iteration maxit (the step closes regardless, :538-540).  Exact, tile by tile: no prediction.
    const bool below_maxit = ep.fused && sk.state_rw && (guard.iter + 1 < sk.state_rw->maxit);
    const bool open_for_sure = below_maxit && (guard.iter + 1 < sk.state_rw->minit);
    const bool open_if_large = below_maxit && guard.iter == 0;
    const double tol_now = (ep.fused && sk.state_rw) ? sk.state_rw->tol : 0.0;

    while (have) {
        const int i0 = tm * BM, j0 = tn * BN;
        const bool head = (k0 == 0);
        // lane coordinates as values the optimiser cannot see through: otherwise it hoists the
        // epilogue's ~50 per-element address terms out of the segment loop and pays for the
        // registers with scratch (whose presence alone costs ~0.2 ms of host time per launch)
        int r16 = r16_c, q4 = q4_c, lane_v = lane;
        asm volatile("" : "+v"(r16), "+v"(q4), "+v"(lane_v));
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                accR[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
                accI[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
                accS[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            }

        QF_TRI_STAMP(seg, 0)
        QF_TRI_NOTE(seg, 4, k0)
        QF_TRI_NOTE(seg, 5, KT)
        QF_STORE_TILE(0, 0)
        __syncthreads();
        if (KT > 2) { QF_LOAD_TILE(2, 0) }
        QF_READ_FRAGS(0, 0, 0)
        __builtin_amdgcn_s_waitcnt(0xC07F);

        QF_TRI_STAMP(seg, 6)
        int kt = 0;
        const bool spread = head && KT >= 10;   // hide the epilogue operand fetch under the K loop
        if (spread) {
            QF_KTILE_STEADY(0, 0, 0)
            QF_KTILE_STEADY(1, 1, 1)
            QF_KTILE_STEADY(2, 0, 2)
            QF_KTILE_STEADY(3, 1, 0)
            QF_KTILE_STEADY(4, 0, 3)
            QF_KTILE_STEADY(5, 1, 0)
            kt = 6;
        }
        for (; kt + (QF_TAIL_STEADY_HERE ? 1 : 4) < KT; kt += 2) {
            QF_KTILE_STEADY(kt, 0, 0)
            QF_KTILE_STEADY(kt + 1, 1, 0)
        }
        QF_KLOOP_TAIL(kt)
        QF_TRI_STAMP(seg, 1)

        int n_t = 0, n_k0 = 0, n_KT = 0, n_tm = 0, n_tn = 0;
        bool have_next = false;
        QF_TRI_NEXT(u, have_next, n_t, n_k0, n_KT, n_tm, n_tn)

        if (!head) {
            const int park_slot = c;
            // a piece of a tile whose head lives in another workgroup: park it (thread-major, one
            // 1 KiB write-through store per wave instruction: no release fence needed), drain, publish
            // (hand-off form: cdna_hip_programming.md section 6, Guideline 16 R1)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const cplx v = make_double2(accR[mi][ni][reg] - accI[mi][ni][reg],
                                                    (accS[mi][ni][reg] - accR[mi][ni][reg]) - accI[mi][ni][reg]);
                        __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4u *>(&v), rsrcP,
                                                               p_voff + (unsigned)(((mi * NT + ni) * 4 + reg) * T * sizeof(cplx)),
                                                               (unsigned)((size_t)park_slot * (BM * BN) * sizeof(cplx)), QF_SK_AUX_ST);
                    }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
            if (have_next) QF_TRI_START_LOADS(n_k0, n_KT, n_tm, n_tn)
            __syncthreads();
            if (tid == 0 && !(sk.debug_drop & 1)) __hip_atomic_store(sk.flags + park_slot, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            QF_TRI_STAMP(seg, 2)
        } else {
            if (!spread) {
                QF_EPI_FETCH(e_c, ep.PW, false)
                QF_EPI_FETCH(e_t, ep.PW, true)
                QF_EPI_COMM
            }
            QF_EPI_FETCH(e_w, ep_W, false)
            QF_EPI_FETCH(e_old, ep_dW_old, false)
            // T = Re/Im of the 3M accumulators, plus the pieces other workgroups parked
            double tre[MT][NT][4], tim[MT][NT][4];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        tre[mi][ni][reg] = accR[mi][ni][reg] - accI[mi][ni][reg];
                        tim[mi][ni][reg] = (accS[mi][ni][reg] - accR[mi][ni][reg]) - accI[mi][ni][reg];
                    }
            if (KT < KTN) {
                // The rest of this tile lies with the workgroups behind this one.  Thread 0 polls a piece's
                // flag, then every wave reads the piece with sc1 loads (never through this CU's L1) and adds
                // it; the pieces are taken in a fixed order.  A workgroup whose range inside this tile covers
                // no K-tile (it lies in the cost positions that stand for the epilogue) parks nothing and is
                // skipped.
                // (the workgroups behind this one, c + 1 .. c_last; slot = workgroup)
                const int tile_org = t * S, tile_end = tile_org + S;
                const int c_first = c + 1;
                int c_last = c;
                while (c_last + 1 < G && (int)((long long)(c_last + 1) * U / G) < tile_end) ++c_last;
#define QF_TRI_HAS_PIECE(c2_)                                                          \
    (QF_TRI_KOF((int)((long long)(c2_) * U / G) - tile_org) <                          \
     QF_TRI_KOF(((int)(((long long)(c2_) + 1) * U / G) < tile_end ? (int)(((long long)(c2_) + 1) * U / G) : tile_end) - tile_org))
                // Farthest piece first: a workgroup whose range starts inside this tile parked its piece at
                // the START of its life, the one whose whole range lies inside the tile (most tiles have
                // one) finishes only now, with this workgroup.  Taking the pieces in descending order puts
                // the early ones' memory round trips behind that wait instead of behind the late one's
                // (fixed order all the same: bit-reproducible runs).
                for (int c2 = c_last; c2 >= c_first; --c2) {
                    const int slot2 = c2;
                    if (!QF_TRI_HAS_PIECE(c2)) continue;
                    if (tid == 0) {
                        unsigned spins = 0;
                        while (__hip_atomic_load(sk.flags + slot2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) {
                            __builtin_amdgcn_s_sleep(QF_POLL_SLEEP);
                            if (++spins > sk.spin_limit) {
                                *sk.fault = 1;
                                break;
                            }
                        }
                    }
                    asm volatile("s_barrier" ::: "memory");   // the polling wave joins after its poll matched
                    const unsigned soff = (unsigned)((size_t)slot2 * (BM * BN) * sizeof(cplx));
                    cplx v[MT * NT * 4];
#pragma unroll
                    for (int q = 0; q < MT * NT * 4; ++q) {
                        const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rsrcP, p_voff + (unsigned)(q * T * sizeof(cplx)), soff, QF_SK_AUX_LD);
                        v[q] = *reinterpret_cast<const cplx *>(&raw);
                    }
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) {
                                tre[mi][ni][reg] += v[(mi * NT + ni) * 4 + reg].x;
                                tim[mi][ni][reg] += v[(mi * NT + ni) * 4 + reg].y;
                            }
                }
            }
            QF_TRI_STAMP(seg, 2)
            // ---- fused epilogue of the tile (as in k_zgemm) and of its mirror image.
            // W and dW_old are EXACTLY skew-Hermitian here (the host checks W; dW_old is zero or this
            // kernel's own output), so the mirrored entries need no loads:
            //   Whalf[j,i] = W[j,i] + dW[j,i] = -conj(Whalf[i,j])  (exact),
            //   |dW_old[j,i] - dW[j,i]| = |dW_old[i,j] - dW[i,j]|: mirror rows' sums = this tile's column sums.
            // Fused step end: only Whalf (and the next step's Whalf) are needed below the diagonal inside
            // the stepper -- they are the right operand of the next first product; dW and the state W are
            // read back by this kernel's epilogue alone, on and above the diagonal tiles.  Their lower
            // triangles are NOT written then: qf_isomp restores them once, at the end of the call
            // (qf_launch_mirror_lower), which takes two tile stores and an LDS pass out of every epilogue.
            // (Two-kernel protocol: k_update reads all of dW, so its mirror image is written here.)
            cplx *Th = reinterpret_cast<cplx *>(smem_raw);                      // [BM][TS] Whalf tile to mirror
            cplx *Ts = reinterpret_cast<cplx *>(smem_raw + TT_BYTES);           // [BM][TS] next step's Whalf tile to mirror
            double *rs = reinterpret_cast<double *>(smem_raw + 2 * TT_BYTES);   // [WN][BM] row sums
            double *cs = rs + WN * BM;                                          // [WM][BN] column sums
            unsigned *open_flag = reinterpret_cast<unsigned *>(cs + WM * BN);   // a partial sum above tol was seen
            const bool offdiag = (tm != tn);
            // (LDS-only barriers from here on: __syncthreads() would also drain the global stores)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the K-loop buffers

            // ---- phase 1: dW = (PW @ Phalf) + comm (isospectral.py:499,509), Whalf = W + dW (:481-482) -- stored as
            // they are formed -- and the sums of |dW_old - dW| (:526,534).  Round 4: the moduli (64 correctly rounded
            // double-precision square roots per lane, ~15 dependent VALU instructions each) are the longest stretch of
            // the epilogue, ~4 us in the stamps; the tile's 32 stores per lane now leave from INSIDE that loop, where
            // they cost issue slots the VALU chain does not use, instead of standing behind it.  The row sums still
            // leave before the candidate tiles, and the ticket is taken before those are stored.
            double csum[NT] = {0.0, 0.0};
            if (tid == 0) *open_flag = 0u;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int li = wm * WTM + mi * 16 + q4 + 4 * reg;
                    double rsum = 0.0;
#pragma unroll
                    for (int ni = 0; ni < NT; ++ni) {
                        const int lj = wn * WTN + ni * 16 + r16;
                        const size_t e = (size_t)(i0 + li) * N + (j0 + lj);
                        const double dr = tre[mi][ni][reg] + e_c[mi][ni][reg].x;
                        const double di = tim[mi][ni][reg] + e_c[mi][ni][reg].y;
                        tre[mi][ni][reg] = dr;
                        tim[mi][ni][reg] = di;
                        const cplx w = e_w[mi][ni][reg];
                        const cplx wh = make_double2(w.x + dr, w.y + di);
                        ep_dW_new[e] = make_double2(dr, di);
                        ep.Whalf[e] = wh;
                        Th[li * TS + lj] = wh;     // (unused on diagonal tiles: cheaper than a branch)
                        if (!ep.fused) Ts[li * TS + lj] = make_double2(dr, di);      // two-kernel protocol: k_update reads all of dW
                        const cplx o = e_old[mi][ni][reg];
                        const double er = o.x - dr, ei = o.y - di;
                        const double a = qf_modulus(er, ei);
                        rsum += a;
                        csum[ni] += a;
                    }
                    rsum = qf_row16_sum(rsum);
                    if (r16 == 0) rs[wn * BM + li] = rsum;
                }
            }
            if (offdiag) {
#pragma unroll
                for (int ni = 0; ni < NT; ++ni) {
                    double s2 = csum[ni];
                    s2 += __shfl_xor(s2, 16, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    if (q4 == 0) cs[wm * BN + wn * WTN + ni * 16 + r16] = s2;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // (relaxed agent-scope stores = write-through: the fused step end's last finisher reads them)
            if (tid < BM) {
                double s2 = 0.0;
#pragma unroll
                for (int cc = 0; cc < WN; ++cc) s2 += rs[cc * BM + tid];
                __hip_atomic_store(ep.rowpart + (size_t)tn * N + i0 + tid, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (open_if_large && s2 > tol_now) *open_flag = 1u;
            } else if (offdiag && tid < BM + BN) {
                const int lj = tid - BM;
                double s2 = 0.0;
#pragma unroll
                for (int cc = 0; cc < WM; ++cc) s2 += cs[cc * BN + lj];
                __hip_atomic_store(ep.rowpart + (size_t)tm * N + j0 + lj, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (open_if_large && s2 > tol_now) *open_flag = 1u;
            }
            unsigned ticket_old = 0u;
            if (ep.fused) {
                // fused step end: the last of the n_tiles epilogues decides.  Every storing wave
                // drains its row sums, one lane takes a ticket (guide section 6 G16: counter form of the
                // hand-off).  Nobody waits for the ticket's answer here: it is looked at behind the candidate
                // tiles and the mirror pass, at the end of the segment.
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (tid == 0 && !((sk.debug_drop & 2) && t == 0))
                    ticket_old = __hip_atomic_fetch_add(sk.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (open_flag, Th / Ts: LDS only)
            }
            QF_TRI_STAMP(seg, 7)
            // (uniform) the stores for "should this iteration close the step" are dead when it cannot
            const bool speculate = ep.fused && !(open_for_sure || (open_if_large && *open_flag != 0u));

            // ---- phase 2, fused step end only: should this iteration turn out to be the step's last -- the next state
            // W + 2 (PW - PW^H) (isospectral.py:547,592) and the next step's first Whalf = that + dW, written
            // speculatively into the spare W buffer / the second Whalf buffer (the decision only flips two indices).
            // The second Whalf tile also goes to LDS for the mirror pass.
            if (speculate) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int li = wm * WTM + mi * 16 + q4 + 4 * reg;
#pragma unroll
                        for (int ni = 0; ni < NT; ++ni) {
                            const int lj = wn * WTN + ni * 16 + r16;
                            const size_t e = (size_t)(i0 + li) * N + (j0 + lj);
                            const cplx w = e_w[mi][ni][reg];
                            const cplx wc = make_double2(w.x + 2.0 * e_c[mi][ni][reg].x, w.y + 2.0 * e_c[mi][ni][reg].y);
                            const cplx whs = make_double2(wc.x + tre[mi][ni][reg], wc.y + tim[mi][ni][reg]);
                            ep_Wnext[e] = wc;
                            ep.Whalf_step[e] = whs;
                            Ts[li * TS + lj] = whs;
                        }
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (offdiag) {
                // row j0+jl of the mirrored tile is column jl of this one: a wave owns 16 such rows
                // and writes each as one coalesced 1 KiB segment
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int jl = wave * 16 + r;
                    const cplx wv = Th[lane_v * TS + jl];
                    const size_t e2 = (size_t)(j0 + jl) * N + (i0 + lane_v);
                    ep.Whalf[e2] = make_double2(-wv.x, wv.y);       // -conj(Whalf[i,j])
                    if (speculate || !ep.fused) {
                        const cplx ws = Ts[lane_v * TS + jl];
                        if (ep.fused) ep.Whalf_step[e2] = make_double2(-ws.x, ws.y);
                        else ep_dW_new[e2] = make_double2(-ws.x, ws.y);         // -conj(dW[i,j])
                    }
                }
            }
            // (after the epilogue, not before it: its operands need the registers)
            if (have_next) QF_TRI_START_LOADS(n_k0, n_KT, n_tm, n_tn)
            if (ep.fused) {
                // was this the last epilogue of all?  (the decision itself runs after the segment loop,
                // when none of the epilogue's registers are live: inlined here it cost the K loop 36
                // register moves per two K-tiles)
                unsigned *last_flag = reinterpret_cast<unsigned *>(rs);     // rs / cs have been consumed
                if (tid == 0) *last_flag = (ticket_old == (unsigned)(sk.n_tiles - 1)) ? 1u : 0u;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (*last_flag != 0u) run_finale = true;
            }
        }
        QF_TRI_STAMP(seg, 3)
        ++seg;
        have = have_next;
        t = n_t;
        k0 = n_k0;
        KT = n_KT;
        tm = n_tm;
        tn = n_tn;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the next segment's prologue rewrites the LDS buffers
    }
    if (run_finale)
        qf_fused_step_end(N, nt, ep.rowpart, sk.ticket, sk.state_rw, sk.rec, guard.iter, tid, reinterpret_cast<double *>(smem_raw));
#undef QF_TRI_NEXT
#undef QF_TRI_KOF
#undef QF_TRI_HAS_PIECE
#undef QF_TRI_START_LOADS
}

// ===========================================================================================
// The second product on the upper triangle for N < 768: 32 x 32 tiles, K split over two workgroups.
//
// Below N = 768 a product launch is one 32x32 tile per CU and the stream-K form above does not pay:
// its partition would give a workgroup 4-5 K-tiles at N = 512 and the gather / epilogue tail of the
// finishing workgroups is longer than that (measured: 38 us against 28 us for the full product).  What
// pays is the symmetry alone: the nt (nt + 1) / 2 tiles on and above the diagonal (136 of 256 at
// N = 512) leave half of the CUs idle, so every off-diagonal tile is given to TWO workgroups, one per
// half of the K range -- 2 * 120 + 16 = 256 workgroups at N = 512, one per CU, each with half the K loop.
//   * a workgroup parks its half-K partial tile (16 KiB, thread-major, write-through), drains, takes a
//     ticket on the tile's arrival counter and leaves if it came first; the one that comes second adds
//     the parked half to its own -- a two-term sum: the same bits whichever half arrives last -- and runs
//     the epilogue for the tile and its mirror image.  Nobody ever waits: no spin, no residency
//     requirement (the counters are monotone: two arrivals per executed launch and tile);
//   * hand-off form: cdna_hip_programming.md section 6, Guideline 16 R1 in its counter form (16-byte sc1
//     stores, every storing wave drains vmcnt(0), barrier, ONE lane's agent-scope atomic add; the wave
//     that learns it came last loads after its add has returned, the others behind a barrier; sc1 loads);
//   * the epilogue is k_zgemm_tri's: residual sums first (write-through) and the step-end ticket before
//     the tile stores, mirror of Whalf (and of the next step's Whalf) through LDS as whole row segments,
//     lower triangles of dW and W not written in the fused protocol (qf_isomp restores them at its end).
// Diagonal tiles are multiplied whole by one workgroup (split_diag = 1) and dealt out first.
// EXACT = false: N is no multiple of 32 -- nt = ceil(N / 32) tile rows, operands of the edge tiles zero-filled (they add
// nothing to the products), stores and row sums guarded, K-tiles = ceil(N / 16).
template <bool EXACT>
__global__ __launch_bounds__(256) void k_zgemm_tri32(int N, int nt, const cplx *__restrict__ A, const cplx *__restrict__ B,
                                                      qf_epilogue ep, qf_guard guard, qf_tri32 sx)
{
    if (!qf_guard_iter(guard)) return;
    constexpr int BM = 32, BN = 32, WM = 2, WN = 2;
    constexpr bool EPI = true, FAST = false;
    using SM = tile_smem<BM, BN, false>;
    constexpr int T = WM * WN * 64;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MT = WTM / 16, NT = WTN / 16;
    static_assert(MT == 1 && NT == 1, "one MFMA tile per wave");
    constexpr int A_STRIDE = SM::A_STRIDE, B_STRIDE = SM::B_STRIDE;
    constexpr int A_PER = (BM * BK) / T, B_PER = (BN * BK) / T;
    constexpr int A_ROWS_PER = T / BK, B_ROWS_PER = T / BN;
    constexpr int A3_STRIDE = SM::A3_STRIDE, B3_STRIDE = SM::B3_STRIDE;
    constexpr int TS = BN + 1;   // row stride (complex) of the tile parked in LDS for the mirror pass
    constexpr int TT_BYTES = BM * TS * (int)sizeof(cplx);
    static_assert((size_t)2 * TT_BYTES + (WN * BM + WM * BN) * sizeof(double) + 16 <= SM::bytes, "epilogue scratch exceeds the K-loop buffers");
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int SG_MFMA = 0x008, SG_VMEM_RD = 0x020, SG_DS_WR = 0x200;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r16 = lane & 15, q4 = lane >> 4;

    // ---- workgroup -> (tile, K range).  Diagonal tiles first (whole K range when split_diag = 1: the long
    // jobs start first), then the off-diagonal ones half-major, so that neighbouring ids are neighbouring
    // tiles of one row with the SAME K range (they share their A panel); XCD-aware over the off-diagonal part.
    const int n_diag_w = nt * sx.split_diag;
    const int n_off = nt * (nt - 1) / 2;
    int tm, tn, h, nh, t;
    if ((int)blockIdx.x < n_diag_w) {
        nh = sx.split_diag;
        tm = tn = (int)blockIdx.x % nt;
        h = (int)blockIdx.x / nt;
        t = tm;
    } else {
        const int nw = (int)gridDim.x - n_diag_w;
        const int w = (n_diag_w % 8 == 0) ? xcd_remap((int)blockIdx.x - n_diag_w, nw) : (int)blockIdx.x - n_diag_w;
        nh = sx.split;
        const int o = w % n_off;
        h = w / n_off;
        tm = 0;
        int rem = o;
        while (rem >= nt - 1 - tm) {
            rem -= nt - 1 - tm;
            ++tm;
        }
        tn = tm + 1 + rem;
        t = nt + o;
    }
    const int i0 = tm * BM, j0 = tn * BN;
    const int KTN = (N + BK - 1) / BK;
    const int kt_begin = (int)((long long)h * KTN / nh);
    const int KT = (int)((long long)(h + 1) * KTN / nh) - kt_begin;

    const cplx zero = make_double2(0.0, 0.0);
    const int parity = guard.state ? guard.state->dw_parity : 0;
    const cplx *__restrict__ ep_dW_old = ep.dW[parity];
    cplx *__restrict__ ep_dW_new = ep.dW[parity ^ 1];
    const int wpar = (ep.fused && guard.state) ? guard.state->w_parity : 0;
    const cplx *__restrict__ ep_W = ep.fused ? ep.Wpair[wpar] : ep.W;
    cplx *__restrict__ ep_Wnext = ep.fused ? ep.Wpair[wpar ^ 1] : nullptr;
    // which candidate stores can be skipped: see k_zgemm_tri
    const bool below_maxit = ep.fused && sx.state_rw && (guard.iter + 1 < sx.state_rw->maxit);
    const bool open_for_sure = below_maxit && (guard.iter + 1 < sx.state_rw->minit);
    const bool open_if_large = below_maxit && guard.iter == 0;
    const double tol_now = (ep.fused && sx.state_rw) ? sx.state_rw->tol : 0.0;

    // ---- per-thread LDS bases and global staging addresses (generic layout of k_zgemm)
    const unsigned char *lds_fa[2] = {
        smem_raw + (size_t)(q4 * A_STRIDE + wm * WTM + (r16 ^ q4)) * sizeof(cplx),
        smem_raw + (size_t)(q4 * A_STRIDE + wm * WTM + (r16 ^ q4 ^ 4)) * sizeof(cplx)};
    const unsigned char *lds_fb = smem_raw + SM::B_OFFSET + (size_t)(q4 * B_STRIDE + wn * WTN + r16) * sizeof(cplx);
    unsigned char *lds_sa = smem_raw + (size_t)((tid % BK) * A_STRIDE + ((tid / BK) ^ (tid % BK & 7))) * sizeof(cplx);
    unsigned char *lds_sb = smem_raw + SM::B_OFFSET + (size_t)tid * sizeof(cplx);
    const unsigned char *lds_fa3[2] = {
        smem_raw + SM::A3_OFFSET + (size_t)(q4 * A3_STRIDE + wm * WTM + r16) * sizeof(double),
        smem_raw + SM::A3_OFFSET + (size_t)(q4 * A3_STRIDE + wm * WTM + r16) * sizeof(double)};
    const unsigned char *lds_fb3 = smem_raw + SM::B3_OFFSET + (size_t)(q4 * B3_STRIDE + wn * WTN + r16) * sizeof(double);
    unsigned char *lds_sa3 = smem_raw + SM::A3_OFFSET + (size_t)((tid % BK) * A3_STRIDE + tid / BK) * sizeof(double);
    unsigned char *lds_sb3 = smem_raw + SM::B3_OFFSET + (size_t)tid * sizeof(double);
    const unsigned a_voff = (unsigned)(((size_t)(tid / BK) * N + (tid % BK)) * sizeof(cplx));
    const unsigned b_voff = (unsigned)(((size_t)(tid / BN) * N + (tid % BN)) * sizeof(cplx));
    const int kt_off = kt_begin;
    // (the K range of this workgroup starts at K-tile kt_begin: folded into the operand bases)
    const unsigned char *a_row = reinterpret_cast<const unsigned char *>(A) + ((size_t)i0 * N + (size_t)kt_begin * BK) * sizeof(cplx);
    const unsigned char *b_col = reinterpret_cast<const unsigned char *>(B) + ((size_t)kt_begin * BK * N + (size_t)j0) * sizeof(cplx);
    const size_t a_pass = (size_t)A_ROWS_PER * N * sizeof(cplx);
    const size_t b_pass = (size_t)B_ROWS_PER * N * sizeof(cplx);
    const size_t b_ktile = (size_t)BK * N * sizeof(cplx);
    // names that only the FAST arms of the shared macros mention; never executed here
    const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(A), 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcB = rsrcA;
    const unsigned fa_voff = 0, fb_voff = 0, fa_soff0 = 0, fb_soff0 = 0, f_rows16 = 0;

    cplx e_c[MT][NT][4], e_t[MT][NT][4], e_w[MT][NT][4], e_old[MT][NT][4];
    v4d accR[MT][NT], accI[MT][NT], accS[MT][NT];
    accR[0][0] = (v4d){0.0, 0.0, 0.0, 0.0};
    accI[0][0] = (v4d){0.0, 0.0, 0.0, 0.0};
    accS[0][0] = (v4d){0.0, 0.0, 0.0, 0.0};
    cplx ra[2][A_PER], rb[2][B_PER];
    cplx fa[2][MT], fb[2][NT];
    double fas[2][MT], fbs[2][NT];

    // ---- K loop over this workgroup's range (the schedule of k_zgemm's 32x32 instantiation)
    QF_LOAD_TILE_A(0, 0)
    if (KT > 1) { QF_LOAD_TILE_A(1, 1) }
    QF_LOAD_TILE_B(0, 0)
    if (KT > 1) { QF_LOAD_TILE_B(1, 1) }
    QF_STORE_TILE(0, 0)
    __syncthreads();
    if (KT > 2) QF_LOAD_TILE(2, 0)
    QF_READ_FRAGS(0, 0, 0)
    __builtin_amdgcn_s_waitcnt(0xC07F);
    int kt = 0;
    const bool spread = KT >= 10;   // enough K-tiles to hide the epilogue operand fetch
    if (spread) {
        QF_KTILE_STEADY(0, 0, 0)
        QF_KTILE_STEADY(1, 1, 1)
        QF_KTILE_STEADY(2, 0, 2)
        QF_KTILE_STEADY(3, 1, 0)
        QF_KTILE_STEADY(4, 0, 3)
        QF_KTILE_STEADY(5, 1, 0)
        kt = 6;
    }
    for (; kt + ((QF_TAIL_STEADY_HERE) ? 1 : 4) < KT; kt += 2) {
        QF_KTILE_STEADY(kt, 0, 0)
        QF_KTILE_STEADY(kt + 1, 1, 0)
    }
    QF_KLOOP_TAIL(kt)
    if (!spread) {
        QF_EPI_FETCH(e_c, ep.PW, false)
        QF_EPI_FETCH(e_t, ep.PW, true)
        QF_EPI_COMM
    }
    QF_EPI_FETCH(e_w, ep_W, false)
    QF_EPI_FETCH(e_old, ep_dW_old, false)

    double tre[4], tim[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        tre[reg] = accR[0][0][reg] - accI[0][0][reg];
        tim[reg] = (accS[0][0][reg] - accR[0][0][reg]) - accI[0][0][reg];
    }
    unsigned *flagw = reinterpret_cast<unsigned *>(smem_raw + 2 * TT_BYTES + (WN * BM + WM * BN) * sizeof(double));
    if (nh > 1) {
        // ---- a piece of a tile: park it, drain, take the arrival ticket; every arrival but the last is done
        const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(sx.partial, 0, 0x7fffffff, 0x00020000);
        const unsigned p_voff = (unsigned)(tid * sizeof(cplx));
        const unsigned slot_bytes = (unsigned)(BM * BN * sizeof(cplx));
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const cplx v = make_double2(tre[reg], tim[reg]);
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4u *>(&v), rsrcP, p_voff + (unsigned)(reg * T * sizeof(cplx)),
                                                   (unsigned)(4 * t + h) * slot_bytes, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
        __syncthreads();                                   // (also: every wave is done with the K-loop buffers)
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(sx.arrive + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flagw = ((old % (unsigned)nh) == (unsigned)(nh - 1)) ? 1u : 0u;
        }
        __syncthreads();
        if (*flagw == 0u) return;
        // all pieces in piece order, this workgroup's own from memory like the others: the same bits whoever came last
        // (two pieces: p0 + p1 either way, as before)
#pragma unroll 1
        for (int hh = 0; hh < nh; ++hh) {
            cplx v[4];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const v4u raw = __builtin_amdgcn_raw_buffer_load_b128(rsrcP, p_voff + (unsigned)(reg * T * sizeof(cplx)),
                                                                      (unsigned)(4 * t + hh) * slot_bytes, 16);
                v[reg] = *reinterpret_cast<const cplx *>(&raw);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                tre[reg] = hh == 0 ? v[reg].x : tre[reg] + v[reg].x;
                tim[reg] = hh == 0 ? v[reg].y : tim[reg] + v[reg].y;
            }
        }
    }

    // ---- fused epilogue of the tile and of its mirror image (k_zgemm_tri's, on 32x32)
    cplx *Th = reinterpret_cast<cplx *>(smem_raw);                      // [BM][TS] Whalf tile to mirror
    cplx *Ts = reinterpret_cast<cplx *>(smem_raw + TT_BYTES);           // [BM][TS] next step's Whalf tile (or dW) to mirror
    double *rs = reinterpret_cast<double *>(smem_raw + 2 * TT_BYTES);   // [WN][BM] row sums
    double *cs = rs + WN * BM;                                          // [WM][BN] column sums
    unsigned *open_flag = flagw;                                        // (the arrival flag has been consumed)
    const bool offdiag = (tm != tn);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the K-loop buffers / the flag
    double csum = 0.0;
    if (tid == 0) *open_flag = 0u;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int li = wm * WTM + q4 + 4 * reg;
        const double dr = tre[reg] + e_c[0][0][reg].x;     // dW = (PW @ Phalf) + comm      (isospectral.py:499,509)
        const double di = tim[reg] + e_c[0][0][reg].y;
        tre[reg] = dr;
        tim[reg] = di;
        const cplx o = e_old[0][0][reg];
        const double er = o.x - dr, ei = o.y - di;
        const double a = qf_modulus(er, ei);          // |dW_old - dW|                  (isospectral.py:526,534)
        double rsum = a;
        csum += a;
        rsum = qf_row16_sum(rsum);      // (xor butterfly 1, 2, 4, 8 on DPP: same tree, same bits as four __shfl_xor steps)
        if (r16 == 0) rs[wn * BM + li] = rsum;
    }
    if (offdiag) {
        double s2 = csum;
        s2 += __shfl_xor(s2, 16, 64);
        s2 += __shfl_xor(s2, 32, 64);
        if (q4 == 0) cs[wm * BN + wn * WTN + r16] = s2;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (tid < BM) {
        double s2 = 0.0;
#pragma unroll
        for (int cc = 0; cc < WN; ++cc) s2 += rs[cc * BM + tid];
        if (EXACT || i0 + tid < N)
            __hip_atomic_store(ep.rowpart + (size_t)tn * N + i0 + tid, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (open_if_large && s2 > tol_now) *open_flag = 1u;
    } else if (offdiag && tid >= 64 && tid < 64 + BN) {
        const int lj = tid - 64;
        double s2 = 0.0;
#pragma unroll
        for (int cc = 0; cc < WM; ++cc) s2 += cs[cc * BN + lj];
        if (EXACT || j0 + lj < N)
            __hip_atomic_store(ep.rowpart + (size_t)tm * N + j0 + lj, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (open_if_large && s2 > tol_now) *open_flag = 1u;
    }
    unsigned ticket_old = 0u;
    if (ep.fused && sx.deferred) {
        // deferred step end: the row sums are all this launch says about the exit test -- the next launch's workgroups
        // take the decision (k_solve / k_decide); nothing to drain, no ticket, no last finisher
        if (tid == 0 && t == 0 && !(sx.debug_drop & 2)) {
            sx.state_rw->pending_iter = guard.iter;
            sx.state_rw->pending = 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else if (ep.fused) {
        // the last of the n_tiles epilogues decides: every storing wave drains its row sums, one lane takes the
        // ticket; its answer is looked at behind the tile stores
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (tid == 0 && !((sx.debug_drop & 2) && t == 0))
            ticket_old = __hip_atomic_fetch_add(sx.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const bool speculate = ep.fused && !(open_for_sure || (open_if_large && *open_flag != 0u));
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int li = wm * WTM + q4 + 4 * reg;
        const int lj = wn * WTN + r16;
        const size_t e = (size_t)(i0 + li) * N + (j0 + lj);
        const cplx d = make_double2(tre[reg], tim[reg]);
        const cplx w = e_w[0][0][reg];
        const cplx wh = make_double2(w.x + d.x, w.y + d.y);      // Whalf = W + dW   (isospectral.py:481-482)
        const bool in = EXACT || (i0 + li < N && j0 + lj < N);
        if (in) {
            ep_dW_new[e] = d;
            ep.Whalf[e] = wh;
        }
        Th[li * TS + lj] = wh;
        if (speculate) {
            // should this be the step's last iteration: W_next = W + 2 comm (isospectral.py:547,592) and the
            // next step's first Whalf = W_next + dW
            const cplx wc = make_double2(w.x + 2.0 * e_c[0][0][reg].x, w.y + 2.0 * e_c[0][0][reg].y);
            const cplx whs = make_double2(wc.x + d.x, wc.y + d.y);
            if (in) {
                ep_Wnext[e] = wc;
                ep.Whalf_step[e] = whs;
            }
            Ts[li * TS + lj] = whs;
        } else if (!ep.fused) {
            Ts[li * TS + lj] = d;      // two-kernel protocol: k_update reads all of dW
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (offdiag) {
        // row j0+jl of the mirrored tile is column jl of this one (32 entries = 512 bytes): a wave writes two
        // such rows per instruction, eight in all
        const int il = lane & 31;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jl = wave * 8 + r * 2 + (lane >> 5);
            if (!EXACT && (j0 + jl >= N || i0 + il >= N)) continue;
            const cplx wv = Th[il * TS + jl];
            const size_t e2 = (size_t)(j0 + jl) * N + (i0 + il);
            ep.Whalf[e2] = make_double2(-wv.x, wv.y);       // -conj(Whalf[i,j])
            if (speculate || !ep.fused) {
                const cplx ws = Ts[il * TS + jl];
                if (ep.fused) ep.Whalf_step[e2] = make_double2(-ws.x, ws.y);
                else ep_dW_new[e2] = make_double2(-ws.x, ws.y);         // -conj(dW[i,j])
            }
        }
    }
    if (ep.fused && !sx.deferred) {
        unsigned *last_flag = reinterpret_cast<unsigned *>(rs);     // rs / cs have been consumed
        if (tid == 0) *last_flag = (ticket_old == (unsigned)(sx.n_tiles - 1)) ? 1u : 0u;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (*last_flag != 0u)
            qf_fused_step_end(N, nt, ep.rowpart, sx.ticket, sx.state_rw, sx.rec, guard.iter, tid, reinterpret_cast<double *>(smem_raw));
    }
}

#undef QF_KTILE_STEADY
#undef QF_KTILE_TAIL
#undef QF_KTILE_LAST
#undef QF_KTILE
#undef QF_KTILE_SPREAD
#undef QF_KTILE_PHASE1
#undef QF_STORE_A
#undef QF_STORE_B
#undef QF_EPI_FETCH
#undef QF_EPI_COMM
#undef QF_LOAD_TILE
#undef QF_STORE_TILE
#undef QF_READ_FRAGS
#undef QF_MFMA

struct gemm_cfg {
    int BM, BN;
};

gemm_cfg pick_gemm(int N)
{
    // Tile size of the full products (the first product of every iteration).  64 x 64 tiles run the software-pipelined
    // FAST path at 0.82-0.87 of the matrix peak -- when N is a multiple of 64 AND the tiles fill whole rounds of the 256
    // CUs (one workgroup per CU): 256 tiles at N = 1024, 1024 at N = 2048.  In between a launch's last round is mostly
    // empty (N = 1088: 289 tiles, two rounds for 1.13 rounds of work), and for N not a multiple of 64 the 64 x 64 kernel
    // falls back to its generic path with bounds checks (N = 1056: 274 us against 134).  The 32 x 32 kernel runs two to
    // three workgroups per CU and has no rounds to speak of: measured (first product us, 64 x 64 / 32 x 32): N = 768
    // 75.9 / 61.7, 832 82.0 / 66.7, 896 88.2 / 92.3, 960 94.7 / 99.3, 1088 204 / 137, 1152 216 / 170, 1280 239 / 218,
    // 1536 424 / 340, 1792 654 / 567; N % 32 == 0 only: 800 109 / 65, 928 124 / 99, 1056 275 / 134, 1248 325 / 197.
    // Rule: multiples of 32 take 32 x 32 tiles unless N >= 896 is a multiple of 64 whose tiles fill >= 85 % of their
    // rounds.
    gemm_cfg c;
    bool big;
    if (N % 32 != 0) {
        big = false;       // generic paths with bounds checks either way: 32 x 32 wins (N = 1000 1,690 -> 1,743 timesteps/s, N = 1500 404 -> 561)
    } else if (N % 64 != 0 || N < 896) {
        big = false;
    } else {
        const long long tiles = (long long)(N / 64) * (N / 64), rounds = (tiles + 255) / 256;
        big = tiles * 100 >= rounds * 256 * 85;
    }
    if (big) { c.BM = 64; c.BN = 64; }
    else { c.BM = 32; c.BN = 32; }
    return c;
}

template <int BM, int BN, int WM, int WN, bool EPI, bool EXACT, bool FUSED>
int launch4(qf_ctx *ctx, const cplx *A, const cplx *B, cplx *C, qf_epilogue ep, const qf_guard &guard)
{
    const int N = ctx->N;
    const int tiles_m = (N + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const size_t smem = tile_smem<BM, BN, EXACT && BM == 64 && BN == 64 && WM * WN == 4>::bytes;
    static qf_smem_attr attr;       // per instantiation
    QF_TRY(qf_smem_attr_set(attr, (const void *)k_zgemm<BM, BN, WM, WN, EPI, EXACT, FUSED>, ctx->device, smem));
    if (FUSED) {   // tile ticket + what the last tile's workgroup updates
        ep.ticket = ctx->ticket + 400;     // a word of the ticket area that k_update's counters never reach
        ep.n_tiles = tiles_m * tiles_n;
        ep.state_rw = ctx->state;
        ep.rec = ctx->host_rec;
        if (ctx->debug_drop == 2) {      // fault injection, one launch (the first of a call is always due)
            ep.debug_drop = 2;
            ctx->debug_drop = 0;
        }
    }
    dim3 grid(tiles_m * tiles_n), block(WM * WN * 64);
    qf_plan_note(ctx, 0x1000000ull | (unsigned long long)(BM << 8 | EPI << 2 | EXACT << 1 | (FUSED ? 1 : 0)),
                 "{\"kernel\": \"k_zgemm<%d,%d%s>\", \"arithmetic\": \"fp64 3M, v_mfma_f64_16x16x4_f64\", \"tile\": [%d, %d], "
                 "\"tiles\": %d, \"tile_share\": 1.0, \"workgroups\": %d, \"threads\": %d, \"exact_tiling\": %s, \"step_end\": \"%s\"}",
                 BM, BN, EPI ? ",EPI" : "", BM, BN, tiles_m * tiles_n, tiles_m * tiles_n, WM * WN * 64, EXACT ? "true" : "false",
                 !EPI ? "none" : FUSED ? "fused (last tile decides)" : "two-kernel");
    hipLaunchKernelGGL((k_zgemm<BM, BN, WM, WN, EPI, EXACT, FUSED>), grid, block, smem, ctx->stream, N, tiles_m,
                       tiles_n, A, B, C, ep, guard);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

template <int BM, int BN, int WM, int WN, bool EPI, bool EXACT>
int launch2(qf_ctx *ctx, const cplx *A, const cplx *B, cplx *C, const qf_epilogue &ep, const qf_guard &guard)
{
    if (EPI && ep.fused) return launch4<BM, BN, WM, WN, EPI, EXACT, EPI>(ctx, A, B, C, ep, guard);
    return launch4<BM, BN, WM, WN, EPI, EXACT, false>(ctx, A, B, C, ep, guard);
}

template <int BM, int BN, int WM, int WN>
int launch(qf_ctx *ctx, const cplx *A, const cplx *B, cplx *C, const qf_epilogue *ep, const qf_guard &guard)
{
    const int N = ctx->N;
    const bool exact = (N % BM == 0) && (N % BN == 0) && (N % BK == 0);
    qf_epilogue none;
    if (ep) {
        if (exact) return launch2<BM, BN, WM, WN, true, true>(ctx, A, B, C, *ep, guard);
        return launch2<BM, BN, WM, WN, true, false>(ctx, A, B, C, *ep, guard);
    }
    if (exact) return launch2<BM, BN, WM, WN, false, true>(ctx, A, B, C, none, guard);
    return launch2<BM, BN, WM, WN, false, false>(ctx, A, B, C, none, guard);
}

}  // namespace

int qf_gemm_tiles_n(int N)
{
    gemm_cfg c = pick_gemm(N);
    return (N + c.BN - 1) / c.BN;
}

int qf_launch_zgemm_tri(qf_ctx *ctx, const cplx *A, const cplx *B, const qf_epilogue *ep, qf_guard guard)
{
    const int N = ctx->N;
    if (!ep || N % 64 != 0 || !ctx->sk_partial || ctx->num_cus < 1) {
        qf_set_error("qf_launch_zgemm_tri: not available for this context (N=%d)", N);
        return QF_ERR_STATE;
    }
    static qf_smem_attr attr;
    QF_TRY(qf_smem_attr_set(attr, (const void *)k_zgemm_tri, ctx->device, TRI_SMEM_BYTES));
    const int nt = N / 64;
    // cost units: per tile its N/16 K-tiles + E units for the finisher's extra work (see the kernel)
    int E = ep->fused ? ctx->sk_epi_units_fused : ctx->sk_epi_units;
    if (E < -8) E = -8;   // (E < 0: finishers given MORE K-tiles than contributors; round 2 measured E = -2 at 2,527 against 2,551 timesteps/s)
    const long long units = (long long)nt * (nt + 1) / 2 * (N / BK + E);
    if (units > 0x7fffffffLL) {       // (the kernel keeps cost positions in 32 bits)
        qf_set_error("qf_launch_zgemm_tri: N=%d is too large for the stream-K partition", N);
        return QF_ERR_INVALID;
    }
    // one workgroup per CU, all resident (the LDS footprint allows one per CU): see the kernel header
    // (short products: at least sk_min_units K-tiles per workgroup, or the exchange dominates)
    long long grid_ll = units / (ctx->sk_min_units > 0 ? ctx->sk_min_units : 1);
    if (grid_ll > ctx->num_cus) grid_ll = ctx->num_cus;
    if (grid_ll < 1) grid_ll = 1;
    const int grid = (int)grid_ll;
    qf_streamk sk;
    const int slots = ctx->sk_slots > 0 ? ctx->sk_slots : ctx->num_cus;
    sk.slots = slots;
    sk.partial = ctx->sk_partial;
    sk.flags = ctx->sk_flags;
    sk.epoch = ++ctx->sk_epoch;
    sk.fault = &ctx->host_rec->fault;
    sk.ticket = ctx->sk_flags + slots;            // one word behind the per-slot flags
    sk.n_tiles = nt * (nt + 1) / 2;
    sk.state_rw = ctx->state;
    sk.rec = ctx->host_rec;
    if (ctx->debug_drop == 1 || (ctx->debug_drop == 2 && ep->fused)) {   // fault injection, one launch
        sk.debug_drop = ctx->debug_drop;
        sk.spin_limit = 1u << 14;        // the injected wait gives up after ~20 ms instead of seconds
        ctx->debug_drop = 0;
    }
    qf_plan_note(ctx, 0x2000000ull | (unsigned long long)(grid << 8 | (E & 0x3f) << 1 | (ep->fused ? 1 : 0)),
                 "{\"kernel\": \"k_zgemm_tri\", \"arithmetic\": \"fp64 3M, v_mfma_f64_16x16x4_f64\", \"tile\": [64, 64], \"tiles\": %d, "
                 "\"tile_share\": %.6f, \"workgroups\": %d, \"threads\": 256, \"partition\": \"stream-K over (tile, K-tile) units, "
                 "contiguous ranges, epilogue weight E = %d K-tiles\", \"step_end\": \"%s\"}",
                 sk.n_tiles, (double)sk.n_tiles / ((double)nt * nt), grid, E, ep->fused ? "fused (last finisher decides)" : "two-kernel");
    hipLaunchKernelGGL(k_zgemm_tri, dim3(grid), dim3(256), TRI_SMEM_BYTES, ctx->stream, N, nt, (int)units, E, A, B, *ep, guard, sk);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_zgemm_tri32(qf_ctx *ctx, const cplx *A, const cplx *B, const qf_epilogue *ep, qf_guard guard)
{
    const int N = ctx->N;
    if (!ep || N < 64 || !ctx->t32_partial || !ctx->t32_arrive) {
        qf_set_error("qf_launch_zgemm_tri32: not available for this context (N=%d)", N);
        return QF_ERR_STATE;
    }
    const int nt = (N + 31) / 32;
    qf_tri32 sx;
    sx.partial = ctx->t32_partial;
    sx.arrive = ctx->t32_arrive;
    sx.split = (ctx->tri32_split == 2 || ctx->tri32_split == 4) ? ctx->tri32_split : 1;
    sx.split_diag = (ctx->tri32_split_diag == 2 || ctx->tri32_split_diag == 4) ? ctx->tri32_split_diag : 1;
    // (a piece is at least two K-tiles)
    while (sx.split > 1 && (N + BK - 1) / BK / sx.split < 2) sx.split >>= 1;
    while (sx.split_diag > 1 && (N + BK - 1) / BK / sx.split_diag < 2) sx.split_diag >>= 1;
    sx.ticket = ctx->ticket + 401;      // (k_zgemm<.., FUSED> owns word 400)
    sx.n_tiles = nt * (nt + 1) / 2;
    sx.state_rw = ctx->state;
    sx.rec = ctx->host_rec;
    if (ctx->debug_drop == 2 && ep->fused) {     // fault injection, one launch
        sx.debug_drop = 2;
        ctx->debug_drop = 0;
    }
    sx.deferred = (ep->fused && ctx->defer && guard.state) ? 1 : 0;
    const int grid = nt * sx.split_diag + nt * (nt - 1) / 2 * sx.split;
    const size_t smem = tile_smem<32, 32, false>::bytes;
    qf_plan_note(ctx, 0x3000000ull | (unsigned long long)(grid << 6 | sx.split << 3 | sx.split_diag << 1 | sx.deferred),
                 "{\"kernel\": \"k_zgemm_tri32<%s>\", \"arithmetic\": \"fp64 3M, v_mfma_f64_16x16x4_f64\", \"tile\": [32, 32], \"tiles\": %d, "
                 "\"tile_share\": %.6f, \"workgroups\": %d, \"threads\": 256, \"partition\": \"K range of an off-diagonal / diagonal tile in "
                 "%d / %d pieces, one workgroup each\", \"step_end\": \"%s\"}",
                 N % 32 == 0 ? "exact" : "guarded edges", sx.n_tiles, (double)sx.n_tiles / ((double)nt * nt), grid, sx.split, sx.split_diag,
                 !ep->fused ? "two-kernel" : sx.deferred ? "deferred (the next solve's workgroups decide)" : "fused (last tile decides)");
    if (N % 32 == 0) hipLaunchKernelGGL(k_zgemm_tri32<true>, dim3(grid), dim3(256), smem, ctx->stream, N, nt, A, B, *ep, guard, sx);
    else hipLaunchKernelGGL(k_zgemm_tri32<false>, dim3(grid), dim3(256), smem, ctx->stream, N, nt, A, B, *ep, guard, sx);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_zgemm(qf_ctx *ctx, const cplx *A, const cplx *B, cplx *C, const qf_epilogue *ep, qf_guard guard)
{
    if (ep && ctx->gemm_tri) return qf_launch_zgemm_tri(ctx, A, B, ep, guard);
    if (ep && ctx->gemm_tri32) return qf_launch_zgemm_tri32(ctx, A, B, ep, guard);
    gemm_cfg c = pick_gemm(ctx->N);
    if (c.BM == 64) return launch<64, 64, 2, 2>(ctx, A, B, C, ep, guard);
    return launch<32, 32, 2, 2>(ctx, A, B, C, ep, guard);
}
