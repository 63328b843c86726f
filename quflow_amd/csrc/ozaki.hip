// Complex128 N x N x N product on the INT8 matrix cores (v_mfma_i32_32x32x32_i8) by digit
// splitting ("Ozaki scheme"), for the two products of the isospectral iteration
// (quflow/integrators/isospectral.py:496,499) -- BASELINE.json config 3's "low-precision-MFMA
// commutator with fp64 Laplacian", built on the int8 rather than the bf16 pipe:
//   * the i8 MFMA runs at twice the bf16 rate (32 cycles for 32x32x32) and accumulates in int32,
//     so every digit product is EXACT for any N here (a bf16 digit product is exact in fp32 only
//     while N 64 64 < 2^24);
//   * an operand costs 10 bytes per complex entry (5 digits x {re, im}) -- less than the 16 bytes
//     of the complex128 itself, where bf16 digits would cost 20.
//
// The kernels are templates on the number of digits KD: 5 (QUFLOW_HIP_GEMM=i8; the figures quoted
// below) or 6 (i8x6: 12 bytes per entry, 21 digit pairs, 2^-42 -- the fp64 fixtures at 1e-11).
//
// Numerics.  Row i of an operand is scaled by a power of two s >= (128/63) max(|re|,|im|) and cut into
// KD base-128 digits from the non-redundant balanced set [-64, 63]: x/s = sum_t d_t 128^-(t+1) + r,
// |r| <= 2^-36 (KD = 5).  The right operand of both products is a skew-Hermitian matrix M (Whalf, then Phalf), B[k][j] = -conj(M[j][k]), so it is sliced by ROWS
// like the left one and one sliced copy of Phalf serves as the left operand of the first product
// and the right operand of the second.  With a = ar + i ai (row of A), m = mr + i mi (row of M):
//     U1 = ar.mr   U2 = ai.mi   U3 = (ar + ai).(mi - mr)      (three real products: "3M")
//     Re(AB) = -U1 - U2        Im(AB) = U3 + U1 - U2
// Each U is a sum over digit pairs (a,b), a+b < KD, of exact int8 GEMMs; pairs of equal a+b share
// an int32 accumulator (|sum| <= 6 N 2^14 < 2^31 for N <= 4096, also for the sum / difference digits in [-128, 127]).  The planes hold the digits in OFFSET form,
// x = d + 64 in [0, 127]: byte-wise sums of two planes then never carry across bytes, so the digits
// of ar+ai and mi-mr are formed IN REGISTERS from the re / im fragments with plain 32-bit adds
//     (xr + xi) ^ 0x80..            = (dr + di)          as int8 (2 VALU per dword)
//     (xi + (xr ^ 0x7f..) + 0x01..) ^ 0x80.. = (di - dr) as int8 (3 VALU per dword)
// and no third plane is stored, moved or read.  U1 and U2 are multiplied on the offset bytes as
// they are; the exact integer correction  sum_k (a+64)(m+64) - a m = 64 sum a + 64 sum m + 4096 N
// comes from per-row digit sums the slicing kernel leaves next to the scales (prefix sums over the
// digit index: one int per accumulator group) and is taken off in the epilogue.  The only error is
// the truncation of the digit series: relative 2^-35 of (row scale x column scale), i.e. the
// accuracy the stepper needs (tools/bf16_split_study.py; the fixed-point tolerance is sqrt(eps)).
//
// Layout.  A sliced operand is stored [row][N/16 k-groups][10 planes][16 bytes]: the 320 bytes a
// workgroup needs of one row for one K-step (32 k) are contiguous, and one ds_read_b128 hands a
// lane its whole MFMA fragment (lane l: row l&31, k = 16 (l>>5) .. +15; probed with exact integer
// data, tools/i8_lanemap_probe.hip, tools/i8_mfma_rate.hip).
//
// Kernel.  64x64 output tile per workgroup, four waves of one 32x32 MFMA tile each; per K-step:
// 20 fragment reads, 100 VALU for the sum / difference fragments and 45 MFMAs (1440 matrix-pipe
// cycles) per wave, placed by hand in the MFMA gaps (sched_barrier between slots); accumulators =
// 15 groups x 16 registers (AGPRs: this file is compiled WITHOUT -amdgpu-mfma-vgpr-form).  Staging
// is LDS-DMA (buffer_load_dwordx4 ... lds: global -> LDS with no register in between; probed in
// tools/dma_probe.hip): an LDS stage is the lane-linear image of 42 wave-instructions = 128 rows x 21
// pieces of 16 bytes (20 data + 1 pad piece per row: bank-conflict-free b128 fragment reads), three
// stages, the DMA pieces of K-step kt+2 issued during K-step kt (every other MFMA gap), retired by
// a counted vmcnt before the one barrier of K-step kt+1 (cdna_hip_programming.md 5, "Pipelining
// across barriers").  Tiles are handed out XCD-aware (compact 4 x 8 tile blocks per XCD).
#include "qf_internal.h"
#include "qf_step_end.h"

// timing-only ablation knobs of the diagnostic build (tools/oz_probe.hip); results are wrong when set
#ifndef OZ_ABL_NOLOAD
#define OZ_ABL_NOLOAD 0     // no LDS-DMA in the K loop
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
#ifndef OZ_STAMP
#define OZ_STAMP 0          // s_memtime stamps (prologue, K loop, epilogue) written over the result
#endif
#ifndef OZ_ABL_NOSWAR
#define OZ_ABL_NOSWAR 0     // third product on the re fragments (no byte-wise add / subtract)
#endif

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int OZ_BK = 32;                 // K per step = one i8 MFMA
constexpr int OZ_T = 64;                  // tile edge
constexpr int STAGES = 3;

// sizes that follow from the number of digits KD per real value (5: the default "i8"; 6: "i8x6")
template <int KD> struct ozc {
    static constexpr int PLANES = 2 * KD;                 // {re, im} x digits
    static constexpr int GROUP_BYTES = PLANES * 16;       // one row, one k-group of 16: 160 / 192 bytes
    static constexpr int ROW_PIECES = 2 * PLANES + 1;     // 16-byte pieces of an LDS row: data + 1 pad (21 / 25)
    static constexpr int ROW_LDS = ROW_PIECES * 16;       // 336 / 400: conflict-free b128 fragment reads
    static constexpr int STAGE_INSTR = 2 * OZ_T * ROW_PIECES / 64;    // wave-wide DMA instructions per stage (42 / 50)
    static constexpr int DMA_PER_WAVE = (STAGE_INSTR + 3) / 4;        // 11 / 13 (the surplus ones land in slack)
    static constexpr int STAGE_BYTES = 4 * DMA_PER_WAVE * 1024;       // 45,056 / 53,248
    static constexpr size_t SMEM = (size_t)STAGES * STAGE_BYTES;      // 135,168 / 159,744 (of 163,840)
    static constexpr int PAIRS = KD * (KD + 1) / 2;                   // MFMAs per real product and K-step (15 / 21)
    static_assert(2 * OZ_T * ROW_PIECES % 64 == 0 && OZ_T * ROW_PIECES % 64 == 0, "A and M rows fall on whole DMA instructions");
    static_assert((ROW_LDS / 4) % 8 == 4, "row pitch = 4 mod 8 dwords: 16 consecutive rows cover the 64 banks");
    static_assert(SMEM <= 160 * 1024, "three stages fit the LDS");
};

// ---- slicing: one workgroup per (job, row), one lane per 4 entries.  The lane keeps its entries
// in registers, the row maximum (wave shuffles + one LDS exchange) gives the power-of-two scale
// s >= (128/63) max(|re|,|im|), each lane cuts its 8 reals into 5 digits (one packed dword per plane, offset
// form d + 64) into an LDS image of the row's planes, which then leaves in coalesced 16-byte stores.
// The row record `scale` = [N] scales (double) followed by [N][10] int32: for {re, im} and s < 5 the
// digit sums  sum_{t <= s} sum_k d_t(row, k)  (the offset correction of accumulator group s).
__device__ __forceinline__ unsigned wave_sum_packed(unsigned x)      // sum over the 64 lanes (no field may overflow)
{
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xb1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4e, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, true);    // row_ror:4
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, true);    // row_ror:8
    return (unsigned)__builtin_amdgcn_readlane((int)x, 0) + (unsigned)__builtin_amdgcn_readlane((int)x, 16) +
           (unsigned)__builtin_amdgcn_readlane((int)x, 32) + (unsigned)__builtin_amdgcn_readlane((int)x, 48);
}

// PAIR (jobs.diag != nullptr, two jobs: A = jobs.j[0], M = jobs.j[1], the operands of one product A @ M with M taken by
// rows): one workgroup cuts row i of BOTH and, on the way, forms in fp64
//     diag[i] = Im (A @ M)_ii = sum_k ( Re A_ik Im M_ik - Im A_ik Re M_ik )            (B[k][i] = -conj(M[i][k]))
// for the product's epilogue to store in place of the digit sum's value.  Why: for skew-Hermitian A and M the terms
// (i, k) and (k, i) of sum_i Im (A @ M)_ii cancel exactly, and the trace of the commutator PW - PW^H the stepper adds to
// W (isospectral.py:509,547) is 2i times that sum.  The truncated digit series keeps the cancellation only between rows
// that share their power-of-two scales (tools/i8_trace_sim.py: three scales among the rows of Phalf at N = 1024 / 2048):
// what is left, 1e-15 per step and the same from step to step, is a LINEAR drift of tr W -- the one Casimir that is linear
// in the rounding (round 4: -1.4e-15 per step at N = 2048, 3.4e-12 after 2,000 steps against the fp64 products' 1e-16).
// N of the N^2 entries in fp64, from rows this kernel holds in registers anyway, put tr W back on the fp64 products' line.
template <int KD>
__global__ __launch_bounds__(1024) void k_oz_slice(int N, qf_oz_jobs jobs, qf_guard guard)
{
    if (!qf_guard_iter(guard)) return;
    constexpr int K_DIG = KD, PLANES = ozc<KD>::PLANES, GROUP_BYTES = ozc<KD>::GROUP_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);                 // [16] row maximum per wave, [16] dot product per wave
    unsigned *isum = reinterpret_cast<unsigned *>(smem + 256);        // [16 waves][PLANES]
    unsigned *img = reinterpret_cast<unsigned *>(smem + 256 + 16 * PLANES * 4);   // [N/16][PLANES][4] dwords
    const bool pair = jobs.diag != nullptr;
    const int row = pair ? (int)blockIdx.x : (int)(blockIdx.x % N);
    const int tid = threadIdx.x, nwaves = blockDim.x >> 6;
    const bool active = 4 * tid < N;
    double keep[8] = {0, 0, 0, 0, 0, 0, 0, 0};                      // PAIR: this lane's entries of A's row
    for (int jj = 0; jj < (pair ? 2 : 1); ++jj) {
    const int job = pair ? jj : (int)(blockIdx.x / N);
    const qf_oz_job jb = jobs.j[job];
    const cplx *X = jb.X;
    if (jb.X_alt && guard.state && guard.state->wh_sel) X = jb.X_alt;   // fused protocol: next step's Whalf
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (active) {
        const double4 *src = reinterpret_cast<const double4 *>(X + (size_t)row * N + 4 * tid);
        const double4 p0 = src[0], p1 = src[1];
        v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w;
        v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
    }
    double m = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmax(m, fabs(v[j]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off, 64));
    if (pair && jj == 1) {       // the diagonal entry's imaginary part: lane partial, wave tree, waves in order
        double d = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) d += keep[2 * j] * v[2 * j + 1] - keep[2 * j + 1] * v[2 * j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
        if ((tid & 63) == 0) red[16 + (tid >> 6)] = d;
    }
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
    for (int w = 1; w < nwaves; ++w) m = fmax(m, red[w]);
    if (pair && jj == 1 && tid == 0) {
        double d = red[16];
        for (int w = 1; w < nwaves; ++w) d += red[16 + w];
        jobs.diag[row] = d;
    }
    // Scale: the smallest power of two with max / s <= 63/128 (round 5; until then s >= 4 max, i.e. a leading digit in
    // [-32, 32] only).  What the truncated series drops -- the digit pairs a + b >= KD -- is of the size of the SCALES'
    // product whatever the entries' size (the low digits of any number are uniform in [-64, 63]), so one bit per
    // operand is a factor four in the product's error: the smooth initial data IC-B (a large stream function: scales
    // 250 x 16 times the white-noise case's) sat 3e-11 from the CPU restatement after two steps at N = 1024.
    int e = 0;
    if (m > 0.0 && m < 1e300) {
        const double f = frexp(m, &e);      // m = f 2^e, f in [0.5, 1)
        e += (f <= 0.984375) ? 1 : 2;       // y = x/s in [-63/128, 63/128]
    }
    const double s = ldexp(1.0, e), inv_s = ldexp(1.0, -e);
    if (tid == 0) jb.scale[row] = s;
    // Digits.  The offset bytes d_t + 64 of the balanced digits d_t in [-64, 63] are the plain base-128 digits of
    // z = y + B, B = sum_t 64 128^-(t+1) = 0.5039... (z in (0.011, 0.997)):
    // one fma puts z + 2^E, E = 52 - 7 KD, into a double whose ulp is 128^-KD, i.e. rounds y to the
    // KD-digit grid, and the 7 KD low mantissa bits ARE the digits.
    double B = 0.0;
#pragma unroll
    for (int t = 0; t < K_DIG; ++t) B += ldexp(1.0, -1 - 7 * t);
    const double C = ldexp(1.0, 52 - 7 * K_DIG) + B;
    unsigned w[2][K_DIG];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < K_DIG; ++t) w[c][t] = 0u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(fma(v[2 * j + c], inv_s, C));
#pragma unroll
            for (int t = 0; t < K_DIG; ++t)
                w[c][t] |= ((unsigned)(bits >> (7 * (K_DIG - 1 - t))) & 127u) << (8 * j);
        }
    if (active) {
        const int g = tid >> 2, q = tid & 3;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < K_DIG; ++t) img[(g * PLANES + c * K_DIG + t) * 4 + q] = w[c][t];
    }
    // digit sums of the row: byte sums (v_sad_u8) of the packed words, two 16-bit fields per word
    // through the wave reduction (64 lanes x 4 x 127 < 2^16)
#pragma unroll
    for (int i = 0; i < PLANES / 2; ++i) {
        unsigned x = 0u;
        if (active) {
            const unsigned s0 = __builtin_amdgcn_sad_u8(w[(2 * i) / K_DIG][(2 * i) % K_DIG], 0u, 0u);
            const unsigned s1 = __builtin_amdgcn_sad_u8(w[(2 * i + 1) / K_DIG][(2 * i + 1) % K_DIG], 0u, 0u);
            x = s0 | (s1 << 16);
        }
        x = wave_sum_packed(x);
        if ((tid & 63) == 0) {
            isum[(tid >> 6) * PLANES + 2 * i] = x & 0xffffu;
            isum[(tid >> 6) * PLANES + 2 * i + 1] = x >> 16;
        }
    }
    __syncthreads();
    if (tid < 2) {       // sum of d = sum of bytes - 64 N; prefix sums over the digit index, re (tid 0) and im (tid 1)
        int *rs = reinterpret_cast<int *>(jb.scale + N) + (size_t)row * PLANES + tid * K_DIG;
        int run = 0;
        for (int t = 0; t < K_DIG; ++t) {
            for (int w = 0; w < nwaves; ++w) run += (int)isum[w * PLANES + tid * K_DIG + t];
            run -= 64 * N;
            rs[t] = run;
        }
    }
    const int pieces = (N / 16) * PLANES;
    v4u *out = reinterpret_cast<v4u *>(jb.planes + (size_t)row * (N / 16) * GROUP_BYTES);
    const v4u *im4 = reinterpret_cast<const v4u *>(img);
    for (int i = tid; i < pieces; i += blockDim.x) out[i] = im4[i];
    if (pair && jj == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) keep[j] = v[j];
        __syncthreads();         // the row image and the sums are free for the second operand
    }
    }
}

// offset bytes x = d + 64 in [0, 127]: byte-wise sums do not carry across bytes
__device__ __forceinline__ int off_add(int xr, int xi)       // int8 digits of (dr + di)
{
    return (int)(((unsigned)xr + (unsigned)xi) ^ 0x80808080u);
}
__device__ __forceinline__ int off_sub(int xi, int xr)       // int8 digits of (di - dr)
{
    return (int)(((unsigned)xi + ((unsigned)xr ^ 0x7f7f7f7fu) + 0x01010101u) ^ 0x80808080u);
}

// LDS-DMA of 16 bytes per lane: LDS destination = dst + 16 lane (wave-uniform dst), source per lane.
// (A non-template wrapper: hipcc drops the host stub of a kernel TEMPLATE that calls this builtin with
// template-dependent arguments.)
__device__ __forceinline__ void oz_dma16(__amdgpu_buffer_rsrc_t rsrc, lds_void *dst, unsigned voffset, unsigned soffset)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, soffset, 0, 0);
}

__device__ __forceinline__ void oz_dma16_sc1(__amdgpu_buffer_rsrc_t rsrc, lds_void *dst, unsigned voffset, unsigned soffset)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voffset, soffset, 0, 16);      // sc1: past this XCD's L2
}
#ifndef OZ_SPIN_LIMIT
#define OZ_SPIN_LIMIT (1u << 22)
#endif

// MFMA order of a sweep over the digit pairs (a, b), a + b < KD: a descending, b ascending -- two
// consecutive MFMAs never share an accumulator (a + b)
__device__ constexpr int OZ_PA5[15] = {4, 3, 3, 2, 2, 2, 1, 1, 1, 1, 0, 0, 0, 0, 0};
__device__ constexpr int OZ_PB5[15] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};
__device__ constexpr int OZ_PA6[21] = {5, 4, 4, 3, 3, 3, 2, 2, 2, 2, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0};
__device__ constexpr int OZ_PB6[21] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 5};

// ---- the product.  C = A @ B, B[k][j] = -conj(M[j][k]) (M skew-Hermitian: B = M), from the
// row-sliced planes of A and M (pa / pm) and their row records (sa / sm: scales, then digit sums).
// FUSEDEPI: the second product of an iteration with the fused epilogue and step end of
// k_zgemm<.., FUSED> (zgemm.hip): dW = C + (PW - PW^H), Whalf = W + dW, the
// speculative next state / next-step Whalf, the residual row sums, the tile ticket and the decision.
// KDM: digits of M's STORAGE layout.  KDM > KD (6 / 5: the second product of QUFLOW_HIP_GEMM=i8x65): M was sliced with
// six digits for the first product; this product multiplies its five leading digits, gathered from the six-digit
// planes by the staging addresses (the LDS image and everything behind it is the five-digit kernel's).  The leading
// five of six digits are the value truncated (not rounded) to 128^-5: |r| <= 2^-35 instead of 2^-36.
template <int KD, bool FUSEDEPI, int KDM = KD>
__global__ __launch_bounds__(256) void k_oz_gemm(int N, const signed char *__restrict__ pa, const double *__restrict__ sa,
                                                  const signed char *__restrict__ pm, const double *__restrict__ sm,
                                                  cplx *__restrict__ C, qf_epilogue ep, qf_guard guard, qf_oz_mirror mir)
{
    if (!qf_guard_iter(guard)) return;
#if OZ_STAMP
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    using cfg = ozc<KD>;
    constexpr int K_DIG = KD, PLANES = cfg::PLANES, GROUP_BYTES = cfg::GROUP_BYTES, ROW_PIECES = cfg::ROW_PIECES;
    constexpr int ROW_LDS = cfg::ROW_LDS, STAGE_INSTR = cfg::STAGE_INSTR, DMA_PER_WAVE = cfg::DMA_PER_WAVE;
    constexpr int STAGE_BYTES = cfg::STAGE_BYTES, PAIRS = cfg::PAIRS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int tiles = N / OZ_T;
    // workgroup -> tile, XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs (own L2
    // each); XCD x works on a compact (tiles/4) x (tiles/2) part of the tile grid, in 4 x 8 blocks,
    // so that the workgroups running together on an XCD share 4 row panels and 8 column panels
    int tm = blockIdx.x / tiles, tn = blockIdx.x % tiles;
    // MIRROR (second product only; T = PW@Phalf is skew-Hermitian): the tiles on and above the
    // diagonal come first in the grid and multiply; a tile below the diagonal waits for its partner's
    // result tile (oz_tbuf, published with write-through stores + a launch-epoch flag, read past the
    // L2 -- the protocol of k_zgemm_tri) and only runs the epilogue on T[i][j] = -conj(T[j][i]).
    // Workgroups are dispatched in order, so a waiting workgroup's partner is running or done.
    const bool mirror = FUSEDEPI && mir.epoch != 0u;
    bool lower = false;
    int upair = 0;                       // index of the upper-triangle tile (this one or the partner)
    if (mirror) {
        const int nup = tiles * (tiles + 1) / 2;
        int b = blockIdx.x;
        lower = b >= nup;
        int row = 0;
        {   // XCD-aware order inside each of the two groups: workgroup ids go round-robin over the 8
            // XCDs; XCD x takes a CONTIGUOUS range of the row-major tile list (one or two row panels
            // shared by its tiles) instead of every eighth tile
            const int cnt = lower ? nup - tiles : nup, b0 = lower ? b - nup : b;
            const int slot = b0 & 7;                          // this workgroup's XCD, counted from the group's first
            const int l = b0 >> 3;                            // how many of that XCD's workgroups came before
            int start = 0;
            for (int y = 0; y < slot; ++y) start += (cnt - y + 7) >> 3;
            b = (lower ? nup : 0) + start + l;
        }
        if (!lower) {
            upair = b;
            while (b >= tiles - row) { b -= tiles - row; ++row; }
            tm = row;
            tn = row + b;
        } else {
            b -= nup;                    // index among the strictly upper tiles, row-major
            while (b >= tiles - 1 - row) { b -= tiles - 1 - row; ++row; }
            const int col = row + 1 + b;
            upair = row * tiles - row * (row - 1) / 2 + (col - row);
            tm = col;                    // the mirrored tile
            tn = row;
        }
    } else if (tiles % 16 == 0) {
        const int x = blockIdx.x & 7, l = blockIdx.x >> 3;
        const int pw_ = tiles / 2;                      // part width in tiles (height tiles / 4)
        const int blk = l >> 5, in = l & 31;            // 4 x 8 block of the part, tile inside it
        const int bpr = pw_ / 8;                        // blocks per part row
        tm = (x >> 1) * (tiles / 4) + (blk / bpr) * 4 + (in >> 3);
        tn = (x & 1) * pw_ + (blk % bpr) * 8 + (in & 7);
    }
    const int i0 = tm * OZ_T, j0 = tn * OZ_T;
    static_assert(KDM >= KD && (KDM == KD || KD <= 5), "M's layout holds at least the digits multiplied (cm in registers)");
    constexpr int GROUP_BYTES_M = 32 * KDM;            // one row, one k-group of 16 in M's storage layout
    const int row_bytes = (N / 16) * GROUP_BYTES;      // one row of a sliced operand (A)
    const int row_bytes_m = (N / 16) * GROUP_BYTES_M;
    const int KT = N / OZ_BK;

    // staging map: wave w issues the DMA instructions w, w+4, ...; instruction n writes the 64
    // pieces [64 n, 64 n + 64) of the stage image, piece P = (row P / 21, piece P % 21); rows
    // 0..63 are A's (instructions 0..20), rows 64..127 M's (21..41); 42, 43 are dummies into slack
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<signed char *>(pa), 0, (int)((size_t)N * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<signed char *>(pm), 0, (int)((size_t)N * row_bytes_m), 0x00020000);
    unsigned voff[DMA_PER_WAVE];
#pragma unroll
    for (int q = 0; q < DMA_PER_WAVE; ++q) {
        const int n = wave + 4 * q;
        int P = n * 64 + lane;
        if (P >= 2 * OZ_T * ROW_PIECES) P -= 2 * OZ_T * ROW_PIECES;     // dummy instructions re-read the first rows
        const int row = P / ROW_PIECES;
        int piece = P % ROW_PIECES;
        if (piece > ROW_PIECES - 2) piece = ROW_PIECES - 2;               // the pad piece repeats the last one
        if (row < OZ_T) {
            voff[q] = (unsigned)((size_t)(i0 + row) * row_bytes + piece * 16);
        } else {
            // piece -> (k-group of the K-step, {re, im}, digit) -> its plane in M's storage layout
            const int kg = piece / PLANES, pl = piece % PLANES;
            const int src_plane = (pl / K_DIG) * KDM + pl % K_DIG;
            voff[q] = (unsigned)((size_t)(j0 + row - OZ_T) * row_bytes_m + kg * GROUP_BYTES_M + src_plane * 16);
        }
    }
    // one DMA instruction q of K-step kt_ into stage st_ (past the last K-step: re-reads the last
    // one into a stage nobody reads any more -- the loop stays branch-free)
#define OZ_DMA1(q_, kt_, st_)                                                          \
    if (!OZ_ABL_NOLOAD) {                                                              \
        const int kk_ = (kt_) < KT ? (kt_) : KT - 1;                                   \
        const int n_ = wave + 4 * (q_);                                                \
        const bool isA_ = (n_ < STAGE_INSTR / 2) || (n_ >= STAGE_INSTR);               \
        oz_dma16(isA_ ? ra : rm, (lds_void *)(smem + (st_) * STAGE_BYTES + n_ * 1024), voff[q_],                    \
                 (unsigned)kk_ * (unsigned)(isA_ ? 2 * GROUP_BYTES : 2 * GROUP_BYTES_M));                          \
    }

    // Accumulators: 3 K_DIG groups of 16 registers.  5 digits: 240, all in AGPRs.  6 digits would be
    // 288 > 256 AGPRs (the register allocator then shuffles whole groups every K-step, 130-350 moves):
    // the three leading-pair groups (s = 0, ONE MFMA per K-step each) are kept as plain VGPR sums S0
    // instead, fed through one transient accumulator T0 (MFMA with C = 0) that is added to its sum a
    // few gaps later, two elements per gap -- 15 + 1 groups = 256 AGPRs exactly.
    constexpr bool PARK0 = K_DIG >= 6;
    v16i acc[3][K_DIG];
    v16i S0[3], T0;
#pragma unroll
    for (int tau = 0; tau < 3; ++tau)
#pragma unroll
        for (int s = 0; s < K_DIG; ++s)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[tau][s][q] = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) S0[0][q] = S0[1][q] = S0[2][q] = T0[q] = 0;
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    const unsigned fragA = (unsigned)((wm * 32 + r) * ROW_LDS + h * GROUP_BYTES);
    const unsigned fragB = (unsigned)((OZ_T + wn * 32 + r) * ROW_LDS + h * GROUP_BYTES);

    // One K-step, 3 PAIRS MFMA slots (45 / 63), the other work placed in the gaps (FR / FI: the re /
    // im digit fragments of this K-step, in registers since the previous K-step):
    //   slots of the first two sweeps (U1 += FR.a x FR.b, then U2 += FI.a x FI.b): the DMA pieces of
    //     K-step kt+2 in every other gap, the 8 KD sum / difference dwords spread over the other gaps
    //   counted vmcnt (this wave's pieces of K-step kt+1 have landed), barrier (everybody's have)
    //   slots of the third sweep (U3 += FS.a x FS.b): the 4 KD fragment reads of K-step kt+1, two per gap
    // All loops below are fully unrolled: every index is a constant by the time registers are assigned.
    struct frag_t { v4i a[K_DIG], b[K_DIG]; };
    frag_t FR, FI, FS;
    constexpr int NSW = 8 * K_DIG;                      // sum / difference dwords per K-step
    constexpr int NGAP = 2 * PAIRS - DMA_PER_WAVE;      // gaps that carry them
    constexpr int NFR = 4 * K_DIG;                      // fragment reads per K-step
    static_assert(2 * DMA_PER_WAVE - 1 <= 2 * PAIRS && NFR <= 2 * PAIRS, "the gaps hold the DMA pieces and the reads");
#define OZ_PAIR_A(i_) (K_DIG == 5 ? OZ_PA5[(i_) < 15 ? (i_) : 0] : OZ_PA6[(i_)])
#define OZ_PAIR_B(i_) (K_DIG == 5 ? OZ_PB5[(i_) < 15 ? (i_) : 0] : OZ_PB6[(i_)])
#define OZ_MFMA(F_, tau_, i_)                                                          \
    {                                                                                  \
        const int a_ = OZ_PAIR_A(i_), b_ = OZ_PAIR_B(i_);                              \
        if (PARK0 && a_ + b_ == 0)                                                     \
            T0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(F_.a[0], F_.b[0], zero16, 0, 0, 0); \
        else if (!OZ_ABL_NOMFMA)                                                       \
            acc[tau_][a_ + b_] = __builtin_amdgcn_mfma_i32_32x32x32_i8(F_.a[a_], F_.b[b_], acc[tau_][a_ + b_], 0, 0, 0); \
        else acc[tau_][a_ + b_][0] += F_.a[a_][0] ^ F_.b[b_][0];                       \
    }
#define OZ_FRAG(j_, base_)    /* fragment read j_ of 4 KD: re a, re m, im a, im m by digit */ \
    if (!OZ_ABL_NOFRAG) {                                                              \
        const int d_ = (j_) % K_DIG, w_ = (j_) / K_DIG;                                \
        const v4i v_ = *reinterpret_cast<const v4i *>((base_) + ((w_ & 1) ? fragB : fragA) + ((w_ >> 1) * K_DIG + d_) * 16); \
        if (w_ == 0) FR.a[d_] = v_;                                                    \
        if (w_ == 1) FR.b[d_] = v_;                                                    \
        if (w_ == 2) FI.a[d_] = v_;                                                    \
        if (w_ == 3) FI.b[d_] = v_;                                                    \
    }
#define OZ_PARK(tau_, gap_)    /* gap gap_ of the sweep AFTER sweep tau_: two elements of S0[tau_] += T0 */ \
    if (PARK0 && (gap_) >= 2 && (gap_) < 10) {                                         \
        S0[tau_][2 * ((gap_) - 2)] += T0[2 * ((gap_) - 2)];                            \
        S0[tau_][2 * ((gap_) - 2) + 1] += T0[2 * ((gap_) - 2) + 1];                    \
    }
#define OZ_FENCE() __builtin_amdgcn_sched_barrier(0)

#if OZ_STAMP
    unsigned long long t_loop = 0;
#endif
    if (!lower) {
#pragma unroll
    for (int q = 0; q < DMA_PER_WAVE; ++q) OZ_DMA1(q, 0, 0)
#pragma unroll
    for (int q = 0; q < DMA_PER_WAVE; ++q) OZ_DMA1(q, 1, 1)
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(DMA_PER_WAVE) : "memory");
#pragma unroll
    for (int j = 0; j < NFR; ++j) OZ_FRAG(j, smem)
    if (OZ_ABL_NOFRAG) {     // diagnostic: some fragments once, so that the MFMAs have defined inputs
        _Pragma("unroll") for (int d = 0; d < K_DIG; ++d)
        {
            FR.a[d] = FR.b[d] = FI.a[d] = FI.b[d] = *reinterpret_cast<const v4i *>(smem + fragA + d * 16);
        }
    }
#if OZ_STAMP
    t_loop = __builtin_amdgcn_s_memtime();
#endif
    int st = 0;                                  // stage of K-step kt
    for (int kt = 0; kt < KT; ++kt) {
        const int st1 = st == STAGES - 1 ? 0 : st + 1;           // stage of K-step kt+1
        const int st2 = st1 == STAGES - 1 ? 0 : st1 + 1;         // stage of K-step kt+2 (= of kt-1: read out)
        const unsigned char *nbase = smem + st1 * STAGE_BYTES;
        OZ_FENCE();
#pragma unroll
        for (int g = 0; g < 2 * PAIRS; ++g) {
            if (g < PAIRS) {
                OZ_MFMA(FR, 0, g)
                OZ_PARK(2, g)             // the third sweep's leading pair of the previous K-step (zero at kt = 0)
            } else {
                OZ_MFMA(FI, 1, g - PAIRS)
                OZ_PARK(0, g - PAIRS)
            }
            if (g % 2 == 0 && g / 2 < DMA_PER_WAVE) {
                OZ_DMA1(g / 2, kt + 2, st2)
            } else {
                const int before = (g + 1) / 2 < DMA_PER_WAVE ? (g + 1) / 2 : DMA_PER_WAVE;   // DMA gaps before this one
                const int u = g - before;
                const int t0 = (u * NSW + NGAP - 1) / NGAP, t1 = ((u + 1) * NSW + NGAP - 1) / NGAP;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = t0 + k;
                    if (t < t1) {
                        const int d = (t % (NSW / 2)) / 4, c = t % 4;
                        if (t < NSW / 2) FS.a[d][c] = OZ_ABL_NOSWAR ? FR.a[d][c] : off_add(FR.a[d][c], FI.a[d][c]);   // ar + ai
                        else FS.b[d][c] = OZ_ABL_NOSWAR ? FR.b[d][c] : off_sub(FI.b[d][c], FR.b[d][c]);               // mi - mr
                    }
                }
            }
            OZ_FENCE();
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DMA_PER_WAVE) : "memory");
        if (!OZ_ABL_NOBARRIER) asm volatile("s_barrier" ::: "memory");
        OZ_FENCE();
#pragma unroll
        for (int g = 0; g < PAIRS; ++g) {
            OZ_MFMA(FS, 2, g)
            OZ_PARK(1, g)
            if (2 * g < NFR) {
                OZ_FRAG(2 * g, nbase)
                OZ_FRAG(2 * g + 1, nbase)
            }
            OZ_FENCE();
        }
        st = st1;
    }
    if (PARK0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) S0[2][q] += T0[q];
    }
    }   // !lower
#if OZ_STAMP
    const unsigned long long t_loop_end = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#undef OZ_FENCE
#undef OZ_PARK
#undef OZ_FRAG
#undef OZ_MFMA
#undef OZ_PAIR_B
#undef OZ_PAIR_A
#undef OZ_DMA1

    // U_tau = s_a s_m sum_s G_s 128^-(s+2);  Re = -U1 - U2, Im = U3 + U1 - U2
    const int gj = j0 + wn * 32 + r;
    const double sbj = sm[gj];
    // offset correction of U1 / U2, group s: 64 (digit sums of the A row + of the M row) + 4096 N (s+1).
    // The A rows' part and scales go through LDS once per tile (the K loop is done with it).
    const int *__restrict__ dsa = reinterpret_cast<const int *>(sa + N);
    const int *__restrict__ dsm = reinterpret_cast<const int *>(sm + N);
    constexpr int CA_PITCH = ((PLANES + 3) / 4) * 4;                     // ints per row: whole 16-byte pieces
    int *ca_lds = reinterpret_cast<int *>(smem + 72 * 1024);             // [64][CA_PITCH]
    double *sa_lds = reinterpret_cast<double *>(smem + 72 * 1024 + 64 * CA_PITCH * 4);   // [64]
    if (!lower) {
        for (int i = tid; i < 64 * PLANES; i += 256) ca_lds[(i / PLANES) * CA_PITCH + i % PLANES] = 64 * dsa[(size_t)i0 * PLANES + i];
        if (tid < 64) sa_lds[tid] = sa[i0 + tid];
    }
    // (the M column's part: one column per lane, in registers -- the fragment registers are free by now.  Round 4: also
    // for 6 digits, and the A rows' part is read from LDS as three 16-byte pieces per row instead of one word per
    // (element, product, digit): 51 LDS reads per lane where the epilogue had 576, behind the last MFMA.)
    // (the six-digit FUSED product keeps the column's part in LDS: twelve more live registers through its long epilogue
    // cost it 8 bytes of scratch)
    constexpr bool CM_REGS = !(K_DIG >= 6 && FUSEDEPI);
    int *cm_lds = reinterpret_cast<int *>(smem + 72 * 1024 + 64 * CA_PITCH * 4 + 64 * 8);   // [64][CA_PITCH]
    int cm[2][K_DIG];
    if constexpr (CM_REGS) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s_ = 0; s_ < K_DIG; ++s_) cm[c][s_] = 64 * dsm[(size_t)gj * (2 * KDM) + c * KDM + s_] + 4096 * N * (s_ + 1);
    } else if (!lower) {
        for (int i = tid; i < 64 * PLANES; i += 256)
            cm_lds[(i / PLANES) * CA_PITCH + i % PLANES] = 64 * dsm[(size_t)j0 * PLANES + i] + 4096 * N * (i % K_DIG + 1);
    }
    __syncthreads();
#define OZ_RESULT(reg_, gi_, tre_, tim_)                                               \
    {                                                                                  \
        const double sc_ = sa_lds[(gi_) - i0] * sbj;                                   \
        int car_[((PLANES + 3) / 4) * 4];                                              \
        _Pragma("unroll") for (int q_ = 0; q_ < (PLANES + 3) / 4; ++q_)                \
        {                                                                              \
            const v4i t4_ = *reinterpret_cast<const v4i *>(ca_lds + ((gi_) - i0) * CA_PITCH + 4 * q_); \
            car_[4 * q_] = t4_[0];                                                     \
            car_[4 * q_ + 1] = t4_[1];                                                 \
            car_[4 * q_ + 2] = t4_[2];                                                 \
            car_[4 * q_ + 3] = t4_[3];                                                 \
        }                                                                              \
        double T_[3];                                                                  \
        _Pragma("unroll") for (int tau = 0; tau < 3; ++tau)                            \
        {                                                                              \
            double t_ = 0.0;                                                           \
            _Pragma("unroll") for (int s_ = K_DIG - 1; s_ >= 0; --s_) /* small terms first */ \
            {                                                                          \
                int g_ = (PARK0 && s_ == 0) ? S0[tau][reg_] : acc[tau][s_][reg_];      \
                if (tau < 2) g_ -= car_[tau * K_DIG + s_] + (CM_REGS ? cm[tau][s_] : cm_lds[(gj - j0) * CA_PITCH + tau * K_DIG + s_]); \
                t_ += (double)g_ * (1.0 / (double)(1ull << (7 * (s_ + 2))));           \
            }                                                                          \
            T_[tau] = t_ * sc_;                                                        \
        }                                                                              \
        tre_ = -(T_[0] + T_[1]);                                                       \
        tim_ = (T_[2] + T_[0]) - T_[1];                                                \
    }
    if constexpr (!FUSEDEPI) {
        // (buffer addressing: one per-lane byte offset, the row of a register as a scalar offset)
        const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc(C, 0, (int)((size_t)N * N * sizeof(cplx)), 0x00020000);
        const unsigned vbase = (unsigned)(((size_t)(i0 + wm * 32 + 4 * h) * N + gj) * sizeof(cplx));
        const unsigned row_stride = (unsigned)N * (unsigned)sizeof(cplx);
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int gi = i0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            double tre, tim;
            OZ_RESULT(reg, gi, tre, tim)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, make_double2(tre, tim)), r_c, vbase,
                                                   (unsigned)((reg & 3) + 8 * (reg >> 2)) * row_stride, 0);
        }
        // Im C_ii in fp64 from the slicing launch (k_oz_slice, PAIR) over the digit sum's value: the lanes of a diagonal
        // tile's diagonal sub-tiles that hold (i, i) -- row (reg & 3) + 8 (reg >> 2) + 4 h = column r -- store it behind
        // their own tile stores (outside the loop above: inside, the pointer's two registers were 8 bytes of scratch)
        if (mir.diag && tm == tn && wm == wn && ((r >> 2) & 1) == h) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            reinterpret_cast<double *>(C + (size_t)gj * N + gj)[1] = mir.diag[gj];
        }
#if OZ_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        if (lane == 0) {      // diagnostic build only: cycle stamps of this wave over the tile's first entries
            double *o = reinterpret_cast<double *>(C) + ((size_t)blockIdx.x * 4 + wave) * 4;
            o[0] = (double)(t_loop - t_begin);
            o[1] = (double)(t_loop_end - t_loop);
            o[2] = (double)(t_end - t_loop_end);
            o[3] = (double)(t_begin & 0xffffffffull);
        }
#endif
    } else {
        const int parity = guard.state ? guard.state->dw_parity : 0;
        const cplx *__restrict__ dW_old = ep.dW[parity];
        cplx *__restrict__ dW_new = ep.dW[parity ^ 1];
        const int wpar = guard.state ? guard.state->w_parity : 0;
        const cplx *__restrict__ Wcur = wpar ? ep.Wpair[1] : ep.Wpair[0];
        cplx *__restrict__ Wnext = wpar ? ep.Wpair[0] : ep.Wpair[1];
        // The K loop is done with LDS.  The transposed operand PW[gj][gi] of the commutator comes
        // through it: the 64 x 64 block PW[j0.., i0..] arrives by LDS-DMA, one whole 1-KiB row per
        // instruction (coalesced; pitch 1040 bytes: the column reads below are conflict-free), where
        // per-lane loads of it would touch 64 cache lines per instruction.
        constexpr int TP = 1040;
        unsigned char *tblk = smem + 64 * TP;                            // mirrored tiles: the partner's result tile
        double *rs = reinterpret_cast<double *>(smem + 2 * 64 * TP);     // [2][64] row sums, flag, step-end scratch
        static_assert(2 * 64 * TP + 2048 <= (int)cfg::SMEM, "epilogue LDS");
        const __amdgpu_buffer_rsrc_t r_t = __builtin_amdgcn_make_buffer_rsrc(
            mir.tbuf, 0, mirror ? (int)((size_t)tiles * (tiles + 1) / 2 * OZ_T * OZ_T * sizeof(cplx)) : 0, 0x00020000);
        // The multiplying workgroup turns its accumulators into T first (they are dead after that) and,
        // when a mirrored tile is waiting for it, publishes T before anything else of its epilogue.
        double tre_[16], tim_[16];
        if (!lower) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int gi = i0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                OZ_RESULT(reg, gi, tre_[reg], tim_[reg])
            }
            if (mirror && tm < tn) {      // row-major 64 x 64, write-through stores; drain; flag
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, make_double2(tre_[reg], tim_[reg])), r_t,
                                                           (unsigned)(((wm * 32 + 4 * h) * OZ_T + wn * 32 + r) * sizeof(cplx)),
                                                           (unsigned)((size_t)upair * OZ_T * OZ_T * sizeof(cplx)) +
                                                               (unsigned)((reg & 3) + 8 * (reg >> 2)) * (unsigned)(OZ_T * sizeof(cplx)),
                                                           16);
                asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                if (tid == 0 && !(mir.debug_drop & 1)) __hip_atomic_store(mir.flags + upair, mir.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // A tile ON the diagonal multiplies both of its triangles, and a digit split's T[i][j] and -conj(T[j][i]) differ by
            // the truncation -- with the five leading of six digits by a BIASED one.  dW, and with it the next Whalf, would be
            // skew-Hermitian only to 1e-11 inside these tiles, and the fp64 diagonal of the next first product (the slicing
            // launch's pair sums, which cancel term by term over an exactly skew-Hermitian Whalf) would no longer add up to
            // zero: tr W leaked 1e-12 per step on smooth data.  The tile parks its T in LDS (the block a mirrored tile receives
            // its partner's T in) and takes the entries below its diagonal from above it, as the mirrored tiles do.
            if (tm == tn) {
                // (the block overlays the offset-correction tables the conversion above read: every wave is done with them first)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int li = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                    *reinterpret_cast<cplx *>(tblk + li * TP + (wn * 32 + r) * 16) = make_double2(tre_[reg], tim_[reg]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the barrier in front of the epilogue loop orders the reads)
            }
        }
        {
            const __amdgpu_buffer_rsrc_t rpw = __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(ep.PW), 0, (int)((size_t)N * N * sizeof(cplx)), 0x00020000);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = wave * 16 + q;
                oz_dma16(rpw, (lds_void *)(smem + row * TP), (unsigned)(((size_t)(j0 + row) * N + i0 + lane) * sizeof(cplx)), 0u);
            }
        }
        // EPI_R rounds of EPI_U rows, the operand loads of EPI_D rounds in flight (the fragment
        // registers are free by now)
        constexpr int EPI_U = 2, EPI_R = 16 / EPI_U, EPI_D = K_DIG <= 5 ? 3 : 2;
        // (buffer addressing: one per-lane byte offset, the row of a register as a scalar offset)
        const size_t mat_bytes = (size_t)N * N * sizeof(cplx);
#define OZ_RSRC(p_) __builtin_amdgcn_make_buffer_rsrc(const_cast<cplx *>(p_), 0, (int)mat_bytes, 0x00020000)
        const __amdgpu_buffer_rsrc_t r_pw = OZ_RSRC(ep.PW), r_w = OZ_RSRC(Wcur), r_dold = OZ_RSRC(dW_old);
        const __amdgpu_buffer_rsrc_t r_dnew = OZ_RSRC(dW_new), r_wh = OZ_RSRC(ep.Whalf), r_wn = OZ_RSRC(Wnext);
        const __amdgpu_buffer_rsrc_t r_whs = OZ_RSRC(ep.Whalf_step);
#undef OZ_RSRC
        const unsigned vbase = (unsigned)(((size_t)(i0 + wm * 32 + 4 * h) * N + gj) * sizeof(cplx));
        const unsigned row_stride = (unsigned)N * (unsigned)sizeof(cplx);
        cplx pw[EPI_D][EPI_U], wv[EPI_D][EPI_U], dold[EPI_D][EPI_U];
#define OZ_EPI_LOAD(q4_)                                                               \
    _Pragma("unroll") for (int u = 0; u < EPI_U; ++u)                                  \
    {                                                                                  \
        const int reg = EPI_U * (q4_) + u;                                             \
        const unsigned so = (unsigned)((reg & 3) + 8 * (reg >> 2)) * row_stride;       \
        pw[(q4_) % EPI_D][u] = __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(r_pw, vbase, so, 0));     \
        wv[(q4_) % EPI_D][u] = __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(r_w, vbase, so, 0));      \
        dold[(q4_) % EPI_D][u] = __builtin_bit_cast(cplx, __builtin_amdgcn_raw_buffer_load_b128(r_dold, vbase, so, 0)); \
    }
#pragma unroll
        for (int q4 = 0; q4 < EPI_D; ++q4) OZ_EPI_LOAD(q4)
        if (lower) {
            // wait for the partner's tile (one lane polls; launch epoch), then fetch it past the L2
            if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(mir.flags + upair, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != mir.epoch) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > OZ_SPIN_LIMIT) {
                        *mir.fault = 1;
                        break;
                    }
                }
            }
            asm volatile("s_barrier" ::: "memory");
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = wave * 16 + q;
                oz_dma16_sc1(r_t, (lds_void *)(tblk + row * TP), (unsigned)((row * OZ_T + lane) * sizeof(cplx)),
                             (unsigned)((size_t)upair * OZ_T * OZ_T * sizeof(cplx)));
            }
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // the DMA'd blocks are in LDS for everybody
#pragma unroll
        for (int q4 = 0; q4 < EPI_R; ++q4) {
            cplx pwt[EPI_U];
#pragma unroll
            for (int u = 0; u < EPI_U; ++u) {
                const int reg = EPI_U * q4 + u;
                const int li = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                pwt[u] = *reinterpret_cast<const cplx *>(smem + (wn * 32 + r) * TP + li * 16);
            }
#pragma unroll
            for (int u = 0; u < EPI_U; ++u) {
                const int reg = EPI_U * q4 + u;
                const int li = wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const unsigned so = (unsigned)((reg & 3) + 8 * (reg >> 2)) * row_stride;
                double tre, tim;
                const bool below = (tm == tn) && li > wn * 32 + r;      // diagonal tile, entry below its diagonal
                if (lower || below) {         // T[gi][gj] = -conj(T[gj][gi]): the partner's (this tile's own) entry (lj, li)
                    const cplx tp = *reinterpret_cast<const cplx *>(tblk + (wn * 32 + r) * TP + li * 16);
                    tre = -tp.x;
                    tim = tp.y;
                } else {
                    tre = ((tm == tn) && li == wn * 32 + r) ? 0.0 : tre_[reg];      // (a skew-Hermitian diagonal is imaginary)
                    tim = tim_[reg];
                }
                const cplx pwv = pw[q4 % EPI_D][u], wvv = wv[q4 % EPI_D][u], dov = dold[q4 % EPI_D][u];
                // comm = PW - PW^H (conj_subtract_, isospectral.py:66-81);  dW = PW@Phalf + comm (:499,509)
                const double cr = pwv.x - pwt[u].x, ci = pwv.y + pwt[u].y;
                const double dr = tre + cr, di = tim + ci;
#define OZ_ST(rs_, x_, y_) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, make_double2(x_, y_)), rs_, vbase, so, 0)
                OZ_ST(r_dnew, dr, di);
                OZ_ST(r_wh, wvv.x + dr, wvv.y + di);                                   // isospectral.py:481-482
                const double wr = wvv.x + 2.0 * cr, wi = wvv.y + 2.0 * ci;             // isospectral.py:547,592
                OZ_ST(r_wn, wr, wi);
                OZ_ST(r_whs, wr + dr, wi + di);
#undef OZ_ST
                const double er = dov.x - dr, ei = dov.y - di;                         // isospectral.py:526,534
                double a = qf_modulus(er, ei);
                a = qf_row16_sum(a);             // the 32 lanes of this row: xor 1, 2, 4, 8 on DPP (registers only) ...
                a += __shfl_xor(a, 16, 64);      // ... and the one step that crosses DPP rows
                if (r == 0) rs[wn * 64 + li] = a;
            }
            if (q4 + EPI_D < EPI_R) OZ_EPI_LOAD(q4 + EPI_D)
        }
#undef OZ_EPI_LOAD
        __syncthreads();
        if (tid < 64)
            __hip_atomic_store(ep.rowpart + (size_t)tn * N + i0 + tid, rs[tid] + rs[64 + tid], __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        // the last tile to get here closes the iteration (qf_step_end.h)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned *last_flag = reinterpret_cast<unsigned *>(rs + 128);
        if (tid == 0) {
            unsigned old = 0u;
            if (!((mir.debug_drop & 2) && blockIdx.x == 0))      // (fault injection: one tile takes no ticket)
                old = __hip_atomic_fetch_add(ep.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *last_flag = (old == (unsigned)(ep.n_tiles - 1)) ? 1u : 0u;
        }
        __syncthreads();
        if (*last_flag != 0u)
            qf_fused_step_end(N, tiles, ep.rowpart, ep.ticket, ep.state_rw, ep.rec, guard.iter, tid, rs + 130);
    }
#undef OZ_RESULT
}

}  // namespace

size_t qf_oz_operand_bytes(int N, int digits) { return (size_t)N * (N / 16) * (size_t)(32 * digits); }
size_t qf_oz_record_bytes(int N, int digits) { return (size_t)N * (sizeof(double) + 2 * digits * sizeof(int)); }

int qf_launch_oz_slice(qf_ctx *ctx, const qf_oz_jobs &jobs, qf_guard guard, int digits)
{
    if (digits != 5 && digits != 6) digits = ctx->oz_digits;
    const int N = ctx->N;
    if (N % 16 != 0 || N > 4096) {
        qf_set_error("qf_launch_oz_slice: N=%d must be a multiple of 16, at most 4096", N);
        return QF_ERR_INVALID;
    }
    const int threads = ((N / 4 + 63) / 64) * 64;
    const int planes = 2 * digits;
    const size_t smem = 256 + 16 * planes * 4 + (size_t)(N / 16) * planes * 16;
    if (jobs.diag && jobs.n != 2) {
        qf_set_error("qf_launch_oz_slice: the diagonal of a product wants its two operands (%d jobs)", jobs.n);
        return QF_ERR_INVALID;
    }
    const int blocks = jobs.diag ? N : jobs.n * N;       // pair mode: one workgroup per row of both operands
    qf_plan_note(ctx, 0x5000000ull | (unsigned long long)(digits << 8 | jobs.n << 1 | (jobs.diag ? 1 : 0)),
                 "{\"kernel\": \"k_oz_slice<%d>\", \"operands\": %d, \"digits\": %d, \"workgroups\": %d, \"threads\": %d, "
                 "\"fp64_diagonal_of_the_product\": %s}", digits, jobs.n, digits, blocks, threads, jobs.diag ? "true" : "false");
    if (digits == 6)
        hipLaunchKernelGGL(k_oz_slice<6>, dim3(blocks), dim3(threads), smem, ctx->stream, N, jobs, guard);
    else
        hipLaunchKernelGGL(k_oz_slice<5>, dim3(blocks), dim3(threads), smem, ctx->stream, N, jobs, guard);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_oz_gemm(qf_ctx *ctx, const signed char *pa, const double *sa, const signed char *pm, const double *sm,
                      cplx *C, const qf_epilogue *ep, qf_guard guard, int digits, int digits_m, const double *diag)
{
    // digits: of A and of the multiplication; digits_m: of M's storage layout (6 / 5: the leading five of six)
    if (digits != 5 && digits != 6) digits = ctx->oz_digits;
    if (digits_m != 5 && digits_m != 6) digits_m = digits;
    if (digits_m < digits || (digits_m != digits && !ep)) {
        qf_set_error("qf_launch_oz_gemm: digits %d / layout %d not available", digits, digits_m);
        return QF_ERR_INVALID;
    }
    const int N = ctx->N;
    if (N % 64 != 0) {
        qf_set_error("qf_launch_oz_gemm: N=%d is not a multiple of 64", N);
        return QF_ERR_INVALID;
    }
    {
        static qf_smem_attr a5p, a5f, a6p, a6f, a56f;
        QF_TRY(qf_smem_attr_set(a56f, (const void *)k_oz_gemm<5, true, 6>, ctx->device, ozc<5>::SMEM));
        QF_TRY(qf_smem_attr_set(a5p, (const void *)k_oz_gemm<5, false>, ctx->device, ozc<5>::SMEM));
        QF_TRY(qf_smem_attr_set(a5f, (const void *)k_oz_gemm<5, true>, ctx->device, ozc<5>::SMEM));
        QF_TRY(qf_smem_attr_set(a6p, (const void *)k_oz_gemm<6, false>, ctx->device, ozc<6>::SMEM));
        QF_TRY(qf_smem_attr_set(a6f, (const void *)k_oz_gemm<6, true>, ctx->device, ozc<6>::SMEM));
    }
    const int tiles = N / 64;
    const dim3 grid(tiles * tiles), block(256);
    qf_epilogue e;
    qf_oz_mirror mir;
    mir.diag = ep ? nullptr : diag;        // plain product: the diagonal's imaginary parts from the slicing launch
    if (ep) {
        e = *ep;
        e.ticket = ctx->ticket + 400;      // the tile-ticket word of the fused step end (cf. zgemm.hip launch4)
        e.n_tiles = tiles * tiles;
        e.state_rw = ctx->state;
        e.rec = ctx->host_rec;
        if (ctx->oz_tbuf && ctx->oz_tflags && tiles > 1) {
            mir.tbuf = ctx->oz_tbuf;
            mir.flags = ctx->oz_tflags;
            mir.epoch = ++ctx->oz_epoch;
            if (mir.epoch == 0u) mir.epoch = ++ctx->oz_epoch;
            mir.fault = &ctx->host_rec->fault;
        }
        // fault injection, one launch (tests/test_hip_faults.py): 1 = no upper tile publishes its result tile's flag (the
        // mirrored tiles' bounded waits run out), 2 = tile 0 takes no step-end ticket (the iteration never closes)
        if ((ctx->debug_drop == 1 && mir.epoch != 0u) || (ctx->debug_drop == 2 && ep->fused)) {
            mir.debug_drop = ctx->debug_drop;
            ctx->debug_drop = 0;
        }
    }
    {
        const bool mirrored = ep && mir.epoch != 0u;
        const int mult = mirrored ? tiles * (tiles + 1) / 2 : tiles * tiles;
        qf_plan_note(ctx, 0x6000000ull | (unsigned long long)(digits << 8 | digits_m << 4 | (ep ? 2 : 0) | (mirrored ? 1 : 0)),
                     "{\"kernel\": \"k_oz_gemm<%d,%s,%d>\", \"arithmetic\": \"int8 digit split, v_mfma_i32_32x32x32_i8, %d digit pairs x 3M\", \"digit_pairs\": %d, "
                     "\"tile\": [64, 64], \"tiles\": %d, \"tile_share\": %.6f, \"workgroups\": %d, \"threads\": 256, \"step_end\": \"%s\"}",
                     digits, ep ? "fused" : "plain", digits_m, digits * (digits + 1) / 2, digits * (digits + 1) / 2, mult, (double)mult / ((double)tiles * tiles),
                     tiles * tiles, ep ? "fused (last tile decides); mirrored tiles wait for their partner's result tile" : "none");
    }
    if (digits == 5 && digits_m == 6) {
        hipLaunchKernelGGL((k_oz_gemm<5, true, 6>), grid, block, ozc<5>::SMEM, ctx->stream, N, pa, sa, pm, sm, C, e, guard, mir);
    } else if (digits == 6) {
        if (ep) hipLaunchKernelGGL((k_oz_gemm<6, true>), grid, block, ozc<6>::SMEM, ctx->stream, N, pa, sa, pm, sm, C, e, guard, mir);
        else hipLaunchKernelGGL((k_oz_gemm<6, false>), grid, block, ozc<6>::SMEM, ctx->stream, N, pa, sa, pm, sm, C, e, guard, mir);
    } else {
        if (ep) hipLaunchKernelGGL((k_oz_gemm<5, true>), grid, block, ozc<5>::SMEM, ctx->stream, N, pa, sa, pm, sm, C, e, guard, mir);
        else hipLaunchKernelGGL((k_oz_gemm<5, false>), grid, block, ozc<5>::SMEM, ctx->stream, N, pa, sa, pm, sm, C, e, guard, mir);
    }
    QF_HIP(hipGetLastError());
    return QF_OK;
}
