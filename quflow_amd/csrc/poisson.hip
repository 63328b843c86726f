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
// * Depth reduction.  Each walk is cut into chunks of L steps (8 for N <= 512, 16 below 768; 9 / 17 in the folded
//   layout of the skew-Hermitian solve from N = 768 on; 32 beyond N = 2175); a thread owns one chunk in registers.  Pass 1 sweeps every chunk with a zero
//   carry-in (all chunks of all diagonals in parallel); pass 2 forms the chunk carries by a
//   wavefront scan of the affine maps y -> a y + b over the chunks of a walk (DPP row_shr /
//   row_bcast lane moves, registers only: log2(C) + 1 steps for C <= 64 chunks, two chunks per lane up to 128; the
//   sequential pass over the chunks is left for longer walks only); pass 3 adds
//   carry * prod(-w) to every entry.  Sequential depth drops from 2N to about
//   2(2L + log2(N/L)) dependent FMAs; HBM/L2 traffic is the algorithmic minimum (W once, the
//   factor table once, P once).
// * Folded walk slots (skew-Hermitian solve, 768 <= N <= 2175): the upper triangle's walks have lengths N .. 1; slot f
//   carries walk f FOLLOWED BY walk N-1-f as one sequence of N + 1 entries -- the zero multiplier at the head of a walk
//   restarts both recurrences at the junction -- so every slot is full: half the workgroups, none of them half idle.
// * The kernel is instantiated for double (complex128 data) and for float (complex64 data: the
//   reference solves those with float32 tables and float32 arithmetic, cpu.py:725).
// * m = 0: tr(W)/N is removed from the right-hand side and tr(P)/N from the solution
//   (cpu.py:311-317,342-352) with deterministic block reductions.
#include "qf_internal.h"
#include "qf_step_end.h"

#pragma clang fp contract(off)  // table arithmetic mirrors the reference's op order; FMAs are explicit

// Diagnostic builds only (tools/solve_probe.hip): switches that drop parts of k_solve so that
// their cost can be read off timing differences.  Never defined in the shipped library.
#ifdef QF_PROBE
__device__ int qf_probe_flags = 0;
__device__ unsigned long long *qf_probe_stamps = nullptr;  // [blocks*waves][16]
#define QF_PROBE_SKIP(bit_) (qf_probe_flags & (1 << (bit_)))
#define QF_PROBE_STAMP(slot_)                                                                     \
    if (qf_probe_stamps && (threadIdx.x & 63) == 0)                                               \
        qf_probe_stamps[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot_)] = __builtin_amdgcn_s_memtime();
#else
#define QF_PROBE_SKIP(bit_) 0
#define QF_PROBE_STAMP(slot_)
#endif

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

// R = float: the reference's float32 table for complex64 input (cpu.py:55-95 with dtype=float32, cpu.py:725):
// the integer-valued diagonal cast to float32 (exact below 2^24), the double-precision square root rounded
// to float32, the boundary condition subtracted in float32.
template <typename R>
__global__ void k_lap_table(int N, int bc, R *__restrict__ lap)
{
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)N * N) return;
    int i = (int)(e / N), j = (int)(e % N);
    R b = (R)lap_b(N, i, j);
    if (bc && e == 0) b -= R(0.5);  // cpu.py:90
    lap[2 * e] = b;
    lap[2 * e + 1] = (R)lap_a(N, i, j);
}

// One thread per flat walk t = 0..N; sequential (runs once per table).
// Same operation order as cpu.py:309,324-325: w = a/b'_{k-1}; b'_k = b_k - w a_k.
template <typename R, typename C2>
__global__ void k_build_factors(int N, const R *__restrict__ lap, C2 *__restrict__ tab)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > N) return;
    size_t NN = (size_t)N * N;
    R bp_prev = R(1);
    for (size_t e = t; e < NN; e += (size_t)N + 1) {
        int i = (int)(e / N), j = (int)(e % N);
        R b = lap[2 * e], a = lap[2 * e + 1];
        R w, bp;
        if (i == 0 || j == 0) {  // head of a diagonal: the reference starts its sweep at k = 1
            w = R(0);
            bp = b;
        } else {
            w = a / bp_prev;
            bp = b - w * a;
        }
        C2 f;
        f.x = w;
        f.y = R(1) / bp;
        tab[e] = f;
        bp_prev = bp;
    }
}

// _dot_cpu_generic, cpu.py:98-108 (coefficients recomputed on the fly: no table traffic)
template <typename R, typename C2>
__global__ void k_laplace(int N, const C2 *__restrict__ P, C2 *__restrict__ W)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= N) return;
    size_t e = (size_t)i * N + j;
    R b = (R)lap_b(N, i, j);
    C2 p = P[e];
    R wr = b * p.x, wi = b * p.y;
    if (i < N - 1 && j < N - 1) {
        R a = (R)lap_a(N, i + 1, j + 1);
        C2 q = P[e + N + 1];
        wr += a * q.x;
        wi += a * q.y;
    }
    if (i > 0 && j > 0) {
        R a = (R)lap_a(N, i, j);
        C2 q = P[e - N - 1];
        wr += a * q.x;
        wi += a * q.y;
    }
    C2 o;
    o.x = wr;
    o.y = wi;
    W[e] = o;
}

// ---- real-type traits: the solve is instantiated for double (complex128 data) and float (complex64 data: the
// reference solves complex64 input with float32 tables and float32 arithmetic, cpu.py:725)
template <typename R> struct rt;
template <> struct rt<double> { typedef double2 C; };
template <> struct rt<float> { typedef float2 C; };
template <typename R> __device__ __forceinline__ typename rt<R>::C mkc(R x, R y);
template <> __device__ __forceinline__ double2 mkc<double>(double x, double y) { return make_double2(x, y); }
template <> __device__ __forceinline__ float2 mkc<float>(float x, float y) { return make_float2(x, y); }
__device__ __forceinline__ double fma_r(double a, double b, double c) { return __fma_rn(a, b, c); }
__device__ __forceinline__ float fma_r(float a, float b, float c) { return __fmaf_rn(a, b, c); }

// DPP lane moves of a double (two dwords) or a float: no LDS crossbar, ~2 VALU issues instead of two
// ds_bpermute round trips (a __shfl_up of a double).  Lanes without a valid source keep their own value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v)
{
    int x = __float_as_int(v);
    x = __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xf, false);
    return __int_as_float(x);
}

// the same move with 0.0 where a lane has no source (reductions)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_or_zero(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_or_zero(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ double read_lane63(double v)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ float read_lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }

// sum over the 64 lanes of a wavefront in a fixed order, returned in every lane: running sums inside the
// rows of 16 (row_shr 1, 2, 4, 8), row totals handed on (row_bcast:15, row_bcast:31), lane 63 read back.
// Registers only -- a __shfl_xor butterfly on doubles is 12 ds_bpermute round trips.
template <typename R>
__device__ __forceinline__ R wave_total(R v)
{
    v += dpp_mov_or_zero<0x111, 0xf>(v);
    v += dpp_mov_or_zero<0x112, 0xf>(v);
    v += dpp_mov_or_zero<0x114, 0xf>(v);
    v += dpp_mov_or_zero<0x118, 0xf>(v);
    v += dpp_mov_or_zero<0x142, 0xa>(v);
    v += dpp_mov_or_zero<0x143, 0xc>(v);
    return read_lane63(v);
}

// deterministic block-wide sum of a complex value in two halves: wave_total inside each wavefront, the wave totals
// left in red[wave] (post); after ONE barrier of the caller's -- an existing one where there is one -- every thread
// adds the wave totals in wave order (read).  red[] must not be written again before every thread has read it.
template <typename R>
__device__ __forceinline__ void block_sum_post(typename rt<R>::C v, typename rt<R>::C *red, int tid)
{
    v.x = wave_total(v.x);
    v.y = wave_total(v.y);
    if ((tid & 63) == 0) red[tid >> 6] = v;
}
template <typename R>
__device__ __forceinline__ typename rt<R>::C block_sum_read(const typename rt<R>::C *red, int nthreads)
{
    typename rt<R>::C r = mkc<R>(R(0), R(0));
    for (int w = 0; w < (nthreads >> 6); ++w) {
        r.x += red[w].x;
        r.y += red[w].y;
    }
    return r;
}

// Inclusive scan of affine maps y -> a*y + b over segments of `width` consecutive lanes (width a power
// of two <= 64, lane = index inside the wavefront): after the call a lane holds the composition of the
// maps of its segment's lanes up to and including itself.  Kogge-Stone inside the rows of 16 lanes
// (DPP row_shr 1, 2, 4, 8), then the row totals are handed on with row_bcast:15 (rows 1 and 3 take
// lane 15 of the row before) and row_bcast:31 (rows 2 and 3 take lane 31): six steps, registers only.
template <typename R>
__device__ __forceinline__ void scan_affine(R &a, typename rt<R>::C &b, int lane, int width)
{
    const int l = lane & (width - 1);
#define QF_SCAN_STEP(D_)                                                               \
    if (width > (D_)) {                                                                \
        const R ap = dpp_mov<0x110 + (D_), 0xf>(a);                                    \
        const R bx = dpp_mov<0x110 + (D_), 0xf>(b.x);                                  \
        const R by = dpp_mov<0x110 + (D_), 0xf>(b.y);                                  \
        if (l >= (D_) && (lane & 15) >= (D_)) {                                        \
            b.x = fma_r(a, bx, b.x);                                                   \
            b.y = fma_r(a, by, b.y);                                                   \
            a *= ap;                                                                   \
        }                                                                              \
    }
    QF_SCAN_STEP(1)
    QF_SCAN_STEP(2)
    QF_SCAN_STEP(4)
    QF_SCAN_STEP(8)
#undef QF_SCAN_STEP
    if (width > 16) {      // row_bcast:15 into rows 1 and 3
        const R ap = dpp_mov<0x142, 0xa>(a);
        const R bx = dpp_mov<0x142, 0xa>(b.x);
        const R by = dpp_mov<0x142, 0xa>(b.y);
        if (lane & 16) {
            b.x = fma_r(a, bx, b.x);
            b.y = fma_r(a, by, b.y);
            a *= ap;
        }
    }
    if (width > 32) {      // row_bcast:31 into rows 2 and 3
        const R ap = dpp_mov<0x143, 0xc>(a);
        const R bx = dpp_mov<0x143, 0xc>(b.x);
        const R by = dpp_mov<0x143, 0xc>(b.y);
        if (lane & 32) {
            b.x = fma_r(a, bx, b.x);
            b.y = fma_r(a, by, b.y);
            a *= ap;
        }
    }
}

// value of the lane before (wave_shr:1), for the exclusive carries; lane 0 keeps its own
template <typename R> __device__ __forceinline__ R lane_before(R v) { return dpp_mov<0x138, 0xf>(v); }

// chunk carries of one recurrence for up to 64 chunks per walk: lane l of a segment of Cp lanes takes chunk l
// (REVERSE: chunk C-1-l), the segment scans the chunks' affine maps, and the carry INTO a chunk is the value at the end
// of the chunk before it.  CPW = 64: the segment is the whole wavefront (33..64 chunks: N = 512's 64 chunks of 8
// entries), known at compile time -- no runtime width in the six scan steps, no division by it; CPW = 0: runtime Cp.
template <typename R, int CPW, bool REVERSE>
__device__ __forceinline__ void scan_chunks(int C, int G, int CE, int CR, int GP, const R *endc, const typename rt<R>::C *endv,
                                            typename rt<R>::C *carry, int lane, int wave, int nwaves)
{
    typedef typename rt<R>::C cplx;
    int Cp = CPW;
    if (!CPW) {
        Cp = 1;
        while (Cp < C) Cp <<= 1;
    }
    const int dpw = 64 / Cp;                 // walks per wavefront
    const int l = lane & (Cp - 1);
    const int q = REVERSE ? C - 1 - l : l;
    for (int gd = wave * dpw + (CPW == 64 ? 0 : lane / Cp); gd < G; gd += nwaves * dpw) {
        R a = R(0);
        cplx b = mkc<R>(R(0), R(0));
        if (l < C) {
            a = endc[gd * CR + q];
            b = endv[gd * CE + q];
        }
        scan_affine(a, b, lane, Cp);
        const R cx = lane_before(b.x), cy = lane_before(b.y);
        if (l < C) carry[q * GP + gd] = (l == 0) ? mkc<R>(R(0), R(0)) : mkc<R>(cx, cy);
    }
}

// Chunked two-level Thomas solve.  Block = G walks x C chunks (G*C threads, lane-fastest in g).
//   SKEWH = 1: walks t = 0..N-1 restricted to the upper triangle (length N-t), result
//              mirrored as P[j,i] = -conj(P[i,j])           (cpu.py:281-362)
//   SKEWH = 0: walks t = 0..N over the whole matrix          (cpu.py:200-278)
// Chunk carries: with C <= 64 one wavefront scans all chunks of a walk with shuffles
// (log2 C steps); longer walks fall back to a sequential pass over the chunks.
// Mirror: the block's results are staged in LDS and written as 16*G-byte row segments
// (the entries (k+t, k) of G consecutive walks t are G consecutive columns of row k+t).
// FOLD = 1 (skew-Hermitian solve of the larger sizes): walk slot f carries walk t = f (length N - f) FOLLOWED BY walk
//   N-1-f (length f + 1) as one sequence of N + 1 entries -- the table's multiplier at the head of a walk is zero, so
//   the recurrences restart at the junction by themselves (the flat walks of SKEWH = 0 rely on the same).  Every slot
//   is full: half the workgroups of the triangle's walk-per-slot layout, none of them half idle.
template <typename R, int L, int SKEWH, int FOLD = 0>
__global__ __launch_bounds__(L <= 17 ? 512 : 256) void k_solve(int N, int G, int C, const typename rt<R>::C *__restrict__ W,
                        typename rt<R>::C *__restrict__ P, const typename rt<R>::C *__restrict__ tab, R scale,
                        qf_guard guard, qf_decide dec, int tail_off)
{
    typedef typename rt<R>::C cplx;      // (shadows the file-level double2 typedef inside the kernel)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // Deferred step end: the second product before this launch left its row sums and
    // qf_dev_state::pending.  Every workgroup forms the decision itself and acts on it at once; thread 0 takes a
    // ticket whose answer is looked at when the workgroup is done: the last arrival (everyone else has read the old
    // state by then) writes the new state and publishes the progress.
    // (Tried: issuing this thread's loads from BOTH Whalf candidates and its table loads first and forming the
    // decision while they travel -- 8,745 against 9,137 timesteps/s at N = 512: twice the sweep loads and 70 more
    // registers cost more than the hidden round trip saved.)
    qf_new_state ns;
    bool decided = false;
    unsigned my_ticket = 0u;
    if (dec.state_rw) {
        // (in this protocol practically every solve follows a second product: the row sums are requested at once,
        // together with the control state, not behind the look at `pending` -- one memory round trip, not two)
        ns = qf_decide_compute(N, dec.slots, dec.rowpart, dec.state_rw, reinterpret_cast<double *>(smem_raw));
        if (dec.state_rw->pending) {
            decided = true;
            if (threadIdx.x == 0) my_ticket = __hip_atomic_fetch_add(dec.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#define QF_SOLVE_EXIT                                                                           \
    {                                                                                           \
        if (decided && threadIdx.x == 0 && my_ticket == gridDim.x - 1) qf_decide_apply(dec.state_rw, dec.rec, dec.ticket, ns); \
    }
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;  // = G*C rounded up to a multiple of 64
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthreads >> 6;
    const int g = tid % G;
    const int jc = tid / G;           // chunk index (>= C for padding threads)
    // workgroup -> walk group, XCD-aware: consecutive workgroup ids go
    // round-robin over the 8 XCDs; neighbouring walk groups share their 128-byte lines (G = 4 walks
    // are 64 bytes of a row), so XCD x takes a contiguous range of walk groups
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x, slot = bid & 7, l = bid >> 3;
        int start = 0;
        for (int y = 0; y < slot; ++y) start += (nb - y + 7) >> 3;
        bid = start + l;
    }
    const int t0 = bid * G;
    const int t = t0 + g;
    const int T = FOLD ? (N + 1) / 2 : (SKEWH ? N : N + 1);
    const size_t NN = (size_t)N * N;
    const size_t stride = (size_t)N + 1;
    // one wavefront scans the chunk carries of a walk: one chunk per lane up to 64 chunks, two up to 128
    const bool use_scan = (C <= 128);
    const bool scan_pairs = (C > 64);

    // LDS carve-up (scan phase): endv[C*G] complex, carry[C*G] complex, red[nthreads] complex,
    // endc[C*G] real.  The mirror staging tile ptile[C*L][G] reuses the same memory afterwards.
    // The chunk-end records are written chunk-by-thread (lanes run over the G walks fastest) and read walk-major
    // by the scan (lanes run over the chunks), the carries the other way round: each is a transposition through
    // LDS.  Padded strides keep both sides conflict-free (round 2's unpadded layout had 4-way conflicts on the
    // strided side: SQ_LDS_BANK_CONFLICT = 50 % of the kernel's LDS cycles): walk stride C + 2 complex entries
    // (G walks x 2 chunks of a b128 lane group land in 8 distinct 4-bank slots), C + 4 reals for the products,
    // G + 1 complex entries per chunk row of the carries.
    const int CE = C + 2, CR = C + 4, GP = G + 1;
    cplx *endv = reinterpret_cast<cplx *>(smem_raw);
    cplx *carry = endv + (size_t)CE * G;
    cplx *red = carry + (size_t)C * GP;
    R *endc = reinterpret_cast<R *>(red + nthreads);
    cplx *ptile = reinterpret_cast<cplx *>(smem_raw);
    cplx *red2 = reinterpret_cast<cplx *>(smem_raw + tail_off);   // 8 entries behind the larger of the two carve-ups
    // chunk-end records: walk-major for the wavefront scan, chunk-major for the serial pass
    const int end_idx = use_scan ? g * CE + jc : jc * G + g;
    const int endc_idx = use_scan ? g * CR + jc : jc * G + g;

    QF_PROBE_STAMP(0)
    int len = 0;
    if (t < T && jc < C) len = SKEWH ? (N - t) : (int)((NN - 1 - (size_t)t) / stride) + 1;
    // FOLD: len1 entries of walk t, then len2 of walk t2 = N-1-t (none when that is walk t itself: odd N's middle)
    const int len1 = len;
    const int t2 = N - 1 - t;
    if (FOLD && len > 0 && t2 != t) len += t + 1;
    // entry k of the sequence lives at base(k) + k * stride
    const long long base1 = t, base2 = (long long)t2 - (long long)len1 * (long long)stride;
    const bool has_trace = (bid == 0);  // the block that owns walk t = 0 (m = 0)
    const bool on_diag = (t == 0 && jc < C);   // (FOLD: for the entries k < len1 of this slot)

    const int k0 = jc * L;
    const size_t e0 = (size_t)t + (size_t)k0 * stride;
#define QF_ENTRY(k_) (FOLD ? (size_t)((((k_) < len1) ? base1 : base2) + (long long)(k_) * (long long)stride) : e0 + (size_t)((k_) - k0) * stride)

    cplx v[L];
    R w[L + 1];
    R inv[L];

    // ---- all global loads of this thread are issued up front and unconditionally (invalid
    // steps read a harmless in-range entry and are masked afterwards): a predicated load sits
    // behind a branch, which would serialise 3*L memory round trips behind the FMA chain.
    // Round 4: the factor table is data independent and nobody writes it -- its L + 1 loads go out BEFORE the tag
    // look-up below (the control state was last written by another XCD: its scalar loads are a memory round trip,
    // during which the table entries now travel; a launch that is not due drops them).  The deferred decision of the
    // small sizes stays ahead of them: everything waits for it, and loads return in order (with the table loads in
    // front of it N = 512 lost 0.5 %).
    // (the 32-entry chunks of the largest sizes keep the old order: their 256 + 100 registers leave no room for it)
    constexpr bool EARLY_TAB = (L <= 17);
    const size_t e_safe = (t < T) ? (size_t)t : 0;
    if constexpr (EARLY_TAB) {
#pragma unroll
        for (int s = 0; s < L; ++s) {
            const bool valid = (k0 + s) < len;
            const size_t e = valid ? QF_ENTRY(k0 + s) : e_safe;
            const cplx tb = tab[e];
            w[s] = tb.x;
            inv[s] = tb.y;
        }
        const bool valid = (k0 + L) < len;
        w[L] = tab[valid ? QF_ENTRY(k0 + L) : e_safe].x;
        if (!valid) w[L] = R(0);   // also the multiplier that links to the next chunk (backward sweep)
    }
    {
        bool due = true;
        int wh_sel = 0;
        if (decided) {
            due = (ns.step_index == guard.step && ns.iters_this_step == guard.iter);
            wh_sel = ns.wh_sel;
        } else if (guard.state) {
            due = qf_guard_iter(guard);       // tagged stepper launch that is not due: no-op
            wh_sel = guard.state->wh_sel;
        }
        if (!due) {
            QF_SOLVE_EXIT
            return;
        }
        // fused step end: the first iteration of a step reads the Whalf the previous step's last
        // product prepared for it (uniform scalar decision)
        if (guard.alt && wh_sel) W = static_cast<const cplx *>(guard.alt);
    }
#pragma unroll
    for (int s = 0; s < L; ++s) {
        const bool valid = (k0 + s) < len;
        const size_t e = valid ? QF_ENTRY(k0 + s) : e_safe;
        v[s] = W[e];          // (nontemporal loads here were tried: 30.7 us instead of 21.4)
        if constexpr (!EARLY_TAB) {
            const cplx tb = tab[e];
            w[s] = tb.x;
            inv[s] = tb.y;
        }
    }
    if constexpr (!EARLY_TAB) {
        const bool valid = (k0 + L) < len;
        w[L] = tab[valid ? QF_ENTRY(k0 + L) : e_safe].x;
        if (!valid) w[L] = R(0);
    }
    // ---- m = 0: circulation tr(W)/N, cpu.py:311-317.  The diagonal IS walk 0: its entries are already in the
    // registers of this workgroup's g = 0 threads (rounds 1-4 loaded them a second time, in a loop of dependent
    // round trips behind the sweep loads: the workgroup that owns walk 0 -- the longest walks of the launch --
    // ended last by that much).  One barrier: nobody writes red[] again before the tr(P) sum, which has its own.
    cplx trW = mkc<R>(R(0), R(0));
    if (has_trace && !QF_PROBE_SKIP(2)) {
        cplx s = mkc<R>(R(0), R(0));
        if (on_diag) {
#pragma unroll
            for (int q = 0; q < L; ++q) {
                if ((k0 + q) < len1) {
                    s.x += v[q].x;
                    s.y += v[q].y;
                }
            }
        }
        block_sum_post<R>(s, red, tid);
        __syncthreads();
        s = block_sum_read<R>(red, nthreads);
        R invN = R(1) / (R)N;
        trW = mkc<R>(s.x * invN, s.y * invN);
    }

#pragma unroll
    for (int s = 0; s < L; ++s) {
        const bool valid = (k0 + s) < len;
        if (!valid) {
            v[s] = mkc<R>(R(0), R(0));
            w[s] = R(0);
            inv[s] = R(0);
        } else if (on_diag && (!FOLD || (k0 + s) < len1)) {
            v[s].x -= trW.x;
            v[s].y -= trW.y;
        }
    }

    QF_PROBE_STAMP(1)
    // ---- pass 1: local forward sweep with zero carry-in
    {
        cplx yprev = mkc<R>(R(0), R(0));
        R cprod = R(1);
#pragma unroll
        for (int s = 0; s < L; ++s) {
            cplx y;
            y.x = fma_r(-w[s], yprev.x, v[s].x);
            y.y = fma_r(-w[s], yprev.y, v[s].y);
            v[s] = y;
            cprod *= -w[s];
            yprev = y;
        }
        if (jc < C) {
            endv[end_idx] = yprev;
            endc[endc_idx] = cprod;
        }
    }
    QF_PROBE_STAMP(2)
    __syncthreads();
    QF_PROBE_STAMP(3)

    // ---- pass 2: chunk carries of the forward recurrence
    if (use_scan && scan_pairs) {
        // lane l composes the maps of chunks 2l and 2l+1, the wavefront scans the 64 compositions, and the
        // second chunk's carry is the first one's map applied to the lane's
        for (int gd = wave; gd < G; gd += nwaves) {
            const int c0 = 2 * lane, c1 = c0 + 1;
            R a0 = R(0), a1 = R(0);
            cplx b0 = mkc<R>(R(0), R(0)), b1 = b0;
            if (c0 < C) {
                a0 = endc[gd * CR + c0];
                b0 = endv[gd * CE + c0];
            }
            if (c1 < C) {
                a1 = endc[gd * CR + c1];
                b1 = endv[gd * CE + c1];
            }
            R a = a1 * a0;
            cplx b = mkc<R>(fma_r(a1, b0.x, b1.x), fma_r(a1, b0.y, b1.y));
            scan_affine(a, b, lane, 64);
            R cx = lane_before(b.x), cy = lane_before(b.y);
            if (lane == 0) cx = cy = R(0);
            if (c0 < C) carry[c0 * GP + gd] = mkc<R>(cx, cy);
            if (c1 < C) carry[c1 * GP + gd] = mkc<R>(fma_r(a0, cx, b0.x), fma_r(a0, cy, b0.y));
        }
    } else if (use_scan) {
        if (C > 32) scan_chunks<R, 64, false>(C, G, CE, CR, GP, endc, endv, carry, lane, wave, nwaves);
        else scan_chunks<R, 0, false>(C, G, CE, CR, GP, endc, endv, carry, lane, wave, nwaves);
    } else if (tid < G) {
        cplx c = mkc<R>(R(0), R(0));
        for (int q = 0; q < C; ++q) {
            carry[q * GP + tid] = c;
            cplx ev = endv[q * G + tid];
            R ec = endc[q * G + tid];
            c.x = fma_r(ec, c.x, ev.x);
            c.y = fma_r(ec, c.y, ev.y);
        }
    }
    QF_PROBE_STAMP(4)
    __syncthreads();
    QF_PROBE_STAMP(5)

    // ---- pass 3: apply the carry, normalise by the pivot:  c_k = y_k / b'_k
    {
        cplx corr = mkc<R>(R(0), R(0));
        if (jc < C) corr = carry[jc * GP + g];
#pragma unroll
        for (int s = 0; s < L; ++s) {
            corr.x *= -w[s];
            corr.y *= -w[s];
            v[s].x = (v[s].x + corr.x) * inv[s];
            v[s].y = (v[s].y + corr.y) * inv[s];
        }
    }

    // ---- pass 4: local backward sweep with zero carry-in:  p_k = c_k - w_{k+1} p_{k+1}
    {
        cplx pnext = mkc<R>(R(0), R(0));
        R dprod = R(1);
#pragma unroll
        for (int s = L - 1; s >= 0; --s) {
            cplx p;
            p.x = fma_r(-w[s + 1], pnext.x, v[s].x);
            p.y = fma_r(-w[s + 1], pnext.y, v[s].y);
            v[s] = p;
            dprod *= -w[s + 1];
            pnext = p;
        }
        __syncthreads();  // every thread has consumed carry[] / endv[] of the forward pass
        if (jc < C) {
            endv[end_idx] = pnext;
            endc[endc_idx] = dprod;
        }
    }
    QF_PROBE_STAMP(6)
    __syncthreads();
    QF_PROBE_STAMP(7)

    // ---- pass 5: chunk carries of the backward recurrence (chunks in reverse order)
    if (use_scan && scan_pairs) {
        for (int gd = wave; gd < G; gd += nwaves) {
            const int c0 = C - 1 - 2 * lane, c1 = c0 - 1;     // (reverse order: c0 is met first)
            R a0 = R(0), a1 = R(0);
            cplx b0 = mkc<R>(R(0), R(0)), b1 = b0;
            if (c0 >= 0) {
                a0 = endc[gd * CR + c0];
                b0 = endv[gd * CE + c0];
            }
            if (c1 >= 0) {
                a1 = endc[gd * CR + c1];
                b1 = endv[gd * CE + c1];
            }
            R a = a1 * a0;
            cplx b = mkc<R>(fma_r(a1, b0.x, b1.x), fma_r(a1, b0.y, b1.y));
            scan_affine(a, b, lane, 64);
            R cx = lane_before(b.x), cy = lane_before(b.y);
            if (lane == 0) cx = cy = R(0);
            if (c0 >= 0) carry[c0 * GP + gd] = mkc<R>(cx, cy);
            if (c1 >= 0) carry[c1 * GP + gd] = mkc<R>(fma_r(a0, cx, b0.x), fma_r(a0, cy, b0.y));
        }
    } else if (use_scan) {
        if (C > 32) scan_chunks<R, 64, true>(C, G, CE, CR, GP, endc, endv, carry, lane, wave, nwaves);
        else scan_chunks<R, 0, true>(C, G, CE, CR, GP, endc, endv, carry, lane, wave, nwaves);
    } else if (tid < G) {
        cplx c = mkc<R>(R(0), R(0));
        for (int q = C - 1; q >= 0; --q) {
            carry[q * GP + tid] = c;
            cplx ev = endv[q * G + tid];
            R ec = endc[q * G + tid];
            c.x = fma_r(ec, c.x, ev.x);
            c.y = fma_r(ec, c.y, ev.y);
        }
    }
    QF_PROBE_STAMP(8)
    __syncthreads();
    QF_PROBE_STAMP(9)

    // ---- pass 6: apply the carry
    {
        cplx corr = mkc<R>(R(0), R(0));
        if (jc < C) corr = carry[jc * GP + g];
#pragma unroll
        for (int s = L - 1; s >= 0; --s) {
            corr.x *= -w[s + 1];
            corr.y *= -w[s + 1];
            v[s].x += corr.x;
            v[s].y += corr.y;
        }
    }

    // ---- m = 0: remove tr(P)/N, cpu.py:342-352.  The wave totals go to red2[] behind everything else in LDS (the
    // staging tile overlays red[]), and the barrier between posting and reading them is the one the staging tile
    // needs anyway: the workgroup that owns walk 0 passes no barrier the others do not
    const bool with_trace = has_trace && !QF_PROBE_SKIP(2);
    if (with_trace) {
        cplx s = mkc<R>(R(0), R(0));
        if (on_diag) {
#pragma unroll
            for (int q = 0; q < L; ++q) {
                if ((k0 + q) < len1) {
                    s.x += v[q].x;
                    s.y += v[q].y;
                }
            }
        }
        block_sum_post<R>(s, red2, tid);
    }

    QF_PROBE_STAMP(10)
    // ---- store (scaled); stage the block's results for the mirrored store
    if (SKEWH || with_trace) __syncthreads();  // carry[] is dead: its memory becomes the staging tile
    if (with_trace && on_diag) {
        const cplx s = block_sum_read<R>(red2, nthreads);
        R invN = R(1) / (R)N;
        R tx = s.x * invN, ty = s.y * invN;
#pragma unroll
        for (int q = 0; q < L; ++q) {
            if (!FOLD || (k0 + q) < len1) {
                v[q].x -= tx;
                v[q].y -= ty;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < L; ++s) {
        const int k = k0 + s;
        cplx p = mkc<R>(v[s].x * scale, v[s].y * scale);
        if (k < len && !QF_PROBE_SKIP(0)) P[QF_ENTRY(k)] = p;
        // (walk-per-slot layout: L * G entries between two chunks are a multiple of 256 bytes -- the four chunks of
        // a 16-lane group would write the same banks; G entries of padding per chunk put them 64 bytes apart.  The
        // folded layout's odd L does that by itself.)
        if (SKEWH && jc < C) ptile[(size_t)(FOLD ? k : k + jc) * G + g] = p;
    }
#undef QF_ENTRY
    if (SKEWH && FOLD) {
        // the mirror of both halves: walks t0+gg (entries k of the slot, target row t0 + u with u = k + gg, columns
        // u .. u-G+1) and walks N-1-t0-gg (entries k' behind the slot's first len1; target row N-1-t0 + u' with
        // u' = k' - gg, columns u' .. u'+G-1): contiguous 16*G-byte row segments either way
        __syncthreads();
        const int gg = tid % G, uu = tid / G, upb = nthreads / G;
        const int tt = t0 + gg;
        const int l1 = (tt < T) ? N - tt : 0;
        const int l2 = (tt < T && N - 1 - tt != tt) ? tt + 1 : 0;
        if (!QF_PROBE_SKIP(1)) {
            const int umax = N - t0 + G - 1;
            if (tt != 0) {
                for (int u0 = uu; u0 < umax; u0 += 4 * upb) {
                    cplx p[4];
                    bool ok[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int u = u0 + q * upb, k = u - gg;
                        ok[q] = (u < umax && k >= 0 && k < l1);
                        p[q] = ptile[(size_t)(ok[q] ? k : 0) * G + gg];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int u = u0 + q * upb, k = u - gg;
                        if (ok[q]) P[(size_t)(t0 + u) * N + k] = mkc<R>(-p[q].x, p[q].y);
                    }
                }
            }
            const int vmax = t0 + G;              // v = u' + G - 1 = 0 .. t0 + G - 1
            for (int v0 = uu; v0 < vmax; v0 += 4 * upb) {
                cplx p[4];
                bool ok[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int kk = v0 + q * upb - (G - 1) + gg;
                    ok[q] = (v0 + q * upb < vmax && kk >= 0 && kk < l2);
                    p[q] = ptile[(size_t)(ok[q] ? l1 + kk : 0) * G + gg];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int up = v0 + q * upb - (G - 1), kk = up + gg;
                    if (ok[q]) P[(size_t)(N - 1 - t0 + up) * N + kk] = mkc<R>(-p[q].x, p[q].y);
                }
            }
        }
    } else if (SKEWH) {
        QF_PROBE_STAMP(11)
        // (i,j) = (k, k+t)  ->  P[j,i] = -conj(P[i,j]), cpu.py:334,340.  With u = k + g the
        // targets of a fixed u are row t0+u, columns u, u-1, .., u-G+1: one contiguous segment.
        __syncthreads();
        const int gg = tid % G, uu = tid / G, upb = nthreads / G;
        const int tt = t0 + gg;
        const int lent = (tt < N) ? N - tt : 0;
        // (k < N - tt means u = k + gg < N - t0: no row beyond that has an entry -- at N = 512 the walk-0 workgroup,
        // which ends last, makes two trips of 256 rows instead of three)
        const int umax = min(C * L + G - 1, N - t0);
        if (tt != 0 && !QF_PROBE_SKIP(1)) {
            // four rows per trip: the LDS reads of a trip are in flight together, then its stores
            for (int u0 = uu; u0 < umax; u0 += 4 * upb) {
                cplx p[4];
                bool ok[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int u = u0 + q * upb, k = u - gg;
                    ok[q] = (u < umax && k >= 0 && k < lent);
                    p[q] = ptile[(size_t)(ok[q] ? k + k / L : 0) * G + gg];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int u = u0 + q * upb, k = u - gg;
                    if (ok[q]) P[(size_t)(t0 + u) * N + k] = mkc<R>(-p[q].x, p[q].y);
                }
            }
        }
    }
    QF_PROBE_STAMP(12)
    QF_SOLVE_EXIT
#undef QF_SOLVE_EXIT
}

// the deferred decision alone: behind the last second product of a call there is no solve to take it
__global__ __launch_bounds__(256) void k_decide(int N, qf_decide dec)
{
    __shared__ double scratch[40];
    if (!dec.state_rw->pending) return;
    const qf_new_state ns = qf_decide_compute(N, dec.slots, dec.rowpart, dec.state_rw, scratch);
    if (threadIdx.x == 0) qf_decide_apply(dec.state_rw, dec.rec, dec.ticket, ns);
}

struct solve_cfg {
    int L, G, C, threads, fold, T;
    size_t smem;
};

// folded walk slots (k_solve<.., FOLD = 1>) for the skew-Hermitian solve from N = 768 on.  Above N = 1024 the
// walk-per-slot layout needs 32-entry chunks (256 VGPRs + spills, one wavefront per SIMD) and, at N = 2048, two rounds
// of 256 workgroups: 50.1 -> 36.9 us there, 40.2 -> 27.0 at N = 1536 (tools/solve_probe.hip); at N = 768 / 1024 the
// 9-entry chunks shorten the four sweeps (17.0 -> 14.5, 19.6 -> 17.8 us; stepper +1.9 % / +0.7 % with two slots per
// workgroup, i.e. 256 workgroups at N = 1024); below, the stepper loses (N = 512: 9,168 -> 8,895 timesteps/s).
// (QUFLOW_HIP_SOLVE_FOLD=0 / 1: never / from N = 256 on, for A/B runs)
bool fold_wanted(int N, int skewh)
{
    const char *e = getenv("QUFLOW_HIP_SOLVE_FOLD");     // (read per launch: the tests switch it inside one process)
    const int forced = e ? atoi(e) : -1;
    if (!skewh || N + 1 > 17 * 128) return false;
    if (forced == 0) return false;
    if (forced == 1) return N >= 256;
    return N >= 768;
}

solve_cfg pick_cfg(int N, size_t csize = sizeof(cplx), bool fold = false)
{
    solve_cfg c;
    c.fold = fold ? 1 : 0;
    c.T = N;
    if (fold) {
        // N + 1 entries per slot, at most 128 chunks (two per lane of the scanning wavefront)
        c.L = (N + 1 <= 9 * 128) ? 9 : 17;
        c.C = (N + 1 + c.L - 1) / c.L;
        c.T = (N + 1) / 2;
        int G = 64;
        while (G > 1 && G * c.C > 512) G >>= 1;
        // a workgroup per CU matters more than 64-byte row segments (N = 1024: two slots, 256 workgroups)
        while (G > 2 && (c.T + G - 1) / G < 256) G >>= 1;
        c.G = G;
        c.threads = ((G * c.C + 63) / 64) * 64;
        const size_t scan_bytes = (size_t)(c.C + 2) * G * csize + (size_t)c.C * (G + 1) * csize + (size_t)c.threads * csize +
                                  (size_t)(c.C + 4) * G * (csize / 2);
        const size_t tile_bytes = (size_t)c.C * c.L * G * csize;
        c.smem = (((scan_bytes > tile_bytes ? scan_bytes : tile_bytes) + 15) & ~(size_t)15) + 8 * csize;   // + red2[] (16-byte aligned): the tr(P) wave totals
        return c;
    }
    c.L = 16;
    c.C = (N + c.L - 1) / c.L;
    // more than 64 chunks per walk cannot be scanned by one wavefront (the serial carry pass takes over):
    // longer chunks from N > 1024 on
    if (c.C > 64) {
        c.L = 32;
        c.C = (N + c.L - 1) / c.L;
    } else if ((N + 3) / 4 <= 64) {
        // N <= 256 (round 6): 4-step chunks, still one chunk per scanning lane -- N = 256 14,891 -> 15,145-15,201 timesteps/s
        // (+1.9 %, three same-box pairs, profiles/r06_solve_l4_ab.txt).  At N = 512 four-step chunks mean 128 chunks per walk,
        // scanned two per lane: 9,783-9,831 -> 9,755-9,787 with 4 walks per workgroup, 9,698-9,708 with 2 -- not taken.
        c.L = 4;
        c.C = (N + 3) / 4;
    } else if ((N + 7) / 8 <= 64) {
        // N <= 512: 8-step chunks still fit one wavefront's scan (<= 64 chunks per walk) -- half the serial depth
        // of the four sweeps for one scan step more: N=512 13.6 -> 12.4 us in tools/solve_probe.hip, 8,749 -> 8,922
        // timesteps/s; N=256 12.2 -> 10.7 us
        c.L = 8;
        c.C = (N + 7) / 8;
    }
    const int max_threads = c.L <= 16 ? 512 : 256;  // register budget of k_solve<L>
    int G = 64;
    while (G > 1 && G * c.C > max_threads) G >>= 1;
    // The solve is bound by per-CU load/store bandwidth, not by HBM: spread it over all 256
    // CUs (64-byte row segments per walk group are still whole L2 requests)
    while (G > 4 && (N + G - 1) / G < 256) G >>= 1;
    c.G = G;
    c.threads = ((G * c.C + 63) / 64) * 64;
    // chunk-end values and carries (complex), chunk-end products (real), reduction scratch (complex)
    const size_t scan_bytes = (size_t)(c.C + 2) * G * csize + (size_t)c.C * (G + 1) * csize + (size_t)c.threads * csize +
                              (size_t)(c.C + 4) * G * (csize / 2);        // (padded strides: see the kernel)
    const size_t tile_bytes = (size_t)c.C * (c.L + 1) * G * csize;   // mirror staging (skew-Hermitian solve), one padding row per chunk
    c.smem = (((scan_bytes > tile_bytes ? scan_bytes : tile_bytes) + 15) & ~(size_t)15) + 8 * csize;   // + red2[] (16-byte aligned): the tr(P) wave totals
    return c;
}

template <typename R>
int launch_solve(qf_ctx *ctx, const typename rt<R>::C *tab, const typename rt<R>::C *W, typename rt<R>::C *P, R scale, int skewh,
                 const qf_guard &guard, const qf_decide *decp = nullptr)
{
    qf_decide dec;
    if (decp) dec = *decp;
    const int N = ctx->N;
    solve_cfg c = pick_cfg(N, sizeof(typename rt<R>::C), fold_wanted(N, skewh));
    if (c.G * c.C > (c.L <= 17 ? 512 : 256)) {
        qf_set_error("qf_launch_solve: N=%d too large for the chunked solver", N);
        return QF_ERR_INVALID;
    }
    const int T = c.fold ? c.T : (skewh ? N : N + 1);
    unsigned blocks = (unsigned)((T + c.G - 1) / c.G);
    dim3 grid(blocks), block(c.threads);
    if (c.smem > 160 * 1024) {
        qf_set_error("qf_launch_solve: N=%d needs %zu bytes of LDS", N, c.smem);
        return QF_ERR_INVALID;
    }
#define QF_SOLVE_F(LL, SK, FO)                                                                      \
    {                                                                                               \
        static qf_smem_attr attr;                                                                   \
        QF_TRY(qf_smem_attr_set(attr, (const void *)k_solve<R, LL, SK, FO>, ctx->device, c.smem));  \
        hipLaunchKernelGGL((k_solve<R, LL, SK, FO>), grid, block, c.smem, ctx->stream, N, c.G, c.C, W, P, tab,  \
                           scale, guard, dec, (int)c.smem - 8 * (int)sizeof(typename rt<R>::C)); \
    }
#define QF_SOLVE(LL, SK) QF_SOLVE_F(LL, SK, 0)
    qf_plan_note(ctx, 0x4000000ull | (unsigned long long)(c.L << 16 | c.G << 8 | c.fold << 2 | (skewh ? 2 : 0) | (sizeof(R) == 4 ? 1 : 0)),
                 "{\"kernel\": \"k_solve<%s, L=%d, %s%s>\", \"chunk\": %d, \"chunks_per_walk\": %d, \"walks_per_workgroup\": %d, "
                 "\"workgroups\": %u, \"threads\": %d, \"lds_bytes\": %zu, \"step_end\": \"%s\"}",
                 sizeof(R) == 4 ? "float" : "double", c.L, skewh ? "skew-Hermitian" : "general", c.fold ? ", folded walk slots" : "", c.L, c.C,
                 c.G, blocks, c.threads, c.smem, dec.state_rw ? "takes the deferred decision of the previous iteration" : "none");
    if (c.fold) {
        if (c.L == 9) QF_SOLVE_F(9, 1, 1) else QF_SOLVE_F(17, 1, 1)
    } else if (c.L == 4) {
        if (skewh) QF_SOLVE(4, 1) else QF_SOLVE(4, 0)
    } else if (c.L == 8) {
        if (skewh) QF_SOLVE(8, 1) else QF_SOLVE(8, 0)
    } else if (c.L == 16) {
        if (skewh) QF_SOLVE(16, 1) else QF_SOLVE(16, 0)
    } else {
        if (skewh) QF_SOLVE(32, 1) else QF_SOLVE(32, 0)
    }
#undef QF_SOLVE
#undef QF_SOLVE_F
    QF_HIP(hipGetLastError());
    return QF_OK;
}

}  // namespace

int qf_launch_lap_table(qf_ctx *ctx, int bc, double *lap_dev)
{
    size_t NN = (size_t)ctx->N * ctx->N;
    int threads = 256;
    unsigned blocks = (unsigned)((NN + threads - 1) / threads);
    hipLaunchKernelGGL(k_lap_table<double>, dim3(blocks), dim3(threads), 0, ctx->stream, ctx->N, bc, lap_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_build_factors(qf_ctx *ctx, const double *lap_dev, qf_factors f)
{
    int threads = 64;
    unsigned blocks = (unsigned)((ctx->N + 1 + threads - 1) / threads);
    hipLaunchKernelGGL((k_build_factors<double, double2>), dim3(blocks), dim3(threads), 0, ctx->stream, ctx->N, lap_dev, f.tab);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_solve(qf_ctx *ctx, const qf_factors &f, const cplx *W, cplx *P, double scale, int skewh,
                    qf_guard guard, const qf_decide *dec)
{
    return launch_solve<double>(ctx, f.tab, W, P, scale, skewh, guard, dec);
}

int qf_launch_decide(qf_ctx *ctx, const qf_decide &dec)
{
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(256), 0, ctx->stream, ctx->N, dec);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_laplace(qf_ctx *ctx, const cplx *P, cplx *W)
{
    const int N = ctx->N;
    dim3 block(256), grid((N + 255) / 256, N);
    hipLaunchKernelGGL((k_laplace<double, double2>), grid, block, 0, ctx->stream, N, P, W);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

// ---- complex64 data: float32 tables, float32 arithmetic (cpu.py:725) ----
int qf_launch_lap_table_f32(qf_ctx *ctx, int bc, float *lap_dev)
{
    size_t NN = (size_t)ctx->N * ctx->N;
    int threads = 256;
    unsigned blocks = (unsigned)((NN + threads - 1) / threads);
    hipLaunchKernelGGL(k_lap_table<float>, dim3(blocks), dim3(threads), 0, ctx->stream, ctx->N, bc, lap_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_build_factors_f32(qf_ctx *ctx, const float *lap_dev, float2 *tab)
{
    int threads = 64;
    unsigned blocks = (unsigned)((ctx->N + 1 + threads - 1) / threads);
    hipLaunchKernelGGL((k_build_factors<float, float2>), dim3(blocks), dim3(threads), 0, ctx->stream, ctx->N, lap_dev, tab);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_solve_f32(qf_ctx *ctx, const float2 *tab, const float2 *W, float2 *P, float scale, int skewh, qf_guard guard)
{
    return launch_solve<float>(ctx, tab, W, P, scale, skewh, guard, nullptr);
}

int qf_launch_laplace_f32(qf_ctx *ctx, const float2 *P, float2 *W)
{
    const int N = ctx->N;
    dim3 block(256), grid((N + 255) / 256, N);
    hipLaunchKernelGGL((k_laplace<float, float2>), grid, block, 0, ctx->stream, N, P, W);
    QF_HIP(hipGetLastError());
    return QF_OK;
}
