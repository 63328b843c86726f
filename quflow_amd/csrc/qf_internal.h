// Internal declarations shared by the HIP translation units of libquflow_hip.so.
// gfx950 (MI355X) only; no CUDA paths, no CPU fallbacks.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/quflow_hip.h"

typedef double2 cplx;  // interleaved (re, im): numpy complex128 layout

// Every device allocation of the library goes through these two (api_context.hip).  Normally they ARE hipMalloc / hipFree.
// With QUFLOW_HIP_DEBUG_GUARD=1 in the environment (read once per process) each allocation is framed by two 64 KiB guard
// zones filled with a byte pattern; the zones are read back when the allocation is freed and on qf_debug_guard_check():
// a kernel that stores outside its operand -- which the allocator's 2 MiB granularity would otherwise swallow silently --
// is reported with the allocation's size and the first damaged offset.  Test infrastructure (tests/conftest.py checks at
// the end of a session run under the variable); costs nothing when the variable is unset.
hipError_t qf_guard_malloc(void **p, size_t bytes);
hipError_t qf_guard_free(void *p);
#define hipMalloc(p, n) qf_guard_malloc((void **)(p), (n))
#define hipFree(p) qf_guard_free((void *)(p))

void qf_set_error(const char *fmt, ...);
struct qf_ctx;
// a launcher's note of what it launches for the role whose prof_scope is open (qf_api.h); `key` != 0 names the
// configuration: the JSON fragment is formatted only when it changes (defined below, behind qf_ctx)
inline void qf_plan_note(qf_ctx *ctx, unsigned long long key, const char *fmt, ...) __attribute__((format(printf, 3, 4)));

#define QF_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess) {                                                        \
            qf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return QF_ERR_HIP;                                                         \
        }                                                                              \
    } while (0)

#define QF_TRY(call)              \
    do {                          \
        int _r = (call);          \
        if (_r != QF_OK) return _r; \
    } while (0)

// Factorisation of one tridiagonal coefficient table (data independent), one interleaved pair per entry:
//   tab[e].x = a_k / b'_{k-1}   (multiplier of the forward sweep; 0 at a diagonal's head)
//   tab[e].y = 1 / b'_k         (reciprocal pivot)
// e is the flat matrix index of entry (i,j); see poisson.hip.  (One 16-byte load per entry and sweep
// step instead of two 8-byte ones: a third fewer load instructions in k_solve.)
struct qf_factors {
    double2 *tab = nullptr;
};

// ---- device-resident control state of the stepper (isospectral.py:463-611 loop nest).
// The data-dependent exit of the fixed-point iteration (isospectral.py:535) is decided ON THE
// DEVICE; every hot-path kernel carries a tag (step, iteration) and turns into a no-op when
// the tag does not match the state, so the host can enqueue ahead without ever blocking on a
// residual read-back.  See api_isomp.hip (qf_isomp) for the protocol.
// values of qf_dev_state::fault / qf_host_record::fault
#define QF_FAULT_WAIT 1
#define QF_FAULT_NONFINITE 2
// A checked residual that is inf / NaN CLOSES THE CALL on the device (the reference raises at that iteration with W as the
// last completed step left it, isospectral.py:534): the deciding thread parks the step counter here, where no launch tag
// can match it -- every launch still queued is a no-op, no step end flips or updates W -- and the host stops enqueueing
// when it sees the counter (fused protocol: in the progress word) or qf_host_record::nonfinite.
#define QF_STEP_ABORTED 0x40000000
#define QF_FINITE(x_) ((x_) <= 1.7976931348623157e308 && (x_) >= -1.7976931348623157e308)     // false for NaN too

struct qf_dev_state {
    double resnorm;              // last checked residual of the current step (inf at step start)
    double tol;
    long long total_iterations;  // isospectral.py:426,478
    long long number_of_maxit;   // isospectral.py:427,540
    int step_index;              // completed steps
    int iters_this_step;         // iterations executed in the current step
    int step_done;               // the break of isospectral.py:535-536 was taken
    int minit, maxit;
    int dw_parity;               // which buffer of the dW ping-pong pair holds the current dW
    int fault;                   // QF_FAULT_*: a bounded device-side wait ran out (k_zgemm_tri) / a checked residual is not finite; checked by qf_isomp
    // fused step end (k_zgemm_tri's last finisher decides, advances and flips these)
    int w_parity;                // which buffer of the W pair holds the current state
    int wh_sel;                  // which Whalf buffer the next iteration reads: 0 = W + dW (same step),
                                 // 1 = W_next + dW (first iteration of the next step)
    // deferred step end (N <= 512, k_zgemm_tri32): the second product of iteration `pending_iter` has
    // left its row sums and nothing else -- the NEXT launch that looks at the state (k_solve, k_decide) takes the
    // exit decision before anything else
    int pending;
    int pending_iter;
};


// what the host polls (pinned, coherent): written by the step bookkeeping at the end of k_update
struct qf_host_record {
    unsigned long long seq;      // number of step-end bookkeeping executions (release-stored last)
    long long total_iterations;
    long long number_of_maxit;
    double resnorm;
    int step_index;
    int last_step_iters;         // iterations the most recently completed step took
    int incomplete;              // 1: the advance found its step still unfinished
    int nonfinite;               // 1: the residual of an exit test was inf / NaN and the call was closed (QF_STEP_ABORTED); written by
                                 // the deciding thread only -- `fault` below belongs to the waiting workgroups, neither overwrites the other
    // fused protocol: (completed steps << 32) | iterations executed in the current step, one
    // 8-byte system-scope store per executed iteration (torn-free for the polling host)
    unsigned long long progress;
    // what qf_isomp needs when the call is over, published with `progress` (no device read-back)
    double tol;                  // tolerance in force (k_state_init; the automatic one is formed on the device)
    int w_parity, wh_sel, dw_parity;
    int fault;                   // QF_FAULT_WAIT: a bounded device-side wait ran out (written by the waiting workgroup itself)
};

// what a deferred decision needs (k_solve, k_decide)
struct qf_decide {
    const double *rowpart = nullptr;
    int slots = 0;
    qf_dev_state *state_rw = nullptr;
    qf_host_record *rec = nullptr;
    unsigned *ticket = nullptr;      // arrivals of the deciding launch's workgroups: the last one writes the state
};

struct qf_guard {
    const qf_dev_state *state = nullptr;  // nullptr: unconditional launch
    int step = 0;
    int iter = 0;
    const void *alt = nullptr;            // fused protocol: the input to use instead when state->wh_sel != 0
};

#ifdef __HIPCC__
// iteration kernels run only for (current step, next iteration, not yet converged)
__device__ __forceinline__ bool qf_guard_iter(const qf_guard &g)
{
    if (!g.state) return true;
    return g.state->step_index == g.step && g.state->step_done == 0 && g.state->iters_this_step == g.iter;
}
// step-end kernels run only once the current step's iteration has finished
__device__ __forceinline__ bool qf_guard_step_end(const qf_guard &g)
{
    if (!g.state) return true;
    return g.state->step_index == g.step && (g.state->step_done != 0 || g.state->iters_this_step >= g.state->maxit);
}
// Sum over the 16 lanes of a DPP row, in every lane, as the xor butterfly 1, 2, 4, 8 -- registers only.  A
// `__shfl_xor` of a double is two ds_bpermute round trips per step (32 dependent LDS round trips for the eight row
// sums of an epilogue).  Same tree, same bits: after the steps 1 and 2 a quad's four lanes hold one value, so the
// half-row / row MIRRORS hand every lane the value its xor-4 / xor-8 partner holds.
template <int CTRL> __device__ __forceinline__ double qf_dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double qf_row16_sum(double v)
{
    v += qf_dpp_f64<0xB1>(v);      // quad_perm [1,0,3,2]
    v += qf_dpp_f64<0x4E>(v);      // quad_perm [2,3,0,1]
    v += qf_dpp_f64<0x141>(v);     // row_half_mirror
    v += qf_dpp_f64<0x140>(v);     // row_mirror
    return v;
}

// |er + i ei| for the residual row sums of |dW_old - dW| (isospectral.py:526,534): every kernel that forms them calls
// THIS function, so the protocols stay bit-identical to each other.  It is hipcc's own correctly rounded double-precision
// square root (v_rsq_f64, one Goldschmidt step, two Newton corrections -- the same operations in the same order, hence
// the same bits) WITHOUT its range scaling: the compare / select / two ldexp that rescue arguments below 2^-767 cost
// 5 of the 17 instructions of each of the 64 roots a lane takes in an epilogue, and a residual entry below 1e-115 is
// zero to any row sum this code can meet (there the result is merely less accurate, never wrong in kind: for
// er^2 + ei^2 below 2^-767 it is NOT correctly rounded).  The "same bits as sqrt()" claim rests on this toolchain's
// expansion of sqrt: tests/test_hip_parity.py::test_modulus_is_the_compilers_square_root compares the two bit for bit
// (qf_debug_modulus) over random and edge arguments, so a compiler that expands sqrt differently is caught there.
__device__ __forceinline__ double qf_modulus(double er, double ei)
{
    const double x = er * er + ei * ei;
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return (x > 0.0 && x < __builtin_inf()) ? g : x;     // (0 -> 0, inf -> inf, NaN -> NaN)
}
#endif

// ---- complex64 data (single.hip; poisson.hip instantiates the solve for float): the float32 working set of a
// context, allocated on first use.  The control plane (qf_dev_state, tickets, progress record, row-sum slots in
// double) is the double-precision path's.
struct qf_c64 {
    float2 *W = nullptr;                       // state
    float2 *dW[2] = {nullptr, nullptr};        // iteration vector, ping-pong
    float2 *Whalf = nullptr, *Phalf = nullptr, *PW = nullptr;
    float2 *kahan_c = nullptr;                 // compensation term (compsum), on demand
    float2 *stage = nullptr;                   // staging of the host-in / host-out entry points
    float *lap = nullptr;                      // (N,N,2) float32 coefficient table of the Poisson problem (cpu.py:725)
    float2 *tab = nullptr;                     // its factorisation {w, 1/b'} in float32
    double *rowpart = nullptr;                 // [column tiles of 64][N] partial row sums of the second product
    int rowpart_tiles = 0;
    int dw_cur = 0;
    bool increment_valid = false;
    float2 *W2 = nullptr, *Whalf2 = nullptr;   // fused step end: second buffers of the W / Whalf pairs (on demand)
    // upper-triangle second product (k_cgemm_tri): exchange area [tiles][4][64*64], arrival counters [tiles], K pieces
    // per off-diagonal / diagonal tile; `tri` = selected for the running call (W exactly skew-Hermitian)
    float2 *tri_partial = nullptr;
    unsigned *tri_arrive = nullptr;
    size_t tri_arrive_count = 0;               // (counters in tri_arrive: one per tile on or above the diagonal)
    int tri_split = 2, tri_split_diag = 2;
    bool tri_allowed = true, tri = false, w_skew_known = false;
};
struct qf_ctri {
    float2 *partial = nullptr;
    unsigned *arrive = nullptr;
    int split = 2, split_diag = 2;
};
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device and is needed once per kernel and
// device: a per-call-site record of what has been set where (one process may drive several devices, one context each)
// The records are function-local statics shared by every host thread: contexts of different sizes driven from different
// threads raise one kernel's limit concurrently (k_solve at N = 512 and N = 1024), so raising is serialised and the
// record only ever grows (a lost update would leave the attribute BELOW a recorded size and fail later launches).
struct qf_smem_attr {
    std::atomic<size_t> bytes[64] = {};
};
inline std::mutex &qf_smem_attr_mutex()
{
    static std::mutex m;
    return m;
}
inline int qf_smem_attr_set(qf_smem_attr &a, const void *fn, int device, size_t bytes)
{
    const int d = device & 63;
    if (bytes > 64 * 1024 && bytes > a.bytes[d].load(std::memory_order_acquire)) {
        std::lock_guard<std::mutex> lock(qf_smem_attr_mutex());
        if (bytes > a.bytes[d].load(std::memory_order_relaxed)) {
            QF_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            a.bytes[d].store(bytes, std::memory_order_release);
        }
    }
    return QF_OK;
}

struct qf_epilogue_f {
    const float2 *PW = nullptr;
    const float2 *W = nullptr;
    float2 *dW[2] = {nullptr, nullptr};
    float2 *Whalf = nullptr;
    double *rowpart = nullptr;
    // fused step end (as qf_epilogue's: DESIGN.md 4b): W pair, the next step's Whalf, the tile ticket and what the
    // last tile's workgroup updates
    int fused = 0;
    float2 *Wpair[2] = {nullptr, nullptr};
    float2 *Whalf_step = nullptr;
    unsigned *ticket = nullptr;
    int n_tiles = 0;
    qf_dev_state *state_rw = nullptr;
    qf_host_record *rec = nullptr;
};

struct qf_event_pair {
    hipEvent_t start, stop;
    int kernel_id;
};

struct qf_ctx {
    int N = 0;
    int device = 0;
    hipStream_t stream = nullptr;

    // state and work matrices, each N*N complex128
    cplx *W = nullptr;       // vorticity state
    bool w_skew_known = false;     // W[j,i] == -conj(W[i,j]) exactly: verified at the entry of a qf_isomp call and
                                   // kept by every isomp update (W += 2 (PW - PW^H)); cleared by whatever else writes W
    cplx *dW[2] = {nullptr, nullptr};  // iteration vector, ping-pong (cur / new)
    int dw_cur = 0;
    bool increment_is_zero = true; // this call starts from dW = 0 (not a qf_isomp_continue)
    bool c64_increment_is_zero = true;   // the same for the complex64 buffers
    int pred_first_iters = 0;      // iterations the cold first step of the previous call needed
    bool increment_valid = false;  // dW[dw_cur] holds the increment of the last qf_isomp call (qf_isomp_continue)
    cplx *W2 = nullptr;      // fused protocol: second buffer of the W pair (allocated on demand)
    cplx *Whalf2 = nullptr;  //                 the next step's Whalf
    bool fused_allowed = true;   // QUFLOW_HIP_FUSED=0 disables the fused step end
    cplx *Whalf = nullptr;   // W + dW
    cplx *Phalf = nullptr;   // eps * Delta^-1 Whalf
    cplx *PW = nullptr;      // Phalf @ Whalf
    cplx *kahan_c = nullptr; // compensation term (compsum), allocated on demand
    cplx *stage = nullptr;   // staging for host-in/host-out entry points
    // int8 digit-split products (ozaki.hip): sliced operands Phalf (A form), Phalf (B form), Whalf (B form),
    // PW (A form), allocated on demand; QUFLOW_HIP_GEMM=i8 / fp64 selects
    signed char *oz_planes[4] = {nullptr, nullptr, nullptr, nullptr};
    double *oz_scale[4] = {nullptr, nullptr, nullptr, nullptr};
    bool gemm_i8 = false;                // decided per qf_isomp call (W exactly skew-Hermitian, fused protocol)
    bool gemm_i8_allowed = false;
    bool gemm_i8_hybrid = false;         // QUFLOW_HIP_GEMM=i8h / i8hx6: fp64 first product, digit-split second product
    bool gemm_i8_first = false;          // QUFLOW_HIP_GEMM=i8x6f: digit-split FIRST product, fp64 upper-triangle second product
    int gemm_i8_min_n = 768;
    int oz_digits = 5;             // base-128 digits per real value of the int8 products: 5 ("i8") or 6 ("i8x6")
    int oz_digits2 = 0;            // "i8x65": digits of the SECOND product (0: as the first) -- PW cut into five, Phalf's leading five of six
    // second int8 product on the upper triangle only: the tiles below the diagonal take their
    // partner's result (T = PW@Phalf is skew-Hermitian) through oz_tbuf instead of multiplying
    cplx *oz_tbuf = nullptr;       // one 64 x 64 result tile per upper-triangle tile
    unsigned *oz_tflags = nullptr; // launch epoch per upper-triangle tile: "its tile is in oz_tbuf"
    unsigned oz_epoch = 0;
    double *oz_diag = nullptr;     // [N] Im (Phalf @ Whalf)_ii formed in fp64 by the slicing launch (ozaki.hip, PAIR)
    std::vector<cplx *> multi;   // per-state buffers of qf_isomp_states (allocated on demand, kept)
    double *multi_rowpart = nullptr;
    cplx *hook_host[3] = {nullptr, nullptr, nullptr};   // pinned staging of the hooked steppers (hooks.hip), on demand
    size_t hook_host_bytes = 0;
    cplx *ns_inv = nullptr;  // Newton-Schulz inverse of I - E (isomp_simple / isomp_quasinewton), on demand
    cplx *ns_tmp = nullptr;

    double *lap = nullptr;   // (N,N,2) coefficient table of the Poisson problem (bc=True)
    qf_factors poisson;      // its factorisation
    // qf_solve_tridiagonal cache: factorised tables by caller key, least-recently-used entries recycled
    // once the byte budget is reached (QUFLOW_HIP_FACTOR_CACHE_MB, default 512); every entry carries a
    // fingerprint of the table it was built from, and a key that comes back with another table is refactored
    struct factor_entry {
        qf_factors f;
        unsigned long long fingerprint = 0;
        unsigned long long last_used = 0;
    };
    std::map<unsigned long long, factor_entry> user_factors;
    unsigned long long factor_clock = 0;
    size_t factor_budget_bytes = (size_t)512 << 20;
    double *lap_user = nullptr;

    // spherical-harmonics transforms (quantization.hip): basis resident in HBM, m-major staging
    double *basis = nullptr;     // N(N+1)(2N+1)/6 doubles (quantization.py:68-113), uploaded once
    cplx *sh_stage = nullptr;    // 4 x N(N+1)/2 complex: packed coefficient / diagonal vectors
    double *sh_omega = nullptr;  // 2 N^2 doubles: omega on the device (real or complex)

    double *rowpart = nullptr;   // [tiles_n][N] partial row sums from the GEMM2 epilogue
    int rowpart_tiles = 0;
    double *rowsum = nullptr;    // [N]
    double *scalars = nullptr;   // small device scratch for reductions (>= 4096 doubles)
    double *host_scalars = nullptr;  // pinned host mirror (>= 16 doubles)
    qf_dev_state *state = nullptr;       // device control state of the stepper
    qf_host_record *host_rec = nullptr;  // pinned + coherent, polled by the host
    // qf_isomp_diag: the diagnostics are enqueued behind the last step, before the call's one synchronisation
    bool diag_at_exit = false, diag_valid = false;
    unsigned *ticket = nullptr;          // block counter of k_update (last block does the step bookkeeping)
    int pred_iters = 3;                  // iterations/step the recent steps needed (enqueue-ahead hint)
    // upper-triangle stream-K form of the second product (k_zgemm_tri): allowed by
    // QUFLOW_HIP_GEMM2 != "full" and N % 64 == 0; switched on per qf_isomp call when W is skew-Hermitian
    bool gemm_tri_allowed = true;
    bool gemm_tri = false;
    int gemm_tri_min_n = 960;            // QUFLOW_HIP_TRI_MIN_N: below (and for N % 64 != 0), the upper triangle is cut into 32x32 tiles
                                         // (k_zgemm_tri32).  Stepper, stream-K / tri32 second product behind a 32x32 first product:
                                         // N = 768 3,926 / 3,933 timesteps/s, 832 3,653 / 3,680, 896 2,917 / 3,015, 960 2,685 / 2,808
                                         // (960 with its 64x64 first product and stream-K: 2,745); 1088 2,000 / 1,891, 1280 1,336 / 1,263
    // upper triangle of 32x32 tiles with the K range of a tile split over two workgroups (k_zgemm_tri32):
    // N % 32 == 0 where the stream-K form is not taken
    bool gemm_tri32 = false;
    int tri32_split = 2, tri32_split_diag = 1;   // pieces per off-diagonal / diagonal tile (qf_fixedpoint_products takes others as an argument)
    cplx *t32_partial = nullptr;
    unsigned *t32_arrive = nullptr;
    // deferred step end with k_zgemm_tri32 (QUFLOW_HIP_DEFER=0 switches it off): decided per call in fused_enter
    bool defer_allowed = true;
    bool defer = false;
    int num_cus = 0;
    cplx *sk_partial = nullptr;          // [sk_slots][64*64] parked partial tiles
    unsigned *sk_flags = nullptr;        // [sk_slots] epoch of the last parked piece, then 16 words (tickets)
    int sk_slots = 0;                    // (0: num_cus -- the diagnostic harnesses that allocate by hand)
    unsigned sk_epoch = 0;
    // weight (in K-tiles) of a finisher's gather + epilogue in the stream-K
    // partition.  0 = plain K-tile split: measured best at N=1024 (E = 0/8/14/20: 91.9/92.5/95.8/100.8 us
    // per second product) -- heavier contributor pieces are parked later than their consumers want
    // them; N=2048 gains 1.4 % at E=8.
    int sk_epi_units = 0;
    int sk_epi_units_fused = 2;          // (round 2, after the epilogue rework: E = 0 / 4 / 8 -> 2506-2515 / 2531-2540 / 2494-2505 steps/s;
                                         //  round 4, epilogue 15.2k -> 11.5k cycles: E = 0 / 2 / 3 / 4 / 6 -> 2595 / 2623 / 2616 / 2611 / 2603)
    int sk_min_units = 8;                // fewest K-tiles a workgroup of k_zgemm_tri takes
    // QUFLOW_HIP_DEBUG_DROP_FLAG (honoured only with QUFLOW_HIP_DEBUG set; tests of the fault paths): the first
    // due second product of this context drops 1 = its piece-flag publications (a device-side wait runs out),
    // 2 = one step-end ticket (the iteration never closes: the host's progress watchdog fires)
    int debug_drop = 0;
    bool needs_reset = false;            // a call ended in an error: counters and flags are rebuilt at the next entry

    qf_c64 *c64 = nullptr;               // complex64 working set (qf_c64_alloc)

    // measurement
    int profile_mask = 0;
    std::vector<qf_event_pair> events_busy;
    std::vector<qf_event_pair> events_free;
    int profile_stride = 1;        // events around every profile_stride-th launch of a kernel id
    // qf_plan_describe: what was launched last for each role (QF_KERNEL_x), written by the launchers (qf_plan_note) while
    // a prof_scope of that role is open; `key` = the configuration the text was formatted for (formatted once per change)
    struct plan_slot {
        unsigned long long key = 0;
        char text[384] = {0};
    } plan[QF_KERNEL_COUNT];
    int plan_role = -1;
    long long prof_seen[QF_KERNEL_COUNT] = {0};       // launches seen while the id was enabled
    long long prof_launches[QF_KERNEL_COUNT] = {0};   // launches measured
    double prof_ms[QF_KERNEL_COUNT] = {0};
    hipEvent_t timer_start = nullptr, timer_stop = nullptr;
};

inline void qf_plan_note(qf_ctx *ctx, unsigned long long key, const char *fmt, ...)
{
    if (!ctx || ctx->plan_role < 0 || ctx->plan_role >= QF_KERNEL_COUNT) return;
    qf_ctx::plan_slot &s = ctx->plan[ctx->plan_role];
    if (key != 0 && s.key == key) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(s.text, sizeof(s.text), fmt, ap);
    va_end(ap);
    s.key = key;
}

// ---- poisson.hip
int qf_launch_lap_table(qf_ctx *ctx, int bc, double *lap_dev);
int qf_launch_build_factors(qf_ctx *ctx, const double *lap_dev, qf_factors f);
int qf_launch_solve(qf_ctx *ctx, const qf_factors &f, const cplx *W, cplx *P, double scale, int skewh,
                    qf_guard guard = qf_guard(), const qf_decide *dec = nullptr);
int qf_launch_decide(qf_ctx *ctx, const qf_decide &dec);      // the deferred decision alone (end of a call)
int qf_launch_laplace(qf_ctx *ctx, const cplx *P, cplx *W);

int qf_launch_lap_table_f32(qf_ctx *ctx, int bc, float *lap_dev);
int qf_launch_build_factors_f32(qf_ctx *ctx, const float *lap_dev, float2 *tab);
int qf_launch_solve_f32(qf_ctx *ctx, const float2 *tab, const float2 *W, float2 *P, float scale, int skewh,
                        qf_guard guard = qf_guard());
int qf_launch_laplace_f32(qf_ctx *ctx, const float2 *P, float2 *W);

// ---- single.hip: complex64 products and elementwise passes
int qf_c64_alloc(qf_ctx *ctx);
void qf_c64_free(qf_c64 *f);
int qf_launch_cgemm(qf_ctx *ctx, const float2 *A, const float2 *B, float2 *C, const qf_epilogue_f *ep, qf_guard guard = qf_guard());
// tile size (32 or 64) of the complex64 second product (and of the row-sum slots it fills): k_cgemm32<EPI> / k_cgemm_tri32 or
// k_cgemm<EPI> / k_cgemm_tri; and of the plain first product: k_cgemm32 or k_cgemm / k_cgemm_ks
int qf_c64_tile(const qf_ctx *ctx);
int qf_c64_tile_first(const qf_ctx *ctx);
// the second product on the upper triangle of 64x64 tiles (requires N % 64 == 0 and qf_c64_tri_alloc)
int qf_c64_tri_alloc(qf_ctx *ctx);
int qf_launch_cgemm_tri(qf_ctx *ctx, const float2 *A, const float2 *B, const qf_epilogue_f *ep, qf_guard guard = qf_guard());
int qf_launch_mirror_lower_f32(qf_ctx *ctx, float2 *X);
int qf_launch_update_f32(qf_ctx *ctx, const float2 *PW, float2 *W, float2 *dW_a, float2 *dW_b, float2 *Whalf, float2 *kahan_c,
                         int reinitialize, qf_guard guard = qf_guard());
int qf_launch_norm_inf_f32(qf_ctx *ctx, const float2 *A, double *out_dev);
int qf_launch_inner2_f32(qf_ctx *ctx, const float2 *A, const float2 *B, double *out_dev);
int qf_launch_skew_defect_f32(qf_ctx *ctx, const float2 *A, double *out_dev);
int qf_launch_lincomb_f32(qf_ctx *ctx, float a, const float2 *X, float b, const float2 *Y, float2 *out);

// ---- zgemm.hip
struct qf_epilogue {
    // fused epilogue of the second product (isospectral.py:499-509,526-534):
    //   dW_new = C + (PW - PW^H);  Whalf = W + dW_new;  rowpart += |dW_old - dW_new|
    const cplx *PW = nullptr;
    const cplx *W = nullptr;
    cplx *dW[2] = {nullptr, nullptr};  // ping-pong pair: old = dW[parity], new = dW[parity ^ 1]
    cplx *Whalf = nullptr;
    double *rowpart = nullptr;
    // fused step end (k_zgemm_tri): W pair (current = Wpair[state->w_parity]; the candidate next state
    // W + 2 (PW - PW^H) goes to the other one) and the next STEP's Whalf = W_next + dW_new
    int fused = 0;
    cplx *Wpair[2] = {nullptr, nullptr};
    cplx *Whalf_step = nullptr;
    // ... and, for the full-product kernel (k_zgemm<.., FUSED>), the tile ticket and what the last
    // tile's workgroup updates (k_zgemm_tri gets these through qf_streamk)
    unsigned *ticket = nullptr;
    int n_tiles = 0;
    qf_dev_state *state_rw = nullptr;
    qf_host_record *rec = nullptr;
    int debug_drop = 0;      // fault-injection build of a launch (QUFLOW_HIP_DEBUG_DROP_FLAG): bit 1 = tile 0 takes no ticket
};
// stream-K exchange area of k_zgemm_tri
struct qf_streamk {
    cplx *partial = nullptr;
    unsigned *flags = nullptr;
    unsigned epoch = 0;
    int *fault = nullptr;
    // fused step end: epilogue ticket (the last of n_tiles epilogues runs the decision)
    unsigned *ticket = nullptr;
    int n_tiles = 0;
    qf_dev_state *state_rw = nullptr;
    qf_host_record *rec = nullptr;
    // bounded waits: polls before a waiting workgroup gives up and raises `fault`
    unsigned spin_limit = 1u << 22;
    // fault injection (QUFLOW_HIP_DEBUG + QUFLOW_HIP_DEBUG_DROP_FLAG, one launch per context; tests only):
    // bit 0 = no workgroup publishes its piece flag, bit 1 = tile 0's epilogue takes no step-end ticket
    int debug_drop = 0;
    int slots = 0;          // 64 KiB slots of the exchange area (and flags in front of the ticket word)
};
// exchange area of k_zgemm_tri32 (upper triangle of 32x32 tiles, K split in two for N < 768)
struct qf_tri32 {
    cplx *partial = nullptr;       // [n_tiles][4][32*32]: the partial tile a workgroup parks before it takes its ticket
    unsigned *arrive = nullptr;    // [n_tiles]: arrivals at a tile (monotone: `split` per executed launch)
    int split = 1;                 // K ranges per off-diagonal tile: 1, 2 or 4
    int split_diag = 1;            // ... per diagonal tile
    unsigned *ticket = nullptr;    // fused step end: epilogue ticket (the last of n_tiles epilogues runs the decision)
    int n_tiles = 0;
    qf_dev_state *state_rw = nullptr;
    qf_host_record *rec = nullptr;
    int debug_drop = 0;            // fault injection: bit 1 = tile 0's epilogue takes no step-end ticket
    int deferred = 0;              // the exit decision is left to the next launch (qf_dev_state::pending)
};
int qf_gemm_tiles_n(int N);
// the second product on the upper triangle of 32x32 tiles (requires ctx->gemm_tri32)
int qf_launch_zgemm_tri32(qf_ctx *ctx, const cplx *A, const cplx *B, const qf_epilogue *ep, qf_guard guard = qf_guard());
// dW = PW @ Phalf + (PW - PW^H) etc. on the upper triangle (requires ctx->gemm_tri)
int qf_launch_zgemm_tri(qf_ctx *ctx, const cplx *A, const cplx *B, const qf_epilogue *ep, qf_guard guard = qf_guard());
int qf_launch_zgemm(qf_ctx *ctx, const cplx *A, const cplx *B, cplx *C, const qf_epilogue *ep,
                    qf_guard guard = qf_guard());

// ---- quantization.hip (device pointers; Nmax = band limit el < Nmax)
int qf_launch_basis(qf_ctx *ctx, double *basis_dev);   // compute_basis on the device
int qf_launch_shr2mat(qf_ctx *ctx, int Nmax, const double *omega_dev, cplx *W_dev);
int qf_launch_mat2shr(qf_ctx *ctx, int Nmax, const cplx *W_dev, double *omega_dev);
int qf_launch_shc2mat(qf_ctx *ctx, const double *omega_dev, cplx *W_dev);
int qf_launch_mat2shc(qf_ctx *ctx, const cplx *W_dev, double *omega_dev);

// ---- ozaki.hip: complex products on the int8 matrix cores from digit-sliced operands
struct qf_oz_job {
    const cplx *X = nullptr;       // matrix to slice, row-wise
    const cplx *X_alt = nullptr;   // fused protocol: used instead when state->wh_sel != 0
    signed char *planes = nullptr;
    double *scale = nullptr;
};
struct qf_oz_mirror {
    cplx *tbuf = nullptr;
    unsigned *flags = nullptr;
    unsigned epoch = 0;            // 0: every tile multiplies
    int *fault = nullptr;
    const double *diag = nullptr;  // plain product: Im C_ii formed in fp64 by the slicing launch (qf_oz_jobs::diag)
    int debug_drop = 0;            // fault injection: 1 = no result-tile flag is published, 2 = tile 0 takes no step-end ticket
};
struct qf_oz_jobs {
    qf_oz_job j[3];
    int n = 0;
    double *diag = nullptr;        // two jobs A, M: one workgroup per row of both, diag[i] = Im (A @ M)_ii in fp64 (ozaki.hip)
};
size_t qf_oz_operand_bytes(int N, int digits);
size_t qf_oz_record_bytes(int N, int digits);      // per operand: N scales, then N x 2 digits int32 digit sums
int qf_launch_oz_slice(qf_ctx *ctx, const qf_oz_jobs &jobs, qf_guard guard = qf_guard(), int digits = 0);
// C = A @ M with M skew-Hermitian, both operands sliced by rows (pa/sa, pm/sm: planes and row scales).
// ep == nullptr: plain product;  ep != nullptr: the second product with the fused epilogue and step end
int qf_launch_oz_gemm(qf_ctx *ctx, const signed char *pa, const double *sa, const signed char *pm, const double *sm,
                      cplx *C, const qf_epilogue *ep = nullptr, qf_guard guard = qf_guard(), int digits = 0, int digits_m = 0,
                      const double *diag = nullptr);

// ---- elementwise.hip
// W += 2(PW - PW^H) at the end of a step.  dW_a/dW_b: the ping-pong pair; the kernel picks the
// current one from the parity of the executed iteration count (device state) when guarded.
int qf_launch_update(qf_ctx *ctx, const cplx *PW, cplx *W, const cplx *dW_a, const cplx *dW_b, cplx *Whalf,
                     cplx *kahan_c, int reinitialize, qf_guard guard = qf_guard());
int qf_launch_debug_modulus(qf_ctx *ctx, int n, const double *er, const double *ei, double *out_mod, double *out_sqrt);
int qf_launch_norm_from_rowpart(qf_ctx *ctx, const double *rowpart, int tiles, double *out_dev);
// residual norm + exit decision of iteration `guard.iter` (isospectral.py:523-536), on device
int qf_launch_norm_decide(qf_ctx *ctx, const double *rowpart, int tiles, qf_guard guard);
// norm_dev != nullptr: automatic tolerance tol = tol_factor * (*norm_dev), formed on the device
// (isospectral.py:440-448: (mach_eps*dt/hb) * |W|_inf) so that the host never waits for the norm
int qf_launch_state_init(qf_ctx *ctx, double tol, int minit, int maxit, const double *norm_dev = nullptr, double tol_factor = 0.0);
int qf_launch_norm_inf(qf_ctx *ctx, const cplx *A, double *out_dev);
// entry of a fused-protocol call in one launch: dW[0] = 0, Whalf = W, |W|_inf -> tolerance (auto_tol), control state reset
int qf_launch_call_begin(qf_ctx *ctx, double tol, int minit, int maxit, int auto_tol, double tol_factor);
// out_dev[0] = sum Re(A conj(B)), out_dev[1] = sum |A|^2 in one pass
int qf_launch_inner2(qf_ctx *ctx, const cplx *A, const cplx *B, double *out_dev);
int qf_launch_inner(qf_ctx *ctx, const cplx *A, const cplx *B, double *out_dev);  // sum Re(A conj(B))
// explicit Runge-Kutta stage on the products A = P@X, B = X@P (B == nullptr: B = A^H, skew-Hermitian case)
int qf_launch_erk_stage(qf_ctx *ctx, const cplx *A, const cplx *B, double inv_hb, const cplx *W, cplx *acc,
                        double c_acc, cplx *Wp, double c_wp, cplx *Wout, double c_fin);
// magmp: magnetic terms into dW / Whalf + residual row sums (slots: ceil(N/32) column tiles of 32)
int qf_launch_magnetic_fix(qf_ctx *ctx, const cplx *BTP, const cplx *BT, cplx *dW, const cplx *dW_old, const cplx *W,
                           cplx *Whalf, double *rowpart);
int qf_launch_magnetic_update(qf_ctx *ctx, const cplx *BT, cplx *W, const cplx *dW, cplx *Whalf);
int qf_launch_lincomb(qf_ctx *ctx, double a, const cplx *X, double b, const cplx *Y, double c, cplx *out);  // a X + b Y + c I
int qf_launch_neg_conj_transpose(qf_ctx *ctx, const cplx *X, cplx *out);                                       // -X^H
int qf_launch_mirror_lower(qf_ctx *ctx, cplx *X);                                                              // X[j,i] = -conj(X[i,j]), i < j
int qf_launch_sum_rowpart(qf_ctx *ctx, const double *rowpart, int tiles, double *rowsum_dev);
// out_dev[0] = max_ij |A[i,j] + conj(A[j,i])|, out_dev[1] = max_ij |A[i,j]|
int qf_launch_skew_defect(qf_ctx *ctx, const cplx *A, double *out_dev);
