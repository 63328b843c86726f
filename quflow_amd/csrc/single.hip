// complex64 data: the float32 arithmetic path of the stepper.
//
// Reference: complex64 input is computed in single precision throughout -- float32 coefficient tables and a
// float32 Thomas solve (quflow/laplacian/cpu.py:725, `dtype=type(W[0,0].real)`), complex64 np.matmul for the two
// products (quflow/integrators/isospectral.py:496,499), complex64 elementwise passes, and the automatic tolerance
// from the float32 machine epsilon (isospectral.py:440-448).  This file holds the float32 kernels that are not
// instantiations of the double-precision ones (poisson.hip instantiates the solve for float):
//   * k_cgemm / k_cgemm32: complex64 N x N x N product on the fp32 matrix cores (v_mfma_f32_32x32x2_f32 / 16x16x4: exact
//     f32 fma chains at 64 flop/clk/SIMD = 157.3 TFLOP/s, MI355X_MICROARCH.md), 3M form like the fp64 kernel, with the
//     fused epilogue of the second product (isospectral.py:499-509, 481-482, 526-534) and the fused step end;
//     k_cgemm_ks / k_cgemm32<.., KS>: the plain product with the K range cut between groups of wavefronts of one
//     workgroup (launches with one tile per CU);
//   * k_cgemm_tri / k_cgemm_tri32: the second product on the upper triangle only (skew-Hermitian state), K pieces per
//     tile exchanged through memory, the last arrival combines and runs the epilogue for the tile and its mirror image;
//   * the end-of-step update (with the Kahan variant, isospectral.py:553-586, contraction off), the infinity
//     norm, the diagnostics' inner products and the skew-Hermitian check on complex64 matrices.
// The control plane is shared with the double-precision path: tagged launches, device-side exit decision on double row
// sums (in the second product's last tile, qf_step_end.h; k_norm_decide in the two-kernel protocol), progress record --
// see api_isomp.hip.
#include "qf_internal.h"
#include "qf_step_end.h"

#pragma clang fp contract(off)  // Kahan summation must not be re-associated or fused; table arithmetic is the reference's

typedef float v16f __attribute__((ext_vector_type(16)));

namespace {

// two neighbouring complex64 entries as one 16-byte load.  With an exact tiling (N a multiple of the tile) every such
// address is 16-byte aligned; on the guarded paths N may be odd, and then the entries of an odd row sit on 8-byte
// boundaries only -- the access is typed accordingly (a plain float4 dereference promises 16 to the compiler;
// global_load_dwordx4 itself takes any dword-aligned address).
// (ONE vector load with the alignment it really has.  A struct of four floats typed 8-byte aligned was tried first: the
// optimiser takes it apart into scalar loads and re-merges them with the single-entry load of the guard's other arm --
// two 8-byte loads per site where there was one 16-byte load, N = 1000 / 1500 at 0.84 of their round-3 rate,
// profiles/r04_size_sweep.jsonl.)
typedef float qf_v4f_a8 __attribute__((ext_vector_type(4), aligned(8)));
template <bool ALIGNED16>
__device__ __forceinline__ float4 ld4(const float2 *p)
{
    if constexpr (ALIGNED16) return *reinterpret_cast<const float4 *>(p);
    else {
        const qf_v4f_a8 v = *reinterpret_cast<const qf_v4f_a8 *>(p);
        return make_float4(v[0], v[1], v[2], v[3]);
    }
}

constexpr int CBM = 64, CBN = 64, CBK = 16;       // block tile (complex entries)
constexpr int SA = CBM + 1;                       // k-major A image: row stride padded by one entry (transposing
                                                  // ds_write_b64 of 8 k-pairs x 2 rows: 16 distinct 8-byte slots)
constexpr int SB = CBN;
constexpr int A_BYTES = CBK * SA * (int)sizeof(float2);
constexpr int B_BYTES = CBK * SB * (int)sizeof(float2);
constexpr int CG_MAIN_BYTES = 2 * (A_BYTES + B_BYTES);
// below N = 768 a 64 x 64 tiling leaves most CUs without a workgroup (64 tiles at N = 512): 32 x 32 block tiles
// there, one 16 x 16 MFMA tile (v_mfma_f32_16x16x4_f32) per wavefront.  Its A image stays row-major with an odd
// row stride: the fragment read of lane (i = l & 15, k = l >> 4) then hits 32 distinct bank pairs per lane group
// (34 i mod 64 are sixteen distinct even banks, the k-neighbours sit two banks further), and so do the staging
// writes (rows r, r+1 x eight k-pairs).
constexpr int SBM = 32, SBN = 32;
constexpr int SAK = CBK + 1;                      // row stride (complex entries) of the small tile's row-major A image
constexpr int SA_BYTES = SBM * SAK * (int)sizeof(float2);
constexpr int SB_BYTES = CBK * SBN * (int)sizeof(float2);
constexpr int TT = CBN + 1;                       // row stride of the mirrored PW tile staged for the epilogue
constexpr int CG_EPI_BYTES = CBM * TT * (int)sizeof(float2) + (2 * CBM + 16) * (int)sizeof(double);
constexpr int CG_SMEM = CG_MAIN_BYTES > CG_EPI_BYTES ? CG_MAIN_BYTES : CG_EPI_BYTES;
constexpr int STT = SBN + 1;
constexpr int SG_MAIN_BYTES = 2 * (SA_BYTES + SB_BYTES);
constexpr int SG_EPI_BYTES = SBM * STT * (int)sizeof(float2) + (2 * SBM + 16) * (int)sizeof(double);
constexpr int SG_SMEM = SG_MAIN_BYTES > SG_EPI_BYTES ? SG_MAIN_BYTES : SG_EPI_BYTES;
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

// C = A @ B (EPI = false) or the second product with its fused epilogue (EPI = true):
//   dW_new = A @ B + (PW - PW^H);  Whalf = W + dW_new;  rowpart[tile column][i] = sum_j |dW_old - dW_new|
// 64 x 64 block tile, 4 wavefronts (2 x 2), one 32 x 32 MFMA tile each; K-tiles of 16 complex entries double-buffered
// in LDS, two K-tiles in flight global -> registers.  3M: T1 = sum ar br, T2 = sum ai bi, T3 = sum (ar+ai)(br+bi).
// MFMA f32 32x32x2 lane maps (cdna_hip_programming.md section 3): A[i = lane & 31][k = lane >> 5],
// B[k = lane >> 5][j = lane & 31], C/D[row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)][col = lane & 31].
template <bool EPI, bool EXACT>
__global__ __launch_bounds__(256) void k_cgemm(int N, int tiles_n, const float2 *__restrict__ A, const float2 *__restrict__ B,
                                               float2 *__restrict__ C, qf_epilogue_f ep, qf_guard guard)
{
    if (!qf_guard_iter(guard)) return;
    // fused step end: the first product of a step's first iteration takes the Whalf prepared for it
    if (!EPI && guard.alt && guard.state->wh_sel) B = static_cast<const float2 *>(guard.alt);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int i0 = tm * CBM, j0 = tn * CBN;
    const int parity = (EPI && guard.state) ? guard.state->dw_parity : 0;
    const float2 *__restrict__ dW_old = ep.dW[parity];
    float2 *__restrict__ dW_new = ep.dW[parity ^ 1];
    // fused step end: the state is Wpair[w_parity]; the candidate next state goes to the other buffer
    const int wpar = (EPI && ep.fused && guard.state) ? guard.state->w_parity : 0;
    const float2 *__restrict__ ep_W = (EPI && ep.fused) ? ep.Wpair[wpar] : ep.W;
    float2 *__restrict__ ep_Wnext = (EPI && ep.fused) ? ep.Wpair[wpar ^ 1] : nullptr;

    // staging maps: A rows (tid / 8) and + 32, k-pair tid % 8;  B k-rows (tid / 32) and + 8, column pair tid % 32
    const int a_row = tid >> 3, a_kp = tid & 7;
    const int b_k = tid >> 5, b_jp = tid & 31;
    float4 ra[2][2], rb[2][2];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    auto load_tile = [&](int kt, float4 (&a)[2], float4 (&b)[2]) {
        const int k0 = kt * CBK;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int gi = i0 + a_row + 32 * r, gk = k0 + 2 * a_kp;
            a[r] = zero4;
            if (EXACT || (gi < N && gk + 1 < N)) a[r] = ld4<EXACT>(A + (size_t)gi * N + gk);
            else if (gi < N && gk < N) { const float2 t = A[(size_t)gi * N + gk]; a[r] = make_float4(t.x, t.y, 0.f, 0.f); }
            const int gkb = k0 + b_k + 8 * r, gj = j0 + 2 * b_jp;
            b[r] = zero4;
            if (EXACT || (gkb < N && gj + 1 < N)) b[r] = ld4<EXACT>(B + (size_t)gkb * N + gj);
            else if (gkb < N && gj < N) { const float2 t = B[(size_t)gkb * N + gj]; b[r] = make_float4(t.x, t.y, 0.f, 0.f); }
        }
    };
    auto store_tile = [&](int buf, const float4 (&a)[2], const float4 (&b)[2]) {
        float2 *As = reinterpret_cast<float2 *>(smem + buf * A_BYTES);
        float2 *Bs = reinterpret_cast<float2 *>(smem + 2 * A_BYTES + buf * B_BYTES);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            As[(2 * a_kp) * SA + a_row + 32 * r] = make_float2(a[r].x, a[r].y);
            As[(2 * a_kp + 1) * SA + a_row + 32 * r] = make_float2(a[r].z, a[r].w);
            *reinterpret_cast<float4 *>(Bs + (b_k + 8 * r) * SB + 2 * b_jp) = b[r];
        }
    };

    v16f t1, t2, t3;
#pragma unroll
    for (int q = 0; q < 16; ++q) { t1[q] = 0.f; t2[q] = 0.f; t3[q] = 0.f; }

    auto compute = [&](int buf) {
        const float2 *As = reinterpret_cast<const float2 *>(smem + buf * A_BYTES) + wm * 32 + l31;
        const float2 *Bs = reinterpret_cast<const float2 *>(smem + 2 * A_BYTES + buf * B_BYTES) + wn * 32 + l31;
#pragma unroll
        for (int s = 0; s < CBK / 2; ++s) {
            const float2 a = As[(2 * s + lh) * SA];
            const float2 b = Bs[(2 * s + lh) * SB];
            t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, t2, 0, 0, 0);
            t3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x + a.y, b.x + b.y, t3, 0, 0, 0);
        }
    };

    const int KT = (N + CBK - 1) / CBK;
    load_tile(0, ra[0], rb[0]);
    if (KT > 1) load_tile(1, ra[1], rb[1]);
    store_tile(0, ra[0], rb[0]);
    __syncthreads();
    if (KT > 2) load_tile(2, ra[0], rb[0]);
    int kt = 0;
    // two K-tiles per trip: LDS buffer and register-set indices are literals.  (Staging K-tile kt+1 and issuing
    // the loads of kt+3 AHEAD of K-tile kt's MFMAs instead of behind them measured slower: 81.9 against 78.5 us
    // at N=1024, 456 against 437 at N=2048 -- the waves then meet the LDS store path all at once.)
    for (; kt + 1 < KT; kt += 2) {
        compute(0);
        store_tile(1, ra[1], rb[1]);                       // K-tile kt+1 (arrived two K-tiles ago)
        if (kt + 3 < KT) load_tile(kt + 3, ra[1], rb[1]);
        __syncthreads();
        compute(1);
        if (kt + 2 < KT) {
            store_tile(0, ra[0], rb[0]);
            if (kt + 4 < KT) load_tile(kt + 4, ra[0], rb[0]);
        }
        __syncthreads();
    }
    if (kt < KT) compute(0);

    if constexpr (!EPI) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int gi = i0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            const int gj = j0 + wn * 32 + l31;
            if (EXACT || (gi < N && gj < N)) C[(size_t)gi * N + gj] = make_float2(t1[q] - t2[q], (t3[q] - t1[q]) - t2[q]);
        }
    } else {
        // ---- fused epilogue.  The mirrored PW tile (rows j0.., columns i0..) goes through LDS so that both
        // global reads of PW are row-coalesced; conj_subtract_ (isospectral.py:71-74): PW[i,j] - conj(PW[j,i]).
        __syncthreads();      // every wave is done with the K-loop buffers
        float2 *Tt = reinterpret_cast<float2 *>(smem);
        double *rs = reinterpret_cast<double *>(smem + CBM * TT * sizeof(float2));   // [2][CBM]
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = (tid >> 5) + 8 * r, cp = tid & 31;
            const int gj = j0 + row, gi = i0 + 2 * cp;
            float4 v = zero4;
            if (EXACT || (gj < N && gi + 1 < N)) v = ld4<EXACT>(ep.PW + (size_t)gj * N + gi);
            else if (gj < N && gi < N) { const float2 t = ep.PW[(size_t)gj * N + gi]; v = make_float4(t.x, t.y, 0.f, 0.f); }
            Tt[row * TT + 2 * cp] = make_float2(v.x, v.y);
            Tt[row * TT + 2 * cp + 1] = make_float2(v.z, v.w);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int li = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            const int lj = wn * 32 + l31;
            const int gi = i0 + li, gj = j0 + lj;
            float a = 0.f;
            if (EXACT || (gi < N && gj < N)) {
                const size_t e = (size_t)gi * N + gj;
                const float2 pw = ep.PW[e];
                const float2 pwt = Tt[lj * TT + li];
                const float cr = pw.x - pwt.x, ci = pw.y + pwt.y;
                const float dr = (t1[q] - t2[q]) + cr;                       // dW = (PW @ Phalf) + comm   (:499,509)
                const float di = ((t3[q] - t1[q]) - t2[q]) + ci;
                dW_new[e] = make_float2(dr, di);
                const float2 w = ep_W[e];
                ep.Whalf[e] = make_float2(w.x + dr, w.y + di);               // Whalf = W + dW             (:481-482)
                if (ep.fused) {
                    // should this be the step's last iteration: W_next = W + 2 comm (:547,592), next Whalf = W_next + dW
                    const float wr = w.x + 2.0f * cr, wi = w.y + 2.0f * ci;
                    ep_Wnext[e] = make_float2(wr, wi);
                    ep.Whalf_step[e] = make_float2(wr + dr, wi + di);
                }
                const float2 o = dW_old[e];
                const float er = o.x - dr, ei = o.y - di;                    // |dW_old - dW|              (:526,534)
                a = sqrtf(er * er + ei * ei);
            }
            // sum over the 32 lanes that share this row (fixed butterfly: deterministic), in double
            double rsum = (double)a;
            rsum = qf_row16_sum(rsum);      // (xor butterfly 1, 2, 4, 8 on DPP: same tree, same bits as four __shfl_xor steps)
            rsum += __shfl_xor(rsum, 16, 64);
            if (l31 == 0) rs[wn * CBM + li] = rsum;
        }
        __syncthreads();
        if (tid < CBM && (EXACT || i0 + tid < N)) {
            const double v = rs[tid] + rs[CBM + tid];
            if (ep.fused) __hip_atomic_store(ep.rowpart + (size_t)tn * N + i0 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ep.rowpart[(size_t)tn * N + i0 + tid] = v;
        }
        if (ep.fused) {
            // the last tile to get here closes the iteration (ticket: guide section 6 G16, counter form)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned *last_flag = reinterpret_cast<unsigned *>(rs + 2 * CBM);
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(ep.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *last_flag = (old == (unsigned)(ep.n_tiles - 1)) ? 1u : 0u;
            }
            __syncthreads();
            if (*last_flag != 0u)
                qf_fused_step_end<1>(N, tiles_n, ep.rowpart, ep.ticket, ep.state_rw, ep.rec, guard.iter, tid, rs + 2 * CBM + 2);
        }
    }
}

// The plain product with the K range of a tile cut into KS parts INSIDE the workgroup: KS groups of four wavefronts, each
// with its own LDS buffers and its own quarter (half) of the K-tiles, partial tiles added in group order at the end.
// Why: at N = 1024 a launch is one 64 x 64 tile per CU, i.e. one wavefront per SIMD, and the fp32 matrix pipe idles
// while that wavefront waits for its LDS reads and barriers -- the same kernel reaches 0.81 of the peak at N = 2048,
// where four workgroups share a CU, against 0.59 at N = 1024.  More wavefronts per SIMD on the SAME tile give the
// scheduler that choice without a second tile.  (N % (16 KS) == 0: every group runs the same number of barriers.)
template <int KS>
__global__ __launch_bounds__(256 * KS) void k_cgemm_ks(int N, int tiles_n, const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                        float2 *__restrict__ C, qf_guard guard)
{
    if (!qf_guard_iter(guard)) return;
    if (guard.alt && guard.state->wh_sel) B = static_cast<const float2 *>(guard.alt);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
    unsigned char *smem = smem_all + grp * CG_MAIN_BYTES;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int i0 = tm * CBM, j0 = tn * CBN;
    const int a_row = tid >> 3, a_kp = tid & 7;
    const int b_k = tid >> 5, b_jp = tid & 31;
    float4 ra[2][2], rb[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) ra[x][y] = rb[x][y] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int KTp = (N / CBK) / KS, kb = grp * KTp;

    auto load_tile = [&](int kt, float4 (&a)[2], float4 (&b)[2]) __attribute__((always_inline)) {
        const int k0 = (kb + kt) * CBK;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            a[r] = *reinterpret_cast<const float4 *>(A + (size_t)(i0 + a_row + 32 * r) * N + k0 + 2 * a_kp);
            b[r] = *reinterpret_cast<const float4 *>(B + (size_t)(k0 + b_k + 8 * r) * N + j0 + 2 * b_jp);
        }
    };
    auto store_tile = [&](int buf, const float4 (&a)[2], const float4 (&b)[2]) __attribute__((always_inline)) {
        float2 *As = reinterpret_cast<float2 *>(smem + buf * A_BYTES);
        float2 *Bs = reinterpret_cast<float2 *>(smem + 2 * A_BYTES + buf * B_BYTES);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            As[(2 * a_kp) * SA + a_row + 32 * r] = make_float2(a[r].x, a[r].y);
            As[(2 * a_kp + 1) * SA + a_row + 32 * r] = make_float2(a[r].z, a[r].w);
            *reinterpret_cast<float4 *>(Bs + (b_k + 8 * r) * SB + 2 * b_jp) = b[r];
        }
    };
    v16f t1, t2, t3;
#pragma unroll
    for (int q = 0; q < 16; ++q) { t1[q] = 0.f; t2[q] = 0.f; t3[q] = 0.f; }
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float2 *As = reinterpret_cast<const float2 *>(smem + buf * A_BYTES) + wm * 32 + l31;
        const float2 *Bs = reinterpret_cast<const float2 *>(smem + 2 * A_BYTES + buf * B_BYTES) + wn * 32 + l31;
#pragma unroll
        for (int s = 0; s < CBK / 2; ++s) {
            const float2 a = As[(2 * s + lh) * SA];
            const float2 b = Bs[(2 * s + lh) * SB];
            t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, t2, 0, 0, 0);
            t3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x + a.y, b.x + b.y, t3, 0, 0, 0);
        }
    };
    load_tile(0, ra[0], rb[0]);
    if (KTp > 1) load_tile(1, ra[1], rb[1]);
    store_tile(0, ra[0], rb[0]);
    __syncthreads();
    if (KTp > 2) load_tile(2, ra[0], rb[0]);
    int kt = 0;
    for (; kt + 1 < KTp; kt += 2) {
        compute(0);
        store_tile(1, ra[1], rb[1]);
        if (kt + 3 < KTp) load_tile(kt + 3, ra[1], rb[1]);
        __syncthreads();
        compute(1);
        if (kt + 2 < KTp) {
            store_tile(0, ra[0], rb[0]);
            if (kt + 4 < KTp) load_tile(kt + 4, ra[0], rb[0]);
        }
        __syncthreads();
    }
    if (kt < KTp) compute(0);
    // the groups' partial tiles through LDS ([group - 1][q][thread]: conflict-free), added in group order
    __syncthreads();
    float2 *X = reinterpret_cast<float2 *>(smem_all);
    if (grp > 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) X[((grp - 1) * 16 + q) * 256 + tid] = make_float2(t1[q] - t2[q], (t3[q] - t1[q]) - t2[q]);
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float re = t1[q] - t2[q], im = (t3[q] - t1[q]) - t2[q];
#pragma unroll
        for (int g = 1; g < KS; ++g) {
            const float2 v = X[((g - 1) * 16 + q) * 256 + tid];
            re += v.x;
            im += v.y;
        }
        const int gi = i0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
        const int gj = j0 + wn * 32 + l31;
        C[(size_t)gi * N + gj] = make_float2(re, im);
    }
}

// ---- the second product on the upper triangle (complex64) ----
// With Phalf and Whalf skew-Hermitian, dW = PW @ Phalf + (PW - PW^H) is skew-Hermitian: only the nt (nt + 1) / 2 tiles
// on and above the diagonal are multiplied, and the finishing workgroup writes the tile of Whalf AND its mirror image
// -conj(.)^T (DESIGN.md 3.1b / 3.1c for the double-precision kernels).  At N = 1024 that is 136 tiles for 256 CUs, so
// the K range of every tile is cut into `split` pieces, one workgroup each (k_zgemm_tri32's exchange): a workgroup
// parks its partial tile (write-through), drains, takes a ticket on the tile's arrival counter and leaves unless it
// came last; the last arrival adds ALL pieces from memory in piece order (its own included: the same bits whoever is
// last) and runs the epilogue.  Nobody waits; the counters are monotone (`split` arrivals per executed launch).
// 67 KiB of LDS: two workgroups share a CU, which is what lets 272 workgroups run on 256 CUs at once.
// Below the diagonal only Whalf is written (the next first product's right operand): W and dW are read back on and
// above the diagonal tiles only and restored once at the end of a call (k_mirror_lower_f).
// On 32 x 32 tiles (any N >= 64; edge tiles guarded): k_cgemm32's K loop (one 16 x 16 MFMA tile per wavefront).  Partial tiles
// are 8 KiB, the epilogue's LDS 17 KiB: many workgroups share a CU.  (A 64 x 64-tile form of this kernel existed in
// rounds 3-4: slower at every size -- N = 1024 71.6 against 59.2 us, N = 2048 320 against 299 -- removed in round 5.)
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
constexpr int ST_TILE_BYTES = SBM * STT * (int)sizeof(float2);
constexpr int ST_EPI_BYTES = 2 * ST_TILE_BYTES + (4 * SBM + 16) * (int)sizeof(double);
constexpr int ST_SMEM = SG_MAIN_BYTES > ST_EPI_BYTES ? SG_MAIN_BYTES : ST_EPI_BYTES;

// EXACT = false: N is no multiple of 32 -- nt = ceil(N / 32) tile rows, loads, stores and row sums of the edge tiles guarded
// (out-of-range operands are zeros: they add nothing to the products), K-tiles = ceil(N / 16).
template <bool EXACT>
__global__ __launch_bounds__(256) void k_cgemm_tri32(int N, int nt, const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                     qf_epilogue_f ep, qf_guard guard, qf_ctri sx)
{
    if (!qf_guard_iter(guard)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int nd_pieces = nt * sx.split_diag, noff = nt * (nt - 1) / 2;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    int tm, tn, h, S, t;
    if (lid < nd_pieces) {
        tm = tn = lid / sx.split_diag;
        h = lid % sx.split_diag;
        S = sx.split_diag;
        t = tm;
    } else {
        const int o2 = lid - nd_pieces;
        int o = o2 % noff;
        h = o2 / noff;
        S = sx.split;
        t = nt + o;
        tm = 0;
        int rowlen = nt - 1;
        while (o >= rowlen) {
            o -= rowlen;
            ++tm;
            --rowlen;
        }
        tn = tm + 1 + o;
    }
    const int i0 = tm * SBM, j0 = tn * SBN;
    const bool offdiag = (tm != tn);
    const int parity = guard.state ? guard.state->dw_parity : 0;
    const float2 *__restrict__ dW_old = ep.dW[parity];
    float2 *__restrict__ dW_new = ep.dW[parity ^ 1];
    const int wpar = (ep.fused && guard.state) ? guard.state->w_parity : 0;
    const float2 *__restrict__ ep_W = ep.fused ? ep.Wpair[wpar] : ep.W;
    float2 *__restrict__ ep_Wnext = ep.fused ? ep.Wpair[wpar ^ 1] : nullptr;

    const int a_row = tid >> 3, a_kp = tid & 7;
    const int b_k = tid >> 4, b_jp = tid & 15;
    float4 ra[2], rb[2];
    ra[0] = ra[1] = rb[0] = rb[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    // this piece's K-tiles: kb .. kb + KTp - 1 (N % 32 == 0 makes the K-tile count even, not a multiple of 4: the pieces
    // of a tile may differ by one K-tile)
    const int KTN = (N + CBK - 1) / CBK;
    const int kb = (int)((long long)KTN * h / S), KTp = (int)((long long)KTN * (h + 1) / S) - kb;

    auto load_tile = [&](int kt, float4 &a, float4 &b) __attribute__((always_inline)) {
        const int k0 = (kb + kt) * CBK;
        if (EXACT) {
            a = *reinterpret_cast<const float4 *>(A + (size_t)(i0 + a_row) * N + k0 + 2 * a_kp);
            b = *reinterpret_cast<const float4 *>(B + (size_t)(k0 + b_k) * N + j0 + 2 * b_jp);
        } else {
            const int gi = i0 + a_row, gk = k0 + 2 * a_kp;
            a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gi < N && gk + 1 < N) a = ld4<false>(A + (size_t)gi * N + gk);
            else if (gi < N && gk < N) { const float2 t = A[(size_t)gi * N + gk]; a = make_float4(t.x, t.y, 0.f, 0.f); }
            const int gkb = k0 + b_k, gj = j0 + 2 * b_jp;
            b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gkb < N && gj + 1 < N) b = ld4<false>(B + (size_t)gkb * N + gj);
            else if (gkb < N && gj < N) { const float2 t = B[(size_t)gkb * N + gj]; b = make_float4(t.x, t.y, 0.f, 0.f); }
        }
    };
    auto store_tile = [&](int buf, const float4 &a, const float4 &b) __attribute__((always_inline)) {
        float2 *As = reinterpret_cast<float2 *>(smem + buf * SA_BYTES);
        float2 *Bs = reinterpret_cast<float2 *>(smem + 2 * SA_BYTES + buf * SB_BYTES);
        As[a_row * SAK + 2 * a_kp] = make_float2(a.x, a.y);
        As[a_row * SAK + 2 * a_kp + 1] = make_float2(a.z, a.w);
        *reinterpret_cast<float4 *>(Bs + b_k * SBN + 2 * b_jp) = b;
    };
    v4f t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f}, t3 = {0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float2 *As = reinterpret_cast<const float2 *>(smem + buf * SA_BYTES) + (wm * 16 + l15) * SAK + lq;
        const float2 *Bs = reinterpret_cast<const float2 *>(smem + 2 * SA_BYTES + buf * SB_BYTES) + lq * SBN + wn * 16 + l15;
#pragma unroll
        for (int s = 0; s < CBK / 4; ++s) {
            const float2 a = As[4 * s];
            const float2 b = Bs[4 * s * SBN];
            t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, t2, 0, 0, 0);
            t3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x + a.y, b.x + b.y, t3, 0, 0, 0);
        }
    };
    load_tile(0, ra[0], rb[0]);
    if (KTp > 1) load_tile(1, ra[1], rb[1]);
    store_tile(0, ra[0], rb[0]);
    __syncthreads();
    if (KTp > 2) load_tile(2, ra[0], rb[0]);
    int kt = 0;
    for (; kt + 1 < KTp; kt += 2) {
        compute(0);
        store_tile(1, ra[1], rb[1]);
        if (kt + 3 < KTp) load_tile(kt + 3, ra[1], rb[1]);
        __syncthreads();
        compute(1);
        if (kt + 2 < KTp) {
            store_tile(0, ra[0], rb[0]);
            if (kt + 4 < KTp) load_tile(kt + 4, ra[0], rb[0]);
        }
        __syncthreads();
    }
    if (kt < KTp) compute(0);

    float re[4], im[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        re[q] = t1[q] - t2[q];
        im[q] = (t3[q] - t1[q]) - t2[q];
    }
    unsigned *flagw = reinterpret_cast<unsigned *>(smem + 2 * ST_TILE_BYTES + 4 * SBM * sizeof(double));
    if (S > 1) {
        const __amdgpu_buffer_rsrc_t rsrcP = __builtin_amdgcn_make_buffer_rsrc(sx.partial, 0, 0x7fffffff, 0x00020000);
        const unsigned p_voff = (unsigned)(tid * sizeof(float2));
        const unsigned slot_bytes = (unsigned)(SBM * SBN * sizeof(float2));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 v = make_float2(re[q], im[q]);
            __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const v2u_t *>(&v), rsrcP, p_voff + (unsigned)(q * 256 * sizeof(float2)),
                                                  (unsigned)(4 * t + h) * slot_bytes, 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(sx.arrive + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flagw = ((old % (unsigned)S) == (unsigned)(S - 1)) ? 1u : 0u;
        }
        __syncthreads();
        if (*flagw == 0u) return;
#pragma unroll 1
        for (int hh = 0; hh < S; ++hh) {
            float2 v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v2u_t raw = __builtin_amdgcn_raw_buffer_load_b64(rsrcP, p_voff + (unsigned)(q * 256 * sizeof(float2)),
                                                                       (unsigned)(4 * t + hh) * slot_bytes, 16);
                v[q] = *reinterpret_cast<const float2 *>(&raw);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                re[q] = hh == 0 ? v[q].x : re[q] + v[q].x;
                im[q] = hh == 0 ? v[q].y : im[q] + v[q].y;
            }
        }
    }

    __syncthreads();
    float2 *Tt = reinterpret_cast<float2 *>(smem);
    float2 *Th = reinterpret_cast<float2 *>(smem + ST_TILE_BYTES);
    double *rs = reinterpret_cast<double *>(smem + 2 * ST_TILE_BYTES);   // [2][32] row sums
    double *cs = rs + 2 * SBM;                                           // [2][32] column sums
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int row = (tid >> 4) + 16 * r, cp = tid & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int gj = j0 + row, gi = i0 + 2 * cp;
        if (EXACT || (gj < N && gi + 1 < N)) v = ld4<EXACT>(ep.PW + (size_t)gj * N + gi);
        else if (gj < N && gi < N) { const float2 t = ep.PW[(size_t)gj * N + gi]; v = make_float4(t.x, t.y, 0.f, 0.f); }
        Tt[row * STT + 2 * cp] = make_float2(v.x, v.y);
        Tt[row * STT + 2 * cp + 1] = make_float2(v.z, v.w);
    }
    __syncthreads();
    float2 dv[4], whv[4], wnv[4], whs[4];
    double csum = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int li = wm * 16 + 4 * lq + q;
        const int lj = wn * 16 + l15;
        const size_t e = (size_t)(i0 + li) * N + (j0 + lj);
        const bool in = EXACT || (i0 + li < N && j0 + lj < N);
        double a = 0.0;
        dv[q] = whv[q] = wnv[q] = whs[q] = make_float2(0.f, 0.f);
        if (in) {
            const float2 pw = ep.PW[e];
            const float2 pwt = Tt[lj * STT + li];
            const float cr = pw.x - pwt.x, ci = pw.y + pwt.y;
            const float dr = re[q] + cr;
            const float di = im[q] + ci;
            dv[q] = make_float2(dr, di);
            const float2 w = ep_W[e];
            whv[q] = make_float2(w.x + dr, w.y + di);
            const float wr = w.x + 2.0f * cr, wi = w.y + 2.0f * ci;
            wnv[q] = make_float2(wr, wi);
            whs[q] = make_float2(wr + dr, wi + di);
            const float2 o = dW_old[e];
            const float er = o.x - dr, ei = o.y - di;
            a = (double)sqrtf(er * er + ei * ei);
        }
        csum += a;
        double rsum = a;
        rsum = qf_row16_sum(rsum);      // (xor butterfly 1, 2, 4, 8 on DPP: same tree, same bits as four __shfl_xor steps)
        if (l15 == 0) rs[wn * SBM + li] = rsum;
    }
    csum += __shfl_xor(csum, 16, 64);     // the four lane groups hold rows 4 lq .. 4 lq + 3 of this column
    csum += __shfl_xor(csum, 32, 64);
    if (lq == 0) cs[wm * SBN + wn * 16 + l15] = csum;
    __syncthreads();
    if (tid < SBM) {
        if (EXACT || i0 + tid < N)
            __hip_atomic_store(ep.rowpart + (size_t)tn * N + i0 + tid, rs[tid] + rs[SBM + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (offdiag && tid >= 64 && tid < 64 + SBN) {
        const int lj = tid - 64;
        if (EXACT || j0 + lj < N)
            __hip_atomic_store(ep.rowpart + (size_t)tm * N + j0 + lj, cs[lj] + cs[SBN + lj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned ticket_old = 0u;
    if (ep.fused) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) ticket_old = __hip_atomic_fetch_add(ep.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int li = wm * 16 + 4 * lq + q;
        const int lj = wn * 16 + l15;
        const size_t e = (size_t)(i0 + li) * N + (j0 + lj);
        const bool in = EXACT || (i0 + li < N && j0 + lj < N);
        if (in) {
            dW_new[e] = dv[q];
            ep.Whalf[e] = whv[q];
        }
        Th[li * STT + lj] = whv[q];
        if (ep.fused) {
            if (in) {
                ep_Wnext[e] = wnv[q];
                ep.Whalf_step[e] = whs[q];
            }
            Tt[li * STT + lj] = whs[q];
        }
    }
    __syncthreads();
    if (offdiag) {
        // row j0 + jl of the mirrored tile is column jl of this one (32 entries = 256 bytes): two per wave instruction
        const int il = lane & 31;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int jl = wave * 8 + r * 2 + (lane >> 5);
            if (!EXACT && (j0 + jl >= N || i0 + il >= N)) continue;
            const float2 wv = Th[il * STT + jl];
            ep.Whalf[(size_t)(j0 + jl) * N + i0 + il] = make_float2(-wv.x, wv.y);
            if (ep.fused) {
                const float2 ws = Tt[il * STT + jl];
                ep.Whalf_step[(size_t)(j0 + jl) * N + i0 + il] = make_float2(-ws.x, ws.y);
            }
        }
    }
    if (ep.fused) {
        __syncthreads();
        unsigned *last_flag = reinterpret_cast<unsigned *>(rs);
        if (tid == 0) *last_flag = (ticket_old == (unsigned)(ep.n_tiles - 1)) ? 1u : 0u;
        __syncthreads();
        if (*last_flag != 0u) qf_fused_step_end<1>(N, nt, ep.rowpart, ep.ticket, ep.state_rw, ep.rec, guard.iter, tid, rs + 2);
    }
}

// X[j,i] = -conj(X[i,j]) for i < j: the lower triangle of a skew-Hermitian complex64 matrix from its upper one
__global__ __launch_bounds__(256) void k_mirror_lower_f(int N, float2 *__restrict__ X)
{
    constexpr int MT = 32;
    __shared__ float2 Ts[MT][MT + 1];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int i0 = bi * MT, j0 = bj * MT;
    for (int r = ty; r < MT; r += 8) {
        const int gj = j0 + r, gi = i0 + tx;
        float2 tv = make_float2(0.f, 0.f);
        if (gj < N && gi < N) tv = X[(size_t)gj * N + gi];
        Ts[r][tx] = tv;
    }
    __syncthreads();
    for (int r = ty; r < MT; r += 8) {
        const int gi = i0 + r, gj = j0 + tx;
        if (gi < N && gj < N && gj < gi) {
            const float2 t = Ts[tx][r];
            X[(size_t)gi * N + gj] = make_float2(-t.x, t.y);
        }
    }
}

// The same product on 32 x 32 block tiles (N < 768): 4 wavefronts (2 x 2), one 16 x 16 tile each.
// MFMA f32 16x16x4 lane maps: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// C/D[row = 4 (lane >> 4) + reg][col = lane & 15].
// KS > 1 (plain product, exact tilings): KS groups of four wavefronts on the same tile, each with its own LDS buffers and
// its own 1/KS of the K-tiles (see k_cgemm_ks).
template <bool EPI, bool EXACT, int KS = 1>
__global__ __launch_bounds__(256 * KS) void k_cgemm32(int N, int tiles_n, const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                      float2 *__restrict__ C, qf_epilogue_f ep, qf_guard guard)
{
    if (!qf_guard_iter(guard)) return;
    if (!EPI && guard.alt && guard.state->wh_sel) B = static_cast<const float2 *>(guard.alt);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int grp = KS > 1 ? (int)(threadIdx.x >> 8) : 0;
    unsigned char *const smem = KS > 1 ? smem_all + grp * SG_MAIN_BYTES : smem_all;
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = lid / tiles_n, tn = lid % tiles_n;
    const int i0 = tm * SBM, j0 = tn * SBN;
    const int parity = (EPI && guard.state) ? guard.state->dw_parity : 0;
    const float2 *__restrict__ dW_old = ep.dW[parity];
    float2 *__restrict__ dW_new = ep.dW[parity ^ 1];
    // fused step end: the state is Wpair[w_parity]; the candidate next state goes to the other buffer
    const int wpar = (EPI && ep.fused && guard.state) ? guard.state->w_parity : 0;
    const float2 *__restrict__ ep_W = (EPI && ep.fused) ? ep.Wpair[wpar] : ep.W;
    float2 *__restrict__ ep_Wnext = (EPI && ep.fused) ? ep.Wpair[wpar ^ 1] : nullptr;

    // staging maps: A row tid / 8, k-pair tid % 8;  B k-row tid / 16, column pair tid % 16
    const int a_row = tid >> 3, a_kp = tid & 7;
    const int b_k = tid >> 4, b_jp = tid & 15;
    float4 ra[2], rb[2];
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    const int KT = ((N + CBK - 1) / CBK) / KS, kb = grp * KT;      // this group's K-tiles
    auto load_tile = [&](int kt, float4 &a, float4 &b) {
        const int k0 = (kb + kt) * CBK;
        const int gi = i0 + a_row, gk = k0 + 2 * a_kp;
        a = zero4;
        if (EXACT || (gi < N && gk + 1 < N)) a = ld4<EXACT>(A + (size_t)gi * N + gk);
        else if (gi < N && gk < N) { const float2 t = A[(size_t)gi * N + gk]; a = make_float4(t.x, t.y, 0.f, 0.f); }
        const int gkb = k0 + b_k, gj = j0 + 2 * b_jp;
        b = zero4;
        if (EXACT || (gkb < N && gj + 1 < N)) b = ld4<EXACT>(B + (size_t)gkb * N + gj);
        else if (gkb < N && gj < N) { const float2 t = B[(size_t)gkb * N + gj]; b = make_float4(t.x, t.y, 0.f, 0.f); }
    };
    auto store_tile = [&](int buf, const float4 &a, const float4 &b) {
        float2 *As = reinterpret_cast<float2 *>(smem + buf * SA_BYTES);
        float2 *Bs = reinterpret_cast<float2 *>(smem + 2 * SA_BYTES + buf * SB_BYTES);
        As[a_row * SAK + 2 * a_kp] = make_float2(a.x, a.y);
        As[a_row * SAK + 2 * a_kp + 1] = make_float2(a.z, a.w);
        *reinterpret_cast<float4 *>(Bs + b_k * SBN + 2 * b_jp) = b;
    };

    v4f t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f}, t3 = {0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const float2 *As = reinterpret_cast<const float2 *>(smem + buf * SA_BYTES) + (wm * 16 + l15) * SAK + lq;
        const float2 *Bs = reinterpret_cast<const float2 *>(smem + 2 * SA_BYTES + buf * SB_BYTES) + lq * SBN + wn * 16 + l15;
#pragma unroll
        for (int s = 0; s < CBK / 4; ++s) {
            const float2 a = As[4 * s];
            const float2 b = Bs[4 * s * SBN];
            t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, t2, 0, 0, 0);
            t3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x + a.y, b.x + b.y, t3, 0, 0, 0);
        }
    };

    load_tile(0, ra[0], rb[0]);
    if (KT > 1) load_tile(1, ra[1], rb[1]);
    store_tile(0, ra[0], rb[0]);
    __syncthreads();
    if (KT > 2) load_tile(2, ra[0], rb[0]);
    int kt = 0;
    for (; kt + 1 < KT; kt += 2) {
        compute(0);
        store_tile(1, ra[1], rb[1]);
        if (kt + 3 < KT) load_tile(kt + 3, ra[1], rb[1]);
        __syncthreads();
        compute(1);
        if (kt + 2 < KT) {
            store_tile(0, ra[0], rb[0]);
            if (kt + 4 < KT) load_tile(kt + 4, ra[0], rb[0]);
        }
        __syncthreads();
    }
    if (kt < KT) compute(0);

    if constexpr (!EPI) {
        float2 *X = reinterpret_cast<float2 *>(smem_all);       // KS > 1: [group - 1][q][thread]
        if constexpr (KS > 1) {
            __syncthreads();
            if (grp > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) X[((grp - 1) * 4 + q) * 256 + tid] = make_float2(t1[q] - t2[q], (t3[q] - t1[q]) - t2[q]);
            }
            __syncthreads();
            if (grp > 0) return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int gi = i0 + wm * 16 + 4 * lq + q;
            const int gj = j0 + wn * 16 + l15;
            float re = t1[q] - t2[q], im = (t3[q] - t1[q]) - t2[q];
            if constexpr (KS > 1) {
#pragma unroll
                for (int g = 1; g < KS; ++g) {
                    const float2 v = X[((g - 1) * 4 + q) * 256 + tid];
                    re += v.x;
                    im += v.y;
                }
            }
            if (EXACT || (gi < N && gj < N)) C[(size_t)gi * N + gj] = make_float2(re, im);
        }
    } else {
        __syncthreads();
        float2 *Tt = reinterpret_cast<float2 *>(smem);
        double *rs = reinterpret_cast<double *>(smem + SBM * STT * sizeof(float2));   // [2][SBM]
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = (tid >> 4) + 16 * r, cp = tid & 15;
            const int gj = j0 + row, gi = i0 + 2 * cp;
            float4 v = zero4;
            if (EXACT || (gj < N && gi + 1 < N)) v = ld4<EXACT>(ep.PW + (size_t)gj * N + gi);
            else if (gj < N && gi < N) { const float2 t = ep.PW[(size_t)gj * N + gi]; v = make_float4(t.x, t.y, 0.f, 0.f); }
            Tt[row * STT + 2 * cp] = make_float2(v.x, v.y);
            Tt[row * STT + 2 * cp + 1] = make_float2(v.z, v.w);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int li = wm * 16 + 4 * lq + q;
            const int lj = wn * 16 + l15;
            const int gi = i0 + li, gj = j0 + lj;
            float a = 0.f;
            if (EXACT || (gi < N && gj < N)) {
                const size_t e = (size_t)gi * N + gj;
                const float2 pw = ep.PW[e];
                const float2 pwt = Tt[lj * STT + li];
                const float cr = pw.x - pwt.x, ci = pw.y + pwt.y;
                const float dr = (t1[q] - t2[q]) + cr;
                const float di = ((t3[q] - t1[q]) - t2[q]) + ci;
                dW_new[e] = make_float2(dr, di);
                const float2 w = ep_W[e];
                ep.Whalf[e] = make_float2(w.x + dr, w.y + di);
                if (ep.fused) {
                    const float wr = w.x + 2.0f * cr, wi = w.y + 2.0f * ci;
                    ep_Wnext[e] = make_float2(wr, wi);
                    ep.Whalf_step[e] = make_float2(wr + dr, wi + di);
                }
                const float2 o = dW_old[e];
                const float er = o.x - dr, ei = o.y - di;
                a = sqrtf(er * er + ei * ei);
            }
            double rsum = (double)a;
            rsum = qf_row16_sum(rsum);      // (xor butterfly 1, 2, 4, 8 on DPP: same tree, same bits as four __shfl_xor steps)
            if (l15 == 0) rs[wn * SBM + li] = rsum;
        }
        __syncthreads();
        if (tid < SBM && (EXACT || i0 + tid < N)) {
            const double v = rs[tid] + rs[SBM + tid];
            if (ep.fused) __hip_atomic_store(ep.rowpart + (size_t)tn * N + i0 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else ep.rowpart[(size_t)tn * N + i0 + tid] = v;
        }
        if (ep.fused) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned *last_flag = reinterpret_cast<unsigned *>(rs + 2 * SBM);
            if (tid == 0) {
                const unsigned old = __hip_atomic_fetch_add(ep.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *last_flag = (old == (unsigned)(ep.n_tiles - 1)) ? 1u : 0u;
            }
            __syncthreads();
            if (*last_flag != 0u)
                qf_fused_step_end<1>(N, tiles_n, ep.rowpart, ep.ticket, ep.state_rw, ep.rec, guard.iter, tid, rs + 2 * SBM + 2);
        }
    }
}

// ---- elementwise passes on complex64 matrices (the double-precision forms are in elementwise.hip)

constexpr int TU = 32;

__device__ void step_advance(qf_dev_state *state, qf_host_record *rec, const qf_guard &guard)
{
    // (as qf_step_advance of elementwise.hip: end-of-step bookkeeping by the last block of the update)
    const bool mine = state->step_index == guard.step;
    const bool complete = mine && (state->step_done != 0 || state->iters_this_step >= state->maxit);
    int incomplete = 0;
    if (complete) {
        if (!state->step_done) state->number_of_maxit += 1;   // for-else, isospectral.py:538-540
        rec->last_step_iters = state->iters_this_step;
        state->step_index += 1;
        state->iters_this_step = 0;
        state->step_done = 0;
        rec->resnorm = state->resnorm;
        state->resnorm = __builtin_inf();                     // isospectral.py:470
    } else if (mine) {
        incomplete = 1;
    }
    rec->total_iterations = state->total_iterations;
    rec->number_of_maxit = state->number_of_maxit;
    rec->step_index = state->step_index;
    rec->incomplete = incomplete;
    if (state->fault == QF_FAULT_NONFINITE) rec->nonfinite = 1;      // k_norm_decide closed the call (QF_STEP_ABORTED)
    __hip_atomic_store(&rec->seq, rec->seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// W += 2 (PW - PW^H) (Kahan-compensated: isospectral.py:568-586, in float32 as the reference's complex64 arrays);
// Whalf = W + dW (the next step's first iterate), or dW = 0 with `reinitialize` (isospectral.py:471-472)
template <bool KAHAN>
__global__ __launch_bounds__(256) void k_update_f(int N, const float2 *__restrict__ PW, float2 *__restrict__ W, float2 *dW_a,
                                                  float2 *dW_b, float2 *__restrict__ Whalf, float2 *__restrict__ kc,
                                                  int reinitialize, qf_guard guard, qf_dev_state *state, qf_host_record *rec,
                                                  unsigned *ticket)
{
    const bool due = qf_guard_step_end(guard);
    if (due) {
        float2 *dWc = (guard.state && guard.state->dw_parity) ? dW_b : dW_a;
        const float2 *dW = reinitialize ? nullptr : dWc;
        __shared__ float2 Ts[TU][TU + 1];
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        const int i0 = blockIdx.y * TU, j0 = blockIdx.x * TU;
        for (int r = ty; r < TU; r += 8) {
            const int gj = j0 + r, gi = i0 + tx;
            float2 tv = make_float2(0.f, 0.f);
            if (gj < N && gi < N) tv = PW[(size_t)gj * N + gi];
            Ts[r][tx] = tv;
        }
        __syncthreads();
        for (int r = ty; r < TU; r += 8) {
            const int gi = i0 + r, gj = j0 + tx;
            if (gi < N && gj < N) {
                const size_t e = (size_t)gi * N + gj;
                const float2 pw = PW[e];
                const float2 pwt = Ts[tx][r];
                const float dr = 2.0f * (pw.x - pwt.x);       // conj_subtract_ then `PWcomm *= 2` (:503,547)
                const float di = 2.0f * (pw.y + pwt.y);
                float2 w = W[e];
                if (KAHAN) {
                    float2 c = kc[e];
                    const float yr = dr - c.x, yi = di - c.y;
                    const float tr = w.x + yr, ti = w.y + yi;
                    c.x = (tr - w.x) - yr;
                    c.y = (ti - w.y) - yi;
                    kc[e] = c;
                    w.x = tr;
                    w.y = ti;
                } else {
                    w.x += dr;
                    w.y += di;
                }
                W[e] = w;
                if (dW) {
                    const float2 d = dW[e];
                    Whalf[e] = make_float2(w.x + d.x, w.y + d.y);
                } else {
                    Whalf[e] = w;
                    dWc[e] = make_float2(0.f, 0.f);
                }
            }
        }
    }
    if (state) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (atomicAdd(&ticket[1 + blockIdx.y], 1u) == gridDim.x - 1) {
                ticket[1 + blockIdx.y] = 0;
                if (atomicAdd(&ticket[0], 1u) == gridDim.y - 1) {
                    ticket[0] = 0;
                    step_advance(state, rec, guard);
                }
            }
        }
    }
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// rowsum[i] = sum_j |A[i,j]| (moduli in float32 as np.abs of a complex64 array, sums in double)
__global__ __launch_bounds__(256) void k_row_abs_sum_f(int N, const float2 *__restrict__ A, double *__restrict__ rowsum)
{
    __shared__ double part[4];
    const int i = blockIdx.x;
    double s = 0.0;
    for (int j = threadIdx.x; j < N; j += 256) {
        const float2 z = A[(size_t)i * N + j];
        s += (double)hypotf(z.x, z.y);
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) rowsum[i] = (part[0] + part[1]) + (part[2] + part[3]);
}

// out[0] = max_i rowsum[i] (NaN-propagating), one block
__global__ __launch_bounds__(1024) void k_max_rows_f(int N, const double *__restrict__ rowsum, double *__restrict__ out)
{
    __shared__ double part[16];
    __shared__ int nanp[16];
    double m = 0.0;
    int nan = 0;
    for (int i = threadIdx.x; i < N; i += 1024) {
        const double s = rowsum[i];
        if (s != s) nan = 1; else m = fmax(m, s);
    }
    m = wave_max_d(m);
    const double nn = wave_max_d((double)nan);
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6] = m;
        nanp[threadIdx.x >> 6] = nn > 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        bool anynan = false;
        for (int w = 0; w < 16; ++w) {
            if (nanp[w]) anynan = true;
            r = fmax(r, part[w]);
        }
        out[0] = anynan ? __builtin_nan("") : r;
    }
}

// partial[b] = sum Re(A conj(B)), partial[1024 + b] = sum |A|^2 over the entries of block b (products in
// float32, sums in double); k_sum2_f adds the partials in block order
__global__ __launch_bounds__(256) void k_inner2_partial_f(size_t n, const float2 *__restrict__ A, const float2 *__restrict__ B,
                                                          double *__restrict__ partial)
{
    __shared__ double p0[4], p1[4];
    double s0 = 0.0, s1 = 0.0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float2 a = A[e], b = B[e];
        s0 += (double)(a.x * b.x + a.y * b.y);
        s1 += (double)(a.x * a.x + a.y * a.y);
    }
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    if ((threadIdx.x & 63) == 0) {
        p0[threadIdx.x >> 6] = s0;
        p1[threadIdx.x >> 6] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (p0[0] + p0[1]) + (p0[2] + p0[3]);
        partial[1024 + blockIdx.x] = (p1[0] + p1[1]) + (p1[2] + p1[3]);
    }
}

__global__ __launch_bounds__(64) void k_sum2_f(int n, const double *__restrict__ partial, double *__restrict__ out)
{
    double s0 = 0.0, s1 = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) {
        s0 += partial[i];
        s1 += partial[1024 + i];
    }
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    if (threadIdx.x == 0) {
        out[0] = s0;
        out[1] = s1;
    }
}

// defect[b] = max |A[i,j] + conj(A[j,i])| over the rows of block b (NaN -> inf)
__global__ __launch_bounds__(256) void k_skew_defect_f(int N, const float2 *__restrict__ A, double *__restrict__ defect)
{
    __shared__ double sd[4];
    double d = 0.0;
    bool nan = false;
    for (int i = blockIdx.x; i < N; i += gridDim.x)
        for (int j = threadIdx.x; j < N; j += 256) {
            const float2 x = A[(size_t)i * N + j], y = A[(size_t)j * N + i];
            const float dr = x.x + y.x, di = x.y - y.y;
            const double dd = fmax(fabs((double)dr), fabs((double)di));
            if (dd != dd) nan = true;
            d = fmax(d, dd);
        }
    if (nan) d = __builtin_inf();
    d = wave_max_d(d);
    if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) defect[blockIdx.x] = fmax(fmax(sd[0], sd[1]), fmax(sd[2], sd[3]));
}

__global__ void k_max_partials_f(int n, const double *__restrict__ p, double *__restrict__ out)
{
    double d = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) d = fmax(d, p[i]);
    d = wave_max_d(d);
    if (threadIdx.x == 0) out[0] = d;
}

// out = a X + b Y (elementwise, float32)
__global__ __launch_bounds__(256) void k_lincomb_f(size_t n, float a, const float2 *X, float b, const float2 *Y, float2 *out)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const float2 x = X[e], y = Y[e];
        out[e] = make_float2(a * x.x + b * y.x, a * x.y + b * y.y);
    }
}

}  // namespace

// Tile sizes of the complex64 products.  The 32 x 32 kernels are small (18-34 KiB of LDS, 50-82 registers): several
// workgroups share a CU, so they neither leave CUs idle when the tile count is no multiple of 256 nor need N % 64 == 0.
// Measured (bench.py --dtype c64, timesteps/s, all products on 64 x 64 / on 32 x 32 tiles): N = 768 9,698 / 12,315; 832
// 9,275 / 11,092; 896 8,856 / 9,232; 960 8,391 / 8,434; 1024 7,216 / 7,757; 1056 3,016 / 6,475 (64 x 64: generic path);
// 1088 4,644 / 6,144; 1152 4,438 / 5,332; 1280 4,058 / 4,307; 1536 2,461 / 2,922; 1792 1,651 / 1,873; 2048 1,339 / 1,328.
// By kernel: the SECOND product (triangle, exchange, epilogue) is faster on 32 x 32 tiles at every size (N = 1024 59.2
// against 71.6 us, N = 2048 299 against 320); the plain FIRST product is faster on 64 x 64 tiles where those fill the
// chip (us, 64 / 32: N = 896 52.3 / 54.2, 960 56.1 / 58.2, 1024 59.0 / 61.8, 2048 402 / 433) and slower elsewhere (768
// 46.1 / 38.5, 1088 113 / 78, 1280 132 / 124, 1536 230 / 188, 1792 348 / 305).
// Rules: second product 32 x 32 at every N;
// first product 64 x 64 for N % 64 == 0 with 896 <= N <= 1024 or, from N = 2048 on, tiles filling >= 85 % of their rounds.
int qf_c64_tile(const qf_ctx *ctx)
{
    (void)ctx;
    // (no exact tiling: the generic paths with bounds checks either way -- N = 1000 3,586 -> 5,922 timesteps/s on the 32 x 32
    // kernels, N = 900 3,907 -> 6,406, N = 1500 1,666 -> 2,180)
    return SBM;
}

int qf_c64_tile_first(const qf_ctx *ctx)
{
    const int N = ctx->N;
    if (N % CBM != 0) return SBM;
    if (N >= 896 && N <= 1024) return CBM;
    if (N >= 2048) {
        const long long tiles = (long long)(N / CBM) * (N / CBM), rounds = (tiles + 255) / 256;
        return tiles * 100 >= rounds * 256 * 85 ? CBM : SBM;
    }
    return SBM;
}

int qf_c64_alloc(qf_ctx *ctx)
{
    if (ctx->c64) return QF_OK;
    const int N = ctx->N;
    const size_t NN = (size_t)N * N, mbytes = NN * sizeof(float2);
    qf_c64 *f = new qf_c64();
    float2 **mats[] = {&f->W, &f->dW[0], &f->dW[1], &f->Whalf, &f->Phalf, &f->PW, &f->stage};
    for (float2 **m : mats) {
        if (hipMalloc((void **)m, mbytes) != hipSuccess || hipMemsetAsync(*m, 0, mbytes, ctx->stream) != hipSuccess) {
            qf_set_error("qf_c64_alloc: out of device memory (N=%d)", N);
            qf_c64_free(f);
            return QF_ERR_HIP;
        }
    }
    if (hipMalloc((void **)&f->lap, 2 * NN * sizeof(float)) != hipSuccess || hipMalloc((void **)&f->tab, mbytes) != hipSuccess) {
        qf_set_error("qf_c64_alloc: out of device memory (N=%d)", N);
        qf_c64_free(f);
        return QF_ERR_HIP;
    }
    f->rowpart_tiles = qf_c64_tile(ctx) == SBM ? (N + SBN - 1) / SBN : (N + CBN - 1) / CBN;     // column tiles of the second product in use
    if (hipMalloc((void **)&f->rowpart, (size_t)f->rowpart_tiles * N * sizeof(double)) != hipSuccess) {
        qf_set_error("qf_c64_alloc: out of device memory (N=%d)", N);
        qf_c64_free(f);
        return QF_ERR_HIP;
    }
    ctx->c64 = f;
    // float32 coefficient table (bc = True) and its factorisation, once per N (cpu.py:725)
    QF_TRY(qf_launch_lap_table_f32(ctx, 1, f->lap));
    QF_TRY(qf_launch_build_factors_f32(ctx, f->lap, f->tab));
    return QF_OK;
}

void qf_c64_free(qf_c64 *f)
{
    if (!f) return;
    void *ptrs[] = {f->W, f->dW[0], f->dW[1], f->Whalf, f->Phalf, f->PW, f->stage, f->kahan_c, f->lap, f->tab, f->rowpart, f->W2,
                    f->Whalf2, f->tri_partial, f->tri_arrive};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete f;
}

// qf_plan_describe: what a complex64 product launcher launches (the role is the caller's open prof_scope)
static void note_c64(qf_ctx *ctx, const char *kernel, int tile, int tiles, int grid_tiles, int wgs, int threads, const char *mfma,
                     const char *step_end)
{
    unsigned long long key = 0x7000000ull ^ ((unsigned long long)tile << 20) ^ ((unsigned long long)wgs << 8) ^ (unsigned long long)threads;
    for (const char *c = kernel; *c; ++c) key = key * 131ull + (unsigned char)*c;
    qf_plan_note(ctx, key | 1ull,
                 "{\"kernel\": \"%s\", \"arithmetic\": \"fp32 3M, %s\", \"tile\": [%d, %d], \"tiles\": %d, \"tile_share\": %.6f, "
                 "\"workgroups\": %d, \"threads\": %d, \"step_end\": \"%s\"}",
                 kernel, mfma, tile, tile, tiles, (double)tiles / (double)grid_tiles, wgs, threads, step_end);
}

int qf_launch_cgemm(qf_ctx *ctx, const float2 *A, const float2 *B, float2 *C, const qf_epilogue_f *ep_in, qf_guard guard)
{
    const int N = ctx->N;
    qf_epilogue_f ep_copy;
    const qf_epilogue_f *ep = ep_in;
    if (ep_in && ep_in->fused) {     // tile ticket + what the last tile's workgroup updates
        ep_copy = *ep_in;
        const int t = qf_c64_tile(ctx) == SBM ? (N + SBM - 1) / SBM : (N + CBM - 1) / CBM;
        ep_copy.ticket = ctx->ticket + 403;
        ep_copy.n_tiles = t * t;
        ep_copy.state_rw = ctx->state;
        ep_copy.rec = ctx->host_rec;
        ep = &ep_copy;
    }
    if ((ep ? qf_c64_tile(ctx) : qf_c64_tile_first(ctx)) == SBM) {
        const int tm = (N + SBM - 1) / SBM, tn = (N + SBN - 1) / SBN;
        const bool ex = (N % SBM == 0) && (N % CBK == 0);
        qf_epilogue_f none_s;
        dim3 grid_s(tm * tn), block_s(256);
        if (!ep && ex) {
            // one tile per CU or fewer: more wavefronts per SIMD on the same tile
            const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
            int ks = tm * tn <= cus ? 4 : tm * tn <= 2 * cus ? 2 : 1;
            while (ks > 1 && (N / CBK) % (2 * ks) != 0) ks >>= 1;
            if (ks == 4 || ks == 2) note_c64(ctx, ks == 4 ? "k_cgemm32<plain, 4 K groups>" : "k_cgemm32<plain, 2 K groups>", SBM, tm * tn, tm * tn, tm * tn, 256 * ks, "v_mfma_f32_16x16x4_f32", "none");
            if (ks == 4) {
                static qf_smem_attr attr4;
                QF_TRY(qf_smem_attr_set(attr4, (const void *)k_cgemm32<false, true, 4>, ctx->device, 4 * SG_MAIN_BYTES));
                hipLaunchKernelGGL((k_cgemm32<false, true, 4>), grid_s, dim3(1024), 4 * SG_MAIN_BYTES, ctx->stream, N, tn, A, B, C, none_s, guard);
                QF_HIP(hipGetLastError());
                return QF_OK;
            }
            if (ks == 2) {
                hipLaunchKernelGGL((k_cgemm32<false, true, 2>), grid_s, dim3(512), 2 * SG_MAIN_BYTES, ctx->stream, N, tn, A, B, C, none_s, guard);
                QF_HIP(hipGetLastError());
                return QF_OK;
            }
        }
        note_c64(ctx, ep ? "k_cgemm32<EPI>" : "k_cgemm32<plain>", SBM, tm * tn, tm * tn, tm * tn, 256, "v_mfma_f32_16x16x4_f32",
                 !ep ? "none" : ep->fused ? "fused (last tile decides)" : "two-kernel");
        if (ep) {
            if (ex) hipLaunchKernelGGL((k_cgemm32<true, true>), grid_s, block_s, SG_SMEM, ctx->stream, N, tn, A, B, C, *ep, guard);
            else hipLaunchKernelGGL((k_cgemm32<true, false>), grid_s, block_s, SG_SMEM, ctx->stream, N, tn, A, B, C, *ep, guard);
        } else {
            if (ex) hipLaunchKernelGGL((k_cgemm32<false, true>), grid_s, block_s, SG_SMEM, ctx->stream, N, tn, A, B, C, none_s, guard);
            else hipLaunchKernelGGL((k_cgemm32<false, false>), grid_s, block_s, SG_SMEM, ctx->stream, N, tn, A, B, C, none_s, guard);
        }
        QF_HIP(hipGetLastError());
        return QF_OK;
    }
    const int tiles_m = (N + CBM - 1) / CBM, tiles_n = (N + CBN - 1) / CBN;
    const bool exact = (N % CBM == 0) && (N % CBK == 0);
    if (!ep && exact) {
        // few tiles per CU: more wavefronts per SIMD on the same tile (k_cgemm_ks)
        const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
        int ks = tiles_m * tiles_n <= cus ? 4 : tiles_m * tiles_n <= 2 * cus ? 2 : 1;
        while (ks > 1 && (N / CBK) % (2 * ks) != 0) ks >>= 1;
        if (ks > 1) {
            static qf_smem_attr attr2, attr4;
            note_c64(ctx, ks == 4 ? "k_cgemm_ks<4>" : "k_cgemm_ks<2>", CBM, tiles_m * tiles_n, tiles_m * tiles_n, tiles_m * tiles_n, 256 * ks, "v_mfma_f32_32x32x2_f32", "none");
            if (ks == 2) {
                QF_TRY(qf_smem_attr_set(attr2, (const void *)k_cgemm_ks<2>, ctx->device, 2 * CG_MAIN_BYTES));
                hipLaunchKernelGGL(k_cgemm_ks<2>, dim3(tiles_m * tiles_n), dim3(512), 2 * CG_MAIN_BYTES, ctx->stream, N, tiles_n, A, B, C, guard);
            } else {
                QF_TRY(qf_smem_attr_set(attr4, (const void *)k_cgemm_ks<4>, ctx->device, 4 * CG_MAIN_BYTES));
                hipLaunchKernelGGL(k_cgemm_ks<4>, dim3(tiles_m * tiles_n), dim3(1024), 4 * CG_MAIN_BYTES, ctx->stream, N, tiles_n, A, B, C, guard);
            }
            QF_HIP(hipGetLastError());
            return QF_OK;
        }
    }
    qf_epilogue_f none;
    dim3 grid(tiles_m * tiles_n), block(256);
    note_c64(ctx, ep ? "k_cgemm<EPI>" : "k_cgemm<plain>", CBM, tiles_m * tiles_n, tiles_m * tiles_n, tiles_m * tiles_n, 256, "v_mfma_f32_32x32x2_f32",
             !ep ? "none" : ep->fused ? "fused (last tile decides)" : "two-kernel");
    if (ep) {
        if (exact) hipLaunchKernelGGL((k_cgemm<true, true>), grid, block, CG_SMEM, ctx->stream, N, tiles_n, A, B, C, *ep, guard);
        else hipLaunchKernelGGL((k_cgemm<true, false>), grid, block, CG_SMEM, ctx->stream, N, tiles_n, A, B, C, *ep, guard);
    } else {
        if (exact) hipLaunchKernelGGL((k_cgemm<false, true>), grid, block, CG_SMEM, ctx->stream, N, tiles_n, A, B, C, none, guard);
        else hipLaunchKernelGGL((k_cgemm<false, false>), grid, block, CG_SMEM, ctx->stream, N, tiles_n, A, B, C, none, guard);
    }
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_c64_tri_alloc(qf_ctx *ctx)
{
    qf_c64 *f = ctx->c64;
    const int tb = SBM;          // the upper-triangle product lives on 32 x 32 tiles (qf_c64_tile)
    if (!f || ctx->N < 64) {
        qf_set_error("qf_c64_tri_alloc: the upper-triangle product needs N >= 64 (N=%d)", ctx->N);
        return QF_ERR_INVALID;
    }
    if (f->tri_arrive) return QF_OK;
    const int nt = (ctx->N + tb - 1) / tb;
    const size_t tiles = (size_t)nt * (nt + 1) / 2;
    // K pieces per off-diagonal / diagonal tile.  A workgroup is small (18 KiB of LDS, 82 registers), several share a CU
    // and the pieces of a tile spread over them.  Measured (timesteps/s; full product / pieces 1,1 / 2,2 / 4,4 / 4,2):
    // N = 512 20,175 / 21,603 / 22,982 / 23,663 / 23,753; N = 704 12,498 / 14,331 / 15,224 / 15,415 / 14,759; N = 256
    // 30,769 / 32,017 / 32,632 / 33,249 / 33,335.
    f->tri_split = 4;
    f->tri_split_diag = 4;
    // (a piece must be at least two K-tiles of 16)
    while (f->tri_split > 1 && (ctx->N + CBK - 1) / CBK / f->tri_split < 2) f->tri_split >>= 1;
    while (f->tri_split_diag > 1 && (ctx->N + CBK - 1) / CBK / f->tri_split_diag < 2) f->tri_split_diag >>= 1;
    QF_HIP(hipMalloc((void **)&f->tri_partial, tiles * 4 * tb * tb * sizeof(float2)));
    QF_HIP(hipMalloc((void **)&f->tri_arrive, tiles * sizeof(unsigned)));
    f->tri_arrive_count = tiles;
    QF_HIP(hipMemsetAsync(f->tri_arrive, 0, tiles * sizeof(unsigned), ctx->stream));
    return QF_OK;
}

int qf_launch_cgemm_tri(qf_ctx *ctx, const float2 *A, const float2 *B, const qf_epilogue_f *ep_in, qf_guard guard)
{
    const int N = ctx->N;
    qf_c64 *f = ctx->c64;
    if (!ep_in || !f || !f->tri_arrive) {
        qf_set_error("qf_launch_cgemm_tri: not available for this context (N=%d)", N);
        return QF_ERR_STATE;
    }
    const int nt = (N + SBM - 1) / SBM;
    qf_epilogue_f ep = *ep_in;
    if (ep.fused) {     // tile ticket + what the last tile's workgroup updates
        ep.ticket = ctx->ticket + 404;
        ep.n_tiles = nt * (nt + 1) / 2;
        ep.state_rw = ctx->state;
        ep.rec = ctx->host_rec;
    }
    qf_ctri sx;
    sx.partial = f->tri_partial;
    sx.arrive = f->tri_arrive;
    sx.split = f->tri_split;
    sx.split_diag = f->tri_split_diag;
    const int grid = nt * sx.split_diag + nt * (nt - 1) / 2 * sx.split;
    note_c64(ctx, "k_cgemm_tri32 (upper triangle, K pieces per tile)", SBM, nt * (nt + 1) / 2, nt * nt, grid, 256, "v_mfma_f32_16x16x4_f32",
             ep.fused ? "fused (last tile decides)" : "two-kernel");
    if (N % SBM == 0) hipLaunchKernelGGL(k_cgemm_tri32<true>, dim3(grid), dim3(256), ST_SMEM, ctx->stream, N, nt, A, B, ep, guard, sx);
    else hipLaunchKernelGGL(k_cgemm_tri32<false>, dim3(grid), dim3(256), ST_SMEM, ctx->stream, N, nt, A, B, ep, guard, sx);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_mirror_lower_f32(qf_ctx *ctx, float2 *X)
{
    const int tiles = (ctx->N + 31) / 32;
    hipLaunchKernelGGL(k_mirror_lower_f, dim3(tiles, tiles), dim3(256), 0, ctx->stream, ctx->N, X);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_update_f32(qf_ctx *ctx, const float2 *PW, float2 *W, float2 *dW_a, float2 *dW_b, float2 *Whalf, float2 *kahan_c,
                         int reinitialize, qf_guard guard)
{
    const int N = ctx->N;
    dim3 grid((N + TU - 1) / TU, (N + TU - 1) / TU), block(256);
    if (kahan_c)
        hipLaunchKernelGGL(k_update_f<true>, grid, block, 0, ctx->stream, N, PW, W, dW_a, dW_b, Whalf, kahan_c, reinitialize, guard,
                           guard.state ? ctx->state : nullptr, ctx->host_rec, ctx->ticket);
    else
        hipLaunchKernelGGL(k_update_f<false>, grid, block, 0, ctx->stream, N, PW, W, dW_a, dW_b, Whalf, kahan_c, reinitialize, guard,
                           guard.state ? ctx->state : nullptr, ctx->host_rec, ctx->ticket);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_norm_inf_f32(qf_ctx *ctx, const float2 *A, double *out_dev)
{
    hipLaunchKernelGGL(k_row_abs_sum_f, dim3(ctx->N), dim3(256), 0, ctx->stream, ctx->N, A, ctx->rowsum);
    QF_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_max_rows_f, dim3(1), dim3(1024), 0, ctx->stream, ctx->N, ctx->rowsum, out_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_inner2_f32(qf_ctx *ctx, const float2 *A, const float2 *B, double *out_dev)
{
    const size_t n = (size_t)ctx->N * ctx->N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    double *partial = ctx->scalars + 64;    // [2][1024]
    hipLaunchKernelGGL(k_inner2_partial_f, dim3(blocks), dim3(256), 0, ctx->stream, n, A, B, partial);
    QF_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_sum2_f, dim3(1), dim3(64), 0, ctx->stream, blocks, partial, out_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_skew_defect_f32(qf_ctx *ctx, const float2 *A, double *out_dev)
{
    const int N = ctx->N;
    const int blocks = N < 1024 ? N : 1024;
    double *partial = ctx->scalars + 64;
    hipLaunchKernelGGL(k_skew_defect_f, dim3(blocks), dim3(256), 0, ctx->stream, N, A, partial);
    QF_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_max_partials_f, dim3(1), dim3(64), 0, ctx->stream, blocks, partial, out_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_lincomb_f32(qf_ctx *ctx, float a, const float2 *X, float b, const float2 *Y, float2 *out)
{
    const size_t n = (size_t)ctx->N * ctx->N;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_lincomb_f, dim3(blocks), dim3(256), 0, ctx->stream, n, a, X, b, Y, out);
    QF_HIP(hipGetLastError());
    return QF_OK;
}
