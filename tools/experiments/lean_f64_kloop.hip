// Probe: how fast is a register-lean fp64 3M K loop (64 x 64 tile, BK = 8, no software pipelining, <= 256 registers so
// that TWO workgroups share a CU) against the library's pipelined one-workgroup-per-CU kernel?  Times C = A @ B pieces:
// `pieces` K ranges per tile, one workgroup each, partial results written to separate buffers (no combine) -- only the
// K-loop rate matters here.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 lean_f64_kloop.hip -o lean_f64_kloop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double2 cplx;
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int BM = 64, BN = 64, BK = 8;
constexpr int ST = 80;                               // doubles per k-row of a plane (64 + 16: k-rows alternate bank halves)
constexpr int PLANE = BK * ST * 8;                   // bytes
constexpr int BUF = 6 * PLANE;                       // Ar Ai As Br Bi Bs
constexpr int SMEM = 2 * BUF;

__global__ __launch_bounds__(256, 2) void k_lean(int N, int tiles_n, int pieces, const cplx *__restrict__ A, const cplx *__restrict__ B, cplx *__restrict__ C)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, q4 = lane >> 4;
    const int piece = blockIdx.x % pieces, tile = blockIdx.x / pieces;
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int i0 = tm * BM, j0 = tn * BN;
    const int KT = N / BK / pieces, kb = piece * KT;
    // staging: A rows tid / 8 (+32), k = tid % 8;  B k-row tid / 32, columns 2 (tid % 32), +1
    const int a_r = tid >> 3, a_k = tid & 7, b_k = tid >> 5, b_j = (tid & 31) * 2;
    cplx ra[2][2], rb[2][2];
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) ra[x][y] = rb[x][y] = make_double2(0.0, 0.0);
    auto load = [&](int kt, cplx (&a)[2], cplx (&b)[2]) __attribute__((always_inline)) {
        const int k0 = (kb + kt) * BK;
        a[0] = A[(size_t)(i0 + a_r) * N + k0 + a_k];
        a[1] = A[(size_t)(i0 + a_r + 32) * N + k0 + a_k];
        const cplx *bp = B + (size_t)(k0 + b_k) * N + j0 + b_j;
        b[0] = bp[0];
        b[1] = bp[1];
    };
    auto store = [&](int buf, const cplx (&a)[2], const cplx (&b)[2]) __attribute__((always_inline)) {
        double *P = reinterpret_cast<double *>(smem + buf * BUF);
        for (int r = 0; r < 2; ++r) {
            const int o = a_k * ST + a_r + 32 * r;
            P[o] = a[r].x;
            P[BK * ST + o] = a[r].y;
            P[2 * BK * ST + o] = a[r].x + a[r].y;
        }
        double *Q = P + 3 * BK * ST + b_k * ST + b_j;
        *reinterpret_cast<double2 *>(Q) = make_double2(b[0].x, b[1].x);
        *reinterpret_cast<double2 *>(Q + BK * ST) = make_double2(b[0].y, b[1].y);
        *reinterpret_cast<double2 *>(Q + 2 * BK * ST) = make_double2(b[0].x + b[0].y, b[1].x + b[1].y);
    };
    v4d acc[3][2][2];
    for (int p = 0; p < 3; ++p) for (int mi = 0; mi < 2; ++mi) for (int ni = 0; ni < 2; ++ni) acc[p][mi][ni] = (v4d){0.0, 0.0, 0.0, 0.0};
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const double *P = reinterpret_cast<const double *>(smem + buf * BUF);
#pragma unroll
        for (int s = 0; s < BK / 4; ++s) {
            double fa[3][2], fb[3][2];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[p][t] = P[p * BK * ST + (4 * s + q4) * ST + wm * 32 + t * 16 + r16];
                    fb[p][t] = P[(3 + p) * BK * ST + (4 * s + q4) * ST + wn * 32 + t * 16 + r16];
                }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[p][mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[p][mi], fb[p][ni], acc[p][mi][ni], 0, 0, 0);
        }
    };
    load(0, ra[0], rb[0]);
    if (KT > 1) load(1, ra[1], rb[1]);
    store(0, ra[0], rb[0]);
    __syncthreads();
    if (KT > 2) load(2, ra[0], rb[0]);
    int kt = 0;
    for (; kt + 1 < KT; kt += 2) {
        compute(0);
        store(1, ra[1], rb[1]);
        if (kt + 3 < KT) load(kt + 3, ra[1], rb[1]);
        __syncthreads();
        compute(1);
        if (kt + 2 < KT) {
            store(0, ra[0], rb[0]);
            if (kt + 4 < KT) load(kt + 4, ra[0], rb[0]);
        }
        __syncthreads();
    }
    if (kt < KT) compute(0);
    cplx *Cp = C + (size_t)piece * N * N;
    for (int mi = 0; mi < 2; ++mi)
        for (int ni = 0; ni < 2; ++ni)
            for (int reg = 0; reg < 4; ++reg) {
                const int gi = i0 + wm * 32 + mi * 16 + q4 + 4 * reg, gj = j0 + wn * 32 + ni * 16 + r16;
                const double re = acc[0][mi][ni][reg] - acc[1][mi][ni][reg];
                const double im = (acc[2][mi][ni][reg] - acc[0][mi][ni][reg]) - acc[1][mi][ni][reg];
                Cp[(size_t)gi * N + gj] = make_double2(re, im);
            }
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 1024;
    const size_t NN = (size_t)N * N;
    std::vector<cplx> hA(NN), hB(NN), hC(4 * NN);
    srand(1);
    for (size_t e = 0; e < NN; ++e) {
        hA[e] = make_double2(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
        hB[e] = make_double2(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
    }
    cplx *A, *B, *C;
    (void)hipMalloc((void **)&A, NN * sizeof(cplx));
    (void)hipMalloc((void **)&B, NN * sizeof(cplx));
    (void)hipMalloc((void **)&C, 4 * NN * sizeof(cplx));
    (void)hipMemcpy(A, hA.data(), NN * sizeof(cplx), hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB.data(), NN * sizeof(cplx), hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute((const void *)k_lean, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int tiles = (N / 64) * (N / 64);
    for (int pieces : {1, 2, 4}) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, 0);
            for (int q = 0; q < 20; ++q) hipLaunchKernelGGL(k_lean, dim3(tiles * pieces), dim3(256), SMEM, 0, N, N / 64, pieces, A, B, C);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double us = best * 1e3 / 20, tf = 6.0 * N * (double)N * N / (us * 1e-6) / 1e12;
        printf("N=%d pieces=%d (%d workgroups): %.1f us per launch, %.1f TFLOP/s executed = %.2f of 78.6\n", N, pieces, tiles * pieces, us, tf, tf / 78.6);
    }
    // check piece sums against a host reference on a few entries
    (void)hipMemset(C, 0, 4 * NN * sizeof(cplx));
    hipLaunchKernelGGL(k_lean, dim3(tiles * 2), dim3(256), SMEM, 0, N, N / 64, 2, A, B, C);
    (void)hipMemcpy(hC.data(), C, 2 * NN * sizeof(cplx), hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int t = 0; t < 64; ++t) {
        const int i = (t * 131) % N, j = (t * 977 + 5) % N;
        double re = 0, im = 0;
        for (int k = 0; k < N; ++k) {
            const cplx a = hA[(size_t)i * N + k], b = hB[(size_t)k * N + j];
            re += a.x * b.x - a.y * b.y;
            im += a.x * b.y + a.y * b.x;
        }
        const cplx c0 = hC[(size_t)i * N + j], c1 = hC[NN + (size_t)i * N + j];
        const double dr = c0.x + c1.x - re, di = c0.y + c1.y - im;
        worst = fmax(worst, fmax(fabs(dr), fabs(di)));
    }
    printf("max error of 64 sampled entries (two pieces summed): %.3e\n", worst);
    return 0;
}
