// What a kernel boundary costs on this machine, and what the shader clock is under load.
//   1. period of back-to-back launches (events around a batch) of
//        - an empty kernel (256 x 256 threads; without / with 132 KiB of dynamic LDS),
//        - a kernel that stores B MiB once (plain / nontemporal / sc1 write-through stores):
//          the end-of-kernel L2 write-back grows with the dirty bytes a kernel leaves behind;
//   2. s_memtime (shader clock) against s_memrealtime (100 MHz) inside a wave that runs f64 MFMAs,
//      with and without the other CUs streaming stores: the clock the chip actually holds;
//   3. per-workgroup start / end stamps (s_memrealtime): dispatch ramp and tail of a grid.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/launch_probe.hip -o tools/launch_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__global__ __launch_bounds__(256) void k_empty(int *sink)
{
    extern __shared__ unsigned char smem[];
    if (sink && threadIdx.x == 1024) sink[0] = smem[0];
}

// every thread stores `per` 16-byte pieces, wave-contiguous; mode 0 plain, 1 nontemporal, 2 sc1, 3 sc0 sc1
template <int MODE>
__global__ __launch_bounds__(256) void k_store(double2 *dst, int per)
{
    const size_t base = ((size_t)blockIdx.x * per) * 256 + threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7fffffff, 0x00020000);
    for (int q = 0; q < per; ++q) {
        const size_t e = base + (size_t)q * 256;
        const double2 v = make_double2((double)q, (double)threadIdx.x);
        if (MODE == 0) dst[e] = v;
        else if (MODE == 1) __builtin_nontemporal_store(*reinterpret_cast<const v4u *>(&v), reinterpret_cast<v4u *>(dst + e));
        else __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const v4u *>(&v), rs, (unsigned)(e * 16), 0,
                                                     MODE == 2 ? 16 : 17);
    }
}

// reads `per` 16-byte pieces per thread (sum kept alive): a kernel that leaves nothing dirty
__global__ __launch_bounds__(256) void k_load(const double2 *src, int per, double *sink)
{
    const size_t base = ((size_t)blockIdx.x * per) * 256 + threadIdx.x;
    double s = 0.0;
    for (int q = 0; q < per; ++q) s += src[base + (size_t)q * 256].x;
    if (s == 12345.678) sink[0] = s;
}

// f64 MFMA chain for `iters` MFMAs per wave; stamps: [block][0..3] = realtime start, realtime end, memtime start, memtime end
__global__ __launch_bounds__(256) void k_mfma(int iters, unsigned long long *stamps, double *sink)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i += 4) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
    }
    const double s = acc0[0] + acc1[1] + acc2[2] + acc3[3];
    if (s == 12345.678) sink[0] = s;
    __syncthreads();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        stamps[blockIdx.x * 4 + 0] = r0;
        stamps[blockIdx.x * 4 + 1] = r1;
        stamps[blockIdx.x * 4 + 2] = c0;
        stamps[blockIdx.x * 4 + 3] = c1;
    }
}

template <typename F>
static double period_us(hipStream_t st, int reps, F launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const size_t big = (size_t)256 << 20;
    double2 *buf;
    CK(hipMalloc((void **)&buf, big));
    CK(hipMemset(buf, 0, big));
    double *sink;
    CK(hipMalloc((void **)&sink, 64));
    unsigned long long *stamps;
    CK(hipMalloc((void **)&stamps, 4096 * 4 * sizeof(unsigned long long)));
    CK(hipFuncSetAttribute((const void *)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024));

    // clock warm-up
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_mfma, dim3(256), dim3(256), 0, st, 40000, stamps, sink);
    CK(hipStreamSynchronize(st));

    printf("== 1. period of back-to-back launches (us per launch, 400 launches)\n");
    printf("empty 256x256, no LDS        : %.2f\n", period_us(st, 400, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, nullptr); }));
    printf("empty 256x256, 132 KiB LDS   : %.2f\n", period_us(st, 400, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 132 * 1024, st, nullptr); }));
    printf("empty 1x64                   : %.2f\n", period_us(st, 400, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, nullptr); }));
    for (int mib : {4, 16, 32, 64, 128}) {
        // 1024 blocks x 256 threads x per x 16 B
        const int blocks = 1024;
        const int per = (int)(((size_t)mib << 20) / ((size_t)blocks * 256 * 16));
        const double p0 = period_us(st, 200, [&] { hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(256), 0, st, buf, per); });
        const double p1 = period_us(st, 200, [&] { hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, st, buf, per); });
        const double p2 = period_us(st, 200, [&] { hipLaunchKernelGGL(k_store<2>, dim3(blocks), dim3(256), 0, st, buf, per); });
        const double p3 = period_us(st, 200, [&] { hipLaunchKernelGGL(k_store<3>, dim3(blocks), dim3(256), 0, st, buf, per); });
        const double pl = period_us(st, 200, [&] { hipLaunchKernelGGL(k_load, dim3(blocks), dim3(256), 0, st, buf, per, sink); });
        printf("store %3d MiB: plain %.2f  nontemporal %.2f  sc1 %.2f  sc0sc1 %.2f   | load %.2f   (GB/s plain %.0f, nt %.0f, load %.0f)\n", mib, p0, p1, p2,
               p3, pl, mib * 1.048576e6 / p0 / 1e3, mib * 1.048576e6 / p1 / 1e3, mib * 1.048576e6 / pl / 1e3);
    }
    // a store kernel followed by a load kernel of the same buffer (producer -> consumer across a boundary)
    for (int mib : {16, 64}) {
        const int blocks = 1024;
        const int per = (int)(((size_t)mib << 20) / ((size_t)blocks * 256 * 16));
        const double pa = period_us(st, 200, [&] {
            hipLaunchKernelGGL(k_store<0>, dim3(blocks), dim3(256), 0, st, buf, per);
            hipLaunchKernelGGL(k_load, dim3(blocks), dim3(256), 0, st, buf, per, sink);
        });
        const double pb = period_us(st, 200, [&] {
            hipLaunchKernelGGL(k_store<1>, dim3(blocks), dim3(256), 0, st, buf, per);
            hipLaunchKernelGGL(k_load, dim3(blocks), dim3(256), 0, st, buf, per, sink);
        });
        printf("store+load %3d MiB pair: plain %.2f  nontemporal %.2f\n", mib, pa, pb);
    }

    printf("== 2. shader clock under f64 MFMA load, 3. dispatch ramp (256 blocks x 4 waves, one per SIMD)\n");
    for (int iters : {400, 4000, 40000}) {
        std::vector<unsigned long long> h(256 * 4);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_mfma, dim3(256), dim3(256), 0, st, iters, stamps, sink);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long rmin = ~0ull, rmax = 0, smax = 0;
            double ghz = 0, life = 0;
            for (int b = 0; b < 256; ++b) {
                rmin = std::min(rmin, h[b * 4]);
                smax = std::max(smax, h[b * 4]);
                rmax = std::max(rmax, h[b * 4 + 1]);
                ghz += (double)(h[b * 4 + 3] - h[b * 4 + 2]) / (double)(h[b * 4 + 1] - h[b * 4]) * 0.1;
                life += (double)(h[b * 4 + 1] - h[b * 4]) * 0.01;
            }
            printf("iters %6d rep %d: events %.2f us | first start -> last end %.2f us, start ramp %.2f us, mean wave life %.2f us | memtime/realtime -> %.3f GHz | cycles per MFMA %.1f\n",
                   iters, rep, ms * 1e3, (rmax - rmin) * 0.01, (smax - rmin) * 0.01, life / 256, ghz / 256,
                   (double)(h[3] - h[2]) / iters);
        }
    }
    return 0;
}
