// In-kernel clock and cycles-per-MFMA probe for v_mfma_f64_16x16x4_f64 on gfx950.
// s_memtime = shader cycles, s_memrealtime = 100 MHz.  The MFMA loop is ONE inline-asm
// statement (hipcc otherwise shuttles loop-carried accumulators between AGPRs and VGPRs).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_clock.hip -o tools/mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));

// MODE 0: 4 independent accumulators; 1: 8 accumulators; 2: 1 accumulator (dependent chain);
// MODE 3: 4 accumulators + 8 independent v_fma_f64 per MFMA group (VALU co-issue)
template <int MODE>
__global__ __launch_bounds__(256) void k_mfma(double *out, unsigned long long *stamps, int iters, double a0, double b0)
{
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    double a2 = a + 1, b2 = b - 1;
    double x0 = a, x1 = b, x2 = a2, x3 = b2, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3, m = 0.999;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 0) {
        asm volatile(
            "s_mov_b32 s20, %6\n"
            "1:\n"
            "v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n"
            "v_mfma_f64_16x16x4_f64 %1, %7, %8, %1\n"
            "v_mfma_f64_16x16x4_f64 %2, %4, %8, %2\n"
            "v_mfma_f64_16x16x4_f64 %3, %7, %5, %3\n"
            "s_sub_u32 s20, s20, 1\n"
            "s_cmp_lg_u32 s20, 0\n"
            "s_cbranch_scc1 1b\n"
            : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)
            : "v"(a), "v"(b), "s"(iters), "v"(a2), "v"(b2)
            : "s20", "scc");
    } else if (MODE == 1) {
        asm volatile(
            "s_mov_b32 s20, %10\n"
            "1:\n"
            "v_mfma_f64_16x16x4_f64 %0, %8, %9, %0\n"
            "v_mfma_f64_16x16x4_f64 %1, %11, %12, %1\n"
            "v_mfma_f64_16x16x4_f64 %2, %8, %12, %2\n"
            "v_mfma_f64_16x16x4_f64 %3, %11, %9, %3\n"
            "v_mfma_f64_16x16x4_f64 %4, %8, %9, %4\n"
            "v_mfma_f64_16x16x4_f64 %5, %11, %12, %5\n"
            "v_mfma_f64_16x16x4_f64 %6, %8, %12, %6\n"
            "v_mfma_f64_16x16x4_f64 %7, %11, %9, %7\n"
            "s_sub_u32 s20, s20, 1\n"
            "s_cmp_lg_u32 s20, 0\n"
            "s_cbranch_scc1 1b\n"
            : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3), "+a"(c4), "+a"(c5), "+a"(c6), "+a"(c7)
            : "v"(a), "v"(b), "s"(iters), "v"(a2), "v"(b2)
            : "s20", "scc");
    } else if (MODE == 2) {
        asm volatile(
            "s_mov_b32 s20, %3\n"
            "1:\n"
            "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n"
            "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n"
            "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n"
            "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n"
            "s_sub_u32 s20, s20, 1\n"
            "s_cmp_lg_u32 s20, 0\n"
            "s_cbranch_scc1 1b\n"
            : "+a"(c0)
            : "v"(a), "v"(b), "s"(iters)
            : "s20", "scc");
    } else {
        asm volatile(
            "s_mov_b32 s20, %6\n"
            "1:\n"
            "v_mfma_f64_16x16x4_f64 %0, %4, %5, %0\n"
            "v_fma_f64 %9, %9, %17, %4\n"
            "v_fma_f64 %10, %10, %17, %4\n"
            "v_fma_f64 %11, %11, %17, %4\n"
            "v_fma_f64 %12, %12, %17, %4\n"
            "v_mfma_f64_16x16x4_f64 %1, %7, %8, %1\n"
            "v_fma_f64 %13, %13, %17, %4\n"
            "v_fma_f64 %14, %14, %17, %4\n"
            "v_fma_f64 %15, %15, %17, %4\n"
            "v_fma_f64 %16, %16, %17, %4\n"
            "v_mfma_f64_16x16x4_f64 %2, %4, %8, %2\n"
            "v_fma_f64 %9, %9, %17, %4\n"
            "v_fma_f64 %10, %10, %17, %4\n"
            "v_fma_f64 %11, %11, %17, %4\n"
            "v_fma_f64 %12, %12, %17, %4\n"
            "v_mfma_f64_16x16x4_f64 %3, %7, %5, %3\n"
            "v_fma_f64 %13, %13, %17, %4\n"
            "v_fma_f64 %14, %14, %17, %4\n"
            "v_fma_f64 %15, %15, %17, %4\n"
            "v_fma_f64 %16, %16, %17, %4\n"
            "s_sub_u32 s20, s20, 1\n"
            "s_cmp_lg_u32 s20, 0\n"
            "s_cbranch_scc1 1b\n"
            : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3)
            : "v"(a), "v"(b), "s"(iters), "v"(a2), "v"(b2), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5),
              "v"(x6), "v"(x7), "v"(m)
            : "s20", "scc");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    v4d s4 = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s4[0] + s4[1] + s4[2] + s4[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int MODE>
void run(const char *name, int blocks, int iters, double *out, unsigned long long *stamps)
{
    const int per_iter = MODE == 1 ? 8 : 4;
    int waves = blocks * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_mfma<MODE>), dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 1.0, 0.5);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> h(2 * waves);
    hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (int w = 0; w < waves; ++w) {
        cyc.push_back((double)h[2 * w] / ((double)iters * per_iter));
        ghz.push_back((double)h[2 * w] / ((double)h[2 * w + 1] * 10.0));
    }
    std::sort(cyc.begin(), cyc.end());
    std::sort(ghz.begin(), ghz.end());
    double tf = (double)waves * iters * per_iter * 2048.0 / (ms * 1e-3) / 1e12;
    printf("%-40s blocks=%4d cyc/MFMA/wave med %.1f (min %.1f max %.1f)  clock med %.3f GHz (min %.3f)  wall %.3f ms  MFMA %.1f TFLOP/s\n",
           name, blocks, cyc[waves / 2], cyc.front(), cyc.back(), ghz[waves / 2], ghz.front(), ms, tf);
}

int main()
{
    double *out; unsigned long long *stamps;
    hipMalloc(&out, sizeof(double) * 256 * 4096);
    hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 4 * 4096);
    const int iters = 20000;
    run<0>("4 acc, 1 wave/SIMD", 256, iters, out, stamps);
    run<1>("8 acc, 1 wave/SIMD", 256, iters, out, stamps);
    run<2>("1 acc (dependent), 1 wave/SIMD", 256, iters, out, stamps);
    run<0>("4 acc, 2 waves/SIMD", 512, iters, out, stamps);
    run<0>("4 acc, 4 waves/SIMD", 1024, iters, out, stamps);
    run<0>("4 acc, ONE block on the chip", 1, iters, out, stamps);
    run<3>("4 acc + 16 v_fma_f64 per 4 MFMA, 1 w/SIMD", 256, iters, out, stamps);
    run<3>("4 acc + 16 v_fma_f64 per 4 MFMA, 2 w/SIMD", 512, iters, out, stamps);
    return 0;
}
