// Micro-benchmarks that calibrate the fp64 roofline of the MI355X for this project:
//   (1) back-to-back v_mfma_f64_16x16x4_f64 issue rate (independent accumulators),
//   (2) plain v_fma_f64 VALU rate,
//   (3) both interleaved in one wave / in different waves of a CU (do the pipes add up?).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters, double a0, double b0)
{
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
__global__ __launch_bounds__(256) void k_valu(double *out, int iters, double a0, double b0)
{
    double x[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = a0 + i + threadIdx.x;
    double m = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i) x[i] = __builtin_fma(x[i], m, a0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave issues NACC MFMAs and NV VALU FMAs per iteration
template <int NACC, int NV>
__global__ __launch_bounds__(256) void k_mix(double *out, int iters, double a0, double b0)
{
    v4d acc[NACC];
    double x[NV];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NV; ++i) x[i] = a0 + i + threadIdx.x;
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3, m = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV / NACC; ++j) x[i * (NV / NACC) + j] = __builtin_fma(x[i * (NV / NACC) + j], m, a0);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < NV; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves 0-3 of a 512-thread block do MFMA, waves 4-7 do VALU (two waves per SIMD, different pipes)
__global__ __launch_bounds__(512) void k_split(double *out, int iters, double a0, double b0)
{
    const bool mf = (threadIdx.x >> 6) < 4;
    v4d acc[4];
    double x[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x;
    double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3, m = b0;
    if (mf) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = __builtin_fma(x[i], m, a0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * 512 * 4096));
    const int iters = 4000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int cus = prop.multiProcessorCount;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        int blocks = cus * bpc;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); }, 5);
        double fl = (double)blocks * 4 /*waves*/ * iters * 4 * 2048.0;
        printf("mfma f64 16x16x4, 4 acc, %d block(s)/CU (x4 waves): %.3f ms  %.1f TFLOP/s\n", bpc, ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); }, 5);
        fl = (double)blocks * 4 * iters * 8 * 2048.0;
        printf("mfma f64 16x16x4, 8 acc, %d block(s)/CU: %.3f ms  %.1f TFLOP/s\n", bpc, ms, fl / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); }, 5);
        fl = (double)blocks * 4 * iters * 1 * 2048.0;
        printf("mfma f64 16x16x4, 1 acc (dependent chain), %d block(s)/CU: %.3f ms  %.1f TFLOP/s\n", bpc, ms, fl / ms / 1e9);
    }
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        int blocks = cus * bpc;
        double ms = time_ms([&] { hipLaunchKernelGGL(k_valu<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); }, 5);
        double fl = (double)blocks * 256 * iters * 16 * 2.0;
        printf("valu v_fma_f64 x16 indep, %d block(s)/CU: %.3f ms  %.1f TFLOP/s\n", bpc, ms, fl / ms / 1e9);
    }
    {
        int blocks = cus;
        double ms = time_ms([&] { hipLaunchKernelGGL((k_mix<4, 16>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); }, 5);
        double flm = (double)blocks * 4 * iters * 4 * 2048.0, flv = (double)blocks * 256 * iters * 16 * 2.0;
        printf("mix same wave: 4 mfma + 16 fma / iter: %.3f ms  mfma %.1f + valu %.1f = %.1f TFLOP/s\n", ms,
               flm / ms / 1e9, flv / ms / 1e9, (flm + flv) / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL((k_mix<4, 32>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); }, 5);
        flv = (double)blocks * 256 * iters * 32 * 2.0;
        printf("mix same wave: 4 mfma + 32 fma / iter: %.3f ms  mfma %.1f + valu %.1f = %.1f TFLOP/s\n", ms,
               flm / ms / 1e9, flv / ms / 1e9, (flm + flv) / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_split, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0, 0.5); }, 5);
        flm = (double)blocks * 4 * iters * 4 * 2048.0;
        flv = (double)blocks * 256 * iters * 64 * 2.0;
        printf("split waves (4 mfma waves + 4 valu waves per CU): %.3f ms  mfma %.1f + valu %.1f = %.1f TFLOP/s\n", ms,
               flm / ms / 1e9, flv / ms / 1e9, (flm + flv) / ms / 1e9);
    }
    hipFree(out);
    return 0;
}
