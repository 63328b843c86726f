// Issue rate of v_mfma_i32_32x32x32_i8: one wave, four waves per CU, whole chip (DVFS-limited floor of
// an N=1024 digit-split product).  Build: hipcc -O2 --offload-arch=gfx950 -Wno-unused-value tools/i8_mfma_rate.hip -o tools/i8probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k_i8(int* out, unsigned long long* cyc, int iters) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)threadIdx.x};
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_bf16(float* out, unsigned long long* cyc, int iters) {
    v8s a = {1,2,3,4,5,6,7,(short)threadIdx.x}, b = {1,2,3,4,5,6,7,8};
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    int* o; float* of; unsigned long long* c; unsigned long long h[4];
    (void)hipMalloc(&o, 4096); (void)hipMalloc(&of, 4096); (void)hipMalloc(&c, 64);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_i8, dim3(1), dim3(64), 0, 0, o, c, iters); (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, c, 8, hipMemcpyDeviceToHost);
        printf("i8 32x32x32 : %.1f cycles per MFMA (one wave)\n", (double)h[0] / (4.0 * iters));
        hipLaunchKernelGGL(k_bf16, dim3(1), dim3(64), 0, 0, of, c, iters); (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, c, 8, hipMemcpyDeviceToHost);
        printf("bf16 32x32x16: %.1f cycles per MFMA (one wave)\n", (double)h[0] / (4.0 * iters));
    }
    // four waves on one CU (one per SIMD)
    hipLaunchKernelGGL(k_i8, dim3(1), dim3(256), 0, 0, o, c, iters); (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, c, 8, hipMemcpyDeviceToHost);
    printf("i8 32x32x32 : %.1f cycles per MFMA (4 waves per CU)\n", (double)h[0] / (4.0 * iters));
    // whole chip: 256 workgroups x 4 waves, 1440 MFMAs per wave (= one N=1024 product), wall time
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_i8, dim3(256), dim3(256), 0, 0, o, c, 360);
        (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("whole chip, 1440 i8 MFMAs per wave (constant small operands): %.1f us (19.2 us at 2.4 GHz)\n", ms * 1e3);
    }
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_i8, dim3(256), dim3(256), 0, 0, o, c, 3600);
        (void)hipEventRecord(e1, 0); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("whole chip, 14400 i8 MFMAs per wave: %.1f us (192 us at 2.4 GHz)\n", ms * 1e3);
    }
    return 0;
}
