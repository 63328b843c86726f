// Does hipExtLaunchKernel(.., hipExtAnyOrderLaunch) let two kernels of ONE stream overlap on gfx950?
// Two spin kernels of 64 workgroups each (a quarter of the chip), back to back: ~2x one kernel's time when they
// serialise, ~1x when the second starts without waiting for the first.
// Build: hipcc -O2 --offload-arch=gfx950 tools/anyorder_probe.hip -o tools/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void k_spin(unsigned long long cycles, unsigned long long *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) out[blockIdx.x] = t0;
}
int main()
{
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long *o;
    (void)hipMalloc(&o, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const unsigned long long cyc = 200000;
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, s);
            hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, cyc, o);
            if (mode == 0) hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, cyc, o + 64);
            else hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, o + 64);
            (void)hipEventRecord(e1, s);
            (void)hipStreamSynchronize(s);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us for two kernels of %llu cycles\n", mode ? "second launch any-order" : "plain launches", ms * 1e3, cyc);
        }
    return 0;
}
