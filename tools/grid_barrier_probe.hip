// What does a grid-wide barrier cost on MI355X -- the price of every kernel boundary a persistent iteration kernel would
// remove?  256 workgroups of 256 threads (one per CU, all resident), `iters` barriers back to back:
// arrive = one agent-scope atomic add by one lane after the workgroup's own barrier; wait = that lane polls the counter
// (agent-scope loads, s_sleep between looks) until all have arrived; then the workgroup's barrier again.
// Variant 1 adds what a data hand-off across XCDs needs on top: every thread stores 16 bytes write-through (sc1) and
// drains them (s_waitcnt vmcnt(0)) before it arrives.
// Build: hipcc -O2 --offload-arch=gfx950 tools/grid_barrier_probe.hip -o tools/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_bar(unsigned *counter, double2 *buf, int iters, int variant, unsigned long long *cyc)
{
    const unsigned G = gridDim.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)(G * 256 * 16), 0x00020000);
    for (int i = 0; i < iters; ++i) {
        if (variant) {
            const v4u v = {(unsigned)i, threadIdx.x, blockIdx.x, 0u};
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, (blockIdx.x * 256 + threadIdx.x) * 16, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(i + 1) * G;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
}
__global__ void k_empty() {}
int main()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int G = prop.multiProcessorCount;
    unsigned *counter;
    double2 *buf;
    unsigned long long *cyc;
    (void)hipMalloc(&counter, 64);
    (void)hipMalloc(&buf, (size_t)G * 256 * 16);
    (void)hipMalloc(&cyc, (size_t)G * 8);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 2000;
    printf("{\"probe\": \"grid_barrier\", \"workgroups\": %d, \"barriers_per_launch\": %d, \"runs\": [\n", G, iters);
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(counter, 0, 64);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_bar, dim3(G), dim3(256), 0, 0, counter, buf, iters, variant, cyc);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf(" {\"variant\": \"%s\", \"us_per_barrier\": %.3f},\n", variant ? "write-through store + drain + barrier" : "barrier only", best * 1e3 / iters);
    }
    // and the thing it would replace: the period of back-to-back dependent (empty) launches in one stream
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_empty, dim3(G), dim3(256), 0, 0);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf(" {\"variant\": \"empty kernel launches back to back (the boundary a barrier would replace)\", \"us_per_launch\": %.3f}\n]}\n", ms * 1e3 / 2000);
    return 0;
}
