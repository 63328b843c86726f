// What overlaps with v_mfma_f64_16x16x4_f64 in ONE wave?  cycles per MFMA when other
// instruction classes are interleaved 1:1 (asm loop, s_memtime), gfx950.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_overlap.hip -o tools/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define BODY(X)                                              \
    "s_mov_b32 s20, %[it]\n"                                 \
    "1:\n"                                                   \
    "v_mfma_f64_16x16x4_f64 %[c0], %[a], %[b], %[c0]\n" X(0) \
    "v_mfma_f64_16x16x4_f64 %[c1], %[a], %[b], %[c1]\n" X(1) \
    "v_mfma_f64_16x16x4_f64 %[c2], %[a], %[b], %[c2]\n" X(2) \
    "v_mfma_f64_16x16x4_f64 %[c3], %[a], %[b], %[c3]\n" X(3) \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                        \
    "s_sub_u32 s20, s20, 1\n"                                \
    "s_cmp_lg_u32 s20, 0\n"                                  \
    "s_cbranch_scc1 1b\n"

#define X_NONE(i) ""
#define X_DSREAD(i) "ds_read_b128 %[d" #i "], %[la] offset:1" #i "24\n"
#define X_DSWRITE(i) "ds_write_b128 %[la], %[d" #i "] offset:1" #i "24\n"
#define X_GLOAD(i) "global_load_dwordx4 %[d" #i "], %[ga], off offset:1" #i "24\n"
#define X_VMOV(i) "v_add_u32 %[t" #i "], %[t" #i "], %[la]\n"
#define X_VMOV2(i) "v_add_u32 %[t" #i "], %[t" #i "], %[la]\nv_add_u32 %[t" #i "], %[t" #i "], %[la]\nv_add_u32 %[t" #i "], %[t" #i "], %[la]\nv_add_u32 %[t" #i "], %[t" #i "], %[la]\n"
#define X_SALU(i) "s_add_u32 s21, s21, 3\ns_add_u32 s22, s22, 5\n"
#define X_DSWRITE2(i) "ds_write_b128 %[la], %[d" #i "] offset:1" #i "24\nds_write_b128 %[la], %[d" #i "] offset:2" #i "24\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *stamps, int iters, const v4i *gsrc)
{
    __shared__ v4i lds[1024];
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
    v4i d0 = {1, 2, 3, 4}, d1 = d0, d2 = d0, d3 = d0;
    int t0 = threadIdx.x, t1 = t0, t2 = t0, t3 = t0;
    unsigned la = (unsigned)(size_t)(&lds[0]) + (threadIdx.x & 63) * 16;
    const v4i *ga = gsrc + threadIdx.x;
    lds[threadIdx.x] = d0;
    __syncthreads();
    unsigned long long s0 = __builtin_amdgcn_s_memtime();
#define RUN(X)                                                                                              \
    asm volatile(BODY(X)                                                                                    \
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [d0] "+v"(d0), [d1] "+v"(d1), \
                   [d2] "+v"(d2), [d3] "+v"(d3), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3)  \
                 : [a] "v"(a), [b] "v"(b), [it] "s"(iters), [la] "v"(la), [ga] "v"(ga)                       \
                 : "s20", "s21", "s22", "scc", "memory");
    if (MODE == 0) { RUN(X_NONE) }
    if (MODE == 1) { RUN(X_DSREAD) }
    if (MODE == 2) { RUN(X_DSWRITE) }
    if (MODE == 3) { RUN(X_GLOAD) }
    if (MODE == 4) { RUN(X_VMOV) }
    if (MODE == 5) { RUN(X_VMOV2) }
    if (MODE == 6) { RUN(X_SALU) }
    if (MODE == 7) { RUN(X_DSWRITE2) }
    unsigned long long s1 = __builtin_amdgcn_s_memtime();
    v4d s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + d0[0] + d1[1] + d2[2] + d3[3] + t0 + t1 + t2 + t3;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + threadIdx.x / 64] = s1 - s0;
}

template <int MODE>
void run(const char *name, int blocks, double *out, unsigned long long *stamps, const v4i *g)
{
    const int iters = 5000;
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, g);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (auto x : h) c.push_back((double)x / (iters * 4.0));
    std::sort(c.begin(), c.end());
    printf("%-44s cycles per MFMA: median %.1f  (min %.1f max %.1f)\n", name, c[c.size() / 2], c.front(), c.back());
}

int main()
{
    double *out; unsigned long long *st; v4i *g;
    hipMalloc(&out, 8 * 256 * 256);
    hipMalloc(&st, 8 * 1024);
    hipMalloc(&g, 16 * 65536);
    hipMemset(g, 0, 16 * 65536);
    run<0>("MFMA only", 256, out, st, g);
    run<1>("+1 ds_read_b128 per MFMA", 256, out, st, g);
    run<2>("+1 ds_write_b128 per MFMA", 256, out, st, g);
    run<7>("+2 ds_write_b128 per MFMA", 256, out, st, g);
    run<3>("+1 global_load_dwordx4 per MFMA", 256, out, st, g);
    run<4>("+1 v_add_u32 per MFMA", 256, out, st, g);
    run<5>("+4 v_add_u32 per MFMA", 256, out, st, g);
    run<6>("+2 s_add_u32 per MFMA", 256, out, st, g);
    return 0;
}
