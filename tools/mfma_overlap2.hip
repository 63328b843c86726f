// Does an LDS-only wave disturb an f64-MFMA-only wave on the same SIMD?  Block = 8 waves:
// waves 0-3 run MFMAs, waves 4-7 run LDS reads / writes / nothing.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, unsigned long long *stamps, int iters)
{
    __shared__ v4i lds[4096];
    const int wave = threadIdx.x >> 6;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
    v4i d0 = {1, 2, 3, 4}, d1 = d0, d2 = d0, d3 = d0;
    unsigned la = (unsigned)(size_t)(&lds[0]) + (threadIdx.x & 255) * 16;
    lds[threadIdx.x] = d0;
    __syncthreads();
    unsigned long long s0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        asm volatile(
            "s_mov_b32 s20, %[it]\n"
            "1:\n"
            "v_mfma_f64_16x16x4_f64 %[c0], %[a], %[b], %[c0]\n"
            "v_mfma_f64_16x16x4_f64 %[c1], %[a], %[b], %[c1]\n"
            "v_mfma_f64_16x16x4_f64 %[c2], %[a], %[b], %[c2]\n"
            "v_mfma_f64_16x16x4_f64 %[c3], %[a], %[b], %[c3]\n"
            "s_sub_u32 s20, s20, 1\n"
            "s_cmp_lg_u32 s20, 0\n"
            "s_cbranch_scc1 1b\n"
            : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3)
            : [a] "v"(a), [b] "v"(b), [it] "s"(iters)
            : "s20", "scc");
    } else if (MODE == 1) {
        asm volatile(
            "s_mov_b32 s20, %[it]\n"
            "1:\n"
            "ds_read_b128 %[d0], %[la]\n"
            "ds_read_b128 %[d1], %[la] offset:4096\n"
            "ds_read_b128 %[d2], %[la] offset:8192\n"
            "ds_read_b128 %[d3], %[la] offset:12288\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_sub_u32 s20, s20, 1\n"
            "s_cmp_lg_u32 s20, 0\n"
            "s_cbranch_scc1 1b\n"
            : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3)
            : [it] "s"(iters), [la] "v"(la)
            : "s20", "scc", "memory");
    } else if (MODE == 2) {
        asm volatile(
            "s_mov_b32 s20, %[it]\n"
            "1:\n"
            "ds_write_b128 %[la], %[d0]\n"
            "ds_write_b128 %[la], %[d1] offset:4096\n"
            "ds_write_b128 %[la], %[d2] offset:8192\n"
            "ds_write_b128 %[la], %[d3] offset:12288\n"
            "s_waitcnt lgkmcnt(0)\n"
            "s_sub_u32 s20, s20, 1\n"
            "s_cmp_lg_u32 s20, 0\n"
            "s_cbranch_scc1 1b\n"
            : [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3)
            : [it] "s"(iters), [la] "v"(la)
            : "s20", "scc", "memory");
    }
    unsigned long long s1 = __builtin_amdgcn_s_memtime();
    v4d s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + d0[0] + d1[1] + d2[2] + d3[3];
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + wave] = s1 - s0;
}

template <int MODE>
void run(const char *name, double *out, unsigned long long *stamps)
{
    const int iters = 5000, blocks = 256;
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, out, stamps, iters);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> m, l;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? m : l).push_back((double)h[b * 8 + w] / (iters * 4.0));
    std::sort(m.begin(), m.end());
    std::sort(l.begin(), l.end());
    printf("%-40s MFMA wave: %.1f cycles/MFMA;  other wave: %.1f cycles per LDS instruction\n", name, m[m.size() / 2], l[l.size() / 2]);
}

int main()
{
    double *out; unsigned long long *st;
    hipMalloc(&out, 8 * 512 * 256);
    hipMalloc(&st, 8 * 8 * 256);
    run<0>("MFMA waves + idle waves", out, st);
    run<1>("MFMA waves + ds_read_b128 waves", out, st);
    run<2>("MFMA waves + ds_write_b128 waves", out, st);
    return 0;
}
