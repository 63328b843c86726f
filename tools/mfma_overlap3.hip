// Cost of a GROUP of ds_read_b128 in front of a group of 16 f64 MFMAs in one wave (the K-loop
// phase structure of zgemm.hip), with C/D in VGPRs or AGPRs.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

#define M4 "v_mfma_f64_16x16x4_f64 %[c0], %[a], %[b], %[c0]\nv_mfma_f64_16x16x4_f64 %[c1], %[a], %[b], %[c1]\nv_mfma_f64_16x16x4_f64 %[c2], %[a], %[b], %[c2]\nv_mfma_f64_16x16x4_f64 %[c3], %[a], %[b], %[c3]\n"
#define R4 "ds_read_b128 %[d0], %[la]\nds_read_b128 %[d1], %[la] offset:4096\nds_read_b128 %[d2], %[la] offset:8192\nds_read_b128 %[d3], %[la] offset:12288\n"
#define LOOPH "s_mov_b32 s20, %[it]\n1:\n"
#define LOOPT "s_sub_u32 s20, s20, 1\ns_cmp_lg_u32 s20, 0\ns_cbranch_scc1 1b\n"

template <int MODE, bool AGPR>
__global__ __launch_bounds__(256) void k(double *out, unsigned long long *stamps, int iters)
{
    __shared__ v4i lds[4096];
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
    v4i d0 = {1, 2, 3, 4}, d1 = d0, d2 = d0, d3 = d0;
    double x0 = a, x1 = b;
    unsigned la = (unsigned)(size_t)(&lds[0]) + (threadIdx.x & 255) * 16;
    lds[threadIdx.x] = d0;
    __syncthreads();
    unsigned long long s0 = __builtin_amdgcn_s_memtime();
#define RUNV(BODY) asm volatile(LOOPH BODY LOOPT : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [x0] "+v"(x0), [x1] "+v"(x1) : [a] "v"(a), [b] "v"(b), [it] "s"(iters), [la] "v"(la) : "s20", "scc", "memory");
#define RUNA(BODY) asm volatile(LOOPH BODY LOOPT : [c0] "+a"(c0), [c1] "+a"(c1), [c2] "+a"(c2), [c3] "+a"(c3), [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [x0] "+v"(x0), [x1] "+v"(x1) : [a] "v"(a), [b] "v"(b), [it] "s"(iters), [la] "v"(la) : "s20", "scc", "memory");
#define RUN(BODY) if (AGPR) { RUNA(BODY) } else { RUNV(BODY) }
    if (MODE == 0) { RUN(M4 M4 M4 M4) }                                   // 16 MFMA
    if (MODE == 1) { RUN(R4 M4 M4 M4 M4) }                                // reads first, no wait
    if (MODE == 2) { RUN(R4 M4 M4 M4 M4 "s_waitcnt lgkmcnt(0)\n") }       // + wait at the end
    if (MODE == 3) { RUN(M4 M4 R4 M4 M4) }                                // reads in the middle
    if (MODE == 4) { RUN(R4 R4 M4 M4 M4 M4) }                             // 8 reads
    if (MODE == 6) { RUN("v_mfma_f64_16x16x4_f64 %[c0], %[a], %[b], %[c0]\nds_write_b128 %[la], %[d0]\nv_mfma_f64_16x16x4_f64 %[c1], %[a], %[b], %[c1]\nds_write_b128 %[la], %[d1] offset:4096\nv_mfma_f64_16x16x4_f64 %[c2], %[a], %[b], %[c2]\nds_write_b128 %[la], %[d2] offset:8192\nv_mfma_f64_16x16x4_f64 %[c3], %[a], %[b], %[c3]\nds_write_b128 %[la], %[d3] offset:12288\n" M4 M4 M4) }   // 4 x (MFMA + ds_write), then 12 MFMA
    if (MODE == 7) { RUN("v_mfma_f64_16x16x4_f64 %[c0], %[a], %[b], %[c0]\nv_add_f64 %[x0], %[x0], %[b]\nv_mfma_f64_16x16x4_f64 %[c1], %[a], %[b], %[c1]\nv_add_f64 %[x1], %[x1], %[b]\nv_mfma_f64_16x16x4_f64 %[c2], %[a], %[b], %[c2]\nv_add_f64 %[x0], %[x0], %[b]\nv_mfma_f64_16x16x4_f64 %[c3], %[a], %[b], %[c3]\nv_add_f64 %[x1], %[x1], %[b]\n" M4 M4 M4) }   // 4 x (MFMA + v_add_f64), then 12 MFMA
    if (MODE == 10) { RUN("v_mfma_f64_16x16x4_f64 %[c0], %[a], %[b], %[c0]\nds_add_f64 %[la], %[x0]\nv_mfma_f64_16x16x4_f64 %[c1], %[a], %[b], %[c1]\nds_add_f64 %[la], %[x1] offset:4096\nv_mfma_f64_16x16x4_f64 %[c2], %[a], %[b], %[c2]\nds_add_f64 %[la], %[x0] offset:8192\nv_mfma_f64_16x16x4_f64 %[c3], %[a], %[b], %[c3]\nds_add_f64 %[la], %[x1] offset:12288\n" M4 M4 M4) }   // 4 x (MFMA + ds_add_f64), then 12 MFMA
    if (MODE == 11) { RUN("v_mfma_f64_16x16x4_f64 %[c0], %[a], %[b], %[c0]\nds_write_b64 %[la], %[x0]\nds_add_f64 %[la], %[x1]\nv_mfma_f64_16x16x4_f64 %[c1], %[a], %[b], %[c1]\nds_write_b64 %[la], %[x0] offset:4096\nds_add_f64 %[la], %[x1] offset:4096\nv_mfma_f64_16x16x4_f64 %[c2], %[a], %[b], %[c2]\nds_write_b64 %[la], %[x0] offset:8192\nds_add_f64 %[la], %[x1] offset:8192\nv_mfma_f64_16x16x4_f64 %[c3], %[a], %[b], %[c3]\nds_write_b64 %[la], %[x0] offset:12288\nds_add_f64 %[la], %[x1] offset:12288\n" M4 M4 M4) }   // 4 x (MFMA + ds_write_b64 + ds_add_f64): a sum formed in LDS
    if (MODE == 8) { RUN(M4 M4 M4 M4 "s_waitcnt lgkmcnt(0)\ns_barrier\n") }   // barrier per 16 MFMA
    if (MODE == 9) { RUN(R4 "s_waitcnt lgkmcnt(0)\n" M4 M4 M4 M4) }   // reads, wait, MFMAs (exposed LDS latency)
    if (MODE == 5) { RUN(M4 "ds_read_b128 %[d0], %[la]\n" M4 "ds_read_b128 %[d1], %[la] offset:4096\n" M4 "ds_read_b128 %[d2], %[la] offset:8192\n" M4 "ds_read_b128 %[d3], %[la] offset:12288\n") }  // 1 read per 4 MFMA
    unsigned long long s1 = __builtin_amdgcn_s_memtime();
    v4d s = c0 + c1 + c2 + c3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + d0[0] + d1[1] + d2[2] + d3[3] + x0 + x1;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = s1 - s0;
}

template <int MODE, bool AGPR>
void run(const char *name, double *out, unsigned long long *stamps)
{
    const int iters = 4000, blocks = 256;
    for (int r = 0; r < 2; ++r) {
        hipLaunchKernelGGL((k<MODE, AGPR>), dim3(blocks), dim3(256), 0, 0, out, stamps, iters);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> m;
    for (auto x : h) m.push_back((double)x / iters);
    std::sort(m.begin(), m.end());
    printf("%-46s %s: %.0f cycles per group of 16 MFMA (ideal 1024)\n", name, AGPR ? "AGPR acc" : "VGPR acc", m[m.size() / 2]);
}

int main()
{
    double *out; unsigned long long *st;
    hipMalloc(&out, 8 * 256 * 256);
    hipMalloc(&st, 8 * 4 * 256);
    run<0, false>("16 MFMA", out, st);
    run<1, false>("4 ds_read + 16 MFMA", out, st);
    run<2, false>("4 ds_read + 16 MFMA + lgkmcnt(0)", out, st);
    run<3, false>("8 MFMA + 4 ds_read + 8 MFMA", out, st);
    run<4, false>("8 ds_read + 16 MFMA", out, st);
    run<5, false>("(4 MFMA + 1 ds_read) x4", out, st);
    run<6, false>("4x(MFMA + ds_write_b128) + 12 MFMA", out, st);
    run<7, false>("4x(MFMA + v_add_f64) + 12 MFMA", out, st);
    run<10, false>("4x(MFMA + ds_add_f64) + 12 MFMA", out, st);
    run<11, false>("4x(MFMA + ds_write_b64 + ds_add_f64) + 12 MFMA", out, st);
    run<8, false>("16 MFMA + lgkmcnt(0) + s_barrier", out, st);
    run<9, false>("4 ds_read + lgkmcnt(0) + 16 MFMA", out, st);
    run<0, true>("16 MFMA", out, st);
    run<1, true>("4 ds_read + 16 MFMA", out, st);
    run<3, true>("8 MFMA + 4 ds_read + 8 MFMA", out, st);
    run<4, true>("8 ds_read + 16 MFMA", out, st);
    run<5, true>("(4 MFMA + 1 ds_read) x4", out, st);
    return 0;
}
