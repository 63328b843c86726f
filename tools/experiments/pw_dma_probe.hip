// Timing-only harness of k_zgemm_pw_dma (first product, both operands by LDS DMA) with ablation knobs.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 [-DPW_ABL_NODMA=1 ...] tools/pw_dma_probe.hip -o tools/pw_dma_probe
#include "zgemm_dma.hip"
#include <cstdarg>
#include <cstdio>
#include <random>
#include <vector>

void qf_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int qf_launch_zgemm(qf_ctx *, const cplx *, const cplx *, cplx *, const qf_epilogue *, qf_guard) { return QF_ERR_STATE; }

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 1024;
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    const size_t NN = (size_t)N * N;
    std::vector<double> h(2 * NN);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    cplx *P, *B, *C;
    for (cplx **p : {&P, &B, &C}) {
        hipMalloc((void **)p, NN * sizeof(cplx));
        for (auto &x : h) x = nd(rng);
        hipMemcpy(*p, h.data(), NN * sizeof(cplx), hipMemcpyHostToDevice);
    }
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipFuncSetAttribute((const void *)k_zgemm_pw_dma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PW_SMEM_BYTES);
    const int nt = N / 64;
    qf_guard g;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double best = 1e30;
    for (int batch = 0; batch < 5; ++batch) {
        hipEventRecord(e0, s);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_zgemm_pw_dma, dim3(nt * nt), dim3(256), PW_SMEM_BYTES, s, N, P, B, C, g);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / reps < best) best = ms / reps;
    }
    printf("k_zgemm_pw_dma N=%d: %.2f us per launch (back-to-back, best of 5 batches)\n", N, 1e3 * best);
    return 0;
}
