// Diagnostic harness for the zgemm kernel: builds quflow_amd/csrc/zgemm.hip with in-kernel
// s_memtime stamps (QF_STAMP) and prints where a K-tile spends its cycles.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/zgemm_probe.hip -o tools/zgemm_probe
#define QF_STAMP 1
#include "../quflow_amd/csrc/zgemm.hip"
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <random>
#include <vector>

void qf_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 1024;
    const int epi = argc > 2 ? atoi(argv[2]) : 0;
    qf_ctx ctx;
    ctx.N = N;
    hipStreamCreate(&ctx.stream);
    const size_t NN = (size_t)N * N;
    std::vector<double> h(2 * NN);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    cplx *A, *B, *C, *W, *D0, *D1, *WH;
    double *rowpart;
    for (cplx **p : {&A, &B, &C, &W, &D0, &D1, &WH}) {
        hipMalloc((void **)p, NN * sizeof(cplx));
        for (auto &x : h) x = nd(rng);
        hipMemcpy(*p, h.data(), NN * sizeof(cplx), hipMemcpyHostToDevice);
    }
    hipMalloc((void **)&rowpart, (size_t)64 * N * sizeof(double));
    const int KT = (N + 15) / 16;
    const int nwaves = ((N + 63) / 64) * ((N + 63) / 64) * 4;
    unsigned long long *stamps;
    hipMalloc((void **)&stamps, (size_t)nwaves * QF_STAMP_SLOTS * sizeof(unsigned long long));
    hipMemset(stamps, 0, (size_t)nwaves * QF_STAMP_SLOTS * sizeof(unsigned long long));
    hipMemcpyToSymbol(HIP_SYMBOL(qf_stamp_buf), &stamps, sizeof(stamps));
    unsigned long long *phases;
    const int nblocks = ((N + 63) / 64) * ((N + 63) / 64);
    hipMalloc((void **)&phases, (size_t)nblocks * 40 * sizeof(unsigned long long));
    hipMemset(phases, 0, (size_t)nblocks * 40 * sizeof(unsigned long long));
    if (getenv("QF_PHASES")) hipMemcpyToSymbol(HIP_SYMBOL(qf_phase_buf), &phases, sizeof(phases));
    if (getenv("QF_NOSTAMPS")) { unsigned long long *nul = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(qf_stamp_buf), &nul, sizeof(nul)); }
    qf_epilogue ep;
    ep.PW = A; ep.W = W; ep.dW[0] = D0; ep.dW[1] = D1; ep.Whalf = WH; ep.rowpart = rowpart;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, ctx.stream);
        qf_launch_zgemm(&ctx, A, B, C, epi ? &ep : nullptr);
        hipEventRecord(e1, ctx.stream);
        hipStreamSynchronize(ctx.stream);
        hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d: %.1f us  (%.1f TFLOP/s)\n", rep, ms * 1e3, 8.0 * N * (double)N * N / (ms * 1e-3) / 1e12);
    }
    std::vector<unsigned long long> st((size_t)nwaves * QF_STAMP_SLOTS);
    hipMemcpy(st.data(), stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    const int last = std::min(KT + 2, QF_STAMP_SLOTS - 1);
    std::vector<double> total, prologue, epilog, per_tile;
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int w = 0; w < nwaves; ++w) {
        const unsigned long long *s = &st[(size_t)w * QF_STAMP_SLOTS];
        tmin = std::min(tmin, s[0]);
        tmax = std::max(tmax, s[last]);
        total.push_back((double)(s[last] - s[0]));
        prologue.push_back((double)(s[1] - s[0]));
        if (KT + 1 < QF_STAMP_SLOTS) epilog.push_back((double)(s[KT + 2] - s[KT + 1]));
        for (int k = 0; k < KT && k + 2 < QF_STAMP_SLOTS; ++k) per_tile.push_back((double)(s[k + 2] - s[k + 1]));
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    auto mx = [](std::vector<double> v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
    auto mn = [](std::vector<double> v) { return v.empty() ? 0.0 : *std::min_element(v.begin(), v.end()); };
    printf("N=%d epi=%d KT=%d waves=%d (cycles; ideal K-tile = %d MFMA x 64 = %d)\n", N, epi, KT, nwaves, 64, 4096);
    printf("first wave start -> last wave end: %llu cycles\n", tmax - tmin);
    printf("wave total      : median %.0f  min %.0f  max %.0f\n", med(total), mn(total), mx(total));
    printf("prologue        : median %.0f  min %.0f  max %.0f\n", med(prologue), mn(prologue), mx(prologue));
    printf("K-tile          : median %.0f  min %.0f  max %.0f\n", med(per_tile), mn(per_tile), mx(per_tile));
    printf("epilogue        : median %.0f  min %.0f  max %.0f\n", med(epilog), mn(epilog), mx(epilog));
    // K-tile profile of wave 0 of block 0 and of a middle block
    for (int w : {0, nwaves / 2}) {
        printf("wave %d K-tile cycles:", w);
        const unsigned long long *s = &st[(size_t)w * QF_STAMP_SLOTS];
        for (int k = 0; k < KT && k + 2 < QF_STAMP_SLOTS; ++k) printf(" %llu", s[k + 2] - s[k + 1]);
        printf("\n");
    }
    if (getenv("QF_PHASES")) {
        std::vector<unsigned long long> ph((size_t)nblocks * 40);
        hipMemcpy(ph.data(), phases, ph.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> d[4];
        for (int b = 0; b < nblocks; ++b)
            for (int t = 0; t < 8; ++t)
                for (int q = 0; q < 4; ++q) d[q].push_back((double)(ph[((size_t)b * 8 + t) * 5 + q + 1] - ph[((size_t)b * 8 + t) * 5 + q]));
        for (int q = 0; q < 4; ++q) printf("phase %d: median %.0f  min %.0f  max %.0f cycles\n", q, med(d[q]), mn(d[q]), mx(d[q]));
    }
    // start skew
    std::vector<double> starts;
    for (int w = 0; w < nwaves; ++w) starts.push_back((double)(st[(size_t)w * QF_STAMP_SLOTS] - tmin));
    printf("wave start skew : median %.0f  max %.0f\n", med(starts), mx(starts));
    return 0;
}
