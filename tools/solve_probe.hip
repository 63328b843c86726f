// Diagnostic harness for k_solve: times the kernel with parts switched off (QF_PROBE flags).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/solve_probe.hip -o tools/solve_probe
#define QF_PROBE 1
#include "../quflow_amd/csrc/poisson.hip"
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
    qf_ctx ctx;
    ctx.N = N;
    hipStreamCreate(&ctx.stream);
    const size_t NN = (size_t)N * N;
    std::vector<double> h(2 * NN);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    for (auto &x : h) x = nd(rng);
    cplx *W, *P;
    hipMalloc((void **)&W, NN * sizeof(cplx));
    hipMalloc((void **)&P, NN * sizeof(cplx));
    hipMemcpy(W, h.data(), NN * sizeof(cplx), hipMemcpyHostToDevice);
    hipMalloc((void **)&ctx.lap, 2 * NN * sizeof(double));
    qf_factors f;
    hipMalloc((void **)&f.tab, NN * sizeof(double2));
    qf_launch_lap_table(&ctx, 1, ctx.lap);
    qf_launch_build_factors(&ctx, ctx.lap, f);
    hipStreamSynchronize(ctx.stream);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char *names[] = {"full", "no forward store", "no mirror store", "no stores", "no trace", "no trace, no stores"};
    const int flags[] = {0, 1, 2, 3, 4, 7};
    for (int v = 0; v < 6; ++v) {
        int fl = flags[v];
        hipMemcpyToSymbol(HIP_SYMBOL(qf_probe_flags), &fl, sizeof(int));
        float best = 1e9f;
        for (int rep = 0; rep < 10; ++rep) {
            float ms;
            hipEventRecord(e0, ctx.stream);
            qf_launch_solve(&ctx, f, W, P, 1.0, 1);
            hipEventRecord(e1, ctx.stream);
            hipStreamSynchronize(ctx.stream);
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("N=%d %-22s %.1f us\n", N, names[v], best * 1e3);
    }
    // phase stamps (full kernel)
    {
        int fl = 0;
        hipMemcpyToSymbol(HIP_SYMBOL(qf_probe_flags), &fl, sizeof(int));
        const int nw = 4096 * 16;
        unsigned long long *st;
        hipMalloc((void **)&st, (size_t)nw * 16 * sizeof(unsigned long long));
        hipMemset(st, 0, (size_t)nw * 16 * sizeof(unsigned long long));
        hipMemcpyToSymbol(HIP_SYMBOL(qf_probe_stamps), &st, sizeof(st));
        qf_launch_solve(&ctx, f, W, P, 1.0, 1);
        hipStreamSynchronize(ctx.stream);
        std::vector<unsigned long long> hs((size_t)nw * 16);
        hipMemcpy(hs.data(), st, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        const char *ph[] = {"setup+trace", "loads", "pass1", "sync", "scan fwd", "sync", "pass3+4", "sync", "scan bwd", "sync", "pass6+trace", "fwd store+stage", "mirror store"};
        std::vector<std::vector<double>> d(13);
        unsigned long long life_max = 0;
        int waves = 0;
        for (int w = 0; w < nw; ++w) {
            const unsigned long long *s = &hs[(size_t)w * 16];
            if (!s[0] || !s[12]) continue;
            ++waves;
            life_max = std::max(life_max, s[12] - s[0]);
            // a layout that skips a phase leaves that slot at 0 (the folded walk slots store the forward half inside
            // the mirror pass: no stamp 11): an interval runs from the last stamp PRESENT to the next one present
            unsigned long long prev = s[0];
            for (int k = 1; k <= 12; ++k) {
                if (!s[k] || s[k] < prev) continue;
                d[k].push_back((double)(s[k] - prev));
                prev = s[k];
            }
        }
        // (no global span: s_memtime has a base per XCD; the intervals below are differences within one wavefront)
        printf("waves=%d  longest wavefront life: %llu cycles\n", waves, life_max);
        for (int k = 1; k <= 12; ++k) {
            std::sort(d[k].begin(), d[k].end());
            if (d[k].empty()) continue;
            printf("  %-18s median %7.0f  max %7.0f\n", ph[k], d[k][d[k].size() / 2], d[k].back());
        }
    }
    return 0;
}
