// Diagnostic harness for k_zgemm_tri (upper-triangle stream-K second product): builds
// quflow_amd/csrc/zgemm.hip with in-kernel s_memtime stamps (QF_STAMP) and prints where a
// workgroup's life goes: per segment K loop, publish / gather, epilogue.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/tri_probe.hip -o tools/tri_probe
#define QF_STAMP 1
#include "../quflow_amd/csrc/zgemm.hip"
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
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
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    ctx.num_cus = prop.multiProcessorCount;
    if (argc > 2) ctx.sk_min_units = atoi(argv[2]);
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
    ctx.sk_slots = 2 * ctx.num_cus;
    hipMalloc((void **)&ctx.sk_partial, (size_t)ctx.sk_slots * 64 * 64 * sizeof(cplx));
    hipMalloc((void **)&ctx.sk_flags, (size_t)(ctx.sk_slots + 16) * sizeof(unsigned));
    hipMemset(ctx.sk_flags, 0, (size_t)(ctx.sk_slots + 16) * sizeof(unsigned));
    hipMalloc((void **)&ctx.state, sizeof(qf_dev_state));
    hipMemset(ctx.state, 0, sizeof(qf_dev_state));
    hipHostMalloc((void **)&ctx.host_rec, sizeof(qf_host_record), hipHostMallocCoherent);
    memset(ctx.host_rec, 0, sizeof(qf_host_record));
    const int fused = getenv("QF_FUSED") ? 1 : 0;     // with the fused step end's speculative stores + decision
    cplx *W2, *WH2;
    hipMalloc((void **)&W2, NN * sizeof(cplx));
    hipMalloc((void **)&WH2, NN * sizeof(cplx));
    {   // the decision reads minit / maxit / tol from the state
        qf_dev_state hs;
        memset(&hs, 0, sizeof(hs));
        hs.minit = 1; hs.maxit = 1 << 30; hs.tol = 0.0; hs.resnorm = 1e300;
        hipMemcpy(ctx.state, &hs, sizeof(hs), hipMemcpyHostToDevice);
    }
    const int nblocks = ctx.num_cus;
    unsigned long long *stamps;
    hipMalloc((void **)&stamps, (size_t)nblocks * 32 * sizeof(unsigned long long));
    hipMemset(stamps, 0, (size_t)nblocks * 32 * sizeof(unsigned long long));
    if (!getenv("QF_NOSTAMPS")) hipMemcpyToSymbol(HIP_SYMBOL(qf_tri_buf), &stamps, sizeof(stamps));
    qf_epilogue ep;
    ep.PW = A; ep.W = W; ep.dW[0] = D0; ep.dW[1] = D1; ep.Whalf = WH; ep.rowpart = rowpart;
    if (fused) { ep.fused = 1; ep.Wpair[0] = W; ep.Wpair[1] = W2; ep.Whalf_step = WH2; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int variant = fused ? 1 : 0; variant < 2; ++variant)
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, ctx.stream);
            if (variant) qf_launch_zgemm_tri(&ctx, A, B, &ep);
            else qf_launch_zgemm(&ctx, A, B, C, &ep);
            hipEventRecord(e1, ctx.stream);
            hipStreamSynchronize(ctx.stream);
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s rep %d: %.1f us\n", variant ? "tri " : "full", rep, ms * 1e3);
        }
    std::vector<unsigned long long> st((size_t)nblocks * 32);
    hipMemcpy(st.data(), stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    // (s_memtime counts per XCD from a base of its own: stamps are only ever compared within ONE workgroup)
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    auto mx = [](std::vector<double> v) { return v.empty() ? 0.0 : *std::max_element(v.begin(), v.end()); };
    std::vector<double> pro, per_kt, pub, gat, epi, life, ph1, ph2;
    for (int b = 0; b < nblocks; ++b) {
        unsigned long long b0 = 0, b1 = 0;
        for (int sgm = 0; sgm < 4; ++sgm) {
            const unsigned long long *s = &st[((size_t)b * 4 + sgm) * 8];
            if (!s[0]) continue;
            if (!b0) b0 = s[0];
            b1 = s[3];
            const double kt = (double)s[5];
            pro.push_back((double)(s[6] - s[0]));
            per_kt.push_back((double)(s[1] - s[6]) / kt);
            if (s[4] != 0) pub.push_back((double)(s[2] - s[1]));     // k0 != 0: a parked piece
            else {
                gat.push_back((double)(s[2] - s[1]));
                epi.push_back((double)(s[3] - s[2]));
                if (s[7]) { ph1.push_back((double)(s[7] - s[2])); ph2.push_back((double)(s[3] - s[7])); }
            }
        }
        life.push_back((double)(b1 - b0));
    }
    {   // where a workgroup is, relative to ITS OWN first stamp (the XCDs' counters have different bases; the workgroups of a
        // one-per-CU launch start within a few hundred cycles of each other): end of the last K loop, end of the life
        std::vector<double> loop_end, life_end;
        for (int b = 0; b < nblocks; ++b) {
            unsigned long long le = 0, en = 0, st0 = 0;
            for (int sgm = 0; sgm < 4; ++sgm) {
                const unsigned long long *s = &st[((size_t)b * 4 + sgm) * 8];
                if (!s[0]) continue;
                if (!st0) st0 = s[0];
                le = s[1];
                en = s[3];
            }
            if (st0) { loop_end.push_back((double)(le - st0)); life_end.push_back((double)(en - st0)); }
        }
        auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
        printf("last K loop ends : min %.0f median %.0f max %.0f cycles after the workgroup's own start\n", pct(loop_end, 0), pct(loop_end, .5), pct(loop_end, 1));
        printf("life ends        : min %.0f 25%% %.0f median %.0f 75%% %.0f max %.0f\n", pct(life_end, 0), pct(life_end, .25), pct(life_end, .5), pct(life_end, .75), pct(life_end, 1));
    }
    printf("prologue        : median %.0f max %.0f\n", med(pro), mx(pro));
    printf("K-tile          : median %.0f max %.0f\n", med(per_kt), mx(per_kt));
    printf("publish         : median %.0f max %.0f  (n=%zu)\n", med(pub), mx(pub), pub.size());
    printf("fetch+gather    : median %.0f max %.0f  (n=%zu)\n", med(gat), mx(gat), gat.size());
    printf("epilogue        : median %.0f max %.0f\n", med(epi), mx(epi));
    printf("  sums+drain+ticket : median %.0f max %.0f\n", med(ph1), mx(ph1));
    printf("  stores+mirror     : median %.0f max %.0f\n", med(ph2), mx(ph2));
    printf("workgroup life  : median %.0f max %.0f\n", med(life), mx(life));
    for (int b : {0, 1, 2, 3, nblocks / 2, nblocks - 1}) {
        printf("block %d:", b);
        unsigned long long own0 = 0;
        for (int sgm = 0; sgm < 4; ++sgm) {
            const unsigned long long *s = &st[((size_t)b * 4 + sgm) * 8];
            if (!s[0]) continue;
            if (!own0) own0 = s[0];
            printf("  [k0=%llu KT=%llu start+%llu pro %llu loop %llu x %llu end %llu]", s[4], s[5], s[0] - own0, s[6] - s[0], s[1] - s[6],
                   s[2] - s[1], s[3] - s[2]);
        }
        printf("\n");
    }
    return 0;
}
