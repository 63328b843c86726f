// Timing-only harness of the two product kernels exactly as the library builds them (no in-kernel
// stamps): A/B of compile-time variants (-DQF_...=...) and of the timing-only ablation knobs
// (-DQF_ABL_NOSTORE=1 ...: results wrong, only the time matters).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 [-D...] tools/gemm_time.hip -o tools/gemm_time
// Run:   tools/gemm_time [N] -> us per launch of the first product, the full second product with
//        epilogue, and the upper-triangle second product without / with the fused step end
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
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    qf_ctx ctx;
    ctx.N = N;
    hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    ctx.num_cus = prop.multiProcessorCount;
    const size_t NN = (size_t)N * N;
    std::vector<double> h(2 * NN);
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    cplx *A, *B, *C, *W, *D0, *D1, *WH, *W2, *WH2;
    for (cplx **p : {&A, &B, &C, &W, &D0, &D1, &WH, &W2, &WH2}) {
        hipMalloc((void **)p, NN * sizeof(cplx));
        for (auto &x : h) x = nd(rng);
        hipMemcpy(*p, h.data(), NN * sizeof(cplx), hipMemcpyHostToDevice);
    }
    double *rowpart;
    hipMalloc((void **)&rowpart, (size_t)64 * N * sizeof(double));
    ctx.rowpart = rowpart;
    ctx.sk_slots = 2 * ctx.num_cus;
    hipMalloc((void **)&ctx.sk_partial, (size_t)ctx.sk_slots * 64 * 64 * sizeof(cplx));
    hipMalloc((void **)&ctx.sk_flags, (size_t)(ctx.sk_slots + 16) * sizeof(unsigned));
    hipMemset(ctx.sk_flags, 0, (size_t)(ctx.sk_slots + 16) * sizeof(unsigned));
    hipMalloc((void **)&ctx.ticket, 512 * sizeof(unsigned));
    hipMemset(ctx.ticket, 0, 512 * sizeof(unsigned));
    hipMalloc((void **)&ctx.state, sizeof(qf_dev_state));
    {
        qf_dev_state hs;
        memset(&hs, 0, sizeof(hs));
        hs.minit = 1; hs.maxit = 1 << 30; hs.tol = 0.0; hs.resnorm = 1e300;
        hipMemcpy(ctx.state, &hs, sizeof(hs), hipMemcpyHostToDevice);
    }
    hipHostMalloc((void **)&ctx.host_rec, sizeof(qf_host_record), hipHostMallocCoherent);
    memset(ctx.host_rec, 0, sizeof(qf_host_record));
    qf_epilogue ep;
    ep.PW = A; ep.W = W; ep.dW[0] = D0; ep.dW[1] = D1; ep.Whalf = WH; ep.rowpart = rowpart;
    qf_epilogue epf = ep;
    epf.fused = 1; epf.Wpair[0] = W; epf.Wpair[1] = W2; epf.Whalf_step = WH2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char *name, auto launch) {
        for (int i = 0; i < 30; ++i) launch();
        hipStreamSynchronize(ctx.stream);
        double best = 1e30, sum = 0;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0, ctx.stream);
            for (int i = 0; i < reps; ++i) launch();
            hipEventRecord(e1, ctx.stream);
            hipStreamSynchronize(ctx.stream);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            best = std::min(best, (double)ms * 1e3 / reps);
            sum += (double)ms * 1e3 / reps;
        }
        printf("%-44s %8.2f us per launch (best of 5 batches; mean %.2f; back-to-back period incl. launch gap)\n", name, best, sum / 5);
    };
    // clock warm-up
    for (int i = 0; i < 2000; ++i) qf_launch_zgemm(&ctx, A, B, C, nullptr);
    hipStreamSynchronize(ctx.stream);
    run("first product (plain store)", [&] { qf_launch_zgemm(&ctx, A, B, C, nullptr); });
    ctx.gemm_tri = false;
    run("second product, full + epilogue", [&] { qf_launch_zgemm(&ctx, A, B, nullptr, &ep); });
    run("second product, full + fused step end", [&] { qf_launch_zgemm(&ctx, A, B, nullptr, &epf); });
    if (N % 64 == 0) {
        run("second product, triangle + epilogue", [&] { qf_launch_zgemm_tri(&ctx, A, B, &ep); });
        run("second product, triangle + fused step end", [&] { qf_launch_zgemm_tri(&ctx, A, B, &epf); });
    }
    if (N % 32 == 0 && N >= 64) {
        const int nt32 = N / 32, ntile = nt32 * (nt32 + 1) / 2;
        hipMalloc((void **)&ctx.t32_partial, (size_t)ntile * 4 * 32 * 32 * sizeof(cplx));
        hipMalloc((void **)&ctx.t32_arrive, (size_t)ntile * sizeof(unsigned));
        hipMemset(ctx.t32_arrive, 0, (size_t)ntile * sizeof(unsigned));
        for (int so = 1; so <= 2; ++so)
            for (int sd = 1; sd <= 2; ++sd) {
                ctx.tri32_split = so;
                ctx.tri32_split_diag = sd;
                char name[96];
                snprintf(name, sizeof name, "second product, tri32 split %d,%d + epilogue", so, sd);
                run(name, [&] { qf_launch_zgemm_tri32(&ctx, A, B, &ep); });
                snprintf(name, sizeof name, "second product, tri32 split %d,%d + fused step end", so, sd);
                run(name, [&] { qf_launch_zgemm_tri32(&ctx, A, B, &epf); });
            }
    }
    if (ctx.host_rec->fault) printf("FAULT flag set\n");
    return 0;
}
