// Diagnostic harness for the int8 digit-split product (quflow_amd/csrc/ozaki.hip): times the
// kernel as built (-DOZ_ABL_* switch parts off; results are wrong then).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=400000 [-DOZ_ABL_NOLOAD=1 ...] tools/oz_probe.hip -o tools/oz_probe
#include "../quflow_amd/csrc/ozaki.hip"
#include <cstdarg>
#include <cstdio>
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
    (void)hipStreamCreate(&ctx.stream);
    const int digits = argc > 2 ? atoi(argv[2]) : 5;
    ctx.oz_digits = digits;
    const size_t ob = qf_oz_operand_bytes(N, digits);
    signed char *pa, *pb;
    double *sa, *sb;
    cplx *C;
    (void)hipMalloc((void **)&pa, ob);
    (void)hipMalloc((void **)&pb, ob);
    (void)hipMalloc((void **)&sa, qf_oz_record_bytes(N, digits));   // row record: scales + digit sums
    (void)hipMemset(sa, 0, qf_oz_record_bytes(N, digits));
    (void)hipMalloc((void **)&sb, qf_oz_record_bytes(N, digits));
    (void)hipMemset(sb, 0, qf_oz_record_bytes(N, digits));
    (void)hipMalloc((void **)&C, (size_t)N * N * sizeof(cplx));
    std::vector<signed char> h(ob);
    unsigned s = 1;
    for (auto &x : h) { s = s * 1664525u + 1013904223u; x = (signed char)((s >> 16) % 128); }   // offset digits in [0, 127]
    (void)hipMemcpy(pa, h.data(), ob, hipMemcpyHostToDevice);
    (void)hipMemcpy(pb, h.data(), ob, hipMemcpyHostToDevice);
    std::vector<double> one(N, 1.0);
    (void)hipMemcpy(sa, one.data(), N * sizeof(double), hipMemcpyHostToDevice);
    (void)hipMemcpy(sb, one.data(), N * sizeof(double), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 10; ++rep) {
        (void)hipEventRecord(e0, ctx.stream);
        qf_launch_oz_gemm(&ctx, pa, sa, pb, sb, C);
        (void)hipEventRecord(e1, ctx.stream);
        (void)hipStreamSynchronize(ctx.stream);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
#if OZ_STAMP
    {
        std::vector<double> st((size_t)(N / 64) * (N / 64) * 16);
        (void)hipMemcpy(st.data(), C, st.size() * sizeof(double), hipMemcpyDeviceToHost);
        double a[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
        const size_t nw = st.size() / 4;
        for (size_t w = 0; w < nw; ++w)
            for (int k = 0; k < 3; ++k) { a[k] += st[4 * w + k]; mx[k] = st[4 * w + k] > mx[k] ? st[4 * w + k] : mx[k]; }
        printf("stamps (shader cycles, mean / max over %zu waves): prologue %.0f / %.0f, K loop %.0f / %.0f (%.0f per K-step), epilogue %.0f / %.0f\n",
               nw, a[0] / nw, mx[0], a[1] / nw, mx[1], a[1] / nw / (N / 32), a[2] / nw, mx[2]);
        if (N == 1024) {      // K-loop ticks of wave 0 of every workgroup, as the 16 x 16 tile map (hundreds of ticks)
            for (int tm = 0; tm < 16; ++tm) {
                for (int tn = 0; tn < 16; ++tn) printf("%4.0f", st[4 * ((size_t)(tm * 16 + tn) * 4) + 1] / 100.0);
                printf("\n");
            }
        }
        printf("  => %.0f cycles in %.1f us: effective clock %.2f GHz (excluding launch)\n", (a[0] + a[1] + a[2]) / nw, best * 1e3,
               (a[0] + a[1] + a[2]) / nw / (best * 1e3) * 1e-3);
    }
#endif
    const double mfma = 1.5 * digits * (digits + 1) * (N / 32) * 32 / 2.4e3;   // us at 2.4 GHz: 45 MFMAs of 32 cycles per K-step
    printf("N=%d digits=%d k_oz_gemm (NOLOAD=%d NOSWAR=%d NOMFMA=%d NOBARRIER=%d NOFRAG=%d): %.1f us  (matrix-pipe floor %.1f us)\n", N, digits, OZ_ABL_NOLOAD,
           OZ_ABL_NOSWAR, OZ_ABL_NOMFMA, OZ_ABL_NOBARRIER, OZ_ABL_NOFRAG, best * 1e3, mfma);
    return 0;
}
