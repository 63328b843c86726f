// Internal declarations shared by the HIP translation units of libquflow_hip.so.
// gfx950 (MI355X) only; no CUDA paths, no CPU fallbacks.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/quflow_hip.h"

typedef double2 cplx;  // interleaved (re, im): numpy complex128 layout

void qf_set_error(const char *fmt, ...);

#define QF_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t _e = (call);                                                        \
        if (_e != hipSuccess) {                                                        \
            qf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return QF_ERR_HIP;                                                         \
        }                                                                              \
    } while (0)

#define QF_TRY(call)              \
    do {                          \
        int _r = (call);          \
        if (_r != QF_OK) return _r; \
    } while (0)

// Factorisation of one tridiagonal coefficient table (data independent):
//   wtab[e]   = a_k / b'_{k-1}   (multiplier of the forward sweep; 0 at a diagonal's head)
//   invtab[e] = 1 / b'_k         (reciprocal pivot)
// e is the flat matrix index of entry (i,j); see poisson.hip.
struct qf_factors {
    double *wtab = nullptr;
    double *invtab = nullptr;
};

struct qf_event_pair {
    hipEvent_t start, stop;
    int kernel_id;
};

struct qf_ctx {
    int N = 0;
    int device = 0;
    hipStream_t stream = nullptr;

    // state and work matrices, each N*N complex128
    cplx *W = nullptr;       // vorticity state
    cplx *dW[2] = {nullptr, nullptr};  // iteration vector, ping-pong (cur / new)
    int dw_cur = 0;
    cplx *Whalf = nullptr;   // W + dW
    cplx *Phalf = nullptr;   // eps * Delta^-1 Whalf
    cplx *PW = nullptr;      // Phalf @ Whalf
    cplx *kahan_c = nullptr; // compensation term (compsum), allocated on demand
    cplx *stage = nullptr;   // staging for host-in/host-out entry points

    double *lap = nullptr;   // (N,N,2) coefficient table of the Poisson problem (bc=True)
    qf_factors poisson;      // its factorisation
    std::map<unsigned long long, qf_factors> user_factors;  // qf_solve_tridiagonal cache
    double *lap_user = nullptr;

    double *rowpart = nullptr;   // [tiles_n][N] partial row sums from the GEMM2 epilogue
    int rowpart_tiles = 0;
    double *rowsum = nullptr;    // [N]
    double *scalars = nullptr;   // small device scratch for reductions (>= 4096 doubles)
    double *host_scalars = nullptr;  // pinned host mirror (>= 16 doubles)

    // measurement
    int profile_mask = 0;
    std::vector<qf_event_pair> events_busy;
    std::vector<qf_event_pair> events_free;
    long long prof_launches[QF_KERNEL_COUNT] = {0};
    double prof_ms[QF_KERNEL_COUNT] = {0};
    hipEvent_t timer_start = nullptr, timer_stop = nullptr;
};

// ---- poisson.hip
int qf_launch_lap_table(qf_ctx *ctx, int bc, double *lap_dev);
int qf_launch_build_factors(qf_ctx *ctx, const double *lap_dev, qf_factors f);
int qf_launch_solve(qf_ctx *ctx, const qf_factors &f, const cplx *W, cplx *P, double scale, int skewh);
int qf_launch_laplace(qf_ctx *ctx, const cplx *P, cplx *W);

// ---- zgemm.hip
struct qf_epilogue {
    // fused epilogue of the second product (isospectral.py:499-509,526-534):
    //   dW_new = C + (PW - PW^H);  Whalf = W + dW_new;  rowpart += |dW_old - dW_new|
    const cplx *PW = nullptr;
    const cplx *W = nullptr;
    const cplx *dW_old = nullptr;
    cplx *dW_new = nullptr;
    cplx *Whalf = nullptr;
    double *rowpart = nullptr;
};
int qf_gemm_tiles_n(int N);
int qf_launch_zgemm(qf_ctx *ctx, const cplx *A, const cplx *B, cplx *C, const qf_epilogue *ep);

// ---- elementwise.hip
int qf_launch_update(qf_ctx *ctx, const cplx *PW, cplx *W, const cplx *dW, cplx *Whalf,
                     cplx *kahan_c, int reinitialize);
int qf_launch_norm_from_rowpart(qf_ctx *ctx, const double *rowpart, int tiles, double *out_dev);
int qf_launch_norm_inf(qf_ctx *ctx, const cplx *A, double *out_dev);
int qf_launch_inner(qf_ctx *ctx, const cplx *A, const cplx *B, double *out_dev);  // sum Re(A conj(B))
