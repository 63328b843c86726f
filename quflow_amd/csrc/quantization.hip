// Spherical-harmonics <-> matrix transforms (quflow/quantization.py:116-396): for every
// diagonal offset m the m-th diagonal of W and the coefficients omega[(el, +-m)], el >= m, are
// related by the dense real (N-m) x (N-m) block B_m of the quantization basis,
//      shr2mat_/shc2mat_ :  diag_m = B_m @ x_m          (quantization.py:188-245, 331-365)
//      mat2shr_/mat2shc_ :  z_m    = diag_m @ B_m       (quantization.py:286-327, 368-396)
// i.e. one sweep over the N^3/3-entry basis (2.9 GB at N=1024) per transform: HBM-bound.
// The basis lives in HBM for the life of the context (qf_basis_upload).
//
// Layouts.  omega is indexed el^2 + el + m (quflow/utils.py:91-105): for fixed m its entries
// are strided by ~2 el, one cache line each.  Small pack/unpack kernels therefore move the
// coefficients to/from an m-major staging array (x_m contiguous, entry j <-> el = m + j, block m
// at offset m N - m (m-1)/2) and apply the real/complex convention factors; the diagonals are
// packed the same way.  The two main kernels then only see contiguous vectors:
//   k_block_matvec : 16 rows of B_m per workgroup, x_m staged through LDS in 256-column chunks,
//                    row-coalesced 512-byte loads of B, wavefront-shuffle reductions;
//   k_block_vecmat : 64 columns of B_m per workgroup, the four waves split the rows, B read in
//                    512-byte row segments, d_m broadcast from LDS, partial sums combined in LDS.
#include "qf_internal.h"

namespace {

__device__ __forceinline__ size_t mmajor_offset(int m, int N) { return (size_t)m * N - (size_t)m * (m - 1) / 2; }
__device__ __forceinline__ size_t basis_offset(int m, int N)
{
    // basis_break_index(m, N), quantization.py:24-42
    const long long a = (long long)m - 1;
    long long ind = a + 2 * a * a - 6 * a * N + 6LL * N * N;
    ind *= 1 + a;
    return (size_t)(ind / 6);
}

constexpr int MODE_SHR = 0, MODE_SHC = 1;

// ---- coefficients -> m-major vectors x (shr: one complex vector per m; shc: two)
// shr (quantization.py:223-241): m = 0: x = omega[el,0];  m > 0: x = (omega[el,m] - i omega[el,-m]) / sqrt(2)
// shc (quantization.py:352-362): x0 = omega[el,m] (lower diagonal), x1 = omega[el,-m] (upper, m != 0)
template <int MODE>
__global__ void k_pack_coeffs(int N, int Nmax, const double *__restrict__ omega, cplx *__restrict__ x0,
                              cplx *__restrict__ x1)
{
    const int m = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= Nmax || j >= Nmax - m) return;
    const long long el = m + j;
    const size_t o = mmajor_offset(m, N) + j;
    if (MODE == MODE_SHR) {
        if (m == 0) {
            x0[o] = make_double2(omega[el * el + el], 0.0);
        } else {
            const double c1dsq2 = 1.0 / sqrt(2.0);
            x0[o] = make_double2(c1dsq2 * omega[el * el + el + m], c1dsq2 * -omega[el * el + el - m]);
        }
    } else {
        const cplx *oc = reinterpret_cast<const cplx *>(omega);
        x0[o] = oc[el * el + el + m];
        if (m != 0) x1[o] = oc[el * el + el - m];
    }
}

// ---- y_m = B_m[:, :J] @ x_m and the assignment to the diagonals of W (followed by W *= 1j)
template <int MODE>
__global__ __launch_bounds__(256) void k_block_matvec(int N, int Nmax, const double *__restrict__ basis,
                                                       const cplx *__restrict__ x0, const cplx *__restrict__ x1,
                                                       cplx *__restrict__ W)
{
    constexpr int ROWS = 16, CH = 256, NV = (MODE == MODE_SHC) ? 2 : 1;
    __shared__ cplx xs[NV][CH];
    const int m = blockIdx.y;
    const int n = N - m;
    const int row0 = blockIdx.x * ROWS;
    if (m >= Nmax || row0 >= n) return;
    const int J = Nmax - m;                   // columns that carry coefficients
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *B = basis + basis_offset(m, N);
    const size_t xo = mmajor_offset(m, N);
    double acc[4][NV][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[r][v][0] = acc[r][v][1] = 0.0;
    for (int c0 = 0; c0 < J; c0 += CH) {
        __syncthreads();
        {
            const int j = c0 + threadIdx.x;
            cplx a = make_double2(0.0, 0.0), b = a;
            if (j < J) {
                a = x0[xo + j];
                if (NV == 2 && m != 0) b = x1[xo + j];
            }
            xs[0][threadIdx.x] = a;
            if (NV == 2) xs[NV - 1][threadIdx.x] = b;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = row0 + wave * 4 + r;
            if (i < n) {
                const double *Brow = B + (size_t)i * n + c0;
#pragma unroll
                for (int t = 0; t < CH / 64; ++t) {
                    const int jj = lane + 64 * t;
                    if (c0 + jj < J) {
                        const double b = Brow[jj];
#pragma unroll
                        for (int v = 0; v < NV; ++v) {
                            const cplx x = xs[v][jj];
                            acc[r][v][0] += b * x.x;
                            acc[r][v][1] += b * x.y;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = row0 + wave * 4 + r;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                double s = acc[r][v][q];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
                acc[r][v][q] = s;
            }
        if (lane == 0 && i < n) {
            const double sgn = (m % 2 == 0) ? 1.0 : -1.0;
            if (MODE == MODE_SHR) {
                // diag_m *= sgn; lower = conj(diag_m), upper = diag_m; W *= 1j   (quantization.py:238-245)
                const double dr = sgn * acc[r][0][0], di = sgn * acc[r][0][1];
                if (m == 0) {
                    W[(size_t)i * N + i] = make_double2(-di, dr);              // 1j * (dr + i di), di == 0
                } else {
                    W[(size_t)i * N + (i + m)] = make_double2(-di, dr);        // 1j * diag_m
                    W[(size_t)(i + m) * N + i] = make_double2(di, dr);         // 1j * conj(diag_m)
                }
            } else {
                // lower: B_m @ omega[el, m]; upper (m != 0): sgn * B_m @ omega[el, -m]; W *= 1j  (:352-365)
                W[(size_t)(i + m) * N + i] = make_double2(-acc[r][0][1], acc[r][0][0]);
                if (m != 0) W[(size_t)i * N + (i + m)] = make_double2(-sgn * acc[r][NV - 1][1], sgn * acc[r][NV - 1][0]);
            }
        }
    }
}

// ---- diagonals of W -> m-major vectors d (lower diagonal m: W[k+m, k]; shc also upper: W[k, k+m])
template <int MODE>
__global__ void k_pack_diags(int N, int Nmax, const cplx *__restrict__ W, cplx *__restrict__ d0, cplx *__restrict__ d1)
{
    const int m = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= Nmax || k >= N - m) return;
    const size_t o = mmajor_offset(m, N) + k;
    d0[o] = W[(size_t)(k + m) * N + k];
    if (MODE == MODE_SHC && m != 0) d1[o] = W[(size_t)k * N + (k + m)];
}

// ---- z_m[j] = sum_k d_m[k] B_m[k, j], j < J
template <int MODE>
__global__ __launch_bounds__(256) void k_block_vecmat(int N, int Nmax, const double *__restrict__ basis,
                                                       const cplx *__restrict__ d0, const cplx *__restrict__ d1,
                                                       cplx *__restrict__ z0, cplx *__restrict__ z1)
{
    constexpr int NV = (MODE == MODE_SHC) ? 2 : 1, KCH = 256;
    __shared__ cplx ds[NV][KCH];
    __shared__ double part[4][NV][2][64];
    const int m = blockIdx.y;
    const int n = N - m;
    const int J = Nmax - m;
    const int j0 = blockIdx.x * 64;
    if (m >= Nmax || j0 >= J) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = j0 + lane;
    const bool live = j < J;
    const double *B = basis + basis_offset(m, N) + (live ? j : j0);
    const size_t dof = mmajor_offset(m, N);
    double acc[NV][2];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v][0] = acc[v][1] = 0.0;
    for (int k0 = 0; k0 < n; k0 += KCH) {
        __syncthreads();
        {
            const int k = k0 + threadIdx.x;
            cplx a = make_double2(0.0, 0.0), b = a;
            if (k < n) {
                a = d0[dof + k];
                if (NV == 2 && m != 0) b = d1[dof + k];
            }
            ds[0][threadIdx.x] = a;
            if (NV == 2) ds[NV - 1][threadIdx.x] = b;
        }
        __syncthreads();
        const int kend = (n - k0 < KCH) ? n - k0 : KCH;
        // the four waves interleave the rows of the chunk; 8 independent row loads in flight per wave
        for (int kk = wave; kk < kend; kk += 32) {
            double b[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = kk + 4 * u;
                b[u] = (k < kend) ? B[(size_t)(k0 + k) * n] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = kk + 4 * u;
                if (k < kend) {
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const cplx d = ds[v][k];
                        acc[v][0] += d.x * b[u];
                        acc[v][1] += d.y * b[u];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        part[wave][v][0][lane] = acc[v][0];
        part[wave][v][1][lane] = acc[v][1];
    }
    __syncthreads();
    if (wave == 0 && live) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const double re = (part[0][v][0][lane] + part[1][v][0][lane]) + (part[2][v][0][lane] + part[3][v][0][lane]);
            const double im = (part[0][v][1][lane] + part[1][v][1][lane]) + (part[2][v][1][lane] + part[3][v][1][lane]);
            (v == 0 ? z0 : z1)[dof + j] = make_double2(re, im);
        }
    }
}

// ---- m-major results z -> omega with the convention factors
// shr (quantization.py:303-327): m = 0: omega[el,0] = Re(z / 1j) = Im z;  m > 0:
//   omega[el,m] = sqrt2*sgn*Im z,  omega[el,-m] = -sqrt2*sgn*Re z;  finally omega /= N
// shc (quantization.py:381-396): omega[el,m] = z0, omega[el,-m] = sgn*z1 (m != 0); omega /= 1j*N
//   (numpy divides by the complex scalar 0 + N i by multiplying with 1/N: out = (Im, -Re) * (1/N))
template <int MODE>
__global__ void k_unpack_coeffs(int N, int Nmax, const cplx *__restrict__ z0, const cplx *__restrict__ z1,
                                double *__restrict__ omega)
{
    const int m = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= Nmax || j >= Nmax - m) return;
    const long long el = m + j;
    const size_t o = mmajor_offset(m, N) + j;
    const double sgn = (m % 2 == 0) ? 1.0 : -1.0;
    const cplx a = z0[o];
    if (MODE == MODE_SHR) {
        if (m == 0) {
            omega[el * el + el] = a.y / (double)N;
        } else {
            const double sqrt2 = sqrt(2.0);
            omega[el * el + el + m] = (sqrt2 * sgn * a.y) / (double)N;
            omega[el * el + el - m] = (-sqrt2 * sgn * a.x) / (double)N;
        }
    } else {
        cplx *oc = reinterpret_cast<cplx *>(omega);
        const double invN = 1.0 / (double)N;
        oc[el * el + el + m] = make_double2(a.y * invN, -a.x * invN);
        if (m != 0) {
            const cplx b = z1[o];
            oc[el * el + el - m] = make_double2((sgn * b.y) * invN, -(sgn * b.x) * invN);
        }
    }
}

template <int MODE>
int forward(qf_ctx *ctx, int Nmax, const double *omega_dev, cplx *W_dev)
{
    const int N = ctx->N;
    cplx *x0 = ctx->sh_stage, *x1 = ctx->sh_stage + (size_t)N * (N + 1) / 2;
    QF_HIP(hipMemsetAsync(W_dev, 0, (size_t)N * N * sizeof(cplx), ctx->stream));   // np.zeros((N,N)), quantization.py:474
    dim3 gp((Nmax + 255) / 256, Nmax);
    hipLaunchKernelGGL(k_pack_coeffs<MODE>, gp, dim3(256), 0, ctx->stream, N, Nmax, omega_dev, x0, x1);
    QF_HIP(hipGetLastError());
    dim3 gm((N + 15) / 16, Nmax);
    hipLaunchKernelGGL(k_block_matvec<MODE>, gm, dim3(256), 0, ctx->stream, N, Nmax, ctx->basis, x0, x1, W_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

template <int MODE>
int backward(qf_ctx *ctx, int Nmax, const cplx *W_dev, double *omega_dev)
{
    const int N = ctx->N;
    const size_t half = (size_t)N * (N + 1) / 2;
    cplx *d0 = ctx->sh_stage, *d1 = d0 + half, *z0 = d1 + half, *z1 = z0 + half;
    dim3 gp((N + 255) / 256, Nmax);
    hipLaunchKernelGGL(k_pack_diags<MODE>, gp, dim3(256), 0, ctx->stream, N, Nmax, W_dev, d0, d1);
    QF_HIP(hipGetLastError());
    dim3 gm((Nmax + 63) / 64, Nmax);
    hipLaunchKernelGGL(k_block_vecmat<MODE>, gm, dim3(256), 0, ctx->stream, N, Nmax, ctx->basis, d0, d1, z0, z1);
    QF_HIP(hipGetLastError());
    dim3 gu((Nmax + 255) / 256, Nmax);
    hipLaunchKernelGGL(k_unpack_coeffs<MODE>, gu, dim3(256), 0, ctx->stream, N, Nmax, z0, z1, omega_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

}  // namespace

int qf_launch_shr2mat(qf_ctx *ctx, int Nmax, const double *omega_dev, cplx *W_dev) { return forward<MODE_SHR>(ctx, Nmax, omega_dev, W_dev); }
int qf_launch_shc2mat(qf_ctx *ctx, const double *omega_dev, cplx *W_dev) { return forward<MODE_SHC>(ctx, ctx->N, omega_dev, W_dev); }
int qf_launch_mat2shr(qf_ctx *ctx, int Nmax, const cplx *W_dev, double *omega_dev) { return backward<MODE_SHR>(ctx, Nmax, W_dev, omega_dev); }
int qf_launch_mat2shc(qf_ctx *ctx, const cplx *W_dev, double *omega_dev) { return backward<MODE_SHC>(ctx, ctx->N, W_dev, omega_dev); }
