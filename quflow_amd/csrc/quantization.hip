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
//   k_block_matvec : 32 rows of B_m per workgroup (8 per wave), x_m staged through LDS in 256-column chunks,
//                    row-coalesced 512-byte loads of B, wavefront-shuffle reductions;
//   k_block_vecmat : 64 columns of B_m per workgroup, the four waves split the rows, B read in
//                    512-byte row segments, d_m broadcast from LDS, partial sums combined in LDS.
#include "qf_internal.h"

#pragma clang fp contract(off)   // the table of the direct Laplacian follows the reference's op order

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


// ---- the quantization basis itself (compute_basis, quantization.py:68-113) on the device.
// Block m of the basis holds the eigenvectors of the (N-m)x(N-m) tridiagonal block T_m of the
// direct Laplacian (quflow/laplacian/direct.py:19-62), column j <-> el = m + j, scaled to norm
// sqrt(N) and oriented by adjust_basis_orientation_ (quantization.py:45-65).  The reference calls
// LAPACK (scipy.linalg.eigh_tridiagonal).  Here the spectrum is known in closed form -- the
// Hoppe-Yau Laplacian has the eigenvalues -el(el+1) exactly -- so every eigenvector is ONE
// twisted factorisation of T_m - lambda I (the getvec step of MRRR, Parlett & Dhillon):
//     forward   D+_k = (d_k - lambda) - e_k^2 / D+_{k-1}
//     backward  D-_k = (d_k - lambda) - e_{k+1}^2 / D-_{k+1}
//     gamma_k = D+_k + D-_k - (d_k - lambda),  twist r = argmin |gamma_k|,  z_r = 1,
//     z_k = -(e_{k+1}/D+_k) z_{k+1} (k < r),   z_k = -(e_k/D-_k) z_{k-1} (k > r),
// repeated once with the Rayleigh-corrected shift lambda + gamma_r/|z|^2 (the matrix entries carry
// rounding errors, so its eigenvalues sit ~eps |T| off the integers).  One thread per eigenvector,
// threads of a block = consecutive columns j: every access B[k*n + j] is coalesced; the vector's
// own storage doubles as the scratch for D+ / D-.
__device__ __forceinline__ double direct_lap_diag(int N, int m, int k)
{
    const double s = (N - 1) / 2.0;
    const double m2 = -s + k, m1 = m2 + m;
    const double c = 2 * (s * (s + 1) - m1 * m2);                 // direct.py:40
    return fabs(c) > 1e-10 ? -c : 0.0;
}
__device__ __forceinline__ double direct_lap_off(int N, int m, int k)   // couples k-1 and k (k >= 1)
{
    const double s = (N - 1) / 2.0;
    const double a2 = -s + (k - 1), a1 = a2 + m;
    const double c = -sqrt(s * (s + 1) - a1 * (a1 + 1)) * sqrt(s * (s + 1) - a2 * (a2 + 1));   // direct.py:50
    return fabs(c) > 1e-10 ? -c : 0.0;
}

__global__ __launch_bounds__(256) void k_basis(int N, double *__restrict__ basis)
{
    const int m = blockIdx.y;
    const int n = N - m;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    double *B = basis + basis_offset(m, N) + j;     // entry k of this vector: B[k*n]
    const size_t nn = (size_t)n;
    const long long el = m + j;
    double lambda = -(double)(el * (el + 1));
    // a pivot this small is replaced (LAPACK's pivmin idea): |T| ~ N^2/2, so 1e-30 |T| is far below
    // rounding and e^2/pivmin stays finite
    const double pivmin = 1e-30 * (1.0 + 0.5 * (double)N * (double)N);
#define QF_PIV(x_) if (fabs(x_) < pivmin) x_ = ((x_) < 0.0 ? -pivmin : pivmin)
    double znorm2 = 1.0;
    for (int pass = 0; pass < 2; ++pass) {
        // (1) forward: D+ into the slots
        double dp = direct_lap_diag(N, m, 0) - lambda;
        QF_PIV(dp);
        B[0] = dp;
        for (int k = 1; k < n; ++k) {
            const double e = direct_lap_off(N, m, k);
            dp = (direct_lap_diag(N, m, k) - lambda) - (e / dp) * e;
            QF_PIV(dp);
            B[k * nn] = dp;
        }
        // (2) backward: twist index
        double dm = direct_lap_diag(N, m, n - 1) - lambda;
        QF_PIV(dm);
        int r = n - 1;
        double gam = dp + dm - (direct_lap_diag(N, m, n - 1) - lambda);
        double gbest = fabs(gam);
        for (int k = n - 2; k >= 0; --k) {
            const double e = direct_lap_off(N, m, k + 1);
            const double dk = direct_lap_diag(N, m, k) - lambda;
            dm = dk - (e / dm) * e;
            QF_PIV(dm);
            const double g = B[k * nn] + dm - dk;
            if (fabs(g) < gbest) {
                gbest = fabs(g);
                gam = g;
                r = k;
            }
        }
        // (3) backward again, D- into the slots above the twist
        dm = direct_lap_diag(N, m, n - 1) - lambda;
        QF_PIV(dm);
        if (n - 1 > r) B[(n - 1) * nn] = dm;
        for (int k = n - 2; k > r; --k) {
            const double e = direct_lap_off(N, m, k + 1);
            dm = (direct_lap_diag(N, m, k) - lambda) - (e / dm) * e;
            QF_PIV(dm);
            B[k * nn] = dm;
        }
        // (4) the vector
        znorm2 = 1.0;
        double z = 1.0;
        for (int k = r - 1; k >= 0; --k) {
            z = -(direct_lap_off(N, m, k + 1) / B[k * nn]) * z;
            B[k * nn] = z;
            znorm2 += z * z;
        }
        z = 1.0;
        for (int k = r + 1; k < n; ++k) {
            z = -(direct_lap_off(N, m, k) / B[k * nn]) * z;
            B[k * nn] = z;
            znorm2 += z * z;
        }
        B[r * nn] = 1.0;
        if (pass == 0) lambda += gam / znorm2;
    }
#undef QF_PIV
    // w2 *= sqrt(N) on unit vectors, then the orientation rule (quantization.py:45-65)
    const double par = (m % 2 == 1) ? -1.0 : 1.0;
    double mult = par;
    const double val = B[(n - 1) * nn];
    if (val < 0.0) {
        mult = -par;
    } else if (val == 0.0) {
        for (int jj = 2; jj < n; ++jj) {
            const double a = B[(n - jj) * nn], b = B[(n - jj - 1) * nn];
            if (fabs(a) > 1e-16 && fabs(b) > 1e-16) {
                const double this_sign = a > 0.0 ? 1.0 : -1.0, prev_sign = b > 0.0 ? 1.0 : -1.0;
                if (this_sign * prev_sign == -1.0) mult = this_sign * par * ((jj % 2 == 0) ? -1.0 : 1.0);
                else mult = this_sign * par;
                break;
            }
        }
    }
    const double scale = mult * (sqrt((double)N) / sqrt(znorm2));
    for (int k = 0; k < n; ++k) B[k * nn] *= scale;
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
    constexpr int RW = 8;                      // rows per wave: 8 x 4 independent 512-byte row loads in flight
    constexpr int ROWS = 4 * RW, CH = 256, NV = (MODE == MODE_SHC) ? 2 : 1;
    __shared__ cplx xs[NV][CH];
    const int m = blockIdx.y;
    const int n = N - m;
    const int row0 = blockIdx.x * ROWS;
    if (m >= Nmax || row0 >= n) return;
    const int J = Nmax - m;                   // columns that carry coefficients
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double *B = basis + basis_offset(m, N);
    const size_t xo = mmajor_offset(m, N);
    double acc[RW][NV][2];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[r][v][0] = acc[r][v][1] = 0.0;
    // rows of this wave (clamped: a row past the block's end re-reads the last one and is dropped)
    const double *Brow[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        int i = row0 + wave * RW + r;
        if (i > n - 1) i = n - 1;
        Brow[r] = B + (size_t)i * n;
    }
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
        for (int t = 0; t < CH / 64; ++t) {
            const int jj = lane + 64 * t;
            if (c0 + jj < J) {
                double b[RW];
#pragma unroll
                for (int r = 0; r < RW; ++r) b[r] = Brow[r][c0 + jj];
                cplx x[NV];
#pragma unroll
                for (int v = 0; v < NV; ++v) x[v] = xs[v][jj];
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        acc[r][v][0] += b[r] * x[v].x;
                        acc[r][v][1] += b[r] * x[v].y;
                    }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int i = row0 + wave * RW + r;
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
    dim3 gm((N + 31) / 32, Nmax);   // 32 rows per workgroup
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

int qf_launch_basis(qf_ctx *ctx, double *basis_dev)
{
    const int N = ctx->N;
    dim3 grid((N + 255) / 256, N);
    hipLaunchKernelGGL(k_basis, grid, dim3(256), 0, ctx->stream, N, basis_dev);
    QF_HIP(hipGetLastError());
    return QF_OK;
}

int qf_launch_shr2mat(qf_ctx *ctx, int Nmax, const double *omega_dev, cplx *W_dev) { return forward<MODE_SHR>(ctx, Nmax, omega_dev, W_dev); }
int qf_launch_shc2mat(qf_ctx *ctx, const double *omega_dev, cplx *W_dev) { return forward<MODE_SHC>(ctx, ctx->N, omega_dev, W_dev); }
int qf_launch_mat2shr(qf_ctx *ctx, int Nmax, const cplx *W_dev, double *omega_dev) { return backward<MODE_SHR>(ctx, Nmax, W_dev, omega_dev); }
int qf_launch_mat2shc(qf_ctx *ctx, const cplx *W_dev, double *omega_dev) { return backward<MODE_SHC>(ctx, ctx->N, W_dev, omega_dev); }
