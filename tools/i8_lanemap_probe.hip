// Lane map of v_mfma_i32_32x32x32_i8 checked with exact integer data (ozaki.hip's fragment layout).
// Build: hipcc -O2 --offload-arch=gfx950 -Wno-unused-value tools/i8_lanemap_probe.hip -o tools/i8_lanemap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
// A[i][k] = (i==I0 && k==K0), B[k][j] = (k==K0) * (j+1): C[I0][j] = j+1 -> find which lane/byte holds A[I0][K0]
__global__ void k(const signed char* A, const signed char* B, int* C) {
    // hypothesised map: lane l (r=l&31,h=l>>5) holds A[r][16h+j], B[16h+j][r], j=0..15
    int l = threadIdx.x, r = l & 31, h = l >> 5;
    signed char a[16], b[16];
    for (int j = 0; j < 16; ++j) { a[j] = A[r*32 + 16*h + j]; b[j] = B[(16*h + j)*32 + r]; }
    v4i av = *reinterpret_cast<v4i*>(a), bv = *reinterpret_cast<v4i*>(b);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) {
        int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
        C[row*32 + col] = c[reg];
    }
}
int main() {
    signed char hA[1024], hB[1024]; int hC[1024], ref[1024];
    unsigned s = 12345;
    for (int i = 0; i < 1024; ++i) { s = s*1664525u+1013904223u; hA[i] = (signed char)((s>>16)%129 - 64); s = s*1664525u+1013904223u; hB[i] = (signed char)((s>>16)%129 - 64); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int t = 0; for (int kk = 0; kk < 32; ++kk) t += (int)hA[i*32+kk]*(int)hB[kk*32+j]; ref[i*32+j] = t; }
    signed char *dA, *dB; int* dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += hC[i] != ref[i];
    printf("i8 32x32x32 map check: %d mismatches of 1024\n", bad);
    return bad != 0;
}
