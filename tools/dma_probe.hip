// LDS-DMA (buffer_load_dwordx4 ... lds) addressing probe: LDS destination = wave-uniform base + 16 lane,
// per-lane global source.  Build: hipcc -O2 --offload-arch=gfx950 -Wno-unused-value tools/dma_probe.hip -o tools/dmaprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned char* src, unsigned* out, int nbytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(src), 0, nbytes, 0x00020000);
    // each wave: 2 instructions; instr i of wave w writes LDS [ (w*2+i)*1024 .. +1024 ) with lane-reversed source
    for (int i = 0; i < 2; ++i) {
        unsigned voff = (unsigned)(((wave * 2 + i) * 64 + (63 - lane)) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (wave * 2 + i) * 1024), 16, voff, 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int j = threadIdx.x; j < 2048; j += blockDim.x) out[j] = reinterpret_cast<unsigned*>(smem)[j];
}
int main()
{
    const int n = 8192;
    std::vector<unsigned> h(n / 4);
    for (int i = 0; i < n / 4; ++i) h[i] = i;
    unsigned char* d; unsigned* o;
    hipMalloc(&d, n); hipMalloc(&o, n);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 8192, 0, d, o, n);
    std::vector<unsigned> r(n / 4);
    hipMemcpy(r.data(), o, n, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 8; ++w) for (int l = 0; l < 64; ++l) for (int q = 0; q < 4; ++q) {
        unsigned want = ((w * 64 + (63 - l)) * 4 + q);
        if (r[(w * 64 + l) * 4 + q] != want) { if (bad < 5) printf("mismatch w%d l%d q%d got %u want %u\n", w, l, q, r[(w*64+l)*4+q], want); ++bad; }
    }
    printf("dma probe: %s (%d bad) err=%s\n", bad ? "FAIL" : "OK", bad, hipGetErrorString(hipGetLastError()));
    return 0;
}
