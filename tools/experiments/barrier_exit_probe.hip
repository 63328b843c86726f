// Does s_barrier count only the surviving wavefronts of a workgroup?  Half of the waves return, the rest go through
// __syncthreads() several times.  (CDNA ISA, S_BARRIER: "if some waves in the threadgroup have already terminated,
// this waits on only the surviving waves".)  Build: hipcc -O3 --offload-arch=gfx950 barrier_exit_probe.hip -o barrier_exit_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k(int *out)
{
    __shared__ int acc[8];
    const int tid = threadIdx.x;
    if (tid < 8) acc[tid] = 0;
    __syncthreads();
    if (tid >= 256) return;          // waves 4..7 (or 4..15) leave
    for (int r = 0; r < 4; ++r) {
        if ((tid & 63) == 0) atomicAdd(&acc[r], 1);
        __syncthreads();
    }
    if (tid == 0) out[blockIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main()
{
    int *d;
    hipMalloc((void **)&d, 1024 * sizeof(int));
    for (int threads : {512, 1024}) {
        hipMemset(d, 0, 1024 * sizeof(int));
        hipLaunchKernelGGL(k, dim3(1024), dim3(threads), 0, 0, d);
        hipError_t e = hipDeviceSynchronize();
        int h[1024];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) bad += (h[i] != 16);
        printf("threads=%d: %s, %d of 1024 workgroups wrong\n", threads, hipGetErrorString(e), bad);
    }
    return 0;
}
