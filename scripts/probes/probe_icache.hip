// Probe: what straight-line code costs a kernel that runs once.  One workgroup of one wave executes N fused multiply-adds on eight
// independent accumulators, as a loop of 8 (64 bytes of code) or fully unrolled (N x 8 bytes); durations from rocprofv3 --kernel-trace.
//   hipcc --offload-arch=gfx950 -O2 -o scripts/probes/probe_icache scripts/probes/probe_icache.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N, int UNROLL>
__global__ void fma_kernel(float *out, float a, float b)
{
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x + i;
#pragma unroll UNROLL
    for (int i = 0; i < N / 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
}
int main()
{
    float *out; hipMalloc(&out, 1 << 20);
    hipStream_t s; hipStreamCreate(&s);
    for (int r = 0; r < 20; ++r) {
        hipLaunchKernelGGL((fma_kernel<4096, 1>), dim3(1), dim3(64), 0, s, out, 1.0001f, 0.5f);
        hipLaunchKernelGGL((fma_kernel<4096, 512>), dim3(1), dim3(64), 0, s, out, 1.0001f, 0.5f);
        hipLaunchKernelGGL((fma_kernel<16384, 1>), dim3(1), dim3(64), 0, s, out, 1.0001f, 0.5f);
        hipLaunchKernelGGL((fma_kernel<16384, 2048>), dim3(1), dim3(64), 0, s, out, 1.0001f, 0.5f);
        hipStreamSynchronize(s);
    }
    printf("done\n");
    return 0;
}
