// Issue rate of v_mfma_f64_4x4x4_4b_f64 on gfx950 (cycles per instruction on one SIMD), alone and
// interleaved with v_fma_f64 VALU work, one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>   // 0: mfma only (8 accumulators), 1: valu fma only, 2: 1 mfma + 3 valu fma interleaved, 3: 1 mfma + 1 cvt
__global__ void rate_kernel(double *out, unsigned long long *cyc, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    double acc[8]; for (int i = 0; i < 8; ++i) acc[i] = i;
    double v[6]; for (int i = 0; i < 6; ++i) v[i] = 0.1 * i;
    float f = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0 || MODE == 2 || MODE == 3) acc[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[u], 0, 0, 0);
            if (MODE == 1 || MODE == 2) { v[0] = fma(a, b, v[0]); v[1] = fma(a, b, v[1]); v[2] = fma(a, b, v[2]); }
            if (MODE == 1) { v[3] = fma(a, b, v[3]); }
            if (MODE == 3) { v[4] += (double)f; f += 1.0f; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 8; ++i) s += acc[i]; for (int i = 0; i < 6; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + f;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char *name, int threads, int per_iter_mfma, int per_iter_valu)
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&cyc, 8 * 16 * 256);
    const int iters = 2000;
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += (double)x; m /= h.size();
    printf("%-34s threads/block %4d: %.1f cycles per loop-iteration-of-8 => %.2f cyc per (mfma x%d + valu x%d)\n",
           name, threads, m / iters, m / iters / 8, per_iter_mfma, per_iter_valu);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int threads : {256, 512}) {
        run<0>("mfma_f64_4x4x4 only", threads, 1, 0);
        run<1>("v_fma_f64 only (4 per slot)", threads, 0, 4);
        run<2>("1 mfma + 3 v_fma_f64", threads, 1, 3);
        run<3>("1 mfma + cvt + add", threads, 1, 2);
    }
    return 0;
}
