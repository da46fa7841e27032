// Dependent-issue latency vs throughput of the fp64 VALU operations the SC-distance kernel is
// made of (gfx950): how many cycles a wave spends per instruction when the instructions form
// 1, 2 or 4 independent chains, one and two waves per SIMD.  Cycles from s_memtime (core clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double quot_core(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    const double q = a * y;
    const double r = fma(-b, q, a);
    return fma(r, y, q);
}

// OP: 0 fma, 1 add, 2 rcp, 3 compiler division, 4 quot_core, 5 sub+mul+add (alignment step), 6 readlane-fed add
template <int OP, int CHAINS>
__global__ void lat_kernel(double *out, unsigned long long *cyc, int iters)
{
    double x[CHAINS];
    for (int i = 0; i < CHAINS; ++i) x[i] = 1.0 + 1e-3 * (threadIdx.x + 7 * i);
    const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-7 * (threadIdx.x + 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / CHAINS; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) x[c] = fma(x[c], a, b);
                if (OP == 1) x[c] = x[c] + b;
                if (OP == 2) x[c] = __builtin_amdgcn_rcp(x[c]);
                if (OP == 3) x[c] = a / x[c];
                if (OP == 4) x[c] = quot_core(a, x[c]);
                if (OP == 5) { const double d = a - x[c]; x[c] = x[c] + d * d; }
                if (OP == 6) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(a), (u * CHAINS + c) & 63);
                    const int hi = __builtin_amdgcn_readlane(__double2hiint(a), (u * CHAINS + c) & 63);
                    x[c] = x[c] + __hiloint2double(hi, lo);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < CHAINS; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int CHAINS>
void run(const char *name, int threads)
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&cyc, 8 * 16 * 256);
    const int iters = 500;
    hipLaunchKernelGGL((lat_kernel<OP, CHAINS>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += (double)x; m /= h.size();
    printf("%-26s chains %d  waves/SIMD %d : %7.2f cycles per op per wave\n", name, CHAINS, threads / 256, m / iters / 16);
    hipFree(out); hipFree(cyc);
}

template <int OP>
void run_all(const char *name)
{
    for (int threads : {256, 512}) {
        run<OP, 1>(name, threads);
        run<OP, 2>(name, threads);
        run<OP, 4>(name, threads);
        run<OP, 8>(name, threads);
    }
}

int main()
{
    run_all<0>("v_fma_f64");
    run_all<1>("v_add_f64");
    run_all<2>("v_rcp_f64");
    run_all<3>("a / b (compiler)");
    run_all<4>("quot_core (8 ops)");
    run_all<5>("sub, mul, add (3 ops)");
    run_all<6>("2 readlane + add");
    return 0;
}
