// Phase-B inner loop of the SC-distance wave kernel in isolation (gfx950): per query ring,
// W+1 = 14 fp64 window values per lane from LDS (7 x ds_read_b128) feed 26 fp64 fmas + 2 cvt.
// How many cycles per ring does one wave need, alone and with a partner on the SIMD, as a
// function of how far ahead the window is requested?  s_memtime (core clock).
//   MODE 0: fmas only (window loaded once)         MODE 1: window of ring r+1 requested before ring r's fmas
//   MODE 2: window of ring r+2 requested (2 ahead) MODE 3: as 1, window as 14 x ds_read_b64
//   MODE 4: as 1, but the 7 reads interleaved with the fmas (sched_group_barrier: 4 VALU, 1 DS read, ...)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int W = 13, NQ = 7, QS = 134, ROWS = 64;

__device__ __forceinline__ void pin_ring(double (&a)[13], double (&b)[13])
{
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                      "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]),
                      "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]),
                      "+v"(b[7]), "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12])
                 :: "memory");
}

template <int MODE>
__global__ __launch_bounds__(768) void b_loop(double *out, unsigned long long *cyc, int reps, const float *kin)
{
    extern __shared__ __attribute__((aligned(16))) double Q[];
    for (int i = threadIdx.x; i < (ROWS + 2) * QS; i += blockDim.x) Q[i] = 1.0 + 1e-3 * (i % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int ll = lane < 60 ? lane : 59;
    const double2 *qwin = reinterpret_cast<const double2 *>(Q + 2 * ll);
    double acc0[W], acc1[W];
#pragma unroll
    for (int t = 0; t < W; ++t) { acc0[t] = 0.0; acc1[t] = 0.0; }
    float kf0 = kin[threadIdx.x], kf1 = kin[threadIdx.x + 768];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        const double2 *qp = qwin;
        double2 qn[NQ], qn2[NQ];
#pragma unroll
        for (int v = 0; v < NQ; ++v) qn[v] = qp[v];
        if (MODE == 2) {
#pragma unroll
            for (int v = 0; v < NQ; ++v) qn2[v] = qp[QS / 2 + v];
        }
#pragma unroll 4
        for (int r = 0; r < ROWS; ++r) {
            double q[W + 1];
#pragma unroll
            for (int v = 0; v < NQ; ++v) { q[2 * v] = qn[v].x; q[2 * v + 1] = qn[v].y; }
            pin_ring(acc0, acc1);
            qp += QS / 2;
            if (MODE == 1 || MODE == 4) {
#pragma unroll
                for (int v = 0; v < NQ; ++v) qn[v] = qp[v];
            }
            if (MODE == 2) {
#pragma unroll
                for (int v = 0; v < NQ; ++v) { qn[v] = qn2[v]; qn2[v] = qp[QS / 2 + v]; }
            }
            if (MODE == 3) {
                const double *qd = reinterpret_cast<const double *>(qp);
#pragma unroll
                for (int v = 0; v < NQ; ++v) { qn[v].x = qd[2 * v]; qn[v].y = qd[2 * v + 1]; }
            }
            const double kd0 = (double)kf0, kd1 = (double)kf1;
            kf0 += 1.0f; kf1 += 0.5f;
#pragma unroll
            for (int t = 0; t < W; ++t) {
                acc0[t] = fma(kd0, q[t], acc0[t]);
                acc1[t] = fma(kd1, q[t + 1], acc1[t]);
            }
            if (MODE == 4) {
#pragma unroll
                for (int v = 0; v < NQ; ++v) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int t = 0; t < W; ++t) s += acc0[t] + acc1[t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

// fp32 variant of the same loop: 14-float window (3 x ds_read_b128 + 1 x ds_read_b64), packed fp32 fmas
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(768) void b_loop_f32(float *out, unsigned long long *cyc, int reps, const float *kin)
{
    extern __shared__ __attribute__((aligned(16))) float Qf[];
    constexpr int QSF = 136;                           // floats per row (16-byte aligned rows)
    for (int i = threadIdx.x; i < (ROWS + 2) * QSF; i += blockDim.x) Qf[i] = 1.0f + 1e-3f * (i % 97);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int ll = lane < 60 ? lane : 59;
    const float *qwin = Qf + 4 * (ll >> 1);            // 16-byte aligned (the kernel would keep a second, 2-float-shifted copy for odd lanes)
    f2 acc0[7], acc1[7];                               // shifts (0,1) (2,3) ... (12,13): 14 slots, 13 used
#pragma unroll
    for (int t = 0; t < 7; ++t) { acc0[t] = f2{0.f, 0.f}; acc1[t] = f2{0.f, 0.f}; }
    float kf0 = kin[threadIdx.x], kf1 = kin[threadIdx.x + 768];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < reps; ++rep) {
        const float *qp = qwin;
        f2 qn[8];
#pragma unroll
        for (int v = 0; v < 4; ++v) { const float4 w = *reinterpret_cast<const float4 *>(qp + 4 * v); qn[2 * v] = f2{w.x, w.y}; qn[2 * v + 1] = f2{w.z, w.w}; }
#pragma unroll 4
        for (int r = 0; r < ROWS; ++r) {
            f2 q[8];
#pragma unroll
            for (int v = 0; v < 8; ++v) q[v] = qn[v];
            asm volatile("" ::: "memory");
            qp += QSF;
#pragma unroll
            for (int v = 0; v < 4; ++v) { const float4 w = *reinterpret_cast<const float4 *>(qp + 4 * v); qn[2 * v] = f2{w.x, w.y}; qn[2 * v + 1] = f2{w.z, w.w}; }
            const f2 k0 = f2{kf0, kf0}, k1 = f2{kf1, kf1};
            kf0 += 1.0f; kf1 += 0.5f;
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                acc0[t] = __builtin_elementwise_fma(k0, q[t], acc0[t]);                       // column 0: q[t] = (w[2t], w[2t+1])
                const f2 qs = f2{q[t].y, q[t + 1].x};                                        // column 1 sees the window one later
                acc1[t] = __builtin_elementwise_fma(k1, qs, acc1[t]);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int t = 0; t < 7; ++t) s += acc0[t].x + acc0[t].y + acc1[t].x + acc1[t].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

void run_f32(int threads)
{
    float *out; unsigned long long *cyc; float *kin;
    (void)hipMalloc(&out, 4 * 768 * 256); (void)hipMalloc(&cyc, 8 * 12 * 256); (void)hipMalloc(&kin, 4 * 2048);
    (void)hipMemset(kin, 0, 4 * 2048);
    const int reps = 40;
    (void)hipFuncSetAttribute((const void *)b_loop_f32, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(b_loop_f32, dim3(256), dim3(threads), 100 * 1024, 0, out, cyc, reps, kin);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * threads / 64);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += (double)x; m /= h.size();
    printf("%-44s waves/CU %d : %7.1f cycles per ring per wave\n", "fp32: packed fmas, 16-float window", threads / 64, m / reps / ROWS);
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(kin);
}

template <int MODE>
void run(const char *name, int threads)
{
    double *out; unsigned long long *cyc; float *kin;
    (void)hipMalloc(&out, 8 * 768 * 256); (void)hipMalloc(&cyc, 8 * 12 * 256); (void)hipMalloc(&kin, 4 * 2048);
    (void)hipMemset(kin, 0, 4 * 2048);
    const int reps = 40;
    const size_t lds = 100 * 1024;                      // > half the CU's LDS: one workgroup per CU
    (void)hipFuncSetAttribute((const void *)b_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(b_loop<MODE>, dim3(256), dim3(threads), lds, 0, out, cyc, reps, kin);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * threads / 64);
    (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto x : h) m += (double)x; m /= h.size();
    printf("%-44s waves/CU %d : %7.1f cycles per ring per wave (fma issue alone = 112)\n", name, threads / 64, m / reps / ROWS);
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(kin);
}

int main()
{
    for (int threads : {64, 256, 512, 768}) {
        run<0>("fmas only", threads);
        run<1>("window one ring ahead, 7 x ds_read_b128", threads);
        run<2>("window two rings ahead, 7 x ds_read_b128", threads);
        run<3>("window one ring ahead, 14 x ds_read_b64", threads);
        run<4>("one ring ahead, reads interleaved 4 fma : 1 read", threads);
        run_f32(threads);
    }
    return 0;
}
