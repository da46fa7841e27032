// Probe: the time between the end of a kernel and the start of the next one on the same stream, as rocprofv3's kernel trace shows it,
// for the shapes of the screening launch group: a launch of many small workgroups (the tail: 256 threads, 20 KB of LDS) followed by
// one workgroup per CU with most of the LDS (the products: 1 024 threads, 126 KB).  usage: probe_dispatch_gap <big threads> <big LDS bytes> <big wgs> <rounds> <mode>
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_gap scripts/probes/probe_dispatch_gap.hip
//   rocprofv3 --kernel-trace --output-format csv -d out -o t -- /tmp/probe_gap 1024 129024 256 40
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>

__global__ void small_kernel(int ticks, int *sink, float *out = nullptr, int wmode = 0)
{
    extern __shared__ int lds_s[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    lds_s[threadIdx.x] = (int)t0;
    if (out) {                                                       // 2 480 x 256 floats = 2.5 MB of results, as the tail launch leaves them
        float *p = out + (size_t)blockIdx.x * 256 + threadIdx.x;
        if (wmode == 1) __builtin_nontemporal_store((float)t0, p); else *p = (float)t0;
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    if (lds_s[(threadIdx.x + 1) & 255] == 12345) *sink = 1;
}
__global__ void big_kernel(int ticks, int *sink, float4 *out = nullptr, int wmode = 0)
{
    extern __shared__ int lds_b[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    lds_b[threadIdx.x] = (int)t0;
    if (out) {                                                       // 20 MB of partial sums, as the products leave them
        for (int r = 0; r < 5; ++r) {
            float4 *p = out + ((size_t)blockIdx.x * 5 + r) * blockDim.x + threadIdx.x;
            const float4 v = make_float4((float)t0, 0.f, 0.f, 0.f);
            if (wmode == 1) __builtin_nontemporal_store(v.x, &p->x); else *p = v;
        }
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
    if (lds_b[(threadIdx.x + 1) & 255] == 12345) *sink = 1;
}
int main(int argc, char **argv)
{
    const int bt = argc > 1 ? atoi(argv[1]) : 1024, bl = argc > 2 ? atoi(argv[2]) : 129024, bw = argc > 3 ? atoi(argv[3]) : 256, rounds = argc > 4 ? atoi(argv[4]) : 40;
    int *sink; hipMalloc(&sink, 4);
    float *wsmall; hipMalloc(&wsmall, (size_t)2480 * 256 * 4);
    float4 *wbig; hipMalloc(&wbig, (size_t)256 * 5 * 1024 * 16);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipFuncSetAttribute((const void *)big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bl);
    // mode (5th argument): 0 nothing between the launches; 1 hipEventRecord between small and big; 2 hipStreamWaitEvent on an event of another
    // stream that fired long ago; 3 the event as the small launch's stop event (hipExtLaunchKernelGGL); 4 record + a second stream waiting on it; 5 as 1 with hipEventDisableSystemFence; 6 a wait for an event of another stream that is pending when it is enqueued (fires within microseconds)
    const int mode = argc > 5 ? atoi(argv[5]) : 0;
    hipStream_t s2; hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t ev, ev_old; hipEventCreateWithFlags(&ev, mode == 5 ? (hipEventDisableTiming | hipEventDisableSystemFence) : hipEventDisableTiming); hipEventCreateWithFlags(&ev_old, hipEventDisableTiming);
    hipEvent_t ev6[64]; for (auto &x : ev6) hipEventCreateWithFlags(&x, hipEventDisableTiming);
    hipLaunchKernelGGL(small_kernel, dim3(8), dim3(256), 20480, s2, 100, sink);
    hipEventRecord(ev_old, s2);
    hipStreamSynchronize(s2);
    for (int r = 0; r < rounds; ++r) {
        if (mode == 7 || mode == 8 || mode == 9) {                   // 7: both kernels leave results (plain stores); 8: nontemporal stores; 9: only the big one writes
            hipLaunchKernelGGL(big_kernel, dim3(bw), dim3(bt), bl, s, 3000, sink, wbig, mode == 8 ? 1 : 0);
            hipLaunchKernelGGL(small_kernel, dim3(2480), dim3(256), 20480, s, 300, sink, mode == 9 ? nullptr : wsmall, mode == 8 ? 1 : 0);
            continue;
        }
        hipLaunchKernelGGL(big_kernel, dim3(bw), dim3(bt), bl, s, 3000, sink);      // 30 us (100 MHz ticks)
        if (mode == 3) hipExtLaunchKernelGGL(small_kernel, dim3(2480), dim3(256), 20480, s, nullptr, ev, 0, 300, sink);
        else hipLaunchKernelGGL(small_kernel, dim3(2480), dim3(256), 20480, s, 300, sink);   // 3 us per workgroup; ~10 workgroups per CU
        if (mode == 1 || mode == 4 || mode == 5) hipEventRecord(ev, s);
        if (mode == 6) { hipLaunchKernelGGL(small_kernel, dim3(8), dim3(256), 20480, s2, 100, sink); hipEventRecord(ev6[r & 63], s2); hipStreamWaitEvent(s, ev6[r & 63], 0); }
        if (mode == 2) hipStreamWaitEvent(s, ev_old, 0);
        if (mode == 3 || mode == 4) { hipStreamWaitEvent(s2, ev, 0); hipLaunchKernelGGL(small_kernel, dim3(8), dim3(256), 20480, s2, 100, sink); }
    }
    hipStreamSynchronize(s2);
    hipStreamSynchronize(s);
    printf("done\n");
    return 0;
}
