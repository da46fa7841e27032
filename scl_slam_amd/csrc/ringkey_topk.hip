// ringkey_topk.hip -- K2: exact brute-force k-nearest ring keys.
//
// Replaces both KD-tree searches of the reference: libnabo on the intra path
// (include/descriptor.h:1631-1642) and nanoflann on the inter path
// (descriptor.h:1699-1716).  KD-tree search with eps = 0 is exact, so a full scan
// that reproduces the metric arithmetic returns the same neighbours:
//   - squared L2 in fp32, four dimensions per step, accumulated exactly as
//     nanoflann's L2_Adaptor::evalMetric: result += ((d0*d0 + d1*d1) + d2*d2) + d3*d3
//     then a sequential tail (include/nanoflann.hpp:383-408); no FMA;
//   - acceptance `dist < worstDist` with worstDist starting at FLT_MAX
//     (nanoflann.hpp:158-163, 1360): NaN / +inf / FLT_MAX never enter the set;
//   - result ordered by ascending distance, equal distances by ascending index
//     (what nanoflann's KNNResultSet yields when points are visited in index order).
// The scan is HBM-bound by N*R*4 bytes: the ring-key table is stored tiled
// ([R/4][cap] float4) so lane i reads 16 contiguous bytes of slot lo+i.
//
// Selection: every thread packs (distance bits << 32 | slot) into a u64 (distances are
// >= 0 so the IEEE bits are monotone); each workgroup extracts its k smallest keys by k
// rounds of block-wide min, a second single-workgroup pass merges the partial lists.
#include <float.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace scl {

namespace {

constexpr unsigned long long kNoKey = ~0ull;
constexpr int kTopkThreads = 256;
constexpr int kTopkPerThread = 4;                       // slots per thread per workgroup
constexpr int kTopkChunk = kTopkThreads * kTopkPerThread;

template <int T>
__device__ __forceinline__ unsigned long long block_min_u64(unsigned long long k, unsigned long long *sred)
{
    k = wave_min_u64(k);
    if (T == kWave) return k;                             // one wave: nothing to exchange
    const int wv = threadIdx.x / kWave;
    __syncthreads();
    if ((threadIdx.x & (kWave - 1)) == 0) sred[wv] = k;
    __syncthreads();
    unsigned long long m = sred[0];
#pragma unroll
    for (int w = 1; w < T / kWave; ++w) m = sred[w] < m ? sred[w] : m;
    return m;
}

// k rounds of (block min, remove).  keys[] are this thread's private keys.
template <int T, int NK>
__device__ __forceinline__ void extract_topk(unsigned long long (&keys)[NK], int k,
                                             unsigned long long *out, unsigned long long *sred)
{
    for (int round = 0; round < k; ++round) {
        unsigned long long mine = keys[0];
#pragma unroll
        for (int i = 1; i < NK; ++i) mine = keys[i] < mine ? keys[i] : mine;
        const unsigned long long m = block_min_u64<T>(mine, sred);
        if (threadIdx.x == 0) out[round] = m;
        if (m == kNoKey) {               // exhausted: fill the rest and stop (uniform)
            for (int r2 = round + 1 + (int)threadIdx.x; r2 < k; r2 += T) out[r2] = kNoKey;
            break;
        }
#pragma unroll
        for (int i = 0; i < NK; ++i) keys[i] = (keys[i] == m) ? kNoKey : keys[i];   // slots are unique
    }
}

// ONE launch: T threads x SPT slots per workgroup and chunk; every workgroup keeps its k best (srun) and writes them to `partial`; the
// workgroup that draws the last ticket merges the lists and writes the result (round 4: a second launch of one workgroup, 4.7 us
// behind the 8.9 us of a scan that ran on ten CUs -- 1 024 slots per workgroup).  At 10k keyframes a workgroup is ONE wave with one
// slot per lane (155 workgroups, sixteen 16-byte loads in flight per lane), the k rounds of the selection are wave reductions without
// a barrier, and the merge reads 155 x k keys.
template <int T, int SPT>
__global__ __launch_bounds__(T) void ringkey_dist_topk_kernel(
    const float4 *rkey4, int cap, const float *qkey, int R, int lo, int hi, int k,
    float exclude_eps, unsigned long long *partial, unsigned int *done, int *out_idx, float *out_d2)
{
    __shared__ unsigned long long sred[T / kWave];
    __shared__ unsigned long long srun[kTopkMaxK];      // this workgroup's running k best
    __shared__ float sq[256];
    __shared__ unsigned int s_ticket;
    for (int r = threadIdx.x; r < R; r += T) sq[r] = qkey[r];
    for (int i = threadIdx.x; i < kTopkMaxK; i += T) srun[i] = kNoKey;
    __syncthreads();

    constexpr int CHUNK = T * SPT;
    const int RGfull = R >> 2;          // full groups of four
    const int tail = R & 3;
    const int nchunks = (hi - lo + CHUNK - 1) / CHUNK;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        unsigned long long keys[SPT + 1];
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            const int slot = lo + chunk * CHUNK + u * T + threadIdx.x;
            unsigned long long key = kNoKey;
            if (slot < hi) {
                float result = 0.0f;
                for (int g = 0; g < RGfull; ++g) {
                    const float4 b = rkey4[(size_t)g * cap + slot];
                    const float d0 = sq[4 * g + 0] - b.x;
                    const float d1 = sq[4 * g + 1] - b.y;
                    const float d2 = sq[4 * g + 2] - b.z;
                    const float d3 = sq[4 * g + 3] - b.w;
                    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
                }
                if (tail) {
                    const float4 b = rkey4[(size_t)RGfull * cap + slot];
                    const float bv[4] = {b.x, b.y, b.z, b.w};
                    for (int i = 0; i < tail; ++i) {
                        const float d0 = sq[4 * RGfull + i] - bv[i];
                        result += d0 * d0;
                    }
                }
                const bool excluded = (exclude_eps > 0.0f) && (result <= exclude_eps);
                if (!excluded && (result < FLT_MAX))
                    key = ((unsigned long long)(unsigned)__float_as_int(result) << 32) | (unsigned)slot;
            }
            keys[u] = key;
        }
        keys[SPT] = (int)threadIdx.x < k ? srun[threadIdx.x] : kNoKey;
        __syncthreads();                     // srun is rewritten by the rounds below
        extract_topk<T>(keys, k, srun, sred);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < k; i += T) partial[(size_t)blockIdx.x * k + i] = srun[i];
    __threadfence();                                       // release: this workgroup's list, before its ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(done, 1u);
    __syncthreads();
    if (s_ticket != gridDim.x - 1) return;
    __threadfence();                                       // acquire: the other workgroups' lists
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int i = threadIdx.x; i < kTopkMaxK; i += T) srun[i] = kNoKey;
    __syncthreads();
    const int count = (int)gridDim.x * k;
    for (int base = 0; base < count; base += 4 * T) {
        unsigned long long keys[5];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = base + u * T + (int)threadIdx.x;
            keys[u] = p < count ? __builtin_nontemporal_load(partial + p) : kNoKey;
        }
        keys[4] = (int)threadIdx.x < k ? srun[threadIdx.x] : kNoKey;
        __syncthreads();
        extract_topk<T>(keys, k, srun, sred);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < k; i += T) {
        const unsigned long long m = srun[i];
        if (m == kNoKey) { out_idx[i] = -1; out_d2[i] = FLT_MAX; }
        else { out_idx[i] = (int)(unsigned)(m & 0xffffffffull); out_d2[i] = __int_as_float((int)(m >> 32)); }
    }
    if (threadIdx.x == 0) *done = 0u;                      // armed for the next launch (stream ordered)
}

__global__ void topk_fill_empty_kernel(int k, int *out_idx, float *out_d2)
{
    for (int i = threadIdx.x; i < k; i += blockDim.x) { out_idx[i] = -1; out_d2[i] = FLT_MAX; }
}

// the k results of a top-k (+ the SC distances of those k) into ONE block of pinned host memory, laid out idx[k] | d2[k] | dist[k] |
// shift[k]: one launch instead of four device-to-host copies of a few bytes each (15-20 us of a 55 us detection call)
__global__ void topk_pack_kernel(const int *idx, const float *d2, const double *dist, const int *shift, int k, int have_dist, char *out)
{
    int *o_idx = reinterpret_cast<int *>(out);
    float *o_d2 = reinterpret_cast<float *>(out + sizeof(int) * k);
    double *o_dist = reinterpret_cast<double *>(out + (sizeof(int) + sizeof(float)) * k);
    int *o_shift = reinterpret_cast<int *>(out + (sizeof(int) + sizeof(float) + sizeof(double)) * k);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        o_idx[i] = idx[i]; o_d2[i] = d2[i];
        if (have_dist) { o_dist[i] = dist[i]; o_shift[i] = shift[i]; }
    }
}

}  // namespace

hipError_t launch_topk_pack(const int *idx, const float *d2, const double *dist, const int *shift, int k, bool have_dist, void *pinned_out, hipStream_t stream)
{
    if (k <= 0 || k > kTopkMaxK || !pinned_out) return hipErrorInvalidValue;
    hipLaunchKernelGGL(topk_pack_kernel, dim3(1), dim3(64), 0, stream, idx, d2, dist, shift, k, have_dist ? 1 : 0, static_cast<char *>(pinned_out));
    return hipGetLastError();
}

hipError_t launch_ringkey_topk(const DbView &db, const float *qkey, int lo, int hi, int k,
                               float exclude_eps, unsigned long long *scratch,
                               int *out_idx, float *out_d2, hipStream_t stream)
{
    if (k <= 0 || k > kTopkMaxK || db.R > 256) return hipErrorInvalidValue;
    const int n = hi - lo;
    if (n <= 0) {
        hipLaunchKernelGGL(topk_fill_empty_kernel, dim3(1), dim3(64), 0, stream, k, out_idx, out_d2);
        return hipGetLastError();
    }
    // scratch: kTopkMaxBlocks * kTopkMaxK partial keys, then the ticket counter (zero between launches)
    unsigned long long *partial = scratch;
    unsigned int *done = reinterpret_cast<unsigned int *>(scratch + (size_t)kTopkMaxBlocks * kTopkMaxK);
    if (k <= 4 && n <= kWave * kTopkMaxBlocks) {
        hipLaunchKernelGGL((ringkey_dist_topk_kernel<kWave, 1>), dim3((n + kWave - 1) / kWave), dim3(kWave), 0, stream,
                           db.rkey4, db.cap, qkey, db.R, lo, hi, k, exclude_eps, partial, done, out_idx, out_d2);
    } else if (n <= kTopkThreads * kTopkMaxBlocks) {
        hipLaunchKernelGGL((ringkey_dist_topk_kernel<kTopkThreads, 1>), dim3((n + kTopkThreads - 1) / kTopkThreads), dim3(kTopkThreads), 0, stream,
                           db.rkey4, db.cap, qkey, db.R, lo, hi, k, exclude_eps, partial, done, out_idx, out_d2);
    } else {
        int blocks = (n + kTopkChunk - 1) / kTopkChunk;
        if (blocks > kTopkMaxBlocks) blocks = kTopkMaxBlocks;    // workgroups stride over the chunks
        hipLaunchKernelGGL((ringkey_dist_topk_kernel<kTopkThreads, kTopkPerThread>), dim3(blocks), dim3(kTopkThreads), 0, stream,
                           db.rkey4, db.cap, qkey, db.R, lo, hi, k, exclude_eps, partial, done, out_idx, out_d2);
    }
    return hipGetLastError();
}

}  // namespace scl
