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
#include "topk_merge.hpp"

namespace scl {

namespace {

constexpr unsigned long long kNoKey = kTopkNoKey;
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

// nanoflann's metric of one slot (NF:391-406: four dimensions per step, sequential tail; no FMA).  The loads of up to sixteen ring
// groups are issued before the first of them is used: the accumulation is one dependent chain, and a load per link of it cost the
// one-wave form of this kernel 16 memory latencies (12 us for 2.5 MB).
__device__ __forceinline__ float ringkey_metric(const float4 *rkey4, int cap, int slot, const float *sq, int RGfull, int tail)
{
    float result = 0.0f;
    for (int g0 = 0; g0 < RGfull; g0 += 16) {
        float4 b[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (g0 + j < RGfull) b[j] = rkey4[(size_t)(g0 + j) * cap + slot];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (g0 + j < RGfull) {
                const int g = g0 + j;
                const float d0 = sq[4 * g + 0] - b[j].x;
                const float d1 = sq[4 * g + 1] - b[j].y;
                const float d2 = sq[4 * g + 2] - b[j].z;
                const float d3 = sq[4 * g + 3] - b[j].w;
                result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
        }
    }
    if (tail) {
        const float4 b = rkey4[(size_t)RGfull * cap + slot];
        const float bv[4] = {b.x, b.y, b.z, b.w};
        for (int i = 0; i < tail; ++i) {
            const float d0 = sq[4 * RGfull + i] - bv[i];
            result += d0 * d0;
        }
    }
    return result;
}

__device__ __forceinline__ unsigned long long ringkey_key(float result, int slot, float exclude_eps)
{
    const bool excluded = (exclude_eps > 0.0f) && (result <= exclude_eps);
    if (!excluded && (result < FLT_MAX))
        return ((unsigned long long)(unsigned)__float_as_int(result) << 32) | (unsigned)slot;
    return kNoKey;
}

// The scan for up to 65 536 slots: 256 slots per workgroup, one per thread (39 workgroups at 10k keyframes).  Every thread counts the
// keys of the workgroup that are smaller than its own (256 broadcast reads of LDS, all independent; keys are unique -- the slot is
// their low word) and the keys of rank < k go to position rank of the workgroup's list: sorted, no rounds, whatever k is (round 4: k
// rounds of a block-wide minimum with two barriers each, 1 024 slots per workgroup on ten CUs).  The rows are requested BEFORE the
// query's ring key: one memory round trip instead of two.  The lists are merged by the kernel that consumes the result (topk_merge.hpp).
__global__ __launch_bounds__(kTopkThreads) void ringkey_lists_kernel(
    const float4 *rkey4, int cap, const float *qkey, int R, int lo, int hi, int k, float exclude_eps, unsigned long long *partial)
{
    constexpr int T = kTopkThreads;
    __shared__ unsigned long long skey[T];
    __shared__ float sq[256];
    const int slot = lo + (int)blockIdx.x * T + (int)threadIdx.x;
    const int RGfull = R >> 2, tail = R & 3;
    unsigned long long mine = kNoKey;
    if (RGfull <= 16 && tail == 0) {
        // (the grids in use: 20, 64, 80 rings) every row of the slot in flight, then the query's key
        float4 b[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < RGfull && slot < hi) b[j] = rkey4[(size_t)j * cap + slot];
        for (int r = threadIdx.x; r < R; r += T) sq[r] = qkey[r];
        __syncthreads();
        if (slot < hi) {
            float result = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (j < RGfull) {
                    const float d0 = sq[4 * j + 0] - b[j].x;
                    const float d1 = sq[4 * j + 1] - b[j].y;
                    const float d2 = sq[4 * j + 2] - b[j].z;
                    const float d3 = sq[4 * j + 3] - b[j].w;
                    result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;     // NF:391-397
                }
            }
            mine = ringkey_key(result, slot, exclude_eps);
        }
    } else {
        for (int r = threadIdx.x; r < R; r += T) sq[r] = qkey[r];
        __syncthreads();
        if (slot < hi) mine = ringkey_key(ringkey_metric(rkey4, cap, slot, sq, RGfull, tail), slot, exclude_eps);
    }
    unsigned long long *list = partial + (size_t)blockIdx.x * k;
    if (k <= 8) {
        // the reference's k (3): k rounds of a wave minimum over registers -- a key is a non-negative finite double by its bit pattern
        // (distance bits in the high word: exponent field <= 0x7f7), "no key" = +inf, so the DPP minimum of device_common.hpp applies
        // and nothing goes through LDS (the 256-key rank count below: 2-3 us of LDS reads) -- then the four waves' lists (4k keys) are
        // ranked by the first wave
        const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
        const double inf = __longlong_as_double(0x7ff0000000000000ll);
        double mk = mine == kNoKey ? inf : __longlong_as_double((long long)mine);
        for (int r = 0; r < k; ++r) {
            const double m = wave_min_f64_dpp(mk);
            if (lane == 0) skey[wv * 8 + r] = m == inf ? kNoKey : (unsigned long long)__double_as_longlong(m);
            mk = mk == m ? inf : mk;                       // keys are unique: exactly one lane held it (or none: all inf)
        }
        __syncthreads();
        if (wv == 0) {
            const int nk = (T / kWave) * k;                // <= 32 keys
            const int w2 = lane / k, r2 = lane - w2 * k;
            const unsigned long long x = lane < nk ? skey[w2 * 8 + r2] : kNoKey;
            int rank = 0, valid = 0;
            for (int j = 0; j < nk; ++j) {
                const int wj = j / k;
                const unsigned long long y = skey[wj * 8 + (j - wj * k)];
                rank += y < x ? 1 : 0;
                valid += y != kNoKey ? 1 : 0;
            }
            if (x != kNoKey && rank < k) list[rank] = x;
            if (lane >= valid && lane < k) list[lane] = kNoKey;      // behind the valid keys
        }
        return;
    }
    skey[threadIdx.x] = mine;
    __syncthreads();
    int rank = 0;
#pragma unroll 8
    for (int j = 0; j < T; ++j) rank += skey[j] < mine ? 1 : 0;
    const int valid = __syncthreads_count(mine != kNoKey);
    if (mine != kNoKey && rank < k) list[rank] = mine;
    for (int i = valid + (int)threadIdx.x; i < k; i += T) list[i] = kNoKey;      // behind the valid keys
}

// The bare search's consumer (scl_ringkey_topk, and every path that needs the merged result in device memory before its next launch):
// one workgroup merges the lists and writes idx / d2 -- and, pinned_out != nullptr, the same as the block idx[k] | d2[k] the host reads.
__global__ __launch_bounds__(kTopkThreads) void topk_merge_pack_kernel(const unsigned long long *lists, int L, int k, int *out_idx, float *out_d2, char *pinned_out)
{
    __shared__ unsigned long long skey[kTopkMergeLds];
    __shared__ unsigned long long scand[kTopkMergeLds];
    __shared__ TopkMergeShared sh;
    __shared__ int r_idx[kTopkMaxK];
    __shared__ float r_d2[kTopkMaxK];
    topk_merge_lists(lists, L, k, skey, scand, &sh, r_idx, r_d2);
    for (int i = threadIdx.x; i < k; i += blockDim.x) {
        out_idx[i] = r_idx[i]; out_d2[i] = r_d2[i];
        if (pinned_out) { reinterpret_cast<int *>(pinned_out)[i] = r_idx[i]; reinterpret_cast<float *>(pinned_out + sizeof(int) * k)[i] = r_d2[i]; }
    }
}

// ... more lists than a workgroup holds in LDS (k large on a large range): rounds of block-wide minima over the lists
__global__ __launch_bounds__(kTopkThreads) void topk_merge_rounds_kernel(const unsigned long long *partial, int count, int k, int *out_idx, float *out_d2)
{
    constexpr int T = kTopkThreads;
    __shared__ unsigned long long sred[T / kWave];
    __shared__ unsigned long long srun[kTopkMaxK];
    for (int i = threadIdx.x; i < kTopkMaxK; i += T) srun[i] = kNoKey;
    __syncthreads();
    for (int base = 0; base < count; base += 4 * T) {
        unsigned long long keys[5];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = base + u * T + (int)threadIdx.x;
            keys[u] = p < count ? partial[p] : kNoKey;
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
}

// More than 65 536 slots: T threads x SPT slots per workgroup and chunk, the workgroups stride over the chunks and keep their k best
// by rounds of block-wide minima (ascending, kNoKey behind the valid keys: the same lists).
template <int T, int SPT>
__global__ __launch_bounds__(T) void ringkey_lists_chunked_kernel(
    const float4 *rkey4, int cap, const float *qkey, int R, int lo, int hi, int k, float exclude_eps, unsigned long long *partial)
{
    __shared__ unsigned long long sred[T / kWave];
    __shared__ unsigned long long srun[kTopkMaxK];      // this workgroup's running k best
    __shared__ float sq[256];
    for (int r = threadIdx.x; r < R; r += T) sq[r] = qkey[r];
    for (int i = threadIdx.x; i < kTopkMaxK; i += T) srun[i] = kNoKey;
    __syncthreads();

    constexpr int CHUNK = T * SPT;
    const int nchunks = (hi - lo + CHUNK - 1) / CHUNK;
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        unsigned long long keys[SPT + 1];
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            const int slot = lo + chunk * CHUNK + u * T + threadIdx.x;
            keys[u] = slot < hi ? ringkey_key(ringkey_metric(rkey4, cap, slot, sq, R >> 2, R & 3), slot, exclude_eps) : kNoKey;
        }
        keys[SPT] = (int)threadIdx.x < k ? srun[threadIdx.x] : kNoKey;
        __syncthreads();                     // srun is rewritten by the rounds below
        extract_topk<T>(keys, k, srun, sred);
        __syncthreads();
    }
    for (int i = threadIdx.x; i < k; i += T) partial[(size_t)blockIdx.x * k + i] = srun[i];
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

// the scan: per-workgroup lists in `scratch` (*n_lists of them, k keys each; 0: empty range)
hipError_t launch_ringkey_lists(const DbView &db, const float *qkey, int lo, int hi, int k, float exclude_eps,
                                unsigned long long *scratch, int *n_lists, hipStream_t stream)
{
    if (k <= 0 || k > kTopkMaxK || db.R > 256 || !n_lists) return hipErrorInvalidValue;
    const int n = hi - lo;
    *n_lists = 0;
    if (n <= 0) return hipSuccess;
    if (n <= kTopkThreads * kTopkMaxBlocks) {
        const int blocks = (n + kTopkThreads - 1) / kTopkThreads;
        hipLaunchKernelGGL(ringkey_lists_kernel, dim3(blocks), dim3(kTopkThreads), 0, stream,
                           db.rkey4, db.cap, qkey, db.R, lo, hi, k, exclude_eps, scratch);
        *n_lists = blocks;
    } else {
        int blocks = (n + kTopkChunk - 1) / kTopkChunk;
        if (blocks > kTopkMaxBlocks) blocks = kTopkMaxBlocks;    // workgroups stride over the chunks
        hipLaunchKernelGGL((ringkey_lists_chunked_kernel<kTopkThreads, kTopkPerThread>), dim3(blocks), dim3(kTopkThreads), 0, stream,
                           db.rkey4, db.cap, qkey, db.R, lo, hi, k, exclude_eps, scratch);
        *n_lists = blocks;
    }
    return hipGetLastError();
}

// the lists merged into out_idx / out_d2 (device), and, pinned_out != nullptr, into the block idx[k] | d2[k] there as well
hipError_t launch_topk_merge(const unsigned long long *scratch, int n_lists, int k, int *out_idx, float *out_d2, void *pinned_out, hipStream_t stream)
{
    if (k <= 0 || k > kTopkMaxK || n_lists < 0 || n_lists > kTopkMaxBlocks) return hipErrorInvalidValue;
    if (n_lists * k <= kTopkMergeLds) {
        hipLaunchKernelGGL(topk_merge_pack_kernel, dim3(1), dim3(kTopkThreads), 0, stream, scratch, n_lists, k, out_idx, out_d2, static_cast<char *>(pinned_out));
    } else {
        hipLaunchKernelGGL(topk_merge_rounds_kernel, dim3(1), dim3(kTopkThreads), 0, stream, scratch, n_lists * k, k, out_idx, out_d2);
        if (pinned_out) hipLaunchKernelGGL(topk_pack_kernel, dim3(1), dim3(64), 0, stream, out_idx, out_d2, (const double *)nullptr, (const int *)nullptr, k, 0, static_cast<char *>(pinned_out));
    }
    return hipGetLastError();
}

hipError_t launch_ringkey_topk(const DbView &db, const float *qkey, int lo, int hi, int k,
                               float exclude_eps, unsigned long long *scratch,
                               int *out_idx, float *out_d2, hipStream_t stream)
{
    int L = 0;
    hipError_t e = launch_ringkey_lists(db, qkey, lo, hi, k, exclude_eps, scratch, &L, stream);
    if (e != hipSuccess) return e;
    return launch_topk_merge(scratch, L, k, out_idx, out_d2, nullptr, stream);
}

}  // namespace scl
