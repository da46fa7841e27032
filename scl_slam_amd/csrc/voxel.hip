// voxel.hip -- pcl::VoxelGrid down-sampling and submap assembly on the GPU.
//
//   voxel grid        DM.h:996-998 (descriptor input), 1183-1185 (submap), 1200-1201 (service input)
//   submap assembly   loopFindNearKeyframes, DM.h:1163-1186: sum of transformPointCloud(keyframe, pose)
//                     (DM.h:234-253) followed by the voxel filter
//
// Algorithm (PCL's, restated in oracle/icp_oracle.c): voxel coordinates floor(p * 1/leaf) - min_b in fp32,
// linear index with x fastest, one output point per occupied voxel = centroid of x, y, z, intensity, output
// in ascending voxel index.  GPU form: (voxel index, point index) pairs sorted by the 32-bit voxel index with
// rocPRIM's radix sort (through hipCUB; LSD radix sort is stable, so the points of a voxel stay in input
// order), run heads flagged and scanned, then ONE thread per voxel adds its points in that order -- the same
// fp32 sums as the serial restatement, bit for bit.  HBM-bound: n * (stride + 4 * 4) bytes; the sort dominates.

#include <cfloat>
#include <cstring>
#include <vector>

#include "device_common.hpp"
#include "icp.hpp"
#include "device_sort.hpp"

namespace scl {

namespace {

struct VoxState { float mn[3]; float mx[3]; long long minb[3]; long long divb[3]; int nfinite; int overflow; int nout; };

__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    return (fabsf(x) <= FLT_MAX) && (fabsf(y) <= FLT_MAX) && (fabsf(z) <= FLT_MAX);
}

__global__ void vox_bbox_partial_kernel(const unsigned char *pts, int n, int stride, float *part, int *cnt)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    int c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *f = reinterpret_cast<const float *>(pts + (size_t)i * stride);
        const float x = f[0], y = f[1], z = f[2];
        if (!finite3(x, y, z)) continue;
        mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
        mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
        ++c;
    }
    __shared__ float s[6][4];
    __shared__ int sc[4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, kWave));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, kWave));
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, kWave);
    const int wv = threadIdx.x / kWave;
    if ((threadIdx.x & 63) == 0) { for (int a = 0; a < 3; ++a) { s[a][wv] = mn[a]; s[3 + a][wv] = mx[a]; } sc[wv] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 4; ++w) tot += sc[w];
        for (int a = 0; a < 3; ++a) {
            float m = s[a][0], M = s[3 + a][0];
            for (int w = 1; w < 4; ++w) { m = fminf(m, s[a][w]); M = fmaxf(M, s[3 + a][w]); }
            part[blockIdx.x * 6 + a] = m; part[blockIdx.x * 6 + 3 + a] = M;
        }
        cnt[blockIdx.x] = tot;
    }
}

__device__ __forceinline__ void vox_setup_kernel_body(const float *part, const int *cnt, int nblocks, float inv, VoxState *st)
{
    // one wave: lane l folds the partial boxes l, l+64, ..., then a butterfly (min / max / integer sum: order independent)
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    int tot = 0;
    for (int b = threadIdx.x; b < nblocks; b += 64) {
        tot += cnt[b];
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], part[b * 6 + a]); mx[a] = fmaxf(mx[a], part[b * 6 + 3 + a]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tot += __shfl_xor(tot, off, kWave);
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, kWave)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, kWave)); }
    }
    if (threadIdx.x != 0) return;
    st->nfinite = tot; st->overflow = 0; st->nout = 0;
    // PCL: min_b = int(floor(min * inv)), the voxel count dx dy dz against int32's range in 64 bits.  A bound outside int32 (a stray
    // point at 1e12 m, a leaf of 1e-9) makes the cast undefined there and the 64-bit product wrap here: found by UBSan in the checker's
    // copy of this test (make sanitize) -- such a cloud is "too large for the leaf" like any other (the input comes back unfiltered)
    bool big = false;
    for (int a = 0; a < 3; ++a) {
        st->mn[a] = mn[a]; st->mx[a] = mx[a];
        const float lo = tot ? floorf(mn[a] * inv) : 0.f, hi = tot ? floorf(mx[a] * inv) : 0.f;
        const bool in_range = lo >= -2147483648.f && hi <= 2147483520.f;      // (NaN fails)
        big |= !in_range;
        st->minb[a] = in_range ? (long long)lo : 0;
        st->divb[a] = in_range ? (long long)hi - st->minb[a] + 1 : 1;
    }
    if (big || st->divb[0] > 2147483647LL || st->divb[1] > 2147483647LL || st->divb[2] > 2147483647LL) st->overflow = 1;
    else {
        const long long xy = st->divb[0] * st->divb[1];                         // < 2^62
        if (xy > 2147483647LL || xy * st->divb[2] > 2147483647LL) st->overflow = 1;
    }
}

__global__ __launch_bounds__(64) void vox_setup_kernel(const float *part, const int *cnt, int nblocks, float inv, VoxState *st)
{
    vox_setup_kernel_body(part, cnt, nblocks, inv, st);
}

constexpr unsigned kNoVoxel = 0xffffffffu;                             // non-finite points sort to the end

__global__ void vox_keys_kernel(const unsigned char *pts, int n, int stride, float inv, const VoxState *st,
                                unsigned *vox, unsigned *idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *f = reinterpret_cast<const float *>(pts + (size_t)i * stride);
    const float x = f[0], y = f[1], z = f[2];
    unsigned key = kNoVoxel;
    if (finite3(x, y, z) && !st->overflow) {
        const long long i0 = (long long)floorf(x * inv) - st->minb[0];
        const long long i1 = (long long)floorf(y * inv) - st->minb[1];
        const long long i2 = (long long)floorf(z * inv) - st->minb[2];
        key = (unsigned)(i0 + i1 * st->divb[0] + i2 * st->divb[0] * st->divb[1]);   // < 2^31 (overflow is flagged)
    }
    vox[i] = key;
    idx[i] = (unsigned)i;
}

__global__ void vox_heads_kernel(const unsigned *vox, int n, int *head)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned k = vox[i];
    head[i] = (k != kNoVoxel && (i == 0 || vox[i - 1] != k)) ? 1 : 0;
}

__global__ void vox_centroid_kernel(const unsigned char *pts, int stride, const unsigned *vox, const unsigned *idx, const int *head,
                                    const int *pos /*exclusive scan of head*/, int n, unsigned char *out, VoxState *st)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    const unsigned v = vox[i];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    const bool has_i = stride >= 20;
    int b = i;
    while (b < n && vox[b] == v) {
        const float *f = reinterpret_cast<const float *>(pts + (size_t)idx[b] * stride);
        sx += f[0]; sy += f[1]; sz += f[2];
        if (has_i) si += f[4];
        ++b;
    }
    const float cnt = (float)(b - i);
    float *o = reinterpret_cast<float *>(out + (size_t)pos[i] * stride);
    for (int k = 0; k < stride / 4; ++k) o[k] = 0.f;
    o[0] = sx / cnt; o[1] = sy / cnt; o[2] = sz / cnt;
    if (has_i) o[4] = si / cnt;
    if (b >= n || vox[b] == kNoVoxel) st->nout = pos[i] + 1;        // the last head also publishes the output count
}

__global__ void transform_append_kernel(const unsigned char *in, int n, int stride, const float *T, unsigned char *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *f = reinterpret_cast<const float *>(in + (size_t)i * stride);
    float *o = reinterpret_cast<float *>(out + (size_t)i * stride);
    const float x = f[0], y = f[1], z = f[2];
    for (int k = 3; k < stride / 4; ++k) o[k] = f[k];
    // distributedMapping.h:247-249 (fp32, left to right, no FMA)
    o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
    o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
    o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
}

// ---- the submaps of one scan's loop candidates, assembled and filtered TOGETHER --------------------------------------------------
// loopFindNearKeyframes for 26 clouds (the scan's own submap + 25 candidates, BASELINE configs[2]) was 26 chains of ~20 short
// launches, ten of them a radix sort of 100 k pairs each.  Here the raw submaps lie behind each other in one buffer and every step is
// ONE launch over all of them: the sort orders (job << 32 | voxel, point) pairs of all jobs at once -- a stable LSD radix sort keeps a
// voxel's points in input order inside its job, so every centroid is the same sum as in the one-by-one form, bit for bit.
struct VoxPiece { const unsigned char *src; int n; int dst_off; int t_index; };   // one keyframe of one job -> its place in the raw buffer
struct VoxJob { int off, n; };                                                    // a job's range of the raw buffer (and of its output)
constexpr int kVoxMaxJobs = 64;

__global__ void transform_pieces_kernel(const VoxPiece *pieces, const float *T_all, int stride, unsigned char *raw)
{
    const VoxPiece pc = pieces[blockIdx.y];
    const float *T = T_all + 16 * pc.t_index;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < pc.n; i += gridDim.x * blockDim.x) {
        const float *f = reinterpret_cast<const float *>(pc.src + (size_t)i * stride);
        float *o = reinterpret_cast<float *>(raw + (size_t)(pc.dst_off + i) * stride);
        const float x = f[0], y = f[1], z = f[2];
        for (int k = 3; k < stride / 4; ++k) o[k] = f[k];
        // distributedMapping.h:247-249 (fp32, left to right, no FMA)
        o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
        o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
        o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
    }
}

// blockIdx.y = job; partial boxes of the job's raw points at part[(job * 256 + block) * 6], finite counts beside them
__global__ void vox_bbox_partial_batch_kernel(const unsigned char *raw, const VoxJob *jobs, int stride, float *part, int *cnt)
{
    const VoxJob jb = jobs[blockIdx.y];
    const unsigned char *pts = raw + (size_t)jb.off * stride;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    int c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < jb.n; i += gridDim.x * blockDim.x) {
        const float *f = reinterpret_cast<const float *>(pts + (size_t)i * stride);
        const float x = f[0], y = f[1], z = f[2];
        if (!finite3(x, y, z)) continue;
        mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
        mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
        ++c;
    }
    __shared__ float s[6][4];
    __shared__ int sc[4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, kWave));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, kWave));
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, kWave);
    const int wv = threadIdx.x / kWave;
    if ((threadIdx.x & 63) == 0) { for (int a = 0; a < 3; ++a) { s[a][wv] = mn[a]; s[3 + a][wv] = mx[a]; } sc[wv] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int slot = blockIdx.y * gridDim.x + blockIdx.x;
        int tot = 0;
        for (int w = 0; w < 4; ++w) tot += sc[w];
        for (int a = 0; a < 3; ++a) {
            float m = s[a][0], M = s[3 + a][0];
            for (int w = 1; w < 4; ++w) { m = fminf(m, s[a][w]); M = fmaxf(M, s[3 + a][w]); }
            part[slot * 6 + a] = m; part[slot * 6 + 3 + a] = M;
        }
        cnt[slot] = tot;
    }
}

__global__ __launch_bounds__(64) void vox_setup_batch_kernel(const float *part, const int *cnt, int nblocks, float inv, VoxState *st)
{
    vox_setup_kernel_body(part + (size_t)blockIdx.x * nblocks * 6, cnt + (size_t)blockIdx.x * nblocks, nblocks, inv, st + blockIdx.x);
}

// one thread per raw point of any job: the job of point i by bisection of the jobs' offsets (in LDS)
__global__ void vox_keys_batch_kernel(const unsigned char *raw, int n_total, const VoxJob *jobs, int n_jobs, int stride, float inv, const VoxState *st,
                                      unsigned long long *keys, unsigned *idx)
{
    __shared__ int s_off[kVoxMaxJobs + 1];
    if ((int)threadIdx.x <= n_jobs) s_off[threadIdx.x] = (int)threadIdx.x < n_jobs ? jobs[threadIdx.x].off : n_total;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_total) return;
    int a = 0, b = n_jobs;                                             // the last job whose offset is <= i (empty jobs share an offset with the next)
    while (b - a > 1) { const int m = (a + b) >> 1; if (s_off[m] <= i) a = m; else b = m; }
    const VoxState *js = st + a;
    const float *f = reinterpret_cast<const float *>(raw + (size_t)i * stride);
    const float x = f[0], y = f[1], z = f[2];
    unsigned key = kNoVoxel;
    if (finite3(x, y, z) && !js->overflow) {
        const long long i0 = (long long)floorf(x * inv) - js->minb[0];
        const long long i1 = (long long)floorf(y * inv) - js->minb[1];
        const long long i2 = (long long)floorf(z * inv) - js->minb[2];
        key = (unsigned)(i0 + i1 * js->divb[0] + i2 * js->divb[0] * js->divb[1]);   // < 2^31 (overflow is flagged)
    }
    keys[i] = ((unsigned long long)(unsigned)a << 32) | key;
    idx[i] = (unsigned)i;
}

__global__ void vox_heads_batch_kernel(const unsigned long long *keys, int n, int *head)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    head[i] = ((unsigned)k != kNoVoxel && (i == 0 || keys[i - 1] != k)) ? 1 : 0;
}

// one thread per voxel: the centroid of its points in input order, written to the job's own output range [off, off + nout)
__global__ void vox_centroid_batch_kernel(const unsigned char *raw, int stride, const unsigned long long *keys, const unsigned *idx, const int *head,
                                          const int *pos /*exclusive scan of head over all jobs*/, int n, const VoxJob *jobs, unsigned char *out, VoxState *st)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !head[i]) return;
    const unsigned long long v = keys[i];
    const int job = (int)(v >> 32);
    const VoxJob jb = jobs[job];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    const bool has_i = stride >= 20;
    int b = i;
    while (b < n && keys[b] == v) {
        const float *f = reinterpret_cast<const float *>(raw + (size_t)idx[b] * stride);
        sx += f[0]; sy += f[1]; sz += f[2];
        if (has_i) si += f[4];
        ++b;
    }
    const float cnt = (float)(b - i);
    const int local = pos[i] - pos[jb.off];                             // (the job's first sorted element is its first head, or the job has no head at all)
    float *o = reinterpret_cast<float *>(out + (size_t)(jb.off + local) * stride);
    for (int k = 0; k < stride / 4; ++k) o[k] = 0.f;
    o[0] = sx / cnt; o[1] = sy / cnt; o[2] = sz / cnt;
    if (has_i) o[4] = si / cnt;
    if (b >= n || (int)(keys[b] >> 32) != job || (unsigned)keys[b] == kNoVoxel) st[job].nout = local + 1;   // the job's last head publishes its count
}

enum VBuf { V_IN = 0, V_KEYS, V_KEYS2, V_HEAD, V_POS, V_TMP, V_OUT, V_STATE, V_PART, V_T };

#define VOX_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) { if (err) *err = std::string(#call) + ": " + hipGetErrorString(e__); return SCL_ERR_HIP; } \
    } while (0)

int vensure(IcpWorkspace *ws, int k, size_t bytes, std::string *err)
{
    if (bytes <= ws->cap[k]) return SCL_OK;
    if (ws->buf[k]) { (void)hipFree(ws->buf[k]); ws->buf[k] = nullptr; ws->cap[k] = 0; }
    size_t nb = bytes + bytes / 4 + 256;
    if (hipMalloc(&ws->buf[k], nb) != hipSuccess) { if (err) *err = "voxel: hipMalloc failed"; return SCL_ERR_NOMEM; }
    ws->cap[k] = nb;
    return SCL_OK;
}

// input already in ws->buf[V_IN]; output left in ws->buf[V_OUT]; *n_out on host
int voxel_device(IcpWorkspace *ws, hipStream_t stream, int n, int stride, float leaf, int *n_out, std::string *err)
{
    int rc;
    *n_out = 0;
    if (n == 0) return SCL_OK;
    if ((rc = vensure(ws, V_KEYS, sizeof(unsigned long long) * (size_t)n, err))) return rc;
    if ((rc = vensure(ws, V_KEYS2, sizeof(unsigned long long) * (size_t)n, err))) return rc;
    if ((rc = vensure(ws, V_HEAD, sizeof(int) * (size_t)n, err))) return rc;
    if ((rc = vensure(ws, V_POS, sizeof(int) * (size_t)n, err))) return rc;
    if ((rc = vensure(ws, V_OUT, (size_t)n * stride + 16, err))) return rc;
    if ((rc = vensure(ws, V_STATE, sizeof(VoxState), err))) return rc;
    if ((rc = vensure(ws, V_PART, (sizeof(float) * 6 + sizeof(int)) * 256, err))) return rc;
    const unsigned char *in = static_cast<const unsigned char *>(ws->buf[V_IN]);
    // V_KEYS / V_KEYS2 hold (voxel, point) as two 32-bit arrays each: unsorted and sorted
    unsigned *vox_in = static_cast<unsigned *>(ws->buf[V_KEYS]), *idx_in = vox_in + n;
    unsigned *vox_out = static_cast<unsigned *>(ws->buf[V_KEYS2]), *idx_out = vox_out + n;
    VoxState *st = static_cast<VoxState *>(ws->buf[V_STATE]);
    float *part = static_cast<float *>(ws->buf[V_PART]);
    int *pcnt = reinterpret_cast<int *>(part + 6 * 256);
    const float inv = 1.0f / leaf;
    int nb = (n + 255) / 256; nb = nb > 256 ? 256 : nb;
    const int pb = (n + 255) / 256;
    hipLaunchKernelGGL(vox_bbox_partial_kernel, dim3(nb), dim3(256), 0, stream, in, n, stride, part, pcnt);
    hipLaunchKernelGGL(vox_setup_kernel, dim3(1), dim3(64), 0, stream, part, pcnt, nb, inv, st);
    hipLaunchKernelGGL(vox_keys_kernel, dim3(pb), dim3(256), 0, stream, in, n, stride, inv, st, vox_in, idx_in);
    // (voxel, point) pairs by voxel, equal voxels in the cloud's order (csrc/device_sort.hip: three passes of eleven bits)
    const size_t tmp_sort = sort_scratch_bytes((size_t)n, 1), tmp_scan = scan_scratch_bytes((size_t)n);
    const size_t tmp = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
    if ((rc = vensure(ws, V_TMP, tmp + 256, err))) return rc;
    VOX_HIP(sort_pairs_u32(ws->buf[V_TMP], vox_in, vox_out, idx_in, idx_out, n, 32, stream));
    hipLaunchKernelGGL(vox_heads_kernel, dim3(pb), dim3(256), 0, stream, vox_out, n, (int *)ws->buf[V_HEAD]);
    VOX_HIP(prefix_sum_i32(ws->buf[V_TMP], (const int *)ws->buf[V_HEAD], (int *)ws->buf[V_POS], n, false, stream));
    hipLaunchKernelGGL(vox_centroid_kernel, dim3(pb), dim3(256), 0, stream, in, stride, vox_out, idx_out, (const int *)ws->buf[V_HEAD],
                       (const int *)ws->buf[V_POS], n, (unsigned char *)ws->buf[V_OUT], st);
    VOX_HIP(hipGetLastError());
    if (!ws->pinned || ws->pinned_cap < sizeof(VoxState)) {
        if (ws->pinned) { (void)hipHostFree(ws->pinned); ws->pinned = nullptr; ws->pinned_cap = 0; }
        VOX_HIP(hipHostMalloc(&ws->pinned, 4096, hipHostMallocDefault));
        ws->pinned_cap = 4096;
    }
    VOX_HIP(hipMemcpyAsync(ws->pinned, st, sizeof(VoxState), hipMemcpyDeviceToHost, stream));
    VOX_HIP(hipStreamSynchronize(stream));
    const VoxState *h = static_cast<const VoxState *>(ws->pinned);
    if (h->overflow) { *n_out = -1; return SCL_OK; }
    *n_out = h->nout;
    return SCL_OK;
}

}  // namespace

// host cloud in, filtered cloud left on the device (*d_result, valid until the workspace is used again);
// *n_out = -1 when PCL would return the input unchanged (index overflow): *d_result is then the unfiltered copy
int voxel_grid_to_device(IcpWorkspace *ws, hipStream_t stream, const void *in, int n, int stride, float leaf,
                         const void **d_result, int *n_out, std::string *err)
{
    if (n < 0 || stride < 12 || (stride & 3) || !(leaf > 0.f)) { if (err) *err = "voxel_grid: bad arguments"; return SCL_ERR_INVALID_ARG; }
    int rc;
    if ((rc = vensure(ws, V_IN, (size_t)n * stride + 16, err))) return rc;
    if (n) VOX_HIP(hipMemcpyAsync(ws->buf[V_IN], in, (size_t)n * stride, hipMemcpyHostToDevice, stream));
    int m = 0;
    if ((rc = voxel_device(ws, stream, n, stride, leaf, &m, err))) return rc;
    *d_result = m < 0 ? ws->buf[V_IN] : ws->buf[V_OUT];
    *n_out = m < 0 ? n : m;
    return SCL_OK;
}

int voxel_grid(IcpWorkspace *ws, hipStream_t stream, const void *in, int n, int stride, float leaf,
               void *out, int out_capacity, int *n_out, std::string *err)
{
    if (n < 0 || stride < 12 || (stride & 3) || !(leaf > 0.f)) { if (err) *err = "voxel_grid: bad arguments"; return SCL_ERR_INVALID_ARG; }
    int rc;
    if ((rc = vensure(ws, V_IN, (size_t)n * stride + 16, err))) return rc;
    if (n) VOX_HIP(hipMemcpyAsync(ws->buf[V_IN], in, (size_t)n * stride, hipMemcpyHostToDevice, stream));
    int m = 0;
    if ((rc = voxel_device(ws, stream, n, stride, leaf, &m, err))) return rc;
    if (m < 0) {                                                       // PCL: leaf too small -> input returned unchanged
        if (n > out_capacity) { if (err) *err = "voxel_grid: output capacity too small"; return SCL_ERR_INVALID_ARG; }
        std::memcpy(out, in, (size_t)n * stride);
        *n_out = n;
        return SCL_OK;
    }
    if (m > out_capacity) { if (err) *err = "voxel_grid: output capacity too small"; return SCL_ERR_INVALID_ARG; }
    if (m) VOX_HIP(hipMemcpyAsync(out, ws->buf[V_OUT], (size_t)m * stride, hipMemcpyDeviceToHost, stream));
    VOX_HIP(hipStreamSynchronize(stream));
    *n_out = m;
    return SCL_OK;
}

// clouds_on_device: clouds[c] are device pointers (the on-device keyframe store) and are read in place.
// d_result != nullptr: the filtered submap stays in the workspace (*d_result, valid until the workspace is
// used again) and nothing crosses PCIe but the 16-float poses and the point count.
int assemble_submap_ex(IcpWorkspace *ws, hipStream_t stream, const void *const *clouds, const int *counts,
                       const float *transforms, int n_clouds, int stride, float leaf, bool clouds_on_device,
                       void *out, int out_capacity, const void **d_result, int *n_out, std::string *err)
{
    if (n_clouds < 0 || stride < 12 || (stride & 3) || !(leaf > 0.f)) { if (err) *err = "assemble_submap: bad arguments"; return SCL_ERR_INVALID_ARG; }
    size_t total = 0;
    for (int c = 0; c < n_clouds; ++c) { if (counts[c] < 0) { if (err) *err = "negative count"; return SCL_ERR_INVALID_ARG; } total += (size_t)counts[c]; }
    if (total > 0x7fffffffull) { if (err) *err = "submap too large"; return SCL_ERR_INVALID_ARG; }
    int rc;
    if ((rc = vensure(ws, V_IN, total * stride + 16, err))) return rc;
    if ((rc = vensure(ws, V_T, sizeof(float) * 16 * (size_t)(n_clouds > 0 ? n_clouds : 1), err))) return rc;
    if (n_clouds) VOX_HIP(hipMemcpyAsync(ws->buf[V_T], transforms, sizeof(float) * 16 * (size_t)n_clouds, hipMemcpyHostToDevice, stream));
    // host clouds are staged behind the concatenated buffer's end in V_OUT (reused as scratch before the filter runs)
    size_t maxc = 0;
    for (int c = 0; c < n_clouds; ++c) maxc = maxc > (size_t)counts[c] ? maxc : (size_t)counts[c];
    if ((rc = vensure(ws, V_OUT, (total > maxc ? total : maxc) * stride + 16, err))) return rc;
    size_t off = 0;
    for (int c = 0; c < n_clouds; ++c) {                               // DM.h:1168-1176
        const int n = counts[c];
        if (!n) continue;
        const unsigned char *d_cloud = static_cast<const unsigned char *>(clouds[c]);
        if (!clouds_on_device) {
            VOX_HIP(hipMemcpyAsync(ws->buf[V_OUT], clouds[c], (size_t)n * stride, hipMemcpyHostToDevice, stream));
            d_cloud = static_cast<const unsigned char *>(ws->buf[V_OUT]);
        }
        hipLaunchKernelGGL(transform_append_kernel, dim3((n + 255) / 256), dim3(256), 0, stream,
                           d_cloud, n, stride, (const float *)ws->buf[V_T] + 16 * c,
                           (unsigned char *)ws->buf[V_IN] + off * stride);
        off += (size_t)n;
    }
    int m = 0;
    if ((rc = voxel_device(ws, stream, (int)total, stride, leaf, &m, err))) return rc;   // DM.h:1181-1185
    const void *res = ws->buf[V_OUT];
    if (m < 0) { m = (int)total; res = ws->buf[V_IN]; }
    if (d_result) *d_result = res;
    if (out) {
        if (m > out_capacity) { if (err) *err = "assemble_submap: output capacity too small"; return SCL_ERR_INVALID_ARG; }
        if (m) VOX_HIP(hipMemcpyAsync(out, res, (size_t)m * stride, hipMemcpyDeviceToHost, stream));
    }
    VOX_HIP(hipStreamSynchronize(stream));
    *n_out = m;
    return SCL_OK;
}

// n_jobs submaps at once (see the batched kernels above): job j = the clouds [first[j], first[j + 1]) of `clouds` (device pointers:
// the keyframe store), each moved by its transform.  d_results[j] / n_out[j]: the filtered submap of job j, left in the workspace
// (valid until the workspace is used again); a job whose voxel index would overflow keeps its unfiltered points (PCL's behaviour).
int assemble_submaps_batch(IcpWorkspace *ws, hipStream_t stream, const void *const *clouds, const int *counts, const float *transforms,
                           const int *first, int n_jobs, int stride, float leaf, const void **d_results, int *n_out, std::string *err)
{
    if (n_jobs < 0 || n_jobs > kVoxMaxJobs || stride < 12 || (stride & 3) || !(leaf > 0.f)) { if (err) *err = "assemble_submaps_batch: bad arguments"; return SCL_ERR_INVALID_ARG; }
    if (n_jobs == 0) return SCL_OK;
    const int n_pieces = first[n_jobs];
    size_t total = 0;
    int max_piece = 0, max_job = 0;
    std::vector<VoxJob> jobs((size_t)n_jobs);
    std::vector<VoxPiece> pieces((size_t)(n_pieces > 0 ? n_pieces : 1));
    for (int j = 0; j < n_jobs; ++j) {
        jobs[(size_t)j].off = (int)total;
        for (int c = first[j]; c < first[j + 1]; ++c) {
            if (counts[c] < 0) { if (err) *err = "negative count"; return SCL_ERR_INVALID_ARG; }
            pieces[(size_t)c] = VoxPiece{static_cast<const unsigned char *>(clouds[c]), counts[c], (int)total, c};
            total += (size_t)counts[c];
            if (total > 0x7fffffffull) { if (err) *err = "submaps too large"; return SCL_ERR_INVALID_ARG; }
            max_piece = counts[c] > max_piece ? counts[c] : max_piece;
        }
        jobs[(size_t)j].n = (int)total - jobs[(size_t)j].off;
        max_job = jobs[(size_t)j].n > max_job ? jobs[(size_t)j].n : max_job;
        d_results[j] = nullptr; n_out[j] = 0;
    }
    if (total == 0) return SCL_OK;
    const int n = (int)total;
    int rc;
    int nb = (max_job + 255) / 256; nb = nb < 1 ? 1 : (nb > 256 ? 256 : nb);
    const size_t tab_bytes = sizeof(float) * 16 * (size_t)(n_pieces > 0 ? n_pieces : 1) + sizeof(VoxPiece) * pieces.size() + sizeof(VoxJob) * jobs.size();
    if ((rc = vensure(ws, V_IN, total * stride + 16, err))) return rc;
    if ((rc = vensure(ws, V_OUT, total * stride + 16, err))) return rc;
    if ((rc = vensure(ws, V_KEYS, (sizeof(unsigned long long) + sizeof(unsigned)) * total + 16, err))) return rc;
    if ((rc = vensure(ws, V_KEYS2, (sizeof(unsigned long long) + sizeof(unsigned)) * total + 16, err))) return rc;
    if ((rc = vensure(ws, V_HEAD, sizeof(int) * total, err))) return rc;
    if ((rc = vensure(ws, V_POS, sizeof(int) * total, err))) return rc;
    if ((rc = vensure(ws, V_STATE, sizeof(VoxState) * (size_t)n_jobs, err))) return rc;
    if ((rc = vensure(ws, V_PART, (sizeof(float) * 6 + sizeof(int)) * (size_t)nb * (size_t)n_jobs, err))) return rc;
    if ((rc = vensure(ws, V_T, tab_bytes + 64, err))) return rc;
    const size_t pin_bytes = tab_bytes > sizeof(VoxState) * (size_t)n_jobs ? tab_bytes : sizeof(VoxState) * (size_t)n_jobs;
    if (!ws->pinned || ws->pinned_cap < pin_bytes) {
        if (ws->pinned) { (void)hipHostFree(ws->pinned); ws->pinned = nullptr; ws->pinned_cap = 0; }
        VOX_HIP(hipHostMalloc(&ws->pinned, pin_bytes + 4096, hipHostMallocDefault));
        ws->pinned_cap = pin_bytes + 4096;
    }
    // the tables: transforms, pieces, jobs -- one copy
    unsigned char *hp = static_cast<unsigned char *>(ws->pinned);
    const size_t o_pieces = sizeof(float) * 16 * (size_t)(n_pieces > 0 ? n_pieces : 1), o_jobs = o_pieces + sizeof(VoxPiece) * pieces.size();
    if (n_pieces) std::memcpy(hp, transforms, sizeof(float) * 16 * (size_t)n_pieces);
    std::memcpy(hp + o_pieces, pieces.data(), sizeof(VoxPiece) * pieces.size());
    std::memcpy(hp + o_jobs, jobs.data(), sizeof(VoxJob) * jobs.size());
    VOX_HIP(hipMemcpyAsync(ws->buf[V_T], hp, tab_bytes, hipMemcpyHostToDevice, stream));
    const float *d_T = static_cast<const float *>(ws->buf[V_T]);
    const VoxPiece *d_pieces = reinterpret_cast<const VoxPiece *>(static_cast<unsigned char *>(ws->buf[V_T]) + o_pieces);
    const VoxJob *d_jobs = reinterpret_cast<const VoxJob *>(static_cast<unsigned char *>(ws->buf[V_T]) + o_jobs);
    unsigned char *raw = static_cast<unsigned char *>(ws->buf[V_IN]);
    unsigned long long *keys_in = static_cast<unsigned long long *>(ws->buf[V_KEYS]), *keys_out = static_cast<unsigned long long *>(ws->buf[V_KEYS2]);
    unsigned *idx_in = reinterpret_cast<unsigned *>(keys_in + total), *idx_out = reinterpret_cast<unsigned *>(keys_out + total);
    VoxState *st = static_cast<VoxState *>(ws->buf[V_STATE]);
    float *part = static_cast<float *>(ws->buf[V_PART]);
    int *pcnt = reinterpret_cast<int *>(part + 6 * (size_t)nb * (size_t)n_jobs);
    const float inv = 1.0f / leaf;
    const int pb = (n + 255) / 256;
    int ebits = 32;
    while ((1 << (ebits - 32)) < n_jobs) ++ebits;                      // the sort looks at the voxel's 32 bits and the job's
    if (n_pieces) {
        int gb = (max_piece + 255) / 256; gb = gb < 1 ? 1 : (gb > 512 ? 512 : gb);
        hipLaunchKernelGGL(transform_pieces_kernel, dim3(gb, n_pieces), dim3(256), 0, stream, d_pieces, d_T, stride, raw);
    }
    hipLaunchKernelGGL(vox_bbox_partial_batch_kernel, dim3(nb, n_jobs), dim3(256), 0, stream, raw, d_jobs, stride, part, pcnt);
    hipLaunchKernelGGL(vox_setup_batch_kernel, dim3(n_jobs), dim3(64), 0, stream, part, pcnt, nb, inv, st);
    hipLaunchKernelGGL(vox_keys_batch_kernel, dim3(pb), dim3(256), 0, stream, raw, n, d_jobs, n_jobs, stride, inv, st, keys_in, idx_in);
    // ((job, voxel), point) pairs by job and voxel, equal keys in the submap's order.  The jobs lie behind each other in the buffer: up to
    // kSortMaxSegments of them are sorted each on its own 32 voxel bits (three passes), more than that all together on (job, voxel)
    const bool segmented = n_jobs <= kSortMaxSegments;
    SortSegments seg{};
    if (segmented) {
        seg.nseg = n_jobs;
        for (int j = 0; j < n_jobs; ++j) seg.off[j] = jobs[(size_t)j].off;
        seg.off[n_jobs] = n;
    }
    const size_t tmp_sort = sort_scratch_bytes((size_t)n, segmented ? n_jobs : 1), tmp_scan = scan_scratch_bytes((size_t)n);
    const size_t tmp = tmp_sort > tmp_scan ? tmp_sort : tmp_scan;
    if ((rc = vensure(ws, V_TMP, tmp + 256, err))) return rc;
    unsigned int seg_hi[kSortMaxSegments];                               // vox_keys_batch_kernel's keys: job << 32 | voxel
    for (int j = 0; j < n_jobs && j < kSortMaxSegments; ++j) seg_hi[j] = (unsigned int)j;
    if (segmented) VOX_HIP(sort_pairs_u64_segmented(ws->buf[V_TMP], keys_in, keys_out, idx_in, idx_out, seg, 32, stream, seg_hi));
    else VOX_HIP(sort_pairs_u64(ws->buf[V_TMP], keys_in, keys_out, idx_in, idx_out, n, ebits, stream));
    hipLaunchKernelGGL(vox_heads_batch_kernel, dim3(pb), dim3(256), 0, stream, keys_out, n, (int *)ws->buf[V_HEAD]);
    VOX_HIP(prefix_sum_i32(ws->buf[V_TMP], (const int *)ws->buf[V_HEAD], (int *)ws->buf[V_POS], n, false, stream));
    hipLaunchKernelGGL(vox_centroid_batch_kernel, dim3(pb), dim3(256), 0, stream, raw, stride, keys_out, idx_out, (const int *)ws->buf[V_HEAD],
                       (const int *)ws->buf[V_POS], n, d_jobs, (unsigned char *)ws->buf[V_OUT], st);
    VOX_HIP(hipGetLastError());
    VOX_HIP(hipMemcpyAsync(ws->pinned, st, sizeof(VoxState) * (size_t)n_jobs, hipMemcpyDeviceToHost, stream));
    VOX_HIP(hipStreamSynchronize(stream));
    const VoxState *h = static_cast<const VoxState *>(ws->pinned);
    for (int j = 0; j < n_jobs; ++j) {
        const bool unfiltered = h[j].overflow != 0;                     // DM.h:1183-1185 through PCL: leaf too small for the cloud's extent
        d_results[j] = (unfiltered ? static_cast<unsigned char *>(ws->buf[V_IN]) : static_cast<unsigned char *>(ws->buf[V_OUT])) + (size_t)jobs[(size_t)j].off * stride;
        n_out[j] = unfiltered ? jobs[(size_t)j].n : h[j].nout;
    }
    return SCL_OK;
}

int assemble_submap(IcpWorkspace *ws, hipStream_t stream, const void *const *clouds, const int *counts,
                    const float *transforms, int n_clouds, int stride, float leaf, void *out, int out_capacity,
                    int *n_out, std::string *err)
{
    return assemble_submap_ex(ws, stream, clouds, counts, transforms, n_clouds, stride, leaf, false,
                              out, out_capacity, nullptr, n_out, err);
}

}  // namespace scl
