// iris.hip -- LiDAR-Iris building blocks on the GPU (include/scl_iris.h; reference include/descriptor.h:462-1302).
//
//   iris_image_kernel     getIris (D.h:532-598): one thread per point, atomicOr of the elevation bit into the
//                         (distance, yaw) cell, atomicMax of the height (order-preserving int image of the float;
//                         cells start at 0 like Eigen::MatrixXf::Zero, so only positive heights register);
//   iris_rowkey_kernel    row means in the reference's left-to-right float order;
//   (host, once)          the circular-convolution kernels of the four log-Gabor scales: response = idft(dft(x) * G)
//                         with both transforms unscaled (cv::dft / cv::idft without DFT_SCALE, D.h:651-653) equals
//                         x (*) h, h[n] = sum_k G[k] e^{2 pi i k n / N} -- computed in fp64 at engine creation;
//   iris_encode_kernel    logFeatureEncode (D.h:661-680): one workgroup per image column n, every (scale, row) response
//                         sum_m x[r][m] h_s[(n - m) mod N] in fp64 in index order, narrowed to float like the reference's
//                         planes, thresholded into the bit-packed templates: T / M as [column][20 words] bit masks over
//                         the 640 template rows, so a column shift is index arithmetic;
//   iris_hamming_kernel   getHammingDistance (D.h:932-964): one wave per (candidate, shift), popcounts of
//                         (T1s ^ T2) & ~(M1s | M2) and of the mask, integers end to end.
#include "scl_iris.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "device_common.hpp"

using namespace scl;

namespace {

__device__ __forceinline__ float iris_atan2f(float y, float x)
{   // the fixed fp64 atan of device_common.hpp behind std::atan2(float, float): same bits as oracle/iris_oracle.c
    const double PI = 3.14159265358979323846;
    if (x != x || y != y) return __int_as_float(0x7fc00000);
    if (y == 0.0f) return (x < 0.0f || (x == 0.0f && (__float_as_int(x) < 0))) ? ((__float_as_int(y) < 0) ? -(float)PI : (float)PI) : y;
    if (x == 0.0f) return y > 0.0f ? (float)(PI / 2) : (float)(-PI / 2);
    const double ay = fabs((double)y), ax = fabs((double)x);
    const bool iy = ay > 1.7976931348623157e308, ix = ax > 1.7976931348623157e308;
    double a = iy ? (ix ? PI / 4 : PI / 2) : (ix ? 0.0 : atan_pos(ay / ax));
    if (x < 0.0f) a = PI - a;
    return (float)(y < 0.0f ? -a : a);
}

__device__ __forceinline__ int floor_to_int_x86(double v)
{
    const double f = floor(v);
    if (!(f >= -2147483648.0 && f <= 2147483647.0)) return (-2147483647 - 1);
    return (int)f;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ void iris_image_kernel(const unsigned char *pts, int n, int stride, int rows, int cols, double add,
                                  unsigned int *cells /* rows*cols words: the byte image widened */, int *zmax)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *f = reinterpret_cast<const float *>(pts + (size_t)i * (size_t)stride);
        const float x = f[0], y = f[1], z = f[2];
        const float dis = sqrtf(x * x + y * y);                                               // D.h:543
        const float arc = (float)((double)(iris_atan2f(z, dis) * 180.0f) / 3.14159265358979323846 + add);
        const float yaw = (float)((double)(iris_atan2f(y, x) * 180.0f) / 3.14159265358979323846 + 180);
        const int q_dis = clampi(floor_to_int_x86((double)dis), 0, rows - 1);
        const int q_arc = clampi(floor_to_int_x86((double)(arc / 4.0f)), 0, 7);
        const int q_yaw = clampi(floor_to_int_x86((double)yaw + 0.5), 0, cols - 1);
        const int cell = q_dis * cols + q_yaw;
        atomicOr(&cells[cell], 1u << q_arc);
        if (z > 0.0f) atomicMax(&zmax[cell], __float_as_int(z));      // positive floats order like their bit patterns; NaN never passes '<'
    }
}

__global__ void iris_rowkey_kernel(const int *zmax, const unsigned int *cells, int rows, int cols, float *rowkey, unsigned char *image)
{
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) image[(size_t)r * cols + c] = (unsigned char)cells[(size_t)r * cols + c];
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int c = 0; c < cols; ++c) s += __int_as_float(zmax[(size_t)r * cols + c]);
        rowkey[r] = s / (float)cols;
    }
}

// (kIrisWords = 20)                 // 640 template rows (2 * 4 scales * 80 rows) as 20 words per column

// one workgroup per image column n
__global__ __launch_bounds__(256) void iris_encode_kernel(const unsigned char *image, const double2 *h, int rows, int N, int nscale,
                                                          unsigned int *Tw, unsigned int *Mw /* [N][words] */, int words)
{
    extern __shared__ unsigned int lds_words[];             // [2 * words]
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * words; i += blockDim.x) lds_words[i] = 0u;
    __syncthreads();
    for (int job = threadIdx.x; job < nscale * rows; job += blockDim.x) {
        const int s = job / rows, r = job - s * rows;
        const unsigned char *xr = image + (size_t)r * N;
        const double2 *hs = h + (size_t)s * N;
        double re = 0.0, im = 0.0;
        for (int m = 0; m < N; ++m) {                       // index order, zeros skipped: the restatement's order
            const unsigned char xv = xr[m];
            if (xv == 0) continue;
            int d = n - m; d = d < 0 ? d + N : d;
            const double2 hv = hs[d];
            re += (double)xv * hv.x; im += (double)xv * hv.y;
        }
        const float fre = (float)re, fim = (float)im;
        const float mag = sqrtf(fre * fre + fim * fim);
        const int ta = s * rows + r, tb = (s + nscale) * rows + r;           // vconcat order, D.h:669-678
        if (fre > 0.0f) atomicOr(&lds_words[ta >> 5], 1u << (ta & 31));
        if (fim > 0.0f) atomicOr(&lds_words[tb >> 5], 1u << (tb & 31));
        if (mag < 0.0001f) { atomicOr(&lds_words[words + (ta >> 5)], 1u << (ta & 31)); atomicOr(&lds_words[words + (tb >> 5)], 1u << (tb & 31)); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < words; i += blockDim.x) { Tw[(size_t)n * words + i] = lds_words[i]; Mw[(size_t)n * words + i] = lds_words[words + i]; }
}

__global__ void iris_unpack_kernel(const unsigned int *W, int N, int words, int trows, unsigned char *out /* [trows][N] */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= trows * N) return;
    const int tr = i / N, n = i - tr * N;
    out[i] = (W[(size_t)n * words + (tr >> 5)] >> (tr & 31)) & 1u ? 255 : 0;
}

// one wave per (candidate, shift index); shift = shifts[job]
__global__ __launch_bounds__(64) void iris_hamming_kernel(const unsigned int *T, const unsigned int *M, size_t feat_words /* per keyframe */,
                                                          int key1, const int *cand, const int *shifts, int shifts_per_cand,
                                                          int N, int words, int trows, int *bits_diff, int *total_bits)
{
    const int job = blockIdx.x, c = job / shifts_per_cand;
    const int key2 = cand[c];
    int sh = shifts[job] % N; sh = sh < 0 ? sh + N : sh;
    const unsigned int *T1 = T + (size_t)key1 * feat_words, *M1 = M + (size_t)key1 * feat_words;
    const unsigned int *T2 = T + (size_t)key2 * feat_words, *M2 = M + (size_t)key2 * feat_words;
    int diff = 0, masked = 0;
    for (int i = threadIdx.x; i < N * words; i += 64) {
        const int k = i / words, w = i - k * words;
        int src = k - sh; src = src < 0 ? src + N : src;                       // circColShift: dst(:, k) = src(:, k - shift), D.h:581-592
        const unsigned int mask = M1[(size_t)src * words + w] | M2[(size_t)k * words + w];
        const unsigned int x = (T1[(size_t)src * words + w] ^ T2[(size_t)k * words + w]) & ~mask;
        diff += __popc(x); masked += __popc(mask);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { diff += __shfl_xor(diff, off, 64); masked += __shfl_xor(masked, off, 64); }
    if (threadIdx.x == 0) { bits_diff[job] = diff; total_bits[job] = trows * N - masked; }
}


// Row-key candidate search (libnabo's exact knn, D.h:1103-1109 / 1209-1215): squared L2 in fp32 between keyframe q's
// row key and those of list[0..n), accumulated four dimensions at a time and unfused -- the metric order the engine
// uses for every KD-tree search of the reference (ringkey_topk.hip; oracle nf_l2).  The k smallest are picked on the host.
__global__ void iris_rowkey_d2_kernel(const float *rowkeys, int rows, int q, const int *list, int n, float *d2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *a = rowkeys + (size_t)q * rows, *b = rowkeys + (size_t)list[i] * rows;
    float result = 0.0f;
    int r = 0;
    for (; r + 3 < rows; r += 4) {
        const float d0 = __fsub_rn(a[r], b[r]), d1 = __fsub_rn(a[r + 1], b[r + 1]), d2v = __fsub_rn(a[r + 2], b[r + 2]), d3 = __fsub_rn(a[r + 3], b[r + 3]);
        const float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1)), __fmul_rn(d2v, d2v)), __fmul_rn(d3, d3));
        result = __fadd_rn(result, t);
    }
    for (; r < rows; ++r) { const float d0 = __fsub_rn(a[r], b[r]); result = __fadd_rn(result, __fmul_rn(d0, d0)); }
    d2[i] = result;
}

}  // namespace

struct scl_iris {
    scl_iris_config cfg;
    int device = 0;
    hipStream_t stream = nullptr;
    mutable std::mutex mu;
    mutable std::string last_error;
    int n = 0, cap = 0;
    int words = 0, trows = 0;
    unsigned char *d_images = nullptr; float *d_rowkeys = nullptr; unsigned int *d_T = nullptr, *d_M = nullptr;
    double2 *d_h = nullptr;
    unsigned int *d_cells = nullptr; int *d_zmax = nullptr; unsigned char *d_points = nullptr; size_t points_cap = 0;
    unsigned char *d_img1 = nullptr; float *d_key1 = nullptr; unsigned char *d_unpack = nullptr;
    int *d_cand = nullptr, *d_shifts = nullptr, *d_diff = nullptr, *d_total = nullptr; size_t job_cap = 0;
    std::vector<int8_t> robots; std::vector<int> indexs;
    // the plugin layer (D.h:1289-1292): per robot the global keys of its keyframes in arrival order
    std::vector<std::vector<int>> local2global;
    int *d_list = nullptr; float *d_d2 = nullptr; size_t list_cap = 0;
};

namespace {

#define IRIS_HIP(h_, call)                                                             \
    do {                                                                               \
        hipError_t err__ = (call);                                                     \
        if (err__ != hipSuccess) {                                                     \
            (h_)->last_error = std::string(#call) + ": " + hipGetErrorString(err__);   \
            return err__ == hipErrorOutOfMemory ? SCL_ERR_NOMEM : SCL_ERR_HIP;         \
        }                                                                              \
    } while (0)

int ifail(const scl_iris *h, int code, const char *msg) { if (h) h->last_error = msg; return code; }

template <class T> int ialloc(scl_iris *h, T **p, size_t count)
{
    void *q = nullptr;
    IRIS_HIP(h, hipMalloc(&q, sizeof(T) * (count ? count : 1)));
    *p = static_cast<T *>(q);
    return SCL_OK;
}

int grow(scl_iris *h, int need)
{
    if (need <= h->cap) return SCL_OK;
    int ncap = h->cap > 0 ? h->cap : 256;
    while (ncap < need) ncap *= 2;
    const size_t cells = (size_t)h->cfg.rows * h->cfg.cols, fw = (size_t)h->cfg.cols * h->words;
    unsigned char *ni = nullptr; float *nk = nullptr; unsigned int *nt = nullptr, *nm = nullptr;
    int rc;
    if ((rc = ialloc(h, &ni, cells * ncap)) || (rc = ialloc(h, &nk, (size_t)h->cfg.rows * ncap)) || (rc = ialloc(h, &nt, fw * ncap)) || (rc = ialloc(h, &nm, fw * ncap))) return rc;
    if (h->n > 0) {
        IRIS_HIP(h, hipMemcpyAsync(ni, h->d_images, cells * h->n, hipMemcpyDeviceToDevice, h->stream));
        IRIS_HIP(h, hipMemcpyAsync(nk, h->d_rowkeys, sizeof(float) * h->cfg.rows * h->n, hipMemcpyDeviceToDevice, h->stream));
        IRIS_HIP(h, hipMemcpyAsync(nt, h->d_T, sizeof(unsigned int) * fw * h->n, hipMemcpyDeviceToDevice, h->stream));
        IRIS_HIP(h, hipMemcpyAsync(nm, h->d_M, sizeof(unsigned int) * fw * h->n, hipMemcpyDeviceToDevice, h->stream));
    }
    IRIS_HIP(h, hipStreamSynchronize(h->stream));
    if (h->d_images) (void)hipFree(h->d_images);
    if (h->d_rowkeys) (void)hipFree(h->d_rowkeys);
    if (h->d_T) (void)hipFree(h->d_T);
    if (h->d_M) (void)hipFree(h->d_M);
    h->d_images = ni; h->d_rowkeys = nk; h->d_T = nt; h->d_M = nm; h->cap = ncap;
    return SCL_OK;
}

// points (host) -> d_img1 / d_key1
int make_image_locked(scl_iris *h, const void *points, int n_points, int stride)
{
    if (n_points < 0 || stride < 12 || (stride & 3) || (n_points > 0 && !points)) return ifail(h, SCL_ERR_INVALID_ARG, "bad point layout");
    const int rows = h->cfg.rows, cols = h->cfg.cols;
    const size_t bytes = (size_t)n_points * stride;
    if (bytes > h->points_cap) {
        if (h->d_points) (void)hipFree(h->d_points);
        h->points_cap = 0;
        int rc = ialloc(h, &h->d_points, bytes + bytes / 4 + 4096);
        if (rc) return rc;
        h->points_cap = bytes + bytes / 4 + 4096;
    }
    if (bytes) IRIS_HIP(h, hipMemcpyAsync(h->d_points, points, bytes, hipMemcpyHostToDevice, h->stream));
    IRIS_HIP(h, hipMemsetAsync(h->d_cells, 0, sizeof(unsigned int) * (size_t)rows * cols, h->stream));
    IRIS_HIP(h, hipMemsetAsync(h->d_zmax, 0, sizeof(int) * (size_t)rows * cols, h->stream));
    if (n_points > 0 && (h->cfg.nscan == 16 || h->cfg.nscan == 64)) {             // D.h:538 / 560: other beam counts leave the image empty
        int blocks = (n_points + 255) / 256; blocks = blocks > 2048 ? 2048 : blocks;
        hipLaunchKernelGGL(iris_image_kernel, dim3(blocks), dim3(256), 0, h->stream, h->d_points, n_points, stride, rows, cols,
                           h->cfg.nscan == 16 ? 15.0 : 24.9, h->d_cells, h->d_zmax);
    }
    hipLaunchKernelGGL(iris_rowkey_kernel, dim3(rows), dim3(128), 0, h->stream, h->d_zmax, h->d_cells, rows, cols, h->d_key1, h->d_img1);
    IRIS_HIP(h, hipGetLastError());
    return SCL_OK;
}

// d_img1 / d_key1 -> database slot n (image, row key, templates)
int append_locked(scl_iris *h, int8_t robot, int index)
{
    if (robot < 0 || robot >= h->cfg.robot_num) return ifail(h, SCL_ERR_INVALID_ARG, "robot id outside [0, robot_num)");
    int rc = grow(h, h->n + 1);
    if (rc) return rc;
    const int rows = h->cfg.rows, cols = h->cfg.cols;
    const size_t cells = (size_t)rows * cols, fw = (size_t)cols * h->words;
    IRIS_HIP(h, hipMemcpyAsync(h->d_images + cells * h->n, h->d_img1, cells, hipMemcpyDeviceToDevice, h->stream));
    IRIS_HIP(h, hipMemcpyAsync(h->d_rowkeys + (size_t)rows * h->n, h->d_key1, sizeof(float) * rows, hipMemcpyDeviceToDevice, h->stream));
    hipLaunchKernelGGL(iris_encode_kernel, dim3(cols), dim3(256), sizeof(unsigned int) * 2 * h->words, h->stream,
                       h->d_img1, h->d_h, rows, cols, h->cfg.nscale, h->d_T + fw * h->n, h->d_M + fw * h->n, h->words);
    IRIS_HIP(h, hipGetLastError());
    IRIS_HIP(h, hipStreamSynchronize(h->stream));
    h->local2global[(size_t)robot].push_back(h->n);                                // D.h:1055
    h->robots.push_back(robot); h->indexs.push_back(index); h->n++;            // D.h:1057
    return SCL_OK;
}

int hamming_jobs_locked(scl_iris *h, int key1, const int *cand, const int *shifts, int n, int per, float *dis, int *bias, bool window)
{
    if (key1 < 0 || key1 >= h->n) return ifail(h, SCL_ERR_OUT_OF_RANGE, "key1 out of range");
    for (int i = 0; i < n; ++i) if (cand[i] < 0 || cand[i] >= h->n) return ifail(h, SCL_ERR_OUT_OF_RANGE, "candidate out of range");
    const size_t jobs = (size_t)n * per;
    if (jobs > h->job_cap) {
        for (int **p : {&h->d_cand, &h->d_shifts, &h->d_diff, &h->d_total}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        h->job_cap = 0;
        int rc;
        if ((rc = ialloc(h, &h->d_cand, jobs + 64)) || (rc = ialloc(h, &h->d_shifts, jobs + 64)) || (rc = ialloc(h, &h->d_diff, jobs + 64)) || (rc = ialloc(h, &h->d_total, jobs + 64))) return rc;
        h->job_cap = jobs + 64;
    }
    IRIS_HIP(h, hipMemcpyAsync(h->d_cand, cand, sizeof(int) * n, hipMemcpyHostToDevice, h->stream));
    IRIS_HIP(h, hipMemcpyAsync(h->d_shifts, shifts, sizeof(int) * jobs, hipMemcpyHostToDevice, h->stream));
    const size_t fw = (size_t)h->cfg.cols * h->words;
    hipLaunchKernelGGL(iris_hamming_kernel, dim3((unsigned)jobs), dim3(64), 0, h->stream, h->d_T, h->d_M, fw, key1, h->d_cand, h->d_shifts, per,
                       h->cfg.cols, h->words, h->trows, h->d_diff, h->d_total);
    IRIS_HIP(h, hipGetLastError());
    std::vector<int> diff(jobs), total(jobs);
    IRIS_HIP(h, hipMemcpyAsync(diff.data(), h->d_diff, sizeof(int) * jobs, hipMemcpyDeviceToHost, h->stream));
    IRIS_HIP(h, hipMemcpyAsync(total.data(), h->d_total, sizeof(int) * jobs, hipMemcpyDeviceToHost, h->stream));
    IRIS_HIP(h, hipStreamSynchronize(h->stream));
    for (int c = 0; c < n; ++c) {                                             // the O(shifts) selection of D.h:937-962, float like the reference
        float best = NAN; int b = -1;
        for (int j = 0; j < per; ++j) {
            const size_t job = (size_t)c * per + j;
            if (total[job] == 0) { if (window) best = NAN; continue; }        // D.h:948-951 resets dis; the exhaustive form just skips
            const float cur = (float)diff[job] / (float)total[job];
            if (cur < best || std::isnan(best)) { best = cur; b = shifts[job]; }
        }
        dis[c] = best; bias[c] = b;
    }
    return SCL_OK;
}

// Candidate search + pairwise comparison shared by the two detections (D.h:1100-1137 / 1205-1242).  `list` = global keys of
// the search set in the order the reference concatenates them (the KD-tree's point order); `cur` = global key of the query.
// Returns the position in `list` of the best candidate (-1: none), its distance and shift.
int detect_core_locked(scl_iris *h, int cur, const std::vector<int> &list, int *best_pos, float *best_dis, int *best_bias)
{
    *best_pos = -1; *best_dis = 10000000.0f; *best_bias = 0;                  // D.h:1104-1106
    const int n = (int)list.size(), k = h->cfg.num_candidates;
    if (n <= 0 || k <= 0) return SCL_OK;
    if ((size_t)n > h->list_cap) {
        if (h->d_list) (void)hipFree(h->d_list);
        if (h->d_d2) (void)hipFree(h->d_d2);
        h->d_list = nullptr; h->d_d2 = nullptr; h->list_cap = 0;
        const size_t cap = (size_t)n + (size_t)n / 2 + 256;
        int rc;
        if ((rc = ialloc(h, &h->d_list, cap)) || (rc = ialloc(h, &h->d_d2, cap))) return rc;
        h->list_cap = cap;
    }
    IRIS_HIP(h, hipMemcpyAsync(h->d_list, list.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(iris_rowkey_d2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_rowkeys, h->cfg.rows, cur, h->d_list, n, h->d_d2);
    IRIS_HIP(h, hipGetLastError());
    std::vector<float> d2((size_t)n);
    IRIS_HIP(h, hipMemcpyAsync(d2.data(), h->d_d2, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    IRIS_HIP(h, hipStreamSynchronize(h->stream));
    // the k nearest: ascending distance, equal distances by ascending position; libnabo without ALLOW_SELF_MATCH skips
    // d2 <= eps; NaN / inf never enter (insertion like the engine's ring-key search)
    std::vector<int> pos; std::vector<float> pd;
    pos.reserve((size_t)k + 1); pd.reserve((size_t)k + 1);
    const float eps = h->cfg.knn_exclude_eps;
    for (int i = 0; i < n; ++i) {
        const float d = d2[(size_t)i];
        if (eps > 0.0f && d <= eps) continue;
        if (!(d < FLT_MAX)) continue;
        if ((int)pos.size() == k && !(d < pd.back())) continue;
        size_t j = pos.size();
        while (j > 0 && pd[j - 1] > d) --j;
        pos.insert(pos.begin() + (long)j, i); pd.insert(pd.begin() + (long)j, d);
        if ((int)pos.size() > k) { pos.pop_back(); pd.pop_back(); }
    }
    if (pos.empty()) return SCL_OK;
    const int m = (int)pos.size(), N = h->cfg.cols;
    std::vector<int> cand((size_t)m), shifts((size_t)m * N), bias((size_t)m);
    std::vector<float> dis((size_t)m);
    for (int c = 0; c < m; ++c) { cand[(size_t)c] = list[(size_t)pos[(size_t)c]]; for (int j = 0; j < N; ++j) shifts[(size_t)c * N + j] = j; }
    int rc = hamming_jobs_locked(h, cur, cand.data(), shifts.data(), m, N, dis.data(), bias.data(), false);
    if (rc) return rc;
    for (int c = 0; c < m; ++c)                                              // D.h:1112-1131: strict <, NaN never wins
        if (dis[(size_t)c] < *best_dis) { *best_dis = dis[(size_t)c]; *best_pos = pos[(size_t)c]; *best_bias = bias[(size_t)c]; }
    return SCL_OK;
}

}  // namespace

extern "C" {

int scl_iris_default_config(scl_iris_config *c)
{
    if (!c) return SCL_ERR_INVALID_ARG;
    c->rows = 80; c->cols = 360; c->nscan = 64; c->nscale = 4; c->min_wavelength = 18; c->mult = 1.6f; c->sigma_onf = 0.75f; c->device = 0;
    c->dist_thres = 0.32; c->num_exclude_recent = 30; c->match_num = 2; c->num_candidates = 10; c->robot_num = 1; c->this_id = 0;
    c->knn_exclude_eps = FLT_EPSILON; c->wire_decode = 0;
    return SCL_OK;
}

const char *scl_iris_last_error(const scl_iris *h) { return h ? h->last_error.c_str() : "null handle"; }

int scl_iris_create(const scl_iris_config *cfg, scl_iris **out)
{
    if (!cfg || !out) return SCL_ERR_INVALID_ARG;
    *out = nullptr;
    if (cfg->rows < 1 || cfg->rows > 512 || cfg->cols < 2 || cfg->cols > 2048 || cfg->nscale < 1 || cfg->nscale > 8 ||
        cfg->min_wavelength < 1 || !(cfg->mult > 0.f) || !(cfg->sigma_onf > 0.f) || cfg->sigma_onf == 1.0f ||
        cfg->robot_num < 1 || cfg->robot_num > 127 || cfg->this_id < 0 || cfg->this_id >= cfg->robot_num || cfg->num_candidates < 1 ||
        cfg->num_candidates > 4096 || cfg->num_exclude_recent < 0 || cfg->match_num < 0 || cfg->match_num > 2 || !(cfg->knn_exclude_eps >= 0.0f))
        return SCL_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SCL_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return SCL_ERR_INVALID_ARG;
    scl_iris *h = new (std::nothrow) scl_iris();
    if (!h) return SCL_ERR_NOMEM;
    h->cfg = *cfg; h->device = cfg->device;
    h->local2global.resize((size_t)cfg->robot_num);                             // D.h:501-509
    h->trows = 2 * cfg->nscale * cfg->rows; h->words = (h->trows + 31) / 32;
    auto bail = [&](int code) { scl_iris_destroy(h); return code; };
    if (hipSetDevice(h->device) != hipSuccess) return bail(SCL_ERR_HIP);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(SCL_ERR_HIP);
    const size_t cells = (size_t)cfg->rows * cfg->cols;
    int rc;
    if ((rc = ialloc(h, &h->d_cells, cells)) || (rc = ialloc(h, &h->d_zmax, cells)) || (rc = ialloc(h, &h->d_img1, cells)) ||
        (rc = ialloc(h, &h->d_key1, (size_t)cfg->rows)) || (rc = ialloc(h, &h->d_h, (size_t)cfg->nscale * cfg->cols)) ||
        (rc = ialloc(h, &h->d_unpack, (size_t)h->trows * cfg->cols))) return bail(rc);
    // the one-sided log-Gabor transfer functions, D.h:622-640 (float arithmetic like cv::log / pow / exp on Mat1f)
    const int N = cfg->cols, ndata = N - (N & 1);
    std::vector<float> g((size_t)cfg->nscale * N, 0.0f);
    double wavelength = cfg->min_wavelength;
    for (int s = 0; s < cfg->nscale; ++s) {
        const double fo = 1.0 / wavelength;
        for (int i = 0; i < ndata / 2 + 1; ++i) {
            const float radius = i == 0 ? 1.0f : (float)i / (float)ndata;
            float t = std::log((float)((double)radius / fo));
            t = t * t;
            const double denom = 2 * std::log((double)cfg->sigma_onf) * std::log((double)cfg->sigma_onf);
            g[(size_t)s * N + i] = std::exp((float)((double)(-t) / denom));
        }
        g[(size_t)s * N] = 0.0f;
        wavelength *= (double)cfg->mult;
    }
    // h[s][n] = sum_k G[k] e^{2 pi i k n / N} in fp64 on the host (once per engine; the CPU restatement forms the same sums
    // with the same libm, so the templates agree bit for bit)
    {
        const double TWO_PI = 6.283185307179586476925286766559;
        std::vector<double2> hh((size_t)cfg->nscale * N);
        for (int s = 0; s < cfg->nscale; ++s)
            for (int n = 0; n < N; ++n) {
                double re = 0.0, im = 0.0;
                for (int k = 0; k < N; ++k) {
                    const float gk = g[(size_t)s * N + k];
                    if (gk == 0.0f) continue;
                    const double a = TWO_PI * (double)(((long long)k * n) % N) / (double)N;
                    re += (double)gk * std::cos(a); im += (double)gk * std::sin(a);
                }
                hh[(size_t)s * N + n] = make_double2(re, im);
            }
        if (hipMemcpy(h->d_h, hh.data(), sizeof(double2) * hh.size(), hipMemcpyHostToDevice) != hipSuccess) return bail(SCL_ERR_HIP);
    }
    *out = h;
    return SCL_OK;
}

int scl_iris_destroy(scl_iris *h)
{
    if (!h) return SCL_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *p : {(void *)h->d_images, (void *)h->d_rowkeys, (void *)h->d_T, (void *)h->d_M, (void *)h->d_h, (void *)h->d_cells, (void *)h->d_zmax,
                    (void *)h->d_points, (void *)h->d_img1, (void *)h->d_key1, (void *)h->d_unpack, (void *)h->d_cand, (void *)h->d_shifts,
                    (void *)h->d_diff, (void *)h->d_total, (void *)h->d_list, (void *)h->d_d2})
        if (p) (void)hipFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SCL_OK;
}

int scl_iris_make_image(scl_iris *h, const void *points, int n_points, int stride_bytes, uint8_t *image, float *rowkey)
{
    if (!h || !image || !rowkey) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    int rc = make_image_locked(h, points, n_points, stride_bytes);
    if (rc) return rc;
    IRIS_HIP(h, hipMemcpyAsync(image, h->d_img1, (size_t)h->cfg.rows * h->cfg.cols, hipMemcpyDeviceToHost, h->stream));
    IRIS_HIP(h, hipMemcpyAsync(rowkey, h->d_key1, sizeof(float) * h->cfg.rows, hipMemcpyDeviceToHost, h->stream));
    IRIS_HIP(h, hipStreamSynchronize(h->stream));
    return SCL_OK;
}

int scl_iris_make_and_save(scl_iris *h, const void *points, int n_points, int stride_bytes, int8_t robot, int index, float *out_values)
{
    if (!h) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    int rc = make_image_locked(h, points, n_points, stride_bytes);
    if (rc) return rc;
    if ((rc = append_locked(h, robot, index))) return rc;
    if (out_values) {                                                         // D.h:1067-1081: image values row-major, then the row key
        const size_t cells = (size_t)h->cfg.rows * h->cfg.cols;
        std::vector<unsigned char> img(cells);
        IRIS_HIP(h, hipMemcpyAsync(img.data(), h->d_img1, cells, hipMemcpyDeviceToHost, h->stream));
        IRIS_HIP(h, hipMemcpyAsync(out_values + cells, h->d_key1, sizeof(float) * h->cfg.rows, hipMemcpyDeviceToHost, h->stream));
        IRIS_HIP(h, hipStreamSynchronize(h->stream));
        for (size_t i = 0; i < cells; ++i) out_values[i] = (float)img[i];
    }
    return SCL_OK;
}

int scl_iris_save_image(scl_iris *h, const uint8_t *image, const float *rowkey, int8_t robot, int index)
{
    if (!h || !image || !rowkey) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    IRIS_HIP(h, hipMemcpyAsync(h->d_img1, image, (size_t)h->cfg.rows * h->cfg.cols, hipMemcpyHostToDevice, h->stream));
    IRIS_HIP(h, hipMemcpyAsync(h->d_key1, rowkey, sizeof(float) * h->cfg.rows, hipMemcpyHostToDevice, h->stream));
    return append_locked(h, robot, index);
}

int scl_iris_save_from_wire(scl_iris *h, const float *values, int8_t robot, int index)
{
    if (!h || !values) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    const int rows = h->cfg.rows, cols = h->cfg.cols;
    std::vector<uint8_t> img((size_t)rows * cols);
    auto to_u8 = [](float f) -> uint8_t {                                     // float -> uchar as x86 does it: cvttss2si, low byte
        if (!(f > -2147483904.0f && f < 2147483648.0f)) return 0;              // (out of int range / NaN -> 0x80000000 -> 0)
        return (uint8_t)(int32_t)f;
    };
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c)
            img[(size_t)r * cols + c] = to_u8(h->cfg.wire_decode ? values[(size_t)r * cols + c]             // D.h:1067-1074's layout
                                                               : values[(size_t)r * (cols + 1) + c + 1]);   // D.h:1035
    IRIS_HIP(h, hipMemcpyAsync(h->d_img1, img.data(), img.size(), hipMemcpyHostToDevice, h->stream));
    IRIS_HIP(h, hipMemcpyAsync(h->d_key1, values + (size_t)rows * cols, sizeof(float) * rows, hipMemcpyHostToDevice, h->stream));   // D.h:1039-1042
    return append_locked(h, robot, index);                                    // synchronises before `img` goes away
}

int scl_iris_get_size_of(const scl_iris *h, int id)
{
    if (!h) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (id == -1) return h->n;                                                // D.h:1262-1265
    if (id < 0 || id >= h->cfg.robot_num) return ifail(h, SCL_ERR_OUT_OF_RANGE, "robot id outside [0, robot_num)");
    return (int)h->local2global[(size_t)id].size();                           // D.h:1268
}

int scl_iris_local_to_global(const scl_iris *h, int robot, int local, int *key)
{
    if (!h || !key) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (robot < 0 || robot >= h->cfg.robot_num) return ifail(h, SCL_ERR_OUT_OF_RANGE, "robot id outside [0, robot_num)");
    const std::vector<int> &l2g = h->local2global[(size_t)robot];
    if (local < 0 || local >= (int)l2g.size()) return ifail(h, SCL_ERR_OUT_OF_RANGE, "local index out of range");
    *key = l2g[(size_t)local];
    return SCL_OK;
}

int scl_iris_detect_intra(scl_iris *h, int cur, int *loop_id, float *bias, float *dist)
{
    if (!h || !loop_id || !bias) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    *loop_id = -1; *bias = 0.0f;
    if (dist) *dist = 10000000.0f;
    const std::vector<int> &mine = h->local2global[(size_t)h->cfg.this_id];
    if (cur < 0 || cur >= (int)mine.size()) return ifail(h, SCL_ERR_OUT_OF_RANGE, "detect_intra: no such keyframe of this robot");
    if (cur < h->cfg.num_exclude_recent + h->cfg.num_candidates + 1) return SCL_OK;     // D.h:1092-1095
    const int history = cur - h->cfg.num_exclude_recent;                                 // D.h:1097-1101
    std::vector<int> list(mine.begin(), mine.begin() + history);
    int pos, b; float d;
    int rc = detect_core_locked(h, mine[(size_t)cur], list, &pos, &d, &b);
    if (rc) return rc;
    if (dist) *dist = d;
    if ((double)d < h->cfg.dist_thres) { *loop_id = pos; *bias = (float)b; }             // D.h:1140-1144: the LOCAL index
    return SCL_OK;
}

int scl_iris_detect_inter(scl_iris *h, int cur, int *loop_id, float *bias, float *dist)
{
    if (!h || !loop_id || !bias) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    *loop_id = -1; *bias = 0.0f;
    if (dist) *dist = 10000000.0f;
    if (cur < 0 || cur >= h->n) return ifail(h, SCL_ERR_OUT_OF_RANGE, "detect_inter: key out of range");
    const int cur_robot = h->robots[(size_t)cur];                                        // D.h:1156
    std::vector<int> list;                                                               // newLocal2Global, D.h:1167-1195
    if (cur_robot == h->cfg.this_id) {
        for (int i = 0; i < h->cfg.robot_num; ++i)
            if (i != h->cfg.this_id) list.insert(list.end(), h->local2global[(size_t)i].begin(), h->local2global[(size_t)i].end());
    } else {
        const std::vector<int> &mine = h->local2global[(size_t)h->cfg.this_id];
        list.assign(mine.begin(), mine.end());
    }
    if ((int)list.size() < h->cfg.num_candidates + 1) return SCL_OK;                     // D.h:1198-1201
    int pos, b; float d;
    int rc = detect_core_locked(h, cur, list, &pos, &d, &b);
    if (rc) return rc;
    if (dist) *dist = d;
    if ((double)d < h->cfg.dist_thres && pos >= 0) { *loop_id = list[(size_t)pos]; *bias = (float)b; }   // D.h:1236, 1245-1248: the GLOBAL key
    return SCL_OK;
}

int scl_iris_get_size(const scl_iris *h)
{
    if (!h) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    return h->n;
}

int scl_iris_get_index(const scl_iris *h, int key, int8_t *robot, int *index)
{
    if (!h || !robot || !index) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (key < 0 || key >= h->n) return ifail(h, SCL_ERR_OUT_OF_RANGE, "key out of range");
    *robot = h->robots[(size_t)key]; *index = h->indexs[(size_t)key];
    return SCL_OK;
}

int scl_iris_get_image(scl_iris *h, int key, uint8_t *image, float *rowkey)
{
    if (!h || !image || !rowkey) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    if (key < 0 || key >= h->n) return ifail(h, SCL_ERR_OUT_OF_RANGE, "key out of range");
    const size_t cells = (size_t)h->cfg.rows * h->cfg.cols;
    IRIS_HIP(h, hipMemcpyAsync(image, h->d_images + cells * key, cells, hipMemcpyDeviceToHost, h->stream));
    IRIS_HIP(h, hipMemcpyAsync(rowkey, h->d_rowkeys + (size_t)h->cfg.rows * key, sizeof(float) * h->cfg.rows, hipMemcpyDeviceToHost, h->stream));
    IRIS_HIP(h, hipStreamSynchronize(h->stream));
    return SCL_OK;
}

int scl_iris_get_feature(scl_iris *h, int key, uint8_t *T, uint8_t *M)
{
    if (!h || !T || !M) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    if (key < 0 || key >= h->n) return ifail(h, SCL_ERR_OUT_OF_RANGE, "key out of range");
    const size_t fw = (size_t)h->cfg.cols * h->words, tot = (size_t)h->trows * h->cfg.cols;
    for (int which = 0; which < 2; ++which) {
        hipLaunchKernelGGL(iris_unpack_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream,
                           (which ? h->d_M : h->d_T) + fw * key, h->cfg.cols, h->words, h->trows, h->d_unpack);
        IRIS_HIP(h, hipGetLastError());
        IRIS_HIP(h, hipMemcpyAsync(which ? M : T, h->d_unpack, tot, hipMemcpyDeviceToHost, h->stream));
        IRIS_HIP(h, hipStreamSynchronize(h->stream));
    }
    return SCL_OK;
}

int scl_iris_hamming_batch(scl_iris *h, int key1, const int *cand, const int *scales, int n, float *dis, int *bias)
{
    if (!h || n < 0 || (n > 0 && (!cand || !scales || !dis || !bias))) return SCL_ERR_INVALID_ARG;
    if (n == 0) return SCL_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    std::vector<int> shifts((size_t)n * 5);
    for (int c = 0; c < n; ++c) for (int j = 0; j < 5; ++j) shifts[(size_t)c * 5 + j] = scales[c] - 2 + j;     // D.h:936
    return hamming_jobs_locked(h, key1, cand, shifts.data(), n, 5, dis, bias, true);
}

int scl_iris_hamming(scl_iris *h, int key1, int key2, int scale, float *dis, int *bias)
{
    return scl_iris_hamming_batch(h, key1, &key2, &scale, 1, dis, bias);
}

int scl_iris_hamming_all_shifts(scl_iris *h, int key1, const int *cand, int n, float *dis, int *bias)
{
    if (!h || n < 0 || (n > 0 && (!cand || !dis || !bias))) return SCL_ERR_INVALID_ARG;
    if (n == 0) return SCL_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    const int N = h->cfg.cols;
    std::vector<int> shifts((size_t)n * N);
    for (int c = 0; c < n; ++c) for (int j = 0; j < N; ++j) shifts[(size_t)c * N + j] = j;
    return hamming_jobs_locked(h, key1, cand, shifts.data(), n, N, dis, bias, false);
}

}  // extern "C"
