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
{   // std::atan2(float, float) of D.h:547-549 = glibc's atan2f (fdlibm's case analysis around atanf(|y / x|), fp32 throughout): the same
    // restatement as oracle/iris_oracle.c's, which equals libm on 4e9 pairs; atanf_glibc: device_common.hpp (all 2^32 inputs)
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int hx = __float_as_int(x), ix = hx & 0x7fffffff, hy = __float_as_int(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf_glibc(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf_glibc(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return __int_as_float(__float_as_int(z) ^ (int)0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
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
                                                          int N, int words, int trows, int *bits_diff, int *total_bits, const int *rolls2)
{
    const int job = blockIdx.x, c = job / shifts_per_cand;
    const int key2 = cand[c];
    int sh = shifts[job] % N; sh = sh < 0 ? sh + N : sh;
    int r2 = rolls2 ? rolls2[c] % N : 0; r2 = r2 < 0 ? r2 + N : r2;           // the second keyframe turned: circShift(T2, 0, roll), D.h:976-977
    const unsigned int *T1 = T + (size_t)key1 * feat_words, *M1 = M + (size_t)key1 * feat_words;
    const unsigned int *T2 = T + (size_t)key2 * feat_words, *M2 = M + (size_t)key2 * feat_words;
    int diff = 0, masked = 0;
    for (int i = threadIdx.x; i < N * words; i += 64) {
        const int k = i / words, w = i - k * words;
        int src = k - sh; src = src < 0 ? src + N : src;                       // circColShift: dst(:, k) = src(:, k - shift), D.h:581-592
        int k2 = k - r2; k2 = k2 < 0 ? k2 + N : k2;
        const unsigned int mask = M1[(size_t)src * words + w] | M2[(size_t)k2 * words + w];
        const unsigned int x = (T1[(size_t)src * words + w] ^ T2[(size_t)k2 * words + w]) & ~mask;
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


// ---- logPolarFFTTemplateMatch (D.h:793-925), the shift estimate in front of compare()'s Hamming windows -------------------------
// Restated from the algorithms OpenCV publishes (oracle/iris_oracle.c: iriso_fft_match states every step; PARITY UNPINNED against
// the reference's binaries).  Everything transcendental -- twiddles, the highpass, the log-polar map, the rotation matrix -- is
// evaluated on the host with the C library the CPU restatement uses and shipped as tables; the device adds, multiplies, divides and
// takes square roots, unfused, in the restatement's order: the two agree bit for bit.  J jobs (candidate, orientation) per launch.
struct FftJob { int key0, roll0, key1; };                  // im0 = image of key0 turned by roll0 columns, im1 = image of key1

// a[j][i] = (float)u8 * (float)(1 / 255)  (convertTo(CV_32FC1, 1.0 / 255.0), D.h:848-849)
__global__ void fm_stage_kernel(const unsigned char *images, const FftJob *jobs, int R, int C, float *a0, float *a1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i >= R * C) return;
    const FftJob jb = jobs[j];
    const int r = i / C, c = i - r * C;
    int s0 = (c - jb.roll0) % C; s0 = s0 < 0 ? s0 + C : s0;                    // circShift(img, 0, roll): dst(:, c) = src(:, c - roll)
    const size_t n = (size_t)R * C;
    a0[(size_t)j * n + i] = (float)images[(size_t)jb.key0 * n + (size_t)r * C + s0] * (float)(1.0 / 255.0);
    a1[(size_t)j * n + i] = (float)images[(size_t)jb.key1 * n + i] * (float)(1.0 / 255.0);
}

__global__ void fm_to_complex_kernel(const float *src, double2 *dst, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[(size_t)blockIdx.y * n + i] = make_double2((double)src[(size_t)blockIdx.y * n + i], 0.0);
}

// out[l][k] = sum_m in[l][m] w^(k m): `lines` lines of n elements (element stride es, line stride ls); sgn -1 forward, +1 inverse.
// One thread per output; consecutive threads run along the dimension that is contiguous in memory.
__global__ __launch_bounds__(256) void fm_dft_lines_kernel(const double2 *in, double2 *out, int n, int es, int lines, int ls, int sgn,
                                                           const double *wc, const double *ws, size_t job_stride)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * lines) return;
    int k, l;
    if (es == 1) { l = t / n; k = t - l * n; } else { k = t / lines; l = t - k * lines; }
    const double2 *src = in + (size_t)blockIdx.y * job_stride + (size_t)l * ls;
    double re = 0.0, im = 0.0;
    int tw = 0;                                                                // (k * m) mod n
    for (int m = 0; m < n; ++m) {
        const double2 x = src[(size_t)m * es];
        const double c = wc[tw], s = sgn < 0 ? -ws[tw] : ws[tw];
        re = re + (x.x * c - x.y * s);
        im = im + (x.x * s + x.y * c);
        tw += k; tw = tw >= n ? tw - n : tw;
    }
    out[(size_t)blockIdx.y * job_stride + (size_t)l * ls + (size_t)k * es] = make_double2(re, im);
}

// recomb (quadrant swap), / (M N), magnitude on the float planes, highpass: f = |F| * h  (D.h:719-764, 856-872)
__global__ void fm_mag_highpass_kernel(const double2 *F, const float *hp, int R, int C, float *f)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * C) return;
    const int r = i / C, c = i - r * C;
    const size_t n = (size_t)R * C;
    const double2 v = F[(size_t)blockIdx.y * n + (size_t)((r + R / 2) % R) * C + (c + C / 2) % C];
    const float mn = (float)(R * C);
    const float re = (float)v.x / mn, im = (float)v.y / mn;
    f[(size_t)blockIdx.y * n + i] = sqrtf(re * re + im * im) * hp[i];
}

__device__ __forceinline__ float fm_bilinear32(const float *src, int R, int C, int sx, int sy)
{
    const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
    const float ax = (float)fx * (1.0f / 32.0f), ay = (float)fy * (1.0f / 32.0f);
    const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
    auto tap = [&](int yy, int xx) { return (yy >= 0 && yy < R && xx >= 0 && xx < C) ? src[(size_t)yy * C + xx] : 0.0f; };
    return ((tap(iy, ix) * w00 + tap(iy, ix + 1) * w01) + tap(iy + 1, ix) * w10) + tap(iy + 1, ix + 1) * w11;
}

// cv::remap through the log-polar map (fixed-point source positions from the host)
__global__ void fm_remap_kernel(const float *f, const int2 *map, int R, int C, float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * C) return;
    const size_t n = (size_t)R * C;
    const int2 p = map[i];
    out[(size_t)blockIdx.y * n + i] = fm_bilinear32(f + (size_t)blockIdx.y * n, R, C, p.x, p.y);
}

// F1 conj(F2) / (|F1 conj(F2)| + FLT_EPSILON)
__global__ void fm_crosspower_kernel(const double2 *A, const double2 *B, double2 *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double2 a = A[(size_t)blockIdx.y * n + i], b = B[(size_t)blockIdx.y * n + i];
    const double pr = a.x * b.x + a.y * b.y, pi = a.y * b.x - a.x * b.y;
    const double mag = sqrt(pr * pr + pi * pi) + (double)FLT_EPSILON;
    out[(size_t)blockIdx.y * n + i] = make_double2(pr / mag, pi / mag);
}

// quadrant swap, first maximum in row-major order, 5 x 5 weighted centroid: one workgroup per job; out = (cols / 2 - cx, rows / 2 - cy)
__global__ __launch_bounds__(256) void fm_peak_kernel(const double2 *Cr, int R, int C, double2 *out)
{
    const size_t n = (size_t)R * C;
    const double2 *src = Cr + (size_t)blockIdx.x * n;
    auto at = [&](int i, int j) { return (float)src[(size_t)((i + R / 2) % R) * C + (j + C / 2) % C].x; };
    float best = -INFINITY; int bidx = 0x7fffffff;
    for (int i = threadIdx.x; i < R * C; i += blockDim.x) {
        const float v = at(i / C, i % C);
        if (v > best) { best = v; bidx = i; }                                  // (ascending i per thread: the first maximum it meets)
    }
    __shared__ float sv[256]; __shared__ int si[256];
    sv[threadIdx.x] = best; si[threadIdx.x] = bidx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float ov = sv[threadIdx.x + off]; const int oi = si[threadIdx.x + off];
            if (ov > sv[threadIdx.x] || (ov == sv[threadIdx.x] && oi < si[threadIdx.x])) { sv[threadIdx.x] = ov; si[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const int pr = si[0] == 0x7fffffff ? 0 : si[0] / C, pc = si[0] == 0x7fffffff ? 0 : si[0] % C;
    int minr = pr - 2, maxr = pr + 2, minc = pc - 2, maxc = pc + 2;
    minr = minr < 0 ? 0 : minr; minc = minc < 0 ? 0 : minc; maxr = maxr > R - 1 ? R - 1 : maxr; maxc = maxc > C - 1 ? C - 1 : maxc;
    double sum = 0.0, cx = 0.0, cy = 0.0;
    for (int y = minr; y <= maxr; ++y)
        for (int x = minc; x <= maxc; ++x) {
            const double v = (double)at(y, x);
            cx += (double)x * v; cy += (double)y * v; sum += v;
        }
    cx /= sum; cy /= sum;
    out[blockIdx.x] = make_double2((double)C / 2.0 - cx, (double)R / 2.0 - cy);
}

// cv::warpAffine with the inverted matrix in fixed point (AB_BITS 10, INTER_BITS 5): mats[j][6]
__global__ void fm_warp_kernel(const float *src, const double *mats, int R, int C, float *dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * C) return;
    const size_t n = (size_t)R * C;
    const double *M = mats + (size_t)blockIdx.y * 6;
    const int y = i / C, x = i - y * C;
    const int X0 = __double2int_rn((M[1] * (double)y + M[2]) * 1024.0) + 16, Y0 = __double2int_rn((M[4] * (double)y + M[5]) * 1024.0) + 16;
    const int X = (X0 + __double2int_rn(M[0] * (double)x * 1024.0)) >> 5, Y = (Y0 + __double2int_rn(M[3] * (double)x * 1024.0)) >> 5;
    dst[(size_t)blockIdx.y * n + i] = fm_bilinear32(src + (size_t)blockIdx.y * n, R, C, X, Y);
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
    // FFT shift estimate (logPolarFFTTemplateMatch): tables made at creation, work buffers for fm_cap jobs
    double *d_wcR = nullptr, *d_wsR = nullptr, *d_wcC = nullptr, *d_wsC = nullptr; float *d_hp = nullptr; int2 *d_lpmap = nullptr; float log_base = 0.f;
    bool fm_ok = false;                                        // even rows / cols (the restatement's quadrant swap)
    size_t fm_cap = 0;
    FftJob *d_fjobs = nullptr; float *d_fa0 = nullptr, *d_fa1 = nullptr, *d_ff = nullptr, *d_flp0 = nullptr, *d_flp1 = nullptr, *d_frs = nullptr;
    double2 *d_fw0 = nullptr, *d_fw1 = nullptr, *d_fw2 = nullptr, *d_fres = nullptr; double *d_fmats = nullptr;
    int *d_rolls = nullptr; size_t rolls_cap = 0;
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

int hamming_jobs_locked(scl_iris *h, int key1, const int *cand, const int *shifts, int n, int per, float *dis, int *bias, bool window, const int *rolls2 = nullptr)
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
    if (rolls2) {
        if ((size_t)n > h->rolls_cap) {
            if (h->d_rolls) (void)hipFree(h->d_rolls);
            h->d_rolls = nullptr; h->rolls_cap = 0;
            int rc = ialloc(h, &h->d_rolls, (size_t)n + 64);
            if (rc) return rc;
            h->rolls_cap = (size_t)n + 64;
        }
        IRIS_HIP(h, hipMemcpyAsync(h->d_rolls, rolls2, sizeof(int) * n, hipMemcpyHostToDevice, h->stream));
    }
    const size_t fw = (size_t)h->cfg.cols * h->words;
    hipLaunchKernelGGL(iris_hamming_kernel, dim3((unsigned)jobs), dim3(64), 0, h->stream, h->d_T, h->d_M, fw, key1, h->d_cand, h->d_shifts, per,
                       h->cfg.cols, h->words, h->trows, h->d_diff, h->d_total, rolls2 ? h->d_rolls : (const int *)nullptr);
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

// fftMatch for J jobs: centre x of the RotatedRect as a float (D.h:927-932); ok[j] = 0 where the reference prints "Images are not
// compatible" and returns an empty rectangle (centre 0).  Two round trips to the host: the rotation / scale of the log-polar stage
// and the translation.
int fft_match_jobs_locked(scl_iris *h, const FftJob *jobs, int J, float *center_x, int *ok_out)
{
    if (J <= 0) return SCL_OK;
    if (!h->fm_ok) return ifail(h, SCL_ERR_UNSUPPORTED, "the FFT shift estimate takes even rows and columns only");
    const int R = h->cfg.rows, C = h->cfg.cols;
    const size_t n = (size_t)R * C;
    if ((size_t)J > h->fm_cap) {
        for (void **p : {(void **)&h->d_fjobs, (void **)&h->d_fa0, (void **)&h->d_fa1, (void **)&h->d_ff, (void **)&h->d_flp0, (void **)&h->d_flp1, (void **)&h->d_frs,
                         (void **)&h->d_fw0, (void **)&h->d_fw1, (void **)&h->d_fw2, (void **)&h->d_fres, (void **)&h->d_fmats}) { if (*p) (void)hipFree(*p); *p = nullptr; }
        h->fm_cap = 0;
        const size_t cap = (size_t)J + 8;
        int rc;
        if ((rc = ialloc(h, &h->d_fjobs, cap)) || (rc = ialloc(h, &h->d_fa0, cap * n)) || (rc = ialloc(h, &h->d_fa1, cap * n)) || (rc = ialloc(h, &h->d_ff, cap * n)) ||
            (rc = ialloc(h, &h->d_flp0, cap * n)) || (rc = ialloc(h, &h->d_flp1, cap * n)) || (rc = ialloc(h, &h->d_frs, cap * n)) || (rc = ialloc(h, &h->d_fw0, cap * n)) ||
            (rc = ialloc(h, &h->d_fw1, cap * n)) || (rc = ialloc(h, &h->d_fw2, cap * n)) || (rc = ialloc(h, &h->d_fres, cap)) || (rc = ialloc(h, &h->d_fmats, cap * 6))) return rc;
        h->fm_cap = cap;
    }
    hipStream_t st = h->stream;
    const dim3 ge((unsigned)((n + 255) / 256), (unsigned)J), blk(256);
    IRIS_HIP(h, hipMemcpyAsync(h->d_fjobs, jobs, sizeof(FftJob) * (size_t)J, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(fm_stage_kernel, ge, blk, 0, st, h->d_images, h->d_fjobs, R, C, h->d_fa0, h->d_fa1);
    // 2-D DFT of a float image: rows, then columns; result in w_out, w_tmp as scratch
    auto dft2 = [&](const float *src, double2 *w_out, double2 *w_tmp) {
        hipLaunchKernelGGL(fm_to_complex_kernel, ge, blk, 0, st, src, w_tmp, (int)n);
        hipLaunchKernelGGL(fm_dft_lines_kernel, ge, blk, 0, st, w_tmp, w_out, C, 1, R, C, -1, h->d_wcC, h->d_wsC, n);
        hipLaunchKernelGGL(fm_dft_lines_kernel, ge, blk, 0, st, w_out, w_tmp, R, C, C, 1, -1, h->d_wcR, h->d_wsR, n);
        (void)hipMemcpyAsync(w_out, w_tmp, sizeof(double2) * n * (size_t)J, hipMemcpyDeviceToDevice, st);
    };
    auto phase_correlate = [&](const float *s1, const float *s2) {             // -> d_fres[j] = (tx, ty)
        dft2(s1, h->d_fw0, h->d_fw2);
        dft2(s2, h->d_fw1, h->d_fw2);
        hipLaunchKernelGGL(fm_crosspower_kernel, ge, blk, 0, st, h->d_fw0, h->d_fw1, h->d_fw2, (int)n);
        hipLaunchKernelGGL(fm_dft_lines_kernel, ge, blk, 0, st, h->d_fw2, h->d_fw0, R, C, C, 1, +1, h->d_wcR, h->d_wsR, n);   // inverse: columns, then rows
        hipLaunchKernelGGL(fm_dft_lines_kernel, ge, blk, 0, st, h->d_fw0, h->d_fw1, C, 1, R, C, +1, h->d_wcC, h->d_wsC, n);
        hipLaunchKernelGGL(fm_peak_kernel, dim3((unsigned)J), blk, 0, st, h->d_fw1, R, C, h->d_fres);
    };
    auto logpolar = [&](const float *img, float *lp) {
        dft2(img, h->d_fw0, h->d_fw1);
        hipLaunchKernelGGL(fm_mag_highpass_kernel, ge, blk, 0, st, h->d_fw0, h->d_hp, R, C, h->d_ff);
        hipLaunchKernelGGL(fm_remap_kernel, ge, blk, 0, st, h->d_ff, h->d_lpmap, R, C, lp);
    };
    logpolar(h->d_fa0, h->d_flp0);
    logpolar(h->d_fa1, h->d_flp1);
    phase_correlate(h->d_flp1, h->d_flp0);
    IRIS_HIP(h, hipGetLastError());
    std::vector<double2> res((size_t)J);
    IRIS_HIP(h, hipMemcpyAsync(res.data(), h->d_fres, sizeof(double2) * (size_t)J, hipMemcpyDeviceToHost, st));
    IRIS_HIP(h, hipStreamSynchronize(st));
    // rotation and scale -> the inverted affine map of warpAffine (D.h:884-912; the expressions of iriso_fft_match)
    std::vector<double> mats((size_t)J * 6);
    std::vector<int> ok((size_t)J, 1);
    for (int j = 0; j < J; ++j) {
        const double rx = res[(size_t)j].x, ry = res[(size_t)j].y;
        float angle = (float)(180.0 * ry / (double)R);
        float scale = (float)std::pow((double)h->log_base, rx);
        if (scale > 1.8f) {
            angle = (float)(-180.0 * ry / (double)R);
            scale = (float)(1.0 / std::pow((double)h->log_base, rx));
            if (scale > 1.8f) ok[(size_t)j] = 0;
        }
        if (angle < -90.0f) angle += 180.0f; else if (angle > 90.0f) angle -= 180.0f;
        const double ang = (double)angle * M_PI / 180.0, sc = 1.0 / (double)scale;
        const double alpha = std::cos(ang) * sc, beta = std::sin(ang) * sc, pcx = (double)(float)(C / 2), pcy = (double)(float)(R / 2);
        double M[6] = {alpha, beta, (1.0 - alpha) * pcx - beta * pcy, -beta, alpha, beta * pcx + (1.0 - alpha) * pcy};
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0.0 ? 1.0 / D : 0.0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
        const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
        for (int k = 0; k < 6; ++k) mats[(size_t)j * 6 + k] = M[k];
    }
    IRIS_HIP(h, hipMemcpyAsync(h->d_fmats, mats.data(), sizeof(double) * mats.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(fm_warp_kernel, ge, blk, 0, st, h->d_fa1, h->d_fmats, R, C, h->d_frs);
    phase_correlate(h->d_frs, h->d_fa0);
    IRIS_HIP(h, hipGetLastError());
    IRIS_HIP(h, hipMemcpyAsync(res.data(), h->d_fres, sizeof(double2) * (size_t)J, hipMemcpyDeviceToHost, st));
    IRIS_HIP(h, hipStreamSynchronize(st));
    for (int j = 0; j < J; ++j) {
        center_x[j] = ok[(size_t)j] ? (float)(res[(size_t)j].x + (double)(C / 2)) : 0.0f;
        if (ok_out) ok_out[j] = ok[(size_t)j];
    }
    return SCL_OK;
}

// compare(cur, candidate) for m candidates (D.h:964-1024): the FFT estimate(s), then the Hamming windows of five shifts around
// them -- the first pass against the candidate as it is, the second against the candidate turned by 180 columns -- as match_num says
int compare_jobs_locked(scl_iris *h, int cur, const int *cand, int m, float *dis, int *bias)
{
    const int mn = h->cfg.match_num, C = h->cfg.cols;
    const bool first = mn == 2 || mn == 0, second = mn == 2 || mn == 1;
    std::vector<FftJob> jobs;
    for (int c = 0; c < m; ++c) {
        if (first) jobs.push_back(FftJob{cand[c], 0, cur});
        if (second) jobs.push_back(FftJob{cand[c], 180, cur});                     // circShift(img2.img, 0, 180), D.h:978
    }
    std::vector<float> cx(jobs.size());
    int rc = fft_match_jobs_locked(h, jobs.data(), (int)jobs.size(), cx.data(), nullptr);
    if (rc) return rc;
    const int per = 5, J = (int)jobs.size();
    std::vector<int> jc((size_t)J), jr((size_t)J), shifts((size_t)J * per), jb((size_t)J);
    std::vector<float> jd((size_t)J);
    for (int j = 0; j < J; ++j) {
        jc[(size_t)j] = jobs[(size_t)j].key0; jr[(size_t)j] = jobs[(size_t)j].roll0;
        const int est = (int)(cx[(size_t)j] - (float)(C / 2));                     // int = float - int, D.h:969 / 980
        for (int t = 0; t < per; ++t) shifts[(size_t)j * per + t] = est - 2 + t;   // D.h:937
    }
    rc = hamming_jobs_locked(h, cur, jc.data(), shifts.data(), J, per, jd.data(), jb.data(), true, jr.data());
    if (rc) return rc;
    int j = 0;
    for (int c = 0; c < m; ++c) {
        float d1 = NAN, d2 = 0.0f; int b1 = -1, b2 = 0;
        if (first) { d1 = jd[(size_t)j]; b1 = jb[(size_t)j]; ++j; }
        if (second) { d2 = jd[(size_t)j]; b2 = jb[(size_t)j]; ++j; }
        if (mn == 2) { if (d1 < d2) { dis[c] = d1; bias[c] = b1; } else { dis[c] = d2; bias[c] = (b2 + 180) % 360; } }   // D.h:986-997
        else if (mn == 1) { dis[c] = d2; bias[c] = (b2 + 180) % 360; }
        else { dis[c] = d1; bias[c] = b1; }
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
    std::vector<int> cand((size_t)m), bias((size_t)m);
    std::vector<float> dis((size_t)m);
    for (int c = 0; c < m; ++c) cand[(size_t)c] = list[(size_t)pos[(size_t)c]];
    int rc;
    if (h->cfg.shift_search == 1) {                                          // every column shift (a superset of compare()'s windows)
        std::vector<int> shifts((size_t)m * N);
        for (int c = 0; c < m; ++c) for (int j = 0; j < N; ++j) shifts[(size_t)c * N + j] = j;
        rc = hamming_jobs_locked(h, cur, cand.data(), shifts.data(), m, N, dis.data(), bias.data(), false);
    } else {
        rc = compare_jobs_locked(h, cur, cand.data(), m, dis.data(), bias.data());   // compare(), D.h:964-1024
    }
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
    c->knn_exclude_eps = FLT_EPSILON; c->wire_decode = 0; c->shift_search = 0;
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
        cfg->num_candidates > 4096 || cfg->num_exclude_recent < 0 || cfg->match_num < 0 || cfg->match_num > 2 || !(cfg->knn_exclude_eps >= 0.0f) ||
        cfg->shift_search < 0 || cfg->shift_search > 1)
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
    // tables of the FFT shift estimate (logPolarFFTTemplateMatch, D.h:719-925): the expressions of oracle/iris_oracle.c's
    // iriso_fft_match, evaluated here with the same C library -- twiddles, highpass, log-polar map in OpenCV's fixed point
    h->fm_ok = !(cfg->rows & 1) && !(cfg->cols & 1) && cfg->rows >= 6 && cfg->cols >= 6;
    if (h->fm_ok) {
        const int R = cfg->rows, C = cfg->cols;
        std::vector<double> wcR((size_t)R), wsR((size_t)R), wcC((size_t)C), wsC((size_t)C);
        for (int k = 0; k < R; ++k) { wcR[(size_t)k] = std::cos(2.0 * M_PI * (double)k / (double)R); wsR[(size_t)k] = std::sin(2.0 * M_PI * (double)k / (double)R); }
        for (int k = 0; k < C; ++k) { wcC[(size_t)k] = std::cos(2.0 * M_PI * (double)k / (double)C); wsC[(size_t)k] = std::sin(2.0 * M_PI * (double)k / (double)C); }
        std::vector<float> a((size_t)R), b((size_t)C), hp(cells);
        { const float step = (float)(M_PI / (double)R); float val = (float)(-M_PI * 0.5); for (int i = 0; i < R; ++i) { a[(size_t)i] = cosf(val); val += step; } }
        { const float step = (float)(M_PI / (double)C); float val = (float)(-M_PI * 0.5); for (int j = 0; j < C; ++j) { b[(size_t)j] = cosf(val); val += step; } }
        for (int i = 0; i < R; ++i)
            for (int j = 0; j < C; ++j) { const float t = a[(size_t)i] * b[(size_t)j]; hp[(size_t)i * C + j] = (1.0f - t) * (2.0f - t); }
        std::vector<int2> map(cells);
        const float radii = (float)C, angles = (float)R, cxf = (float)(C / 2), cyf = (float)(R / 2);
        const float ddx = (float)C - cxf, ddy = (float)R - cyf;
        const float d = (float)std::sqrt((double)ddx * (double)ddx + (double)ddy * (double)ddy);
        const float log_base = (float)std::pow(10.0, (double)(log10f(d) / radii));
        const float d_theta = (float)(M_PI / (double)angles);
        float theta = (float)(M_PI / 2.0);
        for (int i = 0; i < R; ++i) {
            for (int j = 0; j < C; ++j) {
                const float radius = powf(log_base, (float)j);
                const float x = radius * sinf(theta) + cxf, y = radius * cosf(theta) + cyf;
                map[(size_t)i * C + j] = make_int2((int)lrint((double)x * 32.0), (int)lrint((double)y * 32.0));
            }
            theta += d_theta;
        }
        h->log_base = log_base;
        if ((rc = ialloc(h, &h->d_wcR, (size_t)R)) || (rc = ialloc(h, &h->d_wsR, (size_t)R)) || (rc = ialloc(h, &h->d_wcC, (size_t)C)) || (rc = ialloc(h, &h->d_wsC, (size_t)C)) ||
            (rc = ialloc(h, &h->d_hp, cells)) || (rc = ialloc(h, &h->d_lpmap, cells))) return bail(rc);
        if (hipMemcpy(h->d_wcR, wcR.data(), sizeof(double) * R, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(h->d_wsR, wsR.data(), sizeof(double) * R, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_wcC, wcC.data(), sizeof(double) * C, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(h->d_wsC, wsC.data(), sizeof(double) * C, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_hp, hp.data(), sizeof(float) * cells, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(h->d_lpmap, map.data(), sizeof(int2) * cells, hipMemcpyHostToDevice) != hipSuccess)
            return bail(SCL_ERR_HIP);
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
                    (void *)h->d_diff, (void *)h->d_total, (void *)h->d_list, (void *)h->d_d2, (void *)h->d_wcR, (void *)h->d_wsR, (void *)h->d_wcC, (void *)h->d_wsC,
                    (void *)h->d_hp, (void *)h->d_lpmap, (void *)h->d_fjobs, (void *)h->d_fa0, (void *)h->d_fa1, (void *)h->d_ff, (void *)h->d_flp0, (void *)h->d_flp1,
                    (void *)h->d_frs, (void *)h->d_fw0, (void *)h->d_fw1, (void *)h->d_fw2, (void *)h->d_fres, (void *)h->d_fmats, (void *)h->d_rolls})
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

int scl_iris_fft_match(scl_iris *h, int key0, int roll0, int key1, float *center_x, int *compatible)
{
    if (!h || !center_x) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    if (key0 < 0 || key0 >= h->n || key1 < 0 || key1 >= h->n) return ifail(h, SCL_ERR_OUT_OF_RANGE, "fft_match: key out of range");
    const FftJob jb{key0, roll0, key1};
    return fft_match_jobs_locked(h, &jb, 1, center_x, compatible);
}

int scl_iris_compare(scl_iris *h, int key1, const int *cand, int n, float *dis, int *bias)
{
    if (!h || (n > 0 && (!cand || !dis || !bias)) || n < 0) return SCL_ERR_INVALID_ARG;
    if (n == 0) return SCL_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    (void)hipSetDevice(h->device);
    if (key1 < 0 || key1 >= h->n) return ifail(h, SCL_ERR_OUT_OF_RANGE, "compare: key1 out of range");
    for (int i = 0; i < n; ++i) if (cand[i] < 0 || cand[i] >= h->n) return ifail(h, SCL_ERR_OUT_OF_RANGE, "compare: candidate out of range");
    return compare_jobs_locked(h, key1, cand, n, dis, bias);
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
