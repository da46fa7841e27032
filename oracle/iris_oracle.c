/* iris_oracle.c -- see iris_oracle.h.  TEST INFRASTRUCTURE ONLY. */
#include "iris_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* atan2 for the two uses of D.h:547-549: float arguments, float result -- std::atan2(float, float) = the platform's atan2f.  On the
 * x86-64 glibc the reference is built against (2.35 here) that is sysdeps/ieee754/flt-32/e_atan2f.c: fdlibm's case analysis around
 * atanf(|y / x|) -- the float atan restated and pinned in sc_oracle.c (equal to libm on all 2^32 inputs) -- with the pi / pi_lo
 * corrections of the second and third quadrants, fp32 throughout.  Restated here operation by operation; oracle/tools/atan2f_check.c
 * compares it with libm's atan2f on 4e9 pairs (random bit patterns, every pair of special values, realistic coordinates): 0
 * differences; tests/test_iris.py checks a sample on every run.  (Rounds 2-4 evaluated both uses through one fp64 polynomial.) */
float sco_atanf_glibc(float x);     /* sc_oracle.c */
static inline unsigned int iris_f32_bits(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
static inline float iris_f32_from_bits(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
float iriso_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, zero = 0.0f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    float z;
    const int hx = (int)iris_f32_bits(x), ix = hx & 0x7fffffff, hy = (int)iris_f32_bits(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;                 /* x or y is NaN */
    if (hx == 0x3f800000) return sco_atanf_glibc(y);                      /* x = 1.0 */
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);                    /* 2 sign(x) + sign(y) */
    if (iy == 0) {                                                        /* y = 0 */
        switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;        /* x = 0 */
    if (ix == 0x7f800000) {                                               /* x is inf */
        if (iy == 0x7f800000) {
            switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; }
        } else {
            switch (m) { case 0: return zero; case 1: return -zero; case 2: return pi + tiny; default: return -pi - tiny; }
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;   /* y is inf */
    const int k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;                                /* |y / x| > 2^60 */
    else if (hx < 0 && k < -60) z = 0.0f;                                 /* |y| / x < -2^60 */
    else z = sco_atanf_glibc(fabsf(y / x));
    switch (m) {
    case 0: return z;                                                     /* atan(+, +) */
    case 1: return iris_f32_from_bits(iris_f32_bits(z) ^ 0x80000000u);    /* atan(-, +) */
    case 2: return pi - (z - pi_lo);                                      /* atan(+, -) */
    default: return (z - pi_lo) - pi;                                     /* atan(-, -) */
    }
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static int floor_to_int(double v)
{   /* (int)floor(v) as x86-64 evaluates it: NaN / out of range -> INT_MIN */
    double f = floor(v);
    if (!(f >= -2147483648.0 && f <= 2147483647.0)) return (-2147483647 - 1);
    return (int)f;
}

void iriso_make_image(const iriso_config *c, const void *pts, int n, int stride_bytes, uint8_t *image, float *rowkey)
{
    const int rows = c->rows, cols = c->cols;
    float *zmax = (float *)calloc((size_t)rows * cols, sizeof(float));            /* irisRowKeyMat = Zero, D.h:535 */
    memset(image, 0, (size_t)rows * cols);
    const double add = c->nscan == 16 ? 15.0 : 24.9;                              /* D.h:544 / 566 */
    const unsigned char *base = (const unsigned char *)pts;
    for (int p = 0; p < n && (c->nscan == 16 || c->nscan == 64); p++) {
        float v[3];
        memcpy(v, base + (size_t)p * (size_t)stride_bytes, sizeof v);
        const float dis = sqrtf(v[0] * v[0] + v[1] * v[1]);                       /* D.h:543 */
        const float arc = (float)((double)(iriso_atan2f(v[2], dis) * 180.0f) / M_PI + add);   /* float * float, / double, + double -> float */
        const float yaw = (float)((double)(iriso_atan2f(v[1], v[0]) * 180.0f) / M_PI + 180);
        const int q_dis = clampi(floor_to_int((double)dis), 0, rows - 1);
        const int q_arc = clampi(floor_to_int((double)(arc / 4.0f)), 0, 7);
        const int q_yaw = clampi(floor_to_int((double)yaw + 0.5), 0, cols - 1);
        image[(size_t)q_dis * cols + q_yaw] |= (uint8_t)(1 << q_arc);
        if (zmax[(size_t)q_dis * cols + q_yaw] < v[2]) zmax[(size_t)q_dis * cols + q_yaw] = v[2];
    }
    for (int r = 0; r < rows; r++) {                                              /* rowwise().mean(), float */
        float s = 0.0f;
        for (int k = 0; k < cols; k++) s += zmax[(size_t)r * cols + k];
        rowkey[r] = s / (float)cols;
    }
    free(zmax);
}

/* the one-sided log-Gabor transfer function of scale s, D.h:622-640 (float arithmetic like cv::log / pow / exp on Mat1f) */
static void log_gabor(const iriso_config *c, int s, float *g /* [cols] */)
{
    const int ndata = c->cols - (c->cols & 1);
    double wavelength = c->min_wavelength;
    for (int k = 0; k < s; k++) wavelength *= (double)c->mult;
    const double fo = 1.0 / wavelength;
    for (int i = 0; i < c->cols; i++) g[i] = 0.0f;
    for (int i = 0; i < ndata / 2 + 1; i++) {
        const float radius = i == 0 ? 1.0f : (float)i / (float)ndata;
        float t = logf((float)((double)radius / fo));
        t = t * t;
        const double denom = 2 * log((double)c->sigma_onf) * log((double)c->sigma_onf);
        g[i] = expf((float)((double)(-t) / denom));
    }
    g[0] = 0.0f;                                                                  /* D.h:640 */
}

void iriso_responses(const iriso_config *c, const uint8_t *image, double *resp)
{
    const int rows = c->rows, N = c->cols;
    const double TWO_PI = 6.283185307179586476925286766559;
    float *g = (float *)malloc(sizeof(float) * (size_t)N);
    double *hre = (double *)malloc(sizeof(double) * 2 * (size_t)N), *him = hre + N;
    for (int s = 0; s < c->nscale; s++) {
        log_gabor(c, s, g);
        /* response = idft(dft(x) * G), both unscaled (cv::dft / cv::idft without DFT_SCALE, D.h:651-653)
         *          = x (*) h circularly, h[n] = sum_k G[k] e^{+2 pi i k n / N} */
        for (int n = 0; n < N; n++) {
            double re = 0, im = 0;
            for (int k = 0; k < N; k++) {
                if (g[k] == 0.0f) continue;
                const double a = TWO_PI * (double)(((long long)k * n) % N) / (double)N;
                re += (double)g[k] * cos(a); im += (double)g[k] * sin(a);
            }
            hre[n] = re; him[n] = im;
        }
        for (int r = 0; r < rows; r++)
            for (int n = 0; n < N; n++) {
                double re = 0, im = 0;
                for (int m = 0; m < N; m++) {
                    const double x = (double)image[(size_t)r * N + m];
                    if (x == 0.0) continue;
                    int d = n - m; if (d < 0) d += N;
                    re += x * hre[d]; im += x * him[d];
                }
                double *o = resp + (((size_t)s * rows + r) * N + n) * 2;
                o[0] = re; o[1] = im;
            }
    }
    free(g); free(hre);
}

void iriso_encode(const iriso_config *c, const uint8_t *image, uint8_t *T, uint8_t *M)
{
    const int rows = c->rows, N = c->cols, ns = c->nscale;
    double *resp = (double *)malloc(sizeof(double) * 2 * (size_t)ns * rows * N);
    iriso_responses(c, image, resp);
    for (int s = 0; s < ns; s++)
        for (int r = 0; r < rows; r++)
            for (int n = 0; n < N; n++) {
                const double *o = resp + (((size_t)s * rows + r) * N + n) * 2;
                const float re = (float)o[0], im = (float)o[1];                   /* the reference's planes are float */
                const float mag = sqrtf(re * re + im * im);
                const size_t a = ((size_t)s * rows + r) * N + n, b = ((size_t)(s + ns) * rows + r) * N + n;   /* vconcat order, D.h:669-678 */
                T[a] = re > 0 ? 255 : 0; T[b] = im > 0 ? 255 : 0;
                M[a] = mag < 0.0001f ? 255 : 0; M[b] = M[a];
            }
    free(resp);
}

static void hamming_at(const iriso_config *c, const uint8_t *T1, const uint8_t *M1, const uint8_t *T2, const uint8_t *M2, int shift,
                       int *bits_diff, int *total_bits)
{
    const int R = 2 * c->nscale * c->rows, N = c->cols;
    int sh = shift % N; if (sh < 0) sh += N;                                      /* circColShift, D.h:581-592 */
    int mask_bits = 0, diff = 0;
    for (int r = 0; r < R; r++)
        for (int k = 0; k < N; k++) {
            int src = k - sh; if (src < 0) src += N;                              /* dst(:, k) = src(:, k - shift) */
            const uint8_t t1 = T1[(size_t)r * N + src], m1 = M1[(size_t)r * N + src];
            const uint8_t mask = m1 | M2[(size_t)r * N + k];
            if (mask) { mask_bits++; continue; }
            if (t1 ^ T2[(size_t)r * N + k]) diff++;
        }
    *bits_diff = diff; *total_bits = R * N - mask_bits;
}

void iriso_hamming(const iriso_config *c, const uint8_t *T1, const uint8_t *M1, const uint8_t *T2, const uint8_t *M2, int scale, float *dis, int *bias)
{
    *dis = NAN; *bias = -1;
    for (int shift = scale - 2; shift <= scale + 2; shift++) {
        int diff, total;
        hamming_at(c, T1, M1, T2, M2, shift, &diff, &total);
        if (total == 0) { *dis = NAN; continue; }                                 /* D.h:948-951 */
        const float cur = (float)diff / (float)total;
        if (cur < *dis || isnan(*dis)) { *dis = cur; *bias = shift; }
    }
}

void iriso_hamming_all(const iriso_config *c, const uint8_t *T1, const uint8_t *M1, const uint8_t *T2, const uint8_t *M2, float *dis, int *bias)
{
    *dis = NAN; *bias = -1;
    for (int shift = 0; shift < c->cols; shift++) {
        int diff, total;
        hamming_at(c, T1, M1, T2, M2, shift, &diff, &total);
        if (total == 0) continue;
        const float cur = (float)diff / (float)total;
        if (cur < *dis || isnan(*dis)) { *dis = cur; *bias = shift; }
    }
}

/* ---- logPolarFFTTemplateMatch (D.h:793-925) and compare() (D.h:964-1024) ------------------------------------------------------
 * OpenCV is absent: every cv:: call of the chain is restated from the algorithm OpenCV documents / publishes, with float
 * storage where the reference's Mats are CV_32F and fp64 accumulation inside the transforms:
 *   cv::dft / idft      the direct DFT (forward and inverse unscaled), fp64 sums in index order, twiddles cos / sin(2 pi k / n)
 *   cv::magnitude       sqrtf(re^2 + im^2) on the float planes
 *   cv::remap           INTER_LINEAR with OpenCV's fixed-point source coordinates (1/32 pixel: cvRound(x * 32)), the four taps
 *                       weighted (1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx fy in float, BORDER_CONSTANT 0 per tap
 *   cv::phaseCorrelate  no window; F1 conj(F2) / (|F1 conj(F2)| + FLT_EPSILON) on the full complex spectrum; inverse DFT;
 *                       quadrant swap; first maximum in row-major order; 5 x 5 weighted centroid clipped to the image,
 *                       fp64 sums; result = (cols / 2, rows / 2) - centroid
 *   getRotationMatrix2D / warpAffine   the inverse map in fixed point (AB_BITS = 10, INTER_BITS = 5), bilinear taps as remap
 * What is NOT reproduced (cannot be known without the library): the float rounding of OpenCV's mixed-radix FFT, the packed
 * (CCS) spectrum layout of phaseCorrelate's helpers including their handling of the DC and Nyquist terms, SIMD evaluation
 * orders.  PARITY UNPINNED.  The transcendental calls (cos, sin, cosf, sinf, powf, log10f, pow) go to this host's libm, as the
 * reference's do; the engine evaluates the same expressions on the host and ships the tables to the device, so the two
 * agree bit for bit on the same machine (tests/test_iris_fftmatch.py). */

static int iround(double v) { return (int)lrint(v); }                                /* cvRound: round half to even */

typedef struct { double re, im; } cplx;

/* out[l][k] = sum_n in[l][n] w^(k n), one line of `n` elements at stride `es`, `lines` lines at stride `ls`; sign -1 forward, +1 inverse */
static void dft_lines(const cplx *in, cplx *out, int n, int es, int lines, int ls, int sign)
{
    double *wc = (double *)malloc(sizeof(double) * 2 * (size_t)n), *ws = wc + n;
    for (int k = 0; k < n; k++) { wc[k] = cos(2.0 * M_PI * (double)k / (double)n); ws[k] = sin(2.0 * M_PI * (double)k / (double)n); }
    for (int l = 0; l < lines; l++)
        for (int k = 0; k < n; k++) {
            double re = 0.0, im = 0.0;
            int t = 0;                                                             /* (k * m) mod n */
            for (int m = 0; m < n; m++) {
                const cplx x = in[(size_t)l * ls + (size_t)m * es];
                const double c = wc[t], s = sign < 0 ? -ws[t] : ws[t];
                re = re + (x.re * c - x.im * s);
                im = im + (x.re * s + x.im * c);
                t += k; if (t >= n) t -= n;
            }
            out[(size_t)l * ls + (size_t)k * es].re = re; out[(size_t)l * ls + (size_t)k * es].im = im;
        }
    free(wc);
}

static void dft2(const float *src, cplx *dst, cplx *tmp, int R, int C, int sign_unused)
{
    (void)sign_unused;
    for (int i = 0; i < R * C; i++) { tmp[i].re = (double)src[i]; tmp[i].im = 0.0; }
    dft_lines(tmp, dst, C, 1, R, C, -1);                                           /* rows */
    dft_lines(dst, tmp, R, C, C, 1, -1);                                           /* columns */
    memcpy(dst, tmp, sizeof(cplx) * (size_t)R * C);
}

/* bilinear tap sum at fixed-point source position (sx, sy) in 1/32 pixels */
static float bilinear32(const float *src, int R, int C, int sx, int sy)
{
    const int ix = sx >> 5, iy = sy >> 5, fx = sx & 31, fy = sy & 31;
    const float ax = (float)fx * (1.0f / 32.0f), ay = (float)fy * (1.0f / 32.0f);
    const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
#define TAP(yy, xx) (((yy) >= 0 && (yy) < R && (xx) >= 0 && (xx) < C) ? src[(size_t)(yy) * C + (xx)] : 0.0f)
    return ((TAP(iy, ix) * w00 + TAP(iy, ix + 1) * w01) + TAP(iy + 1, ix) * w10) + TAP(iy + 1, ix + 1) * w11;
#undef TAP
}

/* cv::phaseCorrelate(src1, src2) without window: translation (x, y) */
static void phase_correlate(const float *s1, const float *s2, int R, int C, double *tx, double *ty, cplx *w0, cplx *w1, cplx *w2)
{
    dft2(s1, w0, w2, R, C, -1);
    dft2(s2, w1, w2, R, C, -1);
    for (int i = 0; i < R * C; i++) {
        const double pr = w0[i].re * w1[i].re + w0[i].im * w1[i].im, pi = w0[i].im * w1[i].re - w0[i].re * w1[i].im;
        const double mag = sqrt(pr * pr + pi * pi) + (double)FLT_EPSILON;
        w2[i].re = pr / mag; w2[i].im = pi / mag;
    }
    dft_lines(w2, w0, R, C, C, 1, +1);                                             /* inverse: columns, then rows; unscaled (cv::idft without DFT_SCALE) */
    dft_lines(w0, w1, C, 1, R, C, +1);
    /* quadrant swap (even sizes: a circular shift by half), first maximum in row-major order of the shifted array */
    float best = -INFINITY; int pr_ = 0, pc_ = 0;
    float *S = (float *)malloc(sizeof(float) * (size_t)R * C);
    for (int i = 0; i < R; i++)
        for (int j = 0; j < C; j++) S[(size_t)i * C + j] = (float)w1[(size_t)((i + R / 2) % R) * C + (j + C / 2) % C].re;
    for (int i = 0; i < R; i++)
        for (int j = 0; j < C; j++) if (S[(size_t)i * C + j] > best) { best = S[(size_t)i * C + j]; pr_ = i; pc_ = j; }
    int minr = pr_ - 2, maxr = pr_ + 2, minc = pc_ - 2, maxc = pc_ + 2;
    if (minr < 0) minr = 0;
    if (minc < 0) minc = 0;
    if (maxr > R - 1) maxr = R - 1;
    if (maxc > C - 1) maxc = C - 1;
    double sum = 0.0, cx = 0.0, cy = 0.0;
    for (int y = minr; y <= maxr; y++)
        for (int x = minc; x <= maxc; x++) {
            const double v = (double)S[(size_t)y * C + x];
            cx += (double)x * v; cy += (double)y * v; sum += v;
        }
    cx /= sum; cy /= sum;
    free(S);
    *tx = (double)C / 2.0 - cx; *ty = (double)R / 2.0 - cy;
}

/* forwardFFT + magnitude + highpass + logpolar (D.h:719-790, 866-882) of an image given as floats */
static void highpassed_logpolar(const float *img, int R, int C, float *out, float *log_base_out, cplx *w0, cplx *w1)
{
    dft2(img, w0, w1, R, C, -1);
    float *f = (float *)malloc(sizeof(float) * (size_t)R * C);
    /* recomb (quadrant swap), / (M N), magnitude on the float planes */
    const float mn = (float)(R * C);
    for (int i = 0; i < R; i++)
        for (int j = 0; j < C; j++) {
            const cplx v = w0[(size_t)((i + R / 2) % R) * C + (j + C / 2) % C];
            const float re = (float)v.re / mn, im = (float)v.im / mn;
            f[(size_t)i * C + j] = sqrtf(re * re + im * im);
        }
    /* highpass (D.h:740-764): float accumulation of the angle, cosf, (1 - a b)(2 - a b) */
    float *a = (float *)malloc(sizeof(float) * (size_t)(R + C)), *b = a + R;
    { const float step = (float)(M_PI / (double)R); float val = (float)(-M_PI * 0.5); for (int i = 0; i < R; i++) { a[i] = cosf(val); val += step; } }
    { const float step = (float)(M_PI / (double)C); float val = (float)(-M_PI * 0.5); for (int j = 0; j < C; j++) { b[j] = cosf(val); val += step; } }
    for (int i = 0; i < R; i++)
        for (int j = 0; j < C; j++) {
            const float t = a[i] * b[j];
            f[(size_t)i * C + j] = f[(size_t)i * C + j] * ((1.0f - t) * (2.0f - t));
        }
    free(a);
    /* logpolar (D.h:766-791) */
    const float radii = (float)C, angles = (float)R, cxf = (float)(C / 2), cyf = (float)(R / 2);
    const float ddx = (float)C - cxf, ddy = (float)R - cyf;
    const float d = (float)sqrt((double)ddx * (double)ddx + (double)ddy * (double)ddy);
    const float log_base = (float)pow(10.0, (double)(log10f(d) / radii));
    const float d_theta = (float)(M_PI / (double)angles);
    float theta = (float)(M_PI / 2.0);
    for (int i = 0; i < R; i++) {
        for (int j = 0; j < C; j++) {
            const float radius = powf(log_base, (float)j);
            const float x = radius * sinf(theta) + cxf, y = radius * cosf(theta) + cyf;
            out[(size_t)i * C + j] = bilinear32(f, R, C, iround((double)x * 32.0), iround((double)y * 32.0));
        }
        theta += d_theta;
    }
    free(f);
    *log_base_out = log_base;
}

/* fftMatch(im0, im1) (D.h:927-932): the RotatedRect's centre x (float), from which compare() takes its shift; dbg (optional, 6):
 * rotation_and_scale.x, .y, angle, scale, tr.x, tr.y.  Even rows / cols only (returns -1 otherwise). */
int iriso_fft_match(int rows, int cols, const uint8_t *im0u, const uint8_t *im1u, float *center_x, double *dbg)
{
    const int R = rows, C = cols;
    if ((R & 1) || (C & 1) || R < 6 || C < 6) return -1;
    const size_t n = (size_t)R * C;
    float *im0 = (float *)malloc(sizeof(float) * 5 * n), *im1 = im0 + n, *lp0 = im1 + n, *lp1 = lp0 + n, *rs = lp1 + n;
    cplx *w0 = (cplx *)malloc(sizeof(cplx) * 3 * n), *w1 = w0 + n, *w2 = w1 + n;
    for (size_t i = 0; i < n; i++) { im0[i] = (float)im0u[i] * (float)(1.0 / 255.0); im1[i] = (float)im1u[i] * (float)(1.0 / 255.0); }   /* convertTo(CV_32FC1, 1 / 255) */
    float log_base;
    highpassed_logpolar(im0, R, C, lp0, &log_base, w0, w1);
    highpassed_logpolar(im1, R, C, lp1, &log_base, w0, w1);
    double rx, ry;
    phase_correlate(lp1, lp0, R, C, &rx, &ry, w0, w1, w2);
    float angle = (float)(180.0 * ry / (double)R);
    float scale = (float)pow((double)log_base, rx);
    int ok = 1;
    if (scale > 1.8f) {                                                            /* D.h:888-898 (the second phaseCorrelate repeats the first) */
        angle = (float)(-180.0 * ry / (double)R);
        scale = (float)(1.0 / pow((double)log_base, rx));
        if (scale > 1.8f) ok = 0;
    }
    double trx = 0.0, try_ = 0.0;
    if (ok) {
        if (angle < -90.0f) angle += 180.0f; else if (angle > 90.0f) angle -= 180.0f;
        /* getRotationMatrix2D(Point(cols / 2, rows / 2), angle, 1 / scale), then warpAffine's own inversion */
        const double ang = (double)angle * M_PI / 180.0, sc = 1.0 / (double)scale;
        const double alpha = cos(ang) * sc, beta = sin(ang) * sc, pcx = (double)(float)(C / 2), pcy = (double)(float)(R / 2);
        double M[6] = {alpha, beta, (1.0 - alpha) * pcx - beta * pcy, -beta, alpha, beta * pcx + (1.0 - alpha) * pcy};
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0.0 ? 1.0 / D : 0.0;
        const double A11 = M[4] * D, A22 = M[0] * D;
        M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22;
        const double b1 = -M[0] * M[2] - M[1] * M[5], b2 = -M[3] * M[2] - M[4] * M[5];
        M[2] = b1; M[5] = b2;
        for (int y = 0; y < R; y++) {
            const int X0 = iround((M[1] * (double)y + M[2]) * 1024.0) + 16, Y0 = iround((M[4] * (double)y + M[5]) * 1024.0) + 16;
            for (int x = 0; x < C; x++) {
                const int X = (X0 + iround(M[0] * (double)x * 1024.0)) >> 5, Y = (Y0 + iround(M[3] * (double)x * 1024.0)) >> 5;
                rs[(size_t)y * C + x] = bilinear32(im1, R, C, X, Y);
            }
        }
        phase_correlate(rs, im0, R, C, &trx, &try_, w0, w1, w2);
        *center_x = (float)(trx + (double)(C / 2));                                 /* rr.center = tr + Point2d(cols / 2, rows / 2) -> Point2f */
    } else {
        *center_x = 0.0f;                                                          /* cv::RotatedRect(): all zero */
    }
    if (dbg) { dbg[0] = rx; dbg[1] = ry; dbg[2] = (double)angle; dbg[3] = (double)scale; dbg[4] = trx; dbg[5] = try_; }
    free(im0); free(w0);
    return ok;
}

static void roll_cols(const uint8_t *src, uint8_t *dst, int R, int C, int shift)
{   /* circShift(src, 0, shift): dst(:, k) = src(:, k - shift) */
    for (int r = 0; r < R; r++)
        for (int k = 0; k < C; k++) { int s = (k - shift) % C; if (s < 0) s += C; dst[(size_t)r * C + k] = src[(size_t)r * C + s]; }
}

/* compare(img1, img2, &bias), D.h:964-1024.  match_num 2: both passes, 1: the candidate turned by half a revolution only, 0: the
 * first pass only.  T / M: (2 nscale rows) x cols.  shifts_out (optional, 2): the two FFT estimates (INT_MIN where not taken). */
void iriso_compare(const iriso_config *c, int match_num, const uint8_t *img1, const uint8_t *T1, const uint8_t *M1,
                   const uint8_t *img2, const uint8_t *T2, const uint8_t *M2, float *dis, int *bias, int *shifts_out)
{
    const int R = c->rows, C = c->cols, TR = 2 * c->nscale * c->rows, half = 180;   /* the reference turns by 180 COLUMNS whatever cols is (D.h:976-978) */
    float dis1 = NAN, dis2 = 0.0f; int bias1 = -1, bias2 = 0;
    if (shifts_out) { shifts_out[0] = -2147483647 - 1; shifts_out[1] = -2147483647 - 1; }
    if (match_num == 2 || match_num == 0) {
        float cx;
        iriso_fft_match(R, C, img2, img1, &cx, NULL);
        const int first_shift = (int)(cx - (float)(C / 2));                        /* int = float - int, D.h:969 */
        if (shifts_out) shifts_out[0] = first_shift;
        iriso_hamming(c, T1, M1, T2, M2, first_shift, &dis1, &bias1);
    }
    if (match_num == 2 || match_num == 1) {
        uint8_t *T2x = (uint8_t *)malloc((size_t)2 * TR * C + (size_t)R * C), *M2x = T2x + (size_t)TR * C, *i2x = M2x + (size_t)TR * C;
        roll_cols(T2, T2x, TR, C, half); roll_cols(M2, M2x, TR, C, half); roll_cols(img2, i2x, R, C, half);
        float cx;
        iriso_fft_match(R, C, i2x, img1, &cx, NULL);
        const int second_shift = (int)(cx - (float)(C / 2));
        if (shifts_out) shifts_out[1] = second_shift;
        iriso_hamming(c, T1, M1, T2x, M2x, second_shift, &dis2, &bias2);
        free(T2x);
    }
    if (match_num == 2) {
        if (dis1 < dis2) { *dis = dis1; *bias = bias1; }                           /* D.h:986-997 */
        else { *dis = dis2; *bias = (bias2 + 180) % 360; }
    } else if (match_num == 1) { *dis = dis2; *bias = (bias2 + 180) % 360; }
    else { *dis = dis1; *bias = bias1; }
}
