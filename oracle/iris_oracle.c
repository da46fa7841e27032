/* iris_oracle.c -- see iris_oracle.h.  TEST INFRASTRUCTURE ONLY. */
#include "iris_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

double sco_atan_pos(double x);      /* sc_oracle.c: the fixed atan shared with the GPU */

/* atan2 for the two uses of D.h:547-549: float arguments, float result (std::atan2(float, float)), evaluated through
 * the fixed fp64 atan so that CPU and GPU agree on every bin (the platform's atan2f differs in the last bit) */
static float iriso_atan2f(float y, float x)
{
    const double PI = 3.14159265358979323846;
    if (x != x || y != y) return NAN;
    if (y == 0.0f) return (x < 0.0f || (x == 0.0f && signbit(x))) ? (signbit(y) ? -(float)PI : (float)PI) : y;
    if (x == 0.0f) return y > 0.0f ? (float)(PI / 2) : (float)(-PI / 2);
    const double ay = fabs((double)y), ax = fabs((double)x);
    double a = isinf(ay) ? (isinf(ax) ? PI / 4 : PI / 2) : (isinf(ax) ? 0.0 : sco_atan_pos(ay / ax));
    if (x < 0.0f) a = PI - a;
    return (float)(y < 0.0f ? -a : a);
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
