/*
 * sc_oracle.c -- CPU restatement of the reference's Scan Context hot path.
 * TEST INFRASTRUCTURE ONLY (see sc_oracle.h).  Plain C, no Eigen: every
 * reduction is a sequential fp64 (or fp32 where the reference is fp32) sum in
 * the index order of the reference's loops; compiled with -ffp-contract=off.
 *
 * D.h = /root/reference/include/descriptor.h, NF = .../nanoflann.hpp
 */
#include "sc_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* Per-thread scratch of the reference-shaped evaluation.  The reference allocates a fresh Eigen matrix / vector at each
 * of these places (D.h:1498, 1520-1521, 1541-1542, 1559); the COPIES are kept here -- they are what the CPU baseline
 * prices -- but the allocator is not called per pair, per shift and per column: one buffer per call site and thread,
 * grown on demand and kept for the thread's life. */
#define SCO_TLS(name) static __thread double *name; static __thread size_t name##_cap
static double *tls_grow(double **p, size_t *cap, size_t n)
{
    if (*cap < n) { free(*p); *p = (double *)malloc(sizeof(double) * n); *cap = *p ? n : 0; }
    return *p;
}
SCO_TLS(tls_align); SCO_TLS(tls_cols); SCO_TLS(tls_vkeys); SCO_TLS(tls_space); SCO_TLS(tls_shifted); SCO_TLS(tls_fast);

void sco_default_config(sco_config *c)
{   /* D.h:1308-1316 */
    c->num_ring = 20;
    c->num_sector = 60;
    c->num_candidates = 3;
    c->dist_thres = 0.14;
    c->lidar_height = 1.65;
    c->max_radius = 80.0;
    c->num_exclude_recent = 100;
    c->tree_making_period = 10;
    c->search_ratio = 0.1;
    c->knn_exclude_eps = 0.0f;
}

/* ---------------------------------------------------------------------------
 * atan.  The reference calls std::atan(float) (D.h:1357 via `using namespace
 * std`, D.h:19), i.e. the platform's atanf.  On the x86-64 glibc the reference
 * is built against (2.35 in this image; the same file since glibc 2.0) that is
 * sysdeps/ieee754/flt-32/s_atanf.c: fdlibm's float atan -- argument reduction
 * into five intervals, an 11-term polynomial split into odd and even halves,
 * every operation in fp32, no FMA (objdump of libm.so.6: mulss / addss / subss /
 * divss only, a plain FUNC without IFUNC variants).  Restated here operation by
 * operation; oracle/tools/atanf_exhaustive.c compares it with libm's atanf over
 * ALL 2^32 inputs in the build container (0 differences) and writes the block
 * checksums of tests/golden/atanf_blocks.json, which the device's copy
 * (csrc/device_common.hpp: atanf_glibc) is tested against on the GPU.
 * Constants are glibc's, given by their bit patterns.
 * ------------------------------------------------------------------------ */
static inline float f32_from_bits(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned int f32_bits(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }

float sco_atanf_glibc(float x)
{
    static const unsigned int HI[4] = { 0x3eed6338u, 0x3f490fdau, 0x3f7b985eu, 0x3fc90fdau };   /* atan(0.5), atan(1), atan(1.5), atan(inf): high parts */
    static const unsigned int LO[4] = { 0x31ac3769u, 0x33222168u, 0x33140fb4u, 0x33a22168u };   /* ... low parts */
    static const unsigned int AT[11] = { 0x3eaaaaabu, 0xbe4ccccdu, 0x3e124925u, 0xbde38e38u, 0x3dba2e6eu, 0xbd9d8795u,
                                         0x3d886b35u, 0xbd6ef16bu, 0x3d4bda59u, 0xbd15a221u, 0x3c8569d7u };
    const unsigned int hx = f32_bits(x), ix = hx & 0x7fffffffu;
    int id;
    if (ix >= 0x4c000000u) {                              /* |x| >= 2^25 */
        if (ix > 0x7f800000u) return x + x;               /* NaN */
        const float r = f32_from_bits(HI[3]) + f32_from_bits(LO[3]);
        return (hx >> 31) ? -r : r;
    }
    if (ix < 0x3ee00000u) {                               /* |x| < 0.4375 */
        if (ix < 0x31000000u) return x;                   /* |x| < 2^-29 */
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000u) {                           /* |x| < 1.1875 */
            if (ix < 0x3f300000u) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }      /* 7/16 <= |x| < 11/16 */
            else                  { id = 1; x = (x - 1.0f) / (x + 1.0f); }             /* 11/16 <= |x| < 19/16 */
        } else {
            if (ix < 0x401c0000u) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }      /* |x| < 2.4375 */
            else                  { id = 3; x = -1.0f / x; }                           /* 2.4375 <= |x| < 2^25 */
        }
    }
    const float z = x * x;
    const float w = z * z;
#define A(i) f32_from_bits(AT[i])
    const float s1 = z * (A(0) + w * (A(2) + w * (A(4) + w * (A(6) + w * (A(8) + w * A(10))))));
    const float s2 = w * (A(1) + w * (A(3) + w * (A(5) + w * (A(7) + w * A(9)))));
#undef A
    if (id < 0) return x - x * (s1 + s2);
    const float r = f32_from_bits(HI[id]) - ((x * (s1 + s2) - f32_from_bits(LO[id])) - x);
    return (hx >> 31) ? -r : r;
}

/* Rounds 1-4 used this fp64 argument-reduction + polynomial atan narrowed to float on both sides (the platform's atanf
 * "unspecified in the last bit"); kept for the record of tests/test_oracle_envelope.py: its sector bin differs from the
 * platform's for 2 points in 10^8. */
static const double ATAN_HI[4] = {
    4.63647609000806093515e-01, /* atan(0.5) */
    7.85398163397448278999e-01, /* atan(1.0) */
    9.82793723247329054082e-01, /* atan(1.5) */
    1.57079632679489655800e+00, /* atan(inf) */
};
static const double ATAN_LO[4] = {
    2.26987774529616870924e-17,
    3.06161699786838301793e-17,
    1.39033110312309984516e-17,
    6.12323399573676603587e-17,
};
static const double ATAN_T[11] = {
    3.33333333333329318027e-01, -1.99999999998764832476e-01,
    1.42857142725034663711e-01, -1.11111104054623557880e-01,
    9.09088713343650656196e-02, -7.69187620504482999495e-02,
    6.66107313738753120669e-02, -5.83357013379057348645e-02,
    4.97687799461593236017e-02, -3.65315727442169155270e-02,
    1.62858201153657823623e-02,
};

double sco_atan_pos(double x)
{
    int id;
    double z, w, s1, s2;
    if (x != x) return x;                       /* NaN */
    if (x >= 7.378697629483821e19) /* 2^66 */
        return ATAN_HI[3] + ATAN_LO[3];
    if (x < 0.4375) {
        if (x < 1.862645149230957e-09) /* 2^-29 */ return x;
        id = -1;
    } else if (x < 1.1875) {
        if (x < 0.6875) { id = 0; x = (2.0 * x - 1.0) / (2.0 + x); }
        else            { id = 1; x = (x - 1.0) / (x + 1.0); }
    } else {
        if (x < 2.4375) { id = 2; x = (x - 1.5) / (1.0 + 1.5 * x); }
        else            { id = 3; x = -1.0 / x; }
    }
    z = x * x;
    w = z * z;
    s1 = z * (ATAN_T[0] + w * (ATAN_T[2] + w * (ATAN_T[4] + w * (ATAN_T[6] + w * (ATAN_T[8] + w * ATAN_T[10])))));
    s2 = w * (ATAN_T[1] + w * (ATAN_T[3] + w * (ATAN_T[5] + w * (ATAN_T[7] + w * ATAN_T[9]))));
    if (id < 0) return x - x * (s1 + s2);
    return ATAN_HI[id] - ((x * (s1 + s2) - ATAN_LO[id]) - x);
}

static float sco_atanf(float x) { return sco_atanf_glibc(x); }

/* Checksum of sco_atanf_glibc over block `block` (0..255) of 2^24 consecutive float bit patterns: sum mod 2^64 of
 * splitmix64((bits << 32) | result bits), NaN results counted as 0x7fc00000 -- tests/golden/atanf_blocks.json holds the 256
 * values (written by oracle/tools/atanf_exhaustive.c, where every result is also compared with libm's atanf). */
unsigned long long sco_atanf_block_checksum(int block)
{
    unsigned long long h = 0;
    for (unsigned int i = 0; i < (1u << 24); i++) {
        const unsigned int bits = ((unsigned int)block << 24) | i;
        const float a = sco_atanf_glibc(f32_from_bits(bits));
        unsigned long long z = ((unsigned long long)bits << 32) | (a != a ? 0x7fc00000u : f32_bits(a));
        z += 0x9e3779b97f4a7c15ull;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        h += z ^ (z >> 31);
    }
    return h;
}

/* D.h:1352-1374.  180/M_PI is a double; atan(float) is float; the result is
 * narrowed to float on return.  Non-short-circuit '&' has no side effects. */
float sco_xy2theta(float x, float y)
{
    const double k = 180 / M_PI;
    if ((x >= 0) & (y >= 0)) return (float)(k * (double)sco_atanf(y / x));
    if ((x < 0) & (y >= 0))  return (float)(180 - (k * (double)sco_atanf(y / (-x))));
    if ((x < 0) & (y < 0))   return (float)(180 + (k * (double)sco_atanf(y / x)));
    if ((x >= 0) & (y < 0))  return (float)(360 - (k * (double)sco_atanf((-y) / x)));
    return NAN; /* reference falls off the end (UB) for NaN inputs, D.h:1374 */
}

/* int(ceil(v)) as x86-64 evaluates it: NaN / out-of-range -> INT_MIN (cvttsd2si). */
static int ceil_to_int_x86(double v)
{
    double c = ceil(v);
    if (!(c >= -2147483648.0 && c <= 2147483647.0)) return (-2147483647 - 1);
    return (int)c;
}

static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* D.h:1404-1461 */
void sco_make_scancontext(const sco_config *c, const void *pts, int n, int stride_bytes,
                          double *desc, float *vT)
{
    const int R = c->num_ring, S = c->num_sector;
    const int NO_POINT = -1000;                                   /* D.h:1411 */
    for (int i = 0; i < R * S; i++) desc[i] = (double)NO_POINT;   /* D.h:1412 */

    const unsigned char *base = (const unsigned char *)pts;
    for (int p = 0; p < n; p++) {                                 /* D.h:1418 */
        float xyz[3];
        memcpy(xyz, base + (size_t)p * (size_t)stride_bytes, sizeof xyz);
        float px = xyz[0], py = xyz[1];
        float pz = (float)((double)xyz[2] + c->lidar_height);     /* D.h:1422 */

        float azim_range = sqrtf(px * px + py * py);              /* D.h:1425 */
        float azim_angle = sco_xy2theta(px, py);                  /* D.h:1426 */

        if ((double)azim_range > c->max_radius) continue;         /* D.h:1429 */

        int ring_idx = imax(imin(R, ceil_to_int_x86(((double)azim_range / c->max_radius) * R)), 1);  /* D.h:1434 */
        int sctor_idx = imax(imin(S, ceil_to_int_x86(((double)azim_angle / 360.0) * S)), 1);         /* D.h:1435 */

        double *cell = &desc[(size_t)(sctor_idx - 1) * R + (ring_idx - 1)];
        if (*cell < (double)pz) *cell = (double)pz;               /* D.h:1438-1441 */
    }

    for (int r = 0; r < R; r++) {                                 /* D.h:1446-1456 */
        for (int s = 0; s < S; s++) {
            double *cell = &desc[(size_t)s * R + r];
            if (*cell == (double)NO_POINT) *cell = 0;
            if (vT) vT[(size_t)r * S + s] = (float)*cell;
        }
    }
}

/* D.h:1463-1475: float(mean of row) */
void sco_ringkey(int R, int S, const double *desc, float *key)
{
    for (int r = 0; r < R; r++) {
        double sum = 0;
        for (int s = 0; s < S; s++) sum += desc[(size_t)s * R + r];
        key[r] = (float)(sum / (double)S);
    }
}

/* D.h:1477-1489: mean of column (double) */
void sco_sectorkey(int R, int S, const double *desc, double *vkey)
{
    for (int s = 0; s < S; s++) {
        double sum = 0;
        for (int r = 0; r < R; r++) sum += desc[(size_t)s * R + r];
        vkey[s] = sum / (double)R;
    }
}

/* D.h:1376-1395: out.col((c+shift)%S) = in.col(c) */
void sco_circshift(int R, int S, const double *in, int shift, double *out)
{
    if (shift == 0) { memcpy(out, in, sizeof(double) * (size_t)R * S); return; }
    for (int c = 0; c < S; c++) {
        int nl = (c + shift) % S;
        memcpy(out + (size_t)nl * R, in + (size_t)c * R, sizeof(double) * (size_t)R);
    }
}

/* D.h:1491-1511 */
int sco_fast_align(int S, const double *vkey1, const double *vkey2)
{
    int argmin_vkey_shift = 0;
    double min_diff_norm = 10000000;
    double *shifted = tls_grow(&tls_align, &tls_align_cap, (size_t)S);
    for (int shift = 0; shift < S; shift++) {
        sco_circshift(1, S, vkey2, shift, shifted);               /* D.h:1498 */
        double ss = 0;
        for (int j = 0; j < S; j++) {                             /* D.h:1500-1502 */
            double d = vkey1[j] - shifted[j];
            ss += d * d;
        }
        double cur = sqrt(ss);
        if (cur < min_diff_norm) { argmin_vkey_shift = shift; min_diff_norm = cur; }
    }
    return argmin_vkey_shift;
}

static double col_norm(int R, const double *col)
{
    double ss = 0;
    for (int r = 0; r < R; r++) ss += col[r] * col[r];
    return sqrt(ss);
}

/* D.h:1513-1536 (column copies and the double norm evaluation kept on purpose:
 * this is also the reference-shaped CPU baseline) */
double sco_dist_direct(int R, int S, const double *sc1, const double *sc2)
{
    int num_eff_cols = 0;
    double sum_sector_similarity = 0;
    double *c1 = tls_grow(&tls_cols, &tls_cols_cap, 2 * (size_t)R);
    double *c2 = c1 + R;
    for (int col = 0; col < S; col++) {
        memcpy(c1, sc1 + (size_t)col * R, sizeof(double) * (size_t)R);     /* D.h:1520 */
        memcpy(c2, sc2 + (size_t)col * R, sizeof(double) * (size_t)R);     /* D.h:1521 */
        if ((col_norm(R, c1) == 0) | (col_norm(R, c2) == 0)) continue;     /* D.h:1523 */
        double dot = 0;
        for (int r = 0; r < R; r++) dot += c1[r] * c2[r];
        double sim = dot / (col_norm(R, c1) * col_norm(R, c2));            /* D.h:1528 */
        sum_sector_similarity = sum_sector_similarity + sim;
        num_eff_cols = num_eff_cols + 1;
    }
    double sc_sim = sum_sector_similarity / num_eff_cols;                  /* 0/0 -> NaN */
    return 1.0 - sc_sim;
}

static int search_radius(const sco_config *c)
{   /* D.h:1545: int = round(0.5 * SEARCH_RATIO * cols) */
    return (int)round(0.5 * c->search_ratio * (double)c->num_sector);
}

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* D.h:1538-1569 */
void sco_distance(const sco_config *c, const double *sc1, const double *sc2,
                  double *dist, int *shift)
{
    const int R = c->num_ring, S = c->num_sector;
    double *vk1 = tls_grow(&tls_vkeys, &tls_vkeys_cap, 2 * (size_t)S);
    double *vk2 = vk1 + S;
    sco_sectorkey(R, S, sc1, vk1);                                /* D.h:1541 */
    sco_sectorkey(R, S, sc2, vk2);                                /* D.h:1542 */
    int a = sco_fast_align(S, vk1, vk2);                          /* D.h:1543 */

    const int SR = search_radius(c);
    int nsp = 1 + 2 * (SR > 0 ? SR : 0);
    int *space = (int *)tls_grow(&tls_space, &tls_space_cap, ((size_t)nsp + 1) / 2);
    int m = 0;
    space[m++] = a;
    for (int ii = 1; ii < SR + 1; ii++) {                         /* D.h:1547-1551 */
        space[m++] = (a + ii + S) % S;
        space[m++] = (a - ii + S) % S;
    }
    qsort(space, (size_t)m, sizeof(int), cmp_int);                /* D.h:1552 */

    int argmin_shift = 0;
    double min_sc_dist = 10000000;
    double *shifted = tls_grow(&tls_shifted, &tls_shifted_cap, (size_t)R * S);
    for (int t = 0; t < m; t++) {                                 /* D.h:1557-1566 */
        int num_shift = space[t];
        sco_circshift(R, S, sc2, num_shift, shifted);
        double cur = sco_dist_direct(R, S, sc1, shifted);
        if (cur < min_sc_dist) { argmin_shift = num_shift; min_sc_dist = cur; }
    }
    *dist = min_sc_dist;
    *shift = argmin_shift;
}

/* Same arithmetic, same order, no copies: column norms and sector keys are
 * computed once, the shifted candidate is addressed by index.  Bit-identical
 * to sco_distance (asserted in tests/test_oracle_kat.py). */
void sco_distance_fast(const sco_config *c, const double *sc1, const double *sc2,
                       double *dist, int *shift)
{
    const int R = c->num_ring, S = c->num_sector;
    const int nsp_max = 1 + 2 * (search_radius(c) > 0 ? search_radius(c) : 0);
    double *buf = tls_grow(&tls_fast, &tls_fast_cap, 4 * (size_t)S + ((size_t)nsp_max + 1) / 2);
    double *vk1 = buf, *vk2 = buf + S, *n1 = buf + 2 * S, *n2 = buf + 3 * S;
    sco_sectorkey(R, S, sc1, vk1);
    sco_sectorkey(R, S, sc2, vk2);
    for (int s = 0; s < S; s++) {
        n1[s] = col_norm(R, sc1 + (size_t)s * R);
        n2[s] = col_norm(R, sc2 + (size_t)s * R);
    }
    int a = 0;
    double best = 10000000;
    for (int sh = 0; sh < S; sh++) {
        double ss = 0;
        for (int j = 0; j < S; j++) {
            int src = j - sh; if (src < 0) src += S;
            double d = vk1[j] - vk2[src];
            ss += d * d;
        }
        double cur = sqrt(ss);
        if (cur < best) { a = sh; best = cur; }
    }
    const int SR = search_radius(c);
    int nsp = 1 + 2 * (SR > 0 ? SR : 0);
    int *space = (int *)(buf + 4 * (size_t)S);
    int m = 0;
    space[m++] = a;
    for (int ii = 1; ii < SR + 1; ii++) {
        space[m++] = (a + ii + S) % S;
        space[m++] = (a - ii + S) % S;
    }
    qsort(space, (size_t)m, sizeof(int), cmp_int);
    int argmin_shift = 0;
    double min_sc_dist = 10000000;
    for (int t = 0; t < m; t++) {
        int sh = space[t];
        int eff = 0;
        double sum = 0;
        for (int col = 0; col < S; col++) {
            int src = (col - sh) % S; if (src < 0) src += S;
            if ((n1[col] == 0) | (n2[src] == 0)) continue;
            const double *a1 = sc1 + (size_t)col * R, *a2 = sc2 + (size_t)src * R;
            double dot = 0;
            for (int r = 0; r < R; r++) dot += a1[r] * a2[r];
            sum = sum + dot / (n1[col] * n2[src]);
            eff++;
        }
        double cur = 1.0 - sum / eff;
        if (cur < min_sc_dist) { argmin_shift = sh; min_sc_dist = cur; }
    }
    (void)nsp;
    *dist = min_sc_dist;
    *shift = argmin_shift;
}

/* NF:383-408 with the 3-argument call of NF:1358 (worst_dist = -1: no early exit) */
static float nf_l2(const float *a, const float *b, int size)
{
    float result = 0.0f;
    const float *last = a + size;
    const float *lastgroup = last - 3;
    while (a < lastgroup) {
        const float d0 = a[0] - b[0];
        const float d1 = a[1] - b[1];
        const float d2 = a[2] - b[2];
        const float d3 = a[3] - b[3];
        result += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        a += 4; b += 4;
    }
    while (a < last) {
        const float d0 = *a++ - *b++;
        result += d0 * d0;
    }
    return result;
}

/* KNNResultSet::addPoint (NF:177-199) + acceptance test `dist < worstDist` (NF:1360) */
int sco_knn(const float *keys, int N, int R, const float *query, int k,
            float exclude_eps, int *idx, float *d2)
{
    int count = 0;
    for (int i = 0; i < k; i++) { idx[i] = -1; d2[i] = FLT_MAX; }
    if (k <= 0) return 0;
    for (int n = 0; n < N; n++) {
        float dist = nf_l2(query, keys + (size_t)n * R, R);
        if (exclude_eps > 0.0f && dist <= exclude_eps) continue;   /* libnabo self-match rule */
        if (!(dist < d2[k - 1])) continue;
        int i;
        for (i = count; i > 0; --i) {
            if (d2[i - 1] > dist) {
                if (i < k) { d2[i] = d2[i - 1]; idx[i] = idx[i - 1]; }
            } else break;
        }
        if (i < k) { d2[i] = dist; idx[i] = n; }
        if (count < k) count++;
    }
    return count;
}

/* --------------------------------------------------------------------------
 * database (D.h:1768-1800)
 * ----------------------------------------------------------------------- */
struct sco_db {
    sco_config cfg;
    int n, cap;
    double *descs;      /* n * R*S, column-major each   (polarcontexts_)        */
    float  *keys;       /* n * R                        (polarcontextRowKey)    */
    int8_t *robots;     /* polarcontext_indexs_.first  */
    int    *indexs;     /* polarcontext_indexs_.second */
    int tree_counter;   /* tree_making_period_conter, initialised to 0 (repair) */
    int tree_n;         /* number of keys covered by the current inter tree     */
};

sco_db *sco_db_create(const sco_config *c)
{
    sco_db *db = (sco_db *)calloc(1, sizeof *db);
    db->cfg = *c;
    return db;
}

void sco_db_destroy(sco_db *db)
{
    if (!db) return;
    free(db->descs); free(db->keys); free(db->robots); free(db->indexs); free(db);
}

static void db_grow(sco_db *db)
{
    if (db->n < db->cap) return;
    int cap = db->cap ? db->cap * 2 : 256;
    size_t cells = (size_t)db->cfg.num_ring * db->cfg.num_sector;
    db->descs = (double *)realloc(db->descs, sizeof(double) * cells * (size_t)cap);
    db->keys = (float *)realloc(db->keys, sizeof(float) * (size_t)db->cfg.num_ring * (size_t)cap);
    db->robots = (int8_t *)realloc(db->robots, (size_t)cap);
    db->indexs = (int *)realloc(db->indexs, sizeof(int) * (size_t)cap);
    db->cap = cap;
}

/* D.h:1587-1602 */
static void db_save(sco_db *db, const double *sc, int8_t robot, int index)
{
    const int R = db->cfg.num_ring, S = db->cfg.num_sector;
    db_grow(db);
    memcpy(db->descs + (size_t)db->n * R * S, sc, sizeof(double) * (size_t)R * S);
    sco_ringkey(R, S, sc, db->keys + (size_t)db->n * R);
    db->robots[db->n] = robot;
    db->indexs[db->n] = index;
    db->n++;
}

/* D.h:1572-1585 */
void sco_db_save_wire(sco_db *db, const float *values, int8_t robot, int index)
{
    const int R = db->cfg.num_ring, S = db->cfg.num_sector;
    double *sc = (double *)malloc(sizeof(double) * (size_t)R * S);
    for (int r = 0; r < R; r++)
        for (int s = 0; s < S; s++)
            sc[(size_t)s * R + r] = (double)values[(size_t)r * S + s];       /* D.h:1580 */
    db_save(db, sc, robot, index);
    free(sc);
}

/* D.h:1604-1611 */
void sco_db_make_and_save(sco_db *db, const void *pts, int n, int stride_bytes,
                          int8_t robot, int index, float *vT)
{
    const int R = db->cfg.num_ring, S = db->cfg.num_sector;
    double *sc = (double *)malloc(sizeof(double) * (size_t)R * S);
    sco_make_scancontext(&db->cfg, pts, n, stride_bytes, sc, vT);
    db_save(db, sc, robot, index);
    free(sc);
}

int sco_db_size(const sco_db *db) { return db->n; }

void sco_db_get_index(const sco_db *db, int key, int8_t *robot, int *index)
{
    *robot = db->robots[key];
    *index = db->indexs[key];
}

const double *sco_db_desc(const sco_db *db, int key)
{
    return db->descs + (size_t)key * db->cfg.num_ring * db->cfg.num_sector;
}

const float *sco_db_ringkey(const sco_db *db, int key)
{
    return db->keys + (size_t)key * db->cfg.num_ring;
}

/* D.h:1613-1674 */
void sco_db_detect_intra(sco_db *db, int cur, int *loop_id, float *shift,
                         double *dist, double *dist_exact)
{
    const sco_config *c = &db->cfg;
    const int k = c->num_candidates;
    *loop_id = -1; *shift = 0.0f;
    if (dist) *dist = 10000000.0;
    if (dist_exact) *dist_exact = 10000000.0;
    if (cur < 0 || cur >= db->n) return;
    if (cur < c->num_exclude_recent + k + 1) return;              /* D.h:1620-1623 */

    int history = cur - c->num_exclude_recent;                     /* D.h:1627 */
    int *idx = (int *)malloc(sizeof(int) * (size_t)k);
    float *d2 = (float *)malloc(sizeof(float) * (size_t)k);
    sco_knn(db->keys, history, c->num_ring, sco_db_ringkey(db, cur), k,
            c->knn_exclude_eps, idx, d2);                           /* D.h:1631,1642 */

    float minDis = 10000000.0f;                                    /* D.h:1637: float! */
    int minIndex = -1, minBias = 0;
    double exact = 10000000.0;
    for (int i = 0; i < k; i++) {                                  /* D.h:1645-1659 */
        if (idx[i] < 0) continue;
        double cd; int ca;
        sco_distance(c, sco_db_desc(db, cur), sco_db_desc(db, idx[i]), &cd, &ca);
        if (cd < (double)minDis) {                                 /* D.h:1653 */
            minDis = (float)cd;                                    /* D.h:1655: narrowing */
            minIndex = idx[i];
            minBias = ca;
            exact = cd;
        }
    }
    free(idx); free(d2);
    if (dist) *dist = (double)minDis;
    if (dist_exact) *dist_exact = exact;
    if ((double)minDis < c->dist_thres) {                          /* D.h:1662 */
        *loop_id = minIndex;
        *shift = (float)minBias;                                   /* D.h:1665 */
    }
}

/* D.h:1676-1756, repaired: keys from the live table (D.h:1596 is commented out
 * in the reference, leaving polarcontext_invkeys_mat_ empty), the period counter
 * and PC_UNIT_SECTORANGLE initialised (shadowed in the ctor, D.h:1332-1334). */
void sco_db_detect_inter(sco_db *db, int cur, int *loop_id, float *yaw_rad, double *dist)
{
    const sco_config *c = &db->cfg;
    const int k = c->num_candidates;
    *loop_id = -1; *yaw_rad = 0.0f;
    if (dist) *dist = 10000000.0;
    if (cur < 0 || cur >= db->n) return;
    if (db->n < c->num_exclude_recent + 1) return;                 /* D.h:1684-1688 */

    if (db->tree_counter % c->tree_making_period == 0)             /* D.h:1691-1702 */
        db->tree_n = db->n - c->num_exclude_recent;
    db->tree_counter = db->tree_counter + 1;                       /* D.h:1703 */

    int *idx = (int *)malloc(sizeof(int) * (size_t)k);
    float *d2 = (float *)malloc(sizeof(float) * (size_t)k);
    sco_knn(db->keys, db->tree_n, c->num_ring, sco_db_ringkey(db, cur), k, 0.0f, idx, d2); /* D.h:1710-1716 */

    double min_dist = 10000000;
    int nn_align = 0, nn_idx = -1;
    for (int i = 0; i < k; i++) {                                  /* D.h:1721-1737 */
        int ci = idx[i] < 0 ? 0 : idx[i];   /* candidate_indexes is zero-initialised, D.h:1710 */
        double cd; int ca;
        sco_distance(c, sco_db_desc(db, cur), sco_db_desc(db, ci), &cd, &ca);
        if (cd < min_dist) {
            if (ci == cur) continue;                               /* D.h:1731 */
            min_dist = cd; nn_align = ca; nn_idx = ci;
        }
    }
    free(idx); free(d2);
    if (min_dist < c->dist_thres) *loop_id = nn_idx;               /* D.h:1741-1744 */
    const double unit_sector_angle = 360.0 / (double)c->num_sector;   /* D.h:1332 */
    *yaw_rad = (float)(nn_align * unit_sector_angle * M_PI / 180.0);  /* D.h:1752 */
    if (dist) *dist = min_dist;
}

void sco_db_detect_full(sco_db *db, int cur, int *loop_id, int *nn_idx, int *shift, double *dist)
{
    const sco_config *c = &db->cfg;
    *loop_id = -1; *nn_idx = -1; *shift = 0; *dist = 10000000.0;
    if (cur < 0 || cur >= db->n) return;
    int history = cur - c->num_exclude_recent;
    double best = 10000000; int bi = -1, bs = 0;
    for (int i = 0; i < history; i++) {
        double cd; int ca;
        sco_distance_fast(c, sco_db_desc(db, cur), sco_db_desc(db, i), &cd, &ca);
        if (cd < best) { best = cd; bi = i; bs = ca; }
    }
    *dist = best; *nn_idx = bi; *shift = bs;
    if (best < c->dist_thres) *loop_id = bi;
}

void sco_db_distance_batch(sco_db *db, int cur, const int *cand, int n,
                           double *dist, int *shift, int fast)
{
    for (int i = 0; i < n; i++) {
        int ci = cand ? cand[i] : i;
        if (fast) sco_distance_fast(&db->cfg, sco_db_desc(db, cur), sco_db_desc(db, ci), &dist[i], &shift[i]);
        else      sco_distance(&db->cfg, sco_db_desc(db, cur), sco_db_desc(db, ci), &dist[i], &shift[i]);
    }
}

/* ---- CPU baseline variant B: the same reference-shaped evaluation on every host core ----
 * A pool of worker threads created ONCE (and re-made only when the requested count changes); a call publishes one job,
 * the workers take candidates in blocks of 16 from a shared counter (first come first served: cores of a busy host are
 * not equally fast) and the caller sleeps until the last block is done.  No thread is created and nothing is allocated
 * per call, per pair, per shift or per column (the reference-shaped COPIES stay: sco_distance). */
#define SCO_POOL_MAX 1024
#define SCO_POOL_BLOCK 16
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv_work, cv_done;
    pthread_t th[SCO_POOL_MAX];
    int n_threads;
    unsigned long long generation;       /* bumped per job */
    int stop;
    /* the job */
    sco_db *db; int cur; const int *cand; int n; double *dist; int *shift; int fast;
    int next;                            /* next unclaimed candidate (atomic) */
    int busy;                            /* workers still inside the job */
} sco_pool;
static sco_pool g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, {0}, 0, 0, 0,
                           NULL, 0, NULL, 0, NULL, NULL, 0, 0, 0 };
static pthread_mutex_t g_pool_call = PTHREAD_MUTEX_INITIALIZER;   /* one job at a time */

static void pool_run_blocks(sco_pool *p)
{
    for (;;) {
        const int lo = __atomic_fetch_add(&p->next, SCO_POOL_BLOCK, __ATOMIC_RELAXED);
        if (lo >= p->n) break;
        const int hi = lo + SCO_POOL_BLOCK < p->n ? lo + SCO_POOL_BLOCK : p->n;
        for (int i = lo; i < hi; i++) {
            const int ci = p->cand ? p->cand[i] : i;
            if (p->fast) sco_distance_fast(&p->db->cfg, sco_db_desc(p->db, p->cur), sco_db_desc(p->db, ci), &p->dist[i], &p->shift[i]);
            else         sco_distance(&p->db->cfg, sco_db_desc(p->db, p->cur), sco_db_desc(p->db, ci), &p->dist[i], &p->shift[i]);
        }
    }
}

static void *pool_worker(void *arg)
{
    sco_pool *p = (sco_pool *)arg;
    unsigned long long seen = 0;
    pthread_mutex_lock(&p->mu);
    for (;;) {
        while (!p->stop && p->generation == seen) pthread_cond_wait(&p->cv_work, &p->mu);
        if (p->stop) break;
        seen = p->generation;
        pthread_mutex_unlock(&p->mu);
        pool_run_blocks(p);
        pthread_mutex_lock(&p->mu);
        if (--p->busy == 0) pthread_cond_signal(&p->cv_done);
    }
    pthread_mutex_unlock(&p->mu);
    return NULL;
}

static void pool_stop_locked_call(sco_pool *p)
{
    if (p->n_threads == 0) return;
    pthread_mutex_lock(&p->mu);
    p->stop = 1;
    pthread_cond_broadcast(&p->cv_work);
    pthread_mutex_unlock(&p->mu);
    for (int t = 0; t < p->n_threads; t++) pthread_join(p->th[t], NULL);
    p->n_threads = 0; p->stop = 0;
}

/* threads workers besides the caller's own share; returns the number actually running */
static int pool_ensure(sco_pool *p, int threads)
{
    if (p->n_threads == threads) return threads;
    pool_stop_locked_call(p);
    pthread_attr_t at;
    pthread_attr_init(&at);
    pthread_attr_setstacksize(&at, 256 * 1024);
    int made = 0;
    for (int t = 0; t < threads; t++) {
        if (pthread_create(&p->th[t], &at, pool_worker, p) != 0) break;
        made++;
    }
    pthread_attr_destroy(&at);
    p->n_threads = made;
    return made;
}

void sco_pool_shutdown(void)
{
    pthread_mutex_lock(&g_pool_call);
    pool_stop_locked_call(&g_pool);
    pthread_mutex_unlock(&g_pool_call);
}

void sco_db_distance_batch_mt(sco_db *db, int cur, const int *cand, int n,
                              double *dist, int *shift, int fast, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > SCO_POOL_MAX) threads = SCO_POOL_MAX;
    sco_pool *p = &g_pool;
    pthread_mutex_lock(&g_pool_call);
    const int workers = pool_ensure(p, threads - 1);     /* the caller is the last thread of the team */
    pthread_mutex_lock(&p->mu);
    p->db = db; p->cur = cur; p->cand = cand; p->n = n; p->dist = dist; p->shift = shift; p->fast = fast;
    __atomic_store_n(&p->next, 0, __ATOMIC_RELAXED);
    p->busy = workers;
    p->generation++;
    pthread_cond_broadcast(&p->cv_work);
    pthread_mutex_unlock(&p->mu);
    pool_run_blocks(p);
    pthread_mutex_lock(&p->mu);
    while (p->busy > 0) pthread_cond_wait(&p->cv_done, &p->mu);
    pthread_mutex_unlock(&p->mu);
    pthread_mutex_unlock(&g_pool_call);
}

/* ---- Envelope of the parts that cannot be pinned offline (tests only) ------------------------------------------
 * Eigen's .mean() / .norm() / .dot() (D.h:1470-1471, 1484-1485, 1500-1502, 1523, 1528) are packet-vectorised:
 * `lanes` interleaved partial sums (2 with SSE2, 4 with AVX, 8 with AVX-512 or Eigen's 4 x unrolled SSE2 packets),
 * combined pairwise at the end, then a scalar tail.  sco_distance_lanes is sco_distance_fast with every such
 * reduction evaluated in that shape; everything the reference writes as a scalar loop (the sum over sectors in
 * distDirectSC, D.h:1518-1532, and the arg-min scans) stays sequential.  lanes = 1 reproduces sco_distance_fast. */
static double lanes_sum(const double *v, int n, int stride, int lanes)
{
    double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int body = (n / lanes) * lanes;
    for (int i = 0; i < body; i++) p[i % lanes] += v[(size_t)i * stride];
    for (int w = lanes; w > 1; w >>= 1)
        for (int k = 0; k < w / 2; k++) p[k] = p[k] + p[k + w / 2];
    double s = p[0];
    for (int i = body; i < n; i++) s += v[(size_t)i * stride];
    return s;
}

static double lanes_dot(const double *a, const double *b, int n, int lanes)
{
    double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int body = (n / lanes) * lanes;
    for (int i = 0; i < body; i++) p[i % lanes] += a[i] * b[i];
    for (int w = lanes; w > 1; w >>= 1)
        for (int k = 0; k < w / 2; k++) p[k] = p[k] + p[k + w / 2];
    double s = p[0];
    for (int i = body; i < n; i++) s += a[i] * b[i];
    return s;
}

void sco_ringkey_lanes(int R, int S, const double *desc, int lanes, float *key)
{
    for (int r = 0; r < R; r++) key[r] = (float)(lanes_sum(desc + r, S, R, lanes) / (double)S);
}

void sco_distance_lanes(const sco_config *c, const double *sc1, const double *sc2, int lanes,
                        double *dist, int *shift)
{
    const int R = c->num_ring, S = c->num_sector;
    if (lanes < 1) lanes = 1;
    if (lanes > 8) lanes = 8;
    double *buf = (double *)malloc(sizeof(double) * 5 * (size_t)S);
    double *vk1 = buf, *vk2 = buf + S, *n1 = buf + 2 * S, *n2 = buf + 3 * S, *diff = buf + 4 * S;
    for (int s = 0; s < S; s++) {
        vk1[s] = lanes_sum(sc1 + (size_t)s * R, R, 1, lanes) / (double)R;
        vk2[s] = lanes_sum(sc2 + (size_t)s * R, R, 1, lanes) / (double)R;
        n1[s] = sqrt(lanes_dot(sc1 + (size_t)s * R, sc1 + (size_t)s * R, R, lanes));
        n2[s] = sqrt(lanes_dot(sc2 + (size_t)s * R, sc2 + (size_t)s * R, R, lanes));
    }
    int a = 0;
    double best = 10000000;
    for (int sh = 0; sh < S; sh++) {
        for (int j = 0; j < S; j++) { int src = j - sh; if (src < 0) src += S; diff[j] = vk1[j] - vk2[src]; }
        double cur = sqrt(lanes_dot(diff, diff, S, lanes));
        if (cur < best) { a = sh; best = cur; }
    }
    const int SR = search_radius(c);
    int nsp = 1 + 2 * (SR > 0 ? SR : 0);
    int *space = (int *)malloc(sizeof(int) * (size_t)nsp);
    int m = 0;
    space[m++] = a;
    for (int ii = 1; ii < SR + 1; ii++) { space[m++] = (a + ii + S) % S; space[m++] = (a - ii + S) % S; }
    qsort(space, (size_t)m, sizeof(int), cmp_int);
    int argmin_shift = 0;
    double min_sc_dist = 10000000;
    for (int t = 0; t < m; t++) {
        int sh = space[t], eff = 0;
        double sum = 0;
        for (int col = 0; col < S; col++) {
            int src = (col - sh) % S; if (src < 0) src += S;
            if ((n1[col] == 0) | (n2[src] == 0)) continue;
            sum = sum + lanes_dot(sc1 + (size_t)col * R, sc2 + (size_t)src * R, R, lanes) / (n1[col] * n2[src]);
            eff++;
        }
        double cur = 1.0 - sum / eff;
        if (cur < min_sc_dist) { argmin_shift = sh; min_sc_dist = cur; }
    }
    free(space);
    free(buf);
    *dist = min_sc_dist;
    *shift = argmin_shift;
}

/* Census of sector-bin differences between the fixed atan used on both sides of the parity tests (sco_atanf) and
 * this platform's libm atanf, which is what the reference calls (D.h:1357-1372).  n points, uniform in the square
 * [-range, range]^2, xorshift64* stream of `seed`.  Returns the number of points whose sector index
 * (D.h:1435) differs; *theta_diff (optional) = how many xy2theta results differ in any bit. */
static float xy2theta_libm(float x, float y)
{
    const double k = 180 / M_PI;
    if ((x >= 0) & (y >= 0)) return (float)(k * (double)atanf(y / x));
    if ((x < 0) & (y >= 0))  return (float)(180 - (k * (double)atanf(y / (-x))));
    if ((x < 0) & (y < 0))   return (float)(180 + (k * (double)atanf(y / x)));
    if ((x >= 0) & (y < 0))  return (float)(360 - (k * (double)atanf((-y) / x)));
    return NAN;
}

long long sco_theta_census(long long n, unsigned long long seed, double range, int S, long long *theta_diff)
{
    unsigned long long s = seed ? seed : 0x9E3779B97F4A7C15ull;
    long long flips = 0, tdiff = 0;
    for (long long i = 0; i < n; i++) {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        const unsigned long long r = s * 0x2545F4914F6CDD1Dull;
        const float x = (float)(((double)(r >> 40) * (1.0 / 16777216.0) * 2.0 - 1.0) * range);
        const float y = (float)(((double)((r >> 16) & 0xFFFFFF) * (1.0 / 16777216.0) * 2.0 - 1.0) * range);
        const float a = sco_xy2theta(x, y), b = xy2theta_libm(x, y);
        if (memcmp(&a, &b, sizeof a) != 0) {
            tdiff++;
            const int ia = imax(imin(S, ceil_to_int_x86(((double)a / 360.0) * S)), 1);
            const int ib = imax(imin(S, ceil_to_int_x86(((double)b / 360.0) * S)), 1);
            if (ia != ib) flips++;
        }
    }
    if (theta_diff) *theta_diff = tdiff;
    return flips;
}
