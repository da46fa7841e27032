/*
 * icp_oracle.c -- CPU restatement of the geometric-verification path (PCL's
 * published algorithms; see icp_oracle.h: PARITY UNPINNED, test infrastructure only).
 *
 * Numerics: points are fp32; nearest-neighbour distances are fp32
 * ((dx*dx + dy*dy) + dz*dz, no FMA); centroids / cross-covariance / MSE are
 * accumulated sequentially in fp64; the rotation is the closed-form optimum of
 * the orthogonal Procrustes problem with det = +1 (what PCL's
 * TransformationEstimationSVD returns through Eigen::umeyama with its
 * sign fix), computed here by Horn's quaternion form with a cyclic Jacobi
 * eigen-solver -- no SVD library needed, no degenerate-rank special cases.
 */
#include "icp_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

void icpo_default_params(icpo_params *p)
{   /* DM.h:1109-1112 */
    p->max_iterations = 50;
    p->max_correspondence_dist = 100.0;
    p->transformation_epsilon = 1e-6;
    p->euclidean_fitness_epsilon = 1e-6;
    p->estimator = 0;
    p->normal_radius = 1.0;
}

static inline const float *pt(const void *base, int i, int stride)
{
    return (const float *)((const unsigned char *)base + (size_t)i * (size_t)stride);
}

static inline float dist2f(const float *a, const float *b)
{
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

/* ---- exact NN: brute force --------------------------------------------------- */
static void nn_brute(const void *src, int n_src, const void *tgt, int n_tgt, int stride,
                     int *nn_index, float *nn_dist2)
{
    for (int i = 0; i < n_src; i++) {
        const float *p = pt(src, i, stride);
        float best = FLT_MAX; int bi = -1;
        for (int j = 0; j < n_tgt; j++) {
            const float d = dist2f(p, pt(tgt, j, stride));
            if (d < best) { best = d; bi = j; }      /* ascending j: ties keep the lowest index */
        }
        nn_index[i] = bi;
        if (nn_dist2) nn_dist2[i] = best;
    }
}

/* ---- exact NN: uniform grid with shell expansion ------------------------------ */
typedef struct { float mn[3]; float h; int dim[3]; int *start; int *items; } grid_t;

static int cell_of(const grid_t *g, const float *p, int c[3])
{
    for (int a = 0; a < 3; a++) {
        float f = floorf((p[a] - g->mn[a]) / g->h);
        int ci = (f != f) ? 0 : (f < 0 ? 0 : (f >= (float)g->dim[a] ? g->dim[a] - 1 : (int)f));
        c[a] = ci;
    }
    return (c[2] * g->dim[1] + c[1]) * g->dim[0] + c[0];
}

static void grid_build(grid_t *g, const void *tgt, int n, int stride)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; i++) {
        const float *p = pt(tgt, i, stride);
        for (int a = 0; a < 3; a++) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
    }
    float ext = 0; for (int a = 0; a < 3; a++) if (mx[a] - mn[a] > ext) ext = mx[a] - mn[a];
    float h = ext / 96.0f; if (!(h > 1e-6f)) h = 1.0f;
    g->h = h;
    size_t cells = 1;
    for (int a = 0; a < 3; a++) {
        g->mn[a] = mn[a];
        int d = (int)floorf((mx[a] - mn[a]) / h) + 1; if (d < 1) d = 1;
        g->dim[a] = d; cells *= (size_t)d;
    }
    g->start = (int *)calloc(cells + 1, sizeof(int));
    g->items = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    int c[3];
    for (int i = 0; i < n; i++) g->start[cell_of(g, pt(tgt, i, stride), c) + 1]++;
    for (size_t k = 0; k < cells; k++) g->start[k + 1] += g->start[k];
    int *fill = (int *)malloc(sizeof(int) * cells);
    memcpy(fill, g->start, sizeof(int) * cells);
    for (int i = 0; i < n; i++) g->items[fill[cell_of(g, pt(tgt, i, stride), c)]++] = i;   /* ascending i per cell */
    free(fill);
}

static void nn_grid(const void *src, int n_src, const void *tgt, int n_tgt, int stride,
                    int *nn_index, float *nn_dist2)
{
    grid_t g; grid_build(&g, tgt, n_tgt, stride);
    const int maxdim = g.dim[0] > g.dim[1] ? (g.dim[0] > g.dim[2] ? g.dim[0] : g.dim[2])
                                           : (g.dim[1] > g.dim[2] ? g.dim[1] : g.dim[2]);
    for (int i = 0; i < n_src; i++) {
        const float *p = pt(src, i, stride);
        int c[3]; cell_of(&g, p, c);
        float best = FLT_MAX; int bi = -1;
        for (int r = 0; r <= maxdim; r++) {
            int lo[3], hi[3];
            for (int a = 0; a < 3; a++) { lo[a] = c[a] - r; hi[a] = c[a] + r; }
            for (int z = (lo[2] < 0 ? 0 : lo[2]); z <= (hi[2] >= g.dim[2] ? g.dim[2] - 1 : hi[2]); z++)
            for (int y = (lo[1] < 0 ? 0 : lo[1]); y <= (hi[1] >= g.dim[1] ? g.dim[1] - 1 : hi[1]); y++)
            for (int x = (lo[0] < 0 ? 0 : lo[0]); x <= (hi[0] >= g.dim[0] ? g.dim[0] - 1 : hi[0]); x++) {
                const int on_shell = (z == lo[2] || z == hi[2] || y == lo[1] || y == hi[1] || x == lo[0] || x == hi[0]);
                if (!on_shell) continue;
                const int cell = (z * g.dim[1] + y) * g.dim[0] + x;
                for (int k = g.start[cell]; k < g.start[cell + 1]; k++) {
                    const int j = g.items[k];
                    const float d = dist2f(p, pt(tgt, j, stride));
                    if (d < best || (d == best && j < bi)) { best = d; bi = j; }
                }
            }
            /* every unsearched point lies beyond a face of the searched block that still has cells behind it */
            float bound = FLT_MAX; int open = 0;
            for (int a = 0; a < 3; a++) {
                if (lo[a] > 0) { float f = p[a] - (g.mn[a] + (float)lo[a] * g.h); if (f < 0) f = 0; if (f < bound) bound = f; open = 1; }
                if (hi[a] < g.dim[a] - 1) { float f = (g.mn[a] + (float)(hi[a] + 1) * g.h) - p[a]; if (f < 0) f = 0; if (f < bound) bound = f; open = 1; }
            }
            if (!open) break;
            bound *= 0.9999f;                         /* never over-estimate (fp32 face positions) */
            if (bi >= 0 && best < bound * bound) break;
        }
        nn_index[i] = bi;
        if (nn_dist2) nn_dist2[i] = best;
    }
    free(g.start); free(g.items);
}

void icpo_nn(const void *src, int n_src, const void *tgt, int n_tgt, int stride_bytes,
             int use_grid, int *nn_index, float *nn_dist2)
{
    if (use_grid && n_tgt > 0) nn_grid(src, n_src, tgt, n_tgt, stride_bytes, nn_index, nn_dist2);
    else nn_brute(src, n_src, tgt, n_tgt, stride_bytes, nn_index, nn_dist2);
}

/* ---- rotation from the cross-covariance ----------------------------------------
 * S[a][b] = sum (p_a - pbar_a)(q_b - qbar_b)   (src x dst).  Horn 1987: the optimal unit
 * quaternion is the eigenvector of the symmetric 4x4 matrix N(S) with the largest
 * eigenvalue.  Cyclic Jacobi, fixed sweep budget, fp64. */
static void jacobi_eig4(double A[4][4], double V[4][4])
{
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 32; sweep++) {
        double off = 0;
        for (int i = 0; i < 4; i++) for (int j = i + 1; j < 4; j++) off += A[i][j] * A[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; p++) for (int q = p + 1; q < 4; q++) {
            if (A[p][q] == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 4; k++) {            /* A <- A J */
                const double akp = A[k][p], akq = A[k][q];
                A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
            }
            for (int k = 0; k < 4; k++) {            /* A <- J^T A */
                const double apk = A[p][k], aqk = A[q][k];
                A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
            }
            for (int k = 0; k < 4; k++) {
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
            }
        }
    }
}

static void rotation_from_S(const double S[3][3], double R[3][3])
{
    double N[4][4], V[4][4];
    const double Sxx = S[0][0], Sxy = S[0][1], Sxz = S[0][2];
    const double Syx = S[1][0], Syy = S[1][1], Syz = S[1][2];
    const double Szx = S[2][0], Szy = S[2][1], Szz = S[2][2];
    N[0][0] = Sxx + Syy + Szz; N[0][1] = Syz - Szy;       N[0][2] = Szx - Sxz;        N[0][3] = Sxy - Syx;
    N[1][1] = Sxx - Syy - Szz; N[1][2] = Sxy + Syx;       N[1][3] = Szx + Sxz;
    N[2][2] = -Sxx + Syy - Szz; N[2][3] = Syz + Szy;
    N[3][3] = -Sxx - Syy + Szz;
    for (int i = 0; i < 4; i++) for (int j = 0; j < i; j++) N[i][j] = N[j][i];
    jacobi_eig4(N, V);
    int m = 0;
    for (int i = 1; i < 4; i++) if (N[i][i] > N[m][m]) m = i;
    double w = V[0][m], x = V[1][m], y = V[2][m], z = V[3][m];
    const double n = sqrt(w * w + x * x + y * y + z * z);
    if (n > 0) { w /= n; x /= n; y /= n; z /= n; } else { w = 1; x = y = z = 0; }
    R[0][0] = w * w + x * x - y * y - z * z; R[0][1] = 2 * (x * y - w * z);           R[0][2] = 2 * (x * z + w * y);
    R[1][0] = 2 * (x * y + w * z);           R[1][1] = w * w - x * x + y * y - z * z; R[1][2] = 2 * (y * z - w * x);
    R[2][0] = 2 * (x * z - w * y);           R[2][1] = 2 * (y * z + w * x);           R[2][2] = w * w - x * x - y * y + z * z;
}

void icpo_rotation_from_covariance(const double H[9], double R[9])
{   /* H = dst x src (Umeyama's Sigma * N): S = H^T */
    double S[3][3], Rm[3][3];
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) S[a][b] = H[b * 3 + a];
    rotation_from_S(S, Rm);
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) R[a * 3 + b] = Rm[a][b];
}

/* centroids + cross-covariance (fp64, sequential) -> 4x4 float transform src->dst */
static int estimate_rigid(const void *src, const void *tgt, int stride,
                          const int *si, const int *ti, int n, float T[16])
{
    if (n < 3) return -1;
    double pm[3] = {0, 0, 0}, qm[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) {
        const float *p = pt(src, si ? si[i] : i, stride), *q = pt(tgt, ti[i], stride);
        for (int a = 0; a < 3; a++) { pm[a] += (double)p[a]; qm[a] += (double)q[a]; }
    }
    for (int a = 0; a < 3; a++) { pm[a] /= (double)n; qm[a] /= (double)n; }
    double S[3][3] = {{0}};
    for (int i = 0; i < n; i++) {
        const float *p = pt(src, si ? si[i] : i, stride), *q = pt(tgt, ti[i], stride);
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++)
            S[a][b] += ((double)p[a] - pm[a]) * ((double)q[b] - qm[b]);
    }
    double R[3][3];
    rotation_from_S(S, R);
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) T[a * 4 + b] = (float)R[a][b];
        T[a * 4 + 3] = (float)(qm[a] - (R[a][0] * pm[0] + R[a][1] * pm[1] + R[a][2] * pm[2]));
    }
    T[12] = T[13] = T[14] = 0.0f; T[15] = 1.0f;
    return 0;
}

int icpo_rigid_svd(const void *src, const void *tgt, int stride_bytes,
                   const int *src_index, const int *tgt_index, int n_corr, float T[16])
{
    return estimate_rigid(src, tgt, stride_bytes, src_index, tgt_index, n_corr, T);
}

/* DM.h:247-250 */
void icpo_transform(const void *in, int n, int stride_bytes, const float T[16], void *out)
{
    if (out != in) memcpy(out, in, (size_t)n * (size_t)stride_bytes);
    for (int i = 0; i < n; i++) {
        const float *p = pt(in, i, stride_bytes);
        float *o = (float *)((unsigned char *)out + (size_t)i * (size_t)stride_bytes);
        const float x = p[0], y = p[1], z = p[2];
        o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
        o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
        o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
    }
}

static void mat4_mul(const float A[16], const float B[16], float C[16])
{
    float r[16];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
        float s = 0.0f;
        for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
        r[i * 4 + j] = s;
    }
    memcpy(C, r, sizeof r);
}

/* ---- point-to-plane pieces -------------------------------------------------------------------------- */
static void jacobi_eig3(double A[3][3], double V[3][3])
{
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 32; sweep++) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double dia = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off < 1e-300 || off <= 1e-34 * dia) break;      /* (below 1e-17 of the diagonal a rotation changes no bit: icp.hip jacobi_eig3) */
        for (int p = 0; p < 2; p++) for (int q = p + 1; q < 3; q++) {
            if (A[p][q] == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 3; k++) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s * b; A[k][q] = s * a + c * b; }
            for (int k = 0; k < 3; k++) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s * b; A[q][k] = s * a + c * b; }
            for (int k = 0; k < 3; k++) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
        }
    }
}

void icpo_normals(const void *tgt, int n_tgt, int stride, double radius, float *normals)
{
    /* uniform grid with cell = radius: neighbours live in the 27 surrounding cells */
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n_tgt; i++) { const float *p = pt(tgt, i, stride); for (int a = 0; a < 3; a++) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; } }
    int dim[3]; size_t cells = 1;
    for (int a = 0; a < 3; a++) { dim[a] = n_tgt ? (int)floor(((double)mx[a] - mn[a]) / radius) + 1 : 1; cells *= (size_t)dim[a]; }
    int *start = (int *)calloc(cells + 1, sizeof(int));
    int *items = (int *)malloc(sizeof(int) * (size_t)(n_tgt > 0 ? n_tgt : 1));
    int *cell = (int *)malloc(sizeof(int) * (size_t)(n_tgt > 0 ? n_tgt : 1));
    for (int i = 0; i < n_tgt; i++) {
        const float *p = pt(tgt, i, stride);
        int c[3];
        for (int a = 0; a < 3; a++) { c[a] = (int)floor(((double)p[a] - mn[a]) / radius); if (c[a] >= dim[a]) c[a] = dim[a] - 1; if (c[a] < 0) c[a] = 0; }
        cell[i] = (c[2] * dim[1] + c[1]) * dim[0] + c[0];
        start[cell[i] + 1]++;
    }
    for (size_t k = 0; k < cells; k++) start[k + 1] += start[k];
    int *fill = (int *)malloc(sizeof(int) * cells);
    memcpy(fill, start, sizeof(int) * cells);
    for (int i = 0; i < n_tgt; i++) items[fill[cell[i]]++] = i;
    const double r2 = radius * radius;
    for (int i = 0; i < n_tgt; i++) {
        const float *p = pt(tgt, i, stride);
        const int cz = cell[i] / (dim[0] * dim[1]), cy = (cell[i] / dim[0]) % dim[1], cx = cell[i] % dim[0];
        double sum[3] = {0, 0, 0}, sq[6] = {0, 0, 0, 0, 0, 0};
        int cnt = 0;
        for (int z = cz - 1; z <= cz + 1; z++) for (int y = cy - 1; y <= cy + 1; y++) for (int x = cx - 1; x <= cx + 1; x++) {
            if (x < 0 || y < 0 || z < 0 || x >= dim[0] || y >= dim[1] || z >= dim[2]) continue;
            const int c = (z * dim[1] + y) * dim[0] + x;
            for (int k = start[c]; k < start[c + 1]; k++) {
                const float *q = pt(tgt, items[k], stride);
                const double dx = (double)q[0] - p[0], dy = (double)q[1] - p[1], dz = (double)q[2] - p[2];
                if (dx * dx + dy * dy + dz * dz > r2) continue;
                sum[0] += dx; sum[1] += dy; sum[2] += dz;             /* relative to p: well conditioned */
                sq[0] += dx * dx; sq[1] += dx * dy; sq[2] += dx * dz; sq[3] += dy * dy; sq[4] += dy * dz; sq[5] += dz * dz;
                cnt++;
            }
        }
        float *n = normals + (size_t)i * 3;
        n[0] = n[1] = n[2] = 0.0f;
        if (cnt < 3) continue;
        const double N = (double)cnt, m0 = sum[0] / N, m1 = sum[1] / N, m2 = sum[2] / N;
        double C[3][3], V[3][3];
        C[0][0] = sq[0] / N - m0 * m0; C[0][1] = sq[1] / N - m0 * m1; C[0][2] = sq[2] / N - m0 * m2;
        C[1][1] = sq[3] / N - m1 * m1; C[1][2] = sq[4] / N - m1 * m2; C[2][2] = sq[5] / N - m2 * m2;
        C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
        jacobi_eig3(C, V);
        int m = 0;
        for (int a = 1; a < 3; a++) if (C[a][a] < C[m][m]) m = a;
        n[0] = (float)V[0][m]; n[1] = (float)V[1][m]; n[2] = (float)V[2][m];
    }
    free(start); free(items); free(cell); free(fill);
}

/* 6x6 solve by Gaussian elimination with partial pivoting; returns 0 when singular */
static int solve6(double A[6][6], double b[6], double x[6])
{
    int perm[6];
    for (int i = 0; i < 6; i++) perm[i] = i;
    for (int c = 0; c < 6; c++) {
        int piv = c; double best = fabs(A[c][c]);
        for (int r = c + 1; r < 6; r++) if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); piv = r; }
        if (!(best > 1e-300)) return 0;
        if (piv != c) { for (int k = 0; k < 6; k++) { double t = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = t; } double t = b[c]; b[c] = b[piv]; b[piv] = t; }
        for (int r = c + 1; r < 6; r++) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k < 6; k++) A[r][k] -= f * A[c][k];
            b[r] -= f * b[c];
        }
    }
    for (int r = 5; r >= 0; r--) {
        double s = b[r];
        for (int k = r + 1; k < 6; k++) s -= A[r][k] * x[k];
        x[r] = s / A[r][r];
    }
    (void)perm;
    return 1;
}

/* TransformationEstimationPointToPlaneLLS: minimise sum(((R p + t - q) . n)^2) linearised in the angles
 * (Low 2004); the rotation is rebuilt exactly from the three angles (Rz(g) Ry(b) Rx(a)). */
static int estimate_point_to_plane(const void *src, const void *tgt, int stride, const float *normals,
                                   const int *si, const int *ti, int n, float T[16])
{
    double ATA[6][6] = {{0}}, ATb[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        const float *p = pt(src, si[i], stride), *q = pt(tgt, ti[i], stride);
        const float *nn = normals + (size_t)ti[i] * 3;
        const double px = p[0], py = p[1], pz = p[2], nx = nn[0], ny = nn[1], nz = nn[2];
        const double row[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
        const double r = ((double)q[0] - px) * nx + ((double)q[1] - py) * ny + ((double)q[2] - pz) * nz;
        for (int a = 0; a < 6; a++) { for (int b = 0; b < 6; b++) ATA[a][b] += row[a] * row[b]; ATb[a] += row[a] * r; }
    }
    double x[6];
    if (!solve6(ATA, ATb, x)) return -1;
    const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
    T[0] = (float)(cg * cb); T[1] = (float)(-sg * ca + cg * sb * sa); T[2] = (float)(sg * sa + cg * sb * ca);  T[3] = (float)x[3];
    T[4] = (float)(sg * cb); T[5] = (float)(cg * ca + sg * sb * sa);  T[6] = (float)(-cg * sa + sg * sb * ca); T[7] = (float)x[4];
    T[8] = (float)(-sb);     T[9] = (float)(cb * sa);                 T[10] = (float)(cb * ca);                T[11] = (float)x[5];
    T[12] = T[13] = T[14] = 0.0f; T[15] = 1.0f;
    return 0;
}

/* pcl::IterativeClosestPoint::align */
int icpo_icp_align(const void *src, int n_src, const void *tgt, int n_tgt, int stride_bytes,
                   const icpo_params *p, float T[16], float *fitness, int *converged, int *iterations)
{
    const int use_grid = n_tgt > 2048;
    unsigned char *work = (unsigned char *)malloc((size_t)(n_src > 0 ? n_src : 1) * (size_t)stride_bytes);
    memcpy(work, src, (size_t)n_src * (size_t)stride_bytes);
    int *nn = (int *)malloc(sizeof(int) * (size_t)(n_src > 0 ? n_src : 1));
    float *d2 = (float *)malloc(sizeof(float) * (size_t)(n_src > 0 ? n_src : 1));
    int *si = (int *)malloc(sizeof(int) * (size_t)(n_src > 0 ? n_src : 1));
    int *ti = (int *)malloc(sizeof(int) * (size_t)(n_src > 0 ? n_src : 1));
    float final[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float *normals = NULL;
    if (p->estimator == 1) {
        normals = (float *)malloc(sizeof(float) * 3 * (size_t)(n_tgt > 0 ? n_tgt : 1));
        icpo_normals(tgt, n_tgt, stride_bytes, p->normal_radius, normals);
    }
    const float maxd2 = (float)(p->max_correspondence_dist * p->max_correspondence_dist);
    double mse_prev = DBL_MAX;
    int iter = 0, conv = 0;
    while (!conv) {
        icpo_nn(work, n_src, tgt, n_tgt, stride_bytes, use_grid, nn, d2);
        int nc = 0; double sum_d2 = 0;
        for (int i = 0; i < n_src; i++)
            if (nn[i] >= 0 && d2[i] <= maxd2) { si[nc] = i; ti[nc] = nn[i]; sum_d2 += (double)d2[i]; nc++; }
        if (nc < 3) { conv = 0; break; }                                  /* not enough correspondences */
        float Tinc[16];
        if (p->estimator == 1) {
            if (estimate_point_to_plane(work, tgt, stride_bytes, normals, si, ti, nc, Tinc) != 0) { conv = 0; break; }
        } else {
            estimate_rigid(work, tgt, stride_bytes, si, ti, nc, Tinc);
        }
        icpo_transform(work, n_src, stride_bytes, Tinc, work);
        mat4_mul(Tinc, final, final);
        iter++;
        /* DefaultConvergenceCriteria */
        if (iter >= p->max_iterations) { conv = 1; break; }
        const double cos_angle = 0.5 * ((double)Tinc[0] + (double)Tinc[5] + (double)Tinc[10] - 1.0);
        const double tsq = (double)Tinc[3] * Tinc[3] + (double)Tinc[7] * Tinc[7] + (double)Tinc[11] * Tinc[11];
        if (cos_angle >= 1.0 - p->transformation_epsilon && tsq <= p->transformation_epsilon) { conv = 1; break; }
        const double mse = sum_d2 / (double)nc;
        if (fabs(mse - mse_prev) < 1e-12) { conv = 1; break; }
        if (fabs(mse - mse_prev) / mse_prev < p->euclidean_fitness_epsilon) { conv = 1; break; }
        mse_prev = mse;
    }
    /* getFitnessScore(): original source moved by `final`, mean squared NN distance over all points */
    icpo_transform(src, n_src, stride_bytes, final, work);
    icpo_nn(work, n_src, tgt, n_tgt, stride_bytes, use_grid, nn, d2);
    double fs = 0; int nr = 0;
    for (int i = 0; i < n_src; i++) if (nn[i] >= 0) { fs += (double)d2[i]; nr++; }
    if (fitness) *fitness = nr > 0 ? (float)(fs / (double)nr) : FLT_MAX;
    memcpy(T, final, sizeof final);
    if (converged) *converged = conv;
    if (iterations) *iterations = iter;
    free(work); free(nn); free(d2); free(si); free(ti); free(normals);
    return 0;
}

/* ---- RANSAC (deterministic restatement, see icp_oracle.h) ------------------------------------------- */
static unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static void ransac_sample(unsigned long long seed, int h, int n, int idx[3])
{
    unsigned long long ctr = 0;
    for (int m = 0; m < 3; ) {
        const unsigned long long r = splitmix64(seed ^ splitmix64(((unsigned long long)h << 20) + ctr));
        ctr++;
        const int cand = (int)(r % (unsigned long long)n);
        int dup = 0;
        for (int q = 0; q < m; q++) dup |= idx[q] == cand;
        if (!dup) idx[m++] = cand;
    }
}

/* rigid transform of 3 pairs in fp64: T[12] row-major 3x4 */
static void rigid3(const float *p[3], const float *q[3], double T[12])
{
    double pm[3] = {0, 0, 0}, qm[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) for (int a = 0; a < 3; a++) { pm[a] += (double)p[i][a]; qm[a] += (double)q[i][a]; }
    for (int a = 0; a < 3; a++) { pm[a] /= 3.0; qm[a] /= 3.0; }
    double S[3][3] = {{0}};
    for (int i = 0; i < 3; i++) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++)
        S[a][b] += ((double)p[i][a] - pm[a]) * ((double)q[i][b] - qm[b]);
    double R[3][3];
    rotation_from_S(S, R);
    for (int a = 0; a < 3; a++) {
        for (int b = 0; b < 3; b++) T[a * 4 + b] = R[a][b];
        T[a * 4 + 3] = qm[a] - (R[a][0] * pm[0] + R[a][1] * pm[1] + R[a][2] * pm[2]);
    }
}

static int is_inlier(const double T[12], const float *p, const float *q, double thr2)
{
    const double x = p[0], y = p[1], z = p[2];
    const double dx = (T[0] * x + T[1] * y + T[2] * z + T[3]) - (double)q[0];
    const double dy = (T[4] * x + T[5] * y + T[6] * z + T[7]) - (double)q[1];
    const double dz = (T[8] * x + T[9] * y + T[10] * z + T[11]) - (double)q[2];
    return (dx * dx + dy * dy) + dz * dz < thr2;
}

int icpo_ransac(const void *src, const void *tgt, int stride, const int *si, const int *ti,
                int n_corr, int max_iterations, double inlier_threshold, unsigned long long seed,
                int *inlier_mask, int *best_hypothesis, double T_best[12])
{
    if (n_corr < 3) return -1;
    const double thr2 = inlier_threshold * inlier_threshold;
    int best = -1, best_h = -1;
    double Tb[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int h = 0; h < max_iterations; h++) {
        int idx[3];
        ransac_sample(seed, h, n_corr, idx);
        const float *p[3], *q[3];
        for (int m = 0; m < 3; m++) { p[m] = pt(src, si[idx[m]], stride); q[m] = pt(tgt, ti[idx[m]], stride); }
        double T[12];
        rigid3(p, q, T);
        int cnt = 0;
        for (int i = 0; i < n_corr; i++) cnt += is_inlier(T, pt(src, si[i], stride), pt(tgt, ti[i], stride), thr2);
        if (cnt > best) { best = cnt; best_h = h; memcpy(Tb, T, sizeof Tb); }
    }
    if (inlier_mask)
        for (int i = 0; i < n_corr; i++) inlier_mask[i] = is_inlier(Tb, pt(src, si[i], stride), pt(tgt, ti[i], stride), thr2);
    if (best_hypothesis) *best_hypothesis = best_h;
    if (T_best) memcpy(T_best, Tb, sizeof Tb);
    return best < 0 ? 0 : best;
}

int icpo_geometric_verification(const void *src, int n_src, const void *tgt, int n_tgt, int stride,
                                int ransac_iterations, double inlier_threshold, double inlier_ratio,
                                unsigned long long seed, float T[16], int *success, int *n_corr, int *n_inliers)
{
    int *nn = (int *)malloc(sizeof(int) * (size_t)(n_src > 0 ? n_src : 1));
    int *si = (int *)malloc(sizeof(int) * (size_t)(n_src > 0 ? n_src : 1));
    int *ti = (int *)malloc(sizeof(int) * (size_t)(n_src > 0 ? n_src : 1));
    int *mask = (int *)malloc(sizeof(int) * (size_t)(n_src > 0 ? n_src : 1));
    icpo_nn(src, n_src, tgt, n_tgt, stride, n_tgt > 2048, nn, NULL);          /* DM.h:1211-1215 */
    int nc = 0;
    for (int i = 0; i < n_src; i++) if (nn[i] >= 0) { si[nc] = i; ti[nc] = nn[i]; nc++; }
    int ok = 0, ninl = 0;
    for (int k = 0; k < 16; k++) T[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    if (nc >= 3) {
        ninl = icpo_ransac(src, tgt, stride, si, ti, nc, ransac_iterations, inlier_threshold, seed, mask, NULL, NULL);  /* DM.h:1218-1225 */
        int m = 0;
        for (int i = 0; i < nc; i++) if (mask[i]) { si[m] = si[i]; ti[m] = ti[i]; m++; }
        if (m >= 3) estimate_rigid(src, tgt, stride, si, ti, m, T);             /* DM.h:1228-1230 */
        ok = !((double)ninl < inlier_ratio * (double)nc);                       /* DM.h:1238 */
    }
    if (success) *success = ok;
    if (n_corr) *n_corr = nc;
    if (n_inliers) *n_inliers = ninl;
    free(nn); free(si); free(ti); free(mask);
    return 0;
}

/* ---- voxel grid ---------------------------------------------------------------------------------------- */
typedef struct { long long idx; int pt; } vox_key;
static int cmp_vox(const void *a, const void *b)
{
    const vox_key *x = (const vox_key *)a, *y = (const vox_key *)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return (x->pt > y->pt) - (x->pt < y->pt);
}

int icpo_voxel_grid(const void *in, int n, int stride, float leaf, void *out)
{
    const float inv = 1.0f / leaf;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    int nfinite = 0;
    for (int i = 0; i < n; i++) {
        const float *p = pt(in, i, stride);
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
        for (int a = 0; a < 3; a++) { if (p[a] < mn[a]) mn[a] = p[a]; if (p[a] > mx[a]) mx[a] = p[a]; }
        nfinite++;
    }
    if (nfinite == 0) return 0;
    long long minb[3], divb[3];
    for (int a = 0; a < 3; a++) {
        /* a bound outside int32 (PCL casts it to int: undefined) counts as "too large for the leaf", and the voxel count is formed
         * without wrapping (UBSan, make sanitize: 999999874000003969 * 999999937 overflowed here) */
        const float lo = floorf(mn[a] * inv), hi = floorf(mx[a] * inv);
        if (!(lo >= -2147483648.f && hi <= 2147483520.f)) return -1;
        minb[a] = (long long)lo;
        divb[a] = (long long)hi - minb[a] + 1;
        if (divb[a] > 2147483647LL) return -1;
    }
    if (divb[0] * divb[1] > 2147483647LL || divb[0] * divb[1] * divb[2] > 2147483647LL) return -1;
    vox_key *keys = (vox_key *)malloc(sizeof(vox_key) * (size_t)nfinite);
    int m = 0;
    for (int i = 0; i < n; i++) {
        const float *p = pt(in, i, stride);
        if (!(isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2]))) continue;
        const long long i0 = (long long)floorf(p[0] * inv) - minb[0];
        const long long i1 = (long long)floorf(p[1] * inv) - minb[1];
        const long long i2 = (long long)floorf(p[2] * inv) - minb[2];
        keys[m].idx = i0 + i1 * divb[0] + i2 * divb[0] * divb[1];
        keys[m].pt = i; m++;
    }
    qsort(keys, (size_t)m, sizeof(vox_key), cmp_vox);
    const int has_i = stride >= 20;
    int nout = 0;
    for (int a = 0; a < m; ) {
        int b = a;
        float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
        while (b < m && keys[b].idx == keys[a].idx) {
            const float *p = pt(in, keys[b].pt, stride);
            sx += p[0]; sy += p[1]; sz += p[2];
            if (has_i) si += p[4];
            b++;
        }
        const float cnt = (float)(b - a);
        float *o = (float *)((unsigned char *)out + (size_t)nout * (size_t)stride);
        memset(o, 0, (size_t)stride);
        o[0] = sx / cnt; o[1] = sy / cnt; o[2] = sz / cnt;
        if (has_i) o[4] = si / cnt;
        nout++;
        a = b;
    }
    free(keys);
    return nout;
}

void icpo_pose_to_matrix(float x, float y, float z, float roll, float pitch, float yaw, float T[16])
{
    const float A = cosf(yaw), B = sinf(yaw), C = cosf(pitch), D = sinf(pitch), E = cosf(roll), F = sinf(roll);
    const float DE = D * E, DF = D * F;
    T[0] = A * C; T[1] = A * DF - B * E; T[2] = B * F + A * DE; T[3] = x;
    T[4] = B * C; T[5] = A * E + B * DF; T[6] = B * DE - A * F; T[7] = y;
    T[8] = -D;    T[9] = C * F;          T[10] = C * E;         T[11] = z;
    T[12] = 0.f;  T[13] = 0.f;           T[14] = 0.f;           T[15] = 1.f;
}
