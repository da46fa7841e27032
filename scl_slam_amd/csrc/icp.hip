// icp.hip -- geometric verification of loop candidates on the GPU.
//
// Replaces the PCL objects the reference uses inline:
//   pcl::IterativeClosestPoint::align + getFitnessScore   distributedMapping.h:1108-1121
//   CorrespondenceEstimation::determineCorrespondences     distributedMapping.h:1211-1215
//   TransformationEstimationSVD::estimateRigidTransformation  distributedMapping.h:1228-1230
//   paramsServer::transformPointCloud                      distributedMapping.h:234-253
//
// Kernels (SURVEY 8(d) prices an iteration at (n_src + n_tgt) * 16 bytes; an EXACT nearest-neighbour search is bound by its
// comparisons and gathers, not by those bytes: DESIGN.md section 4, K4-K6):
//   K4a grid build   target cloud -> uniform grid (bbox, count, scan, scatter as float4 xyz+index); for the candidates of one scan
//                    as one chain of batched launches (icp_batch_prepare_all)
//   K4b nn search    in memory: lanes of a group share the rows of a shell / of the ball around the previous neighbour (nn_core);
//                    fp32 distances ((dx*dx+dy*dy)+dz*dz), ties -> lowest target index (a total order: the result does not
//                    depend on the order points were scattered into a cell).  Small launches and what the tiles leave over.
//   K4c tile search  the loop's search for large launches: 256 Hilbert-ordered sources per workgroup, the box of the cells their
//                    balls reach staged in LDS, every query walks its cells there; the same neighbours bit for bit; the
//                    workgroup's correspondences reduced in the same launch (icp_tile_search_kernel, icp_tile_finish_kernel)
//   K5  reduce       fp64 sums of the augmented outer product [p;1][q;1]^T over correspondences
//                    (gives the cross-covariance, both centroids and the count in one pass) + sum d2;
//                    on the matrix cores: one v_mfma_f64_4x4x4_4b_f64 per 16 correspondences (tile_reduce: one record per 256
//                    sources, the same records whichever search found the neighbours)
//   K5b solve        the records summed in a fixed order, then one thread: centred covariance, closed-form rotation (polar factor;
//                    Horn quaternion + cyclic Jacobi for near-singular cases, fp64), incremental transform, PCL's convergence
//                    criteria, final = T*final
//   K6  transform    working source cloud <- T_inc * cloud (fp32, no FMA), folded into the next search
//   normals          point to plane: PCA of the neighbours within a radius through the same grid (normals_body)
// The whole ICP loop is enqueued without host round trips: every kernel looks at a device-side `done` flag; the flags travel to the
// host every four iterations and are looked at one period later.
#include "icp.hpp"
#include "device_sort.hpp"


#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_common.hpp"

namespace scl {

namespace {

#ifndef SCL_GRID_DIV
#define SCL_GRID_DIV 96
#endif
constexpr int kGridDiv = SCL_GRID_DIV;   // cells along the longest bbox edge
constexpr int kMaxCells = (kGridDiv + 1) * (kGridDiv + 1) * (kGridDiv + 1);
constexpr int kRedBlocks = 256;
constexpr int kNSum = 17;                // 16 entries of sum [p;1][q;1]^T + sum d2

struct IcpState {
    float mn[3]; float h; int dim[3]; int cells;         // grid
    float final_T[16]; float inc_T[16];
    double mse_prev;
    double fitness;
    int iter; int done; int converged; int n_corr;
    double sums[kNSum];
    unsigned int pad_[2];
};

enum Buf { B_SRC = 0, B_TGT, B_WORK, B_TSORT, B_CSTART, B_CFILL, B_NNI, B_NND, B_PART, B_STATE, B_BBOX, B_SI, B_TI, B_OUT, B_MASK, B_HYP, B_PROB,
           B_NNQ, B_PERM, B_SORT, B_FLAG };
static_assert(B_FLAG < (int)(sizeof(IcpWorkspace::buf) / sizeof(void *)), "IcpWorkspace::buf is too short");
constexpr int B_NORM = B_OUT;             // target normals share the slot of the raw-transform output (never live together)

int ensure(IcpWorkspace *ws, int k, size_t bytes, std::string *err)
{
    if (bytes <= ws->cap[k]) return SCL_OK;
    if (ws->buf[k]) { (void)hipFree(ws->buf[k]); ws->buf[k] = nullptr; ws->cap[k] = 0; }
    size_t nb = bytes + bytes / 4 + 256;
    if (hipMalloc(&ws->buf[k], nb) != hipSuccess) { if (err) *err = "icp: hipMalloc failed"; return SCL_ERR_NOMEM; }
    ws->cap[k] = nb;
    return SCL_OK;
}

#define ICP_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e__ = (call);                                                        \
        if (e__ != hipSuccess) { if (err) *err = std::string(#call) + ": " + hipGetErrorString(e__); return SCL_ERR_HIP; } \
    } while (0)

__device__ __forceinline__ float3 load_xyz(const unsigned char *base, int i, int stride)
{
    const float *f = reinterpret_cast<const float *>(base + (size_t)i * (size_t)stride);
    return make_float3(f[0], f[1], f[2]);
}

// ---- K4a ---------------------------------------------------------------------------
__device__ __forceinline__ void bbox_partial_body(const unsigned char *pts, int n, int stride, float *part)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float3 p = load_xyz(pts, i, stride);
        mn[0] = fminf(mn[0], p.x); mn[1] = fminf(mn[1], p.y); mn[2] = fminf(mn[2], p.z);
        mx[0] = fmaxf(mx[0], p.x); mx[1] = fmaxf(mx[1], p.y); mx[2] = fmaxf(mx[2], p.z);
    }
    __shared__ float s[6][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, kWave));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, kWave));
        }
    }
    const int wv = threadIdx.x / kWave;
    if ((threadIdx.x & 63) == 0) for (int a = 0; a < 3; ++a) { s[a][wv] = mn[a]; s[3 + a][wv] = mx[a]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            float m = s[a][0], M = s[3 + a][0];
            for (int w = 1; w < (int)blockDim.x / kWave; ++w) { m = fminf(m, s[a][w]); M = fmaxf(M, s[3 + a][w]); }
            part[blockIdx.x * 6 + a] = m; part[blockIdx.x * 6 + 3 + a] = M;
        }
    }
}
__global__ void bbox_partial_kernel(const unsigned char *pts, int n, int stride, float *part) { bbox_partial_body(pts, n, stride, part); }

__device__ __forceinline__ void grid_setup_body(const float *part, int nblocks, int n, IcpState *st, int *cell_start)
{
    // one wave: lane l folds the partial boxes l, l+64, ..., then a butterfly (min/max are order independent)
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int b = threadIdx.x; b < nblocks; b += 64)
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], part[b * 6 + a]); mx[a] = fmaxf(mx[a], part[b * 6 + 3 + a]); }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, 64)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, 64)); }
    if (threadIdx.x == 0) {
        if (n <= 0) { for (int a = 0; a < 3; ++a) { mn[a] = 0.f; mx[a] = 0.f; } }
        float ext = 0.f;
        for (int a = 0; a < 3; ++a) ext = fmaxf(ext, mx[a] - mn[a]);
        float h = ext / (float)kGridDiv;
        if (!(h > 1e-6f)) h = 1.0f;
        int cells = 1;
        for (int a = 0; a < 3; ++a) {
            int d = (int)floorf((mx[a] - mn[a]) / h) + 1;
            d = d < 1 ? 1 : (d > kGridDiv + 1 ? kGridDiv + 1 : d);
            st->dim[a] = d; st->mn[a] = mn[a]; cells *= d;
        }
        st->h = h; st->cells = cells;
    }
    (void)cell_start;                                      // zeroed by the caller (hipMemsetAsync of the whole table)
}
__global__ void grid_setup_kernel(const float *part, int nblocks, int n, IcpState *st, int *cell_start) { grid_setup_body(part, nblocks, n, st, cell_start); }

__device__ __forceinline__ int cell_index(const IcpState *st, float3 p, int c[3])
{
    const float v[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float f = floorf((v[a] - st->mn[a]) / st->h);
        int ci = (f != f) ? 0 : (f < 0.f ? 0 : (f >= (float)st->dim[a] ? st->dim[a] - 1 : (int)f));
        c[a] = ci;
    }
    return (c[2] * st->dim[1] + c[1]) * st->dim[0] + c[0];
}

__device__ __forceinline__ void grid_count_body(const unsigned char *pts, int n, int stride, const IcpState *st, int *cell_start)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int c[3];
        const int cell = cell_index(st, load_xyz(pts, i, stride), c);
        atomicAdd(&cell_start[cell + 1], 1);
    }
}

__global__ void grid_count_kernel(const unsigned char *pts, int n, int stride, const IcpState *st, int *cell_start) { grid_count_body(pts, n, stride, st, cell_start); }

__device__ __forceinline__ void grid_scatter_body(const unsigned char *pts, int n, int stride, const IcpState *st,
                                                  int *cell_fill, float4 *sorted)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int c[3];
        const float3 p = load_xyz(pts, i, stride);
        const int cell = cell_index(st, p, c);
        const int pos = atomicAdd(&cell_fill[cell], 1);
        sorted[pos] = make_float4(p.x, p.y, p.z, __int_as_float(i));
    }
}
__global__ void grid_scatter_kernel(const unsigned char *pts, int n, int stride, const IcpState *st, int *cell_fill, float4 *sorted)
{
    grid_scatter_body(pts, n, stride, st, cell_fill, sorted);
}

// ---- K4b ---------------------------------------------------------------------------
// Exact nearest neighbour through the grid, kNnGroup lanes per query.  A query walks cubic shells of cells
// around its own cell; the (z, y) rows of a shell are dealt round robin to the lanes of its group (a row
// on a z- or y-face of the shell is one contiguous range of `sorted`, an interior row contributes its two
// x-face cells), the group's (distance, index) minimum is formed with DPP exchanges, and the walk ends once
// the nearest face of the shell is farther than the best distance.  One lane per query left ~1.5 waves per
// SIMD chasing dependent loads; eight lanes per query give the memory system 8x the requests in flight and
// cut the dependent round trips per shell from (rows) to (rows / 8).  min over (d, index) is order
// independent: results are bit-identical to the serial walk (ties -> lowest target index).
constexpr int kNnGroup = 8;

template <int G>
__device__ __forceinline__ void group_min(float &d, int &j)
{
    // xor 1, xor 2 (quad permutes), then mirror inside each half row (lane i <-> 7 - i): all 8 lanes end equal
#define SCL_GMIN_STEP(CTRL)                                                                        \
    {                                                                                              \
        const float od = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), CTRL, 0xf, 0xf, false)); \
        const int oj = __builtin_amdgcn_update_dpp(0, j, CTRL, 0xf, 0xf, false);                    \
        const bool take = (od < d) | ((od == d) & (oj < j));                                        \
        d = take ? od : d; j = take ? oj : j;                                                       \
    }
    if (G >= 2) SCL_GMIN_STEP(0xB1)
    if (G >= 4) SCL_GMIN_STEP(0x4E)
    if (G >= 8) SCL_GMIN_STEP(0x141)
#undef SCL_GMIN_STEP
}

// apply_iter >= 0: K6 folded in -- the increment of the previous iteration's solve (st->inc_T) moves the working point
// first, and the moved point is written back for the reduction that follows (distributedMapping.h:247-249
// arithmetic: fp32, no FMA).  Every search of the loop follows exactly one solve; a solve that fails or converges sets
// st->done, on which this kernel returns at once -- so no iteration number has to travel with the launch, and the same
// launch serves every iteration, of one alignment or of a whole batch of them.
// warm != 0: nn_idx[i] still holds the query's neighbour of the previous iteration.  Its distance to the moved query is
// an upper bound on the new minimum, and every cell (row) of a shell whose box lies strictly farther away than the
// best distance known so far is skipped: after the first iteration a query typically looks at its own cell and the
// one or two neighbours its small ball reaches into, not at the 26 cells of the first shell.  Exactness is untouched:
// a skipped box cannot hold a point at a smaller or equal distance (ties -> lowest index are decided among the points
// at exactly the minimum distance, whose boxes are never skipped).
// G = lanes per query: 8 for a cold search (whole shells to look at), 2 once the previous neighbour bounds the ball
// (most queries are done after their own cell; eight lanes would mostly idle and only a quarter of the queries
// would be resident at a time).
// The search proper: the group's G lanes (sub = this lane's place in it) find the nearest target point of p, starting from
// (best, bi) -- the previous neighbour's distance and index when warm_start, (FLT_MAX, -1) otherwise.
template <int G>
__device__ __forceinline__ void nn_core(const float3 p, const IcpState *st, const int *cell_start, const float4 *sorted,
                                        const int sub, const bool warm_start, float &best, int &bi)
{
    int c[3];
    cell_index(st, p, c);
    const int dx = st->dim[0], dy = st->dim[1], dz = st->dim[2];
    const float h = st->h;
    const int maxdim = max(dx, max(dy, dz));
    const float pv[3] = {p.x, p.y, p.z};
    const int dm[3] = {dx, dy, dz};
    float dout2[3], dout_sum = 0.f;                              // squared distance from p to the grid's box, per axis
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float below = st->mn[a] - pv[a], above = pv[a] - (st->mn[a] + (float)dm[a] * h);
        float d = fmaxf(fmaxf(below, above), 0.f);
        d = (d == d) ? d : 0.f;                                  // NaN coordinates: no help from this axis
        dout2[a] = d * d;
        dout_sum += dout2[a];
    }
    const float gx0 = st->mn[0], gy0 = st->mn[1], gz0 = st->mn[2];
    // four points per step, their loads issued together (a load per step and a wait behind it left the lane
    // chasing one L2 latency per point); the last step re-reads the range's final point, which cannot change
    // a (distance, index) minimum
    auto scan = [&](int kb, int ke) {
        for (int k = kb; k < ke; k += 4) {
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = sorted[k + u < ke ? k + u : ke - 1];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ex = p.x - q[u].x, ey = p.y - q[u].y, ez = p.z - q[u].z;
                const float d = (ex * ex + ey * ey) + ez * ez;
                const int j = __float_as_int(q[u].w);
                if ((d < best) | ((d == best) & (j < bi))) { best = d; bi = j; }
            }
        }
    };
    // cells [a, b] of one axis that the interval [v - reach, v + reach] meets, clamped to the grid (empty: a > b)
    auto reach_cells = [&](float v, float g0, int dim, float reach, int &a, int &b) {
        const float fa = floorf((v - reach - g0) / h), fb = floorf((v + reach - g0) / h);
        a = 0; b = dim - 1;
        if (fa == fa && fb == fb) {
            const int ia = fa < -1.0e9f ? 0 : (fa > 1.0e9f ? dim : (int)fa), ib = fb < -1.0e9f ? -1 : (fb > 1.0e9f ? dim - 1 : (int)fb);
            a = max(a, ia); b = min(b, ib);
        }
    };
    // One pass over the ball of the best distance known so far: the (z, y) rows of its bounding box, every row ONE contiguous
    // range of `sorted` trimmed to the cells the ball reaches -- O(R^2) rows, each O(1) when empty.  (The shell walk below
    // meets a row once per shell: O(R^3) row visits for a neighbour R cells away, which is what a candidate that does not
    // match costs.)  Every cell that meets the closed ball is visited; the lanes' minima only shrink the ball further.
    auto ball_pass = [&]() {
        const float reach0 = sqrtf(best) * 1.0005f + 1e-6f * h;
        int z0, z1, y0, y1;
        reach_cells(p.z, gz0, dz, reach0, z0, z1);
        reach_cells(p.y, gy0, dy, reach0, y0, y1);
        const int ny = y1 - y0 + 1, nrows = (z1 >= z0 && ny > 0) ? (z1 - z0 + 1) * ny : 0;
        for (int t = sub; t < nrows; t += G) {
            const int zz = t / ny;
            const int z = z0 + zz, y = y0 + (t - zz * ny);
            const int row = (z * dy + y) * dx;
            const float ylo = gy0 + (float)y * h, zlo = gz0 + (float)z * h;
            const float ddy = fmaxf(fmaxf(ylo - p.y, p.y - (ylo + h)), 0.f), ddz = fmaxf(fmaxf(zlo - p.z, p.z - (zlo + h)), 0.f);
            const float dyz2 = (ddy * ddy + ddz * ddz) * 0.9995f;
            if (dyz2 > best) continue;
            int xa, xb;
            reach_cells(p.x, gx0, dx, sqrtf(best - dyz2) * 1.0005f + 1e-6f * h, xa, xb);
            if (xa <= xb) scan(cell_start[row + xa], cell_start[row + xb + 1]);
        }
        group_min<G>(best, bi);
    };
    if (warm_start) ball_pass();                                 // the previous neighbour bounds the ball
    // Cold search: shells until the first one that holds a point (its distance bounds the ball), then the ball once.
    for (int r = 0; !warm_start && r <= maxdim; ++r) {
        const int lo0 = c[0] - r, hi0 = c[0] + r, lo1 = c[1] - r, hi1 = c[1] + r, lo2 = c[2] - r, hi2 = c[2] + r;
        const int z0 = max(lo2, 0), z1 = min(hi2, dz - 1), y0 = max(lo1, 0), y1 = min(hi1, dy - 1);
        const int ny = y1 - y0 + 1, nrows = (z1 - z0 + 1) * ny;
        const int xs = max(lo0, 0), xe = min(hi0, dx - 1);
        for (int t = sub; t < nrows; t += G) {
            const int zz = t / ny;
            const int z = z0 + zz, y = y0 + (t - zz * ny);
            const bool face = (z == lo2) | (z == hi2) | (y == lo1) | (y == hi1);
            const int row = (z * dy + y) * dx;
            // squared distance from p to the row's (y, z) cell interval; boxes are taken a hair smaller than the cells
            // (0.9995 on the squared distance) so that rounding in the box arithmetic can never skip a cell that
            // holds a point at distance <= best
            const float ylo = gy0 + (float)y * h, zlo = gz0 + (float)z * h;
            const float ddy = fmaxf(fmaxf(ylo - p.y, p.y - (ylo + h)), 0.f), ddz = fmaxf(fmaxf(zlo - p.z, p.z - (zlo + h)), 0.f);
            const float dyz2 = (ddy * ddy + ddz * ddz) * 0.9995f;
            if (dyz2 > best) continue;                           // NaN coordinates never skip
            if (face) {
                int xa = xs, xb = xe;
                if (best < FLT_MAX) {                            // trim the row to the cells the ball reaches
                    const float reach = sqrtf(best - dyz2) * 1.0005f + 1e-6f * h;
                    const float fa = floorf((p.x - reach - gx0) / h), fb = floorf((p.x + reach - gx0) / h);
                    if (fa == fa && fb == fb) {
                        const int ia = fa < -1.0e9f ? xs : (fa > 1.0e9f ? xe + 1 : (int)fa), ib = fb < -1.0e9f ? xs - 1 : (fb > 1.0e9f ? xe : (int)fb);
                        xa = max(xa, ia); xb = min(xb, ib);
                    }
                }
                if (xa <= xb) scan(cell_start[row + xa], cell_start[row + xb + 1]);
            } else {
                if (lo0 >= 0) {
                    const float xhi = gx0 + (float)(lo0 + 1) * h;
                    const float ddx = fmaxf(p.x - xhi, 0.f);
                    if (!((ddx * ddx) * 0.9995f + dyz2 > best)) scan(cell_start[row + lo0], cell_start[row + lo0 + 1]);
                }
                if (hi0 <= dx - 1) {
                    const float xlo = gx0 + (float)hi0 * h;
                    const float ddx = fmaxf(xlo - p.x, 0.f);
                    if (!((ddx * ddx) * 0.9995f + dyz2 > best)) scan(cell_start[row + hi0], cell_start[row + hi0 + 1]);
                }
            }
        }
        group_min<G>(best, bi);
        // lower bound on the squared distance to anything not visited yet: such a point lies beyond an open face of
        // the shell (distance f along that axis) and inside the grid's box (so at least dout[a] away along every axis
        // on which p lies outside the box -- what ends the walk of a query far outside the target early)
        float bound2 = FLT_MAX;
        bool open = false;
        const int lo[3] = {lo0, lo1, lo2}, hi[3] = {hi0, hi1, hi2};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float others = dout_sum - dout2[a];
            if (lo[a] > 0) {
                float f = pv[a] - (st->mn[a] + (float)lo[a] * h); f = f < 0.f ? 0.f : f;
                bound2 = fminf(bound2, fmaxf(f * f, dout2[a]) + others); open = true;
            }
            if (hi[a] < dm[a] - 1) {
                float f = (st->mn[a] + (float)(hi[a] + 1) * h) - pv[a]; f = f < 0.f ? 0.f : f;
                bound2 = fminf(bound2, fmaxf(f * f, dout2[a]) + others); open = true;
            }
        }
        if (!open) break;
        if (bi >= 0 && best < bound2 * 0.9998f) break;
        if (bi >= 0 && best < FLT_MAX) { ball_pass(); break; }   // (every lane of the group holds the same best and bi here)
    }
}

template <int G>
__device__ __forceinline__ void nn_search_kernel_t_body(float4 *work, int n_src, const IcpState *st,
                                                          const int *cell_start, const float4 *sorted,
                                                          int *nn_idx, float *nn_d2, int check_done, int apply_iter,
                                                          const unsigned char *tgt_raw, int stride, int warm)
{
    if (check_done && st->done) return;
    if (n_src <= 0) return;
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int sub = threadIdx.x & (G - 1);
    const bool valid = gid < n_src;
    const int i = valid ? gid : n_src - 1;                       // surplus groups shadow the last query (all lanes stay in the exchanges)
    float4 pw = work[i];
    if (apply_iter >= 0 && valid) {                              // (surplus groups would re-apply it to the stored result)
        const float *T = st->inc_T;
        const float x = pw.x, y = pw.y, z = pw.z;
        pw.x = T[0] * x + T[1] * y + T[2] * z + T[3];
        pw.y = T[4] * x + T[5] * y + T[6] * z + T[7];
        pw.z = T[8] * x + T[9] * y + T[10] * z + T[11];
        pw.w = 0.f;
        if (sub == 0) work[i] = pw;
    }
    const float3 p = make_float3(pw.x, pw.y, pw.z);
    float best = FLT_MAX;
    int bi = -1;
    if (warm) {
        const int j = nn_idx[i];
        if (j >= 0) {
            const float3 q = load_xyz(tgt_raw, j, stride);
            const float ex = p.x - q.x, ey = p.y - q.y, ez = p.z - q.z;
            const float d = (ex * ex + ey * ey) + ez * ez;       // the expression the scan uses: the same bits when the scan meets j again
            if (d == d) { best = d; bi = j; }
        }
    }
    nn_core<G>(p, st, cell_start, sorted, sub, warm && bi >= 0, best, bi);
    if (valid && sub == 0) {
        nn_idx[i] = bi;
        nn_d2[i] = best;
    }
}

template <int G>
__global__ __launch_bounds__(256) void nn_search_kernel_t(float4 *work, int n_src, const IcpState *st,
                                                          const int *cell_start, const float4 *sorted,
                                                          int *nn_idx, float *nn_d2, int check_done, int apply_iter,
                                                          const unsigned char *tgt_raw, int stride, int warm)
{
    nn_search_kernel_t_body<G>(work, n_src, st, cell_start, sorted, nn_idx, nn_d2, check_done, apply_iter, tgt_raw, stride, warm);
}

// ---- K5 ----------------------------------------------------------------------------
// sums of [p;1][q;1]^T (16) and of d2 over the accepted correspondences.
// mode 0: pairs (i, nn_idx[i]) with d2 <= maxd2, p from `work`;  mode 1: explicit pairs (si[k], ti[k]).
__device__ __forceinline__ void corr_reduce_kernel_body(const float4 *work, const unsigned char *src_raw,
                                                          const unsigned char *tgt_raw, int stride, int n,
                                                          const int *nn_idx, const float *nn_d2, float maxd2,
                                                          const int *si, const int *ti, int mode,
                                                          const IcpState *st, double *partials, int check_done)
{
    if (check_done && st->done) return;
    double acc[kNSum];
#pragma unroll
    for (int k = 0; k < kNSum; ++k) acc[k] = 0.0;
    const int r_i0 = (int)(blockIdx.x * blockDim.x), r_step = (int)(gridDim.x * blockDim.x), r_i1 = n;
    for (int i = r_i0 + (int)threadIdx.x; i < r_i1; i += r_step) {
        float3 p, q;
        float d2 = 0.f;
        if (mode == 0) {
            const int j = nn_idx[i];
            d2 = nn_d2[i];
            if (j < 0 || !(d2 <= maxd2)) continue;
            const float4 pw = work[i];
            p = make_float3(pw.x, pw.y, pw.z);
            q = load_xyz(tgt_raw, j, stride);
        } else {
            if (nn_idx && nn_idx[i] == 0) continue;          // mode 1: nn_idx doubles as an optional inlier mask
            p = load_xyz(src_raw, si[i], stride);
            q = load_xyz(tgt_raw, ti[i], stride);
        }
        const double pv[4] = {(double)p.x, (double)p.y, (double)p.z, 1.0};
        const double qv[4] = {(double)q.x, (double)q.y, (double)q.z, 1.0};
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a * 4 + b] = fma(pv[a], qv[b], acc[a * 4 + b]);
        acc[16] += (double)d2;
    }
    __shared__ double s[4][kNSum];
#pragma unroll
    for (int k = 0; k < kNSum; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, kWave);
    }
    const int wv = threadIdx.x / kWave;
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < kNSum; ++k) s[wv][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < kNSum)
        partials[blockIdx.x * kNSum + threadIdx.x] = ((s[0][threadIdx.x] + s[1][threadIdx.x]) + s[2][threadIdx.x]) + s[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void corr_reduce_kernel(const float4 *work, const unsigned char *src_raw,
                                                          const unsigned char *tgt_raw, int stride, int n,
                                                          const int *nn_idx, const float *nn_d2, float maxd2,
                                                          const int *si, const int *ti, int mode,
                                                          const IcpState *st, double *partials, int check_done)
{
    corr_reduce_kernel_body(work, src_raw, tgt_raw, stride, n, nn_idx, nn_d2, maxd2, si, ti, mode, st, partials, check_done);
}

// MFMA form of K5.  v_mfma_f64_4x4x4_4b_f64 computes, in each of its four 16-lane blocks,
// D[i][j] += sum_k A[i][k] * B[k][j] (k ascending, a plain fma chain -- probed on gfx950, see
// scripts/probes/probe_mfma_f64.hip).  With A[i][k] = component i of [p;1] and B[k][j] = component j of
// [q;1] for correspondence k of the block, one instruction adds the augmented outer products of 16
// correspondences: cross-covariance, both centroids and the count at once.  Operand layout (probed):
// A[i][k] of block b sits in lane 16k + 4b + i, B[k][j] in lane 16k + 4b + j, D[i][j] in lane 16i + 4b + j.
__device__ __forceinline__ void corr_reduce_mfma_kernel_body(const float4 *work, const unsigned char *src_raw,
                                                               const unsigned char *tgt_raw, int stride, int n,
                                                               const int *nn_idx, const float *nn_d2, float maxd2,
                                                               const int *si, const int *ti, int mode,
                                                               const IcpState *st, double *partials, int check_done)
{
    if (check_done && st->done) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = lane >> 4, blk = (lane >> 2) & 3, comp = lane & 3;
    const int wave_global = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    double acc0 = 0.0, acc1 = 0.0;
    double sum_d2 = 0.0;
    const int b_first = wave_global * 32, b_step = nwaves * 32, b_end = n;
    for (int base = b_first; base < b_end; base += b_step) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {                       // two independent accumulator chains
            const int i = base + half * 16 + 4 * blk + k;            // this lane's correspondence
            double av = 0.0, bv = 0.0;
            if (i < b_end) {
                bool ok = true;
                int pi = i, qi;
                float d2 = 0.f;
                if (mode == 0) { qi = nn_idx[i]; d2 = nn_d2[i]; ok = qi >= 0 && (d2 <= maxd2); }
                else { pi = si[i]; qi = ti[i]; ok = !(nn_idx && nn_idx[i] == 0); }   // optional inlier mask
                if (ok) {
                    float pc, qc;
                    if (comp == 3) { pc = 1.0f; qc = 1.0f; }
                    else {
                        pc = mode == 0 ? reinterpret_cast<const float *>(work + pi)[comp]
                                       : reinterpret_cast<const float *>(src_raw + (size_t)pi * stride)[comp];
                        qc = reinterpret_cast<const float *>(tgt_raw + (size_t)qi * stride)[comp];
                    }
                    av = (double)pc; bv = (double)qc;
                    if (comp == 0) sum_d2 += (double)d2;
                }
            }
            if (half == 0) acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc0, 0, 0, 0);
            else           acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc1, 0, 0, 0);
        }
    }
    double d = acc0 + acc1;                                          // lane 16i + 4b + j holds block b's D[i][j]
    d += __shfl_xor(d, 4, kWave);                                    // fold the four blocks
    d += __shfl_xor(d, 8, kWave);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum_d2 += __shfl_xor(sum_d2, off, kWave);
    __shared__ double s[4][kNSum];
    if (blk == 0) s[wv][k * 4 + comp] = d;                           // lane 16i + j: entry (i, j); here k == i
    if (lane == 0) s[wv][16] = sum_d2;
    __syncthreads();
    if (threadIdx.x < kNSum)
        partials[blockIdx.x * kNSum + threadIdx.x] = ((s[0][threadIdx.x] + s[1][threadIdx.x]) + s[2][threadIdx.x]) + s[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void corr_reduce_mfma_kernel(const float4 *work, const unsigned char *src_raw,
                                                               const unsigned char *tgt_raw, int stride, int n,
                                                               const int *nn_idx, const float *nn_d2, float maxd2,
                                                               const int *si, const int *ti, int mode,
                                                               const IcpState *st, double *partials, int check_done)
{
    corr_reduce_mfma_kernel_body(work, src_raw, tgt_raw, stride, n, nn_idx, nn_d2, maxd2, si, ti, mode, st, partials, check_done);
}

// ---- K5b ---------------------------------------------------------------------------
// Rotation maximising trace(R S) for a well conditioned S with positive determinant (the ICP case:
// thousands of correspondences): the orthogonal polar factor of S^T by Newton's iteration
// X <- (g X + X^-T / g) / 2, Frobenius-scaled (Higham) while far from orthogonal.  A handful of 3x3
// cofactor inverses instead of ~8 sweeps of a 4x4 Jacobi whose dependent fp64 divisions and square
// roots cost ~40 us on one lane.  Returns false (caller falls back to Horn's quaternion) when S is
// close to singular or improper -- e.g. the rank-2 matrices of the 3-point RANSAC fits.
__device__ __forceinline__ bool rotation_polar(const double S[3][3], double R[3][3])
{
    double X[3][3];
    double fro = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { X[i][j] = S[j][i]; fro += S[j][i] * S[j][i]; }
    if (!(fro > 0.0) || !(fro < 1e300)) return false;
    const double inv_fro = 1.0 / sqrt(fro);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) X[i][j] *= inv_fro;
    bool scaled = true;
#pragma unroll 1
    for (int it = 0; it < 24; ++it) {
        double C[3][3];                                    // cofactors: X^-T = C / det
        C[0][0] = X[1][1] * X[2][2] - X[1][2] * X[2][1]; C[0][1] = X[1][2] * X[2][0] - X[1][0] * X[2][2]; C[0][2] = X[1][0] * X[2][1] - X[1][1] * X[2][0];
        C[1][0] = X[0][2] * X[2][1] - X[0][1] * X[2][2]; C[1][1] = X[0][0] * X[2][2] - X[0][2] * X[2][0]; C[1][2] = X[0][1] * X[2][0] - X[0][0] * X[2][1];
        C[2][0] = X[0][1] * X[1][2] - X[0][2] * X[1][1]; C[2][1] = X[0][2] * X[1][0] - X[0][0] * X[1][2]; C[2][2] = X[0][0] * X[1][1] - X[0][1] * X[1][0];
        const double det = X[0][0] * C[0][0] + X[0][1] * C[0][1] + X[0][2] * C[0][2];
        if (it == 0 && !(det > 1e-5)) return false;         // ||X||_F = 1: smallest singular value >~ 2e-5
        if (!(det > 0.0)) return false;
        const double inv_det = 1.0 / det;
        double a = 0.5, b = 0.5 * inv_det;
        if (scaled) {
            double nx = 0.0, nc = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) { nx += X[i][j] * X[i][j]; nc += C[i][j] * C[i][j]; }
            const double g2 = sqrt(nc / nx) * inv_det;      // gamma^2 = ||X^-1||_F / ||X||_F
            const double g = sqrt(g2);
            a = 0.5 * g; b = 0.5 * inv_det / g;
            if (fabs(g2 - 1.0) < 1e-2) scaled = false;      // near orthogonal: plain Newton converges quadratically
        }
        double delta = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double xn = a * X[i][j] + b * C[i][j];
                const double d = xn - X[i][j];
                delta += d * d;
                X[i][j] = xn;
            }
        if (!scaled && delta < 1e-28) break;               // next step would change nothing above rounding
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) R[i][j] = X[i][j];
    return true;
}

__device__ __forceinline__ void rotation_from_S(const double S[3][3], double R[3][3])
{
    if (rotation_polar(S, R)) return;
    double N[4][4], V[4][4];
    const double Sxx = S[0][0], Sxy = S[0][1], Sxz = S[0][2];
    const double Syx = S[1][0], Syy = S[1][1], Syz = S[1][2];
    const double Szx = S[2][0], Szy = S[2][1], Szz = S[2][2];
    N[0][0] = Sxx + Syy + Szz; N[0][1] = Syz - Szy;        N[0][2] = Szx - Sxz;        N[0][3] = Sxy - Syx;
    N[1][1] = Sxx - Syy - Szz; N[1][2] = Sxy + Syx;        N[1][3] = Szx + Sxz;
    N[2][2] = -Sxx + Syy - Szz; N[2][3] = Syz + Szy;
    N[3][3] = -Sxx - Syy + Szz;
    // every loop over matrix indices is unrolled: the 4x4 arrays must stay in registers (a dynamically
    // indexed private array lives in scratch memory, ~0.5 us per access on one lane)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < i; ++j) N[i][j] = N[j][i];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 32; ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j) off += N[i][j] * N[i][j];
        if (off < 1e-300) break;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                if (N[p][q] != 0.0) {
                    const double theta = (N[q][q] - N[p][p]) / (2.0 * N[p][q]);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const double a = N[k][p], b = N[k][q]; N[k][p] = c * a - s * b; N[k][q] = s * a + c * b; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const double a = N[p][k], b = N[q][k]; N[p][k] = c * a - s * b; N[q][k] = s * a + c * b; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
                }
            }
    }
    double best = N[0][0];
    double w = V[0][0], x = V[1][0], y = V[2][0], z = V[3][0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (N[i][i] > best) { best = N[i][i]; w = V[0][i]; x = V[1][i]; y = V[2][i]; z = V[3][i]; }
    const double n = sqrt(w * w + x * x + y * y + z * z);
    if (n > 0.0) { w /= n; x /= n; y /= n; z /= n; } else { w = 1.0; x = y = z = 0.0; }
    R[0][0] = w * w + x * x - y * y - z * z; R[0][1] = 2 * (x * y - w * z);           R[0][2] = 2 * (x * z + w * y);
    R[1][0] = 2 * (x * y + w * z);           R[1][1] = w * w - x * x + y * y - z * z; R[1][2] = 2 * (y * z - w * x);
    R[2][0] = 2 * (x * z - w * y);           R[2][1] = 2 * (y * z + w * x);           R[2][2] = w * w - x * x - y * y + z * z;
}

// final = T_inc * final, ++iter, pcl::registration::DefaultConvergenceCriteria
__device__ __forceinline__ void apply_increment(IcpState *st, const float *T, double sum_d2, double N, int max_iter,
                                double trans_eps, double fit_eps)
{
    float F[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        float s = 0.f;
        for (int k = 0; k < 4; ++k) s += T[i * 4 + k] * st->final_T[k * 4 + j];
        F[i * 4 + j] = s;
    }
    for (int k = 0; k < 16; ++k) st->final_T[k] = F[k];
    st->iter += 1;
    if (st->iter >= max_iter) { st->done = 1; st->converged = 1; return; }
    const double cos_angle = 0.5 * ((double)T[0] + (double)T[5] + (double)T[10] - 1.0);
    const double tsq = (double)T[3] * T[3] + (double)T[7] * T[7] + (double)T[11] * T[11];
    if (cos_angle >= 1.0 - trans_eps && tsq <= trans_eps) { st->done = 1; st->converged = 1; return; }
    const double mse = sum_d2 / N;
    if (fabs(mse - st->mse_prev) < 1e-12) { st->done = 1; st->converged = 1; return; }
    if (fabs(mse - st->mse_prev) / st->mse_prev < fit_eps) { st->done = 1; st->converged = 1; return; }
    st->mse_prev = mse;
}

// Sum of the per-workgroup partials in a fixed order (deterministic run to run): lane l adds workgroups
// l, l+64, ... (independent loads, all in flight together), then entry k is summed over the lanes in lane
// order.  One wave; sums[NS] and tmp[NS][64] in LDS.
template <int NS>
__device__ __forceinline__ void reduce_partials(const double *partials, int nblocks, double *sums, double (*tmp)[64])
{
    // (the workgroup's waves share the entries: wave w takes k = w, w + waves, ... -- every (entry, lane) sum and the order of the
    //  final additions are those of a single wave, so the result does not depend on the number of waves)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    for (int k = wv; k < NS; k += nw) {
        double s = 0.0;
        for (int b = lane; b < nblocks; b += 64) s += partials[b * NS + k];
        tmp[k][lane] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < NS) {
        double s = 0.0;
        for (int l = 0; l < 64; ++l) s += tmp[threadIdx.x][l];
        sums[threadIdx.x] = s;
    }
    __syncthreads();
}

// mode 0: ICP iteration (convergence bookkeeping); mode 1: one-shot rigid estimate; mode 2: fitness only
// one thread: the sums of an iteration -> increment, convergence bookkeeping
__device__ __forceinline__ void icp_solve_from_sums(IcpState *st, const double *sums, int mode, int max_iter, double trans_eps, double fit_eps)
{
    for (int k = 0; k < kNSum; ++k) st->sums[k] = sums[k];
    const double N = sums[15];
    st->n_corr = (int)N;
    if (mode == 2) { st->fitness = N > 0.0 ? sums[16] / N : (double)FLT_MAX; return; }
    if (N < 3.0) { st->done = 1; st->converged = 0; return; }             // not enough correspondences
    double pm[3], qm[3], S[3][3], R[3][3];
    for (int a = 0; a < 3; ++a) { pm[a] = sums[a * 4 + 3] / N; qm[a] = sums[12 + a] / N; }
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) S[a][b] = sums[a * 4 + b] - N * pm[a] * qm[b];
    rotation_from_S(S, R);
    float T[16];
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) T[a * 4 + b] = (float)R[a][b];
        T[a * 4 + 3] = (float)(qm[a] - (R[a][0] * pm[0] + R[a][1] * pm[1] + R[a][2] * pm[2]));
    }
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
    for (int k = 0; k < 16; ++k) st->inc_T[k] = T[k];
    if (mode == 1) { for (int k = 0; k < 16; ++k) st->final_T[k] = T[k]; return; }
    apply_increment(st, T, sums[16], N, max_iter, trans_eps, fit_eps);
}

__device__ __forceinline__ void icp_solve_kernel_body(IcpState *st, const double *partials, int nblocks, int mode,
                                                       int max_iter, double trans_eps, double fit_eps)
{
    __shared__ double sums[kNSum];
    __shared__ double tmp[kNSum][64];
    if (mode == 0 && st->done) return;
    reduce_partials<kNSum>(partials, nblocks, sums, tmp);
    if (threadIdx.x != 0) return;
    icp_solve_from_sums(st, sums, mode, max_iter, trans_eps, fit_eps);
}

__global__ __launch_bounds__(64) void icp_solve_kernel(IcpState *st, const double *partials, int nblocks, int mode,
                                                       int max_iter, double trans_eps, double fit_eps)
{
    icp_solve_kernel_body(st, partials, nblocks, mode, max_iter, trans_eps, fit_eps);
}

// ---- point-to-plane estimator (BASELINE configs[2]; the reference itself is point-to-point) ---------
constexpr int kNPlane = 29;              // 21 (upper triangle of A^T A) + 6 (A^T b) + sum d2 + count

__device__ __forceinline__ void jacobi_eig3(double (&A)[3][3], double (&V)[3][3])
{
    // loops over matrix indices unrolled (registers, not scratch): see rotation_from_S
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 32; ++sweep) {
        // done when the off-diagonal part is below 1e-17 of the diagonal (a rotation by such an angle changes no bit of either);
        // waiting for it to underflow (off < 1e-300) cost three to four more sweeps of nothing -- 40 % of the normals' time
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        const double dia = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off < 1e-300 || off <= 1e-34 * dia) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] != 0.0) {
                    const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s * b; A[k][q] = s * a + c * b; }
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s * b; A[q][k] = s * a + c * b; }
#pragma unroll
                    for (int k = 0; k < 3; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
                }
            }
    }
}

// target normals: PCA of the points within `radius` (searched through the NN grid).  Eight lanes per point: the (z, y) rows of
// the ball's bounding box are dealt round robin to the lanes (one thread per point left 1.5 waves per SIMD chasing dependent
// loads row after row), the ten fp64 sums meet by DPP exchanges in a fixed order.
constexpr int kNormGroup = 8;
constexpr int kNormBlock = 512;           // threads per workgroup: 64 points, whose eigenvectors fill ONE wave (256 threads: half a wave)
__device__ __forceinline__ double group_sum8(double v)
{
    // xor 1, xor 2 (quad permutes), then mirror inside each half row (lane i <-> 7 - i): all 8 lanes end with the same sum
#define SCL_GSUM_STEP(CTRL)                                                                                  \
    {                                                                                                        \
        const long long b = __double_as_longlong(v);                                                         \
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xf, 0xf, false);       \
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);                \
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);                                 \
    }
    SCL_GSUM_STEP(0xB1)
    SCL_GSUM_STEP(0x4E)
    SCL_GSUM_STEP(0x141)
#undef SCL_GSUM_STEP
    return v;
}

__device__ __forceinline__ void normals_body(int n_tgt, const IcpState *st, const int *cell_start, const float4 *sorted, double radius, float4 *normals)
{
    if (n_tgt <= 0) return;
    const int gid = (blockIdx.x * blockDim.x + threadIdx.x) / kNormGroup;
    const int sub = threadIdx.x & (kNormGroup - 1);
    const bool valid = gid < n_tgt;
    // the points are taken in the grid's order (sorted[]: coordinates and, in .w, the point's index in the cloud): the groups of a
    // workgroup then walk the same rows of cells, whatever order the cloud came in
    const float4 me = sorted[valid ? gid : n_tgt - 1];        // surplus groups shadow the last point (all lanes stay in the exchanges)
    const int i = __float_as_int(me.w);
    const float3 p = make_float3(me.x, me.y, me.z);
    const float pv[3] = {p.x, p.y, p.z};
    // Cells the ball can reach, per axis: a point within `radius` of p lies in [p - radius, p + radius]; the grid bins points
    // with an fp32 division (cell_index), which can be off by ~1e-4 of a cell for grids of a thousand cells per axis -- the
    // interval is widened by a hundredth of a cell, not by a whole cell on either side (that box held 18 x the ball's volume).
    // (a product with 1 / h where a quotient stood: the two differ by 1e-14 of a cell, the slack is 1e-2; every point of the cells
    //  looked at is tested against the radius in fp64 below, so a cell too many changes nothing)
    const double hh = (double)st->h, inv_hh = 1.0 / hh, kCellSlack = 0.01;
    auto cells = [&](double v, double reach, int a, int &c0, int &c1) {
        const double l = floor((v - reach - (double)st->mn[a]) * inv_hh - kCellSlack);
        const double h = floor((v + reach - (double)st->mn[a]) * inv_hh + kCellSlack);
        c0 = !(l >= 0.0) ? 0 : (l >= (double)st->dim[a] ? st->dim[a] : (int)l);
        c1 = !(h < (double)st->dim[a]) ? st->dim[a] - 1 : (h < 0.0 ? -1 : (int)h);
    };
    int lo[3], hi[3];
    for (int a = 1; a < 3; ++a) cells((double)pv[a], radius, a, lo[a], hi[a]);
    const double r2 = radius * radius;
    double sum[3] = {0, 0, 0}, sq[6] = {0, 0, 0, 0, 0, 0};
    int cnt = 0;
    const int ny = hi[1] - lo[1] + 1, nrows = (hi[2] >= lo[2] && ny > 0) ? (hi[2] - lo[2] + 1) * ny : 0;
    // Rows are taken eight at a time: lane s of the group finds the range of `sorted` that row t0 + s contributes (its cells along x that
    // the ball reaches), then ALL eight lanes share the points of each of those ranges, lane s taking points s, s + 8, ... -- with a
    // whole row per lane the nine rows of a typical ball left one lane with two rows and every lane with its own row's density, and a
    // wave runs as long as its busiest lane.  A point is first placed by its fp32 squared distance; only when that lies within 1e-5 of
    // r^2 does the fp64 test decide (the same verdict for every point as the fp64 test alone: fp32 errs by 1e-6 at most here).
    const float r2f = (float)r2, r2_lo = r2f * (1.0f - 1e-5f), r2_hi = r2f * (1.0f + 1e-5f);
    const int gbase = (int)(threadIdx.x & 63u) & ~(kNormGroup - 1);
    for (int t0 = 0; t0 < nrows; t0 += kNormGroup) {
        int kb = 0, ke = 0;
        const int t = t0 + sub;
        if (t < nrows) {
            const int zz = t / ny;
            const int z = lo[2] + zz, y = lo[1] + (t - zz * ny);
            // the row's cells along x that the ball reaches: distance from p to the row's (y, z) box, taken a hair smaller
            const double ylo = (double)st->mn[1] + (double)y * hh, zlo = (double)st->mn[2] + (double)z * hh;
            const double ddy = fmax(fmax(ylo - (double)p.y, (double)p.y - (ylo + hh)), 0.0) - kCellSlack * hh;
            const double ddz = fmax(fmax(zlo - (double)p.z, (double)p.z - (zlo + hh)), 0.0) - kCellSlack * hh;
            const double dyz2 = (ddy > 0.0 ? ddy * ddy : 0.0) + (ddz > 0.0 ? ddz * ddz : 0.0);
            if (!(dyz2 > r2)) {
                int xa, xb;
                cells((double)p.x, (double)(__builtin_amdgcn_sqrtf((float)(r2 - dyz2)) * 1.0001f), 0, xa, xb);   // (an fp32 root taken a hair longer: a superset)
                if (xa <= xb) {
                    kb = cell_start[(z * st->dim[1] + y) * st->dim[0] + xa];
                    ke = cell_start[(z * st->dim[1] + y) * st->dim[0] + xb + 1];          // cells along x are contiguous
                }
            }
        }
#pragma unroll 1
        for (int j = 0; j < kNormGroup; ++j) {
            const int rb = __shfl(kb, gbase + j, kWave), re = __shfl(ke, gbase + j, kWave);
            for (int k = rb + sub; k < re; k += 4 * kNormGroup) {         // four loads in flight per step
                float4 qq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int kk = k + u * kNormGroup; qq[u] = sorted[kk < re ? kk : re - 1]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 q = qq[u];
                    if (k + u * kNormGroup >= re) continue;
                    const float fx = q.x - p.x, fy = q.y - p.y, fz = q.z - p.z;
                    const float d2f = fx * fx + fy * fy + fz * fz;
                    if (d2f > r2_hi) continue;
                    const double dx = (double)q.x - (double)p.x, dy = (double)q.y - (double)p.y, dz = (double)q.z - (double)p.z;
                    if (!(d2f < r2_lo) && dx * dx + dy * dy + dz * dz > r2) continue;
                    sum[0] += dx; sum[1] += dy; sum[2] += dz;
                    sq[0] += dx * dx; sq[1] += dx * dy; sq[2] += dx * dz; sq[3] += dy * dy; sq[4] += dy * dz; sq[5] += dz * dz;
                    ++cnt;
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) sum[a] = group_sum8(sum[a]);
#pragma unroll
    for (int a = 0; a < 6; ++a) sq[a] = group_sum8(sq[a]);
    cnt = (int)group_sum8((double)cnt);
    // The eigenvectors: one lane per point, and the workgroup's 64 points side by side in ONE wave -- with the first lane of every
    // group of eight the Jacobi sweeps ran in all four waves at an eighth of their lanes, and they are most of this kernel's
    // instructions.  Same arithmetic per point, so the same normals.
    __shared__ double s_sums[kNormBlock / kNormGroup][10];
    __shared__ int s_idx[kNormBlock / kNormGroup];
    if (sub == 0) {
        double *d = s_sums[threadIdx.x / kNormGroup];
        d[0] = sum[0]; d[1] = sum[1]; d[2] = sum[2];
#pragma unroll
        for (int a = 0; a < 6; ++a) d[3 + a] = sq[a];
        d[9] = (double)cnt;
        s_idx[threadIdx.x / kNormGroup] = valid ? i : -1;
    }
    __syncthreads();
    if (threadIdx.x >= kNormBlock / kNormGroup) return;
    {
        const double *d = s_sums[threadIdx.x];
        sum[0] = d[0]; sum[1] = d[1]; sum[2] = d[2];
#pragma unroll
        for (int a = 0; a < 6; ++a) sq[a] = d[3 + a];
        cnt = (int)d[9];
    }
    const int out_i = s_idx[threadIdx.x];
    if (out_i < 0) return;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cnt >= 3) {
        const double N = (double)cnt, m0 = sum[0] / N, m1 = sum[1] / N, m2 = sum[2] / N;
        double C[3][3], V[3][3];
        C[0][0] = sq[0] / N - m0 * m0; C[0][1] = sq[1] / N - m0 * m1; C[0][2] = sq[2] / N - m0 * m2;
        C[1][1] = sq[3] / N - m1 * m1; C[1][2] = sq[4] / N - m1 * m2; C[2][2] = sq[5] / N - m2 * m2;
        C[1][0] = C[0][1]; C[2][0] = C[0][2]; C[2][1] = C[1][2];
        jacobi_eig3(C, V);
        double low = C[0][0], n0 = V[0][0], n1 = V[1][0], n2 = V[2][0];
#pragma unroll
        for (int a = 1; a < 3; ++a) if (C[a][a] < low) { low = C[a][a]; n0 = V[0][a]; n1 = V[1][a]; n2 = V[2][a]; }
        out = make_float4((float)n0, (float)n1, (float)n2, 0.f);
    }
    normals[out_i] = out;
}
__global__ __launch_bounds__(kNormBlock) void normals_kernel(const unsigned char *tgt, int n_tgt, int stride, const IcpState *st,
                                                      const int *cell_start, const float4 *sorted, double radius, float4 *normals)
{
    (void)tgt; (void)stride;
    normals_body(n_tgt, st, cell_start, sorted, radius, normals);
}

__device__ __forceinline__ void plane_reduce_kernel_body(const float4 *work, const unsigned char *tgt_raw, int stride, int n,
                                                           const int *nn_idx, const float *nn_d2, float maxd2,
                                                           const float4 *normals, const IcpState *st, double *partials)
{
    if (st->done) return;
    double acc[kNPlane];
#pragma unroll
    for (int k = 0; k < kNPlane; ++k) acc[k] = 0.0;
    const int r_i0 = (int)(blockIdx.x * blockDim.x), r_step = (int)(gridDim.x * blockDim.x), r_i1 = n;
    for (int i = r_i0 + (int)threadIdx.x; i < r_i1; i += r_step) {
        const int j = nn_idx[i];
        const float d2 = nn_d2[i];
        if (j < 0 || !(d2 <= maxd2)) continue;
        const float4 pw = work[i];
        const float3 q = load_xyz(tgt_raw, j, stride);
        const float4 nn = normals[j];
        const double px = pw.x, py = pw.y, pz = pw.z, nx = nn.x, ny = nn.y, nz = nn.z;
        const double row[6] = {py * nz - pz * ny, pz * nx - px * nz, px * ny - py * nx, nx, ny, nz};
        const double r = ((double)q.x - px) * nx + ((double)q.y - py) * ny + ((double)q.z - pz) * nz;
        int t = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) acc[t++] += row[a] * row[b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += row[a] * r;
        acc[27] += (double)d2;
        acc[28] += 1.0;
    }
    __shared__ double s[4][kNPlane];
#pragma unroll
    for (int k = 0; k < kNPlane; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, kWave);
    }
    const int wv = threadIdx.x / kWave;
    if ((threadIdx.x & 63) == 0) for (int k = 0; k < kNPlane; ++k) s[wv][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < kNPlane)
        partials[blockIdx.x * kNPlane + threadIdx.x] = ((s[0][threadIdx.x] + s[1][threadIdx.x]) + s[2][threadIdx.x]) + s[3][threadIdx.x];
}

__device__ __forceinline__ void plane_solve_from_sums(IcpState *st, const double *sums, int max_iter, double trans_eps, double fit_eps)
{
    const double N = sums[28];
    st->n_corr = (int)N;
    if (N < 3.0) { st->done = 1; st->converged = 0; return; }
    double A[6][6], b[6], x[6];
    int t = 0;
    for (int a = 0; a < 6; ++a) for (int c = a; c < 6; ++c) { A[a][c] = sums[t]; A[c][a] = sums[t]; ++t; }
    for (int a = 0; a < 6; ++a) b[a] = sums[21 + a];
    for (int c = 0; c < 6; ++c) {                                      // Gaussian elimination, partial pivoting
        int piv = c; double best = fabs(A[c][c]);
        for (int r = c + 1; r < 6; ++r) if (fabs(A[r][c]) > best) { best = fabs(A[r][c]); piv = r; }
        if (!(best > 1e-300)) { st->done = 1; st->converged = 0; return; }
        if (piv != c) { for (int k = 0; k < 6; ++k) { const double tmp = A[c][k]; A[c][k] = A[piv][k]; A[piv][k] = tmp; } const double tmp = b[c]; b[c] = b[piv]; b[piv] = tmp; }
        for (int r = c + 1; r < 6; ++r) {
            const double f = A[r][c] / A[c][c];
            for (int k = c; k < 6; ++k) A[r][k] -= f * A[c][k];
            b[r] -= f * b[c];
        }
    }
    for (int r = 5; r >= 0; --r) { double sacc = b[r]; for (int k = r + 1; k < 6; ++k) sacc -= A[r][k] * x[k]; x[r] = sacc / A[r][r]; }
    const double ca = cos(x[0]), sa = sin(x[0]), cb = cos(x[1]), sb = sin(x[1]), cg = cos(x[2]), sg = sin(x[2]);
    float T[16];
    T[0] = (float)(cg * cb); T[1] = (float)(-sg * ca + cg * sb * sa); T[2] = (float)(sg * sa + cg * sb * ca);  T[3] = (float)x[3];
    T[4] = (float)(sg * cb); T[5] = (float)(cg * ca + sg * sb * sa);  T[6] = (float)(-cg * sa + sg * sb * ca); T[7] = (float)x[4];
    T[8] = (float)(-sb);     T[9] = (float)(cb * sa);                 T[10] = (float)(cb * ca);                T[11] = (float)x[5];
    T[12] = T[13] = T[14] = 0.f; T[15] = 1.f;
    for (int k = 0; k < 16; ++k) st->inc_T[k] = T[k];
    apply_increment(st, T, sums[27], N, max_iter, trans_eps, fit_eps);
}

__device__ __forceinline__ void plane_solve_kernel_body(IcpState *st, const double *partials, int nblocks, int max_iter, double trans_eps, double fit_eps)
{
    __shared__ double sums[kNPlane];
    __shared__ double tmp[kNPlane][64];
    if (st->done) return;
    reduce_partials<kNPlane>(partials, nblocks, sums, tmp);
    if (threadIdx.x != 0) return;
    plane_solve_from_sums(st, sums, max_iter, trans_eps, fit_eps);
}

// ---- K6 ----------------------------------------------------------------------------
__global__ void work_init_kernel(const unsigned char *src, int n, int stride, float4 *work)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float3 p = load_xyz(src, i, stride); work[i] = make_float4(p.x, p.y, p.z, 0.f); }
}

__global__ void raw_transform_kernel(const unsigned char *in, unsigned char *out, int n, int stride, const float *T)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char *pi = in + (size_t)i * stride;
    unsigned char *po = out + (size_t)i * stride;
    const float *f = reinterpret_cast<const float *>(pi);
    const float x = f[0], y = f[1], z = f[2];
    float *o = reinterpret_cast<float *>(po);
    for (int k = 3; k < stride / 4; ++k) o[k] = f[k];                        // intensity & padding copied (DM.h:250)
    o[0] = T[0] * x + T[1] * y + T[2] * z + T[3];
    o[1] = T[4] * x + T[5] * y + T[6] * z + T[7];
    o[2] = T[8] * x + T[9] * y + T[10] * z + T[11];
}

__device__ __forceinline__ void state_init_body(IcpState *st)
{
    if (threadIdx.x == 0) {
        for (int k = 0; k < 16; ++k) { st->final_T[k] = (k % 5 == 0) ? 1.f : 0.f; st->inc_T[k] = (k % 5 == 0) ? 1.f : 0.f; }
        st->mse_prev = DBL_MAX; st->fitness = (double)FLT_MAX;
        st->iter = 0; st->done = 0; st->converged = 0; st->n_corr = 0; st->pad_[0] = 0u; st->pad_[1] = 0u;
    }
}

__global__ void state_init_kernel(IcpState *st) { state_init_body(st); }

// ---- everything an alignment needs from its TARGET, for the candidates of one scan at once (blockIdx.y = candidate) ------------
// The search grid of a target is seven short launches (box, set-up, zero, count, scan, scatter, state) and its normals an eighth;
// for 25 candidates on eight lanes that was 1.6 ms of a query's preparation.  Here every step is one launch over all candidates.
// The prefix sum of a target's cell counts -- only the cells+1 entries its grid has, not the table's capacity -- is one
// workgroup per target (4 096 entries per trip).
struct GridJob { const unsigned char *tgt; int n; IcpState *st; int *cstart; int *cfill; float4 *tsort; float *bbox; float4 *normals; };

__global__ void bbox_partial_batch_kernel(const GridJob *jobs, int stride) { const GridJob j = jobs[blockIdx.y]; bbox_partial_body(j.tgt, j.n, stride, j.bbox); }
__global__ void grid_setup_batch_kernel(const GridJob *jobs, int nblocks) { const GridJob j = jobs[blockIdx.x]; grid_setup_body(j.bbox, nblocks, j.n, j.st, j.cstart); }
__global__ void grid_zero_batch_kernel(const GridJob *jobs)
{
    const GridJob j = jobs[blockIdx.y];
    const int n = j.st->cells + 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) j.cstart[i] = 0;
}
__global__ void grid_count_batch_kernel(const GridJob *jobs, int stride) { const GridJob j = jobs[blockIdx.y]; grid_count_body(j.tgt, j.n, stride, j.st, j.cstart); }
__global__ __launch_bounds__(1024) void grid_scan_batch_kernel(const GridJob *jobs)
{
    // counts sit in cstart[cell + 1]; the inclusive sum of entries 0 .. cells turns them into the cells' start offsets; cfill = a copy
    const GridJob j = jobs[blockIdx.x];
    const int n = j.st->cells + 1;
    __shared__ int s_wave[16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int carry = 0;
    for (int base = 0; base < n; base += 4096) {
        const int i0 = base + 4 * t;
        int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = i0 + u < n ? j.cstart[i0 + u] : 0;
        const int mine = (v[0] + v[1]) + (v[2] + v[3]);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, kWave); if (lane >= off) incl += o; }
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        int before = carry, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int x = s_wave[w]; before += w < wv ? x : 0; total += x; }
        int run = before + incl - mine;
#pragma unroll
        for (int u = 0; u < 4; ++u) { run += v[u]; if (i0 + u < n) { j.cstart[i0 + u] = run; j.cfill[i0 + u] = run; } }
        carry += total;
        __syncthreads();
    }
}
__global__ void grid_scatter_batch_kernel(const GridJob *jobs, int stride) { const GridJob j = jobs[blockIdx.y]; grid_scatter_body(j.tgt, j.n, stride, j.st, j.cfill, j.tsort); }
__global__ void state_init_batch_kernel(const GridJob *jobs) { state_init_body(jobs[blockIdx.x].st); }

// ---- RANSAC correspondence rejection (CorrespondenceRejectorSampleConsensus, DM.h:1218-1225) ----------
// Deterministic restatement (see oracle/icp_oracle.h): hypothesis h = 3 distinct correspondences drawn
// with splitmix64(seed, h, draw), rigid fit of the 3 pairs in fp64, every hypothesis scored against every
// correspondence.  kHypPerBlock hypotheses share one sweep over the pairs.
constexpr int kHypPerBlock = 8;

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ void ransac_model(const unsigned char *src, const unsigned char *tgt, int stride, const int *si, const int *ti,
                             int n, unsigned long long seed, int h, double T[12])
{
    int idx[3];
    unsigned long long ctr = 0;
    for (int m = 0; m < 3;) {
        const unsigned long long r = splitmix64(seed ^ splitmix64(((unsigned long long)h << 20) + ctr));
        ctr++;
        const int cand = (int)(r % (unsigned long long)n);
        bool dup = false;
        for (int q = 0; q < m; ++q) dup |= idx[q] == cand;
        if (!dup) idx[m++] = cand;
    }
    double pm[3] = {0, 0, 0}, qm[3] = {0, 0, 0}, P[3][3], Q[3][3];
    for (int i = 0; i < 3; ++i) {
        const float3 p = load_xyz(src, si[idx[i]], stride), q = load_xyz(tgt, ti[idx[i]], stride);
        P[i][0] = p.x; P[i][1] = p.y; P[i][2] = p.z; Q[i][0] = q.x; Q[i][1] = q.y; Q[i][2] = q.z;
        for (int a = 0; a < 3; ++a) { pm[a] += P[i][a]; qm[a] += Q[i][a]; }
    }
    for (int a = 0; a < 3; ++a) { pm[a] /= 3.0; qm[a] /= 3.0; }
    double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, R[3][3];
    for (int i = 0; i < 3; ++i) for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
        S[a][b] += (P[i][a] - pm[a]) * (Q[i][b] - qm[b]);
    rotation_from_S(S, R);
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) T[a * 4 + b] = R[a][b];
        T[a * 4 + 3] = qm[a] - (R[a][0] * pm[0] + R[a][1] * pm[1] + R[a][2] * pm[2]);
    }
}

__device__ __forceinline__ bool ransac_inlier(const double *T, float3 p, float3 q, double thr2)
{
    const double x = p.x, y = p.y, z = p.z;
    const double dx = (T[0] * x + T[1] * y + T[2] * z + T[3]) - (double)q.x;
    const double dy = (T[4] * x + T[5] * y + T[6] * z + T[7]) - (double)q.y;
    const double dz = (T[8] * x + T[9] * y + T[10] * z + T[11]) - (double)q.z;
    return (dx * dx + dy * dy) + dz * dz < thr2;
}

__global__ __launch_bounds__(256) void ransac_score_kernel(const unsigned char *src, const unsigned char *tgt, int stride,
                                                           const int *si, const int *ti, int n, unsigned long long seed,
                                                           int n_hyp, double thr2, int *counts)
{
    __shared__ double sT[kHypPerBlock][12];
    __shared__ int scnt[kHypPerBlock];
    const int h0 = blockIdx.x * kHypPerBlock;
    if (threadIdx.x < kHypPerBlock) {
        scnt[threadIdx.x] = 0;
        if (h0 + (int)threadIdx.x < n_hyp) ransac_model(src, tgt, stride, si, ti, n, seed, h0 + threadIdx.x, sT[threadIdx.x]);
        else for (int k = 0; k < 12; ++k) sT[threadIdx.x][k] = 0.0;
    }
    __syncthreads();
    int cnt[kHypPerBlock];
#pragma unroll
    for (int u = 0; u < kHypPerBlock; ++u) cnt[u] = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float3 p = load_xyz(src, si[i], stride), q = load_xyz(tgt, ti[i], stride);
#pragma unroll
        for (int u = 0; u < kHypPerBlock; ++u) cnt[u] += ransac_inlier(sT[u], p, q, thr2) ? 1 : 0;
    }
#pragma unroll
    for (int u = 0; u < kHypPerBlock; ++u) {
        int c = cnt[u];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, kWave);
        if ((threadIdx.x & 63) == 0) atomicAdd(&scnt[u], c);
    }
    __syncthreads();
    if (threadIdx.x < kHypPerBlock && h0 + (int)threadIdx.x < n_hyp) counts[h0 + threadIdx.x] = scnt[threadIdx.x];
}

// best hypothesis (most inliers, ties -> lowest index), its model, and the inlier mask
__global__ void ransac_pick_kernel(const int *counts, int n_hyp, int *best /*[2]: h, count*/)
{
    __shared__ long long sk[16];
    long long key = -1;                                  // (count << 32) | (0x7fffffff - h): max = most inliers, lowest h
    for (int h = threadIdx.x; h < n_hyp; h += blockDim.x) {
        const long long k2 = ((long long)counts[h] << 32) | (long long)(0x7fffffff - h);
        key = k2 > key ? k2 : key;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const long long o = __shfl_xor(key, off, kWave); key = o > key ? o : key; }
    if ((threadIdx.x & 63) == 0) sk[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)blockDim.x / 64; ++w) key = sk[w] > key ? sk[w] : key;
        best[0] = key < 0 ? -1 : 0x7fffffff - (int)(key & 0xffffffffll);
        best[1] = key < 0 ? 0 : (int)(key >> 32);
    }
}

__global__ __launch_bounds__(256) void ransac_mask_kernel(const unsigned char *src, const unsigned char *tgt, int stride,
                                                          const int *si, const int *ti, int n, unsigned long long seed,
                                                          const int *best, double thr2, int *mask, double *T_out)
{
    __shared__ double sT[12];
    if (threadIdx.x == 0) {
        if (best[0] >= 0) ransac_model(src, tgt, stride, si, ti, n, seed, best[0], sT);
        else for (int k = 0; k < 12; ++k) sT[k] = (k % 5 == 0) ? 1.0 : 0.0;
        if (blockIdx.x == 0) for (int k = 0; k < 12; ++k) T_out[k] = sT[k];
    }
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        mask[i] = ransac_inlier(sT, load_xyz(src, si[i], stride), load_xyz(tgt, ti[i], stride), thr2) ? 1 : 0;
}

__global__ void iota_pairs_kernel(const int *nn, int n, int *si, int *ti)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { si[i] = i; ti[i] = nn[i]; }
}

// ---- batched forms: one launch serves the same step of many alignments (blockIdx.y = alignment) -------------------
// The loop of ONE alignment is a chain of three dependent, latency-bound launches per iteration (search 35 us with the
// chip a quarter full, reduction 8 us, a single-wave solve 15 us); the loop candidates of one scan (BASELINE configs[2]:
// 25) are independent, so their chains run side by side inside the same launches.  A finished alignment's workgroups
// leave at once (st->done).
struct IcpProblem {
    float4 *work; IcpState *st; const int *cell_start; const float4 *sorted; int *nni; float *nnd; double *part;
    const unsigned char *tgt; const float4 *normals;
    float4 *nnq;                                             // every source's current neighbour: its coordinates and (in .w) its index
    int *flag;                                               // per workgroup of the tile search: lanes left for icp_tile_finish_kernel
    int n_tgt;
};

// How far a part of a batch has got, for the host that enqueues its iterations (icp_batch_run): the last alignment to leave a solve
// launch writes (launches of the part finished so far << 32 | alignments of the part that are done) into pinned host memory.
struct PartSync {
    int *counters;                       // device: [0] arrivals at the current launch's end, [1] alignments done
    unsigned long long *host_word;       // pinned; nullptr: nobody is watching
    unsigned int seq;                    // this launch's number within the part (the cold iteration's solve is 1)
};
__device__ __forceinline__ void part_sync_arrive(const PartSync &ps, const bool newly_done)
{
    if (newly_done) atomicAdd(&ps.counters[1], 1);
    __threadfence();
    if (atomicAdd(&ps.counters[0], 1) == (int)gridDim.x - 1) {   // every alignment of the launch has been here: the next launch starts behind this one
        ps.counters[0] = 0;
        const unsigned int nd = (unsigned int)atomicAdd(&ps.counters[1], 0);
        __hip_atomic_store(ps.host_word, ((unsigned long long)ps.seq << 32) | (unsigned long long)nd, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void icp_solve_batch_kernel(const IcpProblem *pr, int nblocks, int mode, int max_iter, double trans_eps, double fit_eps, PartSync ps)
{
    const IcpProblem p = pr[blockIdx.x];
    const bool was_done = ps.host_word && threadIdx.x == 0 && p.st->done != 0;
    icp_solve_kernel_body(p.st, p.part, nblocks, mode, max_iter, trans_eps, fit_eps);
    if (ps.host_word && threadIdx.x == 0) part_sync_arrive(ps, !was_done && p.st->done != 0);
}

__global__ __launch_bounds__(256) void plane_reduce_batch_kernel(const IcpProblem *pr, int stride, int n, float maxd2)
{
    const IcpProblem p = pr[blockIdx.y];
    plane_reduce_kernel_body(p.work, p.tgt, stride, n, p.nni, p.nnd, maxd2, p.normals, p.st, p.part);
}

__global__ __launch_bounds__(256) void plane_solve_batch_kernel(const IcpProblem *pr, int nblocks, int max_iter, double trans_eps, double fit_eps, PartSync ps)
{
    const IcpProblem p = pr[blockIdx.x];
    const bool was_done = ps.host_word && threadIdx.x == 0 && p.st->done != 0;
    plane_solve_kernel_body(p.st, p.part, nblocks, max_iter, trans_eps, fit_eps);
    if (ps.host_word && threadIdx.x == 0) part_sync_arrive(ps, !was_done && p.st->done != 0);
}

// the normals of the targets of a batch's alignments (icp_batch_run: beside the cold searches)
__global__ __launch_bounds__(kNormBlock) void normals_problems_kernel(const IcpProblem *pr, double radius)
{
    const IcpProblem j = pr[blockIdx.y];
    if ((long long)blockIdx.x * (kNormBlock / kNormGroup) >= j.n_tgt) return;  // (a workgroup is 64 points; the grid is sized for the largest target)
    normals_body(j.n_tgt, j.st, j.cell_start, j.sorted, radius, const_cast<float4 *>(j.normals));
}

// ---- K4c: the search of a loop iteration served from LDS ---------------------------------------------------------------------
// The one-lane-per-query walk above spends its time waiting: previous neighbour -> cell table -> points are dependent round trips
// to L2 / HBM for every query, 8.6 G queries/s however the lanes are grouped.  Here a workgroup takes kTileQ consecutive source
// points -- the batch's sources are sorted along a Hilbert curve once per scan (source_order below), so they are a compact blob
// --, forms the bounding box of the cells its queries' balls reach, stages that box's slice of the cell table and its points
// in LDS with two rounds of coalesced loads, and every query then walks ITS cells out of LDS.  The arithmetic of a comparison, the
// pruning margins and the (distance, index) order are those of nn_core: the neighbour found is the same point, bit for bit.
// A round whose box does not fit (kTileRows rows, kTileTab table entries, kTilePts points) is served by nn_core<1> from
// memory for the lanes that asked -- a blob torn by a jump of the curve, sources far away from the target.
// Cold (first iteration, nothing known): round A looks at each query's own cell, round B at the 3 x 3 x 3 cells around it for
// the queries whose cell was empty, what is still without a candidate walks shells in memory (nn_core<1>); then the ball.
// do_reduce: the workgroup's correspondences are reduced on the spot (fp64 matrix cores, as corr_reduce_mfma_kernel_body) and
// leave as one record of kNSum sums per workgroup: the iteration is this launch + the solve.
#ifndef SCL_TILE_PTS
#define SCL_TILE_PTS 1536
#endif
#ifndef SCL_TILE_TAB
#define SCL_TILE_TAB 1024
#endif
#ifndef SCL_TILE_ABLATE
#define SCL_TILE_ABLATE 0                // experiments only (results invalid): 1 no walk, 2 no reduction, 4 no staging of the points, 8 no rounds at all
#endif
#ifndef SCL_TILE_Q
#define SCL_TILE_Q 256
#endif
#ifndef SCL_TILE_ROWS
#define SCL_TILE_ROWS 128
#endif
constexpr int kTileQ = SCL_TILE_Q, kTilePts = SCL_TILE_PTS, kTileTab = SCL_TILE_TAB, kTileRows = SCL_TILE_ROWS;
#ifdef SCL_DIAGNOSTICS
// [0..7] (SCL_DIAGNOSTICS=1): rounds asked for, rounds that did not fit, lanes finished in memory, table entries / points staged, row
// visits, points compared, tiles left to the finish launch; [8..14] (SCL_DIAGNOSTICS=2, so that the counters' atomics do not sit in the phases they time): ticks of
// thread 0 of every workgroup between the kernel's phases
__device__ unsigned long long g_tile_stats[16];
#endif
#if defined(SCL_DIAGNOSTICS) && SCL_DIAGNOSTICS == 1
#define TILE_STAT(k, v) atomicAdd(&g_tile_stats[k], (unsigned long long)(v))
#else
#define TILE_STAT(k, v) ((void)0)
#endif
#if defined(SCL_DIAGNOSTICS) && SCL_DIAGNOSTICS == 2
#define TILE_STAMP(k) do { if (threadIdx.x == 0) { const unsigned long long now__ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_tile_stats[k], now__ - tile_t0__); tile_t0__ = now__; } } while (0)
#define TILE_STAMP_DECL unsigned long long tile_t0__ = __builtin_amdgcn_s_memrealtime()
#define TILE_STAMP_ARG , unsigned long long &tile_t0__
#define TILE_STAMP_PASS , tile_t0__
#else
#define TILE_STAMP(k) ((void)0)
#define TILE_STAMP_DECL ((void)0)
#define TILE_STAMP_ARG
#define TILE_STAMP_PASS
#endif

struct TileLds {
    float4 pts[kTilePts + 4];            // staged target points (x, y, z, index; a walk reads up to 3 past its range); reused as the reduction's staging area
    int tab[kTileTab];                   // the box's slice of cell_start, row after row, as offsets into pts
    int delta[kTileRows];                // per row of the box: (where its points start in pts) - (where they start in `sorted`)
    int loff[kTileRows + 1];             // where the row's points start in pts; loff[nrows] = their number
    int box[2][6];                       // the requests' bounding box, two rounds' worth (a round prepares the next one's)
    unsigned char prow[kTilePts];        // the row of every staged point
    double red[kTileQ / 64][kNSum];
};
static_assert(kTileRows <= 256 && kTileRows % 64 == 0 && kTileQ % 64 == 0 && kTilePts % kTileQ == 0 && kTileTab % kTileQ == 0, "tile constants");

struct TileGrid { float gx0, gy0, gz0, h, inv_h; int dx, dy, dz; };

// minimum over the wave's 64 lanes, the same value in every lane (a scalar): four DPP steps inside the rows of 16, then the four rows
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));    // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));    // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// n / d and n % d for 0 <= n < 2^20, 0 < d < 2^12 through the reciprocal inv = 1.0f / d (an integer division by a variable is some
// forty instructions; the staging loops did two per table entry): the float quotient is off by at most one, which the remainder tells
__device__ __forceinline__ void fast_divmod(const int n, const int d, const float inv, int &q, int &r)
{
    q = (int)((float)n * inv);
    r = n - q * d;
    if (r < 0) { --q; r += d; }
    else if (r >= d) { ++q; r -= d; }
}

// the cells [a, b] of one axis that the interval [v - reach, v + reach] meets, clamped to the grid (empty: a > b).  A product with 1 / h
// where nn_core divides: the reach is widened by what the two can differ by (2e-5 cells at a hundred cells per axis)
__device__ __forceinline__ void tile_reach(float v, float g0, float h, float inv_h, int dim, float reach, int &a, int &b)
{
    reach += 2.1e-5f * h;
    const float fa = floorf((v - reach - g0) * inv_h), fb = floorf((v + reach - g0) * inv_h);
    a = 0; b = dim - 1;
    if (fa == fa && fb == fb) {
        const int ia = fa < -1.0e9f ? 0 : (fa > 1.0e9f ? dim : (int)fa), ib = fb < -1.0e9f ? -1 : (fb > 1.0e9f ? dim - 1 : (int)fb);
        a = max(a, ia); b = min(b, ib);
    }
}

// LDS operations of one wave execute in issue order; only the compiler must not move them across this point
__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Staging: the lanes with req ask for the cells [xa, xb] x [ya, yb] x [za, zb] (non-empty ranges inside the grid); the box of all
// requests -- its slice of the cell table as offsets into pts, its points -- is put in LDS.  Returns 0 when nobody asked, 1 when
// the box is staged (B describes it), 2 when it does not fit (nothing usable in LDS).  The same value in every lane of the workgroup.
struct TileBox { int X0, Y0, Z0, NY, W1; };
__device__ __forceinline__ int tile_stage(TileLds &L, const TileGrid &g, const int *cell_start, const float4 *sorted, const bool req,
                                          const int xa, const int xb, const int ya, const int yb, const int za, const int zb, TileBox &B, int &par TILE_STAMP_ARG)
{
    // Barriers of a round: the box is complete | the table is in LDS | every point knows its row | the points are in LDS.  (The box
    // of round n + 1 is initialised by round n in the other of two buffers; the rows' prefix sum is done by every wave for itself --
    // the same values from every wave -- and the table keeps its raw entries, a per-row delta turning them into offsets at use: each of
    // the three took a barrier of its own.)  L.box[par] holds INT_MAX / INT_MIN on entry.
    const int t = threadIdx.x, lane = t & 63;
    int lo[3] = {req ? xa : INT_MAX, req ? ya : INT_MAX, req ? za : INT_MAX}, hi[3] = {req ? xb : -INT_MAX, req ? yb : -INT_MAX, req ? zb : -INT_MAX};
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = wave_min_i32(lo[a]); hi[a] = -wave_min_i32(-hi[a]); }
    int *box = L.box[par];
    if (lane == 0 && lo[0] <= hi[0]) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(&box[a], lo[a]); atomicMax(&box[3 + a], hi[a]); }
    }
    __syncthreads();                                             // (also: every lane has left the previous round's table and points)
    const int X0 = box[0], Y0 = box[1], Z0 = box[2], X1 = box[3], Y1 = box[4], Z1 = box[5];
    par ^= 1;
    if (t < 6) L.box[par][t] = t < 3 ? INT_MAX : INT_MIN;        // the next round's (nobody has touched this buffer since the round before last)
    TILE_STAMP(9);
    if (X0 > X1) return 0;
    if (t == 0) TILE_STAT(0, 1);
    const int W1 = X1 - X0 + 2, NY = Y1 - Y0 + 1, NZ = Z1 - Z0 + 1;    // W1 table entries per row: its cells' starts + the end of the last
    if ((long long)NY * NZ > kTileRows || (long long)NY * NZ * W1 > kTileTab) { if (t == 0) TILE_STAT(1, 1); return 2; }
    const int nrows = NY * NZ, nent = nrows * W1;
    const float inv_W1 = 1.0f / (float)W1, inv_NY = 1.0f / (float)NY;
    {                                                            // every load of the table in flight before the first is stored
        int tv[kTileTab / kTileQ];
#pragma unroll
        for (int u = 0; u < kTileTab / kTileQ; ++u) {
            const int e = t + u * kTileQ;
            int r, k, zz, yy;
            fast_divmod(e, W1, inv_W1, r, k);
            fast_divmod(r, NY, inv_NY, zz, yy);
            tv[u] = e < nent ? cell_start[((Z0 + zz) * g.dy + (Y0 + yy)) * g.dx + X0 + k] : 0;
        }
#pragma unroll
        for (int u = 0; u < kTileTab / kTileQ; ++u) { const int e = t + u * kTileQ; if (e < nent) L.tab[e] = tv[u]; }
    }
    __syncthreads();
    TILE_STAMP(10);
    {                                                            // the rows' point counts and their exclusive scan, by every wave: two rows per lane
        constexpr int PER = kTileRows / 64;
        int cnt[PER], first[PER], tot = 0;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int r = lane * PER + u;
            first[u] = r < nrows ? L.tab[r * W1] : 0;
            cnt[u] = r < nrows ? L.tab[r * W1 + W1 - 1] - first[u] : 0;
            tot += cnt[u];
        }
        int incl = tot;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, kWave); if (lane >= off) incl += o; }
        int base = incl - tot;
#pragma unroll
        for (int u = 0; u < PER; ++u) { const int r = lane * PER + u; if (r < nrows) { L.loff[r] = base; L.delta[r] = base - first[u]; } base += cnt[u]; }
        if (lane == 63) L.loff[nrows] = incl;
    }
    wave_fence();                                                // (this wave's own writes: the other waves write the same values)
    const int total = L.loff[nrows];
    TILE_STAMP(11);
    if (total > kTilePts) { if (t == 0) TILE_STAT(1, 1); return 2; }
    if (t == 0) { TILE_STAT(3, nent); TILE_STAT(4, total); }
    for (int r = t >> 4; r < nrows; r += kTileQ / 16) {          // which row every staged point belongs to (16 lanes per row, LDS only)
        const int dst0 = L.loff[r], cnt = L.loff[r + 1] - dst0;
        for (int k = t & 15; k < cnt; k += 16) L.prow[dst0 + k] = (unsigned char)r;
    }
    __syncthreads();
#pragma unroll
    for (int u0 = 0; u0 < ((SCL_TILE_ABLATE & 4) ? 0 : kTilePts / kTileQ); u0 += 4) {   // the points: four loads per lane in flight, then their stores
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = t + (u0 + u) * kTileQ;
            if (k < total) v[u] = sorted[k - L.delta[L.prow[k]]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int k = t + (u0 + u) * kTileQ; if (k < total) L.pts[k] = v[u]; }
    }
    __syncthreads();
    TILE_STAMP(12);
    B.X0 = X0; B.Y0 = Y0; B.Z0 = Z0; B.NY = NY; B.W1 = W1;
    return 1;
}

// One round of the neighbour search: staging, then every asking lane walks ITS cells out of LDS (best / bi / bq updated when 1 is
// returned; 0 and 2 as tile_stage).
__device__ __forceinline__ int tile_round(TileLds &L, const TileGrid &g, const int *cell_start, const float4 *sorted, const bool req,
                                          const int xa, const int xb, const int ya, const int yb, const int za, const int zb,
                                          const float3 p, float &best, int &bi, float3 &bq, int &par TILE_STAMP_ARG)
{
    TileBox B;
    const int rc = tile_stage(L, g, cell_start, sorted, req, xa, xb, ya, yb, za, zb, B, par TILE_STAMP_PASS);
    if (rc != 1) return rc;
    const int X0 = B.X0, Y0 = B.Y0, Z0 = B.Z0, NY = B.NY, W1 = B.W1;
    if (req && !(SCL_TILE_ABLATE & 1)) {
        // (distance, index) as one 64-bit key: the bits of a distance >= 0 order like the distance, the index's sign bit is flipped so
        // that -1 ("none") comes last among equal distances, as in nn_core; a NaN distance has the largest key and is never taken
        unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | ((unsigned int)bi ^ 0x80000000u);
        int kbest = -1;
        for (int z = za; z <= zb; ++z)
            for (int y = ya; y <= yb; ++y) {
                // nn_core's row test: the distance from p to the row's (y, z) box, the box taken a hair smaller than the cells
                const float ylo = g.gy0 + (float)y * g.h, zlo = g.gz0 + (float)z * g.h;
                const float ddy = fmaxf(fmaxf(ylo - p.y, p.y - (ylo + g.h)), 0.f), ddz = fmaxf(fmaxf(zlo - p.z, p.z - (zlo + g.h)), 0.f);
                const float dyz2 = (ddy * ddy + ddz * ddz) * 0.9995f;
                const float bnow = __uint_as_float((unsigned int)(key >> 32));
                if (dyz2 > bnow) continue;
                int x0 = xa, x1 = xb;
                if (bnow < FLT_MAX) {
                    // the cells of the row the ball reaches: a product with 1 / h instead of nn_core's division, the reach widened by
                    // what the two can differ by (2e-5 cells at 100 cells per axis; the 1.0005 covers the square root's last bit)
                    const float reach = __builtin_amdgcn_sqrtf(bnow - dyz2) * 1.0005f + 2.1e-5f * g.h;
                    const float fa = floorf((p.x - reach - g.gx0) * g.inv_h), fb = floorf((p.x + reach - g.gx0) * g.inv_h);
                    if (fa == fa && fb == fb) {
                        const int ia = fa < -1.0e9f ? 0 : (fa > 1.0e9f ? g.dx : (int)fa), ib = fb < -1.0e9f ? -1 : (fb > 1.0e9f ? g.dx - 1 : (int)fb);
                        x0 = max(x0, ia); x1 = min(x1, ib);
                    }
                }
                if (x0 > x1) continue;
                const int rr = (z - Z0) * NY + (y - Y0), row = rr * W1 - X0, dl = L.delta[rr];
                const int kb = L.tab[row + x0] + dl, ke = L.tab[row + x1 + 1] + dl;
                TILE_STAT(5, 1); TILE_STAT(6, ke - kb);
                for (int k = kb; k < ke; k += 4) {
                    const float4 *qp = &L.pts[k];                // (reads past ke stay inside L: masked below)
                    float4 q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) q[u] = qp[u];
                    const int rem = ke - k;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float ex = p.x - q[u].x, ey = p.y - q[u].y, ez = p.z - q[u].z;
                        const float d = (ex * ex + ey * ey) + ez * ez;
                        const unsigned long long ku = ((unsigned long long)__float_as_uint(d) << 32) | (__float_as_uint(q[u].w) ^ 0x80000000u);
                        const bool take = (u < rem) & (ku < key);
                        key = take ? ku : key;
                        kbest = take ? k + u : kbest;
                    }
                }
            }
        best = __uint_as_float((unsigned int)(key >> 32));
        bi = (int)((unsigned int)key ^ 0x80000000u);
        if (kbest >= 0) { const float4 q = L.pts[kbest]; bq = make_float3(q.x, q.y, q.z); }
    }
    TILE_STAMP(13);
    return 1;
}

// A workgroup's kTileQ correspondences (p, its neighbour q, ok) -> one record of kNSum sums: the operands meet in LDS in the layout
// corr_reduce_mfma_kernel_body reads from memory, four fp64 matrix-core products per wave, the waves' sums added in wave order.
__device__ __forceinline__ void tile_reduce(float *pq_area /* kTileQ x 8 floats of LDS */, double (*red)[kNSum] /* [kTileQ / 64] in LDS */,
                                            const float3 p, const float3 q, const bool ok, const float d2, double *record)
{
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    __syncthreads();                                             // every lane has left the area
    float *pq = pq_area + t * 8;
    const float okf = ok ? 1.0f : 0.0f;
    pq[0] = ok ? p.x : 0.f; pq[1] = ok ? p.y : 0.f; pq[2] = ok ? p.z : 0.f; pq[3] = okf;
    pq[4] = ok ? q.x : 0.f; pq[5] = ok ? q.y : 0.f; pq[6] = ok ? q.z : 0.f; pq[7] = okf;
    double sum_d2 = ok ? (double)d2 : 0.0;
    __syncthreads();
    const int k = lane >> 4, blk = (lane >> 2) & 3, comp = lane & 3;
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int base = 0; base < 64; base += 32) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float *e = pq_area + (wv * 64 + base + half * 16 + 4 * blk + k) * 8;
            const double av = (double)e[comp], bv = (double)e[4 + comp];
            if (half == 0) acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc0, 0, 0, 0);
            else           acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc1, 0, 0, 0);
        }
    }
    double d = acc0 + acc1;                                      // lane 16i + 4b + j holds block b's D[i][j]
    d += __shfl_xor(d, 4, kWave);
    d += __shfl_xor(d, 8, kWave);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum_d2 += __shfl_xor(sum_d2, off, kWave);
    if (blk == 0) red[wv][k * 4 + comp] = d;
    if (lane == 0) red[wv][16] = sum_d2;
    __syncthreads();
    if (t < kNSum) {
        double v = red[0][t];
#pragma unroll
        for (int w = 1; w < kTileQ / 64; ++w) v += red[w][t];        // (the waves' sums in wave order)
        record[t] = v;
    }
}

#ifndef SCL_TILE_WAVES
#define SCL_TILE_WAVES 5                 // built for five waves per SIMD (what 32 KB of LDS per workgroup allow): 95 registers instead of 116, the
                                         // same 80 bytes of scratch; 5.71-5.84 against 5.84-5.91 ms per point-to-point query of 25 x 100 k
#endif
__attribute__((amdgpu_waves_per_eu(SCL_TILE_WAVES, SCL_TILE_WAVES)))
__global__ __launch_bounds__(kTileQ) void icp_tile_search_kernel(const IcpProblem *pr, int n_src, int check_done, int apply, int cold,
                                                                  int stride, float maxd2, int do_reduce)
{
    __shared__ TileLds L;
    const IcpProblem P = pr[blockIdx.y];
    const IcpState *st = P.st;
    const int t = threadIdx.x, wv = t >> 6;
    const int i = blockIdx.x * kTileQ + t;
    const bool valid = i < n_src;
    // Every read the prologue needs is issued here, back to back, from addresses that exist for every lane (the last source stands in
    // for the lanes past the end; nnq holds n_src + 1 entries, whatever is in them before the first search), and masked afterwards: a
    // read inside a branch is a basic block of its own that ends in a wait -- the done flag, the grid, the working point, the
    // increment, the previous neighbour's index and its coordinates were six round trips in a row in the life of every workgroup.
    const int ic = valid ? i : (n_src > 0 ? n_src - 1 : 0);
    float4 pw = P.work[ic];
    const float4 prev = P.nnq[ic];
    const int done = st->done;
    TileGrid g;
    g.gx0 = st->mn[0]; g.gy0 = st->mn[1]; g.gz0 = st->mn[2]; g.h = st->h; g.dx = st->dim[0]; g.dy = st->dim[1]; g.dz = st->dim[2];
    float T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = st->inc_T[k];
    asm volatile("" :: "v"(pw.x), "v"(prev.x), "v"(done), "v"(g.h), "v"(T[0]));   // (all of them have arrived before the first branch)
    if (check_done && done) return;
    g.inv_h = 1.0f / g.h;
    if (!valid) pw = make_float4(0.f, 0.f, 0.f, 0.f);
    else if (apply) {                                            // K6: the previous solve's increment moves the working point first
        const float x = pw.x, y = pw.y, z = pw.z;
        pw.x = T[0] * x + T[1] * y + T[2] * z + T[3];
        pw.y = T[4] * x + T[5] * y + T[6] * z + T[7];
        pw.z = T[8] * x + T[9] * y + T[10] * z + T[11];
        pw.w = 0.f;
        P.work[i] = pw;
    }
    const float3 p = make_float3(pw.x, pw.y, pw.z);
    float best = FLT_MAX;
    int bi = -1;
    float3 bq = make_float3(0.f, 0.f, 0.f);
    int par = 0;                                                 // which of the two boxes the next round fills
    if (t < 6) L.box[0][t] = t < 3 ? INT_MAX : INT_MIN;
    __syncthreads();
    bool open = valid;                                           // this lane's search is not finished
    bool deferred = false;                                       // ... and will be finished in memory by the launch behind this one
    TILE_STAMP_DECL;
    if (valid && !cold) {                                        // the previous neighbour bounds the ball
        const int j = __float_as_int(prev.w);
        if (j >= 0) {
            const float ex = p.x - prev.x, ey = p.y - prev.y, ez = p.z - prev.z;
            const float d = (ex * ex + ey * ey) + ez * ez;
            if (d == d) { best = d; bi = j; bq = make_float3(prev.x, prev.y, prev.z); }
        }
    }
    // Stages: 0 = the query's own cell, 1 = the 27 cells around it for the queries that found nothing (both only when cold), 2 = the
    // ball of the best distance known.  A stage's box that does not fit is tried once more wave by wave (64 consecutive sources: a
    // smaller box); what does not fit then, and the queries that reach stage 2 without a candidate, finish in memory: nn_core's
    // own walk from whatever the lane knows.  (One loop, so that tile_round and nn_core are instantiated once: registers.)
    int c[3] = {0, 0, 0};
    if (cold) {                                                  // cell_index's expressions on the grid already loaded
        const float v[3] = {p.x, p.y, p.z}, mn[3] = {g.gx0, g.gy0, g.gz0};
        const int dim[3] = {g.dx, g.dy, g.dz};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float f = floorf((v[a] - mn[a]) / g.h);
            c[a] = (f != f) ? 0 : (f < 0.f ? 0 : (f >= (float)dim[a] ? dim[a] - 1 : (int)f));
        }
    }
#pragma unroll 1
    for (int stage = (SCL_TILE_ABLATE & 8) ? 3 : (cold ? 0 : 2); stage < 3; ++stage) {
        int xa = 0, xb = -1, ya = 0, yb = -1, za = 0, zb = -1;
        bool ask = false, to_memory = false;
        if (stage == 0) { ask = open; xa = xb = c[0]; ya = yb = c[1]; za = zb = c[2]; }
        else if (stage == 1) {
            ask = open && bi < 0;
            xa = max(c[0] - 1, 0); xb = min(c[0] + 1, g.dx - 1); ya = max(c[1] - 1, 0); yb = min(c[1] + 1, g.dy - 1); za = max(c[2] - 1, 0); zb = min(c[2] + 1, g.dz - 1);
        } else {
            to_memory = open && bi < 0;
            if (open && bi >= 0) {
                const float reach0 = __builtin_amdgcn_sqrtf(best) * 1.0005f + 1e-6f * g.h;
                tile_reach(p.x, g.gx0, g.h, g.inv_h, g.dx, reach0, xa, xb);
                tile_reach(p.y, g.gy0, g.h, g.inv_h, g.dy, reach0, ya, yb);
                tile_reach(p.z, g.gz0, g.h, g.inv_h, g.dz, reach0, za, zb);
                ask = xa <= xb && ya <= yb && za <= zb;
            }
        }
#pragma unroll 1
        for (int sub = -1; sub < kTileQ / 64; ++sub) {            // -1: the whole workgroup; 0..3: wave by wave after a box that did not fit
            const bool mine = ask && (sub < 0 || wv == sub);
            if (stage == 2 && sub < 0) TILE_STAMP(8);            // (everything up to the ball round: loads, seeds, a cold search's stages 0 and 1)
            const int rc = tile_round(L, g, P.cell_start, P.sorted, mine, xa, xb, ya, yb, za, zb, p, best, bi, bq, par TILE_STAMP_PASS);
            if ((rc == 2 && sub >= 0 && mine) || to_memory) {     // left to icp_tile_finish_kernel, with what the lane knows as its seed
                TILE_STAT(2, 1);
                deferred = true; open = false; ask = false; to_memory = false;
            }
            if (sub < 0 && rc != 2) break;
        }
    }
    // a deferred lane leaves its seed in nnq and a negative distance as the mark; its workgroup's record is then the finish launch's
    const int any_deferred = __syncthreads_or(deferred ? 1 : 0);
    if (t == 0) { P.flag[blockIdx.x] = any_deferred; TILE_STAT(7, any_deferred ? 1 : 0); }
    if (do_reduce && !any_deferred && !(SCL_TILE_ABLATE & 2))
        tile_reduce(reinterpret_cast<float *>(L.pts), L.red, p, bq, valid && bi >= 0 && (best <= maxd2), best, P.part + (size_t)blockIdx.x * kNSum);
    if (valid) {                                                 // (behind the reduction: a barrier waits for the stores in front of it)
        P.nnq[i] = make_float4(bq.x, bq.y, bq.z, __int_as_float(bi));
        P.nni[i] = bi;
        P.nnd[i] = deferred ? -1.0f : best;
    }
    TILE_STAMP(14);
}

// The lanes icp_tile_search_kernel could not serve from LDS (a box that did not fit twice; a query with no candidate near it): nn_core's
// own walk through memory from the seed the lane left behind, then the workgroup's record.  One launch behind every tile search; a
// workgroup whose flag is down leaves at once.  (A kernel of its own: inlined in the tile search, this walk's registers cost it two
// of its five waves per SIMD.)
// A workgroup of the grid looks at the flags of the tiles w = blockIdx.x, + gridDim.x, ...: the launch is at most kFinishBlocks workgroups per
// alignment, not one per tile (a launch of 9 775 workgroups that read a flag and leave took 16-22 us of every iteration).
#ifndef SCL_ICP_AHEAD
#define SCL_ICP_AHEAD 2
#endif
#ifndef SCL_ICP_PARTS
#define SCL_ICP_PARTS 2                  // parts a large batch runs as (icp_batch_run).  Point to point, 25 x 100 k: 1 part 6.8 ms, 2 parts 5.85 (with HIP's four hardware queues and with eight), 3 parts 5.6 with four queues but 7.6 with eight, 4 parts 7.3
#endif
constexpr int kFinishBlocks = 128;
constexpr int kFinishGroup = 4;                                  // lanes that share one left-over query's walk (a deferred wave's 64 queries go through in one round; 2 and 8 measured slower)
__global__ __launch_bounds__(kTileQ) void icp_tile_finish_kernel(const IcpProblem *pr, int n_src, int n_tiles, int check_done, int stride, float maxd2, int do_reduce)
{
    __shared__ float pq[kTileQ * 8];
    __shared__ double red[kTileQ / 64][kNSum];
    __shared__ int s_count, s_list[kTileQ];
    const IcpProblem P = pr[blockIdx.y];
    const IcpState *st = P.st;
    const int t = threadIdx.x;
    // the done flag and the flags of this workgroup's tiles, all in flight together (one after the other: a round trip each)
    constexpr int kFlagsPer = 8;
    unsigned int flagged = 0u;
    {
        const int done = st->done;
        int fl[kFlagsPer];
#pragma unroll
        for (int u = 0; u < kFlagsPer; ++u) { const int w = blockIdx.x + u * (int)gridDim.x; fl[u] = P.flag[w < n_tiles ? w : n_tiles - 1]; }
#pragma unroll
        for (int u = 0; u < kFlagsPer; ++u) flagged |= (fl[u] != 0 ? 1u : 0u) << u;
        if (check_done && done) return;
    }
    for (int w = blockIdx.x, u = 0; w < n_tiles; w += gridDim.x, ++u) {
        if (!(u < kFlagsPer ? (int)((flagged >> u) & 1u) : P.flag[w])) continue;   // (the same for every lane of the workgroup)
        const int i = w * kTileQ + t;
        const bool valid = i < n_src;
        const int ic = valid ? i : (n_src > 0 ? n_src - 1 : 0);
        const float4 pw = P.work[ic], nq = P.nnq[ic];            // (unconditional, from a position that exists: both in flight together)
        float best = P.nnd[ic];
        const float3 p = make_float3(pw.x, pw.y, pw.z);
        float3 bq = make_float3(nq.x, nq.y, nq.z);
        int bi = __float_as_int(nq.w);
        const bool marked = valid && best < 0.f;
        // The marked queries of the tile -- one or two as a rule, each a long walk through memory -- are taken by groups of
        // kFinishGroup lanes (nn_core's own sharing of a walk: the rows of a shell or ball dealt round robin, the minimum over
        // (distance, index) order independent): with one lane per query the launch lasted as long as the longest single walk,
        // 30-45 us of every iteration.  The seed travels through the reduction's staging area (free until tile_reduce below).
        if (t == 0) s_count = 0;
        __syncthreads();
        if (marked) {
            s_list[atomicAdd(&s_count, 1)] = t;                  // (any order: a query's result does not depend on who walks it)
            pq[t * 8 + 0] = p.x; pq[t * 8 + 1] = p.y; pq[t * 8 + 2] = p.z; pq[t * 8 + 3] = nq.w;
            pq[t * 8 + 4] = bq.x; pq[t * 8 + 5] = bq.y; pq[t * 8 + 6] = bq.z;
        }
        __syncthreads();
        const int n_marked = s_count;
        for (int e = t / kFinishGroup; e < n_marked; e += kTileQ / kFinishGroup) {
            const int owner = s_list[e];
            const float *o = pq + owner * 8;
            const float3 qp = make_float3(o[0], o[1], o[2]), qb = make_float3(o[4], o[5], o[6]);
            int qi = __float_as_int(o[3]);
            float qd = FLT_MAX;                                  // the seed's distance by the walk's own expression, then the walk
            if (qi >= 0) {
                const float ex = qp.x - qb.x, ey = qp.y - qb.y, ez = qp.z - qb.z;
                const float d = (ex * ex + ey * ey) + ez * ez;
                if (d == d) qd = d; else qi = -1;
            }
            nn_core<kFinishGroup>(qp, st, P.cell_start, P.sorted, t & (kFinishGroup - 1), qi >= 0, qd, qi);
            if ((t & (kFinishGroup - 1)) == 0) { pq[owner * 8 + 3] = __int_as_float(qi); pq[owner * 8 + 7] = qd; }
        }
        __syncthreads();
        if (marked) {
            bi = __float_as_int(pq[t * 8 + 3]);
            best = pq[t * 8 + 7];
            if (bi >= 0) bq = load_xyz(P.tgt, bi, stride);
            P.nnq[i] = make_float4(bq.x, bq.y, bq.z, __int_as_float(bi));
            P.nni[i] = bi;
            P.nnd[i] = best;
        }
        if (do_reduce) tile_reduce(pq, red, p, bq, valid && bi >= 0 && (best <= maxd2), best, P.part + (size_t)w * kNSum);
    }
}

// ---- the same iteration for launches too small to fill the chip with tiles (one alignment, a few small ones) --------------------
// A tile workgroup lives through four dependent rounds of loads; with one workgroup per CU nothing hides them (a lone alignment of
// 50 k points: 30 us per search against 18).  Below kTileMinQueries queries per launch the neighbours are searched in memory, two
// lanes per query (nn_search_kernel_t_body, in the sources' working order), and chunk_reduce_batch_kernel forms the SAME records
// from the SAME values (exact neighbours, tile_reduce on kTileQ consecutive sources): which path a launch takes changes no bit of
// the result, so an alignment on its own still equals the same alignment inside a batch.
constexpr long long kTileMinQueries = 300000;
template <int G>
__global__ __launch_bounds__(256) void nn_search_batch_kernel(const IcpProblem *pr, int n_src, int check_done, int apply_iter, int stride, int warm)
{
    const IcpProblem p = pr[blockIdx.y];
    nn_search_kernel_t_body<G>(p.work, n_src, p.st, p.cell_start, p.sorted, p.nni, p.nnd, check_done, apply_iter, p.tgt, stride, warm);
}

__global__ __launch_bounds__(kTileQ) void chunk_reduce_batch_kernel(const IcpProblem *pr, int n_src, int check_done, int stride, float maxd2)
{
    __shared__ float pq[kTileQ * 8];
    __shared__ double red[kTileQ / 64][kNSum];
    const IcpProblem P = pr[blockIdx.y];
    if (check_done && P.st->done) return;
    const int i = blockIdx.x * kTileQ + threadIdx.x;
    float3 p = make_float3(0.f, 0.f, 0.f), q = make_float3(0.f, 0.f, 0.f);
    float d2 = FLT_MAX;
    int j = -1;
    if (i < n_src) {
        const float4 pw = P.work[i];
        p = make_float3(pw.x, pw.y, pw.z);
        j = P.nni[i]; d2 = P.nnd[i];
        if (j >= 0) q = load_xyz(P.tgt, j, stride);
    }
    tile_reduce(pq, red, p, q, i < n_src && j >= 0 && (d2 <= maxd2), d2, P.part + (size_t)blockIdx.x * kNSum);
}

// ---- the order the sources are worked in: along a Hilbert curve through their own bounding box ---------------------------------
// key = the Hilbert index of the point's cell in a 1024^3 grid over the cloud's box (NaN -> cell 0); a stable radix sort of
// (key, index) gives the permutation.  Consecutive cells of the curve share a face, so any run of consecutive sources is one
// blob in space (a Z-order curve jumps: 7 % of the workgroups' boxes did not fit), for any target grid: the order is made once
// per scan and shared by the alignments against all its candidates.
__device__ __forceinline__ unsigned int spread10(unsigned int v)
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ __launch_bounds__(256) void source_key_kernel(const unsigned char *src, int n, int stride, const float *part, int nparts,
                                                         unsigned int *keys, int *vals)
{
    __shared__ float s_box[6];
    if (threadIdx.x < 64) {                                      // the cloud's box from bbox_partial_kernel's records
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int b = threadIdx.x; b < nparts; b += 64)
#pragma unroll
            for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], part[b * 6 + a]); mx[a] = fmaxf(mx[a], part[b * 6 + 3 + a]); }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int a = 0; a < 3; ++a) { mn[a] = fminf(mn[a], __shfl_xor(mn[a], off, kWave)); mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], off, kWave)); }
        if (threadIdx.x == 0) for (int a = 0; a < 3; ++a) { s_box[a] = mn[a]; s_box[3 + a] = mx[a]; }
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float ext = 0.f;
    for (int a = 0; a < 3; ++a) ext = fmaxf(ext, s_box[3 + a] - s_box[a]);
    const float scale = ext > 0.f ? 1023.0f / ext : 0.f;
    const float3 p = load_xyz(src, i, stride);
    const float v[3] = {p.x, p.y, p.z};
    unsigned int X[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float f = (v[a] - s_box[a]) * scale;
        X[a] = !(f > 0.f) ? 0u : (f >= 1023.f ? 1023u : (unsigned int)f);
    }
    // cell coordinates -> the "transposed" Hilbert index (J. Skilling, Programming the Hilbert curve, AIP Conf. Proc. 707, 2004)
#pragma unroll 1
    for (unsigned int Q = 512u; Q > 1u; Q >>= 1) {
        const unsigned int Pm = Q - 1u;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (X[a] & Q) X[0] ^= Pm;
            else { const unsigned int tt = (X[0] ^ X[a]) & Pm; X[0] ^= tt; X[a] ^= tt; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    unsigned int tt = 0;
#pragma unroll 1
    for (unsigned int Q = 512u; Q > 1u; Q >>= 1) if (X[2] & Q) tt ^= Q - 1u;
    X[0] ^= tt; X[1] ^= tt; X[2] ^= tt;
    keys[i] = (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);
    vals[i] = i;
}

__global__ void work_init_batch_kernel(const IcpProblem *pr, const unsigned char *src, const int *perm, int n, int stride)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float3 p = load_xyz(src, perm[k], stride);
    pr[blockIdx.y].work[k] = make_float4(p.x, p.y, p.z, 0.f);
}

// which == 1 only: the final transform applied to the raw source (in the sources' working order) into work
__global__ void work_final_batch_kernel(const IcpProblem *pr, const unsigned char *src, const int *perm, int n, int stride)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const IcpProblem p = pr[blockIdx.y];
    const float *T = p.st->final_T;
    const float3 s = load_xyz(src, perm[k], stride);
    // distributedMapping.h:247-249 (fp32, left-to-right, no FMA)
    const float ox = T[0] * s.x + T[1] * s.y + T[2] * s.z + T[3];
    const float oy = T[4] * s.x + T[5] * s.y + T[6] * s.z + T[7];
    const float oz = T[8] * s.x + T[9] * s.y + T[10] * s.z + T[11];
    p.work[k] = make_float4(ox, oy, oz, 0.f);
}

// results back in the caller's order
__global__ void unpermute_nn_kernel(const int *perm, const int *nni, const float *nnd, int n, int *out_i, float *out_d)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) { out_i[perm[k]] = nni[k]; out_d[perm[k]] = nnd[k]; }
}

// the alignments' done flags and, at the end, their states side by side: one copy to the host instead of one per alignment
__global__ void gather_states_kernel(const IcpProblem *pr, int nprob, IcpState *out)
{
    const int c = blockIdx.x;
    const int words = (int)(sizeof(IcpState) / sizeof(int));
    const int *s = reinterpret_cast<const int *>(pr[c].st);
    int *d = reinterpret_cast<int *>(out + c);
    for (int w = threadIdx.x; w < words; w += blockDim.x) d[w] = s[w];
}


// ---- host helpers -------------------------------------------------------------------
// partial-sum records of an iteration: one per workgroup of the tile search (point to point), kRedBlocks of the plane reduction
static size_t part_bytes(int n_src)
{
    const size_t q = n_src > 0 ? (size_t)n_src : 1;
    const size_t blocks = (q + kTileQ - 1) / kTileQ + 1;
    return sizeof(double) * kNPlane * (blocks > (size_t)kRedBlocks ? blocks : (size_t)kRedBlocks);
}
static bool use_mfma_reduce()
{
    static const bool v = !scl_lab_is("SCL_ICP_REDUCE", "v");   // SCL_ICP_REDUCE=valu
    return v;
}

#define LAUNCH_REDUCE(grid, stream, ...)                                                                  \
    do {                                                                                                  \
        if (use_mfma_reduce()) hipLaunchKernelGGL(corr_reduce_mfma_kernel, dim3(grid), dim3(256), 0, stream, __VA_ARGS__); \
        else hipLaunchKernelGGL(corr_reduce_kernel, dim3(grid), dim3(256), 0, stream, __VA_ARGS__);       \
    } while (0)

int upload(IcpWorkspace *ws, int k, const void *host, size_t bytes, hipStream_t stream, std::string *err)
{
    int rc = ensure(ws, k, bytes + 16, err);
    if (rc) return rc;
    if (bytes) ICP_HIP(hipMemcpyAsync(ws->buf[k], host, bytes, hipMemcpyHostToDevice, stream));
    return SCL_OK;
}

int build_grid(IcpWorkspace *ws, hipStream_t stream, int n_tgt, int stride, std::string *err)
{
    int rc;
    if ((rc = ensure(ws, B_TSORT, sizeof(float4) * (size_t)(n_tgt + 1), err))) return rc;
    if ((rc = ensure(ws, B_CSTART, sizeof(int) * (size_t)(kMaxCells + 2), err))) return rc;
    if ((rc = ensure(ws, B_CFILL, sizeof(int) * (size_t)(kMaxCells + 2), err))) return rc;
    if ((rc = ensure(ws, B_BBOX, sizeof(float) * 6 * 256, err))) return rc;
    if ((rc = ensure(ws, B_STATE, sizeof(IcpState), err))) return rc;
    const unsigned char *tgt = static_cast<const unsigned char *>(ws->buf[B_TGT]);
    IcpState *st = static_cast<IcpState *>(ws->buf[B_STATE]);
    int nb = (n_tgt + 255) / 256; nb = nb < 1 ? 1 : (nb > 256 ? 256 : nb);
    // counts land in cell_start[cell + 1]; an inclusive scan of the whole table (zeros past the used cells) turns
    // them into start offsets; cell_fill = a copy the scatter advances
    int *cstart = static_cast<int *>(ws->buf[B_CSTART]);
    ICP_HIP(hipMemsetAsync(cstart, 0, sizeof(int) * (size_t)(kMaxCells + 2), stream));
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nb), dim3(256), 0, stream, tgt, n_tgt, stride, (float *)ws->buf[B_BBOX]);
    hipLaunchKernelGGL(grid_setup_kernel, dim3(1), dim3(64), 0, stream, (const float *)ws->buf[B_BBOX], nb, n_tgt, st, cstart);
    int gb = (n_tgt + 255) / 256; gb = gb < 1 ? 1 : (gb > 2048 ? 2048 : gb);
    hipLaunchKernelGGL(grid_count_kernel, dim3(gb), dim3(256), 0, stream, tgt, n_tgt, stride, st, cstart);
    if (scan_scratch_bytes((size_t)kMaxCells + 1) > ws->cap[B_CFILL]) { if (err) *err = "icp: scan scratch larger than the cell table"; return SCL_ERR_NOMEM; }
    ICP_HIP(prefix_sum_i32(ws->buf[B_CFILL], cstart, cstart, kMaxCells + 1, true, stream));   // cell_fill doubles as scratch
    ICP_HIP(hipMemcpyAsync(ws->buf[B_CFILL], cstart, sizeof(int) * (size_t)(kMaxCells + 1), hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(gb), dim3(256), 0, stream, tgt, n_tgt, stride, st, (int *)ws->buf[B_CFILL],
                       (float4 *)ws->buf[B_TSORT]);
    ICP_HIP(hipGetLastError());
    return SCL_OK;
}

// one spin of the host's polling loop (icp_batch_run), on whatever the host is
inline void cpu_pause()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

int pinned(IcpWorkspace *ws, size_t bytes, std::string *err)
{
    if (bytes <= ws->pinned_cap) return SCL_OK;
    if (ws->pinned) { (void)hipHostFree(ws->pinned); ws->pinned = nullptr; ws->pinned_cap = 0; }
    ICP_HIP(hipHostMalloc(&ws->pinned, bytes, hipHostMallocDefault));
    ws->pinned_cap = bytes;
    return SCL_OK;
}

constexpr int kNnWarmGroup = 2;

void launch_nn_search(hipStream_t stream, float4 *work, int n_src, const IcpState *st, const int *cell_start, const float4 *sorted,
                      int *nn_idx, float *nn_d2, int check_done, int apply_iter, const unsigned char *tgt_raw, int stride, int warm)
{
    const long long q = n_src > 0 ? n_src : 1;
    if (warm) {
        const int blocks = (int)((q * kNnWarmGroup + 255) / 256);
        hipLaunchKernelGGL(nn_search_kernel_t<kNnWarmGroup>, dim3(blocks), dim3(256), 0, stream, work, n_src, st, cell_start, sorted,
                           nn_idx, nn_d2, check_done, apply_iter, tgt_raw, stride, warm);
    } else {
        const int blocks = (int)((q * kNnGroup + 255) / 256);
        hipLaunchKernelGGL(nn_search_kernel_t<kNnGroup>, dim3(blocks), dim3(256), 0, stream, work, n_src, st, cell_start, sorted,
                           nn_idx, nn_d2, check_done, apply_iter, tgt_raw, stride, warm);
    }
}

int check_cloud_args(int n_src, int n_tgt, int stride, std::string *err)
{
    if (n_src < 0 || n_tgt < 0 || stride < 12 || (stride & 3)) { if (err) *err = "icp: bad cloud layout"; return SCL_ERR_INVALID_ARG; }
    return SCL_OK;
}

}  // namespace

void icp_tile_stats(unsigned long long out[16], bool reset)
{
#ifdef SCL_DIAGNOSTICS
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_stats), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_stats), z, sizeof(z)); }
#else
    (void)reset;
    for (int k = 0; k < 16; ++k) out[k] = 0;
#endif
}

void icp_workspace_free(IcpWorkspace *ws)
{

    for (size_t i = 0; i < sizeof(ws->buf) / sizeof(ws->buf[0]); ++i) if (ws->buf[i]) { (void)hipFree(ws->buf[i]); ws->buf[i] = nullptr; ws->cap[i] = 0; }
    if (ws->side) { (void)hipStreamDestroy(ws->side); ws->side = nullptr; }
    for (int k = 0; k < IcpWorkspace::kMaxParts - 1; ++k) {
        if (ws->part_stream[k]) { (void)hipStreamDestroy(ws->part_stream[k]); ws->part_stream[k] = nullptr; }
        if (ws->ev_join[k]) { (void)hipEventDestroy(ws->ev_join[k]); ws->ev_join[k] = nullptr; }
    }
    for (int k = 0; k < IcpWorkspace::kMaxParts; ++k) if (ws->ev_norm[k]) { (void)hipEventDestroy(ws->ev_norm[k]); ws->ev_norm[k] = nullptr; }
    if (ws->ev_fork) { (void)hipEventDestroy(ws->ev_fork); ws->ev_fork = nullptr; }
    if (ws->pinned) { (void)hipHostFree(ws->pinned); ws->pinned = nullptr; ws->pinned_cap = 0; }
}

int icp_align(IcpWorkspace *ws, hipStream_t stream, int num_cu, const void *src, int n_src,
              const void *tgt, int n_tgt, int stride, const scl_icp_params &p,
              float T[16], float *fitness, int *converged, int *iterations, std::string *err)
{
    (void)num_cu;
    int rc = check_cloud_args(n_src, n_tgt, stride, err);
    if (rc) return rc;
    if ((rc = upload(ws, B_SRC, src, (size_t)n_src * stride, stream, err))) return rc;
    if ((rc = upload(ws, B_TGT, tgt, (size_t)n_tgt * stride, stream, err))) return rc;
    return icp_align_staged(ws, stream, n_src, n_tgt, stride, p, T, fitness, converged, iterations, err);
}

int icp_stage_cloud_host(IcpWorkspace *ws, hipStream_t stream, bool target, const void *h_cloud, int n, int stride,
                         std::string *err)
{
    if (n < 0 || stride < 12 || (stride & 3)) { if (err) *err = "icp_stage_cloud_host: bad arguments"; return SCL_ERR_INVALID_ARG; }
    return upload(ws, target ? B_TGT : B_SRC, h_cloud, (size_t)n * stride, stream, err);
}

int icp_stage_cloud(IcpWorkspace *ws, hipStream_t stream, bool target, const void *d_cloud, int n, int stride,
                    std::string *err)
{
    if (n < 0 || stride < 12 || (stride & 3)) { if (err) *err = "icp_stage_cloud: bad arguments"; return SCL_ERR_INVALID_ARG; }
    const int k = target ? B_TGT : B_SRC;
    const size_t bytes = (size_t)n * stride;
    int rc = ensure(ws, k, bytes + 16, err);
    if (rc) return rc;
    if (bytes) ICP_HIP(hipMemcpyAsync(ws->buf[k], d_cloud, bytes, hipMemcpyDeviceToDevice, stream));
    return SCL_OK;
}

int icp_batch_prepare(IcpWorkspace *ws, hipStream_t stream, const void *d_src, int n_src, int n_tgt, int stride,
                      const scl_icp_params &p, std::string *err);
int icp_batch_run(IcpWorkspace *const *wss, int nprob, IcpWorkspace *ctl, hipStream_t stream, const void *d_src, int n_src,
                  int stride, const scl_icp_params &p, float *T, float *fitness, int *converged, int *iterations, std::string *err);

// One alignment = a batch of one: the same launches, the same order of every sum (tests compare the two bit for bit).
int icp_align_staged(IcpWorkspace *ws, hipStream_t stream, int n_src, int n_tgt, int stride, const scl_icp_params &p,
                     float T[16], float *fitness, int *converged, int *iterations, std::string *err)
{
    int rc = icp_batch_prepare(ws, stream, ws->buf[B_SRC], n_src, n_tgt, stride, p, err);
    if (rc) return rc;
    IcpWorkspace *one[1] = {ws};
    float fit = 0.f; int conv = 0, iters = 0;
    rc = icp_batch_run(one, 1, ws, stream, ws->buf[B_SRC], n_src, stride, p, T, &fit, &conv, &iters, err);
    if (rc) return rc;
    if (fitness) *fitness = fit;
    if (converged) *converged = conv;
    if (iterations) *iterations = iters;
    return SCL_OK;
}

// The order the batch's sources are worked in (B_PERM of ctl): see source_key_kernel.
static int source_order(IcpWorkspace *ctl, hipStream_t stream, const void *d_src, int n_src, int stride, std::string *err)
{
    int rc;
    const size_t n = n_src > 0 ? (size_t)n_src : 1;
    if ((rc = ensure(ctl, B_PERM, sizeof(int) * (n + 1), err))) return rc;
    if (n_src <= 0) return SCL_OK;
    if ((rc = ensure(ctl, B_BBOX, sizeof(float) * 6 * 256, err))) return rc;
    const size_t tmp_bytes = sort_scratch_bytes(n, 1);
    const size_t arr = (sizeof(int) * n + 255) & ~(size_t)255;
    if ((rc = ensure(ctl, B_SORT, 3 * arr + tmp_bytes + 256, err))) return rc;
    unsigned char *base = static_cast<unsigned char *>(ctl->buf[B_SORT]);
    unsigned int *keys_in = reinterpret_cast<unsigned int *>(base), *keys_out = reinterpret_cast<unsigned int *>(base + arr);
    int *vals_in = reinterpret_cast<int *>(base + 2 * arr);
    void *tmp = base + 3 * arr;
    int nb = (n_src + 255) / 256; nb = nb > 256 ? 256 : nb;
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nb), dim3(256), 0, stream, (const unsigned char *)d_src, n_src, stride, (float *)ctl->buf[B_BBOX]);
    hipLaunchKernelGGL(source_key_kernel, dim3((n_src + 255) / 256), dim3(256), 0, stream, (const unsigned char *)d_src, n_src, stride,
                       (const float *)ctl->buf[B_BBOX], nb, keys_in, vals_in);
    ICP_HIP(sort_pairs_u32(tmp, keys_in, keys_out, (unsigned int *)vals_in, (unsigned int *)ctl->buf[B_PERM], n_src, 30, stream));
    ICP_HIP(hipGetLastError());
    return SCL_OK;
}

// ---- the alignments of one scan's loop candidates, every loop step one launch for all of them (BASELINE configs[2]) ------------
// icp_batch_prepare: everything an alignment needs before its first iteration that depends on its target (staged in ws, B_TGT):
// buffers, NN grid, normals (point-to-plane), state.  (The working clouds are set up by icp_batch_run, in the sources' working order.)
int icp_batch_prepare(IcpWorkspace *ws, hipStream_t stream, const void *d_src, int n_src, int n_tgt, int stride,
                      const scl_icp_params &p, std::string *err)
{
    (void)d_src;
    ws->ext_tgt = nullptr;                                       // the target is the one staged in B_TGT
    int rc = check_cloud_args(n_src, n_tgt, stride, err);
    if (rc) return rc;
    if (p.estimator != 0 && p.estimator != 1) { if (err) *err = "unknown estimator"; return SCL_ERR_INVALID_ARG; }
    if (p.estimator == 1 && !(p.normal_radius > 0.0)) { if (err) *err = "normal_radius must be > 0"; return SCL_ERR_INVALID_ARG; }
    if (p.max_iterations < 1) { if (err) *err = "max_iterations < 1"; return SCL_ERR_INVALID_ARG; }
    if ((rc = ensure(ws, B_WORK, sizeof(float4) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_NNQ, sizeof(float4) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_FLAG, sizeof(int) * ((size_t)(n_src > n_tgt ? n_src : n_tgt) / kTileQ + 2), err))) return rc;
    if ((rc = ensure(ws, B_NNI, sizeof(int) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_NND, sizeof(float) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_PART, part_bytes(n_src), err))) return rc;
    ws->n_tgt = n_tgt;
    if ((rc = build_grid(ws, stream, n_tgt, stride, err))) return rc;
    if (p.estimator == 1) {
        if ((rc = ensure(ws, B_NORM, sizeof(float4) * (size_t)(n_tgt + 1), err))) return rc;
        hipLaunchKernelGGL(normals_kernel, dim3(n_tgt > 0 ? (unsigned)(((size_t)n_tgt * kNormGroup + kNormBlock - 1) / kNormBlock) : 1u), dim3(kNormBlock), 0, stream,
                           (const unsigned char *)ws->buf[B_TGT], n_tgt, stride, (const IcpState *)ws->buf[B_STATE],
                           (const int *)ws->buf[B_CSTART], (const float4 *)ws->buf[B_TSORT], p.normal_radius,
                           (float4 *)ws->buf[B_NORM]);
    }
    hipLaunchKernelGGL(state_init_kernel, dim3(1), dim3(64), 0, stream, (IcpState *)ws->buf[B_STATE]);
    ICP_HIP(hipGetLastError());
    return SCL_OK;
}

// icp_batch_prepare for n alignments at once, their targets already on the device (d_tgts[c], n_tgts[c] points: read in place, not
// copied): every step one launch over all of them, on one stream.  ctl keeps the table of the jobs.
int icp_batch_prepare_all(IcpWorkspace *const *wss, int n, IcpWorkspace *ctl, hipStream_t stream, int n_src, const void *const *d_tgts,
                          const int *n_tgts, int stride, const scl_icp_params &p, std::string *err)
{
    if (n <= 0) return SCL_OK;
    if (p.estimator != 0 && p.estimator != 1) { if (err) *err = "unknown estimator"; return SCL_ERR_INVALID_ARG; }
    if (p.estimator == 1 && !(p.normal_radius > 0.0)) { if (err) *err = "normal_radius must be > 0"; return SCL_ERR_INVALID_ARG; }
    if (p.max_iterations < 1) { if (err) *err = "max_iterations < 1"; return SCL_ERR_INVALID_ARG; }
    int rc, max_n = 0;
    std::vector<GridJob> jobs((size_t)n);
    for (int c = 0; c < n; ++c) {
        IcpWorkspace *ws = wss[c];
        const int n_tgt = n_tgts[c];
        if ((rc = check_cloud_args(n_src, n_tgt, stride, err))) return rc;
        if ((rc = ensure(ws, B_WORK, sizeof(float4) * (size_t)(n_src + 1), err))) return rc;
        if ((rc = ensure(ws, B_NNQ, sizeof(float4) * (size_t)(n_src + 1), err))) return rc;
        if ((rc = ensure(ws, B_FLAG, sizeof(int) * ((size_t)(n_src > n_tgt ? n_src : n_tgt) / kTileQ + 2), err))) return rc;
        if ((rc = ensure(ws, B_NNI, sizeof(int) * (size_t)(n_src + 1), err))) return rc;
        if ((rc = ensure(ws, B_NND, sizeof(float) * (size_t)(n_src + 1), err))) return rc;
        if ((rc = ensure(ws, B_PART, part_bytes(n_src), err))) return rc;
        if ((rc = ensure(ws, B_TSORT, sizeof(float4) * (size_t)(n_tgt + 1), err))) return rc;
        if ((rc = ensure(ws, B_CSTART, sizeof(int) * (size_t)(kMaxCells + 2), err))) return rc;
        if ((rc = ensure(ws, B_CFILL, sizeof(int) * (size_t)(kMaxCells + 2), err))) return rc;
        if ((rc = ensure(ws, B_BBOX, sizeof(float) * 6 * 256, err))) return rc;
        if ((rc = ensure(ws, B_STATE, sizeof(IcpState), err))) return rc;
        if (p.estimator == 1 && (rc = ensure(ws, B_NORM, sizeof(float4) * (size_t)(n_tgt + 1), err))) return rc;
        ws->ext_tgt = d_tgts[c];
        ws->n_tgt = n_tgt;
        GridJob &j = jobs[(size_t)c];
        j.tgt = static_cast<const unsigned char *>(d_tgts[c]); j.n = n_tgt; j.st = (IcpState *)ws->buf[B_STATE];
        j.cstart = (int *)ws->buf[B_CSTART]; j.cfill = (int *)ws->buf[B_CFILL]; j.tsort = (float4 *)ws->buf[B_TSORT];
        j.bbox = (float *)ws->buf[B_BBOX]; j.normals = p.estimator == 1 ? (float4 *)ws->buf[B_NORM] : nullptr;
        max_n = n_tgt > max_n ? n_tgt : max_n;
    }
    if ((rc = ensure(ctl, B_PROB, sizeof(GridJob) * (size_t)n, err))) return rc;
    ICP_HIP(hipMemcpyAsync(ctl->buf[B_PROB], jobs.data(), sizeof(GridJob) * (size_t)n, hipMemcpyHostToDevice, stream));   // (pageable: staged before the call returns)
    const GridJob *dj = static_cast<const GridJob *>(ctl->buf[B_PROB]);
    int gb = (max_n + 255) / 256; gb = gb < 1 ? 1 : (gb > 2048 ? 2048 : gb);
    hipLaunchKernelGGL(bbox_partial_batch_kernel, dim3(256, n), dim3(256), 0, stream, dj, stride);
    hipLaunchKernelGGL(grid_setup_batch_kernel, dim3(n), dim3(64), 0, stream, dj, 256);
    hipLaunchKernelGGL(grid_zero_batch_kernel, dim3(64, n), dim3(256), 0, stream, dj);
    hipLaunchKernelGGL(grid_count_batch_kernel, dim3(gb, n), dim3(256), 0, stream, dj, stride);
    hipLaunchKernelGGL(grid_scan_batch_kernel, dim3(n), dim3(1024), 0, stream, dj);
    hipLaunchKernelGGL(grid_scatter_batch_kernel, dim3(gb, n), dim3(256), 0, stream, dj, stride);
    // the normals (point to plane) are first needed by the plane reduction behind the cold search: icp_batch_run launches them beside
    // the cold searches of its parts
    ctl->normals_pending = p.estimator == 1 && max_n > 0;
    ctl->normals_radius = p.normal_radius;
    hipLaunchKernelGGL(state_init_batch_kernel, dim3(n), dim3(64), 0, stream, dj);
    ICP_HIP(hipGetLastError());
    return SCL_OK;
}

static void fill_problem(IcpProblem *hp, IcpWorkspace *ws, bool normals)
{
    hp->work = (float4 *)ws->buf[B_WORK]; hp->st = (IcpState *)ws->buf[B_STATE];
    hp->cell_start = (const int *)ws->buf[B_CSTART]; hp->sorted = (const float4 *)ws->buf[B_TSORT];
    hp->nni = (int *)ws->buf[B_NNI]; hp->nnd = (float *)ws->buf[B_NND]; hp->part = (double *)ws->buf[B_PART];
    hp->tgt = (const unsigned char *)(ws->ext_tgt ? ws->ext_tgt : ws->buf[B_TGT]); hp->normals = normals ? (const float4 *)ws->buf[B_NORM] : nullptr;
    hp->nnq = (float4 *)ws->buf[B_NNQ]; hp->flag = (int *)ws->buf[B_FLAG];
    hp->n_tgt = ws->n_tgt;
}

// icp_batch_run: the ICP loops and the fitness passes of nprob prepared alignments, every step one launch for all of
// them.  ctl keeps the sources' working order, the problem table (device) and the pinned read-back area (it may be the one
// alignment's own workspace).  An iteration (DM.h:1107-1121's loop body) is two launches: icp_tile_search_kernel (the previous
// increment applied, neighbours, this workgroup's sums) and the solve; point to plane keeps its reduction as a launch of its own.
int icp_batch_run(IcpWorkspace *const *wss, int nprob, IcpWorkspace *ctl, hipStream_t stream, const void *d_src, int n_src,
                  int stride, const scl_icp_params &p, float *T, float *fitness, int *converged, int *iterations, std::string *err)
{
    if (nprob <= 0) return SCL_OK;
    int rc;
    constexpr int kAhead = SCL_ICP_AHEAD;                        // iterations the host may be ahead of what a part has finished
    constexpr int kMaxParts = IcpWorkspace::kMaxParts;
    constexpr size_t kWordPitch = 64;                            // a part's progress word has a cache line of pinned memory to itself
    if ((rc = ensure(ctl, B_MASK, sizeof(IcpProblem) * (size_t)nprob, err))) return rc;
    if ((rc = ensure(ctl, B_HYP, sizeof(IcpState) * (size_t)nprob + 2 * sizeof(int) * kMaxParts, err))) return rc;
    if ((rc = pinned(ctl, (sizeof(IcpProblem) + sizeof(IcpState)) * (size_t)nprob + 64 + kWordPitch * kMaxParts, err))) return rc;
    IcpProblem *hp = static_cast<IcpProblem *>(ctl->pinned);
    IcpState *hs = reinterpret_cast<IcpState *>(hp + nprob);
    unsigned char *h_words = reinterpret_cast<unsigned char *>(((uintptr_t)(hs + nprob) + 63) & ~(uintptr_t)63);
    IcpState *d_states = static_cast<IcpState *>(ctl->buf[B_HYP]);
    int *d_counters = reinterpret_cast<int *>(d_states + nprob);
    for (int k = 0; k < kMaxParts; ++k) *reinterpret_cast<volatile unsigned long long *>(h_words + kWordPitch * (size_t)k) = 0ull;
    ICP_HIP(hipMemsetAsync(d_counters, 0, 2 * sizeof(int) * kMaxParts, stream));
    if ((rc = source_order(ctl, stream, d_src, n_src, stride, err))) return rc;
    for (int c = 0; c < nprob; ++c) fill_problem(&hp[c], wss[c], p.estimator == 1);
    const IcpProblem *dp_all = static_cast<const IcpProblem *>(ctl->buf[B_MASK]);
    ICP_HIP(hipMemcpyAsync(ctl->buf[B_MASK], hp, sizeof(IcpProblem) * (size_t)nprob, hipMemcpyHostToDevice, stream));
    const unsigned char *src = static_cast<const unsigned char *>(d_src);
    const int *perm = static_cast<const int *>(ctl->buf[B_PERM]);
    const long long q = n_src > 0 ? n_src : 1;
    const int pb = (int)((q + 255) / 256), tb = (int)((q + kTileQ - 1) / kTileQ);
    const int rb = pb < kRedBlocks ? pb : kRedBlocks;
    const float maxd2 = (float)(p.max_correspondence_dist * p.max_correspondence_dist);
    hipLaunchKernelGGL(work_init_batch_kernel, dim3(pb, nprob), dim3(256), 0, stream, dp_all, src, perm, n_src, stride);
    const bool tiles = (long long)nprob * q >= kTileMinQueries;   // (no bit of the result depends on it: see chunk_reduce_batch_kernel)
    // The alignments are independent, and an iteration is a chain -- search, finish, solve -- whose last two links are short launches
    // that leave the chip almost idle (38-48 of an iteration's 280 us; the search's own last workgroups likewise).  So a batch that is
    // large enough runs as kParts parts, each the same chain over its share of the alignments on a stream of its own: one part's
    // finish and solve run under another part's search.  Which part an alignment is in changes no bit of its result (every launch
    // treats blockIdx.y's alignment on its own); a part keeps at least kTileMinQueries queries, so that it searches as the whole would.
    int parts = 1;
    if (tiles) {
        parts = SCL_ICP_PARTS;
        while (parts > 1 && ((long long)(nprob / parts) * q < kTileMinQueries || nprob / parts < 4)) --parts;
    }
    struct Part { int first, n; hipStream_t s; int next_it; bool finished; PartSync sync; };
    Part part[kMaxParts];
    for (int k = 0; k < parts; ++k) {
        part[k].first = (int)((long long)nprob * k / parts);
        part[k].n = (int)((long long)nprob * (k + 1) / parts) - part[k].first;
        part[k].next_it = 1; part[k].finished = false;
        part[k].s = stream;
        part[k].sync.counters = d_counters + 2 * k;
        part[k].sync.host_word = reinterpret_cast<unsigned long long *>(h_words + kWordPitch * (size_t)k);
        part[k].sync.seq = 0;
    }
    if (parts > 1) {
        if (!ctl->ev_fork) ICP_HIP(hipEventCreateWithFlags(&ctl->ev_fork, hipEventDisableTiming));
        ICP_HIP(hipEventRecord(ctl->ev_fork, stream));
        for (int k = 1; k < parts; ++k) {
            if (!ctl->part_stream[k - 1]) ICP_HIP(hipStreamCreateWithFlags(&ctl->part_stream[k - 1], hipStreamNonBlocking));
            if (!ctl->ev_join[k - 1]) ICP_HIP(hipEventCreateWithFlags(&ctl->ev_join[k - 1], hipEventDisableTiming));
            part[k].s = ctl->part_stream[k - 1];
            ICP_HIP(hipStreamWaitEvent(part[k].s, ctl->ev_fork, 0));
        }
    }
    // Point to plane, the targets' normals still to be computed (icp_batch_prepare_all): a kernel bound by its vector instructions, 1.1-1.4 ms
    // for 25 x 100 k points, beside cold searches that wait for memory.  With two parts and TWO streams: part 1's stream computes part 0's
    // normals and then searches, part 0's stream searches and then computes part 1's normals -- one search and one half of the normals
    // on the chip at any time, and each part's first plane reduction waits for the event behind its normals on the other stream.  (On a
    // side stream of their own the normals shared a hardware queue with a part's stream on some runs, and that part's cold search then
    // started when they ended: 1.2 ms late.)  One part: the side stream as before.
    const bool normals_todo = ctl->normals_pending && p.estimator == 1;
    ctl->normals_pending = false;
    int max_tgt = 0;
    for (int c = 0; c < nprob; ++c) max_tgt = wss[c]->n_tgt > max_tgt ? wss[c]->n_tgt : max_tgt;
    auto launch_normals = [&](const Part &of, hipStream_t on) {
        hipLaunchKernelGGL(normals_problems_kernel, dim3((unsigned)(((size_t)max_tgt * kNormGroup + kNormBlock - 1) / kNormBlock), of.n), dim3(kNormBlock), 0, on,
                           dp_all + of.first, ctl->normals_radius);
    };
    if (normals_todo) {
        for (int k = 0; k < parts; ++k) if (!ctl->ev_norm[k]) ICP_HIP(hipEventCreateWithFlags(&ctl->ev_norm[k], hipEventDisableTiming));
        if (parts == 1) {
            if (!ctl->side) ICP_HIP(hipStreamCreateWithFlags(&ctl->side, hipStreamNonBlocking));
            if (!ctl->ev_fork) ICP_HIP(hipEventCreateWithFlags(&ctl->ev_fork, hipEventDisableTiming));
            ICP_HIP(hipEventRecord(ctl->ev_fork, stream));
            ICP_HIP(hipStreamWaitEvent(ctl->side, ctl->ev_fork, 0));
            launch_normals(part[0], ctl->side);
            ICP_HIP(hipEventRecord(ctl->ev_norm[0], ctl->side));
        }
    }
    // neighbours (the previous increment applied first) and, point to point, the records of the workgroups' sums
    auto search_and_sums = [&](const Part &P, bool cold, int check_done, int apply, float md2, bool sums) {
        const IcpProblem *dp = dp_all + P.first;
        if (tiles) {
            hipLaunchKernelGGL(icp_tile_search_kernel, dim3(tb, P.n), dim3(kTileQ), 0, P.s, dp, n_src, check_done, apply, cold ? 1 : 0, stride, md2, sums ? 1 : 0);
            hipLaunchKernelGGL(icp_tile_finish_kernel, dim3(tb < kFinishBlocks ? tb : kFinishBlocks, P.n), dim3(kTileQ), 0, P.s, dp, n_src, tb, check_done, stride, md2, sums ? 1 : 0);
            return;
        }
        if (cold) hipLaunchKernelGGL(nn_search_batch_kernel<kNnGroup>, dim3((unsigned)((q * kNnGroup + 255) / 256), P.n), dim3(256), 0, P.s,
                                     dp, n_src, check_done, apply ? 1 : -1, stride, 0);
        else hipLaunchKernelGGL(nn_search_batch_kernel<kNnWarmGroup>, dim3((unsigned)((q * kNnWarmGroup + 255) / 256), P.n), dim3(256), 0, P.s,
                                dp, n_src, check_done, apply ? 1 : -1, stride, 1);
        if (sums) hipLaunchKernelGGL(chunk_reduce_batch_kernel, dim3(tb, P.n), dim3(kTileQ), 0, P.s, dp, n_src, check_done, stride, md2);
    };
    bool cold_wait[kMaxParts] = {false, false, false, false};
    auto iteration = [&](Part &P, bool cold) {
        const IcpProblem *dp = dp_all + P.first;
        const int k = (int)(&P - part);
        P.sync.seq += 1;
        if (cold && normals_todo && parts > 1 && (k & 1)) {      // an odd part's stream: the normals of the part before it, then its own search
            launch_normals(part[k - 1], P.s);
            (void)hipEventRecord(ctl->ev_norm[k - 1], P.s);
        }
        search_and_sums(P, cold, 1, cold ? 0 : 1, maxd2, p.estimator == 0);
        if (p.estimator == 1) {
            if (cold && normals_todo) {
                if (parts > 1 && !(k & 1)) {                     // an even part's stream: its own search, then the normals of the part behind it (its own, if it is the last)
                    const int of = k + 1 < parts ? k + 1 : k;
                    launch_normals(part[of], P.s);
                    (void)hipEventRecord(ctl->ev_norm[of], P.s);
                }
                cold_wait[k] = true;                             // (the wait is enqueued when every part's normals have their event: below)
                return;
            }
            hipLaunchKernelGGL(plane_reduce_batch_kernel, dim3(rb, P.n), dim3(256), 0, P.s, dp, stride, n_src, maxd2);
            hipLaunchKernelGGL(plane_solve_batch_kernel, dim3(P.n), dim3(256), 0, P.s, dp, rb, p.max_iterations, p.transformation_epsilon,
                               p.euclidean_fitness_epsilon, P.sync);
        } else {
            hipLaunchKernelGGL(icp_solve_batch_kernel, dim3(P.n), dim3(256), 0, P.s, dp, tb, 0, p.max_iterations, p.transformation_epsilon,
                               p.euclidean_fitness_epsilon, P.sync);
        }
    };
    // The host stays kAhead iterations ahead of what a part has FINISHED (the progress word its solve launches write into pinned
    // memory: launches finished, alignments done): the device always has the next iteration queued, and a part whose alignments are
    // all done costs kAhead empty iterations (a finished alignment's workgroups leave at once), not the four to eight of a look at
    // copied flags every fourth iteration -- 0.26 of a 25-candidate query's 5.8 ms.  The parts are served as they become ready.
    for (int k = parts - 1; k >= 0; --k) iteration(part[k], true);       // (odd parts first: their streams start with the normals the even parts wait for)
    for (int k = 0; k < parts; ++k) {
        if (!cold_wait[k]) continue;                             // point to plane, cold: the rest of the iteration behind the part's normals
        const Part &P = part[k];
        (void)hipStreamWaitEvent(P.s, ctl->ev_norm[k], 0);
        hipLaunchKernelGGL(plane_reduce_batch_kernel, dim3(rb, P.n), dim3(256), 0, P.s, dp_all + P.first, stride, n_src, maxd2);
        hipLaunchKernelGGL(plane_solve_batch_kernel, dim3(P.n), dim3(256), 0, P.s, dp_all + P.first, rb, p.max_iterations, p.transformation_epsilon,
                           p.euclidean_fitness_epsilon, P.sync);
    }
    for (unsigned long long idle = 0;;) {
        bool open = false, progress = false;
        for (int k = 0; k < parts; ++k) {
            Part &P = part[k];
            if (P.finished || P.next_it >= p.max_iterations) continue;
            open = true;
            const int want = P.next_it - kAhead;                 // the solve launch of the part that must have finished (the cold one is 1)
            if (want >= 1) {
                const unsigned long long w = __atomic_load_n(P.sync.host_word, __ATOMIC_ACQUIRE);
                if ((long long)(w >> 32) < (long long)want) continue;
                if ((long long)(w & 0xffffffffull) >= (long long)P.n) { P.finished = true; progress = true; continue; }
            }
            iteration(P, false);
            P.next_it += 1;
            progress = true;
        }
        if (!open) break;
        if (progress) { idle = 0; continue; }
        if ((++idle & 0xfffffull) == 0) {                        // nothing moved for a while: has a stream died, or drained without its word?
            for (int k = 0; k < parts; ++k) {
                const hipError_t qe = hipStreamQuery(part[k].s);
                if (qe != hipSuccess && qe != hipErrorNotReady) { if (err) *err = std::string("icp_batch_run: ") + hipGetErrorString(qe); return SCL_ERR_HIP; }
            }
            if (idle > (1ull << 34)) { if (err) *err = "icp_batch_run: no progress word from the device"; return SCL_ERR_HIP; }
        }
        cpu_pause();
    }
    // fitness: the original source moved by each final transform, mean squared NN distance over all points (the last
    // neighbour bounds the search)
    for (int k = 0; k < parts; ++k) {
        const Part &P = part[k];
        hipLaunchKernelGGL(work_final_batch_kernel, dim3(pb, P.n), dim3(256), 0, P.s, dp_all + P.first, src, perm, n_src, stride);
        search_and_sums(P, false, 0, 0, FLT_MAX, true);
        hipLaunchKernelGGL(icp_solve_batch_kernel, dim3(P.n), dim3(256), 0, P.s, dp_all + P.first, tb, 2, 0, 0.0, 0.0, PartSync{nullptr, nullptr, 0u});
        if (k > 0) {
            ICP_HIP(hipEventRecord(ctl->ev_join[k - 1], P.s));
            ICP_HIP(hipStreamWaitEvent(stream, ctl->ev_join[k - 1], 0));
        }
    }
    hipLaunchKernelGGL(gather_states_kernel, dim3(nprob), dim3(64), 0, stream, dp_all, nprob, d_states);
    ICP_HIP(hipGetLastError());
    ICP_HIP(hipMemcpyAsync(hs, d_states, sizeof(IcpState) * (size_t)nprob, hipMemcpyDeviceToHost, stream));
    ICP_HIP(hipStreamSynchronize(stream));
    for (int c = 0; c < nprob; ++c) {
        std::memcpy(T + 16 * (size_t)c, hs[c].final_T, sizeof(float) * 16);
        if (fitness) fitness[c] = (float)hs[c].fitness;
        if (converged) converged[c] = hs[c].converged;
        if (iterations) iterations[c] = hs[c].iter;
    }
    return SCL_OK;
}

// T_move == nullptr: every source point's nearest target point (a cold search).  T_move: the correspondences of the source moved
// by T_move, searched from those of the unmoved source -- the warm search of a loop iteration, on its own.
int icp_nn_correspondences(IcpWorkspace *ws, hipStream_t stream, int num_cu, const void *src, int n_src,
                           const void *tgt, int n_tgt, int stride, const float *T_move, int *nn_index, float *nn_dist2, std::string *err)
{
    (void)num_cu;
    int rc = check_cloud_args(n_src, n_tgt, stride, err);
    if (rc) return rc;
    if (n_src == 0) return SCL_OK;
    ws->ext_tgt = nullptr;
    if ((rc = upload(ws, B_SRC, src, (size_t)n_src * stride, stream, err))) return rc;
    if ((rc = upload(ws, B_TGT, tgt, (size_t)n_tgt * stride, stream, err))) return rc;
    if ((rc = ensure(ws, B_WORK, sizeof(float4) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_NNQ, sizeof(float4) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_FLAG, sizeof(int) * ((size_t)(n_src > 0 ? n_src : 1) / kTileQ + 2), err))) return rc;
    if ((rc = ensure(ws, B_NNI, sizeof(int) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_NND, sizeof(float) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_PART, part_bytes(n_src), err))) return rc;
    if ((rc = ensure(ws, B_SI, sizeof(int) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_TI, sizeof(float) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_MASK, sizeof(IcpProblem), err))) return rc;
    if ((rc = pinned(ws, sizeof(IcpProblem) + sizeof(IcpState), err))) return rc;
    if ((rc = build_grid(ws, stream, n_tgt, stride, err))) return rc;
    if ((rc = source_order(ws, stream, ws->buf[B_SRC], n_src, stride, err))) return rc;
    IcpProblem *hp = static_cast<IcpProblem *>(ws->pinned);
    fill_problem(hp, ws, false);
    ICP_HIP(hipMemcpyAsync(ws->buf[B_MASK], hp, sizeof(IcpProblem), hipMemcpyHostToDevice, stream));
    const IcpProblem *dp = static_cast<const IcpProblem *>(ws->buf[B_MASK]);
    const int pb = (n_src + 255) / 256, tb = (n_src + kTileQ - 1) / kTileQ;
    hipLaunchKernelGGL(state_init_kernel, dim3(1), dim3(64), 0, stream, (IcpState *)ws->buf[B_STATE]);
    hipLaunchKernelGGL(work_init_batch_kernel, dim3(pb, 1), dim3(256), 0, stream, dp, (const unsigned char *)ws->buf[B_SRC], (const int *)ws->buf[B_PERM], n_src, stride);
    hipLaunchKernelGGL(icp_tile_search_kernel, dim3(tb, 1), dim3(kTileQ), 0, stream, dp, n_src, 0, 0, 1, stride, FLT_MAX, 0);
    hipLaunchKernelGGL(icp_tile_finish_kernel, dim3(tb < kFinishBlocks ? tb : kFinishBlocks, 1), dim3(kTileQ), 0, stream, dp, n_src, tb, 0, stride, FLT_MAX, 0);
    if (T_move) {
        IcpState *st = static_cast<IcpState *>(ws->buf[B_STATE]);
        ICP_HIP(hipMemcpyAsync(st->inc_T, T_move, sizeof(float) * 16, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(icp_tile_search_kernel, dim3(tb, 1), dim3(kTileQ), 0, stream, dp, n_src, 0, 1, 0, stride, FLT_MAX, 0);
        hipLaunchKernelGGL(icp_tile_finish_kernel, dim3(tb < kFinishBlocks ? tb : kFinishBlocks, 1), dim3(kTileQ), 0, stream, dp, n_src, tb, 0, stride, FLT_MAX, 0);
    }
    hipLaunchKernelGGL(unpermute_nn_kernel, dim3(pb), dim3(256), 0, stream, (const int *)ws->buf[B_PERM], (const int *)ws->buf[B_NNI], (const float *)ws->buf[B_NND],
                       n_src, (int *)ws->buf[B_SI], (float *)ws->buf[B_TI]);
    ICP_HIP(hipGetLastError());
    ICP_HIP(hipMemcpyAsync(nn_index, ws->buf[B_SI], sizeof(int) * (size_t)n_src, hipMemcpyDeviceToHost, stream));
    if (nn_dist2) ICP_HIP(hipMemcpyAsync(nn_dist2, ws->buf[B_TI], sizeof(float) * (size_t)n_src, hipMemcpyDeviceToHost, stream));
    ICP_HIP(hipStreamSynchronize(stream));
    return SCL_OK;
}

int icp_rigid_svd(IcpWorkspace *ws, hipStream_t stream, int num_cu, const void *src, int n_src,
                  const void *tgt, int n_tgt, int stride, const int *src_index, const int *tgt_index,
                  int n_corr, float T[16], std::string *err)
{
    (void)num_cu;
    int rc = check_cloud_args(n_src, n_tgt, stride, err);
    if (rc) return rc;
    if (n_corr < 3) { if (err) *err = "rigid_svd needs >= 3 correspondences"; return SCL_ERR_INVALID_ARG; }
    for (int i = 0; i < n_corr; ++i)
        if (src_index[i] < 0 || src_index[i] >= n_src || tgt_index[i] < 0 || tgt_index[i] >= n_tgt) {
            if (err) *err = "rigid_svd: correspondence index out of range"; return SCL_ERR_OUT_OF_RANGE;
        }
    if ((rc = upload(ws, B_SRC, src, (size_t)n_src * stride, stream, err))) return rc;
    if ((rc = upload(ws, B_TGT, tgt, (size_t)n_tgt * stride, stream, err))) return rc;
    if ((rc = upload(ws, B_SI, src_index, sizeof(int) * (size_t)n_corr, stream, err))) return rc;
    if ((rc = upload(ws, B_TI, tgt_index, sizeof(int) * (size_t)n_corr, stream, err))) return rc;
    if ((rc = ensure(ws, B_PART, sizeof(double) * kNSum * kRedBlocks, err))) return rc;
    if ((rc = ensure(ws, B_STATE, sizeof(IcpState), err))) return rc;
    if ((rc = pinned(ws, sizeof(IcpState), err))) return rc;
    IcpState *st = static_cast<IcpState *>(ws->buf[B_STATE]);
    int rb = (n_corr + 255) / 256; rb = rb < 1 ? 1 : (rb > kRedBlocks ? kRedBlocks : rb);
    hipLaunchKernelGGL(state_init_kernel, dim3(1), dim3(64), 0, stream, st);
    LAUNCH_REDUCE(rb, stream, (const float4 *)nullptr,
                  (const unsigned char *)ws->buf[B_SRC], (const unsigned char *)ws->buf[B_TGT], stride, n_corr,
                  (const int *)nullptr, (const float *)nullptr, 0.f, (const int *)ws->buf[B_SI], (const int *)ws->buf[B_TI], 1,
                  st, (double *)ws->buf[B_PART], 0);
    hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(64), 0, stream, st, (const double *)ws->buf[B_PART], rb, 1, 0, 0.0, 0.0);
    ICP_HIP(hipGetLastError());
    IcpState *h = static_cast<IcpState *>(ws->pinned);
    ICP_HIP(hipMemcpyAsync(h, st, sizeof(IcpState), hipMemcpyDeviceToHost, stream));
    ICP_HIP(hipStreamSynchronize(stream));
    std::memcpy(T, h->final_T, sizeof(float) * 16);
    return SCL_OK;
}

int icp_transform_cloud(IcpWorkspace *ws, hipStream_t stream, const void *in, int n, int stride,
                        const float T[16], void *out, std::string *err)
{
    int rc = check_cloud_args(n, 0, stride, err);
    if (rc) return rc;
    if ((rc = upload(ws, B_SRC, in, (size_t)n * stride, stream, err))) return rc;
    if ((rc = ensure(ws, B_OUT, (size_t)n * stride + 16, err))) return rc;
    if ((rc = upload(ws, B_BBOX, T, sizeof(float) * 16, stream, err))) return rc;
    const int pb = (n + 255) / 256 > 0 ? (n + 255) / 256 : 1;
    hipLaunchKernelGGL(raw_transform_kernel, dim3(pb), dim3(256), 0, stream, (const unsigned char *)ws->buf[B_SRC],
                       (unsigned char *)ws->buf[B_OUT], n, stride, (const float *)ws->buf[B_BBOX]);
    ICP_HIP(hipGetLastError());
    ICP_HIP(hipMemcpyAsync(out, ws->buf[B_OUT], (size_t)n * stride, hipMemcpyDeviceToHost, stream));
    ICP_HIP(hipStreamSynchronize(stream));
    return SCL_OK;
}

// RANSAC over explicit pairs already on the device (d_si/d_ti), leaves the mask in ws->buf[B_MASK]
static int ransac_device(IcpWorkspace *ws, hipStream_t stream, int stride, const int *d_si, const int *d_ti, int n_corr,
                         int max_iterations, double thr, unsigned long long seed, int *h_best2, double *h_T12, std::string *err)
{
    int rc;
    if ((rc = ensure(ws, B_MASK, sizeof(int) * (size_t)(n_corr + 1), err))) return rc;
    if ((rc = ensure(ws, B_HYP, sizeof(int) * (size_t)(max_iterations + 8) + 256, err))) return rc;
    if ((rc = pinned(ws, sizeof(IcpState) + 256, err))) return rc;
    int *counts = static_cast<int *>(ws->buf[B_HYP]);
    int *best = counts + max_iterations + 2;
    double *T12 = reinterpret_cast<double *>(reinterpret_cast<char *>(ws->buf[B_HYP]) + ((sizeof(int) * (size_t)(max_iterations + 8) + 63) / 64) * 64);
    const unsigned char *d_src = static_cast<const unsigned char *>(ws->buf[B_SRC]);
    const unsigned char *d_tgt = static_cast<const unsigned char *>(ws->buf[B_TGT]);
    const double thr2 = thr * thr;
    const int hb = (max_iterations + kHypPerBlock - 1) / kHypPerBlock;
    hipLaunchKernelGGL(ransac_score_kernel, dim3(hb), dim3(256), 0, stream, d_src, d_tgt, stride, d_si, d_ti, n_corr, seed,
                       max_iterations, thr2, counts);
    hipLaunchKernelGGL(ransac_pick_kernel, dim3(1), dim3(256), 0, stream, counts, max_iterations, best);
    int mb = (n_corr + 255) / 256; mb = mb < 1 ? 1 : (mb > 1024 ? 1024 : mb);
    hipLaunchKernelGGL(ransac_mask_kernel, dim3(mb), dim3(256), 0, stream, d_src, d_tgt, stride, d_si, d_ti, n_corr, seed,
                       best, thr2, (int *)ws->buf[B_MASK], T12);
    ICP_HIP(hipGetLastError());
    char *hp = static_cast<char *>(ws->pinned);
    ICP_HIP(hipMemcpyAsync(hp, best, sizeof(int) * 2, hipMemcpyDeviceToHost, stream));
    ICP_HIP(hipMemcpyAsync(hp + 64, T12, sizeof(double) * 12, hipMemcpyDeviceToHost, stream));
    ICP_HIP(hipStreamSynchronize(stream));
    std::memcpy(h_best2, hp, sizeof(int) * 2);
    if (h_T12) std::memcpy(h_T12, hp + 64, sizeof(double) * 12);
    return SCL_OK;
}

int icp_ransac(IcpWorkspace *ws, hipStream_t stream, const void *src, int n_src, const void *tgt, int n_tgt, int stride,
               const int *src_index, const int *tgt_index, int n_corr, int max_iterations, double inlier_threshold,
               unsigned long long seed, int *inlier_mask, int *n_inliers, int *best_hypothesis, float T_model[16],
               std::string *err)
{
    int rc = check_cloud_args(n_src, n_tgt, stride, err);
    if (rc) return rc;
    if (n_corr < 3 || max_iterations < 1 || max_iterations > (1 << 20)) { if (err) *err = "ransac: need >= 3 correspondences and 1..2^20 iterations"; return SCL_ERR_INVALID_ARG; }
    for (int i = 0; i < n_corr; ++i)
        if (src_index[i] < 0 || src_index[i] >= n_src || tgt_index[i] < 0 || tgt_index[i] >= n_tgt) {
            if (err) *err = "ransac: correspondence index out of range"; return SCL_ERR_OUT_OF_RANGE;
        }
    if ((rc = upload(ws, B_SRC, src, (size_t)n_src * stride, stream, err))) return rc;
    if ((rc = upload(ws, B_TGT, tgt, (size_t)n_tgt * stride, stream, err))) return rc;
    if ((rc = upload(ws, B_SI, src_index, sizeof(int) * (size_t)n_corr, stream, err))) return rc;
    if ((rc = upload(ws, B_TI, tgt_index, sizeof(int) * (size_t)n_corr, stream, err))) return rc;
    int best2[2]; double T12[12];
    if ((rc = ransac_device(ws, stream, stride, (const int *)ws->buf[B_SI], (const int *)ws->buf[B_TI], n_corr,
                            max_iterations, inlier_threshold, seed, best2, T12, err))) return rc;
    if (inlier_mask) {
        ICP_HIP(hipMemcpyAsync(inlier_mask, ws->buf[B_MASK], sizeof(int) * (size_t)n_corr, hipMemcpyDeviceToHost, stream));
        ICP_HIP(hipStreamSynchronize(stream));
    }
    if (n_inliers) *n_inliers = best2[1];
    if (best_hypothesis) *best_hypothesis = best2[0];
    if (T_model) {
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 4; ++b) T_model[a * 4 + b] = (float)T12[a * 4 + b];
        T_model[12] = T_model[13] = T_model[14] = 0.f; T_model[15] = 1.f;
    }
    return SCL_OK;
}

// geometricVerificationService core, DM.h:1211-1243, entirely on the device
int icp_geometric_verification(IcpWorkspace *ws, hipStream_t stream, int num_cu, const void *src, int n_src,
                               const void *tgt, int n_tgt, int stride, int ransac_iterations, double inlier_threshold,
                               double inlier_ratio, unsigned long long seed, float T[16], int *success, int *n_corr_out,
                               int *n_inliers_out, std::string *err)
{
    (void)num_cu;
    int rc = check_cloud_args(n_src, n_tgt, stride, err);
    if (rc) return rc;
    if (n_src >= 3 && n_tgt >= 1) {
        if ((rc = upload(ws, B_SRC, src, (size_t)n_src * stride, stream, err))) return rc;
        if ((rc = upload(ws, B_TGT, tgt, (size_t)n_tgt * stride, stream, err))) return rc;
    }
    return icp_geometric_verification_staged(ws, stream, n_src, n_tgt, stride, ransac_iterations, inlier_threshold,
                                             inlier_ratio, seed, T, success, n_corr_out, n_inliers_out, err);
}

// the same on clouds already staged in the workspace (icp_stage_cloud / icp_stage_cloud_host)
int icp_geometric_verification_staged(IcpWorkspace *ws, hipStream_t stream, int n_src, int n_tgt, int stride,
                                      int ransac_iterations, double inlier_threshold, double inlier_ratio,
                                      unsigned long long seed, float T[16], int *success, int *n_corr_out,
                                      int *n_inliers_out, std::string *err)
{
    int rc = check_cloud_args(n_src, n_tgt, stride, err);
    if (rc) return rc;
    for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.f : 0.f;
    if (success) *success = 0;
    if (n_corr_out) *n_corr_out = 0;
    if (n_inliers_out) *n_inliers_out = 0;
    if (n_src < 3 || n_tgt < 1) return SCL_OK;
    if (ransac_iterations < 1 || ransac_iterations > (1 << 20)) { if (err) *err = "ransac iterations out of range"; return SCL_ERR_INVALID_ARG; }
    if ((rc = ensure(ws, B_WORK, sizeof(float4) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_NNI, sizeof(int) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_NND, sizeof(float) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_SI, sizeof(int) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_TI, sizeof(int) * (size_t)(n_src + 1), err))) return rc;
    if ((rc = ensure(ws, B_PART, sizeof(double) * kNSum * kRedBlocks, err))) return rc;
    if ((rc = build_grid(ws, stream, n_tgt, stride, err))) return rc;
    IcpState *st = static_cast<IcpState *>(ws->buf[B_STATE]);
    const int pb = (n_src + 255) / 256;
    hipLaunchKernelGGL(work_init_kernel, dim3(pb), dim3(256), 0, stream, (const unsigned char *)ws->buf[B_SRC], n_src, stride,
                       (float4 *)ws->buf[B_WORK]);
    launch_nn_search(stream, (float4 *)ws->buf[B_WORK], n_src, st,
                       (const int *)ws->buf[B_CSTART], (const float4 *)ws->buf[B_TSORT], (int *)ws->buf[B_NNI],
                       (float *)ws->buf[B_NND], 0, -1, (const unsigned char *)ws->buf[B_TGT], stride, 0);   // DM.h:1211-1215
    hipLaunchKernelGGL(iota_pairs_kernel, dim3(pb), dim3(256), 0, stream, (const int *)ws->buf[B_NNI], n_src,
                       (int *)ws->buf[B_SI], (int *)ws->buf[B_TI]);
    int best2[2];
    if ((rc = ransac_device(ws, stream, stride, (const int *)ws->buf[B_SI], (const int *)ws->buf[B_TI], n_src,
                            ransac_iterations, inlier_threshold, seed, best2, nullptr, err))) return rc;   // DM.h:1218-1225
    const int n_inl = best2[1];
    if (n_corr_out) *n_corr_out = n_src;
    if (n_inliers_out) *n_inliers_out = n_inl;
    if (n_inl >= 3) {                                                                              // DM.h:1228-1230
        int rb = (n_src + 255) / 256; rb = rb < 1 ? 1 : (rb > kRedBlocks ? kRedBlocks : rb);
        hipLaunchKernelGGL(state_init_kernel, dim3(1), dim3(64), 0, stream, st);
        LAUNCH_REDUCE(rb, stream, (const float4 *)nullptr, (const unsigned char *)ws->buf[B_SRC],
                      (const unsigned char *)ws->buf[B_TGT], stride, n_src, (const int *)ws->buf[B_MASK],
                      (const float *)nullptr, 0.f, (const int *)ws->buf[B_SI], (const int *)ws->buf[B_TI], 1, st,
                      (double *)ws->buf[B_PART], 0);
        hipLaunchKernelGGL(icp_solve_kernel, dim3(1), dim3(64), 0, stream, st, (const double *)ws->buf[B_PART], rb, 1, 0, 0.0, 0.0);
        ICP_HIP(hipGetLastError());
        IcpState *h = static_cast<IcpState *>(ws->pinned);
        ICP_HIP(hipMemcpyAsync(h, st, sizeof(IcpState), hipMemcpyDeviceToHost, stream));
        ICP_HIP(hipStreamSynchronize(stream));
        std::memcpy(T, h->final_T, sizeof(float) * 16);
    }
    if (success) *success = !((double)n_inl < inlier_ratio * (double)n_src);                      // DM.h:1238
    return SCL_OK;
}

}  // namespace scl
