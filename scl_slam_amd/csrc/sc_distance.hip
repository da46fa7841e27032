// sc_distance.hip -- K1: column-shifted Scan Context distance over the keyframe DB.
//
// Replaces distanceBtnScanContext + fastAlignUsingVkey + distDirectSC + circshift
// (reference include/descriptor.h:1376-1395, 1491-1569).  One launch scores one
// query against n database slots and returns, per slot, the fp64 minimum distance
// and its ring shift, bit-identical to the sequential CPU evaluation:
//   - sector-key alignment: lane s owns shift s and runs the j = 0..S-1 sum of
//     (vq[j]-vk[(j-s) mod S])^2 in order, sqrt, then a strict-< first-wins arg-min
//     (D.h:1496-1508);
//   - shifted cosine distance: lane j owns candidate column j (its R values sit in
//     VGPRs, fetched as float4 from the tiled DB layout), and accumulates, for the
//     W = 2*SR+1 searched shifts at once, the dot product with query column
//     (j+shift) mod S in ring order r = 0..R-1 (the order of Eigen's column dot in
//     D.h:1528); products of two widened floats are exact in fp64, so fma == mul+add;
//   - the per-shift sum over sectors (D.h:1518-1532) is inherently sequential in the
//     reference; lane t of the first wave walks columns 0..S-1 for shift t;
//   - arg-min over shifts in ascending shift order with strict < (D.h:1552-1566).
//
// Mapping: persistent workgroups (<= 1 per CU), each keeps the query as fp64 in LDS
// (columns extended by W-1 so a lane's shift window is contiguous) and loops over
// groups of G candidates; a candidate is served by NW = ceil(S/64) waves.
// Bound: HBM (algorithmic 4*R*S + 8*S bytes per pair) on paper; the LDS read port
// (one ds_read_b64 per fp64 fma) is what limits v1 -- see DESIGN.md.
#include "device_common.hpp"
#include "kernels.hpp"

namespace scl {

namespace {

struct ScArgs {
    const float4 *desc;
    const double *vkey;
    const double *norm;
    const float4 *q_desc;
    const double *q_vkey;
    const double *q_norm;
    const int *cand;
    int slot_base;
    int n;
    int S;
    int SR;
    int NW;   // waves per candidate
    int G;    // candidates per workgroup iteration
    double *out_dist;
    int *out_shift;
};

__device__ __forceinline__ int wrap(int x, int S)
{   // x in (-S, 2S)
    x = x < 0 ? x + S : x;
    return x >= S ? x - S : x;
}

// MAXT bounds the workgroup: 512 threads = 2 waves/SIMD leaves 256 VGPRs per lane, 256 threads
// (one wave per SIMD) the whole 512-entry file.
template <int RG, int W, int MAXT>
__global__ __launch_bounds__(MAXT) void sc_distance_kernel(ScArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int S = a.S, SR = a.SR;
    constexpr int R4 = RG * 4;
    const int QS = S + W - 1;                 // extended query row
    const int gthreads = a.NW * kWave;
    const int g = threadIdx.x / gthreads;     // candidate group within the block
    const int j = threadIdx.x - g * gthreads; // lane within the candidate (column / shift)
    const int wv = j / kWave;
    const bool active = j < S;
    const int jj = active ? j : S - 1;

    double *Qd = smem;                        // [R4][QS]
    double *nq = Qd + R4 * QS;                // [S]
    double *vq = nq + S;                      // [S]
    const int gsz = 4 * S + W * S + 8;
    double *gb = vq + S + g * gsz;
    double *vk2 = gb;                         // [2S] candidate sector key, doubled
    double *nk2 = vk2 + 2 * S;                // [2S] candidate column norms, doubled
    double *simbuf = nk2 + 2 * S;             // [W][S]
    double *redv = simbuf + W * S;            // [4]
    int *redi = (int *)(redv + 4);            // [4] + [4]: the agreed alignment shift

    // ---- stage the query once per (persistent) workgroup ----------------------
    for (int idx = threadIdx.x; idx < RG * S; idx += blockDim.x) {
        const int rg = idx / S, c = idx - rg * S;
        const float4 v = a.q_desc[idx];
        double *dst = Qd + (rg * 4) * QS + c;
        dst[0] = (double)v.x; dst[QS] = (double)v.y; dst[2 * QS] = (double)v.z; dst[3 * QS] = (double)v.w;
        if (c < W - 1) {
            dst += S;
            dst[0] = (double)v.x; dst[QS] = (double)v.y; dst[2 * QS] = (double)v.z; dst[3 * QS] = (double)v.w;
        }
    }
    for (int c = threadIdx.x; c < S; c += blockDim.x) { nq[c] = a.q_norm[c]; vq[c] = a.q_vkey[c]; }

    const int per_iter = gridDim.x * a.G;
    const int iters = (a.n + per_iter - 1) / per_iter;
    for (int it = 0; it < iters; ++it) {
        const int ci = (it * gridDim.x + blockIdx.x) * a.G + g;
        int slot = -1;
        if (ci < a.n) slot = a.cand ? a.cand[ci] : a.slot_base + ci;
        const bool valid = slot >= 0;
        const size_t sl = valid ? (size_t)slot : 0;

        // ---- issue this candidate's loads: keys first, then the column tile ----
        const double vk = a.vkey[sl * S + jj];
        const double nk = a.norm[sl * S + jj];
        float4 kcol[RG];
        const float4 *kp = a.desc + sl * (size_t)(RG * S) + jj;
#pragma unroll
        for (int rg = 0; rg < RG; ++rg) kcol[rg] = kp[rg * S];

        __syncthreads();   // previous iteration is done with vk2/nk2/simbuf (and the query is staged)
        if (active) { vk2[j] = vk; vk2[j + S] = vk; nk2[j] = nk; nk2[j + S] = nk; }
        __syncthreads();

        // ---- phase A: fastAlignUsingVkey (D.h:1491-1511): lane = shift --------
        double best = __longlong_as_double(0x7ff0000000000000LL);
        int bshift = 0x7fffffff;
        if (active) {
            const double *vkp = vk2 + S - j;      // vkp[t] = vk[(t - j) mod S]
            double ss = 0.0;
#pragma unroll 4
            for (int t = 0; t < S; ++t) {
                const double d = vq[t] - vkp[t];
                ss = ss + d * d;
            }
            const double nrm = sqrt(ss);
            if (nrm < kBigDist) { best = nrm; bshift = j; }
        }
        wave_argmin(best, bshift);
        if ((j & (kWave - 1)) == 0) { redv[wv] = best; redi[wv] = bshift; }
        __syncthreads();
        int align = 0;
        {
            double bv = redv[0]; int bs = redi[0];
            for (int w = 1; w < a.NW; ++w) {
                const double ov = redv[w]; const int os = redi[w];
                const bool take = (ov < bv) | ((ov == bv) & (os < bs));
                bv = take ? ov : bv; bs = take ? os : bs;
            }
            align = bv < kBigDist ? bs : 0;       // nothing below 1e7 -> argmin stays 0 (D.h:1493)
        }

        // ---- phase B: W shifted column dot products, ring order ----------------
        const int base = wrap(jj + wrap(align - SR, S), S);   // query column for t = 0
        double acc[W];
#pragma unroll
        for (int t = 0; t < W; ++t) acc[t] = 0.0;
        {
            const double *qrow = Qd + base;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                const float kv[4] = {kcol[rg].x, kcol[rg].y, kcol[rg].z, kcol[rg].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double kd = (double)kv[i];
#pragma unroll
                    for (int t = 0; t < W; ++t) acc[t] = fma(kd, qrow[t], acc[t]);
                    qrow += QS;
                }
            }
        }
        // ---- phase C: cosine similarity per (shift, query column) -------------
        if (active) {
#pragma unroll
            for (int t = 0; t < W; ++t) {
                int c = base + t; c = c >= S ? c - S : c;
                simbuf[t * S + c] = acc[t] / (nq[c] * nk);
            }
        }
        __syncthreads();

        // ---- phase D: per-shift sequential sum over sectors (D.h:1518-1535) ---
        double dmin = __longlong_as_double(0x7ff0000000000000LL);
        int smin = 0x7fffffff;
        if (j < W) {
            const int st = wrap(wrap(align - SR, S) + j, S);   // the shift this lane owns
            const double *nkp = nk2 + S - st;                   // nkp[c] = nk[(c - st) mod S]
            const double *sb = simbuf + j * S;
            double sum = 0.0;
            int eff = 0;
#pragma unroll 4
            for (int c = 0; c < S; ++c) {
                const bool skip = (nq[c] == 0.0) | (nkp[c] == 0.0);   // D.h:1523
                const double v = sb[c];
                if (!skip) { sum = sum + v; eff = eff + 1; }
            }
            const double d = 1.0 - sum / (double)eff;                 // 0/0 -> NaN, never wins
            if (d < kBigDist) { dmin = d; smin = st; }
        }
        if (wv == 0) {
            wave_argmin(dmin, smin);
            if (j == 0 && ci < a.n) {
                const bool ok = valid && (dmin < kBigDist);
                a.out_dist[ci] = ok ? dmin : kBigDist;
                a.out_shift[ci] = ok ? smin : 0;
            }
        }
    }
}

// ---- generic fallback: any (R, S, SR).  One workgroup per candidate, one shift at
// a time, everything read from global/L2.  Correct, not fast; the three BASELINE
// grids (20x60, 64x120, 80x180 with search ratio 0.1) never take this path. --------
__global__ void sc_distance_generic_kernel(ScArgs a, int RG, int R)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int S = a.S, SR = a.SR;
    double *vq = smem, *nq = vq + S, *vk = nq + S, *nk = vk + S, *sim = nk + S, *nrm = sim + S;
    __shared__ int s_align;
    __shared__ double s_best;
    __shared__ int s_bshift;

    const int ci = blockIdx.x;
    const int slot = a.cand ? a.cand[ci] : a.slot_base + ci;
    if (slot < 0) {
        if (threadIdx.x == 0) { a.out_dist[ci] = kBigDist; a.out_shift[ci] = 0; }
        return;
    }
    const size_t sl = (size_t)slot;
    for (int c = threadIdx.x; c < S; c += blockDim.x) {
        vq[c] = a.q_vkey[c]; nq[c] = a.q_norm[c];
        vk[c] = a.vkey[sl * S + c]; nk[c] = a.norm[sl * S + c];
    }
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        double ss = 0.0;
        for (int t = 0; t < S; ++t) {
            int src = t - s; src = src < 0 ? src + S : src;
            const double d = vq[t] - vk[src];
            ss = ss + d * d;
        }
        nrm[s] = sqrt(ss);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int am = 0; double mn = kBigDist;
        for (int s = 0; s < S; ++s) if (nrm[s] < mn) { am = s; mn = nrm[s]; }
        s_align = am; s_best = kBigDist; s_bshift = 0;
    }
    __syncthreads();
    const int align = s_align;
    const float *qf = (const float *)a.q_desc;
    const float *kf = (const float *)(a.desc + sl * (size_t)(RG * S));
    // candidate shifts {align-SR..align+SR} mod S, evaluated in ASCENDING shift order
    // (std::sort at D.h:1552) with strict <; duplicates (2*SR+1 > S) cannot change the result.
    for (int sh = 0; sh < S; ++sh) {
        int dlt = sh - align; dlt = ((dlt % S) + S) % S;            // distance forward from align
        const bool in_space = (dlt <= SR) || (S - dlt <= SR);
        if (!in_space) continue;                                     // uniform across the block
        for (int c = threadIdx.x; c < S; c += blockDim.x) {
            int src = c - sh; src = src < 0 ? src + S : src;
            double dot = 0.0;
            for (int r = 0; r < R; ++r) {
                const double qv = (double)qf[((size_t)(r >> 2) * S + c) * 4 + (r & 3)];
                const double kv = (double)kf[((size_t)(r >> 2) * S + src) * 4 + (r & 3)];
                dot = fma(qv, kv, dot);
            }
            sim[c] = dot / (nq[c] * nk[src]);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double sum = 0.0; int eff = 0;
            for (int c = 0; c < S; ++c) {
                int src = c - sh; src = src < 0 ? src + S : src;
                if ((nq[c] == 0.0) | (nk[src] == 0.0)) continue;
                sum = sum + sim[c]; eff = eff + 1;
            }
            const double d = 1.0 - sum / (double)eff;
            if (d < s_best) { s_best = d; s_bshift = sh; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { a.out_dist[ci] = s_best; a.out_shift[ci] = s_bshift; }
}

// ---- arg-min over the distance vector (full-DB mode) --------------------------
__global__ void argmin_kernel(const double *dist, const int *shift, int n, double *out3)
{
    __shared__ double sv[16];
    __shared__ int si[16];
    double best = __longlong_as_double(0x7ff0000000000000LL);
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double d = dist[i];
        if (d < kBigDist && ((d < best) | ((d == best) & (i < bi)))) { best = d; bi = i; }
    }
    wave_argmin(best, bi);
    const int wv = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) { sv[wv] = best; si[wv] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = blockDim.x / kWave;
        for (int w = 1; w < nw; ++w)
            if ((sv[w] < best) | ((sv[w] == best) & (si[w] < bi))) { best = sv[w]; bi = si[w]; }
        const bool ok = best < kBigDist;
        out3[0] = ok ? best : kBigDist;
        out3[1] = ok ? (double)bi : -1.0;
        out3[2] = ok ? (double)shift[bi] : 0.0;
    }
}

template <int RG, int W, int MAXT>
hipError_t launch_fast(const ScArgs &args_in, int num_cu, hipStream_t stream)
{
    ScArgs a = args_in;
    const int S = a.S;
    const int QS = S + W - 1;
    const size_t fixed = (size_t)(RG * 4 * QS + 2 * S) * sizeof(double);
    const size_t per_group = (size_t)(4 * S + W * S + 8) * sizeof(double);
    const size_t lds_cap = 160 * 1024;
    int G = (int)((lds_cap - fixed) / per_group);
    const int max_g_threads = MAXT / (a.NW * kWave);
    if (G > max_g_threads) G = max_g_threads;
    if (G > 8) G = 8;
    if (G < 1) return hipErrorInvalidValue;
    while (G > 1 && a.n < num_cu * G) G >>= 1;   // few candidates: more workgroups, not fuller ones
    a.G = G;
    int blocks = (a.n + G - 1) / G;
    if (blocks > num_cu) blocks = num_cu;
    const size_t lds = fixed + per_group * G;
    static bool attr_set = false;   // per instantiation
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)sc_distance_kernel<RG, W, MAXT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((sc_distance_kernel<RG, W, MAXT>), dim3(blocks), dim3(G * a.NW * kWave), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_sc_distance(const DbView &db, const QueryView &q, const int *cand, int slot_base,
                              int n, int SR, double *out_dist, int *out_shift, int num_cu,
                              hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    ScArgs a;
    a.desc = db.desc; a.vkey = db.vkey; a.norm = db.norm;
    a.q_desc = q.desc; a.q_vkey = q.vkey; a.q_norm = q.norm;
    a.cand = cand; a.slot_base = slot_base; a.n = n; a.S = db.S; a.SR = SR;
    a.NW = (db.S + kWave - 1) / kWave; a.G = 1;
    a.out_dist = out_dist; a.out_shift = out_shift;
    const int W = 2 * SR + 1;
    if (db.RG == 5 && W == 7 && db.S >= W)   return launch_fast<5, 7, 512>(a, num_cu, stream);
    if (db.RG == 16 && W == 13 && db.S >= W) return launch_fast<16, 13, 512>(a, num_cu, stream);
    if (db.RG == 20 && W == 19 && db.S >= W && db.S <= 180) return launch_fast<20, 19, 256>(a, num_cu, stream);

    const size_t lds = (size_t)6 * db.S * sizeof(double);
    hipLaunchKernelGGL(sc_distance_generic_kernel, dim3(n), dim3(256), lds, stream, a, db.RG, db.R);
    return hipGetLastError();
}

hipError_t launch_argmin(const double *dist, const int *shift, int n, double *out3, hipStream_t stream)
{
    hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(1024), 0, stream, dist, shift, n, out3);
    return hipGetLastError();
}

}  // namespace scl
