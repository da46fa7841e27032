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
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <type_traits>
#include <cstdlib>
#include <vector>

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
    const int *n_dev;         // != nullptr: the number of candidates is read on the device (survivors of the screening pass)
    // survivors of the screening pass (sc_distance_survivors_kernel): every workgroup selects them itself from the
    // approximate distances of the range [slot_base, slot_base + range_n): approx[i] <= min + 2 eps (t_min = ordered
    // image of the minimum, re-armed by the last workgroup), slots written to `cand` in ascending order
    const float *approx; unsigned int *t_min; int range_n; float two_eps;
    const float *ring_d2;     // survivors pass: the ring-key metric of the range (from the screening pass), for the top-k
    int ring_from_keys;       // survivors pass: ... formed here from the tiled ring keys instead (the screening pass left it out)
    int *sel_topk_idx; float *sel_topk_d2; int sel_topk_k; float sel_exclude_eps;
    unsigned long long *surv_stats;   // survivors pass, optional: survivor count statistics (scl_survivor_stats)
    int S;
    int SR;
    int NW;   // waves per candidate
    int G;    // candidates per workgroup iteration
    double *out_dist;
    int *out_shift;
    // optional fused ring-key scan (full-DB mode): squared ring-key distance of every scored slot
    const float4 *rkey4; int rk_cap; const float *q_rkey; float *out_d2;
    // optional fused epilogue (full-DB mode): per-workgroup partials, last workgroup reduces them
    unsigned long long *blk_part; unsigned int *done_counter; double *out3; int *topk_idx; float *topk_d2;
    int topk_k; float exclude_eps;
    int align_filter;             // 1: fp32 correlation filter in front of the exact alignment (SCL_ALIGN_FILTER=0 disables)
    unsigned long long *stamps;   // diagnostic only (SCL_STAMP=1): per-wave phase cycle sums
    int ablate;   // diagnostic only (SCL_ABLATE): bit0 skip alignment loop, bit1 skip ring dots, bit2 skip sector sums
};

// Kernel argument of the wave kernel: up to kMaxQueryBatch complete argument sets, one per query of the launch.
// Workgroups [qi*nb, (qi+1)*nb) serve query qi and read q[qi] straight from the kernel-argument segment (scalar,
// invariant loads the compiler re-issues on demand -- nothing per-query has to stay live in registers).
struct ScBatchArgs {
    ScArgs q[kMaxQueryBatch];
    int nq, nb;
};

__device__ __forceinline__ int wrap(int x, int S)
{   // x in (-S, 2S)
    x = x < 0 ? x + S : x;
    return x >= S ? x - S : x;
}

// MAXT bounds the workgroup: 512 threads = 2 waves/SIMD leaves 256 VGPRs per lane, 256 threads
// (one wave per SIMD) the whole 512-entry file.
template <int RG, int W, int MAXT>
__device__ __forceinline__ void sc_distance_body(const ScArgs &a, double *smem)
{
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

    // a survivor list shorter than the grid (the usual case: one to five survivors, 16 workgroups per query): the workgroups
    // without a candidate leave before they stage 126 KB of query
    if (a.n_dev && (int)blockIdx.x * a.G >= *a.n_dev) return;
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
    const int n_cand = a.n_dev ? *a.n_dev : a.n;          // survivors of the screening pass: counted on the device
    const int iters = (n_cand + per_iter - 1) / per_iter;
    for (int it = 0; it < iters; ++it) {
        const int ci = (it * gridDim.x + blockIdx.x) * a.G + g;
        int slot = -1;
        if (ci < n_cand) slot = a.cand ? a.cand[ci] : a.slot_base + ci;
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
            if (j == 0 && ci < n_cand) {
                const bool ok = valid && (dmin < kBigDist);
                a.out_dist[ci] = ok ? dmin : kBigDist;
                a.out_shift[ci] = ok ? smin : 0;
            }
        }
    }
}

template <int RG, int W, int MAXT>
__global__ __launch_bounds__(MAXT) void sc_distance_kernel(ScArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    sc_distance_body<RG, W, MAXT>(a, smem);
}

// the same program for the survivor lists of up to kWideLegacyBatch queries in one launch: blockIdx.y = query
constexpr int kWideLegacyBatch = 12;         // (twelve argument sets are what the launch's argument block holds)
struct ScArgsBatch { ScArgs q[kWideLegacyBatch]; };
static_assert(sizeof(ScArgsBatch) <= 4096, "kernel arguments end at 4 KB");
template <int RG, int W, int MAXT>
__global__ __launch_bounds__(MAXT) void sc_distance_batch4_kernel(ScArgsBatch ab)
{
    extern __shared__ __attribute__((aligned(16))) double smem_b[];
    sc_distance_body<RG, W, MAXT>(ab.q[blockIdx.y], smem_b);
}

// =================================================================================
// v2: ONE WAVE PER CANDIDATE, two candidate columns per lane.
//
// Why: in the kernel above every fp64 fma needs its own 8-byte LDS operand, so the LDS
// port (256 B/clk/CU) and not the fp64 pipe sets the pace, and three workgroup barriers
// per candidate serialise the phases.  Here lane l owns QUERY columns 2l and 2l+1 and the
// candidate rotates under it: with b the first searched shift, the lane takes candidate
// columns (2l - b) mod S and +1.  The two columns' shift windows overlap in all but one
// query column, so W+1 query values -- always columns 2l .. 2l+W, a fixed, 16-byte aligned,
// never wrapping address: (W+1)/2 conflict-free ds_read_b128 -- feed 2*W fmas (0.54 LDS
// bytes per fma-byte instead of 1).  All per-candidate scratch (doubled sector key,
// similarity rows) is private to the wave: no workgroup barrier after the query has been
// staged, waves drift apart and overlap each other's latency-bound phases.
// Arithmetic order is unchanged (ring order dots, sector order sums): results stay
// bit-identical to the CPU checker.
// =================================================================================
__device__ __forceinline__ void wave_fence()
{   // LDS operations of one wave execute in issue order; only the compiler must not reorder
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Compiler-level ordering point for the ring pipeline of the wave kernel.  An empty asm
// with a memory clobber orders the LDS reads, and taking every accumulator as a read-write
// operand orders the fmas: without the latter SelectionDAG is free to emit all fmas of a
// block after all of its loads (it does), which keeps every ring's window live and spills.
__device__ __forceinline__ void pin_ring(double (&a)[7], double (&b)[7])
{
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                      "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6])
                 :: "memory");
}
__device__ __forceinline__ void pin_ring(double (&a)[13], double (&b)[13])
{
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]),
                      "+v"(a[7]), "+v"(a[8]), "+v"(a[9]), "+v"(a[10]), "+v"(a[11]), "+v"(a[12]),
                      "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]),
                      "+v"(b[7]), "+v"(b[8]), "+v"(b[9]), "+v"(b[10]), "+v"(b[11]), "+v"(b[12])
                 :: "memory");
}

__device__ __forceinline__ void pin3(double &a, double &b, double &c)
{
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c) :: "memory");
}
// a / b for operands whose quotient needs no exponent scaling: v_rcp_f64, two Newton steps on the
// reciprocal, one on the quotient -- the exact instruction sequence of the compiler's correctly rounded
// division (v_div_scale / v_div_fmas / v_div_fixup reduce to the identity in that range).
__device__ __forceinline__ double quot_core(double a, double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    e = fma(-b, y, 1.0);
    y = fma(y, e, y);
    const double q = a * y;
    const double r = fma(-b, q, a);
    return fma(r, y, q);
}
__device__ __forceinline__ bool is_finite(double x) { return fabs(x) <= 0x1.fffffffffffffp+1023; }
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void pin_f2(f2 &a, f2 &b)
{
    asm volatile("" : "+v"(a), "+v"(b) :: "memory");
}
__device__ __forceinline__ void pin1(double &a)
{
    asm volatile("" : "+v"(a) :: "memory");
}

// The whole workgroup program of the wave kernel: `a` = the argument set of this workgroup's query, bid / nbk = its
// index among / the number of workgroups that serve the query.
template <int RG, int W, int CH, int S, int MAXT, bool STAMP>
// sbid / snb: this workgroup's share of the candidates is part sbid of snb (normally bid of nbk; the survivors' pass keeps one
// workgroup free of candidates: it forms the ring-key top-k meanwhile and only takes part in the tail)
__device__ __forceinline__ void sc_wave_body(const ScArgs &a, const int bid, const int nbk, const int sbid, const int snb)
{
    unsigned long long st_t = 0, st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_real0 = 0, st_cyc0 = 0;
    auto stamp = [&]() -> unsigned long long {
        if (!STAMP) { __builtin_amdgcn_sched_barrier(0); return 0ull; }   // phase boundaries stay scheduling fences
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    if (STAMP) { st_cyc0 = stamp(); st_real0 = __builtin_amdgcn_s_memrealtime(); }

    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int R4 = RG * 4;
    static_assert(W % 2 == 1, "2*SR+1 shifts");
    constexpr int HSH = (W + 1) / 2;           // shifts summed per phase-D pass (second pass: W - HSH)
    constexpr int NQ = (W + 1) / 2;            // ds_read_b128 per ring: W+1 query values feed 2*W fmas
    static_assert(RG % CH == 0, "chunk must divide the ring groups");
    static_assert(S % 2 == 0 && S / 2 <= kWave && S >= W + 1, "two sectors per lane, one wave per candidate");
    const int SR = a.SR;
    constexpr int L = S >> 1;                  // active lanes
    constexpr int QS = S + W + 1;              // extended query row (even)
    constexpr int SB = S + 2;                      // similarity row stride: even (b128 rows), rows on distinct banks
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;

    double *Qd = smem;                         // [R4][QS]
    double *nqe = Qd + R4 * QS;                // [QS] query column norms, extended
    double *vq = nqe + QS;                     // [S]  query sector key (also the over-read pad of the ring pipeline)
    // per-wave scratch: the doubled sector key (alignment of candidate i+1) and the similarity rows
    // (phases C/D of candidate i) are never live together -> they share the space
    constexpr int wsz = (2 * S > HSH * SB) ? 2 * S : HSH * SB;
    // fp32 copies of the query's sector key for the alignment filter: vqf0[t] = q[t], vqf1[t] = q[(t+1) mod S]
    float *vqf0 = reinterpret_cast<float *>(vq + S);
    float *vqf1 = vqf0 + S;
    double *wbase = vq + 2 * S;                // per-wave scratch starts here
    double *vk2 = wbase + wave * wsz;          // [2S] candidate sector key, doubled
    double *simbuf = vk2;                      // [HSH][SB]
    // fp32 doubled candidate key, twice: pf[c][t + 2c] = p[t mod S] (c = 0, 1), so that the window of every lane
    // starts on a 16-byte boundary in one of the two copies; lives behind vk2 inside the wave's scratch
    // copy 1 starts PFS floats after copy 0.  Even lanes read copy 0 and odd lanes copy 1, both in 16-byte steps
    // of -16 B per lane pair; with the copies 128 B apart modulo 256 the two streams fall on complementary banks
    // inside every ds_read_b128 lane group (the first fit, 2S + 4 floats, made them collide: 1.5 extra LDS cycles
    // per window read of the filter).  The small grid has no room for the padding.
    constexpr int PFS = (2 * S + 4 <= 288 && 2 * S + (288 + 2 * S + 4 + 1) / 2 <= wsz) ? 288 : 2 * S + 4;
    static_assert(2 * S + (PFS + 2 * S + 4 + 1) / 2 <= wsz, "alignment filter scratch must fit the wave's scratch");
    float *pf = reinterpret_cast<float *>(vk2 + 2 * S);
    int *next_ticket = reinterpret_cast<int *>(wbase + nwaves * wsz);    // workgroup-wide candidate dispenser

    // Candidates of this workgroup: a contiguous range, handed out wave by wave through an LDS counter (below).
    // A workgroup without candidates (few survivors of the screening pass) skips the query staging altogether.
    const int n_cand = a.n_dev ? *a.n_dev : a.n;
    const int c_lo = (int)(((long long)sbid * n_cand) / snb);
    const int c_hi = (int)(((long long)(sbid + 1) * n_cand) / snb);
    for (int idx = threadIdx.x; idx < (c_lo < c_hi ? RG * S : 0); idx += blockDim.x) {
        const int rg = idx / S, c = idx - rg * S;
        const float4 v = a.q_desc[idx];
        double *dst = Qd + (rg * 4) * QS + c;
        dst[0] = (double)v.x; dst[QS] = (double)v.y; dst[2 * QS] = (double)v.z; dst[3 * QS] = (double)v.w;
        if (c < W + 1) {
            dst += S;
            dst[0] = (double)v.x; dst[QS] = (double)v.y; dst[2 * QS] = (double)v.z; dst[3 * QS] = (double)v.w;
        }
    }
    for (int c = threadIdx.x; c < (c_lo < c_hi ? S : 0); c += blockDim.x) {
        const double nv = a.q_norm[c];
        nqe[c] = nv;
        if (c < W + 1) nqe[c + S] = nv;
        const double kv = a.q_vkey[c];
        vq[c] = kv;
        vqf0[c] = (float)kv;
        vqf1[c == 0 ? S - 1 : c - 1] = (float)kv;
    }
    // Candidates of this workgroup: a contiguous range, handed out wave by wave through an LDS
    // counter.  The two waves that share a SIMD are not served equally (the older one wins the
    // VALU arbitration), so a static split leaves waves 4-7 ~25 % behind; first come, first served
    // evens it out.
    if (threadIdx.x == 0) *next_ticket = c_lo + nwaves;
    __syncthreads();                           // the only workgroup barrier

    const bool active = lane < L;
    const int ll = active ? lane : L - 1;
    const int j0 = 2 * ll;                     // first of the lane's two QUERY-side columns
    const double kInf = __longlong_as_double(0x7ff0000000000000LL);
    // Lane l always works on the query window that starts at column 2l; the candidate is what
    // rotates: after the alignment has fixed the first evaluated shift s_start, the lane fetches
    // candidate columns (2l - s_start) mod S and +1.  The window address is the same for every
    // candidate and never wraps, so the b128 window reads are bank-conflict free.
    // (the idle lanes 60..63 read at their own 2 * lane, not at lane L-1's address: four lanes on lane 59's
    // address put a second access on banks 44..47 of their ds_read_b128 lane group -- one extra LDS cycle on
    // every window read, 20 % of the array's cycles in phase B.  What they read and accumulate is never stored.)
    const double2 *qwin = reinterpret_cast<const double2 *>(Qd + 2 * lane);
    // fused ring-key scan: lane g < RG owns ring group g of the query key
    const bool rk_on = a.out_d2 != nullptr;
    float4 qrk = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rk_on && lane < RG) qrk = *reinterpret_cast<const float4 *>(a.q_rkey + 4 * lane);
    constexpr int qstep = QS / 2;

    int ci = c_lo + wave;
    const bool have_work = ci < c_hi;
    // alignment filter: |q|^2 of the query's sector key in fp32 (any summation order: the bound has 5 % of slack)
    const bool use_filter = a.align_filter != 0;
    float qn2 = 0.f;
    if (use_filter) {
        const f2 qv = *reinterpret_cast<const f2 *>(vqf0 + j0);
        qn2 = wave_sum_f32_dpp(active ? qv.x * qv.x + qv.y * qv.y : 0.f);
    }
    bool q_finite;
    {
        bool f = true;
#pragma unroll
        for (int i = 0; i <= W; ++i) f &= is_finite(nqe[j0 + i]);
        q_finite = __all(f);
    }

    // running results of this wave (wave-uniform): best (distance, position) and the KT nearest ring keys
    constexpr int KT = kTailTop;               // ring-key candidates tracked by the fused epilogue
    constexpr unsigned long long kNone = ~0ull;
    double w_best = kInf;
    int w_bidx = 0x7fffffff, w_bshift = 0;
    unsigned long long w_top[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) w_top[t] = kNone;

    // ---- per-candidate state of the software pipeline --------------------------------
    int slot = have_work ? (a.cand ? a.cand[ci] : a.slot_base + ci) : -1;
    int s_start = 0;
    const float4 *kp0 = a.desc, *kp1 = a.desc;
    double2 nk = make_double2(0.0, 0.0);
    float4 k0[CH], k1[CH];

    // alignment of one candidate (fastAlignUsingVkey, D.h:1491-1511) + issue of its first loads
    auto align_and_fetch = [&](int sl_i, const double2 vk, int &o_s_start) {
        wave_fence();
        // lanes >= L mirror lane L-1 (same addresses, same values): no divergent stores needed
        *reinterpret_cast<double2 *>(vk2 + j0) = vk;
        *reinterpret_cast<double2 *>(vk2 + j0 + S) = vk;
        int filtered = -1;                                   // >= 0: the alignment, decided by the filter below
        if (use_filter) {
            // ---- alignment filter -------------------------------------------------------------------
            // argmin_s |q - shift(k, s)|  =  argmax_s c_s,  c_s = sum_t q_t k_(t-s)  (|shift(k, s)| does not depend
            // on s).  c_s is evaluated in fp32 (packed fmas) for every shift; the absolute error of each value is
            // at most eps = 65 * 2^-24 * |q| |k| (60-term chains per accumulator, input conversion, final add;
            // Cauchy-Schwarz on sum |q_t| |k_u|).  Every shift whose c lies more than 4 eps below the maximum is
            // provably not the reference's minimum (2 eps of margin over the error, far above the fp64 rounding
            // of the reference's own sums); if exactly one shift remains it is the answer, otherwise -- near
            // ties, non-finite or huge keys -- the exact fp64 evaluation below decides, as it always did.
            const f2 kf = f2{(float)vk.x, (float)vk.y};
            float *pf1 = pf + PFS;
            *reinterpret_cast<f2 *>(pf + j0) = kf;          *reinterpret_cast<f2 *>(pf + j0 + S) = kf;
            *reinterpret_cast<f2 *>(pf1 + j0 + 2) = kf;     *reinterpret_cast<f2 *>(pf1 + j0 + S + 2) = kf;
            wave_fence();
            const int E = S - 2 * lane;                       // doubled-key index of sector 0 at shift 2l (idle lanes 60..63:
                                                              // their own, unused address -- lane 59's would conflict)
            const float4 *pw = reinterpret_cast<const float4 *>((lane & 1) ? pf1 + E + 2 : pf + E);
            const float4 *q0w = reinterpret_cast<const float4 *>(vqf0);
            const float4 *q1w = reinterpret_cast<const float4 *>(vqf1);
            constexpr int NG = S / 4, FB = 3;                 // sector groups of four, FB groups per batch
            static_assert(S % 4 == 0 && NG % FB == 0, "filter batches must tile the sectors");
            f2 ce = f2{0.f, 0.f}, co = f2{0.f, 0.f};
            float4 pb[2][FB], qa[2][FB], qb[2][FB];
#pragma unroll
            for (int v = 0; v < FB; ++v) { pb[0][v] = pw[v]; qa[0][v] = q0w[v]; qb[0][v] = q1w[v]; }
            if (!(a.ablate & 1)) {
#pragma unroll
            for (int bt = 0; bt < NG / FB; ++bt) {
                if (bt + 1 < NG / FB) {
#pragma unroll
                    for (int v = 0; v < FB; ++v) {
                        pb[(bt + 1) & 1][v] = pw[(bt + 1) * FB + v];
                        qa[(bt + 1) & 1][v] = q0w[(bt + 1) * FB + v];
                        qb[(bt + 1) & 1][v] = q1w[(bt + 1) * FB + v];
                    }
                }
                pin_f2(ce, co);
#pragma unroll
                for (int v = 0; v < FB; ++v) {
                    const float4 pv = pb[bt & 1][v], a0 = qa[bt & 1][v], a1 = qb[bt & 1][v];
                    ce = __builtin_elementwise_fma(f2{a0.x, a0.y}, f2{pv.x, pv.y}, ce);
                    ce = __builtin_elementwise_fma(f2{a0.z, a0.w}, f2{pv.z, pv.w}, ce);
                    co = __builtin_elementwise_fma(f2{a1.x, a1.y}, f2{pv.x, pv.y}, co);
                    co = __builtin_elementwise_fma(f2{a1.z, a1.w}, f2{pv.z, pv.w}, co);
                }
                pin_f2(ce, co);
            }
            }
            const float c_even = ce.x + ce.y, c_odd = co.x + co.y;
            const float kn2 = wave_sum_f32_dpp(active ? kf.x * kf.x + kf.y * kf.y : 0.f);
            const float nsum = sqrtf(qn2) + sqrtf(kn2);
            const float eps = 4.07e-6f * sqrtf(qn2) * sqrtf(kn2) + 1e-12f * (qn2 + kn2);   // 65 * 2^-24 * 1.05
            const float cmax = wave_max_f32_dpp(active ? fmaxf(c_even, c_odd) : -3.0e38f);
            const float cut = cmax - 4.0f * eps;
            const bool fe = active && !(c_even < cut), fo = active && !(c_odd < cut);      // NaN counts as "may be the minimum"
            const unsigned long long me = __builtin_amdgcn_ballot_w64(fe), mo = __builtin_amdgcn_ballot_w64(fo);
            const bool sane = (qn2 < 3.0e38f) && (kn2 < 3.0e38f) && (nsum * nsum < 0.9e14f);   // every norm finite and below the 1e7 start value
            if (sane && __popcll(me) + __popcll(mo) == 1)
                filtered = me ? 2 * (__ffsll((long long)me) - 1) : 2 * (__ffsll((long long)mo) - 1) + 1;
        }
        wave_fence();
        double best = kInf;
        int bshift = 0x7fffffff;
        if (filtered < 0) {
            // lane owns shifts 2l and 2l+1:  p[t] = vk[(t - 2l) mod S]; shift 2l+1 reuses p[t-1].
            // All operands come from LDS (in-order returns => counted waits), one batch of
            // 8 sectors is in flight while the previous one is being consumed.
            const double *p = vk2 + S - j0;
            const double2 *pp = reinterpret_cast<const double2 *>(p);
            double prev = p[-1];
            double ss0 = 0.0, ss1 = 0.0;
            constexpr int npair = S >> 1;
            // Batches of BT sector pairs: the LDS reads of batch b+1 are issued before the arithmetic
            // of batch b (pin3 fixes that order for LLVM), so the two dependent add chains never wait
            // on LDS.
            constexpr int BT = (MAXT > 512) ? 3 : 5;
            static_assert(npair % BT == 0, "alignment batches must tile the sector pairs");
            if (!(a.ablate & 1)) {
                // query key pairs: every lane reads the same 16 bytes (LDS broadcast, one array cycle);
                // cheaper than four v_readlane per pair, which stall the fp64 pipe on the SGPR hazard
                const double2 *qq = reinterpret_cast<const double2 *>(vq);
                double2 pb[2][BT], qb[2][BT];
#pragma unroll
                for (int v = 0; v < BT; ++v) { pb[0][v] = pp[v]; qb[0][v] = qq[v]; }
#pragma unroll
                for (int bt = 0; bt < npair / BT; ++bt) {
                    if (bt + 1 < npair / BT) {
#pragma unroll
                        for (int v = 0; v < BT; ++v) { pb[(bt + 1) & 1][v] = pp[(bt + 1) * BT + v]; qb[(bt + 1) & 1][v] = qq[(bt + 1) * BT + v]; }
                    }
                    pin3(ss0, ss1, prev);
#pragma unroll
                    for (int v = 0; v < BT; ++v) {
                        const double2 pv = pb[bt & 1][v];
                        const double qx = qb[bt & 1][v].x, qy = qb[bt & 1][v].y;
                        const double d0 = qx - pv.x, d1 = qx - prev;
                        ss0 = ss0 + d0 * d0;
                        ss1 = ss1 + d1 * d1;
                        const double e0 = qy - pv.y, e1 = qy - pv.x;
                        ss0 = ss0 + e0 * e0;
                        ss1 = ss1 + e1 * e1;
                        prev = pv.y;
                    }
                    pin3(ss0, ss1, prev);
                }
            }
            const double n0 = sqrt(ss0), n1 = sqrt(ss1);
            if (active && n0 < kBigDist) { best = n0; bshift = j0; }
            if (active && n1 < kBigDist && n1 < best) { best = n1; bshift = j0 + 1; }   // ties keep the lower shift
        }
        int align;
        if (filtered >= 0) {
            align = filtered;
        } else {
            wave_argmin_dpp(best, bshift);
            align = __builtin_amdgcn_readfirstlane(best < kBigDist ? bshift : 0);
        }
        // first shift of the reference's search space; the W evaluated shifts are b .. b+W-1 (mod S)
        o_s_start = wrap(align - SR, S);
        // rotation: lane l keeps query columns 2l, 2l+1 and takes the candidate columns that meet them at
        // shift b; the pair may straddle the sector wrap (kc = S-1), hence two pointers
        const int kc = wrap(j0 - o_s_start, S);
        const int kc1 = kc + 1 == S ? 0 : kc + 1;
        const size_t sl = (size_t)sl_i;
        nk = make_double2(a.norm[sl * S + kc], a.norm[sl * S + kc1]);
        kp0 = a.desc + sl * (size_t)(RG * S) + kc;
        kp1 = a.desc + sl * (size_t)(RG * S) + kc1;
#pragma unroll
        for (int u = 0; u < CH; ++u) { k0[u] = kp0[u * S]; k1[u] = kp1[u * S]; }
    };

    if (slot >= 0) {
        const double2 vk = *reinterpret_cast<const double2 *>(a.vkey + (size_t)slot * S + j0);
        st_t = stamp();
        align_and_fetch(slot, vk, s_start);
        st_a += stamp() - st_t;
    }

    while (have_work) {
        int ticket = 0;
        if (lane == 0) ticket = atomicAdd(next_ticket, 1);
        const int ci_next = __builtin_amdgcn_readfirstlane(ticket);
        int slot_next = -1;
        if (ci_next < c_hi) slot_next = a.cand ? a.cand[ci_next] : a.slot_base + ci_next;
        double2 vk_next = make_double2(0.0, 0.0);
        if (slot_next >= 0) vk_next = *reinterpret_cast<const double2 *>(a.vkey + (size_t)slot_next * S + j0);

        // the candidate's ring key travels under phase B (consumed by the fused ring-key metric after it)
        float4 rk_cand = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot >= 0 && rk_on && lane < RG) rk_cand = a.rkey4[(size_t)lane * a.rk_cap + slot];

        double acc0[W], acc1[W];
        const double2 nk_cur = nk;
        const int s_start_cur = s_start;
        st_t = stamp();
        if (slot >= 0) {
            // ---- phase B: 2 x W shifted column dots in ring order ---------------------
#pragma unroll
            for (int t = 0; t < W; ++t) { acc0[t] = 0.0; acc1[t] = 0.0; }
            // Explicit two-stage pipeline over rings: the query window of ring r+1 is requested
            // before the fmas of ring r; pin_ring() keeps LLVM from batching every ring's LDS
            // window ahead of all fmas of the block (which it otherwise does -- and spills).
            const double2 *qp = qwin;
            double2 qn[NQ];
#pragma unroll
            for (int v = 0; v < NQ; ++v) qn[v] = qp[v];
            // The candidate's two columns stream through a ring of CH ring-groups of registers:
            // as soon as a group has been widened to fp64 its slot is refilled with the group CH
            // ahead, so ~CH*32 B per lane stay in flight all through the phase (HBM latency cover).
            const int nch = (a.ablate & 2) ? 0 : RG / CH;
#pragma unroll 1
            for (int ch = 0; ch < nch; ++ch) {
                const bool more = ch + 1 < RG / CH;
                const float4 *np0 = kp0 + (size_t)(ch + 1) * CH * S, *np1 = kp1 + (size_t)(ch + 1) * CH * S;
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float a0[4] = {k0[u].x, k0[u].y, k0[u].z, k0[u].w};
                    const float a1[4] = {k1[u].x, k1[u].y, k1[u].z, k1[u].w};
                    if (more) { k0[u] = np0[u * S]; k1[u] = np1[u * S]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double q[W + 1];
#pragma unroll
                        for (int v = 0; v < NQ; ++v) { q[2 * v] = qn[v].x; q[2 * v + 1] = qn[v].y; }
                        pin_ring(acc0, acc1);
                        qp += qstep;                        // the last ring over-reads one row: it lands in
                        if (MAXT <= 512) {                  // nqe/vq (inside the allocation), values unused
#pragma unroll
                            for (int v = 0; v < NQ; ++v) qn[v] = qp[v];
                        }
                        const double kd0 = (double)a0[i], kd1 = (double)a1[i];
#pragma unroll
                        for (int t = 0; t < W; ++t) {
                            acc0[t] = fma(kd0, q[t], acc0[t]);
                            acc1[t] = fma(kd1, q[t + 1], acc1[t]);
                        }
                        if (MAXT > 512) {                   // 3 waves/SIMD build: one window buffer, the partner
                            pin_ring(acc0, acc1);           // waves cover the LDS latency instead of a prefetch
#pragma unroll
                            for (int v = 0; v < NQ; ++v) qn[v] = qp[v];
                        }
                    }
                }
            }
        }

        { const unsigned long long t1 = stamp(); st_b += t1 - st_t; st_t = t1; }
        // ---- next candidate: alignment + first loads, hidden under this one's phases C/D ----
        if (slot_next >= 0) align_and_fetch(slot_next, vk_next, s_start);
        { const unsigned long long t1 = stamp(); st_a += t1 - st_t; st_t = t1; }

        // ---- phases C/D in two passes of HSH shifts ------------------------------
        if (slot >= 0 && rk_on) {
            // nanoflann's metric (nanoflann.hpp:383-408): four dimensions per step, fp32, left to right,
            // groups accumulated in order -> lane g forms its group, lane-broadcasts feed the serial sum
            float grp = 0.0f;
            if (lane < RG) {
                const float4 b = rk_cand;
                const float d0 = qrk.x - b.x, d1 = qrk.y - b.y, d2 = qrk.z - b.z, d3 = qrk.w - b.w;
                grp = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
            float result = 0.0f;
#pragma unroll
            for (int g = 0; g < RG; ++g) result += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(grp), g));
            if (lane == 0) a.out_d2[ci] = result;
            const int rbits = __builtin_amdgcn_readfirstlane(__float_as_int(result));   // scalar from here on
            const float rs = __int_as_float(rbits);
            const bool excluded = (a.exclude_eps > 0.0f) && (rs <= a.exclude_eps);
            if (!excluded && (rs < 3.402823466e+38f)) {
                unsigned long long key = ((unsigned long long)(unsigned)rbits << 32) | (unsigned)ci;
#pragma unroll
                for (int t = 0; t < KT; ++t) {                   // sorted insert (ascending), wave-uniform
                    const unsigned long long lo_k = key < w_top[t] ? key : w_top[t];
                    const unsigned long long hi_k = key < w_top[t] ? w_top[t] : key;
                    w_top[t] = lo_k; key = hi_k;
                }
            }
        }
        if (slot >= 0) {
            double dmin = kInf;
            int smin = 0x7fffffff;
            const bool fast_quot = q_finite && __all((int)is_finite(nk_cur.x) & (int)is_finite(nk_cur.y));
            // Two passes (HSH, then W - HSH shifts), each a compile-time instance so the accumulators keep
            // static register indices; a scheduling fence between them keeps LLVM from interleaving both
            // (it would, and spill).  Within a pass: all query norms first, then every quotient (independent
            // division chains the scheduler can interleave), then the stores / effective-sector counts.
            // Similarity rows are stored de-interleaved (even sectors, then odd sectors) so the 8-byte
            // stores of consecutive lanes hit consecutive banks.
            auto cd_pass = [&](auto htag) {
                constexpr int h = decltype(htag)::value;
                constexpr int T0 = h * HSH;
                constexpr int NT = h == 0 ? HSH : W - HSH;
                wave_fence();
                int eff = 0;
                constexpr int HALF = S / 2;
                if (MAXT <= 512) {
                    // 2 waves/SIMD build: everything of the pass in flight at once (ILP hides the division latency)
                    double nqv[NT + 1];
#pragma unroll
                    for (int i = 0; i <= NT; ++i) nqv[i] = nqe[j0 + T0 + i];
                    double s0[NT], s1[NT];
                    bool ok0[NT], ok1[NT];
                    // Descriptor values are widened floats, so for finite inputs every counted quotient has
                    // dot in {0} u [2^-298, 2^262] over norm product in [2^-298, 2^262]: the scaling steps of the
                    // compiler's division (v_div_scale / v_div_fmas / v_div_fixup) are the identity there and
                    // quot_core() -- the same refinement without them -- returns the same bits with three
                    // instructions less and no serialisation through VCC.  Non-finite norms (query: checked
                    // once per wave, candidate: per candidate) take the general division.
                    if (fast_quot) {
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt) {
                            ok0[tt] = !((nqv[tt] == 0.0) | (nk_cur.x == 0.0));              // D.h:1523
                            ok1[tt] = !((nqv[tt + 1] == 0.0) | (nk_cur.y == 0.0));
                            s0[tt] = quot_core(acc0[T0 + tt], nqv[tt] * nk_cur.x);
                            s1[tt] = quot_core(acc1[T0 + tt], nqv[tt + 1] * nk_cur.y);
                        }
                    } else {
#pragma unroll
                        for (int tt = 0; tt < NT; ++tt) {
                            ok0[tt] = !((nqv[tt] == 0.0) | (nk_cur.x == 0.0));
                            ok1[tt] = !((nqv[tt + 1] == 0.0) | (nk_cur.y == 0.0));
                            s0[tt] = acc0[T0 + tt] / (nqv[tt] * nk_cur.x);
                            s1[tt] = acc1[T0 + tt] / (nqv[tt + 1] * nk_cur.y);
                        }
                    }
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const int x0 = j0 + T0 + tt;                                        // even sector iff T0 + tt is even
                        const int c0 = x0 >= S ? x0 - S : x0;
                        const int x1 = x0 + 1;
                        const int c1 = x1 >= S ? x1 - S : x1;
                        double *row = simbuf + tt * SB;
                        // idle lanes hold no sectors: their stores go to the two spare doubles at the end of the row
                        row[active ? (c0 >> 1) + (c0 & 1) * HALF : S] = ok0[tt] ? s0[tt] : 0.0;     // skipped sectors add +0.0: same bits
                        row[active ? (c1 >> 1) + (c1 & 1) * HALF : S + 1] = ok1[tt] ? s1[tt] : 0.0;
                        const int cnt = __popcll(__builtin_amdgcn_ballot_w64(active && ok0[tt])) + __popcll(__builtin_amdgcn_ballot_w64(active && ok1[tt]));
                        eff = (lane == tt) ? cnt : eff;
                    }
                } else {
                    // 3 waves/SIMD build: one shift at a time (few live registers), the partner waves hide the latency
#pragma unroll
                    for (int tt = 0; tt < NT; ++tt) {
                        const int x0 = j0 + T0 + tt, x1 = x0 + 1;
                        const double nq0 = nqe[x0], nq1 = nqe[x1];
                        const bool k0ok = !((nq0 == 0.0) | (nk_cur.x == 0.0));
                        const bool k1ok = !((nq1 == 0.0) | (nk_cur.y == 0.0));
                        const double q0 = acc0[T0 + tt] / (nq0 * nk_cur.x);
                        const double q1 = acc1[T0 + tt] / (nq1 * nk_cur.y);
                        const int c0 = x0 >= S ? x0 - S : x0;
                        const int c1 = x1 >= S ? x1 - S : x1;
                        double *row = simbuf + tt * SB;
                        row[active ? (c0 >> 1) + (c0 & 1) * HALF : S] = k0ok ? q0 : 0.0;
                        row[active ? (c1 >> 1) + (c1 & 1) * HALF : S + 1] = k1ok ? q1 : 0.0;
                        const int cnt = __popcll(__ballot(active && k0ok)) + __popcll(__ballot(active && k1ok));
                        eff = (lane == tt) ? cnt : eff;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                wave_fence();
                { const unsigned long long t1 = stamp(); st_c += t1 - st_t; st_t = t1; }
                if (lane < NT) {
                    const int t = T0 + lane;
                    const double2 *ev = reinterpret_cast<const double2 *>(simbuf + lane * SB);
                    const double2 *od = ev + S / 4;                                     // odd sectors start S/2 doubles in
                    double sum = 0.0;
                    // One dependent add chain of S terms per lane: the row is fetched in batches of DB
                    // steps, batch b+1 requested before the adds of batch b (pin1 fixes that order), so
                    // the chain never waits on an LDS round trip.
                    constexpr int DB = (MAXT > 512) ? 3 : 5;
                    static_assert((S / 4) % DB == 0, "sector-sum batches must tile the row");
                    double2 eb[2][DB], ob[2][DB];
#pragma unroll
                    for (int v = 0; v < DB; ++v) { eb[0][v] = ev[v]; ob[0][v] = od[v]; }
#pragma unroll
                    for (int bt = 0; bt < S / 4 / DB; ++bt) {
                        if (bt + 1 < S / 4 / DB) {
#pragma unroll
                            for (int v = 0; v < DB; ++v) { eb[(bt + 1) & 1][v] = ev[(bt + 1) * DB + v]; ob[(bt + 1) & 1][v] = od[(bt + 1) * DB + v]; }
                        }
                        pin1(sum);
#pragma unroll
                        for (int v = 0; v < DB; ++v) {                                  // sectors 4i .. 4i+3 in order
                            const double2 e = eb[bt & 1][v], o = ob[bt & 1][v];
                            sum = sum + e.x; sum = sum + o.x; sum = sum + e.y; sum = sum + o.y;
                        }
                        pin1(sum);
                    }
                    const double d = 1.0 - sum / (double)eff;                           // 0/0 -> NaN, never wins
                    const int st = wrap(s_start_cur + t, S);
                    if (d < kBigDist && ((d < dmin) | ((d == dmin) & (st < smin)))) { dmin = d; smin = st; }
                }
                { const unsigned long long t1 = stamp(); st_d += t1 - st_t; st_t = t1; }
            };
            static_assert(S % 4 == 0, "sector sums read two aligned pairs per step");
            if (!(a.ablate & 4)) {
                cd_pass(std::integral_constant<int, 0>{});
                __builtin_amdgcn_sched_barrier(0);
                cd_pass(std::integral_constant<int, 1>{});
            }
            wave_argmin_dpp(dmin, smin);
            if (lane == 0) {
                const bool ok = dmin < kBigDist;
                a.out_dist[ci] = ok ? dmin : kBigDist;
                a.out_shift[ci] = ok ? smin : 0;
            }
            {
                const double ds = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(dmin)),
                                                   __builtin_amdgcn_readfirstlane(__double2loint(dmin)));
                const int ss = __builtin_amdgcn_readfirstlane(smin);
                if (ds < kBigDist && ((ds < w_best) | ((ds == w_best) & (ci < w_bidx)))) { w_best = ds; w_bidx = ci; w_bshift = ss; }
            }
        } else if (lane == 0) {
            a.out_dist[ci] = kBigDist;
            a.out_shift[ci] = 0;
        }

        if (ci_next >= c_hi) break;
        ci = ci_next;
        slot = slot_next;
    }
    if (STAMP && a.stamps) {
        const unsigned long long cyc1 = stamp();
        const unsigned long long real1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *o = a.stamps + (size_t)(blockIdx.x * nwaves + wave) * 8;
            o[0] = cyc1 - st_cyc0; o[1] = real1 - st_real0; o[2] = st_a; o[3] = st_b; o[4] = st_c; o[5] = st_d;
        }
    }

    // ---- fused epilogue (full-DB mode): global arg-min and ring-key top-k without another launch ----
    // wave records -> workgroup record (LDS) -> global partial; the last workgroup to arrive (agent-scope
    // release on the producers, acquire on the consumer) reduces the partials and resets the counter.
    if (a.blk_part == nullptr) return;
    constexpr int REC = 2 + KT;
    unsigned long long *wrec = reinterpret_cast<unsigned long long *>(wbase) + (size_t)wave * wsz;   // wave scratch is free now
    if (lane == 0) {
        wrec[0] = (unsigned long long)__double_as_longlong(w_best);
        wrec[1] = ((unsigned long long)(unsigned)w_bidx << 32) | (unsigned)w_bshift;
#pragma unroll
        for (int t = 0; t < KT; ++t) wrec[2 + t] = w_top[t];
    }
    __syncthreads();
    unsigned int *s_ticket = reinterpret_cast<unsigned int *>(next_ticket) + 1;
    if (wave == 0) {
        // workgroup record: lane w < nwaves holds wave w's best, lane l holds one ring key of wave l / KT
        const unsigned long long *base = reinterpret_cast<unsigned long long *>(wbase);
        unsigned long long b0 = ~0ull, b1 = ~0ull;
        if (lane < nwaves) { b0 = base[(size_t)lane * wsz]; b1 = base[(size_t)lane * wsz + 1]; }
        const unsigned long long m0 = wave_min_u64(b0);                       // distance bits order like distances
        const unsigned long long m1 = wave_min_u64(b0 == m0 ? b1 : ~0ull);    // then the lowest position
        unsigned long long key = kNone;
        if (lane < nwaves * KT) key = base[(size_t)(lane / KT) * wsz + 2 + (lane % KT)];
        unsigned long long *bp = a.blk_part + (size_t)bid * REC;
        unsigned long long prev = 0ull;
        bool first = true;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const unsigned long long m = wave_min_u64((first || key > prev) ? key : kNone);
            if (lane == 0) bp[2 + t] = m;
            prev = m; first = false;
            if (m == kNone) { prev = kNone; }
        }
        if (lane == 0) {
            bp[0] = m0; bp[1] = m1;
            __threadfence();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *s_ticket = atomicAdd(a.done_counter, 1u);
        }
    }
    __syncthreads();
    if (*s_ticket != (unsigned)nbk - 1) return;
    if (wave != 0) return;
    __threadfence();                                   // acquire: partials of the other workgroups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        unsigned long long best = (unsigned long long)__double_as_longlong(kInf), tag = ~0ull;
        unsigned long long keys[4 * KT];
#pragma unroll
        for (int t = 0; t < 4 * KT; ++t) keys[t] = kNone;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = lane + u * kWave;
            if (b < nbk) {
                const unsigned long long *bp = a.blk_part + (size_t)b * REC;
                const unsigned long long r0 = __builtin_nontemporal_load(bp), r1 = __builtin_nontemporal_load(bp + 1);
                if (r0 < best || (r0 == best && (r1 >> 32) < (tag >> 32))) { best = r0; tag = r1; }
#pragma unroll
                for (int t = 0; t < KT; ++t) keys[u * KT + t] = __builtin_nontemporal_load(bp + 2 + t);
            }
        }
        double bd = __longlong_as_double((long long)best);
        int bpos = (int)(unsigned)(tag >> 32);
        const int my_shift = (int)(unsigned)(tag & 0xffffffffull);
        const double my_bd = bd; const int my_pos = bpos;
        wave_argmin(bd, bpos);
        if (my_bd == bd && my_pos == bpos && ((lane & 63) == __builtin_amdgcn_readfirstlane(__ffsll((long long)__ballot(my_bd == bd && my_pos == bpos)) - 1))) {
            const bool ok = bd < kBigDist;
            a.out3[0] = ok ? bd : kBigDist;
            // explicit candidate list: report the winner's slot relative to the range start, like the contiguous form
            a.out3[1] = ok ? (double)(a.cand ? a.cand[bpos] - a.slot_base : bpos) : -1.0;
            a.out3[2] = ok ? (double)my_shift : 0.0;
        }
        unsigned long long prev = 0ull;
        bool first = true;
        for (int round = 0; round < a.topk_k; ++round) {
            unsigned long long mine = kNone;
#pragma unroll
            for (int t = 0; t < 4 * KT; ++t) if ((first || keys[t] > prev) && keys[t] < mine) mine = keys[t];
            const unsigned long long m = wave_min_u64(mine);
            if (lane == 0) {
                if (m == kNone) { a.topk_idx[round] = -1; a.topk_d2[round] = 3.402823466e+38f; }
                else { a.topk_idx[round] = a.slot_base + (int)(unsigned)(m & 0xffffffffull); a.topk_d2[round] = __int_as_float((int)(m >> 32)); }
            }
            if (m == kNone) {
                for (int r2 = round + 1 + lane; r2 < a.topk_k; r2 += kWave) { a.topk_idx[r2] = -1; a.topk_d2[r2] = 3.402823466e+38f; }
                break;
            }
            prev = m; first = false;
        }
        if (lane == 0) *a.done_counter = 0u;            // armed for the next launch (stream ordered)
        if (lane == 0 && a.t_min) { *a.t_min = 0xffffffffu; a.t_min[kTminEpsOffset] = 0u; }   // survivors pass: every workgroup has read the minimum (and the bound) by now
    }
}

template <int RG, int W, int CH, int S, int MAXT, bool STAMP>
__global__ __launch_bounds__(MAXT) void sc_distance_wave_kernel(ScBatchArgs ab)
{
    const int nbk = ab.nb;                             // workgroups per query
    const int qi = ab.nq > 1 ? (int)blockIdx.x / nbk : 0;
    const int bid = (int)blockIdx.x - qi * nbk;        // workgroup index within its query
    sc_wave_body<RG, W, CH, S, MAXT, STAMP>(ab.q[qi], bid, nbk, bid, nbk);
}

// Exact pass behind the screening pass (sc_screen.hip): query qi = qargs[qi] (any number of queries: the argument sets
// live in device memory), nb workgroups per query.  Every workgroup first builds the survivor list of its query --
// the slots whose approximate distance lies within 2 eps of the minimum, ascending, written to a.cand (all nb
// workgroups write the same values) -- and then runs the wave program on its share of that list.
template <int RG, int W, int CH, int S, int MAXT>
__global__ __launch_bounds__(MAXT) void sc_distance_survivors_kernel(const ScArgs *qargs, int nb)
{
    const int qi = (int)blockIdx.x / nb;
    const int bid = (int)blockIdx.x - qi * nb;
    ScArgs a = qargs[qi];
    __shared__ int wave_total[MAXT / kWave];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const unsigned int tm = *a.t_min;
    float thr = __int_as_float(0xff800000);                                      // nothing screened: only the "score exactly" marks pass
    if (tm != 0xffffffffu) {
        const unsigned int b = (tm >> 31) ? (tm & 0x7fffffffu) : ~tm;            // inverse of the ordered image
        const unsigned int ew = a.t_min[kTminEpsOffset];                       // the launch's largest per-pair bound (0: not recorded)
        const float two_eps = ew ? fminf(2.0f * __uint_as_float(ew) * 1.0001f, a.two_eps) : a.two_eps;
        thr = __int_as_float((int)b) + two_eps;
    }
    // Every wave owns a contiguous part of the range and walks it 64 entries at a time (coalesced reads; a thread-owned
    // chunk made every load a 64-line gather): count, prefix over the waves, then the same walk writes the list --
    // iteration-major, lane-minor = ascending slots.
    const int nwv = (int)blockDim.x / kWave;
    const int per_wave = (((a.range_n + nwv - 1) / nwv) + kWave - 1) / kWave * kWave;
    const int wlo = wv * per_wave;
    const int whi = wlo + per_wave < a.range_n ? wlo + per_wave : a.range_n;
    int cnt = 0;
    for (int base = wlo; base < whi; base += kWave) {
        const int i = base + lane;
        const bool pass = i < whi && a.approx[i] <= thr;                         // -inf (score exactly) always passes
        cnt += __popcll(__builtin_amdgcn_ballot_w64(pass));
    }
    if (lane == 0) wave_total[wv] = cnt;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < nwv; ++w) { const int t = wave_total[w]; if (w < wv) before += t; total += t; }
    if (bid == 0 && threadIdx.x == 0 && a.out3) a.out3[3] = (double)total;      // (the stream form looks at it: which exact pass the next chunks take)
    if (bid == 0 && threadIdx.x == 0 && a.surv_stats) {
        atomicAdd(a.surv_stats, (unsigned long long)total);
        atomicMax(a.surv_stats + 1, (unsigned long long)total);
        atomicAdd(a.surv_stats + 2, 1ull);
    }
    int *list = const_cast<int *>(a.cand);
    for (int base = wlo; base < whi; base += kWave) {
        const int i = base + lane;
        const bool pass = i < whi && a.approx[i] <= thr;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
        if (pass) list[before + __popcll(m & ((1ull << lane) - 1ull))] = a.slot_base + i;
        before += __popcll(m);
    }
    __threadfence_block();
    __syncthreads();
    // The ring-key top-k of the range (workgroup 0 of the query): k rounds of "smallest key larger than the previous
    // pick" over the metric the screening pass stored; keys (d2 bits << 32 | position) are unique.
    // (the query's LAST workgroup: it scores nothing, so the top-k runs beside the others' scoring, not in front of it)
    const bool topk_block = nb > 1 && bid == nb - 1;
    if ((nb > 1 ? topk_block : bid == 0) && a.sel_topk_k > 0) {
        __shared__ unsigned long long wave_key[MAXT / kWave];
        const unsigned long long none = ~0ull;
        unsigned long long prev = 0ull;
        bool first = true;
        if (a.ring_from_keys) {
            // nanoflann's metric (nanoflann.hpp:383-408) of every keyframe of the range, the arithmetic of sc_screen2_finish_compute:
            // four dimensions per step, fp32, groups accumulated in order; consecutive threads on consecutive slots of the tiled keys
            float *rd = const_cast<float *>(a.ring_d2);
            for (int i = (int)threadIdx.x; i < a.range_n; i += (int)blockDim.x) {
                const int slot = a.slot_base + i;
                float4 bk[RG];
#pragma unroll
                for (int r = 0; r < RG; ++r) bk[r] = a.rkey4[(size_t)r * a.rk_cap + slot];
                float result = 0.0f;
#pragma unroll
                for (int r = 0; r < RG; ++r) {
                    const float4 qk = *reinterpret_cast<const float4 *>(a.q_rkey + 4 * r);
                    const float d0 = qk.x - bk[r].x, d1 = qk.y - bk[r].y, d2 = qk.z - bk[r].z, d3 = qk.w - bk[r].w;
                    const float grp = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
                    result += grp;
                }
                rd[i] = result;
            }
            __threadfence_block();
            __syncthreads();
        }
        for (int round = 0; round < a.sel_topk_k; ++round) {
            unsigned long long mine = none;
            for (int i = (int)threadIdx.x; i < a.range_n; i += (int)blockDim.x) {
                const float r = a.ring_d2[i];
                const bool excluded = (a.sel_exclude_eps > 0.0f) && (r <= a.sel_exclude_eps);
                if (excluded || !(r < 3.402823466e+38f)) continue;
                const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(r) << 32) | (unsigned)i;
                if ((first || key > prev) && key < mine) mine = key;
            }
            mine = wave_min_u64(mine);
            __syncthreads();
            if (lane == 0) wave_key[wv] = mine;
            __syncthreads();
            unsigned long long m = wave_key[0];
            for (int w = 1; w < (int)blockDim.x / kWave; ++w) m = wave_key[w] < m ? wave_key[w] : m;
            if (threadIdx.x == 0) {
                if (m == none) { a.sel_topk_idx[round] = -1; a.sel_topk_d2[round] = 3.402823466e+38f; }
                else { a.sel_topk_idx[round] = a.slot_base + (int)(unsigned)(m & 0xffffffffull); a.sel_topk_d2[round] = __int_as_float((int)(m >> 32)); }
            }
            if (m == none) {
                for (int r2 = round + 1 + (int)threadIdx.x; r2 < a.sel_topk_k; r2 += blockDim.x) { a.sel_topk_idx[r2] = -1; a.sel_topk_d2[r2] = 3.402823466e+38f; }
                break;
            }
            prev = m; first = false;
        }
        __syncthreads();
    }
    a.n = topk_block ? 0 : total; a.n_dev = nullptr;
    if (nb > 1) sc_wave_body<RG, W, CH, S, MAXT, false>(a, bid, nb, topk_block ? 0 : bid, topk_block ? 1 : nb - 1);
    else sc_wave_body<RG, W, CH, S, MAXT, false>(a, bid, nb, bid, nb);
}

constexpr int kTailReadable = 4 * kWave;        // partial records the last workgroup of the fused epilogue merges (u < 4 x 64 lanes)

template <int RG, int W, int CH, int S, int MAXT = 512, bool STAMP = false>
hipError_t launch_wave(const ScBatchArgs &batch_in, int num_cu, hipStream_t stream, int fixed_blocks = 0)
{
    ScBatchArgs ab = batch_in;
    ScArgs &a = ab.q[0];                                   // sizes the launch (batch: the largest n, set by the caller)
    const int n_launch = ab.nq > 1 ? ab.nb : a.n;          // caller passes max n in nb for a batch
    constexpr int QS = S + W + 1;
    constexpr int HSH = (W + 1) / 2;
    const size_t fixed = (size_t)(RG * 4 * QS + QS + 2 * S) * sizeof(double);   // query rows, norms, sector key (fp64 + two fp32 copies)
    const size_t per_wave = (size_t)((2 * S > HSH * (S + 2)) ? 2 * S : HSH * (S + 2)) * sizeof(double);
    const size_t lds_cap = 160 * 1024;
    int waves = (int)((lds_cap - fixed - 16) / per_wave);
    if (waves > MAXT / kWave) waves = MAXT / kWave;
#ifdef SCL_DIAGNOSTICS
    static const int wave_cap = [] { const int v = scl_lab_int("SCL_SC_WAVES", 0); return v > 0 ? v : 1 << 20; }();
    if (waves > wave_cap) waves = wave_cap;               // diagnostic: fewer waves per CU
#endif
    if (waves < 1) return hipErrorInvalidValue;
    if (!fixed_blocks) while (waves > 1 && n_launch < num_cu * waves) waves = (waves + 1) / 2;
    int blocks = (n_launch + waves - 1) / waves;
    if (blocks > num_cu) blocks = num_cu;
    if (fixed_blocks) blocks = fixed_blocks;               // candidate count known only on the device: full-width workgroups
    // the fused epilogue's last workgroup reads at most kTailReadable partials (4 per lane): never launch more
    // workgroups per query than it can merge (a device with more than 256 CUs would otherwise lose partials silently)
    if (a.blk_part && blocks > kTailReadable) blocks = kTailReadable;
    const int grid = blocks * ab.nq;                       // workgroups [qi*blocks, (qi+1)*blocks) serve query qi
    ab.nb = blocks;
    const size_t lds = fixed + per_wave * waves + 16;      // + the candidate dispenser
    static std::atomic<bool> attr_set_dev[64];   // per instantiation and per device (engines on other threads may race: atomic)
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<bool> &attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)sc_distance_wave_kernel<RG, W, CH, S, MAXT, STAMP>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    if (STAMP) {
        // diagnostic build of the kernel (SCL_STAMP=1): per-wave cycle sums per phase, printed to stderr
        const size_t nw = (size_t)blocks * waves;
        unsigned long long *d = nullptr;
        if (hipMalloc(&d, nw * 8 * sizeof(unsigned long long)) != hipSuccess) return hipErrorOutOfMemory;
        (void)hipMemsetAsync(d, 0, nw * 8 * sizeof(unsigned long long), stream);
        a.stamps = d;
        hipLaunchKernelGGL((sc_distance_wave_kernel<RG, W, CH, S, MAXT, STAMP>), dim3(grid), dim3(waves * kWave), lds, stream, ab);
        (void)hipStreamSynchronize(stream);
        std::vector<unsigned long long> h(nw * 8);
        (void)hipMemcpy(h.data(), d, nw * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        (void)hipFree(d);
        static int printed = 0;
        if (printed++ < 3) {
            auto med = [&](int k) {
                std::vector<unsigned long long> v;
                for (size_t w = 0; w < nw; ++w) if (h[w * 8] != 0) v.push_back(h[w * 8 + k]);
                if (v.empty()) return 0.0;
                std::sort(v.begin(), v.end());
                return (double)v[v.size() / 2];
            };
            auto pct = [&](int k, double q) {
                std::vector<unsigned long long> v;
                for (size_t w = 0; w < nw; ++w) if (h[w * 8] != 0) v.push_back(h[w * 8 + k]);
                if (v.empty()) return 0.0;
                std::sort(v.begin(), v.end());
                return (double)v[(size_t)(q * (v.size() - 1))];
            };
            fprintf(stderr, "[scl stamp] wave lifetime us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n",
                    pct(1, 0.0) / 100.0, pct(1, 0.1) / 100.0, pct(1, 0.5) / 100.0, pct(1, 0.9) / 100.0, pct(1, 1.0) / 100.0);
            {   // where do the slow waves sit?  mean lifetime by blockIdx %% 8 (XCD group) and by wave slot
                double sx[8] = {0}, sw[16] = {0}; int nx[8] = {0}, nwv[16] = {0};
                for (size_t w = 0; w < nw; ++w) if (h[w * 8] != 0) {
                    const int blk = (int)(w / waves), wv = (int)(w % waves);
                    sx[blk & 7] += h[w * 8 + 1] / 100.0; nx[blk & 7]++;
                    sw[wv & 15] += h[w * 8 + 1] / 100.0; nwv[wv & 15]++;
                }
                fprintf(stderr, "[scl stamp] mean us by block%%8:");
                for (int k = 0; k < 8; ++k) fprintf(stderr, " %.1f", nx[k] ? sx[k] / nx[k] : 0.0);
                fprintf(stderr, " | by wave slot:");
                for (int k = 0; k < waves && k < 16; ++k) fprintf(stderr, " %.1f", nwv[k] ? sw[k] / nwv[k] : 0.0);
                fprintf(stderr, "\n[scl stamp] slowest blocks:");
                std::vector<std::pair<double, int>> bl;
                for (int b = 0; b < blocks; ++b) { double m = 0; for (int k = 0; k < waves; ++k) m = std::max(m, h[((size_t)b * waves + k) * 8 + 1] / 100.0); bl.push_back({m, b}); }
                std::sort(bl.rbegin(), bl.rend());
                for (int k = 0; k < 24 && k < (int)bl.size(); ++k) fprintf(stderr, " %d:%.0f", bl[k].second, bl[k].first);
                fprintf(stderr, "\n");
            }
            const double cyc = med(0), real = med(1);
            fprintf(stderr, "[scl stamp] waves=%zu n=%d  wave lifetime: %.0f cycles, %.2f us  => clock %.2f GHz | "
                            "per wave: align %.0f  dots %.0f  sims %.0f  sums %.0f  other %.0f cycles\n",
                    nw, a.n, cyc, real / 100.0, real > 0 ? cyc / real * 0.1 : 0.0, med(2), med(3), med(4), med(5),
                    cyc - med(2) - med(3) - med(4) - med(5));
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL((sc_distance_wave_kernel<RG, W, CH, S, MAXT, STAMP>), dim3(grid), dim3(waves * kWave), lds, stream, ab);
    return hipGetLastError();
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

// arg-min over the exact distances of the survivor lists of up to four queries (counts on the device); out3[1] = the
// winner's slot relative to slot_base (ties -> lowest position = lowest slot: the lists are ascending)
struct ArgminBatch { const double *dist[kWideExactBatch]; const int *shift[kWideExactBatch]; const int *n_dev[kWideExactBatch];
                     const int *cand[kWideExactBatch]; int slot_base[kWideExactBatch]; double *out3[kWideExactBatch]; };
__global__ __launch_bounds__(1024) void argmin_survivors_batch_kernel(ArgminBatch b)
{
    const int q = blockIdx.x;
    __shared__ double sv[16];
    __shared__ int si[16];
    const double *dist = b.dist[q]; const int *shift = b.shift[q]; const int *cand = b.cand[q];
    const int n = *b.n_dev[q];
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
        double *out3 = b.out3[q];
        out3[0] = ok ? best : kBigDist;
        out3[1] = ok ? (double)(cand[bi] - b.slot_base[q]) : -1.0;
        out3[2] = ok ? (double)shift[bi] : 0.0;
        out3[3] = (double)n;                             // (the list's length: the stream form looks at it, as in the one-workgroup-per-scan pass)
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
    static std::atomic<bool> attr_set_dev[64];   // per instantiation and per device (one engine per GPU in a process is allowed)
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<bool> &attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)sc_distance_kernel<RG, W, MAXT>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap);
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((sc_distance_kernel<RG, W, MAXT>), dim3(blocks), dim3(G * a.NW * kWave), lds, stream, a);
    return hipGetLastError();
}

}  // namespace

// Phase ablation / in-kernel stamps / occupancy overrides change results or timing on purpose: they exist only in
// builds made with -DSCL_DIAGNOSTICS (scripts/ablate_k1.sh); the product library ignores the environment.
int ablate_flags()
{
#ifdef SCL_DIAGNOSTICS
    static const int f = scl_lab_int("SCL_ABLATE", 0);
    return f;
#else
    return 0;
#endif
}

int align_filter_enabled()
{
    static const int on = scl_lab_int("SCL_ALIGN_FILTER", 1) != 0 ? 1 : 0;
    return on;
}

hipError_t launch_sc_distance(const DbView &db, const QueryView &q, const int *cand, int slot_base,
                              int n, int SR, double *out_dist, int *out_shift, int num_cu,
                              hipStream_t stream, float *out_ring_d2, bool *ring_fused, const FullTail *tail)
{
    if (ring_fused) *ring_fused = false;
    if (n <= 0) return hipSuccess;
    ScArgs a;
    a.desc = db.desc; a.vkey = db.vkey; a.norm = db.norm;
    a.q_desc = q.desc; a.q_vkey = q.vkey; a.q_norm = q.norm;
    a.cand = cand; a.slot_base = slot_base; a.n = n; a.n_dev = nullptr; a.approx = nullptr; a.t_min = nullptr; a.range_n = 0; a.two_eps = 0.f; a.ring_d2 = nullptr; a.ring_from_keys = 0; a.sel_topk_idx = nullptr; a.surv_stats = nullptr; a.sel_topk_d2 = nullptr; a.sel_topk_k = 0; a.sel_exclude_eps = 0.f; a.S = db.S; a.SR = SR;
    a.NW = (db.S + kWave - 1) / kWave; a.G = 1;
    a.ablate = ablate_flags();
    a.align_filter = align_filter_enabled();
    a.stamps = nullptr;
    a.rkey4 = db.rkey4; a.rk_cap = db.cap; a.q_rkey = q.rkey; a.out_d2 = nullptr;
    a.blk_part = nullptr; a.done_counter = nullptr; a.out3 = nullptr; a.topk_idx = nullptr; a.topk_d2 = nullptr;
    a.topk_k = 0; a.exclude_eps = 0.0f;
    a.out_dist = out_dist; a.out_shift = out_shift;
    const int W = 2 * SR + 1;
    static const bool force_v1 = scl_lab_is("SCL_SC_KERNEL", "v1");
    const bool wave_ok = !force_v1;
    const bool wave_grid = wave_ok && ((db.RG == 5 && W == 7 && db.S == 60) || (db.RG == 16 && W == 13 && db.S == 120)) && (db.R % 4 == 0);
    if (wave_grid && out_ring_d2) {
        a.out_d2 = out_ring_d2; if (ring_fused) *ring_fused = true;
        if (tail && tail->k <= kTailTop) {
            a.blk_part = tail->blk_part; a.done_counter = tail->done_counter; a.out3 = tail->out3;
            a.topk_idx = tail->topk_idx; a.topk_d2 = tail->topk_d2; a.topk_k = tail->k; a.exclude_eps = tail->exclude_eps;
        }
    }
    ScBatchArgs one{};
    one.q[0] = a; one.nq = 1; one.nb = 0;
    if (wave_ok && db.RG == 5 && W == 7 && db.S == 60)   return launch_wave<5, 7, 5, 60>(one, num_cu, stream);
#ifdef SCL_DIAGNOSTICS
    static const bool stamp = scl_lab_int("SCL_STAMP", 0) == 1;
    static const int occ = scl_lab_int("SCL_SC_WAVES", 8);
#else
    constexpr bool stamp = false;
    constexpr int occ = 8;
#endif
    if (wave_ok && db.RG == 16 && W == 13 && db.S == 120 && stamp && occ > 8) return launch_wave<16, 13, 2, 120, 768, true>(one, num_cu, stream);
    if (wave_ok && db.RG == 16 && W == 13 && db.S == 120 && stamp) return launch_wave<16, 13, 4, 120, 512, true>(one, num_cu, stream);
    if (wave_ok && db.RG == 16 && W == 13 && db.S == 120 && occ > 8) return launch_wave<16, 13, 2, 120, 768>(one, num_cu, stream);
    if (wave_ok && db.RG == 16 && W == 13 && db.S == 120) return launch_wave<16, 13, 4, 120, 512>(one, num_cu, stream);
    if (db.RG == 5 && W == 7 && db.S >= W)   return launch_fast<5, 7, 512>(a, num_cu, stream);
    if (db.RG == 16 && W == 13 && db.S >= W) return launch_fast<16, 13, 512>(a, num_cu, stream);
    if (db.RG == 20 && W == 19 && db.S >= W && db.S <= 180) return launch_fast<20, 19, 256>(a, num_cu, stream);

    const size_t lds = (size_t)6 * db.S * sizeof(double);
    hipLaunchKernelGGL(sc_distance_generic_kernel, dim3(n), dim3(256), lds, stream, a, db.RG, db.R);
    return hipGetLastError();
}

hipError_t launch_sc_distance_batch(const DbView &db, const QueryBatch &qb, int SR, double *out_dist, int *out_shift,
                                    float *out_ring_d2, const FullTail &tail, int num_cu, hipStream_t stream)
{
    if (qb.nq < 1 || qb.nq > kMaxQueryBatch || !out_ring_d2 || tail.k > kTailTop || !sc_distance_fuses_ring(db, SR))
        return hipErrorInvalidValue;
    ScBatchArgs ab{};
    ab.nq = qb.nq;
    int nmax = 0;
    for (int i = 0; i < qb.nq; ++i) {
        if (qb.n[i] <= 0) return hipErrorInvalidValue;                 // empty ranges are the caller's business
        ScArgs &a = ab.q[i];
        const size_t slot = (size_t)qb.slot[i];
        a.desc = db.desc; a.vkey = db.vkey; a.norm = db.norm;
        a.q_desc = db.desc + slot * (size_t)(db.RG * db.S); a.q_vkey = db.vkey + slot * db.S;
        a.q_norm = db.norm + slot * db.S; a.q_rkey = db.rkey + slot * (size_t)(4 * db.RG);
        a.cand = nullptr; a.slot_base = qb.base[i]; a.n = qb.n[i]; a.n_dev = nullptr; a.approx = nullptr; a.t_min = nullptr; a.range_n = 0; a.two_eps = 0.f; a.ring_d2 = nullptr; a.ring_from_keys = 0; a.sel_topk_idx = nullptr; a.surv_stats = nullptr; a.sel_topk_d2 = nullptr; a.sel_topk_k = 0; a.sel_exclude_eps = 0.f; a.S = db.S; a.SR = SR;
        a.NW = (db.S + kWave - 1) / kWave; a.G = 1;
        a.ablate = ablate_flags(); a.stamps = nullptr; a.align_filter = align_filter_enabled();
        a.rkey4 = db.rkey4; a.rk_cap = db.cap;
        a.out_dist = out_dist + (size_t)i * qb.pair_stride; a.out_shift = out_shift + (size_t)i * qb.pair_stride;
        a.out_d2 = out_ring_d2 + (size_t)i * qb.pair_stride;
        a.blk_part = tail.blk_part + (size_t)i * kTailBlocks * kTailRec; a.done_counter = tail.done_counter + i;
        a.out3 = qb.out3[i];
        a.topk_idx = tail.topk_idx + i * kTailTopMaxK; a.topk_d2 = tail.topk_d2 + i * kTailTopMaxK;
        a.topk_k = tail.k; a.exclude_eps = tail.exclude_eps;
        nmax = qb.n[i] > nmax ? qb.n[i] : nmax;
    }
    for (int i = qb.nq; i < kMaxQueryBatch; ++i) ab.q[i] = ab.q[0];
    ab.nb = nmax;                                                      // launch_wave sizes the grid from it
    const int W = 2 * SR + 1;
    if (db.RG == 5 && W == 7 && db.S == 60) return launch_wave<5, 7, 5, 60>(ab, num_cu, stream);
    return launch_wave<16, 13, 4, 120, 512>(ab, num_cu, stream);
}

hipError_t launch_sc_distance_matrix(const DbView &db, const int *slots, int nq, int base, int n, int SR,
                                     double *out_dist, int *out_shift, size_t row_stride, int num_cu, hipStream_t stream)
{
    if (nq < 1 || nq > kMaxQueryBatch || n <= 0) return hipErrorInvalidValue;
    const int W = 2 * SR + 1;
    const bool wave_grid = ((db.RG == 5 && W == 7 && db.S == 60) || (db.RG == 16 && W == 13 && db.S == 120)) && (db.R % 4 == 0);
    static const bool force_v1 = scl_lab_is("SCL_SC_KERNEL", "v1");
    if (!wave_grid || force_v1) {
        for (int i = 0; i < nq; ++i) {
            QueryView q{};
            const size_t slot = (size_t)slots[i];
            q.desc = db.desc + slot * (size_t)(db.RG * db.S); q.vkey = db.vkey + slot * db.S; q.norm = db.norm + slot * db.S;
            q.rkey = db.rkey + slot * (size_t)(4 * db.RG);
            hipError_t e = launch_sc_distance(db, q, nullptr, base, n, SR, out_dist + (size_t)i * row_stride, out_shift + (size_t)i * row_stride, num_cu, stream);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    ScBatchArgs ab{};
    ab.nq = nq;
    for (int i = 0; i < nq; ++i) {
        ScArgs &a = ab.q[i];
        const size_t slot = (size_t)slots[i];
        a.desc = db.desc; a.vkey = db.vkey; a.norm = db.norm;
        a.q_desc = db.desc + slot * (size_t)(db.RG * db.S); a.q_vkey = db.vkey + slot * db.S;
        a.q_norm = db.norm + slot * db.S; a.q_rkey = db.rkey + slot * (size_t)(4 * db.RG);
        a.cand = nullptr; a.slot_base = base; a.n = n; a.n_dev = nullptr; a.approx = nullptr; a.t_min = nullptr; a.range_n = 0; a.two_eps = 0.f; a.ring_d2 = nullptr; a.ring_from_keys = 0;
        a.sel_topk_idx = nullptr; a.surv_stats = nullptr; a.sel_topk_d2 = nullptr; a.sel_topk_k = 0; a.sel_exclude_eps = 0.f; a.S = db.S; a.SR = SR;
        a.NW = (db.S + kWave - 1) / kWave; a.G = 1;
        a.ablate = ablate_flags(); a.stamps = nullptr; a.align_filter = align_filter_enabled();
        a.rkey4 = db.rkey4; a.rk_cap = db.cap; a.out_d2 = nullptr;
        a.out_dist = out_dist + (size_t)i * row_stride; a.out_shift = out_shift + (size_t)i * row_stride;
        a.blk_part = nullptr; a.done_counter = nullptr; a.out3 = nullptr; a.topk_idx = nullptr; a.topk_d2 = nullptr; a.topk_k = 0; a.exclude_eps = 0.0f;
    }
    for (int i = nq; i < kMaxQueryBatch; ++i) ab.q[i] = ab.q[0];
    ab.nb = n;                                                         // launch_wave sizes the grid from it
    if (db.RG == 5) return launch_wave<5, 7, 5, 60>(ab, num_cu, stream);
    return launch_wave<16, 13, 4, 120, 512>(ab, num_cu, stream);
}

hipError_t launch_sc_distance_survivors(const DbView &db, const SurvivorPass &sp, int SR, int num_cu, hipStream_t stream, int phases)
{
    if (sp.nq < 1 || sp.nq > kMaxSurvivorQueries || !(db.RG == 16 && db.S == 120 && SR == 6) || !sp.d_args || !sp.h_args) return hipErrorInvalidValue;
    constexpr int RG = 16, W = 13, CH = 4, S = 120, MAXT = 512;
    ScArgs *h = reinterpret_cast<ScArgs *>(sp.h_args);
    static_assert(sizeof(ScArgs) <= kSurvivorArgBytes, "argument set must fit the slot the engine reserves");
    for (int i = 0; (phases & kSurvivorArgs) && i < sp.nq; ++i) {
        ScArgs &a = h[i];
        const size_t slot = (size_t)sp.slot[i];
        a.desc = db.desc; a.vkey = db.vkey; a.norm = db.norm;
        a.q_desc = db.desc + slot * (size_t)(db.RG * db.S); a.q_vkey = db.vkey + slot * db.S;
        a.q_norm = db.norm + slot * db.S; a.q_rkey = db.rkey + slot * (size_t)(4 * db.RG);
        a.cand = sp.survivors + (size_t)sp.buf[i] * sp.pair_stride; a.slot_base = sp.base[i]; a.n = 0; a.n_dev = nullptr;
        a.approx = sp.approx + (size_t)sp.buf[i] * sp.pair_stride; a.t_min = sp.t_min + sp.buf[i]; a.range_n = sp.n[i]; a.two_eps = 2.0f * sc_screen_eps();
        a.S = db.S; a.SR = SR; a.NW = (db.S + kWave - 1) / kWave; a.G = 1;
        a.ablate = ablate_flags(); a.stamps = nullptr; a.align_filter = align_filter_enabled();
        a.rkey4 = db.rkey4; a.rk_cap = db.cap;
        a.out_dist = sp.out_dist + (size_t)sp.buf[i] * sp.pair_stride; a.out_shift = sp.out_shift + (size_t)sp.buf[i] * sp.pair_stride;
        a.out_d2 = nullptr;                                            // the ring-key top-k comes from the screening pass
        a.blk_part = sp.blk_part + (size_t)i * kSurvivorBlocks * kTailRec; a.done_counter = sp.done_counter + i;
        a.out3 = sp.out3[i];
        a.topk_idx = nullptr; a.topk_d2 = nullptr; a.topk_k = 0; a.exclude_eps = 0.0f;
        a.ring_d2 = sp.ring_d2 + (size_t)sp.buf[i] * sp.pair_stride; a.ring_from_keys = sp.ring_from_keys;
        a.sel_topk_idx = sp.topk_idx + sp.buf[i] * kTailTopMaxK; a.sel_topk_d2 = sp.topk_d2 + sp.buf[i] * kTailTopMaxK;
        a.sel_topk_k = sp.k; a.sel_exclude_eps = sp.exclude_eps;
        a.surv_stats = sp.surv_stats;
    }
    hipError_t e = hipSuccess;
    if (phases & kSurvivorArgs) {
        e = hipMemcpyAsync(sp.d_args, sp.h_args, sizeof(ScArgs) * (size_t)sp.nq, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
    }
    if (!(phases & kSurvivorKernel)) return hipSuccess;
    constexpr int QS = S + W + 1, HSH = (W + 1) / 2;
    const size_t fixed = (size_t)(RG * 4 * QS + QS + 2 * S) * sizeof(double);
    const size_t per_wave = (size_t)((2 * S > HSH * (S + 2)) ? 2 * S : HSH * (S + 2)) * sizeof(double);
    const int waves = MAXT / kWave;
    const size_t lds = fixed + per_wave * waves + 16;
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<bool> &attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        e = hipFuncSetAttribute((const void *)sc_distance_survivors_kernel<RG, W, CH, S, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // (the kernel also has a few static words)
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    (void)num_cu;
    hipLaunchKernelGGL((sc_distance_survivors_kernel<RG, W, CH, S, MAXT>), dim3(sp.nq * kSurvivorBlocks), dim3(waves * kWave), lds, stream,
                       reinterpret_cast<const ScArgs *>(sp.d_args), kSurvivorBlocks);
    return hipGetLastError();
}

// 80 x 180: exact distances of the survivor list by the one-sector-per-lane kernel, then the arg-min (two launches)
// Exact pass of the 80 x 180 grid over the survivor lists of nq <= 4 queries: ONE launch of the one-sector-per-lane program
// (blockIdx.y = query, the survivor counts stay on the device) and ONE arg-min launch; winners to out3[i].
hipError_t launch_sc_distance_survivors_wide(const DbView &db, int nq, const int *query_slot, const int *slot_base, int SR,
                                             const int *const *survivors, const int *const *n_surv, double *const *out_dist, int *const *out_shift,
                                             double *const *out3, int num_cu, hipStream_t stream, const int *const *starts, const unsigned int *const *smask)
{
    if (!(db.RG == 20 && db.S == 180 && SR == 9) || nq < 1 || nq > kWideExactBatch) return hipErrorInvalidValue;
    constexpr int RG = 20, W = 19, MAXT = 256;
    ScArgsBatch ab{};
    ArgminBatch mb{};
    // The wave program of this grid (sc_masked.hip): every survivor's exact distance at the shifts its screening left open, the
    // first shift taken from the screening's alignment -- instead of the one-sector-per-lane program below, whose workgroups each
    // staged 126 KB of fp64 scan and re-aligned every pair (SCL_WIDE_EXACT=0 keeps it)
    static const bool use_masked = scl_lab_int("SCL_WIDE_EXACT", 1) != 0;
    if (use_masked && starts && smask && sc_masked_supported(db, SR)) {
        MaskedQuery mq[kMaxMaskedQueries];
        static_assert(kWideExactBatch <= kMaxMaskedQueries, "one masked launch per exact batch");
        for (int i = 0; i < nq; ++i) {
            mq[i].qslot = query_slot[i]; mq[i].slot_base = slot_base[i]; mq[i].n = 0; mq[i].n_dev = n_surv[i]; mq[i].cand = survivors[i];
            mq[i].starts = starts[i]; mq[i].smask = smask[i]; mq[i].out_dist = out_dist[i]; mq[i].out_shift = out_shift[i];
            mb.dist[i] = out_dist[i]; mb.shift[i] = out_shift[i]; mb.n_dev[i] = n_surv[i]; mb.cand[i] = survivors[i];
            mb.slot_base[i] = slot_base[i]; mb.out3[i] = out3[i];
        }
        for (int i = nq; i < kWideExactBatch; ++i) { mb.dist[i] = mb.dist[0]; mb.shift[i] = mb.shift[0]; mb.n_dev[i] = mb.n_dev[0]; mb.cand[i] = mb.cand[0]; mb.slot_base[i] = mb.slot_base[0]; mb.out3[i] = mb.out3[0]; }
        hipError_t e = launch_sc_masked(db, SR, mq, nq, 4, stream, true);     // four workgroups per scan: 32 survivors at once, more loop; the scans go round the XCDs
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(argmin_survivors_batch_kernel, dim3(nq), dim3(1024), 0, stream, mb);
        return hipGetLastError();
    }
    if (nq > kWideLegacyBatch) {                                // (SCL_WIDE_EXACT=0 / no shift masks: two launches of the older program)
        hipError_t e = launch_sc_distance_survivors_wide(db, kWideLegacyBatch, query_slot, slot_base, SR, survivors, n_surv, out_dist, out_shift, out3, num_cu, stream, starts, smask);
        if (e != hipSuccess) return e;
        const int o = kWideLegacyBatch;
        return launch_sc_distance_survivors_wide(db, nq - o, query_slot + o, slot_base + o, SR, survivors + o, n_surv + o, out_dist + o, out_shift + o, out3 + o, num_cu, stream,
                                                 starts ? starts + o : nullptr, smask ? smask + o : nullptr);
    }
    for (int i = 0; i < kWideLegacyBatch; ++i) {
        const int j = i < nq ? i : 0;
        ScArgs &a = ab.q[i];
        const size_t slot = (size_t)query_slot[j];
        a.desc = db.desc; a.vkey = db.vkey; a.norm = db.norm;
        a.q_desc = db.desc + slot * (size_t)(db.RG * db.S); a.q_vkey = db.vkey + slot * db.S; a.q_norm = db.norm + slot * db.S;
        a.q_rkey = db.rkey + slot * (size_t)(4 * db.RG);
        a.cand = survivors[j]; a.slot_base = slot_base[j]; a.n = 256;        // sizes nothing here; the kernel loops over *n_dev
        a.n_dev = n_surv[j]; a.approx = nullptr; a.t_min = nullptr; a.range_n = 0; a.two_eps = 0.f;
        a.ring_d2 = nullptr; a.ring_from_keys = 0; a.sel_topk_idx = nullptr; a.surv_stats = nullptr; a.sel_topk_d2 = nullptr; a.sel_topk_k = 0; a.sel_exclude_eps = 0.f;
        a.S = db.S; a.SR = SR; a.NW = (db.S + kWave - 1) / kWave; a.G = 1;
        a.ablate = 0; a.align_filter = 0; a.stamps = nullptr;
        a.rkey4 = db.rkey4; a.rk_cap = db.cap; a.out_d2 = nullptr;
        a.blk_part = nullptr; a.done_counter = nullptr; a.out3 = nullptr; a.topk_idx = nullptr; a.topk_d2 = nullptr; a.topk_k = 0; a.exclude_eps = 0.f;
        a.out_dist = out_dist[j]; a.out_shift = out_shift[j];
        mb.dist[i] = out_dist[j]; mb.shift[i] = out_shift[j]; mb.n_dev[i] = n_surv[j]; mb.cand[i] = survivors[j];
        mb.slot_base[i] = slot_base[j]; mb.out3[i] = out3[j];
    }
    const int S = db.S, QS = S + W - 1;
    const size_t lds = (size_t)(RG * 4 * QS + 2 * S) * sizeof(double) + (size_t)(4 * S + W * S + 8) * sizeof(double);   // G = 1
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<bool> &attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)sc_distance_batch4_kernel<RG, W, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    (void)num_cu;
    // survivors are few (one to a handful per query): 16 workgroups per query cover 16 of them at once, more loop
    hipLaunchKernelGGL((sc_distance_batch4_kernel<RG, W, MAXT>), dim3(16, nq), dim3(ab.q[0].NW * kWave), lds, stream, ab);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(argmin_survivors_batch_kernel, dim3(nq), dim3(1024), 0, stream, mb);
    return hipGetLastError();
}

int sc_align_filter_enabled() { return align_filter_enabled(); }

bool sc_distance_fuses_ring(const DbView &db, int SR)
{
    const int W = 2 * SR + 1;
    if (scl_lab_is("SCL_SC_KERNEL", "v1")) return false;
    return ((db.RG == 5 && W == 7 && db.S == 60) || (db.RG == 16 && W == 13 && db.S == 120)) && (db.R % 4 == 0);
}

hipError_t launch_argmin(const double *dist, const int *shift, int n, double *out3, hipStream_t stream)
{
    hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(1024), 0, stream, dist, shift, n, out3);
    return hipGetLastError();
}

namespace {

// Full-DB epilogue in one launch: arg-min over the SC distances AND the k nearest ring keys from
// the squared ring distances the SC kernel produced on the side.  k rounds of "smallest key larger
// than the previous pick" (keys (d2 bits << 32 | i) are unique), so nothing has to be removed.
__global__ __launch_bounds__(1024) void full_epilogue_kernel(const double *dist, const int *shift, const float *d2,
                                                             int n, int slot_base, int k, float exclude_eps,
                                                             double *out3, int *topk_idx, float *topk_d2)
{
    __shared__ double sv[16];
    __shared__ int si[16];
    __shared__ unsigned long long sk[16];
    const int wv = threadIdx.x / kWave;
    double best = __longlong_as_double(0x7ff0000000000000LL);
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double d = dist[i];
        if (d < kBigDist && ((d < best) | ((d == best) & (i < bi)))) { best = d; bi = i; }
    }
    wave_argmin(best, bi);
    if ((threadIdx.x & (kWave - 1)) == 0) { sv[wv] = best; si[wv] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)blockDim.x / kWave; ++w)
            if ((sv[w] < best) | ((sv[w] == best) & (si[w] < bi))) { best = sv[w]; bi = si[w]; }
        const bool ok = best < kBigDist;
        out3[0] = ok ? best : kBigDist;
        out3[1] = ok ? (double)bi : -1.0;
        out3[2] = ok ? (double)shift[bi] : 0.0;
    }
    const unsigned long long none = ~0ull;
    unsigned long long prev = 0ull;
    bool first = true;
    for (int round = 0; round < k; ++round) {
        unsigned long long mine = none;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const float r = d2[i];
            const bool excluded = (exclude_eps > 0.0f) && (r <= exclude_eps);
            if (excluded || !(r < 3.402823466e+38f)) continue;
            const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(r) << 32) | (unsigned)i;
            if ((first || key > prev) && key < mine) mine = key;
        }
        mine = wave_min_u64(mine);
        __syncthreads();
        if ((threadIdx.x & (kWave - 1)) == 0) sk[wv] = mine;
        __syncthreads();
        unsigned long long m = sk[0];
        for (int w = 1; w < (int)blockDim.x / kWave; ++w) m = sk[w] < m ? sk[w] : m;
        if (threadIdx.x == 0) {
            if (m == none) { topk_idx[round] = -1; topk_d2[round] = 3.402823466e+38f; }
            else { topk_idx[round] = slot_base + (int)(unsigned)(m & 0xffffffffull); topk_d2[round] = __int_as_float((int)(m >> 32)); }
        }
        if (m == none) {
            for (int r2 = round + 1 + (int)threadIdx.x; r2 < k; r2 += blockDim.x) { topk_idx[r2] = -1; topk_d2[r2] = 3.402823466e+38f; }
            break;
        }
        prev = m; first = false;
    }
}

}  // namespace

hipError_t launch_full_epilogue(const double *dist, const int *shift, const float *ring_d2, int n, int slot_base,
                                int k, float exclude_eps, double *out3, int *topk_idx, float *topk_d2,
                                hipStream_t stream)
{
    hipLaunchKernelGGL(full_epilogue_kernel, dim3(1), dim3(1024), 0, stream, dist, shift, ring_d2, n, slot_base, k,
                       exclude_eps, out3, topk_idx, topk_d2);
    return hipGetLastError();
}

}  // namespace scl
