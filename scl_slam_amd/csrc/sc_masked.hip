// sc_masked.hip -- the exact fp64 distance of a (scan, keyframe) pair evaluated ONLY at the shifts that can still hold its minimum.
//
// distanceBtnScanContext (descriptor.h:1538-1569) returns the minimum over 2 SR + 1 shifted cosine distances and its shift.  The
// screening pass (sc_screen.hip) has the pair's first shift -- the reference's own alignment, exactly -- and every shifted distance
// within +-kScreenEps.  A shift whose screened distance lies more than 2 kScreenEps above the pair's smallest screened distance cannot
// be the arg-min (and cannot tie with it): the finishing kernel hands over the set of the others as a bit mask (bit t: shift
// (first + t) mod S), one to three of 13 as a rule.  This kernel evaluates exactly those, in the reference's arithmetic -- the ring-order
// fp64 dot (products of widened floats are exact: fma == mul + add), the quotient by the two column norms, the sum over the sectors in
// ascending sector order, 1 - sum / n_eff, strict < over the shifts in ascending shift value -- so distance and shift are the
// reference's, bit for bit, at a fraction of the 13-shift program's work.  Pairs the screening cannot bound (mask = every shift) cost
// what they always did.
//
// One wave per pair.  Lane l < 60 owns the candidate's columns (l + 60 j - first) mod S, j < S / 60, which meet the scan's columns
// l + 60 j + t at the shift (first + t): the candidate streams from memory (float4 ring groups, coalesced along the lanes), the scan
// is staged once per workgroup in LDS, column-major with a pitch of 4 R + 16 bytes (consecutive columns = consecutive lanes hit
// consecutive 16-byte slots: no bank conflicts).  Up to TMAX shifts of a pair are evaluated in one pass over the candidate; the
// per-shift sums over the sectors are walked by one lane each out of a wave-private LDS row.
//
// Callers: the exact distance MATRIX (scl_sc_distance_matrix on the screened grids: every pair of a batch of scans), and the exact
// pass over the survivors of the 80 x 180 grid (the wave program that grid lacked).
#include <atomic>
#include <cstdio>

#include "align_exact.hpp"
#include "device_common.hpp"
#include "kernels.hpp"
#include "topk_merge.hpp"

namespace scl {

namespace {

__device__ __forceinline__ void wave_fence_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef MASKED_PD
#define MASKED_PD 2
#endif
constexpr int kMaskLanes = 60;                 // active lanes: S / 60 columns per lane (S = 60, 120, 180)
#ifndef MASKED_TM
#define MASKED_TM 4
#endif
constexpr int kMaskTMax = MASKED_TM;         // shifts evaluated per pass over the candidate

template <int RG, int S, int W>
struct MaskedCfg {
    static constexpr int CPL = S / kMaskLanes;                     // columns per lane
    static constexpr int R4 = 4 * RG;
    static constexpr int PITCH = R4 * 4 + 16;                      // bytes per staged scan column
    static constexpr int QCOLS = S + W - 1;                        // columns 0 .. S-1 and the first W-1 again
    static constexpr int WAVES = 8;
    static constexpr size_t LDS_Q = (size_t)QCOLS * PITCH;         // the scan, fp32
    static constexpr size_t LDS_N = (size_t)QCOLS * 8;             // its column norms, fp64 (extended alike)
    static constexpr size_t LDS_WAVE = (size_t)kMaskTMax * S * 8;  // per wave: TMAX rows of S similarities
    static constexpr size_t LDS = LDS_Q + LDS_N + WAVES * LDS_WAVE;
};

// The exact distance of ONE pair at the shifts of `mask` (bit t: shift (first + t) mod S), by one wave: the scan staged in LDS (Qs:
// column-major fp32 with QCOLS columns, nq: its norms extended alike), the candidate's fp32 rows kd and norms kn from memory, simrow:
// the wave's TM rows of S similarities.  best / bshift: the smallest distance and its shift VALUE (ties: the lowest), kInf / INT_MAX
// when no shift has a finite distance.
// ALL: the caller opens every one of TM = W shifts (mask = all ones: slot u is shift first + u): the dots then run without branches, the
// scan's LDS reads one slot ahead of the products that use them, every address a base register + an immediate -- for a lone wave
// (sc_cand_exact_kernel: one wave per candidate, nothing else on the CU) the read latency is otherwise paid 26 times per ring group.
template <int RG, int S, int W, int PD = MASKED_PD, int TM = kMaskTMax, bool ALL = false>
__device__ __forceinline__ void masked_pair(const unsigned char *Qs, const double *nq, double *simrow, const float4 *kd, const double *kn,
                                            const int first, unsigned int mask, const int lane, double &best, int &bshift)
{
    using C = MaskedCfg<RG, S, W>;
    constexpr int CPL = C::CPL, PITCH = C::PITCH;
    const bool active = lane < kMaskLanes;
    const int ll = active ? lane : kMaskLanes - 1;
    const double kInf = __longlong_as_double(0x7ff0000000000000LL);
    // candidate columns of this lane and their norms
    int yc[CPL];
    double nk[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) { int y = ll + kMaskLanes * j - first; y = y < 0 ? y + S : y; yc[j] = y; nk[j] = kn[y]; }
    best = kInf; bshift = 0x7fffffff;
    while (mask) {
        // the next up to TM open shifts of the pair
        int ts[TM]; int nt = 0;
#pragma unroll
        for (int u = 0; u < TM; ++u) { ts[u] = 0; if (mask) { ts[u] = __ffs((int)mask) - 1; mask &= mask - 1; nt = u + 1; } }
        double acc[TM][CPL];
#pragma unroll
        for (int u = 0; u < TM; ++u)
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[u][j] = 0.0;
        // ---- ring-order dots: the candidate's ring groups (one ahead), the scan's from LDS ----
        // (PD ring groups in flight: one group ahead left the wave waiting a memory round trip per group)
        float4 kbuf[PD][CPL];
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int j = 0; j < CPL; ++j) kbuf[d][j] = kd[(size_t)(d < RG ? d : RG - 1) * S + yc[j]];
        static_assert(RG % PD == 0, "ring groups in whole prefetch rounds");
        if constexpr (ALL) {
            const unsigned char *qb[CPL];                                       // the scan's column that meets this lane's at slot 0
#pragma unroll
            for (int j = 0; j < CPL; ++j) qb[j] = Qs + (size_t)(ll + kMaskLanes * j) * PITCH;
            float4 qv[2][CPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) qv[0][j] = *reinterpret_cast<const float4 *>(qb[j]);
#pragma unroll 1
            for (int r0 = 0; r0 < RG; r0 += PD) {
#pragma unroll
                for (int dd = 0; dd < PD; ++dd) {
                    const int rg = r0 + dd;
                    double kx[CPL][4];
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const float4 kv = kbuf[dd][j];
                        kx[j][0] = (double)kv.x; kx[j][1] = (double)kv.y; kx[j][2] = (double)kv.z; kx[j][3] = (double)kv.w;
                    }
                    {
                        const int rn = rg + PD < RG ? rg + PD : RG - 1;
#pragma unroll
                        for (int j = 0; j < CPL; ++j) kbuf[dd][j] = kd[(size_t)rn * S + yc[j]];
                    }
#pragma unroll
                    for (int u = 0; u < TM; ++u) {
                        const int cur = (dd * TM + u) & 1, nxt = cur ^ 1;     // (compile time: a block starts with its first slot in set 0)
                        // the slot that follows: the next shift of this ring group, or the first shift of the next ring group (past the
                        // last group: the column's padding, unused)
                        const int un = u + 1 < TM ? u + 1 : 0, off = un * PITCH + (u + 1 < TM ? dd : dd + 1) * 16;
#pragma unroll
                        for (int j = 0; j < CPL; ++j) qv[nxt][j] = *reinterpret_cast<const float4 *>(qb[j] + off);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < CPL; ++j) {
                            const float4 q = qv[cur][j];
                            acc[u][j] = fma(kx[j][0], (double)q.x, acc[u][j]);
                            acc[u][j] = fma(kx[j][1], (double)q.y, acc[u][j]);
                            acc[u][j] = fma(kx[j][2], (double)q.z, acc[u][j]);
                            acc[u][j] = fma(kx[j][3], (double)q.w, acc[u][j]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if ((PD * TM) & 1) {                                            // an odd number of slots per block: the prefetched one moves to set 0
#pragma unroll
                    for (int j = 0; j < CPL; ++j) qv[0][j] = qv[1][j];
                }
#pragma unroll
                for (int j = 0; j < CPL; ++j) qb[j] += PD * 16;
            }
        } else {
#pragma unroll 1
        for (int r0 = 0; r0 < RG; r0 += PD) {
#pragma unroll
          for (int dd = 0; dd < PD; ++dd) {
            const int rg = r0 + dd;
            float4 kv[CPL];
#pragma unroll
            for (int j = 0; j < CPL; ++j) kv[j] = kbuf[dd][j];
            {   // the slot is refilled with the group PD further on (past the end: the last group again, unused)
                const int rn = rg + PD < RG ? rg + PD : RG - 1;
#pragma unroll
                for (int j = 0; j < CPL; ++j) kbuf[dd][j] = kd[(size_t)rn * S + yc[j]];
            }
#pragma unroll
            for (int u = 0; u < TM; ++u) {
                if (u < nt) {                                               // (wave uniform)
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const float4 qv = *reinterpret_cast<const float4 *>(Qs + (size_t)(ll + kMaskLanes * j + ts[u]) * PITCH + rg * 16);
                        acc[u][j] = fma((double)kv[j].x, (double)qv.x, acc[u][j]);
                        acc[u][j] = fma((double)kv[j].y, (double)qv.y, acc[u][j]);
                        acc[u][j] = fma((double)kv[j].z, (double)qv.z, acc[u][j]);
                        acc[u][j] = fma((double)kv[j].w, (double)qv.w, acc[u][j]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        }
        // ---- cosine similarity per (shift, scan column), into the wave's rows by scan column ----
        wave_fence_lds();
#pragma unroll
        for (int u = 0; u < TM; ++u) {
            if (u < nt && active) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const int cx = ll + kMaskLanes * j + ts[u];             // < S + W - 1
                    const int c = cx >= S ? cx - S : cx;
                    const double nqc = nq[cx];
                    const bool skip = (nqc == 0.0) | (nk[j] == 0.0);        // D.h:1523
                    // a skipped column contributes +0.0 to the sum, which is bit-equivalent to leaving it out
                    simrow[u * S + c] = skip ? 0.0 : acc[u][j] / (nqc * nk[j]);
                }
            }
        }
        // effective columns per shift (both norms non-zero), counted over the lanes
        int eff[TM];
#pragma unroll
        for (int u = 0; u < TM; ++u) {
            int e = 0;
            if (u < nt) {
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    const int cx = ll + kMaskLanes * j + ts[u];
                    const bool use = active && !((nq[cx] == 0.0) | (nk[j] == 0.0));
                    e += __popcll(__builtin_amdgcn_ballot_w64(use));
                }
            }
            eff[u] = e;
        }
        wave_fence_lds();
        // ---- the sum over the scan's columns in ascending order (D.h:1518-1532), one lane per shift ----
        double d = kInf; int sh = 0x7fffffff;
        if (lane < nt) {
            const double *row = simrow + lane * S;
            double sum = 0.0;
#pragma unroll 8
            for (int c = 0; c < S; ++c) sum = sum + row[c];
            int e = eff[0], t = ts[0];
#pragma unroll
            for (int u = 1; u < TM; ++u) if (lane == u) { e = eff[u]; t = ts[u]; }
            const double dd = 1.0 - sum / (double)e;                         // 0 / 0 -> NaN, never wins
            int st = first + t; st = st >= S ? st - S : st;
            if (dd < kBigDist) { d = dd; sh = st; }
        }
        // smallest distance, ties to the lowest shift VALUE (the reference walks the sorted shift space with strict <)
#pragma unroll
        for (int off = 1; off < TM; off <<= 1) {
            const double od = __shfl_xor(d, off, kWave); const int os = __shfl_xor(sh, off, kWave);
            const bool take = (od < d) | ((od == d) & (os < sh));
            d = take ? od : d; sh = take ? os : sh;
        }
        d = readlane_f64(d, 0); sh = __builtin_amdgcn_readfirstlane(sh);
        if ((d < best) | ((d == best) & (sh < bshift))) { best = d; bshift = sh; }
    }
}

template <int RG, int S, int W>
__global__ __launch_bounds__((MaskedCfg<RG, S, W>::WAVES * kWave)) void sc_masked_kernel(MaskedArgs ma)
{
    using C = MaskedCfg<RG, S, W>;
    constexpr int CPL = C::CPL, PITCH = C::PITCH, QCOLS = C::QCOLS;
    static_assert(S % kMaskLanes == 0 && CPL >= 1 && CPL <= 3 && W <= 32, "tiling");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_m[];
    unsigned char *Qs = smem_m;
    double *nq = reinterpret_cast<double *>(smem_m + C::LDS_Q);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double *simrow = reinterpret_cast<double *>(smem_m + C::LDS_Q + C::LDS_N + (size_t)wave * C::LDS_WAVE);   // [TM][S]

    // workgroup -> (query, part): the workgroups of a launch's queries that walk the same part of the range sit next to each other on
    // one XCD (index & 7) and start together: all but the first find the keyframes' rows in that XCD's L2
    const int b = (int)blockIdx.x, xcd = b & 7, jx = b >> 3;
    const int qi = ma.spread ? b % ma.nq : jx % ma.nq;
    const int part = ma.spread ? b / ma.nq : (jx / ma.nq) * 8 + xcd;
    if (part >= ma.parts) return;
    const MaskedQuery mq = ma.q[qi];
    const int n_items = mq.n_dev ? *mq.n_dev : mq.n;
    if (part * C::WAVES >= n_items && part > 0) return;            // (a short survivor list: nothing for this workgroup)

    // ---- stage the scan: columns 0 .. S + W - 2 (column-major, fp32) and their norms ----
    {
        const float4 *qd = ma.desc + (size_t)mq.qslot * (size_t)(RG * S);
        for (int idx = threadIdx.x; idx < RG * QCOLS; idx += blockDim.x) {
            const int rg = idx / QCOLS, cx = idx - rg * QCOLS;
            const int c = cx < S ? cx : cx - S;
            *reinterpret_cast<float4 *>(Qs + (size_t)cx * PITCH + rg * 16) = qd[(size_t)rg * S + c];
        }
        const double *qn = ma.norm + (size_t)mq.qslot * S;
        for (int cx = threadIdx.x; cx < QCOLS; cx += blockDim.x) nq[cx] = qn[cx < S ? cx : cx - S];
    }
    __syncthreads();

    const int stride_items = ma.parts * C::WAVES;
    for (int item = part * C::WAVES + wave; item < n_items; item += stride_items) {
        // ---- the pair: candidate slot, first searched shift, the shifts still open ----
        const int pos = mq.cand ? mq.cand[item] - mq.slot_base : item;         // position in the scan's range (starts / masks are indexed by it)
        const int slot = mq.cand ? mq.cand[item] : mq.slot_base + item;
        int first = mq.starts ? mq.starts[pos] : 0;
        unsigned int mask = mq.smask ? mq.smask[pos] : 0u;
        const unsigned int all = W >= 32 ? 0xffffffffu : ((1u << W) - 1u);
        if (first < 0 || first >= S) { first = 0; mask = 0u; }                  // (an undecided alignment never reaches this kernel: guard only)
        mask &= all;
        const double *kn = ma.norm + (size_t)slot * S;
        const float4 *kd = ma.desc + (size_t)slot * (size_t)(RG * S);
        double best; int bshift;
        masked_pair<RG, S, W>(Qs, nq, simrow, kd, kn, first, mask, lane, best, bshift);
        if (lane == 0) {
            const bool ok = best < kBigDist;
            mq.out_dist[item] = ok ? best : kBigDist;
            mq.out_shift[item] = ok ? bshift : 0;
        }
    }
}

template <int RG, int S, int W>
hipError_t launch_masked(const MaskedArgs &ma, hipStream_t stream)
{
    using C = MaskedCfg<RG, S, W>;
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<bool> &attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)sc_masked_kernel<RG, S, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    const int grid = ma.spread ? ma.parts * ma.nq : 8 * ((ma.parts + 7) / 8) * ma.nq;
    hipLaunchKernelGGL((sc_masked_kernel<RG, S, W>), dim3(grid), dim3(C::WAVES * kWave), C::LDS, stream, ma);
    return hipGetLastError();
}


// ---- the exact pass of a SMALL batch of screened scans (one to four: a blocking call over the whole database) --------------------
// One workgroup per scan does everything the big survivors' pass spreads over launches and workgroups: it reads the smallest
// screened distance, lists the keyframes within the margin of it, scores those few at their open shifts (masked_pair: the first
// shift is the alignment's, the mask the screening's), forms the ring-key top-k from the metric the screening stored, and writes
// the winner to pinned host memory -- one launch, arguments in the kernel-argument segment (the
// survivors' pass of a chunk uploads its argument sets with a copy of its own: 5 us of a blocking call).  A pair that reaches
// it without a first shift (nothing writes kAlignUndecided now; the consumers honour it) is aligned here by the reference's own
// fp64 evaluation and scored at every shift.
constexpr int kSmallWaves = 8;                  // (512 threads: registers for kSmallPD ring groups in flight)
constexpr int kSmallPD = 4;                     // ring groups in flight per wave (8 spill)
constexpr int kSmallList = 1024;                // listed keyframes kept in LDS with their first shift and mask (more: through memory)
constexpr int kSmallTop = 16;                   // ring-key candidates the barrier-free top-k tracks per wave
// EVERY: the instance for scans whose survivors come without shift masks (the 64 x 120 stream): W rows of sums per wave, every shift in one
// pass.  The blocking call's instance (masks, a handful of open shifts) keeps the small rows: with both bodies in one kernel the blocking
// scan took 5 us longer (150 KB of LDS to allocate, twice the code for a launch that runs once).
template <int RG, int S, int W, bool EVERY = false>
struct SmallCfg {
    using M = MaskedCfg<RG, S, W>;
    static constexpr size_t LDS_Q = M::LDS_Q, LDS_N = M::LDS_N;
    // every shift of a survivor in one pass needs W rows of sums per wave: 64 x 120 has the room (150 KB in all), 80 x 180 has not (its
    // launches always form shift masks: passes of kMaskTMax open shifts)
    static constexpr bool kFits = LDS_Q + LDS_N + (size_t)kSmallWaves * W * S * 8 + (size_t)kSmallList * 12 + 4096 <= 160 * 1024;
    static constexpr bool kAll = EVERY && kFits;
    static constexpr size_t ROWS_WAVE = kAll ? (size_t)W * S * 8 : M::LDS_WAVE;
    static constexpr size_t LDS_WAVE = ROWS_WAVE > (size_t)(2 * S + 2) * 8 ? ROWS_WAVE : (size_t)(2 * S + 2) * 8;   // rows of sums, or the doubled key of the slow path
    static constexpr size_t LDS_VQ = (size_t)S * 8;                  // the scan's sector key (slow path)
    static constexpr size_t LDS_REC = (size_t)kSmallWaves * 16 + (size_t)kSmallWaves * kSmallTop * 8;   // a wave's best (distance, position << 8 | shift); its kSmallTop nearest ring keys
    static constexpr size_t LDS_LIST = (size_t)kSmallList * 12;
    static constexpr size_t LDS = LDS_Q + LDS_N + kSmallWaves * LDS_WAVE + LDS_VQ + LDS_REC + 16 + LDS_LIST;
};

template <int RG, int S, int W, bool EVERY>
__global__ __launch_bounds__(kSmallWaves * kWave) void sc_small_exact_kernel(SmallExactArgs sa)
{
    using C = MaskedCfg<RG, S, W>;
    using SC = SmallCfg<RG, S, W, EVERY>;
    constexpr int PITCH = C::PITCH, QCOLS = C::QCOLS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    unsigned char *Qs = smem_s;
    double *nq = reinterpret_cast<double *>(smem_s + SC::LDS_Q);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double *wrow = reinterpret_cast<double *>(smem_s + SC::LDS_Q + SC::LDS_N + (size_t)wave * SC::LDS_WAVE);
    double *vq = reinterpret_cast<double *>(smem_s + SC::LDS_Q + SC::LDS_N + (size_t)kSmallWaves * SC::LDS_WAVE);
    unsigned long long *rec = reinterpret_cast<unsigned long long *>(smem_s + SC::LDS_Q + SC::LDS_N + (size_t)kSmallWaves * SC::LDS_WAVE + SC::LDS_VQ);
    unsigned long long *tkw = rec + 2 * kSmallWaves;                             // [kSmallWaves][kSmallTop]
    int *n_list = reinterpret_cast<int *>(tkw + kSmallWaves * kSmallTop);
    int *llist = n_list + 4;                                                     // [kSmallList][3]: position, first shift, mask
    const SmallExactQuery q = sa.q_dev ? sa.q_dev[blockIdx.x] : sa.q[blockIdx.x];
    const double kInf = __longlong_as_double(0x7ff0000000000000LL);

    // ---- threshold, then the scan's rows and norms on their way while the list is made ----
    const unsigned int tm = *q.t_min;
    float thr = __int_as_float(0xff800000);                                      // nothing screened: only the "score exactly" marks pass
    if (tm != 0xffffffffu) {
        const unsigned int b = (tm >> 31) ? (tm & 0x7fffffffu) : ~tm;            // inverse of the ordered image
        const unsigned int ew = q.t_min[kTminEpsOffset];                       // the launch's largest per-pair bound (0: not recorded)
        thr = __int_as_float((int)b) + (ew ? fminf(2.0f * __uint_as_float(ew) * 1.0001f, sa.two_eps) : sa.two_eps);
    }
    if (threadIdx.x == 0) *n_list = 0;
    {
        const float4 *qd = sa.desc + (size_t)q.qslot * (size_t)(RG * S);
        for (int idx = threadIdx.x; idx < RG * QCOLS; idx += blockDim.x) {
            const int rg = idx / QCOLS, cx = idx - rg * QCOLS;
            const int c = cx < S ? cx : cx - S;
            *reinterpret_cast<float4 *>(Qs + (size_t)cx * PITCH + rg * 16) = qd[(size_t)rg * S + c];
        }
        const double *qn = sa.norm + (size_t)q.qslot * S;
        for (int cx = threadIdx.x; cx < QCOLS; cx += blockDim.x) nq[cx] = qn[cx < S ? cx : cx - S];
        const double *qv = sa.vkey + (size_t)q.qslot * S;
        for (int c = threadIdx.x; c < S; c += blockDim.x) vq[c] = qv[c];
    }
    __syncthreads();
    // ---- one sweep over the range, every thread's loads in flight together (a loop that waited for each load took 100 us here):
    // the keyframes that can still hold the minimum go on the list (positions in the range; any order: the winner is the smallest
    // (distance, position)); the ring-key metric stays in registers for the top-k rounds ----
    constexpr int U = 20, NT = kSmallWaves * kWave;                             // 10 240 positions per sweep
    const unsigned long long none = ~0ull;
    const int sweeps = (q.n + U * NT - 1) / (U * NT);
    unsigned long long tk_prev = 0ull;                                           // (top-k over several sweeps: the rounds below re-read memory)
    float rv[U];
    for (int sw = 0; sw < sweeps; ++sw) {
        float av[U];
        // (UNCONDITIONAL loads from a clamped position, the out-of-range values substituted afterwards: `i < n ? load : constant` made
        //  every load a block of its own that ended with s_waitcnt vmcnt(0) -- forty dependent round trips where this comment's first
        //  line promised all of them in flight, most of the kernel's 22 us; round 5, found in the descriptor scatter first)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = sw * U * NT + u * NT + (int)threadIdx.x, ic = i < q.n ? i : q.n - 1;
            av[u] = q.approx[ic];
            rv[u] = q.ring_d2[ic];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = sw * U * NT + u * NT + (int)threadIdx.x;
            av[u] = i < q.n ? av[u] : __int_as_float(0x7f800000);
            rv[u] = i < q.n ? rv[u] : 3.402823466e+38f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = sw * U * NT + u * NT + (int)threadIdx.x;
            if (i < q.n && av[u] <= thr) {                                      // -inf (score exactly) always passes
                const int at = atomicAdd(n_list, 1);
                if (at < kSmallList) { llist[3 * at] = i; llist[3 * at + 1] = q.starts[i]; llist[3 * at + 2] = q.smask ? (int)q.smask[i] : 0; }
                else q.list[at] = i;
            }
        }
    }
    // ---- the ring-key top-k of the range over the metric the screening stored; keys (d2 bits << 32 | position) are unique.  One sweep
    // (n <= 10 240) and k <= kSmallTop: every wave picks ITS k smallest out of the registers (k rounds of a wave minimum, no barrier),
    // thread 0 merges the waves' picks behind the one barrier below.  Otherwise: k rounds over the whole workgroup, from memory ----
    const bool topk_fast = sa.k > 0 && sa.k <= kSmallTop && sweeps == 1;
    if (topk_fast) {
        unsigned long long prev = 0ull;
        bool first = true;
        for (int round = 0; round < sa.k; ++round) {
            unsigned long long mine = none;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = u * NT + (int)threadIdx.x;
                const float r = rv[u];
                const bool excluded = (sa.exclude_eps > 0.0f) && (r <= sa.exclude_eps);
                const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(r) << 32) | (unsigned)i;
                if (!excluded && (r < 3.402823466e+38f) && (first || key > prev) && key < mine) mine = key;
            }
            mine = wave_min_u64(mine);
            if (lane == 0) tkw[wave * kSmallTop + round] = mine;
            prev = mine; first = false;                                          // (none: every further round finds none as well)
            if (mine == none) { for (int r2 = round + 1; r2 < sa.k; ++r2) if (lane == 0) tkw[wave * kSmallTop + r2] = none; break; }
        }
    } else if (sa.k > 0) {
        bool first = true;
        for (int round = 0; round < sa.k; ++round) {
            unsigned long long mine = none;
            for (int i = (int)threadIdx.x; i < q.n; i += NT) {
                const float r = q.ring_d2[i];
                const bool excluded = (sa.exclude_eps > 0.0f) && (r <= sa.exclude_eps);
                const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(r) << 32) | (unsigned)i;
                if (!excluded && (r < 3.402823466e+38f) && (first || key > tk_prev) && key < mine) mine = key;
            }
            mine = wave_min_u64(mine);
            __syncthreads();                                                     // (the round before has been read)
            if (lane == 0) rec[wave] = mine;
            __syncthreads();
            unsigned long long m = rec[0];
#pragma unroll
            for (int w = 1; w < kSmallWaves; ++w) m = rec[w] < m ? rec[w] : m;
            if (threadIdx.x == 0) {
                if (m == none) { q.topk_idx[round] = -1; q.topk_d2[round] = 3.402823466e+38f; }
                else { q.topk_idx[round] = q.base + (int)(unsigned)(m & 0xffffffffull); q.topk_d2[round] = __int_as_float((int)(m >> 32)); }
            }
            if (m == none) {
                for (int r2 = round + 1 + (int)threadIdx.x; r2 < sa.k; r2 += NT) { q.topk_idx[r2] = -1; q.topk_d2[r2] = 3.402823466e+38f; }
                break;
            }
            tk_prev = m; first = false;
        }
    }
    __threadfence_block();
    __syncthreads();
    const int total = *n_list;
    if (topk_fast && threadIdx.x == 0) {                                         // the k smallest of the waves' k smallest (each list ascending)
        int head[kSmallWaves];
#pragma unroll
        for (int w = 0; w < kSmallWaves; ++w) head[w] = 0;
        for (int round = 0; round < sa.k; ++round) {
            unsigned long long m = none; int mw = -1;
#pragma unroll
            for (int w = 0; w < kSmallWaves; ++w) {
                const unsigned long long c = head[w] < sa.k ? tkw[w * kSmallTop + head[w]] : none;
                if (c < m) { m = c; mw = w; }
            }
            if (mw < 0) { q.topk_idx[round] = -1; q.topk_d2[round] = 3.402823466e+38f; continue; }
#pragma unroll
            for (int w = 0; w < kSmallWaves; ++w) if (w == mw) ++head[w];
            q.topk_idx[round] = q.base + (int)(unsigned)(m & 0xffffffffull); q.topk_d2[round] = __int_as_float((int)(m >> 32));
        }
    }

    // ---- the listed keyframes, one wave each at a time ----
    unsigned long long my_d = ~0ull, my_ps = ~0ull;                              // this wave's best: ordered image of the distance, position << 8 | shift
    for (int it = wave; it < total; it += kSmallWaves) {
        int pos, first; unsigned int mask;
        if (it < kSmallList) { pos = llist[3 * it]; first = llist[3 * it + 1]; mask = (unsigned int)llist[3 * it + 2]; }
        else { pos = q.list[it]; first = q.starts[pos]; mask = q.smask ? q.smask[pos] : 0u; }
        const int slot = q.base + pos;
        const unsigned int all = W >= 32 ? 0xffffffffu : ((1u << W) - 1u);
        const double *kn = sa.norm + (size_t)slot * S;
        const float4 *kd = sa.desc + (size_t)slot * (size_t)(RG * S);
        if (first < 0 || first >= S || !q.smask) {
            // no first shift (or no masks at all): the reference's own alignment (fastAlignUsingVkey, D.h:1491-1511), every shift open
            if (first < 0 || first >= S) {
                int a0;
                if constexpr (S / 2 <= kWave) {
                    const int ll2 = lane < (S >> 1) ? lane : (S >> 1) - 1;
                    const double2 vk = *reinterpret_cast<const double2 *>(sa.vkey + (size_t)slot * S + 2 * ll2);
                    a0 = align_keyframe_exact<S>(vk, lane, wrow, vq);
                } else {
                    constexpr int SPL = (S + kWave - 1) / kWave, LA = S / SPL;
                    const int ll2 = lane < LA ? lane : LA - 1;
                    double vk[SPL];
#pragma unroll
                    for (int u = 0; u < SPL; ++u) vk[u] = sa.vkey[(size_t)slot * S + SPL * ll2 + u];
                    a0 = align_keyframe_wide<S>(vk, lane, wrow, vq);
                }
                // D.h:1545-1551: the searched shifts start SEARCH_RADIUS below the aligned one
                first = a0 - (W - 1) / 2; first = first < 0 ? first + S : first;
                wave_fence_lds();
            }
            mask = all;
        }
        mask &= all;
        double best; int bshift;
        // every shift open (the stream form's launches form no masks; an undecided alignment): all W shifts in ONE pass over the keyframe,
        // the branch-free form of the candidates' kernel -- in passes of four shift slots a survivor of the stream cost its rows four times
        // (44 us per chunk's launch on its own instead of 25)
        constexpr int PDS = C::CPL >= 3 ? 2 : kSmallPD;                           // (three columns per lane, 80 x 180: four ring groups in flight spill)
        if constexpr (SC::kAll) {
            if (mask == all) masked_pair<RG, S, W, PDS, W, true>(Qs, nq, wrow, kd, kn, first, mask, lane, best, bshift);
            else masked_pair<RG, S, W, PDS>(Qs, nq, wrow, kd, kn, first, mask, lane, best, bshift);   // (a lone wave: every round trip counts)
        } else masked_pair<RG, S, W, PDS>(Qs, nq, wrow, kd, kn, first, mask, lane, best, bshift);
        if (best < kBigDist) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(best);
            const unsigned long long od = (b >> 63) ? ~b : (b | 0x8000000000000000ull);     // IEEE order -> unsigned order
            const unsigned long long ps = ((unsigned long long)(unsigned)pos << 8) | (unsigned long long)(bshift & 0xff);
            if (od < my_d || (od == my_d && ps < my_ps)) { my_d = od; my_ps = ps; }
        }
    }
    __syncthreads();                                                             // (rec: the top-k rounds' words have been read)
    if (lane == 0) { rec[2 * wave] = my_d; rec[2 * wave + 1] = my_ps; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long bd = ~0ull, bp = ~0ull;
        for (int w = 0; w < kSmallWaves; ++w) {
            const unsigned long long d = rec[2 * w], p = rec[2 * w + 1];
            if (d < bd || (d == bd && p < bp)) { bd = d; bp = p; }
        }
        volatile double *o = q.out3;
        if (bd == ~0ull) { o[0] = kBigDist; o[1] = -1.0; o[2] = 0.0; }
        else {
            const unsigned long long b = (bd >> 63) ? (bd & 0x7fffffffffffffffull) : ~bd;
            o[0] = __longlong_as_double((long long)b); o[1] = (double)(int)(bp >> 8); o[2] = (double)(int)(bp & 0xff);
        }
        if (sa.q_dev) o[3] = (double)total;
        if (sa.seq) { __threadfence_system(); o[4] = (double)sa.seq; }             // (the record is complete in host memory before its sequence number)
                                             // (the stream form's result records have eight words: the list's length for the host)
        *q.t_min = 0xffffffffu; q.t_min[kTminEpsOffset] = 0u;                    // re-armed for the next screening pass of this buffer set
        if (sa.surv_stats) {
            atomicAdd(sa.surv_stats, (unsigned long long)total);
            atomicMax(sa.surv_stats + 1, (unsigned long long)total);
            atomicAdd(sa.surv_stats + 2, 1ull);
        }
    }
    (void)kInf;
}


// ---- the k candidates of a ring-key search, scored (the reference-faithful detection: D.h:1642-1659, 1710-1737) ----------------
// One workgroup: the scan staged once, one wave per candidate -- the reference's own alignment (fastAlignUsingVkey: align_keyframe_exact),
// then all W shifts in ONE pass over the candidate (masked_pair with W slots) -- and the k results written straight into pinned host
// memory beside the candidates' indices and ring-key distances (idx[k] | d2[k] | dist[k] | shift[k]).  It replaces the 13-shift wave
// program's launch (126 KB of fp64 scan staged for three candidates: 19-20 us) and the packing launch behind it (4 us).
constexpr int kCandWaves = 8;
template <int RG, int S, int W>
struct CandCfg {
    using M = MaskedCfg<RG, S, W>;
    static constexpr size_t LDS_Q = M::LDS_Q, LDS_N = M::LDS_N;
    static constexpr size_t LDS_WAVE = (size_t)W * S * 8 > (size_t)(2 * S + 2) * 8 ? (size_t)W * S * 8 : (size_t)(2 * S + 2) * 8;   // W rows; the alignment's doubled key first
    static constexpr size_t LDS = LDS_Q + LDS_N + kCandWaves * LDS_WAVE + (size_t)S * 8;
};

#ifdef SCL_DIAGNOSTICS
__device__ unsigned long long g_cand_stamps[16];      // ticks (100 MHz) of thread 0 between the candidates' kernel's phases; [15] = launches
#define CAND_STAMP(k) do { if (threadIdx.x == 0) { const unsigned long long now__ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_cand_stamps[k], now__ - cand_t0__); cand_t0__ = now__; } } while (0)
#else
#define CAND_STAMP(k) ((void)0)
#endif
template <int RG, int S, int W>
__global__ __launch_bounds__(kCandWaves * kWave) void sc_cand_exact_kernel(CandExactArgs ca)
{
#ifdef SCL_DIAGNOSTICS
    unsigned long long cand_t0__ = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) atomicAdd(&g_cand_stamps[15], 1ull);
#endif
    using C = MaskedCfg<RG, S, W>;
    using CC = CandCfg<RG, S, W>;
    constexpr int PITCH = C::PITCH, QCOLS = C::QCOLS, SR = (W - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];
    unsigned char *Qs = smem_c;
    double *nq = reinterpret_cast<double *>(smem_c + CC::LDS_Q);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double *wrow = reinterpret_cast<double *>(smem_c + CC::LDS_Q + CC::LDS_N + (size_t)wave * CC::LDS_WAVE);
    double *vq = reinterpret_cast<double *>(smem_c + CC::LDS_Q + CC::LDS_N + (size_t)kCandWaves * CC::LDS_WAVE);
    int *o_idx = reinterpret_cast<int *>(ca.out);
    float *o_d2 = reinterpret_cast<float *>(ca.out + sizeof(int) * ca.k);
    double *o_dist = reinterpret_cast<double *>(ca.out + (sizeof(int) + sizeof(float)) * ca.k);
    int *o_shift = reinterpret_cast<int *>(ca.out + (sizeof(int) + sizeof(float) + sizeof(double)) * ca.k);
    // the candidates first (a dependent load otherwise), then the scan
    int my_slot = -1, my_slot2 = -1;                                             // candidates wave and wave + kCandWaves (k <= 16)
    // Every global read of the prologue first -- the scan (desc, norms, sector key) and the ring-key scan's lists --, then the LDS stores
    // and the merge: one memory round trip where "merge, then stage" was two (phase stamps: 3.4 + 2.3 us of the kernel's 25).
    constexpr int NT = kCandWaves * kWave, NL = (RG * QCOLS + NT - 1) / NT;
    static_assert(QCOLS <= NT && S <= NT && kCandMergeMaxKeys <= 2 * NT, "one norm, one key entry and two list keys per thread");
    float4 sqv[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        int idx = (int)threadIdx.x + u * NT; idx = idx < RG * QCOLS ? idx : RG * QCOLS - 1;
        const int rg = idx / QCOLS, cx = idx - rg * QCOLS;
        sqv[u] = ca.q_desc[(size_t)rg * S + (cx < S ? cx : cx - S)];
    }
    const int cxn = (int)threadIdx.x < QCOLS ? (int)threadIdx.x : QCOLS - 1;
    const double nqv = ca.q_norm[cxn < S ? cxn : cxn - S];
    const double vqv = ca.q_vkey[(int)threadIdx.x < S ? (int)threadIdx.x : S - 1];
    const int n_keys = ca.lists ? ca.n_lists * ca.k : 0;
    unsigned long long lk0 = ~0ull, lk1 = ~0ull;
    if (n_keys > 0) {
        lk0 = ca.lists[(int)threadIdx.x < n_keys ? (int)threadIdx.x : n_keys - 1];
        lk1 = ca.lists[(int)threadIdx.x + NT < n_keys ? (int)threadIdx.x + NT : n_keys - 1];
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
        const int idx = (int)threadIdx.x + u * NT;
        if (idx < RG * QCOLS) {
            const int rg = idx / QCOLS, cx = idx - rg * QCOLS;
            *reinterpret_cast<float4 *>(Qs + (size_t)cx * PITCH + rg * 16) = sqv[u];
        }
    }
    if ((int)threadIdx.x < QCOLS) nq[threadIdx.x] = nqv;
    if ((int)threadIdx.x < S) vq[threadIdx.x] = vqv;
    if (ca.lists) {
        // the ring-key scan left its per-workgroup lists: merged here (topk_merge.hpp), in the LDS of the waves' rows, which nothing uses yet
        unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem_c + CC::LDS_Q + CC::LDS_N);
        unsigned long long *scand = skey + kCandMergeMaxKeys;
        TopkMergeShared *msh = reinterpret_cast<TopkMergeShared *>(scand + kCandMergeMaxKeys);
        int *r_idx = reinterpret_cast<int *>(msh + 1);
        float *r_d2 = reinterpret_cast<float *>(r_idx + 16);
        if ((int)threadIdx.x < n_keys) skey[threadIdx.x] = lk0;
        if ((int)threadIdx.x + NT < n_keys) skey[threadIdx.x + NT] = lk1;
        topk_merge_lists(ca.lists, ca.n_lists, ca.k, skey, scand, msh, r_idx, r_d2, true);
        if (wave < ca.k) my_slot = r_idx[wave];
        if (wave + kCandWaves < ca.k) my_slot2 = r_idx[wave + kCandWaves];
        if (ca.k <= kCandWaves / 2 && wave >= kCandWaves / 2 && wave - kCandWaves / 2 < ca.k) my_slot = r_idx[wave - kCandWaves / 2];   // (two waves per candidate: below)
        if ((int)threadIdx.x < ca.k) {
            const int ci = r_idx[threadIdx.x]; const float cd = r_d2[threadIdx.x];
            o_idx[threadIdx.x] = ci; o_d2[threadIdx.x] = cd;
            ca.cand_idx_out[threadIdx.x] = ci; ca.cand_d2_out[threadIdx.x] = cd;
        }
        CAND_STAMP(0);
    } else {
        if (wave < ca.k) my_slot = ca.cand_idx[wave];
        if (wave + kCandWaves < ca.k) my_slot2 = ca.cand_idx[wave + kCandWaves];
        if (ca.k <= kCandWaves / 2 && wave >= kCandWaves / 2 && wave - kCandWaves / 2 < ca.k) my_slot = ca.cand_idx[wave - kCandWaves / 2];
    }
    if (!ca.lists && (int)threadIdx.x < ca.k) { o_idx[threadIdx.x] = ca.cand_idx[threadIdx.x]; o_d2[threadIdx.x] = ca.cand_d2[threadIdx.x]; }
    __syncthreads();
    CAND_STAMP(1);
    constexpr int PDC = RG % 4 == 0 ? 4 : RG;                                    // ring groups in flight (a lone wave: every round trip counts)
    if (ca.k <= kCandWaves / 2) {
        // Up to four candidates (the reference's three, D.h:1319): TWO waves per candidate.  One wave's 13 shifts were 14 of the kernel's
        // 25 us -- 1 664 fp64 products and as many conversions per candidate on ONE SIMD's fp64 pipe, with half of the workgroup's waves
        // idle.  Wave c aligns candidate c (the reference's own alignment) and scores its first seven shifts, wave c + 4 the other six;
        // the smaller (distance, shift value) of the two is the pair's -- masked_pair's own tie rule (the lowest shift value).
        constexpr int H = kCandWaves / 2, TA = (W + 1) / 2, TB = W - TA;
        __shared__ int s_first[H];
        __shared__ double s_b[kCandWaves];
        __shared__ int s_bs[kCandWaves];
        const int c = wave < H ? wave : wave - H;
        const bool mine = c < ca.k && my_slot >= 0;
        if (mine && wave < H) {
            constexpr int L = S >> 1;
            const int ll2 = lane < L ? lane : L - 1;
            const double2 vk = *reinterpret_cast<const double2 *>(ca.vkey + (size_t)my_slot * S + 2 * ll2);
            const int a0 = align_keyframe_exact<S>(vk, lane, wrow, vq);
            int first = a0 - SR; first = first < 0 ? first + S : first;          // D.h:1545-1551: the searched shifts start SEARCH_RADIUS below
            if (lane == 0) s_first[c] = first;
            wave_fence_lds();
        }
        CAND_STAMP(2);
        __syncthreads();
        double b = __longlong_as_double(0x7ff0000000000000LL); int bs = 0x7fffffff;
        if (mine) {
            int first = s_first[c];
            const float4 *kd = ca.desc + (size_t)my_slot * (size_t)(RG * S);
            const double *kn = ca.norm + (size_t)my_slot * S;
            if (wave < H) masked_pair<RG, S, W, PDC, TA, true>(Qs, nq, wrow, kd, kn, first, (1u << TA) - 1u, lane, b, bs);
            else { first += TA; first = first >= S ? first - S : first; masked_pair<RG, S, W, PDC, TB, true>(Qs, nq, wrow, kd, kn, first, (1u << TB) - 1u, lane, b, bs); }
        }
        if (lane == 0) { s_b[wave] = b; s_bs[wave] = bs; }
        CAND_STAMP(3);
        __syncthreads();
        if (wave < H && c < ca.k && lane == 0) {
            double best = kBigDist; int bshift = 0;
            double b0 = s_b[wave]; int bs0 = s_bs[wave];
            const double b1 = s_b[wave + H]; const int bs1 = s_bs[wave + H];
            if (b1 < b0 || (b1 == b0 && bs1 < bs0)) { b0 = b1; bs0 = bs1; }
            if (my_slot >= 0 && b0 < kBigDist) { best = b0; bshift = bs0; }
            o_dist[c] = best; o_shift[c] = bshift;
        }
    } else
    for (int c = wave; c < ca.k; c += kCandWaves) {
        const int slot = c == wave ? my_slot : my_slot2;
        double best = kBigDist; int bshift = 0;
        if (slot >= 0) {                                                         // (wave uniform; a slot the search left unfilled: (1e7, 0) like launch_sc_distance)
            constexpr int L = S >> 1;
            const int ll2 = lane < L ? lane : L - 1;
            const double2 vk = *reinterpret_cast<const double2 *>(ca.vkey + (size_t)slot * S + 2 * ll2);
            const int a0 = align_keyframe_exact<S>(vk, lane, wrow, vq);
            CAND_STAMP(2);
            int first = a0 - SR; first = first < 0 ? first + S : first;          // D.h:1545-1551: the searched shifts start SEARCH_RADIUS below
            wave_fence_lds();
            double b; int bs;
            masked_pair<RG, S, W, PDC, W, true>(Qs, nq, wrow, ca.desc + (size_t)slot * (size_t)(RG * S), ca.norm + (size_t)slot * S, first,
                                               W >= 32 ? 0xffffffffu : ((1u << W) - 1u), lane, b, bs);
            if (b < kBigDist) { best = b; bshift = bs; }
            CAND_STAMP(3);
        }
        if (lane == 0) { o_dist[c] = best; o_shift[c] = bshift; }
    }
    if (ca.seq) {                                                                // the block is complete in host memory before its sequence number
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) *reinterpret_cast<volatile unsigned int *>(ca.out + cand_seq_offset(ca.k)) = ca.seq;
    }
    CAND_STAMP(4);
}

}  // namespace

#ifdef SCL_DIAGNOSTICS
void cand_stamps_print()
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cand_stamps), sizeof h) != hipSuccess || !h[15]) return;
    static const char *names[5] = {"merge of the scan's lists", "scan staged", "alignment (wave 0's candidate)", "13 shifts", "results out"};
    fprintf(stderr, "candidates' kernel stamps (thread 0, %llu launches, us per launch):", h[15]);
    for (int k = 0; k < 5; ++k) fprintf(stderr, " %s %.2f;", names[k], (double)h[k] / (double)h[15] / 100.0);
    fprintf(stderr, "\n");
}
#endif

bool sc_masked_supported(const DbView &db, int SR)
{
    const int W = 2 * SR + 1;
    return (db.RG == 16 && db.S == 120 && W == 13) || (db.RG == 20 && db.S == 180 && W == 19);
}

hipError_t launch_sc_masked(const DbView &db, int SR, const MaskedQuery *queries, int nq, int parts, hipStream_t stream, bool spread)
{
    if (nq < 1 || nq > kMaxMaskedQueries || parts < 1 || !sc_masked_supported(db, SR)) return hipErrorInvalidValue;
    MaskedArgs ma{};
    ma.desc = db.desc; ma.norm = db.norm; ma.nq = nq; ma.parts = parts; ma.spread = spread ? 1 : 0;
    for (int i = 0; i < nq; ++i) ma.q[i] = queries[i];
    if (db.S == 120) return launch_masked<16, 120, 13>(ma, stream);
    return launch_masked<20, 180, 19>(ma, stream);
}


template <int RG, int S, int W, bool EVERY>
static hipError_t launch_small_t(const SmallExactArgs &sa, hipStream_t stream)
{
    using SC = SmallCfg<RG, S, W, EVERY>;
    static_assert(SC::LDS <= 160 * 1024, "LDS");
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<bool> &attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)sc_small_exact_kernel<RG, S, W, EVERY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC::LDS);
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((sc_small_exact_kernel<RG, S, W, EVERY>), dim3(sa.nq), dim3(kSmallWaves * kWave), SC::LDS, stream, sa);
    return hipGetLastError();
}

bool sc_small_exact_supported(const DbView &db, int SR) { return (db.RG == 16 && db.S == 120 && 2 * SR + 1 == 13) || (db.RG == 20 && db.S == 180 && 2 * SR + 1 == 19); }

hipError_t launch_sc_small_exact(const DbView &db, int SR, const SmallExactArgs &args_in, hipStream_t stream)
{
    if (args_in.nq < 1 || args_in.nq > (args_in.q_dev ? kMaxSmallExactQueries : kMaxQueryBatch) || !sc_small_exact_supported(db, SR)) return hipErrorInvalidValue;
    SmallExactArgs sa = args_in;
    sa.desc = db.desc; sa.norm = db.norm; sa.vkey = db.vkey;
    if (db.S == 120) return sa.every_shift ? launch_small_t<16, 120, 13, true>(sa, stream) : launch_small_t<16, 120, 13, false>(sa, stream);
    return launch_small_t<20, 180, 19, false>(sa, stream);
}


bool sc_cand_exact_supported(const DbView &db, int SR)
{
    const int W = 2 * SR + 1;
    return (db.RG == 16 && db.S == 120 && W == 13) || (db.RG == 5 && db.S == 60 && W == 7 && db.R == 20);
}

template <int RG, int S, int W>
static hipError_t launch_cand_t(const CandExactArgs &ca, hipStream_t stream)
{
    using CC = CandCfg<RG, S, W>;
    static_assert(CC::LDS <= 160 * 1024, "LDS");
    static_assert((size_t)kCandWaves * CC::LDS_WAVE >= 2 * (size_t)kCandMergeMaxKeys * 8 + sizeof(TopkMergeShared) + 16 * 8, "the merge of the scan's lists works in the waves' rows");
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<bool> &attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)sc_cand_exact_kernel<RG, S, W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CC::LDS);
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((sc_cand_exact_kernel<RG, S, W>), dim3(1), dim3(kCandWaves * kWave), CC::LDS, stream, ca);
    return hipGetLastError();
}

hipError_t launch_sc_cand_exact(const DbView &db, const QueryView &q, int SR, int k, const int *cand_idx, const float *cand_d2, void *pinned_out, hipStream_t stream, unsigned int seq,
                                const unsigned long long *lists, int n_lists, int *cand_idx_out, float *cand_d2_out)
{
    if (k < 1 || k > 2 * kCandWaves || !sc_cand_exact_supported(db, SR) || !pinned_out) return hipErrorInvalidValue;
    if (lists && (n_lists < 0 || n_lists * k > kCandMergeMaxKeys || !cand_idx_out || !cand_d2_out)) return hipErrorInvalidValue;
    CandExactArgs ca{};
    ca.lists = lists; ca.n_lists = n_lists; ca.cand_idx_out = cand_idx_out; ca.cand_d2_out = cand_d2_out;
    ca.desc = db.desc; ca.norm = db.norm; ca.vkey = db.vkey; ca.q_desc = q.desc; ca.q_norm = q.norm; ca.q_vkey = q.vkey;
    ca.k = k; ca.cand_idx = cand_idx; ca.cand_d2 = cand_d2; ca.out = static_cast<char *>(pinned_out); ca.seq = seq;
    if (db.S == 120) return launch_cand_t<16, 120, 13>(ca, stream);
    return launch_cand_t<5, 60, 7>(ca, stream);
}

}  // namespace scl
