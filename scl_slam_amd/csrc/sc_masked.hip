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

#include "device_common.hpp"
#include "kernels.hpp"

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

template <int RG, int S, int W>
__global__ __launch_bounds__((MaskedCfg<RG, S, W>::WAVES * kWave)) void sc_masked_kernel(MaskedArgs ma)
{
    using C = MaskedCfg<RG, S, W>;
    constexpr int CPL = C::CPL, PITCH = C::PITCH, QCOLS = C::QCOLS, TM = kMaskTMax;
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
    const int qi = jx % ma.nq;
    const int part = (jx / ma.nq) * 8 + xcd;
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

    const bool active = lane < kMaskLanes;
    const int ll = active ? lane : kMaskLanes - 1;
    const double kInf = __longlong_as_double(0x7ff0000000000000LL);
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
        // candidate columns of this lane and their norms
        int yc[CPL];
        double nk[CPL];
        const double *kn = ma.norm + (size_t)slot * S;
#pragma unroll
        for (int j = 0; j < CPL; ++j) { int y = ll + kMaskLanes * j - first; y = y < 0 ? y + S : y; yc[j] = y; nk[j] = kn[y]; }
        const float4 *kd = ma.desc + (size_t)slot * (size_t)(RG * S);
        double best = kInf; int bshift = 0x7fffffff;
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
            constexpr int PD = MASKED_PD;
            float4 kbuf[PD][CPL];
#pragma unroll
            for (int d = 0; d < PD; ++d)
#pragma unroll
                for (int j = 0; j < CPL; ++j) kbuf[d][j] = kd[(size_t)(d < RG ? d : RG - 1) * S + yc[j]];
            static_assert(RG % PD == 0, "ring groups in whole prefetch rounds");
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
    const int grid = 8 * ((ma.parts + 7) / 8) * ma.nq;
    hipLaunchKernelGGL((sc_masked_kernel<RG, S, W>), dim3(grid), dim3(C::WAVES * kWave), C::LDS, stream, ma);
    return hipGetLastError();
}

}  // namespace

bool sc_masked_supported(const DbView &db, int SR)
{
    const int W = 2 * SR + 1;
    return (db.RG == 16 && db.S == 120 && W == 13) || (db.RG == 20 && db.S == 180 && W == 19);
}

hipError_t launch_sc_masked(const DbView &db, int SR, const MaskedQuery *queries, int nq, int parts, hipStream_t stream)
{
    if (nq < 1 || nq > kMaxMaskedQueries || parts < 1 || !sc_masked_supported(db, SR)) return hipErrorInvalidValue;
    MaskedArgs ma{};
    ma.desc = db.desc; ma.norm = db.norm; ma.nq = nq; ma.parts = parts;
    for (int i = 0; i < nq; ++i) ma.q[i] = queries[i];
    if (db.S == 120) return launch_masked<16, 120, 13>(ma, stream);
    return launch_masked<20, 180, 19>(ma, stream);
}

}  // namespace scl
