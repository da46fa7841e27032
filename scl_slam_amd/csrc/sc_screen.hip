// sc_screen.hip -- K0 + K1s: the screening pass of the full-DB mode (64x120 and 80x180 grids).
//
// Full-DB detection needs the ARG-MIN over the database of distanceBtnScanContext (descriptor.h:1538-1569), not
// every distance.  This pass gives every keyframe a guaranteed interval around its reference distance; only the
// keyframes whose interval reaches below the smallest upper bound are then scored by the exact fp64 kernel
// (sc_distance.hip), so the winner -- index, shift and fp64 distance -- is the reference's, bit for bit.
//
// Per keyframe:
//   1. the reference's own alignment, exactly (fastAlignUsingVkey, D.h:1491-1511: matrix-core correlation filter with the
//      exact fp64 evaluation as fallback -- the same code path as sc_distance.hip);
//   2. the 13 shifted cosine distances of D.h:1545-1566 in reduced precision on the matrix cores: both descriptors
//      with unit columns (x / |column|, fp32) rounded to fp16, products exact, fp32 accumulation:
//        sim[t] = sum over query sectors x and rings r of  Qh[r][x + t] * Kh[r][(x - b) mod S],   b = first shift
//      which is ONE matrix product per 16 keyframes: M = 16 shift rows (13 used), N = 16 keyframes, K = 7 680
//      = (ring group, sector, ring) -- v_mfma_f32_16x16x32_f16, 240 per 16 keyframes (first form; the second form turns the
//      product round: one keyframe against 16 scans).  The effective-sector counts (D.h:1523-1526) are popcounts of the
//      rotated sector masks, exact;
//   3. d~ = min_t (1 - sim[t] / n_eff[t]).
// Error of d~ against the reference distance of the same shift: each cosine is off by at most
//   2u + u^2 (u = 2^-11, fp16 rounding of both unit vectors, Cauchy-Schwarz) + 16 * 2^-25 (fp16 subnormals)
//   + 3 * 2^-23 (fp32 scaling by the reciprocal norms) < 9.78e-4,
// the accumulation (first form: chains of 60 MFMAs = 1 920 products per wave, then 4 partials; second form: chains of 30 MFMAs
// = 960 products, then 8 partials) adds at most 1 924 * 2^-23 = 2.3e-4 even if every addition truncated; the mean over the effective sectors keeps that bound.  kScreenEps = 1.5e-3
// leaves 20 % on top.  A keyframe can hold the minimum only if d~ <= min(d~) + 2 * kScreenEps.
// Keyframes (or queries) with a column norm outside [2^-60, 2^60] or non-finite are never screened out.
//
// What is in this file:
//   sc_align_role / sc_align_kernel         K0: the first shift of every (scan, keyframe) pair -- the reference's alignment exactly,
//                                           as a two-stage matrix-core correlation filter with the fp64 evaluation as fallback
//   sc_screen_role / sc_screen_kernel       K1s, FIRST form: one scan against 16 keyframes per matrix-core tile, the keyframes' rows
//                                           fetched per pair (rotated by the pair's first shift) -- bound by what the L2s deliver;
//                                           the next batch's alignment rides in the same grid (SCL_SCREEN_FORM=1; probes)
//   sc_screen2_kernel                       K1s, SECOND form (default on 64 x 120 and 80 x 180): one KEYFRAME against the launch's
//                                           scans per tile, the scans rotated out of LDS, ring parts in separate workgroups
//   sc_screen2_finish_body / _tail_kernel   the ring parts of a pair meet: bound d~, flags, ring-key metric -- in one launch with the
//                                           next batch's alignment
//   sc_select_kernel                        survivors + ring-key top-k (80 x 180's exact pass; scl_screen_distances)
//   launch_sc_screen_batch                  the launch group of a batch
#include <atomic>
#include <type_traits>

#include "align_exact.hpp"
#include "device_common.hpp"
#include "kernels.hpp"

namespace scl {

namespace {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(8))) u32x4_a8;   // 16 B at an 8-byte boundary (hdesc windows start at any sector)

struct ScreenArgs {
    const double *vkey;
    const uint2 *hdesc; const unsigned int *kmask;
    const double *q_vkey; const float *q_rkey;
    const uint2 *q_hdesc; const unsigned int *q_kmask;
    const _Float16 *hkey; const _Float16 *q_hkey; int hkw;   // dense fp16 sector keys (+ norm), hkw halfs per slot
    const float4 *rkey4; int rk_cap;
    int slot_base, n;
    int *starts;              // [n] first shifts: written by sc_align_kernel, read by sc_screen_kernel
    unsigned long long *fallbacks;   // keyframes whose first shift the filter could not decide (statistics; may be null)
    float *out_approx;        // [n] d~ ; -inf = must be scored exactly, +inf = no finite distance
    float *out_d2;            // [n] squared ring-key distance (nanoflann's metric), for the top-k
    unsigned int *out_smask;  // [n] optional: the shifts of the pair that can still hold its minimum (second form's finishing)
    unsigned int *t_min;      // ordered image of min d~ over the screened keyframes (atomicMin; re-armed by the exact pass)
    int align_filter;
};

// The argument block of a batch: what the queries share, and four words per query (a kernel's arguments end at 4 KB; the
// launch takes two batches).  A role rebuilds its query's ScreenArgs from it (wave-uniform: scalar registers).
struct ScreenQuery { int slot, base, n, buf; };
struct ScreenBatchArgs {
    const double *vkey; const float *rkey; const uint2 *hdesc; const unsigned int *kmask; const float4 *rkey4; const unsigned short *hkey;
    int *starts; float *approx; float *ring_d2; unsigned int *t_min; unsigned long long *fallbacks; unsigned int *smask;
    unsigned long long pair_stride;
    int S, R4, hstride, rk_cap, align_filter, hkw;
    int nq, nb;
    int skip_d2;              // the alignment role leaves the ring-key metric to sc_screen2_finish_kernel
    int self_align;           // first form, a batch nobody aligned in advance (a blocking call of one to three scans): every products workgroup aligns its OWN groups first
    ScreenQuery q[kMaxScreenBatch];
};
__device__ __forceinline__ ScreenArgs screen_args_of(const ScreenBatchArgs &ab, int qi)
{
    const ScreenQuery sq = ab.q[qi];
    ScreenArgs a;
    const size_t slot = (size_t)sq.slot, off = (size_t)sq.buf * (size_t)ab.pair_stride;
    a.vkey = ab.vkey; a.hdesc = ab.hdesc; a.kmask = ab.kmask;
    a.q_vkey = ab.vkey + slot * ab.S; a.q_rkey = ab.rkey + slot * ab.R4;
    a.q_hdesc = ab.hdesc + slot * ab.hstride; a.q_kmask = ab.kmask + slot * 8;
    a.hkey = reinterpret_cast<const _Float16 *>(ab.hkey); a.q_hkey = a.hkey + slot * (size_t)ab.hkw; a.hkw = ab.hkw;
    a.rkey4 = ab.rkey4; a.rk_cap = ab.rk_cap;
    a.slot_base = sq.base; a.n = sq.n;
    a.starts = ab.starts + off; a.fallbacks = ab.fallbacks;
    a.out_approx = ab.approx + off; a.out_d2 = ab.ring_d2 + off; a.out_smask = ab.smask ? ab.smask + off : nullptr;
    a.t_min = ab.t_min + sq.buf; a.align_filter = ab.align_filter;
    return a;
}

__device__ __forceinline__ int wrapS(int x, int S)
{   // x in (-S, 2S)
    x = x < 0 ? x + S : x;
    return x >= S ? x - S : x;
}

__device__ __forceinline__ unsigned int float_to_ordered_u(float f)
{
    const unsigned int b = (unsigned int)__float_as_int(f);
    return (b >> 31) ? ~b : (b | 0x80000000u);
}

constexpr float kScreenEps = 1.5e-3f;          // see the error budget at the top of this file
// ... its part that does not come from the fp16 rounding of the operands, for the products' SECOND form: an accumulator's chain of S / 4
// matrix-core products of 32 terms each, every term's addition truncating (|error| <= 2^-23 of the sum of the terms' magnitudes, which
// is at most n_eff (1 + 1e-3): unit columns), the accumulators and ring parts joined by at most 16 further additions; 2e-6 for the fp32
// scaling of the unit columns (3.6e-7), the finishing kernel's fp32 quotient and difference and this bound's own rounding.
// 1.19e-4 at 64 x 120 (chains of 960 terms), 1.76e-4 at 80 x 180 (1 440).
template <int S> constexpr float screen2_acc_eps() { return (float)((S / 4) * 32 + 16) * 1.1920929e-7f * 1.002f + 2.0e-6f; }
constexpr int kScreenWaves = 4;
constexpr int kGroup = 16;                     // keyframes per matrix product (the MFMA's N)
constexpr int kTileStride = 64;                // bytes per keyframe in a wave's transposition tile (32 fp16); the 16-byte chunk j of keyframe n
                                               // sits at chunk j ^ ((n >> 1) & 2): stores and loads both conflict-free
constexpr int kScreenMaxBlocks = 768;          // workgroups per query (3 per CU at most)

// ---------------------------------------------------------------------------------------------------------------------
// K0: the first shift of every (query, keyframe) pair -- fastAlignUsingVkey (D.h:1491-1511) -- and nanoflann's ring-key
// metric, ahead of the screening products.  One wave per 16 keyframes.
//
// The arg-min over shifts s of |vq - shift(vk, s)| is the arg-max of the circular correlation
//   c[s] = sum_u vq[(u + s) mod S] * vk[u],
// a matrix product: M = shifts, N = 16 keyframes, K = sectors.  It is evaluated as a FILTER in two stages on the matrix
// cores, and only what neither stage can decide goes to the reference's own fp64 evaluation (align_keyframe_exact, the code
// of the exact kernel; ties included, so the reference's "lowest shift wins" is kept):
//   stage 1, always: both keys as unit vectors in fp16 (written at ingest behind the screening copy), v_mfma_f32_16x16x32_f16
//     (MT x SK/32 of them: 32 at 64 x 120).  |c~ - c^| <= 9.9e-4 (fp16 rounding of two unit vectors, Cauchy-Schwarz;
//     subnormal elements; fp32 accumulation): a shift leading every other by more than 3e-3 is the exact arg-max.  On the
//     bench database 96 % of the pairs are decided here.
//   stage 2, for a group with three or more open keyframes: the keys in fp32, v_mfma_f32_16x16x4_f32 (256 per group):
//     |c~ - c| <= 4.07e-6 |vq| |vk| for any summation order (K + 2 roundings of 2^-24 relative; truncating accumulation
//     would still fit); a lead of more than 4 eps decides.  Norms that are not finite or so large that products could
//     overflow decide nothing.
// Lane (m, k) = (lane & 15, lane >> 4) owns A[s = 16t + m][...] read from repeated copies of the query's key in LDS; B comes
// from a per-wave LDS image of the 16 sector keys.  Results: starts[i] = (shift - SR) mod S, out_d2[i].
constexpr int hdesc_rgh(int RG) { return ((RG * 8 + 63) / 64) * 8; }   // 8-byte elements per sector of hdesc: whole k-steps of 64 B
constexpr float kAlign16Margin = 3.0e-3f;      // lead the fp16 stage demands of the best shift (normalised correlation; bound below: 2 x 9.9e-4)
constexpr int kAlignFp32From = 3;              // ambiguous keyframes in a group from which the fp32 stage runs before the exact evaluation

// own_groups: the workgroup aligns the groups the PRODUCTS role of the same (query, block) will walk -- bid, bid + nb, ...: wave w takes
// every NWV-th of them -- instead of its share of a launch of its own (bid NWV + wave, + nb NWV, ...): sc_screen_role calls it so.
template <int RG, int S, int W, int NWV = kScreenWaves>
__device__ __forceinline__ void sc_align_role(const ScreenBatchArgs &ab, const int block, unsigned char *smem_raw, const bool own_groups = false)
{
    constexpr int L = S >> 1;
    constexpr int MT = (S + 15) / 16;                  // tiles of 16 shifts
    constexpr int KB = (S + 15) / 16;                  // blocks of 16 sectors (fp32 stage)
    constexpr int QX = 16 * (MT + KB);                 // the query key, repeated
    constexpr int SK = hkey_halfs(S);                  // fp16 stage: K padded to whole steps of 32
    constexpr int KS = SK / 32;                        // ... its k-steps
    constexpr int QH = 16 * MT + SK + 8;               // halfs of the repeated fp16 query key
    static_assert(S % 4 == 0, "sector keys are read in pairs, the B image in fours");
    const int SR = (W - 1) / 2;

    const int nbk = ab.nb;
    const int qi = ab.nq > 1 ? block / nbk : 0;
    const int bid = block - qi * nbk;
    const ScreenArgs a = screen_args_of(ab, qi);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    double *vq = reinterpret_cast<double *>(smem_raw);                           // [S]
    float *qx = reinterpret_cast<float *>(vq + S);                               // [QX]
    _Float16 *qh0 = reinterpret_cast<_Float16 *>(qx + QX);                       // [QH] the unit fp16 query key, repeated; qh1[i] = qh0[i + 1]
    _Float16 *qh1 = qh0 + QH;
    unsigned char *wbase = reinterpret_cast<unsigned char *>(qh1 + QH) + (size_t)wave * ((2 * S + 2) * 8);
    double *vk2 = reinterpret_cast<double *>(wbase);                             // [2S + 2] scratch of the exact evaluation

    const int m16 = lane & 15, k4 = lane >> 4;
    // (the first group's keys and the query's norm are requested before the set-up's own reads: one memory round trip for all)
    const float qnorm = *reinterpret_cast<const float *>(a.q_hkey + SK);
    const int ngroups = (a.n + kGroup - 1) / kGroup;
    // The keys of a group (16 keyframes x SK halfs, their norms) are fetched into registers one group ahead: a wave's groups are a
    // chain of short phases, and the fetch was a whole memory round trip at the head of every one of them.
    static_assert(kGroup * (SK / 8) == KS * kWave, "one 16-byte chunk per lane and k-step");
    // ... in the matrix cores' B layout: lane (n, j) = (lane & 15, lane >> 4) holds halfs 32 kk + 8 j .. + 7 of keyframe n's key for
    // k-step kk -- the fragments go from memory to the MFMAs without a trip through LDS
    uint4 pre[KS];
    float knorm_pre = 0.f;
    auto fetch = [&](int g) {
        const int fs = a.slot_base + g * kGroup, lr = a.n - 1 - g * kGroup;
        const unsigned char *kp = reinterpret_cast<const unsigned char *>(a.hkey + (size_t)(fs + (m16 < lr ? m16 : lr)) * (size_t)a.hkw);   // the dense key table
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) pre[kk] = *reinterpret_cast<const uint4 *>(kp + (4 * kk + k4) * 16);
        knorm_pre = *reinterpret_cast<const float *>(reinterpret_cast<const _Float16 *>(kp) + SK);
    };
    const int g_first = own_groups ? bid + wave * nbk : bid * NWV + wave;
    if (g_first < ngroups) fetch(g_first);
    {   // every global read of the set-up first, then the LDS stores: three loops of load -> store were three memory round trips
        constexpr int T = NWV * kWave;
        constexpr int N1 = (S + T - 1) / T, N2 = (QX + T - 1) / T, N3 = (QH + T - 1) / T;
        const _Float16 *qk = a.q_hkey;
        double r1[N1]; double r2[N2]; _Float16 r3a[N3], r3b[N3];
#pragma unroll
        for (int u = 0; u < N1; ++u) { const int c = threadIdx.x + u * T; r1[u] = c < S ? a.q_vkey[c] : 0.0; }
#pragma unroll
        for (int u = 0; u < N2; ++u) { const int i = threadIdx.x + u * T; r2[u] = i < QX ? a.q_vkey[i % S] : 0.0; }
#pragma unroll
        for (int u = 0; u < N3; ++u) { const int i = threadIdx.x + u * T; r3a[u] = i < QH ? qk[i % S] : (_Float16)0.f; r3b[u] = i < QH ? qk[(i + 1) % S] : (_Float16)0.f; }
#pragma unroll
        for (int u = 0; u < N1; ++u) { const int c = threadIdx.x + u * T; if (c < S) vq[c] = r1[u]; }
#pragma unroll
        for (int u = 0; u < N2; ++u) { const int i = threadIdx.x + u * T; if (i < QX) qx[i] = (float)r2[u]; }
#pragma unroll
        for (int u = 0; u < N3; ++u) { const int i = threadIdx.x + u * T; if (i < QH) { qh0[i] = r3a[u]; qh1[i] = r3b[u]; } }
    }
    __syncthreads();

    const float *qa = qx + 4 * k4 + m16;                                         // A[s = 16t + m][u = 16b + 4k + e] = qa[16 (b + t) + e]
    // fp16 stage: A[s = 16t + m][u = 32kk + 8k + i] = q^[(32kk + 16t + (8k + m) + i) mod S]: eight consecutive halfs from an
    // arbitrary index -- from the copy shifted by one when that index is odd, so that the reads stay dword aligned
    const int lc = 8 * k4 + m16;
    const unsigned int *qha = reinterpret_cast<const unsigned int *>(((lc & 1) ? qh1 : qh0) + (lc & ~1));
    float qn2 = 0.f;
    for (int i = lane; i < S; i += kWave) qn2 += qx[i] * qx[i];
    qn2 = wave_sum_f32_dpp(qn2);
    const bool use_filter = a.align_filter != 0;
    const float kNegInf = __int_as_float(0xff800000);

    // per keyframe (column n = lane & 15): the largest and the second largest value over all shifts, and the largest's shift
    auto top2 = [&](const f4v (&acc)[MT], float &v1, float &v2, int &a1) {
        v1 = kNegInf; v2 = kNegInf; a1 = 0;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sft = 16 * t + 4 * k4 + i;
                const float c = sft < S ? acc[t][i] : kNegInf;
                const bool gt = c > v1;
                v2 = gt ? v1 : fmaxf(v2, c);
                a1 = gt ? sft : a1;
                v1 = gt ? c : v1;
            }
        }
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float o1 = __shfl_xor(v1, off, kWave), o2 = __shfl_xor(v2, off, kWave);
            const int oa = __shfl_xor(a1, off, kWave);
            const bool gt = o1 > v1;
            v2 = fmaxf(gt ? v1 : o1, fmaxf(v2, o2));
            a1 = gt ? oa : a1;
            v1 = gt ? o1 : v1;
        }
    };

    for (int g = g_first; g < ngroups; g += nbk * NWV) {
        const int c_base = g * kGroup;
        const int first_slot = a.slot_base + c_base;
        const int last_rel = a.n - 1 - c_base;
        const bool mine = lane < kGroup && c_base + lane < a.n;
        // ---- stage 1: the normalised correlation in fp16 on the matrix cores (v_mfma_f32_16x16x32_f16, MT x KS of them).
        // Unit vectors rounded to fp16 are off by <= 2^-11 of themselves (+ 2^-25 per subnormal element), their products are
        // exact, the fp32 accumulation of SK terms adds <= SK 2^-24: |c~ - c^| <= 9.9e-4 for every shift, so a shift that leads
        // every other by more than kAlign16Margin = 3e-3 is the arg-max of the exact correlation, i.e. the reference's arg-min.
        h8 bcur[KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) bcur[kk] = __builtin_bit_cast(h8, pre[kk]);
        const float knorm = knorm_pre;
        if (g + nbk * NWV < ngroups) fetch(g + nbk * NWV);                       // in flight under this group's products and decisions
#ifdef SCL_DIAGNOSTICS
        if (a.align_filter == 2) continue;                                       // probe: the role's key reads alone
#endif
        float v1, v2;
        int a1;
        {
            f4v acc[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = f4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const h8 bfrag = bcur[kk];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const unsigned int *ap = qha + (32 * kk + 16 * t) / 2;
                    u32x4 aw;
                    aw[0] = ap[0]; aw[1] = ap[1]; aw[2] = ap[2]; aw[3] = ap[3];
                    h8 afrag;
                    __builtin_memcpy(&afrag, &aw, 16);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, bfrag, acc[t], 0, 0, 0);
                }
            }
            // The largest and the second largest of the S correlation values of a keyframe (the filter asks whether they are more
            // than the margin apart) and the largest's shift, at 2.5 instructions per value: the shift is written into the low 8
            // mantissa bits of its value (values are normalised correlations in [-1, 1]: a perturbation of < 1.6e-5, which the
            // margin test below allows for), so the maximum carries its own arg-max; pairs of values enter a running (hi, lo)
            // through max3 / med3 (the second largest of {hi, lo, x, y} is max(lo, med3(hi, x, y))), two independent chains.
            static_assert(S <= 256, "8 bits of shift");
            auto tag = [&](float c, int sft) { return sft < S ? __uint_as_float((__float_as_uint(c) & ~255u) | (unsigned int)sft) : kNegInf; };
            float hi[2] = {kNegInf, kNegInf}, lo[2] = {kNegInf, kNegInf};
#pragma unroll
            for (int t = 0; t < MT; ++t) {
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const float x = tag(acc[t][i], 16 * t + 4 * k4 + i), y = tag(acc[t][i + 1], 16 * t + 4 * k4 + i + 1);
                    const int ch = (t & 1);
                    lo[ch] = fmaxf(lo[ch], __builtin_amdgcn_fmed3f(hi[ch], x, y));
                    hi[ch] = fmaxf(fmaxf(hi[ch], x), y);
                }
            }
            auto merge = [](float &h, float &l, float oh, float ol) { l = fmaxf(fmaxf(l, ol), fminf(h, oh)); h = fmaxf(h, oh); };
            merge(hi[0], lo[0], hi[1], lo[1]);
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) merge(hi[0], lo[0], __shfl_xor(hi[0], off, kWave), __shfl_xor(lo[0], off, kWave));
            v1 = hi[0]; v2 = lo[0] + 4.0e-5f;                                   // (the tags moved either value by < 1.6e-5)
            a1 = (int)(__float_as_uint(hi[0]) & 255u);
        }
        // The exact arg-max of the correlation is the reference's arg-min only while its fp64 distances resolve the lead: their
        // rounding noise is ~1e-13 (|vq|^2 + |vk|^2), a lead of 3e-3 |vq| |vk| stands clear of it for norm ratios up to 1e4;
        // distances of 1e7 and more never win in the reference (D.h:1494), so the norms stay below 4e6; tiny norms stay
        // above 1e-30 (no fp64 underflow).  Anything else is left to the exact evaluation.
        const bool in_range = qnorm >= 1e-30f && qnorm <= 4.0e6f && knorm >= 1e-30f && knorm <= 4.0e6f &&
                              qnorm <= 1.0e4f * knorm && knorm <= 1.0e4f * qnorm;
        bool uniq = use_filter && in_range && (v1 == v1) && (v2 < v1 - kAlign16Margin);
        unsigned long long amb = __builtin_amdgcn_ballot_w64(mine && !uniq);
        // ---- stage 2 (a group with several keyframes the first stage left open): the correlation in fp32 ----
        if (use_filter && __popcll(amb) >= kAlignFp32From) {
            // (rare: the keys come straight from the fp64 table, four sectors per lane and block, zero past the last sector)
            const double *src = a.vkey + (size_t)(first_slot + (m16 < last_rel ? m16 : last_rel)) * S;
            f4v acc[MT];
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = f4v{0.f, 0.f, 0.f, 0.f};
            float kn2 = 0.f;
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const int u0 = 16 * b + 4 * k4;
                f4v b4 = {0.f, 0.f, 0.f, 0.f};
                if (u0 < S) {                                                    // S % 4 == 0: a block of four is whole or absent
                    const double2 p0 = *reinterpret_cast<const double2 *>(src + u0), p1 = *reinterpret_cast<const double2 *>(src + u0 + 2);
                    b4 = f4v{(float)p0.x, (float)p0.y, (float)p1.x, (float)p1.y};
                }
                kn2 += (b4[0] * b4[0] + b4[1] * b4[1]) + (b4[2] * b4[2] + b4[3] * b4[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[16 * (b + t) + e], b4[e], acc[t], 0, 0, 0);
                }
            }
            kn2 += __shfl_xor(kn2, 16, kWave);
            kn2 += __shfl_xor(kn2, 32, kWave);
            float w1, w2;
            int b1;
            top2(acc, w1, w2, b1);
            const float nsum = sqrtf(qn2) + sqrtf(kn2);
            const float eps = 4.07e-6f * sqrtf(qn2) * sqrtf(kn2) + 1e-12f * (qn2 + kn2);
            const bool sane = (qn2 < 3.0e38f) && (kn2 < 3.0e38f) && (nsum * nsum < 0.9e14f);
            if (!uniq && sane && (w2 < w1 - 4.0f * eps)) { uniq = true; a1 = b1; }
            amb = __builtin_amdgcn_ballot_w64(mine && !uniq);
        }
        // ---- decide (every lane of column n holds the same numbers; lanes 0..15 write) ----
        if (mine && uniq) a.starts[c_base + lane] = wrapS(a1 - SR, S);
#ifdef SCL_DIAGNOSTICS
        if (a.align_filter == 3) { amb = 0; if (mine) a.starts[c_base + lane] = 0; }
#endif
        if (amb && lane == 0 && a.fallbacks) atomicAdd(a.fallbacks, (unsigned long long)__popcll(amb));
        while (amb) {                                                            // the reference's own evaluation, one keyframe at a time
            const int n = __ffsll((long long)amb) - 1;
            amb &= amb - 1;
            int al;
            if constexpr (S / 2 <= kWave) {
                const int ll = lane < L ? lane : L - 1;
                const double2 vk = *reinterpret_cast<const double2 *>(a.vkey + (size_t)(first_slot + n) * S + 2 * ll);
                al = align_keyframe_exact<S>(vk, lane, vk2, vq);
            } else {                                                             // more than two sectors per lane
                constexpr int SPL = (S + kWave - 1) / kWave, LA = S / SPL;
                const int ll = lane < LA ? lane : LA - 1;
                double vk[SPL];
#pragma unroll
                for (int u = 0; u < SPL; ++u) vk[u] = a.vkey[(size_t)(first_slot + n) * S + SPL * ll + u];
                al = align_keyframe_wide<S>(vk, lane, vk2, vq);
            }
            if (lane == 0) a.starts[c_base + n] = wrapS(al - SR, S);
        }
        // ---- nanoflann's metric (nanoflann.hpp:383-408): four dimensions per step, fp32, groups accumulated in order ----
        if (mine && !ab.skip_d2) {
            const int slot = first_slot + lane;
            float result = 0.0f;
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const float4 bk = a.rkey4[(size_t)r * a.rk_cap + slot];
                const float4 qk = *reinterpret_cast<const float4 *>(a.q_rkey + 4 * r);   // uniform: scalar loads
                const float d0 = qk.x - bk.x, d1 = qk.y - bk.y, d2 = qk.z - bk.z, d3 = qk.w - bk.w;
                const float grp = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
                result += grp;
            }
            a.out_d2[c_base + lane] = result;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// K0, second form: one KEYFRAME against the launch's scans per matrix-core tile (as the products' second form does).
//   c_q[s] = sum_v q_q[v] k[(v - s) mod S]:   A[row][v] = k[(v - s_row) mod S]  (the keyframe),   B[v][q] = q_q[v]  (the scans)
// so B -- the unit fp16 keys of the launch's scans, zero padded to whole k-steps -- is loaded ONCE per wave into registers and
// stays there, and the keyframe's side is the only per-keyframe traffic.  A is a Toeplitz image of the key: lane (m, j) needs the
// eight consecutive entries from (32 kk + 8 j - s_m) mod S.  With the shifts of a tile taken P apart -- row sigma of tile
// (rho, tau) is shift P (16 tau + sigma) + rho, P | S -- that start is rho past a multiple of P for EVERY lane of the tile: the
// fragment is one aligned chunk read of copy rho of the keyframe's alignment image (kernels.hpp: halign_*, written at ingest:
// X_rho[i] = k[(i - rho) mod S]).  P = 8 at S = 120 (15 rows of 16 used, 8 tiles x 4 k-steps = 32 MFMAs per keyframe and launch,
// ds_read_b128), P = 4 at S = 180 (45 rows = 3 x 16, 12 tiles x 6 k-steps, 8-byte aligned reads).
// One wave per keyframe: the image goes from memory (16-byte loads, one keyframe ahead) into the wave's own LDS tile, the
// fragments come out of it.  The filter has the two stages of sc_align_role, both in fp16 products:
//   stage 1, always: kh . qh, the unit keys rounded to fp16.  |c~ - c| <= 9.9e-4 (see sc_align_role); a lead of kAlign16Margin decides.
//   stage 2, when a scan of the launch is still open for this keyframe (half of the keyframes of the bench database; 4 % of its
//     pairs): the keys split in two fp16 parts, k = kh + 2^-11 kl + rk with kl = fp16(2^11 (k - kh)) (scaled: a normal number, no
//     subnormal is ever needed), |rk_v| <= 2^-22 |k_v|.  c~ = sum kh qh + 2^-11 (sum kh ql + kl qh): products of fp16 values are
//     exact in fp32; the leading sum is taken as KS single MFMAs (32 products each, on a zero accumulator) joined by KS - 1
//     additions, the cross sum as one chain whose error is scaled by 2^-11.  Against the correlation of the fp64 unit vectors:
//       accumulation (32 + KS + 1) 2^-23 (every addition truncating)      <= 4.4e-6
//       cross chain 2^-11 (64 KS + 2) 2^-23 x 2,  last fma 2^-24           <= 1.0e-7
//       dropped 2^-22 sum kl ql,  |rq| + |rk|                              <= 7.2e-7
//     kAlignSplitEps = 6e-6 bounds their sum; a shift that leads every other by more than 2 kAlignSplitEps is the arg-max of the
//     exact correlation, i.e. the reference's arg-min (same range conditions on the norms as stage 1).  Values are compared
//     untagged here: the tag's 1.5e-5 would be most of the margin.
//   what neither stage decides (near ties: 5e-4 of the pairs) goes to the reference's own fp64 evaluation, align_keyframe_exact /
//   _wide, one pair at a time -- the code of the exact kernel, ties included ("the lowest shift wins").
// Per 16 scans and keyframe ~110 VALU instructions in stage 1 against 340 per (scan, 16 keyframes) of the first form, no per-scan
// set-up in LDS, and the keyframe's bytes are fetched once per launch instead of once per scan.
constexpr int kAlignUndecided = -1;            // first shift of a pair no alignment has decided (never written by this file's kernels now; the consumers honour it)
constexpr float kAlignSplitEps = 6.0e-6f;      // stage 2: bound of |c~ - c| (see above)
#ifndef SCL_A2_OCC_WIDE
#define SCL_A2_OCC_WIDE 2            // (80 x 180: +2-5 % on the stream against the build for three waves, which spilled 56 bytes)
#endif
#ifndef SCL_A2_OCC
#define SCL_A2_OCC 2                 // (built for two waves per SIMD the tail launch takes 166 registers and spills nothing: three waves per SIMD in fact, and
                                     //  2 us less per launch than the 128-register build for four, which spilled 40 bytes: +3-4 % on the stream)
#endif
#ifndef SCL_A2_PROBE
#define SCL_A2_PROBE 0                         // experiments only (scripts/build_variant.sh): 1 no stage 2 / exact (open pairs marked undecided)
#endif

template <int S>
struct Align2Cfg {
    static constexpr int P = halign_P(S), CP = halign_CP(S);
    static constexpr int ROWS = P ? S / P : 0;                     // shifts per phase
    static constexpr int NTAU = (ROWS + 15) / 16;
    static constexpr int NT = P * NTAU;                            // tiles per keyframe
    static constexpr int SK = hkey_halfs(S), KS = SK / 32;
    static constexpr int TSTEP = 16 * P / 32;                      // a tile row block further on = this many k-steps back
    static constexpr int ND = KS + TSTEP * (NTAU - 1);             // distinct fragment starts of a lane
    static constexpr int IMG = halign_img_bytes(S);                // bytes of one part (hi or lo) of the image
    static constexpr int NLD = (IMG + kWave * 16 - 1) / (kWave * 16);   // 16-byte loads per lane and part
    static constexpr int LDS_WAVE = 2 * IMG > (2 * S + 2) * 8 ? 2 * IMG : ((2 * S + 2) * 8 + 15) / 16 * 16;   // both parts; the exact evaluation's scratch reuses it
    static constexpr int OCC = KS <= 4 ? SCL_A2_OCC : SCL_A2_OCC_WIDE;   // waves per SIMD the kernels are built for
};

template <int S, int W>
__device__ __forceinline__ void sc_align2_role(const ScreenBatchArgs &ab, const unsigned char *halign, int u_lo, int u_n,
                                               const int wave_global, const int waves_total, unsigned char *smem_wave)
{
    using C = Align2Cfg<S>;
    constexpr int P = C::P, CP = C::CP, KS = C::KS, NTAU = C::NTAU, NT = C::NT, ND = C::ND, IMG = C::IMG, SK = C::SK, NLD = C::NLD;
    static_assert(P == 4 || P == 8, "alignment image");
    static_assert(S % P == 0 && (16 * P) % 32 == 0 && CP >= S + 8 && IMG % 16 == 0 && S <= 256 && NLD >= 1 && NLD <= 2, "tiling of the shifts");
    constexpr int SR = (W - 1) / 2;
    constexpr int HAB = halign_bytes(S);
    const int lane = threadIdx.x & (kWave - 1);
    const int c16 = lane & 15, j4 = lane >> 4;                     // A: row sigma = c16; B / output: scan q = c16
    const int cq = c16 < ab.nq ? c16 : 0;                          // columns past the launch's scans shadow scan 0 (never stored)
    const ScreenQuery sq = ab.q[cq];
    const bool q_live = c16 < ab.nq;
    int *starts_q = ab.starts + (size_t)sq.buf * (size_t)ab.pair_stride;
    // ---- B: this lane's part of scan q's key for every k-step (halfs 32 kk + 8 j .. + 7), both parts, and the scan's norm ----
    const _Float16 *qk = reinterpret_cast<const _Float16 *>(ab.hkey) + (size_t)sq.slot * (size_t)ab.hkw;
    h8 bq[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) bq[kk] = *reinterpret_cast<const h8 *>(qk + 32 * kk + 8 * j4);
    const float qnorm = *reinterpret_cast<const float *>(qk + SK);
    const float qerr = *reinterpret_cast<const float *>(qk + SK + 2);          // |q - qh| of the scan's unit key, rounded up (make_sc.hip)
    // ---- A: byte offsets of this lane's fragment starts inside a copy: (8 j - P sigma + 32 d) mod S, d = -TSTEP (NTAU - 1) .. KS - 1
    unsigned int offs[ND];
    {
        int base0 = (8 * j4 - P * c16) % S; base0 = base0 < 0 ? base0 + S : base0;
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            int x = (base0 + 32 * (d - C::TSTEP * (NTAU - 1))) % S; x = x < 0 ? x + S : x;
            offs[d] = (unsigned int)(2 * x);
        }
    }
    const bool use_filter = ab.align_filter != 0;
    const float kNegInf = __int_as_float(0xff800000);
    unsigned long long open_pairs = 0;
    auto frag = [&](const unsigned char *ap) -> h8 {
        if constexpr (P == 8) return *reinterpret_cast<const h8 *>(ap);
        else {
            const uint2 a0 = *reinterpret_cast<const uint2 *>(ap), a1 = *reinterpret_cast<const uint2 *>(ap + 8);
            const u32x4 aw = {a0.x, a0.y, a1.x, a1.y};
            return __builtin_bit_cast(h8, aw);
        }
    };
    // the image's first part (kh) of the wave's next keyframe, requested one keyframe ahead (plain named registers: an array
    // behind a lambda went to scratch memory, and the prefetch with it)
    const unsigned int loff0 = (unsigned int)lane * 16u, loff1 = (unsigned int)(kWave + lane) * 16u;
    const bool ld1 = NLD > 1 && loff1 < (unsigned int)IMG;
    uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = make_uint4(0, 0, 0, 0);
    float knorm_pre = 0.f, kerr_pre = 0.f;
    auto image_of = [&](int idx) { idx = idx < u_n ? idx : u_n - 1; return halign + (size_t)(u_lo + idx) * (size_t)HAB; };
    if (wave_global < u_n) {
        const unsigned char *g = image_of(wave_global);
        pre0 = *reinterpret_cast<const uint4 *>(g + 16 + loff0);
        if (NLD > 1) pre1 = *reinterpret_cast<const uint4 *>(g + 16 + (ld1 ? loff1 : 0u));
        knorm_pre = *reinterpret_cast<const float *>(g);
        kerr_pre = *reinterpret_cast<const float *>(g + 4);
    }
    for (int idx = wave_global; idx < u_n; idx += waves_total) {
        wave_fence();                                                            // the previous keyframe's fragment reads are done
        *reinterpret_cast<uint4 *>(smem_wave + loff0) = pre0;
        if (ld1) *reinterpret_cast<uint4 *>(smem_wave + loff1) = pre1;
        const float knorm = knorm_pre, kerr = kerr_pre;
        wave_fence();
        const unsigned char *g_cur = image_of(idx);
        {
            const unsigned char *g = image_of(idx + waves_total);
            pre0 = *reinterpret_cast<const uint4 *>(g + 16 + loff0);
            if (NLD > 1) pre1 = *reinterpret_cast<const uint4 *>(g + 16 + (ld1 ? loff1 : 0u));
            knorm_pre = *reinterpret_cast<const float *>(g);
            kerr_pre = *reinterpret_cast<const float *>(g + 4);
        }
        // ---- stage 1: the correlation tile by tile; the two largest values per scan with the largest's shift ----
        // (the shift rides in the low 8 mantissa bits of its value: a perturbation of < 3.1e-5, which the margin test allows for;
        //  pairs of values enter a running (hi, lo) through max3 / med3, two independent chains; the next tile's fragments are
        //  requested before this tile's products)
        float hi[2] = {kNegInf, kNegInf}, lo[2] = {kNegInf, kNegInf};
        h8 af[2][KS];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) af[0][kk] = frag(smem_wave + offs[kk + C::TSTEP * (NTAU - 1)]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int rho = t / NTAU, tau = t - rho * NTAU;
            if (t + 1 < NT) {
                const int rn = (t + 1) / NTAU, tn = (t + 1) - rn * NTAU;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) af[(t + 1) & 1][kk] = frag(smem_wave + rn * (CP * 2) + offs[kk + C::TSTEP * (NTAU - 1 - tn)]);
            }
            f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[t & 1][kk], bq[kk], acc, 0, 0, 0);
            const int ch = t & 1;
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                // lane (q, j) holds rows 4 j + i of the tile: shift P (16 tau + 4 j + i) + rho
                const int sg0 = 16 * tau + 4 * j4 + i, sg1 = sg0 + 1;
                const float x = sg0 < C::ROWS ? __uint_as_float((__float_as_uint(acc[i]) & ~255u) | (unsigned int)(P * sg0 + rho)) : kNegInf;
                const float y = sg1 < C::ROWS ? __uint_as_float((__float_as_uint(acc[i + 1]) & ~255u) | (unsigned int)(P * sg1 + rho)) : kNegInf;
                lo[ch] = fmaxf(lo[ch], __builtin_amdgcn_fmed3f(hi[ch], x, y));
                hi[ch] = fmaxf(fmaxf(hi[ch], x), y);
            }
        }
        auto merge = [](float &h, float &l, float oh, float ol) { l = fmaxf(fmaxf(l, ol), fminf(h, oh)); h = fmaxf(h, oh); };
        merge(hi[0], lo[0], hi[1], lo[1]);
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) merge(hi[0], lo[0], __shfl_xor(hi[0], off, kWave), __shfl_xor(lo[0], off, kWave));
        const float v1 = hi[0], v2 = lo[0] + 7.0e-5f;                           // (the tags moved either value by < 3.1e-5)
        int a1 = (int)(__float_as_uint(hi[0]) & 255u);
        // the same range conditions as sc_align_role's first stage (see there)
        const bool in_range = qnorm >= 1e-30f && qnorm <= 4.0e6f && knorm >= 1e-30f && knorm <= 4.0e6f &&
                              qnorm <= 1.0e4f * knorm && knorm <= 1.0e4f * qnorm;
        // the lead the best shift must have: twice the error bound of a value.  |c~ - c| <= |q - qh| |kh| + |q| |k - kh| + the fp32
        // accumulation of 128 exact products ((n - 1) 2^-24 sum |q k| <= 7.6e-6), with the keys' ACTUAL rounding-error norms as recorded
        // at ingest (unit keys: |q| = 1, |kh| <= 1 + |k - kh|) -- never more than the worst case the constant stands for
        const float lead = (qerr > 0.f && kerr > 0.f) ? fminf(kAlign16Margin, 2.0f * (qerr + kerr + qerr * kerr) * 1.0001f + 1.6e-5f) : kAlign16Margin;
        bool uniq = use_filter && in_range && (v1 == v1) && (v2 < v1 - lead);
        const int ci = u_lo + idx - sq.base;
        const bool mine = q_live && ci >= 0 && ci < sq.n;                        // (every lane of column q holds the same numbers)
        unsigned long long amb = __builtin_amdgcn_ballot_w64(mine && !uniq) & 0xffffull;
        // ---- stage 2 (some scan of the launch is still open for this keyframe): the split keys ----
        if (SCL_A2_PROBE != 1 && use_filter && amb) {
            // the image's second part (kl) into the second half of the wave's tile
            const uint4 l0 = *reinterpret_cast<const uint4 *>(g_cur + 16 + IMG + loff0);
            uint4 l1 = make_uint4(0, 0, 0, 0);
            if (NLD > 1) l1 = *reinterpret_cast<const uint4 *>(g_cur + 16 + IMG + (ld1 ? loff1 : 0u));
            *reinterpret_cast<uint4 *>(smem_wave + IMG + loff0) = l0;
            if (ld1) *reinterpret_cast<uint4 *>(smem_wave + IMG + loff1) = l1;
            h8 bql[KS];                                                          // the scans' second parts: needed here only (4 % of the keyframes)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) bql[kk] = *reinterpret_cast<const h8 *>(qk + SK + 8 + 32 * kk + 8 * j4);
            wave_fence();
            float h2 = kNegInf, l2 = kNegInf;
            int g2 = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int rho = t / NTAU, tau = t - rho * NTAU;
                const f4v zero = {0.f, 0.f, 0.f, 0.f};
                f4v hh = zero, xx = zero;
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    const unsigned char *ap = smem_wave + rho * (CP * 2) + offs[kk + C::TSTEP * (NTAU - 1 - tau)];
                    const h8 ah = frag(ap), al = frag(ap + IMG);
                    const f4v one = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bq[kk], zero, 0, 0, 0);   // 32 products on a zero accumulator
                    hh = kk == 0 ? one : hh + one;
                    xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bql[kk], xx, 0, 0, 0);
                    xx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bq[kk], xx, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int sg = 16 * tau + 4 * j4 + i;
                    const float c = sg < C::ROWS ? __builtin_fmaf(xx[i], 0x1p-11f, hh[i]) : kNegInf;     // (2^-11 xx is exact: fma == mul + add)
                    l2 = fmaxf(l2, fminf(c, h2));
                    g2 = c > h2 ? P * sg + rho : g2;
                    h2 = fmaxf(h2, c);
                }
            }
#pragma unroll
            for (int off = 16; off <= 32; off <<= 1) {
                const float oh = __shfl_xor(h2, off, kWave), ol = __shfl_xor(l2, off, kWave);
                const int og = __shfl_xor(g2, off, kWave);
                l2 = fmaxf(fmaxf(l2, ol), fminf(h2, oh));
                g2 = oh > h2 ? og : g2;
                h2 = fmaxf(h2, oh);
            }
            if (!uniq && in_range && (h2 == h2) && (l2 < h2 - 2.0f * kAlignSplitEps)) { uniq = true; a1 = g2; }
            amb = __builtin_amdgcn_ballot_w64(mine && !uniq) & 0xffffull;
        }
        if (mine && j4 == 0 && uniq) starts_q[ci] = wrapS(a1 - SR, S);
        if (SCL_A2_PROBE == 1 && mine && j4 == 0 && !uniq) starts_q[ci] = kAlignUndecided;
        open_pairs += (unsigned long long)__popcll(amb);
        // ---- what the filter left open: the reference's own evaluation, one pair at a time ----
        while (SCL_A2_PROBE != 1 && amb) {
            const int qn = __ffsll((long long)amb) - 1;
            amb &= amb - 1;
            const ScreenQuery oq = ab.q[qn];                                      // (uniform index: scalar loads)
            const double *vq = ab.vkey + (size_t)oq.slot * S;
            const double *vkp = ab.vkey + (size_t)(u_lo + idx) * S;
            double *vk2 = reinterpret_cast<double *>(smem_wave);                  // (the image is not needed any more)
            int al;
            if constexpr (S / 2 <= kWave) {
                constexpr int L = S >> 1;
                const int ll = lane < L ? lane : L - 1;
                const double2 vk = *reinterpret_cast<const double2 *>(vkp + 2 * ll);
                al = align_keyframe_exact<S>(vk, lane, vk2, vq);
            } else {
                constexpr int SPL = (S + kWave - 1) / kWave, LA = S / SPL;
                const int ll = lane < LA ? lane : LA - 1;
                double vk[SPL];
#pragma unroll
                for (int u = 0; u < SPL; ++u) vk[u] = vkp[SPL * ll + u];
                al = align_keyframe_wide<S>(vk, lane, vk2, vq);
            }
            if (lane == 0) (ab.starts + (size_t)oq.buf * (size_t)ab.pair_stride)[u_lo + idx - oq.base] = wrapS(al - SR, S);
        }
    }
    if (lane == 0 && open_pairs && ab.fallbacks) atomicAdd(ab.fallbacks, open_pairs);
}

template <int RG, int S, int W>
__global__ __launch_bounds__(kScreenWaves * kWave, Align2Cfg<S>::OCC) void sc_align2_kernel(ScreenBatchArgs ab, const unsigned char *halign, int u_lo, int u_n)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_align2[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    sc_align2_role<S, W>(ab, halign, u_lo, u_n, (int)blockIdx.x * kScreenWaves + wave, (int)gridDim.x * kScreenWaves,
                         smem_align2 + (size_t)wave * Align2Cfg<S>::LDS_WAVE);
}

// ---------------------------------------------------------------------------------------------------------------------
// K1s: the screening products.
// RGH = 8-byte elements per sector of hdesc (ring groups padded to whole k-steps: 16 at R = 64, 24 at R = 80);
// D = k-steps of loads in flight per wave (1 KB each).
// PROBE (diagnostic builds only, -DSCL_DIAGNOSTICS + SCL_SCREEN_PROBE): 2 = no staging / MFMA (the loads are summed): what
// the access pattern alone costs.  Results are wrong on purpose.

// the S-bit sector mask m (64-bit words, bits >= S zero) rotated right by s: bit x of the result = bit (x + s) mod S of m
template <int S>
__device__ __forceinline__ void rotate_mask(const unsigned long long (&m)[(S + 63) / 64], int s, unsigned long long (&out)[(S + 63) / 64])
{
    constexpr int NW = (S + 63) / 64;
    auto shr = [&](int k, unsigned long long (&o)[NW]) {       // logical shift right by k in [0, 64 NW)
        const int w = k >> 6, b = k & 63;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const unsigned long long lo = i + w < NW ? m[i + w] : 0ull, hi = i + w + 1 < NW ? m[i + w + 1] : 0ull;
            o[i] = b ? (lo >> b) | (hi << (64 - b)) : lo;
        }
    };
    auto shl = [&](int k, unsigned long long (&o)[NW]) {       // logical shift left by k in [0, 64 NW]
        const int w = k >> 6, b = k & 63;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const unsigned long long hi = i - w >= 0 && i - w < NW ? m[i - w] : 0ull, lo = i - w - 1 >= 0 && i - w - 1 < NW ? m[i - w - 1] : 0ull;
            o[i] = b ? (hi << b) | (lo >> (64 - b)) : hi;
        }
    };
    unsigned long long r1[NW], r2[NW];
    shr(s, r1);
    shl(S - s, r2);
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        unsigned long long v = r1[i] | r2[i];
        const int top = S - 64 * i;                             // valid bits of word i
        if (top < 64) v &= top > 0 ? ((1ull << (top & 63)) - 1ull) : 0ull;
        out[i] = v;
    }
}

template <int RGH, int S, int W, int D, int PROBE>
__device__ __forceinline__ void sc_screen_role(const ScreenBatchArgs &ab, const int qi, const int bid, unsigned char *smem_raw)
{
    constexpr int NWV = kScreenWaves;
    constexpr int MT = (W + 15) / 16;                  // tiles of 16 shift rows
    constexpr int QSX = S + 16 * MT;                   // sectors of the extended query
    constexpr int SB = RGH * 8;                        // bytes of one sector in hdesc (all rings, fp16)
    constexpr int KH = SB / 64;                        // k-steps per sector (32 rings = 64 B each)
    constexpr int QST = SB + 32;                       // bytes of one sector of the staged query (the padding keeps the A reads conflict-free)
    constexpr int HS = hdesc_stride_rgh(RGH, S);            // a keyframe's slot in hdesc (elements of 8 B): the copy, the fp16 sector key (+ the chunk-major image)
    constexpr int SPW = S / NWV;                       // query sectors per wave
    constexpr int NST = SPW * KH;                      // k-steps per wave and group
    constexpr int NW64 = (S + 63) / 64;                // 64-bit words of a sector mask
    constexpr int MW = (NW64 + 1) / 2;                 // ... stored as MW x 16 bytes
    static_assert(S % NWV == 0 && SB % 64 == 0 && NST % D == 0 && W <= 32 && S <= 224, "tiling");

    const int nbk = ab.nb;
    const ScreenArgs a = screen_args_of(ab, qi);
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: wave-derived addresses stay in SGPRs

    unsigned char *Qs = smem_raw;                                                // [QSX][QST]: sector-major fp16 query, 16 MT sectors repeated
    uint4 *rotq = reinterpret_cast<uint4 *>(Qs + (size_t)QSX * QST);             // [S][MW] the query's sector mask rotated right by s
    unsigned char *tile = reinterpret_cast<unsigned char *>(rotq + S * MW) + (size_t)wave * (kGroup * kTileStride);   // this wave's transposition tile
    f4v *part = reinterpret_cast<f4v *>(reinterpret_cast<unsigned char *>(rotq + S * MW) + (size_t)NWV * (kGroup * kTileStride));   // [2][NWV][MT][64] partial sums

    // ---- stage the query: its fp16 unit columns extended by 16 MT sectors, its rotated sector masks ----
    for (int idx = threadIdx.x; idx < QSX * (SB / 16); idx += blockDim.x) {
        const int cx = idx / (SB / 16), ch = idx - cx * (SB / 16);
        const int c = cx < S ? cx : cx - S;
        *reinterpret_cast<uint4 *>(Qs + (size_t)cx * QST + ch * 16) =
            *reinterpret_cast<const uint4 *>(reinterpret_cast<const unsigned char *>(a.q_hdesc) + (size_t)c * SB + ch * 16);
    }
    const bool q_bad = a.q_kmask[7] != 0;
    const float q_err = __uint_as_float(a.q_kmask[6]);                           // the scan's summed rounding-error norms (make_sc.hip)
    {
        unsigned long long qm[NW64];
#pragma unroll
        for (int i = 0; i < NW64; ++i) qm[i] = (unsigned long long)a.q_kmask[2 * i] | ((unsigned long long)(2 * i + 1 < 7 ? a.q_kmask[2 * i + 1] : 0u) << 32);
        for (int sft = threadIdx.x; sft < S; sft += blockDim.x) {                // bit x of rotq[s] = query sector (x + s) mod S
            unsigned long long rr[NW64];
            rotate_mask<S>(qm, sft, rr);
#pragma unroll
            for (int i = 0; i < MW; ++i) {
                const unsigned long long lo = rr[2 * i], hi = 2 * i + 1 < NW64 ? rr[2 * i + 1] : 0ull;
                rotq[sft * MW + i] = make_uint4((unsigned int)lo, (unsigned int)(lo >> 32), (unsigned int)hi, (unsigned int)(hi >> 32));
            }
        }
    }

    const int n16 = lane & 15, j4 = lane >> 4;           // MFMA layout: lane = (keyframe n, k-chunk j)
    const int n4 = lane >> 2, jl = lane & 3;             // load layout: four consecutive lanes = 64 consecutive bytes of keyframe n
    const int ngroups = (a.n + kGroup - 1) / kGroup;
    float run_min = __int_as_float(0x7f800000);                                  // wave 0: min d~ over screened keyframes
    float run_eps = 0.0f;                                                        // ... and the largest per-pair bound
    // first shifts (K0's output) of this group's keyframe n4, and of the next group's: requested one group ahead
    auto start_of = [&](int grp) {
        const int ci = grp * kGroup + n4;
        return a.starts[ci < a.n ? ci : a.n - 1];                                // (self-aligned batch: written by this workgroup a moment ago)
    };
    int b_nxt = start_of(bid < ngroups ? bid : 0);
    __syncthreads();

    int par = 0;
    for (int g = bid; g < ngroups; g += nbk, par ^= 1) {
        const int c_base = g * kGroup;
        f4v *part_cur = part + par * (NWV * MT * kWave);
        const int b_cur = b_nxt;
        b_nxt = start_of(g + nbk < ngroups ? g + nbk : g);

        // ---- query sectors SPW*w .. SPW*w + SPW-1 against all 16 keyframes ---------------------------------
        // hdesc is sector-major: the rings of one sector are consecutive (one 128-byte line at R = 64), so the keyframe's
        // ring shift is a rotation of whole sectors, every line is fetched once, and a wave walks SPW consecutive sectors
        // of each keyframe.  One k-step = 32 rings of one sector.  Load layout: lane 4n + j fetches rings 8j .. 8j+7 (16 B)
        // of keyframe n, so four consecutive lanes read 64 consecutive bytes and one instruction is one k-step of all 16
        // keyframes.  (The texture path coalesces per group of four lanes: in the MFMA's own layout, keyframe = lane & 15,
        // every such group touches four lines, which costs a quarter of the kernel.)  The step passes through a 1.5 KB tile
        // of LDS owned by this wave and comes back in the MFMA's B layout, lane (n, j) = lane 16j + n.
        const int ci_n = c_base + n16;
        const int first_slot = a.slot_base + c_base;
        const int last_rel = a.n - 1 - c_base;                                   // groups are consecutive slots: < 16 * 34 KB apart
        // the keyframe sector that meets query sector x at first shift b is (x - b) mod S
        const int cw0 = wrapS(wave * SPW - b_cur, S) * SB;
        // Addresses = one wave-uniform 64-bit base per group + 32-bit per-lane byte offsets.
        const char *hbase = reinterpret_cast<const char *>(a.hdesc + (size_t)first_slot * HS);
        const unsigned int hoff = (unsigned int)(n4 < last_rel ? n4 : last_rel) * (unsigned int)(HS * 8) + (unsigned int)jl * 16u;
        uint4 km[MW];
#pragma unroll
        for (int i = 0; i < MW; ++i) km[i] = make_uint4(0u, 0u, 0u, 0u);
        unsigned int kflag = 0;
        float kerr = 0.0f;
        if (wave == 0) {                                                         // the epilogue's operands: sector mask, rounding-error norm and flag of keyframe n
            const unsigned int *kp = a.kmask + (size_t)(first_slot + (n16 < last_rel ? n16 : last_rel)) * 8;
            const uint2 ef = *reinterpret_cast<const uint2 *>(kp + 6);
            kerr = __uint_as_float(ef.x); kflag = ef.y;
#pragma unroll
            for (int i = 0; i < MW; ++i) km[i] = *reinterpret_cast<const uint4 *>(kp + 4 * i);
            if (MW == 2) { km[1].z = 0u; km[1].w = 0u; }                         // words 6 and 7 are E and the flag, not sector bits
        }
        u32x4 ring[D];
        int cw_iss = cw0;                                                        // byte offset of the next step inside the keyframe
        const char *qbase = reinterpret_cast<const char *>(a.q_hdesc);
        // Every step refills its ring slot with the step D later; past the group's last step the refill reads 16 bytes of
        // the query's own copy instead (cache-resident, never used): a load behind a branch would make the compiler's
        // s_waitcnt counting give up on every load issued before it, and the ring would drain.
        auto issue = [&](int sl, bool real) {
            const char *bsel = real ? hbase : qbase;
            const unsigned int osel = real ? hoff + (unsigned int)cw_iss : (unsigned int)jl * 16u;
            ring[sl] = *reinterpret_cast<const u32x4 *>(bsel + osel);
            cw_iss += 64; cw_iss = cw_iss >= S * SB ? cw_iss - S * SB : cw_iss;
        };
#pragma unroll
        for (int sl = 0; sl < D; ++sl) issue(sl, true);
        unsigned char *tile_wr = tile + n4 * kTileStride + ((jl ^ ((n4 >> 1) & 2)) * 16);
        const unsigned char *tile_rd = tile + n16 * kTileStride + ((j4 ^ ((n16 >> 1) & 2)) * 16);
        f4v acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f4v{0.f, 0.f, 0.f, 0.f};
        // A fragment of step g (sector x = SPW*w + g / KH, rings 32 (g % KH) ..): row t = lane & 15 of tile m is the query
        // at sector x + 16 m + t
        const unsigned char *qlane = Qs + (size_t)(wave * SPW + n16) * QST + j4 * 16;
        // One k-step = 1 global load (issued D steps ahead), 1 LDS write + 1 LDS read through the tile (the step after
        // this one is staged while this one's read is in flight: LDS operations of a wave execute in order, one tile is
        // enough), MT LDS reads of the A fragments (requested one step ahead), MT MFMAs.  The scheduling fences keep LLVM
        // from hoisting every step's loads to the top of the unrolled run.
        h8 a_nxt[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a_nxt[m] = *reinterpret_cast<const h8 *>(qlane + (size_t)(16 * m) * QST);
        if (PROBE < 2) *reinterpret_cast<u32x4 *>(tile_wr) = ring[0];
        issue(0, NST > D);
#pragma unroll 1
        for (int r = 0; r < NST / D; ++r) {
#pragma unroll
            for (int xb = 0; xb < D; ++xb) {
                const int sn = (xb + 1) % D;                                     // slot of the step after this one
                const int st = r * D + xb;                                       // wave-uniform
                const bool more = (st + 1 + D) < NST;
                if (PROBE == 2) {
                    acc[0][0] += __uint_as_float(ring[sn][0] ^ ring[sn][1] ^ ring[sn][2] ^ ring[sn][3]);
                    issue(sn, more);
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
                const h8 bfrag = *reinterpret_cast<const h8 *>(tile_rd);         // this step, staged one step ago
                // (the slot is staged before it is refilled: the old value is dead when the load is issued, so the load
                // returns into the same registers and the ring never has to be copied)
                *reinterpret_cast<u32x4 *>(tile_wr) = ring[sn];
                issue(sn, more);
                // next step's A fragments (past the end they re-read inside the staged query, values unused)
                const int sa = st + 1 < NST ? st + 1 : 0;
                const unsigned char *qn = qlane + ((sa / KH) * QST + (sa % KH) * 64);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const h8 afrag = a_nxt[m];
                    a_nxt[m] = *reinterpret_cast<const h8 *>(qn + (size_t)(16 * m) * QST);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, bfrag, acc[m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) part_cur[(wave * MT + m) * kWave + lane] = acc[m];
        __syncthreads();                                                         // partial sums are in LDS (double-buffered: one barrier per group)

        // ---- epilogue (wave 0): lane (n, q) holds shifts 16m + 4q .. 16m + 4q+3 of keyframe n ----------------------------
        if (wave == 0) {
            const int b_raw = __shfl(b_cur, 4 * n16, kWave);                     // lane 4n holds keyframe n's first shift
            const bool b_open = b_raw < 0;                                       // kAlignUndecided: the exact pass aligns and scores this pair
            const int b_n = b_open ? 0 : b_raw;
            float dmin = __int_as_float(0x7f800000);
            // per shift the interval the reference's distance lies in (as the second form's finishing: the two descriptors' recorded
            // rounding-error norms over n_eff + this form's accumulation -- a wave's chain of NST products of 32 terms, NWV partials)
            const float kInfF = __int_as_float(0x7f800000);
            const bool want_mask = a.out_smask != nullptr;
            const float e_pair = (q_err + kerr) * 1.002f;
            const bool e_ok = e_pair >= 0.0f && e_pair < 1.0f;
            constexpr float kAcc1 = (float)(NST * 32 + NWV + 4) * 1.1920929e-7f * 1.002f + 2.0e-6f;
            float dlo[MT][4], hi_min = kInfF;
            int ne_min = 0x7fffffff;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f4v s = part_cur[m * kWave + lane];
#pragma unroll
                for (int w = 1; w < NWV; ++w) s += part_cur[(w * MT + m) * kWave + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = 16 * m + 4 * j4 + r;
                    // effective sectors (D.h:1523-1526): both columns non-zero; keyframe sector y meets query sector y + b + t
                    int ri = b_n + t; ri = ri >= S ? ri - S : ri;
                    int ne = 0;
#pragma unroll
                    for (int i = 0; i < MW; ++i) {
                        const uint4 rq = rotq[ri * MW + i];
                        ne += __popc(rq.x & km[i].x) + __popc(rq.y & km[i].y) + __popc(rq.z & km[i].z) + __popc(rq.w & km[i].w);
                    }
                    const float d = 1.0f - s[r] * __builtin_amdgcn_rcpf((float)ne);   // (v_rcp_f32, 1 ulp: a screened value; the bound's 2e-6 covers it -- the IEEE quotient is ten instructions, thirteen times per pair)
                    if (t < W && ne > 0 && d < dmin) dmin = d;                   // n_eff = 0: 0/0 in the reference, never wins
                    if (t < W && ne > 0) ne_min = ne < ne_min ? ne : ne_min;
                    dlo[m][r] = kInfF;
                    if (want_mask && t < W && ne > 0) {
                        const float et = e_ok ? fminf(e_pair / (float)ne + kAcc1, kScreenEps) : kScreenEps;
                        dlo[m][r] = d - et; hi_min = fminf(hi_min, d + et);
                    }
                }
            }
            dmin = fminf(dmin, __shfl_xor(dmin, 16, kWave));
            dmin = fminf(dmin, __shfl_xor(dmin, 32, kWave));
            hi_min = fminf(hi_min, __shfl_xor(hi_min, 16, kWave));
            hi_min = fminf(hi_min, __shfl_xor(hi_min, 32, kWave));
            const bool mine = lane < kGroup && ci_n < a.n;
            const bool exact_only = q_bad || kflag != 0 || b_open || !(dmin == dmin);
            if (mine) a.out_approx[ci_n] = exact_only ? __int_as_float(0xff800000) : dmin;
            if (a.out_smask) {
                // the shifts that can still hold (or tie for) the pair's exact minimum: the lower end of the interval not above the
                // smallest upper end; every shift for a pair the screening cannot bound
                unsigned int mb = 0u;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int t = 16 * m + 4 * j4 + r;
                        if (t < W && (exact_only || dlo[m][r] <= hi_min)) mb |= 1u << t;
                    }
                mb |= __shfl_xor(mb, 16, kWave);
                mb |= __shfl_xor(mb, 32, kWave);
                if (mine) a.out_smask[ci_n] = b_open ? 0u : mb;
            }
            float contrib = (mine && !exact_only) ? dmin : __int_as_float(0x7f800000);
            // the pair's bound on |d~ - d| (fewest effective sectors of its shifts), into the launch's largest
            ne_min = min(ne_min, __shfl_xor(ne_min, 16, kWave));
            ne_min = min(ne_min, __shfl_xor(ne_min, 32, kWave));
            float peps = (mine && !exact_only && ne_min != 0x7fffffff) ? (e_ok ? fminf(e_pair / (float)ne_min + kAcc1, kScreenEps) : kScreenEps) : 0.0f;
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) { contrib = fminf(contrib, __shfl_xor(contrib, off, kWave)); peps = fmaxf(peps, __shfl_xor(peps, off, kWave)); }
            run_min = fminf(run_min, __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(contrib))));
            run_eps = fmaxf(run_eps, __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(peps))));
        }
    }
    if (wave == 0 && lane == 0 && run_min < __int_as_float(0x7f800000)) atomicMin(a.t_min, float_to_ordered_u(run_min));
    if (wave == 0 && lane == 0 && run_eps > 0.0f) atomicMax(a.t_min + kTminEpsOffset, __float_as_uint(run_eps));
}

// One launch = the screening products of a batch of scans and, in further workgroups of the same grid, the alignment of
// the NEXT batch (fa.align_blocks = 0: none): the products are HBM bound, the alignment matrix-core bound, and the workgroups
// of the second take the third wave slot per SIMD the first leaves free.  (Two kernels on two streams do the same in
// principle; measured, the dispatcher then lets the alignment crowd out the products.)
//
// Order of the workgroups -- the dispatcher hands them out by index, round robin over the 8 XCDs (index & 7), each with
// its own 4 MB L2.  Along one XCD's sequence (index >> 3) the pattern is: the nq workgroups that walk the SAME keyframe
// groups for the nq queries of the batch, then one workgroup of the alignment.  The nq start together and read the same
// lines within microseconds of each other, so all but the first find them in that XCD's L2: the products' time drops from
// 92 to 56 us per four scans (the same workgroups query-major: every query streams the copy from HBM / the Infinity Cache
// on its own).  The short alignment workgroups are spread over the whole launch.  Slots past either role's count exit.
struct ScreenFusedArgs { ScreenBatchArgs prod; ScreenBatchArgs next; int align_blocks; };

__host__ __device__ inline int fused_patterned_blocks(int nq, int nbk, int align_blocks) { return 8 * ((nbk + 7) >> 3) * (nq + (align_blocks > 0 ? 1 : 0)); }

template <int RG, int S, int W, int D, int OCC, int PROBE = 0>
__global__ __launch_bounds__(kScreenWaves * kWave, OCC) void sc_screen_kernel(ScreenFusedArgs fa)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_fused[];
    const int b = (int)blockIdx.x, nq = fa.prod.nq, nbk = fa.prod.nb, X = fa.align_blocks;
    const int cyc_len = nq + (X > 0 ? 1 : 0);
    const int patterned = fused_patterned_blocks(nq, nbk, X);
    if (b < patterned) {
        const int xcd = b & 7, j = b >> 3;
        const int grp = j / cyc_len, cyc = j - grp * cyc_len;
        const int idx = grp * 8 + xcd;
        if (cyc < nq) {
            if (idx < nbk) {
                if (fa.prod.self_align) {
                    // a batch nobody aligned in advance (a blocking call): the workgroup aligns the groups it is about to walk -- a
                    // phase of one group per wave instead of a launch of 10-13 us in front of this one.  Its first shifts pass through
                    // memory inside the workgroup: one CU, one L1 -- a workgroup-scope fence and the barrier order them (an agent-scope
                    // fence writes the XCD's L2 back: 33 us per launch, measured)
                    sc_align_role<RG, S, W>(fa.prod, cyc * nbk + idx, smem_fused, true);
                    __threadfence_block();
                    __syncthreads();
                }
                sc_screen_role<hdesc_rgh(RG), S, W, D, PROBE>(fa.prod, cyc, idx, smem_fused);
            }
        }
        else if (idx < X) sc_align_role<RG, S, W>(fa.next, idx, smem_fused);
    } else {
        const int idx = 8 * ((nbk + 7) >> 3) + (b - patterned);             // more alignment workgroups than pattern slots
        if (idx < X) sc_align_role<RG, S, W>(fa.next, idx, smem_fused);
    }
}

#ifndef SCL_ALIGN_OCC
#define SCL_ALIGN_OCC 2
#endif
template <int RG, int S, int W>
__global__ __launch_bounds__(kScreenWaves * kWave, SCL_ALIGN_OCC) void sc_align_kernel(ScreenBatchArgs ab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_align[];
    sc_align_role<RG, S, W>(ab, (int)blockIdx.x, smem_align);
}


// ---- select: survivors of the screening (ascending slot order) + the ring-key top-k ----------------------------
// One workgroup per query.  survivors[i] = database slots (ascending) whose d~ <= min d~ + 2 eps, or flagged
// "score exactly"; *n_surv their number.  Also the k nearest ring keys from the metric the screening pass produced
// (k rounds of "smallest key larger than the previous pick"; keys (d2 bits << 32 | position) are unique).
struct SelectArgs {
    const float *approx; const float *d2; int n; int slot_base;
    unsigned int *t_min; int *survivors; int *n_surv;       // survivors == nullptr: ring-key top-k only
    int k; float exclude_eps; int *topk_idx; float *topk_d2;
    unsigned long long *surv_stats;
};
struct SelectBatchArgs { SelectArgs q[kMaxScreenBatch]; };

__global__ __launch_bounds__(1024) void sc_select_kernel(SelectBatchArgs sb)
{
    const SelectArgs &a = sb.q[blockIdx.x];
    __shared__ int wcount[16];
    __shared__ unsigned long long sk[16];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const unsigned int tm = *a.t_min;
    float thr;
    {
        const unsigned int b = (tm >> 31) ? (tm & 0x7fffffffu) : ~tm;            // inverse of float_to_ordered_u
        const unsigned int ew = a.t_min[kTminEpsOffset];                       // the launch's largest per-pair bound (0: not recorded)
        const float two_eps = ew ? fminf(2.0f * __uint_as_float(ew) * 1.0001f, 2.0f * kScreenEps) : 2.0f * kScreenEps;
        thr = tm == 0xffffffffu ? __int_as_float(0xff800000) : __int_as_float((int)b) + two_eps;
    }
    // Every wave owns a contiguous part of the range and walks it 4 x 64 entries at a time (four independent loads in flight,
    // one barrier in all: the walk in steps of the whole workgroup paid a load's latency and three barriers per step): count,
    // prefix over the waves, then the same walk writes the list -- ascending slots.
    {
        const int n_s = a.survivors ? a.n : 0;
        const int nwv = (int)blockDim.x / kWave;
        const int per_wave = (((n_s + nwv - 1) / nwv) + kWave - 1) / kWave * kWave;
        const int wlo = wv * per_wave;
        const int whi = wlo + per_wave < n_s ? wlo + per_wave : n_s;
        int cnt = 0;
        for (int base = wlo; base < whi; base += 4 * kWave) {
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = base + u * kWave + lane; d[u] = i < whi ? a.approx[i] : __int_as_float(0x7fc00000); }   // NaN: never passes
#pragma unroll
            for (int u = 0; u < 4; ++u) cnt += __popcll(__builtin_amdgcn_ballot_w64(d[u] <= thr));      // -inf (score exactly) always passes
        }
        if (lane == 0) wcount[wv] = cnt;
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < nwv; ++w) { const int t = wcount[w]; if (w < wv) before += t; total += t; }
        for (int base = wlo; base < whi; base += 4 * kWave) {
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = base + u * kWave + lane; d[u] = i < whi ? a.approx[i] : __int_as_float(0x7fc00000); }   // NaN: never passes
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool keep = d[u] <= thr;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
                if (keep) a.survivors[before + __popcll(m & ((1ull << lane) - 1ull))] = a.slot_base + base + u * kWave + lane;
                before += __popcll(m);
            }
        }
        if (threadIdx.x == 0 && a.survivors) { *a.n_surv = total; *a.t_min = 0xffffffffu; a.t_min[kTminEpsOffset] = 0u; }   // armed for the next launch (stream ordered)
        if (threadIdx.x == 0 && a.survivors && a.surv_stats) {
            atomicAdd(a.surv_stats, (unsigned long long)total);
            atomicMax(a.surv_stats + 1, (unsigned long long)total);
            atomicAdd(a.surv_stats + 2, 1ull);
        }
    }

    const unsigned long long none = ~0ull;
    unsigned long long prev = 0ull;
    bool first = true;
    for (int round = 0; round < a.k; ++round) {
        unsigned long long mine = none;
        for (int i = threadIdx.x; i < a.n; i += blockDim.x) {
            const float r = a.d2[i];
            const bool excluded = (a.exclude_eps > 0.0f) && (r <= a.exclude_eps);
            if (excluded || !(r < 3.402823466e+38f)) continue;
            const unsigned long long key = ((unsigned long long)(unsigned)__float_as_int(r) << 32) | (unsigned)i;
            if ((first || key > prev) && key < mine) mine = key;
        }
        mine = wave_min_u64(mine);
        __syncthreads();
        if (lane == 0) sk[wv] = mine;
        __syncthreads();
        unsigned long long m = sk[0];
        for (int w = 1; w < (int)blockDim.x / kWave; ++w) m = sk[w] < m ? sk[w] : m;
        if (threadIdx.x == 0) {
            if (m == none) { a.topk_idx[round] = -1; a.topk_d2[round] = 3.402823466e+38f; }
            else { a.topk_idx[round] = a.slot_base + (int)(unsigned)(m & 0xffffffffull); a.topk_d2[round] = __int_as_float((int)(m >> 32)); }
        }
        if (m == none) {
            for (int r2 = round + 1 + (int)threadIdx.x; r2 < a.k; r2 += blockDim.x) { a.topk_idx[r2] = -1; a.topk_d2[r2] = 3.402823466e+38f; }
            break;
        }
        prev = m; first = false;
    }
}

}  // namespace

bool sc_screen_supported(const DbView &db, int SR)
{
    static const bool off = [] { const char *e = getenv("SCL_SCREEN"); return e && e[0] == '0'; }();
    if (off) return false;
    return (db.RG == 16 && db.R == 64 && db.S == 120 && SR == 6) || sc_screen_is_wide(db, SR);
}

bool sc_screen_is_wide(const DbView &db, int SR) { return db.RG == 20 && db.R == 80 && db.S == 180 && SR == 9; }

float sc_screen_eps() { return kScreenEps; }

hipError_t launch_sc_select_batch(const ScreenBatch &sb, hipStream_t stream)
{
    if (sb.nq < 1 || sb.nq > kMaxScreenBatch) return hipErrorInvalidValue;
    SelectBatchArgs sel{};
    for (int i = 0; i < sb.nq; ++i) {
        SelectArgs &s = sel.q[i];
        s.approx = sb.approx + (size_t)sb.buf[i] * sb.pair_stride; s.d2 = sb.ring_d2 + (size_t)sb.buf[i] * sb.pair_stride;
        s.n = sb.n[i]; s.slot_base = sb.base[i]; s.t_min = sb.t_min + sb.buf[i];
        s.survivors = sb.survivors ? sb.survivors + (size_t)sb.buf[i] * sb.pair_stride : nullptr; s.n_surv = sb.n_surv ? sb.n_surv + sb.buf[i] : nullptr;
        s.surv_stats = sb.surv_stats;
        s.k = sb.k; s.exclude_eps = sb.exclude_eps; s.topk_idx = sb.topk_idx + sb.buf[i] * kTailTopMaxK; s.topk_d2 = sb.topk_d2 + sb.buf[i] * kTailTopMaxK;
    }
    hipLaunchKernelGGL(sc_select_kernel, dim3(sb.nq), dim3(1024), 0, stream, sel);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// K1s, second form (64 x 120): the KEYFRAME is the shared operand of the matrix product, the scans of the launch are its
// columns.  With  sim[t] = sum_x sum_r Q[r][x + t] K[r][(x - b) mod S]  and  y = x + t - b ... rewritten over keyframe-side
// sectors:  sim_q[t] = sum_y sum_r K[r][(y - t) mod S] * Q_q[r][(y + b_q) mod S]:
//   A[t][(y, r)] = K[r][(y - t) mod S]     -- a Toeplitz image of ONE keyframe: row t at step y + 1 is row t - 1 at step y (*)
//   B[(y, r)][q] = Q_q[r][(y + b_q) mod S] -- column q = scan q of the launch, rotated by ITS first shift for this keyframe
// so one v_mfma_f32_16x16x32_f16 scores one keyframe against 16 scans, the keyframe's bytes are fetched once per launch and
// workgroup instead of once per pair (the first form pulls 16.6 KB per pair through the L2: that rate bounds it), and the
// per-pair operand -- the rotated scan -- comes out of LDS.  16 scans x 120 sectors x 128 B do not fit 160 KB, so a workgroup
// holds ONE HALF of the rings (32 of 64: 64 B per sector, 123 sectors per scan so that four consecutive steps never wrap =
// 126 KB) and two workgroups on the same XCD walk the same keyframes, one per half (each reads its 64 B of every 128-byte
// line: the second finds the line in that XCD's L2); the two partial sums of a pair meet in sc_screen2_finish_kernel.
// With the rows counted the other way -- row m <-> shift t = 12 - m, A[m][(y, r)] = K[r][(y + m) mod S], the scan rotated by
// b_q + 12 -- the fragment of step y is SIXTEEN CONSECUTIVE SECTORS y .. y+15 of the keyframe (rows 13 .. 15 are spare), and
// the fragment of step y + u is the same sixteen sectors one row further on: ONE load serves FOUR k-steps.  The rows are not even
// moved: step y + u multiplies the UNROTATED fragment, so row m of accumulator u collects what belongs to row m - u (the same
// products in the same order), and the four accumulators are joined once per keyframe with their rows shifted back (row s =
// sum over u of row s + u: three of the sixteen lane groups' values come from the lane 16 further on, ds_bpermute) -- until then
// twelve DPP row rotations per iteration made the VALU the busiest unit of the kernel.  The
// load reads the keyframe's chunk-major image (kernels.hpp, hdesc2: [half][chunk j][sector] x 16 B, 16 sectors repeated at the
// end), where lane (m, j) = (lane & 15, lane >> 4) finds rings 8j .. 8j+7 of sector y + m at 16 (y + m): the 16 lanes of a
// chunk read 256 consecutive bytes, no wrap, no address arithmetic (scalar base + 64 B per iteration).  (From the
// sector-major copy the same fragment is 64 pieces of 16 B in 16 lines: the texture addresser was 86 % busy and the kernel
// took 84 us.)  Four accumulators (step mod 4): chains of 30 MFMAs, 960 products each -- inside the accumulation bound of the
// first form (1 924 x 2^-23 for 8 partial sums).
// One wave per keyframe; the waves of a launch take keyframes gw, gw + W, ... of the union of the scans' ranges.
// LDS image of the scans: four scans share a 256-byte row per sector -- row (p, s) = sector s of scans 4p .. 4p+3, 64 B each --
// so a lane's address is p * QUAD + 256 s + 64 (q & 3) + 16 j and its 16-byte slot in the 256-byte bank row does not
// depend on the sector: whatever the 16 first shifts are, the 16 lanes of every ds_read_b128 group (MI355X_MICROARCH.md,
// LDS: {0-3, 12-15, 20-27}, ...: all 16 scans, eight with chunk j and eight with j + 1) hit 16 different slots when
// QUAD / 16 = 2 mod 4.  (Scan-major rows of 64 B put the eight same-chunk lanes of a group on two slots: 4-way conflicts,
// 99 us per launch instead of ...)
// Shapes of the second form.  64 x 120: two ring halves, 13 shifts in one pass, four k-steps per fragment load (13 + 3 rows),
// 16 scans per launch.  80 x 180 (80 rings, 19 shifts): FIVE ring slices of 16, a k-step's 32 products being 16 rings of TWO consecutive
// sectors (SPK = 2): lane (m, j) of the keyframe fragment reads chunk 2 slice + (j & 1) of sector y + m + (j >> 1) -- the same chunk-major
// image --, the scan fragment is 64 consecutive bytes of a row pair of an image that holds eight scans of 32 B per 256-byte row (a lane's
// slot 2 (q & 7) + (j & 1) whatever its sector: conflict free), an iteration is two k-steps (four sectors, row shift two between them);
// the shifts in two passes of 13 and 6 rows that share the SCAN's fragment (PS = 12: pass 1 on this iteration's keyframe fragment, pass 0 on
// the one of three iterations ago); 16 scans per launch (16 x 183 sectors x 32 B = 94 KB of LDS).  Until the middle of round 4: 96 padded
// rings as three thirds of 32, twelve scans per launch, each pass reading the scan at its own offset (S2_WIDE_THIRDS).
// Waves per workgroup (one workgroup per CU: the scans fill the LDS) and ring slots of the fragment loads; -D overrides for experiments.
// 64 x 120: 113 registers -> 16 waves (four per SIMD) hide the chains that two per SIMD left open (products 54 -> 48 us per 16 scans;
// six ring slots instead of ten: 1 us slower).  80 x 180: 108 registers, sixteen waves, six fragment buffers.
#ifndef S2_FRAG_SHIFT
#define S2_FRAG_SHIFT 1               // products, second form: three of four keyframe fragments by row shifts of loaded ones (0: every fragment loaded)
#endif
#ifndef S2NBUF_A
#define S2NBUF_A 9                    // fragment buffers of the products' second form (loads per keyframe: 9 at S = 120, 12 at S = 180)
#endif
#ifndef S2NBUF_B
#define S2NBUF_B 6
#endif
#ifndef S2RS_A
#define S2RS_A 10
#endif
#ifndef S2RS_B
#define S2RS_B 9
#endif
#ifndef S2WV_B
#define S2WV_B 16
#endif
#ifndef S2WV_A
#define S2WV_A 16
#endif
#ifndef S2WVF_A
#define S2WVF_A 12
#endif
#ifndef S2XW_A
#define S2XW_A 4
#endif
#ifndef S2XPRIO
#define S2XPRIO 0
#endif
#ifndef S2XW_B
#define S2XW_B 2
#endif
#ifdef S2_STAMP
// experiments only (scripts/build_variant.sh): s_memtime of one wave of two workgroups at every iteration of its second keyframe
__device__ unsigned long long g_s2_stamps[4 * 64];
#endif
template <int RG, int S, int W> struct S2Cfg;
// WV: waves of the products per workgroup; XW: EXTRA waves of the same workgroup that align the NEXT batch and finish the PREVIOUS
// one beside the products (sc_screen2_kernel): the products leave two thirds of the vector and matrix-core issue slots idle
// NP ring parts, NPASS passes of 13 shift rows, STEPS k-steps per iteration (= per keyframe fragment), SPK sectors per k-step:
// a k-step's 32 products are 32 rings of one sector (SPK = 1) or 16 rings of two consecutive sectors (SPK = 2)
template <> struct S2Cfg<16, 120, 13> { static constexpr int NP = 2, NPASS = 1, STEPS = 4, SPK = 1, PS = 13, NQ = 16, RS = S2RS_A, WV = S2WV_A, WVF = S2WVF_A, XW = S2XW_A, NBUF = S2NBUF_A; };
#ifdef S2_WIDE_THIRDS
// (rounds 3-4: 96 padded rings as three thirds of 32, twelve scans per launch -- 141 KB of LDS; 37 % of the matrix-core work useful)
template <> struct S2Cfg<20, 180, 19> { static constexpr int NP = 3, NPASS = 2, STEPS = 4, SPK = 1, PS = 13, NQ = 12, RS = 9, WV = S2WV_B, WVF = 8, XW = S2XW_B, NBUF = S2NBUF_B; };
#else
// 80 rings as five slices of 16, two sectors per k-step: no padded rings, sixteen scans per launch (16 x 183 sectors x 32 B = 94 KB)
// PS = 12: the two passes share the scans' fragments -- pass 1 (shifts 5 .. 0 in rows 1 .. 6) multiplies the keyframe fragment of this
// iteration, pass 0 (shifts 18 .. 6) the fragment of three iterations (12 sectors) ago: one LDS read per TWO matrix products (with PS = 13
// each pass read the scan at its own offset, and the LDS array was as busy as the matrix cores)
#ifndef S2_WIDE_PS
#define S2_WIDE_PS 12
#endif
template <> struct S2Cfg<20, 180, 19> { static constexpr int NP = 5, NPASS = 2, STEPS = 2, SPK = 2, PS = S2_WIDE_PS, NQ = 16, RS = 9, WV = S2WV_B, WVF = 8, XW = S2XW_B, NBUF = S2NBUF_B; };
#endif
constexpr int kS2PassRows = 13;                    // shift rows per pass: row m of pass p = shift W - 1 - PS p - m (PS = 13; 12: row 0 of pass 1 repeats row 12 of pass 0)

// LDS image of the scans (see above): quad stride in bytes, 2 mod 4 sixteen-byte slots
// (SECT = sectors per iteration; SPK = 2: eight scans of 32 B share a row, a lane's slot is 2 (q & 7) + (j & 1) whatever its sector:
//  whole rows per block of eight scans)
constexpr int s2_quad(int S, int SECT, int SPK) { return SPK == 2 ? (S + SECT - 1) * 256 : ((((S + SECT - 1) * 256 / 16) % 4 == 2) ? (S + SECT - 1) * 256 : (S + SECT - 1) * 256 + 32); }
template <int RG, int S, int W> constexpr size_t s2_lds()
{
    using C = S2Cfg<RG, S, W>;
    return (size_t)(C::NQ / (4 * C::SPK)) * s2_quad(S, C::STEPS * C::SPK, C::SPK);
}
// The partial sums of a pair and ring part: per pass the groups of four tile rows that hold shifts (rows 4 g .. 4 g + 3 of pass p: shifts
// W - 1 - PS p - 4 g - i), packed -- at 80 x 180 the second pass has six shifts in its sixteen rows: 24 floats per part instead of 32
template <int RG, int S, int W> constexpr bool s2_group_used(int p, int g4) { return 4 * g4 < kS2PassRows && W - 1 - S2Cfg<RG, S, W>::PS * p - 4 * g4 >= 0; }
template <int RG, int S, int W> constexpr int s2_pass_f4(int p) { int n = 0; for (int g = 0; g < 4; ++g) n += s2_group_used<RG, S, W>(p, g) ? 1 : 0; return n; }
template <int RG, int S, int W> constexpr int s2_pass_off(int p) { int o = 0; for (int q = 0; q < p; ++q) o += s2_pass_f4<RG, S, W>(q); return o; }   // in float4
template <int RG, int S, int W> constexpr int s2_part_f4() { return s2_pass_off<RG, S, W>(S2Cfg<RG, S, W>::NPASS); }
template <int RG, int S, int W> constexpr int s2_part_floats() { return S2Cfg<RG, S, W>::NP * s2_part_f4<RG, S, W>() * 4; }   // partial sums per pair
// LDS tile of one extra wave: the alignment image (both parts) / the exact evaluation's scratch, or the finishing's rotated masks
template <int S> constexpr int s2_xlds() { return Align2Cfg<S>::LDS_WAVE > S * ((((S + 63) / 64) + 1) / 2) * 16 ? Align2Cfg<S>::LDS_WAVE : S * ((((S + 63) / 64) + 1) / 2) * 16; }
template <int RG, int S, int W, bool FUSED> constexpr size_t s2_lds_total() { return s2_lds<RG, S, W>() + (FUSED ? (size_t)S2Cfg<RG, S, W>::XW * s2_xlds<S>() : 0); }
template <int RG, int S, int W, bool FUSED> constexpr int s2_waves() { return FUSED ? S2Cfg<RG, S, W>::WVF + S2Cfg<RG, S, W>::XW : S2Cfg<RG, S, W>::WV; }

struct Screen2Args {
    ScreenBatchArgs prod;
    float *part;                                   // [nq][pair_stride][ring parts][passes][16] partial sums (scratch of one launch)
    int u_lo, u_n;                                 // union of the scans' ranges (database slots)
    int nwg;                                       // workgroups of the products (a multiple of 8 x ring parts)
};

template <int RG, int S, int W>
__device__ __forceinline__ void sc_screen2_finish_waves(const Screen2Args &fa, const int nb64, const int unit0, const int unit_stride, unsigned char *lds_wave);

// What the extra waves of a launch do: the alignment of the batch BEHIND this one (a_n > 0) and the finishing of the batch in
// FRONT of it (f_nb64 > 0), whose products the launch before wrote.
struct Screen2Extra {
    ScreenBatchArgs next; const unsigned char *halign; int a_lo, a_n;
    Screen2Args prev; int f_nb64;
};

// FUSED: the workgroup has XW extra waves behind its WVF product waves; otherwise WV product waves and nothing else
template <int RG, int S, int W, bool FUSED>
__global__ __launch_bounds__((s2_waves<RG, S, W, FUSED>() * kWave), 1) void sc_screen2_kernel(Screen2Args fa, Screen2Extra xa)
{
    constexpr int WVP = FUSED ? S2Cfg<RG, S, W>::WVF : S2Cfg<RG, S, W>::WV;
    using C = S2Cfg<RG, S, W>;
    constexpr int NP = C::NP, NPASS = C::NPASS, STEPS = C::STEPS, NQ = C::NQ, SPK = C::SPK, PS = C::PS;
    constexpr bool AOFF = PS != kS2PassRows;           // the passes share the scans' fragments (see S2Cfg)
    constexpr int NB = AOFF ? 1 : NPASS;               // fragment sets of the scans per iteration
    constexpr int LAG = AOFF ? PS / 4 : 0;             // iterations between the keyframe fragments of pass 0 and pass 1
    constexpr int RGH = hdesc_rgh(RG);
    constexpr int SB = RGH * 8;                        // bytes of one sector in hdesc (all rings, fp16)
    constexpr int HS = hdesc_stride(RG, S);
    constexpr int SECT = STEPS * SPK;                  // sectors per iteration
    constexpr int NIT = S / SECT;                      // iterations (of STEPS k-steps) per keyframe
    constexpr int ROWS = S + SECT - 1;                 // sectors per scan in LDS: the reads of an iteration never wrap
    constexpr int QUAD = s2_quad(S, SECT, SPK);        // bytes of a block of SPR scans in the image
    constexpr int SPR = 4 * SPK, SCB = 64 / SPK;       // scans per 256-byte row of the image; a scan's bytes in it (its ring part of one sector)
    static_assert(SB >= SCB * NP && SB == 64 * ((NP * SCB + 63) / 64) && S % SECT == 0 && SECT == 4 && kS2PassRows + SPK * (STEPS - 1) <= 16 && W <= PS * (NPASS - 1) + kS2PassRows && (!AOFF || (NPASS == 2 && PS == 12))
                  && NQ % SPR == 0 && NQ <= kMaxScreenBatch && (SPK == 1 || SPK == 2), "second form");
    static_assert(SPK == 2 ? QUAD % 256 == 0 : (QUAD / 16) % 4 == 2, "bank slots of the image's blocks");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem2[];
    const ScreenBatchArgs &ab = fa.prod;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // workgroup -> (XCD, ring part, index): the parts of index i sit next to each other on one XCD
    const int b = (int)blockIdx.x, xcd = b & 7, jx = b >> 3;
    const int part = jx % NP;
    const int gi = (jx / NP) * 8 + xcd;                // 0 .. nwg / NP - 1
    const int waves_part = (fa.nwg / NP) * WVP;
    const int gw = gi * WVP + wave;

#ifdef S2_STAMP
#define S2_STAMP_AT(slot) do { if ((wave == 0 || wave == 5) && (b == 0 || b == 37) && lane == 0) g_s2_stamps[((b ? 2 : 0) + (wave ? 1 : 0)) * 64 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S2_STAMP_AT(slot) do { } while (0)
#endif
    S2_STAMP_AT(32);
    const bool xwave = FUSED && wave >= WVP;           // an extra wave (alignment / finishing): no part in the products
    const int c16 = lane & 15, j4 = lane >> 4;         // A: row m = c16; B / output: scan q = c16
    // columns past the launch's scans shadow the column 12 below: the SAME address as a lane of the same ds_read_b128 group
    // (lanes 12-15 beside 0-3, 28-31 beside 16-19, ...): a broadcast.  (Shadowing column c - 4 put them on the slots of
    // columns 0-3: 2-way conflicts on every read, half of the LDS cycles.)
    const int cq = c16 < NQ ? c16 : c16 - 12;
    const ScreenQuery sq = ab.q[cq];                   // (entries past nq copy entry 0: valid memory, never stored)
    const int *starts_q = ab.starts + (size_t)sq.buf * (size_t)ab.pair_stride;
    const bool q_live = c16 < ab.nq && c16 < NQ;
    // A side: row m of the fragment of step y is sector y + m of the chunk-major image (no wrap: 16 sectors repeat at its end)
    // (SPK = 2: the fragment's 32 products are chunks 2 part, 2 part + 1 of sector y + m and of sector y + m + 1: lanes j4 = 2, 3 read one
    //  sector further on -- the same 256 consecutive bytes per 16 lanes)
    const int lsec = c16 + (SPK == 2 ? (j4 >> 1) : 0);
    const unsigned int a_lane = (unsigned int)(hdesc2_offset(RG, S) * 8 + (((SPK == 2 ? part * 2 + (j4 & 1) : part * 4 + j4)) * (S + 16) + lsec) * 16);
    const unsigned char *hd = reinterpret_cast<const unsigned char *>(ab.hdesc);
    auto kf_base = [&](int k) -> const unsigned char * {
        int idx = gw + k * waves_part;
        idx = idx < fa.u_n ? idx : fa.u_n - 1;
        idx = idx < 0 ? 0 : idx;
        return hd + (size_t)(fa.u_lo + idx) * (size_t)(HS * 8);
    };
    auto first_shift = [&](int k) -> int {             // scan q's first shift for the wave's k-th keyframe (0 where it has none)
        int idx = gw + k * waves_part;
        idx = idx < fa.u_n ? idx : fa.u_n - 1;
        const int ci = fa.u_lo + idx - sq.base;
        const bool ok = ci >= 0 && ci < sq.n;
        return starts_q[ok ? ci : 0];
    };
    const int nk = gw < fa.u_n ? (fa.u_n - gw + waves_part - 1) / waves_part : 0;
    const unsigned int q_lds = (unsigned int)((cq / SPR) * QUAD + (cq % SPR) * SCB + (SPK == 2 ? (j4 & 1) * 16 + (j4 >> 1) * 256 : j4 * 16));
    const int up16 = ((lane + 16) & 63) * 4;           // ds_bpermute address of the lane that holds the next four rows of this column
    // The fragment of iteration i (keyframe sectors 4 i + m, m = 0 .. 15) shares twelve of its sixteen sectors with the one before: only
    // every FOURTH fragment is loaded (sectors 16 g + m: the same 1 KB load, a quarter as many), the three between come out of two loaded
    // ones by row shifts inside the rows of 16 lanes -- lane m of offset 4 o takes lane m + 4 o of load g, or lane m + 4 o - 16 of load
    // g + 1 (two DPP moves per register).  The texture addresser, the busiest unit of the kernel while every fragment was loaded
    // (TA_BUSY = the kernel's duration), sees a quarter of the requests.
    constexpr int NGR = (NIT + 3) / 4;                  // groups of four iterations per keyframe (the last may be short)
    constexpr int NL = ((NIT - 1) % 4 == 0) ? NGR : NGR + 1;   // loads per keyframe: the last group needs load NGR only if it has a derived fragment
    // NBUF buffers, statically indexed across keyframes (NL % NBUF == 0): load j lives in F[j % NBUF] from its issue -- NBUF - 1 groups
    // ahead of the group that multiplies it -- to the last fragment derived from it.  Measured with three buffers (two groups ahead):
    // every fourth iteration waited 450-1000 cycles for its load (stamps: 250 cycles per iteration, 700-1300 where a group begins).
    constexpr int NBUF = C::NBUF;
    static_assert(NL % NBUF == 0 && NBUF >= 3, "fragment buffers");
    // lane (m, j4) of load g reads sector 16 g + m; the image's rows end at sector S + 15 (the last load's upper lanes stay inside)
    // (sectors past the keyframe's last are its first again: read where they were read before -- those lines are in the L2, the image's
    //  16 repeated sectors per row would come from HBM: 1.5 KB per keyframe, 15 MB per launch)
    auto load_off = [&](int g) -> unsigned int { const int sct = 16 * g + lsec; return a_lane + (unsigned int)((sct < S ? sct : (sct - S < S ? sct - S : 0)) - lsec) * 16u; };
    u32x4 F[NBUF];
    const unsigned char *base_cur = kf_base(0), *base_nxt = kf_base(1);
    int b_cur = 0, b_nxt = 0;
    // ---- stage the scans' ring part (the last STEPS - 1 sectors repeat the first).  Eight 16-byte pieces per thread are requested
    // before the first is stored (one piece at a time, the loop paid a memory round trip per piece: 12.5 k cycles per workgroup, a fifth
    // of the kernel).  Behind the LAST batch's requests and in front of its stores go the wave's first keyframe fragments and first
    // shifts: they do not depend on the image, their round trip to HBM (6-9 k cycles when they followed the barrier) passes under the
    // stores and the barrier, and -- loads return in order -- the scans' pieces, which come out of the L2, are not held up behind them ----
    {
        // One scan per wave and turn (16 scans, 16 waves): the scan -- its slot in the database, its place in the image -- is a scalar, a
        // piece's sector and chunk are shifts of the lane index.  (Dealt out piece by piece over the workgroup's threads, every piece
        // cost a division and a per-lane read of the launch's argument block for its scan's slot: the waves took 5-10 k cycles to ISSUE
        // their requests, and the workgroup met at the barrier 15 k cycles into the kernel -- a quarter of it.)
        constexpr int PPS = 4 / SPK;                           // 16-byte pieces of a scan per sector
        constexpr int NWV = s2_waves<RG, S, W, FUSED>(), NPIECE = ROWS * PPS, BATCH = 8;
        constexpr int NBATCH = (NPIECE + BATCH * kWave - 1) / (BATCH * kWave);
        static_assert(NBATCH <= 2 && NQ <= 2 * NWV, "staging turns");
        // (named variables, straight-line code: as arrays -- behind lambdas or inside macros' loops -- the pieces were placed in scratch)
        uint4 pc0, pc1, pc2, pc3, pc4, pc5, pc6, pc7;
        unsigned int ds0, ds1, ds2, ds3, ds4, ds5, ds6, ds7;
#define S2_STAGE_RQ(BASE, r, PC, DS)                                                                                                            \
        {                                                                                                                                       \
            const int idx0 = (BASE) + (r) * kWave + lane;                                                                                       \
            const int idx = idx0 < NPIECE ? idx0 : NPIECE - 1;                                                                                  \
            const int sx = idx / PPS, ch = idx % PPS;                                                                                           \
            const int sct = sx < S ? sx : sx - S;                                                                                               \
            PC = *reinterpret_cast<const uint4 *>(sbase + (unsigned int)(sct * SB + ch * 16));                                                  \
            DS = dbase + (unsigned int)(sx * 256 + ch * 16);                                                                                    \
        }
#define S2_STAGE_ST(BASE, r, PC, DS) if ((BASE) + (r) * kWave + lane < NPIECE) *reinterpret_cast<uint4 *>(smem2 + DS) = PC;
#define S2_STAGE_REQUEST(BASE) S2_STAGE_RQ(BASE, 0, pc0, ds0) S2_STAGE_RQ(BASE, 1, pc1, ds1) S2_STAGE_RQ(BASE, 2, pc2, ds2) S2_STAGE_RQ(BASE, 3, pc3, ds3) \
                               S2_STAGE_RQ(BASE, 4, pc4, ds4) S2_STAGE_RQ(BASE, 5, pc5, ds5) S2_STAGE_RQ(BASE, 6, pc6, ds6) S2_STAGE_RQ(BASE, 7, pc7, ds7)
#define S2_STAGE_STORE(BASE) S2_STAGE_ST(BASE, 0, pc0, ds0) S2_STAGE_ST(BASE, 1, pc1, ds1) S2_STAGE_ST(BASE, 2, pc2, ds2) S2_STAGE_ST(BASE, 3, pc3, ds3) \
                             S2_STAGE_ST(BASE, 4, pc4, ds4) S2_STAGE_ST(BASE, 5, pc5, ds5) S2_STAGE_ST(BASE, 6, pc6, ds6) S2_STAGE_ST(BASE, 7, pc7, ds7)
        // (a second turn only where the workgroup has fewer waves than the launch has scans: the fused experiment's 12 + 4)
        if (wave + NWV < NQ) {
            const int q2 = wave + NWV;
            const unsigned char *sbase = reinterpret_cast<const unsigned char *>(ab.hdesc + (size_t)ab.q[q2].slot * HS) + part * SCB;
            const unsigned int dbase = (unsigned int)((q2 / SPR) * QUAD + (q2 % SPR) * SCB);
            if constexpr (NBATCH == 2) { S2_STAGE_REQUEST(0) S2_STAGE_STORE(0) }
            S2_STAGE_REQUEST((NBATCH - 1) * BATCH * kWave)
            S2_STAGE_STORE((NBATCH - 1) * BATCH * kWave)
        }
        const int q1 = wave < NQ ? wave : NQ - 1;
        const unsigned char *sbase = reinterpret_cast<const unsigned char *>(ab.hdesc + (size_t)ab.q[q1].slot * HS) + part * SCB;
        const unsigned int dbase = (unsigned int)((q1 / SPR) * QUAD + (q1 % SPR) * SCB);
        if constexpr (NBATCH == 2) { S2_STAGE_REQUEST(0) S2_STAGE_STORE(0) }
        S2_STAGE_REQUEST((NBATCH - 1) * BATCH * kWave)
        S2_STAMP_AT(61);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int NPRE = NBUF - 1 < 4 ? NBUF - 1 : 4;     // (the pieces in flight + every fragment buffer would not fit the registers)
        if (!xwave) {
#pragma unroll
            for (int j = 0; j < NPRE; ++j) F[j] = *reinterpret_cast<const u32x4 *>(base_cur + load_off(j));
            b_cur = first_shift(0); b_nxt = first_shift(1);
        }
        __builtin_amdgcn_sched_barrier(0);
        S2_STAMP_AT(62);
        if (wave < NQ) { S2_STAGE_STORE((NBATCH - 1) * BATCH * kWave) }
        S2_STAMP_AT(63);
        __builtin_amdgcn_sched_barrier(0);
        if (!xwave) {
#pragma unroll
            for (int j = NPRE; j < NBUF - 1; ++j) F[j] = *reinterpret_cast<const u32x4 *>(base_cur + load_off(j));
        }
    }
    __syncthreads();
    S2_STAMP_AT(33);
    if (xwave) {                                       // ---- the extra waves: next batch's alignment, previous batch's finishing ----
        const int xw = wave - WVP;
        const int xg = b * C::XW + xw, xtotal = (int)gridDim.x * C::XW;
#if S2XPRIO > 0
        __builtin_amdgcn_s_setprio(S2XPRIO);           // few instructions, long dependent chains: issue them ahead of the products' waves
#endif
        unsigned char *xs = smem2 + s2_lds<RG, S, W>() + (size_t)xw * s2_xlds<S>();
        if (xa.a_n > 0) sc_align2_role<S, W>(xa.next, xa.halign, xa.a_lo, xa.a_n, xg, xtotal, xs);
        if (xa.f_nb64 > 0) sc_screen2_finish_waves<RG, S, W>(xa.prev, xa.f_nb64, xg, xtotal, xs);
        return;
    }
    if (gw >= fa.u_n) return;
#ifdef S2_NO_PRODUCTS
    return;                                            // experiment: what the extra waves take on their own
#endif
    // Scan sector that meets keyframe sector 0 in pass p: c0 = first shift + W - 1 - 13 p; iteration `it` reads sectors c0 + 4 it .. + 3
    // (mod S; the image repeats STEPS - 1 sectors so that the reads of an iteration never wrap).  The lane's address is ONE of two
    // fixed bases -- before and after its wrap -- plus a compile-time offset of 1 KB per iteration: a compare and a select per
    // iteration and pass instead of the running pointer's five instructions (the vector issue port is what binds this kernel).
    const unsigned char *preB[NB], *postB[NB];            // (pointers: the image's base address is added once per keyframe, not per read)
    int wrapB[NB];                                       // first iteration that reads from the wrapped base
    auto scan_bases = [&](const int first) {
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            int c0 = first + (W - 1) - PS * (AOFF ? 1 : p);   // (shared fragments: the scan where pass 1 wants it)
            c0 = c0 >= S ? c0 - S : c0;
            preB[p] = smem2 + (q_lds + (unsigned int)c0 * 256u);
            postB[p] = preB[p] - S * 256;
            wrapB[p] = (S - c0 + SECT - 1) / SECT;       // smallest it with c0 + 4 it >= S
        }
    };
    auto readB = [&](h8 (&dst)[NB][STEPS], const int it) {
#pragma unroll
        for (int p = 0; p < NB; ++p) {
            const unsigned char *qp = (it < wrapB[p] ? preB[p] : postB[p]) + it * SECT * 256;
#pragma unroll
            for (int u = 0; u < STEPS; ++u) dst[p][u] = *reinterpret_cast<const h8 *>(qp + 256 * SPK * u);
        }
    };
    h8 bfr[2][NB][STEPS];
    scan_bases(b_cur);
    readB(bfr[0], 0);
    for (int k = 0; k < nk; ++k) {
        f4v acc[NPASS][STEPS];
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
#pragma unroll
            for (int u = 0; u < STEPS; ++u) acc[p][u] = f4v{0.f, 0.f, 0.f, 0.f};
        if (k < 6) S2_STAMP_AT(34 + 3 * k);
        // The fragment of iteration it + 1 is formed WHILE iteration it multiplies: its eight row-shift moves have no part in this
        // iteration's products, so they are dealt out between them -- one matrix product, two moves, ... -- and fill the eight cycles in
        // which a matrix product lets the SIMD issue other vector instructions; formed in front of the products they need, the moves
        // were a serial stretch of every iteration.
        auto fragment_of = [&](const int it2) -> u32x4 {
            const int g = it2 >> 2, o = it2 & 3;
            u32x4 fr = F[g % NBUF];
            if (o != 0) {
                const u32x4 f0 = F[g % NBUF], f1 = F[(g + 1) % NBUF];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    // lanes m >= 16 - 4 o: lane m - (16 - 4 o) of the next load; lanes below: lane m + 4 o of this one
                    // (bound_ctrl on the first move: lanes without a source lane take 0 -- no register to initialise; the second keeps them)
                    int t = o == 1 ? __builtin_amdgcn_update_dpp(0, (int)f1[d], 0x11C, 0xf, 0xf, true)        // row_shr:12
                          : o == 2 ? __builtin_amdgcn_update_dpp(0, (int)f1[d], 0x118, 0xf, 0xf, true)        // row_shr:8
                                   : __builtin_amdgcn_update_dpp(0, (int)f1[d], 0x114, 0xf, 0xf, true);       // row_shr:4
                    t = o == 1 ? __builtin_amdgcn_update_dpp(t, (int)f0[d], 0x104, 0xf, 0xf, false)           // row_shl:4
                      : o == 2 ? __builtin_amdgcn_update_dpp(t, (int)f0[d], 0x108, 0xf, 0xf, false)           // row_shl:8
                               : __builtin_amdgcn_update_dpp(t, (int)f0[d], 0x10C, 0xf, 0xf, false);          // row_shl:12
                    fr[d] = (unsigned int)t;
                }
            }
            return fr;
        };
        u32x4 fr_cur = F[0];
        u32x4 fr_h1 = fr_cur, fr_h2 = fr_cur, fr_h3 = fr_cur;   // (shared fragments) the keyframe fragments of the last three iterations
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int g = it >> 2, o = it & 3;
#ifdef S2_STAMP
            if (k == 1 && (wave == 0 || wave == 5) && (b == 0 || b == 37) && lane == 0)
                g_s2_stamps[((b ? 2 : 0) + (wave ? 1 : 0)) * 64 + it] = __builtin_amdgcn_s_memtime();
#endif
            if (o == 0) {                                // the load NBUF - 1 groups ahead: from the next keyframe towards this one's end
                const int j = g + NBUF - 1;                  // (its buffer held load g - 1, whose last fragment was formed two iterations ago)
                // (the keyframe's last load serves the last derived fragment only: sectors up to 4 (NIT - 1) + 15; at S = 120 that is
                //  four of its sixteen sectors -- the other lanes stay out of the request: 0.75 KB per keyframe and ring part less)
                constexpr int kLastLanes = SPK == 2 ? 16 : 4 * (NIT - 1) + 16 - 16 * (NL - 1);
                if (j == NL - 1 && kLastLanes < 16) { if (c16 < kLastLanes) F[j % NBUF] = *reinterpret_cast<const u32x4 *>(base_cur + load_off(j)); }
                else if (j < NL) F[j % NBUF] = *reinterpret_cast<const u32x4 *>(base_cur + load_off(j));
                else if (j - NL < NBUF - 1) F[j % NBUF] = *reinterpret_cast<const u32x4 *>(base_nxt + load_off(j - NL));
            }
            if (it + 1 < NIT) readB(bfr[(it + 1) & 1], it + 1);   // the next iteration's B fragments
#ifndef S2_NO_READ_PIN
            __builtin_amdgcn_sched_barrier(0);           // (pinned ahead of this iteration's products: left alone, the scheduler reuses the
                                                         //  registers the products are reading and issues the reads behind the third of them)
#endif
            u32x4 fr_nxt = fr_cur;
            if (it + 1 < NIT) fr_nxt = fragment_of(it + 1);
            const h8 af = __builtin_bit_cast(h8, fr_cur);
            const int nmm = (AOFF && it < LAG) ? STEPS : STEPS * NPASS;                    // matrix products of this iteration
            if constexpr (AOFF) {
                const h8 ah = __builtin_bit_cast(h8, fr_h3);
#pragma unroll
                for (int u = 0; u < STEPS; ++u) {
                    acc[1][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bfr[it & 1][0][u], acc[1][u], 0, 0, 0);
                    if (it >= LAG) acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bfr[it & 1][0][u], acc[0][u], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < STEPS; ++u)
#pragma unroll
                    for (int p = 0; p < NPASS; ++p)
                        acc[p][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bfr[it & 1][p][u], acc[p][u], 0, 0, 0);
            }
#ifndef S2_NO_INTERLEAVE
            if (it + 1 < NIT && ((it + 1) & 3) != 0) {
                if (nmm == STEPS * NPASS) {
#pragma unroll
                    for (int x = 0; x < STEPS * NPASS; ++x) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // one matrix product
                        __builtin_amdgcn_sched_group_barrier(0x002, 8 / (STEPS * NPASS), 0);   // its share of the eight moves
                    }
                } else {
#pragma unroll
                    for (int x = 0; x < STEPS; ++x) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 8 / STEPS, 0);
                    }
                }
            }
#endif
#ifndef S2_NO_ITER_BARRIER
            __builtin_amdgcn_sched_barrier(0);
#endif
            if constexpr (AOFF) { fr_h3 = fr_h2; fr_h2 = fr_h1; fr_h1 = fr_cur; }
            fr_cur = fr_nxt;
        }
        if constexpr (AOFF) {
            // pass 0's products of the first LAG iterations' scan fragments: with the keyframe's LAST fragments (the sectors wrap), which are
            // at hand only now; the scan fragments are read once more
            h8 bx[LAG][NB][STEPS];
#pragma unroll
            for (int x = 0; x < LAG; ++x) readB(bx[x], x);
#pragma unroll
            for (int x = 0; x < LAG; ++x) {
                const h8 ah = __builtin_bit_cast(h8, x == 0 ? fr_h3 : (x == 1 ? fr_h2 : fr_h1));
#pragma unroll
                for (int u = 0; u < STEPS; ++u) acc[0][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bx[x][0][u], acc[0][u], 0, 0, 0);
            }
        }
        if (k < 6) S2_STAMP_AT(35 + 3 * k);
        // (the loads of the next keyframe that no group of this one reached: their buffers are free now)
#pragma unroll
        for (int j = NGR + NBUF - 1 - NL; j < NBUF - 1; ++j)
            if (j >= 0) F[j % NBUF] = *reinterpret_cast<const u32x4 *>(base_nxt + load_off(j));
        // the next keyframe's first reads of the scans, ahead of this keyframe's tiles (the first buffer is free: the exchanges and
        // stores below used to stand between a keyframe's last product and the next one's first read)
        static_assert(NIT >= 2, "");
        scan_bases(b_nxt);
        if constexpr (((NIT - 1) & 1) == 0) __builtin_amdgcn_sched_barrier(0);   // (the last iteration multiplied out of the first buffer)
        readB(bfr[0], 0);
        // ---- this keyframe's 16 x 16 tiles: lane (q, j) holds rows 4j .. 4j+3 of scan q ----
        {
            const int idx = gw + k * waves_part;
            const int ci = fa.u_lo + idx - sq.base;
            const bool ok = q_live && ci >= 0 && ci < sq.n;
#pragma unroll
            for (int p = 0; p < NPASS; ++p) {
                // row s of the tile = sum over u of row s + SPK u of acc[p][u]: rows 4 j + i + SPK u of this lane while i + SPK u <= 3, of the
                // lane 16 further on (rows 4 (j + 1) ..) beyond; rows 13 .. 15 are not shifts (W - 1 - 13 p - m < 0 there) and take whatever comes
                f4v sum = acc[p][0];
#pragma unroll
                for (int u = 1; u < STEPS; ++u) {
                    constexpr int kMaxSh = SPK * (STEPS - 1);
                    const int sh = SPK * u;
                    float nx[4];
#pragma unroll
                    for (int e2 = 0; e2 < kMaxSh; ++e2)
                        if (e2 < sh) nx[e2] = __int_as_float(__builtin_amdgcn_ds_bpermute(up16, __float_as_int(acc[p][u][e2])));
#pragma unroll
                    for (int i = 0; i < 4; ++i) sum[i] += (i + sh <= 3) ? acc[p][u][(i + sh) & 3] : nx[(i + sh - 4) & 3];
                }
                if (ok && j4 < s2_pass_f4<RG, S, W>(p))
                    *reinterpret_cast<f4v *>(fa.part + (((size_t)c16 * (size_t)ab.pair_stride + (size_t)ci) * NP + part) * (s2_part_f4<RG, S, W>() * 4) + (s2_pass_off<RG, S, W>(p) + j4) * 4) = sum;
            }
        }
        base_cur = base_nxt; base_nxt = kf_base(k + 2);
        b_cur = b_nxt; b_nxt = first_shift(k + 2);
        if (k < 6) S2_STAMP_AT(36 + 3 * k);
    }
    S2_STAMP_AT(60);
}

// the ring parts of every pair meet: d~ = min_t (1 - sim[t] / n_eff[t]), flags, the launch's smallest d~ per scan
// rotq[s] (LDS, built by the caller's threads): the scan's sector mask rotated right by s
template <int S>
__device__ __forceinline__ void build_rotq(const unsigned int *q_kmask, uint4 *rotq, int tid, int nthreads)
{
    constexpr int NW64 = (S + 63) / 64, MW = (NW64 + 1) / 2;
    unsigned long long qm[NW64];
#pragma unroll
    for (int i = 0; i < NW64; ++i) qm[i] = (unsigned long long)q_kmask[2 * i] | ((unsigned long long)(2 * i + 1 < 7 ? q_kmask[2 * i + 1] : 0u) << 32);
    for (int sft = tid; sft < S; sft += nthreads) {
        unsigned long long rr[NW64];
        rotate_mask<S>(qm, sft, rr);
#pragma unroll
        for (int i = 0; i < MW; ++i) {
            const unsigned long long lo = rr[2 * i], hi = 2 * i + 1 < NW64 ? rr[2 * i + 1] : 0ull;
            rotq[sft * MW + i] = make_uint4((unsigned int)lo, (unsigned int)(lo >> 32), (unsigned int)hi, (unsigned int)(hi >> 32));
        }
    }
}

// one pair (scan qi of the batch, position ci of its range): returns its contribution to the launch's smallest d~.
// In two steps: everything the pair reads from memory is REQUESTED first (one batch: the keyframe's mask, the first shift, the partial
// sums, the keyframe's tiled ring key), then -- in the workgroup form behind the barrier that publishes the scan's rotated masks --
// the distances are formed.  Written as one step the compiler waited for the loads two at a time: six memory round trips behind each
// other in a kernel that is nothing but round trips (2.4 waves per SIMD, all resident at once).
template <int RG, int S, int W, bool D2 = true>
struct FinishLoads {
    static constexpr int NPA = S2Cfg<RG, S, W>::NP <= 3 ? S2Cfg<RG, S, W>::NP : 3;   // ring parts whose sums are requested up front (registers: the others follow
                                                                                    //  once these are added up)
    static constexpr int PF4 = s2_part_f4<RG, S, W>();                               // float4 per ring part
    static constexpr int NPF = NPA * PF4;
    static constexpr int MW = (((S + 63) / 64) + 1) / 2;
    static constexpr bool kRingUpFront = NPF * 4 + RG * 4 <= 100;                // (registers: the 80 x 180 grid asks for its ring key later)
    uint4 km[MW]; unsigned int kflag; float kerr; int b_raw; f4v pv[NPF]; const f4v *rest; float4 bk[(D2 && kRingUpFront) ? RG : 1];
};
template <int RG, int S, int W, bool D2 = true>
__device__ __forceinline__ FinishLoads<RG, S, W, D2> sc_screen2_finish_request(const Screen2Args &fa, const ScreenArgs &a, const int qi, const int ci)
{
    using L = FinishLoads<RG, S, W, D2>;
    using C = S2Cfg<RG, S, W>;
    const ScreenBatchArgs &ab = fa.prod;
    L l;
    const unsigned int *kp = a.kmask + (size_t)(a.slot_base + ci) * 8;
#pragma unroll
    for (int i = 0; i < L::MW; ++i) l.km[i] = *reinterpret_cast<const uint4 *>(kp + 4 * i);
    {
        const uint2 ef = *reinterpret_cast<const uint2 *>(kp + 6);              // word 6: the keyframe's summed rounding-error norms E; word 7: the flag
        l.kerr = __uint_as_float(ef.x); l.kflag = ef.y;
    }
    l.b_raw = a.starts[ci];
    const f4v *pp = reinterpret_cast<const f4v *>(fa.part + ((size_t)qi * (size_t)ab.pair_stride + (size_t)ci) * (size_t)s2_part_floats<RG, S, W>());
#pragma unroll
    for (int i = 0; i < L::NPF; ++i) l.pv[i] = pp[i];
    l.rest = pp + L::NPF;
    if constexpr (D2 && L::kRingUpFront) {
        const int slot = a.slot_base + ci;
#pragma unroll
        for (int r = 0; r < RG; ++r) l.bk[r] = a.rkey4[(size_t)r * a.rk_cap + slot];
    }
    return l;
}
// MASKS = false: an instance for launches that ask for no shift masks (the 64 x 120 stream's tail launch): no per-shift interval ends
// are kept -- thirteen registers that the tail launch, built for four waves per SIMD beside the alignment, does not have.
// D2 = false: the ring-key metric of the pair is left to the exact pass of the range (sc_distance_survivors_kernel forms it in
// front of its top-k, on the side stream): sixteen float4 loads and sixty-four registers per pair that the stream's tail launch
// neither waits for nor holds
template <int RG, int S, int W, bool MASKS = true, bool D2 = true>
__device__ __forceinline__ float sc_screen2_finish_compute(const ScreenArgs &a, const int ci, const uint4 *rotq, const bool q_bad, const float q_err, FinishLoads<RG, S, W, D2> &l, float &pair_eps)
{
    using C = S2Cfg<RG, S, W>;
    using L = FinishLoads<RG, S, W, D2>;
    constexpr int NP = C::NP, NPASS = C::NPASS, MW = L::MW, PS = C::PS;
    const float kInf = __int_as_float(0x7f800000);
    if (MW == 2) { l.km[MW - 1].z = 0u; l.km[MW - 1].w = 0u; }               // words 6 and 7 are E and the flag, not sector bits
    const bool b_open = l.b_raw < 0;                                         // kAlignUndecided: scored by the exact pass
    const int b0 = b_open ? 0 : l.b_raw;
    float dmin = kInf;
    float dsh[W];                                                            // the screened distance of every shift (+inf: no effective sector)
    // ... and how far it can be from the reference's distance of that shift: (E_q + E_k)(1 + 1e-3) / n_eff for the fp16 rounding of the
    // two descriptors' unit columns on their recorded error norms (make_sc.hip), screen2_acc_eps for the accumulation and the fp32
    // scaling / quotient / difference; never more than the worst-case kScreenEps
    const bool want_mask = MASKS && a.out_smask != nullptr;
    const float e_pair = (q_err + l.kerr) * 1.002f;
    const bool e_ok = e_pair >= 0.0f && e_pair < 1.0f;                       // (NaN / absurd values: the worst-case bound)
    float hi_min = kInf;                                                     // the smallest upper bound of a shift's exact distance
    int ne_min = 0x7fffffff;                                                 // fewest effective sectors over the pair's shifts
    float dlo[MASKS ? W : 1];
#pragma unroll
    for (int t = 0; t < W; ++t) { dsh[t] = kInf; if (MASKS) dlo[MASKS ? t : 0] = kInf; }
    // the ring parts' sums of a tile row, in the order of the parts; the tiles' rows that are shifts: 13 of pass 0, W - 13 of pass 1
    constexpr int NPA = L::NPA;
    f4v psum[NPASS][4];
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            if (!s2_group_used<RG, S, W>(p, g4)) continue;
            f4v sm = l.pv[s2_pass_off<RG, S, W>(p) + g4];                    // part 0
#pragma unroll
            for (int h = 1; h < NPA; ++h) sm += l.pv[h * L::PF4 + s2_pass_off<RG, S, W>(p) + g4];
            psum[p][g4] = sm;
        }
    if constexpr (NP > NPA) {
        f4v rv[NP - NPA][NPASS][4];
#pragma unroll
        for (int h = NPA; h < NP; ++h)
#pragma unroll
            for (int p = 0; p < NPASS; ++p)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    if (s2_group_used<RG, S, W>(p, g4)) rv[h - NPA][p][g4] = l.rest[(h - NPA) * L::PF4 + s2_pass_off<RG, S, W>(p) + g4];
#pragma unroll
        for (int h = NPA; h < NP; ++h)
#pragma unroll
            for (int p = 0; p < NPASS; ++p)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    if (s2_group_used<RG, S, W>(p, g4)) psum[p][g4] += rv[h - NPA][p][g4];
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            if (!s2_group_used<RG, S, W>(p, g4)) continue;
            const f4v sm = psum[p][g4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 4 * g4 + r;                                    // row m of pass p = shift W - 1 - 13 p - m
                const int t = W - 1 - PS * p - m;
                if (m >= kS2PassRows || t < 0 || (p > 0 && t >= W - 1 - PS * (p - 1) - (kS2PassRows - 1))) continue;   // (a shift the pass in front holds)
                int ri = b0 + t; ri = ri >= S ? ri - S : ri;
                int ne = 0;
#pragma unroll
                for (int i = 0; i < MW; ++i) {
                    const uint4 rq = rotq[ri * MW + i];
                    ne += __popc(rq.x & l.km[i].x) + __popc(rq.y & l.km[i].y) + __popc(rq.z & l.km[i].z) + __popc(rq.w & l.km[i].w);
                }
                const float d = 1.0f - sm[r] * __builtin_amdgcn_rcpf((float)ne);      // (v_rcp_f32, 1 ulp: a screened value; the bound's 2e-6 covers it -- the IEEE quotient is ten instructions, thirteen times per pair)
                if (ne > 0 && d < dmin) dmin = d;
                if (ne > 0) ne_min = ne < ne_min ? ne : ne_min;
                dsh[t] = ne > 0 ? d : kInf;
                if (want_mask && ne > 0) {                                   // (wave uniform: the stream form of 64 x 120 asks for no masks)
                    const float et = e_ok ? fminf(e_pair / (float)ne + screen2_acc_eps<S>(), kScreenEps) : kScreenEps;
                    dlo[MASKS ? t : 0] = d - et; hi_min = fminf(hi_min, d + et);
                }
            }
        }
    }
    const bool exact_only = q_bad || l.kflag != 0 || b_open || !(dmin == dmin);
    a.out_approx[ci] = exact_only ? __int_as_float(0xff800000) : dmin;
    // this pair's bound on |d~ - d| (every shift's is at most this): into the launch's largest (kTminEpsOffset)
    pair_eps = (exact_only || ne_min == 0x7fffffff) ? 0.0f : (e_ok ? fminf(e_pair / (float)ne_min + screen2_acc_eps<S>(), kScreenEps) : kScreenEps);
    if (MASKS && a.out_smask) {
        // a shift can hold (or tie for) the pair's exact minimum only if the lower end of its interval does not lie above the
        // smallest upper end (NaN sums compare false everywhere: such a pair is exact_only); an undecided alignment has no first
        // shift: mask 0
        unsigned int m = 0u;
#pragma unroll
        for (int t = 0; t < W; ++t) m |= ((exact_only || dlo[MASKS ? t : 0] <= hi_min) ? 1u : 0u) << t;
        a.out_smask[ci] = b_open ? 0u : m;
    }
    // nanoflann's metric (nanoflann.hpp:383-408) for the ring-key top-k: four dimensions per step, fp32, groups accumulated in
    // order -- the arithmetic of sc_align_role, here with consecutive threads on consecutive slots of the tiled key table
    if constexpr (D2) {
        const int slot = a.slot_base + ci;
        float result = 0.0f;
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            float4 bk;
            if constexpr (L::kRingUpFront) bk = l.bk[r]; else bk = a.rkey4[(size_t)r * a.rk_cap + slot];
            const float4 qk = *reinterpret_cast<const float4 *>(a.q_rkey + 4 * r);
            const float d0 = qk.x - bk.x, d1 = qk.y - bk.y, d2 = qk.z - bk.z, d3 = qk.w - bk.w;
            const float grp = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            result += grp;
        }
        a.out_d2[ci] = result;
    }
    return exact_only ? kInf : dmin;
}
template <int RG, int S, int W>
__device__ __forceinline__ float sc_screen2_finish_pair(const Screen2Args &fa, const ScreenArgs &a, const int qi, const int ci, const uint4 *rotq, const bool q_bad, float &pair_eps)
{
    FinishLoads<RG, S, W> l = sc_screen2_finish_request<RG, S, W>(fa, a, qi, ci);
    return sc_screen2_finish_compute<RG, S, W>(a, ci, rotq, q_bad, __uint_as_float(a.q_kmask[6]), l, pair_eps);
}

template <int RG, int S, int W, bool MASKS = true, bool D2 = true>
__device__ __forceinline__ void sc_screen2_finish_body(const Screen2Args &fa, const int qi, const int chunk)
{
    constexpr int NW64 = (S + 63) / 64, MW = (NW64 + 1) / 2;
    const ScreenBatchArgs &ab = fa.prod;
    const ScreenArgs a = screen_args_of(ab, qi);
    __shared__ uint4 rotq[S * MW];
    __shared__ float wmin[4], wmax[4];
    const int ci = (int)(chunk * blockDim.x + threadIdx.x);
    const bool live = ci < a.n;
    FinishLoads<RG, S, W, D2> ld = sc_screen2_finish_request<RG, S, W, D2>(fa, a, qi, live ? ci : 0);   // (a.n >= 1 where a workgroup was launched for the scan)
    build_rotq<S>(a.q_kmask, rotq, (int)threadIdx.x, (int)blockDim.x);
    const bool q_bad = a.q_kmask[7] != 0;
    __syncthreads();
    const float kInf = __int_as_float(0x7f800000);
    float contrib = kInf, peps = 0.0f;
    if (live) contrib = sc_screen2_finish_compute<RG, S, W, MASKS, D2>(a, ci, rotq, q_bad, __uint_as_float(a.q_kmask[6]), ld, peps);
    // (wave minimum / maximum by DPP row exchanges: twelve ds_bpermute round trips per thread otherwise)
    contrib = -wave_max_f32_dpp(-contrib);
    peps = wave_max_f32_dpp(peps);
    if ((threadIdx.x & 63) == 0) { wmin[threadIdx.x >> 6] = contrib; wmax[threadIdx.x >> 6] = peps; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fminf(fminf(wmin[0], wmin[1]), fminf(wmin[2], wmin[3]));
        if (m < kInf) atomicMin(a.t_min, float_to_ordered_u(m));
        const float pe = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (pe > 0.0f) atomicMax(a.t_min + kTminEpsOffset, __float_as_uint(pe));      // (positive floats order like their bits)
    }
}

// ... by single waves (the extra waves of the products' launch): wave unit0, unit0 + stride, ... of the (scan, block of 64 pairs)
// units of the batch; rotq in the wave's own LDS tile
template <int RG, int S, int W>
__device__ __forceinline__ void sc_screen2_finish_waves(const Screen2Args &fa, const int nb64, const int unit0, const int unit_stride, unsigned char *lds_wave)
{
    const ScreenBatchArgs &ab = fa.prod;
    const int lane = threadIdx.x & (kWave - 1);
    uint4 *rotq = reinterpret_cast<uint4 *>(lds_wave);
    const float kInf = __int_as_float(0x7f800000);
    for (int u = unit0; u < ab.nq * nb64; u += unit_stride) {
        const int qi = u / nb64, blk = u - qi * nb64;
        const ScreenArgs a = screen_args_of(ab, qi);
        if (blk * kWave >= a.n) continue;                                        // (wave uniform)
        wave_fence();
        build_rotq<S>(a.q_kmask, rotq, lane, kWave);
        const bool q_bad = a.q_kmask[7] != 0;
        wave_fence();
        const int ci = blk * kWave + lane;
        float contrib = kInf, peps = 0.0f;
        if (ci < a.n) contrib = sc_screen2_finish_pair<RG, S, W>(fa, a, qi, ci, rotq, q_bad, peps);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { contrib = fminf(contrib, __shfl_xor(contrib, off, kWave)); peps = fmaxf(peps, __shfl_xor(peps, off, kWave)); }
        if (lane == 0 && contrib < kInf) atomicMin(a.t_min, float_to_ordered_u(contrib));
        if (lane == 0 && peps > 0.0f) atomicMax(a.t_min + kTminEpsOffset, __float_as_uint(peps));
    }
}

template <int RG, int S, int W>
__global__ __launch_bounds__(256) void sc_screen2_finish_kernel(Screen2Args fa)
{
    sc_screen2_finish_body<RG, S, W>(fa, (int)blockIdx.y, (int)blockIdx.x);
}

// The tail of a launch group in ONE launch: the alignment of the NEXT batch (workgroups [0, align_blocks): latency bound, a third
// of the issue slots) beside the finishing of THIS batch (the workgroups behind them: bandwidth bound, short).  On their own
// they take 29.7 + 15.5 us per 16 scans of the 64 x 120 grid.
template <int RG, int S, int W>
__global__ __launch_bounds__(kScreenWaves * kWave, 2) void sc_screen2_tail_kernel(Screen2Args fa, ScreenBatchArgs nb, int align_blocks, int chunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tail[];
    const int b = (int)blockIdx.x;
    if (b < align_blocks) { sc_align_role<RG, S, W>(nb, b, smem_tail); return; }
    const int fb = b - align_blocks;
    sc_screen2_finish_body<RG, S, W>(fa, fb / chunks, fb - (fb / chunks) * chunks);
}

// ... with the alignment in its second form (one keyframe against the next batch's scans per wave)
template <int RG, int S, int W, bool MASKS = true, bool D2 = true>
__global__ __launch_bounds__(kScreenWaves * kWave, Align2Cfg<S>::OCC) void sc_screen2_tail2_kernel(Screen2Args fa, ScreenBatchArgs nb, const unsigned char *halign, int u_lo, int u_n,
                                                                                     int align_blocks, int chunks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tail2[];
#ifdef SCL_TAIL_FINISH_FIRST
    const int fblocks = (int)gridDim.x - align_blocks;
    const int b = (int)blockIdx.x < fblocks ? (int)blockIdx.x + align_blocks : (int)blockIdx.x - fblocks;
#else
    const int b = (int)blockIdx.x;
#endif
    if (b < align_blocks) {
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        sc_align2_role<S, W>(nb, halign, u_lo, u_n, b * kScreenWaves + wave, align_blocks * kScreenWaves, smem_tail2 + (size_t)wave * Align2Cfg<S>::LDS_WAVE);
        return;
    }
    const int fb = b - align_blocks;
    sc_screen2_finish_body<RG, S, W, MASKS, D2>(fa, fb / chunks, fb - (fb / chunks) * chunks);
}

static bool screen_second_form()
{   // SCL_SCREEN_FORM=1 keeps the products' first form (keyframe rows per pair through the L2, alignment riding in the launch)
    static const int form = [] { const char *e = getenv("SCL_SCREEN_FORM"); return e ? atoi(e) : 2; }();
    return form != 1;
}
static int screen_v2_min_env()
{   // SCL_SCREEN_V2_MIN: the smallest batch the second form scores (0 / unset: the grid's own; tests: 1 = always the second form)
    static const int v = [] { const char *e = getenv("SCL_SCREEN_V2_MIN"); return e ? atoi(e) : 0; }();
    return v;
}
int sc_screen_max_batch(const DbView &db, int SR)
{
    return (sc_screen_is_wide(db, SR) && screen_second_form()) ? S2Cfg<20, 180, 19>::NQ : kMaxScreenBatch;
}
size_t sc_screen_scratch_floats(const DbView &db, int SR)
{
    return sc_screen_is_wide(db, SR) ? (size_t)S2Cfg<20, 180, 19>::NQ * s2_part_floats<20, 180, 19>() : (size_t)S2Cfg<16, 120, 13>::NQ * s2_part_floats<16, 120, 13>();
}

// argument block of one batch; returns the largest range or -1
static int fill_screen_args(const DbView &db, const ScreenBatch &sb, int align_filter, ScreenBatchArgs *ab)
{
    ab->vkey = db.vkey; ab->rkey = db.rkey; ab->hdesc = db.hdesc; ab->kmask = db.kmask; ab->rkey4 = db.rkey4; ab->hkey = db.hkey; ab->hkw = hkey_row_halfs(db.S);
    ab->starts = sb.starts; ab->approx = sb.approx; ab->ring_d2 = sb.ring_d2; ab->t_min = sb.t_min; ab->fallbacks = sb.align_fallbacks; ab->smask = sb.smask;
    ab->pair_stride = (unsigned long long)sb.pair_stride;
    ab->S = db.S; ab->R4 = 4 * db.RG; ab->hstride = db.hstride; ab->rk_cap = db.cap; ab->align_filter = align_filter;
    ab->nq = sb.nq;
    int nmax = 0;
    for (int i = 0; i < sb.nq; ++i) {
        if (sb.n[i] <= 0) return -1;
        ab->q[i] = ScreenQuery{sb.slot[i], sb.base[i], sb.n[i], sb.buf[i]};
        nmax = sb.n[i] > nmax ? sb.n[i] : nmax;
    }
    for (int i = sb.nq; i < kMaxScreenBatch; ++i) ab->q[i] = ab->q[0];
    return nmax;
}

// one grid: RG ring groups, S sectors, W shifts; OCC0 / OCC1 = waves per SIMD of the default kernel / of variant 1
template <int RG, int S, int W, int OCC0, int OCC1>
static hipError_t launch_screen_grid(const DbView &db, const ScreenBatch &sb, int align_filter, int num_cu, hipStream_t stream, int phases,
                                     const ScreenBatch *next, const ScreenBatch *prev)
{
    constexpr int RGH = hdesc_rgh(RG);
    ScreenBatchArgs ab{};
    const int nmax = fill_screen_args(db, sb, align_filter, &ab);
    if (nmax < 0) return hipErrorInvalidValue;
    const int ngroups = (nmax + kGroup - 1) / kGroup;
    if (db.hstride != hdesc_stride(RG, S) || !sb.starts) return hipErrorInvalidValue;
    static std::atomic<bool> attr_set_dev[64];                 // per grid (template instance) and device
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<bool> &attr_set = attr_set_dev[dev_ & 63];
    // variants (SCL_SCREEN_VARIANT): 0 = 15 k-steps in flight; 1 = the other register budget; 2 = 5 k-steps in flight
    static const int variant = scl_lab_int("SCL_SCREEN_VARIANT", 0);
    void (*kern)(ScreenFusedArgs) = sc_screen_kernel<RG, S, W, 15, OCC0>;
#ifdef SCL_DIAGNOSTICS
    if (variant == 1) kern = sc_screen_kernel<RG, S, W, 15, OCC1>;
    if (variant == 2) kern = sc_screen_kernel<RG, S, W, 5, OCC0>;
    static const int probe = scl_lab_int("SCL_SCREEN_PROBE", 0);
    if (probe == 2) kern = sc_screen_kernel<RG, S, W, 15, OCC0, 2>;
#else
    const int probe = 0;
#endif
    constexpr int MTA = (S + 15) / 16, MT = (W + 15) / 16, QSX = S + 16 * MT, MW = ((S + 63) / 64 + 1) / 2;
    const size_t lds0 = (size_t)S * 8 + (size_t)(32 * MTA) * 4 + (size_t)2 * (16 * MTA + hkey_halfs(S) + 8) * 2 +
                        (size_t)kScreenWaves * ((2 * S + 2) * 8);
    const size_t lds1 = (size_t)QSX * (RGH * 8 + 32) + (size_t)S * MW * 16 + (size_t)kScreenWaves * kGroup * kTileStride +
                        (size_t)(2 * kScreenWaves * MT) * kWave * 16;
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)sc_screen_kernel<RG, S, W, 15, OCC0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)sc_align_kernel<RG, S, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
#ifdef SCL_DIAGNOSTICS
        (void)hipFuncSetAttribute((const void *)sc_screen_kernel<RG, S, W, 15, OCC1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute((const void *)sc_screen_kernel<RG, S, W, 5, OCC0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute((const void *)sc_screen_kernel<RG, S, W, 15, OCC0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
#endif
        attr_set.store(true, std::memory_order_release);
    }
    // workgroups of the alignment per query: one wave per 16 keyframes, at most two workgroups per CU over all queries of the
    // batch (measured beside the products: 128 per query for four queries on 256 CUs is the best point; 136 and more cost 5 %)
    auto align_blocks = [&](int groups, int nq, bool beside_products) {
        int b = (groups + kScreenWaves - 1) / kScreenWaves;
        static const int cap_env = scl_lab_int("SCL_ALIGN_WGS", 0);
        int cap = cap_env > 0 ? cap_env : (beside_products ? 2 * num_cu / (nq > 0 ? nq : 1) : num_cu);   // on its own: one wave per group
        if (cap < 1) cap = 1;
        return b > cap ? cap : b;
    };
    // second form of the products (64 x 120: the keyframe as the shared operand, the launch's scans as columns): needs the
    // partial-sum scratch; SCL_SCREEN_FORM=1 keeps the first form.  Its finishing kernel also forms the ring-key metric.
    // Which form scores a batch: the second form's cost does not depend on the number of scans (54 us for one or sixteen on
    // 64 x 120, 175 us on 80 x 180), the first form's grows with it (13 / 107 us per scan): batches of up to three scans
    // (one on 80 x 180) -- blocking single-scan calls, short remainders -- take the first form.  A batch's alignment is
    // launched by the batch in front of it and leaves the ring-key metric to the second form's finishing kernel, so the
    // decision is made per batch from its own size.
    const int v2_min = screen_v2_min_env() > 0 ? screen_v2_min_env() : (S <= 128 ? 4 : 2);
    auto second_form_for = [&](const ScreenBatch &b) {
        return screen_second_form() && b.part && probe == 0 && variant == 0 && b.nq >= v2_min && b.nq <= S2Cfg<RG, S, W>::NQ;
    };
    const bool use_v2 = second_form_for(sb);
    const bool next_v2 = next && second_form_for(*next);
    ab.skip_d2 = use_v2 ? 1 : 0;
    // SCL_ALIGN_FORM=1 keeps the alignment's first form (one scan against 16 keyframes per tile, inline fp32 / fp64 fallbacks)
    static const int align_form_env = scl_lab_int("SCL_ALIGN_FORM", 2);
    const bool align2 = align_form_env != 1 && db.halign != nullptr && halign_bytes(S) > 0;
    constexpr size_t lds_a2 = (size_t)kScreenWaves * Align2Cfg<S>::LDS_WAVE;
    auto union_of = [](const ScreenBatch &b, int *lo_out, int *n_out) {
        int lo = b.base[0], hi = b.base[0] + b.n[0];
        for (int i = 1; i < b.nq; ++i) { lo = b.base[i] < lo ? b.base[i] : lo; hi = b.base[i] + b.n[i] > hi ? b.base[i] + b.n[i] : hi; }
        *lo_out = lo; *n_out = hi - lo;
    };
    // workgroups of the alignment's second form: a wave per keyframe, several keyframes per wave (the next one's image in flight)
    auto align2_blocks = [&](int u_n) {
        static const int env = scl_lab_int("SCL_ALIGN2_WGS", 0);
        int b = (u_n + kScreenWaves * 3 - 1) / (kScreenWaves * 3);
        const int cap = env > 0 ? env : 2 * num_cu;
        b = b > cap ? cap : b;
        return b < 1 ? 1 : b;
    };
    // SCL_SCREEN_FUSE=1 (experiment; measured slower): the next batch's alignment and the previous batch's
    // finishing in extra waves inside the products' launch instead of a launch of their own behind it (sc_screen2_tail2_kernel)
    static const int fuse_env = scl_lab_int("SCL_SCREEN_FUSE", 0);
    const bool fuse = fuse_env != 0 && align2;
    if (phases & kScreenFinish) {                                // this batch's finishing alone (the end of a sequence of deferred ones)
        if (!use_v2) return hipErrorInvalidValue;
        Screen2Args f2{};
        f2.prod = ab; f2.part = sb.part;
        hipLaunchKernelGGL((sc_screen2_finish_kernel<RG, S, W>), dim3((nmax + 255) / 256, sb.nq), dim3(256), 0, stream, f2);
        return hipGetLastError();
    }
    // a batch of the first form that has to align for itself and has a products launch coming (a blocking call of one to three
    // scans): the products' workgroups align their own groups (sc_screen_kernel), no launch in front of them
    const bool self_align = (phases & kScreenAlign) && (phases & kScreenProducts) && !use_v2 && probe == 0 && !scl_lab_int("SCL_SELF_ALIGN_OFF", 0);
    // the alignment of this batch, on its own (the first launch of a sequence, or a caller that has only one)
    if ((phases & kScreenAlign) && probe != 3 && !self_align) {
        if (align2 && use_v2) {                                  // (the products' first form takes the ring-key metric from the alignment's first form)
            int ulo, un;
            union_of(sb, &ulo, &un);
            hipLaunchKernelGGL((sc_align2_kernel<RG, S, W>), dim3(align2_blocks(un)), dim3(kScreenWaves * kWave), lds_a2, stream, ab, db.halign, ulo, un);
        } else {
            ab.nb = align_blocks(ngroups, sb.nq, false);
            hipLaunchKernelGGL((sc_align_kernel<RG, S, W>), dim3(ab.nb * sb.nq), dim3(kScreenWaves * kWave), lds0, stream, ab);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (probe == 4 || !(phases & kScreenProducts)) return hipSuccess;
    {
        if (use_v2) {
            static std::atomic<bool> attr2_dev[64];
            std::atomic<bool> &attr2 = attr2_dev[dev_ & 63];
            if (!attr2.load(std::memory_order_acquire)) {
                hipError_t e = hipFuncSetAttribute((const void *)sc_screen2_kernel<RG, S, W, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(s2_lds_total<RG, S, W, false>()));
#ifdef SCL_DIAGNOSTICS
                if (e == hipSuccess) e = hipFuncSetAttribute((const void *)sc_screen2_kernel<RG, S, W, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(s2_lds_total<RG, S, W, true>()));
#endif
                if (e != hipSuccess) return e;
                attr2.store(true, std::memory_order_release);
            }
            Screen2Args f2{};
            f2.prod = ab; f2.part = sb.part;
            int lo = sb.base[0], hi = sb.base[0] + sb.n[0];
            for (int i = 1; i < sb.nq; ++i) { lo = sb.base[i] < lo ? sb.base[i] : lo; hi = sb.base[i] + sb.n[i] > hi ? sb.base[i] + sb.n[i] : hi; }
            f2.u_lo = lo; f2.u_n = hi - lo;
            constexpr int WGU = 8 * S2Cfg<RG, S, W>::NP;                           // the ring parts of an index on one XCD
            int nwg = (num_cu / WGU) * WGU;
            if (nwg < WGU) nwg = WGU;
            f2.nwg = nwg;
            // SCL_ALIGN_SIDE: where the next batch's alignment runs -- 0 (default): in line on the main stream, behind the finishing
            // kernel; 1: on the low-priority side stream from the end of this batch's products, beside the finishing kernel; 2: from
            // the start of the products.  Measured: 1.45 G pairs/s in line, 0.51 G (1) and 0.67 G (2) -- the two event hops between
            // the queues per launch cost more than the alignment itself.
            static const int side_env = scl_lab_int("SCL_ALIGN_SIDE", 0);
            const int side = (next && sb.side) ? side_env : 0;
            hipError_t e = hipSuccess;
            auto launch_align = [&](hipStream_t as) -> hipError_t {
                if (next->nq < 1 || next->nq > kMaxScreenBatch) return hipErrorInvalidValue;
                ScreenBatchArgs nb{};
                const int nmax2 = fill_screen_args(db, *next, align_filter, &nb);
                if (nmax2 < 0) return hipErrorInvalidValue;
                nb.skip_d2 = next_v2 ? 1 : 0;
                if (align2 && next_v2) {
                    int ulo, un;
                    union_of(*next, &ulo, &un);
                    hipLaunchKernelGGL((sc_align2_kernel<RG, S, W>), dim3(align2_blocks(un)), dim3(kScreenWaves * kWave), lds_a2, as, nb, db.halign, ulo, un);
                    return hipGetLastError();
                }
                const int ng2 = (nmax2 + kGroup - 1) / kGroup;
                // persistent workgroups: three per CU (167 registers: three waves per SIMD; 128 spill and double the time) over the
                // whole batch, every wave walks several groups with the next group's keys in flight
                int per_q = 3 * num_cu / next->nq;                                   // (two to nine per CU measure the same 29 us)
                per_q = per_q < 1 ? 1 : per_q;
                nb.nb = (ng2 + kScreenWaves - 1) / kScreenWaves;
                nb.nb = nb.nb > per_q ? per_q : nb.nb;
                hipLaunchKernelGGL((sc_align_kernel<RG, S, W>), dim3(nb.nb * next->nq), dim3(kScreenWaves * kWave), lds0, as, nb);
                return hipGetLastError();
            };
            auto fork = [&]() -> hipError_t {
                hipError_t r = hipEventRecord(sb.ev_fork, stream);
                if (r == hipSuccess) r = hipStreamWaitEvent(sb.side, sb.ev_fork, 0);
                if (r == hipSuccess) r = launch_align(sb.side);
                if (r == hipSuccess) r = hipEventRecord(sb.ev_join, sb.side);
                return r;
            };
            if (side == 2 && (e = fork()) != hipSuccess) return e;

            // what the extra waves of this launch carry: the next batch's alignment, the previous batch's finishing
            Screen2Extra xa{};
            static const int parts_env = scl_lab_int("SCL_FUSE_PARTS", 3);   // experiments: 1 = only the alignment rides, 2 = only the finishing
            const bool ride_align = fuse && side == 0 && next && next_v2 && (parts_env & 1);
            if (ride_align) {
                if (next->nq < 1 || next->nq > kMaxScreenBatch) return hipErrorInvalidValue;
                if (fill_screen_args(db, *next, align_filter, &xa.next) < 0) return hipErrorInvalidValue;
                xa.next.skip_d2 = 1;
                xa.halign = db.halign;
                union_of(*next, &xa.a_lo, &xa.a_n);
            }
            if (prev) {
                if (!fuse || prev->nq < 1 || prev->nq > kMaxScreenBatch || !prev->part) return hipErrorInvalidValue;
                const int pmax = fill_screen_args(db, *prev, align_filter, &xa.prev.prod);
                if (pmax < 0) return hipErrorInvalidValue;
                xa.prev.prod.skip_d2 = 1;
                xa.prev.part = prev->part;
                xa.f_nb64 = (pmax + kWave - 1) / kWave;
            }
#ifdef SCL_DIAGNOSTICS
            if (fuse) hipLaunchKernelGGL((sc_screen2_kernel<RG, S, W, true>), dim3(nwg), dim3(s2_waves<RG, S, W, true>() * kWave), (s2_lds_total<RG, S, W, true>()), stream, f2, xa);
            else
#endif
            hipLaunchKernelGGL((sc_screen2_kernel<RG, S, W, false>), dim3(nwg), dim3(s2_waves<RG, S, W, false>() * kWave), (s2_lds_total<RG, S, W, false>()), stream, f2, xa);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if (fuse && side == 0) {
                // this batch's finishing: deferred to the next launch's extra waves (the caller passes this batch as its `prev`, or
                // ends the sequence with phases = kScreenFinish), or now
                if (!(phases & kScreenDeferFinish)) {
                    hipLaunchKernelGGL((sc_screen2_finish_kernel<RG, S, W>), dim3((nmax + 255) / 256, sb.nq), dim3(256), 0, stream, f2);
                    if ((e = hipGetLastError()) != hipSuccess) return e;
                }
                if (next && !ride_align && (e = launch_align(stream)) != hipSuccess) return e;   // (a batch the first form will score)
                return hipSuccess;
            }
            if (prev || (phases & kScreenDeferFinish)) return hipErrorInvalidValue;
            if (side == 1 && (e = fork()) != hipSuccess) return e;
            static const int tail_env = scl_lab_int("SCL_SCREEN_TAIL", 1);   // 0: finish and alignment as two launches
            if (next && side == 0 && tail_env) {
                if (next->nq < 1 || next->nq > kMaxScreenBatch) return hipErrorInvalidValue;
                ScreenBatchArgs nb{};
                const int nmax2 = fill_screen_args(db, *next, align_filter, &nb);
                if (nmax2 < 0) return hipErrorInvalidValue;
                nb.skip_d2 = next_v2 ? 1 : 0;
                const int chunks = (nmax + 255) / 256;
                if (align2 && next_v2) {
                    // (the first form of the products forms the ring-key metric in its own alignment role: only batches that the
                    //  second form will score take the alignment's second form here)
                    int ulo, un;
                    union_of(*next, &ulo, &un);
                    const int ablocks = align2_blocks(un);
                    const dim3 tgrid(ablocks + chunks * sb.nq), tblock(kScreenWaves * kWave);
                    const bool d2 = !sb.no_ring_metric;                       // (left to the exact pass of the range: kernels.hpp)
                    if (sb.smask && d2) hipLaunchKernelGGL((sc_screen2_tail2_kernel<RG, S, W, true, true>), tgrid, tblock, lds_a2, stream, f2, nb, db.halign, ulo, un, ablocks, chunks);
                    else if (sb.smask) hipLaunchKernelGGL((sc_screen2_tail2_kernel<RG, S, W, true, false>), tgrid, tblock, lds_a2, stream, f2, nb, db.halign, ulo, un, ablocks, chunks);
                    else if (d2) hipLaunchKernelGGL((sc_screen2_tail2_kernel<RG, S, W, false, true>), tgrid, tblock, lds_a2, stream, f2, nb, db.halign, ulo, un, ablocks, chunks);
                    else hipLaunchKernelGGL((sc_screen2_tail2_kernel<RG, S, W, false, false>), tgrid, tblock, lds_a2, stream, f2, nb, db.halign, ulo, un, ablocks, chunks);
                } else {
                    const int ng2 = (nmax2 + kGroup - 1) / kGroup;
                    int per_q = 3 * num_cu / next->nq;
                    per_q = per_q < 1 ? 1 : per_q;
                    nb.nb = (ng2 + kScreenWaves - 1) / kScreenWaves;
                    nb.nb = nb.nb > per_q ? per_q : nb.nb;
                    const int ablocks = nb.nb * next->nq;
                    hipLaunchKernelGGL((sc_screen2_tail_kernel<RG, S, W>), dim3(ablocks + chunks * sb.nq), dim3(kScreenWaves * kWave), lds0, stream, f2, nb, ablocks, chunks);
                }
                if ((e = hipGetLastError()) != hipSuccess) return e;
            } else {
                hipLaunchKernelGGL((sc_screen2_finish_kernel<RG, S, W>), dim3((nmax + 255) / 256, sb.nq), dim3(256), 0, stream, f2);
                if ((e = hipGetLastError()) != hipSuccess) return e;
                if (next && side == 0 && (e = launch_align(stream)) != hipSuccess) return e;
            }
            if (side != 0 && (e = hipStreamWaitEvent(stream, sb.ev_join, 0)) != hipSuccess) return e;
            return hipSuccess;
        }
    }
    // the products of this batch + the alignment of the next one
    ScreenFusedArgs fa{};
    fa.prod = ab;
    int blocks = num_cu * 2;
    if (blocks > ngroups) blocks = ngroups;
    if (blocks > kScreenMaxBlocks) blocks = kScreenMaxBlocks;
    fa.prod.nb = blocks;
    fa.prod.self_align = self_align ? 1 : 0;
    int extra = 0;
    if (next && probe != 3) {
        if (next->nq < 1 || next->nq > kMaxScreenBatch) return hipErrorInvalidValue;
        const int nmax2 = fill_screen_args(db, *next, align_filter, &fa.next);
        if (nmax2 < 0) return hipErrorInvalidValue;
        fa.next.skip_d2 = next_v2 ? 1 : 0;
        fa.next.nb = align_blocks((nmax2 + kGroup - 1) / kGroup, next->nq, true);
        extra = fa.next.nb * next->nq;
#ifdef SCL_DIAGNOSTICS
        if (probe == 5 || probe == 6) fa.next.align_filter = probe == 5 ? 2 : 3;
#endif
    }
    const size_t lds = (extra || self_align) && lds0 > lds1 ? lds0 : lds1;
    fa.align_blocks = extra;
    const int patterned = fused_patterned_blocks(sb.nq, blocks, extra);
    const int slots = 8 * ((blocks + 7) >> 3);                               // alignment slots inside the pattern
    const int grid = patterned + (extra > slots ? extra - slots : 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kScreenWaves * kWave), lds, stream, fa);
    return hipGetLastError();
}

hipError_t launch_sc_screen_batch(const DbView &db, const ScreenBatch &sb, int SR, int align_filter, int num_cu, hipStream_t stream, int phases,
                                  const ScreenBatch *next, const ScreenBatch *prev)
{
    if (sb.nq < 1 || sb.nq > kMaxScreenBatch || !sc_screen_supported(db, SR)) return hipErrorInvalidValue;
    if (sc_screen_is_wide(db, SR)) return launch_screen_grid<20, 180, 19, 2, 2>(db, sb, align_filter, num_cu, stream, phases, next, prev);   // 80 x 180
    // (64 x 120: the first form scores batches of one to three scans -- blocking calls -- and the kernel built for two waves per SIMD takes
    //  4 us less of a one-scan call than the one for three, which was the better one when the first form still scored batches of four)
    return launch_screen_grid<16, 120, 13, 2, 3>(db, sb, align_filter, num_cu, stream, phases, next, prev);                                   // 64 x 120
}

// Can a batch of nq scans have its finishing deferred (second form of the products, extra waves available)?
bool sc_screen_can_defer(const DbView &db, int SR, int nq)
{
    static const int fuse_env = scl_lab_int("SCL_SCREEN_FUSE", 0);
    static const int align_form_env = scl_lab_int("SCL_ALIGN_FORM", 2);
    static const int side_env = scl_lab_int("SCL_ALIGN_SIDE", 0);
    static const int variant = scl_lab_int("SCL_SCREEN_VARIANT", 0);
    static const int probe = scl_lab_int("SCL_SCREEN_PROBE", 0);
    static const int parts_env = scl_lab_int("SCL_FUSE_PARTS", 3);
    if (!(parts_env & 2)) return false;
    if (!fuse_env || align_form_env == 1 || side_env != 0 || variant != 0 || probe != 0 || !screen_second_form() || !db.halign || !sc_screen_supported(db, SR)) return false;
    const bool wide = sc_screen_is_wide(db, SR);
    const int v2_min = screen_v2_min_env() > 0 ? screen_v2_min_env() : (wide ? 2 : 4);
    return nq >= v2_min && nq <= sc_screen_max_batch(db, SR);
}

}  // namespace scl

#ifdef S2_STAMP
extern "C" int scl_debug_s2_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scl::g_s2_stamps), sizeof(unsigned long long) * 4 * 64);
}
#endif
