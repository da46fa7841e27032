// align_exact.hpp -- fastAlignUsingVkey (descriptor.h:1491-1511) for ONE keyframe, wave-wide, in the reference's own fp64 arithmetic:
// what the matrix-core alignment filters (sc_screen.hip) fall back to, and what the small exact pass (sc_masked.hip) runs for a pair
// that reaches it without a first shift.
#pragma once

#include "device_common.hpp"

namespace scl {

namespace {

__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void pin3(double &a, double &b, double &c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c) :: "memory"); }

// fastAlignUsingVkey (D.h:1491-1511) for one keyframe, wave-wide, in the reference's own fp64 arithmetic; returns the
// reference's arg-min shift.  vk = the keyframe's sector key at sectors 2*ll, 2*ll+1 (ll = min(lane, S/2 - 1)).  Same
// arithmetic as the exact evaluation in the alignment phase of sc_distance_wave_kernel (tie rules included).
template <int S>
__device__ __forceinline__ int align_keyframe_exact(const double2 vk, int lane, double *vk2, const double *vq)
{
    constexpr int L = S >> 1;
    const bool active = lane < L;
    const int ll = active ? lane : L - 1;
    const int j0 = 2 * ll;
    const double kInf = __longlong_as_double(0x7ff0000000000000LL);
    wave_fence();
    *reinterpret_cast<double2 *>(vk2 + j0) = vk;
    *reinterpret_cast<double2 *>(vk2 + j0 + S) = vk;
    wave_fence();
    double best = kInf;
    int bshift = 0x7fffffff;
    {
        const double *p = vk2 + S - j0;
        const double2 *pp = reinterpret_cast<const double2 *>(p);
        double prev = p[-1];
        double ss0 = 0.0, ss1 = 0.0;
        constexpr int npair = S >> 1;
        constexpr int BT = 3;                              // (5 in sc_distance.hip; here the path is rare and registers are short)
        static_assert(npair % BT == 0, "alignment batches must tile the sector pairs");
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
        const double n0 = sqrt(ss0), n1 = sqrt(ss1);
        if (active && n0 < kBigDist) { best = n0; bshift = j0; }
        if (active && n1 < kBigDist && n1 < best) { best = n1; bshift = j0 + 1; }
    }
    wave_argmin_dpp(best, bshift);
    return __builtin_amdgcn_readfirstlane(best < kBigDist ? bshift : 0);
}

// The same evaluation for grids with more than two sectors per lane (80 x 180: three shifts per lane).
template <int S>
__device__ __forceinline__ int align_keyframe_wide(const double (&vk)[(S + kWave - 1) / kWave], int lane, double *vk2, const double *vq)
{
    constexpr int SPL = (S + kWave - 1) / kWave;           // shifts (and sectors) per lane
    constexpr int LA = S / SPL;                            // active lanes
    static_assert(S % SPL == 0 && LA <= kWave, "shifts must tile the lanes");
    const double kInf = __longlong_as_double(0x7ff0000000000000LL);
    wave_fence();
    if (lane < LA) {
#pragma unroll
        for (int u = 0; u < SPL; ++u) { vk2[SPL * lane + u] = vk[u]; vk2[SPL * lane + u + S] = vk[u]; }
    }
    if (lane == 0) { vk2[2 * S] = vk[0]; }                 // one past the doubled key: read by the last slide, never used
    wave_fence();
    // lane owns shifts s_k = SPL * lane + k; the key shifted by s at sector t is vk[(t - s) mod S] = p[t - k], p = vk2 + S - SPL * lane
    const int ll = lane < LA ? lane : LA - 1;
    const double *p = vk2 + S - SPL * ll;
    double ss[SPL], w[SPL];
#pragma unroll
    for (int k = 0; k < SPL; ++k) { ss[k] = 0.0; w[k] = p[-k]; }
#pragma unroll 4
    for (int t = 0; t < S; ++t) {                          // sector order, as the reference's norm (D.h:1500-1502)
        const double q = vq[t];
        const double nxt = p[t + 1];
#pragma unroll
        for (int k = 0; k < SPL; ++k) { const double d = q - w[k]; ss[k] = ss[k] + d * d; }
#pragma unroll
        for (int k = SPL - 1; k > 0; --k) w[k] = w[k - 1];
        w[0] = nxt;
    }
    double best = kInf;
    int bshift = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {                        // ascending shifts, strict <: ties keep the lower shift
        const double nk = sqrt(ss[k]);
        if (lane < LA && nk < kBigDist && nk < best) { best = nk; bshift = SPL * lane + k; }
    }
    wave_argmin_dpp(best, bshift);
    return __builtin_amdgcn_readfirstlane(best < kBigDist ? bshift : 0);
}

}  // namespace

}  // namespace scl
