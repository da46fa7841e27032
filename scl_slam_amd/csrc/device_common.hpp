// device_common.hpp -- shared device helpers for the gfx950 kernels.
//
// Numerics contract: every translation unit is built
// with -ffp-contract=off, so a*b+c is never fused behind our back.  fma() is
// written explicitly where (and only where) the product is exact in fp64
// (products of two widened floats), which makes fma(a,b,c) == a*b+c bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"

namespace scl {

constexpr int kWave = 64;                 // gfx950 wavefront
constexpr int kNoPoint = -1000;           // D.h:1411

// ---- ordered-int encoding of floats: signed-int order == float order -------
__device__ __forceinline__ int float_to_ordered(float f)
{
    int b = __float_as_int(f);
    return b >= 0 ? b : (b ^ 0x7fffffff);
}
__device__ __forceinline__ float ordered_to_float(int o)
{
    return __int_as_float(o >= 0 ? o : (o ^ 0x7fffffff));
}

// ---- atanf as the reference's platform computes it (D.h:1357-1372 call std::atan(float)): glibc's float atanf
// (sysdeps/ieee754/flt-32/s_atanf.c, fdlibm's algorithm) restated operation by operation -- fp32 only, no contraction
// (-ffp-contract=off), IEEE division (-fhip-fp32-correctly-rounded-divide-sqrt).  The same restatement in oracle/sc_oracle.c
// equals libm's atanf on all 2^32 inputs in the build container (oracle/tools/atanf_exhaustive.c); this copy is checked on the
// device against the block checksums of tests/golden/atanf_blocks.json (tests/test_gpu_make_sc.py).  Constants by bit pattern.
//
// Written WITHOUT divergent branches: the five intervals of the argument reduction differ only in the operands of ONE division
// (id -1: x / 1 = x exactly), so the interval selects numerator, denominator and the hi / lo constants, every lane divides once and
// evaluates the one polynomial, and the special ranges (|x| >= 2^25, NaN, |x| < 2^-29) are selected at the end.  The branchy form cost
// the descriptor scatter a division per interval AND per quadrant of xy2theta under divergence (an IEEE fp32 division is ~15
// instructions): the kernel was bound by vector instructions, not by HBM.
__device__ __forceinline__ float atanf_glibc(float x)
{
    const unsigned int hx = (unsigned int)__float_as_int(x), ix = hx & 0x7fffffffu;
    const float ax = fabsf(x);
    const bool small = ix < 0x3ee00000u;                  // |x| < 0.4375: no reduction, sign kept
    const bool i0 = ix < 0x3f300000u, i1 = ix < 0x3f980000u, i2 = ix < 0x401c0000u;
    // numerator / denominator of the reduced argument (id 0: (2x-1)/(2+x), 1: (x-1)/(x+1), 2: (x-1.5)/(1+1.5x), 3: -1/x)
    const float num = small ? x : (i0 ? 2.0f * ax - 1.0f : (i1 ? ax - 1.0f : (i2 ? ax - 1.5f : -1.0f)));
    const float den = small ? 1.0f : (i0 ? 2.0f + ax : (i1 ? ax + 1.0f : (i2 ? 1.0f + 1.5f * ax : ax)));
    const float hi = __int_as_float(i0 ? 0x3eed6338 : (i1 ? 0x3f490fda : (i2 ? 0x3f7b985e : 0x3fc90fda)));
    const float lo = __int_as_float(i0 ? 0x31ac3769 : (i1 ? 0x33222168 : (i2 ? 0x33140fb4 : 0x33a22168)));
    const float t = num / den;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (__int_as_float(0x3eaaaaab) + w * (__int_as_float(0x3e124925) + w * (__int_as_float(0x3dba2e6e) +
                     w * (__int_as_float(0x3d886b35) + w * (__int_as_float(0x3d4bda59) + w * __int_as_float(0x3c8569d7))))));
    const float s2 = w * (__int_as_float(0xbe4ccccd) + w * (__int_as_float(0xbde38e38) + w * (__int_as_float(0xbd9d8795) +
                     w * (__int_as_float(0xbd6ef16b) + w * __int_as_float(0xbd15a221)))));
    const float p = t * (s1 + s2);
    const float r_small = t - p;                          // id < 0: x - x*(s1+s2)
    const float r_red = hi - ((p - lo) - t);
    float r = small ? r_small : ((hx >> 31) ? -r_red : r_red);
    if (ix < 0x31000000u) r = x;                          // |x| < 2^-29
    if (ix >= 0x4c000000u) {                              // |x| >= 2^25 (inf included): +-(hi3 + lo3); NaN: x + x
        const float big = __int_as_float(0x3fc90fda) + __int_as_float(0x33a22168);
        r = ix > 0x7f800000u ? x + x : ((hx >> 31) ? -big : big);
    }
    return r;
}

// xy2theta, D.h:1352-1374.  The four quadrant branches of the reference differ in the operands of the division (y / x, y / (-x), y / x,
// (-y) / x) and in how the angle is placed (k a, 180 - k a, 180 + k a, 360 - k a): operands and placement are selected, every lane
// divides once.  The conditions are the reference's own (x = -0.0 counts as x >= 0: y / -0.0 = -inf and the angle is -90 or 450 --
// kept); a NaN coordinate fails all four and yields NaN (the reference falls off the end).
__device__ __forceinline__ float xy2theta(float x, float y)
{
    const double k = 180.0 / 3.14159265358979323846;
    const bool xp = x >= 0, xn = x < 0, yp = y >= 0, yn = y < 0;
    const bool q2 = xn & yp, q3 = xn & yn, q4 = xp & yn;
    const float a = atanf_glibc((q4 ? -y : y) / (q2 ? -x : x));
    const double ka = k * (double)a;
    double r = ka;                                        // x >= 0, y >= 0
    r = q2 ? 180.0 - ka : r;
    r = q3 ? 180.0 + ka : r;
    r = q4 ? 360.0 - ka : r;
    return ((xp | xn) & (yp | yn)) ? (float)r : __int_as_float(0x7fc00000);
}

// int(ceil(v)) the way the reference's x86-64 build evaluates it (NaN -> INT_MIN)
__device__ __forceinline__ int ceil_to_int_x86(double v)
{
    const double c = ceil(v);
    if (!(c >= -2147483648.0 && c <= 2147483647.0)) return (-2147483647 - 1);
    return (int)c;
}

// ---- the descriptor's bins of a point, two ways (make_sc.hip) -------------------------------------------------------------------
// The descriptor depends on a point only through (ring, sector, dropped?).  sc_bin_exact is the reference's chain (D.h:1425-1435):
// IEEE sqrtf, xy2theta with glibc's atanf, the double quotients, ceil -- ~220 vector instructions, four IEEE divisions among them: the
// scatter kernel was bound by them, not by HBM.  sc_bin_fast computes the SAME integers from cheap approximations and says whether it
// is sure: with q~ an approximation of the exact chain's q = range / max_radius * R (resp. theta / 360 * S) and |q~ - q| <= E,
// ceil(q~) = ceil(q) whenever q~ is farther than E from every integer; the guards below are four times the E derived here, and a
// point inside a guard band (2-3 in 10 000) takes the exact chain.  Errors:
//   ring:   s = x*x + y*y is the chain's own float arithmetic; v_sqrt_f32 is within 1 ulp where the chain's sqrtf is correctly
//           rounded (1.5 ulp apart), the product with fl32(R / max_radius) adds two roundings: |q~ - q| <= 3e-7 q <= 2.4e-5 (R <= 80).
//   sector: the chain's float angle is within 4e-5 degrees of the true one (atanf < 1 ulp, the quotient's half ulp, and the final
//           rounding of an angle up to 360: half of 3.05e-5); here: v_rcp_f32 (1 ulp), a degree-17 odd polynomial for atan on [0, 1]
//           (Abramowitz & Stegun 4.4.49, |error| <= 1.4e-8; 9.3e-8 as evaluated in fp32), the octant placement (pi's rounding, one
//           rounding at <= 2 pi: 2.9e-7 rad), the product with fl32(S / 2 pi) (two roundings at <= 180: 2.2e-5): together 6e-5 sectors
//           at S = 180.
//   x or y zero, NaN, or the larger coordinate outside [1e-18, 1e18]: not sure (the chain's quirks live there: y / -0.0, 0 / 0).
// scl_selftest_bin_paths (tests/test_gpu_make_sc.py) compares the two on 2^30 random, boundary-hugging and special points: no
// disagreement where the fast path is sure.
constexpr float kBinGuardRing = 1.0e-4f;
constexpr float kBinGuardSect = 2.5e-4f;

__device__ __forceinline__ void sc_bin_exact(float px, float py, int R, int S, double max_radius, int &ring, int &sect, bool &drop)
{
    const float azim_range = sqrtf(px * px + py * py);             // D.h:1425
    const float azim_angle = xy2theta(px, py);                     // D.h:1426
    drop = (double)azim_range > max_radius;                        // D.h:1429
    ring = max(min(R, ceil_to_int_x86(((double)azim_range / max_radius) * R)), 1);   // D.h:1434
    sect = max(min(S, ceil_to_int_x86(((double)azim_angle / 360.0) * S)), 1);        // D.h:1435
}

__device__ __forceinline__ bool sc_bin_fast(float x, float y, int R, int S, float c_ring, float c_sect, int &ring, int &sect, bool &drop)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float hi = fmaxf(ax, ay), lo = fminf(ax, ay);
    bool sure = (x == x) & (y == y) & (lo > 0.0f) & (hi >= 1.0e-18f) & (hi <= 1.0e18f);
    const float s = x * x + y * y;                                 // the chain's own operations (no contraction)
    const float qr = __builtin_amdgcn_sqrtf(s) * c_ring;
    sure &= fabsf(qr - rintf(qr)) > kBinGuardRing;
    drop = qr > (float)R;
    ring = max(min((int)ceilf(qr), R), 1);
    const float a = lo * __builtin_amdgcn_rcpf(hi);                // in (0, 1]
    const float z = a * a;
    float p = fmaf(z, 0.0028662257f, -0.0161657367f);
    p = fmaf(z, p, 0.0429096138f);
    p = fmaf(z, p, -0.0752896400f);
    p = fmaf(z, p, 0.1065626393f);
    p = fmaf(z, p, -0.1420889944f);
    p = fmaf(z, p, 0.1999355085f);
    p = fmaf(z, p, -0.3333314528f);
    p = fmaf(z, p, 1.0f);
    float t = a * p;                                               // atan(lo / hi), [0, pi / 4]
    t = ay > ax ? 1.57079632679f - t : t;
    t = x < 0.0f ? 3.14159265359f - t : t;
    t = y < 0.0f ? 6.28318530718f - t : t;
    const float qs = t * c_sect;
    sure &= fabsf(qs - rintf(qs)) > kBinGuardSect;
    sect = max(min((int)ceilf(qs), S), 1);
    return sure;
}

// ---- wave-level lexicographic arg-min on (double value, int tag) -----------
__device__ __forceinline__ void wave_argmin(double &v, int &tag)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off, kWave);
        const int ot = __shfl_xor(tag, off, kWave);
        const bool take = (ov < v) | ((ov == v) & (ot < tag));
        v = take ? ov : v;
        tag = take ? ot : tag;
    }
}

// Same result without the LDS round trips of __shfl_xor (ds_bpermute): DPP lane exchanges inside each
// 16-lane row (xor 1, xor 2, half-row mirror, row mirror), then the four row results through scalar
// registers.  Requires all 64 lanes active and NaN-free values.  Every lane returns the wave result.
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int src_lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src_lane),
                            __builtin_amdgcn_readlane(__double2loint(v), src_lane));
}
__device__ __forceinline__ double wave_min_f64_dpp(double v)
{
    double o;
    o = dpp_move_f64<0xB1>(v);  v = o < v ? o : v;      // quad_perm [1,0,3,2]
    o = dpp_move_f64<0x4E>(v);  v = o < v ? o : v;      // quad_perm [2,3,0,1]
    o = dpp_move_f64<0x141>(v); v = o < v ? o : v;      // row_half_mirror
    o = dpp_move_f64<0x140>(v); v = o < v ? o : v;      // row_mirror
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    const double a = r1 < r0 ? r1 : r0, b = r3 < r2 ? r3 : r2;
    return b < a ? b : a;
}
__device__ __forceinline__ int wave_min_i32_dpp(int t)
{
    int o;
    o = __builtin_amdgcn_update_dpp(0, t, 0xB1, 0xf, 0xf, false);  t = o < t ? o : t;
    o = __builtin_amdgcn_update_dpp(0, t, 0x4E, 0xf, 0xf, false);  t = o < t ? o : t;
    o = __builtin_amdgcn_update_dpp(0, t, 0x141, 0xf, 0xf, false); t = o < t ? o : t;
    o = __builtin_amdgcn_update_dpp(0, t, 0x140, 0xf, 0xf, false); t = o < t ? o : t;
    const int r0 = __builtin_amdgcn_readlane(t, 0), r1 = __builtin_amdgcn_readlane(t, 16);
    const int r2 = __builtin_amdgcn_readlane(t, 32), r3 = __builtin_amdgcn_readlane(t, 48);
    const int a = r1 < r0 ? r1 : r0, b = r3 < r2 ? r3 : r2;
    return b < a ? b : a;
}
// sum / max of one float per lane over the wave (same exchange pattern; every lane returns the result)
__device__ __forceinline__ float wave_sum_f32_dpp(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float wave_max_f32_dpp(float v)
{
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xf, 0xf, false)));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// lexicographic arg-min on (value, tag): minimum value, lowest tag among the lanes that hold it
__device__ __forceinline__ void wave_argmin_dpp(double &v, int &tag)
{
    const double m = wave_min_f64_dpp(v);
    const int t = (v == m) ? tag : 0x7fffffff;
    tag = wave_min_i32_dpp(t);
    v = m;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(k, off, kWave);
        k = o < k ? o : k;
    }
    return k;
}

}  // namespace scl
