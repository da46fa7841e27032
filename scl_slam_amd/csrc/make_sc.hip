// make_sc.hip -- K3: Scan Context construction + database ingest.
//
//   make_sc_scatter / make_sc_finalize   makeScancontext, include/descriptor.h:1404-1461
//   ingest                               save(): ring key D.h:1463-1475, sector key
//                                        D.h:1477-1489, wire decode D.h:1572-1585,
//                                        plus the column norms of D.h:1523 (hoisted: they
//                                        depend on one descriptor only)
//
// Descriptor construction is a max-z scatter into an R x S polar image.  Taking the
// maximum is order independent (NaN z never wins `desc < z`, D.h:1438), so the serial
// loop parallelises exactly: every workgroup streams a contiguous slice of one cloud
// (one 16-byte load per point record: x,y,z,+pad), bins it with the reference's
// mixed fp32/fp64 arithmetic, and keeps a private polar tile in LDS updated with
// integer atomicMax on an order-preserving float encoding; tiles are merged into the
// scan's global image with one atomicMax per touched cell.  HBM-bound: n*16 B in,
// R*S*4 out.  A BATCH of up to 16 scans is two launches: the scatter over all of
// them, and ingest_kernel (one workgroup per scan), which finalizes on the way in,
// writes every array of the database slot and leaves the tiles in their initial
// state for the next batch (round 4: init, scatter, finalize, ingest per scan).
#include <atomic>
#include <cstdio>

#include "device_common.hpp"
#include "kernels.hpp"

namespace scl {

namespace {

constexpr int kScThreads = 256;

// The same scatter for a BATCH of scans in one launch (the keyframes of a robot team that arrive together, DM.h:988-1025 once per
// keyframe): workgroup -> (scan, slice of its cloud), a private LDS tile per workgroup, merged into the scan's global tile with one
// atomicMax per touched cell.  The global tiles are in their initial state when the launch starts and are put back into it by the
// kernel that consumes them (ingest_kernel), so a batch costs two launches whatever its size.  Four points per thread are in flight.
__device__ __forceinline__ void scatter_body(const ScanBatch &b, const int wg, int stride, int R, int S,
                                             double lidar_height, double max_radius, int *gtiles, int lab, int *tile /* R*S ints of LDS */)
{
    const bool exact_only = lab & 1;                       // (diagnostics builds: 1 = the reference's chain for every point, 2 / 4 = ablations)
    const int cells = R * S;
    const float c_ring = (float)((double)R / max_radius), c_sect = (float)((double)S / 6.283185307179586);
    const int init = float_to_ordered((float)kNoPoint);
    for (int i = threadIdx.x; i < cells; i += blockDim.x) tile[i] = init;
    int s = 0;
#pragma unroll
    for (int j = 1; j < kMaxScBatch; ++j) s += (j < b.count && wg >= b.first_wg[j]) ? 1 : 0;
    const int slices = b.first_wg[s + 1] - b.first_wg[s];
    const int slice = wg - b.first_wg[s];
    const int n = b.n[s];
    const unsigned char *points = b.points[s];
    const int per_block = (n + slices - 1) / slices;
    const int begin = slice * per_block;
    int end = begin + per_block; end = end > n ? n : end;
    __syncthreads();
    const bool vec = (stride & 15) == 0;
    // SIXTEEN points per thread and round, every record of the round requested before the first is binned: the loop is bound by
    // memory round trips, not by bytes or arithmetic (measured: four points per round, 14.5 us for 30.7 MB whether or not the next
    // round's loads were issued ahead -- two waves per SIMD hide nothing, and a round's arithmetic is a sixth of its latency)
    constexpr int U = 16;
    const int p_end = (lab & 4) ? begin : end;
    for (int p0 = begin + (int)threadIdx.x; p0 < p_end; p0 += U * kScThreads) {
        float px[U], py[U], pzr[U];
        // (unconditional loads from a clamped index: a load inside `if (p < end)` is a block of its own, and the compiler ended every
        //  such block with s_waitcnt vmcnt(0) -- sixteen dependent round trips per round, however the source was arranged)
        if (vec) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * kScThreads;
                const float4 v = *reinterpret_cast<const float4 *>(points + (size_t)(p < end ? p : end - 1) * (size_t)stride);
                px[u] = v.x; py[u] = v.y; pzr[u] = p < end ? v.z : __int_as_float(0x7fc00000);    // past the slice: NaN z never enters a cell
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * kScThreads;
                const float *f = reinterpret_cast<const float *>(points + (size_t)(p < end ? p : end - 1) * (size_t)stride);
                px[u] = f[0]; py[u] = f[1]; pzr[u] = p < end ? f[2] : __int_as_float(0x7fc00000);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float pz = (float)((double)pzr[u] + lidar_height);            // D.h:1422
            int ring, sect; bool drop;
            // the bins from cheap approximations where those are sure of them (device_common.hpp), the reference's chain otherwise
            const bool sure = !exact_only && sc_bin_fast(px[u], py[u], R, S, c_ring, c_sect, ring, sect, drop);
            if (!sure) sc_bin_exact(px[u], py[u], R, S, max_radius, ring, sect, drop);      // D.h:1425-1435
            if (drop) continue;                                                  // D.h:1429
            if (pz != pz) continue;                                              // NaN never passes `<` (D.h:1438)
            atomicMax(&tile[(ring - 1) * S + (sect - 1)], float_to_ordered(pz));
        }
    }
    __syncthreads();
    int *g = gtiles + (size_t)s * cells;
    if (lab & 2) return;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) {
        const int v = tile[i];
        if (v != init) atomicMax(&g[i], v);
    }
}

__global__ __launch_bounds__(kScThreads) void make_sc_batch_scatter_kernel(ScanBatch b, int stride, int R, int S,
                                                                           double lidar_height, double max_radius, int *gtiles, int lab)
{
    extern __shared__ int tile[];
    scatter_body(b, (int)blockIdx.x, stride, R, S, lidar_height, max_radius, gtiles, lab, tile);
}

__global__ void make_sc_init_kernel(int *gtile, int cells)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cells) gtile[i] = float_to_ordered((float)kNoPoint);
}

// descriptor only (scl_make_descriptor: nothing is stored): tiles -> wire-format values, tiles back to their initial state
__global__ void make_sc_finalize_kernel(int *gtile, int cells, float *values)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cells) {
        float v = ordered_to_float(gtile[i]);
        if (v == (float)kNoPoint) v = 0.0f;                            // D.h:1450-1453
        values[i] = v;                                                 // row-major == vT order, D.h:1454
        gtile[i] = float_to_ordered((float)kNoPoint);
    }
}

// One workgroup per descriptor.  values: [count][R*S] row-major floats.
// floor(n / d) for n, d < 2^16 by one multiply-high with M = ceil(2^32 / d) (exact in that range): the index splits of the loops
// below (i -> row, column) were integer divisions by a runtime S -- ~40 instructions each, a third of this kernel's time
__device__ __forceinline__ unsigned int fastdiv_magic(int d) { return 0xffffffffu / (unsigned int)d + 1u; }
__device__ __forceinline__ int fastdiv(int n, unsigned int magic) { return (int)__umulhi((unsigned int)n, magic); }

// sum over a wave of one double per lane (xor exchanges: every lane ends with the same value; a wave that runs it on the same data gets
// the same bits).  For the quantities whose summation ORDER nothing depends on: norms and error bounds of the screening's operands.
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

// tiles != nullptr: the descriptors come straight from the scatter's global tiles (ordered-int max-z images, one per workgroup):
// finalize (D.h:1446-1456: NO_POINT -> 0, row-major floats) happens on the way into LDS, the wire-format values go to vals_out
// (what makeAndSaveDescriptorAndKey returns, D.h:1604-1611) and the tile is put back into its initial state for the next batch.
#ifdef SCL_DIAGNOSTICS
__device__ unsigned long long g_ingest_stamps[16];     // ticks (100 MHz) of workgroup 0 between the ingest's phases; [15] = launches (scl_lab: SCL_INGEST_STAMPS=1 prints them)
#define ING_STAMP(k) do { if (wg == 0 && threadIdx.x == 0) { const unsigned long long now__ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_ingest_stamps[k], now__ - ing_t0__); ing_t0__ = now__; } } while (0)
#else
#define ING_STAMP(k) ((void)0)
#endif
__device__ __forceinline__ void ingest_body(const IngestArgs &ia, const int wg, float *sv /* LDS: [R][S+1], then the S reciprocal norms, ... */)
{
#ifdef SCL_DIAGNOSTICS
    unsigned long long ing_t0__ = __builtin_amdgcn_s_memrealtime();
    if (wg == 0 && threadIdx.x == 0) atomicAdd(&g_ingest_stamps[15], 1ull);
#endif
    const float *values = ia.values; const int first_slot = ia.first_slot; float4 *desc = ia.desc; double *vkey = ia.vkey; double *norm = ia.norm;
    float *rkey = ia.rkey; float4 *rkey4 = ia.rkey4; uint2 *hdesc = ia.hdesc; unsigned int *kmask = ia.kmask; unsigned short *hkey = ia.hkey;
    const int hstride = ia.hstride, cap = ia.cap, R = ia.R, S = ia.S; unsigned char *halign = ia.halign; int *tiles = ia.tiles; float *vals_out = ia.vals_out;
    const int LS = S + 1;                         // odd-ish stride: column walks hit distinct banks
    const int RG = (R + 3) >> 2;
    const unsigned int mS = fastdiv_magic(S);
    const int slot = first_slot + wg;
    const float *src = values ? values + (size_t)wg * R * S : nullptr;
    float *siv = sv + R * LS;
    double *svk = reinterpret_cast<double *>(sv + ((R * LS + S + 1) & ~1));     // [S] the sector key, for its norm
    double *sdd = svk + S;                                                      // [S] squared rounding errors of the fp16 sector key
    double *snorm = sdd + S;                                                    // [S] the columns' fp64 norms (the bound's loop reads them with two lanes per column)
    // (eight loads of a thread in flight before the first is used, from a clamped index: a load per loop step, with the stores that
    //  depend on it behind it, was thirty memory round trips one after the other -- most of this kernel's time)
    constexpr int LU = 8;
    const int cells = R * S;
    if (tiles) {
        int *t = tiles + (size_t)wg * cells;
        float *vo = vals_out ? vals_out + (size_t)wg * cells : nullptr;
        const int init = float_to_ordered((float)kNoPoint);
        for (int i0 = threadIdx.x; i0 < cells; i0 += LU * (int)blockDim.x) {
            int o[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) { const int i = i0 + u * (int)blockDim.x; o[u] = t[i < cells ? i : cells - 1]; }
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int i = i0 + u * (int)blockDim.x;
                if (i < cells) {
                    const int r = fastdiv(i, mS), c = i - r * S;
                    float v = ordered_to_float(o[u]);
                    if (v == (float)kNoPoint) v = 0.0f;                // D.h:1450-1453
                    sv[r * LS + c] = v;
                    if (vo) vo[i] = v;                                 // row-major == vT order, D.h:1454
                    t[i] = init;
                }
            }
        }
    } else {
        for (int i0 = threadIdx.x; i0 < cells; i0 += LU * (int)blockDim.x) {
            float o[LU];
#pragma unroll
            for (int u = 0; u < LU; ++u) { const int i = i0 + u * (int)blockDim.x; o[u] = src[i < cells ? i : cells - 1]; }
#pragma unroll
            for (int u = 0; u < LU; ++u) {
                const int i = i0 + u * (int)blockDim.x;
                if (i < cells) { const int r = fastdiv(i, mS), c = i - r * S; sv[r * LS + c] = o[u]; }
            }
        }
    }
    __syncthreads();
    ING_STAMP(0);

    // tiled copy: element (rg, c) = rows 4rg..4rg+3 of column c
    float4 *dslot = desc + (size_t)slot * RG * S;
    for (int i = threadIdx.x; i < RG * S; i += blockDim.x) {
        const int rg = fastdiv(i, mS), c = i - rg * S;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 4 * rg + k;
            v[k] = r < R ? sv[r * LS + c] : 0.0f;
        }
        dslot[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
    ING_STAMP(1);
    // sector key (column mean, D.h:1482-1486) and column norm (D.h:1523), sequential over rings
    for (int c = threadIdx.x; c < S; c += blockDim.x) {
        double sum = 0.0, ss = 0.0;
        for (int r = 0; r < R; ++r) {
            const double x = (double)sv[r * LS + c];
            sum = sum + x;
            ss = ss + x * x;
        }
        vkey[(size_t)slot * S + c] = sum / (double)R;
        svk[c] = sum / (double)R;
        const double nrm = sqrt(ss);
        norm[(size_t)slot * S + c] = nrm;
        snorm[c] = nrm;
        // screening pass operand (sc_screen.hip): fp32 reciprocal norm; 0 = all-zero column, NaN = score this keyframe exactly
        float iv = __int_as_float(0x7fc00000);
        if (nrm == 0.0) iv = 0.0f;
        else if (nrm >= 0x1p-60 && nrm <= 0x1p60) iv = (float)(1.0 / nrm);
        siv[c] = iv;
    }
    ING_STAMP(2);
    // ring key (row mean narrowed to float, D.h:1468-1472), sequential over sectors
    for (int r = threadIdx.x; r < 4 * RG; r += blockDim.x) {
        float key = 0.0f;
        if (r < R) {
            double sum = 0.0;
            for (int c = 0; c < S; ++c) sum = sum + (double)sv[r * LS + c];
            key = (float)(sum / (double)S);
        }
        rkey[(size_t)slot * 4 * RG + r] = key;
        if (rkey4) reinterpret_cast<float *>(rkey4)[((size_t)(r >> 2) * cap + slot) * 4 + (r & 3)] = key;   // nullptr: staging slot
    }
    ING_STAMP(3);
    // the screening pass's copy (sc_screen.hip): x * inv in fp32, rounded to nearest fp16; sector-major
    __syncthreads();
    ING_STAMP(4);
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 *hslot = reinterpret_cast<h4 *>(hdesc + (size_t)slot * hstride);
    const int RGH = hdesc_sector(RG);                        // ring groups of the copy: rows >= R are zero
    const unsigned int mRGH = fastdiv_magic(RGH);
    for (int i = threadIdx.x; i < S * RGH; i += blockDim.x) {
        const int c = fastdiv(i, mRGH), rg = i - c * RGH;
        const float iv = siv[c];
        h4 hv;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = 4 * rg + k;
            hv[k] = (_Float16)((r < R ? sv[r * LS + c] : 0.0f) * iv);
        }
        hslot[i] = hv;
    }
    ING_STAMP(5);
    // the chunk-major image of the same values: [ring part][chunk][sector 0 .. S+15], 8 rings = 16 B per entry
    if (hdesc2_elems(RG, S)) {
        typedef _Float16 h8v __attribute__((ext_vector_type(8)));
        h8v *h2 = reinterpret_cast<h8v *>(hdesc + (size_t)slot * hstride + (size_t)hdesc2_offset(RG, S));
        const int n2 = hdesc2_elems(RG, S) / 2;                                 // entries of 16 B: ring parts x 4 chunks x (S + 16) sectors
        const unsigned int mS16 = fastdiv_magic(S + 16);
        for (int i = threadIdx.x; i < n2; i += blockDim.x) {
            const int hj = fastdiv(i, mS16), sx = i - hj * (S + 16);
            const int c = sx < S ? sx : sx - S;
            const float iv = siv[c];
            h8v hv;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int r = 8 * hj + k;
                hv[k] = (_Float16)((r < R ? sv[r * LS + c] : 0.0f) * iv);
            }
            h2[i] = hv;
        }
    }
    ING_STAMP(6);
    // the sector key as a unit vector in fp16, behind the copy (first stage of the alignment filter, sc_screen.hip)
    {
        const int SK = hkey_halfs(S);
        _Float16 *hk = reinterpret_cast<_Float16 *>(hdesc + (size_t)slot * hstride + (size_t)RGH * S);
        // |k|^2 of the sector key: lane partial sums + a wave reduction, by every wave for itself (the same bits in every wave).  Not the
        // reference's arithmetic -- the unit key only feeds the alignment FILTER, whose bounds do not depend on this sum's order; the
        // serial sum by every thread was 120 dependent fp64 additions.
        double n2 = 0.0;
        for (int c = (int)(threadIdx.x & (kWave - 1)); c < S; c += kWave) n2 = n2 + svk[c] * svk[c];
        n2 = wave_sum_f64(n2);
        const double nrm = sqrt(n2);
        const bool usable = nrm > 0.0 && nrm < 1.0e300;                   // zero, NaN, inf: all zero -> every shift ties -> exact evaluation
        const double rnrm = 1.0 / nrm;                                    // unit key u = k * (1 / |k|): one division per thread, not one per entry
        _Float16 *hd = reinterpret_cast<_Float16 *>(hkey + (size_t)slot * hkey_row_halfs(S));   // the dense table of the same keys
        // second fp16 part of an entry: fp16(2^11 (u - kh)) -- scaled so that it is a normal number; u - kh is exact in fp64
        auto parts = [&](int c, _Float16 *h, _Float16 *l) {
            const double u = svk[c] * rnrm;
            const _Float16 kh = (_Float16)(float)u;
            *h = kh; *l = (_Float16)(float)((u - (double)(float)kh) * 2048.0);
        };
        for (int c = threadIdx.x; c < SK; c += blockDim.x) {
            _Float16 v = (_Float16)0.0f, vl = (_Float16)0.0f;
            if (c < S && usable) parts(c, &v, &vl);
            hk[c] = v; hd[c] = v; hd[SK + 8 + c] = vl;
        }
        // |u - kh|_2: how far the fp16 key is from the unit key it stands for, rounded UP (second float behind the key).  The
        // alignment's first stage bounds its error by Cauchy-Schwarz on these ACTUAL norms (2.4e-4 as a rule) instead of the worst
        // case of fp16 rounding (4.9e-4 per key): the lead it demands of the best shift falls from 3e-3 to ~1e-3, and fewer pairs go
        // on to the second stage.  An entry below fp16's normal range counts with the larger of its rounding error and its value (a
        // matrix core may take it for zero).  The squares are formed in parallel (sdd), thread 0 adds them up in sector order.
        // (u as stored differs from k / |k| by two roundings, 2.3e-16 |u|: inside the 1e-10 the bound is raised by.)
        for (int c = threadIdx.x; c < S; c += blockDim.x) {
            double d = 0.0;
            if (usable) {
                const double u = svk[c] * rnrm;
                const double kh = (double)(float)(_Float16)(float)u;
                d = fabs(u - kh);
                if (fabs(kh) < 6.103515625e-05) d = fmax(d, fabs(u));
            }
            sdd[c] = d * d;
        }
        __syncthreads();
        float kerr = -1.0f;
        if (threadIdx.x < kWave) {                                        // the first wave adds them up (any order: a bound, rounded up)
            double e2 = 0.0;
            for (int c = (int)threadIdx.x; c < S; c += kWave) e2 = e2 + sdd[c];
            e2 = wave_sum_f64(e2);
            if (usable) kerr = __double2float_ru(sqrt(e2) * (1.0 + 1e-6) + 1e-10);   // (> 0 always: 0 would read as "not recorded")
        }
        if (threadIdx.x == 0) {
            *reinterpret_cast<float *>(hk + SK) = usable ? (float)nrm : -1.0f;   // the filter's range check (negative: no decision)
            *reinterpret_cast<float *>(hd + SK) = usable ? (float)nrm : -1.0f;
            *reinterpret_cast<float *>(hk + SK + 2) = kerr;
            *reinterpret_cast<float *>(hd + SK + 2) = kerr;
        }
        ING_STAMP(7);
        // the alignment image (kernels.hpp: halign_*): norm, P rotated copies of either part; database slots only
        const int P = halign_P(S), CP = halign_CP(S);
        if (halign && P && slot < cap) {
            unsigned char *img = halign + (size_t)slot * (size_t)halign_bytes(S);
            _Float16 *x = reinterpret_cast<_Float16 *>(img + 16);
            const unsigned int mCP = fastdiv_magic(CP);
            for (int i = threadIdx.x; i < P * CP; i += blockDim.x) {
                const int rho = fastdiv(i, mCP), k = i - rho * CP;
                int c = k - rho + S;                                    // (k - rho) mod S, k - rho in [-P, CP)
                c -= fastdiv(c, mS) * S;
                _Float16 v = (_Float16)0.0f, vl = (_Float16)0.0f;
                if (usable) parts(c, &v, &vl);
                x[i] = v; x[P * CP + i] = vl;
            }
            if (threadIdx.x == 0) {
                float *head = reinterpret_cast<float *>(img);
                head[0] = usable ? (float)nrm : -1.0f; head[1] = kerr; head[2] = 0.f; head[3] = 0.f;
            }
        }
    }
    ING_STAMP(8);
    // How far the fp16 unit columns of the screening copy are from the unit columns they stand for: E = sum over the columns of
    // |h_c - x_c / norm_c|_2, rounded up.  By Cauchy-Schwarz the screened cosine of a column pair is off by at most |e_q| + |e_k| +
    // |e_q||e_k|, so the screened distance of a (scan, keyframe, shift) is within (E_q + E_k)(1 + 1e-3) / n_eff (+ accumulation) of
    // the reference's -- on the ACTUAL rounding errors (2e-4 per column as a rule) instead of fp16's worst case (4.9e-4): the
    // finishing kernel's shift masks (sc_screen.hip) get their margin from it.  An entry below fp16's normal range counts with the
    // larger of its rounding error and its value (a matrix core may take it for zero).  Kept in word 6 of the sector mask, which
    // holds no sector bits on grids of up to 192 sectors.
    __syncthreads();                                                 // (svk, the sector key, is no longer read)
    ING_STAMP(10);
    // (two lanes per column where the workgroup has them -- even and odd rings, joined by one exchange --; the bound's sum by a wave
    //  reduction, the mask words by ballots: the serial forms were 6 of the kernel's 22 us)
    {
        const int TPC = 2 * S <= (int)blockDim.x ? 2 : 1;
        const int c = (int)threadIdx.x / TPC, half = (int)threadIdx.x - c * TPC;
        double e2 = 0.0;
        if (c < S) {
            const float iv = siv[c];
            if (iv == iv && iv != 0.0f) {
                const double rn = 1.0 / snorm[c];   // (x * (1 / norm): two roundings from x / norm, covered by the bound's 1e-6)
                for (int r = half; r < R; r += TPC) {
                    const float x = sv[r * LS + c];
                    const double h = (double)(float)(_Float16)(x * iv), u = (double)x * rn;
                    double d = fabs(h - u);
                    if (fabs(h) < 6.103515625e-05) d = fmax(d, fabs(u));
                    e2 = e2 + d * d;
                }
            }
        }
        if (TPC == 2) e2 += __shfl_xor(e2, 1, kWave);                  // (every lane of the workgroup takes part)
        if (c < S && half == 0) svk[c] = sqrt(e2);
    }
    const bool col = (int)threadIdx.x < S;                             // S <= 224 < blockDim: thread c looks at column c
    const float ivc = col ? siv[threadIdx.x] : 0.0f;
    const int any_nan = __syncthreads_or(col && ivc != ivc);           // (also the barrier behind the columns' bounds)
    const unsigned long long nzb = __ballot(col && ivc != 0.0f);       // NaN counts as non-zero
    const int wv = (int)(threadIdx.x >> 6), ln = (int)(threadIdx.x & 63);
    if (wv < 3) {
        if (ln == 0) { kmask[(size_t)slot * 8 + 2 * wv] = (unsigned int)nzb; kmask[(size_t)slot * 8 + 2 * wv + 1] = (unsigned int)(nzb >> 32); }
    } else if (wv == 3) {
        double E = 0.0;
        for (int cc = ln; cc < S; cc += kWave) E = E + svk[cc];
        E = wave_sum_f64(E);
        if (ln == 0) {
            kmask[(size_t)slot * 8 + 6] = S <= 192 ? (unsigned int)__float_as_int(__double2float_ru(E * (1.0 + 1e-6) + 1e-12)) : (unsigned int)nzb;
            kmask[(size_t)slot * 8 + 7] = any_nan ? 1u : 0u;
        }
    }
    ING_STAMP(9);
}

__global__ __launch_bounds__(256) void ingest_kernel(IngestArgs ia)
{
    extern __shared__ float sv_k[];
    ingest_body(ia, (int)blockIdx.x, sv_k);
}

// A group's ingest and the NEXT group's scatter in one launch (the clouds already on the device: nothing to wait for between the groups):
// the ingest is one workgroup per scan, sixteen CUs busy for as long as the scatter of sixteen clouds takes on all of them -- side by
// side they cost what the longer one costs.  Workgroups 0 .. n_ingest-1 (dispatched first: the long ones) ingest the previous group's
// tiles, the others scatter into the other set of tiles.
__global__ __launch_bounds__(256) void front_fused_kernel(ScanBatch b, int stride, double lidar_height, double max_radius, int *gtiles, int lab,
                                                          IngestArgs ia, int n_ingest)
{
    extern __shared__ float sv_f[];
    if ((int)blockIdx.x < n_ingest) ingest_body(ia, (int)blockIdx.x, sv_f);
    else scatter_body(b, (int)blockIdx.x - n_ingest, stride, ia.R, ia.S, lidar_height, max_radius, gtiles, lab, reinterpret_cast<int *>(sv_f));
}

// test hook (tests/test_gpu_make_sc.py): the device's atanf_glibc over whole blocks of 2^24 consecutive float bit patterns, reduced to the
// order-independent checksum of oracle/tools/atanf_exhaustive.c -- sum mod 2^64 of splitmix64((bits << 32) | result bits), NaN as 0x7fc00000
__global__ __launch_bounds__(256) void atanf_checksum_kernel(int first_block, unsigned long long *out)
{
    const unsigned int blk = (unsigned int)first_block + blockIdx.y;
    unsigned long long h = 0;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < (1u << 24); i += gridDim.x * blockDim.x) {
        const unsigned int bits = (blk << 24) | i;
        const float a = atanf_glibc(__int_as_float((int)bits));
        unsigned long long z = ((unsigned long long)bits << 32) | (a != a ? 0x7fc00000u : (unsigned int)__float_as_int(a));
        z += 0x9e3779b97f4a7c15ull;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        h += z ^ (z >> 31);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(&out[blockIdx.y], h);
}

// test hook (tests/test_gpu_make_sc.py): sc_bin_fast against sc_bin_exact on generated points -- out[0] += points on which the fast path
// was sure AND differed from the reference's chain (must stay 0), out[1] += points on which it was sure.
//   mode 0: uniform in the square of +-1.125 max_radius; 1: on ring boundaries, a few float steps either side; 2: on sector boundaries,
//   micro-degrees either side; 3: zeros, signed zeros, denormals, huge values, infinities, NaNs and plain values, all pairs
__global__ __launch_bounds__(256) void bin_paths_selftest_kernel(int mode, unsigned long long seed, unsigned long long n, int R, int S, double max_radius,
                                                                 unsigned long long *out)
{
    const float c_ring = (float)((double)R / max_radius), c_sect = (float)((double)S / 6.283185307179586);
    unsigned long long bad = 0, sure_n = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        unsigned long long z = seed + i * 0x9e3779b97f4a7c15ull;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; z ^= z >> 31;
        const double u0 = (double)(z >> 40) * (1.0 / 16777216.0), u1 = (double)((z >> 16) & 0xffffffull) * (1.0 / 16777216.0);
        const int kk = (int)(z & 31ull) - 16;
        float x, y;
        if (mode == 0) {
            x = (float)((2.0 * u0 - 1.0) * 1.125 * max_radius); y = (float)((2.0 * u1 - 1.0) * 1.125 * max_radius);
        } else if (mode == 1) {
            const int ring = 1 + (int)(u0 * R);
            const double r0 = (double)ring * max_radius / R, ang = 6.283185307179586 * u1;
            const float fx = (float)(r0 * cos(ang)), fy = (float)(r0 * sin(ang));
            x = __int_as_float(__float_as_int(fx) + ((kk >> 2) * (fx != 0.f))); y = __int_as_float(__float_as_int(fy) + ((kk & 3) - 1) * (fy != 0.f));
        } else if (mode == 2) {
            const int sec = (int)(u0 * (S + 1));
            const double ang = (sec * (360.0 / S) + kk * 1.0e-6) * (3.14159265358979323846 / 180.0), r0 = 0.25 + u1 * 1.05 * max_radius;
            x = (float)(r0 * cos(ang)); y = (float)(r0 * sin(ang));
        } else {
            const float tab[16] = {0.0f, -0.0f, 1.0e-42f, -1.0e-42f, 1.0e-20f, -1.0e-20f, 1.0e20f, -1.0e20f, __int_as_float(0x7f800000), __int_as_float((int)0xff800000u),
                                   __int_as_float(0x7fc00000), 1.0f, -1.0f, (float)max_radius, 3.0e-5f, -37.5f};
            x = tab[(z >> 8) & 15]; y = tab[(z >> 12) & 15];
        }
        int rf, sf, re, se; bool df, de;
        const bool sure = sc_bin_fast(x, y, R, S, c_ring, c_sect, rf, sf, df);
        sc_bin_exact(x, y, R, S, max_radius, re, se, de);
        if (sure) {
            sure_n++;
            if (df != de || (!de && (rf != re || sf != se))) bad++;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { bad += __shfl_xor(bad, off, kWave); sure_n += __shfl_xor(sure_n, off, kWave); }
    if ((threadIdx.x & (kWave - 1)) == 0) { atomicAdd(&out[0], bad); atomicAdd(&out[1], sure_n); }
}

__global__ void untile_kernel(const float4 *dslot, int R, int S, float *values)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * S) {
        const int r = i / S, c = i - r * S;
        const float *f = reinterpret_cast<const float *>(dslot + (size_t)(r >> 2) * S + c);
        values[i] = f[r & 3];
    }
}

}  // namespace

hipError_t launch_make_sc_finalize(int *tiles, int count, int R, int S, float *values, hipStream_t stream)
{
    const int cells = R * S * count;
    if (cells <= 0) return hipSuccess;
    hipLaunchKernelGGL(make_sc_finalize_kernel, dim3((cells + 255) / 256), dim3(256), 0, stream, tiles, cells, values);
    return hipGetLastError();
}

hipError_t launch_make_sc_tiles_init(int *tiles, int count, int R, int S, hipStream_t stream)
{
    const int cells = R * S * count;
    if (cells <= 0) return hipSuccess;
    hipLaunchKernelGGL(make_sc_init_kernel, dim3((cells + 255) / 256), dim3(256), 0, stream, tiles, cells);
    return hipGetLastError();
}

// scatter of a batch of clouds (device pointers in b.points, b.n; b.first_wg is filled here) into tiles[count][R*S]
hipError_t launch_make_sc_batch(ScanBatch b, int stride_bytes, int R, int S, double lidar_height, double max_radius,
                                int *tiles, int points_per_wg, int num_cu, hipStream_t stream)
{
    if (b.count <= 0) return hipSuccess;
    if (b.count > kMaxScBatch || points_per_wg < 256) return hipErrorInvalidValue;
    int total = 0;
    for (int i = 0; i < b.count; ++i) {
        b.first_wg[i] = total;
        if (b.n[i] < 0 || (b.n[i] > 0 && !b.points[i])) return hipErrorInvalidValue;
        total += (b.n[i] + points_per_wg - 1) / points_per_wg;           // an empty cloud has no workgroup: its tile stays initial
    }
    for (int i = b.count; i <= kMaxScBatch; ++i) b.first_wg[i] = total;
    if (total == 0) return hipSuccess;
    const size_t lds = sizeof(int) * (size_t)R * S;
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<bool> &attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire) && lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)make_sc_batch_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    (void)num_cu;
    hipLaunchKernelGGL(make_sc_batch_scatter_kernel, dim3(total), dim3(kScThreads), lds, stream, b, stride_bytes, R, S,
                       lidar_height, max_radius, tiles, scl_lab_int("SCL_SC_LAB", 0));
    return hipGetLastError();
}

static size_t ingest_lds_bytes(int R, int S) { return sizeof(float) * ((size_t)R * (S + 1) + S + 2) + sizeof(double) * 3 * (size_t)S; }

hipError_t launch_ingest(const float *values, int count, int first_slot,
                         float4 *desc, double *vkey, double *norm, float *rkey, float4 *rkey4,
                         uint2 *hdesc, unsigned int *kmask, unsigned short *hkey, int hstride,
                         int cap, int R, int S, hipStream_t stream, unsigned char *halign, int *tiles, float *vals_out)
{
    if (count <= 0) return hipSuccess;
    if (S > 224) return hipErrorInvalidValue;              // kmask holds 7 words of sector bits
    const size_t lds = ingest_lds_bytes(R, S);
    // (the attribute is raised to what the launch needs, not to the CU's 160 KB: the kernel has a few bytes of static LDS of its own --
    //  __syncthreads_or -- and static + dynamic must fit.  Per device; engines on different threads may race here: setting it twice is harmless)
    static std::atomic<size_t> attr_lds_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<size_t> &attr_lds = attr_lds_dev[dev_ & 63];
    if (lds > 48 * 1024 && lds > attr_lds.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)ingest_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds.store(lds, std::memory_order_release);
    }
    IngestArgs ia{values, first_slot, desc, vkey, norm, rkey, rkey4, hdesc, kmask, hkey, hstride, cap, R, S, halign, tiles, vals_out};
    hipLaunchKernelGGL(ingest_kernel, dim3(count), dim3(256), lds, stream, ia);
    return hipGetLastError();
}

// ingest of n_ingest scans (ia: tiles of the previous group) + scatter of the batch b into gtiles, one launch (either part may be empty)
hipError_t launch_front_fused(ScanBatch b, int stride_bytes, double lidar_height, double max_radius, int *gtiles, int points_per_wg,
                              const IngestArgs &ia, int n_ingest, hipStream_t stream)
{
    if (b.count < 0 || b.count > kMaxScBatch || n_ingest < 0 || n_ingest > kMaxScBatch || points_per_wg < 256 || ia.S > 224) return hipErrorInvalidValue;
    int total = 0;
    for (int i = 0; i < b.count; ++i) {
        b.first_wg[i] = total;
        if (b.n[i] < 0 || (b.n[i] > 0 && !b.points[i])) return hipErrorInvalidValue;
        total += (b.n[i] + points_per_wg - 1) / points_per_wg;
    }
    for (int i = b.count; i <= kMaxScBatch; ++i) b.first_wg[i] = total;
    if (total + n_ingest == 0) return hipSuccess;
    const size_t lds_i = n_ingest ? ingest_lds_bytes(ia.R, ia.S) : 0, lds_s = sizeof(int) * (size_t)ia.R * ia.S;
    const size_t lds = lds_i > lds_s ? lds_i : lds_s;
    static std::atomic<size_t> attr_lds_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<size_t> &attr_lds = attr_lds_dev[dev_ & 63];
    if (lds > 48 * 1024 && lds > attr_lds.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)front_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_lds.store(lds, std::memory_order_release);
    }
    hipLaunchKernelGGL(front_fused_kernel, dim3(total + n_ingest), dim3(256), lds, stream, b, stride_bytes, lidar_height, max_radius, gtiles,
                       scl_lab_int("SCL_SC_LAB", 0), ia, n_ingest);
    return hipGetLastError();
}

#ifdef SCL_DIAGNOSTICS
void ingest_stamps_print()
{
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ingest_stamps), sizeof h) != hipSuccess || !h[15]) return;
    static const char *names[11] = {"load + finalize", "tiled copy", "sector key + norms", "ring key", "barrier", "hdesc", "h2 image", "unit key, kerr", "halign", "E sum + masks", "E columns' barrier"};
    fprintf(stderr, "ingest stamps (workgroup 0, %llu launches, us per launch):", h[15]);
    for (int k = 0; k < 11; ++k) fprintf(stderr, " %s %.2f;", names[k], (double)h[k] / (double)h[15] / 100.0);
    fprintf(stderr, "\n");
}
#endif

hipError_t launch_atanf_block_checksums(int first_block, int n_blocks, unsigned long long *d_out, hipStream_t stream)
{
    if (first_block < 0 || n_blocks < 1 || first_block + n_blocks > 256) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(d_out, 0, sizeof(unsigned long long) * (size_t)n_blocks, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(atanf_checksum_kernel, dim3(64, n_blocks), dim3(256), 0, stream, first_block, d_out);
    return hipGetLastError();
}

hipError_t launch_bin_paths_selftest(int mode, unsigned long long seed, unsigned long long n, int R, int S, double max_radius, unsigned long long *d_out2, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(d_out2, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bin_paths_selftest_kernel, dim3(2048), dim3(256), 0, stream, mode, seed, n, R, S, max_radius, d_out2);
    return hipGetLastError();
}

hipError_t launch_untile(const float4 *desc_slot, int R, int S, float *values, hipStream_t stream)
{
    const int cells = R * S;
    hipLaunchKernelGGL(untile_kernel, dim3((cells + 255) / 256), dim3(256), 0, stream, desc_slot, R, S, values);
    return hipGetLastError();
}

}  // namespace scl
