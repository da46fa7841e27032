// device_sort.hip -- stable LSD radix sort of (key, value) pairs and prefix sums for the verification path (gfx950).
//
// What is sorted there: PCL's VoxelGrid groups a cloud's points by voxel (DM.h:1183-1185, 1200-1201: the 26 submaps of a query are
// 2.6e6 (voxel, point) pairs in 26 segments of one buffer), the ICP batch takes its sources along a Hilbert curve (1e5 pairs, 30-bit
// keys).  Sizes of 1e5 .. 3e6 pairs and keys of 30 .. 40 bits: few passes matter more than anything else, so a pass takes up to
// ELEVEN bits (2 048 bins) and a sort is three passes for keys of up to 33 bits -- the segments of the batch are sorted each on
// its own 32 voxel bits instead of all together on (job, voxel), 37 bits.
//
// One pass = three launches over tiles of 256 x ITEMS pairs (a tile never straddles a segment):
//   sort_hist_kernel     the tile's histogram of the digit (LDS atomics) -> row `tile` of H[tile][bin]
//   sort_colscan_kernel  exclusive prefix sum down every column of H (16 columns per workgroup: whole 64-byte pieces of every row),
//                        the totals in row `tiles`
//   sort_scatter_kernel  a pair's place = its segment's start + the pairs of its segment with a smaller digit + the pairs with the same
//                        digit in earlier tiles of the segment (all three from E = the scanned H: rows tile, first and end tile of the
//                        segment) + its rank among the tile's pairs with the same digit.  The rank is what keeps the sort stable: a
//                        wave takes 64 consecutive pairs per round, the lanes that hold the same digit find each other with one ballot
//                        per bit, the first of them advances the wave's counter of that digit in LDS (a row of counters per wave: no
//                        atomics, no barrier between rounds) and every one of them ranks behind the counter's old value by the number
//                        of its peers in lower lanes; the waves' rows are then offset against each other.
// The number of passes is odd, so that the pairs end in the caller's output arrays with the input arrays as the second buffer.
//
// Prefix sums (cell counts -> cell starts, voxel heads -> output positions): two launches -- per-chunk totals, then every chunk scanned
// from the sum of the totals before it (at most 1 024 chunks, so that sum is one read per thread).
#include "device_sort.hpp"

#include <cstdint>

namespace scl {

namespace {

constexpr int kSortThreads = 256, kSortWaves = 4, kSortWave = 64;
constexpr int kSortMaxDigit = 11;                // bits per pass at most
constexpr int kSortBigItems = 16, kSortSmallItems = 4;
constexpr size_t kSortBigN = (size_t)1 << 20;    // from here on tiles of 4 096 pairs, below of 1 024 (more workgroups for the small sorts)

struct SortPlan {                                // passed by value: lives in the kernels' argument segment
    int nseg;
    int off[kSortMaxSegments + 1];               // first element of each segment; off[nseg] = n
    int blk[kSortMaxSegments + 1];               // first tile of each segment; blk[nseg] = tiles
    unsigned int hi[kSortMaxSegments];           // packed passes: the high word every key of the segment has
};

// the segment tile b belongs to: its elements [e0, e1), its tiles [b0, b1).  Static indices only: the plan stays in scalar registers.
__device__ __forceinline__ void sort_locate(const SortPlan &pl, const int b, int &e0, int &e1, int &b0, int &b1, unsigned int *hi = nullptr)
{
    e0 = pl.off[0]; e1 = pl.off[1]; b0 = pl.blk[0]; b1 = pl.blk[1];
    unsigned int h = pl.hi[0];
    if (pl.nseg > 1) {
#pragma unroll
        for (int s = 1; s < kSortMaxSegments; ++s) {
            const bool here = s < pl.nseg && b >= pl.blk[s];
            e0 = here ? pl.off[s] : e0; e1 = here ? pl.off[s + 1] : e1;
            b0 = here ? pl.blk[s] : b0; b1 = here ? pl.blk[s + 1] : b1;
            h = here ? pl.hi[s] : h;
        }
    }
    if (hi) *hi = h;
}

template <typename K, int RB, int ITEMS>
__global__ __launch_bounds__(kSortThreads) void sort_hist_kernel(const K *keys, const SortPlan pl, const int shift, const unsigned int mask, int *H)
{
    constexpr int NB = 1 << RB;
    __shared__ int h[NB];
    const int b = blockIdx.x, t = threadIdx.x;
    int e0, e1, b0, b1;
    sort_locate(pl, b, e0, e1, b0, b1);
    for (int d = t; d < NB; d += kSortThreads) h[d] = 0;
    __syncthreads();
    const int tile0 = e0 + (b - b0) * (kSortThreads * ITEMS);
    K kv[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {                            // (every load in flight: clamped positions, masked below)
        const int idx = tile0 + u * kSortThreads + t;
        kv[u] = keys[idx < e1 ? idx : e1 - 1];
    }
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const int idx = tile0 + u * kSortThreads + t;
        if (idx < e1) atomicAdd(&h[(unsigned int)(kv[u] >> shift) & mask], 1);
    }
    __syncthreads();
    for (int d = t; d < NB; d += kSortThreads) H[(size_t)b * NB + d] = h[d];
}

// exclusive prefix sums down the columns of H[tiles + 1][nbins] (row `tiles` receives the totals; what it held is not read)
constexpr int kColRows = 896, kColPitch = 17;
__global__ __launch_bounds__(kSortThreads) void sort_colscan_kernel(int *H, const int tiles, const int nbins)
{
    __shared__ int tile[kColRows * kColPitch];
    __shared__ int part[16][kColPitch];
    const int t = threadIdx.x, dd = t & 15, bl = t >> 4;
    const int col = blockIdx.x * 16 + dd;
    int carry = 0;
    for (int r0 = 0; r0 < tiles + 1; r0 += kColRows) {
        const int nr = tiles + 1 - r0 < kColRows ? tiles + 1 - r0 : kColRows;
#pragma unroll 8
        for (int r = bl; r < nr; r += 16) {
            const int row = r0 + r;
            const int v = H[(size_t)(row < tiles ? row : tiles - 1) * nbins + col];
            tile[r * kColPitch + dd] = row < tiles ? v : 0;
        }
        __syncthreads();
        const int rpl = (nr + 15) / 16, ra = bl * rpl, rb = (ra + rpl < nr) ? ra + rpl : nr;
        int run = 0;
        for (int r = ra; r < rb; ++r) { const int v = tile[r * kColPitch + dd]; tile[r * kColPitch + dd] = run; run += v; }
        part[bl][dd] = run;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int v = part[k][dd]; before += k < bl ? v : 0; total += v; }
        __syncthreads();
        part[bl][dd] = before + carry;
        __syncthreads();
#pragma unroll 8
        for (int r = bl; r < nr; r += 16) H[(size_t)(r0 + r) * nbins + col] = tile[r * kColPitch + dd] + part[r / rpl][dd];
        carry += total;
        __syncthreads();
    }
}

// MODE (64-bit keys whose high word is the same throughout a segment, sorted on their low word): 0 keys and values in arrays of their own,
// in and out; 1 in: as 0, out: ONE 8-byte record per pair (value << 32 | low word of the key); 2 records in and out; 3 records in, out as
// 0 with the segment's high word back in the key.  A pair of a middle pass is then one 8-byte write where it was an 8- and a 4-byte
// write to two arrays -- with 2 048 destinations per tile of 4 096 every write is a transaction of its own.
template <typename K, int RB, int ITEMS, int MODE = 0>
__global__ __launch_bounds__(kSortThreads) void sort_scatter_kernel(const K *kin, K *kout, const unsigned int *vin, unsigned int *vout,
                                                                    const SortPlan pl, const int shift, const unsigned int mask, const int *E)
{
    static_assert(MODE == 0 || sizeof(K) == 8, "packed records are 64-bit keys'");
    constexpr int NB = 1 << RB, DPT = NB / kSortThreads;
    static_assert(RB >= 8 && RB <= kSortMaxDigit, "digit");
    __shared__ int cnt[kSortWaves][NB];
    __shared__ int wtot[kSortWaves];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & (kSortWave - 1), w = t >> 6;
    int e0, e1, b0, b1;
    unsigned int seg_hi = 0u;
    sort_locate(pl, b, e0, e1, b0, b1, &seg_hi);
    // the tile in the order that has to survive: wave w takes pairs [w * 64 ITEMS, (w + 1) * 64 ITEMS), round u the next 64 of them
    const int wave0 = e0 + (b - b0) * (kSortThreads * ITEMS) + w * (kSortWave * ITEMS);
    K kv[ITEMS];
    unsigned int vv[ITEMS];
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const int idx = wave0 + u * kSortWave + lane, ic = idx < e1 ? idx : e1 - 1;
        kv[u] = kin[ic];
        if constexpr (MODE <= 1) vv[u] = vin[ic]; else vv[u] = 0u;
    }
    // this thread's DPT consecutive digits of the three rows of E
    int eb[DPT], es0[DPT], es1[DPT];
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const int d = t * DPT + j;
        eb[j] = E[(size_t)b * NB + d]; es0[j] = E[(size_t)b0 * NB + d]; es1[j] = E[(size_t)b1 * NB + d];
    }
    for (int d = t; d < kSortWaves * NB; d += kSortThreads) (&cnt[0][0])[d] = 0;
    __syncthreads();
    int rank[ITEMS];
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        const bool valid = wave0 + u * kSortWave + lane < e1;
        const unsigned int digit = (unsigned int)(kv[u] >> shift) & mask;
        unsigned long long peers = __builtin_amdgcn_ballot_w64(valid);
#pragma unroll
        for (int bit = 0; bit < RB; ++bit) {
            const bool one = (digit >> bit) & 1u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(one);
            peers &= one ? m : ~m;
        }
        const int leader = __ffsll((long long)peers) - 1;        // (a valid lane is its own peer: never -1 where it is used)
        int old = 0;
        if (valid && lane == leader) { old = cnt[w][digit]; cnt[w][digit] = old + __popcll(peers); }
        old = __shfl(old, leader & (kSortWave - 1), kSortWave);
        rank[u] = old + __popcll(peers & below);
    }
    __syncthreads();
    // where each digit's pairs of this tile start: segment start + smaller digits of the segment + the same digit in earlier tiles
    int tot[DPT], sum = 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) { tot[j] = es1[j] - es0[j]; sum += tot[j]; }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < kSortWave; off <<= 1) { const int o = __shfl_up(incl, off, kSortWave); if (lane >= off) incl += o; }
    if (lane == kSortWave - 1) wtot[w] = incl;
    __syncthreads();
    int base = e0 + incl - sum;
#pragma unroll
    for (int k = 0; k < kSortWaves; ++k) base += k < w ? wtot[k] : 0;
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
        const int d = t * DPT + j;
        const int o = base + (eb[j] - es0[j]);
        base += tot[j];
        const int c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d];
        cnt[0][d] = o; cnt[1][d] = o + c0; cnt[2][d] = o + c0 + c1; cnt[3][d] = o + c0 + c1 + c2;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
        if (wave0 + u * kSortWave + lane < e1) {
            const unsigned int digit = (unsigned int)(kv[u] >> shift) & mask;
            const int pos = cnt[w][digit] + rank[u];
            if constexpr (MODE == 0) { kout[pos] = kv[u]; vout[pos] = vv[u]; }
            else if constexpr (MODE == 1) kout[pos] = (K)(((unsigned long long)vv[u] << 32) | ((unsigned long long)kv[u] & 0xffffffffull));
            else if constexpr (MODE == 2) kout[pos] = kv[u];
            else { kout[pos] = (K)(((unsigned long long)seg_hi << 32) | ((unsigned long long)kv[u] & 0xffffffffull)); vout[pos] = (unsigned int)((unsigned long long)kv[u] >> 32); }
        }
    }
}

inline int sort_items(size_t n) { return n >= kSortBigN ? kSortBigItems : kSortSmallItems; }

template <typename K, int RB, int ITEMS>
hipError_t sort_pass(const K *kin, K *kout, const unsigned int *vin, unsigned int *vout, const SortPlan &pl, int tiles, int shift,
                     unsigned int mask, int *H, hipStream_t stream, int mode)
{
    constexpr int NB = 1 << RB;
    hipLaunchKernelGGL((sort_hist_kernel<K, RB, ITEMS>), dim3(tiles), dim3(kSortThreads), 0, stream, kin, pl, shift, mask, H);
    hipLaunchKernelGGL(sort_colscan_kernel, dim3(NB / 16), dim3(kSortThreads), 0, stream, H, tiles, NB);
    const dim3 g(tiles), t(kSortThreads);
    if constexpr (sizeof(K) == 8) {
        if (mode == 1) hipLaunchKernelGGL((sort_scatter_kernel<K, RB, ITEMS, 1>), g, t, 0, stream, kin, kout, vin, vout, pl, shift, mask, (const int *)H);
        else if (mode == 2) hipLaunchKernelGGL((sort_scatter_kernel<K, RB, ITEMS, 2>), g, t, 0, stream, kin, kout, vin, vout, pl, shift, mask, (const int *)H);
        else if (mode == 3) hipLaunchKernelGGL((sort_scatter_kernel<K, RB, ITEMS, 3>), g, t, 0, stream, kin, kout, vin, vout, pl, shift, mask, (const int *)H);
        else hipLaunchKernelGGL((sort_scatter_kernel<K, RB, ITEMS, 0>), g, t, 0, stream, kin, kout, vin, vout, pl, shift, mask, (const int *)H);
    } else {
        hipLaunchKernelGGL((sort_scatter_kernel<K, RB, ITEMS, 0>), g, t, 0, stream, kin, kout, vin, vout, pl, shift, mask, (const int *)H);
    }
    return hipGetLastError();
}

template <typename K, int ITEMS>
hipError_t sort_pass_rb(int rb, const K *kin, K *kout, const unsigned int *vin, unsigned int *vout, const SortPlan &pl, int tiles, int shift,
                        unsigned int mask, int *H, hipStream_t stream, int mode)
{
    switch (rb) {
    case 11: return sort_pass<K, 11, ITEMS>(kin, kout, vin, vout, pl, tiles, shift, mask, H, stream, mode);
    case 10: return sort_pass<K, 10, ITEMS>(kin, kout, vin, vout, pl, tiles, shift, mask, H, stream, mode);
    case 9: return sort_pass<K, 9, ITEMS>(kin, kout, vin, vout, pl, tiles, shift, mask, H, stream, mode);
    default: return sort_pass<K, 8, ITEMS>(kin, kout, vin, vout, pl, tiles, shift, mask, H, stream, mode);
    }
}

template <typename K>
hipError_t sort_run(void *scratch, K *keys_in, K *keys_out, unsigned int *vals_in, unsigned int *vals_out, const SortSegments &seg, int bits,
                    hipStream_t stream, const unsigned int *seg_hi = nullptr)
{
    if (seg.nseg < 1 || seg.nseg > kSortMaxSegments || bits < 0 || bits > (int)(8 * sizeof(K)) || seg.off[0] != 0) return hipErrorInvalidValue;
    const int n = seg.off[seg.nseg];
    if (n < 0) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    if (!scratch || !keys_in || !keys_out || !vals_in || !vals_out) return hipErrorInvalidValue;
    const int items = sort_items((size_t)n), tile = kSortThreads * items;
    SortPlan pl{};
    pl.nseg = seg.nseg;
    int tiles = 0;
    for (int s = 0; s < seg.nseg; ++s) {
        if (seg.off[s + 1] < seg.off[s]) return hipErrorInvalidValue;
        pl.off[s] = seg.off[s]; pl.blk[s] = tiles;
        tiles += (seg.off[s + 1] - seg.off[s] + tile - 1) / tile;
    }
    pl.off[seg.nseg] = n; pl.blk[seg.nseg] = tiles;
    for (int s = seg.nseg + 1; s <= kSortMaxSegments; ++s) { pl.off[s] = n; pl.blk[s] = tiles; }
    for (int s = 0; s < seg.nseg; ++s) pl.hi[s] = seg_hi ? seg_hi[s] : 0u;
    // an odd number of passes of at most eleven bits
    int passes = (bits + kSortMaxDigit - 1) / kSortMaxDigit;
    if (passes < 1) passes = 1;
    if (!(passes & 1)) ++passes;
    const int digit = (bits + passes - 1) / passes, rb = digit < 8 ? 8 : digit;
    int *H = static_cast<int *>(scratch);
    for (int p = 0; p < passes; ++p) {
        const int shift = p * digit, left = bits - shift, wbits = left < 0 ? 0 : (left < digit ? left : digit);
        const unsigned int mask = wbits ? (1u << wbits) - 1u : 0u;
        const K *kin = (p & 1) ? keys_out : keys_in; K *kout = (p & 1) ? keys_in : keys_out;
        const unsigned int *vin = (p & 1) ? vals_out : vals_in; unsigned int *vout = (p & 1) ? vals_in : vals_out;
        const int sh = shift < (int)(8 * sizeof(K)) ? shift : 0;                 // (a pass with no bits left: mask 0, a stable copy)
        // 64-bit keys that differ in their low word only within a segment: 8-byte records between the first and the last pass
        const bool packed = sizeof(K) == 8 && seg_hi && bits <= 32 && passes >= 3;
        const int mode = !packed ? 0 : (p == 0 ? 1 : (p == passes - 1 ? 3 : 2));
        hipError_t e = items == kSortBigItems ? sort_pass_rb<K, kSortBigItems>(rb, kin, kout, vin, vout, pl, tiles, sh, mask, H, stream, mode)
                                              : sort_pass_rb<K, kSortSmallItems>(rb, kin, kout, vin, vout, pl, tiles, sh, mask, H, stream, mode);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- prefix sums ---------------------------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256, kScanTile = 4 * kScanThreads, kScanMaxChunks = 1024;

__global__ __launch_bounds__(kScanThreads) void scan_totals_kernel(const int *in, const int n, const int chunk, int *part)
{
    __shared__ int ws[kScanThreads / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int a = blockIdx.x * chunk, b = a + chunk < n ? a + chunk : n;
    int s = 0;
    for (int i0 = a; i0 < b; i0 += kScanTile) {
        const int i = i0 + 4 * t;
        if (i + 3 < b) { const int4 v = *reinterpret_cast<const int4 *>(in + i); s += (v.x + v.y) + (v.z + v.w); }
        else for (int k = i; k < b; ++k) s += in[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) ws[w] = s;
    __syncthreads();
    if (t == 0) part[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

__global__ __launch_bounds__(kScanThreads) void scan_apply_kernel(const int *in, int *out, const int n, const int chunk, const int *part, const int inclusive)
{
    __shared__ int ws[kScanThreads / 64];
    __shared__ int wt[kScanThreads / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int before = 0;
#pragma unroll
    for (int k0 = 0; k0 < kScanMaxChunks; k0 += kScanThreads) { const int k = k0 + t; before += k < (int)blockIdx.x ? part[k] : 0; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    if (lane == 0) ws[w] = before;
    __syncthreads();
    int offset = (ws[0] + ws[1]) + (ws[2] + ws[3]);
    const int a = blockIdx.x * chunk, b = a + chunk < n ? a + chunk : n;
    for (int i0 = a; i0 < b; i0 += kScanTile) {
        const int i = i0 + 4 * t;
        int4 v = make_int4(0, 0, 0, 0);
        if (i + 3 < b) v = *reinterpret_cast<const int4 *>(in + i);
        else { if (i < b) v.x = in[i]; if (i + 1 < b) v.y = in[i + 1]; if (i + 2 < b) v.z = in[i + 2]; }
        const int mine = (v.x + v.y) + (v.z + v.w);
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
        __syncthreads();                                         // (the tile before has been read)
        if (lane == 63) wt[w] = incl;
        __syncthreads();
        int e = offset + incl - mine;
#pragma unroll
        for (int k = 0; k < kScanThreads / 64; ++k) e += k < w ? wt[k] : 0;
        const int tile_total = (wt[0] + wt[1]) + (wt[2] + wt[3]);
        int4 o;
        o.x = e + (inclusive ? v.x : 0); e += v.x;
        o.y = e + (inclusive ? v.y : 0); e += v.y;
        o.z = e + (inclusive ? v.z : 0); e += v.z;
        o.w = e + (inclusive ? v.w : 0);
        if (i + 3 < b) *reinterpret_cast<int4 *>(out + i) = o;
        else { if (i < b) out[i] = o.x; if (i + 1 < b) out[i + 1] = o.y; if (i + 2 < b) out[i + 2] = o.z; }
        offset += tile_total;
    }
}

inline int scan_chunk(size_t n)
{
    size_t c = (n + kScanMaxChunks - 1) / kScanMaxChunks;
    c = (c + kScanTile - 1) / kScanTile * kScanTile;
    return (int)(c < (size_t)kScanTile ? (size_t)kScanTile : c);
}

}  // namespace

size_t sort_scratch_bytes(size_t n, int nseg)
{
    const size_t tile = (size_t)kSortThreads * (size_t)sort_items(n);
    const size_t tiles = (n + tile - 1) / tile + (size_t)(nseg > 0 ? nseg : 1);
    return (tiles + 1) * ((size_t)1 << kSortMaxDigit) * sizeof(int);
}

hipError_t sort_pairs_u32(void *scratch, unsigned int *keys_in, unsigned int *keys_out, unsigned int *vals_in, unsigned int *vals_out,
                          int n, int bits, hipStream_t stream)
{
    SortSegments seg{};
    seg.nseg = 1; seg.off[0] = 0; seg.off[1] = n;
    return sort_run<unsigned int>(scratch, keys_in, keys_out, vals_in, vals_out, seg, bits, stream);
}

hipError_t sort_pairs_u64(void *scratch, unsigned long long *keys_in, unsigned long long *keys_out, unsigned int *vals_in,
                          unsigned int *vals_out, int n, int bits, hipStream_t stream)
{
    SortSegments seg{};
    seg.nseg = 1; seg.off[0] = 0; seg.off[1] = n;
    return sort_run<unsigned long long>(scratch, keys_in, keys_out, vals_in, vals_out, seg, bits, stream);
}

hipError_t sort_pairs_u64_segmented(void *scratch, unsigned long long *keys_in, unsigned long long *keys_out, unsigned int *vals_in,
                                    unsigned int *vals_out, const SortSegments &seg, int bits, hipStream_t stream, const unsigned int *seg_hi)
{
    return sort_run<unsigned long long>(scratch, keys_in, keys_out, vals_in, vals_out, seg, bits, stream, seg_hi);
}

size_t scan_scratch_bytes(size_t) { return sizeof(int) * (size_t)kScanMaxChunks; }

hipError_t prefix_sum_i32(void *scratch, const int *in, int *out, int n, bool inclusive, hipStream_t stream)
{
    if (n < 0) return hipErrorInvalidValue;
    if (n == 0) return hipSuccess;
    if (!scratch || !in || !out || ((uintptr_t)in & 15u) || ((uintptr_t)out & 15u)) return hipErrorInvalidValue;
    const int chunk = scan_chunk((size_t)n), chunks = (int)(((size_t)n + (size_t)chunk - 1) / (size_t)chunk);
    int *part = static_cast<int *>(scratch);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(chunks), dim3(kScanThreads), 0, stream, in, n, chunk, part);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(chunks), dim3(kScanThreads), 0, stream, in, out, n, chunk, (const int *)part, inclusive ? 1 : 0);
    return hipGetLastError();
}

}  // namespace scl
