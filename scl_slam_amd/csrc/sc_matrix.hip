// sc_matrix.hip -- the exact fp64 distance MATRIX of a batch of scans against a range of keyframes (scl_sc_distance_matrix on the
// screened grids): distanceBtnScanContext (descriptor.h:1538-1569) of EVERY pair, evaluated at the shifts the screening leaves open.
//
// Same arithmetic as sc_masked.hip (ring-order fp64 dots of widened floats -- fma == mul + add --, the quotient by the two column
// norms, the sum over the scan's sectors in ascending order, 1 - sum / n_eff, strict < over ascending shift VALUES: distance and
// shift are the reference's, bit for bit), turned round for a matrix, where every keyframe meets every scan of the batch:
//
//   * A workgroup holds NS scans in LDS (fp32, column-major, pitch 4 R + 16 B: consecutive columns on consecutive 16-byte slots) and
//     walks a RANGE of keyframes; a wave takes one keyframe at a time (tickets from an LDS counter) and scores it against ALL the
//     workgroup's scans: the keyframe's 30 KB cross L2 -> CU once per NS pairs, and its conversion to fp64 is paid once.
//   * Lane l < 60 owns the KEYFRAME's columns l + 60 j (fixed: one coalesced float4 stream per ring group, requested PD groups
//     ahead, across passes and keyframes); what changes with the pair and the shift is the scan's column, (l + 60 j + shift) mod S
//     -- an LDS address.  So a pass over the keyframe's rings serves up to P (scan, shift) SLOTS, whatever scans they belong to:
//     one accumulator per slot and column.
//   * The P similarity rows of a pass go through the wave's own LDS rows; P lanes walk them (the reference's ascending sum) at once.
//   * Workgroups are small (kr keyframes x NS scans) and ordered so that the workgroups of ONE range -- one per set of scans --
//     follow each other on one XCD (index & 7): they start together, walk the same keyframes in the same order and find them in
//     that XCD's L2.  (sc_masked_kernel's persistent workgroups drifted apart: 2.4 GB of HBM reads per 16-row launch against
//     0.31 GB of keyframes, profiles/r04/matrix_before/.)
#include <atomic>

#include "device_common.hpp"
#include "kernels.hpp"

namespace scl {

namespace {

__device__ __forceinline__ void mat_fence_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int kMatLanes = 60;                  // active lanes: S / 60 keyframe columns per lane (S = 120, 180)
// shapes (scripts/build_variant.sh overrides them for experiments): scans per workgroup, waves, slots per pass, ring groups in flight
#ifndef MAT_A
#define MAT_A 2, 16, 6, 2
#endif
#ifndef MAT_B
#define MAT_B 1, 12, 4, 2
#endif

template <int RG_, int S_, int W_, int NS_, int WAVES_, int P_, int PD_>
struct MatCfg {
    static constexpr int RG = RG_, S = S_, W = W_, NS = NS_, WAVES = WAVES_, P = P_, PD = PD_;
    static constexpr int CPL = S / kMatLanes;                      // keyframe columns per lane
    static constexpr int PITCH = RG * 16 + 16;                     // bytes per staged scan column
    static constexpr size_t LDS_Q = (size_t)NS * S * PITCH;        // the scans, fp32
    static constexpr size_t LDS_N = (size_t)NS * S * 8;            // their column norms, fp64
    static constexpr size_t LDS_WAVE = (size_t)P * S * 8;          // per wave: P rows of S similarities
    static constexpr size_t LDS = LDS_Q + LDS_N + WAVES * LDS_WAVE + 16;   // + the ticket counter
    static_assert(S % kMatLanes == 0 && CPL >= 1 && CPL <= 3 && W <= 32 && NS >= 1 && NS <= 2 && RG % PD == 0 && P >= 3 && P <= 8, "tiling");
    static_assert(LDS <= 160 * 1024, "LDS");
};

struct MatrixArgs {
    const float4 *desc; const double *norm;
    int nq, lo, n, kr, ranges;                 // scans; keyframes [lo, lo + n); keyframes per workgroup; ceil(n / kr)
    int qslot[kMaxScreenBatch];
    const int *starts; const unsigned int *smask; unsigned long long set_stride;   // of scan i at i * set_stride + position in the range
    double *out_dist; int *out_shift; unsigned long long row_stride;               // row i at i * row_stride + position
};

// The ring-order dots of one pass over the keyframe, NP live slots (compile time: no branches between the slots).  The keyframe's
// ring groups come out of the register ring, which is refilled PD groups further on -- in the last block from the stream that follows
// this pass (the same keyframe again, or the next item's) --, converted once, and meet every slot's scan columns out of LDS.  The LDS
// reads run ONE SLOT AHEAD of the products that use them (two register sets; sched_barrier pins the order: left alone, the compiler
// put every read directly in front of its products and waited for it -- a wave spent most of a pass in s_waitcnt lgkmcnt(0)).
template <class C, int NP>
__device__ __forceinline__ void mat_dots(float4 (&kbuf)[C::PD][C::CPL], double (&acc)[C::P][C::CPL], const int (&qoff_in)[C::P][C::CPL],
                                         const unsigned char *Qs, const float4 *kcur, const float4 *knext, const int (&yc)[C::CPL])
{
    constexpr int PD = C::PD, CPL = C::CPL, S = C::S, RG = C::RG;
    constexpr int NQ = NP > 0 ? NP : 1;
    static_assert((PD * NQ) % 2 == 0, "the read buffers' parity is the same at every block");
    int qoff[NQ][CPL];
#pragma unroll
    for (int u = 0; u < NP; ++u)
#pragma unroll
        for (int j = 0; j < CPL; ++j) qoff[u][j] = qoff_in[u][j];
    float4 qv[2][CPL];
    if (NP > 0) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) qv[0][j] = *reinterpret_cast<const float4 *>(Qs + qoff[0][j]);
    }
#pragma unroll 1
    for (int r0 = 0; r0 < RG; r0 += PD) {
        const float4 *src = (r0 == RG - PD) ? knext : kcur + (size_t)(r0 + PD) * S;      // (wave uniform)
#pragma unroll
        for (int dd = 0; dd < PD; ++dd) {
            double kx[CPL][4];
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const float4 kv = kbuf[dd][j];
                kx[j][0] = (double)kv.x; kx[j][1] = (double)kv.y; kx[j][2] = (double)kv.z; kx[j][3] = (double)kv.w;
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) kbuf[dd][j] = src[(size_t)dd * S + yc[j]];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                constexpr int dummy = 0; (void)dummy;
                const int cur = (dd * NP + u) & 1, nxt = cur ^ 1;
                // the reads of the slot that follows (the first slot of the next ring group behind the last; past the last ring
                // group: the column's 16 bytes of padding, unused)
                const int un = u + 1 < NP ? u + 1 : 0, off = (u + 1 < NP ? dd : dd + 1) * 16;
#pragma unroll
                for (int j = 0; j < CPL; ++j) qv[nxt][j] = *reinterpret_cast<const float4 *>(Qs + qoff[un][j] + off);
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
        // (advanced in place: as an induction variable the compiler re-added the block's offset in front of every read)
#pragma unroll
        for (int u = 0; u < NP; ++u)
#pragma unroll
            for (int j = 0; j < CPL; ++j) { qoff[u][j] += PD * 16; asm volatile("" : "+v"(qoff[u][j])); }
    }
}

template <class C>
__global__ __launch_bounds__(C::WAVES * kWave) void sc_matrix_kernel(MatrixArgs ma)
{
    constexpr int RG = C::RG, S = C::S, W = C::W, NS = C::NS, WAVES = C::WAVES, P = C::P, PD = C::PD, CPL = C::CPL, PITCH = C::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];
    unsigned char *Qs = smem_x;
    double *nql = reinterpret_cast<double *>(smem_x + C::LDS_Q);                                        // [NS][S]
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double *rows = reinterpret_cast<double *>(smem_x + C::LDS_Q + C::LDS_N + (size_t)wave * C::LDS_WAVE);   // [P][S]
    int *ticket = reinterpret_cast<int *>(smem_x + C::LDS_Q + C::LDS_N + (size_t)WAVES * C::LDS_WAVE);

    // workgroup -> (range, set of scans): the sets of one range follow each other on one XCD
    const int NG = (ma.nq + NS - 1) / NS;
    const int b = (int)blockIdx.x, xcd = b & 7, jx = b >> 3;
    const int g = jx % NG, range = (jx / NG) * 8 + xcd;
    if (range >= ma.ranges) return;
    const int s0 = g * NS;
    const int ns = ma.nq - s0 < NS ? ma.nq - s0 : NS;
    const int r_lo = range * ma.kr;
    const int r_n = ma.n - r_lo < ma.kr ? ma.n - r_lo : ma.kr;

    // ---- stage the workgroup's scans (column-major, fp32) and their norms ----
    for (int s = 0; s < ns; ++s) {
        const float4 *qd = ma.desc + (size_t)ma.qslot[s0 + s] * (size_t)(RG * S);
        for (int idx = threadIdx.x; idx < RG * S; idx += blockDim.x) {
            const int rg = idx / S, c = idx - rg * S;
            *reinterpret_cast<float4 *>(Qs + (size_t)(s * S + c) * PITCH + rg * 16) = qd[idx];
        }
        const double *qn = ma.norm + (size_t)ma.qslot[s0 + s] * S;
        for (int c = threadIdx.x; c < S; c += blockDim.x) nql[s * S + c] = qn[c];
    }
    if (threadIdx.x == 0) *ticket = WAVES;
    __syncthreads();

    const bool active = lane < kMatLanes;
    const int ll = active ? lane : kMatLanes - 1;
    int yc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) yc[j] = ll + kMatLanes * j;
    const double kInf = __longlong_as_double(0x7ff0000000000000LL);
    const size_t kf_floats4 = (size_t)(RG * S);

    int item = wave;                                                            // (wave uniform)
    if (item >= r_n) return;
    // what an item needs from memory, requested one item ahead: first shifts and masks of its pairs, the keyframe's norms
    int pf_first[NS]; unsigned int pf_mask[NS]; double pf_nk[CPL];
    auto request = [&](const int it) {
        const int pos = r_lo + it;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const size_t o = (size_t)(s0 + (s < ns ? s : 0)) * (size_t)ma.set_stride + (size_t)pos;
            pf_first[s] = ma.starts[o]; pf_mask[s] = ma.smask[o];
        }
        const double *kn = ma.norm + (size_t)(ma.lo + pos) * S;
#pragma unroll
        for (int j = 0; j < CPL; ++j) pf_nk[j] = kn[yc[j]];
    };
    request(item);
    float4 kbuf[PD][CPL];
    {
        const float4 *kd = ma.desc + (size_t)(ma.lo + r_lo + item) * kf_floats4;
#pragma unroll
        for (int d = 0; d < PD; ++d)
#pragma unroll
            for (int j = 0; j < CPL; ++j) kbuf[d][j] = kd[(size_t)d * S + yc[j]];
    }

    while (item < r_n) {
        // the next item's ticket (LDS counter), fetched now so that its loads can be requested during this item's last pass
        int next_item = 0;
        if (lane == 0) next_item = atomicAdd(ticket, 1);
        const int pos = r_lo + item;
        const float4 *kcur = ma.desc + (size_t)(ma.lo + pos) * kf_floats4;
        int first[NS]; unsigned long long work = 0ull;
        double nk[CPL];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            int f = __builtin_amdgcn_readfirstlane(pf_first[s]);
            unsigned int m = (unsigned int)__builtin_amdgcn_readfirstlane((int)pf_mask[s]);
            const unsigned int all = W >= 32 ? 0xffffffffu : ((1u << W) - 1u);
            if (f < 0 || f >= S) { f = 0; m = 0u; }                             // (an undecided alignment never reaches this kernel: guard only)
            if (s >= ns) m = 0u;
            first[s] = f;
            work |= (unsigned long long)(m & all) << (32 * s);
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) nk[j] = pf_nk[j];
        next_item = __builtin_amdgcn_readfirstlane(next_item);
        const bool has_next = next_item < r_n;
        const float4 *knext_item = ma.desc + (size_t)(ma.lo + r_lo + (has_next ? next_item : item)) * kf_floats4;

        double best[NS]; int bshift[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { best[s] = kInf; bshift[s] = 0x7fffffff; }

        bool requested = false;
        do {
            // ---- the next up to P open (scan, shift) slots of the item ----
            int sl_scan[P], sl_shift[P], np = 0;
#pragma unroll
            for (int u = 0; u < P; ++u) {
                sl_scan[u] = 0; sl_shift[u] = 0;
                if (work) {
                    const int bit = __ffsll((long long)work) - 1;
                    work &= work - 1;
                    const int s = bit >> 5, t = bit & 31;
                    int st = first[NS == 1 ? 0 : s] + t; st = st >= S ? st - S : st;
                    sl_scan[u] = s; sl_shift[u] = st; np = u + 1;
                }
            }
            const bool last_pass = work == 0ull;
            const float4 *knext = last_pass ? knext_item : kcur;                // the stream that follows this pass
            int qoff[P][CPL];
            double acc[P][CPL];
#pragma unroll
            for (int u = 0; u < P; ++u)
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    int x = yc[j] + sl_shift[u]; x = x >= S ? x - S : x;        // the scan's column that meets keyframe column yc[j] at this shift
                    qoff[u][j] = (sl_scan[u] * S + x) * PITCH;
                    acc[u][j] = 0.0;
                }
            // ---- ring-order dots ----
            switch (np) {
            case 0: mat_dots<C, 0>(kbuf, acc, qoff, Qs, kcur, knext, yc); break;
            case 1: mat_dots<C, 1>(kbuf, acc, qoff, Qs, kcur, knext, yc); break;
            case 2: mat_dots<C, 2>(kbuf, acc, qoff, Qs, kcur, knext, yc); break;
            case 3: mat_dots<C, 3>(kbuf, acc, qoff, Qs, kcur, knext, yc); break;
            case 4: mat_dots<C, (P >= 4 ? 4 : P)>(kbuf, acc, qoff, Qs, kcur, knext, yc); break;
            case 5: mat_dots<C, (P >= 5 ? 5 : P)>(kbuf, acc, qoff, Qs, kcur, knext, yc); break;
            case 6: mat_dots<C, (P >= 6 ? 6 : P)>(kbuf, acc, qoff, Qs, kcur, knext, yc); break;
            case 7: mat_dots<C, (P >= 7 ? 7 : P)>(kbuf, acc, qoff, Qs, kcur, knext, yc); break;
            default: mat_dots<C, P>(kbuf, acc, qoff, Qs, kcur, knext, yc); break;
            }
            // the next item's first shifts, masks and norms: requested here, in front of the pass's tail (its LDS round trips and serial
            // sums cover the loads; requested in front of the dots they held eight registers through the loop that has none to spare)
            asm volatile("" ::: "memory");                                       // (the loads below stay below the dots)
            if (last_pass && has_next && !requested) { request(next_item); requested = true; }
            // ---- cosine similarity per (slot, scan column), into the wave's rows by scan column; effective columns per slot ----
            mat_fence_lds();
            int eff[P];
#pragma unroll
            for (int u = 0; u < P; ++u) {
                int e = 0;
                if (u < np) {
                    // (the scan's columns are formed again from an opaque copy of the shift: shared with the ones in front of the dots
                    // they would stay in twelve registers through the loop)
                    int shu = sl_shift[u];
                    asm volatile("" : "+s"(shu));
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        int c = yc[j] + shu; c = c >= S ? c - S : c;
                        const double nqc = nql[sl_scan[u] * S + c];
                        const bool skip = (nqc == 0.0) | (nk[j] == 0.0);        // D.h:1523
                        // a skipped column contributes +0.0 to the sum, which is bit-equivalent to leaving it out
                        if (active) rows[u * S + c] = skip ? 0.0 : acc[u][j] / (nqc * nk[j]);
                        e += __popcll(__builtin_amdgcn_ballot_w64(active && !skip));
                    }
                }
                eff[u] = e;
            }
            mat_fence_lds();
            // ---- the sum over the scan's columns in ascending order (D.h:1518-1532), one lane per slot ----
            double d = kInf; int sh = 0x7fffffff;
            if (lane < np) {
                const double *row = rows + lane * S;
                double sum = 0.0;
#pragma unroll 8
                for (int c = 0; c < S; ++c) sum = sum + row[c];
                int e = eff[0], st = sl_shift[0];
#pragma unroll
                for (int u = 1; u < P; ++u) if (lane == u) { e = eff[u]; st = sl_shift[u]; }
                const double dd = 1.0 - sum / (double)e;                         // 0 / 0 -> NaN, never wins
                if (dd < kBigDist) { d = dd; sh = st; }
            }
            // per scan: smallest distance, ties to the lowest shift VALUE (the reference walks the sorted shift space with strict <)
#pragma unroll
            for (int u = 0; u < P; ++u) {
                if (u < np) {
                    const double du = readlane_f64(d, u); const int su = __builtin_amdgcn_readlane(sh, u);
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const bool take = (sl_scan[u] == s) & ((du < best[s]) | ((du == best[s]) & (su < bshift[s])));
                        best[s] = take ? du : best[s]; bshift[s] = take ? su : bshift[s];
                    }
                }
            }
        } while (work);
        if (has_next && !requested) request(next_item);                          // (an item without an open shift)
        if (lane == 0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s < ns) {
                    const bool ok = best[s] < kBigDist;
                    const size_t o = (size_t)(s0 + s) * (size_t)ma.row_stride + (size_t)pos;
                    ma.out_dist[o] = ok ? best[s] : kBigDist;
                    ma.out_shift[o] = ok ? bshift[s] : 0;
                }
            }
        }
        item = has_next ? next_item : r_n;
    }
}

template <class C>
hipError_t launch_matrix_t(const MatrixArgs &ma_in, int nq, hipStream_t stream)
{
    static std::atomic<bool> attr_set_dev[64];
    int dev_ = 0; (void)hipGetDevice(&dev_);
    std::atomic<bool> &attr_set = attr_set_dev[dev_ & 63];
    if (!attr_set.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute((const void *)sc_matrix_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
        if (e != hipSuccess) return e;
        attr_set.store(true, std::memory_order_release);
    }
    MatrixArgs ma = ma_in;
    const int NG = (nq + C::NS - 1) / C::NS;
    const int grid = 8 * ((ma.ranges + 7) / 8) * NG;
    hipLaunchKernelGGL((sc_matrix_kernel<C>), dim3(grid), dim3(C::WAVES * kWave), C::LDS, stream, ma);
    return hipGetLastError();
}

}  // namespace

bool sc_matrix_supported(const DbView &db, int SR)
{
    const int W = 2 * SR + 1;
    return (db.RG == 16 && db.S == 120 && W == 13) || (db.RG == 20 && db.S == 180 && W == 19);
}

hipError_t launch_sc_matrix(const DbView &db, int SR, const int *qslots, int nq, int lo, int n, const int *starts, const unsigned int *smask,
                            size_t set_stride, double *out_dist, int *out_shift, size_t row_stride, int kr, hipStream_t stream)
{
    if (nq < 1 || nq > kMaxScreenBatch || n < 1 || !sc_matrix_supported(db, SR) || !starts || !smask) return hipErrorInvalidValue;
    MatrixArgs ma{};
    ma.desc = db.desc; ma.norm = db.norm; ma.nq = nq; ma.lo = lo; ma.n = n;
    for (int i = 0; i < nq; ++i) ma.qslot[i] = qslots[i];
    ma.starts = starts; ma.smask = smask; ma.set_stride = set_stride;
    ma.out_dist = out_dist; ma.out_shift = out_shift; ma.row_stride = row_stride;
    if (db.S == 120) {
        using C = MatCfg<16, 120, 13, MAT_A>;
        ma.kr = kr > 0 ? kr : 64; if (ma.kr < C::WAVES) ma.kr = C::WAVES;
        ma.ranges = (n + ma.kr - 1) / ma.kr;
        return launch_matrix_t<C>(ma, nq, stream);
    }
    using C = MatCfg<20, 180, 19, MAT_B>;
    ma.kr = kr > 0 ? kr : 64; if (ma.kr < C::WAVES) ma.kr = C::WAVES;
    ma.ranges = (n + ma.kr - 1) / ma.kr;
    return launch_matrix_t<C>(ma, nq, stream);
}

}  // namespace scl
