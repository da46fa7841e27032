// kernels.hpp -- host-side launch interface of the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

namespace scl {

// ---- switches of experiments ---------------------------------------------------------------------------------------------------
// A product build (plain `make`) reads three environment variables, all of which select shipped, tested paths: SCL_SCREEN (0: the
// exact kernel on every pair), SCL_SCREEN_FORM (1: the products' first form on every batch), SCL_SCREEN_V2_MIN (smallest batch the
// second form scores) -- plus SCL_RCCL_LIB (which collective library the sharded front loads).  Every other SCL_* switch selects a
// measured alternative or an ablation and exists only in a diagnostics build (make EXTRA=-DSCL_DIAGNOSTICS): here it is its default,
// a constant, and the code it would select is not compiled in.
#ifdef SCL_DIAGNOSTICS
inline int scl_lab_int(const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; }
inline bool scl_lab_is(const char *name, const char *value) { const char *e = getenv(name); if (!e) return false; while (*value) if (*e++ != *value++) return false; return true; }
#else
constexpr int scl_lab_int(const char *, int dflt) { return dflt; }
constexpr bool scl_lab_is(const char *, const char *) { return false; }
#endif

constexpr double kBigDist = 10000000.0;
// The screening pass keeps, kTminEpsOffset words behind a buffer set's t_min word (the smallest screened distance of the launch), the
// LARGEST per-pair bound of the launch as float bits: eps_max = max over the screened pairs of (E_q + E_k)(1 + 2e-3) / n_eff + accumulation
// (sc_screen.hip).  Every screened distance of the launch is within eps_max of the reference's, so the keyframes that can hold the minimum
// are those within 2 eps_max of the smallest -- not 2 kScreenEps (the worst case of fp16 rounding).  Zero: nothing recorded.
constexpr int kTminEpsOffset = 256;   // D.h:1494,1556,1637,1705 initial minima
constexpr int hdesc_sector(int RG) { return ((RG * 8 + 63) / 64) * 8; }   // 8-byte elements per sector of hdesc: ring groups padded to whole 64-byte k-steps
constexpr int hkey_halfs(int S) { return ((S + 31) / 32) * 32; }         // the unit-norm fp16 sector key behind the copy, zero padded to whole k-steps
constexpr int hkey_store_halfs(int S) { return hkey_halfs(S) + 8; }      // ... followed by the key's norm as a float (and 12 spare bytes)
constexpr int hkey_row_halfs(int S) { return 2 * hkey_halfs(S) + 8; }    // a row of the dense key table: the same, then the key's SECOND fp16 part fp16(2^11 (k - kh)), zero padded alike (stage 2 of the alignment's second form)
// The alignment image of a keyframe (second form of the alignment, sc_screen.hip): P copies of the unit fp16 sector key, copy rho
// rotated right by rho sectors -- X_rho[i] = key[(i - rho) mod S], i in [0, CP) with CP >= S + 8 (the first sectors repeat behind
// the last: eight consecutive entries never wrap).  Layout of a slot: 16 bytes (the key's norm as a float, 12 spare), the P copies
// of the key's first fp16 part kh, the P copies of its second part fp16(2^11 (k - kh)).  With P | S the correlation's matrix-core
// operand (row sigma, phase rho: shift P sigma + rho) is whole, aligned chunks of P halfs of copy rho: P = 8 at S = 120 (16 B per
// chunk), 4 at S = 180 (8 B).  0 bytes: no image for this grid.
constexpr int halign_P(int S) { return S % 8 == 0 ? 8 : (S % 4 == 0 ? 4 : 0); }
constexpr int halign_CP(int S) { return ((S + 8 + 31) / 32) * 32; }
constexpr int halign_img_bytes(int S) { return halign_P(S) * halign_CP(S) * 2; }                      // one part
constexpr int halign_bytes(int S) { return halign_P(S) ? 16 + 2 * halign_img_bytes(S) : 0; }
// A second image of the copy sits behind the key, for the second form of the screening products (sc_screen.hip):
// chunk-major -- [ring part h of 32 rings][16-byte chunk j][sector 0 .. S+15] x 16 B (rings 32h + 8j .. + 7 of sector s mod S), so
// that the 16 consecutive sectors a matrix-core fragment needs per chunk are 256 consecutive bytes; offsets in 8-byte elements,
// 128-byte aligned
constexpr int hdesc2_offset_rgh(int RGH, int S) { return ((RGH * S + hkey_store_halfs(S) / 4 + 15) / 16) * 16; }
constexpr int hdesc2_elems_rgh(int RGH, int S) { return RGH % 8 == 0 ? (RGH / 8) * 8 * (S + 16) : 0; }   // (RGH / 8 ring parts of 32) x 4 chunks x (S + 16) sectors x 16 B
constexpr int hdesc_stride_rgh(int RGH, int S) { return hdesc2_elems_rgh(RGH, S) ? ((hdesc2_offset_rgh(RGH, S) + hdesc2_elems_rgh(RGH, S) + 15) / 16) * 16
                                                                                 : RGH * S + hkey_store_halfs(S) / 4; }
constexpr int hdesc2_offset(int RG, int S) { return hdesc2_offset_rgh(hdesc_sector(RG), S); }
constexpr int hdesc2_elems(int RG, int S) { return hdesc2_elems_rgh(hdesc_sector(RG), S); }
constexpr int hdesc_stride(int RG, int S) { return hdesc_stride_rgh(hdesc_sector(RG), S); }  // ... per slot

// ---- database layout in HBM (one "slot" per keyframe) -----------------------
//   desc   float4 [cap][RG][S]   RG = ceil(R/4); element (rg, c) holds rows
//                                4rg..4rg+3 of column c (rows >= R are 0):
//                                lane c of a wave reads 16 B, the wave 1 KiB+.
//   vkey   double [cap][S]       sector key   (D.h:1477-1489), fp64
//   norm   double [cap][S]       column L2 norms (D.h:1523), fp64
//   rkey   float  [cap][R4]      ring key, row-major (query side), R4 = 4*RG
//   rkey4  float4 [RG][cap]      ring key, tiled for the top-k scan
//   hdesc  half   [cap][S][4*RGH] the screening pass's own copy (RGH = ring groups padded to whole 64-byte steps: 16 at R = 64, 24 at R = 80): unit columns (x * (float)(1 / norm), fp32) rounded to fp16; an all-zero column stays zero, a norm outside [2^-60, 2^60] or non-finite makes the column NaN, sector-major:
//                                all rings of one sector are consecutive (128 B = one cache line on the 64 x 120 grid), so
//                                a ring shift is a rotation of whole lines and every line is read once.  Half the bytes of
//                                desc.  Behind it the sector key as a unit vector in fp16 (vkey / |vkey|, zero padded to a multiple of 32; all zero when the
//                                norm is zero or not finite) and its norm as a float: the first stage of the alignment filter; behind that, 128-byte aligned, the
//                                chunk-major image of the same values (hdesc2_offset / hdesc2_elems above).  hstride = hdesc_stride(RG, S) elements of 8 B.
//   hkey   half   [cap][hkey_row_halfs(S)]  the fp16 sector key and its norm as stored behind hdesc's copy (+ the key's second fp16 part), in a table of its own: the alignment reads
//                                nothing else of a keyframe, and 272 B at a stride of 33 KB cost it a DRAM page and a TLB entry per keyframe
//   kmask  u32    [cap][8]       bit c of words 0..6 = column c has a non-zero norm (grids of up to 192 sectors: word 6 = float E, the summed fp16 rounding-error norms of the screening copy's unit columns, rounded up -- make_sc.hip); word 7 bit 0 = some column norm is outside [2^-60, 2^60] or non-finite (such keyframes are always scored exactly)
struct DbView {
    const float4 *desc;
    const double *vkey;
    const double *norm;
    const uint2  *hdesc;
    const unsigned int *kmask;
    const unsigned short *hkey;   // [cap][hkey_row_halfs(S)] the unit fp16 sector keys + norms once more, DENSE (272 B per keyframe at S = 120): what the alignment reads
    const unsigned char *halign;  // [cap][halign_bytes(S)] the alignment images of the database's keyframes (no staging rows: a scan is the other operand); may be null
    int hstride;
    const float  *rkey;
    const float4 *rkey4;
    int cap;     // slot stride of rkey4
    int R, S, RG;
};

struct QueryView {          // one descriptor in the same layout (a DB slot or the staged query)
    const float4 *desc;     // [RG][S]
    const double *vkey;     // [S]
    const double *norm;     // [S]
    const float  *rkey;     // [R4]
    const uint2  *hdesc;    // [hstride]
    const unsigned int *kmask;   // [8]
};

// K1: distanceBtnScanContext for n candidates.
//   slot(i) = cand ? cand[i] : slot_base + i ; slot < 0 -> (1e7, 0) without compute.
// Returns false when (R,S,SR) has no specialised kernel and the generic one must be used.
// FullTail (optional, needs out_ring_d2 and a fusing grid): the kernel also reduces the global arg-min and the
// ring-key top-k itself (per-workgroup partials + last-workgroup reduction), so no epilogue launch is needed.
// blk_part: kTailRec u64 per workgroup (>= 1024 workgroups); done_counter: one zeroed u32.
constexpr int kTailTop = 4;                 // ring-key candidates the fused epilogue can track (k <= kTailTop)
constexpr int kTailRec = 2 + kTailTop;
constexpr int kTailBlocks = 1024;           // workgroup records per blk_part set
constexpr int kTailTopMaxK = 64;            // entries per topk_idx / topk_d2 set (= kTopkMaxK)
struct FullTail {
    unsigned long long *blk_part; unsigned int *done_counter; double *out3; int *topk_idx; float *topk_d2;
    int k; float exclude_eps;
};
// Several database-resident queries scored by ONE launch of the fused kernel (full-DB mode): query i is
// slot[i], scored against slots base[i] .. base[i]+n[i]-1, its winner written to out3[i] (3 doubles).
// Per-query scratch is the single-query scratch repeated: out_dist / out_shift / out_ring_d2 with stride
// pair_stride, blk_part with stride kTailBlocks*kTailRec, done_counter +1, topk_idx / topk_d2 +kTailTopMaxK.
constexpr int kMaxQueryBatch = 4;
struct QueryBatch {
    int nq;
    int slot[kMaxQueryBatch], base[kMaxQueryBatch], n[kMaxQueryBatch];
    double *out3[kMaxQueryBatch];
    size_t pair_stride;
};
hipError_t launch_sc_distance_batch(const struct DbView &db, const QueryBatch &qb, int SR, double *out_dist, int *out_shift,
                                    float *out_ring_d2, const FullTail &tail, int num_cu, hipStream_t stream);
// The exact distance MATRIX: nq <= kMaxQueryBatch query keyframes (database / staging slots) against the range [base, base + n),
// every pair through the exact fp64 program; row i of the outputs at out_dist / out_shift + i * row_stride.  Grids with a wave
// program score all rows in one launch (a row's workgroups take over CUs as the previous row's retire), the others row by row.
hipError_t launch_sc_distance_matrix(const struct DbView &db, const int *slots, int nq, int base, int n, int SR,
                                     double *out_dist, int *out_shift, size_t row_stride, int num_cu, hipStream_t stream);
// ---- exact distances at the shifts the screening leaves open (sc_masked.hip) ------------------------------------------------------
// Query i: keyframe qslot against the items of its list -- cand[item] (database slots, ascending) or, cand == nullptr, the range
// slot_base + item -- n of them (n_dev != nullptr: counted on the device).  starts / smask are indexed by the slot's position in the
// scan's range (slot - slot_base): the first searched shift and the bit mask of the shifts still open (bit t: shift (first + t) mod S).
// out_dist / out_shift are indexed by item.  parts: workgroups per query (each takes every parts-th group of 8 items).
constexpr int kMaxMaskedQueries = 16;
struct MaskedQuery {
    int qslot, slot_base, n; const int *n_dev; const int *cand; const int *starts; const unsigned int *smask;
    double *out_dist; int *out_shift;
};
struct MaskedArgs { const float4 *desc; const double *norm; int nq, parts, spread; MaskedQuery q[kMaxMaskedQueries]; };
// ---- the exact distance MATRIX on the screened grids (sc_matrix.hip): scans qslots[0 .. nq) against the keyframes [lo, lo + n), every pair at
// the shifts its mask leaves open.  starts / smask of scan i at i * set_stride + position in the range, results of row i at i * row_stride +
// position.  kr: keyframes per workgroup (0: the default).
bool sc_matrix_supported(const struct DbView &db, int SR);
hipError_t launch_sc_matrix(const struct DbView &db, int SR, const int *qslots, int nq, int lo, int n, const int *starts, const unsigned int *smask,
                            size_t set_stride, double *out_dist, int *out_shift, size_t row_stride, int kr, hipStream_t stream);
// ---- the exact pass of a small batch of screened scans (sc_masked.hip: sc_small_exact_kernel) -- one workgroup per scan lists the
// keyframes within the margin of the smallest screened distance, scores them at their open shifts, forms the ring-key top-k and writes
// the winner {distance, position in the range or -1, shift} to out3 (pinned).  list: scratch of n ints; t_min is re-armed.
struct SmallExactQuery {
    int qslot, base, n;
    const float *approx; const int *starts; const unsigned int *smask; const float *ring_d2; unsigned int *t_min;
    int *list; double *out3; int *topk_idx; float *topk_d2;
};
struct SmallExactArgs {
    const float4 *desc; const double *norm; const double *vkey;               // (filled by the launcher)
    int nq, k; float exclude_eps, two_eps; unsigned long long *surv_stats;
    int every_shift;                                                          // the scans' survivors come without shift masks: the instance that scores every shift in one pass
    unsigned int seq;                                                         // != 0: written to out3[4] behind the winner (a blocking call polls the pinned record
                                                                              // instead of an event: the event's packet fires 4-6 us behind the kernel)
    const SmallExactQuery *q_dev;                                             // the queries in device memory (a chunk of the stream form: up to
                                                                              // kMaxSmallExactQueries) instead of q[] (nq <= kMaxQueryBatch)
    SmallExactQuery q[kMaxQueryBatch];
};
constexpr int kMaxSmallExactQueries = 128;
// ---- the k candidates of a ring-key search, scored by one workgroup (sc_masked.hip: sc_cand_exact_kernel): candidate cand_idx[i] (-1:
// none -> (1e7, 0)) against the query; idx[k] | d2[k] | dist[k] | shift[k] written to pinned_out ----
struct CandExactArgs {
    const float4 *desc; const double *norm; const double *vkey; const float4 *q_desc; const double *q_norm; const double *q_vkey;
    int k; const int *cand_idx; const float *cand_d2; char *out;
    const unsigned long long *lists; int n_lists;   // lists != nullptr: the ring-key scan's per-workgroup lists -- the kernel merges them first (topk_merge.hpp) and
    int *cand_idx_out; float *cand_d2_out;          // leaves the k candidates here as well (scl_get_last_topk); cand_idx / cand_d2 are not read then
    unsigned int seq;                  // != 0: written to out + cand_seq_offset(k) when the block is complete (the caller polls it instead of an event)
};
constexpr size_t cand_seq_offset(int k) { return (((size_t)k * (sizeof(int) * 2 + sizeof(float) + sizeof(double))) + 15) / 16 * 16; }
bool sc_cand_exact_supported(const struct DbView &db, int SR);
hipError_t launch_sc_cand_exact(const struct DbView &db, const struct QueryView &q, int SR, int k, const int *cand_idx, const float *cand_d2, void *pinned_out, hipStream_t stream,
                                unsigned int seq = 0, const unsigned long long *lists = nullptr, int n_lists = 0, int *cand_idx_out = nullptr, float *cand_d2_out = nullptr);
constexpr int kCandMergeMaxKeys = 1024;   // lists x k the candidates' kernel merges itself (LDS of its wave rows); more: launch_topk_merge first
bool sc_small_exact_supported(const struct DbView &db, int SR);
hipError_t launch_sc_small_exact(const struct DbView &db, int SR, const SmallExactArgs &args, hipStream_t stream);
bool sc_masked_supported(const struct DbView &db, int SR);
// spread: short survivor lists (only part 0 of a query has work, as a rule): the queries' workgroups go round the XCDs -- the mapping for
// long lists puts every part 0 on XCD 0, and beside a launch that leaves two CUs per XCD free they ran eight behind each other
hipError_t launch_sc_masked(const struct DbView &db, int SR, const MaskedQuery *queries, int nq, int parts, hipStream_t stream, bool spread = false);
// ---- screening pass of the full-DB mode (sc_screen.hip; 64x120 grid) --------------------------------------------
// Per query i: every keyframe of [base, base+n) gets approx[i*pair_stride + pos] = its reference distance within
// +- sc_screen_eps() (fp16 matrix-core evaluation of the reference's own 13 shifts; -inf = must be scored exactly),
// ring_d2 = its ring-key metric (the exact pass forms the top-k from it).  launch_sc_select_batch (diagnostics only):
// survivors = the database slots (ascending) that can still hold the minimum.
// t_min: one word per query, 0xffffffff before the first launch (the select launch re-arms it).
constexpr int kWideExactBatch = 16;         // queries per launch of the 80 x 180 grid's exact pass (select, one-sector-per-lane program, arg-min): a screening launch's worth
constexpr int kMaxScreenBatch = 16;         // scans per screening launch (the fused exact kernel of the 20 x 60 grid keeps kMaxQueryBatch)
struct ScreenBatch {
    int nq;
    int slot[kMaxScreenBatch], base[kMaxScreenBatch], n[kMaxScreenBatch];
    int buf[kMaxScreenBatch];                    // buffer set of query i: approx / ring_d2 / survivors at buf * pair_stride, t_min[buf], top-k set buf
    size_t pair_stride;
    float *approx; float *ring_d2; int *survivors; int *n_surv; unsigned int *t_min;
    int *starts;                                // first shifts (alignment kernel -> screening kernel), like approx
    unsigned long long *align_fallbacks;        // optional counter: keyframes aligned by the exact evaluation
    unsigned long long *surv_stats;             // optional (select launch): [0] += survivors, [1] = max(survivors), [2] += 1 per query
    unsigned int *smask;                        // optional, like approx: per pair the shifts whose screened distance is within 2 eps of the pair's smallest
                                                // (bit t: shift (first + t) mod S; every shift for a pair the screening cannot bound): sc_masked.hip's input
    int k; float exclude_eps; int *topk_idx; float *topk_d2;
    hipStream_t side; hipEvent_t ev_fork, ev_join;   // second form: stream and events for the next batch's alignment beside the products (nullptr: in line)
    float *part;                                // scratch of the second form of the 64 x 120 products: nq * pair_stride * 32 floats (nullptr: first form)
    bool no_ring_metric;                        // the caller's exact pass forms the ring-key metric itself (SurvivorPass::ring_from_keys) or has no use for
                                                // it (the distance matrix): the second form's tail launch leaves ring_d2 alone (the first form and the
                                                // last group's plain finish still write it -- the same values)
};
bool sc_screen_supported(const struct DbView &db, int SR);
bool sc_screen_is_wide(const struct DbView &db, int SR);      // 80 x 180: screening by sc_screen_wide_kernel, exact pass by the one-sector-per-lane kernel
float sc_screen_eps();
int sc_screen_max_batch(const struct DbView &db, int SR);            // scans per screening launch on this grid (16; 12 on 80 x 180)
size_t sc_screen_scratch_floats(const struct DbView &db, int SR);   // partial-sum scratch of the second form per keyframe of a buffer set
// phases: 1 = the alignment kernel (first shifts + ring-key metric into sb.starts / sb.ring_d2), 2 = the screening products
// (which read sb.starts), 3 = both, one after the other on `stream`.  next (optional): the batch that follows; its
// alignment rides in the launch of this batch's products (further workgroups of the same grid: one is matrix-core bound,
// the other HBM bound), so the next call needs phase 2 only.
// kScreenDeferFinish (with kScreenProducts, second form only: sc_screen_can_defer): the batch's finishing -- bound d~, flags, ring-key
// metric: what the exact pass reads -- is left to the extra waves of the NEXT launch, which gets this batch as `prev` (same buffer
// sets, same sb.part, which therefore must not be the next launch's own), or to a last call with phases = kScreenFinish.
// prev: the batch whose finishing rides in this launch.
constexpr int kScreenAlign = 1, kScreenProducts = 2, kScreenDeferFinish = 4, kScreenFinish = 8;
hipError_t launch_sc_screen_batch(const struct DbView &db, const ScreenBatch &sb, int SR, int align_filter, int num_cu, hipStream_t stream,
                                  int phases = kScreenAlign | kScreenProducts, const ScreenBatch *next = nullptr, const ScreenBatch *prev = nullptr);
bool sc_screen_can_defer(const struct DbView &db, int SR, int nq);
hipError_t launch_sc_select_batch(const ScreenBatch &sb, hipStream_t stream);
// Exact pass over the survivors of nq screened queries (any number: the argument sets travel through device memory).
// Query i: keyframe slot[i] against the range [base[i], base[i] + n[i]); its screening results live in buffer set
// buf[i] (approx / survivors / out_dist / out_shift at buf[i] * pair_stride, t_min[buf[i]]).  Every workgroup builds
// the survivor list itself (approx <= min + 2 eps, ascending slots), scores its share with the fp64 wave program, and
// the last workgroup of the query writes the winner (distance, position relative to base[i], shift) to out3[i] and
// re-arms t_min.  blk_part: nq * kSurvivorBlocks * kTailRec words, done_counter: nq zeroed words,
// d_args / h_args (pinned): nq * kSurvivorArgBytes bytes.
// workgroups (of 8 waves) per query: one scores the survivors (one to five as a rule, a wave each), one forms the ring-key top-k
// beside it.  The pass needs 126 KB of LDS per workgroup and cannot share a CU with the screening products (138 KB): every
// CU it holds is a CU the main stream waits for -- 4 per query (256 for a chunk of 64 scans) cost 7 % of the stream's rate, 1 loses
// the overlap of top-k and scoring.
constexpr int kSurvivorBlocks = 2;
constexpr int kSurvivorArgBytes = 384;
constexpr int kMaxSurvivorQueries = 128;
struct SurvivorPass {
    int nq;
    int slot[kMaxSurvivorQueries], base[kMaxSurvivorQueries], n[kMaxSurvivorQueries], buf[kMaxSurvivorQueries];
    double *out3[kMaxSurvivorQueries];
    size_t pair_stride;
    const float *approx; int *survivors; unsigned int *t_min; double *out_dist; int *out_shift;
    const float *ring_d2; int k; float exclude_eps; int *topk_idx; float *topk_d2;   // ring-key top-k of every range (workgroup 0 of the query)
    unsigned long long *blk_part; unsigned int *done_counter;
    unsigned long long *surv_stats;             // optional: [0] += survivors, [1] = max(survivors), [2] += 1 per query scored
    void *d_args; void *h_args;
    int ring_from_keys;                         // 1: ring_d2 is scratch the top-k workgroup of every range fills from the tiled ring keys first
                                                // (ScreenBatch::no_ring_metric: the screening launches did not)
};
// phases: kSurvivorArgs = fill the argument sets and enqueue their copy to the device (may be done ahead of the event the
// kernel has to wait for), kSurvivorKernel = the kernel; both by default
constexpr int kSurvivorArgs = 1, kSurvivorKernel = 2;
hipError_t launch_sc_distance_survivors(const struct DbView &db, const SurvivorPass &sp, int SR, int num_cu, hipStream_t stream,
                                        int phases = kSurvivorArgs | kSurvivorKernel);
hipError_t launch_sc_distance_survivors_wide(const struct DbView &db, int nq, const int *query_slot, const int *slot_base, int SR,
                                             const int *const *survivors, const int *const *n_surv, double *const *out_dist, int *const *out_shift,
                                             double *const *out3, int num_cu, hipStream_t stream,
                                             const int *const *starts = nullptr, const unsigned int *const *smask = nullptr);
int sc_align_filter_enabled();

// out_ring_d2 (optional): also write the squared ring-key distance (nanoflann metric) of every scored
// slot; *ring_fused tells whether the selected kernel supports it (the two-sectors-per-lane grids do).
hipError_t launch_sc_distance(const DbView &db, const QueryView &q, const int *cand, int slot_base,
                              int n, int SR, double *out_dist, int *out_shift, int num_cu,
                              hipStream_t stream, float *out_ring_d2 = nullptr, bool *ring_fused = nullptr,
                              const FullTail *tail = nullptr);

// arg-min over (dist[i], i) for i in [0,n): strict <, first wins, NaN never wins,
// nothing below 1e7 -> idx -1.  out: {dist, (double)idx, (double)shift} packed as 3 doubles.
bool sc_distance_fuses_ring(const DbView &db, int SR);
hipError_t launch_argmin(const double *dist, const int *shift, int n, double *out3, hipStream_t stream);
// arg-min + ring-key top-k (from fused distances) in one launch; idx are slot_base-relative inputs
hipError_t launch_full_epilogue(const double *dist, const int *shift, const float *ring_d2, int n, int slot_base,
                                int k, float exclude_eps, double *out3, int *topk_idx, float *topk_d2,
                                hipStream_t stream);

// K2: exact ring-key top-k over slots [lo,hi).  out_idx[k], out_d2[k] (device).
// scratch must hold kTopkMaxBlocks * kTopkMaxK uint64.
constexpr int kTopkMaxBlocks = 256;
constexpr int kTopkMaxK = 64;
static_assert(kTopkMaxK == kTailTopMaxK, "top-k sets share one size");
// idx[k] | d2[k] | dist[k] | shift[k] into one block of pinned host memory (k results of a top-k and their SC distances)
hipError_t launch_topk_pack(const int *idx, const float *d2, const double *dist, const int *shift, int k, bool have_dist, void *pinned_out, hipStream_t stream);
// the scan alone: per-workgroup lists of k keys in `scratch` (*n_lists of them; 0 = empty range), merged by the consumer (topk_merge.hpp) or by
// launch_topk_merge (-> out_idx / out_d2 in device memory and, pinned_out != nullptr, the block idx[k] | d2[k] the host reads)
hipError_t launch_ringkey_lists(const DbView &db, const float *qkey, int lo, int hi, int k, float exclude_eps,
                                unsigned long long *scratch, int *n_lists, hipStream_t stream);
hipError_t launch_topk_merge(const unsigned long long *scratch, int n_lists, int k, int *out_idx, float *out_d2, void *pinned_out, hipStream_t stream);
hipError_t launch_ringkey_topk(const DbView &db, const float *qkey, int lo, int hi, int k,
                               float exclude_eps, unsigned long long *scratch,
                               int *out_idx, float *out_d2, hipStream_t stream);

// ingest: row-major wire descriptors (device) -> DB slots first_slot.. (all derived data)
hipError_t launch_ingest(const float *values, int count, int first_slot,
                         float4 *desc, double *vkey, double *norm, float *rkey, float4 *rkey4,
                         uint2 *hdesc, unsigned int *kmask, unsigned short *hkey, int hstride,
                         int cap, int R, int S, hipStream_t stream, unsigned char *halign = nullptr,
                         int *tiles = nullptr, float *vals_out = nullptr);   // tiles: descriptors from the batch scatter's tiles (values may be nullptr)
// tiled -> row-major wire format (read back)
hipError_t launch_untile(const float4 *desc_slot, int R, int S, float *values, hipStream_t stream);

// K3 for a batch of scans: one scatter launch over all of them, then launch_ingest(tiles = ...) -- two launches per batch.
constexpr int kMaxScBatch = 16;
constexpr int kScPointsPerWorkgroup = 4096;    // points of a cloud per scatter workgroup (its private LDS tile is merged with <= R*S atomics)
struct ScanBatch {
    const unsigned char *points[kMaxScBatch];      // device pointers
    int n[kMaxScBatch];
    int first_wg[kMaxScBatch + 1];                 // filled by launch_make_sc_batch
    int count;
};
struct IngestArgs {
    const float *values; int first_slot; float4 *desc; double *vkey; double *norm; float *rkey; float4 *rkey4; uint2 *hdesc; unsigned int *kmask;
    unsigned short *hkey; int hstride, cap, R, S; unsigned char *halign; int *tiles; float *vals_out;
};
// ingest of n_ingest scans (ia.tiles: the previous group's tiles) beside the scatter of batch b into gtiles, in ONE launch
hipError_t launch_front_fused(ScanBatch b, int stride_bytes, double lidar_height, double max_radius, int *gtiles, int points_per_wg,
                              const IngestArgs &ia, int n_ingest, hipStream_t stream);
hipError_t launch_make_sc_tiles_init(int *tiles, int count, int R, int S, hipStream_t stream);
// descriptor only: tiles -> values[count][R*S] (row-major wire format), tiles back to their initial state
hipError_t launch_make_sc_finalize(int *tiles, int count, int R, int S, float *values, hipStream_t stream);
hipError_t launch_make_sc_batch(ScanBatch b, int stride_bytes, int R, int S, double lidar_height, double max_radius,
                                int *tiles, int points_per_wg, int num_cu, hipStream_t stream);
// test hook: checksums of the device's atanf over blocks of 2^24 float bit patterns (tests/golden/atanf_blocks.json)
hipError_t launch_atanf_block_checksums(int first_block, int n_blocks, unsigned long long *d_out, hipStream_t stream);
// test hook: the scatter's fast binning against the reference's chain on n generated points: d_out2[0] = disagreements where the fast path was sure, [1] = sure
hipError_t launch_bin_paths_selftest(int mode, unsigned long long seed, unsigned long long n, int R, int S, double max_radius, unsigned long long *d_out2, hipStream_t stream);

}  // namespace scl
