// engine_internal.hpp -- the engine object behind the C ABI, shared by engine.hip (one database on one GPU)
// and sharded_front.hip (one database spread over several GPUs, include/scl_engine.h scl_create_sharded).
#pragma once

#include "scl_engine.h"

#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "icp.hpp"
#include "kernels.hpp"

namespace scl {

enum ProfKind { P_SC = 0, P_TOPK, P_ARGMIN, P_MAKESC, P_INGEST, P_ICPNN, P_ICPRED, P_COUNT };

struct PendingEvent { hipEvent_t start, stop; int kind; int launches = 1; };   // launches: what the pair brackets (a chunk's screening: its launch groups)

struct ShardedFront;                                       // sharded_front.hip

}  // namespace scl

struct scl_engine {
    scl_config cfg;
    scl::ShardedFront *front = nullptr;                    // != nullptr: this object is the front of a sharded database
                                                           // (no device state of its own; every call is forwarded)
    int R = 0, S = 0, RG = 0, R4 = 0, SR = 0;
    int device = 0, num_cu = 256;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;                         // ring-key scan runs beside the SC distance
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    double *h_out3 = nullptr;                              // pinned, device-visible: the arg-min kernel writes it directly
    static constexpr int kSlots = 8;                       // full-DB passes in flight (submit/collect)
    hipEvent_t ev_done[kSlots] = {nullptr};
    hipEvent_t ev_call = nullptr;                          // end of a short blocking call's launches: polled (sync_short)
    int slot_lo[kSlots] = {0}; bool slot_busy[kSlots] = {false}; bool slot_empty[kSlots] = {false};
    unsigned int pinned_seq = 0;                            // != 0: the number the candidates' kernel writes behind the block in h_pinned (topk_finish polls it)
    unsigned int slot_seq[kSlots] = {0}; unsigned int out_seq = 0;   // != 0: the exact pass writes this number behind the slot's result record (polled)
    int slot_ev[kSlots] = {0};                             // which slot's event completes this one (batched launches share one)
    unsigned next_slot = 0;
    // Two locks.  `mu` guards the database -- the arrays, n / cap, the (robot, index) map, the scratch of an append -- and is held by
    // every entry point for as long as it touches them.  `pass_mu` guards what a detection PASS works in (buffer sets, slots, pinned
    // results, the streams' bookkeeping) and is taken first, by every entry point that scores anything; appends take `mu` only.  The
    // stream form keeps pass_mu for the whole call but gives `mu` up while it waits for a chunk (engine.hip: stream_screened_locked),
    // so the LIO thread's appends (DM.h:1001-1003) get in between the chunks of a long stream instead of behind all of it.
    mutable std::mutex mu;
    mutable std::mutex pass_mu;
    mutable std::string last_error;

    // database (layout: kernels.hpp)
    int n = 0, cap = 0;
    float4 *d_desc = nullptr; double *d_vkey = nullptr; double *d_norm = nullptr;
    float *d_rkey = nullptr; float4 *d_rkey4 = nullptr;
    uint2 *d_hdesc = nullptr; unsigned int *d_kmask = nullptr; int hstride = 0;
    unsigned char *d_halign = nullptr;                      // alignment images of the database's keyframes (kernels.hpp: halign_bytes(S) each)
    unsigned short *d_hkey = nullptr; int hkw = 0;          // dense fp16 sector keys (+ norms), hkw halfs per slot   // the screening pass's fp16 copy + sector masks
    std::vector<int8_t> robots;
    std::vector<int> indexs;

    // staged external queries: kStage slots BEHIND the database slots of desc / vkey / norm / rkey (indices
    // cap .. cap+kStage-1; rkey4 has no staging rows), so a staged query is addressed like a keyframe of the
    // database by every kernel, the multi-query launches included.  Query id -1-j = staging slot j
    // (SCL_QUERY_STAGED = -1 = slot 0, the public one; the sharded front uses the others for keyframes that
    // live on another device).
    // (0: the public staged query; 1..9: the sharded front's passes in flight and blocking calls; 12..75: the front's stream form, one
    // block of 64 scans whose keyframes live on another shard)
    // Behind those, an engine that is a shard of a sharded front carries MIRROR rows (stage_rows > kStage; sharded_front.hip): the query-side
    // rows of the newest keyframes of the OTHER shards, copied over when they are appended, so that a scan of a keyframe that lives on
    // another device needs no copy when it is searched for.
    static constexpr int kStage = 76;
    int stage_rows = kStage;                                // rows behind the database slots (>= kStage); fixed before the first allocation
    std::vector<unsigned char> staged = std::vector<unsigned char>(kStage, 0);

    // scratch
    float *d_vals = nullptr; size_t vals_cap = 0;          // wire-format staging (floats)
    unsigned char *d_points = nullptr; size_t points_cap = 0;
    // K3 (make_sc.hip): the batch scatter's kMaxScBatch polar tiles, in their initial state between calls (tiles_clean)
    int *d_tiles = nullptr; bool tiles_clean = false;
    // the pipelined forms (scl_make_and_save_many, scl_stream_from_points): three device buffers for a group's clouds, filled by the
    // copy stream while the group before is binned, ingested and searched for
    unsigned char *d_pbuf[3] = {nullptr, nullptr, nullptr}; size_t pbuf_cap = 0;
    static constexpr int kCopyStreams = 4;
    hipStream_t stream_copy = nullptr;                      // copy stream 0 (also scl_host_copy_rate's)
    hipStream_t stream_copy_x[kCopyStreams - 1] = {nullptr, nullptr, nullptr};   // copy streams 1..
    hipEvent_t ev_copied[kCopyStreams][3] = {}, ev_consumed[3] = {nullptr, nullptr, nullptr};
    std::vector<void *> host_allocs;                       // scl_host_alloc (pinned), freed with the engine at the latest
    double *d_dist = nullptr; int *d_shift = nullptr; int *d_cand = nullptr; float *d_ring_d2 = nullptr; size_t pair_cap = 0;
    // screening pass of the full-DB mode (sc_screen.hip): approximate distances, survivors, their counts, min d~ words
    // Buffers come in kScreenSets sets of `set_stride` entries (one set per query of a chunk of the stream form; the
    // submit / collect form uses sets 0..3): approx, ring_d2, survivors, dist, shift at set * set_stride.
    // (256: chunks of 128 scans = eight launches of 16.  A chunk costs the main stream one event record and one wait for another stream's
    // event, and either keeps the next launch back by 4-6 us (scripts/probes/probe_dispatch_gap.hip): with chunks of 64 that was 4 % of
    // the stream's time.  With chunks of 32 the 80 x 180 products waited 85 us at every chunk end for the exact pass of the chunk before.)
#ifndef SCL_SCREEN_SETS
#define SCL_SCREEN_SETS 256
#endif
    static constexpr int kScreenSets = SCL_SCREEN_SETS;
    float *d_approx = nullptr; int *d_starts = nullptr; int *d_surv = nullptr;
    unsigned int *d_smask = nullptr; size_t smask_cap = 0;  // per pair the shifts still open after the screening (sc_masked.hip); allocated on first use
    float *d_part = nullptr; size_t part_cap = 0;       // partial sums of the screening products' second form (one launch at a time)
    int *d_nsurv = nullptr; unsigned int *d_tmin = nullptr;
    unsigned long long *d_align_fallbacks = nullptr; uint64_t align_pairs = 0;   // statistics of the alignment kernel (scl_alignment_stats)
    unsigned long long *d_surv_stats = nullptr;                                   // [3] survivors of the screening pass: sum, max, queries (scl_survivor_stats)
    size_t set_stride = 0;
    unsigned long long *d_surv_part = nullptr; unsigned int *d_surv_done = nullptr;        // tail of the exact pass
    void *d_surv_args = nullptr; void *h_surv_args = nullptr; unsigned surv_arg_tick = 0;   // argument sets of the exact pass (ring of 8 regions)
    double *h_stream_out = nullptr;                        // pinned: 2 x kScreenSets result records of the stream form
    hipEvent_t ev_chunk[2] = {nullptr, nullptr};
    bool exact_heavy = false;                     // 64 x 120 stream: the last chunks left dozens of survivors per scan -> the survivors' kernel scores the next ones
    hipEvent_t ev_sub0[2] = {nullptr, nullptr};   // 80 x 180: behind the FIRST launch group of a chunk's exact pass (the buffer sets the next-but-one chunk's first alignment writes)
    // stream form: the exact pass over a chunk's survivors (small, latency bound) runs on its own low-priority stream
    // beside the screening products of the next chunk; the main stream carries the products back to back.
    hipStream_t stream_surv = nullptr;
    hipEvent_t ev_k1[2] = {nullptr, nullptr};
    hipEvent_t ev_align_gate = nullptr;
    // second form of the screening products: the next batch's alignment runs as its own kernel on a low-priority stream beside them
    hipStream_t stream_align = nullptr;
    hipEvent_t ev_afork = nullptr, ev_ajoin = nullptr;
    bool screen = false;                                   // grid supported and not switched off (SCL_SCREEN=0)
    bool in_single_fallback = false;                       // submit_full_locked <-> submit_full_many_locked recursion guard
    unsigned long long *d_topk_scratch = nullptr; int *d_topk_idx = nullptr; float *d_topk_d2 = nullptr;
    double *d_out3 = nullptr;
    unsigned long long *d_blk_part = nullptr; unsigned int *d_done_counter = nullptr;   // fused full-DB epilogue
    // Second lane for fused full-DB passes: consecutive passes alternate between `stream` and `stream_alt`
    // (own epilogue scratch and per-pair outputs), so the next pass's workgroups move onto CUs as the previous
    // pass's workgroups retire instead of waiting behind its completion packet.
    hipStream_t stream_alt = nullptr;
    hipEvent_t ev_db = nullptr;                            // database writes on `stream` the alt lane must see
    uint64_t db_version = 0, alt_seen_version = 0;
    unsigned long long *a_blk_part = nullptr; unsigned int *a_done_counter = nullptr;
    int *a_topk_idx = nullptr; float *a_topk_d2 = nullptr;
    double *a_dist = nullptr; int *a_shift = nullptr; float *a_ring_d2 = nullptr; size_t a_pair_cap = 0;
    bool last_pass_alt = false;
    bool last_pass_empty = false;                          // the most recent full-DB pass had nothing to score
    bool alt_lane = false;                                 // SCL_ALT_LANE=1: lowest latency per scan; kernels of the two lanes overlap,
                                                           // so per-kernel durations no longer measure one pass (default off)
    void *h_pinned = nullptr; size_t pinned_cap = 0;       // small result read-back
    // scl_sc_distance_matrix: two halves of (rows of a launch) x (row length) device results and their pinned copies
    double *d_mat_dist = nullptr; int *d_mat_shift = nullptr; void *h_mat = nullptr; size_t mat_cap = 0;
    hipEvent_t ev_mat_k[2] = {nullptr, nullptr}, ev_mat_c[2] = {nullptr, nullptr};

    // inter-robot tree bookkeeping (descriptor.h:1691-1703, counter initialised: see DESIGN.md)
    int tree_counter = 0, tree_n = 0;

    // profiling
    int prof_on = 0;                                       // 0 off, 1 every kernel family, 2 SC distance only, 3 SC distance sampled 1:7
    unsigned prof_tick = 0;
    scl_profile prof{};
    std::vector<scl::PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;

    scl::IcpWorkspace icp_ws;
    scl::IcpWorkspace vox_ws;
    static constexpr int kIcpLanes = 8;                    // streams (and host threads) that prepare the candidates of a batch side by side: 8 measured best (4: 4.9 ms of preparation per 25 candidates from the store, 8: 3.6, 3.1 with GPU_MAX_HW_QUEUES=8, 16: 4.1)
    scl::IcpWorkspace icp_lane_ws[kIcpLanes];
    scl::IcpWorkspace vox_lane_ws[kIcpLanes];              // submap assembly (voxel filter) of the candidates of a batch, one per lane
    // scl_icp_align_batch: one workspace per loop candidate (their ICP loops run fused, every step one launch for the
    // whole batch), prepared on the lane streams; candidates beyond kIcpBatch go in further rounds
    static constexpr int kIcpBatch = 32;
    scl::IcpWorkspace icp_batch_ws[kIcpBatch];
    scl::IcpWorkspace icp_batch_ctl;
    hipEvent_t ev_lane[kIcpLanes] = {nullptr};
    hipStream_t icp_lane_stream[kIcpLanes] = {nullptr};

    // on-device keyframe store (robots[id].keyFrameArray, DM.h:86): clouds live in slabs of HBM, bump allocated
    struct StoredCloud { unsigned char *d = nullptr; int n = -1; size_t cap_bytes = 0; };
    std::vector<std::vector<StoredCloud>> kf;              // [robot][index]; n < 0: never stored
    std::vector<void *> kf_slabs;
    size_t kf_slab_used = 0, kf_slab_cap = 0;
    int kf_stride = 0;                                     // fixed by the first put
};


// ---- hooks for the sharded front (sharded_front.hip); each takes the engine's own lock ---------------------------
namespace scl {

// copy keyframe `src_slot` of `src` (descriptor tile, sector key, norms, ring key) into staging slot j of `dst`
// (query id -1-j); device-to-device, ordered on dst's stream.  src == dst is allowed.
int eng_set_stage_rows(scl_engine *e, int rows);            // before the first keyframe is stored: kStage + mirror rows
int eng_stage_from_peer(scl_engine *dst, int j, scl_engine *src, int src_slot, int count = 1);   // count consecutive rows: slots src_slot.. to staging rows j..
// wire descriptor -> staging slot j
int eng_stage_values(scl_engine *e, int j, const float *values);
// ring-key top-k in [lo, hi) + SC distance of those k: enqueue on the engine's stream / wait and unpack
int eng_topk_enqueue(scl_engine *e, int query, int lo, int hi, int k, float eps, bool want_dist, bool *have_dist);
int eng_topk_finish(scl_engine *e, int k, bool have_dist, int *idx, float *d2, double *dist, int *shift, int *found);
// wait for everything enqueued on the engine's streams (before another device's arrays may move)
int eng_sync_streams(scl_engine *e);
// true when appending `count` keyframes would move the database arrays
bool eng_would_regrow(const scl_engine *e, int count);
int eng_truncate(scl_engine *e, int n_keep);
// device-side exchange of full-DB winners: the pinned (device-visible) result record of a ticket, its range start,
// the stream the pass runs on; release = free the ticket without reading it (the exchange has delivered it)
const double *eng_ticket_record(const scl_engine *e, int ticket, int *slot_lo, bool *empty);
hipStream_t eng_stream(const scl_engine *e);
int eng_release_ticket(scl_engine *e, int ticket);

// the front's side of every public entry point (same arguments)
int front_destroy(scl_engine *e);
int front_make_and_save(scl_engine *e, const void *points, int n_points, int stride_bytes, int8_t robot, int index, float *out_values,
                        bool filtered, float leaf, int *n_filtered);
int front_save_bulk(scl_engine *e, const float *values, int count, const int8_t *robots, const int *indexs);
int front_stage_query(scl_engine *e, const float *values);
int front_detect_intra(scl_engine *e, int cur, int *loop_id, float *shift, double *dist);
int front_detect_inter(scl_engine *e, int cur, int *loop_id, float *yaw_rad, double *dist);
int front_get_index(const scl_engine *e, int key, int8_t *robot, int *index);
int front_get_size(const scl_engine *e);
int front_get_slot(const scl_engine *e, int key, scl_engine **child, int *slot);     // owner of global keyframe `key`
int front_topk(scl_engine *e, int query, int lo, int hi, int k, int *idx, float *d2, double *dist, int *shift, int *found);
int front_sc_distance_batch(scl_engine *e, int query, const int *cand, int n, double *dist, int *shift);
int front_submit_many(scl_engine *e, const int *queries, const int *lo, const int *hi, int nq, int *tickets);
int front_collect(scl_engine *e, int ticket, int *nn_idx, int *shift, double *dist);
int front_detect_full_stream(scl_engine *e, const int *queries, const int *lo, const int *hi, int n_queries,
                             int scans_per_launch, int launches_in_flight, int *nn_idx, int *shift, double *dist);
int front_get_last_topk(scl_engine *e, int k, int *idx, float *d2);
int front_icp_align_batch(scl_engine *e, const void *src, int n_src, const void *const *tgts, const int *n_tgts,
                          int n_targets, int stride_bytes, const scl_icp_params *p,
                          float *T, float *fitness, int *converged, int *iterations);
int front_profile_enable(scl_engine *e, int on);
int front_profile_reset(scl_engine *e);
int front_profile_get(scl_engine *e, scl_profile *out);
int front_sc_distance_matrix(scl_engine *e, const int *queries, int nq, int lo, int hi, double *dist, int *shift);
int front_alignment_stats(scl_engine *e, uint64_t *pairs, uint64_t *fallbacks, int reset);
int front_survivor_stats(scl_engine *e, uint64_t *queries, uint64_t *survivors, uint64_t *max_survivors, int reset);
scl_engine *front_primary(const scl_engine *e);            // the shard that runs unsharded work (geometry, keyframe store)

}  // namespace scl
