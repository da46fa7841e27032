// engine.hip -- host side of the C ABI (include/scl_engine.h): keyframe database in
// HBM, locking, launch orchestration.  No compute happens on the host except the
// O(k) candidate loop of detectIntra/InterLoopClosureID (descriptor.h:1645-1673,
// 1721-1755), whose float-narrowing comparison order is part of the contract.
#include "scl_engine.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "db_file.hpp"
#include "engine_internal.hpp"
#include "device_sort.hpp"

#ifdef SCL_DIAGNOSTICS
namespace scl { void ingest_stamps_print(); void cand_stamps_print(); }   // make_sc.hip / sc_masked.hip: phase stamps (SCL_INGEST_STAMPS=1)
#endif

using namespace scl;

namespace {

#define SCL_HIP(e_, call)                                                              \
    do {                                                                               \
        hipError_t err__ = (call);                                                     \
        if (err__ != hipSuccess) {                                                     \
            (e_)->last_error = std::string(#call) + ": " + hipGetErrorString(err__);   \
            return err__ == hipErrorOutOfMemory ? SCL_ERR_NOMEM : SCL_ERR_HIP;         \
        }                                                                              \
    } while (0)

// one spin of a polling loop, on whatever the host is (the engine's host code also builds for aarch64 robots)
inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

int fail(const scl_engine *e, int code, const char *msg)
{
    if (e) e->last_error = msg;
    return code;
}

struct ProfScope {
    scl_engine *e; int kind; hipStream_t s; hipEvent_t start = nullptr, stop = nullptr;
    ProfScope(scl_engine *e_, int kind_, hipStream_t s_ = nullptr) : e(e_), kind(kind_), s(s_ ? s_ : e_->stream)
    {
        if (!e->prof_on || (e->prof_on >= 2 && kind != P_SC)) return;
#ifndef SCL_PROF_SAMPLE
#define SCL_PROF_SAMPLE 13
#endif
        if (e->prof_on == 3 && (e->prof_tick++ % SCL_PROF_SAMPLE) != 0) return;   // sampled: one launch in thirteen (not a divisor of the launches per chunk;
                                                                                 //  an event pair keeps two launches back by 4-6 us each)
        auto get = [&]() {
            hipEvent_t ev = nullptr;
            if (!e->event_pool.empty()) { ev = e->event_pool.back(); e->event_pool.pop_back(); }
            else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr;
            return ev;
        };
        start = get(); stop = get();
        if (start && stop) (void)hipEventRecord(start, s);
    }
    bool active() const { return start && stop; }
    ~ProfScope()
    {
        if (!e->prof_on || !start || !stop) return;
        (void)hipEventRecord(stop, s);
        e->pending.push_back({start, stop, kind, 1});
    }
};

void collect_profile(scl_engine *e)
{   // folds every finished (start, stop) pair into the profile; unfinished ones stay pending
    std::vector<PendingEvent> keep;
    for (auto &p : e->pending) {
        if (hipEventQuery(p.stop) != hipSuccess) { keep.push_back(p); continue; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            switch (p.kind) {
            case P_SC:     e->prof.sc_distance_ms += ms;  e->prof.sc_distance_launches += (uint64_t)(p.launches > 0 ? p.launches : 1); break;
            case P_TOPK:   e->prof.ringkey_topk_ms += ms; e->prof.ringkey_topk_launches++; break;
            case P_ARGMIN: e->prof.argmin_ms += ms;       e->prof.argmin_launches++; break;
            case P_MAKESC: e->prof.make_sc_ms += ms;      e->prof.make_sc_launches++; break;
            case P_INGEST: e->prof.ingest_ms += ms;       e->prof.ingest_launches++; break;
            case P_ICPNN:  e->prof.icp_nn_ms += ms;       e->prof.icp_nn_launches++; break;
            case P_ICPRED: e->prof.icp_reduce_ms += ms;   e->prof.icp_reduce_launches++; break;
            default: break;
            }
        }
        e->event_pool.push_back(p.start);
        e->event_pool.push_back(p.stop);
    }
    e->pending.swap(keep);
}

int sync(scl_engine *e)
{
    SCL_HIP(e, hipStreamSynchronize(e->stream));
    collect_profile(e);
    return SCL_OK;
}

// The end of a SHORT blocking call (tens of microseconds of device time): an event behind its last launch, polled for up to 300 us,
// then a sleep in the runtime -- hipStreamSynchronize wakes up tens of microseconds late, a fifth of such a call.
int sync_short(scl_engine *e)
{
    if (!e->ev_call) SCL_HIP(e, hipEventCreateWithFlags(&e->ev_call, hipEventDisableTiming));
    SCL_HIP(e, hipEventRecord(e->ev_call, e->stream));
    hipError_t q = hipErrorNotReady;
    const auto t0 = std::chrono::steady_clock::now();
    while ((q = hipEventQuery(e->ev_call)) == hipErrorNotReady)
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) { q = hipEventSynchronize(e->ev_call); break; }
    SCL_HIP(e, q);
    collect_profile(e);
    return SCL_OK;
}

template <class T>
int dev_alloc(scl_engine *e, T **p, size_t count)
{
    void *q = nullptr;
    SCL_HIP(e, hipMalloc(&q, sizeof(T) * (count ? count : 1)));
    *p = static_cast<T *>(q);
    return SCL_OK;
}

template <class T>
void dev_free(T *&p) { if (p) { (void)hipFree(p); p = nullptr; } }

DbView db_view(const scl_engine *e)
{
    DbView v;
    v.desc = e->d_desc; v.vkey = e->d_vkey; v.norm = e->d_norm; v.rkey = e->d_rkey; v.rkey4 = e->d_rkey4;
    v.hdesc = e->d_hdesc; v.kmask = e->d_kmask; v.hkey = e->d_hkey; v.hstride = e->hstride; v.halign = e->d_halign;
    v.cap = e->cap; v.R = e->R; v.S = e->S; v.RG = e->RG;
    return v;
}

int query_view(const scl_engine *e, int query, QueryView *q)
{
    if (query < 0) {                                       // staging slot j = -1 - query
        const int j = -1 - query;
        if (j >= e->stage_rows || !e->staged[j]) return fail(e, SCL_ERR_INVALID_ARG, "no staged query (call scl_stage_query first)");
        query = e->cap + j;
    } else if (query >= e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "query slot out of range");
    q->desc = e->d_desc + (size_t)query * e->RG * e->S;
    q->vkey = e->d_vkey + (size_t)query * e->S;
    q->norm = e->d_norm + (size_t)query * e->S;
    q->rkey = e->d_rkey + (size_t)query * e->R4;
    q->hdesc = e->d_hdesc + (size_t)query * e->hstride;
    q->kmask = e->d_kmask + (size_t)query * 8;
    return SCL_OK;
}

// grow the database to hold at least `need` slots (geometric growth, D2D copies)
int ensure_capacity(scl_engine *e, int need)
{
    if (need <= e->cap) return SCL_OK;
    int ncap = e->cap > 0 ? e->cap : (e->cfg.initial_capacity > 0 ? e->cfg.initial_capacity : 4096);
    while (ncap < need) ncap *= 2;
    const size_t tile = (size_t)e->RG * e->S;
    float4 *nd = nullptr; double *nv = nullptr; double *nn = nullptr; float *nr = nullptr; float4 *nr4 = nullptr;
    uint2 *nh = nullptr; unsigned int *nk = nullptr; unsigned short *nhk = nullptr; unsigned char *nha = nullptr;
    const size_t hab = (size_t)halign_bytes(e->S);            // alignment images: database slots only
    int rc;
    const size_t nst = (size_t)ncap + (size_t)e->stage_rows;  // database slots + the staging (and mirror) rows behind them
    if ((rc = dev_alloc(e, &nd, tile * nst))) return rc;
    if ((rc = dev_alloc(e, &nv, (size_t)e->S * nst))) return rc;
    if ((rc = dev_alloc(e, &nn, (size_t)e->S * nst))) return rc;
    if ((rc = dev_alloc(e, &nr, (size_t)e->R4 * nst))) return rc;
    if ((rc = dev_alloc(e, &nh, (size_t)e->hstride * nst))) return rc;
    if ((rc = dev_alloc(e, &nk, (size_t)8 * nst))) return rc;
    if ((rc = dev_alloc(e, &nhk, (size_t)e->hkw * nst))) return rc;
    if (hab && (rc = dev_alloc(e, &nha, hab * (size_t)ncap))) return rc;
    if ((rc = dev_alloc(e, &nr4, (size_t)e->RG * ncap))) return rc;
    SCL_HIP(e, hipMemsetAsync(nr4, 0, sizeof(float4) * (size_t)e->RG * ncap, e->stream));
    if (e->n > 0) {
        SCL_HIP(e, hipMemcpyAsync(nd, e->d_desc, sizeof(float4) * tile * e->n, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nv, e->d_vkey, sizeof(double) * (size_t)e->S * e->n, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nn, e->d_norm, sizeof(double) * (size_t)e->S * e->n, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nr, e->d_rkey, sizeof(float) * (size_t)e->R4 * e->n, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nh, e->d_hdesc, sizeof(uint2) * (size_t)e->hstride * e->n, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nk, e->d_kmask, sizeof(unsigned int) * (size_t)8 * e->n, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nhk, e->d_hkey, sizeof(unsigned short) * (size_t)e->hkw * e->n, hipMemcpyDeviceToDevice, e->stream));
        if (hab) SCL_HIP(e, hipMemcpyAsync(nha, e->d_halign, hab * (size_t)e->n, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpy2DAsync(nr4, sizeof(float4) * ncap, e->d_rkey4, sizeof(float4) * e->cap,
                                    sizeof(float4) * e->n, e->RG, hipMemcpyDeviceToDevice, e->stream));
    }
    if (e->cap > 0) {                                      // staged queries move with the arrays
        const size_t k = (size_t)e->stage_rows;
        SCL_HIP(e, hipMemcpyAsync(nd + tile * ncap, e->d_desc + tile * e->cap, sizeof(float4) * tile * k, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nv + (size_t)e->S * ncap, e->d_vkey + (size_t)e->S * e->cap, sizeof(double) * e->S * k, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nn + (size_t)e->S * ncap, e->d_norm + (size_t)e->S * e->cap, sizeof(double) * e->S * k, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nr + (size_t)e->R4 * ncap, e->d_rkey + (size_t)e->R4 * e->cap, sizeof(float) * e->R4 * k, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nh + (size_t)e->hstride * ncap, e->d_hdesc + (size_t)e->hstride * e->cap, sizeof(uint2) * e->hstride * k, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nk + (size_t)8 * ncap, e->d_kmask + (size_t)8 * e->cap, sizeof(unsigned int) * 8 * k, hipMemcpyDeviceToDevice, e->stream));
        SCL_HIP(e, hipMemcpyAsync(nhk + (size_t)e->hkw * ncap, e->d_hkey + (size_t)e->hkw * e->cap, sizeof(unsigned short) * e->hkw * k, hipMemcpyDeviceToDevice, e->stream));
    }
    SCL_HIP(e, hipStreamSynchronize(e->stream));
    // passes still reading the old arrays: the exact pass of a stream's chunk on the side stream, the ring-key scan, the alternate lane
    // (an append may now run between the chunks of a stream call: stream_screened_locked)
    for (hipStream_t s2 : {e->stream_alt, e->stream_surv, e->stream_align, e->stream2})
        if (s2) SCL_HIP(e, hipStreamSynchronize(s2));
    dev_free(e->d_desc); dev_free(e->d_vkey); dev_free(e->d_norm); dev_free(e->d_rkey); dev_free(e->d_rkey4);
    dev_free(e->d_hdesc); dev_free(e->d_kmask); dev_free(e->d_hkey); dev_free(e->d_halign);
    e->d_desc = nd; e->d_vkey = nv; e->d_norm = nn; e->d_rkey = nr; e->d_rkey4 = nr4;
    e->d_hdesc = nh; e->d_kmask = nk; e->d_hkey = nhk; e->d_halign = nha;
    e->cap = ncap;
    return SCL_OK;
}

int ensure_vals(scl_engine *e, size_t floats)
{
    if (floats <= e->vals_cap) return SCL_OK;
    dev_free(e->d_vals);
    int rc = dev_alloc(e, &e->d_vals, floats);
    if (rc) { e->vals_cap = 0; return rc; }
    e->vals_cap = floats;
    return SCL_OK;
}

int ensure_points(scl_engine *e, size_t bytes)
{
    if (bytes <= e->points_cap) return SCL_OK;
    dev_free(e->d_points);
    size_t nb = bytes + bytes / 4 + 4096;
    int rc = dev_alloc(e, &e->d_points, nb);
    if (rc) { e->points_cap = 0; return rc; }
    e->points_cap = nb;
    return SCL_OK;
}

int ensure_pairs(scl_engine *e, size_t n)
{
    if (n <= e->pair_cap) return SCL_OK;
    dev_free(e->d_dist); dev_free(e->d_shift); dev_free(e->d_cand); dev_free(e->d_ring_d2);
    dev_free(e->d_approx); dev_free(e->d_surv); dev_free(e->d_starts);
    size_t nn = n + n / 2 + 64;
    int rc;
    e->pair_cap = 0;
    if ((rc = dev_alloc(e, &e->d_approx, nn))) return rc;
    if ((rc = dev_alloc(e, &e->d_starts, nn))) return rc;
    if ((rc = dev_alloc(e, &e->d_surv, nn))) return rc;
    if ((rc = dev_alloc(e, &e->d_ring_d2, nn))) return rc;
    if ((rc = dev_alloc(e, &e->d_dist, nn))) return rc;
    if ((rc = dev_alloc(e, &e->d_shift, nn))) return rc;
    if ((rc = dev_alloc(e, &e->d_cand, nn))) return rc;
    e->pair_cap = nn;
    return SCL_OK;
}

// screening buffers: kScreenSets sets of at least n entries each
int ensure_sets(scl_engine *e, size_t n)
{
    if (n <= e->set_stride && e->pair_cap >= e->set_stride * scl_engine::kScreenSets) return SCL_OK;
    const size_t stride = n + n / 4 + 64;
    int rc = ensure_pairs(e, stride * scl_engine::kScreenSets);
    if (rc) { e->set_stride = 0; return rc; }
    e->set_stride = stride;
    // the shift masks follow the sets (the exact passes evaluate the open shifts only: sc_masked.hip, sc_matrix.hip)
    if (e->smask_cap < stride * scl_engine::kScreenSets) {
        dev_free(e->d_smask); e->d_smask = nullptr; e->smask_cap = 0;
        if ((rc = dev_alloc(e, &e->d_smask, stride * scl_engine::kScreenSets))) { e->set_stride = 0; return rc; }
        e->smask_cap = stride * scl_engine::kScreenSets;
    }
    const size_t need = 2 * stride * sc_screen_scratch_floats(db_view(e), e->SR);   // floats: ring parts x passes x 16 shifts per pair of a launch; two launches' worth (a launch's finishing rides in the next launch)
    if (need > e->part_cap) {
        dev_free(e->d_part); e->part_cap = 0;
        if ((rc = dev_alloc(e, &e->d_part, need))) { e->set_stride = 0; return rc; }
        e->part_cap = need;
    }
    return SCL_OK;
}

int ensure_pinned(scl_engine *e, size_t bytes)
{
    if (bytes <= e->pinned_cap) return SCL_OK;
    if (e->h_pinned) { (void)hipHostFree(e->h_pinned); e->h_pinned = nullptr; e->pinned_cap = 0; }
    SCL_HIP(e, hipHostMalloc(&e->h_pinned, bytes, hipHostMallocDefault));
    memset(e->h_pinned, 0, bytes);                          // (a polled sequence word must never start out as somebody's old number)
    e->pinned_cap = bytes;
    return SCL_OK;
}

// ingest `count` wire descriptors already in e->d_vals into slots [first, first+count)
int ingest_from_vals(scl_engine *e, int count, int first_slot)
{
    ProfScope ps(e, P_INGEST);
    SCL_HIP(e, launch_ingest(e->d_vals, count, first_slot, e->d_desc, e->d_vkey, e->d_norm, e->d_rkey,
                             e->d_rkey4, e->d_hdesc, e->d_kmask, e->d_hkey, e->hstride, e->cap, e->R, e->S, e->stream, e->d_halign));
    e->db_version++;                                       // the alt lane orders itself behind this write
    return SCL_OK;
}

int append_meta(scl_engine *e, int8_t robot, int index)
{
    e->robots.push_back(robot);
    e->indexs.push_back(index);
    e->n++;
    return SCL_OK;
}

// the batch scatter's tiles (make_sc.hip): kMaxScBatch polar images in their initial state between calls
int ensure_tiles(scl_engine *e, hipStream_t s)
{
    int rc;
    if (!e->d_tiles) {
        if ((rc = dev_alloc(e, &e->d_tiles, (size_t)2 * kMaxScBatch * e->R * e->S))) return rc;   // two sets: the resident path scatters group g + 1 beside the ingest of group g
        e->tiles_clean = false;
    }
    if (!e->tiles_clean) {                                   // first use, or a call that failed between scatter and consumer
        SCL_HIP(e, launch_make_sc_tiles_init(e->d_tiles, 2 * kMaxScBatch, e->R, e->S, s));
        e->tiles_clean = true;
    }
    return SCL_OK;
}

int check_cloud(scl_engine *e, const void *points, int n_points, int stride_bytes)
{
    if (n_points < 0 || stride_bytes < 12 || (stride_bytes & 3)) return fail(e, SCL_ERR_INVALID_ARG, "bad point layout");
    if (n_points > 0 && !points) return fail(e, SCL_ERR_INVALID_ARG, "null points");
    return SCL_OK;
}

// K3 for up to kMaxScBatch clouds ALREADY ON THE DEVICE (dptr[i], n[i] points of `stride` bytes): one scatter launch over all of
// them into the tiles.  What consumes the tiles follows: group_ingest (append) or group_values (descriptor only).
int group_scatter(scl_engine *e, const unsigned char *const *dptr, const int *n, int count, int stride, hipStream_t s)
{
    int rc;
    if (count < 1 || count > kMaxScBatch) return fail(e, SCL_ERR_INVALID_ARG, "batch of 1..16 scans");
    if ((rc = ensure_tiles(e, s))) return rc;
    if ((rc = ensure_vals(e, (size_t)e->R * e->S * (size_t)kMaxScBatch))) return rc;
    ScanBatch b{};
    b.count = count;
    uint64_t pts = 0;
    for (int i = 0; i < count; ++i) { b.points[i] = dptr[i]; b.n[i] = n[i]; pts += (uint64_t)n[i]; }
    e->tiles_clean = false;
    ProfScope ps(e, P_MAKESC, s);
    // points of a cloud per workgroup: a workgroup's private tile is merged with up to R*S atomics, so the slice grows with the grid
    // (measured, 16 clouds: 64x120 best at 4 096 of 3 072..16 384; 80x180 at 8 192: 47 us against 60 at 4 096)
    const int slice = e->R * e->S > 10000 ? 2 * kScPointsPerWorkgroup : kScPointsPerWorkgroup;
    SCL_HIP(e, launch_make_sc_batch(b, stride, e->R, e->S, e->cfg.lidar_height, e->cfg.max_radius, e->d_tiles,
                                    scl_lab_int("SCL_SC_SLICE", slice), e->num_cu, s));
    e->prof.make_sc_points += e->prof_on ? pts : 0;
    return SCL_OK;
}

// ... tiles -> database slots [first_slot, first_slot + count) (every array of the slot) + wire-format values in e->d_vals; the tiles
// are initial again afterwards
int group_ingest(scl_engine *e, int count, int first_slot, hipStream_t s)
{
    ProfScope ps(e, P_INGEST, s);
    SCL_HIP(e, launch_ingest(nullptr, count, first_slot, e->d_desc, e->d_vkey, e->d_norm, e->d_rkey, e->d_rkey4, e->d_hdesc, e->d_kmask,
                             e->d_hkey, e->hstride, e->cap, e->R, e->S, s, e->d_halign, e->d_tiles, e->d_vals));
    e->tiles_clean = true;
    e->db_version++;                                       // the alt lane orders itself behind this write
    return SCL_OK;
}

// ... tiles -> wire-format values in e->d_vals only (nothing is stored)
int group_values(scl_engine *e, int count, hipStream_t s)
{
    SCL_HIP(e, launch_make_sc_finalize(e->d_tiles, count, e->R, e->S, e->d_vals, s));
    e->tiles_clean = true;
    return SCL_OK;
}

// one cloud from the host into e->d_points, then the scatter
int scatter_host_cloud(scl_engine *e, const void *points, int n_points, int stride_bytes)
{
    int rc;
    if ((rc = check_cloud(e, points, n_points, stride_bytes))) return rc;
    const size_t bytes = (size_t)n_points * (size_t)stride_bytes;
    if ((rc = ensure_points(e, bytes + 16))) return rc;
    if (bytes) SCL_HIP(e, hipMemcpyAsync(e->d_points, points, bytes, hipMemcpyHostToDevice, e->stream));
    const unsigned char *dp = e->d_points;
    return group_scatter(e, &dp, &n_points, 1, stride_bytes, e->stream);
}

int launch_distance(scl_engine *e, const QueryView &q, const int *d_cand, int slot_base, int n)
{
    ProfScope ps(e, P_SC);
    SCL_HIP(e, launch_sc_distance(db_view(e), q, d_cand, slot_base, n, e->SR, e->d_dist, e->d_shift,
                                  e->num_cu, e->stream));
    if (ps.active()) e->prof.sc_distance_pairs += (uint64_t)n;
    return SCL_OK;
}

int launch_topk(scl_engine *e, const QueryView &q, int lo, int hi, int k, float eps, hipStream_t on = nullptr)
{
    hipStream_t s = on ? on : e->stream;
    ProfScope ps(e, P_TOPK, s);
    SCL_HIP(e, launch_ringkey_topk(db_view(e), q.rkey, lo, hi, k, eps, e->d_topk_scratch,
                                   e->d_topk_idx, e->d_topk_d2, s));
    return SCL_OK;
}

// top-k in [lo,hi) followed by the SC distance of those k candidates; enqueue puts the launches and the copies
// into pinned memory on the engine's stream, finish waits for them and unpacks (the sharded front enqueues on
// every device before it waits on any).
// The ring-key scan leaves per-workgroup lists; the kernel that consumes the k nearest merges them in its prologue
// (topk_merge.hpp): the candidates' kernel of the reference-faithful detection, or the merge + pack kernel of the bare search.
int topk_enqueue_locked(scl_engine *e, int query, int lo, int hi, int k, float eps, bool want_dist, bool *have_dist)
{
    if (k <= 0 || k > kTopkMaxK) return fail(e, SCL_ERR_INVALID_ARG, "k out of range (1..64)");
    QueryView q;
    int rc = query_view(e, query, &q);
    if (rc) return rc;
    if (lo < 0) lo = 0;
    if (hi > e->n) hi = e->n;
    if ((rc = ensure_pairs(e, (size_t)k))) return rc;
    int n_lists = 0;
    {
        ProfScope ps(e, P_TOPK);
        SCL_HIP(e, launch_ringkey_lists(db_view(e), q.rkey, lo, hi, k, eps, e->d_topk_scratch, &n_lists, e->stream));
    }
    *have_dist = hi > lo && want_dist;
    const size_t need = cand_seq_offset(k) + 16;
    if ((rc = ensure_pinned(e, need))) return rc;
    e->pinned_seq = 0;
    if (*have_dist && k <= 16 && sc_cand_exact_supported(db_view(e), e->SR) && !scl_lab_int("SCL_CAND_EXACT_OFF", 0)) {
        // the reference-faithful detection's k (3) candidates: one workgroup merges the scan's lists, aligns and scores the candidates
        // and writes the block (sc_masked.hip)
        const bool fused = n_lists * k <= kCandMergeMaxKeys;
        if (!fused) SCL_HIP(e, launch_topk_merge(e->d_topk_scratch, n_lists, k, e->d_topk_idx, e->d_topk_d2, nullptr, e->stream));
        ProfScope ps(e, P_SC);
        if (++e->out_seq == 0) ++e->out_seq;
        e->pinned_seq = e->out_seq;
        // the word sits at an offset that depends on k: a call with a larger k left idx[] / d2[] data there -- cleared before the launch
        *reinterpret_cast<volatile unsigned int *>(static_cast<char *>(e->h_pinned) + cand_seq_offset(k)) = 0u;
        std::atomic_thread_fence(std::memory_order_release);
        SCL_HIP(e, launch_sc_cand_exact(db_view(e), q, e->SR, k, e->d_topk_idx, e->d_topk_d2, e->h_pinned, e->stream, e->pinned_seq,
                                        fused ? e->d_topk_scratch : nullptr, n_lists, e->d_topk_idx, e->d_topk_d2));
        if (ps.active()) e->prof.sc_distance_pairs += (uint64_t)k;
        return SCL_OK;
    }
    if (*have_dist) {
        SCL_HIP(e, launch_topk_merge(e->d_topk_scratch, n_lists, k, e->d_topk_idx, e->d_topk_d2, nullptr, e->stream));
        if ((rc = launch_distance(e, q, e->d_topk_idx, 0, k))) return rc;
        // the k results travel as ONE block written by a kernel into pinned memory (four copies of a few bytes cost more than the search)
        SCL_HIP(e, launch_topk_pack(e->d_topk_idx, e->d_topk_d2, e->d_dist, e->d_shift, k, true, e->h_pinned, e->stream));
        return SCL_OK;
    }
    // the bare search: merge and block in one launch
    SCL_HIP(e, launch_topk_merge(e->d_topk_scratch, n_lists, k, e->d_topk_idx, e->d_topk_d2, e->h_pinned, e->stream));
    return SCL_OK;
}

int topk_finish_locked(scl_engine *e, int k, bool have_dist, int *idx, float *d2, double *dist, int *shift, int *found)
{
    int rc;
    char *h = static_cast<char *>(e->h_pinned);
    bool seen = false;
    if (e->pinned_seq) {
        // the candidates' kernel writes this call's number behind the block: seen here 4-6 us before an event behind the launch would fire
        const volatile unsigned int *w = reinterpret_cast<const volatile unsigned int *>(h + cand_seq_offset(k));
        const auto t0 = std::chrono::steady_clock::now();
        int spins = 0;
        seen = true;
        while (*w != e->pinned_seq)
            if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) { seen = false; break; }
        std::atomic_thread_fence(std::memory_order_acquire);      // the block behind the word is read below through plain pointers
        if (seen) collect_profile(e);
    }
    if (!seen && (rc = sync_short(e))) return rc;
    const int *h_idx = reinterpret_cast<int *>(h);
    const float *h_d2 = reinterpret_cast<float *>(h + sizeof(int) * k);
    const double *h_dist = reinterpret_cast<double *>(h + (sizeof(int) + sizeof(float)) * k);
    const int *h_shift = reinterpret_cast<int *>(h + (sizeof(int) + sizeof(float) + sizeof(double)) * k);
    int nf = 0;
    for (int i = 0; i < k; ++i) {
        if (idx) idx[i] = h_idx[i];
        if (d2) d2[i] = h_d2[i];
        if (dist) dist[i] = have_dist ? h_dist[i] : kBigDist;
        if (shift) shift[i] = have_dist ? h_shift[i] : 0;
        if (h_idx[i] >= 0) nf++;
    }
    if (found) *found = nf;
    return SCL_OK;
}

int topk_with_distance_locked(scl_engine *e, int query, int lo, int hi, int k, float eps,
                              int *idx, float *d2, double *dist, int *shift, int *found)
{
    bool have_dist = false;
    int rc = topk_enqueue_locked(e, query, lo, hi, k, eps, dist || shift, &have_dist);
    if (rc) return rc;
    return topk_finish_locked(e, k, have_dist, idx, d2, dist, shift, found);
}

}  // namespace

extern "C" {

const char *scl_status_string(int status)
{
    switch (status) {
    case SCL_OK: return "ok";
    case SCL_ERR_INVALID_ARG: return "invalid argument";
    case SCL_ERR_NO_DEVICE: return "no HIP device";
    case SCL_ERR_HIP: return "HIP runtime error";
    case SCL_ERR_OUT_OF_RANGE: return "index out of range";
    case SCL_ERR_NOMEM: return "out of memory";
    case SCL_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
    }
}

const char *scl_last_error(const scl_engine *e) { return e ? e->last_error.c_str() : "null engine"; }

int scl_abi_version(void) { return 5; }

int scl_default_config(scl_config *c)
{
    if (!c) return SCL_ERR_INVALID_ARG;
    c->num_ring = 20; c->num_sector = 60; c->num_candidates = 3;       /* D.h:1308-1310 */
    c->dist_thres = 0.14; c->lidar_height = 1.65; c->max_radius = 80.0; /* D.h:1311-1313 */
    c->num_exclude_recent = 100; c->tree_making_period = 10;            /* D.h:1314-1315 */
    c->search_ratio = 0.1;                                              /* D.h:1316 */
    c->knn_exclude_eps = 0.0f;
    c->device = 0;
    c->initial_capacity = 4096;
    return SCL_OK;
}

// Every environment switch the library reads selects a correct path, but a leftover one can cost an order of magnitude
// (SCL_SCREEN=0 scores every pair with the exact kernel): the first engine of a process says which ones it found set.
static void log_env_overrides_once()
{
    static std::atomic<bool> done{false};
    if (done.exchange(true)) return;
    static const char *const names[] = {"SCL_SCREEN", "SCL_SCREEN_FORM", "SCL_SCREEN_V2_MIN", "SCL_RCCL_LIB",
                                        // experiments: read by a diagnostics build only (kernels.hpp: scl_lab_int)
                                        "SCL_SCREEN_VARIANT", "SCL_SCREEN_PROBE", "SCL_SCREEN_TAIL", "SCL_SCREEN_FUSE", "SCL_FUSE_PARTS", "SCL_ALIGN_FORM", "SCL_ALIGN_WGS",
                                        "SCL_ALIGN2_WGS", "SCL_ALIGN_SIDE", "SCL_ALIGN_FILTER", "SCL_SC_KERNEL", "SCL_SC_WAVES", "SCL_STAMP", "SCL_ABLATE", "SCL_ALT_LANE",
                                        "SCL_ICP_REDUCE", "SCL_MATRIX_PLAIN", "SCL_MATRIX_KERNEL", "SCL_MATRIX_KR", "SCL_WIDE_EXACT", "SCL_STREAM_EXACT", "SCL_SMALL_EXACT_OFF",
                                        "SCL_SELF_ALIGN_OFF", "SCL_CAND_EXACT_OFF"};
    constexpr int kProduct = 4;
    int i = 0;
    for (const char *n : names) {
        const char *v = getenv(n);
#ifdef SCL_DIAGNOSTICS
        if (v) fprintf(stderr, "scl_engine (diagnostics build): environment override in effect: %s=%s\n", n, v);
#else
        if (v && i < kProduct) fprintf(stderr, "scl_engine: environment override in effect: %s=%s\n", n, v);
        else if (v) fprintf(stderr, "scl_engine: %s=%s is an experiment's switch: ignored by this build (make EXTRA=-DSCL_DIAGNOSTICS builds them in)\n", n, v);
#endif
        ++i;
    }
    (void)kProduct;
}

int scl_create(const scl_config *cfg, scl_engine **out)
{
    if (!cfg || !out) return SCL_ERR_INVALID_ARG;
    *out = nullptr;
    log_env_overrides_once();
    if (cfg->num_ring < 1 || cfg->num_ring > 256 || cfg->num_sector < 1 || cfg->num_sector > 1024 ||
        cfg->num_candidates < 1 || cfg->num_candidates > kTopkMaxK || cfg->tree_making_period < 1 ||
        cfg->num_exclude_recent < 0 || !(cfg->max_radius > 0.0) || !(cfg->search_ratio >= 0.0))
        return SCL_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SCL_ERR_NO_DEVICE;
    if (cfg->device < 0 || cfg->device >= ndev) return SCL_ERR_INVALID_ARG;
    scl_engine *e = new (std::nothrow) scl_engine();
    if (!e) return SCL_ERR_NOMEM;
    e->cfg = *cfg;
    e->R = cfg->num_ring; e->S = cfg->num_sector;
    e->RG = (e->R + 3) / 4; e->R4 = e->RG * 4;
    e->hstride = hdesc_stride(e->RG, e->S);
    e->hkw = hkey_row_halfs(e->S);
    e->SR = (int)std::round(0.5 * cfg->search_ratio * (double)e->S);   /* D.h:1545 */
    if (e->SR < 0) e->SR = 0;
    e->device = cfg->device;
    int rc = SCL_OK;
    auto bail = [&](int code) { scl_destroy(e); return code; };
    if (hipSetDevice(e->device) != hipSuccess) return bail(SCL_ERR_HIP);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, e->device) == hipSuccess && prop.multiProcessorCount > 0)
        e->num_cu = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) return bail(SCL_ERR_HIP);
    {   // the ring-key scan is small: give its queue priority so it slips in beside the SC-distance kernel
        int lo_p = 0, hi_p = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
        if (hipStreamCreateWithPriority(&e->stream2, hipStreamNonBlocking, hi_p) != hipSuccess) return bail(SCL_ERR_HIP);
    }
    if (hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) != hipSuccess) return bail(SCL_ERR_HIP);
    if (hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) != hipSuccess) return bail(SCL_ERR_HIP);
    if (hipHostMalloc((void **)&e->h_out3, 64 * scl_engine::kSlots, hipHostMallocDefault) != hipSuccess) return bail(SCL_ERR_HIP);
    memset(e->h_out3, 0, 64 * scl_engine::kSlots);          // o[4] of a record is a polled sequence number
    for (int i = 0; i < scl_engine::kSlots; ++i)
        if (hipEventCreateWithFlags(&e->ev_done[i], hipEventDisableTiming) != hipSuccess) return bail(SCL_ERR_HIP);
    if ((rc = ensure_capacity(e, 1))) return bail(rc);
    if ((rc = dev_alloc(e, &e->d_topk_scratch, (size_t)kTopkMaxBlocks * kTopkMaxK + 2))) return bail(rc);   // partial lists + the scan's ticket counter
    if (hipMemset(e->d_topk_scratch, 0, sizeof(unsigned long long) * ((size_t)kTopkMaxBlocks * kTopkMaxK + 2)) != hipSuccess) return bail(SCL_ERR_HIP);
    if ((rc = dev_alloc(e, &e->d_topk_idx, (size_t)kTopkMaxK * scl_engine::kScreenSets))) return bail(rc);
    if ((rc = dev_alloc(e, &e->d_topk_d2, (size_t)kTopkMaxK * scl_engine::kScreenSets))) return bail(rc);
    if ((rc = dev_alloc(e, &e->d_out3, (size_t)4))) return bail(rc);
    if ((rc = dev_alloc(e, &e->d_blk_part, (size_t)kTailBlocks * kTailRec * kMaxQueryBatch))) return bail(rc);
    static_assert(kMaxQueryBatch <= 4, "done_counter holds four counters");
    if ((rc = dev_alloc(e, &e->d_done_counter, (size_t)4))) return bail(rc);
    if (hipMemset(e->d_done_counter, 0, 16) != hipSuccess) return bail(SCL_ERR_HIP);
    {
        constexpr int NS = scl_engine::kScreenSets;
        if ((rc = dev_alloc(e, &e->d_nsurv, (size_t)NS))) return bail(rc);
        static_assert(NS <= kTminEpsOffset, "the launch's largest bound sits kTminEpsOffset words behind its t_min word");
        if ((rc = dev_alloc(e, &e->d_tmin, (size_t)NS + kTminEpsOffset))) return bail(rc);
        if (hipMemset(e->d_nsurv, 0, sizeof(int) * NS) != hipSuccess) return bail(SCL_ERR_HIP);
        if (hipMemset(e->d_tmin, 0xff, sizeof(unsigned int) * NS) != hipSuccess) return bail(SCL_ERR_HIP);
        if (hipMemset(e->d_tmin + NS, 0, sizeof(unsigned int) * kTminEpsOffset) != hipSuccess) return bail(SCL_ERR_HIP);
        if ((rc = dev_alloc(e, &e->d_surv_part, (size_t)NS * kSurvivorBlocks * kTailRec))) return bail(rc);
        if ((rc = dev_alloc(e, &e->d_surv_done, (size_t)NS))) return bail(rc);
        if (hipMemset(e->d_surv_done, 0, sizeof(unsigned int) * NS) != hipSuccess) return bail(SCL_ERR_HIP);
        if (hipMalloc(&e->d_surv_args, (size_t)8 * NS * kSurvivorArgBytes) != hipSuccess) return bail(SCL_ERR_NOMEM);
        if (hipHostMalloc(&e->h_surv_args, (size_t)8 * NS * kSurvivorArgBytes, hipHostMallocDefault) != hipSuccess) return bail(SCL_ERR_HIP);
        if (hipHostMalloc((void **)&e->h_stream_out, (size_t)2 * NS * 64, hipHostMallocDefault) != hipSuccess) return bail(SCL_ERR_HIP);
        for (auto &ev : e->ev_chunk) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return bail(SCL_ERR_HIP);
        for (auto &ev : e->ev_k1) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return bail(SCL_ERR_HIP);
        for (auto &ev : e->ev_sub0) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return bail(SCL_ERR_HIP);
        if (hipEventCreateWithFlags(&e->ev_align_gate, hipEventDisableTiming) != hipSuccess) return bail(SCL_ERR_HIP);
        {   // lowest priority: its workgroups take the slots the products leave, not the other way round
            int lo_p = 0, hi_p = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo_p, &hi_p);
            // (80 x 180: the exact pass is a dozen short launches per chunk with a handful of survivors each, and the next-but-one chunk is
            //  submitted when it has ended: starved by the products -- 150-180 us per masked launch instead of 35 -- it ended when the
            //  next chunk's screening did and the main stream ran dry for 60 us per chunk.  There it goes FIRST.)
            const bool wide_grid = e->S == 180 && e->RG == 20;
            if (hipStreamCreateWithPriority(&e->stream_surv, hipStreamNonBlocking, wide_grid ? hi_p : lo_p) != hipSuccess) return bail(SCL_ERR_HIP);
            if (hipStreamCreateWithPriority(&e->stream_align, hipStreamNonBlocking, lo_p) != hipSuccess) return bail(SCL_ERR_HIP);
            if (hipEventCreateWithFlags(&e->ev_afork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e->ev_ajoin, hipEventDisableTiming) != hipSuccess) return bail(SCL_ERR_HIP);
        }
    }
    if ((rc = dev_alloc(e, &e->a_blk_part, (size_t)1024 * kTailRec))) return bail(rc);
    if ((rc = dev_alloc(e, &e->a_done_counter, (size_t)4))) return bail(rc);
    if (hipMemset(e->a_done_counter, 0, 16) != hipSuccess) return bail(SCL_ERR_HIP);
    if ((rc = dev_alloc(e, &e->a_topk_idx, (size_t)kTopkMaxK))) return bail(rc);
    if ((rc = dev_alloc(e, &e->a_topk_d2, (size_t)kTopkMaxK))) return bail(rc);
    if (hipStreamCreateWithFlags(&e->stream_alt, hipStreamNonBlocking) != hipSuccess) return bail(SCL_ERR_HIP);
    e->alt_lane = scl_lab_int("SCL_ALT_LANE", 0) == 1;
    if (hipEventCreateWithFlags(&e->ev_db, hipEventDisableTiming) != hipSuccess) return bail(SCL_ERR_HIP);
    if ((rc = dev_alloc(e, &e->d_align_fallbacks, (size_t)1))) return bail(rc);
    if (hipMemset(e->d_align_fallbacks, 0, sizeof(unsigned long long)) != hipSuccess) return bail(SCL_ERR_HIP);
    if ((rc = dev_alloc(e, &e->d_surv_stats, (size_t)3))) return bail(rc);
    if (hipMemset(e->d_surv_stats, 0, 3 * sizeof(unsigned long long)) != hipSuccess) return bail(SCL_ERR_HIP);
    if ((rc = ensure_pairs(e, 1024))) return bail(rc);
    e->screen = sc_screen_supported(db_view(e), e->SR) && cfg->num_candidates <= kTailTopMaxK;
    if ((rc = ensure_pinned(e, 1 << 16))) return bail(rc);
    if ((rc = ensure_vals(e, (size_t)e->R * e->S))) return bail(rc);
    *out = e;
    return SCL_OK;
}

int scl_destroy(scl_engine *e)
{
#ifdef SCL_DIAGNOSTICS
    if (e && !e->front && scl_lab_int("SCL_INGEST_STAMPS", 0)) { (void)hipSetDevice(e->device); scl::ingest_stamps_print(); scl::cand_stamps_print(); }
#endif
    if (!e) return SCL_OK;
    if (e->front) return front_destroy(e);
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (auto &p : e->pending) { (void)hipEventDestroy(p.start); (void)hipEventDestroy(p.stop); }
    for (auto ev : e->event_pool) (void)hipEventDestroy(ev);
    icp_workspace_free(&e->icp_ws);
    icp_workspace_free(&e->vox_ws);
    for (void *slab : e->kf_slabs) (void)hipFree(slab);
    for (int i = 0; i < scl_engine::kIcpBatch; ++i) icp_workspace_free(&e->icp_batch_ws[i]);
    for (int i = 0; i < scl_engine::kIcpLanes; ++i) icp_workspace_free(&e->vox_lane_ws[i]);
    icp_workspace_free(&e->icp_batch_ctl);
    for (int i = 0; i < scl_engine::kIcpLanes; ++i) {
        if (e->ev_lane[i]) (void)hipEventDestroy(e->ev_lane[i]);
        icp_workspace_free(&e->icp_lane_ws[i]);
        if (e->icp_lane_stream[i]) (void)hipStreamDestroy(e->icp_lane_stream[i]);
    }
    dev_free(e->d_desc); dev_free(e->d_vkey); dev_free(e->d_norm); dev_free(e->d_rkey); dev_free(e->d_rkey4);
    dev_free(e->d_hdesc); dev_free(e->d_kmask); dev_free(e->d_hkey); dev_free(e->d_halign);
    dev_free(e->d_vals); dev_free(e->d_points); dev_free(e->d_tiles);
    for (int b = 0; b < 3; ++b) {
        dev_free(e->d_pbuf[b]);
        for (int c = 0; c < scl_engine::kCopyStreams; ++c) if (e->ev_copied[c][b]) (void)hipEventDestroy(e->ev_copied[c][b]);
        if (e->ev_consumed[b]) (void)hipEventDestroy(e->ev_consumed[b]);
    }
    if (e->stream_copy) { (void)hipStreamSynchronize(e->stream_copy); (void)hipStreamDestroy(e->stream_copy); }
    for (hipStream_t cs : e->stream_copy_x) if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
    for (void *hp : e->host_allocs) (void)hipHostFree(hp);
    dev_free(e->d_dist); dev_free(e->d_shift); dev_free(e->d_cand); dev_free(e->d_ring_d2);
    dev_free(e->d_approx); dev_free(e->d_starts); dev_free(e->d_surv); dev_free(e->d_nsurv); dev_free(e->d_tmin); dev_free(e->d_part);
    dev_free(e->d_align_fallbacks);
    dev_free(e->d_smask);
    if (e->d_mat_dist) (void)hipFree(e->d_mat_dist);
    if (e->d_mat_shift) (void)hipFree(e->d_mat_shift);
    if (e->h_mat) (void)hipHostFree(e->h_mat);
    for (auto &ev : e->ev_mat_k) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : e->ev_mat_c) if (ev) (void)hipEventDestroy(ev);
    dev_free(e->d_surv_stats);
    dev_free(e->d_surv_part); dev_free(e->d_surv_done);
    if (e->d_surv_args) (void)hipFree(e->d_surv_args);
    if (e->h_surv_args) (void)hipHostFree(e->h_surv_args);
    if (e->h_stream_out) (void)hipHostFree(e->h_stream_out);
    for (auto ev : e->ev_chunk) if (ev) (void)hipEventDestroy(ev);
    if (e->stream_surv) { (void)hipStreamSynchronize(e->stream_surv); (void)hipStreamDestroy(e->stream_surv); }
    if (e->stream_align) { (void)hipStreamSynchronize(e->stream_align); (void)hipStreamDestroy(e->stream_align); }
    if (e->ev_afork) (void)hipEventDestroy(e->ev_afork);
    if (e->ev_ajoin) (void)hipEventDestroy(e->ev_ajoin);
    for (auto ev : e->ev_k1) if (ev) (void)hipEventDestroy(ev);
    for (auto ev : e->ev_sub0) if (ev) (void)hipEventDestroy(ev);
    if (e->ev_align_gate) (void)hipEventDestroy(e->ev_align_gate);
    dev_free(e->d_topk_scratch); dev_free(e->d_topk_idx); dev_free(e->d_topk_d2); dev_free(e->d_out3);
    dev_free(e->d_blk_part); dev_free(e->d_done_counter);
    dev_free(e->a_blk_part); dev_free(e->a_done_counter); dev_free(e->a_topk_idx); dev_free(e->a_topk_d2);
    dev_free(e->a_dist); dev_free(e->a_shift); dev_free(e->a_ring_d2);
    if (e->ev_db) (void)hipEventDestroy(e->ev_db);
    if (e->stream_alt) { (void)hipStreamSynchronize(e->stream_alt); (void)hipStreamDestroy(e->stream_alt); }
    if (e->h_pinned) (void)hipHostFree(e->h_pinned);
    if (e->h_out3) (void)hipHostFree(e->h_out3);
    if (e->ev_call) (void)hipEventDestroy(e->ev_call);
    for (int i = 0; i < scl_engine::kSlots; ++i) if (e->ev_done[i]) (void)hipEventDestroy(e->ev_done[i]);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->stream2) { (void)hipStreamSynchronize(e->stream2); (void)hipStreamDestroy(e->stream2); }
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return SCL_OK;
}

/* ---- the six virtuals ------------------------------------------------------ */

int scl_make_and_save(scl_engine *e, const void *points, int n_points, int stride_bytes,
                      int8_t robot, int index, float *out_values)
{
    if (!e) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_make_and_save(e, points, n_points, stride_bytes, robot, index, out_values, false, 0.f, nullptr);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    int rc;
    if ((rc = ensure_capacity(e, e->n + 1))) return rc;
    if ((rc = scatter_host_cloud(e, points, n_points, stride_bytes))) return rc;
    if ((rc = group_ingest(e, 1, e->n, e->stream))) return rc;
    if (out_values)
        SCL_HIP(e, hipMemcpyAsync(out_values, e->d_vals, sizeof(float) * (size_t)e->R * e->S,
                                  hipMemcpyDeviceToHost, e->stream));
    if ((rc = sync_short(e))) return rc;                    // (tens of microseconds of device time: polled, not slept on)
    return append_meta(e, robot, index);
}

int scl_make_and_save_filtered(scl_engine *e, const void *points, int n_points, int stride_bytes, float leaf,
                               int8_t robot, int index, float *out_values, int *n_filtered)
{
    if (!e) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_make_and_save(e, points, n_points, stride_bytes, robot, index, out_values, true, leaf, n_filtered);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    int rc;
    if ((rc = check_cloud(e, points, n_points, stride_bytes))) return rc;
    if ((rc = ensure_capacity(e, e->n + 1))) return rc;
    // makeDescriptors, DM.h:996-1002: VoxelGrid(descriptLeafSize) then makeAndSaveDescriptorAndKey -- the filtered cloud
    // never leaves the device
    std::string err;
    const void *d_cloud = nullptr;
    int m = 0;
    if ((rc = voxel_grid_to_device(&e->vox_ws, e->stream, points, n_points, stride_bytes, leaf, &d_cloud, &m, &err))) { e->last_error = err; return rc; }
    const unsigned char *dp = static_cast<const unsigned char *>(d_cloud);
    if ((rc = group_scatter(e, &dp, &m, 1, stride_bytes, e->stream))) return rc;
    if ((rc = group_ingest(e, 1, e->n, e->stream))) return rc;
    if (out_values)
        SCL_HIP(e, hipMemcpyAsync(out_values, e->d_vals, sizeof(float) * (size_t)e->R * e->S,
                                  hipMemcpyDeviceToHost, e->stream));
    if ((rc = sync(e))) return rc;
    if (n_filtered) *n_filtered = m;
    return append_meta(e, robot, index);
}

int scl_make_descriptor(scl_engine *e, const void *points, int n_points, int stride_bytes, float *out_values)
{
    if (!e || !out_values) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    int rc;
    if ((rc = scatter_host_cloud(e, points, n_points, stride_bytes))) return rc;
    if ((rc = group_values(e, 1, e->stream))) return rc;
    SCL_HIP(e, hipMemcpyAsync(out_values, e->d_vals, sizeof(float) * (size_t)e->R * e->S,
                              hipMemcpyDeviceToHost, e->stream));
    return sync(e);
}

int scl_save_from_wire(scl_engine *e, const float *values, int8_t robot, int index)
{
    return scl_save_bulk(e, values, 1, &robot, &index);
}

int scl_save_bulk(scl_engine *e, const float *values, int count, const int8_t *robots, const int *indexs)
{
    if (!e || count < 0 || (count > 0 && !values)) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_save_bulk(e, values, count, robots, indexs);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    int rc;
    if ((rc = ensure_capacity(e, e->n + count))) return rc;
    const size_t cells = (size_t)e->R * e->S;
    const int chunk = 2048;
    if ((rc = ensure_vals(e, cells * (size_t)(count < chunk ? count : chunk)))) return rc;
    for (int done = 0; done < count; done += chunk) {
        const int c = count - done < chunk ? count - done : chunk;
        SCL_HIP(e, hipMemcpyAsync(e->d_vals, values + (size_t)done * cells, sizeof(float) * cells * c,
                                  hipMemcpyHostToDevice, e->stream));
        if ((rc = ingest_from_vals(e, c, e->n + done))) return rc;
        if ((rc = sync(e))) return rc;     // d_vals is reused by the next chunk
    }
    for (int i = 0; i < count; ++i) {
        e->robots.push_back(robots ? robots[i] : (int8_t)0);
        e->indexs.push_back(indexs ? indexs[i] : e->n + i);
    }
    e->n += count;
    return SCL_OK;
}

int scl_stage_query(scl_engine *e, const float *values)
{
    if (!e || !values) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_stage_query(e, values);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    int rc;
    const size_t cells = (size_t)e->R * e->S;
    if ((rc = ensure_vals(e, cells))) return rc;
    SCL_HIP(e, hipMemcpyAsync(e->d_vals, values, sizeof(float) * cells, hipMemcpyHostToDevice, e->stream));
    {
        ProfScope ps(e, P_INGEST);
        // staging slot 0 = database index cap; no tiled ring key for staged queries (rkey4 == nullptr)
        SCL_HIP(e, launch_ingest(e->d_vals, 1, e->cap, e->d_desc, e->d_vkey, e->d_norm, e->d_rkey, nullptr, e->d_hdesc, e->d_kmask, e->d_hkey, e->hstride,
                                 e->cap, e->R, e->S, e->stream));
    }
    if ((rc = sync(e))) return rc;
    e->staged[0] = true;
    return SCL_OK;
}

int scl_detect_intra(scl_engine *e, int cur, int *loop_id, float *shift, double *dist)
{
    if (!e || !loop_id || !shift) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_detect_intra(e, cur, loop_id, shift, dist);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    *loop_id = -1; *shift = 0.0f;                                         /* D.h:1615 */
    if (dist) *dist = kBigDist;
    if (cur < 0 || cur >= e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "cur out of range");
    const int k = e->cfg.num_candidates;
    if (cur < e->cfg.num_exclude_recent + k + 1) return SCL_OK;           /* D.h:1620-1623 */
    const int history = cur - e->cfg.num_exclude_recent;                  /* D.h:1627 */
    int idx[kTopkMaxK]; float d2[kTopkMaxK]; double cd[kTopkMaxK]; int ca[kTopkMaxK];
    int rc = topk_with_distance_locked(e, cur, 0, history, k, e->cfg.knn_exclude_eps, idx, d2, cd, ca, nullptr);
    if (rc) return rc;
    float minDis = 10000000.0f;                                           /* D.h:1637: a float */
    int minIndex = -1, minBias = 0;
    for (int i = 0; i < k; ++i) {                                         /* D.h:1645-1659 */
        if (idx[i] < 0) continue;
        if (cd[i] < (double)minDis) {                                     /* D.h:1653 */
            minDis = (float)cd[i];                                        /* D.h:1655 narrowing */
            minIndex = idx[i];
            minBias = ca[i];
        }
    }
    if (dist) *dist = (double)minDis;
    if ((double)minDis < e->cfg.dist_thres) {                             /* D.h:1662 */
        *loop_id = minIndex;
        *shift = (float)minBias;                                          /* D.h:1665 */
    }
    return SCL_OK;
}

int scl_detect_inter(scl_engine *e, int cur, int *loop_id, float *yaw_rad, double *dist)
{
    if (!e || !loop_id || !yaw_rad) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_detect_inter(e, cur, loop_id, yaw_rad, dist);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    *loop_id = -1; *yaw_rad = 0.0f;                                       /* D.h:1678,1686 */
    if (dist) *dist = kBigDist;
    if (cur < 0 || cur >= e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "cur out of range");
    if (e->n < e->cfg.num_exclude_recent + 1) return SCL_OK;              /* D.h:1684-1688 */
    if (e->tree_counter % e->cfg.tree_making_period == 0)                 /* D.h:1691-1702 */
        e->tree_n = e->n - e->cfg.num_exclude_recent;
    e->tree_counter = e->tree_counter + 1;                                /* D.h:1703 */
    const int k = e->cfg.num_candidates;
    int idx[kTopkMaxK]; float d2[kTopkMaxK]; double cd[kTopkMaxK]; int ca[kTopkMaxK];
    int rc = topk_with_distance_locked(e, cur, 0, e->tree_n, k, 0.0f, idx, d2, cd, ca, nullptr);
    if (rc) return rc;
    // slots the search left unfilled read as index 0 in the reference (zero-initialised
    // candidate_indexes, D.h:1710): score slot 0 for them.
    bool need0 = false;
    for (int i = 0; i < k; ++i) need0 |= idx[i] < 0;
    double cd0 = kBigDist; int ca0 = 0;
    if (need0 && e->tree_n > 0) {
        QueryView q;
        if ((rc = query_view(e, cur, &q))) return rc;
        if ((rc = launch_distance(e, q, nullptr, 0, 1))) return rc;
        SCL_HIP(e, hipMemcpyAsync(e->h_pinned, e->d_dist, sizeof(double), hipMemcpyDeviceToHost, e->stream));
        SCL_HIP(e, hipMemcpyAsync((char *)e->h_pinned + 8, e->d_shift, sizeof(int), hipMemcpyDeviceToHost, e->stream));
        if ((rc = sync(e))) return rc;
        cd0 = *reinterpret_cast<double *>(e->h_pinned);
        ca0 = *reinterpret_cast<int *>((char *)e->h_pinned + 8);
    }
    double min_dist = 10000000;                                           /* D.h:1705 */
    int nn_align = 0, nn_idx = -1;
    for (int i = 0; i < k; ++i) {                                         /* D.h:1721-1737 */
        const int ci = idx[i] < 0 ? 0 : idx[i];
        const double c = idx[i] < 0 ? cd0 : cd[i];
        const int al = idx[i] < 0 ? ca0 : ca[i];
        if (c < min_dist) {
            if (ci == cur) continue;                                      /* D.h:1731 */
            min_dist = c; nn_align = al; nn_idx = ci;
        }
    }
    if (min_dist < e->cfg.dist_thres) *loop_id = nn_idx;                  /* D.h:1741-1744 */
    const double unit_sector_angle = 360.0 / (double)e->S;                /* D.h:1332 */
    *yaw_rad = (float)(nn_align * unit_sector_angle * M_PI / 180.0);      /* D.h:1752 */
    if (dist) *dist = min_dist;
    return SCL_OK;
}

int scl_get_index(const scl_engine *e, int key, int8_t *robot, int *index)
{
    if (!e || !robot || !index) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_get_index(e, key, robot, index);
    std::lock_guard<std::mutex> lk(e->mu);
    if (key < 0 || key >= e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "key out of range");
    *robot = e->robots[key];
    *index = e->indexs[key];
    return SCL_OK;
}

int scl_get_size(const scl_engine *e, int id)
{
    (void)id;                                                             /* D.h:1763-1766 ignores idIn */
    if (!e) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_get_size(e);
    std::lock_guard<std::mutex> lk(e->mu);
    return e->n;
}

/* ---- building blocks -------------------------------------------------------- */

int scl_get_descriptor(const scl_engine *ce, int key, float *values)
{
    scl_engine *e = const_cast<scl_engine *>(ce);
    if (!e || !values) return SCL_ERR_INVALID_ARG;
    if (e->front) { scl_engine *child = nullptr; int slot = 0; const int rc = front_get_slot(e, key, &child, &slot); return rc ? rc : scl_get_descriptor(child, slot, values); }
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    if (key < 0 || key >= e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "key out of range");
    int rc;
    const size_t cells = (size_t)e->R * e->S;
    if ((rc = ensure_vals(e, cells))) return rc;
    SCL_HIP(e, launch_untile(e->d_desc + (size_t)key * e->RG * e->S, e->R, e->S, e->d_vals, e->stream));
    SCL_HIP(e, hipMemcpyAsync(values, e->d_vals, sizeof(float) * cells, hipMemcpyDeviceToHost, e->stream));
    return sync(e);
}

int scl_get_ringkey(const scl_engine *ce, int key, float *ringkey)
{
    scl_engine *e = const_cast<scl_engine *>(ce);
    if (!e || !ringkey) return SCL_ERR_INVALID_ARG;
    if (e->front) { scl_engine *child = nullptr; int slot = 0; const int rc = front_get_slot(e, key, &child, &slot); return rc ? rc : scl_get_ringkey(child, slot, ringkey); }
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    if (key < 0 || key >= e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "key out of range");
    SCL_HIP(e, hipMemcpyAsync(ringkey, e->d_rkey + (size_t)key * e->R4, sizeof(float) * e->R,
                              hipMemcpyDeviceToHost, e->stream));
    return sync(e);
}

int scl_get_sectorkey(const scl_engine *ce, int key, double *sectorkey)
{
    scl_engine *e = const_cast<scl_engine *>(ce);
    if (!e || !sectorkey) return SCL_ERR_INVALID_ARG;
    if (e->front) { scl_engine *child = nullptr; int slot = 0; const int rc = front_get_slot(e, key, &child, &slot); return rc ? rc : scl_get_sectorkey(child, slot, sectorkey); }
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    if (key < 0 || key >= e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "key out of range");
    SCL_HIP(e, hipMemcpyAsync(sectorkey, e->d_vkey + (size_t)key * e->S, sizeof(double) * e->S,
                              hipMemcpyDeviceToHost, e->stream));
    return sync(e);
}

int scl_ringkey_topk(scl_engine *e, int query, int lo, int hi, int k, int *idx, float *d2, int *found)
{
    if (!e || !idx || !d2) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_topk(e, query, lo, hi, k, idx, d2, nullptr, nullptr, found);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    return topk_with_distance_locked(e, query, lo, hi, k, e->cfg.knn_exclude_eps, idx, d2, nullptr, nullptr, found);
}

int scl_topk_with_distance(scl_engine *e, int query, int lo, int hi, int k,
                           int *idx, float *d2, double *dist, int *shift, int *found)
{
    if (!e || !idx || !d2 || !dist || !shift) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_topk(e, query, lo, hi, k, idx, d2, dist, shift, found);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    return topk_with_distance_locked(e, query, lo, hi, k, e->cfg.knn_exclude_eps, idx, d2, dist, shift, found);
}

int scl_sc_distance_batch(scl_engine *e, int query, const int *cand, int n, double *dist, int *shift)
{
    if (!e || n < 0 || !dist || !shift) return SCL_ERR_INVALID_ARG;
    if (n == 0) return SCL_OK;
    if (e->front) return front_sc_distance_batch(e, query, cand, n, dist, shift);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    QueryView q;
    int rc = query_view(e, query, &q);
    if (rc) return rc;
    if (cand) {
        for (int i = 0; i < n; ++i)
            if (cand[i] >= e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "candidate slot out of range");
    } else if (n > e->n) {
        return fail(e, SCL_ERR_OUT_OF_RANGE, "n exceeds database size");
    }
    if ((rc = ensure_pairs(e, (size_t)n))) return rc;
    if (cand) SCL_HIP(e, hipMemcpyAsync(e->d_cand, cand, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, e->stream));
    if ((rc = launch_distance(e, q, cand ? e->d_cand : nullptr, 0, n))) return rc;
    SCL_HIP(e, hipMemcpyAsync(dist, e->d_dist, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
    SCL_HIP(e, hipMemcpyAsync(shift, e->d_shift, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
    return sync(e);
}

namespace { int matrix_screened_locked(scl_engine *e, const int *slots, int nq, int lo, int n, double *dist, int *shift); }

/* The exact distance matrix (north_star: "the column-shifted SC distance matrix over the keyframe database"): rows = queries,
 * columns = keyframes lo .. hi-1, every entry the reference's distanceBtnScanContext (D.h:1538-1569) in fp64 with its shift.
 * Up to kMaxQueryBatch rows per launch of the wave program; a launch's results travel to the host (pinned halves, second
 * stream) while the next launch runs. */
int scl_sc_distance_matrix(scl_engine *e, const int *queries, int nq, int lo, int hi, double *dist, int *shift)
{
    if (!e || nq < 0 || (nq > 0 && (!queries || !dist || !shift))) return SCL_ERR_INVALID_ARG;
    if (nq == 0) return SCL_OK;
    if (e->front) return front_sc_distance_matrix(e, queries, nq, lo, hi, dist, shift);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    if (lo < 0 || hi > e->n || hi < lo) return fail(e, SCL_ERR_OUT_OF_RANGE, "keyframe range out of the database");
    const int n = hi - lo;
    if (n == 0) return SCL_OK;
    std::vector<int> slots((size_t)nq);
    for (int i = 0; i < nq; ++i) {
        const int q = queries[i];
        if (q >= e->n || q < -e->stage_rows) return fail(e, SCL_ERR_OUT_OF_RANGE, "query keyframe out of range");
        if (q < 0 && !e->staged[-1 - q]) return fail(e, SCL_ERR_INVALID_ARG, "no staged query in that slot");
        slots[(size_t)i] = q >= 0 ? q : e->cap + (-1 - q);
    }
    // the screened grids: alignment + screening of 16 rows at a time, then the exact evaluation of the shifts still open (sc_masked.hip)
    if (e->screen && sc_masked_supported(db_view(e), e->SR) && !scl_lab_int("SCL_MATRIX_PLAIN", 0)) return matrix_screened_locked(e, slots.data(), nq, lo, n, dist, shift);
    constexpr int RB = kMaxQueryBatch;                                     // rows per launch
    const size_t row = ((size_t)n + 63) & ~(size_t)63;
    if (e->mat_cap < row) {
        if (e->d_mat_dist) (void)hipFree(e->d_mat_dist);
        if (e->d_mat_shift) (void)hipFree(e->d_mat_shift);
        if (e->h_mat) (void)hipHostFree(e->h_mat);
        e->d_mat_dist = nullptr; e->d_mat_shift = nullptr; e->h_mat = nullptr; e->mat_cap = 0;
        if (hipMalloc((void **)&e->d_mat_dist, sizeof(double) * 2 * RB * row) != hipSuccess ||
            hipMalloc((void **)&e->d_mat_shift, sizeof(int) * 2 * RB * row) != hipSuccess ||
            hipHostMalloc(&e->h_mat, (sizeof(double) + sizeof(int)) * 2 * RB * row, hipHostMallocDefault) != hipSuccess)
            return fail(e, SCL_ERR_NOMEM, "distance matrix buffers");
        e->mat_cap = row;
        for (int h = 0; h < 2; ++h) {
            if (!e->ev_mat_k[h]) SCL_HIP(e, hipEventCreateWithFlags(&e->ev_mat_k[h], hipEventDisableTiming));
            if (!e->ev_mat_c[h]) SCL_HIP(e, hipEventCreateWithFlags(&e->ev_mat_c[h], hipEventDisableTiming));
        }
    }
    const size_t cap = e->mat_cap;
    double *h_dist = static_cast<double *>(e->h_mat);
    int *h_shift = reinterpret_cast<int *>(h_dist + 2 * RB * cap);
    const int groups = (nq + RB - 1) / RB;
    auto deliver = [&](int g) -> int {                                     // group g's rows: pinned half -> the caller's arrays
        const int h = g & 1, r0 = g * RB, rows = nq - r0 < RB ? nq - r0 : RB;
        SCL_HIP(e, hipEventSynchronize(e->ev_mat_c[h]));
        for (int r = 0; r < rows; ++r) {
            std::memcpy(dist + (size_t)(r0 + r) * n, h_dist + ((size_t)h * RB + r) * cap, sizeof(double) * (size_t)n);
            std::memcpy(shift + (size_t)(r0 + r) * n, h_shift + ((size_t)h * RB + r) * cap, sizeof(int) * (size_t)n);
        }
        return SCL_OK;
    };
    int rc = SCL_OK;
    for (int g = 0; g < groups; ++g) {
        const int h = g & 1, r0 = g * RB, rows = nq - r0 < RB ? nq - r0 : RB;
        if (g >= 2 && (rc = deliver(g - 2))) return rc;                    // (also: half h of the device buffers has been copied out)
        {
            ProfScope ps(e, P_SC);
            SCL_HIP(e, launch_sc_distance_matrix(db_view(e), slots.data() + r0, rows, lo, n, e->SR, e->d_mat_dist + (size_t)h * RB * cap,
                                                 e->d_mat_shift + (size_t)h * RB * cap, cap, e->num_cu, e->stream));
            if (ps.active()) e->prof.sc_distance_pairs += (uint64_t)rows * (uint64_t)n;
        }
        SCL_HIP(e, hipEventRecord(e->ev_mat_k[h], e->stream));
        SCL_HIP(e, hipStreamWaitEvent(e->stream2, e->ev_mat_k[h], 0));
        SCL_HIP(e, hipMemcpyAsync(h_dist + (size_t)h * RB * cap, e->d_mat_dist + (size_t)h * RB * cap, sizeof(double) * (size_t)rows * cap, hipMemcpyDeviceToHost, e->stream2));
        SCL_HIP(e, hipMemcpyAsync(h_shift + (size_t)h * RB * cap, e->d_mat_shift + (size_t)h * RB * cap, sizeof(int) * (size_t)rows * cap, hipMemcpyDeviceToHost, e->stream2));
        SCL_HIP(e, hipEventRecord(e->ev_mat_c[h], e->stream2));
    }
    for (int g = groups >= 2 ? groups - 2 : 0; g < groups; ++g)
        if ((rc = deliver(g))) return rc;
    if (e->prof_on) collect_profile(e);
    return SCL_OK;
}

namespace {

// enqueue one full-DB pass; results land in pinned slot `sl` when ev_done[sl] has fired
int submit_full_many_locked(scl_engine *e, const int *queries, const int *los, const int *his, int nq, int *tickets);

// ---- screened full-DB passes (64x120 grid; sc_screen.hip) --------------------------------------------------------
// One screening launch for up to four queries: slots qslot[i] against [lo[i], lo[i] + n[i]), results in buffer sets
// set0 + i.  The event pair of the profile brackets this launch: it is the dominant kernel of a pass.
// One screening launch group: queries qslot[0..nq) against [lo[i], lo[i] + n[i]), buffer sets set0 + i.
struct ScreenGroup { const int *qslot, *lo, *n; int nq, set0; int part_half = 0; bool masks = false; bool keys_later = false; };   // part_half: which half of d_part holds the batch's partial sums;
                                                                                                             // masks: the batch's exact pass evaluates the open shifts only (the shift masks are formed and written)
                                                                                                             // keys_later: ScreenBatch::no_ring_metric
// next (optional, nq > 0): the launch that will follow; its alignment rides in this one (the next call then passes
// kScreenProducts only).
int launch_screen_group(scl_engine *e, const ScreenGroup &cur, int phases = kScreenAlign | kScreenProducts,
                        const ScreenGroup *next = nullptr, hipStream_t stream = nullptr, const ScreenGroup *prev = nullptr)
{
    if (!stream) stream = e->stream;
    auto fill = [&](ScreenBatch &sb, const ScreenGroup &g) {
        sb.nq = g.nq;
        for (int j = 0; j < g.nq; ++j) { sb.slot[j] = g.qslot[j]; sb.base[j] = g.lo[j]; sb.n[j] = g.n[j]; sb.buf[j] = g.set0 + j; }
        sb.pair_stride = e->set_stride;
        sb.approx = e->d_approx; sb.starts = e->d_starts; sb.align_fallbacks = e->d_align_fallbacks; sb.ring_d2 = e->d_ring_d2;
        sb.survivors = nullptr; sb.n_surv = nullptr; sb.t_min = e->d_tmin;
        sb.smask = (g.masks || sc_screen_is_wide(db_view(e), e->SR)) ? e->d_smask : nullptr;   // (the 64 x 120 stream's exact pass scores all 13 shifts of its few survivors: no masks)
        sb.part = e->d_part + (g.part_half ? e->part_cap / 2 : 0);
        sb.no_ring_metric = g.keys_later;
        sb.side = stream == e->stream ? e->stream_align : nullptr; sb.ev_fork = e->ev_afork; sb.ev_join = e->ev_ajoin;
        sb.k = e->cfg.num_candidates; sb.exclude_eps = e->cfg.knn_exclude_eps; sb.topk_idx = e->d_topk_idx; sb.topk_d2 = e->d_topk_d2;
    };
    ScreenBatch sb{}, nx{}, pv{};
    fill(sb, cur);
    const bool has_next = next && next->nq > 0;
    if (has_next) fill(nx, *next);
    const bool has_prev = prev && prev->nq > 0;
    if (has_prev) fill(pv, *prev);
    if (phases == kScreenFinish) {                          // a deferred finishing on its own (the end of a stream)
        SCL_HIP(e, launch_sc_screen_batch(db_view(e), sb, e->SR, sc_align_filter_enabled(), e->num_cu, stream, kScreenFinish, nullptr, nullptr));
        return SCL_OK;
    }
    if (phases & kScreenAlign) for (int j = 0; j < cur.nq; ++j) e->align_pairs += (uint64_t)cur.n[j];
    if (has_next) for (int j = 0; j < next->nq; ++j) e->align_pairs += (uint64_t)next->n[j];
    if ((phases & kScreenAlign) && (phases & kScreenProducts) && has_next) {   // a sequence's first launch: its own alignment outside the
        SCL_HIP(e, launch_sc_screen_batch(db_view(e), sb, e->SR, sc_align_filter_enabled(), e->num_cu, stream, kScreenAlign, nullptr));   // event pair
        phases = kScreenProducts;
    }
    ProfScope ps(e, P_SC, stream);
    SCL_HIP(e, launch_sc_screen_batch(db_view(e), sb, e->SR, sc_align_filter_enabled(), e->num_cu, stream, phases, has_next ? &nx : nullptr, has_prev ? &pv : nullptr));
    if (ps.active()) { for (int j = 0; j < cur.nq; ++j) e->prof.sc_distance_pairs += (uint64_t)cur.n[j]; }
    return SCL_OK;
}

// The exact pass over the survivors of nq screened queries (buffer sets set0 .. set0 + nq - 1); winners to out3[i].
// phases / region: see launch_sc_distance_survivors; a caller that splits the pass passes the same region (from
// survivor_arg_region) to both halves
unsigned survivor_arg_region(scl_engine *e) { return e->surv_arg_tick++ & 7u; }
int launch_survivor_pass(scl_engine *e, const int *qslot, const int *lo, const int *n, int nq, int set0, double *const *out3, hipStream_t stream = nullptr,
                         int phases = kSurvivorArgs | kSurvivorKernel, int region_in = -1, bool ring_from_keys = false)
{
    if (!stream) stream = e->stream;
    SurvivorPass sp{};
    sp.nq = nq; sp.ring_from_keys = ring_from_keys ? 1 : 0;
    for (int j = 0; j < nq; ++j) { sp.slot[j] = qslot[j]; sp.base[j] = lo[j]; sp.n[j] = n[j]; sp.buf[j] = set0 + j; sp.out3[j] = out3[j]; }
    sp.pair_stride = e->set_stride;
    sp.approx = e->d_approx; sp.survivors = e->d_surv; sp.t_min = e->d_tmin; sp.out_dist = e->d_dist; sp.out_shift = e->d_shift;
    sp.blk_part = e->d_surv_part; sp.done_counter = e->d_surv_done; sp.surv_stats = e->d_surv_stats;
    sp.ring_d2 = e->d_ring_d2; sp.k = e->cfg.num_candidates; sp.exclude_eps = e->cfg.knn_exclude_eps; sp.topk_idx = e->d_topk_idx; sp.topk_d2 = e->d_topk_d2;
    const size_t region = (size_t)(region_in >= 0 ? (unsigned)region_in : survivor_arg_region(e)) * scl_engine::kScreenSets * kSurvivorArgBytes;   // 8 passes may be in flight
    sp.d_args = static_cast<char *>(e->d_surv_args) + region; sp.h_args = static_cast<char *>(e->h_surv_args) + region;
    if (!(phases & kSurvivorKernel)) {
        SCL_HIP(e, launch_sc_distance_survivors(db_view(e), sp, e->SR, e->num_cu, stream, phases));
        return SCL_OK;
    }
    ProfScope ps(e, P_ARGMIN, stream);
    SCL_HIP(e, launch_sc_distance_survivors(db_view(e), sp, e->SR, e->num_cu, stream, phases));
    return SCL_OK;
}

// The same pass by sc_small_exact_kernel (sc_masked.hip): ONE workgroup per scan lists the scan's survivors, scores them (every shift:
// the stream's launches form no shift masks), forms the ring-key top-k and writes the winner.  Where the survivors' kernel above holds
// 128 workgroups of 160 KB of LDS for 130 us beside the screening launches -- whose products need every CU: the launch that met them took
// 63 us instead of 38, once per chunk -- this one holds 64 CUs for a fifth of that.  Argument sets through the same regions.
int launch_small_exact_chunk(scl_engine *e, const int *qslot, const int *lo, const int *n, int nq, int set0, double *const *out3, hipStream_t stream,
                             int phases, int region_in)
{
    static_assert(sizeof(SmallExactQuery) <= kSurvivorArgBytes && kMaxSmallExactQueries <= scl_engine::kScreenSets, "argument regions");
    if (nq < 1 || nq > kMaxSmallExactQueries) return fail(e, SCL_ERR_INVALID_ARG, "exact pass: too many scans for one launch");
    const size_t region = (size_t)(region_in >= 0 ? (unsigned)region_in : survivor_arg_region(e)) * scl_engine::kScreenSets * kSurvivorArgBytes;
    SmallExactQuery *h = reinterpret_cast<SmallExactQuery *>(static_cast<char *>(e->h_surv_args) + region);
    SmallExactQuery *d = reinterpret_cast<SmallExactQuery *>(static_cast<char *>(e->d_surv_args) + region);
    const bool wide_masks = sc_screen_is_wide(db_view(e), e->SR) && e->d_smask;   // (80 x 180: every launch forms the shift masks)
    if (phases & kSurvivorArgs) {
        for (int j = 0; j < nq; ++j) {
            const size_t set = (size_t)(set0 + j), off = set * e->set_stride;
            SmallExactQuery &sq = h[j];
            sq.qslot = qslot[j]; sq.base = lo[j]; sq.n = n[j];
            sq.approx = e->d_approx + off; sq.starts = e->d_starts + off; sq.smask = wide_masks ? e->d_smask + off : nullptr; sq.ring_d2 = e->d_ring_d2 + off;
            sq.t_min = e->d_tmin + set; sq.list = e->d_surv + off; sq.out3 = out3[j];
            sq.topk_idx = e->d_topk_idx + set * kTailTopMaxK; sq.topk_d2 = e->d_topk_d2 + set * kTailTopMaxK;
        }
        SCL_HIP(e, hipMemcpyAsync(d, h, sizeof(SmallExactQuery) * (size_t)nq, hipMemcpyHostToDevice, stream));
    }
    if (!(phases & kSurvivorKernel)) return SCL_OK;
    SmallExactArgs sa{};
    sa.nq = nq; sa.k = e->cfg.num_candidates; sa.exclude_eps = e->cfg.knn_exclude_eps; sa.two_eps = 2.0f * sc_screen_eps(); sa.surv_stats = e->d_surv_stats;
    sa.q_dev = d;
    sa.every_shift = wide_masks ? 0 : 1;
    ProfScope ps(e, P_ARGMIN, stream);
    SCL_HIP(e, launch_sc_small_exact(db_view(e), e->SR, sa, stream));
    return SCL_OK;
}

// The exact pass of the 80 x 180 grid over the screened queries of buffer sets set0 .. set0 + nq - 1: per group of up to four
// queries the select launch (survivor lists + ring-key top-k), the one-sector-per-lane program on the survivors and the arg-min.
// first_done (optional): recorded behind the pass's first group of kWideExactBatch scans
int launch_survivor_pass_wide(scl_engine *e, const int *qslot, const int *lo, const int *n, int nq, int set0, double *const *out3, hipStream_t stream, hipEvent_t first_done = nullptr)
{
    ProfScope ps(e, P_ARGMIN, stream);
    for (int g = 0; g < nq; g += kWideExactBatch) {
        const int w = nq - g < kWideExactBatch ? nq - g : kWideExactBatch;
        ScreenBatch sb{};
        sb.nq = w;
        for (int j = 0; j < w; ++j) { sb.slot[j] = qslot[g + j]; sb.base[j] = lo[g + j]; sb.n[j] = n[g + j]; sb.buf[j] = set0 + g + j; }
        sb.pair_stride = e->set_stride;
        sb.approx = e->d_approx; sb.starts = e->d_starts; sb.align_fallbacks = e->d_align_fallbacks; sb.ring_d2 = e->d_ring_d2;
        sb.survivors = e->d_surv; sb.n_surv = e->d_nsurv; sb.t_min = e->d_tmin; sb.surv_stats = e->d_surv_stats;
        sb.k = e->cfg.num_candidates; sb.exclude_eps = e->cfg.knn_exclude_eps; sb.topk_idx = e->d_topk_idx; sb.topk_d2 = e->d_topk_d2;
        SCL_HIP(e, launch_sc_select_batch(sb, stream));
        const int *sv[kWideExactBatch]; const int *ns[kWideExactBatch]; double *od[kWideExactBatch]; int *os[kWideExactBatch];
        const int *st[kWideExactBatch]; const unsigned int *sm[kWideExactBatch];
        for (int j = 0; j < w; ++j) {
            const size_t set = (size_t)(set0 + g + j);
            sv[j] = e->d_surv + set * e->set_stride; ns[j] = e->d_nsurv + set;
            od[j] = e->d_dist + set * e->set_stride; os[j] = e->d_shift + set * e->set_stride;
            st[j] = e->d_starts + set * e->set_stride; sm[j] = e->d_smask ? e->d_smask + set * e->set_stride : nullptr;
        }
        SCL_HIP(e, launch_sc_distance_survivors_wide(db_view(e), w, qslot + g, lo + g, e->SR, sv, ns, od, os, out3 + g, e->num_cu, stream,
                                                     e->d_smask ? st : nullptr, e->d_smask ? sm : nullptr));
        if (g == 0 && first_done) SCL_HIP(e, hipEventRecord(first_done, stream));
    }
    return SCL_OK;
}

// experiments (diagnostics builds only): SCL_MATRIX_KERNEL=masked keeps round 3's one-wave-per-pair kernel, SCL_MATRIX_KR = keyframes per workgroup
static bool matrix_masked_kernel()
{
    return scl_lab_is("SCL_MATRIX_KERNEL", "m");
}
static int matrix_kr()
{
    return scl_lab_int("SCL_MATRIX_KR", 0);
}

// The exact distance matrix on a screened grid.  Per group of up to `mb` rows (a screening launch's worth): alignment, screening
// products and finishing -- which leaves, per pair, the first shift and the mask of the shifts within 2 eps of the pair's smallest
// screened distance -- then sc_masked_kernel evaluates exactly those shifts in fp64.  Every entry is the reference's distance and
// shift, bit for bit; the results of a group travel to the host while the next group runs.
int matrix_screened_locked(scl_engine *e, const int *slots, int nq, int lo, int n, double *dist, int *shift)
{
    const int mb = sc_screen_max_batch(db_view(e), e->SR);
    const bool wide = sc_screen_is_wide(db_view(e), e->SR);
    const int v2_min = wide ? 2 : 4;                                        // smaller batches take the first form, which leaves no masks: padded
    int rc = ensure_sets(e, (size_t)n);
    if (rc) return rc;
    const size_t need = e->set_stride * scl_engine::kScreenSets;
    if (e->smask_cap < need) {
        dev_free(e->d_smask); e->d_smask = nullptr; e->smask_cap = 0;
        if ((rc = dev_alloc(e, &e->d_smask, need))) return rc;
        e->smask_cap = need;
    }
    const size_t row = ((size_t)n + 63) & ~(size_t)63;
    const int RB = kMaxScreenBatch;
    if (e->mat_cap < row * (size_t)(RB / kMaxQueryBatch)) {                  // (the plain form sizes the halves for kMaxQueryBatch rows)
        if (e->d_mat_dist) (void)hipFree(e->d_mat_dist);
        if (e->d_mat_shift) (void)hipFree(e->d_mat_shift);
        if (e->h_mat) (void)hipHostFree(e->h_mat);
        e->d_mat_dist = nullptr; e->d_mat_shift = nullptr; e->h_mat = nullptr; e->mat_cap = 0;
        const size_t cap_rows = (size_t)2 * RB;
        if (hipMalloc((void **)&e->d_mat_dist, sizeof(double) * cap_rows * row) != hipSuccess ||
            hipMalloc((void **)&e->d_mat_shift, sizeof(int) * cap_rows * row) != hipSuccess ||
            hipHostMalloc(&e->h_mat, (sizeof(double) + sizeof(int)) * cap_rows * row, hipHostMallocDefault) != hipSuccess)
            return fail(e, SCL_ERR_NOMEM, "distance matrix buffers");
        e->mat_cap = row * (size_t)(RB / kMaxQueryBatch);
        for (int h = 0; h < 2; ++h) {
            if (!e->ev_mat_k[h]) SCL_HIP(e, hipEventCreateWithFlags(&e->ev_mat_k[h], hipEventDisableTiming));
            if (!e->ev_mat_c[h]) SCL_HIP(e, hipEventCreateWithFlags(&e->ev_mat_c[h], hipEventDisableTiming));
        }
    }
    const size_t cap = row;
    double *h_dist = static_cast<double *>(e->h_mat);
    int *h_shift = reinterpret_cast<int *>(h_dist + (size_t)2 * RB * cap);
    const int groups = (nq + mb - 1) / mb;
    auto deliver = [&](int g) -> int {
        const int h = g & 1, r0 = g * mb, rows = nq - r0 < mb ? nq - r0 : mb;
        SCL_HIP(e, hipEventSynchronize(e->ev_mat_c[h]));
        for (int r = 0; r < rows; ++r) {
            std::memcpy(dist + (size_t)(r0 + r) * n, h_dist + ((size_t)h * RB + r) * cap, sizeof(double) * (size_t)n);
            std::memcpy(shift + (size_t)(r0 + r) * n, h_shift + ((size_t)h * RB + r) * cap, sizeof(int) * (size_t)n);
        }
        return SCL_OK;
    };
    // Consecutive groups alternate between the engine's stream and its second lane (buffer sets, partial sums and output halves of their
    // own) on the 80 x 180 grid: a group is alignment, products and finishing -- launches that each ramp up and drain -- in front of the exact
    // kernel, and on one stream the chip went through them with nothing beside them; now they run under the group before's exact kernel:
    // 97-104 -> 108-116 M pairs/s.  On 64 x 120 the same measured 206-214 -> 194-196 M at 64 rows and nothing at 256: there the exact kernel
    // lives off its keyframes staying in the XCDs' L2s (sc_matrix.hip), which a products launch beside it streams the database through.
    for (int g = 0; g < groups; ++g) {
        const int h = g & 1, r0 = g * mb, rows = nq - r0 < mb ? nq - r0 : mb;
        const int lane = (groups > 1 && wide) ? h : 0, set0 = lane * mb;   // (64 x 120: one lane -- see above)
        hipStream_t ks = lane ? e->stream_alt : e->stream;
        if (lane && e->alt_seen_version != e->db_version) {                  // descriptors written on `stream` since the last pass on the second lane
            SCL_HIP(e, hipEventRecord(e->ev_db, e->stream));
            SCL_HIP(e, hipStreamWaitEvent(e->stream_alt, e->ev_db, 0));
            e->alt_seen_version = e->db_version;
        }
        int qs[kMaxScreenBatch], los[kMaxScreenBatch], ns[kMaxScreenBatch];
        const int padded = rows < v2_min ? v2_min : rows;                   // (the padding rows repeat the last one; their results are dropped)
        for (int j = 0; j < padded; ++j) { qs[j] = slots[r0 + (j < rows ? j : rows - 1)]; los[j] = lo; ns[j] = n; }
        {
            ProfScope ps(e, P_SC, ks);
            ScreenGroup grp{qs, los, ns, padded, set0};
            grp.masks = true; grp.part_half = lane;
            const int prof_saved = e->prof_on;
            e->prof_on = 0;                                                  // (one event pair around the whole group: this scope's)
            rc = launch_screen_group(e, grp, kScreenAlign | kScreenProducts, nullptr, ks);
            e->prof_on = prof_saved;
            if (rc) return rc;
            const int *g_starts = e->d_starts + (size_t)set0 * e->set_stride;
            const unsigned int *g_smask = e->d_smask + (size_t)set0 * e->set_stride;
            MaskedQuery mq[kMaxMaskedQueries];
            for (int j = 0; j < rows; ++j) {
                mq[j].qslot = qs[j]; mq[j].slot_base = lo; mq[j].n = n; mq[j].n_dev = nullptr; mq[j].cand = nullptr;
                mq[j].starts = g_starts + (size_t)j * e->set_stride; mq[j].smask = g_smask + (size_t)j * e->set_stride;
                mq[j].out_dist = e->d_mat_dist + ((size_t)h * RB + j) * cap; mq[j].out_shift = e->d_mat_shift + ((size_t)h * RB + j) * cap;
            }
            if (sc_matrix_supported(db_view(e), e->SR) && !matrix_masked_kernel()) {
                SCL_HIP(e, launch_sc_matrix(db_view(e), e->SR, qs, rows, lo, n, g_starts, g_smask, e->set_stride,
                                            e->d_mat_dist + (size_t)h * RB * cap, e->d_mat_shift + (size_t)h * RB * cap, cap, matrix_kr(), ks));
            } else {
                int parts = 2 * e->num_cu / rows; parts = parts < 1 ? 1 : parts;
                const int max_parts = (n + 7) / 8; parts = parts > max_parts ? max_parts : parts;
                SCL_HIP(e, launch_sc_masked(db_view(e), e->SR, mq, rows, parts, ks));
            }
            if (ps.active()) e->prof.sc_distance_pairs += (uint64_t)rows * (uint64_t)n;
        }
        SCL_HIP(e, hipEventRecord(e->ev_mat_k[h], ks));
        SCL_HIP(e, hipStreamWaitEvent(e->stream2, e->ev_mat_k[h], 0));
        SCL_HIP(e, hipMemcpyAsync(h_dist + (size_t)h * RB * cap, e->d_mat_dist + (size_t)h * RB * cap, sizeof(double) * (size_t)rows * cap, hipMemcpyDeviceToHost, e->stream2));
        SCL_HIP(e, hipMemcpyAsync(h_shift + (size_t)h * RB * cap, e->d_mat_shift + (size_t)h * RB * cap, sizeof(int) * (size_t)rows * cap, hipMemcpyDeviceToHost, e->stream2));
        SCL_HIP(e, hipEventRecord(e->ev_mat_c[h], e->stream2));
        // group g is enqueued: hand over group g - 1 now (its copy ends while g runs; the half it leaves is g + 1's) -- only the last
        // group's copy and hand-over are left behind the last kernel
        if (g >= 1 && (rc = deliver(g - 1))) return rc;
    }
    if ((rc = deliver(groups - 1))) return rc;
    if (e->prof_on) collect_profile(e);
    return SCL_OK;
}

int submit_full_locked(scl_engine *e, int query, int lo, int hi, int *ticket)
{
    if (e->screen && !e->in_single_fallback) {             // one query through the screening pipeline of the batched form
        e->in_single_fallback = true;
        const int rc = submit_full_many_locked(e, &query, &lo, &hi, 1, ticket);
        e->in_single_fallback = false;
        return rc;
    }
    QueryView q;
    int rc = query_view(e, query, &q);
    if (rc) return rc;
    const int sl = (int)(e->next_slot % scl_engine::kSlots);
    if (e->slot_busy[sl]) return fail(e, SCL_ERR_INVALID_ARG, "too many full-DB passes in flight: collect first");
    if (lo < 0) lo = 0;
    if (hi > e->n) hi = e->n;
    const int n = hi - lo;
    e->slot_lo[sl] = lo;
    e->slot_empty[sl] = n <= 0;
    double *out3 = e->h_out3 + (size_t)sl * 8;
    bool use_alt = false;
    if (n > 0) {
        if ((rc = ensure_pairs(e, (size_t)n))) return rc;
        // "full ring-key + shifted SC distance per incoming scan".  On the two-sectors-per-lane grids the
        // ring-key metric is evaluated inside the SC-distance kernel (the wave that scores a slot also reads
        // its ring key) and one epilogue launch does arg-min + top-k; otherwise the stand-alone ring-key
        // scan runs on a second stream beside the SC kernel.
        const int k = e->cfg.num_candidates;
        const bool fuse = k <= kTailTop && sc_distance_fuses_ring(db_view(e), e->SR);
        if (!fuse) SCL_HIP(e, hipEventRecord(e->ev_fork, e->stream));
        bool fused = false;
        // every other fused pass on a database-resident query runs on the alt lane (see scl_engine::stream_alt)
        use_alt = e->alt_lane && fuse && query >= 0 && (e->next_slot & 1u);
        if (use_alt) {
            if ((size_t)n > e->a_pair_cap) {
                dev_free(e->a_dist); dev_free(e->a_shift); dev_free(e->a_ring_d2);
                const size_t nn = (size_t)n + (size_t)n / 2 + 64;
                e->a_pair_cap = 0;
                if ((rc = dev_alloc(e, &e->a_ring_d2, nn))) return rc;
                if ((rc = dev_alloc(e, &e->a_dist, nn))) return rc;
                if ((rc = dev_alloc(e, &e->a_shift, nn))) return rc;
                e->a_pair_cap = nn;
            }
            if (e->alt_seen_version != e->db_version) {               // descriptors written on `stream` since the last alt pass
                SCL_HIP(e, hipEventRecord(e->ev_db, e->stream));
                SCL_HIP(e, hipStreamWaitEvent(e->stream_alt, e->ev_db, 0));
                e->alt_seen_version = e->db_version;
            }
        }
        hipStream_t ks = use_alt ? e->stream_alt : e->stream;
        FullTail tail{use_alt ? e->a_blk_part : e->d_blk_part, use_alt ? e->a_done_counter : e->d_done_counter, out3,
                      use_alt ? e->a_topk_idx : e->d_topk_idx, use_alt ? e->a_topk_d2 : e->d_topk_d2, k, e->cfg.knn_exclude_eps};
        {
            ProfScope ps(e, P_SC, ks);
            SCL_HIP(e, launch_sc_distance(db_view(e), q, nullptr, lo, n, e->SR, use_alt ? e->a_dist : e->d_dist,
                                          use_alt ? e->a_shift : e->d_shift, e->num_cu, ks,
                                          fuse ? (use_alt ? e->a_ring_d2 : e->d_ring_d2) : nullptr, &fused, fuse ? &tail : nullptr));
            if (ps.active()) e->prof.sc_distance_pairs += (uint64_t)n;
        }
        if (fuse && fused) e->last_pass_alt = use_alt;
        if (fuse && fused) {
            // arg-min and top-k were reduced inside the kernel (last workgroup)
        } else {
            if (fuse) return fail(e, SCL_ERR_HIP, "ring-key fusion expected but not provided by the kernel");
            SCL_HIP(e, hipStreamWaitEvent(e->stream2, e->ev_fork, 0));
            if ((rc = launch_topk(e, q, lo, hi, k, e->cfg.knn_exclude_eps, e->stream2))) return rc;
            SCL_HIP(e, hipEventRecord(e->ev_join, e->stream2));
            {
                ProfScope ps(e, P_ARGMIN);
                SCL_HIP(e, launch_argmin(e->d_dist, e->d_shift, n, out3, e->stream));   // writes pinned host memory
            }
            SCL_HIP(e, hipStreamWaitEvent(e->stream, e->ev_join, 0));
        }
    }
    e->last_pass_empty = n <= 0;                            // nothing ran: the top-k buffers still hold an older pass
    SCL_HIP(e, hipEventRecord(e->ev_done[sl], use_alt ? e->stream_alt : e->stream));
    e->slot_ev[sl] = sl; e->slot_seq[sl] = 0;
    e->slot_busy[sl] = true;
    e->next_slot++;
    *ticket = sl;
    return SCL_OK;
}

// nq database-resident queries, one launch (workgroups [i*nb, (i+1)*nb) serve query i); every query gets its own
// result slot, all of them complete with the one event recorded behind the launch.
int submit_full_many_locked(scl_engine *e, const int *queries, const int *los, const int *his, int nq, int *tickets)
{
    const int k = e->cfg.num_candidates;
    const bool wide = e->screen && sc_screen_is_wide(db_view(e), e->SR);      // 80 x 180
    bool batchable = (nq > 1 || e->screen) && nq <= kMaxQueryBatch && (k <= kTailTop || e->screen) && (sc_distance_fuses_ring(db_view(e), e->SR) || wide);
    int qslot[kMaxQueryBatch] = {0};                       // database index of every query (staging slot j = cap + j)
    for (int i = 0; i < nq && batchable; ++i) {
        const int q = queries[i];
        if (q >= 0) { batchable = q < e->n; qslot[i] = q; }
        else { const int j = -1 - q; batchable = j < e->stage_rows && e->staged[j]; qslot[i] = e->cap + j; }
    }
    if (!batchable) {                                      // one pass per query
        for (int i = 0; i < nq; ++i) { int rc = submit_full_locked(e, queries[i], los[i], his[i], &tickets[i]); if (rc) return rc; }
        return SCL_OK;
    }
    for (int i = 0; i < nq; ++i)
        if (e->slot_busy[(e->next_slot + (unsigned)i) % scl_engine::kSlots])
            return fail(e, SCL_ERR_INVALID_ARG, "too many full-DB passes in flight: collect first");
    QueryBatch qb{};
    unsigned int seq_of_launch = 0;                        // (the small exact pass: the number it writes behind every result record)
    int first = -1, nmax = 0;
    int lo_of[kMaxQueryBatch]; bool empty_of[kMaxQueryBatch];
    for (int i = 0; i < nq; ++i) {
        const int sl = (int)((e->next_slot + (unsigned)i) % scl_engine::kSlots);
        int lo = los[i] < 0 ? 0 : los[i], hi = his[i] > e->n ? e->n : his[i];
        const int n = hi - lo;
        tickets[i] = sl;
        lo_of[i] = lo; empty_of[i] = n <= 0;
        if (first < 0) first = sl;
        if (n > 0) {
            const int j = qb.nq++;
            qb.slot[j] = qslot[i]; qb.base[j] = lo; qb.n[j] = n; qb.out3[j] = e->h_out3 + (size_t)sl * 8;
            nmax = n > nmax ? n : nmax;
        }
    }
    if (qb.nq > 0) {
        int rc = e->screen ? SCL_OK : ensure_pairs(e, (size_t)nmax * qb.nq);
        if (rc) return rc;
        qb.pair_stride = (size_t)nmax;
        FullTail tail{e->d_blk_part, e->d_done_counter, nullptr, e->d_topk_idx, e->d_topk_d2, k, e->cfg.knn_exclude_eps};
        if (e->screen) {
            // screening pass (fp16 matrix-core bounds around every reference distance) -> the exact fp64 kernel on the
            // survivors only; the winner is the reference's, bit for bit (sc_screen.hip)
            if ((rc = ensure_sets(e, (size_t)nmax))) return rc;
            ScreenGroup sg{qb.slot, qb.base, qb.n, qb.nq, 0};
            sg.masks = !wide && sc_small_exact_supported(db_view(e), e->SR);
            if ((rc = launch_screen_group(e, sg))) return rc;
            const bool small_pass = (sg.masks || wide) && sc_small_exact_supported(db_view(e), e->SR) && e->d_smask && !scl_lab_int("SCL_SMALL_EXACT_OFF", 0);
            if (wide && !small_pass) {
                if ((rc = launch_survivor_pass_wide(e, qb.slot, qb.base, qb.n, qb.nq, 0, qb.out3, e->stream))) return rc;
            } else if (small_pass) {
                // a blocking call's handful of scans: one workgroup per scan selects, scores the open shifts, forms the top-k and
                // writes the winner (sc_masked.hip) -- one launch, no argument copy
                SmallExactArgs sa{};
                sa.nq = qb.nq; sa.k = k; sa.exclude_eps = e->cfg.knn_exclude_eps; sa.two_eps = 2.0f * sc_screen_eps(); sa.surv_stats = e->d_surv_stats;
                for (int j = 0; j < qb.nq; ++j) {
                    const size_t off = (size_t)j * e->set_stride;
                    SmallExactQuery &sq = sa.q[j];
                    sq.qslot = qb.slot[j]; sq.base = qb.base[j]; sq.n = qb.n[j];
                    sq.approx = e->d_approx + off; sq.starts = e->d_starts + off; sq.smask = e->d_smask + off; sq.ring_d2 = e->d_ring_d2 + off;
                    sq.t_min = e->d_tmin + j; sq.list = e->d_surv + off; sq.out3 = qb.out3[j];
                    sq.topk_idx = e->d_topk_idx + j * kTailTopMaxK; sq.topk_d2 = e->d_topk_d2 + j * kTailTopMaxK;
                }
                if (++e->out_seq == 0) ++e->out_seq;
#ifndef SCL_NO_SEQ
                sa.seq = seq_of_launch = e->out_seq;
#endif
                ProfScope ps(e, P_ARGMIN);
                SCL_HIP(e, launch_sc_small_exact(db_view(e), e->SR, sa, e->stream));
            } else if ((rc = launch_survivor_pass(e, qb.slot, qb.base, qb.n, qb.nq, 0, qb.out3))) return rc;
        } else {
            ProfScope ps(e, P_SC);
            SCL_HIP(e, launch_sc_distance_batch(db_view(e), qb, e->SR, e->d_dist, e->d_shift, e->d_ring_d2, tail, e->num_cu, e->stream));
            if (ps.active()) { for (int j = 0; j < qb.nq; ++j) e->prof.sc_distance_pairs += (uint64_t)qb.n[j]; }
        }
        e->last_pass_alt = false;
    }
    e->last_pass_empty = qb.nq == 0 || empty_of[0];
    SCL_HIP(e, hipEventRecord(e->ev_done[first], e->stream));
    // slot state changes only now that the launch and its event are enqueued (an error above leaves every slot free)
    for (int i = 0; i < nq; ++i) {
        e->slot_lo[tickets[i]] = lo_of[i]; e->slot_empty[tickets[i]] = empty_of[i];
        e->slot_ev[tickets[i]] = first; e->slot_busy[tickets[i]] = true;
        e->slot_seq[tickets[i]] = empty_of[i] ? 0u : seq_of_launch;
    }
    e->next_slot += (unsigned)nq;
    return SCL_OK;
}

int collect_full_locked(scl_engine *e, int ticket, int *nn_idx, int *shift, double *dist)
{
    if (ticket < 0 || ticket >= scl_engine::kSlots || !e->slot_busy[ticket])
        return fail(e, SCL_ERR_INVALID_ARG, "unknown ticket");
    {
        // a pass of one or a few scans ends within tens of microseconds: poll for that long (a blocking wait wakes up tens of
        // microseconds late -- a fifth of a blocking one-scan call), then sleep in the runtime
        hipEvent_t ev = e->ev_done[e->slot_ev[ticket]];
        hipError_t q = hipErrorNotReady;
        const auto t0 = std::chrono::steady_clock::now();
        const unsigned int seq = e->slot_empty[ticket] ? 0u : e->slot_seq[ticket];
        if (seq) {
            // the exact pass's workgroup writes this launch's number behind the record in pinned memory: seen here 4-6 us before the
            // event's packet fires.  (The launches behind it on the stream are ordered by the stream; nothing else of this pass is read here.)
            const volatile double *o = e->h_out3 + (size_t)ticket * 8;
            int spins = 0;
            q = hipSuccess;
            while (o[4] != (double)seq)
                if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) { q = hipEventSynchronize(ev); break; }
        } else {
            while ((q = hipEventQuery(ev)) == hipErrorNotReady)
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) { q = hipEventSynchronize(ev); break; }
        }
        SCL_HIP(e, q);
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    collect_profile(e);
    e->slot_busy[ticket] = false;
    *nn_idx = -1; *shift = 0; *dist = kBigDist;
    if (e->slot_empty[ticket]) return SCL_OK;
    const volatile double *o = e->h_out3 + (size_t)ticket * 8;
    *dist = o[0];
    *nn_idx = o[1] < 0 ? -1 : e->slot_lo[ticket] + (int)o[1];
    *shift = (int)o[2];
    return SCL_OK;
}

}  // namespace

int scl_detect_full_submit(scl_engine *e, int query, int lo, int hi, int *ticket)
{
    if (!e || !ticket) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_submit_many(e, &query, &lo, &hi, 1, ticket);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    return submit_full_locked(e, query, lo, hi, ticket);
}

int scl_detect_full_submit_many(scl_engine *e, const int *queries, const int *lo, const int *hi, int n_queries, int *tickets)
{
    if (!e || !queries || !lo || !hi || !tickets || n_queries < 1 || n_queries > scl_engine::kSlots) return SCL_ERR_INVALID_ARG;
    if (e->front) {
        for (int i = 0; i < n_queries; i += kMaxQueryBatch) {
            const int m = n_queries - i < kMaxQueryBatch ? n_queries - i : kMaxQueryBatch;
            const int rc = front_submit_many(e, queries + i, lo + i, hi + i, m, tickets + i);
            if (rc) return rc;
        }
        return SCL_OK;
    }
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    for (int i = 0; i < n_queries; i += kMaxQueryBatch) {
        const int m = n_queries - i < kMaxQueryBatch ? n_queries - i : kMaxQueryBatch;
        int rc = submit_full_many_locked(e, queries + i, lo + i, hi + i, m, tickets + i);
        if (rc) return rc;
    }
    return SCL_OK;
}

namespace {

// The stream form on the screened grid: a chunk of up to kScreenSets scans = its screening launches back to back
// (spl scans per launch, one buffer set per scan), then ONE exact pass over the survivors of the whole chunk, one
// event.  Two chunks are kept enqueued (their results land in the two halves of h_stream_out), so the device never
// waits for the host between chunks.
// `db`: the caller's lock on the database (e->mu).  The call scores the database it was started on -- n0 keyframes -- and gives the
// lock up while it waits for a chunk's results: appends (which take e->mu only) get in there, behind the launches already enqueued on
// the stream; a capacity doubling waits for every stream of the engine before it frees the old arrays (grow_to), and the launches
// enqueued after it read the new ones (db_view is formed per launch).  pass_mu, held by the caller throughout, keeps every other
// scoring entry point out of the buffer sets.
int stream_screened_locked(scl_engine *e, std::unique_lock<std::mutex> &db, const int *queries, const int *lo, const int *hi, int n_queries, int spl,
                           int *nn_idx, int *shift, double *dist)
{
    const int n0 = e->n;                                     // the database this call scores
    constexpr int NS = scl_engine::kScreenSets;
    constexpr int CH = NS / 2;                               // scans per chunk at most: the two chunks in flight use the two halves of the buffer sets
    // ... and a whole number of launches (a short launch costs the same pass over the database as a full one)
    static_assert(CH <= kMaxSurvivorQueries, "one exact pass takes the survivors of a whole chunk");
    const bool wide = sc_screen_is_wide(db_view(e), e->SR);  // 80 x 180: same launches, its own exact pass
    // (80 x 180: chunks of 64 -- its exact pass is a chain of launches per 16 scans that runs beside the NEXT chunk's screening, and the
    //  host submits the chunk after that only when it has ended: chunks of 128 measured 3 % slower there, 4 % faster on 64 x 120)
#ifndef SCL_WIDE_CHUNK
#define SCL_WIDE_CHUNK 64
#endif
    const int chmax = wide && CH > SCL_WIDE_CHUNK ? SCL_WIDE_CHUNK : CH;
    const int chn = spl >= 1 && spl <= chmax ? (chmax / spl) * spl : chmax;
    struct List { int qslot[CH], qlo[CH], qn[CH], pos[CH], m = 0; };     // the scans of a chunk that have something to score
    struct Chunk { int first = 0, count = 0; bool busy = false, aligned = false, small = false; std::vector<int> lo, empty; };
    Chunk ch[2];
    // A launch's finishing (bound, flags, ring-key metric: what the exact pass reads) rides in the NEXT launch's extra waves
    // (sc_screen.hip): `pend` is the launch whose finishing is still owed, `owed` the chunk whose exact pass waits for it.
    struct Pending { int qslot[kMaxScreenBatch], lo[kMaxScreenBatch], n[kMaxScreenBatch], nq = 0, set0 = 0, half = 0; bool valid = false; } pend;
    struct Owed { List L; int c = 0, region = 0; bool valid = false, small = false; } owed;
    int part_half = 0;
    bool sub0_valid[2] = {false, false};                    // ev_sub0[c] was recorded by the exact pass that ev_chunk[c] ends
    // 64 x 120: the chunk's exact pass by one workgroup per scan (SCL_STREAM_EXACT=survivors keeps round 3's kernel, which also forms the
    // ring-key metric of its ranges -- keys_later)
    const bool small_exact = sc_small_exact_supported(db_view(e), e->SR) && (!wide || e->d_smask) && !scl_lab_is("SCL_STREAM_EXACT", "s");
    const bool keys_later = !wide && !small_exact;
    auto pend_group = [&]() { ScreenGroup g{pend.qslot, pend.lo, pend.n, pend.nq, pend.set0}; g.part_half = pend.half; g.keys_later = keys_later; return g; };
    // the exact pass of chunk `o.c` on the side stream, behind everything the main stream holds now
    auto run_owed = [&]() -> int {
        if (!owed.valid) return SCL_OK;
        const int oc = owed.c;
        double *out3[CH];
        for (int j = 0; j < owed.L.m; ++j) out3[j] = e->h_stream_out + ((size_t)oc * NS + (size_t)owed.L.pos[j]) * 8;
        SCL_HIP(e, hipEventRecord(e->ev_k1[oc], e->stream));
        SCL_HIP(e, hipStreamWaitEvent(e->stream_surv, e->ev_k1[oc], 0));
        int r2 = SCL_OK;
        if (owed.L.m > 0) r2 = owed.small ? launch_small_exact_chunk(e, owed.L.qslot, owed.L.qlo, owed.L.qn, owed.L.m, oc * CH, out3, e->stream_surv, kSurvivorKernel, owed.region)
                               : wide ? launch_survivor_pass_wide(e, owed.L.qslot, owed.L.qlo, owed.L.qn, owed.L.m, oc * CH, out3, e->stream_surv, e->ev_sub0[oc])
                                    : launch_survivor_pass(e, owed.L.qslot, owed.L.qlo, owed.L.qn, owed.L.m, oc * CH, out3, e->stream_surv, kSurvivorKernel, owed.region, keys_later);
        if (r2) return r2;
        sub0_valid[oc] = wide && !owed.small && owed.L.m > 0;
        SCL_HIP(e, hipEventRecord(e->ev_chunk[oc], e->stream_surv));
        owed.valid = false;
        return SCL_OK;
    };
    // the owed finishing as a launch of its own (nothing follows that could carry it)
    auto flush_pending = [&]() -> int {
        if (!pend.valid) return SCL_OK;
        const ScreenGroup g = pend_group();
        pend.valid = false;
        return launch_screen_group(e, g, kScreenFinish);
    };
    int nmax = 1;
    for (int i = 0; i < n_queries; ++i) {
        const int l = lo[i] < 0 ? 0 : lo[i], h = hi[i] > n0 ? n0 : hi[i];
        if (h - l > nmax) nmax = h - l;
    }
    int rc = ensure_sets(e, (size_t)nmax);
    if (rc) return rc;
    // the side stream starts behind everything the main stream holds (ingests wrote the arrays it reads)
    SCL_HIP(e, hipEventRecord(e->ev_align_gate, e->stream));
    SCL_HIP(e, hipStreamWaitEvent(e->stream_surv, e->ev_align_gate, 0));
    auto build = [&](int first, int count, List &L, Chunk *k) -> int {
        L.m = 0;
        for (int i = 0; i < count; ++i) {
            const int q = queries[first + i];
            int slot;
            if (q >= 0) { if (q >= n0) return fail(e, SCL_ERR_OUT_OF_RANGE, "query slot out of range"); slot = q; }
            else { const int j = -1 - q; if (j >= e->stage_rows || !e->staged[j]) return fail(e, SCL_ERR_INVALID_ARG, "no staged query"); slot = e->cap + j; }
            const int l = lo[first + i] < 0 ? 0 : lo[first + i], h = hi[first + i] > n0 ? n0 : hi[first + i];
            if (k) k->lo[(size_t)i] = l;
            if (h - l <= 0) continue;
            if (k) k->empty[(size_t)i] = 0;
            L.qslot[L.m] = slot; L.qlo[L.m] = l; L.qn[L.m] = h - l; L.pos[L.m] = i;
            ++L.m;
        }
        return SCL_OK;
    };
    // c = half of the buffer sets; first / count = this chunk; nfirst / ncount = the chunk behind it (ncount = 0: none)
    auto submit = [&](int c, int first, int count, int nfirst, int ncount) -> int {
        Chunk &k = ch[c];
        k.first = first; k.count = count; k.lo.assign((size_t)count, 0); k.empty.assign((size_t)count, 1);
        List cur, nxt;
        if ((rc = build(first, count, cur, &k))) return rc;
        if (ncount > 0 && (rc = build(nfirst, ncount, nxt, nullptr))) return rc;
        double *out3[CH];
        for (int j = 0; j < cur.m; ++j) out3[j] = e->h_stream_out + ((size_t)c * NS + (size_t)cur.pos[j]) * 8;
        // buffer set = first set of this half + position among the non-empty scans (consecutive within a launch).  The two
        // halves alternate, and a chunk is submitted only after the chunk two before it was collected.
        const int set0 = c * CH, nset0 = (c ^ 1) * CH;
        // Every launch carries the alignment of the launch behind it -- the chunk's last one that of the next chunk's first
        // launch, whose buffer sets the exact pass of the chunk before this one may still be reading: the main stream waits
        // for it there (it finished long ago).  Only the very first launch aligns for itself.
        // the exact pass's argument sets go to the device now, ahead of the products it has to wait for
        // one workgroup per scan scores its survivors at every shift: right for the handful a scan leaves as a rule; dozens per scan (what
        // the chunks collected last reported) go to the survivors' kernel
        k.small = small_exact && !e->exact_heavy;
        const int region = (cur.m > 0 && (k.small || !wide)) ? (int)survivor_arg_region(e) : 0;
        if (cur.m > 0 && (k.small || !wide) && (rc = k.small ? launch_small_exact_chunk(e, cur.qslot, cur.qlo, cur.qn, cur.m, set0, out3, e->stream_surv, kSurvivorArgs, region)
                                                    : launch_survivor_pass(e, cur.qslot, cur.qlo, cur.qn, cur.m, set0, out3, e->stream_surv, kSurvivorArgs, region, keys_later))) return rc;
        if (ncount == 0 && cur.m > 0) SCL_HIP(e, hipEventRecord(e->ev_k1[c], e->stream_surv));   // the call's last chunk: its exact pass runs on the main stream, behind this copy
        // Sampled profile (scl_profile_enable(3)): ONE event pair around the screening launches of every second chunk, its time divided
        // by the chunk's launch groups -- a pair around a single group keeps that group's two launches back by 4-6 us each and
        // measured 5 us (8 %) more than the kernel trace; the events the chunk needs anyway (the exact pass's, the wait for the buffer
        // sets) are inside the span: they are the stream's, not the measurement's.
        struct ProfRestore { scl_engine *e; int saved; ~ProfRestore() { e->prof_on = saved; } } prof_restore{e, e->prof_on};
        hipEvent_t span_start = nullptr, span_stop = nullptr;
        int span_groups = 0;
        if (e->prof_on == 3 && cur.m > 0) {
            if ((e->prof_tick++ & 1) == 0) {
                auto get = [&]() { hipEvent_t ev = nullptr; if (!e->event_pool.empty()) { ev = e->event_pool.back(); e->event_pool.pop_back(); } else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr; return ev; };
                span_start = get(); span_stop = get();
                if (span_start && span_stop) (void)hipEventRecord(span_start, e->stream); else { span_start = span_stop = nullptr; }
            }
            e->prof_on = 0;                                  // (no pairs around single groups inside)
        }
        auto span_end = [&]() {
            if (!span_start) return;
            (void)hipEventRecord(span_stop, e->stream);
            e->pending.push_back({span_start, span_stop, P_SC, span_groups});
            for (int j = 0; j < cur.m; ++j) e->prof.sc_distance_pairs += (uint64_t)cur.qn[j];
            span_start = nullptr;
        };
        bool next_aligned = false;
        if (cur.m == 0) {                                    // nothing to launch: what earlier chunks are owed cannot ride along
            if ((rc = flush_pending())) return rc;
            if ((rc = run_owed())) return rc;
        }
        for (int g = 0; g < cur.m; g += spl) {
            const int w = cur.m - g < spl ? cur.m - g : spl;
            ScreenGroup grp{cur.qslot + g, cur.qlo + g, cur.qn + g, w, set0 + g};
            grp.part_half = part_half; grp.keys_later = keys_later;
            const bool defer = sc_screen_can_defer(db_view(e), e->SR, w);
            if (!defer && (rc = flush_pending())) return rc;   // (a batch the first form scores has no extra waves to carry it)
            ScreenGroup nx{nullptr, nullptr, nullptr, 0, 0};
            if (g + w < cur.m) {
                const int wn = cur.m - g - w < spl ? cur.m - g - w : spl;
                nx = ScreenGroup{cur.qslot + g + w, cur.qlo + g + w, cur.qn + g + w, wn, set0 + g + w};
                nx.keys_later = keys_later;
            } else if (ncount > 0 && nxt.m > 0) {
                const int wn = nxt.m < spl ? nxt.m : spl;
                nx = ScreenGroup{nxt.qslot, nxt.qlo, nxt.qn, wn, nset0};
                nx.keys_later = keys_later;
                // (80 x 180: the exact pass of a chunk is several launch groups, one per kWideExactBatch buffer sets, and nearly as
                //  long as the next chunk's screening: this launch's alignment writes the first group's sets only and waits for those)
                if (ch[c ^ 1].busy) SCL_HIP(e, hipStreamWaitEvent(e->stream, (wide && sub0_valid[c ^ 1] && wn <= kWideExactBatch) ? e->ev_sub0[c ^ 1] : e->ev_chunk[c ^ 1], 0));
                next_aligned = true;
            }
            const int phases = ((g == 0 && !k.aligned) ? (kScreenAlign | kScreenProducts) : kScreenProducts) | (defer ? kScreenDeferFinish : 0);
            const ScreenGroup pv = pend_group();
            if ((rc = launch_screen_group(e, grp, phases, nx.nq > 0 ? &nx : nullptr, nullptr, pend.valid ? &pv : nullptr))) return rc;
            ++span_groups;
            pend.valid = false;
            if (defer) {
                for (int j = 0; j < w; ++j) { pend.qslot[j] = grp.qslot[j]; pend.lo[j] = grp.lo[j]; pend.n[j] = grp.n[j]; }
                pend.nq = w; pend.set0 = grp.set0; pend.half = part_half; pend.valid = true;
                part_half ^= 1;
            }
            if (g == 0 && (rc = run_owed())) return rc;      // the chunk before this one is finished now: its exact pass may start
        }
        span_end();
        // The exact pass of a chunk runs beside the next chunk's products, on the side stream -- except the call's last one,
        // which nothing follows: it stays on the main stream (no hop between streams in front of it; its argument sets are
        // already on the device: wait for their copy only).
        if (ncount > 0 && pend.valid) {
            // this chunk's last finishing rides in the next chunk's first launch: the exact pass is owed until then
            owed.L = cur; owed.c = c; owed.region = region; owed.valid = true; owed.small = k.small;
            k.busy = true;
            k.aligned = false;
            ch[c ^ 1].aligned = next_aligned;
            e->last_pass_empty = cur.m == 0;
            e->last_pass_alt = false;
            return SCL_OK;
        }
        if ((rc = flush_pending())) return rc;
        hipStream_t xs = ncount > 0 ? e->stream_surv : e->stream;
        if (ncount > 0) {
            SCL_HIP(e, hipEventRecord(e->ev_k1[c], e->stream));
            SCL_HIP(e, hipStreamWaitEvent(e->stream_surv, e->ev_k1[c], 0));
        } else if (cur.m > 0) {
            // behind the copy of the argument sets, enqueued on the side stream before this chunk's launches: long done as a rule, and a
            // wait for a pending event keeps the main stream's next launch back by 5 us -- asked for only if the copy is still on its way
            // (the event was recorded behind the copy, at the start of this chunk's submission)
            if (hipEventQuery(e->ev_k1[c]) != hipSuccess) SCL_HIP(e, hipStreamWaitEvent(e->stream, e->ev_k1[c], 0));
        }
        sub0_valid[c] = wide && !k.small && cur.m > 0;
        if (cur.m > 0 && (rc = k.small ? launch_small_exact_chunk(e, cur.qslot, cur.qlo, cur.qn, cur.m, set0, out3, xs, kSurvivorKernel, region)
                             : wide ? launch_survivor_pass_wide(e, cur.qslot, cur.qlo, cur.qn, cur.m, set0, out3, xs, e->ev_sub0[c])
                                    : launch_survivor_pass(e, cur.qslot, cur.qlo, cur.qn, cur.m, set0, out3, xs, kSurvivorKernel, region, keys_later))) return rc;
        SCL_HIP(e, hipEventRecord(e->ev_chunk[c], xs));
        k.busy = true;
        k.aligned = false;
        ch[c ^ 1].aligned = next_aligned;                    // the next chunk's first launch needs no alignment of its own
        e->last_pass_empty = cur.m == 0;
        e->last_pass_alt = false;
        return SCL_OK;
    };
    auto collect = [&](int c, bool last) -> int {
        Chunk &k = ch[c];
        // A chunk's event is polled (for up to 20 ms, then the runtime's blocking wait): the blocking wait wakes up late -- tens of
        // microseconds as a rule, but the next chunk's launches are enqueued only after it, and on the 80 x 180 grid, whose exact
        // pass ends late in the following chunk's screening, the main stream ran dry for 100-120 us of every chunk (a tenth)
        auto wait_chunk = [&]() -> hipError_t {
            hipError_t q;
            const auto t0 = std::chrono::steady_clock::now();
            int spins = 0;
            while ((q = hipEventQuery(e->ev_chunk[c])) == hipErrorNotReady) {
                for (int p = 0; p < 32; ++p) cpu_relax();          // (a thread that appends meanwhile goes through the same runtime)
                if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return hipEventSynchronize(e->ev_chunk[c]);
            }
            return q;
        };
        if (last || owed.valid || pend.valid) {
            SCL_HIP(e, wait_chunk());
        } else {
            // more to submit behind this wait: the database lock is free meanwhile (appends get in; nothing of this call's state
            // names a slot or a pointer that a capacity doubling could move: the chunks in flight are enqueued, the next one is
            // built after the lock is back)
            db.unlock();
            const hipError_t w = wait_chunk();
            db.lock();
            SCL_HIP(e, w);
        }
        collect_profile(e);
        double listed = 0.0; int scored = 0;
        for (int i = 0; i < k.count; ++i) {
            const int o_i = k.first + i;
            nn_idx[o_i] = -1; shift[o_i] = 0; dist[o_i] = kBigDist;
            if (k.empty[(size_t)i]) continue;
            const volatile double *o = e->h_stream_out + ((size_t)c * NS + (size_t)i) * 8;
            listed += o[3]; ++scored;
            dist[o_i] = o[0];
            nn_idx[o_i] = o[1] < 0 ? -1 : k.lo[(size_t)i] + (int)o[1];
            shift[o_i] = (int)o[2];
        }
        if (small_exact && scored > 0) {                      // (with hysteresis: 32 survivors per scan and more, 16 and fewer)
            const double mean = listed / (double)scored;
            if (mean > 32.0) e->exact_heavy = true; else if (mean < 16.0) e->exact_heavy = false;
        }
        k.busy = false;
        return SCL_OK;
    };
    int next = 0, c = 0;
    while (next < n_queries || ch[0].busy || ch[1].busy) {
        if (next < n_queries && !ch[c].busy) {
            const int count = n_queries - next < chn ? n_queries - next : chn;
            const int nfirst = next + count;
            const int ncount = n_queries - nfirst < chn ? n_queries - nfirst : chn;
            if ((rc = submit(c, next, count, nfirst, ncount > 0 ? ncount : 0))) {
                const std::string first_error = e->last_error;
                (void)hipStreamSynchronize(e->stream); (void)hipStreamSynchronize(e->stream_surv);
                e->last_error = first_error;
                return rc;
            }
            next += count;
            c ^= 1;
            continue;
        }
        // both halves enqueued (or nothing left to submit): wait for the older one
        const int older = ch[c].busy ? c : c ^ 1;
        if ((rc = collect(older, next >= n_queries))) return rc;
    }
    return SCL_OK;
}

}  // namespace

namespace {

// scl_detect_full_stream behind its locks (pass_mu held, `lk` = the database lock): also the detection half of scl_stream_from_points
int detect_full_stream_locked(scl_engine *e, std::unique_lock<std::mutex> &lk, const int *queries, const int *lo, const int *hi, int n_queries,
                              int scans_per_launch, int launches_in_flight, int *nn_idx, int *shift, double *dist)
{
    for (int i = 0; i < scl_engine::kSlots; ++i)
        if (e->slot_busy[i]) return fail(e, SCL_ERR_INVALID_ARG, "detect_full_stream: collect the passes in flight first");
    if (n_queries >= 1 && n_queries <= kMaxQueryBatch && e->screen) {
        // a handful of scans (several robots' keyframes arriving together): the blocking form's launch group -- products that align for
        // themselves, one workgroup per scan for the exact pass, the winners read off the sequence number in pinned memory -- instead of
        // the stream's chunk machinery (two scans: 114 -> 80 us); the same winners, bit for bit
        int tk[kMaxQueryBatch];
        int rc = submit_full_many_locked(e, queries, lo, hi, n_queries, tk);
        if (rc) {
            const std::string first_error = e->last_error;
            (void)hipStreamSynchronize(e->stream);
            for (int i = 0; i < scl_engine::kSlots; ++i) e->slot_busy[i] = false;
            e->last_error = first_error;
            return rc;
        }
        for (int i = 0; i < n_queries; ++i) {
            const int r2 = collect_full_locked(e, tk[i], &nn_idx[i], &shift[i], &dist[i]);
            if (r2 && !rc) rc = r2;
        }
        return rc;
    }
    if (e->screen && (sc_distance_fuses_ring(db_view(e), e->SR) || sc_screen_is_wide(db_view(e), e->SR))) {
        // the screened grids take up to kMaxScreenBatch scans per launch: the workgroups of a launch's queries share every
        // keyframe line through the L2 (sc_screen.hip), so more scans per launch read less per scan from HBM
        const int mb = sc_screen_max_batch(db_view(e), e->SR);
        const int spl_s = scans_per_launch < 1 ? 1 : (scans_per_launch > mb ? mb : scans_per_launch);
        return stream_screened_locked(e, lk, queries, lo, hi, n_queries, spl_s, nn_idx, shift, dist);
    }
    const int spl = scans_per_launch < 1 ? 1 : (scans_per_launch > kMaxQueryBatch ? kMaxQueryBatch : scans_per_launch);
    int depth = launches_in_flight < 1 ? 1 : launches_in_flight;
    if (depth * spl > scl_engine::kSlots) depth = scl_engine::kSlots / spl;
    std::vector<int> tk((size_t)n_queries);
    int submitted = 0, collected = 0;
    // error path: the passes already enqueued are waited for and their slots released, so the engine's pipeline is
    // usable again afterwards (their results are dropped; last_error keeps the first failure)
    auto drain = [&]() {
        const std::string first_error = e->last_error;
        (void)hipStreamSynchronize(e->stream);
        for (int i = collected; i < submitted; ++i) e->slot_busy[tk[(size_t)i]] = false;
        e->last_error = first_error;
    };
    while (collected < n_queries) {
        while (submitted < n_queries) {                       // keep `depth` launches enqueued
            const int m = n_queries - submitted < spl ? n_queries - submitted : spl;
            if ((submitted - collected) + m > depth * spl) break;
            const int rc = submit_full_many_locked(e, queries + submitted, lo + submitted, hi + submitted, m, tk.data() + submitted);
            if (rc) { drain(); return rc; }
            submitted += m;
        }
        const int rc = collect_full_locked(e, tk[(size_t)collected], nn_idx + collected, shift + collected, dist + collected);
        ++collected;
        if (rc) { drain(); return rc; }
    }
    return SCL_OK;
}

}  // namespace

int scl_detect_full_stream(scl_engine *e, const int *queries, const int *lo, const int *hi, int n_queries,
                           int scans_per_launch, int launches_in_flight, int *nn_idx, int *shift, double *dist)
{
    if (!e || !queries || !lo || !hi || !nn_idx || !shift || !dist || n_queries < 0) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_detect_full_stream(e, queries, lo, hi, n_queries, scans_per_launch, launches_in_flight, nn_idx, shift, dist);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::unique_lock<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    return detect_full_stream_locked(e, lk, queries, lo, hi, n_queries, scans_per_launch, launches_in_flight, nn_idx, shift, dist);
}

namespace {

constexpr int kDefaultCopyStreams = 2;

// The per-incoming-scan pipeline from raw points (DM.h:988-1025 makeDescriptors once per keyframe, then DM.h:1078 detection):
// the scans go through in GROUPS of up to kMaxScBatch.  A group's clouds are copied into one of three device buffers on the copy
// stream -- by DMA when the caller's buffers are pinned (scl_host_alloc / scl_host_register) -- while the group before is binned (one
// scatter launch), ingested (one launch: every array of its database slots), and, `detect`, searched for over the whole database
// ([0, key - NUM_EXCLUDE_RECENT), D.h:1627) by the stream form's launch group.  PCIe is the floor of this path (n x stride bytes per
// scan against ~10 us of kernels), so the only thing that matters is that the copy engine never waits: group g + 1's copy is
// enqueued before group g's kernels, and the host blocks only on group g's results.
// pass_mu (detect) and `db` = e->mu are held by the caller; the database lock is given up while a group's detection waits.
int points_pipeline_locked(scl_engine *e, std::unique_lock<std::mutex> &db, const void *const *clouds, const int *n_points, int n_scans,
                           int stride_bytes, const int8_t *robots, const int *indexs, float *out_values,
                           bool detect, int *nn_idx, int *shift, double *dist)
{
    int rc;
    for (int i = 0; i < n_scans; ++i)
        if ((rc = check_cloud(e, clouds[i], n_points[i], stride_bytes))) return rc;
    if (n_scans == 0) return SCL_OK;
    constexpr int G = kMaxScBatch;
    const int ng = (n_scans + G - 1) / G;
    const size_t cells = (size_t)e->R * e->S;
    // every group's clouds at 256-byte aligned offsets of one buffer: the largest group sizes the three buffers
    // ... unless the group's clouds lie one behind the other in ONE host allocation (a ring of incoming scans, an arena the host
    // assembles its clouds in): ascending addresses, 16-byte aligned distances, gaps of at most a page.  Such a group travels as ONE copy
    // of its whole span -- at the link's rate for large copies (56.8 GB/s) instead of sixteen start-ups (44.8 GB/s on one stream) -- and
    // lies on the device as it lay on the host (*merged; off = the clouds' distances from the first).
    auto group_bytes = [&](int g, size_t *off, bool *merged_out = nullptr) {
        const int i0 = g * G, i1 = (g + 1) * G < n_scans ? (g + 1) * G : n_scans;
        bool merged = i1 - i0 >= 2;
        const unsigned char *base = static_cast<const unsigned char *>(clouds[i0]), *end = base;
        for (int i = i0; merged && i < i1; ++i) {
            const unsigned char *c = static_cast<const unsigned char *>(clouds[i]);
            const size_t bytes = (size_t)n_points[i] * (size_t)stride_bytes;
            if (!c || bytes == 0 || c < end || (size_t)(c - end) >= 4096 || ((size_t)(c - base) & 15)) merged = false;   // (a gap below a page lies in pages the clouds themselves touch)
            else end = c + bytes;
        }
        if (merged_out) *merged_out = merged;
        if (merged) {
            for (int i = i0; off && i < i1; ++i) off[i - i0] = (size_t)(static_cast<const unsigned char *>(clouds[i]) - base);
            return (size_t)(end - base) + 16;
        }
        size_t at = 0;
        for (int i = i0; i < i1; ++i) {
            if (off) off[i - i0] = at;
            at += (((size_t)n_points[i] * (size_t)stride_bytes + 16) + 255) & ~(size_t)255;
        }
        return at;
    };
    size_t need = 0;
    for (int g = 0; g < ng; ++g) { const size_t b = group_bytes(g, nullptr); need = b > need ? b : need; }
    const int nbuf = ng < 3 ? ng : 3;
    if (!e->stream_copy) SCL_HIP(e, hipStreamCreateWithFlags(&e->stream_copy, hipStreamNonBlocking));
    // several copy streams: a cloud is one copy of a megabyte or two, and the copies of ONE stream run strictly one after the other,
    // each with its own start-up (17 us: 44.8 GB/s for cloud-sized copies against 56.8 for one large copy); with the clouds of a group
    // dealt over several streams another copy is in flight while one is being set up
    const int ncs = std::min(std::max(scl_lab_int("SCL_COPY_STREAMS", kDefaultCopyStreams), 1), (int)scl_engine::kCopyStreams);
    auto cstream = [&](int c) { return c == 0 ? e->stream_copy : e->stream_copy_x[c - 1]; };
    for (int c = 1; c < ncs; ++c)
        if (!e->stream_copy_x[c - 1]) SCL_HIP(e, hipStreamCreateWithFlags(&e->stream_copy_x[c - 1], hipStreamNonBlocking));
    for (int b = 0; b < 3; ++b) {
        for (int c = 0; c < ncs; ++c)
            if (!e->ev_copied[c][b]) SCL_HIP(e, hipEventCreateWithFlags(&e->ev_copied[c][b], hipEventDisableTiming));
        if (!e->ev_consumed[b]) SCL_HIP(e, hipEventCreateWithFlags(&e->ev_consumed[b], hipEventDisableTiming));
    }
    if (need > e->pbuf_cap || !e->d_pbuf[nbuf - 1]) {      // (nothing of an earlier call is in flight: every call ends with its last group consumed)
        const size_t nb = need > e->pbuf_cap ? need + need / 4 + 4096 : e->pbuf_cap;
        for (int b = 0; b < 3; ++b) {
            if (need > e->pbuf_cap) { dev_free(e->d_pbuf[b]); e->d_pbuf[b] = nullptr; }
            if (b < nbuf && !e->d_pbuf[b] && (rc = dev_alloc(e, &e->d_pbuf[b], nb))) { e->pbuf_cap = 0; return rc; }
        }
        e->pbuf_cap = nb;
    }
    if ((rc = ensure_capacity(e, e->n + n_scans))) return rc;     // the slots of every scan of the call: no regrow between groups

    auto enqueue_copy = [&](int g) -> int {
        const int b = g % 3;
        size_t off[G];
        bool merged = false;
        const size_t span = group_bytes(g, off, &merged);
        if (g >= 3)                                          // the group that was binned out of this buffer
            for (int c = 0; c < ncs; ++c) SCL_HIP(e, hipStreamWaitEvent(cstream(c), e->ev_consumed[b], 0));
        if (merged) SCL_HIP(e, hipMemcpyAsync(e->d_pbuf[b], clouds[g * G], span - 16, hipMemcpyHostToDevice, cstream(0)));
        else for (int i = g * G; i < n_scans && i < (g + 1) * G; ++i) {
            const size_t bytes = (size_t)n_points[i] * (size_t)stride_bytes;
            if (bytes) SCL_HIP(e, hipMemcpyAsync(e->d_pbuf[b] + off[i - g * G], clouds[i], bytes, hipMemcpyHostToDevice, cstream(i % ncs)));
        }
        for (int c = 0; c < ncs; ++c) SCL_HIP(e, hipEventRecord(e->ev_copied[c][b], cstream(c)));
        return SCL_OK;
    };
    // error path: nothing may still read the caller's buffers or write the point buffers when the call returns
    auto bail = [&](int code) {
        const std::string first = e->last_error;
        for (int c = 0; c < ncs; ++c) (void)hipStreamSynchronize(cstream(c));
        (void)hipStreamSynchronize(e->stream);
        e->last_error = first;
        return code;
    };

    if ((rc = enqueue_copy(0))) return bail(rc);
    const int excl = e->cfg.num_exclude_recent;
    for (int g = 0; g < ng; ++g) {
        if (g + 1 < ng && (rc = enqueue_copy(g + 1))) return bail(rc);
        const int b = g % 3, i0 = g * G, m = n_scans - i0 < G ? n_scans - i0 : G;
        size_t off[G];
        group_bytes(g, off);
        const unsigned char *dptr[G];
        for (int j = 0; j < m; ++j) dptr[j] = e->d_pbuf[b] + off[j];
        for (int c = 0; c < ncs; ++c)
            if (hipStreamWaitEvent(e->stream, e->ev_copied[c][b], 0) != hipSuccess) return bail(fail(e, SCL_ERR_HIP, "hipStreamWaitEvent(copied)"));
        if ((rc = group_scatter(e, dptr, n_points + i0, m, stride_bytes, e->stream))) return bail(rc);
        if (hipEventRecord(e->ev_consumed[b], e->stream) != hipSuccess) return bail(fail(e, SCL_ERR_HIP, "hipEventRecord(consumed)"));
        if ((rc = ensure_capacity(e, e->n + m))) return bail(rc);   // (no-op unless another thread appended while a group's detection waited)
        const int first_slot = e->n;
        if ((rc = group_ingest(e, m, first_slot, e->stream))) return bail(rc);
        if (out_values && hipMemcpyAsync(out_values + (size_t)i0 * cells, e->d_vals, sizeof(float) * cells * (size_t)m, hipMemcpyDeviceToHost, e->stream) != hipSuccess)
            return bail(fail(e, SCL_ERR_HIP, "descriptor values to the host"));
        for (int j = 0; j < m; ++j) {
            e->robots.push_back(robots ? robots[i0 + j] : (int8_t)0);
            e->indexs.push_back(indexs ? indexs[i0 + j] : first_slot + j);
        }
        e->n += m;                                          // (stream order: whatever scores these slots is enqueued behind their ingest)
        if (detect) {
            int q[G], lo[G], hi[G];
            for (int j = 0; j < m; ++j) { q[j] = first_slot + j; lo[j] = 0; hi[j] = first_slot + j - excl; if (hi[j] < 0) hi[j] = 0; }
            if ((rc = detect_full_stream_locked(e, db, q, lo, hi, m, G, 2, nn_idx + i0, shift + i0, dist + i0))) return bail(rc);
            if (!db.owns_lock()) db.lock();
        } else if (g + 1 == ng || out_values) {
            if ((rc = sync(e))) return bail(rc);            // (d_vals is rewritten by the next group)
        }
    }
    if ((rc = sync(e))) return bail(rc);
    return SCL_OK;
}

}  // namespace

namespace {

// The same path for clouds that are ALREADY on the device (the keyframe store): with no copy to hide, the groups' two launches go
// back to back on the stream (16 scans each), and the detection of all the new keyframes is ONE call of the stream form -- whose
// launch groups overlap (a group's finishing rides beside the next group's alignment), which separate calls per group of 16 cannot.
int device_points_pipeline_locked(scl_engine *e, std::unique_lock<std::mutex> &db, const unsigned char *const *d_clouds, const int *n_points,
                                  int n_scans, int stride_bytes, const int8_t *robots, const int *indexs, float *out_values,
                                  int *nn_idx, int *shift, double *dist)
{
    int rc;
    if (n_scans == 0) return SCL_OK;
    constexpr int G = kMaxScBatch;
    const size_t cells = (size_t)e->R * e->S;
    if ((rc = ensure_capacity(e, e->n + n_scans))) return rc;
    const int first_key = e->n;
    if ((rc = ensure_tiles(e, e->stream))) return rc;
    if ((rc = ensure_vals(e, cells * (size_t)(2 * G)))) return rc;
    for (int i = 0; i < n_scans; ++i)
        if (n_points[i] < 0 || (n_points[i] > 0 && !d_clouds[i])) return fail(e, SCL_ERR_INVALID_ARG, "stored cloud without points");
    // launch g = [ingest of group g - 1 out of tile set (g - 1) & 1] + [scatter of group g into tile set g & 1]: ng + 1 launches back to back
    const int ng = (n_scans + G - 1) / G;
    const int slice = scl_lab_int("SCL_SC_SLICE", e->R * e->S > 10000 ? 2 * kScPointsPerWorkgroup : kScPointsPerWorkgroup);
    e->tiles_clean = false;
    for (int g = 0; g <= ng; ++g) {
        ScanBatch b{};
        if (g < ng) {
            const int i0 = g * G;
            b.count = n_scans - i0 < G ? n_scans - i0 : G;
            for (int j = 0; j < b.count; ++j) { b.points[j] = d_clouds[i0 + j]; b.n[j] = n_points[i0 + j]; }
        }
        IngestArgs ia{};
        int n_ingest = 0;
        ia.R = e->R; ia.S = e->S;
        if (g > 0) {
            const int i0 = (g - 1) * G;
            n_ingest = n_scans - i0 < G ? n_scans - i0 : G;
            ia = IngestArgs{nullptr, first_key + i0, e->d_desc, e->d_vkey, e->d_norm, e->d_rkey, e->d_rkey4, e->d_hdesc, e->d_kmask, e->d_hkey, e->hstride, e->cap,
                            e->R, e->S, e->d_halign, e->d_tiles + (size_t)((g - 1) & 1) * G * cells, e->d_vals + (size_t)((g - 1) & 1) * G * cells};
        }
        {
            ProfScope ps(e, P_MAKESC);
            SCL_HIP(e, launch_front_fused(b, stride_bytes, e->cfg.lidar_height, e->cfg.max_radius, e->d_tiles + (size_t)(g & 1) * G * cells, slice, ia, n_ingest, e->stream));
        }
        if (g > 0 && out_values) {
            const int i0 = (g - 1) * G;
            SCL_HIP(e, hipMemcpyAsync(out_values + (size_t)i0 * cells, e->d_vals + (size_t)((g - 1) & 1) * G * cells, sizeof(float) * cells * (size_t)n_ingest,
                                      hipMemcpyDeviceToHost, e->stream));
        }
    }
    e->tiles_clean = true;
    e->db_version++;
    for (int i = 0; i < n_scans; ++i) {
        e->robots.push_back(robots ? robots[i] : (int8_t)0);
        e->indexs.push_back(indexs ? indexs[i] : first_key + i);
    }
    e->prof.make_sc_points += e->prof_on ? [&] { uint64_t t = 0; for (int i = 0; i < n_scans; ++i) t += (uint64_t)n_points[i]; return t; }() : 0;
    e->n += n_scans;
    std::vector<int> q((size_t)n_scans), lo((size_t)n_scans, 0), hi((size_t)n_scans);
    const int excl = e->cfg.num_exclude_recent;
    for (int i = 0; i < n_scans; ++i) { q[(size_t)i] = first_key + i; hi[(size_t)i] = first_key + i - excl < 0 ? 0 : first_key + i - excl; }
    if ((rc = detect_full_stream_locked(e, db, q.data(), lo.data(), hi.data(), n_scans, G, 2, nn_idx, shift, dist))) return rc;
    if (!db.owns_lock()) db.lock();
    return sync(e);
}

}  // namespace

/* makeDescriptors + full-database detection for keyframes whose clouds are in the on-device keyframe store (scl_keyframe_put) */
int scl_stream_from_store(scl_engine *e, int robot, int first_index, int count, int *nn_idx, int *shift, double *dist, float *out_values)
{
    if (!e || robot < 0 || first_index < 0 || count < 0 || (count > 0 && (!nn_idx || !shift || !dist))) return SCL_ERR_INVALID_ARG;
    if (e->front) return fail(e, SCL_ERR_UNSUPPORTED, "stream_from_store: the keyframe store lives on one engine; call it on a one-GPU engine");
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::unique_lock<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    if ((size_t)robot >= e->kf.size() || (size_t)first_index + (size_t)count > e->kf[(size_t)robot].size())
        return fail(e, SCL_ERR_OUT_OF_RANGE, "stream_from_store: keyframes not in the store");
    std::vector<const unsigned char *> d((size_t)count);
    std::vector<int> n((size_t)count), idx((size_t)count);
    std::vector<int8_t> rb((size_t)count, (int8_t)robot);
    for (int i = 0; i < count; ++i) {
        const auto &c = e->kf[(size_t)robot][(size_t)(first_index + i)];
        if (c.n < 0) return fail(e, SCL_ERR_OUT_OF_RANGE, "stream_from_store: a keyframe of the range was never stored");
        d[(size_t)i] = c.d; n[(size_t)i] = c.n; idx[(size_t)i] = first_index + i;
    }
    return device_points_pipeline_locked(e, lk, d.data(), n.data(), count, e->kf_stride, rb.data(), idx.data(), out_values, nn_idx, shift, dist);
}

/* ---- pinned host buffers ----------------------------------------------------------- */

int scl_host_alloc(scl_engine *e, size_t bytes, void **out)
{
    if (!e || !out || bytes == 0) return SCL_ERR_INVALID_ARG;
    scl_engine *t = e->front ? front_primary(e) : e;
    std::lock_guard<std::mutex> lk(t->mu);
    (void)hipSetDevice(t->device);
    void *p = nullptr;
    SCL_HIP(e, hipHostMalloc(&p, bytes, hipHostMallocPortable));
    t->host_allocs.push_back(p);
    *out = p;
    return SCL_OK;
}

int scl_host_free(scl_engine *e, void *p)
{
    if (!e) return SCL_ERR_INVALID_ARG;
    if (!p) return SCL_OK;
    scl_engine *t = e->front ? front_primary(e) : e;
    std::lock_guard<std::mutex> lk(t->mu);
    for (size_t i = 0; i < t->host_allocs.size(); ++i)
        if (t->host_allocs[i] == p) {
            t->host_allocs.erase(t->host_allocs.begin() + (long)i);
            (void)hipSetDevice(t->device);
            SCL_HIP(e, hipHostFree(p));
            return SCL_OK;
        }
    return fail(e, SCL_ERR_INVALID_ARG, "scl_host_free: not a buffer of scl_host_alloc on this engine");
}

int scl_host_register(scl_engine *e, void *p, size_t bytes)
{
    if (!e || !p || bytes == 0) return SCL_ERR_INVALID_ARG;
    scl_engine *t = e->front ? front_primary(e) : e;
    (void)hipSetDevice(t->device);
    SCL_HIP(e, hipHostRegister(p, bytes, hipHostRegisterPortable));
    return SCL_OK;
}

int scl_host_unregister(scl_engine *e, void *p)
{
    if (!e || !p) return SCL_ERR_INVALID_ARG;
    scl_engine *t = e->front ? front_primary(e) : e;
    (void)hipSetDevice(t->device);
    SCL_HIP(e, hipHostUnregister(p));
    return SCL_OK;
}

/* ---- batches of scans from raw points ------------------------------------------------ */

int scl_make_and_save_many(scl_engine *e, const void *const *clouds, const int *n_points, int count, int stride_bytes,
                           const int8_t *robots, const int *indexs, float *out_values)
{
    if (!e || count < 0 || (count > 0 && (!clouds || !n_points))) return SCL_ERR_INVALID_ARG;
    if (e->front) {                                        // sharded: keyframe by keyframe through the owners (correct, not tuned)
        const size_t cells = (size_t)e->R * e->S;
        for (int i = 0; i < count; ++i) {
            const int rc = scl_make_and_save(e, clouds[i], n_points[i], stride_bytes, robots ? robots[i] : (int8_t)0,
                                             indexs ? indexs[i] : scl_get_size(e, -1), out_values ? out_values + (size_t)i * cells : nullptr);
            if (rc) return rc;
        }
        return SCL_OK;
    }
    std::unique_lock<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    return points_pipeline_locked(e, lk, clouds, n_points, count, stride_bytes, robots, indexs, out_values, false, nullptr, nullptr, nullptr);
}

int scl_stream_from_points(scl_engine *e, const void *const *clouds, const int *n_points, int n_scans, int stride_bytes,
                           const int8_t *robots, const int *indexs, int *nn_idx, int *shift, double *dist, float *out_values)
{
    if (!e || n_scans < 0 || (n_scans > 0 && (!clouds || !n_points || !nn_idx || !shift || !dist))) return SCL_ERR_INVALID_ARG;
    if (e->front) {
        const size_t cells = (size_t)e->R * e->S;
        const int excl = e->cfg.num_exclude_recent;
        for (int i0 = 0; i0 < n_scans; i0 += kMaxScBatch) {
            const int m = n_scans - i0 < kMaxScBatch ? n_scans - i0 : kMaxScBatch;
            int q[kMaxScBatch], lo[kMaxScBatch], hi[kMaxScBatch];
            for (int j = 0; j < m; ++j) {
                const int key = scl_get_size(e, -1);
                if (key < 0) return key;
                const int rc = scl_make_and_save(e, clouds[i0 + j], n_points[i0 + j], stride_bytes, robots ? robots[i0 + j] : (int8_t)0,
                                                 indexs ? indexs[i0 + j] : key, out_values ? out_values + (size_t)(i0 + j) * cells : nullptr);
                if (rc) return rc;
                q[j] = key; lo[j] = 0; hi[j] = key - excl < 0 ? 0 : key - excl;
            }
            const int rc = scl_detect_full_stream(e, q, lo, hi, m, kMaxScBatch, 2, nn_idx + i0, shift + i0, dist + i0);
            if (rc) return rc;
        }
        return SCL_OK;
    }
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::unique_lock<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    return points_pipeline_locked(e, lk, clouds, n_points, n_scans, stride_bytes, robots, indexs, out_values, true, nn_idx, shift, dist);
}

/* measurement: what ONE large copy from pinned host memory reaches on this box's link (the floor of scl_stream_from_points) */
int scl_host_copy_rate(scl_engine *e, size_t bytes, int reps, double *gbytes_per_s)
{
    if (!e || !gbytes_per_s || bytes == 0 || reps < 1) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    void *h = nullptr; unsigned char *d = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t he = hipHostMalloc(&h, bytes, hipHostMallocDefault);
    if (he == hipSuccess) { memset(h, 1, bytes); he = hipMalloc((void **)&d, bytes); }
    if (he == hipSuccess) he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    if (!e->stream_copy && he == hipSuccess) he = hipStreamCreateWithFlags(&e->stream_copy, hipStreamNonBlocking);
    float ms = 0.f;
    if (he == hipSuccess) he = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, e->stream_copy);       // warm
    if (he == hipSuccess) he = hipEventRecord(e0, e->stream_copy);
    for (int i = 0; i < reps && he == hipSuccess; ++i) he = hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, e->stream_copy);
    if (he == hipSuccess) he = hipEventRecord(e1, e->stream_copy);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d) (void)hipFree(d);
    if (h) (void)hipHostFree(h);
    SCL_HIP(e, he);
    *gbytes_per_s = ms > 0.f ? (double)bytes * reps / (ms * 1e-3) / 1e9 : 0.0;
    return SCL_OK;
}

/* test hook: the descriptor scatter's fast binning (device_common.hpp: sc_bin_fast) against the reference's chain (sc_bin_exact,
 * D.h:1425-1435) on n generated points of the engine's grid */
int scl_selftest_bin_paths(scl_engine *e, int mode, uint64_t seed, uint64_t n_points, uint64_t *disagreements, uint64_t *sure)
{
    if (!e || !disagreements || !sure || mode < 0 || mode > 3) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    unsigned long long *d = nullptr, h[2] = {0, 0};
    int rc = dev_alloc(e, &d, (size_t)2);
    if (rc) return rc;
    hipError_t he = launch_bin_paths_selftest(mode, seed, n_points, e->R, e->S, e->cfg.max_radius, d, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    dev_free(d);
    SCL_HIP(e, he);
    *disagreements = h[0]; *sure = h[1];
    return SCL_OK;
}

/* test hook: csrc/device_sort.hip's stable radix sort on the key bits [0, bits) -- key_bytes 4 or 8; n_segments > 1: every segment
 * [segment_offsets[s], segment_offsets[s + 1]) on its own -- as the verification path uses it (voxel.hip, icp.hip) */
int scl_selftest_sort_pairs(scl_engine *e, int key_bytes, const void *keys, const uint32_t *values, int n, int bits, const int *segment_offsets,
                            int n_segments, void *keys_out, uint32_t *values_out)
{
    if (!e || n < 0 || (key_bytes != 4 && key_bytes != 8) || (n > 0 && (!keys || !values || !keys_out || !values_out))) return SCL_ERR_INVALID_ARG;
    if (n_segments < 0 || n_segments > kSortMaxSegments || (n_segments > 1 && (!segment_offsets || key_bytes != 8))) return SCL_ERR_INVALID_ARG;
    if (n == 0) return SCL_OK;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    SortSegments seg{};
    seg.nseg = n_segments > 1 ? n_segments : 1; seg.off[0] = 0; seg.off[1] = n;
    if (n_segments > 1) {
        for (int s = 0; s <= n_segments; ++s) seg.off[s] = segment_offsets[s];
        if (seg.off[0] != 0 || seg.off[n_segments] != n) return SCL_ERR_INVALID_ARG;
    }
    unsigned char *d = nullptr;
    const size_t kb = (size_t)key_bytes * (size_t)n, vb = sizeof(uint32_t) * (size_t)n, kpad = (kb + 255) & ~(size_t)255, vpad = (vb + 255) & ~(size_t)255;
    const size_t sb = sort_scratch_bytes((size_t)n, seg.nseg);
    int rc = dev_alloc(e, &d, 2 * kpad + 2 * vpad + sb);
    if (rc) return rc;
    unsigned char *k0 = d, *k1 = d + kpad, *v0 = d + 2 * kpad, *v1 = v0 + vpad, *scratch = v1 + vpad;
    hipError_t he = hipMemcpyAsync(k0, keys, kb, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(v0, values, vb, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) {
        if (key_bytes == 4) he = sort_pairs_u32(scratch, (unsigned int *)k0, (unsigned int *)k1, (unsigned int *)v0, (unsigned int *)v1, n, bits, e->stream);
        else {
            // keys whose high word is the same throughout every segment take the 8-byte records of voxel.hip's batch (device_sort.hpp)
            unsigned int seg_hi[kSortMaxSegments];
            bool uniform = bits <= 32;
            const unsigned long long *hk = static_cast<const unsigned long long *>(keys);
            for (int s = 0; s < seg.nseg && uniform; ++s) {
                seg_hi[s] = seg.off[s] < seg.off[s + 1] ? (unsigned int)(hk[seg.off[s]] >> 32) : 0u;
                for (int i = seg.off[s]; i < seg.off[s + 1]; ++i) if ((unsigned int)(hk[i] >> 32) != seg_hi[s]) { uniform = false; break; }
            }
            he = sort_pairs_u64_segmented(scratch, (unsigned long long *)k0, (unsigned long long *)k1, (unsigned int *)v0, (unsigned int *)v1, seg, bits, e->stream,
                                          uniform ? seg_hi : nullptr);
        }
    }
    if (he == hipSuccess) he = hipMemcpyAsync(keys_out, k1, kb, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(values_out, v1, vb, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    dev_free(d);
    SCL_HIP(e, he);
    return SCL_OK;
}

/* test hook: csrc/device_sort.hip's prefix sum of n ints (in place on the device, as the cell table's is) */
int scl_selftest_prefix_sum(scl_engine *e, const int32_t *in, int n, int inclusive, int32_t *out)
{
    if (!e || n < 0 || (n > 0 && (!in || !out))) return SCL_ERR_INVALID_ARG;
    if (n == 0) return SCL_OK;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    unsigned char *d = nullptr;
    const size_t nb = (sizeof(int32_t) * (size_t)n + 255) & ~(size_t)255;
    int rc = dev_alloc(e, &d, nb + scan_scratch_bytes((size_t)n));
    if (rc) return rc;
    hipError_t he = hipMemcpyAsync(d, in, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, e->stream);
    if (he == hipSuccess) he = prefix_sum_i32(d + nb, (const int *)d, (int *)d, n, inclusive != 0, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(out, d, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    dev_free(d);
    SCL_HIP(e, he);
    return SCL_OK;
}

/* test hook: the device's atanf (xy2theta, D.h:1352-1374) over blocks of 2^24 float bit patterns, as the checksums of
 * tests/golden/atanf_blocks.json */
int scl_selftest_atanf_blocks(scl_engine *e, int first_block, int n_blocks, uint64_t *checksums)
{
    if (!e || !checksums || first_block < 0 || n_blocks < 1 || first_block + n_blocks > 256) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    unsigned long long *d = nullptr;
    int rc = dev_alloc(e, &d, (size_t)n_blocks);
    if (rc) return rc;
    hipError_t he = launch_atanf_block_checksums(first_block, n_blocks, d, e->stream);
    if (he == hipSuccess) he = hipMemcpyAsync(checksums, d, sizeof(uint64_t) * (size_t)n_blocks, hipMemcpyDeviceToHost, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    dev_free(d);
    SCL_HIP(e, he);
    return SCL_OK;
}

int scl_detect_full_collect(scl_engine *e, int ticket, int *nn_idx, int *shift, double *dist)
{
    if (!e || !nn_idx || !shift || !dist) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_collect(e, ticket, nn_idx, shift, dist);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    return collect_full_locked(e, ticket, nn_idx, shift, dist);
}

int scl_detect_full_range(scl_engine *e, int query, int lo, int hi, int *nn_idx, int *shift, double *dist)
{
    if (!e || !nn_idx || !shift || !dist) return SCL_ERR_INVALID_ARG;
    if (e->front) {
        int t = -1;
        const int rc = front_submit_many(e, &query, &lo, &hi, 1, &t);
        return rc ? rc : front_collect(e, t, nn_idx, shift, dist);
    }
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    *nn_idx = -1; *shift = 0; *dist = kBigDist;
    int ticket = -1;
    int rc = submit_full_locked(e, query, lo, hi, &ticket);
    if (rc) return rc;
    return collect_full_locked(e, ticket, nn_idx, shift, dist);
}

int scl_screen_distances(scl_engine *e, int query, int lo, int hi, float *approx, int *survivors, int *n_survivors, float *eps)
{
    if (!e || !approx) return SCL_ERR_INVALID_ARG;
    if (e->front) return fail(e, SCL_ERR_UNSUPPORTED, "screen_distances: call it on a one-GPU engine");
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    if (eps) *eps = sc_screen_eps();
    if (!e->screen) return fail(e, SCL_ERR_UNSUPPORTED, "no screening pass for this grid (64x120 and 80x180 at search ratio 0.1 only)");
    int qslot;
    if (query >= 0) { if (query >= e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "query slot out of range"); qslot = query; }
    else { const int j = -1 - query; if (j >= e->stage_rows || !e->staged[j]) return fail(e, SCL_ERR_INVALID_ARG, "no staged query"); qslot = e->cap + j; }
    if (lo < 0) lo = 0;
    if (hi > e->n) hi = e->n;
    const int n = hi - lo;
    if (n_survivors) *n_survivors = 0;
    if (n <= 0) return SCL_OK;
    int rc;
    if ((rc = ensure_pairs(e, (size_t)n))) return rc;
    {   // scratch of the products' second form: ring parts x passes x 16 floats per pair
        const size_t per_pair = sc_screen_scratch_floats(db_view(e), e->SR) / (size_t)sc_screen_max_batch(db_view(e), e->SR);
        if ((size_t)n * per_pair > e->part_cap) {
            dev_free(e->d_part); e->part_cap = 0;
            if ((rc = dev_alloc(e, &e->d_part, (size_t)n * per_pair + 1024))) return rc;
            e->part_cap = (size_t)n * per_pair + 1024;
        }
    }
    ScreenBatch sb{};
    sb.part = e->d_part;
    sb.nq = 1; sb.slot[0] = qslot; sb.base[0] = lo; sb.n[0] = n; sb.pair_stride = (size_t)n;
    sb.approx = e->d_approx; sb.starts = e->d_starts; sb.align_fallbacks = e->d_align_fallbacks; sb.ring_d2 = e->d_ring_d2; sb.survivors = e->d_surv; sb.n_surv = e->d_nsurv; sb.t_min = e->d_tmin;
    sb.k = e->cfg.num_candidates; sb.exclude_eps = e->cfg.knn_exclude_eps; sb.topk_idx = e->d_topk_idx; sb.topk_d2 = e->d_topk_d2;
    SCL_HIP(e, launch_sc_screen_batch(db_view(e), sb, e->SR, sc_align_filter_enabled(), e->num_cu, e->stream));
    SCL_HIP(e, launch_sc_select_batch(sb, e->stream));
    SCL_HIP(e, hipMemcpyAsync(approx, e->d_approx, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
    int ns = 0;
    SCL_HIP(e, hipMemcpyAsync(&ns, e->d_nsurv, sizeof(int), hipMemcpyDeviceToHost, e->stream));
    if ((rc = sync(e))) return rc;
    if (n_survivors) *n_survivors = ns;
    if (survivors && ns > 0) {
        SCL_HIP(e, hipMemcpyAsync(survivors, e->d_surv, sizeof(int) * (size_t)ns, hipMemcpyDeviceToHost, e->stream));
        if ((rc = sync(e))) return rc;
    }
    return SCL_OK;
}

int scl_screen_distances_many(scl_engine *e, const int *queries, int n_queries, int lo, int hi, float *approx, float *eps)
{
    if (!e || !queries || !approx || n_queries < 1 || n_queries > kMaxScreenBatch) return SCL_ERR_INVALID_ARG;
    if (e->front) return fail(e, SCL_ERR_UNSUPPORTED, "screen_distances_many: call it on a one-GPU engine");
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    if (eps) *eps = sc_screen_eps();
    if (!e->screen) return fail(e, SCL_ERR_UNSUPPORTED, "no screening pass for this grid (64x120 and 80x180 at search ratio 0.1 only)");
    if (n_queries > sc_screen_max_batch(db_view(e), e->SR)) return fail(e, SCL_ERR_INVALID_ARG, "more scans than a screening launch takes");
    if (lo < 0) lo = 0;
    if (hi > e->n) hi = e->n;
    const int n = hi - lo;
    if (n <= 0) return SCL_OK;
    int rc;
    if ((rc = ensure_pairs(e, (size_t)n * (size_t)n_queries))) return rc;
    {
        const size_t per_pair = sc_screen_scratch_floats(db_view(e), e->SR) / (size_t)sc_screen_max_batch(db_view(e), e->SR);
        const size_t want = (size_t)n * (size_t)n_queries * per_pair + 1024;
        if (want > e->part_cap) {
            dev_free(e->d_part); e->part_cap = 0;
            if ((rc = dev_alloc(e, &e->d_part, want))) return rc;
            e->part_cap = want;
        }
    }
    ScreenBatch sb{};
    sb.part = e->d_part;
    sb.nq = n_queries; sb.pair_stride = (size_t)n;
    for (int i = 0; i < n_queries; ++i) {
        const int q = queries[i];
        if (q >= 0) { if (q >= e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "query slot out of range"); sb.slot[i] = q; }
        else { const int j = -1 - q; if (j >= e->stage_rows || !e->staged[j]) return fail(e, SCL_ERR_INVALID_ARG, "no staged query"); sb.slot[i] = e->cap + j; }
        sb.base[i] = lo; sb.n[i] = n; sb.buf[i] = i;
    }
    sb.approx = e->d_approx; sb.starts = e->d_starts; sb.align_fallbacks = e->d_align_fallbacks; sb.ring_d2 = e->d_ring_d2; sb.survivors = e->d_surv; sb.n_surv = e->d_nsurv; sb.t_min = e->d_tmin;
    sb.k = e->cfg.num_candidates; sb.exclude_eps = e->cfg.knn_exclude_eps; sb.topk_idx = e->d_topk_idx; sb.topk_d2 = e->d_topk_d2;
    SCL_HIP(e, launch_sc_screen_batch(db_view(e), sb, e->SR, sc_align_filter_enabled(), e->num_cu, e->stream));
    SCL_HIP(e, hipMemsetAsync(e->d_tmin, 0xff, sizeof(unsigned int) * (size_t)n_queries, e->stream));          // (no select launch re-arms the words)
    SCL_HIP(e, hipMemsetAsync(e->d_tmin + kTminEpsOffset, 0, sizeof(unsigned int) * (size_t)n_queries, e->stream));
    SCL_HIP(e, hipMemcpyAsync(approx, e->d_approx, sizeof(float) * (size_t)n * (size_t)n_queries, hipMemcpyDeviceToHost, e->stream));
    return sync(e);
}

int scl_get_last_topk(scl_engine *e, int k, int *idx, float *d2)
{
    if (!e || !idx || !d2 || k < 1 || k > kTopkMaxK) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_get_last_topk(e, k, idx, d2);
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    if (e->last_pass_empty) {                              // the last pass had an empty range: no neighbours
        for (int i = 0; i < k; ++i) { idx[i] = -1; d2[i] = FLT_MAX; }
        return SCL_OK;
    }
    if (e->last_pass_alt) {                                // the most recent pass ran on the alt lane
        SCL_HIP(e, hipMemcpyAsync(idx, e->a_topk_idx, sizeof(int) * k, hipMemcpyDeviceToHost, e->stream_alt));
        SCL_HIP(e, hipMemcpyAsync(d2, e->a_topk_d2, sizeof(float) * k, hipMemcpyDeviceToHost, e->stream_alt));
        SCL_HIP(e, hipStreamSynchronize(e->stream_alt));
        return SCL_OK;
    }
    SCL_HIP(e, hipMemcpyAsync(idx, e->d_topk_idx, sizeof(int) * k, hipMemcpyDeviceToHost, e->stream));
    SCL_HIP(e, hipMemcpyAsync(d2, e->d_topk_d2, sizeof(float) * k, hipMemcpyDeviceToHost, e->stream));
    return sync(e);
}

int scl_detect_full(scl_engine *e, int cur, int *loop_id, int *nn_idx, int *shift, double *dist)
{
    if (!e || !loop_id || !nn_idx || !shift || !dist) return SCL_ERR_INVALID_ARG;
    *loop_id = -1;
    if (cur < 0) return SCL_ERR_OUT_OF_RANGE;
    const int hi = cur - e->cfg.num_exclude_recent;                       /* D.h:1627 */
    int rc = scl_detect_full_range(e, cur, 0, hi, nn_idx, shift, dist);
    if (rc) return rc;
    if (*nn_idx >= 0 && *dist < e->cfg.dist_thres) *loop_id = *nn_idx;
    return SCL_OK;
}

/* ---- database dump / load, index map ----------------------------------------------- */

int scl_get_descriptors(const scl_engine *ce, int first, int count, float *values)
{
    scl_engine *e = const_cast<scl_engine *>(ce);
    if (!e || !values || first < 0 || count < 0) return SCL_ERR_INVALID_ARG;
    const size_t cells = (size_t)e->R * e->S;
    if (e->front) {                                         // sharded: keyframe by keyframe through the owners
        for (int i = 0; i < count; ++i) { const int rc = scl_get_descriptor(e, first + i, values + (size_t)i * cells); if (rc) return rc; }
        return SCL_OK;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    if (first + count > e->n) return fail(e, SCL_ERR_OUT_OF_RANGE, "get_descriptors: range exceeds the database");
    const int chunk = 512;
    int rc;
    if ((rc = ensure_vals(e, cells * (size_t)(count < chunk ? (count > 0 ? count : 1) : chunk)))) return rc;
    for (int done = 0; done < count; done += chunk) {
        const int c = count - done < chunk ? count - done : chunk;
        for (int i = 0; i < c; ++i)
            SCL_HIP(e, launch_untile(e->d_desc + (size_t)(first + done + i) * e->RG * e->S, e->R, e->S, e->d_vals + (size_t)i * cells, e->stream));
        SCL_HIP(e, hipMemcpyAsync(values + (size_t)done * cells, e->d_vals, sizeof(float) * cells * (size_t)c, hipMemcpyDeviceToHost, e->stream));
        if ((rc = sync(e))) return rc;
    }
    return SCL_OK;
}

int scl_find_key(const scl_engine *e, int8_t robot, int index, int *key)
{
    if (!e || !key) return SCL_ERR_INVALID_ARG;
    *key = -1;
    const int n = scl_get_size(e, -1);
    if (n < 0) return n;
    // newest first: the callers of DM.h:1281-1284 ask about recent keyframes
    for (int k = n - 1; k >= 0; --k) {
        int8_t r = 0; int idx = 0;
        const int rc = scl_get_index(e, k, &r, &idx);
        if (rc) return rc;
        if (r == robot && idx == index) { *key = k; break; }
    }
    return SCL_OK;
}

int scl_db_dump_file(scl_engine *e, const char *path)
{
    if (!e || !path) return SCL_ERR_INVALID_ARG;
    const int n = scl_get_size(e, -1);
    if (n < 0) return n;
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(e, SCL_ERR_INVALID_ARG, "db_dump: cannot open the file for writing");
    DbFileHeader h{};
    std::memcpy(h.magic, kDbMagic, 8);
    h.version = 1; h.num_ring = e->R; h.num_sector = e->S; h.count = n;
    // state of detectInterLoopClosureID's periodic tree (D.h:1691-1703): reserved[0] = 1 marks the two words as present
    if (!e->front) { std::lock_guard<std::mutex> lk(e->mu); h.reserved[0] = 1; h.reserved[1] = e->tree_counter; h.reserved[2] = e->tree_n; }
    int rc = SCL_OK;
    if (std::fwrite(&h, sizeof h, 1, f) != 1) rc = SCL_ERR_INVALID_ARG;
    const size_t cells = (size_t)e->R * e->S;
    const int chunk = 512;
    std::vector<float> buf(cells * (size_t)chunk);
    for (int done = 0; done < n && !rc; done += chunk) {    // float32[N][R*S], wire order (D.h:1446-1455)
        const int c = n - done < chunk ? n - done : chunk;
        rc = scl_get_descriptors(e, done, c, buf.data());
        if (!rc && std::fwrite(buf.data(), sizeof(float) * cells, (size_t)c, f) != (size_t)c) rc = fail(e, SCL_ERR_INVALID_ARG, "db_dump: short write");
    }
    for (int k = 0; k < n && !rc; ++k) {                    // (robot, index) map, D.h:1758-1761
        int8_t r = 0; int32_t idx = 0;
        rc = scl_get_index(e, k, &r, &idx);
        const int32_t rec[2] = {(int32_t)r, idx};
        if (!rc && std::fwrite(rec, sizeof rec, 1, f) != 1) rc = fail(e, SCL_ERR_INVALID_ARG, "db_dump: short write");
    }
    if (std::fclose(f) != 0 && !rc) rc = fail(e, SCL_ERR_INVALID_ARG, "db_dump: close failed");
    return rc;
}

int scl_db_load_file(scl_engine *e, const char *path, int *n_loaded)
{
    if (!e || !path) return SCL_ERR_INVALID_ARG;
    if (n_loaded) *n_loaded = 0;
    FILE *f = std::fopen(path, "rb");
    if (!f) return fail(e, SCL_ERR_INVALID_ARG, "db_load: cannot open the file");
    const int n_before = scl_get_size(e, -1);
    DbFileHeader h{};
    int sink_rc = 0;
    // the parser (db_file.hpp: host code, fuzzed under ASan / UBSan by `make sanitize`) proves the file's length against its header before
    // anything is sized by it, and hands the descriptors over in chunks
    const int st = db_file_parse(f, e->R, e->S, &h, [&](const float *vals, int c, const int8_t *robots, const int *indexs) {
        const int rc = scl_save_bulk(e, vals, c, robots, indexs);
        if (!rc && n_loaded) *n_loaded += c;
        return rc;
    }, &sink_rc);
    std::fclose(f);
    if (st == DBF_SINK) return sink_rc;                     // (scl_save_bulk left its own message)
    if (st == DBF_NOMEM) return fail(e, SCL_ERR_NOMEM, db_file_status_string(st));
    if (st != DBF_OK) return fail(e, SCL_ERR_INVALID_ARG, db_file_status_string(st));
    // an engine that was empty takes over the dump's inter-robot tree state too (D.h:1691-1703): the next
    // detectInterLoopClosureID searches the range the dumped engine would have searched
    if (n_before == 0 && !e->front && h.reserved[0] == 1) {
        std::lock_guard<std::mutex> lk(e->mu);
        e->tree_counter = h.reserved[1]; e->tree_n = h.reserved[2];
    }
    return SCL_OK;
}

/* ---- pose algebra behind the ICP block (DM.h:1130-1141, 1249-1259) ------------------------ */

int scl_matrix_to_pose(const float T[16], float *x, float *y, float *z, float *roll, float *pitch, float *yaw)
{
    if (!T || !x || !y || !z || !roll || !pitch || !yaw) return SCL_ERR_INVALID_ARG;
    *x = T[3]; *y = T[7]; *z = T[11];                       /* pcl::getTranslationAndEulerAngles, float like the reference */
    *roll = std::atan2(T[9], T[10]);
    *pitch = std::asin(-T[8]);
    *yaw = std::atan2(T[4], T[0]);
    return SCL_OK;
}

int scl_loop_pose_between(const float T_icp[16], const float pose_cur[6], const float pose_pre[6], double between_xyz_q[7], double between_rpy[3])
{
    if (!T_icp || !pose_cur || !pose_pre || !between_xyz_q) return SCL_ERR_INVALID_ARG;
    float Tw[16], Tc[16];
    scl_pose_to_matrix(pose_cur[0], pose_cur[1], pose_cur[2], pose_cur[3], pose_cur[4], pose_cur[5], Tw);   /* tfWrong, DM.h:1137 */
    for (int r = 0; r < 4; ++r)                                                                              /* tfCorrect = tfICP * tfWrong (Affine3f), DM.h:1138 */
        for (int c = 0; c < 4; ++c) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += T_icp[4 * r + k] * Tw[4 * k + c];
            Tc[4 * r + c] = s;
        }
    float x, y, z, roll, pitch, yaw;
    scl_matrix_to_pose(Tc, &x, &y, &z, &roll, &pitch, &yaw);                                                 /* DM.h:1139 */
    auto rzryrx = [](double ro, double pi, double ya, double R[9]) {                                         /* gtsam::Rot3::RzRyRx */
        const double cr = std::cos(ro), sr = std::sin(ro), cp = std::cos(pi), sp = std::sin(pi), cy = std::cos(ya), sy = std::sin(ya);
        R[0] = cy * cp; R[1] = cy * sp * sr - sy * cr; R[2] = cy * sp * cr + sy * sr;
        R[3] = sy * cp; R[4] = sy * sp * sr + cy * cr; R[5] = sy * sp * cr - cy * sr;
        R[6] = -sp;     R[7] = cp * sr;                R[8] = cp * cr;
    };
    double Rf[9], Rt[9];
    rzryrx((double)roll, (double)pitch, (double)yaw, Rf);                                                    /* poseFrom, DM.h:1140 */
    rzryrx((double)pose_pre[3], (double)pose_pre[4], (double)pose_pre[5], Rt);                               /* poseTo, DM.h:1141 + 214-218 */
    const double tf[3] = {(double)x, (double)y, (double)z}, tt[3] = {(double)pose_pre[0], (double)pose_pre[1], (double)pose_pre[2]};
    double Rb[9], tb[3];                                                                                     /* poseFrom.between(poseTo) = poseFrom^-1 * poseTo */
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += Rf[3 * k + r] * Rt[3 * k + c]; Rb[3 * r + c] = s; }
        double s = 0; for (int k = 0; k < 3; ++k) s += Rf[3 * k + r] * (tt[k] - tf[k]);
        tb[r] = s;
    }
    between_xyz_q[0] = tb[0]; between_xyz_q[1] = tb[1]; between_xyz_q[2] = tb[2];
    // rotation matrix -> unit quaternion with w >= 0 (q and -q are the same rotation; the sign GTSAM / Eigen pick is not pinned)
    double qw, qx, qy, qz;
    const double tr = Rb[0] + Rb[4] + Rb[8];
    if (tr > 0) { const double s = std::sqrt(tr + 1.0) * 2; qw = 0.25 * s; qx = (Rb[7] - Rb[5]) / s; qy = (Rb[2] - Rb[6]) / s; qz = (Rb[3] - Rb[1]) / s; }
    else if (Rb[0] > Rb[4] && Rb[0] > Rb[8]) { const double s = std::sqrt(1.0 + Rb[0] - Rb[4] - Rb[8]) * 2; qw = (Rb[7] - Rb[5]) / s; qx = 0.25 * s; qy = (Rb[1] + Rb[3]) / s; qz = (Rb[2] + Rb[6]) / s; }
    else if (Rb[4] > Rb[8]) { const double s = std::sqrt(1.0 + Rb[4] - Rb[0] - Rb[8]) * 2; qw = (Rb[2] - Rb[6]) / s; qx = (Rb[1] + Rb[3]) / s; qy = 0.25 * s; qz = (Rb[5] + Rb[7]) / s; }
    else { const double s = std::sqrt(1.0 + Rb[8] - Rb[0] - Rb[4]) * 2; qw = (Rb[3] - Rb[1]) / s; qx = (Rb[2] + Rb[6]) / s; qy = (Rb[5] + Rb[7]) / s; qz = 0.25 * s; }
    if (qw < 0) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }
    between_xyz_q[3] = qx; between_xyz_q[4] = qy; between_xyz_q[5] = qz; between_xyz_q[6] = qw;
    if (between_rpy) {                                                                                       /* Rot3::roll/pitch/yaw (DM.h:1258) */
        between_rpy[0] = std::atan2(Rb[7], Rb[8]);
        between_rpy[1] = std::asin(-Rb[6] > 1 ? 1 : (-Rb[6] < -1 ? -1 : -Rb[6]));
        between_rpy[2] = std::atan2(Rb[3], Rb[0]);
    }
    return SCL_OK;
}

/* ---- geometric verification -------------------------------------------------- */

int scl_icp_default_params(scl_icp_params *p)
{
    if (!p) return SCL_ERR_INVALID_ARG;
    p->max_iterations = 50;               /* DM.h:1110 */
    p->max_correspondence_dist = 100.0;   /* DM.h:1109 */
    p->transformation_epsilon = 1e-6;     /* DM.h:1111 */
    p->euclidean_fitness_epsilon = 1e-6;  /* DM.h:1112 */
    p->estimator = 0;
    p->normal_radius = 1.0;
    return SCL_OK;
}

int scl_icp_align(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                  int stride_bytes, const scl_icp_params *p,
                  float T[16], float *fitness, int *converged, int *iterations)
{
    if (!e || !src || !tgt || !p || !T) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    std::string err;
    int rc = icp_align(&e->icp_ws, e->stream, e->num_cu, src, n_src, tgt, n_tgt, stride_bytes, *p,
                       T, fitness, converged, iterations, &err);
    if (rc) e->last_error = err;
    return rc;
}

int scl_icp_align_batch(scl_engine *e, const void *src, int n_src, const void *const *tgts, const int *n_tgts,
                        int n_targets, int stride_bytes, const scl_icp_params *p,
                        float *T, float *fitness, int *converged, int *iterations)
{
    if (!e || !src || !p || !T || n_targets < 0 || (n_targets > 0 && (!tgts || !n_tgts))) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_icp_align_batch(e, src, n_src, tgts, n_tgts, n_targets, stride_bytes, p, T, fitness, converged, iterations);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    const int lanes = n_targets < scl_engine::kIcpLanes ? n_targets : scl_engine::kIcpLanes;
    for (int l = 0; l < lanes; ++l) {
        if (!e->icp_lane_stream[l]) SCL_HIP(e, hipStreamCreateWithFlags(&e->icp_lane_stream[l], hipStreamNonBlocking));
        if (!e->ev_lane[l]) SCL_HIP(e, hipEventCreateWithFlags(&e->ev_lane[l], hipEventDisableTiming));
    }
    if (n_src < 0 || stride_bytes < 12 || (stride_bytes & 3)) return fail(e, SCL_ERR_INVALID_ARG, "icp_align_batch: bad cloud layout");
    for (int c = 0; c < n_targets; ++c)
        if ((!tgts[c] && n_tgts[c] > 0) || n_tgts[c] < 0) return fail(e, SCL_ERR_INVALID_ARG, "icp_align_batch: bad target");
    // the source crosses PCIe once; every alignment's working cloud is made from it on the device
    int rc0 = ensure_points(e, (size_t)n_src * stride_bytes + 16);
    if (rc0) return rc0;
    if (n_src) SCL_HIP(e, hipMemcpyAsync(e->d_points, src, (size_t)n_src * stride_bytes, hipMemcpyHostToDevice, e->stream));
    SCL_HIP(e, hipStreamSynchronize(e->stream));
    for (int first = 0; first < n_targets; first += scl_engine::kIcpBatch) {
        const int m = n_targets - first < scl_engine::kIcpBatch ? n_targets - first : scl_engine::kIcpBatch;
        // stage the targets and build their search grids (normals): independent per candidate, issued from a few host
        // threads onto the lane streams, so that one candidate's kernels run under another's PCIe copy (the batched chain of
        // scl_loop_icp_batch_from_store behind ALL the copies was measured slower here: 8.3 against 8.1 ms, 6.2 against 5.1 point to plane)
        std::atomic<int> next{0};
        std::atomic<int> first_rc{SCL_OK};
        std::string errs[scl_engine::kIcpLanes];
        auto worker = [&](int l) {
            (void)hipSetDevice(e->device);
            hipStream_t st = e->icp_lane_stream[l];
            for (;;) {
                const int c = next.fetch_add(1);
                if (c >= m) break;
                IcpWorkspace *ws = &e->icp_batch_ws[c];
                int rc = icp_stage_cloud_host(ws, st, true, tgts[first + c], n_tgts[first + c], stride_bytes, &errs[l]);
                if (!rc) rc = icp_batch_prepare(ws, st, e->d_points, n_src, n_tgts[first + c], stride_bytes, *p, &errs[l]);
                if (rc) { int ok = SCL_OK; first_rc.compare_exchange_strong(ok, rc); }
            }
            (void)hipEventRecord(e->ev_lane[l], st);
        };
        const int nl = m < lanes ? m : lanes;
        std::vector<std::thread> pool;
        for (int l = 1; l < nl; ++l) pool.emplace_back(worker, l);
        worker(0);
        for (auto &t : pool) t.join();
        int rc = first_rc.load();
        if (rc) { for (auto &msg : errs) if (!msg.empty()) { e->last_error = msg; break; } (void)hipDeviceSynchronize(); return rc; }
        for (int l = 0; l < nl; ++l) SCL_HIP(e, hipStreamWaitEvent(e->stream, e->ev_lane[l], 0));
        IcpWorkspace *wss[scl_engine::kIcpBatch];
        for (int c = 0; c < m; ++c) wss[c] = &e->icp_batch_ws[c];
        std::string err;
        rc = icp_batch_run(wss, m, &e->icp_batch_ctl, e->stream, e->d_points, n_src, stride_bytes, *p, T + 16 * (size_t)first,
                           fitness ? fitness + first : nullptr, converged ? converged + first : nullptr,
                           iterations ? iterations + first : nullptr, &err);
        if (rc) { e->last_error = err; return rc; }
    }
    return SCL_OK;
}

int scl_nn_correspondences(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                           int stride_bytes, int *nn_index, float *nn_dist2)
{
    if (!e || !src || !tgt || !nn_index) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    std::string err;
    int rc = icp_nn_correspondences(&e->icp_ws, e->stream, e->num_cu, src, n_src, tgt, n_tgt, stride_bytes, nullptr,
                                    nn_index, nn_dist2, &err);
    if (rc) e->last_error = err;
    return rc;
}

void scl_debug_icp_tile_stats(unsigned long long out[16], int reset) { icp_tile_stats(out, reset != 0); }

int scl_nn_correspondences_moved(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                                 int stride_bytes, const float T[16], int *nn_index, float *nn_dist2)
{
    if (!e || !src || !tgt || !nn_index || !T) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    std::string err;
    int rc = icp_nn_correspondences(&e->icp_ws, e->stream, e->num_cu, src, n_src, tgt, n_tgt, stride_bytes, T,
                                    nn_index, nn_dist2, &err);
    if (rc) e->last_error = err;
    return rc;
}

int scl_rigid_svd(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                  int stride_bytes, const int *src_index, const int *tgt_index, int n_corr, float T[16])
{
    if (!e || !src || !tgt || !src_index || !tgt_index || !T) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    std::string err;
    int rc = icp_rigid_svd(&e->icp_ws, e->stream, e->num_cu, src, n_src, tgt, n_tgt, stride_bytes,
                           src_index, tgt_index, n_corr, T, &err);
    if (rc) e->last_error = err;
    return rc;
}

int scl_ransac_correspondences(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                               int stride_bytes, const int *src_index, const int *tgt_index, int n_corr,
                               int max_iterations, double inlier_threshold, uint64_t seed,
                               int *inlier_mask, int *n_inliers, int *best_hypothesis, float T_model[16])
{
    if (!e || !src || !tgt || !src_index || !tgt_index) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    std::string err;
    int rc = icp_ransac(&e->icp_ws, e->stream, src, n_src, tgt, n_tgt, stride_bytes, src_index, tgt_index, n_corr,
                        max_iterations, inlier_threshold, (unsigned long long)seed, inlier_mask, n_inliers,
                        best_hypothesis, T_model, &err);
    if (rc) e->last_error = err;
    return rc;
}

int scl_geometric_verification(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                               int stride_bytes, int ransac_iterations, double inlier_threshold,
                               double inlier_ratio, uint64_t seed, float T[16], int *success,
                               int *n_correspondences, int *n_inliers)
{
    if (!e || !src || !tgt || !T) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    std::string err;
    int rc = icp_geometric_verification(&e->icp_ws, e->stream, e->num_cu, src, n_src, tgt, n_tgt, stride_bytes,
                                        ransac_iterations, inlier_threshold, inlier_ratio, (unsigned long long)seed,
                                        T, success, n_correspondences, n_inliers, &err);
    if (rc) e->last_error = err;
    return rc;
}

int scl_voxel_grid(scl_engine *e, const void *points, int n_points, int stride_bytes, float leaf,
                   void *out, int out_capacity, int *n_out)
{
    if (!e || (!points && n_points > 0) || !out || !n_out) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    std::string err;
    int rc = voxel_grid(&e->vox_ws, e->stream, points, n_points, stride_bytes, leaf, out, out_capacity, n_out, &err);
    if (rc) e->last_error = err;
    return rc;
}

int scl_pose_to_matrix(float x, float y, float z, float roll, float pitch, float yaw, float T[16])
{
    if (!T) return SCL_ERR_INVALID_ARG;
    const float A = std::cos(yaw), B = std::sin(yaw), C = std::cos(pitch), D = std::sin(pitch), E = std::cos(roll), F = std::sin(roll);
    const float DE = D * E, DF = D * F;
    T[0] = A * C; T[1] = A * DF - B * E; T[2] = B * F + A * DE; T[3] = x;
    T[4] = B * C; T[5] = A * E + B * DF; T[6] = B * DE - A * F; T[7] = y;
    T[8] = -D;    T[9] = C * F;          T[10] = C * E;         T[11] = z;
    T[12] = 0.f;  T[13] = 0.f;           T[14] = 0.f;           T[15] = 1.f;
    return SCL_OK;
}

int scl_assemble_submap(scl_engine *e, const void *const *clouds, const int *counts, const float *transforms,
                        int n_clouds, int stride_bytes, float leaf, void *out, int out_capacity, int *n_out)
{
    if (!e || (n_clouds > 0 && (!clouds || !counts || !transforms)) || !out || !n_out) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    std::string err;
    int rc = assemble_submap(&e->vox_ws, e->stream, clouds, counts, transforms, n_clouds, stride_bytes, leaf, out,
                             out_capacity, n_out, &err);
    if (rc) e->last_error = err;
    return rc;
}

int scl_transform_cloud(scl_engine *e, const void *in, int n, int stride_bytes, const float T[16], void *out)
{
    if (!e || !in || !out || !T) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    std::string err;
    int rc = icp_transform_cloud(&e->icp_ws, e->stream, in, n, stride_bytes, T, out, &err);
    if (rc) e->last_error = err;
    return rc;
}

/* ---- on-device keyframe store ------------------------------------------------ */

namespace {

constexpr size_t kKfSlabBytes = (size_t)256 << 20;         // 256 MiB slabs: ~100 keyframes of 100 k points each

int kf_alloc(scl_engine *e, size_t bytes, unsigned char **out)
{
    bytes = (bytes + 255) & ~(size_t)255;
    if (e->kf_slabs.empty() || e->kf_slab_used + bytes > e->kf_slab_cap) {
        const size_t cap = bytes > kKfSlabBytes ? bytes : kKfSlabBytes;
        void *slab = nullptr;
        if (hipMalloc(&slab, cap) != hipSuccess) return fail(e, SCL_ERR_NOMEM, "keyframe store: hipMalloc failed");
        e->kf_slabs.push_back(slab);
        e->kf_slab_used = 0;
        e->kf_slab_cap = cap;
    }
    *out = static_cast<unsigned char *>(e->kf_slabs.back()) + e->kf_slab_used;
    e->kf_slab_used += bytes;
    return SCL_OK;
}

// keyframes key-search_num .. key+search_num that exist in the store (DM.h:1168-1176); poses[i] belongs to
// keyframe key - search_num + i
int kf_window(scl_engine *e, int robot, int key, int search_num, const float *poses,
              std::vector<const void *> *clouds, std::vector<int> *counts, std::vector<float> *T)
{
    if (robot < 0 || search_num < 0 || !poses) return fail(e, SCL_ERR_INVALID_ARG, "keyframe window: bad arguments");
    clouds->clear(); counts->clear(); T->clear();
    if ((size_t)robot >= e->kf.size()) return SCL_OK;
    const auto &arr = e->kf[robot];
    for (int i = -search_num; i <= search_num; ++i) {
        const long long k = (long long)key + i;
        if (k < 0 || k >= (long long)arr.size()) continue;             // DM.h:1171-1174
        const auto &c = arr[(size_t)k];
        if (c.n < 0) return fail(e, SCL_ERR_INVALID_ARG, "keyframe window: a keyframe inside the window was never stored");
        clouds->push_back(c.d);
        counts->push_back(c.n);
        const float *m = poses + 16 * (size_t)(i + search_num);
        T->insert(T->end(), m, m + 16);
    }
    return SCL_OK;
}

}  // namespace

int scl_keyframe_put(scl_engine *e, int robot, int index, const void *points, int n_points, int stride_bytes)
{
    if (!e || robot < 0 || index < 0 || n_points < 0 || (!points && n_points > 0)) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    if (stride_bytes < 12 || (stride_bytes & 3)) return fail(e, SCL_ERR_INVALID_ARG, "keyframe_put: bad stride");
    if (e->kf_stride == 0) e->kf_stride = stride_bytes;
    if (stride_bytes != e->kf_stride) return fail(e, SCL_ERR_INVALID_ARG, "keyframe_put: stride differs from the store's");
    (void)hipSetDevice(e->device);
    if ((size_t)robot >= e->kf.size()) e->kf.resize((size_t)robot + 1);
    auto &arr = e->kf[robot];
    if ((size_t)index >= arr.size()) arr.resize((size_t)index + 1);
    auto &c = arr[(size_t)index];
    const size_t bytes = (size_t)n_points * stride_bytes;
    if (bytes > c.cap_bytes) {                                         // replacing with a larger cloud: old space is retired
        unsigned char *d = nullptr;
        int rc = kf_alloc(e, bytes, &d);
        if (rc) return rc;
        c.d = d; c.cap_bytes = (bytes + 255) & ~(size_t)255;
    }
    if (bytes) {
        SCL_HIP(e, hipMemcpyAsync(c.d, points, bytes, hipMemcpyHostToDevice, e->stream));
        SCL_HIP(e, hipStreamSynchronize(e->stream));                   // the caller's buffer is free on return
    }
    c.n = n_points;
    return SCL_OK;
}

int scl_keyframe_count(const scl_engine *e, int robot)
{
    if (!e || robot < 0) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    return (size_t)robot < e->kf.size() ? (int)e->kf[robot].size() : 0;
}

int scl_keyframe_get(scl_engine *e, int robot, int index, void *out, int out_capacity, int *n_points)
{
    if (!e || robot < 0 || index < 0 || !n_points) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    if ((size_t)robot >= e->kf.size() || (size_t)index >= e->kf[robot].size() || e->kf[robot][index].n < 0)
        return fail(e, SCL_ERR_INVALID_ARG, "keyframe_get: no such keyframe");
    const auto &c = e->kf[robot][index];
    *n_points = c.n;
    if (!out) return SCL_OK;
    if (c.n > out_capacity) return fail(e, SCL_ERR_INVALID_ARG, "keyframe_get: output capacity too small");
    (void)hipSetDevice(e->device);
    if (c.n) {
        SCL_HIP(e, hipMemcpyAsync(out, c.d, (size_t)c.n * e->kf_stride, hipMemcpyDeviceToHost, e->stream));
        SCL_HIP(e, hipStreamSynchronize(e->stream));
    }
    return SCL_OK;
}

int scl_submap_from_store(scl_engine *e, int robot, int key, int search_num, const float *poses, float leaf,
                          void *out, int out_capacity, int *n_out)
{
    if (!e || !out || !n_out) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    std::vector<const void *> clouds; std::vector<int> counts; std::vector<float> T;
    int rc = kf_window(e, robot, key, search_num, poses, &clouds, &counts, &T);
    if (rc) return rc;
    std::string err;
    rc = assemble_submap_ex(&e->vox_ws, e->stream, clouds.data(), counts.data(), T.data(), (int)clouds.size(),
                            e->kf_stride ? e->kf_stride : 16, leaf, true, out, out_capacity, nullptr, n_out, &err);
    if (rc) e->last_error = err;
    return rc;
}

int scl_loop_icp_from_store(scl_engine *e, int robot, int key_cur, const float *pose_cur,
                            int key_pre, int search_num, const float *poses_pre, float leaf,
                            const scl_icp_params *p, int min_src_points, int min_tgt_points,
                            float T[16], float *fitness, int *converged, int *iterations, int *n_src, int *n_tgt)
{
    if (!e || !pose_cur || !poses_pre || !p || !T) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    const int stride = e->kf_stride ? e->kf_stride : 16;
    std::vector<const void *> clouds; std::vector<int> counts; std::vector<float> Tw;
    std::string err;
    int ns = 0, nt = 0;
    const void *d_res = nullptr;
    // source: loopFindNearKeyframes(cur, 0), DM.h:1105
    int rc = kf_window(e, robot, key_cur, 0, pose_cur, &clouds, &counts, &Tw);
    if (rc) return rc;
    rc = assemble_submap_ex(&e->vox_ws, e->stream, clouds.data(), counts.data(), Tw.data(), (int)clouds.size(), stride, leaf,
                            true, nullptr, 0, &d_res, &ns, &err);
    if (!rc) rc = icp_stage_cloud(&e->icp_ws, e->stream, false, d_res, ns, stride, &err);
    if (rc) { e->last_error = err; return rc; }
    // target: loopFindNearKeyframes(pre, historyKeyframeSearchNum), DM.h:1107
    rc = kf_window(e, robot, key_pre, search_num, poses_pre, &clouds, &counts, &Tw);
    if (rc) return rc;
    rc = assemble_submap_ex(&e->vox_ws, e->stream, clouds.data(), counts.data(), Tw.data(), (int)clouds.size(), stride, leaf,
                            true, nullptr, 0, &d_res, &nt, &err);
    if (!rc) rc = icp_stage_cloud(&e->icp_ws, e->stream, true, d_res, nt, stride, &err);
    if (rc) { e->last_error = err; return rc; }
    if (n_src) *n_src = ns;
    if (n_tgt) *n_tgt = nt;
    if (converged) *converged = 0;
    if (iterations) *iterations = 0;
    if (fitness) *fitness = 0.0f;
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    if (ns < min_src_points || nt < min_tgt_points) return SCL_OK;    // DM.h:1108-1111: too small, no alignment attempted
    rc = icp_align_staged(&e->icp_ws, e->stream, ns, nt, stride, *p, T, fitness, converged, iterations, &err);
    if (rc) e->last_error = err;
    return rc;
}

int scl_loop_icp_batch_from_store(scl_engine *e, int robot, int key_cur, const float *pose_cur,
                                  int n_candidates, const int *keys_pre, int search_num, const float *poses_pre, float leaf,
                                  const scl_icp_params *p, int min_src_points, int min_tgt_points,
                                  float *T, float *fitness, int *converged, int *iterations, int *n_src, int *n_tgts)
{
    if (!e || !pose_cur || !p || !T || n_candidates < 0 || (n_candidates > 0 && (!keys_pre || !poses_pre))) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    const int stride = e->kf_stride ? e->kf_stride : 16;
    std::vector<const void *> clouds; std::vector<int> counts; std::vector<float> Tw;
    std::string err;
    const int win = 2 * search_num + 1;
    for (int c = 0; c < n_candidates; ++c) {
        for (int i = 0; i < 16; ++i) T[16 * (size_t)c + i] = (i % 5 == 0) ? 1.0f : 0.0f;
        if (fitness) fitness[c] = 0.0f;
        if (converged) converged[c] = 0;
        if (iterations) iterations[c] = 0;
        if (n_tgts) n_tgts[c] = 0;
    }
    if (n_src) *n_src = 0;
    int rc = SCL_OK;
    for (int first = 0; first < n_candidates || first == 0; first += scl_engine::kIcpBatch) {
        const int m = n_candidates - first < scl_engine::kIcpBatch ? n_candidates - first : scl_engine::kIcpBatch;
        // The submaps of a round -- the scan's own (loopFindNearKeyframes(cur, 0), DM.h:1105) in front, then the candidates'
        // (loopFindNearKeyframes(pre, historyKeyframeSearchNum), DM.h:1107) -- are assembled and filtered TOGETHER: every step of the
        // voxel filter is one launch over all of them (voxel.hip, assemble_submaps_batch), where each used to be a chain of ~20 short
        // launches on its lane.  (The scan's submap is redone per round of kIcpBatch candidates: rounds beyond the first are rare.)
        std::vector<const void *> cl_all; std::vector<int> cn_all, first_of; std::vector<float> tw_all;
        first_of.push_back(0);
        {
            clouds.clear(); counts.clear(); Tw.clear();
            if ((rc = kf_window(e, robot, key_cur, 0, pose_cur, &clouds, &counts, &Tw))) return rc;
            cl_all.insert(cl_all.end(), clouds.begin(), clouds.end()); cn_all.insert(cn_all.end(), counts.begin(), counts.end()); tw_all.insert(tw_all.end(), Tw.begin(), Tw.end());
            first_of.push_back((int)cl_all.size());
        }
        for (int c = 0; c < m; ++c) {
            clouds.clear(); counts.clear(); Tw.clear();
            if ((rc = kf_window(e, robot, keys_pre[first + c], search_num, poses_pre + (size_t)(first + c) * win * 16, &clouds, &counts, &Tw))) return rc;
            cl_all.insert(cl_all.end(), clouds.begin(), clouds.end()); cn_all.insert(cn_all.end(), counts.begin(), counts.end()); tw_all.insert(tw_all.end(), Tw.begin(), Tw.end());
            first_of.push_back((int)cl_all.size());
        }
        std::vector<const void *> d_sub((size_t)m + 1); std::vector<int> n_sub((size_t)m + 1);
        rc = assemble_submaps_batch(&e->vox_ws, e->stream, cl_all.data(), cn_all.data(), tw_all.data(), first_of.data(), m + 1, stride, leaf,
                                    d_sub.data(), n_sub.data(), &err);
        if (rc) { e->last_error = err; return rc; }
        const int ns = n_sub[0];
        const void *d_src = d_sub[0];                                                    // (stays in the filter's workspace until the round is over)
        if (n_src) *n_src = ns;
        if (m <= 0) break;
        // search grids (+ normals) of the candidates that are large enough (DM.h:1108), their submaps read where the filter left them:
        // one launch per step over all of them (icp.hip, icp_batch_prepare_all), on the engine's stream
        IcpWorkspace *wss[scl_engine::kIcpBatch]; int which[scl_engine::kIcpBatch]; int live = 0;
        const void *tg[scl_engine::kIcpBatch]; int tn[scl_engine::kIcpBatch];
        for (int c = 0; c < m; ++c) {
            const int nt = n_sub[(size_t)c + 1];
            if (n_tgts) n_tgts[first + c] = nt;
            if (ns < min_src_points || nt < min_tgt_points) continue;
            wss[live] = &e->icp_batch_ws[c]; which[live] = first + c; tg[live] = d_sub[(size_t)c + 1]; tn[live] = nt; ++live;
        }
        if (live == 0) continue;
        rc = icp_batch_prepare_all(wss, live, &e->icp_batch_ctl, e->stream, ns, tg, tn, stride, *p, &err);
        if (rc) { e->last_error = err; return rc; }
        std::vector<float> Tl(16 * (size_t)live), fl((size_t)live); std::vector<int> cl((size_t)live), il((size_t)live);
        rc = icp_batch_run(wss, live, &e->icp_batch_ctl, e->stream, d_src, ns, stride, *p, Tl.data(), fl.data(), cl.data(), il.data(), &err);
        if (rc) { e->last_error = err; return rc; }
        for (int j = 0; j < live; ++j) {
            std::memcpy(T + 16 * (size_t)which[j], Tl.data() + 16 * (size_t)j, 16 * sizeof(float));
            if (fitness) fitness[which[j]] = fl[(size_t)j];
            if (converged) converged[which[j]] = cl[(size_t)j];
            if (iterations) iterations[which[j]] = il[(size_t)j];
        }
    }
    return SCL_OK;
}

int scl_geometric_verification_from_store(scl_engine *e, const void *src, int n_src, int stride_bytes, float src_leaf,
                                          int robot, int key_pre, int search_num, const float *poses_pre, float leaf,
                                          int min_src_points, int min_tgt_points,
                                          int ransac_iterations, double inlier_threshold, double inlier_ratio, uint64_t seed,
                                          float T[16], int *success, int *n_src_filtered, int *n_tgt,
                                          int *n_correspondences, int *n_inliers)
{
    if (!e || (!src && n_src > 0) || !poses_pre || !T) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    const int stride = e->kf_stride ? e->kf_stride : stride_bytes;
    if (stride != stride_bytes) return fail(e, SCL_ERR_INVALID_ARG, "geometric_verification_from_store: stride differs from the store's");
    std::string err;
    int ns = 0, nt = 0, rc;
    const void *d_res = nullptr;
    // received cloud: downSizeFilterICP (DM.h:1199-1201), stays on the device
    rc = voxel_grid_to_device(&e->vox_ws, e->stream, src, n_src, stride, src_leaf, &d_res, &ns, &err);
    if (!rc) rc = icp_stage_cloud(&e->icp_ws, e->stream, false, d_res, ns, stride, &err);
    if (rc) { e->last_error = err; return rc; }
    // submap around the matched keyframe (DM.h:1202) from the keyframe store
    std::vector<const void *> clouds; std::vector<int> counts; std::vector<float> Tw;
    rc = kf_window(e, robot, key_pre, search_num, poses_pre, &clouds, &counts, &Tw);
    if (rc) return rc;
    rc = assemble_submap_ex(&e->vox_ws, e->stream, clouds.data(), counts.data(), Tw.data(), (int)clouds.size(), stride, leaf,
                            true, nullptr, 0, &d_res, &nt, &err);
    if (!rc) rc = icp_stage_cloud(&e->icp_ws, e->stream, true, d_res, nt, stride, &err);
    if (rc) { e->last_error = err; return rc; }
    if (n_src_filtered) *n_src_filtered = ns;
    if (n_tgt) *n_tgt = nt;
    for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0f : 0.0f;
    if (success) *success = 0;
    if (n_correspondences) *n_correspondences = 0;
    if (n_inliers) *n_inliers = 0;
    if (ns < min_src_points || nt < min_tgt_points) return SCL_OK;    // DM.h:1204: clouds too small
    rc = icp_geometric_verification_staged(&e->icp_ws, e->stream, ns, nt, stride, ransac_iterations, inlier_threshold,
                                           inlier_ratio, (unsigned long long)seed, T, success, n_correspondences, n_inliers, &err);
    if (rc) e->last_error = err;
    return rc;
}

/* ---- measurement ------------------------------------------------------------- */

int scl_profile_enable(scl_engine *e, int on)
{
    if (!e) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_profile_enable(e, on);
    std::lock_guard<std::mutex> lk(e->mu);
    e->prof_on = on < 0 ? 0 : (on > 3 ? 1 : on);
    e->prof_tick = 0;
    return SCL_OK;
}

int scl_profile_reset(scl_engine *e)
{
    if (!e) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_profile_reset(e);
    std::lock_guard<std::mutex> lk(e->mu);
    std::memset(&e->prof, 0, sizeof e->prof);
    return SCL_OK;
}

int scl_profile_get(scl_engine *e, scl_profile *out)
{
    if (!e || !out) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_profile_get(e, out);
    std::lock_guard<std::mutex> lk(e->mu);
    *out = e->prof;
    return SCL_OK;
}

int scl_alignment_stats(scl_engine *e, uint64_t *pairs, uint64_t *fallbacks, int reset)
{
    if (!e) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_alignment_stats(e, pairs, fallbacks, reset);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    unsigned long long f = 0;
    SCL_HIP(e, hipStreamSynchronize(e->stream));
    SCL_HIP(e, hipMemcpy(&f, e->d_align_fallbacks, sizeof f, hipMemcpyDeviceToHost));
    if (pairs) *pairs = e->align_pairs;
    if (fallbacks) *fallbacks = (uint64_t)f;
    if (reset) {
        SCL_HIP(e, hipMemset(e->d_align_fallbacks, 0, sizeof f));
        e->align_pairs = 0;
    }
    return SCL_OK;
}

int scl_survivor_stats(scl_engine *e, uint64_t *queries, uint64_t *survivors, uint64_t *max_survivors, int reset)
{
    if (!e) return SCL_ERR_INVALID_ARG;
    if (e->front) return front_survivor_stats(e, queries, survivors, max_survivors, reset);
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    unsigned long long s[3] = {0, 0, 0};
    SCL_HIP(e, hipStreamSynchronize(e->stream));
    if (e->stream_surv) SCL_HIP(e, hipStreamSynchronize(e->stream_surv));
    SCL_HIP(e, hipMemcpy(s, e->d_surv_stats, sizeof s, hipMemcpyDeviceToHost));
    if (survivors) *survivors = (uint64_t)s[0];
    if (max_survivors) *max_survivors = (uint64_t)s[1];
    if (queries) *queries = (uint64_t)s[2];
    if (reset) SCL_HIP(e, hipMemset(e->d_surv_stats, 0, sizeof s));
    return SCL_OK;
}

int scl_device_name(const scl_engine *e, char *buf, int buflen)
{
    if (!e || !buf || buflen <= 0) return SCL_ERR_INVALID_ARG;
    if (e->front) e = front_primary(e);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, e->device) != hipSuccess) return SCL_ERR_HIP;
    std::snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return SCL_OK;
}

}  // extern "C"

// ---- hooks for the sharded front (engine_internal.hpp) ---------------------------------------------------------
namespace scl {

int eng_stage_from_peer(scl_engine *dst, int j, scl_engine *src, int src_slot, int count)
{
    if (!dst || !src || count < 1 || j < 0 || j + count > dst->stage_rows) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(dst->mu);
    if (src_slot < 0 || src_slot + count > src->n) return fail(dst, SCL_ERR_OUT_OF_RANGE, "stage_from_peer: source slot out of range");
    (void)hipSetDevice(dst->device);
    const size_t tile = (size_t)dst->RG * dst->S, S = (size_t)dst->S, R4 = (size_t)dst->R4, m = (size_t)count;
    const size_t d = (size_t)dst->cap + (size_t)j, s = (size_t)src_slot;
    // consecutive slots are consecutive rows of every array on both sides (the strides agree: same grid): one copy per array
    if (dst->hstride != src->hstride || dst->hkw != src->hkw) return fail(dst, SCL_ERR_INVALID_ARG, "stage_from_peer: the shards' layouts differ");
    SCL_HIP(dst, hipMemcpyPeerAsync(dst->d_desc + d * tile, dst->device, src->d_desc + s * tile, src->device, sizeof(float4) * tile * m, dst->stream));
    SCL_HIP(dst, hipMemcpyPeerAsync(dst->d_vkey + d * S, dst->device, src->d_vkey + s * S, src->device, sizeof(double) * S * m, dst->stream));
    SCL_HIP(dst, hipMemcpyPeerAsync(dst->d_norm + d * S, dst->device, src->d_norm + s * S, src->device, sizeof(double) * S * m, dst->stream));
    SCL_HIP(dst, hipMemcpyPeerAsync(dst->d_rkey + d * R4, dst->device, src->d_rkey + s * R4, src->device, sizeof(float) * R4 * m, dst->stream));
    SCL_HIP(dst, hipMemcpyPeerAsync(dst->d_hdesc + d * dst->hstride, dst->device, src->d_hdesc + s * src->hstride, src->device, sizeof(uint2) * src->hstride * m, dst->stream));
    SCL_HIP(dst, hipMemcpyPeerAsync(dst->d_kmask + d * 8, dst->device, src->d_kmask + s * 8, src->device, sizeof(unsigned int) * 8 * m, dst->stream));
    SCL_HIP(dst, hipMemcpyPeerAsync(dst->d_hkey + d * dst->hkw, dst->device, src->d_hkey + s * src->hkw, src->device, sizeof(unsigned short) * src->hkw * m, dst->stream));
    for (int i = 0; i < count; ++i) dst->staged[j + i] = true;
    return SCL_OK;
}

int eng_set_stage_rows(scl_engine *e, int rows)
{
    if (!e || rows < scl_engine::kStage) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->n > 0) return fail(e, SCL_ERR_INVALID_ARG, "stage rows are fixed once a keyframe is stored");
    (void)hipSetDevice(e->device);
    const int old_cap = e->cap;
    e->stage_rows = rows;
    e->staged.assign((size_t)rows, 0);
    e->cap = 0;                                            // empty database: new arrays of the new shape, nothing to copy (the old ones are freed there)
    return ensure_capacity(e, old_cap > 0 ? old_cap : 1);
}

int eng_stage_values(scl_engine *e, int j, const float *values)
{
    if (!e || !values || j < 0 || j >= e->stage_rows) return SCL_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    int rc;
    const size_t cells = (size_t)e->R * e->S;
    if ((rc = ensure_vals(e, cells))) return rc;
    SCL_HIP(e, hipMemcpyAsync(e->d_vals, values, sizeof(float) * cells, hipMemcpyHostToDevice, e->stream));
    {
        ProfScope ps(e, P_INGEST);
        SCL_HIP(e, launch_ingest(e->d_vals, 1, e->cap + j, e->d_desc, e->d_vkey, e->d_norm, e->d_rkey, nullptr, e->d_hdesc, e->d_kmask, e->d_hkey, e->hstride,
                                 e->cap, e->R, e->S, e->stream));
    }
    if ((rc = sync(e))) return rc;
    e->staged[j] = true;
    return SCL_OK;
}

int eng_topk_enqueue(scl_engine *e, int query, int lo, int hi, int k, float eps, bool want_dist, bool *have_dist)
{
    std::lock_guard<std::mutex> pk(e->pass_mu);            // pass state (pinned results, slots, buffer sets): pass_mu first, like every scoring entry point
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    return topk_enqueue_locked(e, query, lo, hi, k, eps, want_dist, have_dist);
}

int eng_topk_finish(scl_engine *e, int k, bool have_dist, int *idx, float *d2, double *dist, int *shift, int *found)
{
    std::lock_guard<std::mutex> pk(e->pass_mu);            // pass state (pinned results, slots, buffer sets): pass_mu first, like every scoring entry point
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    return topk_finish_locked(e, k, have_dist, idx, d2, dist, shift, found);
}

int eng_sync_streams(scl_engine *e)
{
    std::lock_guard<std::mutex> lk(e->mu);
    (void)hipSetDevice(e->device);
    SCL_HIP(e, hipStreamSynchronize(e->stream));
    if (e->stream_alt) SCL_HIP(e, hipStreamSynchronize(e->stream_alt));
    if (e->stream2) SCL_HIP(e, hipStreamSynchronize(e->stream2));
    return SCL_OK;
}

bool eng_would_regrow(const scl_engine *e, int count)
{
    std::lock_guard<std::mutex> lk(e->mu);
    return e->n + count > e->cap;
}

// Drop the keyframes from n_keep on (the sharded front undoes a multi-shard append that failed on a later shard; the slots'
// contents are simply overwritten by the next append).  Passes in flight must have been collected by the caller.
int eng_truncate(scl_engine *e, int n_keep)
{
    std::lock_guard<std::mutex> lk(e->mu);
    if (n_keep < 0 || n_keep > e->n) return SCL_ERR_INVALID_ARG;
    e->robots.resize((size_t)n_keep); e->indexs.resize((size_t)n_keep);
    e->n = n_keep;
    return SCL_OK;
}

const double *eng_ticket_record(const scl_engine *e, int ticket, int *slot_lo, bool *empty)
{
    std::lock_guard<std::mutex> pk(e->pass_mu);
    std::lock_guard<std::mutex> lk(e->mu);
    if (ticket < 0 || ticket >= scl_engine::kSlots || !e->slot_busy[ticket]) return nullptr;
    *slot_lo = e->slot_lo[ticket];
    *empty = e->slot_empty[ticket];
    return e->h_out3 + (size_t)ticket * 8;
}

hipStream_t eng_stream(const scl_engine *e) { return e->stream; }

int eng_release_ticket(scl_engine *e, int ticket)
{
    std::lock_guard<std::mutex> pk(e->pass_mu);            // pass state (pinned results, slots, buffer sets): pass_mu first, like every scoring entry point
    std::lock_guard<std::mutex> lk(e->mu);
    if (ticket < 0 || ticket >= scl_engine::kSlots || !e->slot_busy[ticket]) return fail(e, SCL_ERR_INVALID_ARG, "unknown ticket");
    e->slot_busy[ticket] = false;
    collect_profile(e);
    return SCL_OK;
}

}  // namespace scl
