// sharded_front.hip -- ONE keyframe database spread over several GPUs behind the same C ABI
// (include/scl_engine.h: scl_create_sharded).  One process, one engine state per device.
//
// Global keyframe g lives on shard g % G in local slot g / G (round robin keeps the shards balanced while
// the database grows append-only).  Every (query, keyframe) pair is independent, so scoring needs no
// data-path exchange; what is exchanged is
//   * reference-faithful detection (descriptor.h:1613-1674, 1676-1756): every shard's local ring-key top-k with
//     the SC distance / shift of those k (k records per shard, merged on the host: ascending ring distance,
//     ties -> lowest global index, then the reference's candidate loop with its float narrowing);
//   * full-DB detection: the per-shard (distance, index, shift) winners, reduced either on the host (G records of
//     24 bytes in pinned memory) or on the devices with two RCCL min all-reduces on packed 64-bit keys
//     (ncclAllReduce(ncclUint64, ncclMin): first the order-preserving image of the fp64 distance, then
//     index << 16 | shift of the shards that hold the minimum) -- the same reduction scl_slam_amd/sharded.py
//     runs across processes.
// The search-range rule [0, cur - NUM_EXCLUDE_RECENT) (descriptor.h:1627) is applied on GLOBAL indices: shard c
// may use local slots l with l * G + c < cur - exclude.
// A query keyframe lives on one shard; the others get a device-to-device copy of its slot in one of their
// staging slots (engine_internal.hpp), so the same kernels run everywhere.
#include "engine_internal.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define SCL_HAVE_RCCL_HEADER 1
#endif

namespace scl {

namespace {

constexpr int kMaxShards = 16;
constexpr int kFrontSlots = 8;                             // full-DB passes in flight at the front (= the shards' slots)
constexpr unsigned long long kNoKey = ~0ull;
static_assert(1 + kFrontSlots < scl_engine::kStage, "staging slots: 0 public, 1..8 passes in flight, 9 blocking calls");
constexpr int kStreamStage0 = 12, kStreamBlock = 64;       // staging rows of the stream form's block (keyframes that are not mirrored)
// Mirror rows: behind the staging rows every shard keeps the query-side rows (descriptor, keys, norms, fp16 copy,
// masks) of the newest kMirrorPerOwner keyframes of EVERY other shard, copied device to device when the keyframe is appended.  A scan is
// searched for right after it was appended, so the shards that do not own it find it in place: no copy at query time, whatever the
// order of the queries.  Row of slot s of owner o: kStage + o kMirrorPerOwner + s mod kMirrorPerOwner (consecutive slots = consecutive
// rows: a bulk append travels in one copy per array and destination).  Older keyframes still go through the staging rows.
constexpr int kMirrorPerOwner = 1024;
constexpr int kStreamCall = 1024;                          // scans per shard call of the stream form (a block ends earlier when its 64 staging rows are used up)
static_assert(kStreamStage0 > 1 + kFrontSlots && kStreamStage0 + kStreamBlock <= scl_engine::kStage, "staging rows of the stream form");

int local_count(int global_hi, int c, int G)
{   // number of local slots l of shard c with l * G + c < global_hi
    if (global_hi <= c) return 0;
    return (global_hi - c + G - 1) / G;
}

// ---- RCCL, loaded on demand (the library is large; a one-GPU engine never touches it) -------------------------
#ifdef SCL_HAVE_RCCL_HEADER
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    bool shared_devices_ok = false;                        // the library says it can run several ranks on one device (RCCL cannot; the tests' stand-in can)
};

Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // SCL_RCCL_LIB: the collective library to load instead of the system's librccl (a newer build of it -- or the stand-in the tests
        // link, tests/cpp/mock_rccl.cpp, which forms the ranks' element-wise minimum on the host so that the G > 1 control flow of the
        // exchange runs on a one-GPU box).  If it is set and cannot be loaded there is no fallback.
        if (const char *path = getenv("SCL_RCCL_LIB")) {
            r.lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.lib) break;
            }
        }
        if (!r.lib) return;
        r.shared_devices_ok = dlsym(r.lib, "scl_collective_allows_shared_devices") != nullptr;
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.ok = r.CommInitAll && r.CommDestroy && r.AllReduce && r.GroupStart && r.GroupEnd;
    });
    return &r;
}

#endif

// ---- device side of the RCCL exchange ---------------------------------------------------------------------------
// rec[i] -> the pinned (device-visible) record {dist, position or -1, shift} the fused SC kernel wrote for pass i of
// this shard.  key1 = order-preserving image of the distance (no candidate: all ones); key2 = global index << 16 | shift.
struct PackArgs {
    const double *rec[kMaxQueryBatch]; int slot_lo[kMaxQueryBatch]; int empty[kMaxQueryBatch];
    int m, shard, G;
};

__global__ void pack_winner_keys_kernel(PackArgs a, unsigned long long *key1, unsigned long long *key2)
{
    const int i = threadIdx.x;
    if (i >= a.m) return;
    unsigned long long k1 = kNoKey, k2 = kNoKey;
    if (!a.empty[i]) {
        const volatile double *r = a.rec[i];
        const double d = r[0], pos = r[1];
        if (pos >= 0.0) {
            const unsigned long long b = (unsigned long long)__double_as_longlong(d);
            k1 = (b >> 63) ? ~b : (b | 0x8000000000000000ull);           // IEEE order -> unsigned order
            const unsigned long long g = (unsigned long long)(a.slot_lo[i] + (int)pos) * (unsigned long long)a.G + (unsigned long long)a.shard;
            k2 = (g << 16) | ((unsigned long long)(int)r[2] & 0xffffull);
        }
    }
    key1[i] = k1; key2[i] = k2;
}

__global__ void select_winner_keys_kernel(const unsigned long long *key1, const unsigned long long *min1,
                                          unsigned long long *key2, int m)
{   // only the shards that hold the minimum distance take part in the second reduction
    const int i = threadIdx.x;
    if (i < m && key1[i] != min1[i]) key2[i] = kNoKey;
}

double key_to_dist(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double d; std::memcpy(&d, &b, 8);
    return d;
}

}  // namespace

struct ShardedFront {
    int G = 0;
    scl_engine *sh[kMaxShards] = {nullptr};
    int dev[kMaxShards] = {0};
    int n = 0;                                             // keyframes in the whole database
    std::vector<int8_t> robots;                            // (robot, index) map, replicated on the host (D.h:1758-1761)
    std::vector<int> indexs;
    int tree_counter = 0, tree_n = 0;                      // D.h:1691-1703
    bool staged0 = false;                                  // scl_stage_query was called on the front
    int mirror_lo[kMaxShards] = {0};                       // per owner: slots [mirror_lo, its size) have been copied to the other shards' mirror rows
    bool mirror_on = false;

    struct Pass { bool busy = false; int tk[kMaxShards]; int group = -1; };
    Pass pass[kFrontSlots];
    unsigned next_pass = 0;
    int last_pass = -1;

    // device-side exchange (exchange == 2)
    int exchange = 1;                                      // 1 host merge, 2 RCCL min all-reduce on packed keys
#ifdef SCL_HAVE_RCCL_HEADER
    ncclComm_t comm[kMaxShards] = {nullptr};
    Rccl *coll = nullptr;                                  // the collective's function table (librccl, or what SCL_RCCL_LIB names)
#endif
    unsigned long long *d_key[kMaxShards] = {nullptr};     // per shard: kFrontSlots groups x {key1, key2, min1, min2}[kMaxQueryBatch]
    unsigned long long *h_keys = nullptr;                  // pinned: kFrontSlots groups x {min1, min2}[kMaxQueryBatch]
    hipEvent_t ev_group[kFrontSlots] = {nullptr};
    struct Group { bool active = false; int m = 0; int first_ticket = -1; bool delivered = false;
                   double dist[kMaxQueryBatch]; int idx[kMaxQueryBatch]; int shift[kMaxQueryBatch]; };
    Group group[kFrontSlots];
    unsigned next_group = 0;
};

namespace {

int ffail(const scl_engine *e, int code, const std::string &msg)
{
    e->last_error = msg;
    return code;
}

int child_fail(const scl_engine *e, const scl_engine *child, int rc, const char *where)
{
    e->last_error = std::string(where) + ": " + scl_last_error(child);
    return rc;
}

inline int mirror_row(int owner, int slot) { return scl_engine::kStage + owner * kMirrorPerOwner + slot % kMirrorPerOwner; }

// is global keyframe g in the other shards' mirror rows?
inline bool mirrored(const ShardedFront *f, int g)
{
    if (!f->mirror_on || g < 0 || g >= f->n) return false;
    const int o = g % f->G, s = g / f->G;
    return s >= f->mirror_lo[o] && s >= local_count(f->n, o, f->G) - kMirrorPerOwner;
}

// slots [s0, s0 + cnt) of owner o have just been appended (the owner's ingest has completed): copy their rows to the mirror rows of
// every other shard.  A failure only switches the mirror off for what it could not copy: those keyframes take the staging rows.
void mirror_appended(scl_engine *e, int o, int s0, int cnt)
{
    ShardedFront *f = e->front;
    if (!f->mirror_on || cnt <= 0) return;
    const int end = s0 + cnt;
    if (cnt > kMirrorPerOwner) { s0 = end - kMirrorPerOwner; cnt = kMirrorPerOwner; }
    bool in_flight = false;
    for (int t = 0; t < kFrontSlots; ++t) in_flight |= f->pass[t].busy;
    bool ok = true;
    for (int c = 0; c < f->G && ok; ++c) {
        if (c == o) continue;
        // a pass in flight may still read the rows that are about to be replaced (its exact pass runs on a side stream)
        if (in_flight && eng_sync_streams(f->sh[c])) { ok = false; break; }
        for (int s = s0; s < end && ok; ) {
            const int r = s % kMirrorPerOwner;
            const int len = end - s < kMirrorPerOwner - r ? end - s : kMirrorPerOwner - r;
            ok = eng_stage_from_peer(f->sh[c], mirror_row(o, s), f->sh[o], s, len) == SCL_OK;
            s += len;
        }
    }
    // (mirror_lo[o] <= the owner's old size always: a successful copy keeps [mirror_lo, end) contiguous)
    if (!ok) f->mirror_lo[o] = end;                        // nothing of this owner up to here counts as mirrored
}

// make global keyframe / staged query `query` available on every shard; qid[c] = the id shard c uses for it.
// stage_slot: which staging slot the non-owners use (1 + front pass for passes in flight, 1 for blocking calls).
int place_query(scl_engine *e, int query, int stage_slot, int *qid)
{
    ShardedFront *f = e->front;
    if (query < 0) {
        if (query != SCL_QUERY_STAGED || !f->staged0) return ffail(e, SCL_ERR_INVALID_ARG, "no staged query (call scl_stage_query first)");
        for (int c = 0; c < f->G; ++c) qid[c] = SCL_QUERY_STAGED;
        return SCL_OK;
    }
    if (query >= f->n) return ffail(e, SCL_ERR_OUT_OF_RANGE, "query keyframe out of range");
    const int owner = query % f->G, slot = query / f->G;
    const bool in_mirror = mirrored(f, query);
    for (int c = 0; c < f->G; ++c) {
        if (c == owner) { qid[c] = slot; continue; }
        if (in_mirror) { qid[c] = -1 - mirror_row(owner, slot); continue; }
        const int rc = eng_stage_from_peer(f->sh[c], stage_slot, f->sh[owner], slot);
        if (rc) return child_fail(e, f->sh[c], rc, "stage query on shard");
        qid[c] = -1 - stage_slot;
    }
    return SCL_OK;
}

struct TopRec { float d2; int g; double dist; int shift; };

// global ring-key top-k (+ SC distance of those) over global range [lo, hi): per-shard top-k, merged
int sharded_topk(scl_engine *e, int query, int lo, int hi, int k, float eps, bool want_dist, std::vector<TopRec> *out)
{
    ShardedFront *f = e->front;
    if (k <= 0 || k > kTopkMaxK) return ffail(e, SCL_ERR_INVALID_ARG, "k out of range (1..64)");
    if (lo < 0) lo = 0;
    if (hi > f->n) hi = f->n;
    int qid[kMaxShards];
    int rc = place_query(e, query, 1 + kFrontSlots, qid);   // the staging slot of blocking calls
    if (rc) return rc;
    bool have[kMaxShards] = {false};
    for (int c = 0; c < f->G; ++c) {                       // enqueue on every device, then wait
        rc = eng_topk_enqueue(f->sh[c], qid[c], local_count(lo, c, f->G), local_count(hi, c, f->G), k, eps, want_dist, &have[c]);
        if (rc) return child_fail(e, f->sh[c], rc, "top-k on shard");
    }
    out->clear();
    for (int c = 0; c < f->G; ++c) {
        int idx[kTopkMaxK]; float d2[kTopkMaxK]; double dist[kTopkMaxK]; int shift[kTopkMaxK];
        rc = eng_topk_finish(f->sh[c], k, have[c], idx, d2, dist, shift, nullptr);
        if (rc) return child_fail(e, f->sh[c], rc, "top-k on shard");
        for (int i = 0; i < k; ++i)
            if (idx[i] >= 0) out->push_back({d2[i], idx[i] * f->G + c, dist[i], shift[i]});
    }
    // ascending ring distance, ties -> lowest global index: the order one database's scan returns
    std::sort(out->begin(), out->end(), [](const TopRec &a, const TopRec &b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.g < b.g); });
    if ((int)out->size() > k) out->resize((size_t)k);
    return SCL_OK;
}

}  // namespace

// ---- life cycle -----------------------------------------------------------------------------------------------------

scl_engine *front_primary(const scl_engine *e) { return e->front->sh[0]; }

int front_destroy(scl_engine *e)
{
    ShardedFront *f = e->front;
    for (int c = 0; c < f->G; ++c) {
        if (f->sh[c]) { (void)hipSetDevice(f->dev[c]); (void)eng_sync_streams(f->sh[c]); }
    }
#ifdef SCL_HAVE_RCCL_HEADER
    for (int c = 0; c < f->G; ++c) if (f->comm[c] && f->coll) (void)f->coll->CommDestroy(f->comm[c]);
#endif
    for (int c = 0; c < f->G; ++c) {
        if (f->d_key[c]) { (void)hipSetDevice(f->dev[c]); (void)hipFree(f->d_key[c]); }
        if (f->sh[c]) (void)scl_destroy(f->sh[c]);
    }
    if (f->h_keys) (void)hipHostFree(f->h_keys);
    for (auto &ev : f->ev_group) if (ev) (void)hipEventDestroy(ev);
    delete f;
    delete e;
    return SCL_OK;
}

}  // namespace scl

using namespace scl;

extern "C" int scl_create_sharded(const scl_config *cfg, const int *devices, int n_devices, int exchange, scl_engine **out)
{
    if (!cfg || !devices || !out || n_devices < 1 || n_devices > kMaxShards || exchange < 0 || exchange > 2) return SCL_ERR_INVALID_ARG;
    *out = nullptr;
    scl_engine *e = new (std::nothrow) scl_engine();
    ShardedFront *f = new (std::nothrow) ShardedFront();
    if (!e || !f) { delete e; delete f; return SCL_ERR_NOMEM; }
    e->front = f;
    e->cfg = *cfg;
    e->R = cfg->num_ring; e->S = cfg->num_sector;
    f->G = n_devices;
    bool distinct = true;
    for (int c = 0; c < n_devices; ++c) {
        f->dev[c] = devices[c];
        for (int d = 0; d < c; ++d) distinct &= devices[d] != devices[c];
    }
    for (int c = 0; c < n_devices; ++c) {
        scl_config cc = *cfg;
        cc.device = devices[c];
        cc.initial_capacity = cfg->initial_capacity > 0 ? (cfg->initial_capacity + n_devices - 1) / n_devices + 1 : cfg->initial_capacity;
        int rc = scl_create(&cc, &f->sh[c]);
        if (rc) { front_destroy(e); return rc; }
        if (n_devices > 1 && (rc = eng_set_stage_rows(f->sh[c], scl_engine::kStage + n_devices * kMirrorPerOwner))) { front_destroy(e); return rc; }
    }
    f->mirror_on = n_devices > 1;
    // the exchange of full-DB winners: 0 = RCCL when it can be had (more than one shard, every shard on its own
    // device), else the host merge; 1 = host merge; 2 = RCCL or fail
    f->exchange = 1;
    if (exchange >= 2 || (exchange == 0 && n_devices > 1 && distinct)) {
        int rc = SCL_ERR_UNSUPPORTED;
#ifdef SCL_HAVE_RCCL_HEADER
        Rccl *r = rccl();
        if (r->ok && (distinct || r->shared_devices_ok)) {
            const ncclResult_t nr = r->CommInitAll(f->comm, n_devices, f->dev);
            rc = nr == ncclSuccess ? SCL_OK : SCL_ERR_HIP;
            if (rc) for (auto &cm : f->comm) cm = nullptr;
            else f->coll = r;
        }
#endif
        if (rc == SCL_OK) {
            const size_t words = (size_t)kFrontSlots * 4 * kMaxQueryBatch;
            for (int c = 0; c < n_devices && rc == SCL_OK; ++c) {
                if (hipSetDevice(f->dev[c]) != hipSuccess || hipMalloc((void **)&f->d_key[c], words * 8) != hipSuccess) rc = SCL_ERR_HIP;
            }
            if (rc == SCL_OK && hipHostMalloc((void **)&f->h_keys, (size_t)kFrontSlots * 2 * kMaxQueryBatch * 8, hipHostMallocDefault) != hipSuccess) rc = SCL_ERR_HIP;
            (void)hipSetDevice(f->dev[0]);
            for (int g = 0; g < kFrontSlots && rc == SCL_OK; ++g)
                if (hipEventCreateWithFlags(&f->ev_group[g], hipEventDisableTiming) != hipSuccess) rc = SCL_ERR_HIP;
        }
        if (rc == SCL_OK) f->exchange = 2;
        else if (exchange >= 2) { front_destroy(e); return rc; }
    }
    *out = e;
    return SCL_OK;
}

extern "C" int scl_shard_info(const scl_engine *e, int *n_shards, int *exchange)
{
    if (!e) return SCL_ERR_INVALID_ARG;
    if (n_shards) *n_shards = e->front ? e->front->G : 1;
    if (exchange) *exchange = e->front ? e->front->exchange : 0;
    return SCL_OK;
}

namespace scl {

// ---- ingest -----------------------------------------------------------------------------------------------------------

namespace {
// the database arrays of a shard are about to move: nothing (peer copies included) may still be reading them
int quiesce_if_regrow(scl_engine *e, int shard, int count)
{
    ShardedFront *f = e->front;
    if (!eng_would_regrow(f->sh[shard], count)) return SCL_OK;
    for (int c = 0; c < f->G; ++c) {
        const int rc = eng_sync_streams(f->sh[c]);
        if (rc) return child_fail(e, f->sh[c], rc, "sync before regrow");
    }
    return SCL_OK;
}
}  // namespace

int front_make_and_save(scl_engine *e, const void *points, int n_points, int stride_bytes, int8_t robot, int index, float *out_values,
                        bool filtered, float leaf, int *n_filtered)
{
    std::lock_guard<std::mutex> lk(e->mu);
    ShardedFront *f = e->front;
    const int c = f->n % f->G;                             // the new keyframe's shard builds its descriptor
    int rc = quiesce_if_regrow(e, c, 1);
    if (rc) return rc;
    rc = filtered ? scl_make_and_save_filtered(f->sh[c], points, n_points, stride_bytes, leaf, robot, index, out_values, n_filtered)
                  : scl_make_and_save(f->sh[c], points, n_points, stride_bytes, robot, index, out_values);
    if (rc) return child_fail(e, f->sh[c], rc, "make_and_save on shard");
    f->robots.push_back(robot); f->indexs.push_back(index); f->n++;
    mirror_appended(e, c, (f->n - 1) / f->G, 1);
    return SCL_OK;
}

int front_save_bulk(scl_engine *e, const float *values, int count, const int8_t *robots, const int *indexs)
{
    std::lock_guard<std::mutex> lk(e->mu);
    ShardedFront *f = e->front;
    const size_t cells = (size_t)e->R * e->S;
    if (count == 1) {
        const int c = f->n % f->G;
        int rc = quiesce_if_regrow(e, c, 1);
        if (rc) return rc;
        rc = scl_save_bulk(f->sh[c], values, 1, robots, indexs);
        if (rc) return child_fail(e, f->sh[c], rc, "save on shard");
    } else {
        std::vector<float> buf;
        std::vector<int8_t> rb; std::vector<int> ib;
        // A failure on shard c must not leave shards 0 .. c-1 longer than the front believes they are (global g lives in slot
        // g / G of shard g % G): the shards that already appended are cut back to their old length
        int n_old[kMaxShards];
        for (int c = 0; c < f->G; ++c) n_old[c] = scl_get_size(f->sh[c], -1);
        auto undo = [&](int upto) { for (int d = 0; d < upto; ++d) (void)eng_truncate(f->sh[d], n_old[d]); };
        for (int c = 0; c < f->G; ++c) {
            // descriptors i with (n + i) % G == c, in order
            int first = (c - f->n % f->G + f->G) % f->G;
            const int m = first < count ? (count - first + f->G - 1) / f->G : 0;
            if (m == 0) continue;
            buf.resize((size_t)m * cells); rb.resize((size_t)m); ib.resize((size_t)m);
            for (int j = 0; j < m; ++j) {
                const int i = first + j * f->G;
                std::memcpy(buf.data() + (size_t)j * cells, values + (size_t)i * cells, cells * sizeof(float));
                rb[(size_t)j] = robots ? robots[i] : (int8_t)0;
                ib[(size_t)j] = indexs ? indexs[i] : f->n + i;
            }
            int rc = quiesce_if_regrow(e, c, m);
            if (rc) { undo(c); return rc; }
            rc = scl_save_bulk(f->sh[c], buf.data(), m, rb.data(), ib.data());
            if (rc) { undo(c); return child_fail(e, f->sh[c], rc, "save on shard"); }
        }
    }
    for (int i = 0; i < count; ++i) {
        f->robots.push_back(robots ? robots[i] : (int8_t)0);
        f->indexs.push_back(indexs ? indexs[i] : f->n + i);
    }
    const int n_before = f->n;
    f->n += count;
    for (int c = 0; c < f->G; ++c) {
        const int s0 = local_count(n_before, c, f->G), s1 = local_count(f->n, c, f->G);
        mirror_appended(e, c, s0, s1 - s0);
    }
    return SCL_OK;
}

int front_stage_query(scl_engine *e, const float *values)
{
    std::lock_guard<std::mutex> lk(e->mu);
    ShardedFront *f = e->front;
    for (int c = 0; c < f->G; ++c) {
        const int rc = eng_stage_values(f->sh[c], 0, values);
        if (rc) return child_fail(e, f->sh[c], rc, "stage query on shard");
    }
    f->staged0 = true;
    return SCL_OK;
}

int front_get_index(const scl_engine *e, int key, int8_t *robot, int *index)
{
    std::lock_guard<std::mutex> lk(e->mu);
    const ShardedFront *f = e->front;
    if (key < 0 || key >= f->n) return ffail(e, SCL_ERR_OUT_OF_RANGE, "key out of range");
    *robot = f->robots[(size_t)key]; *index = f->indexs[(size_t)key];
    return SCL_OK;
}

int front_get_size(const scl_engine *e)
{
    std::lock_guard<std::mutex> lk(e->mu);
    return e->front->n;
}

int front_get_slot(const scl_engine *e, int key, scl_engine **child, int *slot)
{
    std::lock_guard<std::mutex> lk(e->mu);
    const ShardedFront *f = e->front;
    if (key < 0 || key >= f->n) return ffail(e, SCL_ERR_OUT_OF_RANGE, "key out of range");
    *child = f->sh[key % f->G]; *slot = key / f->G;
    return SCL_OK;
}

// ---- reference-faithful detection ---------------------------------------------------------------------------------------

int front_detect_intra(scl_engine *e, int cur, int *loop_id, float *shift, double *dist)
{
    std::lock_guard<std::mutex> lk(e->mu);
    ShardedFront *f = e->front;
    *loop_id = -1; *shift = 0.0f;                                         /* D.h:1615 */
    if (dist) *dist = kBigDist;
    if (cur < 0 || cur >= f->n) return ffail(e, SCL_ERR_OUT_OF_RANGE, "cur out of range");
    const int k = e->cfg.num_candidates;
    if (cur < e->cfg.num_exclude_recent + k + 1) return SCL_OK;           /* D.h:1620-1623 */
    const int history = cur - e->cfg.num_exclude_recent;                  /* D.h:1627, on global indices */
    std::vector<TopRec> top;
    const int rc = sharded_topk(e, cur, 0, history, k, e->cfg.knn_exclude_eps, true, &top);
    if (rc) return rc;
    float minDis = 10000000.0f;                                           /* D.h:1637: a float */
    int minIndex = -1, minBias = 0;
    for (const TopRec &t : top) {                                         /* D.h:1645-1659 */
        if (t.dist < (double)minDis) {                                    /* D.h:1653 */
            minDis = (float)t.dist;                                       /* D.h:1655 narrowing */
            minIndex = t.g; minBias = t.shift;
        }
    }
    if (dist) *dist = (double)minDis;
    if ((double)minDis < e->cfg.dist_thres) {                             /* D.h:1662 */
        *loop_id = minIndex;
        *shift = (float)minBias;                                          /* D.h:1665 */
    }
    return SCL_OK;
}

int front_detect_inter(scl_engine *e, int cur, int *loop_id, float *yaw_rad, double *dist)
{
    std::lock_guard<std::mutex> lk(e->mu);
    ShardedFront *f = e->front;
    *loop_id = -1; *yaw_rad = 0.0f;                                       /* D.h:1678,1686 */
    if (dist) *dist = kBigDist;
    if (cur < 0 || cur >= f->n) return ffail(e, SCL_ERR_OUT_OF_RANGE, "cur out of range");
    if (f->n < e->cfg.num_exclude_recent + 1) return SCL_OK;              /* D.h:1684-1688 */
    if (f->tree_counter % e->cfg.tree_making_period == 0)                 /* D.h:1691-1702 */
        f->tree_n = f->n - e->cfg.num_exclude_recent;
    f->tree_counter = f->tree_counter + 1;                                /* D.h:1703 */
    const int k = e->cfg.num_candidates;
    std::vector<TopRec> top;
    int rc = sharded_topk(e, cur, 0, f->tree_n, k, 0.0f, true, &top);
    if (rc) return rc;
    // slots the search left unfilled read as index 0 in the reference (zero-initialised candidate_indexes,
    // D.h:1710): score keyframe 0 for them
    double cd0 = kBigDist; int ca0 = 0;
    if ((int)top.size() < k && f->tree_n > 0) {
        int qid[kMaxShards];
        if ((rc = place_query(e, cur, 1 + kFrontSlots, qid))) return rc;
        const int zero = 0;
        rc = scl_sc_distance_batch(f->sh[0], qid[0], &zero, 1, &cd0, &ca0);
        if (rc) return child_fail(e, f->sh[0], rc, "distance to keyframe 0");
    }
    double min_dist = 10000000;                                           /* D.h:1705 */
    int nn_align = 0, nn_idx = -1;
    for (int i = 0; i < k; ++i) {                                         /* D.h:1721-1737 */
        const bool filled = i < (int)top.size();
        const int ci = filled ? top[(size_t)i].g : 0;
        const double c = filled ? top[(size_t)i].dist : cd0;
        const int al = filled ? top[(size_t)i].shift : ca0;
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

int front_topk(scl_engine *e, int query, int lo, int hi, int k, int *idx, float *d2, double *dist, int *shift, int *found)
{
    std::lock_guard<std::mutex> lk(e->mu);
    std::vector<TopRec> top;
    const int rc = sharded_topk(e, query, lo, hi, k, e->cfg.knn_exclude_eps, dist || shift, &top);
    if (rc) return rc;
    for (int i = 0; i < k; ++i) {
        const bool filled = i < (int)top.size();
        idx[i] = filled ? top[(size_t)i].g : -1;
        d2[i] = filled ? top[(size_t)i].d2 : FLT_MAX;
        if (dist) dist[i] = filled ? top[(size_t)i].dist : kBigDist;
        if (shift) shift[i] = filled ? top[(size_t)i].shift : 0;
    }
    if (found) *found = (int)top.size();
    return SCL_OK;
}

int front_sc_distance_batch(scl_engine *e, int query, const int *cand, int n, double *dist, int *shift)
{
    std::lock_guard<std::mutex> lk(e->mu);
    ShardedFront *f = e->front;
    if (!cand && n > f->n) return ffail(e, SCL_ERR_OUT_OF_RANGE, "n exceeds database size");
    int qid[kMaxShards];
    int rc = place_query(e, query, 1 + kFrontSlots, qid);
    if (rc) return rc;
    std::vector<int> local, where;
    std::vector<double> dl; std::vector<int> sl;
    for (int c = 0; c < f->G; ++c) {
        local.clear(); where.clear();
        for (int i = 0; i < n; ++i) {
            const int g = cand ? cand[i] : i;
            if (g >= f->n) return ffail(e, SCL_ERR_OUT_OF_RANGE, "candidate keyframe out of range");
            if (g < 0) { if (c == 0) { dist[i] = kBigDist; shift[i] = 0; } continue; }
            if (g % f->G == c) { local.push_back(g / f->G); where.push_back(i); }
        }
        if (local.empty()) continue;
        dl.resize(local.size()); sl.resize(local.size());
        rc = scl_sc_distance_batch(f->sh[c], qid[c], local.data(), (int)local.size(), dl.data(), sl.data());
        if (rc) return child_fail(e, f->sh[c], rc, "distance batch on shard");
        for (size_t j = 0; j < local.size(); ++j) { dist[where[j]] = dl[j]; shift[where[j]] = sl[j]; }
    }
    return SCL_OK;
}

int front_sc_distance_matrix(scl_engine *e, const int *queries, int nq, int lo, int hi, double *dist, int *shift)
{   // row by row through the per-shard candidate lists (a diagnostic / bulk-export call on a sharded database)
    int n_all;
    { std::lock_guard<std::mutex> lk(e->mu); n_all = e->front->n; }
    if (lo < 0 || hi > n_all || hi < lo) return ffail(e, SCL_ERR_OUT_OF_RANGE, "keyframe range out of the database");
    const int n = hi - lo;
    if (n == 0) return SCL_OK;
    std::vector<int> cand((size_t)n);
    for (int i = 0; i < n; ++i) cand[(size_t)i] = lo + i;
    for (int r = 0; r < nq; ++r) {
        const int rc = front_sc_distance_batch(e, queries[r], cand.data(), n, dist + (size_t)r * n, shift + (size_t)r * n);
        if (rc) return rc;
    }
    return SCL_OK;
}

// ---- full-DB detection --------------------------------------------------------------------------------------------------

namespace {

// the RCCL reduction of one launch group (m <= kMaxQueryBatch passes submitted together on every shard)
int enqueue_group_exchange(scl_engine *e, const int *front_tickets, int m)
{
#ifdef SCL_HAVE_RCCL_HEADER
    ShardedFront *f = e->front;
    Rccl *r = f->coll;
    const int g = (int)(f->next_group % kFrontSlots);
    if (f->group[g].active) return ffail(e, SCL_ERR_INVALID_ARG, "too many full-DB passes in flight: collect first");
    const size_t goff = (size_t)g * 4 * kMaxQueryBatch;
    for (int c = 0; c < f->G; ++c) {
        PackArgs a{};
        a.m = m; a.shard = c; a.G = f->G;
        for (int i = 0; i < m; ++i) {
            bool empty = true; int lo = 0;
            const int tk = f->pass[front_tickets[i]].tk[c];
            const double *rec = tk >= 0 ? eng_ticket_record(f->sh[c], tk, &lo, &empty) : nullptr;
            a.rec[i] = rec; a.slot_lo[i] = lo; a.empty[i] = (!rec || empty) ? 1 : 0;
        }
        if (hipSetDevice(f->dev[c]) != hipSuccess) return ffail(e, SCL_ERR_HIP, "hipSetDevice");
        unsigned long long *k = f->d_key[c] + goff;
        hipLaunchKernelGGL(pack_winner_keys_kernel, dim3(1), dim3(64), 0, eng_stream(f->sh[c]), a, k, k + kMaxQueryBatch);
        if (hipGetLastError() != hipSuccess) return ffail(e, SCL_ERR_HIP, "pack_winner_keys_kernel launch");
    }
    auto all_reduce = [&](int in_word, int out_word) -> int {
        if (r->GroupStart() != ncclSuccess) return SCL_ERR_HIP;
        for (int c = 0; c < f->G; ++c) {
            (void)hipSetDevice(f->dev[c]);
            unsigned long long *k = f->d_key[c] + goff;
            if (r->AllReduce(k + in_word * kMaxQueryBatch, k + out_word * kMaxQueryBatch, (size_t)m, ncclUint64, ncclMin,
                             f->comm[c], eng_stream(f->sh[c])) != ncclSuccess) { (void)r->GroupEnd(); return SCL_ERR_HIP; }
        }
        return r->GroupEnd() == ncclSuccess ? SCL_OK : SCL_ERR_HIP;
    };
    if (all_reduce(0, 2)) return ffail(e, SCL_ERR_HIP, "ncclAllReduce(min) on the distance keys failed");
    for (int c = 0; c < f->G; ++c) {
        (void)hipSetDevice(f->dev[c]);
        unsigned long long *k = f->d_key[c] + goff;
        hipLaunchKernelGGL(select_winner_keys_kernel, dim3(1), dim3(64), 0, eng_stream(f->sh[c]), k, k + 2 * kMaxQueryBatch, k + kMaxQueryBatch, m);
        if (hipGetLastError() != hipSuccess) return ffail(e, SCL_ERR_HIP, "select_winner_keys_kernel launch");
    }
    if (all_reduce(1, 3)) return ffail(e, SCL_ERR_HIP, "ncclAllReduce(min) on the index keys failed");
    (void)hipSetDevice(f->dev[0]);
    unsigned long long *k0 = f->d_key[0] + goff;
    unsigned long long *h = f->h_keys + (size_t)g * 2 * kMaxQueryBatch;
    if (hipMemcpyAsync(h, k0 + 2 * kMaxQueryBatch, sizeof(unsigned long long) * 2 * kMaxQueryBatch, hipMemcpyDeviceToHost, eng_stream(f->sh[0])) != hipSuccess ||
        hipEventRecord(f->ev_group[g], eng_stream(f->sh[0])) != hipSuccess)
        return ffail(e, SCL_ERR_HIP, "result copy of the exchange");
    ShardedFront::Group &gr = f->group[g];
    gr.active = true; gr.m = m; gr.delivered = false; gr.first_ticket = front_tickets[0];
    for (int i = 0; i < m; ++i) f->pass[front_tickets[i]].group = g * kMaxQueryBatch + i;
    f->next_group++;
    return SCL_OK;
#else
    (void)front_tickets; (void)m;
    return ffail(e, SCL_ERR_UNSUPPORTED, "built without the RCCL header");
#endif
}

int submit_many_locked(scl_engine *e, const int *queries, const int *lo, const int *hi, int nq, int *tickets)
{
    ShardedFront *f = e->front;
    if (nq < 1 || nq > kMaxQueryBatch) return ffail(e, SCL_ERR_INVALID_ARG, "1..4 queries per launch");
    for (int i = 0; i < nq; ++i)
        if (f->pass[(f->next_pass + (unsigned)i) % kFrontSlots].busy)
            return ffail(e, SCL_ERR_INVALID_ARG, "too many full-DB passes in flight: collect first");
    int qid[kMaxQueryBatch][kMaxShards];
    for (int i = 0; i < nq; ++i) {
        const int t = (int)((f->next_pass + (unsigned)i) % kFrontSlots);
        const int rc = place_query(e, queries[i], 1 + t, qid[i]);         // staging slot 1 + t is free while pass t is
        if (rc) return rc;
    }
    int child_tk[kMaxShards][kMaxQueryBatch];
    for (int c = 0; c < f->G; ++c) {
        int q[kMaxQueryBatch], l[kMaxQueryBatch], h[kMaxQueryBatch];
        for (int i = 0; i < nq; ++i) {
            const int glo = lo[i] < 0 ? 0 : lo[i], ghi = hi[i] > f->n ? f->n : hi[i];
            q[i] = qid[i][c]; l[i] = local_count(glo, c, f->G); h[i] = local_count(ghi, c, f->G);
        }
        const int rc = scl_detect_full_submit_many(f->sh[c], q, l, h, nq, child_tk[c]);
        if (rc) {
            // unwind: passes already enqueued on the shards before c are waited for and dropped
            for (int d = 0; d < c; ++d)
                for (int i = 0; i < nq; ++i) { int a, b; double x; (void)scl_detect_full_collect(f->sh[d], child_tk[d][i], &a, &b, &x); }
            return child_fail(e, f->sh[c], rc, "submit on shard");
        }
    }
    for (int i = 0; i < nq; ++i) {
        const int t = (int)((f->next_pass + (unsigned)i) % kFrontSlots);
        ShardedFront::Pass &p = f->pass[t];
        p.busy = true; p.group = -1;
        for (int c = 0; c < f->G; ++c) p.tk[c] = child_tk[c][i];
        tickets[i] = t;
    }
    f->next_pass += (unsigned)nq;
    f->last_pass = tickets[nq - 1];
    if (f->exchange >= 2) {
        const int rc = enqueue_group_exchange(e, tickets, nq);
        if (rc) {
            // the caller gets no tickets: wait for the passes the shards hold and give their slots back
            const std::string why = e->last_error;
            for (int i = 0; i < nq; ++i) {
                ShardedFront::Pass &p = f->pass[tickets[i]];
                for (int c = 0; c < f->G; ++c) { int a, b; double x; if (p.tk[c] >= 0) (void)scl_detect_full_collect(f->sh[c], p.tk[c], &a, &b, &x); }
                p.busy = false; p.group = -1;
            }
            e->last_error = why;
            return rc;
        }
    }
    return SCL_OK;
}

int collect_locked(scl_engine *e, int ticket, int *nn_idx, int *shift, double *dist)
{
    ShardedFront *f = e->front;
    if (ticket < 0 || ticket >= kFrontSlots || !f->pass[ticket].busy) return ffail(e, SCL_ERR_INVALID_ARG, "unknown ticket");
    ShardedFront::Pass &p = f->pass[ticket];
    *nn_idx = -1; *shift = 0; *dist = kBigDist;
    if (p.group >= 0) {
        // device-side exchange: the reduced keys of the whole launch group arrive together on shard 0's stream
        const int g = p.group / kMaxQueryBatch, i = p.group % kMaxQueryBatch;
        ShardedFront::Group &gr = f->group[g];
        if (!gr.delivered) {
            (void)hipSetDevice(f->dev[0]);
            if (hipEventSynchronize(f->ev_group[g]) != hipSuccess) return ffail(e, SCL_ERR_HIP, "hipEventSynchronize(exchange)");
            const unsigned long long *h = f->h_keys + (size_t)g * 2 * kMaxQueryBatch;
            for (int j = 0; j < gr.m; ++j) {
                const unsigned long long m1 = h[j], m2 = h[kMaxQueryBatch + j];
                if (m1 == kNoKey || m2 == kNoKey) { gr.dist[j] = kBigDist; gr.idx[j] = -1; gr.shift[j] = 0; }
                else { gr.dist[j] = key_to_dist(m1); gr.idx[j] = (int)(m2 >> 16); gr.shift[j] = (int)(m2 & 0xffffull); }
            }
            gr.delivered = true;
        }
        *dist = gr.dist[i]; *nn_idx = gr.idx[i]; *shift = gr.shift[i];
        for (int c = 0; c < f->G; ++c) (void)eng_release_ticket(f->sh[c], p.tk[c]);
        p.busy = false; p.group = -1;
        bool any = false;
        for (int t = 0; t < kFrontSlots; ++t) any |= f->pass[t].busy && f->pass[t].group / kMaxQueryBatch == g && f->pass[t].group >= 0;
        if (!any) gr.active = false;
        return SCL_OK;
    }
    // host merge: strict <, ties -> lowest global index (one database's arg-min)
    int rc_first = SCL_OK;
    for (int c = 0; c < f->G; ++c) {
        int nn = -1, sh = 0; double d = kBigDist;
        const int rc = scl_detect_full_collect(f->sh[c], p.tk[c], &nn, &sh, &d);
        if (rc) { if (!rc_first) rc_first = child_fail(e, f->sh[c], rc, "collect on shard"); continue; }
        if (nn < 0) continue;
        const int g = nn * f->G + c;
        if (d < *dist || (d == *dist && (*nn_idx < 0 || g < *nn_idx))) { *dist = d; *nn_idx = g; *shift = sh; }
    }
    p.busy = false;
    return rc_first;
}

}  // namespace

int front_submit_many(scl_engine *e, const int *queries, const int *lo, const int *hi, int nq, int *tickets)
{
    std::lock_guard<std::mutex> lk(e->mu);
    return submit_many_locked(e, queries, lo, hi, nq, tickets);
}

int front_collect(scl_engine *e, int ticket, int *nn_idx, int *shift, double *dist)
{
    std::lock_guard<std::mutex> lk(e->mu);
    return collect_locked(e, ticket, nn_idx, shift, dist);
}

int front_detect_full_stream(scl_engine *e, const int *queries, const int *lo, const int *hi, int n_queries,
                             int scans_per_launch, int launches_in_flight, int *nn_idx, int *shift, double *dist)
{
    std::lock_guard<std::mutex> lk(e->mu);
    ShardedFront *f = e->front;
    for (int t = 0; t < kFrontSlots; ++t)
        if (f->pass[t].busy) return ffail(e, SCL_ERR_INVALID_ARG, "detect_full_stream: collect the passes in flight first");
    // Block form (winners merged on the host whatever the exchange mode of the single passes: the results of a stream go to host arrays
    // anyway): the scans go to the shards in blocks of up to kStreamCall.  A scan whose keyframe the other shards hold in their mirror
    // rows (every recent keyframe: copied when it was appended) needs nothing; the others -- at most 64 per block -- are copied to the
    // staging rows of the shards that do not hold them.  Then every shard runs its own stream form over the block (up to 16 scans per
    // launch, one exact pass per chunk: scl_detect_full_stream of the shard, one host thread per shard), and the per-shard winners of
    // the block are merged as one database's arg-min would (smallest distance, ties to the lowest global index).
    const int G = f->G;
    const int spl_s = scans_per_launch < 1 ? 1 : scans_per_launch;              // the shard clamps it to what its grid takes per launch
    std::vector<int> q((size_t)G * kStreamCall), l((size_t)G * kStreamCall), h((size_t)G * kStreamCall);
    std::vector<int> nn((size_t)G * kStreamCall), sh((size_t)G * kStreamCall);
    std::vector<double> dd((size_t)G * kStreamCall);
    std::vector<int> row_of((size_t)kStreamCall);
    for (int b0 = 0, m = 0; b0 < n_queries; b0 += m) {
        int n_copy = 0;
        for (m = 0; b0 + m < n_queries && m < kStreamCall; ++m) {
            const int g = queries[b0 + m];
            const bool copy = G > 1 && g >= 0 && !mirrored(f, g);
            if (copy && n_copy == kStreamBlock) break;
            n_copy += copy ? 1 : 0;
            row_of[(size_t)m] = -1;
        }
        // Staging rows in the order (owner shard, position in the block): keyframes that follow each other on their owner land in
        // consecutive rows, and a run of them travels to a shard in one copy per array instead of one per keyframe.
        int next_row = kStreamStage0;
        for (int o = 0; o < G && n_copy > 0; ++o) {
            int run_row = -1, run_slot = -1, run_len = 0;
            auto flush = [&]() -> int {
                for (int c = 0; c < G && run_len > 0; ++c) {
                    if (c == o) continue;
                    const int rc = eng_stage_from_peer(f->sh[c], run_row, f->sh[o], run_slot, run_len);
                    if (rc) return child_fail(e, f->sh[c], rc, "stage queries on shard");
                }
                run_len = 0;
                return SCL_OK;
            };
            for (int i = 0; i < m; ++i) {
                const int g = queries[b0 + i];
                if (g < 0 || g % G != o || mirrored(f, g)) continue;
                if (g >= f->n) return ffail(e, SCL_ERR_OUT_OF_RANGE, "query keyframe out of range");
                const int slot = g / G;
                row_of[(size_t)i] = next_row++;
                if (run_len > 0 && slot == run_slot + run_len) { ++run_len; continue; }
                int rc = flush();
                if (rc) return rc;
                run_row = row_of[(size_t)i]; run_slot = slot; run_len = 1;
            }
            const int rc = flush();
            if (rc) return rc;
        }
        for (int i = 0; i < m; ++i) {
            const int g = queries[b0 + i];
            if (g < 0 && (g != SCL_QUERY_STAGED || !f->staged0)) return ffail(e, SCL_ERR_INVALID_ARG, "no staged query (call scl_stage_query first)");
            const int glo = lo[b0 + i] < 0 ? 0 : lo[b0 + i], ghi = hi[b0 + i] > f->n ? f->n : hi[b0 + i];
            for (int c = 0; c < G; ++c) {
                const size_t k = (size_t)c * kStreamCall + (size_t)i;
                q[k] = g < 0 ? SCL_QUERY_STAGED : (c == g % G ? g / G : -1 - (row_of[(size_t)i] >= 0 ? row_of[(size_t)i] : mirror_row(g % G, g / G)));
                l[k] = local_count(glo, c, G); h[k] = local_count(ghi, c, G);
            }
        }
        int rcs[kMaxShards] = {0};
        auto run = [&](int c) {
            const size_t k = (size_t)c * kStreamCall;
            rcs[c] = scl_detect_full_stream(f->sh[c], q.data() + k, l.data() + k, h.data() + k, m, spl_s, launches_in_flight,
                                            nn.data() + k, sh.data() + k, dd.data() + k);
        };
        std::vector<std::thread> pool;
        for (int c = 1; c < G; ++c) pool.emplace_back(run, c);
        run(0);
        for (auto &t : pool) t.join();
        for (int c = 0; c < G; ++c) if (rcs[c]) return child_fail(e, f->sh[c], rcs[c], "stream on shard");
        for (int i = 0; i < m; ++i) {
            int bi = -1, bs = 0; double bd = kBigDist;
            for (int c = 0; c < G; ++c) {
                const size_t k = (size_t)c * kStreamCall + (size_t)i;
                if (nn[k] < 0) continue;
                const int g = nn[k] * G + c;
                if (dd[k] < bd || (dd[k] == bd && (bi < 0 || g < bi))) { bd = dd[k]; bi = g; bs = sh[k]; }
            }
            nn_idx[b0 + i] = bi; shift[b0 + i] = bs; dist[b0 + i] = bd;
        }
    }
    return SCL_OK;
}

int front_get_last_topk(scl_engine *e, int k, int *idx, float *d2)
{
    std::lock_guard<std::mutex> lk(e->mu);
    ShardedFront *f = e->front;
    std::vector<TopRec> all;
    for (int c = 0; c < f->G; ++c) {
        int li[kTopkMaxK]; float ld[kTopkMaxK];
        const int rc = scl_get_last_topk(f->sh[c], k, li, ld);
        if (rc) return child_fail(e, f->sh[c], rc, "last top-k of shard");
        for (int i = 0; i < k; ++i) if (li[i] >= 0) all.push_back({ld[i], li[i] * f->G + c, 0.0, 0});
    }
    std::sort(all.begin(), all.end(), [](const TopRec &a, const TopRec &b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.g < b.g); });
    for (int i = 0; i < k; ++i) {
        const bool filled = i < (int)all.size();
        idx[i] = filled ? all[(size_t)i].g : -1;
        d2[i] = filled ? all[(size_t)i].d2 : FLT_MAX;
    }
    return SCL_OK;
}

// ---- geometric verification: the loop candidates of one scan are independent -> candidate c runs on shard c % G ----

int front_icp_align_batch(scl_engine *e, const void *src, int n_src, const void *const *tgts, const int *n_tgts,
                          int n_targets, int stride_bytes, const scl_icp_params *p,
                          float *T, float *fitness, int *converged, int *iterations)
{
    std::lock_guard<std::mutex> lk(e->mu);
    ShardedFront *f = e->front;
    const int G = f->G;
    std::vector<std::thread> pool;
    std::atomic<int> first_rc{SCL_OK};
    auto run = [&](int c) {
        std::vector<const void *> tp; std::vector<int> tn; std::vector<int> which;
        for (int i = c; i < n_targets; i += G) { tp.push_back(tgts[i]); tn.push_back(n_tgts[i]); which.push_back(i); }
        if (which.empty()) return;
        const size_t m = which.size();
        std::vector<float> Tl(16 * m), fl(m); std::vector<int> cl(m), il(m);
        const int rc = scl_icp_align_batch(f->sh[c], src, n_src, tp.data(), tn.data(), (int)m, stride_bytes, p, Tl.data(), fl.data(), cl.data(), il.data());
        if (rc) { int ok = SCL_OK; if (first_rc.compare_exchange_strong(ok, rc)) e->last_error = std::string("icp_align_batch on shard: ") + scl_last_error(f->sh[c]); return; }
        for (size_t j = 0; j < m; ++j) {
            std::memcpy(T + 16 * (size_t)which[j], Tl.data() + 16 * j, 16 * sizeof(float));
            if (fitness) fitness[which[j]] = fl[j];
            if (converged) converged[which[j]] = cl[j];
            if (iterations) iterations[which[j]] = il[j];
        }
    };
    for (int c = 1; c < G; ++c) pool.emplace_back(run, c);
    run(0);
    for (auto &t : pool) t.join();
    return first_rc.load();
}

// ---- measurement -----------------------------------------------------------------------------------------------------------

int front_profile_enable(scl_engine *e, int on)
{
    for (int c = 0; c < e->front->G; ++c) { const int rc = scl_profile_enable(e->front->sh[c], on); if (rc) return rc; }
    return SCL_OK;
}

int front_profile_reset(scl_engine *e)
{
    for (int c = 0; c < e->front->G; ++c) { const int rc = scl_profile_reset(e->front->sh[c]); if (rc) return rc; }
    return SCL_OK;
}

int front_alignment_stats(scl_engine *e, uint64_t *pairs, uint64_t *fallbacks, int reset)
{
    uint64_t p = 0, f = 0;
    for (int c = 0; c < e->front->G; ++c) {
        uint64_t pc = 0, fc = 0;
        const int rc = scl_alignment_stats(e->front->sh[c], &pc, &fc, reset);
        if (rc) return rc;
        p += pc; f += fc;
    }
    if (pairs) *pairs = p;
    if (fallbacks) *fallbacks = f;
    return SCL_OK;
}

int front_survivor_stats(scl_engine *e, uint64_t *queries, uint64_t *survivors, uint64_t *max_survivors, int reset)
{   // a scan is scored on every shard: survivors add up over the shards of a scan, `queries` counts (scan, shard) passes
    uint64_t q = 0, s = 0, m = 0;
    for (int c = 0; c < e->front->G; ++c) {
        uint64_t qc = 0, sc = 0, mc = 0;
        const int rc = scl_survivor_stats(e->front->sh[c], &qc, &sc, &mc, reset);
        if (rc) return rc;
        q += qc; s += sc; m = mc > m ? mc : m;
    }
    if (queries) *queries = q;
    if (survivors) *survivors = s;
    if (max_survivors) *max_survivors = m;
    return SCL_OK;
}

int front_profile_get(scl_engine *e, scl_profile *out)
{
    std::memset(out, 0, sizeof *out);
    for (int c = 0; c < e->front->G; ++c) {
        scl_profile p;
        const int rc = scl_profile_get(e->front->sh[c], &p);
        if (rc) return rc;
        out->sc_distance_ms += p.sc_distance_ms; out->sc_distance_launches += p.sc_distance_launches; out->sc_distance_pairs += p.sc_distance_pairs;
        out->ringkey_topk_ms += p.ringkey_topk_ms; out->ringkey_topk_launches += p.ringkey_topk_launches;
        out->argmin_ms += p.argmin_ms; out->argmin_launches += p.argmin_launches;
        out->make_sc_ms += p.make_sc_ms; out->make_sc_launches += p.make_sc_launches; out->make_sc_points += p.make_sc_points;
        out->ingest_ms += p.ingest_ms; out->ingest_launches += p.ingest_launches;
        out->icp_nn_ms += p.icp_nn_ms; out->icp_nn_launches += p.icp_nn_launches;
        out->icp_reduce_ms += p.icp_reduce_ms; out->icp_reduce_launches += p.icp_reduce_launches;
    }
    return SCL_OK;
}

}  // namespace scl
