/*
 * scl_engine.h -- C ABI of the MI355X-native Scan Context loop-closure engine.
 *
 * This is the drop-in boundary for ONE path of thisparticle/scl_slam (ROS
 * package dlc_slam): Scan Context place recognition + geometric verification.
 * Plain pointers and sizes only; no C++/torch types; never throws.
 *
 * Each entry point names the reference interface it replaces
 *   D.h  = include/descriptor.h          (class scan_descriptor, D.h:21-36;
 *                                          scan_context_descriptor, D.h:1304-1801)
 *   DM.h = include/distributedMapping.h  (loop-closure driver)
 * A header-only C++ adapter with the reference's six virtuals lives in
 * include/scl/scan_context_hip_descriptor.hpp; INTEGRATION.md shows the
 * one-line change at DM.h:404.
 *
 * Conventions
 *   - every function returns an scl_status (0 = ok, < 0 = error);
 *   - "values" is the wire format of msg/global_descriptor.msg:8 `float32[]
 *     values`: R*S floats, ROW-major (ring-major), D.h:1446-1455 / 1576-1582;
 *   - point clouds are (base pointer, count, stride_bytes) with x,y,z as three
 *     consecutive floats at the start of each record (pcl::PointXYZI: stride 32);
 *   - "no loop" is loop_id == -1 exactly as D.h:1615,1678;
 *   - an engine is internally synchronised: append and detect may be called
 *     from different threads (the reference calls detect* without mtxSC,
 *     DM.h:1078,1280, while save() runs under it, DM.h:1001-1003).
 *   - all compute runs on the GPU; there is no CPU fallback.  Without a HIP
 *     device scl_create fails with SCL_ERR_NO_DEVICE.
 */
#ifndef SCL_ENGINE_H
#define SCL_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct scl_engine scl_engine;

typedef enum scl_status {
    SCL_OK               =  0,
    SCL_ERR_INVALID_ARG  = -1,
    SCL_ERR_NO_DEVICE    = -2,
    SCL_ERR_HIP          = -3,
    SCL_ERR_OUT_OF_RANGE = -4,
    SCL_ERR_NOMEM        = -5,
    SCL_ERR_UNSUPPORTED  = -6
} scl_status;

/* Constructor arguments of scan_context_descriptor, D.h:1307-1316 (same
 * defaults), plus engine-side knobs. */
typedef struct scl_config {
    int    num_ring;            /* PC_NUM_RING              = 20   */
    int    num_sector;          /* PC_NUM_SECTOR            = 60   */
    int    num_candidates;      /* NUM_CANDIDATES_FROM_TREE = 3    */
    double dist_thres;          /* SC_DIST_THRES            = 0.14 */
    double lidar_height;        /* LIDAR_HEIGHT             = 1.65 */
    double max_radius;          /* PC_MAX_RADIUS            = 80.0 */
    int    num_exclude_recent;  /* NUM_EXCLUDE_RECENT       = 100  */
    int    tree_making_period;  /* TREE_MAKING_PERIOD_      = 10   */
    double search_ratio;        /* SEARCH_RATIO             = 0.1  */
    float  knn_exclude_eps;     /* 0 = nanoflann semantics (D.h:1716); FLT_EPSILON =
                                   libnabo self-match exclusion (D.h:1642)          */
    int    device;              /* HIP device ordinal                     = 0    */
    int    initial_capacity;    /* keyframe slots preallocated (grows x2) = 4096 */
} scl_config;

/* Accumulated device time per kernel family, measured with HIP events on the
 * engine's own stream while profiling is enabled (scl_profile_enable). */
typedef struct scl_profile {
    double   sc_distance_ms;   uint64_t sc_distance_launches;  uint64_t sc_distance_pairs;
    double   ringkey_topk_ms;  uint64_t ringkey_topk_launches;
    double   argmin_ms;        uint64_t argmin_launches;
    double   make_sc_ms;       uint64_t make_sc_launches;      uint64_t make_sc_points;
    double   ingest_ms;        uint64_t ingest_launches;
    double   icp_nn_ms;        uint64_t icp_nn_launches;
    double   icp_reduce_ms;    uint64_t icp_reduce_launches;
} scl_profile;

const char *scl_status_string(int status);
const char *scl_last_error(const scl_engine *e);       /* detail of the last failure */
int  scl_abi_version(void);

int  scl_default_config(scl_config *cfg);
int  scl_create(const scl_config *cfg, scl_engine **out);     /* ctor, D.h:1307-1344 */
int  scl_destroy(scl_engine *e);
/* The same constructor for ONE keyframe database sharded over the GPUs of a node (BASELINE configs[3]): one
 * process, one engine state per entry of devices[], keyframe g on shard g % n_devices; every entry point of this
 * header works on the returned engine exactly as on a one-GPU engine (same results, bit for bit: the search
 * range of D.h:1627 is applied on global indices, ties go to the lowest global index).  Scoring needs no
 * exchange; the per-shard winners of the full-DB mode are reduced with two RCCL min all-reduces on packed 64-bit
 * keys (exchange = 2; needs every shard on its own device), or on the host from pinned memory (exchange = 1);
 * exchange = 0 picks RCCL when n_devices > 1 and the devices are distinct, else the host merge.  The same device
 * may be listed more than once (several shards on one GPU: rehearsal of the multi-GPU path on a one-GPU box).
 * The collective library is librccl unless the environment names another one (SCL_RCCL_LIB=/path/to/lib: a newer RCCL build, or
 * the stand-in the tests link -- tests/cpp/mock_rccl.cpp, the ranks' element-wise minimum formed on the host --, which is how the
 * G > 1 control flow of exchange = 2 runs on a one-GPU box: RCCL itself refuses two ranks on one device).
 * With more than one shard every shard also keeps the query-side rows of the newest 1024 keyframes of each OTHER shard (copied
 * device to device when a keyframe is appended: 66 KB per keyframe at 64x120, n_devices x 1024 rows per shard), so that searching
 * for a recent keyframe -- what detection does -- moves nothing between the devices; older keyframes are copied when asked for.
 * Geometric verification of one scan's candidates (scl_icp_align_batch) is spread over the shards by candidate;
 * other geometry calls and the keyframe store live on devices[0].  cfg->device is ignored. */
int  scl_create_sharded(const scl_config *cfg, const int *devices, int n_devices, int exchange, scl_engine **out);
/* number of shards (1 for scl_create) and the exchange in use (0 none, 1 host merge, 2 min all-reduces through the collective library); either may be NULL */
int  scl_shard_info(const scl_engine *e, int *n_shards, int *exchange);

/* ---- the six virtuals of scan_descriptor (D.h:21-36) ----------------------- */

/* makeAndSaveDescriptorAndKey, D.h:25 / 1604-1611 (called at DM.h:1002).
 * out_values (R*S floats, may be NULL) receives the vector the reference returns. */
int  scl_make_and_save(scl_engine *e, const void *points, int n_points, int stride_bytes,
                       int8_t robot, int index, float *out_values);
/* saveDescriptorAndKey, D.h:27 / 1572-1585 (called at DM.h:627). */
int  scl_save_from_wire(scl_engine *e, const float *values, int8_t robot, int index);
/* detectIntraLoopClosureID, D.h:29 / 1613-1674 (called at DM.h:1078).
 * shift = ring shift as float (D.h:1665); dist (may be NULL) = the float-narrowed
 * running minimum of D.h:1655 widened back to double. */
int  scl_detect_intra(scl_engine *e, int cur, int *loop_id, float *shift, double *dist);
/* detectInterLoopClosureID, D.h:31 / 1676-1756 (called at DM.h:1280);
 * yaw_rad as D.h:1752.  Latent defects of the reference are repaired, see DESIGN.md. */
int  scl_detect_inter(scl_engine *e, int cur, int *loop_id, float *yaw_rad, double *dist);
/* getIndex, D.h:33 / 1758-1761 */
int  scl_get_index(const scl_engine *e, int key, int8_t *robot, int *index);
/* getSize, D.h:35 / 1763-1766: returns the size (>= 0) or a negative status. */
int  scl_get_size(const scl_engine *e, int id);

/* ---- bulk / building-block entry points ----------------------------------- */

/* makeDescriptors (DM.h:996-1002) in one call: pcl::VoxelGrid with leaf `leaf` (descriptLeafSize) on the raw scan,
 * then makeAndSaveDescriptorAndKey on the filtered cloud, which stays on the device.  *n_filtered (optional) =
 * points after the filter.  Same descriptor, key and database state as scl_voxel_grid + scl_make_and_save. */
int  scl_make_and_save_filtered(scl_engine *e, const void *points, int n_points, int stride_bytes, float leaf,
                                int8_t robot, int index, float *out_values, int *n_filtered);
/* makeDescriptors for a BATCH of keyframes (DM.h:988-1025 runs once per keyframe; a robot team's keyframes arrive together, and a
 * map that is loaded or replayed arrives all at once): `count` clouds -> `count` descriptors appended in order, as `count` calls of
 * scl_make_and_save would (same descriptors, keys and database state, bit for bit).  Groups of up to 16 scans cost two kernel
 * launches each (one scatter over all the clouds, one ingest that writes every array of their slots); a group's clouds travel to
 * the device while the group before is processed -- by DMA when the buffers are pinned (scl_host_alloc / scl_host_register).
 * robots / indexs may be NULL (robot 0, index = slot); out_values may be NULL, else count x R*S floats (the wire format of
 * global_descriptor.values, D.h:1446-1456, per scan). */
int  scl_make_and_save_many(scl_engine *e, const void *const *clouds, const int *n_points, int count, int stride_bytes,
                            const int8_t *robots, const int *indexs, float *out_values);
/* The per-incoming-scan pipeline from raw points in one call: for scan i, makeAndSaveDescriptorAndKey (DM.h:1002; keyframe
 * key_i = size before the call + i) and then the full-database detection of that keyframe over [0, key_i - NUM_EXCLUDE_RECENT)
 * (D.h:1627; scl_detect_full's rule): nn_idx[i] = arg-min keyframe (-1: empty range), shift[i], dist[i] = its fp64 SC distance --
 * what scl_make_and_save + scl_detect_full_range give scan by scan, bit for bit.  Same grouping and copy overlap as
 * scl_make_and_save_many; the caller applies the threshold (dist < dist_thres, D.h:1662). */
int  scl_stream_from_points(scl_engine *e, const void *const *clouds, const int *n_points, int n_scans, int stride_bytes,
                            const int8_t *robots, const int *indexs, int *nn_idx, int *shift, double *dist, float *out_values);
/* The same for keyframes whose clouds are already in the on-device keyframe store (scl_keyframe_put, DM.h:674): keyframes
 * first_index .. first_index + count - 1 of `robot` get their descriptors built from the stored clouds and are appended as
 * (robot, index); then each is searched for over [0, key - NUM_EXCLUDE_RECENT).  Nothing crosses PCIe but the results (and the
 * descriptor values when out_values is given): the path's rate with its inputs resident in HBM.  Same results as
 * scl_stream_from_points on the same clouds. */
int  scl_stream_from_store(scl_engine *e, int robot, int first_index, int count, int *nn_idx, int *shift, double *dist, float *out_values);
/* Pinned host memory for point clouds: a cloud handed over from such a buffer goes to the device by DMA, without the runtime's
 * staging copy, and overlaps the kernels of the scans before it.  scl_host_alloc buffers are freed by scl_host_free or with the
 * engine; scl_host_register pins memory the caller owns (e.g. a pcl::PointCloud's points) until scl_host_unregister. */
int  scl_host_alloc(scl_engine *e, size_t bytes, void **out);
int  scl_host_free(scl_engine *e, void *p);
int  scl_host_register(scl_engine *e, void *p, size_t bytes);
int  scl_host_unregister(scl_engine *e, void *p);
/* makeScancontext only (D.h:1404-1461), nothing is stored. */
int  scl_make_descriptor(scl_engine *e, const void *points, int n_points, int stride_bytes,
                         float *out_values);
/* count descriptors at once; robots/indexs may be NULL (robot 0, index = slot). */
int  scl_save_bulk(scl_engine *e, const float *values, int count,
                   const int8_t *robots, const int *indexs);
/* read back: descriptor in wire format, ring key (D.h:1463-1475, R floats),
 * sector key (D.h:1477-1489, S doubles) */
int  scl_get_descriptor(const scl_engine *e, int key, float *values);
int  scl_get_ringkey(const scl_engine *e, int key, float *ringkey);
int  scl_get_sectorkey(const scl_engine *e, int key, double *sectorkey);

/* `count` descriptors first .. first+count-1 in wire format (count * R*S floats) */
int  scl_get_descriptors(const scl_engine *e, int first, int count, float *values);
/* Reverse of getIndex (D.h:1758-1761): the database key of keyframe `index` of robot `robot`, or -1 (*key) when it
 * is not in the database -- the index mapping of performInterLoopClosure, DM.h:1281-1284. */
int  scl_find_key(const scl_engine *e, int8_t robot, int index, int *key);
/* The whole keyframe database as a flat file: header {magic, version, R, S, N}, float32[N][R*S] descriptors in
 * wire order (D.h:1446-1455), then N x {int32 robot, int32 index} (the map of D.h:1758-1761).  Keys, norms and the
 * tiled layouts are derived data and are rebuilt on load.  load appends to the engine's database (an empty one-GPU engine
 * reproduces the dumped one exactly: same keys, same detections -- the header's reserved words carry the counter and range
 * of detectInterLoopClosureID's periodic tree, D.h:1691-1703; a sharded engine restarts that period); the grid must match.
 * load checks the file's length against its header before sizing anything by it and fails with SCL_ERR_INVALID_ARG /
 * SCL_ERR_NOMEM, never by an exception; a load that fails midway (I/O error, device memory) leaves the keyframes appended
 * so far in the database: *n_loaded says how many. */
int  scl_db_dump_file(scl_engine *e, const char *path);
int  scl_db_load_file(scl_engine *e, const char *path, int *n_loaded);

/* Stage an external query descriptor (wire format) that is NOT stored in the DB;
 * afterwards pass SCL_QUERY_STAGED as `query`.  Used by the sharded (multi-GPU)
 * driver, where the query keyframe may live on another rank. */
#define SCL_QUERY_STAGED (-1)
int  scl_stage_query(scl_engine *e, const float *values);

/* Exact k nearest ring keys among DB slots [lo, hi): replaces both KD-trees
 * (libnabo D.h:1631-1642, nanoflann D.h:1699-1716).  Squared L2 in fp32 with
 * nanoflann's accumulation order (nanoflann.hpp:383-408); ascending distance,
 * ties -> lower index; unfilled slots idx = -1, d2 = FLT_MAX.
 * Returns the number found in *found (may be NULL). */
int  scl_ringkey_topk(scl_engine *e, int query, int lo, int hi, int k,
                      int *idx, float *d2, int *found);
/* distanceBtnScanContext (D.h:1538-1569) of `query` against cand[0..n)
 * (cand == NULL: slots 0..n-1).  dist[i] is bit-identical to the fp64 value of
 * the sequential CPU evaluation; shift[i] the arg-min ring shift. */
int  scl_sc_distance_batch(scl_engine *e, int query, const int *cand, int n,
                           double *dist, int *shift);
/* The distance MATRIX of north_star ("the column-shifted SC distance matrix over the keyframe database"): row r =
 * distanceBtnScanContext (D.h:1538-1569) of keyframe queries[r] (or a staged query, -1 - slot) against the keyframes
 * lo .. hi-1; dist / shift are nq x (hi - lo), row-major.  Every entry is the exact fp64 evaluation (bit-identical to the
 * sequential CPU evaluation, as scl_sc_distance_batch).  On the screened grids (64x120, 80x180) the screening pass first tells,
 * per pair, which of the 2 SR + 1 shifts can still hold the minimum (a guaranteed bound: DESIGN.md section 4), and the fp64
 * arithmetic of the reference is then carried out at those shifts only -- the result is the same number; the environment switch
 * SCL_MATRIX_PLAIN=1 evaluates every shift of every pair.  Several rows share a launch and a launch's results travel to the
 * host while the next one runs. */
int  scl_sc_distance_matrix(scl_engine *e, const int *queries, int nq, int lo, int hi, double *dist, int *shift);
/* BASELINE "full-DB" mode: ring-key top-k AND the shifted SC distance against
 * every eligible slot [0, hi) with hi = cur - num_exclude_recent (D.h:1627), then
 * the global arg-min (ties -> lowest slot).  nn_idx/shift/dist describe the best
 * slot; loop_id = nn_idx iff dist < dist_thres else -1.
 * With query == SCL_QUERY_STAGED, `hi` is given explicitly via scl_detect_full_range. */
int  scl_detect_full(scl_engine *e, int cur, int *loop_id, int *nn_idx, int *shift, double *dist);
int  scl_detect_full_range(scl_engine *e, int query, int lo, int hi,
                           int *nn_idx, int *shift, double *dist);
/* Pipelined form of scl_detect_full_range for streams of scans: submit enqueues the pass and returns a
 * ticket at once (up to 8 may be in flight, results are collected in any order); collect blocks until
 * that pass has finished.  scl_detect_full_range == submit + collect. */
int  scl_detect_full_submit(scl_engine *e, int query, int lo, int hi, int *ticket);
/* Several keyframes of the database as queries in ONE launch (scans of several robots arriving together,
 * or a backlog): query i = keyframe queries[i] against [lo[i], hi[i]); tickets[i] is collected like a
 * ticket of scl_detect_full_submit.  Up to 4 queries share a launch (more are split); staged queries
 * (SCL_QUERY_STAGED) and grids without the fused kernel fall back to one launch per query. */
int  scl_detect_full_submit_many(scl_engine *e, const int *queries, const int *lo, const int *hi, int n_queries, int *tickets);
/* A backlog of n_queries scans in one call: the submit / collect pipeline of the two calls above run natively
 * (scans_per_launch queries per kernel launch, launches_in_flight launches enqueued ahead), results in
 * submission order.  Blocks until the last scan is done; no other pass may be in flight. */
int  scl_detect_full_stream(scl_engine *e, const int *queries, const int *lo, const int *hi, int n_queries,
                            int scans_per_launch, int launches_in_flight, int *nn_idx, int *shift, double *dist);
int  scl_detect_full_collect(scl_engine *e, int ticket, int *nn_idx, int *shift, double *dist);
/* Diagnostic view of the screening pass behind the full-DB mode on the 64x120 grid (scl_slam_amd/csrc/sc_screen.hip):
 * approx[i] = the reduced-precision (fp16 matrix-core) evaluation of distanceBtnScanContext (D.h:1538-1569) of
 * `query` against slot lo + i -- the reference's own alignment, its 13 shifts -- guaranteed within *eps of the fp64
 * value (-inf: the keyframe is always scored exactly); survivors[0 .. *n_survivors) = the slots (ascending) that can
 * still hold the minimum and are re-scored by the exact fp64 kernel.  The full-DB entry points return the exact
 * winner; this call exists so that tests can check the bound on the hardware.  survivors (n entries), n_survivors
 * and eps may be NULL.  SCL_ERR_UNSUPPORTED on other grids. */
int  scl_screen_distances(scl_engine *e, int query, int lo, int hi, float *approx, int *survivors, int *n_survivors, float *eps);
/* ... of up to 16 scans in ONE screening launch (the stream form's batch: the products' second form, every column of its matrix
 * products): approx[i * n + k] = scan queries[i] against slot lo + k, n = the clipped range's length.  No survivor lists. */
int  scl_screen_distances_many(scl_engine *e, const int *queries, int n_queries, int lo, int hi, float *approx, float *eps);
/* The ring-key top-k (num_candidates entries) computed as part of the last scl_detect_full[_range]. */
int  scl_get_last_topk(scl_engine *e, int k, int *idx, float *d2);
/* Reference-faithful candidates for the sharded driver: local ring-key top-k in
 * [lo,hi) plus the SC distance/shift of each (one device pass, no host round trip). */
int  scl_topk_with_distance(scl_engine *e, int query, int lo, int hi, int k,
                            int *idx, float *d2, double *dist, int *shift, int *found);

/* ---- geometric verification (PCL objects inlined at DM.h:1108-1121, 1211-1230) */

typedef struct scl_icp_params {
    int    max_iterations;            /* icp.setMaximumIterations(50)        DM.h:1110 */
    double max_correspondence_dist;   /* icp.setMaxCorrespondenceDistance(100) DM.h:1109 */
    double transformation_epsilon;    /* icp.setTransformationEpsilon(1e-6)  DM.h:1111 */
    double euclidean_fitness_epsilon; /* icp.setEuclideanFitnessEpsilon(1e-6) DM.h:1112 */
    int    estimator;                 /* 0 = point-to-point SVD (reference, DM.h:1108), 1 = point-to-plane LLS
                                         (BASELINE configs[2]; target normals by PCA within normal_radius) */
    double normal_radius;             /* neighbourhood radius of the target normals, metres (1.0) */
} scl_icp_params;

int  scl_icp_default_params(scl_icp_params *p);
/* pcl::IterativeClosestPoint::align + getFitnessScore, DM.h:1108-1121.
 * T = 4x4 row-major final transformation (source -> target). */
int  scl_icp_align(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                   int stride_bytes, const scl_icp_params *p,
                   float T[16], float *fitness, int *converged, int *iterations);
/* The same alignment for the n_targets loop candidates of one scan (BASELINE configs[2]: ICP on the
 * top-25 candidates): one source against tgts[c] (n_tgts[c] points), up to four alignments in flight on
 * internal streams.  T: n_targets x 16 floats; fitness / converged / iterations: n_targets entries
 * (each may be NULL).  Per-candidate results are those of scl_icp_align. */
int  scl_icp_align_batch(scl_engine *e, const void *src, int n_src, const void *const *tgts, const int *n_tgts,
                         int n_targets, int stride_bytes, const scl_icp_params *p,
                         float *T, float *fitness, int *converged, int *iterations);
/* CorrespondenceEstimation::determineCorrespondences, DM.h:1211-1215:
 * exact 1-NN of every source point in the target (ties -> lowest target index). */
int  scl_nn_correspondences(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                            int stride_bytes, int *nn_index, float *nn_dist2);
/* The correspondences of the source moved by T (row-major 4x4, DM.h:247-249 arithmetic), searched from those of the
 * unmoved source: the warm neighbour search of an ICP iteration (pcl::IterativeClosestPoint's loop body, DM.h:1119) on
 * its own.  Results equal scl_nn_correspondences on the moved cloud. */
int  scl_nn_correspondences_moved(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                                  int stride_bytes, const float T[16], int *nn_index, float *nn_dist2);
/* TransformationEstimationSVD::estimateRigidTransformation, DM.h:1228-1230,
 * over correspondence pairs (src_index[i], tgt_index[i]). */
int  scl_rigid_svd(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                   int stride_bytes, const int *src_index, const int *tgt_index, int n_corr,
                   float T[16]);
/* CorrespondenceRejectorSampleConsensus::getCorrespondences, DM.h:1218-1225 (ransacMaxIter DM.h:187,
 * ransacOutlierTreshold DM.h:188).  Deterministic: hypothesis h draws its 3 correspondences from a
 * counter-based generator of (seed, h); every one of max_iterations hypotheses is scored; best = most
 * inliers, ties -> lowest h.  inlier_mask[n_corr] (0/1), T_model = the winning 3-point model (may be NULL). */
int  scl_ransac_correspondences(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                                int stride_bytes, const int *src_index, const int *tgt_index, int n_corr,
                                int max_iterations, double inlier_threshold, uint64_t seed,
                                int *inlier_mask, int *n_inliers, int *best_hypothesis, float T_model[16]);
/* The compute core of geometricVerificationService, DM.h:1211-1243: NN correspondences -> RANSAC ->
 * SVD transform on the inliers -> gate `inliers >= inlier_ratio * correspondences` (inlierTreshold DM.h:189). */
int  scl_geometric_verification(scl_engine *e, const void *src, int n_src, const void *tgt, int n_tgt,
                                int stride_bytes, int ransac_iterations, double inlier_threshold,
                                double inlier_ratio, uint64_t seed, float T[16], int *success,
                                int *n_correspondences, int *n_inliers);
/* pcl::VoxelGrid::filter with one leaf size (DM.h:501,503; DM.h:996-998, 1183-1185, 1200-1201): one point
 * per occupied voxel = centroid of x, y, z and intensity, ascending voxel index.  out must hold out_capacity
 * records (n_points always suffices).  When the voxel index range overflows int32 the input is returned
 * unchanged, as PCL does. */
int  scl_voxel_grid(scl_engine *e, const void *points, int n_points, int stride_bytes, float leaf,
                    void *out, int out_capacity, int *n_out);
/* pcl::getTransformation(x, y, z, roll, pitch, yaw) (DM.h:223,241) as a row-major 4x4; host-side helper */
int  scl_pose_to_matrix(float x, float y, float z, float roll, float pitch, float yaw, float T[16]);
/* pcl::getTranslationAndEulerAngles (DM.h:1133, 1139): roll = atan2(R21, R22), pitch = asin(-R20), yaw = atan2(R10, R00) */
int  scl_matrix_to_pose(const float T[16], float *x, float *y, float *z, float *roll, float *pitch, float *yaw);
/* The tail of the ICP block, DM.h:1130-1141 (and DM.h:1249-1259 with the SVD transform): tfCorrect = T_icp * tfWrong
 * (pose_cur = x, y, z, roll, pitch, yaw of the current keyframe, float like Affine3f), poseFrom = its Euler pose,
 * poseTo = pose_pre, result = poseFrom.between(poseTo) in double: between_xyz_q = translation x, y, z and the unit
 * quaternion x, y, z, w (w >= 0) -- the fields of loop_info.betPose / poseBetween --, between_rpy (optional) =
 * roll, pitch, yaw of the same rotation. */
int  scl_loop_pose_between(const float T_icp[16], const float pose_cur[6], const float pose_pre[6], double between_xyz_q[7], double between_rpy[3]);
/* loopFindNearKeyframes, DM.h:1163-1186: concatenation of transformPointCloud(cloud_i, T_i) (DM.h:234-253)
 * followed by the voxel filter.  transforms = n_clouds row-major 4x4 matrices. */
int  scl_assemble_submap(scl_engine *e, const void *const *clouds, const int *counts, const float *transforms,
                         int n_clouds, int stride_bytes, float leaf, void *out, int out_capacity, int *n_out);
/* paramsServer::transformPointCloud, DM.h:234-253 (xyz transformed, rest copied) */
int  scl_transform_cloud(scl_engine *e, const void *in, int n, int stride_bytes,
                         const float T[16], void *out);

/* ---- on-device keyframe store ------------------------------------------------
 * robots[id].keyFrameArray (DM.h:86; filled at DM.h:983-985) kept resident in HBM, so that submap
 * assembly and the ICP of a loop candidate move only poses across PCIe, not clouds.
 * Records have one stride per engine (fixed by the first put).  Keyframes are dense per robot
 * (index = position in keyFrameArray); putting an existing index replaces that cloud. */
int  scl_keyframe_put(scl_engine *e, int robot, int index, const void *points, int n_points, int stride_bytes);
/* keyFrameArray.size() of one robot (highest stored index + 1) */
int  scl_keyframe_count(const scl_engine *e, int robot);
/* read one stored cloud back (out may be NULL to query the size only) */
int  scl_keyframe_get(scl_engine *e, int robot, int index, void *out, int out_capacity, int *n_points);
/* loopFindNearKeyframes(nearKeyframes, key, searchNum), DM.h:1163-1186, from the store: keyframes
 * key-searchNum .. key+searchNum that exist, each moved by its pose, concatenated in index order and
 * voxel-filtered with `leaf`.  poses: (2*search_num+1) row-major 4x4 matrices, poses[i] belonging to
 * keyframe key-search_num+i (entries of keyframes outside [0, count) are ignored). */
int  scl_submap_from_store(scl_engine *e, int robot, int key, int search_num, const float *poses, float leaf,
                           void *out, int out_capacity, int *n_out);
/* performIntraLoopClosure stage 2, DM.h:1104-1121, entirely on the device: source = submap(key_cur, 0),
 * target = submap(key_pre, search_num), size gate (DM.h:1108: fewer than min_src_points / min_tgt_points
 * -> no alignment, *converged = 0, T = identity), then scl_icp_align's alignment. */
int  scl_loop_icp_from_store(scl_engine *e, int robot, int key_cur, const float *pose_cur,
                             int key_pre, int search_num, const float *poses_pre, float leaf,
                             const scl_icp_params *p, int min_src_points, int min_tgt_points,
                             float T[16], float *fitness, int *converged, int *iterations, int *n_src, int *n_tgt);
/* The same stage for the n_candidates loop candidates of ONE scan (BASELINE configs[2]: ICP on the top-25 candidates):
 * the source submap is built once, every candidate's target submap comes from the store (keys_pre[c], its window of
 * 2 * search_num + 1 poses at poses_pre + c * (2 * search_num + 1) * 16), and the ICP loops of all candidates run fused
 * -- every loop step is one launch for the whole batch.  Outputs are per candidate (T: n_candidates x 16); candidates
 * that fail the size gate keep T = identity, converged = 0.  Per-candidate results equal scl_loop_icp_from_store's. */
int  scl_loop_icp_batch_from_store(scl_engine *e, int robot, int key_cur, const float *pose_cur,
                                   int n_candidates, const int *keys_pre, int search_num, const float *poses_pre, float leaf,
                                   const scl_icp_params *p, int min_src_points, int min_tgt_points,
                                   float *T, float *fitness, int *converged, int *iterations, int *n_src, int *n_tgts);

/* geometricVerificationService (DM.h:1189-1268) with the submap taken from the keyframe store: voxel filter of
 * the received cloud (src_leaf), submap(key_pre, search_num) from stored keyframes (leaf), size gate
 * (DM.h:1204), then exactly scl_geometric_verification's correspondences -> RANSAC -> SVD -> inlier gate;
 * no cloud but the received one crosses PCIe. */
int  scl_geometric_verification_from_store(scl_engine *e, const void *src, int n_src, int stride_bytes, float src_leaf,
                                           int robot, int key_pre, int search_num, const float *poses_pre, float leaf,
                                           int min_src_points, int min_tgt_points,
                                           int ransac_iterations, double inlier_threshold, double inlier_ratio, uint64_t seed,
                                           float T[16], int *success, int *n_src_filtered, int *n_tgt,
                                           int *n_correspondences, int *n_inliers);

/* ---- measurement ----------------------------------------------------------- */
int  scl_profile_enable(scl_engine *e, int on);   /* 0 off, 1 every kernel family, 2 SC distance only,
                                                     3 SC distance only, sampled: one launch in thirteen -- in the stream form one
                                                     pair around every second chunk's launch groups, counted as that many launches
                                                     (an event keeps the launch behind it back by 4-6 us) */
int  scl_profile_reset(scl_engine *e);
int  scl_profile_get(scl_engine *e, scl_profile *out);
/* Counters of the ICP loop's LDS-tiled neighbour search (icp.hip K4c) since the last reset, in builds made with
 * -DSCL_DIAGNOSTICS (zeros otherwise): [0] rounds asked for, [1] rounds whose box did not fit, [2] lanes finished in memory,
 * [3] table entries staged, [4] points staged, [5] row visits, [6] points compared; [8..14] 10 ns ticks of every workgroup's
 * first thread per phase (up to the ball round, box, table, row scan, staging, the queries' walk, reduction).  Device-wide. */
void scl_debug_icp_tile_stats(unsigned long long out[16], int reset);
/* The full-database pass aligns every (scan, keyframe) pair (fastAlignUsingVkey, D.h:1491-1511) with an fp32 correlation
 * filter on the matrix cores and falls back to the reference's own fp64 evaluation wherever two shifts are closer than the
 * filter's error margin.  pairs = pairs aligned since the last reset, fallbacks = those decided by the fp64 evaluation (the
 * rate depends on the data: flat or periodic sector keys fall back).  Either pointer may be NULL; reset != 0 clears both. */
int  scl_alignment_stats(scl_engine *e, uint64_t *pairs, uint64_t *fallbacks, int reset);
/* The full-database pass scores exactly (fp64, D.h:1538-1569) only the keyframes whose screening bound reaches the smallest
 * bound of their scan (sc_screen.hip): queries = scans that went through an exact pass since the last reset, survivors = keyframes
 * scored exactly for them, max_survivors = the largest such count of one scan.  The rate of the pass depends on it: a database
 * where every keyframe survives runs at the exact kernel's rate.  On a sharded engine a scan counts once per shard.  Any
 * pointer may be NULL; reset != 0 clears the counters. */
int  scl_survivor_stats(scl_engine *e, uint64_t *queries, uint64_t *survivors, uint64_t *max_survivors, int reset);
int  scl_device_name(const scl_engine *e, char *buf, int buflen);
/* GB/s of `reps` copies of `bytes` from pinned host memory to the device, one after the other on the engine's copy stream (HIP events):
 * the link's rate, i.e. the floor of scl_stream_from_points per byte of point cloud. */
int  scl_host_copy_rate(scl_engine *e, size_t bytes, int reps, double *gbytes_per_s);
/* Self test of the device's float atan -- xy2theta (D.h:1352-1374) calls std::atan(float), here glibc's atanf restated in fp32
 * (csrc/device_common.hpp) --: checksums[b] = sum mod 2^64 over the 2^24 float bit patterns of block first_block + b of
 * splitmix64((bits << 32) | result bits), NaN results counted as 0x7fc00000.  tests/golden/atanf_blocks.json holds the 256 values
 * of libm's atanf (oracle/tools/atanf_exhaustive.c). */
int  scl_selftest_atanf_blocks(scl_engine *e, int first_block, int n_blocks, uint64_t *checksums);
/* Self test of the descriptor kernel's two ways to a point's (ring, sector, dropped?) -- the reference's chain (D.h:1425-1435) and the
 * cheap approximations that stand in for it wherever they are provably the same integers (csrc/device_common.hpp) -- on n_points
 * generated points of the engine's grid (mode 0: uniform; 1: on ring boundaries; 2: on sector boundaries; 3: zeros, denormals, huge,
 * inf, NaN): *sure = points the fast path answered, *disagreements = those of them where the chain says otherwise (must be 0). */
int  scl_selftest_bin_paths(scl_engine *e, int mode, uint64_t seed, uint64_t n_points, uint64_t *disagreements, uint64_t *sure);

/* Self test of the verification path's own stable radix sort (csrc/device_sort.hip; it replaced hipCUB's): the n (key, value) pairs
 * sorted on the key bits [0, bits), equal keys in their input order.  key_bytes 4 or 8; n_segments > 1 (8-byte keys): every segment
 * [segment_offsets[s], segment_offsets[s + 1]) sorted on its own, as the 26 submaps of a query are (DM.h:1183-1185 through PCL's VoxelGrid). */
int  scl_selftest_sort_pairs(scl_engine *e, int key_bytes, const void *keys, const uint32_t *values, int n, int bits, const int *segment_offsets,
                             int n_segments, void *keys_out, uint32_t *values_out);
/* ... and of its prefix sums (cell counts -> cell starts): out[i] = in[0] + ... + in[i - 1], + in[i] when inclusive != 0 */
int  scl_selftest_prefix_sum(scl_engine *e, const int32_t *in, int n, int inclusive, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif /* SCL_ENGINE_H */
