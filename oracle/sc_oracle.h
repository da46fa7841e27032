/*
 * sc_oracle.h -- CPU restatement of the reference's Scan Context hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference lines it restates.  Abbreviations:
 *   D.h  = /root/reference/include/descriptor.h
 *   NF   = /root/reference/include/nanoflann.hpp
 *
 * Parity status: the reference has no tests and its Eigen/PCL/libnabo
 * dependencies are absent, so the SC arithmetic is "parity unpinned" by the
 * reference itself; it is pinned by hand-derived known-answer cases
 * (tests/test_oracle_kat.py).  The ring-key kNN *is* pinned: golden lists
 * were produced by the reference's own vendored nanoflann (oracle/_ref).
 */
#ifndef SC_ORACLE_H
#define SC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors the scan_context_descriptor ctor arguments, D.h:1307-1316. */
typedef struct sco_config {
    int    num_ring;            /* PC_NUM_RING              (20)   */
    int    num_sector;          /* PC_NUM_SECTOR            (60)   */
    int    num_candidates;      /* NUM_CANDIDATES_FROM_TREE (3)    */
    double dist_thres;          /* SC_DIST_THRES            (0.14) */
    double lidar_height;        /* LIDAR_HEIGHT             (1.65) */
    double max_radius;          /* PC_MAX_RADIUS            (80.0) */
    int    num_exclude_recent;  /* NUM_EXCLUDE_RECENT       (100)  */
    int    tree_making_period;  /* TREE_MAKING_PERIOD_      (10)   */
    double search_ratio;        /* SEARCH_RATIO             (0.1)  */
    float  knn_exclude_eps;     /* 0 = nanoflann semantics; FLT_EPSILON = libnabo self-match exclusion */
} sco_config;

void   sco_default_config(sco_config *c);

/* scalar helpers */
double sco_atan_pos(double x);                       /* rounds 1-4's fp64 atan, x >= 0 (kept for the envelope record) */
unsigned long long sco_atanf_block_checksum(int block);   /* see tests/golden/atanf_blocks.json */
float  sco_atanf_glibc(float x);                     /* glibc's float atanf restated (fp32 ops only); == libm over all 2^32 inputs */
float  sco_xy2theta(float x, float y);               /* D.h:1352-1374 */

/* descriptor + keys.  Every `double *desc/sc` is an R x S matrix stored
 * COLUMN-major (desc[c*R + r]) exactly like the reference's Eigen::MatrixXd;
 * the wire vector vT / `values` is ROW-major floats (D.h:1446-1455, 1576-1582). */
void   sco_make_scancontext(const sco_config *c, const void *pts, int n, int stride_bytes,
                            double *desc, float *vT);                      /* D.h:1404-1461 */
void   sco_ringkey(int R, int S, const double *desc, float *key);          /* D.h:1463-1475 */
void   sco_sectorkey(int R, int S, const double *desc, double *vkey);      /* D.h:1477-1489 */
void   sco_circshift(int R, int S, const double *in, int shift, double *out); /* D.h:1376-1395 */
int    sco_fast_align(int S, const double *vkey1, const double *vkey2);    /* D.h:1491-1511 */
double sco_dist_direct(int R, int S, const double *sc1, const double *sc2);/* D.h:1513-1536 */
/* reference-shaped: per-shift matrix copy, norms evaluated twice.  D.h:1538-1569 */
void   sco_distance(const sco_config *c, const double *sc1, const double *sc2,
                    double *dist, int *shift);
/* same results, no copies / precomputed norms (CPU baseline variant C) */
void   sco_distance_fast(const sco_config *c, const double *sc1, const double *sc2,
                         double *dist, int *shift);

/* exact brute-force kNN with nanoflann's fp32 accumulation order (NF:383-408)
 * and result-set rule (NF:177-199, 1360).  keys: N x R row-major.
 * Visit order = ascending index, so equal distances keep the lower index first.
 * Returns number found (<= k); unfilled slots get idx -1 / d2 FLT_MAX. */
int    sco_knn(const float *keys, int N, int R, const float *query, int k,
               float exclude_eps, int *idx, float *d2);

/* database object mirroring scan_context_descriptor state, D.h:1768-1800 */
typedef struct sco_db sco_db;
sco_db *sco_db_create(const sco_config *c);
void    sco_db_destroy(sco_db *db);
void    sco_db_save_wire(sco_db *db, const float *values, int8_t robot, int index); /* D.h:1572-1602 */
void    sco_db_make_and_save(sco_db *db, const void *pts, int n, int stride_bytes,
                             int8_t robot, int index, float *vT);                    /* D.h:1604-1611 */
int     sco_db_size(const sco_db *db);                                                /* D.h:1763-1766 */
void    sco_db_get_index(const sco_db *db, int key, int8_t *robot, int *index);       /* D.h:1758-1761 */
const double *sco_db_desc(const sco_db *db, int key);
const float  *sco_db_ringkey(const sco_db *db, int key);
/* D.h:1613-1674.  *dist receives the (float-narrowed) running minimum widened to double;
 * *dist_exact the un-narrowed fp64 distance of the winning candidate. */
void    sco_db_detect_intra(sco_db *db, int cur, int *loop_id, float *shift,
                            double *dist, double *dist_exact);
/* D.h:1676-1756 with the latent bugs repaired as documented in DESIGN.md:
 * ring keys come from the live key table, the tree covers [0, N-exclude). */
void    sco_db_detect_inter(sco_db *db, int cur, int *loop_id, float *yaw_rad, double *dist);
/* BASELINE "full-DB" mode: argmin of sco_distance over every eligible keyframe
 * [0, cur - exclude_recent); ties -> lowest index; fp64 compare.  nn_idx/shift/dist
 * describe the best keyframe; loop_id = nn_idx iff dist < dist_thres, else -1. */
void    sco_db_detect_full(sco_db *db, int cur, int *loop_id, int *nn_idx, int *shift, double *dist);
/* distance of keyframe `cur` against candidates cand[0..n) (NULL = 0..n-1) */
void    sco_db_distance_batch(sco_db *db, int cur, const int *cand, int n,
                              double *dist, int *shift, int fast);

/* the same batch on a team of `threads` threads (CPU baseline on all host cores): a persistent pool, made on the first
 * call and re-made only when `threads` changes; candidates are dealt in blocks of 16, first come first served */
void    sco_db_distance_batch_mt(sco_db *db, int cur, const int *cand, int n,
                                 double *dist, int *shift, int fast, int threads);
void    sco_pool_shutdown(void);          /* joins the pool's threads (optional; the pool is re-made on demand) */

/* envelope of what cannot be pinned offline (tests only; see sc_oracle.c) */
void sco_ringkey_lanes(int R, int S, const double *desc, int lanes, float *key);
void sco_distance_lanes(const sco_config *c, const double *sc1, const double *sc2, int lanes, double *dist, int *shift);
long long sco_theta_census(long long n, unsigned long long seed, double range, int S, long long *theta_diff);

#ifdef __cplusplus
}
#endif
#endif
