/*
 * icp_oracle.h -- CPU restatement of the geometric-verification path.
 * TEST INFRASTRUCTURE ONLY (see sc_oracle.h for the rules).
 *
 * PARITY UNPINNED: the reference delegates this path to PCL
 * (pcl::IterativeClosestPoint, CorrespondenceEstimation,
 * TransformationEstimationSVD; call sites DM.h:1108-1121, 1211-1230), PCL is
 * pinned only as `pcl_catkin @ afe789a` (dependencies.rosinstall:33-36), is not
 * in /root/reference nor installed, and the reference has no tests.  The
 * functions below restate PCL's published algorithm;
 * they are pinned by known-answer cases (cloud vs rigidly moved copy).
 *
 * DM.h = /root/reference/include/distributedMapping.h
 */
#ifndef ICP_ORACLE_H
#define ICP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct icpo_params {
    int    max_iterations;             /* DM.h:1110  (50)   */
    double max_correspondence_dist;    /* DM.h:1109  (100)  */
    double transformation_epsilon;     /* DM.h:1111  (1e-6) */
    double euclidean_fitness_epsilon;  /* DM.h:1112  (1e-6) */
} icpo_params;

void icpo_default_params(icpo_params *p);

/* exact 1-NN of every source point in the target: squared distance in fp32,
 * ((dx*dx + dy*dy) + dz*dz), ties -> lowest target index.
 * CorrespondenceEstimation::determineCorrespondences, DM.h:1211-1215.
 * use_grid = 0: brute force; 1: uniform-grid accelerated (same results). */
void icpo_nn(const void *src, int n_src, const void *tgt, int n_tgt, int stride_bytes,
             int use_grid, int *nn_index, float *nn_dist2);

/* TransformationEstimationSVD (Umeyama without scale) over pairs
 * (src_index[i], tgt_index[i]); T = 4x4 row-major float.  DM.h:1228-1230. */
int  icpo_rigid_svd(const void *src, const void *tgt, int stride_bytes,
                    const int *src_index, const int *tgt_index, int n_corr, float T[16]);

/* paramsServer::transformPointCloud, DM.h:234-253 (fp32, no FMA) */
void icpo_transform(const void *in, int n, int stride_bytes, const float T[16], void *out);

/* pcl::IterativeClosestPoint::align + getFitnessScore, DM.h:1108-1121 */
int  icpo_icp_align(const void *src, int n_src, const void *tgt, int n_tgt, int stride_bytes,
                    const icpo_params *p, float T[16], float *fitness, int *converged, int *iterations);

/* 3x3 SVD-based rotation for a cross-covariance matrix (exposed for tests):
 * H = sum (q - qbar)(p - pbar)^T  (dst x src), R = U diag(1,1,det) V^T */
void icpo_rotation_from_covariance(const double H[9], double R[9]);

#ifdef __cplusplus
}
#endif
#endif
