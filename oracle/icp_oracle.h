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
    int    estimator;                  /* 0 point-to-point (reference), 1 point-to-plane (BASELINE configs[2]) */
    double normal_radius;              /* neighbourhood of the target normals (1.0 m) */
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

/* Target normals for the point-to-plane estimator: PCA of all target points within `radius` of the point
 * (itself included), fp64 covariance, eigenvector of the smallest eigenvalue (cyclic Jacobi); fewer than
 * 3 neighbours -> (0,0,0).  normals: n_tgt x 3 floats. */
void icpo_normals(const void *tgt, int n_tgt, int stride_bytes, double radius, float *normals);

/* CorrespondenceRejectorSampleConsensus (DM.h:1218-1225), restated as a deterministic RANSAC:
 * hypothesis h draws 3 distinct correspondences with a counter-based generator (splitmix64 of
 * (seed, h, draw)), fits the rigid transform of the 3 pairs, and scores every correspondence with
 * |T p - q|^2 < thr^2 (fp64).  ALL max_iterations hypotheses are scored (no probabilistic early exit:
 * at least as good as PCL's adaptive loop); best = most inliers, ties -> lowest h.  PCL's own random
 * sequence cannot be reproduced offline, so parity with the reference is statistical.
 * inlier_mask[n_corr] receives 0/1; returns the inlier count (or -1 if n_corr < 3). */
int  icpo_ransac(const void *src, const void *tgt, int stride_bytes, const int *src_index, const int *tgt_index,
                 int n_corr, int max_iterations, double inlier_threshold, unsigned long long seed,
                 int *inlier_mask, int *best_hypothesis, double T_best[12]);

/* geometricVerificationService core (DM.h:1211-1243): NN correspondences -> RANSAC -> SVD on the
 * inliers -> inlier-ratio gate.  success = n_inliers >= inlier_ratio * n_corr (DM.h:1238). */
int  icpo_geometric_verification(const void *src, int n_src, const void *tgt, int n_tgt, int stride_bytes,
                                 int ransac_iterations, double inlier_threshold, double inlier_ratio,
                                 unsigned long long seed, float T[16], int *success, int *n_corr, int *n_inliers);

/* pcl::VoxelGrid::filter with one leaf size for all axes (DM.h:501,503; call sites DM.h:996-998,
 * 1183-1185, 1200-1201), restated from PCL's published algorithm (PARITY UNPINNED):
 * bounding box of the finite points -> voxel coords floor(p * (1/leaf)) - min_b (fp32) -> linear index
 * (x fastest) -> one output point per occupied voxel = centroid of x, y, z and intensity (the float at
 * byte offset 16 when the record has one), fp32 sums in ascending input order within the voxel, divided
 * by the count; output in ascending voxel index; other record bytes are zero.
 * Returns the number of output points, or -1 when the index range would overflow int32 (PCL then
 * returns the input unchanged). */
int  icpo_voxel_grid(const void *in, int n, int stride_bytes, float leaf, void *out);

/* pcl::getTransformation(x, y, z, roll, pitch, yaw) as used at DM.h:223,241: fp32, row-major 4x4 */
void icpo_pose_to_matrix(float x, float y, float z, float roll, float pitch, float yaw, float T[16]);

/* 3x3 SVD-based rotation for a cross-covariance matrix (exposed for tests):
 * H = sum (q - qbar)(p - pbar)^T  (dst x src), R = U diag(1,1,det) V^T */
void icpo_rotation_from_covariance(const double H[9], double R[9]);

#ifdef __cplusplus
}
#endif
#endif
