// icp.hpp -- geometric verification on the GPU (host-side interface used by engine.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <string>

#include "scl_engine.h"

namespace scl {

struct IcpWorkspace {
    void *buf[24] = {nullptr};
    size_t cap[24] = {0};
    void *pinned = nullptr;
    size_t pinned_cap = 0;
    const void *ext_tgt = nullptr;             // the alignment's target when it is read in place (icp_batch_prepare_all) instead of staged in the workspace
    hipStream_t side = nullptr;                // icp_batch_prepare_all: the candidates' normals run beside the batch's cold search
    bool normals_pending = false;              // icp_batch_prepare_all: the targets' normals are still to be computed (icp_batch_run launches them beside its cold searches)
    double normals_radius = 0.0;
    int n_tgt = 0;                             // the alignment's target as prepared
    // icp_batch_run: a batch's alignments run as parts on streams of their own (one part's solve under another part's search)
    static constexpr int kMaxParts = 4;
    hipStream_t part_stream[kMaxParts - 1] = {nullptr, nullptr, nullptr};   // (part 0 runs on the caller's stream)
    hipEvent_t ev_fork = nullptr, ev_join[kMaxParts - 1] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_norm[kMaxParts] = {nullptr, nullptr, nullptr, nullptr};   // the normals of a part's targets are done
};

void icp_workspace_free(IcpWorkspace *ws);
// counters of the tile search (builds with -DSCL_DIAGNOSTICS; zeros otherwise): scripts/probe_icp_tiles.py
void icp_tile_stats(unsigned long long out[16], bool reset);

int icp_align(IcpWorkspace *ws, hipStream_t stream, int num_cu, const void *src, int n_src,
              const void *tgt, int n_tgt, int stride_bytes, const scl_icp_params &p,
              float T[16], float *fitness, int *converged, int *iterations, std::string *err);
// device-resident clouds: icp_stage_cloud() copies one (device to device) into the workspace, then
// icp_align_staged() runs the same alignment as icp_align() on what has been staged
int icp_stage_cloud(IcpWorkspace *ws, hipStream_t stream, bool target, const void *d_cloud, int n, int stride_bytes,
                    std::string *err);
int icp_stage_cloud_host(IcpWorkspace *ws, hipStream_t stream, bool target, const void *h_cloud, int n, int stride_bytes,
                         std::string *err);
int icp_align_staged(IcpWorkspace *ws, hipStream_t stream, int n_src, int n_tgt, int stride_bytes,
                     const scl_icp_params &p, float T[16], float *fitness, int *converged, int *iterations,
                     std::string *err);
// the alignments of one scan's loop candidates, every loop step one launch for all of them (icp.hip)
int icp_batch_prepare(IcpWorkspace *ws, hipStream_t stream, const void *d_src, int n_src, int n_tgt, int stride,
                      const scl_icp_params &p, std::string *err);
// the same for n alignments whose targets are already on the device (read in place), every step one launch over all of them
int icp_batch_prepare_all(IcpWorkspace *const *wss, int n, IcpWorkspace *ctl, hipStream_t stream, int n_src, const void *const *d_tgts,
                          const int *n_tgts, int stride, const scl_icp_params &p, std::string *err);
int icp_batch_run(IcpWorkspace *const *wss, int nprob, IcpWorkspace *ctl, hipStream_t stream, const void *d_src, int n_src,
                  int stride, const scl_icp_params &p, float *T, float *fitness, int *converged, int *iterations, std::string *err);
int icp_nn_correspondences(IcpWorkspace *ws, hipStream_t stream, int num_cu, const void *src, int n_src,
                           const void *tgt, int n_tgt, int stride_bytes, const float *T_move, int *nn_index, float *nn_dist2,
                           std::string *err);
int icp_rigid_svd(IcpWorkspace *ws, hipStream_t stream, int num_cu, const void *src, int n_src,
                  const void *tgt, int n_tgt, int stride_bytes, const int *src_index, const int *tgt_index,
                  int n_corr, float T[16], std::string *err);
int icp_transform_cloud(IcpWorkspace *ws, hipStream_t stream, const void *in, int n, int stride_bytes,
                        const float T[16], void *out, std::string *err);

int icp_ransac(IcpWorkspace *ws, hipStream_t stream, const void *src, int n_src, const void *tgt, int n_tgt, int stride,
               const int *src_index, const int *tgt_index, int n_corr, int max_iterations, double inlier_threshold,
               unsigned long long seed, int *inlier_mask, int *n_inliers, int *best_hypothesis, float T_model[16],
               std::string *err);
int icp_geometric_verification(IcpWorkspace *ws, hipStream_t stream, int num_cu, const void *src, int n_src,
                               const void *tgt, int n_tgt, int stride, int ransac_iterations, double inlier_threshold,
                               double inlier_ratio, unsigned long long seed, float T[16], int *success, int *n_corr_out,
                               int *n_inliers_out, std::string *err);

int icp_geometric_verification_staged(IcpWorkspace *ws, hipStream_t stream, int n_src, int n_tgt, int stride,
                                      int ransac_iterations, double inlier_threshold, double inlier_ratio,
                                      unsigned long long seed, float T[16], int *success, int *n_corr_out,
                                      int *n_inliers_out, std::string *err);

// voxel.hip (a separate workspace instance is used: buffer slots differ from icp.hip's)
int voxel_grid(IcpWorkspace *ws, hipStream_t stream, const void *in, int n, int stride, float leaf,
               void *out, int out_capacity, int *n_out, std::string *err);
int voxel_grid_to_device(IcpWorkspace *ws, hipStream_t stream, const void *in, int n, int stride, float leaf,
                         const void **d_result, int *n_out, std::string *err);
int assemble_submap(IcpWorkspace *ws, hipStream_t stream, const void *const *clouds, const int *counts,
                    const float *transforms, int n_clouds, int stride, float leaf, void *out, int out_capacity,
                    int *n_out, std::string *err);

// the same for n_jobs submaps at once, every step one launch over all of them; job j = clouds [first[j], first[j + 1]) (device pointers)
int assemble_submaps_batch(IcpWorkspace *ws, hipStream_t stream, const void *const *clouds, const int *counts, const float *transforms,
                           const int *first, int n_jobs, int stride, float leaf, const void **d_results, int *n_out, std::string *err);

int assemble_submap_ex(IcpWorkspace *ws, hipStream_t stream, const void *const *clouds, const int *counts,
                       const float *transforms, int n_clouds, int stride, float leaf, bool clouds_on_device,
                       void *out, int out_capacity, const void **d_result, int *n_out, std::string *err);

}  // namespace scl
