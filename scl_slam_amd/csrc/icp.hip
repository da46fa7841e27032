// icp.hip -- placeholder until the ICP kernels land (next milestone of this round).
#include "icp.hpp"

namespace scl {

void icp_workspace_free(IcpWorkspace *ws)
{
    for (int i = 0; i < 16; ++i) if (ws->buf[i]) { (void)hipFree(ws->buf[i]); ws->buf[i] = nullptr; ws->cap[i] = 0; }
    if (ws->pinned) { (void)hipHostFree(ws->pinned); ws->pinned = nullptr; ws->pinned_cap = 0; }
}

int icp_align(IcpWorkspace *, hipStream_t, int, const void *, int, const void *, int, int, const scl_icp_params &,
              float *, float *, int *, int *, std::string *err)
{ if (err) *err = "icp_align: not implemented yet"; return SCL_ERR_UNSUPPORTED; }
int icp_nn_correspondences(IcpWorkspace *, hipStream_t, int, const void *, int, const void *, int, int, int *, float *,
                           std::string *err)
{ if (err) *err = "nn_correspondences: not implemented yet"; return SCL_ERR_UNSUPPORTED; }
int icp_rigid_svd(IcpWorkspace *, hipStream_t, int, const void *, int, const void *, int, int, const int *, const int *,
                  int, float *, std::string *err)
{ if (err) *err = "rigid_svd: not implemented yet"; return SCL_ERR_UNSUPPORTED; }
int icp_transform_cloud(IcpWorkspace *, hipStream_t, const void *, int, int, const float *, void *, std::string *err)
{ if (err) *err = "transform_cloud: not implemented yet"; return SCL_ERR_UNSUPPORTED; }

}  // namespace scl
