// hull_kernels.hip — hidden-point removal (Katz): spherical flip + convex hull on the GPU.
//   sphericalFlip / convexHull / hidden_pts_removal   /root/reference/src/tools.py:38-85
#include "common.hpp"

extern "C" size_t tohip_hpr_workspace_bytes(int64_t n) {
    (void)n;
    return 256;
}

extern "C" int tohip_spherical_flip(const float* xyz, int64_t n, float param, float* flipped, float* radius_out,
                                    void* workspace, size_t workspace_bytes, void* stream_) {
    if (!xyz || !flipped || !workspace || n <= 0) return TOHIP_EINVAL;
    if (workspace_bytes < 256) return TOHIP_ENOSPC;
    return launch_flip(xyz, n, param, flipped, radius_out, (int*)workspace, (hipStream_t)stream_);
}

extern "C" int tohip_hidden_pts_removal(const float* xyz, int64_t n, float param, int32_t* visible_idx,
                                        int32_t* visible_count, float* mask, void* workspace, size_t workspace_bytes,
                                        void* stream_) {
    (void)xyz; (void)n; (void)param; (void)visible_idx; (void)visible_count; (void)mask; (void)workspace;
    (void)workspace_bytes; (void)stream_;
    return TOHIP_EINVAL;  // hull construction lands in the next commit
}
