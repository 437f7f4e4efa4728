// pose_kernels.hip — ModelPose (single camera pose) forward/backward and the element-wise helper
// functions of /root/reference/src/model.py for gfx950.
//
//   ModelPose.forward / criterion   model.py:98-127
//   to_camera_frame                 model.py:50-57   (exact f32 op order of the reference)
//   get_dist_mask / get_fov_mask    model.py:13-47
#include <type_traits>

#include "common.hpp"

// k_bwd_finish2 (the quaternion chain) comes from traj_kernels.hip (same translation unit, see trajopt_hip.hip)

// the pose's camera record: F.normalize (model.py:53), m = R(q/|q|)^T
__global__ void k_prep_posecam(const float* __restrict__ trans, const float* __restrict__ quat, WayHot* __restrict__ hot,
                               WayCold* __restrict__ cold) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float q[4] = {quat[0], quat[1], quat[2], quat[3]};
    float ss = q[0] * q[0];
    ss = ss + q[1] * q[1];
    ss = ss + q[2] * q[2];
    ss = ss + q[3] * q[3];
    float n = sqrtf(ss);
    n = n < 1e-12f ? 1e-12f : n;
    for (int i = 0; i < 4; ++i) q[i] = q[i] / n;
    WayCold cd;
    for (int i = 0; i < 4; ++i) cd.qn[i] = q[i];
    cd.nrm = n;
    cd.pad[0] = cd.pad[1] = cd.pad[2] = 0.f;
    cold[0] = cd;
    float R[9];
    quat_to_R(q, R);
    WayHot h;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) h.m[3 * i + j] = R[3 * j + i];
    h.t[0] = trans[0]; h.t[1] = trans[1]; h.t[2] = trans[2];
    h.a = 0.f; h.invM = 1.f; h.thr = INFINITY; h.sthr = INFINITY;
    hot[0] = h;
}

// observations + block partial sums (grid-stride, one point per lane per step: streaming 12 B in, 4 B out)
template <bool PINHOLE>
__global__ void __launch_bounds__(TO_BLOCK)
k_pose_fwd(CloudView cv, const WayHot* __restrict__ hot, CamConsts cc, const float* __restrict__ mask,
           float* __restrict__ obs, double* __restrict__ part) {
    __shared__ double lds[TO_BLOCK];
    const WayHot h = hot[0];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < cv.n; i += stride) {
        float X, Y, Z, y0, y1, y2;
        to_cam(h, cv.soa[i], cv.soa[cv.npad + i], cv.soa[2 * cv.npad + i], X, Y, Z, y0, y1, y2);
        float p = soft_vis<PINHOLE>(cc, X, Y, Z, nullptr);
        const int o = cv.perm[i];  // the caller's point order
        if (mask) p = mask[o] * p;  // model.py:115
        obs[o] = p;
        s += (double)p;
    }
    const double tot = block_sum_double(s, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(TO_BLOCK)
k_pose_fwd_finish(const double* __restrict__ part, int nparts, float eps, float* __restrict__ scalars) {
    __shared__ double lds[TO_BLOCK];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += TO_BLOCK) s += part[i];
    const double tot = block_sum_double(s, lds);
    if (threadIdx.x == 0) {
        const float sum = (float)tot;
        scalars[0] = sum;
        scalars[1] = 1.0f / (sum + eps);  // model.py:126
    }
}

// dL/d obs_n = -loss^2 * gout (model.py:126); 12 sums per block: sum w g [3], sum w y (x) g [9]
template <bool PINHOLE>
__global__ void __launch_bounds__(TO_BLOCK)
k_pose_bwd(CloudView cv, const WayHot* __restrict__ hot, CamConsts cc, const float* __restrict__ mask,
           const float* __restrict__ grad_obs, const float* __restrict__ scalars, const float* __restrict__ gout,
           double* __restrict__ part) {
    __shared__ double lds[TO_BLOCK];
    const WayHot h = hot[0];
    // dL/d obs_n: a caller-supplied vector (general criterion), else the fused loss 1/(sum+eps)
    const float coef = grad_obs ? 0.f : -scalars[1] * scalars[1] * gout[0];
    double acc[12];
    for (int k = 0; k < 12; ++k) acc[k] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < cv.n; i += stride) {
        float X, Y, Z, y0, y1, y2;
        to_cam(h, cv.soa[i], cv.soa[cv.npad + i], cv.soa[2 * cv.npad + i], X, Y, Z, y0, y1, y2);
        Vis s;
        soft_vis<PINHOLE>(cc, X, Y, Z, &s);
        float g[3];
        dvis_dc<PINHOLE>(cc, X, Y, Z, s, g);
        const int o = cv.perm[i];
        const float go = grad_obs ? grad_obs[o] : coef;
        const float wgt = mask ? go * mask[o] : go;
        const float w0 = wgt * g[0], w1 = wgt * g[1], w2 = wgt * g[2];
        acc[0] += w0; acc[1] += w1; acc[2] += w2;
        acc[3] += y0 * w0; acc[4] += y0 * w1; acc[5] += y0 * w2;
        acc[6] += y1 * w0; acc[7] += y1 * w1; acc[8] += y1 * w2;
        acc[9] += y2 * w0; acc[10] += y2 * w1; acc[11] += y2 * w2;
    }
    for (int k = 0; k < 12; ++k) {
        const double r = block_sum_double(acc[k], lds);
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * 12 + k] = r;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(TO_BLOCK)
k_pose_bwd_finish(const double* __restrict__ part, int nparts, float* __restrict__ vgrad) {
    __shared__ double lds[TO_BLOCK];
    for (int k = 0; k < 12; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nparts; i += TO_BLOCK) s += part[(int64_t)i * 12 + k];
        const double r = block_sum_double(s, lds);
        if (threadIdx.x == 0) vgrad[k] = (float)r;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// to_camera_frame / ego_to_cam_torch with the reference's exact f32 arithmetic: (x - t), then
// quaternion_apply(q_inv, .) = two raw Hamilton products, every term rounded, left to right
// (no fma: this translation unit is built with -ffp-contract=off).

struct ExactPose {
    float q[4], t[3];
};

__device__ __forceinline__ ExactPose exact_pose(const float* __restrict__ quat, const float* __restrict__ trans, int normalize) {
    ExactPose e;
    float q[4] = {quat[0], quat[1], quat[2], quat[3]};
    if (normalize) {
        float ss = q[0] * q[0];
        ss = ss + q[1] * q[1];
        ss = ss + q[2] * q[2];
        ss = ss + q[3] * q[3];
        float nn = sqrtf(ss);
        nn = nn < 1e-12f ? 1e-12f : nn;
        for (int i = 0; i < 4; ++i) q[i] = q[i] / nn;
    }
    for (int i = 0; i < 4; ++i) e.q[i] = q[i];
    e.t[0] = trans[0]; e.t[1] = trans[1]; e.t[2] = trans[2];
    return e;
}

__device__ __forceinline__ void exact_to_cam(const ExactPose& e, float x, float y, float z, float& r0, float& r1, float& r2) {
    const float aw = e.q[0], ax = -e.q[1], ay = -e.q[2], az = -e.q[3];
    const float cw = e.q[0], cx = e.q[1], cy = e.q[2], cz = e.q[3];
    const float bx = x - e.t[0], by = y - e.t[1], bz = z - e.t[2];
    const float bw = 0.f;
    const float ow = aw * bw - ax * bx - ay * by - az * bz;
    const float ox = aw * bx + ax * bw + ay * bz - az * by;
    const float oy = aw * by - ax * bz + ay * bw + az * bx;
    const float oz = aw * bz + ax * by - ay * bx + az * bw;
    r0 = ow * cx + ox * cw + oy * cz - oz * cy;
    r1 = ow * cy - ox * cz + oy * cw + oz * cx;
    r2 = ow * cz + ox * cy - oy * cx + oz * cw;
}

__global__ void __launch_bounds__(TO_BLOCK)
k_to_camera_frame(const float* __restrict__ xyz, int64_t n, const float* __restrict__ quat,
                  const float* __restrict__ trans, int normalize, int out_layout, float* __restrict__ out) {
    const ExactPose e = exact_pose(quat, trans, normalize);
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) {
        float r0, r1, r2;
        exact_to_cam(e, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], r0, r1, r2);
        if (out_layout == 0) {
            out[3 * i] = r0; out[3 * i + 1] = r1; out[3 * i + 2] = r2;
        } else {
            out[i] = r0; out[n + i] = r1; out[2 * n + i] = r2;
        }
    }
}

// get_dist_mask and soft get_fov_mask as separate outputs (reference op order; IEEE divisions)
__global__ void __launch_bounds__(TO_BLOCK)
k_soft_masks(const float* __restrict__ cam_xyz, int64_t n, CamConsts cc, float std_, float img_w, float img_h,
             float* __restrict__ dist_mask, float* __restrict__ fov_mask) {
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) {
        const float X = cam_xyz[3 * i], Y = cam_xyz[3 * i + 1], Z = cam_xyz[3 * i + 2];
        if (dist_mask) {
            const float dx = X - cc.mean, dy = Y - cc.mean, dz = Z - cc.mean;
            const float dist = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
            const float ds = dist / std_;
            dist_mask[i] = expf(-0.5f * (ds * ds));
        }
        if (fov_mask) {
            const float h0 = fmaf(cc.k[2], Z, fmaf(cc.k[1], Y, cc.k[0] * X));
            const float h1 = fmaf(cc.k[5], Z, fmaf(cc.k[4], Y, cc.k[3] * X));
            const float h2 = fmaf(cc.k[8], Z, fmaf(cc.k[7], Y, cc.k[6] * X));
            const float S = 1.0f / (1.0f + expf(-h2));
            const float z = h2 + cc.eps;
            const float au = (h0 / z - cc.halfw) / img_w, av = (h1 / z - cc.halfh) / img_h;
            const float Gw = expf(-0.5f * (au * au)), Gh = expf(-0.5f * (av * av));
            fov_mask[i] = S * Gw * Gh;
        }
    }
}

// Backward of the two masks w.r.t. the camera-frame points: grad_xyz[n] = g_dist[n] dD/dp + g_fov[n] dF/dp (either upstream
// gradient may be absent).  D = exp(-|p - mean|^2 / (2 std^2)); F = S Gw Gh, ln F = ln S - au^2/2 - av^2/2 with
// au = (h0/z - W/2)/W, av = (h1/z - H/2)/H, z = h2 + eps, h = K p  (model.py:13-47).
__global__ void __launch_bounds__(TO_BLOCK)
k_soft_masks_bwd(const float* __restrict__ cam_xyz, int64_t n, CamConsts cc, float std_, float img_w, float img_h,
                 const float* __restrict__ g_dist, const float* __restrict__ g_fov, float* __restrict__ grad_xyz) {
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) {
        const float X = cam_xyz[3 * i], Y = cam_xyz[3 * i + 1], Z = cam_xyz[3 * i + 2];
        float g[3] = {0.f, 0.f, 0.f};
        if (g_dist) {
            const float d[3] = {X - cc.mean, Y - cc.mean, Z - cc.mean};
            const float dist = sqrtf(fmaf(d[2], d[2], fmaf(d[1], d[1], d[0] * d[0])));
            const float ds = dist / std_;
            const float D = expf(-0.5f * (ds * ds));
            const float w = -g_dist[i] * D / (std_ * std_);
            for (int k = 0; k < 3; ++k) g[k] = w * d[k];
        }
        if (g_fov) {
            const float h0 = fmaf(cc.k[2], Z, fmaf(cc.k[1], Y, cc.k[0] * X));
            const float h1 = fmaf(cc.k[5], Z, fmaf(cc.k[4], Y, cc.k[3] * X));
            const float h2 = fmaf(cc.k[8], Z, fmaf(cc.k[7], Y, cc.k[6] * X));
            const float S = 1.0f / (1.0f + expf(-h2));
            const float z = h2 + cc.eps, rz = 1.0f / z;
            const float u = h0 * rz, v = h1 * rz;
            const float au = (u - cc.halfw) / img_w, av = (v - cc.halfh) / img_h;
            const float F = S * expf(-0.5f * (au * au)) * expf(-0.5f * (av * av));
            const float gf = g_fov[i] * F;
            const float cu = -au / img_w * rz, cv = -av / img_h * rz;      // d(-au^2/2)/dh0, d(-av^2/2)/dh1
            const float c2 = (1.0f - S) - (cu * u + cv * v);               // d ln F / d h2
            if (F > 0.f)
                for (int k = 0; k < 3; ++k) g[k] += gf * (cu * cc.k[k] + cv * cc.k[3 + k] + c2 * cc.k[6 + k]);
        }
        grad_xyz[3 * i] = g[0]; grad_xyz[3 * i + 1] = g[1]; grad_xyz[3 * i + 2] = g[2];
    }
}

// Backward of to_camera_frame, c = R(q/|q|)^T (x - t): grad_xyz[n] = R g_n; per block the 12 sums (sum g, sum y (x) g) that
// k_pose_bwd_finish + k_bwd_finish2 turn into the translation and quaternion gradients.
__global__ void __launch_bounds__(TO_BLOCK)
k_to_camera_frame_bwd(const float* __restrict__ xyz, int64_t n, const WayHot* __restrict__ hot, const float* __restrict__ g_out,
                      float* __restrict__ grad_xyz, double* __restrict__ part) {
    __shared__ double lds[TO_BLOCK];
    const WayHot h = hot[0];
    double acc[12];
    for (int k = 0; k < 12; ++k) acc[k] = 0.0;
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) {
        const float y0 = xyz[3 * i] - h.t[0], y1 = xyz[3 * i + 1] - h.t[1], y2 = xyz[3 * i + 2] - h.t[2];
        const float g0 = g_out[3 * i], g1 = g_out[3 * i + 1], g2 = g_out[3 * i + 2];
        if (grad_xyz) {   // dL/dx = R g,  R[j][i] = m[3*i+j]
            grad_xyz[3 * i] = fmaf(h.m[6], g2, fmaf(h.m[3], g1, h.m[0] * g0));
            grad_xyz[3 * i + 1] = fmaf(h.m[7], g2, fmaf(h.m[4], g1, h.m[1] * g0));
            grad_xyz[3 * i + 2] = fmaf(h.m[8], g2, fmaf(h.m[5], g1, h.m[2] * g0));
        }
        acc[0] += g0; acc[1] += g1; acc[2] += g2;
        acc[3] += y0 * g0; acc[4] += y0 * g1; acc[5] += y0 * g2;
        acc[6] += y1 * g0; acc[7] += y1 * g1; acc[8] += y1 * g2;
        acc[9] += y2 * g0; acc[10] += y2 * g1; acc[11] += y2 * g2;
    }
    for (int k = 0; k < 12; ++k) {
        const double r = block_sum_double(acc[k], lds);
        if (threadIdx.x == 0) part[(int64_t)blockIdx.x * 12 + k] = r;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
namespace {
constexpr int kPoseBlocks = 1024;
struct PosePlan {
    size_t off_hot, off_cold, off_part, off_vgrad, total;
};
inline PosePlan pose_plan() {
    PosePlan p;
    size_t o = 0;
    p.off_hot = o;   o += align_up(sizeof(WayHot), 256);
    p.off_cold = o;  o += align_up(sizeof(WayCold), 256);
    p.off_part = o;  o += align_up((size_t)kPoseBlocks * 12 * sizeof(double), 256);
    p.off_vgrad = o; o += align_up(12 * sizeof(float), 256);
    p.total = o;
    return p;
}
inline int pose_blocks(int64_t n) {
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    return (int)(nb > kPoseBlocks ? kPoseBlocks : nb);
}
}  // namespace

extern "C" size_t tohip_pose_workspace_bytes(int64_t n_points) {
    (void)n_points;
    return pose_plan().total;
}

extern "C" int tohip_pose_forward(const void* packed, int64_t n, const float* trans, const float* quat,
                                  const tohip_camera* cam, const float* mask, float* obs, float* scalars,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    if (!packed || !trans || !quat || !cam || !obs || !scalars || !workspace || n <= 0) return TOHIP_EINVAL;
    const PosePlan pl = pose_plan();
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    WayHot* hot = (WayHot*)(ws + pl.off_hot);
    WayCold* cold = (WayCold*)(ws + pl.off_cold);
    double* part = (double*)(ws + pl.off_part);
    const CamConsts cc = make_consts(cam);
    const CloudView cv = cloud_view(packed, n);
    k_prep_posecam<<<1, 64, 0, st>>>(trans, quat, hot, cold);
    TO_HIP_CHECK_LAUNCH();
    const int nb = pose_blocks(n);
    if (cc.pinhole) k_pose_fwd<true><<<nb, TO_BLOCK, 0, st>>>(cv, hot, cc, mask, obs, part);
    else k_pose_fwd<false><<<nb, TO_BLOCK, 0, st>>>(cv, hot, cc, mask, obs, part);
    TO_HIP_CHECK_LAUNCH();
    k_pose_fwd_finish<<<1, TO_BLOCK, 0, st>>>(part, nb, cam->eps, scalars);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_pose_backward(const void* packed, int64_t n, const float* trans, const float* quat,
                                   const tohip_camera* cam, const float* mask, const float* grad_obs,
                                   const float* scalars, const float* gout, float* trans_grad, float* quat_grad,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    if (!packed || !trans || !quat || !cam || !trans_grad || !quat_grad || !workspace || n <= 0 ||
        (!grad_obs && (!scalars || !gout)))
        return TOHIP_EINVAL;
    const PosePlan pl = pose_plan();
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    WayHot* hot = (WayHot*)(ws + pl.off_hot);
    WayCold* cold = (WayCold*)(ws + pl.off_cold);
    double* part = (double*)(ws + pl.off_part);
    float* vgrad = (float*)(ws + pl.off_vgrad);
    const CamConsts cc = make_consts(cam);
    const CloudView cv = cloud_view(packed, n);
    k_prep_posecam<<<1, 64, 0, st>>>(trans, quat, hot, cold);
    TO_HIP_CHECK_LAUNCH();
    const int nb = pose_blocks(n);
    if (cc.pinhole) k_pose_bwd<true><<<nb, TO_BLOCK, 0, st>>>(cv, hot, cc, mask, grad_obs, scalars, gout, part);
    else k_pose_bwd<false><<<nb, TO_BLOCK, 0, st>>>(cv, hot, cc, mask, grad_obs, scalars, gout, part);
    TO_HIP_CHECK_LAUNCH();
    k_pose_bwd_finish<<<1, TO_BLOCK, 0, st>>>(part, nb, vgrad);
    TO_HIP_CHECK_LAUNCH();
    k_bwd_finish2<<<1, 64, 0, st>>>(vgrad, hot, cold, 1, 1, nullptr, nullptr, trans_grad, quat_grad);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_to_camera_frame(const float* xyz, int64_t n, const float* quat, const float* trans, int normalize,
                                     int out_layout, float* out, void* stream_) {
    if (!xyz || !quat || !trans || !out || n < 0) return TOHIP_EINVAL;
    if (n == 0) return TOHIP_OK;
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    if (nb > 4096) nb = 4096;
    k_to_camera_frame<<<(int)nb, TO_BLOCK, 0, (hipStream_t)stream_>>>(xyz, n, quat, trans, normalize, out_layout, out);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_soft_masks(const float* cam_xyz, int64_t n, const tohip_camera* cam, float* dist_mask,
                                float* fov_mask, void* stream_) {
    if (!cam_xyz || !cam || n < 0) return TOHIP_EINVAL;
    if (n == 0) return TOHIP_OK;
    const CamConsts cc = make_consts(cam);
    const float std_ = (float)(((double)cam->max_dist - (double)cam->min_dist) / 2.0);
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    if (nb > 4096) nb = 4096;
    k_soft_masks<<<(int)nb, TO_BLOCK, 0, (hipStream_t)stream_>>>(cam_xyz, n, cc, std_, cam->img_width, cam->img_height,
                                                                 dist_mask, fov_mask);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_soft_masks_backward(const float* cam_xyz, int64_t n, const tohip_camera* cam, const float* grad_dist,
                                         const float* grad_fov, float* grad_xyz, void* stream_) {
    if (!cam_xyz || !cam || !grad_xyz || n < 0) return TOHIP_EINVAL;
    if (n == 0) return TOHIP_OK;
    const CamConsts cc = make_consts(cam);
    const float std_ = (float)(((double)cam->max_dist - (double)cam->min_dist) / 2.0);
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    if (nb > 4096) nb = 4096;
    k_soft_masks_bwd<<<(int)nb, TO_BLOCK, 0, (hipStream_t)stream_>>>(cam_xyz, n, cc, std_, cam->img_width, cam->img_height, grad_dist,
                                                                     grad_fov, grad_xyz);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_to_camera_frame_backward(const float* xyz, int64_t n, const float* quat, const float* trans,
                                              const float* grad_out, float* grad_xyz, float* grad_quat, float* grad_trans,
                                              void* workspace, size_t workspace_bytes, void* stream_) {
    if (!xyz || !quat || !trans || !grad_out || !grad_quat || !grad_trans || !workspace || n <= 0) return TOHIP_EINVAL;
    const PosePlan pl = pose_plan();
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    WayHot* hot = (WayHot*)(ws + pl.off_hot);
    WayCold* cold = (WayCold*)(ws + pl.off_cold);
    double* part = (double*)(ws + pl.off_part);
    float* vgrad = (float*)(ws + pl.off_vgrad);
    k_prep_posecam<<<1, 64, 0, st>>>(trans, quat, hot, cold);
    TO_HIP_CHECK_LAUNCH();
    const int nb = pose_blocks(n);
    k_to_camera_frame_bwd<<<nb, TO_BLOCK, 0, st>>>(xyz, n, hot, grad_out, grad_xyz, part);
    TO_HIP_CHECK_LAUNCH();
    k_pose_bwd_finish<<<1, TO_BLOCK, 0, st>>>(part, nb, vgrad);
    TO_HIP_CHECK_LAUNCH();
    k_bwd_finish2<<<1, 64, 0, st>>>(vgrad, hot, cold, 1, 1, nullptr, nullptr, grad_trans, grad_quat);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}
