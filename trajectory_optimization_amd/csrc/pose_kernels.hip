// pose_kernels.hip — ModelPose (single camera pose) forward/backward and the element-wise helper
// functions of /root/reference/src/model.py for gfx950.
//
//   ModelPose.forward / criterion   model.py:98-127
//   to_camera_frame                 model.py:50-57   (exact f32 op order of the reference)
//   get_dist_mask / get_fov_mask    model.py:13-47
#include <type_traits>

#include "common.hpp"

// k_bwd_finish2 (the quaternion chain), prep_wayrec and the packed evaluation come from traj_kernels.hip / common.hpp (same
// translation unit, see trajopt_hip.hip)

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

// ---------------------------------------------------------------------------------------------
// ModelPose: ONE streaming pass over the cloud for everything a step needs.
//
//   observations[n] = dist_mask * fov_mask (* occlusion mask)      model.py:98-122     12 B read + 4 B written per point
//   sum -> loss = 1 / (sum + eps)                                  model.py:124-127
//   gradient sums  sum_n w_n dp_n/dy  and  sum_n y_n (x) w_n dp_n/dy  (y = x - t, world-aligned)
//
// The gradient of the loss is -loss^2 x (sums taken with w_n = mask_n): the sums do not depend on the loss, so the pass that
// computes the observations takes them too and the finish kernel scales them once the sum is known — loss.backward() then costs
// no pass over the cloud at all.  A lane owns EIGHT consecutive points (two 16-byte loads per coordinate plane, four packed
// evaluations: the arithmetic of traj_kernels.hip's pair kernel, two points per instruction); the observations leave as two
// 16-byte stores per lane when the cloud keeps the caller's order (a pose-only cloud is packed unsorted: nothing here culls) and
// through the permutation otherwise.  Per-lane sums in f32 (a lane sees a few dozen points), one DPP tree per sum and wave, the
// waves' and the blocks' totals in f64, in a fixed order: bitwise reproducible.
#define TO_POSE_PTS 8                       // points per lane
#define TO_POSE_CHUNK (TO_BLOCK * TO_POSE_PTS)
#define TO_POSE_NSUM 13                     // sum of the observations + 12 gradient sums
#define TO_POSE_MAXBLOCKS 1024

struct PoseArgs {
    CloudView cv;
    const float* trans;
    const float* quat;
    EvalK k;
    const float* mask;       // caller's order, may be NULL
    const float* grad_obs;   // caller's order, may be NULL: dL/d observations (general upstream); NULL: unit weights
    float* obs;              // caller's order (FWD)
    double* part;            // gridDim.x x 16 doubles
};

// the camera record of the pose in LDS (prep_wayrec: F.normalize, R, the projection rows) — every block builds its own: no launch
__device__ __forceinline__ void pose_record(const PoseArgs& a, WayRec* srec, WayCold* scold) {
    if (threadIdx.x == 0) prep_wayrec(0, a.trans, a.quat, 1, nullptr, nullptr, a.k, srec, scold, nullptr, 1);
    __syncthreads();
}

template <bool FWD, bool GRAD>
__global__ void __launch_bounds__(TO_BLOCK) k_pose_stream(PoseArgs a) {
    __shared__ WayRec srec;
    __shared__ WayCold scold;
    __shared__ float swave[TO_WAVES_PER_BLOCK][16];
    pose_record(a, &srec, &scold);
    // the record's first line as scalars: a packed instruction takes one scalar operand
    WayRec r;
    {
        const float* src = reinterpret_cast<const float*>(&srec);
        float* dst = reinterpret_cast<float*>(&r);
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, src[i])));
    }
    const EvalK& k = a.k;
    const bool ident = a.cv.hdr[0] == 0;   // the points keep the caller's order: observations, masks and upstream gradients are read / written in place
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t n = a.cv.n, npad = a.cv.npad;
    float asum = 0.f;
    f2 acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = pk_splat(0.f);
    const int64_t nchunks = npad / TO_POSE_CHUNK;
    // two blocks to a CU, each striding over the chunks: the next chunk's points are requested before this one is evaluated
    float nx[TO_POSE_PTS], ny[TO_POSE_PTS], nz[TO_POSE_PTS];
    if ((int64_t)blockIdx.x < nchunks) load_points<TO_POSE_PTS>(a.cv.soa, npad, (int64_t)blockIdx.x * TO_POSE_CHUNK + (int64_t)t * TO_POSE_PTS, nx, ny, nz);
    for (int64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int64_t base = c * TO_POSE_CHUNK + (int64_t)t * TO_POSE_PTS;
        float x[TO_POSE_PTS], y[TO_POSE_PTS], z[TO_POSE_PTS];
#pragma unroll
        for (int i = 0; i < TO_POSE_PTS; ++i) { x[i] = nx[i]; y[i] = ny[i]; z[i] = nz[i]; }
        if (c + gridDim.x < nchunks) load_points<TO_POSE_PTS>(a.cv.soa, npad, (c + gridDim.x) * TO_POSE_CHUNK + (int64_t)t * TO_POSE_PTS, nx, ny, nz);
        if (base >= n) continue;   // pads only
        const bool whole = base + TO_POSE_PTS <= n && ident;   // eight real points, in the caller's order: 16-byte accesses
        int o[TO_POSE_PTS];
        if (!ident) {
            const int4 p0 = *reinterpret_cast<const int4*>(a.cv.perm + base), p1 = *reinterpret_cast<const int4*>(a.cv.perm + base + 4);
            o[0] = p0.x; o[1] = p0.y; o[2] = p0.z; o[3] = p0.w; o[4] = p1.x; o[5] = p1.y; o[6] = p1.z; o[7] = p1.w;
        } else {
#pragma unroll
            for (int i = 0; i < TO_POSE_PTS; ++i) o[i] = (int)(base + i);
        }
        float w[TO_POSE_PTS];   // what multiplies p in the observation (the occlusion mask), 0 for pads
#pragma unroll
        for (int i = 0; i < TO_POSE_PTS; ++i) w[i] = base + i < n ? 1.0f : 0.f;
        if (a.mask != nullptr) {
            if (whole && ((((uintptr_t)a.mask) & 15) == 0)) {
                const float4 m0 = *reinterpret_cast<const float4*>(a.mask + base), m1 = *reinterpret_cast<const float4*>(a.mask + base + 4);
                w[0] = m0.x; w[1] = m0.y; w[2] = m0.z; w[3] = m0.w; w[4] = m1.x; w[5] = m1.y; w[6] = m1.z; w[7] = m1.w;
            } else {
#pragma unroll
                for (int i = 0; i < TO_POSE_PTS; ++i)
                    if (base + i < n) w[i] = a.mask[o[i]];
            }
        }
        float gw[TO_POSE_PTS];   // GRAD: the weight of dp/dy — the mask, times the upstream gradient where one is given
#pragma unroll
        for (int i = 0; i < TO_POSE_PTS; ++i) gw[i] = w[i];
        if (GRAD && a.grad_obs != nullptr) {
#pragma unroll
            for (int i = 0; i < TO_POSE_PTS; ++i) gw[i] = base + i < n ? a.grad_obs[o[i]] * w[i] : 0.f;
        }
        float ob[TO_POSE_PTS];
#pragma unroll
        for (int i = 0; i < TO_POSE_PTS; i += 2) {
            VisGrad2 vg;
            const f2 p = vis_p_pk_grad(r, k, f2{x[i], x[i + 1]}, f2{y[i], y[i + 1]}, f2{z[i], z[i + 1]}, vg);
            const f2 obs2 = f2{w[i], w[i + 1]} * p;   // model.py:115: mask * observations
            ob[i] = obs2.x; ob[i + 1] = obs2.y;
            asum += obs2.x;
            asum += obs2.y;
            if (GRAD) {
                f2 g[3];
                dvis_dy_pk(r, k, p, vg, g);
                const f2 wg = f2{gw[i], gw[i + 1]};
                const f2 w0 = wg * g[0], w1 = wg * g[1], w2 = wg * g[2];
                acc[0] = acc[0] + w0; acc[1] = acc[1] + w1; acc[2] = acc[2] + w2;
                acc[3] = pk_fma(vg.y0, w0, acc[3]); acc[4] = pk_fma(vg.y0, w1, acc[4]); acc[5] = pk_fma(vg.y0, w2, acc[5]);
                acc[6] = pk_fma(vg.y1, w0, acc[6]); acc[7] = pk_fma(vg.y1, w1, acc[7]); acc[8] = pk_fma(vg.y1, w2, acc[8]);
                acc[9] = pk_fma(vg.y2, w0, acc[9]); acc[10] = pk_fma(vg.y2, w1, acc[10]); acc[11] = pk_fma(vg.y2, w2, acc[11]);
            }
        }
        if (FWD) {
            if (whole && ((((uintptr_t)a.obs) & 15) == 0)) {
                __builtin_nontemporal_store(f4v{ob[0], ob[1], ob[2], ob[3]}, reinterpret_cast<f4v*>(a.obs + base));
                __builtin_nontemporal_store(f4v{ob[4], ob[5], ob[6], ob[7]}, reinterpret_cast<f4v*>(a.obs + base + 4));
            } else {
#pragma unroll
                for (int i = 0; i < TO_POSE_PTS; ++i)
                    if (base + i < n) a.obs[o[i]] = ob[i];
            }
        }
    }
    // per wave one DPP tree per sum (valid in lane 63), then the block's four waves in order, in f64
    float sums[TO_POSE_NSUM];
    sums[0] = wave_sum63(asum);
    if (GRAD) {
#pragma unroll
        for (int j = 0; j < 12; ++j) sums[1 + j] = wave_sum63(acc[j].x + acc[j].y);
    }
    if (lane == 63) {
#pragma unroll
        for (int j = 0; j < (GRAD ? TO_POSE_NSUM : 1); ++j) swave[wave][j] = sums[j];
    }
    __syncthreads();
    if (t < (GRAD ? TO_POSE_NSUM : 1)) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < TO_WAVES_PER_BLOCK; ++wv) s += (double)swave[wv][t];
        a.part[(int64_t)blockIdx.x * 16 + t] = s;
    }
}

// what follows the pass: the blocks' totals -> scalars (sum, loss), and — GRAD — the gradient sums scaled, rotated into the camera
// frame and chained to (position, raw quaternion).  One block.
//   coef_mode 0: the sums carry their upstream gradient already (grad_obs)       1: x -loss^2 gout[0] (scalars_in[1] = loss)
//             2: x -loss^2 gout[0] with the loss of THIS pass (forward + backward in one call; gout NULL: 1)
struct PoseFinish {
    const double* part;
    int nparts;
    const float *trans, *quat;
    EvalK k;
    float eps;
    float* scalars_out;        // may be NULL
    const float* scalars_in;   // coef_mode 1
    const float* gout;
    int coef_mode, grad;
    float *trans_grad, *quat_grad;
    // a device-resident optimisation step (optimizer.optimize_pose): Adam on (trans, quat) with the gradient just computed
    int adam;
    float *mt, *vt, *mq, *vq;
    float lr_pose, lr_quat, beta1, beta2, adam_eps;
    int step;                  // 1-based
    float* loss_log;           // adam: loss_log[step - 1] = the loss of this step
};

__global__ void __launch_bounds__(1024) k_pose_finish(PoseFinish f) {
    __shared__ double stot[16];
    __shared__ double sred[1024 / 16][16];
    __shared__ WayRec srec;
    __shared__ WayCold scold;
    __shared__ float vgrad[12];
    const int t = threadIdx.x, col = t & 15, row = t >> 4;
    if (t == 0) prep_wayrec(0, f.trans, f.quat, 1, nullptr, nullptr, f.k, &srec, &scold, nullptr, 1);
    // column `col` of the partials, rows row, row + 16, ... in order; then the sixteen row groups in order (fixed: deterministic)
    double s = 0.0;
    const int nrows = (int)blockDim.x / 16;
    for (int i = row; i < f.nparts; i += nrows) s += f.part[(int64_t)i * 16 + col];
    sred[row][col] = s;
    __syncthreads();
    if (t < 16) {
        double q = 0.0;
        for (int i = 0; i < nrows; ++i) q += sred[i][t];
        stot[t] = q;
    }
    __syncthreads();
    const float sum = (float)stot[0];
    const float loss = 1.0f / (sum + f.eps);   // model.py:126
    if (t == 0 && f.scalars_out) { f.scalars_out[0] = sum; f.scalars_out[1] = loss; f.scalars_out[2] = 0.f; f.scalars_out[3] = 0.f; }
    if (t == 0 && f.adam) f.loss_log[f.step - 1] = loss;
    if (!f.grad) return;
    double coef = 1.0;
    if (f.coef_mode == 1) coef = -(double)f.scalars_in[1] * (double)f.scalars_in[1] * (double)f.gout[0];
    if (f.coef_mode == 2) coef = -(double)loss * (double)loss * (double)(f.gout ? f.gout[0] : 1.0f);
    // c = m y: dL/dc = m gy, y (x) dL/dc = (y (x) gy) m^T   (k_traj_finish's last step)
    if (t < 12) {
        const WayRec& r = srec;
        double out;
        const double g0 = coef * stot[1], g1 = coef * stot[2], g2 = coef * stot[3];
        if (t < 3) out = (double)r.m[3 * t] * g0 + (double)r.m[3 * t + 1] * g1 + (double)r.m[3 * t + 2] * g2;
        else {
            const int j = (t - 3) / 3, i = (t - 3) % 3;
            out = coef * (stot[4 + 3 * j] * (double)r.m[3 * i] + stot[4 + 3 * j + 1] * (double)r.m[3 * i + 1] + stot[4 + 3 * j + 2] * (double)r.m[3 * i + 2]);
        }
        vgrad[t] = (float)out;
    }
    __syncthreads();
    if (t == 0) {
        float o[7];
        finish_waypoint(0, vgrad, RecRows{&srec}, &scold, 1, nullptr, nullptr, o);
        if (f.trans_grad) { f.trans_grad[0] = o[0]; f.trans_grad[1] = o[1]; f.trans_grad[2] = o[2]; }
        if (f.quat_grad) { f.quat_grad[0] = o[3]; f.quat_grad[1] = o[4]; f.quat_grad[2] = o[5]; f.quat_grad[3] = o[6]; }
        if (f.adam) {
            const AdamConsts cp = adam_consts(f.lr_pose, f.beta1, f.beta2, f.step), cq = adam_consts(f.lr_quat, f.beta1, f.beta2, f.step);
            for (int i = 0; i < 3; ++i) adam_apply(const_cast<float*>(f.trans), o[i], f.mt, f.vt, i, f.beta1, f.beta2, f.adam_eps, cp);
            for (int i = 0; i < 4; ++i) adam_apply(const_cast<float*>(f.quat), o[3 + i], f.mq, f.vq, i, f.beta1, f.beta2, f.adam_eps, cq);
        }
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
    // 12 B in, 8 B out per point: the memory system's job — provided the four exponentials, the square root and the four divisions
    // of the reference's expressions do not take a hundred instructions.  Hardware v_exp / v_rcp / v_sqrt with the argument's
    // rounding error folded back in (to_exp: 1.5 ulp) instead of libm's: 85 -> 67 us at 16 M points, results within 2 ulp.
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    const float inv_std = to_rcp(std_), inv_w = to_rcp(img_w), inv_h = to_rcp(img_h);
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) {
        const float X = cam_xyz[3 * i], Y = cam_xyz[3 * i + 1], Z = cam_xyz[3 * i + 2];
        if (dist_mask) {
            const float dx = X - cc.mean, dy = Y - cc.mean, dz = Z - cc.mean;
            const float ds = __builtin_sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx))) * inv_std;
            dist_mask[i] = to_exp(-0.5f * (ds * ds));
        }
        if (fov_mask) {
            const float h0 = fmaf(cc.k[2], Z, fmaf(cc.k[1], Y, cc.k[0] * X));
            const float h1 = fmaf(cc.k[5], Z, fmaf(cc.k[4], Y, cc.k[3] * X));
            const float h2 = fmaf(cc.k[8], Z, fmaf(cc.k[7], Y, cc.k[6] * X));
            const float S = to_rcp(1.0f + to_exp(-h2));
            const float rz = to_rcp(h2 + cc.eps);
            const float au = (h0 * rz - cc.halfw) * inv_w, av = (h1 * rz - cc.halfh) * inv_h;
            fov_mask[i] = S * to_exp(-0.5f * fmaf(au, au, av * av));
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
constexpr int kPoseBlocks = 1024;   // k_to_camera_frame_bwd's grid
struct PosePlan {
    size_t off_hot, off_cold, off_part, off_vgrad, total;
};
inline PosePlan pose_plan() {
    PosePlan p;
    size_t o = 0;
    p.off_hot = o;   o += align_up(sizeof(WayHot), 256);
    p.off_cold = o;  o += align_up(sizeof(WayCold), 256);
    p.off_part = o;  o += align_up((size_t)TO_POSE_MAXBLOCKS * 16 * sizeof(double), 256);   // (>= kPoseBlocks x 12 doubles)
    p.off_vgrad = o; o += align_up(12 * sizeof(float), 256);
    p.total = o;
    return p;
}
inline int pose_blocks(int64_t n) {
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    return (int)(nb > kPoseBlocks ? kPoseBlocks : nb);
}

// blocks of the streaming pass: one per 2048-point chunk while they are all resident at once (occupancy x CUs), else that many
// persistent ones striding over the chunks
template <bool FWD, bool GRAD>
inline int pose_stream_blocks(int64_t npad) {
    static const int resident = [] {
        int dev = 0, cus = 256, per = 4;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, k_pose_stream<FWD, GRAD>, TO_BLOCK, 0) != hipSuccess || per <= 0) per = 2;
        if (per > 2) per = 2;   // two blocks to a CU keep the chip's HBM pipes full with the next chunk's loads in flight (12 MB), and
                                // the finish has a quarter of the partials to add
        const int r = per * cus;
        return r > TO_POSE_MAXBLOCKS ? TO_POSE_MAXBLOCKS : r;
    }();
    const int64_t nchunks = npad / TO_POSE_CHUNK;
    return (int)(nchunks < resident ? nchunks : resident);
}

struct PoseCall {
    hipStream_t st;
    PoseArgs a;
    PoseFinish f;
};
inline int pose_call_init(PoseCall& c, const void* packed, int64_t n, const float* trans, const float* quat, const tohip_camera* cam,
                          const float* mask, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!packed || !trans || !quat || !cam || !workspace || n <= 0) return TOHIP_EINVAL;
    const PosePlan pl = pose_plan();
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    c.st = (hipStream_t)stream_;
    c.a.cv = cloud_view(packed, n);
    c.a.trans = trans; c.a.quat = quat; c.a.k = make_evalk(cam); c.a.mask = mask; c.a.grad_obs = nullptr; c.a.obs = nullptr;
    c.a.part = (double*)((char*)workspace + pl.off_part);
    PoseFinish& f = c.f;
    f.part = c.a.part; f.nparts = 0; f.trans = trans; f.quat = quat; f.k = c.a.k; f.eps = cam->eps;
    f.scalars_out = nullptr; f.scalars_in = nullptr; f.gout = nullptr; f.coef_mode = 0; f.grad = 0; f.trans_grad = nullptr; f.quat_grad = nullptr;
    f.adam = 0; f.mt = f.vt = f.mq = f.vq = nullptr; f.lr_pose = f.lr_quat = f.beta1 = f.beta2 = f.adam_eps = 0.f; f.step = 0; f.loss_log = nullptr;
    return TOHIP_OK;
}
template <bool FWD, bool GRAD>
inline int pose_launch(PoseCall& c) {
    const int nb = pose_stream_blocks<FWD, GRAD>(c.a.cv.npad);
    k_pose_stream<FWD, GRAD><<<nb, TO_BLOCK, 0, c.st>>>(c.a);
    TO_HIP_CHECK_LAUNCH();
    c.f.nparts = nb;
    c.f.grad = GRAD ? 1 : 0;
    k_pose_finish<<<1, nb > 64 ? 1024 : TO_BLOCK, 0, c.st>>>(c.f);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}
}  // namespace

extern "C" size_t tohip_pose_workspace_bytes(int64_t n_points) {
    (void)n_points;
    return pose_plan().total;
}

extern "C" int tohip_pose_forward(const void* packed, int64_t n, const float* trans, const float* quat,
                                  const tohip_camera* cam, const float* mask, float* obs, float* scalars,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
    if (!obs || !scalars) return TOHIP_EINVAL;
    PoseCall c;
    const int rc = pose_call_init(c, packed, n, trans, quat, cam, mask, workspace, workspace_bytes, stream_);
    if (rc != TOHIP_OK) return rc;
    c.a.obs = obs;
    c.f.scalars_out = scalars;
    return pose_launch<true, false>(c);
}

extern "C" int tohip_pose_backward(const void* packed, int64_t n, const float* trans, const float* quat,
                                   const tohip_camera* cam, const float* mask, const float* grad_obs,
                                   const float* scalars, const float* gout, float* trans_grad, float* quat_grad,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    if (!trans_grad || !quat_grad || (!grad_obs && (!scalars || !gout))) return TOHIP_EINVAL;
    PoseCall c;
    const int rc = pose_call_init(c, packed, n, trans, quat, cam, mask, workspace, workspace_bytes, stream_);
    if (rc != TOHIP_OK) return rc;
    c.a.grad_obs = grad_obs;
    c.f.coef_mode = grad_obs ? 0 : 1;
    c.f.scalars_in = scalars; c.f.gout = gout;
    c.f.trans_grad = trans_grad; c.f.quat_grad = quat_grad;
    return pose_launch<false, true>(c);
}

extern "C" int tohip_pose_forward_backward(const void* packed, int64_t n, const float* trans, const float* quat,
                                           const tohip_camera* cam, const float* mask, float* obs, float* scalars, const float* gout,
                                           float* trans_grad, float* quat_grad, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!obs || !scalars || !trans_grad || !quat_grad) return TOHIP_EINVAL;
    PoseCall c;
    const int rc = pose_call_init(c, packed, n, trans, quat, cam, mask, workspace, workspace_bytes, stream_);
    if (rc != TOHIP_OK) return rc;
    c.a.obs = obs;
    c.f.scalars_out = scalars;
    c.f.coef_mode = 2; c.f.gout = gout;
    c.f.trans_grad = trans_grad; c.f.quat_grad = quat_grad;
    return pose_launch<true, true>(c);
}

extern "C" int tohip_pose_opt_step(const void* packed, int64_t n, float* trans, float* quat, const tohip_camera* cam, const float* mask,
                                   float* obs, float* scalars, float* trans_grad, float* quat_grad, float* exp_avg_t, float* exp_avg_sq_t,
                                   float* exp_avg_q, float* exp_avg_sq_q, float lr_pose, float lr_quat, float beta1, float beta2,
                                   float adam_eps, int32_t step, float* loss_log, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!obs || !scalars || !exp_avg_t || !exp_avg_sq_t || !exp_avg_q || !exp_avg_sq_q || !loss_log || step < 1) return TOHIP_EINVAL;
    PoseCall c;
    const int rc = pose_call_init(c, packed, n, trans, quat, cam, mask, workspace, workspace_bytes, stream_);
    if (rc != TOHIP_OK) return rc;
    c.a.obs = obs;
    c.f.scalars_out = scalars;
    c.f.coef_mode = 2;
    c.f.trans_grad = trans_grad; c.f.quat_grad = quat_grad;
    c.f.adam = 1; c.f.mt = exp_avg_t; c.f.vt = exp_avg_sq_t; c.f.mq = exp_avg_q; c.f.vq = exp_avg_sq_q;
    c.f.lr_pose = lr_pose; c.f.lr_quat = lr_quat; c.f.beta1 = beta1; c.f.beta2 = beta2; c.f.adam_eps = adam_eps; c.f.step = step; c.f.loss_log = loss_log;
    return pose_launch<true, true>(c);
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
