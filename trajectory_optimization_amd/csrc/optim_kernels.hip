// optim_kernels.hip — the O(W) remainder of one optimisation step on the device:
// the regularisers of ModelTraj.criterion with their analytic gradients, the total loss, Adam, and the
// reference's early-stop rule, so that an optimisation run enqueues kernels only and never syncs the host.
//
//   criterion: l2 / smooth / length   /root/reference/src/model.py:244-260, :135-155
//   Adam (two parameter groups)       /root/reference/src/trajectory_optimization.py:91-94 (torch.optim.Adam defaults)
//   early stop                        /root/reference/src/trajectory_optimization.py:100-124
//
// Everything here is a single small block: W is tens to a few thousand waypoints.
#include "common.hpp"

// state[] (floats, device):  [0] reward0 (mean reward of the first step)   [1] smooth0 (smooth loss of the first step)
//                            [2] stopped (0/1)   [3] steps taken   [4] last visibility gain   [5] last smooth gain
#define TO_OPT_STATE 8

// loss_terms[0..4] = vis, l2, length, smooth, total.  grad_poses (W,3) receives the regularisers' gradient
// ADDED to what the visibility backward wrote (pass accumulate = 1), scaled by gout (dL/d total = 1).
__global__ void __launch_bounds__(TO_BLOCK)
k_traj_regularizers(const float* __restrict__ poses, const float* __restrict__ poses0, int W, float smooth_w,
                    float length_w, float eps, const float* __restrict__ scalars /* [1] = loss_vis */,
                    float* __restrict__ loss_terms, float* __restrict__ grad_poses, int accumulate,
                    const float* __restrict__ state /* may be NULL; else loss row = state[3] (steps taken so far) */,
                    float* __restrict__ grad_terms /* may be NULL; else (3, W, 3): d l2, d length, d smooth separately */) {
    if (state) loss_terms += 8 * (int)state[3];
    __shared__ double lds[TO_BLOCK];
    __shared__ double sh[4];
    const int t = threadIdx.x;
    auto P = [&](const float* a, int i, int k) { return (double)a[3 * i + k]; };
    // ---- forward sums -----------------------------------------------------------------------------
    double len = 0, len0 = 0, ang = 0;
    for (int i = t; i < W - 1; i += TO_BLOCK) {
        double s = 0, s0 = 0;
        for (int k = 0; k < 3; ++k) {
            const double d = P(poses, i + 1, k) - P(poses, i, k), d0 = P(poses0, i + 1, k) - P(poses0, i, k);
            s += d * d; s0 += d0 * d0;
        }
        len += sqrt(s); len0 += sqrt(s0);
    }
    for (int i = 1 + t; i < W - 1; i += TO_BLOCK) {
        double ab2 = 0, ac2 = 0, dot = 0;
        for (int k = 0; k < 3; ++k) {
            const double ab = P(poses, i - 1, k) - P(poses, i, k), ac = P(poses, i + 1, k) - P(poses, i, k);
            ab2 += ab * ab; ac2 += ac * ac; dot += ab * ac;
        }
        double c = dot / (sqrt(ab2) * sqrt(ac2) + (double)eps);
        c = fmin(1.0, fmax(-1.0, c));
        ang += acos(c);
    }
    const double L = block_sum_double(len, lds);
    if (t == 0) sh[0] = L;
    __syncthreads();
    const double L0 = block_sum_double(len0, lds);
    if (t == 0) sh[1] = L0;
    __syncthreads();
    const double A = block_sum_double(ang, lds);
    if (t == 0) sh[2] = A;
    __syncthreads();
    const double mean_angle = sh[2] / (double)(W - 2);
    const double smooth = (double)smooth_w / (mean_angle + (double)eps);
    const double dlen = sh[0] - sh[1];
    const double length = (double)length_w * fabs(dlen);
    double l2sq = 0;
    for (int k = 0; k < 3; ++k) { const double d = P(poses, 0, k) - P(poses0, 0, k); l2sq += d * d; }
    const double l2 = sqrt(l2sq);
    if (t == 0) {
        const double vis = (double)scalars[1];
        loss_terms[0] = (float)vis; loss_terms[1] = (float)l2; loss_terms[2] = (float)length; loss_terms[3] = (float)smooth;
        loss_terms[4] = (float)(vis + l2 + length + smooth);
    }
    if (!grad_poses && !grad_terms) return;
    // ---- gradients: thread per waypoint gathers the terms it appears in ------------------------------
    const double dsm_dang = -smooth / (mean_angle + (double)eps) / (double)(W - 2);  // d smooth / d phi_i
    const double dlen_w = (double)length_w * (dlen > 0 ? 1.0 : (dlen < 0 ? -1.0 : 0.0));
    for (int j = t; j < W; j += TO_BLOCK) {
        double gl[3] = {0, 0, 0}, g2[3] = {0, 0, 0}, gs[3] = {0, 0, 0};  // length, l2, smooth
        // length: segments (j-1, j) and (j, j+1)
        if (j > 0) {
            double d[3], s = 0;
            for (int k = 0; k < 3; ++k) { d[k] = P(poses, j, k) - P(poses, j - 1, k); s += d[k] * d[k]; }
            s = sqrt(s);
            if (s > 0) for (int k = 0; k < 3; ++k) gl[k] += dlen_w * d[k] / s;
        }
        if (j < W - 1) {
            double d[3], s = 0;
            for (int k = 0; k < 3; ++k) { d[k] = P(poses, j + 1, k) - P(poses, j, k); s += d[k] * d[k]; }
            s = sqrt(s);
            if (s > 0) for (int k = 0; k < 3; ++k) gl[k] -= dlen_w * d[k] / s;
        }
        // l2 on the first waypoint
        if (j == 0 && l2 > 0) for (int k = 0; k < 3; ++k) g2[k] += (P(poses, 0, k) - P(poses0, 0, k)) / l2;
        // smoothness: waypoint j is the corner of angle j and an end point of angles j-1 and j+1
        for (int i = j - 1; i <= j + 1; ++i) {
            if (i < 1 || i > W - 2) continue;
            double ab[3], ac[3], nab = 0, nac = 0, dot = 0;
            for (int k = 0; k < 3; ++k) {
                ab[k] = P(poses, i - 1, k) - P(poses, i, k); ac[k] = P(poses, i + 1, k) - P(poses, i, k);
                nab += ab[k] * ab[k]; nac += ac[k] * ac[k]; dot += ab[k] * ac[k];
            }
            nab = sqrt(nab); nac = sqrt(nac);
            const double den = nab * nac + (double)eps;
            const double c = dot / den;
            if (!(c > -1.0 && c < 1.0)) continue;  // arccos' derivative is unbounded at +-1 (torch: inf/nan)
            const double dphi_dc = -1.0 / sqrt(1.0 - c * c);
            // dc/dAB = AC/den - c * nac * AB/(nab*den),  dc/dAC symmetric
            double dab[3], dac[3];
            for (int k = 0; k < 3; ++k) {
                dab[k] = ac[k] / den - (nab > 0 ? c * nac * ab[k] / (nab * den) : 0.0);
                dac[k] = ab[k] / den - (nac > 0 ? c * nab * ac[k] / (nac * den) : 0.0);
            }
            for (int k = 0; k < 3; ++k) {
                double dc;
                if (j == i - 1) dc = dab[k];
                else if (j == i + 1) dc = dac[k];
                else dc = -dab[k] - dac[k];
                gs[k] += dsm_dang * dphi_dc * dc;
            }
        }
        for (int k = 0; k < 3; ++k) {
            if (grad_poses) {
                const float prev = accumulate ? grad_poses[3 * j + k] : 0.f;
                grad_poses[3 * j + k] = prev + (float)(gl[k] + g2[k] + gs[k]);
            }
            if (grad_terms) {
                grad_terms[3 * j + k] = (float)g2[k];
                grad_terms[3 * (W + j) + k] = (float)gl[k];
                grad_terms[3 * (2 * W + j) + k] = (float)gs[k];
            }
        }
    }
}

// scatter the gradient rows of the evaluated waypoints (every wps_step-th) into full (W,3)/(W,4) arrays
__global__ void k_scatter_rows(const float* __restrict__ src, int n_rows, int cols, int step, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * cols) return;
    const int r = i / cols, c = i - r * cols;
    dst[(int64_t)r * step * cols + c] = src[i];
}
__global__ void k_gather_rows(const float* __restrict__ src, int n_rows, int cols, int step, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * cols) return;
    const int r = i / cols, c = i - r * cols;
    dst[i] = src[(int64_t)r * step * cols + c];
}

// torch.optim.Adam (defaults betas=(0.9,0.999), eps=1e-8, no weight decay / amsgrad), one call per parameter group.
// No-op once state[2] (stopped) is set.  `t` = 1-based step index.
__global__ void k_adam(float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ m, float* __restrict__ v,
                       int n, float lr, float beta1, float beta2, float eps, int t, const float* __restrict__ state) {
    if (state && state[2] != 0.f) return;
    if (t <= 0) t = (int)state[3] + 1;  // step index kept on the device: the same launch can be replayed from a hipGraph
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = grad[i];
    const float mi = beta1 * m[i] + (1.0f - beta1) * g;      // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    const float step_size = (float)((double)lr / bc1);
    const float denom = sqrtf(vi) / (float)sqrt(bc2) + eps;
    param[i] = param[i] - step_size * (mi / denom);
}

// The reference's early-stop rule, evaluated after the step: gains relative to the first step's values.
__global__ void k_early_stop(const float* __restrict__ scalars /* [0] = mean reward */, const float* __restrict__ loss_terms,
                             float rewards_th, float smoothness_th, float* __restrict__ state, int row_from_state) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (row_from_state) loss_terms += 8 * (int)state[3];
    if (state[2] != 0.f) return;
    const float mean_r = scalars[0], smooth = loss_terms[3];  // loss_terms: this step's row
    if (state[3] == 0.f) { state[0] = mean_r; state[1] = smooth; }
    state[3] += 1.f;
    const float vg = mean_r / state[0], sg = state[1] / smooth;
    state[4] = vg; state[5] = sg;
    if (vg > rewards_th && sg > smoothness_th) state[2] = 1.f;
}

extern "C" int tohip_traj_regularizers(const float* poses, const float* poses0, int64_t W, float smoothness_weight,
                                       float traj_length_weight, float eps, const float* scalars, float* loss_terms,
                                       float* grad_poses, int accumulate, const float* state, float* grad_terms,
                                       void* stream_) {
    if (!poses || !poses0 || !scalars || !loss_terms || W < 3) return TOHIP_EINVAL;
    k_traj_regularizers<<<1, TO_BLOCK, 0, (hipStream_t)stream_>>>(poses, poses0, (int)W, smoothness_weight,
                                                                  traj_length_weight, eps, scalars, loss_terms, grad_poses,
                                                                  accumulate, state, grad_terms);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_rows_strided(const float* src, int64_t n_rows, int cols, int step, int scatter, float* dst,
                                  void* stream_) {
    if (!src || !dst || n_rows <= 0 || cols <= 0 || step <= 0) return TOHIP_EINVAL;
    const int n = (int)(n_rows * cols);
    if (scatter) k_scatter_rows<<<(n + 255) / 256, 256, 0, (hipStream_t)stream_>>>(src, (int)n_rows, cols, step, dst);
    else k_gather_rows<<<(n + 255) / 256, 256, 0, (hipStream_t)stream_>>>(src, (int)n_rows, cols, step, dst);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               float beta1, float beta2, float eps, int32_t step, const float* state, void* stream_) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || (step < 1 && !state)) return TOHIP_EINVAL;
    k_adam<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream_>>>(param, grad, exp_avg, exp_avg_sq, (int)n, lr, beta1, beta2,
                                                                      eps, step, state);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_early_stop(const float* scalars, const float* loss_terms, float rewards_th, float smoothness_th,
                                float* state, int row_from_state, void* stream_) {
    if (!scalars || !loss_terms || !state) return TOHIP_EINVAL;
    k_early_stop<<<1, 64, 0, (hipStream_t)stream_>>>(scalars, loss_terms, rewards_th, smoothness_th, state, row_from_state);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}
