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
#include "opt_step.hpp"

// loss_terms[0..4] = vis, l2, length, smooth, total.  grad_poses (W,3) receives the regularisers' gradient (ADDED to what the
// visibility backward wrote when accumulate != 0).  state (may be NULL): the loss row written is state[3] (steps taken so far).
__device__ __forceinline__ void
regularizers_block(const float* __restrict__ poses, const float* __restrict__ poses0, int W, float smooth_w,
                   float length_w, float eps, const float* __restrict__ scalars /* [1] = loss_vis */,
                   float* __restrict__ loss_terms, float* __restrict__ grad_poses, int accumulate,
                   const float* __restrict__ state, float* __restrict__ grad_terms, double* lds, double* sh) {
    if (state) loss_terms += 8 * (int)state[3];
    const RegOut o = regularizers_eval(poses, poses0, W, smooth_w, length_w, eps, grad_poses, accumulate, grad_terms, lds, sh);
    if (threadIdx.x == 0) write_loss_terms(loss_terms, (double)scalars[1], o);
}

__global__ void __launch_bounds__(TO_BLOCK)
k_traj_regularizers(const float* __restrict__ poses, const float* __restrict__ poses0, int W, float smooth_w,
                    float length_w, float eps, const float* __restrict__ scalars, float* __restrict__ loss_terms,
                    float* __restrict__ grad_poses, int accumulate, const float* __restrict__ state,
                    float* __restrict__ grad_terms) {
    __shared__ double lds[TO_BLOCK / 64];
    __shared__ double sh[4];
    regularizers_block(poses, poses0, W, smooth_w, length_w, eps, scalars, loss_terms, grad_poses, accumulate, state,
                       grad_terms, lds, sh);
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
    if (i < n) adam_element(param, grad[i], m, v, i, lr, beta1, beta2, eps, t);
}

// The reference's early-stop rule, evaluated after the step: gains relative to the first step's values.
__device__ __forceinline__ void early_stop_rule(const float* __restrict__ scalars, const float* __restrict__ loss_terms,
                                                float rewards_th, float smoothness_th, float* __restrict__ state,
                                                int row_from_state) {
    if (row_from_state) loss_terms += 8 * (int)state[3];
    early_stop_next(state, state, scalars[0], loss_terms[3], rewards_th, smoothness_th);   // loss_terms: this step's row
}

__global__ void k_early_stop(const float* __restrict__ scalars /* [0] = mean reward */, const float* __restrict__ loss_terms,
                             float rewards_th, float smoothness_th, float* __restrict__ state, int row_from_state) {
    if (threadIdx.x == 0 && blockIdx.x == 0) early_stop_rule(scalars, loss_terms, rewards_th, smoothness_th, state, row_from_state);
}

// One block does the whole O(W) remainder of an optimisation step (optimizer.optimize_trajectory): scatter of the
// evaluated waypoints' visibility gradients into full (W,3)/(W,4) arrays, criterion regularisers + their gradient on top,
// the two Adam updates, the early-stop rule — five launches and two memsets otherwise.  Every gradient is complete
// before any parameter moves (the regularisers read their neighbours' positions).
struct StepTail {
    float *poses, *quats;
    const float* poses0;
    const float *pg_eval, *qg_eval;  // (n_eval, 3), (n_eval, 4): rows r -> waypoint r * step
    float *pg, *qg;                  // (W, 3), (W, 4): full gradients (outputs)
    float *mp, *vp, *mq, *vq;        // Adam moments
    const float* scalars;
    float* loss_terms;               // (n_steps, 8) log, row = state[3]
    float* state;
    int W, n_eval, step;
    float smooth_w, length_w, eps, lr_pose, lr_quat, beta1, beta2, adam_eps, rewards_th, smoothness_th;
    int64_t log_stride;              // floats between two trajectories' loss logs (several trajectories: one block each)
};

__global__ void __launch_bounds__(TO_BLOCK) k_traj_step_tail(StepTail a) {
    __shared__ double lds[TO_BLOCK];
    __shared__ double sh[4];
    const int t = threadIdx.x;
    {   // block b = trajectory b: equal-length trajectories laid end to end, each with its own state, scalars and log
        const int64_t b = blockIdx.x;
        a.poses += b * a.W * 3; a.quats += b * a.W * 4; a.poses0 += b * a.W * 3;
        a.pg_eval += b * a.n_eval * 3; a.qg_eval += b * a.n_eval * 4;
        a.pg += b * a.W * 3; a.qg += b * a.W * 4;
        a.mp += b * a.W * 3; a.vp += b * a.W * 3; a.mq += b * a.W * 4; a.vq += b * a.W * 4;
        a.scalars += b * 4; a.state += b * 8; a.loss_terms += b * a.log_stride;
    }
    for (int i = t; i < a.W * 3; i += TO_BLOCK) {
        const int j = i / 3, k = i - 3 * j, r = j / a.step;
        a.pg[i] = (j == r * a.step && r < a.n_eval) ? a.pg_eval[3 * r + k] : 0.f;
    }
    for (int i = t; i < a.W * 4; i += TO_BLOCK) {
        const int j = i >> 2, k = i & 3, r = j / a.step;
        a.qg[i] = (j == r * a.step && r < a.n_eval) ? a.qg_eval[4 * r + k] : 0.f;
    }
    __syncthreads();
    regularizers_block(a.poses, a.poses0, a.W, a.smooth_w, a.length_w, a.eps, a.scalars, a.loss_terms, a.pg, 1, a.state,
                       nullptr, lds, sh);
    __syncthreads();
    if (a.state[2] == 0.f) {  // not stopped yet (uniform)
        const int step_idx = (int)a.state[3] + 1;
        for (int i = t; i < a.W * 3; i += TO_BLOCK)
            adam_element(a.poses, a.pg[i], a.mp, a.vp, i, a.lr_pose, a.beta1, a.beta2, a.adam_eps, step_idx);
        for (int i = t; i < a.W * 4; i += TO_BLOCK)
            adam_element(a.quats, a.qg[i], a.mq, a.vq, i, a.lr_quat, a.beta1, a.beta2, a.adam_eps, step_idx);
    }
    __syncthreads();
    if (t == 0) early_stop_rule(a.scalars, a.loss_terms, a.rewards_th, a.smoothness_th, a.state, 1);
}

// poses_e[r] = poses[r * step], quats_e[r] = quats[r * step] in one launch (model.py:217's waypoint selection)
__global__ void k_gather_waypoints(const float* __restrict__ poses, const float* __restrict__ quats, int n_eval, int step,
                                   float* __restrict__ poses_e, float* __restrict__ quats_e) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_eval * 3) { const int r = i / 3, k = i - 3 * r; poses_e[i] = poses[(int64_t)r * step * 3 + k]; }
    if (i < n_eval * 4) { const int r = i >> 2, k = i & 3; quats_e[i] = quats[(int64_t)r * step * 4 + k]; }
}

// several equal-length trajectories laid end to end: rows r*step of each (blockIdx.y = trajectory)
__global__ void k_gather_waypoints_multi(const float* __restrict__ poses, const float* __restrict__ quats, int W, int n_eval, int step,
                                         float* __restrict__ poses_e, float* __restrict__ quats_e) {
    const int64_t b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_eval * 3) { const int r = i / 3, kk = i - 3 * r; poses_e[b * n_eval * 3 + i] = poses[(b * W + (int64_t)r * step) * 3 + kk]; }
    if (i < n_eval * 4) { const int r = i >> 2, kk = i & 3; quats_e[b * n_eval * 4 + i] = quats[(b * W + (int64_t)r * step) * 4 + kk]; }
}

extern "C" int tohip_gather_waypoints_multi(const float* poses, const float* quats, int64_t W, int64_t n_traj, int64_t n_eval, int step,
                                            float* poses_e, float* quats_e, void* stream_) {
    if (!poses || !quats || !poses_e || !quats_e || n_eval <= 0 || step <= 0 || W <= 0 || n_traj <= 0 || n_traj > 65535) return TOHIP_EINVAL;
    const int n = (int)(n_eval * 4);
    k_gather_waypoints_multi<<<dim3((n + 255) / 256, (unsigned)n_traj), 256, 0, (hipStream_t)stream_>>>(poses, quats, (int)W, (int)n_eval, step,
                                                                                                       poses_e, quats_e);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_gather_waypoints(const float* poses, const float* quats, int64_t n_eval, int step, float* poses_e,
                                      float* quats_e, void* stream_) {
    if (!poses || !quats || !poses_e || !quats_e || n_eval <= 0 || step <= 0) return TOHIP_EINVAL;
    const int n = (int)(n_eval * 4);
    k_gather_waypoints<<<(n + 255) / 256, 256, 0, (hipStream_t)stream_>>>(poses, quats, (int)n_eval, step, poses_e, quats_e);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_traj_step_tail_multi(float* poses, float* quats, const float* poses0, int64_t W, int64_t n_traj,
                                          const float* poses_grad_eval, const float* quats_grad_eval, int64_t n_eval, int step,
                                          float* poses_grad, float* quats_grad, float* exp_avg_p, float* exp_avg_sq_p, float* exp_avg_q,
                                          float* exp_avg_sq_q, float smoothness_weight, float traj_length_weight, float eps,
                                          float lr_pose, float lr_quat, float beta1, float beta2, float adam_eps, float rewards_th,
                                          float smoothness_th, const float* scalars, float* loss_terms, int64_t loss_terms_stride,
                                          float* state, void* stream_) {
    if (!poses || !quats || !poses0 || !poses_grad_eval || !quats_grad_eval || !poses_grad || !quats_grad || !exp_avg_p ||
        !exp_avg_sq_p || !exp_avg_q || !exp_avg_sq_q || !scalars || !loss_terms || !state || W < 3 || n_eval <= 0 || step <= 0 ||
        (n_eval - 1) * step >= W || n_traj <= 0)
        return TOHIP_EINVAL;
    StepTail a;
    a.poses = poses; a.quats = quats; a.poses0 = poses0; a.pg_eval = poses_grad_eval; a.qg_eval = quats_grad_eval;
    a.pg = poses_grad; a.qg = quats_grad; a.mp = exp_avg_p; a.vp = exp_avg_sq_p; a.mq = exp_avg_q; a.vq = exp_avg_sq_q;
    a.scalars = scalars; a.loss_terms = loss_terms; a.state = state; a.W = (int)W; a.n_eval = (int)n_eval; a.step = step;
    a.smooth_w = smoothness_weight; a.length_w = traj_length_weight; a.eps = eps; a.lr_pose = lr_pose; a.lr_quat = lr_quat;
    a.beta1 = beta1; a.beta2 = beta2; a.adam_eps = adam_eps; a.rewards_th = rewards_th; a.smoothness_th = smoothness_th;
    a.log_stride = loss_terms_stride;
    k_traj_step_tail<<<(int)n_traj, TO_BLOCK, 0, (hipStream_t)stream_>>>(a);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_traj_step_tail(float* poses, float* quats, const float* poses0, int64_t W, const float* poses_grad_eval,
                                    const float* quats_grad_eval, int64_t n_eval, int step, float* poses_grad, float* quats_grad,
                                    float* exp_avg_p, float* exp_avg_sq_p, float* exp_avg_q, float* exp_avg_sq_q,
                                    float smoothness_weight, float traj_length_weight, float eps, float lr_pose, float lr_quat,
                                    float beta1, float beta2, float adam_eps, float rewards_th, float smoothness_th,
                                    const float* scalars, float* loss_terms, float* state, void* stream_) {
    return tohip_traj_step_tail_multi(poses, quats, poses0, W, 1, poses_grad_eval, quats_grad_eval, n_eval, step, poses_grad, quats_grad,
                                      exp_avg_p, exp_avg_sq_p, exp_avg_q, exp_avg_sq_q, smoothness_weight, traj_length_weight, eps, lr_pose,
                                      lr_quat, beta1, beta2, adam_eps, rewards_th, smoothness_th, scalars, loss_terms, 0, state, stream_);
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

// torch.optim.Adam's update of up to TOHIP_ADAM_MAX_GROUPS parameter tensors in ONE launch (blockIdx.y = tensor): the reference's
// optimiser holds two (poses @ lr_pose, quats @ lr_quat; trajectory_optimization.py:91-94), torch launches ~7 kernels for each.
struct AdamMulti { tohip_adam_group g[TOHIP_ADAM_MAX_GROUPS]; };
__global__ void __launch_bounds__(256) k_adam_multi(AdamMulti a) {
    const tohip_adam_group& g = a.g[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += (int64_t)gridDim.x * blockDim.x)
        adam_element(g.param, g.grad[i], g.exp_avg, g.exp_avg_sq, (int)i, g.lr, g.beta1, g.beta2, g.eps, g.step);
}

extern "C" int tohip_adam_step_multi(const tohip_adam_group* groups, int32_t n_groups, void* stream_) {
    if (!groups || n_groups <= 0 || n_groups > TOHIP_ADAM_MAX_GROUPS) return TOHIP_EINVAL;
    AdamMulti a;
    int64_t nmax = 0;
    for (int i = 0; i < n_groups; ++i) {
        const tohip_adam_group& g = groups[i];
        if (!g.param || !g.grad || !g.exp_avg || !g.exp_avg_sq || g.n <= 0 || g.n > 0x7fffffff || g.step < 1) return TOHIP_EINVAL;
        a.g[i] = g;
        if (g.n > nmax) nmax = g.n;
    }
    int64_t nb = (nmax + 255) / 256;
    if (nb > 1024) nb = 1024;
    k_adam_multi<<<dim3((unsigned)nb, (unsigned)n_groups), 256, 0, (hipStream_t)stream_>>>(a);
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
