// opt_step.hpp — the O(W) remainder of an optimisation step as device functions that ride inside the visibility launches.
//
//   criterion: l2 / smooth / length   /root/reference/src/model.py:244-260, :135-155
//   Adam (two parameter groups)       /root/reference/src/trajectory_optimization.py:91-94 (torch.optim.Adam defaults)
//   early stop                        /root/reference/src/trajectory_optimization.py:100-124
//
// The regularisers depend on the positions only, which are known when a step starts; the Adam update of a waypoint's rows
// needs that waypoint's visibility gradient only, which its own block of k_traj_finish has in hand.  So a step needs no
// launch of its own for any of this:
//   prologue   one extra block per trajectory in the sparse kernel's launch (traj_kernels.hip: a quarter of that launch's blocks
//              leave at once, it runs beside the others; in the probe's launch it was the longest block): the regularisers' values
//              and their gradient (W,3), and the step's Adam constants, into the step's scratch
//   epilogue   every block of k_traj_finish (one per evaluated waypoint) updates the rows [r*step, (r+1)*step) of its
//              trajectory — full gradient = visibility row (first row only) + regularisers — and the block of a trajectory's
//              first waypoint also writes the loss log and the next row of the early-stop state
// The state of step i is ROW i of a log (n_steps + 1 rows of 8 floats per trajectory, row 0 zero): every block reads row i,
// one block writes row i + 1 — no block waits for another inside the launch.
//   state row: [0] reward0 (mean reward of the first step)   [1] smooth0 (smooth loss of the first step)
//              [2] stopped (0/1)   [3] steps taken   [4] last visibility gain   [5] last smooth gain
#pragma once
#include "common.hpp"

#define TO_OPT_STATE 8

// ---- criterion's regularisers with their analytic gradients ---------------------------------------------------------------
// Works for any blockDim.x >= TO_BLOCK: the first TO_BLOCK threads do the work in the order a TO_BLOCK-thread block would, the
// others add zeros — the sums are the same bits whatever launch the block rides in.
struct RegOut {
    double l2, length, smooth;   // the three terms (model.py:249-258)
};

// grad_poses (W,3), may be NULL: the regularisers' gradient (ADDED to its content when accumulate != 0); grad_terms, may be
// NULL: (3, W, 3) = d l2, d length, d smooth separately.  Returns the terms in every thread.
__device__ __forceinline__ RegOut
regularizers_eval(const float* __restrict__ poses, const float* __restrict__ poses0, int W, float smooth_w, float length_w,
                  float eps, float* __restrict__ grad_poses, int accumulate, float* __restrict__ grad_terms,
                  double* lds /* one double per wave */, double* sh /* 4 doubles */) {
    const int t = threadIdx.x;
    const bool worker = t < TO_BLOCK;
    auto P = [&](const float* a, int i, int k) { return (double)a[3 * i + k]; };
    // ---- forward sums -----------------------------------------------------------------------------
    double len = 0, len0 = 0, ang = 0;
    if (worker) {
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
            double c = dot / (sqrt(ab2 * ac2) + (double)eps);
            c = fmin(1.0, fmax(-1.0, c));
            ang += acos(c);
        }
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
    RegOut o;
    o.smooth = (double)smooth_w / (mean_angle + (double)eps);
    const double dlen = sh[0] - sh[1];
    o.length = (double)length_w * fabs(dlen);
    double l2sq = 0;
    for (int k = 0; k < 3; ++k) { const double d = P(poses, 0, k) - P(poses0, 0, k); l2sq += d * d; }
    o.l2 = sqrt(l2sq);
    if ((!grad_poses && !grad_terms) || !worker) return o;
    // ---- gradients: thread per waypoint gathers the terms it appears in ------------------------------
    const double dsm_dang = -o.smooth / (mean_angle + (double)eps) / (double)(W - 2);  // d smooth / d phi_i
    const double dlen_w = (double)length_w * (dlen > 0 ? 1.0 : (dlen < 0 ? -1.0 : 0.0));
    for (int j = t; j < W; j += TO_BLOCK) {
        double gl[3] = {0, 0, 0}, g2[3] = {0, 0, 0}, gs[3] = {0, 0, 0};  // length, l2, smooth
        // length: segments (j-1, j) and (j, j+1)
        if (j > 0) {
            double d[3], s = 0;
            for (int k = 0; k < 3; ++k) { d[k] = P(poses, j, k) - P(poses, j - 1, k); s += d[k] * d[k]; }
            if (s > 0) { const double f = dlen_w / sqrt(s); for (int k = 0; k < 3; ++k) gl[k] += f * d[k]; }
        }
        if (j < W - 1) {
            double d[3], s = 0;
            for (int k = 0; k < 3; ++k) { d[k] = P(poses, j + 1, k) - P(poses, j, k); s += d[k] * d[k]; }
            if (s > 0) { const double f = dlen_w / sqrt(s); for (int k = 0; k < 3; ++k) gl[k] -= f * d[k]; }
        }
        // l2 on the first waypoint
        if (j == 0 && o.l2 > 0) for (int k = 0; k < 3; ++k) g2[k] += (P(poses, 0, k) - P(poses0, 0, k)) / o.l2;
        // smoothness: waypoint j is the corner of angle j and an end point of angles j-1 and j+1
        for (int i = j - 1; i <= j + 1; ++i) {
            if (i < 1 || i > W - 2) continue;
            double ab[3], ac[3], nab = 0, nac = 0, dot = 0;
            for (int k = 0; k < 3; ++k) {
                ab[k] = P(poses, i - 1, k) - P(poses, i, k); ac[k] = P(poses, i + 1, k) - P(poses, i, k);
                nab += ab[k] * ab[k]; nac += ac[k] * ac[k]; dot += ab[k] * ac[k];
            }
            nab = sqrt(nab); nac = sqrt(nac);
            const double inv_den = 1.0 / (nab * nac + (double)eps);
            const double c = dot * inv_den;
            if (!(c > -1.0 && c < 1.0)) continue;  // arccos' derivative is unbounded at +-1 (torch: inf/nan)
            const double coef = -dsm_dang / sqrt(1.0 - c * c);   // d smooth / d phi_i x d phi / d c
            // dc/dAB = AC/den - c * nac * AB/(nab*den),  dc/dAC symmetric (three divisions per angle: f64 division is ~35 instructions)
            const double kab = nab > 0 ? c * nac * inv_den / nab : 0.0, kac = nac > 0 ? c * nab * inv_den / nac : 0.0;
            for (int k = 0; k < 3; ++k) {
                const double dab = ac[k] * inv_den - kab * ab[k], dac = ab[k] * inv_den - kac * ac[k];
                double dc;
                if (j == i - 1) dc = dab;
                else if (j == i + 1) dc = dac;
                else dc = -dab - dac;
                gs[k] += coef * dc;
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
    return o;
}

// loss_terms[0..4] = vis, l2, length, smooth, total
__device__ __forceinline__ void write_loss_terms(float* __restrict__ loss_terms, double vis, const RegOut& o) {
    loss_terms[0] = (float)vis; loss_terms[1] = (float)o.l2; loss_terms[2] = (float)o.length; loss_terms[3] = (float)o.smooth;
    loss_terms[4] = (float)(vis + o.l2 + o.length + o.smooth);
}

// ---- torch.optim.Adam (defaults betas=(0.9,0.999), eps=1e-8, no weight decay / amsgrad) --------------------------------------
// The bias corrections of step t (1-based) as torch applies them: step_size = lr / (1 - beta1^t) computed in double and used as
// a float scalar, denominator sqrt(v) / sqrt(1 - beta2^t) + eps.
struct AdamConsts {
    float step_size, sqrt_bc2;
};
__device__ __forceinline__ AdamConsts adam_consts(float lr, float beta1, float beta2, int t) {
    const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
    AdamConsts c;
    c.step_size = (float)((double)lr / bc1);
    c.sqrt_bc2 = (float)sqrt(bc2);
    return c;
}
__device__ __forceinline__ void adam_apply(float* __restrict__ param, float g, float* __restrict__ m, float* __restrict__ v, int i,
                                           float beta1, float beta2, float eps, const AdamConsts& c) {
    const float mi = beta1 * m[i] + (1.0f - beta1) * g;      // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = beta2 * v[i] + (1.0f - beta2) * g * g;   // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / c.sqrt_bc2 + eps;
    param[i] = param[i] - c.step_size * (mi / denom);
}
__device__ __forceinline__ void adam_element(float* __restrict__ param, float g, float* __restrict__ m, float* __restrict__ v,
                                             int i, float lr, float beta1, float beta2, float eps, int t) {
    adam_apply(param, g, m, v, i, beta1, beta2, eps, adam_consts(lr, beta1, beta2, t));
}

// ---- the reference's early-stop rule: gains relative to the first step's values ------------------------------------------------
// in = the state before the step, out = after it (may alias).  mean_r: the step's mean reward; smooth: its smooth loss.
__device__ __forceinline__ void early_stop_next(const float* __restrict__ in, float* __restrict__ out, float mean_r, float smooth,
                                                float rewards_th, float smoothness_th) {
    float s[TO_OPT_STATE];
    for (int i = 0; i < TO_OPT_STATE; ++i) s[i] = in[i];
    if (s[2] == 0.f) {
        if (s[3] == 0.f) { s[0] = mean_r; s[1] = smooth; }
        s[3] += 1.f;
        const float vg = mean_r / s[0], sg = s[1] / smooth;
        s[4] = vg; s[5] = sg;
        if (vg > rewards_th && sg > smoothness_th) s[2] = 1.f;
    }
    for (int i = 0; i < TO_OPT_STATE; ++i) out[i] = s[i];
}

// ---- one optimisation step's constants, shared by the prologue and the epilogue ----------------------------------------------
// What the prologue leaves per trajectory in the step's scratch (256-byte aligned): the regularisers' terms and the Adam
// constants of this step; the gradient rows follow in `reg` (n_traj x W x 3 floats).
struct __attribute__((aligned(64))) OptPro {
    double l2, length, smooth;
    float ss_p, ss_q, sqrt_bc2;   // Adam: lr_pose / bc1, lr_quat / bc1, sqrt(bc2) of step (state row)[3] + 1
    float pad;
};

struct OptStep {
    int mode;                  // 0: off   1: an optimisation step (regularisers, Adam, early stop)   2: model() / loss.backward() (loss_kernels.hip)
    // per trajectory b: poses / quats / poses0 / moments / full gradients at row b * W
    float *poses, *quats;
    const float* poses0;
    float *mp, *vp, *mq, *vq;  // mode 1: Adam moments
    float *pg, *qg;            // (n_traj * W, 3 / 4): the step's full gradients (outputs of the epilogue)
    OptPro* pro;               // n_traj
    float* reg;                // n_traj x W x 3: the regularisers' gradient (prologue -> epilogue)
    float* reg_terms;          // mode 2, may be NULL: (3, W, 3) = d l2, d length, d smooth separately
    float* loss_log;           // mode 1: trajectory b's log at + b * log_stride: (n_steps, 8); mode 2: the caller's loss_terms (8 floats)
    const float* state_in;     // mode 1: trajectory b's state row of this step at + b * state_stride
    float* state_out;          //   ... and of the next one
    const float* gout;         // mode 2: dL/d loss (device)
    int64_t log_stride, state_stride;
    int W, n_eval, step, n_traj;
    float smooth_w, length_w, eps, lr_pose, lr_quat, beta1, beta2, adam_eps, rewards_th, smoothness_th;
};

// prologue: block `b` of the extra blocks of the SPARSE kernel's launch = trajectory b (what it leaves — pro, reg — is for the
// launches after that one: the pairs' extra block, the finish blocks)
__device__ __forceinline__ void opt_prologue_block(const OptStep& a, int b, double* lds, double* sh) {
    const float* poses = a.poses + (int64_t)b * a.W * 3;
    const float* poses0 = a.poses0 + (int64_t)b * a.W * 3;
    // Adam's bias corrections are two f64 pow() — a few hundred instructions of one thread: a wave the regularisers leave idle
    // takes them meanwhile (a block of TO_BLOCK threads has none: thread 0, afterwards)
    const int adam_thread = blockDim.x > TO_BLOCK ? TO_BLOCK : 0;
    OptPro p;
    p.ss_p = p.ss_q = p.sqrt_bc2 = p.pad = 0.f;
    auto adam = [&]() {
        const int t = (int)a.state_in[(int64_t)b * a.state_stride + 3] + 1;
        const double bc1 = 1.0 - pow((double)a.beta1, (double)t), bc2 = 1.0 - pow((double)a.beta2, (double)t);
        p.ss_p = (float)((double)a.lr_pose / bc1); p.ss_q = (float)((double)a.lr_quat / bc1); p.sqrt_bc2 = (float)sqrt(bc2);
    };
    if (a.mode == 1 && adam_thread != 0 && threadIdx.x == adam_thread) {
        adam();
        a.pro[b].ss_p = p.ss_p; a.pro[b].ss_q = p.ss_q; a.pro[b].sqrt_bc2 = p.sqrt_bc2; a.pro[b].pad = 0.f;
    }
    const RegOut o = regularizers_eval(poses, poses0, a.W, a.smooth_w, a.length_w, a.eps, a.reg + (int64_t)b * a.W * 3, 0,
                                       a.mode == 2 ? a.reg_terms : nullptr, lds, sh);
    if (threadIdx.x == 0) {
        a.pro[b].l2 = o.l2; a.pro[b].length = o.length; a.pro[b].smooth = o.smooth;
        if (a.mode == 1 && adam_thread == 0) adam();
        if (a.mode != 1 || adam_thread == 0) { a.pro[b].ss_p = p.ss_p; a.pro[b].ss_q = p.ss_q; a.pro[b].sqrt_bc2 = p.sqrt_bc2; a.pro[b].pad = 0.f; }
    }
}

// epilogue of evaluated waypoint r of trajectory b: element e = 7 * jj + k of the rows [r * step, r * step + step) — the caller
// spreads e over its threads (e < 7 * step).  What an element reads does not depend on the step's results, so a kernel asks
// for it when it starts (opt_elem_load) and uses it when its waypoint's gradient row is known (opt_elem_apply: vis = that row,
// 3 + 4 floats, any address space).  mode 1: full gradient, then Adam unless the run has stopped; mode 2: full gradient of
// the loss (vis is already scaled by gout).
struct OptElem {
    int64_t at;      // index into the (n_traj W, 3) / (n_traj W, 4) arrays; < 0: no such element
    int k;           // 0..2 position, 3..6 quaternion component
    bool first;      // the evaluated waypoint's own row (the others of the stride carry the regularisers' gradient only)
    float reg, m, v, param;
};
__device__ __forceinline__ OptElem opt_elem_load(const OptStep& a, int b, int r, int e) {
    OptElem o;
    const int jj = e / 7, k = e - 7 * jj;
    const int j = r * a.step + jj;
    o.k = k; o.first = jj == 0; o.reg = o.m = o.v = o.param = 0.f;
    o.at = -1;
    if (j >= a.W || jj >= a.step) return o;
    const int64_t row = (int64_t)b * a.W + j;
    o.at = k < 3 ? row * 3 + k : row * 4 + (k - 3);
    if (k < 3) o.reg = a.reg[o.at];
    if (a.mode == 1) {
        o.m = (k < 3 ? a.mp : a.mq)[o.at];
        o.v = (k < 3 ? a.vp : a.vq)[o.at];
        o.param = (k < 3 ? a.poses : a.quats)[o.at];
    }
    return o;
}
__device__ __forceinline__ void opt_elem_apply(const OptStep& a, const OptElem& o, const float* vis, bool stopped, const OptPro& p) {
    if (o.at < 0) return;
    const float gv = o.first ? vis[o.k] : 0.f;
    const bool pos = o.k < 3;
    const float g = !pos ? gv : (a.mode == 2 ? gv + a.gout[0] * o.reg : gv + o.reg);
    (pos ? a.pg : a.qg)[o.at] = g;
    if (a.mode != 1 || stopped) return;
    // torch.optim.Adam's update (adam_apply) on the values loaded ahead
    const float mi = a.beta1 * o.m + (1.0f - a.beta1) * g;
    const float vi = a.beta2 * o.v + (1.0f - a.beta2) * g * g;
    (pos ? a.mp : a.mq)[o.at] = mi;
    (pos ? a.vp : a.vq)[o.at] = vi;
    const float denom = sqrtf(vi) / p.sqrt_bc2 + a.adam_eps;
    (pos ? a.poses : a.quats)[o.at] = o.param - (pos ? p.ss_p : p.ss_q) * (mi / denom);
}
__device__ __forceinline__ void opt_update_element(const OptStep& a, int b, int r, int e, const float* vis, bool stopped,
                                                   const OptPro& p) {
    opt_elem_apply(a, opt_elem_load(a, b, r, e), vis, stopped, p);
}

// the trajectory's loss log row and its next state row (one thread of the block that holds the trajectory's first waypoint);
// scalars = the trajectory's (mean reward, loss_vis, ...) of this step
__device__ __forceinline__ void opt_commit(const OptStep& a, int b, const float* scalars, const OptPro& p) {
    const float* in = a.state_in + (int64_t)b * a.state_stride;
    float* out = a.state_out + (int64_t)b * a.state_stride;
    RegOut o;
    o.l2 = p.l2; o.length = p.length; o.smooth = p.smooth;
    if (in[2] == 0.f) write_loss_terms(a.loss_log + (int64_t)b * a.log_stride + 8 * (int)in[3], (double)scalars[1], o);
    early_stop_next(in, out, scalars[0], (float)p.smooth, a.rewards_th, a.smoothness_th);
}
