// loss_kernels.hip — ModelTraj.forward() and loss.backward() as ONE library call each.
//
// The reference's optimisation loop (/root/reference/src/trajectory_optimization.py:109-116) is
//     optimizer.zero_grad(); loss = model(); loss.backward(); optimizer.step()
// with model() = /root/reference/src/model.py:200-260 (waypoint selection :214-217, visibility + log-odds :217-231,
// rewards :237, criterion :244-260).  On an MI355X the kernels of one step take 0.05-0.15 ms, so the loop is bound by what the
// host does between them: this file gives the host one call for everything model() computes and one for everything
// loss.backward() computes, over a caller-owned description of the model (tohip_traj_loss: pointers and constants only; the
// library keeps no state).
//
//   tohip_traj_loss_forward    probe (every wps_step-th waypoint, read in place), pass 1, the fused sparse kernel (log-odds,
//                              rewards, their sum, the gradient sums of the flagged pairs with unit upstream gradient) ->
//                              criterion's regularisers with their analytic gradients and the visibility scalars
//   tohip_traj_loss_backward   the per-waypoint finish of the visibility gradient (scaled by dL/d loss, read on the device)
//                              -> full (W,3) / (W,4) gradients: evaluated rows scattered, the regularisers' gradient on top
#include "common.hpp"

namespace {

struct LossLayout {
    int64_t n_eval, V, npad;
    size_t off_pe, off_qe, off_lo, off_mm, off_sc, off_pge, off_qge, off_reg, total;
};

inline LossLayout loss_layout(int64_t n, int64_t W, int step, int n_cams) {
    LossLayout l;
    l.n_eval = (W + step - 1) / step;
    l.V = l.n_eval * (n_cams > 0 ? n_cams : 1);
    l.npad = tohip_padded_points(n);
    size_t o = 0;
    l.off_pe = o;  o += align_up((size_t)l.n_eval * 3 * sizeof(float), 256);
    l.off_qe = o;  o += align_up((size_t)l.n_eval * 4 * sizeof(float), 256);
    l.off_lo = o;  o += align_up((size_t)l.npad * sizeof(float), 256);
    l.off_mm = o;  o += align_up((size_t)l.V * 2 * sizeof(float), 256);
    l.off_sc = o;  o += 256;
    l.off_pge = o; o += align_up((size_t)l.n_eval * 3 * sizeof(float), 256);
    l.off_qge = o; o += align_up((size_t)l.n_eval * 4 * sizeof(float), 256);
    l.off_reg = o; o += align_up((size_t)W * 3 * sizeof(float), 256);
    l.total = o;
    return l;
}

inline bool loss_plan_ok(const tohip_traj_loss* p, LossLayout* l, int* C) {
    if (!p || !p->packed || !p->poses0 || !p->workspace || !p->scratch || p->n_points <= 0 || p->n_wps < 3 || p->wps_step < 1) return false;
    *C = (p->rig.n_cams > 0 && p->rig.rig_quats) ? p->rig.n_cams : 1;
    *l = loss_layout(p->n_points, p->n_wps, p->wps_step, *C);
    return true;
}

}  // namespace

// full gradients of the loss: row w of the evaluated waypoints' visibility gradient (already scaled by dL/d loss in the finish
// kernel) where w is a multiple of `step`, zero elsewhere; the regularisers' gradient, scaled here, on top
__global__ void k_traj_loss_grad(const float* __restrict__ pg_e, const float* __restrict__ qg_e, const float* __restrict__ reg,
                                 const float* __restrict__ gout, int W, int n_eval, int step, float* __restrict__ pg,
                                 float* __restrict__ qg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float g = gout[0];
    if (i < W * 3) {
        const int j = i / 3, k = i - 3 * j, r = j / step;
        const float v = (j == r * step && r < n_eval) ? pg_e[3 * r + k] : 0.f;
        pg[i] = v + g * reg[i];
    }
    if (i < W * 4) {
        const int j = i >> 2, k = i & 3, r = j / step;
        qg[i] = (j == r * step && r < n_eval) ? qg_e[4 * r + k] : 0.f;
    }
}

extern "C" size_t tohip_traj_loss_scratch_bytes(int64_t n_points, int64_t n_wps, int32_t wps_step, int32_t n_cams) {
    if (n_points <= 0 || n_wps <= 0 || wps_step < 1) return 0;
    return loss_layout(n_points, n_wps, wps_step, n_cams).total;
}

extern "C" int tohip_traj_loss_scratch_layout(int64_t n_points, int64_t n_wps, int32_t wps_step, int32_t n_cams, int64_t* offsets_host) {
    if (n_points <= 0 || n_wps <= 0 || wps_step < 1 || !offsets_host) return TOHIP_EINVAL;
    const LossLayout l = loss_layout(n_points, n_wps, wps_step, n_cams);
    const size_t o[8] = {l.off_pe, l.off_qe, l.off_lo, l.off_mm, l.off_sc, l.off_pge, l.off_qge, l.off_reg};
    for (int i = 0; i < 8; ++i) offsets_host[i] = (int64_t)o[i];
    return TOHIP_OK;
}

extern "C" int tohip_traj_loss_forward(const tohip_traj_loss* p, const float* poses, const float* quats, float* rewards,
                                       float* loss_terms, void* stream_) {
    LossLayout l;
    int C;
    if (!loss_plan_ok(p, &l, &C) || !poses || !quats || !rewards || !loss_terms) return TOHIP_EINVAL;
    if (p->scratch_bytes < l.total) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    char* sc = (char*)p->scratch;
    float* lo = (float*)(sc + l.off_lo);
    float* mm = (float*)(sc + l.off_mm);
    float* scal = (float*)(sc + l.off_sc);
    float* reg = (float*)(sc + l.off_reg);
    const tohip_rig* rig = C > 1 || p->rig.rig_quats ? &p->rig : nullptr;
    // probe, pass 1, k_traj_sparse<FUSED>: log-odds, rewards, their integer sum, the pair sums with unit upstream gradient
    TrajStep s;
    int rc = traj_step_init(s, p->packed, p->n_points, l.n_eval, 1, nullptr, &p->cam, rig, p->flags, nullptr, p->workspace, p->workspace_bytes, st, true);
    if (rc != TOHIP_OK) return rc;
    s.wp_stride = p->wps_step;   // every wps_step-th waypoint is evaluated (model.py:215-217): the probe reads them in place
    rc = traj_fused_forward(s, poses, quats, lo, mm, rewards);
    if (rc != TOHIP_OK) return rc;
    // criterion: the scalars of the visibility term come out of the integer reward sum first
    k_traj_regularizers<<<1, TO_BLOCK, 0, st>>>(poses, p->poses0, (int)p->n_wps, p->smoothness_weight, p->traj_length_weight, p->cam.eps,
                                                scal, loss_terms, reg, 0, nullptr, p->reg_terms, s.acc, p->n_points, s.shift);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_traj_loss_backward(const tohip_traj_loss* p, const float* gout, float* poses_grad, float* quats_grad, void* stream_) {
    LossLayout l;
    int C;
    if (!loss_plan_ok(p, &l, &C) || !gout || !poses_grad || !quats_grad) return TOHIP_EINVAL;
    if (p->scratch_bytes < l.total) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    char* sc = (char*)p->scratch;
    float* pge = (float*)(sc + l.off_pge);
    float* qge = (float*)(sc + l.off_qge);
    const float* reg = (const float*)(sc + l.off_reg);
    const tohip_rig* rig = C > 1 || p->rig.rig_quats ? &p->rig : nullptr;
    TrajStep s;
    int rc = traj_step_init(s, p->packed, p->n_points, l.n_eval, 1, nullptr, &p->cam, rig, p->flags, nullptr, p->workspace, p->workspace_bytes, st, false);
    if (rc != TOHIP_OK) return rc;
    rc = launch_finish(s, finish_post(s, 2, nullptr, gout, nullptr, p->cam.eps), pge, qge);
    if (rc != TOHIP_OK) return rc;
    const int n = (int)(p->n_wps * 4);
    k_traj_loss_grad<<<(n + 255) / 256, 256, 0, st>>>(pge, qge, reg, gout, (int)p->n_wps, (int)l.n_eval, p->wps_step, poses_grad, quats_grad);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}
