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
//   tohip_traj_loss_forward    four launches: probe (every wps_step-th waypoint, read in place; one block more computes
//                              criterion's regularisers with their analytic gradients: they depend on the positions only),
//                              pass 1, the fused sparse kernel (log-odds, rewards, their integer sum), the gradient sums of the
//                              flagged pairs with unit upstream gradient (one block more turns the reward sum and the
//                              regularisers' terms into model()'s scalars and loss terms)
//   tohip_traj_loss_backward   one launch: the per-waypoint finish of the visibility gradient (scaled by dL/d loss, read on the
//                              device); every block also writes its waypoint's rows of the full (W,3) / (W,4) gradients —
//                              visibility row + dL/d loss x the regularisers' gradient (opt_step.hpp, mode 2)
#include "common.hpp"
#include "opt_step.hpp"

namespace {

struct LossLayout {
    int64_t n_eval, V, npad;
    size_t off_pe, off_qe, off_lo, off_mm, off_sc, off_pge, off_qge, off_reg, off_pro, total;
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
    l.off_pro = o; o += align_up(sizeof(OptPro), 256);   // the regularisers' terms (prologue -> the pairs' extra block)
    l.total = o;
    return l;
}

inline bool loss_plan_ok(const tohip_traj_loss* p, LossLayout* l, int* C) {
    if (!p || !p->packed || !p->poses0 || !p->workspace || !p->scratch || p->n_points <= 0 || p->n_wps < 3 || p->wps_step < 1) return false;
    *C = (p->rig.n_cams > 0 && p->rig.rig_quats) ? p->rig.n_cams : 1;
    *l = loss_layout(p->n_points, p->n_wps, p->wps_step, *C);
    return true;
}

// what the prologue / epilogue blocks need of the plan (opt_step.hpp, mode 2)
inline void loss_opt(OptStep& a, const tohip_traj_loss* p, const LossLayout& l) {
    char* sc = (char*)p->scratch;
    a.mode = 2;
    a.poses0 = p->poses0;
    a.pro = (OptPro*)(sc + l.off_pro);
    a.reg = (float*)(sc + l.off_reg);
    a.reg_terms = p->reg_terms;
    a.W = (int)p->n_wps; a.n_eval = (int)l.n_eval; a.step = p->wps_step; a.n_traj = 1;
    a.smooth_w = p->smoothness_weight; a.length_w = p->traj_length_weight; a.eps = p->cam.eps;
}

}  // namespace

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
    const tohip_rig* rig = C > 1 || p->rig.rig_quats ? &p->rig : nullptr;
    // probe (+ the regularisers' block), pass 1, k_traj_sparse<FUSED>, the pair sums with unit upstream gradient (+ the scalars' block)
    TrajStep s;
    int rc = traj_step_init(s, p->packed, p->n_points, l.n_eval, 1, nullptr, &p->cam, rig, p->flags & 0xff, nullptr, p->workspace, p->workspace_bytes, st, true);
    if (rc != TOHIP_OK) return rc;
    s.wp_stride = p->wps_step;   // every wps_step-th waypoint is evaluated (model.py:215-217): the probe reads them in place
    loss_opt(s.opt, p, l);
    s.opt.poses = const_cast<float*>(poses);
    s.opt.loss_log = loss_terms;
    s.opt_scalars = scal;
    return traj_fused_forward(s, poses, quats, lo, mm, rewards);
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
    const tohip_rig* rig = C > 1 || p->rig.rig_quats ? &p->rig : nullptr;
    TrajStep s;
    int rc = traj_step_init(s, p->packed, p->n_points, l.n_eval, 1, nullptr, &p->cam, rig, p->flags & 0xff, nullptr, p->workspace, p->workspace_bytes, st, false);
    if (rc != TOHIP_OK) return rc;
    loss_opt(s.opt, p, l);
    s.opt.pg = poses_grad; s.opt.qg = quats_grad; s.opt.gout = gout;
    return launch_finish(s, finish_post(s, 2, nullptr, gout, nullptr, p->cam.eps), pge, qge);
}

// tohip_traj_backward (a general dL/d rewards, or the fused loss through the separate kernels) leaves ITS pair sums — scaled by
// its upstream gradient — where tohip_traj_loss_backward expects the sums taken with unit upstream gradient: this takes those
// again (the step's pair list, log-odds vector and extrema are untouched by a backward: same pairs, same bits).
extern "C" int tohip_traj_loss_refresh(const tohip_traj_loss* p, void* stream_) {
    LossLayout l;
    int C;
    if (!loss_plan_ok(p, &l, &C)) return TOHIP_EINVAL;
    if (p->scratch_bytes < l.total) return TOHIP_ENOSPC;
    const tohip_rig* rig = C > 1 || p->rig.rig_quats ? &p->rig : nullptr;
    TrajStep s;
    int rc = traj_step_init(s, p->packed, p->n_points, l.n_eval, 1, nullptr, &p->cam, rig, p->flags & 0xff, nullptr, p->workspace, p->workspace_bytes, stream_, false);
    if (rc != TOHIP_OK) return rc;
    return launch_pairs(s, sparse_args(s, (float*)((char*)p->scratch + l.off_lo)));
}
