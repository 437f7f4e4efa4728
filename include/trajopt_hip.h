/*
 * trajopt_hip.h — C ABI of libtrajopt_hip.so: the MI355X (gfx950) implementation of the
 * differentiable point-cloud visibility + coverage-reward path of ctu-vras/trajectory_optimization.
 *
 * The reference has no FFI/plugin seam; its boundary is the Python object API of src/model.py and
 * src/tools.py.  Each entry point below therefore names the reference code it replaces (file:line,
 * relative to the reference checkout).  INTEGRATION.md shows the ctypes stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the caller allocates every input, output and workspace buffer, and no DEVICE memory is ever allocated or freed by the
 *     library.  Nothing a call computes is kept for a later call.  The two things the library does keep, both HOST-side
 *     scratch that never changes a result:
 *       * the hull builds (tohip_convex_hull_vertices, tohip_hidden_pts_removal and its _batched form: the calls that read
 *         counters back and therefore synchronise): two
 *         pinned read-back buffers (~2 KB each) and two events per (calling thread, device), created at the first such call
 *         of the thread on that device and released when the thread ends;
 *       * tohip_profile_enable(1): a pool of timing events per process, grown on demand while profiling is on, released by
 *         tohip_profile_enable(0) after tohip_profile_read;
 *     and the process-wide switches tohip_profile_enable / tohip_profile_clock themselves.  Everything else is stateless:
 *     two threads may call any entry point concurrently on different streams with different workspaces;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are only enqueued,
 *     no entry point synchronises unless its comment says so;
 *   - return value: 0 = ok, >0 = a hipError_t from a launch, <0 = an argument error (TOHIP_E*);
 *     nothing throws, nothing calls exit();
 *   - quaternions are (w,x,y,z) like the reference's models (model.py:69,162);
 *   - float = IEEE binary32 everywhere; indices are int32; counts int64.
 */
#ifndef TRAJOPT_HIP_H
#define TRAJOPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 12 (r05): + tohip_render_points_blend / tohip_render_blend_workspace_bytes; + TOHIP_TRAJ_OPT_LAST_OUTPUTS; tohip_voxel_grid reports
 * PCL's "leaf size too small" case as *out_count = -1.  (11: r04) */
#define TOHIP_ABI_VERSION 12

#define TOHIP_OK 0
#define TOHIP_EINVAL (-1)   /* bad size / null pointer */
#define TOHIP_ENOSPC (-2)   /* caller-provided workspace or output capacity too small */
#define TOHIP_ENOTCONV (-3) /* hull construction did not converge within its round limit */
#define TOHIP_ENAN (-4)     /* hull input holds a NaN (a zero-norm point flips to NaN, tools.py:49-52): scipy raises ValueError */

/* Points are processed in tiles of this many; packed clouds are padded to a multiple of it. */
#define TOHIP_POINT_TILE 2048

/* Camera model shared by all waypoints: load_intrinsics (tools.py:320-325) + the constants the
 * models keep (model.py:91-94,186-189).  Passed by pointer in HOST memory. */
typedef struct tohip_camera {
    float K[9];      /* row-major 3x3 intrinsics */
    float img_width; /* 1232. */
    float img_height;/* 1616. */
    float min_dist;  /* pc_clip_limits[0]  (1.0) */
    float max_dist;  /* pc_clip_limits[1]  (5.0) */
    float eps;       /* 1e-6 */
} tohip_camera;

/* Optional multi-camera rig (BASELINE.json config 5; the reference has no fusion code — each
 * (camera, waypoint) pair is a virtual waypoint with its own min/max normalisation, all log-odds
 * summed).  rig_quats: (C,4) unit wxyz rotating camera->body; rig_trans: (C,3) lever arms in the body
 * frame.  n_cams = 0 or NULL pointers mean "one camera at the body frame". */
typedef struct tohip_rig {
    int32_t n_cams;
    const float *rig_quats; /* device */
    const float *rig_trans; /* device */
} tohip_rig;

int tohip_abi_version(void);
const char *tohip_error_string(int code);

/* ---- cloud packing -------------------------------------------------------------------------
 * The cloud is constant over an optimisation run (model.py:80,174), so it is packed once into an opaque
 * device blob of tohip_packed_cloud_bytes(N) bytes: the points in Morton order as x[Npad] | y[Npad] |
 * z[Npad] (Npad = tohip_padded_points(N); the pad repeats the last sorted point), the permutation back to
 * the caller's order and its inverse, and one bounding sphere per 256 sorted points.  sort = 0 keeps the caller's order
 * (no spatial coherence: the exact culling then rarely fires — the layout for ModelPose, which culls nothing and then reads
 * masks and writes observations in place, 16 bytes at a time).  Needs tohip_pack_workspace_bytes(N)
 * bytes of scratch. */
int64_t tohip_padded_points(int64_t n_points);
size_t tohip_packed_cloud_bytes(int64_t n_points);
size_t tohip_pack_workspace_bytes(int64_t n_points);
int tohip_pack_cloud(const float *xyz, int64_t n_points, int sort, void *packed, void *workspace,
                     size_t workspace_bytes, void *stream);

/* ---- ModelTraj (model.py:200-242 forward, :246 visibility term, autograd backward) ----------
 * Workspace bytes needed by the calls below for n_points and n_virtual = W * max(1,n_cams) (at most 65 536 virtual waypoints).
 * THE WORKSPACE MUST BE ZERO-FILLED ONCE BEFORE ITS FIRST USE (hipMemset): tohip_traj_reward's accumulator word is expected
 * zero and left zero; every forward resets what a step accumulates into, that word included.  The forward leaves its state
 * there (waypoint records, per-waypoint extrema, the list of the pairs that contribute, the candidate slots) and the backward
 * of the same step reads it: do not touch the workspace between the two. */
size_t tohip_traj_workspace_bytes(int64_t n_points, int64_t n_virtual);

/* flags */
#define TOHIP_TRAJ_DENSE 1 /* evaluate every (point, waypoint) pair; default: skip pairs that provably
                              contribute exactly nothing (bitwise identical results, see traj_kernels.hip) */
/* bits 8..23 of flags: the waypoint selection of model.py:215-217 as a stride of the forward's reads — with
 * TOHIP_TRAJ_STRIDE(step) the n_wps evaluated waypoints are rows 0, step, 2 step, ... of poses / quats, read in place (one
 * trajectory; the backward entry points ignore the bits: gradient rows are compact, one per evaluated waypoint). */
#define TOHIP_TRAJ_STRIDE(step) ((((step) - 1) & 0xffff) << 8)
/* tohip_traj_opt.flags only: the two N-sized OUTPUTS of a step — lo_sum and rewards, per trajectory — are written by the run's
 * last step (step_index == n_steps - 1) only.  Every step still computes every reward (their sum, the loss and the gradients need
 * them); what the earlier steps skip is refilling and scattering two vectors of N floats per trajectory that the next step
 * overwrites unread — 64 MB of stores per step for eight trajectories over 1 M points
 * (/root/reference/src/trajectory_optimization.py:147-157 publishes model.rewards once, after the loop). */
#define TOHIP_TRAJ_OPT_LAST_OUTPUTS 2

/* Forward over the W evaluated waypoints (caller has applied wps_step, model.py:214-217):
 * to_camera_frame -> get_dist_mask * get_fov_mask -> per-waypoint (p-min)/max -> clip -> log-odds,
 * summed over waypoints into lo_sum[0..Npad) IN PACKED (sorted) ORDER (overwritten; this rank's partial
 * sum when the waypoints are sharded over GPUs — every rank packs the same cloud the same way).
 * minmax[v] = (min p, max(p - min p)) per virtual waypoint (for inspection; the backward reads the workspace).
 * replaces model.py:217-231. */
int tohip_traj_forward(const void *packed, int64_t n_points, const float *poses, const float *quats, int64_t n_wps,
                       const tohip_camera *cam_host, const tohip_rig *rig_host, int flags, const uint32_t *occlusion_bits,
                       float *lo_sum, float *minmax, float *rewards_half, void *workspace, size_t workspace_bytes,
                       void *stream);
/* rewards_half (may be NULL): N floats that the forward fills with sigmoid(0) = 0.5 on its way — hand the same array to
 * tohip_traj_reward with prefilled = 1 and that call only stores the rewards of the points with a non-zero log-odds (2 % of
 * the cloud on the BASELINE workloads) instead of scattering all N. */
/* occlusion_bits (may be NULL = nothing occluded): per virtual waypoint a row of Npad/32 words, bit i = 1 when the
 * packed (sorted) point i is NOT occluded from that waypoint; an occluded pair has p = 0.  The per-waypoint
 * analogue of ModelPose's occlusion mask (model.py:112-115) that the reference leaves as a TODO (tools.py:61-62).
 * Rows are built with tohip_occlusion_row(s) from a hard-frustum cull + HPR (or z-buffer) of the camera-frame cloud;
 * the bits of the pad positions [N, Npad) repeat the bit of the last sorted point. */
int tohip_inverse_permutation(const void *packed, int64_t n_points, int32_t *inv_perm, void *stream);
int tohip_occlusion_row(int64_t n_points, const int32_t *inv_perm, const int32_t *kept_idx, const int32_t *kept_count,
                        const int32_t *visible_idx_in_kept, const int32_t *visible_count, uint32_t *row, void *stream);
/* The same for n_wps waypoints in four launches.  kept_idx (n_wps, n): waypoint w's kept points in its first kept_count[w]
 * entries (the layout tohip_cull_waypoints writes); vis_idx: the visible ones as positions in that list, waypoint w's in
 * [vis_off[w], vis_off[w+1]) (n_wps+1 device int32); all_visible[w] != 0: nothing of w is occluded.  rows: (n_wps, Npad/32). */
int tohip_occlusion_rows(int64_t n_points, const int32_t *inv_perm, const int32_t *kept_idx, const int32_t *kept_count,
                         const int32_t *vis_idx, const int32_t *vis_off, const int32_t *all_visible, int64_t n_wps,
                         uint32_t *rows, void *stream);

/* The same from per-point visibility VALUES instead of index lists: waypoint w's kept point j is hidden when
 * visible[seg_off[w] + j] == 0 (seg_off: n_wps device int64) — the mask tohip_hidden_pts_removal_batched writes over the waypoints'
 * kept points laid end to end, or tohip_zbuffer_visible_batched's (seg_off[w] = w * n).  A waypoint with fewer than min_points kept
 * points keeps its row of ones (a hull needs 4).  Three launches for all waypoints.  inv_perm may be NULL: kept_idx then holds
 * positions in the packed cloud's order already (the cull ran over the sorted points) in ascending order, and the hidden bits of a
 * word are combined on their way (one atomic per run of neighbours instead of one per hidden point). */
int tohip_occlusion_rows_masked(int64_t n_points, const int32_t *inv_perm, const int32_t *kept_idx, const int32_t *kept_count,
                                const float *visible, const int64_t *seg_off, int32_t min_points, int64_t n_wps, uint32_t *rows,
                                void *stream);

/* rewards[0..N) = sigmoid(lo_sum) in the CALLER'S point order (model.py:237); scalars[0] = mean(rewards),
 * scalars[1] = loss_vis = 1/(mean+eps) (model.py:246), scalars[2] = -loss_vis^2/N (d loss_vis / d reward_n).
 * One launch.  prefilled != 0: the caller promises rewards[0..N) == 0.5 on entry (tohip_traj_forward's rewards_half).
 * workspace: a zero-filled-once region of at least 256 bytes — normally the forward's workspace (none of the forward's
 * state is touched). */
int tohip_traj_reward(const void *packed, const float *lo_sum, int64_t n_points, float eps, int prefilled, float *rewards,
                      float *scalars, void *workspace, size_t workspace_bytes, void *stream);

/* Backward w.r.t. this rank's waypoints: poses_grad (W,3), quats_grad (W,4), for the step whose tohip_traj_forward last
 * used `workspace` (same n_points, n_wps, rig, flags, occlusion_bits).
 * lo_sum = the (all-reduced) log-odds vector in packed order that tohip_traj_reward turned into rewards.
 * The upstream gradient is either grad_rewards (N floats in the caller's order, dL/d rewards: any
 * criterion built on model.rewards, as torch autograd would hand it over), or, when grad_rewards is NULL,
 * the fused visibility loss: scalars (from tohip_traj_reward) and gout = device pointer to dL/d loss_vis.
 * Deterministic: no float atomics anywhere, tie sets of the per-waypoint min()/max() included. */
int tohip_traj_backward(const void *packed, int64_t n_points, int64_t n_wps, const tohip_camera *cam_host,
                        const tohip_rig *rig_host, int flags, const uint32_t *occlusion_bits, const float *lo_sum,
                        const float *grad_rewards, const float *scalars, const float *gout, float *poses_grad,
                        float *quats_grad, void *workspace, size_t workspace_bytes, void *stream);

/* ---- several trajectories over one cloud in one pass ("many trajectories optimised concurrently") ----------
 * n_traj trajectories' waypoints laid end to end (poses (W,3), quats (W,4), W = their total); traj_offsets: n_traj + 1 ascending
 * body-waypoint offsets on the DEVICE (int32; [0] = 0, [n_traj] = W; may be NULL when n_traj == 1).  Every trajectory has its
 * own log-odds vector (lo_sum: n_traj x Npad), rewards (n_traj x N), scalars (n_traj x 4) and upstream gradient (gout: n_traj
 * floats; grad_rewards: n_traj x N); the per-waypoint outputs (minmax, gradients) are simply concatenated.  Each trajectory's
 * results are, bit for bit, those of a call with that trajectory alone.  The single-trajectory entry points above are these
 * with n_traj = 1. */
size_t tohip_traj_workspace_bytes_multi(int64_t n_points, int64_t n_virtual, int64_t n_traj);
int tohip_traj_forward_multi(const void *packed, int64_t n_points, const float *poses, const float *quats, int64_t n_wps,
                             const int32_t *traj_offsets, int64_t n_traj, const tohip_camera *cam_host, const tohip_rig *rig_host,
                             int flags, const uint32_t *occlusion_bits, float *lo_sum, float *minmax, float *rewards_half,
                             void *workspace, size_t workspace_bytes, void *stream);
int tohip_traj_reward_multi(const void *packed, const float *lo_sum, int64_t n_points, int64_t n_traj, float eps, int prefilled,
                            float *rewards, float *scalars, void *workspace, size_t workspace_bytes, void *stream);
int tohip_traj_backward_multi(const void *packed, int64_t n_points, int64_t n_wps, int64_t n_traj, const tohip_camera *cam_host,
                              const tohip_rig *rig_host, int flags, const uint32_t *occlusion_bits, const float *lo_sum,
                              const float *grad_rewards, const float *scalars, const float *gout, float *poses_grad,
                              float *quats_grad, void *workspace, size_t workspace_bytes, void *stream);

/* tohip_traj_reward + tohip_traj_backward of the fused visibility loss (model.py:237,246 and loss.backward() through them) in two
 * launches instead of three: the rewards / mean / loss scalars and the gradient sums both need only the complete log-odds vector,
 * so they share a launch (the sums are taken with unit dL/d reward and scaled by scalars[2] * gout afterwards: they are linear in
 * it).  Arguments as in the two separate calls; `scalars` and `rewards` are outputs.  rewards and scalars are bitwise those of
 * tohip_traj_reward; the gradients agree with tohip_traj_backward's to rounding (the scale factor is applied once per waypoint
 * in f64 instead of once per point in f32). */
int tohip_traj_reward_backward(const void *packed, int64_t n_points, int64_t n_wps, const tohip_camera *cam_host,
                               const tohip_rig *rig_host, int flags, const uint32_t *occlusion_bits, const float *lo_sum, float eps,
                               int prefilled, float *rewards, float *scalars, const float *gout, float *poses_grad,
                               float *quats_grad, void *workspace, size_t workspace_bytes, void *stream);
int tohip_traj_reward_backward_multi(const void *packed, int64_t n_points, int64_t n_wps, int64_t n_traj,
                                     const tohip_camera *cam_host, const tohip_rig *rig_host, int flags,
                                     const uint32_t *occlusion_bits, const float *lo_sum, float eps, int prefilled, float *rewards,
                                     float *scalars, const float *gout, float *poses_grad, float *quats_grad, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* The whole step when NO collective sits between forward and backward (one GPU, or every rank holding all waypoints):
 * tohip_traj_forward + tohip_traj_reward + tohip_traj_backward of the fused visibility loss in FIVE launches — records + probe,
 * pass 1, the sparse kernel (flags, log-odds, rewards, their sum; a block per candidate slot), the gradient sums of the flagged
 * (slot, waypoint) pairs (a wave per pair, dealt evenly to the whole chip), the per-waypoint finish.
 * Outputs as in the separate calls (lo_sum, minmax, rewards, scalars, poses_grad, quats_grad); gout = device pointer(s) to
 * dL/d loss_vis.  rewards, scalars and lo_sum are bitwise those of the separate calls; the gradients agree to rounding (the
 * dL/d reward factor is applied once per waypoint in f64 instead of once per point in f32).  replaces model.py:217-231,:237,:246
 * and loss.backward() through them. */
int tohip_traj_forward_backward(const void *packed, int64_t n_points, const float *poses, const float *quats, int64_t n_wps,
                                const tohip_camera *cam_host, const tohip_rig *rig_host, int flags, const uint32_t *occlusion_bits,
                                float *lo_sum, float *minmax, float *rewards, float *scalars, const float *gout,
                                float *poses_grad, float *quats_grad, void *workspace, size_t workspace_bytes, void *stream);
int tohip_traj_forward_backward_multi(const void *packed, int64_t n_points, const float *poses, const float *quats, int64_t n_wps,
                                      const int32_t *traj_offsets, int64_t n_traj, const tohip_camera *cam_host,
                                      const tohip_rig *rig_host, int flags, const uint32_t *occlusion_bits, float *lo_sum,
                                      float *minmax, float *rewards, float *scalars, const float *gout, float *poses_grad,
                                      float *quats_grad, void *workspace, size_t workspace_bytes, void *stream);

/* ---- a POINT-sharded step (the alternative to sharding waypoints) ------------------------------------------------
 * Every rank packs N / R of the points and evaluates ALL W waypoints on them.  What the ranks exchange does not grow with N:
 *   tohip_traj_pshard_pass1   records, probe and pass 1 on this rank's points: per-waypoint extrema of ITS points in the workspace
 *   (collective 1)            element-wise MAX over the int32 words tohip_traj_extrema_view points at (4 per virtual waypoint:
 *                             -bits(min p), bits(max p), 0, 0 — p >= 0, so the integer order is the float order), IN PLACE
 *   tohip_traj_pshard_local   flags against the (now global) extrema, log-odds, rewards of this rank's points (complete: the rank
 *                             has every waypoint), the gradient sums of its flagged pairs with unit upstream gradient, and per
 *                             waypoint the 40 sums everything after is linear in -> partial[tohip_traj_pshard_partial_count(V)]
 *   (collective 2)            SUM over `partial` (doubles; [0] the reward sum in fixed point, [1] NaN marks, [2] point counts)
 *   tohip_traj_pshard_finish  mean reward of ALL points, loss_vis, dL/d reward, tie shares, chain -> scalars, poses_grad, quats_grad
 *                             (identical on every rank)
 * n_global = the points of all ranks (the fixed-point scale of the reward sum and the mean's denominator).  lo_sum (Npad_local),
 * minmax (V,2), rewards (n_local, this rank's points in the caller's order): as in tohip_traj_forward_backward.  One trajectory. */
size_t tohip_traj_pshard_partial_count(int64_t n_virtual);
int tohip_traj_extrema_view(int64_t n_points, int64_t n_virtual, void *workspace, size_t workspace_bytes, int32_t **words_out_host,
                            int64_t *n_words_out_host);
int tohip_traj_pshard_pass1(const void *packed, int64_t n_local, int64_t n_global, const float *poses, const float *quats,
                            int64_t n_wps, const tohip_camera *cam_host, const tohip_rig *rig_host, int flags,
                            const uint32_t *occlusion_bits, float *lo_sum, float *rewards, void *workspace, size_t workspace_bytes,
                            void *stream);
int tohip_traj_pshard_local(const void *packed, int64_t n_local, int64_t n_global, int64_t n_wps, const tohip_camera *cam_host,
                            const tohip_rig *rig_host, int flags, const uint32_t *occlusion_bits, float *lo_sum, float *minmax,
                            float *rewards, double *partial, void *workspace, size_t workspace_bytes, void *stream);
int tohip_traj_pshard_finish(int64_t n_local, int64_t n_global, int64_t n_wps, const tohip_camera *cam_host,
                             const tohip_rig *rig_host, const double *partial, const float *gout, float *scalars,
                             float *poses_grad, float *quats_grad, void *workspace, size_t workspace_bytes, void *stream);

/* ---- a waypoint-sharded step's all-reduce, compacted (the one data-path collective) ------------------------------
 * A rank's partial log-odds vector is exactly zero outside the 256-point slots its forward listed as candidates (6-8 % of the
 * slots on the BASELINE workloads).  Instead of all-reducing N floats: (1) tohip_traj_candidate_flags -> one 0/1 int32 per slot
 * of this rank's last forward (Npad/256 of them, device); the ranks MAX-reduce them (RCCL has no OR); (2) tohip_slot_flags_prefix
 * -> prefix[s] = set flags below slot s, prefix[nslots] = their number = the union's slots (nslots + 1 int32, device); (3)
 * tohip_slots_pack(pack = 1) gathers the union's slots of lo_sum into `compact` (256 floats each, in slot order; at most
 * capacity_slots of them); the ranks sum-reduce compact[0 .. 256 * count); (4) tohip_slots_pack(pack = 0) scatters it back.  The
 * slots outside the union are zero on every rank and are not touched. */
int tohip_traj_candidate_flags(int64_t n_points, int64_t n_virtual, int64_t n_traj, const void *workspace, size_t workspace_bytes,
                               int32_t *slot_flags, void *stream);
int tohip_slot_flags_prefix(const int32_t *slot_flags, int64_t n_points, int32_t *prefix, void *stream);
int tohip_slots_pack(const int32_t *slot_flags, const int32_t *prefix, int64_t n_points, float *lo_sum, float *compact,
                     int64_t capacity_slots, int pack, void *stream);

/* ---- ModelTraj.forward() / loss.backward() as one call each ------------------------------------------
 * The reference's loop (trajectory_optimization.py:109-116) is `optimizer.zero_grad(); loss = model(); loss.backward();
 * optimizer.step()`; one step's kernels take 0.05-0.15 ms on an MI355X, so the loop is bound by the host work between them.
 * tohip_traj_loss describes a model once (HOST struct, caller-owned like everything it points to; the library keeps no state):
 * model() is then tohip_traj_loss_forward — waypoint selection (model.py:214-217), visibility + log-odds (:217-231), rewards
 * (:237), criterion (:244-260) — and loss.backward() is tohip_traj_loss_backward.  One trajectory, no occlusion rows, no
 * sharding (those go through the separate calls above). */
typedef struct tohip_traj_loss {
    const void *packed;      /* tohip_pack_cloud's blob */
    int64_t n_points;
    int64_t n_wps;           /* W: ALL waypoints (criterion uses every one, model.py:249-258) */
    int32_t wps_step;        /* every wps_step-th waypoint is evaluated for visibility (model.py:215-217) */
    int32_t flags;           /* TOHIP_TRAJ_* */
    tohip_camera cam;
    tohip_rig rig;           /* n_cams = 0: one camera at the body frame */
    const float *poses0;     /* (W,3) initial positions (model.py:176) */
    float smoothness_weight; /* model.py:166 */
    float traj_length_weight;
    void *workspace;         /* tohip_traj_workspace_bytes(n_points, n_eval * max(1, n_cams)), n_eval = ceil(W / wps_step); zero-filled once */
    size_t workspace_bytes;
    void *scratch;           /* tohip_traj_loss_scratch_bytes(...) bytes: the step's intermediate vectors (layout below) */
    size_t scratch_bytes;
    float *reg_terms;        /* NULL, or (3, W, 3) floats: the gradients of l2, length and smooth separately (callers that
                                differentiate a single entry of model.loss) */
} tohip_traj_loss;
size_t tohip_traj_loss_scratch_bytes(int64_t n_points, int64_t n_wps, int32_t wps_step, int32_t n_cams);
/* byte offsets into `scratch` of: [0] poses_e (n_eval,3)  [1] quats_e (n_eval,4)  [2] lo_sum (Npad, packed order)
 * [3] minmax (V,2)  [4] scalars (4: mean reward, loss_vis, d loss_vis / d reward, -)  [5] poses_grad_eval (n_eval,3)
 * [6] quats_grad_eval (n_eval,4)  [7] regularisers' gradient (W,3) — for callers that want to look at them. */
int tohip_traj_loss_scratch_layout(int64_t n_points, int64_t n_wps, int32_t wps_step, int32_t n_cams, int64_t *offsets_host);
/* model(): rewards (N floats, caller's point order), loss_terms[0..4] = vis, l2, length, smooth, total (8 floats).  Leaves the
 * step's state in workspace + scratch for tohip_traj_loss_backward (and for tohip_traj_backward with a general dL/d rewards:
 * the workspace is in the state tohip_traj_forward leaves, lo_sum / scalars are in scratch; see tohip_traj_loss_refresh).
 * Four launches; loss.backward() is one. */
int tohip_traj_loss_forward(const tohip_traj_loss *plan_host, const float *poses, const float *quats, float *rewards,
                            float *loss_terms, void *stream);
/* loss.backward(): gout = DEVICE pointer to dL/d loss (autograd's incoming gradient); poses_grad (W,3), quats_grad (W,4) =
 * gout * d loss / d (poses, quats) of the step whose forward last used the plan's workspace and scratch. */
int tohip_traj_loss_backward(const tohip_traj_loss *plan_host, const float *gout, float *poses_grad, float *quats_grad,
                             void *stream);
/* tohip_traj_backward over the plan's workspace (a general dL/d rewards between model() and loss.backward()) OVERWRITES the pair
 * sums tohip_traj_loss_backward reads: they are then scaled by that call's upstream gradient.  Call this before the next
 * tohip_traj_loss_backward of the same step: it takes the unit-gradient sums again (one launch; same pairs, same bits). */
int tohip_traj_loss_refresh(const tohip_traj_loss *plan_host, void *stream);

/* ---- one step of TrajOpt.run (trajectory_optimization.py:100-127) as one call ----------------------------------
 * `optimizer.zero_grad(); loss = model(); loss.backward(); optimizer.step()` and the early-stop bookkeeping (:119-124) for
 * n_traj equal-length trajectories over one cloud, in the FIVE launches of tohip_traj_forward_backward: the waypoint selection
 * (model.py:215-217) is a stride of the first launch's reads, criterion's regularisers (model.py:244-260) with their gradient and
 * the step's Adam constants are extra blocks of the first launch, and every block of the last launch (one per evaluated waypoint)
 * updates its waypoint's rows of poses / quats (torch.optim.Adam, two groups: trajectory_optimization.py:91-94) and, for a
 * trajectory's first waypoint, writes the loss log and the early-stop state.  One camera or a rig; no occlusion rows, no sharding
 * (those use the separate calls + tohip_traj_step_tail).  HOST struct, caller-owned like everything it points to. */
typedef struct tohip_traj_opt {
    const void *packed;        /* tohip_pack_cloud's blob */
    int64_t n_points;
    int64_t n_wps;             /* W: waypoints of ONE trajectory */
    int32_t wps_step;          /* every wps_step-th waypoint is evaluated for visibility */
    int32_t flags;             /* TOHIP_TRAJ_DENSE, TOHIP_TRAJ_OPT_LAST_OUTPUTS or 0 */
    int32_t n_traj;            /* trajectories laid end to end: rows b * W .. b * W + W - 1 of every per-waypoint array */
    int32_t n_steps;           /* rows of the logs below */
    const int32_t *traj_offsets; /* n_traj + 1 device int32: b * ceil(W / wps_step) (NULL when n_traj == 1) */
    tohip_camera cam;
    tohip_rig rig;             /* n_cams = 0: one camera at the body frame */
    float *poses, *quats;      /* (n_traj W, 3), (n_traj W, 4): the parameters, updated in place */
    const float *poses0;       /* (n_traj W, 3) */
    float smoothness_weight, traj_length_weight;
    float lr_pose, lr_quat, beta1, beta2, adam_eps;   /* torch.optim.Adam: two groups, shared betas / eps */
    float rewards_th, smoothness_th;                  /* trajectory_optimization.py:100 */
    float *exp_avg_p, *exp_avg_sq_p, *exp_avg_q, *exp_avg_sq_q;   /* Adam moments, zero before the first step */
    float *poses_grad, *quats_grad;   /* outputs (n_traj W, 3 / 4): the step's full gradients (what .grad would hold) */
    float *poses_grad_eval, *quats_grad_eval; /* outputs, may be NULL: (n_traj n_eval, 3 / 4) visibility gradient rows */
    float *lo_sum;             /* n_traj x Npad (packed order); with TOHIP_TRAJ_OPT_LAST_OUTPUTS valid after the last step only */
    float *minmax;             /* (V, 2), V = n_traj * n_eval * max(1, n_cams) */
    float *rewards;            /* n_traj x N, caller's point order; with TOHIP_TRAJ_OPT_LAST_OUTPUTS valid after the last step only */
    float *scalars;            /* n_traj x 4: mean reward, loss_vis, d loss_vis / d reward, - */
    float *loss_log;           /* n_traj x n_steps x 8: row s of a trajectory = (vis, l2, length, smooth, total) of its s-th step */
    float *state_log;          /* n_traj x (n_steps + 1) x 8, row 0 ZERO before the first step: row i = the early-stop state before
                                  step i ([0] reward0 [1] smooth0 [2] stopped [3] steps taken [4] visibility gain [5] smoothness
                                  gain); step i reads row i and writes row i + 1; row n_steps is the run's result */
    void *workspace;           /* tohip_traj_workspace_bytes_multi(n_points, V, n_traj); zero-filled once */
    size_t workspace_bytes;
    void *scratch;             /* tohip_traj_opt_scratch_bytes(n_wps, n_traj) */
    size_t scratch_bytes;
} tohip_traj_opt;
size_t tohip_traj_opt_scratch_bytes(int64_t n_wps, int64_t n_traj);
/* step_index = 0, 1, ... n_steps - 1, in order.  A trajectory that has stopped (state[2]) stays put; its rewards are still
 * refreshed.  Nothing synchronises. */
int tohip_traj_opt_step(const tohip_traj_opt *opt_host, int32_t step_index, void *stream);

/* ---- ModelPose (model.py:98-127) -------------------------------------------------------------- */
size_t tohip_pose_workspace_bytes(int64_t n_points);

/* observations, occlusion_mask and grad_obs are in the caller's point order.
 * observations[n] = dist_mask*fov_mask (* occlusion_mask[n] when non-NULL, model.py:112-115);
 * scalars[0] = sum(observations), scalars[1] = loss = 1/(sum+eps). */
int tohip_pose_forward(const void *packed, int64_t n_points, const float *trans, const float *quat,
                       const tohip_camera *cam_host, const float *occlusion_mask, float *observations, float *scalars,
                       void *workspace, size_t workspace_bytes, void *stream);
/* Upstream gradient: grad_obs (N floats, dL/d observations) or, when NULL, the fused loss through
 * scalars (from tohip_pose_forward) and gout (device pointer to dL/d loss). */
int tohip_pose_backward(const void *packed, int64_t n_points, const float *trans, const float *quat,
                        const tohip_camera *cam_host, const float *occlusion_mask, const float *grad_obs,
                        const float *scalars, const float *gout, float *trans_grad, float *quat_grad, void *workspace,
                        size_t workspace_bytes, void *stream);

/* tohip_pose_forward + tohip_pose_backward of the fused loss in ONE pass over the cloud (the gradient sums do not depend on the
 * loss: the pass that writes the observations takes them with unit weights, and the finish scales them by -loss^2 gout once the
 * sum is known): observations, scalars, and trans_grad (3) / quat_grad (4) = gout x d loss / d (trans, raw quat); gout = device
 * pointer to dL/d loss, NULL = 1.  Two launches (the pass, its one-block finish).  replaces model.py:98-127 + loss.backward(). */
int tohip_pose_forward_backward(const void *packed, int64_t n_points, const float *trans, const float *quat,
                                const tohip_camera *cam_host, const float *occlusion_mask, float *observations, float *scalars,
                                const float *gout, float *trans_grad, float *quat_grad, void *workspace, size_t workspace_bytes,
                                void *stream);
/* One step of PoseOpt's loop (pose_optimization.py:124-141): the call above, then torch.optim.Adam (two groups: trans @ lr_pose,
 * quat @ lr_quat, pose_optimization.py:93-97) on trans / quat IN PLACE, in the finish launch — still two launches.  step = the
 * 1-based iteration; loss_log[step - 1] = the loss of this step (before the update); trans_grad / quat_grad may be NULL. */
int tohip_pose_opt_step(const void *packed, int64_t n_points, float *trans, float *quat, const tohip_camera *cam_host,
                        const float *occlusion_mask, float *observations, float *scalars, float *trans_grad, float *quat_grad,
                        float *exp_avg_t, float *exp_avg_sq_t, float *exp_avg_q, float *exp_avg_sq_q, float lr_pose, float lr_quat,
                        float beta1, float beta2, float adam_eps, int32_t step, float *loss_log, void *workspace,
                        size_t workspace_bytes, void *stream);

/* ---- element-wise helpers of model.py (forward values) ---------------------------------------- */
/* to_camera_frame (model.py:50-57; normalize=1) / ego_to_cam_torch (pc_processor.py:63-70;
 * normalize=0): bit-identical to the reference's f32 op order.  out_layout 0: (N,3), 1: (3,N). */
int tohip_to_camera_frame(const float *xyz, int64_t n_points, const float *quat, const float *trans, int normalize,
                          int out_layout, float *out, void *stream);
/* get_dist_mask (model.py:13-24) and soft get_fov_mask (model.py:27-47) on (N,3) camera-frame points. */
int tohip_soft_masks(const float *cam_xyz, int64_t n_points, const tohip_camera *cam_host, float *dist_mask,
                     float *fov_mask, void *stream);
/* Their backward passes (the reference's helpers are plain torch ops, differentiable by autograd):
 * grad_xyz (N,3) = grad_dist[n] dD/dp + grad_fov[n] dF/dp (either upstream gradient may be NULL = zero);
 * to_camera_frame: grad_xyz (N,3, may be NULL) = R grad_out, grad_quat (4) w.r.t. the raw quaternion (through F.normalize),
 * grad_trans (3); workspace of tohip_pose_workspace_bytes bytes. */
int tohip_soft_masks_backward(const float *cam_xyz, int64_t n_points, const tohip_camera *cam_host, const float *grad_dist,
                              const float *grad_fov, float *grad_xyz, void *stream);
int tohip_to_camera_frame_backward(const float *xyz, int64_t n_points, const float *quat, const float *trans,
                                   const float *grad_out, float *grad_xyz, float *grad_quat, float *grad_trans,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* ---- hard frustum cull (tools.py:176-187, pc_processor.py:72-83, model.py:34-39) -------------- */
size_t tohip_frustum_workspace_bytes(int64_t n_points);
/* cam_3xN: camera-frame points as (3,N).  dist_mask/fov_mask: N bytes of 0/1 (either may be NULL).
 * kept_idx (capacity N int32, may be NULL): ascending indices with both masks set; *kept_count (device
 * int32) their number.  Bit-exact with the reference's CPU path. */
int tohip_frustum_cull(const float *cam_3xN, int64_t n_points, const tohip_camera *cam_host, float min_dist,
                       float max_dist, uint8_t *dist_mask, uint8_t *fov_mask, int32_t *kept_idx, int32_t *kept_count,
                       void *workspace, size_t workspace_bytes, void *stream);
/* The cull stage of the per-camera pipeline (pc_processor.py:158-170) for n_wps poses at once: exact transform of the cloud
 * (to_camera_frame arithmetic; normalize as in tohip_to_camera_frame), hard frustum test, ordered compaction.
 * kept_idx (n_wps, n) int32 and kept_pts (n_wps, n, 3) f32 (camera frame) receive each pose's kept points in input
 * order in their first kept_count[w] rows (device int32 per pose) — worst-case sized, no host round trip. */
size_t tohip_cull_waypoints_workspace_bytes(int64_t n_points, int64_t n_wps);
int tohip_cull_waypoints(const float *xyz, int64_t n_points, const float *poses, const float *quats, int64_t n_wps,
                         int normalize, const tohip_camera *cam_host, float min_dist, float max_dist, int32_t *kept_idx,
                         float *kept_pts, int32_t *kept_count, void *workspace, size_t workspace_bytes, void *stream);
/* The same with the poses' kept points laid END TO END in kept_pts (what tohip_hidden_pts_removal_batched reads: the occlusion
 * refresh's next stage, pc_processor.py:171-187 per camera): pose w's rows are [seg_off[w], seg_off[w+1]) of kept_pts, seg_off
 * (n_wps + 1 device int64) is written here; kept_idx keeps its (n_wps, n) layout.  kept_pts must still hold n_wps * n rows
 * (nobody knows the counts beforehand).  One launch more than tohip_cull_waypoints, no copy afterwards. */
int tohip_cull_waypoints_packed(const float *xyz, int64_t n_points, const float *poses, const float *quats, int64_t n_wps,
                                int normalize, const tohip_camera *cam_host, float min_dist, float max_dist, int32_t *kept_idx,
                                float *kept_pts, int32_t *kept_count, int64_t *seg_off, void *workspace, size_t workspace_bytes,
                                void *stream);
/* gather rows: out[i,:] = xyz[idx[i],:] for i < *count (device), xyz (N,3) or (3,N) by in_layout. */
int tohip_gather_points(const float *xyz, int64_t n_points, int in_layout, const int32_t *idx, const int32_t *count,
                        int64_t capacity, float *out_xyz, void *stream);

/* ---- hidden-point removal (tools.py:38-85) ------------------------------------------------------ */
/* Recommended workspace.  The hull's face pool takes every byte beyond the per-point part; the recommendation holds
 * n/2 faces (HPR of a scene keeps a few % of the points).  A hull that creates more returns TOHIP_ENOSPC: call again
 * with a larger workspace (4x is what the Python host does; 8 faces per point is never exceeded). */
size_t tohip_hpr_workspace_bytes(int64_t n_points);
/* sphericalFlip (tools.py:38-53): flipped (N,3), radius_out[0] = max||p|| * 10^param. Bit-exact.  workspace: at least
 * TOHIP_FLIP_WORKSPACE_BYTES of scratch (the blocks' maxima of the norms). */
#define TOHIP_FLIP_WORKSPACE_BYTES 8192
int tohip_spherical_flip(const float *xyz, int64_t n_points, float param, float *flipped, float *radius_out,
                         void *workspace, size_t workspace_bytes, void *stream);
/* hidden_pts_removal (tools.py:67-85): flip, convex hull of flipped points + origin (double
 * precision quickhull on the GPU), visible = hull vertices in ascending index order minus the LAST
 * one (the reference drops hull.vertices[-1] unconditionally).  visible_idx capacity N int32;
 * *visible_count device int32; mask (N floats of 0/1, may be NULL).  SYNCHRONISES the stream (the hull
 * is built in rounds whose convergence is read back).  The read-back is a 128-byte record that a one-wave kernel writes into mapped
 * host memory and the calling thread polls: the library keeps two such records per (host thread, device) — the one piece of state it
 * holds between calls, allocated on the thread's first build on that device and freed when the thread ends. */
int tohip_hidden_pts_removal(const float *xyz, int64_t n_points, float param, int32_t *visible_idx,
                             int32_t *visible_count, float *mask, void *workspace, size_t workspace_bytes,
                             void *stream);

/* hidden_pts_removal of n_segments independent clouds in one pass, each seen from its own origin: the
 * per-camera use of HPR in pc_processor.py:171-178 (one segment per camera topic / per waypoint).  xyz holds the
 * segments end to end; segment s = rows [seg_offsets_host[s], seg_offsets_host[s+1]) (HOST array, n_segments+1,
 * seg_offsets_host[0] = 0).  Every segment gets its own flip radius (max norm of ITS points) and its own
 * "drop the last hull vertex".  All hulls advance in the same rounds.
 *   visible_idx          capacity n_total int32: rows of xyz, ascending; segment s's visible rows are
 *                        visible_idx[seg_visible_offsets[s] : seg_visible_offsets[s+1]]
 *   seg_visible_offsets  n_segments+1 int32 (device)
 *   mask                 n_total floats 0/1, or NULL
 *   seg_status           n_segments int32 (device) or NULL: 0 ok; 1 fewer than 4 points, 2 flat — scipy raises
 *                        QhullError for both; 3 a NaN among its (flipped) points —
 *                        scipy raises ValueError; here such a segment reports no visible points
 * SYNCHRONISES the stream. */
size_t tohip_hpr_batched_workspace_bytes(int64_t n_total, int32_t n_segments);
int tohip_hidden_pts_removal_batched(const float *xyz, const int64_t *seg_offsets_host, int32_t n_segments, float param,
                                     int32_t *visible_idx, int32_t *seg_visible_offsets, float *mask, int32_t *seg_status,
                                     void *workspace, size_t workspace_bytes, void *stream);

/* convexHull (tools.py:56-64): ascending hull-vertex indices of pts (n,3), optionally with the origin
 * appended as point n.  idx capacity n+1 int32; *count device int32; *rounds_host (may be NULL) the
 * number of insertion rounds.  SYNCHRONISES the stream. */
int tohip_convex_hull_vertices(const float *pts, int64_t n_points, int with_origin, int32_t *idx, int32_t *count,
                               int32_t *rounds_host, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the O(W) remainder of an optimisation step, on the device (no host sync inside a run) ---------
 * criterion's regularisers (model.py:244-260) with analytic gradients.  loss_terms[0..4] = vis (copied from
 * scalars[1]), l2, length, smooth, total.  grad_poses (W,3), may be NULL: the regularisers' gradient, added to
 * the existing content when accumulate != 0 (i.e. on top of the visibility gradient). */
int tohip_traj_regularizers(const float *poses, const float *poses0, int64_t n_wps, float smoothness_weight,
                            float traj_length_weight, float eps, const float *scalars, float *loss_terms,
                            float *grad_poses, int accumulate, const float *state, float *grad_terms, void *stream);
/* grad_terms (may be NULL): (3, W, 3) floats, the gradients of l2, length and smooth separately (their sum is what
 * grad_poses receives) — for callers that differentiate a single term of model.loss. */
/* state (may be NULL): when given, loss_terms is the base of an (n_steps, 8) log and the row written is
 * state[3] = steps taken so far — the launch then carries no per-step host value and can be replayed from a
 * hipGraph.  Same convention for tohip_adam_step (step <= 0: step index = state[3] + 1) and tohip_early_stop. */
/* rows r*step of a (.., cols) array <-> a compact (n_rows, cols) array: the every-wps_step-th waypoint selection
 * of model.py:217 (scatter = 0: gather src[r*step] -> dst[r]; 1: scatter src[r] -> dst[r*step]). */
int tohip_rows_strided(const float *src, int64_t n_rows, int cols, int step, int scatter, float *dst, void *stream);
/* torch.optim.Adam update of one parameter group (defaults of trajectory_optimization.py:91-94); step is the
 * 1-based iteration; a no-op once state[2] != 0 (early stop reached).  state may be NULL. */
int tohip_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, int32_t step, const float *state, void *stream);
/* The same update for up to TOHIP_ADAM_MAX_GROUPS parameter tensors in ONE launch (the reference's optimiser holds two:
 * poses @ lr_pose, quats @ lr_quat, trajectory_optimization.py:91-94).  groups_host: HOST array. */
#define TOHIP_ADAM_MAX_GROUPS 8
typedef struct tohip_adam_group {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t n;
    float lr, beta1, beta2, eps;
    int32_t step; /* 1-based */
} tohip_adam_group;
int tohip_adam_step_multi(const tohip_adam_group *groups_host, int32_t n_groups, void *stream);
/* early-stop rule of trajectory_optimization.py:100-124 on the device.  state (8 floats, zero-initialised by
 * the caller): [0] reward0 [1] smooth0 [2] stopped [3] steps taken [4] visibility gain [5] smoothness gain. */
int tohip_early_stop(const float *scalars, const float *loss_terms, float rewards_th, float smoothness_th, float *state,
                     int row_from_state, void *stream);

/* The same step remainder in ONE launch (one block; W up to a few thousand): scatter of the evaluated waypoints'
 * visibility gradients (rows r -> waypoint r*step, zero elsewhere) into poses_grad (W,3) / quats_grad (W,4), criterion
 * regularisers and their gradient on top, both Adam updates (step index state[3]+1, skipped once stopped), the loss log
 * row state[3] and the early-stop rule.  Equivalent to tohip_rows_strided x2 + tohip_traj_regularizers +
 * tohip_adam_step x2 + tohip_early_stop with `state`. */
int tohip_traj_step_tail(float *poses, float *quats, const float *poses0, int64_t n_wps, const float *poses_grad_eval,
                         const float *quats_grad_eval, int64_t n_eval, int step, float *poses_grad, float *quats_grad,
                         float *exp_avg_p, float *exp_avg_sq_p, float *exp_avg_q, float *exp_avg_sq_q,
                         float smoothness_weight, float traj_length_weight, float eps, float lr_pose, float lr_quat,
                         float beta1, float beta2, float adam_eps, float rewards_th, float smoothness_th,
                         const float *scalars, float *loss_terms, float *state, void *stream);
/* The same for n_traj equal-length trajectories laid end to end (block b = trajectory b): poses / quats / poses0 / gradients /
 * Adam moments (n_traj x W rows), poses_grad_eval / quats_grad_eval (n_traj x n_eval rows), scalars (n_traj x 4),
 * state (n_traj x 8), loss_terms: trajectory b's log starts at loss_terms + b * loss_terms_stride floats. */
int tohip_traj_step_tail_multi(float *poses, float *quats, const float *poses0, int64_t n_wps, int64_t n_traj,
                               const float *poses_grad_eval, const float *quats_grad_eval, int64_t n_eval, int step,
                               float *poses_grad, float *quats_grad, float *exp_avg_p, float *exp_avg_sq_p, float *exp_avg_q,
                               float *exp_avg_sq_q, float smoothness_weight, float traj_length_weight, float eps, float lr_pose,
                               float lr_quat, float beta1, float beta2, float adam_eps, float rewards_th, float smoothness_th,
                               const float *scalars, float *loss_terms, int64_t loss_terms_stride, float *state, void *stream);
int tohip_gather_waypoints_multi(const float *poses, const float *quats, int64_t n_wps, int64_t n_traj, int64_t n_eval, int step,
                                 float *poses_e, float *quats_e, void *stream);
/* poses_e[r] = poses[r*step], quats_e[r] = quats[r*step] for r < n_eval, in one launch. */
int tohip_gather_waypoints(const float *poses, const float *quats, int64_t n_eval, int step, float *poses_e, float *quats_e,
                           void *stream);

/* ---- input formats (pointcloud_utils.py, launch/voxels_filtering.launch) --------------------------------
 * PointCloud2 payload -> (N,3) f32 with non-finite rows removed, in message order
 * (pointcloud2_to_xyz_array, pointcloud_utils.py:197-198, + the callers' cast to f32).  data: the message's
 * byte buffer on the device, n_points = width*height; x/y/z_off and datatype (7 = FLOAT32, 8 = FLOAT64) from
 * its PointFields.  out_xyz capacity n_points rows (rows at and beyond *out_count are unspecified); *out_count device int32.
 * The workspace may hold anything on entry; REUSE it from message to message: from 2 M points on, one of its words remembers
 * whether the last message had invalid rows, and a message after one without any is unpacked with one read of its bytes instead of
 * two (16 M points: 94 instead of 140 us; same output either way — the hint only chooses the schedule). */
size_t tohip_ingest_workspace_bytes(int64_t n_points);
int tohip_pointcloud2_to_xyz(const uint8_t *data, int64_t n_points, int32_t point_step, int32_t x_off, int32_t y_off,
                             int32_t z_off, int32_t datatype, int32_t is_bigendian, int32_t remove_nans, float *out_xyz,
                             int32_t *out_count, void *workspace, size_t workspace_bytes, void *stream);
/* pcl::VoxelGrid as configured by launch/voxels_filtering.launch:11-21: drop non-finite points and points whose
 * filter field (0/1/2 = x/y/z, -1 = none) lies outside [limit_min, limit_max]; one centroid per occupied voxel,
 * voxels in ascending key order.  out_xyz capacity n rows; *out_count device int32: the number of voxels, or -1 when the grid
 * would have more than 2^31 - 1 cells (PCL: "leaf size is too small", it returns its input unfiltered). */
size_t tohip_voxel_grid_workspace_bytes(int64_t n_points);
int tohip_voxel_grid(const float *xyz, int64_t n_points, float leaf_x, float leaf_y, float leaf_z, int32_t filter_field,
                     float limit_min, float limit_max, float *out_xyz, int32_t *out_count, void *workspace,
                     size_t workspace_bytes, void *stream);
/* pc_to_voxel (pointcloud_utils.py:279-288): float64 occupancy grid (nx,ny,nz), 1.0 where a point falls.
 * pc: (n, cols>=3) f32 rows. */
int tohip_pc_to_voxel(const float *pc, int64_t n_points, int32_t cols, double resolution, double x0, double x1, double y0,
                      double y1, double z0, double z1, int32_t nx, int32_t ny, int32_t nz, double *voxel, void *stream);

/* ---- frustum rasterisation (render_pc_image, tools.py:122-173; pulsar itself cannot be pinned, see
 * render_kernels.hip): nearest-depth sphere splat of camera-frame points.  K9_host: row-major intrinsics in
 * HOST memory; image (height,width,3) f32; owner (may be NULL) (height,width) int32 winning point or -1;
 * owns_pixel (may be NULL) n int32 flags: the z-buffer visibility set. */
size_t tohip_render_workspace_bytes(int32_t width, int32_t height);
int tohip_render_points(const float *verts, int64_t n_points, const float *K9_host, int32_t width, int32_t height,
                        float radius, float znear, float zfar, float background, float *image, int32_t *owner,
                        int32_t *owns_pixel, void *workspace, size_t workspace_bytes, void *stream);

/* render_pc_image's `gamma` (tools.py:122, 160-171: pulsar's blending softness): every disc covering a pixel centre contributes with
 * weight (1 - distance to the disc's centre / its radius) * exp(normalised depth / gamma), the background with exp(0); the statement
 * of the blend is in render_kernels.hip (parity unpinned).  Deterministic: integer atomics in fixed point.  gamma > 0, zfar > znear. */
size_t tohip_render_blend_workspace_bytes(int32_t width, int32_t height);
int tohip_render_points_blend(const float *verts, int64_t n_points, const float *K9_host, int32_t width, int32_t height, float radius,
                              float znear, float zfar, float gamma, float background, float *image, void *workspace,
                              size_t workspace_bytes, void *stream);

/* The z-buffer visibility sets of n_clouds camera-frame clouds at once (tohip_render_points' owns_pixel for each): cloud w = the
 * first count[w] (device int32) rows of verts + w * n_stride * 3 — the layout tohip_cull_waypoints writes — with its own z-buffer;
 * visible[w * n_stride + j] = 1.0f when its point j owns a pixel, else 0.  Clouds go through in chunks of as many z-buffers as the
 * workspace holds (tohip_zbuffer_batched_workspace_bytes: up to 2 GB): three launches per chunk. */
size_t tohip_zbuffer_batched_workspace_bytes(int32_t width, int32_t height, int64_t n_clouds);
int tohip_zbuffer_visible_batched(const float *verts, int64_t n_stride, const int32_t *count, int64_t n_clouds, const float *K9_host,
                                  int32_t width, int32_t height, float radius, float znear, float zfar, float *visible,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* ---- optional per-kernel timing (bench.py's roofline leg) -----------------------------------------
 * When enabled, every launch of the big kernels is bracketed by hipEventRecord on its own stream.
 * tohip_profile_read synchronises on those events and returns, per kernel id < TOHIP_PROF_NKERNELS,
 * the summed milliseconds and the number of launches since the last read (HOST arrays). */
#define TOHIP_PROF_NKERNELS 6
int tohip_profile_enable(int on);
const char *tohip_profile_name(int id);
int tohip_profile_read(double *ms_sum_host, int64_t *counts_host);
/* Diagnostic: while a device buffer is set, every block of k_traj_pass1 (the dense kernel) stores two 64-bit words — its
 * lifetime in shader-clock ticks (s_memtime) and in 100 MHz ticks (s_memrealtime) — at [2*block], then four words per block
 * (start, end, HW_ID, XCC_ID) from [2*grid]; grid = tohip_profile_clock_blocks(n_points, n_virtual, flags, with_occlusion) blocks
 * (6 words each); NULL switches it off.  Their quotient x 100 MHz is the clock
 * the chip actually holds under this kernel (it gives clock back under load).  Never set during a timed pass. */
int tohip_profile_clock(void *device_buffer);
int64_t tohip_profile_clock_blocks(int64_t n_points, int64_t n_virtual, int flags, int with_occlusion);

/* Diagnostic: what the last forward over `workspace` found.  stats (DEVICE int64 x 5, zero-filled by the caller):
 * [0] flagged (256-point slot, virtual waypoint) pairs — the pairs with a non-zero log-odds term or a gradient;
 * [1] candidate (slot, trajectory) items of pass 1; [2] slots; [3] virtual waypoints; [4] the (slot, waypoint) pairs the last
 * culled pass 1 evaluated (stale after a dense one). */
int tohip_traj_step_stats(int64_t n_points, int64_t n_virtual, int64_t n_traj, const void *workspace, size_t workspace_bytes,
                          int64_t *stats, void *stream);

/* ---- self tests of cross-lane primitives (used by tests/, cheap) -------------------------------- */
int tohip_selftest_wave_reduce(const float *in64xK, int32_t k, float *out_sum, float *out_min, float *out_max,
                               void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRAJOPT_HIP_H */
