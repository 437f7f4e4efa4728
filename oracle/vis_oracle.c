/*
 * vis_oracle.c — CPU restatement of the reference's visibility / coverage-reward path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under trajectory_optimization_amd/ may link, load or
 * call this file; it exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg have an independent checker for the HIP kernels.  It is pinned against
 * golden vectors produced by the reference itself (tests/golden/make_golden.py).
 *
 * Every routine follows the reference's arithmetic in the reference's op order, in the
 * precision REAL (compiled twice: float = what the reference computes, double = the same
 * formulas as a tighter yardstick).  Compile with -ffp-contract=off: the reference's
 * element-wise torch ops round after every operation; the only fused multiply-adds are the
 * k-ordered FMA chain of its 3x3 CPU sgemm (K @ points), written with fma() explicitly.
 *
 *   to_camera_frame   /root/reference/src/model.py:50-57   (+ pytorch3d quaternion_apply,
 *                     restated: q (x) (0,v) (x) conj(q), Hamilton products left to right)
 *   get_dist_mask     /root/reference/src/model.py:13-24
 *   get_fov_mask      /root/reference/src/model.py:27-47
 *   ModelPose         /root/reference/src/model.py:98-127
 *   ModelTraj.forward /root/reference/src/model.py:200-242  (visibility term of criterion :246)
 *   backward          torch autograd of the above, written out analytically
 *   get_cam_frustum_pts /root/reference/src/tools.py:176-187
 *   ego_to_cam_torch  /root/reference/src/pc_processor.py:63-70
 *   sphericalFlip     /root/reference/src/tools.py:38-53
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#error "compile with -DREAL=float|double -DSFX=f32|f64"
#endif

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SFX)

#define IS_F32 (sizeof(REAL) == 4)
static inline REAL r_exp(REAL x) { return IS_F32 ? (REAL)expf((float)x) : (REAL)exp((double)x); }
static inline REAL r_log(REAL x) { return IS_F32 ? (REAL)logf((float)x) : (REAL)log((double)x); }
static inline REAL r_sqrt(REAL x) { return IS_F32 ? (REAL)sqrtf((float)x) : (REAL)sqrt((double)x); }
static inline REAL r_fma(REAL a, REAL b, REAL c) {
    return IS_F32 ? (REAL)fmaf((float)a, (float)b, (float)c) : (REAL)fma((double)a, (double)b, (double)c);
}

/* Per-waypoint camera, prepared once per waypoint exactly as to_camera_frame does. */
typedef struct {
    REAL qi[4]; /* q_inv = normalize(q) * (1,-1,-1,-1)        model.py:53-54 */
    REAL qn[4]; /* invert(q_inv) = normalize(q)               quaternion_apply's second factor */
    REAL t[3];
    REAL nrm;   /* max(||q||, 1e-12)                            F.normalize */
} FN(cam_t);

typedef struct {
    REAL K[9];
    REAL w, h;       /* img_width, img_height */
    REAL halfw, halfh;
    REAL mean, std;  /* (min+max)/2, (max-min)/2                model.py:20-21 */
    REAL eps;        /* float32(1e-6)                           model.py:93,188 */
    REAL clip_hi;    /* float32(1.0 - 1e-6) as torch.clip casts it, model.py:229 */
} FN(consts_t);

static void FN(make_consts)(FN(consts_t) * c, const float *K, float img_w, float img_h, float min_dist, float max_dist) {
    for (int i = 0; i < 9; ++i) c->K[i] = (REAL)K[i];
    c->w = (REAL)img_w;
    c->h = (REAL)img_h;
    c->halfw = (REAL)(float)((double)img_w / 2.0);
    c->halfh = (REAL)(float)((double)img_h / 2.0);
    c->mean = (REAL)(float)(((double)min_dist + (double)max_dist) / 2.0);
    c->std = (REAL)(float)(((double)max_dist - (double)min_dist) / 2.0);
    c->eps = (REAL)(float)1e-6;
    c->clip_hi = (REAL)(float)(1.0 - 1e-6);
}

static void FN(make_cam)(FN(cam_t) * cam, const float *quat, const float *trans, int normalize) {
    REAL q[4] = {(REAL)quat[0], (REAL)quat[1], (REAL)quat[2], (REAL)quat[3]};
    REAL n = 1;
    if (normalize) {
        REAL ss = q[0] * q[0];
        ss = ss + q[1] * q[1];
        ss = ss + q[2] * q[2];
        ss = ss + q[3] * q[3];
        n = r_sqrt(ss);
        if (n < (REAL)1e-12) n = (REAL)1e-12;
        for (int i = 0; i < 4; ++i) q[i] = q[i] / n;
    }
    cam->nrm = n;
    for (int i = 0; i < 4; ++i) cam->qn[i] = q[i];
    cam->qi[0] = q[0];
    cam->qi[1] = -q[1];
    cam->qi[2] = -q[2];
    cam->qi[3] = -q[3];
    for (int i = 0; i < 3; ++i) cam->t[i] = (REAL)trans[i];
}

/* quaternion_apply(q_inv, x - t): two raw Hamilton products, each term rounded, left to right */
static inline void FN(to_cam)(const FN(cam_t) * cam, const REAL x[3], REAL c[3]) {
    const REAL aw = cam->qi[0], ax = cam->qi[1], ay = cam->qi[2], az = cam->qi[3];
    const REAL bw = 0, bx = x[0] - cam->t[0], by = x[1] - cam->t[1], bz = x[2] - cam->t[2];
    const REAL ow = aw * bw - ax * bx - ay * by - az * bz;
    const REAL ox = aw * bx + ax * bw + ay * bz - az * by;
    const REAL oy = aw * by - ax * bz + ay * bw + az * bx;
    const REAL oz = aw * bz + ax * by - ay * bx + az * bw;
    const REAL cw = cam->qn[0], cx = cam->qn[1], cy = cam->qn[2], cz = cam->qn[3];
    c[0] = ow * cx + ox * cw + oy * cz - oz * cy;
    c[1] = ow * cy - ox * cz + oy * cw + oz * cx;
    c[2] = ow * cz + ox * cy - oy * cx + oz * cw;
}

/* K @ c as the reference's CPU sgemm evaluates it: k-ordered FMA chain (pinned bit-exactly
 * against the fixture hard_pipeline_bundled.npz, see tests/test_oracle_golden.py) */
static inline void FN(project)(const FN(consts_t) * k, const REAL c[3], REAL h[3]) {
    for (int i = 0; i < 3; ++i) h[i] = r_fma(k->K[3 * i + 2], c[2], r_fma(k->K[3 * i + 1], c[1], k->K[3 * i] * c[0]));
}

typedef struct {
    REAL D, S, Gw, Gh, p, u, v, z;
} FN(vis_t);

static inline REAL FN(soft_vis)(const FN(consts_t) * k, const REAL c[3], FN(vis_t) * o) {
    /* get_dist_mask */
    const REAL dx = c[0] - k->mean, dy = c[1] - k->mean, dz = c[2] - k->mean;
    /* torch.linalg.norm(dim=1) on CPU accumulates with an FMA chain (pinned bit-exactly on the
     * sphericalFlip fixture, which uses the same norm) */
    const REAL dist = r_sqrt(r_fma(dz, dz, r_fma(dy, dy, dx * dx)));
    const REAL ds = dist / k->std;
    const REAL D = r_exp((REAL)-0.5 * (ds * ds));
    /* get_fov_mask (soft branch) */
    REAL h[3];
    FN(project)(k, c, h);
    const REAL S = (REAL)1 / ((REAL)1 + r_exp(-h[2]));
    const REAL z = h[2] + k->eps;
    const REAL u = h[0] / z, v = h[1] / z;
    const REAL au = (u - k->halfw) / k->w, av = (v - k->halfh) / k->h;
    const REAL Gw = r_exp((REAL)-0.5 * (au * au));
    const REAL Gh = r_exp((REAL)-0.5 * (av * av));
    const REAL fov = S * Gw * Gh;
    const REAL p = D * fov;
    if (o) {
        o->D = D; o->S = S; o->Gw = Gw; o->Gh = Gh; o->p = p; o->u = u; o->v = v; o->z = z;
    }
    return p;
}

/* d p / d c  (camera-frame point), SURVEY.md §8a row G */
static inline void FN(dvis_dc)(const FN(consts_t) * k, const REAL c[3], const FN(vis_t) * s, REAL g[3]) {
    if (!(s->p > 0)) { g[0] = g[1] = g[2] = 0; return; }
    const REAL s2 = k->std * k->std;
    const REAL gw = -(s->u - k->halfw) / (k->w * k->w);
    const REAL gh = -(s->v - k->halfh) / (k->h * k->h);
    const REAL a0 = gw / s->z, a1 = gh / s->z;
    const REAL a2 = ((REAL)1 - s->S) - (gw * s->u + gh * s->v) / s->z;
    for (int j = 0; j < 3; ++j) {
        const REAL kta = k->K[j] * a0 + k->K[3 + j] * a1 + k->K[6 + j] * a2; /* (K^T a)_j */
        g[j] = s->p * (-(c[j] - k->mean) / s2 + kta);
    }
}

/* rotation of the unit quaternion qn (camera -> world), c = R^T (x - t) */
static void FN(rot_from_q)(const REAL q[4], REAL R[9]) {
    const REAL w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z);         R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);         R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);         R[7] = 2 * (y * z + w * x);         R[8] = w * w - x * x - y * y + z * z;
}

/* chain (sum_n dL/dc_n, sum_n y_n dL/dc_n^T) -> (dL/dt, dL/dq_raw) */
static void FN(pose_chain)(const FN(cam_t) * cam, const double Gt[3], const double GR[9], REAL *dt, REAL *dq) {
    REAL R[9];
    FN(rot_from_q)(cam->qn, R);
    for (int j = 0; j < 3; ++j) dt[j] = (REAL)(-(R[3 * j] * Gt[0] + R[3 * j + 1] * Gt[1] + R[3 * j + 2] * Gt[2]));
    const double w = cam->qn[0], x = cam->qn[1], y = cam->qn[2], z = cam->qn[3];
#define A(j, i) GR[3 * (j) + (i)]
    double dh[4];
    dh[0] = 2 * (w * (A(0, 0) + A(1, 1) + A(2, 2)) + z * (A(1, 0) - A(0, 1)) + y * (A(0, 2) - A(2, 0)) + x * (A(2, 1) - A(1, 2)));
    dh[1] = 2 * (x * (A(0, 0) - A(1, 1) - A(2, 2)) + y * (A(0, 1) + A(1, 0)) + z * (A(0, 2) + A(2, 0)) + w * (A(2, 1) - A(1, 2)));
    dh[2] = 2 * (y * (-A(0, 0) + A(1, 1) - A(2, 2)) + x * (A(0, 1) + A(1, 0)) + w * (A(0, 2) - A(2, 0)) + z * (A(1, 2) + A(2, 1)));
    dh[3] = 2 * (z * (-A(0, 0) - A(1, 1) + A(2, 2)) + w * (A(1, 0) - A(0, 1)) + x * (A(0, 2) + A(2, 0)) + y * (A(1, 2) + A(2, 1)));
#undef A
    /* through F.normalize: (I - qn qn^T) / ||q|| */
    const double dot = w * dh[0] + x * dh[1] + y * dh[2] + z * dh[3];
    dq[0] = (REAL)((dh[0] - w * dot) / cam->nrm);
    dq[1] = (REAL)((dh[1] - x * dot) / cam->nrm);
    dq[2] = (REAL)((dh[2] - y * dot) / cam->nrm);
    dq[3] = (REAL)((dh[3] - z * dot) / cam->nrm);
}

/* ------------------------------------------------------------------ ModelTraj */

/* Forward of the visibility term over the W evaluated waypoints (caller has applied wps_step).
 * xyz: (N,3) row-major f32.  Outputs (caller-allocated): lo_sum[N], rewards[N], pmin[W] (=min p),
 * pmax[W] (=max(p - min p)), *mean_reward, *loss_vis.  scratch: N REALs. */
int FN(oracle_traj_forward)(const float *xyz, int64_t N, const float *poses, const float *quats, int64_t W,
                            const float *K, float img_w, float img_h, float min_dist, float max_dist, const float *occ,
                            REAL *lo_sum, REAL *rewards, REAL *pmin, REAL *pmax, double *mean_reward, double *loss_vis) {
    /* occ (may be NULL): (W,N) 0/1 occlusion masks multiplied into p, the per-waypoint analogue of
     * ModelPose's `mask = occlusion_mask * mask` (model.py:112-115); SURVEY.md 8f.3 */
    FN(consts_t) k;
    FN(make_consts)(&k, K, img_w, img_h, min_dist, max_dist);
    REAL *p = (REAL *)malloc(sizeof(REAL) * (size_t)(N > 0 ? N : 1));
    if (!p) return -1;
    for (int64_t n = 0; n < N; ++n) lo_sum[n] = 0;
    for (int64_t w = 0; w < W; ++w) {
        FN(cam_t) cam;
        FN(make_cam)(&cam, quats + 4 * w, poses + 3 * w, 1);
        REAL a = INFINITY;
        int has_nan = 0;
#pragma omp parallel for reduction(min : a) reduction(| : has_nan) schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            REAL x[3] = {(REAL)xyz[3 * n], (REAL)xyz[3 * n + 1], (REAL)xyz[3 * n + 2]}, c[3];
            FN(to_cam)(&cam, x, c);
            p[n] = FN(soft_vis)(&k, c, NULL);
            if (occ) p[n] = (REAL)occ[w * N + n] * p[n];
            if (p[n] < a) a = p[n];
            if (p[n] != p[n]) has_nan = 1;
        }
        if (has_nan) a = (REAL)NAN; /* torch.min() / max() propagate a NaN (a NaN or inf coordinate in the cloud): model.py:226 */
        REAL M = -INFINITY;
#pragma omp parallel for reduction(max : M) schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            p[n] = p[n] - a; /* model.py:226 */
            if (p[n] > M) M = p[n];
        }
        if (has_nan) M = (REAL)NAN;
        pmin[w] = a;
        pmax[w] = M;
#pragma omp parallel for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            REAL ph = p[n] / M; /* :227 */
            ph = ph < (REAL)0.5 ? (REAL)0.5 : (ph > k.clip_hi ? k.clip_hi : ph); /* :229 (a NaN passes through, as in torch.clip) */
            const REAL lo = r_log(ph / ((REAL)1 - ph)); /* :230 */
            lo_sum[n] = lo_sum[n] + lo;                  /* :231 */
        }
    }
    double acc = 0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        rewards[n] = (REAL)1 / ((REAL)1 + r_exp(-lo_sum[n])); /* :237 */
        acc += (double)rewards[n];
    }
    const double mean = (double)(REAL)(acc / (double)(N > 0 ? N : 1));
    *mean_reward = mean;
    *loss_vis = (double)((REAL)1 / ((REAL)mean + k.eps)); /* :246 */
    free(p);
    return 0;
}

/* Diagnostic knob of the backward below (tests/test_hip_conditioning.py): the lower activity threshold of the clipped log-odds
 * (model.py:229: p_hat >= 1/2 carries gradient) moved by `shift`.  The gradient with the threshold at 1/2 - d minus the one with
 * it at 1/2 + d is what the points within d of the threshold are worth — the amount by which two correct f32 evaluations of the
 * reference's formula may differ on a waypoint that has such a point.  0 = the reference's rule. */
static double FN(g_act_shift) = 0.0;
void FN(oracle_set_act_shift)(double shift) { FN(g_act_shift) = shift; }
/* A second diagnostic knob (tools/stress_models.py): EVERY p_hat of the backward moved by `shift` — both activity thresholds and
 * the weight 1 / (p_hat (1 - p_hat)), which amplifies an error of p_hat by 1 / (1 - p_hat) just below the upper threshold.  The
 * gradient at +d minus the one at -d is what an uncertainty of d in p_hat (f32: a few 1e-7) is worth to a waypoint: it matters where
 * a handful of points carry a waypoint's whole gradient and one of them sits at p_hat = 0.999...  0 = the reference's rule. */
static double FN(g_phat_shift) = 0.0;
void FN(oracle_set_phat_shift)(double shift) { FN(g_phat_shift) = shift; }

/* The sums of one waypoint that everything after is linear in, for given extrema (a = min p, M = max p - a: the waypoint's
 * own, or — a point-sharded run, tests/test_distributed_cpu.py — those over ALL ranks' points) and dL/d reward_n = coef for every n:
 *   out[0..2] sum G/M dp/dc   [3..11] sum y (x) G/M dp/dc   [12] S1 = sum G (p_hat - 1)/M   [13] S2 = sum G (-p_hat)/M
 *   [14..25] the argmin set's sum dp/dc, sum y (x) dp/dc, [26] its size      [27..38], [39] the same for the argmax set
 * p: scratch of N REALs.  All of it is additive over the points. */
static void FN(bwd_sums)(const FN(consts_t) *kp, const FN(cam_t) *camp, const float *xyz, int64_t N, const float *occ_row,
                         const REAL *rewards, REAL a, REAL M, double coef, REAL *p, double out[40]) {
    const FN(consts_t) k = *kp;
    const FN(cam_t) cam = *camp;
    double S1 = 0, S2 = 0, Gt[3] = {0, 0, 0}, GR[9] = {0};
    double At_min[3] = {0}, AR_min[9] = {0}, At_max[3] = {0}, AR_max[9] = {0};
    long n_min = 0, n_max = 0;
#pragma omp parallel
    {
        double s1 = 0, s2 = 0, gt[3] = {0}, gr[9] = {0}, atn[3] = {0}, arn[9] = {0}, atx[3] = {0}, arx[9] = {0};
        long cmin = 0, cmax = 0;
#pragma omp for schedule(static) nowait
        for (int64_t n = 0; n < N; ++n) {
            const REAL pp = p[n] - a;
            const REAL ph0 = pp / M;
            const int is_min = (p[n] == a), is_max = (pp == M);
            const REAL ph = FN(g_phat_shift) != 0.0 ? (REAL)((double)ph0 + FN(g_phat_shift)) : ph0;
            const int act = ((double)ph >= 0.5 + FN(g_act_shift) && ph <= k.clip_hi);
            if (!act && !is_min && !is_max) continue;
            REAL x[3] = {(REAL)xyz[3 * n], (REAL)xyz[3 * n + 1], (REAL)xyz[3 * n + 2]}, c[3], g[3];
            FN(vis_t) s;
            FN(to_cam)(&cam, x, c);
            FN(soft_vis)(&k, c, &s);
            FN(dvis_dc)(&k, c, &s, g);
            if (occ_row) for (int i = 0; i < 3; ++i) g[i] = (REAL)occ_row[n] * g[i]; /* d(occ*p)/dc */
            const double y[3] = {(double)x[0] - (double)cam.t[0], (double)x[1] - (double)cam.t[1], (double)x[2] - (double)cam.t[2]};
            if (act) {
                const double r = (double)rewards[n];
                const double G = coef * r * (1.0 - r) / ((double)ph * (1.0 - (double)ph));
                s1 += G * ((double)ph - 1.0) / (double)M;
                s2 += G * (-(double)ph) / (double)M;
                const double wgt = G / (double)M;
                for (int i = 0; i < 3; ++i) gt[i] += wgt * (double)g[i];
                for (int j = 0; j < 3; ++j)
                    for (int i = 0; i < 3; ++i) gr[3 * j + i] += wgt * y[j] * (double)g[i];
            }
            if (is_min) {
                ++cmin;
                for (int i = 0; i < 3; ++i) atn[i] += (double)g[i];
                for (int j = 0; j < 3; ++j)
                    for (int i = 0; i < 3; ++i) arn[3 * j + i] += y[j] * (double)g[i];
            }
            if (is_max) {
                ++cmax;
                for (int i = 0; i < 3; ++i) atx[i] += (double)g[i];
                for (int j = 0; j < 3; ++j)
                    for (int i = 0; i < 3; ++i) arx[3 * j + i] += y[j] * (double)g[i];
            }
        }
#pragma omp critical
        {
            S1 += s1; S2 += s2; n_min += cmin; n_max += cmax;
            for (int i = 0; i < 3; ++i) { Gt[i] += gt[i]; At_min[i] += atn[i]; At_max[i] += atx[i]; }
            for (int i = 0; i < 9; ++i) { GR[i] += gr[i]; AR_min[i] += arn[i]; AR_max[i] += arx[i]; }
        }
    }
    for (int i = 0; i < 3; ++i) { out[i] = Gt[i]; out[14 + i] = At_min[i]; out[27 + i] = At_max[i]; }
    for (int i = 0; i < 9; ++i) { out[3 + i] = GR[i]; out[17 + i] = AR_min[i]; out[30 + i] = AR_max[i]; }
    out[12] = S1; out[13] = S2; out[26] = (double)n_min; out[39] = (double)n_max;
}

/* sums (scaled by `scale`: the dL/d reward factor when they were taken with coef = 1) -> the waypoint's gradient: the shares of
 * the argmin / argmax sets (torch splits the gradient of min() / max() evenly among ties), then the chain to (position, quaternion) */
static void FN(bwd_final)(const FN(cam_t) *cam, const double in[40], double scale, REAL *pg, REAL *qg) {
    double Gt[3], GR[9];
    const double S1 = scale * in[12], S2 = scale * in[13];
    const double wmin = in[26] > 0 ? S1 / in[26] : 0.0, wmax = in[39] > 0 ? S2 / in[39] : 0.0;
    for (int i = 0; i < 3; ++i) Gt[i] = scale * in[i] + wmin * in[14 + i] + wmax * in[27 + i];
    for (int i = 0; i < 9; ++i) GR[i] = scale * in[3 + i] + wmin * in[17 + i] + wmax * in[30 + i];
    FN(pose_chain)(cam, Gt, GR, pg, qg);
}

/* p of every point for one waypoint into p[]; returns its min (NaN when some p is NaN: torch.min() propagates it, model.py:226) */
static REAL FN(eval_waypoint)(const FN(consts_t) *k, const FN(cam_t) *cam, const float *xyz, int64_t N, const float *occ_row, REAL *p,
                              int *has_nan_out) {
    REAL a = INFINITY;
    int has_nan = 0;
#pragma omp parallel for reduction(min : a) reduction(| : has_nan) schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        REAL x[3] = {(REAL)xyz[3 * n], (REAL)xyz[3 * n + 1], (REAL)xyz[3 * n + 2]}, c[3];
        FN(to_cam)(cam, x, c);
        p[n] = FN(soft_vis)(k, c, NULL);
        if (occ_row) p[n] = (REAL)occ_row[n] * p[n];
        if (p[n] < a) a = p[n];
        if (p[n] != p[n]) has_nan = 1;
    }
    if (has_nan) a = (REAL)NAN;
    *has_nan_out = has_nan;
    return a;
}

/* Backward of loss_vis w.r.t. the evaluated waypoints' (poses, quats).  rewards/pmin/pmax from the
 * forward; gout = dL/d loss_vis.  Outputs poses_grad[W*3], quats_grad[W*4]. */
int FN(oracle_traj_backward)(const float *xyz, int64_t N, const float *poses, const float *quats, int64_t W,
                             const float *K, float img_w, float img_h, float min_dist, float max_dist, const float *occ,
                             const REAL *rewards, double mean_reward, double gout, REAL *poses_grad, REAL *quats_grad) {
    FN(consts_t) k;
    FN(make_consts)(&k, K, img_w, img_h, min_dist, max_dist);
    REAL *p = (REAL *)malloc(sizeof(REAL) * (size_t)(N > 0 ? N : 1));
    if (!p) return -1;
    const double vis = 1.0 / (mean_reward + (double)k.eps);
    const double coef = -gout * vis * vis / (double)N; /* dL/d reward_n */
    for (int64_t w = 0; w < W; ++w) {
        FN(cam_t) cam;
        FN(make_cam)(&cam, quats + 4 * w, poses + 3 * w, 1);
        int has_nan;
        const REAL a = FN(eval_waypoint)(&k, &cam, xyz, N, occ ? occ + w * N : NULL, p, &has_nan);
        REAL M = -INFINITY;
#pragma omp parallel for reduction(max : M) schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            const REAL pp = p[n] - a;
            if (pp > M) M = pp;
        }
        if (has_nan) M = (REAL)NAN;
        double sums[40];
        FN(bwd_sums)(&k, &cam, xyz, N, occ ? occ + w * N : NULL, rewards, a, M, coef, p, sums);
        FN(bwd_final)(&cam, sums, 1.0, poses_grad + 3 * w, quats_grad + 4 * w);
        /* autograd through a NaN min / max, through p_hat = 0 / 0 of a waypoint whose p are all equal (in f32: all underflowed to 0),
         * or from a NaN loss (some waypoint was one of those: NaN log-odds for every point): every entry of the waypoint's gradient
         * is NaN — torch multiplies the NaN upstream by the clip's 0 / 1 mask (tests/golden/traj_stress_23_134.npz) */
        if (has_nan || !(M > (REAL)0) || coef != coef) {
            for (int i = 0; i < 3; ++i) poses_grad[3 * w + i] = (REAL)NAN;
            for (int i = 0; i < 4; ++i) quats_grad[4 * w + i] = (REAL)NAN;
        }
    }
    free(p);
    return 0;
}

/* ---- the same path cut where a POINT-sharded run cuts it (every rank a part of the cloud and all the waypoints; the CPU rehearsal
 * of distributed.PointShard in tests/test_distributed_cpu.py) ------------------------------------------------------------------ */
/* this part's extrema per waypoint: pmin[w] = min p, pmax[w] = max p (NOT minus the minimum: ranks combine them with min / max) */
int FN(oracle_traj_extrema)(const float *xyz, int64_t N, const float *poses, const float *quats, int64_t W, const float *K, float img_w,
                            float img_h, float min_dist, float max_dist, REAL *pmin, REAL *pmax) {
    FN(consts_t) k;
    FN(make_consts)(&k, K, img_w, img_h, min_dist, max_dist);
    REAL *p = (REAL *)malloc(sizeof(REAL) * (size_t)(N > 0 ? N : 1));
    if (!p) return -1;
    for (int64_t w = 0; w < W; ++w) {
        FN(cam_t) cam;
        FN(make_cam)(&cam, quats + 4 * w, poses + 3 * w, 1);
        int has_nan;
        pmin[w] = FN(eval_waypoint)(&k, &cam, xyz, N, NULL, p, &has_nan);
        REAL mx = -INFINITY;
        for (int64_t n = 0; n < N; ++n) if (p[n] > mx) mx = p[n];
        pmax[w] = has_nan ? (REAL)NAN : mx;
    }
    free(p);
    return 0;
}

/* log-odds sums and rewards of this part's points for GIVEN extrema (the global ones), model.py:226-237 */
int FN(oracle_traj_forward_ext)(const float *xyz, int64_t N, const float *poses, const float *quats, int64_t W, const float *K,
                                float img_w, float img_h, float min_dist, float max_dist, const REAL *ext_min, const REAL *ext_max,
                                REAL *lo_sum, REAL *rewards) {
    FN(consts_t) k;
    FN(make_consts)(&k, K, img_w, img_h, min_dist, max_dist);
    REAL *p = (REAL *)malloc(sizeof(REAL) * (size_t)(N > 0 ? N : 1));
    if (!p) return -1;
    for (int64_t n = 0; n < N; ++n) lo_sum[n] = 0;
    for (int64_t w = 0; w < W; ++w) {
        FN(cam_t) cam;
        FN(make_cam)(&cam, quats + 4 * w, poses + 3 * w, 1);
        int has_nan;
        (void)FN(eval_waypoint)(&k, &cam, xyz, N, NULL, p, &has_nan);
        const REAL a = ext_min[w], M = ext_max[w] - ext_min[w];
#pragma omp parallel for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            REAL ph = (p[n] - a) / M;
            ph = ph < (REAL)0.5 ? (REAL)0.5 : (ph > k.clip_hi ? k.clip_hi : ph);
            lo_sum[n] = lo_sum[n] + r_log(ph / ((REAL)1 - ph));
        }
    }
    for (int64_t n = 0; n < N; ++n) rewards[n] = (REAL)1 / ((REAL)1 + r_exp(-lo_sum[n]));
    free(p);
    return 0;
}

/* this part's 40 sums per waypoint (bwd_sums) with dL/d reward = 1 for given extrema: partial[W * 40] */
int FN(oracle_traj_backward_partial)(const float *xyz, int64_t N, const float *poses, const float *quats, int64_t W, const float *K,
                                     float img_w, float img_h, float min_dist, float max_dist, const REAL *ext_min, const REAL *ext_max,
                                     const REAL *rewards, double *partial) {
    FN(consts_t) k;
    FN(make_consts)(&k, K, img_w, img_h, min_dist, max_dist);
    REAL *p = (REAL *)malloc(sizeof(REAL) * (size_t)(N > 0 ? N : 1));
    if (!p) return -1;
    for (int64_t w = 0; w < W; ++w) {
        FN(cam_t) cam;
        FN(make_cam)(&cam, quats + 4 * w, poses + 3 * w, 1);
        int has_nan;
        (void)FN(eval_waypoint)(&k, &cam, xyz, N, NULL, p, &has_nan);
        FN(bwd_sums)(&k, &cam, xyz, N, NULL, rewards, ext_min[w], ext_max[w] - ext_min[w], 1.0, p, partial + 40 * w);
    }
    free(p);
    return 0;
}

/* the ranks' sums added up -> gradients; scale = dL/d reward (the same for every point: -gout vis^2 / N_all) */
int FN(oracle_traj_backward_final)(const float *poses, const float *quats, int64_t W, const double *partial, double scale,
                                   REAL *poses_grad, REAL *quats_grad) {
    for (int64_t w = 0; w < W; ++w) {
        FN(cam_t) cam;
        FN(make_cam)(&cam, quats + 4 * w, poses + 3 * w, 1);
        FN(bwd_final)(&cam, partial + 40 * w, scale, poses_grad + 3 * w, quats_grad + 4 * w);
    }
    return 0;
}

/* ------------------------------------------------------------------ ModelPose */

/* observations[n] = D*fov (* mask[n] if mask != NULL), loss = 1/(sum + eps). model.py:98-127 */
int FN(oracle_pose_forward)(const float *xyz, int64_t N, const float *trans, const float *quat, const float *K,
                            float img_w, float img_h, float min_dist, float max_dist, const float *mask,
                            REAL *observations, double *loss) {
    FN(consts_t) k;
    FN(make_consts)(&k, K, img_w, img_h, min_dist, max_dist);
    FN(cam_t) cam;
    FN(make_cam)(&cam, quat, trans, 1);
    double acc = 0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        REAL x[3] = {(REAL)xyz[3 * n], (REAL)xyz[3 * n + 1], (REAL)xyz[3 * n + 2]}, c[3];
        FN(to_cam)(&cam, x, c);
        REAL o = FN(soft_vis)(&k, c, NULL);
        if (mask) o = (REAL)mask[n] * o; /* :115 */
        observations[n] = o;
        acc += (double)o;
    }
    *loss = (double)((REAL)1 / ((REAL)acc + k.eps));
    return 0;
}

int FN(oracle_pose_backward)(const float *xyz, int64_t N, const float *trans, const float *quat, const float *K,
                             float img_w, float img_h, float min_dist, float max_dist, const float *mask,
                             double loss, double gout, REAL *trans_grad, REAL *quat_grad) {
    FN(consts_t) k;
    FN(make_consts)(&k, K, img_w, img_h, min_dist, max_dist);
    FN(cam_t) cam;
    FN(make_cam)(&cam, quat, trans, 1);
    const double coef = -gout * loss * loss;
    double Gt[3] = {0}, GR[9] = {0};
#pragma omp parallel
    {
        double gt[3] = {0}, gr[9] = {0};
#pragma omp for schedule(static) nowait
        for (int64_t n = 0; n < N; ++n) {
            REAL x[3] = {(REAL)xyz[3 * n], (REAL)xyz[3 * n + 1], (REAL)xyz[3 * n + 2]}, c[3], g[3];
            FN(vis_t) s;
            FN(to_cam)(&cam, x, c);
            FN(soft_vis)(&k, c, &s);
            FN(dvis_dc)(&k, c, &s, g);
            const double wgt = coef * (mask ? (double)mask[n] : 1.0);
            const double y[3] = {(double)x[0] - (double)cam.t[0], (double)x[1] - (double)cam.t[1], (double)x[2] - (double)cam.t[2]};
            for (int i = 0; i < 3; ++i) gt[i] += wgt * (double)g[i];
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i) gr[3 * j + i] += wgt * y[j] * (double)g[i];
        }
#pragma omp critical
        {
            for (int i = 0; i < 3; ++i) Gt[i] += gt[i];
            for (int i = 0; i < 9; ++i) GR[i] += gr[i];
        }
    }
    FN(pose_chain)(&cam, Gt, GR, trans_grad, quat_grad);
    return 0;
}

/* ------------------------------------------------------------------ element-wise helpers */

/* to_camera_frame (normalize=1) / ego_to_cam_torch (normalize=0, pc_processor.py:63-70). out: (N,3) */
int FN(oracle_to_camera_frame)(const float *xyz, int64_t N, const float *quat, const float *trans, int normalize, REAL *out) {
    FN(cam_t) cam;
    FN(make_cam)(&cam, quat, trans, normalize);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        REAL x[3] = {(REAL)xyz[3 * n], (REAL)xyz[3 * n + 1], (REAL)xyz[3 * n + 2]};
        FN(to_cam)(&cam, x, out + 3 * n);
    }
    return 0;
}

/* get_dist_mask, get_fov_mask(soft) on camera-frame points (N,3) */
int FN(oracle_soft_masks)(const float *cam_xyz, int64_t N, const float *K, float img_w, float img_h, float min_dist,
                          float max_dist, REAL *dist_mask, REAL *fov_mask) {
    FN(consts_t) k;
    FN(make_consts)(&k, K, img_w, img_h, min_dist, max_dist);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        REAL c[3] = {(REAL)cam_xyz[3 * n], (REAL)cam_xyz[3 * n + 1], (REAL)cam_xyz[3 * n + 2]};
        FN(vis_t) s;
        FN(soft_vis)(&k, c, &s);
        dist_mask[n] = s.D;
        fov_mask[n] = s.S * s.Gw * s.Gh;
    }
    return 0;
}

#if defined(ORACLE_F32_ONLY_PARTS)
/* ------------------------------------------------------------------ hard (boolean) path, f32 only */

/* get_cam_frustum_pts on a (3,N) camera-frame array (tools.py:176-187; get_fov_mask binary branch
 * model.py:34-39 with min_dist=-inf,max_dist=+inf).  Masks are 0/1 bytes. */
int oracle_frustum_masks(const float *cam_3xN, int64_t N, const float *K, float img_w, float img_h, float min_dist,
                         float max_dist, uint8_t *dist_mask, uint8_t *fov_mask) {
    const float *X = cam_3xN, *Y = cam_3xN + N, *Z = cam_3xN + 2 * N;
    const float wl = (float)((double)img_w - 1.0), hl = (float)((double)img_h - 1.0);
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        float h[3];
        for (int i = 0; i < 3; ++i) h[i] = fmaf(K[3 * i + 2], Z[n], fmaf(K[3 * i + 1], Y[n], K[3 * i] * X[n]));
        const float u = h[0] / h[2], v = h[1] / h[2];
        dist_mask[n] = (Z[n] > min_dist) & (Z[n] < max_dist);
        fov_mask[n] = (h[2] > 0.0f) & (u > 1.0f) & (u < wl) & (v > 1.0f) & (v < hl);
    }
    return 0;
}

/* sphericalFlip (tools.py:38-53): norm, radius = max(norm) * 10**param, flipped = 2*((radius-norm)*p)/norm + p */
int oracle_spherical_flip(const float *xyz, int64_t N, double param, float *flipped, float *radius_out) {
    float mx = -INFINITY;
    float *nrm = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    if (!nrm) return -1;
    for (int64_t n = 0; n < N; ++n) {
        const float x = xyz[3 * n], y = xyz[3 * n + 1], z = xyz[3 * n + 2];
        nrm[n] = sqrtf(fmaf(z, z, fmaf(y, y, x * x))); /* torch.linalg.norm(dim=1): FMA chain */
        if (nrm[n] > mx || nrm[n] != nrm[n]) mx = nrm[n]; /* torch.max propagates NaN */
    }
    const float radius = mx * (float)pow(10.0, param); /* 0-d f32 tensor * python float */
    for (int64_t n = 0; n < N; ++n)
        for (int j = 0; j < 3; ++j) {
            const float t = (radius - nrm[n]) * xyz[3 * n + j];
            flipped[3 * n + j] = (2.0f * t) / nrm[n] + xyz[3 * n + j];
        }
    if (radius_out) *radius_out = radius;
    free(nrm);
    return 0;
}
#endif
