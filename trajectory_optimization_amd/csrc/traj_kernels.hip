// traj_kernels.hip — ModelTraj visibility term, forward + analytic backward, for gfx950.
//
// Replaces the per-waypoint Python loop of /root/reference/src/model.py:217-231 (forward),
// :237/:246 (rewards, visibility loss) and its torch-autograd backward (SURVEY.md §8a rows A-C,E,G).
//
// Data layout in HBM
//   cloud     Morton-sorted SoA x|y|z (npad each) + permutation + one bounding sphere per 256 points
//             (packed once, tohip_pack_cloud)                                          16 B/point
//   WayHot    one 64-B record per virtual waypoint (rotation, translation, min, 1/max, cull bound) -> SGPRs
//   lo_sum (sorted order) / rewards (original order)                                    4 B/point each
//   partials  [virtual waypoint][wave slot] min/max pairs (8 B) and gradient sums (64 B)
//
// Kernel shape: one lane owns P consecutive sorted points in registers and loops over the waypoints, whose
// constants arrive through scalar loads; HBM traffic is ~N*16 B per pass, independent of W.
//
// Two evaluation modes with bitwise identical results:
//   DENSE  every (point, waypoint) pair is evaluated (the streaming reference semantics; bench headline)
//   CULL   pairs that provably contribute exactly nothing are skipped: the log-odds of a pair is exactly 0
//          unless p_hat > 0.5, and p <= exp(-0.5 d2/sigma^2) bounds p by the squared distance d2 of the
//          camera-frame point from (mu,mu,mu).  A wave first tests its 256-point bounding sphere, then the
//          per-point d2, and only then evaluates.  The per-waypoint max is searched the same way against a
//          lower bound L <= max found by a strided probe, which also proves min == 0 by exhibiting a zero.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "profile.hpp"

// ---------------------------------------------------------------------------------------------
// virtual waypoint records

// thread per virtual waypoint v = w*C + c.  F.normalize (model.py:53), rig composition
// R_v = R(qn_w) R(q_c), t_v = t_w + R(qn_w) l_c.
// With `minmax` (the backward, which rebuilds the records from the forward's result) the normalisation constants go
// in right away (apply_minmax) and the waypoint's row of the tie accumulators is cleared: one launch
// where there were three.
__device__ __forceinline__ void apply_minmax(WayHot& h, float& auxM, float a, float M, float inv_var, int cull);

__device__ __forceinline__ void prep_waycam(int v, const float* __restrict__ poses, const float* __restrict__ quats, int C,
                                            const float* __restrict__ rig_q, const float* __restrict__ rig_t,
                                            WayHot* __restrict__ hot, WayCold* __restrict__ cold, WayAux* __restrict__ aux,
                                            const float* __restrict__ minmax, float inv_var, int cull,
                                            float* __restrict__ ties) {
    const int w = v / C, c = v - w * C;
    float q[4] = {quats[4 * w], quats[4 * w + 1], quats[4 * w + 2], quats[4 * w + 3]};
    float ss = q[0] * q[0];
    ss = ss + q[1] * q[1];
    ss = ss + q[2] * q[2];
    ss = ss + q[3] * q[3];
    float n = sqrtf(ss);
    n = n < 1e-12f ? 1e-12f : n;
    for (int i = 0; i < 4; ++i) q[i] = q[i] / n;
    if (c == 0) {
        WayCold cd;
        for (int i = 0; i < 4; ++i) cd.qn[i] = q[i];
        cd.nrm = n;
        cd.pad[0] = cd.pad[1] = cd.pad[2] = 0.f;
        cold[w] = cd;
    }
    float Rw[9], R[9];
    quat_to_R(q, Rw);
    float t[3] = {poses[3 * w], poses[3 * w + 1], poses[3 * w + 2]};
    if (rig_q != nullptr) {
        const float qc[4] = {rig_q[4 * c], rig_q[4 * c + 1], rig_q[4 * c + 2], rig_q[4 * c + 3]};
        float Rc[9];
        quat_to_R(qc, Rc);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) R[3 * i + j] = Rw[3 * i] * Rc[j] + Rw[3 * i + 1] * Rc[3 + j] + Rw[3 * i + 2] * Rc[6 + j];
        if (rig_t != nullptr) {
            const float l[3] = {rig_t[3 * c], rig_t[3 * c + 1], rig_t[3 * c + 2]};
            for (int i = 0; i < 3; ++i) t[i] += Rw[3 * i] * l[0] + Rw[3 * i + 1] * l[1] + Rw[3 * i + 2] * l[2];
        }
    } else {
        for (int i = 0; i < 9; ++i) R[i] = Rw[i];
    }
    WayHot h;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) h.m[3 * i + j] = R[3 * j + i];
    h.t[0] = t[0]; h.t[1] = t[1]; h.t[2] = t[2];
    h.a = 0.f; h.invM = 1.f; h.thr = INFINITY; h.sthr = INFINITY;
    float auxM = 1.f;
    if (minmax) apply_minmax(h, auxM, minmax[2 * v], minmax[2 * v + 1], inv_var, cull);
    hot[v] = h;
    if (aux) {
        WayAux a;
        a.M = auxM; a.L = 0.f; a.thr1 = INFINITY; a.sthr1 = INFINITY; a.azero = 0.f;
        a.pad[0] = a.pad[1] = a.pad[2] = 0.f;
        aux[v] = a;
    }
    if (ties) {
        float4* tz = reinterpret_cast<float4*>(ties + (int64_t)v * 32);
        for (int i = 0; i < 8; ++i) tz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__global__ void k_prep_waycams(const float* __restrict__ poses, const float* __restrict__ quats, int W, int C,
                               const float* __restrict__ rig_q, const float* __restrict__ rig_t,
                               WayHot* __restrict__ hot, WayCold* __restrict__ cold, WayAux* __restrict__ aux,
                               const float* __restrict__ minmax = nullptr, float inv_var = 0.f, int cull = 0,
                               float* __restrict__ ties = nullptr) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < W * C) prep_waycam(v, poses, quats, C, rig_q, rig_t, hot, cold, aux, minmax, inv_var, cull, ties);
}

// ---------------------------------------------------------------------------------------------
// point loads: lane owns P consecutive points

template <int P>
__device__ __forceinline__ void load_points(const float* __restrict__ soa, int64_t npad, int64_t base, float (&x)[P],
                                            float (&y)[P], float (&z)[P]) {
    if constexpr (P == 4) {
        const float4 a = *reinterpret_cast<const float4*>(soa + base);
        const float4 b = *reinterpret_cast<const float4*>(soa + npad + base);
        const float4 c = *reinterpret_cast<const float4*>(soa + 2 * npad + base);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w;
        y[0] = b.x; y[1] = b.y; y[2] = b.z; y[3] = b.w;
        z[0] = c.x; z[1] = c.y; z[2] = c.z; z[3] = c.w;
    } else if constexpr (P == 2) {
        const float2 a = *reinterpret_cast<const float2*>(soa + base);
        const float2 b = *reinterpret_cast<const float2*>(soa + npad + base);
        const float2 c = *reinterpret_cast<const float2*>(soa + 2 * npad + base);
        x[0] = a.x; x[1] = a.y; y[0] = b.x; y[1] = b.y; z[0] = c.x; z[1] = c.y;
    } else {
        x[0] = soa[base]; y[0] = soa[npad + base]; z[0] = soa[2 * npad + base];
    }
}

template <int P>
__device__ __forceinline__ void load_vec(const float* __restrict__ src, int64_t base, float (&v)[P]) {
    if constexpr (P == 4) {
        const float4 a = *reinterpret_cast<const float4*>(src + base);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    } else if constexpr (P == 2) {
        const float2 a = *reinterpret_cast<const float2*>(src + base);
        v[0] = a.x; v[1] = a.y;
    } else {
        v[0] = src[base];
    }
}

// Optional per-(virtual waypoint, point) occlusion bits in packed order: row v holds npad bits,
// bit i = 1 when sorted point i is NOT occluded from waypoint v.  om[] = 1.0f / 0.0f multipliers of p; without a
// bit array every multiplier is exactly 1 (p * 1.0f == p, so the unoccluded results do not change by a bit).
template <int P, bool OCC = true>
__device__ __forceinline__ void load_occ(const uint32_t* __restrict__ occ, int64_t occw, int v, int64_t base, float (&om)[P]) {
    if constexpr (!OCC) {
#pragma unroll
        for (int i = 0; i < P; ++i) om[i] = 1.0f;  // compile-time ones: the multiplications fold away
    } else {
        unsigned bits = ~0u;
        if (occ) bits = occ[(int64_t)v * occw + (base >> 5)] >> (unsigned)(base & 31);
#pragma unroll
        for (int i = 0; i < P; ++i) om[i] = ((bits >> i) & 1u) ? 1.0f : 0.0f;
    }
}

// wave-uniform bounding sphere of the 256-point tile this wave's points belong to
__device__ __forceinline__ float4 wave_tile_bound(const CloudView& cv, int64_t base) {
    const int tile = __builtin_amdgcn_readfirstlane((int)(base >> 8));
    float4 b = cv.bounds[tile];
    b.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.x)));
    b.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.y)));
    b.z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.z)));
    b.w = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.w)));
    return b;
}

// One (tile, waypoint) liveness decision, written once so that the kernels that skip a pair and the
// finishing kernel that skips its partial agree bit for bit.  q0..q3 = the waypoint's WayHot as 4 float4.
__device__ __forceinline__ bool tile_live(const float4& q0, const float4& q1, const float4& q2, float thr, float sthr,
                                          const float4& tb, float mean) {
    const float y0 = tb.x - q2.y, y1 = tb.y - q2.z, y2 = tb.z - q2.w;
    const float X = fmaf(q0.z, y2, fmaf(q0.y, y1, q0.x * y0));
    const float Y = fmaf(q1.y, y2, fmaf(q1.x, y1, q0.w * y0));
    const float Z = fmaf(q2.x, y2, fmaf(q1.w, y1, q1.z * y0));
    const float D2 = dist2_mean(X, Y, Z, mean);
    const float bound = fmaf(tb.w, fmaf(2.0f, sthr, tb.w), thr) * 1.00001f;  // (sthr + r)^2, rounded up
    return !(D2 > bound);
}

// The same test for 64 waypoints at once: lane l tests waypoint vc + l against the wave's tile and the
// ballot is the set of waypoints whose active sphere may reach the tile.  FROM_AUX: pass-1 bound (thr1)
// and "min proven zero" flag from the probe; waypoints without the proof always survive (dense search).
template <bool FROM_AUX>
__device__ __forceinline__ unsigned long long tile_survivors(const WayHot* __restrict__ hot, const WayAux* __restrict__ aux,
                                                             int vc, int v1, const float4& tb, float mean) {
    const int v = vc + (int)(threadIdx.x & 63);
    bool ok = false;
    if (v < v1) {
        const float4* hp = reinterpret_cast<const float4*>(hot + v);
        const float4 q0 = hp[0], q1 = hp[1], q2 = hp[2], q3 = hp[3];  // m0..3 | m4..7 | m8 t0 t1 t2 | a invM thr sthr
        float thr = q3.z, sthr = q3.w;
        bool force = false;
        if (FROM_AUX) {
            const WayAux a = aux[v];
            thr = a.thr1; sthr = a.sthr1;
            force = a.azero == 0.f;
        }
        ok = force || tile_live(q0, q1, q2, thr, sthr, tb, mean);
    }
    return __ballot(ok);
}

// ---------------------------------------------------------------------------------------------
// probe (CULL mode): block per virtual waypoint evaluates a strided sample of the sorted cloud.
//   L     = max p over the sample  (a lower bound of the true max: an actual value of p)
//   azero = some sample has p == 0 exactly  =>  min_n p == 0 (p is never negative)

// The block first builds its waypoint's record itself (k_prep_waycams' work: one launch less on the forward), and it is
// 1024 threads wide: the samples are scattered single loads, so the kernel is as long as one thread's chain of them.
#define TO_PROBE_THREADS 1024
template <bool PINHOLE>
__global__ void __launch_bounds__(TO_PROBE_THREADS)
k_traj_probe(CloudView cv, const float* __restrict__ poses, const float* __restrict__ quats, int C,
             const float* __restrict__ rig_q, const float* __restrict__ rig_t, WayHot* __restrict__ hot,
             WayCold* __restrict__ cold, WayAux* __restrict__ aux, CamConsts cc, int step,
             const uint32_t* __restrict__ occ, int64_t occw) {
    __shared__ float smx[TO_PROBE_THREADS / 64];
    __shared__ int szero[TO_PROBE_THREADS / 64];
    const int v = blockIdx.x, t = threadIdx.x;
    if (t == 0) prep_waycam(v, poses, quats, C, rig_q, rig_t, hot, cold, aux, nullptr, 0.f, 0, nullptr);
    __syncthreads();
    const WayHot h = hot[v];
    float mx = 0.f;
    int zero = 0;
    for (int64_t i = (int64_t)t * step; i < cv.n; i += (int64_t)TO_PROBE_THREADS * step) {
        float X, Y, Z, y0, y1, y2;
        to_cam(h, cv.soa[i], cv.soa[cv.npad + i], cv.soa[2 * cv.npad + i], X, Y, Z, y0, y1, y2);
        float om[1];
        load_occ<1>(occ, occw, v, i, om);
        const float p = soft_vis<PINHOLE>(cc, X, Y, Z, nullptr) * om[0];
        mx = fmaxf(mx, p);
        zero |= (p == 0.f);
    }
    for (int s = 32; s > 0; s >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, s)); zero |= __shfl_xor(zero, s); }
    if ((t & 63) == 0) { smx[t >> 6] = mx; szero[t >> 6] = zero; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < TO_PROBE_THREADS / 64; ++w) { mx = fmaxf(mx, smx[w]); zero |= szero[w]; }
        WayAux a = aux[v];
        a.L = mx;
        a.azero = zero ? 1.f : 0.f;
        cull_threshold(a.L, cc.inv_var, &a.thr1, &a.sthr1);
        aux[v] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// pass 1: per-waypoint min / max of p over the cloud.  grid = (point blocks, waypoint tiles).
// part[v * nslots + slot] = (min, max) over the 64*P points of one wave.

template <int P, bool PINHOLE>
__device__ __forceinline__ void pass1_dense_wp(const CamConsts& cc, const WayHot& h, const float (&x)[P], const float (&y)[P],
                                               const float (&z)[P], const float (&om)[P], float& mn, float& mx) {
    if constexpr (P >= 2) {  // two points per packed instruction
#pragma unroll
        for (int i = 0; i < P; i += 2) {
            f2 X, Y, Z, y0, y1, y2;
            to_cam_pk(h, f2{x[i], x[i + 1]}, f2{y[i], y[i + 1]}, f2{z[i], z[i + 1]}, X, Y, Z, y0, y1, y2);
            const f2 p = soft_vis_pk<PINHOLE>(cc, X, Y, Z, nullptr) * f2{om[i], om[i + 1]};
            mn = fminf(mn, fminf(p.x, p.y));
            mx = fmaxf(mx, fmaxf(p.x, p.y));
        }
    } else {
        float X, Y, Z, y0, y1, y2;
        to_cam(h, x[0], y[0], z[0], X, Y, Z, y0, y1, y2);
        const float p = soft_vis<PINHOLE>(cc, X, Y, Z, nullptr) * om[0];
        mn = fminf(mn, p);
        mx = fmaxf(mx, p);
    }
}

template <int P, bool PINHOLE, bool CULL, bool OCC>
__global__ void __launch_bounds__(TO_BLOCK)
k_traj_pass1(CloudView cv, const WayHot* __restrict__ hot, const WayAux* __restrict__ aux, int V, int vtile,
             CamConsts cc, float2* __restrict__ part, int nslots, const uint32_t* __restrict__ occ, int64_t occw) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * TO_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t base = ((int64_t)blockIdx.x * TO_BLOCK + threadIdx.x) * P;
    float x[P], y[P], z[P];
    load_points<P>(cv.soa, cv.npad, base, x, y, z);
    const int v0 = blockIdx.y * vtile;
    const int v1 = min(V, v0 + vtile);
    if (!CULL) {
        for (int v = v0; v < v1; ++v) {
            const WayHot h = hot[v];
            float mn = INFINITY, mx = 0.f;  // p >= +0 always
            float om[P];
            load_occ<P, OCC>(occ, occw, v, base, om);
            pass1_dense_wp<P, PINHOLE>(cc, h, x, y, z, om, mn, mx);
            mn = wave_min63_nn(mn);
            mx = wave_max63_nn(mx);
            if (lane == 63) part[(int64_t)v * nslots + slot] = make_float2(mn, mx);
        }
        return;
    }
    const float4 tb = wave_tile_bound(cv, base);
    for (int vc = v0; vc < v1; vc += 64) {
        unsigned long long live = tile_survivors<true>(hot, aux, vc, v1, tb, cc.mean);
        // waypoints whose max cannot be beaten from this tile: min is the proven 0, max unknown (-inf)
        if (vc + lane < v1 && !((live >> lane) & 1ull)) part[(int64_t)(vc + lane) * nslots + slot] = make_float2(0.f, -INFINITY);
        while (live) {
            const int v = vc + __builtin_ctzll(live);
            live &= live - 1ull;
            const WayHot h = hot[v];
            const WayAux a = aux[v];
            float mn = INFINITY, mx = 0.f;  // p >= +0 always (an all-culled wave reports max 0 <= the probe's L)
            float om[P];
            load_occ<P, OCC>(occ, occw, v, base, om);
            if (a.azero != 0.f) {
                mn = 0.f;  // proven by the probe; only the max is searched, among points that can reach L
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    float X, Y, Z, y0, y1, y2;
                    to_cam(h, x[i], y[i], z[i], X, Y, Z, y0, y1, y2);
                    if (__any(dist2_mean(X, Y, Z, cc.mean) <= a.thr1)) mx = fmaxf(mx, soft_vis<PINHOLE>(cc, X, Y, Z, nullptr) * om[i]);
                }
            } else {
                pass1_dense_wp<P, PINHOLE>(cc, h, x, y, z, om, mn, mx);
                mn = wave_min63_nn(mn);
            }
            mx = wave_max63_nn(mx);
            if (lane == 63) part[(int64_t)v * nslots + slot] = make_float2(mn, mx);
        }
    }
}

// block per virtual waypoint: a = min, M = max - a (== max(p - a): rounding is monotone), cull bound of
// the active set {p_hat > 0.5} = {p > a + M/2}
__global__ void __launch_bounds__(1024)
k_minmax_finish(const float2* __restrict__ part, int nslots, WayHot* __restrict__ hot, WayAux* __restrict__ aux,
                float inv_var, int cull, float* __restrict__ minmax) {
    __shared__ float smn[16], smx[16];
    const int v = blockIdx.x, t = threadIdx.x, nthreads = blockDim.x;
    float mn = INFINITY, mx = -INFINITY;
    for (int s = t; s < nslots; s += nthreads) {
        const float2 q = part[(int64_t)v * nslots + s];
        mn = fminf(mn, q.x);
        mx = fmaxf(mx, q.y);
    }
    for (int s = 32; s > 0; s >>= 1) { mn = fminf(mn, __shfl_xor(mn, s)); mx = fmaxf(mx, __shfl_xor(mx, s)); }
    if ((t & 63) == 0) { smn[t >> 6] = mn; smx[t >> 6] = mx; }
    __syncthreads();
    if (t == 0) {
        float a = smn[0], pmax = smx[0];
        for (int w = 1; w < (nthreads + 63) / 64; ++w) { a = fminf(a, smn[w]); pmax = fmaxf(pmax, smx[w]); }
        if (cull) pmax = fmaxf(pmax, aux[v].L);  // L is an attained value of p (defensive: it is never skipped)
        const float M = pmax - a;
        hot[v].a = a;
        hot[v].invM = 1.0f / M;
        float thr = INFINITY, sthr = INFINITY;
        // cull == 2 (the forward also records the backward's need mask): the argmin set {p == a} carries gradient when
        // a > 0 and lies outside the active set, so such a waypoint is not culled — the rule of apply_minmax below
        if (cull && M > 0.f && (cull == 1 || a == 0.f)) cull_threshold(fmaf(0.5f, M, a), inv_var, &thr, &sthr);
        hot[v].thr = thr;
        hot[v].sthr = sthr;
        aux[v].M = M;
        minmax[2 * v] = a;
        minmax[2 * v + 1] = M;
    }
}

// restore (a, M, 1/M, cull bound) from a caller-kept minmax array (backward entry point).  With a > 0 the
// argmin set {p == a} carries gradient and lies outside the active set, so culling is disabled.
__device__ __forceinline__ void apply_minmax(WayHot& h, float& auxM, float a, float M, float inv_var, int cull) {
    h.a = a; h.invM = 1.0f / M;
    float thr = INFINITY, sthr = INFINITY;
    if (cull && M > 0.f && a == 0.f) cull_threshold(0.5f * M, inv_var, &thr, &sthr);
    h.thr = thr; h.sthr = sthr;
    auxM = M;
}

// ---------------------------------------------------------------------------------------------
// pass 2: p_hat = (p - a)/M, clip to [0.5, 1-eps], log-odds, summed over the waypoints in order
// (model.py:226-231).  ATOMIC=false: one block column owns all waypoints and stores lo_sum once.

// `need` (optional): set when the pair will carry gradient in the backward — p_hat >= 1/2 before the clip, or a member of
// the argmin set when min p > 0 — the predicate of k_traj_bwd_scan, recorded by the forward for the backward to use.
template <bool PINHOLE>
__device__ __forceinline__ float log_odds(const CamConsts& cc, const WayHot& h, float X, float Y, float Z, float om,
                                          bool* need = nullptr) {
    const float p = soft_vis<PINHOLE>(cc, X, Y, Z, nullptr) * om;
    float ph = (p - h.a) * h.invM;
    if (need) *need |= (ph >= 0.5f) | ((h.a > 0.f) & (p == h.a));
    ph = __builtin_amdgcn_fmed3f(ph, 0.5f, cc.clip_hi);
    // log(ph/(1-ph)) as a difference of logs: exactly 0 at ph = 0.5
    return (to_log2(ph) - to_log2(1.0f - ph)) * 0.693147180559945f;
}

template <bool PINHOLE>
__device__ __forceinline__ f2 log_odds_pk(const CamConsts& cc, const WayHot& h, f2 X, f2 Y, f2 Z, f2 om, bool* need = nullptr) {
    const f2 p = soft_vis_pk<PINHOLE>(cc, X, Y, Z, nullptr) * om;
    f2 ph = (p - pk_splat(h.a)) * pk_splat(h.invM);
    if (need) *need |= (ph.x >= 0.5f) | (ph.y >= 0.5f) | ((h.a > 0.f) & ((p.x == h.a) | (p.y == h.a)));
    ph = f2{__builtin_amdgcn_fmed3f(ph.x, 0.5f, cc.clip_hi), __builtin_amdgcn_fmed3f(ph.y, 0.5f, cc.clip_hi)};
    const f2 q = pk_splat(1.0f) - ph;
    const f2 l = f2{to_log2(ph.x), to_log2(ph.y)} - f2{to_log2(q.x), to_log2(q.y)};
    return l * pk_splat(0.693147180559945f);
}

// NEED: also record, per (backward wave slot, waypoint), whether any pair will carry gradient (bit (v & 63) of
// need_out[(v >> 6) * nslots + slot], the layout of k_traj_bwd_scan) — the forward has p_hat of every pair it evaluates in
// hand, so the backward need not look for the active pairs again.  A backward slot is 64 * Pb points: `wps` = Pb / P waves
// of this kernel, consecutive in a block, whose bits are ORed through LDS.
template <int P, bool PINHOLE, bool CULL, bool OCC, bool NEED = false>
__global__ void __launch_bounds__(TO_BLOCK)
k_traj_pass2(CloudView cv, const WayHot* __restrict__ hot, int V, CamConsts cc, float* __restrict__ lo_sum,
             const uint32_t* __restrict__ occ, int64_t occw, unsigned long long* __restrict__ need_out = nullptr,
             int nslots = 0, int wps = 1) {
    __shared__ unsigned long long sbits[TO_WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t base = ((int64_t)blockIdx.x * TO_BLOCK + threadIdx.x) * P;
    // all waves of the block meet here once per 64 waypoints (V is uniform): the group's first wave writes the word
    auto put_need = [&](int vword, unsigned long long bits) {
        if constexpr (NEED) {
            if (wps == 1) {
                if (lane == 0) need_out[(int64_t)vword * nslots + (blockIdx.x * TO_WAVES_PER_BLOCK + wave)] = bits;
            } else {
                if (lane == 0) sbits[wave] = bits;
                __syncthreads();
                if (lane == 0 && wave % wps == 0) {
                    unsigned long long b = 0ull;
                    for (int j = 0; j < wps; ++j) b |= sbits[wave + j];
                    need_out[(int64_t)vword * nslots + (blockIdx.x * TO_WAVES_PER_BLOCK + wave) / wps] = b;
                }
                __syncthreads();
            }
        }
    };
    float x[P], y[P], z[P], acc[P];
    load_points<P>(cv.soa, cv.npad, base, x, y, z);
#pragma unroll
    for (int i = 0; i < P; ++i) acc[i] = 0.f;
    bool degenerate = false;  // M == 0: the reference divides 0/0 -> NaN for every point (model.py:227)
    if (CULL) {
        const float4 tb = wave_tile_bound(cv, base);
        for (int vc = 0; vc < V; vc += 64) {
            const int vl = vc + (int)(threadIdx.x & 63);
            degenerate |= __any(vl < V && !(hot[min(vl, V - 1)].invM < INFINITY));
            unsigned long long live = tile_survivors<false>(hot, nullptr, vc, V, tb, cc.mean);
            unsigned long long nbits = 0ull;
            // ascending waypoint order: the same summation order as the dense loop.  The next survivor's record is
            // requested before the current one is worked on: its index comes out of the bit mask, so without this the
            // scalar load's latency would be paid once per survivor — the tail of the heavy tiles.
            WayHot hn;
            int vn = 0;
            if (live) { vn = vc + __builtin_ctzll(live); hn = hot[vn]; }
            while (live) {
                const int v = vn;
                const WayHot h = hn;
                live &= live - 1ull;
                if (live) { vn = vc + __builtin_ctzll(live); hn = hot[vn]; }
                float om[P];
                load_occ<P, OCC>(occ, occw, v, base, om);
                bool need = false;
                bool* np = NEED ? &need : nullptr;
                // lanes beyond the bound get p_hat < 0.5 -> exactly 0, so evaluating them too changes nothing
                if constexpr (P >= 2) {
                    // two points per instruction, like the dense loop (the packed twins give the scalar results bit for bit)
#pragma unroll
                    for (int i = 0; i < P; i += 2) {
                        f2 X, Y, Z, y0, y1, y2;
                        to_cam_pk(h, f2{x[i], x[i + 1]}, f2{y[i], y[i + 1]}, f2{z[i], z[i + 1]}, X, Y, Z, y0, y1, y2);
                        const bool in0 = dist2_mean(X.x, Y.x, Z.x, cc.mean) <= h.thr, in1 = dist2_mean(X.y, Y.y, Z.y, cc.mean) <= h.thr;
                        if (__any(in0 | in1)) {
                            const f2 lo = log_odds_pk<PINHOLE>(cc, h, X, Y, Z, f2{om[i], om[i + 1]}, np);
                            acc[i] += lo.x;
                            acc[i + 1] += lo.y;
                        }
                    }
                } else {
                    float X, Y, Z, y0, y1, y2;
                    to_cam(h, x[0], y[0], z[0], X, Y, Z, y0, y1, y2);
                    if (__any(dist2_mean(X, Y, Z, cc.mean) <= h.thr)) acc[0] += log_odds<PINHOLE>(cc, h, X, Y, Z, om[0], np);
                }
                if (NEED && __any(need)) nbits |= 1ull << (v - vc);
            }
            put_need(vc >> 6, nbits);
        }
    } else {
        unsigned long long nbits = 0ull;
        for (int v = 0; v < V; ++v) {
            const WayHot h = hot[v];
            degenerate |= !(h.invM < INFINITY);
            float om[P];
            load_occ<P, OCC>(occ, occw, v, base, om);
            bool need = false;
            bool* np = NEED ? &need : nullptr;
            if constexpr (P >= 2) {
#pragma unroll
                for (int i = 0; i < P; i += 2) {
                    f2 X, Y, Z, y0, y1, y2;
                    to_cam_pk(h, f2{x[i], x[i + 1]}, f2{y[i], y[i + 1]}, f2{z[i], z[i + 1]}, X, Y, Z, y0, y1, y2);
                    const f2 lo = log_odds_pk<PINHOLE>(cc, h, X, Y, Z, f2{om[i], om[i + 1]}, np);
                    acc[i] += lo.x;
                    acc[i + 1] += lo.y;
                }
            } else {
                float X, Y, Z, y0, y1, y2;
                to_cam(h, x[0], y[0], z[0], X, Y, Z, y0, y1, y2);
                acc[0] += log_odds<PINHOLE>(cc, h, X, Y, Z, om[0], np);
            }
            if constexpr (NEED) {
                if (__any(need)) nbits |= 1ull << (v & 63);
                if ((v & 63) == 63 || v == V - 1) {
                    put_need(v >> 6, nbits);
                    nbits = 0ull;
                }
            }
        }
    }
    if (degenerate) {
#pragma unroll
        for (int i = 0; i < P; ++i) acc[i] = __builtin_nanf("");
    }
    // one block column owns all waypoints of its points: a single store, fixed summation order
    if constexpr (P == 4) *reinterpret_cast<float4*>(lo_sum + base) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    else if constexpr (P == 2) *reinterpret_cast<float2*>(lo_sum + base) = make_float2(acc[0], acc[1]);
    else lo_sum[base] = acc[0];
}

// ---------------------------------------------------------------------------------------------
// rewards = sigmoid(lo_sum) (model.py:237) scattered back to the caller's point order, mean and
// visibility loss (model.py:246)

__global__ void __launch_bounds__(TO_BLOCK)
k_reward(const float* __restrict__ lo_sum, const int* __restrict__ inv, int64_t n, float* __restrict__ rewards,
         double* __restrict__ part) {
    __shared__ double lds[TO_BLOCK];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    // thread per OUTPUT element (the caller's order): a scattered 4-byte read of lo_sum and a coalesced store, rather
    // than a coalesced read and a scattered store
    for (int64_t o = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; o < n; o += stride) {
        const float lo = lo_sum[inv[o]];
        float r = to_rcp(1.0f + to_exp(-lo));
        if (lo != lo) r = lo;  // a degenerate waypoint (max == min) makes the reference's rewards NaN: propagate
        rewards[o] = r;
        s += (double)r;
    }
    const double tot = block_sum_double(s, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(TO_BLOCK)
k_reward_finish(const double* __restrict__ part, int nparts, int64_t n, float eps, float* __restrict__ scalars) {
    __shared__ double lds[TO_BLOCK];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += TO_BLOCK) s += part[i];
    const double tot = block_sum_double(s, lds);
    if (threadIdx.x == 0) {
        const float mean = (float)(tot / (double)n);
        const float vis = 1.0f / (mean + eps);
        scalars[0] = mean;
        scalars[1] = vis;
        scalars[2] = (float)(-(double)vis * (double)vis / (double)n);
        scalars[3] = 0.f;  // reserved; written so that callers need not clear the vector
    }
}

// ---------------------------------------------------------------------------------------------
// backward.  Per (point, waypoint): G = dL/dp_hat = g_n [0.5 <= p_hat <= 1-eps] / (p_hat (1 - p_hat)),
// dL/dp = G / M, plus the shares of the min/max points (torch splits them evenly among ties):
//   S1 = sum G (p_hat - 1)/M  -> argmin set,   S2 = sum G (-p_hat)/M -> argmax set.
// Per wave and waypoint 14 sums leave through part[(v*nslots+slot)*16 ..]:
//   [0..2] sum w g   [3..11] sum w y (x) g   [12] S1   [13] S2        (w = G/M, g = dp/dc, y = x - t)
// The min/max sets (normally one point each) add their unweighted (g, y (x) g) into ties[v*32 ..] with
// float atomics: [0..11] argmin set, [12..23] argmax set, [24] n_min, [25] n_max.

#define TO_BWD_NSUM 14

template <bool PINHOLE>
__device__ __forceinline__ bool bwd_accum(const CamConsts& cc, const WayHot& h, float M, const Vis& s, float X, float Y,
                                          float Z, float y0, float y1, float y2, float gn, bool valid,
                                          float (&acc)[TO_BWD_NSUM], float* __restrict__ tb);

template <bool PINHOLE>
__device__ __forceinline__ bool bwd_eval(const CamConsts& cc, const WayHot& h, float M, float X, float Y, float Z,
                                         float y0, float y1, float y2, float gn, bool valid, float om,
                                         float (&acc)[TO_BWD_NSUM], float* __restrict__ tb) {
    Vis s;
    soft_vis<PINHOLE>(cc, X, Y, Z, &s);
    s.p *= om;  // an occluded pair has p = 0: inactive, and dp/dc = p * (...) = 0
    return bwd_accum<PINHOLE>(cc, h, M, s, X, Y, Z, y0, y1, y2, gn, valid, acc, tb);
}

template <bool PINHOLE>
__device__ __forceinline__ bool bwd_accum(const CamConsts& cc, const WayHot& h, float M, const Vis& s, float X, float Y,
                                          float Z, float y0, float y1, float y2, float gn, bool valid,
                                          float (&acc)[TO_BWD_NSUM], float* __restrict__ tb) {
    const float p = s.p;
    const float pp = p - h.a;
    const float ph = pp * h.invM;
    const bool act = (ph >= 0.5f) && (ph <= cc.clip_hi);
    const bool is_min = valid && (p == h.a) && (p > 0.f);
    const bool is_max = valid && (pp == M) && (M > 0.f);
    if (act || is_min || is_max) {
        float g[3];
        dvis_dc<PINHOLE>(cc, X, Y, Z, s, g);
        if (act) {
            const float G = gn * to_rcp(ph * (1.0f - ph));
            const float wgt = G * h.invM;
            acc[12] = fmaf(wgt, ph - 1.0f, acc[12]);
            acc[13] = fmaf(-wgt, ph, acc[13]);
            const float w0 = wgt * g[0], w1 = wgt * g[1], w2 = wgt * g[2];
            acc[0] += w0; acc[1] += w1; acc[2] += w2;
            acc[3] = fmaf(y0, w0, acc[3]); acc[4] = fmaf(y0, w1, acc[4]); acc[5] = fmaf(y0, w2, acc[5]);
            acc[6] = fmaf(y1, w0, acc[6]); acc[7] = fmaf(y1, w1, acc[7]); acc[8] = fmaf(y1, w2, acc[8]);
            acc[9] = fmaf(y2, w0, acc[9]); acc[10] = fmaf(y2, w1, acc[10]); acc[11] = fmaf(y2, w2, acc[11]);
        }
        if (is_min || is_max) {
            const float yy[3] = {y0, y1, y2};
            if (is_min) {
                for (int k = 0; k < 3; ++k) atomicAdd(tb + k, g[k]);
                for (int j = 0; j < 3; ++j)
                    for (int k = 0; k < 3; ++k) atomicAdd(tb + 3 + 3 * j + k, yy[j] * g[k]);
                atomicAdd(tb + 24, 1.0f);
            }
            if (is_max) {
                for (int k = 0; k < 3; ++k) atomicAdd(tb + 12 + k, g[k]);
                for (int j = 0; j < 3; ++j)
                    for (int k = 0; k < 3; ++k) atomicAdd(tb + 15 + 3 * j + k, yy[j] * g[k]);
                atomicAdd(tb + 25, 1.0f);
            }
        }
    }
    return act;
}

// First half of the dense backward, split off so that it can run while the log-odds vector is still being all-reduced
// (multi-GPU): for every (wave of points, virtual waypoint) whether any pair needs the gradient path.  That depends on
// p and the waypoint's min/max only — not on lo_sum.  Every pair is evaluated here (the packed, branch-free phase 1 of
// k_traj_bwd); bit (v & 63) of need[(v >> 6) * nslots + slot].  Waypoint tiles (grid.y) start at multiples of 64, so a
// word has one writer.
template <int P, bool PINHOLE, bool OCC>
__global__ void __launch_bounds__(TO_BLOCK)
k_traj_bwd_scan(CloudView cv, const WayHot* __restrict__ hot, int V, int vtile, CamConsts cc,
                unsigned long long* __restrict__ need_out, int nslots, const uint32_t* __restrict__ occ, int64_t occw) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * TO_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t base = ((int64_t)blockIdx.x * TO_BLOCK + threadIdx.x) * P;
    float x[P], y[P], z[P];
    load_points<P>(cv.soa, cv.npad, base, x, y, z);
    const int v0 = blockIdx.y * vtile;
    const int v1 = min(V, v0 + vtile);
    unsigned long long bits = 0ull;
    for (int v = v0; v < v1; ++v) {
        const WayHot h = hot[v];
        float om[P];
        load_occ<P, OCC>(occ, occw, v, base, om);
        bool need = false;
        if constexpr (P >= 2) {
#pragma unroll
            for (int i = 0; i < P; i += 2) {
                f2 X, Y, Z, y0, y1, y2;
                to_cam_pk(h, f2{x[i], x[i + 1]}, f2{y[i], y[i + 1]}, f2{z[i], z[i + 1]}, X, Y, Z, y0, y1, y2);
                const f2 p = soft_vis_pk<PINHOLE>(cc, X, Y, Z, nullptr) * f2{om[i], om[i + 1]};
                const f2 ph = (p - pk_splat(h.a)) * pk_splat(h.invM);
                need |= (ph.x >= 0.5f) | (ph.y >= 0.5f) | ((h.a > 0.f) & ((p.x == h.a) | (p.y == h.a)));
            }
        } else {
            float X, Y, Z, y0, y1, y2;
            to_cam(h, x[0], y[0], z[0], X, Y, Z, y0, y1, y2);
            const float p = soft_vis<PINHOLE>(cc, X, Y, Z, nullptr) * om[0];
            const float ph = (p - h.a) * h.invM;
            need = (ph >= 0.5f) | ((h.a > 0.f) & (p == h.a));
        }
        if (__any(need)) bits |= 1ull << (v & 63);
        if ((v & 63) == 63 || v == v1 - 1) {
            if (lane == 0) need_out[(int64_t)(v >> 6) * nslots + slot] = bits;
            bits = 0ull;
        }
    }
}

template <int P, bool PINHOLE, bool CULL, bool OCC, bool MASKED = false>
__global__ void __launch_bounds__(TO_BLOCK)
k_traj_bwd(CloudView cv, const WayHot* __restrict__ hot, const WayAux* __restrict__ aux, int V, int vtile,
           CamConsts cc, const float* __restrict__ lo_sum, const float* __restrict__ grad_rewards,
           const float* __restrict__ scalars, const float* __restrict__ gout, float* __restrict__ part, int nslots,
           float* __restrict__ ties, const uint32_t* __restrict__ occ, int64_t occw,
           unsigned long long* __restrict__ tmask, const unsigned long long* __restrict__ need_in = nullptr) {
    const int lane = threadIdx.x & 63;
    const int slot = blockIdx.x * TO_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t base = ((int64_t)blockIdx.x * TO_BLOCK + threadIdx.x) * P;
    float x[P], y[P], z[P], gn[P];
    bool valid[P];
    load_points<P>(cv.soa, cv.npad, base, x, y, z);
    // dL/d reward_n: a caller-supplied vector (general criterion), else the fused visibility loss
    const float coef = grad_rewards ? 0.f : scalars[2] * gout[0];
    float lo[P];
    load_vec<P>(lo_sum, base, lo);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        valid[i] = base + i < cv.n;  // pads carry no gradient
        float r = to_rcp(1.0f + to_exp(-lo[i]));  // == k_reward's value of rewards[perm[base+i]]
        if (lo[i] != lo[i]) r = lo[i];
        float gr = coef;
        if (grad_rewards) {
            const int pi = cv.perm[base + i];  // the caller's index (-1 for pads)
            gr = pi >= 0 ? grad_rewards[pi] : 0.f;
        }
        gn[i] = valid[i] ? gr * r * (1.0f - r) : 0.f;  // dL/d lo_sum_n
    }
    const int v0 = blockIdx.y * vtile;
    const int v1 = min(V, v0 + vtile);
    auto store = [&](int v, const float (&acc)[TO_BWD_NSUM]) {
        float4* dst = reinterpret_cast<float4*>(part + ((int64_t)v * nslots + slot) * 16);
        dst[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        dst[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        dst[2] = make_float4(acc[8], acc[9], acc[10], acc[11]);
        dst[3] = make_float4(acc[12], acc[13], 0.f, 0.f);
    };
    // Dense mode writes a partial only where the wave had an active pair and records that in tmask (bit (v & 63) of
    // tmask[(v >> 6) * nslots + slot]; tiles start at multiples of 64: one writer per word): k_bwd_finish1 sums the
    // recorded partials.  Most (wave, waypoint) combinations are untouched, and 64 bytes of zeros each were a quarter of
    // this kernel's time and most of the finish kernel's.
    if constexpr (!CULL && MASKED) {
        // second half of the split backward: k_traj_bwd_scan has evaluated every pair; only the flagged
        // (wave, waypoint) combinations take the gradient path (the scalar twin: bit-identical p)
        float acc[TO_BWD_NSUM];
#pragma unroll
        for (int k = 0; k < TO_BWD_NSUM; ++k) acc[k] = 0.f;
        // a block row owns 16 waypoints — a quarter of a mask word (vtile = 16 here): the flagged waypoints of the few
        // waves near the path are walked one after the other, and sixteen at most keeps that tail short
        {
            const int vc = v0;
            const int64_t word = (int64_t)(vc >> 6) * nslots + slot;
            unsigned live = (unsigned)(need_in[word] >> (vc & 63)) & 0xffffu;
            unsigned done = 0u;
            while (live) {
                const int b = __builtin_ctz(live);
                live &= live - 1u;
                const int v = vc + b;
                const WayHot h = hot[v];
                const float M = aux[v].M;
                float om[P];
                load_occ<P, OCC>(occ, occw, v, base, om);
                bool any_act = false;
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    float X, Y, Z, y0, y1, y2;
                    to_cam(h, x[i], y[i], z[i], X, Y, Z, y0, y1, y2);
                    any_act |= bwd_eval<PINHOLE>(cc, h, M, X, Y, Z, y0, y1, y2, gn[i], valid[i], om[i], acc,
                                                 ties + (int64_t)v * 32);
                }
                if (__any(any_act)) {
#pragma unroll
                    for (int k = 0; k < TO_BWD_NSUM; ++k) acc[k] = wave_sum63(acc[k]);
                    if (lane == 63) store(v, acc);
#pragma unroll
                    for (int k = 0; k < TO_BWD_NSUM; ++k) acc[k] = 0.f;
                    done |= 1u << b;
                }
            }
            // the quarter of the (little-endian) 64-bit word this row owns
            if (lane == 0) reinterpret_cast<unsigned short*>(tmask)[word * 4 + ((vc & 63) >> 4)] = (unsigned short)done;
        }
        return;
    }
    if (!CULL) {
        float acc[TO_BWD_NSUM];
#pragma unroll
        for (int k = 0; k < TO_BWD_NSUM; ++k) acc[k] = 0.f;
        unsigned long long bits = 0ull;
        for (int v = v0; v < v1; ++v) {
            const WayHot h = hot[v];
            bool any_act = false;
            float om[P];
            load_occ<P, OCC>(occ, occw, v, base, om);
            if constexpr (P >= 2) {
                // phase 1 (packed, branch-free): p and p_hat of every pair, and whether any element of the wave needs
                // the gradient path; phase 2, under ONE wave-uniform branch per waypoint, re-evaluates those few
                // elements with the scalar twin (bit-identical p), so nothing but `need` stays live across the branch
                bool need = false;
#pragma unroll
                for (int i = 0; i < P; i += 2) {
                    f2 X, Y, Z, y0, y1, y2;
                    to_cam_pk(h, f2{x[i], x[i + 1]}, f2{y[i], y[i + 1]}, f2{z[i], z[i + 1]}, X, Y, Z, y0, y1, y2);
                    const f2 p = soft_vis_pk<PINHOLE>(cc, X, Y, Z, nullptr) * f2{om[i], om[i + 1]};
                    const f2 ph = (p - pk_splat(h.a)) * pk_splat(h.invM);
                    // superset of (act | is_min | is_max); the argmin set only matters when a > 0 (bwd_accum requires
                    // p == a && p > 0; with a = 0 — the usual case — half the cloud has p == 0)
                    need |= (ph.x >= 0.5f) | (ph.y >= 0.5f) | ((h.a > 0.f) & ((p.x == h.a) | (p.y == h.a)));
                }
                if (__any(need)) {
                    const float M = aux[v].M;  // only the gradient path compares against the max
#pragma unroll
                    for (int i = 0; i < P; ++i) {
                        float X, Y, Z, y0, y1, y2;
                        to_cam(h, x[i], y[i], z[i], X, Y, Z, y0, y1, y2);
                        any_act |= bwd_eval<PINHOLE>(cc, h, M, X, Y, Z, y0, y1, y2, gn[i], valid[i], om[i], acc,
                                                     ties + (int64_t)v * 32);
                    }
                }
            } else {
                float X, Y, Z, y0, y1, y2;
                to_cam(h, x[0], y[0], z[0], X, Y, Z, y0, y1, y2);
                any_act |= bwd_eval<PINHOLE>(cc, h, aux[v].M, X, Y, Z, y0, y1, y2, gn[0], valid[0], om[0], acc,
                                             ties + (int64_t)v * 32);
            }
            // every pair has been evaluated; when no lane of the wave was active all 14 sums are exact zeros: nothing to
            // reduce and nothing to store
            if (__any(any_act)) {
#pragma unroll
                for (int k = 0; k < TO_BWD_NSUM; ++k) acc[k] = wave_sum63(acc[k]);
                if (lane == 63) store(v, acc);
#pragma unroll
                for (int k = 0; k < TO_BWD_NSUM; ++k) acc[k] = 0.f;
                bits |= 1ull << (v & 63);
            }
            if ((v & 63) == 63 || v == v1 - 1) {
                if (lane == 0) tmask[(int64_t)(v >> 6) * nslots + slot] = bits;
                bits = 0ull;
            }
        }
        return;
    }
    const float4 tb = wave_tile_bound(cv, base);
    for (int vc = v0; vc < v1; vc += 64) {
        unsigned long long live = tile_survivors<false>(hot, nullptr, vc, v1, tb, cc.mean);
        // (tile, waypoint) pairs that are not live write nothing: k_bwd_finish1 repeats the test and skips them
        while (live) {
            const int v = vc + __builtin_ctzll(live);
            live &= live - 1ull;
            const WayHot h = hot[v];
            const float M = aux[v].M;
            float acc[TO_BWD_NSUM];
#pragma unroll
            for (int k = 0; k < TO_BWD_NSUM; ++k) acc[k] = 0.f;
            bool any_act = false;
            float om[P];
            load_occ<P, OCC>(occ, occw, v, base, om);
#pragma unroll
            for (int i = 0; i < P; ++i) {
                float X, Y, Z, y0, y1, y2;
                to_cam(h, x[i], y[i], z[i], X, Y, Z, y0, y1, y2);
                if (__any(dist2_mean(X, Y, Z, cc.mean) <= h.thr))
                    any_act |= bwd_eval<PINHOLE>(cc, h, M, X, Y, Z, y0, y1, y2, gn[i], valid[i], om[i], acc, ties + (int64_t)v * 32);
            }
            if (__any(any_act)) {  // otherwise every lane's sums are exact zeros: nothing to reduce
#pragma unroll
                for (int k = 0; k < TO_BWD_NSUM; ++k) acc[k] = wave_sum63(acc[k]);
            }
            if (lane == 63) store(v, acc);
        }
    }
}

// thread per body waypoint: rig composition, dL/dt = -R sum dL/dc, dL/dR = sum y (x) dL/dc,
// quaternion chain through the homogeneous form of R and through F.normalize.
__device__ void finish_waypoint(int w, const float* __restrict__ vgrad, const WayHot* __restrict__ hot,
                                const WayCold* __restrict__ cold, int C, const float* __restrict__ rig_q,
                                const float* __restrict__ rig_t, float* __restrict__ poses_grad,
                                float* __restrict__ quats_grad) {
    double dt[3] = {0, 0, 0}, A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // A[3*j+i] = dL/dR_w[j][i]
    for (int c = 0; c < C; ++c) {
        const int v = w * C + c;
        const float* gv = vgrad + (int64_t)v * 12;
        const WayHot h = hot[v];
        // dL/dt_v = -R_v Gt,   R_v[j][i] = h.m[3*i+j]
        double dtv[3];
        for (int j = 0; j < 3; ++j)
            dtv[j] = -((double)h.m[j] * gv[0] + (double)h.m[3 + j] * gv[1] + (double)h.m[6 + j] * gv[2]);
        for (int j = 0; j < 3; ++j) dt[j] += dtv[j];
        if (rig_q != nullptr) {
            const float qc[4] = {rig_q[4 * c], rig_q[4 * c + 1], rig_q[4 * c + 2], rig_q[4 * c + 3]};
            float Rc[9];
            quat_to_R(qc, Rc);
            // dL/dR_w = dL/dR_v Rc^T + dL/dt_v l^T
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i) {
                    double a = 0;
                    for (int k = 0; k < 3; ++k) a += (double)gv[3 + 3 * j + k] * (double)Rc[3 * i + k];
                    if (rig_t != nullptr) a += dtv[j] * (double)rig_t[3 * c + i];
                    A[3 * j + i] += a;
                }
        } else {
            for (int k = 0; k < 9; ++k) A[k] += (double)gv[3 + k];
        }
    }
    for (int j = 0; j < 3; ++j) poses_grad[3 * w + j] = (float)dt[j];
    const WayCold cd = cold[w];
    const double qw = cd.qn[0], qx = cd.qn[1], qy = cd.qn[2], qz = cd.qn[3];
#define AA(j, i) A[3 * (j) + (i)]
    double dh[4];
    dh[0] = 2 * (qw * (AA(0, 0) + AA(1, 1) + AA(2, 2)) + qz * (AA(1, 0) - AA(0, 1)) + qy * (AA(0, 2) - AA(2, 0)) + qx * (AA(2, 1) - AA(1, 2)));
    dh[1] = 2 * (qx * (AA(0, 0) - AA(1, 1) - AA(2, 2)) + qy * (AA(0, 1) + AA(1, 0)) + qz * (AA(0, 2) + AA(2, 0)) + qw * (AA(2, 1) - AA(1, 2)));
    dh[2] = 2 * (qy * (-AA(0, 0) + AA(1, 1) - AA(2, 2)) + qx * (AA(0, 1) + AA(1, 0)) + qw * (AA(0, 2) - AA(2, 0)) + qz * (AA(1, 2) + AA(2, 1)));
    dh[3] = 2 * (qz * (-AA(0, 0) - AA(1, 1) + AA(2, 2)) + qw * (AA(1, 0) - AA(0, 1)) + qx * (AA(0, 2) + AA(2, 0)) + qy * (AA(1, 2) + AA(2, 1)));
#undef AA
    const double dot = qw * dh[0] + qx * dh[1] + qy * dh[2] + qz * dh[3];
    const double inv = 1.0 / (double)cd.nrm;
    quats_grad[4 * w + 0] = (float)((dh[0] - qw * dot) * inv);
    quats_grad[4 * w + 1] = (float)((dh[1] - qx * dot) * inv);
    quats_grad[4 * w + 2] = (float)((dh[2] - qy * dot) * inv);
    quats_grad[4 * w + 3] = (float)((dh[3] - qz * dot) * inv);
}

__global__ void k_bwd_finish2(const float* __restrict__ vgrad, const WayHot* __restrict__ hot,
                              const WayCold* __restrict__ cold, int W, int C, const float* __restrict__ rig_q,
                              const float* __restrict__ rig_t, float* __restrict__ poses_grad,
                              float* __restrict__ quats_grad) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < W) finish_waypoint(w, vgrad, hot, cold, C, rig_q, rig_t, poses_grad, quats_grad);
}

// block per virtual waypoint: sum the wave partials (double, fixed order), add the min/max shares,
// write vgrad[v*12 ..] = (sum dL/dc [3], sum y (x) dL/dc [9]).  In CULL mode a partial exists only where the
// (tile, waypoint) pair is live — the same predicate, on the same records, as k_traj_bwd evaluated; in dense mode where
// k_traj_bwd recorded one in tmask.
#define TO_FINISH_THREADS 256  // (1024 threads measured slower: 19 vs 16 us — the 14 block sums then cross 16 waves)
__global__ void __launch_bounds__(TO_FINISH_THREADS)
k_bwd_finish1(const float* __restrict__ part, int nslots, int slots_per_tile_shift, const float* __restrict__ ties,
              const WayHot* __restrict__ hot, const float4* __restrict__ bounds, float mean, int cull,
              const unsigned long long* __restrict__ tmask, float* __restrict__ vgrad,
              const WayCold* __restrict__ cold, int C, float* __restrict__ poses_grad, float* __restrict__ quats_grad) {
    __shared__ double lds[TO_BLOCK];
    __shared__ double tot[TO_BWD_NSUM];
    const int v = blockIdx.x, t = threadIdx.x;
    const float4* hp = reinterpret_cast<const float4*>(hot + v);
    const float4 q0 = hp[0], q1 = hp[1], q2 = hp[2], q3 = hp[3];
    double s[TO_BWD_NSUM];
    for (int k = 0; k < TO_BWD_NSUM; ++k) s[k] = 0.0;
    // four slots per trip: their liveness loads are issued together (the loop is a chain of dependent loads otherwise);
    // the partials are still added in increasing slot order
    for (int sl0 = t; sl0 < nslots; sl0 += 4 * TO_FINISH_THREADS) {
        bool live[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sl = sl0 + j * TO_FINISH_THREADS;
            live[j] = false;
            if (sl < nslots) {
                if (cull) {
                    // slot -> 256-point tile: P=4: slot == tile, P=2: two slots per tile, P=1: four
                    const float4 tb = bounds[sl >> slots_per_tile_shift];
                    live[j] = tile_live(q0, q1, q2, q3.z, q3.w, tb, mean);
                } else {
                    live[j] = (tmask[(int64_t)(v >> 6) * nslots + sl] >> (v & 63)) & 1ull;  // dense: the recorded partials
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!live[j]) continue;
            const int sl = sl0 + j * TO_FINISH_THREADS;
            const float4* src = reinterpret_cast<const float4*>(part + ((int64_t)v * nslots + sl) * 16);
            const float4 a = src[0], b = src[1], c = src[2], d = src[3];
            s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
            s[4] += b.x; s[5] += b.y; s[6] += b.z; s[7] += b.w;
            s[8] += c.x; s[9] += c.y; s[10] += c.z; s[11] += c.w;
            s[12] += d.x; s[13] += d.y;
        }
    }
    for (int k = 0; k < TO_BWD_NSUM; ++k) {
        const double r = block_sum_double(s[k], lds);
        if (t == 0) tot[k] = r;
        __syncthreads();
    }
    if (t < 12) {
        const float* tb = ties + (int64_t)v * 32;
        const double nmin = tb[24], nmax = tb[25];
        const double wmin = nmin > 0.0 ? tot[12] / nmin : 0.0;
        const double wmax = nmax > 0.0 ? tot[13] / nmax : 0.0;
        vgrad[v * 12 + t] = (float)(tot[t] + wmin * (double)tb[t] + wmax * (double)tb[12 + t]);
    }
    if (C == 1) {  // one camera per waypoint: the waypoint's gradient follows at once (k_bwd_finish2's work, no launch)
        __syncthreads();
        if (t == 0) finish_waypoint(v, vgrad, hot, cold, 1, nullptr, nullptr, poses_grad, quats_grad);
    }
}

// ---------------------------------------------------------------------------------------------
// occlusion bits: row = all ones, then every point kept by the hard frustum cull is cleared and
// every point HPR (or the z-buffer) found visible among the kept ones is set again.

__global__ void k_inverse_perm(const int* __restrict__ perm, int64_t n, int* __restrict__ inv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) inv[perm[i]] = (int)i;
}

__global__ void k_occ_clear_kept(const int* __restrict__ inv, const int32_t* __restrict__ kept_idx,
                                 const int32_t* __restrict__ kept_count, uint32_t* __restrict__ row) {
    const int m = *kept_count;
    const int stride = gridDim.x * blockDim.x;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const int s = inv[kept_idx[j]];
        atomicAnd(&row[s >> 5], ~(1u << (s & 31)));
    }
}

__global__ void k_occ_set_visible(const int* __restrict__ inv, const int32_t* __restrict__ kept_idx,
                                  const int32_t* __restrict__ vis_idx, const int32_t* __restrict__ vis_count,
                                  uint32_t* __restrict__ row) {
    const int m = *vis_count;
    const int stride = gridDim.x * blockDim.x;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const int s = inv[kept_idx[vis_idx[j]]];
        atomicOr(&row[s >> 5], 1u << (s & 31));
    }
}

// all waypoints' rows in three launches: grid.y = waypoint.  kept_idx (W, n): waypoint w's kept points in its first
// kept_count[w] entries; vis_idx: the visible ones among them as positions in that list, waypoint w's in
// [vis_off[w], vis_off[w+1]); all_visible[w] != 0: nothing of w is occluded (its row stays all ones).
__global__ void k_occ_rows_clear(const int* __restrict__ inv, const int32_t* __restrict__ kept_idx, int64_t n,
                                 const int32_t* __restrict__ kept_count, const int32_t* __restrict__ all_visible,
                                 uint32_t* __restrict__ rows, int64_t roww) {
    const int w = blockIdx.y;
    if (all_visible[w]) return;
    const int m = kept_count[w];
    const int32_t* k = kept_idx + (int64_t)w * n;
    uint32_t* row = rows + (int64_t)w * roww;
    const int stride = gridDim.x * blockDim.x;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const int s = inv[k[j]];
        atomicAnd(&row[s >> 5], ~(1u << (s & 31)));
    }
}

__global__ void k_occ_rows_set(const int* __restrict__ inv, const int32_t* __restrict__ kept_idx, int64_t n,
                               const int32_t* __restrict__ vis_idx, const int32_t* __restrict__ vis_off,
                               const int32_t* __restrict__ all_visible, uint32_t* __restrict__ rows, int64_t roww) {
    const int w = blockIdx.y;
    if (all_visible[w]) return;
    const int j0 = vis_off[w], j1 = vis_off[w + 1];
    const int32_t* k = kept_idx + (int64_t)w * n;
    uint32_t* row = rows + (int64_t)w * roww;
    const int stride = gridDim.x * blockDim.x;
    for (int j = j0 + blockIdx.x * blockDim.x + threadIdx.x; j < j1; j += stride) {
        const int s = inv[k[vis_idx[j]]];
        atomicOr(&row[s >> 5], 1u << (s & 31));
    }
}

extern "C" int tohip_occlusion_rows(int64_t n, const int32_t* inv_perm, const int32_t* kept_idx, const int32_t* kept_count,
                                    const int32_t* vis_idx, const int32_t* vis_off, const int32_t* all_visible, int64_t n_wps,
                                    uint32_t* rows, void* stream_) {
    if (!inv_perm || !kept_idx || !kept_count || !vis_idx || !vis_off || !all_visible || !rows || n <= 0 || n_wps <= 0 ||
        n_wps > 65535)
        return TOHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    const int64_t roww = tohip_padded_points(n) / 32;
    hipError_t e = hipMemsetAsync(rows, 0xff, (size_t)roww * (size_t)n_wps * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    int nb = (int)((n + 255) / 256);
    if (nb > 256) nb = 256;
    const dim3 grid((unsigned)nb, (unsigned)n_wps);
    k_occ_rows_clear<<<grid, 256, 0, st>>>(inv_perm, kept_idx, n, kept_count, all_visible, rows, roww);
    TO_HIP_CHECK_LAUNCH();
    k_occ_rows_set<<<grid, 256, 0, st>>>(inv_perm, kept_idx, n, vis_idx, vis_off, all_visible, rows, roww);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_inverse_permutation(const void* packed, int64_t n, int32_t* inv, void* stream_) {
    if (!packed || !inv || n <= 0) return TOHIP_EINVAL;
    const CloudView cv = cloud_view(packed, n);
    k_inverse_perm<<<(int)((n + 255) / 256), 256, 0, (hipStream_t)stream_>>>(cv.perm, n, inv);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_occlusion_row(int64_t n, const int32_t* inv_perm, const int32_t* kept_idx, const int32_t* kept_count,
                                   const int32_t* vis_idx, const int32_t* vis_count, uint32_t* row, void* stream_) {
    if (!inv_perm || !kept_idx || !kept_count || !vis_idx || !vis_count || !row || n <= 0) return TOHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    const int64_t npad = tohip_padded_points(n);
    hipError_t e = hipMemsetAsync(row, 0xff, (size_t)(npad / 32) * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    int nb = (int)((n + 255) / 256);
    if (nb > 1024) nb = 1024;
    k_occ_clear_kept<<<nb, 256, 0, st>>>(inv_perm, kept_idx, kept_count, row);
    TO_HIP_CHECK_LAUNCH();
    k_occ_set_visible<<<nb, 256, 0, st>>>(inv_perm, kept_idx, vis_idx, vis_count, row);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// host side: workspace layout + launches

namespace {

struct TrajPlan {
    int64_t npad;
    int P;         // points per lane
    int nblk;      // point blocks
    int nslots;    // wave slots = nblk * 4
    int V;
    size_t off_hot, off_cold, off_aux, off_mm, off_rpart, off_bpart, off_ties, off_vgrad, off_tmask, total;
};

inline int choose_P(int64_t n) {
    static const int forced = [] { const char* e = getenv("TOHIP_FORCE_P"); return e ? atoi(e) : 0; }();  // experiments
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    return n >= (int64_t)512 * 1024 ? 4 : (n >= (int64_t)128 * 1024 ? 2 : 1);
}

inline TrajPlan make_plan(int64_t n, int64_t V, int64_t W) {
    TrajPlan p;
    p.npad = tohip_padded_points(n);
    p.P = choose_P(n);
    p.nblk = (int)(p.npad / (TO_BLOCK * p.P));
    p.nslots = p.nblk * TO_WAVES_PER_BLOCK;
    p.V = (int)V;
    // P is a pure function of n, hence so are the slot count and the whole plan
    const size_t max_slots = (size_t)p.nslots;
    size_t o = 0;
    p.off_rpart = o; o += align_up((size_t)4096 * sizeof(double), 256);  // first: tohip_traj_reward uses only this
    p.off_hot = o;   o += align_up((size_t)V * sizeof(WayHot), 256);
    p.off_cold = o;  o += align_up((size_t)W * sizeof(WayCold), 256);
    p.off_aux = o;   o += align_up((size_t)V * sizeof(WayAux), 256);
    p.off_mm = o;    o += align_up((size_t)V * max_slots * sizeof(float2), 256);
    p.off_bpart = o; o += align_up((size_t)V * max_slots * 16 * sizeof(float), 256);
    p.off_ties = o;  o += align_up((size_t)V * 32 * sizeof(float), 256);
    p.off_vgrad = o; o += align_up((size_t)V * 12 * sizeof(float), 256);
    p.off_tmask = o; o += align_up((size_t)((V + 63) / 64) * max_slots * sizeof(unsigned long long), 256);
    p.total = o;
    return p;
}

// waypoint tiling of the grid's y dimension: enough blocks to fill 256 CUs a few times over
inline void choose_tiles(int nblk, int V, bool cull, int* vtile, int* ntiles) {
    int nt = 1;
    if (nblk < 512) nt = (1024 + nblk - 1) / nblk;
    // culled kernels: the work sits in the few point blocks near the path; splitting their waypoint range over
    // grid.y spreads it over more CUs (partials are per (waypoint, wave): results do not depend on the split)
    if (cull && nt < 8) nt = 8;  // (1, 2, 4, 16 tiles measured at 1 M x 128: 0.26, 0.25, 0.21, 0.19 ms per step against 0.19 for 8)
    if (nt > V) nt = V;
    if (nt < 1) nt = 1;
    *vtile = (V + nt - 1) / nt;
    if (!cull) *vtile = (*vtile + 63) / 64 * 64;  // dense: a 64-waypoint word of the touched/need masks has one writer
    *ntiles = (V + *vtile - 1) / *vtile;
}

template <typename F>
inline void dispatch(int P, bool pinhole, bool cull, bool occ, F&& f) {
    auto with_p = [&](auto Pc) {
        auto with_o = [&](auto Oc) {
            if (pinhole) { if (cull) f(Pc, std::true_type(), std::true_type(), Oc); else f(Pc, std::true_type(), std::false_type(), Oc); }
            else { if (cull) f(Pc, std::false_type(), std::true_type(), Oc); else f(Pc, std::false_type(), std::false_type(), Oc); }
        };
        if (occ) with_o(std::true_type()); else with_o(std::false_type());
    };
    if (P == 4) with_p(std::integral_constant<int, 4>());
    else if (P == 2) with_p(std::integral_constant<int, 2>());
    else with_p(std::integral_constant<int, 1>());
}

inline int rig_cams(const tohip_rig* rig) { return (rig && rig->n_cams > 0 && rig->rig_quats) ? rig->n_cams : 1; }

}  // namespace

extern "C" size_t tohip_traj_workspace_bytes(int64_t n_points, int64_t n_virtual) {
    if (n_points <= 0 || n_virtual <= 0) return 0;
    return make_plan(n_points, n_virtual, n_virtual).total;
}

extern "C" int tohip_traj_forward(const void* packed, int64_t n, const float* poses, const float* quats, int64_t W,
                                  const tohip_camera* cam, const tohip_rig* rig, int flags, const uint32_t* occlusion_bits,
                                  float* lo_sum, float* minmax, void* need_mask_out, void* workspace, size_t workspace_bytes,
                                  void* stream_) {
    if (!packed || !poses || !quats || !cam || !lo_sum || !minmax || !workspace || n <= 0 || W <= 0) return TOHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    const int C = rig_cams(rig);
    const int64_t V = W * C;
    if (V > (1 << 24)) return TOHIP_EINVAL;
    const TrajPlan pl = make_plan(n, V, W);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    char* ws = (char*)workspace;
    WayHot* hot = (WayHot*)(ws + pl.off_hot);
    WayCold* cold = (WayCold*)(ws + pl.off_cold);
    WayAux* aux = (WayAux*)(ws + pl.off_aux);
    float2* mm = (float2*)(ws + pl.off_mm);
    const CamConsts cc = make_consts(cam);
    const CloudView cv = cloud_view(packed, n);
    const bool cull = !(flags & TOHIP_TRAJ_DENSE);
    const float* rq = (C > 1 || (rig && rig->rig_quats)) ? rig->rig_quats : nullptr;
    const float* rt = rq ? rig->rig_trans : nullptr;

    {
        TO_PROF(TOHIP_PROF_SMALL, st);
        if (cull) {
            int step = (int)(n / 4096);
            if (step < 1) step = 1;
            if (cc.pinhole)
                k_traj_probe<true><<<(int)V, TO_PROBE_THREADS, 0, st>>>(cv, poses, quats, C, rq, rt, hot, cold, aux, cc, step,
                                                                        occlusion_bits, cv.npad / 32);
            else
                k_traj_probe<false><<<(int)V, TO_PROBE_THREADS, 0, st>>>(cv, poses, quats, C, rq, rt, hot, cold, aux, cc, step,
                                                                         occlusion_bits, cv.npad / 32);
        } else {
            k_prep_waycams<<<(int)((V + 127) / 128), 128, 0, st>>>(poses, quats, (int)W, C, rq, rt, hot, cold, aux);
        }
        TO_HIP_CHECK_LAUNCH();
    }
    int vtile, ntiles;
    choose_tiles(pl.nblk, (int)V, cull, &vtile, &ntiles);
    {
        TO_PROF(TOHIP_PROF_PASS1, st);
        dispatch(pl.P, cc.pinhole != 0, cull, occlusion_bits != nullptr, [&](auto Pc, auto Ph, auto Cu, auto Oc) {
            k_traj_pass1<decltype(Pc)::value, decltype(Ph)::value, decltype(Cu)::value, decltype(Oc)::value>
                <<<dim3(pl.nblk, ntiles), TO_BLOCK, 0, st>>>(cv, hot, aux, (int)V, vtile, cc, mm, pl.nslots, occlusion_bits,
                                                             cv.npad / 32);
        });
    }
    TO_HIP_CHECK_LAUNCH();
    {
        TO_PROF(TOHIP_PROF_SMALL, st);
        k_minmax_finish<<<(int)V, 1024, 0, st>>>(mm, pl.nslots, hot, aux, cc.inv_var, cull ? (need_mask_out ? 2 : 1) : 0, minmax);
    }
    TO_HIP_CHECK_LAUNCH();
    TO_PROF(TOHIP_PROF_PASS2, st);
    // Culled mode: one point per lane.  The little work there is sits in the few tiles near the path, where a wave walks
    // its survivors one after the other; pass 2 has no per-wave partials, so the points-per-lane factor is free to choose
    // and P = 1 cuts that tail four-fold (1 M x 128: 45 -> 28 us).  The result does not depend on P.
    const int P2 = cull ? 1 : pl.P;
    const int wps = pl.P / P2;  // waves of this launch per backward wave slot (64 * pl.P points)
    dispatch(P2, cc.pinhole != 0, cull, occlusion_bits != nullptr, [&](auto Pc, auto Ph, auto Cu, auto Oc) {
        constexpr int Pv = decltype(Pc)::value;
        constexpr bool Phv = decltype(Ph)::value, Cuv = decltype(Cu)::value, Ocv = decltype(Oc)::value;
        const dim3 grid((unsigned)(pl.npad / (TO_BLOCK * P2)), 1);
        if (need_mask_out)
            k_traj_pass2<Pv, Phv, Cuv, Ocv, true><<<grid, TO_BLOCK, 0, st>>>(cv, hot, (int)V, cc, lo_sum, occlusion_bits, cv.npad / 32,
                                                                             (unsigned long long*)need_mask_out, pl.nslots, wps);
        else
            k_traj_pass2<Pv, Phv, Cuv, Ocv><<<grid, TO_BLOCK, 0, st>>>(cv, hot, (int)V, cc, lo_sum, occlusion_bits, cv.npad / 32);
    });
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_traj_reward(const void* packed, const float* lo_sum, int64_t n, float eps, float* rewards,
                                 float* scalars, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!packed || !lo_sum || !rewards || !scalars || !workspace || n <= 0) return TOHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    if (workspace_bytes < 4096 * sizeof(double)) return TOHIP_ENOSPC;
    double* rpart = (double*)workspace;  // TrajPlan::off_rpart == 0
    const CloudView cv = cloud_view(packed, n);
    int nb = (int)((n + TO_BLOCK - 1) / TO_BLOCK);
    if (nb > 2048) nb = 2048;
    TO_PROF(TOHIP_PROF_REWARD, st);
    k_reward<<<nb, TO_BLOCK, 0, st>>>(lo_sum, cv.inv, n, rewards, rpart);
    TO_HIP_CHECK_LAUNCH();
    k_reward_finish<<<1, TO_BLOCK, 0, st>>>(rpart, nb, n, eps, scalars);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" size_t tohip_traj_need_mask_bytes(int64_t n, int64_t n_virtual) {
    if (n <= 0 || n_virtual <= 0) return 256;
    const TrajPlan pl = make_plan(n, n_virtual, n_virtual);
    return align_up((size_t)((n_virtual + 63) / 64) * (size_t)pl.nslots * sizeof(unsigned long long), 256);
}

extern "C" int tohip_traj_backward_scan(const void* packed, int64_t n, const float* poses, const float* quats, int64_t W,
                                        const tohip_camera* cam, const tohip_rig* rig, int flags,
                                        const uint32_t* occlusion_bits, const float* minmax, void* need_mask,
                                        void* workspace, size_t workspace_bytes, void* stream_) {
    if (!packed || !poses || !quats || !cam || !minmax || !need_mask || !workspace || n <= 0 || W <= 0) return TOHIP_EINVAL;
    if (!(flags & TOHIP_TRAJ_DENSE)) return TOHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    const int C = rig_cams(rig);
    const int64_t V = W * C;
    const TrajPlan pl = make_plan(n, V, W);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    char* ws = (char*)workspace;
    WayHot* hot = (WayHot*)(ws + pl.off_hot);
    WayCold* cold = (WayCold*)(ws + pl.off_cold);
    WayAux* aux = (WayAux*)(ws + pl.off_aux);
    const CamConsts cc = make_consts(cam);
    const CloudView cv = cloud_view(packed, n);
    const float* rq = (C > 1 || (rig && rig->rig_quats)) ? rig->rig_quats : nullptr;
    const float* rt = rq ? rig->rig_trans : nullptr;
    {
        TO_PROF(TOHIP_PROF_SMALL, st);
        k_prep_waycams<<<(int)((V + 127) / 128), 128, 0, st>>>(poses, quats, (int)W, C, rq, rt, hot, cold, aux, minmax,
                                                                cc.inv_var, 0);
        TO_HIP_CHECK_LAUNCH();
    }
    int vtile, ntiles;
    choose_tiles(pl.nblk, (int)V, false, &vtile, &ntiles);
    {
        TO_PROF(TOHIP_PROF_BWD, st);
        dispatch(pl.P, cc.pinhole != 0, false, occlusion_bits != nullptr, [&](auto Pc, auto Ph, auto Cu, auto Oc) {
            if constexpr (!decltype(Cu)::value)
                k_traj_bwd_scan<decltype(Pc)::value, decltype(Ph)::value, decltype(Oc)::value>
                    <<<dim3(pl.nblk, ntiles), TO_BLOCK, 0, st>>>(cv, hot, (int)V, vtile, cc, (unsigned long long*)need_mask,
                                                                 pl.nslots, occlusion_bits, cv.npad / 32);
        });
    }
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_traj_backward(const void* packed, int64_t n, const float* poses, const float* quats, int64_t W,
                                   const tohip_camera* cam, const tohip_rig* rig, int flags,
                                   const uint32_t* occlusion_bits, const float* lo_sum, const float* grad_rewards,
                                   const float* scalars, const float* minmax,
                                   const float* gout, const void* need_mask, float* poses_grad, float* quats_grad,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    if (!packed || !poses || !quats || !cam || !lo_sum || !minmax || !poses_grad || !quats_grad || !workspace ||
        n <= 0 || W <= 0 || (!grad_rewards && (!scalars || !gout)))
        return TOHIP_EINVAL;

    hipStream_t st = (hipStream_t)stream_;
    const int C = rig_cams(rig);
    const int64_t V = W * C;
    const TrajPlan pl = make_plan(n, V, W);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    char* ws = (char*)workspace;
    WayHot* hot = (WayHot*)(ws + pl.off_hot);
    WayCold* cold = (WayCold*)(ws + pl.off_cold);
    WayAux* aux = (WayAux*)(ws + pl.off_aux);
    float* bpart = (float*)(ws + pl.off_bpart);
    float* ties = (float*)(ws + pl.off_ties);
    float* vgrad = (float*)(ws + pl.off_vgrad);
    unsigned long long* tmask = (unsigned long long*)(ws + pl.off_tmask);
    const CamConsts cc = make_consts(cam);
    const CloudView cv = cloud_view(packed, n);
    // with a need mask (from tohip_traj_forward or tohip_traj_backward_scan) the flagged (wave, waypoint) combinations are
    // walked directly, whatever the mode: no culling tests, partials recorded in the touched mask
    const bool cull = !(flags & TOHIP_TRAJ_DENSE) && !need_mask;
    const float* rq = (C > 1 || (rig && rig->rig_quats)) ? rig->rig_quats : nullptr;
    const float* rt = rq ? rig->rig_trans : nullptr;

    {
        // the workspace may have been reused since the forward: rebuild the waypoint records
        TO_PROF(TOHIP_PROF_SMALL, st);
        k_prep_waycams<<<(int)((V + 127) / 128), 128, 0, st>>>(poses, quats, (int)W, C, rq, rt, hot, cold, aux, minmax,
                                                                cc.inv_var, cull ? 1 : 0, ties);
        TO_HIP_CHECK_LAUNCH();
    }
    int vtile, ntiles;
    choose_tiles(pl.nblk, (int)V, cull, &vtile, &ntiles);
    if (need_mask) {
        // the masked half only walks flagged waypoints, a handful per wave near the path: a quarter of a mask word
        // (16 waypoints) per block row spreads those tails over more CUs
        vtile = 16;
        ntiles = (int)((V + 15) / 16);
    }
    {
        TO_PROF(TOHIP_PROF_BWD, st);
        dispatch(pl.P, cc.pinhole != 0, cull, occlusion_bits != nullptr, [&](auto Pc, auto Ph, auto Cu, auto Oc) {
            constexpr int Pv = decltype(Pc)::value;
            constexpr bool Phv = decltype(Ph)::value, Cuv = decltype(Cu)::value, Ocv = decltype(Oc)::value;
            if constexpr (!Cuv) {
                if (need_mask) {
                    k_traj_bwd<Pv, Phv, false, Ocv, true><<<dim3(pl.nblk, ntiles), TO_BLOCK, 0, st>>>(
                        cv, hot, aux, (int)V, vtile, cc, lo_sum, grad_rewards, scalars, gout, bpart, pl.nslots, ties,
                        occlusion_bits, cv.npad / 32, tmask, (const unsigned long long*)need_mask);
                    return;
                }
            }
            k_traj_bwd<Pv, Phv, Cuv, Ocv><<<dim3(pl.nblk, ntiles), TO_BLOCK, 0, st>>>(
                cv, hot, aux, (int)V, vtile, cc, lo_sum, grad_rewards, scalars, gout, bpart, pl.nslots, ties, occlusion_bits,
                cv.npad / 32, tmask);
        });
    }
    TO_HIP_CHECK_LAUNCH();
    TO_PROF(TOHIP_PROF_SMALL, st);
    const int shift = pl.P == 4 ? 0 : (pl.P == 2 ? 1 : 2);
    const bool single = C == 1 && rq == nullptr;
    k_bwd_finish1<<<(int)V, TO_FINISH_THREADS, 0, st>>>(bpart, pl.nslots, shift, ties, hot, cv.bounds, cc.mean, cull ? 1 : 0, tmask, vgrad,
                                               cold, single ? 1 : 0, poses_grad, quats_grad);
    TO_HIP_CHECK_LAUNCH();
    if (!single) {
        k_bwd_finish2<<<(int)((W + 63) / 64), 64, 0, st>>>(vgrad, hot, cold, (int)W, C, rq, rt, poses_grad, quats_grad);
        TO_HIP_CHECK_LAUNCH();
    }
    return TOHIP_OK;
}
