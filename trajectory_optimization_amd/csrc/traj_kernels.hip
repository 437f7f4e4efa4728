// traj_kernels.hip — ModelTraj visibility term, forward + analytic backward, for gfx950.
//
// Replaces the per-waypoint Python loop of /root/reference/src/model.py:217-231 (forward),
// :237/:246 (rewards, visibility loss) and its torch-autograd backward (SURVEY.md §8a rows A-C,E,G).
//
// Every (point, waypoint) pair is evaluated ONCE, in pass 1, which only keeps the per-waypoint extrema; everything
// after it works on the few pairs that can contribute:
//
//   k_traj_prep / k_traj_probe   waypoint records (WayRec, common.hpp); the probe (CULL mode) also samples the cloud
//   k_traj_pass1                 p of every pair -> (min, max) per (waypoint, wave of 256 points)   the dense kernel
//   k_traj_select                block per waypoint: a = min p, M = max (p - a); a (256-point slot, waypoint) pair
//                                is FLAGGED when its maximum reaches p_hat >= 1/2 (or it holds an argmin point and
//                                a > 0): only flagged pairs have a non-zero log-odds term or a gradient.  Flags in
//                                both orientations, the list of flagged slots, the list of flagged pairs, and the
//                                rows that hold the extrema (tie sets)
//   k_traj_lo_sparse             block per flagged slot: log-odds of its flagged waypoints, summed in a fixed order
//   k_traj_reward                rewards = sigmoid(lo_sum) in the caller's order, mean, visibility loss (one launch)
//   k_traj_bwd_sparse            wave per flagged pair: the 14 gradient sums of its 256 points
//   k_traj_bwd_finish            block per waypoint: partials in slot order (f64), argmin/argmax shares from the
//                                recorded rows (deterministic: no float atomics), chain to (position, quaternion)
//
// About 0.7 % of the (slot, waypoint) pairs are flagged on the BASELINE workloads (1 M x 128: 3 546 of 500 224), so
// the sparse kernels are launch-latency sized and pass 1 is the whole cost: ~N*16 B of HBM traffic per launch, bound
// by VALU issue (26 FMA-class + 4 transcendental instructions per evaluation, common.hpp).
//
// Data layout in HBM
//   cloud     Morton-sorted SoA x|y|z (npad each) + permutation + one bounding sphere per 256 points
//             (packed once, tohip_pack_cloud)                                          16 B/point
//   WayRec    two 64-B lines per virtual waypoint; line 0 -> SGPRs by one s_load_dwordx16
//   lo_sum (sorted order) / rewards (original order)                                    4 B/point each
//   part      [virtual waypoint][slot]: (min, max) of p over the 256 points of a wave       8 B
//   bpart     [virtual waypoint][slot]: 14 gradient sums of a flagged pair                 64 B (written where flagged)
//
// Two evaluation modes with bitwise identical results:
//   DENSE  pass 1 evaluates every (point, waypoint) pair (the streaming reference semantics; bench headline)
//   CULL   pass 1 skips pairs that provably can neither be a waypoint's maximum nor be flagged: p <= 2^(-cd d2)
//          bounds p by the squared distance d2 = |y - sp|^2; a wave tests its tile's bounding sphere against
//          64 waypoints at once, then the per-point d2.  The bound is L/2 with L an attained value of p found by a
//          strided probe, which also proves min p == 0 by exhibiting a zero (else that waypoint is searched densely).
//
// The forward leaves its state (records, flags, lists) in the workspace; the backward reads it there: the workspace
// must not be touched between tohip_traj_forward and tohip_traj_backward of the same step.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "profile.hpp"

#define TO_SLOT 256            // points per flag slot (= bounding-sphere tile)
#define TO_TIE_CAP 7           // recorded slots per extremum and waypoint; more -> the finish kernel scans all slots
#define TO_BWD_NSUM 14

// ---------------------------------------------------------------------------------------------
// workspace control block

// head of the workspace: k_traj_reward's accumulators, one 64-bit word per trajectory (arrivals | NaN marks | fixed-point
// sum); zero between launches

struct TieRec {            // slots (ascending) whose max equals the waypoint's max / whose min equals its min (a > 0)
    int nmax, nmin;        // counts; > TO_TIE_CAP: overflow, scan every slot
    int maxrow[TO_TIE_CAP];
    int minrow[TO_TIE_CAP];
};

// ---------------------------------------------------------------------------------------------
// virtual waypoint records: thread per virtual waypoint v = w*C + c.  F.normalize (model.py:53), rig composition
// R_v = R(qn_w) R(q_c), t_v = t_w + R(qn_w) l_c, then the three projection rows and the Gaussian's centre.

// traj_off (optional): n_traj + 1 ascending body-waypoint offsets of several trajectories laid end to end
__device__ __forceinline__ void prep_wayrec(int v, const float* __restrict__ poses, const float* __restrict__ quats, int C,
                                            const float* __restrict__ rig_q, const float* __restrict__ rig_t,
                                            const EvalK& k, WayRec* __restrict__ rec, WayCold* __restrict__ cold,
                                            const int* __restrict__ traj_off, int n_traj) {
    const int w = v / C, c = v - w * C;
    int seg = 0;
    if (traj_off != nullptr)
        while (seg + 1 < n_traj && w >= traj_off[seg + 1]) ++seg;
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
    WayRec r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[3 * i + j] = R[3 * j + i];
    r.t[0] = t[0]; r.t[1] = t[1]; r.t[2] = t[2];
    // rows of K m (double, rounded once): h_i = sum_j K[i][j] c_j = sum_k (sum_j K[i][j] m[j][k]) y_k
    for (int kk = 0; kk < 3; ++kk) {
        double h0 = 0, h1 = 0, h2 = 0, s = 0;
        for (int j = 0; j < 3; ++j) {
            const double mjk = (double)r.m[3 * j + kk];
            h0 += (double)k.k[j] * mjk;
            h1 += (double)k.k[3 + j] * mjk;
            h2 += (double)k.k[6 + j] * mjk;
            s += mjk;   // (R mu)_k = mu * sum_j m[j][k]
        }
        r.f0[kk] = (float)((double)k.su * h0);
        r.f1[kk] = (float)((double)k.sv * h1);
        r.f2[kk] = (float)h2;
        r.sp[kk] = (float)((double)k.mean * s);
    }
    r.a = 0.f; r.invM = 1.f; r.M = 1.f; r.L = 0.f; r.thr1 = INFINITY; r.sthr1 = INFINITY; r.azero = 0.f; r.seg = seg;
    rec[v] = r;
}

// clears what k_traj_select accumulates into (the slot-major flags)
__device__ __forceinline__ void clear_select_state(unsigned long long* __restrict__ ft, int64_t ft_words) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < ft_words; i += nth) ft[i] = 0ull;
}

__global__ void __launch_bounds__(256)
k_traj_prep(const float* __restrict__ poses, const float* __restrict__ quats, int V, int C, const float* __restrict__ rig_q,
            const float* __restrict__ rig_t, EvalK k, WayRec* __restrict__ rec, WayCold* __restrict__ cold,
            unsigned long long* __restrict__ ft, int64_t ft_words, const int* __restrict__ traj_off, int n_traj) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V) prep_wayrec(v, poses, quats, C, rig_q, rig_t, k, rec, cold, traj_off, n_traj);
    clear_select_state(ft, ft_words);
}

// ---------------------------------------------------------------------------------------------
// point loads: lane owns P consecutive points

template <int P>
__device__ __forceinline__ void load_points(const float* __restrict__ soa, int64_t npad, int64_t base, float (&x)[P],
                                            float (&y)[P], float (&z)[P]) {
#pragma unroll
    for (int j = 0; j < P; j += 4) {
        const float4 a = *reinterpret_cast<const float4*>(soa + base + j);
        const float4 b = *reinterpret_cast<const float4*>(soa + npad + base + j);
        const float4 c = *reinterpret_cast<const float4*>(soa + 2 * npad + base + j);
        x[j] = a.x; x[j + 1] = a.y; x[j + 2] = a.z; x[j + 3] = a.w;
        y[j] = b.x; y[j + 1] = b.y; y[j + 2] = b.z; y[j + 3] = b.w;
        z[j] = c.x; z[j + 1] = c.y; z[j + 2] = c.z; z[j + 3] = c.w;
    }
}

// Optional per-(virtual waypoint, point) occlusion bits in packed order: row v holds npad bits,
// bit i = 1 when sorted point i is NOT occluded from waypoint v.  om[] = 1.0f / 0.0f multipliers of p; without a
// bit array every multiplier is a compile-time one (p * 1.0f == p: the unoccluded results do not change by a bit).
template <int P, bool OCC>
__device__ __forceinline__ void load_occ(const uint32_t* __restrict__ occ, int64_t occw, int v, int64_t base, float (&om)[P]) {
    if constexpr (!OCC) {
#pragma unroll
        for (int i = 0; i < P; ++i) om[i] = 1.0f;
    } else {
        const unsigned bits = occ[(int64_t)v * occw + (base >> 5)] >> (unsigned)(base & 31);  // P <= 8 consecutive bits, aligned to P
#pragma unroll
        for (int i = 0; i < P; ++i) om[i] = ((bits >> i) & 1u) ? 1.0f : 0.0f;
    }
}
__device__ __forceinline__ float occ_one(const uint32_t* __restrict__ occ, int64_t occw, int v, int64_t i) {
    if (!occ) return 1.0f;
    return ((occ[(int64_t)v * occw + (i >> 5)] >> (unsigned)(i & 31)) & 1u) ? 1.0f : 0.0f;
}

// wave-uniform bounding sphere of the 256-point tile this wave's points belong to (P = 4: one tile per wave)
__device__ __forceinline__ float4 wave_tile_bound(const CloudView& cv, int64_t base) {
    const int tile = __builtin_amdgcn_readfirstlane((int)(base >> 8));
    float4 b = cv.bounds[tile];
    b.x = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.x)));
    b.y = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.y)));
    b.z = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.z)));
    b.w = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, b.w)));
    return b;
}

// squared distance of a world point from the Gaussian's centre of record r — the expression vis_p uses
__device__ __forceinline__ float dist2_sp(const WayRec& r, float x, float y, float z) {
    const float d0 = (x - r.t[0]) - r.sp[0], d1 = (y - r.t[1]) - r.sp[1], d2 = (z - r.t[2]) - r.sp[2];
    return fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
}

// (tile, waypoint) liveness for 64 waypoints at once: lane l tests waypoint vc + l against the wave's tile; the ballot is
// the set of waypoints whose sphere {d2 <= thr1} may reach the tile.  Waypoints without the probe's "min is zero" proof
// always survive (they are searched densely).
__device__ __forceinline__ unsigned long long tile_survivors(const WayRec* __restrict__ rec, int vc, int v1, const float4& tb) {
    const int v = vc + (int)(threadIdx.x & 63);
    bool ok = false;
    if (v < v1) {
        const float4* rp = reinterpret_cast<const float4*>(rec + v);
        const float4 q0 = rp[0], q3 = rp[3], q4 = rp[4], q5 = rp[5];  // t0 t1 t2 f00 | sp0 sp1 sp2 a | invM M L thr1 | sthr1 azero ..
        const float d0 = (tb.x - q0.x) - q3.x, d1 = (tb.y - q0.y) - q3.y, d2 = (tb.z - q0.z) - q3.z;
        const float D2 = fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
        const float thr = q4.w, sthr = q5.x;
        const float bound = fmaf(tb.w, fmaf(2.0f, sthr, tb.w), thr) * 1.00001f;  // (sthr + r)^2, rounded up
        ok = (q5.y == 0.f) || !(D2 > bound);
    }
    return __ballot(ok);
}

// Squared-distance bound thr such that  d2 > thr  =>  2^(-cd d2) < tau * (1 - 1e-4): p = S * 2^-A <= 2^(-cd d2) cannot
// reach tau.  The 1e-4 margin covers every rounding in p (~1e-6).  tau outside (0,1) -> +inf (never cull).
__device__ inline void cull_bound(float tau, float inv_var, float* thr, float* sthr) {
    cull_threshold(tau, inv_var, thr, sthr);
}

// ---------------------------------------------------------------------------------------------
// probe (CULL mode): block per virtual waypoint evaluates a strided sample of the sorted cloud.
//   L     = max p over the sample  (a lower bound of the true max: an attained value of p)
//   azero = some sample has p == 0 exactly  =>  min_n p == 0 (p is never negative)
// The block first builds its waypoint's record (k_traj_prep's work: one launch less), and it is 1024 threads wide:
// the samples are scattered single loads, so the kernel is as long as one thread's chain of them.
// THREADS = 1024 for up to a few hundred waypoints (the shortest chain); 256 beyond (eight blocks to a CU instead of two: the
// 1 024 waypoints of eight concurrent trajectories no longer queue).  Maximum and "some p is zero" do not depend on the order.
template <int TO_PROBE_THREADS>
__global__ void __launch_bounds__(TO_PROBE_THREADS)
k_traj_probe(CloudView cv, const float* __restrict__ poses, const float* __restrict__ quats, int C,
             const float* __restrict__ rig_q, const float* __restrict__ rig_t, EvalK k, WayRec* __restrict__ rec,
             WayCold* __restrict__ cold, const uint32_t* __restrict__ occ, int64_t occw,
             unsigned long long* __restrict__ ft, int64_t ft_words, const int* __restrict__ traj_off, int n_traj) {
    __shared__ float smx[TO_PROBE_THREADS / 64];
    __shared__ int szero[TO_PROBE_THREADS / 64];
    const int v = blockIdx.x, t = threadIdx.x;
    clear_select_state(ft, ft_words);
    // the samples (a contiguous copy of every step-th sorted point, made at pack time) are requested first, the record is
    // built meanwhile
    constexpr int kBatch = 8, kRounds = TO_PROBE_MAX / (TO_PROBE_THREADS * kBatch);
    float px[kBatch], py[kBatch], pz[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) {
        const int sj = t + j * TO_PROBE_THREADS;
        const int sc = sj < cv.nsamples ? sj : 0;
        px[j] = cv.samples[sc]; py[j] = cv.samples[TO_PROBE_MAX + sc]; pz[j] = cv.samples[2 * TO_PROBE_MAX + sc];
    }
    if (t == 0) prep_wayrec(v, poses, quats, C, rig_q, rig_t, k, rec, cold, traj_off, n_traj);
    __syncthreads();
    const WayRec r = rec[v];
    float mx = 0.f;
    int zero = 0;
    for (int rd = 0; rd < kRounds; ++rd) {
        if (rd > 0) {
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int sj = t + (rd * kBatch + j) * TO_PROBE_THREADS;
                const int sc = sj < cv.nsamples ? sj : 0;
                px[j] = cv.samples[sc]; py[j] = cv.samples[TO_PROBE_MAX + sc]; pz[j] = cv.samples[2 * TO_PROBE_MAX + sc];
            }
        }
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int sj = t + (rd * kBatch + j) * TO_PROBE_THREADS;
            if (sj < cv.nsamples) {
                const float p = vis_p(r, k, px[j], py[j], pz[j]) * occ_one(occ, occw, v, (int64_t)sj * cv.sample_step);
                mx = fmaxf(mx, p);
                zero |= (p == 0.f);
            }
        }
    }
    for (int s = 32; s > 0; s >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, s)); zero |= __shfl_xor(zero, s); }
    if ((t & 63) == 0) { smx[t >> 6] = mx; szero[t >> 6] = zero; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < TO_PROBE_THREADS / 64; ++w) { mx = fmaxf(mx, smx[w]); zero |= szero[w]; }
        float thr, sthr;
        cull_bound(0.5f * mx, k.inv_var, &thr, &sthr);   // d2 > thr  =>  p < L/2 <= M/2: neither the max nor flagged
        rec[v].L = mx;
        rec[v].thr1 = thr;
        rec[v].sthr1 = sthr;
        rec[v].azero = zero ? 1.f : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// pass 1: p of every (point, waypoint) pair of the block's tile -> per-wave (min, max).  grid = (point blocks, waypoint
// tiles).  A lane owns 4 consecutive sorted points, a wave one 256-point slot; part[v * nslots + slot] = (min, max).

#define TO_P 4   // points per lane: a wave is one slot (and one bounding-sphere tile)

__device__ __forceinline__ void pass1_eval(const EvalK& k, const WayRec& r, const float (&x)[TO_P], const float (&y)[TO_P],
                                           const float (&z)[TO_P], const float (&om)[TO_P], float& mn, float& mx) {
    const f2 p0 = vis_p_pk(r, k, f2{x[0], x[1]}, f2{y[0], y[1]}, f2{z[0], z[1]}) * f2{om[0], om[1]};
    const f2 p1 = vis_p_pk(r, k, f2{x[2], x[3]}, f2{y[2], y[3]}, f2{z[2], z[3]}) * f2{om[2], om[3]};
    mn = fminf(fminf(p0.x, p0.y), fminf(p1.x, p1.y));   // p >= +0 always
    mx = fmaxf(fmaxf(p0.x, p0.y), fmaxf(p1.x, p1.y));
}

// the log-odds vector starts from zero (k_traj_lo_sparse fills the flagged slots) and, when the caller asks for it, the rewards
// vector from sigmoid(0) = 1/2 (k_traj_reward then only stores the others): done by whoever evaluates a point block's first waypoint
typedef float f4v __attribute__((ext_vector_type(4)));
struct OutInit {   // n_traj log-odds vectors of npad floats (and rewards vectors of n floats) one after the other
    float* lo_zero;
    float* rewards_half;
    int64_t npad, n;
    int n_traj;
};
__device__ __forceinline__ void init_outputs(int64_t base, const OutInit& o) {
    // streaming stores: nothing of this is read again by this kernel
    for (int b = 0; b < o.n_traj; ++b) {
        __builtin_nontemporal_store(f4v{0.f, 0.f, 0.f, 0.f}, reinterpret_cast<f4v*>(o.lo_zero + (int64_t)b * o.npad + base));
        if (o.rewards_half != nullptr && base < o.n) {
            float* rh = o.rewards_half + (int64_t)b * o.n;
            if (base + 4 <= o.n && ((((int64_t)b * o.n) & 3) == 0))   // (n need not be a multiple of 4: a later vector may start unaligned)
                __builtin_nontemporal_store(f4v{0.5f, 0.5f, 0.5f, 0.5f}, reinterpret_cast<f4v*>(rh + base));
            else
                for (int64_t i = base; i < base + 4 && i < o.n; ++i) rh[i] = 0.5f;
        }
    }
}

// DENSE: persistent blocks, as many as the chip holds at once (host: occupancy x CUs).  A lane owns EIGHT consecutive points
// (four independent packed evaluation chains per waypoint: a SIMD holds only ~5 of these waves, which issue in age order, so the
// instruction-level parallelism has to come from inside the wave), a wave two 256-point slots, a block 2048 points.  The
// (point block, waypoint) pairs are one flat range cut into gridDim.x equal pieces: a block's piece is a run of consecutive
// waypoints of one point block (seldom two), so every block does the same number of evaluations to within one waypoint, loads
// its points once (twice), and all blocks end together: no dispatch order, no tail.
#define TO_PD 8
template <bool OCC>
__global__ void __launch_bounds__(TO_BLOCK)
k_traj_pass1_dense(CloudView cv, const WayRec* __restrict__ rec, int V, int nblk, EvalK k, float2* __restrict__ part, int nslots,
                   const uint32_t* __restrict__ occ, int64_t occw, OutInit oi, unsigned long long* __restrict__ stamps) {
    constexpr int P = TO_PD;
    const int lane = threadIdx.x & 63;
    // diagnostic only (tohip_profile_clock): shader-clock and 100 MHz real-time stamps of this block, to a buffer nothing else reads
    unsigned long long st0 = 0, sr0 = 0;
    if (stamps != nullptr && threadIdx.x == 0) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
    const int64_t total = (int64_t)nblk * V;
    int64_t u = total * blockIdx.x / gridDim.x;
    const int64_t u_end = total * (blockIdx.x + 1) / gridDim.x;
    // The SIMD arbiter serves its waves oldest first, so equal pieces do not end together: the old waves race ahead, retire, and
    // the last wave of a SIMD finishes alone at ~70 % of the issue rate.  Priority outranks age: a wave steps its priority down
    // as it completes quarters of its piece, so whoever is behind is served first and all waves stay within a quarter of each other.
    const int64_t q1 = u + (u_end - u) / 4, q2 = u + (u_end - u) / 2, q3 = u + 3 * (u_end - u) / 4;
    __builtin_amdgcn_s_setprio(3);
    while (u < u_end) {
        const int pb = (int)(u / V);
        const int v0 = (int)(u - (int64_t)pb * V);
        const int v1 = (int)min((int64_t)V, v0 + (u_end - u));
        const int gthread = pb * TO_BLOCK + threadIdx.x;
        const int64_t base = (int64_t)gthread * P;
        const int slot = gthread >> 5;   // 32 lanes x 8 points
        float x[P], y[P], z[P];
        load_points<P>(cv.soa, cv.npad, base, x, y, z);
        if (v0 == 0) { init_outputs(base, oi); init_outputs(base + 4, oi); }
        for (int v = v0; v < v1; ++v) {
            const int64_t uu = u + (v - v0);
            if (uu == q1) __builtin_amdgcn_s_setprio(2);
            if (uu == q2) __builtin_amdgcn_s_setprio(1);
            if (uu == q3) __builtin_amdgcn_s_setprio(0);
            const WayRec& r = rec[v];
            float om[P];
            load_occ<P, OCC>(occ, occw, v, base, om);
            f2 p[P / 2];
#pragma unroll
            for (int i = 0; i < P; i += 2)
                p[i / 2] = vis_p_pk(r, k, f2{x[i], x[i + 1]}, f2{y[i], y[i + 1]}, f2{z[i], z[i + 1]}) * f2{om[i], om[i + 1]};
            float mn = fminf(fminf(fminf(p[0].x, p[0].y), fminf(p[1].x, p[1].y)), fminf(fminf(p[2].x, p[2].y), fminf(p[3].x, p[3].y)));
            float mx = fmaxf(fmaxf(fmaxf(p[0].x, p[0].y), fmaxf(p[1].x, p[1].y)), fmaxf(fmaxf(p[2].x, p[2].y), fmaxf(p[3].x, p[3].y)));
            mn = half_min31_nn_fused(mn);
            mx = half_max31_nn_fused(mx);
            if ((lane & 31) == 31) part[(int64_t)v * nslots + slot] = make_float2(mn, mx);
        }
        u += v1 - v0;
    }
    if (stamps != nullptr && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - sr0;
        unsigned long long* ext = stamps + 2 * (int64_t)gridDim.x + 4 * (int64_t)blockIdx.x;   // where and when the block ran
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ext[0] = sr0; ext[1] = __builtin_amdgcn_s_memrealtime(); ext[2] = hw; ext[3] = xcc;
    }
}

// CULL: grid = (point blocks, waypoint tiles of <= 64): one ballot covers the block row's waypoints.  A live (tile, waypoint)
// pair is evaluated like in the dense kernel (packed, every point of the wave): per-point distance tests cost as much as they
// save once the tile is live.  The work sits in the few point blocks near the path, hence many short block rows.
template <bool OCC>
__global__ void __launch_bounds__(TO_BLOCK)
k_traj_pass1_cull(CloudView cv, const WayRec* __restrict__ rec, int V, int vtile, EvalK k, float2* __restrict__ part, int nslots,
                  const uint32_t* __restrict__ occ, int64_t occw, OutInit oi) {
    constexpr int P = TO_P;
    const int lane = threadIdx.x & 63;
    const int gthread = blockIdx.x * TO_BLOCK + threadIdx.x;
    const int64_t base = (int64_t)gthread * P;
    const int slot = gthread >> 6;
    float x[P], y[P], z[P];
    load_points<P>(cv.soa, cv.npad, base, x, y, z);
    if (blockIdx.y == 0) init_outputs(base, oi);
    const int v0 = blockIdx.y * vtile;
    const int v1 = min(V, v0 + vtile);
    const float4 tb = wave_tile_bound(cv, base);
    unsigned long long live = tile_survivors(rec, v0, v1, tb);
    // waypoints that cannot be affected from this tile: min is the proven 0, max unknown (-inf: never flagged)
    if (v0 + lane < v1 && !((live >> lane) & 1ull)) part[(int64_t)(v0 + lane) * nslots + slot] = make_float2(0.f, -INFINITY);
    while (live) {
        const int v = v0 + __builtin_ctzll(live);
        live &= live - 1ull;
        const WayRec& r = rec[v];
        float mn, mx, om[P];
        load_occ<P, OCC>(occ, occw, v, base, om);
        pass1_eval(k, r, x, y, z, om, mn, mx);
        mn = wave_min63_nn_fused(mn);
        mx = wave_max63_nn_fused(mx);
        if (lane == 63) part[(int64_t)v * nslots + slot] = make_float2(mn, mx);
    }
}

// ---------------------------------------------------------------------------------------------
// select: block per virtual waypoint.
//   sweep 1  a = min, pmax = max over the slot partials; M = pmax - a (== max(p - a): rounding is monotone)
//   sweep 2  thread per 256-point slot: flagged when its maximum has p_hat >= 1/2 — the predicate the sparse kernels
//            evaluate per point, applied to an attained value — or it holds an argmin point while a > 0 (that set carries
//            gradient, model.py:226); the slots holding the extrema are recorded for the finish kernel.
// Flags: fv[v][slot word] (finish kernel), ft[slot][v word] (forward); vlist[v][0..vcnt[v]) = the flagged slots of v
// (backward).  Everything a block appends to is its own: the only global atomics are the ft bits, fire and forget.

#define TO_SELECT_FAST_SLOTS 4096   // up to this many slots (1 M points) a thread keeps its slots' partials in registers
// TO_SELECT_THREADS = 1024 for up to a few hundred waypoints, 256 beyond (eight blocks to a CU instead of two); minima, maxima, flags
// and the sorted tie rows do not depend on it
template <bool FAST, int TO_SELECT_THREADS>
__global__ void __launch_bounds__(TO_SELECT_THREADS)
k_traj_select(const float2* __restrict__ part, int nslots, int V, WayRec* __restrict__ rec, int cull, float* __restrict__ minmax,
              unsigned long long* __restrict__ fv, int fv_words, unsigned long long* __restrict__ ft, int vwords,
              int* __restrict__ vlist, int* __restrict__ vcnt, TieRec* __restrict__ ties, float* __restrict__ lo_sum, int64_t npad,
              int* __restrict__ sflag, int* __restrict__ slist, int64_t nmark, const int* __restrict__ toff, int C, float inv_var) {
    __shared__ float smn[TO_SELECT_THREADS / 64], smx[TO_SELECT_THREADS / 64];
    __shared__ float s_a, s_pmax;
    __shared__ int s_nmax, s_nmin, s_npairs, s_maxrow[TO_TIE_CAP], s_minrow[TO_TIE_CAP];
    const int v = blockIdx.x, t = threadIdx.x, lane = t & 63;
    const float2* pv = part + (int64_t)v * nslots;
    float mn = INFINITY, mx = -INFINITY;
    bool nan = false;
    constexpr int NQ = TO_SELECT_FAST_SLOTS / TO_SELECT_THREADS;
    float2 q[NQ];   // FAST: thread t owns slots t, t + THREADS, ...: requested at once, kept for sweep 2
    if constexpr (FAST) {
#pragma unroll
        for (int kk = 0; kk < NQ; ++kk) {
            const int s = t + kk * TO_SELECT_THREADS;
            q[kk] = s < nslots ? pv[s] : make_float2(INFINITY, -INFINITY);
        }
#pragma unroll
        for (int kk = 0; kk < NQ; ++kk) {
            mn = fminf(mn, q[kk].x);
            mx = fmaxf(mx, q[kk].y);
            nan |= (q[kk].y != q[kk].y);   // a NaN p wins the max in pass 1 (integer order): the reference's max() is NaN too
        }
    } else {
        for (int s0 = 0; s0 < nslots; s0 += 4 * TO_SELECT_THREADS) {  // four loads in flight per thread
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sidx = s0 + j * TO_SELECT_THREADS + t;
                q[j] = sidx < nslots ? pv[sidx] : make_float2(INFINITY, -INFINITY);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mn = fminf(mn, q[j].x);
                mx = fmaxf(mx, q[j].y);
                nan |= (q[j].y != q[j].y);
            }
        }
    }
    for (int s = 32; s > 0; s >>= 1) { mn = fminf(mn, __shfl_xor(mn, s)); mx = fmaxf(mx, __shfl_xor(mx, s)); }
    nan = __any(nan);
    if (lane == 0) { smn[t >> 6] = mn; smx[t >> 6] = nan ? __builtin_nanf("") : mx; }
    if (t == 0) { s_nmax = 0; s_nmin = 0; s_npairs = 0; }
    __syncthreads();
    if (t < 64) {   // wave 0: the 16 wave results
        float a = t < TO_SELECT_THREADS / 64 ? smn[t] : INFINITY, pm = t < TO_SELECT_THREADS / 64 ? smx[t] : -INFINITY;
        bool anynan = pm != pm;
        for (int s = 8; s > 0; s >>= 1) { a = fminf(a, __shfl_xor(a, s)); pm = fmaxf(pm, __shfl_xor(pm, s)); }
        anynan = __any(anynan);
        if (t == 0) {
            float pmax = pm;
            if (cull) pmax = fmaxf(pmax, rec[v].L);  // L is an attained value of p (defensive: its tile is never skipped)
            if (anynan) pmax = __builtin_nanf("");
            const float M = pmax - a;
            rec[v].a = a;
            rec[v].invM = 1.0f / M;
            rec[v].M = M;
            // the probe's bound has done its work (pass 1); from here on thr1 serves the sparse kernels: a point farther than this
            // from the Gaussian's centre has p < a + M/2, i.e. p_hat < 1/2 — no log-odds, no gradient (a whole wave of such points
            // is skipped).  Degenerate waypoints (NaN, M <= 0) keep every point.
            float thr2 = INFINITY, sthr2 = INFINITY;
            if (M > 0.f && M < INFINITY) cull_bound(a + 0.5f * M, inv_var, &thr2, &sthr2);
            rec[v].thr1 = thr2;
            rec[v].sthr1 = sthr2;
            minmax[2 * v] = a;
            minmax[2 * v + 1] = M;
            s_a = a;
            s_pmax = pmax;
        }
    }
    __syncthreads();
    const float a = s_a, pmax = s_pmax;
    const float M = pmax - a, invM = 1.0f / M;
    const bool amin = a > 0.f;
    if (!(M > 0.f) || !(invM < INFINITY)) {
        // max == min, or a NaN: the reference's p / max is 0/0 for EVERY point of this waypoint (model.py:227), so every
        // log-odds sum is NaN.  Rare: this block stores it; k_traj_lo_sparse adds onto it.
        const float nanv = __builtin_nanf("");
        float* lo_t = lo_sum + (int64_t)rec[v].seg * npad;   // its trajectory's vector
        for (int64_t i = t; i < npad; i += TO_SELECT_THREADS) lo_t[i] = nanv;
    }
    int* myl = vlist + (int64_t)v * nslots;
    // the waypoint's trajectory and that trajectory's bits of this waypoint's flag word
    const int seg = rec[v].seg;
    unsigned long long seg_mask = ~0ull;
    if (toff != nullptr) {
        const int v_lo = toff[seg] * C, v_hi = toff[seg + 1] * C, w0 = (v >> 6) * 64;
        if (w0 < v_lo) seg_mask &= ~0ull << (v_lo - w0);
        if (v_hi - w0 < 64) seg_mask &= (1ull << (v_hi - w0)) - 1ull;
    }
    auto sweep2 = [&](const float2 qq, const int s0) {
        const int s = s0 + t;
        bool flag = false;
        if (s < nslots) {
            flag = ((qq.y - a) * invM >= 0.5f) | (amin & (qq.x == a));
            if (qq.y == pmax && M > 0.f) { const int i = atomicAdd(&s_nmax, 1); if (i < TO_TIE_CAP) s_maxrow[i] = s; }
            if (amin && qq.x == a) { const int i = atomicAdd(&s_nmin, 1); if (i < TO_TIE_CAP) s_minrow[i] = s; }
        }
        const unsigned long long b = __ballot(flag);
        if (lane == 0 && (s0 + (t & ~63)) < nslots) fv[(int64_t)v * fv_words + ((s0 + t) >> 6)] = b;
        if (b) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&s_npairs, __popcll(b));   // LDS: one reservation per wave
            base = __shfl(base, 0);
            bool first = false;
            if (flag) {
                myl[base + __popcll(b & ((1ull << lane) - 1ull))] = s;
                // the slot's first flag (of any waypoint) puts it on the list k_traj_lo_sparse walks: a zero word before this bit
                // is necessary, the exchange on the slot's own marker decides (other words, other blocks)
                // (the word may hold other trajectories' bits: a zero word is only the cheap way out for the common case)
                const unsigned long long old = atomicOr(&ft[(int64_t)s * vwords + (v >> 6)], 1ull << (v & 63));
                if ((old & seg_mask) == 0ull) first = atomicExch(&sflag[(int64_t)seg * nslots + s], 1) == 0;
            }
            const unsigned long long fb = __ballot(first);
            if (fb) {   // one counter update per wave
                int at = 0;
                if (lane == 0) at = atomicAdd(&sflag[nmark], __popcll(fb));
                at = __shfl(at, 0);
                if (first) {
                    const int e = at + __popcll(fb & ((1ull << lane) - 1ull));
                    slist[2 * e] = s; slist[2 * e + 1] = seg;
                }
            }
        }
    };
    if constexpr (FAST) {
#pragma unroll
        for (int kk = 0; kk < NQ; ++kk)
            if (kk * TO_SELECT_THREADS < nslots) sweep2(q[kk], kk * TO_SELECT_THREADS);
    } else {
        for (int s0 = 0; s0 < nslots; s0 += TO_SELECT_THREADS) sweep2(s0 + t < nslots ? pv[s0 + t] : make_float2(0.f, 0.f), s0);
    }
    __syncthreads();
    if (t == 0) {
        vcnt[v] = s_npairs;
        // ascending slot order: the tie sums are added in a fixed order (insertion sort of <= 7 entries, in LDS)
        const int na = min(s_nmax, TO_TIE_CAP), nb = min(s_nmin, TO_TIE_CAP);
        for (int i = 1; i < na; ++i) {
            const int x = s_maxrow[i];
            int j = i - 1;
            while (j >= 0 && s_maxrow[j] > x) { s_maxrow[j + 1] = s_maxrow[j]; --j; }
            s_maxrow[j + 1] = x;
        }
        for (int i = 1; i < nb; ++i) {
            const int x = s_minrow[i];
            int j = i - 1;
            while (j >= 0 && s_minrow[j] > x) { s_minrow[j + 1] = s_minrow[j]; --j; }
            s_minrow[j + 1] = x;
        }
        int* tp = reinterpret_cast<int*>(ties + v);   // TieRec: nmax, nmin, maxrow[7], minrow[7]
        tp[0] = s_nmax;
        tp[1] = s_nmin;
        for (int i = 0; i < TO_TIE_CAP; ++i) { tp[2 + i] = i < na ? s_maxrow[i] : 0; tp[2 + TO_TIE_CAP + i] = i < nb ? s_minrow[i] : 0; }
    }
}

// ---------------------------------------------------------------------------------------------
// sparse forward: block per flagged slot, 1024 threads = 256 points x 4 waypoint groups.  Group g takes the slot's flagged
// waypoints of rank g, g+4, ... (ascending); the four partial sums are added in group order: a fixed summation order that
// depends only on the flag set, which DENSE and CULL share.
//   p_hat = (p - a)/M, clip to [0.5, 1-eps], log-odds (model.py:226-231); unflagged pairs contribute exactly 0.

__device__ __forceinline__ float log_odds(const EvalK& k, const WayRec& r, float p) {
    float ph = (p - r.a) * r.invM;
    ph = __builtin_amdgcn_fmed3f(ph, 0.5f, k.clip_hi);
    // log(ph/(1-ph)) as a difference of logs: exactly 0 at ph = 0.5
    return (to_log2(ph) - to_log2(1.0f - ph)) * 0.693147180559945f;
}

__global__ void __launch_bounds__(1024)
k_traj_lo_sparse(CloudView cv, const WayRec* __restrict__ rec, EvalK k, const unsigned long long* __restrict__ ft, int vwords,
                 float* __restrict__ lo_sum, const uint32_t* __restrict__ occ, int64_t occw, const int* __restrict__ toff, int n_traj,
                 int C, const int* __restrict__ slist, const int* __restrict__ nlisted) {
    // blocks walk the list of flagged (slot, trajectory) pairs k_traj_select made (6 % of the slots on the BASELINE workloads; a block per SLOT
    // spent 9 of its 13 us on 3 700 blocks that read their flag words and left, two 1024-thread blocks to a CU).  The list's
    // order is arrival order; a slot's sum does not depend on it.  The sum is ADDED to what lo_sum holds: zero from
    // pass 1, or the NaN k_traj_select stored everywhere for a degenerate waypoint (the reference divides 0/0 for every point
    // then, model.py:227).  Several trajectories (toff: their n_traj + 1 body-waypoint offsets; C cameras each): each has its own vector and
    // its own rank count, so its sum is the one a run of that trajectory alone produces.
    __shared__ float spart[3][TO_SLOT];
    const int pt = threadIdx.x & (TO_SLOT - 1), g = threadIdx.x >> 8;
    const int nl = *nlisted;
    // a list entry = (slot, trajectory): that trajectory's waypoints [v_lo, v_hi) only, its own log-odds vector — the trajectories
    // of a slot run side by side, and nobody looks at a (slot, trajectory) pair without a flag
    for (int li = blockIdx.x; li < nl; li += gridDim.x) {   // block-uniform
        const int s = slist[2 * li], tr = slist[2 * li + 1];
        const int v_lo = toff ? toff[tr] * C : 0, v_hi = toff ? toff[tr + 1] * C : 0x7fffffff;
        const unsigned long long* fts = ft + (int64_t)s * vwords;
        const int64_t i = (int64_t)s * TO_SLOT + pt;
        const float x = cv.soa[i], y = cv.soa[cv.npad + i], z = cv.soa[2 * cv.npad + i];
        float acc = 0.f;
        int rank = 0;
        for (int w = v_lo >> 6; w < vwords && w * 64 < v_hi; ++w) {
            unsigned long long bits = fts[w];
            if (w * 64 < v_lo) bits &= ~0ull << (v_lo - w * 64);
            if (v_hi - w * 64 < 64) bits &= (1ull << (v_hi - w * 64)) - 1ull;
            while (bits) {
                const int v = w * 64 + __builtin_ctzll(bits);
                bits &= bits - 1ull;
                if (((rank++) & 3) == g) {
                    const WayRec& r = rec[v];
                    // a wave (64 neighbouring points) none of whose points can reach p_hat = 1/2 adds exact zeros
                    if (__any(!(dist2_sp(r, x, y, z) > r.thr1))) acc += log_odds(k, r, vis_p(r, k, x, y, z) * occ_one(occ, occw, v, i));
                }
            }
        }
        if (g) spart[g - 1][pt] = acc;
        __syncthreads();
        if (!g) {
            float* dst = lo_sum + (int64_t)tr * cv.npad + i;
            *dst = *dst + (((acc + spart[0][pt]) + spart[1][pt]) + spart[2][pt]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// rewards = sigmoid(lo_sum) (model.py:237) scattered back to the caller's point order, mean and
// visibility loss (model.py:246).

// f64 sum of one value per lane in a fixed order (xor butterfly); every lane gets the result
__device__ __forceinline__ double wave_sum_double(double v) {
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

// One launch, thread per PACKED position: lo_sum and the permutation are read coalesced, the reward goes to the caller's
// order with a scattered 4-byte store — which is skipped for the points whose log-odds is exactly 0 when the caller says the
// rewards vector already holds sigmoid(0) = 1/2 everywhere (`prefilled`: tohip_traj_forward's rewards_half output; 98 % of the
// points on the BASELINE workloads).
// The mean needs one more dependent step only: every block adds ONE 64-bit word to an accumulator — its f64 partial as a
// fixed-point integer (bits 0..47), a NaN mark (bits 48..55) and its arrival (bits 56..63).  Integer addition commutes, so the
// total does not depend on the arrival order; the block whose add returns the last arrival has the complete sum in hand
// (returned value + its own word) and writes the scalars.  The fixed-point step is 2^-shift with shift = 47 - ceil(log2 n): at
// most 128 roundings of 2^-(shift+1) on a sum of >= n/2 (1 M points: 5e-13 relative).  acc: zero before and after the launch.
#define TO_REWARD_THREADS 1024
#define TO_REWARD_BLOCKS 128
// block bx of nbx of trajectory `traj` (its own log-odds vector, rewards vector, accumulator word and scalars)
__device__ __forceinline__ void reward_block(const float* __restrict__ lo_sum, const int* __restrict__ perm, int64_t n, int64_t npad, float eps,
                                             int shift, int prefilled, float* __restrict__ rewards, unsigned long long* __restrict__ acc,
                                             float* __restrict__ scalars, int bx, int nbx, int traj, double* lds) {
    lo_sum += (int64_t)traj * npad;
    rewards += (int64_t)traj * n;
    acc += traj;
    scalars += 4 * traj;
    double s = 0.0;
    const int64_t stride = (int64_t)nbx * TO_REWARD_THREADS * 4;
    for (int64_t i0 = ((int64_t)bx * TO_REWARD_THREADS + threadIdx.x) * 4; i0 < n; i0 += stride) {
        const float4 lo4 = *reinterpret_cast<const float4*>(lo_sum + i0);   // npad is a multiple of 2048: aligned, in bounds
        const float lo[4] = {lo4.x, lo4.y, lo4.z, lo4.w};
        const bool all0 = (lo4.x == 0.f) & (lo4.y == 0.f) & (lo4.z == 0.f) & (lo4.w == 0.f);
        if (prefilled && all0) {
            // sigmoid(0) evaluated as below is exactly 0.5: rcp(1 + 1)
            const int cnt = (int)min((int64_t)4, n - i0);
            s += 0.5 * (double)cnt;
            continue;
        }
        const int4 o4 = *reinterpret_cast<const int4*>(perm + i0);
        const int o[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i0 + j < n) {
                float r = to_rcp(1.0f + to_exp(-lo[j]));
                if (lo[j] != lo[j]) r = lo[j];  // a degenerate waypoint (max == min) makes the reference's rewards NaN: propagate
                if (!prefilled || lo[j] != 0.f) rewards[o[j]] = r;
                s += (double)r;
            }
        }
    }
    const double tot = block_sum_double(s, lds);
    if (threadIdx.x != 0) return;
    const bool isnan_ = tot != tot;
    const unsigned long long fixed = isnan_ ? 0ull : (unsigned long long)__double2ll_rn(ldexp(tot, shift));
    const unsigned long long word = (1ull << 56) | (isnan_ ? (1ull << 48) : 0ull) | fixed;
    const unsigned long long old = __hip_atomic_fetch_add(acc, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(old >> 56) != nbx - 1) return;
    const unsigned long long all = old + word;
    __hip_atomic_store(acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    const bool anynan = ((all >> 48) & 0xffull) != 0ull;
    const double sum = ldexp((double)(all & 0xffffffffffffull), -shift);
    const float mean = anynan ? __builtin_nanf("") : (float)(sum / (double)n);
    const float vis = 1.0f / (mean + eps);
    scalars[0] = mean;
    scalars[1] = vis;
    scalars[2] = (float)(-(double)vis * (double)vis / (double)n);
    scalars[3] = 0.f;  // reserved; written so that callers need not clear the vector
}

__global__ void __launch_bounds__(TO_REWARD_THREADS)
k_traj_reward(const float* __restrict__ lo_sum, const int* __restrict__ perm, int64_t n, int64_t npad, float eps, int shift, int prefilled,
              float* __restrict__ rewards, unsigned long long* __restrict__ acc, float* __restrict__ scalars) {
    __shared__ double lds[TO_REWARD_THREADS / 64];
    reward_block(lo_sum, perm, n, npad, eps, shift, prefilled, rewards, acc, scalars, blockIdx.x, gridDim.x, blockIdx.y, lds);
}

// ---------------------------------------------------------------------------------------------
// backward.  Per (point, waypoint): G = dL/dp_hat = g_n [0.5 <= p_hat <= 1-eps] / (p_hat (1 - p_hat)),
// dL/dp = G / M, plus the shares of the min/max points (torch splits them evenly among ties):
//   S1 = sum G (p_hat - 1)/M  -> argmin set,   S2 = sum G (-p_hat)/M -> argmax set.
// Per flagged pair 14 sums, bpart[(v*nslots+slot)*16 ..]:
//   [0..2] sum w gy   [3..11] sum w y (x) gy   [12] S1   [13] S2        (w = G/M, gy = dp/dy, y = x - t)

#define TO_BWD_GX 8    // blocks per waypoint; wave (x, w) of a waypoint takes the flagged slots vlist[v][4x + w], [.. + 32], ...
// One WAVE per flagged pair, four points per lane (point lane + 64 j of the slot): the 14 sums are added per lane over its four
// points, then once across the wave (DPP tree, total in lane 63) — a quarter of the cross-lane work a 256-thread block with one
// point per thread spent, no LDS and no block barrier.  A fixed order that depends on nothing but the pair.
// wave wv of nwv of waypoint v.  scalars == NULL (and no grad_rewards): the sums are taken with dL/d reward = 1 and the finish kernel
// scales them (they are linear in it) — so that this work does not have to wait for the mean of the rewards.
__device__ __forceinline__ void bwd_pairs(const CloudView& cv, const WayRec* __restrict__ rec, const EvalK& k, const int* __restrict__ vlist,
                                          const int* __restrict__ vcnt, int nslots, const float* __restrict__ lo_sum,
                                          const float* __restrict__ grad_rewards, const float* __restrict__ scalars,
                                          const float* __restrict__ gout, float* __restrict__ bpart, const uint32_t* __restrict__ occ,
                                          int64_t occw, int v, int wv, int nwv) {
    const int lane = threadIdx.x & 63;
    const int npairs = vcnt[v];
    if (wv >= npairs) return;
    // the waypoint's trajectory: its log-odds vector, upstream gradient and loss scalars
    const int seg = rec[v].seg;
    lo_sum += (int64_t)seg * cv.npad;
    if (grad_rewards) grad_rewards += (int64_t)seg * cv.n;
    const float coef = grad_rewards ? 0.f : (scalars ? scalars[4 * seg + 2] * gout[seg] : 1.0f);
    const WayRec& r = rec[v];
    for (int it = wv; it < npairs; it += nwv) {
        const int s = vlist[(int64_t)v * nslots + it];
        float acc[TO_BWD_NSUM];
#pragma unroll
        for (int j = 0; j < TO_BWD_NSUM; ++j) acc[j] = 0.f;
        bool any = false;
#pragma unroll
        for (int q = 0; q < TO_SLOT / 64; ++q) {
            const int64_t i = (int64_t)s * TO_SLOT + q * 64 + lane;
            const float x = cv.soa[i], y = cv.soa[cv.npad + i], z = cv.soa[2 * cv.npad + i];
            const float lo_i = lo_sum[i];   // (npad floats per vector) requested with the coordinates, not after the distance test
            if (!__any(!(dist2_sp(r, x, y, z) > r.thr1))) continue;   // none of these 64 points can reach p_hat = 1/2 (k_traj_select)
            // dL/d lo_sum_n: through the caller's dL/d rewards vector (general criterion) or the fused visibility loss
            float gn = 0.f;
            if (i < cv.n) {
                const float lo = lo_i;
                float rw = to_rcp(1.0f + to_exp(-lo));  // == k_traj_reward's value of rewards[perm[i]]
                if (lo != lo) rw = lo;
                const float gr = grad_rewards ? grad_rewards[cv.perm[i]] : coef;
                gn = gr * rw * (1.0f - rw);
            }
            VisGrad vg;
            const float p = vis_p(r, k, x, y, z, &vg) * occ_one(occ, occw, v, i);
            const float ph = (p - r.a) * r.invM;
            const bool act = (ph >= 0.5f) && (ph <= k.clip_hi);
            if (act) {
                float g[3];
                dvis_dy(r, k, p, vg, g);
                const float G = gn * to_rcp(ph * (1.0f - ph));
                const float wgt = G * r.invM;
                acc[12] += wgt * (ph - 1.0f);
                acc[13] += -wgt * ph;
                const float w0 = wgt * g[0], w1 = wgt * g[1], w2 = wgt * g[2];
                acc[0] += w0; acc[1] += w1; acc[2] += w2;
                acc[3] += vg.y0 * w0; acc[4] += vg.y0 * w1; acc[5] += vg.y0 * w2;
                acc[6] += vg.y1 * w0; acc[7] += vg.y1 * w1; acc[8] += vg.y1 * w2;
                acc[9] += vg.y2 * w0; acc[10] += vg.y2 * w1; acc[11] += vg.y2 * w2;
                any = true;
            }
        }
        if (__any(any)) {
#pragma unroll
            for (int j = 0; j < TO_BWD_NSUM; ++j) acc[j] = wave_sum63(acc[j]);
        }
        if (lane == 63) {
            float* dst = bpart + ((int64_t)v * nslots + s) * 16;
#pragma unroll
            for (int j = 0; j < TO_BWD_NSUM; ++j) dst[j] = acc[j];
            dst[14] = 0.f; dst[15] = 0.f;   // the finish kernel adds all 16 columns of a row
        }
    }
}

__global__ void __launch_bounds__(TO_SLOT)
k_traj_bwd_sparse(CloudView cv, const WayRec* __restrict__ rec, EvalK k, const int* __restrict__ vlist, const int* __restrict__ vcnt,
                  int nslots, const float* __restrict__ lo_sum, const float* __restrict__ grad_rewards,
                  const float* __restrict__ scalars, const float* __restrict__ gout, float* __restrict__ bpart,
                  const uint32_t* __restrict__ occ, int64_t occw) {
    bwd_pairs(cv, rec, k, vlist, vcnt, nslots, lo_sum, grad_rewards, scalars, gout, bpart, occ, occw, blockIdx.y,
              blockIdx.x * (TO_SLOT / 64) + (threadIdx.x >> 6), gridDim.x * (TO_SLOT / 64));
}

// rewards + mean + loss (the first nbx * n_traj blocks) and the gradient sums with unit upstream gradient (the other blocks, 16
// waves each = half a waypoint's 32) in ONE launch: both need the complete log-odds vector and nothing of each other.
__global__ void __launch_bounds__(TO_REWARD_THREADS)
k_traj_reward_bwd(const float* __restrict__ lo_sum, int64_t n, float eps, int shift, int prefilled, float* __restrict__ rewards,
                  unsigned long long* __restrict__ acc, float* __restrict__ scalars, int nbx, int n_traj,
                  CloudView cv, const WayRec* __restrict__ rec, EvalK k, const int* __restrict__ vlist, const int* __restrict__ vcnt,
                  int nslots, float* __restrict__ bpart, const uint32_t* __restrict__ occ, int64_t occw, int V) {
    __shared__ double lds[TO_REWARD_THREADS / 64];
    const int R = nbx * n_traj;
    if ((int)blockIdx.x < R) {
        reward_block(lo_sum, cv.perm, n, cv.npad, eps, shift, prefilled, rewards, acc, scalars, blockIdx.x % nbx, nbx, blockIdx.x / nbx, lds);
        return;
    }
    const int64_t gw = ((int64_t)blockIdx.x - R) * (TO_REWARD_THREADS / 64) + (threadIdx.x >> 6);
    const int v = (int)(gw >> 5);
    if (v < V) bwd_pairs(cv, rec, k, vlist, vcnt, nslots, lo_sum, nullptr, nullptr, nullptr, bpart, occ, occw, v, (int)(gw & 31), 32);
}

// thread per body waypoint: rig composition, dL/dt = -R sum dL/dc, dL/dR = sum y (x) dL/dc,
// quaternion chain through the homogeneous form of R and through F.normalize.
// vgrad[v*12 ..] = (sum dL/dc [3], sum y (x) dL/dc [9]); mrow(v) = the 9 floats m[3*i+j] = R_v[j][i].
template <typename MRow>
__device__ void finish_waypoint(int w, const float* __restrict__ vgrad, MRow mrow, const WayCold* __restrict__ cold, int C,
                                const float* __restrict__ rig_q, const float* __restrict__ rig_t,
                                float* __restrict__ poses_grad, float* __restrict__ quats_grad) {
    double dt[3] = {0, 0, 0}, A[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // A[3*j+i] = dL/dR_w[j][i]
    for (int c = 0; c < C; ++c) {
        const int v = w * C + c;
        const float* gv = vgrad + (int64_t)v * 12;
        const float* m = mrow(v);
        // dL/dt_v = -R_v Gt,   R_v[j][i] = m[3*i+j]
        double dtv[3];
        for (int j = 0; j < 3; ++j)
            dtv[j] = -((double)m[j] * gv[0] + (double)m[3 + j] * gv[1] + (double)m[6 + j] * gv[2]);
        for (int j = 0; j < 3; ++j) dt[j] += dtv[j];
        if (rig_q != nullptr) {
            const float qc[4] = {rig_q[4 * c], rig_q[4 * c + 1], rig_q[4 * c + 2], rig_q[4 * c + 3]};
            float Rc[9];
            quat_to_R(qc, Rc);
            // dL/dR_w = dL/dR_v Rc^T + dL/dt_v l^T
            for (int j = 0; j < 3; ++j)
                for (int i = 0; i < 3; ++i) {
                    double a = 0;
                    for (int kk = 0; kk < 3; ++kk) a += (double)gv[3 + 3 * j + kk] * (double)Rc[3 * i + kk];
                    if (rig_t != nullptr) a += dtv[j] * (double)rig_t[3 * c + i];
                    A[3 * j + i] += a;
                }
        } else {
            for (int kk = 0; kk < 9; ++kk) A[kk] += (double)gv[3 + kk];
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

struct RecRows {
    const WayRec* rec;
    __device__ const float* operator()(int v) const { return rec[v].m; }
};
struct HotRows {
    const WayHot* hot;
    __device__ const float* operator()(int v) const { return hot[v].m; }
};

__global__ void k_traj_bwd_finish2(const float* __restrict__ vgrad, const WayRec* __restrict__ rec,
                                   const WayCold* __restrict__ cold, int W, int C, const float* __restrict__ rig_q,
                                   const float* __restrict__ rig_t, float* __restrict__ poses_grad,
                                   float* __restrict__ quats_grad) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < W) finish_waypoint(w, vgrad, RecRows{rec}, cold, C, rig_q, rig_t, poses_grad, quats_grad);
}

// the ModelPose path keeps its own record type (pose_kernels.hip)
__global__ void k_bwd_finish2(const float* __restrict__ vgrad, const WayHot* __restrict__ hot,
                              const WayCold* __restrict__ cold, int W, int C, const float* __restrict__ rig_q,
                              const float* __restrict__ rig_t, float* __restrict__ poses_grad,
                              float* __restrict__ quats_grad) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < W) finish_waypoint(w, vgrad, HotRows{hot}, cold, C, rig_q, rig_t, poses_grad, quats_grad);
}

// block per virtual waypoint: adds the partials of the flagged slots in 16 sums x 64 slot groups — group g takes the slots
// s = g (mod 64), i.e. bit g of every flag word, in ascending order; the 64 group sums are then added in group order (double).
// THREADS = 1024: a thread per (sum, group), the shortest chain (up to a few hundred waypoints).  THREADS = 256: a thread keeps
// four groups (g, g + 16, g + 32, g + 48) — the same 64 sums in the same order, bit for bit — and eight blocks share a CU where
// two 1024-thread blocks made the 1 024 waypoints of eight concurrent trajectories queue (32 -> 13 us) — and the shares of the extremal points: the rows recorded by k_traj_select are re-evaluated by
// wave 0 in ascending row order with a fixed DPP tree, so the result does not depend on any arrival order (torch splits the
// gradient of min()/max() evenly among ties, model.py:226-227).
//   vgrad[v*12 ..] = (sum dL/dc [3], sum y (x) dL/dc [9]) with c = R^T y:  R^T gy,  (y (x) gy) R.
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_traj_bwd_finish(CloudView cv, const float* __restrict__ bpart, int nslots, const unsigned long long* __restrict__ fv, int fv_words,
                  const WayRec* __restrict__ rec, EvalK k, const TieRec* __restrict__ ties, const float2* __restrict__ part,
                  const uint32_t* __restrict__ occ, int64_t occw, float* __restrict__ vgrad,
                  const WayCold* __restrict__ cold, int single, float* __restrict__ poses_grad, float* __restrict__ quats_grad,
                  const float* __restrict__ post_scalars, const float* __restrict__ post_gout) {
    constexpr int NG = THREADS / 16, PER = 64 / NG;   // groups per pass, groups per thread
    __shared__ double sgrp[64][16];
    __shared__ double stie[2][13];   // [0] argmin set, [1] argmax set: 12 sums + count
    const int v = blockIdx.x, t = threadIdx.x, kq = t & 15, g = t >> 4;
    const WayRec& r = rec[v];
    // the waypoint's flag words first (one parallel load), then each group's bit of every word
    __shared__ unsigned long long sfv[1024];
    double acc[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) acc[j] = 0.0;
    for (int w0 = 0; w0 < fv_words; w0 += 1024) {
        const int nw = min(1024, fv_words - w0);
        __syncthreads();
        for (int j = t; j < nw; j += THREADS) sfv[j] = fv[(int64_t)v * fv_words + w0 + j];
        __syncthreads();
        for (int w = 0; w < nw; ++w) {
            const unsigned long long word = sfv[w];
            if (word == 0ull) continue;
#pragma unroll
            for (int j = 0; j < PER; ++j)
                if ((word >> (g + NG * j)) & 1ull) acc[j] += (double)bpart[((int64_t)v * nslots + ((w0 + w) * 64 + g + NG * j)) * 16 + kq];
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) sgrp[g + NG * j][kq] = acc[j];
    // ---- argmin / argmax sets: waves 0..3 take a quarter (64 points) of every recorded slot each ----
    __shared__ double stie4[2][4][13];
    if (t < 256) {
        const int wq = t >> 6, ln = t & 63;
        const int* tp = reinterpret_cast<const int*>(ties + v);   // TieRec: nmax, nmin, maxrow[7], minrow[7]
        const float a = r.a, M = r.M;
        for (int set = 0; set < 2; ++set) {
            const int cnt = set ? tp[0] : tp[1];
            double tot[13];
            for (int j = 0; j < 13; ++j) tot[j] = 0.0;
            auto do_slot = [&](int rr) {   // rr: a slot holding the extremum
                const int64_t i = (int64_t)rr * TO_SLOT + wq * 64 + ln;
                float gq[13];
                for (int j = 0; j < 13; ++j) gq[j] = 0.f;
                if (i < cv.n) {  // pads are not members
                    VisGrad vg;
                    const float p = vis_p(r, k, cv.soa[i], cv.soa[cv.npad + i], cv.soa[2 * cv.npad + i], &vg) * occ_one(occ, occw, v, i);
                    const bool member = set ? ((p - a == M) && (M > 0.f)) : ((p == a) && (p > 0.f));
                    if (member) {
                        float gy[3];
                        dvis_dy(r, k, p, vg, gy);
                        const float yy[3] = {vg.y0, vg.y1, vg.y2};
                        for (int q = 0; q < 3; ++q) gq[q] = gy[q];
                        for (int j = 0; j < 3; ++j)
                            for (int q = 0; q < 3; ++q) gq[3 + 3 * j + q] = yy[j] * gy[q];
                        gq[12] = 1.0f;
                    }
                }
                for (int j = 0; j < 13; ++j) tot[j] += (double)wave_sum63(gq[j]);  // valid in lane 63
            };
            if (cnt <= TO_TIE_CAP) {
                for (int q = 0; q < cnt; ++q) do_slot(tp[2 + (set ? 0 : TO_TIE_CAP) + q]);
            } else {
                // more slots hold the extremum than were recorded: walk every slot partial (rare: many exact duplicates)
                for (int rr = 0; rr < nslots; ++rr) {
                    const float2 q = part[(int64_t)v * nslots + rr];
                    const bool hit = set ? (q.y - a == M) : (q.x == a);
                    if (hit) do_slot(rr);
                }
            }
            if (ln == 63)
                for (int j = 0; j < 13; ++j) stie4[set][wq][j] = tot[j];
        }
    }
    __syncthreads();
    if (t < 26) {   // ascending slot order inside a quarter, quarters in order: a fixed summation order
        const int set = t / 13, j = t % 13;
        stie[set][j] = ((stie4[set][0][j] + stie4[set][1][j]) + stie4[set][2][j]) + stie4[set][3][j];
    }
    __shared__ double stot[16];
    if (t < 16) {
        double q = 0.0;
        for (int gg = 0; gg < 64; ++gg) q += sgrp[gg][t];
        // sums taken with unit upstream gradient (k_traj_reward_bwd) get the trajectory's dL/d reward here: they are linear in it
        if (post_scalars != nullptr) q *= (double)(post_scalars[4 * r.seg + 2] * post_gout[r.seg]);
        stot[t] = q;
    }
    __syncthreads();
    __shared__ float sgy[12];
    if (t < 12) {
        const double nmin = stie[0][12], nmax = stie[1][12];
        const double wmin = nmin > 0.0 ? stot[12] / nmin : 0.0;
        const double wmax = nmax > 0.0 ? stot[13] / nmax : 0.0;
        sgy[t] = (float)(stot[t] + wmin * stie[0][t] + wmax * stie[1][t]);   // sum gy [3], sum y (x) gy [9], world-aligned
    }
    __syncthreads();
    if (t < 12) {
        // c = m y:  dL/dc = m gy;  y (x) dL/dc = (y (x) gy) m^T     (m[3*i+j] = R[j][i])
        float out;
        if (t < 3) out = (float)((double)r.m[3 * t] * sgy[0] + (double)r.m[3 * t + 1] * sgy[1] + (double)r.m[3 * t + 2] * sgy[2]);
        else {
            const int j = (t - 3) / 3, i = (t - 3) % 3;
            out = (float)((double)sgy[3 + 3 * j] * r.m[3 * i] + (double)sgy[3 + 3 * j + 1] * r.m[3 * i + 1] + (double)sgy[3 + 3 * j + 2] * r.m[3 * i + 2]);
        }
        vgrad[v * 12 + t] = out;
    }
    if (single) {  // one camera per waypoint: the waypoint's gradient follows at once (k_traj_bwd_finish2's work, no launch)
        __syncthreads();
        if (t == 0) finish_waypoint(v, vgrad, RecRows{rec}, cold, 1, nullptr, nullptr, poses_grad, quats_grad);
    }
}

// ---------------------------------------------------------------------------------------------
// occlusion bits: row = all ones, then every point kept by the hard frustum cull is cleared and
// every point HPR (or the z-buffer) found visible among the kept ones is set again; the pad bits (positions n..npad-1,
// copies of the last sorted point) then take the bit of position n-1, so that a pad never sees more than its original.

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

// thread per waypoint row: positions n..npad-1 (at most 1023 bits) := bit n-1
__global__ void k_occ_rows_pad(int64_t n, int64_t npad, uint32_t* __restrict__ rows, int64_t roww, int n_wps) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_wps || n >= npad) return;
    uint32_t* row = rows + (int64_t)w * roww;
    const uint32_t last = (row[(n - 1) >> 5] >> ((n - 1) & 31)) & 1u;
    for (int64_t i = n; i < npad;) {
        const int64_t wi = i >> 5;
        const int b0 = (int)(i & 31);
        const int nb = (int)min((int64_t)32 - b0, npad - i);
        const uint32_t mask = (nb == 32 ? 0xffffffffu : ((1u << nb) - 1u)) << b0;
        row[wi] = last ? (row[wi] | mask) : (row[wi] & ~mask);
        i += nb;
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
    const int32_t* kk = kept_idx + (int64_t)w * n;
    uint32_t* row = rows + (int64_t)w * roww;
    const int stride = gridDim.x * blockDim.x;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const int s = inv[kk[j]];
        atomicAnd(&row[s >> 5], ~(1u << (s & 31)));
    }
}

__global__ void k_occ_rows_set(const int* __restrict__ inv, const int32_t* __restrict__ kept_idx, int64_t n,
                               const int32_t* __restrict__ vis_idx, const int32_t* __restrict__ vis_off,
                               const int32_t* __restrict__ all_visible, uint32_t* __restrict__ rows, int64_t roww) {
    const int w = blockIdx.y;
    if (all_visible[w]) return;
    const int j0 = vis_off[w], j1 = vis_off[w + 1];
    const int32_t* kk = kept_idx + (int64_t)w * n;
    uint32_t* row = rows + (int64_t)w * roww;
    const int stride = gridDim.x * blockDim.x;
    for (int j = j0 + blockIdx.x * blockDim.x + threadIdx.x; j < j1; j += stride) {
        const int s = inv[kk[vis_idx[j]]];
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
    const int64_t npad = tohip_padded_points(n);
    const int64_t roww = npad / 32;
    hipError_t e = hipMemsetAsync(rows, 0xff, (size_t)roww * (size_t)n_wps * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    int nb = (int)((n + 255) / 256);
    if (nb > 256) nb = 256;
    const dim3 grid((unsigned)nb, (unsigned)n_wps);
    k_occ_rows_clear<<<grid, 256, 0, st>>>(inv_perm, kept_idx, n, kept_count, all_visible, rows, roww);
    TO_HIP_CHECK_LAUNCH();
    k_occ_rows_set<<<grid, 256, 0, st>>>(inv_perm, kept_idx, n, vis_idx, vis_off, all_visible, rows, roww);
    TO_HIP_CHECK_LAUNCH();
    k_occ_rows_pad<<<(int)((n_wps + 63) / 64), 64, 0, st>>>(n, npad, rows, roww, (int)n_wps);
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
    k_occ_rows_pad<<<1, 64, 0, st>>>(n, npad, row, npad / 32, 1);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// host side: workspace layout + launches

namespace {

// diagnostic: device buffer for k_traj_pass1's per-block clock stamps (tohip_profile_clock); normally none
inline unsigned long long*& clock_stamps() { static unsigned long long* p = nullptr; return p; }

struct TrajPlan {
    int64_t npad;
    int nblk;      // pass-1 point blocks (1024 points each)
    int nslots;    // npad / 256
    int fv_words;  // (nslots + 63) / 64
    int vwords;    // (V + 63) / 64
    int V;
    size_t off_ctl, off_rec, off_cold, off_part, off_fv, off_ft, off_sflag, off_slist, off_vcnt, off_vlist, off_ties, off_bpart, off_vgrad, total;
    int64_t nmark;           // (trajectory, slot) markers = capacity of the pair list
    int64_t ft_zero_words;   // ft and, right behind it, one int per slot ("listed") + the slot list's counter: cleared together
};

inline TrajPlan make_plan(int64_t n, int64_t V, int64_t W, int64_t n_traj = 1) {
    TrajPlan p;
    p.npad = tohip_padded_points(n);
    p.nblk = (int)(p.npad / (TO_BLOCK * TO_P));
    p.nslots = (int)(p.npad / TO_SLOT);
    p.fv_words = (p.nslots + 63) / 64;
    p.vwords = (int)((V + 63) / 64);
    p.V = (int)V;
    size_t o = 0;
    p.off_ctl = o;   o += align_up(sizeof(unsigned long long) * (size_t)(n_traj < 1 ? 1 : n_traj), 256);   // first: all tohip_traj_reward uses
    p.off_rec = o;   o += align_up((size_t)V * sizeof(WayRec), 256);
    p.off_cold = o;  o += align_up((size_t)W * sizeof(WayCold), 256);
    p.off_part = o;  o += align_up((size_t)V * (size_t)p.nslots * sizeof(float2), 256);
    p.off_fv = o;    o += align_up((size_t)V * (size_t)p.fv_words * sizeof(unsigned long long), 256);
    p.off_ft = o;    o += (size_t)p.nslots * (size_t)p.vwords * sizeof(unsigned long long);
    const size_t nmark = (size_t)p.nslots * (size_t)(n_traj < 1 ? 1 : n_traj);   // one marker per (trajectory, slot)
    p.nmark = (int64_t)nmark;
    p.off_sflag = o; o += align_up((nmark + 2) * sizeof(int), 256);   // [nmark] listed?  [nmark] = entries of slist
    p.ft_zero_words = (int64_t)((o - p.off_ft) / sizeof(unsigned long long));
    p.off_slist = o; o += align_up(nmark * 2 * sizeof(int), 256);     // (slot, trajectory) pairs
    p.off_vcnt = o;  o += align_up((size_t)V * sizeof(int), 256);
    p.off_vlist = o; o += align_up((size_t)V * (size_t)p.nslots * sizeof(int), 256);
    p.off_ties = o;  o += align_up((size_t)V * sizeof(TieRec), 256);
    p.off_bpart = o; o += align_up((size_t)V * (size_t)p.nslots * 16 * sizeof(float), 256);
    p.off_vgrad = o; o += align_up((size_t)V * 12 * sizeof(float), 256);
    p.total = o;
    return p;
}

// culled pass 1: block rows of at most this many waypoints (one ballot; a wave walks its live waypoints one after the other)
inline void cull_tiles(int V, int* vtile, int* ntiles) {
    static const int cvt = [] { const char* e = getenv("TOHIP_CULL_VTILE"); return e ? atoi(e) : 0; }();  // experiments
    int vt = cvt > 0 ? cvt : 32;
    if (vt > 64) vt = 64;
    if (vt > V) vt = V;
    if (vt < 1) vt = 1;
    *vtile = vt;
    *ntiles = (V + vt - 1) / vt;
}

// dense pass 1: as many persistent blocks as are resident at once (never more than one per (point block, waypoint) pair)
inline int dense_blocks(int nblk, int V, bool occ) {
    static const int forced = [] { const char* e = getenv("TOHIP_DENSE_BLOCKS"); return e ? atoi(e) : 0; }();  // experiments
    static int per_cu[2] = {0, 0}, cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
        int a = 0, b = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, k_traj_pass1_dense<false>, TO_BLOCK, 0) != hipSuccess || a <= 0) a = 4;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, k_traj_pass1_dense<true>, TO_BLOCK, 0) != hipSuccess || b <= 0) b = 4;
        // the API answers one block per CU too many at this kernel's SGPR count (82-96; measured: 6 resident where it says 7,
        // MI355X_MICROARCH.md "Residency"); a block that is not resident from the start would run alone at the end
        per_cu[0] = (a > 8 ? 8 : a) - 1;
        per_cu[1] = (b > 8 ? 8 : b) - 1;
        if (per_cu[0] < 1) per_cu[0] = 1;
        if (per_cu[1] < 1) per_cu[1] = 1;
    }
    int64_t nb = forced > 0 ? forced : (int64_t)per_cu[occ ? 1 : 0] * cus;
    const int64_t total = (int64_t)nblk * V;
    if (nb > total) nb = total;
    return (int)nb;
}

inline int rig_cams(const tohip_rig* rig) { return (rig && rig->n_cams > 0 && rig->rig_quats) ? rig->n_cams : 1; }

}  // namespace

extern "C" size_t tohip_traj_workspace_bytes_multi(int64_t n_points, int64_t n_virtual, int64_t n_traj) {
    if (n_points <= 0 || n_virtual <= 0 || n_traj <= 0) return 0;
    return make_plan(n_points, n_virtual, n_virtual, n_traj).total;
}
extern "C" size_t tohip_traj_workspace_bytes(int64_t n_points, int64_t n_virtual) {
    return tohip_traj_workspace_bytes_multi(n_points, n_virtual, 1);
}

extern "C" int tohip_traj_forward_multi(const void* packed, int64_t n, const float* poses, const float* quats, int64_t W,
                                        const int32_t* traj_offsets, int64_t n_traj, const tohip_camera* cam, const tohip_rig* rig,
                                        int flags, const uint32_t* occlusion_bits, float* lo_sum, float* minmax, float* rewards_half,
                                        void* workspace, size_t workspace_bytes, void* stream_) {
    if (!packed || !poses || !quats || !cam || !lo_sum || !minmax || !workspace || n <= 0 || W <= 0 || n_traj <= 0 || n_traj > 65535 ||
        (n_traj > 1 && !traj_offsets))
        return TOHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    const int C = rig_cams(rig);
    const int64_t V = W * C;
    if (V > (1 << 24)) return TOHIP_EINVAL;
    const TrajPlan pl = make_plan(n, V, W, n_traj);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    char* ws = (char*)workspace;
    WayRec* rec = (WayRec*)(ws + pl.off_rec);
    WayCold* cold = (WayCold*)(ws + pl.off_cold);
    float2* part = (float2*)(ws + pl.off_part);
    unsigned long long* fv = (unsigned long long*)(ws + pl.off_fv);
    unsigned long long* ft = (unsigned long long*)(ws + pl.off_ft);
    int* vcnt = (int*)(ws + pl.off_vcnt);
    int* vlist = (int*)(ws + pl.off_vlist);
    TieRec* ties = (TieRec*)(ws + pl.off_ties);
    const EvalK k = make_evalk(cam);
    const CloudView cv = cloud_view(packed, n);
    const bool cull = !(flags & TOHIP_TRAJ_DENSE);
    const float* rq = (C > 1 || (rig && rig->rig_quats)) ? rig->rig_quats : nullptr;
    const float* rt = rq ? rig->rig_trans : nullptr;
    const int64_t ft_words = pl.ft_zero_words;   // the flag words and the slot list's markers + counter behind them
    int* sflag = (int*)(ws + pl.off_sflag);
    int* slist = (int*)(ws + pl.off_slist);
    const int64_t occw = cv.npad / 32;
    const int* toff = n_traj > 1 ? traj_offsets : nullptr;
    const OutInit oi{lo_sum, rewards_half, cv.npad, n, (int)n_traj};

    {
        TO_PROF(TOHIP_PROF_SMALL, st);
        if (cull) {
            if (V <= 512)
                k_traj_probe<1024><<<(int)V, 1024, 0, st>>>(cv, poses, quats, C, rq, rt, k, rec, cold, occlusion_bits, occw, ft, ft_words, toff,
                                                            (int)n_traj);
            else
                k_traj_probe<256><<<(int)V, 256, 0, st>>>(cv, poses, quats, C, rq, rt, k, rec, cold, occlusion_bits, occw, ft, ft_words, toff,
                                                          (int)n_traj);
        } else {
            int nb = (int)((V + 255) / 256);
            const int want = (int)((ft_words + 255) / 256);
            if (nb < want) nb = want > 256 ? 256 : want;
            k_traj_prep<<<nb, 256, 0, st>>>(poses, quats, (int)V, C, rq, rt, k, rec, cold, ft, ft_words, toff, (int)n_traj);
        }
        TO_HIP_CHECK_LAUNCH();
    }
    {
        TO_PROF(TOHIP_PROF_PASS1, st);
        const bool occ = occlusion_bits != nullptr;
        if (cull) {
            int vtile, ntiles;
            cull_tiles((int)V, &vtile, &ntiles);
            const dim3 grid(pl.nblk, ntiles);
            if (occ) k_traj_pass1_cull<true><<<grid, TO_BLOCK, 0, st>>>(cv, rec, (int)V, vtile, k, part, pl.nslots, occlusion_bits, occw, oi);
            else k_traj_pass1_cull<false><<<grid, TO_BLOCK, 0, st>>>(cv, rec, (int)V, vtile, k, part, pl.nslots, occlusion_bits, occw, oi);
        } else {
            const int nblk8 = (int)(pl.npad / (TO_BLOCK * TO_PD));
            const int nb = dense_blocks(nblk8, (int)V, occ);
            if (occ) k_traj_pass1_dense<true><<<nb, TO_BLOCK, 0, st>>>(cv, rec, (int)V, nblk8, k, part, pl.nslots, occlusion_bits, occw, oi, clock_stamps());
            else k_traj_pass1_dense<false><<<nb, TO_BLOCK, 0, st>>>(cv, rec, (int)V, nblk8, k, part, pl.nslots, occlusion_bits, occw, oi, clock_stamps());
        }
        TO_HIP_CHECK_LAUNCH();
    }
    {
        TO_PROF(TOHIP_PROF_SMALL, st);
        const bool fast = pl.nslots <= TO_SELECT_FAST_SLOTS;
        if (V <= 512) {
            if (fast) k_traj_select<true, 1024><<<(int)V, 1024, 0, st>>>(part, pl.nslots, (int)V, rec, cull ? 1 : 0, minmax, fv, pl.fv_words, ft,
                                                                      pl.vwords, vlist, vcnt, ties, lo_sum, cv.npad, sflag, slist, pl.nmark, toff, C, k.inv_var);
            else k_traj_select<false, 1024><<<(int)V, 1024, 0, st>>>(part, pl.nslots, (int)V, rec, cull ? 1 : 0, minmax, fv, pl.fv_words, ft,
                                                                      pl.vwords, vlist, vcnt, ties, lo_sum, cv.npad, sflag, slist, pl.nmark, toff, C, k.inv_var);
        } else {
            if (fast) k_traj_select<true, 256><<<(int)V, 256, 0, st>>>(part, pl.nslots, (int)V, rec, cull ? 1 : 0, minmax, fv, pl.fv_words, ft,
                                                                      pl.vwords, vlist, vcnt, ties, lo_sum, cv.npad, sflag, slist, pl.nmark, toff, C, k.inv_var);
            else k_traj_select<false, 256><<<(int)V, 256, 0, st>>>(part, pl.nslots, (int)V, rec, cull ? 1 : 0, minmax, fv, pl.fv_words, ft,
                                                                      pl.vwords, vlist, vcnt, ties, lo_sum, cv.npad, sflag, slist, pl.nmark, toff, C, k.inv_var);
        }
        TO_HIP_CHECK_LAUNCH();
    }
    {
        TO_PROF(TOHIP_PROF_PASS2, st);
        const int64_t lob = std::min<int64_t>(pl.nmark, 512 * n_traj);   // blocks: the expected number of listed pairs, each a chain of its own
        k_traj_lo_sparse<<<(int)std::min<int64_t>(lob, 4096), 1024, 0, st>>>(cv, rec, k, ft, pl.vwords, lo_sum, occlusion_bits, occw, toff, (int)n_traj, C,
                                                                           slist, sflag + pl.nmark);
        TO_HIP_CHECK_LAUNCH();
    }
    return TOHIP_OK;
}

extern "C" int tohip_traj_forward(const void* packed, int64_t n, const float* poses, const float* quats, int64_t W,
                                  const tohip_camera* cam, const tohip_rig* rig, int flags, const uint32_t* occlusion_bits,
                                  float* lo_sum, float* minmax, float* rewards_half, void* workspace, size_t workspace_bytes,
                                  void* stream_) {
    return tohip_traj_forward_multi(packed, n, poses, quats, W, nullptr, 1, cam, rig, flags, occlusion_bits, lo_sum, minmax, rewards_half,
                                    workspace, workspace_bytes, stream_);
}

extern "C" int tohip_traj_reward_multi(const void* packed, const float* lo_sum, int64_t n, int64_t n_traj, float eps, int prefilled,
                                       float* rewards, float* scalars, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!packed || !lo_sum || !rewards || !scalars || !workspace || n <= 0 || n_traj <= 0 || n_traj > 65535) return TOHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    if (workspace_bytes < sizeof(unsigned long long) * (size_t)n_traj) return TOHIP_ENOSPC;
    unsigned long long* acc = (unsigned long long*)workspace;   // TrajPlan::off_ctl == 0
    const CloudView cv = cloud_view(packed, n);
    int nb = (int)((n + 4 * TO_REWARD_THREADS - 1) / (4 * TO_REWARD_THREADS));
    if (nb > TO_REWARD_BLOCKS) nb = TO_REWARD_BLOCKS;
    int lg = 0;
    while (((int64_t)1 << lg) < n) ++lg;
    TO_PROF(TOHIP_PROF_REWARD, st);
    k_traj_reward<<<dim3(nb, (unsigned)n_traj), TO_REWARD_THREADS, 0, st>>>(lo_sum, cv.perm, n, cv.npad, eps, 47 - lg, prefilled ? 1 : 0, rewards,
                                                                            acc, scalars);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_traj_reward(const void* packed, const float* lo_sum, int64_t n, float eps, int prefilled, float* rewards,
                                 float* scalars, void* workspace, size_t workspace_bytes, void* stream_) {
    return tohip_traj_reward_multi(packed, lo_sum, n, 1, eps, prefilled, rewards, scalars, workspace, workspace_bytes, stream_);
}

namespace {
// backward of the step; with `fused` also the rewards, their mean and the loss scalars (tohip_traj_reward's work) in the
// backward's first launch
struct FusedReward { float eps; int prefilled; float* rewards; float* scalars; };
int traj_backward_impl(const void* packed, int64_t n, int64_t W, int64_t n_traj, const tohip_camera* cam,
                       const tohip_rig* rig, int flags, const uint32_t* occlusion_bits, const float* lo_sum,
                       const float* grad_rewards, const float* scalars, const float* gout, float* poses_grad,
                       float* quats_grad, void* workspace, size_t workspace_bytes, void* stream_, const FusedReward* fused,
                       int phases = 3 /* 1: the pair sums (with `fused`: + rewards, mean, loss scalars); 2: the finish kernels */) {
    if (!packed || !cam || !lo_sum || ((phases & 2) && (!poses_grad || !quats_grad)) || !workspace || n <= 0 || W <= 0 || n_traj <= 0 ||
        (!grad_rewards && (!scalars || ((phases & 2) && !gout))) || n_traj > 65535)
        return TOHIP_EINVAL;
    const float* fused_scalars = fused ? scalars : nullptr;
    hipStream_t st = (hipStream_t)stream_;
    const int C = rig_cams(rig);
    const int64_t V = W * C;
    const TrajPlan pl = make_plan(n, V, W, n_traj);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    char* ws = (char*)workspace;
    WayRec* rec = (WayRec*)(ws + pl.off_rec);
    WayCold* cold = (WayCold*)(ws + pl.off_cold);
    float2* part = (float2*)(ws + pl.off_part);
    unsigned long long* fv = (unsigned long long*)(ws + pl.off_fv);
    const int* vcnt = (const int*)(ws + pl.off_vcnt);
    const int* vlist = (const int*)(ws + pl.off_vlist);
    TieRec* ties = (TieRec*)(ws + pl.off_ties);
    float* bpart = (float*)(ws + pl.off_bpart);
    float* vgrad = (float*)(ws + pl.off_vgrad);
    const EvalK k = make_evalk(cam);
    const CloudView cv = cloud_view(packed, n);
    const float* rq = (C > 1 || (rig && rig->rig_quats)) ? rig->rig_quats : nullptr;
    const float* rt = rq ? rig->rig_trans : nullptr;
    const int64_t occw = cv.npad / 32;
    (void)flags;
    if (!(phases & 1)) {
    } else if (fused) {
        TO_PROF(TOHIP_PROF_BWD, st);
        int nbx = (int)((n + 4 * TO_REWARD_THREADS - 1) / (4 * TO_REWARD_THREADS));
        if (nbx > TO_REWARD_BLOCKS) nbx = TO_REWARD_BLOCKS;
        int lg = 0;
        while (((int64_t)1 << lg) < n) ++lg;
        const int64_t blocks = (int64_t)nbx * n_traj + (V * 32 + TO_REWARD_THREADS / 64 - 1) / (TO_REWARD_THREADS / 64);
        if (blocks > 0x7fffffff) return TOHIP_EINVAL;
        k_traj_reward_bwd<<<(int)blocks, TO_REWARD_THREADS, 0, st>>>(lo_sum, n, fused->eps, 47 - lg, fused->prefilled ? 1 : 0, fused->rewards,
                                                                     (unsigned long long*)workspace, fused->scalars, nbx, (int)n_traj, cv, rec, k,
                                                                     vlist, vcnt, pl.nslots, bpart, occlusion_bits, occw, (int)V);
        TO_HIP_CHECK_LAUNCH();
    } else {
        TO_PROF(TOHIP_PROF_BWD, st);
        for (int64_t v0 = 0; v0 < V; v0 += 65535) {  // grid.y limit
            const int nv = (int)(V - v0 < 65535 ? V - v0 : 65535);
            k_traj_bwd_sparse<<<dim3(TO_BWD_GX, nv), TO_SLOT, 0, st>>>(cv, rec + v0, k, vlist + v0 * pl.nslots, vcnt + v0, pl.nslots, lo_sum,
                                                                        grad_rewards, scalars, gout, bpart + v0 * pl.nslots * 16,
                                                                        occlusion_bits ? occlusion_bits + v0 * occw : nullptr, occw);
        }
        TO_HIP_CHECK_LAUNCH();
    }
    if (!(phases & 2)) return TOHIP_OK;
    TO_PROF(TOHIP_PROF_SMALL, st);
    const bool single = C == 1 && rq == nullptr;
    if (V <= 512)
        k_traj_bwd_finish<1024><<<(int)V, 1024, 0, st>>>(cv, bpart, pl.nslots, fv, pl.fv_words, rec, k, ties, part, occlusion_bits, occw, vgrad,
                                                         cold, single ? 1 : 0, poses_grad, quats_grad, fused_scalars, fused_scalars ? gout : nullptr);
    else
        k_traj_bwd_finish<256><<<(int)V, 256, 0, st>>>(cv, bpart, pl.nslots, fv, pl.fv_words, rec, k, ties, part, occlusion_bits, occw, vgrad,
                                                       cold, single ? 1 : 0, poses_grad, quats_grad, fused_scalars, fused_scalars ? gout : nullptr);
    TO_HIP_CHECK_LAUNCH();
    if (!single) {
        k_traj_bwd_finish2<<<(int)((W + 63) / 64), 64, 0, st>>>(vgrad, rec, cold, (int)W, C, rq, rt, poses_grad, quats_grad);
        TO_HIP_CHECK_LAUNCH();
    }
    return TOHIP_OK;
}

}  // namespace

extern "C" int tohip_traj_backward_multi(const void* packed, int64_t n, int64_t W, int64_t n_traj, const tohip_camera* cam,
                                         const tohip_rig* rig, int flags, const uint32_t* occlusion_bits, const float* lo_sum,
                                         const float* grad_rewards, const float* scalars, const float* gout, float* poses_grad,
                                         float* quats_grad, void* workspace, size_t workspace_bytes, void* stream_) {
    return traj_backward_impl(packed, n, W, n_traj, cam, rig, flags, occlusion_bits, lo_sum, grad_rewards, scalars, gout, poses_grad, quats_grad,
                              workspace, workspace_bytes, stream_, nullptr);
}

extern "C" int tohip_traj_reward_backward_multi(const void* packed, int64_t n, int64_t W, int64_t n_traj, const tohip_camera* cam,
                                                const tohip_rig* rig, int flags, const uint32_t* occlusion_bits, const float* lo_sum,
                                                float eps, int prefilled, float* rewards, float* scalars, const float* gout,
                                                float* poses_grad, float* quats_grad, void* workspace, size_t workspace_bytes,
                                                void* stream_) {
    if (!rewards || !scalars || !gout) return TOHIP_EINVAL;
    const FusedReward f{eps, prefilled, rewards, scalars};
    return traj_backward_impl(packed, n, W, n_traj, cam, rig, flags, occlusion_bits, lo_sum, nullptr, scalars, gout, poses_grad, quats_grad,
                              workspace, workspace_bytes, stream_, &f);
}

extern "C" int tohip_traj_reward_backward(const void* packed, int64_t n, int64_t W, const tohip_camera* cam, const tohip_rig* rig, int flags,
                                          const uint32_t* occlusion_bits, const float* lo_sum, float eps, int prefilled, float* rewards,
                                          float* scalars, const float* gout, float* poses_grad, float* quats_grad, void* workspace,
                                          size_t workspace_bytes, void* stream_) {
    return tohip_traj_reward_backward_multi(packed, n, W, 1, cam, rig, flags, occlusion_bits, lo_sum, eps, prefilled, rewards, scalars, gout,
                                            poses_grad, quats_grad, workspace, workspace_bytes, stream_);
}

extern "C" int tohip_traj_backward(const void* packed, int64_t n, int64_t W, const tohip_camera* cam, const tohip_rig* rig,
                                   int flags, const uint32_t* occlusion_bits, const float* lo_sum, const float* grad_rewards,
                                   const float* scalars, const float* gout, float* poses_grad, float* quats_grad,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    return tohip_traj_backward_multi(packed, n, W, 1, cam, rig, flags, occlusion_bits, lo_sum, grad_rewards, scalars, gout, poses_grad,
                                     quats_grad, workspace, workspace_bytes, stream_);
}

// Diagnostic (bench.py's roofline leg, never on in a timed pass): k_traj_pass1 stamps s_memtime / s_memrealtime per block into
// `buffer` (16 bytes per block; capacity for grid.x * grid.y blocks: tohip_profile_clock_blocks) while it is set; NULL turns it off.
// The in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
extern "C" int tohip_profile_clock(void* buffer) {
    clock_stamps() = (unsigned long long*)buffer;
    return TOHIP_OK;
}
extern "C" int64_t tohip_profile_clock_blocks(int64_t n_points, int64_t n_virtual, int flags) {
    if (n_points <= 0 || n_virtual <= 0 || !(flags & TOHIP_TRAJ_DENSE)) return 0;   // only the dense kernel stamps
    const TrajPlan pl = make_plan(n_points, n_virtual, n_virtual);
    const int nblk8 = (int)(pl.npad / (TO_BLOCK * TO_PD));
    return dense_blocks(nblk8, (int)n_virtual, true) > dense_blocks(nblk8, (int)n_virtual, false)
               ? dense_blocks(nblk8, (int)n_virtual, true) : dense_blocks(nblk8, (int)n_virtual, false);
}
