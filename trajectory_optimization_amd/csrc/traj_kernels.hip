// traj_kernels.hip — ModelTraj visibility term, forward + analytic backward, for gfx950.
//
// Replaces the per-waypoint Python loop of /root/reference/src/model.py:217-231 (forward),
// :237/:246 (rewards, visibility loss) and its torch-autograd backward (SURVEY.md §8a rows A-C,E,G).
//
// Every (point, waypoint) pair is evaluated ONCE, in pass 1, which keeps the extrema only; everything after it works on
// the few pairs that can contribute.  A step is FIVE launches:
//
//   k_traj_probe    block per waypoint: its record (WayRec, common.hpp), the extrema of p over a sample of the cloud
//                   (attained values: L <= max p, U >= min p), the reset of what the step accumulates into (running extrema,
//                   flag rows, tie lists, candidate bits, the pair list's length); CULL: the slots the waypoint can reach
//   k_traj_pass1    p of every pair (DENSE) / of the pairs the probe marked reachable (CULL) -> (min, max) per (256-point
//                   slot, waypoint) in `part`, the waypoint's running extrema by integer atomicMin / atomicMax (order
//                   independent), and a candidate bit per (trajectory, slot) whose maximum can be flagged
//   k_traj_sparse   block per candidate (slot, trajectory): which waypoints are FLAGGED for it — slot maximum reaches
//                   p_hat >= 1/2, or it holds an argmin point while min p > 0: only flagged pairs have a non-zero log-odds term
//                   or a gradient — then the log-odds of its 256 points over the flagged waypoints (fixed order); FUSED (no
//                   collective between forward and backward) also the rewards and their fixed-point sum; FWD stops at lo_sum
//                   (the all-reduce of a waypoint-sharded run comes next).  Appends its flagged pairs to the step's pair list.
//   k_traj_pairs    wave per flagged (slot, waypoint) pair, dealt evenly to the whole chip: the 14 gradient sums of the pair.
//                   Packed f32 throughout (two points per register pair).  In a split step it shares its launch with the
//                   rewards' scan of the complete lo_sum (k_traj_reward_bwd)
//   k_traj_finish   block per waypoint: its flagged slots' partials in slot order (f64), argmin/argmax shares from the
//                   recorded slots (deterministic: no float atomics), scale by dL/d reward, chain to (position, quaternion)
//
// About 0.7 % of the (slot, waypoint) pairs are flagged on the BASELINE workloads (1 M x 128: 3 546 of 500 224); a dense
// indoor cloud flags 10-20 %.  Pass 1 moves ~N*16 B per launch whatever W is and is bound by VALU issue (26 FMA-class + 4
// transcendental instructions per evaluation, common.hpp); the sparse kernel costs in proportion to the flagged pairs.
//
// Data layout in HBM
//   cloud     Morton-sorted SoA x|y|z (npad each) + permutation + one bounding sphere per 256 points
//             (packed once, tohip_pack_cloud)                                          16 B/point
//   WayRec    two 64-B lines per virtual waypoint; line 0 -> SGPRs by one s_load_dwordx16
//   Extrema   16 B per virtual waypoint: (min p, max p) as integers
//   lo_sum (sorted order) / rewards (original order)                                    4 B/point each
//   part      [slot][virtual waypoint]: (min, max) of p over the slot's 256 points         8 B
//   fv        flag bits [waypoint][slot word] (k_traj_finish walks its row); plist: the step's flagged (slot, waypoint) pairs
//   cbits     candidate bits [trajectory][slot word], one word per 128-byte line; live (CULL): reachable-slot bits [waypoint][slot word]
//   bpart     [virtual waypoint][slot]: 14 gradient sums of a flagged pair                 64 B (written where flagged)
//
// Two evaluation modes with bitwise identical results:
//   DENSE  pass 1 evaluates every (point, waypoint) pair (the streaming reference semantics; bench headline)
//   CULL   pass 1 skips pairs that provably can neither be a waypoint's maximum nor be flagged: p <= 2^(-cd d2)
//          bounds p by the squared distance d2 = |y - sp|^2; the probe tests the waypoint's sphere against every slot's
//          bounding sphere.  The bound is L/2; waypoints whose sample did not exhibit p == 0 (U > 0: the minimum is not
//          known to be zero) are searched densely.
//
// The forward leaves its state (records, extrema, flags) in the workspace; the backward reads it there: the workspace
// must not be touched between tohip_traj_forward and tohip_traj_backward of the same step.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "opt_step.hpp"
#include "profile.hpp"

#define TO_SLOT 256            // points per flag slot (= bounding-sphere tile)
#define TO_TIE_CAP 7           // recorded slots per extremum and waypoint; more -> the finish kernel scans all slots
#define TO_BWD_NSUM 14
#define TO_SP_THREADS 1024     // the block of the kernels that walk slot lists (culled pass 1, sparse, pairs)
#define TO_SP_WAVES (TO_SP_THREADS / 64)
#define TO_CBIT_STRIDE 16      // words between two words of the candidate bits: one to a 128-byte line (mark_candidate)
#define TO_PROBE_MAXFW 1024   // words of a `live` row the culled pass 1 holds in LDS: culling up to 65 536 slots (16.7 M points), dense beyond
#define TO_SP_MAXW 1024        // flag words of one slot held in LDS by k_traj_sparse: at most 65 536 virtual waypoints
// The step's list of flagged (slot, waypoint) pairs is 64 sub-lists, slot s appending to sub-list s & 63, each with its own length
// word on its own 128-byte line: a returning atomic on ONE word is served ~88 times per microsecond (MI355X_MICROARCH.md,
// "dequeue"), and a trajectory that has moved for a hundred optimiser steps has 1 400 candidate slots to append — 16 us of a 24 us
// kernel with one word.  Sub-list l holds the entries [l * cap, l * cap + length l), cap = ceil(nslots / 64) * V.
#define TO_PL_SHARDS 64
#define TO_PL_STRIDE 32        // ints between two length words

// ---------------------------------------------------------------------------------------------
// Diagnostic build (-DTOHIP_STAMPS, tools/kernel_timeline.py; never the shipped library): thread 0 of every block notes the
// 100 MHz real-time counter at a few places of the small kernels — where a 7-12 us kernel of dependent memory accesses spends its
// time cannot be read off a profile (k_traj_finish's 4.3 us walk over its flag words was found this way).
#ifdef TOHIP_STAMPS
#define TO_STAMP_BLOCKS 1024
#define TO_STAMP_N 12
enum { TO_STAMP_PROBE = 0, TO_STAMP_CULL, TO_STAMP_SPARSE, TO_STAMP_PAIRS, TO_STAMP_FINISH, TO_STAMP_KERNELS };
__device__ unsigned long long g_stamps[TO_STAMP_KERNELS][TO_STAMP_BLOCKS][TO_STAMP_N];
#define TO_STAMP(kern, i)                                                                                            \
    do {                                                                                                             \
        const unsigned lb_ = blockIdx.y * gridDim.x + blockIdx.x;                                                    \
        if (threadIdx.x == 0 && lb_ < TO_STAMP_BLOCKS) g_stamps[kern][lb_][i] = __builtin_amdgcn_s_memrealtime();    \
    } while (0)
// the same by lane 0 of EVERY wave, the latest kept: when the block's last wave passed the place
#define TO_STAMP_LAST(kern, i)                                                                                       \
    do {                                                                                                             \
        const unsigned lb_ = blockIdx.y * gridDim.x + blockIdx.x;                                                    \
        if ((threadIdx.x & 63) == 0 && lb_ < TO_STAMP_BLOCKS) atomicMax(&g_stamps[kern][lb_][i], (unsigned long long)__builtin_amdgcn_s_memrealtime()); \
    } while (0)
extern "C" int tohip_stamps_read(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * TO_STAMP_KERNELS * TO_STAMP_BLOCKS * TO_STAMP_N);
}
extern "C" int tohip_stamps_clear() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_stamps)) != hipSuccess) return -1;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * TO_STAMP_KERNELS * TO_STAMP_BLOCKS * TO_STAMP_N);
}
#else
#define TO_STAMP(kern, i) do { } while (0)
#define TO_STAMP_LAST(kern, i) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------
// workspace control block, per trajectory: the sum of the rewards as integers.  Every reward enters as rn(r * 2^shift)
// (exact for r in [1/2, 1): a reward is sigmoid of a non-negative log-odds sum), so the total is the same whatever kernel,
// block or order added it up: scalars are bitwise reproducible across the fused and the split step.
//   a[0..7]   k_traj_sparse<FUSED>: eight partial sums of r - 1/2 over the points it touches, one 128-byte line each
//             (blockIdx & 7 picks one: an address takes ~90 atomics / us); cleared by the next k_traj_probe, read by
//             k_traj_finish / the criterion kernel
//   b         k_traj_reward: ONE word — the sum (bits 0..47), the blocks that met a NaN (48..55), the blocks that have
//             arrived (56..63) — so that one relaxed atomic per block both adds and counts: the block whose add returns the
//             last arrival has the total in hand.  Zero between launches (that block clears it).
struct __attribute__((aligned(128))) RewardAcc {
    struct __attribute__((aligned(128))) Line { long long sum; unsigned nan; unsigned pad[29]; } a[8];
    struct __attribute__((aligned(128))) { unsigned long long word; unsigned long long pad[15]; } b;
};
static_assert(sizeof(RewardAcc) == 9 * 128, "RewardAcc is nine lines");
#define TO_REWARD_BLOCKS 128   // arrivals and NaN marks are 8-bit fields

__host__ __device__ inline int reward_shift(int64_t n) {
    int lg = 0;
    while (((int64_t)1 << lg) < n) ++lg;
    return 47 - lg;   // n * 2^shift <= 2^47: the sum stays below bit 48
}
__device__ __forceinline__ long long reward_fixed(float r, int shift) { return __double2ll_rn(ldexp((double)r, shift)); }
// scalars[0] = mean(rewards), [1] = loss_vis = 1/(mean+eps) (model.py:246), [2] = d loss_vis / d reward_n, [3] reserved
__device__ __forceinline__ void reward_scalars(long long total, bool anynan, int64_t n, int shift, float eps, float out[4]) {
    const float mean = anynan ? __builtin_nanf("") : (float)(ldexp((double)total, -shift) / (double)n);
    const float vis = 1.0f / (mean + eps);
    out[0] = mean;
    out[1] = vis;
    out[2] = (float)(-(double)vis * (double)vis / (double)n);
    out[3] = 0.f;
}
__device__ __forceinline__ void reward_scalars_from_a(const RewardAcc* acc, int64_t n, int shift, float eps, float out[4]) {
    long long s = 0;
    unsigned nan = 0;
    for (int i = 0; i < 8; ++i) { s += acc->a[i].sum; nan |= acc->a[i].nan; }
    reward_scalars((long long)n * (1ll << (shift - 1)) + s, nan != 0, n, shift, eps, out);   // the untouched points hold 1/2
}

struct TieRec {            // slots whose max equals the waypoint's max / whose min equals its min (a > 0), in arrival order
    int nmax, nmin;        // counts; > TO_TIE_CAP: overflow, scan every slot
    int maxrow[TO_TIE_CAP];
    int minrow[TO_TIE_CAP];
};

// ---------------------------------------------------------------------------------------------
// virtual waypoint records: v = w*C + c.  F.normalize (model.py:53), rig composition
// R_v = R(qn_w) R(q_c), t_v = t_w + R(qn_w) l_c, then the three projection rows and the Gaussian's centre.

// traj_off (optional): n_traj + 1 ascending body-waypoint offsets of several trajectories laid end to end
__device__ __forceinline__ void prep_wayrec(int v, const float* __restrict__ poses, const float* __restrict__ quats, int C,
                                            const float* __restrict__ rig_q, const float* __restrict__ rig_t,
                                            const EvalK& k, WayRec* __restrict__ rec, WayCold* __restrict__ cold,
                                            const int* __restrict__ traj_off, int n_traj, int wp_stride = 1, int traj_rows = 0, int seg_known = -1) {
    const int w = v / C, c = v - w * C;
    int seg = seg_known;
    if (traj_off != nullptr && seg < 0) {
        seg = 0;
        while (seg + 1 < n_traj && w >= traj_off[seg + 1]) ++seg;
    }
    if (seg < 0) seg = 0;
    // wp_stride > 1: the evaluated waypoints are every wp_stride-th row of the caller's arrays (model.py:215-217), read in place;
    // traj_rows > 0: every trajectory owns traj_rows rows of them (its evaluated waypoints are rows 0, wp_stride, ... of its own)
    {
        const int64_t row = (traj_rows > 0 && traj_off != nullptr) ? (int64_t)seg * traj_rows + (int64_t)(w - traj_off[seg]) * wp_stride
                                                                   : (int64_t)w * wp_stride;
        poses += 3 * (row - w);
        quats += 4 * (row - w);
    }
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
        r.sp[kk] = (float)(-(double)k.scd * (double)k.mean * s);
    }
    r.Lh = 0.f; r.L = 0.f; r.U = INFINITY; r.thr1 = INFINITY; r.sthr1 = INFINITY; r.azero = 0.f; r.pad = 0.f; r.seg = seg;
    rec[v] = r;
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

// (tile, waypoint) liveness: may the waypoint's sphere {d2 <= thr1} reach the tile (bounding sphere tb)?  Waypoints without the
// probe's "min is zero" proof (azero == 0) are searched densely: every tile is live for them.
// t = the waypoint's position, sp = -sqrt(cd) R mu (the record's fields)
__device__ __forceinline__ bool tile_live(const float (&t)[3], const float (&sp)[3], float thr, float sthr, float azero, const float4& tb, float inv_scd) {
    // the sphere's centre from the Gaussian's, in metres: (c - t) - R mu, with R mu = -sp / sqrt(cd)
    const float d0 = fmaf(sp[0], inv_scd, tb.x - t[0]), d1 = fmaf(sp[1], inv_scd, tb.y - t[1]), d2 = fmaf(sp[2], inv_scd, tb.z - t[2]);
    const float D2 = fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
    const float bound = fmaf(tb.w, fmaf(2.0f, sthr, tb.w), thr) * 1.00001f;  // (sthr + r)^2, rounded up
    return (azero == 0.f) || !(D2 > bound);
}

// Squared-distance bound thr such that  d2 > thr  =>  2^(-cd d2) < tau * (1 - 1e-4): p = S * 2^-A <= 2^(-cd d2) cannot
// reach tau.  The 1e-4 margin covers every rounding in p (~1e-6).  tau outside (0,1) -> +inf (never cull).
__device__ inline void cull_bound(float tau, float inv_var, float* thr, float* sthr) {
    cull_threshold(tau, inv_var, thr, sthr);
}

// the log-odds vector starts from zero (k_traj_sparse fills the flagged slots) and, when the caller asks for it, the rewards
// vector from sigmoid(0) = 1/2 (only the others are stored later): done by whoever evaluates a point block's first waypoint
typedef float f4v __attribute__((ext_vector_type(4)));
struct OutInit {   // n_traj log-odds vectors of npad floats (and rewards vectors of n floats) one after the other
    float* lo_zero;
    float* rewards_half;
    int64_t npad, n;
    int n_traj;
};
__device__ __forceinline__ void init_outputs(int64_t base, const OutInit& o) {
    if (o.lo_zero == nullptr) return;   // (a step of an optimisation run that leaves the N-sized outputs to the last one)
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

// what the probe does besides for the culled pass 1
struct ProbeCull {
    int on;
    int nslots;
    unsigned long long* live;      // V x fv_words: bit s of row v = slot s may hold a point waypoint v sees at all
};

// ---------------------------------------------------------------------------------------------
// probe: block per virtual waypoint.  Builds the waypoint's record, evaluates a strided sample of the sorted cloud
//   L = max p over the sample  (a lower bound of the true max: an attained value of p)
//   U = min p over the sample  (an upper bound of the true min; U == 0  =>  min_n p == 0: p is never negative)
// and resets what the step accumulates into: the waypoint's running extrema := (U, L), its tie lists, its row of the
// waypoint-major flag bits, and (block 0) the trajectories' reward sums.
// THREADS = 1024 for up to a few hundred waypoints (the samples are scattered single loads: the kernel is as long as one
// thread's chain of them); 256 beyond (eight blocks to a CU instead of two: the 1 024 waypoints of eight concurrent
// trajectories no longer queue).  Maximum and minimum do not depend on the order.
template <int TO_PROBE_THREADS>
__global__ void __launch_bounds__(TO_PROBE_THREADS)
k_traj_probe(CloudView cv, const float* __restrict__ poses, const float* __restrict__ quats, int C,
             const float* __restrict__ rig_q, const float* __restrict__ rig_t, EvalK k, WayRec* __restrict__ rec,
             WayCold* __restrict__ cold, Extrema* __restrict__ ext, TieRec* __restrict__ ties, const uint32_t* __restrict__ occ,
             int64_t occw, unsigned long long* __restrict__ fv, int fv_words, RewardAcc* __restrict__ acc,
             const int* __restrict__ traj_off, int* __restrict__ traj_off_ws, int n_traj, unsigned long long* __restrict__ cbits,
             int ncbits, int* __restrict__ ctr, int wp_stride, int traj_rows, ProbeCull pc, int V) {
    __shared__ float smx[TO_PROBE_THREADS / 64], smn[TO_PROBE_THREADS / 64];
    __shared__ float scull[4];
    const int v = blockIdx.x, t = threadIdx.x;
    TO_STAMP(TO_STAMP_PROBE, 0);
    for (int j = t; j < fv_words; j += TO_PROBE_THREADS) fv[(int64_t)v * fv_words + j] = 0ull;
    for (int j = v * TO_PROBE_THREADS + t; j < ncbits; j += V * TO_PROBE_THREADS) cbits[(int64_t)j * TO_CBIT_STRIDE] = 0ull;   // the candidate (slot, trajectory) bits
    if (v == 0 && t < TO_PL_SHARDS) ctr[t * TO_PL_STRIDE] = 0;   // the pair sub-lists' lengths
    if (v == 0) {
        for (int j = t; j < n_traj * 8; j += TO_PROBE_THREADS) { acc[j >> 3].a[j & 7].sum = 0; acc[j >> 3].a[j & 7].nan = 0u; }
        for (int j = t; j < n_traj; j += TO_PROBE_THREADS) acc[j].b.word = 0ull;   // (k_traj_reward leaves it zero; a launch that was cut short may not have)
        if (traj_off != nullptr)   // the calls that follow the forward (backward, finish) take no offsets: they read this copy
            for (int j = t; j <= n_traj; j += TO_PROBE_THREADS) traj_off_ws[j] = traj_off[j];
    }
    // CULL: the bounding spheres of the first slots this wave will test, requested now
    constexpr int kPre = 4, kWaves = TO_PROBE_THREADS / 64;
    float4 tb[kPre];
    if (pc.on) {
#pragma unroll
        for (int i = 0; i < kPre; ++i) {
            const int sl = ((t >> 6) + i * kWaves) * 64 + (t & 63);
            tb[i] = cv.bounds[sl < pc.nslots ? sl : 0];
        }
    }
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
    __shared__ WayRec srec;   // the block's record: built by thread 0, read by everybody from LDS (not back from global memory)
    // which trajectory is the waypoint's?  Wave 0 looks at up to 64 offsets at once (a walk along them was a dependent load per
    // trajectory: 1.2 us more for the eighth of eight)
    int seg = -1;
    if (traj_off != nullptr && t < 64) {
        seg = 0;
        const int w = v / C;
        for (int j0 = 1; j0 < n_traj; j0 += 64) {
            const int j = j0 + t;
            seg += __popcll(__ballot(j < n_traj && w >= traj_off[j]));   // the offsets ascend: the count of those at or below w
        }
    }
    if (t == 0) { prep_wayrec(v, poses, quats, C, rig_q, rig_t, k, &srec - v, cold, traj_off, n_traj, wp_stride, traj_rows, seg); rec[v] = srec; }
    __syncthreads();
    TO_STAMP(TO_STAMP_PROBE, 1);   // record built, samples requested
    const WayRec r = srec;
    float mx = 0.f, mn = INFINITY;
    const int rounds = min(kRounds, (cv.nsamples + TO_PROBE_THREADS * kBatch - 1) / (TO_PROBE_THREADS * kBatch));   // (a round is a memory round trip)
    for (int rd = 0; rd < rounds; ++rd) {
        if (rd > 0) {
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const int sj = t + (rd * kBatch + j) * TO_PROBE_THREADS;
                const int sc = sj < cv.nsamples ? sj : 0;
                px[j] = cv.samples[sc]; py[j] = cv.samples[TO_PROBE_MAX + sc]; pz[j] = cv.samples[2 * TO_PROBE_MAX + sc];
            }
        }
        // two samples to a packed instruction (per element the operations of vis_p: the same bits): with a thousand waypoints the
        // probe is ~6 % of a dense pass 1 and bound by instruction issue, not by its chain of loads
#pragma unroll
        for (int j = 0; j < kBatch; j += 2) {
            const int sj0 = t + (rd * kBatch + j) * TO_PROBE_THREADS, sj1 = sj0 + TO_PROBE_THREADS;
            if (sj0 < cv.nsamples) {   // (sj1 > sj0: the second one is the first to run out)
                const f2 p = vis_p_pk(r, k, f2{px[j], px[j + 1]}, f2{py[j], py[j + 1]}, f2{pz[j], pz[j + 1]}) *
                             f2{occ_one(occ, occw, v, (int64_t)sj0 * cv.sample_step), sj1 < cv.nsamples ? occ_one(occ, occw, v, (int64_t)sj1 * cv.sample_step) : 1.0f};
                mx = fmaxf(mx, p.x);
                mn = fminf(mn, p.x);
                if (sj1 < cv.nsamples) { mx = fmaxf(mx, p.y); mn = fminf(mn, p.y); }
            }
        }
    }
    for (int s = 32; s > 0; s >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, s)); mn = fminf(mn, __shfl_xor(mn, s)); }
    if ((t & 63) == 0) { smx[t >> 6] = mx; smn[t >> 6] = mn; }
    __syncthreads();
    TO_STAMP(TO_STAMP_PROBE, 2);   // sample evaluated and reduced
    if (t == 0) {
        for (int w = 1; w < TO_PROBE_THREADS / 64; ++w) { mx = fmaxf(mx, smx[w]); mn = fminf(mn, smn[w]); }
        if (!(mn <= mx)) { mx = 0.f; mn = INFINITY; }   // a NaN among the sampled p (fmax/fmin drop it): no usable bound
        // A NaN / inf coordinate anywhere in the cloud (noted by tohip_pack_cloud): the reference's p is NaN for that point, its
        // min() and max() over p are NaN (torch propagates it), and with them every reward of every waypoint and every gradient
        // (model.py:226-231).  The waypoint's maximum is set to NaN here — it sorts above every float in the integer order, so
        // no atomicMax replaces it — which makes the waypoint degenerate: every slot is searched, flagged, and adds NaN.
        const bool poisoned = cv.hdr[1] != 0;
        if (poisoned) { mx = __builtin_nanf(""); mn = INFINITY; }
        float thr, sthr;
        cull_bound(0.5f * mx, k.inv_var, &thr, &sthr);   // d2 > thr  =>  p < L/2 <= M/2: neither the max nor flagged
        rec[v].Lh = (mn == 0.f && !poisoned) ? 0.49f * mx : 0.f;
        rec[v].L = mx;
        rec[v].U = mn;
        rec[v].thr1 = thr;
        rec[v].sthr1 = sthr;
        rec[v].azero = (mn == 0.f) ? 1.f : 0.f;
        scull[0] = thr; scull[1] = sthr; scull[2] = (mn == 0.f) ? 1.f : 0.f;
        Extrema e;
        e.nmn = -__builtin_bit_cast(int, mn);
        e.mx = __builtin_bit_cast(int, mx);
        e.pad[0] = e.pad[1] = 0;
        ext[v] = e;
        ties[v].nmax = 0;
        ties[v].nmin = 0;
    }
    TO_STAMP(TO_STAMP_PROBE, 3);   // bounds, cull distance, resets
    if (!pc.on) return;
    // CULL: which 256-point slots can waypoint v reach at all — row v of `live`, a bit per slot: the culled pass 1 deals the set
    // bits to its waves, k_traj_sparse asks them whether a pair was evaluated
    __syncthreads();
    const float thr = scull[0], sthr = scull[1], azero = scull[2], inv_scd = 1.0f / k.scd;
    const int lane = t & 63;
    auto test_word = [&](int w, const float4& b) {
        const int sl = w * 64 + lane;
        const unsigned long long word = __ballot(sl < pc.nslots && tile_live(r.t, r.sp, thr, sthr, azero, b, inv_scd));
        if (lane == 0) pc.live[(int64_t)v * fv_words + w] = word;
    };
#pragma unroll
    for (int i = 0; i < kPre; ++i)
        if ((t >> 6) + i * kWaves < fv_words) test_word((t >> 6) + i * kWaves, tb[i]);
    for (int w0 = (t >> 6) + kPre * kWaves; w0 < fv_words; w0 += kPre * kWaves) {   // the rest, kPre loads in flight at a time
#pragma unroll
        for (int i = 0; i < kPre; ++i) {
            const int sl = (w0 + i * kWaves) * 64 + lane;
            tb[i] = cv.bounds[sl < pc.nslots ? sl : 0];
        }
#pragma unroll
        for (int i = 0; i < kPre; ++i)
            if (w0 + i * kWaves < fv_words) test_word(w0 + i * kWaves, tb[i]);
    }
    TO_STAMP(TO_STAMP_PROBE, 4);   // reachable slots
}

// ---------------------------------------------------------------------------------------------
// pass 1: p of every (point, waypoint) pair -> part[slot * V + v] = (min, max) over the slot's 256 points, and the
// waypoint's running extrema.  Only a wave whose value improves on the probe's (U, L) issues an atomic: the number of
// points above a 4 096-sample maximum is ~n/4096, in a few dozen slots per waypoint.

#define TO_P 4   // points per lane where a wave is one slot (the sparse and the pair kernel; the plan's point blocks)

// What the lane holding a slot's (min, max) of waypoint v does with them besides storing them: a slot whose maximum reaches Lh is
// a candidate (-> returns true; the common case is ONE compare), and only a candidate can improve on L; the minimum needs a look
// only while the probe has not exhibited a zero (U != 0: a wave-uniform branch).
__device__ __forceinline__ bool fold_extrema(Extrema* __restrict__ ext, int v, float mn, float mx, const WayRec& r) {
    bool cand = false;
    if (!(mx < r.Lh)) {
        cand = true;
        const int mxb = __builtin_bit_cast(int, mx);
        if (mxb > __builtin_bit_cast(int, r.L)) atomicMax(&ext[v].mx, mxb);
    }
    const int Ub = __builtin_bit_cast(int, r.U);
    if (Ub != 0) {
        const int mnb = __builtin_bit_cast(int, mn);
        if (mnb < Ub) atomicMax(&ext[v].nmn, -mnb);
    }
    return cand;
}

// A candidate (slot, trajectory) sets its bit — bit `slot` of the trajectory's row of fv_words words; k_traj_sparse walks the
// set bits.  No answer is waited for.  The words sit TO_CBIT_STRIDE words apart, one to a 128-byte line: device-scope atomics on
// one line are served one after the other (~80 ns each, measured), and a step sets thousands of these bits.
__device__ __forceinline__ unsigned long long* cbit_word(unsigned long long* cbits, int fv_words, int seg, int w) {
    return cbits + ((int64_t)seg * fv_words + w) * TO_CBIT_STRIDE;
}
__device__ __forceinline__ void mark_candidate(unsigned long long* __restrict__ cbits, int fv_words, int slot, int seg) {
    atomicOr(cbit_word(cbits, fv_words, seg, slot >> 6), 1ull << (slot & 63));
}

// DENSE: persistent blocks, as many as the chip holds at once (host: occupancy x CUs).  A lane owns EIGHT consecutive points
// (four independent packed evaluation chains per waypoint: a SIMD holds only ~5 of these waves, which issue in age order, so the
// instruction-level parallelism has to come from inside the wave), a wave two 256-point slots, a block 2048 points.  The
// (point block, waypoint) pairs are one flat range cut into gridDim.x equal pieces: a block's piece is a run of consecutive
// waypoints of one point block (seldom two), so every block does the same number of evaluations to within one waypoint, loads
// its points once (twice), and all blocks end together: no dispatch order, no tail.  A lane's (min, max) stores of consecutive
// waypoints are adjacent in `part`.
#define TO_PD 8
template <bool OCC>
__global__ void __launch_bounds__(TO_BLOCK)
k_traj_pass1_dense(CloudView cv, const WayRec* __restrict__ rec, int V, int nblk, EvalK k, float2* __restrict__ part,
                   Extrema* __restrict__ ext, unsigned long long* __restrict__ cbits, int fv_words,
                   const uint32_t* __restrict__ occ, int64_t occw, OutInit oi, unsigned long long* __restrict__ stamps) {
    constexpr int P = TO_PD;
    const int lane = threadIdx.x & 63;
    // diagnostic only (tohip_profile_clock): shader-clock and 100 MHz real-time stamps of this block, to a buffer nothing else reads
    unsigned long long st0 = 0, sr0 = 0;
    if (stamps != nullptr && threadIdx.x == 0) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
    // two of the evaluation's constants live in vector registers for the whole kernel (common.hpp, vis_p_pk)
    f2 eps2 = pk_splat(k.eps), l2e2 = pk_splat(k.l2e_eps), scd2 = pk_splat(k.scd);
    asm volatile("" : "+v"(eps2), "+v"(l2e2), "+v"(scd2));
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
        float2* prow = part + (int64_t)slot * V;
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
                p[i / 2] = vis_p_pk(r, k, f2{x[i], x[i + 1]}, f2{y[i], y[i + 1]}, f2{z[i], z[i + 1]}, eps2, l2e2, scd2) * f2{om[i], om[i + 1]};
            // (three-operand maxima: four instructions for eight values)
            const float m1 = fmaxf(fmaxf(p[0].x, p[0].y), p[1].x), m2 = fmaxf(fmaxf(p[1].y, p[2].x), p[2].y);
            float mx = fmaxf(m2, fmaxf(fmaxf(p[3].x, p[3].y), m1));
            mx = half_max31_nn_fused(mx);
            // the minimum is wanted only while the probe has not exhibited a zero (U != 0, wave-uniform): p is never negative, so
            // with one zero in the cloud min p = 0 whatever this slot holds, and nothing downstream looks at a slot's minimum then
            float mn = 0.f;
            if (__builtin_bit_cast(int, r.U) != 0) {
                mn = fminf(fminf(fminf(p[0].x, p[0].y), fminf(p[1].x, p[1].y)), fminf(fminf(p[2].x, p[2].y), fminf(p[3].x, p[3].y)));
                mn = half_min31_nn_fused(mn);
            }
            if ((lane & 31) == 31) {
                prow[v] = make_float2(mn, mx);
                if (fold_extrema(ext, v, mn, mx, r)) mark_candidate(cbits, fv_words, slot, r.seg);
            }
        }
        u += v1 - v0;
    }
    if (stamps != nullptr && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - sr0;
        unsigned long long* ext_st = stamps + 2 * (int64_t)gridDim.x + 4 * (int64_t)blockIdx.x;   // where and when the block ran
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        ext_st[0] = sr0; ext_st[1] = __builtin_amdgcn_s_memrealtime(); ext_st[2] = hw; ext_st[3] = xcc;
    }
}

__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// CULL: most (slot, waypoint) pairs cannot contribute — on the BASELINE slab a waypoint reaches 2.5 % of the slots — and the
// probe has marked, per waypoint, the slots it can reach.  grid = (2 or 1, V) blocks of sixteen waves: the waves of row v
// deal the waypoint's list among themselves, one (slot, waypoint) pair at a time — the record sits in scalar registers for the
// wave's whole life, the next pair's points are requested before the current pair is evaluated; every pair costs the same, so
// the chip is evenly loaded whatever the pairs' distribution over the slots.  The waypoint's extrema and its candidates' bits
// are combined in LDS and leave the block as ONE atomic per word: thousands of device-scope atomics on a few lines, one per
// pair, were what the kernel's time consisted of.  A pair that is not listed is not written: k_traj_sparse takes (min, max) =
// (the proven 0, -inf: never flagged) for it from the `live` bit.
#define TO_CULL_LIST 2048
struct CullLds {
    unsigned long long cand[TO_PROBE_MAXFW];   // the candidates this block finds
    unsigned long long row[TO_PROBE_MAXFW];    // the waypoint's reachable slots (its row of `live`)
    int pre[TO_PROBE_MAXFW];                   // set bits before each word
    int list[TO_CULL_LIST];                    // the reachable slots in ascending order, while they fit
    int mx[2 * TO_SP_WAVES], mn[2 * TO_SP_WAVES];   // (of the block's NW <= TO_SP_WAVES waves)
    int total;
};

// the j-th reachable slot of the block's waypoint (j < total, wave-uniform): the word by a 64-ary search of the prefix, the bit
// by its rank inside the word — every lane looks at one candidate
__device__ __forceinline__ int cull_select(const CullLds& L, int fv_words, int j, int lane) {
    int lo = 0, hi = fv_words;   // the word is in [lo, hi): pre[lo] <= j
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) >> 6;
        const int idx = lo + lane * step;
        const int cnt = __popcll(__ballot(idx < hi && L.pre[idx] <= j));   // >= 1
        lo += (cnt - 1) * step;
        hi = min(hi, lo + step);
    }
    const unsigned long long word = uniform_u64(L.row[lo]);
    const int r = j - L.pre[lo];
    const bool hit = ((word >> lane) & 1ull) && __popcll(word & ((1ull << lane) - 1ull)) == r;
    return lo * 64 + __builtin_ctzll(__ballot(hit));
}

template <bool OCC, int NW>
__global__ void __launch_bounds__(NW * 64)
k_traj_pass1_cull(CloudView cv, const WayRec* __restrict__ rec, int V, EvalK k, float2* __restrict__ part, Extrema* __restrict__ ext,
                  unsigned long long* __restrict__ cbits, int fv_words, const unsigned long long* __restrict__ live,
                  const uint32_t* __restrict__ occ, int64_t occw, OutInit oi) {
    __shared__ CullLds L;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int v = blockIdx.y;
    TO_STAMP(TO_STAMP_CULL, 0);
    for (int w = t; w < fv_words; w += NW * 64) { L.row[w] = live[(int64_t)v * fv_words + w]; L.cand[w] = 0ull; }
    // the outputs' start values (the dense pass 1 sets them itself): stores nobody here waits for
    if (oi.lo_zero != nullptr)
        for (int64_t i = (((int64_t)v * gridDim.x + blockIdx.x) * (NW * 64) + t) * 4; i < oi.npad; i += (int64_t)gridDim.y * gridDim.x * (NW * 64) * 4)
            init_outputs(i, oi);
    __syncthreads();
    TO_STAMP(TO_STAMP_CULL, 1);   // the waypoint's row of reachable slots is in LDS
    if (t < 64) {   // exclusive prefix of the words' popcounts, 64 words at a time
        int carry = 0;
        for (int w0 = 0; w0 < fv_words; w0 += 64) {
            const int c = (w0 + lane < fv_words) ? __popcll(L.row[w0 + lane]) : 0;
            int incl = c;
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                const int up = __shfl_up(incl, sh);
                if (lane >= sh) incl += up;
            }
            if (w0 + lane < fv_words) L.pre[w0 + lane] = carry + incl - c;
            carry += __shfl(incl, 63);
        }
        if (lane == 0) L.total = carry;
    }
    __syncthreads();
    const int n = L.total;
    const bool listed = n <= TO_CULL_LIST;   // block-uniform: the set bits written out once, a pair is then one LDS read
    if (listed) {
        for (int w = wave; w < fv_words; w += NW) {
            const unsigned long long word = L.row[w];
            if ((word >> lane) & 1ull) L.list[L.pre[w] + __popcll(word & ((1ull << lane) - 1ull))] = w * 64 + lane;
        }
        __syncthreads();
    }
    TO_STAMP(TO_STAMP_CULL, 2);   // prefix and list
    const int wr = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * NW) + wave), WR = (int)gridDim.x * NW;
    const WayRec& r = rec[v];
    int bmx = __builtin_bit_cast(int, r.L), bmn = __builtin_bit_cast(int, r.U);   // p >= +0: the bit patterns order like the values
    // A wave takes TWO reachable slots at a time — lanes 0..31 the one, 32..63 the other, eight points per lane: the dense kernel's
    // four independent packed chains and half-wave reductions (708 issue cycles for two pairs where one pair alone took 584).
    const int half = lane >> 5, nunits = (n + 1) >> 1;
    auto unit_slot = [&](int u) {   // lane's slot of unit u (the second half of a last, odd unit repeats the first: not stored)
        if (listed) return L.list[min(2 * u + half, n - 1)];
        const int s0 = cull_select(L, fv_words, 2 * u, lane), s1 = cull_select(L, fv_words, min(2 * u + 1, n - 1), lane);   // (wave-uniform ranks)
        return half ? s1 : s0;
    };
    if (wr < nunits) {
        constexpr int P8 = TO_PD;
        const f2 eps2 = pk_splat(k.eps), l2e2 = pk_splat(k.l2e_eps), scd2 = pk_splat(k.scd);
        // one unit: the lane's eight points of slot `cur` against the block's waypoint
        auto eval_unit = [&](int u, int cur, const float (&x)[P8], const float (&y)[P8], const float (&z)[P8]) {
            const bool valid = 2 * u + half < n;
            float om[P8];
            load_occ<P8, OCC>(occ, occw, v, (int64_t)cur * TO_SLOT + (lane & 31) * P8, om);
            f2 p[P8 / 2];
#pragma unroll
            for (int i = 0; i < P8; i += 2)
                p[i / 2] = vis_p_pk(r, k, f2{x[i], x[i + 1]}, f2{y[i], y[i + 1]}, f2{z[i], z[i + 1]}, eps2, l2e2, scd2) * f2{om[i], om[i + 1]};
            const float m1 = fmaxf(fmaxf(p[0].x, p[0].y), p[1].x), m2 = fmaxf(fmaxf(p[1].y, p[2].x), p[2].y);
            const float mx = half_max31_nn_fused(fmaxf(m2, fmaxf(fmaxf(p[3].x, p[3].y), m1)));
            float mn = 0.f;   // wanted only while the probe has not exhibited a zero (see the dense kernel)
            if (__builtin_bit_cast(int, r.U) != 0) {
                mn = fminf(fminf(fminf(p[0].x, p[0].y), fminf(p[1].x, p[1].y)), fminf(fminf(p[2].x, p[2].y), fminf(p[3].x, p[3].y)));
                mn = half_min31_nn_fused(mn);
            }
            if ((lane & 31) == 31 && valid) {
                part[(int64_t)cur * V + v] = make_float2(mn, mx);
                // what fold_extrema does with atomics, on the half-wave's own running values
                if (!(mx < r.Lh)) {
                    atomicOr(&L.cand[cur >> 6], 1ull << (cur & 63));
                    bmx = max(bmx, __builtin_bit_cast(int, mx));
                }
                if (__builtin_bit_cast(int, r.U) != 0) bmn = min(bmn, __builtin_bit_cast(int, mn));
            }
        };
        // Two register sets take turns: while one unit is evaluated the next one's points arrive in the other set (a single set
        // refilled by 24 moves per unit cost a seventh of the loop's vector instructions).
        float xa[P8], ya[P8], za[P8], xb[P8], yb[P8], zb[P8];
        int sa = unit_slot(wr), sb = 0;
        load_points<P8>(cv.soa, cv.npad, (int64_t)sa * TO_SLOT + (lane & 31) * P8, xa, ya, za);
        for (int u = wr; u < nunits; u += 2 * WR) {
            const bool more1 = u + WR < nunits, more2 = u + 2 * WR < nunits;
            if (more1) {
                sb = unit_slot(u + WR);
                load_points<P8>(cv.soa, cv.npad, (int64_t)sb * TO_SLOT + (lane & 31) * P8, xb, yb, zb);
            }
            eval_unit(u, sa, xa, ya, za);
            if (!more1) break;
            if (more2) {
                sa = unit_slot(u + 2 * WR);
                load_points<P8>(cv.soa, cv.npad, (int64_t)sa * TO_SLOT + (lane & 31) * P8, xa, ya, za);
            }
            eval_unit(u + WR, sb, xb, yb, zb);
        }
    }
    TO_STAMP(TO_STAMP_CULL, 3);   // wave 0's pairs evaluated
    if ((lane & 31) == 31) { L.mx[2 * wave + half] = bmx; L.mn[2 * wave + half] = bmn; }
    __syncthreads();
    TO_STAMP(TO_STAMP_CULL, 4);   // every wave's
    if (t == 0) {
        for (int w = 0; w < 2 * NW; ++w) { bmx = max(bmx, L.mx[w]); bmn = min(bmn, L.mn[w]); }
        if (bmx > __builtin_bit_cast(int, r.L)) atomicMax(&ext[v].mx, bmx);
        if (bmn < __builtin_bit_cast(int, r.U)) atomicMax(&ext[v].nmn, -bmn);
    }
    for (int w = t; w < fv_words; w += NW * 64) {
        const unsigned long long word = L.cand[w];
        if (word) atomicOr(cbit_word(cbits, fv_words, r.seg, w), word);
    }
    TO_STAMP(TO_STAMP_CULL, 5);
}

// ---------------------------------------------------------------------------------------------
// sparse: blocks of NW waves (4 by default: up to four blocks to a CU, 1 024 resident; 16 behind TOHIP_SPARSE_WAVES for comparison)
// take the candidate (slot, trajectory) bits pass 1 set (6-8 % of the slots on the BASELINE workloads; 15-37 % once an optimisation
// has moved the trajectory or in an indoor cloud).  Every wave holds the slot's 256 points (four consecutive points per lane, as
// two packed pairs); the slot's flagged waypoints of a trajectory, in ascending order, are dealt by rank & 15 into SIXTEEN partial
// sums — wave w keeps the 16 / NW of them with (rank & 15) % NW == w — a fixed order that depends only on the flag set, which DENSE
// and CULL share, and bitwise the same sums whatever NW is.
//   flags     lane per waypoint: the slot's (min, max) against the waypoint's final extrema — flagged when its maximum
//             has p_hat >= 1/2 (the predicate applied per point below, on an attained value), or it holds an argmin point
//             while a > 0 (that set carries gradient, model.py:226); a degenerate waypoint (max == min, or a NaN: the
//             reference's 0/0 for EVERY point, model.py:227) flags every slot and adds NaN.  Slots holding an extremum
//             enter the waypoint's tie list.  The flag bits go to fv (bit of this slot), the flagged pairs to the pair list.
//   staging   what an evaluation reads of a flagged waypoint's record, with a and 1/M, goes to LDS in one parallel load per
//             256 waypoints: a wave's chain is then arithmetic, not a scalar load per waypoint
//   forward   p_hat = (p - a)/M, clip to [0.5, 1-eps], log-odds (model.py:226-231), summed per wave, the sixteen waves in
//             order; unflagged pairs contribute exactly 0.  lo_sum[slot] is complete when the block is done with it:
//   rewards   r = sigmoid(lo) to the caller's order (points with lo == 0 keep the prefilled 1/2), their sum as integers

enum { TO_SP_FWD = 0, TO_SP_FUSED = 2 };
#define TO_SP_CW 4                      // flag words (x 64 waypoints) staged at a time
#define TO_SP_STAGE (TO_SP_CW * 64)

struct SparseArgs {
    CloudView cv;
    const WayRec* rec;
    const Extrema* ext;
    EvalK k;
    const float2* part;
    int V, nslots, vwords, fv_words;
    const unsigned long long* cbits;   // candidate (slot, trajectory) bits (pass 1): n_traj rows of fv_words words
    const unsigned long long* live;    // culled pass 1: bit (v, slot) = the pair was evaluated (part holds it); NULL: all were
    int2* plist;                 // the step's flagged (slot, waypoint) pairs: TO_PL_SHARDS sub-lists of plcap entries, in no particular order
    int* npairs;                 //   their lengths, TO_PL_STRIDE ints apart (FWD / FUSED append, the pair kernel reads)
    int64_t plcap;
    unsigned long long* fv;
    TieRec* ties;
    float* lo_sum;               // n_traj x npad; FWD / FUSED write the flagged slots, BWD reads
    float* minmax;               // FWD / FUSED: (a, M) per virtual waypoint (block 0 writes it)
    const uint32_t* occ;
    int64_t occw;
    const int* toff;
    int n_traj, C;
    float* rewards;              // FUSED: n_traj x n, caller's order
    int prefilled;               // FUSED: rewards hold 1/2 everywhere
    RewardAcc* acc;              // FUSED
    int shift;
    const float* grad_rewards;   // BWD upstream: dL/d rewards (n_traj x n, caller's order), or
    const float* scalars;        //               scalars[4*traj+2] * gout[traj], or (all three NULL) 1
    const float* gout;
    float* bpart;
    float* unit;                 // n_traj x npad, packed order: r (1 - r) of a candidate slot's points (0 for pads) — FUSED writes it with the
                                 //   slot's rewards, the pair kernel reads it when the upstream gradient is the unit one
};

// a flagged waypoint as the evaluations read it (LDS): line 0 of its record, its normalisation, its index
struct __attribute__((aligned(16))) StagedWay {
    float t[3], f0[3], f1[3], f2[3], sp[3];
    float a, invM;
    int v;
    float pad[2];
};
static_assert(sizeof(StagedWay) == 80, "StagedWay is 20 floats");

// NW = waves of the block (16 or 4); SFW = flag words held (TO_SP_MAXW, or 8 where one trajectory of at most TO_SP_STAGE virtual
// waypoints is all there is: 36 KB instead of 44, four 4-wave blocks to a CU)
template <int NW, int SFW>
struct SparseLds {
    unsigned long long sflag[SFW];
    StagedWay stage[TO_SP_STAGE];
    float4 spart[TO_SP_WAVES][64];   // the SIXTEEN partial sums of a slot (rank & 15), whatever NW is
    int any[NW];
    int pbase;
};

__device__ __forceinline__ f2 log_odds_pk(const EvalK& k, float a, float invM, f2 p) {
    f2 ph = (p - pk_splat(a)) * pk_splat(invM);
    ph = f2{__builtin_amdgcn_fmed3f(ph.x, 0.5f, k.clip_hi), __builtin_amdgcn_fmed3f(ph.y, 0.5f, k.clip_hi)};
    // log(ph/(1-ph)) as a difference of logs: exactly 0 at ph = 0.5
    const f2 om = pk_splat(1.0f) - ph;
    return (f2{to_log2(ph.x), to_log2(ph.y)} - f2{to_log2(om.x), to_log2(om.y)}) * pk_splat(0.693147180559945f);
}

// what an evaluation reads of waypoint v, with its final normalisation, into the stage
__device__ __forceinline__ void stage_way(const SparseArgs& a, int v, StagedWay& dst) {
    const WayRec& r = a.rec[v];
    StagedWay sw;
    for (int i = 0; i < 3; ++i) { sw.t[i] = r.t[i]; sw.f0[i] = r.f0[i]; sw.f1[i] = r.f1[i]; sw.f2[i] = r.f2[i]; sw.sp[i] = r.sp[i]; }
    float pmax, M;
    load_norm(a.ext[v], sw.a, pmax, M, sw.invM);
    if (!(M > 0.f) || !(sw.invM < INFINITY)) sw.invM = __builtin_nanf("");   // degenerate: every log-odds is NaN (0/0 in the reference)
    sw.v = v;
    sw.pad[0] = sw.pad[1] = 0.f;
    dst = sw;
}

// one candidate slot for trajectory tr — its virtual waypoints [v_lo, v_hi), its own log-odds vector, rewards, sums and rank
// count, so that its results are the ones a run of that trajectory alone produces; blockDim.x = 64 NW.  Returns with
// every thread past its last use of the LDS.
// The slot's flagged waypoints, in ascending order, go to SIXTEEN partial sums by rank & 15, added up in the order 0..15: a
// 16-wave block keeps one per wave, a 4-wave block four per wave (ranks w, w + 4, w + 8, w + 12 mod 16) — the same sixteen sums
// in the same order, hence the same bits whichever block shape the host picks.
template <int MODE, bool OCC, int NW, int SFW>
__device__ __forceinline__ void sparse_slot(const SparseArgs& a, int slot, int tr, int acc_line, SparseLds<NW, SFW>& L) {
    constexpr int NSET = TO_SP_WAVES / NW;   // partial sums a wave keeps
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    TO_STAMP(TO_STAMP_SPARSE, 1);   // the block's candidate is known
    const EvalK& k = a.k;
    const int64_t base = (int64_t)slot * TO_SLOT + lane * 4;
    float x[4], y[4], z[4];
    load_points<4>(a.cv.soa, a.cv.npad, base, x, y, z);
    // where the slot's points live in the caller's order (rewards): requested now, wanted after the forward
    int4 o4 = make_int4(0, 0, 0, 0);
    if (MODE == TO_SP_FUSED && wave == 0) o4 = *reinterpret_cast<const int4*>(a.cv.perm + base);

    const int v_lo = a.toff ? a.toff[tr] * a.C : 0, v_hi = a.toff ? a.toff[tr + 1] * a.C : a.V;
    const int w_lo = v_lo >> 6, w_hi = min(a.vwords, (v_hi + 63) >> 6);
    const int wb = SFW < TO_SP_MAXW ? w_lo : 0;   // L.sflag[w - wb]: the small array holds the trajectory's own words only
    // A trajectory of up to TO_SP_STAGE virtual waypoints is staged whole, entry v - v_lo, with the block's first loads: which
    // waypoints are flagged is known two dependent loads later, and what is staged does not depend on it.
    const bool direct = v_hi - v_lo <= TO_SP_STAGE;   // block-uniform
    const bool stager = direct && t < v_hi - v_lo;
    StagedWay sw;
    if (stager) stage_way(a, v_lo + t, sw);   // into registers: its loads are in flight beside those of the flags below
    // ---- flags of this slot, one word per 64 waypoints ----
    int mine = 0;
    for (int w = w_lo + wave; w < w_hi; w += NW) {
        const int v = w * 64 + lane;
        bool flag = false;
        if (v >= v_lo && v < v_hi) {
            float2 pq = a.part[(int64_t)slot * a.V + v];
            // a pair the culled pass 1 did not evaluate: min is the proven 0, max unknown (-inf: never flagged)
            if (a.live != nullptr && !((a.live[(int64_t)v * a.fv_words + (slot >> 6)] >> (slot & 63)) & 1ull)) pq = make_float2(0.f, -INFINITY);
            float av, pmax, M, invM;
            load_norm(a.ext[v], av, pmax, M, invM);
            const bool amin = av > 0.f;
            const bool degenerate = !(M > 0.f) || !(invM < INFINITY);
            flag = ((pq.y - av) * invM >= 0.5f) | (amin & (pq.x == av)) | degenerate;
            if (pq.y == pmax && M > 0.f) { const int i = atomicAdd(&a.ties[v].nmax, 1); if (i < TO_TIE_CAP) a.ties[v].maxrow[i] = slot; }
            if (amin && pq.x == av) { const int i = atomicAdd(&a.ties[v].nmin, 1); if (i < TO_TIE_CAP) a.ties[v].minrow[i] = slot; }
            if (flag) atomicOr(&a.fv[(int64_t)v * a.fv_words + (slot >> 6)], 1ull << (slot & 63));
        }
        const unsigned long long word = __ballot(flag);
        if (lane == 0) L.sflag[w - wb] = word;
        mine += __popcll(word);
    }
    if (stager) L.stage[t] = sw;
    if (lane == 0) L.any[wave] = mine;
    __syncthreads();
    TO_STAMP(TO_STAMP_SPARSE, 2);   // flags, staged records
    int npairs = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) npairs += L.any[w];
    if (npairs == 0) { __syncthreads(); return; }   // a candidate that is not flagged after all: lo_sum 0, rewards 1/2 (pass 1 wrote them)
    // room for the slot's pairs in the step's pair list: asked for by a thread with no other load in flight, wanted at the end
    // (an offset the compiler cannot see through keeps the add a plain one-lane atomic: the wave-aggregated form it would build
    // for a uniform address waits for the result on the spot)
    int pbase = 0;
    if (t == NW * 64 - 1) {
        int zero = 0;
        asm volatile("" : "+v"(zero));
        pbase = atomicAdd(a.npairs + (slot & (TO_PL_SHARDS - 1)) * TO_PL_STRIDE + zero, npairs);
    }

    {
        auto flagged = [&](int w) { return uniform_u64(L.sflag[w - wb]); };
        // stage the flagged waypoints of words [wc, wc + CW): entry = rank inside the chunk; returns their number
        auto stage_chunk = [&](int wc) {
            int cnt = 0, my_rank = -1, my_v = -1;
#pragma unroll
            for (int j = 0; j < TO_SP_CW; ++j) {
                const int w = wc + j;
                const unsigned long long bits = w < w_hi ? flagged(w) : 0ull;
                if ((t >> 6) == j && ((bits >> lane) & 1ull)) { my_rank = cnt + __popcll(bits & ((1ull << lane) - 1ull)); my_v = w * 64 + lane; }
                cnt += __popcll(bits);
            }
            if (my_rank >= 0) stage_way(a, my_v, L.stage[my_rank]);   // threads 0 .. 255: one flagged waypoint each
            __syncthreads();
            return cnt;
        };
        float lo[4];
        {
            f2 acc0[NSET], acc1[NSET];
#pragma unroll
            for (int q = 0; q < NSET; ++q) { acc0[q] = pk_splat(0.f); acc1[q] = pk_splat(0.f); }
            auto add_way = [&](const StagedWay& r, f2& s0, f2& s1) {
                float om[4];
                load_occ<4, OCC>(a.occ, a.occw, r.v, base, om);
                const f2 p0 = vis_p_pk(r, k, f2{x[0], x[1]}, f2{y[0], y[1]}, f2{z[0], z[1]}) * f2{om[0], om[1]};
                const f2 p1 = vis_p_pk(r, k, f2{x[2], x[3]}, f2{y[2], y[3]}, f2{z[2], z[3]}) * f2{om[2], om[3]};
                // a degenerate waypoint (staged with 1/M = NaN) makes every log-odds NaN, whatever med3 does with one; else + 0
                const f2 poison = pk_splat(r.invM != r.invM ? __builtin_nanf("") : 0.f);
                s0 = s0 + (log_odds_pk(k, r.a, r.invM, p0) + poison);
                s1 = s1 + (log_odds_pk(k, r.a, r.invM, p1) + poison);
            };
            // the wave's share of rank kk (wave == kk mod NW): partial sum (kk & 15) = set (kk / NW) mod NSET of this wave
            auto add_rank = [&](const StagedWay& r, int kk) {
                const int set = (kk / NW) & (NSET - 1);
#pragma unroll
                for (int q = 0; q < NSET; ++q)
                    if (q == set) add_way(r, acc0[q], acc1[q]);
            };
            int rank0 = 0;
            if (direct) {   // the flagged waypoints in ascending order, rank r to partial sum r & 15 — the order of the chunked walk below.
                // A wave goes straight to its ranks (a walk over every flagged bit by every wave was 2 us of a dense cloud's slot)
                for (int kk = wave; kk < npairs; kk += NW) {
                    int r = kk, w = w_lo;
                    unsigned long long bits = flagged(w);
                    while (r >= __popcll(bits)) { r -= __popcll(bits); bits = flagged(++w); }   // at most five words
                    const bool hit = ((bits >> lane) & 1ull) && __popcll(bits & ((1ull << lane) - 1ull)) == r;
                    add_rank(L.stage[w * 64 + __builtin_ctzll(__ballot(hit)) - v_lo], kk);
                }
            }
            for (int wc = w_lo; wc < w_hi && !direct; wc += TO_SP_CW) {
                const int cnt = stage_chunk(wc);
                for (int e = ((wave - rank0) & (NW - 1)); e < cnt; e += NW) add_rank(L.stage[e], rank0 + e);   // rank0 + e == wave (mod NW)
                rank0 += cnt;
                if (wc + TO_SP_CW < w_hi) __syncthreads();   // the stage is rewritten by the next chunk
            }
            TO_STAMP(TO_STAMP_SPARSE, 3);   // wave 0's share of the forward sweep
#pragma unroll
            for (int q = 0; q < NSET; ++q) L.spart[wave + NW * q][lane] = make_float4(acc0[q].x, acc0[q].y, acc1[q].x, acc1[q].y);
            __syncthreads();
            TO_STAMP(TO_STAMP_SPARSE, 4);   // every wave's
            float4 s = L.spart[0][lane];
#pragma unroll
            for (int w = 1; w < TO_SP_WAVES; ++w) {
                const float4 q = L.spart[w][lane];
                s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
            }
            lo[0] = s.x; lo[1] = s.y; lo[2] = s.z; lo[3] = s.w;
            if (wave == 0 && a.lo_sum != nullptr) *reinterpret_cast<float4*>(a.lo_sum + (int64_t)tr * a.cv.npad + base) = s;
        }

        // ---- rewards of the slot's points (wave 0) ----
        if (MODE == TO_SP_FUSED && wave == 0) {
            const int o[4] = {o4.x, o4.y, o4.z, o4.w};
            long long fsum = 0;
            bool fnan = false;
            float un[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ex = to_exp(-lo[j]);
                float rw = to_rcp(1.0f + ex);   // == k_traj_reward's value of rewards[perm[i]]
                float om = rw * ex;             // 1 - r = e / (1 + e): taken from the exponential, not from r (see pair_sums)
                if (lo[j] != lo[j]) { rw = lo[j]; om = lo[j]; }
                un[j] = base + j < a.cv.n ? 1.0f * rw * om : 0.f;   // the pair kernel's d reward / d lo_sum of this point, bit for bit
                if (base + j < a.cv.n) {                      // pads are not points
                    if (a.rewards != nullptr && (!a.prefilled || lo[j] != 0.f)) a.rewards[(int64_t)tr * a.cv.n + o[j]] = rw;
                    if (rw != rw) fnan = true;
                    else fsum += reward_fixed(rw, a.shift) - (1ll << (a.shift - 1));
                }
            }
            *reinterpret_cast<float4*>(a.unit + (int64_t)tr * a.cv.npad + base) = make_float4(un[0], un[1], un[2], un[3]);
            for (int s = 32; s > 0; s >>= 1) fsum += __shfl_xor(fsum, s);
            fnan = __any(fnan);
            if (lane == 0) {
                if (fsum != 0) atomicAdd(reinterpret_cast<unsigned long long*>(&a.acc[tr].a[acc_line].sum), (unsigned long long)fsum);
                if (fnan) atomicOr(&a.acc[tr].a[acc_line].nan, 1u);
            }
        }
        TO_STAMP(TO_STAMP_SPARSE, 5);   // sums, rewards
        if (t == NW * 64 - 1) L.pbase = pbase;   // (a device-scope atomic's answer takes microseconds: first read here)
        __syncthreads();
        TO_STAMP(TO_STAMP_SPARSE, 6);   // the pair list's answer
    }
    // the slot's pairs: a position each (the waves' counts, then the wave's words in its order, then the bit's rank)
    {
        int64_t at = (int64_t)(slot & (TO_PL_SHARDS - 1)) * a.plcap + L.pbase;
#pragma unroll
        for (int w = 0; w < NW; ++w) at += w < wave ? L.any[w] : 0;
        for (int w = w_lo + wave; w < w_hi; w += NW) {
            const unsigned long long word = uniform_u64(L.sflag[w - wb]);
            if ((word >> lane) & 1ull) a.plist[at + __popcll(word & ((1ull << lane) - 1ull))] = make_int2(slot, w * 64 + lane);
            at += __popcll(word);
        }
    }
    __syncthreads();   // the flag words are free for the block's next item
}

// (a, M) per virtual waypoint for the caller (tohip_traj_forward's minmax output): block 0, before its first slot
__device__ __forceinline__ void write_minmax(const SparseArgs& a) {
    if (a.minmax == nullptr) return;
    for (int v = threadIdx.x; v < a.V; v += blockDim.x) {
        float av, pmax, M, invM;
        load_norm(a.ext[v], av, pmax, M, invM);
        a.minmax[2 * v] = av;
        a.minmax[2 * v + 1] = M;
    }
}

// block b of nb works for trajectory b % n_traj: the q-th, (q + S)-th, ... set bit of that trajectory's candidate bits (bit s =
// slot s), q = b / n_traj, S = nb / n_traj (the host launches a multiple of n_traj blocks).  Every wave finds them from the same
// popcount prefix, 64 words at a time.
template <int MODE, bool OCC, int NW, int SFW>
__device__ __forceinline__ void sparse_walk(const SparseArgs& a, int b, int nb, SparseLds<NW, SFW>& L) {
    const int lane = threadIdx.x & 63;
    const int tr = b % a.n_traj, S = nb / a.n_traj;
    const unsigned long long* bits = a.cbits + (int64_t)tr * a.fv_words * TO_CBIT_STRIDE;
    int carry = 0, next = b / a.n_traj;
    for (int c0 = 0; c0 < a.fv_words; c0 += 64) {
        const unsigned long long word = (c0 + lane < a.fv_words) ? bits[(int64_t)(c0 + lane) * TO_CBIT_STRIDE] : 0ull;
        const int pc = __popcll(word);
        int incl = pc;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const int up = __shfl_up(incl, sh);
            if (lane >= sh) incl += up;
        }
        const int tot = __shfl(incl, 63);
        while (next < carry + tot) {   // wave- and block-uniform
            const int rnk = next - carry;
            const int f = __builtin_ctzll(__ballot(incl > rnk));
            const int rr = rnk - __shfl(incl - pc, f);
            const unsigned long long wf = uniform_u64((unsigned long long)__shfl((long long)word, f));
            const bool hit = ((wf >> lane) & 1ull) && __popcll(wf & ((1ull << lane) - 1ull)) == rr;
            sparse_slot<MODE, OCC, NW, SFW>(a, (c0 + f) * 64 + __builtin_ctzll(__ballot(hit)), tr, b & 7, L);
            next += S;
        }
        carry += tot;
    }
}

template <int MODE, bool OCC, int NW, int SFW>
__global__ void __launch_bounds__(NW * 64, (NW == 16 || SFW == 8) ? 4 : 3) k_traj_sparse(SparseArgs a, OptStep os) {   // (4 waves per SIMD: 128 registers; 3 where the LDS allows no more)
    __shared__ SparseLds<NW, SFW> L;
    // The step's prologue (opt_step.hpp: a trajectory's regularisers with their gradient, the step's Adam constants) needs the
    // positions only and is wanted by the finish kernel: it rides HERE, as one block per trajectory behind the candidates' blocks —
    // a quarter of those find no candidate and leave after 2 us, so it runs beside the others.  (In the probe's launch, until r04,
    // its 8 us of serial f64 were that launch's longest block: +1.5 us per optimiser step.)
    const int extra = os.mode ? os.n_traj : 0;
    if ((int)blockIdx.x >= (int)gridDim.x - extra) {
        __shared__ double olds[NW], osh[4];
        opt_prologue_block(os, (int)blockIdx.x - ((int)gridDim.x - extra), olds, osh);
        return;
    }
    TO_STAMP(TO_STAMP_SPARSE, 0);
    if (blockIdx.x == 0) write_minmax(a);
    sparse_walk<MODE, OCC, NW, SFW>(a, (int)blockIdx.x, (int)gridDim.x - extra, L);
    TO_STAMP(TO_STAMP_SPARSE, 7);
}

// ---------------------------------------------------------------------------------------------
// The gradient sums of the flagged (slot, waypoint) pairs, one WAVE per pair, the pairs dealt round-robin to every wave of the
// grid: a slot seen by forty waypoints and one seen by a single waypoint cost their blocks the same in k_traj_sparse, and
// the chip is evenly loaded here whatever the pairs' distribution over the slots (a dense indoor cloud flags 6x the pairs of the
// BASELINE slab in a third of the slots).  The pairs come from the list k_traj_sparse appended them to (their order there is
// arrival order; a pair's row does not depend on it).  A wave reads its pair's waypoint record with scalar loads (the index is
// uniform), the slot's four points per lane, their log-odds and, when the upstream gradient is per point, its entries.
//   per flagged pair: G = dL/dp_hat = g_n [0.5 <= p_hat <= 1-eps] / (p_hat (1 - p_hat)), dL/dp = G / M, plus
//   the shares of the min/max points (S1 = sum G (p_hat - 1)/M -> argmin set, S2 = sum G (-p_hat)/M -> argmax
//   set; torch splits them evenly among ties).  bpart[(v*nslots+slot)*16 ..]:
//     [0..2] sum w gy   [3..11] sum w y (x) gy   [12] S1   [13] S2        (w = G/M, gy = dp/dy, y = x - t)
//   per lane over its four points, then one DPP tree over the wave.  The fused step takes the sums with dL/d reward = 1
//   (they are linear in it; k_traj_finish scales them once the mean of the rewards is known).
// UNIT: the upstream gradient is the unit one AND this step's fused sparse kernel has left r (1 - r) of the slot's points in
// a.unit: four loads instead of four sigmoids per lane and pair (a slot's points meet dozens of waypoints).
template <bool OCC, bool UNIT>
__device__ __forceinline__ void pair_sums(const SparseArgs& a, int slot, int v, int lane) {
    const EvalK& k = a.k;
    const WayRec& r = a.rec[v];
    const int tr = a.toff ? r.seg : 0;
    const int64_t base = (int64_t)slot * TO_SLOT + lane * 4;
    float x[4], y[4], z[4];
    load_points<4>(a.cv.soa, a.cv.npad, base, x, y, z);
    const float4 l4 = *reinterpret_cast<const float4*>((UNIT ? a.unit : a.lo_sum) + (int64_t)tr * a.cv.npad + base);
    int4 o4 = make_int4(0, 0, 0, 0);
    if (!UNIT && a.grad_rewards != nullptr) o4 = *reinterpret_cast<const int4*>(a.cv.perm + base);
    float om[4];
    load_occ<4, OCC>(a.occ, a.occw, v, base, om);
    float av, pmax, M, invM;
    load_norm(a.ext[v], av, pmax, M, invM);
    if (!(M > 0.f) || !(invM < INFINITY)) invM = __builtin_nanf("");   // degenerate: nothing is active (NaN compares false)
    const float coef = (!UNIT && a.scalars) ? a.scalars[4 * tr + 2] * a.gout[tr] : 1.0f;
    const float lo[4] = {l4.x, l4.y, l4.z, l4.w};
    const int o[4] = {o4.x, o4.y, o4.z, o4.w};
    float gn[4];   // dL/d lo_sum_n
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (UNIT) { gn[j] = lo[j]; continue; }   // (l4 holds r (1 - r) then)
        // d reward / d lo_sum = r (1 - r) with 1 - r = e / (1 + e) = r e, e = exp(-lo_sum): where r -> 1 the difference 1 - r taken
        // from an f32 r keeps one or two digits (r = 0.9999999: 54 % per half ulp of r — the stress fixture traj_stress_31_101,
        // where the reference's own f32 sits within 2e-6 of the f64 result and r04's 1 - r was 8.5 % off)
        const float ex = to_exp(-lo[j]);
        float rw = to_rcp(1.0f + ex);
        float om = rw * ex;
        if (lo[j] != lo[j]) { rw = lo[j]; om = lo[j]; }
        const bool valid = base + j < a.cv.n;        // pads are not points
        float gr = coef;
        if (a.grad_rewards != nullptr) gr = valid ? a.grad_rewards[(int64_t)tr * a.cv.n + o[j]] : 0.f;
        gn[j] = valid ? gr * rw * om : 0.f;
    }
    f2 acc[TO_BWD_NSUM];
#pragma unroll
    for (int j = 0; j < TO_BWD_NSUM; ++j) acc[j] = pk_splat(0.f);
    const WayRecPk rp = wayrec_pk(r);   // the record's first line in register pairs (common.hpp)
    bool any_act = false;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        VisGrad2 vg;
        const f2 p = vis_p_pk_grad(rp, k, f2{x[2 * h], x[2 * h + 1]}, f2{y[2 * h], y[2 * h + 1]}, f2{z[2 * h], z[2 * h + 1]}, vg) *
                     f2{om[2 * h], om[2 * h + 1]};
        const f2 ph = (p - pk_splat(av)) * pk_splat(invM);
        const bool act0 = (ph.x >= 0.5f) && (ph.x <= k.clip_hi), act1 = (ph.y >= 0.5f) && (ph.y <= k.clip_hi);
        if (act0 | act1) {
            f2 g[3];
            dvis_dy_pk(rp, k, p, vg, g);
            const f2 G = f2{gn[2 * h], gn[2 * h + 1]} * pk_rcp(ph * (pk_splat(1.0f) - ph));
            f2 wgt = G * pk_splat(invM);
            wgt = f2{act0 ? wgt.x : 0.f, act1 ? wgt.y : 0.f};
            acc[12] = pk_fma(wgt, ph - pk_splat(1.0f), acc[12]);
            acc[13] = pk_fma(-wgt, ph, acc[13]);
            const f2 w0 = wgt * g[0], w1 = wgt * g[1], w2 = wgt * g[2];
            acc[0] = acc[0] + w0; acc[1] = acc[1] + w1; acc[2] = acc[2] + w2;
            acc[3] = pk_fma(vg.y0, w0, acc[3]); acc[4] = pk_fma(vg.y0, w1, acc[4]); acc[5] = pk_fma(vg.y0, w2, acc[5]);
            acc[6] = pk_fma(vg.y1, w0, acc[6]); acc[7] = pk_fma(vg.y1, w1, acc[7]); acc[8] = pk_fma(vg.y1, w2, acc[8]);
            acc[9] = pk_fma(vg.y2, w0, acc[9]); acc[10] = pk_fma(vg.y2, w1, acc[10]); acc[11] = pk_fma(vg.y2, w2, acc[11]);
            any_act = true;
        }
    }
    // the pair's 14 sums over the wave in one transposed reduction (common.hpp): lane l ends up with the sum wave_sum16_index(l),
    // and the first sixteen lanes store the pair's row (columns 14 and 15 are zero: the finish kernel adds all 16 of a row)
    float sum[16];
#pragma unroll
    for (int j = 0; j < TO_BWD_NSUM; ++j) sum[j] = acc[j].x + acc[j].y;
    sum[14] = sum[15] = 0.f;
    float total = 0.f;
    if (__any(any_act)) total = wave_sum16_transposed(sum, lane);
    if (lane < 16) a.bpart[((int64_t)v * a.nslots + slot) * 16 + wave_sum16_index(lane)] = total;
}

// block b of nb (TO_SP_THREADS threads each): wave gw = wave * nb + b of 16 nb takes the pairs gw, gw + 16 nb, ...
template <bool OCC, bool UNIT>
__device__ __forceinline__ void pair_walk(const SparseArgs& a, int b, int nb) {
    const int lane = threadIdx.x & 63;
    TO_STAMP(TO_STAMP_PAIRS, 0);
    // wave-major numbering: the step's pairs rarely divide by the grid's waves, and the waves that take one pair more should be ONE per
    // SIMD of every CU, not all sixteen waves of the first blocks (21.5 k pairs over 4 096 waves: 65 CUs did six rounds, 191 five)
    const int gw = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) * nb + b), GW = nb * TO_SP_WAVES;
    // lane l holds sub-list l's length; the step's pairs are numbered through the sub-lists in order (an inclusive scan), and the
    // grid's waves take the numbers gw, gw + GW, ...: the chip is evenly loaded whatever the sub-lists' lengths
    const int cnt = a.npairs[lane * TO_PL_STRIDE];
    int incl = cnt;
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        const int up = __shfl_up(incl, sh);
        if (lane >= sh) incl += up;
    }
    const int excl = incl - cnt;
    const int P = __shfl(incl, 63);
    auto entry = [&](int p) {   // the p-th pair: in the first sub-list whose inclusive count exceeds p
        const int l = __popcll(__ballot(incl <= p));
        return a.plist[(int64_t)l * a.plcap + (p - __shfl(excl, l))];
    };
    int2 next = gw < P ? entry(gw) : make_int2(0, 0);
    TO_STAMP(TO_STAMP_PAIRS, 1);   // the lists' lengths and the wave's first pair
    for (int p = gw; p < P; p += GW) {
        const int2 cur = next;
        if (p + GW < P) next = entry(p + GW);
        pair_sums<OCC, UNIT>(a, __builtin_amdgcn_readfirstlane(cur.x), __builtin_amdgcn_readfirstlane(cur.y), lane);
    }
    TO_STAMP(TO_STAMP_PAIRS, 2);
    TO_STAMP_LAST(TO_STAMP_PAIRS, 3);   // the block's last wave
}

// os.mode == 2 (model() as one library call, loss_kernels.hip): one block more than the pairs need turns the integer reward sum
// — complete since k_traj_sparse — and the prologue's regulariser terms into model()'s scalars and loss terms.
template <bool OCC, bool UNIT>
__global__ void __launch_bounds__(TO_SP_THREADS) k_traj_pairs(SparseArgs a, OptStep os, float* __restrict__ scalars_out) {
    const int extra = os.mode == 2 ? 1 : 0;
    if (extra && blockIdx.x == gridDim.x - 1) {
        if (threadIdx.x == 0) {
            float sc[4];
            reward_scalars_from_a(a.acc, a.cv.n, a.shift, os.eps, sc);
            scalars_out[0] = sc[0]; scalars_out[1] = sc[1]; scalars_out[2] = sc[2]; scalars_out[3] = sc[3];
            const OptPro p = os.pro[0];
            RegOut o;
            o.l2 = p.l2; o.length = p.length; o.smooth = p.smooth;
            write_loss_terms(os.loss_log, (double)sc[1], o);
        }
        return;
    }
    pair_walk<OCC, UNIT>(a, (int)blockIdx.x, (int)gridDim.x - extra);
}

// ---------------------------------------------------------------------------------------------
// rewards = sigmoid(lo_sum) (model.py:237) scattered back to the caller's point order, mean and
// visibility loss (model.py:246) — the split step (a collective sits between forward and backward) and callers that hand
// in a log-odds vector of their own.
// One launch, thread per PACKED position: lo_sum and the permutation are read coalesced, the reward goes to the caller's
// order with a scattered 4-byte store — which is skipped for the points whose log-odds is exactly 0 when the caller says the
// rewards vector already holds sigmoid(0) = 1/2 everywhere (`prefilled`: tohip_traj_forward's rewards_half output; 98 % of the
// points on the BASELINE workloads).  Every block adds its integer sum to the trajectory's accumulator; the block whose
// arrival completes the count has the total (integer addition commutes: no dependence on the order), writes the scalars and
// clears the accumulator for the next launch.
// block bx of nbx of trajectory `traj` (its own log-odds vector, rewards vector, accumulator and scalars); blockDim.x = 1024
__device__ __forceinline__ void reward_block(const float* __restrict__ lo_sum, const int* __restrict__ perm, int64_t n, int64_t npad, float eps,
                                             int shift, int prefilled, float* __restrict__ rewards, RewardAcc* __restrict__ acc,
                                             float* __restrict__ scalars, int bx, int nbx, int traj, long long* lds) {
    lo_sum += (int64_t)traj * npad;
    rewards += (int64_t)traj * n;
    acc += traj;
    scalars += 4 * traj;
    long long s = 0;
    bool anynan = false;
    const long long half = 1ll << (shift - 1);
    const int64_t stride = (int64_t)nbx * TO_SP_THREADS * 4;
    for (int64_t i0 = ((int64_t)bx * TO_SP_THREADS + threadIdx.x) * 4; i0 < n; i0 += stride) {
        const float4 lo4 = *reinterpret_cast<const float4*>(lo_sum + i0);   // npad is a multiple of 2048: aligned, in bounds
        const float lo[4] = {lo4.x, lo4.y, lo4.z, lo4.w};
        const bool all0 = (lo4.x == 0.f) & (lo4.y == 0.f) & (lo4.z == 0.f) & (lo4.w == 0.f);
        if (all0 && prefilled) {   // sigmoid(0) evaluated as below is exactly 0.5
            s += half * min((int64_t)4, n - i0);
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
                if (r != r) anynan = true;
                else s += reward_fixed(r, shift);
            }
        }
    }
    for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
    anynan = __any(anynan);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { lds[wave] = s; lds[TO_SP_WAVES + wave] = anynan ? 1ll : 0ll; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    long long tot = 0;
    bool nan_ = false;
    for (int w = 0; w < TO_SP_WAVES; ++w) { tot += lds[w]; nan_ |= lds[TO_SP_WAVES + w] != 0ll; }
    const unsigned long long word = (1ull << 56) | (nan_ ? (1ull << 48) : 0ull) | ((unsigned long long)tot & 0xffffffffffffull);
    const unsigned long long old = __hip_atomic_fetch_add(&acc->b.word, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(old >> 56) != nbx - 1) return;
    const unsigned long long all = old + word;
    __hip_atomic_store(&acc->b.word, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
    float out[4];
    reward_scalars((long long)(all & 0xffffffffffffull), ((all >> 48) & 0xffull) != 0ull, n, shift, eps, out);
    scalars[0] = out[0]; scalars[1] = out[1]; scalars[2] = out[2]; scalars[3] = out[3];
}

__global__ void __launch_bounds__(TO_SP_THREADS)
k_traj_reward(const float* __restrict__ lo_sum, const int* __restrict__ perm, int64_t n, int64_t npad, float eps, int shift, int prefilled,
              float* __restrict__ rewards, RewardAcc* __restrict__ acc, float* __restrict__ scalars) {
    __shared__ long long lds[2 * TO_SP_WAVES];
    reward_block(lo_sum, perm, n, npad, eps, shift, prefilled, rewards, acc, scalars, blockIdx.x, gridDim.x, blockIdx.y, lds);
}

// rewards + mean + loss (the first nbx * n_traj blocks) and the gradient sums with unit upstream gradient (the other blocks:
// k_traj_pairs' waves) in ONE launch: both need the complete log-odds vector and nothing of each other.
template <bool OCC>
__global__ void __launch_bounds__(TO_SP_THREADS)
k_traj_reward_bwd(const float* __restrict__ lo_sum, int64_t n, float eps, int prefilled, float* __restrict__ rewards,
                  float* __restrict__ scalars, int nbx, SparseArgs a) {
    __shared__ long long lds[2 * TO_SP_WAVES];
    const int R = nbx * a.n_traj;
    if ((int)blockIdx.x < R) {
        reward_block(lo_sum, a.cv.perm, n, a.cv.npad, eps, a.shift, prefilled, rewards, a.acc, scalars, blockIdx.x % nbx, nbx, blockIdx.x / nbx, lds);
        return;
    }
    pair_walk<OCC, false>(a, (int)blockIdx.x - R, (int)gridDim.x - R);   // (the rewards of this very launch: no r (1 - r) left by a fused forward)
}

// thread per body waypoint: rig composition, dL/dt = -R sum dL/dc, dL/dR = sum y (x) dL/dc,
// quaternion chain through the homogeneous form of R and through F.normalize.
// vgrad[v*12 ..] = (sum dL/dc [3], sum y (x) dL/dc [9]); mrow(v) = the 9 floats m[3*i+j] = R_v[j][i].
template <typename MRow>
__device__ void finish_waypoint(int w, const float* __restrict__ vgrad, MRow mrow, const WayCold* __restrict__ cold, int C,
                                const float* __restrict__ rig_q, const float* __restrict__ rig_t, float (&out)[7]) {
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
    for (int j = 0; j < 3; ++j) out[j] = (float)dt[j];
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
    out[3] = (float)((dh[0] - qw * dot) * inv);
    out[4] = (float)((dh[1] - qx * dot) * inv);
    out[5] = (float)((dh[2] - qy * dot) * inv);
    out[6] = (float)((dh[3] - qz * dot) * inv);
}
// the gradient row of body waypoint w to the caller's arrays (either may be NULL: the row is wanted in registers only)
__device__ __forceinline__ void store_grad_row(int w, const float (&o)[7], float* __restrict__ poses_grad, float* __restrict__ quats_grad) {
    if (poses_grad) { poses_grad[3 * w] = o[0]; poses_grad[3 * w + 1] = o[1]; poses_grad[3 * w + 2] = o[2]; }
    if (quats_grad) { quats_grad[4 * w] = o[3]; quats_grad[4 * w + 1] = o[4]; quats_grad[4 * w + 2] = o[5]; quats_grad[4 * w + 3] = o[6]; }
}

struct RecRows {
    const WayRec* rec;
    __device__ const float* operator()(int v) const { return rec[v].m; }
};
struct HotRows {
    const WayHot* hot;
    __device__ const float* operator()(int v) const { return hot[v].m; }
};

// os.mode != 0: the thread goes on with its waypoint's share of the step's epilogue (opt_step.hpp); scalars: the trajectories'
// (mean reward, loss_vis, ...) the finish kernel wrote in the launch before
__global__ void k_traj_bwd_finish2(const float* __restrict__ vgrad, const WayRec* __restrict__ rec,
                                   const WayCold* __restrict__ cold, int W, int C, const float* __restrict__ rig_q,
                                   const float* __restrict__ rig_t, float* __restrict__ poses_grad,
                                   float* __restrict__ quats_grad, OptStep os, const float* __restrict__ scalars) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    float o[7];
    finish_waypoint(w, vgrad, RecRows{rec}, cold, C, rig_q, rig_t, o);
    store_grad_row(w, o, poses_grad, quats_grad);
    if (os.mode) {
        const int b = w / os.n_eval, rr = w - b * os.n_eval;
        const OptPro p = os.pro[b];
        const bool stopped = os.mode == 1 && os.state_in[(int64_t)b * os.state_stride + 2] != 0.f;
        for (int e = 0; e < 7 * os.step; ++e) opt_update_element(os, b, rr, e, o, stopped, p);
        if (os.mode == 1 && rr == 0) opt_commit(os, b, scalars + 4 * b, p);
    }
}

// the ModelPose path keeps its own record type (pose_kernels.hip)
__global__ void k_bwd_finish2(const float* __restrict__ vgrad, const WayHot* __restrict__ hot,
                              const WayCold* __restrict__ cold, int W, int C, const float* __restrict__ rig_q,
                              const float* __restrict__ rig_t, float* __restrict__ poses_grad,
                              float* __restrict__ quats_grad) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    float o[7];
    finish_waypoint(w, vgrad, HotRows{hot}, cold, C, rig_q, rig_t, o);
    store_grad_row(w, o, poses_grad, quats_grad);
}


// block per virtual waypoint: adds the partials of the flagged slots in 16 sums x 64 slot groups — group g takes the slots
// s = g (mod 64), i.e. bit g of every flag word, in ascending order; the 64 group sums are then added in group order (double).
// THREADS = 1024: a thread per (sum, group), the shortest chain (up to a few hundred waypoints).  THREADS = 256: a thread keeps
// four groups (g, g + 16, g + 32, g + 48) — the same 64 sums in the same order, bit for bit — and eight blocks share a CU where
// two 1024-thread blocks made the 1 024 waypoints of eight concurrent trajectories queue (32 -> 13 us) — and the shares of the
// extremal points: the slots recorded by k_traj_sparse (arrival order: sorted here) are re-evaluated by waves 0..3 in ascending
// slot order with a fixed DPP tree, so the result does not depend on any arrival order (torch splits the gradient of min()/max()
// evenly among ties, model.py:226-227).
//   vgrad[v*12 ..] = (sum dL/dc [3], sum y (x) dL/dc [9]) with c = R^T y:  R^T gy,  (y (x) gy) R.
// The sums of a FUSED step were taken with unit dL/d reward: post = 1 scales them by scalars[4*seg+2] * gout[seg] (scalars
// written by k_traj_reward in the same step), post = 2 by the same factor computed here from the integer reward sum
// (RewardAcc::a), which the first block of each trajectory also turns into that trajectory's scalars.
#define TO_PSHARD_HDR 8      // doubles in front of the per-waypoint partials of a point-sharded step
#define TO_PSHARD_NSUM 40    // per virtual waypoint: 14 sums of the flagged pairs, 13 + 13 of the argmin / argmax sets
struct FinishPost {
    const unsigned long long* live;   // culled pass 1: bit (v, slot) = the pair was evaluated (`part` holds it); NULL: all were
    double* partial;           // mode 3: TO_PSHARD_HDR + V x TO_PSHARD_NSUM doubles (this rank's sums: the ranks add them up)
    int mode;                  // 0: none   1: scalars + gout   2: acc + gout (+ scalars_out)   3: stop at the sums (point-sharded step)
    const float* scalars;
    const float* gout;
    const RewardAcc* acc;
    float* scalars_out;
    int64_t n;
    int shift;
    float eps;
    const int* toff;
    int C;
};

template <int THREADS>
__global__ void __launch_bounds__(THREADS)
k_traj_finish(CloudView cv, const float* __restrict__ bpart, int nslots, const unsigned long long* __restrict__ fv, int fv_words,
              const WayRec* __restrict__ rec, const Extrema* __restrict__ ext, EvalK k, const TieRec* __restrict__ ties,
              const float2* __restrict__ part, int V, const uint32_t* __restrict__ occ, int64_t occw, float* __restrict__ vgrad,
              const WayCold* __restrict__ cold, int single, float* __restrict__ poses_grad, float* __restrict__ quats_grad,
              FinishPost post, OptStep os) {
    constexpr int NG = THREADS / 16, PER = 64 / NG;   // groups per pass, groups per thread
    __shared__ double sgrp[64][16];
    __shared__ double stie[2][13];   // [0] argmin set, [1] argmax set: 12 sums + count
    __shared__ int srow[2][TO_TIE_CAP];
    const int v = blockIdx.x, t = threadIdx.x, kq = t & 15, g = t >> 4;
    TO_STAMP(TO_STAMP_FINISH, 0);
    // the step's epilogue (opt_step.hpp) reads nothing this kernel computes: asked for now, used at the end
    OptElem oel;
    OptPro opro;
    bool ostopped = false;
    oel.at = -1;
    if (os.mode && single) {
        const int b = v / os.n_eval;
        if (t < 7 * os.step) oel = opt_elem_load(os, b, v - b * os.n_eval, t);
        opro = os.pro[b];
        ostopped = os.mode == 1 && os.state_in[(int64_t)b * os.state_stride + 2] != 0.f;
    }
    const WayRec& r = rec[v];
    float a, pmax, M, invM;
    load_norm(ext[v], a, pmax, M, invM);
    // the recorded tie slots in ascending order: the tie sums are added in a fixed order (insertion sort of <= 7 entries)
    if (t < 2) {
        const int* tp = reinterpret_cast<const int*>(ties + v);   // TieRec: nmax, nmin, maxrow[7], minrow[7]
        const int cnt = min(t ? tp[0] : tp[1], TO_TIE_CAP);
        const int* src = tp + 2 + (t ? 0 : TO_TIE_CAP);
        int* dst = srow[t];
        for (int i = 0; i < cnt; ++i) {
            const int xv = src[i];
            int j = i - 1;
            while (j >= 0 && dst[j] > xv) { dst[j + 1] = dst[j]; --j; }
            dst[j + 1] = xv;
        }
    }
    // the waypoint's flag words first (one parallel load), then each group's bit of every word
    __shared__ unsigned long long sfv[1024];
    __shared__ double stie4[2][4][13];
    constexpr int FIN_ROWS = 64;
    __shared__ int snzw[1024];
    __shared__ int snz_n;
    __shared__ float svals[THREADS == 1024 ? FIN_ROWS : 1][16];
    const int* tp = reinterpret_cast<const int*>(ties + v);
    // ---- argmin / argmax sets: waves 0..3 take a quarter (64 points) of every recorded slot each ----
    auto tie_sets = [&]() {
        const int wq = t >> 6, ln = t & 63;
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
                for (int q = 0; q < cnt; ++q) do_slot(srow[set][q]);
            } else {
                // more slots hold the extremum than were recorded: walk every slot partial (rare: many exact duplicates)
                for (int rr = 0; rr < nslots; ++rr) {
                    if (post.live != nullptr && !((post.live[(int64_t)v * fv_words + (rr >> 6)] >> (rr & 63)) & 1ull)) continue;   // not evaluated: holds neither
                    const float2 q = part[(int64_t)rr * V + v];
                    const bool hit = set ? (q.y - a == M) : (q.x == a);
                    if (hit) do_slot(rr);
                }
            }
            if (ln == 63)
                for (int j = 0; j < 13; ++j) stie4[set][wq][j] = tot[j];
        }
    };
    if (THREADS == 1024 && fv_words <= 1024) {
        // The common shape: the two jobs run side by side.  Waves 0..3 re-evaluate the tie slots; waves 4..15 add the partials —
        // 1024 (group, column) items on 768 threads, a third of them two — into the same 64 x 16 group sums, each in ascending slot order.
        for (int j = t; j < fv_words; j += THREADS) sfv[j] = fv[(int64_t)v * fv_words + j];
        __syncthreads();   // the flag words and the sorted tie slots
        TO_STAMP(TO_STAMP_FINISH, 1);
        // The flagged slots' rows of partials: with up to FIN_ROWS of them (one word chunk of flag bits) every row is requested at
        // once — row j of the ascending list to svals[j] — and the ordered sums below read LDS; a thread that loaded its group's
        // rows one after the other paid a memory round trip per row (4.6 of the kernel's 8 us at three or four rows per group).
        int pre = 0, nflag = 0;   // lane = word: flagged slots in the words before it; all of them
        unsigned long long myword = 0ull, nz = 0ull;   // lane = word: its flag bits; the words that have any
        if (fv_words <= 64) {
            const int lane = t & 63;
            myword = lane < fv_words ? sfv[lane] : 0ull;
            nz = __ballot(myword != 0ull);
            const int c = __popcll(myword);
            int incl = c;
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                const int up = __shfl_up(incl, sh);
                if (lane >= sh) incl += up;
            }
            pre = incl - c;
            nflag = __shfl(incl, 63);
        }
        // (many rows — a dense indoor cloud flags a few hundred slots per waypoint — are cheaper added from memory by their own
        // (group, column) threads: every thread has several in flight)
        const bool rows_in_lds = fv_words <= 64 && nflag <= FIN_ROWS;   // block-uniform
        if (t < 256) {
            tie_sets();
        } else if (rows_in_lds) {
            // waves 4..15: thread i takes column i % 16 of row i / 16 of the ascending list (then i + 768, ...): every load of the
            // block is in flight at once
            const int lane = t & 63;
            for (int i0 = (t - 256) & ~63; i0 < nflag * 16; i0 += THREADS - 256) {   // wave-uniform: the ballots below want every lane
                const int i = i0 + lane;
                const bool valid = i < nflag * 16;
                const int j = valid ? i >> 4 : nflag - 1, kcol = i & 15;
                // row j: the word whose prefix is the last one <= j, then its (j - prefix)-th set bit.  Four rows to a wave: four
                // ballots (the row is wave-uniform per quarter: ask per quarter)
                int w = 0;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int jq = __shfl(j, qd * 16);
                    const int wq = __popcll(__ballot(lane < fv_words && pre <= jq)) - 1;   // pre[0] = 0 <= jq
                    if ((lane >> 4) == qd) w = wq;
                }
                unsigned long long w2 = sfv[w];
                for (int r = j - __shfl(pre, w); r > 0; --r) w2 &= w2 - 1ull;
                if (valid) svals[j][kcol] = bpart[((int64_t)v * nslots + (w * 64 + __builtin_ctzll(w2))) * 16 + kcol];
            }
        } else {
            const int i0 = t - 256, i1 = i0 < 256 ? 768 + i0 : -1;
            const int g0 = i0 >> 4, k0 = i0 & 15, g1 = i1 >> 4, k1 = i1 & 15;
            double acc0 = 0.0, acc1 = 0.0;
            auto add_word = [&](int w, unsigned long long word) {
                if ((word >> g0) & 1ull) acc0 += (double)bpart[((int64_t)v * nslots + (w * 64 + g0)) * 16 + k0];
                if (i1 >= 0 && ((word >> g1) & 1ull)) acc1 += (double)bpart[((int64_t)v * nslots + (w * 64 + g1)) * 16 + k1];
            };
            if (fv_words <= 64) {   // the words that hold a flag, from the lanes that hold them (no LDS round trip per word)
                for (unsigned long long m = nz; m; m &= m - 1ull) {
                    const int w = __builtin_ctzll(m);
                    add_word(w, (unsigned long long)__shfl((long long)myword, w));
                }
            } else {
                for (int w = 0; w < fv_words; ++w) {
                    const unsigned long long word = sfv[w];
                    if (word != 0ull) add_word(w, word);
                }
            }
            sgrp[g0][k0] = acc0;
            if (i1 >= 0) sgrp[g1][k1] = acc1;
        }
        if (rows_in_lds) {
            __syncthreads();   // the rows
            // thread (group g, column kq): its group's rows in ascending slot order, as before, from LDS
            double acc0 = 0.0;
            for (unsigned long long m = nz; m; m &= m - 1ull) {   // the words that hold a flag, in ascending order (a few): a walk over all
                const int w = __builtin_ctzll(m);                // words is a chain of as many LDS round trips
                const unsigned long long word = (unsigned long long)__shfl((long long)myword, w);
                const int pw = __shfl(pre, w);                   // (every lane takes part: m is the same for all)
                if ((word >> g) & 1ull) acc0 += (double)svals[pw + __popcll(word & ((1ull << g) - 1ull))][kq];
            }
            sgrp[g][kq] = acc0;
        }
    } else {
        double acc[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) acc[j] = 0.0;
        for (int w0 = 0; w0 < fv_words; w0 += 1024) {
            const int nw = min(1024, fv_words - w0);
            __syncthreads();
            for (int j = t; j < nw; j += THREADS) sfv[j] = fv[(int64_t)v * fv_words + w0 + j];
            __syncthreads();
            // the words that hold a flag, in ascending order (a few of many: a walk over all of them is a chain of as many LDS round trips)
            if (t < 64) {
                int cnt = 0;
                for (int c0 = 0; c0 < nw; c0 += 64) {
                    const bool has = c0 + t < nw && sfv[c0 + t] != 0ull;
                    const unsigned long long m = __ballot(has);
                    if (has) snzw[cnt + __popcll(m & ((1ull << t) - 1ull))] = c0 + t;
                    cnt += __popcll(m);
                }
                if (t == 0) snz_n = cnt;
            }
            __syncthreads();
            const int nnz = snz_n;
            for (int i = 0; i < nnz; ++i) {
                const int w = snzw[i];
                const unsigned long long word = sfv[w];
#pragma unroll
                for (int j = 0; j < PER; ++j)
                    if ((word >> (g + NG * j)) & 1ull) acc[j] += (double)bpart[((int64_t)v * nslots + ((w0 + w) * 64 + g + NG * j)) * 16 + kq];
            }
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) sgrp[g + NG * j][kq] = acc[j];
        __syncthreads();   // srow
        if (t < 256) tie_sets();
    }
    TO_STAMP(TO_STAMP_FINISH, 2);   // wave 0: the tie slots re-evaluated
    __syncthreads();
    TO_STAMP(TO_STAMP_FINISH, 3);   // every wave: tie slots, rows, group sums
    if (t < 26) {   // ascending slot order inside a quarter, quarters in order: a fixed summation order
        const int set = t / 13, j = t % 13;
        stie[set][j] = ((stie4[set][0][j] + stie4[set][1][j]) + stie4[set][2][j]) + stie4[set][3][j];
    }
    __shared__ double stot[16];
    __shared__ float s_sc[4];   // the trajectory's scalars (the step's epilogue wants them)
    if (post.mode == 3) {
        // a point-sharded step: this rank holds a part of the cloud, so these are PARTIAL sums — everything below (the factor
        // dL/d reward from the mean of ALL rewards, the tie sets' shares, the chain) is linear in them or needs the other ranks'
        // too: they leave here, the ranks add them up (one all-reduce), k_traj_pshard_final goes on
        __syncthreads();   // stie
        double* dst = post.partial + TO_PSHARD_HDR + (int64_t)v * TO_PSHARD_NSUM;
        if (t < 14) {
            double q = 0.0;
            for (int gg = 0; gg < 64; ++gg) q += sgrp[gg][t];
            dst[t] = q;
        } else if (t < 40) {
            dst[t] = stie[(t - 14) / 13][(t - 14) % 13];
        }
        if (v == 0 && t == 64) {   // the rank's reward sum (fixed point, exact in a double), NaN mark and point count
            long long sfix = 0;
            unsigned nan = 0;
            for (int i = 0; i < 8; ++i) { sfix += post.acc->a[i].sum; nan |= post.acc->a[i].nan; }
            post.partial[0] = (double)((long long)cv.n * (1ll << (post.shift - 1)) + sfix);
            post.partial[1] = nan ? 1.0 : 0.0;
            post.partial[2] = (double)cv.n;
            for (int i = 3; i < TO_PSHARD_HDR; ++i) post.partial[i] = 0.0;
        }
        return;
    }
    if (t < 16) {
        double q = 0.0;
        for (int gg = 0; gg < 64; ++gg) q += sgrp[gg][t];
        // sums taken with unit upstream gradient get the trajectory's dL/d reward here: they are linear in it
        if (post.mode == 1) q *= (double)(post.scalars[4 * r.seg + 2] * post.gout[r.seg]);
        if (post.mode == 2) {
            float sc[4];
            reward_scalars_from_a(post.acc + r.seg, post.n, post.shift, post.eps, sc);
            q *= (double)(sc[2] * (post.gout ? post.gout[r.seg] : 1.0f));
            const int v_first = post.toff ? post.toff[r.seg] * post.C : 0;
            if (t == 0) { s_sc[0] = sc[0]; s_sc[1] = sc[1]; s_sc[2] = sc[2]; s_sc[3] = sc[3]; }
            if (t == 0 && v == v_first && post.scalars_out) {
                float* so = post.scalars_out + 4 * r.seg;
                so[0] = sc[0]; so[1] = sc[1]; so[2] = sc[2]; so[3] = sc[3];
            }
        }
        stot[t] = q;
    }
    __syncthreads();
    TO_STAMP(TO_STAMP_FINISH, 4);   // the 64 groups added, the factor dL/d reward
    __shared__ float sgy[12];
    if (t < 12) {
        const double nmin = stie[0][12], nmax = stie[1][12];
        const double wmin = nmin > 0.0 ? stot[12] / nmin : 0.0;
        const double wmax = nmax > 0.0 ? stot[13] / nmax : 0.0;
        sgy[t] = (float)(stot[t] + wmin * stie[0][t] + wmax * stie[1][t]);   // sum gy [3], sum y (x) gy [9], world-aligned
        // a degenerate waypoint (max == min, or a NaN among its p: the reference's p_hat is 0/0 or NaN for EVERY point, model.py:227):
        // autograd's gradient through it is NaN in every entry, like the rewards
        if (!(M > 0.f) || !(invM < INFINITY)) sgy[t] = __builtin_nanf("");
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
        __shared__ float s_vis[8];
        if (t == 0) {
            float o[7];
            finish_waypoint(v, vgrad, RecRows{rec}, cold, 1, nullptr, nullptr, o);
            store_grad_row(v, o, poses_grad, quats_grad);
            for (int kk = 0; kk < 7; ++kk) s_vis[kk] = o[kk];
        }
        if (os.mode) {
            // the step's epilogue for this waypoint's rows (opt_step.hpp): full gradient, and in an optimisation step Adam; the
            // block of a trajectory's first waypoint also writes the loss log and the next row of the early-stop state
            __syncthreads();
            const int b = v / os.n_eval, rr = v - b * os.n_eval;
            opt_elem_apply(os, oel, s_vis, ostopped, opro);
            for (int e = t + THREADS; e < 7 * os.step; e += THREADS) opt_update_element(os, b, rr, e, s_vis, ostopped, opro);   // (a stride beyond 36 waypoints)
            if (os.mode == 1 && rr == 0 && t == 0) opt_commit(os, b, s_sc, opro);
        }
    }
    TO_STAMP(TO_STAMP_FINISH, 5);
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

// rows from per-point visibility values instead of index lists: waypoint w's kept point j (kept_idx[w][j], j < kept_count[w]) is
// hidden when visible[seg_off[w] + j] == 0 — the mask a batched hull pass writes over the waypoints' kept points laid end to end,
// or the batched z-buffer's (seg_off[w] = w x n).  A waypoint with fewer than min_points kept points keeps its row of ones.
__global__ void k_occ_rows_masked(const int* __restrict__ inv, const int32_t* __restrict__ kept_idx, int64_t n,
                                  const int32_t* __restrict__ kept_count, const float* __restrict__ visible,
                                  const int64_t* __restrict__ seg_off, int min_points, uint32_t* __restrict__ rows, int64_t roww) {
    const int w = blockIdx.y;
    const int m = kept_count[w];
    if (m < min_points) return;
    const int32_t* kk = kept_idx + (int64_t)w * n;
    const float* vis = visible + seg_off[w];
    uint32_t* row = rows + (int64_t)w * roww;
    const int stride = gridDim.x * blockDim.x;
    if (inv != nullptr) {
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
            if (vis[j] == 0.f) {
                const int s = inv[kk[j]];
                atomicAnd(&row[s >> 5], ~(1u << (s & 31)));
            }
        }
        return;
    }
    // inv == NULL: kept_idx holds positions in the packed cloud's order already (the cull ran over the sorted points), ascending —
    // the hidden bits of one word sit on neighbouring lanes: OR-ed along their run by a segmented scan, one atomic per run instead of
    // one per hidden point (12 M scattered atomics, 0.6 ms of a 128-waypoint refresh, until r06)
    const int lane = threadIdx.x & 63;
    for (int j0 = blockIdx.x * blockDim.x; j0 < m; j0 += stride) {   // (all lanes of a wave take part)
        const int j = j0 + threadIdx.x;
        int word = -1 - lane;   // (distinct negative keys: lanes without a hidden point join nobody's run)
        unsigned bits = 0u;
        if (j < m && vis[j] == 0.f) { const int s = kk[j]; word = s >> 5; bits = 1u << (s & 31); }
        for (int d = 1; d < 64; d <<= 1) {   // runs are at most 32 lanes long, but need not be aligned
            const unsigned ob = (unsigned)__shfl_down((int)bits, d);
            const int ow = __shfl_down(word, d);
            if (lane + d < 64 && ow == word) bits |= ob;
        }
        const int pw = __shfl_up(word, 1);
        if (word >= 0 && (lane == 0 || pw != word)) atomicAnd(&row[word], ~bits);
    }
}

extern "C" int tohip_occlusion_rows_masked(int64_t n, const int32_t* inv_perm, const int32_t* kept_idx, const int32_t* kept_count,
                                           const float* visible, const int64_t* seg_off, int32_t min_points, int64_t n_wps, uint32_t* rows,
                                           void* stream_) {
    if (!kept_idx || !kept_count || !visible || !seg_off || !rows || n <= 0 || n_wps <= 0 || n_wps > 65535) return TOHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    const int64_t npad = tohip_padded_points(n);
    const int64_t roww = npad / 32;
    hipError_t e = hipMemsetAsync(rows, 0xff, (size_t)roww * (size_t)n_wps * sizeof(uint32_t), st);
    if (e != hipSuccess) return (int)e;
    int nb = (int)((n + 255) / 256);
    if (nb > 256) nb = 256;
    k_occ_rows_masked<<<dim3((unsigned)nb, (unsigned)n_wps), 256, 0, st>>>(inv_perm, kept_idx, n, kept_count, visible, seg_off, min_points, rows, roww);
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
    int nblk;      // culled pass-1 point blocks (1024 points each)
    int nslots;    // npad / 256
    int ncbits;    // trajectories x fv_words: the words of the candidate (slot, trajectory) bits
    int fv_words;  // (nslots + 63) / 64
    int vwords;    // (V + 63) / 64
    int V;
    int64_t plcap; // entries of one pair sub-list
    size_t off_ctl, off_toff, off_rec, off_cold, off_ext, off_cbits, off_ctr, off_part, off_fv, off_live, off_plist, off_ties, off_bpart, off_vgrad, off_unit, total;
};

inline TrajPlan make_plan(int64_t n, int64_t V, int64_t W, int64_t n_traj = 1) {
    TrajPlan p;
    p.npad = tohip_padded_points(n);
    p.nblk = (int)(p.npad / (TO_BLOCK * TO_P));
    p.nslots = (int)(p.npad / TO_SLOT);
    p.fv_words = (p.nslots + 63) / 64;
    p.ncbits = (int)((int64_t)p.fv_words * (n_traj < 1 ? 1 : n_traj));
    p.vwords = (int)((V + 63) / 64);
    p.V = (int)V;
    size_t o = 0;
    p.off_ctl = o;   o += sizeof(RewardAcc) * (size_t)(n_traj < 1 ? 1 : n_traj);   // first: all tohip_traj_reward uses
    p.off_toff = o;  o += align_up(sizeof(int) * (size_t)((n_traj < 1 ? 1 : n_traj) + 1), 256);   // the forward's copy of the trajectory offsets
    p.off_rec = o;   o += align_up((size_t)V * sizeof(WayRec), 256);
    (void)W;
    p.off_cold = o;  o += align_up((size_t)V * sizeof(WayCold), 256);   // (one per BODY waypoint; sized by V so that the layout depends on (n, V, n_traj) only)
    p.off_ext = o;   o += align_up((size_t)V * sizeof(Extrema), 256);
    p.off_cbits = o; o += align_up((size_t)p.ncbits * TO_CBIT_STRIDE * sizeof(unsigned long long), 256);   // a bit per (trajectory, slot): pass 1's candidates
    p.off_ctr = o;   o += (size_t)TO_PL_SHARDS * TO_PL_STRIDE * sizeof(int);                // the pair sub-lists' lengths, one to a 128-byte line
    p.off_part = o;  o += align_up((size_t)V * (size_t)p.nslots * sizeof(float2), 256);
    p.off_fv = o;    o += align_up((size_t)V * (size_t)p.fv_words * sizeof(unsigned long long), 256);
    p.off_live = o;  o += align_up((size_t)V * (size_t)p.fv_words * sizeof(unsigned long long), 256);   // culled pass 1: the slots a waypoint can reach, as bits
    p.plcap = (int64_t)((p.nslots + TO_PL_SHARDS - 1) / TO_PL_SHARDS) * V;   // a sub-list's capacity: every pair of its slots flagged
    p.off_plist = o; o += align_up((size_t)p.plcap * TO_PL_SHARDS * sizeof(int2), 256);
    p.off_ties = o;  o += align_up((size_t)V * sizeof(TieRec), 256);
    p.off_bpart = o; o += align_up((size_t)V * (size_t)p.nslots * 16 * sizeof(float), 256);
    p.off_vgrad = o; o += align_up((size_t)V * 12 * sizeof(float), 256);
    p.off_unit = o;  o += align_up((size_t)n_traj * (size_t)p.npad * sizeof(float), 256);   // r (1 - r) of the candidate slots' points (sparse -> pairs)
    p.total = o;
    return p;
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

// everything the launches of one step share
struct TrajStep {
    hipStream_t st;
    TrajPlan pl;
    CloudView cv;
    EvalK k;
    int C;
    int64_t n, W, V, n_traj;
    const float *rq, *rt;
    const int* toff;     // trajectory offsets (NULL: one trajectory)
    int* toff_ws;        // their copy in the workspace: the calls after the forward take none
    const uint32_t* occ;
    int64_t occw;
    bool cull;
    int max_traj_v = 0;  // several trajectories: the virtual waypoints of the longest one when the host knows (0: it does not)
    int wp_stride = 1;   // rows between two evaluated waypoints in the arrays handed to the probe
    int traj_rows = 0;   // > 0: rows every trajectory owns in those arrays (its evaluated waypoints are rows 0, wp_stride, ... of its own)
    bool lean = false;   // an optimisation step that is not the run's last: lo_sum and rewards are nobody's to read (TOHIP_TRAJ_OPT_LAST_OUTPUTS)
    OptStep opt = OptStep{};   // mode != 0: the step's prologue / epilogue ride in the sparse kernel's, the pairs' and the finish launches (opt_step.hpp)
    float* opt_scalars = nullptr;   // opt.mode == 2: model()'s scalars (the extra block of the pairs' launch writes them)
    RewardAcc* acc;
    WayRec* rec;
    WayCold* cold;
    Extrema* ext;
    unsigned long long* cbits;
    int* ctr;
    int2* plist;
    float2* part;
    unsigned long long *fv, *live;
    TieRec* ties;
    float *bpart, *vgrad;
    float* unit;
    int shift;
};

inline int traj_step_init(TrajStep& s, const void* packed, int64_t n, int64_t W, int64_t n_traj, const int32_t* traj_offsets,
                          const tohip_camera* cam, const tohip_rig* rig, int flags, const uint32_t* occ, void* workspace,
                          size_t workspace_bytes, void* stream_, bool forward) {
    if (!packed || !cam || !workspace || n <= 0 || W <= 0 || n_traj <= 0 || n_traj > 65535 || (forward && n_traj > 1 && !traj_offsets)) return TOHIP_EINVAL;
    s.st = (hipStream_t)stream_;
    s.C = rig_cams(rig);
    s.n = n; s.W = W; s.V = W * s.C; s.n_traj = n_traj;
    if (s.V > 64 * TO_SP_MAXW || (int64_t)(tohip_padded_points(n) / TO_SLOT / 64 + 1) * n_traj > (int64_t)1 << 30) return TOHIP_EINVAL;
    s.pl = make_plan(n, s.V, W, n_traj);
    if (workspace_bytes < s.pl.total) return TOHIP_ENOSPC;
    char* ws = (char*)workspace;
    s.acc = (RewardAcc*)(ws + s.pl.off_ctl);
    s.rec = (WayRec*)(ws + s.pl.off_rec);
    s.cold = (WayCold*)(ws + s.pl.off_cold);
    s.ext = (Extrema*)(ws + s.pl.off_ext);
    s.cbits = (unsigned long long*)(ws + s.pl.off_cbits);
    s.ctr = (int*)(ws + s.pl.off_ctr);
    s.plist = (int2*)(ws + s.pl.off_plist);
    s.live = (unsigned long long*)(ws + s.pl.off_live);
    s.part = (float2*)(ws + s.pl.off_part);
    s.fv = (unsigned long long*)(ws + s.pl.off_fv);
    s.ties = (TieRec*)(ws + s.pl.off_ties);
    s.bpart = (float*)(ws + s.pl.off_bpart);
    s.vgrad = (float*)(ws + s.pl.off_vgrad);
    s.unit = (float*)(ws + s.pl.off_unit);
    s.k = make_evalk(cam);
    s.cv = cloud_view(packed, n);
    s.cull = !(flags & TOHIP_TRAJ_DENSE) && s.pl.fv_words <= TO_PROBE_MAXFW;   // (beyond 16.7 M points every pair is evaluated)
    s.wp_stride = ((flags >> 8) & 0xffff) + 1;   // TOHIP_TRAJ_STRIDE(step): the evaluated waypoints are every step-th row of poses / quats
    if (s.wp_stride > 1 && n_traj > 1) return TOHIP_EINVAL;   // (several trajectories read in place: tohip_traj_opt_step)
    s.rq = (s.C > 1 || (rig && rig->rig_quats)) ? rig->rig_quats : nullptr;
    s.rt = s.rq ? rig->rig_trans : nullptr;
    s.toff_ws = (int*)(ws + s.pl.off_toff);
    s.toff = n_traj > 1 ? (forward ? traj_offsets : s.toff_ws) : nullptr;
    s.occ = occ;
    s.occw = s.cv.npad / 32;
    s.shift = reward_shift(n);
    return TOHIP_OK;
}

// launches 1 and 2 of a step: records + probe, pass 1
inline int launch_probe_pass1(const TrajStep& s, const float* poses, const float* quats, float* lo_sum, float* rewards_half) {
    const int V = (int)s.V;
    const OutInit oi{s.lean ? nullptr : lo_sum, s.lean ? nullptr : rewards_half, s.cv.npad, s.n, (int)s.n_traj};
    {
        TO_PROF(TOHIP_PROF_PROBE, s.st);
        const ProbeCull pc{s.cull ? 1 : 0, s.pl.nslots, s.live};
        const int grid = V;
        if (V <= 512)
            k_traj_probe<1024><<<grid, 1024, 0, s.st>>>(s.cv, poses, quats, s.C, s.rq, s.rt, s.k, s.rec, s.cold, s.ext, s.ties, s.occ, s.occw, s.fv,
                                                        s.pl.fv_words, s.acc, s.toff, s.toff_ws, (int)s.n_traj, s.cbits, s.pl.ncbits, s.ctr, s.wp_stride,
                                                        s.traj_rows, pc, V);
        else
            k_traj_probe<256><<<grid, 256, 0, s.st>>>(s.cv, poses, quats, s.C, s.rq, s.rt, s.k, s.rec, s.cold, s.ext, s.ties, s.occ, s.occw, s.fv,
                                                      s.pl.fv_words, s.acc, s.toff, s.toff_ws, (int)s.n_traj, s.cbits, s.pl.ncbits, s.ctr, s.wp_stride,
                                                      s.traj_rows, pc, V);
        TO_HIP_CHECK_LAUNCH();
    }
    {
        TO_PROF(TOHIP_PROF_PASS1, s.st);
        const bool occ = s.occ != nullptr;
        if (s.cull) {
            // blocks of four waves, up to eight to a row: a waypoint's reachable slots are evaluated by 32 waves on up to eight CUs.
            // (Two 16-wave blocks to a row kept a heavy row — 800 reachable slots in a 10 m room, 100 on the slab — on the eight
            // SIMDs of two CUs: 17 us of instruction issue there while the light rows' CUs idled.)
            static const int cull_waves = [] { const char* e = getenv("TOHIP_CULL_WAVES"); return e && atoi(e) == 16 ? 16 : 4; }();
            if (cull_waves == 16) {
                const dim3 grid(V <= 256 ? 2 : 1, V);
                if (occ) k_traj_pass1_cull<true, 16><<<grid, 1024, 0, s.st>>>(s.cv, s.rec, V, s.k, s.part, s.ext, s.cbits, s.pl.fv_words, s.live, s.occ, s.occw, oi);
                else k_traj_pass1_cull<false, 16><<<grid, 1024, 0, s.st>>>(s.cv, s.rec, V, s.k, s.part, s.ext, s.cbits, s.pl.fv_words, s.live, s.occ, s.occw, oi);
            } else {
                const dim3 grid(std::max(1, std::min(8, 1024 / V)), V);
                if (occ) k_traj_pass1_cull<true, 4><<<grid, 256, 0, s.st>>>(s.cv, s.rec, V, s.k, s.part, s.ext, s.cbits, s.pl.fv_words, s.live, s.occ, s.occw, oi);
                else k_traj_pass1_cull<false, 4><<<grid, 256, 0, s.st>>>(s.cv, s.rec, V, s.k, s.part, s.ext, s.cbits, s.pl.fv_words, s.live, s.occ, s.occw, oi);
            }
        } else {
            const int nblk8 = (int)(s.pl.npad / (TO_BLOCK * TO_PD));
            const int nb = dense_blocks(nblk8, V, occ);
            if (occ) k_traj_pass1_dense<true><<<nb, TO_BLOCK, 0, s.st>>>(s.cv, s.rec, V, nblk8, s.k, s.part, s.ext, s.cbits, s.pl.fv_words, s.occ, s.occw, oi, clock_stamps());
            else k_traj_pass1_dense<false><<<nb, TO_BLOCK, 0, s.st>>>(s.cv, s.rec, V, nblk8, s.k, s.part, s.ext, s.cbits, s.pl.fv_words, s.occ, s.occw, oi, clock_stamps());
        }
        TO_HIP_CHECK_LAUNCH();
    }
    return TOHIP_OK;
}

inline SparseArgs sparse_args(const TrajStep& s, float* lo_sum) {
    SparseArgs a;
    a.cv = s.cv; a.rec = s.rec; a.ext = s.ext; a.k = s.k; a.part = s.part;
    a.V = (int)s.V; a.nslots = s.pl.nslots; a.vwords = s.pl.vwords; a.fv_words = s.pl.fv_words; a.cbits = s.cbits; a.live = s.cull ? s.live : nullptr; a.plist = s.plist; a.npairs = s.ctr; a.plcap = s.pl.plcap;
    a.fv = s.fv; a.ties = s.ties; a.lo_sum = lo_sum; a.minmax = nullptr; a.occ = s.occ; a.occw = s.occw;
    a.toff = s.toff; a.n_traj = (int)s.n_traj; a.C = s.C;
    a.rewards = nullptr; a.prefilled = 0; a.acc = s.acc; a.shift = s.shift;
    a.grad_rewards = nullptr; a.scalars = nullptr; a.gout = nullptr; a.bpart = s.bpart; a.unit = s.unit;
    return a;
}

// list walkers: the expected number of candidate slots on the workloads this is tuned for (6-8 % of the slots), each a chain
// of its own; a dense cloud lists every slot and the blocks loop
// 4-wave blocks, four to a CU (all resident): a candidate slot is a chain of dependent accesses of ~9 us whatever the block's
// shape, and a 16-wave block holds a whole CU for it — 1 441 candidates (the BASELINE trajectory after a hundred optimiser steps)
// were 5.6 rounds of 256 such blocks, 28 us.  Sixteen-wave blocks remain for experiments (TOHIP_SPARSE_WAVES=16).
inline int sparse_waves() {
    static const int nw = [] { const char* e = getenv("TOHIP_SPARSE_WAVES"); const int v = e ? atoi(e) : 0; return v == 16 ? 16 : 4; }();
    return nw;
}
// one 16-wave block per CU, all resident: the pairs are dealt to the grid's waves
inline int pair_blocks() {
    static const int cus = [] {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) return prop.multiProcessorCount;
        return 256;
    }();
    return cus;
}

// Do the flag words of any ONE trajectory fit the small array (8 words: 36 KB of LDS, four 4-wave blocks to a CU instead of three)?
// One trajectory starts at word 0; one of several may start anywhere inside a word.  max_traj_v = 0: the lengths are on the device
// only (the generic tohip_*_multi calls), and the blocks hold the large array.
inline bool sparse_small_flags(const TrajStep& s) {
    if (s.n_traj == 1) return s.V <= 8 * 64;
    return s.max_traj_v > 0 && s.max_traj_v <= 7 * 64;
}
inline int sparse_blocks(const TrajStep& s, int nw) {
    // as many blocks as are RESIDENT at once — four 4-wave blocks to a CU with the small flag array, three with the large one, two of
    // sixteen waves — shared evenly by the trajectories: a block walks its trajectory's candidates q, q + S, ..., and a block that
    // waits for a CU starts its walk when the others are half way through theirs (r04 launched 1 024 blocks of which 768 were
    // resident for eight trajectories: 72 us where 55 do)
    const int64_t resident = (int64_t)pair_blocks() * (nw == 16 ? 2 : (sparse_small_flags(s) ? 4 : 3));
    const int64_t per_traj = std::max<int64_t>(1, std::min<int64_t>(s.pl.nslots, resident / s.n_traj));
    return (int)(per_traj * s.n_traj);
}

template <int MODE, bool OCC, int NW>
inline void launch_sparse_nw(const TrajStep& s, const SparseArgs& a) {
    const int grid = sparse_blocks(s, NW) + (s.opt.mode ? s.opt.n_traj : 0);   // (+ the step's prologue blocks)
    if (sparse_small_flags(s)) k_traj_sparse<MODE, OCC, NW, 8><<<grid, NW * 64, 0, s.st>>>(a, s.opt);
    else k_traj_sparse<MODE, OCC, NW, TO_SP_MAXW><<<grid, NW * 64, 0, s.st>>>(a, s.opt);
}

template <int MODE>
inline int launch_sparse(const TrajStep& s, const SparseArgs& a) {
    const bool w16 = sparse_waves() == 16;
    if (s.occ) { if (w16) launch_sparse_nw<MODE, true, 16>(s, a); else launch_sparse_nw<MODE, true, 4>(s, a); }
    else { if (w16) launch_sparse_nw<MODE, false, 16>(s, a); else launch_sparse_nw<MODE, false, 4>(s, a); }
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

inline int launch_pairs(const TrajStep& s, const SparseArgs& a) {
    const int grid = pair_blocks() + (s.opt.mode == 2 ? 1 : 0);
    const bool unit = a.unit != nullptr && a.grad_rewards == nullptr && a.scalars == nullptr;   // the callers with a unit upstream follow a fused forward
    if (s.occ) {
        if (unit) k_traj_pairs<true, true><<<grid, TO_SP_THREADS, 0, s.st>>>(a, s.opt, s.opt_scalars);
        else k_traj_pairs<true, false><<<grid, TO_SP_THREADS, 0, s.st>>>(a, s.opt, s.opt_scalars);
    } else {
        if (unit) k_traj_pairs<false, true><<<grid, TO_SP_THREADS, 0, s.st>>>(a, s.opt, s.opt_scalars);
        else k_traj_pairs<false, false><<<grid, TO_SP_THREADS, 0, s.st>>>(a, s.opt, s.opt_scalars);
    }
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

inline int launch_finish(const TrajStep& s, const FinishPost& post, float* poses_grad, float* quats_grad) {
    TO_PROF(TOHIP_PROF_FINISH, s.st);
    const bool single = s.C == 1 && s.rq == nullptr;
    const int V = (int)s.V;
    if (V <= 512)
        k_traj_finish<1024><<<V, 1024, 0, s.st>>>(s.cv, s.bpart, s.pl.nslots, s.fv, s.pl.fv_words, s.rec, s.ext, s.k, s.ties, s.part, V, s.occ, s.occw,
                                                  s.vgrad, s.cold, single ? 1 : 0, poses_grad, quats_grad, post, s.opt);
    else
        k_traj_finish<256><<<V, 256, 0, s.st>>>(s.cv, s.bpart, s.pl.nslots, s.fv, s.pl.fv_words, s.rec, s.ext, s.k, s.ties, s.part, V, s.occ, s.occw,
                                                s.vgrad, s.cold, single ? 1 : 0, poses_grad, quats_grad, post, s.opt);
    TO_HIP_CHECK_LAUNCH();
    if (!single && post.mode != 3) {
        k_traj_bwd_finish2<<<(int)((s.W + 63) / 64), 64, 0, s.st>>>(s.vgrad, s.rec, s.cold, (int)s.W, s.C, s.rq, s.rt, poses_grad, quats_grad, s.opt,
                                                                    post.scalars_out);
        TO_HIP_CHECK_LAUNCH();
    }
    return TOHIP_OK;
}

inline FinishPost finish_post(const TrajStep& s, int mode, const float* scalars, const float* gout, float* scalars_out, float eps) {
    FinishPost p;
    p.live = s.cull ? s.live : nullptr;
    p.partial = nullptr;
    p.mode = mode; p.scalars = scalars; p.gout = gout; p.acc = s.acc; p.scalars_out = scalars_out; p.n = s.n; p.shift = s.shift; p.eps = eps;
    p.toff = s.toff; p.C = s.C;
    return p;
}

inline int reward_blocks(int64_t n) {
    int64_t nb = (n + 4 * TO_SP_THREADS - 1) / (4 * TO_SP_THREADS);
    return (int)(nb > TO_REWARD_BLOCKS ? TO_REWARD_BLOCKS : nb);
}

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
    if (!poses || !quats || !lo_sum || !minmax) return TOHIP_EINVAL;
    TrajStep s;
    int rc = traj_step_init(s, packed, n, W, n_traj, traj_offsets, cam, rig, flags, occlusion_bits, workspace, workspace_bytes, stream_, true);
    if (rc != TOHIP_OK) return rc;
    rc = launch_probe_pass1(s, poses, quats, lo_sum, rewards_half);
    if (rc != TOHIP_OK) return rc;
    TO_PROF(TOHIP_PROF_PASS2, s.st);
    SparseArgs a = sparse_args(s, lo_sum);
    a.minmax = minmax;
    return launch_sparse<TO_SP_FWD>(s, a);
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
    if (workspace_bytes < sizeof(RewardAcc) * (size_t)n_traj) return TOHIP_ENOSPC;
    RewardAcc* acc = (RewardAcc*)workspace;   // TrajPlan::off_ctl == 0
    const CloudView cv = cloud_view(packed, n);
    TO_PROF(TOHIP_PROF_REWARD, st);
    k_traj_reward<<<dim3(reward_blocks(n), (unsigned)n_traj), TO_SP_THREADS, 0, st>>>(lo_sum, cv.perm, n, cv.npad, eps, reward_shift(n), prefilled ? 1 : 0,
                                                                                 rewards, acc, scalars);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_traj_reward(const void* packed, const float* lo_sum, int64_t n, float eps, int prefilled, float* rewards,
                                 float* scalars, void* workspace, size_t workspace_bytes, void* stream_) {
    return tohip_traj_reward_multi(packed, lo_sum, n, 1, eps, prefilled, rewards, scalars, workspace, workspace_bytes, stream_);
}

namespace {
// backward of a split step (a complete lo_sum comes in); with `fused` also the rewards, their mean and the loss scalars
// (tohip_traj_reward's work) in the backward's first launch.  phases: 1 = the pair sums (with `fused`: + rewards), 2 = the finish.
struct FusedReward { float eps; int prefilled; float* rewards; float* scalars; };
int traj_backward_impl(const void* packed, int64_t n, int64_t W, int64_t n_traj, const tohip_camera* cam,
                       const tohip_rig* rig, int flags, const uint32_t* occlusion_bits, float* lo_sum,
                       const float* grad_rewards, const float* scalars, const float* gout, float* poses_grad,
                       float* quats_grad, void* workspace, size_t workspace_bytes, void* stream_, const FusedReward* fused) {
    if (!lo_sum || !poses_grad || !quats_grad || (!grad_rewards && (!scalars || !gout))) return TOHIP_EINVAL;
    TrajStep s;
    int rc = traj_step_init(s, packed, n, W, n_traj, nullptr, cam, rig, flags, occlusion_bits, workspace, workspace_bytes, stream_, false);
    if (rc != TOHIP_OK) return rc;
    SparseArgs a = sparse_args(s, lo_sum);
    if (fused) {
        TO_PROF(TOHIP_PROF_BWD, s.st);
        const int nbx = reward_blocks(n);
        const int64_t blocks = (int64_t)nbx * n_traj + pair_blocks();
        if (s.occ) k_traj_reward_bwd<true><<<(int)blocks, TO_SP_THREADS, 0, s.st>>>(lo_sum, n, fused->eps, fused->prefilled ? 1 : 0, fused->rewards, fused->scalars, nbx, a);
        else k_traj_reward_bwd<false><<<(int)blocks, TO_SP_THREADS, 0, s.st>>>(lo_sum, n, fused->eps, fused->prefilled ? 1 : 0, fused->rewards, fused->scalars, nbx, a);
        TO_HIP_CHECK_LAUNCH();
        return launch_finish(s, finish_post(s, 1, scalars, gout, nullptr, fused->eps), poses_grad, quats_grad);
    }
    {
        TO_PROF(TOHIP_PROF_BWD, s.st);
        a.grad_rewards = grad_rewards;
        a.scalars = grad_rewards ? nullptr : scalars;
        a.gout = grad_rewards ? nullptr : gout;
        a.unit = nullptr;   // (the forward before this call need not have been a fused one: r (1 - r) from the log-odds vector)
        rc = launch_pairs(s, a);
        if (rc != TOHIP_OK) return rc;
    }
    return launch_finish(s, finish_post(s, 0, nullptr, nullptr, nullptr, 0.f), poses_grad, quats_grad);
}

}  // namespace

extern "C" int tohip_traj_backward_multi(const void* packed, int64_t n, int64_t W, int64_t n_traj, const tohip_camera* cam,
                                         const tohip_rig* rig, int flags, const uint32_t* occlusion_bits, const float* lo_sum,
                                         const float* grad_rewards, const float* scalars, const float* gout, float* poses_grad,
                                         float* quats_grad, void* workspace, size_t workspace_bytes, void* stream_) {
    return traj_backward_impl(packed, n, W, n_traj, cam, rig, flags, occlusion_bits, const_cast<float*>(lo_sum), grad_rewards, scalars, gout,
                              poses_grad, quats_grad, workspace, workspace_bytes, stream_, nullptr);
}

extern "C" int tohip_traj_reward_backward_multi(const void* packed, int64_t n, int64_t W, int64_t n_traj, const tohip_camera* cam,
                                                const tohip_rig* rig, int flags, const uint32_t* occlusion_bits, const float* lo_sum,
                                                float eps, int prefilled, float* rewards, float* scalars, const float* gout,
                                                float* poses_grad, float* quats_grad, void* workspace, size_t workspace_bytes,
                                                void* stream_) {
    if (!rewards || !scalars || !gout) return TOHIP_EINVAL;
    const FusedReward f{eps, prefilled, rewards, scalars};
    return traj_backward_impl(packed, n, W, n_traj, cam, rig, flags, occlusion_bits, const_cast<float*>(lo_sum), nullptr, scalars, gout, poses_grad,
                              quats_grad, workspace, workspace_bytes, stream_, &f);
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

namespace {
// the fused step, launches 1-3: probe, pass 1, k_traj_sparse<FUSED> (log-odds, rewards, their integer sum, the pair sums with
// unit upstream gradient).  The finish (launch 4) needs dL/d loss_vis and may follow later (loss_kernels.hip).
int traj_fused_forward(TrajStep& s, const float* poses, const float* quats, float* lo_sum, float* minmax, float* rewards) {
    int rc = launch_probe_pass1(s, poses, quats, lo_sum, rewards);
    if (rc != TOHIP_OK) return rc;
    SparseArgs a = sparse_args(s, s.lean ? nullptr : lo_sum);
    a.minmax = minmax;
    a.rewards = s.lean ? nullptr : rewards;
    a.prefilled = 1;
    {
        TO_PROF(TOHIP_PROF_PASS2, s.st);
        rc = launch_sparse<TO_SP_FUSED>(s, a);
    }
    if (rc != TOHIP_OK) return rc;
    TO_PROF(TOHIP_PROF_BWD, s.st);
    return launch_pairs(s, a);
}
}  // namespace

extern "C" int tohip_traj_forward_backward_multi(const void* packed, int64_t n, const float* poses, const float* quats, int64_t W,
                                                 const int32_t* traj_offsets, int64_t n_traj, const tohip_camera* cam,
                                                 const tohip_rig* rig, int flags, const uint32_t* occlusion_bits, float* lo_sum,
                                                 float* minmax, float* rewards, float* scalars, const float* gout, float* poses_grad,
                                                 float* quats_grad, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!poses || !quats || !lo_sum || !minmax || !rewards || !scalars || !gout || !poses_grad || !quats_grad) return TOHIP_EINVAL;
    TrajStep s;
    int rc = traj_step_init(s, packed, n, W, n_traj, traj_offsets, cam, rig, flags, occlusion_bits, workspace, workspace_bytes, stream_, true);
    if (rc != TOHIP_OK) return rc;
    rc = traj_fused_forward(s, poses, quats, lo_sum, minmax, rewards);
    if (rc != TOHIP_OK) return rc;
    return launch_finish(s, finish_post(s, 2, nullptr, gout, scalars, cam->eps), poses_grad, quats_grad);
}

extern "C" int tohip_traj_forward_backward(const void* packed, int64_t n, const float* poses, const float* quats, int64_t W,
                                           const tohip_camera* cam, const tohip_rig* rig, int flags, const uint32_t* occlusion_bits,
                                           float* lo_sum, float* minmax, float* rewards, float* scalars, const float* gout,
                                           float* poses_grad, float* quats_grad, void* workspace, size_t workspace_bytes, void* stream_) {
    return tohip_traj_forward_backward_multi(packed, n, poses, quats, W, nullptr, 1, cam, rig, flags, occlusion_bits, lo_sum, minmax, rewards,
                                             scalars, gout, poses_grad, quats_grad, workspace, workspace_bytes, stream_);
}

// ---- one step of TrajOpt.run (trajectory_optimization.py:100-127) as ONE call and FIVE launches ------------------------------
// zero_grad(); loss = model(); loss.backward(); optimizer.step(); early-stop bookkeeping — for n_traj trajectories over one cloud.
// The waypoint selection is a stride of the probe's reads, the regularisers and the step's Adam constants are extra blocks of
// the SPARSE kernel's launch (consumers of pro / reg come after it), the parameter update and the bookkeeping are the tail of
// every k_traj_finish block (opt_step.hpp).
namespace {
struct OptLayout { size_t off_pro, off_reg, total; };
inline OptLayout opt_layout(int64_t W, int64_t n_traj) {
    OptLayout l;
    size_t o = 0;
    l.off_pro = o; o += align_up(sizeof(OptPro) * (size_t)n_traj, 256);
    l.off_reg = o; o += align_up(sizeof(float) * 3 * (size_t)W * (size_t)n_traj, 256);
    l.total = o;
    return l;
}
}  // namespace

extern "C" size_t tohip_traj_opt_scratch_bytes(int64_t n_wps, int64_t n_traj) {
    if (n_wps <= 0 || n_traj <= 0) return 0;
    return opt_layout(n_wps, n_traj).total;
}

extern "C" int tohip_traj_opt_step(const tohip_traj_opt* o, int32_t step_index, void* stream_) {
    if (!o || !o->packed || !o->poses || !o->quats || !o->poses0 || !o->exp_avg_p || !o->exp_avg_sq_p || !o->exp_avg_q || !o->exp_avg_sq_q ||
        !o->poses_grad || !o->quats_grad || !o->lo_sum || !o->minmax || !o->rewards || !o->scalars || !o->loss_log || !o->state_log ||
        !o->workspace || !o->scratch || o->n_points <= 0 || o->n_wps < 3 || o->wps_step < 1 || o->n_traj < 1 || step_index < 0 ||
        step_index >= o->n_steps || (o->n_traj > 1 && !o->traj_offsets) || (o->flags & ~(TOHIP_TRAJ_DENSE | TOHIP_TRAJ_OPT_LAST_OUTPUTS)) != 0)
        return TOHIP_EINVAL;
    const int64_t W = o->n_wps, B = o->n_traj, n_eval = (W + o->wps_step - 1) / o->wps_step;
    const OptLayout l = opt_layout(W, B);
    if (o->scratch_bytes < l.total) return TOHIP_ENOSPC;
    const tohip_rig* rig = (o->rig.n_cams > 0 && o->rig.rig_quats) ? &o->rig : nullptr;
    TrajStep s;
    int rc = traj_step_init(s, o->packed, o->n_points, B * n_eval, B, o->traj_offsets, &o->cam, rig, o->flags & TOHIP_TRAJ_DENSE, nullptr, o->workspace,
                            o->workspace_bytes, stream_, true);
    if (rc != TOHIP_OK) return rc;
    s.lean = (o->flags & TOHIP_TRAJ_OPT_LAST_OUTPUTS) && step_index < o->n_steps - 1;
    s.wp_stride = o->wps_step;
    s.traj_rows = (int)W;
    s.max_traj_v = (int)(n_eval * s.C);
    OptStep& a = s.opt;
    a.mode = 1;
    a.poses = o->poses; a.quats = o->quats; a.poses0 = o->poses0;
    a.mp = o->exp_avg_p; a.vp = o->exp_avg_sq_p; a.mq = o->exp_avg_q; a.vq = o->exp_avg_sq_q;
    a.pg = o->poses_grad; a.qg = o->quats_grad;
    a.pro = (OptPro*)((char*)o->scratch + l.off_pro);
    a.reg = (float*)((char*)o->scratch + l.off_reg);
    a.reg_terms = nullptr;
    a.loss_log = o->loss_log;
    a.log_stride = (int64_t)o->n_steps * 8;
    a.state_stride = ((int64_t)o->n_steps + 1) * TO_OPT_STATE;
    a.state_in = o->state_log + (int64_t)step_index * TO_OPT_STATE;
    a.state_out = o->state_log + ((int64_t)step_index + 1) * TO_OPT_STATE;
    a.gout = nullptr;
    a.W = (int)W; a.n_eval = (int)n_eval; a.step = o->wps_step; a.n_traj = (int)B;
    a.smooth_w = o->smoothness_weight; a.length_w = o->traj_length_weight; a.eps = o->cam.eps;
    a.lr_pose = o->lr_pose; a.lr_quat = o->lr_quat; a.beta1 = o->beta1; a.beta2 = o->beta2; a.adam_eps = o->adam_eps;
    a.rewards_th = o->rewards_th; a.smoothness_th = o->smoothness_th;
    rc = traj_fused_forward(s, o->poses, o->quats, o->lo_sum, o->minmax, o->rewards);
    if (rc != TOHIP_OK) return rc;
    return launch_finish(s, finish_post(s, 2, nullptr, nullptr, o->scalars, o->cam.eps), o->poses_grad_eval, o->quats_grad_eval);
}

// ---- point-sharded step (the alternative to waypoint sharding) ------------------------------------------------
// Every rank holds N / R points and evaluates ALL waypoints on them.  What crosses ranks is small and does not grow with N:
//   after pass 1    the per-waypoint extrema: ONE element-wise MAX all-reduce over the int32 view of the Extrema array (the
//                   minimum is kept negated), 16 bytes per virtual waypoint
//   after the sums  TO_PSHARD_HDR + V x 40 doubles: the rank's reward sum (fixed point), and per waypoint the 14 gradient sums
//                   of its flagged pairs and the 13 + 13 of its argmin / argmax sets — all of it additive over the points
// The log-odds sum of a point is complete on its own rank (it has every waypoint), so rewards never travel.
__global__ void __launch_bounds__(64)
k_traj_pshard_final(const double* __restrict__ partial, const WayRec* __restrict__ rec, const Extrema* __restrict__ ext,
                    const WayCold* __restrict__ cold, int single, int shift, float eps, const float* __restrict__ gout,
                    float* __restrict__ scalars_out, float* __restrict__ vgrad, float* __restrict__ poses_grad,
                    float* __restrict__ quats_grad) {
    __shared__ float sgy[12];
    const int v = blockIdx.x, t = threadIdx.x;
    const double* src = partial + TO_PSHARD_HDR + (int64_t)v * TO_PSHARD_NSUM;
    const WayRec& r = rec[v];
    float a, pmax, M, invM;
    load_norm(ext[v], a, pmax, M, invM);
    // the scalars from the ranks' total: every reward of every rank, N = all points
    float sc[4];
    reward_scalars((long long)partial[0], partial[1] != 0.0, (int64_t)partial[2], shift, eps, sc);
    if (v == 0 && t == 0 && scalars_out) { scalars_out[0] = sc[0]; scalars_out[1] = sc[1]; scalars_out[2] = sc[2]; scalars_out[3] = sc[3]; }
    if (t < 12) {
        const double coef = (double)(sc[2] * (gout ? gout[0] : 1.0f));
        const double nmin = src[14 + 12], nmax = src[27 + 12];
        const double wmin = nmin > 0.0 ? coef * src[12] / nmin : 0.0;
        const double wmax = nmax > 0.0 ? coef * src[13] / nmax : 0.0;
        sgy[t] = (float)(coef * src[t] + wmin * src[14 + t] + wmax * src[27 + t]);
        if (!(M > 0.f) || !(invM < INFINITY)) sgy[t] = __builtin_nanf("");   // degenerate waypoint: NaN like its rewards (k_traj_finish)
    }
    __syncthreads();
    if (t < 12) {
        float out;
        if (t < 3) out = (float)((double)r.m[3 * t] * sgy[0] + (double)r.m[3 * t + 1] * sgy[1] + (double)r.m[3 * t + 2] * sgy[2]);
        else {
            const int j = (t - 3) / 3, i = (t - 3) % 3;
            out = (float)((double)sgy[3 + 3 * j] * r.m[3 * i] + (double)sgy[3 + 3 * j + 1] * r.m[3 * i + 1] + (double)sgy[3 + 3 * j + 2] * r.m[3 * i + 2]);
        }
        vgrad[v * 12 + t] = out;
    }
    if (single) {
        __syncthreads();
        if (t == 0) {
            float o[7];
            finish_waypoint(v, vgrad, RecRows{rec}, cold, 1, nullptr, nullptr, o);
            store_grad_row(v, o, poses_grad, quats_grad);
        }
    }
}

extern "C" size_t tohip_traj_pshard_partial_count(int64_t n_virtual) {
    return n_virtual > 0 ? (size_t)(TO_PSHARD_HDR + n_virtual * TO_PSHARD_NSUM) : 0;
}

// host helper: where the step's per-waypoint extrema live in the workspace, as int32 words (4 per virtual waypoint: -bits(min),
// bits(max), 0, 0) — the array a point-sharded run MAX-all-reduces in place between tohip_traj_pshard_pass1 and _local
extern "C" int tohip_traj_extrema_view(int64_t n_points, int64_t n_virtual, void* workspace, size_t workspace_bytes, int32_t** words,
                                       int64_t* n_words) {
    if (n_points <= 0 || n_virtual <= 0 || !workspace || !words || !n_words) return TOHIP_EINVAL;
    const TrajPlan pl = make_plan(n_points, n_virtual, n_virtual, 1);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    *words = (int32_t*)((char*)workspace + pl.off_ext);
    *n_words = n_virtual * 4;
    return TOHIP_OK;
}

extern "C" int tohip_traj_pshard_pass1(const void* packed, int64_t n_local, int64_t n_global, const float* poses, const float* quats, int64_t W,
                                       const tohip_camera* cam, const tohip_rig* rig, int flags, const uint32_t* occlusion_bits,
                                       float* lo_sum, float* rewards, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!poses || !quats || !lo_sum || !rewards || n_global < n_local) return TOHIP_EINVAL;
    TrajStep s;
    int rc = traj_step_init(s, packed, n_local, W, 1, nullptr, cam, rig, flags, occlusion_bits, workspace, workspace_bytes, stream_, true);
    if (rc != TOHIP_OK) return rc;
    return launch_probe_pass1(s, poses, quats, lo_sum, rewards);
}

extern "C" int tohip_traj_pshard_local(const void* packed, int64_t n_local, int64_t n_global, int64_t W, const tohip_camera* cam,
                                       const tohip_rig* rig, int flags, const uint32_t* occlusion_bits, float* lo_sum, float* minmax,
                                       float* rewards, double* partial, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!lo_sum || !minmax || !rewards || !partial || n_global < n_local) return TOHIP_EINVAL;
    TrajStep s;
    int rc = traj_step_init(s, packed, n_local, W, 1, nullptr, cam, rig, flags, occlusion_bits, workspace, workspace_bytes, stream_, false);
    if (rc != TOHIP_OK) return rc;
    s.shift = reward_shift(n_global);   // one fixed-point scale on every rank: the ranks' sums add up exactly
    SparseArgs a = sparse_args(s, lo_sum);
    a.minmax = minmax;
    a.rewards = rewards;
    a.prefilled = 1;
    rc = launch_sparse<TO_SP_FUSED>(s, a);
    if (rc != TOHIP_OK) return rc;
    rc = launch_pairs(s, a);
    if (rc != TOHIP_OK) return rc;
    FinishPost post = finish_post(s, 3, nullptr, nullptr, nullptr, cam->eps);
    post.partial = partial;
    return launch_finish(s, post, nullptr, nullptr);
}

extern "C" int tohip_traj_pshard_finish(int64_t n_local, int64_t n_global, int64_t W, const tohip_camera* cam, const tohip_rig* rig,
                                        const double* partial, const float* gout, float* scalars, float* poses_grad, float* quats_grad,
                                        void* workspace, size_t workspace_bytes, void* stream_) {
    if (!cam || !partial || !scalars || !poses_grad || !quats_grad || !workspace || n_local <= 0 || n_global < n_local || W <= 0) return TOHIP_EINVAL;
    const int C = rig_cams(rig);
    const int64_t V = W * C;
    const TrajPlan pl = make_plan(n_local, V, W, 1);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    const WayRec* rec = (const WayRec*)(ws + pl.off_rec);
    const WayCold* cold = (const WayCold*)(ws + pl.off_cold);
    float* vgrad = (float*)(ws + pl.off_vgrad);
    const bool single = C == 1 && !(rig && rig->rig_quats);
    k_traj_pshard_final<<<(int)V, 64, 0, st>>>(partial, rec, (const Extrema*)(ws + pl.off_ext), cold, single ? 1 : 0, reward_shift(n_global), cam->eps,
                                               gout, scalars, vgrad, poses_grad, quats_grad);
    TO_HIP_CHECK_LAUNCH();
    if (!single) {
        k_traj_bwd_finish2<<<(int)((W + 63) / 64), 64, 0, st>>>(vgrad, rec, cold, (int)W, C, rig->rig_quats, rig->rig_trans, poses_grad, quats_grad,
                                                                OptStep{}, nullptr);
        TO_HIP_CHECK_LAUNCH();
    }
    return TOHIP_OK;
}

// ---- the log-odds vector of a waypoint-sharded step, compacted for its all-reduce ---------------------------------------------
// A rank's partial log-odds vector is exactly zero outside the slots its pass 1 listed as candidates (6-8 % of the slots on the
// BASELINE workloads).  The ranks MAX-reduce a 0/1 flag per slot (16 KB at 1 M points), pack the slots of the union — the same
// set on every rank — into one contiguous buffer, all-reduce that, and unpack: the slots outside the union are zero on every
// rank and stay untouched.

// flag[s] = 1 when slot s is a candidate of this rank's last forward, else 0 (one int32 per slot: RCCL reduces with MAX, it has no OR)
__global__ void k_candidate_flags(const unsigned long long* __restrict__ cbits, int fv_words, int n_traj, int nslots, int* __restrict__ flag) {
    for (int sl = blockIdx.x * blockDim.x + threadIdx.x; sl < nslots; sl += gridDim.x * blockDim.x) {
        unsigned long long any = 0ull;
        for (int tr = 0; tr < n_traj; ++tr) any |= cbits[((int64_t)tr * fv_words + (sl >> 6)) * TO_CBIT_STRIDE];   // of any trajectory
        flag[sl] = (int)((any >> (sl & 63)) & 1ull);
    }
}

// prefix[s] = number of set flags below slot s; prefix[nslots] = their total (one block)
__global__ void __launch_bounds__(1024) k_flag_prefix(const int* __restrict__ flag, int nslots, int* __restrict__ prefix) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (nslots + 1023) / 1024;
    int c = 0;
    for (int j = 0; j < per; ++j) { const int s = t * per + j; if (s < nslots) c += flag[s] != 0; }
    part[t] = c;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {   // inclusive scan
        const int v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = t ? part[t - 1] : 0;
    for (int j = 0; j < per; ++j) {
        const int s = t * per + j;
        if (s < nslots) { prefix[s] = run; run += flag[s] != 0; }
    }
    if (t == 1023) prefix[nslots] = part[1023];
}

// block per slot of the union: its 256-float block <-> block prefix[s] of the compact buffer (pack = 1: gather, 0: scatter)
__global__ void __launch_bounds__(TO_SLOT)
k_slots_move(const int* __restrict__ flag, const int* __restrict__ prefix, int nslots, float* __restrict__ lo_sum,
             float* __restrict__ compact, int64_t capacity_slots, int pack) {
    for (int s = blockIdx.x; s < nslots; s += gridDim.x) {
        if (!flag[s]) continue;
        const int e = prefix[s];
        if (e >= capacity_slots) continue;
        if (pack) compact[(int64_t)e * TO_SLOT + threadIdx.x] = lo_sum[(int64_t)s * TO_SLOT + threadIdx.x];
        else lo_sum[(int64_t)s * TO_SLOT + threadIdx.x] = compact[(int64_t)e * TO_SLOT + threadIdx.x];
    }
}

extern "C" int tohip_traj_candidate_flags(int64_t n_points, int64_t n_virtual, int64_t n_traj, const void* workspace, size_t workspace_bytes,
                                          int32_t* slot_flags, void* stream_) {
    if (n_points <= 0 || n_virtual <= 0 || n_traj <= 0 || !workspace || !slot_flags) return TOHIP_EINVAL;
    const TrajPlan pl = make_plan(n_points, n_virtual, n_virtual, n_traj);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    const char* ws = (const char*)workspace;
    k_candidate_flags<<<16, 256, 0, st>>>((const unsigned long long*)(ws + pl.off_cbits), pl.fv_words, (int)n_traj, pl.nslots, slot_flags);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_slot_flags_prefix(const int32_t* slot_flags, int64_t n_points, int32_t* prefix, void* stream_) {
    if (!slot_flags || !prefix || n_points <= 0) return TOHIP_EINVAL;
    const int nslots = (int)(tohip_padded_points(n_points) / TO_SLOT);
    k_flag_prefix<<<1, 1024, 0, (hipStream_t)stream_>>>(slot_flags, nslots, prefix);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_slots_pack(const int32_t* slot_flags, const int32_t* prefix, int64_t n_points, float* lo_sum, float* compact,
                                int64_t capacity_slots, int pack, void* stream_) {
    if (!slot_flags || !prefix || !lo_sum || !compact || n_points <= 0 || capacity_slots < 0) return TOHIP_EINVAL;
    const int nslots = (int)(tohip_padded_points(n_points) / TO_SLOT);
    k_slots_move<<<nslots < 2048 ? nslots : 2048, TO_SLOT, 0, (hipStream_t)stream_>>>(slot_flags, prefix, nslots, lo_sum, compact, capacity_slots, pack);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// Diagnostic: what the last forward over `workspace` found — stats[0] = flagged (slot, waypoint) pairs, stats[1] = candidate slots
// of pass 1, stats[2] = slots, stats[3] = virtual waypoints, stats[4] = the pairs the last CULLED pass 1 evaluated (device int64 x 5; the
// caller zero-fills it).
__global__ void k_traj_stats(const unsigned long long* __restrict__ cbits, int ncbits, const int* __restrict__ ctr, int nslots, int V,
                             const unsigned long long* __restrict__ live, int nlive_words, unsigned long long* __restrict__ stats) {
    unsigned long long c = 0, e = 0;
    for (int i = threadIdx.x; i < ncbits; i += blockDim.x) c += __popcll(cbits[(int64_t)i * TO_CBIT_STRIDE]);
    for (int i = threadIdx.x; i < nlive_words; i += blockDim.x) e += (unsigned long long)__popcll(live[i]);
    for (int s = 32; s > 0; s >>= 1) { c += __shfl_xor(c, s); e += __shfl_xor(e, s); }
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&stats[1], c);
    if ((threadIdx.x & 63) == 0 && e) atomicAdd(&stats[4], e);
    if (threadIdx.x == 0) {
        unsigned long long np = 0;
        for (int l = 0; l < TO_PL_SHARDS; ++l) np += (unsigned long long)ctr[l * TO_PL_STRIDE];
        stats[0] = np;
    }
    if (threadIdx.x == 0) { stats[2] = (unsigned long long)nslots; stats[3] = (unsigned long long)V; }
}

extern "C" int tohip_traj_step_stats(int64_t n_points, int64_t n_virtual, int64_t n_traj, const void* workspace, size_t workspace_bytes,
                                     int64_t* stats, void* stream_) {
    if (n_points <= 0 || n_virtual <= 0 || n_traj <= 0 || !workspace || !stats) return TOHIP_EINVAL;
    const TrajPlan pl = make_plan(n_points, n_virtual, n_virtual, n_traj);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    const char* ws = (const char*)workspace;
    k_traj_stats<<<1, 1024, 0, (hipStream_t)stream_>>>((const unsigned long long*)(ws + pl.off_cbits), pl.ncbits, (const int*)(ws + pl.off_ctr), pl.nslots,
                                                       (int)n_virtual, (const unsigned long long*)(ws + pl.off_live), (int)(n_virtual * pl.fv_words), (unsigned long long*)stats);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// Diagnostic (bench.py's roofline leg, never on in a timed pass): k_traj_pass1 stamps s_memtime / s_memrealtime per block into
// `buffer` (48 bytes per block; capacity: tohip_profile_clock_blocks) while it is set; NULL turns it off.
// The in-kernel clock is d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
extern "C" int tohip_profile_clock(void* buffer) {
    clock_stamps() = (unsigned long long*)buffer;
    return TOHIP_OK;
}
// the dense kernel's grid for this problem (with_occlusion selects the build it is asked about)
extern "C" int64_t tohip_profile_clock_blocks(int64_t n_points, int64_t n_virtual, int flags, int with_occlusion) {
    if (n_points <= 0 || n_virtual <= 0 || !(flags & TOHIP_TRAJ_DENSE)) return 0;   // only the dense kernel stamps
    const TrajPlan pl = make_plan(n_points, n_virtual, n_virtual);
    const int nblk8 = (int)(pl.npad / (TO_BLOCK * TO_PD));
    return dense_blocks(nblk8, (int)n_virtual, with_occlusion != 0);
}
