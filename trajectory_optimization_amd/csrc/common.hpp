// common.hpp — device-side building blocks shared by the gfx950 kernels.
//
// Compiled with -ffp-contract=off: every fused multiply-add below is written as fmaf(), so the
// same source gives the same bits in every kernel that recomputes a visibility value (pass 1,
// pass 2 and the backward compare p against the per-waypoint min/max with ==).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/trajopt_hip.h"

#define TO_WAVE 64
#define TO_BLOCK 256
#define TO_WAVES_PER_BLOCK (TO_BLOCK / TO_WAVE)

#define TO_HIP_CHECK_LAUNCH()                       \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

// ----------------------------------------------------------------------------------------------
// Camera constants (kernel argument -> SGPRs) and per-virtual-waypoint camera records.

struct CamConsts {
    float k[9];
    float halfw, halfh;      // img_width/2, img_height/2             model.py:44-45
    float inv_w, inv_h;      // 1/img_width, 1/img_height
    float mean;              // (min+max)/2                           model.py:20
    float inv_var;           // 1/std^2, std=(max-min)/2              model.py:21
    float eps;               // 1e-6                                  model.py:44
    float clip_hi;           // float32(1-1e-6)                       model.py:229
    int pinhole;             // K = [[fx,0,cx],[0,fy,cy],[0,0,1]]
    // The two image-plane Gaussians with their constants folded: with s = sqrt(1/2),
    //   a_u = ((K0 . c)/z - W/2)/W * s = (ks0 . c) * (1/z) - cw,   ks0 = K[0,:] * s / W,  cw = s / 2   (same for v, H)
    // so that  -1/2 ((dist/std)^2 + ((u-W/2)/W)^2 + ((v-H/2)/H)^2) = -(d2 * half_inv_var + a_u^2 + a_v^2):
    // 9 instead of 14 FMA-class operations per evaluation in the three big kernels.
    float ks0[3], ks1[3];
    float cw, ch;
    float half_inv_var;
};

// Hot record, one 64-byte line per virtual waypoint: c = m * (x - t).  a / invM / thr / sthr are filled by
// the min/max finishing kernel: p_hat = (p - a) * invM;  d2 > thr  =>  p_hat < 0.5 (log-odds exactly 0).
struct __attribute__((aligned(64))) WayHot {
    float m[9];   // m[3*i+j] = R[j][i]  (R = rotation camera->world of the virtual waypoint)
    float t[3];
    float a;      // min_n p
    float invM;   // 1 / max_n (p - a)
    float thr;    // squared-distance bound of the active set (+inf = no culling)
    float sthr;   // sqrt(thr)
};

// Cold record used by the gradient chain.
struct WayCold {
    float qn[4];  // normalised body quaternion
    float nrm;    // max(||q||, 1e-12)
    float pad[3];
};

// The packed cloud as the kernels see it (tohip_pack_cloud): Morton-sorted SoA + permutation + tile bounds.
struct CloudView {
    const float* soa;       // x | y | z, npad each, sorted order, pads repeat the last sorted point
    const int* perm;        // sorted position -> original index (-1 for pads)
    const float4* bounds;   // per 256 sorted points: bounding-sphere centre xyz, radius (conservative)
    const int* inv;         // original index -> sorted position (n entries)
    const float* samples;   // every sample_step-th sorted point, x | y | z (TO_PROBE_MAX each): the probe's strided sample, contiguous
    const int* hdr;         // 64 ints: [0] = 1 when the points are Morton-sorted, 0 when they keep the caller's order (perm = identity);
                            // [1] != 0 when some coordinate of the cloud is NaN or inf
    int64_t npad;
    int64_t n;
    int nsamples, sample_step;
};

#define TO_PROBE_MAX 8192   // capacity of the sample section (its layout since r02)
// every step-th sorted point is a sample: step = ceil(n / 4096) gives at most 4 096 of them — a whole number of rounds of the probe's
// blocks (the floor of r01-r04 gave 4 099 samples at 1 M points: a third round of loads for three samples in the 256-thread blocks)
static inline int probe_step(int64_t n) { const int64_t s = (n + 4095) / 4096; return (int)(s < 1 ? 1 : s); }

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// LSD radix sort of (key, value) pairs on bits [begin_bit, end_bit) — rocPRIM's device primitive, stable (equal keys keep their
// order: the Morton sorts rely on it for determinism).  tmp == NULL: only the scratch size is returned in `bytes`.
// rocPRIM takes a MERGE sort for up to 1 048 576 items by default — twenty launches of 6-8 us at a million points, whatever the bits —
// and Onesweep (one histogram, one scan, one launch per 8-bit digit) beyond: the limit is lowered so that every cloud worth the
// name gets the radix passes, whose number the callers cut by sorting on the bits they need only.
using SortConfig = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 65536>;
template <class Key, class Val>
static inline hipError_t sort_pairs(void* tmp, size_t& bytes, const Key* keys_in, Key* keys_out, const Val* vals_in, Val* vals_out, int n,
                                    int begin_bit, int end_bit, hipStream_t st) {
    return rocprim::radix_sort_pairs<SortConfig>(tmp, bytes, keys_in, keys_out, vals_in, vals_out, (unsigned)n, (unsigned)begin_bit, (unsigned)end_bit, st);
}

// Packed cloud blob (tohip_pack_cloud): [x|y|z f32, 3*npad] [perm i32, npad] [bounds float4, npad/256] [inv i32, npad]
// [samples x|y|z f32, 3*TO_PROBE_MAX] [header, 64 i32]
static inline size_t packed_cloud_bytes(int64_t n) {
    const int64_t npad = tohip_padded_points(n);
    return (size_t)npad * 16 + (size_t)(npad / 256) * 16 + (size_t)npad * 4 + (size_t)TO_PROBE_MAX * 12 + 256;
}
static inline CloudView cloud_view(const void* packed, int64_t n) {
    CloudView cv;
    cv.npad = tohip_padded_points(n);
    cv.n = n;
    cv.soa = (const float*)packed;
    cv.perm = (const int*)((const char*)packed + (size_t)cv.npad * 12);
    cv.bounds = (const float4*)((const char*)packed + (size_t)cv.npad * 16);
    cv.inv = (const int*)((const char*)packed + (size_t)cv.npad * 16 + (size_t)(cv.npad / 256) * 16);
    cv.samples = (const float*)((const char*)packed + (size_t)cv.npad * 16 + (size_t)(cv.npad / 256) * 16 + (size_t)cv.npad * 4);
    cv.hdr = (const int*)((const char*)cv.samples + (size_t)TO_PROBE_MAX * 12);
    cv.sample_step = probe_step(n);
    cv.nsamples = (int)((n + cv.sample_step - 1) / cv.sample_step);
    return cv;
}

static inline CamConsts make_consts(const tohip_camera* c) {
    CamConsts k;
    for (int i = 0; i < 9; ++i) k.k[i] = c->K[i];
    k.halfw = (float)((double)c->img_width / 2.0);
    k.halfh = (float)((double)c->img_height / 2.0);
    k.inv_w = 1.0f / c->img_width;
    k.inv_h = 1.0f / c->img_height;
    k.mean = (float)(((double)c->min_dist + (double)c->max_dist) / 2.0);
    const double sd = ((double)c->max_dist - (double)c->min_dist) / 2.0;
    k.inv_var = (float)(1.0 / (sd * sd));
    k.eps = c->eps;
    k.clip_hi = (float)(1.0 - (double)c->eps);
    k.pinhole = (c->K[1] == 0.f && c->K[3] == 0.f && c->K[6] == 0.f && c->K[7] == 0.f && c->K[8] == 1.f) ? 1 : 0;
    const double sq = 0.70710678118654752440;
    for (int i = 0; i < 3; ++i) {
        k.ks0[i] = (float)((double)c->K[i] * sq / (double)c->img_width);
        k.ks1[i] = (float)((double)c->K[3 + i] * sq / (double)c->img_height);
    }
    k.cw = (float)(0.5 * sq);
    k.ch = (float)(0.5 * sq);
    k.half_inv_var = (float)(0.5 / (sd * sd));
    return k;
}

// ----------------------------------------------------------------------------------------------
// Transcendentals on the hardware units (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each).

__device__ __forceinline__ float to_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// exp(x), ~1.5 ulp: the rounding error of x*log2(e) is recovered with one fma and folded back in.
// x is clamped to [-150, 88] (results below 2^-149 flush to 0; above: stays finite).
__device__ __forceinline__ float to_exp(float x) {
    x = __builtin_amdgcn_fmed3f(x, -150.0f, 88.0f);
    const float L2E = 1.44269504088896341f, L2E_LO = 1.925963033500519e-8f;
    const float e = x * L2E;
    float r = fmaf(x, L2E, -e);
    r = fmaf(x, L2E_LO, r);
    const float y = __builtin_amdgcn_exp2f(e);
    return fmaf(y, r * 0.693147180559945f, y);
}

__device__ __forceinline__ float to_log2(float x) { return __builtin_amdgcn_logf(x); }

// Squared-distance bound thr such that  d2 > thr  =>  exp(-0.5 * d2 * inv_var) < tau * (1 - 1e-4), i.e. the
// soft visibility p = S * exp(-0.5 (d2 inv_var + ...)) <= that bound cannot reach tau.  The 1e-4 margin covers
// every rounding in p (~1e-6).  tau outside (0,1) -> +inf (never cull).
// (f32 on the hardware's log unit, 1 ulp, rounded away by 1e-5: the probe's thread 0 has every block of its launch waiting, and
// the double-precision log of r01-r04 was half a microsecond of it.  A larger bound only evaluates a few more pairs.)
__device__ inline void cull_threshold(float tau, float inv_var, float* thr, float* sthr) {
    const float t = tau * (1.0f - 1e-4f);
    float th = INFINITY;
    if (t > 0.0f && t < 1.0f) {
        th = (-2.0f * 0.693147180559945f) * to_log2(t) / inv_var * (1.0f + 1e-5f);
        th = nextafterf(th, INFINITY);
    }
    *thr = th;
    *sthr = sqrtf(th) * 1.000001f;
}

// ----------------------------------------------------------------------------------------------
// Packed f32: two points per lane in a 64-bit register pair, so that the FMA-class arithmetic issues as v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 (one wave64 VALU instruction occupies its SIMD for 4 cycles whether it carries one float per lane
// or two).  Every element goes through exactly the operation sequence of the scalar function (vis_p / dvis_dy below), each
// operation IEEE-rounded per element, so the results are bit-identical to the scalar path; transcendentals, med3 and min/max
// stay per element.

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 pk_splat(float s) { return (f2)(s); }
__device__ __forceinline__ f2 pk_splat(f2 v) { return v; }   // (a record whose fields are register pairs already: WayRecPk)
__device__ __forceinline__ f2 pk_rcp(f2 x) { return f2{to_rcp(x.x), to_rcp(x.y)}; }

// ----------------------------------------------------------------------------------------------
// Wave64 reductions on the DPP network; the result is valid in lane 63.
//   quad_perm[1,0,3,2]=0xB1  quad_perm[2,3,0,1]=0x4E  row_half_mirror=0x141  row_mirror=0x140
//   row_bcast:15=0x142 (row_mask 0xA)  row_bcast:31=0x143 (row_mask 0xC)

#define TO_DPP_I(old, v, ctrl, rmask, bc) __builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, rmask, 0xF, bc)
#define TO_DPP_F(old, v, ctrl, rmask, bc) \
    __builtin_bit_cast(float, TO_DPP_I(__builtin_bit_cast(int, (float)(old)), __builtin_bit_cast(int, (float)(v)), ctrl, rmask, bc))

// The four in-row steps read only valid lanes, so bound_ctrl:1 changes nothing but lets hipcc fold the DPP
// move into the ALU op (v_add_f32_dpp); the two row_bcast steps write only some rows and keep the move.
__device__ __forceinline__ float wave_sum63(float v) {
    v += TO_DPP_F(0.f, v, 0xB1, 0xF, true);
    v += TO_DPP_F(0.f, v, 0x4E, 0xF, true);
    v += TO_DPP_F(0.f, v, 0x141, 0xF, true);
    v += TO_DPP_F(0.f, v, 0x140, 0xF, true);
    v += TO_DPP_F(0.f, v, 0x142, 0xA, false);
    v += TO_DPP_F(0.f, v, 0x143, 0xC, false);
    return v;
}
__device__ __forceinline__ float wave_min63(float v) {
    v = fminf(v, TO_DPP_F(v, v, 0xB1, 0xF, true));
    v = fminf(v, TO_DPP_F(v, v, 0x4E, 0xF, true));
    v = fminf(v, TO_DPP_F(v, v, 0x141, 0xF, true));
    v = fminf(v, TO_DPP_F(v, v, 0x140, 0xF, true));
    v = fminf(v, TO_DPP_F(v, v, 0x142, 0xA, false));
    v = fminf(v, TO_DPP_F(v, v, 0x143, 0xC, false));
    return v;
}
__device__ __forceinline__ float wave_max63(float v) {
    v = fmaxf(v, TO_DPP_F(v, v, 0xB1, 0xF, true));
    v = fmaxf(v, TO_DPP_F(v, v, 0x4E, 0xF, true));
    v = fmaxf(v, TO_DPP_F(v, v, 0x141, 0xF, true));
    v = fmaxf(v, TO_DPP_F(v, v, 0x140, 0xF, true));
    v = fmaxf(v, TO_DPP_F(v, v, 0x142, 0xA, false));
    v = fmaxf(v, TO_DPP_F(v, v, 0x143, 0xC, false));
    return v;
}
// min / max of NON-NEGATIVE floats (the visibility p = S * E >= +0, +inf allowed) on their bit patterns: for such
// values the signed-integer order is the float order, v_min_i32/v_max_i32 need no NaN canonicalisation and take the
// DPP operand directly (8 VALU per reduction instead of 18).  A NaN (0x7fc00000) sorts above +inf: it wins the max.
__device__ __forceinline__ float wave_min63_nn(float f) {
    int v = __builtin_bit_cast(int, f);
    v = min(v, TO_DPP_I(v, v, 0xB1, 0xF, true));
    v = min(v, TO_DPP_I(v, v, 0x4E, 0xF, true));
    v = min(v, TO_DPP_I(v, v, 0x141, 0xF, true));
    v = min(v, TO_DPP_I(v, v, 0x140, 0xF, true));
    v = min(v, TO_DPP_I(v, v, 0x142, 0xA, false));
    v = min(v, TO_DPP_I(v, v, 0x143, 0xC, false));
    return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float wave_max63_nn(float f) {
    int v = __builtin_bit_cast(int, f);
    v = max(v, TO_DPP_I(v, v, 0xB1, 0xF, true));
    v = max(v, TO_DPP_I(v, v, 0x4E, 0xF, true));
    v = max(v, TO_DPP_I(v, v, 0x141, 0xF, true));
    v = max(v, TO_DPP_I(v, v, 0x140, 0xF, true));
    v = max(v, TO_DPP_I(v, v, 0x142, 0xA, false));
    v = max(v, TO_DPP_I(v, v, 0x143, 0xC, false));
    return __builtin_bit_cast(float, v);
}

// SIXTEEN sums over the wave at once (a transposed reduction): at each of four in-row steps the two lanes of a pair split the
// values still alive between them — one keeps the even ones and takes the partner's, the other the odd ones — so the number of
// live values halves while the lanes summed double: 8 + 4 + 2 + 1 exchanges of 3 instructions instead of 16 x 6 for sixteen
// separate trees.  Afterwards lane l holds the row's total of value  bit3(l) + 2 bit2(l) + 4 bit1(l) + 8 bit0(l)  (the partners
// are the DPP network's: row_mirror flips all four lane bits, row_half_mirror the low three, quad_perm [2,3,0,1] bit 1, [1,0,3,2]
// bit 0; the selector of a step is a bit its exchange flips and the later ones do not), and two cross-row butterflies add the
// four rows.  Fixed order: deterministic.  wave_sum16_index(lane) names the value a lane ends up with.
__device__ __forceinline__ int wave_sum16_index(int lane) {
    return ((lane >> 3) & 1) | (((lane >> 2) & 1) << 1) | (((lane >> 1) & 1) << 2) | ((lane & 1) << 3);
}
__device__ __forceinline__ float wave_sum16_transposed(const float (&v)[16], int lane) {
    float a[8], b[4], c[2];
    const bool s3 = (lane >> 3) & 1, s2 = (lane >> 2) & 1, s1 = (lane >> 1) & 1, s0 = lane & 1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = s3 ? v[2 * i + 1] : v[2 * i], give = s3 ? v[2 * i] : v[2 * i + 1];
        a[i] = keep + TO_DPP_F(0.f, give, 0x140, 0xF, true);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = s2 ? a[2 * i + 1] : a[2 * i], give = s2 ? a[2 * i] : a[2 * i + 1];
        b[i] = keep + TO_DPP_F(0.f, give, 0x141, 0xF, true);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = s1 ? b[2 * i + 1] : b[2 * i], give = s1 ? b[2 * i] : b[2 * i + 1];
        c[i] = keep + TO_DPP_F(0.f, give, 0x4E, 0xF, true);
    }
    const float keep = s0 ? c[1] : c[0], give = s0 ? c[0] : c[1];
    float r = keep + TO_DPP_F(0.f, give, 0xB1, 0xF, true);
    r += __shfl_xor(r, 16);
    r += __shfl_xor(r, 32);
    return r;
}

// one (x, y, z) row of an (n,3) f32 array as ONE 12-byte load (global_load_dwordx3; rows are 4-byte aligned): a gather through a
// permutation then touches a row's line once, where three dword loads fetched it three times (2.8 GB for 16 M points)
struct __attribute__((packed, aligned(4))) F3 { float x, y, z; };
__device__ __forceinline__ F3 load_row3(const float* __restrict__ xyz, int64_t row) { return *reinterpret_cast<const F3*>(xyz + 3 * row); }

// ----------------------------------------------------------------------------------------------
// Bounds of a cloud (the pack's box, the voxel grid's cell range): a pass over (n,3) f32 whose result is six numbers.
// Device-scope atomics on ONE cache line are served one after the other (~80 ns each): six per wave from 1 000 waves were 77 us
// of an 80 us kernel over 12 MB (r04: 2 % of the HBM rate) and 560 us with 8 000 waves.  Here a lane takes FOUR points per
// iteration as three 16-byte loads (when the array is 16-byte aligned), a block combines in LDS, and the block's six atomics go to
// copy (block & 63) of the result — 64 copies, each on a line of its own; whoever reads the result folds the copies (bounds_fold).
#define TO_BOUNDS_COPIES 64
#define TO_BOUNDS_STRIDE 32   // 4-byte words between two copies (a 128-byte line)
#define TO_BOUNDS_WORDS (2 * TO_BOUNDS_COPIES * TO_BOUNDS_STRIDE)   // minima (3 used of each copy), then maxima

// f(x, y, z) for every point of xyz (n,3), grid-strided; blockDim.x == TO_BLOCK
template <class F>
__device__ __forceinline__ void for_each_point(const float* __restrict__ xyz, int64_t n, F f) {
    const int64_t tid = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x, nthreads = (int64_t)gridDim.x * TO_BLOCK;
    if ((reinterpret_cast<uintptr_t>(xyz) & 15) == 0) {
        const int64_t quads = n >> 2;
        for (int64_t q = tid; q < quads; q += nthreads) {
            const float4* p = reinterpret_cast<const float4*>(xyz) + 3 * q;
            const float4 a = p[0], b = p[1], c = p[2];
            f(a.x, a.y, a.z); f(a.w, b.x, b.y); f(b.z, b.w, c.x); f(c.y, c.z, c.w);
        }
        for (int64_t i = 4 * quads + tid; i < n; i += nthreads) f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    } else {
        for (int64_t i = tid; i < n; i += nthreads) f(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    }
}
// the block's (min, max) of three signed-comparable ints each -> one atomic per value on the block's copy
__device__ __forceinline__ void bounds_commit(int (&mn)[3], int (&mx)[3], int* __restrict__ out) {
    __shared__ int smn[TO_WAVES_PER_BLOCK][3], smx[TO_WAVES_PER_BLOCK][3];
    for (int k = 0; k < 3; ++k)
        for (int s = 32; s > 0; s >>= 1) { mn[k] = min(mn[k], __shfl_xor(mn[k], s)); mx[k] = max(mx[k], __shfl_xor(mx[k], s)); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
        for (int k = 0; k < 3; ++k) { smn[wave][k] = mn[k]; smx[wave][k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x % 3;
        const bool is_max = threadIdx.x >= 3;
        int v = is_max ? smx[0][k] : smn[0][k];
        for (int w = 1; w < TO_WAVES_PER_BLOCK; ++w) v = is_max ? max(v, smx[w][k]) : min(v, smn[w][k]);
        int* dst = out + (is_max ? TO_BOUNDS_COPIES * TO_BOUNDS_STRIDE : 0) + (blockIdx.x & (TO_BOUNDS_COPIES - 1)) * TO_BOUNDS_STRIDE + k;
        if (is_max) atomicMax(dst, v); else atomicMin(dst, v);
    }
}
// fold the copies: every thread of the block gets (mn[3], mx[3]); blockDim.x >= 64
__device__ __forceinline__ void bounds_fold(const int* __restrict__ src, int (&mn)[3], int (&mx)[3]) {
    __shared__ int sres[6];
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        for (int k = 0; k < 3; ++k) {
            int a = src[lane * TO_BOUNDS_STRIDE + k], b = src[TO_BOUNDS_COPIES * TO_BOUNDS_STRIDE + lane * TO_BOUNDS_STRIDE + k];
            for (int s = 32; s > 0; s >>= 1) { a = min(a, __shfl_xor(a, s)); b = max(b, __shfl_xor(b, s)); }
            if (lane == 0) { sres[k] = a; sres[3 + k] = b; }
        }
    }
    __syncthreads();
    for (int k = 0; k < 3; ++k) { mn[k] = sres[k]; mx[k] = sres[3 + k]; }
}
// start values: minima INT_MAX, maxima INT_MIN (one launch; the two halves are not a memset pattern)
__global__ void k_bounds_init(int* __restrict__ b) {
    for (int i = threadIdx.x; i < TO_BOUNDS_WORDS; i += blockDim.x) b[i] = i < TO_BOUNDS_COPIES * TO_BOUNDS_STRIDE ? 0x7fffffff : (int)0x80000000;
}

// block-wide double sum (fixed order -> deterministic), valid in every thread: a butterfly inside each wave, then the
// waves' totals through LDS — two barriers instead of the nine of a tree over all 256 threads (the finishing kernels call
// this up to 14 times in a row and were bound by the barriers).
__device__ __forceinline__ double block_sum_double(double v, double* lds /* >= one double per wave */) {
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (blockDim.x + 63) >> 6;
    __syncthreads();  // a previous call's totals may still be being read
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < nwaves; ++w) r += lds[w];
    return r;
}

// Rotation matrix (row-major R, camera->world) from a quaternion, homogeneous quadratic form.
__host__ __device__ inline void quat_to_R(const float q[4], float R[9]) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = w * w + x * x - y * y - z * z; R[1] = 2.f * (x * y - w * z);         R[2] = 2.f * (x * z + w * y);
    R[3] = 2.f * (x * y + w * z);         R[4] = w * w - x * x + y * y - z * z; R[5] = 2.f * (y * z - w * x);
    R[6] = 2.f * (x * z - w * y);         R[7] = 2.f * (y * z + w * x);         R[8] = w * w - x * x - y * y + z * z;
}

// ==============================================================================================
// ModelTraj records and evaluation (traj_kernels.hip).
//
// Per evaluation the reference computes c = R^T (x - t), h = K c, z = h2 + eps, u = h0/z, v = h1/z and
//   p = sigmoid(h2) * exp(-1/2 (|c - mu|^2/sigma^2 + ((u - W/2)/W)^2 + ((v - H/2)/H)^2))      model.py:13-47,:223
// Everything the camera contributes is linear in y = x - t, so the per-waypoint record holds three row vectors
//   g0 = f0 . y,  g1 = f1 . y,  z = f2 . y + eps       f0 = sqrt(L2E/2)/W * K[0,:] R^T,  f1 likewise,  f2 = K[2,:] R^T
// and the world-aligned offset of the Gaussian's centre, R mu (|c - mu| = |y - R mu|: rotations keep lengths), pre-scaled:
// sp = -sqrt(cd) R mu with cd = L2E/(2 sigma^2), so that d = sqrt(cd) y + sp = sqrt(cd) (y - R mu) is one FMA per axis and
// the exponent arrives in base 2 without further scaling (cw = ch = sqrt(L2E/2)/2):
//   A = |d|^2 + (g0/z - cw)^2 + (g1/z - ch)^2,    p = 2^-A / (1 + 2^(-L2E (z - eps)))
// 25 FMA-class operations and 4 transcendentals (v_rcp x2, v_exp x2) per evaluation; any K (no pinhole special case).
// The scalar and the packed function apply the same IEEE operations per element, so pass 1 (packed) and the sparse
// kernels (scalar) see bit-identical p — they compare it with == against the per-waypoint extrema.

#define TO_L2E 1.4426950408889634

// 128 bytes = two 64-byte lines per virtual waypoint; line 0 is all a plain evaluation reads (one s_load_dwordx16).
// L and U are the probe's results: attained values of p (maximum and minimum over a sample of the cloud).  Pass 1 folds a
// wave's (min, max) into the waypoint's running extrema (Extrema, integer atomics) only when it improves on them, which a
// handful of waves per waypoint do, and lists a slot as a CANDIDATE when its maximum reaches Lh for some waypoint: with
// min p = 0 (U == 0) a flagged pair has max >= M/2 (1 - 2^-22) > 0.49 M >= 0.49 L; while the minimum is unknown every slot is one.
struct __attribute__((aligned(128))) WayRec {
    float t[3];    //  0..2   translation of the virtual waypoint
    float f0[3];   //  3..5
    float f1[3];   //  6..8
    float f2[3];   //  9..11
    float sp[3];   // 12..14  -sqrt(cd) R mu
    float Lh;      // 15      0.49 L (0 while the minimum is not known to be zero): a slot whose maximum stays below it cannot be flagged
    float L;       // 16      probe: max of p over the sample (a lower bound of the max)
    float U;       // 17      probe: min of p over the sample (an upper bound of the min)
    float thr1;    // 18      probe: squared-distance bound, d2 > thr1  =>  p < L/2  (+inf = never cull)
    float sthr1;   // 19      sqrt(thr1), rounded up
    float azero;   // 20      probe: 1 = U == 0, hence min_n p == 0 exactly (p is never negative)
    float pad;     // 21
    float m[9];    // 22..30  m[3*i+j] = R[j][i]: c = m y (the gradient chain wants R)
    int seg;       // 31      trajectory the waypoint belongs to (several trajectories evaluated in one pass; else 0)
};
static_assert(sizeof(WayRec) == 128, "WayRec is two 64-byte lines");

// Per-waypoint extrema of p as bit patterns (p >= +0: the integer order is the float order; a NaN sorts above +inf).
// Initialised by the probe with (U, L), improved by pass 1 with atomicMax — order independent, hence deterministic.  The
// minimum is kept NEGATED (nmn = -bits(min p)) so that both fields only ever grow: a point-sharded run (every rank holds a part
// of the cloud, distributed.PointShard) combines the ranks' arrays with ONE element-wise MAX all-reduce over the int32 view, in
// place (the pad words are zero everywhere).  a = min p, M = max p - a, p_hat = (p - a) / M  (model.py:226-227).
struct __attribute__((aligned(16))) Extrema {
    int nmn, mx;
    int pad[2];
};
__device__ __forceinline__ void load_norm(const Extrema& e, float& a, float& pmax, float& M, float& invM) {
    a = __builtin_bit_cast(float, -e.nmn);
    pmax = __builtin_bit_cast(float, e.mx);
    M = pmax - a;          // == max(p - a): rounding is monotone
    invM = 1.0f / M;
}

struct EvalK {
    float eps, cw, ch, scd;   // scd = sqrt(cd), cd = L2E/(2 sigma^2)
    float nl2e;      // -L2E
    float l2e_eps;   //  L2E * eps
    float clip_hi;   // float32(1 - 1e-6)                                   model.py:229
    float inv_var;   // 1/sigma^2 (cull bounds)
    float su, sv;    // sqrt(L2E/2)/W, sqrt(L2E/2)/H (record construction)
    float mean;      // (min+max)/2
    float k[9];      // intrinsics (record construction)
};

static inline EvalK make_evalk(const tohip_camera* c) {
    EvalK k;
    const double sd = ((double)c->max_dist - (double)c->min_dist) / 2.0;
    const double s = sqrt(TO_L2E * 0.5);
    k.eps = c->eps;
    k.cw = k.ch = (float)(0.5 * s);
    k.scd = (float)sqrt(TO_L2E * 0.5 / (sd * sd));
    k.nl2e = (float)(-TO_L2E);
    k.l2e_eps = (float)(TO_L2E * (double)c->eps);
    k.clip_hi = (float)(1.0 - (double)c->eps);
    k.inv_var = (float)(1.0 / (sd * sd));
    k.su = (float)(s / (double)c->img_width);
    k.sv = (float)(s / (double)c->img_height);
    k.mean = (float)(((double)c->min_dist + (double)c->max_dist) / 2.0);
    for (int i = 0; i < 9; ++i) k.k[i] = c->K[i];
    return k;
}

__device__ __forceinline__ float to_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// everything the gradient wants besides p
struct VisGrad {
    float y0, y1, y2, d0, d1, d2, g0, g1, rz, au, av, S;
};

__device__ __forceinline__ float vis_p(const WayRec& r, const EvalK& k, float x, float y, float z, VisGrad* o = nullptr) {
    const float y0 = x - r.t[0], y1 = y - r.t[1], y2 = z - r.t[2];
    const float g0 = fmaf(r.f0[2], y2, fmaf(r.f0[1], y1, r.f0[0] * y0));
    const float g1 = fmaf(r.f1[2], y2, fmaf(r.f1[1], y1, r.f1[0] * y0));
    const float zz = fmaf(r.f2[2], y2, fmaf(r.f2[1], y1, fmaf(r.f2[0], y0, k.eps)));
    const float d0 = fmaf(y0, k.scd, r.sp[0]), d1 = fmaf(y1, k.scd, r.sp[1]), d2 = fmaf(y2, k.scd, r.sp[2]);   // sqrt(cd) (y - R mu)
    const float dd = fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
    const float rz = to_rcp(zz);
    const float au = fmaf(g0, rz, -k.cw), av = fmaf(g1, rz, -k.ch);
    const float A = fmaf(av, av, fmaf(au, au, dd));
    const float E = to_exp2(-A);
    const float e = to_exp2(fmaf(zz, k.nl2e, k.l2e_eps));
    const float S = to_rcp(1.0f + e);
    if (o) { o->y0 = y0; o->y1 = y1; o->y2 = y2; o->d0 = d0; o->d1 = d1; o->d2 = d2; o->g0 = g0; o->g1 = g1; o->rz = rz; o->au = au; o->av = av; o->S = S; }
    return S * E;
}

// (eps2, l2e2: k.eps and k.l2e_eps in both halves of a register pair — an instruction takes one scalar operand, so a second
// constant has to sit in vector registers; the dense pass 1 keeps these two there for its whole loop)
template <class Rec>
__device__ __forceinline__ f2 vis_p_pk(const Rec& r, const EvalK& k, f2 x, f2 y, f2 z, f2 eps2, f2 l2e2, f2 scd2) {
    const f2 y0 = x - pk_splat(r.t[0]), y1 = y - pk_splat(r.t[1]), y2 = z - pk_splat(r.t[2]);
    const f2 g0 = pk_fma(pk_splat(r.f0[2]), y2, pk_fma(pk_splat(r.f0[1]), y1, pk_splat(r.f0[0]) * y0));
    const f2 g1 = pk_fma(pk_splat(r.f1[2]), y2, pk_fma(pk_splat(r.f1[1]), y1, pk_splat(r.f1[0]) * y0));
    const f2 zz = pk_fma(pk_splat(r.f2[2]), y2, pk_fma(pk_splat(r.f2[1]), y1, pk_fma(pk_splat(r.f2[0]), y0, eps2)));
    const f2 d0 = pk_fma(y0, scd2, pk_splat(r.sp[0])), d1 = pk_fma(y1, scd2, pk_splat(r.sp[1])), d2 = pk_fma(y2, scd2, pk_splat(r.sp[2]));
    const f2 dd = pk_fma(d2, d2, pk_fma(d1, d1, d0 * d0));
    const f2 rz = pk_rcp(zz);
    const f2 au = pk_fma(g0, rz, pk_splat(-k.cw)), av = pk_fma(g1, rz, pk_splat(-k.ch));
    const f2 A = pk_fma(av, av, pk_fma(au, au, dd));
    const f2 E = f2{to_exp2(-A.x), to_exp2(-A.y)};
    const f2 ea = pk_fma(zz, pk_splat(k.nl2e), l2e2);
    const f2 S = pk_rcp(pk_splat(1.0f) + f2{to_exp2(ea.x), to_exp2(ea.y)});
    return S * E;
}
template <class Rec>
__device__ __forceinline__ f2 vis_p_pk(const Rec& r, const EvalK& k, f2 x, f2 y, f2 z) {
    return vis_p_pk(r, k, x, y, z, pk_splat(k.eps), pk_splat(k.l2e_eps), pk_splat(k.scd));
}

// Line 0 of a record with every field in BOTH halves of a vector register pair.  A packed instruction takes ONE scalar operand;
// the gradient's chains use two fields of the record in one instruction all the time (f0 - q f2, scd y + sp, ...), and the
// compiler then moves a scalar into a register pair on the spot, every time (87 v_mov_b32 per pair in the pair kernel).  A wave
// that evaluates 256 points against one waypoint fills these fifteen pairs once.
struct WayRecPk {
    f2 t[3], f0[3], f1[3], f2[3], sp[3];
};
__device__ __forceinline__ WayRecPk wayrec_pk(const WayRec& r) {
    WayRecPk o;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o.t[i] = pk_splat(r.t[i]); o.f0[i] = pk_splat(r.f0[i]); o.f1[i] = pk_splat(r.f1[i]); o.f2[i] = pk_splat(r.f2[i]); o.sp[i] = pk_splat(r.sp[i]);
        asm volatile("" : "+v"(o.t[i]), "+v"(o.f0[i]), "+v"(o.f1[i]), "+v"(o.f2[i]), "+v"(o.sp[i]));
    }
    return o;
}

// d p / d y (y = x - t, world-aligned) of the same evaluation; zero where p underflowed or the pair is occluded.
//   ln p = ln S - ln2 A,   d ln S / dy = (1 - S) f2,
//   dA/dy = 2 cd (y - sp) + 2 rz [ au (f0 - g0 rz f2) + av (f1 - g1 rz f2) ]
__device__ __forceinline__ void dvis_dy(const WayRec& r, const EvalK& k, float p, const VisGrad& s, float g[3]) {
    const float q0 = s.g0 * s.rz, q1 = s.g1 * s.rz;
    const float oneS = 1.0f - s.S;
    const float c2 = 2.0f * 0.693147180559945f;
    const float dk[3] = {s.d0, s.d1, s.d2};
    const bool live = p > 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float w0 = fmaf(-q0, r.f2[i], r.f0[i]);   // f0 - g0 rz f2
        const float w1 = fmaf(-q1, r.f2[i], r.f1[i]);
        const float dA = fmaf(s.rz, fmaf(s.av, w1, s.au * w0), k.scd * dk[i]);   // (dA/dy)/2: cd (y - R mu) = sqrt(cd) d
        g[i] = live ? p * fmaf(-c2, dA, oneS * r.f2[i]) : 0.0f;
    }
}

// packed twins (two points per lane) of vis_p with its by-products and of dvis_dy: per element the operation sequence of the
// scalar functions, so p compares equal (==) with what pass 1 saw
struct VisGrad2 {
    f2 y0, y1, y2, d0, d1, d2, g0, g1, rz, au, av, S;
};

template <class Rec>
__device__ __forceinline__ f2 vis_p_pk_grad(const Rec& r, const EvalK& k, f2 x, f2 y, f2 z, VisGrad2& o) {
    const f2 y0 = x - pk_splat(r.t[0]), y1 = y - pk_splat(r.t[1]), y2 = z - pk_splat(r.t[2]);
    const f2 g0 = pk_fma(pk_splat(r.f0[2]), y2, pk_fma(pk_splat(r.f0[1]), y1, pk_splat(r.f0[0]) * y0));
    const f2 g1 = pk_fma(pk_splat(r.f1[2]), y2, pk_fma(pk_splat(r.f1[1]), y1, pk_splat(r.f1[0]) * y0));
    const f2 zz = pk_fma(pk_splat(r.f2[2]), y2, pk_fma(pk_splat(r.f2[1]), y1, pk_fma(pk_splat(r.f2[0]), y0, pk_splat(k.eps))));
    const f2 d0 = pk_fma(y0, pk_splat(k.scd), pk_splat(r.sp[0])), d1 = pk_fma(y1, pk_splat(k.scd), pk_splat(r.sp[1])),
             d2 = pk_fma(y2, pk_splat(k.scd), pk_splat(r.sp[2]));
    const f2 dd = pk_fma(d2, d2, pk_fma(d1, d1, d0 * d0));
    const f2 rz = pk_rcp(zz);
    const f2 au = pk_fma(g0, rz, pk_splat(-k.cw)), av = pk_fma(g1, rz, pk_splat(-k.ch));
    const f2 A = pk_fma(av, av, pk_fma(au, au, dd));
    const f2 E = f2{to_exp2(-A.x), to_exp2(-A.y)};
    const f2 ea = pk_fma(zz, pk_splat(k.nl2e), pk_splat(k.l2e_eps));
    const f2 S = pk_rcp(pk_splat(1.0f) + f2{to_exp2(ea.x), to_exp2(ea.y)});
    o.y0 = y0; o.y1 = y1; o.y2 = y2; o.d0 = d0; o.d1 = d1; o.d2 = d2; o.g0 = g0; o.g1 = g1; o.rz = rz; o.au = au; o.av = av; o.S = S;
    return S * E;
}

template <class Rec>
__device__ __forceinline__ void dvis_dy_pk(const Rec& r, const EvalK& k, f2 p, const VisGrad2& s, f2 g[3]) {
    const f2 q0 = s.g0 * s.rz, q1 = s.g1 * s.rz;
    const f2 oneS = pk_splat(1.0f) - s.S;
    const float c2 = 2.0f * 0.693147180559945f;
    const f2 dk[3] = {s.d0, s.d1, s.d2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const f2 w0 = pk_fma(-q0, pk_splat(r.f2[i]), pk_splat(r.f0[i]));
        const f2 w1 = pk_fma(-q1, pk_splat(r.f2[i]), pk_splat(r.f1[i]));
        const f2 dA = pk_fma(s.rz, pk_fma(s.av, w1, s.au * w0), pk_splat(k.scd) * dk[i]);
        const f2 gi = p * pk_fma(pk_splat(-c2), dA, oneS * pk_splat(r.f2[i]));
        g[i] = f2{p.x > 0.0f ? gi.x : 0.0f, p.y > 0.0f ? gi.y : 0.0f};
    }
}

// in-row (16 lanes) reductions of non-negative floats on their bit patterns: every lane of the row ends up with the result
__device__ __forceinline__ float row_min16_nn(float f) {
    int v = __builtin_bit_cast(int, f);
    v = min(v, TO_DPP_I(v, v, 0xB1, 0xF, true));
    v = min(v, TO_DPP_I(v, v, 0x4E, 0xF, true));
    v = min(v, TO_DPP_I(v, v, 0x141, 0xF, true));
    v = min(v, TO_DPP_I(v, v, 0x140, 0xF, true));
    return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float row_max16_nn(float f) {
    int v = __builtin_bit_cast(int, f);
    v = max(v, TO_DPP_I(v, v, 0xB1, 0xF, true));
    v = max(v, TO_DPP_I(v, v, 0x4E, 0xF, true));
    v = max(v, TO_DPP_I(v, v, 0x141, 0xF, true));
    v = max(v, TO_DPP_I(v, v, 0x140, 0xF, true));
    return __builtin_bit_cast(float, v);
}

// Min / max of non-negative floats over each HALF of the wave (lanes 0..31 and 32..63), valid in lanes 31 and 63 — both pass-1
// kernels hold two 256-point slots per wave (8 points per lane): the four in-row steps above, then the cross-row step as a
// single instruction (v_max_i32_dpp with dst == src1: the rows outside row_mask keep their value, which is what the step
// wants; hipcc emits a v_mov_b32_dpp plus the ALU op for the same thing).  DPP reads a VGPR written by the previous VALU
// instruction only after two wait states: s_nop 1.
__device__ __forceinline__ float half_min31_nn_fused(float f) {
    int v = __builtin_bit_cast(int, row_min16_nn(f));
    asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float half_max31_nn_fused(float f) {
    int v = __builtin_bit_cast(int, row_max16_nn(f));
    asm volatile("s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
    return __builtin_bit_cast(float, v);
}
