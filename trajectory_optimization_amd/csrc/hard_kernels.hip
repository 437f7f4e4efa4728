// hard_kernels.hip — cloud packing, the hard (boolean) frustum cull with ordered compaction, row gather,
// spherical flip, and a self-test of the cross-lane primitives, for gfx950.
//
//   get_cam_frustum_pts     /root/reference/src/tools.py:176-187, /root/reference/src/pc_processor.py:72-83
//   get_fov_mask(binary)    /root/reference/src/model.py:34-39
//   sphericalFlip           /root/reference/src/tools.py:38-53
//
// These produce index sets that must equal the reference's bit for bit, so the arithmetic is the
// reference CPU path's, operation by operation: K @ points as the k-ordered FMA chain of its 3x3 sgemm,
// IEEE division, torch.linalg.norm's FMA chain (all pinned on fixtures, tests/test_oracle_golden.py).

#include "common.hpp"

// ---------------------------------------------------------------------------------------------
// cloud packing: (N,3) -> Morton-sorted x|y|z (padded with copies of the last sorted point), the
// permutation back to the caller's order, and one bounding sphere per 256 sorted points.

__device__ __forceinline__ unsigned fkey(float f) {  // order-preserving float -> uint (hull_kernels.hip's boxes)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// order-preserving float -> int (the SIGNED integer order is the float order: what common.hpp's bounds_commit compares)
__device__ __forceinline__ int fkey_i(float f) {
    const int u = __float_as_int(f);
    return u >= 0 ? u : (int)(0x80000000u - (unsigned)u);   // negative floats: magnitude grows downwards
}
__device__ __forceinline__ float fkey_i_inv(int k) {
    return __int_as_float(k >= 0 ? k : (int)(0x80000000u - (unsigned)k));
}

// the box of the finite coordinates (common.hpp: bounds_commit)
__global__ void __launch_bounds__(TO_BLOCK) k_bbox(const float* __restrict__ xyz, int64_t n, int* __restrict__ bbox) {
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for_each_point(xyz, n, [&](float x, float y, float z) {
        const float p[3] = {x, y, z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (p[k] != p[k]) continue;  // NaNs do not shape the box
            const int u = fkey_i(p[k]);
            mn[k] = min(mn[k], u); mx[k] = max(mx[k], u);
        }
    });
    bounds_commit(mn, mx, bbox);
}

__device__ __forceinline__ unsigned spread10(unsigned v) {  // 10 bits -> every third bit
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

__global__ void __launch_bounds__(TO_BLOCK)
k_morton(const float* __restrict__ xyz, int64_t n, const int* __restrict__ bbox, unsigned* __restrict__ keys,
         int* __restrict__ vals) {
    int bmn[3], bmx[3];
    bounds_fold(bbox, bmn, bmx);
    const float lo[3] = {fkey_i_inv(bmn[0]), fkey_i_inv(bmn[1]), fkey_i_inv(bmn[2])};
    const float hi[3] = {fkey_i_inv(bmx[0]), fkey_i_inv(bmx[1]), fkey_i_inv(bmx[2])};
    // one cell size for all axes (cubic cells: compact tiles), set by the longest side of the box
    const float ext = fmaxf(fmaxf(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
    float sc[3];
    for (int k = 0; k < 3; ++k) sc[k] = ext > 0.f ? 1023.0f / ext : 0.f;
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) {
        unsigned q[3];
        for (int k = 0; k < 3; ++k) {
            const float f = (xyz[3 * i + k] - lo[k]) * sc[k];
            q[k] = (unsigned)fminf(fmaxf(f, 0.f), 1023.f);  // NaN -> 0
        }
        keys[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
        vals[i] = (int)i;
    }
}

__global__ void __launch_bounds__(TO_BLOCK) k_iota(int* __restrict__ vals, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) vals[i] = (int)i;
}

// block per 256 sorted points (a tile), tiles strided over the grid: gather the tile's points into the SoA rows, the permutation
// and its inverse, the probe's strided sample — and, the tile's points being in the block's registers, its bounding sphere:
// centre of the bounding box, radius = max distance to it, padded for the float rounding of the kernels' camera-frame arithmetic
// (conservative: a larger sphere only culls less).  (Until r05 the bounds were a launch of their own.)
__global__ void __launch_bounds__(TO_BLOCK)
k_pack_cloud(const float* __restrict__ xyz, const int* __restrict__ order, int64_t n, int64_t npad, float* __restrict__ soa,
             int* __restrict__ perm, int* __restrict__ inv, float* __restrict__ samples, int sample_step, int* __restrict__ hdr, int sorted,
             float4* __restrict__ bounds) {
    __shared__ float smn[3][TO_WAVES_PER_BLOCK], smx[3][TO_WAVES_PER_BLOCK], srad[TO_WAVES_PER_BLOCK];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (blockIdx.x == 0 && t == 0) hdr[0] = sorted;   // (the header was cleared before the launch)
    const int64_t ntiles = npad / TO_BLOCK;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t i = tile * TO_BLOCK + t;
        const int64_t s = order[i < n ? i : n - 1];
        const F3 row = load_row3(xyz, s);
        const float px = row.x, py = row.y, pz = row.z;
        // a NaN / inf coordinate: the reference's min() over p makes every reward of every waypoint NaN (model.py:226) — noted once
        // here, honoured by k_traj_probe (fmax / fmin and the culling would drop such a point silently)
        if (i < n && !(isfinite(px) && isfinite(py) && isfinite(pz))) atomicOr(&hdr[1], 1);
        soa[i] = px;
        soa[npad + i] = py;
        soa[2 * npad + i] = pz;
        if (i < n && i % sample_step == 0) {  // the probe's strided sample of the sorted cloud, kept contiguous
            const int64_t j = i / sample_step;
            samples[j] = px; samples[TO_PROBE_MAX + j] = py; samples[2 * TO_PROBE_MAX + j] = pz;
        }
        perm[i] = i < n ? (int)s : -1;
        if (i < n) inv[s] = (int)i;  // the way back, for kernels that produce their output in the caller's order
        // ---- the tile's bounding sphere (fminf / fmaxf drop a NaN: the radius below does not) ----
        const float p[3] = {px, py, pz};
        float mn[3], mx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            mn[k] = p[k]; mx[k] = p[k];
            for (int sh = 32; sh > 0; sh >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], sh)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], sh)); }
            if (lane == 0) { smn[k][wave] = mn[k]; smx[k][wave] = mx[k]; }
        }
        __syncthreads();
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float a = smn[k][0], b = smx[k][0];
            for (int w = 1; w < TO_WAVES_PER_BLOCK; ++w) { a = fminf(a, smn[k][w]); b = fmaxf(b, smx[k][w]); }
            c[k] = 0.5f * (a + b);
        }
        const float dx = px - c[0], dy = py - c[1], dz = pz - c[2];
        float r = sqrtf(dx * dx + dy * dy + dz * dz);
        if (!(r == r)) r = INFINITY;  // a NaN point: never cull this tile
        for (int sh = 32; sh > 0; sh >>= 1) r = fmaxf(r, __shfl_xor(r, sh));
        if (lane == 0) srad[wave] = r;
        __syncthreads();
        if (t == 0) {
            float rm = srad[0];
            for (int w = 1; w < TO_WAVES_PER_BLOCK; ++w) rm = fmaxf(rm, srad[w]);
            const float amax = fmaxf(fmaxf(fabsf(c[0]), fabsf(c[1])), fabsf(c[2])) + rm;
            bounds[tile] = make_float4(c[0], c[1], c[2], rm * 1.0001f + 1e-5f * amax + 1e-6f);
        }
        __syncthreads();   // the LDS is the next tile's
    }
}

// start values of the pack: the box's copies (k_bounds_init's), the blob's header cleared — one launch
__global__ void k_pack_init(int* __restrict__ bbox, int* __restrict__ hdr) {
    if (bbox != nullptr)
        for (int i = threadIdx.x; i < TO_BOUNDS_WORDS; i += blockDim.x) bbox[i] = i < TO_BOUNDS_COPIES * TO_BOUNDS_STRIDE ? 0x7fffffff : (int)0x80000000;
    for (int i = threadIdx.x; i < 64; i += blockDim.x) hdr[i] = 0;
}

extern "C" int64_t tohip_padded_points(int64_t n) {
    if (n <= 0) return 0;
    return (n + TOHIP_POINT_TILE - 1) / TOHIP_POINT_TILE * TOHIP_POINT_TILE;
}

extern "C" size_t tohip_packed_cloud_bytes(int64_t n) { return n > 0 ? packed_cloud_bytes(n) : 0; }

namespace {
struct PackPlan { size_t off_keys, off_keys2, off_vals, off_vals2, off_bbox, off_tmp, tmp_bytes, total; };
inline PackPlan pack_plan(int64_t n) {
    PackPlan p;
    size_t o = 0;
    p.off_keys = o;  o += align_up(sizeof(unsigned) * (size_t)n, 256);
    p.off_keys2 = o; o += align_up(sizeof(unsigned) * (size_t)n, 256);
    p.off_vals = o;  o += align_up(sizeof(int) * (size_t)n, 256);
    p.off_vals2 = o; o += align_up(sizeof(int) * (size_t)n, 256);
    p.off_bbox = o;  o += align_up(sizeof(int) * TO_BOUNDS_WORDS, 256);
    size_t tmp = 0;
    (void)sort_pairs(nullptr, tmp, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                             (int*)nullptr, (int)n, 0, 30, (hipStream_t)0);
    p.tmp_bytes = tmp;
    p.off_tmp = o;   o += align_up(tmp, 256);
    p.total = o;
    return p;
}
}  // namespace

extern "C" size_t tohip_pack_workspace_bytes(int64_t n) { return n > 0 ? pack_plan(n).total : 0; }

extern "C" int tohip_pack_cloud(const float* xyz, int64_t n, int sort, void* packed, void* workspace,
                                size_t workspace_bytes, void* stream_) {
    if (!xyz || !packed || !workspace || n <= 0 || n > (int64_t)0x7fffffff) return TOHIP_EINVAL;
    const PackPlan pl = pack_plan(n);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    unsigned* keys = (unsigned*)(ws + pl.off_keys);
    unsigned* keys2 = (unsigned*)(ws + pl.off_keys2);
    int* vals = (int*)(ws + pl.off_vals);
    int* vals2 = (int*)(ws + pl.off_vals2);
    int* bbox = (int*)(ws + pl.off_bbox);
    const int64_t npad = tohip_padded_points(n);
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    if (nb > 2048) nb = 2048;
    const int* order = vals;
    const CloudView cv = cloud_view(packed, n);
    k_pack_init<<<1, TO_BLOCK, 0, st>>>(sort ? bbox : nullptr, (int*)cv.hdr);
    TO_HIP_CHECK_LAUNCH();
    if (sort) {
        hipError_t e;
        k_bbox<<<(int)std::min<int64_t>(1024, (n / 4 + TO_BLOCK - 1) / TO_BLOCK + 1), TO_BLOCK, 0, st>>>(xyz, n, bbox);
        TO_HIP_CHECK_LAUNCH();
        k_morton<<<(int)nb, TO_BLOCK, 0, st>>>(xyz, n, bbox, keys, vals);
        TO_HIP_CHECK_LAUNCH();
        size_t tmp = pl.tmp_bytes;
        // by the 21 most significant bits (7 per axis: cells of 1/128 of the box's longest side; the stable sort keeps the caller's
        // order inside a cell): a radix pass less than all 30 bits, and a 256-point tile spans several cells either way
        // (TOHIP_PACK_SORT_LOW_BIT: an experiment knob, clamped to the bits the sort's plan was sized for — begin_bit < end_bit = 30)
        static const int low_bit = [] { const char* ev = getenv("TOHIP_PACK_SORT_LOW_BIT"); const int v = ev ? atoi(ev) : 9; return v < 0 ? 0 : (v > 29 ? 29 : v); }();
        e = sort_pairs(ws + pl.off_tmp, tmp, keys, keys2, vals, vals2, (int)n, low_bit, 30, st);
        if (e != hipSuccess) return (int)e;
        order = vals2;  // radix sort is stable: equal cells keep the caller's order (deterministic)
    } else {
        k_iota<<<(int)nb, TO_BLOCK, 0, st>>>(vals, n);
        TO_HIP_CHECK_LAUNCH();
    }
    int64_t nbp = npad / TO_BLOCK;
    if (nbp > 4096) nbp = 4096;
    k_pack_cloud<<<(int)nbp, TO_BLOCK, 0, st>>>(xyz, order, n, npad, (float*)cv.soa, (int*)cv.perm, (int*)cv.inv, (float*)cv.samples,
                                                cv.sample_step, (int*)cv.hdr, sort ? 1 : 0, (float4*)cv.bounds);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// hard frustum

struct FrustumConsts {
    float k[9];
    float wl, hl;  // img_width - 1, img_height - 1
    float dmin, dmax;
};

__device__ __forceinline__ void frustum_pred(const FrustumConsts& f, float X, float Y, float Z, bool& dist, bool& fov) {
    const float h0 = fmaf(f.k[2], Z, fmaf(f.k[1], Y, f.k[0] * X));
    const float h1 = fmaf(f.k[5], Z, fmaf(f.k[4], Y, f.k[3] * X));
    const float h2 = fmaf(f.k[8], Z, fmaf(f.k[7], Y, f.k[6] * X));
    const float u = h0 / h2, v = h1 / h2;  // IEEE-rounded division (tools.py:182)
    dist = (Z > f.dmin) && (Z < f.dmax);
    fov = (h2 > 0.0f) && (u > 1.0f) && (u < f.wl) && (v > 1.0f) && (v < f.hl);
}

#define TO_CULL_TILE 1024

// pass A: masks + number of kept points per 1024-point tile; the kept points of every 64 as one word (keep[i / 64]): pass C
// reads those 2 bits per 16 points instead of the points again
__global__ void __launch_bounds__(TO_BLOCK)
k_frustum_count(const float* __restrict__ cam, int64_t n, FrustumConsts f, uint8_t* __restrict__ dist_mask,
                uint8_t* __restrict__ fov_mask, int32_t* __restrict__ tile_count, unsigned long long* __restrict__ keep) {
    __shared__ int wave_cnt[TO_WAVES_PER_BLOCK];
    const int64_t tile0 = (int64_t)blockIdx.x * TO_CULL_TILE;
    int cnt = 0;
    for (int j = 0; j < TO_CULL_TILE / TO_BLOCK; ++j) {
        const int64_t i = tile0 + j * TO_BLOCK + threadIdx.x;
        bool d = false, v = false;
        if (i < n) {
            frustum_pred(f, cam[i], cam[n + i], cam[2 * n + i], d, v);
            if (dist_mask) dist_mask[i] = d ? 1 : 0;
            if (fov_mask) fov_mask[i] = v ? 1 : 0;
        }
        const unsigned long long b = __ballot(d && v);
        if ((threadIdx.x & 63) == 0 && tile0 + j * TO_BLOCK + threadIdx.x < n) keep[(tile0 + j * TO_BLOCK + threadIdx.x) >> 6] = b;
        cnt += __popcll(b);
    }
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// pass B: exclusive scan of the tile counts, one block, in rounds of 4 counts per thread: a 16-byte load (coalesced), the
// thread's own prefix, a scan of the threads' sums (shuffles inside a wave, the waves' totals through LDS), the running carry.
// (A 256-wide Hillis-Steele scan per 256 counts — 18 barriers each — took 69 us for the 15 625 tiles of a 16 M-point cloud.)
__global__ void __launch_bounds__(1024)
k_scan_tiles(const int32_t* __restrict__ tile_count, int ntiles, int32_t* __restrict__ tile_off,
             int32_t* __restrict__ total) {
    __shared__ int wsum[2][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = (blockDim.x + 63) >> 6;
    const bool vec = ((((uintptr_t)tile_count) | ((uintptr_t)tile_off)) & 15) == 0;
    int carry = 0, buf = 0;
    for (int c0 = 0; c0 < ntiles; c0 += 4 * (int)blockDim.x, buf ^= 1) {
        const int i = c0 + 4 * t;
        int v[4] = {0, 0, 0, 0};
        if (vec && i + 4 <= ntiles) {
            const int4 q = *reinterpret_cast<const int4*>(tile_count + i);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            for (int k = 0; k < 4; ++k) if (i + k < ntiles) v[k] = tile_count[i + k];
        }
        const int s = v[0] + v[1] + v[2] + v[3];
        int incl = s;
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const int up = __shfl_up(incl, sh);
            if (lane >= sh) incl += up;
        }
        if (lane == 63) wsum[buf][wave] = incl;
        __syncthreads();   // (the two buffers alternate: the next round's writes cannot overtake this round's reads)
        int base = carry, all = 0;
        for (int w = 0; w < nw; ++w) { if (w < wave) base += wsum[buf][w]; all += wsum[buf][w]; }
        int run = base + incl - s;
        if (vec && i + 4 <= ntiles) {
            *reinterpret_cast<int4*>(tile_off + i) = make_int4(run, run + v[0], run + v[0] + v[1], run + v[0] + v[1] + v[2]);
        } else {
            for (int k = 0; k < 4; ++k) { if (i + k < ntiles) tile_off[i + k] = run; run += v[k]; }
        }
        carry += all;
    }
    if (t == 0) *total = carry;
}
inline void launch_scan_tiles(const int32_t* tile_count, int ntiles, int32_t* tile_off, int32_t* total, hipStream_t st) {
    k_scan_tiles<<<1, ntiles > 2048 ? 1024 : TO_BLOCK, 0, st>>>(tile_count, ntiles, tile_off, total);
}

// pass C: ascending indices of kept points.  A WAVE per tile: its sixteen keep words arrive in one request (lanes 0-15), the wave
// walks them in point order with the running offset in a scalar, and no wave waits for another — a block per tile was 15 625
// blocks of a few instructions at 16 M points; a block striding over tiles chained eight dependent loads (26 us for 4 MB written).
// tile_off == nullptr (clouds of up to 2 M points: at most 2 048 tiles): no scan launch — the wave adds up the counts of the
// tiles before its own itself (32 independent loads per lane at most, one or two memory round trips, where the single-block scan
// and its launch were a third of a 1 M-point call), and the wave of the last tile writes the total.
__global__ void __launch_bounds__(TO_BLOCK)
k_frustum_write(int64_t n, const unsigned long long* __restrict__ keep, const int32_t* __restrict__ tile_off,
                const int32_t* __restrict__ tile_count, int32_t* __restrict__ total, int32_t* __restrict__ kept_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t nwords = (n + 63) >> 6;
    const int ntiles = (int)((n + TO_CULL_TILE - 1) / TO_CULL_TILE);
    const int nwaves = gridDim.x * TO_WAVES_PER_BLOCK;
    for (int tile = blockIdx.x * TO_WAVES_PER_BLOCK + (threadIdx.x >> 6); tile < ntiles; tile += nwaves) {
        const int64_t tile0 = (int64_t)tile * TO_CULL_TILE;
        const int64_t w0 = tile0 >> 6;
        const unsigned long long mine = (lane < TO_CULL_TILE / 64 && w0 + lane < nwords) ? keep[w0 + lane] : 0ull;
        int off;
        if (tile_off != nullptr) {
            off = tile_off[tile];
        } else {
            int s = 0;
            for (int j = lane; j < tile; j += 64) s += tile_count[j];
            for (int sh = 32; sh > 0; sh >>= 1) s += __shfl_xor(s, sh);
            off = s;
            if (tile == ntiles - 1 && lane == 0 && total != nullptr) *total = s + tile_count[tile];
        }
        if (kept_idx == nullptr) continue;
#pragma unroll
        for (int q = 0; q < TO_CULL_TILE / 64; ++q) {
            const unsigned long long b = (unsigned long long)__shfl((long long)mine, q);
            if ((b >> lane) & 1ull) kept_idx[off + __popcll(b & ((1ull << lane) - 1ull))] = (int32_t)(tile0 + (int64_t)q * 64 + lane);
            off += __popcll(b);
        }
    }
}

extern "C" size_t tohip_frustum_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    const size_t ntiles = (size_t)((n + TO_CULL_TILE - 1) / TO_CULL_TILE);
    return 2 * ((ntiles * sizeof(int32_t) + 255) / 256 * 256) + 256 + ((size_t)((n + 63) / 64) * 8 + 255) / 256 * 256;   // counts, offsets, total, keep bits
}

extern "C" int tohip_frustum_cull(const float* cam_3xN, int64_t n, const tohip_camera* cam, float min_dist,
                                  float max_dist, uint8_t* dist_mask, uint8_t* fov_mask, int32_t* kept_idx,
                                  int32_t* kept_count, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!cam || n < 0 || (n > 0 && !cam_3xN) || n > (int64_t)0x7fffffff || !workspace) return TOHIP_EINVAL;
    if (workspace_bytes < tohip_frustum_workspace_bytes(n)) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    if (n == 0) {
        if (kept_count) {
            hipError_t e = hipMemsetAsync(kept_count, 0, sizeof(int32_t), st);
            if (e != hipSuccess) return (int)e;
        }
        return TOHIP_OK;
    }
    FrustumConsts f;
    for (int i = 0; i < 9; ++i) f.k[i] = cam->K[i];
    f.wl = (float)((double)cam->img_width - 1.0);
    f.hl = (float)((double)cam->img_height - 1.0);
    f.dmin = min_dist;
    f.dmax = max_dist;
    const int ntiles = (int)((n + TO_CULL_TILE - 1) / TO_CULL_TILE);
    const size_t seg = ((size_t)ntiles * sizeof(int32_t) + 255) / 256 * 256;
    int32_t* tile_count = (int32_t*)workspace;
    int32_t* tile_off = (int32_t*)((char*)workspace + seg);
    int32_t* total_scratch = (int32_t*)((char*)workspace + 2 * seg);
    unsigned long long* keep = (unsigned long long*)((char*)workspace + 2 * seg + 256);
    k_frustum_count<<<ntiles, TO_BLOCK, 0, st>>>(cam_3xN, n, f, dist_mask, fov_mask, tile_count, keep);
    TO_HIP_CHECK_LAUNCH();
    const int wblocks = (ntiles + TO_WAVES_PER_BLOCK - 1) / TO_WAVES_PER_BLOCK;
    static const int own_prefix = getenv("TOHIP_FRUSTUM_OWN_PREFIX") ? atoi(getenv("TOHIP_FRUSTUM_OWN_PREFIX")) : 1;   // experiments: 0 = always scan
    if (own_prefix && ntiles <= 2048 && (kept_idx || kept_count)) {   // two launches: every wave of the write pass finds its own offset
        k_frustum_write<<<wblocks, TO_BLOCK, 0, st>>>(n, keep, nullptr, tile_count, kept_count, kept_idx);
        TO_HIP_CHECK_LAUNCH();
        return TOHIP_OK;
    }
    if (kept_idx || kept_count) {
        launch_scan_tiles(tile_count, ntiles, tile_off, kept_count ? kept_count : total_scratch, st);
        TO_HIP_CHECK_LAUNCH();
    }
    if (kept_idx) {
        k_frustum_write<<<wblocks < 8192 ? wblocks : 8192, TO_BLOCK, 0, st>>>(n, keep, tile_off, nullptr, nullptr, kept_idx);
        TO_HIP_CHECK_LAUNCH();
    }
    return TOHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// The cull stage of the per-camera pipeline (pc_processor.py:158-170) for W waypoints at once: exact transform of the
// whole cloud into every waypoint's camera frame, hard frustum test, ordered compaction — three launches instead of
// five per waypoint.  kept_idx[w*n ..] / kept_pts[(w*n + j)*3 ..] receive waypoint w's kept points in input order,
// kept_count[w] their number.
constexpr int kCullWords = (TO_CULL_TILE / TO_BLOCK) * TO_WAVES_PER_BLOCK;   // verdict words per (waypoint, tile): [trip][wave]
__global__ void __launch_bounds__(TO_BLOCK)
k_cull_wps_count(const float* __restrict__ xyz, int64_t n, const float* __restrict__ poses, const float* __restrict__ quats,
                 int normalize, FrustumConsts f, int ntiles, int32_t* __restrict__ tile_count, unsigned long long* __restrict__ keep) {
    __shared__ int wave_cnt[TO_WAVES_PER_BLOCK];
    const int w = blockIdx.y;
    const ExactPose e = exact_pose(quats + 4 * w, poses + 3 * w, normalize);
    const int64_t tile0 = (int64_t)blockIdx.x * TO_CULL_TILE;
    // the verdicts go to memory as well — a 64-bit word per (wave, trip), kCullWords per (waypoint, tile) — so that the write pass
    // transforms the kept points only (a ninth of the pairs on the occlusion workload; the test is 90 instructions per pair with its
    // three exact divisions, and both passes ran it: 0.37 + 0.46 ms for 128 x 1 M until r06)
    unsigned long long* kw = keep + ((int64_t)w * ntiles + blockIdx.x) * kCullWords;
    int cnt = 0;
    for (int j = 0; j < TO_CULL_TILE / TO_BLOCK; ++j) {
        const int64_t i = tile0 + j * TO_BLOCK + threadIdx.x;
        bool d = false, v = false;
        if (i < n) {
            float X, Y, Z;
            exact_to_cam(e, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], X, Y, Z);
            frustum_pred(f, X, Y, Z, d, v);
        }
        const unsigned long long bal = __ballot(d && v);
        if ((threadIdx.x & 63) == 0) kw[j * TO_WAVES_PER_BLOCK + (threadIdx.x >> 6)] = bal;
        cnt += __popcll(bal);
    }
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[(int64_t)w * ntiles + blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// block per waypoint: exclusive scan of its tile counts (in place) and its total
__global__ void __launch_bounds__(TO_BLOCK)
k_cull_wps_scan(int32_t* __restrict__ tile_count, int ntiles, int32_t* __restrict__ kept_count) {
    __shared__ int lds[TO_BLOCK];
    __shared__ int carry;
    int32_t* tc = tile_count + (int64_t)blockIdx.x * ntiles;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < ntiles; c0 += TO_BLOCK) {
        const int i = c0 + threadIdx.x;
        const int v = i < ntiles ? tc[i] : 0;
        lds[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < TO_BLOCK; s <<= 1) {
            const int add = (int)threadIdx.x >= s ? lds[threadIdx.x - s] : 0;
            __syncthreads();
            lds[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < ntiles) tc[i] = carry + lds[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += lds[TO_BLOCK - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) kept_count[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(TO_BLOCK)
k_cull_wps_write(const float* __restrict__ xyz, int64_t n, const float* __restrict__ poses, const float* __restrict__ quats,
                 int normalize, int ntiles, const int32_t* __restrict__ tile_off, const unsigned long long* __restrict__ keep,
                 int32_t* __restrict__ kept_idx, float* __restrict__ kept_pts, const int64_t* __restrict__ seg_off) {
    const int w = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long* kw = keep + ((int64_t)w * ntiles + blockIdx.x) * kCullWords;
    // this wave's verdict words and where its kept points go: the tile's offset + everything kept in earlier trips + in earlier waves
    // of the same trip — sixteen words, read by everybody, no barrier
    unsigned long long mine[TO_CULL_TILE / TO_BLOCK];
    int off[TO_CULL_TILE / TO_BLOCK];
    int run = tile_off[(int64_t)w * ntiles + blockIdx.x];
    bool any = false;
#pragma unroll
    for (int j = 0; j < TO_CULL_TILE / TO_BLOCK; ++j) {
#pragma unroll
        for (int k = 0; k < TO_WAVES_PER_BLOCK; ++k) {
            const unsigned long long word = kw[j * TO_WAVES_PER_BLOCK + k];
            if (k == wave) { mine[j] = word; off[j] = run; }
            run += __popcll(word);
        }
        any = any || mine[j] != 0ull;
    }
    if (!any) return;   // (wave-uniform) most waves of a waypoint's tiles keep nothing
    const ExactPose e = exact_pose(quats + 4 * w, poses + 3 * w, normalize);
    const int64_t tile0 = (int64_t)blockIdx.x * TO_CULL_TILE;
    int32_t* ki = kept_idx + (int64_t)w * n;
    // seg_off: the waypoints' kept points end to end (tohip_cull_waypoints_packed) instead of n rows apart
    float* kp = kept_pts + (seg_off ? seg_off[w] : (int64_t)w * n) * 3;
#pragma unroll
    for (int j = 0; j < TO_CULL_TILE / TO_BLOCK; ++j) {
        if (!((mine[j] >> lane) & 1ull)) continue;
        const int64_t i = tile0 + j * TO_BLOCK + threadIdx.x;
        float X, Y, Z;
        exact_to_cam(e, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], X, Y, Z);
        const int dst = off[j] + __popcll(mine[j] & ((1ull << lane) - 1ull));
        ki[dst] = (int32_t)i;
        kp[3 * (int64_t)dst] = X; kp[3 * (int64_t)dst + 1] = Y; kp[3 * (int64_t)dst + 2] = Z;
    }
}

// seg_off[w] = kept_count[0] + ... + kept_count[w-1], seg_off[n_wps] = the total: one block (n_wps <= 65 535)
__global__ void __launch_bounds__(TO_BLOCK)
k_cull_wps_offsets(const int32_t* __restrict__ kept_count, int n_wps, int64_t* __restrict__ seg_off) {
    __shared__ long long lds[TO_BLOCK];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n_wps; c0 += TO_BLOCK) {
        const int i = c0 + threadIdx.x;
        const long long v = i < n_wps ? kept_count[i] : 0;
        lds[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < TO_BLOCK; s <<= 1) {
            const long long add = (int)threadIdx.x >= s ? lds[threadIdx.x - s] : 0;
            __syncthreads();
            lds[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_wps) seg_off[i] = carry + lds[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 0) carry += lds[TO_BLOCK - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) seg_off[n_wps] = carry;
}

extern "C" size_t tohip_cull_waypoints_workspace_bytes(int64_t n, int64_t n_wps) {
    if (n <= 0 || n_wps <= 0) return 256;
    const size_t ntiles = (size_t)((n + TO_CULL_TILE - 1) / TO_CULL_TILE);
    return (ntiles * (size_t)n_wps * sizeof(int32_t) + 255) / 256 * 256 + ntiles * (size_t)n_wps * kCullWords * sizeof(unsigned long long) + 256;
}

static int cull_waypoints_impl(const float* xyz, int64_t n, const float* poses, const float* quats, int64_t n_wps,
                               int normalize, const tohip_camera* cam, float min_dist, float max_dist, int32_t* kept_idx,
                               float* kept_pts, int32_t* kept_count, int64_t* seg_off, void* workspace, size_t workspace_bytes,
                               void* stream_) {
    if (!xyz || !poses || !quats || !cam || !kept_idx || !kept_pts || !kept_count || !workspace || n <= 0 || n_wps <= 0 ||
        n > (int64_t)0x7fffffff || n_wps > 65535)
        return TOHIP_EINVAL;
    if (workspace_bytes < tohip_cull_waypoints_workspace_bytes(n, n_wps)) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    const int ntiles = (int)((n + TO_CULL_TILE - 1) / TO_CULL_TILE);
    int32_t* tile_count = (int32_t*)workspace;
    unsigned long long* keep = (unsigned long long*)((char*)workspace + ((size_t)ntiles * (size_t)n_wps * sizeof(int32_t) + 255) / 256 * 256);
    FrustumConsts f;
    for (int i = 0; i < 9; ++i) f.k[i] = cam->K[i];
    f.wl = (float)((double)cam->img_width - 1.0);
    f.hl = (float)((double)cam->img_height - 1.0);
    f.dmin = min_dist;
    f.dmax = max_dist;
    const dim3 grid((unsigned)ntiles, (unsigned)n_wps);
    k_cull_wps_count<<<grid, TO_BLOCK, 0, st>>>(xyz, n, poses, quats, normalize, f, ntiles, tile_count, keep);
    TO_HIP_CHECK_LAUNCH();
    k_cull_wps_scan<<<(unsigned)n_wps, TO_BLOCK, 0, st>>>(tile_count, ntiles, kept_count);
    TO_HIP_CHECK_LAUNCH();
    if (seg_off) {
        k_cull_wps_offsets<<<1, TO_BLOCK, 0, st>>>(kept_count, (int)n_wps, seg_off);
        TO_HIP_CHECK_LAUNCH();
    }
    k_cull_wps_write<<<grid, TO_BLOCK, 0, st>>>(xyz, n, poses, quats, normalize, ntiles, tile_count, keep, kept_idx, kept_pts, seg_off);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

extern "C" int tohip_cull_waypoints(const float* xyz, int64_t n, const float* poses, const float* quats, int64_t n_wps,
                                    int normalize, const tohip_camera* cam, float min_dist, float max_dist, int32_t* kept_idx,
                                    float* kept_pts, int32_t* kept_count, void* workspace, size_t workspace_bytes,
                                    void* stream_) {
    return cull_waypoints_impl(xyz, n, poses, quats, n_wps, normalize, cam, min_dist, max_dist, kept_idx, kept_pts, kept_count, nullptr,
                               workspace, workspace_bytes, stream_);
}

extern "C" int tohip_cull_waypoints_packed(const float* xyz, int64_t n, const float* poses, const float* quats, int64_t n_wps,
                                           int normalize, const tohip_camera* cam, float min_dist, float max_dist, int32_t* kept_idx,
                                           float* kept_pts, int32_t* kept_count, int64_t* seg_off, void* workspace,
                                           size_t workspace_bytes, void* stream_) {
    if (!seg_off) return TOHIP_EINVAL;
    return cull_waypoints_impl(xyz, n, poses, quats, n_wps, normalize, cam, min_dist, max_dist, kept_idx, kept_pts, kept_count, seg_off,
                               workspace, workspace_bytes, stream_);
}

// out[i,:] = xyz[idx[i],:]  for i < *count
__global__ void __launch_bounds__(TO_BLOCK)
k_gather_points(const float* __restrict__ xyz, int64_t n, int in_layout, const int32_t* __restrict__ idx,
                const int32_t* __restrict__ count, float* __restrict__ out) {
    const int m = *count;
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < m; i += stride) {
        const int64_t s = idx[i];
        if (in_layout == 0) {
            out[3 * i] = xyz[3 * s]; out[3 * i + 1] = xyz[3 * s + 1]; out[3 * i + 2] = xyz[3 * s + 2];
        } else {
            out[3 * i] = xyz[s]; out[3 * i + 1] = xyz[n + s]; out[3 * i + 2] = xyz[2 * n + s];
        }
    }
}

extern "C" int tohip_gather_points(const float* xyz, int64_t n, int in_layout, const int32_t* idx, const int32_t* count,
                                   int64_t capacity, float* out, void* stream_) {
    if (!xyz || !idx || !count || !out || n <= 0 || capacity < 0) return TOHIP_EINVAL;
    if (capacity == 0) return TOHIP_OK;
    int64_t nb = (capacity + TO_BLOCK - 1) / TO_BLOCK;
    if (nb > 2048) nb = 2048;
    k_gather_points<<<(int)nb, TO_BLOCK, 0, (hipStream_t)stream_>>>(xyz, n, in_layout, idx, count, out);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// spherical flip

__device__ __forceinline__ float flip_norm(float x, float y, float z) { return sqrtf(fmaf(z, z, fmaf(y, y, x * x))); }

// max of the norms on the float bits: norms are >= 0 so the integer order is the float order, and a NaN (0x7fc00000) sorts
// above +inf, i.e. propagates like torch.max does.  Every block leaves its maximum in parts[block] (no atomics: a few thousand of
// them on one line would be served one after the other); k_flip's blocks each take the maximum of the parts.
#define TO_FLIP_PARTS 2048
__global__ void __launch_bounds__(TO_BLOCK)
k_norm_max(const float* __restrict__ xyz, int64_t n, int* __restrict__ parts) {
    __shared__ int sw[TO_WAVES_PER_BLOCK];
    int m = 0;
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) {
        const float nr = flip_norm(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        m = max(m, __float_as_int(nr) & 0x7fffffff);
    }
    for (int s = 32; s > 0; s >>= 1) m = max(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) parts[blockIdx.x] = max(max(sw[0], sw[1]), max(sw[2], sw[3]));
}

__global__ void __launch_bounds__(TO_BLOCK)
k_flip(const float* __restrict__ xyz, int64_t n, const int* __restrict__ parts, int nparts, float scale,
       float* __restrict__ flipped, float* __restrict__ radius_out) {
    __shared__ int sw[TO_WAVES_PER_BLOCK];
    int m = 0;
    for (int i = threadIdx.x; i < nparts; i += TO_BLOCK) m = max(m, parts[i]);
    for (int s = 32; s > 0; s >>= 1) m = max(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = m;
    __syncthreads();
    const float radius = __int_as_float(max(max(sw[0], sw[1]), max(sw[2], sw[3]))) * scale;  // tools.py:45
    if (radius_out && blockIdx.x == 0 && threadIdx.x == 0) radius_out[0] = radius;
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const float nr = flip_norm(x, y, z);
        const float r = radius - nr;
        flipped[3 * i] = (2.0f * (r * x)) / nr + x;  // tools.py:46-52
        flipped[3 * i + 1] = (2.0f * (r * y)) / nr + y;
        flipped[3 * i + 2] = (2.0f * (r * z)) / nr + z;
    }
}

// parts: TO_FLIP_PARTS ints of scratch
static int launch_flip(const float* xyz, int64_t n, float param, float* flipped, float* radius_out, int* parts,
                       hipStream_t st) {
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    if (nb > TO_FLIP_PARTS) nb = TO_FLIP_PARTS;
    k_norm_max<<<(int)nb, TO_BLOCK, 0, st>>>(xyz, n, parts);
    TO_HIP_CHECK_LAUNCH();
    const float scale = (float)pow(10.0, (double)param);
    k_flip<<<(int)nb, TO_BLOCK, 0, st>>>(xyz, n, parts, (int)nb, scale, flipped, radius_out);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// self test: column sums / mins / maxs of a (64, K) matrix with the DPP reductions

__global__ void k_selftest_wave_reduce(const float* __restrict__ in, int k, float* __restrict__ osum,
                                       float* __restrict__ omin, float* __restrict__ omax) {
    const int lane = threadIdx.x;
    for (int c = 0; c < k; ++c) {
        const float v = in[lane * k + c];
        const float s = wave_sum63(v), mn = wave_min63(v), mx = wave_max63(v);
        // the non-negative variants, checked on |v| and folded back in: any mismatch poisons the outputs
        const float a = fabsf(v);
        const bool ok = wave_min63_nn(a) == wave_min63(a) && wave_max63_nn(a) == wave_max63(a);
        if (lane == 63) { osum[c] = ok ? s : __builtin_nanf(""); omin[c] = ok ? mn : __builtin_nanf(""); omax[c] = mx; }
    }
    // the transposed reduction of sixteen columns at a time (the pair kernel's): each column's total must land in the lane
    // wave_sum16_index names, and agree with the single-value tree to rounding — a mismatch poisons the column's sum
    for (int c0 = 0; c0 < k; c0 += 16) {
        float v16[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v16[j] = c0 + j < k ? in[lane * k + c0 + j] : 0.f;
        const float tot = wave_sum16_transposed(v16, lane);
        const int col = c0 + wave_sum16_index(lane);
        if (lane < 16 && col < k) {
            const float ref = osum[col];
            if (!(fabsf(tot - ref) <= 1e-4f * (1.0f + fabsf(ref)))) osum[col] = __builtin_nanf("");
        }
    }
}

extern "C" int tohip_selftest_wave_reduce(const float* in, int32_t k, float* osum, float* omin, float* omax, void* stream_) {
    if (!in || !osum || !omin || !omax || k <= 0) return TOHIP_EINVAL;
    k_selftest_wave_reduce<<<1, 64, 0, (hipStream_t)stream_>>>(in, k, osum, omin, omax);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}
