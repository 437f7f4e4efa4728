// ingest_kernels.hip — the data formats on the input side of the hot path, for gfx950.
//
//   PointCloud2 bytes -> (N,3) f32 with NaN rows removed   /root/reference/src/pointcloud_utils.py:22-80,180-198
//                                                          (callers cast to f32: trajectory_optimization.py:62-63)
//   VoxelGrid down-sampling in front of the optimisers      /root/reference/launch/voxels_filtering.launch:11-21
//                                                          (pcl::VoxelGrid, third-party C++, restated from its published
//                                                          algorithm: field filter -> voxel keys -> centroid per voxel,
//                                                          output in ascending voxel-key order)
//   pc_to_voxel occupancy grid                              /root/reference/src/pointcloud_utils.py:279-288

#include "common.hpp"

// ---------------------------------------------------------------------------------------------
// PointCloud2 unpack.  datatype: sensor_msgs/PointField FLOAT32 = 7, FLOAT64 = 8.

__device__ __forceinline__ float load_field(const uint8_t* p, int datatype, int big_endian) {
    if (datatype == 8) {
        unsigned long long u = 0;
        for (int k = 0; k < 8; ++k) u |= (unsigned long long)p[big_endian ? 7 - k : k] << (8 * k);
        return (float)__longlong_as_double((long long)u);
    }
    unsigned u = 0;
    for (int k = 0; k < 4; ++k) u |= (unsigned)p[big_endian ? 3 - k : k] << (8 * k);
    return __uint_as_float(u);
}

// `be` carries the layout class besides the byte order (bit 0): bit 1 = little-endian FLOAT32 fields at 4-byte aligned addresses
// (point_step, offsets and the buffer all multiples of 4: the layout every ROS driver writes) — one dword load per field instead
// of four byte loads; bit 2 = the same for FLOAT64 (8-byte aligned).
#define TO_PC2_F32_ALIGNED 2
#define TO_PC2_F64_ALIGNED 4
#define TO_PC2_VEC16 8      // with F32_ALIGNED: 16-byte points at 16-byte aligned addresses (x, y, z, intensity): one 16-byte load
__device__ __forceinline__ bool unpack_point(const uint8_t* data, int64_t i, int point_step, int xo, int yo, int zo,
                                             int datatype, int be, int remove_nans, float& x, float& y, float& z) {
    const uint8_t* p = data + i * point_step;
    if (be & TO_PC2_F32_ALIGNED) {
        if (be & TO_PC2_VEC16) {
            const float4 v = *reinterpret_cast<const float4*>(p);
            const float f[4] = {v.x, v.y, v.z, v.w};
            x = f[xo >> 2]; y = f[yo >> 2]; z = f[zo >> 2];
        } else {
            const float* q = reinterpret_cast<const float*>(p);
            x = q[xo >> 2]; y = q[yo >> 2]; z = q[zo >> 2];
        }
        return !remove_nans || (isfinite(x) && isfinite(y) && isfinite(z));
    }
    if (be & TO_PC2_F64_ALIGNED) {
        const double* q = reinterpret_cast<const double*>(p);
        const double dx = q[xo >> 3], dy = q[yo >> 3], dz = q[zo >> 3];
        x = (float)dx; y = (float)dy; z = (float)dz;
        return !remove_nans || (isfinite(dx) && isfinite(dy) && isfinite(dz));   // np.isfinite on the stored f64 values
    }
    be &= 1;
    x = load_field(p + xo, datatype, be);
    y = load_field(p + yo, datatype, be);
    z = load_field(p + zo, datatype, be);
    // np.isfinite on the stored values (pointcloud_utils.py:186); a finite f64 beyond f32 range stays "kept"
    if (!remove_nans) return true;
    if (datatype == 8) {
        // classify on the f64 bits: exponent all ones = inf/nan
        bool fin = true;
        const int offs[3] = {xo, yo, zo};
        for (int a = 0; a < 3; ++a) {
            const uint8_t* q = p + offs[a];
            const unsigned hi = ((unsigned)q[be ? 0 : 7] << 8) | q[be ? 1 : 6];
            fin = fin && (((hi >> 4) & 0x7ffu) != 0x7ffu);
        }
        return fin;
    }
    return isfinite(x) && isfinite(y) && isfinite(z);
}

// The unpack reads the message twice (count | scan | write: 58 + 74 us at 16 M points, each pass at the rate its kind of traffic
// gets on this chip; four one-pass variants with look-back descriptors measured 170-290 us).  What it need not do twice
// is a message WITHOUT invalid rows — what a lidar driver publishes (PointCloud2.is_dense; the reference ignores the flag and masks
// anyway, pointcloud_utils.py:186): there every row stays where it is.  So when the LAST message through this workspace had no
// invalid row (`hint`, a word of the caller's workspace that survives between calls), the count pass also writes every row to
// its own place, and the write pass only touches the tiles at and behind the first invalid row — none for a dense message:
// one read of the message, 16 + 12 bytes per point.  A message with invalid rows after a dense one pays the identity writes once
// (155 instead of 140 us at 16 M points) and flips the hint; a fresh workspace starts in the two-pass mode.  Same outputs either way.
#define TO_PC2_HINT_DENSE 0x44454e53
struct Pc2Mode { int hint; int mode_a; };   // hint: persistent; mode_a: what THIS call's count pass did (read by its write pass)

__global__ void __launch_bounds__(TO_BLOCK)
k_pc2_count(const uint8_t* __restrict__ data, int64_t n, int point_step, int xo, int yo, int zo, int datatype, int be,
            int remove_nans, int32_t* __restrict__ tile_count, Pc2Mode* __restrict__ mode, float* __restrict__ out) {
    __shared__ int wave_cnt[TO_WAVES_PER_BLOCK];
    const int64_t tile0 = (int64_t)blockIdx.x * 1024;
    const bool dense = mode != nullptr && (mode->hint == TO_PC2_HINT_DENSE || !remove_nans);   // (nobody writes the hint during this launch)
    if (mode != nullptr && blockIdx.x == 0 && threadIdx.x == 0) mode->mode_a = dense ? 1 : 0;
    int cnt = 0;
    for (int j = 0; j < 4; ++j) {
        const int64_t i = tile0 + j * TO_BLOCK + threadIdx.x;
        float x, y, z;
        const bool keep = i < n && unpack_point(data, i, point_step, xo, yo, zo, datatype, be, remove_nans, x, y, z);
        if (dense && i < n) { out[3 * i] = x; out[3 * i + 1] = y; out[3 * i + 2] = z; }   // where it belongs if no row before it is dropped
        cnt += __popcll(__ballot(keep));
    }
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// one 1024-point tile's kept rows to out[base ..) in message order; all threads of the block
__device__ __forceinline__ void pc2_write_tile(const uint8_t* __restrict__ data, int64_t n, int point_step, int xo, int yo, int zo, int datatype,
                                               int be, int remove_nans, int64_t tile0, int base, float* __restrict__ out, int* wave_cnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = 0; j < 4; ++j) {
        const int64_t i = tile0 + j * TO_BLOCK + threadIdx.x;
        float x = 0, y = 0, z = 0;
        const bool keep = i < n && unpack_point(data, i, point_step, xo, yo, zo, datatype, be, remove_nans, x, y, z);
        const unsigned long long b = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(b);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (keep) {
            const int64_t d = off + __popcll(b & ((1ull << lane) - 1ull));
            out[3 * d] = x; out[3 * d + 1] = y; out[3 * d + 2] = z;
        }
        base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

// a block per tile.  Adaptive (mode != nullptr): a tile that the count pass has put in place already (no row dropped before or
// inside it) is left alone — after a dense message every block leaves after three loads.  (Measured and not kept: 2 048 resident
// blocks striding over the tiles with eight (offset, count) pairs requested at once — 95 us instead of 94 for a dense message at
// 16 M points, 154 instead of 142 with 1 % NaN rows: the block turnover is not what the dense case waits for.)
__global__ void __launch_bounds__(TO_BLOCK)
k_pc2_write(const uint8_t* __restrict__ data, int64_t n, int point_step, int xo, int yo, int zo, int datatype, int be,
            int remove_nans, const int32_t* __restrict__ tile_off, const int32_t* __restrict__ tile_count, Pc2Mode* __restrict__ mode,
            const int32_t* __restrict__ total, float* __restrict__ out) {
    __shared__ int wave_cnt[TO_WAVES_PER_BLOCK];
    const int64_t tile0 = (int64_t)blockIdx.x * 1024;
    const int base = tile_off[blockIdx.x];
    if (mode != nullptr) {
        const bool placed = mode->mode_a != 0;   // the count pass wrote every row to its own place
        // the next call's hint (this call's passes read mode_a and the old hint, which nobody touches any more)
        if (blockIdx.x == 0 && threadIdx.x == 0) mode->hint = (*total == (int32_t)n) ? TO_PC2_HINT_DENSE : 0;
        const int64_t rows = n - tile0 < 1024 ? n - tile0 : 1024;
        if (placed && base == (int)tile0 && tile_count[blockIdx.x] == (int)rows) return;   // (block-uniform)
    }
    pc2_write_tile(data, n, point_step, xo, yo, zo, datatype, be, remove_nans, tile0, base, out, wave_cnt);
}

extern "C" size_t tohip_ingest_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    const size_t ntiles = (size_t)((n + 1023) / 1024);
    return 2 * align_up(ntiles * sizeof(int32_t), 256) + 256;
}

extern "C" int tohip_pointcloud2_to_xyz(const uint8_t* data, int64_t n_points, int32_t point_step, int32_t x_off,
                                        int32_t y_off, int32_t z_off, int32_t datatype, int32_t is_bigendian,
                                        int32_t remove_nans, float* out_xyz, int32_t* out_count, void* workspace,
                                        size_t workspace_bytes, void* stream_) {
    if (!out_count || !workspace || n_points < 0 || n_points > 0x7fffffffLL || point_step <= 0 ||
        (datatype != 7 && datatype != 8))
        return TOHIP_EINVAL;
    const int fsz = datatype == 8 ? 8 : 4;
    if (x_off < 0 || y_off < 0 || z_off < 0 || x_off + fsz > point_step || y_off + fsz > point_step || z_off + fsz > point_step)
        return TOHIP_EINVAL;
    if (workspace_bytes < tohip_ingest_workspace_bytes(n_points)) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    if (n_points == 0) {
        hipError_t e = hipMemsetAsync(out_count, 0, sizeof(int32_t), st);
        return e == hipSuccess ? TOHIP_OK : (int)e;
    }
    if (!data || !out_xyz) return TOHIP_EINVAL;
    const int ntiles = (int)((n_points + 1023) / 1024);
    const size_t sg = align_up((size_t)ntiles * sizeof(int32_t), 256);
    int32_t* tile_count = (int32_t*)workspace;
    int32_t* tile_off = (int32_t*)((char*)workspace + sg);
    int layout = is_bigendian ? 1 : 0;
    const int al = fsz - 1;
    if (!is_bigendian && !(((uintptr_t)data | (uintptr_t)point_step | (uintptr_t)x_off | (uintptr_t)y_off | (uintptr_t)z_off) & (uintptr_t)al))
        layout |= datatype == 8 ? TO_PC2_F64_ALIGNED : TO_PC2_F32_ALIGNED;
    if ((layout & TO_PC2_F32_ALIGNED) && point_step == 16 && !((uintptr_t)data & 15)) layout |= TO_PC2_VEC16;
    // large messages only: below ~2 M points the call is its three launches' boundaries either way
    static const int adapt_env = getenv("TOHIP_PC2_ADAPTIVE") ? atoi(getenv("TOHIP_PC2_ADAPTIVE")) : 1;   // experiments: 0 = always two passes
    Pc2Mode* mode = (adapt_env && n_points >= (int64_t)2 << 20) ? (Pc2Mode*)((char*)workspace + 2 * sg + 128) : nullptr;
    k_pc2_count<<<ntiles, TO_BLOCK, 0, st>>>(data, n_points, point_step, x_off, y_off, z_off, datatype, layout,
                                              remove_nans, tile_count, mode, out_xyz);
    TO_HIP_CHECK_LAUNCH();
    launch_scan_tiles(tile_count, ntiles, tile_off, out_count, st);
    TO_HIP_CHECK_LAUNCH();
    k_pc2_write<<<ntiles, TO_BLOCK, 0, st>>>(data, n_points, point_step, x_off, y_off, z_off, datatype, layout,
                                              remove_nans, tile_off, tile_count, mode, out_count, out_xyz);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// VoxelGrid (pcl::VoxelGrid<PointXYZ>::applyFilter restated): points with a non-finite coordinate or with the
// filter field outside [lim_min, lim_max] are dropped; the grid origin is floor(min corner / leaf) of the kept
// points; a voxel's output is the centroid of its points, voxels in ascending key order
// key = i + j*div_x + k*div_x*div_y.

struct VoxParams {
    float inv_leaf[3];
    int field;  // 0,1,2 = x,y,z ; -1 = no field filter
    float lim_min, lim_max;
};

__device__ __forceinline__ bool vox_keep(const VoxParams& vp, float x, float y, float z) {
    if (!(isfinite(x) && isfinite(y) && isfinite(z))) return false;
    if (vp.field >= 0) {
        const float f = vp.field == 0 ? x : (vp.field == 1 ? y : z);
        if (f > vp.lim_max || f < vp.lim_min) return false;
    }
    return true;
}

// the cell range of the kept points (common.hpp: bounds_commit); ctrl[8] = overflow ("leaf size too small")
__global__ void __launch_bounds__(TO_BLOCK)
k_vox_bounds(const float* __restrict__ xyz, int64_t n, VoxParams vp, int* __restrict__ bounds) {
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for_each_point(xyz, n, [&](float x, float y, float z) {
        if (!vox_keep(vp, x, y, z)) return;
        const float p[3] = {x, y, z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            // floor(min_p * inv_leaf) == min over points of floor(p * inv_leaf): floor and the scaling are monotone
            const int c = (int)floorf(p[k] * vp.inv_leaf[k]);
            mn[k] = min(mn[k], c); mx[k] = max(mx[k], c);
        }
    });
    bounds_commit(mn, mx, bounds);
}

__global__ void __launch_bounds__(TO_BLOCK)
k_vox_keys(const float* __restrict__ xyz, int64_t n, VoxParams vp, const int* __restrict__ bounds, int* __restrict__ ctrl,
           unsigned* __restrict__ keys, int* __restrict__ vals) {
    int cmn[3], cmx[3];
    bounds_fold(bounds, cmn, cmx);
    const long long dx = (long long)cmx[0] - cmn[0] + 1, dy = (long long)cmx[1] - cmn[1] + 1, dz = (long long)cmx[2] - cmn[2] + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctrl[8] = (cmn[0] <= cmx[0] && dx * dy * dz > 0x7fffffffLL) ? 1 : 0;  // PCL: "leaf size too small"
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        unsigned key = ~0u;  // dropped points sort to the end (a voxel key is below 2^31: PCL refuses larger grids, ctrl[8])
        if (vox_keep(vp, x, y, z)) {
            const long long ci = (long long)((int)floorf(x * vp.inv_leaf[0]) - cmn[0]);
            const long long cj = (long long)((int)floorf(y * vp.inv_leaf[1]) - cmn[1]);
            const long long ck = (long long)((int)floorf(z * vp.inv_leaf[2]) - cmn[2]);
            key = (unsigned)(ci + cj * dx + ck * dx * dy);
        }
        keys[i] = key;
        vals[i] = (int)i;
    }
}

// run heads in the sorted key array -> head flags (int) for the ordered compaction
__global__ void __launch_bounds__(TO_BLOCK)
k_vox_heads(const unsigned* __restrict__ keys, int64_t n, int* __restrict__ head) {
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride)
        head[i] = (keys[i] != ~0u && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
}

// thread per voxel: sequential f32 centroid over its run (stable sort: the caller's point order inside a voxel)
__global__ void __launch_bounds__(TO_BLOCK)
k_vox_centroids(const float* __restrict__ xyz, const unsigned* __restrict__ keys, const int* __restrict__ order,
                int64_t n, const int* __restrict__ head_pos, int* __restrict__ n_vox, const int* __restrict__ ctrl, float* __restrict__ out) {
    // pcl::VoxelGrid: "Leaf size is too small for the input dataset. Integer indices would overflow." -> it hands back its input;
    // here the count says so: -1.  ctrl[8] was written by k_vox_keys (an earlier launch), so EVERY block takes this branch or none
    // does: nobody reads *n_vox in a launch that overwrites it (r05 had block 0 write the -1 at its end while other blocks could still
    // be reading the count), and the meaningless centroid pass is skipped
    if (ctrl[8]) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *n_vox = -1;
        return;
    }
    const int m = *n_vox;
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t v = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; v < m; v += stride) {
        int64_t i = head_pos[v];
        const unsigned key = keys[i];
        float sx = 0.f, sy = 0.f, sz = 0.f;
        int cnt = 0;
        for (; i < n && keys[i] == key; ++i) {
            const F3 row = load_row3(xyz, order[i]);
            sx += row.x; sy += row.y; sz += row.z;
            ++cnt;
        }
        const float c = (float)cnt;
        out[3 * v] = sx / c; out[3 * v + 1] = sy / c; out[3 * v + 2] = sz / c;
    }
}

namespace {
struct VoxPlan { size_t off_keys, off_keys2, off_vals, off_vals2, off_head, off_hpos, off_tcnt, off_toff, off_ctrl, off_tmp, tmp_bytes, total; };
inline VoxPlan vox_plan(int64_t n) {
    VoxPlan p;
    size_t o = 0;
    const size_t ntiles = (size_t)((n + 1023) / 1024);
    p.off_keys = o;  o += align_up(sizeof(unsigned) * (size_t)n, 256);
    p.off_keys2 = o; o += align_up(sizeof(unsigned) * (size_t)n, 256);
    p.off_vals = o;  o += align_up(sizeof(int) * (size_t)n, 256);
    p.off_vals2 = o; o += align_up(sizeof(int) * (size_t)n, 256);
    p.off_head = o;  o += align_up(sizeof(int) * (size_t)n, 256);
    p.off_hpos = o;  o += align_up(sizeof(int) * (size_t)n, 256);
    p.off_tcnt = o;  o += align_up(sizeof(int) * ntiles, 256);
    p.off_toff = o;  o += align_up(sizeof(int) * ntiles, 256);
    p.off_ctrl = o;  o += 256 + align_up(sizeof(int) * TO_BOUNDS_WORDS, 256);   // [8] the overflow mark; the cell range's copies behind it
    size_t tmp = 0;
    (void)sort_pairs(nullptr, tmp, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, (int)n, 0, 32, (hipStream_t)0);
    p.tmp_bytes = tmp;
    p.off_tmp = o;   o += align_up(tmp, 256);
    p.total = o;
    return p;
}
}  // namespace

extern "C" size_t tohip_voxel_grid_workspace_bytes(int64_t n) { return n > 0 ? vox_plan(n).total : 256; }

// out_xyz capacity n rows; *out_count (device) = number of voxels, or -1 when the grid has more than 2^31 - 1 cells (PCL then returns
// its input unfiltered: the caller's move).  filter_field: 0/1/2 or -1.
extern "C" int tohip_voxel_grid(const float* xyz, int64_t n, float leaf_x, float leaf_y, float leaf_z, int32_t filter_field,
                                float limit_min, float limit_max, float* out_xyz, int32_t* out_count, void* workspace,
                                size_t workspace_bytes, void* stream_) {
    if (!out_count || !workspace || n < 0 || n > 0x7fffffffLL || !(leaf_x > 0.f) || !(leaf_y > 0.f) || !(leaf_z > 0.f) ||
        filter_field > 2)
        return TOHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    if (n == 0) {
        hipError_t e = hipMemsetAsync(out_count, 0, sizeof(int32_t), st);
        return e == hipSuccess ? TOHIP_OK : (int)e;
    }
    if (!xyz || !out_xyz) return TOHIP_EINVAL;
    const VoxPlan pl = vox_plan(n);
    if (workspace_bytes < pl.total) return TOHIP_ENOSPC;
    char* ws = (char*)workspace;
    auto* keys = (unsigned*)(ws + pl.off_keys);
    auto* keys2 = (unsigned*)(ws + pl.off_keys2);
    int* vals = (int*)(ws + pl.off_vals);
    int* vals2 = (int*)(ws + pl.off_vals2);
    int* head = (int*)(ws + pl.off_head);
    int* hpos = (int*)(ws + pl.off_hpos);
    int* tcnt = (int*)(ws + pl.off_tcnt);
    int* toff = (int*)(ws + pl.off_toff);
    int* ctrl = (int*)(ws + pl.off_ctrl);
    VoxParams vp;
    vp.inv_leaf[0] = 1.0f / leaf_x; vp.inv_leaf[1] = 1.0f / leaf_y; vp.inv_leaf[2] = 1.0f / leaf_z;  // pcl: inverse_leaf_size_
    vp.field = filter_field < 0 ? -1 : filter_field;
    vp.lim_min = limit_min; vp.lim_max = limit_max;
    int* bounds = ctrl + 64;
    hipError_t e;
    k_bounds_init<<<1, TO_BLOCK, 0, st>>>(bounds);   // (r04 copied the start values from the host's stack and synchronised the stream for it)
    TO_HIP_CHECK_LAUNCH();
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    if (nb > 2048) nb = 2048;
    k_vox_bounds<<<(int)std::min<int64_t>(1024, (n / 4 + TO_BLOCK - 1) / TO_BLOCK + 1), TO_BLOCK, 0, st>>>(xyz, n, vp, bounds);
    TO_HIP_CHECK_LAUNCH();
    k_vox_keys<<<(int)nb, TO_BLOCK, 0, st>>>(xyz, n, vp, bounds, ctrl, keys, vals);
    TO_HIP_CHECK_LAUNCH();
    size_t tmp = pl.tmp_bytes;
    // a voxel key is below 2^31 (k_vox_keys marks the overflow PCL refuses), a dropped point's key is all ones: 32-bit keys, four passes
    e = sort_pairs(ws + pl.off_tmp, tmp, keys, keys2, vals, vals2, (int)n, 0, 32, st);
    if (e != hipSuccess) return (int)e;
    k_vox_heads<<<(int)nb, TO_BLOCK, 0, st>>>(keys2, n, head);
    TO_HIP_CHECK_LAUNCH();
    const int ntiles = (int)((n + 1023) / 1024);
    hull::k_flag_count<<<ntiles, TO_BLOCK, 0, st>>>(head, (int)n, tcnt);
    TO_HIP_CHECK_LAUNCH();
    launch_scan_tiles(tcnt, ntiles, toff, out_count, st);
    TO_HIP_CHECK_LAUNCH();
    hull::k_flag_write<<<ntiles, TO_BLOCK, 0, st>>>(head, (int)n, toff, hpos, (int)n);
    TO_HIP_CHECK_LAUNCH();
    k_vox_centroids<<<(int)nb, TO_BLOCK, 0, st>>>(xyz, keys2, vals2, n, hpos, out_count, ctrl, out_xyz);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// ---------------------------------------------------------------------------------------------
// pc_to_voxel: 1.0 at the cell of every point inside [x0,x1) x [y0,y1) x [z0,z1), float64 like np.zeros

__global__ void __launch_bounds__(TO_BLOCK)
k_pc_to_voxel(const float* __restrict__ pc, int64_t n, int cols, double res, double x0, double x1, double y0, double y1,
              double z0, double z1, int nx, int ny, int nz, double* __restrict__ voxel) {
    const int64_t stride = (int64_t)gridDim.x * TO_BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * TO_BLOCK + threadIdx.x; i < n; i += stride) {
        const double x = pc[cols * i], y = pc[cols * i + 1], z = pc[cols * i + 2];
        if (!(x >= x0 && x < x1 && y >= y0 && y < y1 && z >= z0 && z < z1)) continue;
        const int ci = (int)((x - x0) / res), cj = (int)((y - y0) / res), ck = (int)((z - z0) / res);  // astype(int32) truncates
        if (ci < nx && cj < ny && ck < nz) voxel[((int64_t)ci * ny + cj) * nz + ck] = 1.0;
    }
}

extern "C" int tohip_pc_to_voxel(const float* pc, int64_t n, int32_t cols, double resolution, double x0, double x1, double y0,
                                 double y1, double z0, double z1, int32_t nx, int32_t ny, int32_t nz, double* voxel,
                                 void* stream_) {
    if (!voxel || n < 0 || cols < 3 || !(resolution > 0) || nx <= 0 || ny <= 0 || nz <= 0) return TOHIP_EINVAL;
    hipStream_t st = (hipStream_t)stream_;
    hipError_t e = hipMemsetAsync(voxel, 0, sizeof(double) * (size_t)nx * ny * nz, st);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return TOHIP_OK;
    if (!pc) return TOHIP_EINVAL;
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    if (nb > 2048) nb = 2048;
    k_pc_to_voxel<<<(int)nb, TO_BLOCK, 0, st>>>(pc, n, cols, resolution, x0, x1, y0, y1, z0, z1, nx, ny, nz, voxel);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}
