// hull_kernels.hip — hidden-point removal (Katz): spherical flip + 3-D convex hull on the GPU.
//   sphericalFlip / convexHull / hidden_pts_removal   /root/reference/src/tools.py:38-85
//
// The reference hands the flipped cloud (+ the origin) to scipy.spatial.ConvexHull (Qhull, float64 on
// the float32-computed coordinates) and uses only hull.vertices.  The vertex set of a convex hull is a
// property of the point set, so any exact hull algorithm yields the same set on non-degenerate input;
// this file builds it with a data-parallel quickhull in float64:
//
//   state    points (x,y,z as f64, the origin appended as point M), per point its conflict face
//            (a face it lies strictly outside of, or -1), triangles with CCW outward orientation,
//            neighbour links per edge, unnormalised plane normals
//   round    (1) per face: farthest outside point = apex candidate (atomicMax on ordered f64 bits)
//            (2) label propagation over the face graph: each candidate claims the connected set of
//                faces its apex sees; a hashed total order breaks overlaps
//            (3) a candidate is accepted iff it owns every face its apex sees and no better candidate
//                owns a face across its horizon -> accepted regions are pairwise non-adjacent, so the
//                insertions commute and equal the sequential result
//            (4) one new triangle per horizon edge, linked to its two siblings by rotating around the
//                shared horizon vertex; (5) points of deleted faces move to a new face or retire
//   end      no point is outside any face; hull vertices = vertices of the live faces
//
// Predicates are plain f64 (coordinate differences of f32 inputs are exact; products are rounded):
// like Qhull's own, they are not exact on nearly coplanar quadruples — DESIGN.md §6.
#include "common.hpp"

namespace hull {

constexpr int kNone = -1;
constexpr int kCtrlNFaces = 0, kCtrlAnyOutside = 1, kCtrlChanged = 2, kCtrlError = 3, kCtrlAccepted = 4,
              kCtrlRound = 5, kCtrlChanged2 = 6, kCtrlAnyOutside2 = 7, kCtrlAccepted2 = 9 /* the round's values, published by
              k_commit for the host while the live counters are cleared for the next round; [8] = staged face counter */,
              kCtrlInts = 16;
constexpr int kErrCapacity = 1, kErrFlat = 2, kErrTopology = 4;

struct Bufs {
    double *px, *py, *pz;  // M1 points
    int* pface;            // M1
    int* fv;               // 3 * fcap
    int* fn;               // 3 * fcap
    double *nx, *ny, *nz;  // fcap
    unsigned long long* fmax;  // fcap
    int* fapex;            // fcap
    int* fowner;           // fcap
    int* fflags;           // fcap   bit0 alive, bit1 candidate accepted
    int* nfhead;           // fcap
    int* nfnext;           // fcap
    int* newface;          // 3 * fcap
    int* ctrl;             // kCtrlInts
    int* vflag;            // M1
    int* tile_cnt;         // ntiles(M1)
    int* tile_off;         // ntiles(M1)
    float* flipped;        // 3 * M (only used by hidden_pts_removal)
    int* flip_max;         // 1
    int m1;
    int fcap;
};

__host__ inline size_t seg(size_t bytes) { return align_up(bytes, 256); }

__host__ inline int face_capacity(int64_t m1) {
    int64_t c = 8 * m1 + 1024;
    return (int)(c > 0x3fffffff ? 0x3fffffff : c);
}

__host__ inline size_t carve(Bufs* b, char* base, int64_t n_points) {
    const int64_t m1 = n_points + 1;
    const int fcap = face_capacity(m1);
    const int ntiles = (int)((m1 + 1023) / 1024);
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += seg(bytes); return p; };
    char* p;
    p = take(sizeof(double) * m1); if (b) b->px = (double*)p;
    p = take(sizeof(double) * m1); if (b) b->py = (double*)p;
    p = take(sizeof(double) * m1); if (b) b->pz = (double*)p;
    p = take(sizeof(int) * m1); if (b) b->pface = (int*)p;
    p = take(sizeof(int) * 3 * (size_t)fcap); if (b) b->fv = (int*)p;
    p = take(sizeof(int) * 3 * (size_t)fcap); if (b) b->fn = (int*)p;
    p = take(sizeof(double) * (size_t)fcap); if (b) b->nx = (double*)p;
    p = take(sizeof(double) * (size_t)fcap); if (b) b->ny = (double*)p;
    p = take(sizeof(double) * (size_t)fcap); if (b) b->nz = (double*)p;
    p = take(sizeof(unsigned long long) * (size_t)fcap); if (b) b->fmax = (unsigned long long*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->fapex = (int*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->fowner = (int*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->fflags = (int*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->nfhead = (int*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->nfnext = (int*)p;
    p = take(sizeof(int) * 3 * (size_t)fcap); if (b) b->newface = (int*)p;
    p = take(sizeof(int) * kCtrlInts); if (b) b->ctrl = (int*)p;
    p = take(sizeof(int) * m1); if (b) b->vflag = (int*)p;
    p = take(sizeof(int) * ntiles); if (b) b->tile_cnt = (int*)p;
    p = take(sizeof(int) * ntiles); if (b) b->tile_off = (int*)p;
    p = take(sizeof(float) * 3 * (size_t)n_points); if (b) b->flipped = (float*)p;
    p = take(sizeof(int) * 4); if (b) b->flip_max = (int*)p;
    if (b) { b->m1 = (int)m1; b->fcap = fcap; }
    return o;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long dkey(double d) {
    // order-preserving map of non-negative doubles (and of all doubles) to unsigned integers
    unsigned long long u = (unsigned long long)__double_as_longlong(d);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned long long prio(int f, int round) {
    return ((unsigned long long)hash32((unsigned)f * 0x9E3779B9u + (unsigned)round) << 32) | (unsigned)f;
}

// signed (unnormalised) distance of point i from the plane of face f; > 0 = strictly outside
__device__ __forceinline__ double fdist(const Bufs& b, int f, int i) {
    const int v0 = b.fv[3 * f];
    const double dx = b.px[i] - b.px[v0], dy = b.py[i] - b.py[v0], dz = b.pz[i] - b.pz[v0];
    return b.nx[f] * dx + b.ny[f] * dy + b.nz[f] * dz;
}

__device__ __forceinline__ void set_plane(const Bufs& b, int f) {
    const int a = b.fv[3 * f], c1 = b.fv[3 * f + 1], c2 = b.fv[3 * f + 2];
    const double ux = b.px[c1] - b.px[a], uy = b.py[c1] - b.py[a], uz = b.pz[c1] - b.pz[a];
    const double vx = b.px[c2] - b.px[a], vy = b.py[c2] - b.py[a], vz = b.pz[c2] - b.pz[a];
    b.nx[f] = uy * vz - uz * vy;
    b.ny[f] = uz * vx - ux * vz;
    b.nz[f] = ux * vy - uy * vx;
}

// atomicMax of `key` into fmax[f] for every lane with f >= 0; when all live lanes of the wave share one face the
// maximum is reduced in the wave first and ONE atomic is issued (same-address atomics serialise at ~90/us, and
// in the early rounds a handful of faces own a million points).  Must be called by all lanes of the wave.
__device__ __forceinline__ void wave_face_max(const Bufs& b, int f, unsigned long long key) {
    unsigned long long todo = __ballot(f >= 0);
    // up to 8 distinct faces per wave are reduced in the wave and cost one atomic each; the rest go lane by lane
    for (int it = 0; it < 8 && todo != 0ull; ++it) {
        const int f0 = __shfl(f, __builtin_ctzll(todo));
        const bool mine = f == f0;
        unsigned long long m = mine ? key : 0ull;
        for (int s = 32; s > 0; s >>= 1) {
            const unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)(m >> 32), s) << 32) | (unsigned)__shfl_xor((int)m, s);
            m = o > m ? o : m;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(&b.fmax[f0], m);
        todo &= ~__ballot(mine);
        if (mine) f = kNone;
    }
    if (f >= 0) atomicMax(&b.fmax[f], key);
}

__global__ void __launch_bounds__(TO_BLOCK)
k_load(Bufs b, const float* __restrict__ pts, int n, int with_origin) {
    const int stride = gridDim.x * TO_BLOCK;
    for (int i = blockIdx.x * TO_BLOCK + threadIdx.x; i < b.m1; i += stride) {
        if (i < n) {
            b.px[i] = (double)pts[3 * i]; b.py[i] = (double)pts[3 * i + 1]; b.pz[i] = (double)pts[3 * i + 2];
        } else {
            // the appended origin (tools.py:60); without it the slot repeats point 0 (never a new vertex)
            b.px[i] = with_origin ? 0.0 : (double)pts[0];
            b.py[i] = with_origin ? 0.0 : (double)pts[1];
            b.pz[i] = with_origin ? 0.0 : (double)pts[2];
        }
        b.pface[i] = kNone;
        b.vflag[i] = 0;
    }
    if (blockIdx.x == 0 && threadIdx.x < kCtrlInts) b.ctrl[threadIdx.x] = 0;
}

// block-wide argmax of (key, lowest index on ties); all threads get the winner
#define HULL_INIT_THREADS 1024
__device__ int block_argmax(double key, int idx, double* skey, int* sidx, double* out_key) {
    const int t = threadIdx.x;
    skey[t] = key; sidx[t] = idx;
    __syncthreads();
    for (int s = HULL_INIT_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) {
            const double k2 = skey[t + s]; const int i2 = sidx[t + s];
            if (k2 > skey[t] || (k2 == skey[t] && i2 < sidx[t])) { skey[t] = k2; sidx[t] = i2; }
        }
        __syncthreads();
    }
    const int r = sidx[0];
    if (out_key) *out_key = skey[0];
    __syncthreads();
    return r;
}

// initial tetrahedron: lowest-x point, the point farthest from it, farthest from their line, farthest
// from their plane.  One block.
__global__ void __launch_bounds__(HULL_INIT_THREADS) k_init(Bufs b) {
    __shared__ double skey[HULL_INIT_THREADS];
    __shared__ int sidx[HULL_INIT_THREADS];
    const int t = threadIdx.x, m1 = b.m1;
    double best = -INFINITY; int bi = 0x7fffffff;
    for (int i = t; i < m1; i += HULL_INIT_THREADS) { const double k = -b.px[i]; if (k > best) { best = k; bi = i; } }
    const int i0 = block_argmax(best, bi, skey, sidx, nullptr);
    const double x0 = b.px[i0], y0 = b.py[i0], z0 = b.pz[i0];
    best = -INFINITY; bi = 0x7fffffff;
    for (int i = t; i < m1; i += HULL_INIT_THREADS) {
        const double dx = b.px[i] - x0, dy = b.py[i] - y0, dz = b.pz[i] - z0;
        const double k = dx * dx + dy * dy + dz * dz;
        if (k > best) { best = k; bi = i; }
    }
    double kk;
    const int i1 = block_argmax(best, bi, skey, sidx, &kk);
    const double ex = b.px[i1] - x0, ey = b.py[i1] - y0, ez = b.pz[i1] - z0;
    best = -INFINITY; bi = 0x7fffffff;
    for (int i = t; i < m1; i += HULL_INIT_THREADS) {
        const double dx = b.px[i] - x0, dy = b.py[i] - y0, dz = b.pz[i] - z0;
        const double cx = dy * ez - dz * ey, cy = dz * ex - dx * ez, cz = dx * ey - dy * ex;
        const double k = cx * cx + cy * cy + cz * cz;
        if (k > best) { best = k; bi = i; }
    }
    double k2;
    const int i2 = block_argmax(best, bi, skey, sidx, &k2);
    const double fx = b.px[i2] - x0, fy = b.py[i2] - y0, fz = b.pz[i2] - z0;
    const double nx = ey * fz - ez * fy, ny = ez * fx - ex * fz, nz = ex * fy - ey * fx;
    best = -INFINITY; bi = 0x7fffffff;
    for (int i = t; i < m1; i += HULL_INIT_THREADS) {
        const double k = fabs(nx * (b.px[i] - x0) + ny * (b.py[i] - y0) + nz * (b.pz[i] - z0));
        if (k > best) { best = k; bi = i; }
    }
    double k3;
    const int i3 = block_argmax(best, bi, skey, sidx, &k3);
    if (t == 0) {
        if (!(kk > 0.0) || !(k2 > 0.0) || !(k3 > 0.0)) { b.ctrl[kCtrlError] = kErrFlat; return; }
        int a = i0, c1 = i1, c2 = i2;
        const int d = i3;
        const double s = nx * (b.px[d] - x0) + ny * (b.py[d] - y0) + nz * (b.pz[d] - z0);
        if (s > 0.0) { const int tmp = c1; c1 = c2; c2 = tmp; }  // d must lie below face (a,c1,c2)
        const int F[4][3] = {{a, c1, c2}, {c1, a, d}, {c2, c1, d}, {a, c2, d}};
        const int Nb[4][3] = {{1, 2, 3}, {0, 3, 2}, {0, 1, 3}, {0, 2, 1}};
        for (int f = 0; f < 4; ++f) {
            for (int k = 0; k < 3; ++k) { b.fv[3 * f + k] = F[f][k]; b.fn[3 * f + k] = Nb[f][k]; }
            set_plane(b, f);
            b.fflags[f] = 1;
            b.fowner[f] = kNone;
            b.fmax[f] = 0ull;
            b.fapex[f] = 0x7fffffff;
        }
        b.ctrl[kCtrlNFaces] = 4;
        b.ctrl[kCtrlNFaces + 8] = 4;  // staged face counter (k_new_faces allocates from it, k_commit publishes it)
    }
}

__global__ void __launch_bounds__(TO_BLOCK) k_assign0(Bufs b) {
    if (b.ctrl[kCtrlError]) return;
    const int stride = gridDim.x * TO_BLOCK;
    int sv[4];
    sv[0] = b.fv[0]; sv[1] = b.fv[1]; sv[2] = b.fv[2]; sv[3] = b.fv[5];
    const int nloop = (b.m1 + stride - 1) / stride;
    for (int it = 0; it < nloop; ++it) {
        const int i = blockIdx.x * TO_BLOCK + threadIdx.x + it * stride;
        double best = 0.0; int bf = kNone;
        if (i < b.m1 && !(i == sv[0] || i == sv[1] || i == sv[2] || i == sv[3])) {
            for (int f = 0; f < 4; ++f) { const double d = fdist(b, f, i); if (d > best) { best = d; bf = f; } }
            b.pface[i] = bf;
        }
        wave_face_max(b, bf, dkey(best));
    }
}

// ---- round --------------------------------------------------------------------------------------
__global__ void __launch_bounds__(TO_BLOCK) k_round_reset(Bufs b) {
    // per-round state of every face; faces that still have points outside them become candidates (owner = self)
    const int nf = b.ctrl[kCtrlNFaces];
    const int stride = gridDim.x * TO_BLOCK;
    bool any = false;
    for (int f = blockIdx.x * TO_BLOCK + threadIdx.x; f < nf; f += stride) {
        const int alive = b.fflags[f] & 1;
        const bool cand = alive && b.fapex[f] != 0x7fffffff;  // fmax / fapex persist: an outside set is fixed at creation
        b.fowner[f] = cand ? f : kNone;
        b.nfhead[f] = kNone;
        b.fflags[f] = alive | (cand ? 2 : 0);
        any |= cand;
    }
    if (any) b.ctrl[kCtrlAnyOutside] = 1;
}

// apex of the faces with id >= f_lo (the faces created since the last call): lowest-index point among those at
// the face's maximum distance.  A face's outside set never changes after its creation round, so its apex is
// computed once; older faces keep theirs.
__global__ void __launch_bounds__(TO_BLOCK) k_far_arg(Bufs b, int f_lo) {
    const int stride = gridDim.x * TO_BLOCK;
    for (int i = blockIdx.x * TO_BLOCK + threadIdx.x; i < b.m1; i += stride) {
        const int f = b.pface[i];
        if (f < f_lo) continue;
        if (dkey(fdist(b, f, i)) == b.fmax[f]) atomicMin(&b.fapex[f], i);
    }
}


// each live face adopts the best-priority owner among its neighbours whose apex sees it
__global__ void __launch_bounds__(TO_BLOCK) k_owner_prop(Bufs b) {
    const int nf = b.ctrl[kCtrlNFaces], round = b.ctrl[kCtrlRound];
    const int stride = gridDim.x * TO_BLOCK;
    bool changed = false;
    for (int g = blockIdx.x * TO_BLOCK + threadIdx.x; g < nf; g += stride) {
        if (!(b.fflags[g] & 1)) continue;
        int best = b.fowner[g];
        unsigned long long bp = best >= 0 ? prio(best, round) : ~0ull;
        for (int k = 0; k < 3; ++k) {
            const int o = b.fowner[b.fn[3 * g + k]];
            if (o < 0 || o == best) continue;
            const unsigned long long op = prio(o, round);
            if (op < bp && fdist(b, g, b.fapex[o]) > 0.0) { best = o; bp = op; }
        }
        if (best != b.fowner[g]) { b.fowner[g] = best; changed = true; }
    }
    if (__any(changed) && (threadIdx.x & 63) == 0) b.ctrl[kCtrlChanged] = 1;
}

__global__ void __launch_bounds__(TO_BLOCK) k_accept(Bufs b) {
    const int nf = b.ctrl[kCtrlNFaces], round = b.ctrl[kCtrlRound];
    const int stride = gridDim.x * TO_BLOCK;
    for (int g = blockIdx.x * TO_BLOCK + threadIdx.x; g < nf; g += stride) {
        if (!(b.fflags[g] & 1)) continue;
        const int o = b.fowner[g];
        if (o < 0) continue;
        if (b.fowner[o] != o) { continue; }  // o lost its own face: not a candidate (its flag is cleared below)
        const int apex = b.fapex[o];
        bool ok = true;
        for (int k = 0; k < 3; ++k) {
            const int n = b.fn[3 * g + k];
            const int on = b.fowner[n];
            if (on == o) continue;
            if (fdist(b, n, apex) > 0.0) ok = false;                                  // visible but not ours
            else if (on >= 0 && prio(on, round) < prio(o, round)) ok = false;          // adjacent to a better region
        }
        if (!ok) atomicAnd(&b.fflags[o], ~2);
    }
}


__device__ __forceinline__ bool owned_accepted(const Bufs& b, int g, int* owner) {
    const int o = b.fowner[g];
    *owner = o;
    return o >= 0 && b.fowner[o] == o && (b.fflags[o] & 2);
}

// one new triangle (u, v, apex) per horizon edge (u, v) of an accepted region
__global__ void __launch_bounds__(TO_BLOCK) k_new_faces(Bufs b) {
    const int nf = b.ctrl[kCtrlNFaces];
    const int stride = gridDim.x * TO_BLOCK;
    for (int g = blockIdx.x * TO_BLOCK + threadIdx.x; g < nf; g += stride) {
        if (!(b.fflags[g] & 1)) continue;
        int o;
        if (!owned_accepted(b, g, &o)) continue;
        if (g == o) atomicAdd(&b.ctrl[kCtrlAccepted], 1);
        const int apex = b.fapex[o];
        for (int k = 0; k < 3; ++k) {
            const int n = b.fn[3 * g + k];
            if (b.fowner[n] == o) continue;
            const int id = atomicAdd(&b.ctrl[kCtrlNFaces + 8], 1);  // staged counter, folded in by k_commit
            if (id >= b.fcap) { b.ctrl[kCtrlError] |= kErrCapacity; continue; }
            const int u = b.fv[3 * g + k], v = b.fv[3 * g + (k + 1) % 3];
            b.fv[3 * id] = u; b.fv[3 * id + 1] = v; b.fv[3 * id + 2] = apex;
            b.fn[3 * id] = n; b.fn[3 * id + 1] = kNone; b.fn[3 * id + 2] = kNone;
            set_plane(b, id);
            b.fflags[id] = 1; b.fowner[id] = kNone; b.fmax[id] = 0ull; b.fapex[id] = 0x7fffffff; b.nfhead[id] = kNone;
            b.newface[3 * g + k] = id;
            b.nfnext[id] = atomicExch(&b.nfhead[o], id);
            for (int j = 0; j < 3; ++j)
                if (b.fn[3 * n + j] == g) b.fn[3 * n + j] = id;
        }
    }
}

// sibling links: rotate around the horizon vertex v through the region's faces to the next horizon edge
__global__ void __launch_bounds__(TO_BLOCK) k_link_faces(Bufs b, int nf_before) {
    const int stride = gridDim.x * TO_BLOCK;
    for (int g = blockIdx.x * TO_BLOCK + threadIdx.x; g < nf_before; g += stride) {
        if (!(b.fflags[g] & 1)) continue;
        int o;
        if (!owned_accepted(b, g, &o)) continue;
        for (int k = 0; k < 3; ++k) {
            // NB fn[g][k] of a region face is never rewritten, so "across a horizon edge" is still decidable
            if (b.fowner[b.fn[3 * g + k]] == o) continue;
            const int my = b.newface[3 * g + k];
            if (my < 0 || my >= b.fcap) continue;
            int cur = g, e = (k + 1) % 3, steps = 0;
            while (true) {
                const int n2 = b.fn[3 * cur + e];
                if (b.fowner[n2] != o) break;
                int j = 0;
                while (j < 3 && b.fn[3 * n2 + j] != cur) ++j;
                if (j == 3 || ++steps > 4096) { b.ctrl[kCtrlError] |= kErrTopology; break; }
                cur = n2; e = (j + 1) % 3;
            }
            const int nxt = b.newface[3 * cur + e];
            if (nxt < 0 || nxt >= b.fcap) { b.ctrl[kCtrlError] |= kErrTopology; continue; }
            b.fn[3 * my + 1] = nxt;
            b.fn[3 * nxt + 2] = my;
        }
    }
}

// points of deleted faces: the apex retires as a vertex, the rest move to the new face they are farthest
// outside of, or retire inside the hull
__global__ void __launch_bounds__(TO_BLOCK) k_reassign(Bufs b) {
    const int stride = gridDim.x * TO_BLOCK;
    const int nloop = (b.m1 + stride - 1) / stride;
    for (int it = 0; it < nloop; ++it) {
        const int i = blockIdx.x * TO_BLOCK + threadIdx.x + it * stride;
        double best = 0.0; int bf = kNone;
        const int g = i < b.m1 ? b.pface[i] : kNone;
        int o;
        if (g >= 0 && owned_accepted(b, g, &o)) {
            if (i != b.fapex[o]) {
                int steps = 0;
                for (int f = b.nfhead[o]; f >= 0 && steps < (1 << 20); f = b.nfnext[f], ++steps) {
                    const double d = fdist(b, f, i);
                    if (d > best) { best = d; bf = f; }
                }
            }
            b.pface[i] = bf;  // the apex retires as a vertex; a point outside no new face retires inside the hull
        }
        wave_face_max(b, bf, dkey(best));  // feeds the apex search of the new face
    }
}

__global__ void __launch_bounds__(TO_BLOCK) k_kill_faces(Bufs b, int nf_before) {
    const int stride = gridDim.x * TO_BLOCK;
    for (int g = blockIdx.x * TO_BLOCK + threadIdx.x; g < nf_before; g += stride) {
        int o;
        if ((b.fflags[g] & 1) && owned_accepted(b, g, &o)) b.newface[3 * g] = -2;  // mark; cleared in k_commit
    }
}

__global__ void __launch_bounds__(TO_BLOCK) k_commit(Bufs b, int nf_before) {
    const int stride = gridDim.x * TO_BLOCK;
    for (int g = blockIdx.x * TO_BLOCK + threadIdx.x; g < nf_before; g += stride)
        if (b.newface[3 * g] == -2) { b.fflags[g] = 0; b.newface[3 * g] = kNone; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int nf = b.ctrl[kCtrlNFaces + 8];
        if (nf > b.fcap) nf = b.fcap;
        b.ctrl[kCtrlNFaces] = nf;
        b.ctrl[kCtrlRound] += 1;
        b.ctrl[kCtrlChanged2] = b.ctrl[kCtrlChanged];
        b.ctrl[kCtrlAnyOutside2] = b.ctrl[kCtrlAnyOutside];
        b.ctrl[kCtrlAccepted2] = b.ctrl[kCtrlAccepted];
        b.ctrl[kCtrlChanged] = 0;
        b.ctrl[kCtrlAnyOutside] = 0;
        b.ctrl[kCtrlAccepted] = 0;
    }
}

__global__ void __launch_bounds__(TO_BLOCK) k_mark_vertices(Bufs b) {
    const int nf = b.ctrl[kCtrlNFaces];
    const int stride = gridDim.x * TO_BLOCK;
    for (int f = blockIdx.x * TO_BLOCK + threadIdx.x; f < nf; f += stride)
        if (b.fflags[f] & 1) { b.vflag[b.fv[3 * f]] = 1; b.vflag[b.fv[3 * f + 1]] = 1; b.vflag[b.fv[3 * f + 2]] = 1; }
}

// ---- ordered compaction of the flagged indices (same scheme as the frustum cull) ----------------
__global__ void __launch_bounds__(TO_BLOCK) k_flag_count(const int* __restrict__ flag, int n, int* __restrict__ tile_cnt) {
    __shared__ int wave_cnt[TO_WAVES_PER_BLOCK];
    const int tile0 = blockIdx.x * 1024;
    int cnt = 0;
    for (int j = 0; j < 4; ++j) {
        const int i = tile0 + j * TO_BLOCK + threadIdx.x;
        cnt += __popcll(__ballot(i < n && flag[i] != 0));
    }
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ void __launch_bounds__(TO_BLOCK)
k_flag_write(const int* __restrict__ flag, int n, const int* __restrict__ tile_off, int* __restrict__ out, int cap) {
    __shared__ int wave_cnt[TO_WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile0 = blockIdx.x * 1024;
    int base = tile_off[blockIdx.x];
    for (int j = 0; j < 4; ++j) {
        const int i = tile0 + j * TO_BLOCK + threadIdx.x;
        const bool keep = i < n && flag[i] != 0;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        const int dst = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (keep && dst < cap) out[dst] = i;
        base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
}

// visible = all hull vertices but the last (tools.py:79); count and 0/1 mask
__global__ void __launch_bounds__(TO_BLOCK)
k_finish_visible(const int* __restrict__ idx, const int* __restrict__ total, int drop_last, int n_points,
                 int32_t* __restrict__ out_count, float* __restrict__ mask) {
    const int m = max(0, *total - (drop_last ? 1 : 0));
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = m;
    if (!mask) return;
    const int stride = gridDim.x * TO_BLOCK;
    for (int i = blockIdx.x * TO_BLOCK + threadIdx.x; i < m; i += stride) {
        const int v = idx[i];
        if (v < n_points) mask[v] = 1.0f;
    }
}

inline int nblocks(int64_t n, int cap = 2048) {
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    return (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}

// Builds the hull of pts (n,3) [+ origin]; leaves vflag set.  Synchronises the stream.
static int build(const Bufs& b, const float* pts, int n, int with_origin, hipStream_t st, int* rounds_out) {
    k_load<<<nblocks(b.m1), TO_BLOCK, 0, st>>>(b, pts, n, with_origin);
    TO_HIP_CHECK_LAUNCH();
    k_init<<<1, HULL_INIT_THREADS, 0, st>>>(b);
    TO_HIP_CHECK_LAUNCH();
    k_assign0<<<nblocks(b.m1), TO_BLOCK, 0, st>>>(b);
    k_far_arg<<<nblocks(b.m1), TO_BLOCK, 0, st>>>(b, 0);
    TO_HIP_CHECK_LAUNCH();
    int h[kCtrlInts];
    hipError_t e = hipMemcpyAsync(h, b.ctrl, sizeof(h), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    if (h[kCtrlError] & kErrFlat) return TOHIP_EINVAL;  // all points coplanar: Qhull refuses too (QH6154)
    int nf = h[kCtrlNFaces];
    const int max_rounds = 100000;
    int round = 0;
    bool careful = false;  // after a round without progress: propagate ownership to convergence (host-checked)
    int sweeps = 8;        // ownership sweeps per round, adapted from the previous round's convergence flag
    for (; round < max_rounds; ++round) {
        const int gf = nblocks(nf), gp = nblocks(b.m1);
        k_round_reset<<<gf, TO_BLOCK, 0, st>>>(b);
        TO_HIP_CHECK_LAUNCH();
        if (!careful) {
            // Fast path: a fixed number of propagation sweeps, no readback.  Unconverged ownership is safe — a
            // candidate is accepted only if it owns every face its apex sees (k_accept) — it can only cost progress.
            for (int it = 0; it < sweeps - 1; ++it) k_owner_prop<<<gf, TO_BLOCK, 0, st>>>(b);
            e = hipMemsetAsync(b.ctrl + kCtrlChanged, 0, sizeof(int), st);
            if (e != hipSuccess) return (int)e;
            k_owner_prop<<<gf, TO_BLOCK, 0, st>>>(b);  // still changing here = not converged (seen in the round's readback)
            TO_HIP_CHECK_LAUNCH();
        } else {
            while (true) {
                e = hipMemsetAsync(b.ctrl + kCtrlChanged, 0, sizeof(int), st);
                if (e != hipSuccess) return (int)e;
                k_owner_prop<<<gf, TO_BLOCK, 0, st>>>(b);
                k_owner_prop<<<gf, TO_BLOCK, 0, st>>>(b);
                TO_HIP_CHECK_LAUNCH();
                e = hipMemcpyAsync(h, b.ctrl, sizeof(h), hipMemcpyDeviceToHost, st);
                if (e != hipSuccess) return (int)e;
                e = hipStreamSynchronize(st);
                if (e != hipSuccess) return (int)e;
                if (!h[kCtrlChanged]) break;
            }
        }
        k_accept<<<gf, TO_BLOCK, 0, st>>>(b);
        k_new_faces<<<gf, TO_BLOCK, 0, st>>>(b);
        k_link_faces<<<gf, TO_BLOCK, 0, st>>>(b, nf);
        k_reassign<<<gp, TO_BLOCK, 0, st>>>(b);
        k_far_arg<<<gp, TO_BLOCK, 0, st>>>(b, nf);  // apexes of the faces created this round (ids >= nf)
        k_kill_faces<<<gf, TO_BLOCK, 0, st>>>(b, nf);
        k_commit<<<gf, TO_BLOCK, 0, st>>>(b, nf);
        TO_HIP_CHECK_LAUNCH();
        e = hipMemcpyAsync(h, b.ctrl, sizeof(h), hipMemcpyDeviceToHost, st);  // the round's one readback
        if (e != hipSuccess) return (int)e;
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) return (int)e;
        if (h[kCtrlError]) return (h[kCtrlError] & kErrCapacity) ? TOHIP_ENOSPC : TOHIP_ENOTCONV;
        if (!h[kCtrlAnyOutside2]) break;  // no point outside any face: the round was a no-op and the hull is complete
        if (h[kCtrlAccepted2] <= 0) {
            if (careful) return TOHIP_ENOTCONV;  // converged ownership always admits the best candidate: inconsistent predicates
            careful = true;
        } else {
            if (!careful) sweeps = h[kCtrlChanged2] ? (sweeps < 64 ? sweeps * 2 : 64) : (sweeps > 4 ? sweeps - 1 : 4);
            careful = false;
        }
        nf = h[kCtrlNFaces];
    }
    if (round >= max_rounds) return TOHIP_ENOTCONV;
    if (rounds_out) *rounds_out = round;
    k_mark_vertices<<<nblocks(nf), TO_BLOCK, 0, st>>>(b);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// ascending indices of the flagged points -> out (capacity cap), *total on the device
static int compact(const Bufs& b, int* out, int cap, int* total_dev, hipStream_t st) {
    const int ntiles = (b.m1 + 1023) / 1024;
    k_flag_count<<<ntiles, TO_BLOCK, 0, st>>>(b.vflag, b.m1, b.tile_cnt);
    TO_HIP_CHECK_LAUNCH();
    k_scan_tiles<<<1, TO_BLOCK, 0, st>>>(b.tile_cnt, ntiles, b.tile_off, total_dev);
    TO_HIP_CHECK_LAUNCH();
    k_flag_write<<<ntiles, TO_BLOCK, 0, st>>>(b.vflag, b.m1, b.tile_off, out, cap);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

}  // namespace hull

extern "C" size_t tohip_hpr_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    return hull::carve(nullptr, nullptr, n);
}

extern "C" int tohip_spherical_flip(const float* xyz, int64_t n, float param, float* flipped, float* radius_out,
                                    void* workspace, size_t workspace_bytes, void* stream_) {
    if (!xyz || !flipped || !workspace || n <= 0) return TOHIP_EINVAL;
    if (workspace_bytes < 256) return TOHIP_ENOSPC;
    return launch_flip(xyz, n, param, flipped, radius_out, (int*)workspace, (hipStream_t)stream_);
}

// Hull vertices (ascending) of pts (n,3), optionally with the origin appended as point n
// (tools.py:56-64).  idx capacity n+1; *count on the device.  Synchronises the stream.
extern "C" int tohip_convex_hull_vertices(const float* pts, int64_t n, int with_origin, int32_t* idx, int32_t* count,
                                          int32_t* rounds_host, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!pts || !idx || !count || !workspace || n < 4 || n > (int64_t)100000000) return TOHIP_EINVAL;
    if (workspace_bytes < hull::carve(nullptr, nullptr, n)) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    hull::Bufs b;
    hull::carve(&b, (char*)workspace, n);
    int rounds = 0;
    int rc = hull::build(b, pts, (int)n, with_origin, st, &rounds);
    if (rc != TOHIP_OK) return rc;
    if (rounds_host) *rounds_host = rounds;
    return hull::compact(b, idx, (int)n + 1, count, st);
}

extern "C" int tohip_hidden_pts_removal(const float* xyz, int64_t n, float param, int32_t* visible_idx,
                                        int32_t* visible_count, float* mask, void* workspace, size_t workspace_bytes,
                                        void* stream_) {
    if (!xyz || !visible_idx || !visible_count || !workspace || n < 4 || n > (int64_t)100000000) return TOHIP_EINVAL;
    if (workspace_bytes < hull::carve(nullptr, nullptr, n)) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    hull::Bufs b;
    hull::carve(&b, (char*)workspace, n);
    int rc = launch_flip(xyz, n, param, b.flipped, nullptr, b.flip_max, st);
    if (rc != TOHIP_OK) return rc;
    rc = hull::build(b, b.flipped, (int)n, 1, st, nullptr);
    if (rc != TOHIP_OK) return rc;
    // the hull lists at most n+1 vertices; visible_idx holds n: the last one is dropped anyway
    rc = hull::compact(b, visible_idx, (int)n, b.ctrl + hull::kCtrlChanged, st);
    if (rc != TOHIP_OK) return rc;
    if (mask) {
        hipError_t e = hipMemsetAsync(mask, 0, sizeof(float) * (size_t)n, st);
        if (e != hipSuccess) return (int)e;
    }
    hull::k_finish_visible<<<hull::nblocks(n), TO_BLOCK, 0, st>>>(visible_idx, b.ctrl + hull::kCtrlChanged, 1, (int)n,
                                                                   visible_count, mask);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}
