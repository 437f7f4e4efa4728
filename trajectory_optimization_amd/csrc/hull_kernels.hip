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
//   round    (1) per face: farthest outside point = apex candidate (one atomicMax on (float distance, ~position): apex_key)
//            (2) label propagation over the face graph: each candidate claims the connected set of
//                faces its apex sees; a hashed total order breaks overlaps
//            (3) a candidate is accepted iff it owns every face its apex sees and no better candidate
//                owns a face across its horizon -> accepted regions are pairwise non-adjacent, so the
//                insertions commute and equal the sequential result (r06: decided inside step (2)'s walk —
//                k_owner_claim's `verdict` — the k_accept launch is the careful path's)
//            (4) one new triangle per horizon edge, linked to its two siblings by rotating around the
//                shared horizon vertex; (5) points of deleted faces move to a new face or retire
//   end      no point is outside any face; hull vertices = vertices of the live faces
//
// Predicates are plain f64 (coordinate differences of f32 inputs are exact; products are rounded):
// like Qhull's own, they are not exact on nearly coplanar quadruples — DESIGN.md §6.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <map>
#include <memory>

#include "common.hpp"

namespace hull {

constexpr int kNone = -1;
constexpr int kCtrlNFaces = 0 /* published face count; [8] = staged face counter */, kCtrlChanged = 2, kCtrlError = 3,
              kCtrlAccepted = 4, kCtrlOverflow = 5 /* a sub-list of claimed faces ran full this round */,
              kCtrlNLive = 14 /* entries of the live-point list */, kCtrlInts = 16;
// The faces a round works on are kept in lists, each cut into kSubLists sub-lists with counters 128 bytes apart (appends to
// ONE counter serialise at ~90 per microsecond; a block appends to, and later walks, the sub-list blockIdx % kSubLists):
//   candidates  the faces with points outside them, two lists alternating by round parity (the tail of round r fills r+1's)
//   claimed     the faces a candidate's walk claimed this round, same parity scheme
// Counters live behind the scalars of the control block: ccnt(par, s), ocnt(par, s).
constexpr int kSubLists = 64, kCntStride = 32;
constexpr int kCtrlTotal = kCtrlInts + 2 * 2 * kSubLists * kCntStride;
constexpr int kErrCapacity = 1, kErrFlat = 2, kErrTopology = 4, kErrNaN = 8;

// what a distance test needs of a face, in one 64-byte line: the unnormalised normal, the coordinates of the face's
// first vertex (a copy: the differences below are the same numbers) and the link of the owner's new-face list
struct __attribute__((aligned(64))) FaceRec {
    double nx, ny, nz, x0, y0, z0;
    int next;
    float inv_norm;        // 1 / |n|: heights (distance in length units) rank the candidates
    int pad[2];            // [0]: head of the overflow chain of region_tab (an accepted candidate's region of more than ten faces)
};

struct Bufs {
    double *px, *py, *pz;  // M1 points
    int* pface;            // M1
    int* fv;               // 3 * fcap
    int* fn;               // 3 * fcap
    FaceRec* frec;         // fcap
    unsigned long long* fmax;  // fcap   apex_key of the face's farthest outside point (0: none)
    int* fowner;           // fcap
    unsigned long long* fprio;  // fcap   rank of a candidate this round (smaller = better), written by the previous round's tail
    int* fflags;           // fcap   bit0 alive, bit1 candidate / accepted, bit2 dies at the end of this round
    int* nfhead;           // fcap   per candidate: how many new faces its accepted region has got so far this round (region_tab)
    int* newface;          // 3 * fcap
    int* cand[2];          // fcap each: candidate faces of even / odd rounds, kSubLists sub-lists of fcap / kSubLists
    int* olist;            // fcap: faces claimed this round, same sub-list layout
    int* oclaim;           // fcap: the candidate that claimed each of them (an entry counts while that candidate still owns the face)
    int* ctrl;             // kCtrlInts
    int* vflag;            // M1
    int* tile_cnt;         // ntiles(M1)
    int* tile_off;         // ntiles(M1)
    float* flipped;        // 3 * M (only used by hidden_pts_removal)
    int* flip_max;         // max(nseg, TO_FLIP_PARTS): per-viewpoint maxima (batched) / the single cloud's block maxima (launch_flip)
    // segments: independent point sets (one hull each) laid end to end; segment s owns the expanded indices
    // [seg_off[s], seg_off[s+1]) — its points followed by ONE slot for the appended origin — and the faces it
    // grows never reference another segment, so every round kernel below serves all hulls at once
    int* seg_off;          // nseg + 1
    int* seg_status;       // nseg   0 ok, 1 fewer than 4 points (no hull), 2 flat (Qhull: QH6154), 3 NaN coordinates
    int* seg_nan;          // nseg   set by k_bbox when a coordinate of the segment is NaN
    int* seg_start;        // nseg + 1   first position of each segment in the compacted vertex list
    int* seg_cnt;          // nseg
    int* idx_all;          // M1 (batched result staging)
    // points still outside some face, in position order; shrinks as the hull grows and is compacted every few rounds, so
    // that the per-round point kernels walk the tens of thousands of live points of the late rounds, not all M1
    int *live, *live2;     // M1 each
    // spatial order: the points are worked on in Morton order of their position inside the segment's bounding box, so
    // that the lanes of a wave hold neighbouring points — which share conflict faces and new-face lists for the whole
    // build (coalesced, mostly wave-uniform reads instead of 64 scattered lists per wave).  px/py/pz, pface and the
    // faces' vertex ids live in this order; perm translates to the caller's (expanded) numbering.
    int* perm;             // M1   position -> expanded index
    unsigned long long *keys, *keys2;  // M1 each
    int* vals;             // M1
    unsigned* seg_bbox;    // 6 * nseg (ordered-float keys: min x,y,z, max x,y,z)
    char* sort_tmp;
    size_t sort_tmp_bytes;
    int m1;
    int fcap;
    int nseg;
    int hbits;             // bits of the apex height in a candidate's rank (make_prio)
    int early_out;         // k_owner_claim: candidates beaten by a neighbouring candidate do not walk
    int share_edges;       // k_owner_claim: regions that share a horizon edge convexly are accepted together (convex_across)
    int origin;            // 1: the last slot of a segment is the appended origin; 0: it repeats the segment's first point and never takes part
    int sub;               // > 1 while only every sub-th point (in Morton order) takes part: the first rounds of a large build
    int key32;             // 1: segment number + Morton cell fit 32-bit sort keys
    int serial;            // 1: the sample's hull is built by k_sample_hull (a block per segment, sequential insertion out of LDS);
                           //    the sample is then every sample_stride()-th position of a segment, counted from its first
};

// A face's farthest outside point lives in ONE word, fmax[f] (r06; until r05 a 64-bit distance there and a second pass over the live
// points to find who attains it): the distance as a float in the high word, ~position in the low one — atomicMax keeps the farthest
// point, the lowest position among equals (float rounding is monotone; any outside point is a valid apex, the farthest a good one).
// 0 = no point outside the face.
__device__ __forceinline__ unsigned long long apex_key(double d, int pos) {
    return d > 0.0 ? (((unsigned long long)__float_as_uint((float)d)) << 32) | (unsigned long long)(0xffffffffu - (unsigned)pos) : 0ull;
}
__device__ __forceinline__ int apex_pos(unsigned long long k) { return (int)(0xffffffffu - (unsigned)(k & 0xffffffffull)); }
__device__ __forceinline__ float apex_dist(unsigned long long k) { return __uint_as_float((unsigned)(k >> 32)); }

__host__ inline size_t seg(size_t bytes) { return align_up(bytes, 256); }

// Faces are never recycled within a build, so the capacity bounds the faces ever CREATED (a few per hull vertex).
// The default suits clouds whose hull is a small fraction of the points (HPR of a scene: 2-5 %); a build that runs out
// returns TOHIP_ENOSPC and the caller retries with a larger workspace — every byte beyond the fixed part is used for
// faces (faces_for_bytes), up to the never-exceeded-in-practice 8 per point.
constexpr size_t kBytesPerFace = 16 * sizeof(int) + 2 * sizeof(double) + 64;  // the per-face arrays carved below
constexpr int kFaceArrays = 13;

__host__ inline int default_face_capacity(int64_t m1, int64_t nseg) {
    int64_t c = m1 / 2 + 64 * nseg + 4096;
    if (c < 262144) c = 262144 < 8 * m1 + 1024 ? 262144 : 8 * m1 + 1024;  // small clouds: memory is no concern, a retry is
    return (int)(c > 0x3fffffff ? 0x3fffffff : c);
}

__host__ inline size_t carve(Bufs* b, char* base, int64_t n_points, int64_t nseg, int fcap) {
    const int64_t m1 = n_points + nseg;
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
    p = take(sizeof(FaceRec) * (size_t)fcap); if (b) b->frec = (FaceRec*)p;
    p = take(sizeof(unsigned long long) * (size_t)fcap); if (b) b->fmax = (unsigned long long*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->fowner = (int*)p;
    p = take(sizeof(unsigned long long) * (size_t)fcap); if (b) b->fprio = (unsigned long long*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->fflags = (int*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->nfhead = (int*)p;
    p = take(sizeof(int) * 3 * (size_t)fcap); if (b) b->newface = (int*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->cand[0] = (int*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->cand[1] = (int*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->olist = (int*)p;
    p = take(sizeof(int) * (size_t)fcap); if (b) b->oclaim = (int*)p;
    p = take(sizeof(int) * kCtrlTotal); if (b) b->ctrl = (int*)p;
    p = take(sizeof(int) * m1); if (b) b->vflag = (int*)p;
    p = take(sizeof(int) * ntiles); if (b) b->tile_cnt = (int*)p;
    p = take(sizeof(int) * ntiles); if (b) b->tile_off = (int*)p;
    p = take(sizeof(float) * 3 * (size_t)n_points); if (b) b->flipped = (float*)p;
    p = take(sizeof(int) * (nseg > TO_FLIP_PARTS ? nseg : TO_FLIP_PARTS)); if (b) b->flip_max = (int*)p;
    p = take(sizeof(int) * (nseg + 1)); if (b) b->seg_off = (int*)p;
    p = take(sizeof(int) * nseg); if (b) b->seg_status = (int*)p;
    p = take(sizeof(int) * nseg); if (b) b->seg_nan = (int*)p;
    p = take(sizeof(int) * (nseg + 1)); if (b) b->seg_start = (int*)p;
    p = take(sizeof(int) * nseg); if (b) b->seg_cnt = (int*)p;
    p = take(sizeof(int) * (nseg > 1 ? m1 : 1)); if (b) b->idx_all = (int*)p;
    p = take(sizeof(int) * m1); if (b) b->live = (int*)p;
    p = take(sizeof(int) * m1); if (b) b->live2 = (int*)p;
    p = take(sizeof(int) * m1); if (b) b->perm = (int*)p;
    p = take(sizeof(unsigned long long) * m1); if (b) b->keys = (unsigned long long*)p;
    p = take(sizeof(unsigned long long) * m1); if (b) b->keys2 = (unsigned long long*)p;
    p = take(sizeof(int) * m1); if (b) b->vals = (int*)p;
    p = take(sizeof(unsigned) * 6 * nseg); if (b) b->seg_bbox = (unsigned*)p;
    size_t tmp = 0;
    (void)sort_pairs(nullptr, tmp, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (const int*)nullptr, (int*)nullptr, (int)m1, 0, 64,
                     (hipStream_t)0);
    {
        size_t tmp32 = 0;   // up to 256 segments sort 32-bit keys
        (void)sort_pairs(nullptr, tmp32, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr, (int*)nullptr, (int)m1, 0, 32, (hipStream_t)0);
        if (tmp32 > tmp) tmp = tmp32;
    }
    p = take(tmp); if (b) { b->sort_tmp = p; b->sort_tmp_bytes = tmp; }
    if (b) { b->m1 = (int)m1; b->fcap = fcap; b->nseg = (int)nseg; b->sub = 1; b->origin = 1; b->hbits = 0; b->early_out = 0; b->share_edges = 0; b->serial = 0; b->key32 = 1; }
    return o;
}

// the largest face capacity a workspace of `bytes` holds for this input (0 if not even the minimum fits)
__host__ inline int faces_for_bytes(int64_t n_points, int64_t nseg, size_t bytes) {
    const size_t fixed = carve(nullptr, nullptr, n_points, nseg, 0) + (size_t)kFaceArrays * 256;
    if (bytes <= fixed) return 0;
    size_t f = (bytes - fixed) / kBytesPerFace;
    const size_t hi = (size_t)8 * (size_t)(n_points + nseg) + 1024;
    if (f > hi) f = hi;
    if (f > 0x3fffffff) f = 0x3fffffff;
    if ((int64_t)f < 4 * nseg + 64) return 0;
    return (int)f;
}

// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// Rank of candidate face f in round `round` (smaller = better): the top `hbits` bits of its apex's HEIGHT above the face
// (float: exponent first — higher apexes first, they bury more of the others), then a hash salted per round, then the id.
// hbits = 0: the hashed total order alone.
__device__ __forceinline__ unsigned long long make_prio(int f, int round, float height, int hbits) {
    const unsigned h = hash32((unsigned)f * 0x9E3779B9u + (unsigned)round);
    if (hbits <= 0) return ((unsigned long long)h << 32) | (unsigned)f;
    const unsigned hb = (__float_as_uint(height) & 0x7fffffffu) >> (31 - hbits);
    const unsigned inv = ((1u << hbits) - 1u) - hb;
    return ((unsigned long long)inv << 48) | ((unsigned long long)(h & 0xffffu) << 32) | (unsigned)f;
}

// an ACCEPTED candidate's record doubles as the table of its region's new faces (k_new_faces): words 0..11 = the plane it no longer needs
constexpr int kRegionTab = 12;
__device__ __forceinline__ int* region_tab(const Bufs& b, int o) { return reinterpret_cast<int*>(&b.frec[o]); }

// signed (unnormalised) distance of point i from the plane of face f; > 0 = strictly outside
__device__ __forceinline__ double plane_dist(const FaceRec& r, double x, double y, double z) {
    const double dx = x - r.x0, dy = y - r.y0, dz = z - r.z0;
    return r.nx * dx + r.ny * dy + r.nz * dz;
}
__device__ __forceinline__ double fdist(const Bufs& b, int f, int i) { return plane_dist(b.frec[f], b.px[i], b.py[i], b.pz[i]); }

// A point with the coordinates of one of the face's vertices lies ON the face whatever the rounding of its distance says (exact
// copies of a hull vertex: duplicated input rows, and the slot that repeats a segment's first point when no origin is appended)
__device__ __forceinline__ bool on_a_vertex(const Bufs& b, int f, double x, double y, double z) {
    for (int k = 0; k < 3; ++k) {
        const int v = b.fv[3 * f + k];
        if (b.px[v] == x && b.py[v] == y && b.pz[v] == z) return true;
    }
    return false;
}

__device__ __forceinline__ void set_plane(const Bufs& b, int f) {
    const int a = b.fv[3 * f], c1 = b.fv[3 * f + 1], c2 = b.fv[3 * f + 2];
    const double ax = b.px[a], ay = b.py[a], az = b.pz[a];
    const double ux = b.px[c1] - ax, uy = b.py[c1] - ay, uz = b.pz[c1] - az;
    const double vx = b.px[c2] - ax, vy = b.py[c2] - ay, vz = b.pz[c2] - az;
    FaceRec& r = b.frec[f];
    r.nx = uy * vz - uz * vy;
    r.ny = uz * vx - ux * vz;
    r.nz = ux * vy - uy * vx;
    r.x0 = ax; r.y0 = ay; r.z0 = az;
    r.next = kNone;
    r.pad[0] = kNone;      // (region_tab's overflow head: only ever written once, when the face is an accepted candidate)
    r.inv_norm = (float)(1.0 / sqrt(r.nx * r.nx + r.ny * r.ny + r.nz * r.nz));
}

// Per-face maximum of `key` over many points.  Early rounds aim a million points at a handful of faces, and
// same-address global atomics serialise (~90 per microsecond), so the maximum is reduced in three levels: in the
// wave (up to 8 distinct faces per call), in a 256-entry table in LDS shared by the block across all iterations of
// its point loop, and only then — once per block and face — in global memory.  A face that does not get a table
// slot (late rounds: thousands of faces, no contention) goes to global memory directly.
constexpr int kFaceMaxSlots = TO_BLOCK;
struct FaceMaxTable {
    unsigned long long key[kFaceMaxSlots];
    int face[kFaceMaxSlots];
};

__device__ __forceinline__ void face_max_init(FaceMaxTable& t) {
    t.key[threadIdx.x] = 0ull; t.face[threadIdx.x] = kNone;
    __syncthreads();
}

__device__ __forceinline__ void face_max_put(const Bufs& b, FaceMaxTable& t, int f, unsigned long long key) {
    const int slot = (int)(hash32((unsigned)f) & (unsigned)(kFaceMaxSlots - 1));
    const int old = atomicCAS(&t.face[slot], kNone, f);
    if (old == kNone || old == f) atomicMax(&t.key[slot], key);
    else atomicMax(&b.fmax[f], key);
}

// must be called by all lanes of the wave
__device__ __forceinline__ void wave_face_max(const Bufs& b, FaceMaxTable& t, int f, unsigned long long key) {
    unsigned long long todo = __ballot(f >= 0);
    for (int it = 0; it < 8 && todo != 0ull; ++it) {
        const int f0 = __shfl(f, __builtin_ctzll(todo));
        const bool mine = f == f0;
        unsigned long long m = mine ? key : 0ull;
        for (int s = 32; s > 0; s >>= 1) {
            const unsigned long long o = ((unsigned long long)(unsigned)__shfl_xor((int)(m >> 32), s) << 32) | (unsigned)__shfl_xor((int)m, s);
            m = o > m ? o : m;
        }
        if ((threadIdx.x & 63) == 0) face_max_put(b, t, f0, m);
        todo &= ~__ballot(mine);
        if (mine) f = kNone;
    }
    if (f >= 0) face_max_put(b, t, f, key);
}

__device__ __forceinline__ void face_max_flush(const Bufs& b, FaceMaxTable& t) {
    __syncthreads();
    if (t.face[threadIdx.x] != kNone) atomicMax(&b.fmax[t.face[threadIdx.x]], t.key[threadIdx.x]);
}

// Reserve `want` (>= 0) consecutive slots per thread from *counter with ONE atomic per block (a same-address atomic
// per lane or wave serialises); returns the thread's first slot.  Must be called by all threads of the block.
__device__ __forceinline__ int block_alloc(int* counter, int want) {
    __shared__ int wave_tot[TO_WAVES_PER_BLOCK];
    __shared__ int block_base;
    int incl = want;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int s = 1; s < 64; s <<= 1) {
        const int o = __shfl_up(incl, s);
        if (lane >= s) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < TO_WAVES_PER_BLOCK; ++w) { const int c = wave_tot[w]; total += c; if (w < wave) before += c; }
    if (threadIdx.x == 0) block_base = total > 0 ? atomicAdd(counter, total) : 0;
    __syncthreads();
    const int base = block_base;
    __syncthreads();  // block_base / wave_tot are reused by the next call
    return base + before + incl - want;
}

// segment of expanded index i: the largest s with seg_off[s] <= i
__device__ __forceinline__ int find_seg(const Bufs& b, int i) {
    int lo = 0, hi = b.nseg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (b.seg_off[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}
// The set-up kernels walk the points 64 at a time, and 64 consecutive indices share their segment but for the waves on the segments'
// 127 boundaries: the search runs ONCE per wave on the scalar unit (r06; every lane's own search was seven dependent vector loads per
// point — most of what k_bbox, k_sort_keys, k_load, k_norm_max_seg and k_flip_seg cost for 128 views).  e0 must be wave-uniform;
// *end = the first index behind the segment.  The lanes' segment: wave_seg's if e0 + 63 < *end, their own find_seg otherwise.
__device__ __forceinline__ int wave_seg(const Bufs& b, int e0, int* end) {
    const int e = __builtin_amdgcn_readfirstlane(e0);
    int lo = 0, hi = b.nseg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (b.seg_off[mid] <= e) lo = mid; else hi = mid;
    }
    *end = b.seg_off[lo + 1];
    return lo;
}
__device__ __forceinline__ int lane_seg(const Bufs& b, int e0, int e) {
    int end;
    const int s0 = wave_seg(b, e0, &end);
    return e0 + 63 < end ? s0 : find_seg(b, e);   // (wave-uniform branch)
}

__device__ __forceinline__ int* ccnt(const Bufs& b, int par, int s) { return b.ctrl + kCtrlInts + (par * kSubLists + s) * kCntStride; }
__device__ __forceinline__ int* ocnt(const Bufs& b, int par, int s) { return ccnt(b, 2 + par, s); }
__device__ __forceinline__ int sub_cap(const Bufs& b) { return b.fcap / kSubLists; }

// The faces a per-round face kernel has to look at: the round's candidates (their own faces are theirs without a claim) and
// the claimed faces — a few thousand of the hundreds of thousands alive.  Virtual block vb of nvb (a multiple of kSubLists)
// walks sub-list vb % kSubLists of both, part vb / kSubLists.  A face can show up more than once (a candidate whose face
// was claimed by a better one; a face claimed twice): the kernels are idempotent or de-duplicate (k_new_faces).
struct ListWalk {
    const int *cl, *ol, *oc;
    int ncand, total, first, step, loops;
};
__device__ __forceinline__ ListWalk list_walk(const Bufs& b, int par, int vb, int nvb) {
    ListWalk w;
    const int sl = vb % kSubLists, cap = sub_cap(b);
    w.cl = b.cand[par] + (size_t)sl * cap;
    w.ol = b.olist + (size_t)sl * cap;
    w.oc = b.oclaim + (size_t)sl * cap;
    w.ncand = min(*ccnt(b, par, sl), cap);
    w.total = w.ncand + min(*ocnt(b, par, sl), cap);  // one index space: the candidates, then the claimed faces
    w.first = (vb / kSubLists) * TO_BLOCK + threadIdx.x;
    w.step = (nvb / kSubLists) * TO_BLOCK;
    w.loops = (w.total + w.step - 1) / w.step;  // the same for every thread of the block
    return w;
}
// claimer (optional): the candidate whose entry this is — the face itself for a candidate's own entry
__device__ __forceinline__ int list_face(const ListWalk& w, int it, bool* is_cand = nullptr, int* claimer = nullptr) {
    const int j = w.first + it * w.step;
    if (is_cand) *is_cand = j < w.ncand;
    if (claimer) *claimer = j < w.ncand ? w.cl[j] : (j < w.total ? w.oc[j - w.ncand] : kNone);
    return j < w.ncand ? w.cl[j] : (j < w.total ? w.ol[j - w.ncand] : kNone);
}

__global__ void k_single_segment(Bufs b) { b.seg_off[0] = 0; b.seg_off[1] = b.m1; }

// coordinates of expanded index e of segment sg: a row of the source array (every earlier segment has spent one slot
// on its origin), or — the segment's last slot — the appended origin (tools.py:60); without an origin that slot
// repeats the segment's first point (never a new vertex)
// a row of an (n,3) f32 array in ONE 12-byte access (three 4-byte loads of a gathered row are three trips through the address unit)
struct __attribute__((packed, aligned(4))) Row3 { float x, y, z; };
__device__ __forceinline__ Row3 load_row(const float* __restrict__ xyz, int64_t r) { return reinterpret_cast<const Row3*>(xyz)[r]; }

__device__ __forceinline__ void source_point(const Bufs& b, const float* __restrict__ pts, int with_origin, int e, int sg,
                                             float* x, float* y, float* z) {
    const bool extra = e == b.seg_off[sg + 1] - 1;
    const int r = extra ? b.seg_off[sg] - sg : e - sg;
    const bool zero = extra && with_origin;
    const Row3 p = load_row(pts, r);
    *x = zero ? 0.0f : p.x;
    *y = zero ? 0.0f : p.y;
    *z = zero ? 0.0f : p.z;
}

__global__ void __launch_bounds__(TO_BLOCK) k_bbox_init(Bufs b) {
    const int i = blockIdx.x * TO_BLOCK + threadIdx.x;
    if (i < 6 * b.nseg) b.seg_bbox[i] = (i % 6) < 3 ? 0xffffffffu : 0u;
    if (i < b.nseg) b.seg_nan[i] = 0;
}

// Every wave walks ONE contiguous run of points and keeps the running box of the segment it is in in registers; the box
// goes to memory (six atomics) when the run leaves the segment and at the end — a few thousand atomics in all, where one
// set per 64 points serialised on the segments' addresses (1.6 ms for 13.6 M points).
__global__ void __launch_bounds__(TO_BLOCK) k_bbox(Bufs b, const float* __restrict__ pts, int with_origin) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * TO_WAVES_PER_BLOCK + (threadIdx.x >> 6), nwaves = gridDim.x * TO_WAVES_PER_BLOCK;
    const int chunk = ((b.m1 + nwaves - 1) / nwaves + 63) / 64 * 64;
    const int64_t begin64 = (int64_t)wave * chunk;
    const int begin = begin64 < b.m1 ? (int)begin64 : b.m1, end = begin64 + chunk < b.m1 ? (int)(begin64 + chunk) : b.m1;
    int cur = -1, cur_seg = -1, cur_end = -1;
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    auto flush = [&]() {  // wave-uniform `cur`: called by all lanes
        if (cur >= 0) {
            for (int k = 0; k < 3; ++k) {
                for (int s = 32; s > 0; s >>= 1) {
                    lo[k] = min(lo[k], (unsigned)__shfl_xor((int)lo[k], s));
                    hi[k] = max(hi[k], (unsigned)__shfl_xor((int)hi[k], s));
                }
            }
            if (lane == 0)
                for (int k = 0; k < 3; ++k) { atomicMin(&b.seg_bbox[6 * cur + k], lo[k]); atomicMax(&b.seg_bbox[6 * cur + 3 + k], hi[k]); }
        }
        for (int k = 0; k < 3; ++k) { lo[k] = 0xffffffffu; hi[k] = 0u; }
    };
    for (int e0 = begin; e0 < end; e0 += 64) {
        const int e = e0 + lane;
        int sg = -1;
        unsigned key[3] = {0u, 0u, 0u};
        bool ok[3] = {false, false, false};
        if (e0 >= cur_end) cur_seg = wave_seg(b, e0, &cur_end);   // (the run has left the segment it was in: wave-uniform)
        if (e < end) {
            sg = e0 + 63 < cur_end ? cur_seg : find_seg(b, e);
            float c[3];
            source_point(b, pts, with_origin, e, sg, &c[0], &c[1], &c[2]);
            for (int k = 0; k < 3; ++k) {
                if (c[k] == c[k]) { key[k] = fkey(c[k]); ok[k] = true; }  // NaN coordinates take no part in the box ...
                else b.seg_nan[sg] = 1;                                    // ... and void the segment (scipy: "Points cannot contain NaN")
            }
        }
        const int s0 = __shfl(sg, 0);  // lane 0 is always inside the run
        if (__all(sg == s0 || sg < 0)) {
            if (s0 != cur) { flush(); cur = s0; }
            for (int k = 0; k < 3; ++k)
                if (ok[k]) { lo[k] = min(lo[k], key[k]); hi[k] = max(hi[k], key[k]); }
        } else {  // the 64 points straddle a segment boundary: settle them one by one
            flush();
            cur = -1;
            for (int k = 0; k < 3; ++k)
                if (ok[k]) { atomicMin(&b.seg_bbox[6 * sg + k], key[k]); atomicMax(&b.seg_bbox[6 * sg + 3 + k], key[k]); }
        }
    }
    flush();
}

// ONE segment (r06): the walk above keeps a wave on one long run so that a segment's six words see few atomics — 512 waves on a
// million points, 59 us.  With a single segment every block can reduce its points in LDS and send six atomics to copy
// (block & 63) of the box — 64 copies on lines of their own in `copies` (the sort's output buffer, free until the sort) — and one
// small block folds them: 8 us.  (The pack's k_bbox does the same with its own key function.)
__global__ void __launch_bounds__(TO_BLOCK) k_bbox1_init(unsigned* __restrict__ copies) {
    for (int i = threadIdx.x; i < 64 * 32; i += TO_BLOCK) copies[i] = (i & 31) < 3 ? 0xffffffffu : 0u;   // [copy * 32 + k]: min x,y,z | max x,y,z
}
__global__ void __launch_bounds__(TO_BLOCK) k_bbox1(Bufs b, const float* __restrict__ pts, int with_origin, unsigned* __restrict__ copies) {
    __shared__ unsigned slo[3][TO_WAVES_PER_BLOCK], shi[3][TO_WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0u, 0u, 0u};
    bool nan = false;
    for (int e = blockIdx.x * TO_BLOCK + threadIdx.x; e < b.m1; e += gridDim.x * TO_BLOCK) {
        float c[3];
        source_point(b, pts, with_origin, e, 0, &c[0], &c[1], &c[2]);
        for (int k = 0; k < 3; ++k) {
            if (c[k] == c[k]) { const unsigned key = fkey(c[k]); lo[k] = min(lo[k], key); hi[k] = max(hi[k], key); }
            else nan = true;   // NaN coordinates take no part in the box and void the segment (scipy: "Points cannot contain NaN")
        }
    }
    for (int k = 0; k < 3; ++k)
        for (int s = 32; s > 0; s >>= 1) {
            lo[k] = min(lo[k], (unsigned)__shfl_xor((int)lo[k], s));
            hi[k] = max(hi[k], (unsigned)__shfl_xor((int)hi[k], s));
        }
    if (__any(nan) && lane == 0) b.seg_nan[0] = 1;
    if (lane == 0) for (int k = 0; k < 3; ++k) { slo[k][wave] = lo[k]; shi[k][wave] = hi[k]; }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x % 3;
        unsigned v = threadIdx.x < 3 ? slo[k][0] : shi[k][0];
        for (int w = 1; w < TO_WAVES_PER_BLOCK; ++w) v = threadIdx.x < 3 ? min(v, slo[k][w]) : max(v, shi[k][w]);
        unsigned* dst = copies + (blockIdx.x & 63) * 32 + threadIdx.x;
        if (threadIdx.x < 3) atomicMin(dst, v); else atomicMax(dst, v);
    }
}
__global__ void __launch_bounds__(64) k_bbox1_fold(Bufs b, const unsigned* __restrict__ copies) {
    const int lane = threadIdx.x;
    for (int k = 0; k < 6; ++k) {
        unsigned v = copies[lane * 32 + k];
        for (int s = 32; s > 0; s >>= 1) {
            const unsigned o = (unsigned)__shfl_xor((int)v, s);
            v = k < 3 ? min(v, o) : max(v, o);
        }
        if (lane == 0) b.seg_bbox[k] = v;
    }
}

__device__ __forceinline__ unsigned spread8(unsigned v) {  // 8 bits -> every third bit
    v &= 0xffu;
    v = (v | (v << 8)) & 0x0000f00fu;
    v = (v | (v << 4)) & 0x000c30c3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}
// The Morton cell of a point inside its segment's bounding box: 8 bits per axis (r06; 10 until r05 — cells of 1/256 of the box keep a
// wave's lanes on neighbouring points just as well, and 24 bits are three radix passes where 30 were four; with the segment number above
// them the keys of up to 256 segments are 32-bit words, four passes where the 64-bit keys took five).  ONE function for the sort key
// and for lowest_identical_row's walk over "equal cells": the two must agree (r05 found out with 208 ms builds).
constexpr int kMortonBits = 24;
__device__ __forceinline__ unsigned morton_of(const Bufs& b, int sg, float x, float y, float z) {
    const float c[3] = {x, y, z};
    unsigned q[3];
    for (int k = 0; k < 3; ++k) {
        const float lo = fkey_inv(b.seg_bbox[6 * sg + k]), hi = fkey_inv(b.seg_bbox[6 * sg + 3 + k]);
        const float t = (c[k] - lo) / (hi - lo) * 255.0f;  // NaN / empty extent -> 0 below
        q[k] = t >= 0.0f ? (t < 255.0f ? (unsigned)t : 255u) : 0u;
    }
    return spread8(q[0]) | (spread8(q[1]) << 1) | (spread8(q[2]) << 2);
}

// sort key: segment in the high bits (segments stay where they are), the Morton cell of the position in the segment's bounding box
// below; the value is the expanded index.  32-bit keys while segment number and cell fit (b.key32), 64-bit ones beyond 256 segments
__global__ void __launch_bounds__(TO_BLOCK) k_sort_keys(Bufs b, const float* __restrict__ pts, int with_origin) {
    const int stride = gridDim.x * TO_BLOCK;
    for (int e = blockIdx.x * TO_BLOCK + threadIdx.x; e - (int)(threadIdx.x & 63) < b.m1; e += stride) {
        const int sg = lane_seg(b, e - (int)(threadIdx.x & 63), e);   // (all lanes of the wave take part)
        if (e >= b.m1) continue;
        float c[3];
        source_point(b, pts, with_origin, e, sg, &c[0], &c[1], &c[2]);
        const unsigned m = morton_of(b, sg, c[0], c[1], c[2]);
        if (b.key32) reinterpret_cast<unsigned*>(b.keys)[e] = ((unsigned)sg << kMortonBits) | m;   // 32-bit keys sort in half the time
        else b.keys[e] = ((unsigned long long)(unsigned)sg << kMortonBits) | m;
        b.vals[e] = e;
    }
}

// the sorted key of position j (keys2 keeps the sort's output)
__device__ __forceinline__ unsigned long long sorted_key(const Bufs& b, int j) {
    return b.key32 ? (unsigned long long)reinterpret_cast<const unsigned*>(b.keys2)[j] : b.keys2[j];
}

// position j takes the point perm[j]
__global__ void __launch_bounds__(TO_BLOCK)
k_load(Bufs b, const float* __restrict__ pts, int with_origin) {
    const int stride = gridDim.x * TO_BLOCK;
    for (int j = blockIdx.x * TO_BLOCK + threadIdx.x; j < b.m1; j += stride) {
        const int e = b.perm[j];
        float x, y, z;
        source_point(b, pts, with_origin, e, (int)(sorted_key(b, j) >> kMortonBits), &x, &y, &z);   // (the key's high bits: its segment)
        b.px[j] = (double)x; b.py[j] = (double)y; b.pz[j] = (double)z;
        b.pface[j] = kNone;
        b.vflag[j] = 0;
        b.vals[j] = 0;        // (the sort's input values are free now: k_mark_vertices' "this vertex has been looked up" marks)
        b.live[j] = j;
    }
    // the control block was zero-filled by the host; no face is published before the first round's k_accept
    if (blockIdx.x == 0 && threadIdx.x == 0) { b.ctrl[kCtrlNFaces + 8] = 4 * b.nseg; b.ctrl[kCtrlNLive] = b.m1; }
}

// block-wide argmax of (key, lowest index on ties); all threads get the winner
#define HULL_INIT_THREADS 1024
// thread 0 of a segment's block: status, and the four faces of its initial tetrahedron (face slots 4*sg .. 4*sg+3)
__device__ void init_tetrahedron(const Bufs& b, int sg, bool enough, bool has_nan, int i0, int i1, int i2, int i3, double kk, double k2,
                                 double k3, double nx, double ny, double nz, double x0, double y0, double z0) {
    const int lo = b.seg_off[sg], fb = 4 * sg;
    {
        const bool flat = enough && (!(kk > 0.0) || !(k2 > 0.0) || !(k3 > 0.0));
        b.seg_status[sg] = has_nan ? 3 : (!enough ? 1 : (flat ? 2 : 0));
        if (!enough || flat) {
            // the single-hull entry points refuse, like scipy (ValueError) and Qhull (QH6154/QH6214)
            if (b.nseg == 1) b.ctrl[kCtrlError] = has_nan ? kErrNaN : kErrFlat;
            // no hull for this segment: its four face slots stay dead (never candidates, never neighbours of a live face)
            for (int f = fb; f < fb + 4; ++f) {
                for (int k = 0; k < 3; ++k) { b.fv[3 * f + k] = lo; b.fn[3 * f + k] = f; }
                b.frec[f] = FaceRec{0.0, 0.0, 0.0, b.px[lo], b.py[lo], b.pz[lo], kNone, 0.f, {kNone, 0}};
                b.fflags[f] = 0; b.fowner[f] = kNone; b.fmax[f] = 0ull; b.nfhead[f] = 0;
                b.newface[3 * f] = kNone;
            }
            return;
        }
        int a = i0, c1 = i1, c2 = i2;
        const int d = i3;
        const double s = nx * (b.px[d] - x0) + ny * (b.py[d] - y0) + nz * (b.pz[d] - z0);
        if (s > 0.0) { const int tmp = c1; c1 = c2; c2 = tmp; }  // d must lie below face (a,c1,c2)
        const int F[4][3] = {{a, c1, c2}, {c1, a, d}, {c2, c1, d}, {a, c2, d}};
        const int Nb[4][3] = {{1, 2, 3}, {0, 3, 2}, {0, 1, 3}, {0, 2, 1}};
        for (int j = 0; j < 4; ++j) {
            const int f = fb + j;
            for (int k = 0; k < 3; ++k) { b.fv[3 * f + k] = F[j][k]; b.fn[3 * f + k] = fb + Nb[j][k]; }
            set_plane(b, f);
            b.fflags[f] = 1;
            b.fowner[f] = kNone;
            b.fmax[f] = 0ull;
            b.nfhead[f] = 0;
            b.newface[3 * f] = kNone;
        }
    }
}

// A candidate of the arg-max passes is (key, caller's index, position): ties go to the caller's lowest index, and the position comes
// along in the same word (r06; until then the winner's position was looked up in an inverse permutation that k_load had to scatter,
// 4 random bytes per point).
typedef unsigned long long tie_t;
__device__ __forceinline__ tie_t tie_word(int e, int pos) { return ((tie_t)(unsigned)e << 32) | (unsigned)pos; }
constexpr tie_t kTieNone = ~0ull;
__device__ __forceinline__ int tie_pos(tie_t w) { return (int)(unsigned)(w & 0xffffffffull); }

__device__ tie_t block_argmax(double key, tie_t idx, double* skey, tie_t* sidx, double* out_key) {
    const int t = threadIdx.x;
    skey[t] = key; sidx[t] = idx;
    __syncthreads();
    for (int s = HULL_INIT_THREADS / 2; s > 0; s >>= 1) {
        if (t < s) {
            const double k2 = skey[t + s]; const tie_t i2 = sidx[t + s];
            if (k2 > skey[t] || (k2 == skey[t] && i2 < sidx[t])) { skey[t] = k2; sidx[t] = i2; }
        }
        __syncthreads();
    }
    const tie_t r = sidx[0];
    if (out_key) *out_key = skey[0];
    __syncthreads();
    return r;
}

// the k-th point of a segment's 1-in-st sample: runs of 64 consecutive points, 64 st apart — a wave reads whole lines (every st-th
// point touched as many lines as all of them: the four passes of a 128-view build moved 1.3 GB for a sample of an eighth)
__device__ __forceinline__ int sample_at(int lo, int k, int st) { return lo + (k >> 6) * (64 * st) + (k & 63); }

// initial tetrahedron of one segment per block: lowest-x point, the point farthest from it, farthest from their
// line, farthest from their plane.  Segment s owns the face slots 4s .. 4s+3.
__global__ void __launch_bounds__(HULL_INIT_THREADS) k_init(Bufs b) {
    __shared__ double skey[HULL_INIT_THREADS];
    __shared__ tie_t sidx[HULL_INIT_THREADS];
    const int t = threadIdx.x, sg = blockIdx.x, lo = b.seg_off[sg], hi = b.seg_off[sg + 1], fb = 4 * sg;
    double kk = 0.0, k2 = 0.0, k3 = 0.0, nx = 0.0, ny = 0.0, nz = 0.0, x0 = 0.0, y0 = 0.0, z0 = 0.0;
    int i0 = lo, i1 = lo, i2 = lo, i3 = lo;
    const bool has_nan = b.seg_nan[sg] != 0;
    const bool enough = hi - lo - 1 >= 4 && !has_nan;  // Qhull needs d+1 input points; the appended origin comes on top
    if (enough) {
        // Any four input points that span a volume start a hull; extreme ones start it well.  A large segment looks for them among
        // every eighth point (Morton order: a uniform sample — the four passes over 107 k points per view were 0.35 ms of a 128-view
        // build, r06) and only falls back to all points if the sample's tetrahedron is degenerate (a flat verdict must come from all).
        // ties go to the caller's lowest index (perm), whatever the internal order
        for (int st = (hi - lo >= 32768) ? 8 : 1; ; st = 1) {
            double best = -INFINITY; tie_t bi = kTieNone;
            for (int q = t, i = sample_at(lo, q, st); i < hi; q += HULL_INIT_THREADS, i = sample_at(lo, q, st)) {
                const double k = -b.px[i]; const tie_t e = tie_word(b.perm[i], i);
                if (k > best || (k == best && e < bi)) { best = k; bi = e; }
            }
            i0 = tie_pos(block_argmax(best, bi, skey, sidx, nullptr));
            x0 = b.px[i0]; y0 = b.py[i0]; z0 = b.pz[i0];
            best = -INFINITY; bi = kTieNone;
            for (int q = t, i = sample_at(lo, q, st); i < hi; q += HULL_INIT_THREADS, i = sample_at(lo, q, st)) {
                const double dx = b.px[i] - x0, dy = b.py[i] - y0, dz = b.pz[i] - z0;
                const double k = dx * dx + dy * dy + dz * dz; const tie_t e = tie_word(b.perm[i], i);
                if (k > best || (k == best && e < bi)) { best = k; bi = e; }
            }
            i1 = tie_pos(block_argmax(best, bi, skey, sidx, &kk));
            const double ex = b.px[i1] - x0, ey = b.py[i1] - y0, ez = b.pz[i1] - z0;
            best = -INFINITY; bi = kTieNone;
            for (int q = t, i = sample_at(lo, q, st); i < hi; q += HULL_INIT_THREADS, i = sample_at(lo, q, st)) {
                const double dx = b.px[i] - x0, dy = b.py[i] - y0, dz = b.pz[i] - z0;
                const double cx = dy * ez - dz * ey, cy = dz * ex - dx * ez, cz = dx * ey - dy * ex;
                const double k = cx * cx + cy * cy + cz * cz; const tie_t e = tie_word(b.perm[i], i);
                if (k > best || (k == best && e < bi)) { best = k; bi = e; }
            }
            i2 = tie_pos(block_argmax(best, bi, skey, sidx, &k2));
            const double fx = b.px[i2] - x0, fy = b.py[i2] - y0, fz = b.pz[i2] - z0;
            nx = ey * fz - ez * fy; ny = ez * fx - ex * fz; nz = ex * fy - ey * fx;
            best = -INFINITY; bi = kTieNone;
            for (int q = t, i = sample_at(lo, q, st); i < hi; q += HULL_INIT_THREADS, i = sample_at(lo, q, st)) {
                const double k = fabs(nx * (b.px[i] - x0) + ny * (b.py[i] - y0) + nz * (b.pz[i] - z0)); const tie_t e = tie_word(b.perm[i], i);
                if (k > best || (k == best && e < bi)) { best = k; bi = e; }
            }
            i3 = tie_pos(block_argmax(best, bi, skey, sidx, &k3));
            if (st == 1 || (kk > 0.0 && k2 > 0.0 && k3 > 0.0)) break;   // (block-uniform: block_argmax hands every thread the same values)
        }
    }
    if (t == 0) init_tetrahedron(b, sg, enough, has_nan, i0, i1, i2, i3, kk, k2, k3, nx, ny, nz, x0, y0, z0);
}

// ---- the same for ONE large segment, spread over many blocks: four passes over the points, each block leaving its best
// (key, caller's index) in `part`; the next launch's blocks first fold the previous pass's partials (all of them, redundantly)
// and go on.  A single block walking a million points four times was 1.5 ms of a 19 ms build.
constexpr int kInitBlocks = 128;
struct InitSel { int i[4]; double key[4]; };

__device__ __forceinline__ int init_fold(const Bufs& b, const double* __restrict__ pkey, const tie_t* __restrict__ pidx, int nparts,
                                         double* skey, tie_t* sidx, double* out_key) {
    const int t = threadIdx.x;
    double best = -INFINITY; tie_t bi = kTieNone;
    for (int j = t; j < nparts; j += HULL_INIT_THREADS) {
        const double k = pkey[j]; const tie_t e = pidx[j];
        if (k > best || (k == best && e < bi)) { best = k; bi = e; }
    }
    return tie_pos(block_argmax(best, bi, skey, sidx, out_key));
}

__global__ void __launch_bounds__(HULL_INIT_THREADS) k_init1_pass(Bufs b, int pass, double* __restrict__ pkey, tie_t* __restrict__ pidx,
                                                                  InitSel* __restrict__ sel) {
    __shared__ double skey[HULL_INIT_THREADS];
    __shared__ tie_t sidx[HULL_INIT_THREADS];
    const int t = threadIdx.x, lo = b.seg_off[0], hi = b.seg_off[1];
    const int nparts = gridDim.x;
    double x0 = 0, y0 = 0, z0 = 0, ex = 0, ey = 0, ez = 0, nx = 0, ny = 0, nz = 0;
    if (b.seg_nan[0] != 0) return;  // refused in the finish kernel; no keys to compare
    if (pass >= 1) {
        // previous pass's result (every block folds the same partials: same answer everywhere; block 0 records it)
        double key;
        const int ip = init_fold(b, pkey + (size_t)(pass - 1) * kInitBlocks, pidx + (size_t)(pass - 1) * kInitBlocks, nparts, skey, sidx, &key);
        if (blockIdx.x == 0 && t == 0) { sel->i[pass - 1] = ip; sel->key[pass - 1] = key; }
        const int i0 = pass == 1 ? ip : sel->i[0];
        x0 = b.px[i0]; y0 = b.py[i0]; z0 = b.pz[i0];
        if (pass >= 2) {
            const int i1 = pass == 2 ? ip : sel->i[1];
            ex = b.px[i1] - x0; ey = b.py[i1] - y0; ez = b.pz[i1] - z0;
        }
        if (pass >= 3) {
            const double fx = b.px[ip] - x0, fy = b.py[ip] - y0, fz = b.pz[ip] - z0;
            nx = ey * fz - ez * fy; ny = ez * fx - ex * fz; nz = ex * fy - ey * fx;
        }
    }
    double best = -INFINITY; tie_t bi = kTieNone;
    for (int i = lo + blockIdx.x * HULL_INIT_THREADS + t; i < hi; i += nparts * HULL_INIT_THREADS) {
        const double dx = b.px[i] - x0, dy = b.py[i] - y0, dz = b.pz[i] - z0;
        double k;
        if (pass == 0) k = -b.px[i];
        else if (pass == 1) k = dx * dx + dy * dy + dz * dz;
        else if (pass == 2) { const double cx = dy * ez - dz * ey, cy = dz * ex - dx * ez, cz = dx * ey - dy * ex; k = cx * cx + cy * cy + cz * cz; }
        else k = fabs(nx * dx + ny * dy + nz * dz);
        const tie_t e = tie_word(b.perm[i], i);
        if (k > best || (k == best && e < bi)) { best = k; bi = e; }
    }
    double bk;
    const tie_t r = block_argmax(best, bi, skey, sidx, &bk);
    if (t == 0) { pkey[(size_t)pass * kInitBlocks + blockIdx.x] = bk; pidx[(size_t)pass * kInitBlocks + blockIdx.x] = r; }
}

__global__ void __launch_bounds__(HULL_INIT_THREADS) k_init1_finish(Bufs b, int nparts, const double* __restrict__ pkey,
                                                                    const tie_t* __restrict__ pidx, const InitSel* __restrict__ sel) {
    __shared__ double skey[HULL_INIT_THREADS];
    __shared__ tie_t sidx[HULL_INIT_THREADS];
    if (b.seg_nan[0] != 0) {
        const int lo = b.seg_off[0];
        if (threadIdx.x == 0) init_tetrahedron(b, 0, false, true, lo, lo, lo, lo, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0);
        return;
    }
    double k3;
    const int i3 = init_fold(b, pkey + (size_t)3 * kInitBlocks, pidx + (size_t)3 * kInitBlocks, nparts, skey, sidx, &k3);
    if (threadIdx.x == 0) {
        const int i0 = sel->i[0], i1 = sel->i[1], i2 = sel->i[2];
        const double x0 = b.px[i0], y0 = b.py[i0], z0 = b.pz[i0];
        const double ex = b.px[i1] - x0, ey = b.py[i1] - y0, ez = b.pz[i1] - z0;
        const double fx = b.px[i2] - x0, fy = b.py[i2] - y0, fz = b.pz[i2] - z0;
        init_tetrahedron(b, 0, true, false, i0, i1, i2, i3, sel->key[1], sel->key[2], k3, ey * fz - ez * fy, ez * fx - ex * fz,
                         ex * fy - ey * fx, x0, y0, z0);
    }
}

__global__ void __launch_bounds__(TO_BLOCK) k_assign0(Bufs b) {
    __shared__ FaceMaxTable tab;
    face_max_init(tab);
    const int stride = gridDim.x * TO_BLOCK;
    const int nloop = (b.m1 + stride - 1) / stride;
    for (int it = 0; it < nloop; ++it) {
        const int i = blockIdx.x * TO_BLOCK + threadIdx.x + it * stride;
        double best = 0.0; int bf = kNone;
        if (i < b.m1) {
            const int sgi = find_seg(b, i), fb = 4 * sgi;
            // the tetrahedron's corners: faces fb and fb+1 are (a,c1,c2) and (c1,a,d)
            const bool filler = !b.origin && b.perm[i] == b.seg_off[sgi + 1] - 1;
            if ((b.fflags[fb] & 1) && i % b.sub == 0 && !filler && !(i == b.fv[3 * fb] || i == b.fv[3 * fb + 1] || i == b.fv[3 * fb + 2] || i == b.fv[3 * fb + 5])) {
                for (int f = fb; f < fb + 4; ++f) {
                    const double d = fdist(b, f, i);
                    if (d > best && !on_a_vertex(b, f, b.px[i], b.py[i], b.pz[i])) { best = d; bf = f; }
                }
                b.pface[i] = bf;
            }
        }
        wave_face_max(b, tab, bf, apex_key(best, i));
    }
    face_max_flush(b, tab);
}

// ---- round --------------------------------------------------------------------------------------
// Four launches (r06: the claim walk gives the verdict; five with k_accept in the careful path): claim | new faces | link + reassign | tail,
// all of them over the round's face LISTS (candidates and
// claimed faces, list_walk) and the live points — never over every face of the hull.  `par` is the round's parity (which of
// the two candidate lists it reads), `round` salts the candidates' ranking.  What each reads of the control block:
//   kCtrlNFaces      published face count = the faces that existed before THIS round's insertions (written by k_accept)
//   kCtrlNFaces + 8  staged count: every face created so far (k_new_faces allocates from it)
//   kCtrlAccepted    regions accepted this round
// The tail prepares the NEXT round (per-face reset, next candidate list), so a round has no reset launch of its own.

// The ownership propagation of a round in ONE launch, one WAVE per candidate: breadth first over the region the candidate's
// apex sees, the frontier in LDS, every (frontier face, edge) pair on its own lane — a level costs one chain of dependent
// loads (neighbour id -> its owner and plane -> the claim) whatever the frontier's size, where one thread per candidate
// paid that chain once per face.  A face is claimed unless a better candidate holds it (worse ones are robbed); every claim
// is logged (LDS, flushed to the block's sub-list of claimed faces with one counter update per candidate).  Which faces end
// up with a LOSING candidate depends on arrival order, the winners' regions do not: k_accept admits a candidate only if it
// owns every face its apex sees and borders no better region, so an incomplete walk (frontier or claim budget exhausted, a
// face stolen later) can only cost that candidate this round; a round that accepts nobody is repeated with k_owner_prop
// run to convergence.
constexpr int kClaimFront = 128, kClaimMax = 4096, kClaimLog = 192;

// Two regions that share a horizon edge (u, v) can BOTH go in this round when the new faces on it, (u, v, mine) and (v, u, theirs),
// meet convexly: their apex lies strictly below the plane of my new face (and then mine below theirs).  In the sequential order
// either apex, put in second, then sees exactly its own region — what an apex sees is connected across edges, the faces across its
// horizon are the old ones (unchanged) or such a new face — so the two insertions commute like those of regions that do not touch
// (r06: until then any two adjacent regions excluded each other, and a round accepted 7 % of its candidates).  Each side asks with
// its own rounding; a side that says no fails the worse of the two, and failing is always safe.  `cg`, `k`: my region's face and its
// edge on the horizon; (ax, ay, az): my apex; `co`: the candidate that holds the face across.
constexpr double kConvexTol = 1e-9;   // relative: a pair within rounding of coplanar waits a round
__device__ __forceinline__ bool convex_across(const Bufs& b, int cg, int k, double ax, double ay, double az, int co) {
    const int u = b.fv[3 * cg + k], v = b.fv[3 * cg + (k == 2 ? 0 : k + 1)];
    const unsigned long long theirs = b.fmax[co];
    if (theirs == 0ull) return false;
    const int q = apex_pos(theirs);
    const double ux = b.px[u], uy = b.py[u], uz = b.pz[u];
    const double ex = b.px[v] - ux, ey = b.py[v] - uy, ez = b.pz[v] - uz;
    const double wx = ax - ux, wy = ay - uy, wz = az - uz;
    const double tx = b.px[q] - ux, ty = b.py[q] - uy, tz = b.pz[q] - uz;
    const double nx = ey * wz - ez * wy, ny = ez * wx - ex * wz, nz = ex * wy - ey * wx;
    const double d = nx * tx + ny * ty + nz * tz;
    return d < 0.0 && d * d > (kConvexTol * kConvexTol) * (nx * nx + ny * ny + nz * nz) * (tx * tx + ty * ty + tz * tz);
}
// verdict != 0 (r06, the fast path): the walk also DECIDES who is accepted, so the round needs no k_accept launch.  k_accept's rule —
// a candidate stays accepted iff it owns every face its apex sees and no better candidate owns a face across its horizon — is
// settled where ownership changes hands: a walk that meets a face it sees in better hands, or a better owner across its horizon,
// fails itself; a walk that robs a face, or finds a worse owner across its horizon, fails that one; an incomplete walk fails itself.
// Every claimed face's three neighbours are looked at AFTER the claim's CAS has returned (the next level of the walk), and claims
// and looks are atomics at the memory side: of two candidates that take adjacent faces at the same time at least one sees the
// other, whichever order the CASes land in.  The verdict is what k_accept computed from the final ownership: the accepted bit
// (fflags bit 1) of the candidates nobody failed.  Block 0 does k_accept's housekeeping.
__global__ void __launch_bounds__(TO_BLOCK) k_owner_claim(Bufs b, int round, int par, int verdict) {
    __shared__ int fr[TO_WAVES_PER_BLOCK][2][kClaimFront];
    if (b.ctrl[kCtrlError] != 0) return;   // (see round_dead)
    if (verdict && blockIdx.x == 0) {
        // nothing has been inserted yet this round: the staged count is the face count; it is published here for the
        // kernels that run while k_new_faces raises the staged one.  The next round's lists start empty.
        if (threadIdx.x == 0) { b.ctrl[kCtrlNFaces] = min(b.ctrl[kCtrlNFaces + 8], b.fcap); b.ctrl[kCtrlAccepted] = 0; }
        if (threadIdx.x < kSubLists) { *ccnt(b, par ^ 1, threadIdx.x) = 0; *ocnt(b, par ^ 1, threadIdx.x) = 0; }
    }
    __shared__ int lg[TO_WAVES_PER_BLOCK][kClaimLog];
    __shared__ int lgc[TO_WAVES_PER_BLOCK][kClaimLog];   // who claimed it
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int sl = blockIdx.x % kSubLists, cap = sub_cap(b);
    const int* __restrict__ cand = b.cand[par] + (size_t)sl * cap;
    int* __restrict__ own = b.olist + (size_t)sl * cap;
    int* __restrict__ ownc = b.oclaim + (size_t)sl * cap;
    int* own_n = ocnt(b, par, sl);
    const int ncand = min(*ccnt(b, par, sl), cap);
    const int wstep = (gridDim.x / kSubLists) * TO_WAVES_PER_BLOCK;
    int logn = 0;
    auto flush = [&]() {  // wave-uniform
        if (logn == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(own_n, logn);
        base = __shfl(base, 0);
        for (int i = lane; i < logn; i += 64) {
            if (base + i < cap) { own[base + i] = lg[wid][i]; ownc[base + i] = lgc[wid][i]; }
            else { b.ctrl[kCtrlOverflow] = 1; b.ctrl[kCtrlError] |= kErrCapacity; }  // k_accept turns everybody down; the build ends
        }
        logn = 0;
    };
    for (int c = (blockIdx.x / kSubLists) * TO_WAVES_PER_BLOCK + wid; c < ncand; c += wstep) {
        const int o = cand[c];
        const unsigned long long ax = b.fmax[o];
        const int n0 = (b.early_out && lane < 3) ? b.fn[3 * o + lane] : kNone;   // for the look at the neighbours below: requested with the apex
        if (ax == 0ull) {  // a candidate without a point outside its face (never seen): not a candidate
            if (lane == 0) { b.fowner[o] = kNone; atomicAnd(&b.fflags[o], ~2); }
            continue;
        }
        if (__hip_atomic_load(&b.fowner[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != o) continue;  // taken by a better one
        const unsigned long long po = b.fprio[o];
        // the apex's coordinates are requested before the look at the neighbours below, not after it: two chains of dependent
        // loads side by side
        const int apex = apex_pos(ax);
        const double px = b.px[apex], py = b.py[apex], pz = b.pz[apex];
        {
            // A candidate whose own face is seen by the apex of a BETTER candidate next door has lost before it starts: that
            // neighbour's walk takes this face at its first level.  Not walking keeps its claims off the faces that worse
            // candidates need (experiments: TOHIP_HULL_EARLY_OUT=0 switches it off)
            bool doomed = false;
            if (n0 >= 0) {
                const int n = n0;
                if ((b.fflags[n] & 3) == 3 && b.fprio[n] < po) {
                    const unsigned long long an = b.fmax[n];
                    if (an != 0ull) {
                        const int pa = apex_pos(an);
                        doomed = plane_dist(b.frec[o], b.px[pa], b.py[pa], b.pz[pa]) > 0.0;
                    }
                }
            }
            if (__any(doomed)) {
                if (verdict && lane == 0) atomicAnd(&b.fflags[o], ~2);
                continue;
            }
        }
        int cur = 0, ncur = 1, claimed = 0;
        bool failed = false;   // (wave-uniform)
        if (lane == 0) fr[wid][0][0] = o;
        while (ncur > 0 && claimed < kClaimMax) {
            int nnext = 0;
            for (int base = 0; base < 3 * ncur; base += 64) {
                const int t = base + lane;
                bool mine = false;
                int n = kNone;
                bool fail_me = false;
                int fail_other = kNone;
                if (t < 3 * ncur) {
                    const int cg = fr[wid][cur][t / 3];
                    n = b.fn[3 * cg + t % 3];
                    int co = __hip_atomic_load(&b.fowner[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool sees = plane_dist(b.frec[n], px, py, pz) > 0.0;  // loaded alongside the owner, not after it
                    if (co >= b.fcap) co = kNone;   // (never: an owner is a face id)
                    if (sees && !(co == o || (co >= 0 && b.fprio[co] <= po))) {
                        while (true) {
                            const int old = atomicCAS(&b.fowner[n], co, o);
                            if (old == co) { mine = true; fail_other = co; break; }  // two lanes on the same face: the second finds `o` there; a robbed owner has lost
                            co = old;
                            if (co == o) break;
                            if (co >= 0 && b.fprio[co] <= po) { fail_me = true; break; }   // a better one was quicker
                        }
                    } else if (sees) {
                        fail_me = co != o;                     // a face my apex sees, in better hands
                    } else if (co >= 0 && co != o && verdict) { // across my horizon, in other hands: unless the two fit, the worse one loses
                        if (!b.share_edges || !convex_across(b, cg, t % 3, px, py, pz, co)) {
                            if (b.fprio[co] < po) fail_me = true; else fail_other = co;
                        }
                    }
                }
                if (verdict) {
                    if (fail_other >= 0) atomicAnd(&b.fflags[fail_other], ~2);
                    failed = failed || __any(fail_me);
                }
                const unsigned long long bal = __ballot(mine);
                const int cnt = __popcll(bal), rank = __popcll(bal & ((1ull << lane) - 1ull));
                if (logn + cnt > kClaimLog) flush();
                if (mine) { lg[wid][logn + rank] = n; lgc[wid][logn + rank] = o; }
                logn += cnt;
                if (mine && nnext + rank < kClaimFront) fr[wid][cur ^ 1][nnext + rank] = n;  // beyond: the walk stays incomplete (safe)
                nnext += cnt;
            }
            claimed += nnext;
            if (nnext > kClaimFront) failed = true;   // faces were claimed whose neighbours nobody will look at: an incomplete walk
            ncur = nnext < kClaimFront ? nnext : kClaimFront;
            cur ^= 1;
        }
        if (ncur > 0) failed = true;                  // the claim budget ran out with a frontier left
        if (verdict && failed && lane == 0) atomicAnd(&b.fflags[o], ~2);
    }
    flush();
}

// The same walk for rounds with TENS OF THOUSANDS of candidates (a batch of views: up to 150 k per round), four or eight candidates
// to a wave (r06; eight is the default).  There a wave's walk is a chain of dependent accesses (neighbour -> owner and plane -> CAS, ~4.5 us per candidate) on
// the 10-20 lanes its region's edges fill, every wave of the chip holds one, and the launch is as long as the 18 candidates a wave
// walks one after the other (80 us per round for 128 views).  Regions are small in such rounds, so each QUARTER (or eighth) of a wave
// takes a candidate of its own: sixteen (eight) (face, edge) pairs per step, a frontier of at most 32 faces per level — a walk that outgrows it is
// incomplete: the candidate fails itself and comes back in a later round — four chains in flight per wave.  The four walks are
// independent but move in lock-step through one instruction stream: every ballot is the whole wave's, a quarter reads its bits of
// it.  Same claims, same log, same verdict rule as k_owner_claim (verdict = 1 always: this kernel is the fast path's).
constexpr int kSubFront = 32;
template <int SL>   // lanes per candidate: 16 (four to a wave) or 8 (eight)
__global__ void __launch_bounds__(TO_BLOCK) k_owner_claim_sub(Bufs b, int round, int par) {
    constexpr int NG = 64 / SL;
    __shared__ int fr[TO_WAVES_PER_BLOCK][NG][2][kSubFront];
    __shared__ int lg[TO_WAVES_PER_BLOCK][kClaimLog];
    __shared__ int lgc[TO_WAVES_PER_BLOCK][kClaimLog];
    if (b.ctrl[kCtrlError] != 0) return;   // (see round_dead)
    if (blockIdx.x == 0) {   // k_accept's housekeeping (see k_owner_claim)
        if (threadIdx.x == 0) { b.ctrl[kCtrlNFaces] = min(b.ctrl[kCtrlNFaces + 8], b.fcap); b.ctrl[kCtrlAccepted] = 0; }
        if (threadIdx.x < kSubLists) { *ccnt(b, par ^ 1, threadIdx.x) = 0; *ocnt(b, par ^ 1, threadIdx.x) = 0; }
    }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, grp = lane / SL, gl = lane % SL;
    const unsigned long long gm = ((1ull << SL) - 1ull) << (SL * grp), below = (1ull << lane) - 1ull;
    const int sl = blockIdx.x % kSubLists, cap = sub_cap(b);
    const int* __restrict__ cand = b.cand[par] + (size_t)sl * cap;
    int* __restrict__ own = b.olist + (size_t)sl * cap;
    int* __restrict__ ownc = b.oclaim + (size_t)sl * cap;
    int* own_n = ocnt(b, par, sl);
    const int ncand = min(*ccnt(b, par, sl), cap);
    const int cstep = (gridDim.x / kSubLists) * TO_WAVES_PER_BLOCK * NG;
    int c = ((blockIdx.x / kSubLists) * TO_WAVES_PER_BLOCK + wid) * NG + grp;   // this quarter's next candidate
    int logn = 0;
    auto flush = [&]() {  // wave-uniform
        if (logn == 0) return;
        int base = 0;
        if (lane == 0) base = atomicAdd(own_n, logn);
        base = __shfl(base, 0);
        for (int i = lane; i < logn; i += 64) {
            if (base + i < cap) { own[base + i] = lg[wid][i]; ownc[base + i] = lgc[wid][i]; }
            else { b.ctrl[kCtrlOverflow] = 1; b.ctrl[kCtrlError] |= kErrCapacity; }
        }
        logn = 0;
    };
    // a quarter's walk (the same values in its sixteen lanes)
    bool active = false, failed = false;
    int o = kNone, cur = 0, ncur = 0, nnext = 0, claimed = 0, base = 0;
    unsigned long long po = 0ull;
    double px = 0.0, py = 0.0, pz = 0.0;
    while (true) {
        // ---- quarters without a walk take their next candidate (one try per turn)
        const bool fetch = !active && c < ncand;
        if (!__any(active || fetch)) break;
        {
            int oc = kNone;
            unsigned long long ax = 0ull;
            bool ok = false, doomed = false;
            if (fetch) {
                oc = cand[c];
                ax = b.fmax[oc];
                const int n0 = (b.early_out && gl < 3) ? b.fn[3 * oc + gl] : kNone;
                if (ax == 0ull) {   // (never seen: a candidate without a point outside its face)
                    if (gl == 0) { b.fowner[oc] = kNone; atomicAnd(&b.fflags[oc], ~2); }
                } else if (__hip_atomic_load(&b.fowner[oc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == oc) {
                    ok = true;
                    po = b.fprio[oc];
                    const int apex = apex_pos(ax);
                    px = b.px[apex]; py = b.py[apex]; pz = b.pz[apex];
                    if (n0 >= 0 && (b.fflags[n0] & 3) == 3 && b.fprio[n0] < po) {   // a better candidate next door whose apex sees this face
                        const unsigned long long an = b.fmax[n0];
                        if (an != 0ull) {
                            const int pa = apex_pos(an);
                            doomed = plane_dist(b.frec[oc], b.px[pa], b.py[pa], b.pz[pa]) > 0.0;
                        }
                    }
                }
                c += cstep;
            }
            const bool gd = (__ballot(doomed) & gm) != 0ull;   // (the whole wave's ballot; this quarter's bits)
            if (fetch && ok) {
                if (gd) { if (gl == 0) atomicAnd(&b.fflags[oc], ~2); }
                else {
                    o = oc; active = true; failed = false; cur = 0; ncur = 1; nnext = 0; claimed = 0; base = 0;
                    if (gl == 0) fr[wid][grp][0][0] = o;
                }
            }
        }
        // ---- one step of every walking quarter: sixteen (face, edge) pairs of its current level
        const int t = base + gl;
        bool mine = false, fail_me = false;
        int n = kNone, fail_other = kNone;
        if (active && t < 3 * ncur) {
            const int cg = fr[wid][grp][cur][t / 3];
            n = b.fn[3 * cg + t % 3];
            int co = __hip_atomic_load(&b.fowner[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool sees = plane_dist(b.frec[n], px, py, pz) > 0.0;
            if (co >= b.fcap) co = kNone;   // (never: an owner is a face id)
            if (sees && !(co == o || (co >= 0 && b.fprio[co] <= po))) {
                while (true) {
                    const int old = atomicCAS(&b.fowner[n], co, o);
                    if (old == co) { mine = true; fail_other = co; break; }
                    co = old;
                    if (co == o) break;
                    if (co >= 0 && b.fprio[co] <= po) { fail_me = true; break; }
                }
            } else if (sees) {
                fail_me = co != o;
            } else if (co >= 0 && co != o) {   // (see k_owner_claim)
                if (!b.share_edges || !convex_across(b, cg, t % 3, px, py, pz, co)) {
                    if (b.fprio[co] < po) fail_me = true; else fail_other = co;
                }
            }
        }
        if (fail_other >= 0) atomicAnd(&b.fflags[fail_other], ~2);
        const unsigned long long bal = __ballot(mine), balf = __ballot(fail_me);
        const int total = __popcll(bal);
        if (logn + total > kClaimLog) flush();
        if (mine) { const int r = logn + __popcll(bal & below); lg[wid][r] = n; lgc[wid][r] = o; }
        logn += total;
        if (active) {
            const int cnt = __popcll(bal & gm), rank = __popcll(bal & gm & below);
            if (mine && nnext + rank < kSubFront) fr[wid][grp][cur ^ 1][nnext + rank] = n;   // beyond: the walk stays incomplete
            nnext += cnt;
            failed = failed || (balf & gm) != 0ull;
            base += SL;
            if (base >= 3 * ncur) {   // the level is done (the same in all lanes of the quarter)
                claimed += nnext;
                if (nnext > kSubFront) failed = true;
                ncur = nnext < kSubFront ? nnext : kSubFront;
                cur ^= 1; nnext = 0; base = 0;
                if (ncur == 0 || claimed >= kClaimMax) {
                    if (ncur > 0) failed = true;
                    if (failed && gl == 0) atomicAnd(&b.fflags[o], ~2);
                    active = false;
                }
            }
        }
    }
    flush();
}

// The careful path (after a round that accepted nobody): each live face adopts the best-priority owner among its neighbours
// whose apex sees it, launched until nothing changes (host-checked); k_owned_list then lists the owned faces.  Both walk
// every face ever created — rare enough not to deserve a list of the alive ones.
__global__ void __launch_bounds__(TO_BLOCK) k_owner_prop(Bufs b, int round) {
    const int nf = min(b.ctrl[kCtrlNFaces + 8], b.fcap);
    const int stride = gridDim.x * TO_BLOCK;
    bool changed = false;
    for (int g = blockIdx.x * TO_BLOCK + threadIdx.x; g < nf; g += stride) {
        if (!(b.fflags[g] & 1)) continue;
        int best = b.fowner[g];
        if (best == g && b.fmax[g] == 0ull) { b.fowner[g] = kNone; atomicAnd(&b.fflags[g], ~2); continue; }
        unsigned long long bp = best >= 0 ? b.fprio[best] : ~0ull;
        for (int k = 0; k < 3; ++k) {
            const int o = b.fowner[b.fn[3 * g + k]];
            if (o < 0 || o == best || b.fmax[o] == 0ull) continue;
            const unsigned long long op = b.fprio[o];
            if (op < bp && fdist(b, g, apex_pos(b.fmax[o])) > 0.0) { best = o; bp = op; }
        }
        if (best != b.fowner[g]) { b.fowner[g] = best; changed = true; }
    }
    if (__any(changed) && (threadIdx.x & 63) == 0) b.ctrl[kCtrlChanged] = 1;
}

__global__ void __launch_bounds__(TO_BLOCK) k_owned_list(Bufs b, int par) {
    const int nf = min(b.ctrl[kCtrlNFaces + 8], b.fcap);
    const int sl = blockIdx.x % kSubLists, cap = sub_cap(b);
    const int stride = gridDim.x * TO_BLOCK;
    const int nloop = (nf + stride - 1) / stride;
    for (int it = 0; it < nloop; ++it) {
        const int g = blockIdx.x * TO_BLOCK + threadIdx.x + it * stride;
        const bool put = g < nf && (b.fflags[g] & 1) && b.fowner[g] >= 0 && b.fowner[g] != g;
        const int slot = block_alloc(ocnt(b, par, sl), put ? 1 : 0);
        if (put) {
            if (slot < cap) { b.olist[(size_t)sl * cap + slot] = g; b.oclaim[(size_t)sl * cap + slot] = b.fowner[g]; }
            else b.ctrl[kCtrlError] |= kErrCapacity;
        }
    }
}

__global__ void __launch_bounds__(TO_BLOCK) k_accept(Bufs b, int round, int par) {
    if (b.ctrl[kCtrlError] != 0) return;   // (see round_dead)
    if (blockIdx.x == 0) {
        // nothing has been inserted yet this round: the staged count is the face count; it is published here for the
        // kernels that run while k_new_faces raises the staged one.  The next round's lists start empty.
        if (threadIdx.x == 0) { b.ctrl[kCtrlNFaces] = min(b.ctrl[kCtrlNFaces + 8], b.fcap); b.ctrl[kCtrlAccepted] = 0; }
        if (threadIdx.x < kSubLists) { *ccnt(b, par ^ 1, threadIdx.x) = 0; *ocnt(b, par ^ 1, threadIdx.x) = 0; }
    }
    const bool overflow = b.ctrl[kCtrlOverflow] != 0;  // a claimed face is missing from the lists: nobody can be checked
    const ListWalk w = list_walk(b, par, blockIdx.x, gridDim.x);
    for (int it = 0; it < w.loops; ++it) {
        const int g = list_face(w, it);
        if (g < 0 || !(b.fflags[g] & 1)) continue;
        const int o = b.fowner[g];
        if (o < 0) continue;
        if (b.fowner[o] != o) { continue; }  // o lost its own face: not a candidate (owned_accepted() checks the same)
        const int apex = apex_pos(b.fmax[o]);
        bool ok = !overflow;
        for (int k = 0; k < 3; ++k) {
            const int n = b.fn[3 * g + k];
            const int on = b.fowner[n];
            if (on == o) continue;
            if (fdist(b, n, apex) > 0.0) ok = false;                                  // visible but not ours
            else if (on >= 0 && b.fprio[on] < b.fprio[o]) ok = false;                 // adjacent to a better region
        }
        if (!ok) atomicAnd(&b.fflags[o], ~2);
    }
}


__device__ __forceinline__ bool owned_accepted(const Bufs& b, int g, int* owner) {
    const int o = b.fowner[g];
    *owner = o;
    return o >= 0 && b.fowner[o] == o && (b.fflags[o] & 2);
}

// one new triangle (u, v, apex) per horizon edge (u, v) of an accepted region; the region's faces get their death mark
// (bit 2; the alive bit goes in the tail, when nobody looks at the region any more).  A face can be listed more than once (claimed,
// then taken by a better candidate): the entry that counts is the one made by the candidate that owns the face now — exactly one
// (a candidate claims a face at most once), decided without an atomic whose answer the thread would have to wait for.
__global__ void __launch_bounds__(TO_BLOCK) k_new_faces(Bufs b, int par) {
    // An error (face capacity, a claimed-face list that ran full, inconsistent topology) ends the build — but the host only learns of
    // it from its next readback, up to two batches of rounds later, and a round cut short by the capacity leaves horizon edges without
    // faces, links unset and `newface` words stale: walking that structure reads owners out of records nobody wrote (r06: a memory
    // fault in tools/stress_hpr_repeat.py, on the build that runs out of faces and is retried).  So every round kernel leaves at
    // once when the error word is set: the rounds behind the failing one do nothing (round_dead).
    if (b.ctrl[kCtrlError] != 0 || b.ctrl[kCtrlOverflow] != 0) return;
    const ListWalk w = list_walk(b, par, blockIdx.x, gridDim.x);
    for (int it = 0; it < w.loops; ++it) {
        int claimer = kNone;
        const int g = list_face(w, it, nullptr, &claimer);
        int o = kNone, want = 0;
        bool hor[3] = {false, false, false}, ndies[3] = {false, false, false};
        const int fl = g >= 0 ? b.fflags[g] : 0;
        // Everything the new faces need is asked for BEFORE the ids are (block_alloc waits for a returning atomic and two barriers;
        // r05 loaded apex, vertices and the neighbours' links behind it: three more dependent accesses per round): the face's
        // neighbours and vertices, their coordinates, the apex and its coordinates, the links of the neighbours across the horizon
        int nn[3] = {kNone, kNone, kNone}, vv[3] = {0, 0, 0}, nfn[3][3], apex = 0;
        double vx[3] = {0, 0, 0}, vy[3] = {0, 0, 0}, vz[3] = {0, 0, 0}, ax = 0, ay = 0, az = 0;
        if (g >= 0 && (fl & 1) && owned_accepted(b, g, &o) && claimer == o) {
            b.fflags[g] = fl | 4;   // nobody else writes this face's flags in this launch
            for (int k = 0; k < 3; ++k) { nn[k] = b.fn[3 * g + k]; vv[k] = b.fv[3 * g + k]; }
            int on[3];
            for (int k = 0; k < 3; ++k) { on[k] = b.fowner[nn[k]]; hor[k] = on[k] != o; want += hor[k]; }
            // a face across the horizon that dies in ANOTHER accepted region this round (regions that share an edge, convex_across):
            // its links stay as they are — both regions' walks read them — and the two new faces on the edge find each other in link_faces
            for (int k = 0; k < 3; ++k) ndies[k] = hor[k] && on[k] >= 0 && b.fowner[on[k]] == on[k] && (b.fflags[on[k]] & 2);
            if (want) {
                apex = apex_pos(b.fmax[o]);
                ax = b.px[apex]; ay = b.py[apex]; az = b.pz[apex];
                for (int k = 0; k < 3; ++k) { vx[k] = b.px[vv[k]]; vy[k] = b.py[vv[k]]; vz[k] = b.pz[vv[k]]; }
                for (int k = 0; k < 3; ++k)
                    for (int j = 0; j < 3; ++j) nfn[k][j] = hor[k] ? b.fn[3 * nn[k] + j] : kNone;
            }
        } else {
            o = kNone;
        }
        const unsigned long long acc = __ballot(o >= 0 && g == o);
        if (acc != 0ull && (threadIdx.x & 63) == 0) atomicAdd(&b.ctrl[kCtrlAccepted], __popcll(acc));
        int id = block_alloc(&b.ctrl[kCtrlNFaces + 8], want);
        if (want == 0) continue;
        for (int k = 0; k < 3; ++k) {
            if (!hor[k]) continue;
            if (id >= b.fcap) { b.ctrl[kCtrlError] |= kErrCapacity; break; }
            const int n = nn[k], k1 = (k + 1) % 3;
            const int u = vv[k], v = vv[k1];
            const int slot = atomicAdd(&b.nfhead[o], 1);   // the region's table of new faces: asked for first, stored last
            b.fv[3 * id] = u; b.fv[3 * id + 1] = v; b.fv[3 * id + 2] = apex;
            b.fn[3 * id] = n; b.fn[3 * id + 1] = kNone; b.fn[3 * id + 2] = kNone;
            {   // set_plane's expression on the coordinates already in hand: a = u, c1 = v, c2 = the apex
                const double ux = vx[k1] - vx[k], uy = vy[k1] - vy[k], uz = vz[k1] - vz[k];
                const double wx = ax - vx[k], wy = ay - vy[k], wz = az - vz[k];
                FaceRec r;
                r.nx = uy * wz - uz * wy; r.ny = uz * wx - ux * wz; r.nz = ux * wy - uy * wx;
                r.x0 = vx[k]; r.y0 = vy[k]; r.z0 = vz[k];
                r.next = kNone; r.pad[0] = kNone; r.pad[1] = 0;
                r.inv_norm = (float)(1.0 / sqrt(r.nx * r.nx + r.ny * r.ny + r.nz * r.nz));
                b.frec[id] = r;
            }
            b.fflags[id] = 1; b.fowner[id] = kNone; b.fmax[id] = 0ull; b.nfhead[id] = 0;
            b.newface[3 * g + k] = id;
            for (int j = 0; j < 3; ++j)
                if (nfn[k][j] == g && !ndies[k]) b.fn[3 * n + j] = id;   // (each of n's three words changes only from its own region face to that face's new one)
            // The candidate's own record is free from here on (its plane was last read by k_accept; it dies at the end of the round):
            // its first twelve words take the region's new faces — a moving point then asks for all of them at once instead of
            // walking a linked list, one dependent load per face (13 -> 8 us per round at 1 M points, r06); a region of more than
            // ten faces chains the rest through the records' `next` as before
            if (slot < kRegionTab) region_tab(b, o)[slot] = id;
            else { const int prev = atomicExch(&b.frec[o].pad[0], id); b.frec[id].next = prev; }   // (NOT o's own `next`: that is o's link in the list it was born into)
            ++id;
        }
    }
}

// sibling links: rotate around the horizon vertex v through the region's faces to the next horizon edge
__device__ __forceinline__ void link_faces(const Bufs& b, int par, int vblock, int nvblocks) {
    const ListWalk w = list_walk(b, par, vblock, nvblocks);
    for (int it = 0; it < w.loops; ++it) {
        const int g = list_face(w, it);
        if (g < 0 || !(b.fflags[g] & 4)) continue;  // faces of the accepted regions only (a second entry repeats the same writes)
        const int o = b.fowner[g];
        for (int k = 0; k < 3; ++k) {
            // NB fn[g][k] of a region face is never rewritten, so "across a horizon edge" is still decidable
            if (b.fowner[b.fn[3 * g + k]] == o) continue;
            const int my = b.newface[3 * g + k];
            if (my < 0 || my >= b.fcap) continue;
            {   // the face across died in another region this round: my neighbour across the edge is ITS new face on it
                const int n = b.fn[3 * g + k];
                if (b.fflags[n] & 4) {
                    int j = 0;
                    while (j < 3 && b.fn[3 * n + j] != g) ++j;
                    const int other = j < 3 ? b.newface[3 * n + j] : kNone;
                    if (other < 0 || other >= b.fcap) { b.ctrl[kCtrlError] |= kErrTopology; continue; }
                    b.fn[3 * my] = other;
                }
            }
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
__device__ __forceinline__ void reassign_points(const Bufs& b, FaceMaxTable& tab, int vblock, int nvblocks) {
    face_max_init(tab);
    const int stride = nvblocks * TO_BLOCK;
    const int nlive = b.ctrl[kCtrlNLive];
    // Most live points' faces survive a round: the loop is a scan of (live -> pface -> fflags), three dependent loads per point.
    // Four points per thread and trip keep four such chains in flight (12.3 M live points right after the join: 165 -> see DESIGN 6).
    constexpr int U = 4;
    const int nloop = (nlive + U * stride - 1) / (U * stride);
    for (int it = 0; it < nloop; ++it) {
        int jj[U], ii[U], gg[U], fl[U];
        for (int u = 0; u < U; ++u) { jj[u] = vblock * TO_BLOCK + threadIdx.x + (it * U + u) * stride; ii[u] = jj[u] < nlive ? b.live[jj[u]] : 0; }
        for (int u = 0; u < U; ++u) gg[u] = jj[u] < nlive ? b.pface[ii[u]] : kNone;
        int fo[U];
        for (int u = 0; u < U; ++u) { fl[u] = gg[u] >= 0 ? b.fflags[gg[u]] : 0; fo[u] = gg[u] >= 0 ? b.fowner[gg[u]] : kNone; }   // (both by the face: one trip)
        for (int u = 0; u < U; ++u) {
            const int i = ii[u], g = gg[u];
            double best = 0.0; int bf = kNone;
            if (g >= 0 && (fl[u] & 4)) {
                const int o = fo[u];
                // everything that hangs on the candidate is asked for at once: its apex (position), how many new faces its region
                // has, the table of their ids (three 16-byte loads out of its own record) — then the apex's coordinates, the
                // point's and the new faces' records together; a region of more than ten faces goes on through the records' `next`
                const unsigned long long akey = b.fmax[o];
                const int cnt = b.nfhead[o];
                const int4* rt = reinterpret_cast<const int4*>(region_tab(b, o));
                const int4 t0 = rt[0], t1 = rt[1], t2 = rt[2];
                const int over = cnt > kRegionTab ? b.frec[o].pad[0] : kNone;
                const int ap = apex_pos(akey);
                const double x = b.px[i], y = b.py[i], z = b.pz[i];
                const double apx = b.px[ap], apy = b.py[ap], apz = b.pz[ap];
                const int ids[kRegionTab] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w, t2.x, t2.y, t2.z, t2.w};
                const int inl = cnt < kRegionTab ? cnt : kRegionTab;
#pragma unroll
                for (int k = 0; k < kRegionTab; ++k) {
                    if (k >= inl) break;
                    const FaceRec r = b.frec[ids[k]];
                    const double d = plane_dist(r, x, y, z);
                    if (d > best) { best = d; bf = ids[k]; }
                }
                int steps = 0;
                for (int f = over; f >= 0 && steps < (1 << 20); ++steps) {
                    const FaceRec r = b.frec[f];  // one line per step of the list: plane, anchor and link
                    const double d = plane_dist(r, x, y, z);
                    if (d > best) { best = d; bf = f; }
                    f = r.next;
                }
                // (every new face has the apex as a vertex and two older ones, whose copies retired when THEY went in)
                if (i == ap || (x == apx && y == apy && z == apz)) { best = 0.0; bf = kNone; }
                b.pface[i] = bf;  // the apex retires as a vertex; a point outside no new face retires inside the hull
            }
            wave_face_max(b, tab, bf, apex_key(best, i));  // the new face's farthest point so far
        }
    }
    face_max_flush(b, tab);
}

// the new faces' links and the points' new conflict faces need the same thing — every face of the round created — and
// nothing of each other: one launch, the first `link_blocks` blocks link, the others move points
__global__ void __launch_bounds__(TO_BLOCK) k_link_reassign(Bufs b, int par, int link_blocks) {
    __shared__ FaceMaxTable tab;
    if (b.ctrl[kCtrlError] != 0) return;   // (k_new_faces ran out of faces in THIS round, or an earlier round failed: round_dead)
    if ((int)blockIdx.x < link_blocks) link_faces(b, par, blockIdx.x, link_blocks);
    else reassign_points(b, tab, blockIdx.x - link_blocks, gridDim.x - link_blocks);
}
// (experiments, TOHIP_HULL_SPLIT_LINK=1: the two halves as launches of their own, so that a profile shows which one the round waits for)
__global__ void __launch_bounds__(TO_BLOCK) k_link_only(Bufs b, int par) {
    if (b.ctrl[kCtrlError] != 0) return;
    link_faces(b, par, blockIdx.x, gridDim.x);
}
__global__ void __launch_bounds__(TO_BLOCK) k_reassign_only(Bufs b) {
    __shared__ FaceMaxTable tab;
    if (b.ctrl[kCtrlError] != 0) return;
    reassign_points(b, tab, blockIdx.x, gridDim.x);
}

// End of a round = start of the next: apexes of the faces created this round (point loop), then per listed face: the accepted
// regions' faces die and the round's ownership is cleared; the candidates that are still alive and the new faces with points
// outside them enter the next round's candidate list as their own owners.  Also run once after the initial tetrahedra
// (no lists yet; every face is new).  Needs a grid that is a multiple of kSubLists.
__global__ void __launch_bounds__(TO_BLOCK) k_round_tail(Bufs b, int par, int next_round) {
    if (b.ctrl[kCtrlError] != 0) return;   // (round_dead)
    const int sl = blockIdx.x % kSubLists, cap = sub_cap(b);
    int* __restrict__ next = b.cand[par ^ 1] + (size_t)sl * cap;
    int* next_n = ccnt(b, par ^ 1, sl);
    // One pass over both sources of the next round's candidates — the listed faces (a surviving candidate re-enters through its own
    // entry) and the faces created this round — with ONE reservation per step for both: an allocation is an atomic whose answer
    // the whole block waits for.
    const ListWalk w = list_walk(b, par, blockIdx.x, gridDim.x);
    const int f_lo = b.ctrl[kCtrlNFaces], nf = min(b.ctrl[kCtrlNFaces + 8], b.fcap);  // final since k_new_faces ended
    const int stride = gridDim.x * TO_BLOCK;
    const int nloop = (nf - f_lo + stride - 1) / stride;
    for (int it = 0; it < max(w.loops, nloop); ++it) {
        int f1 = kNone, f2 = kNone;
        bool c1 = false, c2 = false;
        if (it < w.loops) {
            bool from_cand;
            f1 = list_face(w, it, &from_cand);
            bool cand = false;
            if (f1 >= 0) {
                const int fl = b.fflags[f1];
                const int alive = (fl & 1) && !(fl & 4);
                cand = alive && b.fmax[f1] != 0ull;  // fmax persists: an outside set is fixed at the face's creation
                b.fowner[f1] = cand ? f1 : kNone;
                b.fflags[f1] = alive | (cand ? 2 : 0);
                if (from_cand) b.nfhead[f1] = 0;
                if (cand && from_cand) b.fprio[f1] = make_prio(f1, next_round, apex_dist(b.fmax[f1]) * b.frec[f1].inv_norm, b.hbits);
            }
            c1 = cand && from_cand;  // a candidate enters through its own entry, not through a claim's
        }
        if (it < nloop) {
            // consecutive faces to consecutive blocks: a round's few hundred new faces spread over all sub-lists
            f2 = f_lo + (it * TO_BLOCK + threadIdx.x) * gridDim.x + blockIdx.x;
            c2 = f2 < nf && (b.fflags[f2] & 1) && b.fmax[f2] != 0ull;
            if (c2) {
                b.fowner[f2] = f2; b.fflags[f2] = 3;
                b.fprio[f2] = make_prio(f2, next_round, apex_dist(b.fmax[f2]) * b.frec[f2].inv_norm, b.hbits);
            }
        }
        const int slot = block_alloc(next_n, (c1 ? 1 : 0) + (c2 ? 1 : 0));
        if (c1) {
            if (slot < cap) next[slot] = f1;
            else b.ctrl[kCtrlError] |= kErrCapacity;
        }
        if (c2) {
            const int s2 = slot + (c1 ? 1 : 0);
            if (s2 < cap) next[s2] = f2;
            else b.ctrl[kCtrlError] |= kErrCapacity;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) b.ctrl[kCtrlOverflow] = 0;  // read by k_accept only
}

// ---- the first rounds of a large build on a sample of the points ----------------------------------------------------
// While a hull has a few hundred faces, a round inserts a handful of points but moves EVERY point still outside it (the point
// kernels of the first 40-60 rounds were half the time of a 128 x 107 k batched build).  So those rounds run on every sub-th
// point (Morton order: a uniform spatial sample); the hull they grow is a convex polytope on input points like any other
// intermediate state.  Then every other point is given its conflict face by testing it against ALL faces of its segment's
// polytope (k_assign_all: f64 plane tests from LDS, ~1 ms for 13.6 M points x 500 faces) — points inside retire without ever
// having been moved — and the build goes on with all of them.

// ---- the sample's hull by sequential insertion, a block per segment (r06) ---------------------------------------------------
// The rounds above cost ~45 us each whatever they insert (five dependent launches of latency-bound kernels), and while a hull has
// a few hundred faces a round inserts a handful of points: the first ~22 rounds of a million-point build were 1 ms of its 4.8.
// A polytope of a few hundred faces and a sample of <= 4 096 points fit ONE workgroup: faces, planes and the polytope's vertices
// in LDS, eight sample points per thread in registers, and classic quickhull — insert the point that is highest above its
// conflict face, one at a time: the region it sees (breadth first from its conflict face: connected by construction, coplanar
// faces elsewhere cannot join it by rounding), one new face per horizon edge, sibling links by matching the horizon's vertices,
// the dead faces' points to the new faces.  An insertion is a dozen block barriers (2-3 us), no launch and no memory round trip.
// The block leaves what the join (k_seg_faces, k_assign_all, k_rebuild_candidates) expects of a sample phase:
// faces with planes and links, dead faces unflagged, the sample points' conflict faces and the faces' maxima over them.
// Face ids: the tetrahedron's 4 sg .. 4 sg + 3, then a block of KL - 4 ids per segment behind all tetrahedra (unused ones stay
// dead).  Any intermediate state of quickhull is a convex polytope on input points, so — like the sample rounds it replaces —
// this changes the schedule, not the result.
constexpr int kSerialThreads = 512, kSerialPts = 8, kSerialHmax = 512, kSerialReg = 512;

__device__ __forceinline__ int sample_stride(const Bufs& b, int lo, int hi) {
    return b.serial ? max(1, (hi - lo + kSerialThreads * kSerialPts - 1) / (kSerialThreads * kSerialPts)) : b.sub;
}
__device__ __forceinline__ bool is_sample(const Bufs& b, int lo, int hi, int j) {
    return b.serial ? ((j - lo) % sample_stride(b, lo, hi) == 0) : (j % b.sub == 0);
}

__host__ __device__ inline size_t sample_hull_lds_bytes(int KL) {
    // per face: plane 6 doubles, inv_norm, 3 vertices, 3 neighbours, a mark, a flag byte; per vertex (<= KL / 2 + 4): 3 doubles + position
    const size_t nv = (size_t)KL / 2 + 8;
    return sizeof(double) * 6 * KL + sizeof(double) * 3 * nv + sizeof(unsigned long long) * 24 + sizeof(float) * KL +
           sizeof(int) * (7 * (size_t)KL + 2 * nv + 4 * kSerialHmax + kSerialReg + 16) + (size_t)KL + 64;
}

// maximum of 64-bit keys over the wave on the DPP network (common.hpp: wave_max63_nn); valid in lane 63
__device__ __forceinline__ unsigned long long wave_max63_u64(unsigned long long key) {
    unsigned hi = (unsigned)(key >> 32), lo = (unsigned)key;
#define TO_U64_MAX_STEP(ctrl, rmask, bc) { \
        const unsigned h2 = (unsigned)TO_DPP_I(hi, hi, ctrl, rmask, bc), l2 = (unsigned)TO_DPP_I(lo, lo, ctrl, rmask, bc); \
        const bool take = h2 > hi || (h2 == hi && l2 > lo); \
        hi = take ? h2 : hi; lo = take ? l2 : lo; }
    TO_U64_MAX_STEP(0xB1, 0xF, true) TO_U64_MAX_STEP(0x4E, 0xF, true) TO_U64_MAX_STEP(0x141, 0xF, true) TO_U64_MAX_STEP(0x140, 0xF, true)
    TO_U64_MAX_STEP(0x142, 0xA, false) TO_U64_MAX_STEP(0x143, 0xC, false)
#undef TO_U64_MAX_STEP
    return ((unsigned long long)hi << 32) | lo;
}

__global__ void __launch_bounds__(kSerialThreads) k_sample_hull(Bufs b, int KL, int target) {
    extern __shared__ double smem[];
    const int NV = KL / 2 + 8;
    double* const s_nx = smem;            double* const s_ny = s_nx + KL;  double* const s_nz = s_ny + KL;
    double* const s_x0 = s_nz + KL;       double* const s_y0 = s_x0 + KL;  double* const s_z0 = s_y0 + KL;
    double* const s_vx = s_z0 + KL;       double* const s_vy = s_vx + NV;  double* const s_vz = s_vy + NV;
    unsigned long long* const s_red = (unsigned long long*)(s_vz + NV);   // [0..15] wave maxima, [16..19] the apex: x, y, z (bits), conflict face
    float* const s_inv = (float*)(s_red + 24);
    int* const s_fv = (int*)(s_inv + KL);     // [k * KL + f]: LOCAL vertex numbers
    int* const s_fn = s_fv + 3 * KL;          // [k * KL + f]: local face numbers
    int* const s_vpos = s_fn + 3 * KL;        // local vertex -> position
    int* const s_hu = s_vpos + NV;            // horizon: start vertex, end vertex, neighbour across, (visible face, edge)
    int* const s_hv = s_hu + kSerialHmax;
    int* const s_hn = s_hv + kSerialHmax;
    int* const s_hg = s_hn + kSerialHmax;
    int* const s_reg = s_hg + kSerialHmax;    // the region the apex sees (faces), breadth-first order
    int* const s_vslot = s_reg + kSerialReg;  // local vertex -> the horizon entry that starts at it (this insertion's; stale otherwise)
    int* const s_misc = s_vslot + NV;         // [1] horizon size of this insertion (-1: it does not fit), [2] error
    int* const s_vis = s_misc + 16;           // per face: seen by the current apex
    unsigned char* const s_alive = (unsigned char*)(s_vis + KL);

    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    const int sg = blockIdx.x, lo = b.seg_off[sg], hi = b.seg_off[sg + 1], fb = 4 * sg;
    const int K = KL - 4, base = 4 * b.nseg + sg * K;
    auto gid = [&](int f) { return f < 4 ? fb + f : base + f - 4; };
    if (sg == 0 && t == 0) b.ctrl[kCtrlNFaces + 8] = 4 * b.nseg + b.nseg * K;   // every id handed out: the unused ones are dead faces
    if (!(b.fflags[fb] & 1)) {   // no hull for this segment (too few points, flat, NaN): its ids stay dead
        for (int f = t; f < K; f += kSerialThreads) b.fflags[base + f] = 0;
        return;
    }
    // ---- the tetrahedron: faces 0..3, vertices 0..3 = a, c1, c2, d (init_tetrahedron's F = {a,c1,c2},{c1,a,d},{c2,c1,d},{a,c2,d})
    const int cpos[4] = {b.fv[3 * fb], b.fv[3 * fb + 1], b.fv[3 * fb + 2], b.fv[3 * fb + 5]};
    if (t < 4) {
        s_vpos[t] = cpos[t]; s_vx[t] = b.px[cpos[t]]; s_vy[t] = b.py[cpos[t]]; s_vz[t] = b.pz[cpos[t]];
        const FaceRec r = b.frec[fb + t];
        s_nx[t] = r.nx; s_ny[t] = r.ny; s_nz[t] = r.nz; s_x0[t] = r.x0; s_y0[t] = r.y0; s_z0[t] = r.z0; s_inv[t] = r.inv_norm;
        for (int k = 0; k < 3; ++k) {
            const int v = b.fv[3 * (fb + t) + k];
            s_fv[k * KL + t] = v == cpos[0] ? 0 : (v == cpos[1] ? 1 : (v == cpos[2] ? 2 : 3));
            s_fn[k * KL + t] = b.fn[3 * (fb + t) + k] - fb;
        }
        s_alive[t] = 1;
    }
    for (int f = t; f < KL; f += kSerialThreads) { s_vis[f] = 0; if (f >= 4) s_alive[f] = 0; }
    if (t < 16) s_misc[t] = 0;
    __syncthreads();
    auto pdist = [&](int f, double x, double y, double z) {   // plane_dist's expression on the LDS copy of the record
        const double dx = x - s_x0[f], dy = y - s_y0[f], dz = z - s_z0[f];
        return s_nx[f] * dx + s_ny[f] * dy + s_nz[f] * dz;
    };
    // ---- this thread's sample points: positions lo + (p * 1024 + t) * stride
    const int stride = sample_stride(b, lo, hi);
    double px[kSerialPts], py[kSerialPts], pz[kSerialPts];
    float hgt[kSerialPts];
    int cf[kSerialPts];
#pragma unroll
    for (int p = 0; p < kSerialPts; ++p) {
        const long long jl = (long long)lo + (long long)(p * kSerialThreads + t) * stride;
        const int j = jl < hi ? (int)jl : -1;
        bool ok = j >= 0 && j != cpos[0] && j != cpos[1] && j != cpos[2] && j != cpos[3] && (b.origin || b.perm[j] != hi - 1);
        px[p] = ok ? b.px[j] : 0.0; py[p] = ok ? b.py[j] : 0.0; pz[p] = ok ? b.pz[j] : 0.0;
        cf[p] = kNone; hgt[p] = 0.f;
        if (ok) {
            bool copy = false;   // an exact copy of a corner lies ON the tetrahedron whatever the rounding says (k_assign0: on_a_vertex)
            for (int c = 0; c < 4; ++c) copy |= px[p] == s_vx[c] && py[p] == s_vy[c] && pz[p] == s_vz[c];
            double best = 0.0;
            if (!copy)
                for (int f = 0; f < 4; ++f) {
                    const double d = pdist(f, px[p], py[p], pz[p]);
                    if (d > best) { best = d; cf[p] = f; }
                }
            if (cf[p] >= 0) hgt[p] = (float)best * s_inv[cf[p]];
        }
    }
#ifdef TOHIP_SH_STAMPS   // diagnostic build (tools/hpr_sample_stamps.sh): where an insertion's time goes, in shader clocks and 100 MHz ticks
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_t = clock64();
    const unsigned long long st_w0 = wall_clock64();
#define SH_STAMP(i) { const unsigned long long st_n = clock64(); st_acc[i] += st_n - st_t; st_t = st_n; }
#else
#define SH_STAMP(i)
#endif
    int nf = 4, nv = 4, created = 0;
    // Two block barriers per insertion: (1) after every wave has put its best point forward, (2) after the topology.  Face ids are
    // never reused, so a dead face keeps its mark (nobody's neighbour, nobody's conflict face: never looked at again) and nothing
    // has to be retired between insertions.
    while (created < target) {
        // ---- 1. the apex: the sample point highest above its conflict face (ties: the lowest position)
        unsigned long long key = 0ull;
#pragma unroll
        for (int p = 0; p < kSerialPts; ++p)
            if (cf[p] >= 0) {
                const unsigned j = (unsigned)(lo + (p * kSerialThreads + t) * stride);
                const unsigned long long k2 = ((unsigned long long)__float_as_uint(hgt[p]) << 32) | (0xffffffffu - j);
                key = k2 > key ? k2 : key;
            }
        key = wave_max63_u64(key);
        if (lane == 63) s_red[wid] = key;
        SH_STAMP(0)
        __syncthreads();
        SH_STAMP(1)
        key = 0ull;
        for (int w = 0; w < kSerialThreads / 64; ++w) key = s_red[w] > key ? s_red[w] : key;
        if (key == 0ull) break;   // nobody outside: the sample's hull is complete (block-uniform)
        const int ja = (int)(0xffffffffu - (unsigned)(key & 0xffffffffull));
        const int ka = (ja - lo) / stride;
        // ---- 2. ONE wave does the topology — the wave that owns the apex (its coordinates are in its registers): the region the apex
        // sees, breadth first from its conflict face, 64 (face, edge) pairs per step and no block barrier; an edge whose far side the
        // apex does not see is a horizon edge, found by the same test; then one new face per horizon edge and the sibling links.
        // The other waves wait at the barrier below.
        if (wid == (ka % kSerialThreads) >> 6) {
            double ax = 0.0, ay = 0.0, az = 0.0; int acf = 0;
#pragma unroll
            for (int p = 0; p < kSerialPts; ++p)
                if (p == ka / kSerialThreads) { ax = px[p]; ay = py[p]; az = pz[p]; acf = cf[p]; }
            const int src = (ka % kSerialThreads) & 63;
            ax = __shfl(ax, src); ay = __shfl(ay, src); az = __shfl(az, src); acf = __shfl(acf, src);
            if (lane == 0) { s_reg[0] = acf; s_vis[acf] = 1; }
            int r_lo = 0, r_hi = 1, H = 0;   // the frontier: s_reg[r_lo .. r_hi); horizon entries so far
            bool full = false;
            while (r_lo < r_hi && !full) {
                int r_new = r_hi;
                for (int e0 = 0; e0 < 3 * (r_hi - r_lo); e0 += 64) {
                    const int e = e0 + lane;
                    bool add = false, hor = false; int n = 0, f = 0, k = 0;
                    if (e < 3 * (r_hi - r_lo)) {
                        f = s_reg[r_lo + e / 3]; k = e % 3;
                        n = s_fn[k * KL + f];
                        const bool seen = pdist(n, ax, ay, az) > 0.0;   // (a face of the region already passes this test too)
                        if (seen) add = atomicExch(&s_vis[n], 1) == 0;      // two lanes on one face: one adds it
                        hor = !seen;
                    }
                    const unsigned long long bal = __ballot(add), bah = __ballot(hor);
                    const int pos = r_new + __popcll(bal & ((1ull << lane) - 1ull));
                    if (add && pos < kSerialReg) s_reg[pos] = n;
                    r_new += __popcll(bal);
                    const int slot = H + __popcll(bah & ((1ull << lane) - 1ull));
                    if (hor && slot < kSerialHmax) {
                        const int u = s_fv[k * KL + f];
                        s_hu[slot] = u; s_hv[slot] = s_fv[((k + 1) % 3) * KL + f]; s_hn[slot] = n; s_hg[slot] = f;
                        s_vslot[u] = slot;   // the horizon is a cycle: every vertex starts exactly one of its edges
                    }
                    H += __popcll(bah);
                    if (r_new > kSerialReg || H > kSerialHmax) { full = true; break; }
                }
                r_lo = r_hi; r_hi = full ? r_hi : r_new;
            }
            const int nreg = r_hi;
            const bool fits = !full && nf + H <= KL && nv + 1 <= NV;
            if (!fits) {   // no room for this insertion: the polytope stays as it is, the loop ends
                if (full) { for (int f = lane; f < nf; f += 64) s_vis[f] = 0; }   // (faces were marked that the list could not take)
                else { for (int i = lane; i < nreg; i += 64) s_vis[s_reg[i]] = 0; }
                if (lane == 0) s_misc[1] = -1;
            } else {
                if (lane == 0) { s_vpos[nv] = ja; s_vx[nv] = ax; s_vy[nv] = ay; s_vz[nv] = az; s_misc[1] = H;
                                 s_red[16] = (unsigned long long)__double_as_longlong(ax); s_red[17] = (unsigned long long)__double_as_longlong(ay);
                                 s_red[18] = (unsigned long long)__double_as_longlong(az); }
                for (int i = lane; i < nreg; i += 64) s_alive[s_reg[i]] = 0;   // the region dies (its marks stay: the points look at them)
                // one new face (u, v, apex) per horizon edge (u, v); its neighbour across the horizon re-pointed; sibling links: edge 1
                // of (u, v, apex) is (v, apex) -> the new face that STARTS at v, and that face's edge 2 comes back
                for (int q = lane; q < H; q += 64) {
                    const int id = nf + q, u = s_hu[q], v = s_hv[q], n = s_hn[q], g = s_hg[q];
                    const int nxt = s_vslot[v];
                    s_fv[id] = u; s_fv[KL + id] = v; s_fv[2 * KL + id] = nv;
                    s_fn[id] = n;
                    // set_plane's expression: a = u, c1 = v, c2 = apex
                    const double x0 = s_vx[u], y0 = s_vy[u], z0 = s_vz[u];
                    const double ux = s_vx[v] - x0, uy = s_vy[v] - y0, uz = s_vz[v] - z0;
                    const double wx = ax - x0, wy = ay - y0, wz = az - z0;
                    const double nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx;
                    s_nx[id] = nx; s_ny[id] = ny; s_nz[id] = nz; s_x0[id] = x0; s_y0[id] = y0; s_z0[id] = z0;
                    s_inv[id] = (float)__builtin_amdgcn_rsq(nx * nx + ny * ny + nz * nz);   // ranks candidates only (set_plane: 1 / sqrt)
                    s_alive[id] = 1;
                    for (int jn = 0; jn < 3; ++jn)
                        if (s_fn[jn * KL + n] == g) s_fn[jn * KL + n] = id;   // (an edge belongs to one horizon entry: nobody else writes this word)
                    if (nxt < 0 || nxt >= H || s_hu[nxt] != v) s_misc[2] = 1;   // the horizon is not a closed cycle: inconsistent predicates
                    else { s_fn[KL + id] = nf + nxt; s_fn[2 * KL + nf + nxt] = id; }
                }
            }
        }
        SH_STAMP(2)
        __syncthreads();
        SH_STAMP(3)
        const int H = s_misc[1];
        if (H < 0 || s_misc[2]) break;   // (block-uniform)
        const double ax = __longlong_as_double((long long)s_red[16]), ay = __longlong_as_double((long long)s_red[17]),
                     az = __longlong_as_double((long long)s_red[18]);
        // ---- 3. the points of the faces that died: to the new face they are farthest outside of, or inside.  Faces outside, points
        // inside: a plane is read once for the thread's (up to eight) orphans
        unsigned orphan = 0u;
#pragma unroll
        for (int p = 0; p < kSerialPts; ++p)
            if (cf[p] >= 0 && s_vis[cf[p]]) orphan |= 1u << p;
        if (orphan) {
            double best[kSerialPts];
            int bf[kSerialPts];
            unsigned test = 0u;   // the orphans that look for a new face: not the apex (a vertex now) nor its exact copies
#pragma unroll
            for (int p = 0; p < kSerialPts; ++p) {
                best[p] = 0.0; bf[p] = kNone;
                const int j = lo + (p * kSerialThreads + t) * stride;
                if (((orphan >> p) & 1u) && j != ja && !(px[p] == ax && py[p] == ay && pz[p] == az)) test |= 1u << p;
                if ((orphan >> p) & 1u) { cf[p] = kNone; hgt[p] = 0.f; }
            }
            for (int q = 0; q < H; ++q) {
                const int f = nf + q;
                const double nx = s_nx[f], ny = s_ny[f], nz = s_nz[f], x0 = s_x0[f], y0 = s_y0[f], z0 = s_z0[f];
#pragma unroll
                for (int p = 0; p < kSerialPts; ++p)
                    if ((test >> p) & 1u) {
                        const double dx = px[p] - x0, dy = py[p] - y0, dz = pz[p] - z0;   // plane_dist's expression
                        const double d = nx * dx + ny * dy + nz * dz;
                        if (d > best[p]) { best[p] = d; bf[p] = f; }
                    }
            }
#pragma unroll
            for (int p = 0; p < kSerialPts; ++p)
                if ((test >> p) & 1u) { cf[p] = bf[p]; hgt[p] = bf[p] >= 0 ? (float)best[p] * s_inv[bf[p]] : 0.f; }
        }
        nf += H; nv += 1; created += H;
        SH_STAMP(4)
    }
#ifdef TOHIP_SH_STAMPS
    if (sg == 0 && lane == 0) {   // per wave: [phase 1 | barrier 1 wait | topology (or nothing) | barrier 2 wait | phase 3], insertions, wall ticks
        unsigned long long* o = b.keys + 16 * wid;
        for (int i = 0; i < 5; ++i) o[i] = st_acc[i];
        o[5] = (unsigned long long)nv - 4; o[6] = wall_clock64() - st_w0; o[7] = (unsigned long long)nf;
    }
#endif
    __syncthreads();
    __syncthreads();
    if (s_misc[2] && t == 0) atomicOr(&b.ctrl[kCtrlError], kErrTopology);
    // ---- what the join expects: the faces ...
    for (int f = t; f < KL; f += kSerialThreads) {
        const int g = gid(f);
        if (f >= nf) { b.fflags[g] = 0; continue; }
        for (int k = 0; k < 3; ++k) { b.fv[3 * g + k] = s_vpos[s_fv[k * KL + f]]; b.fn[3 * g + k] = gid(s_fn[k * KL + f]); }
        FaceRec r;
        r.nx = s_nx[f]; r.ny = s_ny[f]; r.nz = s_nz[f]; r.x0 = s_x0[f]; r.y0 = s_y0[f]; r.z0 = s_z0[f];
        r.next = kNone; r.inv_norm = s_inv[f]; r.pad[0] = kNone; r.pad[1] = 0;
        b.frec[g] = r;
        b.fflags[g] = s_alive[f] ? 1 : 0;
        b.fowner[g] = kNone; b.fmax[g] = 0ull; b.nfhead[g] = 0;
        b.newface[3 * g] = kNone; b.newface[3 * g + 1] = kNone; b.newface[3 * g + 2] = kNone;
    }
    __syncthreads();   // (the maxima below go to words this block has just zeroed)
    // ... and the sample points: conflict face, and the faces' maxima over them (the other points add theirs in k_assign_all)
#pragma unroll
    for (int p = 0; p < kSerialPts; ++p) {
        const long long jl = (long long)lo + (long long)(p * kSerialThreads + t) * stride;
        if (jl >= hi) continue;
        const int j = (int)jl;
        if (j == cpos[0] || j == cpos[1] || j == cpos[2] || j == cpos[3]) continue;
        b.pface[j] = cf[p] >= 0 ? gid(cf[p]) : kNone;
        if (cf[p] >= 0) atomicMax(&b.fmax[gid(cf[p])], apex_key(pdist(cf[p], px[p], py[p], pz[p]), j));
    }
}

__global__ void __launch_bounds__(TO_BLOCK) k_live_stride(Bufs b) {  // live list = the sample
    const int cnt = (b.m1 + b.sub - 1) / b.sub;
    const int stride = gridDim.x * TO_BLOCK;
    for (int k = blockIdx.x * TO_BLOCK + threadIdx.x; k < cnt; k += stride) b.live[k] = k * b.sub;
    if (blockIdx.x == 0 && threadIdx.x == 0) b.ctrl[kCtrlNLive] = cnt;
}

__global__ void __launch_bounds__(TO_BLOCK) k_live_all(Bufs b) {
    const int stride = gridDim.x * TO_BLOCK;
    for (int j = blockIdx.x * TO_BLOCK + threadIdx.x; j < b.m1; j += stride) b.live[j] = j;
    if (blockIdx.x == 0 && threadIdx.x == 0) b.ctrl[kCtrlNLive] = b.m1;
}

// alive faces grouped by segment: counts, then (after a scan) the faces themselves into `list` (b.olist between two rounds)
__global__ void __launch_bounds__(TO_BLOCK) k_seg_faces(Bufs b, int* __restrict__ cnt, const int* __restrict__ off, int* __restrict__ list) {
    const int nf = min(b.ctrl[kCtrlNFaces + 8], b.fcap);
    const int stride = gridDim.x * TO_BLOCK;
    const int lane = threadIdx.x & 63;
    for (int f0 = blockIdx.x * TO_BLOCK; f0 < nf; f0 += stride) {   // (all lanes of a wave take part)
        const int f = f0 + threadIdx.x;
        const bool alive = f < nf && (b.fflags[f] & 1);
        const int sg = alive ? find_seg(b, b.fv[3 * f]) : -1;  // positions are grouped by segment like the expanded indices
        // neighbouring faces belong to the same segment (the sample's ids come in blocks per segment): ONE counter update per wave
        // and segment — 32 k returning atomics on the four cache lines of 128 counters were 69 us per pass
        unsigned long long todo = __ballot(alive);
        int k = 0;
        while (todo != 0ull) {
            const int s0 = __shfl(sg, (int)__builtin_ctzll(todo));
            const bool mine = alive && sg == s0;
            const unsigned long long m = __ballot(mine);
            const int lead = (int)__builtin_ctzll(m);
            int base = 0;
            if (lane == lead) base = atomicAdd(&cnt[s0], __popcll(m));
            base = __shfl(base, lead);
            if (mine) k = base + __popcll(m & ((1ull << lane) - 1ull));
            todo &= ~m;
        }
        if (alive && list) list[off[sg] + k] = f;
        // (the dormant points that join a face's outside set raise its fmax — and with it its apex — as they come: k_assign_all)
    }
}

constexpr int kAssignPts = 8, kAssignTile = 256;
__global__ void __launch_bounds__(TO_BLOCK) k_assign_all(Bufs b, const int* __restrict__ off, const int* __restrict__ list) {
    // per face: the plane as n.p - c with c = n.p0 LOWERED by a bound of both expressions' rounding (a filter: 3 FMAs), and the
    // record of the kernels' own distance expression (plane_dist) for the pairs that pass it
    struct Plane { double nx, ny, nz, cm, x0, y0, z0; int f; int pad; };
    __shared__ Plane pl[kAssignTile];
    __shared__ FaceMaxTable tab;
    face_max_init(tab);
    const int sg = blockIdx.y, lo = b.seg_off[sg], hi = b.seg_off[sg + 1];
    const int f0 = off[sg], f1 = off[sg + 1];
    // the tetrahedron's corners were picked among ALL points: a dormant one is a vertex already (its distance to its own faces
    // is rounding noise, not a conflict)
    const int fb = 4 * sg;
    const int c0 = b.fv[3 * fb], c1 = b.fv[3 * fb + 1], c2 = b.fv[3 * fb + 2], c3 = b.fv[3 * fb + 5];
    // largest coordinate magnitude of the segment (its bounding box)
    double rmax = 0.0;
    for (int k = 0; k < 6; ++k) rmax = fmax(rmax, fabs((double)fkey_inv(b.seg_bbox[6 * sg + k])));
    const int chunk = TO_BLOCK * kAssignPts;
    for (int base = lo + blockIdx.x * chunk; base < hi; base += gridDim.x * chunk) {  // block-uniform
        double x[kAssignPts], y[kAssignPts], z[kAssignPts], best[kAssignPts];
        int bf[kAssignPts];
        bool mine[kAssignPts];
        for (int k = 0; k < kAssignPts; ++k) {
            const int j = base + k * TO_BLOCK + threadIdx.x;
            // the sample's points have their faces (or have retired) already
            mine[k] = j < hi && !is_sample(b, lo, hi, j) && j != c0 && j != c1 && j != c2 && j != c3 && (b.origin || b.perm[j] != hi - 1);
            x[k] = mine[k] ? b.px[j] : 0.0; y[k] = mine[k] ? b.py[j] : 0.0; z[k] = mine[k] ? b.pz[j] : 0.0;
            best[k] = mine[k] ? 0.0 : INFINITY; bf[k] = kNone;   // (nothing beats infinity: the other lanes never ask for a test)
        }
        for (int t0 = f0; t0 < f1; t0 += kAssignTile) {
            const int nt = min(kAssignTile, f1 - t0);
            __syncthreads();
            if ((int)threadIdx.x < nt) {
                const int f = list[t0 + threadIdx.x];
                const FaceRec r = b.frec[f];
                const double c = r.nx * r.x0 + r.ny * r.y0 + r.nz * r.z0;
                // |n.p - c - plane_dist| <= ~8u (|nx|+|ny|+|nz|)(rmax + |p0|): 2e-15 covers 16u
                const double m = 2e-15 * ((fabs(r.nx) + fabs(r.ny) + fabs(r.nz)) * (2.0 * rmax) + fabs(c));
                pl[threadIdx.x] = Plane{r.nx, r.ny, r.nz, c - m, r.x0, r.y0, r.z0, f, 0};
            }
            __syncthreads();
            for (int q = 0; q < nt; ++q) {
                const double nx = pl[q].nx, ny = pl[q].ny, nz = pl[q].nz, cm = pl[q].cm;
                bool any = false;
                bool need[kAssignPts];
                for (int k = 0; k < kAssignPts; ++k) {
                    need[k] = fma(nx, x[k], fma(ny, y[k], fma(nz, z[k], -cm))) > best[k];  // an upper bound of the distance
                    any |= need[k];
                }
                if (!__any(any)) continue;
                FaceRec r;  // the same expression as every other distance test
                r.nx = nx; r.ny = ny; r.nz = nz; r.x0 = pl[q].x0; r.y0 = pl[q].y0; r.z0 = pl[q].z0;
                const int f = pl[q].f;
                for (int k = 0; k < kAssignPts; ++k) {
                    if (!need[k]) continue;
                    const double d = plane_dist(r, x[k], y[k], z[k]);
                    if (d > best[k] && !on_a_vertex(b, f, x[k], y[k], z[k])) { best[k] = d; bf[k] = f; }
                }
            }
        }
        for (int k = 0; k < kAssignPts; ++k) {
            const int j = base + k * TO_BLOCK + threadIdx.x;
            if (mine[k]) b.pface[j] = bf[k];
            wave_face_max(b, tab, mine[k] ? bf[k] : kNone, apex_key(mine[k] ? best[k] : 0.0, j));
        }
    }
    face_max_flush(b, tab);
}

// the candidate list of parity `par` from scratch: every alive face with points outside it, as its own owner
__global__ void __launch_bounds__(TO_BLOCK) k_rebuild_candidates(Bufs b, int par, int next_round) {
    const int sl = blockIdx.x % kSubLists, cap = sub_cap(b);
    int* __restrict__ next = b.cand[par] + (size_t)sl * cap;
    int* next_n = ccnt(b, par, sl);
    const int nf = min(b.ctrl[kCtrlNFaces + 8], b.fcap);
    const int stride = gridDim.x * TO_BLOCK;
    const int nloop = (nf + stride - 1) / stride;
    for (int it = 0; it < nloop; ++it) {
        const int f = (it * TO_BLOCK + threadIdx.x) * gridDim.x + blockIdx.x;
        const bool alive = f < nf && (b.fflags[f] & 1);
        const bool cand = alive && b.fmax[f] != 0ull;
        if (alive) { b.fowner[f] = cand ? f : kNone; b.fflags[f] = 1 | (cand ? 2 : 0); b.nfhead[f] = 0; }
        if (cand) b.fprio[f] = make_prio(f, next_round, apex_dist(b.fmax[f]) * b.frec[f].inv_norm, b.hbits);
        const int slot = block_alloc(next_n, cand ? 1 : 0);
        if (cand) {
            if (slot < cap) next[slot] = f;
            else b.ctrl[kCtrlError] |= kErrCapacity;
        }
    }
}

// Which of several IDENTICAL rows carries a hull vertex is the build's business (copies of a vertex are treated as lying on the
// hull), and it depends on the schedule (a large build's sample may meet a later copy first).  Reported is always the row with
// the LOWEST original index: identical rows share a sort key (segment, Morton cell), the sort is stable, so inside the run of
// equal keys around the vertex the first row with its coordinates is that one.  The sorted keys are still in keys2 (r06: until then
// the walk recomputed every row's cell from its coordinates — 13 us per build with 30-bit cells, 78 with the 24-bit ones).  (Qhull's
// own choice among identical rows follows its insertion history — the first copy in ~70 % of the cases, the last in the others —
// and cannot be reproduced by a parallel build.)
__device__ __forceinline__ int lowest_identical_row(const Bufs& b, int j) {
    const unsigned long long key = sorted_key(b, j);
    const double x = b.px[j], y = b.py[j], z = b.pz[j];
    int first = j;
    for (int i = j - 1; i >= 0 && sorted_key(b, i) == key; --i)
        if (b.px[i] == x && b.py[i] == y && b.pz[i] == z) first = i;
    return first;
}

__global__ void __launch_bounds__(TO_BLOCK) k_mark_vertices(Bufs b) {
    const int nf = min(b.ctrl[kCtrlNFaces + 8], b.fcap);  // every face created (the published count lags by the last round)
    const int stride = gridDim.x * TO_BLOCK;
    for (int f = blockIdx.x * TO_BLOCK + threadIdx.x; f < nf; f += stride)
        if (b.fflags[f] & 1)
            for (int k = 0; k < 3; ++k) {
                const int v = b.fv[3 * f + k];
                // a vertex belongs to ~6 faces: the first of them looks its row up (a walk over the run of equal sort keys)
                if (atomicExch(&b.vals[v], 1) == 0) b.vflag[b.perm[lowest_identical_row(b, v)]] = 1;  // flags in the caller's (expanded) numbering
            }
}

// ---- compaction of the live-point list (ordered: the Morton locality stays) -----------------------
__global__ void __launch_bounds__(TO_BLOCK) k_live_count(Bufs b, int nlive, int* __restrict__ tile_cnt) {
    __shared__ int wave_cnt[TO_WAVES_PER_BLOCK];
    const int tile0 = blockIdx.x * 1024;
    int cnt = 0;
    for (int k = 0; k < 4; ++k) {
        const int j = tile0 + k * TO_BLOCK + threadIdx.x;
        cnt += __popcll(__ballot(j < nlive && b.pface[b.live[j]] >= 0));
    }
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

__global__ void __launch_bounds__(TO_BLOCK) k_live_write(Bufs b, int nlive_old, const int* __restrict__ tile_off) {
    __shared__ int wave_cnt[TO_WAVES_PER_BLOCK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile0 = blockIdx.x * 1024;
    int base = tile_off[blockIdx.x];
    for (int k = 0; k < 4; ++k) {
        const int j = tile0 + k * TO_BLOCK + threadIdx.x;
        const int i = j < nlive_old ? b.live[j] : 0;
        const bool keep = j < nlive_old && b.pface[i] >= 0;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (keep) b.live2[off + __popcll(bal & ((1ull << lane) - 1ull))] = i;
        base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
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

// What the host wants to know of a batch of rounds, written by ONE wave straight into host memory the GPU has mapped (r06; until
// then a 16 KB copy of the control block behind an event: the copy's kernel was 1.4 us between two gaps of 12 and 6 us — the
// system-scope flush and the signal — on every one of a build's ten readbacks): the scalars, the two candidate counts summed
// over their sub-lists, and last, released at system scope, the sequence number the host is polling for.
constexpr int kRepCand = kCtrlInts, kRepSeq = kCtrlInts + 2, kRepInts = 32;
__global__ void __launch_bounds__(64) k_report(Bufs b, int* __restrict__ rec, int seq) {
    static_assert(kSubLists == 64, "a lane per sub-list");
    const int lane = threadIdx.x;
    int c0 = *ccnt(b, 0, lane), c1 = *ccnt(b, 1, lane);
    for (int s = 32; s > 0; s >>= 1) { c0 += __shfl_xor(c0, s); c1 += __shfl_xor(c1, s); }
    if (lane < kCtrlInts) rec[lane] = b.ctrl[lane];
    if (lane == 0) { rec[kRepCand] = c0; rec[kRepCand + 1] = c1; }
    __threadfence_system();   // (one wave: every lane's stores are out before lane 0 goes on)
    if (lane == 0) __hip_atomic_store(&rec[kRepSeq], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

inline int nblocks(int64_t n, int cap = 2048) {
    int64_t nb = (n + TO_BLOCK - 1) / TO_BLOCK;
    return (int)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}

// Builds the hull of pts (n,3) [+ origin]; leaves vflag set.  Synchronises the stream.
// b.seg_off must already be on the device (k_single_segment for one hull).
static int build(const Bufs& b_in, const float* pts, int with_origin, int64_t max_seg_points, hipStream_t st, int* rounds_out) {
    Bufs b = b_in;  // local copy: the two live-point buffers swap roles at every compaction
    b.origin = with_origin ? 1 : 0;
    {
        // 9 = exponent + one mantissa bit: apexes more than ~1.4x higher go first, the hash decides among the rest.  Measured
        // (tools/hpr_sweep.sh): 0 bits (hash only) 177 rounds / 10.7 ms at 1 M points, 6: 116 / 7.0, 9: 106 / 6.2, 12: 120 / 6.7, 16: 129 / 7.2
        static const int hb = getenv("TOHIP_HULL_HBITS") ? atoi(getenv("TOHIP_HULL_HBITS")) : 9;  // experiments
        b.hbits = hb < 0 ? 0 : (hb > 16 ? 16 : hb);
        static const int eo = getenv("TOHIP_HULL_EARLY_OUT") ? atoi(getenv("TOHIP_HULL_EARLY_OUT")) : 1;  // experiments
        b.early_out = eo;
        static const int se = getenv("TOHIP_HULL_SHARE_EDGES") ? atoi(getenv("TOHIP_HULL_SHARE_EDGES")) : 1;    // experiments: 0 = adjacent regions exclude each other (r05)
        b.share_edges = se;
    }
    {
        // large segments: the first rounds on a sample (see k_assign_all)
        static const int force_sub = getenv("TOHIP_HULL_SUB") ? atoi(getenv("TOHIP_HULL_SUB")) : 0;  // experiments: 1 = off
        const int64_t avg = b.m1 / b.nseg;
        b.sub = (avg >= 32768 && b.nseg <= 65535) ? (int)std::min<int64_t>(128, std::max<int64_t>(2, avg / 1024)) : 1;
        if (force_sub > 0) b.sub = force_sub;
    }
    hipError_t e = hipMemsetAsync(b.ctrl, 0, sizeof(int) * kCtrlTotal, st);
    if (e != hipSuccess) return (int)e;
    k_bbox_init<<<(6 * b.nseg + TO_BLOCK - 1) / TO_BLOCK, TO_BLOCK, 0, st>>>(b);
    // six same-address atomics per wave and segment: 4096 waves on ONE segment's box cost 0.27 ms, 512 cost 0.03
    if (b.nseg == 1 && b.m1 >= 65536) {
        unsigned* copies = reinterpret_cast<unsigned*>(b.keys2);
        k_bbox1_init<<<1, TO_BLOCK, 0, st>>>(copies);
        k_bbox1<<<nblocks(b.m1, 1024), TO_BLOCK, 0, st>>>(b, pts, with_origin, copies);
        k_bbox1_fold<<<1, 64, 0, st>>>(b, copies);
    } else {
        k_bbox<<<nblocks(b.m1, b.nseg == 1 ? 128 : 1024), TO_BLOCK, 0, st>>>(b, pts, with_origin);
    }
    int seg_bits = 0;
    while ((1 << seg_bits) < b.nseg) ++seg_bits;
    b.key32 = kMortonBits + seg_bits <= 32 ? 1 : 0;
    k_sort_keys<<<nblocks(b.m1), TO_BLOCK, 0, st>>>(b, pts, with_origin);
    TO_HIP_CHECK_LAUNCH();
    {
        size_t tmp = b.sort_tmp_bytes;
        // stable: equal cells keep the caller's order
        const hipError_t es = b.key32
            ? sort_pairs(b.sort_tmp, tmp, reinterpret_cast<const unsigned*>(b.keys), reinterpret_cast<unsigned*>(b.keys2),
                                                 b.vals, b.perm, b.m1, 0, kMortonBits + seg_bits, st)
            : sort_pairs(b.sort_tmp, tmp, b.keys, b.keys2, b.vals, b.perm, b.m1, 0, kMortonBits + seg_bits, st);
        if (es != hipSuccess) return (int)es;
    }
    k_load<<<nblocks(b.m1), TO_BLOCK, 0, st>>>(b, pts, with_origin);
    TO_HIP_CHECK_LAUNCH();
    if (b.nseg == 1 && b.m1 >= 65536) {
        // the UNSORTED keys (8 bytes per point) are free after the sort: the passes' partials and the selections live there
        // (keys2 keeps the sorted keys: k_mark_vertices compares them)
        double* pkey = (double*)b.keys;
        tie_t* pidx = (tie_t*)(pkey + 4 * kInitBlocks);
        InitSel* sel = (InitSel*)(pidx + 4 * kInitBlocks);
        for (int pass = 0; pass < 4; ++pass) k_init1_pass<<<kInitBlocks, HULL_INIT_THREADS, 0, st>>>(b, pass, pkey, pidx, sel);
        k_init1_finish<<<1, HULL_INIT_THREADS, 0, st>>>(b, kInitBlocks, pkey, pidx, sel);
    } else {
        k_init<<<b.nseg, HULL_INIT_THREADS, 0, st>>>(b);
    }
    TO_HIP_CHECK_LAUNCH();
    // Large segments: the sample's hull by sequential insertion, one block per segment out of LDS (k_sample_hull) instead of the
    // first ~20-25 rounds; then the join below, immediately.  TOHIP_HULL_SERIAL=0: the sample's rounds as until r05 (experiments);
    // TOHIP_HULL_SERIAL_IDS: face ids per segment (LDS: 92 bytes each), TOHIP_HULL_SERIAL_FACES: stop after that many created.
    int serial_ids = 0;
    if (b.sub > 1) {
        static const int want = getenv("TOHIP_HULL_SERIAL") ? atoi(getenv("TOHIP_HULL_SERIAL")) : 1;
        // measured (tools/hpr_serial_sweep.sh, r06): an insertion costs ~5 us here, a round ~45 us for one hull and ~160 us for 128 —
        // one hull is best served by 192-448 ids (flat), a batch by 768-896
        static const int ids_env = getenv("TOHIP_HULL_SERIAL_IDS") ? atoi(getenv("TOHIP_HULL_SERIAL_IDS")) : 0;
        const int ids = std::max(64, std::min(1536, (ids_env > 0 ? ids_env : (b.nseg == 1 ? 192 : 512)) / 2 * 2));   // (with regions sharing edges and the polled report, r06: one hull 2.19 / 2.19 / 2.17 / 2.23 / 2.26 / 2.39 ms at 64 / 128 / 192 / 256 / 320 / 448, the sample's rounds instead 2.22; 128 views 9.0 / 8.8 / 8.7 / 8.9 / 9.2 ms at 256 / 384 / 512 / 768 / 1024)
        if (want && (int64_t)4 * b.nseg + (int64_t)b.nseg * (ids - 4) <= (int64_t)b.fcap) {
            static int lds_ok_for = 0;   // the dynamic-LDS limit of the kernel is raised once per size
            const size_t lds = sample_hull_lds_bytes(ids);
            if (lds_ok_for != ids) {
                if (hipFuncSetAttribute((const void*)k_sample_hull, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess) lds_ok_for = ids;
                else (void)hipGetLastError();
            }
            if (lds_ok_for == ids) serial_ids = ids;
        }
    }
    if (serial_ids) {
        static const int faces_env = getenv("TOHIP_HULL_SERIAL_FACES") ? atoi(getenv("TOHIP_HULL_SERIAL_FACES")) : 0;
        const int target = faces_env > 0 ? faces_env : serial_ids;   // (the loop also stops when an insertion no longer fits the ids)
        b.serial = 1;
        k_sample_hull<<<b.nseg, kSerialThreads, sample_hull_lds_bytes(serial_ids), st>>>(b, serial_ids, target);
#ifdef TOHIP_SH_STAMPS
        {
            unsigned long long hst[16 * 8];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(hst, b.keys, sizeof(hst), hipMemcpyDeviceToHost);
            for (int w = 0; w < 8; ++w)
                fprintf(stderr, "sample hull wave %d: phase1 %llu  wait1 %llu  topology %llu  wait2 %llu  phase3 %llu clocks; %llu insertions, %llu faces, %.1f us wall\n", w,
                        hst[16 * w], hst[16 * w + 1], hst[16 * w + 2], hst[16 * w + 3], hst[16 * w + 4], hst[16 * w + 5], hst[16 * w + 7], hst[16 * w + 6] * 0.01);
        }
#endif
    } else {
        if (b.sub > 1) k_live_stride<<<nblocks((b.m1 + b.sub - 1) / b.sub), TO_BLOCK, 0, st>>>(b);
        k_assign0<<<nblocks(b.m1, 1024), TO_BLOCK, 0, st>>>(b);
        k_round_tail<<<kSubLists, TO_BLOCK, 0, st>>>(b, 1, 0);  // round 0's candidates: the tetrahedra's faces with points outside
    }
    TO_HIP_CHECK_LAUNCH();
    // One readback = k_report's record in mapped host memory, the host polling its sequence number: the host keeps ONE batch of
    // rounds enqueued ahead of the readback it is waiting for, so the GPU never idles while the host looks at the counts.
    // Rounds enqueued after the hull is complete find no candidate and change nothing; the counts only size grids, and every
    // kernel strides over its lists.
    struct Pinned {
        int* rec[2] = {nullptr, nullptr};    // host addresses
        int* drec[2] = {nullptr, nullptr};   // the same records as the device sees them
        int seq = 0;                         // last sequence number handed out (never reused: a stale record cannot match)
        bool ok = false;
        Pinned() {   // mapped, coherent host memory: the GPU's stores land in it without a copy
            ok = true;
            for (int i = 0; i < 2 && ok; ++i) {
                ok = hipHostMalloc((void**)&rec[i], sizeof(int) * kRepInts, hipHostMallocPortable | hipHostMallocMapped) == hipSuccess &&
                     hipHostGetDevicePointer((void**)&drec[i], rec[i], 0) == hipSuccess;
                if (ok) for (int k = 0; k < kRepInts; ++k) rec[i][k] = 0;
            }
        }
        ~Pinned() {
            for (int i = 0; i < 2; ++i)
                if (rec[i]) (void)hipHostFree(rec[i]);
        }
        Pinned(const Pinned&) = delete;
        Pinned& operator=(const Pinned&) = delete;
    };
    // one set per (thread, device): it lives as long as the thread (a build must not pay two pinned allocations)
    static thread_local std::map<int, std::unique_ptr<Pinned>> pins;
    int cur_dev = 0;
    if (hipGetDevice(&cur_dev) != hipSuccess) return TOHIP_EINVAL;
    std::unique_ptr<Pinned>& pin_slot = pins[cur_dev];
    if (!pin_slot) pin_slot.reset(new Pinned());
    Pinned& pin = *pin_slot;
    if (!pin.ok) return TOHIP_EINVAL;
    const int* h = pin.rec[0];
    int wslot = 0, rslot = 0, inflight = 0;
    int want_seq[2] = {0, 0};
    auto post_readback = [&]() -> hipError_t {  // enqueue the report; collect() waits for it
        want_seq[wslot] = ++pin.seq;
        k_report<<<1, 64, 0, st>>>(b, pin.drec[wslot], want_seq[wslot]);
        const hipError_t er = hipGetLastError();
        wslot ^= 1; ++inflight;
        return er;
    };
    double host_enqueue_us = 0.0, host_wait_us = 0.0;  // for the trace: where the host's time goes
    auto collect = [&]() -> hipError_t {
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t er = hipSuccess;
        const volatile int* seqp = pin.rec[rslot] + kRepSeq;
        // poll; every so often ask the stream whether it is still working (a fault, or a report that never ran, must not hang the host)
        for (long spin = 1; __atomic_load_n(seqp, __ATOMIC_ACQUIRE) != want_seq[rslot]; ++spin) {
            if ((spin & 0x3fff) == 0) {
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {   // everything enqueued has run: the record is there, or it never will be
                    if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) != want_seq[rslot]) er = hipErrorUnknown;
                    break;
                }
                if (q != hipErrorNotReady) { er = q; break; }
            }
            __builtin_ia32_pause();
        }
        host_wait_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        h = pin.rec[rslot];
        rslot ^= 1; --inflight;
        return er;
    };
    auto candidates = [&](int par) { return h[kRepCand + par]; };
    auto cdiv = [](int64_t a, int64_t d) { return (a + d - 1) / d; };
    e = post_readback();
    if (e == hipSuccess) e = collect();
    if (e != hipSuccess) return (int)e;
    if (b.nseg == 1 && (h[kCtrlError] & kErrNaN)) return TOHIP_ENAN;
    if (b.nseg == 1 && (h[kCtrlError] & kErrFlat)) return TOHIP_EINVAL;
    int nf = h[kCtrlNFaces + 8];
    const int max_rounds = 100000;
    int round = 0 /* rounds enqueued */, ncand = candidates(0), live_bound = (b.m1 + b.sub - 1) / b.sub;
    static const bool batch_env = getenv("TOHIP_HULL_BATCH") != nullptr;
    static const int batch = batch_env ? std::max(1, atoi(getenv("TOHIP_HULL_BATCH"))) : 3;  // experiments: rounds per readback (1 M points: 2.38 / 2.29 / 2.26 / 2.29 / 2.30 / 2.37 ms at 1 / 2 / 3 / 4 / 6 / 8)
    static const bool always_careful = getenv("TOHIP_HULL_CAREFUL") != nullptr;                   // experiments: the slow path only
    static const int compact_every = getenv("TOHIP_HULL_COMPACT") ? atoi(getenv("TOHIP_HULL_COMPACT")) : 2;
    static const bool trace = getenv("TOHIP_HULL_TRACE") != nullptr;   // experiments: the build's progress, one line per readback
    static const int fused_verdict = getenv("TOHIP_HULL_FUSED_ACCEPT") ? atoi(getenv("TOHIP_HULL_FUSED_ACCEPT")) : 1;   // experiments: 0 = a k_accept launch per round
    static const int sub_claim = getenv("TOHIP_HULL_SUB_CLAIM") ? atoi(getenv("TOHIP_HULL_SUB_CLAIM")) : 4;   // experiments: candidates per wave from which the quarter-wave walk takes over (0 = never)
    static const int sub_lanes = getenv("TOHIP_HULL_SUB_LANES") ? atoi(getenv("TOHIP_HULL_SUB_LANES")) : 8;   // eight candidates to a wave (measured: 12.0-12.1 ms for 128 views; 16 lanes each: 12.3-12.4; a wave each: 13.9-14.0)
    static const int reassign_cap = getenv("TOHIP_HULL_REASSIGN_CAP") ? std::max(64, atoi(getenv("TOHIP_HULL_REASSIGN_CAP"))) : 1024;   // experiments: blocks of the point kernel
    static const int grid_scale = getenv("TOHIP_HULL_GRID_SCALE") ? std::max(1, atoi(getenv("TOHIP_HULL_GRID_SCALE"))) : 4;   // experiments (1 M points: 2.97 / 2.89 / 2.85 / 2.84 ms at 1 / 2 / 4 / 8)
    static const int split_link = getenv("TOHIP_HULL_SPLIT_LINK") ? atoi(getenv("TOHIP_HULL_SPLIT_LINK")) : 0;   // experiments
    static const int join_faces = getenv("TOHIP_HULL_JOIN_FACES") ? atoi(getenv("TOHIP_HULL_JOIN_FACES")) : 192;  // experiments

    // `careful`: ownership propagated to convergence with the host checking (after a batch that accepted nobody)
    auto enqueue_rounds_inner = [&](int nrounds, bool careful) -> int {
        // grids: a wave per candidate; a thread per listed face (a candidate claims a handful of faces; lists grow within a batch)
        // (`ncand` is what a readback said two batches ago, and while a hull grows its candidates multiply by ~1.25 per round: the
        // grids are cut for `grid_scale` times as many — a wave that finds no candidate leaves, one that finds three walks them in turn)
        const int64_t nc_grid = (int64_t)ncand * grid_scale;
        const int ga = kSubLists * (int)std::min<int64_t>(32, std::max<int64_t>(1, cdiv(nc_grid, kSubLists * TO_WAVES_PER_BLOCK)));
        const int gl = kSubLists * (int)std::min<int64_t>(16, std::max<int64_t>(1, cdiv(nc_grid * 16, kSubLists * TO_BLOCK)));
        for (int r = 0; r < nrounds; ++r, ++round) {
            const int par = round & 1;
            if (!careful) {
                // Fast path: a wave per candidate walks and claims its region (one launch, no readback).  Incomplete
                // ownership is safe — a candidate is accepted only if it owns every face its apex sees (k_accept).
                // many candidates (a batch of views): four to a wave (k_owner_claim_sub); else a wave each
                if (fused_verdict && sub_claim && (int64_t)ncand >= (int64_t)sub_claim * ga * TO_WAVES_PER_BLOCK)
                    { if (sub_lanes == 8) k_owner_claim_sub<8><<<ga, TO_BLOCK, 0, st>>>(b, round, par); else k_owner_claim_sub<16><<<ga, TO_BLOCK, 0, st>>>(b, round, par); }
                else
                    k_owner_claim<<<ga, TO_BLOCK, 0, st>>>(b, round, par, fused_verdict);
                TO_HIP_CHECK_LAUNCH();
            } else {
                while (true) {
                    hipError_t ec = hipMemsetAsync(b.ctrl + kCtrlChanged, 0, sizeof(int), st);
                    if (ec != hipSuccess) return (int)ec;
                    k_owner_prop<<<nblocks(nf), TO_BLOCK, 0, st>>>(b, round);
                    k_owner_prop<<<nblocks(nf), TO_BLOCK, 0, st>>>(b, round);
                    TO_HIP_CHECK_LAUNCH();
                    ec = post_readback();
                    if (ec == hipSuccess) ec = collect();
                    if (ec != hipSuccess) return (int)ec;
                    if (!h[kCtrlChanged]) break;
                }
                k_owned_list<<<kSubLists * (int)std::min<int64_t>(16, std::max<int64_t>(1, cdiv(nf, kSubLists * TO_BLOCK))), TO_BLOCK, 0, st>>>(b, par);
                TO_HIP_CHECK_LAUNCH();
            }
            const int gr = nblocks(live_bound, reassign_cap);
            if (careful || !fused_verdict) k_accept<<<gl, TO_BLOCK, 0, st>>>(b, round, par);   // (the fast path's walk has given its verdict)
            k_new_faces<<<gl, TO_BLOCK, 0, st>>>(b, par);
            if (split_link) {
                k_link_only<<<gl, TO_BLOCK, 0, st>>>(b, par); k_reassign_only<<<gr, TO_BLOCK, 0, st>>>(b);
                if (split_link == 2) k_reassign_only<<<gr, TO_BLOCK, 0, st>>>(b);   // (experiments: the second pass moves nobody — what the scan alone costs)
            }
            else k_link_reassign<<<gl + gr, TO_BLOCK, 0, st>>>(b, par, gl);
            const int gt = gl;   // (a list walk: the apexes of the new faces come with their maxima since r06, no pass over the live points)
            k_round_tail<<<gt, TO_BLOCK, 0, st>>>(b, par, round + 1);
            TO_HIP_CHECK_LAUNCH();
        }
        return TOHIP_OK;
    };

    auto enqueue_rounds = [&](int nrounds, bool careful) -> int {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = enqueue_rounds_inner(nrounds, careful);
        host_enqueue_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        return rc;
    };
    int rounds_seen = 0;                 // rounds covered by the last collected readback
    int round_of[2] = {0, 0};            // rounds enqueued when each in-flight readback was posted
    int batches_since_compaction = 0;
    int compaction_pending_until = -1;   // a compaction was enqueued when `round` was this: older readbacks hold the old live count
    bool stalled = false;                // the last collected batch accepted nobody
    int careful_at = -1;                 // `round` right after the careful round was enqueued
    const int ahead = always_careful ? 1 : 2;
    auto drain = [&](int rc) { while (inflight > 0) (void)collect(); return rc; };
    // the sample's rounds end when its hulls have a few hundred faces each (faces created ~ 3x faces alive), are complete, or late
    const int64_t switch_faces = (int64_t)join_faces * b.nseg;
    auto join_all_points = [&]() -> int {
        hipError_t ej = hipMemsetAsync(b.seg_cnt, 0, sizeof(int) * (size_t)b.nseg, st);
        if (ej != hipSuccess) return (int)ej;
        k_seg_faces<<<nblocks(nf), TO_BLOCK, 0, st>>>(b, b.seg_cnt, nullptr, nullptr);
        launch_scan_tiles(b.seg_cnt, b.nseg, b.seg_start, b.seg_start + b.nseg, st);
        ej = hipMemsetAsync(b.flip_max, 0, sizeof(int) * (size_t)b.nseg, st);  // free since the flip: the scatter's cursors
        if (ej != hipSuccess) return (int)ej;
        k_seg_faces<<<nblocks(nf), TO_BLOCK, 0, st>>>(b, b.flip_max, b.seg_start, b.olist);  // olist is free between two rounds
        const int gx = (int)std::min<int64_t>(1024, std::max<int64_t>(1, cdiv(max_seg_points + 1, TO_BLOCK * kAssignPts)));
        k_assign_all<<<dim3(gx, b.nseg), TO_BLOCK, 0, st>>>(b, b.seg_start, b.olist);
        TO_HIP_CHECK_LAUNCH();
        // the live list: every point that has a conflict face now
        k_live_all<<<nblocks(b.m1), TO_BLOCK, 0, st>>>(b);
        const int ntl = (b.m1 + 1023) / 1024;
        k_live_count<<<ntl, TO_BLOCK, 0, st>>>(b, b.m1, b.tile_cnt);
        launch_scan_tiles(b.tile_cnt, ntl, b.tile_off, b.ctrl + kCtrlNLive, st);
        k_live_write<<<ntl, TO_BLOCK, 0, st>>>(b, b.m1, b.tile_off);
        TO_HIP_CHECK_LAUNCH();
        int* t = b.live; b.live = b.live2; b.live2 = t;
        const int par = round & 1;
        ej = hipMemsetAsync(b.ctrl + kCtrlInts + par * kSubLists * kCntStride, 0, sizeof(int) * kSubLists * kCntStride, st);
        if (ej != hipSuccess) return (int)ej;
        k_rebuild_candidates<<<kSubLists * (int)std::min<int64_t>(16, std::max<int64_t>(1, cdiv(nf, kSubLists * TO_BLOCK))), TO_BLOCK, 0, st>>>(b, par, round);
        TO_HIP_CHECK_LAUNCH();
        b.sub = 1; b.serial = 0;
        return TOHIP_OK;
    };
    while ((ncand > 0 || b.sub > 1) && round < max_rounds) {
        if (b.sub > 1 && (b.serial || nf >= switch_faces || ncand == 0 || round >= 96)) {
            while (inflight > 0) {  // the rounds in flight belong to the sample
                e = collect();
                if (e != hipSuccess) return drain((int)e);
                if (h[kCtrlError]) return drain((h[kCtrlError] & kErrCapacity) ? TOHIP_ENOSPC : TOHIP_ENOTCONV);
            }
            nf = h[kCtrlNFaces + 8] < b.fcap ? h[kCtrlNFaces + 8] : b.fcap;
            const int rc = join_all_points();
            if (rc != TOHIP_OK) return rc;
            e = post_readback();
            if (e == hipSuccess) e = collect();
            if (e != hipSuccess) return drain((int)e);
            if (h[kCtrlError]) return (h[kCtrlError] & kErrCapacity) ? TOHIP_ENOSPC : TOHIP_ENOTCONV;
            ncand = candidates(round & 1);
            if (trace) fprintf(stderr, "hull: all points joined after round %d: faces %d live %d candidates %d\n", round, nf, h[kCtrlNLive], ncand);
            stalled = false;
            live_bound = std::max(1, h[kCtrlNLive]);   // the join compacted the list: this readback holds its length
            batches_since_compaction = 0;
            compaction_pending_until = round;
            continue;
        }
        if (!stalled) {
            while (inflight < ahead) {  // keep one batch ahead of the readback being waited for
                // few candidates left: the build is about to end, and every round enqueued beyond its end is five launches of
                // nothing (up to eight such rounds at four per readback: 0.1 ms of a build) -> two per readback from here on
                // ... and with tens of thousands of candidates (a batch of views) a round is hundreds of microseconds: the host is
                // ahead anyway, and one round per readback wastes the fewest at the end (128 views: 9.5 ms at 4, 9.0 at 2, 8.9 at 1)
                const int per = batch_env ? batch : (ncand >= 16384 ? 1 : (ncand <= 512 ? 2 : batch));
                const int rc = enqueue_rounds(always_careful ? 1 : per, always_careful);
                if (rc != TOHIP_OK) return drain(rc);
                round_of[wslot] = round;
                e = post_readback();
                if (e != hipSuccess) return drain((int)e);
            }
        } else if (inflight == 0) {
            // every enqueued round has reported and the last one accepted nobody: one round with converged ownership
            if (trace) fprintf(stderr, "hull: careful round %d (candidates %d)\n", round, ncand);
            const int rc = enqueue_rounds(1, true);
            if (rc != TOHIP_OK) return drain(rc);
            round_of[wslot] = round;
            careful_at = round;
            e = post_readback();
            if (e != hipSuccess) return drain((int)e);
        }
        const int posted_at = round_of[rslot];
        e = collect();
        if (e != hipSuccess) return drain((int)e);
        rounds_seen = posted_at;
        if (h[kCtrlError]) {
            if (trace) fprintf(stderr, "hull: error bits %d at round %d\n", h[kCtrlError], posted_at);
            return drain((h[kCtrlError] & kErrCapacity) ? TOHIP_ENOSPC : TOHIP_ENOTCONV);
        }
        nf = h[kCtrlNFaces + 8] < b.fcap ? h[kCtrlNFaces + 8] : b.fcap;
        ncand = candidates(posted_at & 1);
        if (trace) fprintf(stderr, "hull: round %d faces %d live %d candidates %d accepted(last) %d\n", posted_at, nf, h[kCtrlNLive], ncand,
                           h[kCtrlAccepted]);
        if (ncand == 0) {  // no face has a point outside it: the hull is complete (rounds still in flight are no-ops) ...
            if (b.sub > 1) continue;  // ... of the sample: time for the other points
            break;
        }
        if (h[kCtrlAccepted] <= 0) {
            // converged ownership always admits the best candidate: a careful round without progress = inconsistent predicates
            if (posted_at == careful_at || always_careful) return drain(TOHIP_ENOTCONV);
            stalled = true;   // let what is in flight report, then run a careful round
            continue;
        }
        stalled = false;
        if (++batches_since_compaction >= compact_every && h[kCtrlNLive] > 4096 && posted_at > compaction_pending_until) {
            // drop the points that have retired inside the hull from the list the point kernels walk
            const int nlive = h[kCtrlNLive], ntl = (nlive + 1023) / 1024;
            k_live_count<<<ntl, TO_BLOCK, 0, st>>>(b, nlive, b.tile_cnt);
            launch_scan_tiles(b.tile_cnt, ntl, b.tile_off, b.ctrl + kCtrlNLive, st);
            k_live_write<<<ntl, TO_BLOCK, 0, st>>>(b, nlive, b.tile_off);
            {
                const hipError_t el = hipGetLastError();   // a report may be in flight into the shared mapped records: drain before leaving
                if (el != hipSuccess) return drain((int)el);
            }
            int* t = b.live; b.live = b.live2; b.live2 = t;
            batches_since_compaction = 0;
            live_bound = nlive;  // the new count is on the device only; this bounds it
            compaction_pending_until = round;  // readbacks posted up to now still show the count before this compaction
        }
    }
    if (round >= max_rounds && ncand > 0) return drain(TOHIP_ENOTCONV);
    round = rounds_seen;
    if (trace) fprintf(stderr, "hull: %d rounds; host: %.0f us enqueueing rounds, %.0f us waiting for readbacks\n", round, host_enqueue_us, host_wait_us);
    if (rounds_out) *rounds_out = round;
    k_mark_vertices<<<nblocks(nf), TO_BLOCK, 0, st>>>(b);
    TO_HIP_CHECK_LAUNCH();
    while (inflight > 0) {  // the mapped records are this thread's next build's too
        e = collect();
        if (e != hipSuccess) return (int)e;
    }
    return TOHIP_OK;
}

// ascending indices of the flagged points -> out (capacity cap), *total on the device
static int compact(const Bufs& b, int* out, int cap, int* total_dev, hipStream_t st) {
    const int ntiles = (b.m1 + 1023) / 1024;
    k_flag_count<<<ntiles, TO_BLOCK, 0, st>>>(b.vflag, b.m1, b.tile_cnt);
    TO_HIP_CHECK_LAUNCH();
    launch_scan_tiles(b.tile_cnt, ntiles, b.tile_off, total_dev, st);
    TO_HIP_CHECK_LAUNCH();
    k_flag_write<<<ntiles, TO_BLOCK, 0, st>>>(b.vflag, b.m1, b.tile_off, out, cap);
    TO_HIP_CHECK_LAUNCH();
    return TOHIP_OK;
}

// ---- batched hidden-point removal: per-segment flip radius, per-segment "drop the last hull vertex" -------
__device__ __forceinline__ int find_src_seg(const Bufs& b, int r) {  // segment of source row r (seg_off[s] - s <= r)
    int lo = 0, hi = b.nseg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (b.seg_off[mid] - mid <= r) lo = mid; else hi = mid;
    }
    return lo;
}

// (wave_seg for source rows: r0 wave-uniform, *end = the first row behind the segment)
__device__ __forceinline__ int wave_src_seg(const Bufs& b, int r0, int* end) {
    const int r = __builtin_amdgcn_readfirstlane(r0);
    int lo = 0, hi = b.nseg;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (b.seg_off[mid] - mid <= r) lo = mid; else hi = mid;
    }
    *end = b.seg_off[lo + 1] - (lo + 1);
    return lo;
}

__global__ void __launch_bounds__(TO_BLOCK) k_norm_max_seg(Bufs b, const float* __restrict__ xyz, int n) {
    // contiguous run per wave, running maximum of the current segment in a register (as k_bbox)
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * TO_WAVES_PER_BLOCK + (threadIdx.x >> 6), nwaves = gridDim.x * TO_WAVES_PER_BLOCK;
    const int chunk = ((n + nwaves - 1) / nwaves + 63) / 64 * 64;
    const int64_t begin64 = (int64_t)wave * chunk;
    const int begin = begin64 < n ? (int)begin64 : n, end = begin64 + chunk < n ? (int)(begin64 + chunk) : n;
    int cur = -1, m = 0, cur_seg = -1, cur_end = -1;
    auto flush = [&]() {
        if (cur >= 0) {
            for (int s = 32; s > 0; s >>= 1) m = max(m, __shfl_xor(m, s));
            if (lane == 0) atomicMax(&b.flip_max[cur], m);
        }
        m = 0;
    };
    for (int i0 = begin; i0 < end; i0 += 64) {
        const int i = i0 + lane;
        int sg = -1, v = 0;
        if (i0 >= cur_end) cur_seg = wave_src_seg(b, i0, &cur_end);   // (wave-uniform)
        if (i < end) {
            sg = i0 + 63 < cur_end ? cur_seg : find_src_seg(b, i);
            const Row3 p = load_row(xyz, i);
            v = __float_as_int(flip_norm(p.x, p.y, p.z)) & 0x7fffffff;
        }
        const int s0 = __shfl(sg, 0);
        if (__all(sg == s0 || sg < 0)) {
            if (s0 != cur) { flush(); cur = s0; }
            m = max(m, v);
        } else {
            flush();
            cur = -1;
            if (sg >= 0) atomicMax(&b.flip_max[sg], v);
        }
    }
    flush();
}

__global__ void __launch_bounds__(TO_BLOCK)
k_flip_seg(Bufs b, const float* __restrict__ xyz, int n, float scale, float* __restrict__ flipped) {
    const int stride = gridDim.x * TO_BLOCK;
    for (int i = blockIdx.x * TO_BLOCK + threadIdx.x; i - (int)(threadIdx.x & 63) < n; i += stride) {
        int send;
        const int i0 = i - (int)(threadIdx.x & 63), s0 = wave_src_seg(b, i0, &send);   // (all lanes of the wave take part)
        if (i >= n) continue;
        const float radius = __int_as_float(b.flip_max[i0 + 63 < send ? s0 : find_src_seg(b, i)]) * scale;  // tools.py:45, per viewpoint
        const Row3 p = load_row(xyz, i);
        const float x = p.x, y = p.y, z = p.z;
        const float nr = flip_norm(x, y, z);
        const float r = radius - nr;
        Row3 o;
        o.x = (2.0f * (r * x)) / nr + x;  // tools.py:46-52
        o.y = (2.0f * (r * y)) / nr + y;
        o.z = (2.0f * (r * z)) / nr + z;
        reinterpret_cast<Row3*>(flipped)[i] = o;
    }
}

// first position of every segment in the ascending vertex list (+ the total as entry nseg), and the number of
// visible points it yields: all of its hull vertices but the last (tools.py:79)
__global__ void __launch_bounds__(TO_BLOCK) k_seg_ranges(Bufs b, const int* __restrict__ total) {
    const int T = *total;
    const int stride = gridDim.x * TO_BLOCK;
    for (int s = blockIdx.x * TO_BLOCK + threadIdx.x; s <= b.nseg; s += stride) {
        const int key = b.seg_off[s];
        int lo = 0, hi = T;  // lower bound of key in idx_all[0:T)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (b.idx_all[mid] < key) lo = mid + 1; else hi = mid;
        }
        b.seg_start[s] = lo;
    }
}

__global__ void __launch_bounds__(TO_BLOCK) k_seg_counts(Bufs b) {
    const int stride = gridDim.x * TO_BLOCK;
    for (int s = blockIdx.x * TO_BLOCK + threadIdx.x; s < b.nseg; s += stride)
        b.seg_cnt[s] = max(0, b.seg_start[s + 1] - b.seg_start[s] - 1);
}

__global__ void __launch_bounds__(TO_BLOCK)
k_seg_write(Bufs b, const int* __restrict__ total, const int32_t* __restrict__ out_off, int32_t* __restrict__ out,
            float* __restrict__ mask) {
    const int T = *total;
    const int stride = gridDim.x * TO_BLOCK;
    for (int p = blockIdx.x * TO_BLOCK + threadIdx.x; p < T; p += stride) {
        int lo = 0, hi = b.nseg;  // the last segment whose start is <= p (earlier ones with the same start are empty)
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (b.seg_start[mid] <= p) lo = mid; else hi = mid;
        }
        if (p >= b.seg_start[lo + 1] - 1) continue;  // the segment's last hull vertex is dropped
        const int row = b.idx_all[p] - lo;           // back to a row of the caller's concatenated array
        out[out_off[lo] + p - b.seg_start[lo]] = row;
        if (mask) mask[row] = 1.0f;
    }
}

}  // namespace hull

extern "C" size_t tohip_hpr_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    return hull::carve(nullptr, nullptr, n, 1, hull::default_face_capacity(n + 1, 1));
}

extern "C" size_t tohip_hpr_batched_workspace_bytes(int64_t n_total, int32_t n_segments) {
    if (n_total < 0 || n_segments <= 0) return 256;
    return hull::carve(nullptr, nullptr, n_total, n_segments, hull::default_face_capacity(n_total + n_segments, n_segments));
}

extern "C" int tohip_spherical_flip(const float* xyz, int64_t n, float param, float* flipped, float* radius_out,
                                    void* workspace, size_t workspace_bytes, void* stream_) {
    if (!xyz || !flipped || !workspace || n <= 0) return TOHIP_EINVAL;
    if (workspace_bytes < TOHIP_FLIP_WORKSPACE_BYTES) return TOHIP_ENOSPC;
    return launch_flip(xyz, n, param, flipped, radius_out, (int*)workspace, (hipStream_t)stream_);
}

// Hull vertices (ascending) of pts (n,3), optionally with the origin appended as point n
// (tools.py:56-64).  idx capacity n+1; *count on the device.  Synchronises the stream.
extern "C" int tohip_convex_hull_vertices(const float* pts, int64_t n, int with_origin, int32_t* idx, int32_t* count,
                                          int32_t* rounds_host, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!pts || !idx || !count || !workspace || n < 4 || n > (int64_t)100000000) return TOHIP_EINVAL;
    const int fcap = hull::faces_for_bytes(n, 1, workspace_bytes);
    if (fcap == 0) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    hull::Bufs b;
    if (hull::carve(&b, (char*)workspace, n, 1, fcap) > workspace_bytes) return TOHIP_ENOSPC;   // (faces_for_bytes and carve disagree: a build error)
    hull::k_single_segment<<<1, 1, 0, st>>>(b);
    TO_HIP_CHECK_LAUNCH();
    int rounds = 0;
    int rc = hull::build(b, pts, with_origin, n, st, &rounds);
    if (rc != TOHIP_OK) return rc;
    if (rounds_host) *rounds_host = rounds;
    return hull::compact(b, idx, (int)n + 1, count, st);
}

extern "C" int tohip_hidden_pts_removal(const float* xyz, int64_t n, float param, int32_t* visible_idx,
                                        int32_t* visible_count, float* mask, void* workspace, size_t workspace_bytes,
                                        void* stream_) {
    if (!xyz || !visible_idx || !visible_count || !workspace || n < 4 || n > (int64_t)100000000) return TOHIP_EINVAL;
    const int fcap = hull::faces_for_bytes(n, 1, workspace_bytes);
    if (fcap == 0) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    hull::Bufs b;
    if (hull::carve(&b, (char*)workspace, n, 1, fcap) > workspace_bytes) return TOHIP_ENOSPC;   // (faces_for_bytes and carve disagree: a build error)
    hull::k_single_segment<<<1, 1, 0, st>>>(b);
    TO_HIP_CHECK_LAUNCH();
    int rc = launch_flip(xyz, n, param, b.flipped, nullptr, b.flip_max, st);
    if (rc != TOHIP_OK) return rc;
    rc = hull::build(b, b.flipped, 1, n, st, nullptr);
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

// hidden_pts_removal of n_segments independent clouds in ONE pass (each seen from its own origin: the per-camera
// use of /root/reference/src/pc_processor.py:171-178, one segment per camera or waypoint).  xyz holds the segments
// end to end, segment s = rows [seg_offsets_host[s], seg_offsets_host[s+1]).  All hulls advance in the same rounds,
// so the cost of a round (launches + one readback) is shared by every viewpoint.
//   visible_idx          n_total ints: rows of xyz, ascending; segment s's visible rows are
//                        visible_idx[seg_visible_offsets[s] : seg_visible_offsets[s+1]]
//   seg_visible_offsets  n_segments + 1 ints (device)
//   mask                 n_total floats 0/1, or NULL
//   seg_status           n_segments ints (device) or NULL: 0 ok; 1 fewer than 4 points and 2 flat — Qhull raises for
//                        both (the reference would throw); here such a segment just reports no visible points
extern "C" int tohip_hidden_pts_removal_batched(const float* xyz, const int64_t* seg_offsets_host, int32_t n_segments,
                                                float param, int32_t* visible_idx, int32_t* seg_visible_offsets,
                                                float* mask, int32_t* seg_status, void* workspace, size_t workspace_bytes,
                                                void* stream_) {
    if (!seg_offsets_host || n_segments <= 0 || !visible_idx || !seg_visible_offsets || !workspace) return TOHIP_EINVAL;
    if (seg_offsets_host[0] != 0) return TOHIP_EINVAL;
    for (int32_t s = 0; s < n_segments; ++s)
        if (seg_offsets_host[s + 1] < seg_offsets_host[s]) return TOHIP_EINVAL;
    const int64_t n = seg_offsets_host[n_segments];
    if (n + n_segments > (int64_t)100000000 || (n > 0 && !xyz)) return TOHIP_EINVAL;
    const int fcap = hull::faces_for_bytes(n, n_segments, workspace_bytes);
    if (fcap == 0) return TOHIP_ENOSPC;
    hipStream_t st = (hipStream_t)stream_;
    hull::Bufs b;
    if (hull::carve(&b, (char*)workspace, n, n_segments, fcap) > workspace_bytes) return TOHIP_ENOSPC;   // (faces_for_bytes and carve disagree: a build error)
    std::vector<int> off((size_t)n_segments + 1);
    for (int32_t s = 0; s <= n_segments; ++s) off[s] = (int)(seg_offsets_host[s] + s);  // + one origin slot per segment
    hipError_t e = hipMemcpyAsync(b.seg_off, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(st);  // `off` is pageable host memory owned by this call
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(b.flip_max, 0, sizeof(int) * (size_t)n_segments, st);
    if (e != hipSuccess) return (int)e;
    if (n > 0) {
        hull::k_norm_max_seg<<<hull::nblocks(n, 1024), TO_BLOCK, 0, st>>>(b, xyz, (int)n);
        TO_HIP_CHECK_LAUNCH();
        hull::k_flip_seg<<<hull::nblocks(n), TO_BLOCK, 0, st>>>(b, xyz, (int)n, (float)pow(10.0, (double)param), b.flipped);
        TO_HIP_CHECK_LAUNCH();
    }
    int64_t max_seg = 0;
    for (int32_t s = 0; s < n_segments; ++s) max_seg = std::max<int64_t>(max_seg, seg_offsets_host[s + 1] - seg_offsets_host[s]);
    int rc = hull::build(b, b.flipped, 1, max_seg, st, nullptr);
    if (rc != TOHIP_OK) return rc;
    int* total = b.ctrl + hull::kCtrlChanged;
    rc = hull::compact(b, b.idx_all, b.m1, total, st);
    if (rc != TOHIP_OK) return rc;
    hull::k_seg_ranges<<<hull::nblocks(n_segments + 1), TO_BLOCK, 0, st>>>(b, total);
    TO_HIP_CHECK_LAUNCH();
    hull::k_seg_counts<<<hull::nblocks(n_segments), TO_BLOCK, 0, st>>>(b);
    TO_HIP_CHECK_LAUNCH();
    launch_scan_tiles(b.seg_cnt, n_segments, seg_visible_offsets, seg_visible_offsets + n_segments, st);
    TO_HIP_CHECK_LAUNCH();
    if (mask && n > 0) {
        e = hipMemsetAsync(mask, 0, sizeof(float) * (size_t)n, st);
        if (e != hipSuccess) return (int)e;
    }
    hull::k_seg_write<<<hull::nblocks(b.m1), TO_BLOCK, 0, st>>>(b, total, seg_visible_offsets, visible_idx, mask);
    TO_HIP_CHECK_LAUNCH();
    if (seg_status) {
        e = hipMemcpyAsync(seg_status, b.seg_status, sizeof(int) * (size_t)n_segments, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return (int)e;
    }
    return TOHIP_OK;
}
