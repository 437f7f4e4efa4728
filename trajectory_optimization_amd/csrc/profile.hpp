// profile.hpp — optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default; when enabled every instrumented launch is bracketed by two hipEventRecord calls on the
// same stream the kernel is enqueued on, and tohip_profile_read() sums the elapsed times per kernel id.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#define TOHIP_PROF_PASS1 0
#define TOHIP_PROF_PASS2 1
#define TOHIP_PROF_BWD 2
#define TOHIP_PROF_REWARD 3
#define TOHIP_PROF_PROBE 4
#define TOHIP_PROF_FINISH 5
#define TOHIP_PROF_NKERNELS 6

namespace toprof {
struct Rec { int id; hipEvent_t a, b; };
struct State {
    bool on = false;
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
};
inline State& state() { static State s; return s; }
inline hipEvent_t get_event() {
    State& s = state();
    if (s.used == s.pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        s.pool.push_back(e);
    }
    return s.pool[s.used++];
}
struct Scope {
    hipStream_t st; hipEvent_t b = nullptr;
    Scope(int id, hipStream_t st_) : st(st_) {
        State& s = state();
        if (!s.on) return;
        hipEvent_t a = get_event();
        b = get_event();
        if (!a || !b) { b = nullptr; return; }
        (void)hipEventRecord(a, st);
        s.recs.push_back({id, a, b});
    }
    ~Scope() { if (b) (void)hipEventRecord(b, st); }
};
}  // namespace toprof
#define TO_PROF(id, st) toprof::Scope prof_scope__(id, st)
