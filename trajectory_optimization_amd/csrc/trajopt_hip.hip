// trajopt_hip.hip — the one translation unit of libtrajopt_hip.so (gfx950 only).
//
// Build:  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared trajopt_hip.hip -o libtrajopt_hip.so
// (-ffp-contract=off: every FMA in the kernels is an explicit fmaf(); see common.hpp.)
#include "traj_kernels.hip"
#include "pose_kernels.hip"
#include "hard_kernels.hip"
#include "hull_kernels.hip"
#include "optim_kernels.hip"
#include "loss_kernels.hip"
#include "ingest_kernels.hip"
#include "render_kernels.hip"

extern "C" int tohip_abi_version(void) { return TOHIP_ABI_VERSION; }

extern "C" const char* tohip_error_string(int code) {
    switch (code) {
        case TOHIP_OK: return "ok";
        case TOHIP_EINVAL: return "invalid argument (null pointer or bad size)";
        case TOHIP_ENOSPC: return "workspace or output capacity too small";
        case TOHIP_ENOTCONV: return "convex hull did not converge within the round limit";
        case TOHIP_ENAN: return "points cannot contain NaN";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

// ---- optional kernel timing (profile.hpp) ---------------------------------------------------------
extern "C" int tohip_profile_enable(int on) {
    toprof::State& s = toprof::state();
    s.on = on != 0;
    s.recs.clear();
    s.used = 0;
    if (!s.on) {   // the header's contract: profiling off = nothing kept
        for (hipEvent_t e : s.pool) (void)hipEventDestroy(e);
        s.pool.clear();
    }
    return TOHIP_OK;
}

extern "C" const char* tohip_profile_name(int id) {
    static const char* names[TOHIP_PROF_NKERNELS] = {"k_traj_pass1", "k_traj_sparse", "k_traj_pairs / k_traj_reward_bwd", "k_traj_reward",
                                                      "k_traj_probe", "k_traj_finish"};
    return (id >= 0 && id < TOHIP_PROF_NKERNELS) ? names[id] : "?";
}

// Synchronises on the recorded events.  ms_sum / counts: TOHIP_PROF_NKERNELS entries each (host).
extern "C" int tohip_profile_read(double* ms_sum, int64_t* counts) {
    if (!ms_sum || !counts) return TOHIP_EINVAL;
    for (int i = 0; i < TOHIP_PROF_NKERNELS; ++i) { ms_sum[i] = 0.0; counts[i] = 0; }
    toprof::State& s = toprof::state();
    for (const auto& r : s.recs) {
        hipError_t e = hipEventSynchronize(r.b);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, r.a, r.b);
        if (e != hipSuccess) return (int)e;
        ms_sum[r.id] += (double)ms;
        counts[r.id] += 1;
    }
    s.recs.clear();
    s.used = 0;
    return TOHIP_OK;
}
