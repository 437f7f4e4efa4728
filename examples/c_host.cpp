// A host with no Python and no torch in it: the visibility forward + reward + backward of a small trajectory through
// the C ABI of include/trajopt_hip.h — what a C or C++ caller links against — as the three separate calls a sharded run
// makes around its all-reduce, and as the one fused call of a single GPU.
//
//   hipcc --offload-arch=gfx950 -O2 -Iinclude examples/c_host.cpp -Ltrajectory_optimization_amd -ltrajopt_hip
//         -Wl,-rpath,$PWD/trajectory_optimization_amd -o c_host && ./c_host [n_points] [n_wps]
//
// Prints one line of JSON (mean reward, visibility loss, the first waypoint's gradients); tests/test_hip_c_host.py
// checks it against the Python path on the same inputs.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "trajopt_hip.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define TO_OK(x) do { int rc_ = (x); if (rc_ != TOHIP_OK) { std::fprintf(stderr, "%s: %s\n", #x, tohip_error_string(rc_)); return 3; } } while (0)

template <typename T>
static T* dev_alloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) + 256) != hipSuccess) return nullptr;
    return static_cast<T*>(p);
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 20000;
    const int64_t W = argc > 2 ? std::atoll(argv[2]) : 6;
    if (tohip_abi_version() != TOHIP_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }

    // inputs: a slab of points (a 32-bit LCG the test reproduces) and waypoints along x looking down +x
    std::vector<float> pts(3 * n), poses(3 * W), quats(4 * W);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    for (int64_t i = 0; i < n; ++i) {
        pts[3 * i] = rnd() * 30.0f - 15.0f; pts[3 * i + 1] = rnd() * 30.0f - 15.0f; pts[3 * i + 2] = rnd() * 4.0f - 2.0f;
    }
    for (int64_t w = 0; w < W; ++w) {
        poses[3 * w] = -8.0f + 16.0f * (float)w / (float)(W > 1 ? W - 1 : 1); poses[3 * w + 1] = 0.5f * (float)w; poses[3 * w + 2] = 0.0f;
        // optical frame (z forward) looking along +x: R = Ry(90deg) Rz(-90deg)  ->  quaternion (0.5, -0.5, 0.5, -0.5)
        quats[4 * w] = 0.5f; quats[4 * w + 1] = -0.5f; quats[4 * w + 2] = 0.5f; quats[4 * w + 3] = -0.5f;
    }
    tohip_camera cam = {{758.03967f, 0.f, 621.46572f, 0.f, 761.62359f, 756.86402f, 0.f, 0.f, 1.f}, 1232.f, 1616.f, 1.0f, 5.0f, 1e-6f};

    const int64_t npad = tohip_padded_points(n);
    float* d_pts = dev_alloc<float>(3 * n);
    float* d_poses = dev_alloc<float>(3 * W);
    float* d_quats = dev_alloc<float>(4 * W);
    char* d_packed = dev_alloc<char>(tohip_packed_cloud_bytes(n));
    const size_t pack_ws = tohip_pack_workspace_bytes(n), traj_ws = tohip_traj_workspace_bytes(n, W);
    char* d_ws = dev_alloc<char>(pack_ws > traj_ws ? pack_ws : traj_ws);
    float* d_lo = dev_alloc<float>(npad);
    float* d_minmax = dev_alloc<float>(2 * W);
    float* d_rewards = dev_alloc<float>(n);
    float* d_scalars = dev_alloc<float>(4);
    float* d_gout = dev_alloc<float>(1);
    float* d_pg = dev_alloc<float>(3 * W);
    float* d_qg = dev_alloc<float>(4 * W);
    if (!d_pts || !d_poses || !d_quats || !d_packed || !d_ws || !d_lo || !d_minmax || !d_rewards || !d_scalars || !d_gout || !d_pg || !d_qg) return 2;
    HIP_OK(hipMemcpy(d_pts, pts.data(), pts.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_poses, poses.data(), poses.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_quats, quats.data(), quats.size() * sizeof(float), hipMemcpyHostToDevice));
    const float one = 1.0f;
    HIP_OK(hipMemcpy(d_gout, &one, sizeof(float), hipMemcpyHostToDevice));

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    TO_OK(tohip_pack_cloud(d_pts, n, 1, d_packed, d_ws, pack_ws, st));
    HIP_OK(hipMemsetAsync(d_ws, 0, traj_ws, st));  // the workspace is zero-filled once before its first use
    TO_OK(tohip_traj_forward(d_packed, n, d_poses, d_quats, W, &cam, nullptr, 0, nullptr, d_lo, d_minmax, d_rewards, d_ws, traj_ws, st));
    TO_OK(tohip_traj_reward(d_packed, d_lo, n, cam.eps, 1, d_rewards, d_scalars, d_ws, traj_ws, st));
    TO_OK(tohip_traj_backward(d_packed, n, W, &cam, nullptr, 0, nullptr, d_lo, nullptr, d_scalars, d_gout, d_pg, d_qg, d_ws, traj_ws,
                              st));
    HIP_OK(hipStreamSynchronize(st));
    // the same step as ONE call (four launches): what a single-GPU caller uses, no collective between forward and backward
    float* d_pg2 = dev_alloc<float>(3 * W);
    float* d_qg2 = dev_alloc<float>(4 * W);
    float* d_scalars2 = dev_alloc<float>(4);
    float* d_rewards2 = dev_alloc<float>(n);
    if (!d_pg2 || !d_qg2 || !d_scalars2 || !d_rewards2) return 2;
    TO_OK(tohip_traj_forward_backward(d_packed, n, d_poses, d_quats, W, &cam, nullptr, 0, nullptr, d_lo, d_minmax, d_rewards2, d_scalars2, d_gout,
                                      d_pg2, d_qg2, d_ws, traj_ws, st));
    HIP_OK(hipStreamSynchronize(st));
    float scalars2[4];
    std::vector<float> pg2(3 * W), qg2(4 * W), r1(n), r2(n);
    HIP_OK(hipMemcpy(scalars2, d_scalars2, sizeof(scalars2), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(pg2.data(), d_pg2, pg2.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(qg2.data(), d_qg2, qg2.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(r1.data(), d_rewards, r1.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(r2.data(), d_rewards2, r2.size() * sizeof(float), hipMemcpyDeviceToHost));
    bool fused_rewards_equal = scalars2[0] == 0.f ? false : true;
    for (int64_t i = 0; i < n; ++i) fused_rewards_equal = fused_rewards_equal && r1[i] == r2[i];

    float scalars[4];
    std::vector<float> pg(3 * W), qg(4 * W);
    HIP_OK(hipMemcpy(scalars, d_scalars, sizeof(scalars), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(pg.data(), d_pg, pg.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(qg.data(), d_qg, qg.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::printf("{\"n_points\": %lld, \"n_wps\": %lld, \"mean_reward\": %.9g, \"loss_vis\": %.9g, \"poses_grad\": [", (long long)n,
                (long long)W, scalars[0], scalars[1]);
    for (size_t i = 0; i < pg.size(); ++i) std::printf("%s%.9g", i ? ", " : "", pg[i]);
    std::printf("], \"quats_grad\": [");
    for (size_t i = 0; i < qg.size(); ++i) std::printf("%s%.9g", i ? ", " : "", qg[i]);
    std::printf("], \"fused\": {\"mean_reward\": %.9g, \"loss_vis\": %.9g, \"rewards_equal\": %s, \"poses_grad\": [", scalars2[0], scalars2[1],
                fused_rewards_equal ? "true" : "false");
    for (size_t i = 0; i < pg2.size(); ++i) std::printf("%s%.9g", i ? ", " : "", pg2[i]);
    std::printf("], \"quats_grad\": [");
    for (size_t i = 0; i < qg2.size(); ++i) std::printf("%s%.9g", i ? ", " : "", qg2[i]);
    std::printf("]}}\n");
    return 0;
}
