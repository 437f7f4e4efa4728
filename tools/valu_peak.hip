// valu_peak.hip — what one gfx950 SIMD sustains per wave64 VALU instruction, measured.
//
// The big kernels of this library are bound by vector-instruction issue, so their roofline
// needs the price of each instruction class on THIS chip, with 1..8 waves per SIMD:
//   v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_exp_f32 / v_rcp_f32 / v_log_f32, v_min_i32 with a DPP operand,
//   v_mov_b32 DPP (row_bcast), v_med3_f32, v_cmp + s_or (the need mask).
// Output: JSON, cycles of SIMD time per wave-instruction = (kernel time x in-kernel clock x 1024 SIMDs) / wave-instrs.
//
//   hipcc -O3 --offload-arch=gfx950 tools/valu_peak.hip -o gpurun_out/valu_peak && gpurun_out/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

#define ITERS 2048
#define NCHAIN 16

enum { OP_FMA, OP_PK_FMA, OP_PK_MUL, OP_PK_ADD, OP_EXP, OP_RCP, OP_LOG, OP_MIN_DPP, OP_MOV_DPP_BCAST, OP_MED3, OP_CMP, OP_MIX, OP_N };
static const char* op_names[OP_N] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_exp_f32", "v_rcp_f32", "v_log_f32",
                                     "v_min_i32_dpp", "v_mov_b32_dpp_row_bcast", "v_med3_f32", "v_cmp_ge_f32", "mix_6pk_2exp"};

template <int OP>
__global__ void __launch_bounds__(256) k_peak(float* out, unsigned long long* clk, float seed) {
    float a[NCHAIN];
    f2 p[NCHAIN / 2];
    const float b = seed * 1.0000001f, c = seed * 1e-7f;
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) a[i] = seed + (float)(threadIdx.x + i) * 1e-3f;
#pragma unroll
    for (int i = 0; i < NCHAIN / 2; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    const f2 pb = f2{b, b}, pc = f2{c, c};
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < NCHAIN; ++i) {
            if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == OP_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == OP_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            if (OP == OP_MIN_DPP) asm volatile("v_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            if (OP == OP_MOV_DPP_BCAST) asm volatile("v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]));
            if (OP == OP_MED3) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b));
            if (OP == OP_CMP) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
        }
#pragma unroll
        for (int i = 0; i < NCHAIN / 2; ++i) {
            if (OP == OP_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            if (OP == OP_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            if (OP == OP_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
        }
        if (OP == OP_MIX) {
            // the shape of one evaluation pair in the forward kernels: 6 packed FMA-class per 2 transcendentals
#pragma unroll
            for (int i = 0; i < NCHAIN / 2; ++i) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[2 * i]));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
                asm volatile("v_exp_f32 %0, %0" : "+v"(a[2 * i + 1]));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCHAIN; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < NCHAIN / 2; ++i) s += p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;  // keeps the chains alive
    if (threadIdx.x == 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

static int instrs_per_iter(int op) {
    switch (op) {
        case OP_PK_FMA: case OP_PK_MUL: case OP_PK_ADD: return NCHAIN / 2;
        case OP_MIX: return (NCHAIN / 2) * 8;
        default: return NCHAIN;
    }
}

template <int OP>
static void run(int waves_per_simd, float* out, unsigned long long* clk, std::vector<unsigned long long>& hclk, bool first) {
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks: 4 waves, one per SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) k_peak<OP><<<blocks, 256>>>(out, clk, 1.0f);
    CHECK(hipDeviceSynchronize());
    const int reps = 10;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) k_peak<OP><<<blocks, 256>>>(out, clk, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipMemcpy(hclk.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
    // in-kernel clock: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz), median over blocks
    std::vector<double> ghz(blocks);
    for (int b = 0; b < blocks; ++b) ghz[b] = (double)hclk[2 * b] / (double)hclk[2 * b + 1] * 0.1;
    std::sort(ghz.begin(), ghz.end());
    const double clock_ghz = ghz[blocks / 2];
    const double wave_instrs = (double)blocks * 4.0 * ITERS * instrs_per_iter(OP);
    const double per_simd = wave_instrs / 1024.0;
    const double t = ms * 1e-3 / reps;
    const double cyc = t * clock_ghz * 1e9 / per_simd;
    // block lifetime in shader cycles per wave-instruction (what one wave saw)
    std::vector<double> life(blocks);
    for (int b = 0; b < blocks; ++b) life[b] = (double)hclk[2 * b] / ((double)ITERS * instrs_per_iter(OP));
    std::sort(life.begin(), life.end());
    printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"kernel_us\": %.2f, \"clock_ghz\": %.3f, \"simd_cycles_per_wave_instr\": %.3f, "
           "\"wave_lifetime_cycles_per_instr\": %.3f}",
           first ? "" : ",\n", op_names[OP], waves_per_simd, t * 1e6, clock_ghz, cyc, life[blocks / 2]);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

#include <algorithm>

int main() {
    float* out;
    unsigned long long* clk;
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * 256 * 8));
    std::vector<unsigned long long> hclk(2 * 256 * 8);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz_max\": %d, \"rows\": [\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    bool first = true;
    const int wps[] = {1, 2, 4, 8};
    for (int wi = 0; wi < 4; ++wi) {
        const int w = wps[wi];
        run<OP_FMA>(w, out, clk, hclk, first); first = false;
        run<OP_PK_FMA>(w, out, clk, hclk, false);
        run<OP_PK_MUL>(w, out, clk, hclk, false);
        run<OP_PK_ADD>(w, out, clk, hclk, false);
        run<OP_EXP>(w, out, clk, hclk, false);
        run<OP_RCP>(w, out, clk, hclk, false);
        run<OP_LOG>(w, out, clk, hclk, false);
        run<OP_MIN_DPP>(w, out, clk, hclk, false);
        run<OP_MOV_DPP_BCAST>(w, out, clk, hclk, false);
        run<OP_MED3>(w, out, clk, hclk, false);
        run<OP_CMP>(w, out, clk, hclk, false);
        run<OP_MIX>(w, out, clk, hclk, false);
    }
    printf("\n]}\n");
    return 0;
}
