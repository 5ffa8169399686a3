// valu_rate.hip -- what does a wave64 VALU instruction cost on gfx950, as a function of the waves sharing a SIMD?
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.hip && tools/valu_rate > profiles/r03_valu_rate.txt
//
// The blend kernels of this library are bound by VALU issue, not by HBM; their floor is
// (wave64 VALU instructions per SIMD) x (cycles per instruction).  MI355X_MICROARCH.md gives 2 cycles for v_fma_f32 on
// the SIMD-32, rocprofv3's SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU gives 4.15 for the backward blend: this program measures
// it.  Every wave runs `iters` trips of 64 independent-enough instructions of one kind (8 accumulators, so a lone wave
// is not limited by the dependent-issue latency) and reads s_memtime around the loop; grids put 1, 2, 4, 5 or 8
// single-wave workgroups on every SIMD (1024 SIMDs).  Reported: cycles per instruction as ONE wave sees it (its own
// elapsed cycles / its instructions) and as the SIMD sees it (elapsed / instructions of all its waves), plus the
// wall-clock rate, which folds the actual shader clock in.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                          \
    do {                                                                                  \
        hipError_t r_ = (e);                                                              \
        if (r_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));                       \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

enum Kind { FMA, MUL, ADD, PK_FMA, PK_MUL, EXP, RCP, DPP_ADD, CNDMASK, FMA_EXP_MIX, MAD_U32, NUM_KINDS };
static const char* kNames[NUM_KINDS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32 (2 lanes-ops/lane)",
                                        "v_pk_mul_f32 (2 lanes-ops/lane)", "v_exp_f32", "v_rcp_f32",
                                        "v_add_f32_dpp row_ror:4", "v_cndmask_b32", "7 x v_fma_f32 + 1 x v_exp_f32",
                                        "v_mad_u32_u24"};

#define R8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

template <int KIND>
__global__ __launch_bounds__(64) void rate_kernel(unsigned long long* cycles, float* sink, int iters, float s)
{
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    v2f ps = {s, s};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == FMA)
                asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                             "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
            else if (KIND == MUL)
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
            else if (KIND == ADD)
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
            else if (KIND == PK_FMA)
                asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                             "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(ps));
            else if (KIND == PK_MUL)
                asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                             "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(ps));
            else if (KIND == EXP)
                asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                             "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (KIND == RCP)
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (KIND == DPP_ADD)   // each reads a register written 8 instructions earlier through the DPP path
                asm volatile("s_nop 1\n"
                             "v_add_f32_dpp %0, %1, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %2, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %2, %3, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %4, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %4, %5, %4 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %6, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n"
                             "v_add_f32_dpp %6, %7, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %0, %7 row_ror:4 row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            else if (KIND == CNDMASK)
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s) : "vcc");
            else if (KIND == FMA_EXP_MIX)
                asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_exp_f32 %3, %3\n"
                             "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
            else if (KIND == MAD_U32)
                asm volatile("v_mad_u32_u24 %0, %0, %8, %0\n v_mad_u32_u24 %1, %1, %8, %1\n v_mad_u32_u24 %2, %2, %8, %2\n v_mad_u32_u24 %3, %3, %8, %3\n"
                             "v_mad_u32_u24 %4, %4, %8, %4\n v_mad_u32_u24 %5, %5, %8, %5\n v_mad_u32_u24 %6, %6, %8, %6\n v_mad_u32_u24 %7, %7, %8, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (r == 123.456f) sink[0] = r;   // never true: keeps the chains alive
}

template <int KIND>
static void run(int waves_per_simd, int iters, unsigned long long* d_cyc, float* d_sink, hipEvent_t e0, hipEvent_t e1)
{
    const int simds = 1024;
    const int blocks = simds * waves_per_simd;
    const float s = 0.999f;
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, d_cyc, d_sink, iters / 8, s);   // warm-up
    CHECK(hipDeviceSynchronize());
    float best_ms = 1e30f;
    std::vector<unsigned long long> h(blocks);
    double mean_cyc = 0;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, d_cyc, d_sink, iters, s);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best_ms) {
            best_ms = ms;
            CHECK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
            double sum = 0;
            for (auto c : h) sum += (double)c;
            mean_cyc = sum / blocks;
        }
    }
    const double insts = (double)iters * 64.0;
    // s_memtime ticks: MI355X_MICROARCH.md says one tick = one shader cycle; the wall-clock columns do not depend on it
    const double per_wave = mean_cyc / insts;
    const double per_simd = mean_cyc / (insts * waves_per_simd);
    const double ginst_s = insts * blocks / (best_ms * 1e-3) / 1e9;            // wave-instructions per second, whole chip
    const double wall_cyc_2p4 = 2.4e9 * (best_ms * 1e-3) / (insts * waves_per_simd);   // cycles per inst per SIMD if the clock were 2.4 GHz
    printf("%-34s %2d waves/SIMD  %7.2f ticks/inst/wave  %6.2f ticks/inst/SIMD  %8.1f Ginst/s  %6.2f cyc/inst/SIMD@2.4GHz  (%.3f ms)\n",
           kNames[KIND], waves_per_simd, per_wave, per_simd, ginst_s, wall_cyc_2p4, best_ms);
}

template <int KIND>
static void sweep(int iters, unsigned long long* d_cyc, float* d_sink, hipEvent_t e0, hipEvent_t e1)
{
    for (int w : {1, 2, 4, 5, 8}) run<KIND>(w, iters, d_cyc, d_sink, e0, e1);
    printf("\n");
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;   // x 64 instructions per wave
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs, clockRate %d kHz; %d x 64 instructions per wave; single-wave workgroups, 1024 x W of them\n",
           prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, iters);
    unsigned long long* d_cyc;
    float* d_sink;
    CHECK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 1024 * 8));
    CHECK(hipMalloc(&d_sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    sweep<FMA>(iters, d_cyc, d_sink, e0, e1);
    sweep<MUL>(iters, d_cyc, d_sink, e0, e1);
    sweep<ADD>(iters, d_cyc, d_sink, e0, e1);
    sweep<PK_FMA>(iters, d_cyc, d_sink, e0, e1);
    sweep<PK_MUL>(iters, d_cyc, d_sink, e0, e1);
    sweep<EXP>(iters, d_cyc, d_sink, e0, e1);
    sweep<RCP>(iters, d_cyc, d_sink, e0, e1);
    sweep<DPP_ADD>(iters, d_cyc, d_sink, e0, e1);
    sweep<CNDMASK>(iters, d_cyc, d_sink, e0, e1);
    sweep<FMA_EXP_MIX>(iters, d_cyc, d_sink, e0, e1);
    sweep<MAD_U32>(iters, d_cyc, d_sink, e0, e1);
    return 0;
}
