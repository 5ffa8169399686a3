// launch_bench.hip -- host cost of getting a chain of small kernels onto an MI355X queue, measured four ways.
// Decides how the rasterizer issues its ~20 launches per pass (direct launches vs one hipGraphLaunch vs a graph whose
// kernels read their pointer arguments from a device-resident block that a one-kernel "argument writer" refreshes).
//   hipcc --offload-arch=gfx950 -O2 -o tools/launch_bench tools/launch_bench.hip && tools/launch_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e)                                                                              \
    do {                                                                                   \
        hipError_t r_ = (e);                                                               \
        if (r_ != hipSuccess) {                                                            \
            printf("%s failed: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__);     \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

struct ArgBlock {
    uint32_t* out;
    uint32_t add;
    uint32_t pad[29];
};

__global__ void step_kernel(uint32_t* out, uint32_t add) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += add; }
__global__ void step_indirect_kernel(const ArgBlock* a) { if (threadIdx.x == 0 && blockIdx.x == 0) a->out[0] += a->add; }
__global__ void write_args_kernel(ArgBlock* dst, ArgBlock v) { if (threadIdx.x == 0) *dst = v; }
__global__ void flag_kernel(volatile uint32_t* host_flag, uint32_t seq)
{
    if (threadIdx.x == 0) {
        __threadfence_system();
        *host_flag = seq;
    }
}

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
    const int chain = argc > 1 ? atoi(argv[1]) : 20, reps = argc > 2 ? atoi(argv[2]) : 300;
    hipStream_t s, side;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    uint32_t* d;
    CK(hipMalloc(&d, 4096));
    CK(hipMemset(d, 0, 4096));
    ArgBlock* dargs;
    CK(hipMalloc(&dargs, sizeof(ArgBlock)));
    hipEvent_t fork, join;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));

    auto direct = [&](bool with_side) {
        for (int k = 0; k < chain; k++) {
            hipLaunchKernelGGL(step_kernel, dim3(64), dim3(256), 0, s, d, 1u);
            if (with_side && k == 1) {
                CK(hipEventRecord(fork, s));
                CK(hipStreamWaitEvent(side, fork, 0));
                hipLaunchKernelGGL(step_kernel, dim3(64), dim3(256), 0, side, d + 64, 1u);
                CK(hipEventRecord(join, side));
            }
            if (with_side && k == chain - 2) CK(hipStreamWaitEvent(s, join, 0));
        }
    };
    auto measure = [&](const char* name, auto&& fn) {
        for (int i = 0; i < 20; i++) fn();
        CK(hipStreamSynchronize(s));
        const double t0 = now_us();
        for (int i = 0; i < reps; i++) fn();
        const double t1 = now_us();
        CK(hipStreamSynchronize(s));
        const double t2 = now_us();
        printf("%-44s host enqueue %8.2f us/rep   enqueue+drain %8.2f us/rep\n", name, (t1 - t0) / reps, (t2 - t0) / reps);
    };
    printf("chain = %d kernels, reps = %d\n", chain, reps);
    measure("direct launches", [&] { direct(false); });
    measure("direct launches + side-stream fork/join", [&] { direct(true); });

    // graph of the same chain (stream capture)
    hipGraph_t g;
    hipGraphExec_t ge, ge_side, ge_ind;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    direct(false);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    measure("hipGraphLaunch (linear chain)", [&] { CK(hipGraphLaunch(ge, s)); });
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    direct(true);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge_side, g, nullptr, nullptr, 0));
    measure("hipGraphLaunch (chain + side branch)", [&] { CK(hipGraphLaunch(ge_side, s)); });

    // graph whose kernels read their arguments from a device block; the block is refreshed by one ordinary launch
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < chain; k++) hipLaunchKernelGGL(step_indirect_kernel, dim3(64), dim3(256), 0, s, dargs);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge_ind, g, nullptr, nullptr, 0));
    uint32_t* d2;
    CK(hipMalloc(&d2, 4096));
    CK(hipMemset(d2, 0, 4096));
    int flip = 0;
    measure("arg-writer launch + hipGraphLaunch (indirect)", [&] {
        ArgBlock a{};
        a.out = (flip++ & 1) ? d2 : d;
        a.add = 1;
        hipLaunchKernelGGL(write_args_kernel, dim3(1), dim3(64), 0, s, dargs, a);
        CK(hipGraphLaunch(ge_ind, s));
    });
    uint32_t h[2] = {0, 0};
    CK(hipMemcpy(&h[0], d, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&h[1], d2, 4, hipMemcpyDeviceToHost));
    printf("indirect graph: counters %u + %u (alternating targets honoured: %s)\n", h[0], h[1],
           h[1] >= (uint32_t)(chain * (reps + 20) / 2 - chain) ? "yes" : "NO");

    // kernel -> host-mapped flag latency (replaces hipMemcpyAsync + event + hipEventQuery spin)
    volatile uint32_t* hflag;
    CK(hipHostMalloc((void**)&hflag, 64, hipHostMallocMapped | hipHostMallocCoherent));
    *hflag = 0;
    uint32_t* dflag;
    CK(hipHostGetDevicePointer((void**)&dflag, (void*)hflag, 0));
    double worst = 0, sum = 0;
    for (uint32_t i = 1; i <= 200; i++) {
        const double t0 = now_us();
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, dflag, i);
        while (*hflag != i) {
            if (now_us() - t0 > 2e6) {
                printf("flag never arrived\n");
                return 1;
            }
        }
        const double dt = now_us() - t0;
        sum += dt;
        worst = dt > worst ? dt : worst;
    }
    printf("launch -> host sees kernel-written flag: mean %.2f us, worst %.2f us\n", sum / 200, worst);
    // flag visible BEFORE the stream drains?  long chain behind the flag kernel
    {
        const double t0 = now_us();
        hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(64), 0, s, dflag, 1000u);
        for (int i = 0; i < 50; i++) direct(false);
        const double t1 = now_us();
        while (*hflag != 1000u) {}
        const double t2 = now_us();
        CK(hipStreamSynchronize(s));
        const double t3 = now_us();
        printf("flag seen %.1f us after launch (enqueue of the tail took %.1f us, stream drained at %.1f us)\n", t2 - t0,
               t1 - t0, t3 - t0);
    }
    // for comparison: the event path used so far
    {
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        uint32_t* pinned;
        CK(hipHostMalloc((void**)&pinned, 64, hipHostMallocDefault));
        double sum2 = 0;
        for (int i = 0; i < 200; i++) {
            const double t0 = now_us();
            hipLaunchKernelGGL(step_kernel, dim3(1), dim3(64), 0, s, d, 1u);
            CK(hipMemcpyAsync(pinned, d, 16, hipMemcpyDeviceToHost, s));
            CK(hipEventRecord(ev, s));
            while (hipEventQuery(ev) == hipErrorNotReady) {}
            sum2 += now_us() - t0;
        }
        printf("launch + memcpyAsync D2H + event spin: mean %.2f us\n", sum2 / 200);
    }
    return 0;
}
