// launch_bench2.hip -- follow-up of launch_bench.hip: how to refresh the argument block of a replayed graph safely
// while the host runs ahead of the GPU.
//   A. hipGraphExecKernelNodeSetParams on node 0 (by-value block) before every hipGraphLaunch of ONE exec
//   B. node 0 reads the block from a host-mapped ring slot selected by a device-side launch counter
// Both are checked for correctness with the GPU kept busy (launches pile up) and timed on the host.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e)                                                                          \
    do {                                                                               \
        hipError_t r_ = (e);                                                           \
        if (r_ != hipSuccess) {                                                        \
            printf("%s failed: %s (line %d)\n", #e, hipGetErrorString(r_), __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

struct ArgBlock {
    uint32_t* out;
    uint32_t add;
    uint32_t pad[125];   // 512 B
};

__global__ void write_args_kernel(ArgBlock* dst, ArgBlock v) { if (threadIdx.x == 0) *dst = v; }
__global__ void ring_args_kernel(ArgBlock* dst, const ArgBlock* host_ring, uint32_t ring, uint32_t* cursor)
{
    const uint32_t slot = *cursor % ring;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(host_ring + slot);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    for (uint32_t k = threadIdx.x; k < sizeof(ArgBlock) / 4; k += blockDim.x) d[k] = s[k];
    __syncthreads();
    if (threadIdx.x == 0) *cursor = *cursor + 1;
}
__global__ void step_indirect_kernel(const ArgBlock* a) { if (threadIdx.x == 0 && blockIdx.x == 0) a->out[0] += a->add; }
__global__ void busy_kernel(uint32_t* p, long cycles)
{
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    if (threadIdx.x == 0) p[0] += 1;
}
__global__ void plain_kernel(uint32_t* p) { if (threadIdx.x == 0) p[0] += 1; }

static double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const int chain = 20, n = 64;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    uint32_t *d, *cursor, *scratch;
    CK(hipMalloc(&d, 4 * n));
    CK(hipMalloc(&cursor, 4));
    CK(hipMalloc(&scratch, 4));
    ArgBlock* dargs;
    CK(hipMalloc(&dargs, sizeof(ArgBlock)));

    // ---- A: SetParams on node 0 of one exec ----
    hipGraph_t g;
    hipGraphExec_t ge;
    ArgBlock a0{};
    a0.out = d;
    a0.add = 0;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(write_args_kernel, dim3(1), dim3(64), 0, s, dargs, a0);
    for (int k = 0; k < chain; k++) hipLaunchKernelGGL(step_indirect_kernel, dim3(64), dim3(256), 0, s, dargs);
    CK(hipStreamEndCapture(s, &g));
    size_t nn = 0;
    CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CK(hipGraphGetNodes(g, nodes.data(), &nn));
    hipGraphNode_t node0 = nullptr;
    for (auto nd : nodes) {
        hipKernelNodeParams kp{};
        if (hipGraphKernelNodeGetParams(nd, &kp) == hipSuccess && kp.func == (void*)write_args_kernel) node0 = nd;
    }
    printf("graph has %zu nodes, node0 %s\n", nn, node0 ? "found" : "NOT FOUND");
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int trial = 0; trial < 2 && node0; trial++) {
        CK(hipMemset(d, 0, 4 * n));
        CK(hipStreamSynchronize(s));
        if (trial == 1) hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s, scratch, 300000L);   // ~3 ms at 100 MHz
        const double t0 = now_us();
        for (int i = 0; i < n; i++) {
            ArgBlock a{};
            a.out = d + i;
            a.add = (uint32_t)(i + 1);
            ArgBlock* dst = dargs;
            void* args[2] = {&dst, &a};
            hipKernelNodeParams kp{};
            kp.func = (void*)write_args_kernel;
            kp.gridDim = dim3(1);
            kp.blockDim = dim3(64);
            kp.kernelParams = args;
            CK(hipGraphExecKernelNodeSetParams(ge, node0, &kp));
            CK(hipGraphLaunch(ge, s));
        }
        const double t1 = now_us();
        CK(hipStreamSynchronize(s));
        const double t2 = now_us();
        std::vector<uint32_t> h(n);
        CK(hipMemcpy(h.data(), d, 4 * n, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; i++) bad += h[i] != (uint32_t)(chain * (i + 1));
        printf("A SetParams+launch (%s): host %.2f us/launch, drained after %.1f us, wrong slots %d/%d\n",
               trial ? "GPU busy, host ahead" : "GPU idle", (t1 - t0) / n, t2 - t0, bad, n);
    }

    // ---- B: node 0 pulls the block from a host-mapped ring ----
    const uint32_t ring = 128;
    ArgBlock* hring;
    CK(hipHostMalloc((void**)&hring, sizeof(ArgBlock) * ring, hipHostMallocMapped | hipHostMallocCoherent));
    ArgBlock* dring;
    CK(hipHostGetDevicePointer((void**)&dring, hring, 0));
    CK(hipMemset(cursor, 0, 4));
    hipGraphExec_t gb;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(ring_args_kernel, dim3(1), dim3(128), 0, s, dargs, (const ArgBlock*)dring, ring, cursor);
    for (int k = 0; k < chain; k++) hipLaunchKernelGGL(step_indirect_kernel, dim3(64), dim3(256), 0, s, dargs);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&gb, g, nullptr, nullptr, 0));
    uint32_t seq = 0;
    for (int trial = 0; trial < 2; trial++) {
        CK(hipMemset(d, 0, 4 * n));
        CK(hipStreamSynchronize(s));
        if (trial == 1) hipLaunchKernelGGL(busy_kernel, dim3(1), dim3(64), 0, s, scratch, 300000L);
        const double t0 = now_us();
        for (int i = 0; i < n; i++) {
            ArgBlock& a = hring[seq % ring];
            a.out = d + i;
            a.add = (uint32_t)(i + 1);
            seq++;
            __atomic_thread_fence(__ATOMIC_RELEASE);
            CK(hipGraphLaunch(gb, s));
        }
        const double t1 = now_us();
        CK(hipStreamSynchronize(s));
        const double t2 = now_us();
        std::vector<uint32_t> h(n);
        CK(hipMemcpy(h.data(), d, 4 * n, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; i++) bad += h[i] != (uint32_t)(chain * (i + 1));
        printf("B host ring + launch     (%s): host %.2f us/launch, drained after %.1f us, wrong slots %d/%d\n",
               trial ? "GPU busy, host ahead" : "GPU idle", (t1 - t0) / n, t2 - t0, bad, n);
    }

    // ---- C: graph launches interleaved with ordinary launches (torch ops between our passes) ----
    for (int mix = 0; mix < 3; mix++) {
        CK(hipStreamSynchronize(s));
        const int reps = 300;
        const double t0 = now_us();
        for (int i = 0; i < reps; i++) {
            hring[seq % ring].out = d;
            hring[seq % ring].add = 1;
            seq++;
            for (int k = 0; k < mix; k++) hipLaunchKernelGGL(plain_kernel, dim3(1), dim3(64), 0, s, scratch);
            CK(hipGraphLaunch(gb, s));
        }
        const double t1 = now_us();
        CK(hipStreamSynchronize(s));
        const double t2 = now_us();
        printf("C %d plain launch(es) + graph: host %.2f us/rep, enqueue+drain %.2f us/rep\n", mix, (t1 - t0) / reps,
               (t2 - t0) / reps);
    }
    return 0;
}
