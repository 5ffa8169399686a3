// tools/sort_bench.hip -- times rocPRIM radix_sort_pairs configurations for the two sorts of the binning stage
// (P = 500k 32-bit depth keys; R = 3.64M pairs on 13 tile bits).  Tuning aid, not part of the product.
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <rocprim/rocprim.hpp>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <class Cfg>
int run(const char* name, size_t n, int bits, uint32_t* kin, uint32_t* kout, uint32_t* vin, uint32_t* vout)
{
    size_t bytes = 0;
    CK(rocprim::radix_sort_pairs<Cfg>(nullptr, bytes, kin, kout, vin, vout, n, 0, bits));
    void* tmp;
    CK(hipMalloc(&tmp, bytes + 256));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) CK(rocprim::radix_sort_pairs<Cfg>(tmp, bytes, kin, kout, vin, vout, n, 0, bits));
    CK(hipEventRecord(a));
    const int reps = 20;
    for (int i = 0; i < reps; i++) CK(rocprim::radix_sort_pairs<Cfg>(tmp, bytes, kin, kout, vin, vout, n, 0, bits));
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("%-34s n=%8zu bits=%2d  %8.1f us  temp=%zu\n", name, n, bits, 1000.f * ms / reps, bytes);
    CK(hipFree(tmp));
    return 0;
}

using namespace rocprim;
template <int IPT, int RB>
using OS = radix_sort_config<default_config, default_config, radix_sort_onesweep_config<kernel_config<256, IPT>, kernel_config<256, IPT>, RB>, 0>;
using DefOnesweep = radix_sort_config<default_config, default_config, default_config, 0>;

template <class K>
int run_k(const char* name, size_t n, int bits)
{
    K *kin, *kout; uint32_t *vin, *vout;
    CK(hipMalloc(&kin, n * sizeof(K))); CK(hipMalloc(&kout, n * sizeof(K))); CK(hipMalloc(&vin, n * 4)); CK(hipMalloc(&vout, n * 4));
    std::vector<K> h(n); std::mt19937 rng(2);
    for (auto& x : h) x = (K)(rng() % 6700u);
    CK(hipMemcpy(kin, h.data(), n * sizeof(K), hipMemcpyHostToDevice));
    size_t bytes = 0;
    CK(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, bits));
    void* tmp; CK(hipMalloc(&tmp, bytes + 256));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) CK(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0, bits));
    CK(hipEventRecord(a));
    for (int i = 0; i < 20; i++) CK(rocprim::radix_sort_pairs(tmp, bytes, kin, kout, vin, vout, n, 0, bits));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-34s n=%8zu bits=%2d  %8.1f us\n", name, n, bits, 1000.f * ms / 20);
    return 0;
}

int main()
{
    run_k<uint16_t>("u16 keys default", 3640000, 13);
    run_k<uint16_t>("u16 keys default 16 bits", 3640000, 16);
    run_k<uint32_t>("u32 keys default", 3640000, 13);
    run_k<uint16_t>("u16 keys default (1.8M)", 1800000, 13);
    run_k<uint32_t>("u32 keys default (1.8M)", 1800000, 13);

    const size_t sizes[2] = {500000, 3640000};
    const int bitsv[2] = {32, 13};
    for (int t = 0; t < 2; t++) {
        const size_t n = sizes[t];
        std::vector<uint32_t> h(n);
        std::mt19937 rng(1);
        for (auto& x : h) x = bitsv[t] == 32 ? (0x40000000u + (rng() & 0x00ffffffu) * 3u) : (rng() % 6700u);
        uint32_t *kin, *kout, *vin, *vout;
        CK(hipMalloc(&kin, n * 4)); CK(hipMalloc(&kout, n * 4)); CK(hipMalloc(&vin, n * 4)); CK(hipMalloc(&vout, n * 4));
        CK(hipMemcpy(kin, h.data(), n * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(vin, h.data(), n * 4, hipMemcpyHostToDevice));
        run<default_config>("default", n, bitsv[t], kin, kout, vin, vout);
        run<DefOnesweep>("onesweep default cfg", n, bitsv[t], kin, kout, vin, vout);
        run<OS<12, 8>>("onesweep 256x12 r8", n, bitsv[t], kin, kout, vin, vout);
        run<OS<8, 8>>("onesweep 256x8 r8", n, bitsv[t], kin, kout, vin, vout);
        run<OS<4, 8>>("onesweep 256x4 r8", n, bitsv[t], kin, kout, vin, vout);
        run<OS<8, 7>>("onesweep 256x8 r7", n, bitsv[t], kin, kout, vin, vout);
        run<OS<8, 6>>("onesweep 256x8 r6", n, bitsv[t], kin, kout, vin, vout);
        run<OS<16, 8>>("onesweep 256x16 r8", n, bitsv[t], kin, kout, vin, vout);
        hipFree(kin); hipFree(kout); hipFree(vin); hipFree(vout);
    }
    return 0;
}
