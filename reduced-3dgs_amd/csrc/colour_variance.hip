// colour_variance.hip -- per-camera accumulation step of the SH-band culling statistics.
//
// Replaces, for one camera, the chain of ~25 small torch operators + the calculateColourCUDA kernel of
// /root/reference/submodules/diff-gaussian-rasterization/reduced_3dgs.cu:143-198 and
// reduced_3dgs/sh_culling.cu:6-90 by ONE fused per-Gaussian kernel: truncated colours for bands 0..deg,
// transmittance weight, weighted colour distances, running weighted mean and variance, all updated in place.
// The counter-mode forward (forward.cu:560-564; blend.hip COUNTERS) supplies touched pixels and summed
// transmittance.  Reference quirks kept (see oracle/colour_variance.py): the variance update uses the updated
// mean in both factors (tensor aliasing at reduced_3dgs.cu:185), colour slots above a Gaussian's own degree are
// 0, Gaussians not present in a view keep their statistics.
#include "common.h"

namespace r3 {

constexpr int kCvBlock = 256;
constexpr int kCvWaveShFloats = 64 * 48 + (64 * 48) / 32;

__device__ __forceinline__ int cskew(int e) { return e + (e >> 5); }

struct CvShRowLds {
    const float* base;
    int roff;
    __device__ __forceinline__ float at(int e) const { return base[cskew(roff + e)]; }
};

struct CvArgs {
    int P, M, max_deg;
    const int* degs;
    const float* means;
    const float* campos;
    const float* shs;
    const int* radii;
    const int* touched;
    const float* transm;
    float* wSum;      // [P]
    float* wSumSq;    // [P]
    float* mean;      // [P][3]
    float* variance;  // [P][3]
    float* accum;     // [P][max_deg]
};

__global__ __launch_bounds__(kCvBlock) void colour_variance_accumulate_kernel(CvArgs a)
{
    __shared__ float s_sh[kCvBlock / 64][kCvWaveShFloats];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.P, M = a.M;
    const int i = blockIdx.x * kCvBlock + tid;
    const bool valid = i < P;
    const int wave_first = blockIdx.x * kCvBlock + wave * 64;
    const bool present = valid && a.radii[i] > 0;
    const int nrows = max(0, min(64, P - wave_first));
    const int span_len = nrows * 3 * M;
    const long span_first = 3L * M * wave_first;
    float* lds = s_sh[wave];
    if (__ballot(present) != 0ull) {
        const float* src = a.shs + span_first;
        for (int e = lane; e < span_len; e += 64) lds[cskew(e)] = src[e];
    }
    __syncthreads();
    if (!valid) return;

    // weight of this view: mean transmittance over the pixels the Gaussian blended into (reduced_3dgs.cu:145)
    const float w = a.transm[i] / fmaxf((float)a.touched[i], 1.0f);
    const float ws = a.wSum[i] + w;
    a.wSum[i] = ws;
    a.wSumSq[i] += w * w;
    if (!present) return;  // colours are zeroed for absent Gaussians: distances 0, mean / variance untouched

    float col[12];
#pragma unroll
    for (int k = 0; k < 12; k++) col[k] = 0.f;
    const float campos[3] = {a.campos[0], a.campos[1], a.campos[2]};
    CvShRowLds row{lds, lane * 3 * M};
    sh_truncated_colours(a.degs[i], a.max_deg + 1, row, a.means[3 * i], a.means[3 * i + 1], a.means[3 * i + 2], campos, col);
    const float* full = col + 3 * a.max_deg;
    for (int cur = 0; cur < a.max_deg; cur++) {
        const float d0 = full[0] - col[3 * cur], d1 = full[1] - col[3 * cur + 1], d2 = full[2] - col[3 * cur + 2];
        float d = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        if (d != d) d = 0.f;
        a.accum[(size_t)i * a.max_deg + cur] += w * d;
    }
    float coeff = w / ws;
    if (coeff != coeff) coeff = 0.f;
    for (int ch = 0; ch < 3; ch++) {
        const float m_old = a.mean[3 * (size_t)i + ch];
        const float m_new = m_old + coeff * (full[ch] - m_old);
        a.mean[3 * (size_t)i + ch] = m_new;
        a.variance[3 * (size_t)i + ch] += w * (full[ch] - m_new) * (full[ch] - m_new);
    }
}

void launch_colour_variance_accumulate(int P, const int* D, int M, int max_sh_deg, const float* means3D,
                                       const float* cam_pos, const float* shs, const int* radii, const int* touched,
                                       const float* transmittance, float* wSum, float* wSumSq, float* mean,
                                       float* variance, float* accum, hipStream_t s)
{
    CvArgs a;
    a.P = P;
    a.M = M;
    a.max_deg = max_sh_deg;
    a.degs = D;
    a.means = means3D;
    a.campos = cam_pos;
    a.shs = shs;
    a.radii = radii;
    a.touched = touched;
    a.transm = transmittance;
    a.wSum = wSum;
    a.wSumSq = wSumSq;
    a.mean = mean;
    a.variance = variance;
    a.accum = accum;
    hipLaunchKernelGGL(colour_variance_accumulate_kernel, dim3((P + kCvBlock - 1) / kCvBlock), dim3(kCvBlock), 0, s, a);
}

}  // namespace r3
