// reduction_ops.hip -- the redundancy-score operators around the rasterizer, declared in
// include/r3dgs_reduction.h.  What the reference does (/root/reference/submodules/diff-gaussian-rasterization):
// reduced_3dgs.cu:205-287 on the host, reduced_3dgs/redundancy_score.cu on the device.  How it is issued here:
//   * pixel size: ONE launch, the camera loop runs inside the kernel with the matrices and image sizes read
//     through wave-uniform loads (the reference launches per camera and does two .item() host syncs per camera);
//   * intersection / scatter-min: one lane per (Gaussian, neighbour) PAIR, so the index rows and the mask bytes
//     are read/written fully coalesced; the per-Gaussian count is a ballot+popcount per wave segment.
// Compiled with -ffp-contract=off and correctly rounded divide/sqrt (build.py EXACT): operation order follows
// the scalar reference (GLM evaluation order for the matrix products).
#include "../../include/r3dgs_reduction.h"

#include "common.h"

namespace {

constexpr int kBlock = 256;

// GLM mat4 * vec4 on the raw floats: (col0*x + col1*y) + (col2*z + col3*w); m[4c+r].
struct V4 {
    float x, y, z, w;
};
__device__ inline V4 mat4_mul(const float* __restrict__ m, float x, float y, float z, float w)
{
    V4 r;
    r.x = (m[0] * x + m[4] * y) + (m[8] * z + m[12] * w);
    r.y = (m[1] * x + m[5] * y) + (m[9] * z + m[13] * w);
    r.z = (m[2] * x + m[6] * y) + (m[10] * z + m[14] * w);
    r.w = (m[3] * x + m[7] * y) + (m[11] * z + m[15] * w);
    return r;
}

// redundancy_score.cu:46-97, all cameras in one pass.
__global__ __launch_bounds__(kBlock) void min_pixel_size_kernel(int P, int C, const float* __restrict__ w2ndc,
                                                                const float* __restrict__ w2ndc_inv,
                                                                const float* __restrict__ means3D,
                                                                const int* __restrict__ image_height,
                                                                const int* __restrict__ image_width,
                                                                float* __restrict__ pixel_sizes)
{
    const int idx = blockIdx.x * kBlock + threadIdx.x;
    if (idx >= P) return;
    const float px = means3D[3 * idx], py = means3D[3 * idx + 1], pz = means3D[3 * idx + 2];
    float best = 10000.f;
    for (int c = 0; c < C; c++) {
        const float* m = w2ndc + 16 * c;        // wave-uniform addresses -> scalar loads
        const float* mi = w2ndc_inv + 16 * c;
        const V4 hom = mat4_mul(m, px, py, pz, 1.f);
        float pw = 1.0f / (hom.w + 0.0000001f);
        const float nx = hom.x * pw, ny = hom.y * pw, depth = hom.z * pw;
        const bool inside = nx <= 1.f && ny <= 1.f && depth <= 1.f && nx >= -1.f && ny >= -1.f && depth >= 0.f;
        if (!inside) continue;
        const int W = image_width[c], H = image_height[c];
        float ex = 0.f, ey = 0.f;
        if (W > H)
            ex = 2.f / (float)W;
        else
            ey = 2.f / (float)H;
        const V4 e = mat4_mul(mi, ex, ey, depth, 1.f);
        pw = 1.f / (e.w + 0.0000001f);
        const float e0 = e.x * pw, e1 = e.y * pw, e2 = e.z * pw;
        const V4 s = mat4_mul(mi, 0.f, 0.f, depth, 1.f);
        pw = 1.f / (s.w + 0.0000001f);
        const float d0 = e0 - s.x * pw, d1 = e1 - s.y * pw, d2 = e2 - s.z * pw;
        const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        best = fminf(best, len);  // CUDA min(float,float): a NaN operand yields the other one
    }
    pixel_sizes[idx] = best;
}

// redundancy_score.cu:121-159 with the rotation of :185-207 rebuilt in registers (the reference materialises a
// [P,3,3] tensor first).  One lane per pair; pairs of a Gaussian are contiguous, so a wave holds whole runs.
__global__ __launch_bounds__(kBlock) void intersection_kernel(int P, int knn, const float* __restrict__ means3D,
                                                              const float* __restrict__ scales,
                                                              const float* __restrict__ rotations,
                                                              const int* __restrict__ neighbours,
                                                              const float* __restrict__ sphere_radius,
                                                              int* __restrict__ redundancy,
                                                              uint8_t* __restrict__ mask)
{
    const long long total = (long long)P * knn;
    const long long pair = (long long)blockIdx.x * kBlock + threadIdx.x;
    const bool live = pair < total;
    bool hit = false;
    int g = 0;
    if (live) {
        g = (int)(pair / knn);
        const int n = neighbours[pair];
        const float r = rotations[4 * g], x = rotations[4 * g + 1], y = rotations[4 * g + 2], z = rotations[4 * g + 3];
        const float d0 = means3D[3 * g] - means3D[3 * n];
        const float d1 = means3D[3 * g + 1] - means3D[3 * n + 1];
        const float d2 = means3D[3 * g + 2] - means3D[3 * n + 2];
        const float rad = sphere_radius[g];
        const float a0 = scales[3 * n] + rad, a1 = scales[3 * n + 1] + rad, a2 = scales[3 * n + 2] + rad;
        // vec3 * mat3 = (dot(col0, v), dot(col1, v), dot(col2, v)), dot = (x*x' + y*y') + z*z'
        const float l0 = ((1.f - 2.f * (y * y + z * z)) * d0 + (2.f * (x * y + r * z)) * d1) + (2.f * (x * z - r * y)) * d2;
        const float l1 = ((2.f * (x * y - r * z)) * d0 + (1.f - 2.f * (x * x + z * z)) * d1) + (2.f * (y * z + r * x)) * d2;
        const float l2 = ((2.f * (x * z + r * y)) * d0 + (2.f * (y * z - r * x)) * d1) + (1.f - 2.f * (x * x + y * y)) * d2;
        const float v = ((l0 * l0) * (1.f / (a0 * a0)) + (l1 * l1) * (1.f / (a1 * a1))) + (l2 * l2) * (1.f / (a2 * a2));
        hit = v < 1.f;
        mask[pair] = hit ? 1 : 0;
    }
    // count per Gaussian: lanes of one Gaussian form a contiguous segment of the wave
    const unsigned long long hits = __ballot(hit);
    if (live) {
        const long long wave_base = pair - (threadIdx.x & 63);
        const long long seg_lo = (long long)g * knn - wave_base, seg_hi = seg_lo + knn;
        const int lo = seg_lo < 0 ? 0 : (int)seg_lo, hi = seg_hi > 64 ? 64 : (int)seg_hi;
        if ((int)(threadIdx.x & 63) == lo) {
            const unsigned long long seg = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
            const int c = __popcll(hits & seg);
            if (lo == (int)seg_lo && hi == (int)seg_hi)
                redundancy[g] = c;            // the whole run is in this wave: plain store, no zero-fill needed
            else if (c)
                atomicAdd(&redundancy[g], c);  // run cut by a wave boundary (redundancy zeroed by the launcher)
        }
    }
}

// runs cut by a wave boundary accumulate with atomics, so those entries must start at zero
__global__ __launch_bounds__(kBlock) void zero_cut_runs_kernel(int P, int knn, int* __restrict__ redundancy)
{
    const int g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= P) return;
    const long long a = (long long)g * knn, b = a + knn - 1;
    if ((a >> 6) != (b >> 6)) redundancy[g] = 0;
}

// redundancy_score.cu:6-27, one lane per pair.
__global__ __launch_bounds__(kBlock) void fill_int_kernel(int n, int v, int* __restrict__ out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = v;
}
__global__ __launch_bounds__(kBlock) void min_redundancy_kernel(long long total, int knn,
                                                                const int* __restrict__ redundancy,
                                                                const int* __restrict__ neighbours,
                                                                const uint8_t* __restrict__ mask,
                                                                int* __restrict__ min_redundancy)
{
    const long long pair = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (pair >= total || !mask[pair]) return;
    atomicMin(&min_redundancy[neighbours[pair]], redundancy[pair / knn]);
}

// densification statistics of one view, for the view-parallel exchange (multiview.py)
__global__ __launch_bounds__(kBlock) void pack_view_stats_kernel(int P, const float* __restrict__ vg,
                                                                 const int* __restrict__ radii,
                                                                 float* __restrict__ grad_norm,
                                                                 float* __restrict__ visible, int* __restrict__ radii_out)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= P) return;
    const int r = radii[i];
    const float gx = vg[3 * i], gy = vg[3 * i + 1];
    grad_norm[i] = r > 0 ? sqrtf(gx * gx + gy * gy) : 0.f;
    visible[i] = r > 0 ? 1.f : 0.f;
    radii_out[i] = r;
}

// Local half of the view-parallel exchange (multiview.py): after the all-to-all every rank holds, for ITS shard of the
// flat buffer, one copy per rank; combine them in rank order -- SUM for the fp32 part of the buffer (gradients,
// densification statistics), MAX for the int32 tail (radii).  Fixed order: every replica computes identical bits.
__global__ __launch_bounds__(kBlock) void reduce_shards_kernel(const float* __restrict__ recv, int world, long long shard,
                                                               long long shard_begin, long long sum_len,
                                                               float* __restrict__ out)
{
    const long long j = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (j >= shard) return;
    if (shard_begin + j < sum_len) {
        float acc = recv[j];
        for (int w = 1; w < world; w++) acc += recv[(long long)w * shard + j];
        out[j] = acc;
    } else {
        const int* ri = reinterpret_cast<const int*>(recv);
        int m = ri[j];
        for (int w = 1; w < world; w++) m = max(m, ri[(long long)w * shard + j]);
        reinterpret_cast<int*>(out)[j] = m;
    }
}

// The same with a middle region of bfloat16 PAIRS (the opt-in reduced-precision transport of the 45 higher-band SH
// gradients per Gaussian, multiview.py sh_rest_bf16): words [sum_len, half_end) of the buffer hold two bfloat16 each;
// every half is widened, summed in fp32 in rank order and rounded once to nearest-even -- the value every replica,
// the owner of the shard included, continues with.
__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t f32_to_bf16_rne(float f)
{
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;   // NaN stays NaN (quiet)
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__global__ __launch_bounds__(kBlock) void reduce_shards_mixed_kernel(const float* __restrict__ recv, int world, long long shard,
                                                                     long long shard_begin, long long sum_len,
                                                                     long long half_end, float* __restrict__ out)
{
    const long long j = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (j >= shard) return;
    const long long e = shard_begin + j;
    if (e < sum_len) {
        float acc = recv[j];
        for (int w = 1; w < world; w++) acc += recv[(long long)w * shard + j];
        out[j] = acc;
    } else if (e < half_end) {
        const uint32_t* ru = reinterpret_cast<const uint32_t*>(recv);
        uint32_t v = ru[j];
        float lo = bf16_to_f32(v & 0xFFFFu), hi = bf16_to_f32(v >> 16);
        for (int w = 1; w < world; w++) {
            v = ru[(long long)w * shard + j];
            lo += bf16_to_f32(v & 0xFFFFu);
            hi += bf16_to_f32(v >> 16);
        }
        reinterpret_cast<uint32_t*>(out)[j] = f32_to_bf16_rne(lo) | (f32_to_bf16_rne(hi) << 16);
    } else {
        const int* ri = reinterpret_cast<const int*>(recv);
        int m = ri[j];
        for (int w = 1; w < world; w++) m = max(m, ri[(long long)w * shard + j]);
        reinterpret_cast<int*>(out)[j] = m;
    }
}

int grid_for(long long n) { return (int)((n + kBlock - 1) / kBlock); }

}  // namespace

extern "C" {

int r3dgs_min_pixel_size(int P, int n_cameras, const float* w2ndc, const float* w2ndc_inv, const float* means3D,
                         const int* image_height, const int* image_width, float* pixel_sizes, void* stream)
{
    return r3::guarded_call([&]() {
        if (P <= 0) return 0;
        if (n_cameras < 0) throw r3::Error("n_cameras must be >= 0");
        if (!means3D || !pixel_sizes || (n_cameras && (!w2ndc || !w2ndc_inv || !image_height || !image_width)))
            throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        min_pixel_size_kernel<<<grid_for(P), kBlock, 0, s>>>(P, n_cameras, w2ndc, w2ndc_inv, means3D, image_height,
                                                             image_width, pixel_sizes);
        r3::check_launch("min pixel size", s, false);
        return 0;
    });
}

int r3dgs_sphere_ellipsoid_intersection(int P, int knn, const float* means3D, const float* scales,
                                        const float* rotations, const int* neighbours, const float* sphere_radius,
                                        int* redundancy, uint8_t* mask, void* stream)
{
    return r3::guarded_call([&]() {
        if (P <= 0) return 0;
        if (knn < 0) throw r3::Error("knn must be >= 0");
        if (!redundancy) throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (knn == 0) {
            fill_int_kernel<<<grid_for(P), kBlock, 0, s>>>(P, 0, redundancy);
            r3::check_launch("intersection (no neighbours)", s, false);
            return 0;
        }
        if (!means3D || !scales || !rotations || !neighbours || !sphere_radius || !mask)
            throw r3::Error("a required pointer is NULL");
        zero_cut_runs_kernel<<<grid_for(P), kBlock, 0, s>>>(P, knn, redundancy);
        intersection_kernel<<<grid_for((long long)P * knn), kBlock, 0, s>>>(P, knn, means3D, scales, rotations,
                                                                              neighbours, sphere_radius, redundancy,
                                                                              mask);
        r3::check_launch("sphere/ellipsoid intersection", s, false);
        return 0;
    });
}

int r3dgs_min_redundancy(int P, int knn, const int* redundancy, const int* neighbours, const uint8_t* mask,
                         int* min_redundancy, void* stream)
{
    return r3::guarded_call([&]() {
        if (P <= 0) return 0;
        if (knn < 0) throw r3::Error("knn must be >= 0");
        if (!min_redundancy || (knn && (!redundancy || !neighbours || !mask)))
            throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        fill_int_kernel<<<grid_for(P), kBlock, 0, s>>>(P, P, min_redundancy);
        if (knn)
            min_redundancy_kernel<<<grid_for((long long)P * knn), kBlock, 0, s>>>((long long)P * knn, knn, redundancy,
                                                                                    neighbours, mask, min_redundancy);
        r3::check_launch("min redundancy", s, false);
        return 0;
    });
}

int r3dgs_pack_view_stats(int P, const float* viewspace_grad, const int* radii, float* grad_norm, float* visible,
                          int* radii_out, void* stream)
{
    return r3::guarded_call([&]() {
        if (P <= 0) return 0;
        if (!viewspace_grad || !radii || !grad_norm || !visible || !radii_out) throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        pack_view_stats_kernel<<<grid_for(P), kBlock, 0, s>>>(P, viewspace_grad, radii, grad_norm, visible, radii_out);
        r3::check_launch("pack view stats", s, false);
        return 0;
    });
}

int r3dgs_reduce_shards(int world, long long shard, long long shard_begin, long long sum_len, const float* recv,
                        float* out, void* stream)
{
    return r3::guarded_call([&]() {
        if (shard <= 0) return 0;
        if (world < 1 || !recv || !out) throw r3::Error("reduce_shards: bad arguments");
        hipStream_t s = static_cast<hipStream_t>(stream);
        reduce_shards_kernel<<<grid_for(shard), kBlock, 0, s>>>(recv, world, shard, shard_begin, sum_len, out);
        r3::check_launch("reduce shards", s, false);
        return 0;
    });
}

int r3dgs_reduce_shards_mixed(int world, long long shard, long long shard_begin, long long sum_len, long long half_end,
                              const float* recv, float* out, void* stream)
{
    return r3::guarded_call([&]() {
        if (shard <= 0) return 0;
        if (world < 1 || !recv || !out || half_end < sum_len) throw r3::Error("reduce_shards_mixed: bad arguments");
        hipStream_t s = static_cast<hipStream_t>(stream);
        reduce_shards_mixed_kernel<<<grid_for(shard), kBlock, 0, s>>>(recv, world, shard, shard_begin, sum_len, half_end, out);
        r3::check_launch("reduce shards (mixed)", s, false);
        return 0;
    });
}

}  // extern "C"
