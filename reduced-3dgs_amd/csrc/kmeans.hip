// kmeans.hip -- 1-D k-means codebooks (r3dgs_kmeans, include/r3dgs_reduction.h); replaces Reduced3DGS::kmeans,
// /root/reference/submodules/diff-gaussian-rasterization/reduced_3dgs.cu:290-340 with its kernels
// reduced_3dgs/kmeans.cu:12-53 (updateCentersCUDA) and :73-105 (updateIdsCUDA).
//
// What the reference computes per Lloyd update: id(v) = FIRST centre i minimising sqrt((c_i - v)^2) (strict <, so
// the lowest index wins among equal distances), centre_i = sum of its values / their number (0 if empty), stop when
// sum |old - new| < tol; then one more assignment.  It does so with a 256-way linear scan per value and per update
// and a host read-back of the shift per update.
//
// Here the problem's 1-D structure is used instead (results identical, see the exactness note):
//   * the values are radix-sorted ONCE and an fp64 exclusive prefix sum of the sorted values is kept;
//   * per update the centres are sorted (value, index) and deduplicated in one block; each sorted value finds its
//     centre by an 8-10 step binary search in LDS, and only the few hundred positions where the id changes between
//     consecutive sorted values touch the accumulators: cluster sum = difference of two prefix entries, cluster size =
//     difference of two positions.  One pass over 4 bytes per value per update, no per-value atomics;
//   * the convergence flag stays on the device; all updates are enqueued at once and become no-ops once converged;
//   * the final assignment runs over the values in their original order.
// Exactness: fl(c - v), its square and the correctly rounded sqrt are monotone in c on either side of v, so the
// minimum distance is attained at one of the two sorted neighbours of v and every centre that TIES with it (after
// rounding) is contiguous with them; scanning outwards while the distance is equal and taking the smallest original
// index reproduces the reference's first-index rule bit for bit, including duplicate and near-duplicate centres.
// Sums: fp64 prefix differences rounded once to fp32 (the reference adds fp32 partials in atomic order, so its
// centres are only defined to ~1e-6 relative; the oracle sums in fp64 too).  NaN VALUES are not supported (the
// reference turns centre 0 into 0 when one is present); NaN / inf centres behave as in the reference.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "../../include/r3dgs_reduction.h"
#include "common.h"

namespace {

constexpr int kMaxCenters = 1024;
constexpr int kBlock = 256;
constexpr int kItems = 8;           // consecutive sorted values per thread in the segment pass
constexpr int kPad = 0x7fffffff;

struct ToDouble {
    __host__ __device__ double operator()(float v) const { return (double)v; }
};

struct KmeansWork {
    float* centers;      // [1024] current centres, original order
    float* s_val;        // [1024] sorted, deduplicated centre values
    int* s_idx;          // [1024] smallest original index of each
    double* sums;        // [1024]
    int* counts;         // [1024] (wrapping adds of +-position)
    int* flags;          // m, done, iters
    float* sorted;       // [n]
    double* prefix;      // [n + 1] exclusive prefix of sorted
    char* temp;
    static KmeansWork carve(char* base, size_t n, size_t temp_bytes)
    {
        KmeansWork w;
        char* p = base;
        auto take = [&](size_t bytes) {
            char* r = p;
            p += (bytes + 255) / 256 * 256;
            return r;
        };
        w.centers = reinterpret_cast<float*>(take(4 * kMaxCenters));
        w.s_val = reinterpret_cast<float*>(take(4 * kMaxCenters));
        w.s_idx = reinterpret_cast<int*>(take(4 * kMaxCenters));
        w.sums = reinterpret_cast<double*>(take(8 * kMaxCenters));
        w.counts = reinterpret_cast<int*>(take(4 * kMaxCenters));
        w.flags = reinterpret_cast<int*>(take(256));
        w.sorted = reinterpret_cast<float*>(take(4 * n));
        w.prefix = reinterpret_cast<double*>(take(8 * (n + 1)));
        w.temp = take(temp_bytes);
        (void)p;
        return w;
    }
};

size_t temp_bytes_for(size_t n)
{
    if (n == 0) return 0;
    size_t a = 0, b = 0;
    R3_HIP(rocprim::radix_sort_keys(nullptr, a, (float*)nullptr, (float*)nullptr, n, 0, 32, (hipStream_t)0));
    auto in = rocprim::make_transform_iterator((const float*)nullptr, ToDouble{});
    R3_HIP(rocprim::inclusive_scan(nullptr, b, in, (double*)nullptr, n, rocprim::plus<double>(), (hipStream_t)0));
    return a > b ? a : b;
}

// ---- exact assignment against the sorted, deduplicated centres (LDS or global pointers) -------------------------
__device__ inline float centre_dist(float c, float v)
{
    const float t = c - v;
    return sqrtf(t * t);
}

// Squares that differ by more than kSure relative cannot round to the same correctly rounded sqrt (sqrt halves a
// relative gap; one fp32 ulp is 2^-23), so the common case is decided on the squares and the sqrt is only formed for
// near-ties.
constexpr float kSure = 1.00001f;

__device__ inline float centre_sq(float c, float v)
{
    const float t = c - v;
    return t * t;
}

__device__ inline int assign_sorted(float v, const float* __restrict__ sv, const int* __restrict__ si, int m)
{
    int lo = 0, hi = m;                 // first entry with sv >= v (NaN entries sort last and compare as ">= v")
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sv[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    const int left = lo - 1, right = lo;
    const float al = left >= 0 ? centre_sq(sv[left], v) : INFINITY;
    const float ar = right < m ? centre_sq(sv[right], v) : INFINITY;
    if (al * kSure < ar) {              // left neighbour strictly nearer; ties can only continue leftwards
        int id = si[left];
        for (int j = left - 1; j >= 0; j--) {
            const float a = centre_sq(sv[j], v);
            if (al * kSure < a || sqrtf(a) != sqrtf(al)) break;
            id = min(id, si[j]);
        }
        return id;
    }
    if (ar * kSure < al) {
        int id = si[right];
        for (int j = right + 1; j < m; j++) {
            const float a = centre_sq(sv[j], v);
            if (ar * kSure < a || sqrtf(a) != sqrtf(ar)) break;
            id = min(id, si[j]);
        }
        return id;
    }
    // near-tie between the two sides (or nothing finite): the exact rule
    const float dl = sqrtf(al), dr = sqrtf(ar);
    const float mind = dl < dr ? dl : dr;
    if (!(mind < INFINITY)) return 0;   // nothing ever beats the initial min_dist = INFINITY: closest_centroid stays 0
    int id = kPad;
    for (int j = left; j >= 0 && centre_dist(sv[j], v) == mind; j--) id = min(id, si[j]);
    for (int j = right; j < m && centre_dist(sv[j], v) == mind; j++) id = min(id, si[j]);
    return id;
}

// ---- one block: (optionally) finish an update, then sort + deduplicate the centres for the next pass -----------
__device__ inline bool key_less(float av, int ai, float bv, int bi)
{
    const int ca = ai == kPad ? 2 : (av != av ? 1 : 0), cb = bi == kPad ? 2 : (bv != bv ? 1 : 0);
    if (ca != cb) return ca < cb;
    if (ca == 0 && av != bv) return av < bv;
    return ai < bi;
}

template <bool UPDATE>
__global__ __launch_bounds__(kMaxCenters) void kmeans_centres_kernel(int n_centers, float* __restrict__ centers,
                                                                     double* __restrict__ sums,
                                                                     int* __restrict__ counts, float tol,
                                                                     float* __restrict__ s_val_out,
                                                                     int* __restrict__ s_idx_out,
                                                                     int* __restrict__ flags)
{
    int* m_out = flags;
    int* done = flags + 1;
    int* iters = flags + 2;
    if (UPDATE && *done) return;
    __shared__ float k_val[kMaxCenters];
    __shared__ int k_idx[kMaxCenters];
    __shared__ int s_scan[kMaxCenters];
    __shared__ float s_shift[kMaxCenters / 64];
    const int t = threadIdx.x;
    const int N = blockDim.x;           // next power of two >= n_centers (>= 64)

    float mine = 0.f;
    if (t < n_centers) mine = centers[t];
    if (UPDATE) {
        // reduced_3dgs.cu:318-324: centre = sum / size, NaN (empty cluster) -> 0, shift = sum |old - new|
        float shift = 0.f;
        if (t < n_centers) {
            float nc = (float)sums[t] / (float)counts[t];
            if (nc != nc) nc = 0.f;
            shift = fabsf(mine - nc);
            mine = nc;
            centers[t] = nc;
        }
        for (int off = 32; off; off >>= 1) shift += __shfl_down(shift, off);
        if ((t & 63) == 0) s_shift[t >> 6] = shift;
        __syncthreads();
        if (t == 0) {
            float tot = 0.f;
            for (int w = 0; w < N / 64; w++) tot += s_shift[w];
            *iters += 1;
            if (tot < tol) *done = 1;
        }
    }
    if (t < n_centers) {
        sums[t] = 0.0;
        counts[t] = 0;
    }

    // bitonic sort of (value, index) over blockDim.x = next power of two >= n_centers entries,
    // NaNs after the numbers, padding last
    k_val[t] = mine;
    k_idx[t] = t < n_centers ? t : kPad;
    __syncthreads();
    for (int k = 2; k <= N; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int o = t ^ j;
            if (o > t) {
                const float av = k_val[t], bv = k_val[o];
                const int ai = k_idx[t], bi = k_idx[o];
                const bool up = (t & k) == 0;
                if (key_less(bv, bi, av, ai) == up) {
                    k_val[t] = bv;
                    k_idx[t] = bi;
                    k_val[o] = av;
                    k_idx[o] = ai;
                }
            }
            __syncthreads();
        }
    // deduplicate equal values (the first of a run carries the smallest index) and compact
    const bool keep = t < n_centers && (t == 0 || !(k_val[t] == k_val[t - 1]));
    s_scan[t] = keep ? 1 : 0;
    __syncthreads();
    for (int off = 1; off < N; off <<= 1) {
        const int add = t >= off ? s_scan[t - off] : 0;
        __syncthreads();
        s_scan[t] += add;
        __syncthreads();
    }
    if (keep) {
        s_val_out[s_scan[t] - 1] = k_val[t];
        s_idx_out[s_scan[t] - 1] = k_idx[t];
    }
    if (t == N - 1) *m_out = s_scan[t];
}

// ---- the per-update pass over the sorted values: only id changes touch the accumulators ------------------------
__global__ __launch_bounds__(kBlock) void kmeans_segments_kernel(int n, const float* __restrict__ sorted,
                                                                 const double* __restrict__ prefix,
                                                                 const float* __restrict__ s_val,
                                                                 const int* __restrict__ s_idx,
                                                                 const int* __restrict__ flags,
                                                                 double* __restrict__ sums, int* __restrict__ counts)
{
    if (flags[1]) return;               // converged
    __shared__ float sv[kMaxCenters];
    __shared__ int si[kMaxCenters];
    const int m = flags[0];
    for (int i = threadIdx.x; i < m; i += kBlock) {
        sv[i] = s_val[i];
        si[i] = s_idx[i];
    }
    __syncthreads();
    const long long start = ((long long)blockIdx.x * kBlock + threadIdx.x) * kItems;
    if (start >= n) return;
    float v[kItems];
    if (start + kItems <= n) {
        const float4 a = *reinterpret_cast<const float4*>(sorted + start);
        const float4 b = *reinterpret_cast<const float4*>(sorted + start + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
        for (int k = 0; k < kItems; k++) v[k] = start + k < n ? sorted[start + k] : 0.f;
    }
    int prev = start > 0 ? assign_sorted(sorted[start - 1], sv, si, m) : -1;
    for (int k = 0; k < kItems; k++) {
        const long long p = start + k;
        if (p >= n) break;
        const int id = assign_sorted(v[k], sv, si, m);
        if (id != prev) {               // a run of `prev` ends before p, a run of `id` starts at p
            const double pre = prefix[p];
            if (prev >= 0) {
                atomicAdd(&sums[prev], pre);
                atomicAdd(&counts[prev], (int)p);
            }
            atomicAdd(&sums[id], -pre);
            atomicAdd(&counts[id], -(int)p);
            prev = id;
        }
        if (p == n - 1) {
            atomicAdd(&sums[id], prefix[n]);
            atomicAdd(&counts[id], n);
        }
    }
}

__global__ __launch_bounds__(kBlock) void kmeans_assign_kernel(int n, const float* __restrict__ values,
                                                               const float* __restrict__ s_val,
                                                               const int* __restrict__ s_idx,
                                                               const int* __restrict__ flags, int* __restrict__ ids)
{
    __shared__ float sv[kMaxCenters];
    __shared__ int si[kMaxCenters];
    const int m = flags[0];
    for (int i = threadIdx.x; i < m; i += kBlock) {
        sv[i] = s_val[i];
        si[i] = s_idx[i];
    }
    __syncthreads();
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        ids[i] = assign_sorted(values[i], sv, si, m);
}

__global__ void kmeans_finish_kernel(int n_centers, const float* __restrict__ centers, float* __restrict__ out,
                                     const int* __restrict__ flags, int* __restrict__ iters_out)
{
    for (int i = threadIdx.x; i < n_centers; i += blockDim.x) out[i] = centers[i];
    if (threadIdx.x == 0 && iters_out) *iters_out = flags[2];
}

__global__ void zero_first_kernel(double* p) { *p = 0.0; }

}  // namespace

extern "C" {

size_t r3dgs_kmeans_workspace_bytes(int n_values, int n_centers)
{
    size_t out = 0;
    r3::guarded_call([&]() {
        if (n_centers < 1 || n_centers > kMaxCenters) throw r3::Error("n_centers must be in [1,1024]");
        if (n_values < 0) throw r3::Error("negative size");
        const size_t temp = temp_bytes_for((size_t)n_values);
        KmeansWork w = KmeansWork::carve(nullptr, (size_t)n_values, temp);
        out = (size_t)reinterpret_cast<uintptr_t>(w.temp) + temp + 256;
        return 0;
    });
    return out;
}

int r3dgs_kmeans(int n_values, int n_centers, const float* values, const float* centers_in, float tol,
                 int max_iterations, int* ids, float* centers_out, int* iterations_run, char* workspace, void* stream)
{
    return r3::guarded_call([&]() {
        if (n_centers < 1 || n_centers > kMaxCenters) throw r3::Error("n_centers must be in [1,1024]");
        if (n_values < 0 || max_iterations < 0) throw r3::Error("negative size");
        if (!centers_in || !centers_out || !workspace || (n_values && (!values || !ids)))
            throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        const size_t n = (size_t)n_values;
        size_t temp = temp_bytes_for(n);
        KmeansWork w = KmeansWork::carve(workspace, n, temp);
        R3_HIP(hipMemcpyAsync(w.centers, centers_in, sizeof(float) * n_centers, hipMemcpyDeviceToDevice, s));
        R3_HIP(hipMemsetAsync(w.flags, 0, 256, s));
        if (n && max_iterations) {
            size_t t1 = temp;
            R3_HIP(rocprim::radix_sort_keys(w.temp, t1, values, w.sorted, n, 0, 32, s));
            zero_first_kernel<<<1, 1, 0, s>>>(w.prefix);
            size_t t2 = temp;
            auto in = rocprim::make_transform_iterator((const float*)w.sorted, ToDouble{});
            R3_HIP(rocprim::inclusive_scan(w.temp, t2, in, w.prefix + 1, n, rocprim::plus<double>(), s));
        }
        int sort_n = 64;
        while (sort_n < n_centers) sort_n <<= 1;
        kmeans_centres_kernel<false><<<1, sort_n, 0, s>>>(n_centers, w.centers, w.sums, w.counts, tol, w.s_val,
                                                               w.s_idx, w.flags);
        const long long per_block = (long long)kBlock * kItems;
        const int seg_blocks = (int)((n + per_block - 1) / per_block);
        for (int it = 0; it < max_iterations; it++) {
            if (seg_blocks)
                kmeans_segments_kernel<<<seg_blocks, kBlock, 0, s>>>(n_values, w.sorted, w.prefix, w.s_val, w.s_idx,
                                                                     w.flags, w.sums, w.counts);
            kmeans_centres_kernel<true><<<1, sort_n, 0, s>>>(n_centers, w.centers, w.sums, w.counts, tol, w.s_val,
                                                                  w.s_idx, w.flags);
        }
        if (n) {
            int blocks = (int)((n + kBlock - 1) / kBlock);
            blocks = blocks > 8192 ? 8192 : blocks;
            kmeans_assign_kernel<<<blocks, kBlock, 0, s>>>(n_values, values, w.s_val, w.s_idx, w.flags, ids);
        }
        kmeans_finish_kernel<<<1, 256, 0, s>>>(n_centers, w.centers, centers_out, w.flags, iterations_run);
        r3::check_launch("kmeans", s, false);
        return 0;
    });
}

}  // extern "C"
