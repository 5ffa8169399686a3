// binning.hip -- tile binning: which Gaussians touch which 16x16 tile, in front-to-back order.
//
// Replaces rasterizer_impl.cu:441 (InclusiveSum), :78-119 duplicateWithKeys, :465-473 the 64-bit
// (tile|depth) radix sort and :124-146 identifyTileRanges of
// /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer, with the SAME resulting order
// (tile-major; inside a tile ascending depth bits, ties by ascending Gaussian index) but far less sort
// traffic, which is what bounds this stage on MI355X:
//
//   reference: sort R (u64 key, u32 value) pairs on 32+log2(tiles) bits  -> ~6 passes x 24 B x R
//   here:      1. stable sort the P Gaussians once by their 32 depth bits (4 passes x 16 B x P, P << R)
//              2. inclusive scan of tiles_touched in that depth order
//              3. emit the R (tile, id) pairs in depth order, load-balanced per wave
//              4. stable sort the pairs on the log2(tiles) tile bits only (2 passes x 16 B x R)
//   Stability of both sorts + ascending-id input order reproduces the reference's tie-break exactly.
//
// The sorts / scan use rocPRIM device primitives (onesweep radix sort, decoupled-lookback scan).
#include <cstdlib>
#include <cstring>
#include <string>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace r3 {

struct GatherTiles {
    const uint32_t* tiles;
    __host__ __device__ uint32_t operator()(uint32_t id) const { return tiles[id]; }
};

size_t depth_sort_temp_bytes(size_t P)
{
    size_t a = 0, b = 0;
    uint32_t* n = nullptr;
    R3_HIP(rocprim::radix_sort_pairs(nullptr, a, n, n, rocprim::counting_iterator<uint32_t>(0), n, P, 0, 32));
    auto it = rocprim::make_transform_iterator(n, GatherTiles{n});
    R3_HIP(rocprim::inclusive_scan(nullptr, b, it, n, P, rocprim::plus<uint32_t>()));
    return (a > b ? a : b) + 256;
}

size_t tile_sort_temp_bytes(size_t R)
{
    size_t a = 0;
    uint32_t* n = nullptr;
    R3_HIP(rocprim::radix_sort_pairs(nullptr, a, n, n, n, n, R ? R : 1, 0, 32));
    return a + 256;
}

void run_depth_sort_and_scan(int P, GeomState& g, hipStream_t s)
{
    size_t bytes = g.temp_bytes;
    R3_HIP(rocprim::radix_sort_pairs(g.temp, bytes, g.depth_key, g.key_sorted, rocprim::counting_iterator<uint32_t>(0),
                                     g.order, (size_t)P, 0, 32, s));
    bytes = g.temp_bytes;
    auto it = rocprim::make_transform_iterator(g.order, GatherTiles{g.tiles});
    R3_HIP(rocprim::inclusive_scan(g.temp, bytes, it, g.offsets, (size_t)P, rocprim::plus<uint32_t>(), s));
}

// Inclusive scan across a 256-thread workgroup: shuffles inside each wave, one LDS exchange of the four wave totals
// (two barriers instead of the sixteen of a Hillis-Steele pass in LDS).  s_w: 4 words of LDS.
template <class T>
__device__ inline T block256_inclusive_scan(T v, T* s_w)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        const T up = __shfl_up(v, off);
        if (lane >= off) v += up;
    }
    __syncthreads();   // s_w may still be read from a previous use
    if (lane == 63) s_w[w] = v;
    __syncthreads();
    T add = 0;
    for (int k = 0; k < w; k++) add += s_w[k];
    return v + add;
}

// ---- bucketed depth sort ---------------------------------------------------------------------------------------
// The generic device sort above is launch-latency bound at this size (block sort + 9 merge passes, ~125 us for
// 500k keys).  The keys are view depths, so:
//   (1) every workgroup histograms its 4096 Gaussians over kDepthBuckets equal-width intervals of [min depth, max
//       depth] (the range comes from the preprocess kernel) and stores its row -- no global atomics: device-scope
//       atomics on the same few lines were the cost of the first version of this pass;
//   (2) a column-wise scan of the rows (each workgroup's first slot inside each bucket) and the bucket totals;
//       raises header.sort_overflow if a bucket exceeds kBucketCap;
//   (3) (key, id) scatter into the bucket regions -- slot = bucket start (scan of the totals, redone per workgroup
//       in LDS) + row base + LDS rank, again no global atomics;
//   (4) one workgroup per bucket sorts its pairs in LDS as 64-bit (key << 32 | id) words and scans tiles_touched in
//       that order on top of the bucket's base.
// The bucket function is monotone in the key, so concatenating the sorted buckets is the stable sort by depth bits
// the reference's 64-bit key sort implies.  On overflow (e.g. a fronto-parallel plane of splats) the host, which
// sees the flag with the num_rendered read-back, reruns the generic path.
constexpr int kHistPerThread = kHistPerBlock / 256;
constexpr int kCountBits = 24;
constexpr unsigned long long kCountMask = (1ull << kCountBits) - 1;

struct DepthRange {
    float zmin, scale;
};
__device__ inline DepthRange make_depth_range(uint32_t mx, uint32_t mi)
{
    DepthRange r;
    const float zmax = __uint_as_float(mx), zmin = __uint_as_float(~mi);
    r.zmin = zmin;
    r.scale = zmax > zmin ? (float)kDepthBuckets / (zmax - zmin) : 0.f;   // no visible Gaussian: nothing is looked up
    return r;
}
__device__ inline DepthRange load_depth_range(const DepthSortScratch* ds)
{
    return make_depth_range(ds->depth_max, ds->depth_inv_min);
}
// Sum / max of the preprocess workgroups' partials by one workgroup of 256 or 1024 threads (s_red: 4 x 16 words).
__device__ inline PrePartial reduce_partials(const PrePartial* __restrict__ parts, int n, uint32_t (*s_red)[4])
{
    PrePartial acc = {0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(parts)[i];
        acc.visible += v.x;
        acc.num_rendered += v.y;
        acc.depth_max = max(acc.depth_max, v.z);
        acc.depth_inv_min = max(acc.depth_inv_min, v.w);
    }
    for (int off = 32; off > 0; off >>= 1) {
        acc.visible += (uint32_t)__shfl_xor((int)acc.visible, off);
        acc.num_rendered += (uint32_t)__shfl_xor((int)acc.num_rendered, off);
        acc.depth_max = max(acc.depth_max, (uint32_t)__shfl_xor((int)acc.depth_max, off));
        acc.depth_inv_min = max(acc.depth_inv_min, (uint32_t)__shfl_xor((int)acc.depth_inv_min, off));
    }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_red[w][0] = acc.visible;
        s_red[w][1] = acc.num_rendered;
        s_red[w][2] = acc.depth_max;
        s_red[w][3] = acc.depth_inv_min;
    }
    __syncthreads();
    PrePartial out = {0u, 0u, 0u, 0u};
    for (int k = 0; k < nw; k++) {
        out.visible += s_red[k][0];
        out.num_rendered += s_red[k][1];
        out.depth_max = max(out.depth_max, s_red[k][2]);
        out.depth_inv_min = max(out.depth_inv_min, s_red[k][3]);
    }
    __syncthreads();
    return out;
}
__device__ inline int depth_bucket(uint32_t key, DepthRange r)
{
    if (key == 0xFFFFFFFFu) return kDepthBuckets;   // culled
    const int b = (int)((__uint_as_float(key) - r.zmin) * r.scale);
    return min(max(b, 0), kDepthBuckets - 1);
}

__global__ __launch_bounds__(256) void depth_hist_kernel(int P, const uint32_t* __restrict__ key,
                                                         const uint32_t* __restrict__ tiles,
                                                         const PrePartial* __restrict__ parts, int n_parts,
                                                         const GeomHeader* hdr, DepthSortScratch* ds,
                                                         unsigned long long* __restrict__ rows)
{
    __shared__ unsigned long long hist[kDepthBuckets + 1];
    __shared__ uint32_t s_red[16][4];
    for (int b = threadIdx.x; b <= kDepthBuckets; b += 256) hist[b] = 0;
    // depth range: from the header if it is already there, else every workgroup derives it from the preprocess
    // partials itself (header being produced on another stream); workgroup 0 leaves it for the scatter kernel
    PrePartial all;
    if (hdr) {   // the header was reduced on this stream before the launch
        all.depth_max = hdr->depth_max;
        all.depth_inv_min = hdr->depth_inv_min;
    } else {
        all = reduce_partials(parts, n_parts, s_red);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ds->depth_max = all.depth_max;
        ds->depth_inv_min = all.depth_inv_min;
    }
    const DepthRange rng = make_depth_range(all.depth_max, all.depth_inv_min);
    const int base = blockIdx.x * kHistPerBlock;
    uint32_t kv[kHistPerThread], tv[kHistPerThread];
#pragma unroll
    for (int k = 0; k < kHistPerThread; k++) {   // all loads in flight before the first LDS atomic
        const int i = base + k * 256 + threadIdx.x;
        kv[k] = i < P ? key[i] : 0xFFFFFFFFu;
        tv[k] = i < P ? tiles[i] : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kHistPerThread; k++)
        if (base + k * 256 + (int)threadIdx.x < P)
            atomicAdd(&hist[depth_bucket(kv[k], rng)], ((unsigned long long)tv[k] << kCountBits) | 1ull);
    __syncthreads();
    unsigned long long* row = rows + (size_t)blockIdx.x * (kDepthBuckets + 1);
    for (int b = threadIdx.x; b <= kDepthBuckets; b += 256) row[b] = hist[b];
}

// 64 columns per workgroup, a contiguous band of rows per wave: per column the exclusive prefix of the counts down
// the rows (each histogram workgroup's first slot inside the bucket) and the column total.
constexpr int kColWaves = 16;
__global__ __launch_bounds__(64 * kColWaves) void depth_colscan_kernel(int n_rows,
                                                                       const unsigned long long* __restrict__ rows,
                                                                       uint32_t* __restrict__ row_base,
                                                                       DepthSortScratch* ds, GeomHeader* hdr)
{
    __shared__ unsigned long long s_part[kColWaves][64];
    __shared__ uint32_t s_over;
    if (threadIdx.x == 0) s_over = 0u;
    // every overflow slot the host reads is written: this workgroup's own below, the unused ones here
    if (blockIdx.x == 0 && threadIdx.x >= gridDim.x && threadIdx.x < kOverflowSlots) hdr->sort_overflow[threadIdx.x] = 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool live = c <= kDepthBuckets;
    constexpr int kStride = kDepthBuckets + 1;
    const int band = (n_rows + kColWaves - 1) / kColWaves;
    const int r_lo = min(n_rows, w * band), r_hi = min(n_rows, r_lo + band);
    unsigned long long mine = 0;
    if (live) {
#pragma unroll 8
        for (int r = r_lo; r < r_hi; r++) mine += rows[(size_t)r * kStride + c];
    }
    s_part[w][lane] = mine;
    __syncthreads();
    unsigned long long acc = 0, total = 0;
    if (live) {
    for (int k = 0; k < kColWaves; k++) {
        if (k < w) acc += s_part[k][lane];
        total += s_part[k][lane];
    }
#pragma unroll 8
    for (int r = r_lo; r < r_hi; r++) {
        const unsigned long long v = rows[(size_t)r * kStride + c];
        row_base[(size_t)r * kStride + c] = (uint32_t)(acc & kCountMask);
        acc += v;
    }
    if (w == 0) {
        ds->total[c] = total;
        if (c < kDepthBuckets && (uint32_t)(total & kCountMask) > (uint32_t)kBucketCap) s_over = 1u;
    }
    }
    __syncthreads();
    if (threadIdx.x == 0) hdr->sort_overflow[blockIdx.x] = s_over;   // every slot the host reads is written
}

__global__ __launch_bounds__(1024) void header_reduce_kernel(const PrePartial* __restrict__ parts, int n_parts,
                                                             GeomHeader* hdr)
{
    __shared__ uint32_t s_red[16][4];
    const PrePartial all = reduce_partials(parts, n_parts, s_red);
    if (threadIdx.x == 0) {
        hdr->visible = all.visible;
        hdr->num_rendered = all.num_rendered;
        hdr->depth_max = all.depth_max;
        hdr->depth_inv_min = all.depth_inv_min;
    }
}

// Exclusive scan of the 1025 bucket totals (both packed fields at once; the count field cannot carry, P < 2^24),
// by a 256-thread workgroup into LDS.
__device__ inline void scan_bucket_totals(const DepthSortScratch* ds, uint32_t* s_start, uint32_t* s_tile,
                                          unsigned long long* s_tmp)
{
    constexpr int kPer = (kDepthBuckets + 1 + 255) / 256;
    unsigned long long c[kPer], sum = 0;
    for (int k = 0; k < kPer; k++) {
        const int b = threadIdx.x * kPer + k;
        c[k] = b <= kDepthBuckets ? ds->total[b] : 0ull;
        sum += c[k];
    }
    unsigned long long run = block256_inclusive_scan(sum, s_tmp) - sum;
    for (int k = 0; k < kPer; k++) {
        const int b = threadIdx.x * kPer + k;
        if (b <= kDepthBuckets + 1) {
            s_start[b] = (uint32_t)(run & kCountMask);
            s_tile[b] = (uint32_t)(run >> kCountBits);
        }
        run += c[k];
    }
    __syncthreads();
}

// (key, id) into the bucket regions; the culled Gaussians go straight to their final place.
__global__ __launch_bounds__(256) void depth_scatter_kernel(int P, const uint32_t* __restrict__ key,
                                                            const GeomHeader* hdr, DepthSortScratch* ds,
                                                            const uint32_t* __restrict__ row_base,
                                                            uint32_t* __restrict__ out_key,
                                                            uint32_t* __restrict__ out_id,
                                                            uint32_t* __restrict__ order,
                                                            uint32_t* __restrict__ offsets)
{
    __shared__ uint32_t slot0[kDepthBuckets + 2];   // bucket starts, then this workgroup's first slot of each bucket
    __shared__ uint32_t s_tile[kDepthBuckets + 2];
    __shared__ uint32_t rank[kDepthBuckets + 1];
    __shared__ unsigned long long s_tmp[256];
    scan_bucket_totals(ds, slot0, s_tile, s_tmp);
    if (blockIdx.x == 0)   // published for the per-bucket sort kernel
        for (int b = threadIdx.x; b <= kDepthBuckets + 1; b += 256) {
            ds->start[b] = slot0[b];
            ds->tile_base[b] = s_tile[b];
        }
    const uint32_t R = s_tile[kDepthBuckets];   // culled Gaussians add no tiles: their scan value is the total
    __syncthreads();
    const uint32_t* mybase = row_base + (size_t)blockIdx.x * (kDepthBuckets + 1);
    for (int b = threadIdx.x; b <= kDepthBuckets; b += 256) {
        slot0[b] += mybase[b];
        rank[b] = 0;
    }
    const DepthRange rng = load_depth_range(ds);
    const int base = blockIdx.x * kHistPerBlock;
    uint32_t kv[kHistPerThread];
#pragma unroll
    for (int k = 0; k < kHistPerThread; k++) {
        const int i = base + k * 256 + threadIdx.x;
        kv[k] = i < P ? key[i] : 0xFFFFFFFFu;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kHistPerThread; k++) {
        const uint32_t id = (uint32_t)(base + k * 256 + threadIdx.x);
        if ((int)id < P) {
            const int b = depth_bucket(kv[k], rng);
            const uint32_t slot = slot0[b] + atomicAdd(&rank[b], 1u);   // any order: the bucket is sorted next
            if (b == kDepthBuckets) {
                order[slot] = id;
                offsets[slot] = R;
            } else {
                out_key[slot] = kv[k];
                out_id[slot] = id;
            }
        }
    }
}

// One workgroup per bucket: sort the (key << 32 | id) words in LDS, then the inclusive scan of tiles_touched in
// that order on top of the bucket's base (rasterizer_impl.cu:441 InclusiveSum, fused).
__global__ __launch_bounds__(256) void depth_bucket_sort_kernel(const DepthSortScratch* __restrict__ ds,
                                                                const uint32_t* __restrict__ in_key,
                                                                const uint32_t* __restrict__ in_id,
                                                                const uint32_t* __restrict__ tiles,
                                                                uint32_t* __restrict__ order,
                                                                uint32_t* __restrict__ offsets)
{
    __shared__ unsigned long long s[kBucketCap];
    __shared__ uint32_t s_sum[256];
    const int b = blockIdx.x;
    const uint32_t start = ds->start[b], n = ds->start[b + 1] - start;
    if (n == 0) return;
    // A bucket that overflows the LDS (sort_overflow is set and the host reruns the generic sort + scan) is passed
    // through unsorted so that `order` holds valid ids.
    if (n > (uint32_t)kBucketCap) {
        for (uint32_t r = threadIdx.x; r < n; r += 256) order[start + r] = in_id[start + r];
        return;
    }
    uint32_t N = 2;
    while (N < n) N <<= 1;
    for (uint32_t r = threadIdx.x; r < N; r += 256)
        s[r] = r < n ? ((unsigned long long)in_key[start + r] << 32) | in_id[start + r] : ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += 256) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;   // lo has bit j clear
                const unsigned long long a = s[lo], c = s[hi];
                const bool up = (lo & k) == 0;
                if ((a > c) == up) {
                    s[lo] = c;
                    s[hi] = a;
                }
            }
            __syncthreads();
        }
    // scan: each thread owns `per` consecutive sorted entries
    const uint32_t per = (n + 255) / 256;
    const uint32_t r0 = threadIdx.x * per;
    uint32_t tl[kBucketCap / 256], mine = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t r = r0 + k;
        tl[k] = r < n ? tiles[(uint32_t)s[r]] : 0u;
        mine += tl[k];
    }
    uint32_t run = ds->tile_base[b] + block256_inclusive_scan(mine, s_sum) - mine;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t r = r0 + k;
        if (r < n) {
            run += tl[k];
            order[start + r] = (uint32_t)s[r];
            offsets[start + r] = run;
        }
    }
}

constexpr int kColBlocks = (kDepthBuckets + 1 + 63) / 64;
static_assert(kColBlocks <= kOverflowSlots, "one overflow slot per column-scan workgroup");

void run_header_reduce(int P, GeomState& g, hipStream_t s)
{
    header_reduce_kernel<<<1, 1024, 0, s>>>(g.partials, (int)pre_partials((size_t)P), g.header);
}

void run_depth_histogram(int P, GeomState& g, bool header_ready, hipStream_t s)
{
    const int rows = (int)depth_hist_rows((size_t)P), np = (int)pre_partials((size_t)P);
    depth_hist_kernel<<<rows, 256, 0, s>>>(P, g.depth_key, g.tiles, g.partials, np, header_ready ? g.header : nullptr,
                                           g.dsort, g.hist_rows);
    depth_colscan_kernel<<<kColBlocks, 64 * kColWaves, 0, s>>>(rows, g.hist_rows, g.hist_base, g.dsort, g.header);
}

void run_depth_bucket_sort_and_scan(int P, GeomState& g, hipStream_t s)
{
    const int rows = (int)depth_hist_rows((size_t)P);
    depth_scatter_kernel<<<rows, 256, 0, s>>>(P, g.depth_key, g.header, g.dsort, g.hist_base, g.key_sorted,
                                              g.bucket_id, g.order, g.offsets);
    depth_bucket_sort_kernel<<<kDepthBuckets, 256, 0, s>>>(g.dsort, g.key_sorted, g.bucket_id, g.tiles, g.order,
                                                           g.offsets);
}

int tile_rank_bits(int P, size_t n_tiles)
{
    static const bool force_pairs = [] {   // R3DGS_TILE_SORT=pairs | rocprim | (default: own radix on packed words)
        const char* v = getenv("R3DGS_TILE_SORT");
        return v && std::string(v) == "pairs";
    }();
    if (force_pairs) return 0;
    const uint32_t tb = higher_msb((uint32_t)n_tiles), rb = higher_msb((uint32_t)P);
    return tb + rb <= 32 ? (int)rb : 0;
}

// Pair emission, balanced over OUTPUT positions.  The reference loops one thread over all tiles of its
// Gaussian (rasterizer_impl.cu:106-117); in depth order the nearest -- largest -- splats sit next to each
// other, so any per-Gaussian (or per-64-Gaussian) work split has a tail: the first GPU version of this
// kernel spent 234 us waiting for its first few waves.  Here every workgroup owns kEmitPerBlock
// consecutive output slots: two lanes locate the slot range's first/last source Gaussian by binary
// search in the inclusive scan, the block stages that <= kEmitPerBlock+1 long slice (end offset, id,
// tile rect) in LDS, and every lane resolves its slots with an LDS binary search.  Stores are coalesced.
constexpr int kEmitPerBlock = 1024;
constexpr int kEmitSlice = kEmitPerBlock + 8;

// Smallest j in [0, n] with a[j] > pos (n if none), a non-decreasing, searched by a whole wave: 64 probes per step,
// four dependent loads for P = 500k instead of the nineteen of a scalar binary search.  All 64 lanes must call it;
// the result is wave-uniform.
__device__ __forceinline__ uint32_t upper_bound_wave(const uint32_t* __restrict__ a, uint32_t n, uint32_t pos, int lane)
{
    uint32_t lo = 0, hi = n;  // the answer (smallest j with a[j] > pos, n if none) lies in [lo, hi]
    while (hi - lo > 64u) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t idx = lo + ((uint32_t)lane + 1u) * step - 1u;   // ascending probes; the last one reaches >= hi - 1
        const bool gt = idx < hi ? a[idx] > pos : true;
        const unsigned long long m = __ballot(gt);
        if (m == 0ull) return hi;   // even a[hi - 1] <= pos (lane 63 probed it): the answer is hi
        const uint32_t f = (uint32_t)__builtin_ctzll(m);
        const uint32_t probe_f = lo + (f + 1u) * step - 1u;
        const uint32_t new_lo = f == 0u ? lo : lo + f * step;          // one past the last probe that was <= pos
        hi = probe_f < hi ? probe_f : hi;
        lo = new_lo;
    }
    const uint32_t idx = lo + (uint32_t)lane;
    const unsigned long long m = __ballot(idx < hi && a[idx] > pos);
    return m ? lo + (uint32_t)__builtin_ctzll(m) : hi;
}

__global__ __launch_bounds__(256) void emit_pairs_kernel(int P, uint32_t R, const uint32_t* __restrict__ order,
                                                         const uint32_t* __restrict__ offsets,
                                                         const ushort4* __restrict__ rect, int gx, GRec* rec,
                                                         int rank_bits, uint32_t* __restrict__ tile_out,
                                                         uint32_t* __restrict__ id_out, uint2* __restrict__ ranges,
                                                         uint32_t n_tiles, uint32_t* __restrict__ radix_rows)
{
    __shared__ uint32_t s_hist[kRadixBins];   // first radix digit of the packed tile sort, counted while emitting
    if (radix_rows && threadIdx.x < kRadixBins) s_hist[threadIdx.x] = 0;
    // the tile ranges start out as (0, 0) (rasterizer_impl.cu:475 memset): cleared here, ahead of the sort, instead of
    // by a fill of its own on the stream
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n_tiles; t += gridDim.x * 256u) ranges[t] = make_uint2(0u, 0u);
    __shared__ uint32_t s_end[kEmitSlice];
    __shared__ uint32_t s_id[kEmitSlice];
    __shared__ ushort4 s_rect[kEmitSlice];
    __shared__ uint32_t s_j[2];
    __shared__ uint32_t s_start0;
    const uint32_t pos0 = blockIdx.x * (uint32_t)kEmitPerBlock;
    const uint32_t pos1 = min(R, pos0 + (uint32_t)kEmitPerBlock);  // exclusive
    if (threadIdx.x < 128) {   // wave 0 locates the first slot's Gaussian, wave 1 the last slot's
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const uint32_t j = upper_bound_wave(offsets, (uint32_t)P, w == 0 ? pos0 : pos1 - 1, lane);
        if (lane == 0) {
            s_j[w] = j;
            if (w == 0) s_start0 = j == 0 ? 0u : offsets[j - 1];
        }
    }
    __syncthreads();
    const uint32_t j_lo = s_j[0], j_hi = s_j[1];
    // every Gaussian inside the slice owns >= 1 slot (culled ones sort to the very end), so n <= slots + 1
    const uint32_t n = min(j_hi - j_lo + 1u, (uint32_t)kEmitSlice);
    for (uint32_t k = threadIdx.x; k < n; k += 256) {
        const uint32_t id = order[j_lo + k];
        s_end[k] = offsets[j_lo + k];
        s_id[k] = id;
        s_rect[k] = rect[id];
    }
    __syncthreads();
    const uint32_t start0 = s_start0;
    // record where each staged Gaussian's pairs begin (blocks sharing a Gaussian write the same value)
    for (uint32_t k = threadIdx.x; k < n; k += 256) rec[s_id[k]].pair_start = k == 0 ? start0 : s_end[k - 1];
#pragma unroll
    for (int e = 0; e < kEmitPerBlock / 256; e++) {
        const uint32_t pos = pos0 + threadIdx.x + (uint32_t)e * 256u;
        if (pos < pos1) {
            uint32_t lo = 0, hi = n - 1;  // smallest k with s_end[k] > pos; exists because pos < s_end[n-1]
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (s_end[mid] > pos)
                    hi = mid;
                else
                    lo = mid + 1;
            }
            const uint32_t start = lo == 0 ? start0 : s_end[lo - 1];
            const uint32_t local = pos - start;
            const ushort4 r = s_rect[lo];
            const uint32_t w = (uint32_t)r.z - (uint32_t)r.x;
            const uint32_t ty = local / w, tx = local - ty * w;  // row-major (y, x), rasterizer_impl.cu:106-117
            const uint32_t tile = ((uint32_t)r.y + ty) * (uint32_t)gx + (uint32_t)r.x + tx;
            if (rank_bits) {
                tile_out[pos] = (tile << rank_bits) | (j_lo + lo);   // one word: tile | rank in depth order
                if (radix_rows) atomicAdd(&s_hist[tile & (kRadixBins - 1)], 1u);
            } else {
                tile_out[pos] = tile;
                id_out[pos] = s_id[lo];
            }
        }
    }
    if (radix_rows) {
        __syncthreads();
        if (threadIdx.x < kRadixBins) radix_rows[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_hist[threadIdx.x];
    }
}

// ---- tile sort of the packed pair words: LSD radix, two 7-bit digits ------------------------------------------------
// rocPRIM's onesweep spends more time around its two passes (per-pass fills of the look-back state, histogram and scan
// kernels: ~65 us of launches and gaps for ~48 us of sorting at R = 3.6 M) than in them.  Here a pass is: per-workgroup
// digit counts (for the first digit they come out of the emission kernel), one workgroup per digit scanning its row
// of counts, and a scatter that ranks its 1024 keys stably -- wave-level match by seven ballots per round, running
// per-wave digit counts in LDS -- no fills, no look-back chain.  Stable, so the two passes give the tile-major order
// with the emission (depth) order preserved inside a tile.
__global__ __launch_bounds__(256) void radix_hist_kernel(uint32_t R, const uint32_t* __restrict__ in, int shift,
                                                         uint32_t* __restrict__ rows)
{
    __shared__ uint32_t s_hist[kRadixBins];
    if (threadIdx.x < kRadixBins) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (uint32_t)kRadixBlock;
#pragma unroll
    for (int k = 0; k < kRadixBlock / 256; k++) {
        const uint32_t i = base + k * 256u + threadIdx.x;
        if (i < R) atomicAdd(&s_hist[(in[i] >> shift) & (kRadixBins - 1)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kRadixBins) rows[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = s_hist[threadIdx.x];
}

// one workgroup per digit: exclusive scan of that digit's per-workgroup counts, and the digit total
__global__ __launch_bounds__(256) void radix_digit_scan_kernel(uint32_t nb, const uint32_t* __restrict__ rows,
                                                               uint32_t* __restrict__ base, uint32_t* __restrict__ total)
{
    __shared__ uint32_t s_sum[256];
    const uint32_t* row = rows + (size_t)blockIdx.x * nb;
    uint32_t* out = base + (size_t)blockIdx.x * nb;
    const uint32_t per = (nb + 255u) / 256u, e0 = threadIdx.x * per;
    uint32_t mine = 0;
#pragma unroll 8
    for (uint32_t k = 0; k < per; k++) mine += e0 + k < nb ? row[e0 + k] : 0u;
    const uint32_t incl = block256_inclusive_scan(mine, s_sum);
    uint32_t run = incl - mine;
#pragma unroll 8
    for (uint32_t k = 0; k < per; k++)
        if (e0 + k < nb) {
            const uint32_t v = row[e0 + k];
            out[e0 + k] = run;
            run += v;
        }
    if (threadIdx.x == 255) total[blockIdx.x] = incl;
}

// (Materialising point_list[pos] = order[rank] here in the last pass was measured: the dependent gather lengthens this
// kernel by 19 us and saves 13 us in tile_ranges_kernel, so it stays there.)
__global__ __launch_bounds__(256) void radix_scatter_kernel(uint32_t R, const uint32_t* __restrict__ in,
                                                            uint32_t* __restrict__ out, int shift,
                                                            const uint32_t* __restrict__ base,
                                                            const uint32_t* __restrict__ total)
{
    constexpr int kWaves = 4, kRounds = kRadixBlock / 256;
    __shared__ uint32_t s_wcount[kWaves][kRadixBins];   // running per-wave digit counts
    __shared__ uint32_t s_start[kRadixBins];            // exclusive scan of the digit totals
    __shared__ uint32_t s_off[kWaves][kRadixBins];      // digit start + workgroup base + waves below
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < kWaves * kRadixBins; t += 256) (&s_wcount[0][0])[t] = 0;
    if (w == 0) {   // 128 totals, two per lane, scanned with shuffles
        const uint32_t v0 = total[2 * lane], v1 = total[2 * lane + 1];
        uint32_t incl = v0 + v1;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += up;
        }
        const uint32_t excl = incl - (v0 + v1);
        s_start[2 * lane] = excl;
        s_start[2 * lane + 1] = excl + v0;
    }
    __syncthreads();
    const uint32_t blk = blockIdx.x * (uint32_t)kRadixBlock + (uint32_t)w * (kRadixBlock / kWaves);
    uint32_t key[kRounds], lrank[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const uint32_t i = blk + r * 64u + lane;
        key[r] = i < R ? in[i] : 0u;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const bool valid = blk + r * 64u + lane < R;
        const uint32_t d = (key[r] >> shift) & (kRadixBins - 1);
        unsigned long long peers = __ballot(valid);   // lanes of this round holding the same digit
#pragma unroll
        for (int b = 0; b < kRadixBits; b++) {
            const bool bit = (d >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const uint32_t seen = s_wcount[w][d];          // same-digit keys of this wave in earlier rounds
        lrank[r] = seen + (uint32_t)__popcll(peers & below);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & below) == 0ull) s_wcount[w][d] = seen + (uint32_t)__popcll(peers);   // group leader
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (threadIdx.x < kRadixBins) {
        const int d = threadIdx.x;
        uint32_t run = s_start[d] + base[(size_t)d * gridDim.x + blockIdx.x];
#pragma unroll
        for (int k = 0; k < kWaves; k++) {
            s_off[k][d] = run;
            run += s_wcount[k][d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        if (blk + r * 64u + lane < R) {
            const uint32_t d = (key[r] >> shift) & (kRadixBins - 1);
            out[s_off[w][d] + lrank[r]] = key[r];
        }
    }
}

// rasterizer_impl.cu:124-146 identifyTileRanges on the sorted 32-bit keys; the packed sort's Gaussian ids
// (point_list[i] = order[rank]) are materialised here too.
__global__ __launch_bounds__(256) void tile_ranges_kernel(int R, const uint32_t* __restrict__ tile_sorted,
                                                          int rank_bits, const uint32_t* __restrict__ order,
                                                          uint32_t* __restrict__ point_list, uint2* ranges,
                                                          unsigned char* __restrict__ pair_flag)
{
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;   // four consecutive entries per thread (arrays 256-B aligned)
    if (i0 >= R) return;
    // the backward's "row written" flags start out zero (it puts the ones it consumed back itself)
    if (i0 + 4 <= R)
        *reinterpret_cast<uint32_t*>(pair_flag + i0) = 0u;
    else
        for (int k = i0; k < R; k++) pair_flag[k] = 0;
    uint32_t key[4];
    const int n = min(4, R - i0);
    if (n == 4) {
        const uint4 k4 = *reinterpret_cast<const uint4*>(tile_sorted + i0);
        key[0] = k4.x; key[1] = k4.y; key[2] = k4.z; key[3] = k4.w;
    } else {
        for (int k = 0; k < 4; k++) key[k] = k < n ? tile_sorted[i0 + k] : 0u;
    }
    uint32_t prev = i0 ? tile_sorted[i0 - 1] >> rank_bits : 0xFFFFFFFFu;
    if (rank_bits && order) {
        const uint32_t mask = (1u << rank_bits) - 1u;
        uint32_t id[4];
        for (int k = 0; k < 4; k++) id[k] = k < n ? order[key[k] & mask] : 0u;
        if (n == 4)
            *reinterpret_cast<uint4*>(point_list + i0) = make_uint4(id[0], id[1], id[2], id[3]);
        else
            for (int k = 0; k < n; k++) point_list[i0 + k] = id[k];
    }
    for (int k = 0; k < n; k++) {
        const uint32_t cur = key[k] >> rank_bits;
        const int i = i0 + k;
        if (i == 0)
            ranges[cur].x = 0;
        else if (cur != prev) {
            ranges[prev].y = i;
            ranges[cur].x = i;
        }
        if (i == R - 1) ranges[cur].y = R;
        prev = cur;
    }
}

void run_tile_binning(int P, int R, int gx, int gy, GeomState& g, BinState& b, ImageState& img, hipStream_t s)
{
    const size_t Tn = (size_t)gx * gy;
    if (R <= 0) {
        R3_HIP(hipMemsetAsync(img.ranges, 0, Tn * sizeof(uint2), s));  // rasterizer_impl.cu:475
        return;
    }
    const int rank_bits = tile_rank_bits(P, Tn);
    const int bits = (int)higher_msb((uint32_t)Tn);
    const uint32_t nb = (uint32_t)((R + kEmitPerBlock - 1) / kEmitPerBlock);
    static const bool rocprim_tiles = [] {   // R3DGS_TILE_SORT=rocprim: packed keys through rocPRIM's onesweep (A/B)
        const char* v = getenv("R3DGS_TILE_SORT");
        return v && std::string(v) == "rocprim";
    }();
    const bool own_radix = rank_bits && bits <= 2 * kRadixBits && !rocprim_tiles;
    hipLaunchKernelGGL(emit_pairs_kernel, dim3(nb), dim3(256), 0, s, P, (uint32_t)R, g.order, g.offsets, g.rect, gx, g.rec,
                       rank_bits, b.tile_in, b.gauss_in, img.ranges, (uint32_t)Tn, own_radix ? b.radix_rows : nullptr);
    size_t bytes = b.temp_bytes;
    if (own_radix) {
        // pass 1: low 7 tile bits (counts from the emission kernel), tile_in -> gauss_in (free in the packed sort)
        hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(kRadixBins), dim3(256), 0, s, nb, b.radix_rows, b.radix_base,
                           b.radix_total);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(256), 0, s, (uint32_t)R, b.tile_in, b.gauss_in, rank_bits,
                           b.radix_base, b.radix_total);
        // pass 2: the remaining tile bits, gauss_in -> tile_sorted
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nb), dim3(256), 0, s, (uint32_t)R, b.gauss_in, rank_bits + kRadixBits,
                           b.radix_rows);
        hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(kRadixBins), dim3(256), 0, s, nb, b.radix_rows, b.radix_base,
                           b.radix_total + kRadixBins);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nb), dim3(256), 0, s, (uint32_t)R, b.gauss_in, b.tile_sorted,
                           rank_bits + kRadixBits, b.radix_base, b.radix_total + kRadixBins);
    } else if (rank_bits)
        R3_HIP(rocprim::radix_sort_keys(b.temp, bytes, b.tile_in, b.tile_sorted, (size_t)R, rank_bits, rank_bits + bits, s));
    else
        R3_HIP(rocprim::radix_sort_pairs(b.temp, bytes, b.tile_in, b.tile_sorted, b.gauss_in, b.point_list, (size_t)R, 0,
                                         bits, s));
    hipLaunchKernelGGL(tile_ranges_kernel, dim3((R + 1023) / 1024), dim3(256), 0, s, R, b.tile_sorted, rank_bits,
                       (const uint32_t*)g.order,
                       b.point_list, img.ranges, b.pair_flag);
}

// debug accessor: rebuild the reference's 64-bit keys (tile << 32 | depth bits) of the sorted list
__global__ __launch_bounds__(256) void export_keys_kernel(int R, int rank_bits,
                                                          const uint32_t* __restrict__ tile_sorted,
                                                          const uint32_t* __restrict__ point_list,
                                                          const uint32_t* __restrict__ depth_key, uint64_t* keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    keys[i] = ((uint64_t)(tile_sorted[i] >> rank_bits) << 32) | (uint64_t)depth_key[point_list[i]];
}

void launch_export_keys(int P, int R, size_t n_tiles, const BinState& b, const GeomState& g, uint64_t* keys_out,
                        hipStream_t s)
{
    if (R <= 0) return;
    hipLaunchKernelGGL(export_keys_kernel, dim3((R + 255) / 256), dim3(256), 0, s, R, tile_rank_bits(P, n_tiles),
                       b.tile_sorted, b.point_list,
                       g.depth_key, keys_out);
}

}  // namespace r3
