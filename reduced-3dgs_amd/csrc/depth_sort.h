// depth_sort.h -- the bucketed depth sort of the P Gaussians (+ the fused inclusive scan of tiles_touched), written
// as workgroup ROLES so that the kernels that run them can carry a second, independent role in their spare
// workgroups (preprocess.hip fuses the SH -> RGB stream into them: the sort kernels are latency-bound and leave most
// of the chip idle, the colour kernel is bandwidth-bound, and a graph with a side branch costs the host as much as
// twenty direct launches).
//
// Replaces, together with binning.hip, rasterizer_impl.cu:441 (InclusiveSum) and the depth half of the 64-bit
// (tile|depth) key sort of rasterizer_impl.cu:465-473 of
// /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer.
//
// A generic device sort is launch-latency bound at this size (rocPRIM: block sort + 9 merge passes, ~125 us for 500k
// keys).  The keys are view depths whose range the preprocess kernel already knows, so:
//   (1) every workgroup histograms its 4096 Gaussians over nb monotone buckets of [min depth, max depth] and stores its
//       row -- no global atomics: device-scope atomics on the same few lines were the cost of the first version;
//   (2) a column-wise scan of the rows (each workgroup's first slot inside each bucket) and the bucket totals;
//   (3) (key, id) scatter into the bucket regions -- slot = bucket start (scan of the totals, redone per workgroup
//       in LDS) + row base + LDS rank, again no global atomics;
//   (4) one workgroup per bucket sorts its pairs in LDS as 64-bit (key << 32 | id) words and scans tiles_touched in
//       that order on top of the bucket's base.
// What travels: the scatter (3) writes ONE 16-byte record per Gaussian -- depth key, id, tile rect -- into its bucket's
// region (one write transaction instead of two 4-byte ones), and the bucket sort (4) sorts 64-bit words
//   (key - smallest key of the bucket) << 40 | id << 16 | position in the bucket
// (a bucket is at most 2^31 / 1024 key values wide, ids are below 2^24, a bucket sorted here holds at most 4096), so that
// every sorted entry knows where its record is: the tile count for the scan is the area of the rect in the record --
// no gather from tiles[id], which at 6 M Gaussians was 6 M random reads over 24 MB -- and the rect leaves in DEPTH ORDER
// (rect_sorted) for the pair emission, which had the same gather.
// The bucket function is monotone in the key, so concatenating the sorted buckets is the stable sort by depth bits
// the reference's 64-bit key sort implies.  Buckets are equal-width in the key's BIT PATTERN (positive floats order
// like their bits), i.e. roughly logarithmic in depth: an unbounded scene with its content at 2..8 units and a
// background out to 100 puts ~a quarter of the buckets under the content instead of 6% of them.  nb grows with P
// (mean load <= 512).  A bucket that still exceeds the LDS capacity (a fronto-parallel plane of splats) is sorted by
// its workgroup with a radix sort in global memory -- correct, slow, and flagged in the header / PassInfo so that
// the host can route later passes of such a view through the generic sort.
#pragma once
#include "common.h"

namespace r3 {

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

// Stable partition of one wave's 64 digits: lanes holding the same digit as this lane (including itself).
template <int BITS>
__device__ __forceinline__ unsigned long long match_digit(uint32_t d, bool valid)
{
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; b++) {
        const bool bit = (d >> b) & 1u;
        const unsigned long long bal = __ballot(bit);
        peers &= bit ? bal : ~bal;
    }
    return peers;
}

constexpr int kWaveSortMax = 512;    // largest bucket one wave sorts in registers (8 words per lane)
constexpr int kGlobalSplitMax = 64;                                  // sub-buckets of a bucket that does not fit the LDS sort
constexpr uint32_t kGlobalSplitCap = kGlobalSplitMax * 2048u;       // ... which it can hold at an average of half the LDS capacity
constexpr int kBucketsPerGroup = 4;  // buckets per 256-thread workgroup of the bucket-sort kernel: one per wave
constexpr int kHistPerThread = kHistBatch / 256;
constexpr int kCountBits = 24;
constexpr unsigned long long kCountMask = (1ull << kCountBits) - 1;

// offsets[slot] = end of the pairs of the Gaussian at depth rank `slot` (inclusive scan of tiles_touched); the Gaussian
// holding pair m * kRadixBlock also notes its rank for the emission workgroup that starts there.
__device__ __forceinline__ void publish_offset(const DepthArgs& a, uint32_t slot, uint32_t end, uint32_t count)
{
    a.offsets[slot] = end;
    if (a.block_first && count) {
        const uint32_t first = end - count;
        for (uint32_t m = (first + (uint32_t)kRadixBlock - 1u) / (uint32_t)kRadixBlock;
             m * (uint32_t)kRadixBlock < end && m < a.block_cap; m++)
            a.block_first[m] = slot;
    }
}

struct DepthRange {
    uint32_t kmin;
    float scale;
    int nb;
};
__device__ inline DepthRange make_depth_range(uint32_t mx, uint32_t mi, int nb)
{
    DepthRange r;
    const uint32_t kmax = mx, kmin = ~mi;
    r.kmin = kmin;
    r.nb = nb;
    r.scale = kmax >= kmin ? (float)nb / ((float)(kmax - kmin) + 1.0f) : 0.0f;   // no visible Gaussian: nothing is looked up
    return r;
}
// Monotone in key (conversion, multiplication by a positive constant and truncation all are), which is all the sort
// needs; fp32 rounding only moves bucket boundaries.
__device__ inline int depth_bucket(uint32_t key, const DepthRange& r)
{
    if (key == 0xFFFFFFFFu) return r.nb;   // culled
    const int b = (int)((float)(key - r.kmin) * r.scale);
    return min(max(b, 0), r.nb - 1);
}

// LDS bytes of each role (the kernels size their dynamic LDS as the maximum over the roles they carry)
inline size_t depth_hist_lds(int nb) { return (size_t)(nb + 1) * sizeof(unsigned long long); }
inline size_t depth_scatter_lds(int nb) { return (size_t)2 * (nb + 2) * sizeof(uint32_t) + 256 * sizeof(unsigned long long); }
constexpr size_t kBucketSortLds = (size_t)kBucketCap * 8 + 256 * 4 + (256 + 256 + 4 * 256 + 4) * 4;

// Sum / max of the preprocess workgroups' partials by one workgroup (s_red: 16 x 6 words).
__device__ inline PrePartial reduce_partials(const PrePartial* __restrict__ parts, int n, uint32_t (*s_red)[6])
{
    PrePartial acc = {0u, 0u, 0u, 0u, 0u, 0u, {0u, 0u}};
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const uint4 v = reinterpret_cast<const uint4*>(parts)[2 * i], w = reinterpret_cast<const uint4*>(parts)[2 * i + 1];
        acc.visible += v.x;
        acc.num_rendered += v.y;
        acc.depth_max = max(acc.depth_max, v.z);
        acc.depth_inv_min = max(acc.depth_inv_min, v.w);
        acc.binned += w.x;
        acc.rendered_ref += w.y;
    }
    for (int off = 32; off > 0; off >>= 1) {
        acc.visible += (uint32_t)__shfl_xor((int)acc.visible, off);
        acc.num_rendered += (uint32_t)__shfl_xor((int)acc.num_rendered, off);
        acc.depth_max = max(acc.depth_max, (uint32_t)__shfl_xor((int)acc.depth_max, off));
        acc.depth_inv_min = max(acc.depth_inv_min, (uint32_t)__shfl_xor((int)acc.depth_inv_min, off));
        acc.binned += (uint32_t)__shfl_xor((int)acc.binned, off);
        acc.rendered_ref += (uint32_t)__shfl_xor((int)acc.rendered_ref, off);
    }
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_red[w][0] = acc.visible;
        s_red[w][1] = acc.num_rendered;
        s_red[w][2] = acc.depth_max;
        s_red[w][3] = acc.depth_inv_min;
        s_red[w][4] = acc.binned;
        s_red[w][5] = acc.rendered_ref;
    }
    __syncthreads();
    PrePartial out = {0u, 0u, 0u, 0u, 0u, 0u, {0u, 0u}};
    for (int k = 0; k < nw; k++) {
        out.visible += s_red[k][0];
        out.num_rendered += s_red[k][1];
        out.depth_max = max(out.depth_max, s_red[k][2]);
        out.depth_inv_min = max(out.depth_inv_min, s_red[k][3]);
        out.binned += s_red[k][4];
        out.rendered_ref += s_red[k][5];
    }
    __syncthreads();
    return out;
}

// The device header every later kernel reads (visible count, the pairs wanted, that count clamped to the reservation,
// depth range), and the pass's numbers published in its host-mapped PassInfo slot: the host never has to wait for the
// pass, and when it wants them (strict mode's check, exact-size path) it polls plain memory.  One thread.
__device__ inline void write_header(const HeaderArgs& a, const PrePartial& all)
{
    GeomHeader* hdr = a.hdr;
    hdr->visible = all.visible;
    hdr->num_rendered = all.num_rendered;
    hdr->depth_max = all.depth_max;
    hdr->depth_inv_min = all.depth_inv_min;
    hdr->num_pairs = min(all.num_rendered, a.reserve);
    hdr->reserve = a.reserve;
    hdr->sort_overflow = 0u;
    hdr->binned = all.binned;
    hdr->rendered_ref = all.rendered_ref;
    hdr->sh_cache = a.sh_cache;
    hdr->ckpt = a.ckpt;
    {   // lists at least this long get checkpoints (common.h): a multiple of the pass's mean list length, two segments at least
        const unsigned long long mean_pct = (unsigned long long)hdr->num_pairs * a.ckpt_factor_pct / (a.n_tiles ? a.n_tiles : 1u);
        const uint32_t thr = (uint32_t)min(mean_pct / 100ull, 0x7FFFFFFFull);
        hdr->ckpt_thr = a.ckpt ? max(thr, 2u << a.ckpt) : 0xFFFFFFFFu;
    }
    volatile PassInfo* info = a.info;
    if (info) {
        info->num_rendered = all.rendered_ref;
        info->pairs = all.num_rendered;
        info->visible = all.visible;
        info->reserve = a.reserve;
        info->sort_overflow = 0u;
        __threadfence_system();
        if (a.stamp_sort) info->sort_seq = a.ticket;
        info->seq = a.ticket;
    }
}

// (1) workgroup `wg` of depth_hist_rows(P): histogram of its per_block Gaussians, kHistBatch at a time.  With
// a.fuse_header every histogram workgroup first reduces the preprocess partials itself (2 k x 16 B out of L2) and
// workgroup 0 writes the header: the one-workgroup header kernel and its launch gap leave the chain.
__device__ inline void depth_hist_role(const DepthArgs& a, const HeaderArgs& h, char* smem, int wg)
{
    __shared__ uint32_t s_red[16][6];
    unsigned long long* hist = reinterpret_cast<unsigned long long*>(smem);   // [nb + 1]
    const int P = a.P, nb = a.nb;
    for (int b = threadIdx.x; b <= nb; b += 256) hist[b] = 0;
    uint32_t dmax, dinv;
    if (wg == 0 && threadIdx.x == 0) a.ds->big_count = 0u;   // (the column scan, which fills the list, runs later)
    if (a.fuse_header) {
        const PrePartial all = reduce_partials(h.parts, h.n_parts, s_red);
        if (wg == 0 && threadIdx.x == 0) write_header(h, all);
        dmax = all.depth_max;
        dinv = all.depth_inv_min;
    } else {
        dmax = a.hdr->depth_max;
        dinv = a.hdr->depth_inv_min;
    }
    const DepthRange rng = make_depth_range(dmax, dinv, nb);
    __syncthreads();
    for (int base = wg * a.per_block; base < min(P, (wg + 1) * a.per_block); base += kHistBatch) {
        uint32_t kv[kHistPerThread], tv[kHistPerThread];
#pragma unroll
        for (int k = 0; k < kHistPerThread; k++) {   // all loads in flight before the first LDS atomic
            const int i = base + k * 256 + threadIdx.x;
            kv[k] = i < P ? a.key[i] : 0xFFFFFFFFu;
            tv[k] = i < P ? a.tiles[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < kHistPerThread; k++)
            if (base + k * 256 + (int)threadIdx.x < P)
                atomicAdd(&hist[depth_bucket(kv[k], rng)], ((unsigned long long)tv[k] << kCountBits) | 1ull);
    }
    __syncthreads();
    unsigned long long* row = a.hist_rows + (size_t)wg * (nb + 1);
    for (int b = threadIdx.x; b <= nb; b += 256) row[b] = hist[b];
}

// (2) 64 columns per workgroup of 1024 threads, a contiguous band of rows per wave: per column the exclusive prefix
// of the counts down the rows (each histogram workgroup's first slot inside the bucket) and the column total.
constexpr int kColWaves = 16;
__device__ inline void depth_colscan_role(const DepthArgs& a, int wg)
{
    __shared__ unsigned long long s_part[kColWaves][64];
    __shared__ uint32_t s_over;
    const int n_rows = a.rows, nb = a.nb;
    const unsigned long long* __restrict__ rows = a.hist_rows;
    uint32_t* __restrict__ row_base = a.hist_base;
    if (threadIdx.x == 0) s_over = 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = wg * 64 + lane;
    const bool live = c <= nb;
    const int kStride = nb + 1;
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
            a.ds->total[c] = total;
            if (c < nb && (uint32_t)(total & kCountMask) > kGlobalSplitCap) s_over = 1u;
        }
    }
    if (w == 0) {   // the buckets one wave cannot sort in registers go on the list the bucket-sort kernel's workgroups share
        const bool big = live && c < nb && (uint32_t)(total & kCountMask) > (uint32_t)kWaveSortMax;
        const unsigned long long m = __ballot(big);
        if (m) {
            uint32_t base = 0u;
            const int leader = __builtin_ctzll(m);
            if (lane == leader) base = atomicAdd(&a.ds->big_count, (uint32_t)__builtin_popcountll(m));
            base = (uint32_t)__shfl((int)base, leader);
            if (big) a.ds->big_list[base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)c;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_over) a.hdr->sort_overflow = 1u;   // zeroed by the header earlier in this pass
}

// Exclusive scan of the nb + 1 bucket totals (both packed fields at once; the count field cannot carry, P < 2^24) by a
// 256-thread workgroup: bucket starts into LDS, and -- by the publishing workgroup only -- starts and tile bases into
// global memory for the bucket-sort kernel.  The totals are read twice (L2) instead of being held in registers.
__device__ inline void scan_bucket_totals(DepthSortScratch* ds, int nb, uint32_t* s_start, unsigned long long* s_tmp,
                                          bool publish)
{
    const int per = (nb + 2 + 255) / 256;
    const int b0 = threadIdx.x * per;
    unsigned long long sum = 0;
    for (int k = 0; k < per; k++) sum += b0 + k <= nb ? ds->total[b0 + k] : 0ull;
    unsigned long long run = block256_inclusive_scan(sum, s_tmp) - sum;
    for (int k = 0; k < per; k++) {
        const int b = b0 + k;
        if (b <= nb + 1) {
            s_start[b] = (uint32_t)(run & kCountMask);
            if (publish) {
                ds->start[b] = (uint32_t)(run & kCountMask);
                ds->tile_base[b] = (uint32_t)(run >> kCountBits);
            }
            run += b <= nb ? ds->total[b] : 0ull;
        }
    }
    __syncthreads();
}

// (3) (key, id) into the bucket regions; the culled Gaussians go straight to their final place.
__device__ inline void depth_scatter_role(const DepthArgs& a, char* smem, int wg)
{
    const int P = a.P, nb = a.nb;
    unsigned long long* s_tmp = reinterpret_cast<unsigned long long*>(smem);   // [256]
    uint32_t* slot0 = reinterpret_cast<uint32_t*>(smem + 256 * sizeof(unsigned long long));   // bucket starts, then this
    uint32_t* rank = slot0 + (nb + 2);                                         // workgroup's first slot of each bucket
    scan_bucket_totals(a.ds, nb, slot0, s_tmp, wg == 0);   // workgroup 0 publishes for the per-bucket sort kernel
    // The host-visible overflow hint carries its own stamp: the header (PassInfo::seq) is published by an earlier
    // kernel, and a host looking in between must not take "no overflow" from a slot this pass has not written yet.
    // Written here, one kernel behind the column scan that decides it, by one thread.
    if (wg == 0 && threadIdx.x == 0 && a.info) {
        volatile PassInfo* info = a.info;
        info->sort_overflow = a.hdr->sort_overflow;
        __threadfence_system();
        info->sort_seq = a.ticket;
    }
    const uint32_t R = a.hdr->num_rendered;   // culled Gaussians add no tiles: their scan value is the total
    const uint32_t* mybase = a.hist_base + (size_t)wg * (nb + 1);
    for (int b = threadIdx.x; b <= nb; b += 256) {
        slot0[b] += mybase[b];
        rank[b] = 0;
    }
    const DepthRange rng = make_depth_range(a.hdr->depth_max, a.hdr->depth_inv_min, nb);
    __syncthreads();
    const uint2* __restrict__ rects = reinterpret_cast<const uint2*>(a.rect);   // (x0 | y0 << 16, x1 | y1 << 16)
    for (int base = wg * a.per_block; base < min(P, (wg + 1) * a.per_block); base += kHistBatch) {
        uint32_t kv[kHistPerThread];
        uint2 rv[kHistPerThread];
#pragma unroll
        for (int k = 0; k < kHistPerThread; k++) {
            const int i = base + k * 256 + threadIdx.x;
            kv[k] = i < P ? a.key[i] : 0xFFFFFFFFu;
            rv[k] = i < P ? rects[i] : make_uint2(0u, 0u);   // (not written for a culled Gaussian: never looked at)
        }
#pragma unroll
        for (int k = 0; k < kHistPerThread; k++) {
            const uint32_t id = (uint32_t)(base + k * 256 + threadIdx.x);
            if ((int)id < P) {
                const int b = depth_bucket(kv[k], rng);
                const uint32_t slot = slot0[b] + atomicAdd(&rank[b], 1u);   // any order: the bucket is sorted next
                if (b == nb) {
                    a.order[slot] = id;
                    a.offsets[slot] = R;
                } else {
                    a.rec16[slot] = make_uint4(kv[k], id, rv[k].x, rv[k].y);
                }
            }
        }
    }
}

// tiles a rect covers (gauss_math.h preprocess_one: tiles_touched IS this area)
__device__ __forceinline__ uint32_t rect_tiles(uint2 rc)
{
    return ((rc.y & 0xffffu) - (rc.x & 0xffffu)) * ((rc.y >> 16) - (rc.x >> 16));
}
// The sort word of the bucket sorts: (key - bucket minimum) << 40 | id << 16 | position.  Its three fields rest on invariants
// that live elsewhere and are pinned here:
//   * key span of a bucket < 2^24: keys are the bit patterns of positive floats (< 2^31) cut into >= kMinDepthBuckets
//     equal-width buckets (common.h depth_bucket_count) -- and a sub-bucket only narrows its range;
//   * id < 2^24: P >= 2^24 is routed to the generic sort (capi.hip plan_forward: generic_depth_sort);
//   * position < 2^16: a word-sorted bucket holds at most kBucketCap records; a bigger one is first split in global memory
//     into pieces of ~2048 (kGlobalSplitCap / kGlobalSplitMax) and a piece above kBucketCap takes the radix path.
static_assert((1u << 31) / (unsigned)kMinDepthBuckets <= (1u << 24), "a depth bucket's key span must fit the sort word's 24 key bits");
static_assert(kBucketCap <= 65536, "a word-sorted bucket's positions must fit the sort word's 16 position bits");
static_assert(kGlobalSplitCap / kGlobalSplitMax <= (unsigned)kBucketCap, "a piece of a globally split bucket must fit the LDS sort");
__device__ __forceinline__ unsigned long long sort_word(uint32_t key, uint32_t kmin, uint32_t id, uint32_t pos)
{
    return ((unsigned long long)(key - kmin) << 40) | ((unsigned long long)id << 16) | (unsigned long long)pos;
}
__device__ __forceinline__ uint32_t word_id(unsigned long long w) { return (uint32_t)(w >> 16) & 0xFFFFFFu; }
__device__ __forceinline__ uint32_t word_pos(unsigned long long w) { return (uint32_t)w & 0xFFFFu; }

// Bitonic sort of 64 * E words held E per lane by ONE wave (element r * 64 + lane in v[r]): no barriers, no LDS
// arrays -- compare-exchanges with a partner >= 64 elements away are register-local, the others one 64-bit
// shuffle per element.  Ascending; pad with ~0.
template <int E>
__device__ __forceinline__ void wave_bitonic_sort(unsigned long long (&v)[E], int lane)
{
    constexpr int N = 64 * E;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int dr = j / 64;
#pragma unroll
                for (int r = 0; r < E; r++) {
                    if ((r & dr) == 0) {
                        const bool up = ((r * 64) & k) == 0;   // k > j >= 64: decided by the register index alone
                        const unsigned long long x = v[r], y = v[r | dr];
                        const bool sw = (x > y) == up;
                        v[r] = sw ? y : x;
                        v[r | dr] = sw ? x : y;
                    }
                }
            } else {
                const bool lower = (lane & j) == 0;
#pragma unroll
                for (int r = 0; r < E; r++) {
                    const unsigned long long other = __shfl_xor(v[r], j);
                    const bool up = ((r * 64 + lane) & k) == 0;
                    const bool take_min = lower == up;
                    v[r] = ((v[r] < other) == take_min) ? v[r] : other;
                }
            }
        }
    }
}

// full-wave inclusive scan (result in every lane) and total
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)v, off);
        if (lane >= off) v += up;
    }
    return v;
}

// Slow path of the bucket sort: n (key, id) pairs in global memory, sorted by (key, id) by ONE 256-thread workgroup
// with a stable LSD radix sort (8-bit digits; digits on which all pairs agree are skipped -- a bucket's keys share
// their high bits).  Ping-pongs between (k0, v0) and (k1, v1); returns 0 / 1 = which pair of arrays holds the result.
// lds: 256 + 256 + 4 * 256 + 4 words.
__device__ inline int block_radix_sort_pairs(uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, uint32_t n, uint32_t* lds)
{
    uint32_t* s_hist = lds;
    uint32_t* s_base = lds + 256;
    uint32_t(*s_wc)[256] = reinterpret_cast<uint32_t(*)[256]>(lds + 512);
    uint32_t* s_skip = lds + 512 + 4 * 256;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int cur = 0;
    for (int pass = 0; pass < 7; pass++) {   // id bytes 0..2 (P < 2^24), then key bytes 0..3
        const uint32_t* kin = cur ? k1 : k0;
        const uint32_t* vin = cur ? v1 : v0;
        uint32_t* kout = cur ? k0 : k1;
        uint32_t* vout = cur ? v0 : v1;
        const bool on_id = pass < 3;
        const int shift = on_id ? 8 * pass : 8 * (pass - 3);
        s_hist[threadIdx.x] = 0;
        if (threadIdx.x == 0) *s_skip = 0;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 256)
            atomicAdd(&s_hist[((on_id ? vin[i] : kin[i]) >> shift) & 255u], 1u);
        __syncthreads();
        const uint32_t cnt = s_hist[threadIdx.x];
        if (cnt == n) *s_skip = 1;   // every pair has this digit: the pass would not move anything
        const uint32_t incl = block256_inclusive_scan(cnt, s_base);   // s_base doubles as the scan's 4-word scratch
        __syncthreads();
        const bool skip = *s_skip != 0;
        __syncthreads();
        s_base[threadIdx.x] = incl - cnt;
        __syncthreads();
        if (skip) continue;
        for (uint32_t c0 = 0; c0 < n; c0 += 256) {   // chunks in order: stability
            const uint32_t i = c0 + threadIdx.x;
            const bool valid = i < n;
            const uint32_t kk = valid ? kin[i] : 0u, vv = valid ? vin[i] : 0u;
            const uint32_t d = ((on_id ? vv : kk) >> shift) & 255u;
            for (int t = threadIdx.x; t < 4 * 256; t += 256) (&s_wc[0][0])[t] = 0;
            __syncthreads();
            const unsigned long long peers = match_digit<8>(d, valid);
            if (valid && (peers & below) == 0ull) s_wc[w][d] = (uint32_t)__popcll(peers);
            __syncthreads();
            if (valid) {
                uint32_t pos = s_base[d] + (uint32_t)__popcll(peers & below);
                for (int k = 0; k < w; k++) pos += s_wc[k][d];
                kout[pos] = kk;
                vout[pos] = vv;
            }
            __syncthreads();
            s_base[threadIdx.x] += s_wc[0][threadIdx.x] + s_wc[1][threadIdx.x] + s_wc[2][threadIdx.x] + s_wc[3][threadIdx.x];
            __syncthreads();
        }
        __threadfence();
        __syncthreads();
        cur ^= 1;
    }
    return cur;
}

// One wave sorts m <= 64 * E sort words it finds in LDS (src[0 .. m)) in registers and writes the entries' outputs at
// depth ranks out0 .. out0 + m: ids, rects (fetched from the records `rec` by the position the word carries), and the
// inclusive scan of the tile counts on top of `tile_base`.
template <int E>
__device__ __forceinline__ void wave_sort_words(const DepthArgs& a, const unsigned long long* src, uint32_t m,
                                                const uint4* __restrict__ rec, uint32_t out0, uint32_t tile_base, int lane)
{
    unsigned long long v[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const uint32_t i = (uint32_t)(r * 64 + lane);
        v[r] = i < m ? src[i] : ~0ull;
    }
    wave_bitonic_sort<E>(v, lane);
    uint32_t t[E];
    uint2 rc[E];
#pragma unroll
    for (int r = 0; r < E; r++) {
        const bool have = (uint32_t)(r * 64 + lane) < m;
        rc[r] = have ? *reinterpret_cast<const uint2*>(&rec[word_pos(v[r])].z) : make_uint2(0u, 0u);
        t[r] = have ? rect_tiles(rc[r]) : 0u;
    }
    uint2* __restrict__ rect_out = reinterpret_cast<uint2*>(a.rect_sorted) + out0;
    uint32_t run = tile_base;
#pragma unroll
    for (int r = 0; r < E; r++) {
        const uint32_t i = (uint32_t)(r * 64 + lane);
        const uint32_t incl = wave_inclusive_scan(t[r], lane);
        if (i < m) {
            a.order[out0 + i] = word_id(v[r]);
            rect_out[i] = rc[r];
            publish_offset(a, out0 + i, run + incl, t[r]);
        }
        run += (uint32_t)__shfl((int)incl, 63);
    }
}

constexpr int kSplitMax = 32;       // sub-buckets a big bucket is split into at most
constexpr int kSplitTarget = 192;   // ... aiming at this many entries each (a sub-bucket above kWaveSortMax: merge path)

// Sorts the n <= kBucketCap records rec[0 .. n) by (key, id) with the whole workgroup -- the sort words in LDS -- and writes
// ids, rects and the inclusive scan of the tile counts (on top of `tb`) at depth ranks out0 .. out0 + n.
__device__ __forceinline__ void sort_records_lds(const DepthArgs& a, char* smem, const uint4* __restrict__ rec, uint32_t n,
                                                 uint32_t out0, uint32_t tb)
{
    unsigned long long* s = reinterpret_cast<unsigned long long*>(smem);                     // [kBucketCap]
    uint32_t* s_sum = reinterpret_cast<uint32_t*>(smem + (size_t)kBucketCap * 8);            // [256]
    uint32_t* s_radix = s_sum + 256;                                                          // scratch of the split
    uint32_t* __restrict__ order = a.order;
    uint2* __restrict__ rect_out = reinterpret_cast<uint2*>(a.rect_sorted) + out0;
    // smallest key of the bucket: the words hold keys relative to it
    uint32_t kmin = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < n; i += 256) kmin = min(kmin, rec[i].x);
    for (int off = 32; off > 0; off >>= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off));
    __syncthreads();   // s_sum may still be read by a previous bucket of this workgroup
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = kmin;
    __syncthreads();
    kmin = min(min(s_sum[0], s_sum[1]), min(s_sum[2], s_sum[3]));
    __syncthreads();
    // Split path: the bucket's keys are spread over [kmin, kmax]; one more monotone split of that range (LDS counters, a
    // scan over <= 32 numbers) leaves sub-buckets a WAVE sorts in registers -- no bitonic merge levels through LDS with a
    // barrier each (10 for 1024 entries, 33 for 4096).  Large scenes aim for ~1000 Gaussians per depth bucket (the
    // histogram / scatter tables grow with the bucket count) and a real scene's depth distribution fills the buckets of its
    // foreground several times over: there every bucket comes this way.  A sub-bucket above 512 entries (keys piled on a
    // few values): the merge path below sorts the bucket.
    {
        uint32_t* s_cnt = s_radix;                  // [kSplitMax] entries per sub-bucket, then the placement cursor
        uint32_t* s_tiles = s_radix + kSplitMax;    // [kSplitMax] tile counts per sub-bucket
        uint32_t* s_first = s_radix + 2 * kSplitMax;   // [kSplitMax + 1] first position of each sub-bucket
        uint32_t* s_tbase = s_radix + 3 * kSplitMax + 1;   // [kSplitMax] tiles in front of each sub-bucket
        uint32_t* s_flag = s_radix + 4 * kSplitMax + 1;    // kmax, then "a sub-bucket is too big"
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        // (two passes over the bucket's records -- the second finds them in the L1 / L2 -- instead of sixteen records per
        // thread in registers: the kernel's register count is the maximum over every path through it)
        uint32_t kmax = 0u;
        for (uint32_t i = threadIdx.x; i < n; i += 256) kmax = max(kmax, rec[i].x);
        for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off));
        if (threadIdx.x < (uint32_t)kSplitMax) s_cnt[threadIdx.x] = s_tiles[threadIdx.x] = 0u;
        if (lane == 0) s_sum[w] = kmax;
        __syncthreads();
        kmax = max(max(s_sum[0], s_sum[1]), max(s_sum[2], s_sum[3]));
        const uint32_t nsub = min((uint32_t)kSplitMax, (n + (uint32_t)kSplitTarget - 1u) / (uint32_t)kSplitTarget);
        const float scale = (float)nsub / ((float)(kmax - kmin) + 1.0f);   // monotone in the key, as depth_bucket
        auto sub_of = [&](uint32_t key) { return min((uint32_t)((float)(key - kmin) * scale), nsub - 1u); };
#pragma unroll 4
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint4 r = rec[i];
            const uint32_t sb = sub_of(r.x);
            atomicAdd(&s_cnt[sb], 1u);
            atomicAdd(&s_tiles[sb], rect_tiles(make_uint2(r.z, r.w)));
        }
        __syncthreads();
        if (w == 0) {   // scan of the <= 32 sub-bucket sizes and tile sums
            const uint32_t c = lane < (int)nsub ? s_cnt[lane] : 0u, tl = lane < (int)nsub ? s_tiles[lane] : 0u;
            const uint32_t ci = wave_inclusive_scan(c, lane), ti = wave_inclusive_scan(tl, lane);
            const unsigned long long big = __ballot(c > (uint32_t)kWaveSortMax);
            if (lane < (int)nsub) {
                s_first[lane] = ci - c;
                s_tbase[lane] = ti - tl;
                s_cnt[lane] = 0u;   // from here on: the placement cursor
            }
            if (lane == 0) {
                s_first[nsub] = n;
                s_flag[0] = big != 0ull;
            }
        }
        __syncthreads();
        if (s_flag[0] == 0u) {
#pragma unroll 4
            for (uint32_t i = threadIdx.x; i < n; i += 256) {
                const uint2 ki = *reinterpret_cast<const uint2*>(rec + i);
                const uint32_t sb = sub_of(ki.x);
                s[s_first[sb] + atomicAdd(&s_cnt[sb], 1u)] = sort_word(ki.x, kmin, ki.y, i);
            }
            __syncthreads();
            for (uint32_t sb = (uint32_t)w; sb < nsub; sb += 4u) {   // wave-uniform
                const uint32_t f0 = s_first[sb], m = s_first[sb + 1u] - f0;
                if (m == 0u) {
                } else if (m <= 64u) {
                    wave_sort_words<1>(a, s + f0, m, rec, out0 + f0, tb + s_tbase[sb], lane);
                } else if (m <= 128u) {
                    wave_sort_words<2>(a, s + f0, m, rec, out0 + f0, tb + s_tbase[sb], lane);
                } else if (m <= 256u) {
                    wave_sort_words<4>(a, s + f0, m, rec, out0 + f0, tb + s_tbase[sb], lane);
                } else {
                    wave_sort_words<8>(a, s + f0, m, rec, out0 + f0, tb + s_tbase[sb], lane);
                }
            }
            __syncthreads();   // (the window is reused by the workgroup's next big bucket)
            return;
        }
        __syncthreads();
    }
    uint32_t N = 2 * kWaveSortMax;
    while (N < n) N <<= 1;
    // levels k <= kWaveSortMax of the bitonic network: every wave sorts 512-word chunks in registers and parks them in
    // LDS, ascending for even chunks and descending for odd ones (what the network leaves after its k = 512 level)
    {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (uint32_t c = (uint32_t)w; c < N / kWaveSortMax; c += 4) {
            unsigned long long v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t i = c * kWaveSortMax + (uint32_t)(r * 64 + lane);
                if (i < n) {
                    const uint2 ki = *reinterpret_cast<const uint2*>(rec + i);
                    v[r] = sort_word(ki.x, kmin, ki.y, i);
                } else {
                    v[r] = ~0ull;
                }
            }
            wave_bitonic_sort<8>(v, lane);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t i = (uint32_t)(r * 64 + lane);
                s[c * kWaveSortMax + ((c & 1u) ? (uint32_t)kWaveSortMax - 1u - i : i)] = v[r];
            }
        }
    }
    __syncthreads();
    for (uint32_t k = 2 * kWaveSortMax; k <= N; k <<= 1)   // the remaining levels merge through LDS
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < N / 2; t += 256) {
                const uint32_t lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;   // lo has bit j clear
                const unsigned long long x = s[lo], y = s[hi];
                const bool up = (lo & k) == 0;
                if ((x > y) == up) {
                    s[lo] = y;
                    s[hi] = x;
                }
            }
            __syncthreads();
        }
    // scan: each thread owns `per` consecutive sorted entries (two passes over them: no per-thread arrays)
    const uint32_t per = (n + 255) / 256;
    const uint32_t r0 = threadIdx.x * per;
    uint32_t mine = 0;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t r = r0 + k;
        if (r < n) mine += rect_tiles(*reinterpret_cast<const uint2*>(&rec[word_pos(s[r])].z));
    }
    uint32_t run = tb + block256_inclusive_scan(mine, s_sum) - mine;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t r = r0 + k;
        if (r < n) {
            const uint2 rc = *reinterpret_cast<const uint2*>(&rec[word_pos(s[r])].z);
            const uint32_t t = rect_tiles(rc);
            run += t;
            order[out0 + r] = word_id(s[r]);
            rect_out[r] = rc;
            publish_offset(a, out0 + r, run, t);
        }
    }
}

// Workgroup-level sort of ONE bucket b (the big ones, see depth_bucket_group_role), and the inclusive scan of tiles_touched
// in that order on top of the bucket's base.  Up to kBucketCap entries: sort_records_lds.  Above -- a real scene's
// foreground at 2 M Gaussians puts 4000-5000 into each of a few hundred buckets -- the bucket is first split in GLOBAL
// memory (the same monotone split of its key range, records copied into the sub-buckets' places of a second record array)
// and every piece sorted that way; only a bucket whose pieces do not fit either (keys piled on a few values: a
// fronto-parallel plane of splats) takes the radix sort in global memory.
__device__ __forceinline__ void depth_bucket_sort_role(const DepthArgs& a, char* smem, int b)
{
    uint32_t* s_sum = reinterpret_cast<uint32_t*>(smem + (size_t)kBucketCap * 8);            // [256]
    uint32_t* s_radix = s_sum + 256;                                                          // slow path scratch
    const DepthSortScratch* __restrict__ ds = a.ds;
    uint32_t* __restrict__ order = a.order;
    const uint32_t start = ds->start[b], n = ds->start[b + 1] - start;
    if (n == 0) return;
    const uint4* __restrict__ rec = a.rec16 + start;
    uint2* __restrict__ rect_out = reinterpret_cast<uint2*>(a.rect_sorted) + start;
    if (n <= (uint32_t)kBucketCap) {
        sort_records_lds(a, smem, rec, n, start, ds->tile_base[b]);
        return;
    }
    if (n <= kGlobalSplitCap) {
        __shared__ uint32_t g_cnt[kGlobalSplitMax], g_tiles[kGlobalSplitMax], g_first[kGlobalSplitMax + 1], g_tbase[kGlobalSplitMax];
        __shared__ uint32_t g_ok;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint32_t k = rec[i].x;
            kmin = min(kmin, k);
            kmax = max(kmax, k);
        }
        for (int off = 32; off > 0; off >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off));
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off));
        }
        __syncthreads();
        if (lane == 0) {
            s_sum[w] = kmin;
            s_sum[4 + w] = kmax;
        }
        if (threadIdx.x < (uint32_t)kGlobalSplitMax) g_cnt[threadIdx.x] = g_tiles[threadIdx.x] = 0u;
        __syncthreads();
        kmin = min(min(s_sum[0], s_sum[1]), min(s_sum[2], s_sum[3]));
        kmax = max(max(s_sum[4], s_sum[5]), max(s_sum[6], s_sum[7]));
        const uint32_t nsub = min((uint32_t)kGlobalSplitMax, (n + 2047u) / 2048u);
        const float scale = (float)nsub / ((float)(kmax - kmin) + 1.0f);   // monotone in the key, as depth_bucket
        auto sub_of = [&](uint32_t key) { return min((uint32_t)((float)(key - kmin) * scale), nsub - 1u); };
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint4 r = rec[i];
            const uint32_t sb = sub_of(r.x);
            atomicAdd(&g_cnt[sb], 1u);
            atomicAdd(&g_tiles[sb], rect_tiles(make_uint2(r.z, r.w)));
        }
        __syncthreads();
        if (w == 0) {
            const uint32_t c = lane < (int)nsub ? g_cnt[lane] : 0u, tl = lane < (int)nsub ? g_tiles[lane] : 0u;
            const uint32_t ci = wave_inclusive_scan(c, lane), ti = wave_inclusive_scan(tl, lane);
            const unsigned long long big = __ballot(c > (uint32_t)kBucketCap);
            if (lane < (int)nsub) {
                g_first[lane] = ci - c;
                g_tbase[lane] = ti - tl;
                g_cnt[lane] = 0u;   // from here on: the placement cursor
            }
            if (lane == 0) {
                g_first[nsub] = n;
                g_ok = big == 0ull;
            }
        }
        __syncthreads();
        if (g_ok) {
            uint4* __restrict__ dst = a.rec16_b + start;
            for (uint32_t i = threadIdx.x; i < n; i += 256) {
                const uint4 r = rec[i];
                const uint32_t sb = sub_of(r.x);
                dst[g_first[sb] + atomicAdd(&g_cnt[sb], 1u)] = r;
            }
            __threadfence();
            __syncthreads();
            const uint32_t tb = ds->tile_base[b];
            for (uint32_t sb = 0; sb < nsub; sb++) {   // workgroup-uniform
                const uint32_t f0 = g_first[sb], m = g_first[sb + 1u] - f0;
                if (m) sort_records_lds(a, smem, dst + f0, m, start + f0, tb + g_tbase[sb]);
                __syncthreads();
            }
            return;
        }
    }
    {
        // the bucket fits neither: its (key, id) pairs are copied out of the records, sorted by a radix sort in global memory,
        // then the scan in strides of 256 (tile counts and rects gathered by id: the slow path)
        for (uint32_t i = threadIdx.x; i < n; i += 256) {
            const uint4 r = rec[i];
            a.key_sorted[start + i] = r.x;
            a.bucket_id[start + i] = r.y;
        }
        __threadfence();
        __syncthreads();
        const int cur = block_radix_sort_pairs(a.key_sorted + start, a.bucket_id + start, a.ovf_key + start,
                                               a.ovf_id + start, n, s_radix);
        const uint32_t* ids = (cur ? a.ovf_id : a.bucket_id) + start;
        const uint2* __restrict__ rects = reinterpret_cast<const uint2*>(a.rect);
        uint32_t run = ds->tile_base[b];
        for (uint32_t c0 = 0; c0 < n; c0 += 256) {
            const uint32_t r = c0 + threadIdx.x;
            const uint32_t id = r < n ? ids[r] : 0u;
            const uint2 rc = r < n ? rects[id] : make_uint2(0u, 0u);
            const uint32_t t = r < n ? rect_tiles(rc) : 0u;
            const uint32_t incl = block256_inclusive_scan(t, s_sum);
            if (r < n) {
                order[start + r] = id;
                rect_out[r] = rc;
                publish_offset(a, start + r, run + incl, t);
            }
            __syncthreads();
            if (threadIdx.x == 255) s_sum[4] = incl;
            __syncthreads();
            run += s_sum[4];
        }
    }
}

// One wave sorts bucket b (n <= 64 * E pairs) in registers and scans tiles_touched in sorted order.  (Staging the records'
// rect halves in LDS instead of fetching them again by position measured slower: depth sort + colour 0.108 vs 0.096 ms at
// 500 k Gaussians, 0.319 vs 0.291 at 2 M, 0.867 vs 0.828 at 6 M -- the bucket's 16 KB are still in the L1 / L2.)
template <int E>
__device__ __forceinline__ void wave_sort_bucket(const DepthArgs& a, uint32_t start, uint32_t n, uint32_t tile_base,
                                                 int lane)
{
    const uint4* __restrict__ rec = a.rec16 + start;
    unsigned long long v[E];
    uint2 ki[E];
    uint32_t kmin = 0xFFFFFFFFu;
#pragma unroll
    for (int r = 0; r < E; r++) {
        const uint32_t i = (uint32_t)(r * 64 + lane);
        ki[r] = i < n ? *reinterpret_cast<const uint2*>(rec + i) : make_uint2(0xFFFFFFFFu, 0u);
        kmin = min(kmin, ki[r].x);
    }
    for (int off = 32; off > 0; off >>= 1) kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off));
#pragma unroll
    for (int r = 0; r < E; r++) {
        const uint32_t i = (uint32_t)(r * 64 + lane);
        v[r] = i < n ? sort_word(ki[r].x, kmin, ki[r].y, i) : ~0ull;
    }
    wave_bitonic_sort<E>(v, lane);
    uint32_t t[E];
    uint2 rc[E];
#pragma unroll
    for (int r = 0; r < E; r++) {   // the record of the entry that sorted here: its rect (-> tile count) is the other half
        const bool have = (uint32_t)(r * 64 + lane) < n;
        rc[r] = have ? *reinterpret_cast<const uint2*>(&rec[word_pos(v[r])].z) : make_uint2(0u, 0u);
        t[r] = have ? rect_tiles(rc[r]) : 0u;
    }
    uint2* __restrict__ rect_out = reinterpret_cast<uint2*>(a.rect_sorted) + start;
    uint32_t run = tile_base;
#pragma unroll
    for (int r = 0; r < E; r++) {
        const uint32_t i = (uint32_t)(r * 64 + lane);
        const uint32_t incl = wave_inclusive_scan(t[r], lane);
        if (i < n) {
            a.order[start + i] = word_id(v[r]);
            rect_out[i] = rc[r];
            publish_offset(a, start + i, run + incl, t[r]);
        }
        run += (uint32_t)__shfl((int)incl, 63);
    }
}


// (4) workgroup wg of nb / 4: each of its four waves sorts one bucket's (key << 32 | id) words in registers and scans
// tiles_touched in that order on top of the bucket's base (rasterizer_impl.cu:441 InclusiveSum, fused); buckets above
// kWaveSortMax pairs are then sorted by the whole workgroup in LDS (<= kBucketCap) or in global memory.
__device__ __forceinline__ void depth_bucket_group_role(const DepthArgs& a, char* smem, int wg, int n_groups)
{
    const DepthSortScratch* __restrict__ ds = a.ds;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        const int b = wg * kBucketsPerGroup + w;
        if (b < a.nb) {
            const uint32_t start = ds->start[b], n = ds->start[b + 1] - start, tb = ds->tile_base[b];
            if (n == 0) {
            } else if (n <= 64) {
                wave_sort_bucket<1>(a, start, n, tb, lane);
            } else if (n <= 128) {
                wave_sort_bucket<2>(a, start, n, tb, lane);
            } else if (n <= 256) {
                wave_sort_bucket<4>(a, start, n, tb, lane);
            } else if (n <= (uint32_t)kWaveSortMax) {
                wave_sort_bucket<8>(a, start, n, tb, lane);
            }
        }
    }
    // The big buckets, all four waves together -- dealt out from the compact list the column scan made of them, workgroup g
    // taking entries g, g + groups, ...: with each workgroup sorting the big ones among ITS four buckets, the few hundred
    // neighbouring buckets that hold a real scene's foreground kept ~70 workgroups busy for four sorts each while the others
    // had long left (depth sort + colour 0.156 -> 0.135 ms on the clustered 500 k scene).  (Handing them out through a
    // shared cursor instead was measured too: device-scope atomics on one address cost ~20 ns each -- 0.72 -> 0.84 ms at
    // 6 M Gaussians, where all 8192 buckets are big.)
    const uint32_t n_big = ds->big_count;
    for (uint32_t i = (uint32_t)wg; i < n_big; i += (uint32_t)n_groups) {
        __syncthreads();
        depth_bucket_sort_role(a, smem, (int)ds->big_list[i]);
    }
}

}  // namespace r3
