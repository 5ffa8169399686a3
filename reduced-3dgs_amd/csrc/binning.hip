// binning.hip -- tile binning: which Gaussians touch which 16x16 tile, in front-to-back order.
//
// Replaces rasterizer_impl.cu:441 (InclusiveSum), :78-119 duplicateWithKeys, :465-473 the 64-bit
// (tile|depth) radix sort and :124-146 identifyTileRanges of
// /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer, with the SAME resulting order
// (tile-major; inside a tile ascending depth bits, ties by ascending Gaussian index) but far less sort
// traffic, which is what bounds this stage on MI355X:
//
//   reference: sort R (u64 key, u32 value) pairs on 32+log2(tiles) bits  -> ~6 passes x 24 B x R
//   here:      1. stable sort the P Gaussians once by their 32 depth bits (P << R): one bucketing pass + an LDS sort
//              2. inclusive scan of tiles_touched in that depth order (fused into 1.)
//              3. emit the R pairs in depth order as words (tile | Gaussian id), balanced over OUTPUT slots
//              4. stable LSD radix sort of the words on the log2(tiles) tile bits only (2 passes x 8 or 12 B x R:
//                 32-bit words, or a 16-bit key array + a 32-bit id array above 2^19 Gaussians)
//   Stability of both sorts + ascending-id input order reproduces the reference's tie-break exactly.
//
// Every kernel takes a pointer into the pass's device argument block (common.h) and reads the pair count from the
// device header: nothing here depends on a host copy of num_rendered, grids are sized by the pair reservation.
// rocPRIM is used only by the generic depth sort (fallback for P >= 2^24 and R3DGS_DEPTH_SORT=generic).
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include <rocprim/rocprim.hpp>

#include "depth_sort.h"

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

void run_generic_depth_sort(int P, GeomState& g, hipStream_t s)
{
    size_t bytes = g.temp_bytes;
    R3_HIP(rocprim::radix_sort_pairs(g.temp, bytes, g.depth_key, g.key_sorted, rocprim::counting_iterator<uint32_t>(0),
                                     g.order, (size_t)P, 0, 32, s));
    bytes = g.temp_bytes;
    auto it = rocprim::make_transform_iterator(g.order, GatherTiles{g.tiles});
    R3_HIP(rocprim::inclusive_scan(g.temp, bytes, it, g.offsets, (size_t)P, rocprim::plus<uint32_t>(), s));
}

// ---- per-view header ----------------------------------------------------------------------------------------------
// One workgroup turns the preprocess partials into the device header (depth_sort.h write_header).  Used by the
// exact-size path, whose host waits for num_rendered between the geometry kernel and everything else; the asynchronous
// path lets the histogram workgroups do it (DepthArgs::fuse_header).
__global__ __launch_bounds__(1024) void header_reduce_kernel(const HeaderArgs* __restrict__ ap)
{
    __shared__ uint32_t s_red[16][6];
    const HeaderArgs a = *ap;
    const PrePartial all = reduce_partials(a.parts, a.n_parts, s_red);
    if (threadIdx.x == 0) write_header(a, all);
}

void issue_header_reduce(const HeaderArgs* a, hipStream_t s)
{
    hipLaunchKernelGGL(header_reduce_kernel, dim3(1), dim3(1024), 0, s, a);
}

// ---- packed pair words -------------------------------------------------------------------------------------------
PairLayout pair_layout(int P, size_t n_tiles)
{
    static const int forced = [] {   // R3DGS_TILE_SORT=wide | split: that layout for every shape it can hold (A/B runs, tests)
        const char* v = getenv("R3DGS_TILE_SORT");
        return !v ? 0 : std::string(v) == "wide" ? 1 : std::string(v) == "split" ? 2 : 0;
    }();
    PairLayout l;
    l.tile_bits = (int)higher_msb((uint32_t)n_tiles);
    const int rb = (int)higher_msb((uint32_t)P);
    const bool fits16 = n_tiles <= 65536;
    if (forced == 1 || (forced == 2 && !fits16))
        l.wide = 1;
    else if (forced == 2)
        l.wide = 2;
    else
        l.wide = l.tile_bits + rb > 32 ? (fits16 ? 2 : 1) : 0;
    l.rank_bits = l.wide ? 32 : rb;
    l.digit_bits = l.tile_bits <= 14 ? 7 : 8;
    l.passes = (l.tile_bits + l.digit_bits - 1) / l.digit_bits;
    if (l.passes < 2) l.passes = 2;   // the first digit's counts come out of the emission kernel; keep one shape
    if (l.passes > kMaxRadixPasses) throw Error("more than 2^24 tiles are not supported");
    return l;
}


// How the pair words live in memory.  In registers a word is always  tile << rank_bits | Gaussian id.
struct IoNarrow {   // one 32-bit array
    typedef uint32_t Reg;
    static constexpr bool kIdsApart = false;
    static __device__ __forceinline__ Reg load(const char* buf, uint32_t, uint32_t i) { return reinterpret_cast<const uint32_t*>(buf)[i]; }
    static __device__ __forceinline__ void store(char* buf, uint32_t*, uint32_t, uint32_t i, Reg w) { reinterpret_cast<uint32_t*>(buf)[i] = w; }
    static __device__ __forceinline__ uint32_t tile_at(const char* buf, uint32_t cap, uint32_t i, int rank_bits) { return load(buf, cap, i) >> rank_bits; }
};
struct IoWide {     // one 64-bit array, rank_bits = 32
    typedef unsigned long long Reg;
    static constexpr bool kIdsApart = false;
    static __device__ __forceinline__ Reg load(const char* buf, uint32_t, uint32_t i) { return reinterpret_cast<const Reg*>(buf)[i]; }
    static __device__ __forceinline__ void store(char* buf, uint32_t*, uint32_t, uint32_t i, Reg w) { reinterpret_cast<Reg*>(buf)[i] = w; }
    static __device__ __forceinline__ uint32_t tile_at(const char* buf, uint32_t cap, uint32_t i, int) { return (uint32_t)(load(buf, cap, i) >> 32); }
};
struct IoSplit {    // [cap] 32-bit ids, then [cap] 16-bit tile keys; rank_bits = 32.  6 bytes per pair and pass instead of 8,
                    // the last pass drops its ids straight into point_list, and the emitted ids ARE the backward's run keys
    typedef unsigned long long Reg;
    static constexpr bool kIdsApart = true;
    static __device__ __forceinline__ const unsigned short* keys(const char* buf, uint32_t cap) { return reinterpret_cast<const unsigned short*>(buf + 4 * (size_t)cap); }
    static __device__ __forceinline__ Reg load(const char* buf, uint32_t cap, uint32_t i)
    {
        return ((Reg)keys(buf, cap)[i] << 32) | reinterpret_cast<const uint32_t*>(buf)[i];
    }
    static __device__ __forceinline__ void store(char* buf, uint32_t* ids_out, uint32_t cap, uint32_t i, Reg w)
    {
        reinterpret_cast<unsigned short*>(buf + 4 * (size_t)cap)[i] = (unsigned short)(w >> 32);
        (ids_out ? ids_out : reinterpret_cast<uint32_t*>(buf))[i] = (uint32_t)w;
    }
    static __device__ __forceinline__ uint32_t tile_at(const char* buf, uint32_t cap, uint32_t i, int) { return keys(buf, cap)[i]; }
};

// Pair emission, balanced over OUTPUT positions.  The reference loops one thread over all tiles of its
// Gaussian (rasterizer_impl.cu:106-117); in depth order the nearest -- largest -- splats sit next to each
// other, so any per-Gaussian (or per-64-Gaussian) work split has a tail: the first GPU version of this
// kernel spent 234 us waiting for its first few waves.  Here every workgroup owns kEmitPerBlock
// consecutive output slots: two waves locate the slot range's first/last source Gaussian by a 64-ary
// search in the inclusive scan, the block stages that <= kEmitPerBlock+1 long slice (end offset, id,
// tile rect) in LDS, and every lane resolves its slots with an LDS binary search.  Stores are coalesced.
// With num_rendered above the pass's reservation the slots stop at the reservation: emission is in depth order, so
// what is dropped are the FARTHEST pairs.
constexpr int kEmitPerBlock = kRadixBlock;
constexpr int kEmitSlice = kEmitPerBlock + 8;
constexpr int kEmitStage = 512;   // Gaussians whose start / id / rect are staged in LDS per round

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

// One logical block = kEmitPerBlock consecutive output slots.  The pair-sized kernels are launched with about as many
// workgroups as the pair count of recent passes needs (FwdPlan::grid_pairs) and stride over the logical blocks: sizing
// the grid by the reservation left a third of the workgroups with nothing to do but three dependent loads to find that
// out (+10 us on the backward's segmented sum alone).
template <class IO>
__device__ __forceinline__ void emit_pairs_block(const EmitArgs& a, uint32_t R, uint32_t blk)
{
    __shared__ uint32_t s_hist[kMaxRadixBins];   // first radix digit of the tile sort, counted while emitting
    __shared__ uint32_t s_start[kEmitStage];     // staged Gaussians of the current round: first pair, id, tile rect
    __shared__ uint32_t s_id[kEmitStage];
    __shared__ ushort4 s_rect[kEmitStage];
    __shared__ unsigned short s_owner[kEmitPerBlock];   // slice index of the slot's Gaussian (< kEmitSlice <= 65535)
    __shared__ uint32_t s_wmax[4];
    __shared__ uint32_t s_j[2];
    __shared__ uint32_t s_start0;
    const int bins = 1 << a.digit_bits;
    const uint32_t pos0 = blk * (uint32_t)kEmitPerBlock;
    if ((int)threadIdx.x < bins) s_hist[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < kEmitPerBlock; i += 256) s_owner[i] = 0;
    const uint32_t* __restrict__ offsets = a.offsets;
    const uint32_t pos1 = min(R, pos0 + (uint32_t)kEmitPerBlock);  // exclusive
    if (threadIdx.x < 128) {   // wave 0 locates the first slot's Gaussian, wave 1 the last slot's
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const uint32_t* __restrict__ bf = a.block_first;
        const bool last_blk = pos1 == R;   // the slice ends with the pass's last pair, not at a workgroup border
        uint32_t j;
        if (bf && w == 0) {
            j = bf[blk];   // noted by the depth sort's scan
        } else if (bf && !last_blk) {
            // the Gaussian holding pair pos1 (the next workgroup's first) holds pair pos1 - 1 too unless it starts there
            const uint32_t g = bf[blk + 1];
            j = (g > 0u && offsets[g - 1] == pos1) ? g - 1u : g;
        } else if (bf && R == a.hdr->num_rendered) {
            j = a.hdr->binned - 1u;   // the last Gaussian in depth order that owns pairs
        } else {
            j = upper_bound_wave(offsets, (uint32_t)a.P, w == 0 ? pos0 : pos1 - 1, lane);
        }
        if (lane == 0) {
            s_j[w] = j;
            if (w == 0) s_start0 = j == 0 ? 0u : offsets[j - 1];
        }
    }
    __syncthreads();
    const uint32_t j_lo = s_j[0], j_hi = s_j[1];
    // every Gaussian inside the slice owns >= 1 slot (culled ones sort to the very end), so n <= slots + 1
    const uint32_t n = min(j_hi - j_lo + 1u, (uint32_t)kEmitSlice);
    const uint32_t start0 = s_start0;
    // The Gaussians' data is staged kEmitStage at a time (a block of 2048 slots typically holds ~200 Gaussians; staging
    // for the worst case of one per slot cost 33 KB of LDS and two thirds of the kernel's resident workgroups): every
    // round emits the slots whose Gaussian it holds.  The first round is staged before anything else needs it.
    auto stage = [&](uint32_t r0, uint32_t nr) {
        for (uint32_t k = threadIdx.x; k < nr; k += 256) {
            const uint32_t g = j_lo + r0 + k, id = a.order[g];
            s_start[k] = r0 + k == 0 ? start0 : offsets[g - 1];
            s_id[k] = id;
            s_rect[k] = a.rect_sorted ? a.rect_sorted[g] : a.rect[id];   // depth-ordered copy: a coalesced read, no gather
        }
    };
    const uint32_t nr0 = min(n, (uint32_t)kEmitStage);
    stage(0u, nr0);
    __syncthreads();
    // record where each Gaussian's pairs begin (blocks sharing a Gaussian write the same value), and mark the slot it
    // starts at with its slice index: a running maximum over the slots then names every slot's Gaussian (a binary
    // search per slot over the slice's end offsets cost ~11 dependent LDS reads per pair instead)
    for (uint32_t k = threadIdx.x; k < n; k += 256) {
        const uint32_t start = k < nr0 ? s_start[k] : offsets[j_lo + k - 1];
        a.rec[k < nr0 ? s_id[k] : a.order[j_lo + k]].pair_start = start;
        if (start >= pos0) s_owner[start - pos0] = (unsigned short)k;   // start < pos1: the slice ends at the last slot's Gaussian
    }
    __syncthreads();
    {
        constexpr int E = kEmitPerBlock / 256;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        uint32_t v[E];
#pragma unroll
        for (int i = 0; i < E; i++) v[i] = s_owner[threadIdx.x * E + i];
#pragma unroll
        for (int i = 1; i < E; i++) v[i] = max(v[i], v[i - 1]);
        uint32_t incl = v[E - 1];
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl = max(incl, up);
        }
        if (lane == 63) s_wmax[w] = incl;
        uint32_t before = (uint32_t)__shfl_up((int)incl, 1);
        if (lane == 0) before = 0;
        __syncthreads();
        for (int q = 0; q < w; q++) before = max(before, s_wmax[q]);
#pragma unroll
        for (int i = 0; i < E; i++) s_owner[threadIdx.x * E + i] = (unsigned short)max(v[i], before);
    }
    typedef typename IO::Reg Word;
    const int rank_bits = a.rank_bits;
    for (uint32_t r0 = 0; r0 < n; r0 += (uint32_t)kEmitStage) {
        const uint32_t nr = min(n - r0, (uint32_t)kEmitStage);
        __syncthreads();   // s_owner complete / the previous round's staging consumed
        if (r0 > 0) {
            stage(r0, nr);
            __syncthreads();
        }
#pragma unroll
        for (int e = 0; e < kEmitPerBlock / 256; e++) {
            const uint32_t pos = pos0 + threadIdx.x + (uint32_t)e * 256u;
            if (pos < pos1) {
                const uint32_t lo = (uint32_t)s_owner[pos - pos0] - r0;   // the staged Gaussian this slot belongs to
                if (lo < nr) {
                    const uint32_t local = pos - s_start[lo];
                    const ushort4 r = s_rect[lo];
                    const uint32_t w = (uint32_t)r.z - (uint32_t)r.x;
                    const uint32_t ty = local / w, tx = local - ty * w;  // row-major (y, x), rasterizer_impl.cu:106-117
                    const uint32_t tile = ((uint32_t)r.y + ty) * (uint32_t)a.gx + (uint32_t)r.x + tx;
                    IO::store(a.words_out, nullptr, a.cap, pos, ((Word)tile << rank_bits) | (Word)s_id[lo]);   // tile | Gaussian id
                    if (a.pair_rank) a.pair_rank[pos] = s_id[lo];
                    atomicAdd(&s_hist[tile & (uint32_t)(bins - 1)], 1u);
                }
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < bins) a.radix_rows[(size_t)threadIdx.x * a.row_stride + blk] = s_hist[threadIdx.x];
}

template <class IO>
__global__ __launch_bounds__(256) void emit_pairs_kernel(const EmitArgs* __restrict__ ap)
{
    const EmitArgs a = *ap;
    const uint32_t R = a.hdr->num_pairs;
    // the tile ranges start out as (0, 0) (rasterizer_impl.cu:475 memset): cleared here, ahead of the sort, instead of
    // by a fill of its own on the stream
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < a.n_tiles; t += gridDim.x * 256u) a.ranges[t] = make_uint2(0u, 0u);
    for (uint32_t blk = blockIdx.x; blk * (uint32_t)kEmitPerBlock < R; blk += gridDim.x) {
        emit_pairs_block<IO>(a, R, blk);
        __syncthreads();
    }
}

// ---- tile sort of the packed pair words: LSD radix -------------------------------------------------------------------
// rocPRIM's onesweep spends more time around its two passes (per-pass fills of the look-back state, histogram and scan
// kernels: ~65 us of launches and gaps for ~48 us of sorting at R = 3.6 M) than in them.  Here a pass is: per-workgroup
// digit counts (for the first digit they come out of the emission kernel), one workgroup per digit scanning its row
// of counts, and a scatter that ranks its kRadixBlock keys stably -- wave-level match by one ballot per digit bit per round,
// running per-wave digit counts in LDS -- no fills, no look-back chain.  Stable, so the passes give the tile-major
// order with the emission (depth) order preserved inside a tile.
template <class IO>
__device__ __forceinline__ void radix_hist_block(const RadixArgs& a, uint32_t R, uint32_t blk)
{
    __shared__ uint32_t s_hist[kMaxRadixBins];
    const uint32_t base = blk * (uint32_t)kRadixBlock;
    const int bins = 1 << a.digit_bits;
    if ((int)threadIdx.x < bins) s_hist[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRadixBlock / 256; k++) {
        const uint32_t i = base + k * 256u + threadIdx.x;
        if (i < R) atomicAdd(&s_hist[(IO::tile_at(a.in, a.cap, i, a.rank_bits) >> a.tile_shift) & (uint32_t)(bins - 1)], 1u);
    }
    __syncthreads();
    if ((int)threadIdx.x < bins) a.rows[(size_t)threadIdx.x * a.row_stride + blk] = s_hist[threadIdx.x];
}

template <class IO>
__global__ __launch_bounds__(256) void radix_hist_kernel(const RadixArgs* __restrict__ ap)
{
    const RadixArgs a = *ap;
    const uint32_t R = a.hdr->num_pairs;
    for (uint32_t blk = blockIdx.x; blk * (uint32_t)kRadixBlock < R; blk += gridDim.x) {
        radix_hist_block<IO>(a, R, blk);
        __syncthreads();
    }
}

// one workgroup per digit: exclusive scan of that digit's per-workgroup counts, and the digit total
__global__ __launch_bounds__(256) void radix_digit_scan_kernel(const RadixArgs* __restrict__ ap)
{
    __shared__ uint32_t s_sum[256];
    const RadixArgs a = *ap;
    const uint32_t nb = (a.hdr->num_pairs + (uint32_t)kRadixBlock - 1u) / (uint32_t)kRadixBlock;   // live workgroups
    const uint32_t* row = a.rows + (size_t)blockIdx.x * a.row_stride;
    uint32_t* out = a.base + (size_t)blockIdx.x * a.row_stride;
    // 1024 counts per trip, four consecutive ones per thread as one 16-byte access (rows are padded to a multiple of
    // four counts and 16-byte aligned, capi.hip); counts of workgroups that do not exist read as zero
    uint32_t run = 0;
    for (uint32_t c0 = 0; c0 < nb; c0 += 1024u) {
        const uint32_t i = c0 + threadIdx.x * 4u;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < nb) v = *reinterpret_cast<const uint4*>(row + i);
        if (i + 1 >= nb) v.y = 0u;
        if (i + 2 >= nb) v.z = 0u;
        if (i + 3 >= nb) v.w = 0u;
        const uint32_t mine = v.x + v.y + v.z + v.w;
        const uint32_t incl = block256_inclusive_scan(mine, s_sum);
        const uint32_t b0 = run + incl - mine;
        if (i < nb) *reinterpret_cast<uint4*>(out + i) = make_uint4(b0, b0 + v.x, b0 + v.x + v.y, b0 + v.x + v.y + v.z);
        run += s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];   // stays valid: the next scan starts with a barrier
    }
    if (threadIdx.x == 0) a.total[blockIdx.x] = run;
}

// (Materialising point_list[pos] = order[rank] here in the last pass was measured: the dependent gather lengthens this
// kernel by 19 us and saves 13 us in tile_ranges_kernel, so it stays there.)
template <class IO, int BITS>
__device__ __forceinline__ void radix_scatter_block(const RadixArgs& a, uint32_t R, uint32_t blk_id)
{
    constexpr int kWaves = 4, kRounds = kRadixBlock / 256, kBins = 1 << BITS;
    __shared__ uint32_t s_wcount[kWaves][kBins];   // running per-wave digit counts
    __shared__ uint32_t s_start[kBins];            // exclusive scan of the digit totals
    __shared__ uint32_t s_off[kWaves][kBins];      // digit start + workgroup base + waves below
    __shared__ uint32_t s_delta[kBins];
    __shared__ typename IO::Reg s_keys[kRadixBlock];
    typedef typename IO::Reg Word;
    const int shift = a.shift;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < kWaves * kBins; t += 256) (&s_wcount[0][0])[t] = 0;
    if (w == 0) {   // kBins totals, kBins/64 per lane, scanned with shuffles
        constexpr int kPer = kBins / 64;
        uint32_t v[kPer], sum = 0;
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            v[k] = a.total[kPer * lane + k];
            sum += v[k];
        }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += up;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            s_start[kPer * lane + k] = run;
            run += v[k];
        }
    }
    __syncthreads();
    const uint32_t blk = blk_id * (uint32_t)kRadixBlock + (uint32_t)w * (kRadixBlock / kWaves);
    Word key[kRounds];
    uint32_t lrank[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const uint32_t i = blk + r * 64u + lane;
        key[r] = i < R ? IO::load(a.in, a.cap, i) : (Word)0;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const bool valid = blk + r * 64u + lane < R;
        const uint32_t d = (uint32_t)(key[r] >> shift) & (uint32_t)(kBins - 1);
        const unsigned long long peers = match_digit<BITS>(d, valid);   // lanes of this round holding the same digit
        const uint32_t seen = s_wcount[w][d];          // same-digit keys of this wave in earlier rounds
        lrank[r] = seen + (uint32_t)__popcll(peers & below);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & below) == 0ull) s_wcount[w][d] = seen + (uint32_t)__popcll(peers);   // group leader
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // The keys leave through LDS in digit-major order: consecutive lanes then store to consecutive addresses inside a
    // digit's run (kRadixBlock / kBins keys long on average), instead of every lane to a slot of its own.
    if (w == 0) {   // exclusive scan of the workgroup's digit totals -> where each digit's run starts in the staging array
        constexpr int kPer = kBins / 64;
        uint32_t v[kPer], sum = 0;
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const int d = kPer * lane + k;
            v[k] = s_wcount[0][d] + s_wcount[1][d] + s_wcount[2][d] + s_wcount[3][d];
            sum += v[k];
        }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += up;
        }
        uint32_t run = incl - sum;
#pragma unroll
        for (int k = 0; k < kPer; k++) {
            const int d = kPer * lane + k;
            // global slot of staging slot i of digit d = i + s_delta[d] (wrapping arithmetic)
            s_delta[d] = s_start[d] + a.base[(size_t)d * a.row_stride + blk_id] - run;
            uint32_t r2 = run;
#pragma unroll
            for (int q = 0; q < kWaves; q++) {
                s_off[q][d] = r2;
                r2 += s_wcount[q][d];
            }
            run += v[k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        if (blk + r * 64u + lane < R) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & (uint32_t)(kBins - 1);
            s_keys[s_off[w][d] + lrank[r]] = key[r];
        }
    }
    __syncthreads();
    const uint32_t base0 = blk_id * (uint32_t)kRadixBlock;
    const uint32_t nvalid = min((uint32_t)kRadixBlock, R - base0);
#pragma unroll
    for (int r = 0; r < kRounds; r++) {
        const uint32_t i = (uint32_t)r * 256u + threadIdx.x;
        if (i < nvalid) {
            const Word k = s_keys[i];
            const uint32_t d = (uint32_t)(k >> shift) & (uint32_t)(kBins - 1);
            IO::store(a.out, a.ids_out, a.cap, i + s_delta[d], k);
        }
    }
}

template <class IO, int BITS>
__global__ __launch_bounds__(256) void radix_scatter_kernel(const RadixArgs* __restrict__ ap)
{
    const RadixArgs a = *ap;
    const uint32_t R = a.hdr->num_pairs;
    for (uint32_t blk = blockIdx.x; blk * (uint32_t)kRadixBlock < R; blk += gridDim.x) {
        radix_scatter_block<IO, BITS>(a, R, blk);
        __syncthreads();
    }
}

// rasterizer_impl.cu:124-146 identifyTileRanges on the sorted words; the Gaussian ids (the low bits of the words)
// are split off into point_list here too.
template <class IO>
__global__ __launch_bounds__(256) void tile_ranges_kernel(const RangesArgs* __restrict__ ap)
{
    typedef typename IO::Reg Word;
    const RangesArgs a = *ap;
    const int R = (int)a.hdr->num_pairs;
  for (int blk = (int)blockIdx.x; blk * 1024 < R; blk += (int)gridDim.x) {
    const int i0 = (blk * 256 + threadIdx.x) * 4;   // four consecutive entries per thread (arrays 256-B aligned)
    if (i0 >= R) continue;
    unsigned char* __restrict__ pair_flag = a.pair_flag;
    uint32_t* __restrict__ point_list = a.point_list;
    uint2* ranges = a.ranges;
    const int rank_bits = a.rank_bits;
    // the backward's "row written" flags start out zero (it puts the ones it consumed back itself)
    if (i0 + 4 <= R)
        *reinterpret_cast<uint32_t*>(pair_flag + i0) = 0u;
    else
        for (int k = i0; k < R; k++) pair_flag[k] = 0;
    const int n = min(4, R - i0);
    uint32_t tile[4];
    if (IO::kIdsApart) {   // the last pass wrote the ids into point_list itself; only the 16-bit keys are read here
        for (int k = 0; k < 4; k++) tile[k] = k < n ? IO::tile_at(a.sorted, a.cap, (uint32_t)(i0 + k), rank_bits) : 0u;
    } else {
        Word key[4];
        for (int k = 0; k < 4; k++) key[k] = k < n ? IO::load(a.sorted, a.cap, (uint32_t)(i0 + k)) : (Word)0;
        const Word mask = (((Word)1) << rank_bits) - 1;
        uint32_t id[4];
        for (int k = 0; k < 4; k++) {
            id[k] = (uint32_t)(key[k] & mask);
            tile[k] = (uint32_t)(key[k] >> rank_bits);
        }
        if (n == 4)
            *reinterpret_cast<uint4*>(point_list + i0) = make_uint4(id[0], id[1], id[2], id[3]);
        else
            for (int k = 0; k < n; k++) point_list[i0 + k] = id[k];
    }
    uint32_t prev = i0 ? IO::tile_at(a.sorted, a.cap, (uint32_t)(i0 - 1), rank_bits) : 0xFFFFFFFFu;
    for (int k = 0; k < n; k++) {
        const uint32_t cur = tile[k];
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
}

template <class IO>
static void issue_tile_binning_t(const FwdPlan& p, const FwdPassArgs* a, hipStream_t s)
{
    const PairLayout& l = p.layout;
    const uint32_t nbk = (p.grid_pairs + (uint32_t)kRadixBlock - 1u) / (uint32_t)kRadixBlock;   // <= row_stride; blocks stride
    const int bins = 1 << l.digit_bits;
    hipLaunchKernelGGL(emit_pairs_kernel<IO>, dim3(nbk), dim3(256), 0, s, &a->emit);
    for (int k = 0; k < l.passes; k++) {
        const RadixArgs* ra = &a->radix[k];
        if (k > 0) hipLaunchKernelGGL(radix_hist_kernel<IO>, dim3(nbk), dim3(256), 0, s, ra);
        hipLaunchKernelGGL(radix_digit_scan_kernel, dim3(bins), dim3(256), 0, s, ra);
        if (l.digit_bits == 7)
            hipLaunchKernelGGL((radix_scatter_kernel<IO, 7>), dim3(nbk), dim3(256), 0, s, ra);
        else
            hipLaunchKernelGGL((radix_scatter_kernel<IO, 8>), dim3(nbk), dim3(256), 0, s, ra);
    }
    hipLaunchKernelGGL(tile_ranges_kernel<IO>, dim3((p.grid_pairs + 1023u) / 1024u), dim3(256), 0, s, &a->ranges);
}

void issue_tile_binning(const FwdPlan& p, const FwdPassArgs* a, hipStream_t s)
{
    if (p.layout.wide == 2)
        issue_tile_binning_t<IoSplit>(p, a, s);
    else if (p.layout.wide == 1)
        issue_tile_binning_t<IoWide>(p, a, s);
    else
        issue_tile_binning_t<IoNarrow>(p, a, s);
}

// debug accessor: rebuild the reference's 64-bit keys (tile << 32 | depth bits) of the sorted list
template <class IO>
__global__ __launch_bounds__(256) void export_keys_kernel(int R, uint32_t cap, int rank_bits, const char* __restrict__ sorted,
                                                          const uint32_t* __restrict__ point_list,
                                                          const uint32_t* __restrict__ depth_key, const GeomHeader* hdr,
                                                          uint64_t* keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    // a caller holding only the reference's num_rendered may ask for more entries than the pass binned (opacity-aware
    // rects): what lies behind the list is not a list entry
    if ((uint32_t)i >= hdr->num_pairs) {
        keys[i] = ~(uint64_t)0;
        return;
    }
    keys[i] = ((uint64_t)IO::tile_at(sorted, cap, (uint32_t)i, rank_bits) << 32) | (uint64_t)depth_key[point_list[i]];
}

// debug accessor: the first min(count, pairs binned) entries of the point list (the caller has filled `out` with ~0)
__global__ __launch_bounds__(256) void export_point_list_kernel(int count, const uint32_t* __restrict__ point_list,
                                                                const GeomHeader* hdr, uint32_t* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < count && (uint32_t)i < hdr->num_pairs) out[i] = point_list[i];
}

void launch_export_point_list(int count, const uint32_t* point_list, const GeomHeader* hdr, uint32_t* out, hipStream_t s)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(export_point_list_kernel, dim3((count + 255) / 256), dim3(256), 0, s, count, point_list, hdr, out);
}

// which buffer of the blob holds the sorted words (split layout: their keys) after the passes (see the fill of RadixArgs
// in capi.hip)
const char* sorted_words(const BinState& b, const PairLayout& l)
{
    if (l.wide) return (l.passes & 1) ? b.words_b : b.words_a;   // a -> b -> a (-> b)
    return (l.passes & 1) ? b.words_b : b.words_c;                // a -> b -> c (-> b)
}

void launch_export_keys(int P, int R, int cap, size_t n_tiles, const BinState& b, const GeomState& g, uint64_t* keys_out,
                        hipStream_t s)
{
    if (R <= 0) return;
    const PairLayout l = pair_layout(P, n_tiles);
    const char* sorted = sorted_words(b, l);
    const dim3 grid((R + 255) / 256), block(256);
    if (l.wide == 2)
        hipLaunchKernelGGL(export_keys_kernel<IoSplit>, grid, block, 0, s, R, (uint32_t)cap, l.rank_bits, sorted, b.point_list,
                           g.depth_key, g.header, keys_out);
    else if (l.wide == 1)
        hipLaunchKernelGGL(export_keys_kernel<IoWide>, grid, block, 0, s, R, (uint32_t)cap, l.rank_bits, sorted, b.point_list,
                           g.depth_key, g.header, keys_out);
    else
        hipLaunchKernelGGL(export_keys_kernel<IoNarrow>, grid, block, 0, s, R, (uint32_t)cap, l.rank_bits, sorted,
                           b.point_list, g.depth_key, g.header, keys_out);
}

}  // namespace r3
