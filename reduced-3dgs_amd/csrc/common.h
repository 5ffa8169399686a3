// common.h -- buffer layouts, pass argument blocks, launch helpers and error plumbing shared by the HIP
// translation units.
//
// The three caller-owned byte buffers play the role of the reference's GeometryState / BinningState /
// ImageState blobs (cuda_rasterizer/rasterizer_impl.h:21-73, rasterizer_impl.cu:163-202): produced by
// forward, kept alive by the caller, handed back verbatim to backward.
// Their internal layout is private to this library and is MI355X-first, not the reference's:
//   * one 48-byte gather record per Gaussian (GRec) instead of five separate arrays,
//   * depth-sorted Gaussian order + packed (tile | depth rank) words instead of 64-bit (tile|depth) keys,
//   * a per-(tile, Gaussian)-pair gradient slab for the backward (no float atomics anywhere).
// The parts the backward needs sit at the FRONT of each blob so that their offsets do not depend on
// library temp-storage sizes.
//
// How a pass is issued (capi.hip): every kernel of a pass reads its pointers and scalars from ONE device-resident
// argument block (FwdPassArgs / BwdPassArgs) instead of from its kernel arguments.  The block is (re)written by a
// one-workgroup kernel whose by-value argument IS the block, so a whole pass is "write the block, run a fixed chain
// of kernels whose launch parameters never change": that chain is captured once per shape into a hipGraph and
// replayed with one hipGraphLaunch per pass (the measured host cost of 20 direct launches is 54 us idle and
// milliseconds on a loaded host; one graph launch is ~8 us either way).  Nothing the host does depends on
// num_rendered any more: grids are sized by the caller's pair RESERVATION, the kernels read the pair count from the
// device header.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <stdexcept>
#include <string>

#include "gauss_math.h"

namespace r3 {

constexpr size_t kAlign = 256;
constexpr int kPairGrad = 9;  // floats per (tile, Gaussian) pair parked by the backward blend:
                              // dmean2D.xy, dconic.xyw, dopacity, dcolor.rgb
#ifndef R3_ACC_STRIDE
#define R3_ACC_STRIDE 12
#endif
#ifndef R3_PAIR_STRIDE
#define R3_PAIR_STRIDE 12
#endif
// floats per piece of wave_part (the leading / trailing partial run sum of a 64-pair group): 9 packed (the round-2 layout, nine
// 4-byte stores by one lane), or 12 with -DR3_WP_VEC (experiment: three 16-byte stores and loads per piece)
#ifdef R3_WP_VEC
constexpr int kPieceStride = 12;
#else
constexpr int kPieceStride = 9;
#endif
constexpr int kAccStride = R3_ACC_STRIDE;    // floats per Gaussian in the reduced 2D-stage gradient row (9 used)
constexpr int kPairStride = R3_PAIR_STRIDE;  // floats per row of the per-pair slab: 48-B rows, written / read as three float4
                                // (36-B rows measured 2.4x write amplification: partial lines, nine dword accesses)

// Per-view counters.  Every preprocess workgroup stores one PrePartial (no atomics, nothing to pre-clear: the first
// GPU profile showed 7.8k same-address atomics costing ~90 us, a sharded version still needed a fill of the header
// in front of every pass); one workgroup reduces them into the header.
constexpr int kPreBlockSize = 256;
struct PrePartial {
    uint32_t visible;        // #Gaussians with radii > 0 in the workgroup (SH-sparsity normaliser, rasterizer_impl.cu:549-566)
    uint32_t num_rendered;   // sum of tiles_touched of the rects the Gaussians are BINNED into: the pairs the pass wants
    uint32_t depth_max;      // max depth bits over the binned Gaussians (0 if none)
    uint32_t depth_inv_min;  // max of ~(depth bits), i.e. ~min (0 if none)
    uint32_t binned;         // #Gaussians with tiles_touched > 0 (<= visible: opacity-aware rects, gauss_math.h tighten_rect)
    uint32_t rendered_ref;   // sum of the tile counts of the REFERENCE's rects: what the reference calls num_rendered
    uint32_t pad[2];
};
static_assert(sizeof(PrePartial) == 32, "two 16-byte loads per partial");
struct GeomHeader {
    uint32_t visible, num_rendered, depth_max, depth_inv_min;   // num_rendered: pairs wanted (see PrePartial)
    uint32_t num_pairs;       // min(num_rendered, reserve): the pairs every later kernel of the pass works on
    uint32_t reserve;         // pair capacity of the binning blob of this pass
    uint32_t sort_overflow;   // a depth bucket exceeded the LDS sort capacity (handled on the device; host hint only)
    uint32_t binned;          // Gaussians that own pairs: the first `binned` entries of the depth order
    uint32_t rendered_ref;    // the reference's num_rendered (reported; nothing on the device uses it)
    uint32_t sh_cache;        // 1: GeomState::sh_ddir holds this pass's SH direction derivatives (dense SH input)
    uint32_t ckpt;            // log2 of the segment length the forward blend left checkpoints for (BinState::ckpt); 0: none
    uint32_t ckpt_thr;        // ... in the tiles whose list has at least this many entries
    uint32_t pad[52];
};
static_assert(sizeof(GeomHeader) == 256, "header = 256 B");
inline size_t pre_partials(size_t P) { return (P + kPreBlockSize - 1) / kPreBlockSize; }

// What the host may want to know about a pass, written by the pass's kernels into HOST-mapped memory (one ring slot
// per pass ticket): the host never waits for it on the asynchronous path, and polls plain memory (no HIP calls) on
// the exact path.  `seq` is stored last, after a system-scope fence.
struct PassInfo {
    uint32_t seq;            // low 32 bits of the pass number once num_rendered / pairs / visible are valid
    uint32_t num_rendered;   // sum of tiles_touched as the reference defines it (rasterizer_impl.cu:441-446)
    uint32_t visible;
    uint32_t reserve;
    uint32_t sort_overflow;  // hint: a depth bucket overflowed (slow in-kernel path was taken); valid once ...
    uint32_t sort_seq;       // ... this equals the pass number (stamped by the depth sort's scan kernel, which runs
                             // after the header is published)
    uint32_t pairs;          // (tile, Gaussian) pairs the pass wants to emit; above `reserve` the farthest are dropped
    uint32_t pad;
};
static_assert(sizeof(PassInfo) == 32, "PassInfo = 32 B");

// Bucketed depth sort (binning.hip): `nb` monotone buckets over [min depth bits, max depth bits], each sorted by one
// workgroup in LDS.  nb scales with P (<= kMaxDepthBuckets).
constexpr int kMinDepthBuckets = 1024;
constexpr int kMaxDepthBuckets = 16384;
constexpr int kBucketCap = 4096;     // (key, id) pairs one workgroup sorts in LDS (32 KB)
#ifndef R3_HIST_BATCH
#define R3_HIST_BATCH 4096
#endif
constexpr int kHistBatch = R3_HIST_BATCH;   // Gaussians a histogram / scatter workgroup handles per round (16 per thread)
// Gaussians per histogram / scatter workgroup: rounds of kHistBatch, as many as keep the row count near 256 (one
// workgroup per CU: their nb-entry LDS tables leave room for no more at large nb).  Every such workgroup carries
// nb-entry tables (its histogram row, its scan of the bucket totals), so with a fixed 4096 Gaussians per workgroup
// the table traffic grew like P * nb: 0.44 ms for the scatter alone at 6 M Gaussians.
inline size_t depth_hist_per_block(size_t P)
{
    const size_t want = (P + 255) / 256;
    const size_t per = ((want + kHistBatch - 1) / kHistBatch) * kHistBatch;
    return per < (size_t)kHistBatch ? (size_t)kHistBatch : per;
}
int depth_bucket_load();   // mean Gaussians per bucket aimed for up to 1 M Gaussians (R3DGS_DEPTH_BUCKET_LOAD; default 128); capi.hip
inline int depth_bucket_count(size_t P)
{
    int nb = kMinDepthBuckets;
    // every histogram / scatter workgroup carries nb-entry tables, so large scenes take coarser buckets.  Stage time:
    // 500 k Gaussians: 2048 / 4096 / 8192 buckets 0.099 / 0.084 / 0.106 ms; 2 M: 4096 / 8192 / 16384 0.26 / 0.30 / 0.38 ms;
    // 6 M: 8192 / 16384 0.76 / 0.98 ms -- i.e. ~128 per bucket up to 1 M, ~512 up to 4 M, ~1024 above
    const size_t load = (size_t)depth_bucket_load() * (P > (1u << 22) ? 8 : P > (1u << 20) ? 4 : 1);
    while (nb < kMaxDepthBuckets && P / (size_t)nb > load) nb <<= 1;
    return nb;
}
struct DepthSortScratch {
    uint32_t depth_max, depth_inv_min;               // depth range of the visible Gaussians
    uint32_t big_count, pad_;                        // buckets above one wave's capacity (big_list): dealt out over all the
                                                     // workgroups of the bucket-sort kernel (a real scene piles its foreground
                                                     // into a few hundred neighbouring buckets; cleared by the histogram
                                                     // kernel, filled by the column scan)
    unsigned long long total[kMaxDepthBuckets + 1];  // per bucket (tile sum << 24 | count); [nb] = culled
    uint32_t start[kMaxDepthBuckets + 2];            // exclusive scan of the bucket sizes
    uint32_t tile_base[kMaxDepthBuckets + 2];        // exclusive scan of the buckets' tiles_touched sums
    uint32_t big_list[kMaxDepthBuckets];             // the buckets above one wave's capacity, in no particular order
};
inline size_t depth_hist_rows(size_t P) { const size_t per = depth_hist_per_block(P); return (P + per - 1) / per; }

struct Carver {
    char* p;
    explicit Carver(char* base)
        : p(reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(base) + kAlign - 1) & ~(uintptr_t)(kAlign - 1)))
    {
    }
    template <class T>
    T* take(size_t count)
    {
        T* r = reinterpret_cast<T*>(p);
        size_t bytes = (count * sizeof(T) + kAlign - 1) & ~(kAlign - 1);
        p += bytes;
        return r;
    }
};

struct GeomState {
    GeomHeader* header;
    PrePartial* partials;     // [ceil(P / 256)]
    DepthSortScratch* dsort;
    unsigned long long* hist_rows;  // [rows][nb + 1]  per-workgroup (tile sum << 24 | count) histograms
    uint32_t* hist_base;            // [rows][nb + 1]  first slot of each workgroup inside each bucket
    GRec* rec;            // [P]
    float* acc;           // [P * kAccStride]  per-Gaussian sums of the per-pair gradients (backward): 9 sums, then the 64-bit stamp of
                          // the backward pass that wrote the row (a row without the current stamp reads as zeros)
    ushort4* rect;        // [P]  tile rect (minx, miny, maxx, maxy)
    uint32_t* depth_key;  // [P]  float bits of view depth, 0xFFFFFFFF when culled
    uint32_t* tiles;      // [P]  tiles_touched
    uint32_t* key_sorted; // [P]  sorted keys (generic sort); bucketed sort: the keys of a bucket too big for the LDS sort
    uint32_t* bucket_id;  // [P]  ... and its ids (ping-pong partners of ovf_key / ovf_id on that slow path)
    uint4* rec16;         // [P]  (depth key, id, rect x0 | y0 << 16, x1 | y1 << 16) grouped by depth bucket (depth_sort.h)
    ushort4* rect_sorted; // [P]  tile rects in depth order, for the pair emission (bucketed sort only)
    uint4* rec16_b;       // [P]  second record array: where a bucket too big for the LDS sort is split into sub-buckets
    uint32_t* order;      // [P]  Gaussian ids in (depth, id) order
    uint32_t* offsets;    // [P]  inclusive scan of tiles[order[j]]
    int* radii_internal;  // [P]  used when the caller passes radii == nullptr (rasterizer_impl.cu:393-396)
    float* sh_ddir;       // [P * 9]  d(colour channel)/d(view direction) of the SH expansion (gauss_math.h sh_dir_derivs),
                          //          left by the forward's colour stream for visible Gaussians of degree > 0: the backward
                          //          then needs the 12 * M-byte SH row only for the sparsity term
    uint32_t* ovf_key;    // [P]  ping-pong partner of key_sorted for a depth bucket that overflows the LDS sort
    uint32_t* ovf_id;     // [P]  ... and of bucket_id
    char* temp;           // rocPRIM temp storage of the generic depth sort
    size_t temp_bytes;
    static GeomState carve(char* base, size_t P, size_t temp_bytes)
    {
        Carver c(base);
        GeomState g;
        const size_t nb = (size_t)depth_bucket_count(P);
        g.header = c.take<GeomHeader>(1);
        g.partials = c.take<PrePartial>(pre_partials(P));
        g.dsort = c.take<DepthSortScratch>(1);
        g.hist_rows = c.take<unsigned long long>(depth_hist_rows(P) * (nb + 1));
        g.hist_base = c.take<uint32_t>(depth_hist_rows(P) * (nb + 1));
        g.rec = c.take<GRec>(P);
        g.acc = c.take<float>(P * kAccStride);
        g.rect = c.take<ushort4>(P);
        g.depth_key = c.take<uint32_t>(P);
        g.tiles = c.take<uint32_t>(P);
        g.key_sorted = c.take<uint32_t>(P);
        g.bucket_id = c.take<uint32_t>(P);
        g.order = c.take<uint32_t>(P);
        g.offsets = c.take<uint32_t>(P);
        g.radii_internal = c.take<int>(P);
        g.ovf_key = c.take<uint32_t>(P);
        g.ovf_id = c.take<uint32_t>(P);
        g.rec16 = c.take<uint4>(P);
        g.rect_sorted = c.take<ushort4>(P);
        g.rec16_b = c.take<uint4>(P);
        g.temp = c.take<char>(temp_bytes);
        g.temp_bytes = temp_bytes;
        // LAST, so that a blob without it is a valid blob: a forward that will not leave the derivatives (inference /
        // ragged SH / precomputed colours / r3dgs_forward_hint(0)) may be given a blob of lean_end's size (36 B per
        // Gaussian less: 216 MB at 6 M Gaussians); the pass header says whether they are there, nothing else looks
        g.lean_end = c.p;
        g.sh_ddir = c.take<float>(P * 9);
        g.end = c.p;
        return g;
    }
    char* lean_end;
    char* end;
};

// rasterizer_impl.cu:43-58 getHigherMsb
inline uint32_t higher_msb(uint32_t n)
{
    uint32_t msb = sizeof(n) * 4, step = msb;
    while (step > 1) {
        step /= 2;
        if (n >> msb)
            msb += step;
        else
            msb -= step;
    }
    if (n >> msb) msb++;
    return msb;
}

// Tile sort of the packed pair words (binning.hip): a pair travels as ONE word, tile << rank_bits | Gaussian id,
// through a key-only, stable LSD radix sort on the tile bits; emission is in depth order, so the order inside a
// tile is (depth, id) without the depth ever being part of the key.
// 32-bit words while tile bits + id bits <= 32 (the BASELINE shape: 13 + 19); above -- every real scene -- the word
// lives as a 16-bit tile key array + a 32-bit id array (up to 65536 tiles; 64-bit words tile << 32 | id beyond).
// Two or three stable passes of <= 8-bit digits, kRadixBlock keys per workgroup.
// Forward and backward derive the same layout from (P, #tiles) alone.
constexpr int kRadixBlock = 2048;   // pairs per workgroup of the emission / radix kernels; 1024 / 2048 / 4096: tile binning
                                    // 0.097 / 0.092 / 0.107 ms at 500 k Gaussians, 0.327 / 0.280 / 0.294 ms at 2 M
constexpr int kMaxRadixBins = 256;
constexpr int kMaxRadixPasses = 3;
struct PairLayout {
    int tile_bits;   // bits holding the tile id
    int rank_bits;   // shift of the tile id inside the word = bits holding the Gaussian id (32 for wide words)
    int wide;        // 0: 32-bit words (tile | id);  1: 64-bit words (tile << 32 | id);  2: split -- a 16-bit tile key array
                     // and a 32-bit id array (6 bytes per pair through the passes instead of 8; needs <= 65536 tiles)
    int passes;      // radix passes over the tile bits
    int digit_bits;  // bits per pass
};
PairLayout pair_layout(int P, size_t n_tiles);   // binning.hip (honours R3DGS_TILE_SORT=wide for A/B runs and tests)

// workgroups of the emission / radix grids for R pairs, padded so that every digit's row of per-workgroup counts starts
// 16-byte aligned (the digit scan moves four counts per access)
inline uint32_t radix_row_stride(size_t R)
{
    return (uint32_t)(((R + kRadixBlock - 1) / kRadixBlock + 3) & ~(size_t)3);
}

// ---- list segments of the backward blend (blend.hip) ------------------------------------------------------------------
// A tile whose list is long is walked by SEVERAL workgroups of the backward blend, each over one segment of the list:
// a real scene has tiles many times heavier than the mean (a foreground object in front of an empty sky), and a kernel of
// one wave per tile is then as long as its heaviest tile is slow -- the chip drains for the last third of it
// (profiles/r05_bwd_timeline_clustered.txt).  A segment that does not start at the list's end needs the per-pixel state
// there: the forward blend checkpoints (T, accumulated colour) of every pixel each S = 2^GeomHeader::ckpt entries while it
// walks such a list, and its final colour.  Which lists: those at least GeomHeader::ckpt_thr entries long -- the pass's
// mean list length times a factor, at least 2 S (the forward only knows the LENGTH of a list, not how deep its pixels will
// look; a uniform scene then pays nothing).  Slot of checkpoint k >= 1 (state in front of entry k * S; k = 0: the final
// colour) of tile t whose list starts at pair `first`:  first / S + t + k  -- lists are contiguous and in tile order, so
// the slots of different tiles are disjoint without a prefix sum (as quad_mask_slot).
constexpr int kBwdSegMinLog2 = 7;      // smallest segment: 128 entries (two 64-entry chunks); the checkpoint pool is sized for it
constexpr int kBwdSegMaxLog2 = 8;
constexpr int kBwdSegMax = 32;         // segments per tile at most: the last one takes whatever is left
constexpr int kFwdCkptMax = 256;       // checkpoints per tile at most (a pass may walk segments of up to 8 S: 31 x 8 < 256)
constexpr uint32_t kUnitTileBits = 20; // a unit word of the backward's launch order: tile | segment << 20 | segments << 26
R3_HD size_t ckpt_slot(uint32_t first, uint32_t k, uint32_t tile, uint32_t seg_log2) { return (size_t)(first >> seg_log2) + tile + k; }
inline size_t ckpt_slots(size_t R, size_t Tn) { return (R >> kBwdSegMinLog2) + Tn + 2; }
// The unit order of the backward blend is kept as kOrderLists lists, each made by a workgroup of its own; workgroup b of the
// backward blend takes entry b / 8 of list b % 8, and the hardware deals consecutive workgroups over the eight XCDs -- list g
// is what XCD g walks.  List g holds the kListBlock x kListBlock TILE BLOCKS g, g + 8, ... of the image (blocks row-major):
// neighbouring tiles share most of their Gaussians, so a record is fetched into few L2s (as every 8th TILE per list the
// kernel's FETCH_SIZE was 220 MB on the metric shape, as 4 x 4 blocks 162 MB, WRITE_SIZE 132 -> 119 MB; time equal or 0.5 %
// better: profiles/r05_exp_list_blocks.txt), and 425 blocks dealt over 8 lists are still statistically alike tile sets.
constexpr uint32_t kOrderLists = 8, kListBlock = 4, kListBlockTiles = kListBlock * kListBlock;
struct TileGrid {
    uint32_t gx, gy;
    R3_HD size_t n() const { return (size_t)gx * gy; }
    R3_HD uint32_t blocks_x() const { return (gx + kListBlock - 1) / kListBlock; }
    R3_HD uint32_t blocks() const { return blocks_x() * ((gy + kListBlock - 1) / kListBlock); }
    // slots of list `list` (a block at the image's edge has slots without a tile), and the tile of slot j (~0: none)
    R3_HD uint32_t list_slots(uint32_t list) const
    {
        return blocks() > list ? (blocks() - list + kOrderLists - 1) / kOrderLists * kListBlockTiles : 0u;
    }
    R3_HD uint32_t list_tile(uint32_t list, uint32_t j) const
    {
        const uint32_t k = j / kListBlockTiles * kOrderLists + list, o = j % kListBlockTiles, bx = blocks_x();
        const uint32_t tx = k % bx * kListBlock + o % kListBlock, ty = k / bx * kListBlock + o / kListBlock;
        return tx < gx && ty < gy ? ty * gx + tx : 0xFFFFFFFFu;
    }
    // no list holds more tiles than this
    R3_HD uint32_t list_tiles_max() const
    {
        const size_t m = (size_t)((blocks() + kOrderLists - 1) / kOrderLists) * kListBlockTiles;
        return (uint32_t)(m < n() ? m : n());
    }
};
// workgroups of the backward blend = capacity of its unit order: per list its tiles once, plus its share of the extra
// segments a pass may have (bounded by the lists: sum over tiles of len / S <= R / S; capped -- a list that wants more
// walks longer segments)
R3_HD uint32_t bwd_list_fit(uint32_t pairs, const TileGrid& g)   // units a list may hold in a pass of `pairs` pairs
{
    const size_t extra = (size_t)pairs >> kBwdSegMinLog2, most = 8 * g.n();
    return g.list_tiles_max() + (uint32_t)((extra < most ? extra : most) / kOrderLists);
}
inline uint32_t bwd_units_cap(uint32_t reserve, const TileGrid& g) { return (bwd_list_fit(reserve, g) + 1u) * kOrderLists; }
int bwd_segment_log2();         // R3DGS_BWD_SEG_LEN = 128 (default) | 256 -> 7 | 8; capi.hip
int bwd_segment_factor_pct();   // R3DGS_BWD_SEG_FACTOR: lists >= this percentage of the pass's mean length are split (75)

struct BinState {
    uint32_t* point_list;  // [R] Gaussian ids, tile-major, (depth, id) order inside a tile
    float* pair_grad;      // [R * kPairGrad] per-pair gradients in EMISSION order (Gaussian-major), backward only
    float* wave_part;      // [(R/64+1) * 2 * kPairGrad] leading / trailing partial run sums of each 64-pair group
    unsigned char* pair_flag;  // [R] 1 = the backward blend wrote this pair's row (only this is zeroed per pass)
    uint32_t* pair_rank;   // [R] Gaussian id of the emitted pair, emission order (run key of the backward's segmented
                           //     sum); narrow words: aliases words_a (the low rank_bits of the word)
    char* words_a;         // [R] packed words in emission order (narrow) / ping-pong buffer A (wide)
    char* words_b;         // [R] ping-pong buffer B
    char* words_c;         // [R] narrow only: third buffer, so that words_a survives for the backward
    uint32_t* radix_rows;  // [bins][R/kRadixBlock + 1] per-workgroup digit counts, digit-major
    uint32_t* radix_base;  // same shape: first slot of each workgroup inside each digit
    uint32_t* radix_total; // [kMaxRadixPasses][kMaxRadixBins] digit totals of the passes
    uint32_t* block_first;           // [R/kRadixBlock + 2] depth rank of the Gaussian holding pair m * kRadixBlock: written by the
                                     // depth sort's scan (asynchronous path), spares the emission kernel its search
    unsigned long long* quad_masks;  // [R/64 + Tn + 2][4] region pre-test of the forward blend, kept for the backward:
                                     // bit j of [slot][q] = entry j of a 64-entry chunk of a tile's list may reach 8x8
                                     // quadrant q; slot = quad_mask_slot(first pair of the tile, chunk, tile)
    uint32_t* unit_order;            // [bwd_units_cap + 2 * kOrderLists] launch order of the backward blend's (tile, segment)
                                     // units: kOrderLists lists of cap / kOrderLists slots, each heaviest first; behind
                                     // them per list how many units it has and log2 of the segment length they walk
                                     // (blend.hip unit_order_kernel)
    float4* ckpt;                    // [ckpt_slots][256] forward checkpoints of the segmented tiles: (T, C0, C1, C2) per pixel
    char* end;
    static BinState carve(char* base, size_t R, int wide, const TileGrid& grid)
    {
        const size_t Tn = grid.n();
        Carver c(base);
        BinState b;
        b.point_list = c.take<uint32_t>(R);
        b.pair_grad = c.take<float>(R * kPairStride);
        b.wave_part = c.take<float>((R / 64 + 1) * 2 * kPieceStride);
        b.pair_flag = c.take<unsigned char>(R);
        if (wide == 1) {
            b.pair_rank = c.take<uint32_t>(R);
            b.words_a = reinterpret_cast<char*>(c.take<unsigned long long>(R));
            b.words_b = reinterpret_cast<char*>(c.take<unsigned long long>(R));
            b.words_c = nullptr;
        } else if (wide == 2) {   // per buffer: [R] ids, then [R] 16-bit keys; the emission-order ids of buffer A survive
            b.words_a = reinterpret_cast<char*>(c.take<unsigned long long>(R));
            b.words_b = reinterpret_cast<char*>(c.take<unsigned long long>(R));
            b.words_c = nullptr;
            b.pair_rank = reinterpret_cast<uint32_t*>(b.words_a);
        } else {
            b.words_a = reinterpret_cast<char*>(c.take<uint32_t>(R));
            b.words_b = reinterpret_cast<char*>(c.take<uint32_t>(R));
            b.words_c = reinterpret_cast<char*>(c.take<uint32_t>(R));
            b.pair_rank = reinterpret_cast<uint32_t*>(b.words_a);
        }
        b.radix_rows = c.take<uint32_t>((size_t)kMaxRadixBins * radix_row_stride(R));
        b.radix_base = c.take<uint32_t>((size_t)kMaxRadixBins * radix_row_stride(R));
        b.radix_total = c.take<uint32_t>(kMaxRadixPasses * kMaxRadixBins);
        b.quad_masks = c.take<unsigned long long>((R / 64 + Tn + 2) * 4);
        b.block_first = c.take<uint32_t>(R / kRadixBlock + 2);
        b.unit_order = c.take<uint32_t>((size_t)bwd_units_cap((uint32_t)R, grid) + 2 * kOrderLists);
        // LAST: only touched for the tiles a pass segments (32 B per pair + 4 KB per tile of address space)
        b.ckpt = c.take<float4>(ckpt_slots(R, Tn) * 256);
        b.end = c.p;
        return b;
    }
};

// Slot of chunk `chunk` (64 entries) of the list of tile `tile`, whose first pair is `first`: lists are contiguous and in
// tile order, so floor(first / 64) + chunk grows by at least (chunks of the tile) - 1 from a tile to the next; adding the
// tile id makes the slots of different tiles disjoint without a prefix sum over the tiles' chunk counts.
R3_HD size_t quad_mask_slot(uint32_t first, uint32_t chunk, uint32_t tile)
{
    return (size_t)(first >> 6) + chunk + tile;
}

struct ImageState {
    float* final_T;       // [N]
    uint32_t* n_contrib;  // [N]
    uint2* ranges;        // [Tn]
    uint32_t* quad_depth; // [Tn][4] deepest contributor of each 8x8 quadrant (max of n_contrib), left by the forward blend
    uint32_t* tile_order; // [Tn] unused since the launch order became a list of (tile, segment) units (BinState::unit_order);
                          // kept so that the image blob keeps its size
    char* end;
    static ImageState carve(char* base, size_t N, size_t Tn)
    {
        Carver c(base);
        ImageState s;
        s.final_T = c.take<float>(N);
        s.n_contrib = c.take<uint32_t>(N);
        s.ranges = c.take<uint2>(Tn);
        s.quad_depth = c.take<uint32_t>(Tn * 4);
        s.tile_order = c.take<uint32_t>(Tn);
        s.end = c.p;
        return s;
    }
};

template <class S, class... A>
size_t required_bytes(A... a)
{
    S s = S::carve(nullptr, a...);
    return (size_t)reinterpret_cast<uintptr_t>(s.end) + kAlign;
}

// temp-storage query of the generic (rocPRIM) depth sort (binning.hip); needs a visible GPU
size_t depth_sort_temp_bytes(size_t P);

// Launch size of the pair-sized kernels for a binning blob of `reserve` pairs: the reservation is ~1.5x the largest
// recent pair count plus a constant (capi.hip reserve_hint), the grids cover that count with a little headroom and
// their blocks stride over whatever lies beyond.  A function of `reserve` alone, so forward, backward and the graph
// cache agree on it.
inline uint32_t grid_pairs_for(uint32_t reserve)
{
    const uint64_t g = (uint64_t)reserve * 7 / 10;
    return (uint32_t)(g < 1024 ? (reserve < 1024 ? reserve : 1024) : g);
}

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Pointers in the pass blocks are plain (generic) pointers; a load through one is a FLAT load, which also counts on the LDS
// counter -- every wait for an LDS result then drains the memory loads in flight.  Kernels that keep loads in flight across LDS
// work cast to the global address space first.
#if defined(__HIP_DEVICE_COMPILE__)
#define R3_GLOBAL __attribute__((address_space(1)))
#else
#define R3_GLOBAL   // host pass: the kernels are only parsed
#endif
template <class T>
__device__ __forceinline__ R3_GLOBAL T* global_ptr(T* p)
{
    return (R3_GLOBAL T*)p;
}

// Runs f, turning exceptions into (-1, r3dgs_last_error()); defined in capi.hip, shared by every extern "C" TU.
int guarded_call(const std::function<int()>& f);

#define R3_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            throw r3::Error(std::string(#expr) + " failed: " + hipGetErrorString(e_) + " (" + __FILE__ + \
                            ":" + std::to_string(__LINE__) + ")");                                     \
    } while (0)

inline void check_launch(const char* what, hipStream_t s, bool debug)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw Error(std::string(what) + ": launch failed: " + hipGetErrorString(e));
    if (debug) {  // mirrors the reference's CHECK_CUDA(debug) (auxiliary.h:161-168): sync + surface the error here
        e = hipStreamSynchronize(s);
        if (e != hipSuccess) throw Error(std::string(what) + ": " + hipGetErrorString(e));
    }
}

// Per-view parameters as they arrive at the boundary: the matrices, camera position and background
// are DEVICE tensors (gaussian_renderer/__init__.py:37-50 passes CUDA tensors), so kernels read them
// through wave-uniform (scalar) loads.
struct ViewParams {
    const float* view;    // [16] world->view, transposed/row-vector layout (scene/cameras.py:54)
    const float* proj;    // [16] full projection, same layout
    const float* campos;  // [3]
    const float* bg;      // [3]
    float tan_fovx, tan_fovy;
    int W, H;
    float scale_modifier;
};

__device__ __forceinline__ Camera load_camera(const ViewParams& v)
{
    Camera c;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        c.view[k] = v.view[k];
        c.proj[k] = v.proj[k];
    }
    c.campos[0] = v.campos[0];
    c.campos[1] = v.campos[1];
    c.campos[2] = v.campos[2];
    c.tan_fovx = v.tan_fovx;
    c.tan_fovy = v.tan_fovy;
    c.focal_y = v.H / (2.0f * v.tan_fovy);  // rasterizer_impl.cu:386-387
    c.focal_x = v.W / (2.0f * v.tan_fovx);
    c.W = v.W;
    c.H = v.H;
    c.gx = (v.W + kTile - 1) / kTile;
    c.gy = (v.H + kTile - 1) / kTile;
    c.scale_modifier = v.scale_modifier;
    return c;
}

struct FwdInputs {
    int P, M;
    const int* degrees;
    const float* means3D;
    const float* scales;
    const float* rotations;
    const float* opacities;
    const float* shs;
    const float* cov3D_precomp;
    const float* colors_precomp;
    // ragged SH (inference variant, forward.cu:19-36); null for the dense path
    const int* coeffs_num;
    const int* per_band_count;
    const int* cumsum_count;
};

struct BwdOutputs {
    float* dL_dmean2D;   // [P,3]
    float* dL_dopacity;  // [P,1]
    float* dL_dcolor;    // [P,3]
    float* dL_dmean3D;   // [P,3]
    float* dL_dcov3D;    // [P,6]
    float* dL_dsh;       // [P,M,3]
    float* dL_dscale;    // [P,3]
    float* dL_drot;      // [P,4]
    float* dL_dconic;    // [P,4] optional (nullptr: not exported)
};

// ---- per-kernel argument blocks (members of the pass blocks below) -----------------------------------------------
struct PreArgs {          // preprocess.hip
    FwdInputs in;
    ViewParams view;
    GRec* rec;
    ushort4* rect;
    uint32_t* depth_key;
    uint32_t* tiles;
    PrePartial* partials;
    int* radii;
    int color_blocks;     // workgroup-sized chunks of the colour kernel's persistent loop
    int tight;            // 1: bin into the opacity-aware rect (gauss_math.h tighten_rect), 0: into the reference's
    float* sh_ddir;       // GeomState::sh_ddir (null: ragged SH / precomputed colours -- nothing to leave)
};
struct HeaderArgs {       // binning.hip header_reduce_kernel
    const PrePartial* parts;
    int n_parts;
    GeomHeader* hdr;
    PassInfo* info;       // host-mapped slot of this pass
    uint32_t ticket;
    uint32_t reserve;
    uint32_t stamp_sort;  // 1: no depth-scan kernel follows (generic sort): the header stamps PassInfo::sort_seq itself
    uint32_t sh_cache;    // -> GeomHeader::sh_cache
    uint32_t ckpt;        // -> GeomHeader::ckpt (log2 of the segment length; 0: the forward leaves no checkpoints)
    uint32_t ckpt_factor_pct, n_tiles;   // GeomHeader::ckpt_thr = max(2 S, factor * num_pairs / n_tiles)
};
struct DepthArgs {        // depth_sort.h bucketed depth sort
    int P, nb, rows, per_block;
    int fuse_header;      // 1: the histogram workgroups reduce the preprocess partials themselves and workgroup 0 writes
                          // the header / PassInfo (no header_reduce launch in front: asynchronous path)
    const uint32_t* key;
    const uint32_t* tiles;
    GeomHeader* hdr;
    PassInfo* info;
    uint32_t ticket;      // pass number (stamp of PassInfo::sort_seq)
    DepthSortScratch* ds;
    unsigned long long* hist_rows;
    uint32_t* hist_base;
    uint32_t* key_sorted;
    uint32_t* bucket_id;
    uint32_t* ovf_key;
    uint32_t* ovf_id;
    const ushort4* rect;     // GeomState::rect (by Gaussian id)
    uint4* rec16;            // GeomState::rec16
    uint4* rec16_b;          // GeomState::rec16_b
    ushort4* rect_sorted;    // GeomState::rect_sorted (written by the bucket sort)
    uint32_t* order;
    uint32_t* offsets;
    uint32_t* block_first;   // BinState::block_first or nullptr (exact-size path: the binning blob does not exist yet)
    uint32_t block_cap;      // entries of block_first
};
struct EmitArgs {         // binning.hip emit_pairs_kernel
    int P, gx;
    const GeomHeader* hdr;
    const uint32_t* order;
    const uint32_t* offsets;
    const uint32_t* block_first;   // as DepthArgs::block_first (nullptr: search the scan)
    const ushort4* rect;          // by Gaussian id (generic depth sort)
    const ushort4* rect_sorted;   // in depth order, left by the bucketed depth sort (nullptr: gather rect[id])
    GRec* rec;
    int rank_bits, digit_bits;
    char* words_out;
    uint32_t cap;          // pair capacity of the word buffers (the split layout's key array starts behind cap ids)
    uint32_t* pair_rank;   // 64-bit words only (else nullptr: it aliases the emitted words / ids)
    uint2* ranges;
    uint32_t n_tiles;
    uint32_t* radix_rows;
    uint32_t row_stride;   // workgroups of the emission / radix grids (reserve / kRadixBlock, rounded up)
};
struct RadixArgs {        // binning.hip one LSD pass (hist -> digit scan -> scatter)
    const GeomHeader* hdr;
    const char* in;
    char* out;
    uint32_t* ids_out;     // split layout, last pass: the ids go straight into point_list (else nullptr)
    uint32_t cap;
    int shift, digit_bits; // shift = rank_bits + tile_shift: position of the pass's digit inside a register word
    int rank_bits, tile_shift;
    uint32_t* rows;
    uint32_t* base;
    uint32_t* total;
    uint32_t row_stride;
};
struct RangesArgs {       // binning.hip tile_ranges_kernel
    const GeomHeader* hdr;
    const char* sorted;
    uint32_t cap;
    int rank_bits;
    const uint32_t* order;
    uint32_t* point_list;
    uint2* ranges;
    unsigned char* pair_flag;
};
struct BlendFwdArgs {     // blend.hip
    const uint2* ranges;
    const uint32_t* point_list;
    const GRec* rec;
    int W, H, gx;
    uint32_t nblocks;
    const float* bg;
    float* out_color;
    float* final_T;
    uint32_t* n_contrib;
    int* touched;
    float* transmittance;
    unsigned long long* quad_masks;   // BinState::quad_masks (null: not kept)
    uint32_t* quad_depth;             // ImageState::quad_depth
    float4* ckpt;                     // BinState::ckpt (null: no backward will follow / segments off: nothing is left)
    const GeomHeader* hdr;            // hdr->ckpt, hdr->ckpt_thr: segment length and which lists get checkpoints
};
struct BlendBwdArgs {     // blend.hip
    const uint2* ranges;
    const uint32_t* point_list;
    const GRec* rec;
    const float* final_T;
    const uint32_t* n_contrib;
    const float* dL_dpix;
    int W, H, gx;
    uint32_t nblocks;
    const float* bg;
    float* pair_grad;  // [R][kPairGrad]: mx, my, cA, cB, cC, op, r, g, b per (tile, Gaussian) pair, emission order
    unsigned char* pair_flag;  // [R] set for rows written in this pass
    const unsigned long long* quad_masks;   // BinState::quad_masks as the forward left them (null: recompute)
    uint32_t* tile_order;                   // BinState::unit_order: launch order of the (tile, segment) units, heaviest
                                            // first (null: one workgroup per tile, row-major bands, one per XCD)
    uint32_t units_cap;                     // slots of the kOrderLists lists together (= workgroups launched)
    const uint32_t* quad_depth;             // ImageState::quad_depth, what the order is built from
    const float4* ckpt;                     // BinState::ckpt
    const GeomHeader* hdr;                  // hdr->ckpt / ckpt_thr: what the forward checkpointed (0: every tile is one unit)
    int segments;                           // 0: never split a list (r3dgs_set_bwd_segments(0))
};
struct PairReduceArgs {   // preprocess_bwd.hip
    const GeomHeader* hdr;
    const float* pair_grad;
    unsigned char* pair_flag;
    const uint32_t* pair_rank;
    uint32_t rank_mask;
    const uint32_t* order;
    const GRec* rec;
    const uint32_t* tiles;
    float* acc;
    float* wave_part;
    uint32_t stamp0, stamp1;   // this backward pass's stamp (process nonce, pass number): a run whose nine sums are all zero
                               // -- a third of the visible Gaussians of the metric scene: nothing contributed in any of their
                               // tiles -- stores NO row; a stored row carries the stamp in its pad words and the per-Gaussian
                               // kernel reads a row without this pass's stamp as zeros (preprocess_bwd.hip)
};
struct PreBwdArgs {       // preprocess_bwd.hip
    FwdInputs in;
    ViewParams view;
    const int* radii;
    const GRec* rec;
    const uint32_t* tiles;
    const float* acc;
    const float* wave_part;
    const float* pair_grad;   // -DR3_ACC_IN_SLAB (experiment): the run sums stay in the slab, in the row of the run's last pair
    uint32_t stamp0, stamp1;  // as PairReduceArgs
    const GeomHeader* header;
    float lambda_sh;
    const float* sh_ddir; // GeomState::sh_ddir when the backward may use it (no sparsity term, not switched off), else null
    int stagger;   // start-up delay step of the first generation of workgroups, in 64-clock units (preprocess_bwd.hip)
    BwdOutputs out;
};

// The device-resident argument block of a forward / backward pass.
struct FwdPassArgs {
    PreArgs pre;
    HeaderArgs header;
    DepthArgs depth;
    EmitArgs emit;
    RadixArgs radix[kMaxRadixPasses];
    RangesArgs ranges;
    BlendFwdArgs blend;
};
struct BwdPassArgs {
    BlendBwdArgs blend;
    PairReduceArgs reduce;
    PreBwdArgs pre;
};
static_assert(sizeof(FwdPassArgs) <= 3072 && sizeof(BwdPassArgs) <= 3072, "pass blocks travel as by-value kernel arguments");

// Host-side constants of a pass: everything that shapes the launch chain (and therefore keys the graph cache).
struct FwdPlan {
    int P, M, W, H, gx, gy;
    uint32_t reserve;      // pair capacity the binning blob was carved with (>= 1)
    uint32_t grid_pairs;   // pairs the pair-sized grids are launched for (their blocks stride beyond it): the pair count
                           // the reservation was derived from, without its slack
    PairLayout layout;
    int nb;                // depth buckets
    int ragged, counters;  // ragged SH addressing; counter mode (calculate_mean_transmittance)
    int color_in_geom;     // the geometry kernel's workgroups colour their own Gaussians (large scenes; preprocess.hip)
    int color_side;        // the colour kernel runs on a second stream beside the depth sort and the binning (capi.hip)
    int fwd_ppl;           // pixels per lane of the forward blend
    int color_grid;        // workgroups of the SH -> RGB stream (per launch that carries it)
    int color_fuse;        // 1: the colour chunks ride in spare workgroups of the depth-sort kernels
    int color_split[3];    // percent of the colour chunks in the histogram / scatter / bucket-sort launches
    int generic_depth_sort;  // rocPRIM sort + scan instead of the bucketed sort (never inside a graph)
    int tight;             // opacity-aware tile rects (default) or the reference's 3-sigma squares
};
struct BwdPlan {
    int P, M, W, H, gx, gy;
    uint32_t reserve, grid_pairs;
    uint32_t units_cap;    // workgroups of the backward blend when it runs in unit order (bwd_units_cap)
    PairLayout layout;
    int bwd_ppl;
    int has_pairs;         // 0: the forward ran with an empty reservation (P > 0, no binning blob)
    int f64_chain;         // the per-Gaussian backward evaluates the covariance chain in double (gauss_math.h; default)
};

// stage ids of the optional per-stage timers (capi.hip)
// kBlendBwdKernel: the backward blend kernel ALONE, inside kBlendBwd's events (which also cover the unit order and the pair
// reduction): the duration bench.py's roofline.kernel_frac divides by
enum Stage { kPre = 0, kDepthSort, kBinning, kBlendFwd, kBlendBwd, kPreBwd, kColor, kBlendBwdKernel, kNumStages };

// ---- per-stage issue functions (one per translation unit).  `a` points INTO the device pass block. ----------------
// The first kernel of a pass also installs the pass block: it receives the block BY VALUE (its workgroups read their own
// arguments straight from the kernarg segment) and workgroup 0 copies it to `dst` for the kernels that follow.
void issue_preprocess_geom(const FwdPlan& p, FwdPassArgs* dst, const FwdPassArgs& v, hipStream_t s);
void issue_preprocess_color(const FwdPlan& p, const PreArgs* a, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* view, bool* present, hipStream_t s);

void issue_header_reduce(const HeaderArgs* a, hipStream_t s);
void prepare_depth_bucket_sort(int nb);   // LDS opt-in of the depth-sort kernels, once per device (not a stream op)
void issue_depth_sort_and_color(const FwdPlan& p, const FwdPassArgs* a, hipStream_t s);   // preprocess.hip
void run_generic_depth_sort(int P, GeomState& g, hipStream_t s);   // rocPRIM, host pointers: direct issue only
void issue_tile_binning(const FwdPlan& p, const FwdPassArgs* a, hipStream_t s);
const char* sorted_words(const BinState& b, const PairLayout& l);   // which blob buffer holds the sorted words
void launch_export_keys(int P, int R, int cap, size_t n_tiles, const BinState& b, const GeomState& g, uint64_t* keys_out,
                        hipStream_t s);
void launch_export_point_list(int count, const uint32_t* point_list, const GeomHeader* hdr, uint32_t* out, hipStream_t s);

void issue_blend_forward(const FwdPlan& p, const BlendFwdArgs* a, hipStream_t s);
void issue_unit_order(const BwdPlan& p, BwdPassArgs* dst, const BwdPassArgs& v, hipStream_t s);   // installs the block (if it runs)
void issue_blend_backward(const BwdPlan& p, BwdPassArgs* dst, const BwdPassArgs& v, hipStream_t s);   // installs the block otherwise
void issue_pair_reduce(const BwdPlan& p, const PairReduceArgs* a, hipStream_t s);
void issue_preprocess_backward(const BwdPlan& p, const PreBwdArgs* a, hipStream_t s);

void launch_colour_variance_accumulate(int P, const int* D, int M, int max_sh_deg, const float* means3D,
                                       const float* cam_pos, const float* shs, const int* radii, const int* touched,
                                       const float* transmittance, float* wSum, float* wSumSq, float* mean,
                                       float* variance, float* accum, hipStream_t s);

int env_int(const char* env, int dflt, int lo, int hi);   // capi.hip

#if defined(__HIPCC__)
// Copies the pass block, which the calling kernel received as its SECOND by-value argument after a `Block* dst`, word by
// word out of the kernarg segment to *dst (taking the parameter's address instead would spill it to scratch first).
// Call from one workgroup; `nthreads` threads take part.
template <class Block>
__device__ __forceinline__ void install_block_from_kernarg(Block* dst, int tid, int nthreads)
{
    static_assert(alignof(Block) == 8 && sizeof(Block) % 4 == 0, "kernarg layout: [dst (8 B)][block]");
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const uint32_t __attribute__((address_space(4))) * KernargWords;
    KernargWords src = (KernargWords)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(Block*) / 4;
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    for (uint32_t k = (uint32_t)tid; k < sizeof(Block) / 4; k += (uint32_t)nthreads) d[k] = src[k];
#endif
}
#endif

}  // namespace r3
