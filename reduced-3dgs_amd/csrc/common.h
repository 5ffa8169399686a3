// common.h -- buffer layouts, launch helpers and error plumbing shared by the HIP translation units.
//
// The three caller-owned byte buffers play the role of the reference's GeometryState / BinningState /
// ImageState blobs (cuda_rasterizer/rasterizer_impl.h:21-73, rasterizer_impl.cu:163-202): produced by
// forward through allocator callbacks, kept alive by the caller, handed back verbatim to backward.
// Their internal layout is private to this library and is MI355X-first, not the reference's:
//   * one 48-byte gather record per Gaussian (GRec) instead of five separate arrays,
//   * depth-sorted Gaussian order + 32-bit tile keys instead of 64-bit (tile|depth) keys,
//   * a per-(tile, Gaussian)-pair gradient slab for the backward (no float atomics anywhere).
// The parts the backward needs sit at the FRONT of each blob so that their offsets do not depend on
// library temp-storage sizes.
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
constexpr int kAccStride = 12;  // floats per Gaussian in the reduced 2D-stage gradient row (9 used)

// Per-view counters.  Every preprocess workgroup stores one PrePartial (no atomics, nothing to pre-clear: the first
// GPU profile showed 7.8k same-address atomics costing ~90 us, a sharded version still needed a fill of the header
// in front of every pass); the depth-sort kernels reduce them, and one workgroup writes the header the host reads.
constexpr int kPreBlockSize = 256;
struct PrePartial {
    uint32_t visible;        // #Gaussians with radii > 0 in the workgroup (SH-sparsity normaliser, rasterizer_impl.cu:549-566)
    uint32_t num_rendered;   // sum of tiles_touched
    uint32_t depth_max;      // max depth bits over the visible Gaussians (0 if none)
    uint32_t depth_inv_min;  // max of ~(depth bits), i.e. ~min (0 if none)
};
constexpr int kOverflowSlots = 32;
struct GeomHeader {
    uint32_t visible, num_rendered, depth_max, depth_inv_min;
    uint32_t sort_overflow[kOverflowSlots];  // one per column-scan workgroup: a depth bucket exceeds kBucketCap
    uint32_t pad[28];
};
static_assert(sizeof(GeomHeader) == 256, "header = 256 B");
inline size_t pre_partials(size_t P) { return (P + kPreBlockSize - 1) / kPreBlockSize; }

// Bucketed depth sort (binning.hip): kDepthBuckets equal-width depth intervals between the view's min and max
// depth, each sorted by one workgroup in LDS.
constexpr int kDepthBuckets = 1024;
constexpr int kBucketCap = 4096;   // (key, id) pairs one workgroup sorts in LDS (32 KB)
constexpr int kHistPerBlock = 4096;  // Gaussians per histogram workgroup (16 per thread)
struct DepthSortScratch {
    uint32_t depth_max, depth_inv_min, pad_[2];   // depth range of the visible Gaussians (histogram workgroup 0)
    unsigned long long total[kDepthBuckets + 1];  // per bucket (tile sum << 24 | count); [kDepthBuckets] = culled
    uint32_t start[kDepthBuckets + 2];       // exclusive scan of the bucket sizes
    uint32_t tile_base[kDepthBuckets + 2];   // exclusive scan of the buckets' tiles_touched sums
};
inline size_t depth_hist_rows(size_t P) { return (P + kHistPerBlock - 1) / kHistPerBlock; }

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
    unsigned long long* hist_rows;  // [rows][kDepthBuckets + 1]  per-workgroup (tile sum << 24 | count) histograms
    uint32_t* hist_base;            // [rows][kDepthBuckets + 1]  first slot of each workgroup inside each bucket
    GRec* rec;            // [P]
    float* acc;           // [P * kAccStride]  per-Gaussian sums of the per-pair gradients (backward)
    ushort4* rect;        // [P]  tile rect (minx, miny, maxx, maxy)
    uint32_t* depth_key;  // [P]  float bits of view depth, 0xFFFFFFFF when culled
    uint32_t* tiles;      // [P]  tiles_touched
    uint32_t* key_sorted; // [P]  keys grouped by depth bucket (bucketed sort) / sorted keys (generic sort)
    uint32_t* bucket_id;  // [P]  Gaussian ids grouped by depth bucket
    uint32_t* order;      // [P]  Gaussian ids in (depth, id) order
    uint32_t* offsets;    // [P]  inclusive scan of tiles[order[j]]
    int* radii_internal;  // [P]  used when the caller passes radii == nullptr (rasterizer_impl.cu:393-396)
    char* temp;           // sort / scan temp storage
    size_t temp_bytes;
    static GeomState carve(char* base, size_t P, size_t temp_bytes)
    {
        Carver c(base);
        GeomState g;
        g.header = c.take<GeomHeader>(1);
        g.partials = c.take<PrePartial>(pre_partials(P));
        g.dsort = c.take<DepthSortScratch>(1);
        g.hist_rows = c.take<unsigned long long>(depth_hist_rows(P) * (kDepthBuckets + 1));
        g.hist_base = c.take<uint32_t>(depth_hist_rows(P) * (kDepthBuckets + 1));
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
        g.temp = c.take<char>(temp_bytes);
        g.temp_bytes = temp_bytes;
        g.end = c.p;
        return g;
    }
    char* end;
};

// Tile sort of the packed pair words (binning.hip): LSD radix, 7-bit digits, 1024 keys per workgroup.
constexpr int kRadixBits = 7;
constexpr int kRadixBins = 1 << kRadixBits;
constexpr int kRadixBlock = 1024;

struct BinState {
    uint32_t* point_list;  // [R] Gaussian ids, tile-major, (depth, id) order inside a tile
    float* pair_grad;      // [R * kPairGrad] per-pair gradients in EMISSION order (Gaussian-major), backward only
    float* wave_part;      // [(R/64+1) * 2 * kPairGrad] leading / trailing partial run sums of each 64-pair group
    unsigned char* pair_flag;  // [R] 1 = the backward blend wrote this pair's row (only this is zeroed per pass)
    uint32_t* tile_sorted; // [R] sorted keys: tile id (pair sort) or tile << rank_bits | depth rank (packed sort)
    uint32_t* tile_in;     // [R] the same keys in emission (Gaussian-major) order
    uint32_t* gauss_in;    // [R] Gaussian id per emitted pair (pair sort) / ping-pong buffer of the packed sort
    uint32_t* radix_rows;  // [kRadixBins][R/kRadixBlock + 1] per-workgroup digit counts, digit-major
    uint32_t* radix_base;  // same shape: first slot of each workgroup inside each digit
    uint32_t* radix_total; // [2][kRadixBins] digit totals of the two passes
    char* temp;
    size_t temp_bytes;
    char* end;
    static BinState carve(char* base, size_t R, size_t temp_bytes)
    {
        Carver c(base);
        BinState b;
        b.point_list = c.take<uint32_t>(R);
        b.pair_grad = c.take<float>(R * kPairGrad);
        b.wave_part = c.take<float>((R / 64 + 1) * 2 * kPairGrad);
        b.pair_flag = c.take<unsigned char>(R);
        b.tile_sorted = c.take<uint32_t>(R);
        b.tile_in = c.take<uint32_t>(R);
        b.gauss_in = c.take<uint32_t>(R);
        b.radix_rows = c.take<uint32_t>((size_t)kRadixBins * (R / kRadixBlock + 1));
        b.radix_base = c.take<uint32_t>((size_t)kRadixBins * (R / kRadixBlock + 1));
        b.radix_total = c.take<uint32_t>(2 * kRadixBins);
        b.temp = c.take<char>(temp_bytes);
        b.temp_bytes = temp_bytes;
        b.end = c.p;
        return b;
    }
};

struct ImageState {
    float* final_T;       // [N]
    uint32_t* n_contrib;  // [N]
    uint2* ranges;        // [Tn]
    char* end;
    static ImageState carve(char* base, size_t N, size_t Tn)
    {
        Carver c(base);
        ImageState s;
        s.final_T = c.take<float>(N);
        s.n_contrib = c.take<uint32_t>(N);
        s.ranges = c.take<uint2>(Tn);
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

// temp-storage queries (binning.hip)
size_t depth_sort_temp_bytes(size_t P);
size_t tile_sort_temp_bytes(size_t R);

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

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
// through wave-uniform (scalar) loads; only the plain scalars travel in the kernel argument.
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

// ---- per-stage host launchers (one per translation unit) ------------------------------------
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

void launch_preprocess(const FwdInputs& in, const ViewParams& view, GeomState& g, int* radii, hipStream_t s);
void launch_preprocess_color(const FwdInputs& in, const ViewParams& view, GeomState& g, hipStream_t s);
void launch_mark_visible(int P, const float* means3D, const float* view, bool* present, hipStream_t s);

// depth sort + scan; returns nothing (R is read back by the caller from g.offsets[P-1])
void run_header_reduce(int P, GeomState& g, hipStream_t s);                // generic path: partials -> header
void run_depth_sort_and_scan(int P, GeomState& g, hipStream_t s);          // generic (rocPRIM) path
void run_depth_histogram(int P, GeomState& g, bool header_ready, hipStream_t s);  // bucketed path, step 1 (sets sort_overflow)
void run_depth_bucket_sort_and_scan(int P, GeomState& g, hipStream_t s);   // bucketed path, steps 2-4
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
// Packed tile sort: when (tile bits + depth-rank bits) fit 32 bits the pairs travel as ONE word
// (tile << rank_bits | rank in depth order) through a key-only radix sort on the tile bits -- half the sort traffic;
// the Gaussian id is order[rank].  Returns rank_bits, or 0 for the (key, value) pair sort.  Forward and backward
// take the same decision from (P, #tiles) alone.  R3DGS_TILE_SORT=pairs forces the pair sort.
int tile_rank_bits(int P, size_t n_tiles);
void run_tile_binning(int P, int R, int gx, int gy, GeomState& g, BinState& b, ImageState& img, hipStream_t s);
void launch_export_keys(int P, int R, size_t n_tiles, const BinState& b, const GeomState& g, uint64_t* keys_out,
                        hipStream_t s);

void launch_blend_forward(const ViewParams& view, const GeomState& g, const BinState& b,
                          ImageState& img, float* out_color, int* touched, float* transmittance, hipStream_t s);
void launch_blend_backward(const ViewParams& view, const GeomState& g, BinState& b,
                           const ImageState& img, const float* dL_dpix, hipStream_t s);

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
void launch_colour_variance_accumulate(int P, const int* D, int M, int max_sh_deg, const float* means3D,
                                       const float* cam_pos, const float* shs, const int* radii, const int* touched,
                                       const float* transmittance, float* wSum, float* wSumSq, float* mean,
                                       float* variance, float* accum, hipStream_t s);
void launch_pair_reduce(int P, int R, size_t n_tiles, const GeomState& g, const BinState& b, hipStream_t s);
void launch_preprocess_backward(const FwdInputs& in, const ViewParams& view, const int* radii, const GeomState& g,
                                const BinState& b, const BwdOutputs& out, float lambda_sh_sparsity, hipStream_t s);

}  // namespace r3
