// capi.hip -- the extern "C" boundary declared in include/r3dgs_rasterizer.h.
//
// Orchestration of one forward / backward pass; replaces CudaRasterizer::Rasterizer::{forward,
// inferenceForward, backward, markVisible} (cuda_rasterizer/rasterizer_impl.cu:149-161, :206-355,
// :359-504, :508-630 of /root/reference/submodules/diff-gaussian-rasterization).
// Differences in how the work is issued (results are the same):
//   * everything runs on the caller's HIP stream; the only host synchronisation is the read-back of
//     num_rendered, which sizes the caller-owned binning blob (same structural sync as the reference, but
//     R is produced by the first kernel and fetched on a side stream while the depth sort runs, so the GPU
//     does not idle during the host round trip);
//   * the SH -> RGB kernel (the forward's HBM-heavy stream) runs on that side stream too, underneath the
//     launch-latency-bound sorts;
//   * no per-call hipMalloc/hipFree: all scratch lives in the three caller blobs;
//   * `debug` makes every stage synchronise and surface its error (the reference's CHECK_CUDA).
#include "../../include/r3dgs_rasterizer.h"

#include <cstddef>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "common.h"

namespace {

thread_local std::string g_last_error;

size_t cached_depth_temp(size_t P)
{
    static std::mutex mu;
    static std::unordered_map<size_t, size_t> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(P);
    if (it != cache.end()) return it->second;
    size_t b = r3::depth_sort_temp_bytes(P);
    cache[P] = b;
    return b;
}

// tile-sort temp grows with R; query on a rounded-up size so the cache stays small
size_t round_up_R(size_t R)
{
    size_t g = 1 << 16;
    return ((R + g - 1) / g) * g;
}
size_t cached_tile_temp(size_t R)
{
    static std::mutex mu;
    static std::unordered_map<size_t, size_t> cache;
    std::lock_guard<std::mutex> lk(mu);
    const size_t key = round_up_R(R ? R : 1);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    size_t b = r3::tile_sort_temp_bytes(key);
    cache[key] = b;
    return b;
}

// ---- optional per-stage timing with HIP events on the caller's stream (r3dgs_profile_*) ----------
enum Stage { kPre = 0, kDepthSort, kBinning, kBlendFwd, kBlendBwd, kPreBwd, kColor, kNumStages };
struct Profiler {
    std::mutex mu;
    unsigned mask = 0;   // bit (stage): record an event pair around that stage
    std::vector<std::pair<hipEvent_t, hipEvent_t>> used[kNumStages];
    std::vector<hipEvent_t> pool;
    hipEvent_t get()
    {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        R3_HIP(hipEventCreate(&e));
        return e;
    }
} g_prof;

struct StageTimer {
    int stage;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    StageTimer(int stage_, hipStream_t s_) : stage(stage_), s(s_)
    {
        if (!((g_prof.mask >> stage_) & 1u)) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        a = g_prof.get();
        b = g_prof.get();
        R3_HIP(hipEventRecord(a, s));
    }
    void stop()
    {
        if (!a) return;
        R3_HIP(hipEventRecord(b, s));
        std::lock_guard<std::mutex> lk(g_prof.mu);
        g_prof.used[stage].push_back({a, b});
        a = nullptr;
    }
};

// Host-side resources for the num_rendered read-back: a non-blocking side stream, two events and one
// pinned word, per (host thread, device).  The copy is ordered after the preprocess kernel by an event and
// runs beside the depth sort, so the structural host round trip of the forward is hidden behind GPU work.
struct ReadbackCtx {
    hipStream_t side = nullptr, side2 = nullptr;
    hipEvent_t after_pre = nullptr, copied = nullptr, colored = nullptr, after_hist = nullptr, copied2 = nullptr;
    hipEvent_t geom_done = nullptr;
    r3::GeomHeader* pinned = nullptr;
};
ReadbackCtx& readback_ctx()
{
    thread_local std::unordered_map<int, ReadbackCtx> per_device;
    int dev = 0;
    R3_HIP(hipGetDevice(&dev));
    ReadbackCtx& c = per_device[dev];
    if (!c.side) {
        // the SH -> RGB kernel on `side` is bandwidth-heavy filler under the latency-bound sort kernels of the
        // caller's stream: lowest priority, so that their workgroups are dispatched first (at equal priority the
        // depth scatter kernel took 46 us instead of 14 us next to it)
        int prio_low = 0, prio_high = 0;
        R3_HIP(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
        R3_HIP(hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, prio_low));
        R3_HIP(hipStreamCreateWithPriority(&c.side2, hipStreamNonBlocking, prio_high));
        R3_HIP(hipEventCreateWithFlags(&c.geom_done, hipEventDisableTiming));
        R3_HIP(hipEventCreateWithFlags(&c.after_hist, hipEventDisableTiming));
        R3_HIP(hipEventCreateWithFlags(&c.copied2, hipEventDisableTiming));
        R3_HIP(hipEventCreateWithFlags(&c.after_pre, hipEventDisableTiming));
        R3_HIP(hipEventCreateWithFlags(&c.copied, hipEventDisableTiming));
        R3_HIP(hipEventCreateWithFlags(&c.colored, hipEventDisableTiming));
        R3_HIP(hipHostMalloc(reinterpret_cast<void**>(&c.pinned), sizeof(r3::GeomHeader), hipHostMallocDefault));
    }
    return c;
}

template <class F>
int guarded(F&& f)
{
    try {
        g_last_error.clear();
        return f();
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return -1;
    }
}

int forward_impl(r3dgs_alloc_fn geometryBuffer, void* geometry_user, r3dgs_alloc_fn binningBuffer, void* binning_user,
                 r3dgs_alloc_fn imageBuffer, void* image_user, int P, const int* D, int M, const int* coeffsNum,
                 const int* perBand, const int* cumSum, const float* background, int width, int height,
                 const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                 const float* scales, float scale_modifier, const float* rotations, const float* cov3D_precomp,
                 const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                 float* out_color, int* out_touched_pixels, float* out_transmittance, int* radii,
                 int calculate_mean_transmittance, int debug, void* stream)
{
    using namespace r3;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (P <= 0) return 0;
    if (!geometryBuffer || !binningBuffer || !imageBuffer) throw Error("allocator callbacks must not be NULL");
    if (!means3D || !opacities || !viewmatrix || !projmatrix || !cam_pos || !background || !out_color)
        throw Error("a required pointer is NULL");
    if (!colors_precomp && !shs) throw Error("provide SHs or precomputed colours");
    if (!cov3D_precomp && (!scales || !rotations)) throw Error("provide scale/rotation or a precomputed 3D covariance");
    if (!colors_precomp && !coeffsNum && (M < 1 || M > 16)) throw Error("SH coefficient count M must be in [1,16]");
    if (!colors_precomp && !coeffsNum && !D) throw Error("per-Gaussian degrees must be provided with SHs");
    if (width <= 0 || height <= 0) throw Error("image size must be positive");
    if (calculate_mean_transmittance && (!out_touched_pixels || !out_transmittance))
        throw Error("counter mode needs out_touched_pixels and out_transmittance");

    const int gx = (width + kTile - 1) / kTile, gy = (height + kTile - 1) / kTile;
    const size_t depth_temp = cached_depth_temp((size_t)P);
    char* gptr = geometryBuffer(required_bytes<GeomState>((size_t)P, depth_temp), geometry_user);
    if (!gptr) throw Error("geometry allocator returned NULL");
    GeomState geom = GeomState::carve(gptr, (size_t)P, depth_temp);
    char* iptr = imageBuffer(required_bytes<ImageState>((size_t)width * height, (size_t)gx * gy), image_user);
    if (!iptr) throw Error("image allocator returned NULL");
    ImageState img = ImageState::carve(iptr, (size_t)width * height, (size_t)gx * gy);
    if (!radii) radii = geom.radii_internal;

    // (Clearing the tile ranges / pair flags on the side stream instead of the main one was tried: each cross-stream
    // wait costs the main queue more than the ~5 us fill it saves -- step time went up by ~30 us.)
    ReadbackCtx& rb = readback_ctx();
    static const bool generic_env = [] {   // R3DGS_DEPTH_SORT=generic forces the rocPRIM path (A/B runs, tests)
        const char* v = getenv("R3DGS_DEPTH_SORT");
        return v && std::string(v) == "generic";
    }();
    const bool generic_sort = generic_env || P >= (1 << 24);   // the bucket histogram packs the count in 24 bits

    FwdInputs in;
    in.P = P;
    in.M = M;
    in.degrees = D;
    in.means3D = means3D;
    in.scales = scales;
    in.rotations = rotations;
    in.opacities = opacities;
    in.shs = shs;
    in.cov3D_precomp = cov3D_precomp;
    in.colors_precomp = colors_precomp;
    in.coeffs_num = coeffsNum;
    in.per_band_count = perBand;
    in.cumsum_count = cumSum;
    ViewParams view;
    view.view = viewmatrix;
    view.proj = projmatrix;
    view.campos = cam_pos;
    view.bg = background;
    view.tan_fovx = tan_fovx;
    view.tan_fovy = tan_fovy;
    view.W = width;
    view.H = height;
    view.scale_modifier = scale_modifier;

    StageTimer t0(kPre, s);
    launch_preprocess(in, view, geom, radii, s);
    t0.stop();
    check_launch("preprocess", s, debug);
    // The SH -> RGB kernel needs only the geometry kernel's visibility: it starts now on the low-priority side stream,
    // underneath the (latency-bound) header / depth-sort kernels of the main stream.
    R3_HIP(hipEventRecord(rb.geom_done, s));
    R3_HIP(hipStreamWaitEvent(rb.side, rb.geom_done, 0));
    {
        StageTimer tc(kColor, rb.side);
        launch_preprocess_color(in, view, geom, rb.side);
        tc.stop();
        R3_HIP(hipEventRecord(rb.colored, rb.side));
    }
    // The header the host needs: one workgroup turns the preprocess partials into num_rendered / visible count and
    // the 16 bytes start their way to the host on the copy stream.  R does not depend on the depth order, so the
    // structural host round trip -- size the binning blob, then enqueue the binning -- overlaps the whole depth sort.
    // (Running the reduction on the copy stream as well keeps 12 us off the main chain but delays R by the
    // cross-queue latency: 991 vs 1010 it/s on the same box, R3DGS_HEADER_SIDE=1 selects it.)
    static const bool header_on_main = [] {
        const char* v = getenv("R3DGS_HEADER_SIDE");
        return !(v && v[0] == '1');
    }();
    if (header_on_main) {
        run_header_reduce(P, geom, s);
        R3_HIP(hipEventRecord(rb.after_pre, s));
        R3_HIP(hipStreamWaitEvent(rb.side2, rb.after_pre, 0));
    } else {
        R3_HIP(hipStreamWaitEvent(rb.side2, rb.geom_done, 0));
        run_header_reduce(P, geom, rb.side2);
    }
    R3_HIP(hipMemcpyAsync(rb.pinned, geom.header, offsetof(GeomHeader, sort_overflow), hipMemcpyDeviceToHost, rb.side2));
    R3_HIP(hipEventRecord(rb.copied, rb.side2));
    StageTimer t1(kDepthSort, s);
    if (generic_sort) {
        run_depth_sort_and_scan(P, geom, s);
    } else {
        run_depth_histogram(P, geom, header_on_main, s);
        // the bucket-overflow flags follow on the same copy stream
        R3_HIP(hipEventRecord(rb.after_hist, s));
        R3_HIP(hipStreamWaitEvent(rb.side2, rb.after_hist, 0));
        R3_HIP(hipMemcpyAsync(rb.pinned->sort_overflow, geom.header->sort_overflow, sizeof(uint32_t) * kOverflowSlots,
                              hipMemcpyDeviceToHost, rb.side2));
        R3_HIP(hipEventRecord(rb.copied2, rb.side2));
        run_depth_bucket_sort_and_scan(P, geom, s);
    }
    t1.stop();
    // spin on the event instead of hipEventSynchronize: a blocking wait parks the host thread, and on an otherwise
    // idle many-core host its wake-up (deep C-state exit) was observed to cost more than the whole forward
    auto spin = [](hipEvent_t ev) {
        for (;;) {
            const hipError_t q = hipEventQuery(ev);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) R3_HIP(q);
        }
    };
    spin(rb.copied);
    const uint32_t R = rb.pinned->num_rendered;
    if (R > 0x7fffffffu) throw Error("num_rendered exceeds 2^31-1");
    const size_t tile_temp = cached_tile_temp(R);
    char* bptr = binningBuffer(required_bytes<BinState>((size_t)R, tile_temp), binning_user);   // host work, overlapped
    if (!bptr) throw Error("binning allocator returned NULL");
    BinState bin = BinState::carve(bptr, (size_t)R, tile_temp);
    if (!generic_sort) {
        spin(rb.copied2);
        bool overflow = false;
        for (int k = 0; k < kOverflowSlots; k++) overflow |= rb.pinned->sort_overflow[k] != 0;
        if (overflow) {
            // a depth bucket did not fit one workgroup's LDS (many splats at one depth): redo with the generic sort
            StageTimer t1b(kDepthSort, s);
            run_depth_sort_and_scan(P, geom, s);
            t1b.stop();
        }
    }
    check_launch("depth sort + scan", s, debug);

    StageTimer t2(kBinning, s);
    run_tile_binning(P, (int)R, gx, gy, geom, bin, img, s);
    t2.stop();
    check_launch("tile binning", s, debug);
    R3_HIP(hipStreamWaitEvent(s, rb.colored, 0));  // the blend needs the colours
    StageTimer t3(kBlendFwd, s);
    launch_blend_forward(view, geom, bin, img, out_color, calculate_mean_transmittance ? out_touched_pixels : nullptr,
                         calculate_mean_transmittance ? out_transmittance : nullptr, s);
    t3.stop();
    check_launch("blend forward", s, debug);
    if (debug) R3_HIP(hipStreamSynchronize(rb.side));
    return (int)R;
}

}  // namespace

int r3::guarded_call(const std::function<int()>& f) { return guarded(f); }

extern "C" {

const char* r3dgs_version(void) { return "r3dgs-hip gfx950 0.1"; }
const char* r3dgs_last_error(void) { return g_last_error.c_str(); }

// The temp-storage part of the blob sizes comes from rocPRIM queries, which need a visible GPU;
// without one these return 0 and set r3dgs_last_error().
size_t r3dgs_geometry_bytes(int P)
{
    try {
        g_last_error.clear();
        return r3::required_bytes<r3::GeomState>((size_t)P, cached_depth_temp((size_t)(P > 0 ? P : 1)));
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return 0;
    }
}
size_t r3dgs_binning_bytes(int R)
{
    try {
        g_last_error.clear();
        return r3::required_bytes<r3::BinState>((size_t)R, cached_tile_temp((size_t)(R > 0 ? R : 1)));
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return 0;
    }
}
size_t r3dgs_image_bytes(int width, int height)
{
    const int gx = (width + r3::kTile - 1) / r3::kTile, gy = (height + r3::kTile - 1) / r3::kTile;
    return r3::required_bytes<r3::ImageState>((size_t)width * height, (size_t)gx * gy);
}

int r3dgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                       unsigned char* present, void* stream)
{
    (void)projmatrix;  // unused by the reference too (rasterizer_impl.cu:62-74)
    return guarded([&]() {
        if (P <= 0) return 0;
        if (!means3D || !viewmatrix || !present) throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        r3::launch_mark_visible(P, means3D, viewmatrix, reinterpret_cast<bool*>(present), s);
        r3::check_launch("mark_visible", s, false);
        return 0;
    });
}

int r3dgs_forward(r3dgs_alloc_fn geometryBuffer, void* geometry_user, r3dgs_alloc_fn binningBuffer, void* binning_user,
                  r3dgs_alloc_fn imageBuffer, void* image_user, int P, const int* D, int M, const float* background,
                  int width, int height, const float* means3D, const float* shs, const float* colors_precomp,
                  const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                  const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix, const float* cam_pos,
                  float tan_fovx, float tan_fovy, int prefiltered, float* out_color, int* out_touched_pixels,
                  float* out_transmittance, int* radii, int calculate_mean_transmittance, int debug, void* stream)
{
    (void)prefiltered;
    return guarded([&]() {
        return forward_impl(geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user, P, D, M,
                            nullptr, nullptr, nullptr, background, width, height, means3D, shs, colors_precomp,
                            opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix, projmatrix, cam_pos,
                            tan_fovx, tan_fovy, out_color, out_touched_pixels, out_transmittance, radii,
                            calculate_mean_transmittance, debug, stream);
    });
}

int r3dgs_inference_forward(r3dgs_alloc_fn geometryBuffer, void* geometry_user, r3dgs_alloc_fn binningBuffer,
                            void* binning_user, r3dgs_alloc_fn imageBuffer, void* image_user, int P, const int* D,
                            int bandsNum, const int* coeffsNum, const int* perBandPrimitiveCount,
                            const int* cumSumPrimitiveCount, const float* background, int width, int height,
                            const float* means3D, const float* shs, const float* colors_precomp,
                            const float* opacities, const float* scales, float scale_modifier,
                            const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                            const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                            int prefiltered, float* out_color, int* out_touched_pixels, float* out_transmittance,
                            int* radii, int calculate_mean_transmittance, int debug, void* stream)
{
    (void)prefiltered;
    return guarded([&]() {
        if (!colors_precomp) {
            if (bandsNum != 4) throw r3::Error("ragged SH path expects 4 bands (degrees 0..3)");
            if (!coeffsNum || !perBandPrimitiveCount || !cumSumPrimitiveCount)
                throw r3::Error("ragged SH path needs coeffsNum / perBandPrimitiveCount / cumSumPrimitiveCount");
        } else {
            coeffsNum = perBandPrimitiveCount = cumSumPrimitiveCount = nullptr;
        }
        return forward_impl(geometryBuffer, geometry_user, binningBuffer, binning_user, imageBuffer, image_user, P, D, 16,
                            coeffsNum, perBandPrimitiveCount, cumSumPrimitiveCount, background, width, height, means3D,
                            shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp, viewmatrix,
                            projmatrix, cam_pos, tan_fovx, tan_fovy, out_color, out_touched_pixels, out_transmittance,
                            radii, calculate_mean_transmittance, debug, stream);
    });
}

int r3dgs_backward(int P, const int* D, int M, int R, const float* background, int width, int height,
                   const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                   float scale_modifier, const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                   const float* projmatrix, const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                   char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix, float* dL_dmean2D,
                   float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                   float* dL_dsh, float* dL_dscale, float* dL_drot, float lambda_sh_sparsity, int debug, void* stream)
{
    return guarded([&]() {
        using namespace r3;
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (P <= 0) return 0;
        if (!geom_buffer || !image_buffer || (R > 0 && !binning_buffer)) throw Error("state buffers must not be NULL");
        if (!means3D || !viewmatrix || !projmatrix || !campos || !background || !dL_dpix)
            throw Error("a required pointer is NULL");
        if (!dL_dmean2D || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D || !dL_dscale || !dL_drot)
            throw Error("a gradient output pointer is NULL");
        if (shs && (!dL_dsh || !D || M < 1 || M > 16)) throw Error("SH gradients need dL_dsh, degrees and 1 <= M <= 16");
        const int gx = (width + kTile - 1) / kTile, gy = (height + kTile - 1) / kTile;
        GeomState geom = GeomState::carve(geom_buffer, (size_t)P, cached_depth_temp((size_t)P));
        ImageState img = ImageState::carve(image_buffer, (size_t)width * height, (size_t)gx * gy);
        BinState bin = BinState::carve(binning_buffer, (size_t)R, cached_tile_temp((size_t)R));
        if (R <= 0) bin.pair_grad = bin.wave_part = nullptr;
        if (!radii) radii = geom.radii_internal;

        FwdInputs in;
        in.P = P;
        in.M = M;
        in.degrees = D;
        in.means3D = means3D;
        in.scales = scales;
        in.rotations = rotations;
        in.opacities = nullptr;
        in.shs = colors_precomp ? nullptr : shs;
        in.cov3D_precomp = cov3D_precomp;
        in.colors_precomp = colors_precomp;
        in.coeffs_num = in.per_band_count = in.cumsum_count = nullptr;
        ViewParams view;
        view.view = viewmatrix;
        view.proj = projmatrix;
        view.campos = campos;
        view.bg = background;
        view.tan_fovx = tan_fovx;
        view.tan_fovy = tan_fovy;
        view.W = width;
        view.H = height;
        view.scale_modifier = scale_modifier;

        StageTimer t4(kBlendBwd, s);
        if (R > 0) {
            // bin.pair_flag is all zero here: the forward's tile_ranges kernel clears it and pair_reduce puts every
            // flag it consumed back to zero, so neither pass pays for a fill of its own
            launch_blend_backward(view, geom, bin, img, dL_dpix, s);
            launch_pair_reduce(P, R, (size_t)gx * gy, geom, bin, s);
        }
        t4.stop();
        check_launch("blend backward", s, debug);
        BwdOutputs out;
        out.dL_dmean2D = dL_dmean2D;
        out.dL_dopacity = dL_dopacity;
        out.dL_dcolor = dL_dcolor;
        out.dL_dmean3D = dL_dmean3D;
        out.dL_dcov3D = dL_dcov3D;
        out.dL_dsh = dL_dsh;
        out.dL_dscale = dL_dscale;
        out.dL_drot = dL_drot;
        out.dL_dconic = dL_dconic;
        StageTimer t5(kPreBwd, s);
        launch_preprocess_backward(in, view, radii, geom, bin, out, lambda_sh_sparsity, s);
        t5.stop();
        check_launch("preprocess backward", s, debug);
        return 0;
    });
}

int r3dgs_colour_variance_accumulate(int P, const int* D, int M, int max_sh_deg, const float* means3D,
                                     const float* cam_pos, const float* shs, const int* radii,
                                     const int* touched_pixels, const float* transmittance, float* wSum, float* wSumSq,
                                     float* mean, float* variance, float* colourDistancesAccum, void* stream)
{
    return guarded([&]() {
        if (P <= 0) return 0;
        if (max_sh_deg < 1 || max_sh_deg > 3) throw r3::Error("max_sh_deg must be in [1,3]");
        if (M < (max_sh_deg + 1) * (max_sh_deg + 1) || M > 16) throw r3::Error("SH tensor too small for max_sh_deg");
        if (!D || !means3D || !cam_pos || !shs || !radii || !touched_pixels || !transmittance || !wSum || !wSumSq ||
            !mean || !variance || !colourDistancesAccum)
            throw r3::Error("a required pointer is NULL");
        hipStream_t s = static_cast<hipStream_t>(stream);
        r3::launch_colour_variance_accumulate(P, D, M, max_sh_deg, means3D, cam_pos, shs, radii, touched_pixels,
                                              transmittance, wSum, wSumSq, mean, variance, colourDistancesAccum, s);
        r3::check_launch("colour variance accumulate", s, false);
        return 0;
    });
}

int r3dgs_profile_enable(int on)
{
    // on == 0: off; on == 1: every stage; otherwise bit (s + 1) of `on` selects stage s alone -- each event record is
    // a packet on the stream, and a dozen of them per pass cost a few percent of a 1 ms iteration
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.mask = on == 0 ? 0u : (on == 1 ? ~0u : ((unsigned)on >> 1));
    return 0;
}

int r3dgs_profile_stage_count(void) { return kNumStages; }

const char* r3dgs_profile_stage_name(int stage)
{
    static const char* names[kNumStages] = {"preprocess_fwd", "depth_sort_scan", "tile_binning", "blend_fwd",
                                            "blend_bwd",      "preprocess_bwd",  "sh_color_overlapped"};
    return (stage >= 0 && stage < kNumStages) ? names[stage] : "";
}

int r3dgs_profile_read(double* total_ms, int* launches)
{
    return guarded([&]() {
        std::lock_guard<std::mutex> lk(g_prof.mu);
        for (int st = 0; st < kNumStages; st++) {
            double sum = 0.0;
            for (auto& ev : g_prof.used[st]) {
                R3_HIP(hipEventSynchronize(ev.second));
                float ms = 0.f;
                R3_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
                sum += ms;
                g_prof.pool.push_back(ev.first);
                g_prof.pool.push_back(ev.second);
            }
            if (total_ms) total_ms[st] = sum;
            if (launches) launches[st] = (int)g_prof.used[st].size();
            g_prof.used[st].clear();
        }
        return 0;
    });
}

int r3dgs_export_binning(int P, int R, int width, int height, char* geom_buffer, char* binning_buffer,
                         char* image_buffer, uint64_t* keys, uint32_t* point_list, uint32_t* ranges, uint32_t* n_contrib,
                         float* final_T, uint32_t* tiles_touched, void* stream)
{
    return guarded([&]() {
        using namespace r3;
        hipStream_t s = static_cast<hipStream_t>(stream);
        if (P <= 0) return 0;
        const int gx = (width + kTile - 1) / kTile, gy = (height + kTile - 1) / kTile;
        const size_t N = (size_t)width * height, Tn = (size_t)gx * gy;
        GeomState geom = GeomState::carve(geom_buffer, (size_t)P, cached_depth_temp((size_t)P));
        ImageState img = ImageState::carve(image_buffer, N, Tn);
        if (R > 0) {
            BinState bin = BinState::carve(binning_buffer, (size_t)R, cached_tile_temp((size_t)R));
            if (keys) launch_export_keys(P, R, Tn, bin, geom, keys, s);
            if (point_list)
                R3_HIP(hipMemcpyAsync(point_list, bin.point_list, sizeof(uint32_t) * (size_t)R, hipMemcpyDeviceToDevice, s));
        }
        if (ranges) R3_HIP(hipMemcpyAsync(ranges, img.ranges, sizeof(uint2) * Tn, hipMemcpyDeviceToDevice, s));
        if (n_contrib) R3_HIP(hipMemcpyAsync(n_contrib, img.n_contrib, sizeof(uint32_t) * N, hipMemcpyDeviceToDevice, s));
        if (final_T) R3_HIP(hipMemcpyAsync(final_T, img.final_T, sizeof(float) * N, hipMemcpyDeviceToDevice, s));
        if (tiles_touched)
            R3_HIP(hipMemcpyAsync(tiles_touched, geom.tiles, sizeof(uint32_t) * (size_t)P, hipMemcpyDeviceToDevice, s));
        check_launch("export", s, false);
        return 0;
    });
}

}  // extern "C"
