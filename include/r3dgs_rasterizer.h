/*
 * r3dgs_rasterizer.h -- C ABI of the MI355X-native differentiable Gaussian-splatting rasterizer
 * (shared library libr3dgs_hip.so, built from reduced-3dgs_amd/csrc/ for gfx950).
 *
 * This is the drop-in boundary for the hot path of graphdeco-inria/reduced-3dgs: each entry point
 * replaces one static method of `CudaRasterizer::Rasterizer`
 * (submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer.h:20-117), with the same argument
 * order and meaning, expressed in plain C:
 *   - `std::function<char*(size_t)>` allocator callbacks  ->  r3dgs_alloc_fn + user pointer
 *   - `bool`                                               ->  int (0 / 1)
 *   - an explicit HIP stream (`void*` = hipStream_t; the reference uses the legacy default stream)
 *   - errors: the reference throws std::runtime_error; here functions return a negative status and
 *     r3dgs_last_error() returns the message (thread-local).  A host shim turns that into the
 *     exception its language expects.
 * All pointers are DEVICE pointers (fp32 / int32, contiguous) unless stated otherwise; optional inputs
 * are NULL when absent (the reference tests `ptr != nullptr`, forward.cu:403,441).
 * No torch / HIP types appear in any signature.
 */
#ifndef R3DGS_RASTERIZER_H
#define R3DGS_RASTERIZER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Resizable-buffer callback: must return a device pointer to at least `bytes` bytes that stays valid
 * until the matching r3dgs_backward call.  Replaces resizeFunctional() of rasterize_points.cu:33-41. */
typedef char* (*r3dgs_alloc_fn)(size_t bytes, void* user);

/* Library identification string, e.g. "r3dgs-hip gfx950 0.1". Host memory, never freed. */
const char* r3dgs_version(void);

/* Message of the last failed call on this thread ("" if none). Host memory owned by the library. */
const char* r3dgs_last_error(void);

/* Sizes of the three opaque state blobs (the reference's required<GeometryState>(P) etc.,
 * rasterizer_impl.h:66-72).  Layouts are private to the library.  The binning blob is sized by a pair capacity
 * (`reserve`: num_rendered for an exact-size pass, the reservation for r3dgs_forward_reserved); its layout also
 * depends on (P, width, height) -- 64-bit pair words once tile bits + log2(P) exceed 32.
 * r3dgs_binning_capacity inverts r3dgs_binning_bytes: the largest capacity whose layout fits `bytes` (it
 * reproduces the carve of the pass that sized the blob), 0 if none fits, negative on error.
 * r3dgs_geometry_bytes_lean: the geometry blob of a forward that leaves no SH direction derivatives for a backward --
 * r3dgs_inference_forward*, precomputed colours, or a forward announced with r3dgs_forward_hint(0): 36 bytes per Gaussian
 * less (the array is the blob's last one; the pass header says whether it was written, nothing else reads it). */
size_t r3dgs_geometry_bytes(int P);
size_t r3dgs_geometry_bytes_lean(int P);
size_t r3dgs_binning_bytes(int P, int width, int height, int reserve);
size_t r3dgs_image_bytes(int width, int height);
int r3dgs_binning_capacity(int P, int width, int height, size_t bytes);

/* Rasterizer::markVisible (rasterizer.h:25-30, rasterizer_impl.cu:149-161): present[i] = view-space z > 0.2.
 * `present` is a device array of P bytes (0/1).  Returns 0 or a negative status. */
int r3dgs_mark_visible(int P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                       unsigned char* present, void* stream);

/* Rasterizer::forward (rasterizer.h:31-56, rasterizer_impl.cu:359-504).
 * Returns num_rendered (>= 0) or a negative status.  Exact-size contract of the reference: the binning blob is
 * requested through the callback once the pair count is known and is sized for the RETURNED num_rendered (the pairs
 * actually binned, r3dgs_forward_pairs, are fewer when the opacity-aware rects left tiles out), so the returned value is
 * the R r3dgs_backward / r3dgs_export_binning expect, as in the reference; i.e. this entry point waits for that one number
 * (the reference's cudaMemcpy at rasterizer_impl.cu:446) -- by polling host-mapped memory the pass writes, with a
 * deadline (R3DGS_SYNC_TIMEOUT_MS).  Training loops should use r3dgs_forward_reserved, which never waits.
 *   D        : per-Gaussian SH degree [P] (int32)
 *   M        : SH coefficients per Gaussian in the dense `shs` tensor [P, M, 3] (<= 16)
 *   opacities: RAW (pre-sigmoid); scales: ACTIVATED; rotations: UNIT quaternions (r,x,y,z)
 *   viewmatrix / projmatrix: 4x4 in the reference's transposed layout (scene/cameras.py:54-56)
 *   out_color: [3, H, W], fully written;  radii: [P] int32 or NULL
 *   out_touched_pixels [P] int32 / out_transmittance [P] fp32: accumulated into (caller zeroes) when
 *     calculate_mean_transmittance != 0 (forward.cu:560-564), else ignored
 *   prefiltered: accepted for signature parity and ignored (the reference only uses it to trap) */
int r3dgs_forward(r3dgs_alloc_fn geometryBuffer, void* geometry_user, r3dgs_alloc_fn binningBuffer,
                  void* binning_user, r3dgs_alloc_fn imageBuffer, void* image_user, int P, const int* D, int M,
                  const float* background, int width, int height, const float* means3D, const float* shs,
                  const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                  const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                  const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                  float* out_color, int* out_touched_pixels, float* out_transmittance, int* radii,
                  int calculate_mean_transmittance, int debug, void* stream);

/* Rasterizer::inferenceForward (rasterizer.h:89-116, rasterizer_impl.cu:206-355): forward with the ragged,
 * degree-sorted SH buffer of the variable-SH-band inference path (forward.cu:19-36).
 * coeffsNum / perBandPrimitiveCount / cumSumPrimitiveCount: device int32[bandsNum], bandsNum == 4. */
int r3dgs_inference_forward(r3dgs_alloc_fn geometryBuffer, void* geometry_user, r3dgs_alloc_fn binningBuffer,
                            void* binning_user, r3dgs_alloc_fn imageBuffer, void* image_user, int P, const int* D,
                            int bandsNum, const int* coeffsNum, const int* perBandPrimitiveCount,
                            const int* cumSumPrimitiveCount, const float* background, int width, int height,
                            const float* means3D, const float* shs, const float* colors_precomp,
                            const float* opacities, const float* scales, float scale_modifier,
                            const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                            const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                            int prefiltered, float* out_color, int* out_touched_pixels, float* out_transmittance,
                            int* radii, int calculate_mean_transmittance, int debug, void* stream);

/* ---- asynchronous forward (extension; the hot path) -------------------------------------------------------------
 * The reference's forward blocks the host in the middle of every pass to size the binning buffer
 * (rasterizer_impl.cu:441-450).  r3dgs_forward_reserved removes that: the caller passes the three blobs up front
 * (r3dgs_geometry_bytes / r3dgs_binning_bytes(..., reserve) / r3dgs_image_bytes) and a pair reservation; everything
 * is enqueued on `stream` -- as ONE hipGraph launch per pass once the shape has been seen -- and the call returns
 * a pass ticket (> 0) without waiting.  The pair count lives on the device; if it exceeds `reserve` the FARTHEST
 * pairs are dropped for that pass (emission is in depth order) and the pass is flagged R3DGS_PASS_TRUNCATED: a caller
 * that wants the reference's results MUST look at the flag before the pass's outputs are consumed and redo a flagged
 * pass with r3dgs_forward (the Python host does: diff_gaussian_rasterization/_C.py, strict mode, the default).  The
 * numbers are published ~40 us into the pass (long before it ends), so that check does not drain the GPU.
 * Everything else (arguments, outputs, numerics) is r3dgs_forward's.
 *   r3dgs_reserve_hint_view: reservation proposed from earlier passes of this camera (key: the device address of its
 *     view matrix), falling back to the largest recent pair count of this (device, W, H); scaled to P, with slack
 *     (R3DGS_RESERVE_SLACK_PCT, default 150), rounded up to a geometric grid (a reservation keys the captured graph).
 *     0 = nothing known yet about this image size (run r3dgs_forward once) or R3DGS_RESERVE=off.
 *     r3dgs_reserve_hint is the same without a camera.
 *   r3dgs_pass_query: num_rendered / visible / reserve (-1 for exact-size passes) / flags of a ticket; wait = 0
 *     returns 0 if the pass has not produced them yet, wait = 1 blocks (host-memory poll with deadline).
 *     Returns 1 when filled in, negative on error (e.g. the ticket is more than ~1000 passes old).  A ticket knows
 *     its device: the query may come from any thread, with any device current.
 *   r3dgs_reserve_overflow_events: number of truncated passes seen so far (+ the last one's numbers). */
#define R3DGS_PASS_TRUNCATED 1
#define R3DGS_PASS_DEPTH_BUCKET_OVERFLOW 2
int r3dgs_reserve_hint(int P, int width, int height);
int r3dgs_reserve_hint_view(int P, int width, int height, const float* viewmatrix);
long long r3dgs_forward_reserved(char* geom_buffer, char* binning_buffer, char* image_buffer, int reserve, int P,
                                 const int* D, int M, const float* background, int width, int height,
                                 const float* means3D, const float* shs, const float* colors_precomp,
                                 const float* opacities, const float* scales, float scale_modifier,
                                 const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                                 const float* projmatrix, const float* cam_pos, float tan_fovx, float tan_fovy,
                                 int prefiltered, float* out_color, int* out_touched_pixels, float* out_transmittance,
                                 int* radii, int calculate_mean_transmittance, int debug, void* stream);
long long r3dgs_inference_forward_reserved(char* geom_buffer, char* binning_buffer, char* image_buffer, int reserve,
                                           int P, const int* D, int bandsNum, const int* coeffsNum,
                                           const int* perBandPrimitiveCount, const int* cumSumPrimitiveCount,
                                           const float* background, int width, int height, const float* means3D,
                                           const float* shs, const float* colors_precomp, const float* opacities,
                                           const float* scales, float scale_modifier, const float* rotations,
                                           const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                           const float* cam_pos, float tan_fovx, float tan_fovy, int prefiltered,
                                           float* out_color, int* out_touched_pixels, float* out_transmittance,
                                           int* radii, int calculate_mean_transmittance, int debug, void* stream);
int r3dgs_pass_query(long long ticket, int wait, int* num_rendered, int* visible, int* reserve, int* flags);
long long r3dgs_reserve_overflow_events(int* last_num_rendered, int* last_reserve);
/* (tile, Gaussian) pairs of a pass, as opposed to its num_rendered.  num_rendered keeps the reference's definition (sum
 * over the visible Gaussians of the tile count of the 3-sigma bounding square, rasterizer_impl.cu:441-446); the lists
 * this library builds leave out the tiles of that square which the Gaussian cannot reach with alpha >= 1/255
 * (opacity-aware rects, on by default: no pixel decision changes, so image and gradients are those of the full lists),
 * so pairs <= num_rendered.  The binning blob and R3DGS_PASS_TRUNCATED are about pairs.
 *   r3dgs_pass_pairs: pair count of a ticket (wait as for r3dgs_pass_query); negative on error.
 *   r3dgs_forward_pairs: pair count of the last r3dgs_forward / r3dgs_inference_forward call of this thread
 *     (informational: the binning blob of such a call is carved for the num_rendered it returned).
 *   r3dgs_set_tight_rects(0): bin into the reference's rects (lists identical to the reference's; pairs ==
 *     num_rendered); returns the previous setting (a negative argument only queries).  Also R3DGS_TIGHT_RECT=0.
 *   r3dgs_export_rects: debug accessor, the tile rect (x0, y0, x1, y1; exclusive maxima) each Gaussian was binned
 *     into, [P][4] uint16 device array (undefined for culled Gaussians). */
int r3dgs_pass_pairs(long long ticket, int wait);
int r3dgs_forward_pairs(void);
int r3dgs_set_tight_rects(int on);
int r3dgs_export_rects(int P, char* geom_buffer, unsigned short* rects, void* stream);

/* Launch order of the backward blend's tiles.  1 (default): heaviest first -- the tiles whose lists the backward has to walk
 * deepest (the forward leaves the depth of every 8x8 quadrant's last contributor in the image blob) are started first, so
 * that the chip does not drain while the few long walks finish; 0: row-major, as the forward.  Every tile's arithmetic is
 * its own: the gradients are bit-identical either way.  Returns the previous setting (a negative argument only queries).
 * Also R3DGS_TILE_ORDER=0.  (No counterpart in the reference, whose backward.cu:405-618 takes tiles in grid order.) */
int r3dgs_set_tile_order(int on);
/* SH direction derivatives.  A forward over a dense SH tensor leaves d(colour)/d(view direction) of every visible Gaussian
 * in the geometry blob (36 B each, evaluated while the SH row is staged for the colour anyway); a backward without a
 * sparsity term (lambda_sh_sparsity == 0) then does not read the SH tensor at all (backward.cu:20-172 reads 12 * M bytes
 * per Gaussian for exactly these nine numbers).  0: the backward reads the rows, as with a sparsity term.  Bit-identical
 * gradients either way.  Returns the previous setting (a negative argument only queries).  Also R3DGS_SH_CACHE=0. */
int r3dgs_set_sh_cache(int on);
/* The per-Gaussian backward evaluates the covariance chain (conic -> cov2D -> cov3D -> scale / quaternion,
 * backward.cu:228-306 and 311-374) in double from the same fp32 inputs and rounds once: the reference's fp32 evaluation of
 * that chain is 2e-4 of max |dL_drotations| away from the exact value at the benchmark shape.
 * r3dgs_set_f64_chain(0) selects the fp32 restatement of the reference's arithmetic; returns the previous setting (a
 * negative argument only queries).  Also R3DGS_F64_CHAIN=0. */
int r3dgs_set_f64_chain(int on);
/* r3dgs_forward_hint(0): the forwards this thread issues from now on will not be followed by a backward (rendering under
 * no_grad): they leave no SH direction derivatives (their header says so; a backward on such a state reads the SH rows).
 * r3dgs_forward_hint(1), the initial state: they do. */
void r3dgs_forward_hint(int will_backward);

/* r3dgs_set_bwd_segments(0): the backward blend walks every tile's list with ONE workgroup; 1 (default): a list of at
 * least thr = max(2 S, 75 % of the pass's mean list length) entries whose deepest contributor lies behind entry S is walked
 * in ceil(deepest / S) segments (at most 32) of S = 128 entries by several workgroups, each starting from the per-pixel
 * state the forward blend checkpointed there (real scenes have tiles many times heavier than the mean; DESIGN.md section
 * 6).  A pass whose segments would exceed what it launches (tiles + min(pairs / 128, 8 x tiles) units) walks 2 S, 4 S or
 * 8 S instead.  R3DGS_BWD_SEG_LEN=128|256 sets S, R3DGS_BWD_SEG_FACTOR=<percent> the factor (75); thr is written into the
 * pass header by the forward (depth_sort.h, where the header is written; common.h bwd_segment_factor_pct).  The forward
 * and the backward of a state must run under the same setting only in the sense that a forward issued with segments off
 * leaves no checkpoints and its backward then never splits (the pass header says which).  Returns the previous setting (a
 * negative argument only queries).  Also R3DGS_BWD_SEG=0. */
int r3dgs_set_bwd_segments(int on);

/* Debug accessor: the forward's per-quadrant depths ([tiles][4] uint32: quadrant q = (x half) + 2 * (y half) of the
 * 16x16 tile) and, after a backward with the order on, the launch order it used: [cap + 16] uint32, cap =
 * r3dgs_bwd_units_cap(R, W, H) -- eight lists of cap / 8 slots (list g: the units of the 4 x 4 tile blocks g, g + 8, ... of the image, blocks counted
 * row-major, heaviest first;
 * entry = tile | segment << 20 | segments of the tile << 26), then per list the number of its units and log2 of the segment
 * length they walk (0: whole tiles).  Workgroup b of the backward blend takes entry b / 8 of list b % 8.  P, R: as given to
 * the backward.  Device arrays, either may be NULL. */
int r3dgs_export_tile_order(int P, int R, int width, int height, char* binning_buffer, char* image_buffer,
                            unsigned int* quad_depth, unsigned int* unit_order, void* stream);
int r3dgs_bwd_units_cap(int R, int width, int height);

/* Forget every pair count learnt so far (a new scene is about to be loaded; tests): the next pass of each image size
 * takes the exact-size path again. */
void r3dgs_reserve_forget(void);
/* Per-camera pair counts are remembered under the device address of the camera's view matrix (r3dgs_reserve_hint_view).
 * A caller that frees a view matrix should say so, or the allocator may hand the address to another camera whose first
 * pass then inherits a stranger's count (harmless in strict mode -- the pass is redone exactly -- but it costs that redo;
 * the hint is also never taken below 0.3 x the largest recent count of the image size). */
void r3dgs_reserve_forget_view(const float* viewmatrix);

/* Rasterizer::backward (rasterizer.h:58-87, rasterizer_impl.cu:508-630).  Returns 0 or a negative status.
 * No host synchronisation; one hipGraph launch once the shape has been seen.  R is the pair capacity the forward
 * sized the binning blob with: the num_rendered r3dgs_forward returned (the reference's contract), or the `reserve`
 * passed to r3dgs_forward_reserved (r3dgs_binning_capacity recovers a capacity with the same layout from the blob's
 * size); the pair count itself is read from the device.  Takes lambda_sh_sparsity (the reference's public wrapper argument,
 * rasterize_points.cu:245; the multiplier lambda / (visible * 45) is formed on the device).
 * Every element of every output is written (zeros where the reference relies on zero-initialised
 * tensors), so outputs may be uninitialised.  dL_dconic ([P,2,2]) may be NULL.
 * The three state blobs are NOT read-only here: the per-pair gradient slab, its flags, the per-Gaussian sums and the tile
 * launch order are scratch areas inside them (the reference's backward allocates nothing either).  Several backward passes
 * over one forward state (retain_graph) are fine one after the other -- each leaves the scratch as it found it -- but
 * must not run concurrently on different streams.
 * Shapes: dL_dpix [3,H,W]; dL_dmean2D [P,3]; dL_dopacity [P]; dL_dcolor [P,3]; dL_dmean3D [P,3];
 * dL_dcov3D [P,6]; dL_dsh [P,M,3]; dL_dscale [P,3]; dL_drot [P,4]. */
int r3dgs_backward(int P, const int* D, int M, int R, const float* background, int width, int height,
                   const float* means3D, const float* shs, const float* colors_precomp, const float* scales,
                   float scale_modifier, const float* rotations, const float* cov3D_precomp,
                   const float* viewmatrix, const float* projmatrix, const float* campos, float tan_fovx,
                   float tan_fovy, const int* radii, char* geom_buffer, char* binning_buffer, char* image_buffer,
                   const float* dL_dpix, float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                   float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                   float lambda_sh_sparsity, int debug, void* stream);

/* Debug accessor for bit-exact checks of the integer stages (SURVEY.md 8b "provide a debug accessor"):
 * copies, out of the opaque blobs of a finished forward, the sorted list in the REFERENCE's format --
 * keys[i] = (tile << 32) | depth_bits (rasterizer_impl.cu:110-113), point_list, per-tile ranges,
 * n_contrib and final T.  R = pair capacity of the binning blob (as for r3dgs_backward), count = entries of the
 * list to export (<= num_rendered).  Any output pointer may be NULL.  Device pointers. */
int r3dgs_export_binning(int P, int R, int count, int width, int height, char* geom_buffer, char* binning_buffer,
                         char* image_buffer, uint64_t* keys, uint32_t* point_list, uint32_t* ranges /*[Tn][2]*/,
                         uint32_t* n_contrib, float* final_T, uint32_t* tiles_touched, void* stream);

/* Next-tier operator: one camera's contribution to `calculate_colours_variance`
 * (reduced_3dgs.cu:143-198 + reduced_3dgs/sh_culling.cu:6-90), fused into one per-Gaussian kernel.  Call after
 * r3dgs_forward(..., calculate_mean_transmittance = 1) of that camera with its radii / out_touched_pixels /
 * out_transmittance.  Updates in place: wSum[P], wSumSq[P], mean[P,3], variance[P,3],
 * colourDistancesAccum[P,max_sh_deg] (all zero before the first camera).  The caller finishes with
 * colourDistancesAccum / wSum, variance / wSum, mean (reduced_3dgs.cu:202). */
int r3dgs_colour_variance_accumulate(int P, const int* D, int M, int max_sh_deg, const float* means3D,
                                     const float* cam_pos, const float* shs, const int* radii,
                                     const int* touched_pixels, const float* transmittance, float* wSum, float* wSumSq,
                                     float* mean, float* variance, float* colourDistancesAccum, void* stream);

/* Optional per-stage timing (not in the reference; it times with torch.cuda.Event pairs from Python,
 * train.py:52-53, gaussian_renderer/__init__.py:95-98).  When enabled, forward/backward record a HIP event pair
 * around each selected stage ON THE CALLER'S STREAM.  on = 0: off; 1: every stage; otherwise bit (s + 1) selects
 * stage s.  A pass with a timed stage is issued with direct launches instead of its graph (the events sit between
 * its kernels), so time only what you need: bench.py times the dominant stage alone inside its timed region.
 * r3dgs_profile_read() waits for the recorded events, writes per-stage total milliseconds and launch counts
 * (arrays of r3dgs_profile_stage_count() entries, host memory) and resets the counters.
 * Stages: preprocess_fwd, depth_sort_scan, tile_binning, blend_fwd, blend_bwd (unit order + blend kernel + pair
 * reduction), preprocess_bwd, sh_color (generic-sort route only) and blend_bwd_kernel -- the backward blend kernel ALONE,
 * nested inside blend_bwd's events (round 6: the duration a per-kernel roofline divides by). */
int r3dgs_profile_enable(int on);
int r3dgs_profile_stage_count(void);
const char* r3dgs_profile_stage_name(int stage);
int r3dgs_profile_read(double* total_ms, int* launches);

#ifdef __cplusplus
}
#endif
#endif /* R3DGS_RASTERIZER_H */
