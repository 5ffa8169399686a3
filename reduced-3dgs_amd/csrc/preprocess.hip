// preprocess.hip -- per-view, per-Gaussian forward stage on gfx950 (wave64).
//
// Replaces cuda_rasterizer/forward.cu:353-456 preprocessCUDA (dense SH [P,M,3] + per-Gaussian degrees),
// forward.cu:245-350 variableSHPreprocessCUDA (ragged degree-sorted SH buffer) and
// rasterizer_impl.cu:62-74 checkFrustum of /root/reference/submodules/diff-gaussian-rasterization.
//
// One lane per Gaussian, 256-thread workgroups, two roles: geometry (everything the binning needs; its sort keys go
// to the depth sort at once -- this kernel also installs the pass block, its by-value argument) and colour (the
// HBM-heavy SH stream, which rides in spare workgroups of the three depth-sort launches so that it runs underneath
// that sort inside one linear launch chain).  In the colour role each wave first copies the SH rows of
// its 64 Gaussians -- one contiguous span of the [P,M,3] tensor (or of the ragged buffer) -- into LDS with
// fully coalesced loads, then every lane evaluates its own row out of LDS (bank-skewed index), instead of 64
// lanes striding 192 B apart through global memory.  Output is one 48-byte record per visible Gaussian (GRec),
// the tile rect, the depth sort key and tiles_touched.
// Built with -ffp-contract=off: radii / rects / tiles_touched are bit-exact against the oracle.
#include <cstdlib>
#include <map>
#include <mutex>

#include "depth_sort.h"

namespace r3 {

constexpr int kPreBlock = kPreBlockSize;
constexpr int kMaxCoeff = 16;
constexpr int kRowFloats = 3 * kMaxCoeff;                        // 48
constexpr int kWaveShFloats = 64 * kRowFloats + (64 * kRowFloats) / 32;  // + bank skew

// Where float e of a wave's SH span sits in LDS (the lanes read the same element of 64 different rows: the rows must start
// in different banks).  ROWS48 -- a dense [P,16,3] tensor, rows of 48 floats: one word of padding per row, element e of
// the lane's row is base[49 * lane + e] and e folds into the instruction's offset field; otherwise (ragged rows, other M)
// one word of padding per 32 and the address is computed per access.
template <bool ROWS48>
__device__ __forceinline__ int skew(int e)
{
    if (ROWS48) return e + (int)(((uint32_t)e * 43691u) >> 21);   // e + e / 48 for e < 2^16
    return e + (e >> 5);
}

template <bool ROWS48>
struct ShRowLds {
    const float* base;   // ROWS48: the lane's row (span + 49 * lane); else the wave's span
    int roff;            // ROWS48: 0; else first float of the lane's row in the span
    __device__ __forceinline__ float at(int e) const { return ROWS48 ? base[e] : base[skew<false>(roff + e)]; }
};

// the wave's span (n4 float4s at src4) -> LDS, twelve loads per lane in flight
template <bool ROWS48>
__device__ __forceinline__ void stage_span(const float4* __restrict__ src4, int n4, float* dst, int lane)
{
    constexpr int kBatch = 12;
    for (int base = 0; base < n4; base += 64 * kBatch) {
        float4 v[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const int e4 = base + k * 64 + lane;
            v[k] = e4 < n4 ? src4[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < kBatch; k++) {
            const int e4 = base + k * 64 + lane;
            if (e4 < n4) {
                const int e = e4 << 2;
                if (ROWS48) {   // 48 % 4 == 0: the four floats are in one row
                    float* d = dst + skew<true>(e);
                    d[0] = v[k].x;
                    d[1] = v[k].y;
                    d[2] = v[k].z;
                    d[3] = v[k].w;
                } else {
                    dst[skew<false>(e)] = v[k].x;
                    dst[skew<false>(e + 1)] = v[k].y;
                    dst[skew<false>(e + 2)] = v[k].z;
                    dst[skew<false>(e + 3)] = v[k].w;
                }
            }
        }
    }
}

// forward.cu:19-36 getSHOffset (float3 units)
__device__ __forceinline__ int ragged_offset(int idx, const int* coeffs, const int* perband, const int* cumsum, int* deg)
{
    int off = 0;
    *deg = 0;
    if (idx < cumsum[0]) return idx * coeffs[0];
    *deg = 1;
    off += perband[0] * coeffs[0];
    if (idx < cumsum[1]) return off + (idx - cumsum[0]) * coeffs[1];
    *deg = 2;
    off += perband[1] * coeffs[1];
    if (idx < cumsum[2]) return off + (idx - cumsum[1]) * coeffs[2];
    *deg = 3;
    off += perband[2] * coeffs[2];
    return off + (idx - cumsum[2]) * coeffs[3];
}

template <bool RAGGED, int BLOCK>
__device__ __forceinline__ void color_role(const PreArgs& a, char* smem, int first, int last, int wg, int n_wg);

// ---- kernel 1: geometry ---------------------------------------------------------------------------
// cull, projection, conic, radius, tile rect, depth key, per-view counters.  Reads 44 B per Gaussian.  Its
// outputs are everything the depth sort / binning needs, so the SH -> RGB kernel below can run on a side
// stream underneath the (launch-latency-bound) sort.
// COLOR != 0 (large scenes, FwdPlan::color_in_geom): the workgroup goes on to colour its own 256 Gaussians (color_role
// below; COLOR 2 = ragged SH) while their records are still in the L2 -- the separate colour stream otherwise re-opens every
// 48-byte record for a 16-byte partial write, which HBM pays as a read-modify-write of the whole line.
template <int COLOR>
__global__ __launch_bounds__(kPreBlock) void preprocess_geom_kernel(FwdPassArgs* dst, FwdPassArgs v)
{
    extern __shared__ __attribute__((aligned(16))) char geom_smem[];
    // first kernel of the forward: it gets the pass block by value, installs it for the kernels behind it ...
    if (blockIdx.x == 0) install_block_from_kernarg(dst, (int)threadIdx.x, kPreBlock);
    const PreArgs a = v.pre;   // ... and reads its own arguments from the kernarg segment (scalar loads)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.in.P;
    const int i = blockIdx.x * kPreBlock + tid;
    const bool valid = i < P;
    const Camera cam = load_camera(a.view);

    PreOut o;
    o.radius = 0;
    o.tiles = 0;
    o.tiles_ref = 0;
    o.depth = 0.f;
    if (valid) {
        const float mx = a.in.means3D[3 * i], my = a.in.means3D[3 * i + 1], mz = a.in.means3D[3 * i + 2];
        float sc[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, c6[6];
        const float* c6p = nullptr;
        if (a.in.cov3D_precomp) {
            for (int k = 0; k < 6; k++) c6[k] = a.in.cov3D_precomp[6 * i + k];
            c6p = c6;
        } else {
            for (int k = 0; k < 3; k++) sc[k] = a.in.scales[3 * i + k];
            const float4 qv = reinterpret_cast<const float4*>(a.in.rotations)[i];
            q[0] = qv.x;
            q[1] = qv.y;
            q[2] = qv.z;
            q[3] = qv.w;
        }
        preprocess_one(cam, mx, my, mz, sc, q, c6p, a.in.opacities[i], &o, a.tight != 0);
        uint32_t dkey = 0xFFFFFFFFu;
        if (o.radius > 0) {
            GRec r;
            r.x = o.px;
            r.y = o.py;
            r.cA = o.conic[0];
            r.cB = o.conic[1];
            r.cC = o.conic[2];
            r.op = o.opacity;
            r.r = r.g = r.b = 0.f;  // filled in by the colour kernel
            r.rect_min = (uint32_t)o.rmin[0] | ((uint32_t)o.rmin[1] << 16);
            r.width_clamp = (uint32_t)(o.rmax[0] - o.rmin[0]);  // clamp bits OR-ed in by the colour kernel
            r.pair_start = 0xFFFFFFFFu;   // filled in by the pair-emission kernel (stays ~0 if no pair was emitted)
            a.rec[i] = r;
            a.rect[i] = make_ushort4((unsigned short)o.rmin[0], (unsigned short)o.rmin[1], (unsigned short)o.rmax[0],
                                     (unsigned short)o.rmax[1]);
            // a visible Gaussian whose opacity-aware rect is empty owns no pair: it stays out of the depth order like a
            // culled one (its radius, record and gradients are those of a visible Gaussian)
            if (o.tiles > 0) dkey = __float_as_uint(o.depth);
        }
        a.radii[i] = o.radius;
        a.tiles[i] = o.tiles;
        a.depth_key[i] = dkey;
    }
    // per-workgroup totals, stored (not accumulated): visible count (SH-sparsity normaliser of the backward),
    // num_rendered, and the depth range of the visible Gaussians (bucketed depth sort, binning.hip).  R does not depend
    // on the depth order, so the host can fetch it while the GPU is busy with the depth sort (capi.hip).
    const unsigned long long vmask = __ballot(o.radius > 0);
    const bool binned = o.tiles > 0;
    const unsigned long long bmask = __ballot(binned);
    uint32_t tsum = o.tiles, rsum = o.tiles_ref;
    uint32_t dmax = binned ? __float_as_uint(o.depth) : 0u, dinv = binned ? ~__float_as_uint(o.depth) : 0u;
    for (int off = 32; off > 0; off >>= 1) {
        tsum += (uint32_t)__shfl_xor((int)tsum, off);
        rsum += (uint32_t)__shfl_xor((int)rsum, off);
        dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, off));
        dinv = max(dinv, (uint32_t)__shfl_xor((int)dinv, off));
    }
    __shared__ uint32_t s_cnt[kPreBlock / 64][6];
    if (lane == 0) {
        s_cnt[wave][0] = (uint32_t)__popcll(vmask);
        s_cnt[wave][1] = tsum;
        s_cnt[wave][2] = dmax;
        s_cnt[wave][3] = dinv;
        s_cnt[wave][4] = (uint32_t)__popcll(bmask);
        s_cnt[wave][5] = rsum;
    }
    __syncthreads();
    if (tid == 0) {
        PrePartial pp = {0u, 0u, 0u, 0u, 0u, 0u, {0u, 0u}};
        for (int k = 0; k < kPreBlock / 64; k++) {
            pp.visible += s_cnt[k][0];
            pp.num_rendered += s_cnt[k][1];
            pp.depth_max = max(pp.depth_max, s_cnt[k][2]);
            pp.depth_inv_min = max(pp.depth_inv_min, s_cnt[k][3]);
            pp.binned += s_cnt[k][4];
            pp.rendered_ref += s_cnt[k][5];
        }
        a.partials[blockIdx.x] = pp;
    }
    if (COLOR) {
        __syncthreads();
        if (COLOR == 2)
            color_role<true, kPreBlock>(a, geom_smem, (int)blockIdx.x, (int)blockIdx.x + 1, 0, 1);
        else
            color_role<false, kPreBlock>(a, geom_smem, (int)blockIdx.x, (int)blockIdx.x + 1, 0, 1);
    }
}

// ---- kernel 2: colour -------------------------------------------------------------------------------
// SH -> RGB (forward.cu:105-159, ragged variant :19-100) or copy of the precomputed colours into the
// records of the visible Gaussians.  Streams the SH tensor (192 B per Gaussian at degree 3): the wave's 64 rows
// are one contiguous span, staged through LDS with dwordx4 loads, evaluated per lane from LDS.
// Launched as a SMALL persistent grid (launch_preprocess_color): the kernel is bandwidth-bound filler next to the
// latency-bound depth-sort kernels of the main stream, and a full grid's LDS footprint (3 x 49 KB per CU) left their
// workgroups no room -- the depth scatter kernel took 46 us instead of 14 us beside it.
constexpr size_t kColorLds = sizeof(float) * (kPreBlock / 64) * kWaveShFloats;

// chunks [first + wg, last) in steps of n_wg, BLOCK Gaussians each (BLOCK = workgroup size); smem: BLOCK / 64 wave windows.
// BLOCK = 64 (standalone kernel only): single-wave workgroups -- 12 per CU instead of 3 of four waves: no barrier couples
// the waves' load / evaluate phases, so they drift apart and one wave's loads fly while another evaluates.
template <bool RAGGED, int BLOCK>
__device__ __forceinline__ void color_role(const PreArgs& a, char* smem, int first, int last, int wg, int n_wg)
{
    float(*s_sh)[kWaveShFloats] = reinterpret_cast<float(*)[kWaveShFloats]>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.in.P, M = a.in.M;
    const bool rows48 = !RAGGED && M == 16;   // dense degree-3 tensor (wave-uniform)
  for (int blk = first + wg; blk < last; blk += n_wg) {
    const int i = blk * BLOCK + tid;
    const bool valid = i < P;
    const int wave_first = blk * BLOCK + wave * 64;
    const bool vis = valid && a.tiles[i] > 0;
    const bool need_sh = vis && (a.in.colors_precomp == nullptr);

    // ---- per-lane SH row placement inside the wave's contiguous span --------------------------
    int deg = 0, roff = 0;  // roff in floats relative to the span start
    long span_first = 0;    // first float of the wave's span in the SH buffer
    int span_len = 0;       // floats
    if (a.in.colors_precomp == nullptr) {
        if (RAGGED) {
            const int last_i = min(wave_first + 63, P - 1);
            int d0, dl;
            const int off_first = wave_first < P ? ragged_offset(wave_first, a.in.coeffs_num, a.in.per_band_count, a.in.cumsum_count, &d0) : 0;
            const int off_last = wave_first < P ? ragged_offset(last_i, a.in.coeffs_num, a.in.per_band_count, a.in.cumsum_count, &dl) : 0;
            span_first = 3L * off_first;
            span_len = wave_first < P ? 3 * (off_last + (dl + 1) * (dl + 1) - off_first) : 0;
            if (valid) {
                const int off = ragged_offset(i, a.in.coeffs_num, a.in.per_band_count, a.in.cumsum_count, &deg);
                roff = 3 * (off - off_first);
            }
        } else {
            const int nrows = max(0, min(64, P - wave_first));
            span_first = 3L * M * wave_first;
            span_len = nrows * 3 * M;
            roff = lane * 3 * M;
            if (valid) deg = a.in.degrees[i];
        }
    }
    // wave-uniform: does any lane of this wave need its SH row?
    const bool wave_needs = __ballot(need_sh) != 0ull;
    if (wave_needs) {
        const float* src = a.in.shs + span_first;
        float* dst = s_sh[wave];
        if (((span_first | span_len) & 3) == 0) {  // 16-B aligned span (always true for M = 16): dwordx4 loads
            // Twelve loads per lane cover a full degree-3 span (64 rows x 192 B).  They are issued back to back before the
            // first LDS store: with one load in flight per wave (load, wait, store, next load) the kernel ran at the
            // 2 TB/s that 12 waves/CU x 1 KB per memory latency allow.
            const float4* src4 = reinterpret_cast<const float4*>(src);
            if (rows48)
                stage_span<true>(src4, span_len >> 2, dst, lane);
            else
                stage_span<false>(src4, span_len >> 2, dst, lane);
        } else {
            for (int e = lane; e < span_len; e += 64) dst[rows48 ? skew<true>(e) : skew<false>(e)] = src[e];
        }
    }
    __syncthreads();

    if (vis) {
        float rgb[3];
        uint32_t cbits = 0;
        if (need_sh) {
            const float campos[3] = {a.view.campos[0], a.view.campos[1], a.view.campos[2]};
            const float mx = a.in.means3D[3 * i], my = a.in.means3D[3 * i + 1], mz = a.in.means3D[3 * i + 2];
            // the colour, and -- while the row is here -- its derivatives with respect to the view direction, which is all
            // the backward wants from the row (GeomState::sh_ddir)
            auto eval = [&](const auto& row) {
                sh_to_rgb(deg, row, mx, my, mz, campos, rgb, &cbits);
                if (!RAGGED && a.sh_ddir && deg > 0) {
                    float d9[9];
                    sh_dir_derivs_at(deg, row, mx, my, mz, campos, d9);
                    float* o = a.sh_ddir + 9 * (size_t)i;
#pragma unroll
                    for (int k = 0; k < 9; k++) o[k] = d9[k];
                }
            };
            if (rows48)
                eval(ShRowLds<true>{s_sh[wave] + 49 * lane, 0});
            else
                eval(ShRowLds<false>{s_sh[wave], roff});
        } else {
            rgb[0] = a.in.colors_precomp[3 * i];
            rgb[1] = a.in.colors_precomp[3 * i + 1];
            rgb[2] = a.in.colors_precomp[3 * i + 2];
        }
        GRec* r = a.rec + i;
        r->r = rgb[0];
        r->g = rgb[1];
        r->b = rgb[2];
        if (cbits) r->width_clamp |= cbits << 16;  // same lane wrote the width in the geometry kernel
    }
    __syncthreads();   // the staging buffer is reused by the next round
  }
}

// standalone colour kernel (generic depth sort path, R3DGS_COLOR_FUSE=0): small persistent grid
template <bool RAGGED, int BLOCK>
__global__ __launch_bounds__(BLOCK) void preprocess_color_kernel(const PreArgs* __restrict__ ap)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PreArgs a = *ap;
    color_role<RAGGED, BLOCK>(a, smem, 0, (a.in.P + BLOCK - 1) / BLOCK, (int)blockIdx.x, (int)gridDim.x);
}

// ---- depth-sort kernels carrying a share of the colour stream in extra workgroups -------------------------------
// Workgroups [0, n_sort) run the depth-sort role, workgroups [n_sort, n_sort + n_color) the colour chunks
// [c0, c1).  One dynamic LDS window serves whichever role a workgroup has.
template <int STEP, bool RAGGED>
__global__ __launch_bounds__(kPreBlock) void depth_sort_color_kernel(const FwdPassArgs* __restrict__ pa, int n_sort,
                                                                     int c0, int c1)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wg = (int)blockIdx.x;
    if (wg < n_sort) {
        const DepthArgs d = pa->depth;
        if (STEP == 0)
            depth_hist_role(d, pa->header, smem, wg);
        else if (STEP == 1)
            depth_scatter_role(d, smem, wg);
        else
            depth_bucket_group_role(d, smem, wg, n_sort);
    } else {
        const PreArgs a = pa->pre;
        color_role<RAGGED, kPreBlock>(a, smem, c0, c1, wg - n_sort, (int)gridDim.x - n_sort);
    }
}

__global__ __launch_bounds__(64 * kColWaves) void depth_colscan_kernel(const DepthArgs* __restrict__ ap)
{
    const DepthArgs d = *ap;
    depth_colscan_role(d, (int)blockIdx.x);
}

void issue_preprocess_geom(const FwdPlan& p, FwdPassArgs* dst, const FwdPassArgs& v, hipStream_t s)
{
    const int blocks = (p.P + kPreBlock - 1) / kPreBlock;
    if (p.color_in_geom && p.ragged)
        hipLaunchKernelGGL(preprocess_geom_kernel<2>, dim3(blocks), dim3(kPreBlock), kColorLds, s, dst, v);
    else if (p.color_in_geom)
        hipLaunchKernelGGL(preprocess_geom_kernel<1>, dim3(blocks), dim3(kPreBlock), kColorLds, s, dst, v);
    else
        hipLaunchKernelGGL(preprocess_geom_kernel<0>, dim3(blocks), dim3(kPreBlock), 0, s, dst, v);
}

void issue_preprocess_color(const FwdPlan& p, const PreArgs* a, hipStream_t s)
{
    static const int block = env_int("R3DGS_COLOR_BLOCK", 256, 64, 256) == 64 ? 64 : 256;   // workgroup size of THIS kernel
    const int blocks = (p.P + block - 1) / block;
    const int want = p.color_grid > 0 ? p.color_grid * (kPreBlock / block) : blocks;
    const int grid = blocks > want ? want : blocks;
    const size_t lds = kColorLds / (kPreBlock / block);
    if (block == 64) {
        if (p.ragged)
            hipLaunchKernelGGL((preprocess_color_kernel<true, 64>), dim3(grid), dim3(64), lds, s, a);
        else
            hipLaunchKernelGGL((preprocess_color_kernel<false, 64>), dim3(grid), dim3(64), lds, s, a);
    } else if (p.ragged) {
        hipLaunchKernelGGL((preprocess_color_kernel<true, kPreBlock>), dim3(grid), dim3(kPreBlock), lds, s, a);
    } else {
        hipLaunchKernelGGL((preprocess_color_kernel<false, kPreBlock>), dim3(grid), dim3(kPreBlock), lds, s, a);
    }
}

template <int STEP, bool RAGGED>
static void launch_sort_color(const FwdPassArgs* pa, int n_sort, int n_color, int c0, int c1, size_t lds, hipStream_t s)
{
    hipLaunchKernelGGL((depth_sort_color_kernel<STEP, RAGGED>), dim3(n_sort + n_color), dim3(kPreBlock), lds, s, pa,
                       n_sort, c0, c1);
}

static size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

template <int STEP, bool RAGGED>
static void opt_in_lds(size_t bytes)
{
    if (bytes > 48 * 1024)
        R3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(depth_sort_color_kernel<STEP, RAGGED>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

// Not a stream operation: called before a chain is captured / issued.  hipFuncSetAttribute applies to the CURRENT
// device, so what has been prepared is remembered per device.
void prepare_depth_bucket_sort(int nb)
{
    static std::mutex mu;
    static std::map<int, int> prepared;
    std::lock_guard<std::mutex> lk(mu);
    int dev = 0;
    R3_HIP(hipGetDevice(&dev));
    int& prepared_nb = prepared[dev];
    if (nb <= prepared_nb) return;
    const size_t h = max_sz(depth_hist_lds(nb), kColorLds), sc = max_sz(depth_scatter_lds(nb), kColorLds),
                 bs = max_sz(kBucketSortLds, kColorLds);
    opt_in_lds<0, false>(h);
    opt_in_lds<0, true>(h);
    opt_in_lds<1, false>(sc);
    opt_in_lds<1, true>(sc);
    opt_in_lds<2, false>(bs);
    opt_in_lds<2, true>(bs);
    if (kColorLds > 48 * 1024) {
        R3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(preprocess_geom_kernel<1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kColorLds));
        R3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(preprocess_geom_kernel<2>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kColorLds));
        R3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(preprocess_color_kernel<false, kPreBlock>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kColorLds));
        R3_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(preprocess_color_kernel<true, kPreBlock>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kColorLds));
    }
    prepared_nb = nb;
}

// Bucketed depth sort with the SH -> RGB stream riding in spare workgroups of three of its four kernels.  The colour
// chunks are split p.color_split[0..2] percent over the histogram / scatter / bucket-sort launches (each share about
// as long as the sort role it hides behind); with fusion off (p.color_fuse == 0) the colour kernel runs on its own
// after the sort.
void issue_depth_sort_and_color(const FwdPlan& p, const FwdPassArgs* pa, hipStream_t s)
{
    const int rows = (int)depth_hist_rows((size_t)p.P), nb = p.nb;
    const int chunks = (p.P + kPreBlock - 1) / kPreBlock;
    int c[4] = {0, 0, 0, chunks};
    if (p.color_fuse) {
        c[1] = (int)((long long)chunks * p.color_split[0] / 100);
        c[2] = c[1] + (int)((long long)chunks * p.color_split[1] / 100);
    } else {
        c[1] = c[2] = c[3] = 0;
    }
    const int cw = p.color_grid > 0 ? p.color_grid : 512;
    auto n_color = [&](int k) { const int n = c[k + 1] - c[k]; return n < cw ? n : cw; };
    const size_t lds0 = n_color(0) ? max_sz(depth_hist_lds(nb), kColorLds) : depth_hist_lds(nb);
    const size_t lds1 = n_color(1) ? max_sz(depth_scatter_lds(nb), kColorLds) : depth_scatter_lds(nb);
    const size_t lds2 = n_color(2) ? max_sz(kBucketSortLds, kColorLds) : kBucketSortLds;
    if (p.ragged) {
        launch_sort_color<0, true>(pa, rows, n_color(0), c[0], c[1], lds0, s);
        hipLaunchKernelGGL(depth_colscan_kernel, dim3((nb + 1 + 63) / 64), dim3(64 * kColWaves), 0, s, &pa->depth);
        launch_sort_color<1, true>(pa, rows, n_color(1), c[1], c[2], lds1, s);
        launch_sort_color<2, true>(pa, (nb + kBucketsPerGroup - 1) / kBucketsPerGroup, n_color(2), c[2], c[3], lds2, s);
    } else {
        launch_sort_color<0, false>(pa, rows, n_color(0), c[0], c[1], lds0, s);
        hipLaunchKernelGGL(depth_colscan_kernel, dim3((nb + 1 + 63) / 64), dim3(64 * kColWaves), 0, s, &pa->depth);
        launch_sort_color<1, false>(pa, rows, n_color(1), c[1], c[2], lds1, s);
        launch_sort_color<2, false>(pa, (nb + kBucketsPerGroup - 1) / kBucketsPerGroup, n_color(2), c[2], c[3], lds2, s);
    }
    if (!p.color_fuse && !p.color_in_geom && !p.color_side) issue_preprocess_color(p, &pa->pre, s);
}

// rasterizer_impl.cu:62-74 checkFrustum: present[i] = (view * p).z > 0.2
__global__ __launch_bounds__(256) void mark_visible_kernel(int P, const float* means, const float* view, bool* present)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float pv[3];
    xform4x3(view, means[3 * i], means[3 * i + 1], means[3 * i + 2], pv);
    present[i] = pv[2] > 0.2f;
}

void launch_mark_visible(int P, const float* means3D, const float* view, bool* present, hipStream_t s)
{
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
}

}  // namespace r3
