// preprocess_bwd.hip -- per-Gaussian backward stage on gfx950, ONE fused kernel.
//
// Replaces the two reference kernels cuda_rasterizer/backward.cu:177-307 computeCov2DCUDA and
// backward.cu:379-434 preprocessCUDA (+ :20-172 SH, :311-374 cov3D), and the count(radii>0) reduction of
// rasterizer_impl.cu:549-571 (the visible count was already produced by the forward; no host sync, no
// malloc/free in the backward) of /root/reference/submodules/diff-gaussian-rasterization.
//
// The 2D-stage gradients arrive as one 9-float row per (tile, Gaussian) pair; pair_reduce_kernel below sums
// them per Gaussian first (segmented sum, no atomics).
// HBM-bound streaming kernel: one lane per Gaussian.  The wave's 64 SH rows are staged through LDS with
// coalesced loads, the dL/dsh rows are built IN PLACE in the same LDS span and written back with
// coalesced stores -- including the zeros the API contract demands for bands above a Gaussian's degree
// and for culled Gaussians -- so the caller does not have to memset the 12*M*P-byte tensor first.
// Every output element of every Gaussian is written (zeros where the reference leaves its
// zero-initialised tensors untouched), so outputs may be uninitialised memory.
#include "common.h"

namespace r3 {

// Workgroup size of preprocess_bwd_kernel.  One wave per workgroup (round 4): every wave owns its LDS window and its 64
// Gaussians anyway, and without a four-wave barrier the twelve waves of a CU drift apart, so loads, evaluation and the
// dL_dsh row stores of different waves overlap: 0.194 -> 0.181 ms at 2 M Gaussians, 0.574 -> 0.539 at 6 M, unchanged at
// 500 k (0.062-0.064 either way); 128: in between.  -DR3_PREBWD_BLOCK=256 for the round-3 shape.
#ifndef R3_PREBWD_BLOCK
#define R3_PREBWD_BLOCK 64
#endif
constexpr int kBwdBlock = R3_PREBWD_BLOCK;
constexpr int kBwdWaveShFloats = 64 * 48 + (64 * 48) / 32;

// Where float e of the wave's span (64 rows x 3M floats, row after row) sits in LDS.  The lanes of a wave read the same
// element of 64 different rows, so the rows must start in different banks.
//   ROWS48 (M == 16, every dense degree-3 tensor): one word of padding per row, i.e. rows 49 words apart.  Element e of
//     lane's row is base[49 * lane + e]: the compiler folds e into the instruction's offset field, no address arithmetic
//     per access (the general scheme below spent 30 % of the kernel's vector instructions on it).
//   otherwise: one word of padding per 32.
template <bool ROWS48>
__device__ __forceinline__ int bskew(int e)
{
    if (ROWS48) return e + (int)(((uint32_t)e * 43691u) >> 21);   // e + e / 48 for e < 2^16
    return e + (e >> 5);
}

template <bool ROWS48>
struct ShRowLdsRW {
    float* base;   // ROWS48: already advanced to the lane's row (base + 49 * lane); else the wave's span
    int roff;      // ROWS48: 0; else first float of the lane's row in the span
    __device__ __forceinline__ float at(int e) const { return ROWS48 ? base[e] : base[bskew<false>(roff + e)]; }
    __device__ __forceinline__ void put(int e, float v) const
    {
        if (ROWS48)
            base[e] = v;
        else
            base[bskew<false>(roff + e)] = v;
    }
};

// ------------------------------------------------------------------------------------------------
// Per-Gaussian sums of the per-(tile, Gaussian)-pair gradients written by the backward blend.
// The slab is in emission order, i.e. a Gaussian's pairs are contiguous, so this is a segmented sum over a
// sorted key (the Gaussian id of each pair = the unsorted value array of the tile sort).  One lane per pair,
// coalesced loads, a 6-step segmented scan inside each wave.  A run that lies inside one 64-pair group is
// final and goes to acc[gid]; a run cut by a group boundary leaves its piece in the group's
// leading / trailing slot and the per-Gaussian kernel adds the <= (tiles/64 + 2) pieces in order.
// No atomics, fixed summation order: the backward is bit-reproducible.  (Letting each Gaussian's lane loop
// over its own pairs instead cost 0.86 ms: the largest splats own 600+ pairs.)
// ------------------------------------------------------------------------------------------------
constexpr int kReduceGroups = 1;   // 64-pair groups per wave and trip (2, with all their loads in flight together, measured the
                                   // same 49 us: the kernel moves whole 128-byte lines of the slab for the ~45 % of its 48-byte
                                   // rows that are flagged, ~3.6 TB/s of DRAM traffic)

// segmented sum of one 64-pair group (one pair per lane) and the stores of its run totals
__device__ __forceinline__ void pair_reduce_group(const PairReduceArgs& a, uint32_t e, int lane, bool valid, uint32_t key,
                                                  float (&v)[kPairGrad], uint32_t key_before, uint32_t key_after)
{
    // Inclusive segmented scan over equal-key runs, on DPP (VALU) moves only: Kogge-Stone inside each row of 16 lanes
    // (row_shr 1, 2, 4, 8), then the classic row_bcast:15 / row_bcast:31 pair carries the row totals across -- valid
    // for a SEGMENTED scan because runs are contiguous: a lane shares the key of the broadcast lane iff its run reaches
    // back to it.  The first version went through ds_bpermute (60 LDS-pipe instructions per wave: 61% issue stall).
    // A lane without a source keeps `old`: ~key for the key (never equal), so it takes nothing.
#define R3_SEG_STEP(CTRL, RMASK)                                                                                          \
    {                                                                                                                     \
        const uint32_t ku = (uint32_t)__builtin_amdgcn_update_dpp((int)~key, (int)key, CTRL, RMASK, 0xf, false);          \
        const bool take = ku == key;                                                                                      \
        _Pragma("unroll") for (int k = 0; k < kPairGrad; k++)                                                             \
        {                                                                                                                 \
            const float vu = __builtin_bit_cast(                                                                          \
                float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[k]), CTRL, RMASK, 0xf, false));           \
            if (take) v[k] += vu;                                                                                         \
        }                                                                                                                 \
    }
    R3_SEG_STEP(0x111, 0xf)   // row_shr:1
    R3_SEG_STEP(0x112, 0xf)   // row_shr:2
    R3_SEG_STEP(0x114, 0xf)   // row_shr:4
    R3_SEG_STEP(0x118, 0xf)   // row_shr:8
    R3_SEG_STEP(0x142, 0xa)   // row_bcast:15 -> rows 1 and 3
    R3_SEG_STEP(0x143, 0xc)   // row_bcast:31 -> rows 2 and 3
#undef R3_SEG_STEP
    const uint32_t knext = (uint32_t)__shfl_down((int)key, 1);
    const uint32_t key0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);   // the group's first run (all lanes active here)
    if (valid && (lane == 63 || knext != key)) {  // last lane of a run: holds the run's sum inside this group
        const uint32_t gid = a.order ? a.order[key] : key;
        // does the run reach across the borders of this group?  The neighbouring pairs' keys say so (runs are contiguous:
        // same key <=> same run); asking the Gaussian's record for its pair range was a dependent gather per run
        const bool from_before = key == key0 && key_before == key;
        const bool into_next = lane == 63 && key_after == key;
#ifdef R3_EXP_PR_NOSTORE   // timing experiment only (wrong results): what the scattered 48-byte run-sum stores cost --
        // 17 of pair_reduce's 41 us at 500 k, 52 of 116 at 2 M, 152 of 282 at 6 M (profiles/r06_exp_pair_reduce_stores.txt)
        if (v[0] != 12345.678f) return;
#endif
        if (!from_before && !into_next) {
#ifdef R3_ACC_IN_SLAB   // experiment: the run sum stays where the run ends (a write next to the rows this wave just read) and the
                        // per-Gaussian kernel gathers it by pair_start instead of reading a row this store scattered
            float4* dst = reinterpret_cast<float4*>(const_cast<float*>(a.pair_grad) + (size_t)e * kPairStride);
#else
            float4* dst = reinterpret_cast<float4*>(a.acc + (size_t)gid * kAccStride);   // 48-B row: three 16-B stores
#endif
#ifndef R3_NO_ZERO_ROW_SKIP
            // A run without a contributing pair (all nine sums exactly zero: a third of the visible Gaussians of the metric scene,
            // more in a densified one) stores nothing: the reader takes a row that does not carry THIS pass's stamp for zeros.
            // 40-54 % of this kernel is these scattered stores (profiles/r06_exp_pair_reduce_stores.txt); pair_reduce 41 -> 39 us
            // at 500 k, 127 -> 92 at 2 M, 282 -> 187 at 6 M.  -DR3_NO_ZERO_ROW_SKIP: every run stores its row (A/B builds).
            bool nz = false;
#pragma unroll
            for (int k = 0; k < kPairGrad; k++) nz |= v[k] != 0.f;
            if (nz) {
                dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                dst[1] = make_float4(v[4], v[5], v[6], v[7]);
                dst[2] = make_float4(v[8], __uint_as_float(a.stamp0), __uint_as_float(a.stamp1), 0.f);
            }
#else
            dst[0] = make_float4(v[0], v[1], v[2], v[3]);
            dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            dst[2] = make_float4(v[8], 0.f, 0.f, 0.f);
            if (kAccStride >= 16) dst[3] = make_float4(0.f, 0.f, 0.f, 0.f);   // 64-byte rows: the whole burst is written
#endif
        } else {
            float* wp = a.wave_part + (size_t)(e >> 6) * 2 * kPieceStride;
#ifdef R3_WP_VEC
            if (from_before) {  // continues a run of the previous group: this group's leading piece
                float4* w4 = reinterpret_cast<float4*>(wp);
                w4[0] = make_float4(v[0], v[1], v[2], v[3]);
                w4[1] = make_float4(v[4], v[5], v[6], v[7]);
                w4[2] = make_float4(v[8], 0.f, 0.f, 0.f);
            }
            if (into_next) {  // continues into the next group: trailing piece
                float4* w4 = reinterpret_cast<float4*>(wp + kPieceStride);
                w4[0] = make_float4(v[0], v[1], v[2], v[3]);
                w4[1] = make_float4(v[4], v[5], v[6], v[7]);
                w4[2] = make_float4(v[8], 0.f, 0.f, 0.f);
            }
#else
            if (from_before) {  // continues a run of the previous group: this group's leading piece
#pragma unroll
                for (int k = 0; k < kPairGrad; k++) wp[k] = v[k];
            }
            if (into_next) {  // continues into the next group: trailing piece
#pragma unroll
                for (int k = 0; k < kPairGrad; k++) wp[kPieceStride + k] = v[k];
            }
#endif
        }
    }
}

// R3_PR_WAVES (experiment): waves per SIMD the register allocation is held to (80 VGPRs = 6 by default)
#ifdef R3_PR_WAVES
__attribute__((amdgpu_waves_per_eu(R3_PR_WAVES, R3_PR_WAVES)))
#endif
__global__ __launch_bounds__(256) void pair_reduce_kernel(const PairReduceArgs* __restrict__ ap)
{
    const PairReduceArgs a = *ap;
    const uint32_t R = a.hdr->num_pairs;
    constexpr uint32_t kPerBlock = 256u * kReduceGroups;
  for (uint32_t blk = blockIdx.x; blk * kPerBlock < R; blk += gridDim.x) {   // logical blocks strided over the grid (common.h)
    const float* __restrict__ pair_grad = a.pair_grad;
    unsigned char* __restrict__ pair_flag = a.pair_flag;
    const uint32_t* __restrict__ pair_gid = a.pair_rank;
    const uint32_t rank_mask = a.rank_mask;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t e0 = blk * kPerBlock + (uint32_t)wave * (64u * kReduceGroups) + (uint32_t)lane;
    float v[kReduceGroups][kPairGrad];
    uint32_t key[kReduceGroups], kb[kReduceGroups], ka[kReduceGroups];
    bool flag[kReduceGroups];
    // every load of the wave's groups is issued before the first use: keys, flags, neighbours' keys, then the rows
#pragma unroll
    for (int g = 0; g < kReduceGroups; g++) {
        const uint32_t e = e0 + 64u * g, gbase = e & ~63u;
        key[g] = e < R ? pair_gid[e] & rank_mask : 0xFFFFFFFFu;   // run key: the Gaussian id
        flag[g] = e < R && pair_flag[e] != 0;
        kb[g] = gbase > 0u && gbase < R ? pair_gid[gbase - 1u] & rank_mask : 0xFFFFFFFFu;
        ka[g] = gbase + 64u < R ? pair_gid[gbase + 64u] & rank_mask : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int g = 0; g < kReduceGroups; g++) {
        const uint32_t e = e0 + 64u * g;
#pragma unroll
        for (int k = 0; k < kPairGrad; k++) v[g][k] = 0.f;
#ifdef R3_EXP_PR_NOLOAD    // timing experiment only (wrong results): what fetching the flagged slab rows costs
        if (flag[g] && e == 0xFFFFFFF0u) {
#else
        if (flag[g]) {  // ~1/3 of the pairs contribute; the rest of the slab is stale memory, never read
#endif
            pair_flag[e] = 0;  // consumed: all flags are zero again when this kernel ends (next backward pass)
            const float4* src = reinterpret_cast<const float4*>(pair_grad + (size_t)e * kPairStride);
            const float4 r0 = src[0], r1 = src[1];
            v[g][0] = r0.x; v[g][1] = r0.y; v[g][2] = r0.z; v[g][3] = r0.w;
            v[g][4] = r1.x; v[g][5] = r1.y; v[g][6] = r1.z; v[g][7] = r1.w;
            v[g][8] = src[2].x;
        }
    }
#pragma unroll
    for (int g = 0; g < kReduceGroups; g++) {
        const uint32_t e = e0 + 64u * g;
        pair_reduce_group(a, e, lane, e < R, key[g], v[g], kb[g], ka[g]);
    }
  }
}

void issue_pair_reduce(const BwdPlan& p, const PairReduceArgs* a, hipStream_t s)
{
    if (!p.has_pairs) return;
    constexpr uint32_t per = 256u * kReduceGroups;
    hipLaunchKernelGGL(pair_reduce_kernel, dim3((p.grid_pairs + per - 1u) / per), dim3(256), 0, s, a);
}

// F64: the covariance chain in double (gauss_math.h cov2d_backward_f64 / cov3d_backward_f64; the default)
template <bool ROWS48, bool F64>
__global__ __launch_bounds__(kBwdBlock) void preprocess_bwd_kernel(const PreBwdArgs* __restrict__ ap)
{
    __shared__ float s_sh[kBwdBlock / 64][kBwdWaveShFloats];
    const PreBwdArgs a = *ap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = a.in.P, M = a.in.M;
    const int i = blockIdx.x * kBwdBlock + tid;
    const bool valid = i < P;
    const int wave_first = blockIdx.x * kBwdBlock + wave * 64;
    const Camera cam = load_camera(a.view);
    const bool has_sh = a.in.shs != nullptr;

    const bool vis = valid && a.radii[i] > 0;
    const int nrows = max(0, min(64, P - wave_first));
    const int span_len = has_sh ? nrows * 3 * M : 0;
    const long span_first = 3L * M * wave_first;
    float* lds = s_sh[wave];
    const bool wave_vis = __ballot(vis) != 0ull;
    // the forward left the SH direction derivatives and there is no sparsity term: the SH rows are not read at all
    const bool cached = has_sh && a.sh_ddir != nullptr && a.header->sh_cache != 0u;
    // The workgroups that start together (three per CU) would load their SH rows together, compute together and store
    // together: HBM idle while they compute, the SIMDs idle while they wait.  The second and third of a CU start a step
    // later each.  (Only when the rows are read: without that phase the stagger costs 2 us instead of saving 4.)
    if (!cached && a.stagger > 0 && blockIdx.x < 768u * (256 / kBwdBlock)) {
        const int steps = (int)(blockIdx.x / (256u * (256 / kBwdBlock))) * a.stagger;
        for (int k = 0; k < steps; k += 127) __builtin_amdgcn_s_sleep(127);
    }
    if (has_sh && wave_vis && !cached) {
        const float* src = a.in.shs + span_first;
        if (((span_first | span_len) & 3) == 0) {   // 16-B aligned span (always for M = 16): dwordx4 loads, six in
            const float4* src4 = reinterpret_cast<const float4*>(src);   // flight before the first LDS store (twelve, as in
            const int n4 = span_len >> 2;                                // the forward's colour kernel, cost a wave of occupancy)
            constexpr int kBatch = 6;
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
                            float* d = lds + bskew<true>(e);
                            d[0] = v[k].x;
                            d[1] = v[k].y;
                            d[2] = v[k].z;
                            d[3] = v[k].w;
                        } else {
                            lds[bskew<false>(e)] = v[k].x;
                            lds[bskew<false>(e + 1)] = v[k].y;
                            lds[bskew<false>(e + 2)] = v[k].z;
                            lds[bskew<false>(e + 3)] = v[k].w;
                        }
                    }
                }
            }
        } else {
            for (int e = lane; e < span_len; e += 64) lds[bskew<ROWS48>(e)] = src[e];
        }
    }
    __syncthreads();

    float dmean[3] = {0.f, 0.f, 0.f}, dcov6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dscale[3] = {0.f, 0.f, 0.f}, dq[4] = {0.f, 0.f, 0.f, 0.f};
    float g2x = 0.f, g2y = 0.f, dop = 0.f, dcol[3] = {0.f, 0.f, 0.f}, gcon[3] = {0.f, 0.f, 0.f};
    int K = 0;
    const ShRowLdsRW<ROWS48> row{ROWS48 ? lds + 49 * lane : lds, ROWS48 ? 0 : lane * 3 * M};
    if (vis) {
        const float mx = a.in.means3D[3 * i], my = a.in.means3D[3 * i + 1], mz = a.in.means3D[3 * i + 2];
        const GRec r = a.rec[i];
        // 2D-stage gradient row: final in acc[], or in <= tiles/64 + 2 ordered pieces.  A Gaussian whose pairs did not
        // all fit the pass's pair reservation (num_rendered > reserve: the farthest pairs were dropped and the pass is
        // flagged) gets zero 2D-stage gradients instead of a partial sum.
        const uint32_t start = r.pair_start, ntile = a.tiles[i];
        if (a.wave_part && start != 0xFFFFFFFFu && start + ntile <= a.header->num_pairs) {
            float acc9[kPairGrad];
            const uint32_t last = start + ntile - 1u;
            const uint32_t w0 = start >> 6, w1 = last >> 6;
            if (w0 == w1) {
#ifdef R3_ACC_IN_SLAB
                const float4* ap = reinterpret_cast<const float4*>(a.pair_grad + (size_t)last * kPairStride);
#else
                const float4* ap = reinterpret_cast<const float4*>(a.acc + (size_t)i * kAccStride);
#endif
                float4 a0 = ap[0], a1 = ap[1], a2 = ap[2];
#ifndef R3_NO_ZERO_ROW_SKIP
                if (__float_as_uint(a2.y) != a.stamp0 || __float_as_uint(a2.z) != a.stamp1)   // not written by this pass: zeros
                    a0 = a1 = a2 = make_float4(0.f, 0.f, 0.f, 0.f);
#endif
                acc9[0] = a0.x; acc9[1] = a0.y; acc9[2] = a0.z; acc9[3] = a0.w;
                acc9[4] = a1.x; acc9[5] = a1.y; acc9[6] = a1.z; acc9[7] = a1.w;
                acc9[8] = a2.x;
            } else {
#ifdef R3_WP_VEC
                const float4* w4 = reinterpret_cast<const float4*>(a.wave_part + (size_t)w0 * 2 * kPieceStride + kPieceStride);
                float4 p0 = w4[0], p1 = w4[1], p2 = w4[2];
                acc9[0] = p0.x; acc9[1] = p0.y; acc9[2] = p0.z; acc9[3] = p0.w;
                acc9[4] = p1.x; acc9[5] = p1.y; acc9[6] = p1.z; acc9[7] = p1.w;
                acc9[8] = p2.x;
                for (uint32_t w = w0 + 1; w <= w1; w++) {  // leading piece of every following group
                    w4 = reinterpret_cast<const float4*>(a.wave_part + (size_t)w * 2 * kPieceStride);
                    p0 = w4[0]; p1 = w4[1]; p2 = w4[2];
                    acc9[0] += p0.x; acc9[1] += p0.y; acc9[2] += p0.z; acc9[3] += p0.w;
                    acc9[4] += p1.x; acc9[5] += p1.y; acc9[6] += p1.z; acc9[7] += p1.w;
                    acc9[8] += p2.x;
                }
#else
                const float* wp = a.wave_part + (size_t)w0 * 2 * kPieceStride + kPieceStride;  // trailing piece of w0
#pragma unroll
                for (int k = 0; k < kPairGrad; k++) acc9[k] = wp[k];
                for (uint32_t w = w0 + 1; w <= w1; w++) {  // leading piece of every following group
                    wp = a.wave_part + (size_t)w * 2 * kPieceStride;
#pragma unroll
                    for (int k = 0; k < kPairGrad; k++) acc9[k] += wp[k];
                }
#endif
            }
            g2x = acc9[0];
            g2y = acc9[1];
            gcon[0] = acc9[2];
            gcon[1] = acc9[3];
            gcon[2] = acc9[4];
            dop = acc9[5];
            dcol[0] = acc9[6];
            dcol[1] = acc9[7];
            dcol[2] = acc9[8];
        }
        float sc[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, c6[6];
        if (a.in.cov3D_precomp) {
            for (int k = 0; k < 6; k++) c6[k] = a.in.cov3D_precomp[6 * i + k];
        } else {
            for (int k = 0; k < 3; k++) sc[k] = a.in.scales[3 * i + k];
            for (int k = 0; k < 4; k++) q[k] = a.in.rotations[4 * i + k];
            cov3d_from_scale_rot(sc, cam.scale_modifier, q, c6);  // recomputed, not stored by the forward
        }
        double dcov6d[6];
        if (F64) {
            cov2d_backward_f64(cam, mx, my, mz, c6, gcon[0], gcon[1], gcon[2], dcov6d, dmean);
#pragma unroll
            for (int k = 0; k < 6; k++) dcov6[k] = (float)dcov6d[k];
        } else {
            cov2d_backward(cam, mx, my, mz, c6, gcon[0], gcon[1], gcon[2], dcov6, dmean);
        }
        project_backward(cam, mx, my, mz, g2x, g2y, dmean);
        if (has_sh) {
            float mult = 0.f;
            if (a.lambda_sh != 0.f) {
                const uint32_t V = a.header->visible;
                mult = a.lambda_sh / (float)((int)V * 15 * 3);
            }
            const int deg = a.in.degrees[i];
            K = (deg + 1) * (deg + 1);
            if (cached) {
                float d9[9];
#pragma unroll
                for (int k = 0; k < 9; k++)   // (a Gaussian binned into no tile got no colour and left nothing: its dcol is 0)
                    d9[k] = (deg > 0 && ntile > 0u) ? a.sh_ddir[9 * (size_t)i + k] : 0.f;
                sh_backward<true>(deg, row, row, d9, mx, my, mz, cam.campos, r.width_clamp >> 16, dcol, 0.f, dmean);
            } else {
                sh_backward<false>(deg, row, row, nullptr, mx, my, mz, cam.campos, r.width_clamp >> 16, dcol, mult, dmean);
            }
        }
        if (a.in.scales) {
            if (F64)
                cov3d_backward_f64(sc, cam.scale_modifier, q, dcov6d, dscale, dq);
            else
                cov3d_backward(sc, cam.scale_modifier, q, dcov6, dscale, dq);
        }
        dop = opacity_backward(dop, r.op);
    }
    if (has_sh && valid) {
        if (wave_vis) {
            for (int e = 3 * K; e < 3 * M; e++) row.put(e, 0.f);  // bands above this Gaussian's degree / culled rows
        }
    }
    __syncthreads();
    if (has_sh) {
        float* dst = a.out.dL_dsh + span_first;
        if (((span_first | span_len) & 3) == 0) {   // dwordx4 stores of the gradient rows (or of zeros)
            float4* dst4 = reinterpret_cast<float4*>(dst);
            const int n4 = span_len >> 2;
            if (wave_vis) {
                for (int e4 = lane; e4 < n4; e4 += 64) {
                    const int e = e4 << 2;
                    if (ROWS48) {
                        const float* q = lds + bskew<true>(e);
                        dst4[e4] = make_float4(q[0], q[1], q[2], q[3]);
                    } else {
                        dst4[e4] = make_float4(lds[bskew<false>(e)], lds[bskew<false>(e + 1)], lds[bskew<false>(e + 2)],
                                               lds[bskew<false>(e + 3)]);
                    }
                }
            } else {
                for (int e4 = lane; e4 < n4; e4 += 64) dst4[e4] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else if (wave_vis) {
            for (int e = lane; e < span_len; e += 64) dst[e] = lds[bskew<ROWS48>(e)];
        } else {
            for (int e = lane; e < span_len; e += 64) dst[e] = 0.f;
        }
    }
    if (valid) {
        float* o;
        o = a.out.dL_dmean2D + 3 * (size_t)i;
        o[0] = g2x;
        o[1] = g2y;
        o[2] = 0.f;
        a.out.dL_dopacity[i] = dop;
        o = a.out.dL_dcolor + 3 * (size_t)i;
        o[0] = dcol[0];
        o[1] = dcol[1];
        o[2] = dcol[2];
        o = a.out.dL_dmean3D + 3 * (size_t)i;
        o[0] = dmean[0];
        o[1] = dmean[1];
        o[2] = dmean[2];
        o = a.out.dL_dcov3D + 6 * (size_t)i;
        for (int k = 0; k < 6; k++) o[k] = dcov6[k];
        o = a.out.dL_dscale + 3 * (size_t)i;
        o[0] = dscale[0];
        o[1] = dscale[1];
        o[2] = dscale[2];
        o = a.out.dL_drot + 4 * (size_t)i;
        o[0] = dq[0];
        o[1] = dq[1];
        o[2] = dq[2];
        o[3] = dq[3];
        if (a.out.dL_dconic) {
            o = a.out.dL_dconic + 4 * (size_t)i;
            o[0] = gcon[0];
            o[1] = gcon[1];
            o[2] = 0.f;
            o[3] = gcon[2];
        }
    }
}

void issue_preprocess_backward(const BwdPlan& p, const PreBwdArgs* a, hipStream_t s)
{
    const int blocks = (p.P + kBwdBlock - 1) / kBwdBlock;
    static const int lds_pad = env_int("R3DGS_PREBWD_LDS_PAD", 0, 0, 65536);
    if (p.M == 16) {
        if (p.f64_chain)
            hipLaunchKernelGGL((preprocess_bwd_kernel<true, true>), dim3(blocks), dim3(kBwdBlock), lds_pad, s, a);
        else
            hipLaunchKernelGGL((preprocess_bwd_kernel<true, false>), dim3(blocks), dim3(kBwdBlock), lds_pad, s, a);
    } else {
        if (p.f64_chain)
            hipLaunchKernelGGL((preprocess_bwd_kernel<false, true>), dim3(blocks), dim3(kBwdBlock), lds_pad, s, a);
        else
            hipLaunchKernelGGL((preprocess_bwd_kernel<false, false>), dim3(blocks), dim3(kBwdBlock), lds_pad, s, a);
    }
}

}  // namespace r3
