// blend.hip -- per-tile front-to-back alpha blending (forward) and its per-pixel backward, gfx950.
//
// Replaces cuda_rasterizer/forward.cu:461-582 renderCUDA and backward.cu:437-595 renderCUDA of
// /root/reference/submodules/diff-gaussian-rasterization.
//
// MI355X-first structure (not the reference's 256-thread block with two barriers per chunk):
//   * ONE 64-lane wave owns a pixel region of a tile and streams the tile's list on its own: single-wave
//     workgroups, no cross-wave barriers, so the CU's wave slots hide each other's gather latency.  A lane
//     carries PPL pixels (PPL 8x8 quadrants of the 16x16 tile): every LDS broadcast read of a Gaussian record
//     is amortised over PPL pixels per lane, and the backward's cross-lane reduction over PPL x 64 pixels.
//   * the list is fetched 64 entries at a time: lane j gathers the 48-byte GRec of entry j (one
//     cache-line-local load instead of the reference's five scattered arrays) into registers one chunk
//     ahead of its use, then parks it in LDS.
//   * while it holds entry j, lane j also runs the exact region pre-test (blend_math.h): more than half of
//     the reference's bounding-square tile entries reach no pixel of the tile, 72% no pixel of a given
//     quadrant.  The wave then walks only the surviving entries (bit scan over a wave-uniform mask) and
//     only their surviving quadrants.  Per-pixel decisions are untouched, so image, n_contrib and gradients
//     are what they are without the pre-test.
//   * the forward keeps its pre-test masks (one 64-bit word per chunk and quadrant) and the backward loads them
//     instead of repeating the minimisation.
//   * the conic is pre-scaled when an entry is staged, so that log2(G) is three FMAs per (pixel, entry); the
//     backward accumulates moments of m = G * dL/dalpha (sum m, sum m d, sum m d d^T) per pixel and forms the
//     gradients once per (tile, entry).
//   * backward: the 9 sums of an entry are added over the lane's pixels, reduced across the wave with DPP row
//     operations (bank-masked adds, wave_reduce9 below; finishing the last steps with 4 same-address LDS float
//     atomics instead was measured 45% slower), parked in LDS per list entry, and flushed once per 64-entry chunk
//     with plain stores into the entry's own slot of a per-(tile, Gaussian)-pair slab.  The reference issues
//     one global float atomic per (pixel, Gaussian, component); a first version here issued one per (tile,
//     Gaussian, component) and still spent 0.25 of 0.87 ms on them (11.8 M atomics per pass at the BASELINE
//     shape).  Now there are none: the per-Gaussian kernel sums a Gaussian's slots, which are contiguous.
//   * workgroup ids are remapped so that consecutive tiles run on the same XCD (shared L2 for the records of
//     Gaussians straddling neighbouring tiles).
#include <cstdlib>

#include "blend_math.h"
#include "common.h"

namespace r3 {

// (pointers out of a pass block go through global_ptr(), common.h: FLAT accesses would count on lgkmcnt and every wait for an
// LDS read would wait for the global gather in flight)
constexpr int kChunk = 64;

// debug builds only (tools/bwd_timeline.py): when and where every workgroup of the backward (-DR3_TIMELINE) or of the
// forward (-DR3_TIMELINE_FWD) blend ran
#if defined(R3_TIMELINE) || defined(R3_TIMELINE_FWD)
__device__ unsigned long long g_timeline[4 * 65536];
#define R3_TL_BEGIN_(id) const unsigned long long tl0_ = __builtin_amdgcn_s_memtime(); const uint32_t tlid_ = (id);
#define R3_TL_END_(extra)                                                                                                \
    if (threadIdx.x == 0 && tlid_ < 65536u) {                                                                            \
        g_timeline[4 * tlid_] = tl0_;                                                                                     \
        g_timeline[4 * tlid_ + 1] = __builtin_amdgcn_s_memtime();                                                         \
        g_timeline[4 * tlid_ + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |     \
                                    ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); \
        g_timeline[4 * tlid_ + 3] = (unsigned long long)(extra);                                                          \
    }
#endif
#if defined(R3_TIMELINE)
#define R3_TL_BEGIN(id) R3_TL_BEGIN_(id)
#define R3_TL_END(extra) R3_TL_END_(extra)
#else
#define R3_TL_BEGIN(id)
#define R3_TL_END(extra)
#endif
#if defined(R3_TIMELINE_FWD)
#define R3_TLF_BEGIN(id) R3_TL_BEGIN_(id)
#define R3_TLF_END(extra) R3_TL_END_(extra)
#else
#define R3_TLF_BEGIN(id)
#define R3_TLF_END(extra)
#endif

struct LdsRec {  // 48 B: a staged list entry, conic pre-scaled (QSplat); c.yzw = GRec's rect_min, width_clamp, pair_start
    float4 a;    // x, y, qa, qb
    float4 b;    // qc, op, r, g
    float4 c;    // b, -, -, -
};

__device__ __forceinline__ QSplat load_splat(const LdsRec& r)
{
    QSplat s;
    s.x = r.a.x;
    s.y = r.a.y;
    s.qa = r.a.z;
    s.qb = r.a.w;
    s.qc = r.b.x;
    s.op = r.b.y;
    s.r = r.b.z;
    s.g = r.b.w;
    s.b = r.c.x;
    return s;
}

__device__ __forceinline__ Splat splat_from_regs(const float4& a, const float4& b, const float4& c)
{
    Splat s;
    s.x = a.x;
    s.y = a.y;
    s.cA = a.z;
    s.cB = a.w;
    s.cC = b.x;
    s.op = b.y;
    s.r = b.z;
    s.g = b.w;
    s.b = c.x;
    return s;
}

// Lane j parks the entry it gathered (registers a, b, c = the GRec) with the conic pre-scaled.  Six narrow stores, each
// either straight out of the gather's destination registers or of freshly computed values: built as three float4
// the compiler shuffles the loaded components into new register tuples right behind the gather -- and waits for the
// gather there, which serialises the prefetch of chunk k+1 with the blending of chunk k.
__device__ __forceinline__ void stage_entry(LdsRec& dst, const float4& a, const float4& b, const float4& c)
{
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    // volatile: or the stores are merged back into float4; explicitly LDS: volatile accesses are not address-space inferred
#define R3_LDS_V(T) volatile T __attribute__((address_space(3)))*
    R3_LDS_V(float) d = (R3_LDS_V(float))reinterpret_cast<float*>(&dst);
    *(R3_LDS_V(v2f))d = v2f{a.x, a.y};
    *(R3_LDS_V(v2f))(d + 2) = v2f{(-0.5f * kLog2e) * a.z, -kLog2e * a.w};
    d[4] = (-0.5f * kLog2e) * b.x;
    d[5] = b.y;
    *(R3_LDS_V(v2f))(d + 6) = v2f{b.z, b.w};
    *(R3_LDS_V(v4f))(d + 8) = v4f{c.x, c.y, c.z, c.w};
#undef R3_LDS_V
}

// consecutive logical ids on one XCD: hardware places workgroup b on XCD b % 8 (speed only, never correctness)
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nblocks)
{
    const uint32_t per = nblocks >> 3;
    if (b >= per * 8) return b;  // tail
    return (b & 7u) * per + (b >> 3);
}

#define R3_DPP_ADD(v, ctrl, rmask) \
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, 0xf, true))

// sum over each 16-lane DPP row, result in every lane of the row
__device__ __forceinline__ float row_sum(float v)
{
    R3_DPP_ADD(v, 0xb1, 0xf);   // quad_perm [1,0,3,2]
    R3_DPP_ADD(v, 0x4e, 0xf);   // quad_perm [2,3,0,1]
    R3_DPP_ADD(v, 0x124, 0xf);  // row_ror:4
    R3_DPP_ADD(v, 0x128, 0xf);  // row_ror:8
    return v;
}

// full-wave sum; result valid in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
    v = row_sum(v);
    R3_DPP_ADD(v, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    R3_DPP_ADD(v, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave sum
    return v;
}

// Transposing wave reduction of the 9 per-entry sums (SplatSums, in its field order: components 0..8).  A plain butterfly
// costs 6 cross-lane adds per component (54, and the two cross-row steps need an extra move each).  Here every step
// over a lane bit exchanges DIFFERENT components between partner lanes (keep one, send the other), halving the live
// registers: 9 -> 5 -> 3 -> 2 -> 1, and only that one register crosses the 16-lane rows (2 ds_bpermute).  The two
// steps that start with the most registers go over lane bits 2 and 3 (row_ror:4 / row_ror:8), because a DPP bank mask
// selects exactly those bits (bank = 4 consecutive lanes of a row): two bank-masked v_add_f32_dpp into the same
// destination do "keep mine, add the partner's" without a v_cndmask.  The compiler cannot be asked for a bank-masked
// DPP add (it only folds full-mask moves), so those 14 instructions are inline assembly; the leading s_nop is the
// VALU-write -> DPP-read wait states the hazard recogniser cannot see INTO an asm block (it does pad behind one).
// Afterwards every lane with (lane & 2) == 0 holds the wave total of component reduce9_component(lane), every other
// lane component 8.  25 VALU + 2 DS per entry (was 38 + 6).
__device__ __forceinline__ int reduce9_component(int lane)
{
    return (lane & 2) ? 8 : 4 * (lane & 1) + 2 * ((lane >> 3) & 1) + ((lane >> 2) & 1);
}

__device__ __forceinline__ float wave_reduce9(const SplatSums& g, int lane)
{
    float u0, u1, u2, u3, t8, w0, w1;
    asm("s_nop 1\n\t"
        // lane bit 2: banks 0 and 2 keep the first of a pair, banks 1 and 3 the second
        "v_add_f32_dpp %[u0], %[mx], %[mx] row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[u0], %[my], %[my] row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[u1], %[ca], %[ca] row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[u1], %[cb], %[cb] row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[u2], %[cc], %[cc] row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[u2], %[op], %[op] row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[u3], %[cr], %[cr] row_ror:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %[u3], %[cg], %[cg] row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %[t8], %[bl], %[bl] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        // lane bit 3: banks 0 and 1 keep the first, banks 2 and 3 the second
        "v_add_f32_dpp %[w0], %[u0], %[u0] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[w0], %[u1], %[u1] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[w1], %[u2], %[u2] row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %[w1], %[u3], %[u3] row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %[t8], %[t8], %[t8] row_ror:8 row_mask:0xf bank_mask:0xf"
        : [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3), [t8] "=&v"(t8), [w0] "=&v"(w0), [w1] "=&v"(w1)
        : [mx] "v"(g.sx), [my] "v"(g.sy), [ca] "v"(g.sxx), [cb] "v"(g.sxy), [cc] "v"(g.syy), [op] "v"(g.sm), [cr] "v"(g.r),
          [cg] "v"(g.g), [bl] "v"(g.b));
    // now, summed over the 4 lanes of the row with the same lane & 3:  w0 = component 2 * bit3 + bit2,  w1 = 4 + that,  t8 = b
    R3_DPP_ADD(w0, 0xb1, 0xf);   // quad_perm [1,0,3,2]
    R3_DPP_ADD(w1, 0xb1, 0xf);
    R3_DPP_ADD(t8, 0xb1, 0xf);
    float z = (lane & 1) ? w1 : w0;
    R3_DPP_ADD(z, 0x4e, 0xf);    // quad_perm [2,3,0,1]
    R3_DPP_ADD(t8, 0x4e, 0xf);
    z = (lane & 2) ? t8 : z;
    z += __shfl_xor(z, 16);
    z += __shfl_xor(z, 32);
    return z;
}

template <int PPL>
__device__ __forceinline__ void pixel_of(int tile_x, int tile_y, int part, int q, int lane, int* px, int* py)
{
    const int b4 = part * PPL + q;  // 8x8 quadrant inside the 16x16 tile
    *px = tile_x * kTile + (b4 & 1) * 8 + (lane & 7);
    *py = tile_y * kTile + (b4 >> 1) * 8 + (lane >> 3);
}

constexpr int kFwdBatch = 4;   // entries per trip of the forward's entry loop: 2 / 3 / 4 -> 0.163 / 0.158 / 0.158 ms

// Blends batches of N surviving entries (front to back) while at least N survive; returns the entries left over.
template <int N, int PPL>
__device__ __forceinline__ unsigned long long fwd_batches(unsigned long long anymask, const unsigned long long* qmask,
                                                          const LdsRec* s_rec, const float* pxf, const float* pyf,
                                                          uint32_t first_pos, FwdPix* pix)
{
    while (__builtin_popcountll(anymask) >= N) {
        int j[N];
        QSplat sp[N];
        float al[N][PPL];
        bool inb[N][PPL];
#pragma unroll
        for (int i = 0; i < N; i++) {
            j[i] = __builtin_ctzll(anymask);
            anymask &= anymask - 1ull;
            sp[i] = load_splat(s_rec[j[i]]);
        }
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int q = 0; q < PPL; q++) al[i][q] = fwd_alpha(sp[i], pxf[q], pyf[q], &inb[i][q]);
#pragma unroll
        for (int i = 0; i < N; i++)
#pragma unroll
            for (int q = 0; q < PPL; q++)
                if (PPL == 1 || ((qmask[q] >> j[i]) & 1ull)) {   // one quadrant per wave: anymask IS its mask
                    float Tb;
                    fwd_apply(sp[i], al[i][q], inb[i][q], first_pos + (uint32_t)j[i] + 1u, pix[q], &Tb);
                }
    }
    return anymask;
}

template <int PPL, bool COUNTERS>
__global__ __launch_bounds__(64) void blend_fwd_kernel(const BlendFwdArgs* __restrict__ ap)
{
    __shared__ LdsRec s_rec[kChunk];
    const BlendFwdArgs a = *ap;   // pass block in device memory: scalar loads, once
    R3_TLF_BEGIN(blockIdx.x)
    __shared__ uint32_t s_id[kChunk];
    constexpr int PARTS = 4 / PPL;
    const int lane = threadIdx.x;
    const uint32_t wg = xcd_remap(blockIdx.x, a.nblocks);
    const uint32_t tile = wg / PARTS;
    const int part = (int)(wg % PARTS);
    const int tile_x = (int)(tile % (uint32_t)a.gx), tile_y = (int)(tile / (uint32_t)a.gx);
    // (stays in VGPRs: made wave-uniform by readfirstlane, the scalar loop control cost 0.174 -> 0.185 ms here)
    const uint2 range = global_ptr(a.ranges)[tile];

    float pxf[PPL], pyf[PPL], qx0[PPL], qy0[PPL];
    FwdPix pix[PPL];
    bool inside[PPL];
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        int px, py;
        pixel_of<PPL>(tile_x, tile_y, part, q, lane, &px, &py);
        const int b4 = part * PPL + q;
        qx0[q] = (float)(tile_x * kTile + (b4 & 1) * 8);   // first pixel column / row of quadrant q: wave-uniform,
        qy0[q] = (float)(tile_y * kTile + (b4 >> 1) * 8);  // computed from scalars so that it stays in SGPRs
        pxf[q] = (float)px;
        pyf[q] = (float)py;
        inside[q] = px < a.W && py < a.H;
        fwd_pix_init(pix[q], inside[q]);
    }

    // software pipeline: the records of chunk k+1 are gathered into registers while chunk k is blended
    // A list this long may be walked in segments by the backward (common.h): the state in front of every S-th entry is
    // parked for it while the walk is here anyway (4 KB per tile and checkpoint; nothing for the other tiles)
    const uint32_t seg_log2 = a.ckpt ? global_ptr(a.hdr)->ckpt : 0u;
    const bool segmented = seg_log2 != 0u && range.y - range.x >= global_ptr(a.hdr)->ckpt_thr;
    float4 nxa, nxb, nxc;
    uint32_t nxid = 0;
    nxa = nxb = nxc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (range.x + lane < range.y) {
        nxid = global_ptr(a.point_list)[range.x + lane];
        const auto* g = (const R3_GLOBAL float4*)global_ptr(a.rec + nxid);
        nxa = g[0];
        nxb = g[1];
        nxc = g[2];
    }
    // the list ids run one more chunk ahead than the records they index: one memory round trip per chunk, not two
    uint32_t nnid = 0;
    if (range.x + kChunk + lane < range.y) nnid = global_ptr(a.point_list)[range.x + kChunk + lane];
    for (uint32_t base = range.x; base < range.y; base += kChunk) {
        {
            bool live = false;
#pragma unroll
            for (int q = 0; q < PPL; q++) live |= fwd_pix_live(pix[q]);
            if (__ballot(live) == 0ull) break;  // every pixel of the region saturated
        }
        if (segmented) {
            const uint32_t rel = base - range.x;
            if (rel != 0u && (rel & ((1u << seg_log2) - 1u)) == 0u && (rel >> seg_log2) < (uint32_t)kFwdCkptMax) {
                auto* dst = global_ptr(a.ckpt) + ckpt_slot(range.x, rel >> seg_log2, tile, seg_log2) * 256;
#pragma unroll
                for (int q = 0; q < PPL; q++)   // a pixel that is done by now is never looked up here
                    dst[(part * PPL + q) * 64 + lane] = make_float4(pix[q].T, pix[q].C0, pix[q].C1, pix[q].C2);
            }
        }
        __syncthreads();
        stage_entry(s_rec[lane], nxa, nxb, nxc);
        if (COUNTERS) s_id[lane] = nxid;
        // region pre-test: lane j decides for entry j which of this wave's quadrants it can reach at all
        unsigned long long qmask[PPL], anymask = 0ull;
        {
            const Splat mine = splat_from_regs(nxa, nxb, nxc);
            const bool have = base + lane < range.y;
#pragma unroll
            for (int q = 0; q < PPL; q++) {
                qmask[q] = __ballot(have && region_may_contribute(mine, qx0[q], qx0[q] + 7.f, qy0[q], qy0[q] + 7.f));
                anymask |= qmask[q];
            }
            if (a.quad_masks) {   // kept for the backward, which walks the same chunks
                auto* dst =
                    global_ptr(a.quad_masks) + quad_mask_slot(range.x, (base - range.x) >> 6, tile) * 4 + (uint32_t)(part * PPL);
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < PPL; q++) dst[q] = qmask[q];
                }
            }
        }
        __syncthreads();
        {
            const uint32_t idx = base + kChunk + lane;
            if (idx < range.y) {
                nxid = nnid;
                const auto* g = (const R3_GLOBAL float4*)global_ptr(a.rec + nxid);
                nxa = g[0];
                nxb = g[1];
                nxc = g[2];
            }
            if (idx + kChunk < range.y) nnid = global_ptr(a.point_list)[idx + kChunk];
        }
        // Several surviving entries per trip: their alphas (the exp and the quadratic form, ~70 % of a step) do not depend on
        // the pixel state, so they are evaluated side by side before the sequential compositing of one after the other
        // -- independent work between dependent instructions, and the scalar loop control (bit scan, LDS address, branch)
        // is paid once per batch: the forward is as sensitive to those as to vector instructions (one entry per trip 0.176 ms,
        // two 0.171, two without per-entry "is there a second one" tests 0.159, four 0.158).
        if (!COUNTERS) {
            anymask = fwd_batches<kFwdBatch, PPL>(anymask, qmask, s_rec, pxf, pyf, base - range.x, pix);
            anymask = fwd_batches<2, PPL>(anymask, qmask, s_rec, pxf, pyf, base - range.x, pix);
        }
        while (anymask) {  // surviving entries, front to back
            const int j = __builtin_ctzll(anymask);
            anymask &= anymask - 1ull;
            const QSplat s = load_splat(s_rec[j]);
            const uint32_t pos1 = base - range.x + (uint32_t)j + 1u;
            int cnt = 0;
            float tsum = 0.f;
#pragma unroll
            for (int q = 0; q < PPL; q++) {
                if ((qmask[q] >> j) & 1ull) {
                    float Tb;
                    const int r = fwd_step(s, pxf[q], pyf[q], pos1, pix[q], &Tb);
                    if (COUNTERS && r == 1) {
                        cnt++;
                        tsum += Tb;
                    }
                }
            }
            if (COUNTERS) {  // forward.cu:560-564, one atomic pair per (region, Gaussian) instead of per pixel
                if (__ballot(cnt != 0) != 0ull) {
                    const float c = wave_sum_to_lane63((float)cnt);
                    const float t = wave_sum_to_lane63(tsum);
                    if (lane == 63) {
                        atomicAdd(a.touched + s_id[j], (int)c);
                        atomicAdd(a.transmittance + s_id[j], t);
                    }
                }
            }
        }
    }
    // deepest contributor of each quadrant: how far the backward will have to walk this tile's list (its launch order)
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        uint32_t m = pix[q].last;
        for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
        if (lane == 0) global_ptr(a.quad_depth)[tile * 4u + (uint32_t)(part * PPL + q)] = m;
    }
    if (segmented) {   // slot 0: the colour every contributor left (without the background): what lies behind a checkpoint
        auto* dst = global_ptr(a.ckpt) + ckpt_slot(range.x, 0u, tile, seg_log2) * 256;
#pragma unroll
        for (int q = 0; q < PPL; q++)
            dst[(part * PPL + q) * 64 + lane] = make_float4(fwd_pix_T(pix[q]), pix[q].C0, pix[q].C1, pix[q].C2);
    }
    const size_t plane = (size_t)a.W * a.H;
    const float bg0 = global_ptr(a.bg)[0], bg1 = global_ptr(a.bg)[1], bg2 = global_ptr(a.bg)[2];
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        if (inside[q]) {
            const size_t p = (size_t)a.W * (size_t)pyf[q] + (size_t)pxf[q];
            const float T = fwd_pix_T(pix[q]);
            global_ptr(a.final_T)[p] = T;
            global_ptr(a.n_contrib)[p] = pix[q].last;
            global_ptr(a.out_color)[p] = pix[q].C0 + T * bg0;
            global_ptr(a.out_color)[plane + p] = pix[q].C1 + T * bg1;
            global_ptr(a.out_color)[2 * plane + p] = pix[q].C2 + T * bg2;
        }
    }
    R3_TLF_END(range.y - range.x)
}

template <int PPL>
static void launch_fwd_ppl(const FwdPlan& p, const BlendFwdArgs* a, hipStream_t s)
{
    const uint32_t nblocks = (uint32_t)(p.gx * p.gy * (4 / PPL));   // == a->nblocks
    if (p.counters)
        hipLaunchKernelGGL((blend_fwd_kernel<PPL, true>), dim3(nblocks), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((blend_fwd_kernel<PPL, false>), dim3(nblocks), dim3(64), 0, s, a);
}

void issue_blend_forward(const FwdPlan& p, const BlendFwdArgs* a, hipStream_t s)
{
    if (p.fwd_ppl == 4)
        launch_fwd_ppl<4>(p, a, s);
    else if (p.fwd_ppl == 2)
        launch_fwd_ppl<2>(p, a, s);
    else
        launch_fwd_ppl<1>(p, a, s);
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
constexpr int kGradStride = 9;   // the 9 sums of a list entry (odd stride: the flush reads without bank conflicts)

// REUSE: the region pre-test masks are the forward's (BinState::quad_masks) instead of being recomputed per chunk.
// 6 waves per SIMD (80 VGPRs, three spilled values outside the entry loop).  With round 2's 96-register body 5 and 6
// measured the same; since the accumulated-colour recurrence and the offset products took 10 registers out of the
// loop, 6 is the faster by 1 % (0.3602 / 0.3603 vs 0.3644 / 0.3627 ms, two runs each, same box).
#ifndef R3_BWD_OCC
#define R3_BWD_OCC 6
#endif
// One workgroup = one UNIT: a tile, or -- for a tile whose list the pass splits (common.h) -- entries [lo, hi) of
// its list.  A segment that ends in front of a pixel's last contributor starts that pixel from the forward's checkpoint at
// `hi`: T there, and the colour that lies behind it = (final colour - colour in front of hi) / T, projected on the pixel's
// upstream gradient (the accumulated-colour state A of blend_math.h BwdPix); a pixel whose last contributor is inside
// [lo, hi) starts as ever, one that ends in front of lo has nothing to do here.  Every list entry belongs to exactly one
// segment, so every row of the per-pair slab still has ONE writer: no atomics, bit-reproducible.
// FIRST: this is the first kernel of the backward (no unit order): it installs the pass block for the kernels behind it and
// reads its own arguments from the kernarg segment; otherwise unit_order_kernel did and the arguments come from the block
// (a graph replay refreshes the by-value arguments of its first node only).
template <int PPL, bool REUSE, bool FIRST>
__global__ __launch_bounds__(64, R3_BWD_OCC) void blend_bwd_kernel(BwdPassArgs* dst, BwdPassArgs v)
{
    __shared__ LdsRec s_rec[kChunk];
    if (FIRST && blockIdx.x == 0) install_block_from_kernarg(dst, (int)threadIdx.x, 64);
    R3_TL_BEGIN(blockIdx.x)
    const BlendBwdArgs a = FIRST ? v.blend : dst->blend;
    __shared__ float s_grad[kChunk * kGradStride];
    constexpr int PARTS = 4 / PPL;
    const int lane = threadIdx.x;
    uint32_t wg, seg = 0u, nseg = 1u, walk_log2 = 0u;
    if (a.tile_order) {
        // entry b / lists of list b % lists (each list is heaviest first; the grid covers the most units a pass may have)
        const uint32_t g = blockIdx.x % kOrderLists, i = blockIdx.x / kOrderLists, per_list = a.units_cap / kOrderLists;
        const uint32_t* trailer = a.tile_order + a.units_cap + 2u * g;
        if (i >= trailer[0]) return;
        walk_log2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)trailer[1]);   // segment length this list's units walk
        const uint32_t u = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.tile_order[g * per_list + i]);
        wg = u & ((1u << kUnitTileBits) - 1u);
        seg = (u >> kUnitTileBits) & 63u;
        nseg = u >> (kUnitTileBits + 6u);
    } else {
        wg = xcd_remap(blockIdx.x, a.nblocks);
    }
    const uint32_t tile = wg / PARTS;
    const int part = (int)(wg % PARTS);
    const int tile_x = (int)(tile % (uint32_t)a.gx), tile_y = (int)(tile / (uint32_t)a.gx);
    uint2 range = a.ranges[tile];
    range.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)range.x);   // wave-uniform: kept in SGPRs
    range.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)range.y);
    const size_t plane = (size_t)a.W * a.H;
    const float bg0 = a.bg[0], bg1 = a.bg[1], bg2 = a.bg[2];

    float pxf[PPL], pyf[PPL], qx0[PPL], qy0[PPL];
    BwdPix pix[PPL];
    uint32_t lmax = 0;
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        int px, py;
        pixel_of<PPL>(tile_x, tile_y, part, q, lane, &px, &py);
        const int b4 = part * PPL + q;
        qx0[q] = (float)(tile_x * kTile + (b4 & 1) * 8);
        qy0[q] = (float)(tile_y * kTile + (b4 >> 1) * 8);
        pxf[q] = (float)px;
        pyf[q] = (float)py;
        const bool inside = px < a.W && py < a.H;
        BwdPix& p = pix[q];
        bwd_pix_init(p, 0.f, 0u, 0.f, 0.f, 0.f, 0.f);
        if (inside) {
            const size_t id = (size_t)a.W * py + px;
            const float g0 = a.dL_dpix[id], g1 = a.dL_dpix[plane + id], g2 = a.dL_dpix[2 * plane + id];
            bwd_pix_init(p, a.final_T[id], a.n_contrib[id], g0, g1, g2, bg0 * g0 + bg1 * g1 + bg2 * g2);
        }
        lmax = max(lmax, p.last);
    }
    // deepest contributor of each quadrant and of the whole region: nothing behind it matters to any pixel there
    uint32_t qlast[PPL];
#pragma unroll
    for (int q = 0; q < PPL; q++) {
        uint32_t m = pix[q].last;
        for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
        qlast[q] = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
    }
    for (int off = 32; off > 0; off >>= 1) lmax = max(lmax, (uint32_t)__shfl_xor((int)lmax, off));
    lmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)lmax);
    if (lmax == 0) {
        R3_TL_END(0)
        return;
    }
    // this unit's part of the list: [lo, hi)
    uint32_t lo = 0u, hi = lmax;
    if (nseg > 1u) {
        lo = seg << walk_log2;
        if (seg + 1u < nseg) hi = lo + (1u << walk_log2);   // (the unit order only makes segments with lo < lmax)
        if (hi < lmax) {
            // pixels whose last contributor lies behind this segment pass through it: their state in front of entry `hi`
            // (the forward's checkpoints are S = 2^hdr->ckpt apart; a pass may walk segments of 2 S, 4 S ...)
            const uint32_t seg_log2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.hdr->ckpt);
            const float4* ck = a.ckpt + ckpt_slot(range.x, hi >> seg_log2, tile, seg_log2) * 256;
            const float4* fin = a.ckpt + ckpt_slot(range.x, 0u, tile, seg_log2) * 256;
#pragma unroll
            for (int q = 0; q < PPL; q++) {
                BwdPix& p = pix[q];
                if (p.last > hi) {
                    const float4 c = ck[(part * PPL + q) * 64 + lane], f = fin[(part * PPL + q) * 64 + lane];
                    // p.T is the final transmittance here and p.A the background term bg . g
                    p.A = ((f.y - c.y) * p.g0 + (f.z - c.z) * p.g1 + (f.w - c.w) * p.g2 + p.T * p.A) * R3_RCP(c.x);
                    p.T = c.x;
                }
            }
#pragma unroll
            for (int q = 0; q < PPL; q++) qlast[q] = min(qlast[q], hi);
        }
    }

    const float half_w = 0.5f * (float)a.W, half_h = 0.5f * (float)a.H;  // backward.cu:498-499
    float4 nxa, nxb, nxc;
    uint32_t nxid = 0;
    nxa = nxb = nxc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int cfirst = (int)((hi - 1) / kChunk) * kChunk;
    if ((uint32_t)cfirst + (uint32_t)lane < hi) {
        nxid = a.point_list[range.x + (uint32_t)cfirst + (uint32_t)lane];
        const float4* g = reinterpret_cast<const float4*>(a.rec + nxid);
        nxa = g[0];
        nxb = g[1];
        nxc = g[2];
    }
    uint32_t nnid = 0;   // list ids run one more chunk ahead than the records they index
    if (cfirst >= kChunk) nnid = a.point_list[range.x + (uint32_t)(cfirst - kChunk) + (uint32_t)lane];
    // the forward's masks of a chunk travel with its records: lane q < 4 fetches quadrant q's word one chunk ahead
    const unsigned long long* const masks0 = REUSE ? a.quad_masks + quad_mask_slot(range.x, 0u, tile) * 4 : nullptr;
    unsigned long long nxm = 0ull;
    if (REUSE && lane < 4) nxm = masks0[(size_t)(cfirst >> 6) * 4 + lane];
    // lanes that park the reduced sums of an entry: lane -> component as wave_reduce9 leaves them
    const bool writer = lane < 16 && ((lane & 2) == 0 || lane == 2);
    float* const s_grad_slot = s_grad + reduce9_component(lane);
    SplatSums sg;   // zero whenever an entry starts: cleared after every reduction, untouched by entries without a hit
    sg.sx = sg.sy = sg.sxx = sg.sxy = sg.syy = sg.sm = sg.r = sg.g = sg.b = 0.f;
    for (int cbase = cfirst; cbase >= (int)lo; cbase -= kChunk) {
        __syncthreads();
        stage_entry(s_rec[lane], nxa, nxb, nxc);
        unsigned long long qmask[PPL], anymask = 0ull;
        {
            const Splat mine = splat_from_regs(nxa, nxb, nxc);
            const bool have = (uint32_t)cbase + (uint32_t)lane < hi;
#pragma unroll
            for (int q = 0; q < PPL; q++) {
                if (REUSE) {
                    const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)nxm, q);
                    const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(nxm >> 32), q);
                    qmask[q] = ((unsigned long long)mhi << 32) | mlo;
                } else {
                    qmask[q] = __ballot(have && region_may_contribute(mine, qx0[q], qx0[q] + 7.f, qy0[q], qy0[q] + 7.f));
                }
                // entries behind the quadrant's deepest contributor / behind this unit's segment (scalar arithmetic)
                const uint32_t left = qlast[q] > (uint32_t)cbase ? qlast[q] - (uint32_t)cbase : 0u;
                if (left < (uint32_t)kChunk) qmask[q] &= (1ull << left) - 1ull;
                anymask |= qmask[q];
            }
        }
        if (cbase >= (int)lo + kChunk) {  // gather the next (shallower) chunk while this one is processed; it is always full
            nxid = nnid;
            if (cbase >= 2 * kChunk) nnid = a.point_list[range.x + (uint32_t)(cbase - 2 * kChunk) + (uint32_t)lane];
            const float4* g = reinterpret_cast<const float4*>(a.rec + nxid);
            nxa = g[0];
            nxb = g[1];
            nxc = g[2];
            if (REUSE && lane < 4) nxm = masks0[(size_t)((cbase - kChunk) >> 6) * 4 + lane];
        }
        __syncthreads();
        const int n = (int)min((uint32_t)kChunk, hi - (uint32_t)cbase);
        unsigned long long contributed = 0ull;   // wave-uniform: entries of this chunk that got sums
        while (anymask) {  // surviving entries, back to front: highest set bit first
            const int j = 63 - __builtin_clzll(anymask);
            anymask &= ~(1ull << j);
            const QSplat s = load_splat(s_rec[j]);
            const uint32_t pos = (uint32_t)(cbase + j);
            unsigned long long hit = 0ull;
#pragma unroll
            for (int q = 0; q < PPL; q++)
                if ((qmask[q] >> j) & 1ull) {
                    BwdEval e;
                    const bool valid = bwd_test(s, pxf[q], pyf[q], pos, pix[q], e);
                    // ballot of ONE compare is that compare's SGPR pair; of the conjunction it is a v_cndmask + v_cmp.
                    // `hit` only gates the reduction, so it may over-approximate: power > 0 (a degenerate conic) is
                    // left out, such an entry then reduces and stores the zeros it accumulated.
                    hit |= __builtin_amdgcn_ballot_w64(e.in_list) & __builtin_amdgcn_ballot_w64(e.visible);
                    if (valid) bwd_accumulate(s, e, pix[q], sg);
                }
            if (hit != 0ull) {
                const float z = wave_reduce9(sg, lane);
                int joff = j * kGradStride;
                asm("" : "+s"(joff));   // stays a scalar multiply + v_add (else: one quarter-rate v_mad_u64_u32)
                if (writer) s_grad_slot[joff] = z;
                contributed |= 1ull << j;
                sg.sx = sg.sy = sg.sxx = sg.sxy = sg.syy = sg.sm = sg.r = sg.g = sg.b = 0.f;
            }
        }
        __syncthreads();
        if (lane < n && ((contributed >> lane) & 1ull)) {
            // park the entry's sums in ITS slot of the per-pair slab: slot = first pair of the Gaussian + row-major
            // index of this tile inside the Gaussian's tile rect (exactly the emission order of binning.hip).
            // Plain stores, one owner per slot -- the per-Gaussian kernel adds a Gaussian's slots up in order, so
            // the backward has no float atomics and is bit-reproducible.  Only a 1-byte flag per pair is zeroed
            // before the pass (3.6 MB instead of the 131 MB slab); rows without flag are never read.
            const float4 c = s_rec[lane].c;
            const uint32_t rect_min = __float_as_uint(c.y), width = __float_as_uint(c.z) & 0xffffu;
            const uint32_t slot = __float_as_uint(c.w) + ((uint32_t)tile_y - (rect_min >> 16)) * width +
                                  ((uint32_t)tile_x - (rect_min & 0xffffu));
            float4* dst = reinterpret_cast<float4*>(a.pair_grad + (size_t)slot * kPairStride);   // 48-B row, 3 x 16 B
            const float* src = s_grad + lane * kGradStride;
            SplatSums u;
            u.sx = src[0];
            u.sy = src[1];
            u.sxx = src[2];
            u.sxy = src[3];
            u.syy = src[4];
            u.sm = src[5];
            u.r = src[6];
            u.g = src[7];
            u.b = src[8];
            // moments -> gradients, once per (tile, entry); viewport factors of backward.cu:498-499 applied here
            const SplatGrad g = splat_grad_of(load_splat(s_rec[lane]), u);
            dst[0] = make_float4(g.mx * half_w, g.my * half_h, g.cA, g.cB);
            dst[1] = make_float4(g.cC, g.op, g.r, g.g);
            dst[2] = make_float4(g.b, 0.f, 0.f, 0.f);
            if (kPairStride >= 16) dst[3] = make_float4(0.f, 0.f, 0.f, 0.f);   // 64-byte rows: the whole burst is written
            a.pair_flag[slot] = 1;
        }
    }
    R3_TL_END(hi - lo)
}

#if defined(R3_TIMELINE) || defined(R3_TIMELINE_FWD)
extern "C" int r3dgs_debug_timeline(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 4 * (size_t)n);
}
#endif

// Launch order of the backward blend's units: heaviest first.  All units are resident at once for the first half of the
// kernel and each wave's duration is set by how many share its SIMD, so in row-major order the chip drains for the whole
// second half (profiles/r03_bwd_timeline_row_major.txt: 20 resident waves per CU for five tenths of a CU's span, then 17, 12, 9, 6,
// 3); started by decreasing weight, the long walks are under way when the short ones fill the gaps (0.367 -> 0.313 ms on
// the metric shape).  A unit is a tile, or one segment of a tile whose list is long and deep (common.h):
// on a scene with the load of a real capture the heaviest tile walks three times the mean and the kernel was as long as
// that walk.  Weight of a unit = sum over the four quadrants of the entries it will visit there (ImageState::quad_depth
// clamped to the segment).  Tiles nothing contributed to are left out.  Counting sort over 1024 weight classes (64
// classes: +2 us for the stage, 16: +10, 4: +22), one workgroup per unit list (below); the order inside a class is whatever
// the LDS atomics make it -- every unit's arithmetic is its own, so the gradients do not depend on the order
// (tests/test_gpu_parity.py compares the two orders bit for bit).  The hardware deals consecutive workgroups over the eight
// XCDs: as ONE sorted list every XCD got every eighth unit -- equal work per XCD, and every XCD's L2 saw every Gaussian's
// record (FETCH_SIZE 62 -> 150 MB); keeping each XCD on a BAND of the image and ordering inside the band only (74 MB)
// measured 0.315 ms against 0.305 ms for the stage.  The eight lists of 4 x 4 tile blocks (common.h TileGrid) are the
// middle: each XCD walks blocks from all over the image, heaviest first, and neighbouring tiles stay on one XCD.
constexpr int kOrderThreads = 1024, kOrderClasses = 1024;

// segments of 2^walk entries the backward walks a tile in: 1 unless the forward left checkpoints for it (list length)
// and it is deeper than one segment
__device__ __forceinline__ uint32_t unit_segments(uint32_t list_len, uint32_t deepest, uint32_t thr, uint32_t walk)
{
    if (list_len < thr || deepest <= (1u << walk)) return 1u;
    return min((deepest + (1u << walk) - 1u) >> walk, (uint32_t)kBwdSegMax);
}
// entries segment `s` of `n` visits in the four quadrants whose deepest contributors are d
__device__ __forceinline__ uint32_t unit_weight(const uint4& d, uint32_t s, uint32_t n, uint32_t walk)
{
    if (n == 1u) return d.x + d.y + d.z + d.w;
    const uint32_t lo = s << walk, hi = s + 1u < n ? lo + (1u << walk) : 0xFFFFFFFFu;
    auto part = [&](uint32_t q) { return min(max(q, lo), hi) - lo; };
    return part(d.x) + part(d.y) + part(d.z) + part(d.w);
}

// KEEP: tiles a thread holds in registers between the passes.  Since the order became kOrderLists lists a thread of the metric
// shape holds ONE tile (864 slots per list), and every further slot still issued its (clamped) loads and its unrolled counting
// code: 12.7 us with eight slots, 7.5 with two (profiles/r05_exp_unit_order.txt); the launch takes the smallest that covers
// the image's longest list (1: <= 1024 slots per list, i.e. up to ~8 k tiles; 2; 8: up to 64 k tiles, beyond that the
// passes re-read).
template <int KEEP>
__global__ __launch_bounds__(kOrderThreads) void unit_order_kernel(BwdPassArgs* dst, BwdPassArgs v)
{
    __shared__ uint32_t s_count[kOrderClasses];
    __shared__ uint32_t s_scan[kOrderThreads / 64];
    constexpr int kWalks = 4;   // segment lengths a pass may walk: S, 2 S, 4 S, 8 S (the first whose units fit the launch)
    __shared__ uint32_t s_max;
    const uint32_t tid = threadIdx.x, lane_id = tid & 63u;
    if (blockIdx.x == 0) install_block_from_kernarg(dst, (int)tid, kOrderThreads);   // first kernel of the backward: the pass block
    // One workgroup per LIST: workgroup g orders the tiles of list g (common.h TileGrid: the 4 x 4 tile blocks g, g + lists, ...
    // of the image) with its own class counters, no word exchanged with the others; the blend kernel's workgroup b takes entry
    // b / lists of list b % lists, so the lists are consumed side by side and the launch order is the interleaving of eight
    // orders of statistically alike tile sets.
    // (As ONE workgroup over all tiles the kernel was bound by the vector issue of the one compute unit it ran on: 18 us.)
    const uint32_t list = blockIdx.x, all_tiles = v.blend.nblocks;
    const TileGrid grid{(uint32_t)v.blend.gx, all_tiles / (uint32_t)v.blend.gx};
    const uint4* __restrict__ qd = reinterpret_cast<const uint4*>(v.blend.quad_depth);
    const uint2* __restrict__ ranges = v.blend.ranges;
    const uint32_t cap = v.blend.units_cap / kOrderLists;                 // slots of this list
    uint32_t* __restrict__ order = v.blend.tile_order + list * cap;
    uint32_t* __restrict__ trailer = v.blend.tile_order + v.blend.units_cap + 2u * list;
    const uint32_t n_tiles = grid.list_slots(list);   // slots of this list; a slot of a block at the image's edge may hold no tile
    auto tile_of = [&](uint32_t j) { return grid.list_tile(list, j); };   // slot -> tile (~0: none)
    const uint32_t seg_log2 = (v.blend.segments != 0 && v.blend.ckpt != nullptr && all_tiles <= (1u << kUnitTileBits))
                                  ? v.blend.hdr->ckpt : 0u;   // 0: the forward left no checkpoints / segments are off
    const uint32_t thr = seg_log2 ? v.blend.hdr->ckpt_thr : 0xFFFFFFFFu;
    // Units a list may hold.  The bound is taken from the pass's pair count, not from the capacity its binning blob happens
    // to have: the exact-size path and the reserved path of one view then walk the same segments and give the same bits.
    const uint32_t fit = bwd_list_fit(v.blend.hdr->num_pairs, grid);   // <= cap: the reservation holds the pairs
    s_count[tid] = 0u;
    if (tid == 0) s_max = 0u;
    __syncthreads();
    // up to KEEP tiles per thread stay in registers between the passes (one memory round trip instead of four: the
    // kernel is a chain of latencies, 9.5 us when every pass went back to memory); a larger grid re-reads
    const bool keep = n_tiles <= (uint32_t)(KEEP * kOrderThreads);
    uint4 w[KEEP];
    uint32_t len[KEEP], tl[KEEP];   // weights, list length and tile of the kept slots
    auto load = [&](uint32_t t, uint4& d, uint32_t& l) {
        d = make_uint4(0u, 0u, 0u, 0u);
        l = 0u;
        const uint32_t tile = t < n_tiles ? tile_of(t) : 0xFFFFFFFFu;
        if (tile != 0xFFFFFFFFu) {
            d = qd[tile];
            const uint2 r = ranges[tile];
            l = r.y - r.x;
        }
    };
    if (keep) {
        // every load in flight before the first use (written as `load` per tile the compiler reused one register pair for the
        // ranges and waited for each of the loads in turn: 12 us of the kernel's 22)
        uint2 rr[KEEP];
#pragma unroll
        for (int k = 0; k < KEEP; k++) {
            const uint32_t j = tid + (uint32_t)(k * kOrderThreads);
            tl[k] = j < n_tiles ? tile_of(j) : 0xFFFFFFFFu;
            w[k] = qd[min(tl[k], all_tiles - 1u)];
            rr[k] = ranges[min(tl[k], all_tiles - 1u)];
        }
#pragma unroll
        for (int k = 0; k < KEEP; k++) {
            const bool in = tl[k] != 0xFFFFFFFFu;
            if (!in) w[k] = make_uint4(0u, 0u, 0u, 0u);
            len[k] = in ? rr[k].y - rr[k].x : 0u;
        }
    }
    auto wave_prefix = [&](uint32_t x, uint32_t* total) {   // exclusive prefix and total of x over the wave
        uint32_t incl = x;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
            if (lane_id >= (uint32_t)off) incl += up;
        }
        *total = (uint32_t)__shfl((int)incl, 63);
        return incl - x;
    };
    // (the kernel runs on ONE compute unit per list and is bound by its vector-instruction issue -- integer min / max at half
    // rate -- so nothing is worked out twice: segment counts and the classes of a tile's first three partial segments wait in
    // registers for the placement pass)
    constexpr uint32_t kCached = 3u;
    uint32_t nf[KEEP], pc[KEEP];   // segments | full segments << 8, and the cached classes, of the kept tiles
    // The segment length is settled by TRYING: the counting pass runs with the shortest segments (2^seg_log2 entries) and its
    // class scan says how many units that makes; only a pass whose lists are so long that they exceed what the list may
    // launch (tens of millions of pairs: fit = tiles + min(pairs / S, 8 tiles) / lists) counts again with 2 S, 4 S, 8 S and
    // finally whole tiles.  (Round 5 counted the units of all four candidate lengths in a pass of their own before the
    // counting pass, on every backward: 1.3 of the kernel's 12.7 us)
    uint32_t walk = seg_log2, total_units = 0u, full_class = 0u;
    float scale = 0.f;
    for (;;) {
        // The heaviest unit.  A pass that splits knows a bound without looking: a whole tile is shorter than `thr` entries (or it
        // would be split) and no deeper than its list, a segment weighs at most 4 x its length -- only the last segment of a tile
        // capped at kBwdSegMax can be heavier, and lands in the heaviest class.  Without segments: one more pass over the tiles.
        uint32_t heaviest;
        if (walk) {
            heaviest = max(4u << walk, 4u * min(thr, 1u << 20));
        } else {
            uint32_t kmax = 0u;
            if (keep) {
#pragma unroll
                for (int k = 0; k < KEEP; k++) kmax = max(kmax, w[k].x + w[k].y + w[k].z + w[k].w);
            } else {
                for (uint32_t t = tid; t < n_tiles; t += kOrderThreads) {
                    uint4 d;
                    uint32_t l;
                    load(t, d, l);
                    kmax = max(kmax, d.x + d.y + d.z + d.w);
                }
            }
            for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off));
            if (lane_id == 0u) atomicMax(&s_max, kmax);
            __syncthreads();
            heaviest = s_max;
        }
        scale = (float)(kOrderClasses - 1) / (float)max(heaviest, 1u);
        // class 0 = the heaviest
        auto klass = [&](uint32_t wt) {
            return (uint32_t)(kOrderClasses - 1) - min((uint32_t)((float)wt * scale), (uint32_t)(kOrderClasses - 1));
        };
        full_class = klass(4u << walk);
        auto segs = [&](const uint4& d, uint32_t l) -> uint32_t {
            if (d.x + d.y + d.z + d.w == 0u) return 0u;   // nothing contributed to this tile: no unit
            return walk ? unit_segments(l, max(max(d.x, d.y), max(d.z, d.w)), thr, walk) : 1u;
        };
        // Leading segments of a split tile that all four quadrants walk in full: they all weigh 4 x the segment length -- one
        // class, thousands of units in a deep scene, and one LDS counter they would all queue on (the kernel took 52 us at
        // 2 M Gaussians that way).  A tile's FULL segments are therefore counted and placed as a group, and the groups of the
        // tiles a wave holds at the same step add up among themselves (a shuffle scan): one atomic per wave and step.
        auto full_segments = [&](const uint4& d, uint32_t n) -> uint32_t {
            if (n <= 1u) return 0u;
            const uint32_t f = min(min(d.x, d.y), min(d.z, d.w)) >> walk;
            return min(f, n == (uint32_t)kBwdSegMax ? n - 1u : n);   // (a capped tile's last segment takes the rest: not "full")
        };
        // (a) the partial units of a tile: one LDS atomic each; (b) its full segments: the lanes of a wave add theirs up (one
        // shuffle scan per pass over the sum of a thread's tiles) and send one atomic.
        auto count_partials = [&](const uint4& d, uint32_t n, uint32_t f, uint32_t* classes) {
            uint32_t packed = 0u;
            for (uint32_t k = f; k < n; k++) {
                const uint32_t c = klass(unit_weight(d, k, n, walk));
                atomicAdd(&s_count[c], 1u);
                if (k - f < kCached) packed |= c << (10u * (k - f));
            }
            *classes = packed;
        };
        if (keep) {
            uint32_t fsum = 0u, total;
#pragma unroll
            for (int k = 0; k < KEEP; k++) {
                const uint32_t n = segs(w[k], len[k]), f = full_segments(w[k], n);
                nf[k] = n | (f << 8);
                fsum += f;
                count_partials(w[k], n, f, &pc[k]);
            }
            wave_prefix(fsum, &total);
            if (lane_id == 0u && total) atomicAdd(&s_count[full_class], total);
        } else {
            for (uint32_t t0 = 0; t0 < n_tiles; t0 += kOrderThreads) {   // uniform trip count: the wave scan needs every lane
                uint4 d;
                uint32_t l, unused, total;
                load(t0 + tid, d, l);
                const uint32_t n = segs(d, l), f = full_segments(d, n);
                count_partials(d, n, f, &unused);
                wave_prefix(f, &total);
                if (lane_id == 0u && total) atomicAdd(&s_count[full_class], total);
            }
        }
        __syncthreads();
        // exclusive scan of the class counts (one class per thread)
        const uint32_t mine = s_count[tid];
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off);
            if (lane_id >= (uint32_t)off) incl += up;
        }
        if (lane_id == 63u) s_scan[tid >> 6] = incl;
        __syncthreads();
        uint32_t before = 0u;
        total_units = 0u;
        for (uint32_t k = 0; k < (uint32_t)(kOrderThreads / 64); k++) {
            const uint32_t c = s_scan[k];
            if (k < (tid >> 6)) before += c;
            total_units += c;
        }
        __syncthreads();
        const bool fits = walk == 0u || total_units <= fit;   // the same for every thread
        s_count[tid] = fits ? before + incl - mine : 0u;       // first slot of the class, then its cursor / cleared for the retry
        if (tid == 0) s_max = 0u;
        __syncthreads();
        if (fits) break;
        walk = walk + 1u < seg_log2 + (uint32_t)kWalks ? walk + 1u : 0u;   // longer segments; none fits: whole tiles
    }
    if (tid == 0) {
        trailer[0] = total_units;   // the units of this list (<= cap by the choice of `walk`; whole tiles: <= list_tiles_max)
        trailer[1] = walk;          // log2 of the segment length they walk (0: whole tiles)
    }
    auto klass = [&](uint32_t wt) {
        return (uint32_t)(kOrderClasses - 1) - min((uint32_t)((float)wt * scale), (uint32_t)(kOrderClasses - 1));
    };
    auto place = [&](uint32_t tile, const uint4& d, uint32_t n, uint32_t f, uint32_t base, uint32_t classes, bool cached) {
        const uint32_t word = tile | (n << (kUnitTileBits + 6u));
        for (uint32_t k = 0; k < f; k++) order[base + k] = word | (k << kUnitTileBits);
        for (uint32_t k = f; k < n; k++) {
            const uint32_t c = (cached && k - f < kCached) ? (classes >> (10u * (k - f))) & 1023u
                                                           : klass(unit_weight(d, k, n, walk));
            order[atomicAdd(&s_count[c], 1u)] = word | (k << kUnitTileBits);
        }
    };
    auto full_base = [&](uint32_t f) {   // where this lane's full segments go: one atomic per wave
        uint32_t total;
        const uint32_t ahead = wave_prefix(f, &total);
        uint32_t base = 0u;
        if (lane_id == 0u && total) base = atomicAdd(&s_count[full_class], total);
        return (uint32_t)__shfl((int)base, 0) + ahead;
    };
    if (keep) {
        uint32_t fsum = 0u;
#pragma unroll
        for (int k = 0; k < KEEP; k++) fsum += nf[k] >> 8;
        uint32_t base = full_base(fsum);
#pragma unroll
        for (int k = 0; k < KEEP; k++) {
            place(tl[k], w[k], nf[k] & 255u, nf[k] >> 8, base, pc[k], true);
            base += nf[k] >> 8;
        }
    } else {
        for (uint32_t t0 = 0; t0 < n_tiles; t0 += kOrderThreads) {
            uint4 d;
            uint32_t l;
            load(t0 + tid, d, l);
            const uint32_t n = d.x + d.y + d.z + d.w == 0u ? 0u
                               : (walk ? unit_segments(l, max(max(d.x, d.y), max(d.z, d.w)), thr, walk) : 1u);
            uint32_t f = 0u;
            if (n > 1u) {
                f = min(min(d.x, d.y), min(d.z, d.w)) >> walk;
                f = min(f, n == (uint32_t)kBwdSegMax ? n - 1u : n);
            }
            place(t0 + tid < n_tiles ? tile_of(t0 + tid) : 0u, d, n, f, full_base(f), 0u, false);
        }
    }
}

template <bool REUSE>
static void launch_bwd(uint32_t nblocks, uint32_t units_cap, BwdPassArgs* dst, const BwdPassArgs& v, hipStream_t s)
{
    if (v.blend.tile_order) {
        static const int lds_pad = env_int("R3DGS_BWD_LDS_PAD", 0, 0, 65536);
        // one workgroup per unit the pass MAY have (those beyond its count leave at once)
        hipLaunchKernelGGL((blend_bwd_kernel<4, REUSE, false>), dim3(units_cap), dim3(64), lds_pad, s, dst, v);
    } else {
        hipLaunchKernelGGL((blend_bwd_kernel<4, REUSE, true>), dim3(nblocks), dim3(64), 0, s, dst, v);
    }
}

// the launch order of the backward blend's units (and the pass block: this is then the first kernel of the backward)
void issue_unit_order(const BwdPlan& p, BwdPassArgs* dst, const BwdPassArgs& v, hipStream_t s)
{
    if (!v.blend.tile_order) return;
    // slots of the longest list decide how many tiles a thread of the order kernel keeps in registers
    const uint32_t slots = TileGrid{(uint32_t)p.gx, (uint32_t)p.gy}.list_slots(0);
    if (slots <= (uint32_t)kOrderThreads)
        hipLaunchKernelGGL(unit_order_kernel<1>, dim3(kOrderLists), dim3(kOrderThreads), 0, s, dst, v);
    else if (slots <= 2u * (uint32_t)kOrderThreads)
        hipLaunchKernelGGL(unit_order_kernel<2>, dim3(kOrderLists), dim3(kOrderThreads), 0, s, dst, v);
    else
        hipLaunchKernelGGL(unit_order_kernel<8>, dim3(kOrderLists), dim3(kOrderThreads), 0, s, dst, v);
}

void issue_blend_backward(const BwdPlan& p, BwdPassArgs* dst, const BwdPassArgs& v, hipStream_t s)
{
    // one wave per tile (PPL = 4): the per-pair gradient slab has exactly one owner per (tile, Gaussian)
    const uint32_t nblocks = (uint32_t)(p.gx * p.gy);   // == v.blend.nblocks
    if (v.blend.quad_masks)
        launch_bwd<true>(nblocks, p.units_cap, dst, v, s);
    else
        launch_bwd<false>(nblocks, p.units_cap, dst, v, s);
}

}  // namespace r3
