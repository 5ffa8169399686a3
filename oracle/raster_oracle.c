/*
 * raster_oracle.c -- CPU restatement of the reduced-3dgs tile rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (reduced-3dgs_amd/) may
 * include, link or call this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker.
 *
 * PARITY STATUS: "parity unpinned" against the reference CUDA kernels -- the
 * reference ships no tests / golden vectors and cannot be compiled here (GLM
 * submodule absent, no nvcc).  What IS pinned (tests/golden,
 * tests/test_oracle_pins.py): SH colour vs utils/sh_utils.py:eval_sh, camera
 * matrices vs utils/graphics_utils.py, and the analytic backward vs an fp64
 * autograd restatement (oracle/torch_ref.py).
 *
 * Every function cites the reference file:line (relative to /root/reference,
 * DGR = submodules/diff-gaussian-rasterization) whose arithmetic it follows.
 * All arithmetic is IEEE fp32, one rounding per operation (build with
 * -ffp-contract=off); sums are evaluated left-to-right / ascending index as
 * written in the reference expressions (GLM products: ascending k).
 *
 * Matrix convention (DGR/cuda_rasterizer/auxiliary.h:58-77): flat m[4*c + r] is
 * entry (row r, col c) of the mathematical column-vector matrix.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16 /* DGR/cuda_rasterizer/config.h: BLOCK_X = BLOCK_Y = 16 */

/* ---- constants: DGR/cuda_rasterizer/auxiliary.h:22-39 ------------------- */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* float -> int truncation with the saturating behaviour of the GPU conversion */
static inline int f2i(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483520.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int)(-2147483647 - 1);
    return (int)v;
}

/* auxiliary.h:58-77 transformPoint4x3 / 4x4 */
static inline void xform4x3(const float* m, const float* p, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform4x4(const float* m, const float* p, float* o)
{
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* auxiliary.h:41-44 ndc2Pix -- evaluated in double exactly like the reference
 * (the literals 1.0 / 0.5 are double), then rounded once to float. */
static inline float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

/* auxiliary.h:46-56 getRect */
static inline void get_rect(float px, float py, int radius, int gx, int gy, int* rmin, int* rmax)
{
    float r = (float)radius;
    rmin[0] = imin(gx, imax(0, f2i((px - r) / (float)TILE)));
    rmin[1] = imin(gy, imax(0, f2i((py - r) / (float)TILE)));
    rmax[0] = imin(gx, imax(0, f2i((px + r + (float)(TILE - 1)) / (float)TILE)));
    rmax[1] = imin(gy, imax(0, f2i((py + r + (float)(TILE - 1)) / (float)TILE)));
}

/* world->view rotation Rw(i,j) = viewmatrix[4*j + i] */
#define RW(vm, i, j) ((vm)[4 * (j) + (i)])

/* forward.cu:207-241 computeCov3D: Sigma = R diag(s*mod)^2 R^T, q=(r,x,y,z) assumed unit.
 * GLM: M = S * R_glm with R_glm = R^T, Sigma = M^T M, so
 * Sigma(a,b) = sum_k (s_k R(a,k)) * (s_k R(b,k)), k ascending. */
static void quat_to_R(const float* q, float R[3][3])
{
    float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = 1.f - 2.f * (y * y + z * z);
    R[0][1] = 2.f * (x * y - r * z);
    R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z);
    R[1][1] = 1.f - 2.f * (x * x + z * z);
    R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y);
    R[2][1] = 2.f * (y * z + r * x);
    R[2][2] = 1.f - 2.f * (x * x + y * y);
}
static void cov3d_from_scale_rot(const float* scale, float mod, const float* q, float* cov6)
{
    float R[3][3], Mk[3][3]; /* Mk[k][j] = s_k * R(j,k)  (= M_glm(row k, col j)) */
    float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    quat_to_R(q, R);
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 3; j++) Mk[k][j] = s[k] * R[j][k];
#define SIG(a, b) (Mk[0][a] * Mk[0][b] + Mk[1][a] * Mk[1][b] + Mk[2][a] * Mk[2][b])
    cov6[0] = SIG(0, 0);
    cov6[1] = SIG(0, 1);
    cov6[2] = SIG(0, 2);
    cov6[3] = SIG(1, 1);
    cov6[4] = SIG(1, 2);
    cov6[5] = SIG(2, 2);
#undef SIG
}

/* A = J * Rw (2x3), shared by forward.cu:162-202 and backward.cu:199-227.
 * Returns clamped t, and the clamp masks used by the backward. */
static void ewa_A(const float* mean, const float* vm, float fx, float fy, float tanx, float tany,
                  float A[2][3], float t[3], float* xmul, float* ymul)
{
    xform4x3(vm, mean, t);
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fminf_(limx, fmaxf_(-limx, txtz)) * t[2];
    t[1] = fminf_(limy, fmaxf_(-limy, tytz)) * t[2];
    if (xmul) *xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    if (ymul) *ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float J00 = fx / t[2], J02 = -(fx * t[0]) / (t[2] * t[2]);
    const float J11 = fy / t[2], J12 = -(fy * t[1]) / (t[2] * t[2]);
    /* T_glm = W_glm * J_glm : A(i,j) = sum_k Rw(k,j) * J(i,k), k ascending */
    for (int j = 0; j < 3; j++) {
        A[0][j] = RW(vm, 0, j) * J00 + RW(vm, 1, j) * 0.0f + RW(vm, 2, j) * J02;
        A[1][j] = RW(vm, 0, j) * 0.0f + RW(vm, 1, j) * J11 + RW(vm, 2, j) * J12;
    }
}

/* forward.cu:162-202 computeCov2D: cov = (A Sigma) A^T, +0.3 on the diagonal */
static void cov2d(const float A[2][3], const float* c6, float* a, float* b, float* c)
{
    const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    float AS[2][3];
    for (int i = 0; i < 2; i++)
        for (int k = 0; k < 3; k++) AS[i][k] = A[i][0] * S[0][k] + A[i][1] * S[1][k] + A[i][2] * S[2][k];
    *a = (AS[0][0] * A[0][0] + AS[0][1] * A[0][1] + AS[0][2] * A[0][2]) + 0.3f;
    *b = AS[1][0] * A[0][0] + AS[1][1] * A[0][1] + AS[1][2] * A[0][2]; /* cov[0][1] = col0,row1 */
    *c = (AS[1][0] * A[1][0] + AS[1][1] * A[1][1] + AS[1][2] * A[1][2]) + 0.3f;
}

/* real SH basis, forward.cu:105-159 (same association as the reference expressions) */
static void sh_basis(int deg, float x, float y, float z, float* Y)
{
    Y[0] = SH_C0;
    if (deg > 0) {
        Y[1] = -(SH_C1 * y);
        Y[2] = SH_C1 * z;
        Y[3] = -(SH_C1 * x);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Y[4] = SH_C2[0] * xy;
            Y[5] = SH_C2[1] * yz;
            Y[6] = SH_C2[2] * (2.0f * zz - xx - yy);
            Y[7] = SH_C2[3] * xz;
            Y[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                Y[9] = SH_C3[0] * y * (3.0f * xx - yy);
                Y[10] = SH_C3[1] * xy * z;
                Y[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
                Y[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                Y[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
                Y[14] = SH_C3[5] * z * (xx - yy);
                Y[15] = SH_C3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}

/* forward.cu:19-36 getSHOffset: offset (in float3 units) into the ragged,
 * degree-sorted SH buffer; degree implied by position. */
static int ragged_sh_offset(int idx, const int* coeffs, const int* perband, const int* cumsum, int* deg)
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

/* forward.cu:105-159 computeColorFromSH */
static void sh_to_rgb(int deg, const float* sh /* [K][3] */, const float* mean, const float* campos,
                      float* rgb, unsigned char* clamped)
{
    float d[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
    float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] = d[0] / len;
    d[1] = d[1] / len;
    d[2] = d[2] / len;
    float Y[16];
    sh_basis(deg, d[0], d[1], d[2], Y);
    int K = (deg + 1) * (deg + 1);
    for (int ch = 0; ch < 3; ch++) {
        float r = Y[0] * sh[ch];
        for (int k = 1; k < K; k++) r = r + Y[k] * sh[3 * k + ch];
        r += 0.5f;
        clamped[ch] = (r < 0);
        rgb[ch] = fmaxf_(r, 0.0f);
    }
}

/* -------------------------------------------------------------------------
 * Per-Gaussian forward: forward.cu:353-456 preprocessCUDA (dense SH, stride M)
 * and forward.cu:245-350 variableSHPreprocessCUDA (ragged SH, when coeffs!=NULL).
 * Optional inputs are NULL when absent (forward.cu:403,441).
 * Outputs for culled Gaussians: radii = tiles_touched = 0, rest untouched.
 * ------------------------------------------------------------------------- */
void orc_preprocess(int P, int M, const int* degs, const float* means, const float* scales, float mod,
                    const float* rots, const float* opac_raw, const float* shs, const float* cov3D_precomp,
                    const float* colors_precomp, const float* vm, const float* pm, const float* campos,
                    int W, int H, float tanx, float tany, const int* coeffs, const int* perband,
                    const int* cumsum, int* radii, float* xy, float* depths, float* cov3D, float* conic_op,
                    float* rgb, unsigned char* clamped, uint32_t* tiles_touched)
{
    const float fy = H / (2.0f * tany), fx = W / (2.0f * tanx); /* rasterizer_impl.cu:386-387 */
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    for (int i = 0; i < P; i++) {
        radii[i] = 0;
        tiles_touched[i] = 0;
        const float* p = means + 3 * i;
        float pv[3];
        xform4x3(vm, p, pv);
        if (pv[2] <= 0.2f) continue; /* auxiliary.h:139-159 in_frustum */
        float ph[4];
        xform4x4(pm, p, ph);
        float pw = 1.0f / (ph[3] + 0.0000001f);
        float pp[2] = {ph[0] * pw, ph[1] * pw};
        const float* c6;
        if (cov3D_precomp)
            c6 = cov3D_precomp + 6 * i;
        else {
            cov3d_from_scale_rot(scales + 3 * i, mod, rots + 4 * i, cov3D + 6 * i);
            c6 = cov3D + 6 * i;
        }
        const float opacity = 1.0f / (1.0f + expf(-opac_raw[i])); /* auxiliary.h:134-137 */
        float A[2][3], t[3], a, b, c;
        ewa_A(p, vm, fx, fy, tanx, tany, A, t, NULL, NULL);
        cov2d(A, c6, &a, &b, &c);
        float det = a * c - b * b;
        if (det == 0.0f) continue;
        float det_inv = 1.f / det;
        float conic[3] = {c * det_inv, -b * det_inv, a * det_inv};
        float mid = 0.5f * (a + c);
        float lam1 = mid + sqrtf(fmaxf_(0.1f, mid * mid - det));
        float lam2 = mid - sqrtf(fmaxf_(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf_(lam1, lam2)));
        float px = ndc2pix(pp[0], W), py = ndc2pix(pp[1], H);
        int rmin[2], rmax[2];
        get_rect(px, py, f2i(my_radius), gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (!colors_precomp) {
            if (coeffs) {
                int deg;
                int off = ragged_sh_offset(i, coeffs, perband, cumsum, &deg);
                sh_to_rgb(deg, shs + 3 * (size_t)off, p, campos, rgb + 3 * i, clamped + 3 * i);
            } else {
                sh_to_rgb(degs[i], shs + 3 * (size_t)M * i, p, campos, rgb + 3 * i, clamped + 3 * i);
            }
        }
        depths[i] = pv[2];
        radii[i] = f2i(my_radius);
        xy[2 * i] = px;
        xy[2 * i + 1] = py;
        conic_op[4 * i] = conic[0];
        conic_op[4 * i + 1] = conic[1];
        conic_op[4 * i + 2] = conic[2];
        conic_op[4 * i + 3] = opacity;
        tiles_touched[i] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    }
}

/* rasterizer_impl.cu:62-74 checkFrustum */
void orc_mark_visible(int P, const float* means, const float* vm, unsigned char* present)
{
    for (int i = 0; i < P; i++) {
        float pv[3];
        xform4x3(vm, means + 3 * i, pv);
        present[i] = pv[2] > 0.2f;
    }
}

/* rasterizer_impl.cu:43-58 getHigherMsb */
uint32_t orc_higher_msb(uint32_t n)
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

/* -------------------------------------------------------------------------
 * Binning: rasterizer_impl.cu:441 (inclusive scan), :78-119 duplicateWithKeys,
 * :465-473 stable radix sort on bits [0, 32+msb), :124-146 identifyTileRanges.
 * Call with keys==NULL to obtain R only.  Returns R.
 * ------------------------------------------------------------------------- */
/* rects == NULL: the reference's binning (rasterizer_impl.cu:78-146): every tile of getRect(radius).
 * rects != NULL ([P][4] = x0, y0, x1, y1 in tiles, exclusive maxima; the rows of the Gaussians with radii > 0 are used):
 * the same algorithm over GIVEN rects -- the product bins a Gaussian into the sub-rect of the reference's square that it
 * can reach with alpha >= 1/255 (csrc/gauss_math.h tighten_rect), the tests hand the product's rects in here and check
 * with orc_culled_tile_violations that every tile left out is one the reference's own per-pixel test rejects. */
int64_t orc_bin_rects(int P, int W, int H, const int* radii, const float* xy, const float* depths,
                      const uint32_t* tiles_touched, const uint16_t* rects, uint64_t* keys /*R*/,
                      uint32_t* point_list /*R*/, uint32_t* ranges /*2*Tn*/)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    int64_t R = 0;
    for (int i = 0; i < P; i++) {
        if (!rects)
            R += tiles_touched[i];
        else if (radii[i] > 0)
            R += (int64_t)(rects[4 * i + 2] - rects[4 * i]) * (int64_t)(rects[4 * i + 3] - rects[4 * i + 1]);
    }
    if (!keys) return R;
    uint64_t* ku = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(R ? R : 1));
    uint32_t* vu = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(R ? R : 1));
    int64_t off = 0;
    for (int i = 0; i < P; i++) {
        if (radii[i] > 0) {
            int rmin[2], rmax[2];
            if (rects) {
                rmin[0] = rects[4 * i];
                rmin[1] = rects[4 * i + 1];
                rmax[0] = rects[4 * i + 2];
                rmax[1] = rects[4 * i + 3];
            } else {
                get_rect(xy[2 * i], xy[2 * i + 1], radii[i], gx, gy, rmin, rmax);
            }
            uint32_t dbits;
            memcpy(&dbits, depths + i, 4);
            for (int y = rmin[1]; y < rmax[1]; y++)
                for (int x = rmin[0]; x < rmax[0]; x++) {
                    uint64_t key = (uint64_t)(y * gx + x);
                    key <<= 32;
                    key |= dbits;
                    ku[off] = key;
                    vu[off] = (uint32_t)i;
                    off++;
                }
        }
    }
    /* stable LSD radix sort, 16-bit digits, on the low 32+msb bits */
    int end_bit = 32 + (int)orc_higher_msb((uint32_t)(gx * gy));
    uint64_t* ka = ku;
    uint32_t* va = vu;
    uint64_t* kb = keys;
    uint32_t* vb = point_list;
    size_t* cnt = (size_t*)malloc(sizeof(size_t) * 65537);
    for (int shift = 0; shift < end_bit; shift += 16) {
        int nb = end_bit - shift < 16 ? end_bit - shift : 16;
        uint64_t mask = ((uint64_t)1 << nb) - 1;
        memset(cnt, 0, sizeof(size_t) * 65537);
        for (int64_t i = 0; i < R; i++) cnt[((ka[i] >> shift) & mask) + 1]++;
        for (int d = 0; d < 65536; d++) cnt[d + 1] += cnt[d];
        for (int64_t i = 0; i < R; i++) {
            size_t d = (size_t)((ka[i] >> shift) & mask);
            kb[cnt[d]] = ka[i];
            vb[cnt[d]] = va[i];
            cnt[d]++;
        }
        uint64_t* tk = ka;
        ka = kb;
        kb = tk;
        uint32_t* tv = va;
        va = vb;
        vb = tv;
    }
    if (ka != keys) {
        memcpy(keys, ka, sizeof(uint64_t) * (size_t)R);
        memcpy(point_list, va, sizeof(uint32_t) * (size_t)R);
    }
    free(cnt);
    free(ku);
    free(vu);
    memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)(gx * gy)); /* rasterizer_impl.cu:475 */
    for (int64_t i = 0; i < R; i++) {
        uint32_t cur = (uint32_t)(keys[i] >> 32);
        if (i == 0)
            ranges[2 * cur] = 0;
        else {
            uint32_t prev = (uint32_t)(keys[i - 1] >> 32);
            if (cur != prev) {
                ranges[2 * prev + 1] = (uint32_t)i;
                ranges[2 * cur] = (uint32_t)i;
            }
        }
        if (i == R - 1) ranges[2 * cur + 1] = (uint32_t)R;
    }
    return R;
}

/* -------------------------------------------------------------------------
 * Per-pixel forward blend: forward.cu:461-582 renderCUDA.
 * `colors` is rgb[P][3] (geomState.rgb or colors_precomp, rasterizer_impl.cu:486).
 * Counter mode (forward.cu:560-564): touched/transmittance non-NULL.
 * `ambig` (optional, N bytes): set to 1 for pixels where a decision of the
 * sequential blend (power>0, alpha<1/255, T(1-alpha)<1e-4) was within `ambig_rel`
 * of its threshold -- a different-but-valid exp() rounding could flip it; parity
 * tests exclude exactly those pixels from the 1e-5 bound.  The band widens with the
 * CONDITIONING of `power`: it is a sum of three products that cancel for an anisotropic
 * splat seen off its axes (|power| = 5 from terms of 470 each was met in the clustered
 * workload), so another valid fp32 evaluation of the same expression -- nvcc contracts
 * it into FMAs by default, this library pre-scales the conic -- moves power by a few
 * ulps of the LARGEST term, and alpha = opacity * exp(power) by that much relatively:
 * slack = 4 * 2^-23 * (|A dx^2| / 2 + |C dy^2| / 2 + |B dx dy|), added to `ambig_rel` for
 * the alpha test and to the absolute band of the power > 0 test (well-conditioned
 * entries: slack <= 3e-6, the band is what it was).
 * ------------------------------------------------------------------------- */
int64_t orc_bin(int P, int W, int H, const int* radii, const float* xy, const float* depths,
                const uint32_t* tiles_touched, uint64_t* keys /*R*/, uint32_t* point_list /*R*/,
                uint32_t* ranges /*2*Tn*/)
{
    return orc_bin_rects(P, W, H, radii, xy, depths, tiles_touched, NULL, keys, point_list, ranges);
}

/* Number of (tile, Gaussian, pixel) triples where the Gaussian's GIVEN rect leaves out a tile of the reference's rect
 * (auxiliary.h:46-56) although a pixel of that tile passes the reference's own per-pixel test (forward.cu:534-546:
 * power <= 0 and alpha = min(0.99, opacity * exp(power)) >= 1/255), plus the number of given rects that are not inside
 * the reference's.  0 <=> blending the shorter lists takes exactly the per-pixel decisions of the full lists.
 * Same fp32 expressions as orc_blend_fwd. */
int64_t orc_culled_tile_violations(int P, int W, int H, const int* radii, const float* xy, const float* conic_op,
                                   const uint16_t* rects, int64_t* tiles_left_out)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    int64_t bad = 0, left = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : bad, left)
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        int rmin[2], rmax[2];
        get_rect(xy[2 * i], xy[2 * i + 1], radii[i], gx, gy, rmin, rmax);
        const int x0 = rects[4 * i], y0 = rects[4 * i + 1], x1 = rects[4 * i + 2], y1 = rects[4 * i + 3];
        if (x1 > x0 && y1 > y0 && (x0 < rmin[0] || y0 < rmin[1] || x1 > rmax[0] || y1 > rmax[1])) bad++;
        const float* co = conic_op + 4 * i;
        for (int ty = rmin[1]; ty < rmax[1]; ty++)
            for (int tx = rmin[0]; tx < rmax[0]; tx++) {
                if (tx >= x0 && tx < x1 && ty >= y0 && ty < y1) continue; /* kept */
                left++;
                for (int py = ty * TILE; py < imin(H, (ty + 1) * TILE); py++)
                    for (int px = tx * TILE; px < imin(W, (tx + 1) * TILE); px++) {
                        const float dx = xy[2 * i] - (float)px, dy = xy[2 * i + 1] - (float)py;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float alpha = fminf_(0.99f, co[3] * expf(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        bad++;
                    }
            }
    }
    if (tiles_left_out) *tiles_left_out = left;
    return bad;
}

void orc_blend_fwd(int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* xy,
                   const float* colors, const float* conic_op, const float* bg, float* out_color,
                   float* final_T, uint32_t* n_contrib, int* touched, float* transmittance,
                   unsigned char* ambig, float ambig_rel)
{
    const int gx = (W + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 4) if (!touched)
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / TILE) * gx + (px / TILE);
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            const float pxf = (float)px, pyf = (float)py;
            float T = 1.0f, C[3] = {0, 0, 0};
            uint32_t contributor = 0, last = 0;
            unsigned char amb = 0;
            for (uint32_t k = r0; k < r1; k++) {
                contributor++;
                const uint32_t id = point_list[k];
                const float dx = xy[2 * id] - pxf, dy = xy[2 * id + 1] - pyf;
                const float* co = conic_op + 4 * id;
                const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                float slack = 0.0f;   /* what another association / FMA contraction of `power` may move it by */
                if (ambig)
                    slack = 4.0f * 1.1920929e-7f *
                            (0.5f * (fabsf(co[0] * dx * dx) + fabsf(co[2] * dy * dy)) + fabsf(co[1] * dx * dy));
                if (ambig && fabsf(power) <= 1e-6f + slack) amb = 1;
                if (power > 0.0f) continue;
                const float alpha = fminf_(0.99f, co[3] * expf(power));
                if (ambig && fabsf(alpha - 1.0f / 255.0f) <= (ambig_rel + slack) * (1.0f / 255.0f)) amb = 1;
                if (alpha < 1.0f / 255.0f) continue;
                const float test_T = T * (1 - alpha);
                if (ambig && fabsf(test_T - 0.0001f) <= ambig_rel * 0.0001f) amb = 1;
                if (test_T < 0.0001f) break; /* done = true */
                for (int ch = 0; ch < 3; ch++) C[ch] += colors[3 * id + ch] * alpha * T;
                if (touched) {
                    touched[id] += 1;
                    transmittance[id] += T;
                }
                T = test_T;
                last = contributor;
            }
            const size_t pix = (size_t)W * py + px;
            final_T[pix] = T;
            n_contrib[pix] = last;
            for (int ch = 0; ch < 3; ch++) out_color[(size_t)ch * H * W + pix] = C[ch] + T * bg[ch];
            if (ambig) ambig[pix] = amb;
        }
}

/* -------------------------------------------------------------------------
 * Per-pixel backward blend: backward.cu:437-595 renderCUDA.
 * The reference sums the per-(pixel,Gaussian) terms with unordered fp32 atomics;
 * the oracle evaluates every term in fp32 exactly as written and accumulates the
 * per-Gaussian sums in double (the order-independent value the atomics approximate).
 * acc layout per Gaussian (double[9]): dmean2D.x, dmean2D.y, dconic.x, dconic.y,
 * dconic.w, dopacity, dcolor.r, dcolor.g, dcolor.b.
 * ------------------------------------------------------------------------- */
void orc_blend_bwd(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* bg,
                   const float* xy, const float* conic_op, const float* colors, const float* final_T,
                   const uint32_t* n_contrib, const float* dL_dpix, float* dL_dmean2D /*P*3*/,
                   float* dL_dconic /*P*4*/, float* dL_dopacity /*P*/, float* dL_dcolor /*P*3*/)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    double* acc = (double*)calloc((size_t)P * 9, sizeof(double));
    const float ddelx_dx = (float)(0.5 * W), ddely_dy = (float)(0.5 * H);
    /* one tile per task; per-tile partial sums (double) are flushed once per list entry, so the
     * threads only meet in a few atomic adds and the result does not depend on the thread count
     * beyond double rounding */
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const int tile = ty * gx + tx;
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            if (r1 <= r0) continue;
            double* loc = (double*)calloc((size_t)(r1 - r0) * 9, sizeof(double));
            for (int py = ty * TILE; py < imin(H, (ty + 1) * TILE); py++)
                for (int px = tx * TILE; px < imin(W, (tx + 1) * TILE); px++) {
                    const size_t pix = (size_t)W * py + px;
                    const float pxf = (float)px, pyf = (float)py;
                    const float T_final = final_T[pix];
                    float T = T_final;
                    const uint32_t last_contributor = n_contrib[pix];
                    float accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0, dpx[3];
                    for (int ch = 0; ch < 3; ch++) dpx[ch] = dL_dpix[(size_t)ch * H * W + pix];
                    /* walk positions last_contributor-1 .. 0 of the tile list (backward.cu:524-526) */
                    for (int64_t pos = (int64_t)last_contributor - 1; pos >= 0; pos--) {
                        const uint32_t id = point_list[r0 + pos];
                        const float dx = xy[2 * id] - pxf, dy = xy[2 * id + 1] - pyf;
                        const float* co = conic_op + 4 * id;
                        const float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        const float G = expf(power);
                        const float alpha = fminf_(0.99f, co[3] * G);
                        if (alpha < 1.0f / 255.0f) continue;
                        T = T / (1.f - alpha);
                        const float dchannel_dcolor = alpha * T;
                        float dL_dalpha = 0.0f;
                        double* a = loc + 9 * (size_t)pos;
                        for (int ch = 0; ch < 3; ch++) {
                            const float c = colors[3 * id + ch];
                            accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                            last_color[ch] = c;
                            dL_dalpha += (c - accum_rec[ch]) * dpx[ch];
                            a[6 + ch] += (double)(dchannel_dcolor * dpx[ch]);
                        }
                        dL_dalpha *= T;
                        last_alpha = alpha;
                        float bg_dot = 0;
                        for (int ch = 0; ch < 3; ch++) bg_dot += bg[ch] * dpx[ch];
                        dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                        const float dL_dG = co[3] * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = -gdx * co[0] - gdy * co[1];
                        const float dG_ddely = -gdy * co[2] - gdx * co[1];
                        a[0] += (double)(dL_dG * dG_ddelx * ddelx_dx);
                        a[1] += (double)(dL_dG * dG_ddely * ddely_dy);
                        a[2] += (double)(-0.5f * gdx * dx * dL_dG);
                        a[3] += (double)(-0.5f * gdx * dy * dL_dG);
                        a[4] += (double)(-0.5f * gdy * dy * dL_dG);
                        a[5] += (double)(G * dL_dalpha);
                    }
                }
            for (uint32_t k = 0; k < r1 - r0; k++) {
                const double* l = loc + 9 * (size_t)k;
                double* a = acc + 9 * (size_t)point_list[r0 + k];
                for (int c = 0; c < 9; c++)
                    if (l[c] != 0.0) {
#pragma omp atomic
                        a[c] += l[c];
                    }
            }
            free(loc);
        }
    for (int i = 0; i < P; i++) {
        const double* a = acc + 9 * (size_t)i;
        dL_dmean2D[3 * i] = (float)a[0];
        dL_dmean2D[3 * i + 1] = (float)a[1];
        dL_dmean2D[3 * i + 2] = 0.f;
        dL_dconic[4 * i] = (float)a[2];
        dL_dconic[4 * i + 1] = (float)a[3];
        dL_dconic[4 * i + 2] = 0.f;
        dL_dconic[4 * i + 3] = (float)a[4];
        dL_dopacity[i] = (float)a[5];
        dL_dcolor[3 * i] = (float)a[6];
        dL_dcolor[3 * i + 1] = (float)a[7];
        dL_dcolor[3 * i + 2] = (float)a[8];
    }
    free(acc);
}

static inline float signf_(float v) { return (float)((v > 0.f) - (v < 0.f)); } /* glm::sign */

/* -------------------------------------------------------------------------
 * Per-Gaussian backward: backward.cu:177-307 computeCov2DCUDA followed by
 * backward.cu:379-434 preprocessCUDA (+ :20-172 SH, :311-374 cov3D).
 * In/out: dL_dopacity (sigmoid chain applied in place), dL_dcolor (input).
 * Outputs must be zero-initialised by the caller (rasterize_points.cu:259-267);
 * Gaussians with radii<=0 are skipped entirely.
 * n_visible: count(radii>0) used for the SH-sparsity multiplier
 * (rasterizer_impl.cu:549-566); lambda_sh==0 disables it.
 * ------------------------------------------------------------------------- */
void orc_preprocess_bwd(int P, int M, const int* degs, const float* means, const int* radii,
                        const float* shs, const unsigned char* clamped, const float* scales,
                        const float* rots, float mod, const float* cov3Ds /* precomp or geom */,
                        const float* vm, const float* pm, const float* campos, int W, int H, float tanx,
                        float tany, const float* dL_dmean2D /*P*3*/, const float* conic_op,
                        const float* dL_dconic /*P*4*/, float* dL_dmean3D, const float* dL_dcolor,
                        float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                        float* dL_dopacity, float lambda_sh)
{
    const float fy = H / (2.0f * tany), fx = W / (2.0f * tanx);
    float mult = 0.f;
    if (lambda_sh != 0.f) {
        int V = 0;
        for (int i = 0; i < P; i++) V += radii[i] > 0;
        mult = lambda_sh / (float)(V * 15 * 3);
    }
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        const float* p = means + 3 * i;
        const float* c6 = cov3Ds + 6 * i;
        /* ---- conic -> cov2D -> cov3D, A, t, mean (backward.cu:197-306) ---- */
        float A[2][3], t[3], xmul, ymul, a, b, c;
        ewa_A(p, vm, fx, fy, tanx, tany, A, t, &xmul, &ymul);
        cov2d(A, c6, &a, &b, &c);
        const float gA = dL_dconic[4 * i], gB = dL_dconic[4 * i + 1], gC = dL_dconic[4 * i + 3];
        const float denom = a * c - b * b;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float* dcov = dL_dcov3D + 6 * i;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * gA + 2 * b * c * gB + (denom - a * c) * gC);
            dL_dc = denom2inv * (-a * a * gC + 2 * a * b * gB + (denom - a * c) * gA);
            dL_db = denom2inv * 2 * (b * c * gA - (denom + 2 * b * b) * gB + a * b * gC);
            dcov[0] = (A[0][0] * A[0][0] * dL_da + A[0][0] * A[1][0] * dL_db + A[1][0] * A[1][0] * dL_dc);
            dcov[3] = (A[0][1] * A[0][1] * dL_da + A[0][1] * A[1][1] * dL_db + A[1][1] * A[1][1] * dL_dc);
            dcov[5] = (A[0][2] * A[0][2] * dL_da + A[0][2] * A[1][2] * dL_db + A[1][2] * A[1][2] * dL_dc);
            dcov[1] = 2 * A[0][0] * A[0][1] * dL_da + (A[0][0] * A[1][1] + A[0][1] * A[1][0]) * dL_db +
                      2 * A[1][0] * A[1][1] * dL_dc;
            dcov[2] = 2 * A[0][0] * A[0][2] * dL_da + (A[0][0] * A[1][2] + A[0][2] * A[1][0]) * dL_db +
                      2 * A[1][0] * A[1][2] * dL_dc;
            dcov[4] = 2 * A[0][2] * A[0][1] * dL_da + (A[0][1] * A[1][2] + A[0][2] * A[1][1]) * dL_db +
                      2 * A[1][1] * A[1][2] * dL_dc;
        } else {
            for (int k = 0; k < 6; k++) dcov[k] = 0;
        }
        const float S[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
        float dA[2][3];
        for (int j = 0; j < 3; j++) {
            float a0 = A[0][0] * S[j][0] + A[0][1] * S[j][1] + A[0][2] * S[j][2];
            float a1 = A[1][0] * S[j][0] + A[1][1] * S[j][1] + A[1][2] * S[j][2];
            dA[0][j] = 2 * a0 * dL_da + a1 * dL_db;
            dA[1][j] = 2 * a1 * dL_dc + a0 * dL_db;
        }
        const float dJ00 = RW(vm, 0, 0) * dA[0][0] + RW(vm, 0, 1) * dA[0][1] + RW(vm, 0, 2) * dA[0][2];
        const float dJ02 = RW(vm, 2, 0) * dA[0][0] + RW(vm, 2, 1) * dA[0][1] + RW(vm, 2, 2) * dA[0][2];
        const float dJ11 = RW(vm, 1, 0) * dA[1][0] + RW(vm, 1, 1) * dA[1][1] + RW(vm, 1, 2) * dA[1][2];
        const float dJ12 = RW(vm, 2, 0) * dA[1][0] + RW(vm, 2, 1) * dA[1][1] + RW(vm, 2, 2) * dA[1][2];
        const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const float dtx = xmul * -fx * tz2 * dJ02;
        const float dty = ymul * -fy * tz2 * dJ12;
        const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * t[0]) * tz3 * dJ02 +
                          (2 * fy * t[1]) * tz3 * dJ12;
        float dmean[3]; /* transformVec4x3Transpose: assigns (backward.cu:301-306) */
        dmean[0] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
        dmean[1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
        dmean[2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;

        /* ---- mean2D -> mean3D through the perspective divide (backward.cu:406-423) ---- */
        float mh[4];
        xform4x4(pm, p, mh);
        const float mw = 1.0f / (mh[3] + 0.0000001f);
        const float mul1 = (pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12]) * mw * mw;
        const float mul2 = (pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13]) * mw * mw;
        const float g2x = dL_dmean2D[3 * i], g2y = dL_dmean2D[3 * i + 1];
        dmean[0] += (pm[0] * mw - pm[3] * mul1) * g2x + (pm[1] * mw - pm[3] * mul2) * g2y;
        dmean[1] += (pm[4] * mw - pm[7] * mul1) * g2x + (pm[5] * mw - pm[7] * mul2) * g2y;
        dmean[2] += (pm[8] * mw - pm[11] * mul1) * g2x + (pm[9] * mw - pm[11] * mul2) * g2y;

        /* ---- colour -> SH and view direction (backward.cu:20-172) ---- */
        if (shs) {
            const float* sh = shs + 3 * (size_t)M * i;
            float* dsh = dL_dsh + 3 * (size_t)M * i;
            float dorig[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
            float len = sqrtf(dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2]);
            const float x = dorig[0] / len, y = dorig[1] / len, z = dorig[2] / len;
            float dRGB[3];
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcolor[3 * i + ch] * (clamped[3 * i + ch] ? 0.f : 1.f);
            const int deg = degs[i];
            const int K = (deg + 1) * (deg + 1);
            float Y[16];
            sh_basis(deg, x, y, z, Y);
            for (int k = 0; k < K; k++)
                for (int ch = 0; ch < 3; ch++) {
                    float g = Y[k] * dRGB[ch];
                    if (k >= 1 && mult != 0.f) g = g + mult * signf_(sh[3 * k + ch]);
                    dsh[3 * k + ch] = g;
                }
            float dRdx[3] = {0, 0, 0}, dRdy[3] = {0, 0, 0}, dRdz[3] = {0, 0, 0};
#define SHK(k) (sh[3 * (k) + ch])
            for (int ch = 0; ch < 3; ch++) {
                if (deg > 0) {
                    dRdx[ch] = -SH_C1 * SHK(3);
                    dRdy[ch] = -SH_C1 * SHK(1);
                    dRdz[ch] = SH_C1 * SHK(2);
                    if (deg > 1) {
                        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                        dRdx[ch] += SH_C2[0] * y * SHK(4) + SH_C2[2] * 2.f * -x * SHK(6) + SH_C2[3] * z * SHK(7) +
                                    SH_C2[4] * 2.f * x * SHK(8);
                        dRdy[ch] += SH_C2[0] * x * SHK(4) + SH_C2[1] * z * SHK(5) + SH_C2[2] * 2.f * -y * SHK(6) +
                                    SH_C2[4] * 2.f * -y * SHK(8);
                        dRdz[ch] += SH_C2[1] * y * SHK(5) + SH_C2[2] * 2.f * 2.f * z * SHK(6) + SH_C2[3] * x * SHK(7);
                        if (deg > 2) {
                            dRdx[ch] += (SH_C3[0] * SHK(9) * 3.f * 2.f * xy + SH_C3[1] * SHK(10) * yz +
                                         SH_C3[2] * SHK(11) * -2.f * xy + SH_C3[3] * SHK(12) * -3.f * 2.f * xz +
                                         SH_C3[4] * SHK(13) * (-3.f * xx + 4.f * zz - yy) +
                                         SH_C3[5] * SHK(14) * 2.f * xz + SH_C3[6] * SHK(15) * 3.f * (xx - yy));
                            dRdy[ch] += (SH_C3[0] * SHK(9) * 3.f * (xx - yy) + SH_C3[1] * SHK(10) * xz +
                                         SH_C3[2] * SHK(11) * (-3.f * yy + 4.f * zz - xx) +
                                         SH_C3[3] * SHK(12) * -3.f * 2.f * yz + SH_C3[4] * SHK(13) * -2.f * xy +
                                         SH_C3[5] * SHK(14) * -2.f * yz + SH_C3[6] * SHK(15) * -3.f * 2.f * xy);
                            dRdz[ch] += (SH_C3[1] * SHK(10) * xy + SH_C3[2] * SHK(11) * 4.f * 2.f * yz +
                                         SH_C3[3] * SHK(12) * 3.f * (2.f * zz - xx - yy) +
                                         SH_C3[4] * SHK(13) * 4.f * 2.f * xz + SH_C3[5] * SHK(14) * (xx - yy));
                        }
                    }
                }
            }
#undef SHK
            const float ddir[3] = {dRdx[0] * dRGB[0] + dRdx[1] * dRGB[1] + dRdx[2] * dRGB[2],
                                   dRdy[0] * dRGB[0] + dRdy[1] * dRGB[1] + dRdy[2] * dRGB[2],
                                   dRdz[0] * dRGB[0] + dRdz[1] * dRGB[1] + dRdz[2] * dRGB[2]};
            /* auxiliary.h:107-117 dnormvdv */
            const float vx = dorig[0], vy = dorig[1], vz = dorig[2];
            const float sum2 = vx * vx + vy * vy + vz * vz;
            const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
            dmean[0] += ((+sum2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * invsum32;
            dmean[1] += (-vx * vy * ddir[0] + (sum2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * invsum32;
            dmean[2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (sum2 - vz * vz) * ddir[2]) * invsum32;
        }
        dL_dmean3D[3 * i] = dmean[0];
        dL_dmean3D[3 * i + 1] = dmean[1];
        dL_dmean3D[3 * i + 2] = dmean[2];

        /* ---- cov3D -> scale, rotation (backward.cu:311-374) ---- */
        if (scales) {
            const float* q = rots + 4 * i;
            const float r = q[0], qx = q[1], qy = q[2], qz = q[3];
            float R[3][3];
            quat_to_R(q, R);
            const float s[3] = {mod * scales[3 * i], mod * scales[3 * i + 1], mod * scales[3 * i + 2]};
            /* M(k,j) = s_k R(j,k);  dL_dSigma symmetric with halved off-diagonals */
            const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                    {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                    {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
            /* dL_dM = 2 * M * dL_dSigma : dM(k,j) = sum_m (2*M(k,m)) * dS(m,j) */
            float dM[3][3];
            for (int k = 0; k < 3; k++)
                for (int j = 0; j < 3; j++) {
                    float m0 = 2.0f * (s[k] * R[0][k]), m1 = 2.0f * (s[k] * R[1][k]), m2 = 2.0f * (s[k] * R[2][k]);
                    dM[k][j] = m0 * dS[0][j] + m1 * dS[1][j] + m2 * dS[2][j];
                }
            /* dL_dscale_k = dot(R(:,k), dM(k,:))  (backward.cu:355-358) */
            for (int k = 0; k < 3; k++)
                dL_dscale[3 * i + k] = R[0][k] * dM[k][0] + R[1][k] * dM[k][1] + R[2][k] * dM[k][2];
            /* dL_dMt[k][j] (glm col k,row j) = dM(k,j); scaled by s_k (backward.cu:360-362) */
            float D[3][3];
            for (int k = 0; k < 3; k++)
                for (int j = 0; j < 3; j++) D[k][j] = dM[k][j] * s[k];
            float* dq = dL_drot + 4 * i;
            dq[0] = 2 * qz * (D[0][1] - D[1][0]) + 2 * qy * (D[2][0] - D[0][2]) + 2 * qx * (D[1][2] - D[2][1]);
            dq[1] = 2 * qy * (D[1][0] + D[0][1]) + 2 * qz * (D[2][0] + D[0][2]) + 2 * r * (D[1][2] - D[2][1]) -
                    4 * qx * (D[2][2] + D[1][1]);
            dq[2] = 2 * qx * (D[1][0] + D[0][1]) + 2 * r * (D[2][0] - D[0][2]) + 2 * qz * (D[1][2] + D[2][1]) -
                    4 * qy * (D[2][2] + D[0][0]);
            dq[3] = 2 * r * (D[0][1] - D[1][0]) + 2 * qx * (D[2][0] + D[0][2]) + 2 * qy * (D[1][2] + D[2][1]) -
                    4 * qz * (D[1][1] + D[0][0]);
        }
        /* sigmoid chain, evaluated in double like the reference literal 1.0 (backward.cu:433) */
        const float o = conic_op[4 * i + 3];
        dL_dopacity[i] = (float)((double)dL_dopacity[i] * ((double)o * (1.0 - (double)o)));
    }
}

/* -------------------------------------------------------------------------
 * SH-band culling statistics -- next-tier operator `calculate_colours_variance`
 * (DGR/reduced_3dgs.cu:41-203).  Per-camera, per-Gaussian part:
 * DGR/reduced_3dgs/sh_culling.cu:6-57 computeColorFromSH: colour truncated after band k for
 * k = 0..deg (note: +0.5 is added right after the DC term there, unlike forward.cu:151), slots above the
 * Gaussian's own degree stay 0.  `stride` = max_sh_deg + 1 slots of 3 floats (the reference hard-codes 4).
 * ------------------------------------------------------------------------- */
void orc_truncated_colours(int P, int M, int stride, const int* degs, const float* means, const float* campos,
                           const float* shs, float* colours /* [P][stride][3], pre-zeroed */)
{
    for (int i = 0; i < P; i++) {
        const float* p = means + 3 * i;
        const float* sh = shs + 3 * (size_t)M * i;
        float d[3] = {p[0] - campos[0], p[1] - campos[1], p[2] - campos[2]};
        float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float x = d[0] / len, y = d[1] / len, z = d[2] / len;
        int deg = degs[i];
        if (deg > stride - 1) deg = stride - 1;
        float Y[16];
        sh_basis(deg, x, y, z, Y);
        for (int ch = 0; ch < 3; ch++) {
            float r = Y[0] * sh[ch];
            r += 0.5f;
            colours[((size_t)i * stride + 0) * 3 + ch] = fmaxf_(r, 0.0f);
            int k = 1;
            for (int band = 1; band <= deg; band++) {
                for (; k < (band + 1) * (band + 1); k++) r = r + Y[k] * sh[3 * k + ch];
                colours[((size_t)i * stride + band) * 3 + ch] = fmaxf_(r, 0.0f);
            }
        }
    }
}
