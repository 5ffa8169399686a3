// gauss_math.h -- per-Gaussian arithmetic of the rasterizer hot path (gfx950 build).
//
// Everything here is straight-line fp32 math on ONE Gaussian, written as
// __host__ __device__ inline functions so that (a) the HIP kernels in
// preprocess.hip / preprocess_bwd.hip call it per lane, and (b) the CPU test shim
// tests/hostcheck/hostcheck.hip can run exactly the same source on the host against
// the oracle before any GPU time is spent.  The translation units that include this
// header are compiled with -ffp-contract=off and correctly rounded fp32 divide/sqrt,
// so integer outputs (radii, tile rects, tiles_touched) are reproducible bit-for-bit.
//
// Semantics follow (file:line relative to /root/reference/submodules/diff-gaussian-rasterization):
//   cuda_rasterizer/forward.cu:353-456   preprocessCUDA           -> preprocess_one()
//   cuda_rasterizer/forward.cu:105-159   computeColorFromSH       -> sh_to_rgb()
//   cuda_rasterizer/backward.cu:177-307  computeCov2DCUDA         -> cov2d_backward()
//   cuda_rasterizer/backward.cu:379-434  preprocessCUDA (bwd)     -> project_backward(), sh_backward(),
//   cuda_rasterizer/backward.cu:311-374  computeCov3D (bwd)          cov3d_backward()
// Matrix convention (auxiliary.h:58-77): flat m[4*c + r] = entry (row r, col c).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace r3 {

#define R3_HD __host__ __device__ __forceinline__

constexpr int kTile = 16;  // config.h BLOCK_X = BLOCK_Y

// SH basis constants, auxiliary.h:22-39
#define R3_SH_C0 0.28209479177387814f
#define R3_SH_C1 0.4886025119029199f
#define R3_SH_C2_0 1.0925484305920792f
#define R3_SH_C2_1 -1.0925484305920792f
#define R3_SH_C2_2 0.31539156525252005f
#define R3_SH_C2_3 -1.0925484305920792f
#define R3_SH_C2_4 0.5462742152960396f
#define R3_SH_C3_0 -0.5900435899266435f
#define R3_SH_C3_1 2.890611442640554f
#define R3_SH_C3_2 -0.4570457994644658f
#define R3_SH_C3_3 0.3731763325901154f
#define R3_SH_C3_4 -0.4570457994644658f
#define R3_SH_C3_5 1.445305721320277f
#define R3_SH_C3_6 -0.5900435899266435f

// 48-byte per-Gaussian record consumed by the blend kernels (one gather per tile-list entry).
struct alignas(16) GRec {
    float x, y, cA, cB;       // pixel-space mean, conic (A,B)
    float cC, op, r, g;       // conic C, activated opacity, colour r,g
    float b;                  // colour b
    uint32_t rect_min;        // first tile of the rect: x | y << 16
    uint32_t width_clamp;     // rect width in tiles | clamp bits << 16 (bit ch: colour channel clamped at 0,
                              // forward.cu:155-157)
    uint32_t pair_start;      // index of this Gaussian's first (tile, Gaussian) pair in emission order; written by
                              // the pair-emission kernel.  Its pairs are [pair_start, pair_start + tiles_touched),
                              // row-major over the rect -- where the backward parks per-pair gradients.
};

struct Camera {
    float view[16];
    float proj[16];
    float campos[3];
    float tan_fovx, tan_fovy, focal_x, focal_y;
    int W, H, gx, gy;
    float scale_modifier;
};

R3_HD float fmin_(float a, float b) { return a < b ? a : b; }
R3_HD float fmax_(float a, float b) { return a > b ? a : b; }
R3_HD int imin_(int a, int b) { return a < b ? a : b; }
R3_HD int imax_(int a, int b) { return a > b ? a : b; }

// float -> int truncation, saturating (the behaviour of v_cvt_i32_f32)
R3_HD int f2i(float v)
{
    if (!(v == v)) return 0;
    if (v >= 2147483520.0f) return 2147483647;
    if (v <= -2147483648.0f) return -2147483647 - 1;
    return (int)v;
}

R3_HD void xform4x3(const float* m, float px, float py, float pz, float* o)
{
    o[0] = m[0] * px + m[4] * py + m[8] * pz + m[12];
    o[1] = m[1] * px + m[5] * py + m[9] * pz + m[13];
    o[2] = m[2] * px + m[6] * py + m[10] * pz + m[14];
}
R3_HD void xform4x4(const float* m, float px, float py, float pz, float* o)
{
    o[0] = m[0] * px + m[4] * py + m[8] * pz + m[12];
    o[1] = m[1] * px + m[5] * py + m[9] * pz + m[13];
    o[2] = m[2] * px + m[6] * py + m[10] * pz + m[14];
    o[3] = m[3] * px + m[7] * py + m[11] * pz + m[15];
}

// auxiliary.h:41-44 -- the reference evaluates this in double (1.0 / 0.5 literals)
R3_HD float ndc2pix(float v, int S) { return (float)((((double)v + 1.0) * S - 1.0) * 0.5); }

// auxiliary.h:46-56 getRect
R3_HD void tile_rect(float px, float py, int radius, int gx, int gy, int* rmin, int* rmax)
{
    const float r = (float)radius;
    rmin[0] = imin_(gx, imax_(0, f2i((px - r) / (float)kTile)));
    rmin[1] = imin_(gy, imax_(0, f2i((py - r) / (float)kTile)));
    rmax[0] = imin_(gx, imax_(0, f2i((px + r + (float)(kTile - 1)) / (float)kTile)));
    rmax[1] = imin_(gy, imax_(0, f2i((py + r + (float)(kTile - 1)) / (float)kTile)));
}

// rotation matrix of a unit quaternion q = (r, x, y, z); row-major R[3*i + j]
R3_HD void quat_to_R(const float* q, float* R)
{
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1.f - 2.f * (y * y + z * z);
    R[1] = 2.f * (x * y - r * z);
    R[2] = 2.f * (x * z + r * y);
    R[3] = 2.f * (x * y + r * z);
    R[4] = 1.f - 2.f * (x * x + z * z);
    R[5] = 2.f * (y * z - r * x);
    R[6] = 2.f * (x * z - r * y);
    R[7] = 2.f * (y * z + r * x);
    R[8] = 1.f - 2.f * (x * x + y * y);
}

// forward.cu:207-241: Sigma = R diag(s*mod)^2 R^T as 6 floats (00,01,02,11,12,22).
// Sigma(a,b) = sum_k (s_k R(a,k)) (s_k R(b,k)), k ascending.
R3_HD void cov3d_from_scale_rot(const float* scale, float mod, const float* q, float* c6)
{
    float R[9];
    quat_to_R(q, R);
    const float s0 = mod * scale[0], s1 = mod * scale[1], s2 = mod * scale[2];
    float M[9];  // M[3*k + j] = s_k * R(j,k)
    for (int j = 0; j < 3; j++) {
        M[0 + j] = s0 * R[3 * j + 0];
        M[3 + j] = s1 * R[3 * j + 1];
        M[6 + j] = s2 * R[3 * j + 2];
    }
#define R3_SIG(a, b) (M[0 + a] * M[0 + b] + M[3 + a] * M[3 + b] + M[6 + a] * M[6 + b])
    c6[0] = R3_SIG(0, 0);
    c6[1] = R3_SIG(0, 1);
    c6[2] = R3_SIG(0, 2);
    c6[3] = R3_SIG(1, 1);
    c6[4] = R3_SIG(1, 2);
    c6[5] = R3_SIG(2, 2);
#undef R3_SIG
}

// A = J * Rw (2x3 as A[3*i + j]) with the 1.3*tanfov clamp (forward.cu:168-187, backward.cu:199-225).
// Rw(i,j) = view[4*j + i].
R3_HD void ewa_A(const Camera& cam, float mx, float my, float mz, float* A, float* t, float* xmul, float* ymul)
{
    const float* vm = cam.view;
    xform4x3(vm, mx, my, mz, t);
    const float limx = 1.3f * cam.tan_fovx, limy = 1.3f * cam.tan_fovy;
    const float txtz = t[0] / t[2], tytz = t[1] / t[2];
    t[0] = fmin_(limx, fmax_(-limx, txtz)) * t[2];
    t[1] = fmin_(limy, fmax_(-limy, tytz)) * t[2];
    *xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    *ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float J00 = cam.focal_x / t[2], J02 = -(cam.focal_x * t[0]) / (t[2] * t[2]);
    const float J11 = cam.focal_y / t[2], J12 = -(cam.focal_y * t[1]) / (t[2] * t[2]);
    for (int j = 0; j < 3; j++) {
        const float r0 = vm[4 * j + 0], r1 = vm[4 * j + 1], r2 = vm[4 * j + 2];
        A[0 + j] = r0 * J00 + r1 * 0.0f + r2 * J02;
        A[3 + j] = r0 * 0.0f + r1 * J11 + r2 * J12;
    }
}

// forward.cu:189-201: cov2D = (A Sigma) A^T, +0.3 low-pass on the diagonal
R3_HD void cov2d(const float* A, const float* c6, float* a, float* b, float* c)
{
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    float AS[6];
    for (int i = 0; i < 2; i++)
        for (int k = 0; k < 3; k++)
            AS[3 * i + k] = A[3 * i + 0] * S[0 + k] + A[3 * i + 1] * S[3 + k] + A[3 * i + 2] * S[6 + k];
    *a = (AS[0] * A[0] + AS[1] * A[1] + AS[2] * A[2]) + 0.3f;
    *b = AS[3] * A[0] + AS[4] * A[1] + AS[5] * A[2];
    *c = (AS[3] * A[3] + AS[4] * A[4] + AS[5] * A[5]) + 0.3f;
}

// real SH basis up to `deg` for unit direction (x,y,z); forward.cu:115-148 association
R3_HD void sh_basis(int deg, float x, float y, float z, float* Y)
{
    Y[0] = R3_SH_C0;
    if (deg > 0) {
        Y[1] = -(R3_SH_C1 * y);
        Y[2] = R3_SH_C1 * z;
        Y[3] = -(R3_SH_C1 * x);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            Y[4] = R3_SH_C2_0 * xy;
            Y[5] = R3_SH_C2_1 * yz;
            Y[6] = R3_SH_C2_2 * (2.0f * zz - xx - yy);
            Y[7] = R3_SH_C2_3 * xz;
            Y[8] = R3_SH_C2_4 * (xx - yy);
            if (deg > 2) {
                Y[9] = R3_SH_C3_0 * y * (3.0f * xx - yy);
                Y[10] = R3_SH_C3_1 * xy * z;
                Y[11] = R3_SH_C3_2 * y * (4.0f * zz - xx - yy);
                Y[12] = R3_SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                Y[13] = R3_SH_C3_4 * x * (4.0f * zz - xx - yy);
                Y[14] = R3_SH_C3_5 * z * (xx - yy);
                Y[15] = R3_SH_C3_6 * x * (xx - 3.0f * yy);
            }
        }
    }
}

// Accessor for one Gaussian's SH row [K][3] living anywhere (global row, or an LDS staging
// area with a bank-skewed index).  `at(e)` returns float e of the row (e = 3*k + ch).
struct ShRowPlain {
    const float* p;
    R3_HD float at(int e) const { return p[e]; }
};

// forward.cu:105-159: colour = sum_k Y_k(dir) sh_k + 0.5, clamp >= 0, record clamps
template <class ShRow>
R3_HD void sh_to_rgb(int deg, const ShRow& sh, float mx, float my, float mz, const float* campos, float* rgb,
                     uint32_t* clamp_bits)
{
    float dx = mx - campos[0], dy = my - campos[1], dz = mz - campos[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    dx = dx / len;
    dy = dy / len;
    dz = dz / len;
    float Y[16];
    sh_basis(deg, dx, dy, dz, Y);
    const int K = (deg + 1) * (deg + 1);
    uint32_t bits = 0;
    for (int ch = 0; ch < 3; ch++) {
        float r = Y[0] * sh.at(ch);
        for (int k = 1; k < K; k++) r = r + Y[k] * sh.at(3 * k + ch);
        r += 0.5f;
        if (r < 0) bits |= 1u << ch;
        rgb[ch] = fmax_(r, 0.0f);
    }
    *clamp_bits = bits;
}

// reduced_3dgs/sh_culling.cu:6-57: colour truncated after band k, k = 0..min(deg, nslots-1); the +0.5 is added
// right after the DC term there.  out[3*k + ch]; slots above the Gaussian's own degree are left untouched
// (the caller zero-fills).
template <class ShRow>
R3_HD void sh_truncated_colours(int deg, int nslots, const ShRow& sh, float mx, float my, float mz, const float* campos,
                                float* out)
{
    float dx = mx - campos[0], dy = my - campos[1], dz = mz - campos[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    dx = dx / len;
    dy = dy / len;
    dz = dz / len;
    if (deg > nslots - 1) deg = nslots - 1;
    float Y[16];
    sh_basis(deg, dx, dy, dz, Y);
    for (int ch = 0; ch < 3; ch++) {
        float r = Y[0] * sh.at(ch);
        r += 0.5f;
        out[ch] = fmax_(r, 0.0f);
        int k = 1;
        for (int band = 1; band <= deg; band++) {
            for (; k < (band + 1) * (band + 1); k++) r = r + Y[k] * sh.at(3 * k + ch);
            out[3 * band + ch] = fmax_(r, 0.0f);
        }
    }
}

struct PreOut {
    int radius;        // 0 => culled (the reference's radii output)
    int rmin[2], rmax[2];
    float px, py, depth;
    float conic[3], opacity;
    uint32_t tiles;      // tiles of the rect the Gaussian is binned into (0: visible, but it can reach no pixel)
    uint32_t tiles_ref;  // tiles of the reference's rect (auxiliary.h:46-56): what num_rendered counts
};

// Opacity-aware tile rect.  The reference bins a Gaussian into every tile of the bounding SQUARE of 3 sigma of its major
// axis (auxiliary.h:46-56) and lets each pixel find out that it is not reached (alpha < 1/255 -> skip, forward.cu:540-546).
// alpha >= 1/255  <=>  q(d) = 0.5 d^T conic d <= tau = ln(255 opacity), an ellipse whose axis-aligned bounding box has
// the half extents sqrt(2 tau cov_xx), sqrt(2 tau cov_yy) (cov = conic^-1, the +0.3 low-pass included): tiles of the
// reference's rect outside that box hold no pixel the Gaussian can touch, so leaving them out of the lists changes no
// pixel decision, i.e. neither the image nor any gradient -- only the lists get shorter (35 % fewer pairs at the
// benchmark shape).
// Conservative against the REFERENCE's fp32 evaluation of q, not only against the exact one: the per-pixel
// test forms q from three products that cancel for an anisotropic splat seen at an angle, sum |terms| <= 4 kappa q with
// kappa = cov_xx cov_yy / det >= 1 (= AC / (AC - B^2) of the conic; 1 for an axis-aligned splat), and the conic itself
// carries the rounding of det = ac - b^2 (relative 2 eps kappa, a common factor of q).  A pixel with exact q > tau can
// therefore pass the fp32 test as long as q (1 - r) <= tau, r ~ 26 eps kappa = 1.6e-6 kappa (eps = 2^-24).  tau is taken
// 0.2 % + 2e-3 larger as before (the margin of the blend kernels' region pre-test, blend_math.h) and then divided by
// 1 - 4e-6 kappa; kappa > 6e4 (r > 0.25: a needle whose far pixels the fp32 test decides by rounding) keeps the
// reference's rect.  The box is 0.2 % + half a pixel wider; a conic that is not positive definite, or any NaN, keeps the
// reference's rect.  tests/test_gpu_parity.py checks every left-out (tile, Gaussian) pair pixel by pixel with the oracle,
// tests/test_hostcheck.py does the same on the CPU with needle scenes (median radius 500 px).
constexpr float kCancelMargin = 4e-6f;   // per unit of kappa; shared with blend_math.h region_may_contribute

R3_HD void tighten_rect(float px, float py, float a, float b, float c, float opacity, int* rmin, int* rmax)
{
    const float det = a * c - b * b;
    if (!(a > 0.f) || !(c > 0.f) || !(det > 0.f)) return;
    const float r = kCancelMargin * ((a * c) / det);
    if (!(r < 0.25f)) return;
    float tau = logf(255.0f * opacity);
    tau = tau + 2e-3f * fabsf(tau) + 2e-3f;
    if (!(tau == tau)) return;
    if (!(tau > 0.f)) {   // opacity below 1/255 (with margin): alpha <= opacity < 1/255 everywhere (power <= 0 where blended)
        rmax[0] = rmin[0];
        rmax[1] = rmin[1];
        return;
    }
    tau = tau / (1.0f - r);
    const float hx = sqrtf(2.f * tau * a) * 1.002f + 0.5f, hy = sqrtf(2.f * tau * c) * 1.002f + 0.5f;
    if (!(hx == hx) || !(hy == hy)) return;
    // pixels are at integer coordinates: those with |px - x| <= hx lie in tiles floor((x - hx) / 16) .. floor((x + hx) / 16)
    const float lo_x = floorf((px - hx) / (float)kTile), hi_x = floorf((px + hx) / (float)kTile) + 1.f;
    const float lo_y = floorf((py - hy) / (float)kTile), hi_y = floorf((py + hy) / (float)kTile) + 1.f;
    rmin[0] = imax_(rmin[0], f2i(lo_x));
    rmin[1] = imax_(rmin[1], f2i(lo_y));
    rmax[0] = imin_(rmax[0], f2i(hi_x));
    rmax[1] = imin_(rmax[1], f2i(hi_y));
    if (rmax[0] < rmin[0]) rmax[0] = rmin[0];
    if (rmax[1] < rmin[1]) rmax[1] = rmin[1];
}

// forward.cu:353-456 without the colour step.  cov6 = precomputed covariance or nullptr.
R3_HD void preprocess_one(const Camera& cam, float mx, float my, float mz, const float* scale, const float* rot,
                          const float* cov6_precomp, float opacity_raw, PreOut* o, bool tight = false)
{
    o->radius = 0;
    o->tiles = 0;
    o->tiles_ref = 0;
    float pv[3];
    xform4x3(cam.view, mx, my, mz, pv);
    if (pv[2] <= 0.2f) return;  // auxiliary.h:139-159
    float ph[4];
    xform4x4(cam.proj, mx, my, mz, ph);
    const float pw = 1.0f / (ph[3] + 0.0000001f);
    const float ppx = ph[0] * pw, ppy = ph[1] * pw;
    float c6[6];
    if (cov6_precomp) {
        for (int k = 0; k < 6; k++) c6[k] = cov6_precomp[k];
    } else {
        cov3d_from_scale_rot(scale, cam.scale_modifier, rot, c6);
    }
    float A[6], t[3], xm, ym, a, b, c;
    ewa_A(cam, mx, my, mz, A, t, &xm, &ym);
    cov2d(A, c6, &a, &b, &c);
    const float det = a * c - b * b;
    if (det == 0.0f) return;
    const float det_inv = 1.f / det;
    const float mid = 0.5f * (a + c);
    const float lam1 = mid + sqrtf(fmax_(0.1f, mid * mid - det));
    const float lam2 = mid - sqrtf(fmax_(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmax_(lam1, lam2)));
    const float px = ndc2pix(ppx, cam.W), py = ndc2pix(ppy, cam.H);
    const int rad = f2i(my_radius);
    tile_rect(px, py, rad, cam.gx, cam.gy, o->rmin, o->rmax);
    const int area = (o->rmax[0] - o->rmin[0]) * (o->rmax[1] - o->rmin[1]);
    if (area == 0) return;
    o->radius = rad;
    o->tiles_ref = (uint32_t)area;
    o->px = px;
    o->py = py;
    o->depth = pv[2];
    o->conic[0] = c * det_inv;
    o->conic[1] = -b * det_inv;
    o->conic[2] = a * det_inv;
    o->opacity = 1.0f / (1.0f + expf(-opacity_raw));  // auxiliary.h:134-137
    if (tight) tighten_rect(px, py, a, b, c, o->opacity, o->rmin, o->rmax);
    o->tiles = (uint32_t)((o->rmax[0] - o->rmin[0]) * (o->rmax[1] - o->rmin[1]));
}

// ---------------------------------------------------------------------------------------------
// backward pieces
// ---------------------------------------------------------------------------------------------

// backward.cu:177-307: (dL/dconic A,B,C) -> dL/dcov3D[6] and the covariance part of dL/dmean.
R3_HD void cov2d_backward(const Camera& cam, float mx, float my, float mz, const float* c6, float gA, float gB,
                          float gC, float* dcov6, float* dmean)
{
    float A[6], t[3], xmul, ymul, a, b, c;
    ewa_A(cam, mx, my, mz, A, t, &xmul, &ymul);
    cov2d(A, c6, &a, &b, &c);
    const float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * gA + 2 * b * c * gB + (denom - a * c) * gC);
        dL_dc = denom2inv * (-a * a * gC + 2 * a * b * gB + (denom - a * c) * gA);
        dL_db = denom2inv * 2 * (b * c * gA - (denom + 2 * b * b) * gB + a * b * gC);
        dcov6[0] = (A[0] * A[0] * dL_da + A[0] * A[3] * dL_db + A[3] * A[3] * dL_dc);
        dcov6[3] = (A[1] * A[1] * dL_da + A[1] * A[4] * dL_db + A[4] * A[4] * dL_dc);
        dcov6[5] = (A[2] * A[2] * dL_da + A[2] * A[5] * dL_db + A[5] * A[5] * dL_dc);
        dcov6[1] = 2 * A[0] * A[1] * dL_da + (A[0] * A[4] + A[1] * A[3]) * dL_db + 2 * A[3] * A[4] * dL_dc;
        dcov6[2] = 2 * A[0] * A[2] * dL_da + (A[0] * A[5] + A[2] * A[3]) * dL_db + 2 * A[3] * A[5] * dL_dc;
        dcov6[4] = 2 * A[2] * A[1] * dL_da + (A[1] * A[5] + A[2] * A[4]) * dL_db + 2 * A[4] * A[5] * dL_dc;
    } else {
        for (int k = 0; k < 6; k++) dcov6[k] = 0;
    }
    const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    float dA[6];
    for (int j = 0; j < 3; j++) {
        const float a0 = A[0] * S[3 * j + 0] + A[1] * S[3 * j + 1] + A[2] * S[3 * j + 2];
        const float a1 = A[3] * S[3 * j + 0] + A[4] * S[3 * j + 1] + A[5] * S[3 * j + 2];
        dA[0 + j] = 2 * a0 * dL_da + a1 * dL_db;
        dA[3 + j] = 2 * a1 * dL_dc + a0 * dL_db;
    }
    const float* vm = cam.view;  // Rw(i,j) = vm[4*j + i]
    const float dJ00 = vm[0] * dA[0] + vm[4] * dA[1] + vm[8] * dA[2];
    const float dJ02 = vm[2] * dA[0] + vm[6] * dA[1] + vm[10] * dA[2];
    const float dJ11 = vm[1] * dA[3] + vm[5] * dA[4] + vm[9] * dA[5];
    const float dJ12 = vm[2] * dA[3] + vm[6] * dA[4] + vm[10] * dA[5];
    const float tz = 1.f / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const float hx = cam.focal_x, hy = cam.focal_y;
    const float dtx = xmul * -hx * tz2 * dJ02;
    const float dty = ymul * -hy * tz2 * dJ12;
    const float dtz = -hx * tz2 * dJ00 - hy * tz2 * dJ11 + (2 * hx * t[0]) * tz3 * dJ02 + (2 * hy * t[1]) * tz3 * dJ12;
    dmean[0] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
    dmean[1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
    dmean[2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;
}

// backward.cu:406-423: dL/dmean2D -> dL/dmean3D through the perspective divide (added to dmean)
R3_HD void project_backward(const Camera& cam, float mx, float my, float mz, float g2x, float g2y, float* dmean)
{
    const float* pm = cam.proj;
    float mh[4];
    xform4x4(pm, mx, my, mz, mh);
    const float mw = 1.0f / (mh[3] + 0.0000001f);
    const float mul1 = (pm[0] * mx + pm[4] * my + pm[8] * mz + pm[12]) * mw * mw;
    const float mul2 = (pm[1] * mx + pm[5] * my + pm[9] * mz + pm[13]) * mw * mw;
    dmean[0] += (pm[0] * mw - pm[3] * mul1) * g2x + (pm[1] * mw - pm[3] * mul2) * g2y;
    dmean[1] += (pm[4] * mw - pm[7] * mul1) * g2x + (pm[5] * mw - pm[7] * mul2) * g2y;
    dmean[2] += (pm[8] * mw - pm[11] * mul1) * g2x + (pm[9] * mw - pm[11] * mul2) * g2y;
}

R3_HD float sign_(float v) { return (float)((v > 0.f) - (v < 0.f)); }

// Sink for one Gaussian's dL/dsh row; `put(e, v)` stores float e (= 3*k + ch).
struct ShGradPlain {
    float* p;
    R3_HD void put(int e, float v) const { p[e] = v; }
};

// backward.cu:61-150: d(colour channel ch)/d(unit view direction), d9[3 * ch + axis], for deg > 0.  A function of the SH
// row and the direction only -- the forward can evaluate it while it has the row staged (preprocess.hip) and spare the
// backward the 12 * M bytes per Gaussian of reading the row again.
template <class ShRow>
R3_HD void sh_dir_derivs(int deg, const ShRow& sh, float x, float y, float z, float* d9)
{
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    for (int ch = 0; ch < 3; ch++) {
#define R3_SH(k) sh.at(3 * (k) + ch)
        float dx_ = -R3_SH_C1 * R3_SH(3);
        float dy_ = -R3_SH_C1 * R3_SH(1);
        float dz_ = R3_SH_C1 * R3_SH(2);
        if (deg > 1) {
            dx_ += R3_SH_C2_0 * y * R3_SH(4) + R3_SH_C2_2 * 2.f * -x * R3_SH(6) + R3_SH_C2_3 * z * R3_SH(7) +
                   R3_SH_C2_4 * 2.f * x * R3_SH(8);
            dy_ += R3_SH_C2_0 * x * R3_SH(4) + R3_SH_C2_1 * z * R3_SH(5) + R3_SH_C2_2 * 2.f * -y * R3_SH(6) +
                   R3_SH_C2_4 * 2.f * -y * R3_SH(8);
            dz_ += R3_SH_C2_1 * y * R3_SH(5) + R3_SH_C2_2 * 2.f * 2.f * z * R3_SH(6) + R3_SH_C2_3 * x * R3_SH(7);
            if (deg > 2) {
                dx_ += (R3_SH_C3_0 * R3_SH(9) * 3.f * 2.f * xy + R3_SH_C3_1 * R3_SH(10) * yz +
                        R3_SH_C3_2 * R3_SH(11) * -2.f * xy + R3_SH_C3_3 * R3_SH(12) * -3.f * 2.f * xz +
                        R3_SH_C3_4 * R3_SH(13) * (-3.f * xx + 4.f * zz - yy) + R3_SH_C3_5 * R3_SH(14) * 2.f * xz +
                        R3_SH_C3_6 * R3_SH(15) * 3.f * (xx - yy));
                dy_ += (R3_SH_C3_0 * R3_SH(9) * 3.f * (xx - yy) + R3_SH_C3_1 * R3_SH(10) * xz +
                        R3_SH_C3_2 * R3_SH(11) * (-3.f * yy + 4.f * zz - xx) +
                        R3_SH_C3_3 * R3_SH(12) * -3.f * 2.f * yz + R3_SH_C3_4 * R3_SH(13) * -2.f * xy +
                        R3_SH_C3_5 * R3_SH(14) * -2.f * yz + R3_SH_C3_6 * R3_SH(15) * -3.f * 2.f * xy);
                dz_ += (R3_SH_C3_1 * R3_SH(10) * xy + R3_SH_C3_2 * R3_SH(11) * 4.f * 2.f * yz +
                        R3_SH_C3_3 * R3_SH(12) * 3.f * (2.f * zz - xx - yy) +
                        R3_SH_C3_4 * R3_SH(13) * 4.f * 2.f * xz + R3_SH_C3_5 * R3_SH(14) * (xx - yy));
            }
        }
#undef R3_SH
        d9[3 * ch] = dx_;
        d9[3 * ch + 1] = dy_;
        d9[3 * ch + 2] = dz_;
    }
}

// the forward's side of it: the direction exactly as sh_backward forms it
template <class ShRow>
R3_HD void sh_dir_derivs_at(int deg, const ShRow& sh, float mx, float my, float mz, const float* campos, float* d9)
{
    const float vx = mx - campos[0], vy = my - campos[1], vz = mz - campos[2];
    const float len = sqrtf(vx * vx + vy * vy + vz * vz);
    sh_dir_derivs(deg, sh, vx / len, vy / len, vz / len, d9);
}

// backward.cu:20-172: colour gradient -> SH coefficients (+ L1 sparsity term on bands >= 1) and,
// through the view direction, an extra contribution added to dmean.
// CACHED: `d9` holds sh_dir_derivs of this Gaussian as the forward left them and `sh` is never read (the sparsity term
// needs the coefficients' signs: callers take this form only with sparsity_mult == 0).
template <bool CACHED, class ShRow, class ShGrad>
R3_HD void sh_backward(int deg, const ShRow& sh, const ShGrad& dsh, const float* d9_cached, float mx, float my, float mz,
                       const float* campos, uint32_t clamp_bits, const float* dL_dcolor, float sparsity_mult, float* dmean)
{
    const float vx = mx - campos[0], vy = my - campos[1], vz = mz - campos[2];
    const float len = sqrtf(vx * vx + vy * vy + vz * vz);
    const float x = vx / len, y = vy / len, z = vz / len;
    float dRGB[3];
    for (int ch = 0; ch < 3; ch++) dRGB[ch] = dL_dcolor[ch] * (((clamp_bits >> ch) & 1u) ? 0.f : 1.f);
    float ddir[3] = {0.f, 0.f, 0.f};
    if (deg > 0) {
        float d9[9];
        if (CACHED) {
            for (int k = 0; k < 9; k++) d9[k] = d9_cached[k];
        } else {
            sh_dir_derivs(deg, sh, x, y, z, d9);
        }
        for (int ch = 0; ch < 3; ch++) {
            ddir[0] += d9[3 * ch] * dRGB[ch];
            ddir[1] += d9[3 * ch + 1] * dRGB[ch];
            ddir[2] += d9[3 * ch + 2] * dRGB[ch];
        }
    }
    // dL/dsh is written only now: `dsh` may alias the storage `sh` reads from (LDS staging row reused in place),
    // every element is read (sign) before it is overwritten and never read again.
    float Y[16];
    sh_basis(deg, x, y, z, Y);
    const int K = (deg + 1) * (deg + 1);
    for (int k = 0; k < K; k++)
        for (int ch = 0; ch < 3; ch++) {
            float g = Y[k] * dRGB[ch];
            if (!CACHED && k >= 1 && sparsity_mult != 0.f) g = g + sparsity_mult * sign_(sh.at(3 * k + ch));
            dsh.put(3 * k + ch, g);
        }
    // auxiliary.h:107-117 dnormvdv
    const float sum2 = vx * vx + vy * vy + vz * vz;
    const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmean[0] += ((+sum2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * invsum32;
    dmean[1] += (-vx * vy * ddir[0] + (sum2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * invsum32;
    dmean[2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (sum2 - vz * vz) * ddir[2]) * invsum32;
}

// backward.cu:311-374: dL/dcov3D[6] -> dL/dscale[3] (activated scale) and dL/dq[4]
R3_HD void cov3d_backward(const float* scale, float mod, const float* q, const float* dcov6, float* dscale, float* dq)
{
    const float r = q[0], qx = q[1], qy = q[2], qz = q[3];
    float R[9];
    quat_to_R(q, R);
    const float s[3] = {mod * scale[0], mod * scale[1], mod * scale[2]};
    const float dS[9] = {dcov6[0],        0.5f * dcov6[1], 0.5f * dcov6[2], 0.5f * dcov6[1], dcov6[3],
                         0.5f * dcov6[4], 0.5f * dcov6[2], 0.5f * dcov6[4], dcov6[5]};
    float dM[9];  // dM[3*k + j] = sum_m 2*M(k,m) * dS(m,j),  M(k,m) = s_k R(m,k)
    for (int k = 0; k < 3; k++) {
        const float m0 = 2.0f * (s[k] * R[0 + k]), m1 = 2.0f * (s[k] * R[3 + k]), m2 = 2.0f * (s[k] * R[6 + k]);
        for (int j = 0; j < 3; j++) dM[3 * k + j] = m0 * dS[0 + j] + m1 * dS[3 + j] + m2 * dS[6 + j];
    }
    for (int k = 0; k < 3; k++) dscale[k] = R[0 + k] * dM[3 * k + 0] + R[3 + k] * dM[3 * k + 1] + R[6 + k] * dM[3 * k + 2];
    float D[9];
    for (int k = 0; k < 3; k++)
        for (int j = 0; j < 3; j++) D[3 * k + j] = dM[3 * k + j] * s[k];
#define R3_D(a, b) D[3 * (a) + (b)]
    dq[0] = 2 * qz * (R3_D(0, 1) - R3_D(1, 0)) + 2 * qy * (R3_D(2, 0) - R3_D(0, 2)) + 2 * qx * (R3_D(1, 2) - R3_D(2, 1));
    dq[1] = 2 * qy * (R3_D(1, 0) + R3_D(0, 1)) + 2 * qz * (R3_D(2, 0) + R3_D(0, 2)) + 2 * r * (R3_D(1, 2) - R3_D(2, 1)) -
            4 * qx * (R3_D(2, 2) + R3_D(1, 1));
    dq[2] = 2 * qx * (R3_D(1, 0) + R3_D(0, 1)) + 2 * r * (R3_D(2, 0) - R3_D(0, 2)) + 2 * qz * (R3_D(1, 2) + R3_D(2, 1)) -
            4 * qy * (R3_D(2, 2) + R3_D(0, 0));
    dq[3] = 2 * r * (R3_D(0, 1) - R3_D(1, 0)) + 2 * qx * (R3_D(2, 0) + R3_D(0, 2)) + 2 * qy * (R3_D(1, 2) + R3_D(2, 1)) -
            4 * qz * (R3_D(1, 1) + R3_D(0, 0));
#undef R3_D
}

// ---------------------------------------------------------------------------------------------
// The covariance chain in double (round 4).
//
// backward.cu:228-306 (conic -> cov2D -> cov3D, and the covariance part of dL/dmean) and :311-374 (cov3D -> scale,
// quaternion) are quadratic forms of matrices whose entries are orders of magnitude larger than the result (for a
// splat of scale ratio 20 the 1e5-sized entries of dL/dcov3D contract to a dL/dscale of 10): evaluated in fp32 -- as
// the reference does and as cov2d_backward / cov3d_backward above restate -- the rounding of the intermediates alone
// puts dL/drotations 2e-4 and dL/dscales 5e-5 of the tensor's maximum away from the exact value at the benchmark shape,
// measured with the oracle's double evaluation (oracle/backward_f64.c; the same fp32 inputs through a double chain:
// 2e-6 / 4e-6).  Two fp32 evaluations then differ from EACH OTHER by more than north_star's 1e-4.  This stage is
// HBM-bound (a v_fma_f64 costs a SIMD 4.2 cycles per wave against 2.7 for v_fma_f32, tools/valu_rate.hip; the kernel waits
// for memory either way: 0.0625 vs 0.0614 ms), so the per-Gaussian backward evaluates the chain in double from
// the same fp32 inputs and rounds once at the end: the closest fp32 number to the exact gradient of what the forward
// computed.  Decisions (the 1.3 tan(fov) clamp masks) are taken in fp32 exactly as the forward took them, and the
// reference's conventions are kept (fp32 focal lengths and clamp limits, 1 / (det^2 + 1e-7), no quaternion-normalisation
// Jacobian, dL/dscale w.r.t. the modifier-scaled scale).  R3DGS_F64_CHAIN=0 / r3dgs_set_f64_chain(0) selects the fp32
// restatement above (A/B runs, the host-check shim).
// ---------------------------------------------------------------------------------------------
R3_HD void cov2d_backward_f64(const Camera& cam, float mx, float my, float mz, const float* c6f, float gA, float gB,
                              float gC, double* dcov6, float* dmean)
{
    const float* vm = cam.view;
    // masks and clamp side from the fp32 evaluation the forward made (ewa_A)
    float tf[3];
    xform4x3(vm, mx, my, mz, tf);
    const float limxf = 1.3f * cam.tan_fovx, limyf = 1.3f * cam.tan_fovy;
    const float txtz = tf[0] / tf[2], tytz = tf[1] / tf[2];
    const bool clx = txtz < -limxf || txtz > limxf, cly = tytz < -limyf || tytz > limyf;
    double t[3];
    for (int r = 0; r < 3; r++)
        t[r] = (double)vm[r] * mx + (double)vm[4 + r] * my + (double)vm[8 + r] * mz + (double)vm[12 + r];
    if (clx) t[0] = (txtz < 0.f ? -(double)limxf : (double)limxf) * t[2];
    if (cly) t[1] = (tytz < 0.f ? -(double)limyf : (double)limyf) * t[2];
    const double fx = cam.focal_x, fy = cam.focal_y, itz = 1.0 / t[2], itz2 = itz * itz;
    const double J00 = fx * itz, J02 = -fx * t[0] * itz2, J11 = fy * itz, J12 = -fy * t[1] * itz2;
    double A[6];   // J * Rw, Rw(i,j) = vm[4*j + i]
    for (int j = 0; j < 3; j++) {
        A[j] = J00 * vm[4 * j] + J02 * vm[4 * j + 2];
        A[3 + j] = J11 * vm[4 * j + 1] + J12 * vm[4 * j + 2];
    }
    const double S[9] = {c6f[0], c6f[1], c6f[2], c6f[1], c6f[3], c6f[4], c6f[2], c6f[4], c6f[5]};
    double AS[6];
    for (int i = 0; i < 2; i++)
        for (int k = 0; k < 3; k++) AS[3 * i + k] = A[3 * i] * S[k] + A[3 * i + 1] * S[3 + k] + A[3 * i + 2] * S[6 + k];
    const double a = AS[0] * A[0] + AS[1] * A[1] + AS[2] * A[2] + 0.3;
    const double b = AS[0] * A[3] + AS[1] * A[4] + AS[2] * A[5];
    const double c = AS[3] * A[3] + AS[4] * A[4] + AS[5] * A[5] + 0.3;
    const double det = a * c - b * b;
    const double k2 = 1.0 / (det * det + 0.0000001);
    // backward.cu:234-246 with (det - ac) = -b^2; gB is half the derivative w.r.t. the off-diagonal (backward.cu:572-577)
    const double da = k2 * (-c * c * gA + 2 * b * c * gB - b * b * gC);
    const double dc = k2 * (-a * a * gC + 2 * a * b * gB - b * b * gA);
    const double db = k2 * 2 * (b * c * gA - (det + 2 * b * b) * gB + a * b * gC);
    dcov6[0] = A[0] * A[0] * da + A[0] * A[3] * db + A[3] * A[3] * dc;
    dcov6[3] = A[1] * A[1] * da + A[1] * A[4] * db + A[4] * A[4] * dc;
    dcov6[5] = A[2] * A[2] * da + A[2] * A[5] * db + A[5] * A[5] * dc;
    dcov6[1] = 2 * A[0] * A[1] * da + (A[0] * A[4] + A[1] * A[3]) * db + 2 * A[3] * A[4] * dc;
    dcov6[2] = 2 * A[0] * A[2] * da + (A[0] * A[5] + A[2] * A[3]) * db + 2 * A[3] * A[5] * dc;
    dcov6[4] = 2 * A[2] * A[1] * da + (A[1] * A[5] + A[2] * A[4]) * db + 2 * A[4] * A[5] * dc;
    double dA[6];
    for (int j = 0; j < 3; j++) {
        dA[j] = 2 * AS[j] * da + AS[3 + j] * db;
        dA[3 + j] = 2 * AS[3 + j] * dc + AS[j] * db;
    }
    const double dJ00 = vm[0] * dA[0] + vm[4] * dA[1] + vm[8] * dA[2];
    const double dJ02 = vm[2] * dA[0] + vm[6] * dA[1] + vm[10] * dA[2];
    const double dJ11 = vm[1] * dA[3] + vm[5] * dA[4] + vm[9] * dA[5];
    const double dJ12 = vm[2] * dA[3] + vm[6] * dA[4] + vm[10] * dA[5];
    const double itz3 = itz2 * itz;
    const double dtx = clx ? 0.0 : -fx * itz2 * dJ02;
    const double dty = cly ? 0.0 : -fy * itz2 * dJ12;
    const double dtz = -fx * itz2 * dJ00 - fy * itz2 * dJ11 + 2 * fx * t[0] * itz3 * dJ02 + 2 * fy * t[1] * itz3 * dJ12;
    dmean[0] = (float)(vm[0] * dtx + vm[1] * dty + vm[2] * dtz);
    dmean[1] = (float)(vm[4] * dtx + vm[5] * dty + vm[6] * dtz);
    dmean[2] = (float)(vm[8] * dtx + vm[9] * dty + vm[10] * dtz);
}

// backward.cu:311-374 in double: Sigma = R S^2 R^T, G = dL/dSigma as a symmetric matrix (off-diagonals of dcov6 halved)
//   dL/ds_k = 2 s_k (R^T G R)_kk,   dL/dR = 2 G R S^2,   dL/dq_n = sum_ij dL/dR_ij dR_ij/dq_n   (q not normalised here)
R3_HD void cov3d_backward_f64(const float* scale, float mod, const float* qf, const double* dcov6, float* dscale, float* dq)
{
    const double r = qf[0], x = qf[1], y = qf[2], z = qf[3];
    const double R[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)};
    const double s[3] = {(double)mod * scale[0], (double)mod * scale[1], (double)mod * scale[2]};
    const double G[9] = {dcov6[0], 0.5 * dcov6[1], 0.5 * dcov6[2], 0.5 * dcov6[1], dcov6[3],
                         0.5 * dcov6[4], 0.5 * dcov6[2], 0.5 * dcov6[4], dcov6[5]};
    double E[9];   // E = 2 G R S^2 (= dL/dR), column k scaled by s_k^2
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) E[3 * i + k] = 2 * (G[3 * i] * R[k] + G[3 * i + 1] * R[3 + k] + G[3 * i + 2] * R[6 + k]);
    for (int k = 0; k < 3; k++)   // dL/ds_k = s_k * sum_i R(i,k) * (2 G R)(i,k)
        dscale[k] = (float)(s[k] * (R[k] * E[k] + R[3 + k] * E[3 + k] + R[6 + k] * E[6 + k]));
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) E[3 * i + k] *= s[k] * s[k];
    // dR/dq of the formula above, contracted with E
    dq[0] = (float)(2 * (z * (E[3] - E[1]) + y * (E[2] - E[6]) + x * (E[7] - E[5])));
    dq[1] = (float)(2 * (y * (E[1] + E[3]) + z * (E[2] + E[6]) + r * (E[7] - E[5])) - 4 * x * (E[4] + E[8]));
    dq[2] = (float)(2 * (x * (E[1] + E[3]) + r * (E[2] - E[6]) + z * (E[5] + E[7])) - 4 * y * (E[0] + E[8]));
    dq[3] = (float)(2 * (r * (E[3] - E[1]) + x * (E[2] + E[6]) + y * (E[5] + E[7])) - 4 * z * (E[0] + E[4]));
}

// backward.cu:433 -- sigmoid chain, evaluated in double like the reference's `1.0 - w`
R3_HD float opacity_backward(float dL_dact, float o) { return (float)((double)dL_dact * ((double)o * (1.0 - (double)o))); }

}  // namespace r3
