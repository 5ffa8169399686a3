// blend_math.h -- per-(pixel, Gaussian) arithmetic of the alpha-blend forward and backward.
//
// __host__ __device__ so that tests/hostcheck/hostcheck.hip can run the very same step functions
// on the CPU.  The blend translation units are compiled with FMA contraction allowed and the
// hardware exp (v_exp_f32): results are compared with the oracle to 1e-5 (colour) / 1e-4 (grads),
// not bit-for-bit.
//
// Follows (relative to /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer):
//   forward.cu:528-570  inner loop of renderCUDA (fwd)   -> fwd_step()
//   backward.cu:520-593 inner loop of renderCUDA (bwd)   -> bwd_step()
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace r3 {

#ifndef R3_HD
#define R3_HD __host__ __device__ __forceinline__
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define R3_EXP2(x) __builtin_amdgcn_exp2f(x)  // v_exp_f32
#define R3_LOG2(x) __builtin_amdgcn_logf(x)  // v_log_f32
#define R3_RCP(x) __builtin_amdgcn_rcpf(x)  // v_rcp_f32, 1 ulp
#else
#define R3_EXP2(x) exp2f(x)
#define R3_LOG2(x) log2f(x)
#define R3_RCP(x) (1.0f / (x))
#endif

// the 9 floats a pixel needs from a Gaussian
struct Splat {
    float x, y, cA, cB, cC, op, r, g, b;
};

// ---- region pre-test ----------------------------------------------------------------------------
// The reference bins a Gaussian into every tile of the bounding SQUARE of its 3-sigma radius
// (auxiliary.h:46-56), so in a typical scene more than half of a tile's list entries reach no pixel of the
// tile with alpha >= 1/255 (54% at the BASELINE shape; 72% per 8x8 quadrant).  Each such entry still costs the
// per-pixel evaluation.  Because  alpha >= 1/255  <=>  q(d) <= ln(255*opacity)  with the convex quadratic
// q(d) = 0.5*(A dx^2 + C dy^2) + B dx dy, one exact minimisation of q over a pixel rectangle decides for the
// whole rectangle.  The blend kernels run this once per (entry, 8x8 quadrant) -- one lane per entry, while the
// entry is being staged -- and skip entries/quadrants that provably contribute nothing.  The skip is
// conservative: a 2e-3 margin, and on top of it the cancellation margin of gauss_math.h tighten_rect -- an anisotropic
// splat seen at an angle has q = a small difference of large products, whose fp32 evaluation (the reference's, the
// oracle's, this library's) is only good to a relative ~1.6e-6 kappa, kappa = AC / (AC - B^2); tau is divided by
// 1 - 4e-6 kappa and a splat with kappa > 6e4 is never skipped.  So every per-pixel decision, n_contrib and the image
// are exactly what they are without the pre-test.

// min over pixels (px,py) in [X0,X1]x[Y0,Y1] of q(x-px, y-py); requires A > 0 and C > 0
R3_HD float region_qmin(float x, float y, float A, float B, float C, float X0, float X1, float Y0, float Y1)
{
    const float dx0 = x - X1, dx1 = x - X0, dy0 = y - Y1, dy1 = y - Y0;  // d-rectangle [dx0,dx1]x[dy0,dy1]
    if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return 0.f;  // centre inside
    const float nbc = -B * R3_RCP(C), nba = -B * R3_RCP(A);
    float qm;
    {   // edges dx = dx0 and dx = dx1: minimise over dy (convex, coefficient C/2)
        float t = fminf(fmaxf(nbc * dx0, dy0), dy1);
        qm = 0.5f * (A * dx0 * dx0 + C * t * t) + B * dx0 * t;
        t = fminf(fmaxf(nbc * dx1, dy0), dy1);
        qm = fminf(qm, 0.5f * (A * dx1 * dx1 + C * t * t) + B * dx1 * t);
    }
    {   // edges dy = dy0 and dy = dy1
        float t = fminf(fmaxf(nba * dy0, dx0), dx1);
        qm = fminf(qm, 0.5f * (A * t * t + C * dy0 * dy0) + B * t * dy0);
        t = fminf(fmaxf(nba * dy1, dx0), dx1);
        qm = fminf(qm, 0.5f * (A * t * t + C * dy1 * dy1) + B * t * dy1);
    }
    return qm;
}

// false only if NO pixel of the rectangle can reach alpha >= 1/255 for this splat
R3_HD bool region_may_contribute(const Splat& s, float X0, float X1, float Y0, float Y1)
{
    if (!(s.cA > 0.f) || !(s.cC > 0.f)) return true;  // degenerate conic: never skip
    const float tau = 0.6931471805599453f * R3_LOG2(255.0f * s.op);   // alpha >= 1/255  <=>  q <= tau
    const float qmin = region_qmin(s.x, s.y, s.cA, s.cB, s.cC, X0, X1, Y0, Y1);
    const float ac = s.cA * s.cC, det = fmaf(-s.cB, s.cB, ac);
    const float keep = 1.0f - 4e-6f * (ac * R3_RCP(det));             // 1 - r; det <= 0 or kappa > 6e4: never skip
    if (!(det > 0.f) || !(keep > 0.75f)) return true;
    return !(qmin * keep > tau + 2e-3f * fabsf(tau) + 2e-3f);  // NaN anywhere => keep
}

// The same 9 floats with the conic pre-scaled so that  log2(G) = qa*dx^2 + qb*dx*dy + qc*dy^2  (G = exp(power) of
// forward.cu:534-537): the -0.5 and the log2(e) in front of v_exp_f32 are paid once per list entry by the lane that
// stages it (3 multiplies per 64-entry chunk) instead of once per (pixel, entry).  Forward and backward evaluate the
// falloff through the one function below, with explicit FMAs, so both take the same per-pixel decisions.
struct QSplat {
    float x, y, qa, qb, qc, op, r, g, b;
};

constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

R3_HD QSplat scale_splat(const Splat& s)
{
    QSplat q;
    q.x = s.x;
    q.y = s.y;
    q.qa = (-0.5f * kLog2e) * s.cA;
    q.qb = -kLog2e * s.cB;
    q.qc = (-0.5f * kLog2e) * s.cC;
    q.op = s.op;
    q.r = s.r;
    q.g = s.g;
    q.b = s.b;
    return q;
}

// The three products of the pixel offset are formed first and kept: the backward's second moments are sums of
// m * dx^2, m * dx dy, m * dy^2 and take them as they are (one FMA each instead of a multiply and an FMA).
struct Offset2 {
    float dx, dy, dxx, dxy, dyy;
};

R3_HD Offset2 pixel_offset(const QSplat& s, float pxf, float pyf)
{
    Offset2 o;
    o.dx = s.x - pxf;
    o.dy = s.y - pyf;
    o.dxx = o.dx * o.dx;
    o.dxy = o.dx * o.dy;
    o.dyy = o.dy * o.dy;
    return o;
}

R3_HD float log2_falloff(const QSplat& s, const Offset2& o)
{
    return fmaf(s.qa, o.dxx, fmaf(s.qb, o.dxy, s.qc * o.dyy));
}

struct FwdPix {
    float T;        // running product of (1 - alpha) over every entry that passed the alpha test.  The reference stops a
                    // pixel at the first entry with T (1 - alpha) < 0.0001 (forward.cu:547-552, `done`); the product only
                    // falls, so "T has dropped below 0.0001" IS that flag: the running product is simply carried on, every
                    // later entry fails the same test by itself, and nothing has to be selected per entry (v_cndmask
                    // issues at half the rate of an FMA).  A pixel outside the image starts at -1: never blended.
    float Tf;       // T after the last BLENDED entry: the reference's final T
    float C0, C1, C2;
    uint32_t last;  // 1-based list position of the last blended entry (n_contrib)
};

R3_HD void fwd_pix_init(FwdPix& p, bool inside)
{
    p.T = inside ? 1.0f : -1.0f;
    p.Tf = 1.0f;
    p.C0 = p.C1 = p.C2 = 0.f;
    p.last = 0;
}

R3_HD bool fwd_pix_live(const FwdPix& p) { return p.T >= 0.0001f; }
R3_HD float fwd_pix_T(const FwdPix& p) { return p.Tf; }

// alpha of one list entry at one pixel (forward.cu:534-546) and whether the reference's first skip (power > 0, which
// only a conic that is not positive definite can trigger) leaves it in.  The two skips (power > 0, alpha < 1/255) stay
// ONE divergence point: two compares whose lane masks are AND-ed on the scalar unit (a compare + select + compare costs
// one more half-rate VALU instruction: v_cmp and v_cndmask issue at half the rate of an FMA, profiles/r03_valu_rate.txt).
R3_HD float fwd_alpha(const QSplat& s, float pxf, float pyf, bool* in_bound)
{
    const float p2 = log2_falloff(s, pixel_offset(s, pxf, pyf));
    *in_bound = p2 <= 0.0f;
    return fminf(0.99f, s.op * R3_EXP2(p2));
}

// Blends an entry of opacity-weighted falloff `alpha` into pixel `p`.  Returns 0: skipped, 1: blended, 2: pixel
// saturated (done).  `pos1` = 1-based position of the entry in the tile list.  *T_before = transmittance it saw.
R3_HD int fwd_apply(const QSplat& s, float alpha, bool in_bound, uint32_t pos1, FwdPix& p, float* T_before)
{
    if (!(in_bound && alpha >= 1.0f / 255.0f)) return 0;
    const float w = alpha * p.T;
    const float Tn = p.T - w;                 // T * (1 - alpha), forward.cu:547
    *T_before = p.T;
    p.T = Tn;
    if (!(Tn >= 0.0001f)) return 2;           // saturated here, earlier, or outside the image: not blended
    p.C0 += s.r * w;
    p.C1 += s.g * w;
    p.C2 += s.b * w;
    p.Tf = Tn;
    p.last = pos1;
    return 1;
}

// One list entry against one pixel (forward.cu:528-570).
R3_HD int fwd_step(const QSplat& s, float pxf, float pyf, uint32_t pos1, FwdPix& p, float* T_before)
{
    bool in_bound;
    const float alpha = fwd_alpha(s, pxf, pyf, &in_bound);
    return fwd_apply(s, alpha, in_bound, pos1, p, T_before);
}

R3_HD int fwd_step(const Splat& s, float pxf, float pyf, uint32_t pos1, FwdPix& p, float* T_before)
{
    return fwd_step(scale_splat(s), pxf, pyf, pos1, p, T_before);
}

// per-pixel state of the back-to-front walk.  The reference keeps accum_rec[3] and last_color[3] per
// pixel (backward.cu:487-494) and forms  dL_dalpha = sum_ch (c_ch - accum_rec_ch) * g_ch.  Both recurrences
// are linear, so only their projection on the pixel's upstream gradient g is needed:
//   A = accum_rec . g,      A <- A + alpha * (c . g - A)     after an entry (alpha, c) has been processed,
// which is the reference's  accum_rec = last_alpha * last_color + (1 - last_alpha) * accum_rec  evaluated when the entry
// is left instead of when the next one is entered -- the same arithmetic up to association (the difference c . g - A is
// needed for dL_dalpha anyway), 13 fewer registers per pixel than carrying accum_rec, last_color and last_alpha.
// A starts at bg . g instead of 0: the background is what lies behind the last contributor, and with it inside the
// recurrence (c . g - A) T already contains the reference's separate term -T_final / (1 - alpha) * (bg . g)
// (backward.cu:569-572):  A_k = sum_{j behind k} alpha_j c_j.g prod_{k<i<j} (1 - alpha_i) + bg.g prod_{i behind k} (1 - alpha_i)
// and T_k prod_{i behind k} (1 - alpha_i) = T_final / (1 - alpha_k).
struct BwdPix {
    float T;           // transmittance in front of the current entry (recovered by division)
    float A;           // see above
    float g0, g1, g2;  // dL_dpixel
    uint32_t last;     // n_contrib
};

R3_HD void bwd_pix_init(BwdPix& p, float T_final, uint32_t last, float g0, float g1, float g2, float bg_dot)
{
    p.T = T_final;
    p.A = bg_dot;
    p.g0 = g0;
    p.g1 = g1;
    p.g2 = g2;
    p.last = last;
}

// per-Gaussian partial gradient (2D stage): dmean2D WITHOUT the 0.5*W / 0.5*H viewport factors of backward.cu:498-499
struct SplatGrad {
    float mx, my, cA, cB, cC, op, r, g, b;
};

// What a lane accumulates per list entry over its pixels and the wave then reduces.  The entry's conic and opacity are
// the same for every pixel, so the five geometry gradients of backward.cu:553-566 are kept as the moments
//   m = G * dL_dalpha,   sm = sum m,  (sx, sy) = sum m * d,  (sxx, sxy, syy) = sum m * d d^T
// and turned into dL_dmean2D / dL_dconic / dL_dopacity ONCE per (tile, entry) after the reduction (splat_grad_of):
// 9 VALU per (pixel, entry) instead of 15.
struct SplatSums {
    float sx, sy, sxx, sxy, syy, sm, r, g, b;
};

R3_HD SplatGrad splat_grad_of(const QSplat& s, const SplatSums& u)
{
    SplatGrad g;
    const float k = s.op * kLn2;   // conic = -ln2 * (2 qa, qb, 2 qc)
    g.mx = k * (2.f * s.qa * u.sx + s.qb * u.sy);
    g.my = k * (2.f * s.qc * u.sy + s.qb * u.sx);
    const float h = -0.5f * s.op;
    g.cA = h * u.sxx;
    g.cB = h * u.sxy;
    g.cC = h * u.syy;
    g.op = u.sm;
    g.r = u.r;
    g.g = u.g;
    g.b = u.b;
    return g;
}

// One list entry (0-based position `pos`) against one pixel, in two halves so that the kernel can ballot the decision
// before it branches on it.  bwd_test: the reference's three skips -- entry behind this pixel's last contributor
// (backward.cu:524-526), power > 0, alpha < 1/255 -- evaluated together: one divergence point instead of three.
struct BwdEval {
    Offset2 o;
    float G, alpha;
    bool in_list, in_bound, visible;   // the three decisions, kept apart: a kernel can ballot each compare for free
    R3_HD bool valid() const { return in_list && in_bound && visible; }
};

R3_HD bool bwd_test(const QSplat& s, float pxf, float pyf, uint32_t pos, const BwdPix& p, BwdEval& e)
{
    e.o = pixel_offset(s, pxf, pyf);
    const float p2 = log2_falloff(s, e.o);
    e.G = R3_EXP2(p2);
    e.alpha = fminf(0.99f, s.op * e.G);
    e.in_list = pos < p.last;
    e.in_bound = p2 <= 0.0f;
    e.visible = e.alpha >= 1.0f / 255.0f;
    return e.valid();
}

R3_HD void bwd_accumulate(const QSplat& s, const BwdEval& e, BwdPix& p, SplatSums& a)
{
    const float ra = R3_RCP(1.0f - e.alpha);
    p.T = p.T * ra;  // T recovered by division (backward.cu:541)
    const float dch = e.alpha * p.T;
    a.r += dch * p.g0;
    a.g += dch * p.g1;
    a.b += dch * p.g2;
    const float cg = s.r * p.g0 + s.g * p.g1 + s.b * p.g2;
    const float behind = cg - p.A;
    const float dL_dalpha = behind * p.T;
    p.A = fmaf(e.alpha, behind, p.A);
    const float m = e.G * dL_dalpha;
    a.sm += m;
    a.sx = fmaf(m, e.o.dx, a.sx);
    a.sy = fmaf(m, e.o.dy, a.sy);
    a.sxx = fmaf(m, e.o.dxx, a.sxx);
    a.sxy = fmaf(m, e.o.dxy, a.sxy);
    a.syy = fmaf(m, e.o.dyy, a.syy);
}

// Whole step for one (pixel, entry) pair, adding into `a`; returns true when the entry contributed.
R3_HD bool bwd_step(const Splat& s0, float pxf, float pyf, uint32_t pos, BwdPix& p, SplatGrad& a)
{
    const QSplat s = scale_splat(s0);
    BwdEval e;
    if (!bwd_test(s, pxf, pyf, pos, p, e)) return false;
    SplatSums u;
    u.sx = u.sy = u.sxx = u.sxy = u.syy = u.sm = u.r = u.g = u.b = 0.f;
    bwd_accumulate(s, e, p, u);
    const SplatGrad g = splat_grad_of(s, u);
    a.mx += g.mx;
    a.my += g.my;
    a.cA += g.cA;
    a.cB += g.cB;
    a.cC += g.cC;
    a.op += g.op;
    a.r += g.r;
    a.g += g.g;
    a.b += g.b;
    return true;
}

}  // namespace r3
