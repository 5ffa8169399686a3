/*
 * backward_f64.c -- double-precision evaluation of the rasterizer's backward pass.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as raster_oracle.c: tests/, smoke() and bench.py's cpu_baseline leg may use it,
 * the product may not).
 *
 * The forward-state arrays (pixel means, conic + opacity, colours, 3D covariances) are passed as double so that the pin
 * test can hand over an fp64 forward's values; in normal use they are the fp32 numbers of the oracle's state, widened.
 *
 * Purpose: the reference (DGR/cuda_rasterizer/backward.cu) evaluates its backward in fp32 and sums the per-pixel terms
 * with unordered float atomics; raster_oracle.c restates that arithmetic in fp32 (sums in double).  Two fp32
 * evaluations of the cancellation-heavy covariance chain (backward.cu:228-306 conic -> cov2D -> cov3D, :311-374
 * cov3D -> scale / quaternion) can differ from each other by more than either differs from the exact value, so a test that
 * only compares the two cannot tell which one is off.  This file gives the third number: the EXACT gradient (double
 * arithmetic, ~1e-15) of the function the forward evaluated, with
 *   - the forward's discrete decisions taken from the fp32 state (which entries a pixel blended: power > 0 and
 *     alpha < 1/255 skips decided in fp32 exactly as forward.cu:531-546 / backward.cu:527-538 decide them; the last
 *     contributor from n_contrib; the 1.3 tan(fov) clamp masks and the colour clamps as the forward recorded them),
 *   - the per-Gaussian quantities the reference's backward reads from the forward's buffers (pixel means, conic, opacity,
 *     colour, 3D covariance) taken as the fp32 numbers stored there.
 *
 * It is deliberately NOT a transcription of the reference's closed forms (raster_oracle.c is): every stage is written
 * as the matrix calculus it is, so that a shared misreading of backward.cu in raster_oracle.c and in the product's
 * gauss_math.h would show up here.
 *   blend        : transmittances by forward products from T = 1 (not by division from final_T, backward.cu:541),
 *                  colour behind an entry as a suffix sum;  dL/dalpha_k = g . (c_k T_k - S_k / (1 - alpha_k))
 *   conic        : Q = Sigma'^-1  =>  dL/dSigma' = -Q (dL/dQ) Q, with the reference's regularised 1 / (det^2 + 1e-7)
 *   cov2D        : Sigma' = A Sigma A^T + 0.3 I  =>  dL/dSigma = A^T G A,  dL/dA = 2 G A Sigma
 *   projection   : A = J(t) Rw, t = Rw mu + t0; the reference's convention that a clamped t.x / t.y is a constant
 *                  for d/dt.x, d/dt.y but is the value used in dJ/dt.z (backward.cu:283-297)
 *   covariance   : Sigma = R S^2 R^T  =>  dL/ds_k = 2 s_k (R^T G R)_kk,  dL/dR = 2 G R S^2,  dL/dq = sum dL/dR_ij dR_ij/dq
 *                  (no normalisation Jacobian, dL/dscale w.r.t. the modifier-scaled scale: the reference's conventions)
 *   SH           : dL/dsh_k = Y_k(d) dRGB,  dL/dd = sum_k grad Y_k (sh_k . dRGB),  through d = v / |v|
 * Conventions of the outputs are the reference's: dL_dmean2D in NDC units (x 0.5 W, x 0.5 H), dL_dconic.y = HALF the
 * derivative w.r.t. the symmetric off-diagonal (backward.cu:572-577), dL_dcov3D off-diagonals = the FULL derivative
 * w.r.t. the shared entry (backward.cu:262-270).
 * Pinned by tests/test_oracle_f64.py against fp64 autograd through oracle/torch_ref.py (an independent statement of the
 * forward), and compared with raster_oracle.c's fp32 backward there.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16

static inline int imin_(int a, int b) { return a < b ? a : b; }
static inline float fminf__(float a, float b) { return a < b ? a : b; }

/* acc layout per Gaussian (double[9]): dmean2D.x, .y (NDC units), dL/dA, HALF dL/dB, dL/dC, dL/do, dL/drgb[3] */
void orc_blend_bwd_f64(int P, int W, int H, const uint32_t* ranges, const uint32_t* point_list, const float* bg,
                       const double* xy, const double* conic_op, const double* colors, const uint32_t* n_contrib,
                       const float* dL_dpix, double* acc /* [P][9], zeroed here */)
{
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    memset(acc, 0, sizeof(double) * 9 * (size_t)P);
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int ty = 0; ty < gy; ty++)
        for (int tx = 0; tx < gx; tx++) {
            const int tile = ty * gx + tx;
            const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
            if (r1 <= r0) continue;
            const uint32_t n = r1 - r0;
            double* loc = (double*)calloc((size_t)n * 9, sizeof(double));
            double* Tk = (double*)malloc(sizeof(double) * n);   /* transmittance in front of entry k */
            double* Ak = (double*)malloc(sizeof(double) * n);   /* its alpha (0: skipped) */
            double* Gk = (double*)malloc(sizeof(double) * n);
            for (int py = ty * TILE; py < imin_(H, (ty + 1) * TILE); py++)
                for (int px = tx * TILE; px < imin_(W, (tx + 1) * TILE); px++) {
                    const size_t pix = (size_t)W * py + px;
                    const uint32_t last = n_contrib[pix];
                    if (!last) continue;
                    const float pxf = (float)px, pyf = (float)py;
                    double g[3];
                    for (int ch = 0; ch < 3; ch++) g[ch] = (double)dL_dpix[(size_t)ch * H * W + pix];
                    /* forward products */
                    double T = 1.0;
                    for (uint32_t k = 0; k < last; k++) {
                        const uint32_t id = point_list[r0 + k];
                        const double* co = conic_op + 4 * id;
                        /* the decisions, in fp32 as the kernels take them */
                        const float cf[4] = {(float)co[0], (float)co[1], (float)co[2], (float)co[3]};
                        const float dxf = (float)xy[2 * id] - pxf, dyf = (float)xy[2 * id + 1] - pyf;
                        const float powf_ = -0.5f * (cf[0] * dxf * dxf + cf[2] * dyf * dyf) - cf[1] * dxf * dyf;
                        Ak[k] = 0.0;
                        Tk[k] = T;
                        if (powf_ > 0.0f) continue;
                        if (fminf__(0.99f, cf[3] * expf(powf_)) < 1.0f / 255.0f) continue;
                        /* the values, in double */
                        const double dx = xy[2 * id] - (double)pxf, dy = xy[2 * id + 1] - (double)pyf;
                        const double q = -0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        const double G = exp(q);
                        double a = co[3] * G;
                        if (a > 0.99) a = 0.99;   /* straight-through in the backward (backward.cu:537,576) */
                        Gk[k] = G;
                        Ak[k] = a;
                        T *= 1.0 - a;
                    }
                    /* suffix sums: S = colour composited behind entry k (not normalised) + T_end * bg */
                    double S[3] = {T * (double)bg[0], T * (double)bg[1], T * (double)bg[2]};
                    for (int64_t k = (int64_t)last - 1; k >= 0; k--) {
                        const double a = Ak[k];
                        if (a == 0.0) continue;
                        const uint32_t id = point_list[r0 + k];
                        const double* co = conic_op + 4 * id;
                        const double Tf = Tk[k];
                        double* l = loc + 9 * (size_t)k;
                        double dL_da = 0;
                        for (int ch = 0; ch < 3; ch++) {
                            const double c = colors[3 * id + ch];
                            l[6 + ch] += a * Tf * g[ch];
                            dL_da += g[ch] * (c * Tf - S[ch] / (1.0 - a));
                            S[ch] += c * a * Tf;
                        }
                        const double o = co[3], G = Gk[k];
                        const double dL_dG = o * dL_da;
                        const double dx = xy[2 * id] - (double)pxf, dy = xy[2 * id + 1] - (double)pyf;
                        const double A = co[0], B = co[1], C = co[2];
                        /* G = exp(-0.5 (A dx^2 + C dy^2) - B dx dy), d = mean - pixel */
                        l[0] += dL_dG * G * (-(A * dx + B * dy)) * (0.5 * W);
                        l[1] += dL_dG * G * (-(C * dy + B * dx)) * (0.5 * H);
                        l[2] += dL_dG * G * (-0.5 * dx * dx);
                        l[3] += dL_dG * G * (-0.5 * dx * dy);   /* half of d/dB (-dx dy) */
                        l[4] += dL_dG * G * (-0.5 * dy * dy);
                        l[5] += G * dL_da;
                    }
                }
            for (uint32_t k = 0; k < n; k++) {
                const double* l = loc + 9 * (size_t)k;
                double* a = acc + 9 * (size_t)point_list[r0 + k];
                for (int c = 0; c < 9; c++)
                    if (l[c] != 0.0) {
#pragma omp atomic
                        a[c] += l[c];
                    }
            }
            free(loc);
            free(Tk);
            free(Ak);
            free(Gk);
        }
}

/* Pin-test switch.  0 (default): the reference's conventions where they deviate from the pure derivative -- focal lengths
 * and clamp limits formed in fp32 (rasterizer_impl.cu:386-387, forward.cu:168-169) and the regularised 1 / (det^2 + 1e-7)
 * of backward.cu:234 (which is NOT the derivative of forward.cu:419-423's 1 / det: it costs up to ~1e-5 of a small
 * Gaussian's gradient).  1: the pure derivative of the forward, which fp64 autograd must reproduce to ~1e-12. */
static int g_pure = 0;
void orc_f64_set_pure(int on) { g_pure = on; }

/* 3x3 helpers, row-major */
static void mat3_mul(const double* a, const double* b, double* o)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) o[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

static void quat_R(const double* q, double* R)
{
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z);
    R[1] = 2 * (x * y - r * z);
    R[2] = 2 * (x * z + r * y);
    R[3] = 2 * (x * y + r * z);
    R[4] = 1 - 2 * (x * x + z * z);
    R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (x * z - r * y);
    R[7] = 2 * (y * z + r * x);
    R[8] = 1 - 2 * (x * x + y * y);
}

/* dR[i][9]: derivative of the nine entries w.r.t. q_i, q = (r, x, y, z) taken as four free variables */
static void quat_dR(const double* q, double dR[4][9])
{
    const double r = q[0], x = q[1], y = q[2], z = q[3];
    const double d_r[9] = {0, -2 * z, 2 * y, 2 * z, 0, -2 * x, -2 * y, 2 * x, 0};
    const double d_x[9] = {0, 2 * y, 2 * z, 2 * y, -4 * x, -2 * r, 2 * z, 2 * r, -4 * x};
    const double d_y[9] = {-4 * y, 2 * x, 2 * r, 2 * x, 0, 2 * z, -2 * r, 2 * z, -4 * y};
    const double d_z[9] = {-4 * z, -2 * r, 2 * x, 2 * r, -4 * z, 2 * y, 2 * x, 2 * y, 0};
    memcpy(dR[0], d_r, sizeof d_r);
    memcpy(dR[1], d_x, sizeof d_x);
    memcpy(dR[2], d_y, sizeof d_y);
    memcpy(dR[3], d_z, sizeof d_z);
}

/* real SH basis (the polynomial form of utils/sh_utils.py / forward.cu:115-148) and its gradient w.r.t. (x, y, z) */
static void sh_basis_grad(int deg, double x, double y, double z, double* Y, double (*dY)[3])
{
    const double C0 = 0.28209479177387814, C1 = 0.4886025119029199;
    const double C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
                          0.5462742152960396};
    const double C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                          -0.4570457994644658, 1.445305721320277, -0.5900435899266435};
    for (int k = 0; k < 16; k++) {
        Y[k] = 0;
        dY[k][0] = dY[k][1] = dY[k][2] = 0;
    }
    Y[0] = C0;
    if (deg < 1) return;
    Y[1] = -C1 * y, dY[1][1] = -C1;
    Y[2] = C1 * z, dY[2][2] = C1;
    Y[3] = -C1 * x, dY[3][0] = -C1;
    if (deg < 2) return;
    Y[4] = C2[0] * x * y, dY[4][0] = C2[0] * y, dY[4][1] = C2[0] * x;
    Y[5] = C2[1] * y * z, dY[5][1] = C2[1] * z, dY[5][2] = C2[1] * y;
    Y[6] = C2[2] * (2 * z * z - x * x - y * y), dY[6][0] = C2[2] * -2 * x, dY[6][1] = C2[2] * -2 * y, dY[6][2] = C2[2] * 4 * z;
    Y[7] = C2[3] * x * z, dY[7][0] = C2[3] * z, dY[7][2] = C2[3] * x;
    Y[8] = C2[4] * (x * x - y * y), dY[8][0] = C2[4] * 2 * x, dY[8][1] = C2[4] * -2 * y;
    if (deg < 3) return;
    Y[9] = C3[0] * y * (3 * x * x - y * y), dY[9][0] = C3[0] * 6 * x * y, dY[9][1] = C3[0] * (3 * x * x - 3 * y * y);
    Y[10] = C3[1] * x * y * z, dY[10][0] = C3[1] * y * z, dY[10][1] = C3[1] * x * z, dY[10][2] = C3[1] * x * y;
    Y[11] = C3[2] * y * (4 * z * z - x * x - y * y);
    dY[11][0] = C3[2] * -2 * x * y, dY[11][1] = C3[2] * (4 * z * z - x * x - 3 * y * y), dY[11][2] = C3[2] * 8 * y * z;
    Y[12] = C3[3] * z * (2 * z * z - 3 * x * x - 3 * y * y);
    dY[12][0] = C3[3] * -6 * x * z, dY[12][1] = C3[3] * -6 * y * z, dY[12][2] = C3[3] * (6 * z * z - 3 * x * x - 3 * y * y);
    Y[13] = C3[4] * x * (4 * z * z - x * x - y * y);
    dY[13][0] = C3[4] * (4 * z * z - 3 * x * x - y * y), dY[13][1] = C3[4] * -2 * x * y, dY[13][2] = C3[4] * 8 * x * z;
    Y[14] = C3[5] * z * (x * x - y * y), dY[14][0] = C3[5] * 2 * x * z, dY[14][1] = C3[5] * -2 * y * z, dY[14][2] = C3[5] * (x * x - y * y);
    Y[15] = C3[6] * x * (x * x - 3 * y * y), dY[15][0] = C3[6] * (3 * x * x - 3 * y * y), dY[15][1] = C3[6] * -6 * x * y;
}

/* Per-Gaussian backward in double.  `acc` is orc_blend_bwd_f64's output.  vm / pm: the transposed 4x4 matrices the
 * reference passes (flat m[4*c + r] = entry (r, c)).  cov3Ds: the fp32 covariances the forward stored (or the
 * precomputed ones).  clamped: the forward's colour clamp flags.  Outputs (double): dmean3D[P][3], dcov3D[P][6],
 * dsh[P][M][3], dscale[P][3], drot[P][4], dopacity[P] (w.r.t. the RAW opacity), all zero for radii <= 0. */
void orc_preprocess_bwd_f64(int P, int M, const int* degs, const float* means, const int* radii, const float* shs,
                            const unsigned char* clamped, const float* scales, const float* rots, float mod,
                            const double* cov3Ds, const float* vm, const float* pm, const float* campos, int W, int H,
                            float tanx, float tany, const double* conic_op, const double* acc, double* dmean3D,
                            double* dcov3D, double* dsh, double* dscale, double* drot, double* dopacity, float lambda_sh)
{
    /* focal lengths as the reference forms them: fp32 (rasterizer_impl.cu:386-387) */
    const double fx = g_pure ? W / (2.0 * (double)tanx) : (double)(W / (2.0f * tanx));
    const double fy = g_pure ? H / (2.0 * (double)tany) : (double)(H / (2.0f * tany));
    const double limx = g_pure ? 1.3 * (double)tanx : (double)(1.3f * tanx);
    const double limy = g_pure ? 1.3 * (double)tany : (double)(1.3f * tany);
    double mult = 0;
    if (lambda_sh != 0.f) {
        int V = 0;
        for (int i = 0; i < P; i++) V += radii[i] > 0;
        mult = (double)(lambda_sh / (float)(V * 15 * 3));
    }
    double Rw[9], Fm[16];   /* Rw(i,j) = vm[4j + i]; F(i,j) = pm[4j + i] */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rw[3 * i + j] = vm[4 * j + i];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) Fm[4 * i + j] = pm[4 * j + i];
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        for (int k = 0; k < 3; k++) dmean3D[3 * i + k] = 0;
        for (int k = 0; k < 6; k++) dcov3D[6 * i + k] = 0;
        for (int k = 0; k < 3 * M; k++) dsh[(size_t)3 * M * i + k] = 0;
        if (scales) {
            for (int k = 0; k < 3; k++) dscale[3 * i + k] = 0;
            for (int k = 0; k < 4; k++) drot[4 * i + k] = 0;
        }
        dopacity[i] = 0;
        if (!(radii[i] > 0)) continue;
        const double mu[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
        const double* a9 = acc + 9 * (size_t)i;
        /* ---- camera-space point, clamp decisions in fp32 as the kernels take them (forward.cu:168-176) ---- */
        double t[3];
        for (int r = 0; r < 3; r++) t[r] = Rw[3 * r] * mu[0] + Rw[3 * r + 1] * mu[1] + Rw[3 * r + 2] * mu[2] + vm[12 + r];
        float tf[3];
        for (int r = 0; r < 3; r++)
            tf[r] = vm[r] * means[3 * i] + vm[4 + r] * means[3 * i + 1] + vm[8 + r] * means[3 * i + 2] + vm[12 + r];
        const float txtz = tf[0] / tf[2], tytz = tf[1] / tf[2];
        const int clx = (txtz < -(1.3f * tanx) || txtz > (1.3f * tanx)), cly = (tytz < -(1.3f * tany) || tytz > (1.3f * tany));
        double txc = t[0], tyc = t[1];
        if (clx) txc = (txtz < 0 ? -limx : limx) * t[2];
        if (cly) tyc = (tytz < 0 ? -limy : limy) * t[2];
        const double tz = t[2];
        const double J[6] = {fx / tz, 0, -fx * txc / (tz * tz), 0, fy / tz, -fy * tyc / (tz * tz)};
        double A[6];   /* A = J Rw, 2x3 */
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++) A[3 * r + c] = J[3 * r] * Rw[c] + J[3 * r + 1] * Rw[3 + c] + J[3 * r + 2] * Rw[6 + c];
        const double* c6 = cov3Ds + 6 * (size_t)i;
        const double S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
        double AS[6];
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++) AS[3 * r + c] = A[3 * r] * S[c] + A[3 * r + 1] * S[3 + c] + A[3 * r + 2] * S[6 + c];
        const double ca = AS[0] * A[0] + AS[1] * A[1] + AS[2] * A[2] + 0.3;
        const double cb = AS[0] * A[3] + AS[1] * A[4] + AS[2] * A[5];
        const double cc = AS[3] * A[3] + AS[4] * A[4] + AS[5] * A[5] + 0.3;
        /* ---- conic = Sigma'^-1: dL/dSigma' = -k adj(Sigma') Gq adj(Sigma'), k = 1 / (det^2 + 1e-7) (backward.cu:234) ---- */
        const double det = ca * cc - cb * cb;
        const double kk = 1.0 / (det * det + (g_pure ? 0.0 : 1e-7));
        const double Gq[4] = {a9[2], a9[3], a9[3], a9[4]};   /* symmetric matrix gradient w.r.t. the conic */
        const double adj[4] = {cc, -cb, -cb, ca};
        double t1[4], Gc[4];
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 2; c++) t1[2 * r + c] = adj[2 * r] * Gq[c] + adj[2 * r + 1] * Gq[2 + c];
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 2; c++) Gc[2 * r + c] = -kk * (t1[2 * r] * adj[c] + t1[2 * r + 1] * adj[2 + c]);
        /* ---- Sigma' = A Sigma A^T + 0.3 I ---- */
        double GA[6];   /* Gc A, 2x3 */
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++) GA[3 * r + c] = Gc[2 * r] * A[c] + Gc[2 * r + 1] * A[3 + c];
        double Gs[9];   /* A^T Gc A: matrix gradient w.r.t. Sigma */
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Gs[3 * r + c] = A[r] * GA[c] + A[3 + r] * GA[3 + c];
        double* dc = dcov3D + 6 * (size_t)i;
        dc[0] = Gs[0], dc[3] = Gs[4], dc[5] = Gs[8];
        dc[1] = Gs[1] + Gs[3], dc[2] = Gs[2] + Gs[6], dc[4] = Gs[5] + Gs[7];
        double dA[6];   /* 2 Gc A Sigma */
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++) dA[3 * r + c] = 2 * (GA[3 * r] * S[c] + GA[3 * r + 1] * S[3 + c] + GA[3 * r + 2] * S[6 + c]);
        double dJ[6];   /* dA Rw^T */
        for (int r = 0; r < 2; r++)
            for (int c = 0; c < 3; c++) dJ[3 * r + c] = dA[3 * r] * Rw[3 * c] + dA[3 * r + 1] * Rw[3 * c + 1] + dA[3 * r + 2] * Rw[3 * c + 2];
        /* J00 = fx/tz, J02 = -fx tx/tz^2, J11 = fy/tz, J12 = -fy ty/tz^2 */
        double dt[3];
        dt[0] = clx ? 0.0 : dJ[2] * (-fx / (tz * tz));
        dt[1] = cly ? 0.0 : dJ[5] * (-fy / (tz * tz));
        dt[2] = dJ[0] * (-fx / (tz * tz)) + dJ[4] * (-fy / (tz * tz)) + dJ[2] * (2 * fx * txc / (tz * tz * tz)) +
                dJ[5] * (2 * fy * tyc / (tz * tz * tz));
        double dmu[3];
        for (int c = 0; c < 3; c++) dmu[c] = Rw[c] * dt[0] + Rw[3 + c] * dt[1] + Rw[6 + c] * dt[2];
        /* ---- pixel mean: p = h.xy / (h.w + 1e-7), h = F (mu, 1); acc[0..1] are dL/dp (NDC units) ---- */
        double h[4];
        for (int r = 0; r < 4; r++) h[r] = Fm[4 * r] * mu[0] + Fm[4 * r + 1] * mu[1] + Fm[4 * r + 2] * mu[2] + Fm[4 * r + 3];
        const double iw = 1.0 / (h[3] + 1e-7);
        for (int c = 0; c < 3; c++)
            dmu[c] += a9[0] * (Fm[c] - h[0] * iw * Fm[12 + c]) * iw + a9[1] * (Fm[4 + c] - h[1] * iw * Fm[12 + c]) * iw;
        /* ---- colour ---- */
        if (shs) {
            const float* sh = shs + (size_t)3 * M * i;
            double* ds = dsh + (size_t)3 * M * i;
            const double v[3] = {mu[0] - (double)campos[0], mu[1] - (double)campos[1], mu[2] - (double)campos[2]};
            const double len = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const double d[3] = {v[0] / len, v[1] / len, v[2] / len};
            double Y[16], dY[16][3], dRGB[3];
            const int deg = degs[i], K = (deg + 1) * (deg + 1);
            sh_basis_grad(deg, d[0], d[1], d[2], Y, dY);
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = clamped[3 * i + ch] ? 0.0 : a9[6 + ch];
            double dd[3] = {0, 0, 0};
            for (int k = 0; k < K; k++) {
                double w = 0;
                for (int ch = 0; ch < 3; ch++) {
                    const double s = sh[3 * k + ch];
                    ds[3 * k + ch] = Y[k] * dRGB[ch] + ((k >= 1 && mult != 0) ? mult * ((s > 0) - (s < 0)) : 0.0);
                    w += s * dRGB[ch];
                }
                for (int c = 0; c < 3; c++) dd[c] += dY[k][c] * w;
            }
            const double dot = d[0] * dd[0] + d[1] * dd[1] + d[2] * dd[2];
            for (int c = 0; c < 3; c++) dmu[c] += (dd[c] - d[c] * dot) / len;
        }
        for (int c = 0; c < 3; c++) dmean3D[3 * i + c] = dmu[c];
        /* ---- Sigma = R S^2 R^T ---- */
        if (scales) {
            const double q[4] = {rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]};
            const double s[3] = {(double)mod * scales[3 * i], (double)mod * scales[3 * i + 1], (double)mod * scales[3 * i + 2]};
            double R[9], GR[9], dRq[4][9];
            quat_R(q, R);
            double Gsym[9];
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) Gsym[3 * r + c] = 0.5 * (Gs[3 * r + c] + Gs[3 * c + r]);
            mat3_mul(Gsym, R, GR);
            for (int k = 0; k < 3; k++) {
                const double rtgr = R[k] * GR[k] + R[3 + k] * GR[3 + k] + R[6 + k] * GR[6 + k];
                dscale[3 * i + k] = 2 * s[k] * rtgr;   /* w.r.t. the modifier-scaled scale (backward.cu:355-358) */
            }
            quat_dR(q, dRq);
            for (int n = 0; n < 4; n++) {
                double sum = 0;
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) sum += 2 * GR[3 * r + c] * s[c] * s[c] * dRq[n][3 * r + c];
                drot[4 * i + n] = sum;
            }
        }
        const double o = conic_op[4 * i + 3];
        dopacity[i] = a9[5] * o * (1.0 - o);
    }
}
