/*
 * mvf_oracle.c -- CPU restatement of the Mono-ViFI view-synthesis + photometric-loss
 * hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may build, load or call it.  The product path
 * (mono-vifi_amd/) never links or imports it and has no CPU fallback.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit (integer
 * sampling indices, depth, cam points, grid, SSIM / reprojection maps, argmin) or to
 * the stated tolerance (bilinear values, reductions, gradients) against golden vectors
 * captured by importing the reference itself (tests/golden/make_golden.py ->
 * tests/golden/g*.npz; checked by tests/test_oracle_golden.py).
 *
 * Arithmetic contract ("exact mode"): plain IEEE fp32, no FMA
 * contraction (build with -ffp-contract=off), explicit fmaf() only where the reference's
 * batched matmul accumulates (k-sequential chain seeded by a multiply), true divides.
 *
 * Each function cites the reference lines it restates (paths relative to
 * /root/reference).  Nothing here is copied from the reference: the reference is
 * ~30 lines of PyTorch tensor expressions; this is their per-pixel scalar meaning.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MVFO_API __attribute__((visibility("default")))

#ifdef _OPENMP
#include <omp.h>
#endif
/* number of OpenMP threads the timing leg of bench.py uses (0 = leave as is) */
MVFO_API int mvfo_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------
 * disp -> depth.  reference: layers.py:16-25 (called from train.py:961).
 * min_disp = (float)(1/max_depth), range = (float)(1/min_depth - 1/max_depth) are
 * rounded to fp32 by the caller exactly as the Python scalars are when they meet the
 * fp32 tensor; multiply then add (two roundings), then a true reciprocal.
 * ---------------------------------------------------------------------------------- */
static inline float depth_of(float disp, float min_disp, float range)
{
    float scaled = min_disp + range * disp;
    return 1.0f / scaled;
}

MVFO_API void mvfo_disp_to_depth(const float *disp, float *scaled, float *depth, long n,
                                 float min_disp, float range)
{
    for (long i = 0; i < n; ++i) {
        float s = min_disp + range * disp[i];
        if (scaled) scaled[i] = s;
        if (depth) depth[i] = 1.0f / s;
    }
}

/* ------------------------------------------------------------------------------------
 * BackprojectDepth.forward.  reference: layers.py:192-197 (pix_coords built 178-190:
 * rows [x, y, 1], flattened row-major idx = y*W + x).
 * inv_K[:, :3, :3] @ pix is a k-sequential fmaf chain seeded by a multiply.
 * cam: [B,4,N]
 * ---------------------------------------------------------------------------------- */
static inline void ray_of(const float *iK /*4x4*/, float x, float y, float r[3])
{
    for (int i = 0; i < 3; ++i) {
        float a = iK[i * 4 + 0] * x;
        a = fmaf(iK[i * 4 + 1], y, a);
        a = fmaf(iK[i * 4 + 2], 1.0f, a);
        r[i] = a;
    }
}

MVFO_API void mvfo_backproject(const float *depth, const float *invK, float *cam,
                               int B, int H, int W)
{
    long N = (long)H * W;
    for (int b = 0; b < B; ++b) {
        const float *iK = invK + b * 16;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                long i = (long)y * W + x;
                float r[3];
                ray_of(iK, (float)x, (float)y, r);
                float d = depth[b * N + i];
                cam[(b * 4 + 0) * N + i] = d * r[0];
                cam[(b * 4 + 1) * N + i] = d * r[1];
                cam[(b * 4 + 2) * N + i] = d * r[2];
                cam[(b * 4 + 3) * N + i] = 1.0f;
            }
    }
}

/* P = (K @ T)[:3]   reference: layers.py:212.
 * A 4x4 @ 4x4 batched product is below ATen's small-matrix threshold (rows*cols*k < 400),
 * so the reference runs its plain C++ loop there: products and sums rounded separately,
 * k in order -- NOT an fma chain (unlike the two [3xk]@[kxN] products, which go to the
 * BLAS kernel).  Pinned by tests/golden/g1_*.npz key "P*". */
static inline void proj_matrix(const float *K, const float *T, float P[12])
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 4; ++j) {
            float a = K[i * 4 + 0] * T[0 * 4 + j];
            a = a + K[i * 4 + 1] * T[1 * 4 + j];
            a = a + K[i * 4 + 2] * T[2 * 4 + j];
            a = a + K[i * 4 + 3] * T[3 * 4 + j];
            P[i * 4 + j] = a;
        }
}

MVFO_API void mvfo_proj_matrix(const float *K, const float *T, float *P, int B)
{
    for (int b = 0; b < B; ++b) proj_matrix(K + b * 16, T + b * 16, P + b * 12);
}

/* one point through P, the perspective divide and the [-1,1] normalisation.
 * reference: layers.py:214-221 */
static inline void project_point(const float P[12], const float X[4], float eps,
                                 float wm1, float hm1, float *gx, float *gy)
{
    float c[3];
    for (int i = 0; i < 3; ++i) {
        float a = P[i * 4 + 0] * X[0];
        a = fmaf(P[i * 4 + 1], X[1], a);
        a = fmaf(P[i * 4 + 2], X[2], a);
        a = fmaf(P[i * 4 + 3], X[3], a);
        c[i] = a;
    }
    float z = c[2] + eps;
    float u = c[0] / z;
    float v = c[1] / z;
    u = u / wm1;
    v = v / hm1;
    *gx = (u - 0.5f) * 2.0f;
    *gy = (v - 0.5f) * 2.0f;
}

/* Project3D.forward: cam [B,4,N] -> pix [B,H,W,2].  reference: layers.py:211-222 */
MVFO_API void mvfo_project(const float *cam, const float *K, const float *T, float *pix,
                           int B, int H, int W, float eps)
{
    long N = (long)H * W;
    float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    for (int b = 0; b < B; ++b) {
        float P[12];
        proj_matrix(K + b * 16, T + b * 16, P);
#pragma omp parallel for schedule(static)
        for (long i = 0; i < N; ++i) {
            float X[4] = {cam[(b * 4 + 0) * N + i], cam[(b * 4 + 1) * N + i],
                          cam[(b * 4 + 2) * N + i], cam[(b * 4 + 3) * N + i]};
            float gx, gy;
            project_point(P, X, eps, wm1, hm1, &gx, &gy);
            pix[(b * N + i) * 2 + 0] = gx;
            pix[(b * N + i) * 2 + 1] = gy;
        }
    }
}

/* ------------------------------------------------------------------------------------
 * F.grid_sample(img, grid, mode=bilinear, padding_mode="border", align_corners=True)
 * reference call site: train.py:966-969.  The arithmetic is ATen's (third-party,
 * torch pinned 1.11.0+cu113 in README.md:57): unnormalise ((g+1)/2)*(size-1), clip to
 * [0,size-1], floor, weights w = i - floor(i), e = 1 - w.  x0,y0 are the bit-exact
 * integers; values are tolerance-checked.
 * ---------------------------------------------------------------------------------- */
typedef struct {
    int x0, y0, x1, y1;
    float wx, wy;     /* fractional parts */
    int inx, iny;     /* 1 when the coordinate was strictly inside (grad passes) */
} tap_t;

static inline float clipf(float v, float hi)
{
    /* NaN -> 0 (never indexes out of bounds) */
    float a = (v > 0.0f) ? v : 0.0f;
    return (a < hi) ? a : hi;
}

static inline tap_t tap_of(float gx, float gy, int H, int W)
{
    tap_t t;
    float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    float ix = ((gx + 1.0f) / 2.0f) * wm1;
    float iy = ((gy + 1.0f) / 2.0f) * hm1;
    t.inx = (ix > 0.0f) && (ix < wm1);
    t.iny = (iy > 0.0f) && (iy < hm1);
    ix = clipf(ix, wm1);
    iy = clipf(iy, hm1);
    float fx = floorf(ix), fy = floorf(iy);
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    t.x1 = (t.x0 + 1 < W) ? t.x0 + 1 : W - 1;
    t.y1 = (t.y0 + 1 < H) ? t.y0 + 1 : H - 1;
    t.wx = ix - fx;
    t.wy = iy - fy;
    return t;
}

static inline float bilerp(const float *im, int W, const tap_t *t)
{
    float w = t->wx, e = 1.0f - w, n = t->wy, s = 1.0f - n;
    float nw = im[t->y0 * W + t->x0], ne = im[t->y0 * W + t->x1];
    float sw = im[t->y1 * W + t->x0], se = im[t->y1 * W + t->x1];
    return nw * (s * e) + ne * (s * w) + sw * (n * e) + se * (n * w);
}

MVFO_API void mvfo_grid_sample(const float *img, const float *grid, float *out, int32_t *x0,
                               int32_t *y0, int B, int C, int H, int W)
{
    long N = (long)H * W;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < N; ++i) {
            tap_t t = tap_of(grid[(b * N + i) * 2], grid[(b * N + i) * 2 + 1], H, W);
            if (x0) x0[b * N + i] = t.x0;
            if (y0) y0[b * N + i] = t.y0;
            if (out)
                for (int c = 0; c < C; ++c)
                    out[((long)(b * C + c)) * N + i] = bilerp(img + ((long)(b * C + c)) * N, W, &t);
        }
    }
}

/* grad of grid_sample w.r.t. the grid only (images never require grad in this trainer,
 * SURVEY.md section 8a row a4).  ggrid [B,H,W,2]. */
static inline void bilerp_grad(const float *im, int W, const tap_t *t, float *dx, float *dy)
{
    float w = t->wx, e = 1.0f - w, n = t->wy, s = 1.0f - n;
    float nw = im[t->y0 * W + t->x0], ne = im[t->y0 * W + t->x1];
    float sw = im[t->y1 * W + t->x0], se = im[t->y1 * W + t->x1];
    *dx = (ne - nw) * s + (se - sw) * n;
    *dy = (sw - nw) * e + (se - ne) * w;
}

MVFO_API void mvfo_grid_sample_bwd(const float *img, const float *grid, const float *gout,
                                   float *ggrid, int B, int C, int H, int W)
{
    long N = (long)H * W;
    float sx = (float)(W - 1) / 2.0f, sy = (float)(H - 1) / 2.0f;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < N; ++i) {
            tap_t t = tap_of(grid[(b * N + i) * 2], grid[(b * N + i) * 2 + 1], H, W);
            float gx = 0.f, gy = 0.f;
            for (int c = 0; c < C; ++c) {
                float dx, dy;
                bilerp_grad(img + ((long)(b * C + c)) * N, W, &t, &dx, &dy);
                float g = gout[((long)(b * C + c)) * N + i];
                gx += g * dx;
                gy += g * dy;
            }
            ggrid[(b * N + i) * 2 + 0] = t.inx ? gx * sx : 0.0f;
            ggrid[(b * N + i) * 2 + 1] = t.iny ? gy * sy : 0.0f;
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Trainer.generate_images_pred for ONE source: disp -> depth -> backproject -> project
 * -> grid_sample, fused per pixel.  reference: train.py:956-971.
 * Optional outputs: pix [B,H,W,2], x0/y0 int32 [B,H,W], warped [B,3,H,W].
 * ---------------------------------------------------------------------------------- */
MVFO_API void mvfo_warp_fwd(const float *disp, const float *invK, const float *K, const float *T,
                            const float *src, float *warped, float *pix, int32_t *x0, int32_t *y0,
                            int B, int H, int W, float min_disp, float range, float eps)
{
    long N = (long)H * W;
    float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    for (int b = 0; b < B; ++b) {
        float P[12];
        proj_matrix(K + b * 16, T + b * 16, P);
        const float *iK = invK + b * 16;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                long i = (long)y * W + x;
                float r[3], X[4], gx, gy;
                ray_of(iK, (float)x, (float)y, r);
                float d = depth_of(disp[b * N + i], min_disp, range);
                X[0] = d * r[0]; X[1] = d * r[1]; X[2] = d * r[2]; X[3] = 1.0f;
                project_point(P, X, eps, wm1, hm1, &gx, &gy);
                if (pix) { pix[(b * N + i) * 2] = gx; pix[(b * N + i) * 2 + 1] = gy; }
                tap_t t = tap_of(gx, gy, H, W);
                if (x0) x0[b * N + i] = t.x0;
                if (y0) y0[b * N + i] = t.y0;
                if (warped)
                    for (int c = 0; c < 3; ++c)
                        warped[((long)(b * 3 + c)) * N + i] = bilerp(src + ((long)(b * 3 + c)) * N, W, &t);
            }
    }
}

/* Backward of the fused warp for one source: grad_warped [B,3,H,W] ->
 * grad_disp [B,1,H,W] (ACCUMULATED into), grad_T [B,4,4] (overwritten).
 * Chain: bilinear adjoint w.r.t. (ix,iy) (zero where clipped; x(W-1)/2) -> normalise
 * -> perspective divide -> P -> (X = depth*ray) -> depth -> disp; grad_P = sum g_c X^T,
 * grad_T = K^T [grad_P; 0].  Autograd of reference train.py:956-971. */
MVFO_API void mvfo_warp_bwd(const float *disp, const float *invK, const float *K, const float *T,
                            const float *src, const float *gwarped, float *gdisp, float *gT,
                            int B, int H, int W, float min_disp, float range, float eps)
{
    long N = (long)H * W;
    float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    float sx = wm1 / 2.0f, sy = hm1 / 2.0f;
    for (int b = 0; b < B; ++b) {
        float P[12];
        proj_matrix(K + b * 16, T + b * 16, P);
        const float *iK = invK + b * 16;
        double gP[12];
        for (int k = 0; k < 12; ++k) gP[k] = 0.0;
#pragma omp parallel
        {
            double lP[12];
            for (int k = 0; k < 12; ++k) lP[k] = 0.0;
#pragma omp for schedule(static)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    long i = (long)y * W + x;
                    float r[3], X[4], c[3];
                    ray_of(iK, (float)x, (float)y, r);
                    float scaled = min_disp + range * disp[b * N + i];
                    float d = 1.0f / scaled;
                    X[0] = d * r[0]; X[1] = d * r[1]; X[2] = d * r[2]; X[3] = 1.0f;
                    for (int q = 0; q < 3; ++q) {
                        float a = P[q * 4 + 0] * X[0];
                        a = fmaf(P[q * 4 + 1], X[1], a);
                        a = fmaf(P[q * 4 + 2], X[2], a);
                        a = fmaf(P[q * 4 + 3], X[3], a);
                        c[q] = a;
                    }
                    float z = c[2] + eps;
                    float u = c[0] / z, v = c[1] / z;
                    float gx = (u / wm1 - 0.5f) * 2.0f, gy = (v / hm1 - 0.5f) * 2.0f;
                    tap_t t = tap_of(gx, gy, H, W);
                    float gix = 0.f, giy = 0.f;
                    for (int ch = 0; ch < 3; ++ch) {
                        float dx, dy;
                        bilerp_grad(src + ((long)(b * 3 + ch)) * N, W, &t, &dx, &dy);
                        float g = gwarped[((long)(b * 3 + ch)) * N + i];
                        gix += g * dx;
                        giy += g * dy;
                    }
                    /* d(ix)/d(gx) = (W-1)/2 ; d(gx)/d(u) = 2/(W-1) */
                    float ggx = t.inx ? gix * sx : 0.0f;
                    float ggy = t.iny ? giy * sy : 0.0f;
                    float gu = ggx * 2.0f / wm1;
                    float gv = ggy * 2.0f / hm1;
                    float gc[3];
                    gc[0] = gu / z;
                    gc[1] = gv / z;
                    gc[2] = -(gu * u + gv * v) / z;
                    float gd = 0.f;
                    for (int j = 0; j < 3; ++j) {
                        float gXj = gc[0] * P[0 * 4 + j] + gc[1] * P[1 * 4 + j] + gc[2] * P[2 * 4 + j];
                        gd += gXj * r[j];
                    }
                    for (int q = 0; q < 3; ++q)
                        for (int j = 0; j < 4; ++j) lP[q * 4 + j] += (double)gc[q] * X[j];
                    /* depth = 1/scaled ; scaled = min + range*disp */
                    gdisp[b * N + i] += -gd * d * d * range;
                }
#pragma omp critical
            for (int k = 0; k < 12; ++k) gP[k] += lP[k];
        }
        const float *Kb = K + b * 16;
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) {
                double a = 0.0;
                for (int q = 0; q < 3; ++q) a += (double)Kb[q * 4 + k] * gP[q * 4 + j];
                gT[b * 16 + k * 4 + j] = (float)a;
            }
    }
}

/* ------------------------------------------------------------------------------------
 * SSIM.  reference: layers.py:261-290.  ReflectionPad2d(1); AvgPool2d(3,1) = the 9 taps
 * summed sequentially in row-major order from 0, then one true divide by 9; literal
 * expression order; no contraction.  out [B,C,H,W] in [0,1].
 * ---------------------------------------------------------------------------------- */
static inline int refl(int j, int n)
{
    if (j < 0) return -j;
    if (j >= n) return 2 * (n - 1) - j;
    return j;
}

typedef struct { float mu_x, mu_y, exx, eyy, exy; } win_t;

static inline win_t window_stats(const float *x, const float *y, int H, int W, int py, int px)
{
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
    for (int dy = -1; dy <= 1; ++dy) {
        int yy = refl(py + dy, H);
        for (int dx = -1; dx <= 1; ++dx) {
            int xx = refl(px + dx, W);
            float a = x[yy * W + xx], b = y[yy * W + xx];
            sx = sx + a;
            sy = sy + b;
            sxx = sxx + a * a;
            syy = syy + b * b;
            sxy = sxy + a * b;
        }
    }
    win_t w = {sx / 9.0f, sy / 9.0f, sxx / 9.0f, syy / 9.0f, sxy / 9.0f};
    return w;
}

static inline float ssim_from(const win_t *w, float C1, float C2, float *raw_out)
{
    float sigma_x = w->exx - w->mu_x * w->mu_x;
    float sigma_y = w->eyy - w->mu_y * w->mu_y;
    float sigma_xy = w->exy - w->mu_x * w->mu_y;
    float n = (2.0f * w->mu_x * w->mu_y + C1) * (2.0f * sigma_xy + C2);
    float d = (w->mu_x * w->mu_x + w->mu_y * w->mu_y + C1) * (sigma_x + sigma_y + C2);
    float raw = (1.0f - n / d) / 2.0f;
    if (raw_out) *raw_out = raw;
    float c = raw < 0.0f ? 0.0f : raw;
    return c > 1.0f ? 1.0f : c;
}

static inline float ssim_c1(void) { return (float)(0.01 * 0.01); }
static inline float ssim_c2(void) { return (float)(0.03 * 0.03); }

MVFO_API void mvfo_ssim(const float *x, const float *y, float *out, int B, int C, int H, int W)
{
    long N = (long)H * W;
    float C1 = ssim_c1(), C2 = ssim_c2();
#pragma omp parallel for schedule(static) collapse(2)
    for (int bc = 0; bc < B * C; ++bc)
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                win_t w = window_stats(x + bc * N, y + bc * N, H, W, py, px);
                out[bc * N + (long)py * W + px] = ssim_from(&w, C1, C2, 0);
            }
}

/* d SSIM / d (mu_x, Exx, Exy) and d/d(mu_y, Eyy) of the un-clamped map; zero when the
 * clamp is active (torch clamp passes grad on the closed interval). */
typedef struct { float dmux, dexx, dexy, dmuy, deyy; } dwin_t;

static inline dwin_t ssim_partials(const win_t *w, float C1, float C2)
{
    float mx = w->mu_x, my = w->mu_y;
    float sigma_x = w->exx - mx * mx, sigma_y = w->eyy - my * my, sigma_xy = w->exy - mx * my;
    float A1 = 2.0f * mx * my + C1, A2 = 2.0f * sigma_xy + C2;
    float B1 = mx * mx + my * my + C1, B2 = sigma_x + sigma_y + C2;
    float n = A1 * A2, d = B1 * B2;
    float raw = (1.0f - n / d) / 2.0f;
    dwin_t g = {0, 0, 0, 0, 0};
    if (!(raw >= 0.0f && raw <= 1.0f)) return g;
    /* raw = (1 - n/d)/2 : draw = -0.5*(dn/d - n*dd/d^2) */
    float inv_d = 1.0f / d;
    float kn = -0.5f * inv_d;            /* d raw / d n */
    float kd = 0.5f * n * inv_d * inv_d; /* d raw / d d */
    /* n: dA1/dmx = 2my, dA2/dmx = -2my (via sigma_xy), dA2/dExy = 2 */
    float dn_dmx = 2.0f * my * A2 - 2.0f * my * A1;
    float dn_dmy = 2.0f * mx * A2 - 2.0f * mx * A1;
    float dn_dexy = 2.0f * A1;
    /* d: dB1/dmx = 2mx, dB2/dmx = -2mx (via sigma_x), dB2/dExx = 1 */
    float dd_dmx = 2.0f * mx * B2 - 2.0f * mx * B1;
    float dd_dmy = 2.0f * my * B2 - 2.0f * my * B1;
    g.dmux = kn * dn_dmx + kd * dd_dmx;
    g.dmuy = kn * dn_dmy + kd * dd_dmy;
    g.dexy = kn * dn_dexy;
    g.dexx = kd * B1;
    g.deyy = kd * B1;
    return g;
}

/* Backward of SSIM: gout [B,C,H,W] -> gx, gy (either may be NULL), overwritten. */
MVFO_API void mvfo_ssim_bwd(const float *x, const float *y, const float *gout, float *gx, float *gy,
                            int B, int C, int H, int W)
{
    long N = (long)H * W;
    float C1 = ssim_c1(), C2 = ssim_c2();
    if (gx) memset(gx, 0, sizeof(float) * B * C * N);
    if (gy) memset(gy, 0, sizeof(float) * B * C * N);
#pragma omp parallel for schedule(static)
    for (int bc = 0; bc < B * C; ++bc) {
        const float *xb = x + bc * N, *yb = y + bc * N;
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                float g = gout[bc * N + (long)py * W + px];
                if (g == 0.0f) continue;
                win_t w = window_stats(xb, yb, H, W, py, px);
                dwin_t p = ssim_partials(&w, C1, C2);
                for (int dy = -1; dy <= 1; ++dy) {
                    int yy = refl(py + dy, H);
                    for (int dx = -1; dx <= 1; ++dx) {
                        int xx = refl(px + dx, W);
                        float a = xb[yy * W + xx], bq = yb[yy * W + xx];
                        if (gx) gx[bc * N + yy * W + xx] += g * (p.dmux + 2.0f * a * p.dexx + bq * p.dexy) / 9.0f;
                        if (gy) gy[bc * N + yy * W + xx] += g * (p.dmuy + 2.0f * bq * p.deyy + a * p.dexy) / 9.0f;
                    }
                }
            }
    }
}

/* ------------------------------------------------------------------------------------
 * Trainer.compute_reprojection_loss.  reference: train.py:973-985.
 * l1 = ((|t-p|_0 + |t-p|_1) + |t-p|_2) / 3 ; ssim likewise ; 0.85*ssim + 0.15*l1.
 * pred, tgt [B,3,H,W] -> out [B,1,H,W]
 * ---------------------------------------------------------------------------------- */
static inline float reproj_px(const float *pred, const float *tgt, long N, int H, int W, int py,
                              int px, int no_ssim, float C1, float C2)
{
    long i = (long)py * W + px;
    float a0 = fabsf(tgt[i] - pred[i]);
    float a1 = fabsf(tgt[N + i] - pred[N + i]);
    float a2 = fabsf(tgt[2 * N + i] - pred[2 * N + i]);
    float l1 = ((a0 + a1) + a2) / 3.0f;
    if (no_ssim) return l1;
    float s[3];
    for (int c = 0; c < 3; ++c) {
        win_t w = window_stats(pred + c * N, tgt + c * N, H, W, py, px);
        s[c] = ssim_from(&w, C1, C2, 0);
    }
    float ss = ((s[0] + s[1]) + s[2]) / 3.0f;
    return 0.85f * ss + 0.15f * l1;
}

MVFO_API void mvfo_reprojection(const float *pred, const float *tgt, float *out, int B, int H, int W,
                                int no_ssim)
{
    long N = (long)H * W;
    float C1 = ssim_c1(), C2 = ssim_c2();
#pragma omp parallel for schedule(static) collapse(2)
    for (int b = 0; b < B; ++b)
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px)
                out[b * N + (long)py * W + px] =
                    reproj_px(pred + b * 3 * N, tgt + b * 3 * N, N, H, W, py, px, no_ssim, C1, C2);
}

/* ------------------------------------------------------------------------------------
 * get_smooth_loss on the mean-normalised disparity.
 * reference: train.py:1044-1047 + layers.py:231-242.
 * returns smooth = mean_x + mean_y ; also the per-image mean disp (mean over H of the
 * column... reference takes mean(2) then mean(3); equal up to rounding).
 * ---------------------------------------------------------------------------------- */
MVFO_API double mvfo_smooth(const float *disp, const float *img, int B, int H, int W, int normalise,
                            float *mean_out)
{
    long N = (long)H * W;
    double sx = 0.0, sy = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *d = disp + b * N;
        float denom = 1.0f;
        if (normalise) {
            double m = 0.0;
            for (long i = 0; i < N; ++i) m += d[i];
            float mean = (float)(m / (double)N);
            if (mean_out) mean_out[b] = mean;
            denom = mean + 1e-7f;
        }
        const float *im = img + b * 3 * N;
        double lx = 0.0, ly = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : lx, ly)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                long i = (long)y * W + x;
                float nd = d[i] / denom;
                if (x + 1 < W) {
                    float gd = fabsf(nd - d[i + 1] / denom);
                    float gi = ((fabsf(im[i] - im[i + 1]) + fabsf(im[N + i] - im[N + i + 1])) +
                                fabsf(im[2 * N + i] - im[2 * N + i + 1])) / 3.0f;
                    lx += (double)(gd * expf(-gi));
                }
                if (y + 1 < H) {
                    float gd = fabsf(nd - d[i + W] / denom);
                    float gi = ((fabsf(im[i] - im[i + W]) + fabsf(im[N + i] - im[N + i + W])) +
                                fabsf(im[2 * N + i] - im[2 * N + i + W])) / 3.0f;
                    ly += (double)(gd * expf(-gi));
                }
            }
        sx += lx;
        sy += ly;
    }
    return sx / ((double)B * H * (W - 1)) + sy / ((double)B * (H - 1) * W);
}

/* grad of `scale * smooth(disp / (mean_hw(disp)+1e-7), img)` (normalise=1) or of
 * `scale * smooth(disp, img)` (normalise=0) w.r.t. disp, ACCUMULATED into gdisp. */
MVFO_API void mvfo_smooth_bwd(const float *disp, const float *img, float *gdisp, int B, int H, int W,
                              int normalise, float scale)
{
    long N = (long)H * W;
    double cx = (double)scale / ((double)B * H * (W - 1));
    double cy = (double)scale / ((double)B * (H - 1) * W);
    float *gn = (float *)malloc(sizeof(float) * N);
    for (int b = 0; b < B; ++b) {
        const float *d = disp + b * N;
        const float *im = img + b * 3 * N;
        float denom = 1.0f;
        if (normalise) {
            double m = 0.0;
            for (long i = 0; i < N; ++i) m += d[i];
            denom = (float)(m / (double)N) + 1e-7f;
        }
        /* Gather form of the scatter `gn[i] += g; gn[i+1] -= g` (x term of pixel i) and
         * `gn[i] += g; gn[i+W] -= g` (y term): pixel i receives, in the scatter's own order,
         * -gy(i-W), -gx(i-1), +gx(i), +gy(i) -- the same additions in the same order, so the same
         * bits, and every pixel is independent (parallel over rows; the scatter was serial). */
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                long i = (long)y * W + x;
                float acc = 0.0f;
                if (y >= 1) {           /* y term of pixel i - W */
                    long j = i - W;
                    float df = d[j] / denom - d[i] / denom;
                    float gi = ((fabsf(im[j] - im[i]) + fabsf(im[N + j] - im[N + i])) +
                                fabsf(im[2 * N + j] - im[2 * N + i])) / 3.0f;
                    float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
                    acc -= (float)cy * expf(-gi) * sg;
                }
                if (x >= 1) {           /* x term of pixel i - 1 */
                    long j = i - 1;
                    float df = d[j] / denom - d[i] / denom;
                    float gi = ((fabsf(im[j] - im[i]) + fabsf(im[N + j] - im[N + i])) +
                                fabsf(im[2 * N + j] - im[2 * N + i])) / 3.0f;
                    float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
                    acc -= (float)cx * expf(-gi) * sg;
                }
                float nd = d[i] / denom;
                if (x + 1 < W) {
                    float df = nd - d[i + 1] / denom;
                    float gi = ((fabsf(im[i] - im[i + 1]) + fabsf(im[N + i] - im[N + i + 1])) +
                                fabsf(im[2 * N + i] - im[2 * N + i + 1])) / 3.0f;
                    float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
                    acc += (float)cx * expf(-gi) * sg;
                }
                if (y + 1 < H) {
                    float df = nd - d[i + W] / denom;
                    float gi = ((fabsf(im[i] - im[i + W]) + fabsf(im[N + i] - im[N + i + W])) +
                                fabsf(im[2 * N + i] - im[2 * N + i + W])) / 3.0f;
                    float sg = (df > 0.f) ? 1.f : ((df < 0.f) ? -1.f : 0.f);
                    acc += (float)cy * expf(-gi) * sg;
                }
                gn[i] = acc;
            }
        if (normalise) {
            /* nd = d/denom, denom = mean+eps : g_d = gn/denom - (sum gn*d)/denom^2 / N */
            double dot = 0.0;
            for (long i = 0; i < N; ++i) dot += (double)gn[i] * d[i];
            float corr = (float)(dot / ((double)denom * denom) / (double)N);
#pragma omp parallel for schedule(static)
            for (long i = 0; i < N; ++i) gdisp[b * N + i] += gn[i] / denom - corr;
        } else {
#pragma omp parallel for schedule(static)
            for (long i = 0; i < N; ++i) gdisp[b * N + i] += gn[i];
        }
    }
    free(gn);
}

/* ------------------------------------------------------------------------------------
 * Trainer.compute_losses_base, forward.  reference: train.py:987-1051.
 *   warped [S][B,3,H,W], src [S][B,3,H,W] (identity candidates), noise: standard-normal
 *   draw [B,n_id,H,W] (n_id = 1 if avg else S) scaled by 1e-5 here (train.py:1023-1024),
 *   mask_rec [B,1,H,W] or NULL.
 * outputs (any may be NULL): rp [B,S,H,W], idl [B,S,H,W], to_opt [B,H,W], idx int32 [B,H,W]
 *   (-1 when a single candidate), returns the photometric mean; *smooth_out the
 *   smoothness term; loss = photometric + smoothness_weight*smooth is formed by the caller.
 * ---------------------------------------------------------------------------------- */
#define MVFO_NO_SSIM 1
#define MVFO_AVG_REPROJ 2
#define MVFO_NO_AUTOMASK 4

MVFO_API double mvfo_losses_base_fwd(const float *tgt, const float *const *warped,
                                     const float *const *src, const float *noise,
                                     const float *mask_rec, int S, int flags, float *rp_out,
                                     float *idl_out, float *to_opt, int32_t *idx, int B, int H, int W)
{
    long N = (long)H * W;
    int no_ssim = flags & MVFO_NO_SSIM, avg = flags & MVFO_AVG_REPROJ, automask = !(flags & MVFO_NO_AUTOMASK);
    int n_id = automask ? (avg ? 1 : S) : 0;
    float C1 = ssim_c1(), C2 = ssim_c2();
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        double lsum = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : lsum)
        for (int py = 0; py < H; ++py)
            for (int px = 0; px < W; ++px) {
                long i = (long)py * W + px;
                float cand[16];
                int nc = 0;
                float rp[8], idl[8];
                for (int k = 0; k < S; ++k) {
                    rp[k] = reproj_px(warped[k] + b * 3 * N, tgt + b * 3 * N, N, H, W, py, px, no_ssim, C1, C2);
                    if (rp_out) rp_out[((long)b * S + k) * N + i] = rp[k];
                    if (automask) {
                        idl[k] = reproj_px(src[k] + b * 3 * N, tgt + b * 3 * N, N, H, W, py, px, no_ssim, C1, C2);
                        if (idl_out) idl_out[((long)b * S + k) * N + i] = idl[k];
                    }
                }
                if (automask) {
                    if (avg) {
                        float m = idl[0];
                        for (int k = 1; k < S; ++k) m = m + idl[k];
                        m = m / (float)S;
                        cand[nc++] = m + noise[(long)b * N + i] * 0.00001f;
                    } else {
                        for (int k = 0; k < S; ++k)
                            cand[nc++] = idl[k] + noise[((long)b * S + k) * N + i] * 0.00001f;
                    }
                }
                if (avg) {
                    float m = rp[0];
                    for (int k = 1; k < S; ++k) m = m + rp[k];
                    cand[nc++] = m / (float)S;
                } else {
                    for (int k = 0; k < S; ++k) cand[nc++] = rp[k];
                }
                float best = cand[0];
                int bi = 0;
                for (int k = 1; k < nc; ++k)
                    if (cand[k] < best) { best = cand[k]; bi = k; }
                if (mask_rec) best = best * mask_rec[b * N + i];
                if (to_opt) to_opt[b * N + i] = best;
                if (idx) idx[b * N + i] = (nc > 1) ? bi : -1;
                lsum += (double)best;
            }
        total += lsum;
        (void)n_id;
    }
    return total / ((double)B * N);
}

/* Backward of compute_losses_base w.r.t. the warped images: given the argmin map and the
 * upstream scalar gradient `gloss`, write gwarped[k] [B,3,H,W] (overwritten).
 * Only pixels whose argmin selected source k (or the averaged reprojection channel)
 * contribute, through SSIM's 3x3 reflect adjoint and the L1 sign (autograd of
 * train.py:973-1043). */
MVFO_API void mvfo_losses_base_bwd(const float *tgt, const float *const *warped, const int32_t *idx,
                                   const float *mask_rec, int S, int flags, float gloss,
                                   float *const *gwarped, int B, int H, int W)
{
    long N = (long)H * W;
    int no_ssim = flags & MVFO_NO_SSIM, avg = flags & MVFO_AVG_REPROJ, automask = !(flags & MVFO_NO_AUTOMASK);
    int n_id = automask ? (avg ? 1 : S) : 0;
    float C1 = ssim_c1(), C2 = ssim_c2();
    float gpix = gloss / (float)((double)B * N);
    for (int k = 0; k < S; ++k) {
        float *gw = gwarped[k];
        memset(gw, 0, sizeof(float) * B * 3 * N);
        /* The 3x3 scatter of a pixel reaches one row up and down, so row bands of one colour (every
         * second band) never touch the same row: all (plane, band) tasks of a colour run in parallel
         * (the per-plane loop alone stopped scaling at B*3 = 36 threads).  Within a row the additions
         * keep the sequential order; a band's first row receives its upper neighbour's contribution
         * after its own instead of before (gradients are tolerance-level quantities). */
        const int BAND = 8, nband = (H + BAND - 1) / BAND;
        for (int colour = 0; colour < 2; ++colour) {
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
        for (int bc = 0; bc < B * 3; ++bc)
        for (int band = colour; band < nband; band += 2) {
            int b = bc / 3, c = bc % 3;
            const float *xb = warped[k] + bc * N, *yb = tgt + bc * N;
            int py_end = (band + 1) * BAND < H ? (band + 1) * BAND : H;
            for (int py = band * BAND; py < py_end; ++py)
                for (int px = 0; px < W; ++px) {
                    long i = (long)py * W + px;
                    int sel = idx ? idx[b * N + i] : -1;
                    float wgt;
                    if (sel < 0) wgt = avg ? 1.0f / (float)S : 1.0f;  /* single candidate */
                    else if (avg) wgt = (sel == n_id) ? 1.0f / (float)S : 0.0f;
                    else wgt = (sel == n_id + k) ? 1.0f : 0.0f;
                    if (wgt == 0.0f) continue;
                    float g = gpix * wgt;
                    if (mask_rec) g *= mask_rec[b * N + i];
                    if (g == 0.0f) continue;
                    /* L1: d|t-p|/dp = -sign(t-p), mean over 3 channels */
                    float df = yb[i] - xb[i];
                    float sg = (df > 0.f) ? -1.f : ((df < 0.f) ? 1.f : 0.f);
                    float l1w = no_ssim ? 1.0f : 0.15f;
                    gw[bc * N + i] += g * l1w * sg / 3.0f;
                    if (no_ssim) continue;
                    win_t w = window_stats(xb, yb, H, W, py, px);
                    dwin_t p = ssim_partials(&w, C1, C2);
                    float gs = g * 0.85f / 3.0f;
                    for (int dy = -1; dy <= 1; ++dy) {
                        int yy = refl(py + dy, H);
                        for (int dx = -1; dx <= 1; ++dx) {
                            int xx = refl(px + dx, W);
                            float a = xb[yy * W + xx], bq = yb[yy * W + xx];
                            gw[bc * N + yy * W + xx] += gs * (p.dmux + 2.0f * a * p.dexx + bq * p.dexy) / 9.0f;
                        }
                    }
                }
            (void)c;
        }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Pose glue.  reference: layers.py:28-103 (transformation_from_parameters,
 * get_translation_matrix, rot_from_axisangle).  axisangle, translation [B,3] -> M [B,4,4]
 * M = T*R (forward) or R^T * T(-t) (invert).  Tolerance-checked (sin/cos/norm).
 * ---------------------------------------------------------------------------------- */
static void rot_of(const float v[3], float R[16])
{
    float angle = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    float inv = angle + 1e-7f;
    float x = v[0] / inv, y = v[1] / inv, z = v[2] / inv;
    float ca = cosf(angle), sa = sinf(angle), C = 1.0f - ca;
    float xs = x * sa, ys = y * sa, zs = z * sa;
    float xC = x * C, yC = y * C, zC = z * C;
    float xyC = x * yC, yzC = y * zC, zxC = z * xC;
    memset(R, 0, sizeof(float) * 16);
    R[0] = x * xC + ca;  R[1] = xyC - zs;     R[2] = zxC + ys;
    R[4] = xyC + zs;     R[5] = y * yC + ca;  R[6] = yzC - xs;
    R[8] = zxC - ys;     R[9] = yzC + xs;     R[10] = z * zC + ca;
    R[15] = 1.0f;
}

static void mat4_mul(const float *A, const float *Bm, float *C)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            /* 4x4 @ 4x4: ATen small-matrix loop, no fma (see proj_matrix) */
            float a = A[i * 4] * Bm[j];
            for (int k = 1; k < 4; ++k) a = a + A[i * 4 + k] * Bm[k * 4 + j];
            C[i * 4 + j] = a;
        }
}

MVFO_API void mvfo_pose(const float *axisangle, const float *translation, int invert, float *M, int B)
{
    for (int b = 0; b < B; ++b) {
        float R[16], T[16], Rt[16];
        rot_of(axisangle + b * 3, R);
        float sgn = invert ? -1.0f : 1.0f;
        memset(T, 0, sizeof(T));
        T[0] = T[5] = T[10] = T[15] = 1.0f;
        T[3] = sgn * translation[b * 3 + 0];
        T[7] = sgn * translation[b * 3 + 1];
        T[11] = sgn * translation[b * 3 + 2];
        if (invert) {
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) Rt[i * 4 + j] = R[j * 4 + i];
            mat4_mul(Rt, T, M + b * 16);
        } else {
            mat4_mul(T, R, M + b * 16);
        }
    }
}

/* ------------------------------------------------------------------------------------
 * f1: IFRNet.warp / FusionModule.warp_features.  reference: networks/IFRNet.py:7-15
 * (called from networks/fusion_module.py:80-90 and IFRNet.py:215-250,428-429).
 * grid = linspace(-1,1) + flow / ((size-1)/2) ; then grid_sample(border, align_corners).
 * xs[W], ys[H]: the linspace values (torch.linspace on the CPU), passed in.
 * ---------------------------------------------------------------------------------- */
static inline tap_t flow_tap(const float *flow, const float *xs, const float *ys, int b, long i,
                             int x, int y, int H, int W)
{
    long N = (long)H * W;
    float fx = flow[((long)b * 2 + 0) * N + i], fy = flow[((long)b * 2 + 1) * N + i];
    float gx = xs[x] + fx / (((float)W - 1.0f) / 2.0f);
    float gy = ys[y] + fy / (((float)H - 1.0f) / 2.0f);
    return tap_of(gx, gy, H, W);
}

MVFO_API void mvfo_flow_warp(const float *img, const float *flow, const float *xs, const float *ys,
                             float *out, int32_t *x0, int32_t *y0, int B, int C, int H, int W)
{
    long N = (long)H * W;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                long i = (long)y * W + x;
                tap_t t = flow_tap(flow, xs, ys, b, i, x, y, H, W);
                if (x0) x0[b * N + i] = t.x0;
                if (y0) y0[b * N + i] = t.y0;
                if (out)
                    for (int c = 0; c < C; ++c)
                        out[((long)b * C + c) * N + i] = bilerp(img + ((long)b * C + c) * N, W, &t);
            }
    }
}

/* grads w.r.t. img (scatter-add, g_img zero-initialised by the caller) and flow */
MVFO_API void mvfo_flow_warp_bwd(const float *img, const float *flow, const float *xs, const float *ys,
                                 const float *gout, float *g_img, float *g_flow, int B, int C, int H,
                                 int W)
{
    long N = (long)H * W;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                long i = (long)y * W + x;
                tap_t t = flow_tap(flow, xs, ys, b, i, x, y, H, W);
                float w = t.wx, e = 1.0f - w, n = t.wy, s = 1.0f - n;
                float gx = 0.f, gy = 0.f;
                for (int c = 0; c < C; ++c) {
                    float g = gout[((long)b * C + c) * N + i];
                    if (g_img) {
                        float *gi = g_img + ((long)b * C + c) * N;
                        gi[t.y0 * W + t.x0] += g * (s * e);
                        gi[t.y0 * W + t.x1] += g * (s * w);
                        gi[t.y1 * W + t.x0] += g * (n * e);
                        gi[t.y1 * W + t.x1] += g * (n * w);
                    }
                    float dx, dy;
                    bilerp_grad(img + ((long)b * C + c) * N, W, &t, &dx, &dy);
                    gx += g * dx;
                    gy += g * dy;
                }
                if (g_flow) {
                    g_flow[((long)b * 2 + 0) * N + i] = t.inx ? gx : 0.0f;
                    g_flow[((long)b * 2 + 1) * N + i] = t.iny ? gy : 0.0f;
                }
            }
}

/* ------------------------------------------------------------------------------------
 * f2: Trainer.compute_SI_log_depth_loss.  reference: train.py:924-941.
 * returns the loss; g_pred / g_target (nullable) receive gloss * d loss / d pred|target.
 * ---------------------------------------------------------------------------------- */
MVFO_API double mvfo_silog(const float *pred, const float *target, const float *mask, float beta,
                           float gloss, float *g_pred, float *g_target, int B, long N)
{
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        const float *p = pred + b * N, *t = target + b * N;
        const float *m = mask ? mask + b * N : 0;
        double s1 = 0.0, s2 = 0.0, n = 0.0;
        for (long i = 0; i < N; ++i) {
            float mk = m ? m[i] : 1.0f;
            float ld = logf(p[i] + 1e-7f) * mk - logf(t[i] + 1e-7f) * mk;
            s1 += ld; s2 += (double)ld * ld; n += mk;
        }
        n += 1e-8;
        total += s2 / n - (double)beta * s1 * s1 / (n * n);
        if (g_pred || g_target)
            for (long i = 0; i < N; ++i) {
                float mk = m ? m[i] : 1.0f;
                float lp = p[i] + 1e-7f, lt = t[i] + 1e-7f;
                float ld = logf(lp) * mk - logf(lt) * mk;
                double g = (double)gloss / B * (2.0 * ld / n - 2.0 * beta * s1 / (n * n));
                if (g_pred) g_pred[b * N + i] = (float)(g * mk / lp);
                if (g_target) g_target[b * N + i] = (float)(-g * mk / lt);
            }
    }
    return total / B;
}

/* ------------------------------------------------------------------------------------
 * f2: Trainer.affine_transform (reference: train.py:888-902) and the depth "restore" of
 * Trainer.compute_depth_consistency_loss_affine (reference: train.py:909-916).
 *
 * Both compose torchvision.transforms.functional.rotate(img, angle, interpolation=2)
 * (bilinear, zero fill: an inverse-mapped pixel-centre grid sampled with
 * grid_sample(align_corners=False, padding_mode="zeros")) with a box crop / paste and
 * F.interpolate(mode="bilinear", align_corners=False).  torchvision is absent here, so the
 * rotate step follows its published algorithm (v0.12, _get_inverse_affine_matrix +
 * _gen_affine_grid): PARITY UNPINNED for that step; the interpolate / crop / paste steps are
 * checked against torch itself (tests/test_oracle_golden.py).
 *
 * rot_pos(): source position in pixels of output pixel (y,x) for a rotation by `deg`:
 *   c,s = cos,sin(deg*pi/180) evaluated in double and rounded to fp32 (torchvision builds the
 *   matrix with python floats); grid g = (c*xs - s*ys)/(0.5*W) with xs = x + 0.5 - W/2;
 *   pixel = ((g + 1)*W - 1)/2  (ATen grid_sampler_unnormalize, align_corners=False).
 * ---------------------------------------------------------------------------------- */
typedef struct { float c, s; } trig_t;
static inline trig_t trig_of(float deg)
{
    double a = (double)deg * 3.14159265358979323846 / 180.0;
    trig_t t = {(float)cos(a), (float)sin(a)};
    return t;
}
static inline void rot_pos(trig_t t, int y, int x, int H, int W, float *px, float *py)
{
    float xs = (float)x + 0.5f - (float)W / 2.0f, ys = (float)y + 0.5f - (float)H / 2.0f;
    float gx = (t.c * xs - t.s * ys) / (0.5f * (float)W);
    float gy = (t.s * xs + t.c * ys) / (0.5f * (float)H);
    *px = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
    *py = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
}
/* bilinear, zero padding, of one plane at a real position */
static inline float sample_zeros(const float *im, int H, int W, float px, float py)
{
    float fx = floorf(px), fy = floorf(py);
    int x0 = (int)fx, y0 = (int)fy;
    float lx = px - fx, ly = py - fy, v = 0.0f;
    if (y0 >= 0 && y0 < H) {
        if (x0 >= 0 && x0 < W) v += im[(long)y0 * W + x0] * ((1.0f - lx) * (1.0f - ly));
        if (x0 + 1 >= 0 && x0 + 1 < W) v += im[(long)y0 * W + x0 + 1] * (lx * (1.0f - ly));
    }
    if (y0 + 1 >= 0 && y0 + 1 < H) {
        if (x0 >= 0 && x0 < W) v += im[(long)(y0 + 1) * W + x0] * ((1.0f - lx) * ly);
        if (x0 + 1 >= 0 && x0 + 1 < W) v += im[(long)(y0 + 1) * W + x0 + 1] * (lx * ly);
    }
    return v;
}
/* F.interpolate(bilinear, align_corners=False) source index of output index `o`:
 * in_size/out_size scale, clamp at 0, upper tap clamped (ATen area_pixel_compute_source_index) */
typedef struct { int i0, i1; float l; } rs_t;
static inline rs_t resize_src(int o, int in_size, int out_size)
{
    float sc = (float)in_size / (float)out_size;
    float t = ((float)o + 0.5f) * sc - 0.5f;
    if (t < 0.0f) t = 0.0f;
    rs_t r;
    r.i0 = (int)t;
    if (r.i0 > in_size - 1) r.i0 = in_size - 1;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l = t - (float)r.i0;
    if (r.l < 0.0f) r.l = 0.0f;
    if (r.l > 1.0f) r.l = 1.0f;
    return r;
}

/* out[b,c] = resize( rotate(img[b,c], angle[b]) [y0:y0+h, x0:x0+w] -> (H,W) ) */
MVFO_API void mvfo_affine_transform(const float *img, const float *angle, const int32_t *box,
                                    float *out, int B, int C, int H, int W)
{
    long N = (long)H * W;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y) {
            trig_t t = trig_of(angle[b]);
            int x0 = box[b * 4], y0 = box[b * 4 + 1], w = box[b * 4 + 2], h = box[b * 4 + 3];
            rs_t ry = resize_src(y, h, H);
            for (int x = 0; x < W; ++x) {
                rs_t rx = resize_src(x, w, W);
                float p[4][2];
                rot_pos(t, y0 + ry.i0, x0 + rx.i0, H, W, &p[0][0], &p[0][1]);
                rot_pos(t, y0 + ry.i0, x0 + rx.i1, H, W, &p[1][0], &p[1][1]);
                rot_pos(t, y0 + ry.i1, x0 + rx.i0, H, W, &p[2][0], &p[2][1]);
                rot_pos(t, y0 + ry.i1, x0 + rx.i1, H, W, &p[3][0], &p[3][1]);
                for (int c = 0; c < C; ++c) {
                    const float *im = img + ((long)b * C + c) * N;
                    float v00 = sample_zeros(im, H, W, p[0][0], p[0][1]);
                    float v01 = sample_zeros(im, H, W, p[1][0], p[1][1]);
                    float v10 = sample_zeros(im, H, W, p[2][0], p[2][1]);
                    float v11 = sample_zeros(im, H, W, p[3][0], p[3][1]);
                    /* ATen upsample_bilinear2d: h0lambda*(w0lambda*a + w1lambda*b) + h1lambda*(...) */
                    out[((long)b * C + c) * N + (long)y * W + x] =
                        (1.0f - ry.l) * ((1.0f - rx.l) * v00 + rx.l * v01) +
                        ry.l * ((1.0f - rx.l) * v10 + rx.l * v11);
                }
            }
        }
}

/* canvas value at (Y,X): inside the box the (H,W)->(h,w) bilinear resize of depth, else 0 */
static inline float canvas_at(const float *d, int H, int W, int x0, int y0, int w, int h, int Y, int X)
{
    if (X < x0 || X >= x0 + w || Y < y0 || Y >= y0 + h) return 0.0f;
    rs_t ry = resize_src(Y - y0, H, h), rx = resize_src(X - x0, W, w);
    const float *r0 = d + (long)ry.i0 * W, *r1 = d + (long)ry.i1 * W;
    return (1.0f - ry.l) * ((1.0f - rx.l) * r0[rx.i0] + rx.l * r0[rx.i1]) +
           ry.l * ((1.0f - rx.l) * r1[rx.i0] + rx.l * r1[rx.i1]);
}

/* out[b,c] = ratio[b] * rotate( paste( resize(depth[b,c] -> (h,w)) at (x0,y0) on zeros ), -angle[b] ) */
MVFO_API void mvfo_affine_restore(const float *depth, const float *angle, const int32_t *box,
                                  const float *ratio, float *out, int B, int C, int H, int W)
{
    long N = (long)H * W;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y) {
            trig_t t = trig_of(-angle[b]);
            int x0 = box[b * 4], y0 = box[b * 4 + 1], w = box[b * 4 + 2], h = box[b * 4 + 3];
            for (int x = 0; x < W; ++x) {
                float px, py;
                rot_pos(t, y, x, H, W, &px, &py);
                float fx = floorf(px), fy = floorf(py);
                int X = (int)fx, Y = (int)fy;
                float lx = px - fx, ly = py - fy;
                for (int c = 0; c < C; ++c) {
                    const float *d = depth + ((long)b * C + c) * N;
                    float v = 0.0f;
                    if (Y >= 0 && Y < H) {
                        if (X >= 0 && X < W) v += canvas_at(d, H, W, x0, y0, w, h, Y, X) * ((1.0f - lx) * (1.0f - ly));
                        if (X + 1 >= 0 && X + 1 < W) v += canvas_at(d, H, W, x0, y0, w, h, Y, X + 1) * (lx * (1.0f - ly));
                    }
                    if (Y + 1 >= 0 && Y + 1 < H) {
                        if (X >= 0 && X < W) v += canvas_at(d, H, W, x0, y0, w, h, Y + 1, X) * ((1.0f - lx) * ly);
                        if (X + 1 >= 0 && X + 1 < W) v += canvas_at(d, H, W, x0, y0, w, h, Y + 1, X + 1) * (lx * ly);
                    }
                    out[((long)b * C + c) * N + (long)y * W + x] = v * ratio[b];
                }
            }
        }
}

/* adjoint of mvfo_affine_restore w.r.t. depth: plain scatter in double (single thread per
 * plane), g_depth [B,C,H,W] */
MVFO_API void mvfo_affine_restore_bwd(const float *g_out, const float *angle, const int32_t *box,
                                      const float *ratio, float *g_depth, int B, int C, int H, int W)
{
    long N = (long)H * W;
#pragma omp parallel for collapse(2)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            double *acc = (double *)calloc((size_t)N, sizeof(double));
            trig_t t = trig_of(-angle[b]);
            int x0 = box[b * 4], y0 = box[b * 4 + 1], w = box[b * 4 + 2], h = box[b * 4 + 3];
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float px, py;
                    rot_pos(t, y, x, H, W, &px, &py);
                    float fx = floorf(px), fy = floorf(py);
                    int X = (int)fx, Y = (int)fy;
                    float lx = px - fx, ly = py - fy;
                    double g = (double)g_out[((long)b * C + c) * N + (long)y * W + x] * ratio[b];
                    for (int k = 0; k < 4; ++k) {
                        int YY = Y + (k >> 1), XX = X + (k & 1);
                        if (YY < 0 || YY >= H || XX < 0 || XX >= W) continue;
                        if (XX < x0 || XX >= x0 + w || YY < y0 || YY >= y0 + h) continue;
                        double wr = (double)((k & 1) ? lx : 1.0f - lx) * (double)((k >> 1) ? ly : 1.0f - ly);
                        rs_t ry = resize_src(YY - y0, H, h), rx = resize_src(XX - x0, W, w);
                        acc[(long)ry.i0 * W + rx.i0] += g * wr * (1.0 - ry.l) * (1.0 - rx.l);
                        acc[(long)ry.i0 * W + rx.i1] += g * wr * (1.0 - ry.l) * rx.l;
                        acc[(long)ry.i1 * W + rx.i0] += g * wr * ry.l * (1.0 - rx.l);
                        acc[(long)ry.i1 * W + rx.i1] += g * wr * ry.l * rx.l;
                    }
                }
            for (long i = 0; i < N; ++i) g_depth[((long)b * C + c) * N + i] = (float)acc[i];
            free(acc);
        }
}

/* ====================================================================================
 * The adjoint of one unit evaluated in DOUBLE (test infrastructure for the 1e-4 per-element gradient bar).
 * Same chain as mvfo_losses_base_bwd -> mvfo_warp_bwd -> mvfo_smooth_bwd above (autograd of reference
 * train.py:956-1051, layers.py:231-290).  Every DECISION is the fp32 forward's -- the argmin map it is given, the
 * bilinear cell and the clipped-coordinate flags (tap_of on the fp32 chain), whether SSIM's clamp passes the gradient
 * (the fp32 raw value), the sign of a disparity difference -- so this differentiates the SAME piecewise function the
 * fp32 forward evaluated; every VALUE (window means, SSIM partials, projection chain, bilinear weights, smoothness
 * weights, all sums) is formed in double from the fp32 inputs.  Three legitimate fp32 evaluation orders of this
 * adjoint (the reference's autograd, the fp32 oracle above, the HIP kernel) sit up to 2.7e-4 of the tensor max
 * apart at their worst pixel; against this double evaluation each of them can be held to 1e-4.  Pinned to the
 * reference evaluated in float64 (tests/golden/g4_f64_C2_*.npz) by tests/test_oracle_golden.py.
 * ==================================================================================== */
typedef struct { double mu_x, mu_y, exx, eyy, exy; } win64_t;

static inline win64_t window_stats64(const float *x, const float *y, int H, int W, int py, int px)
{
    double sx = 0.0, sy = 0.0, sxx = 0.0, syy = 0.0, sxy = 0.0;
    for (int dy = -1; dy <= 1; ++dy) {
        int yy = refl(py + dy, H);
        for (int dx = -1; dx <= 1; ++dx) {
            int xx = refl(px + dx, W);
            double a = x[yy * W + xx], b = y[yy * W + xx];
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
    }
    win64_t w = {sx / 9.0, sy / 9.0, sxx / 9.0, syy / 9.0, sxy / 9.0};
    return w;
}

typedef struct { double dmux, dexx, dexy; } dwin64_t;

/* d SSIM / d (mu_x, Exx, Exy) of the un-clamped map (the caller has checked that the clamp is inactive) */
static inline dwin64_t ssim_partials64(const win64_t *w)
{
    const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
    double mx = w->mu_x, my = w->mu_y;
    double sigma_x = w->exx - mx * mx, sigma_y = w->eyy - my * my, sigma_xy = w->exy - mx * my;
    double A1 = 2.0 * mx * my + C1, A2 = 2.0 * sigma_xy + C2;
    double B1 = mx * mx + my * my + C1, B2 = sigma_x + sigma_y + C2;
    double n = A1 * A2, d = B1 * B2;
    double kn = -0.5 / d, kd = 0.5 * n / (d * d);
    dwin64_t g;
    g.dmux = kn * (2.0 * my * A2 - 2.0 * my * A1) + kd * (2.0 * mx * B2 - 2.0 * mx * B1);
    g.dexy = kn * 2.0 * A1;
    g.dexx = kd * B1;
    return g;
}

/* gwarped[k] [B,3,H,W] double, overwritten */
MVFO_API void mvfo_losses_base_bwd_f64(const float *tgt, const float *const *warped, const int32_t *idx,
                                       const float *mask_rec, int S, int flags, double gloss,
                                       double *const *gwarped, int B, int H, int W)
{
    long N = (long)H * W;
    int no_ssim = flags & MVFO_NO_SSIM, avg = flags & MVFO_AVG_REPROJ, automask = !(flags & MVFO_NO_AUTOMASK);
    int n_id = automask ? (avg ? 1 : S) : 0;
    float C1 = ssim_c1(), C2 = ssim_c2();
    double gpix = gloss / ((double)B * (double)N);
    for (int k = 0; k < S; ++k) {
        double *gw = gwarped[k];
        memset(gw, 0, sizeof(double) * B * 3 * N);
        const int BAND = 8, nband = (H + BAND - 1) / BAND;      /* row bands of one colour never touch the same row */
        for (int colour = 0; colour < 2; ++colour) {
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
        for (int bc = 0; bc < B * 3; ++bc)
        for (int band = colour; band < nband; band += 2) {
            int b = bc / 3;
            const float *xb = warped[k] + bc * N, *yb = tgt + bc * N;
            int py_end = (band + 1) * BAND < H ? (band + 1) * BAND : H;
            for (int py = band * BAND; py < py_end; ++py)
                for (int px = 0; px < W; ++px) {
                    long i = (long)py * W + px;
                    int sel = idx ? idx[b * N + i] : -1;
                    double wgt;
                    if (sel < 0) wgt = avg ? 1.0 / (double)S : 1.0;
                    else if (avg) wgt = (sel == n_id) ? 1.0 / (double)S : 0.0;
                    else wgt = (sel == n_id + k) ? 1.0 : 0.0;
                    if (wgt == 0.0) continue;
                    double g = gpix * wgt;
                    if (mask_rec) g *= (double)mask_rec[b * N + i];
                    if (g == 0.0) continue;
                    float df = yb[i] - xb[i];
                    double sg = (df > 0.f) ? -1.0 : ((df < 0.f) ? 1.0 : 0.0);
                    gw[bc * N + i] += g * (no_ssim ? 1.0 : 0.15) * sg / 3.0;
                    if (no_ssim) continue;
                    /* the clamp decision is the fp32 forward's */
                    win_t wf = window_stats(xb, yb, H, W, py, px);
                    float raw;
                    (void)ssim_from(&wf, C1, C2, &raw);
                    if (!(raw >= 0.0f && raw <= 1.0f)) continue;
                    win64_t w = window_stats64(xb, yb, H, W, py, px);
                    dwin64_t p = ssim_partials64(&w);
                    double gs = g * 0.85 / 3.0;
                    for (int dy = -1; dy <= 1; ++dy) {
                        int yy = refl(py + dy, H);
                        for (int dx = -1; dx <= 1; ++dx) {
                            int xx = refl(px + dx, W);
                            double a = xb[yy * W + xx], bq = yb[yy * W + xx];
                            gw[bc * N + yy * W + xx] += gs * (p.dmux + 2.0 * a * p.dexx + bq * p.dexy) / 9.0;
                        }
                    }
                }
        }
        }
    }
}

/* gwarped [B,3,H,W] double -> gdisp [B,1,H,W] double (ACCUMULATED into), gT [B,4,4] double (overwritten) */
MVFO_API void mvfo_warp_bwd_f64(const float *disp, const float *invK, const float *K, const float *T,
                                const float *src, const double *gwarped, double *gdisp, double *gT,
                                int B, int H, int W, float min_disp, float range, float eps)
{
    long N = (long)H * W;
    float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    for (int b = 0; b < B; ++b) {
        float P[12];
        proj_matrix(K + b * 16, T + b * 16, P);
        double Pd[12];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 4; ++j) {
                double a = 0.0;
                for (int q = 0; q < 4; ++q) a += (double)K[b * 16 + i * 4 + q] * (double)T[b * 16 + q * 4 + j];
                Pd[i * 4 + j] = a;
            }
        const float *iK = invK + b * 16;
        double gP[12];
        for (int k = 0; k < 12; ++k) gP[k] = 0.0;
#pragma omp parallel
        {
            double lP[12];
            for (int k = 0; k < 12; ++k) lP[k] = 0.0;
#pragma omp for schedule(static)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    long i = (long)y * W + x;
                    /* the fp32 forward chain: the cell and the clipped-coordinate flags */
                    float r[3], X[4], c[3];
                    ray_of(iK, (float)x, (float)y, r);
                    float scaled = min_disp + range * disp[b * N + i];
                    float d = 1.0f / scaled;
                    X[0] = d * r[0]; X[1] = d * r[1]; X[2] = d * r[2]; X[3] = 1.0f;
                    for (int q = 0; q < 3; ++q) {
                        float a = P[q * 4 + 0] * X[0];
                        a = fmaf(P[q * 4 + 1], X[1], a);
                        a = fmaf(P[q * 4 + 2], X[2], a);
                        a = fmaf(P[q * 4 + 3], X[3], a);
                        c[q] = a;
                    }
                    float z = c[2] + eps;
                    float u = c[0] / z, v = c[1] / z;
                    tap_t t = tap_of((u / wm1 - 0.5f) * 2.0f, (v / hm1 - 0.5f) * 2.0f, H, W);
                    /* the same chain in double: values */
                    double rd[3], Xd[4], cd[3];
                    for (int q = 0; q < 3; ++q)
                        rd[q] = (double)iK[q * 4 + 0] * x + (double)iK[q * 4 + 1] * y + (double)iK[q * 4 + 2];
                    double dd = 1.0 / ((double)min_disp + (double)range * (double)disp[b * N + i]);
                    Xd[0] = dd * rd[0]; Xd[1] = dd * rd[1]; Xd[2] = dd * rd[2]; Xd[3] = 1.0;
                    for (int q = 0; q < 3; ++q)
                        cd[q] = Pd[q * 4 + 0] * Xd[0] + Pd[q * 4 + 1] * Xd[1] + Pd[q * 4 + 2] * Xd[2] + Pd[q * 4 + 3];
                    double zd = cd[2] + (double)eps, ud = cd[0] / zd, vd = cd[1] / zd;
                    /* un-normalised sample position = (u, v) clipped to the image; weights inside the fp32 cell */
                    double ix = ud < 0.0 ? 0.0 : (ud > (double)wm1 ? (double)wm1 : ud);
                    double iy = vd < 0.0 ? 0.0 : (vd > (double)hm1 ? (double)hm1 : vd);
                    double wx = ix - (double)t.x0, wy = iy - (double)t.y0;
                    double gix = 0.0, giy = 0.0;
                    for (int ch = 0; ch < 3; ++ch) {
                        const float *im = src + ((long)(b * 3 + ch)) * N;
                        double nw = im[t.y0 * W + t.x0], ne = im[t.y0 * W + t.x1];
                        double sw = im[t.y1 * W + t.x0], se = im[t.y1 * W + t.x1];
                        double g = gwarped[((long)(b * 3 + ch)) * N + i];
                        gix += g * ((ne - nw) * (1.0 - wy) + (se - sw) * wy);
                        giy += g * ((sw - nw) * (1.0 - wx) + (se - ne) * wx);
                    }
                    /* d(ix)/d(gx) * d(gx)/d(u) = (W-1)/2 * 2/(W-1) = 1 where the coordinate was not clipped */
                    double gu = t.inx ? gix : 0.0, gv = t.iny ? giy : 0.0;
                    double gc[3];
                    gc[0] = gu / zd;
                    gc[1] = gv / zd;
                    gc[2] = -(gu * ud + gv * vd) / zd;
                    double gd = 0.0;
                    for (int j = 0; j < 3; ++j)
                        gd += (gc[0] * Pd[0 * 4 + j] + gc[1] * Pd[1 * 4 + j] + gc[2] * Pd[2 * 4 + j]) * rd[j];
                    for (int q = 0; q < 3; ++q)
                        for (int j = 0; j < 4; ++j) lP[q * 4 + j] += gc[q] * Xd[j];
                    gdisp[b * N + i] += -gd * dd * dd * (double)range;
                }
#pragma omp critical
            for (int k = 0; k < 12; ++k) gP[k] += lP[k];
        }
        const float *Kb = K + b * 16;
        for (int k = 0; k < 4; ++k)
            for (int j = 0; j < 4; ++j) {
                double a = 0.0;
                for (int q = 0; q < 3; ++q) a += (double)Kb[q * 4 + k] * gP[q * 4 + j];
                gT[b * 16 + k * 4 + j] = a;
            }
    }
}

/* grad of `scale * smooth(disp / (mean_hw(disp)+1e-7), img)` w.r.t. disp, ACCUMULATED into gdisp (double) */
MVFO_API void mvfo_smooth_bwd_f64(const float *disp, const float *img, double *gdisp, int B, int H, int W,
                                  int normalise, double scale)
{
    long N = (long)H * W;
    double cx = scale / ((double)B * H * (W - 1));
    double cy = scale / ((double)B * (H - 1) * W);
    double *gn = (double *)malloc(sizeof(double) * N);
    for (int b = 0; b < B; ++b) {
        const float *d = disp + b * N;
        const float *im = img + b * 3 * N;
        float denom = 1.0f;
        double denom_d = 1.0;
        if (normalise) {
            double m = 0.0;
            for (long i = 0; i < N; ++i) m += d[i];
            denom = (float)(m / (double)N) + 1e-7f;       /* the forward's (signs) */
            denom_d = m / (double)N + 1e-7;
        }
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                long i = (long)y * W + x;
                double acc = 0.0;
#define MVFO_SM_TERM(j_, i_, c_, sign_)                                                                       \
                {                                                                                              \
                    float df = d[j_] / denom - d[i_] / denom;                                                  \
                    double gi = ((fabs((double)im[j_] - (double)im[i_]) +                                      \
                                  fabs((double)im[N + (j_)] - (double)im[N + (i_)])) +                         \
                                 fabs((double)im[2 * N + (j_)] - (double)im[2 * N + (i_)])) / 3.0;             \
                    double sg = (df > 0.f) ? 1.0 : ((df < 0.f) ? -1.0 : 0.0);                                  \
                    acc += (sign_) * (c_) * exp(-gi) * sg;                                                     \
                }
                if (y >= 1) MVFO_SM_TERM(i - W, i, cy, -1.0)
                if (x >= 1) MVFO_SM_TERM(i - 1, i, cx, -1.0)
                if (x + 1 < W) MVFO_SM_TERM(i, i + 1, cx, 1.0)
                if (y + 1 < H) MVFO_SM_TERM(i, i + W, cy, 1.0)
#undef MVFO_SM_TERM
                gn[i] = acc;
            }
        if (normalise) {
            double dot = 0.0;
            for (long i = 0; i < N; ++i) dot += gn[i] * (double)d[i];
            double corr = dot / (denom_d * denom_d) / (double)N;
#pragma omp parallel for schedule(static)
            for (long i = 0; i < N; ++i) gdisp[b * N + i] += gn[i] / denom_d - corr;
        } else {
#pragma omp parallel for schedule(static)
            for (long i = 0; i < N; ++i) gdisp[b * N + i] += gn[i];
        }
    }
    free(gn);
}
