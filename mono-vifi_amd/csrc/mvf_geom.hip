// mvf_geom.hip -- geometry stages of the hot path for gfx950 (MI355X):
// disp_to_depth, BackprojectDepth, Project3D, grid_sample(border, align_corners=True), the
// per-pixel fusion of the four (Trainer.generate_images_pred) and the pose glue.
//
// All kernels are one-pixel-per-lane streaming kernels: lanes of a wavefront walk
// consecutive x, so every planar [B,C,H,W] access is a fully coalesced 256-B wave
// transaction; the 2x2 bilinear taps of neighbouring lanes fall in the same or adjacent
// cache lines for realistic ego-motion and are served by L2.  Bound: HBM bandwidth.
// Compiled with -ffp-contract=off (see mvf_common.hpp for the arithmetic contract).
#include "mvf_common.hpp"

using namespace mvf;

namespace {

constexpr int NT = 256;

inline dim3 pix_grid(int H, int W, int B) { return dim3((unsigned)((H * W + NT - 1) / NT), (unsigned)B); }

// ---------------------------------------------------------------- a1 disp_to_depth
__global__ void __launch_bounds__(NT) k_disp_to_depth(const float *__restrict__ disp,
                                                      float *__restrict__ scaled,
                                                      float *__restrict__ depth, int64_t n,
                                                      float min_disp, float range)
{
    int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * NT;
    for (; i < n; i += stride) {
        float s = min_disp + range * disp[i];
        if (scaled) scaled[i] = s;
        if (depth) depth[i] = 1.0f / s;
    }
}

__global__ void __launch_bounds__(NT) k_disp_to_depth_bwd(const float *__restrict__ disp,
                                                          const float *__restrict__ g_scaled,
                                                          const float *__restrict__ g_depth,
                                                          float *__restrict__ g_disp, int64_t n,
                                                          float min_disp, float range)
{
    int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * NT;
    for (; i < n; i += stride) {
        float s = min_disp + range * disp[i];
        float d = 1.0f / s;
        float g = 0.0f;
        if (g_scaled) g += g_scaled[i] * range;
        if (g_depth) g += -g_depth[i] * d * d * range;
        g_disp[i] = g;
    }
}

// ---------------------------------------------------------------- a2 BackprojectDepth
__global__ void __launch_bounds__(NT) k_backproject(const float *__restrict__ depth,
                                                    const float *__restrict__ invK,
                                                    float *__restrict__ cam, int H, int W)
{
    int N = H * W, b = blockIdx.y;
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    int y = i / W, x = i - y * W;
    float r[3];
    ray_of(invK + b * 16, (float)x, (float)y, r);
    float d = depth[(size_t)b * N + i];
    float *o = cam + (size_t)b * 4 * N + i;
    o[0] = d * r[0];
    o[(size_t)N] = d * r[1];
    o[(size_t)2 * N] = d * r[2];
    o[(size_t)3 * N] = 1.0f;
}

__global__ void __launch_bounds__(NT) k_backproject_bwd(const float *__restrict__ g_cam,
                                                        const float *__restrict__ invK,
                                                        float *__restrict__ g_depth, int H, int W)
{
    int N = H * W, b = blockIdx.y;
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    int y = i / W, x = i - y * W;
    float r[3];
    ray_of(invK + b * 16, (float)x, (float)y, r);
    const float *g = g_cam + (size_t)b * 4 * N + i;
    g_depth[(size_t)b * N + i] = g[0] * r[0] + g[(size_t)N] * r[1] + g[(size_t)2 * N] * r[2];
}

// ---------------------------------------------------------------- a3 Project3D
MVF_DEV void load_P(const float *__restrict__ K, const float *__restrict__ T, float P[12])
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) P[i * 4 + j] = proj_entry(K, T, i, j);
}

__global__ void __launch_bounds__(NT) k_project(const float *__restrict__ cam,
                                                const float *__restrict__ K,
                                                const float *__restrict__ T,
                                                float *__restrict__ pix, int H, int W, float eps)
{
    int N = H * W, b = blockIdx.y;
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    float P[12];
    load_P(K + b * 16, T + b * 16, P);
    const float *cp = cam + (size_t)b * 4 * N + i;
    float X0 = cp[0], X1 = cp[(size_t)N], X2 = cp[(size_t)2 * N], X3 = cp[(size_t)3 * N];
    float c[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float a = P[q * 4 + 0] * X0;
        a = fmaf(P[q * 4 + 1], X1, a);
        a = fmaf(P[q * 4 + 2], X2, a);
        a = fmaf(P[q * 4 + 3], X3, a);
        c[q] = a;
    }
    float gx, gy, u, v, z;
    normalise_uv(c, eps, (float)(W - 1), (float)(H - 1), gx, gy, u, v, z);
    float2 o = make_float2(gx, gy);
    reinterpret_cast<float2 *>(pix)[(size_t)b * N + i] = o;
}

// per-block partial of sum_i gc[q]*Xh[j] (12 values) -> ws[(b*nblk + blk)*12 + k]
template <int NTT>
MVF_DEV void reduce12_to_ws(float acc[12], float *__restrict__ ws, float *scratch)
{
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        float s = block_sum<NTT>(acc[k], scratch);
        if (threadIdx.x == 0) ws[k] = s;
    }
}

__global__ void __launch_bounds__(NT) k_project_bwd(const float *__restrict__ cam,
                                                    const float *__restrict__ K,
                                                    const float *__restrict__ T,
                                                    const float *__restrict__ g_pix,
                                                    float *__restrict__ g_cam,
                                                    float *__restrict__ ws, int H, int W, float eps)
{
    __shared__ float scratch[NT / kWave];
    int N = H * W, b = blockIdx.y;
    int i = blockIdx.x * NT + threadIdx.x;
    float P[12];
    load_P(K + b * 16, T + b * 16, P);
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0f;
    if (i < N) {
        const float *cp = cam + (size_t)b * 4 * N + i;
        float Xh[4] = {cp[0], cp[(size_t)N], cp[(size_t)2 * N], cp[(size_t)3 * N]};
        float c[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float a = P[q * 4 + 0] * Xh[0];
            a = fmaf(P[q * 4 + 1], Xh[1], a);
            a = fmaf(P[q * 4 + 2], Xh[2], a);
            a = fmaf(P[q * 4 + 3], Xh[3], a);
            c[q] = a;
        }
        float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
        float z = c[2] + eps, u = c[0] / z, v = c[1] / z;
        float2 g = reinterpret_cast<const float2 *>(g_pix)[(size_t)b * N + i];
        float gu = g.x * 2.0f / wm1, gv = g.y * 2.0f / hm1;
        float gc[3] = {gu / z, gv / z, -(gu * u + gv * v) / z};
        if (g_cam) {
            float *o = g_cam + (size_t)b * 4 * N + i;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[(size_t)j * N] = gc[0] * P[j] + gc[1] * P[4 + j] + gc[2] * P[8 + j];
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[q * 4 + j] = gc[q] * Xh[j];
    }
    if (ws) reduce12_to_ws<NT>(acc, ws + ((size_t)b * gridDim.x + blockIdx.x) * 12, scratch);
}

// fold per-block grad_P partials in fixed order (fp64) and form grad_T = K^T [grad_P ; 0]
// grid (B, S): ws [S][B][nblk][12], K [B,4,4], gT [S][B][4][4]
__global__ void __launch_bounds__(NT) k_finish_gT(const float *__restrict__ ws,
                                                  const float *__restrict__ K,
                                                  float *__restrict__ gT, int nblk)
{
    __shared__ double sh[NT / kWave][12];
    __shared__ double gP[12];
    int b = blockIdx.x, s = blockIdx.y, B = gridDim.x;
    const float *w = ws + ((size_t)s * B + b) * nblk * 12;
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0;
    for (int blk = threadIdx.x; blk < nblk; blk += NT)
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[k] += (double)w[(size_t)blk * 12 + k];
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) sh[wid][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        double v = 0.0;
        for (int i = 0; i < NT / kWave; ++i) v += sh[i][threadIdx.x];
        gP[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        int k = threadIdx.x >> 2, j = threadIdx.x & 3;
        const float *Kb = K + b * 16;
        double a = 0.0;
        for (int q = 0; q < 3; ++q) a += (double)Kb[q * 4 + k] * gP[q * 4 + j];
        gT[((size_t)s * B + b) * 16 + threadIdx.x] = (float)a;
    }
}

// ---------------------------------------------------------------- a4 grid_sample
__global__ void __launch_bounds__(NT) k_grid_sample(const float *__restrict__ img,
                                                    const float *__restrict__ grid,
                                                    float *__restrict__ out,
                                                    int32_t *__restrict__ idx_xy, int C, int H, int W)
{
    int N = H * W, b = blockIdx.y;
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    float2 g = reinterpret_cast<const float2 *>(grid)[(size_t)b * N + i];
    Tap t = tap_of(g.x, g.y, H, W);
    if (idx_xy) reinterpret_cast<int2 *>(idx_xy)[(size_t)b * N + i] = make_int2(t.x0, t.y0);
    if (out)
        for (int c = 0; c < C; ++c)
            out[((size_t)b * C + c) * N + i] = bilerp(img + ((size_t)b * C + c) * N, W, t);
}

MVF_DEV void scatter_taps(float *__restrict__ gi, int W, const Tap &t, float g)
{
    float w = t.wx, e = 1.0f - w, n = t.wy, s = 1.0f - n;
    atomicAdd(gi + t.y0 * W + t.x0, g * (s * e));
    atomicAdd(gi + t.y0 * W + t.x1, g * (s * w));
    atomicAdd(gi + t.y1 * W + t.x0, g * (n * e));
    atomicAdd(gi + t.y1 * W + t.x1, g * (n * w));
}

__global__ void __launch_bounds__(NT) k_grid_sample_bwd(const float *__restrict__ img,
                                                        const float *__restrict__ grid,
                                                        const float *__restrict__ g_out,
                                                        float *__restrict__ g_grid,
                                                        float *__restrict__ g_img, int C, int H, int W)
{
    int N = H * W, b = blockIdx.y;
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    float2 g = reinterpret_cast<const float2 *>(grid)[(size_t)b * N + i];
    Tap t = tap_of(g.x, g.y, H, W);
    float gx = 0.0f, gy = 0.0f;
    for (int c = 0; c < C; ++c) {
        float go = g_out[((size_t)b * C + c) * N + i];
        float dx, dy;
        bilerp_grad(img + ((size_t)b * C + c) * N, W, t, dx, dy);
        gx += go * dx;
        gy += go * dy;
        if (g_img) scatter_taps(g_img + ((size_t)b * C + c) * N, W, t, go);
    }
    float sx = (float)(W - 1) / 2.0f, sy = (float)(H - 1) / 2.0f;
    if (g_grid)
        reinterpret_cast<float2 *>(g_grid)[(size_t)b * N + i] =
            make_float2(t.inx ? gx * sx : 0.0f, t.iny ? gy * sy : 0.0f);
}

// ---------------------------------------------------------------- f1 flow warp
// IFRNet.warp / FusionModule.warp_features (reference: networks/IFRNet.py:7-15,
// networks/fusion_module.py:80-90): grid = linspace(-1,1) + flow / ((size-1)/2), then the same
// bilinear / border / align_corners=True gather as a4.  xs[W], ys[H] are the linspace values
// (passed in so that they are the reference's own fp32 values).  One lane per pixel, CCH
// channels per lane: the tap is computed once and reused by every channel of the chunk.
constexpr int FW_CCH = 8;

__global__ void __launch_bounds__(NT) k_flow_warp_fwd(const float *__restrict__ img,
                                                      const float *__restrict__ flow,
                                                      const float *__restrict__ xs,
                                                      const float *__restrict__ ys,
                                                      float *__restrict__ out,
                                                      int32_t *__restrict__ idx_xy, int C, int H, int W)
{
    int N = H * W, b = blockIdx.z;
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    int y = i / W, x = i - y * W;
    Tap t = flow_tap(flow, xs, ys, b, i, x, y, H, W);
    if (idx_xy && blockIdx.y == 0)
        reinterpret_cast<int2 *>(idx_xy)[(size_t)b * N + i] = make_int2(t.x0, t.y0);
    if (!out) return;
    int c0 = blockIdx.y * FW_CCH, c1 = min(c0 + FW_CCH, C);
    for (int c = c0; c < c1; ++c)
        out[((size_t)b * C + c) * N + i] = bilerp(img + ((size_t)b * C + c) * N, W, t);
}

__global__ void __launch_bounds__(NT) k_flow_warp_bwd(const float *__restrict__ img,
                                                      const float *__restrict__ flow,
                                                      const float *__restrict__ xs,
                                                      const float *__restrict__ ys,
                                                      const float *__restrict__ g_out,
                                                      float *__restrict__ g_img,
                                                      float *__restrict__ g_flow_part, int C, int H,
                                                      int W)
{
    int N = H * W, b = blockIdx.z;
    const int i_raw = blockIdx.x * NT + threadIdx.x;
    const bool active = i_raw < N;
    const int i = active ? i_raw : N - 1;
    int y = i / W, x = i - y * W;
    Tap t = flow_tap(flow, xs, ys, b, i, x, y, H, W);
    const ScatterLinks links = scatter_links(t, active);
    int c0 = blockIdx.y * FW_CCH, c1 = min(c0 + FW_CCH, C);
    float gx = 0.0f, gy = 0.0f;
    for (int c = c0; c < c1; ++c) {
        float go = active ? g_out[((size_t)b * C + c) * N + i] : 0.0f;
        if (g_img) scatter_taps_linked(g_img + ((size_t)b * C + c) * N, W, t, go, active, links);
        if (g_flow_part && active) {
            float dx, dy;
            bilerp_grad(img + ((size_t)b * C + c) * N, W, t, dx, dy);
            gx += go * dx;
            gy += go * dy;
        }
    }
    if (g_flow_part && active) {
        // d(ix)/d(flow_x) = ((W-1)/2) / ((W-1)/2) = 1 where the coordinate was not clipped;
        // chunk partials are summed by the caller-visible second pass (k_flow_grad_fold)
        size_t o = (((size_t)blockIdx.y * gridDim.z + b) * 2) * N + i;
        g_flow_part[o] = t.inx ? gx : 0.0f;
        g_flow_part[o + N] = t.iny ? gy : 0.0f;
    }
}

// g_flow[b,2,N] = sum over channel chunks of the partials (fixed order)
__global__ void __launch_bounds__(NT) k_flow_grad_fold(const float *__restrict__ part,
                                                       float *__restrict__ g_flow, int nchunk, size_t n)
{
    size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int k = 0; k < nchunk; ++k) s += part[(size_t)k * n + i];
    g_flow[i] = s;
}

// ---------------------------------------------------------------- a5 fused warp
__global__ void __launch_bounds__(NT) k_warp_fwd(const float *__restrict__ disp,
                                                 const float *__restrict__ invK,
                                                 const float *__restrict__ K,
                                                 const float *__restrict__ T,
                                                 const float *__restrict__ src,
                                                 float *__restrict__ warped, float *__restrict__ pix,
                                                 int32_t *__restrict__ idx_xy, int H, int W,
                                                 float min_disp, float range, float eps)
{
    int N = H * W, b = blockIdx.y;
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    int y = i / W, x = i - y * W;
    float P[12];
    load_P(K + b * 16, T + b * 16, P);
    WarpPoint w = warp_point(disp[(size_t)b * N + i], invK + b * 16, P, x, y, H, W, min_disp,
                             range, eps);
    if (pix) reinterpret_cast<float2 *>(pix)[(size_t)b * N + i] = make_float2(w.gx, w.gy);
    if (idx_xy) reinterpret_cast<int2 *>(idx_xy)[(size_t)b * N + i] = make_int2(w.t.x0, w.t.y0);
    if (warped) {
        const float *sp = src + (size_t)b * 3 * N;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            warped[((size_t)b * 3 + c) * N + i] = bilerp(sp + (size_t)c * N, W, w.t);
    }
}

__global__ void __launch_bounds__(NT) k_warp_bwd(const float *__restrict__ disp,
                                                 const float *__restrict__ invK,
                                                 const float *__restrict__ K,
                                                 const float *__restrict__ T,
                                                 const float *__restrict__ src,
                                                 const float *__restrict__ g_warped,
                                                 float *__restrict__ g_disp,
                                                 float *__restrict__ g_src, float *__restrict__ ws,
                                                 int accumulate, int H, int W, float min_disp,
                                                 float range, float eps)
{
    __shared__ float scratch[NT / kWave];
    int N = H * W, b = blockIdx.y;
    int i = blockIdx.x * NT + threadIdx.x;
    float P[12];
    load_P(K + b * 16, T + b * 16, P);
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0f;
    if (i < N) {
        int y = i / W, x = i - y * W;
        WarpPoint w = warp_point(disp[(size_t)b * N + i], invK + b * 16, P, x, y, H, W, min_disp,
                                 range, eps);
        const float *sp = src + (size_t)b * 3 * N;
        float gix = 0.0f, giy = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float go = g_warped[((size_t)b * 3 + c) * N + i];
            float dx, dy;
            bilerp_grad(sp + (size_t)c * N, W, w.t, dx, dy);
            gix += go * dx;
            giy += go * dy;
            if (g_src) scatter_taps(g_src + ((size_t)b * 3 + c) * N, W, w.t, go);
        }
        float gc[3];
        float gd = warp_point_bwd(w, P, gix, giy, H, W, gc);
        float g = -gd * w.depth * w.depth * range;
        size_t o = (size_t)b * N + i;
        g_disp[o] = accumulate ? g_disp[o] + g : g;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            acc[q * 4 + 0] = gc[q] * w.X[0];
            acc[q * 4 + 1] = gc[q] * w.X[1];
            acc[q * 4 + 2] = gc[q] * w.X[2];
            acc[q * 4 + 3] = gc[q];
        }
    }
    reduce12_to_ws<NT>(acc, ws + ((size_t)b * gridDim.x + blockIdx.x) * 12, scratch);
}

// ---------------------------------------------------------------- f2 scale-invariant log loss
// Trainer.compute_SI_log_depth_loss (reference: train.py:924-941): per image
//   ld = log(pred+1e-7)*m - log(target+1e-7)*m ; n = sum m + 1e-8
//   loss_b = sum ld^2 / n - beta * (sum ld)^2 / n^2 ; loss = mean_b loss_b
// Forward: per-block partials {sum ld, sum ld^2, sum m} -> ws ; finishing kernel (fp64) writes
// loss[0] and sums[B][4] = {s1, s2, n, -} for the backward.  Backward is element-wise.
constexpr int SIL_NB = 64;     // blocks per image

__global__ void __launch_bounds__(NT) k_silog_partial(const float *__restrict__ pred,
                                                      const float *__restrict__ target,
                                                      const float *__restrict__ mask,
                                                      float *__restrict__ ws, int N)
{
    __shared__ float scratch[3 * (NT / 16)];
    int b = blockIdx.y;
    const float *p = pred + (size_t)b * N, *t = target + (size_t)b * N;
    const float *m = mask ? mask + (size_t)b * N : nullptr;
    float v[3] = {0.0f, 0.0f, 0.0f};
    for (int i = blockIdx.x * NT + threadIdx.x; i < N; i += SIL_NB * NT) {
        float mk = m ? m[i] : 1.0f;
        float ld = logf(p[i] + 1e-7f) * mk - logf(t[i] + 1e-7f) * mk;
        v[0] += ld;
        v[1] += ld * ld;
        v[2] += mk;
    }
    float tot = block_sum_many<NT, 3>(v, scratch);
    if (threadIdx.x < 3) ws[((size_t)b * SIL_NB + blockIdx.x) * 4 + threadIdx.x] = tot;
}

__global__ void k_silog_finish(const float *__restrict__ ws, float *__restrict__ loss,
                               float *__restrict__ sums, int B, float beta)
{
    // one lane per image, 64 images per pass, any batch size (images folded in index order)
    __shared__ double acc[64];
    double tot = 0.0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + threadIdx.x;
        double lb = 0.0;
        if (b < B) {
            double s1 = 0.0, s2 = 0.0, n = 0.0;
            for (int k = 0; k < SIL_NB; ++k) {
                const float *q = ws + ((size_t)b * SIL_NB + k) * 4;
                s1 += q[0]; s2 += q[1]; n += q[2];
            }
            n += 1e-8;
            lb = s2 / n - (double)beta * s1 * s1 / (n * n);
            sums[b * 4 + 0] = (float)s1;
            sums[b * 4 + 1] = (float)s2;
            sums[b * 4 + 2] = (float)n;
            sums[b * 4 + 3] = 0.0f;
        }
        acc[threadIdx.x] = lb;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < 64; ++i) tot += acc[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(tot / (double)B);
}

__global__ void __launch_bounds__(NT) k_silog_bwd(const float *__restrict__ pred,
                                                  const float *__restrict__ target,
                                                  const float *__restrict__ mask,
                                                  const float *__restrict__ sums,
                                                  const float *__restrict__ g_loss,
                                                  float *__restrict__ g_pred,
                                                  float *__restrict__ g_target, int B, int N, float beta)
{
    int b = blockIdx.y;
    int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    size_t o = (size_t)b * N + i;
    float mk = mask ? mask[o] : 1.0f;
    float lp = pred[o] + 1e-7f, lt = target[o] + 1e-7f;
    float ld = logf(lp) * mk - logf(lt) * mk;
    float s1 = sums[b * 4 + 0], n = sums[b * 4 + 2];
    float g = g_loss[0] / (float)B * (2.0f * ld / n - 2.0f * beta * s1 / (n * n));
    if (g_pred) g_pred[o] = g * mk / lp;
    if (g_target) g_target[o] = -g * mk / lt;
}

// ---- the SI-log losses of a step as ONE forward and ONE backward launch (round 5) -----------------------------------
// process_batch evaluates nine of them (train.py:813-815, 868-882), each on 12 images: nine partial + nine finishing
// launches forward and nine backward, every one of them bound by its launch, not by its 6 MB.  Jobs: per-image base +
// stride (the depth views of an interleaved decoder output are read where they lie).  Forward: blocks (chunk, image,
// job) write partials; one finishing block per job folds its images in index order (fp64, the order of
// k_silog_finish) into sums[job] and loss[job]; the last job to finish adds the losses in job order into total[0].
struct SilogJob {
    const float *pred, *target, *mask;
    size_t ps, ts, ms;                   // image strides (floats)
    float *g_pred, *g_target;            // backward outputs [B,N] contiguous, nullable
};
struct SilogJobs {
    SilogJob j[MVF_MAX_SILOG_JOBS];
    int n;
};
__global__ void __launch_bounds__(NT) k_silog_many_partial(SilogJobs J, float *__restrict__ ws, int B, int N)
{
    __shared__ float scratch[3 * (NT / 16)];
    const int b = blockIdx.y, jb = blockIdx.z;
    const SilogJob &job = J.j[jb];
    const float *p = job.pred + (size_t)b * job.ps, *t = job.target + (size_t)b * job.ts;
    const float *m = job.mask ? job.mask + (size_t)b * job.ms : nullptr;
    float v[3] = {0.0f, 0.0f, 0.0f};
    for (int i = blockIdx.x * NT + threadIdx.x; i < N; i += SIL_NB * NT) {
        float mk = m ? m[i] : 1.0f;
        float ld = logf(p[i] + 1e-7f) * mk - logf(t[i] + 1e-7f) * mk;
        v[0] += ld;
        v[1] += ld * ld;
        v[2] += mk;
    }
    const float tot = block_sum_many<NT, 3>(v, scratch);
    if (threadIdx.x < 3) ws[(((size_t)jb * B + b) * SIL_NB + blockIdx.x) * 4 + threadIdx.x] = tot;
}
// block = job: its images folded as k_silog_finish folds them; the last job to arrive (n_jobs tickets in all) adds the
// losses in job order.  (Folding by the last partial block of a job instead -- one launch -- was measured: 108 us, its
// 768 tickets per job serialise on one address; two launches: 20 us.)
__global__ void __launch_bounds__(64) k_silog_many_finish(const float *__restrict__ ws, float *__restrict__ sums,
                                                          float *__restrict__ losses, float *__restrict__ total,
                                                          int *__restrict__ tickets, int n_jobs, int B, float beta)
{
    __shared__ double acc[64];
    const int jb = blockIdx.x;
    const float *wj = ws + (size_t)jb * B * SIL_NB * 4;
    double lt = 0.0;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int bb = b0 + (int)threadIdx.x;
        double lb = 0.0;
        if (bb < B) {
            double s1 = 0.0, s2 = 0.0, n = 0.0;
            for (int k = 0; k < SIL_NB; ++k) {
                const float *q = wj + ((size_t)bb * SIL_NB + k) * 4;
                s1 += q[0]; s2 += q[1]; n += q[2];
            }
            n += 1e-8;
            lb = s2 / n - (double)beta * s1 * s1 / (n * n);
            float *sj = sums + ((size_t)jb * B + bb) * 4;
            sj[0] = (float)s1; sj[1] = (float)s2; sj[2] = (float)n; sj[3] = 0.0f;
        }
        acc[threadIdx.x] = lb;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < 64; ++i) lt += acc[i];
        __syncthreads();
    }
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        publish(losses + jb, (float)(lt / (double)B));
        s_last = (take_ticket(tickets) == n_jobs - 1);
    }
    __syncthreads();
    if (!s_last) return;
    // (the jobs' losses fetched by as many lanes at once, summed in job order by one: a lane fetching them one after the
    // other pays a round trip to the device-coherent level per job)
    float tt = 0.0f;
    for (int k0 = 0; k0 < n_jobs; k0 += 64) {
        const int k = k0 + (int)threadIdx.x;
        if (k < n_jobs) acc[threadIdx.x] = (double)fetch_published(losses + k);
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < min(64, n_jobs - k0); ++i) tt += (float)acc[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        total[0] = tt;
        publish(tickets, 0);
    }
}
// element-wise adjoint of every job: upstream gradient = g_total[0] (nullable) + g_losses[job] (nullable)
__global__ void __launch_bounds__(NT) k_silog_many_bwd(SilogJobs J, const float *__restrict__ sums,
                                                       const float *__restrict__ g_total,
                                                       const float *__restrict__ g_losses, int B, int N, float beta)
{
    const int b = blockIdx.y, jb = blockIdx.z;
    const SilogJob &job = J.j[jb];
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    const float mk = job.mask ? job.mask[(size_t)b * job.ms + i] : 1.0f;
    const float lp = job.pred[(size_t)b * job.ps + i] + 1e-7f, lt = job.target[(size_t)b * job.ts + i] + 1e-7f;
    const float ld = logf(lp) * mk - logf(lt) * mk;
    const float *sj = sums + ((size_t)jb * B + b) * 4;
    const float s1 = sj[0], n = sj[2];
    const float gl = (g_total ? g_total[0] : 0.0f) + (g_losses ? g_losses[jb] : 0.0f);
    const float g = gl / (float)B * (2.0f * ld / n - 2.0f * beta * s1 / (n * n));
    const size_t o = (size_t)b * N + i;
    if (job.g_pred) job.g_pred[o] = g * mk / lp;
    if (job.g_target) job.g_target[o] = -g * mk / lt;
}

// ---------------------------------------------------------------- Conv3x3's ReflectionPad2d(1)
// reference: layers.py:121-138 (every 3x3 convolution of the decoders pads by reflection
// first; the tensors are the largest of the step, up to [72,16,192,640]).  Forward: one lane
// per output element of the padded plane, lanes walk x (coalesced stores, near-coalesced
// loads).  Backward: a gather -- each input element sums its own copy and the reflected
// copies that land on it (rows 1 and H-2, columns 1 and W-2) -- so no atomics.
// Bound: HBM (read N, write N + border).
constexpr int RP_PL = 8;     // planes per lane: index math once, 8 independent loads in flight

__global__ void __launch_bounds__(NT) k_reflect_pad1_fwd(const float *__restrict__ in,
                                                         float *__restrict__ out, int planes, int H,
                                                         int W)
{
    const int Wp = W + 2, Hp = H + 2;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= Hp * Wp) return;
    const int Y = i / Wp, X = i - Y * Wp;
    int y = Y - 1, x = X - 1;
    y = (y < 0) ? -y : ((y >= H) ? 2 * (H - 1) - y : y);
    x = (x < 0) ? -x : ((x >= W) ? 2 * (W - 1) - x : x);
    const size_t p0 = (size_t)blockIdx.y * RP_PL;
    const int np = min(RP_PL, planes - (int)p0);
    const float *src = in + p0 * H * W + (size_t)y * W + x;
    float *dst = out + p0 * Hp * Wp + i;
    float v[RP_PL];
#pragma unroll
    for (int k = 0; k < RP_PL; ++k) v[k] = (k < np) ? src[(size_t)k * H * W] : 0.0f;
#pragma unroll
    for (int k = 0; k < RP_PL; ++k)
        if (k < np) dst[(size_t)k * Hp * Wp] = v[k];
}

__global__ void __launch_bounds__(NT) k_reflect_pad1_bwd(const float *__restrict__ g_out,
                                                         float *__restrict__ g_in, int planes, int H,
                                                         int W)
{
    const int Wp = W + 2;
    const size_t PP = (size_t)(H + 2) * Wp;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    // padded rows / columns that map onto (y, x): its own, plus the reflected border ones
    int rows[3], cols[3], nr = 1, nc = 1;
    rows[0] = y + 1;
    cols[0] = x + 1;
    if (y == 1) rows[nr++] = 0;
    if (y == H - 2) rows[nr++] = H + 1;
    if (x == 1) cols[nc++] = 0;
    if (x == W - 2) cols[nc++] = W + 1;
    const size_t p0 = (size_t)blockIdx.y * RP_PL;
    const int np = min(RP_PL, planes - (int)p0);
    const float *g = g_out + p0 * PP;
    float *dst = g_in + p0 * H * W + i;
    if (nr <= 2 && nc <= 2) {
        // own copy for every lane, with all plane loads in flight; the (few) border lanes add
        // the reflected copies afterwards
        const size_t o = (size_t)(y + 1) * Wp + x + 1;
        float v[RP_PL];
#pragma unroll
        for (int k = 0; k < RP_PL; ++k) v[k] = (k < np) ? g[(size_t)k * PP + o] : 0.0f;
        if (nr > 1 || nc > 1) {
            const size_t oc = (size_t)(y + 1) * Wp + cols[nc - 1];
            const size_t orw = (size_t)rows[nr - 1] * Wp + x + 1;
            const size_t orc = (size_t)rows[nr - 1] * Wp + cols[nc - 1];
#pragma unroll
            for (int k = 0; k < RP_PL; ++k) {
                if (k >= np) continue;
                const float *gk = g + (size_t)k * PP;
                if (nc > 1) v[k] += gk[oc];
                if (nr > 1) v[k] += gk[orw];
                if (nr > 1 && nc > 1) v[k] += gk[orc];
            }
        }
#pragma unroll
        for (int k = 0; k < RP_PL; ++k)
            if (k < np) dst[(size_t)k * H * W] = v[k];
        return;
    }
    for (int k = 0; k < np; ++k) {      // H or W == 3: a row / column is the image of both borders
        float acc = 0.0f;
        for (int a = 0; a < nr; ++a)
            for (int c = 0; c < nc; ++c) acc += g[(size_t)k * PP + (size_t)rows[a] * Wp + cols[c]];
        dst[(size_t)k * H * W] = acc;
    }
}

// ---------------------------------------------------------------- a10 pose glue
// reference: layers.py:28-103.  One lane per batch element.
struct Rot {
    float R[9];
};

MVF_DEV void rodrigues(const float v[3], float R[9], float &angle, float axis[3], float &sa,
                       float &ca)
{
    angle = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    float den = angle + 1e-7f;
    float x = v[0] / den, y = v[1] / den, z = v[2] / den;
    axis[0] = x; axis[1] = y; axis[2] = z;
    ca = cosf(angle);
    sa = sinf(angle);
    float C = 1.0f - ca;
    float xs = x * sa, ys = y * sa, zs = z * sa;
    float xC = x * C, yC = y * C, zC = z * C;
    float xyC = x * yC, yzC = y * zC, zxC = z * xC;
    R[0] = x * xC + ca; R[1] = xyC - zs;     R[2] = zxC + ys;
    R[3] = xyC + zs;    R[4] = y * yC + ca;  R[5] = yzC - xs;
    R[6] = zxC - ys;    R[7] = yzC + xs;     R[8] = z * zC + ca;
}

__global__ void k_pose_fwd(const float *__restrict__ aa, const float *__restrict__ tr,
                           float *__restrict__ M, int invert, int B)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float v[3] = {aa[b * 3], aa[b * 3 + 1], aa[b * 3 + 2]};
    float t[3] = {tr[b * 3], tr[b * 3 + 1], tr[b * 3 + 2]};
    float R[9], angle, axis[3], sa, ca;
    rodrigues(v, R, angle, axis, sa, ca);
    float *m = M + b * 16;
    if (!invert) {
        // M = T * R : rotation block R, translation column t
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            m[i * 4 + 0] = R[i * 3 + 0];
            m[i * 4 + 1] = R[i * 3 + 1];
            m[i * 4 + 2] = R[i * 3 + 2];
            m[i * 4 + 3] = t[i];
        }
    } else {
        // M = R^T * T(-t) : rotation block R^T, translation column R^T * (-t)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float r0 = R[0 * 3 + i], r1 = R[1 * 3 + i], r2 = R[2 * 3 + i];
            m[i * 4 + 0] = r0;
            m[i * 4 + 1] = r1;
            m[i * 4 + 2] = r2;
            // 4x4 @ 4x4: the reference's small-matrix loop, no fma (see proj_entry)
            float a = r0 * (-t[0]);
            a = a + r1 * (-t[1]);
            a = a + r2 * (-t[2]);
            m[i * 4 + 3] = a;
        }
    }
    m[12] = 0.0f; m[13] = 0.0f; m[14] = 0.0f; m[15] = 1.0f;
}

__global__ void k_pose_bwd(const float *__restrict__ aa, const float *__restrict__ tr,
                           const float *__restrict__ gM, float *__restrict__ g_aa,
                           float *__restrict__ g_tr, int invert, int B)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float v[3] = {aa[b * 3], aa[b * 3 + 1], aa[b * 3 + 2]};
    float t[3] = {tr[b * 3], tr[b * 3 + 1], tr[b * 3 + 2]};
    float R[9], angle, ax[3], sa, ca;
    rodrigues(v, R, angle, ax, sa, ca);
    const float *g = gM + b * 16;
    // gradient w.r.t. R (3x3) and t
    float gR[9], gt[3];
    if (!invert) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) gR[i * 3 + j] = g[i * 4 + j];
            gt[i] = g[i * 4 + 3];
        }
    } else {
        // M[i][j] = R[j][i] ; M[i][3] = -sum_j R[j][i] t[j]
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                gR[j * 3 + i] = g[i * 4 + j] - g[i * 4 + 3] * t[j];
                acc += -g[i * 4 + 3] * R[j * 3 + i];
            }
            gt[j] = acc;
        }
    }
    // R entries in terms of (x,y,z, sa, ca, C=1-ca)
    float x = ax[0], y = ax[1], z = ax[2], C = 1.0f - ca;
    // d/dx, d/dy, d/dz, d/dsa, d/dca (C depends on ca: dC/dca = -1)
    float gx = gR[0] * 2.0f * x * C + (gR[1] + gR[3]) * y * C + (gR[2] + gR[6]) * z * C +
               (gR[7] - gR[5]) * sa;
    float gy = gR[4] * 2.0f * y * C + (gR[1] + gR[3]) * x * C + (gR[5] + gR[7]) * z * C +
               (gR[2] - gR[6]) * sa;
    float gz = gR[8] * 2.0f * z * C + (gR[2] + gR[6]) * x * C + (gR[5] + gR[7]) * y * C +
               (gR[3] - gR[1]) * sa;
    float gsa = (gR[3] - gR[1]) * z + (gR[2] - gR[6]) * y + (gR[7] - gR[5]) * x;
    float gC = gR[0] * x * x + gR[4] * y * y + gR[8] * z * z + (gR[1] + gR[3]) * x * y +
               (gR[2] + gR[6]) * z * x + (gR[5] + gR[7]) * y * z;
    float gca = gR[0] + gR[4] + gR[8] - gC;
    // angle = |v| ; axis = v / (angle + 1e-7) ; sa = sin(angle) ; ca = cos(angle)
    float g_angle = gsa * ca - gca * sa;
    float den = angle + 1e-7f;
    float gax[3] = {gx, gy, gz};
    float dot = gax[0] * v[0] + gax[1] * v[1] + gax[2] * v[2];
    g_angle += -dot / (den * den);
    // d angle / d v = v / angle  (torch.norm backward gives 0 at angle == 0)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float dn = (angle > 0.0f) ? v[i] / angle : 0.0f;
        g_aa[b * 3 + i] = gax[i] / den + g_angle * dn;
        g_tr[b * 3 + i] = gt[i];
    }
}

}  // namespace

// ---------------------------------------------------------------- measurement hooks
#include <algorithm>
#include <mutex>
#include <vector>
namespace mvf {
namespace {
struct ProfRec {
    hipEvent_t e0, e1;
    int64_t work;
    int tag;
};
struct ProfDone {
    double ms;
    int64_t work;
    int tag;
};
struct ProfState {
    std::mutex mu;
    int level = 0;
    std::vector<ProfRec> ev[MVF_PROF_COUNT];
    std::vector<ProfDone> done[MVF_PROF_COUNT];
    hipEvent_t open_start[MVF_PROF_COUNT] = {};
    double acc_ms[MVF_PROF_COUNT] = {};
    int64_t acc_n[MVF_PROF_COUNT] = {};
    int64_t acc_work[MVF_PROF_COUNT] = {};
};
ProfState &prof() { static ProfState p; return p; }
constexpr size_t kMaxPairs = 1 << 14, kMaxDone = 1 << 16;

void drain(ProfState &p, int id)
{
    for (auto &r : p.ev[id]) {
        float ms = 0.0f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            p.acc_ms[id] += ms;
            p.acc_n[id] += 1;
            if (p.done[id].size() < kMaxDone) p.done[id].push_back({(double)ms, r.work, r.tag});
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    p.ev[id].clear();
}
inline bool prof_active(const ProfState &p, int id)
{
    return p.level >= 2 || (p.level == 1 && id < MVF_PROF_UNITS_FINISH);
}
}  // namespace

void prof_begin(int id, hipStream_t st)
{
    ProfState &p = prof();
    if (!p.level) return;
    std::lock_guard<std::mutex> g(p.mu);
    if (!prof_active(p, id)) return;
    if (p.ev[id].size() >= kMaxPairs) drain(p, id);
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    p.open_start[id] = e;
}

void prof_end(int id, hipStream_t st, int64_t work, int tag)
{
    ProfState &p = prof();
    if (!p.level) return;
    std::lock_guard<std::mutex> g(p.mu);
    if (!p.open_start[id]) return;
    p.acc_work[id] += work;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    (void)hipEventRecord(e, st);
    p.ev[id].push_back({p.open_start[id], e, work, tag});
    p.open_start[id] = nullptr;
}
}  // namespace mvf

// shared with mvf_photo.hip: fold the fused backward's per-tile grad_P partials
namespace mvf_geom {
int finish_gT(const float *ws, const float *K, float *gT, int B, int S, int nblk, void *stream)
{
    hipLaunchKernelGGL(k_finish_gT, dim3(B, S), dim3(NT), 0, (hipStream_t)stream, ws, K, gT, nblk);
    return hip_check_launch();
}
}  // namespace mvf_geom

// wide adjoint of the reflection pad (mvf_glue.hip)
namespace mvf_glue {
bool reflect_pad1_fwd_wide(const float *in, float *out, int planes, int H, int W, hipStream_t st);
bool reflect_pad1_bwd_wide(const float *g_out, float *g_in, int planes, int H, int W, hipStream_t st);
}

// ============================================================================ C ABI
extern "C" {

int mvf_abi_version(void) { return MVF_ABI_VERSION; }

int mvf_profile_enable(int level)
{
    auto &p = mvf::prof();
    std::lock_guard<std::mutex> g(p.mu);
    p.level = level < 0 ? 0 : level;
    return 0;
}

int mvf_profile_reset(void)
{
    auto &p = mvf::prof();
    std::lock_guard<std::mutex> g(p.mu);
    for (int i = 0; i < MVF_PROF_COUNT; ++i) {
        mvf::drain(p, i);
        p.done[i].clear();
        p.acc_ms[i] = 0.0;
        p.acc_n[i] = 0;
        p.acc_work[i] = 0;
    }
    return 0;
}

int mvf_profile_read(int id, double *total_ms, int64_t *launches)
{
    if (id < 0 || id >= MVF_PROF_COUNT) return (int)hipErrorInvalidValue;
    auto &p = mvf::prof();
    std::lock_guard<std::mutex> g(p.mu);
    mvf::drain(p, id);
    if (total_ms) *total_ms = p.acc_ms[id];
    if (launches) *launches = p.acc_n[id];
    return 0;
}

int mvf_profile_read_work(int id, int64_t *work)
{
    if (id < 0 || id >= MVF_PROF_COUNT || !work) return (int)hipErrorInvalidValue;
    auto &p = mvf::prof();
    std::lock_guard<std::mutex> g(p.mu);
    *work = p.acc_work[id];
    return 0;
}

int64_t mvf_profile_read_launches(int id, double *ms, int64_t *work, int32_t *tag, int64_t cap)
{
    if (id < 0 || id >= MVF_PROF_COUNT || cap < 0) return -(int64_t)hipErrorInvalidValue;
    auto &p = mvf::prof();
    std::lock_guard<std::mutex> g(p.mu);
    mvf::drain(p, id);
    const int64_t n = std::min<int64_t>(cap, (int64_t)p.done[id].size());
    for (int64_t i = 0; i < n; ++i) {
        if (ms) ms[i] = p.done[id][i].ms;
        if (work) work[i] = p.done[id][i].work;
        if (tag) tag[i] = p.done[id][i].tag;
    }
    return n;
}

const char *mvf_profile_name(int id)
{
    static const char *const names[MVF_PROF_COUNT] = {
        "k_photo_fwd<fused>", "k_photo_bwd<fused>", "k_photo_fwd", "k_photo_bwd", "k_warp_fwd", "k_warp_bwd",
        "k_unit_fb", "k_units_finish", "k_fb_scale", "k_disp_mean", "k_bias_act_fwd", "k_bias_act_bwd",
        "k_act_bwd_flat", "k_up2cat_pad_fwd", "k_up2cat_pad_bwd_x", "k_up2cat_pad_bwd_skip", "k_reflect_pad1_fwd",
        "k_reflect_pad1_bwd", "k_maxpool3s2_fwd", "k_maxpool3s2_bwd", "k_fusion_level_fwd",
        "k_fusion_level_bwd_gather", "k_flow_warp_fwd", "k_disp_head_fwd", "k_disp_head_bwd",
        "k_resize_bilinear_fwd", "k_resize_bilinear_bwd", "k_upsample_nearest_fwd", "k_upsample_nearest_bwd",
        "k_silog_fwd", "k_silog_bwd", "k_affine", "k_regroup_fwd", "k_regroup_bwd", "k_interleave_fwd", "k_sum_act_fwd"};
    return (id >= 0 && id < MVF_PROF_COUNT) ? names[id] : "?";
}

const char *mvf_error_string(int err) { return hipGetErrorString((hipError_t)err); }

size_t mvf_workspace_floats(int B, int H, int W)
{
    size_t nblk = ((size_t)H * W + NT - 1) / NT;
    // (+ B*16: mvf_unit_fwdbwd keeps its ticket counters behind the one-unit workspace)
    return (size_t)B * nblk * 16 * MVF_MAX_SRC + 1024 + (size_t)B * 16;
}

int mvf_disp_to_depth_fwd(const float *disp, float *scaled, float *depth, int64_t n,
                          float min_disp, float range, void *stream)
{
    if (n <= 0) return 0;
    int64_t blocks = (n + NT - 1) / NT;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_disp_to_depth, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream,
                       disp, scaled, depth, n, min_disp, range);
    return hip_check_launch();
}

int mvf_disp_to_depth_bwd(const float *disp, const float *g_scaled, const float *g_depth,
                          float *g_disp, int64_t n, float min_disp, float range, void *stream)
{
    if (n <= 0) return 0;
    int64_t blocks = (n + NT - 1) / NT;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_disp_to_depth_bwd, dim3((unsigned)blocks), dim3(NT), 0,
                       (hipStream_t)stream, disp, g_scaled, g_depth, g_disp, n, min_disp, range);
    return hip_check_launch();
}

int mvf_backproject_fwd(const float *depth, const float *inv_K, float *cam, int B, int H, int W,
                        void *stream)
{
    if (B * H * W <= 0) return 0;
    hipLaunchKernelGGL(k_backproject, pix_grid(H, W, B), dim3(NT), 0, (hipStream_t)stream, depth,
                       inv_K, cam, H, W);
    return hip_check_launch();
}

int mvf_backproject_bwd(const float *g_cam, const float *inv_K, float *g_depth, int B, int H,
                        int W, void *stream)
{
    if (B * H * W <= 0) return 0;
    hipLaunchKernelGGL(k_backproject_bwd, pix_grid(H, W, B), dim3(NT), 0, (hipStream_t)stream,
                       g_cam, inv_K, g_depth, H, W);
    return hip_check_launch();
}

int mvf_project_fwd(const float *cam, const float *K, const float *T, float *pix, int B, int H,
                    int W, float eps, void *stream)
{
    if (B * H * W <= 0) return 0;
    hipLaunchKernelGGL(k_project, pix_grid(H, W, B), dim3(NT), 0, (hipStream_t)stream, cam, K, T,
                       pix, H, W, eps);
    return hip_check_launch();
}

int mvf_project_bwd(const float *cam, const float *K, const float *T, const float *g_pix,
                    float *g_cam, float *g_T, float *workspace, int B, int H, int W, float eps,
                    void *stream)
{
    if (B * H * W <= 0) return 0;
    dim3 grid = pix_grid(H, W, B);
    hipLaunchKernelGGL(k_project_bwd, grid, dim3(NT), 0, (hipStream_t)stream, cam, K, T, g_pix,
                       g_cam, g_T ? workspace : nullptr, H, W, eps);
    if (g_T)
        hipLaunchKernelGGL(k_finish_gT, dim3(B, 1), dim3(NT), 0, (hipStream_t)stream, workspace, K,
                           g_T, (int)grid.x);
    return hip_check_launch();
}

int mvf_grid_sample_fwd(const float *img, const float *grid, float *out, int32_t *idx_xy, int B,
                        int C, int H, int W, void *stream)
{
    if (B * H * W <= 0) return 0;
    hipLaunchKernelGGL(k_grid_sample, pix_grid(H, W, B), dim3(NT), 0, (hipStream_t)stream, img,
                       grid, out, idx_xy, C, H, W);
    return hip_check_launch();
}

int mvf_grid_sample_bwd(const float *img, const float *grid, const float *g_out, float *g_grid,
                        float *g_img, int B, int C, int H, int W, void *stream)
{
    if (B * H * W <= 0) return 0;
    hipLaunchKernelGGL(k_grid_sample_bwd, pix_grid(H, W, B), dim3(NT), 0, (hipStream_t)stream, img,
                       grid, g_out, g_grid, g_img, C, H, W);
    return hip_check_launch();
}

int mvf_warp_fwd(const float *disp, const float *inv_K, const float *K, const float *T,
                 const float *src, float *warped, float *pix, int32_t *idx_xy, int B, int H,
                 int W, float min_disp, float range, float eps, void *stream)
{
    if (B * H * W <= 0) return 0;
    {
        ProfScope ps(MVF_PROF_WARP_FWD, (hipStream_t)stream);
        hipLaunchKernelGGL(k_warp_fwd, pix_grid(H, W, B), dim3(NT), 0, (hipStream_t)stream, disp,
                           inv_K, K, T, src, warped, pix, idx_xy, H, W, min_disp, range, eps);
    }
    return hip_check_launch();
}

int mvf_warp_bwd(const float *disp, const float *inv_K, const float *K, const float *T,
                 const float *src, const float *g_warped, float *g_disp, float *g_T, float *g_src,
                 float *workspace, int accumulate, int B, int H, int W, float min_disp,
                 float range, float eps, void *stream)
{
    if (B * H * W <= 0) return 0;
    dim3 grid = pix_grid(H, W, B);
    {
        ProfScope ps(MVF_PROF_WARP_BWD, (hipStream_t)stream);
        hipLaunchKernelGGL(k_warp_bwd, grid, dim3(NT), 0, (hipStream_t)stream, disp, inv_K, K, T,
                           src, g_warped, g_disp, g_src, workspace, accumulate, H, W, min_disp,
                           range, eps);
    }
    hipLaunchKernelGGL(k_finish_gT, dim3(B, 1), dim3(NT), 0, (hipStream_t)stream, workspace, K,
                       g_T, (int)grid.x);
    return hip_check_launch();
}

int mvf_flow_warp_fwd(const float *img, const float *flow, const float *xs, const float *ys,
                      float *out, int32_t *idx_xy, int B, int C, int H, int W, void *stream)
{
    if (B * C * H * W <= 0) return 0;
    dim3 grid((unsigned)((H * W + NT - 1) / NT), (unsigned)((C + FW_CCH - 1) / FW_CCH), (unsigned)B);
    // image read once, flow read once, warped image written once
    ProfScope ps(MVF_PROF_FLOW_WARP_FWD, stream, 4LL * B * H * W * (2LL * C + 2));
    hipLaunchKernelGGL(k_flow_warp_fwd, grid, dim3(NT), 0, (hipStream_t)stream, img, flow, xs, ys, out,
                       idx_xy, C, H, W);
    return hip_check_launch();
}

size_t mvf_flow_warp_workspace_floats(int B, int C, int H, int W)
{
    return (size_t)((C + FW_CCH - 1) / FW_CCH) * B * 2 * H * W;
}

int mvf_flow_warp_bwd(const float *img, const float *flow, const float *xs, const float *ys,
                      const float *g_out, float *g_img, float *g_flow, float *workspace, int B, int C,
                      int H, int W, void *stream)
{
    if (B * C * H * W <= 0) return 0;
    const int nchunk = (C + FW_CCH - 1) / FW_CCH;
    dim3 grid((unsigned)((H * W + NT - 1) / NT), (unsigned)nchunk, (unsigned)B);
    hipLaunchKernelGGL(k_flow_warp_bwd, grid, dim3(NT), 0, (hipStream_t)stream, img, flow, xs, ys,
                       g_out, g_img, g_flow ? workspace : nullptr, C, H, W);
    if (g_flow) {
        size_t n = (size_t)B * 2 * H * W;
        hipLaunchKernelGGL(k_flow_grad_fold, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0,
                           (hipStream_t)stream, workspace, g_flow, nchunk, n);
    }
    return hip_check_launch();
}

int mvf_silog_fwd(const float *pred, const float *target, const float *mask, float *loss,
                  float *sums, float *workspace, int B, int N, float beta, void *stream)
{
    if (B <= 0 || N <= 0) return 0;
    if (B > 65535) return (int)hipErrorInvalidValue;   // grid.y
    ProfScope ps(MVF_PROF_SILOG_FWD, stream, 4LL * B * N * (2 + (mask ? 1 : 0)));
    hipLaunchKernelGGL(k_silog_partial, dim3(SIL_NB, B), dim3(NT), 0, (hipStream_t)stream, pred, target,
                       mask, workspace, N);
    hipLaunchKernelGGL(k_silog_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, workspace, loss, sums,
                       B, beta);
    return hip_check_launch();
}

int mvf_silog_bwd(const float *pred, const float *target, const float *mask, const float *sums,
                  const float *g_loss, float *g_pred, float *g_target, int B, int N, float beta,
                  void *stream)
{
    if (B <= 0 || N <= 0) return 0;
    ProfScope ps(MVF_PROF_SILOG_BWD, stream, 4LL * B * N * (2 + (mask ? 1 : 0) + (g_pred ? 1 : 0) + (g_target ? 1 : 0)));
    hipLaunchKernelGGL(k_silog_bwd, dim3((unsigned)((N + NT - 1) / NT), (unsigned)B), dim3(NT), 0,
                       (hipStream_t)stream, pred, target, mask, sums, g_loss, g_pred, g_target, B, N, beta);
    return hip_check_launch();
}

size_t mvf_silog_many_workspace_floats(int n_jobs, int B) { return (size_t)n_jobs * B * SIL_NB * 4; }

static int silog_jobs(const mvf_silog_job *jobs, int n_jobs, int N, bool bwd, SilogJobs &J)
{
    if (n_jobs < 1 || n_jobs > MVF_MAX_SILOG_JOBS || !jobs) return (int)hipErrorInvalidValue;
    J.n = n_jobs;
    for (int i = 0; i < n_jobs; ++i) {
        const mvf_silog_job &d = jobs[i];
        if (!d.pred || !d.target) return (int)hipErrorInvalidValue;
        SilogJob &j = J.j[i];
        j.pred = d.pred; j.target = d.target; j.mask = d.mask;
        j.ps = d.pred_stride ? (size_t)d.pred_stride : (size_t)N;
        j.ts = d.target_stride ? (size_t)d.target_stride : (size_t)N;
        j.ms = d.mask_stride ? (size_t)d.mask_stride : (size_t)N;
        j.g_pred = bwd ? d.g_pred : nullptr; j.g_target = bwd ? d.g_target : nullptr;
    }
    return 0;
}

int mvf_silog_many_fwd(const mvf_silog_job *jobs, int n_jobs, float *losses, float *total, float *sums,
                       float *workspace, int32_t *tickets, int B, int N, float beta, void *stream)
{
    if (B <= 0 || N <= 0) return 0;
    if (B > 65535 || !losses || !total || !sums || !workspace || !tickets) return (int)hipErrorInvalidValue;
    SilogJobs J = {};
    if (int e = silog_jobs(jobs, n_jobs, N, false, J)) return e;
    int64_t bytes = 0;
    for (int i = 0; i < n_jobs; ++i) bytes += 4LL * B * N * (2 + (jobs[i].mask ? 1 : 0));
    ProfScope ps(MVF_PROF_SILOG_FWD, stream, bytes);
    hipLaunchKernelGGL(k_silog_many_partial, dim3(SIL_NB, (unsigned)B, (unsigned)n_jobs), dim3(NT), 0, (hipStream_t)stream,
                       J, workspace, B, N);
    hipLaunchKernelGGL(k_silog_many_finish, dim3((unsigned)n_jobs), dim3(64), 0, (hipStream_t)stream, workspace, sums,
                       losses, total, tickets, n_jobs, B, beta);
    return hip_check_launch();
}

int mvf_silog_many_bwd(const mvf_silog_job *jobs, int n_jobs, const float *sums, const float *g_total,
                       const float *g_losses, int B, int N, float beta, void *stream)
{
    if (B <= 0 || N <= 0) return 0;
    if (B > 65535 || !sums || (!g_total && !g_losses)) return (int)hipErrorInvalidValue;
    SilogJobs J = {};
    if (int e = silog_jobs(jobs, n_jobs, N, true, J)) return e;
    int64_t bytes = 0;
    for (int i = 0; i < n_jobs; ++i)
        bytes += 4LL * B * N * (2 + (jobs[i].mask ? 1 : 0) + (jobs[i].g_pred ? 1 : 0) + (jobs[i].g_target ? 1 : 0));
    ProfScope ps(MVF_PROF_SILOG_BWD, stream, bytes);
    hipLaunchKernelGGL(k_silog_many_bwd, dim3((unsigned)((N + NT - 1) / NT), (unsigned)B, (unsigned)n_jobs), dim3(NT), 0,
                       (hipStream_t)stream, J, sums, g_total, g_losses, B, N, beta);
    return hip_check_launch();
}

int mvf_reflect_pad1_fwd(const float *in, float *out, int planes, int H, int W, void *stream)
{
    if (planes <= 0) return 0;
    if (!in || !out || H < 2 || W < 2 || planes > 65535 * RP_PL) return (int)hipErrorInvalidValue;
    const int n = (H + 2) * (W + 2);
    ProfScope ps(MVF_PROF_REFLECT_PAD_FWD, stream, 4LL * planes * ((int64_t)H * W + n));
    if (mvf_glue::reflect_pad1_fwd_wide(in, out, planes, H, W, (hipStream_t)stream)) return hip_check_launch();
    hipLaunchKernelGGL(k_reflect_pad1_fwd, dim3((unsigned)((n + NT - 1) / NT), (unsigned)((planes + RP_PL - 1) / RP_PL)),
                       dim3(NT), 0, (hipStream_t)stream, in, out, planes, H, W);
    return hip_check_launch();
}

int mvf_reflect_pad1_bwd(const float *g_out, float *g_in, int planes, int H, int W, void *stream)
{
    if (planes <= 0) return 0;
    if (!g_out || !g_in || H < 2 || W < 2 || planes > 65535 * RP_PL) return (int)hipErrorInvalidValue;
    const int n = H * W;
    ProfScope ps(MVF_PROF_REFLECT_PAD_BWD, stream, 4LL * planes * ((int64_t)n + (int64_t)(H + 2) * (W + 2)));
    if (mvf_glue::reflect_pad1_bwd_wide(g_out, g_in, planes, H, W, (hipStream_t)stream)) return hip_check_launch();
    hipLaunchKernelGGL(k_reflect_pad1_bwd, dim3((unsigned)((n + NT - 1) / NT), (unsigned)((planes + RP_PL - 1) / RP_PL)),
                       dim3(NT), 0, (hipStream_t)stream, g_out, g_in, planes, H, W);
    return hip_check_launch();
}

int mvf_pose_fwd(const float *axisangle, const float *translation, float *M, int invert, int B,
                 void *stream)
{
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_pose_fwd, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, axisangle,
                       translation, M, invert, B);
    return hip_check_launch();
}

int mvf_pose_bwd(const float *axisangle, const float *translation, const float *g_M,
                 float *g_axisangle, float *g_translation, int invert, int B, void *stream)
{
    if (B <= 0) return 0;
    hipLaunchKernelGGL(k_pose_bwd, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, axisangle,
                       translation, g_M, g_axisangle, g_translation, invert, B);
    return hip_check_launch();
}

}  // extern "C"
