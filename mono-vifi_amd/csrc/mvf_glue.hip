// mvf_glue.hip -- SURVEY.md section 8f-4: step glue either side of the hot path.
//
// (1) Decoder glue.  Every stage of the depth decoder does
//         x = upsample(upconv_0(x)) ; x = cat([x, skip], 1) ; x = upconv_1(x)     (networks/monodepth2.py:84-90)
//     where upconv_1 is Conv3x3 = ReflectionPad2d(1) + conv (layers.py:121-138) and upsample is
//     nearest x2 (layers.py:225-228).  As stock ops that is three full passes over the largest
//     tensors of the step (nearest upsample, cat copy, reflection pad) forward and three
//     backward; here the padded convolution input [B, C1+C2, 2h+2, 2w+2] is written ONCE from x
//     [B,C1,h,w] and the skip feature [B,C2,2h,2w], and the adjoint is one gather pass per source
//     (reflect-pad adjoint + 2x2 fold for x; reflect-pad adjoint for the skip) -- no atomics.
// (2) Disparity head.  outputs[("disp", s)] = sigmoid(dispconv(x)) (monodepth2.py:93) followed by
//     disp_to_depth (layers.py:16-25, called train.py:961 and by the depth-consistency losses):
//     one pass emits disp, depth and the per-image mean partials of disp the unit kernel needs.
// Bound: HBM streaming.  -ffp-contract=off (mvf_common.hpp); sigmoid as ATen writes it,
// 1 / (1 + exp(-x)).
#include <cstdlib>

#include "mvf_common.hpp"

using namespace mvf;

namespace {

constexpr int NT = 256;
constexpr int PL = 8;            // planes per lane: index math once, PL independent loads in flight
constexpr int NMEAN = 32;        // per-image mean partials (matches mvf_photo.hip / mvf_unit_fb.hip)

MVF_DEV int reflect1(int v, int n)
{
    return (v < 0) ? -v : ((v >= n) ? 2 * (n - 1) - v : v);
}

// out[b, c, Y, X] = src(c)[reflect(Y-1), reflect(X-1)], src = nearest-x2 of x for c < C1, skip else.
// grid (padded pixels, plane chunks of the concatenated channel axis, B)
__global__ void __launch_bounds__(NT) k_up2cat_pad_fwd(const float *__restrict__ x,
                                                       const float *__restrict__ skip,
                                                       float *__restrict__ out, int C1, int C2, int h, int w)
{
    const int H = 2 * h, W = 2 * w, Hp = H + 2, Wp = W + 2;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= Hp * Wp) return;
    const int Y = i / Wp, X = i - Y * Wp;
    const int y = reflect1(Y - 1, H), xx = reflect1(X - 1, W);
    const int b = blockIdx.z, C = C1 + C2;
    const int c0 = blockIdx.y * PL;
    float *dst = out + ((size_t)b * C + c0) * Hp * Wp + i;
    const size_t n1 = (size_t)h * w, n2 = (size_t)H * W;
    const size_t o1 = (size_t)(y >> 1) * w + (xx >> 1), o2 = (size_t)y * W + xx;
    float v[PL];
#pragma unroll
    for (int k = 0; k < PL; ++k) {
        const int c = c0 + k;
        v[k] = 0.0f;
        if (c < C1) v[k] = x[((size_t)b * C1 + c) * n1 + o1];
        else if (c < C) v[k] = skip[((size_t)b * C2 + (c - C1)) * n2 + o2];
    }
#pragma unroll
    for (int k = 0; k < PL; ++k)
        if (c0 + k < C) dst[(size_t)k * Hp * Wp] = v[k];
}

// adjoint of ReflectionPad2d(1) at (y, x) of an H x W plane: own copy + the reflected border copies
MVF_DEV float pad_adjoint(const float *__restrict__ g, int y, int x, int H, int W)
{
    const int Wp = W + 2;
    float v = g[(size_t)(y + 1) * Wp + x + 1];
    const int ry = (y == 1) ? 0 : ((y == H - 2) ? H + 1 : -1);       // H >= 4: at most one reflected row
    const int rx = (x == 1) ? 0 : ((x == W - 2) ? W + 1 : -1);
    if (rx >= 0) v += g[(size_t)(y + 1) * Wp + rx];
    if (ry >= 0) {
        v += g[(size_t)ry * Wp + x + 1];
        if (rx >= 0) v += g[(size_t)ry * Wp + rx];
    }
    return v;
}

// g_x[b,c,Y,X] = sum over the 2x2 children of pad_adjoint(g[b,c]) ; grid (h*w, plane chunks of C1, B)
__global__ void __launch_bounds__(NT) k_up2cat_pad_bwd_x(const float *__restrict__ g,
                                                         float *__restrict__ g_x, int C1, int C2, int h, int w)
{
    const int H = 2 * h, W = 2 * w;
    const size_t PP = (size_t)(H + 2) * (W + 2);
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= h * w) return;
    const int Y = i / w, X = i - Y * w;
    const int b = blockIdx.z, C = C1 + C2, c0 = blockIdx.y * PL;
    float v[PL];
#pragma unroll
    for (int k = 0; k < PL; ++k) {
        v[k] = 0.0f;
        if (c0 + k < C1) {
            const float *gp = g + ((size_t)b * C + c0 + k) * PP;
            v[k] = (pad_adjoint(gp, 2 * Y, 2 * X, H, W) + pad_adjoint(gp, 2 * Y, 2 * X + 1, H, W)) +
                   (pad_adjoint(gp, 2 * Y + 1, 2 * X, H, W) + pad_adjoint(gp, 2 * Y + 1, 2 * X + 1, H, W));
        }
    }
#pragma unroll
    for (int k = 0; k < PL; ++k)
        if (c0 + k < C1) g_x[((size_t)b * C1 + c0 + k) * h * w + i] = v[k];
}

// g_skip[b,c,y,x] = pad_adjoint(g[b, C1 + c]) ; grid (H*W, plane chunks of C2, B)
__global__ void __launch_bounds__(NT) k_up2cat_pad_bwd_skip(const float *__restrict__ g,
                                                            float *__restrict__ g_skip, int C1, int C2, int h,
                                                            int w)
{
    const int H = 2 * h, W = 2 * w;
    const size_t PP = (size_t)(H + 2) * (W + 2);
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    const int b = blockIdx.z, C = C1 + C2, c0 = blockIdx.y * PL;
    float v[PL];
#pragma unroll
    for (int k = 0; k < PL; ++k)
        v[k] = (c0 + k < C2) ? pad_adjoint(g + ((size_t)b * C + C1 + c0 + k) * PP, y, x, H, W) : 0.0f;
#pragma unroll
    for (int k = 0; k < PL; ++k)
        if (c0 + k < C2) g_skip[((size_t)b * C2 + c0 + k) * H * W + i] = v[k];
}

// ---- the pad adjoints, wide form (round 4) ------------------------------------------------------------------
// The one-element-per-lane gathers above move 4 bytes per lane and instruction (0.40-0.51 of the HBM peak in the
// training step).  Here a lane owns FOUR consecutive columns of an unpadded row: its own copies are one 16-byte load
// from the padded plane (4-byte aligned: a padded row is W + 2 floats), the reflected border copies are added by the
// few lanes that sit on rows 1 / H-2 or hold columns 1 / W-2, and the result leaves as one aligned 16-byte store.
// Same additions in the same order as pad_adjoint.  Needs W % 4 == 0, W >= 8, H >= 4 (else the forms above run).
struct __attribute__((packed, aligned(4))) F4U {
    float x, y, z, w;
};
MVF_DEV float4 ldg4u(const float *__restrict__ p)
{
    const F4U v = *reinterpret_cast<const F4U *>(p);
    return make_float4(v.x, v.y, v.z, v.w);
}
MVF_DEV float4 pad_adj4(const float *__restrict__ g, int y, int x0, int H, int W)
{
    const int Wp = W + 2;
    const float *row = g + (size_t)(y + 1) * Wp;
    float4 v = ldg4u(row + x0 + 1);
    const bool first = x0 == 0, last = x0 + 4 == W;
    if (first) v.y += row[0];                  // column 1 also receives padded column 0
    if (last) v.z += row[W + 1];               // column W-2 also receives padded column W+1
    const int ry = (y == 1) ? 0 : ((y == H - 2) ? H + 1 : -1);
    if (ry >= 0) {
        const float *rr = g + (size_t)ry * Wp;
        const float4 r = ldg4u(rr + x0 + 1);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        if (first) v.y += rr[0];
        if (last) v.z += rr[W + 1];
    }
    return v;
}
constexpr int PADW_PL = 4;      // planes per lane
// dst plane p (of `planes`) <- padded source plane (p / Cd) * Cs + Coff + p % Cd; grid (H * W/4 blocks, plane chunks)
__global__ void __launch_bounds__(NT) k_pad1_bwd_w4(const float *__restrict__ g, float *__restrict__ dst, int planes,
                                                    int Cd, int Cs, int Coff, int H, int W)
{
    const int W4 = W >> 2;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H * W4) return;
    const int y = i / W4, x0 = (i - y * W4) * 4;
    const size_t PP = (size_t)(H + 2) * (W + 2), n = (size_t)H * W;
    const int p0 = blockIdx.y * PADW_PL;
    float4 v[PADW_PL];
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k) {
        const int p = min(p0 + k, planes - 1);
        const int b = p / Cd, c = p - b * Cd;
        v[k] = pad_adj4(g + ((size_t)b * Cs + Coff + c) * PP, y, x0, H, W);
    }
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k)
        if (p0 + k < planes) *reinterpret_cast<float4 *>(dst + (size_t)(p0 + k) * n + (size_t)y * W + x0) = v[k];
}
// g_x of the x2-upsampled part: two outputs per lane = the 2 x 4 children of rows 2Y, 2Y+1
__global__ void __launch_bounds__(NT) k_up2cat_pad_bwd_x_w4(const float *__restrict__ g, float *__restrict__ g_x, int planes,
                                                            int C1, int C, int h, int w)
{
    const int H = 2 * h, W = 2 * w, w2 = w >> 1;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= h * w2) return;
    const int Y = i / w2, j = i - Y * w2;
    const size_t PP = (size_t)(H + 2) * (W + 2), n = (size_t)h * w;
    const int p0 = blockIdx.y * PADW_PL;
    float4 a[PADW_PL], bq[PADW_PL];
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k) {
        const int p = min(p0 + k, planes - 1);
        const int b = p / C1, c = p - b * C1;
        const float *gp = g + ((size_t)b * C + c) * PP;
        a[k] = pad_adj4(gp, 2 * Y, 4 * j, H, W);
        bq[k] = pad_adj4(gp, 2 * Y + 1, 4 * j, H, W);
    }
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k)
        if (p0 + k < planes)
            *reinterpret_cast<float2 *>(g_x + (size_t)(p0 + k) * n + (size_t)Y * w + 2 * j) =
                make_float2((a[k].x + a[k].y) + (bq[k].x + bq[k].y), (a[k].z + a[k].w) + (bq[k].z + bq[k].w));
}
// forward counterpart: four consecutive columns of an unpadded row go out as one (4-byte aligned) 16-byte store
// into the padded row; the lanes on the border also write the reflected copies (rows 0 / H+1, columns 0 / W+1)
MVF_DEV void stg4u(float *__restrict__ p, float4 v)
{
    F4U t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    *reinterpret_cast<F4U *>(p) = t;
}
MVF_DEV void pad_put4(float *__restrict__ dst, int y, int x0, float4 v, int H, int W)
{
    const int Wp = W + 2;
    float *row = dst + (size_t)(y + 1) * Wp;
    const bool first = x0 == 0, last = x0 + 4 == W;
    stg4u(row + x0 + 1, v);
    if (first) row[0] = v.y;
    if (last) row[W + 1] = v.z;
    const int ry = (y == 1) ? 0 : ((y == H - 2) ? H + 1 : -1);
    if (ry >= 0) {
        float *rr = dst + (size_t)ry * Wp;
        stg4u(rr + x0 + 1, v);
        if (first) rr[0] = v.y;
        if (last) rr[W + 1] = v.z;
    }
}
__global__ void __launch_bounds__(NT) k_pad1_fwd_w4(const float *__restrict__ in, float *__restrict__ out, int planes, int H,
                                                    int W)
{
    const int W4 = W >> 2;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H * W4) return;
    const int y = i / W4, x0 = (i - y * W4) * 4;
    const size_t PP = (size_t)(H + 2) * (W + 2), n = (size_t)H * W;
    const int p0 = blockIdx.y * PADW_PL;
    float4 v[PADW_PL];
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k)
        v[k] = *reinterpret_cast<const float4 *>(in + (size_t)min(p0 + k, planes - 1) * n + (size_t)y * W + x0);
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k)
        if (p0 + k < planes) pad_put4(out + (size_t)(p0 + k) * PP, y, x0, v[k], H, W);
}
// grid (H * W/4 blocks, chunks of the C1 + C2 output planes, B)
__global__ void __launch_bounds__(NT) k_up2cat_pad_fwd_w4(const float *__restrict__ x, const float *__restrict__ skip,
                                                          float *__restrict__ out, int C1, int C2, int h, int w)
{
    const int H = 2 * h, W = 2 * w, W4 = W >> 2;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H * W4) return;
    const int y = i / W4, j = i - y * W4, x0 = 4 * j;
    const int b = blockIdx.z, C = C1 + C2, c0 = blockIdx.y * PADW_PL;
    const size_t PP = (size_t)(H + 2) * (W + 2), n1 = (size_t)h * w, n2 = (size_t)H * W;
    float4 v[PADW_PL];
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k) {
        const int c = min(c0 + k, C - 1);
        if (c < C1) {
            const float2 t = *reinterpret_cast<const float2 *>(x + ((size_t)b * C1 + c) * n1 + (size_t)(y >> 1) * w + 2 * j);
            v[k] = make_float4(t.x, t.x, t.y, t.y);
        } else {
            v[k] = *reinterpret_cast<const float4 *>(skip + ((size_t)b * C2 + (c - C1)) * n2 + (size_t)y * W + x0);
        }
    }
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k)
        if (c0 + k < C) pad_put4(out + ((size_t)b * C + c0 + k) * PP, y, x0, v[k], H, W);
}

// ---- round 5: the convolution epilogue inside its consumer ---------------------------------
// In the depth decoder every ConvBlock output (networks/monodepth2.py:84-93, layers.py:106-118: conv + bias + ELU) is
// read next by exactly one pad kernel (the next Conv3x3's ReflectionPad2d(1), or upsample + cat + pad).  The pad kernels
// can apply `ELU(y + bias[c])` on load: the activated tensor is then written ONCE, already padded, instead of written
// by an epilogue pass, read back and written again padded (16 -> 8 bytes per element forward).  Backward: the padded
// tensor (saved by the convolution that consumes it anyway) holds the activated values in its interior, so one kernel
// gathers the pad adjoint, multiplies by ELU'(out) = out > 0 ? 1 : out + 1 and forms the bias gradient's partial
// sums (20 -> 12 bytes per element).  Wide form only (W % 4 == 0 ...): other shapes keep the separate kernels.
MVF_DEV float4 bias_elu4(float4 v, float bz)
{
    v.x += bz; v.y += bz; v.z += bz; v.w += bz;
    v.x = (v.x > 0.0f) ? v.x : expm1f(v.x);
    v.y = (v.y > 0.0f) ? v.y : expm1f(v.y);
    v.z = (v.z > 0.0f) ? v.z : expm1f(v.z);
    v.w = (v.w > 0.0f) ? v.w : expm1f(v.w);
    return v;
}
__global__ void __launch_bounds__(NT) k_pad1_act_fwd_w4(const float *__restrict__ in, const float *__restrict__ bias,
                                                        float *__restrict__ out, int planes, int C, int H, int W)
{
    const int W4 = W >> 2;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H * W4) return;
    const int y = i / W4, x0 = (i - y * W4) * 4;
    const size_t PP = (size_t)(H + 2) * (W + 2), n = (size_t)H * W;
    const int p0 = blockIdx.y * PADW_PL;
    float4 v[PADW_PL];
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k) {
        const int p = min(p0 + k, planes - 1);
        v[k] = bias_elu4(*reinterpret_cast<const float4 *>(in + (size_t)p * n + (size_t)y * W + x0), bias[p % C]);
    }
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k)
        if (p0 + k < planes) pad_put4(out + (size_t)(p0 + k) * PP, y, x0, v[k], H, W);
}
// (the x2-upsampled part gets bias + ELU, the skip feature is copied)
__global__ void __launch_bounds__(NT) k_up2cat_pad_act_fwd_w4(const float *__restrict__ x, const float *__restrict__ bias,
                                                              const float *__restrict__ skip, float *__restrict__ out,
                                                              int C1, int C2, int h, int w)
{
    const int H = 2 * h, W = 2 * w, W4 = W >> 2;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H * W4) return;
    const int y = i / W4, j = i - y * W4, x0 = 4 * j;
    const int b = blockIdx.z, C = C1 + C2, c0 = blockIdx.y * PADW_PL;
    const size_t PP = (size_t)(H + 2) * (W + 2), n1 = (size_t)h * w, n2 = (size_t)H * W;
    float4 v[PADW_PL];
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k) {
        const int c = min(c0 + k, C - 1);
        if (c < C1) {
            const float2 t = *reinterpret_cast<const float2 *>(x + ((size_t)b * C1 + c) * n1 + (size_t)(y >> 1) * w + 2 * j);
            const float4 a = bias_elu4(make_float4(t.x, t.y, 0.0f, 0.0f), bias[c]);
            v[k] = make_float4(a.x, a.x, a.y, a.y);
        } else {
            v[k] = *reinterpret_cast<const float4 *>(skip + ((size_t)b * C2 + (c - C1)) * n2 + (size_t)y * W + x0);
        }
    }
#pragma unroll
    for (int k = 0; k < PADW_PL; ++k)
        if (c0 + k < C) pad_put4(out + ((size_t)b * C + c0 + k) * PP, y, x0, v[k], H, W);
}
MVF_DEV float elu_grad(float g, float o) { return (o > 0.0f) ? g : g * (o + 1.0f); }
// block (chunk, plane): up to 4 x 256 column quads of one plane, one shot; part[(n * chunks + chunk) * C + c]
constexpr int PADA_U = 4;
__global__ void __launch_bounds__(NT) k_pad1_act_bwd_w4(const float *__restrict__ g, const float *__restrict__ padded,
                                                        float *__restrict__ gx, float *__restrict__ part, int C, int H, int W)
{
    __shared__ float scratch[NT / kWave];
    const int W4 = W >> 2, per = H * W4;
    const int plane = blockIdx.y, n = plane / C, c = plane - n * C;
    const size_t PP = (size_t)(H + 2) * (W + 2);
    const float *gp = g + (size_t)plane * PP, *op = padded + (size_t)plane * PP;
    float *dst = gx + (size_t)plane * H * W;
    float4 gv[PADA_U], ov[PADA_U];
    int yy[PADA_U], xx[PADA_U];
#pragma unroll
    for (int k = 0; k < PADA_U; ++k) {
        const int i = min((int)blockIdx.x * (NT * PADA_U) + k * NT + (int)threadIdx.x, per - 1);
        yy[k] = i / W4; xx[k] = (i - yy[k] * W4) * 4;
        gv[k] = pad_adj4(gp, yy[k], xx[k], H, W);
        ov[k] = ldg4u(op + (size_t)(yy[k] + 1) * (W + 2) + xx[k] + 1);
    }
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < PADA_U; ++k) {
        if ((int)blockIdx.x * (NT * PADA_U) + k * NT + (int)threadIdx.x >= per) continue;
        const float4 t = make_float4(elu_grad(gv[k].x, ov[k].x), elu_grad(gv[k].y, ov[k].y), elu_grad(gv[k].z, ov[k].z),
                                     elu_grad(gv[k].w, ov[k].w));
        *reinterpret_cast<float4 *>(dst + (size_t)yy[k] * W + xx[k]) = t;
        acc += (t.x + t.y) + (t.z + t.w);
    }
    const float r = block_sum<NT>(acc, scratch);
    if (threadIdx.x == 0) part[((size_t)n * gridDim.x + blockIdx.x) * C + c] = r;
}
// g_x of the x2-upsampled, activated part: (sum over the 2 x 2 children of the pad adjoint) * ELU'(x_act), x_act read
// from the padded tensor's interior (a child of the pixel); block (chunk, plane of B * C1)
__global__ void __launch_bounds__(NT) k_up2cat_pad_act_bwd_x_w4(const float *__restrict__ g, const float *__restrict__ padded,
                                                                float *__restrict__ g_x, float *__restrict__ part, int C1,
                                                                int C, int h, int w)
{
    __shared__ float scratch[NT / kWave];
    const int H = 2 * h, W = 2 * w, w2 = w >> 1, per = h * w2;
    const int plane = blockIdx.y, b = plane / C1, c = plane - b * C1;
    const size_t PP = (size_t)(H + 2) * (W + 2);
    const float *gp = g + ((size_t)b * C + c) * PP, *op = padded + ((size_t)b * C + c) * PP;
    float *dst = g_x + (size_t)plane * h * w;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = (int)blockIdx.x * (NT * 2) + k * NT + (int)threadIdx.x;
        if (i >= per) continue;
        const int Y = i / w2, j = i - Y * w2;
        const float4 a = pad_adj4(gp, 2 * Y, 4 * j, H, W), bq = pad_adj4(gp, 2 * Y + 1, 4 * j, H, W);
        const float4 o = ldg4u(op + (size_t)(2 * Y + 1) * (W + 2) + 4 * j + 1);
        const float2 t = make_float2(elu_grad((a.x + a.y) + (bq.x + bq.y), o.x), elu_grad((a.z + a.w) + (bq.z + bq.w), o.z));
        *reinterpret_cast<float2 *>(dst + (size_t)Y * w + 2 * j) = t;
        acc += t.x + t.y;
    }
    const float r = block_sum<NT>(acc, scratch);
    if (threadIdx.x == 0) part[((size_t)b * gridDim.x + blockIdx.x) * C1 + c] = r;
}

static bool pad_wide_ok(const void *padded, const void *plain, int H, int W)
{
    return (W & 3) == 0 && W >= 8 && H >= 4 && (((uintptr_t)plain) & 15) == 0 && (((uintptr_t)padded) & 3) == 0 &&
           !getenv("MVF_PAD_NARROW");
}

// ---- disparity head: disp = sigmoid(logit), depth = 1 / (min_disp + range * disp), mean partials
// grid (NMEAN, B): block (chunk, b) walks its slice of the image
__global__ void __launch_bounds__(NT) k_disp_head_fwd(const float *__restrict__ logit,
                                                      float *__restrict__ disp, float *__restrict__ depth,
                                                      float *__restrict__ mean_part, int N, float min_disp,
                                                      float range)
{
    __shared__ float scratch[NT / kWave];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const int per = (N + NMEAN - 1) / NMEAN;
    const int lo = chunk * per, hi = min(lo + per, N);
    const size_t base = (size_t)b * N;
    // same accumulation pattern as k_disp_mean (mvf_photo.hip): the unit kernels read these partials
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    auto one = [&](int i) {
        const float d = 1.0f / (1.0f + expf(-logit[base + i]));
        disp[base + i] = d;
        if (depth) depth[base + i] = 1.0f / (min_disp + range * d);
        return d;
    };
    int i = lo + threadIdx.x;
    for (; i + 3 * NT < hi; i += 4 * NT) {
        s0 += one(i);
        s1 += one(i + NT);
        s2 += one(i + 2 * NT);
        s3 += one(i + 3 * NT);
    }
    for (; i < hi; i += NT) s0 += one(i);
    const float r = block_sum<NT>((s0 + s1) + (s2 + s3), scratch);
    if (threadIdx.x == 0 && mean_part) mean_part[b * NMEAN + chunk] = r;
}

// g_logit = (g_disp + g_depth * d depth/d disp) * disp * (1 - disp),  d depth/d disp = -range * depth^2
__global__ void __launch_bounds__(NT) k_disp_head_bwd(const float *__restrict__ disp,
                                                      const float *__restrict__ g_disp,
                                                      const float *__restrict__ g_depth,
                                                      float *__restrict__ g_logit, int64_t n, float min_disp,
                                                      float range)
{
    int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * NT;
    for (; i < n; i += stride) {
        const float d = disp[i];
        float g = g_disp ? g_disp[i] : 0.0f;
        if (g_depth) {
            const float dep = 1.0f / (min_disp + range * d);
            g -= g_depth[i] * range * dep * dep;
        }
        g_logit[i] = g * d * (1.0f - d);
    }
}

// The same adjoint with the hot-path units' disparity gradients taken RAW, as the unit kernel left them (upstream
// gradient 1, without the per-image shift of the mean-normalised smoothness term): (raw - shift_b) * g is applied on
// load -- the pass k_fb_scale makes over every unit's gradient for a consumer that needs a tensor -- and the units' images
// are read where the unit launch wrote them: no scaled copy, no re-interleaving `stack` of the group gradients.
// grid (chunks, images of the head's batch); a unit covers the images first, first + step, ... (count of them).
struct HeadUnit {
    const float *g_raw, *stats, *g_loss, *g_sum;
    size_t raw_stride;
    float smoothness;
    int first, step, count;
};
struct HeadUnits {
    HeadUnit u[MVF_MAX_UNITS];
    int n;
};
__global__ void __launch_bounds__(NT) k_disp_head_bwd_units(const float *__restrict__ disp,
                                                            const float *__restrict__ g_disp,
                                                            const float *__restrict__ g_depth,
                                                            float *__restrict__ g_logit, int N, int vec, float min_disp,
                                                            float range, HeadUnits hu)
{
    const int img = blockIdx.y;
    const size_t base = (size_t)img * N;
    // the (at most a few) units that cover this image: block-uniform
    const float *raw[MVF_MAX_UNITS];
    float shift[MVF_MAX_UNITS], gs[MVF_MAX_UNITS];
    int m = 0;
    for (int e = 0; e < hu.n; ++e) {
        const HeadUnit &u = hu.u[e];
        const int rel = img - u.first;
        if (rel < 0 || rel % u.step != 0 || rel / u.step >= u.count) continue;
        const int b = rel / u.step;
        // shift_b and g exactly as k_fb_scale forms them (mvf_unit_fb.hip)
        const float den = u.stats[b * 4 + 1];
        const float smooth_b = u.stats[b * 4 + 2] + u.stats[b * 4 + 3];
        shift[m] = (u.smoothness * smooth_b / (float)N) / den;
        gs[m] = (u.g_loss ? u.g_loss[0] : 0.0f) + (u.g_sum ? u.g_sum[0] : 0.0f);
        raw[m] = u.g_raw + (size_t)b * u.raw_stride;
        ++m;
    }
    auto one = [&](float d, float g, float gdep) {
        if (g_depth) {
            const float dep = 1.0f / (min_disp + range * d);
            g -= gdep * range * dep * dep;
        }
        return g * d * (1.0f - d);
    };
    if (vec) {
        const int i = blockIdx.x * NT + threadIdx.x;
        if (i >= N / 4) return;
        const float4 d = reinterpret_cast<const float4 *>(disp + base)[i];
        float4 g = g_disp ? reinterpret_cast<const float4 *>(g_disp + base)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < m; ++k) {
            const float4 r = reinterpret_cast<const float4 *>(raw[k])[i];
            const float4 t = make_float4((r.x - shift[k]) * gs[k], (r.y - shift[k]) * gs[k], (r.z - shift[k]) * gs[k],
                                         (r.w - shift[k]) * gs[k]);
            if (k == 0 && !g_disp) g = t;          // (0 + t would turn a -0 into +0: keep the bits of the two-kernel path)
            else { g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w; }
        }
        const float4 gd = g_depth ? reinterpret_cast<const float4 *>(g_depth + base)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4 *>(g_logit + base)[i] =
            make_float4(one(d.x, g.x, gd.x), one(d.y, g.y, gd.y), one(d.z, g.z, gd.z), one(d.w, g.w, gd.w));
    } else {
        for (int i = blockIdx.x * NT + threadIdx.x; i < N; i += gridDim.x * NT) {
            float g = g_disp ? g_disp[base + i] : 0.0f;
            for (int k = 0; k < m; ++k) {
                const float t = (raw[k][i] - shift[k]) * gs[k];
                g = (k == 0 && !g_disp) ? t : g + t;
            }
            g_logit[base + i] = one(disp[base + i], g, g_depth ? g_depth[base + i] : 0.0f);
        }
    }
}

// ---- convolution epilogue: bias + activation (+ residual) --------------------------------------
// Every biased convolution of the step is `conv -> + bias[c] -> activation` (decoder ConvBlock:
// layers.py:106-118 ELU; IFRNet convrelu / ResBlock: networks/IFRNet.py:128-157 PReLU, with the
// block input added before the last PReLU; transposed convolutions and 1x1 merges: bias only).
// As stock ops the bias is a strided broadcast add_ (one full pass), the activation another, and
// in backward the bias gradient a separate reduction over the activation's gradient.  Here the
// convolution runs without bias and one pass does  out = act(x + bias[c] (+ res));  backward one
// pass does  g_x = g * act'(out)  AND the per-channel partial sums of g_x (deterministic: fixed
// split, partials folded in index order by k_bias_grad_finish).
// act: 0 none, 1 ELU(alpha=1), 2 ReLU, 3 PReLU (slope per channel, or one slope if slope_n == 1).
// ELU as ATen writes it: x > 0 ? x : expm1(x); its derivative from the result: out > 0 ? 1 : out + 1.
enum { ACT_NONE = 0, ACT_ELU = 1, ACT_RELU = 2, ACT_PRELU = 3 };

MVF_DEV float act_apply(float v, int act, float slope)
{
    switch (act) {
    case ACT_ELU: return (v > 0.0f) ? v : expm1f(v);
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_PRELU: return (v > 0.0f) ? v : slope * v;
    default: return v;
    }
}

// flat over the tensor: lane i handles float4 (VEC) or scalar element i, i + NT, ...; the channel of an
// element is (i / plane) % C (one integer divide per access: the pass is HBM-bound)
template <bool VEC>
__global__ void __launch_bounds__(NT) k_bias_act_fwd(const float *__restrict__ x, const float *__restrict__ bias,
                                                     const float *__restrict__ slope,
                                                     const float *__restrict__ res, float *__restrict__ out,
                                                     int C, int per_plane, int64_t total, int act, int slope_n)
{
    constexpr int U = 4;
    const int64_t i0 = (int64_t)blockIdx.x * (NT * U) + threadIdx.x;
    if (VEC) {
        const float4 *xp = reinterpret_cast<const float4 *>(x);
        const float4 *rp = res ? reinterpret_cast<const float4 *>(res) : nullptr;
        float4 *op = reinterpret_cast<float4 *>(out);
        float4 v[U], r[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int64_t i = i0 + k * NT;
            if (i < total) {
                v[k] = xp[i];
                if (rp) r[k] = rp[i];
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int64_t i = i0 + k * NT;
            if (i >= total) continue;
            const int c = (int)(((unsigned)i / (unsigned)per_plane) % (unsigned)C);     // total < 2^31 (launcher)
            const float bv = bias ? bias[c] : 0.0f;
            const float sv = (act == ACT_PRELU) ? slope[slope_n == 1 ? 0 : c] : 0.0f;
            float4 t = v[k];
            t.x += bv; t.y += bv; t.z += bv; t.w += bv;
            if (rp) { t.x += r[k].x; t.y += r[k].y; t.z += r[k].z; t.w += r[k].w; }
            op[i] = make_float4(act_apply(t.x, act, sv), act_apply(t.y, act, sv), act_apply(t.z, act, sv),
                                act_apply(t.w, act, sv));
        }
    } else {
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int64_t i = i0 + k * NT;
            if (i >= total) continue;
            const int c = (int)(((unsigned)i / (unsigned)per_plane) % (unsigned)C);
            float t = x[i] + (bias ? bias[c] : 0.0f);
            if (res) t += res[i];
            out[i] = act_apply(t, act, (act == ACT_PRELU) ? slope[slope_n == 1 ? 0 : c] : 0.0f);
        }
    }
}

// out = act(((t0 + t1) + t2) + ...): the branch sum of an HRNet fuse layer (reference networks/hrnet_encoder.py:
// `y = y + ...` per branch, then ReLU) in one pass -- the term-at-a-time form reads and writes the accumulator once
// per branch (11 tensor passes for four branches; 5 here).  Same left-to-right sum, same bits.
constexpr int SUM_MAX = 8;
MVF_DEV float relu_nan(float v) { return (v < 0.0f) ? 0.0f : v; }       // ATen's relu: NaN stays NaN, -0.0 stays -0.0
struct SumTerms {
    const float *t[SUM_MAX];
    int n;
};
template <bool VEC>
__global__ void __launch_bounds__(NT) k_sum_act_fwd(SumTerms ts, float *__restrict__ out, int64_t total, int act)
{
    constexpr int U = 4;
    const int64_t i0 = (int64_t)blockIdx.x * (NT * U) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int64_t i = i0 + k * NT;
        if (i >= total) continue;
        if (VEC) {
            float4 a = reinterpret_cast<const float4 *>(ts.t[0])[i];
            for (int j = 1; j < ts.n; ++j) {
                const float4 b = reinterpret_cast<const float4 *>(ts.t[j])[i];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            if (act == ACT_RELU) { a.x = relu_nan(a.x); a.y = relu_nan(a.y); a.z = relu_nan(a.z); a.w = relu_nan(a.w); }
            reinterpret_cast<float4 *>(out)[i] = a;
        } else {
            float a = ts.t[0][i];
            for (int j = 1; j < ts.n; ++j) a += ts.t[j][i];
            out[i] = (act == ACT_RELU) ? relu_nan(a) : a;
        }
    }
}

// g_x = g * act'(out) alone (no bias to reduce for): flat, ELU / ReLU
template <bool VEC>
__global__ void __launch_bounds__(NT) k_act_bwd_flat(const float *__restrict__ g, const float *__restrict__ out,
                                                     float *__restrict__ gx, int64_t total, int act)
{
    constexpr int U = 4;
    const int64_t i0 = (int64_t)blockIdx.x * (NT * U) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int64_t i = i0 + k * NT;
        if (i >= total) continue;
        if (VEC) {
            float4 gv = reinterpret_cast<const float4 *>(g)[i];
            const float4 ov = reinterpret_cast<const float4 *>(out)[i];
            if (act == ACT_ELU) {
                gv.x = (ov.x > 0.0f) ? gv.x : gv.x * (ov.x + 1.0f);
                gv.y = (ov.y > 0.0f) ? gv.y : gv.y * (ov.y + 1.0f);
                gv.z = (ov.z > 0.0f) ? gv.z : gv.z * (ov.z + 1.0f);
                gv.w = (ov.w > 0.0f) ? gv.w : gv.w * (ov.w + 1.0f);
            } else {
                gv.x = (ov.x > 0.0f) ? gv.x : 0.0f;
                gv.y = (ov.y > 0.0f) ? gv.y : 0.0f;
                gv.z = (ov.z > 0.0f) ? gv.z : 0.0f;
                gv.w = (ov.w > 0.0f) ? gv.w : 0.0f;
            }
            reinterpret_cast<float4 *>(gx)[i] = gv;
        } else {
            const float ov = out[i], gv = g[i];
            gx[i] = (ov > 0.0f) ? gv : ((act == ACT_ELU) ? gv * (ov + 1.0f) : 0.0f);
        }
    }
}

// block (c, s): the s-th of `nsplit` equal runs of channel c's N*HW elements (n-major).
// g_x = g * act'(out) (not written for act none: it IS g), part[s][c] = sum of g_x over the run.
// Round 4: a lane keeps FOUR element groups in flight per iteration (the one-group loop left two 16-byte loads in
// flight per lane: 0.50 of the HBM peak) and the (image, offset) split of an element index is 32-bit arithmetic
// (N * HW / 4 < 2^31 is checked by the launcher; the 64-bit divide was ~60 instructions per 16 bytes).
MVF_DEV float4 act_grad4(float4 gv, float4 ov, int act)
{
    if (act == ACT_ELU) {
        gv.x = (ov.x > 0.0f) ? gv.x : gv.x * (ov.x + 1.0f);
        gv.y = (ov.y > 0.0f) ? gv.y : gv.y * (ov.y + 1.0f);
        gv.z = (ov.z > 0.0f) ? gv.z : gv.z * (ov.z + 1.0f);
        gv.w = (ov.w > 0.0f) ? gv.w : gv.w * (ov.w + 1.0f);
    } else {
        gv.x = (ov.x > 0.0f) ? gv.x : 0.0f;
        gv.y = (ov.y > 0.0f) ? gv.y : 0.0f;
        gv.z = (ov.z > 0.0f) ? gv.z : 0.0f;
        gv.w = (ov.w > 0.0f) ? gv.w : 0.0f;
    }
    return gv;
}
template <bool VEC>
__global__ void __launch_bounds__(NT) k_bias_act_bwd(const float *__restrict__ g, const float *__restrict__ out,
                                                     float *__restrict__ gx, float *__restrict__ part, int N, int C,
                                                     int HW, int act, int nsplit)
{
    __shared__ float scratch[NT / 16];
    constexpr int U = 4;
    const int c = blockIdx.x % C, s = blockIdx.x / C;
    const unsigned per_plane = VEC ? (unsigned)(HW >> 2) : (unsigned)HW;
    const unsigned total = (unsigned)N * per_plane;
    const unsigned per = (total + nsplit - 1) / nsplit;
    const unsigned lo = min((unsigned)s * per, total), hi = min(lo + per, total);
    float acc = 0.0f;
    for (unsigned i0 = lo + threadIdx.x; i0 < hi; i0 += U * NT) {
        size_t o[U];
        bool ok[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const unsigned i = i0 + k * NT;
            ok[k] = i < hi;
            const unsigned ic = ok[k] ? i : lo;
            const unsigned n = ic / per_plane, q = ic - n * per_plane;
            o[k] = ((size_t)n * C + c) * HW + (VEC ? (size_t)q * 4 : q);
        }
        if (VEC) {
            float4 gv[U], ov[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                gv[k] = *reinterpret_cast<const float4 *>(g + o[k]);
                if (act != ACT_NONE) ov[k] = *reinterpret_cast<const float4 *>(out + o[k]);
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (!ok[k]) continue;
                float4 t = gv[k];
                if (act != ACT_NONE) {
                    t = act_grad4(t, ov[k], act);
                    *reinterpret_cast<float4 *>(gx + o[k]) = t;
                }
                acc += (t.x + t.y) + (t.z + t.w);
            }
        } else {
            float gv[U], ov[U];
#pragma unroll
            for (int k = 0; k < U; ++k) {
                gv[k] = g[o[k]];
                if (act != ACT_NONE) ov[k] = out[o[k]];
            }
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (!ok[k]) continue;
                float t = gv[k];
                if (act != ACT_NONE) {
                    t = (ov[k] > 0.0f) ? t : ((act == ACT_ELU) ? t * (ov[k] + 1.0f) : 0.0f);
                    gx[o[k]] = t;
                }
                acc += t;
            }
        }
    }
    const float r = block_sum<NT>(acc, scratch);
    if (threadIdx.x == 0) part[(size_t)s * C + c] = r;
}

// The same pass, ONE SHOT per block (round 4): block (chunk, plane) handles 1024 float4 of one (n, c) plane with its
// four 16-byte loads per stream in flight and exits -- measured on this chip the many-short-blocks form streams at
// 0.69 of the HBM peak (k_act_bwd_flat, k_bias_act_fwd) where blocks that loop over long runs reach 0.48-0.50.
// part[(n * chunks + chunk) * C + c]; folded per channel in that order by k_bias_grad_finish_tree.
__global__ void __launch_bounds__(NT) k_bias_act_bwd_shot(const float4 *__restrict__ g, const float4 *__restrict__ out,
                                                          float4 *__restrict__ gx, float *__restrict__ part, int C,
                                                          int per_plane, int act)
{
    __shared__ float scratch[NT / kWave];
    constexpr int U = 4;
    const int plane = blockIdx.y, n = plane / C, c = plane - n * C;
    const size_t base = (size_t)plane * per_plane;
    const int q0 = blockIdx.x * (NT * U) + threadIdx.x;
    float4 gv[U], ov[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int q = q0 + k * NT;
        if (q < per_plane) {
            gv[k] = g[base + q];
            if (act != ACT_NONE) ov[k] = out[base + q];
        }
    }
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int q = q0 + k * NT;
        if (q >= per_plane) continue;
        float4 t = gv[k];
        if (act != ACT_NONE) {
            t = act_grad4(t, ov[k], act);
            gx[base + q] = t;
        }
        acc += (t.x + t.y) + (t.z + t.w);
    }
    const float r = block_sum<NT>(acc, scratch);
    if (threadIdx.x == 0) part[((size_t)n * gridDim.x + blockIdx.x) * C + c] = r;
}
// block c: lane t adds the partials t, t + 256, ... of channel c in index order, then one fixed-shape block sum
__global__ void __launch_bounds__(NT) k_bias_grad_finish_tree(const float *__restrict__ part, float *__restrict__ gb,
                                                              int C, int nsplit)
{
    __shared__ float scratch[NT / kWave];
    const int c = blockIdx.x;
    float a = 0.0f;
    for (int s = threadIdx.x; s < nsplit; s += NT) a += part[(size_t)s * C + c];
    const float r = block_sum<NT>(a, scratch);
    if (threadIdx.x == 0) gb[c] = r;
}

__global__ void __launch_bounds__(NT) k_bias_grad_finish(const float *__restrict__ part, float *__restrict__ gb,
                                                         int C, int nsplit)
{
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    float a = 0.0f;
    for (int s = 0; s < nsplit; ++s) a += part[(size_t)s * C + c];
    gb[c] = a;
}

// ---- bilinear resize of feature maps ----------------------------------------------------------
// F.interpolate(x, mode="bilinear") as the networks use it: HRNet's fuse layers
// (align_corners=True, coarse branch -> fine branch, up to 8x), Lite-Mono's decoder
// (scale_factor=2, align_corners=False).  ATen's NCHW kernel takes one thread per OUTPUT POSITION
// and loops over batch x channels inside it: at 48x160 with 96x18 planes that is 7,680 threads
// on 256 CUs (26 ms per DHRNet step for 42 launches).  Here one lane per output element, planes
// along grid.y.  Source index and weights as ATen writes them (UpSample.h:
// area_pixel_compute_source_index): align_corners ? scale*dst : max(scale*(dst+0.5)-0.5, 0);
// value = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11).
// Backward: deterministic gather.  The source index is monotonic in dst, so the output pixels
// that touch input row iy form a contiguous run; each candidate of a conservatively widened run
// is re-evaluated with the forward's own arithmetic and contributes its exact weight
// (no float atomics; ATen's backward scatters with atomicAdd).
struct Lin1 {
    int i0, i1;
    float w0, w1;
};
MVF_DEV Lin1 resize_lin(int dst, float scale, int n_in, int align)
{
    float src = align ? scale * (float)dst : scale * ((float)dst + 0.5f) - 0.5f;
    if (!align && src < 0.0f) src = 0.0f;
    Lin1 l;
    l.i0 = min((int)src, n_in - 1);
    l.i1 = (l.i0 < n_in - 1) ? l.i0 + 1 : l.i0;
    l.w1 = src - (float)l.i0;
    l.w0 = 1.0f - l.w1;
    return l;
}

// grid (output pixel blocks, plane chunks); PLR planes per lane share the index arithmetic
constexpr int PLR = 4;
__global__ void __launch_bounds__(NT) k_resize_bilinear_fwd(const float *__restrict__ x, float *__restrict__ out,
                                                            int planes, int ih, int iw, int oh, int ow,
                                                            float sh, float sw, int align)
{
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= oh * ow) return;
    const int oy = i / ow, ox = i - oy * ow;
    const Lin1 ly = resize_lin(oy, sh, ih, align), lx = resize_lin(ox, sw, iw, align);
    const int p0 = blockIdx.y * PLR;
    const size_t ni = (size_t)ih * iw, no = (size_t)oh * ow;
    const int a00 = ly.i0 * iw + lx.i0, a01 = ly.i0 * iw + lx.i1, a10 = ly.i1 * iw + lx.i0, a11 = ly.i1 * iw + lx.i1;
    float v[PLR][4];
#pragma unroll
    for (int k = 0; k < PLR; ++k) {
        if (p0 + k < planes) {
            const float *p = x + (size_t)(p0 + k) * ni;
            v[k][0] = p[a00]; v[k][1] = p[a01]; v[k][2] = p[a10]; v[k][3] = p[a11];
        }
    }
#pragma unroll
    for (int k = 0; k < PLR; ++k)
        if (p0 + k < planes)
            out[(size_t)(p0 + k) * no + i] = ly.w0 * (lx.w0 * v[k][0] + lx.w1 * v[k][1]) +
                                            ly.w1 * (lx.w0 * v[k][2] + lx.w1 * v[k][3]);
}

// candidate run of output indices whose taps can include input index `i` (widened by one)
MVF_DEV void resize_run(int i, float scale, int n_out, int align, int &lo, int &hi)
{
    if (!(scale > 0.0f)) { lo = 0; hi = n_out - 1; return; }
    const float off = align ? 0.0f : 0.5f;
    const float a = ((float)i - 1.0f + off) / scale - off, b = ((float)i + 1.0f + off) / scale - off;
    lo = max((int)floorf(a) - 1, 0);
    hi = min((int)ceilf(b) + 1, n_out - 1);
}
// weight of output index d for input index i (0 when d does not touch i)
MVF_DEV float resize_weight(int d, int i, float scale, int n_in, int align)
{
    const Lin1 l = resize_lin(d, scale, n_in, align);
    return ((l.i0 == i) ? l.w0 : 0.0f) + ((l.i1 == i) ? l.w1 : 0.0f);
}

// grid (input pixel blocks, plane chunks)
__global__ void __launch_bounds__(NT) k_resize_bilinear_bwd(const float *__restrict__ g, float *__restrict__ gx,
                                                            int planes, int ih, int iw, int oh, int ow, float sh,
                                                            float sw, int align)
{
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= ih * iw) return;
    const int iy = i / iw, ix = i - iy * iw;
    int ylo, yhi, xlo, xhi;
    resize_run(iy, sh, oh, align, ylo, yhi);
    resize_run(ix, sw, ow, align, xlo, xhi);
    // tighten the x run to the touching candidates (weights are zero outside anyway)
    while (xlo <= xhi && resize_weight(xlo, ix, sw, iw, align) == 0.0f) ++xlo;
    while (xhi >= xlo && resize_weight(xhi, ix, sw, iw, align) == 0.0f) --xhi;
    const int p0 = blockIdx.y * PLR;
    const size_t ni = (size_t)ih * iw, no = (size_t)oh * ow;
    float acc[PLR];
#pragma unroll
    for (int k = 0; k < PLR; ++k) acc[k] = 0.0f;
    for (int dy = ylo; dy <= yhi; ++dy) {
        const float wy = resize_weight(dy, iy, sh, ih, align);
        if (wy == 0.0f) continue;
        for (int dx = xlo; dx <= xhi; ++dx) {
            const float w = wy * resize_weight(dx, ix, sw, iw, align);
            const float *gp = g + (size_t)p0 * no + (size_t)dy * ow + dx;
#pragma unroll
            for (int k = 0; k < PLR; ++k)
                if (p0 + k < planes) acc[k] += w * gp[(size_t)k * no];
        }
    }
#pragma unroll
    for (int k = 0; k < PLR; ++k)
        if (p0 + k < planes) gx[(size_t)(p0 + k) * ni + i] = acc[k];
}

// The same adjoint as two separable passes (round 4): bilinear resizing is out = Ry x Rx^T, so g_x = Ry^T (g Rx).
// The one-pass gather above walks (2f+3) x (2f+1) candidate outputs per input pixel with a dependent load behind every
// weight test -- 0.08 of the HBM peak on the HRNet fuse layers (up to 8x: ~300 iterations per lane on a 6 x 20 input) and
// only ih * iw * planes / 4 lanes.  Pass 1 folds the rows: t[iy][ox] = sum over the output rows touching iy (a
// wave-uniform run, every load a coalesced 16-byte access along ox); pass 2 folds the columns of t (1/f of the size of
// g).  Same weights (the forward's own arithmetic), fixed order, no atomics; the order of the additions differs from
// the one-pass form (tolerance-level quantity).
template <bool VEC>
__global__ void __launch_bounds__(NT) k_resize_bwd_rows(const float *__restrict__ g, float *__restrict__ t, int planes,
                                                        int ih, int oh, int ow, float sh, int align)
{
    constexpr int PV = 2;                                   // planes per lane
    const int per = VEC ? ow >> 2 : ow;
    // lanes walk (iy, q) row-major: a row of the fine grid is often narrower than a workgroup (ow / 4 = 40 at 48 x 160)
    const int i = blockIdx.x * NT + threadIdx.x, p0 = blockIdx.y * PV;
    if (i >= ih * per) return;
    const int iy = i / per, q = i - iy * per;
    int ylo, yhi;
    resize_run(iy, sh, oh, align, ylo, yhi);
    const size_t no = (size_t)oh * ow, nt = (size_t)ih * ow;
    float4 acc[PV];
#pragma unroll
    for (int k = 0; k < PV; ++k) acc[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int dy = ylo; dy <= yhi; ++dy) {
        const float wy = resize_weight(dy, iy, sh, ih, align);
        if (wy == 0.0f) continue;
#pragma unroll
        for (int k = 0; k < PV; ++k) {
            const float *gp = g + (size_t)min(p0 + k, planes - 1) * no + (size_t)dy * ow;
            if (VEC) {
                const float4 v = reinterpret_cast<const float4 *>(gp)[q];
                acc[k].x += wy * v.x; acc[k].y += wy * v.y; acc[k].z += wy * v.z; acc[k].w += wy * v.w;
            } else {
                acc[k].x += wy * gp[q];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PV; ++k) {
        if (p0 + k >= planes) continue;
        float *tp = t + (size_t)(p0 + k) * nt + (size_t)iy * ow;
        if (VEC) reinterpret_cast<float4 *>(tp)[q] = acc[k];
        else tp[q] = acc[k].x;
    }
}
// grid (input pixel blocks, plane chunks): g_x[iy][ix] = sum over the columns of t touching ix
__global__ void __launch_bounds__(NT) k_resize_bwd_cols(const float *__restrict__ t, float *__restrict__ gx, int planes,
                                                        int ih, int iw, int ow, float sw, int align)
{
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= ih * iw) return;
    const int iy = i / iw, ix = i - iy * iw;
    int xlo, xhi;
    resize_run(ix, sw, ow, align, xlo, xhi);
    const int p0 = blockIdx.y * PLR;
    const size_t ni = (size_t)ih * iw, nt = (size_t)ih * ow;
    float acc[PLR];
#pragma unroll
    for (int k = 0; k < PLR; ++k) acc[k] = 0.0f;
    for (int dx = xlo; dx <= xhi; ++dx) {
        const float wx = resize_weight(dx, ix, sw, iw, align);
        if (wx == 0.0f) continue;
        const float *tp = t + (size_t)p0 * nt + (size_t)iy * ow + dx;
#pragma unroll
        for (int k = 0; k < PLR; ++k)
            if (p0 + k < planes) acc[k] += wx * tp[(size_t)k * nt];
    }
#pragma unroll
    for (int k = 0; k < PLR; ++k)
        if (p0 + k < planes) gx[(size_t)(p0 + k) * ni + i] = acc[k];
}

// nearest-neighbour upsampling by an integer factor f (layers.py:225-228 `upsample`, the DHRNet
// decoder's branch merges): out[Y][X] = x[Y/f][X/f]; adjoint = the f x f block sum (a gather).
// ATen's NCHW kernels are position-parallel like the bilinear ones (4.6 ms backward per DHRNet step).
__global__ void __launch_bounds__(NT) k_upsample_nearest_fwd(const float *__restrict__ x, float *__restrict__ out,
                                                             int planes, int ih, int iw, int f)
{
    const int oh = ih * f, ow = iw * f;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= oh * ow) return;
    const int oy = i / ow, ox = i - oy * ow;
    const int src = (oy / f) * iw + ox / f;
    const int p0 = blockIdx.y * PLR;
    const size_t ni = (size_t)ih * iw, no = (size_t)oh * ow;
    float v[PLR];
#pragma unroll
    for (int k = 0; k < PLR; ++k)
        if (p0 + k < planes) v[k] = x[(size_t)(p0 + k) * ni + src];
#pragma unroll
    for (int k = 0; k < PLR; ++k)
        if (p0 + k < planes) out[(size_t)(p0 + k) * no + i] = v[k];
}

__global__ void __launch_bounds__(NT) k_upsample_nearest_bwd(const float *__restrict__ g, float *__restrict__ gx,
                                                             int planes, int ih, int iw, int f)
{
    const int ow = iw * f;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= ih * iw) return;
    const int iy = i / iw, ix = i - iy * iw;
    const int p0 = blockIdx.y * PLR;
    const size_t ni = (size_t)ih * iw, no = ni * f * f;
    float acc[PLR];
#pragma unroll
    for (int k = 0; k < PLR; ++k) acc[k] = 0.0f;
    for (int dy = 0; dy < f; ++dy)
        for (int dx = 0; dx < f; ++dx) {
            const float *gp = g + (size_t)p0 * no + (size_t)(iy * f + dy) * ow + ix * f + dx;
#pragma unroll
            for (int k = 0; k < PLR; ++k)
                if (p0 + k < planes) acc[k] += gp[(size_t)k * no];
        }
#pragma unroll
    for (int k = 0; k < PLR; ++k)
        if (p0 + k < planes) gx[(size_t)(p0 + k) * ni + i] = acc[k];
}

// factor 2, even input width, aligned planes (every `upsample` of the decoders): a lane owns TWO input pixels = a 2 x 4
// output block -- one 8-byte access on the input side, two 16-byte accesses on the output side (the element-per-lane
// kernels above: 0.45 / 0.34 of the HBM peak).  Same values; the adjoint adds its four terms in the same order.
__global__ void __launch_bounds__(NT) k_upsample_nearest2_fwd_w(const float *__restrict__ x, float *__restrict__ out,
                                                                int planes, int ih, int iw)
{
    const int iw2 = iw >> 1, ow = iw * 2;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= ih * iw2) return;
    const int iy = i / iw2, j = i - iy * iw2;
    const int p0 = blockIdx.y * PLR;
    const size_t ni = (size_t)ih * iw, no = ni * 4;
    float2 v[PLR];
#pragma unroll
    for (int k = 0; k < PLR; ++k)
        v[k] = *reinterpret_cast<const float2 *>(x + (size_t)min(p0 + k, planes - 1) * ni + (size_t)iy * iw + 2 * j);
#pragma unroll
    for (int k = 0; k < PLR; ++k) {
        if (p0 + k >= planes) continue;
        float *op = out + (size_t)(p0 + k) * no + (size_t)(2 * iy) * ow + 4 * j;
        const float4 w = make_float4(v[k].x, v[k].x, v[k].y, v[k].y);
        *reinterpret_cast<float4 *>(op) = w;
        *reinterpret_cast<float4 *>(op + ow) = w;
    }
}
__global__ void __launch_bounds__(NT) k_upsample_nearest2_bwd_w(const float *__restrict__ g, float *__restrict__ gx,
                                                                int planes, int ih, int iw)
{
    const int iw2 = iw >> 1, ow = iw * 2;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= ih * iw2) return;
    const int iy = i / iw2, j = i - iy * iw2;
    const int p0 = blockIdx.y * PLR;
    const size_t ni = (size_t)ih * iw, no = ni * 4;
    float4 a[PLR], b[PLR];
#pragma unroll
    for (int k = 0; k < PLR; ++k) {
        const float *gp = g + (size_t)min(p0 + k, planes - 1) * no + (size_t)(2 * iy) * ow + 4 * j;
        a[k] = *reinterpret_cast<const float4 *>(gp);
        b[k] = *reinterpret_cast<const float4 *>(gp + ow);
    }
#pragma unroll
    for (int k = 0; k < PLR; ++k) {
        if (p0 + k >= planes) continue;
        // (dy, dx) = (0,0), (0,1), (1,0), (1,1), starting from 0.0f like the element-per-lane kernel
        const float2 r = make_float2((((0.0f + a[k].x) + a[k].y) + b[k].x) + b[k].y,
                                     (((0.0f + a[k].z) + a[k].w) + b[k].z) + b[k].w);
        *reinterpret_cast<float2 *>(gx + (size_t)(p0 + k) * ni + (size_t)iy * iw + 2 * j) = r;
    }
}

// ---- 3x3 / stride 2 / pad 1 max pooling of the ResNet stems ------------------------------------
// nn.MaxPool2d(3, 2, 1) between conv1 and layer1 of every ResNet trunk (networks/monodepth2.py:39,
// networks/posenet.py:21, 87) on the largest activation of the step ([96, 64, 96, 320] for the depth
// encoder's grouped call: 755 MB).  ATen's pair keeps an int64 index per output and its backward ran
// at 1.2 ms per launch (max_pool_backward_nchw: 2.4 ms per step, forward 1.0 ms).  Here the forward
// stores the window-local position of the maximum as ONE byte (0..8 = kh*3 + kw, relative to the
// window's unclamped origin (2*oy-1, 2*ox-1)) and the backward is a gather: an input pixel lies
// in at most 2 x 2 windows.  Selection rule as ATen writes it (MaxPoolForward: scan kh then kw,
// `val > max || isnan(val)`, start = first in-bounds element); the backward adds the (<= 4)
// matching windows in ATen's order (oy, then ox ascending), so sums are bit-identical.
__global__ void __launch_bounds__(NT) k_maxpool3s2_fwd(const float *__restrict__ x, float *__restrict__ out,
                                                       uint8_t *__restrict__ idx, int planes, int H, int W, int OH,
                                                       int OW)
{
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= OH * OW) return;
    const int oy = i / OW, ox = i - oy * OW;
    const int y0 = 2 * oy - 1, x0 = 2 * ox - 1;
    const int p0 = blockIdx.y * PLR;
    const size_t ni = (size_t)H * W, no = (size_t)OH * OW;
    float best[PLR];
    int bi[PLR];
    const int first = (y0 < 0 ? 3 : 0) + (x0 < 0 ? 1 : 0);
#pragma unroll
    for (int k = 0; k < PLR; ++k) { best[k] = -INFINITY; bi[k] = first; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int yy = y0 + kh;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int xx = x0 + kw;
            if (xx < 0 || xx >= W) continue;
            const float *xp = x + (size_t)p0 * ni + (size_t)yy * W + xx;
#pragma unroll
            for (int k = 0; k < PLR; ++k) {
                if (p0 + k >= planes) continue;
                const float v = xp[(size_t)k * ni];
                if ((v > best[k]) || (v != v)) { best[k] = v; bi[k] = kh * 3 + kw; }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PLR; ++k)
        if (p0 + k < planes) {
            out[(size_t)(p0 + k) * no + i] = best[k];
            idx[(size_t)(p0 + k) * no + i] = (uint8_t)bi[k];
        }
}

template <bool ADD>   // ADD: gx = addend + gathered gradient (the pooled tensor's other consumer, see mvf_maxpool3s2_bwd_add)
__global__ void __launch_bounds__(NT) k_maxpool3s2_bwd(const float *__restrict__ g, const uint8_t *__restrict__ idx,
                                                       const float *__restrict__ addend, float *__restrict__ gx,
                                                       int planes, int H, int W, int OH, int OW)
{
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, xq = i - y * W;
    const int p0 = blockIdx.y * PLR;
    const size_t ni = (size_t)H * W, no = (size_t)OH * OW;
    // windows covering row y: 2*oy-1 <= y <= 2*oy+1  <=>  oy in {y>>1, (y+1)>>1}: the second one
    // exists for odd y (and oy < OH).  All (<= 4) index bytes and gradients of a plane are loaded
    // unconditionally from clamped addresses, so the 8 loads x PLR planes of a lane are in flight
    // together (a loop with data-dependent bounds and a load behind each compare ran
    // latency-bound at 1 TB/s); invalid windows are masked in the select.
    const int oy0 = y >> 1, ox0 = xq >> 1;
    const bool vy = (y & 1) && (oy0 + 1 < OH), vx = (xq & 1) && (ox0 + 1 < OW);
    const int oy1 = vy ? oy0 + 1 : oy0, ox1 = vx ? ox0 + 1 : ox0;
    const int cy0 = (y - 2 * oy0 + 1) * 3, cy1 = (y - 2 * oy1 + 1) * 3;
    const int cx0 = xq - 2 * ox0 + 1, cx1 = xq - 2 * ox1 + 1;
    const size_t o00 = (size_t)oy0 * OW + ox0, o01 = (size_t)oy0 * OW + ox1, o10 = (size_t)oy1 * OW + ox0,
                 o11 = (size_t)oy1 * OW + ox1;
    uint8_t c[PLR][4];
    float v[PLR][4];
#pragma unroll
    for (int k = 0; k < PLR; ++k) {
        if (p0 + k >= planes) continue;
        const uint8_t *ip = idx + (size_t)(p0 + k) * no;
        const float *gp = g + (size_t)(p0 + k) * no;
        c[k][0] = ip[o00]; c[k][1] = ip[o01]; c[k][2] = ip[o10]; c[k][3] = ip[o11];
        v[k][0] = gp[o00]; v[k][1] = gp[o01]; v[k][2] = gp[o10]; v[k][3] = gp[o11];
    }
#pragma unroll
    for (int k = 0; k < PLR; ++k) {
        if (p0 + k >= planes) continue;
        float acc = 0.0f;          // ATen's order: (oy0,ox0), (oy0,ox1), (oy1,ox0), (oy1,ox1)
        if (c[k][0] == (uint8_t)(cy0 + cx0)) acc += v[k][0];
        if (vx && c[k][1] == (uint8_t)(cy0 + cx1)) acc += v[k][1];
        if (vy && c[k][2] == (uint8_t)(cy1 + cx0)) acc += v[k][2];
        if (vy && vx && c[k][3] == (uint8_t)(cy1 + cx1)) acc += v[k][3];
        gx[(size_t)(p0 + k) * ni + i] = ADD ? addend[(size_t)(p0 + k) * ni + i] + acc : acc;
    }
}

// ---- the same pool, wide form (W % 4 == 0, 16-byte aligned planes: every ResNet stem of the step) --------
// Round 4: the one-output-per-lane kernels above ran at 0.28 / 0.31 of the HBM peak -- nine (forward) / eight
// (backward) 4-byte loads per element and plane, each wave instruction touching 256 useful bytes.  Here a lane
// owns TWO horizontally adjacent outputs (forward: their 3 x 5 input patch is one 16-byte load + one scalar per
// row) resp. a 2 x 4 INPUT block (backward: the six windows that can touch it are a float2 + a float and a
// uchar2 + a uchar per window row; two 16-byte stores).  Same selection rule, same accumulation order as above
// (bit-identical results: tests/test_hip_parity.py::test_maxpool3s2_vs_oracle runs both forms).
constexpr int PLW = 4;      // planes per lane
MVF_DEV void pool_take(float v, int code, float &best, int &bi)
{
    if ((v > best) || (v != v)) { best = v; bi = code; }
}
__global__ void __launch_bounds__(NT) k_maxpool3s2_fwd_w4(const float *__restrict__ x, float *__restrict__ out,
                                                          uint8_t *__restrict__ idx, int planes, int H, int W,
                                                          int OH, int OW)
{
    const int OW2 = OW >> 1;                       // output pairs per row (OW = W / 2 is even)
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= OH * OW2) return;
    const int oy = i / OW2, j = i - oy * OW2;
    const int y0 = 2 * oy - 1, xa = 4 * j;         // window rows y0..y0+2; columns xa-1 .. xa+3
    const int p0 = blockIdx.y * PLW;
    const size_t ni = (size_t)H * W, no = (size_t)OH * OW;
    const bool left = xa > 0;
    float4 r[PLW][3];
    float l[PLW][3];
#pragma unroll
    for (int k = 0; k < PLW; ++k) {
        const float *xp = x + (size_t)min(p0 + k, planes - 1) * ni;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int yy = min(max(y0 + kh, 0), H - 1);          // clamped address; invalid rows are skipped below
            r[k][kh] = *reinterpret_cast<const float4 *>(xp + (size_t)yy * W + xa);
            l[k][kh] = xp[(size_t)yy * W + (left ? xa - 1 : xa)];
        }
    }
    const int first = (y0 < 0 ? 3 : 0);
#pragma unroll
    for (int k = 0; k < PLW; ++k) {
        if (p0 + k >= planes) continue;
        float b0 = -INFINITY, b1 = -INFINITY;
        int i0 = first + (left ? 0 : 1), i1 = first;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int yy = y0 + kh;
            if (yy < 0 || yy >= H) continue;
            if (left) pool_take(l[k][kh], kh * 3 + 0, b0, i0);
            pool_take(r[k][kh].x, kh * 3 + 1, b0, i0);
            pool_take(r[k][kh].y, kh * 3 + 2, b0, i0);
            pool_take(r[k][kh].y, kh * 3 + 0, b1, i1);
            pool_take(r[k][kh].z, kh * 3 + 1, b1, i1);
            pool_take(r[k][kh].w, kh * 3 + 2, b1, i1);
        }
        const size_t o = (size_t)(p0 + k) * no + (size_t)oy * OW + 2 * j;
        *reinterpret_cast<float2 *>(out + o) = make_float2(b0, b1);
        *reinterpret_cast<uchar2 *>(idx + o) = make_uchar2((uint8_t)i0, (uint8_t)i1);
    }
}

// one input pixel's gradient from the (<= 4) windows around it; wg / wc: gradients and codes of the window block
// rows {a, a+1} x columns {2b, 2b+1, 2b+2}; (ry, rx) = pixel position inside the 2 x 4 block
MVF_DEV float pool_gather(const float (&wg)[2][3], const uint8_t (&wc)[2][3], int ry, int rx, bool vy1, const bool (&vx)[3])
{
    // windows of row y = 2a + ry: oy0 = a (+ a+1 for odd y); of column x = 4b + rx: ox0 = 2b + (rx >> 1) (+1 for odd x)
    const int c0 = rx >> 1, c1 = c0 + 1;
    const bool oddy = ry & 1, oddx = rx & 1;
    const int cy0 = (ry + 1) * 3, cy1 = (ry - 1) * 3;          // (y - 2 oy + 1) * 3 for oy = a, a + 1
    const int cx0 = rx - 2 * c0 + 1, cx1 = rx - 2 * c1 + 1;
    float acc = 0.0f;
    if (wc[0][c0] == (uint8_t)(cy0 + cx0)) acc += wg[0][c0];
    if (oddx && vx[c1] && wc[0][c1] == (uint8_t)(cy0 + cx1)) acc += wg[0][c1];
    if (oddy && vy1 && wc[1][c0] == (uint8_t)(cy1 + cx0)) acc += wg[1][c0];
    if (oddy && vy1 && oddx && vx[c1] && wc[1][c1] == (uint8_t)(cy1 + cx1)) acc += wg[1][c1];
    return acc;
}
template <bool ADD>
__global__ void __launch_bounds__(NT) k_maxpool3s2_bwd_w4(const float *__restrict__ g, const uint8_t *__restrict__ idx,
                                                          const float *__restrict__ addend, float *__restrict__ gx,
                                                          int planes, int H, int W, int OH, int OW)
{
    const int W4 = W >> 2, H2 = H >> 1;            // H even, W % 4 == 0
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= H2 * W4) return;
    const int a = i / W4, b = i - a * W4;
    const int p0 = blockIdx.y * PLW;
    const size_t ni = (size_t)H * W, no = (size_t)OH * OW;
    const bool vy1 = a + 1 < OH;
    const bool vx[3] = {true, true, 2 * b + 2 < OW};
    const int oy1 = vy1 ? a + 1 : a, ox2 = vx[2] ? 2 * b + 2 : 2 * b + 1;
    float2 g01[PLW][2];
    float g2[PLW][2];
    uchar2 c01[PLW][2];
    uint8_t c2[PLW][2];
    float4 ad[PLW][2];
#pragma unroll
    for (int k = 0; k < PLW; ++k) {
        const size_t pb = (size_t)min(p0 + k, planes - 1) * no;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const size_t o = pb + (size_t)(r ? oy1 : a) * OW;
            g01[k][r] = *reinterpret_cast<const float2 *>(g + o + 2 * b);
            g2[k][r] = g[o + ox2];
            c01[k][r] = *reinterpret_cast<const uchar2 *>(idx + o + 2 * b);
            c2[k][r] = idx[o + ox2];
            if (ADD)
                ad[k][r] = *reinterpret_cast<const float4 *>(addend + (size_t)min(p0 + k, planes - 1) * ni +
                                                             (size_t)(2 * a + r) * W + 4 * b);
        }
    }
#pragma unroll
    for (int k = 0; k < PLW; ++k) {
        if (p0 + k >= planes) continue;
        float wg[2][3];
        uint8_t wc[2][3];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            wg[r][0] = g01[k][r].x; wg[r][1] = g01[k][r].y; wg[r][2] = g2[k][r];
            wc[r][0] = c01[k][r].x; wc[r][1] = c01[k][r].y; wc[r][2] = c2[k][r];
        }
        float *op = gx + (size_t)(p0 + k) * ni + (size_t)(2 * a) * W + 4 * b;
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
            float4 v;
            v.x = pool_gather(wg, wc, ry, 0, vy1, vx);
            v.y = pool_gather(wg, wc, ry, 1, vy1, vx);
            v.z = pool_gather(wg, wc, ry, 2, vy1, vx);
            v.w = pool_gather(wg, wc, ry, 3, vy1, vx);
            if (ADD) { v.x = ad[k][ry].x + v.x; v.y = ad[k][ry].y + v.y; v.z = ad[k][ry].z + v.z; v.w = ad[k][ry].w + v.w; }
            *reinterpret_cast<float4 *>(op + (size_t)ry * W) = v;
        }
    }
}

// ---- on-device colour augmentation -------------------------------------------------------------
// MonoDataset.__getitem__ / preprocess (datasets/mono_dataset.py:102-184, 214-256): with
// probability 1/2 a sample's frames are flipped horizontally, and with probability 1/2 all of its
// frames get ONE torchvision ColorJitter draw (brightness, contrast, saturation in [0.8,1.2], hue
// in [-0.1,0.1], the four adjustments in a random order).  The reference does this per item on the
// host with PIL; here the draw (factors, order, flags) is a few numbers per sample and the pixels
// are touched on the device: a reduction pass for the grey-level mean the contrast step blends
// with (it depends on the adjustments ordered before it), then one streaming pass.
// torchvision is on neither box: the adjustments restate its published float-tensor algorithms
// (_blend / rgb_to_grayscale / _rgb2hsv / _hsv2rgb) -- parity unpinned for this glue.
struct Rgb {
    float r, g, b;
};
MVF_DEV float clamp1(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
MVF_DEV float grey(const Rgb &c) { return 0.2989f * c.r + 0.587f * c.g + 0.114f * c.b; }
MVF_DEV Rgb blend(const Rgb &a, const Rgb &o, float ratio)
{
    const float q = 1.0f - ratio;
    return {clamp1(ratio * a.r + q * o.r), clamp1(ratio * a.g + q * o.g), clamp1(ratio * a.b + q * o.b)};
}
MVF_DEV Rgb adjust_hue(const Rgb &c, float f)
{
    const float maxc = fmaxf(c.r, fmaxf(c.g, c.b)), minc = fminf(c.r, fminf(c.g, c.b));
    const bool eq = maxc == minc;
    const float cr = maxc - minc;
    const float sat = cr / (eq ? 1.0f : maxc);
    const float dv = eq ? 1.0f : cr;
    const float rc = (maxc - c.r) / dv, gc = (maxc - c.g) / dv, bc = (maxc - c.b) / dv;
    float h = 0.0f;
    if (maxc == c.r) h = bc - gc;
    else if (maxc == c.g) h = 2.0f + rc - bc;
    else h = 4.0f + gc - rc;
    h = fmodf(h / 6.0f + 1.0f, 1.0f);
    h = h + f;
    h = h - floorf(h);                       // python-style % 1.0
    const float h6 = h * 6.0f, fi = floorf(h6), fr = h6 - fi;
    const int i = ((int)fi) % 6;
    const float v = maxc;
    const float p = clamp1(v * (1.0f - sat)), q = clamp1(v * (1.0f - sat * fr)), t = clamp1(v * (1.0f - sat * (1.0f - fr)));
    switch (i) {
    case 0: return {v, t, p};
    case 1: return {q, v, p};
    case 2: return {p, v, t};
    case 3: return {p, q, v};
    case 4: return {t, p, v};
    default: return {v, p, q};
    }
}
// adjustments order[0..n) of one sample; `mean` is consumed by the contrast step
MVF_DEV Rgb jitter_chain(Rgb c, const float *__restrict__ fac, const int *__restrict__ order, int n, float mean)
{
    for (int k = 0; k < n; ++k) {
        const int op = order[k];
        if (op == 0) c = blend(c, {0.0f, 0.0f, 0.0f}, fac[0]);
        else if (op == 1) c = blend(c, {mean, mean, mean}, fac[1]);
        else if (op == 2) { const float g = grey(c); c = blend(c, {g, g, g}, fac[2]); }
        else c = adjust_hue(c, fac[3]);
    }
    return c;
}

constexpr int JIT_NB = 64;     // reduction blocks per image

// partial sums of the grey level of (image after the adjustments ordered before contrast)
__global__ void __launch_bounds__(NT) k_jitter_mean(const float *__restrict__ img, const float *__restrict__ fac,
                                                    const int *__restrict__ order, const int *__restrict__ apply,
                                                    float *__restrict__ part, int N, int F)
{
    __shared__ float scratch[NT / kWave];
    const int n = blockIdx.y, s = n / F;           // image, sample
    float acc = 0.0f;
    if (apply[s]) {
        int pre = 0;
        while (pre < 4 && order[s * 4 + pre] != 1) ++pre;
        const float *p = img + (size_t)n * 3 * N;
        for (int i = blockIdx.x * NT + threadIdx.x; i < N; i += JIT_NB * NT) {
            Rgb c = {p[i], p[N + i], p[2 * (size_t)N + i]};
            acc += grey(jitter_chain(c, fac + s * 4, order + s * 4, pre, 0.0f));
        }
    }
    const float r = block_sum<NT>(acc, scratch);
    if (threadIdx.x == 0) part[(size_t)n * JIT_NB + blockIdx.x] = r;
}

// out_raw (nullable) = flipped copy; out_aug = flipped + jittered (copy when apply == 0)
__global__ void __launch_bounds__(NT) k_jitter_apply(const float *__restrict__ img, const float *__restrict__ fac,
                                                     const int *__restrict__ order, const int *__restrict__ apply,
                                                     const int *__restrict__ flip, const float *__restrict__ part,
                                                     float *__restrict__ out_raw, float *__restrict__ out_aug, int H,
                                                     int W, int F)
{
    __shared__ float smean;
    const int N = H * W, n = blockIdx.y, s = n / F;
    if (threadIdx.x == 0) {
        double m = 0.0;
        for (int k = 0; k < JIT_NB; ++k) m += (double)part[(size_t)n * JIT_NB + k];
        smean = (float)(m / (double)N);
    }
    __syncthreads();
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= N) return;
    const int y = i / W, x = i - y * W;
    const int sx = flip[s] ? W - 1 - x : x;
    const float *p = img + (size_t)n * 3 * N + (size_t)y * W + sx;
    Rgb c = {p[0], p[N], p[2 * (size_t)N]};
    const size_t o = (size_t)n * 3 * N + i;
    if (out_raw) { out_raw[o] = c.r; out_raw[o + N] = c.g; out_raw[o + 2 * (size_t)N] = c.b; }
    if (apply[s]) c = jitter_chain(c, fac + s * 4, order + s * 4, 4, smean);
    out_aug[o] = c.r; out_aug[o + N] = c.g; out_aug[o + 2 * (size_t)N] = c.b;
}


// ---- grouped batch norm: the per-layer vector glue (networks/grouped.py) -----------------------------------------
// A grouped layer normalises [B, G*C, ...] with its C-vectors repeated G times.  Around every MIOpen batch-norm call
// that was five tiny ATen launches forward (stack, repeat, two matrix-vector updates of the running statistics, the
// step counter) and two backward (stack, sum): 325 such layers in an HRNet18 step.  One launch each here.
// tiled[4][G*C] = (weight, bias, running_mean, running_var) repeated G times
__global__ void __launch_bounds__(NT) k_bn_tile(const float *__restrict__ w, const float *__restrict__ b,
                                                const float *__restrict__ rm, const float *__restrict__ rv,
                                                float *__restrict__ tiled, int C, int G)
{
    const int i = blockIdx.x * NT + threadIdx.x, n = G * C;
    if (i >= n) return;
    const int c = i % C;
    tiled[i] = w[c]; tiled[n + i] = b[c]; tiled[2 * n + i] = rm[c]; tiled[3 * n + i] = rv[c];
}
// The tiled statistics come back holding, per group, ONE momentum update from the common start.  The G sequential
// updates of the per-call form are r <- beta r + sum_g coef[g] upd[g][c] (GroupedBatchNorm2d._fold_running); both
// statistics and the step counter in one launch.
struct BnFoldCoef { float c[32]; };
__global__ void __launch_bounds__(NT) k_bn_fold_running(float *__restrict__ run_mean, float *__restrict__ run_var,
                                                        const float *__restrict__ upd_mean, const float *__restrict__ upd_var,
                                                        BnFoldCoef coef, float beta, int C, int G, int64_t *tracked)
{
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c == 0 && tracked) *tracked += G;
    if (c >= C) return;
    float am = 0.0f, av = 0.0f;
    for (int g = 0; g < G; ++g) {
        am = fmaf(coef.c[g], upd_mean[g * C + c], am);
        av = fmaf(coef.c[g], upd_var[g * C + c], av);
    }
    run_mean[c] = fmaf(beta, run_mean[c], am);
    run_var[c] = fmaf(beta, run_var[c], av);
}
// adjoint of the tiling: out[0][c] = sum_g gw[g*C + c], out[1][c] = sum_g gb[g*C + c] (group order)
__global__ void __launch_bounds__(NT) k_bn_untile(const float *__restrict__ gw, const float *__restrict__ gb,
                                                  float *__restrict__ out, int C, int G)
{
    const int c = blockIdx.x * NT + threadIdx.x;
    if (c >= C) return;
    float a = 0.0f, b = 0.0f;
    for (int g = 0; g < G; ++g) { a += gw[g * C + c]; b += gb[g * C + c]; }
    out[c] = a; out[C + c] = b;
}

// The same two steps for EVERY grouped batch-norm layer of a network in one launch each (round 4): a device table
// with one row per layer; the tiling runs when the grouped call begins, the fold when it ends (325 layers in an
// HRNet18 encoder: 650 launches per step less).
struct BnRow {
    const float *w, *b;
    float *rm, *rv, *tiled;
    int64_t *tracked;
    int64_t C, pad;
};
__global__ void __launch_bounds__(NT) k_bn_tile_many(const BnRow *__restrict__ rows, int G)
{
    const BnRow r = rows[blockIdx.y];
    const int C = (int)r.C, n = G * C;
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
        const int c = i % C;
        r.tiled[i] = r.w[c]; r.tiled[n + i] = r.b[c]; r.tiled[2 * n + i] = r.rm[c]; r.tiled[3 * n + i] = r.rv[c];
    }
}
__global__ void __launch_bounds__(NT) k_bn_fold_many(const BnRow *__restrict__ rows, BnFoldCoef coef, float beta, int G)
{
    const BnRow r = rows[blockIdx.y];
    const int C = (int)r.C, n = G * C;
    if (blockIdx.x == 0 && threadIdx.x == 0 && r.tracked) *r.tracked += G;
    for (int c = blockIdx.x * NT + threadIdx.x; c < C; c += gridDim.x * NT) {
        float am = 0.0f, av = 0.0f;
        for (int g = 0; g < G; ++g) {
            am = fmaf(coef.c[g], r.tiled[2 * n + g * C + c], am);
            av = fmaf(coef.c[g], r.tiled[3 * n + g * C + c], av);
        }
        r.rm[c] = fmaf(beta, r.rm[c], am);
        r.rv[c] = fmaf(beta, r.rv[c], av);
    }
}

// ---- regrouping of interleaved group batches (networks/grouped.py) ----------------------------------------------
// One optimisation step of the reference calls the depth encoder once per input and hands each call's feature
// pyramid to the decoder / fusion module calls that need it (train.py:745-747, 788-797, 830-868).  Here the G
// encoder inputs are ONE interleaved batch (sample n = b * G + g) and every consumer takes its own interleaved
// batch of some of the groups: dst_k sample b * Gk + j = src sample b * G + group_k[j].  Forward: every (output,
// position) slot copies one group, ONE launch for all outputs of a pyramid level (ATen's stack per consumer ran at
// 0.25 of the HBM peak: strided CatArrayBatchedCopy).  Backward: the gradient of a source group is the sum over the
// slots that read it, in (output, position) order, written once -- no zero fill for unused groups, no accumulation
// passes of the autograd engine, no stack of the group gradients.
constexpr int RG_MAX_OUT = 8, RG_MAX_SLOTS = 32;
struct RegroupPlan {
    float *out[RG_MAX_OUT];                 // forward: outputs; backward: their gradients (nullable)
    int64_t ostride[RG_MAX_OUT];            // distance between consecutive samples of an output, in kernel elements
    int gk[RG_MAX_OUT];                     // groups per output
    unsigned char slot_k[RG_MAX_SLOTS], slot_j[RG_MAX_SLOTS], slot_g[RG_MAX_SLOTS];
    unsigned char first[RG_MAX_SLOTS + 1];  // backward: slots [first[g], first[g+1]) of `by_g` read source group g
    unsigned char by_g[RG_MAX_SLOTS];
    int nslots;
};

// grid (chunk blocks, slots, B): one shot per block, four 16-byte (VEC) accesses per lane in flight
template <bool VEC>
__global__ void __launch_bounds__(NT) k_regroup_fwd(const float *__restrict__ src, RegroupPlan pl, int G, int64_t per)
{
    constexpr int U = 4;
    const int s = blockIdx.y, b = blockIdx.z;
    const int k = pl.slot_k[s], j = pl.slot_j[s], g = pl.slot_g[s];
    const int64_t q0 = (int64_t)blockIdx.x * (NT * U) + threadIdx.x;
    const int64_t so = ((int64_t)b * G + g) * per, dof = ((int64_t)b * pl.gk[k] + j) * per;
    float *__restrict__ dst = pl.out[k];
    if (VEC) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q0 + u * NT < per) v[u] = reinterpret_cast<const float4 *>(src)[so + q0 + u * NT];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q0 + u * NT < per) reinterpret_cast<float4 *>(dst)[dof + q0 + u * NT] = v[u];
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q0 + u * NT < per) dst[dof + q0 + u * NT] = src[so + q0 + u * NT];
    }
}

// grid (chunk blocks, G, B): the gradient of source group g = sum of its slots' gradients in slot order
template <bool VEC>
__global__ void __launch_bounds__(NT) k_regroup_bwd(float *__restrict__ g_src, RegroupPlan pl, int G, int64_t per)
{
    constexpr int U = 4;
    const int g = blockIdx.y, b = blockIdx.z;
    const int64_t q0 = (int64_t)blockIdx.x * (NT * U) + threadIdx.x;
    const int64_t so = ((int64_t)b * G + g) * per;
    const int lo = pl.first[g], hi = pl.first[g + 1];
    if (VEC) {
        float4 acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        for (int e = lo; e < hi; ++e) {
            const int s = pl.by_g[e], k = pl.slot_k[s];
            const float4 *__restrict__ gp = reinterpret_cast<const float4 *>(pl.out[k]) + ((int64_t)b * pl.gk[k] + pl.slot_j[s]) * pl.ostride[k];
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (q0 + u * NT < per) v[u] = gp[q0 + u * NT];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (q0 + u * NT < per) {
                    if (e == lo) acc[u] = v[u];
                    else { acc[u].x += v[u].x; acc[u].y += v[u].y; acc[u].z += v[u].z; acc[u].w += v[u].w; }
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q0 + u * NT < per) reinterpret_cast<float4 *>(g_src)[so + q0 + u * NT] = acc[u];
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t q = q0 + u * NT;
            if (q >= per) continue;
            float acc = 0.0f;
            for (int e = lo; e < hi; ++e) {
                const int s = pl.by_g[e], k = pl.slot_k[s];
                const float v = pl.out[k][((int64_t)b * pl.gk[k] + pl.slot_j[s]) * pl.ostride[k] + q];
                acc = (e == lo) ? v : acc + v;
            }
            g_src[so + q] = acc;
        }
    }
}

// ---- G x P separate [B, len] tensors -> ONE interleaved group batch (input side of a grouped call) -------------
// dst sample b * G + g = the parts of group g back to back (e.g. the two frames of a pose pair, train.py:943-946,
// for the six pairs of a step: the reference concatenates per call; here cat per pair + stack of the pairs were
// seven ATen launches moving every image twice).  Forward only: the inputs are images.
constexpr int IL_MAX_SLOTS = 32;
struct InterleavePlan {
    const float *src[IL_MAX_SLOTS];
    int64_t len[IL_MAX_SLOTS], off[IL_MAX_SLOTS];     // part length and offset inside the destination sample
    unsigned char group[IL_MAX_SLOTS];
};
template <bool VEC>
__global__ void __launch_bounds__(NT) k_interleave_fwd(InterleavePlan pl, float *__restrict__ dst, int G, int64_t total)
{
    constexpr int U = 4;
    const int s = blockIdx.y, b = blockIdx.z;
    const int64_t len = pl.len[s], q0 = (int64_t)blockIdx.x * (NT * U) + threadIdx.x;
    if ((int64_t)blockIdx.x * (NT * U) >= len) return;
    const float *__restrict__ src = pl.src[s];
    const int64_t so = (int64_t)b * len, dof = ((int64_t)b * G + pl.group[s]) * total + pl.off[s];
    if (VEC) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q0 + u * NT < len) v[u] = reinterpret_cast<const float4 *>(src)[so + q0 + u * NT];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q0 + u * NT < len) reinterpret_cast<float4 *>(dst)[dof + q0 + u * NT] = v[u];
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (q0 + u * NT < len) dst[dof + q0 + u * NT] = src[so + q0 + u * NT];
    }
}

// host side of both directions: the plan from (counts, groups); slots whose output pointer is null are left out
// (backward: an output nobody differentiated)
static int regroup_plan(RegroupPlan &pl, float *const *out, int n_out, const int32_t *counts, const int32_t *groups,
                        int G, bool skip_null, bool &vec_ok, int64_t chunk, const int64_t *strides = nullptr)
{
    if (n_out <= 0 || n_out > RG_MAX_OUT || G <= 0 || G > RG_MAX_SLOTS || !counts || !groups || !out)
        return (int)hipErrorInvalidValue;
    int ns = 0, at = 0;
    for (int k = 0; k < n_out; ++k) {
        if (counts[k] <= 0) return (int)hipErrorInvalidValue;
        pl.out[k] = out[k];
        pl.gk[k] = counts[k];
        pl.ostride[k] = strides ? strides[k] : chunk;
        if (!out[k] && !skip_null) return (int)hipErrorInvalidValue;
        if (out[k] && pl.ostride[k] < chunk) return (int)hipErrorInvalidValue;
        if (out[k] && ((((uintptr_t)out[k]) & 15) || (pl.ostride[k] & 3))) vec_ok = false;
        for (int j = 0; j < counts[k]; ++j, ++at) {
            const int g = groups[at];
            if (g < 0 || g >= G) return (int)hipErrorInvalidValue;
            if (!out[k]) continue;
            if (ns >= RG_MAX_SLOTS) return (int)hipErrorInvalidValue;
            pl.slot_k[ns] = (unsigned char)k; pl.slot_j[ns] = (unsigned char)j; pl.slot_g[ns] = (unsigned char)g;
            ++ns;
        }
    }
    for (int k = n_out; k < RG_MAX_OUT; ++k) { pl.out[k] = nullptr; pl.gk[k] = 0; pl.ostride[k] = 0; }
    if (vec_ok)
        for (int k = 0; k < n_out; ++k) pl.ostride[k] /= 4;
    pl.nslots = ns;
    int e = 0;
    for (int g = 0; g < G; ++g) {               // stable: slot order within a group = (output, position) order
        pl.first[g] = (unsigned char)e;
        for (int s = 0; s < ns; ++s)
            if (pl.slot_g[s] == g) pl.by_g[e++] = (unsigned char)s;
    }
    for (int g = G; g <= RG_MAX_SLOTS; ++g) pl.first[g] = (unsigned char)e;
    return 0;
}
}  // namespace

// wide adjoint of ReflectionPad2d(1) for mvf_reflect_pad1_bwd (mvf_geom.hip); false when the shape needs the narrow form
namespace mvf_glue {
bool reflect_pad1_fwd_wide(const float *in, float *out, int planes, int H, int W, hipStream_t st)
{
    if (!pad_wide_ok(out, in, H, W) || planes > 65535 * PADW_PL) return false;
    hipLaunchKernelGGL(k_pad1_fwd_w4, dim3((unsigned)((H * (W / 4) + NT - 1) / NT), (unsigned)((planes + PADW_PL - 1) / PADW_PL)),
                       dim3(NT), 0, st, in, out, planes, H, W);
    return true;
}
bool reflect_pad1_bwd_wide(const float *g_out, float *g_in, int planes, int H, int W, hipStream_t st)
{
    if (!pad_wide_ok(g_out, g_in, H, W) || planes > 65535 * PADW_PL) return false;
    hipLaunchKernelGGL(k_pad1_bwd_w4, dim3((unsigned)((H * (W / 4) + NT - 1) / NT), (unsigned)((planes + PADW_PL - 1) / PADW_PL)),
                       dim3(NT), 0, st, g_out, g_in, planes, 1, 1, 0, H, W);
    return true;
}
}  // namespace mvf_glue

extern "C" {

int mvf_up2cat_pad_fwd(const float *x, const float *skip, float *out, int B, int C1, int C2, int h, int w,
                       void *stream)
{
    if (B <= 0 || C1 <= 0 || h <= 0 || w <= 0) return 0;
    if (!x || !out || C2 < 0 || (C2 > 0 && !skip)) return (int)hipErrorInvalidValue;
    const int n = (2 * h + 2) * (2 * w + 2), C = C1 + C2;
    // algorithmic bytes: x and skip read once, the padded concatenation written once
    ProfScope ps(MVF_PROF_UP2CAT_FWD, stream, 4LL * B * ((int64_t)C1 * h * w + (int64_t)C2 * 4 * h * w + (int64_t)C * n));
    if (pad_wide_ok(out, C2 > 0 ? skip : x, 2 * h, 2 * w) && (((uintptr_t)x) & 7) == 0 && (C + PADW_PL - 1) / PADW_PL <= 65535 &&
        B <= 65535) {
        hipLaunchKernelGGL(k_up2cat_pad_fwd_w4, dim3((unsigned)((2 * h * (2 * w / 4) + NT - 1) / NT), (unsigned)((C + PADW_PL - 1) / PADW_PL), (unsigned)B),
                           dim3(NT), 0, (hipStream_t)stream, x, skip, out, C1, C2, h, w);
        return hip_check_launch();
    }
    hipLaunchKernelGGL(k_up2cat_pad_fwd, dim3((unsigned)((n + NT - 1) / NT), (unsigned)((C + PL - 1) / PL), (unsigned)B),
                       dim3(NT), 0, (hipStream_t)stream, x, skip, out, C1, C2, h, w);
    return hip_check_launch();
}

int mvf_up2cat_pad_bwd(const float *g_out, float *g_x, float *g_skip, int B, int C1, int C2, int h, int w,
                       void *stream)
{
    if (B <= 0 || C1 <= 0 || h <= 0 || w <= 0) return 0;
    if (!g_out || C2 < 0 || h < 2 || w < 2) return (int)hipErrorInvalidValue;   // 2h, 2w >= 4
    const int64_t PPb = (int64_t)(2 * h + 2) * (2 * w + 2);
    if (g_x) {
        ProfScope ps(MVF_PROF_UP2CAT_BWD_X, stream, 4LL * B * C1 * (PPb + (int64_t)h * w));
        if (pad_wide_ok(g_out, g_x, 2 * h, 2 * w) && (((uintptr_t)g_x) & 7) == 0 && (int64_t)B * C1 <= 65535LL * PADW_PL)
            hipLaunchKernelGGL(k_up2cat_pad_bwd_x_w4, dim3((unsigned)((h * (w / 2) + NT - 1) / NT), (unsigned)((B * C1 + PADW_PL - 1) / PADW_PL)),
                               dim3(NT), 0, (hipStream_t)stream, g_out, g_x, B * C1, C1, C1 + C2, h, w);
        else
        hipLaunchKernelGGL(k_up2cat_pad_bwd_x, dim3((unsigned)((h * w + NT - 1) / NT), (unsigned)((C1 + PL - 1) / PL), (unsigned)B),
                           dim3(NT), 0, (hipStream_t)stream, g_out, g_x, C1, C2, h, w);
    }
    if (g_skip && C2 > 0) {
        ProfScope ps(MVF_PROF_UP2CAT_BWD_SKIP, stream, 4LL * B * C2 * (PPb + 4LL * h * w));
        if (pad_wide_ok(g_out, g_skip, 2 * h, 2 * w) && (int64_t)B * C2 <= 65535LL * PADW_PL)
            hipLaunchKernelGGL(k_pad1_bwd_w4, dim3((unsigned)((2 * h * (2 * w / 4) + NT - 1) / NT), (unsigned)((B * C2 + PADW_PL - 1) / PADW_PL)),
                               dim3(NT), 0, (hipStream_t)stream, g_out, g_skip, B * C2, C2, C1 + C2, C1, 2 * h, 2 * w);
        else
        hipLaunchKernelGGL(k_up2cat_pad_bwd_skip, dim3((unsigned)((4 * h * w + NT - 1) / NT), (unsigned)((C2 + PL - 1) / PL), (unsigned)B),
                           dim3(NT), 0, (hipStream_t)stream, g_out, g_skip, C1, C2, h, w);
    }
    return hip_check_launch();
}

// ---- pad kernels with the producer's bias + ELU applied on load (see k_pad1_act_fwd_w4) ------------------------
int mvf_pad_act_supported(int H, int W) { return ((W & 3) == 0 && W >= 8 && H >= 4 && !getenv("MVF_PAD_NARROW")) ? 1 : 0; }

size_t mvf_pad_act_workspace_floats(int B, int C, int H, int W)
{
    const int chunks = (H * (W / 4) + NT * PADA_U - 1) / (NT * PADA_U);
    return (size_t)B * (chunks > 0 ? chunks : 1) * C + 16;
}

int mvf_reflect_pad1_act_fwd(const float *in, const float *bias, float *out, int B, int C, int H, int W, void *stream)
{
    if (B <= 0 || C <= 0) return 0;
    const int64_t planes = (int64_t)B * C;
    if (!in || !bias || !out || !pad_wide_ok(out, in, H, W) || planes > 65535LL * PADW_PL) return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_REFLECT_PAD_FWD, stream, 4LL * planes * ((int64_t)H * W + (int64_t)(H + 2) * (W + 2)));
    hipLaunchKernelGGL(k_pad1_act_fwd_w4, dim3((unsigned)((H * (W / 4) + NT - 1) / NT), (unsigned)((planes + PADW_PL - 1) / PADW_PL)),
                       dim3(NT), 0, (hipStream_t)stream, in, bias, out, (int)planes, C, H, W);
    return hip_check_launch();
}

int mvf_reflect_pad1_act_bwd(const float *g_padded, const float *padded, float *g_in, float *g_bias, float *workspace,
                             int B, int C, int H, int W, void *stream)
{
    if (B <= 0 || C <= 0) return 0;
    const int64_t planes = (int64_t)B * C;
    if (!g_padded || !padded || !g_in || !g_bias || !workspace || !pad_wide_ok(g_padded, g_in, H, W) ||
        (((uintptr_t)padded) & 3) || planes > 65535)
        return (int)hipErrorInvalidValue;
    const int chunks = (H * (W / 4) + NT * PADA_U - 1) / (NT * PADA_U);
    // padded gradient and padded activations read once, the input gradient written once
    ProfScope ps(MVF_PROF_REFLECT_PAD_BWD, stream, 4LL * planes * (2LL * (H + 2) * (W + 2) + (int64_t)H * W));
    hipLaunchKernelGGL(k_pad1_act_bwd_w4, dim3((unsigned)chunks, (unsigned)planes), dim3(NT), 0, (hipStream_t)stream, g_padded,
                       padded, g_in, workspace, C, H, W);
    hipLaunchKernelGGL(k_bias_grad_finish_tree, dim3((unsigned)C), dim3(NT), 0, (hipStream_t)stream, workspace, g_bias, C,
                       B * chunks);
    return hip_check_launch();
}

int mvf_up2cat_pad_act_fwd(const float *x, const float *bias, const float *skip, float *out, int B, int C1, int C2, int h,
                           int w, void *stream)
{
    if (B <= 0 || C1 <= 0 || h <= 0 || w <= 0) return 0;
    const int C = C1 + C2, n = (2 * h + 2) * (2 * w + 2);
    if (!x || !bias || !out || C2 < 0 || (C2 > 0 && !skip) || !pad_wide_ok(out, C2 > 0 ? skip : x, 2 * h, 2 * w) ||
        (((uintptr_t)x) & 7) || (C + PADW_PL - 1) / PADW_PL > 65535 || B > 65535)
        return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_UP2CAT_FWD, stream, 4LL * B * ((int64_t)C1 * h * w + (int64_t)C2 * 4 * h * w + (int64_t)C * n));
    hipLaunchKernelGGL(k_up2cat_pad_act_fwd_w4, dim3((unsigned)((2 * h * (2 * w / 4) + NT - 1) / NT), (unsigned)((C + PADW_PL - 1) / PADW_PL), (unsigned)B),
                       dim3(NT), 0, (hipStream_t)stream, x, bias, skip, out, C1, C2, h, w);
    return hip_check_launch();
}

size_t mvf_up2cat_pad_act_workspace_floats(int B, int C1, int h, int w)
{
    const int chunks = (h * (w / 2) + NT * 2 - 1) / (NT * 2);
    return (size_t)B * (chunks > 0 ? chunks : 1) * C1 + 16;
}

int mvf_up2cat_pad_act_bwd(const float *g_padded, const float *padded, float *g_x, float *g_bias, float *g_skip,
                           float *workspace, int B, int C1, int C2, int h, int w, void *stream)
{
    if (B <= 0 || C1 <= 0 || h <= 0 || w <= 0) return 0;
    if (!g_padded || !padded || !g_x || !g_bias || !workspace || C2 < 0 || h < 2 || w < 2 ||
        !pad_wide_ok(g_padded, g_x, 2 * h, 2 * w) || (((uintptr_t)g_x) & 7) || (((uintptr_t)padded) & 3) ||
        (int64_t)B * C1 > 65535)
        return (int)hipErrorInvalidValue;
    const int64_t PPb = (int64_t)(2 * h + 2) * (2 * w + 2);
    {
        const int chunks = (h * (w / 2) + NT * 2 - 1) / (NT * 2);
        ProfScope ps(MVF_PROF_UP2CAT_BWD_X, stream, 4LL * B * C1 * (PPb + (int64_t)h * w + (int64_t)h * w));
        hipLaunchKernelGGL(k_up2cat_pad_act_bwd_x_w4, dim3((unsigned)chunks, (unsigned)(B * C1)), dim3(NT), 0, (hipStream_t)stream,
                           g_padded, padded, g_x, workspace, C1, C1 + C2, h, w);
        hipLaunchKernelGGL(k_bias_grad_finish_tree, dim3((unsigned)C1), dim3(NT), 0, (hipStream_t)stream, workspace, g_bias,
                           C1, B * chunks);
    }
    if (g_skip && C2 > 0) {
        ProfScope ps(MVF_PROF_UP2CAT_BWD_SKIP, stream, 4LL * B * C2 * (PPb + 4LL * h * w));
        if (pad_wide_ok(g_padded, g_skip, 2 * h, 2 * w) && (int64_t)B * C2 <= 65535LL * PADW_PL)
            hipLaunchKernelGGL(k_pad1_bwd_w4, dim3((unsigned)((2 * h * (2 * w / 4) + NT - 1) / NT), (unsigned)((B * C2 + PADW_PL - 1) / PADW_PL)),
                               dim3(NT), 0, (hipStream_t)stream, g_padded, g_skip, B * C2, C2, C1 + C2, C1, 2 * h, 2 * w);
        else
            hipLaunchKernelGGL(k_up2cat_pad_bwd_skip, dim3((unsigned)((4 * h * w + NT - 1) / NT), (unsigned)((C2 + PL - 1) / PL), (unsigned)B),
                               dim3(NT), 0, (hipStream_t)stream, g_padded, g_skip, C1, C2, h, w);
    }
    return hip_check_launch();
}

int mvf_disp_head_fwd(const float *logit, float *disp, float *depth, float *mean_partials, int B, int N,
                      float min_disp, float range, void *stream)
{
    if (B <= 0 || N <= 0) return 0;
    if (!logit || !disp || B > 65535) return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_DISP_HEAD_FWD, stream, 4LL * B * N * (2 + (depth ? 1 : 0)));
    hipLaunchKernelGGL(k_disp_head_fwd, dim3(NMEAN, (unsigned)B), dim3(NT), 0, (hipStream_t)stream, logit, disp,
                       depth, mean_partials, N, min_disp, range);
    return hip_check_launch();
}

int mvf_disp_head_bwd(const float *disp, const float *g_disp, const float *g_depth, float *g_logit, int64_t n,
                      float min_disp, float range, void *stream)
{
    if (n <= 0) return 0;
    if (!disp || !g_logit) return (int)hipErrorInvalidValue;
    int64_t blocks = (n + NT - 1) / NT;
    if (blocks > 8192) blocks = 8192;
    ProfScope ps(MVF_PROF_DISP_HEAD_BWD, stream, 4LL * n * (2 + (g_disp ? 1 : 0) + (g_depth ? 1 : 0)));
    hipLaunchKernelGGL(k_disp_head_bwd, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, disp, g_disp,
                       g_depth, g_logit, n, min_disp, range);
    return hip_check_launch();
}

int mvf_disp_head_bwd_units(const float *disp, const float *g_disp, const float *g_depth, float *g_logit, int B,
                            int N, float min_disp, float range, const mvf_head_unit_grad *units, int n_units,
                            void *stream)
{
    if (B <= 0 || N <= 0) return 0;
    if (!disp || !g_logit || B > 65535 || n_units < 0 || n_units > MVF_MAX_UNITS || (n_units && !units))
        return (int)hipErrorInvalidValue;
    HeadUnits hu = {};
    hu.n = n_units;
    bool vec = (N % 4 == 0) && ((((uintptr_t)disp | (uintptr_t)g_disp | (uintptr_t)g_depth | (uintptr_t)g_logit) & 15) == 0);
    int64_t imgs = 0;
    for (int i = 0; i < n_units; ++i) {
        const mvf_head_unit_grad &d = units[i];
        if (!d.g_disp_raw || !d.stats || (!d.g_loss && !d.g_sum) || d.first < 0 || d.step < 1 || d.count < 1 ||
            d.first + (int64_t)(d.count - 1) * d.step >= B)
            return (int)hipErrorInvalidValue;
        HeadUnit &u = hu.u[i];
        u.g_raw = d.g_disp_raw; u.stats = d.stats; u.g_loss = d.g_loss; u.g_sum = d.g_sum;
        u.raw_stride = d.raw_stride ? (size_t)d.raw_stride : (size_t)N;
        u.smoothness = d.smoothness; u.first = d.first; u.step = d.step; u.count = d.count;
        vec = vec && ((((uintptr_t)d.g_disp_raw) & 15) == 0) && (u.raw_stride % 4 == 0);
        imgs += d.count;
    }
    const int per = vec ? N / 4 : N;
    int64_t gx = (per + NT - 1) / NT;
    if (!vec && gx > 64) gx = 64;
    // disp read, g_logit written, the optional planes, one raw gradient per covered image
    ProfScope ps(MVF_PROF_DISP_HEAD_BWD, stream, 4LL * N * ((int64_t)B * (2 + (g_disp ? 1 : 0) + (g_depth ? 1 : 0)) + imgs));
    hipLaunchKernelGGL(k_disp_head_bwd_units, dim3((unsigned)gx, (unsigned)B), dim3(NT), 0, (hipStream_t)stream, disp,
                       g_disp, g_depth, g_logit, N, vec ? 1 : 0, min_disp, range, hu);
    return hip_check_launch();
}

static int bias_act_nsplit(int N, int C, int HW)
{
    // >= 2048 blocks when the tensor allows it, runs of at least 4096 elements
    int64_t total = (int64_t)N * HW;
    int64_t s = (2048 + C - 1) / C;
    const int64_t smax = (total + 4095) / 4096;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    return (int)s;
}

// one-shot form of the backward pass: 16-byte vectors, planes of >= 512 float4, plane count within grid.y
static bool bias_shot_ok(int N, int C, int HW, bool vec)
{
    return vec && (HW >> 2) >= 512 && (int64_t)N * C <= 65535 && !getenv("MVF_BIAS_RUNS");
}
static int bias_shot_chunks(int HW) { return ((HW >> 2) + NT * 4 - 1) / (NT * 4); }

size_t mvf_bias_act_workspace_floats(int N, int C, int HW)
{
    const size_t runs = (size_t)bias_act_nsplit(N, C, HW) * C;
    const size_t shot = ((HW & 3) == 0) ? (size_t)N * bias_shot_chunks(HW) * C : 0;
    return runs > shot ? runs : shot;
}

int mvf_bias_act_fwd(const float *x, const float *bias, const float *slope, const float *res, float *out, int N,
                     int C, int HW, int act, int slope_n, void *stream)
{
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    if (!x || !out || act < 0 || act > 3 || (act == ACT_PRELU && (!slope || (slope_n != 1 && slope_n != C))))
        return (int)hipErrorInvalidValue;
    const bool vec = (HW & 3) == 0 && ((((uintptr_t)x | (uintptr_t)out | (uintptr_t)res) & 15) == 0);
    const int per_plane = vec ? HW / 4 : HW;
    const int64_t total = (int64_t)N * C * per_plane;
    const int64_t blocks = (total + NT * 4 - 1) / (NT * 4);
    if (total >= 0x7fffffffLL) return (int)hipErrorInvalidValue;       // 32-bit element index inside the kernel
    ProfScope ps(MVF_PROF_BIAS_ACT_FWD, stream, 4LL * N * C * HW * (2 + (res ? 1 : 0)));
    if (vec)
        hipLaunchKernelGGL(k_bias_act_fwd<true>, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, x, bias,
                           slope, res, out, C, per_plane, total, act, slope_n);
    else
        hipLaunchKernelGGL(k_bias_act_fwd<false>, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, x, bias,
                           slope, res, out, C, per_plane, total, act, slope_n);
    return hip_check_launch();
}

int mvf_bias_act_bwd(const float *g, const float *out, float *g_x, float *g_bias, float *workspace, int N, int C,
                     int HW, int act, void *stream)
{
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    if (!g || act < 0 || act > 2 || (act != ACT_NONE && (!out || !g_x)) || (g_bias && !workspace))
        return (int)hipErrorInvalidValue;
    const bool vec = (HW & 3) == 0 && ((((uintptr_t)g | (uintptr_t)out | (uintptr_t)g_x) & 15) == 0);
    if (!g_bias) {                       // no bias gradient wanted: the activation's adjoint alone
        if (act == ACT_NONE) return 0;
        const int64_t total = (int64_t)N * C * (vec ? HW / 4 : HW);
        const int64_t blocks = (total + NT * 4 - 1) / (NT * 4);
        if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
        ProfScope ps(MVF_PROF_ACT_BWD_FLAT, stream, 12LL * N * C * HW);
        if (vec)
            hipLaunchKernelGGL(k_act_bwd_flat<true>, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, g, out,
                               g_x, total, act);
        else
            hipLaunchKernelGGL(k_act_bwd_flat<false>, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, g, out,
                               g_x, total, act);
        return hip_check_launch();
    }
    const int nsplit = bias_act_nsplit(N, C, HW);
    if ((int64_t)N * HW >= 0x7fffffffLL) return (int)hipErrorInvalidValue;      // 32-bit element index per channel
    // g read once; with an activation: its output read and g_x written as well
    ProfScope ps(MVF_PROF_BIAS_ACT_BWD, stream, 4LL * N * C * HW * (act != ACT_NONE ? 3 : 1));
    if (bias_shot_ok(N, C, HW, vec)) {
        const int chunks = bias_shot_chunks(HW);
        hipLaunchKernelGGL(k_bias_act_bwd_shot, dim3((unsigned)chunks, (unsigned)(N * C)), dim3(NT), 0, (hipStream_t)stream,
                           reinterpret_cast<const float4 *>(g), reinterpret_cast<const float4 *>(out),
                           reinterpret_cast<float4 *>(g_x), workspace, C, HW >> 2, act);
        hipLaunchKernelGGL(k_bias_grad_finish_tree, dim3((unsigned)C), dim3(NT), 0, (hipStream_t)stream, workspace, g_bias,
                           C, N * chunks);
        return hip_check_launch();
    }
    if (vec)
        hipLaunchKernelGGL(k_bias_act_bwd<true>, dim3((unsigned)(C * nsplit)), dim3(NT), 0, (hipStream_t)stream, g,
                           out, g_x, workspace, N, C, HW, act, nsplit);
    else
        hipLaunchKernelGGL(k_bias_act_bwd<false>, dim3((unsigned)(C * nsplit)), dim3(NT), 0, (hipStream_t)stream, g,
                           out, g_x, workspace, N, C, HW, act, nsplit);
    hipLaunchKernelGGL(k_bias_grad_finish, dim3((unsigned)((C + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream,
                       workspace, g_bias, C, nsplit);
    return hip_check_launch();
}

int mvf_resize_bilinear_fwd(const float *x, float *out, int planes, int ih, int iw, int oh, int ow, float scale_h,
                            float scale_w, int align_corners, void *stream)
{
    if (planes <= 0 || oh <= 0 || ow <= 0) return 0;
    if (!x || !out || ih <= 0 || iw <= 0 || (planes + PLR - 1) / PLR > 65535) return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_RESIZE_FWD, stream, 4LL * planes * ((int64_t)ih * iw + (int64_t)oh * ow));
    hipLaunchKernelGGL(k_resize_bilinear_fwd, dim3((unsigned)((oh * ow + NT - 1) / NT), (unsigned)((planes + PLR - 1) / PLR)),
                       dim3(NT), 0, (hipStream_t)stream, x, out, planes, ih, iw, oh, ow, scale_h, scale_w,
                       align_corners);
    return hip_check_launch();
}

size_t mvf_resize_bilinear_bwd_workspace_floats(int planes, int ih, int ow)
{
    return (planes > 0 && ih > 0 && ow > 0) ? (size_t)planes * ih * ow : 0;
}

int mvf_resize_bilinear_bwd(const float *g_out, float *g_x, float *workspace, int planes, int ih, int iw, int oh, int ow,
                            float scale_h, float scale_w, int align_corners, void *stream)
{
    if (planes <= 0 || ih <= 0 || iw <= 0) return 0;
    if (!g_out || !g_x || oh <= 0 || ow <= 0 || (planes + PLR - 1) / PLR > 65535) return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_RESIZE_BWD, stream, 4LL * planes * ((int64_t)ih * iw + (int64_t)oh * ow));
    hipStream_t st = (hipStream_t)stream;
    if (workspace && (planes + 1) / 2 <= 65535 && !getenv("MVF_RESIZE_ONEPASS")) {
        // rows, then columns (workspace: mvf_resize_bilinear_bwd_workspace_floats(planes, ih, ow) floats)
        const bool vec = (ow & 3) == 0 && (((uintptr_t)g_out) & 15) == 0 && (((uintptr_t)workspace) & 15) == 0;
        const int per = vec ? ow >> 2 : ow;
        const dim3 ga((unsigned)((ih * per + NT - 1) / NT), (unsigned)((planes + 1) / 2));
        if (vec) hipLaunchKernelGGL(k_resize_bwd_rows<true>, ga, dim3(NT), 0, st, g_out, workspace, planes, ih, oh, ow, scale_h, align_corners);
        else hipLaunchKernelGGL(k_resize_bwd_rows<false>, ga, dim3(NT), 0, st, g_out, workspace, planes, ih, oh, ow, scale_h, align_corners);
        hipLaunchKernelGGL(k_resize_bwd_cols, dim3((unsigned)((ih * iw + NT - 1) / NT), (unsigned)((planes + PLR - 1) / PLR)),
                           dim3(NT), 0, st, workspace, g_x, planes, ih, iw, ow, scale_w, align_corners);
        return hip_check_launch();
    }
    hipLaunchKernelGGL(k_resize_bilinear_bwd, dim3((unsigned)((ih * iw + NT - 1) / NT), (unsigned)((planes + PLR - 1) / PLR)),
                       dim3(NT), 0, st, g_out, g_x, planes, ih, iw, oh, ow, scale_h, scale_w,
                       align_corners);
    return hip_check_launch();
}

int mvf_upsample_nearest_fwd(const float *x, float *out, int planes, int ih, int iw, int factor, void *stream)
{
    if (planes <= 0 || ih <= 0 || iw <= 0) return 0;
    if (!x || !out || factor < 1 || (planes + PLR - 1) / PLR > 65535) return (int)hipErrorInvalidValue;
    const int no = ih * factor * iw * factor;
    ProfScope ps(MVF_PROF_NEAREST_FWD, stream, 4LL * planes * ((int64_t)ih * iw + no));
    if (factor == 2 && (iw & 1) == 0 && (((uintptr_t)x) & 7) == 0 && (((uintptr_t)out) & 15) == 0 && !getenv("MVF_NEAREST_NARROW")) {
        hipLaunchKernelGGL(k_upsample_nearest2_fwd_w, dim3((unsigned)((ih * (iw / 2) + NT - 1) / NT), (unsigned)((planes + PLR - 1) / PLR)),
                           dim3(NT), 0, (hipStream_t)stream, x, out, planes, ih, iw);
        return hip_check_launch();
    }
    hipLaunchKernelGGL(k_upsample_nearest_fwd, dim3((unsigned)((no + NT - 1) / NT), (unsigned)((planes + PLR - 1) / PLR)),
                       dim3(NT), 0, (hipStream_t)stream, x, out, planes, ih, iw, factor);
    return hip_check_launch();
}

int mvf_upsample_nearest_bwd(const float *g_out, float *g_x, int planes, int ih, int iw, int factor, void *stream)
{
    if (planes <= 0 || ih <= 0 || iw <= 0) return 0;
    if (!g_out || !g_x || factor < 1 || (planes + PLR - 1) / PLR > 65535) return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_NEAREST_BWD, stream, 4LL * planes * (int64_t)ih * iw * (1 + factor * factor));
    if (factor == 2 && (iw & 1) == 0 && (((uintptr_t)g_x) & 7) == 0 && (((uintptr_t)g_out) & 15) == 0 && !getenv("MVF_NEAREST_NARROW")) {
        hipLaunchKernelGGL(k_upsample_nearest2_bwd_w, dim3((unsigned)((ih * (iw / 2) + NT - 1) / NT), (unsigned)((planes + PLR - 1) / PLR)),
                           dim3(NT), 0, (hipStream_t)stream, g_out, g_x, planes, ih, iw);
        return hip_check_launch();
    }
    hipLaunchKernelGGL(k_upsample_nearest_bwd, dim3((unsigned)((ih * iw + NT - 1) / NT), (unsigned)((planes + PLR - 1) / PLR)),
                       dim3(NT), 0, (hipStream_t)stream, g_out, g_x, planes, ih, iw, factor);
    return hip_check_launch();
}

// wide kernels: W % 4 == 0 (so OW = W / 2 is even and every plane row starts 16-byte aligned), aligned bases
static bool pool_wide_ok(const float *full, const float *pooled, const uint8_t *idx, int H, int W)
{
    return (W & 3) == 0 && W >= 4 && H >= 2 && (((uintptr_t)full) & 15) == 0 && (((uintptr_t)pooled) & 7) == 0 &&
           (((uintptr_t)idx) & 1) == 0;
}

int mvf_maxpool3s2_fwd(const float *x, float *out, uint8_t *idx, int planes, int H, int W, void *stream)
{
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    if (!x || !out || !idx || (planes + PLR - 1) / PLR > 65535) return (int)hipErrorInvalidValue;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    // input read once; pooled values (4 B) and window-local argmax (1 B) written
    ProfScope ps(MVF_PROF_MAXPOOL_FWD, stream, (int64_t)planes * (4LL * H * W + 5LL * OH * OW));
    if (pool_wide_ok(x, out, idx, H, W) && !getenv("MVF_POOL_NARROW")) {
        hipLaunchKernelGGL(k_maxpool3s2_fwd_w4, dim3((unsigned)((OH * (OW / 2) + NT - 1) / NT), (unsigned)((planes + PLW - 1) / PLW)),
                           dim3(NT), 0, (hipStream_t)stream, x, out, idx, planes, H, W, OH, OW);
        return hip_check_launch();
    }
    hipLaunchKernelGGL(k_maxpool3s2_fwd, dim3((unsigned)((OH * OW + NT - 1) / NT), (unsigned)((planes + PLR - 1) / PLR)),
                       dim3(NT), 0, (hipStream_t)stream, x, out, idx, planes, H, W, OH, OW);
    return hip_check_launch();
}

static int maxpool3s2_bwd_any(const float *g_out, const uint8_t *idx, const float *addend, float *g_x, int planes, int H,
                              int W, void *stream)
{
    if (planes <= 0 || H <= 0 || W <= 0) return 0;
    if (!g_out || !idx || !g_x || (planes + PLR - 1) / PLR > 65535) return (int)hipErrorInvalidValue;
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    ProfScope ps(MVF_PROF_MAXPOOL_BWD, stream, (int64_t)planes * ((addend ? 8LL : 4LL) * H * W + 5LL * OH * OW));
    const bool wide = pool_wide_ok(g_x, g_out, idx, H, W) && (H & 1) == 0 && !getenv("MVF_POOL_NARROW") &&
                      (((uintptr_t)addend) & 15) == 0;
    const dim3 gw((unsigned)(((H / 2) * (W / 4) + NT - 1) / NT), (unsigned)((planes + PLW - 1) / PLW));
    const dim3 gn((unsigned)((H * W + NT - 1) / NT), (unsigned)((planes + PLR - 1) / PLR));
    hipStream_t st = (hipStream_t)stream;
    if (wide && addend) hipLaunchKernelGGL(k_maxpool3s2_bwd_w4<true>, gw, dim3(NT), 0, st, g_out, idx, addend, g_x, planes, H, W, OH, OW);
    else if (wide) hipLaunchKernelGGL(k_maxpool3s2_bwd_w4<false>, gw, dim3(NT), 0, st, g_out, idx, addend, g_x, planes, H, W, OH, OW);
    else if (addend) hipLaunchKernelGGL(k_maxpool3s2_bwd<true>, gn, dim3(NT), 0, st, g_out, idx, addend, g_x, planes, H, W, OH, OW);
    else hipLaunchKernelGGL(k_maxpool3s2_bwd<false>, gn, dim3(NT), 0, st, g_out, idx, addend, g_x, planes, H, W, OH, OW);
    return hip_check_launch();
}

int mvf_maxpool3s2_bwd(const float *g_out, const uint8_t *idx, float *g_x, int planes, int H, int W, void *stream)
{
    return maxpool3s2_bwd_any(g_out, idx, nullptr, g_x, planes, H, W, stream);
}

int mvf_maxpool3s2_bwd_add(const float *g_out, const uint8_t *idx, const float *addend, float *g_x, int planes, int H,
                           int W, void *stream)
{
    if (planes > 0 && H > 0 && W > 0 && !addend) return (int)hipErrorInvalidValue;
    return maxpool3s2_bwd_any(g_out, idx, addend, g_x, planes, H, W, stream);
}

int mvf_bn_tile_many(const void *rows, int n_layers, int max_channels, int G, void *stream)
{
    if (n_layers <= 0 || G <= 0 || max_channels <= 0) return 0;
    if (!rows || n_layers > 65535) return (int)hipErrorInvalidValue;
    const int bx = (G * max_channels + NT - 1) / NT;
    hipLaunchKernelGGL(k_bn_tile_many, dim3((unsigned)(bx < 8 ? bx : 8), (unsigned)n_layers), dim3(NT), 0, (hipStream_t)stream,
                       reinterpret_cast<const BnRow *>(rows), G);
    return hip_check_launch();
}

int mvf_bn_fold_many(const void *rows, int n_layers, int max_channels, const float *coef, float beta, int G, void *stream)
{
    if (n_layers <= 0 || G <= 0 || max_channels <= 0) return 0;
    if (!rows || !coef || n_layers > 65535 || G > 32) return (int)hipErrorInvalidValue;
    BnFoldCoef cf;
    for (int g = 0; g < 32; ++g) cf.c[g] = g < G ? coef[g] : 0.0f;
    const int bx = (max_channels + NT - 1) / NT;
    hipLaunchKernelGGL(k_bn_fold_many, dim3((unsigned)(bx < 8 ? bx : 8), (unsigned)n_layers), dim3(NT), 0, (hipStream_t)stream,
                       reinterpret_cast<const BnRow *>(rows), cf, beta, G);
    return hip_check_launch();
}


int mvf_sum_act_fwd(const float *const *terms, int n_terms, float *out, int64_t total, int act, void *stream)
{
    if (total <= 0 || n_terms <= 0) return 0;
    if (!terms || !out || n_terms > SUM_MAX || (act != ACT_NONE && act != ACT_RELU)) return (int)hipErrorInvalidValue;
    SumTerms ts;
    bool vec = (total & 3) == 0 && (((uintptr_t)out) & 15) == 0;
    for (int j = 0; j < SUM_MAX; ++j) {
        ts.t[j] = j < n_terms ? terms[j] : nullptr;
        if (j < n_terms && (!terms[j] || (((uintptr_t)terms[j]) & 15))) { if (!terms[j]) return (int)hipErrorInvalidValue; vec = false; }
    }
    ts.n = n_terms;
    const int64_t per = vec ? total / 4 : total, blocks = (per + NT * 4 - 1) / (NT * 4);
    if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_SUM_ACT_FWD, stream, 4LL * (n_terms + 1) * total);
    if (vec) hipLaunchKernelGGL(k_sum_act_fwd<true>, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, ts, out, per, act);
    else hipLaunchKernelGGL(k_sum_act_fwd<false>, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream, ts, out, per, act);
    return hip_check_launch();
}


int mvf_regroup_fwd(const float *src, int G, int B, int64_t chunk, int n_out, float *const *dst, const int32_t *counts,
                    const int32_t *groups, void *stream)
{
    if (B <= 0 || chunk <= 0) return 0;
    if (!src || B > 65535 || (int64_t)B * G * chunk > 0x7fffffffffffLL) return (int)hipErrorInvalidValue;
    RegroupPlan pl;
    bool vec = (chunk & 3) == 0 && (((uintptr_t)src) & 15) == 0;
    const int err = regroup_plan(pl, dst, n_out, counts, groups, G, false, vec, chunk);
    if (err) return err;
    const int64_t per = vec ? chunk / 4 : chunk, blocks = (per + NT * 4 - 1) / (NT * 4);
    if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    // every slot reads one group and writes one group
    ProfScope ps(MVF_PROF_REGROUP_FWD, stream, 8LL * pl.nslots * B * chunk);
    const dim3 grid((unsigned)blocks, (unsigned)pl.nslots, (unsigned)B);
    if (vec) hipLaunchKernelGGL(k_regroup_fwd<true>, grid, dim3(NT), 0, (hipStream_t)stream, src, pl, G, per);
    else hipLaunchKernelGGL(k_regroup_fwd<false>, grid, dim3(NT), 0, (hipStream_t)stream, src, pl, G, per);
    return hip_check_launch();
}

int mvf_regroup_bwd(const float *const *g_dst, const int64_t *g_strides, int G, int B, int64_t chunk, int n_out,
                    const int32_t *counts, const int32_t *groups, float *g_src, void *stream)
{
    if (B <= 0 || chunk <= 0) return 0;
    if (!g_src || B > 65535 || (int64_t)B * G * chunk > 0x7fffffffffffLL) return (int)hipErrorInvalidValue;
    RegroupPlan pl;
    bool vec = (chunk & 3) == 0 && (((uintptr_t)g_src) & 15) == 0;
    const int err = regroup_plan(pl, const_cast<float *const *>(g_dst), n_out, counts, groups, G, true, vec, chunk,
                                 g_strides);
    if (err) return err;
    const int64_t per = vec ? chunk / 4 : chunk, blocks = (per + NT * 4 - 1) / (NT * 4);
    if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    // every differentiated slot read once, every source group written once
    ProfScope ps(MVF_PROF_REGROUP_BWD, stream, 4LL * (pl.nslots + G) * B * chunk);
    const dim3 grid((unsigned)blocks, (unsigned)G, (unsigned)B);
    if (vec) hipLaunchKernelGGL(k_regroup_bwd<true>, grid, dim3(NT), 0, (hipStream_t)stream, g_src, pl, G, per);
    else hipLaunchKernelGGL(k_regroup_bwd<false>, grid, dim3(NT), 0, (hipStream_t)stream, g_src, pl, G, per);
    return hip_check_launch();
}


int mvf_interleave_fwd(const float *const *src, const int64_t *len, const int64_t *offset, const int32_t *group,
                       int n_slots, float *dst, int G, int B, int64_t total, void *stream)
{
    if (B <= 0 || total <= 0 || n_slots <= 0) return 0;
    if (!src || !len || !offset || !group || !dst || n_slots > IL_MAX_SLOTS || G <= 0 || G > 255 || B > 65535)
        return (int)hipErrorInvalidValue;
    InterleavePlan pl;
    bool vec = (total & 3) == 0 && (((uintptr_t)dst) & 15) == 0;
    int64_t longest = 0, moved = 0;
    for (int s = 0; s < n_slots; ++s) {
        if (!src[s] || len[s] <= 0 || offset[s] < 0 || offset[s] + len[s] > total || group[s] < 0 || group[s] >= G)
            return (int)hipErrorInvalidValue;
        if ((len[s] & 3) || (offset[s] & 3) || (((uintptr_t)src[s]) & 15)) vec = false;
        longest = len[s] > longest ? len[s] : longest;
        moved += len[s];
    }
    for (int s = 0; s < IL_MAX_SLOTS; ++s) {
        const bool on = s < n_slots;
        pl.src[s] = on ? src[s] : nullptr;
        pl.len[s] = on ? (vec ? len[s] / 4 : len[s]) : 0;
        pl.off[s] = on ? (vec ? offset[s] / 4 : offset[s]) : 0;
        pl.group[s] = on ? (unsigned char)group[s] : 0;
    }
    const int64_t per = vec ? longest / 4 : longest, blocks = (per + NT * 4 - 1) / (NT * 4);
    if (blocks > 0x7fffffffLL) return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_INTERLEAVE_FWD, stream, 8LL * B * moved);     // every part read once, written once
    const dim3 grid((unsigned)blocks, (unsigned)n_slots, (unsigned)B);
    if (vec) hipLaunchKernelGGL(k_interleave_fwd<true>, grid, dim3(NT), 0, (hipStream_t)stream, pl, dst, G, total / 4);
    else hipLaunchKernelGGL(k_interleave_fwd<false>, grid, dim3(NT), 0, (hipStream_t)stream, pl, dst, G, total);
    return hip_check_launch();
}


int mvf_bn_tile(const float *weight, const float *bias, const float *running_mean, const float *running_var, float *tiled,
                int C, int G, void *stream)
{
    if (C <= 0 || G <= 0) return 0;
    if (!weight || !bias || !running_mean || !running_var || !tiled) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bn_tile, dim3((unsigned)((G * C + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, weight, bias,
                       running_mean, running_var, tiled, C, G);
    return hip_check_launch();
}

int mvf_bn_fold_running(float *running_mean, float *running_var, const float *upd_mean, const float *upd_var,
                        const float *coef, float beta, int C, int G, int64_t *num_batches_tracked, void *stream)
{
    if (C <= 0 || G <= 0) return 0;
    if (!running_mean || !running_var || !upd_mean || !upd_var || !coef || G > 32) return (int)hipErrorInvalidValue;
    BnFoldCoef cf;
    for (int g = 0; g < 32; ++g) cf.c[g] = g < G ? coef[g] : 0.0f;
    hipLaunchKernelGGL(k_bn_fold_running, dim3((unsigned)((C + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream,
                       running_mean, running_var, upd_mean, upd_var, cf, beta, C, G, num_batches_tracked);
    return hip_check_launch();
}

int mvf_bn_untile(const float *g_weight_tiled, const float *g_bias_tiled, float *g_out, int C, int G, void *stream)
{
    if (C <= 0 || G <= 0) return 0;
    if (!g_weight_tiled || !g_bias_tiled || !g_out) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bn_untile, dim3((unsigned)((C + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, g_weight_tiled,
                       g_bias_tiled, g_out, C, G);
    return hip_check_launch();
}


size_t mvf_color_jitter_workspace_floats(int images) { return (size_t)images * JIT_NB; }

int mvf_color_jitter(const float *img, const float *factors, const int32_t *order, const int32_t *apply,
                     const int32_t *flip, float *out_raw, float *out_aug, float *workspace, int samples,
                     int frames, int H, int W, void *stream)
{
    if (samples <= 0 || frames <= 0 || H <= 0 || W <= 0) return 0;
    if (!img || !factors || !order || !apply || !flip || !out_aug || !workspace || samples * frames > 65535)
        return (int)hipErrorInvalidValue;
    const int n = samples * frames, N = H * W;
    hipLaunchKernelGGL(k_jitter_mean, dim3(JIT_NB, (unsigned)n), dim3(NT), 0, (hipStream_t)stream, img, factors,
                       order, apply, workspace, N, frames);
    hipLaunchKernelGGL(k_jitter_apply, dim3((unsigned)((N + NT - 1) / NT), (unsigned)n), dim3(NT), 0,
                       (hipStream_t)stream, img, factors, order, apply, flip, workspace, out_raw, out_aug, H, W,
                       frames);
    return hip_check_launch();
}

}  // extern "C"
