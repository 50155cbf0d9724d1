// mvf_affine.hip -- the affine-augmentation glue either side of the hot path (SURVEY.md
// section 8f-2) for gfx950 (MI355X):
//
//   Trainer.affine_transform                       reference: train.py:888-902
//   depth_restore of compute_depth_consistency_loss_affine          train.py:909-916
//
// The reference runs both as a per-sample Python loop with five .item() host syncs per sample
// (angle, box) and 3-4 separate ATen launches per sample (rotate = affine grid + grid_sample,
// slice, interpolate, paste).  Here angle / box / ratio stay device tensors and each
// direction is one launch over the whole batch: every output pixel evaluates the *nested*
// bilinear interpolation (rotate of a resize, or resize of a rotate: 16 taps) directly, so no
// intermediate canvas is written.  The backward of the restore is two deterministic gather
// passes (no atomics): the rotation is an isometry, so the output pixels that touch a canvas
// pixel lie in a 3x3 neighbourhood of its inverse image; the resize adjoint is separable.
//
// One lane per pixel, lanes walk x: stores are coalesced; for |angle| <= 5 deg neighbouring
// lanes read neighbouring taps.  Bound: HBM bandwidth (8-16 B/px), in practice L1/TA.
//
// torchvision's rotate (bilinear, zero fill) is restated from its published algorithm
// (inverse-mapped pixel-centre grid, grid_sample align_corners=False): cos/sin are evaluated
// in double and rounded to fp32 like torchvision's python-float matrix.  Tolerance-level
// arithmetic (parity of the rotate step is unpinned: torchvision is on neither box).
#include "mvf_common.hpp"

using namespace mvf;

namespace {

constexpr int NT = 256;
constexpr int TX = 64, TY = 4;     // pixels per workgroup: 64 wide (one wave per row) x 4 rows

struct Trig {
    float c, s;
};

// per-workgroup: lane 0 evaluates the double-precision trig once
MVF_DEV Trig block_trig(float deg, Trig *sh)
{
    if (threadIdx.x == 0) {
        double a = (double)deg * (3.14159265358979323846 / 180.0);
        sh->c = (float)cos(a);
        sh->s = (float)sin(a);
    }
    __syncthreads();
    return *sh;
}

// source position (pixels) of output pixel (y,x) under torchvision's rotate:
// grid = (R [xs,ys]) / (0.5*size), then grid_sampler_unnormalize(align_corners=False)
MVF_DEV void rot_pos(Trig t, int y, int x, int H, int W, float &px, float &py)
{
    float xs = (float)x + 0.5f - (float)W / 2.0f, ys = (float)y + 0.5f - (float)H / 2.0f;
    float gx = (t.c * xs - t.s * ys) / (0.5f * (float)W);
    float gy = (t.s * xs + t.c * ys) / (0.5f * (float)H);
    px = ((gx + 1.0f) * (float)W - 1.0f) / 2.0f;
    py = ((gy + 1.0f) * (float)H - 1.0f) / 2.0f;
}

struct ZTap {                       // bilinear / zero padding at a real position
    int x0, y0;
    float lx, ly;
};
MVF_DEV ZTap ztap_of(float px, float py)
{
    float fx = floorf(px), fy = floorf(py);
    ZTap z;
    // positions far outside the image (or non-finite) contribute nothing: park them at -2
    z.x0 = (fx >= -2.0f && fx <= 1.0e6f) ? (int)fx : -2;
    z.y0 = (fy >= -2.0f && fy <= 1.0e6f) ? (int)fy : -2;
    z.lx = px - fx;
    z.ly = py - fy;
    return z;
}
MVF_DEV float sample_zeros(const float *__restrict__ im, int H, int W, const ZTap &z)
{
    const bool xa = z.x0 >= 0 && z.x0 < W, xb = z.x0 + 1 >= 0 && z.x0 + 1 < W;
    const bool ya = z.y0 >= 0 && z.y0 < H, yb = z.y0 + 1 >= 0 && z.y0 + 1 < H;
    const float *r0 = im + (size_t)max(z.y0, 0) * W, *r1 = im + (size_t)max(min(z.y0 + 1, H - 1), 0) * W;
    float v = 0.0f;
    if (ya && xa) v += r0[z.x0] * ((1.0f - z.lx) * (1.0f - z.ly));
    if (ya && xb) v += r0[z.x0 + 1] * (z.lx * (1.0f - z.ly));
    if (yb && xa) v += r1[z.x0] * ((1.0f - z.lx) * z.ly);
    if (yb && xb) v += r1[z.x0 + 1] * (z.lx * z.ly);
    return v;
}

// F.interpolate(bilinear, align_corners=False): source taps of output index o
// (ATen area_pixel_compute_source_index + the clamped upper tap)
struct RTap {
    int i0, i1;
    float l;
};
MVF_DEV RTap resize_src(int o, int in_size, int out_size)
{
    float sc = (float)in_size / (float)out_size;
    float t = fmaxf(((float)o + 0.5f) * sc - 0.5f, 0.0f);
    RTap r;
    r.i0 = min((int)t, in_size - 1);
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l = fminf(fmaxf(t - (float)r.i0, 0.0f), 1.0f);
    return r;
}

struct Box {
    int x0, y0, w, h;
};
// the box is the caller's contract (inside the image, w,h >= 1); clamped for memory safety
MVF_DEV Box box_of(const int32_t *__restrict__ box, int b, int H, int W)
{
    Box k;
    k.x0 = min(max(box[b * 4 + 0], 0), W - 1);
    k.y0 = min(max(box[b * 4 + 1], 0), H - 1);
    k.w = min(max(box[b * 4 + 2], 1), W - k.x0);
    k.h = min(max(box[b * 4 + 3], 1), H - k.y0);
    return k;
}

// ---------------------------------------------------------------- affine_transform forward
// out = resize( rotate(img, angle)[box] -> (H,W) ): 4 resize taps x 4 rotate taps per channel
__global__ void __launch_bounds__(NT) k_affine_transform(const float *__restrict__ img,
                                                         const float *__restrict__ angle,
                                                         const int32_t *__restrict__ box,
                                                         float *__restrict__ out, int C, int H, int W, int Bm)
{
    __shared__ Trig sh;
    const int b = blockIdx.z, bm = b % Bm;       // image b takes angle / box of sample b % Bm (several views per sample)
    const Trig t = block_trig(angle[bm], &sh);
    const int x = blockIdx.x * TX + (threadIdx.x & (TX - 1)), y = blockIdx.y * TY + threadIdx.x / TX;
    if (x >= W || y >= H) return;
    const Box k = box_of(box, bm, H, W);
    const RTap ry = resize_src(y, k.h, H), rx = resize_src(x, k.w, W);
    ZTap z[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float px, py;
        rot_pos(t, k.y0 + ((q >> 1) ? ry.i1 : ry.i0), k.x0 + ((q & 1) ? rx.i1 : rx.i0), H, W, px, py);
        z[q] = ztap_of(px, py);
    }
    const size_t N = (size_t)H * W;
    // Round 5: a pixel whose four rotate-tap sets all lie inside the image (everywhere but a thin border for the
    // +-5 degree rotations of the augmentation) takes its 16 taps as eight unconditional 8-byte row pairs per channel,
    // weights formed once -- sample_zeros' sixteen predicated 4-byte loads per channel were the kernel's cost
    // (50 us for 17.7 MB).  Same products, same left-to-right sums (0 + a == a): the same bits.
    bool inner = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) inner = inner && z[q].x0 >= 0 && z[q].x0 + 1 < W && z[q].y0 >= 0 && z[q].y0 + 1 < H;
    if (inner) {
        float w4[4][4];
        unsigned o0[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w4[q][0] = (1.0f - z[q].lx) * (1.0f - z[q].ly);
            w4[q][1] = z[q].lx * (1.0f - z[q].ly);
            w4[q][2] = (1.0f - z[q].lx) * z[q].ly;
            w4[q][3] = z[q].lx * z[q].ly;
            o0[q] = (unsigned)(z[q].y0 * W + z[q].x0) * 4u;
        }
        const unsigned W4b = (unsigned)W * 4u;
        for (int c = 0; c < C; ++c) {
            const float *im = img + ((size_t)b * C + c) * N;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float2 r0 = ldg2_at(im, o0[q]), r1 = ldg2_at(im, o0[q] + W4b);
                float a = r0.x * w4[q][0];
                a += r0.y * w4[q][1];
                a += r1.x * w4[q][2];
                a += r1.y * w4[q][3];
                v[q] = a;
            }
            out[((size_t)b * C + c) * N + (size_t)y * W + x] =
                (1.0f - ry.l) * ((1.0f - rx.l) * v[0] + rx.l * v[1]) + ry.l * ((1.0f - rx.l) * v[2] + rx.l * v[3]);
        }
        return;
    }
    for (int c = 0; c < C; ++c) {
        const float *im = img + ((size_t)b * C + c) * N;
        float v00 = sample_zeros(im, H, W, z[0]), v01 = sample_zeros(im, H, W, z[1]);
        float v10 = sample_zeros(im, H, W, z[2]), v11 = sample_zeros(im, H, W, z[3]);
        out[((size_t)b * C + c) * N + (size_t)y * W + x] =
            (1.0f - ry.l) * ((1.0f - rx.l) * v00 + rx.l * v01) + ry.l * ((1.0f - rx.l) * v10 + rx.l * v11);
    }
}

// ---------------------------------------------------------------- affine_restore forward
// canvas(Y,X) = inside the box ? resize(depth -> (h,w))(Y-y0, X-x0) : 0
MVF_DEV float canvas_at(const float *__restrict__ d, int H, int W, const Box &k, int Y, int X)
{
    if (X < k.x0 || X >= k.x0 + k.w || Y < k.y0 || Y >= k.y0 + k.h) return 0.0f;
    const RTap ry = resize_src(Y - k.y0, H, k.h), rx = resize_src(X - k.x0, W, k.w);
    const float *r0 = d + (size_t)ry.i0 * W, *r1 = d + (size_t)ry.i1 * W;
    return (1.0f - ry.l) * ((1.0f - rx.l) * r0[rx.i0] + rx.l * r0[rx.i1]) +
           ry.l * ((1.0f - rx.l) * r1[rx.i0] + rx.l * r1[rx.i1]);
}

// out = ratio * rotate( paste(resize(depth)) , -angle )
__global__ void __launch_bounds__(NT) k_affine_restore_fwd(const float *__restrict__ depth,
                                                           const float *__restrict__ angle,
                                                           const int32_t *__restrict__ box,
                                                           const float *__restrict__ ratio,
                                                           float *__restrict__ out, int C, int H, int W,
                                                           size_t in_stride)
{
    __shared__ Trig sh;
    const int b = blockIdx.z;
    const Trig t = block_trig(-angle[b], &sh);
    const int x = blockIdx.x * TX + (threadIdx.x & (TX - 1)), y = blockIdx.y * TY + threadIdx.x / TX;
    if (x >= W || y >= H) return;
    const Box k = box_of(box, b, H, W);
    float px, py;
    rot_pos(t, y, x, H, W, px, py);
    const ZTap z = ztap_of(px, py);
    const bool xa = z.x0 >= 0 && z.x0 < W, xb = z.x0 + 1 >= 0 && z.x0 + 1 < W;
    const bool ya = z.y0 >= 0 && z.y0 < H, yb = z.y0 + 1 >= 0 && z.y0 + 1 < H;
    const float r = ratio[b];
    const size_t N = (size_t)H * W;
    for (int c = 0; c < C; ++c) {
        const float *d = depth + (size_t)b * in_stride + (size_t)c * N;
        float v = 0.0f;
        if (ya && xa) v += canvas_at(d, H, W, k, z.y0, z.x0) * ((1.0f - z.lx) * (1.0f - z.ly));
        if (ya && xb) v += canvas_at(d, H, W, k, z.y0, z.x0 + 1) * (z.lx * (1.0f - z.ly));
        if (yb && xa) v += canvas_at(d, H, W, k, z.y0 + 1, z.x0) * ((1.0f - z.lx) * z.ly);
        if (yb && xb) v += canvas_at(d, H, W, k, z.y0 + 1, z.x0 + 1) * (z.lx * z.ly);
        out[((size_t)b * C + c) * N + (size_t)y * W + x] = v * r;
    }
}

// ---------------------------------------------------------------- affine_restore backward
// pass 1: g_canvas(Y,X) = ratio * sum over output pixels p whose rotate taps include (Y,X).
// The forward map p -> s(p) is a rotation about the centre, so those p lie within sqrt(2) of
// s^-1(Y,X): the three integers in (c-1.5, c+1.5) per axis.  Weights are recomputed with the
// forward's own expressions.  Only canvas pixels inside the box are written (pass 2 reads
// nothing else).
__global__ void __launch_bounds__(NT) k_affine_restore_bwd_rot(const float *__restrict__ g_out,
                                                               const float *__restrict__ angle,
                                                               const int32_t *__restrict__ box,
                                                               const float *__restrict__ ratio,
                                                               float *__restrict__ g_canvas, int C, int H,
                                                               int W)
{
    __shared__ Trig sh;
    const int b = blockIdx.z;
    const Trig t = block_trig(-angle[b], &sh);
    const int X = blockIdx.x * TX + (threadIdx.x & (TX - 1)), Y = blockIdx.y * TY + threadIdx.x / TX;
    if (X >= W || Y >= H) return;
    const Box k = box_of(box, b, H, W);
    if (X < k.x0 || X >= k.x0 + k.w || Y < k.y0 || Y >= k.y0 + k.h) return;
    // inverse rotation of the canvas pixel
    const float u = (float)X - ((float)W / 2.0f - 0.5f), v = (float)Y - ((float)H / 2.0f - 0.5f);
    const float cx = (t.c * u + t.s * v) - 0.5f + (float)W / 2.0f;
    const float cy = (-t.s * u + t.c * v) - 0.5f + (float)H / 2.0f;
    const int fx = (int)floorf(cx - 1.5f), fy = (int)floorf(cy - 1.5f);
    const size_t N = (size_t)H * W;
    const float r = ratio[b];
    float wgt[9];
    int off[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int y = fy + 1 + q / 3, x = fx + 1 + q % 3;
        wgt[q] = 0.0f;
        off[q] = 0;
        if (x < 0 || x >= W || y < 0 || y >= H) continue;
        float px, py;
        rot_pos(t, y, x, H, W, px, py);
        const ZTap z = ztap_of(px, py);
        float wx = (X == z.x0) ? 1.0f - z.lx : (X == z.x0 + 1) ? z.lx : 0.0f;
        float wy = (Y == z.y0) ? 1.0f - z.ly : (Y == z.y0 + 1) ? z.ly : 0.0f;
        wgt[q] = wx * wy;
        off[q] = y * W + x;
    }
    for (int c = 0; c < C; ++c) {
        const float *g = g_out + ((size_t)b * C + c) * N;
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 9; ++q)
            if (wgt[q] != 0.0f) acc += g[off[q]] * wgt[q];
        g_canvas[((size_t)b * C + c) * N + (size_t)Y * W + X] = acc * r;
    }
}

// candidate output indices k in [0,out) of a resize whose taps may include input index j
MVF_DEV void resize_candidates(int j, int in_size, int out_size, int &lo, int &hi)
{
    const float inv = (float)out_size / (float)in_size;      // 1/scale
    lo = (int)floorf(((float)j - 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)j + 1.5f) * inv - 0.5f) + 1;
    if (j >= in_size - 1) hi = out_size - 1;                  // clamped upper taps pile up here
    lo = max(lo, 0);
    hi = min(hi, out_size - 1);
}
MVF_DEV float resize_weight(int k, int j, int in_size, int out_size)
{
    const RTap r = resize_src(k, in_size, out_size);
    float w = 0.0f;
    if (r.i0 == j) w += 1.0f - r.l;
    if (r.i1 == j) w += r.l;
    return w;
}

// pass 2: g_depth(i,j) = sum over box pixels (ky,kx) of wy(ky,i) * wx(kx,j) * g_canvas
__global__ void __launch_bounds__(NT) k_affine_restore_bwd_resize(const float *__restrict__ g_canvas,
                                                                  const int32_t *__restrict__ box,
                                                                  float *__restrict__ g_depth, int C, int H,
                                                                  int W)
{
    const int b = blockIdx.z;
    const int j = blockIdx.x * TX + (threadIdx.x & (TX - 1)), i = blockIdx.y * TY + threadIdx.x / TX;
    if (j >= W || i >= H) return;
    const Box k = box_of(box, b, H, W);
    int ylo, yhi, xlo, xhi;
    resize_candidates(i, H, k.h, ylo, yhi);
    resize_candidates(j, W, k.w, xlo, xhi);
    const size_t N = (size_t)H * W;
    for (int c = 0; c < C; ++c) {
        const float *g = g_canvas + ((size_t)b * C + c) * N;
        float acc = 0.0f;
        for (int ky = ylo; ky <= yhi; ++ky) {
            const float wy = resize_weight(ky, i, H, k.h);
            if (wy == 0.0f) continue;
            float row = 0.0f;
            for (int kx = xlo; kx <= xhi; ++kx) {
                const float wx = resize_weight(kx, j, W, k.w);
                if (wx != 0.0f) row += wx * g[(size_t)(k.y0 + ky) * W + k.x0 + kx];
            }
            acc += wy * row;
        }
        g_depth[((size_t)b * C + c) * N + (size_t)i * W + j] = acc;
    }
}

inline dim3 tile_grid(int B, int H, int W)
{
    return dim3((unsigned)((W + TX - 1) / TX), (unsigned)((H + TY - 1) / TY), (unsigned)B);
}

}  // namespace

extern "C" {

int mvf_affine_transform_fwd(const float *img, const float *angle_deg, const int32_t *box, float *out,
                             int B, int C, int H, int W, void *stream)
{
    return mvf_affine_transform_views_fwd(img, angle_deg, box, out, B, B, C, H, W, stream);
}

int mvf_affine_transform_views_fwd(const float *img, const float *angle_deg, const int32_t *box, float *out,
                                   int B, int B_meta, int C, int H, int W, void *stream)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    if (!img || !angle_deg || !box || !out || B > 65535 || B_meta < 1 || B % B_meta) return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_AFFINE, stream, 8LL * B * C * H * W);
    hipLaunchKernelGGL(k_affine_transform, tile_grid(B, H, W), dim3(NT), 0, (hipStream_t)stream, img,
                       angle_deg, box, out, C, H, W, B_meta);
    return hip_check_launch();
}

int mvf_affine_restore_fwd(const float *depth, const float *angle_deg, const int32_t *box,
                           const float *ratio, float *out, int B, int C, int H, int W, void *stream)
{
    return mvf_affine_restore_strided_fwd(depth, 0, angle_deg, box, ratio, out, B, C, H, W, stream);
}

int mvf_affine_restore_strided_fwd(const float *depth, int64_t image_stride, const float *angle_deg, const int32_t *box,
                                   const float *ratio, float *out, int B, int C, int H, int W, void *stream)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    if (!depth || !angle_deg || !box || !ratio || !out || B > 65535 || image_stride < 0) return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_AFFINE, stream, 8LL * B * C * H * W);
    hipLaunchKernelGGL(k_affine_restore_fwd, tile_grid(B, H, W), dim3(NT), 0, (hipStream_t)stream, depth,
                       angle_deg, box, ratio, out, C, H, W, image_stride ? (size_t)image_stride : (size_t)C * H * W);
    return hip_check_launch();
}

int mvf_affine_restore_bwd(const float *g_out, const float *angle_deg, const int32_t *box,
                           const float *ratio, float *workspace, float *g_depth, int B, int C, int H,
                           int W, void *stream)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    if (!g_out || !angle_deg || !box || !ratio || !workspace || !g_depth || B > 65535)
        return (int)hipErrorInvalidValue;
    ProfScope ps(MVF_PROF_AFFINE, stream, 8LL * B * C * H * W);
    hipLaunchKernelGGL(k_affine_restore_bwd_rot, tile_grid(B, H, W), dim3(NT), 0, (hipStream_t)stream,
                       g_out, angle_deg, box, ratio, workspace, C, H, W);
    hipLaunchKernelGGL(k_affine_restore_bwd_resize, tile_grid(B, H, W), dim3(NT), 0, (hipStream_t)stream,
                       workspace, box, g_depth, C, H, W);
    return hip_check_launch();
}

}  // extern "C"
