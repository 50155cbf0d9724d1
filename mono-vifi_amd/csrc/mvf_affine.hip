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

// cos / sin of an angle in degrees, evaluated in double and rounded to fp32 (torchvision builds its matrix from python
// floats).  Rounds 2-5 had lane 0 of every workgroup call libm's double cos / sin and the other 255 lanes wait at a barrier
// for it: the four affine kernels took 24-39 us each for 24 MB (kernel trace of round 6) -- ocml's generic routines
// (argument reduction for any magnitude) run for microseconds on one lane.  Now EVERY lane evaluates it itself, without
// a barrier: Cody-Waite reduction by pi/2 in two pieces and fdlibm's kernel polynomials on [-pi/4, pi/4] (error below
// one double ulp: the fp32 rounding of the result can differ from libm's only when the exact value lies within
// ~1e-16 of a rounding boundary), ~30 double operations per lane.  |deg| < 1e6 (the augmentation draws +-5 degrees).
MVF_DEV Trig trig_of(float deg)
{
    const double x = (double)deg * (3.14159265358979323846 / 180.0);
    const double kd = rint(x * 0.63661977236758134308);
    const double r = fma(-kd, 6.12323399573676603587e-17, fma(-kd, 1.57079632679489655800e+00, x));
    const double z = r * r;
    const double sp = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double sn = r + (z * r) * (-1.66666666666666324348e-01 + z * sp);
    const double cp = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double cs = 1.0 - (0.5 * z - z * cp);
    const int k = (int)kd & 3;
    Trig t;
    t.c = (float)(k == 0 ? cs : k == 1 ? -sn : k == 2 ? -cs : sn);
    t.s = (float)(k == 0 ? sn : k == 1 ? cs : k == 2 ? -sn : -cs);
    return t;
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
// out = resize( rotate(img, angle)[box] -> (H,W) ): 4 resize taps x 4 rotate taps per channel.
// Round 6: the crop is UP-scaled (ratio 1.2 .. 2.0, mono_dataset.py:41), so neighbouring output pixels share their
// resize taps: a 64 x 8 output tile reads at most (64 + 2) x (8 + 2) crop pixels.  Rounds 2-5 evaluated the rotate at
// the four resize taps of EVERY output pixel (16 gathers + four position chains per pixel and channel: 52 us for
// 47 MB); now the workgroup rotates each crop pixel of its patch ONCE into LDS (stage A) and the output pixels take
// their four resize taps from there (stage B).  Same products in the same order per rotated sample and per output:
// the same bits as the per-pixel form.
constexpr int ATX = 64, ATY = 8;                   // output tile of the transform (two rows per lane)
constexpr int APW = ATX + 2, APH = ATY + 2;        // largest patch of crop pixels a tile can touch (scale <= 1)
constexpr int ACH_LDS = 4;                         // channels held in LDS at a time
constexpr int APOS = (APW * APH + NT - 1) / NT;    // patch positions per lane (3)

// rotated sample of crop pixel (cy, cx) (image coordinates), channel plane `im`: the four products in sample_zeros' order
struct RotTap {
    unsigned o0;          // byte offset of the row-y0 tap pair (inner positions)
    float w[4];
    ZTap z;
    bool inner;
};
MVF_DEV RotTap rot_tap(Trig t, int cy, int cx, int H, int W)
{
    RotTap r;
    float px, py;
    rot_pos(t, cy, cx, H, W, px, py);
    r.z = ztap_of(px, py);
    r.inner = r.z.x0 >= 0 && r.z.x0 + 1 < W && r.z.y0 >= 0 && r.z.y0 + 1 < H;
    r.w[0] = (1.0f - r.z.lx) * (1.0f - r.z.ly);
    r.w[1] = r.z.lx * (1.0f - r.z.ly);
    r.w[2] = (1.0f - r.z.lx) * r.z.ly;
    r.w[3] = r.z.lx * r.z.ly;
    r.o0 = r.inner ? (unsigned)(r.z.y0 * W + r.z.x0) * 4u : 0u;
    return r;
}
MVF_DEV float rot_sample(const float *__restrict__ im, int H, int W, const RotTap &r)
{
    if (r.inner) {
        const float2 r0 = ldg2_at(im, r.o0), r1 = ldg2_at(im, r.o0 + (unsigned)W * 4u);
        float a = r0.x * r.w[0];
        a += r0.y * r.w[1];
        a += r1.x * r.w[2];
        a += r1.y * r.w[3];
        return a;
    }
    return sample_zeros(im, H, W, r.z);
}

__global__ void __launch_bounds__(NT) k_affine_transform(const float *__restrict__ img,
                                                         const float *__restrict__ angle,
                                                         const int32_t *__restrict__ box,
                                                         float *__restrict__ out, int C, int H, int W, int Bm)
{
    __shared__ float patch[ACH_LDS][APH][APW + 1];
    const int b = blockIdx.z, bm = b % Bm;       // image b takes angle / box of sample b % Bm (several views per sample)
    const Trig t = trig_of(angle[bm]);
    const Box k = box_of(box, bm, H, W);
    const int lx = threadIdx.x & (ATX - 1), ly = threadIdx.x / ATX;          // 64 x 4 lanes, two output rows each
    const int x_lo = blockIdx.x * ATX, y_lo = blockIdx.y * ATY;
    const int x_hi = min(x_lo + ATX, W) - 1, y_hi = min(y_lo + ATY, H) - 1;
    // the patch: crop pixels [cy0, cy1] x [cx0, cx1] (resize taps are monotone in the output index)
    const int cx0 = resize_src(x_lo, k.w, W).i0, cx1 = resize_src(x_hi, k.w, W).i1;
    const int cy0 = resize_src(y_lo, k.h, H).i0, cy1 = resize_src(y_hi, k.h, H).i1;
    const int pw = cx1 - cx0 + 1, ph = cy1 - cy0 + 1;
    const bool fits = pw <= APW && ph <= APH;    // always for crops no larger than the image (the caller's contract)
    const size_t N = (size_t)H * W;
    const int x = x_lo + lx;
    RTap rx = resize_src(min(x, W - 1), k.w, W), ry[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) ry[j] = resize_src(min(y_lo + ly + 4 * j, H - 1), k.h, H);
    if (!fits) {
        // (a box larger than the contract allows: per-pixel form, no sharing)
        for (int j = 0; j < 2; ++j) {
            const int y = y_lo + ly + 4 * j;
            if (x >= W || y >= H) continue;
            for (int c = 0; c < C; ++c) {
                const float *im = img + ((size_t)b * C + c) * N;
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    v[q] = rot_sample(im, H, W, rot_tap(t, k.y0 + ((q >> 1) ? ry[j].i1 : ry[j].i0),
                                                        k.x0 + ((q & 1) ? rx.i1 : rx.i0), H, W));
                out[((size_t)b * C + c) * N + (size_t)y * W + x] =
                    (1.0f - ry[j].l) * ((1.0f - rx.l) * v[0] + rx.l * v[1]) + ry[j].l * ((1.0f - rx.l) * v[2] + rx.l * v[3]);
            }
        }
        return;
    }
    // stage A positions of this lane (row-major over the patch) and their rotate taps, shared by all channels
    RotTap rt[APOS];
    int pr[APOS], pc[APOS];
    const int npos = pw * ph;
#pragma unroll
    for (int i = 0; i < APOS; ++i) {
        const int p = min((int)threadIdx.x + i * NT, npos - 1);
        pr[i] = p / pw; pc[i] = p - pr[i] * pw;
        rt[i] = rot_tap(t, k.y0 + cy0 + pr[i], k.x0 + cx0 + pc[i], H, W);
    }
    for (int c0 = 0; c0 < C; c0 += ACH_LDS) {
        const int nc = min(ACH_LDS, C - c0);
        if (c0 > 0) __syncthreads();             // the previous chunk's reads are done
#pragma unroll
        for (int i = 0; i < APOS; ++i) {
            if ((int)threadIdx.x + i * NT >= npos) break;
            for (int c = 0; c < nc; ++c)
                patch[c][pr[i]][pc[i]] = rot_sample(img + ((size_t)b * C + c0 + c) * N, H, W, rt[i]);
        }
        __syncthreads();
        if (x < W) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int y = y_lo + ly + 4 * j;
                if (y >= H) continue;
                const int r0 = ry[j].i0 - cy0, r1 = ry[j].i1 - cy0, q0 = rx.i0 - cx0, q1 = rx.i1 - cx0;
                for (int c = 0; c < nc; ++c) {
                    const float v0 = patch[c][r0][q0], v1 = patch[c][r0][q1], v2 = patch[c][r1][q0], v3 = patch[c][r1][q1];
                    out[((size_t)b * C + c0 + c) * N + (size_t)y * W + x] =
                        (1.0f - ry[j].l) * ((1.0f - rx.l) * v0 + rx.l * v1) + ry[j].l * ((1.0f - rx.l) * v2 + rx.l * v3);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- affine_restore forward
// canvas(Y,X) = inside the box ? resize(depth -> (h,w))(Y-y0, X-x0) : 0
// out = ratio * rotate( paste(resize(depth)) , -angle )
__global__ void __launch_bounds__(NT) k_affine_restore_fwd(const float *__restrict__ depth,
                                                           const float *__restrict__ angle,
                                                           const int32_t *__restrict__ box,
                                                           const float *__restrict__ ratio,
                                                           float *__restrict__ out, int C, int H, int W,
                                                           size_t in_stride)
{
    const int b = blockIdx.z;
    const Trig t = trig_of(-angle[b]);
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
    // Round 6: everything about the four rotate taps that does not depend on the channel -- inside the image? inside
    // the box? the resize taps of the canvas pixel (two IEEE divides each) and the weights -- is formed ONCE; rounds
    // 2-5 re-derived it per channel inside canvas_at (three depth maps per image in the training step: 24 divides and
    // the box tests three times over).  Same loads, same products, same order of sums per channel: the same bits.
    bool use[4], inb[4];
    int o00[4], o01[4], o10[4], o11[4];
    float ly[4], lx[4], wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int Y = z.y0 + (q >> 1), X = z.x0 + (q & 1);
        use[q] = ((q >> 1) ? yb : ya) && ((q & 1) ? xb : xa);
        wq[q] = ((q & 1) ? z.lx : 1.0f - z.lx) * ((q >> 1) ? z.ly : 1.0f - z.ly);
        inb[q] = use[q] && !(X < k.x0 || X >= k.x0 + k.w || Y < k.y0 || Y >= k.y0 + k.h);
        o00[q] = o01[q] = o10[q] = o11[q] = 0;
        ly[q] = lx[q] = 0.0f;
        if (inb[q]) {
            const RTap ry = resize_src(Y - k.y0, H, k.h), rx = resize_src(X - k.x0, W, k.w);
            o00[q] = ry.i0 * W + rx.i0; o01[q] = ry.i0 * W + rx.i1;
            o10[q] = ry.i1 * W + rx.i0; o11[q] = ry.i1 * W + rx.i1;
            ly[q] = ry.l; lx[q] = rx.l;
        }
    }
    for (int c = 0; c < C; ++c) {
        const float *d = depth + (size_t)b * in_stride + (size_t)c * N;
        float v = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!use[q]) continue;
            float cv = 0.0f;
            if (inb[q])
                cv = (1.0f - ly[q]) * ((1.0f - lx[q]) * d[o00[q]] + lx[q] * d[o01[q]]) +
                     ly[q] * ((1.0f - lx[q]) * d[o10[q]] + lx[q] * d[o11[q]]);
            v += cv * wq[q];
        }
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
    const int b = blockIdx.z;
    const Trig t = trig_of(-angle[b]);
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
    // (round 6: the separable weights -- a resize_src with an IEEE divide each -- once per pixel, not once per channel;
    // ranges longer than the arrays below, i.e. crops far smaller than the contract's, take the per-channel form)
    constexpr int MAXC = 8;
    const int ny = yhi - ylo + 1, nx = xhi - xlo + 1;
    if (ny <= MAXC && nx <= MAXC) {
        float wy[MAXC], wx[MAXC];
#pragma unroll
        for (int a = 0; a < MAXC; ++a) {
            wy[a] = a < ny ? resize_weight(ylo + a, i, H, k.h) : 0.0f;
            wx[a] = a < nx ? resize_weight(xlo + a, j, W, k.w) : 0.0f;
        }
        for (int c = 0; c < C; ++c) {
            const float *g = g_canvas + ((size_t)b * C + c) * N;
            float acc = 0.0f;
#pragma unroll
            for (int a = 0; a < MAXC; ++a) {
                if (a >= ny || wy[a] == 0.0f) continue;
                float row = 0.0f;
#pragma unroll
                for (int e = 0; e < MAXC; ++e)
                    if (e < nx && wx[e] != 0.0f) row += wx[e] * g[(size_t)(k.y0 + ylo + a) * W + k.x0 + xlo + e];
                acc += wy[a] * row;
            }
            g_depth[((size_t)b * C + c) * N + (size_t)i * W + j] = acc;
        }
        return;
    }
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
    hipLaunchKernelGGL(k_affine_transform, dim3((unsigned)((W + ATX - 1) / ATX), (unsigned)((H + ATY - 1) / ATY), (unsigned)B),
                       dim3(NT), 0, (hipStream_t)stream, img, angle_deg, box, out, C, H, W, B_meta);
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
