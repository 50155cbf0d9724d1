// mvf_common.hpp -- device-side building blocks shared by the gfx950 kernels.
//
// "Exact mode" arithmetic contract: this
// translation unit is compiled with -ffp-contract=off, so `a*b + c` is two roundings;
// fmaf() appears ONLY where the reference's BLAS-backed [3xk]@[kxN] products accumulate
// (k-sequential chain seeded by a multiply); every `/` is the correctly rounded IEEE
// divide hipcc emits by default.  The integer sampling indices this produces are
// bit-identical to the reference's CPU run (pinned by tests/golden).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvf_hotpath.h"

#define MVF_DEV __device__ __forceinline__

namespace mvf {

constexpr int kWave = 64;

struct SrcPtrs {
    const float *p[MVF_MAX_SRC];
};
struct DstPtrs {
    float *p[MVF_MAX_SRC];
};

// ------------------------------------------------------------------------------- geometry
// depth = 1 / (min_disp + range*disp)          reference: layers.py:21-24
MVF_DEV float depth_of(float disp, float min_disp, float range)
{
    float scaled = min_disp + range * disp;
    return 1.0f / scaled;
}

// inv_K[:3,:3] @ [x,y,1]                        reference: layers.py:193
MVF_DEV void ray_of(const float *__restrict__ iK, float x, float y, float r[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float a = iK[i * 4 + 0] * x;
        a = fmaf(iK[i * 4 + 1], y, a);
        a = fmaf(iK[i * 4 + 2], 1.0f, a);
        r[i] = a;
    }
}

// P = (K @ T)[:3]                               reference: layers.py:212
// 4x4 @ 4x4 runs ATen's small-matrix loop on the reference's CPU path: products and sums
// rounded separately (pinned by golden key "P*").
MVF_DEV float proj_entry(const float *__restrict__ K, const float *__restrict__ T, int i, int j)
{
    float a = K[i * 4 + 0] * T[0 * 4 + j];
    a = a + K[i * 4 + 1] * T[1 * 4 + j];
    a = a + K[i * 4 + 2] * T[2 * 4 + j];
    a = a + K[i * 4 + 3] * T[3 * 4 + j];
    return a;
}

// c = P @ [X;1]                                 reference: layers.py:214
MVF_DEV void apply_P(const float P[12], const float X[3], float c[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float a = P[i * 4 + 0] * X[0];
        a = fmaf(P[i * 4 + 1], X[1], a);
        a = fmaf(P[i * 4 + 2], X[2], a);
        a = fmaf(P[i * 4 + 3], 1.0f, a);
        c[i] = a;
    }
}

// perspective divide + [-1,1] normalisation    reference: layers.py:216-221
MVF_DEV void normalise_uv(const float c[3], float eps, float wm1, float hm1, float &gx, float &gy,
                          float &u, float &v, float &z)
{
    z = c[2] + eps;
    u = c[0] / z;
    v = c[1] / z;
    float un = u / wm1;
    float vn = v / hm1;
    gx = (un - 0.5f) * 2.0f;
    gy = (vn - 0.5f) * 2.0f;
}

// bilinear tap of grid_sample(border, align_corners=True)   call site train.py:966-969
struct Tap {
    int x0, y0, x1, y1;
    float wx, wy;
    bool inx, iny;   // coordinate strictly inside (0, size-1): gradient passes
};

MVF_DEV float clip_coord(float v, float hi)
{
    // clamp to [0, hi] as one v_med3_f32 (the two selects it replaces are four instructions): the same value
    // for every non-NaN input, and a NaN comes out as 0 (with a NaN operand the instruction returns the
    // minimum of the others), as before: never indexes out of bounds
    return __builtin_amdgcn_fmed3f(v, 0.0f, hi);
}

MVF_DEV Tap tap_of_ixy(float ix, float iy, int H, int W);
MVF_DEV Tap tap_of(float gx, float gy, int H, int W)
{
    float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    float ix = ((gx + 1.0f) / 2.0f) * wm1;
    float iy = ((gy + 1.0f) / 2.0f) * hm1;
    return tap_of_ixy(ix, iy, H, W);
}
// the same tap from the un-normalised sample position (ix, iy)
MVF_DEV Tap tap_of_ixy(float ix, float iy, int H, int W)
{
    Tap t;
    float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    t.inx = (ix > 0.0f) && (ix < wm1);
    t.iny = (iy > 0.0f) && (iy < hm1);
    ix = clip_coord(ix, wm1);
    iy = clip_coord(iy, hm1);
    float fx = floorf(ix), fy = floorf(iy);
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    t.x1 = min(t.x0 + 1, W - 1);
    t.y1 = min(t.y0 + 1, H - 1);
    t.wx = ix - fx;
    t.wy = iy - fy;
    return t;
}

// The two horizontally adjacent taps of a row are fetched with ONE 8-byte load (4-byte
// aligned global_load_dwordx2).  The texture path turns a per-lane dword gather into 16-32
// cache accesses per wave instruction (measured: TCP_TOTAL_CACHE_ACCESSES /
// TA_FLAT_READ_WAVEFRONTS = 32), so halving the number of gather instructions halves the
// pressure on it.  At the right border (x0 == W-1, whose x1 weight is 0) the pair is
// anchored at W-2 and both taps read its second element.  Needs W >= 2.
struct __attribute__((packed, aligned(4))) F2U {
    float a, b;
};
MVF_DEV float2 ldg2(const float *__restrict__ p)
{
    F2U v = *reinterpret_cast<const F2U *>(p);
    return make_float2(v.a, v.b);
}
// the same load addressed as (plane base) + (32-bit BYTE offset).  With a wave-uniform base this
// is the scalar-base form of the instruction (global_load_dwordx2 v, v_off, s[base:base+1]): no
// 64-bit address arithmetic in the VALU.  Round 2's ISA spent ~28 VALU instructions per tap set
// of a source pair on it (v_mad_u64_u32, v_lshlrev_b64 and twelve v_lshl_add_u64, partly
// quarter-rate); a plane is far below 4 GiB, so the byte offset fits 32 bits by construction.
// (global address space stated explicitly: a pointer rebuilt from scalar registers is otherwise
// a generic one and the access becomes a flat_load)
#if defined(__HIP_DEVICE_COMPILE__)
#define MVF_GLOBAL __attribute__((address_space(1)))
#else
#define MVF_GLOBAL       /* host pass of the single-source compile: only parsed */
#endif
MVF_DEV float2 ldg2_at(const float *__restrict__ base, unsigned byte_off)
{
    F2U v = *(const MVF_GLOBAL F2U *)((const MVF_GLOBAL char *)base + byte_off);
    return make_float2(v.a, v.b);
}

MVF_DEV float ldg_at(const float *__restrict__ base, unsigned byte_off)
{
    return *(const MVF_GLOBAL float *)((const MVF_GLOBAL char *)base + byte_off);
}
MVF_DEV float2 ldg_f2_at(const float *__restrict__ base, unsigned byte_off)     // 8-byte aligned
{
    return *(const MVF_GLOBAL float2 *)((const MVF_GLOBAL char *)base + byte_off);
}
MVF_DEV void stg_at(float *__restrict__ base, unsigned byte_off, float v)
{
    *(MVF_GLOBAL float *)((MVF_GLOBAL char *)base + byte_off) = v;
}
MVF_DEV void stg_f2_at(float *__restrict__ base, unsigned byte_off, float2 v)     // 8-byte aligned
{
    *(MVF_GLOBAL float2 *)((MVF_GLOBAL char *)base + byte_off) = v;
}
MVF_DEV void stg_u8_at(uint8_t *__restrict__ base, unsigned byte_off, uint8_t v)
{
    *((MVF_GLOBAL uint8_t *)base + byte_off) = v;
}
// A pointer every lane of the wave holds the same value of (derived from blockIdx), moved to
// scalar registers.  hipcc computes 64-bit `base + b * stride` in the VALU (there is no scalar
// 64-bit multiply on gfx950) and then keeps every address derived from it in VGPR pairs.
template <typename T>
MVF_DEV T *uniform_ptr(T *p)
{
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T *>(((uint64_t)hi << 32) | lo);
}

struct TapRows {
    unsigned o0, o1;   // BYTE offsets of the tap pairs in rows y0 and y1 within one [H,W] plane
    bool sh;           // pair anchored one pixel left of x0 (right border)
};
MVF_DEV TapRows taprows_of(const Tap &t, int W)
{
    TapRows q;
    int xb = min(t.x0, W - 2);
    q.sh = t.x0 > xb;
    // 24-bit multiplies (full rate; y < 2^24 rows, 4 W < 2^24): v_mad_u32_u24
    const unsigned w4 = (unsigned)W * 4u, x4 = (unsigned)xb * 4u;
    q.o0 = __umul24((unsigned)t.y0, w4) + x4;
    q.o1 = __umul24((unsigned)t.y1, w4) + x4;
    return q;
}
MVF_DEV void load_taps(const float *__restrict__ im, const TapRows &q, float &nw, float &ne, float &sw,
                       float &se)
{
    float2 r0 = ldg2_at(im, q.o0), r1 = ldg2_at(im, q.o1);
    nw = q.sh ? r0.y : r0.x;
    ne = r0.y;
    sw = q.sh ? r1.y : r1.x;
    se = r1.y;
}

MVF_DEV float bilerp(const float *__restrict__ im, int W, const Tap &t)
{
    float w = t.wx, e = 1.0f - w, n = t.wy, s = 1.0f - n;
    float nw, ne, sw, se;
    load_taps(im, taprows_of(t, W), nw, ne, sw, se);
    return nw * (s * e) + ne * (s * w) + sw * (n * e) + se * (n * w);
}

MVF_DEV void bilerp_grad(const float *__restrict__ im, int W, const Tap &t, float &dx, float &dy)
{
    float w = t.wx, e = 1.0f - w, n = t.wy, s = 1.0f - n;
    float nw, ne, sw, se;
    load_taps(im, taprows_of(t, W), nw, ne, sw, se);
    dx = (ne - nw) * s + (se - sw) * n;
    dy = (sw - nw) * e + (se - ne) * w;
}

// full per-pixel chain of generate_images_pred up to the tap (reference train.py:956-969)
struct WarpPoint {
    Tap t;
    float gx, gy;       // normalised grid
    float u, v, z;      // perspective quotient and (z + eps)
    float X[3];         // camera point
    float r[3];         // ray
    float depth;
};

MVF_DEV WarpPoint warp_point(float disp, const float *__restrict__ iK, const float P[12], int x,
                             int y, int H, int W, float min_disp, float range, float eps)
{
    WarpPoint w;
    ray_of(iK, (float)x, (float)y, w.r);
    w.depth = depth_of(disp, min_disp, range);
    w.X[0] = w.depth * w.r[0];
    w.X[1] = w.depth * w.r[1];
    w.X[2] = w.depth * w.r[2];
    float c[3];
    apply_P(P, w.X, c);
    normalise_uv(c, eps, (float)(W - 1), (float)(H - 1), w.gx, w.gy, w.u, w.v, w.z);
    w.t = tap_of(w.gx, w.gy, H, W);
    return w;
}

// adjoint of the chain: (g_ix, g_iy) w.r.t. the un-normalised sample coordinate ->
// gc (grad of c = P@[X;1]) ; returns grad of depth.
MVF_DEV float warp_point_bwd(const WarpPoint &w, const float P[12], float gix, float giy, int H,
                             int W, float gc[3])
{
    float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    float ggx = w.t.inx ? gix * (wm1 / 2.0f) : 0.0f;
    float ggy = w.t.iny ? giy * (hm1 / 2.0f) : 0.0f;
    float gu = ggx * 2.0f / wm1;
    float gv = ggy * 2.0f / hm1;
    gc[0] = gu / w.z;
    gc[1] = gv / w.z;
    gc[2] = -(gu * w.u + gv * w.v) / w.z;
    float gd = 0.0f;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float gX = gc[0] * P[0 * 4 + j] + gc[1] * P[1 * 4 + j] + gc[2] * P[2 * 4 + j];
        gd += gX * w.r[j];
    }
    return gd;
}

// ------------------------------------------------------------------------------- flow warp (f1)
// IFRNet.warp / FusionModule.warp_features (reference: networks/IFRNet.py:7-15,
// networks/fusion_module.py:80-90): grid = linspace(-1,1) + flow / ((size-1)/2)
MVF_DEV Tap flow_tap_xy(float fx, float fy, const float *__restrict__ xs, const float *__restrict__ ys,
                        int x, int y, int H, int W)
{
    float gx = xs[x] + fx / (((float)W - 1.0f) / 2.0f);
    float gy = ys[y] + fy / (((float)H - 1.0f) / 2.0f);
    return tap_of(gx, gy, H, W);
}
MVF_DEV Tap flow_tap(const float *__restrict__ flow, const float *__restrict__ xs,
                     const float *__restrict__ ys, int b, int i, int x, int y, int H, int W)
{
    size_t N = (size_t)H * W;
    float fx = flow[((size_t)b * 2 + 0) * N + i], fy = flow[((size_t)b * 2 + 1) * N + i];
    return flow_tap_xy(fx, fy, xs, ys, x, y, H, W);
}


// The L2 executes fp32 atomics at about one dword per clock per channel, so the scatter is
// bound by the NUMBER of atomics (4 per element), not by bytes.  Lanes of a wavefront walk
// consecutive x; for a locally uniform integer displacement the east tap of lane l is the
// west tap of lane l+1 (same rows): lane l+1 then takes that contribution over with a
// cross-lane move and ONE atomic serves both -- 2 atomics per element instead of 4.  The
// hand-over is decided per lane pair from the integer taps, so any flow field is handled
// (lanes that do not line up keep their own atomics).
struct ScatterLinks {
    bool from_left;    // my west column == left neighbour's east column, same rows: I add its east values
    bool to_right;     // ... and the right neighbour takes my east values
};
MVF_DEV ScatterLinks scatter_links(const Tap &t, bool active)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int act = active ? 1 : 0;
    const int lx1 = __shfl_up(t.x1, 1), ly0 = __shfl_up(t.y0, 1), ly1 = __shfl_up(t.y1, 1),
              la = __shfl_up(act, 1);
    const int rx0 = __shfl_down(t.x0, 1), ry0 = __shfl_down(t.y0, 1), ry1 = __shfl_down(t.y1, 1),
              ra = __shfl_down(act, 1);
    ScatterLinks k;
    k.from_left = active && lane > 0 && la && lx1 == t.x0 && ly0 == t.y0 && ly1 == t.y1;
    k.to_right = active && lane < kWave - 1 && ra && rx0 == t.x1 && ry0 == t.y0 && ry1 == t.y1;
    return k;
}
// all lanes of the wavefront must call this (cross-lane moves); inactive lanes pass g = 0
MVF_DEV void scatter_taps_linked(float *__restrict__ gi, int W, const Tap &t, float g, bool active,
                                 const ScatterLinks &k)
{
    float w = t.wx, e = 1.0f - w, n = t.wy, s = 1.0f - n;
    float nw = g * (s * e), ne = g * (s * w), sw = g * (n * e), se = g * (n * w);
    const float lne = __shfl_up(ne, 1), lse = __shfl_up(se, 1);
    if (k.from_left) { nw += lne; sw += lse; }
    if (active) {
        atomicAdd(gi + t.y0 * W + t.x0, nw);
        atomicAdd(gi + t.y1 * W + t.x0, sw);
        if (!k.to_right) {
            atomicAdd(gi + t.y0 * W + t.x1, ne);
            atomicAdd(gi + t.y1 * W + t.x1, se);
        }
    }
}


// ------------------------------------------------------------------------------- const divides
// x/9 (3x3 window mean, reference layers.py:266-270) and x/3 (channel mean, train.py:977,982)
// as  q = x*c ; r = fma(-d,q,x) ; q' = fma(r,c,q)  with c = RN(1/d): bit-identical to the IEEE
// quotient for EVERY finite float (exhaustively verified by oracle/check_constdiv.c, run by
// tests/test_constdiv.py) at 3 instructions instead of ~10.
// MVF_FAST_SSIM (opt-in build, NOT the default and NOT parity mode): the window / channel means
// by reciprocal multiplies and the SSIM quotient by v_rcp -- the tolerance-mode arithmetic
// SURVEY.md section 7 ("SSIM exactness vs speed") asks to be measured beside exact mode.  The
// integer sampling indices stay exact; argmin / auto-mask may flip where candidates nearly tie.
#ifdef MVF_FAST_SSIM
MVF_DEV float div9(float x) { return x * (1.0f / 9.0f); }
MVF_DEV float div3(float x) { return x * (1.0f / 3.0f); }
#else
MVF_DEV float div9(float x)
{
    constexpr float c = 1.0f / 9.0f;
    float q = x * c;
    float r = fmaf(-9.0f, q, x);
    return fmaf(r, c, q);
}
MVF_DEV float div3(float x)
{
    constexpr float c = 1.0f / 3.0f;
    float q = x * c;
    float r = fmaf(-3.0f, q, x);
    return fmaf(r, c, q);
}
#endif

// packed pairs: two IEEE fp32 operations per lane per instruction (v_pk_*_f32)
#ifndef MVF_SCALAR_F2
typedef float f2 __attribute__((ext_vector_type(2)));

MVF_DEV f2 f2s(float a)
{
    f2 r = {a, a};
    return r;
}
MVF_DEV f2 mk2(float a, float b)
{
    f2 r = {a, b};
    return r;
}
MVF_DEV f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
MVF_DEV f2 pk_abs(f2 a) { return __builtin_elementwise_abs(a); }
#else
// Experiment (round 4): the same pair type as two independent scalars.  On gfx950 a v_pk_*_f32 instruction costs
// the issue time of two plain ones (tools/valu_ubench.hip: 1.8-2.2 ns vs 1.0), so packing buys nothing by itself
// and pays register moves wherever a pair has to be assembled from scalars; this form lets the compiler keep
// the halves wherever they are.  Same IEEE operations per half: exact mode is untouched.
struct alignas(8) f2 {
    float x, y;
};
MVF_DEV f2 f2s(float a) { return f2{a, a}; }
MVF_DEV f2 mk2(float a, float b) { return f2{a, b}; }
MVF_DEV f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
MVF_DEV f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
MVF_DEV f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
MVF_DEV f2 operator/(f2 a, f2 b) { return f2{a.x / b.x, a.y / b.y}; }
MVF_DEV f2 operator-(f2 a) { return f2{-a.x, -a.y}; }
MVF_DEV f2 operator*(float a, f2 b) { return f2{a * b.x, a * b.y}; }
MVF_DEV f2 operator*(f2 a, float b) { return f2{a.x * b, a.y * b}; }
MVF_DEV f2 operator/(f2 a, float b) { return f2{a.x / b, a.y / b}; }
MVF_DEV f2 &operator+=(f2 &a, f2 b) { a.x += b.x; a.y += b.y; return a; }
MVF_DEV f2 pk_fma(f2 a, f2 b, f2 c) { return f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
MVF_DEV f2 pk_abs(f2 a) { return f2{fabsf(a.x), fabsf(a.y)}; }
#endif
#ifdef MVF_FAST_SSIM
MVF_DEV f2 div9(f2 x) { return x * f2s(1.0f / 9.0f); }
MVF_DEV f2 div3(f2 x) { return x * f2s(1.0f / 3.0f); }
#else
MVF_DEV f2 div9(f2 x)
{
    const f2 c = f2s(1.0f / 9.0f);
    f2 q = x * c;
    f2 r = pk_fma(f2s(-9.0f), q, x);
    return pk_fma(r, c, q);
}
MVF_DEV f2 div3(f2 x)
{
    const f2 c = f2s(1.0f / 3.0f);
    f2 q = x * c;
    f2 r = pk_fma(f2s(-3.0f), q, x);
    return pk_fma(r, c, q);
}
#endif

// ---- correctly rounded fp32 division without the range guards --------------------------
// The compiler lowers `n / d` to v_div_scale x2, v_rcp, Newton step, quotient, two residual
// corrections, v_div_fmas, v_div_fixup (11 instructions).  The scale / fixup instructions
// only act when an exponent is extreme (|d| or |n| outside ~[2^-96, 2^96], denormals, 0,
// inf, NaN); everywhere else they are the identity and the result is that of the core
// below -- the same operations in the same order, so the same bits.  Used where the operand
// range is known by construction or checked by the caller (div_safe), it (a) drops the three
// guard instructions, (b) runs packed for a candidate pair, and (c) lets several quotients
// over the same denominator share the refined reciprocal.
MVF_DEV float recip_refined(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    float e = fmaf(-d, r, 1.0f);
    return fmaf(e, r, r);
}
MVF_DEV float div_core(float n, float d, float r1)
{
    float q = n * r1;
    float rem = fmaf(-d, q, n);
    float q1 = fmaf(rem, r1, q);
    float rem1 = fmaf(-d, q1, n);
    return fmaf(rem1, r1, q1);
}
MVF_DEV f2 recip_refined(f2 d)
{
    f2 r = mk2(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y));
    f2 e = pk_fma(-d, r, f2s(1.0f));
    return pk_fma(e, r, r);
}
MVF_DEV f2 div_core(f2 n, f2 d, f2 r1)
{
    f2 q = n * r1;
    f2 rem = pk_fma(-d, q, n);
    f2 q1 = pk_fma(rem, r1, q);
    f2 rem1 = pk_fma(-d, q1, n);
    return pk_fma(rem1, r1, q1);
}
// reciprocal and quotient of the SSIM formula (n/d with d > 0, O(1)): exact mode = the
// guard-free IEEE core; MVF_FAST_SSIM = v_rcp and one multiply
MVF_DEV f2 ssim_recip(f2 d)
{
    return recip_refined(d);              // v_rcp + one Newton step; in fast mode the quotient is then ONE multiply
}
MVF_DEV f2 ssim_quot(f2 n, f2 d, f2 r1)
{
#ifdef MVF_FAST_SSIM
    return n * r1;
#else
    return div_core(n, d, r1);
#endif
}
// exponent of |x| within [-60, 60] (finite, normal, far from over/underflow in the core)
MVF_DEV bool div_safe(float x)
{
    return (unsigned)(__builtin_amdgcn_frexp_expf(x) + 60) <= 121u;
}

// ------------------------------------------------------------------------------- SSIM
MVF_DEV int refl(int j, int n)
{
    j = (j < 0) ? -j : j;
    j = (j >= n) ? 2 * (n - 1) - j : j;
    return j;
}

// (float)(0.01**2), (float)(0.03**2)            reference: layers.py:274-275
constexpr float kC1 = (float)(0.01 * 0.01);
constexpr float kC2 = (float)(0.03 * 0.03);

struct Win {
    float mu_x, mu_y, exx, eyy, exy;
};

// reference: layers.py:281-290 -- literal expression order, no contraction
MVF_DEV float ssim_raw(const Win &w)
{
    float sigma_x = w.exx - w.mu_x * w.mu_x;
    float sigma_y = w.eyy - w.mu_y * w.mu_y;
    float sigma_xy = w.exy - w.mu_x * w.mu_y;
    float n = (2.0f * w.mu_x * w.mu_y + kC1) * (2.0f * sigma_xy + kC2);
    float d = (w.mu_x * w.mu_x + w.mu_y * w.mu_y + kC1) * (sigma_x + sigma_y + kC2);
    return (1.0f - n / d) / 2.0f;
}

MVF_DEV float clamp01(float v)
{
    float c = v < 0.0f ? 0.0f : v;
    return c > 1.0f ? 1.0f : c;
}

struct DWin {
    float dmux, dexx, dexy, dmuy, deyy;
};

// partial derivatives of the clamped SSIM map w.r.t. the window statistics
MVF_DEV DWin ssim_partials(const Win &w)
{
    float mx = w.mu_x, my = w.mu_y;
    float sigma_x = w.exx - mx * mx, sigma_y = w.eyy - my * my, sigma_xy = w.exy - mx * my;
    float A1 = 2.0f * mx * my + kC1, A2 = 2.0f * sigma_xy + kC2;
    float B1 = mx * mx + my * my + kC1, B2 = sigma_x + sigma_y + kC2;
    float n = A1 * A2, d = B1 * B2;
    float raw = (1.0f - n / d) / 2.0f;
    DWin g = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (!(raw >= 0.0f && raw <= 1.0f)) return g;
    float inv_d = 1.0f / d;
    float kn = -0.5f * inv_d;
    float kd = 0.5f * n * inv_d * inv_d;
    float dn_dmx = 2.0f * my * A2 - 2.0f * my * A1;
    float dn_dmy = 2.0f * mx * A2 - 2.0f * mx * A1;
    float dd_dmx = 2.0f * mx * B2 - 2.0f * mx * B1;
    float dd_dmy = 2.0f * my * B2 - 2.0f * my * B1;
    g.dmux = kn * dn_dmx + kd * dd_dmx;
    g.dmuy = kn * dn_dmy + kd * dd_dmy;
    g.dexy = kn * 2.0f * A1;
    g.dexx = kd * B1;
    g.deyy = kd * B1;
    return g;
}

// multiplicity of image column q in the reflect-padded 3-window centred on column p
MVF_DEV float refl_mult(int p, int q, int n)
{
    return ((p == 0 && q == 1) || (p == n - 1 && q == n - 2)) ? 2.0f : 1.0f;
}

// ------------------------------------------------------------------------------- reductions
// Wavefront sum on the DPP path (no LDS crossbar): four row-shift steps leave each 16-lane row's
// total in its last lane, two row broadcasts fold the rows; the total is valid in LANE 63 only.
// (hipcc lowers __shfl_down to ds_bpermute: 6 LDS operations per value; a workgroup reduction of
// 27 values was 162 of them.)  The compiler folds each move into its add (v_add_f32_dpp).
MVF_DEV float dpp_add(float v, int ctrl_is /*0: shr1, 1: shr2, 2: shr4, 3: shr8, 4: bcast15, 5: bcast31*/)
{
    int r;
    const int x = __builtin_bit_cast(int, v);
    switch (ctrl_is) {
    case 0: r = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true); break;
    case 1: r = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true); break;
    case 2: r = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true); break;
    case 3: r = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true); break;
    case 4: r = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); break;
    default: r = __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); break;
    }
#ifdef MVF_PIN_DPP
    asm volatile("" : "+v"(r));
#endif
    return v + __builtin_bit_cast(float, r);
}
MVF_DEV float wave_sum(float v)      // result in lane 63
{
#pragma unroll
    for (int k = 0; k < 6; ++k) v = dpp_add(v, k);
    return v;
}

// ---- message passing between the blocks of the small kernels (unit preparing / finishing, SI-log finishing) -----------------------------
// A block publishes a few values and takes a ticket; the block that draws the last ticket reads everybody's values.
// __threadfence() on this chip is `buffer_wbl2 sc1` + wait: a write-back of the XCD's whole L2 (4 MiB, full of the unit
// kernel's gradient planes) per call -- measured: 60 us for the 2,304 blocks of one preparing launch.  Instead every
// published value is an agent-scope atomic store (sc1: written through to the device-coherent level), the ticket is
// taken after `s_waitcnt vmcnt(0)` (the stores have completed), and the reader uses agent-scope atomic loads (sc1:
// not served from a stale L2 line): the same ordering for exactly the values involved, no cache maintenance.
// This ordering argument is the gfx9 family's: ONE counter (vmcnt) covers loads AND stores, and sc1 stores write
// through.  Targets with a separate store counter (vscnt: gfx10 and later) would let the ticket overtake the published
// values -- refuse to build the device code for anything else; this library is written for gfx950.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "publish / take_ticket rely on gfx9 vmcnt semantics (stores counted by vmcnt): build for gfx950"
#endif
template <typename Tv>
MVF_DEV void publish(Tv *p, Tv v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename Tv>
MVF_DEV Tv fetch_published(const Tv *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
MVF_DEV int take_ticket(int *tk)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // everything published before is complete
    return __hip_atomic_fetch_add(tk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum over the workgroup; result valid in thread 0.  `scratch` holds >= nwaves floats.
template <int NT>
MVF_DEV float block_sum(float v, float *scratch)
{
    constexpr int NW = NT / kWave;
    v = wave_sum(v);
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) scratch[wid] = v;
    __syncthreads();
    float r = 0.0f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) r += scratch[i];
    }
    return r;
}

// sum NV per-lane values over the workgroup with TWO barriers in total: four DPP row-shift steps
// leave every 16-lane row's sum in its last lane, those lanes park it in LDS (one slot per row),
// then lane q < NV folds the rows of all waves in a fixed order.  Result for value q is returned in
// lane q of the workgroup (threadIdx.x == q); `scratch` holds >= (NT/16)*NV floats.
template <int NT, int NV>
MVF_DEV float block_sum_many(const float (&v)[NV], float *scratch)
{
    constexpr int NR = NT / 16;                     // 16-lane rows in the workgroup
    const int row = threadIdx.x >> 4;
    const bool last = (threadIdx.x & 15) == 15;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        float s = v[q];
#pragma unroll
        for (int k = 0; k < 4; ++k) s = dpp_add(s, k);
        if (last) scratch[row * NV + q] = s;
    }
    __syncthreads();
    float r = 0.0f;
    if (threadIdx.x < NV) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r += scratch[i * NV + threadIdx.x];
    }
    return r;
}

// The same sum through an LDS transpose: every lane parks its NV values (value-major, stride NT + 8 so that
// both the writes and the strided reads below are bank-conflict free), then lane (q, c) -- NV x 8 of them --
// adds up the 32 entries c, c + 8, ... of value q, and lane q folds the 8 chunk sums.  31 + 8 additions per
// lane instead of 4 DPP steps for each of the NV values (NV = 27: 108) -- the workgroup reduction was 5 % of
// the unit kernel's VALU instructions.  Fixed order: deterministic.  Result for value q in lane q;
// `scratch` holds >= NV * (NT + 8) + NV * 8 floats (CH = 8 chunks; 4 when NV * 8 > NT) and must be free of live data (barrier on entry).
template <int NT, int NV>
MVF_DEV float block_sum_many_lds(const float (&v)[NV], float *scratch)
{
    constexpr int STR = NT + 8, CH = (NV * 8 <= NT) ? 8 : 4, PER = NT / CH;      // (128-lane workgroups: four chunks)
    static_assert(NT % CH == 0 && NV * CH <= NT, "one (value, chunk) pair per lane");
    const int t = threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) scratch[q * STR + t] = v[q];
    __syncthreads();
    float *part = scratch + NV * STR;
    if (t < NV * CH) {
        const int q = t / CH, c = t - q * CH;
        const float *p = scratch + q * STR + c;
        float s = p[0];
#pragma unroll
        for (int i = 1; i < PER; ++i) s += p[CH * i];
        part[t] = s;
    }
    __syncthreads();
    float r = 0.0f;
    if (t < NV) {
#pragma unroll
        for (int c = 0; c < CH; ++c) r += part[t * CH + c];
    }
    return r;
}

inline int hip_check_launch()
{
    return (int)hipGetLastError();
}

// measurement hooks (implemented in mvf_geom.hip): event pair around one kernel launch (or one launcher's
// group of launches).  `work`: pixels for the hot-path ids (< MVF_PROF_UNITS_FINISH), algorithmic bytes for
// the others; `tag`: launch kind (MVF_TAG_*)
void prof_begin(int kernel_id, hipStream_t st);
void prof_end(int kernel_id, hipStream_t st, int64_t work, int tag);
struct ProfScope {
    int id, tag;
    hipStream_t st;
    int64_t work;
    ProfScope(int kernel_id, hipStream_t s, int64_t w = 0, int t = 0) : id(kernel_id), tag(t), st(s), work(w) { prof_begin(id, st); }
    ProfScope(int kernel_id, void *s, int64_t w = 0, int t = 0) : ProfScope(kernel_id, (hipStream_t)s, w, t) {}
    ~ProfScope() { prof_end(id, st, work, tag); }
};

}  // namespace mvf
