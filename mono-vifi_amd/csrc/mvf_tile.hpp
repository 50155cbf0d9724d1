// mvf_tile.hpp -- the tile engine shared by the photometric kernels (mvf_photo.hip) and the
// forward+backward unit kernel (mvf_unit_fb.hip): region / plane geometry, XCD-aware tile order,
// reflect-addressed plane staging, packed-pair window statistics, the SSIM formula and its
// partial derivatives for a candidate pair, and the fused warp of a source pair into LDS.
// Included into an anonymous namespace by each translation unit (internal linkage).
// Arithmetic contract: mvf_common.hpp (exact mode, -ffp-contract=off).
#pragma once
#include "mvf_common.hpp"

using namespace mvf;

// Geometry knobs, set by the including translation unit BEFORE the include (each .hip is its own
// anonymous-namespace instance): region width and pixels per lane.  64 x 16 / 4 px (256 lanes,
// ~75 KB LDS in the forward+backward kernel: 2 workgroups per CU) or 32 x 16 / 2 px (256 lanes,
// ~40 KB: 3-4 workgroups per CU, half the per-lane register state).
#ifndef MVF_TILE_TW
#define MVF_TILE_TW 64
#endif
#ifndef MVF_TILE_PX
#define MVF_TILE_PX 4
#endif
#ifndef MVF_TILE_TH
#define MVF_TILE_TH 16
#endif

namespace {

#ifndef MVF_TILE_PY
#define MVF_TILE_PY 1
#endif
constexpr int TW = MVF_TILE_TW, TH = MVF_TILE_TH; // compute region
constexpr int PX = MVF_TILE_PX;          // pixels per lane along x (one row segment)
constexpr int PY = MVF_TILE_PY;          // rows per lane (mvf_unit_fb.hip: a lane owns a PY x PX block of the region)
constexpr int RW = PX + 2;               // plane columns a lane's 3x3 windows span
static_assert(PX == 4 || PX == 2, "row loaders below handle 4 or 2 pixels per lane");
static_assert(MVF_TILE_TH % MVF_TILE_PY == 0, "whole row blocks");
constexpr int NT = (TW / PX) * (TH / PY);       // lanes per workgroup (256 at PY = 1)
constexpr int PW = TW + 2;               // staged plane width (1-px halo)
constexpr int PH = TH + 2;
constexpr int LDW = TW + 4;              // LDS row stride (floats), multiple of 4
constexpr int PLANE = PH * LDW;          // floats per LDS plane
constexpr int RPLANE = TH * LDW;         // floats per region-sized LDS plane
constexpr int NMEAN = 32;                // partial sums per image of the disp mean
constexpr int NPART = 4;                 // floats per tile partial (photo, sx, sy, pad)

static_assert(NT % kWave == 0 && TW / PX == 16, "whole waves; 16 lanes per region row (DPP rows)");

// XCD-aware tile order.  The dispatcher places workgroup i on XCD i % 8 (private 4 MiB L2
// each).  Re-number so that each XCD owns one contiguous run of tiles (neighbouring tiles of
// the same image rows): the 1-px halo columns/rows and the bilinear taps a tile shares with
// its neighbours then hit in that XCD's L2 instead of being fetched once per XCD.  Pure
// speed choice -- any placement gives the same results.
struct TileId {
    int bx, by, b;
};
MVF_DEV TileId tile_of_block(int tiles_x, int tiles_y, int B)
{
    const int total = tiles_x * tiles_y * B;
    const int lin = blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    const int q = total >> 3, r = total & 7;
    const int vid = xcd * q + min(xcd, r) + slot;
    TileId t;
    t.bx = vid % tiles_x;
    const int rest = vid / tiles_x;
    t.by = rest % tiles_y;
    t.b = rest / tiles_y;
    return t;
}

MVF_DEV int refl_clamp(int j, int n)
{
    j = (j < 0) ? -j : j;
    j = (j >= n) ? 2 * (n - 1) - j : j;
    return min(max(j, 0), n - 1);
}

// byte offset of pixel (gy, gx) inside one [H,W] fp32 plane: 24-bit multiply (full rate), 32-bit
// result -- with a wave-uniform plane base the load takes the scalar-base form (mvf_common.hpp)
MVF_DEV unsigned plane_off4(int gy, int gx, int W)
{
    return __umul24((unsigned)gy, (unsigned)W * 4u) + (unsigned)gx * 4u;
}

// stage one [H,W] plane into LDS with reflect addressing; plane origin (py0, px0)
constexpr int NSTAGE = (PH * PW + NT - 1) / NT;   // plane elements per lane (5)

// Byte offsets (inside one [H,W] plane) of the NSTAGE plane elements a lane stages.  `inner`
// (workgroup-uniform: the whole staged plane lies inside the image -- 78 % of the tiles at 640x192)
// skips the reflect / clamp mapping of both coordinates behind ONE scalar branch: ~15 VALU
// instructions per element that only the tiles on the image border need.
MVF_DEV void stage_offsets(unsigned (&o)[NSTAGE], int H, int W, int py0, int px0, bool inner)
{
    if (inner) {
        const unsigned base = plane_off4(py0, px0, W);
#pragma unroll
        for (int it = 0; it < NSTAGE; ++it) {
            int idx = min((int)threadIdx.x + it * NT, PH * PW - 1);
            int r = idx / PW, c = idx - r * PW;
            o[it] = base + plane_off4(r, c, W);
        }
    } else {
#pragma unroll
        for (int it = 0; it < NSTAGE; ++it) {
            int idx = min((int)threadIdx.x + it * NT, PH * PW - 1);
            int r = idx / PW, c = idx - r * PW;
            o[it] = plane_off4(refl_clamp(py0 + r, H), refl_clamp(px0 + c, W), W);
        }
    }
}

// All global loads of a lane are issued before its first LDS store (fully unrolled,
// constant trip count): one exposed memory latency per staging phase instead of five.
MVF_DEV void stage_plane(float *__restrict__ lds, const float *__restrict__ img, int H, int W,
                         int py0, int px0, bool inner = false)
{
    float v[NSTAGE];
    unsigned o[NSTAGE];
    stage_offsets(o, H, W, py0, px0, inner);
#pragma unroll
    for (int it = 0; it < NSTAGE; ++it) v[it] = ldg_at(img, o[it]);
#pragma unroll
    for (int it = 0; it < NSTAGE; ++it) {
        int idx = threadIdx.x + it * NT;
        int r = idx / PW, c = idx - r * PW;
        if (idx < PH * PW) lds[r * LDW + c] = v[it];
    }
}

// stage 3 channel planes with all loads of a lane in flight together
MVF_DEV void stage_planes3(float *__restrict__ lds, const float *__restrict__ img, size_t N, int H,
                           int W, int py0, int px0, bool inner = false)
{
    float v[NSTAGE][3];
    unsigned off[NSTAGE];
    stage_offsets(off, H, W, py0, px0, inner);
#pragma unroll
    for (int it = 0; it < NSTAGE; ++it) {
        const unsigned o = off[it];
        v[it][0] = ldg_at(img, o); v[it][1] = ldg_at(img + N, o); v[it][2] = ldg_at(img + 2 * N, o);
    }
#pragma unroll
    for (int it = 0; it < NSTAGE; ++it) {
        int idx = threadIdx.x + it * NT;
        int r = idx / PW, c = idx - r * PW;
        if (idx < PH * PW) {
            lds[r * LDW + c] = v[it][0];
            lds[PLANE + r * LDW + c] = v[it][1];
            lds[2 * PLANE + r * LDW + c] = v[it][2];
        }
    }
}

// ---- packed fp32 pairs -----------------------------------------------------------------
// gfx950 VALU executes v_pk_{add,mul,fma}_f32: two IEEE fp32 operations per lane per
// instruction.  The tile kernels evaluate candidates TWO AT A TIME (e.g. the two warped
// sources, then the two identity sources): the pair is stored interleaved in LDS as
// float2, so one ds_read_b128 feeds both, every window add / product / formula step is one
// packed instruction for both candidates, and no register shuffles are needed.  Each lane
// of a packed op is the same IEEE operation as the scalar form, so exact mode is unchanged.
constexpr int PPLANE = PH * LDW;   // f2 elements per staged pair plane
constexpr int RPPLANE = TH * LDW;  // f2 elements per region-sized pair plane

// RW = PX+2 consecutive floats of an LDS plane row, starting at column PX*seg (16-B aligned for
// PX = 4, 8-B aligned for PX = 2)
struct Row6 {
    float v[RW];
};
MVF_DEV Row6 load_row6(const float *__restrict__ p)
{
    Row6 r;
    if constexpr (PX == 4) {
        float4 a = *reinterpret_cast<const float4 *>(p);
        float2 b = *reinterpret_cast<const float2 *>(p + 4);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y;
    } else {
        float2 a = *reinterpret_cast<const float2 *>(p), b = *reinterpret_cast<const float2 *>(p + 2);
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = b.x; r.v[3] = b.y;
    }
    return r;
}
// RW consecutive pairs of a pair-plane row (16-B aligned)
struct Row6P {
    f2 v[RW];
};
MVF_DEV Row6P load_row6p(const f2 *__restrict__ p)
{
    Row6P r;
    const float4 *q = reinterpret_cast<const float4 *>(p);
    float4 a = q[0], b = q[1];
    r.v[0] = mk2(a.x, a.y); r.v[1] = mk2(a.z, a.w); r.v[2] = mk2(b.x, b.y); r.v[3] = mk2(b.z, b.w);
    if constexpr (PX == 4) {
        float4 c = q[2];
        r.v[4] = mk2(c.x, c.y); r.v[5] = mk2(c.z, c.w);
    }
    return r;
}

// Window sums of the 4 pixels of a lane for a candidate PAIR and for the target, row-major
// sequential order (exact mode).  xs/ys point at plane element (row, 4*seg): window of pixel
// j = cols j..j+2.  The target statistics ride along as a packed (y, y*y) accumulator; they
// are recomputed per pair instead of being held in 24 registers across the whole kernel.
struct Stats4P {
    f2 sx[PX], sxx[PX], sxy[PX];
    f2 sy[PX];      // (sum y, sum y*y)
    f2 xc[PX];      // centre values of the pair
    float yc[PX];   // centre values of the target
};

MVF_DEV void window_xp(const f2 *__restrict__ xs, const float *__restrict__ ys, Stats4P &o)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        Row6P x = load_row6p(xs + r * LDW);
        Row6 y = load_row6(ys + r * LDW);
        f2 xx[RW], xy[RW], yy[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            xx[i] = x.v[i] * x.v[i];
            xy[i] = x.v[i] * f2s(y.v[i]);
            yy[i] = mk2(y.v[i], y.v[i] * y.v[i]);
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (r == 0 && d == 0) {
                    o.sx[j] = x.v[j];
                    o.sxx[j] = xx[j];
                    o.sxy[j] = xy[j];
                    o.sy[j] = yy[j];
                } else {
                    o.sx[j] = o.sx[j] + x.v[j + d];
                    o.sxx[j] = o.sxx[j] + xx[j + d];
                    o.sxy[j] = o.sxy[j] + xy[j + d];
                    o.sy[j] = o.sy[j] + yy[j + d];
                }
            }
            if (r == 1) {
                o.xc[j] = x.v[j + 1];
                o.yc[j] = y.v[j + 1];
            }
        }
    }
}

// reference: layers.py:281-290 for a candidate pair -- literal expression order per lane
MVF_DEV f2 ssim_raw_pk(f2 mu_x, f2 mu_y, f2 exx, f2 eyy, f2 exy)
{
#ifdef MVF_FAST_SSIM     // opt-in fast mode: the same formula contracted (mvf_unit_fb.hip: ssim_val_partials_pk)
    {
        const f2 mxx = mu_x * mu_x, myy = mu_y * mu_y, mxy = mu_x * mu_y;
        const f2 A1 = pk_fma(f2s(2.0f), mxy, f2s(kC1)), A2 = pk_fma(f2s(2.0f), exy - mxy, f2s(kC2));
        const f2 B1 = (mxx + myy) + f2s(kC1), B2 = ((exx - mxx) + (eyy - myy)) + f2s(kC2);
        const f2 dd = B1 * B2;
        return pk_fma(ssim_quot(A1 * A2, dd, ssim_recip(dd)), f2s(-0.5f), f2s(0.5f));
    }
#endif
    f2 sigma_x = exx - mu_x * mu_x;
    f2 sigma_y = eyy - mu_y * mu_y;
    f2 sigma_xy = exy - mu_x * mu_y;
    f2 n = (2.0f * mu_x * mu_y + f2s(kC1)) * (2.0f * sigma_xy + f2s(kC2));
    f2 d = (mu_x * mu_x + mu_y * mu_y + f2s(kC1)) * (sigma_x + sigma_y + f2s(kC2));
    // d >= C1*(C2 - rounding) > 0 and |n|, d = O(1) for images in [0,1]: the guard-free
    // division core gives the correctly rounded quotient (see mvf_common.hpp)
    return (f2s(1.0f) - ssim_quot(n, d, ssim_recip(d))) / 2.0f;
}

MVF_DEV f2 clamp01_pk(f2 v) { return mk2(clamp01(v.x), clamp01(v.y)); }

// x-side partial derivatives of the clamped SSIM map for a candidate pair
MVF_DEV void ssim_partials_pk(f2 mx, f2 my, f2 exx, f2 eyy, f2 exy, f2 &dmux, f2 &dexx, f2 &dexy)
{
    f2 sigma_x = exx - mx * mx, sigma_y = eyy - my * my, sigma_xy = exy - mx * my;
    f2 A1 = 2.0f * mx * my + f2s(kC1), A2 = 2.0f * sigma_xy + f2s(kC2);
    f2 B1 = mx * mx + my * my + f2s(kC1), B2 = sigma_x + sigma_y + f2s(kC2);
    f2 n = A1 * A2, d = B1 * B2;
    const f2 r1 = ssim_recip(d);          // shared by n/d and 1/d
    f2 raw = (f2s(1.0f) - ssim_quot(n, d, r1)) / 2.0f;
    f2 live = mk2((raw.x >= 0.0f && raw.x <= 1.0f) ? 1.0f : 0.0f,
                  (raw.y >= 0.0f && raw.y <= 1.0f) ? 1.0f : 0.0f);
    f2 inv_d = div_core(f2s(1.0f), d, r1);
    f2 kn = -0.5f * inv_d * live;
    f2 kd = 0.5f * n * inv_d * inv_d * live;
    f2 dn_dmx = 2.0f * my * A2 - 2.0f * my * A1;
    f2 dd_dmx = 2.0f * mx * B2 - 2.0f * mx * B1;
    dmux = kn * dn_dmx + kd * dd_dmx;
    dexy = kn * 2.0f * A1;
    dexx = kd * B1;
}

// reprojection maps of a staged candidate pair against the staged target for the 4 pixels
// of this lane.   reference: train.py:973-985.  Channel sums accumulate in the reference's
// order ((c0 + c1) + c2).
MVF_DEV void reproj4p(const f2 *__restrict__ pair, const float *__restrict__ tgt, int off,
                      bool no_ssim, f2 out[PX])
{
    f2 ab[PX], ss[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) ab[j] = ss[j] = f2s(0.0f);
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        if (no_ssim) {
            Row6P x = load_row6p(pair + c * PPLANE + off + LDW);
            Row6 y = load_row6(tgt + c * PLANE + off + LDW);
#pragma unroll
            for (int j = 0; j < PX; ++j) ab[j] = ab[j] + pk_abs(f2s(y.v[j + 1]) - x.v[j + 1]);
        } else {
            Stats4P s;
            window_xp(pair + c * PPLANE + off, tgt + c * PLANE + off, s);
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                f2 my = div9(s.sy[j]);     // (mu_y, E[y*y])
                f2 raw = ssim_raw_pk(div9(s.sx[j]), f2s(my.x), div9(s.sxx[j]), f2s(my.y),
                                     div9(s.sxy[j]));
                ss[j] = ss[j] + clamp01_pk(raw);
                ab[j] = ab[j] + pk_abs(f2s(s.yc[j]) - s.xc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        f2 l1 = div3(ab[j]);
        if (no_ssim) {
            out[j] = l1;
        } else {
            f2 sm = div3(ss[j]);
            out[j] = 0.85f * sm + 0.15f * l1;
        }
    }
}

struct PoseLds {
    float P[MVF_MAX_SRC][12];
    float den;    // per-image mean disparity + 1e-7
    float gpix;   // backward: upstream grad / (B*H*W)
    float gloss;
    float rden;   // 1 / den (forward+backward unit kernel: phase 8)
};

MVF_DEV void load_pose_pair(const PoseLds &sh, int ka, int kb, f2 P2[12])
{
#pragma unroll
    for (int i = 0; i < 12; ++i) P2[i] = mk2(sh.P[ka][i], sh.P[kb][i]);
}

// overwrite one lane (0/1) of the 3 pair planes with a staged [3,H,W] image (reflect
// addressing); all 15 loads of a lane are in flight before the first LDS store
MVF_DEV void stage_lane3(f2 *__restrict__ pairP, int lane, const float *__restrict__ im, size_t N,
                         int H, int W, int py0, int px0)
{
    float *base = reinterpret_cast<float *>(pairP) + lane;
    float v[NSTAGE][3];
#pragma unroll
    for (int it = 0; it < NSTAGE; ++it) {
        int idx = min((int)threadIdx.x + it * NT, PH * PW - 1);
        int r = idx / PW, c = idx - r * PW;
        int gy = refl_clamp(py0 + r, H), gx = refl_clamp(px0 + c, W);
        const unsigned o = plane_off4(gy, gx, W);
        v[it][0] = ldg_at(im, o); v[it][1] = ldg_at(im + N, o); v[it][2] = ldg_at(im + 2 * N, o);
    }
#pragma unroll
    for (int it = 0; it < NSTAGE; ++it) {
        int idx = threadIdx.x + it * NT;
        int r = idx / PW, c = idx - r * PW;
        if (idx < PH * PW) {
            base[2 * (r * LDW + c)] = v[it][0];
            base[2 * (PPLANE + r * LDW + c)] = v[it][1];
            base[2 * (2 * PPLANE + r * LDW + c)] = v[it][2];
        }
    }
}

// stage two images interleaved into the pair planes (two passes of 15 loads per lane)
MVF_DEV void stage_pair3(f2 *__restrict__ pairP, const float *__restrict__ im0,
                         const float *__restrict__ im1, size_t N, int H, int W, int py0, int px0)
{
    stage_lane3(pairP, 0, im0, N, H, W, py0, px0);
    asm volatile("" ::: "memory");   // keep the passes apart: 15 live values, not 30
    stage_lane3(pairP, 1, im1, N, H, W, py0, px0);
}

// Kernel prologue: target (3 planes), disparity and a first candidate pair taken straight
// from global planes, with ALL 35 loads of a lane in flight before the first LDS store -- one
// exposed memory latency for what would otherwise be three staging phases.
MVF_DEV void stage_first(float *__restrict__ tgtP, float *__restrict__ dispP, f2 *__restrict__ pairP,
                         const float *__restrict__ tgt, const float *__restrict__ disp,
                         const float *__restrict__ im0, const float *__restrict__ im1, size_t N, int H,
                         int W, int py0, int px0, bool inner = false)
{
    float vt[NSTAGE][3], vd[NSTAGE], va[NSTAGE][3], vb[NSTAGE][3];
    unsigned off[NSTAGE];
    stage_offsets(off, H, W, py0, px0, inner);
#pragma unroll
    for (int it = 0; it < NSTAGE; ++it) {
        const unsigned o = off[it];
        vt[it][0] = ldg_at(tgt, o); vt[it][1] = ldg_at(tgt + N, o); vt[it][2] = ldg_at(tgt + 2 * N, o);
        vd[it] = ldg_at(disp, o);
        va[it][0] = ldg_at(im0, o); va[it][1] = ldg_at(im0 + N, o); va[it][2] = ldg_at(im0 + 2 * N, o);
        vb[it][0] = ldg_at(im1, o); vb[it][1] = ldg_at(im1 + N, o); vb[it][2] = ldg_at(im1 + 2 * N, o);
    }
#pragma unroll
    for (int it = 0; it < NSTAGE; ++it) {
        int idx = threadIdx.x + it * NT;
        int r = idx / PW, c = idx - r * PW;
        if (idx < PH * PW) {
            const int e = r * LDW + c;
            tgtP[e] = vt[it][0]; tgtP[PLANE + e] = vt[it][1]; tgtP[2 * PLANE + e] = vt[it][2];
            dispP[e] = vd[it];
            pairP[e] = mk2(va[it][0], vb[it][0]);
            pairP[PPLANE + e] = mk2(va[it][1], vb[it][1]);
            pairP[2 * PPLANE + e] = mk2(va[it][2], vb[it][2]);
        }
    }
}

// ---- the same prologues for INNER tiles, two pixels per load (round 4) -------------------------------------
// A plane that lies inside the image has an even origin column (px0 = bx * (TW - 2) - 2) and an even row stride,
// so a plane row is PW / 2 aligned 8-byte pairs: half the address arithmetic, half the load and LDS-store
// instructions of the element-wise form above (staging was 141 VALU instructions per output pixel and 16 % of the
// unit kernel's time: profiles/r04_unit_kernel_phase_budget.txt), and only the first wave has a second round.
constexpr int PW2 = PW / 2;
constexpr int NSTAGE2 = (PH * PW2 + NT - 1) / NT;
static_assert(PW % 2 == 0 && LDW % 2 == 0 && PLANE % 2 == 0, "8-byte pairs of a plane row");
struct StagePos2 {
    unsigned o[NSTAGE2];     // byte offset of the pair inside one [H,W] plane
    int e[NSTAGE2];          // LDS plane element of its first pixel
    bool ok[NSTAGE2];
};
MVF_DEV StagePos2 stage_pos2(int W, int py0, int px0)
{
    StagePos2 p;
    const unsigned base = plane_off4(py0, px0, W);
#pragma unroll
    for (int it = 0; it < NSTAGE2; ++it) {
        const int idx = (int)threadIdx.x + it * NT;
        p.ok[it] = idx < PH * PW2;
        const int ic = min(idx, PH * PW2 - 1);
        const int r = ic / PW2, c2 = ic - r * PW2;
        p.o[it] = base + plane_off4(r, 2 * c2, W);
        p.e[it] = r * LDW + 2 * c2;
    }
    return p;
}
// rounds after the first only exist for the leading waves (wave-uniform)
MVF_DEV bool stage2_round_live(int it) { return ((int)(threadIdx.x & ~(kWave - 1)) + it * NT) < PH * PW2; }

MVF_DEV void stage_first_inner(float *__restrict__ tgtP, float *__restrict__ dispP, f2 *__restrict__ pairP,
                               const float *__restrict__ tgt, const float *__restrict__ disp,
                               const float *__restrict__ im0, const float *__restrict__ im1, size_t N, int W, int py0,
                               int px0)
{
    const StagePos2 p = stage_pos2(W, py0, px0);
    float2 vt[NSTAGE2][3], vd[NSTAGE2], va[NSTAGE2][3], vb[NSTAGE2][3];
#pragma unroll
    for (int it = 0; it < NSTAGE2; ++it) {
        if (it > 0 && !stage2_round_live(it)) break;
        const unsigned o = p.o[it];
        vt[it][0] = ldg_f2_at(tgt, o); vt[it][1] = ldg_f2_at(tgt + N, o); vt[it][2] = ldg_f2_at(tgt + 2 * N, o);
        vd[it] = ldg_f2_at(disp, o);
        va[it][0] = ldg_f2_at(im0, o); va[it][1] = ldg_f2_at(im0 + N, o); va[it][2] = ldg_f2_at(im0 + 2 * N, o);
        vb[it][0] = ldg_f2_at(im1, o); vb[it][1] = ldg_f2_at(im1 + N, o); vb[it][2] = ldg_f2_at(im1 + 2 * N, o);
    }
#pragma unroll
    for (int it = 0; it < NSTAGE2; ++it) {
        if (it > 0 && !stage2_round_live(it)) break;
        if (!p.ok[it]) continue;
        const int e = p.e[it];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            *reinterpret_cast<float2 *>(tgtP + c * PLANE + e) = vt[it][c];
            *reinterpret_cast<float4 *>(pairP + c * PPLANE + e) =
                make_float4(va[it][c].x, vb[it][c].x, va[it][c].y, vb[it][c].y);
        }
        *reinterpret_cast<float2 *>(dispP + e) = vd[it];
    }
}
// target + disparity only (units that take their identity maps from another unit, or run without auto-masking)
MVF_DEV void stage_tgt_disp_inner(float *__restrict__ tgtP, float *__restrict__ dispP, const float *__restrict__ tgt,
                                  const float *__restrict__ disp, size_t N, int W, int py0, int px0)
{
    const StagePos2 p = stage_pos2(W, py0, px0);
    float2 vt[NSTAGE2][3], vd[NSTAGE2];
#pragma unroll
    for (int it = 0; it < NSTAGE2; ++it) {
        if (it > 0 && !stage2_round_live(it)) break;
        const unsigned o = p.o[it];
        vt[it][0] = ldg_f2_at(tgt, o); vt[it][1] = ldg_f2_at(tgt + N, o); vt[it][2] = ldg_f2_at(tgt + 2 * N, o);
        vd[it] = ldg_f2_at(disp, o);
    }
#pragma unroll
    for (int it = 0; it < NSTAGE2; ++it) {
        if (it > 0 && !stage2_round_live(it)) break;
        if (!p.ok[it]) continue;
        const int e = p.e[it];
#pragma unroll
        for (int c = 0; c < 3; ++c) *reinterpret_cast<float2 *>(tgtP + c * PLANE + e) = vt[it][c];
        *reinterpret_cast<float2 *>(dispP + e) = vd[it];
    }
}

// generate_images_pred for TWO sources at one pixel: the ray, depth and camera point are
// shared, the projection runs packed (lane 0 = source a, lane 1 = source b).
struct WarpPair {
    Tap ta, tb;
    f2 u, v, z;
    float X[3], r[3], depth;
};

MVF_DEV WarpPair warp_point_pair(float disp, const float *__restrict__ iK, const f2 P2[12], int x,
                                 int y, int H, int W, float min_disp, float range, float eps)
{
    WarpPair w;
    ray_of(iK, (float)x, (float)y, w.r);
    w.depth = depth_of(disp, min_disp, range);
    w.X[0] = w.depth * w.r[0];
    w.X[1] = w.depth * w.r[1];
    w.X[2] = w.depth * w.r[2];
    f2 c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        f2 a = P2[i * 4 + 0] * f2s(w.X[0]);
        a = pk_fma(P2[i * 4 + 1], f2s(w.X[1]), a);
        a = pk_fma(P2[i * 4 + 2], f2s(w.X[2]), a);
        a = pk_fma(P2[i * 4 + 3], f2s(1.0f), a);
        c[i] = a;
    }
    w.z = c[2] + f2s(eps);
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    f2 un, vn;
#ifndef MVF_NO_FAST_CHAIN
    // The eight correctly rounded divides c0/z, c1/z, u/(W-1), v/(H-1) of the source pair run
    // packed through the guard-free core when every operand sits in [2^-40, 2^40]: there the
    // compiler's IEEE sequence reduces to exactly these operations (mvf_common.hpp), so the
    // bits -- and the integer sampling indices -- are unchanged, the quotients by z share one
    // refined reciprocal and the reciprocals of W-1 / H-1 are wave-uniform.  Anything else
    // (zeros, huge or tiny magnitudes, denormals, inf) takes the IEEE path below; a NaN slips
    // through the min/max test and yields NaN on either path.
    const float hi = fmaxf(fmaxf(fmaxf(fabsf(c[0].x), fabsf(c[0].y)), fmaxf(fabsf(c[1].x), fabsf(c[1].y))),
                           fmaxf(fabsf(w.z.x), fabsf(w.z.y)));
    const float lo = fminf(fminf(fminf(fabsf(c[0].x), fabsf(c[0].y)), fminf(fabsf(c[1].x), fabsf(c[1].y))),
                           fminf(fabsf(w.z.x), fabsf(w.z.y)));
    if (__builtin_expect(lo >= 0x1p-40f && hi <= 0x1p40f, 1)) {
        const f2 rz = recip_refined(w.z);
        w.u = div_core(c[0], w.z, rz);
        w.v = div_core(c[1], w.z, rz);
        un = div_core(w.u, f2s(wm1), f2s(recip_refined(wm1)));
        vn = div_core(w.v, f2s(hm1), f2s(recip_refined(hm1)));
    } else
#endif
    {
        w.u = c[0] / w.z;
        w.v = c[1] / w.z;
        un = w.u / f2s(wm1);
        vn = w.v / f2s(hm1);
    }
#ifndef MVF_NO_FOLD_GRID
    // grid_sample un-normalises what Project3D normalised: ix = (((un - 0.5) * 2 + 1) / 2) * (W - 1)
    // (layers.py:219-221, then ATen's grid_sampler_unnormalize).  Scaling by 2 is exact and rounding commutes with
    // it, so RN(2 t + 1) / 2 == RN(t + 0.5) bit for bit (t = RN(un - 0.5) is at most ~0.5 in magnitude: no overflow;
    // and where t + 0.5 falls below the normal range it is an exact difference of two nearby floats): two packed
    // operations less per coordinate pair, same sampling position -- the index SHA-256 tests pin it.
    const f2 ixp = ((un - f2s(0.5f)) + f2s(0.5f)) * f2s(wm1);
    const f2 iyp = ((vn - f2s(0.5f)) + f2s(0.5f)) * f2s(hm1);
    w.ta = tap_of_ixy(ixp.x, iyp.x, H, W);
    w.tb = tap_of_ixy(ixp.y, iyp.y, H, W);
#else
    f2 gx = (un - f2s(0.5f)) * 2.0f;
    f2 gy = (vn - f2s(0.5f)) * 2.0f;
    w.ta = tap_of(gx.x, gy.x, H, W);
    w.tb = tap_of(gx.y, gy.y, H, W);
#endif
    return w;
}

struct Taps4 {
    TapRows q;
    float wnw, wne, wsw, wse;
};
MVF_DEV Taps4 taps_of(const Tap &t, int W)
{
    Taps4 r;
    r.q = taprows_of(t, W);
    float fw = t.wx, fe = 1.0f - fw, fn = t.wy, fs = 1.0f - fn;
    r.wnw = fs * fe; r.wne = fs * fw; r.wsw = fn * fe; r.wse = fn * fw;
    return r;
}

// fused warp of a source pair into the pair planes: bilinear samples of src_a / src_b at
// the projected position of every plane pixel (reflect-mapped into the image).  Two plane
// positions are processed per iteration: their projection chains (long dependent sequences
// of divides) interleave, and all 48 taps are in flight before the first use.
struct WarpSlot {
    Taps4 qa, qb;
    int r, c, x0a, y0a, x0b, y0b;
    float wxa, wya, wxb, wyb;     // fractional tap position of both sources
    unsigned fla, flb;            // bit 2: x strictly inside (gradient passes), bit 3: y strictly inside
    bool live;
};

// the warp of plane position (r, c); `inner`: the plane lies inside the image (no reflect mapping)
MVF_DEV WarpSlot warp_slot_rc(int r, int c, bool live, const float *__restrict__ dispP,
                              const float *__restrict__ iK, const f2 P2[12], int H, int W, int py0, int px0,
                              float min_disp, float range, float eps, bool inner)
{
    WarpSlot s;
    s.live = live;
    s.r = r;
    s.c = c;
    int gy, gx;
    if (inner) {        // plane inside the image (workgroup-uniform): no reflect mapping
        gy = py0 + s.r; gx = px0 + s.c;
    } else {
        gy = refl_clamp(py0 + s.r, H); gx = refl_clamp(px0 + s.c, W);
    }
#ifdef MVF_ABL_NOCHAIN
    WarpPair w = {};
#else
    WarpPair w = warp_point_pair(dispP[s.r * LDW + s.c], iK, P2, gx, gy, H, W, min_disp, range, eps);
#endif
    s.qa = taps_of(w.ta, W);
    s.qb = taps_of(w.tb, W);
#ifdef MVF_ABL_COALESCED   // ablation: taps at the pixel itself (perfectly coalesced gathers)
    s.qa.q.o0 = s.qa.q.o1 = s.qb.q.o0 = s.qb.q.o1 = ((unsigned)gy * W + min(gx, W - 2)) * 4u;
#endif
#ifdef MVF_ABL_NOCHAIN     // ablation: no projection chain (taps from the disparity bits)
    s.qa.q.o0 = s.qa.q.o1 = s.qb.q.o0 = s.qb.q.o1 = ((unsigned)gy * W + min(gx, W - 2)) * 4u;
    s.qa.wnw = s.qb.wnw = dispP[s.r * LDW + s.c];
#endif
    s.x0a = w.ta.x0; s.y0a = w.ta.y0; s.x0b = w.tb.x0; s.y0b = w.tb.y0;
    s.wxa = w.ta.wx; s.wya = w.ta.wy; s.wxb = w.tb.wx; s.wyb = w.tb.wy;
    s.fla = (w.ta.inx ? 4u : 0u) | (w.ta.iny ? 8u : 0u);
    s.flb = (w.tb.inx ? 4u : 0u) | (w.tb.iny ? 8u : 0u);
    return s;
}
// plane positions in row-major order: position idx = row * PW + col
MVF_DEV WarpSlot warp_slot(int idx, const float *__restrict__ dispP, const float *__restrict__ iK,
                           const f2 P2[12], int H, int W, int py0, int px0, float min_disp,
                           float range, float eps, bool inner = false)
{
    const bool live = idx < PH * PW;
    idx = min(idx, PH * PW - 1);
    const int r = idx / PW;
    return warp_slot_rc(r, idx - r * PW, live, dispP, iK, P2, H, W, py0, px0, min_disp, range, eps, inner);
}

template <int U>   // plane positions per iteration (1: fewest registers, 2: more overlap)
MVF_DEV void warp_pair_into_lds(f2 *__restrict__ pairP, const float *__restrict__ dispP,
                                const float *__restrict__ sa, const float *__restrict__ sb,
                                const float *__restrict__ iK, const f2 P2[12], int H, int W, int py0,
                                int px0, float min_disp, float range, float eps,
                                int32_t *__restrict__ idx_a, int32_t *__restrict__ idx_b, int ty0,
                                int tx0, int oh = TH, int ow = TW)
{
    const size_t N = (size_t)H * W;
    constexpr int NIT = (NSTAGE + U - 1) / U;
#pragma unroll 1
    for (int it = 0; it < NIT; ++it) {
        WarpSlot s[U];
        float a[U][3][4], bq[U][3][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
            s[u] = warp_slot((int)threadIdx.x + (U * it + u) * NT, dispP, iK, P2, H, W, py0, px0,
                             min_disp, range, eps);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
#ifdef MVF_ABL_LDSGATHER   // ablation: taps from an LDS plane (access-pattern cost of an LDS-staged source)
                {
                    int ya = min(max(s[u].y0a - py0, 0), PH - 2), xa = min(max(s[u].x0a - px0, 0), PW - 2);
                    int yb = min(max(s[u].y0b - py0, 0), PH - 2), xb = min(max(s[u].x0b - px0, 0), PW - 2);
                    const float *la = dispP + ya * LDW + xa, *lb = dispP + yb * LDW + xb;
                    a[u][ch][0] = la[0] + ch; a[u][ch][1] = la[1]; a[u][ch][2] = la[LDW]; a[u][ch][3] = la[LDW + 1];
                    bq[u][ch][0] = lb[0] + ch; bq[u][ch][1] = lb[1]; bq[u][ch][2] = lb[LDW]; bq[u][ch][3] = lb[LDW + 1];
                }
#else
                load_taps(sa + ch * N, s[u].qa.q, a[u][ch][0], a[u][ch][1], a[u][ch][2], a[u][ch][3]);
                load_taps(sb + ch * N, s[u].qb.q, bq[u][ch][0], bq[u][ch][1], bq[u][ch][2], bq[u][ch][3]);
#endif
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!s[u].live) continue;
            const Taps4 &qa = s[u].qa, &qb = s[u].qb;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float va = a[u][ch][0] * qa.wnw + a[u][ch][1] * qa.wne + a[u][ch][2] * qa.wsw +
                           a[u][ch][3] * qa.wse;
                float vb = bq[u][ch][0] * qb.wnw + bq[u][ch][1] * qb.wne + bq[u][ch][2] * qb.wsw +
                           bq[u][ch][3] * qb.wse;
                pairP[ch * PPLANE + s[u].r * LDW + s[u].c] = mk2(va, vb);
            }
            if (idx_a) {
                // the un-reflected pixels of this tile own their index entry
                int y = py0 + s[u].r, x = px0 + s[u].c;
                if (y >= ty0 && y < min(ty0 + oh, H) && x >= tx0 && x < min(tx0 + ow, W)) {
                    reinterpret_cast<int2 *>(idx_a)[(size_t)y * W + x] = make_int2(s[u].x0a, s[u].y0a);
                    if (idx_b != idx_a)
                        reinterpret_cast<int2 *>(idx_b)[(size_t)y * W + x] = make_int2(s[u].x0b, s[u].y0b);
                }
            }
        }
    }
}

}  // namespace
