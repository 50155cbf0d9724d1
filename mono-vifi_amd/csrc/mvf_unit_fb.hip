// mvf_unit_fb.hip -- forward AND backward of hot-path units in a single tile kernel
// (mvf_units_fwdbwd): per unit, S <= 2 x generate_images_pred (train.py:956-971) +
// compute_losses_base (train.py:987-1051) + its whole adjoint down to grad_disp and grad_T.
// A training step runs nine units (train.py:747-883); the three of each group (single-frame /
// multi-frame / affine: train.py:747-760, 795-810, 837-882) are mutually independent and go
// out as ONE launch: workgroup -> (unit, image, tile), per-unit descriptors in the kernel
// arguments, images addressed by (base, image stride) so that the interleaved output of a
// grouped decoder call is read where it lies.
//
// In training both directions of a unit always run, and everything the backward needs from the
// forward is per-pixel local -- the candidates, their min / argmin, the mask -- except two
// per-image scalars: the mean disparity (the disparity head's partials, or k_disp_mean first)
// and the smoothness sum of the mean-normalisation term, which enters grad_disp as a per-image
// constant and is applied by mvf_units_fwdbwd_scale together with the upstream gradient (the
// backward is linear in it).
//
// One workgroup (256 lanes, 2 px per lane) owns a 32x16 region and emits its 30x14 interior, so
// the 3x3 (reflect-aware) SSIM adjoint never leaves the workgroup.  Phases:
//   1  stage target, disparity and the identity pair (one memory phase)
//   2  identity candidates (SSIM + L1, packed for the pair) -- or, for the second unit of a
//      (single-frame, multi-frame) pair that shares target and sources (train.py:747-812), the
//      identity maps the first unit wrote (ident_out -> ident_in: 8 B/px instead of 24 B/px of
//      staging and the identity pair's SSIM; the tie-break noise is still per unit)
//   3  fused warp of the source pair into LDS: exact projection chain (guard-free divides)
//   4  warped candidates: ONE pass over the window statistics yields the SSIM value AND its
//      partial derivatives, which stay in registers (unweighted) until the argmin is known
//   5  min / argmin / mask / outputs
//   6  per channel: weighted coefficients -> LDS -> 3x3 adjoint gather -> grad_warped parked
//      in the channel's (consumed) pair plane
//   7  bilinear + projection adjoint per output pixel -> grad_disp, grad_P partials
//   8  smoothness (value and gradient), store, ONE reduction of all 27 tile partials
// One finishing launch per unit launch (k_units_finish: loss, stats, grad_T of every unit).
// Compiled with -ffp-contract=off (arithmetic contract in mvf_common.hpp): everything that
// feeds an integer (sampling indices, argmin) follows the reference's evaluation order.
// Region geometry (mvf_tile.hpp): 32 x 16 pixels, 2 per lane: 40 KB of LDS and 128 VGPRs, so FOUR
// workgroups share a CU (4 waves per SIMD).
// Addressing: every image base is wave-uniform (blockIdx) and held in scalar registers; pixels
// and taps are 32-bit byte offsets from it (scalar-base global loads, 24-bit multiplies) -- round
// 2's ISA spent 236 64-bit VALU address instructions, partly quarter rate, on the same accesses.
// geometry (mvf_tile.hpp): 32 x 16 region, 2 px per lane, 256 lanes; the register budget must allow 4 waves per SIMD.
// (What other geometries measured: HISTORY.md, "Why this geometry".)
#ifndef MVF_FB_PY
#define MVF_FB_PY 1      // rows per lane: 1 = 256 lanes x (1 x 2) px, 4 waves per SIMD (shipped); 2 = 128 lanes x (2 x 2) px, 2 waves
#endif                   // per SIMD with twice the registers (round 6's structural build: measured)
#define MVF_TILE_TW 32
#define MVF_TILE_PX 2
#define MVF_TILE_PY MVF_FB_PY
#define MVF_TILE_TH 16
#define MVF_FB_WAVES (MVF_FB_PY == 2 ? 2 : 4)
#include "mvf_tile.hpp"

namespace {

constexpr int OW = TW - 2, OH = TH - 2;   // output interior of a region
constexpr int NRED = 27;                  // 24 grad_P + photo + smooth x + smooth y
constexpr int NIMG = 4;                   // doubles per image of the folded loss terms (photo, sx, sy, pad)

// one unit of a launch (device copy of mvf_unit_desc, include/mvf_hotpath.h)
struct UnitArgs {
    const float *disp, *tgt, *src0, *src1, *mask, *T, *K, *invK, *noise, *mean_ws, *ident_in;
    size_t disp_stride, tgt_stride, src0_stride, src1_stride, mask_stride, g_stride;
    float *g_disp;        // [B,1,H,W] for an upstream gradient of 1, WITHOUT the per-image shift
    float *g_T;           // [S,B,4,4]
    float *loss, *stats;  // [3], [B,4]
    float *ident_out;
    uint8_t *argmin_out;
    float *auto_mask_out, *to_opt_out, *noise_out;
    int32_t *idx_xy;
    uint32_t seed0, seed1;   // in-kernel tie-break noise (noise == nullptr)
};

struct FbArgs {
    UnitArgs u[MVF_MAX_UNITS];
    float *gp_ws;         // [U][S][B][ntiles][12] grad_P partials
    float *part;          // [U][B][ntiles][NPART] loss partials
    double *img_ws;       // [U][B][NIMG] per-image folded loss terms
    int *tickets;         // [U + 1] one per unit + one per launch (k_units_finish); zero on entry, zero on exit
    float *loss_sum;      // nullable: sum of the units' loss[0]
    const float *loss_sum_in;   // nullable: running total the sum starts from
    float *tab;           // [U][B] ImgTab: what a workgroup needs of its image besides the planes (k_units_prepare)
    int nunits, flags, B, H, W, tiles_x, tiles_y;
    int flags_int;        // bit 0: every image base / stride of the launch is 8-byte aligned (pair staging allowed)
    uint32_t mg_tx, mg_ty;   // floor(2^32 / tiles_x) + 1, floor(2^32 / tiles_y) + 1 (0: one tile that way): workgroup -> tile
    float smoothness, min_disp, range, eps;
    float gpix, cxs, cys; // launch constants of the adjoint: 1 / (B N), smoothness / (B H (W-1)), smoothness / (B (H-1) W)
};

// Per-image constants of a launch, written ONCE per (unit, image) by k_units_prepare in front of the unit kernel:
// rounds 2-4 had every one of the image's 308 workgroups rebuild them -- lane 0 folding the 32 mean partials
// serially and dividing twice, 24 lanes re-multiplying K @ T -- behind a chain of five dependent memory round
// trips (kernel arguments -> unit descriptor -> partials pointer -> partials -> ...) in front of its first barrier.
struct ImgTab {
    float P[12][2];       // (K @ T)[:3], the two sources' entries side by side (S == 1: the first source's twice): a workgroup
                          // reads them as the twelve packed pairs it computes with (stored per source they cost 18 register
                          // moves per load to interleave, twice per lane)
    float mean, den, rden;   // mean disparity, mean + 1e-7, 1 / den
    float pad[5];
};
static_assert(sizeof(ImgTab) == 128, "one 128-byte line per image");
constexpr int TABF = sizeof(ImgTab) / sizeof(float);

// LDS carve (floats): target 3 planes | pair 3 f2 planes | disparity | coefficient 3 f2 region
// planes (A,B,G) | pose | scratch
constexpr int FB_TGT = 0, FB_PAIR = 3 * PLANE, FB_DISP = FB_PAIR + 6 * PLANE,
              FB_COEF = FB_DISP + PLANE, FB_POSE = FB_COEF + 6 * RPLANE;
static_assert((NT / 16) * NRED <= 6 * RPLANE, "the final reduction parks its row sums in the (then free) coefficient planes");
inline size_t fb_smem() { return FB_POSE * sizeof(float) + sizeof(ImgTab); }
MVF_DEV void load_pose_pair(const ImgTab &sh, int ka, int kb, f2 P2[12])
{
#pragma unroll
    for (int i = 0; i < 12; ++i) P2[i] = mk2(sh.P[i][ka], sh.P[i][kb]);
}
// workgroup -> tile with the launcher's reciprocals (mg = floor(2^32 / d) + 1: exact for n * d < 2^32, which the
// launcher checks; mg == 0 stands for d == 1).  Three 32-bit integer divisions on the scalar unit -- a v_rcp round trip
// through a vector register each -- were the first thing every wave of the launch did.
MVF_DEV int div_magic(int n, uint32_t mg) { return mg ? (int)__umulhi((uint32_t)n, mg) : n; }
MVF_DEV TileId tile_of_block_mg(int tiles_x, int tiles_y, int B, uint32_t mg_tx, uint32_t mg_ty)
{
    const int total = tiles_x * tiles_y * B;
    const int lin = blockIdx.x;
    const int xcd = lin & 7, slot = lin >> 3;
    const int q = total >> 3, r = total & 7;
    const int vid = xcd * q + min(xcd, r) + slot;
    TileId t;
    const int rest = div_magic(vid, mg_tx);
    t.bx = vid - rest * tiles_x;
    t.b = div_magic(rest, mg_ty);
    t.by = rest - t.b * tiles_y;
    return t;
}

// ---- counter-based tie-break noise ------------------------------------------------------
// train.py:1023-1024 draws torch.randn(identity_reprojection_loss.shape) * 1e-5 per call.  With
// noise == nullptr the kernel draws its own standard normals from a counter-based generator
// keyed by (seed, element index): two rounds of a 32-bit avalanche hash per uniform, Box-Muller
// for the pair of identity candidates of a pixel.  A tie-breaker, not a statistics engine; the
// draw can be written out (noise_out) so that a test can replay it through the oracle.
// One 32-bit avalanche hash per pixel (two xorshift-multiply rounds, the second keyed by seed1) gives
// both uniforms: 16 bits each -- 65,536 radius levels up to 4.7 sigma and 65,536 angles, ample for a
// tie-breaker scaled by 1e-5.  The transcendental steps are the hardware's own: v_log_f32 (base 2),
// v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in revolutions, so u2 goes in as it is).  Round 2 drew
// two hashes (four quarter-rate 32-bit multiplies) and went through libm's sqrtf / sincosf (correctly
// rounded square root, range reduction for arguments that never exceed 2 pi): ~50 instructions per pixel
// more for the same purpose.
MVF_DEV uint32_t mix32(uint32_t x, uint32_t key)
{
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= key;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
MVF_DEV f2 normal_pair(uint32_t seed0, uint32_t seed1, uint32_t idx)
{
    const uint32_t a = mix32(idx + seed0, seed1);
    // u1 in (0,1], u2 in [0,1)
    const float u1 = ((float)(a >> 16) + 1.0f) * 0x1p-16f, u2 = (float)(a & 0xffffu) * 0x1p-16f;
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
    return mk2(r * __builtin_amdgcn_cosf(u2), r * __builtin_amdgcn_sinf(u2));
}

// clamp(v, 0, 1) as ONE v_med3_f32 per component (the select form is four instructions).  Same value as
// torch.clamp for every non-NaN input; a NaN comes out as 0 here, and the pixel that produced it still
// poisons the loss through its own L1 term |t - p| (the SSIM map of the unit kernel is internal; the
// stand-alone mvf_ssim_fwd keeps the NaN-propagating select).
MVF_DEV f2 clamp01_med3_pk(f2 v)
{
    return mk2(__builtin_amdgcn_fmed3f(v.x, 0.0f, 1.0f), __builtin_amdgcn_fmed3f(v.y, 0.0f, 1.0f));
}

// SSIM value (exact: reference layers.py:281-290, literal order) AND its x-side partial
// derivatives (tolerance) from ONE set of window means of a candidate pair.
MVF_DEV void ssim_val_partials_pk(f2 mx, f2 my, f2 exx, f2 eyy, f2 exy, f2 &val, f2 &dmux, f2 &dexx2,
                                  f2 &dexy)
{
#ifdef MVF_FAST_SSIM     // contracted: every a * b + c of the formula is one fused multiply-add
    const f2 mxx = mx * mx, myy = my * my, mxy = mx * my;
    f2 sigma_x = exx - mxx, sigma_y = eyy - myy, sigma_xy = exy - mxy;
    f2 A1 = pk_fma(f2s(2.0f), mxy, f2s(kC1)), A2 = pk_fma(f2s(2.0f), sigma_xy, f2s(kC2));
    f2 B1 = (mxx + myy) + f2s(kC1), B2 = (sigma_x + sigma_y) + f2s(kC2);
#else
    f2 sigma_x = exx - mx * mx, sigma_y = eyy - my * my, sigma_xy = exy - mx * my;
    f2 A1 = 2.0f * mx * my + f2s(kC1), A2 = 2.0f * sigma_xy + f2s(kC2);
    f2 B1 = mx * mx + my * my + f2s(kC1), B2 = sigma_x + sigma_y + f2s(kC2);
#endif
    f2 n = A1 * A2, d = B1 * B2;
    // d >= C1*(C2 - rounding) > 0 and |n|, d = O(1) for images in [0,1]: the guard-free
    // division core gives the correctly rounded quotient (see mvf_common.hpp)
    const f2 r1 = ssim_recip(d);
    const f2 q = ssim_quot(n, d, r1);
#ifdef MVF_FAST_SSIM
    const f2 raw = pk_fma(q, f2s(-0.5f), f2s(0.5f));
#else
    const f2 raw = (f2s(1.0f) - q) / 2.0f;
#endif
    val = clamp01_med3_pk(raw);
    // the clamp passes the gradient where it changed nothing (NaN compares unequal: no gradient)
    const f2 live = mk2(val.x == raw.x ? 1.0f : 0.0f, val.y == raw.y ? 1.0f : 0.0f);
    // partial derivatives (tolerance arithmetic, constants folded): with L = live / d,
    //   d raw / d n = -L / 2,  d raw / d d = L q / 2  (q = n / d), and every partial below carries a factor 2
    const f2 L = r1 * live, Lq = L * q;
    dmux = pk_fma(Lq, mx * (B2 - B1), -(L * (my * (A2 - A1))));
    dexy = -(L * A1);
    dexx2 = Lq * B1;
}

// ---- target statistics computed ONCE per pixel and channel ------------------------------------------
// The window means of the target (mu_y, E[y*y]) are the same for the identity pair and for the
// warped pair.  The identity pass stashes them, lane-privately, in the coefficient planes (free
// until the adjoint: each lane writes and later reads only the entries of its own pixels, so no
// barrier is involved), and the warped pass reads them back instead of re-accumulating 9 taps of
// y and y*y per pixel and channel.  Exact mode is untouched: same sums in the same order.
// (sum y, sum y*y) over the 3x3 target windows of a lane's TWO pixels, packed over the PIXELS: the three
// column pairs (c0,c1), (c1,c2), (c2,c3) of a row are the (d = 0, 1, 2) taps of (pixel 0, pixel 1), so each
// half still adds its nine taps in the reference's row-major order -- exact mode untouched -- while a row
// costs 2 packed multiplies + 2 pair moves + 6 packed adds (the (y, y*y)-per-pixel form: 4 multiplies +
// 4 moves + 6 adds).  The stash then holds (mu_y px0, mu_y px1, E[yy] px0, E[yy] px1).
struct TStat2 {
    f2 sy, syy;
};
MVF_DEV void tstat_row(const Row6 &y, bool first, TStat2 &t)
{
    static_assert(PX == 2, "pixel-packed target statistics: two pixels per lane");
    const f2 y01 = mk2(y.v[0], y.v[1]), y12 = mk2(y.v[1], y.v[2]), y23 = mk2(y.v[2], y.v[3]);
    const f2 q01 = y01 * y01, q23 = y23 * y23, q12 = mk2(q01.y, q23.x);
    if (first) { t.sy = y01; t.syy = q01; }
    else { t.sy = t.sy + y01; t.syy = t.syy + q01; }
    t.sy = t.sy + y12; t.syy = t.syy + q12;
    t.sy = t.sy + y23; t.syy = t.syy + q23;
}
// (mu_y, E[y*y]) of pixel j out of the pixel-packed means
MVF_DEV f2 tstat_of(const f2 tm[PX], int j) { return j == 0 ? mk2(tm[0].x, tm[1].x) : mk2(tm[0].y, tm[1].y); }

// sign(v) in {-1, 0, +1} without compares: v * 2^127 * 2^127 is +-inf for every non-zero v (the smallest
// subnormal times 2^254 is 2^105), 0 for a zero and NaN for a NaN; v_med3_f32(., -1, 1) clamps that to +-1 / 0
// (a NaN operand yields the minimum of the others: -1 -- the compare form gave 0; a NaN difference only arises from
// a NaN image, which poisons the loss anyway).  Two full-rate multiplies + one half-rate op instead of two compares
// and two selects (tools/valu_ubench.hip: 3.7 vs 6.8 ns of issue time per value).
MVF_DEV float sign_of(float v)
{
    return __builtin_amdgcn_fmed3f((v * 0x1p127f) * 0x1p127f, -1.0f, 1.0f);
}

struct Stats4X {
    f2 sx[PX], sxx[PX], sxy[PX];
    f2 xc[PX];
    float yc[PX];
};
// Window statistics of a lane's PY x PX block of region pixels.  The PY + 2 plane rows stream through ONCE: the products
// x*x, x*y of a row are formed once and feed every output row whose window holds that row (PY = 2: 16 positions per four
// outputs instead of 24), and each output still adds its nine taps in the reference's row-major order -- its three rows
// arrive in order (exact mode untouched).  xs / ys point at plane element (first row of the block, PX * seg).
#ifdef MVF_FAST_SSIM
// Fast mode (opt-in build, never the default, never in the parity suite): what the reference's evaluation order
// (layers.py:277-290: nine taps row-major, products rounded before they are summed) costs on this chip is measured
// against THIS form -- separable 3x3 sums: the three rows of a column first (products folded in by fused
// multiply-adds), then the horizontal 3-sums of the lane's two outputs with the middle column pair shared (41 packed
// operations for the three window sums of two pixels instead of 72; the target's sums likewise).  Same quantities up to rounding (a few ulp of the
// sums): argmin / auto-mask may flip where candidates nearly tie; the integer sampling indices are untouched
// (tools/fast_mode_report.py).
MVF_DEV void window_blk(const f2 *__restrict__ xs, const float *__restrict__ ys, Stats4X (&o)[PY], TStat2 *t = nullptr)
{
    static_assert(PX == 2 && PY == 1, "fast-mode window sums: 1 x 2 pixels per lane");
    f2 cx[RW], cxx[RW], cxy[RW];
    float cy[RW], cyy[RW];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        Row6P x = load_row6p(xs + r * LDW);
        Row6 y = load_row6(ys + r * LDW);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            if (r == 0) {
                cx[i] = x.v[i]; cxx[i] = x.v[i] * x.v[i]; cxy[i] = x.v[i] * f2s(y.v[i]);
                cy[i] = y.v[i]; cyy[i] = y.v[i] * y.v[i];
            } else {
                cx[i] = cx[i] + x.v[i];
                cxx[i] = pk_fma(x.v[i], x.v[i], cxx[i]);
                cxy[i] = pk_fma(x.v[i], f2s(y.v[i]), cxy[i]);
                cy[i] = cy[i] + y.v[i];
                cyy[i] = fmaf(y.v[i], y.v[i], cyy[i]);
            }
        }
        if (r == 1) {
#pragma unroll
            for (int j = 0; j < PX; ++j) { o[0].xc[j] = x.v[j + 1]; o[0].yc[j] = y.v[j + 1]; }
        }
    }
    const f2 mx = cx[1] + cx[2], mxx = cxx[1] + cxx[2], mxy = cxy[1] + cxy[2];
    o[0].sx[0] = cx[0] + mx; o[0].sx[1] = mx + cx[3];
    o[0].sxx[0] = cxx[0] + mxx; o[0].sxx[1] = mxx + cxx[3];
    o[0].sxy[0] = cxy[0] + mxy; o[0].sxy[1] = mxy + cxy[3];
    if (t) {
        const float my = cy[1] + cy[2], myy = cyy[1] + cyy[2];
        t[0].sy = mk2(cy[0] + my, my + cy[3]);
        t[0].syy = mk2(cyy[0] + myy, myy + cyy[3]);
    }
}
MVF_DEV void target_rows(const float *__restrict__ ys, TStat2 (&t)[PY])
{
    float cy[RW], cyy[RW];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const Row6 y = load_row6(ys + r * LDW);
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            cy[i] = r == 0 ? y.v[i] : cy[i] + y.v[i];
            cyy[i] = r == 0 ? y.v[i] * y.v[i] : fmaf(y.v[i], y.v[i], cyy[i]);
        }
    }
    const float m = cy[1] + cy[2], mm = cyy[1] + cyy[2];
    t[0].sy = mk2(cy[0] + m, m + cy[3]);
    t[0].syy = mk2(cyy[0] + mm, mm + cyy[3]);
}
#else
MVF_DEV void window_blk(const f2 *__restrict__ xs, const float *__restrict__ ys, Stats4X (&o)[PY], TStat2 *t = nullptr)
{
#pragma unroll
    for (int k = 0; k < PY + 2; ++k) {
        const Row6P x = load_row6p(xs + k * LDW);
        const Row6 y = load_row6(ys + k * LDW);
        f2 xx[RW], xy[RW];
#pragma unroll
        for (int i = 0; i < RW; ++i) {
            xx[i] = x.v[i] * x.v[i];
            xy[i] = x.v[i] * f2s(y.v[i]);
        }
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            const int r = k - i;                 // which row of output row i's window this plane row is
            if (r < 0 || r > 2) continue;
            if (t) tstat_row(y, r == 0, t[i]);
#pragma unroll
            for (int j = 0; j < PX; ++j) {
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    if (r == 0 && d == 0) {
                        o[i].sx[j] = x.v[j];
                        o[i].sxx[j] = xx[j];
                        o[i].sxy[j] = xy[j];
                    } else {
                        o[i].sx[j] = o[i].sx[j] + x.v[j + d];
                        o[i].sxx[j] = o[i].sxx[j] + xx[j + d];
                        o[i].sxy[j] = o[i].sxy[j] + xy[j + d];
                    }
                }
                if (r == 1) {
                    o[i].xc[j] = x.v[j + 1];
                    o[i].yc[j] = y.v[j + 1];
                }
            }
        }
    }
}
MVF_DEV void target_rows(const float *__restrict__ ys, TStat2 (&t)[PY])
{
#pragma unroll
    for (int k = 0; k < PY + 2; ++k) {
        const Row6 y = load_row6(ys + k * LDW);
#pragma unroll
        for (int i = 0; i < PY; ++i)
            if (k - i >= 0 && k - i <= 2) tstat_row(y, k == i, t[i]);
    }
}
#endif
MVF_DEV void stash_tstats(f2 *__restrict__ statP, const f2 my[PX])
{
    float4 *p = reinterpret_cast<float4 *>(statP);
#pragma unroll
    for (int j = 0; j < PX; j += 2) p[j / 2] = make_float4(my[j].x, my[j].y, my[j + 1].x, my[j + 1].y);
}
MVF_DEV void fetch_tstats(const f2 *__restrict__ statP, f2 my[PX])
{
    const float4 *p = reinterpret_cast<const float4 *>(statP);
#pragma unroll
    for (int j = 0; j < PX; j += 2) {
        const float4 v = p[j / 2];
        my[j] = mk2(v.x, v.y);
        my[j + 1] = mk2(v.z, v.w);
    }
}
// window means of the target alone (no auto-masking, or the identity maps are handed over): stashed per row of the block
MVF_DEV void target_stats(const float *__restrict__ ys, f2 *__restrict__ statP)
{
    TStat2 t[PY];
    target_rows(ys, t);
#pragma unroll
    for (int i = 0; i < PY; ++i) {
        f2 my[PX];
        my[0] = div9(t[i].sy);        // (mu_y px0, mu_y px1)
        my[1] = div9(t[i].syy);       // (E[yy] px0, E[yy] px1)
        stash_tstats(statP + i * LDW, my);
    }
}
// identity candidates (reference: train.py:973-985 on the raw sources), stashing the target means
MVF_DEV void reproj_identity(const f2 *__restrict__ pair, const float *__restrict__ tgt, int off,
                             bool no_ssim, f2 (&out)[PY][PX], f2 *__restrict__ statP, int roff)
{
    f2 ab[PY][PX], ss[PY][PX];
#pragma unroll
    for (int i = 0; i < PY; ++i)
#pragma unroll
        for (int j = 0; j < PX; ++j) ab[i][j] = ss[i][j] = f2s(0.0f);
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        if (no_ssim) {
#pragma unroll
            for (int i = 0; i < PY; ++i) {
                Row6P x = load_row6p(pair + c * PPLANE + off + (i + 1) * LDW);
                Row6 y = load_row6(tgt + c * PLANE + off + (i + 1) * LDW);
#pragma unroll
                for (int j = 0; j < PX; ++j) ab[i][j] = ab[i][j] + pk_abs(f2s(y.v[j + 1]) - x.v[j + 1]);
            }
        } else {
            Stats4X s[PY];
            TStat2 t[PY];
            window_blk(pair + c * PPLANE + off, tgt + c * PLANE + off, s, t);
#pragma unroll
            for (int i = 0; i < PY; ++i) {
                f2 my[PX];
                my[0] = div9(t[i].sy);        // (mu_y px0, mu_y px1)
                my[1] = div9(t[i].syy);       // (E[yy] px0, E[yy] px1)
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    const f2 m = tstat_of(my, j);
                    f2 raw = ssim_raw_pk(div9(s[i].sx[j]), f2s(m.x), div9(s[i].sxx[j]), f2s(m.y), div9(s[i].sxy[j]));
                    ss[i][j] = ss[i][j] + clamp01_med3_pk(raw);
                    ab[i][j] = ab[i][j] + pk_abs(f2s(s[i].yc[j]) - s[i].xc[j]);
                }
                stash_tstats(statP + c * RPPLANE + roff + i * LDW, my);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < PY; ++i)
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            f2 l1 = div3(ab[i][j]);
            out[i][j] = no_ssim ? l1 : 0.85f * div3(ss[i][j]) + 0.15f * l1;
        }
}

// What phase 7 needs of the forward's projection chain at an output pixel, kept in SIX registers per
// position: per source the byte offset of the row-y0 tap pair (< 2^28) with four flags above it (pair
// anchored one pixel left, row y1 below y0, x / y strictly inside) and the fractional tap position (two floats).
// For this to work a lane must warp in phase 3 the pixels whose adjoint it evaluates in phases 7 + 8: the warp
// walks the 30 x 14 interior first, in the lane order of phase 7, then the border ring of the 34 x 18 plane.
struct TapStash {
    uint32_t oa, ob;
    float wxa, wya, wxb, wyb;
};
constexpr uint32_t kOffMask = 0x0fffffffu;
MVF_DEV uint32_t pack_w(float wx, float wy)
{
    return (uint32_t)(wx * 65536.0f) | ((uint32_t)(wy * 65536.0f) << 16);      // wx, wy in [0, 1): truncation
}
MVF_DEV TapStash pack_taps(const WarpSlot &s)
{
    TapStash t;
    t.oa = s.qa.q.o0 | (s.qa.q.sh ? 1u << 28 : 0u) | (s.qa.q.o1 != s.qa.q.o0 ? 1u << 29 : 0u) | (s.fla << 28);
    t.ob = s.qb.q.o0 | (s.qb.q.sh ? 1u << 28 : 0u) | (s.qb.q.o1 != s.qb.q.o0 ? 1u << 29 : 0u) | (s.flb << 28);
    t.wxa = s.wxa; t.wya = s.wya; t.wxb = s.wxb; t.wyb = s.wyb;
    return t;
}
// plane position (r, c) of warp slot q (0 .. NSTAGE-1) of this lane; false beyond the plane
MVF_DEV bool fb_slot_pos(int q, int &r, int &c)
{
    // The interior ROWS first, all TW region columns of each (plane columns 1 .. TW): the order phase 7 walks them in.
    // Round 6: rows of TW = 32 lanes instead of the OW = 30 output columns -- a 30-wide walk wraps inside every
    // 32-lane LDS access group and the two wrapped lanes land LDW - OW = 6 banks further, on banks the group already
    // uses (2-way conflict on every ds_read_b32 of phases 7 + 8: the smoothness reads alone were 23 % of the kernel's
    // SQ_LDS_BANK_CONFLICT, profiles/r06_unit_kernel_lds_conflicts.csv).  Two lanes per row idle in phase 7 (their
    // positions are ring columns, warped here like any other); the round count is the same (448 positions: 2 rounds).
    constexpr int NI = TW * (TH - 2);
    const int s = (int)threadIdx.x + q * NT;
    if (s < NI) {
        const int pr = s / TW;
        r = pr + 2; c = s - pr * TW + 1;
        return true;
    }
    const int j = s - NI;                                // the ring: rows 0, 1, PH-2, PH-1, then the two side columns
    if (j < 4 * PW) {
        const int rr = j / PW;
        r = (rr < 2) ? rr : rr + (PH - 4); c = j - rr * PW;
        return true;
    }
    const int jj = min(j - 4 * PW, 2 * (PH - 4) - 1), rr = jj >> 1;
    r = rr + 2; c = (jj & 1) ? PW - 1 : 0;
    return j - 4 * PW < 2 * (PH - 4);
}

struct WarpCtx {
    f2 *pairP;
    const float *dispP, *sa, *sb, *iK;
    const f2 *P2;
    int H, W, py0, px0, oy0, ox0;
    float min_disp, range, eps;
    int32_t *idx_a, *idx_b;
    bool inner;        // the staged plane lies inside the image: no reflect mapping (workgroup-uniform)
    TapStash *stash;   // [2] registers of the lane: taps of its (up to) two interior positions
};

// one batch of U plane positions: chain + tap loads issued (issue), bilinear combine + store (finish).
// The tap rows stay the 8-byte pairs they were loaded as: (west, east) x (west weight, east weight)
// is ONE packed multiply per row, and the right-border case (pair anchored one pixel left, both taps
// its second element, east weight 0 there) becomes a swap of the WEIGHT pair -- once per source and
// position instead of two selects per channel.  Same products, same left-to-right sum:
// nw*wnw + ne*wne + sw*wsw + se*wse, with exact zeros where a weight is 0.
template <int U>
struct WarpBatch {
    WarpSlot s[U];
    float2 a0[U][3], a1[U][3], b0[U][3], b1[U][3];     // rows y0 / y1 of source a / b
};


template <int U>
MVF_DEV void warp_issue(const WarpCtx &k, int slot0, WarpBatch<U> &w)
{
    const size_t N = (size_t)k.H * k.W;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int r, c;
        const bool live = fb_slot_pos(slot0 + u, r, c);
        w.s[u] = warp_slot_rc(r, c, live, k.dispP, k.iK, k.P2, k.H, k.W, k.py0, k.px0, k.min_disp, k.range, k.eps,
                              k.inner);
        if (slot0 + u < (TW * (TH - 2) + NT - 1) / NT) k.stash[slot0 + u] = pack_taps(w.s[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            w.a0[u][ch] = ldg2_at(k.sa + ch * N, w.s[u].qa.q.o0);
            w.a1[u][ch] = ldg2_at(k.sa + ch * N, w.s[u].qa.q.o1);
            w.b0[u][ch] = ldg2_at(k.sb + ch * N, w.s[u].qb.q.o0);
            w.b1[u][ch] = ldg2_at(k.sb + ch * N, w.s[u].qb.q.o1);
        }
}

// (west, east) weight pairs of the two tap rows; at the right border the pair sits one pixel left
// of x0 and the tap is its second element (whose partner weight is exactly 0 there: wx == 0)
MVF_DEV void row_weights(const Taps4 &q, f2 &top, f2 &bot)
{
    top = q.q.sh ? mk2(0.0f, q.wnw) : mk2(q.wnw, q.wne);
    bot = q.q.sh ? mk2(0.0f, q.wsw) : mk2(q.wsw, q.wse);
}

template <int U>
MVF_DEV void warp_finish(const WarpCtx &k, const WarpBatch<U> &w)
{
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!w.s[u].live) continue;
        f2 wat, wab, wbt, wbb;
        row_weights(w.s[u].qa, wat, wab);
        row_weights(w.s[u].qb, wbt, wbb);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float va = w.a0[u][ch].x * wat.x + w.a0[u][ch].y * wat.y + w.a1[u][ch].x * wab.x + w.a1[u][ch].y * wab.y;
            const float vb = w.b0[u][ch].x * wbt.x + w.b0[u][ch].y * wbt.y + w.b1[u][ch].x * wbb.x + w.b1[u][ch].y * wbb.y;
            k.pairP[ch * PPLANE + w.s[u].r * LDW + w.s[u].c] = mk2(va, vb);
        }
        if (k.idx_a) {
            // the region's 62x14 interior owns its entries of the (optional) index maps
            int y = k.py0 + w.s[u].r, x = k.px0 + w.s[u].c;
            if (y >= k.oy0 && y < min(k.oy0 + OH, k.H) && x >= k.ox0 && x < min(k.ox0 + OW, k.W)) {
                reinterpret_cast<int2 *>(k.idx_a)[(size_t)y * k.W + x] = make_int2(w.s[u].x0a, w.s[u].y0a);
                if (k.idx_b != k.idx_a)
                    reinterpret_cast<int2 *>(k.idx_b)[(size_t)y * k.W + x] = make_int2(w.s[u].x0b, w.s[u].y0b);
            }
        }
    }
}

template <int U>
MVF_DEV void warp_slots(const WarpCtx &k, int slot0)
{
    WarpBatch<U> w;
    warp_issue<U>(k, slot0, w);
    warp_finish<U>(k, w);
}

// plane positions first .. NSTAGE-1, ONE per batch (24 tap pairs in flight; two per batch cost a workgroup per CU:
// 128 VGPRs + spills, +4 %, HISTORY.md); waves whose lanes all lie beyond the plane at a position skip it (wave-uniform)
MVF_DEV void warp_pair_into_lds_fb(const WarpCtx &k, int first)
{
    const int wave_base = (int)threadIdx.x & ~(kWave - 1);
#pragma unroll
    for (int q = first; q < NSTAGE; ++q)
        if ((wave_base + q * NT) < PH * PW) warp_slots<1>(k, q);
}

// ---- static analysis build (tools/isa_cost.py; never shipped) -----------------------------------------------
// -DMVF_PHASE_MARKERS puts a comment line into the ISA at every phase boundary; -DMVF_ANALYSIS=k freezes the
// run-time switches to one hot configuration so that the compiler drops the branches that configuration never
// takes and the static instruction stream IS the executed one (phases are straight-line code, loops unrolled):
//   1 = single-frame launch (auto-masking, identity candidates evaluated, in-kernel noise, no mask), inner tile
//   2 = multi-frame launch (identity maps handed over), inner tile      3 = affine launch (as 1 + mask_rec)
#ifdef MVF_PHASE_MARKERS
#define MVF_PHASE(name) asm volatile("; MVF_PHASE " name ::: "memory")
#else
#define MVF_PHASE(name)
#endif
#ifndef MVF_ANALYSIS
#define MVF_ANALYSIS 0
#endif

// =============================================================================== the kernel
template <int S, bool AVG>     // AVG: --avg_reprojection (both sources carry gradient)
__global__ void __launch_bounds__(NT, MVF_FB_WAVES) k_unit_fb(FbArgs a)
{
    static_assert(S == 1 || S == 2, "one source pair");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tgtP = smem + FB_TGT;
    f2 *pairP = reinterpret_cast<f2 *>(smem + FB_PAIR);
    float *dispP = smem + FB_DISP;
    f2 *coefP = reinterpret_cast<f2 *>(smem + FB_COEF);
    ImgTab &sh = *reinterpret_cast<ImgTab *>(smem + FB_POSE);

    // workgroup -> (unit, image, tile): the images of all units form one batch of nunits * B
    const TileId tid = tile_of_block_mg(a.tiles_x, a.tiles_y, a.B * a.nunits, a.mg_tx, a.mg_ty);
    const int H = a.H, W = a.W;
    const int unit = __builtin_amdgcn_readfirstlane(tid.b / a.B);
    const int b = __builtin_amdgcn_readfirstlane(tid.b - unit * a.B);
    const int ub = unit * a.B + b;          // image slot of this launch
    const UnitArgs &u = a.u[unit];
    const size_t N = (size_t)H * W;
    const int cy0 = tid.by * OH - 1, cx0 = tid.bx * OW - 1;   // region origin
    const int py0 = cy0 - 1, px0 = cx0 - 1;                   // plane origin
    // (MVF_ANALYSIS: compile-time constants -- the branches the frozen configuration never takes fold away)
    const bool no_ssim = MVF_ANALYSIS ? false : bool(a.flags & MVF_NO_SSIM);
    const bool automask = MVF_ANALYSIS ? true : !(a.flags & MVF_NO_AUTOMASK);
    constexpr bool avg = AVG;
    const int n_id = automask ? (avg ? 1 : S) : 0;
    constexpr bool hasb = S > 1;
    constexpr int kb = hasb ? 1 : 0;

    // image bases in scalar registers
    const float *tgt_b = uniform_ptr(u.tgt + (size_t)b * u.tgt_stride);
    const float *disp_b = uniform_ptr(u.disp + (size_t)b * u.disp_stride);
    const float *sa = uniform_ptr(u.src0 + (size_t)b * u.src0_stride);
    const float *sb = hasb ? uniform_ptr(u.src1 + (size_t)b * u.src1_stride) : sa;
    const float *iK = u.invK + b * 16;
    const bool ident_given = MVF_ANALYSIS ? (MVF_ANALYSIS == 2) : (automask && (u.ident_in != nullptr));

    // the image's constants: one 128-byte line of the table k_units_prepare wrote, in flight with the staging loads
    if (threadIdx.x < TABF)
        reinterpret_cast<float *>(&sh)[threadIdx.x] = a.tab[(size_t)ub * TABF + threadIdx.x];
    // a lane owns a PY x PX block of region pixels: rows row0 .. row0 + PY - 1, columns seg * PX ..
    const int seg = threadIdx.x & (TW / PX - 1), row0 = (threadIdx.x / (TW / PX)) * PY;
    const int off = row0 * LDW + seg * PX;    // plane element of the window's top-left (of the block's first row)
    const int roff = off;                     // region-plane element of this lane's first pixel
    const int y0 = cy0 + row0, x0 = cx0 + seg * PX;

    // ---- 1: target, disparity (and the identity pair) -> LDS
    // tiles whose staged plane lies inside the image need no reflect / clamp mapping of the
    // coordinates they stage and warp (scalar branch; 240 of 308 tiles at 640x192)
    const bool inner = MVF_ANALYSIS ? true : ((py0 >= 0) && (px0 >= 0) && (py0 + PH <= H) && (px0 + PW <= W));
    MVF_PHASE("1_stage");
#ifndef MVF_ABL_NOSTAGE   // timing ablation: nothing staged (the planes keep what the previous workgroup left there)
    // (the pair form needs 8-byte aligned image bases: checked by the launcher, which clears FB_PAIR_OK otherwise)
    const bool pairs_ok = inner && (a.flags_int & 1);
    if (automask && !ident_given) {
        if (pairs_ok) stage_first_inner(tgtP, dispP, pairP, tgt_b, disp_b, sa, sb, N, W, py0, px0);
        else stage_first(tgtP, dispP, pairP, tgt_b, disp_b, sa, sb, N, H, W, py0, px0, inner);
    } else {
        if (pairs_ok) stage_tgt_disp_inner(tgtP, dispP, tgt_b, disp_b, N, W, py0, px0);
        else {
            stage_planes3(tgtP, tgt_b, N, H, W, py0, px0, inner);
            stage_plane(dispP, disp_b, H, W, py0, px0, inner);
        }
    }
#endif
    __syncthreads();

    f2 P2[12];
    load_pose_pair(sh, 0, kb, P2);
    WarpCtx wk_ctx;
    {
        WarpCtx &k = wk_ctx;
        k.pairP = pairP; k.dispP = dispP;
        k.sa = sa; k.sb = sb;
        k.iK = iK; k.P2 = P2;
        k.H = H; k.W = W; k.py0 = py0; k.px0 = px0; k.oy0 = cy0 + 1; k.ox0 = cx0 + 1;
        k.min_disp = a.min_disp; k.range = a.range; k.eps = a.eps;
        k.idx_a = (!MVF_ANALYSIS && u.idx_xy) ? u.idx_xy + ((size_t)b) * N * 2 : nullptr;
        k.idx_b = (!MVF_ANALYSIS && u.idx_xy) ? u.idx_xy + ((size_t)kb * a.B + b) * N * 2 : nullptr;
        k.inner = inner;
    }
    constexpr int NSTASH = (TW * (TH - 2) + NT - 1) / NT;     // adjoint positions per lane (phase 7 walks TW x OH)
    TapStash stash[NSTASH] = {};
    wk_ctx.stash = stash;

    // ---- 2: identity candidates of every region pixel
    MVF_PHASE("2_identity");
    f2 vid[PY][PX];
#pragma unroll
    for (int i = 0; i < PY; ++i)
#pragma unroll
        for (int j = 0; j < PX; ++j) vid[i][j] = f2s(0.0f);
    if (automask && !ident_given) {
#ifndef MVF_ABL_NOID     // timing ablation: identity candidates not evaluated
        reproj_identity(pairP, tgtP, off, no_ssim, vid, coefP, roff);
#else
        for (int i = 0; i < PY; ++i)
            for (int j = 0; j < PX; ++j) vid[i][j] = pairP[off + (i + 1) * LDW + 1 + j];
#endif
        __syncthreads();                       // identity pair consumed
    } else {
        if (ident_given) {
            // the identity maps of the unit this one shares target and sources with (pre-noise;
            // exactly the values reproj_identity would produce here): one 8-byte load per pixel,
            // in flight while the target statistics below are evaluated
            const float *idb = uniform_ptr(u.ident_in + (size_t)b * N * 2);
#pragma unroll
            for (int i = 0; i < PY; ++i)
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    const int yc = min(max(y0 + i, 0), H - 1), xc = min(max(x0 + j, 0), W - 1);
                    const float2 v = ldg_f2_at(idb, plane_off4(yc, xc, W) * 2u);
                    vid[i][j] = mk2(v.x, v.y);
                }
        }
        if (!no_ssim) {
#pragma unroll 1
            for (int c = 0; c < 3; ++c) target_stats(tgtP + c * PLANE + off, coefP + c * RPPLANE + roff);
        }
    }

    // ---- 3: fused warp of the source pair
    MVF_PHASE("3_warp");
#ifdef MVF_ABL_NOWARP    // timing ablation: the raw sources instead of the warped pair
    stage_pair3(pairP, wk_ctx.sa, wk_ctx.sb, N, H, W, py0, px0);
    if (false)
#endif
        warp_pair_into_lds_fb(wk_ctx, 0);
    __syncthreads();

    // ---- 4: warped candidates; the SSIM partials of the three channels stay in registers
    // (channel loops unrolled: round 2 rolled them and rotated the partials through three slots,
    // 24 register moves per channel here and again in phase 6)
    MVF_PHASE("4_ssim_warped");
    f2 pm[3][PY][PX], px2[3][PY][PX], pg[3][PY][PX];      // d/d mu_x, 2 d/d E[xx], d/d E[xy]
    f2 vw[PY][PX];
    {
        f2 ab[PY][PX], ss[PY][PX];
#pragma unroll
        for (int i = 0; i < PY; ++i)
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                ab[i][j] = ss[i][j] = f2s(0.0f);
#pragma unroll
                for (int q = 0; q < 3; ++q) pm[q][i][j] = px2[q][i][j] = pg[q][i][j] = f2s(0.0f);
            }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#ifdef MVF_ABL_NOSSIM4
            if (true) {
#else
            if (no_ssim) {
#endif
#pragma unroll
                for (int i = 0; i < PY; ++i) {
                    Row6P x = load_row6p(pairP + c * PPLANE + off + (i + 1) * LDW);
                    Row6 yv = load_row6(tgtP + c * PLANE + off + (i + 1) * LDW);
#pragma unroll
                    for (int j = 0; j < PX; ++j) ab[i][j] = ab[i][j] + pk_abs(f2s(yv.v[j + 1]) - x.v[j + 1]);
                }
            } else {
                Stats4X st[PY];
                window_blk(pairP + c * PPLANE + off, tgtP + c * PLANE + off, st);
#pragma unroll
                for (int i = 0; i < PY; ++i) {
                    f2 tm[PX];                 // (mu_y, E[y*y]) stashed by the identity / target pass
                    fetch_tstats(coefP + c * RPPLANE + roff + i * LDW, tm);
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
                        const f2 my = tstat_of(tm, j);
                        f2 val;
                        ssim_val_partials_pk(div9(st[i].sx[j]), f2s(my.x), div9(st[i].sxx[j]), f2s(my.y),
                                             div9(st[i].sxy[j]), val, pm[c][i][j], px2[c][i][j], pg[c][i][j]);
                        ss[i][j] = ss[i][j] + val;
                        ab[i][j] = ab[i][j] + pk_abs(f2s(st[i].yc[j]) - st[i].xc[j]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < PY; ++i)
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                f2 l1 = div3(ab[i][j]);
                vw[i][j] = no_ssim ? l1 : 0.85f * div3(ss[i][j]) + 0.15f * l1;
            }
    }

    // mask and tie-break noise of this lane's region pixels.  Fetched / drawn here rather than in
    // the prologue: six registers less across the warp and SSIM phases (128-VGPR budget of 4
    // waves per SIMD); with 4 workgroups per CU the one exposed load latency is covered
    MVF_PHASE("5a_mask_noise");
    float mraw[PY][PX];        // mask value (1 without a mask), 0 outside the image
    f2 nz[PY][PX];
    const float *mask_b = (MVF_ANALYSIS ? (MVF_ANALYSIS == 3) : (u.mask != nullptr)) ? uniform_ptr(u.mask + (size_t)b * u.mask_stride) : nullptr;
#pragma unroll
    for (int i = 0; i < PY; ++i)
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int x = x0 + j, y = y0 + i;
        const bool in = (y >= 0) && (y < H) && (x >= 0) && (x < W);
        const unsigned pix4 = plane_off4(min(max(y, 0), H - 1), min(max(x, 0), W - 1), W);
        mraw[i][j] = in ? (mask_b ? ldg_at(mask_b, pix4) : 1.0f) : 0.0f;
        nz[i][j] = f2s(0.0f);
#ifdef MVF_ABL_NONOISE   // timing ablation: no tie-break noise
        if (false) {
#else
        if (automask && in) {
#endif
            if (!MVF_ANALYSIS && u.noise) {
                const float *nb = uniform_ptr(u.noise + (size_t)b * (avg ? 1 : S) * N);
                if (avg) nz[i][j] = f2s(ldg_at(nb, pix4));
                else nz[i][j] = mk2(ldg_at(nb, pix4), hasb ? ldg_at(nb + N, pix4) : 0.0f);
            } else {
                nz[i][j] = normal_pair(u.seed0, u.seed1, (uint32_t)b * (uint32_t)N + (pix4 >> 2));
                if (!MVF_ANALYSIS && u.noise_out) {
                    float *nb = uniform_ptr(u.noise_out + (size_t)b * (avg ? 1 : S) * N);
                    stg_at(nb, pix4, nz[i][j].x);
                    if (!avg && hasb) stg_at(nb + N, pix4, nz[i][j].y);
                }
            }
        }
    }

    // ---- 5: min / argmin / mask / outputs (reference: train.py:1010-1043)
    MVF_PHASE("5b_argmin_outputs");
    f2 wk[PY][PX];             // adjoint weight of the two warped candidates
    float fb_photo = 0.0f;
#ifdef MVF_ABL_NO5       // timing ablation: no min / argmin / mask / outputs (every candidate value stays live)
#pragma unroll
    for (int i = 0; i < PY; ++i)
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        asm volatile("" :: "v"(vw[i][j]), "v"(vid[i][j]), "v"(nz[i][j]));
        wk[i][j] = f2s(a.gpix * mraw[i][j]);
    }
    if (false)
#endif
#pragma unroll
    for (int i = 0; i < PY; ++i)
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int x = x0 + j, col = seg * PX + j, y = y0 + i, row = row0 + i;
        const bool rowin = (y >= 0) && (y < H);
        const bool row_out = (row >= 1) && (row <= OH) && rowin;   // interior (= output) rows
        const bool in = rowin && (x >= 0) && (x < W);
        const unsigned pix4 = plane_off4(min(max(y, 0), H - 1), min(max(x, 0), W - 1), W);
        float best = 0.0f;
        int bi = 0, nc = 0;
        if (automask) {
            if (avg) {
                float m = vid[i][j].x;
                if (hasb) m = m + vid[i][j].y;
                m = m / (float)S;
                best = m + nz[i][j].x * 0.00001f;
                nc = 1;
            } else {
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    float v = (k == 0 ? vid[i][j].x : vid[i][j].y) + (k == 0 ? nz[i][j].x : nz[i][j].y) * 0.00001f;
                    if (nc == 0 || v < best) { best = v; bi = nc; }
                    ++nc;
                }
            }
        }
        if (avg) {
            float m = vw[i][j].x;
            if (hasb) m = m + vw[i][j].y;
            m = m / (float)S;
            if (nc == 0 || m < best) { best = m; bi = nc; }
            ++nc;
        } else {
#pragma unroll
            for (int k = 0; k < S; ++k) {
                float v = (k == 0) ? vw[i][j].x : vw[i][j].y;
                if (nc == 0 || v < best) { best = v; bi = nc; }
                ++nc;
            }
        }
        if (mask_b) best = best * mraw[i][j];
        const int sel = (nc > 1) ? bi : 255;
        const bool outp = row_out && (col >= 1) && (col <= OW) && in;
        if (outp) {
            if (u.argmin_out) stg_u8_at(uniform_ptr(u.argmin_out + (size_t)b * N), pix4 >> 2, (uint8_t)sel);
            if (!MVF_ANALYSIS && u.auto_mask_out) stg_at(uniform_ptr(u.auto_mask_out + (size_t)b * N), pix4, (bi > n_id - 1) ? 1.0f : 0.0f);
            if (!MVF_ANALYSIS && u.to_opt_out) stg_at(uniform_ptr(u.to_opt_out + (size_t)b * N), pix4, best);
            if ((MVF_ANALYSIS ? MVF_ANALYSIS == 1 : (u.ident_out != nullptr)) && automask)
                stg_f2_at(uniform_ptr(u.ident_out + (size_t)b * N * 2), pix4 * 2u, make_float2(vid[i][j].x, vid[i][j].y));
            fb_photo += best;
        }
        // selection weight of the two sources: the argmin picked it (or the averaged channel)
        float wa, wb;
        if (sel == 255) wa = wb = avg ? 1.0f / (float)S : 1.0f;        // single candidate
        else if (avg) wa = wb = (sel == n_id) ? 1.0f / (float)S : 0.0f;
        else { wa = (sel == n_id) ? 1.0f : 0.0f; wb = (sel == n_id + 1) ? 1.0f : 0.0f; }
        const float wbase = a.gpix * mraw[i][j];                          // 0 outside the image
        wk[i][j] = mk2(wbase * wa, hasb ? wbase * wb : 0.0f);
    }

    // ---- 6: SSIM adjoint, one channel at a time through the coefficient planes.
    // g_x(q) = sum over the 3x3 windows p around q of A(p) + x(q) B(p) + y(q) G(p).  The
    // horizontal 3-sums are formed in registers before the LDS round trip: a lane owns PX
    // consecutive columns, the two missing neighbours come from lanes seg-1 / seg+1 of the same
    // 16-lane row by DPP row shifts (zero-filled at the row ends: those columns do not exist
    // and would only reach the region's border columns, which are never outputs).  The planes then
    // hold row sums, and the vertical step reads two rows instead of nine row segments.
    // Reflect-pad multiplicities only exist next to the image border (rows 1, H-2, cols 1, W-2).
    MVF_PHASE("6_ssim_adjoint");
    float mlx[PX], mrx[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        mlx[j] = (x0 + j == 1) ? 2.0f : 1.0f;
        mrx[j] = (x0 + j == W - 2) ? 2.0f : 1.0f;
    }
    const bool col_border = !MVF_ANALYSIS && ((cx0 <= 1) || (cx0 + TW >= W - 1));          // workgroup-uniform
    // DPP row shifts inside the 16-lane rows (= the 16 column segments of a region row).  The empty
    // asm pins each 32-bit move: without it hipcc 7.2 folds the two halves of a pair into ONE move
    // and broadcasts it (observed in the ISA: v_mov_b32_dpp + op_sel_hi:[0,1]).
    auto dpp1 = [](float v, bool right) {
        int r = right ? __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, true)
                      : __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true);
        asm volatile("" : "+v"(r));
        return __builtin_bit_cast(float, r);
    };
    auto from_left = [&](f2 v) { return mk2(dpp1(v.x, false), dpp1(v.y, false)); };    // lane seg-1 (0 at seg 0)
    auto from_right = [&](f2 v) { return mk2(dpp1(v.x, true), dpp1(v.y, true)); };     // lane seg+1 (0 at seg 15)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        f2 hs[3][PY][PX];          // row sums of A, B, G at this lane's columns, per row of its block
#ifdef MVF_ABL_NO6H      // timing ablation: no SSIM adjoint at all (coefficients, DPP row sums, LDS round trip, gather)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < PY; ++i)
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    asm volatile("" :: "v"(pm[c][i][j]), "v"(px2[c][i][j]), "v"(pg[c][i][j]));
                    hs[pl][i][j] = f2s(0.0f);
                }
        if (false) {
#else
        if (!no_ssim) {
#endif
            if (c > 0) __syncthreads();        // vertical reads of the previous channel done
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int i = 0; i < PY; ++i) {
                    f2 cf[PX + 2];
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
                        const f2 g = wk[i][j] * ((0.85f / 3.0f) / 9.0f);
                        cf[j + 1] = g * (pl == 0 ? pm[c][i][j] : (pl == 1 ? px2[c][i][j] : pg[c][i][j]));
                    }
                    cf[0] = from_left(cf[PX]);
                    cf[PX + 1] = from_right(cf[1]);
                    if (col_border) {
#pragma unroll
                        for (int j = 0; j < PX; ++j)
                            hs[pl][i][j] = pk_fma(f2s(mrx[j]), cf[j + 2], pk_fma(f2s(mlx[j]), cf[j], cf[j + 1]));
                    } else {
#pragma unroll
                        for (int j = 0; j < PX; ++j) hs[pl][i][j] = (cf[j] + cf[j + 1]) + cf[j + 2];
                    }
                    // (with PY > 1 only the block's first and last row are read by other lanes, but every row is some
                    // block's first or last for PY <= 2)
                    float4 *cp = reinterpret_cast<float4 *>(coefP + pl * RPPLANE + roff + i * LDW);
#pragma unroll
                    for (int j = 0; j < PX; j += 2)
                        cp[j / 2] = make_float4(hs[pl][i][j].x, hs[pl][i][j].y, hs[pl][i][j + 1].x, hs[pl][i][j + 1].y);
                }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < PY; ++i) {
            const int row = row0 + i, y = y0 + i;
            const float myu = (y == 1) ? 2.0f : 1.0f, myd = (y == H - 2) ? 2.0f : 1.0f;
            // own centre values (the pair plane of this channel is only read by its owner from now on)
            f2 xq[PX], gw[PX];
            float yq[PX];
            {
                const f2 *xc = pairP + c * PPLANE + (row + 1) * LDW + seg * PX + 1;
                const float *yc = tgtP + c * PLANE + (row + 1) * LDW + seg * PX + 1;
#pragma unroll
                for (int j = 0; j < PX; ++j) { xq[j] = xc[j]; yq[j] = yc[j]; }
            }
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                // L1 term: d|t-p|/dp = -sign(t-p), channel mean
                f2 df = f2s(yq[j]) - xq[j];
                f2 sg = mk2(sign_of(df.x), sign_of(df.y));      // d|t - p| / dp = -sign(t - p): the minus sits in the constant
                gw[j] = wk[i][j] * (-(no_ssim ? 1.0f : 0.15f) * (1.0f / 3.0f)) * sg;
            }
#if defined(MVF_ABL_NOGATHER) || defined(MVF_ABL_NO6H)
            if (false) {
#else
            if (!no_ssim && row >= 1 && row <= OH) {
#endif
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    // the rows above and below: the block's own (registers) or another lane's (LDS)
                    f2 uu[PX], dd[PX];
                    if (i == 0) {
                        const float4 *up = reinterpret_cast<const float4 *>(coefP + pl * RPPLANE + roff - LDW);
#pragma unroll
                        for (int j = 0; j < PX; j += 2) {
                            const float4 uv = up[j / 2];
                            uu[j] = mk2(uv.x, uv.y); uu[j + 1] = mk2(uv.z, uv.w);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < PX; ++j) uu[j] = hs[pl][i > 0 ? i - 1 : 0][j];
                    }
                    if (i == PY - 1) {
                        const float4 *dn = reinterpret_cast<const float4 *>(coefP + pl * RPPLANE + roff + PY * LDW);
#pragma unroll
                        for (int j = 0; j < PX; j += 2) {
                            const float4 d = dn[j / 2];
                            dd[j] = mk2(d.x, d.y); dd[j + 1] = mk2(d.z, d.w);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < PX; ++j) dd[j] = hs[pl][i < PY - 1 ? i + 1 : i][j];
                    }
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
                        const f2 t = pk_fma(f2s(myd), dd[j], pk_fma(f2s(myu), uu[j], hs[pl][i][j]));     // adjoint: tolerance arithmetic
                        if (pl == 0) gw[j] += t;
                        else if (pl == 1) gw[j] = pk_fma(xq[j], t, gw[j]);
                        else gw[j] = pk_fma(f2s(yq[j]), t, gw[j]);
                    }
                }
            }
            // park grad_warped of this channel in its own pair plane (own entries only)
            {
                f2 *gp = pairP + c * PPLANE + (row + 1) * LDW + seg * PX + 1;
#pragma unroll
                for (int j = 0; j < PX; ++j) gp[j] = gw[j];
            }
        }
    }
    __syncthreads();      // every grad_warped is parked

    // ---- 7 + 8: bilinear + projection adjoint, smoothness value + gradient, store grad_disp.
    // Lanes now walk the region linearly (position p = tid + k*256, TW per row): neighbouring
    // lanes handle neighbouring pixels, so the bilinear taps of a wave fall into a few cache
    // lines (with the owner mapping a wave's taps were PX pixels apart per lane).
    MVF_PHASE("7_8_adjoint_smooth");
    load_pose_pair(sh, 0, kb, P2);
    float *gd_b = uniform_ptr(u.g_disp + (size_t)b * u.g_stride);
    f2 accP[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        accP[q] = f2s(0.0f);
        // one definition, here: without the pin the compiler re-creates these zeros at every level of the
        // position loop's condition nest (36 register moves per wave in the first pass alone)
        asm volatile("" : "+v"(accP[q]));
    }
    float fb_sx = 0.0f, fb_sy = 0.0f;
    const float rden = sh.rden;
    const float cxs = a.cxs, cys = a.cys;
    // (the OH interior rows are enumerated, TW lanes each: 448 positions = 7 wave passes, same as the 420 of an OW-wide
    // walk; see fb_slot_pos for why the rows are TW wide)
    constexpr int NPOS7 = TW * OH;
#pragma unroll
    for (int k = 0; k < (NPOS7 + NT - 1) / NT; ++k) {
        const int p = (int)threadIdx.x + k * NT;
        if ((NPOS7 % NT) && ((int)(threadIdx.x & ~(kWave - 1)) + k * NT >= NPOS7)) break;   // whole wave beyond: wave-uniform
        const int r = 1 + p / TW, cc = p - (p / TW) * TW;
        const int yy = cy0 + r, xx = cx0 + cc;
        const bool outp = (p < NPOS7) && (r >= 1) && (r <= OH) && (cc >= 1) && (cc <= OW) && (yy < H) && (xx < W);
        if (!outp) continue;       // yy, xx >= 0 for interior positions
        const int e = (r + 1) * LDW + cc + 1;
        float gdp = 0.0f;
#ifdef MVF_ABL_NO7      // timing ablation: no bilinear / projection adjoint
        gdp = pairP[e].x;
#else
        const f2 g0 = pairP[e], g1 = pairP[PPLANE + e], g2 = pairP[2 * PPLANE + e];
        // grad_warped of a source is non-zero wherever ANY window of the pixel's 3x3 neighbourhood
        // selected it, so both sources are usually live; pixels deep inside auto-masked areas
        // (every neighbour won by an identity candidate) fetch no taps at all
        const bool live = (g0.x != 0.0f) || (g1.x != 0.0f) || (g2.x != 0.0f) ||
                          (hasb && ((g0.y != 0.0f) || (g1.y != 0.0f) || (g2.y != 0.0f)));
        if (live) {
            // the forward's taps come out of the registers phase 3 left them in (same lane, same position);
            // what the projection adjoint needs besides -- the camera point, z, u, v -- is tolerance
            // arithmetic: one v_rcp each for the depth (one Newton step) and for z
            WarpPair w;
            {
                const TapStash ts = stash[k];
                ray_of(iK, (float)xx, (float)yy, w.r);
                const float scaled = a.min_disp + a.range * dispP[e];
                float rd = __builtin_amdgcn_rcpf(scaled);
                rd = fmaf(fmaf(-scaled, rd, 1.0f), rd, rd);
                w.depth = rd;
                w.X[0] = rd * w.r[0]; w.X[1] = rd * w.r[1]; w.X[2] = rd * w.r[2];
                f2 c[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    f2 acc = P2[i * 4 + 0] * f2s(w.X[0]);
                    acc = pk_fma(P2[i * 4 + 1], f2s(w.X[1]), acc);
                    acc = pk_fma(P2[i * 4 + 2], f2s(w.X[2]), acc);
                    c[i] = acc + P2[i * 4 + 3];
                }
                w.z = c[2] + f2s(a.eps);
                const f2 rzz = mk2(__builtin_amdgcn_rcpf(w.z.x), __builtin_amdgcn_rcpf(w.z.y));
                w.u = c[0] * rzz;
                w.v = c[1] * rzz;
                w.ta.wx = ts.wxa; w.ta.wy = ts.wya;
                w.tb.wx = ts.wxb; w.tb.wy = ts.wyb;
                w.ta.inx = ts.oa & (1u << 30); w.ta.iny = ts.oa & (1u << 31);
                w.tb.inx = ts.ob & (1u << 30); w.tb.iny = ts.ob & (1u << 31);
            }
            const unsigned W4 = (unsigned)W * 4u;
            TapRows qa, qb;
            qa.o0 = stash[k].oa & kOffMask; qa.o1 = qa.o0 + ((stash[k].oa & (1u << 29)) ? W4 : 0u);
            qa.sh = stash[k].oa & (1u << 28);
            qb.o0 = stash[k].ob & kOffMask; qb.o1 = qb.o0 + ((stash[k].ob & (1u << 29)) ? W4 : 0u);
            qb.sh = stash[k].ob & (1u << 28);
            float dxa[3], dya[3], dxb[3], dyb[3];
            {
                // the tap rows as the 8-byte pairs they are loaded as: d/dy of the bilinear sample is
                // (row1 - row0) . (e, w) -- one packed subtract and one packed multiply per channel; at
                // the right border (pair anchored one pixel left, both taps its second element, w == 0)
                // the weight pair is swapped instead of selecting taps.  d/dx there is masked by `inx`.
                float2 ra0[3], ra1[3], rb0[3], rb1[3];
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    ra0[ch] = ldg2_at(sa + ch * N, qa.o0); ra1[ch] = ldg2_at(sa + ch * N, qa.o1);
                    rb0[ch] = ldg2_at(sb + ch * N, qb.o0); rb1[ch] = ldg2_at(sb + ch * N, qb.o1);
                }
                const float na = w.ta.wy, sna = 1.0f - na, nb = w.tb.wy, snb = 1.0f - nb;
                const f2 ewa = qa.sh ? mk2(w.ta.wx, 1.0f - w.ta.wx) : mk2(1.0f - w.ta.wx, w.ta.wx);
                const f2 ewb = qb.sh ? mk2(w.tb.wx, 1.0f - w.tb.wx) : mk2(1.0f - w.tb.wx, w.tb.wx);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    // (the adjoint is tolerance arithmetic: explicit fused multiply-adds from here on; the
                    // translation unit's -ffp-contract=off only guards the forward's exact-mode expressions)
                    dxa[ch] = fmaf(ra1[ch].y - ra1[ch].x, na, (ra0[ch].y - ra0[ch].x) * sna);
                    dxb[ch] = fmaf(rb1[ch].y - rb1[ch].x, nb, (rb0[ch].y - rb0[ch].x) * snb);
                    dya[ch] = fmaf(ra1[ch].y - ra0[ch].y, ewa.y, (ra1[ch].x - ra0[ch].x) * ewa.x);
                    dyb[ch] = fmaf(rb1[ch].y - rb0[ch].y, ewb.y, (rb1[ch].x - rb0[ch].x) * ewb.x);
                }
            }
            const f2 gix = pk_fma(g2, mk2(dxa[2], dxb[2]), pk_fma(g1, mk2(dxa[1], dxb[1]), g0 * mk2(dxa[0], dxb[0])));
            const f2 giy = pk_fma(g2, mk2(dya[2], dyb[2]), pk_fma(g1, mk2(dya[1], dyb[1]), g0 * mk2(dya[0], dyb[0])));
            // adjoint of unnormalise / normalise ((W-1)/2 * 2/(W-1) = 1) and of the perspective
            // divide; tolerance arithmetic: one reciprocal of z per source (see warp_point_bwd)
            const f2 gu = mk2(w.ta.inx ? gix.x : 0.0f, w.tb.inx ? gix.y : 0.0f);
            const f2 gv = mk2(w.ta.iny ? giy.x : 0.0f, w.tb.iny ? giy.y : 0.0f);
            const f2 rz = mk2(__builtin_amdgcn_rcpf(w.z.x), __builtin_amdgcn_rcpf(w.z.y));
            f2 gc[3];
            gc[0] = gu * rz;
            gc[1] = gv * rz;
            gc[2] = -pk_fma(gc[1], w.v, gc[0] * w.u);
            f2 gd = f2s(0.0f);
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const f2 gX = pk_fma(gc[2], P2[2 * 4 + jj], pk_fma(gc[1], P2[1 * 4 + jj], gc[0] * P2[0 * 4 + jj]));
                gd = pk_fma(gX, f2s(w.r[jj]), gd);
            }
            gdp = -(hasb ? gd.x + gd.y : gd.x) * w.depth * w.depth * a.range;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                accP[q * 4 + 0] = pk_fma(gc[q], f2s(w.X[0]), accP[q * 4 + 0]);
                accP[q * 4 + 1] = pk_fma(gc[q], f2s(w.X[1]), accP[q * 4 + 1]);
                accP[q * 4 + 2] = pk_fma(gc[q], f2s(w.X[2]), accP[q * 4 + 2]);
                accP[q * 4 + 3] += gc[q];
            }
        }
#endif
        // smoothness: d/d disp_j of s*smooth(disp/den) = gn_j/den - (per-image constant); the
        // constant needs the per-image smoothness sum and is applied by mvf_units_fwdbwd_scale.
        const float *dc = dispP + e;
        const float *t0 = tgtP + e;
        auto wgt = [&](int o) {
            // (the smoothness term is tolerance arithmetic -- its exp already is: the channel mean as a multiply)
            float gi = ((fabsf(t0[0] - t0[o]) + fabsf(t0[PLANE] - t0[PLANE + o])) +
                        fabsf(t0[2 * PLANE] - t0[2 * PLANE + o])) * (1.0f / 3.0f);
            return __expf(-gi);
        };
        auto sgn = [](float v) { return sign_of(v); };
        // only the SIGN of differences of normalised disparities is needed for the gradient:
        // dividing by the positive per-image constant cannot change it
        const float nd = dc[0];
        float gn = 0.0f;
#ifndef MVF_ABL_NOSMOOTH
        if (xx + 1 < W) {
            const float w1 = wgt(1), df = nd - dc[1];
            gn = fmaf(cxs * w1, sgn(df), gn);
            fb_sx = fmaf(fabsf(df) * rden, w1, fb_sx);
        }
        if (xx - 1 >= 0) gn = fmaf(-cxs * wgt(-1), sgn(dc[-1] - nd), gn);
        if (yy + 1 < H) {
            const float wl = wgt(LDW), df = nd - dc[LDW];
            gn = fmaf(cys * wl, sgn(df), gn);
            fb_sy = fmaf(fabsf(df) * rden, wl, fb_sy);
        }
        if (yy - 1 >= 0) gn = fmaf(-cys * wgt(-LDW), sgn(dc[-LDW] - nd), gn);
#endif
#ifdef MVF_ABL_NOSTORE    // timing ablation: grad_disp not stored
        { const float gout = fmaf(gn, rden, gdp); asm volatile("" :: "v"(gout)); }
#else
        stg_at(gd_b, plane_off4(yy, xx, W), fmaf(gn, rden, gdp));
#endif
    }

    // ---- one reduction for all tile partials: grad_P of both sources, photo, smoothness sums
    MVF_PHASE("9_reduce");
    const size_t ntiles = (size_t)a.tiles_x * a.tiles_y;
    const size_t tile = (size_t)tid.by * a.tiles_x + tid.bx;
    const size_t UB = (size_t)a.nunits * a.B;
    {
        float flat[NRED];
#pragma unroll
        for (int q = 0; q < 12; ++q) { flat[q] = accP[q].x; flat[12 + q] = accP[q].y; }
        flat[24] = fb_photo; flat[25] = fb_sx; flat[26] = fb_sy;
#ifdef MVF_ABL_NORED     // timing ablation: no workgroup reduction (the partials stay live)
        float tot = 0.0f;
#pragma unroll
        for (int q = 0; q < NRED; ++q) asm volatile("" :: "v"(flat[q]));
        tot = (threadIdx.x & 1) ? flat[1] : flat[0];
#else
        // every plane is dead by now: the transpose uses the workgroup's LDS from its start (the pose block
        // behind the planes stays untouched)
        static_assert(NRED * (NT + 8) + NRED * 8 <= FB_POSE, "the reduction's transpose fits in front of the pose block");
        const float tot = block_sum_many_lds<NT, NRED>(flat, smem);
#endif
        const int t = threadIdx.x;
        // gp_ws [S][U*B][ntiles][12], part [U*B][ntiles][NPART]; folded by k_units_finish
        if (t < 12) a.gp_ws[(((size_t)ub) * ntiles + tile) * 12 + t] = tot;
        else if (t < 24) { if (hasb) a.gp_ws[((UB + ub) * ntiles + tile) * 12 + t - 12] = tot; }
        else if (t < NRED) a.part[((size_t)ub * ntiles + tile) * NPART + (t - 24)] = tot;
    }
}

// ---- preparing kernel: ONE small launch in front of the unit kernel, blocks (chunk, image, unit).
// For a unit whose disparity head supplied no mean partials (`mean_mask` bit set) the NMEAN blocks of an image first
// sum their chunk of the disparity (k_disp_mean's partition and order) and the LAST of them to arrive (device-scope
// ticket) writes the image's table line; for the others only block 0 of an image runs, straight to the table.
// Table line: lanes 0 .. 23 form (K @ T)[:3] (proj_entry: ATen's small-matrix order, golden key "P*"); lane 32 folds
// the NMEAN partials in index order -- whoever arrives last: the result does not depend on the arrival order -- and
// divides (the compiler's correctly rounded IEEE divides, as the unit kernel's lane 0 did before round 5).
__global__ void __launch_bounds__(256) k_units_prepare(FbArgs a, int S, unsigned mean_mask, int *tk)
{
    __shared__ float scratch[4];
    __shared__ int s_last;
    const int chunk = blockIdx.x, b = blockIdx.y, unit = blockIdx.z;
    const UnitArgs &u = a.u[unit];
    const int N = a.H * a.W;
    const int t = threadIdx.x;
    if ((mean_mask >> unit) & 1u) {
        const int per = (N + NMEAN - 1) / NMEAN;
        const int lo = chunk * per, hi = min(lo + per, N);
        const float *d = u.disp + (size_t)b * u.disp_stride;
        // k_disp_mean's sums, bit for bit (four strided accumulators over a lane's full groups of four, the up to
        // three elements left over added to the first), with sixteen loads in flight per lane instead of four: the
        // rolled loop paid a memory round trip per group (17 us for the six disparity tensors of a launch)
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        const int nt = (hi - lo - t + 255) / 256;        // this lane's elements (<= 0: none)
        const int n4 = nt & ~3;                          // ... of which in full groups of four
        for (int k0 = 0; k0 < nt; k0 += 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = (k0 + k < nt) ? d[lo + t + (k0 + k) * 256] : 0.0f;
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                const bool full = k0 + k < n4;           // (a group starts at a multiple of four: all of it or none)
                s0 += v[k];
                s1 += full ? v[k + 1] : 0.0f;
                s2 += full ? v[k + 2] : 0.0f;
                s3 += full ? v[k + 3] : 0.0f;
                if (!full) { s0 += v[k + 1]; s0 += v[k + 2]; s0 += v[k + 3]; }
            }
        }
        const float r = block_sum<256>((s0 + s1) + (s2 + s3), scratch);
        if (t == 0) {
            publish(const_cast<float *>(u.mean_ws) + b * NMEAN + chunk, r);
            s_last = (take_ticket(tk + unit * a.B + b) == NMEAN - 1);
        }
        __syncthreads();
        if (!s_last) return;
        if (t == 0) publish(tk + unit * a.B + b, 0);    // leave the counter as it was found
    } else if (chunk != 0) return;
    ImgTab &tb = reinterpret_cast<ImgTab *>(a.tab)[(size_t)unit * a.B + b];
    // the NMEAN partials fetched by NMEAN lanes at once, then folded in index order by one (a lane fetching them one after
    // the other paid 32 dependent round trips to the device-coherent level: 16 us of the launch's 17.5)
    __shared__ float parts[NMEAN];
    if (t >= 64 && t < 64 + NMEAN) parts[t - 64] = fetch_published(u.mean_ws + b * NMEAN + (t - 64));
    __syncthreads();
    if (t < 24) {
        const int k = (t < 12 || S < 2) ? 0 : 1, e = t < 12 ? t : t - 12;
        tb.P[e][t < 12 ? 0 : 1] = proj_entry(u.K + b * 16, u.T + ((size_t)k * a.B + b) * 16, e >> 2, e & 3);
    } else if (t == 32) {
        float m = 0.0f;
        for (int i = 0; i < NMEAN; ++i) m += parts[i];
        const float mean = m / (float)((size_t)a.H * a.W);
        const float den = mean + 1e-7f;
        tb.mean = mean;
        tb.den = den;
        tb.rden = 1.0f / den;             // reciprocal of the mean-normalisation constant (phase 8)
    }
}

// ---- finishing kernel: ONE launch for all units of a unit launch ------------------------------
// block (unit, image): folds the image's tile partials (27 values x ntiles; tiles strided over 32
// slices in fp64, then the slices in order) into grad_T of both sources (= K^T [grad_P ; 0]),
// stats[b] and the image's loss terms; the block of a unit that finishes LAST (device-scope
// ticket) folds the images in index order into loss[3].  Every fold reads its inputs in index
// order whoever performs it: the results do not depend on the arrival order.
// (Round 2: one launch PER UNIT whose B*S+1 blocks each walked all tiles serially, 10.2 us.
//  Folding inside the unit kernel by its last-arriving workgroup was built and measured this
//  round: with release/acquire fences 545 us per unit instead of 102 -- buffer_wbl2 / buffer_inv
//  act on the whole L2 --, with device-scope sc1 accesses instead of fences 107: every workgroup
//  then waits for a write-through and a ticket round trip while it holds 40 KB of LDS.)
constexpr int FIN_SLICES = 32;          // tile slices per (unit, image) block: 32 values x 32 slices = 1024 lanes
constexpr int FIN_UNROLL = 4;           // independent loads in flight per lane
__global__ void __launch_bounds__(32 * FIN_SLICES) k_units_finish(FbArgs a, int S)
{
    __shared__ double red[FIN_SLICES * 32];
    __shared__ int s_last;
    const int ub = blockIdx.x, unit = ub / a.B, b = ub - unit * a.B;
    const UnitArgs &u = a.u[unit];
    const int H = a.H, W = a.W;
    const size_t N = (size_t)H * W;
    const size_t ntiles = (size_t)a.tiles_x * a.tiles_y;
    const size_t UB = (size_t)a.nunits * a.B;
    const int q = threadIdx.x & 31, slice = threadIdx.x >> 5;
    double acc = 0.0;
    if (q < NRED) {
        const float *srcp;
        size_t stride;
        if (q < 12) { srcp = a.gp_ws + ((size_t)ub * ntiles) * 12 + q; stride = 12; }
        else if (q < 24) { srcp = a.gp_ws + ((UB + ub) * ntiles) * 12 + (q - 12); stride = 12; }
        else { srcp = a.part + ((size_t)ub * ntiles) * NPART + (q - 24); stride = NPART; }
        if (q < 12 || q >= 24 || S > 1) {
            // a lane's tiles slice, slice + 32, ...: FIN_UNROLL loads issued before the first add (the
            // round-2 form walked 39 tiles per lane one exposed memory latency at a time: 17 us)
            for (size_t t0 = slice; t0 < ntiles; t0 += (size_t)FIN_SLICES * FIN_UNROLL) {
                float v[FIN_UNROLL];
#pragma unroll
                for (int k = 0; k < FIN_UNROLL; ++k) {
                    const size_t t = t0 + (size_t)k * FIN_SLICES;
                    v[k] = (t < ntiles) ? srcp[t * stride] : 0.0f;
                }
#pragma unroll
                for (int k = 0; k < FIN_UNROLL; ++k) acc += (double)v[k];
            }
        }
    }
    red[slice * 32 + q] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        double v = 0.0;
#pragma unroll
        for (int sl = 0; sl < FIN_SLICES; ++sl) v += red[sl * 32 + q];
        red[q] = v;          // slice 0's row now holds the totals (each lane wrote only its own q)
    }
    __syncthreads();
    if (threadIdx.x < 16 * S) {
        const int s = threadIdx.x >> 4, e16 = threadIdx.x & 15, kk = e16 >> 2, j = e16 & 3;
        const float *Kb = u.K + b * 16;
        double v = 0.0;
        for (int qq = 0; qq < 3; ++qq) v += (double)Kb[qq * 4 + kk] * red[s * 12 + qq * 4 + j];
        u.g_T[((size_t)s * a.B + b) * 16 + e16] = (float)v;
    }
    if (threadIdx.x == 0) {
        const double sxb = red[25] / ((double)a.B * H * (W - 1)), syb = red[26] / ((double)a.B * (H - 1) * W);
        const ImgTab &tb = reinterpret_cast<const ImgTab *>(a.tab)[ub];
        u.stats[b * 4 + 0] = tb.mean;
        u.stats[b * 4 + 1] = tb.den;
        u.stats[b * 4 + 2] = (float)sxb;
        u.stats[b * 4 + 3] = (float)syb;
        double *iw = a.img_ws + (size_t)ub * NIMG;
        publish(iw, red[24]);
        publish(iw + 1, sxb + syb);
        s_last = (take_ticket(a.tickets + unit) == a.B - 1);       // the image's terms are complete before its ticket
    }
    __syncthreads();          // (also: everybody is done with `red`)
    if (!s_last) return;      // workgroup-uniform
    // the unit's last block folds its images in index order.  The published terms are fetched by as many lanes at once and
    // summed from LDS by one (lane 0 fetching them one after the other: 2 B dependent round trips, 10 us per launch)
    double photo = 0.0, smooth = 0.0;
    for (int i0 = 0; i0 < a.B; i0 += 512) {
        const int which = threadIdx.x >> 9, ii = i0 + (threadIdx.x & 511);
        if (ii < a.B) red[threadIdx.x] = fetch_published(a.img_ws + ((size_t)unit * a.B + ii) * NIMG + which);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nb = min(512, a.B - i0);
            for (int k = 0; k < nb; ++k) {
                photo += red[k];
                smooth += red[512 + k];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double pmn = photo / ((double)a.B * (double)N);
        const float l0 = (float)(pmn + (double)a.smoothness * smooth);
        u.loss[0] = l0;
        u.loss[1] = (float)pmn;
        u.loss[2] = (float)smooth;
        publish(a.tickets + unit, 0);      // leave the counter as it was found
        s_last = 0;
        if (a.loss_sum) {
            // the launch's last unit to finish adds the units' losses in UNIT order (whoever it is: the
            // result does not depend on the arrival order); each finisher publishes its loss[0] first
            publish(u.loss, l0);
            s_last = (take_ticket(a.tickets + a.nunits) == a.nunits - 1);
        }
    }
    __syncthreads();
    if (!s_last) return;
    float *lsum = reinterpret_cast<float *>(red);
    if ((int)threadIdx.x < a.nunits) lsum[threadIdx.x] = fetch_published(a.u[threadIdx.x].loss);
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = a.loss_sum_in ? a.loss_sum_in[0] : 0.0f;
        for (int i = 0; i < a.nunits; ++i) tot += lsum[i];
        a.loss_sum[0] = tot;
        publish(a.tickets + a.nunits, 0);
    }
}

// backward(): out = (raw - shift_b) * g_loss for grad_disp, g_T_raw * g_loss for grad_T.
// shift_b = (smoothness * smooth_b / N) / den_b is the mean-normalisation term of the smoothness
// gradient; (x - s) * g in this order reproduces the bits of the two-kernel path for g = 1.
struct ScaleUnit {
    const float *g_raw, *gT_raw, *stats, *g_loss, *g_sum;
    float *g_disp, *gT;
    size_t in_stride, out_stride;
};
struct ScaleArgs {
    ScaleUnit u[MVF_MAX_UNITS];
    float smoothness;
    int N4, N, nT;
};
__global__ void __launch_bounds__(256) k_fb_scale(ScaleArgs a)
{
    const ScaleUnit &u = a.u[blockIdx.z];
    const float g = (u.g_loss ? u.g_loss[0] : 0.0f) + (u.g_sum ? u.g_sum[0] : 0.0f);
    if (blockIdx.y == gridDim.y - 1) {            // the extra row of blocks scales grad_T
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < a.nT) u.gT[i] = u.gT_raw[i] * g;
        return;
    }
    if (!u.g_disp) return;                        // the unit's disparity gradient is taken raw by its consumer
    const int b = blockIdx.y;
    const int N = a.N, N4 = a.N4;
    const float den = u.stats[b * 4 + 1];
    const float smooth_b = u.stats[b * 4 + 2] + u.stats[b * 4 + 3];
    const float shift = (a.smoothness * smooth_b / (float)N) / den;
    const float *in = u.g_raw + (size_t)b * u.in_stride;
    float *out = u.g_disp + (size_t)b * u.out_stride;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < N4) {
        const float4 v = reinterpret_cast<const float4 *>(in)[i];
        reinterpret_cast<float4 *>(out)[i] =
            make_float4((v.x - shift) * g, (v.y - shift) * g, (v.z - shift) * g, (v.w - shift) * g);
    } else {
        for (int k = 4 * i; k < N && k < 4 * i + 4; ++k)
            if (k >= 4 * N4) out[k] = (in[k] - shift) * g;
    }
}

}  // namespace

namespace {
// floats of workspace: mean partials | loss partials | grad_P partials | per-image folds (doubles)
struct WsLayout {
    size_t mean, part, gp, img, tab, total;
};
WsLayout ws_layout(int U, int B, int H, int W)
{
    const size_t tiles = (size_t)((W + OW - 1) / OW) * ((H + OH - 1) / OH);
    WsLayout l;
    l.mean = 0;
    l.part = l.mean + (size_t)U * B * NMEAN;
    l.gp = l.part + (size_t)U * B * tiles * NPART;
    l.img = l.gp + (size_t)2 * U * B * tiles * 12;
    l.img = (l.img + 1) & ~(size_t)1;                   // doubles: 8-byte aligned
    l.tab = l.img + (size_t)U * B * NIMG * 2;
    l.tab = (l.tab + 31) & ~(size_t)31;                 // the per-image table: 128-byte lines (workspace itself: see the launcher)
    l.total = l.tab + (size_t)U * B * TABF;
    return l;
}
}  // namespace

extern "C" {

size_t mvf_units_workspace_floats(int n_units, int B, int H, int W)
{
    return ws_layout(n_units, B, H, W).total + 2;
}


// one per unit + one per launch (finishing kernel), then one per (unit, image) (preparing kernel)
size_t mvf_units_ticket_ints(int n_units, int B) { return (size_t)n_units + 1 + (size_t)n_units * (B > 0 ? B : 0); }

int mvf_units_fwdbwd(const mvf_unit_desc *units, int n_units, int S, int flags, float smoothness,
                     float min_disp, float range, float eps, float *workspace, int32_t *tickets, int B,
                     int H, int W, void *stream)
{
    if (n_units < 1 || n_units > MVF_MAX_UNITS || !units) return (int)hipErrorInvalidValue;
    if (S < 1 || S > 2) return (int)hipErrorInvalidValue;      // one source pair
    if (!workspace || !tickets) return (int)hipErrorInvalidValue;
    if (B * H * W <= 0) return 0;
    if ((double)H * W * 4.0 >= 268435456.0 || H >= (1 << 22) || W >= (1 << 22))
        return (int)hipErrorInvalidValue;     // byte offsets inside a plane fit 28 bits (4 flag bits ride above them)
    if ((((uintptr_t)workspace) & 7) != 0) return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int N = H * W;
    const bool automask = !(flags & MVF_NO_AUTOMASK);
    FbArgs a = {};
    a.nunits = n_units; a.flags = flags; a.B = B; a.H = H; a.W = W;
    a.tiles_x = (W + OW - 1) / OW; a.tiles_y = (H + OH - 1) / OH;
    a.smoothness = smoothness; a.min_disp = min_disp; a.range = range; a.eps = eps;
    const size_t ntiles = (size_t)a.tiles_x * a.tiles_y;
    const WsLayout l = ws_layout(n_units, B, H, W);
    a.part = workspace + l.part;
    a.gp_ws = workspace + l.gp;
    a.img_ws = reinterpret_cast<double *>(workspace + l.img);
    a.tab = workspace + l.tab;
    a.tickets = tickets;
    {
        // launch constants the adjoint used to derive per lane (two double multiplies and an IEEE divide each): the
        // same expressions, evaluated here once -- host float / double arithmetic is IEEE, the bits are the same
        a.gpix = 1.0f / (float)((double)B * (double)N);
        a.cxs = smoothness / (float)((double)B * H * (W - 1));
        a.cys = smoothness / (float)((double)B * (H - 1) * W);
        const uint64_t nmax = (uint64_t)ntiles * B * n_units;          // largest dividend of the tile decode
        if (nmax * (uint64_t)max(a.tiles_x, a.tiles_y) >= (1ull << 32)) return (int)hipErrorInvalidValue;
        a.mg_tx = a.tiles_x > 1 ? (uint32_t)((1ull << 32) / a.tiles_x) + 1u : 0u;
        a.mg_ty = a.tiles_y > 1 ? (uint32_t)((1ull << 32) / a.tiles_y) + 1u : 0u;
    }
    a.loss_sum = units[0].loss_sum;
    a.loss_sum_in = units[0].loss_sum ? units[0].loss_sum_in : nullptr;
    // units without disparity-mean partials of their own: the preparing launch computes them
    unsigned mean_mask = 0;
    bool pair_ok = (W % 2) == 0;
    for (int i = 0; i < n_units; ++i) {
        const mvf_unit_desc &d = units[i];
        if (!d.disp || !d.tgt || !d.src[0] || (S > 1 && !d.src[1]) || !d.T || !d.K || !d.inv_K ||
            !d.g_disp_raw || !d.g_T_raw || !d.loss || !d.stats)
            return (int)hipErrorInvalidValue;
        UnitArgs &u = a.u[i];
        u.disp = d.disp; u.tgt = d.tgt; u.src0 = d.src[0]; u.src1 = S > 1 ? d.src[1] : d.src[0];
        u.mask = d.mask_rec; u.T = d.T; u.K = d.K; u.invK = d.inv_K; u.noise = d.noise;
        u.ident_in = automask ? d.ident_in : nullptr; u.ident_out = automask ? d.ident_out : nullptr;
        u.disp_stride = d.disp_stride ? (size_t)d.disp_stride : (size_t)N;
        u.tgt_stride = d.tgt_stride ? (size_t)d.tgt_stride : (size_t)3 * N;
        u.src0_stride = d.src_stride[0] ? (size_t)d.src_stride[0] : (size_t)3 * N;
        u.src1_stride = S > 1 ? (d.src_stride[1] ? (size_t)d.src_stride[1] : (size_t)3 * N) : u.src0_stride;
        u.mask_stride = d.mask_stride ? (size_t)d.mask_stride : (size_t)N;
        u.g_stride = d.g_stride ? (size_t)d.g_stride : (size_t)N;
        pair_ok = pair_ok && ((((uintptr_t)u.disp | (uintptr_t)u.tgt | (uintptr_t)u.src0 | (uintptr_t)u.src1) & 7) == 0) &&
                  (((u.disp_stride | u.tgt_stride | u.src0_stride | u.src1_stride) & 1) == 0) && ((N & 1) == 0);
        u.g_disp = d.g_disp_raw; u.g_T = d.g_T_raw; u.loss = d.loss; u.stats = d.stats;
        u.argmin_out = d.argmin; u.auto_mask_out = d.auto_mask; u.to_opt_out = d.to_opt;
        u.noise_out = d.noise_out; u.idx_xy = d.idx_xy;
        u.seed0 = (uint32_t)d.noise_seed; u.seed1 = (uint32_t)(d.noise_seed >> 32);
        if (d.disp_mean_partials) u.mean_ws = d.disp_mean_partials;
        else {
            u.mean_ws = workspace + l.mean + (size_t)i * B * NMEAN;
            mean_mask |= 1u << i;
        }
    }
    a.flags_int = pair_ok ? 1 : 0;
    {
        // (the disparity planes are read once when a unit brings no mean partials)
        int nm = 0;
        for (int i = 0; i < n_units; ++i) nm += (mean_mask >> i) & 1;
        ProfScope ps(MVF_PROF_DISP_MEAN, st, 4LL * nm * B * N + (int64_t)sizeof(ImgTab) * n_units * B);
        hipLaunchKernelGGL(k_units_prepare, dim3(mean_mask ? NMEAN : 1, (unsigned)B, (unsigned)n_units), dim3(256), 0, st,
                           a, S, mean_mask, tickets + n_units + 1);
    }
    {
        // launch kind for the per-type medians of bench.py: identity maps taken over / mask supplied / neither
        int tag = MVF_TAG_SINGLE_FRAME, nmask = 0;
        for (int i = 0; i < n_units; ++i) {
            if (a.u[i].ident_in) tag = MVF_TAG_MULTI_FRAME;
            nmask += a.u[i].mask != nullptr;
        }
        if (tag != MVF_TAG_MULTI_FRAME && nmask) tag = (nmask == n_units) ? MVF_TAG_AFFINE : MVF_TAG_MIXED;
        ProfScope ps(MVF_PROF_UNIT_FWDBWD, st, (int64_t)n_units * B * N, tag);
        const dim3 grid((unsigned)(ntiles * B * n_units));
        const bool avg = flags & MVF_AVG_REPROJ;
        if (S == 1 && !avg) hipLaunchKernelGGL((k_unit_fb<1, false>), grid, dim3(NT), fb_smem(), st, a);
        else if (S == 1) hipLaunchKernelGGL((k_unit_fb<1, true>), grid, dim3(NT), fb_smem(), st, a);
        else if (!avg) hipLaunchKernelGGL((k_unit_fb<2, false>), grid, dim3(NT), fb_smem(), st, a);
        else hipLaunchKernelGGL((k_unit_fb<2, true>), grid, dim3(NT), fb_smem(), st, a);
    }
    {
        // tile partials read once (27 floats per tile and image, 24 of them only with two sources)
        ProfScope ps(MVF_PROF_UNITS_FINISH, st, 4LL * n_units * B * (int64_t)ntiles * (NPART + 12 * S));
        hipLaunchKernelGGL(k_units_finish, dim3((unsigned)(n_units * B)), dim3(32 * FIN_SLICES), 0, st, a, S);
    }
    return hip_check_launch();
}

int mvf_units_fwdbwd_scale(const mvf_unit_scale_desc *units, int n_units, float smoothness, int B, int S,
                           int H, int W, void *stream)
{
    if (n_units < 1 || n_units > MVF_MAX_UNITS || !units) return (int)hipErrorInvalidValue;
    if (B * H * W <= 0) return 0;
    const int N = H * W;
    ScaleArgs a = {};
    bool vec = (N % 4 == 0);
    bool any_disp = false;
    for (int i = 0; i < n_units; ++i) {
        const mvf_unit_scale_desc &d = units[i];
        // g_disp == NULL: only grad_T of that unit (its disparity gradient goes raw to mvf_disp_head_bwd_units)
        if (!d.g_T_raw || !d.stats || (!d.g_loss && !d.g_sum) || !d.g_T || (d.g_disp && !d.g_disp_raw))
            return (int)hipErrorInvalidValue;
        any_disp = any_disp || d.g_disp;
        ScaleUnit &u = a.u[i];
        u.g_raw = d.g_disp_raw; u.gT_raw = d.g_T_raw; u.stats = d.stats; u.g_loss = d.g_loss; u.g_sum = d.g_sum;
        u.g_disp = d.g_disp; u.gT = d.g_T;
        u.in_stride = d.in_stride ? (size_t)d.in_stride : (size_t)N;
        u.out_stride = d.out_stride ? (size_t)d.out_stride : (size_t)N;
        // float4 path needs 16-B aligned image rows: N % 4 == 0, aligned bases and strides
        vec = vec && ((((uintptr_t)d.g_disp_raw) | ((uintptr_t)d.g_disp)) % 16 == 0) &&
              (u.in_stride % 4 == 0) && (u.out_stride % 4 == 0);
    }
    a.smoothness = smoothness;
    a.N = N;
    a.N4 = vec ? N / 4 : 0;
    a.nT = S * B * 16;
    const int per = any_disp ? (vec ? a.N4 : (N + 3) / 4) : 0;
    const unsigned gx = (unsigned)((max(per, a.nT) + 255) / 256);
    int nd = 0;
    for (int i = 0; i < n_units; ++i) nd += units[i].g_disp != nullptr;
    ProfScope ps(MVF_PROF_FB_SCALE, stream, 8LL * nd * B * N);            // raw gradient read, scaled gradient written
    hipLaunchKernelGGL(k_fb_scale, dim3(gx, (unsigned)(any_disp ? B : 0) + 1, (unsigned)n_units), dim3(256), 0,
                       (hipStream_t)stream, a);
    return hip_check_launch();
}

// one unit, contiguous tensors (the round-2 entry point; kept for C callers -- INTEGRATION.md).
// The ticket counters live at the end of the workspace and are zeroed here (one memset node).
int mvf_unit_fwdbwd(const float *disp, const float *tgt, const float *const *src, const float *T,
                    const float *K, const float *inv_K, const float *noise, const float *mask_rec,
                    int S, int flags, float smoothness, float min_disp, float range, float eps,
                    float *loss, uint8_t *argmin, float *auto_mask, float *to_opt, float *stats,
                    int32_t *idx_xy, float *g_disp, float *g_T, float *workspace, int B, int H, int W,
                    uint64_t noise_seed, float *noise_out, const float *disp_mean_partials, void *stream)
{
    if (S < 1 || S > 2) return (int)hipErrorInvalidValue;
    if (!g_disp || !g_T || !loss || !stats || !workspace) return (int)hipErrorInvalidValue;
    if (B * H * W <= 0) return 0;
    mvf_unit_desc d = {};
    d.disp = disp; d.tgt = tgt; d.src[0] = src[0]; d.src[1] = S > 1 ? src[1] : nullptr;
    d.T = T; d.K = K; d.inv_K = inv_K; d.mask_rec = mask_rec; d.noise = noise;
    d.disp_mean_partials = disp_mean_partials; d.noise_seed = noise_seed;
    d.loss = loss; d.stats = stats; d.g_disp_raw = g_disp; d.g_T_raw = g_T;
    d.argmin = argmin; d.auto_mask = auto_mask; d.to_opt = to_opt; d.idx_xy = idx_xy; d.noise_out = noise_out;
    // workspace = mvf_workspace_floats(B,H,W) floats (mvf_geom.hip), larger than one unit needs:
    // the tickets go behind the unit workspace
    const size_t wsf = (mvf_units_workspace_floats(1, B, H, W) + 1) & ~(size_t)1;
    int32_t *tickets = reinterpret_cast<int32_t *>(workspace + wsf);
    hipError_t e = hipMemsetAsync(tickets, 0, mvf_units_ticket_ints(1, B) * sizeof(int32_t), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return mvf_units_fwdbwd(&d, 1, S, flags, smoothness, min_disp, range, eps, workspace, tickets, B, H, W,
                            stream);
}

int mvf_unit_fwdbwd_scale(const float *g_disp_raw, const float *g_T_raw, const float *stats,
                          const float *g_loss, float smoothness, float *g_disp, float *g_T, int B,
                          int S, int H, int W, void *stream)
{
    if (B * H * W <= 0) return 0;
    mvf_unit_scale_desc d = {};
    d.g_disp_raw = g_disp_raw; d.g_T_raw = g_T_raw; d.stats = stats; d.g_loss = g_loss;
    d.g_disp = g_disp; d.g_T = g_T;
    return mvf_units_fwdbwd_scale(&d, 1, smoothness, B, S, H, W, stream);
}

}  // extern "C"
