// mvf_photo.hip -- photometric half of the hot path for gfx950 (MI355X):
//   SSIM (layers.py:261-290), compute_reprojection_loss (train.py:973-985),
//   compute_losses_base (train.py:987-1051), get_smooth_loss (layers.py:231-242), and the
//   fully fused UNIT (S x generate_images_pred + compute_losses_base) whose warped images
//   live only in LDS.
//
// Tile engine.  A workgroup of 256 lanes owns a 64x16-pixel compute region.  Every image
// plane it needs is staged once into LDS as an 18-row x 68-float plane (1-px reflect halo,
// rows padded to a multiple of 16 B), and every lane then owns a 4-pixel row segment:
// its 3x3 windows are three `ds_read_b128 + ds_read_b64` per plane (18 floats for 4
// pixels instead of 36 scalar reads), and global traffic is the coalesced plane staging
// plus the L2-served bilinear taps.
//   forward : region = output tile (grid steps 64x16)
//   backward: region = where the SSIM adjoint coefficients are formed; outputs are its
//             62x14 interior (grid steps 62x14), so the 3x3 adjoint gather never leaves
//             the workgroup and needs no atomics.
// Bound: the exact-mode window arithmetic (no shared partial sums, true divides) makes
// these kernels VALU-bound below the HBM roofline -- see DESIGN.md section 5.
// Compiled with -ffp-contract=off (arithmetic contract in mvf_common.hpp).
#include "mvf_common.hpp"

using namespace mvf;

namespace {

constexpr int TW = 64, TH = 16;          // compute region
constexpr int PX = 4;                    // pixels per lane (one row segment)
constexpr int NT = (TW / PX) * TH;       // 256 lanes
constexpr int PW = TW + 2;               // staged plane width (1-px halo)
constexpr int PH = TH + 2;
constexpr int LDW = TW + 4;              // LDS row stride (floats), multiple of 4
constexpr int PLANE = PH * LDW;          // floats per LDS plane
constexpr int RPLANE = TH * LDW;         // floats per region-sized LDS plane
constexpr int NMEAN = 32;                // partial sums per image of the disp mean
constexpr int NPART = 4;                 // floats per tile partial (photo, sx, sy, pad)

static_assert(NT == 256, "tile engine assumes 256 lanes");

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

// stage one [H,W] plane into LDS with reflect addressing; plane origin (py0, px0)
MVF_DEV void stage_plane(float *__restrict__ lds, const float *__restrict__ img, int H, int W,
                         int py0, int px0)
{
    for (int idx = threadIdx.x; idx < PH * PW; idx += NT) {
        int r = idx / PW, c = idx - r * PW;
        int gy = refl_clamp(py0 + r, H), gx = refl_clamp(px0 + c, W);
        lds[r * LDW + c] = img[(size_t)gy * W + gx];
    }
}

// stage 3 channel planes with all loads of a lane in flight together
MVF_DEV void stage_planes3(float *__restrict__ lds, const float *__restrict__ img, size_t N, int H,
                           int W, int py0, int px0)
{
    for (int idx = threadIdx.x; idx < PH * PW; idx += NT) {
        int r = idx / PW, c = idx - r * PW;
        int gy = refl_clamp(py0 + r, H), gx = refl_clamp(px0 + c, W);
        size_t o = (size_t)gy * W + gx;
        float v0 = img[o], v1 = img[N + o], v2 = img[2 * N + o];
        lds[r * LDW + c] = v0;
        lds[PLANE + r * LDW + c] = v1;
        lds[2 * PLANE + r * LDW + c] = v2;
    }
}

// 6 consecutive floats of an LDS plane row, starting at a 16-B aligned column
struct Row6 {
    float v[6];
};
MVF_DEV Row6 load_row6(const float *__restrict__ p)
{
    Row6 r;
    float4 a = *reinterpret_cast<const float4 *>(p);
    float2 b = *reinterpret_cast<const float2 *>(p + 4);
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y;
    return r;
}

// Window sums of the 4 pixels of a lane, row-major sequential order (exact mode).
// xs/ys point at plane element (row, 4*seg): the window of pixel j covers cols j..j+2.
struct Stats4 {
    float sx[PX], sxx[PX], sxy[PX];
    float xc[PX], yc[PX];   // centre values
};

MVF_DEV void window_x(const float *__restrict__ xs, const float *__restrict__ ys, Stats4 &o)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        Row6 x = load_row6(xs + r * LDW);
        Row6 y = load_row6(ys + r * LDW);
        float xx[6], xy[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            xx[i] = x.v[i] * x.v[i];
            xy[i] = x.v[i] * y.v[i];
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (r == 0 && d == 0) {
                    o.sx[j] = x.v[j];
                    o.sxx[j] = xx[j];
                    o.sxy[j] = xy[j];
                } else {
                    o.sx[j] = o.sx[j] + x.v[j + d];
                    o.sxx[j] = o.sxx[j] + xx[j + d];
                    o.sxy[j] = o.sxy[j] + xy[j + d];
                }
            }
            if (r == 1) {
                o.xc[j] = x.v[j + 1];
                o.yc[j] = y.v[j + 1];
            }
        }
    }
}

struct StatsY4 {
    float mu[PX], eyy[PX];
};

MVF_DEV void window_y(const float *__restrict__ ys, StatsY4 &o)
{
    float sy[PX], syy[PX];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        Row6 y = load_row6(ys + r * LDW);
        float yy[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) yy[i] = y.v[i] * y.v[i];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (r == 0 && d == 0) {
                    sy[j] = y.v[j];
                    syy[j] = yy[j];
                } else {
                    sy[j] = sy[j] + y.v[j + d];
                    syy[j] = syy[j] + yy[j + d];
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        o.mu[j] = div9(sy[j]);
        o.eyy[j] = div9(syy[j]);
    }
}

// reprojection map of one staged pred (3 planes) against the staged target for the 4
// pixels of this lane.   reference: train.py:973-985
MVF_DEV void reproj4(const float *__restrict__ pred, const float *__restrict__ tgt, int off,
                     const StatsY4 ty[3], bool no_ssim, float out[PX])
{
    float ab[3][PX], ss[3][PX];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (no_ssim) {
            Row6 x = load_row6(pred + c * PLANE + off + LDW);
            Row6 y = load_row6(tgt + c * PLANE + off + LDW);
#pragma unroll
            for (int j = 0; j < PX; ++j) ab[c][j] = fabsf(y.v[j + 1] - x.v[j + 1]);
        } else {
            Stats4 s;
            window_x(pred + c * PLANE + off, tgt + c * PLANE + off, s);
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                Win w = {div9(s.sx[j]), ty[c].mu[j], div9(s.sxx[j]), ty[c].eyy[j], div9(s.sxy[j])};
                ss[c][j] = clamp01(ssim_raw(w));
                ab[c][j] = fabsf(s.yc[j] - s.xc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        float l1 = div3((ab[0][j] + ab[1][j]) + ab[2][j]);
        if (no_ssim) {
            out[j] = l1;
        } else {
            float sm = div3((ss[0][j] + ss[1][j]) + ss[2][j]);
            out[j] = 0.85f * sm + 0.15f * l1;
        }
    }
}

// ---- fused warp into LDS: pred planes <- bilinear samples of src at the projected
// positions of every plane pixel (reflect-mapped into the image)
struct PoseLds {
    float P[MVF_MAX_SRC][12];
    float den;    // per-image mean disparity + 1e-7
    float gpix;   // backward: upstream grad / (B*H*W)
    float gloss;
    float pad;
};

MVF_DEV void load_pose_regs(const PoseLds &sh, int k, float P[12])
{
#pragma unroll
    for (int i = 0; i < 12; ++i) P[i] = sh.P[k][i];
}

MVF_DEV void warp_into_lds(float *__restrict__ pred, const float *__restrict__ dispP,
                           const float *__restrict__ src, const float *__restrict__ iK,
                           const float P[12], int H, int W, int py0, int px0, float min_disp,
                           float range, float eps, int32_t *__restrict__ idx_xy, int ty0, int tx0)
{
    size_t N = (size_t)H * W;
    for (int idx = threadIdx.x; idx < PH * PW; idx += NT) {
        int r = idx / PW, c = idx - r * PW;
        int gy = refl_clamp(py0 + r, H), gx = refl_clamp(px0 + c, W);
        WarpPoint w = warp_point(dispP[r * LDW + c], iK, P, gx, gy, H, W, min_disp, range, eps);
        const float *s00 = src + w.t.y0 * W + w.t.x0, *s01 = src + w.t.y0 * W + w.t.x1;
        const float *s10 = src + w.t.y1 * W + w.t.x0, *s11 = src + w.t.y1 * W + w.t.x1;
        // all 12 taps in flight before the first use
        float a0 = s00[0], b0 = s01[0], c0 = s10[0], d0 = s11[0];
        float a1 = s00[N], b1 = s01[N], c1 = s10[N], d1 = s11[N];
        float a2 = s00[2 * N], b2 = s01[2 * N], c2 = s10[2 * N], d2 = s11[2 * N];
        float fw = w.t.wx, fe = 1.0f - fw, fn = w.t.wy, fs = 1.0f - fn;
        float wnw = fs * fe, wne = fs * fw, wsw = fn * fe, wse = fn * fw;
        pred[r * LDW + c] = a0 * wnw + b0 * wne + c0 * wsw + d0 * wse;
        pred[PLANE + r * LDW + c] = a1 * wnw + b1 * wne + c1 * wsw + d1 * wse;
        pred[2 * PLANE + r * LDW + c] = a2 * wnw + b2 * wne + c2 * wsw + d2 * wse;
        if (idx_xy) {
            // the un-reflected pixels of this tile own their index entry
            int y = py0 + r, x = px0 + c;
            if (y >= ty0 && y < min(ty0 + TH, H) && x >= tx0 && x < min(tx0 + TW, W))
                reinterpret_cast<int2 *>(idx_xy)[(size_t)y * W + x] = make_int2(w.t.x0, w.t.y0);
        }
    }
}

// =============================================================================== forward
struct FwdArgs {
    const float *disp, *tgt, *noise, *mask, *T, *K, *invK;
    SrcPtrs warped, src;
    float *ws;             // [B*NMEAN] mean partials, then [B*ntiles*NPART] tile partials
    uint8_t *argmin;
    float *auto_mask, *to_opt;
    int32_t *idx_xy;
    int S, flags, B, H, W, tiles_x, tiles_y;
    float min_disp, range, eps;
};

template <bool FUSED, int S>
__global__ void __launch_bounds__(NT) k_photo_fwd(FwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tgtP = smem;                 // 3 planes
    float *predP = smem + 3 * PLANE;    // 3 planes
    float *dispP = smem + 6 * PLANE;    // 1 plane
    PoseLds &sh = *reinterpret_cast<PoseLds *>(smem + 7 * PLANE);
    float *scratch = smem + 7 * PLANE + sizeof(PoseLds) / 4;

    const TileId tid = tile_of_block(a.tiles_x, a.tiles_y, a.B);
    const int H = a.H, W = a.W, b = tid.b;
    const size_t N = (size_t)H * W;
    const int ty0 = tid.by * TH, tx0 = tid.bx * TW;
    const int py0 = ty0 - 1, px0 = tx0 - 1;
    const bool no_ssim = a.flags & MVF_NO_SSIM, avg = a.flags & MVF_AVG_REPROJ;
    const bool automask = !(a.flags & MVF_NO_AUTOMASK);

    if (threadIdx.x == 0) {
        float m = 0.0f;
        for (int i = 0; i < NMEAN; ++i) m += a.ws[b * NMEAN + i];
        sh.den = m / (float)N + 1e-7f;
    }
    if (FUSED && threadIdx.x < 12 * S) {
        int k = threadIdx.x / 12, e = threadIdx.x - k * 12;
        sh.P[k][e] = proj_entry(a.K + b * 16, a.T + ((size_t)k * a.B + b) * 16, e >> 2, e & 3);
    }
    stage_planes3(tgtP, a.tgt + (size_t)b * 3 * N, N, H, W, py0, px0);
    stage_plane(dispP, a.disp + (size_t)b * N, H, W, py0, px0);
    __syncthreads();

    const int seg = threadIdx.x & (TW / PX - 1), row = threadIdx.x / (TW / PX);
    const int off = row * LDW + seg * PX;   // plane element of the window's top-left
    const int y = ty0 + row, x0 = tx0 + seg * PX;

    StatsY4 ty[3];
    if (!no_ssim) {
#pragma unroll
        for (int c = 0; c < 3; ++c) window_y(tgtP + c * PLANE + off, ty[c]);
    }

    float rp[S][PX], idl[S][PX];
    const int npred = automask ? 2 * S : S;
#pragma unroll 1
    for (int p = 0; p < npred; ++p) {
        const bool is_id = p >= S;
        const int k = is_id ? p - S : p;
        __syncthreads();   // previous pred fully consumed
        if (FUSED && !is_id) {
            float P[12];
            load_pose_regs(sh, k, P);
            warp_into_lds(predP, dispP, a.src.p[k] + (size_t)b * 3 * N, a.invK + b * 16, P, H, W,
                          py0, px0, a.min_disp, a.range, a.eps,
                          a.idx_xy ? a.idx_xy + ((size_t)k * a.B + b) * N * 2 : nullptr, ty0, tx0);
        } else {
            const float *im = (is_id ? a.src.p[k] : a.warped.p[k]) + (size_t)b * 3 * N;
            stage_planes3(predP, im, N, H, W, py0, px0);
        }
        __syncthreads();
        float out[PX];
        reproj4(predP, tgtP, off, ty, no_ssim, out);
#pragma unroll
        for (int kk = 0; kk < S; ++kk)     // static register indices only
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                if (kk == k && is_id) idl[kk][j] = out[j];
                if (kk == k && !is_id) rp[kk][j] = out[j];
            }
    }

    // ---- candidates, min / argmin, mask, outputs (reference: train.py:1010-1043)
    float photo = 0.0f, sx = 0.0f, sy = 0.0f;
    const float den = sh.den;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int x = x0 + j;
        const bool live = (y < H) && (x < W);
        const size_t pix = (size_t)min(y, H - 1) * W + min(x, W - 1);
        const size_t pi = (size_t)b * N + pix;
        float best = 0.0f;
        int bi = 0, nc = 0;
        if (automask) {
            if (avg) {
                float m = idl[0][j];
#pragma unroll
                for (int k = 1; k < S; ++k) m = m + idl[k][j];
                m = m / (float)S;
                best = m + a.noise[pi] * 0.00001f;
                nc = 1;
            } else {
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    float v = idl[k][j] + a.noise[((size_t)b * S + k) * N + pix] * 0.00001f;
                    if (nc == 0 || v < best) { best = v; bi = nc; }
                    ++nc;
                }
            }
        }
        if (avg) {
            float m = rp[0][j];
#pragma unroll
            for (int k = 1; k < S; ++k) m = m + rp[k][j];
            m = m / (float)S;
            if (nc == 0 || m < best) { best = m; bi = nc; }
            ++nc;
        } else {
#pragma unroll
            for (int k = 0; k < S; ++k) {
                float v = rp[k][j];
                if (nc == 0 || v < best) { best = v; bi = nc; }
                ++nc;
            }
        }
        if (a.mask) best = best * a.mask[pi];
        if (live) {
            const int n_id = automask ? (avg ? 1 : S) : 0;
            a.argmin[pi] = (nc > 1) ? (uint8_t)bi : (uint8_t)255;
            if (a.auto_mask) a.auto_mask[pi] = (bi > n_id - 1) ? 1.0f : 0.0f;
            if (a.to_opt) a.to_opt[pi] = best;
            photo += best;
            // ---- edge-aware smoothness (reference: layers.py:231-242 on disp/(mean+1e-7))
            const float *dc = dispP + (row + 1) * LDW + seg * PX + 1 + j;
            const float *t0 = tgtP + (row + 1) * LDW + seg * PX + 1 + j;
            float nd = dc[0] / den;
            if (x + 1 < W) {
                float gd = fabsf(nd - dc[1] / den);
                float gi = div3((fabsf(t0[0] - t0[1]) + fabsf(t0[PLANE] - t0[PLANE + 1])) +
                                fabsf(t0[2 * PLANE] - t0[2 * PLANE + 1]));
                sx += gd * expf(-gi);
            }
            if (y + 1 < H) {
                float gd = fabsf(nd - dc[LDW] / den);
                float gi = div3((fabsf(t0[0] - t0[LDW]) + fabsf(t0[PLANE] - t0[PLANE + LDW])) +
                                fabsf(t0[2 * PLANE] - t0[2 * PLANE + LDW]));
                sy += gd * expf(-gi);
            }
        }
    }
    float *part = a.ws + (size_t)a.B * NMEAN +
                  (((size_t)b * a.tiles_y + tid.by) * a.tiles_x + tid.bx) * NPART;
    float r0 = block_sum<NT>(photo, scratch);
    float r1 = block_sum<NT>(sx, scratch);
    float r2 = block_sum<NT>(sy, scratch);
    if (threadIdx.x == 0) { part[0] = r0; part[1] = r1; part[2] = r2; part[3] = 0.0f; }
}

// per-image mean of disp: NMEAN partial sums per image, folded in fixed order by consumers
__global__ void __launch_bounds__(256) k_disp_mean(const float *__restrict__ disp,
                                                   float *__restrict__ ws, int N)
{
    __shared__ float scratch[4];
    int b = blockIdx.y, chunk = blockIdx.x;
    int per = (N + NMEAN - 1) / NMEAN;
    int lo = chunk * per, hi = min(lo + per, N);
    const float *d = disp + (size_t)b * N;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256) {
        s0 += d[i];
        s1 += d[i + 256];
        s2 += d[i + 512];
        s3 += d[i + 768];
    }
    for (; i < hi; i += 256) s0 += d[i];
    float r = block_sum<256>((s0 + s1) + (s2 + s3), scratch);
    if (threadIdx.x == 0) ws[b * NMEAN + chunk] = r;
}

// fold tile partials: loss[0..2], stats[B][4] = {mean, den, sx_b/Nx, sy_b/Ny}.
// One wave per image (fixed order inside the wave), images of a batch folded in order.
__global__ void __launch_bounds__(1024) k_finish_fwd(const float *__restrict__ ws,
                                                     float *__restrict__ loss,
                                                     float *__restrict__ stats, int B, int H, int W,
                                                     int ntiles, float smoothness, int want_photo)
{
    __shared__ double shp[16], shs[16];
    __shared__ double acc_photo, acc_smooth;
    const float *tp = ws + (size_t)B * NMEAN;
    const double N = (double)H * W;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) { acc_photo = 0.0; acc_smooth = 0.0; }
    __syncthreads();
    for (int b0 = 0; b0 < B; b0 += 16) {
        const int b = b0 + wid;
        if (b < B) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            for (int t = lane; t < ntiles; t += 64) {
                const float *q = tp + ((size_t)b * ntiles + t) * NPART;
                a0 += (double)q[0];
                a1 += (double)q[1];
                a2 += (double)q[2];
            }
            for (int off = 32; off > 0; off >>= 1) {
                a0 += __shfl_down(a0, off, 64);
                a1 += __shfl_down(a1, off, 64);
                a2 += __shfl_down(a2, off, 64);
            }
            if (lane == 0) {
                double sxb = a1 / ((double)B * H * (W - 1));
                double syb = a2 / ((double)B * (H - 1) * W);
                shp[wid] = a0;
                shs[wid] = sxb + syb;
                if (stats) {
                    float m = 0.0f;
                    for (int i = 0; i < NMEAN; ++i) m += ws[b * NMEAN + i];
                    float mean = m / (float)N;
                    stats[b * 4 + 0] = mean;
                    stats[b * 4 + 1] = mean + 1e-7f;
                    stats[b * 4 + 2] = (float)sxb;
                    stats[b * 4 + 3] = (float)syb;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 0; i < 16 && b0 + i < B; ++i) {
                acc_photo += shp[i];
                acc_smooth += shs[i];
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double pm = acc_photo / ((double)B * N);
        if (want_photo) {
            loss[0] = (float)(pm + (double)smoothness * acc_smooth);
            loss[1] = (float)pm;
            loss[2] = (float)acc_smooth;
        } else {
            loss[0] = (float)acc_smooth;
        }
    }
}

// =============================================================================== backward
struct BwdArgs {
    const float *disp, *tgt, *mask, *T, *K, *invK, *stats, *g_loss;
    const uint8_t *argmin;
    SrcPtrs warped, src;
    DstPtrs g_warped;
    float *g_disp, *ws;
    int S, flags, B, H, W, tiles_x, tiles_y;
    float smoothness, min_disp, range, eps;
};

constexpr int OW = TW - 2, OH = TH - 2;   // output interior of a backward region

template <bool FUSED, int S>
__global__ void __launch_bounds__(NT) k_photo_bwd(BwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tgtP = smem;                 // 3 planes
    float *predP = smem + 3 * PLANE;    // 3 planes
    float *dispP = smem + 6 * PLANE;    // 1 plane
    float *coefP = smem + 7 * PLANE;    // 3 region planes (A, B, G); reused for grad_warped
    float *gdP = coefP + 3 * RPLANE;    // 1 region plane: grad_depth -> grad_disp accumulator
    PoseLds &sh = *reinterpret_cast<PoseLds *>(gdP + RPLANE);
    float *scratch = gdP + RPLANE + sizeof(PoseLds) / 4;

    const TileId tid = tile_of_block(a.tiles_x, a.tiles_y, a.B);
    const int H = a.H, W = a.W, b = tid.b;
    const size_t N = (size_t)H * W;
    const int cy0 = tid.by * OH - 1, cx0 = tid.bx * OW - 1;   // region origin
    const int py0 = cy0 - 1, px0 = cx0 - 1;                           // plane origin
    const bool no_ssim = a.flags & MVF_NO_SSIM, avg = a.flags & MVF_AVG_REPROJ;
    const bool automask = !(a.flags & MVF_NO_AUTOMASK);
    const int n_id = automask ? (avg ? 1 : S) : 0;

    if (threadIdx.x == 0) {
        float g = a.g_loss[0];
        sh.gloss = g;
        sh.gpix = g / (float)((double)a.B * (double)N);
        sh.den = a.stats[b * 4 + 1];
    }
    if (FUSED && threadIdx.x < 12 * S) {
        int k = threadIdx.x / 12, e = threadIdx.x - k * 12;
        sh.P[k][e] = proj_entry(a.K + b * 16, a.T + ((size_t)k * a.B + b) * 16, e >> 2, e & 3);
    }
    stage_planes3(tgtP, a.tgt + (size_t)b * 3 * N, N, H, W, py0, px0);
    stage_plane(dispP, a.disp + (size_t)b * N, H, W, py0, px0);
    for (int i = threadIdx.x; i < RPLANE; i += NT) gdP[i] = 0.0f;
    __syncthreads();

    const int seg = threadIdx.x & (TW / PX - 1), row = threadIdx.x / (TW / PX);
    const int off = row * LDW + seg * PX;
    const int roff = row * LDW + seg * PX;   // region-plane element of this lane's first pixel
    const int y = cy0 + row, x0 = cx0 + seg * PX;
    const bool rowin = (y >= 0) && (y < H);
    const bool row_out = (row >= 1) && (row <= OH) && rowin;   // interior (= output) rows

    // base weight of every region pixel: gpix * mask (0 outside the image), and its argmin
    float wbase[PX];
    int sel[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int x = x0 + j;
        const bool in = rowin && (x >= 0) && (x < W);
        const size_t pi = (size_t)b * N + (size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
        sel[j] = in ? (int)a.argmin[pi] : 254;
        float m = (in && a.mask) ? a.mask[pi] : 1.0f;
        wbase[j] = in ? sh.gpix * m : 0.0f;
    }

    StatsY4 ty[3];
    if (!no_ssim) {
#pragma unroll
        for (int c = 0; c < 3; ++c) window_y(tgtP + c * PLANE + off, ty[c]);
    }
    const float myu = (y == 1) ? 2.0f : 1.0f, myd = (y == H - 2) ? 2.0f : 1.0f;

#pragma unroll 1
    for (int k = 0; k < S; ++k) {
        asm volatile("" ::: "memory");   // keep k-invariant LDS reads inside the loop (VGPRs)
        __syncthreads();
        if (FUSED) {
            float P[12];
            load_pose_regs(sh, k, P);
            warp_into_lds(predP, dispP, a.src.p[k] + (size_t)b * 3 * N, a.invK + b * 16, P, H, W,
                          py0, px0, a.min_disp, a.range, a.eps, nullptr, 0, 0);
        } else {
            stage_planes3(predP, a.warped.p[k] + (size_t)b * 3 * N, N, H, W, py0, px0);
        }
        __syncthreads();

        // selection weight of this source: the argmin picked it (or the averaged channel)
        float wk[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            float w;
            if (sel[j] == 255) w = avg ? 1.0f / (float)S : 1.0f;        // single candidate
            else if (avg) w = (sel[j] == n_id) ? 1.0f / (float)S : 0.0f;
            else w = (sel[j] == n_id + k) ? 1.0f : 0.0f;
            wk[j] = wbase[j] * w;
        }

        float gw[3][PX];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float *xp = predP + c * PLANE, *yp = tgtP + c * PLANE;
            float xq[PX], yq[PX];
            if (no_ssim) {
                Row6 xr = load_row6(xp + off + LDW), yr = load_row6(yp + off + LDW);
#pragma unroll
                for (int j = 0; j < PX; ++j) { xq[j] = xr.v[j + 1]; yq[j] = yr.v[j + 1]; }
            } else {
                Stats4 s;
                window_x(xp + off, yp + off, s);
                float cA[PX], cB[PX], cG[PX];
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    Win w = {div9(s.sx[j]), ty[c].mu[j], div9(s.sxx[j]), ty[c].eyy[j],
                             div9(s.sxy[j])};
                    DWin d = ssim_partials(w);
                    float g = wk[j] * ((0.85f / 3.0f) / 9.0f);
                    cA[j] = g * d.dmux;
                    cB[j] = g * 2.0f * d.dexx;
                    cG[j] = g * d.dexy;
                    xq[j] = s.xc[j];
                    yq[j] = s.yc[j];
                }
                float *cp = coefP + roff;
                *reinterpret_cast<float4 *>(cp) = make_float4(cA[0], cA[1], cA[2], cA[3]);
                *reinterpret_cast<float4 *>(cp + RPLANE) = make_float4(cB[0], cB[1], cB[2], cB[3]);
                *reinterpret_cast<float4 *>(cp + 2 * RPLANE) = make_float4(cG[0], cG[1], cG[2], cG[3]);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                // L1 term: d|t-p|/dp = -sign(t-p), channel mean
                float df = yq[j] - xq[j];
                float sg = (df > 0.0f) ? -1.0f : ((df < 0.0f) ? 1.0f : 0.0f);
                gw[c][j] = wk[j] * (no_ssim ? 1.0f : 0.15f) * sg / 3.0f;
            }
            if (!no_ssim && row >= 1 && row <= OH) {
                float sA[PX] = {0.f, 0.f, 0.f, 0.f}, sB[PX] = {0.f, 0.f, 0.f, 0.f},
                      sG[PX] = {0.f, 0.f, 0.f, 0.f};
                const bool hasl = seg > 0, hasr = seg < TW / PX - 1;
#pragma unroll
                for (int dr = -1; dr <= 1; ++dr) {
                    const float my = (dr < 0) ? myu : ((dr > 0) ? myd : 1.0f);
                    // coefficient columns 4*seg-1 .. 4*seg+4 of region row (row+dr)
                    const float *cr = coefP + roff + dr * LDW;
                    float4 mA = *reinterpret_cast<const float4 *>(cr);
                    float4 mB = *reinterpret_cast<const float4 *>(cr + RPLANE);
                    float4 mG = *reinterpret_cast<const float4 *>(cr + 2 * RPLANE);
                    float vA[6] = {hasl ? cr[-1] : 0.f, mA.x, mA.y, mA.z, mA.w, hasr ? cr[4] : 0.f};
                    float vB[6] = {hasl ? cr[RPLANE - 1] : 0.f, mB.x, mB.y, mB.z, mB.w,
                                   hasr ? cr[RPLANE + 4] : 0.f};
                    float vG[6] = {hasl ? cr[2 * RPLANE - 1] : 0.f, mG.x, mG.y, mG.z, mG.w,
                                   hasr ? cr[2 * RPLANE + 4] : 0.f};
#pragma unroll
                    for (int j = 0; j < PX; ++j) {
                        // reflect-pad multiplicity: neighbour column 0 seen twice from column 1, ...
                        const float ml = (x0 + j == 1) ? 2.0f : 1.0f;
                        const float mr = (x0 + j == W - 2) ? 2.0f : 1.0f;
                        sA[j] += my * (ml * vA[j] + vA[j + 1] + mr * vA[j + 2]);
                        sB[j] += my * (ml * vB[j] + vB[j + 1] + mr * vB[j + 2]);
                        sG[j] += my * (ml * vG[j] + vG[j + 1] + mr * vG[j + 2]);
                    }
                }
#pragma unroll
                for (int j = 0; j < PX; ++j) gw[c][j] += sA[j] + xq[j] * sB[j] + yq[j] * sG[j];
            }
            __syncthreads();   // coefficient planes free for the next channel
        }

        // ---- consume grad_warped of this source
        if (!FUSED) {
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const int x = x0 + j, col = seg * PX + j;
                const bool outp = row_out && (col >= 1) && (col <= OW) && (x >= 0) && (x < W);
                if (!outp) continue;
                const size_t pi = (size_t)y * W + x;
#pragma unroll
                for (int c = 0; c < 3; ++c) a.g_warped.p[k][((size_t)b * 3 + c) * N + pi] = gw[c][j];
            }
        } else {
            // park grad_warped in the (now free) coefficient planes, then walk this lane's
            // pixels one at a time through the projection adjoint (keeps the live set small)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                *reinterpret_cast<float4 *>(coefP + c * RPLANE + roff) =
                    make_float4(gw[c][0], gw[c][1], gw[c][2], gw[c][3]);
            float P[12];
            load_pose_regs(sh, k, P);
            const float *sp = a.src.p[k] + (size_t)b * 3 * N;
            float accP[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) accP[q] = 0.0f;
#pragma unroll 1
            for (int j = 0; j < PX; ++j) {
                const int x = x0 + j, col = seg * PX + j;
                const bool outp = row_out && (col >= 1) && (col <= OW) && (x >= 0) && (x < W);
                if (!outp) continue;
                const float g0 = coefP[roff + j], g1 = coefP[RPLANE + roff + j],
                            g2 = coefP[2 * RPLANE + roff + j];
                WarpPoint w = warp_point(dispP[(row + 1) * LDW + col + 1], a.invK + b * 16, P, x, y,
                                         H, W, a.min_disp, a.range, a.eps);
                float dx0, dy0, dx1, dy1, dx2, dy2;
                bilerp_grad(sp, W, w.t, dx0, dy0);
                bilerp_grad(sp + N, W, w.t, dx1, dy1);
                bilerp_grad(sp + 2 * N, W, w.t, dx2, dy2);
                float gix = g0 * dx0 + g1 * dx1 + g2 * dx2;
                float giy = g0 * dy0 + g1 * dy1 + g2 * dy2;
                float gc[3];
                float gd = warp_point_bwd(w, P, gix, giy, H, W, gc);
                gdP[roff + j] += -gd * w.depth * w.depth * a.range;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    accP[q * 4 + 0] += gc[q] * w.X[0];
                    accP[q * 4 + 1] += gc[q] * w.X[1];
                    accP[q * 4 + 2] += gc[q] * w.X[2];
                    accP[q * 4 + 3] += gc[q];
                }
            }
            // per-tile partial of grad_P for source k: ws[((k*B + b)*ntiles + tile)*12 + q]
            const size_t ntiles = (size_t)a.tiles_x * a.tiles_y;
            float *part = a.ws + ((((size_t)k * a.B + b) * ntiles) +
                                  (size_t)tid.by * a.tiles_x + tid.bx) * 12;
#pragma unroll
            for (int q = 0; q < 12; ++q) {
                float s = block_sum<NT>(accP[q], scratch);
                if (threadIdx.x == 0) part[q] = s;
            }
        }
    }

    // ---- smoothness gradient + store grad_disp (reference: layers.py:231-242, train.py:1044-1049)
    if (a.g_disp) {
        const float den = sh.den;
        const float scale = sh.gloss * a.smoothness;
        const float cx = scale / (float)((double)a.B * H * (W - 1));
        const float cy = scale / (float)((double)a.B * (H - 1) * W);
        const float smooth_b = a.stats[b * 4 + 2] + a.stats[b * 4 + 3];
        // d/d disp_j of scale*smooth(disp/den): gn_j/den - (sum_i gn_i d_i)/(den^2 N); the sum is
        // den*scale*smooth_b because the per-image term is positively homogeneous of degree 1
        const float corr = scale * smooth_b / (float)N;
#pragma unroll 1
        for (int j = 0; j < PX; ++j) {
            const int x = x0 + j;
            const int col = seg * PX + j;
            const bool outp = row_out && (col >= 1) && (col <= OW) && (x >= 0) && (x < W);
            if (!outp) continue;
            const float *dc = dispP + (row + 1) * LDW + col + 1;
            const float *t0 = tgtP + (row + 1) * LDW + col + 1;
            float nd = dc[0] / den;
            float gn = 0.0f;
            auto wgt = [&](int o) {
                float gi = div3((fabsf(t0[0] - t0[o]) + fabsf(t0[PLANE] - t0[PLANE + o])) +
                                fabsf(t0[2 * PLANE] - t0[2 * PLANE + o]));
                return expf(-gi);
            };
            auto sgn = [](float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); };
            if (x + 1 < W) gn += cx * wgt(1) * sgn(nd - dc[1] / den);
            if (x - 1 >= 0) gn -= cx * wgt(-1) * sgn(dc[-1] / den - nd);
            if (y + 1 < H) gn += cy * wgt(LDW) * sgn(nd - dc[LDW] / den);
            if (y - 1 >= 0) gn -= cy * wgt(-LDW) * sgn(dc[-LDW] / den - nd);
            a.g_disp[(size_t)b * N + (size_t)y * W + x] = gdP[roff + j] + gn / den - corr / den;
        }
    }
}

// =============================================================================== standalone
// SSIM map, general channel count (layers.py:277-290).  One lane per pixel.
__global__ void __launch_bounds__(256) k_ssim_fwd(const float *__restrict__ x,
                                                  const float *__restrict__ y,
                                                  float *__restrict__ out, int H, int W)
{
    int N = H * W;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    size_t base = (size_t)blockIdx.y * N;
    int py = i / W, px = i - py * W;
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        int yy = refl_clamp(py + dy, H);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int xx = refl_clamp(px + dx, W);
            float a = x[base + (size_t)yy * W + xx], bq = y[base + (size_t)yy * W + xx];
            sx = sx + a;
            sy = sy + bq;
            sxx = sxx + a * a;
            syy = syy + bq * bq;
            sxy = sxy + a * bq;
        }
    }
    Win w = {div9(sx), div9(sy), div9(sxx), div9(syy), div9(sxy)};
    out[base + i] = clamp01(ssim_raw(w));
}

MVF_DEV Win window_at(const float *__restrict__ x, const float *__restrict__ y, int H, int W, int py,
                      int px)
{
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        int yy = refl_clamp(py + dy, H);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int xx = refl_clamp(px + dx, W);
            float a = x[(size_t)yy * W + xx], bq = y[(size_t)yy * W + xx];
            sx = sx + a;
            sy = sy + bq;
            sxx = sxx + a * a;
            syy = syy + bq * bq;
            sxy = sxy + a * bq;
        }
    }
    Win w = {div9(sx), div9(sy), div9(sxx), div9(syy), div9(sxy)};
    return w;
}

// SSIM adjoint in gather form: every pixel q re-derives the coefficients of its 3x3
// neighbours (API-completeness kernel; the hot path uses the tiled adjoint above).
// mode 0: plain SSIM with upstream map g_out [planes,H,W] -> g_x, g_y
__global__ void __launch_bounds__(256) k_ssim_bwd(const float *__restrict__ x,
                                                  const float *__restrict__ y,
                                                  const float *__restrict__ g_out,
                                                  float *__restrict__ g_x, float *__restrict__ g_y,
                                                  int H, int W)
{
    int N = H * W;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    size_t base = (size_t)blockIdx.y * N;
    const float *xb = x + base, *yb = y + base, *gb = g_out + base;
    int qy = i / W, qx = i - qy * W;
    float xq = xb[i], yq = yb[i];
    float ax = 0.f, ay = 0.f;
    for (int dy = -1; dy <= 1; ++dy) {
        int py = qy + dy;
        if (py < 0 || py >= H) continue;
        float my = refl_mult(py, qy, H);
        for (int dx = -1; dx <= 1; ++dx) {
            int px = qx + dx;
            if (px < 0 || px >= W) continue;
            float m = my * refl_mult(px, qx, W);
            float g = gb[(size_t)py * W + px];
            if (g == 0.0f) continue;
            Win w = window_at(xb, yb, H, W, py, px);
            DWin d = ssim_partials(w);
            ax += m * g * (d.dmux + 2.0f * xq * d.dexx + yq * d.dexy);
            ay += m * g * (d.dmuy + 2.0f * yq * d.deyy + xq * d.dexy);
        }
    }
    if (g_x) g_x[base + i] = ax / 9.0f;
    if (g_y) g_y[base + i] = ay / 9.0f;
}

// reprojection map of a standalone (pred, target) pair (train.py:973-985)
__global__ void __launch_bounds__(256) k_reproj_fwd(const float *__restrict__ pred,
                                                    const float *__restrict__ tgt,
                                                    float *__restrict__ out, int H, int W,
                                                    int no_ssim)
{
    int N = H * W;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    int b = blockIdx.y;
    const float *pb = pred + (size_t)b * 3 * N, *tb = tgt + (size_t)b * 3 * N;
    int py = i / W, px = i - py * W;
    float ab[3], ss[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ab[c] = fabsf(tb[(size_t)c * N + i] - pb[(size_t)c * N + i]);
        if (!no_ssim) {
            Win w = window_at(pb + (size_t)c * N, tb + (size_t)c * N, H, W, py, px);
            ss[c] = clamp01(ssim_raw(w));
        }
    }
    float l1 = div3((ab[0] + ab[1]) + ab[2]);
    if (no_ssim) {
        out[(size_t)b * N + i] = l1;
    } else {
        float sm = div3((ss[0] + ss[1]) + ss[2]);
        out[(size_t)b * N + i] = 0.85f * sm + 0.15f * l1;
    }
}

__global__ void __launch_bounds__(256) k_reproj_bwd(const float *__restrict__ pred,
                                                    const float *__restrict__ tgt,
                                                    const float *__restrict__ g_out,
                                                    float *__restrict__ g_pred, int H, int W,
                                                    int no_ssim)
{
    int N = H * W;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    int b = blockIdx.y;
    const float *gb = g_out + (size_t)b * N;
    int qy = i / W, qx = i - qy * W;
    for (int c = 0; c < 3; ++c) {
        const float *xb = pred + ((size_t)b * 3 + c) * N, *yb = tgt + ((size_t)b * 3 + c) * N;
        float xq = xb[i], yq = yb[i];
        float df = yq - xq;
        float sg = (df > 0.0f) ? -1.0f : ((df < 0.0f) ? 1.0f : 0.0f);
        float g = gb[i] * (no_ssim ? 1.0f : 0.15f) * sg / 3.0f;
        if (!no_ssim) {
            float ax = 0.f;
            for (int dy = -1; dy <= 1; ++dy) {
                int py = qy + dy;
                if (py < 0 || py >= H) continue;
                float my = refl_mult(py, qy, H);
                for (int dx = -1; dx <= 1; ++dx) {
                    int px = qx + dx;
                    if (px < 0 || px >= W) continue;
                    float go = gb[(size_t)py * W + px];
                    if (go == 0.0f) continue;
                    float m = my * refl_mult(px, qx, W);
                    Win w = window_at(xb, yb, H, W, py, px);
                    DWin d = ssim_partials(w);
                    ax += m * go * (d.dmux + 2.0f * xq * d.dexx + yq * d.dexy);
                }
            }
            g += ax * (0.85f / 3.0f) / 9.0f;
        }
        g_pred[((size_t)b * 3 + c) * N + i] = g;
    }
}

// standalone smoothness (layers.py:231-242): one lane per pixel, tile partials in ws
__global__ void __launch_bounds__(256) k_smooth_fwd(const float *__restrict__ disp,
                                                    const float *__restrict__ img,
                                                    float *__restrict__ ws, int normalise, int B,
                                                    int H, int W)
{
    __shared__ float scratch[4];
    __shared__ float sden;
    int N = H * W, b = blockIdx.y;
    if (threadIdx.x == 0) {
        float den = 1.0f;
        if (normalise) {
            float m = 0.0f;
            for (int i = 0; i < NMEAN; ++i) m += ws[b * NMEAN + i];
            den = m / (float)N + 1e-7f;
        }
        sden = den;
    }
    __syncthreads();
    const float den = sden;
    int i = blockIdx.x * 256 + threadIdx.x;
    float sx = 0.f, sy = 0.f;
    if (i < N) {
        int y = i / W, x = i - y * W;
        const float *d = disp + (size_t)b * N;
        const float *im = img + (size_t)b * 3 * N;
        float nd = normalise ? d[i] / den : d[i];
        if (x + 1 < W) {
            float o = normalise ? d[i + 1] / den : d[i + 1];
            float gi = ((fabsf(im[i] - im[i + 1]) + fabsf(im[N + i] - im[N + i + 1])) +
                        fabsf(im[2 * N + i] - im[2 * N + i + 1])) / 3.0f;
            sx = fabsf(nd - o) * expf(-gi);
        }
        if (y + 1 < H) {
            float o = normalise ? d[i + W] / den : d[i + W];
            float gi = ((fabsf(im[i] - im[i + W]) + fabsf(im[N + i] - im[N + i + W])) +
                        fabsf(im[2 * N + i] - im[2 * N + i + W])) / 3.0f;
            sy = fabsf(nd - o) * expf(-gi);
        }
    }
    float r1 = block_sum<256>(sx, scratch);
    float r2 = block_sum<256>(sy, scratch);
    if (threadIdx.x == 0) {
        float *part = ws + (size_t)B * NMEAN + ((size_t)b * gridDim.x + blockIdx.x) * NPART;
        part[0] = 0.0f; part[1] = r1; part[2] = r2; part[3] = 0.0f;
    }
}

__global__ void __launch_bounds__(256) k_smooth_bwd(const float *__restrict__ disp,
                                                    const float *__restrict__ img,
                                                    const float *__restrict__ stats,
                                                    const float *__restrict__ g_loss, float scale_in,
                                                    float *__restrict__ g_disp, int accumulate,
                                                    int normalise, int B, int H, int W)
{
    int N = H * W, b = blockIdx.y;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float den = normalise ? stats[b * 4 + 1] : 1.0f;
    const float scale = g_loss[0] * scale_in;
    const float cx = scale / (float)((double)B * H * (W - 1));
    const float cy = scale / (float)((double)B * (H - 1) * W);
    int y = i / W, x = i - y * W;
    const float *d = disp + (size_t)b * N + i;
    const float *t0 = img + (size_t)b * 3 * N + i;
    auto nrm = [&](float v) { return normalise ? v / den : v; };
    auto wgt = [&](int o) {
        float gi = ((fabsf(t0[0] - t0[o]) + fabsf(t0[N] - t0[N + o])) +
                    fabsf(t0[2 * (size_t)N] - t0[2 * (size_t)N + o])) / 3.0f;
        return expf(-gi);
    };
    auto sgn = [](float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); };
    float nd = nrm(d[0]);
    float gn = 0.0f;
    if (x + 1 < W) gn += cx * wgt(1) * sgn(nd - nrm(d[1]));
    if (x - 1 >= 0) gn -= cx * wgt(-1) * sgn(nrm(d[-1]) - nd);
    if (y + 1 < H) gn += cy * wgt(W) * sgn(nd - nrm(d[W]));
    if (y - 1 >= 0) gn -= cy * wgt(-W) * sgn(nrm(d[-W]) - nd);
    float g;
    if (normalise) {
        float smooth_b = stats[b * 4 + 2] + stats[b * 4 + 3];
        g = gn / den - scale * smooth_b / (float)N / den;
    } else {
        g = gn;
    }
    size_t o = (size_t)b * N + i;
    g_disp[o] = accumulate ? g_disp[o] + g : g;
}

inline size_t fwd_smem() { return (7 * PLANE) * sizeof(float) + sizeof(PoseLds) + 8 * sizeof(float); }
inline size_t bwd_smem()
{
    return (7 * PLANE + 4 * RPLANE) * sizeof(float) + sizeof(PoseLds) + 8 * sizeof(float);
}

template <bool FUSED>
void launch_fwd_kernel(const FwdArgs &a, dim3 grid, hipStream_t st)
{
    switch (a.S) {
    case 1: hipLaunchKernelGGL((k_photo_fwd<FUSED, 1>), grid, dim3(NT), fwd_smem(), st, a); break;
    case 2: hipLaunchKernelGGL((k_photo_fwd<FUSED, 2>), grid, dim3(NT), fwd_smem(), st, a); break;
    case 3: hipLaunchKernelGGL((k_photo_fwd<FUSED, 3>), grid, dim3(NT), fwd_smem(), st, a); break;
    default: hipLaunchKernelGGL((k_photo_fwd<FUSED, 4>), grid, dim3(NT), fwd_smem(), st, a); break;
    }
}

template <bool FUSED>
void launch_bwd_kernel(const BwdArgs &a, dim3 grid, hipStream_t st)
{
    switch (a.S) {
    case 1: hipLaunchKernelGGL((k_photo_bwd<FUSED, 1>), grid, dim3(NT), bwd_smem(), st, a); break;
    case 2: hipLaunchKernelGGL((k_photo_bwd<FUSED, 2>), grid, dim3(NT), bwd_smem(), st, a); break;
    case 3: hipLaunchKernelGGL((k_photo_bwd<FUSED, 3>), grid, dim3(NT), bwd_smem(), st, a); break;
    default: hipLaunchKernelGGL((k_photo_bwd<FUSED, 4>), grid, dim3(NT), bwd_smem(), st, a); break;
    }
}

int launch_fwd(bool fused, FwdArgs &a, float smoothness, float *loss, float *stats, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int N = a.H * a.W;
    a.tiles_x = (a.W + TW - 1) / TW;
    a.tiles_y = (a.H + TH - 1) / TH;
    hipLaunchKernelGGL(k_disp_mean, dim3(NMEAN, a.B), dim3(256), 0, st, a.disp, a.ws, N);
    dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.B));
    {
        ProfScope ps(fused ? MVF_PROF_UNIT_FWD : MVF_PROF_PHOTO_FWD, st);
        if (fused) launch_fwd_kernel<true>(a, grid, st);
        else launch_fwd_kernel<false>(a, grid, st);
    }
    hipLaunchKernelGGL(k_finish_fwd, dim3(1), dim3(1024), 0, st, a.ws, loss, stats, a.B, a.H, a.W,
                       a.tiles_x * a.tiles_y, smoothness, 1);
    return hip_check_launch();
}

}  // namespace

// from mvf_geom.hip
namespace mvf_geom {
int finish_gT(const float *ws, const float *K, float *gT, int B, int S, int nblk, void *stream);
}

// ============================================================================ C ABI
extern "C" {

int mvf_ssim_fwd(const float *x, const float *y, float *out, int B, int C, int H, int W,
                 void *stream)
{
    if (B * C * H * W <= 0) return 0;
    hipLaunchKernelGGL(k_ssim_fwd, dim3((H * W + 255) / 256, B * C), dim3(256), 0,
                       (hipStream_t)stream, x, y, out, H, W);
    return hip_check_launch();
}

int mvf_ssim_bwd(const float *x, const float *y, const float *g_out, float *g_x, float *g_y,
                 int B, int C, int H, int W, void *stream)
{
    if (B * C * H * W <= 0) return 0;
    hipLaunchKernelGGL(k_ssim_bwd, dim3((H * W + 255) / 256, B * C), dim3(256), 0,
                       (hipStream_t)stream, x, y, g_out, g_x, g_y, H, W);
    return hip_check_launch();
}

int mvf_reprojection_fwd(const float *pred, const float *target, float *out, int B, int H, int W,
                         int no_ssim, void *stream)
{
    if (B * H * W <= 0) return 0;
    hipLaunchKernelGGL(k_reproj_fwd, dim3((H * W + 255) / 256, B), dim3(256), 0,
                       (hipStream_t)stream, pred, target, out, H, W, no_ssim);
    return hip_check_launch();
}

int mvf_reprojection_bwd(const float *pred, const float *target, const float *g_out, float *g_pred,
                         int B, int H, int W, int no_ssim, void *stream)
{
    if (B * H * W <= 0) return 0;
    hipLaunchKernelGGL(k_reproj_bwd, dim3((H * W + 255) / 256, B), dim3(256), 0,
                       (hipStream_t)stream, pred, target, g_out, g_pred, H, W, no_ssim);
    return hip_check_launch();
}

int mvf_smooth_fwd(const float *disp, const float *img, float *out, float *stats, float *workspace,
                   int normalise, int B, int H, int W, void *stream)
{
    if (B * H * W <= 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int nblk = (H * W + 255) / 256;
    hipLaunchKernelGGL(k_disp_mean, dim3(NMEAN, B), dim3(256), 0, st, disp, workspace, H * W);
    hipLaunchKernelGGL(k_smooth_fwd, dim3(nblk, B), dim3(256), 0, st, disp, img, workspace,
                       normalise, B, H, W);
    hipLaunchKernelGGL(k_finish_fwd, dim3(1), dim3(1024), 0, st, workspace, out, stats, B, H, W,
                       nblk, 1.0f, 0);
    return hip_check_launch();
}

int mvf_smooth_bwd(const float *disp, const float *img, const float *stats, const float *g_loss,
                   float scale, float *g_disp, int accumulate, int normalise, int B, int H, int W,
                   void *stream)
{
    if (B * H * W <= 0) return 0;
    hipLaunchKernelGGL(k_smooth_bwd, dim3((H * W + 255) / 256, B), dim3(256), 0,
                       (hipStream_t)stream, disp, img, stats, g_loss, scale, g_disp, accumulate,
                       normalise, B, H, W);
    return hip_check_launch();
}

int mvf_photo_fwd(const float *disp, const float *tgt, const float *const *warped,
                  const float *const *src, const float *noise, const float *mask_rec, int S,
                  int flags, float smoothness, float *loss, uint8_t *argmin, float *auto_mask,
                  float *to_opt, float *stats, float *workspace, int B, int H, int W, void *stream)
{
    if (S < 1 || S > MVF_MAX_SRC) return (int)hipErrorInvalidValue;
    if (B * H * W <= 0) return 0;
    FwdArgs a = {};
    a.disp = disp; a.tgt = tgt; a.noise = noise; a.mask = mask_rec;
    for (int k = 0; k < S; ++k) {
        a.warped.p[k] = warped[k];
        a.src.p[k] = (flags & MVF_NO_AUTOMASK) ? nullptr : src[k];
    }
    a.ws = workspace; a.argmin = argmin; a.auto_mask = auto_mask; a.to_opt = to_opt;
    a.S = S; a.flags = flags; a.B = B; a.H = H; a.W = W;
    return launch_fwd(false, a, smoothness, loss, stats, stream);
}

int mvf_unit_fwd(const float *disp, const float *tgt, const float *const *src, const float *T,
                 const float *K, const float *inv_K, const float *noise, const float *mask_rec,
                 int S, int flags, float smoothness, float min_disp, float range, float eps,
                 float *loss, uint8_t *argmin, float *auto_mask, float *to_opt, float *stats,
                 int32_t *idx_xy, float *workspace, int B, int H, int W, void *stream)
{
    if (S < 1 || S > MVF_MAX_SRC) return (int)hipErrorInvalidValue;
    if (B * H * W <= 0) return 0;
    FwdArgs a = {};
    a.disp = disp; a.tgt = tgt; a.noise = noise; a.mask = mask_rec; a.T = T; a.K = K; a.invK = inv_K;
    for (int k = 0; k < S; ++k) a.src.p[k] = src[k];
    a.ws = workspace; a.argmin = argmin; a.auto_mask = auto_mask; a.to_opt = to_opt;
    a.idx_xy = idx_xy;
    a.S = S; a.flags = flags; a.B = B; a.H = H; a.W = W;
    a.min_disp = min_disp; a.range = range; a.eps = eps;
    return launch_fwd(true, a, smoothness, loss, stats, stream);
}

int mvf_photo_bwd(const float *disp, const float *tgt, const float *const *warped,
                  const uint8_t *argmin, const float *mask_rec, const float *stats,
                  const float *g_loss, int S, int flags, float smoothness, float *const *g_warped,
                  float *g_disp, int B, int H, int W, void *stream)
{
    if (S < 1 || S > MVF_MAX_SRC) return (int)hipErrorInvalidValue;
    if (B * H * W <= 0) return 0;
    BwdArgs a = {};
    a.disp = disp; a.tgt = tgt; a.mask = mask_rec; a.stats = stats; a.g_loss = g_loss;
    a.argmin = argmin;
    for (int k = 0; k < S; ++k) { a.warped.p[k] = warped[k]; a.g_warped.p[k] = g_warped[k]; }
    a.g_disp = g_disp;
    a.S = S; a.flags = flags; a.B = B; a.H = H; a.W = W;
    a.tiles_x = (W + OW - 1) / OW; a.tiles_y = (H + OH - 1) / OH;
    a.smoothness = smoothness;
    {
        ProfScope ps(MVF_PROF_PHOTO_BWD, (hipStream_t)stream);
        launch_bwd_kernel<false>(a, dim3((unsigned)(a.tiles_x * a.tiles_y * B)), (hipStream_t)stream);
    }
    return hip_check_launch();
}

int mvf_unit_bwd(const float *disp, const float *tgt, const float *const *src, const float *T,
                 const float *K, const float *inv_K, const uint8_t *argmin, const float *mask_rec,
                 const float *stats, const float *g_loss, int S, int flags, float smoothness,
                 float min_disp, float range, float eps, float *g_disp, float *g_T,
                 float *workspace, int B, int H, int W, void *stream)
{
    if (S < 1 || S > MVF_MAX_SRC) return (int)hipErrorInvalidValue;
    if (B * H * W <= 0) return 0;
    BwdArgs a = {};
    a.disp = disp; a.tgt = tgt; a.mask = mask_rec; a.stats = stats; a.g_loss = g_loss;
    a.argmin = argmin; a.T = T; a.K = K; a.invK = inv_K;
    for (int k = 0; k < S; ++k) a.src.p[k] = src[k];
    a.g_disp = g_disp; a.ws = workspace;
    a.S = S; a.flags = flags; a.B = B; a.H = H; a.W = W;
    a.tiles_x = (W + OW - 1) / OW; a.tiles_y = (H + OH - 1) / OH;
    a.smoothness = smoothness; a.min_disp = min_disp; a.range = range; a.eps = eps;
    {
        ProfScope ps(MVF_PROF_UNIT_BWD, (hipStream_t)stream);
        launch_bwd_kernel<true>(a, dim3((unsigned)(a.tiles_x * a.tiles_y * B)), (hipStream_t)stream);
    }
    return mvf_geom::finish_gT(workspace, K, g_T, B, S, a.tiles_x * a.tiles_y, stream);
}

}  // extern "C"
