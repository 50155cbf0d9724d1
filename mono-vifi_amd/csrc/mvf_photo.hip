// mvf_photo.hip -- photometric half of the hot path for gfx950 (MI355X):
//   SSIM (layers.py:261-290), compute_reprojection_loss (train.py:973-985),
//   compute_losses_base (train.py:987-1051), get_smooth_loss (layers.py:231-242), and the
//   fully fused UNIT (S x generate_images_pred + compute_losses_base) whose warped images
//   live only in LDS.
//
// Tile engine.  A workgroup of 256 lanes owns a 64x16-pixel compute region.  Every image
// plane it needs is staged once into LDS as an 18-row x 68-float plane (1-px reflect halo,
// rows padded to a multiple of 16 B), and every lane then owns a 4-pixel row segment:
// its 3x3 windows are a few wide LDS reads per plane (18 values for 4 pixels instead of 36
// scalar reads), candidates are evaluated two at a time with packed fp32 math (see "packed
// fp32 pairs" below), and global traffic is the coalesced plane staging plus the L2-served
// bilinear taps.
//   forward : region = output tile (grid steps 64x16)
//   backward: region = where the SSIM adjoint coefficients are formed; outputs are its
//             62x14 interior (grid steps 62x14), so the 3x3 adjoint gather never leaves
//             the workgroup and needs no atomics.
// Bound: the exact-mode window arithmetic (no shared partial sums, true divides) makes
// these kernels VALU-bound below the HBM roofline -- see DESIGN.md section 5.
// Compiled with -ffp-contract=off (arithmetic contract in mvf_common.hpp).
#include "mvf_tile.hpp"

namespace {


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

// LDS carve (floats): target 3 planes | pair 3 planes of f2 | disparity 1 plane | pose | scratch
constexpr int FWD_TGT = 0, FWD_PAIR = 3 * PLANE, FWD_DISP = FWD_PAIR + 6 * PLANE,
              FWD_POSE = FWD_DISP + PLANE;

#ifndef MVF_FWD_WAVES
#define MVF_FWD_WAVES 3   // <=168 VGPRs: 3 waves/SIMD, matching the 3 workgroups/CU the 49 KB of LDS allow
#endif
#ifndef MVF_BWD_WAVES
#define MVF_BWD_WAVES 1
#endif
template <bool FUSED, int S>
__global__ void __launch_bounds__(NT, MVF_FWD_WAVES) k_photo_fwd(FwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tgtP = smem + FWD_TGT;
    f2 *pairP = reinterpret_cast<f2 *>(smem + FWD_PAIR);
    float *dispP = smem + FWD_DISP;
    PoseLds &sh = *reinterpret_cast<PoseLds *>(smem + FWD_POSE);
    float *scratch = smem + FWD_POSE + sizeof(PoseLds) / 4;

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
    // candidates: S warped sources, then (auto-masking) S identity sources; evaluated two at
    // a time, LAST pair first: a pair that comes straight from global planes (identity
    // sources; every pair of the staged kernel) is fetched together with the target and the
    // disparity in the prologue, and the fused warp (which needs the staged disparity) follows
    constexpr int NC = 2 * S;
    float val[NC][PX];
    const int ncand = automask ? 2 * S : S;
#ifdef MVF_ABL_ONEPAIR
    const int npair = 1;
#else
    const int npair = (ncand + 1) >> 1;
#endif
    bool prologue_pair = false;
    {
        const int ca = 2 * (npair - 1), cb = (ca + 1 < ncand) ? ca + 1 : ca;
        const bool wa = ca < S, wb = cb < S;
        const int ka = wa ? ca : ca - S, kb = wb ? cb : cb - S;
#ifdef MVF_ABL_NOWARP
        prologue_pair = true;
        const float *ima = a.src.p[ka], *imb = a.src.p[kb];
#else
        prologue_pair = !(FUSED && wa);
        const float *ima = FUSED ? a.src.p[ka] : (wa ? a.warped.p[ka] : a.src.p[ka]);
        const float *imb = FUSED ? a.src.p[kb] : (wb ? a.warped.p[kb] : a.src.p[kb]);
#endif
        if (prologue_pair) {
            stage_first(tgtP, dispP, pairP, a.tgt + (size_t)b * 3 * N, a.disp + (size_t)b * N,
                        ima + (size_t)b * 3 * N, imb + (size_t)b * 3 * N, N, H, W, py0, px0);
        } else {
            stage_planes3(tgtP, a.tgt + (size_t)b * 3 * N, N, H, W, py0, px0);
            stage_plane(dispP, a.disp + (size_t)b * N, H, W, py0, px0);
        }
    }
    __syncthreads();

    const int seg = threadIdx.x & (TW / PX - 1), row = threadIdx.x / (TW / PX);
    const int off = row * LDW + seg * PX;   // plane element of the window's top-left
    const int y = ty0 + row, x0 = tx0 + seg * PX;

#pragma unroll 1
    for (int pr = npair - 1; pr >= 0; --pr) {
        const int ca = 2 * pr, cb = (2 * pr + 1 < ncand) ? 2 * pr + 1 : 2 * pr;
        const bool wa = ca < S, wb = cb < S;              // warped (vs identity) candidates
        const int ka = wa ? ca : ca - S, kb = wb ? cb : cb - S;
        const bool staged = prologue_pair && pr == npair - 1;
        if (!staged && pr != npair - 1) __syncthreads();   // previous pair fully consumed
#ifdef MVF_ABL_NOWARP
        if (staged) {
        } else if (false) {
            if (wa) {
#else
        if (staged) {
        } else if (FUSED) {
            if (wa) {
#endif
                // lane 0 (and lane 1 when it is a warped source too) by the fused warp
                f2 P2[12];
                const int kb2 = wb ? kb : ka;
                load_pose_pair(sh, ka, kb2, P2);
                int32_t *ia = a.idx_xy ? a.idx_xy + ((size_t)ka * a.B + b) * N * 2 : nullptr;
                int32_t *ib = a.idx_xy ? a.idx_xy + ((size_t)kb2 * a.B + b) * N * 2 : nullptr;
                warp_pair_into_lds<1>(pairP, dispP, a.src.p[ka] + (size_t)b * 3 * N,
                                      a.src.p[kb2] + (size_t)b * 3 * N, a.invK + b * 16, P2, H, W, py0,
                                   px0, a.min_disp, a.range, a.eps, ia, ib, ty0, tx0);
                if (!wb) stage_lane3(pairP, 1, a.src.p[kb] + (size_t)b * 3 * N, N, H, W, py0, px0);
            } else {
                stage_pair3(pairP, a.src.p[ka] + (size_t)b * 3 * N, a.src.p[kb] + (size_t)b * 3 * N, N,
                            H, W, py0, px0);
            }
        } else {
#ifdef MVF_ABL_NOWARP
            const float *ima = a.src.p[ka] + (size_t)b * 3 * N;
            const float *imb = a.src.p[kb] + (size_t)b * 3 * N;
#else
            const float *ima = (wa ? a.warped.p[ka] : a.src.p[ka]) + (size_t)b * 3 * N;
            const float *imb = (wb ? a.warped.p[kb] : a.src.p[kb]) + (size_t)b * 3 * N;
#endif
            stage_pair3(pairP, ima, imb, N, H, W, py0, px0);
        }
        if (!staged) __syncthreads();
        f2 out[PX];
#ifdef MVF_ABL_NOREPROJ
        {
            f2 t = pairP[off + LDW + 1] + f2s(tgtP[off + LDW + 1]);
            for (int j = 0; j < PX; ++j) out[j] = t;
        }
#else
        reproj4p(pairP, tgtP, off, no_ssim, out);
#endif
#pragma unroll
        for (int cc = 0; cc < NC; ++cc)     // static register indices only
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                if (cc == ca) val[cc][j] = out[j].x;
                if (cc == cb && cb != ca) val[cc][j] = out[j].y;
            }
    }

    // ---- candidates, min / argmin, mask, outputs (reference: train.py:1010-1043)
    // rp[k] = val[k], idl[k] = val[S + k]
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
                float m = val[S][j];
#pragma unroll
                for (int k = 1; k < S; ++k) m = m + val[S + k][j];
                m = m / (float)S;
                best = m + a.noise[pi] * 0.00001f;
                nc = 1;
            } else {
#pragma unroll
                for (int k = 0; k < S; ++k) {
                    float v = val[S + k][j] + a.noise[((size_t)b * S + k) * N + pix] * 0.00001f;
                    if (nc == 0 || v < best) { best = v; bi = nc; }
                    ++nc;
                }
            }
        }
        if (avg) {
            float m = val[0][j];
#pragma unroll
            for (int k = 1; k < S; ++k) m = m + val[k][j];
            m = m / (float)S;
            if (nc == 0 || m < best) { best = m; bi = nc; }
            ++nc;
        } else {
#pragma unroll
            for (int k = 0; k < S; ++k) {
                float v = val[k][j];
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
#ifndef MVF_ABL_NOSMOOTH
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
#endif
        }
    }
    float *part = a.ws + (size_t)a.B * NMEAN +
                  (((size_t)b * a.tiles_y + tid.by) * a.tiles_x + tid.bx) * NPART;
    const float sums[4] = {photo, sx, sy, 0.0f};
    const float tot = block_sum_many<NT, 4>(sums, scratch);
    if (threadIdx.x < 4) part[threadIdx.x] = tot;
}

// per-image mean of disp: NMEAN partial sums per image, folded in fixed order by consumers
__global__ void __launch_bounds__(256) k_disp_mean(const float *__restrict__ disp,
                                                   float *__restrict__ ws, int N, size_t stride)
{
    __shared__ float scratch[4];
    int b = blockIdx.y, chunk = blockIdx.x;
    int per = (N + NMEAN - 1) / NMEAN;
    int lo = chunk * per, hi = min(lo + per, N);
    const float *d = disp + (size_t)b * stride;
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

// the same partials for up to MVF_MAX_UNITS disparity tensors in ONE launch (grid.z = tensor): the units of a
// batched launch each needed a k_disp_mean launch of their own when no disparity head supplied the partials
struct DispMeanMany {
    const float *disp[MVF_MAX_UNITS];
    size_t stride[MVF_MAX_UNITS];
    float *ws[MVF_MAX_UNITS];
};
__global__ void __launch_bounds__(256) k_disp_mean_many(DispMeanMany j, int N)
{
    __shared__ float scratch[4];
    const int b = blockIdx.y, chunk = blockIdx.x, u = blockIdx.z;
    const int per = (N + NMEAN - 1) / NMEAN;
    const int lo = chunk * per, hi = min(lo + per, N);
    const float *d = j.disp[u] + (size_t)b * j.stride[u];
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int i = lo + threadIdx.x;
    for (; i + 3 * 256 < hi; i += 4 * 256) {
        s0 += d[i];
        s1 += d[i + 256];
        s2 += d[i + 512];
        s3 += d[i + 768];
    }
    for (; i < hi; i += 256) s0 += d[i];
    const float r = block_sum<256>((s0 + s1) + (s2 + s3), scratch);
    if (threadIdx.x == 0) j.ws[u][b * NMEAN + chunk] = r;
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

// LDS carve (floats): target 3 planes | pair 3 f2 planes | disparity | coefficient 3 f2 region
// planes (A,B,G; reused to park grad_warped) | grad_disp accumulator region plane | pose | scratch
constexpr int BWD_TGT = 0, BWD_PAIR = 3 * PLANE, BWD_DISP = BWD_PAIR + 6 * PLANE,
              BWD_COEF = BWD_DISP + PLANE, BWD_GD = BWD_COEF + 6 * RPLANE, BWD_POSE = BWD_GD + RPLANE;

template <bool FUSED, int S>
__global__ void __launch_bounds__(NT, MVF_BWD_WAVES) k_photo_bwd(BwdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tgtP = smem + BWD_TGT;
    f2 *pairP = reinterpret_cast<f2 *>(smem + BWD_PAIR);
    float *dispP = smem + BWD_DISP;
    f2 *coefP = reinterpret_cast<f2 *>(smem + BWD_COEF);
    float *gdP = smem + BWD_GD;
    PoseLds &sh = *reinterpret_cast<PoseLds *>(smem + BWD_POSE);
    float *scratch = smem + BWD_POSE + sizeof(PoseLds) / 4;

    const TileId tid = tile_of_block(a.tiles_x, a.tiles_y, a.B);
    const int H = a.H, W = a.W, b = tid.b;
    const size_t N = (size_t)H * W;
    const int cy0 = tid.by * OH - 1, cx0 = tid.bx * OW - 1;   // region origin
    const int py0 = cy0 - 1, px0 = cx0 - 1;                   // plane origin
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
    const int seg = threadIdx.x & (TW / PX - 1), row = threadIdx.x / (TW / PX);
    const int off = row * LDW + seg * PX;
    const int roff = row * LDW + seg * PX;   // region-plane element of this lane's first pixel
    const int y = cy0 + row, x0 = cx0 + seg * PX;
    const bool rowin = (y >= 0) && (y < H);
    const bool row_out = (row >= 1) && (row <= OH) && rowin;   // interior (= output) rows

    // argmin and mask of every region pixel: issued with the prologue's plane loads (one
    // memory phase), weighted after the barrier
    float wbase[PX];
    int sel[PX];
#pragma unroll
    for (int j = 0; j < PX; ++j) {
        const int x = x0 + j;
        const bool in = rowin && (x >= 0) && (x < W);
        const size_t pi = (size_t)b * N + (size_t)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1);
        sel[j] = in ? (int)a.argmin[pi] : 254;
        float m = (in && a.mask) ? a.mask[pi] : 1.0f;
        wbase[j] = in ? m : 0.0f;
    }
    stage_planes3(tgtP, a.tgt + (size_t)b * 3 * N, N, H, W, py0, px0);
    stage_plane(dispP, a.disp + (size_t)b * N, H, W, py0, px0);
    for (int i = threadIdx.x; i < RPLANE; i += NT) gdP[i] = 0.0f;
    __syncthreads();
    // base weight of every region pixel: gpix * mask (0 outside the image)
#pragma unroll
    for (int j = 0; j < PX; ++j) wbase[j] = sh.gpix * wbase[j];

    const float myu = (y == 1) ? 2.0f : 1.0f, myd = (y == H - 2) ? 2.0f : 1.0f;

    constexpr int NPAIR = (S + 1) / 2;
#pragma unroll 1
    for (int pr = 0; pr < NPAIR; ++pr) {
        const int ka = 2 * pr;
        const bool hasb = 2 * pr + 1 < S;
        const int kb = hasb ? 2 * pr + 1 : ka;
        asm volatile("" ::: "memory");
        __syncthreads();
        f2 P2[12];
#ifdef MVF_ABL_NOWARPB
        load_pose_pair(sh, ka, kb, P2);
        if (false) {
#else
        if (FUSED) {
            load_pose_pair(sh, ka, kb, P2);
#endif
            warp_pair_into_lds<2>(pairP, dispP, a.src.p[ka] + (size_t)b * 3 * N,
                                  a.src.p[kb] + (size_t)b * 3 * N, a.invK + b * 16, P2, H, W, py0, px0,
                               a.min_disp, a.range, a.eps, nullptr, nullptr, 0, 0);
        } else {
#ifdef MVF_ABL_NOWARPB
            stage_pair3(pairP, a.src.p[ka] + (size_t)b * 3 * N, a.src.p[kb] + (size_t)b * 3 * N, N,
                        H, W, py0, px0);
#else
            stage_pair3(pairP, a.warped.p[ka] + (size_t)b * 3 * N, a.warped.p[kb] + (size_t)b * 3 * N, N,
                        H, W, py0, px0);
#endif
        }
        __syncthreads();

        // selection weights of the two sources: the argmin picked it (or the averaged channel)
        f2 wk[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            float wa, wb;
            if (sel[j] == 255) wa = wb = avg ? 1.0f / (float)S : 1.0f;        // single candidate
            else if (avg) wa = wb = (sel[j] == n_id) ? 1.0f / (float)S : 0.0f;
            else { wa = (sel[j] == n_id + ka) ? 1.0f : 0.0f; wb = (sel[j] == n_id + kb) ? 1.0f : 0.0f; }
            wk[j] = mk2(wbase[j] * wa, hasb ? wbase[j] * wb : 0.0f);
        }

#pragma unroll 1
        for (int c = 0; c < 3; ++c) {
            f2 gw[PX];
            const f2 *xp = pairP + c * PPLANE;
            const float *yp = tgtP + c * PLANE;
            f2 xq[PX];
            float yq[PX];
            if (no_ssim) {
                Row6P xr = load_row6p(xp + off + LDW);
                Row6 yr = load_row6(yp + off + LDW);
#pragma unroll
                for (int j = 0; j < PX; ++j) { xq[j] = xr.v[j + 1]; yq[j] = yr.v[j + 1]; }
#ifdef MVF_ABL_NOSTATS
            } else if (false) {
#else
            } else {
#endif
                Stats4P s;
                window_xp(xp + off, yp + off, s);
                f2 cA[PX], cB[PX], cG[PX];
#pragma unroll
                for (int j = 0; j < PX; ++j) {
                    f2 dmux, dexx, dexy;
                    f2 my = div9(s.sy[j]);     // (mu_y, E[y*y])
                    ssim_partials_pk(div9(s.sx[j]), f2s(my.x), div9(s.sxx[j]), f2s(my.y),
                                     div9(s.sxy[j]), dmux, dexx, dexy);
                    f2 g = wk[j] * ((0.85f / 3.0f) / 9.0f);
                    cA[j] = g * dmux;
                    cB[j] = g * 2.0f * dexx;
                    cG[j] = g * dexy;
                    xq[j] = s.xc[j];
                    yq[j] = s.yc[j];
                }
                float4 *cp = reinterpret_cast<float4 *>(coefP + roff);
                cp[0] = make_float4(cA[0].x, cA[0].y, cA[1].x, cA[1].y);
                cp[1] = make_float4(cA[2].x, cA[2].y, cA[3].x, cA[3].y);
                cp = reinterpret_cast<float4 *>(coefP + RPPLANE + roff);
                cp[0] = make_float4(cB[0].x, cB[0].y, cB[1].x, cB[1].y);
                cp[1] = make_float4(cB[2].x, cB[2].y, cB[3].x, cB[3].y);
                cp = reinterpret_cast<float4 *>(coefP + 2 * RPPLANE + roff);
                cp[0] = make_float4(cG[0].x, cG[0].y, cG[1].x, cG[1].y);
                cp[1] = make_float4(cG[2].x, cG[2].y, cG[3].x, cG[3].y);
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                // L1 term: d|t-p|/dp = -sign(t-p), channel mean
                f2 df = f2s(yq[j]) - xq[j];
                f2 sg = mk2((df.x > 0.0f) ? -1.0f : ((df.x < 0.0f) ? 1.0f : 0.0f),
                            (df.y > 0.0f) ? -1.0f : ((df.y < 0.0f) ? 1.0f : 0.0f));
                gw[j] = wk[j] * (no_ssim ? 1.0f : 0.15f) * sg / 3.0f;
            }
#ifdef MVF_ABL_NOGATHER
            if (false) {
#else
            if (!no_ssim && row >= 1 && row <= OH) {
#endif
                f2 sA[PX], sB[PX], sG[PX];
#pragma unroll
                for (int j = 0; j < PX; ++j) sA[j] = sB[j] = sG[j] = f2s(0.0f);
                const bool hasl = seg > 0, hasr = seg < TW / PX - 1;
#pragma unroll
                for (int dr = -1; dr <= 1; ++dr) {
                    const float my = (dr < 0) ? myu : ((dr > 0) ? myd : 1.0f);
                    // coefficient columns 4*seg-1 .. 4*seg+4 of region row (row+dr)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const f2 *cr = coefP + pl * RPPLANE + roff + dr * LDW;
                        const float4 *q = reinterpret_cast<const float4 *>(cr);
                        float4 m0 = q[0], m1 = q[1];
                        f2 v[6];
                        v[0] = hasl ? cr[-1] : f2s(0.0f);
                        v[1] = mk2(m0.x, m0.y); v[2] = mk2(m0.z, m0.w);
                        v[3] = mk2(m1.x, m1.y); v[4] = mk2(m1.z, m1.w);
                        v[5] = hasr ? cr[4] : f2s(0.0f);
#pragma unroll
                        for (int j = 0; j < PX; ++j) {
                            // reflect-pad multiplicity: column 0 is seen twice from column 1, ...
                            const float ml = (x0 + j == 1) ? 2.0f : 1.0f;
                            const float mr = (x0 + j == W - 2) ? 2.0f : 1.0f;
                            f2 t = my * (ml * v[j] + v[j + 1] + mr * v[j + 2]);
                            if (pl == 0) sA[j] += t;
                            else if (pl == 1) sB[j] += t;
                            else sG[j] += t;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < PX; ++j) gw[j] += sA[j] + xq[j] * sB[j] + f2s(yq[j]) * sG[j];
            }
            // park grad_warped of this channel in its own pair plane: every window read of
            // the plane finished before the barrier above, and each lane only touches the
            // entries of its own 4 pixels
            {
                f2 *gp = pairP + c * PPLANE + (row + 1) * LDW + seg * PX + 1;
#pragma unroll
                for (int j = 0; j < PX; ++j) gp[j] = gw[j];
            }
            __syncthreads();   // coefficient planes free for the next channel
        }

        // ---- consume grad_warped of this source pair
        if (!FUSED) {
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const int x = x0 + j, col = seg * PX + j;
                const bool outp = row_out && (col >= 1) && (col <= OW) && (x >= 0) && (x < W);
                if (!outp) continue;
                const size_t pi = (size_t)y * W + x;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f2 g = pairP[c * PPLANE + (row + 1) * LDW + col + 1];
                    a.g_warped.p[ka][((size_t)b * 3 + c) * N + pi] = g.x;
                    if (hasb) a.g_warped.p[kb][((size_t)b * 3 + c) * N + pi] = g.y;
                }
            }
        } else {
            // walk this lane's pixels one at a time through the projection adjoint (keeps
            // the live set small); grad_warped comes back from the pair planes
            const float *sa = a.src.p[ka] + (size_t)b * 3 * N, *sb = a.src.p[kb] + (size_t)b * 3 * N;
            f2 accP[12];
#pragma unroll
            for (int q = 0; q < 12; ++q) accP[q] = f2s(0.0f);
#pragma unroll 4
            for (int j = 0; j < PX; ++j) {
#ifdef MVF_ABL_NOCHAINB
                continue;
#endif
                const int x = x0 + j, col = seg * PX + j;
                const bool outp = row_out && (col >= 1) && (col <= OW) && (x >= 0) && (x < W);
                if (!outp) continue;
                const f2 *gp = pairP + (row + 1) * LDW + col + 1;
                const f2 g0 = gp[0], g1 = gp[PPLANE], g2 = gp[2 * PPLANE];
                WarpPair w = warp_point_pair(dispP[(row + 1) * LDW + col + 1], a.invK + b * 16, P2, x, y,
                                             H, W, a.min_disp, a.range, a.eps);
                float dxa[3], dya[3], dxb[3], dyb[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
#ifdef MVF_ABL_CHAINLDS   // ablation: adjoint taps from an LDS plane instead of global memory
                    const float *la = tgtP + c * PLANE + (row + 1) * LDW + col;
                    dxa[c] = (la[1] - la[0]) * w.ta.wy + (la[LDW + 1] - la[LDW]);
                    dya[c] = (la[LDW] - la[0]) * w.ta.wx + (la[LDW + 1] - la[1]);
                    dxb[c] = (la[2] - la[1]) * w.tb.wy + (la[LDW + 2] - la[LDW + 1]);
                    dyb[c] = (la[LDW + 1] - la[1]) * w.tb.wx + (la[LDW + 2] - la[2]);
#elif defined(MVF_ABL_CHAINNOTAP)
                    dxa[c] = w.ta.wx; dya[c] = w.ta.wy; dxb[c] = w.tb.wx; dyb[c] = w.tb.wy;
#else
                    bilerp_grad(sa + c * N, W, w.ta, dxa[c], dya[c]);
                    bilerp_grad(sb + c * N, W, w.tb, dxb[c], dyb[c]);
#endif
                }
                f2 gix = g0 * mk2(dxa[0], dxb[0]) + g1 * mk2(dxa[1], dxb[1]) + g2 * mk2(dxa[2], dxb[2]);
                f2 giy = g0 * mk2(dya[0], dyb[0]) + g1 * mk2(dya[1], dyb[1]) + g2 * mk2(dya[2], dyb[2]);
                // adjoint of unnormalise / normalise / perspective divide (see warp_point_bwd)
                const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
                f2 ggx = mk2(w.ta.inx ? gix.x * (wm1 / 2.0f) : 0.0f, w.tb.inx ? gix.y * (wm1 / 2.0f) : 0.0f);
                f2 ggy = mk2(w.ta.iny ? giy.x * (hm1 / 2.0f) : 0.0f, w.tb.iny ? giy.y * (hm1 / 2.0f) : 0.0f);
                f2 gu = ggx * 2.0f / wm1, gv = ggy * 2.0f / hm1;
                f2 gc[3];
                gc[0] = gu / w.z;
                gc[1] = gv / w.z;
                gc[2] = -(gu * w.u + gv * w.v) / w.z;
                f2 gd = f2s(0.0f);
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    f2 gX = gc[0] * P2[0 * 4 + jj] + gc[1] * P2[1 * 4 + jj] + gc[2] * P2[2 * 4 + jj];
                    gd += gX * w.r[jj];
                }
                const float gdsum = hasb ? gd.x + gd.y : gd.x;
                gdP[roff + j] += -gdsum * w.depth * w.depth * a.range;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    accP[q * 4 + 0] += gc[q] * w.X[0];
                    accP[q * 4 + 1] += gc[q] * w.X[1];
                    accP[q * 4 + 2] += gc[q] * w.X[2];
                    accP[q * 4 + 3] += gc[q];
                }
            }
            // per-tile partials of grad_P: ws[((k*B + b)*ntiles + tile)*12 + q]
            const size_t ntiles = (size_t)a.tiles_x * a.tiles_y;
            const size_t tile = (size_t)tid.by * a.tiles_x + tid.bx;
            float *parta = a.ws + (((size_t)ka * a.B + b) * ntiles + tile) * 12;
            float *partb = a.ws + (((size_t)kb * a.B + b) * ntiles + tile) * 12;
            float flat[24];
#pragma unroll
            for (int q = 0; q < 12; ++q) { flat[q] = accP[q].x; flat[12 + q] = accP[q].y; }
#ifdef MVF_ABL_NOBSUM
            const float tot = flat[threadIdx.x % 24];
#else
            const float tot = block_sum_many<NT, 24>(flat, scratch);
#endif
            if (threadIdx.x < 12) parta[threadIdx.x] = tot;
            else if (threadIdx.x < 24 && hasb) partb[threadIdx.x - 12] = tot;
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
        const float rden = 1.0f / den, corr_den = corr / den;
#pragma unroll 1
        for (int j = 0; j < PX; ++j) {
            const int x = x0 + j;
            const int col = seg * PX + j;
            const bool outp = row_out && (col >= 1) && (col <= OW) && (x >= 0) && (x < W);
            if (!outp) continue;
            const float *dc = dispP + (row + 1) * LDW + col + 1;
            const float *t0 = tgtP + (row + 1) * LDW + col + 1;
            // only the SIGN of differences of normalised disparities is needed: dividing by
            // the positive per-image constant cannot change it, so compare the raw values
            float nd = dc[0];
            float gn = 0.0f;
            auto wgt = [&](int o) {
                float gi = div3((fabsf(t0[0] - t0[o]) + fabsf(t0[PLANE] - t0[PLANE + o])) +
                                fabsf(t0[2 * PLANE] - t0[2 * PLANE + o]));
                return expf(-gi);
            };
            auto sgn = [](float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); };
#ifndef MVF_ABL_NOSMOOTHB
            if (x + 1 < W) {
                gn += cx * wgt(1) * sgn(nd - dc[1]);
            }
            if (x - 1 >= 0) gn -= cx * wgt(-1) * sgn(dc[-1] - nd);
            if (y + 1 < H) {
                gn += cy * wgt(LDW) * sgn(nd - dc[LDW]);
            }
            if (y - 1 >= 0) gn -= cy * wgt(-LDW) * sgn(dc[-LDW] - nd);
#endif
            a.g_disp[(size_t)b * N + (size_t)y * W + x] = gdP[roff + j] + gn * rden - corr_den;
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

inline size_t fwd_smem() { return FWD_POSE * sizeof(float) + sizeof(PoseLds) + 128 * sizeof(float); }
inline size_t bwd_smem() { return BWD_POSE * sizeof(float) + sizeof(PoseLds) + 24 * (NT / 16) * sizeof(float); }

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
    hipLaunchKernelGGL(k_disp_mean, dim3(NMEAN, a.B), dim3(256), 0, st, a.disp, a.ws, N, (size_t)N);
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

// shared with mvf_unit_fb.hip
namespace mvf_photo {
struct DispMeanJobs {
    const float *disp[MVF_MAX_UNITS];
    size_t stride[MVF_MAX_UNITS];
    float *ws[MVF_MAX_UNITS];
    int n;
};
void launch_disp_mean_many(const DispMeanJobs &jobs, int B, int N, hipStream_t st)
{
    DispMeanMany j = {};
    for (int i = 0; i < jobs.n; ++i) { j.disp[i] = jobs.disp[i]; j.stride[i] = jobs.stride[i]; j.ws[i] = jobs.ws[i]; }
    ProfScope ps(MVF_PROF_DISP_MEAN, st, 4LL * jobs.n * B * N);
    hipLaunchKernelGGL(k_disp_mean_many, dim3(NMEAN, B, jobs.n), dim3(256), 0, st, j, N);
}
void launch_disp_mean(const float *disp, size_t image_stride, float *ws, int B, int N, hipStream_t st)
{
    ProfScope ps(MVF_PROF_DISP_MEAN, st, 4LL * B * N);
    hipLaunchKernelGGL(k_disp_mean, dim3(NMEAN, B), dim3(256), 0, st, disp, ws, N, image_stride);
}
}  // namespace mvf_photo

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
    hipLaunchKernelGGL(k_disp_mean, dim3(NMEAN, B), dim3(256), 0, st, disp, workspace, H * W, (size_t)H * W);
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
