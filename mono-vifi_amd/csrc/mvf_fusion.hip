// mvf_fusion.hip -- SURVEY.md section 8f-1: the tensor that enters each 1x1 convolution of the
// multi-frame FusionModule (reference: networks/fusion_module.py:65-130), produced in ONE launch
// per pyramid level instead of ~40 ATen kernels (60 sin/cos launches, three F.interpolate, eight
// cat copies, the mask merge):
//
//   out[b] = cat[ feat_0 ‖ emb(0) ‖ m * (warp(feat_n1, fl_n1) ‖ emb(e_n1)) + (1-m) * (warp(feat_p1, fl_p1) ‖ emb(e_p1)) ]
//
//   fl_*  = F.interpolate(flow, size=(h,w), bilinear, align_corners=False) * (w/W, h/H)   (warp_features, :80-90)
//   e_*   = the cascaded half-resolution flow, halved at each step                        (get_embedding_flow, :65-78)
//   m     = F.interpolate(merge_mask, size=(h,w))                                         (merge_features, :92-103)
//   emb   = [x, sin(2^k x), cos(2^k x)], k = 0..9 (42 channels)                           (Embedder, :7-37)
//
// A small "prep" kernel per level evaluates the three resizes (the cascade depends on the
// previous level) into a 9-plane side tensor; the level kernel then streams: every lane owns a
// pixel, computes its two flow taps once per 8-channel chunk (paired 8-byte tap loads), and one
// extra chunk writes the 84 embedding channels.  Backward: features need gradients, flows and
// mask come from the frozen teacher: grad_feat_0 is a view of grad_out, the two warps scatter
// m * g and (1-m) * g through the shared flow-warp scatter (mvf_common.hpp).
// Arithmetic: fp32, expression order of the reference (m*a + (1-m)*b as two products and a sum);
// sinf/cosf with full range reduction (arguments reach 2^9 * |flow|).
#include "mvf_common.hpp"

using namespace mvf;

namespace {

constexpr int NT = 256;
constexpr int CCH = 8;           // feature channels per lane
constexpr int NFREQ = 10;
constexpr int EMB = 2 * (1 + 2 * NFREQ);   // 42
// prep planes: 0-1 e_n1, 2-3 e_p1, 4-5 fl_n1, 6-7 fl_p1, 8 mask
constexpr int PREP = 9;

// ATen's bilinear source index for align_corners=False: src = scale*(dst+0.5)-0.5, clamped at 0
struct Lin {
    int i0, i1;
    float w0, w1;
};
MVF_DEV Lin lin_of(int dst, float scale, int n_in)
{
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    Lin l;
    l.i0 = min((int)src, n_in - 1);
    l.i1 = min(l.i0 + 1, n_in - 1);
    l.w1 = src - (float)l.i0;
    l.w0 = 1.0f - l.w1;
    return l;
}
MVF_DEV float lerp2(const float *__restrict__ p, int W, const Lin &ly, const Lin &lx)
{
    const float top = lx.w0 * p[(size_t)ly.i0 * W + lx.i0] + lx.w1 * p[(size_t)ly.i0 * W + lx.i1];
    const float bot = lx.w0 * p[(size_t)ly.i1 * W + lx.i0] + lx.w1 * p[(size_t)ly.i1 * W + lx.i1];
    return ly.w0 * top + ly.w1 * bot;
}

// one level of the side tensor.  halvings: how many scale-0.5 steps lead from `esrc`
// ([B,2,eh,ew]: the full-resolution flow for the first level, the previous level's e planes
// otherwise) to this level (2 for Lite-Mono's first level, else 1).
__global__ void __launch_bounds__(NT) k_fusion_prep(const float *__restrict__ flow_n1,
                                                    const float *__restrict__ flow_p1,
                                                    const float *__restrict__ mask,
                                                    const float *__restrict__ esrc_n1,
                                                    const float *__restrict__ esrc_p1, size_t esrc_bstride,
                                                    float *__restrict__ prep, int h, int w, int Hf, int Wf,
                                                    int eh, int ew, int halvings)
{
    const int n = h * w, b = blockIdx.y;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const int y = i / w, x = i - y * w;
    float *o = prep + (size_t)b * PREP * n + i;
    // ---- direct resizes from full resolution (flows scaled with the size; mask)
    {
        const Lin ly = lin_of(y, (float)Hf / (float)h, Hf), lx = lin_of(x, (float)Wf / (float)w, Wf);
        const size_t Nf = (size_t)Hf * Wf;
        const float sx = (float)((double)w / (double)Wf), sy = (float)((double)h / (double)Hf);
        o[4 * (size_t)n] = lerp2(flow_n1 + ((size_t)b * 2 + 0) * Nf, Wf, ly, lx) * sx;
        o[5 * (size_t)n] = lerp2(flow_n1 + ((size_t)b * 2 + 1) * Nf, Wf, ly, lx) * sy;
        o[6 * (size_t)n] = lerp2(flow_p1 + ((size_t)b * 2 + 0) * Nf, Wf, ly, lx) * sx;
        o[7 * (size_t)n] = lerp2(flow_p1 + ((size_t)b * 2 + 1) * Nf, Wf, ly, lx) * sy;
        o[8 * (size_t)n] = lerp2(mask + (size_t)b * Nf, Wf, ly, lx);
    }
    // ---- cascaded half resolution: x <- 0.5 * interpolate(x, scale_factor=0.5), `halvings` times
    const size_t Ne = (size_t)eh * ew;
#pragma unroll
    for (int f = 0; f < 2; ++f) {
        const float *src = (f == 0 ? esrc_n1 : esrc_p1) + (size_t)b * esrc_bstride;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float *p = src + (size_t)c * Ne;
            float v;
            if (halvings == 1) {
                v = lerp2(p, ew, lin_of(y, 2.0f, eh), lin_of(x, 2.0f, ew)) * 0.5f;
            } else {
                // two steps: this pixel of the second step reads a 2x2 block of the first step
                const int mh = eh / 2, mw = ew / 2;
                const Lin ly = lin_of(y, 2.0f, mh), lx = lin_of(x, 2.0f, mw);
                float q[2][2];
#pragma unroll
                for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 2; ++dx)
                        q[dy][dx] = lerp2(p, ew, lin_of(dy ? ly.i1 : ly.i0, 2.0f, eh),
                                          lin_of(dx ? lx.i1 : lx.i0, 2.0f, ew)) * 0.5f;
                const float top = lx.w0 * q[0][0] + lx.w1 * q[0][1], bot = lx.w0 * q[1][0] + lx.w1 * q[1][1];
                v = (ly.w0 * top + ly.w1 * bot) * 0.5f;
            }
            o[(size_t)(2 * f + c) * n] = v;
        }
    }
}

// grid (pixel blocks, nchunk + 1, B): chunk < nchunk -> 8 feature channels (copy of feat_0 and
// the merged warps); chunk == nchunk -> the 84 embedding channels.
__global__ void __launch_bounds__(NT) k_fusion_level_fwd(const float *__restrict__ f0,
                                                         const float *__restrict__ fn1,
                                                         const float *__restrict__ fp1,
                                                         const float *__restrict__ prep,
                                                         const float *__restrict__ xs,
                                                         const float *__restrict__ ys,
                                                         float *__restrict__ out, int C, int h, int w)
{
    const int n = h * w, b = blockIdx.z;
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= n) return;
    const int y = i / w, x = i - y * w;
    const float *pp = prep + (size_t)b * PREP * n + i;
    const int CT = 2 * (C + EMB);
    float *ob = out + (size_t)b * CT * n + i;
    const float m = pp[8 * (size_t)n], om = 1.0f - m;
    const int nchunk = (C + CCH - 1) / CCH;
    if ((int)blockIdx.y < nchunk) {
        const Tap tn = flow_tap_xy(pp[4 * (size_t)n], pp[5 * (size_t)n], xs, ys, x, y, h, w);
        const Tap tp = flow_tap_xy(pp[6 * (size_t)n], pp[7 * (size_t)n], xs, ys, x, y, h, w);
        const int c0 = blockIdx.y * CCH, c1 = min(c0 + CCH, C);
        for (int c = c0; c < c1; ++c) {
            const size_t pl = ((size_t)b * C + c) * n;
            ob[(size_t)c * n] = f0[pl + i];
            const float a = bilerp(fn1 + pl, w, tn), q = bilerp(fp1 + pl, w, tp);
            ob[(size_t)(C + EMB + c) * n] = m * a + om * q;
        }
        return;
    }
    // ---- embedding channels: [x, sin(2^k x), cos(2^k x)] of the (zero | n1 | p1) flows
    const float en[2] = {pp[0], pp[(size_t)n]}, ep[2] = {pp[2 * (size_t)n], pp[3 * (size_t)n]};
    float *o0 = ob + (size_t)C * n;                 // emb(0)
    float *o1 = ob + (size_t)(2 * C + EMB) * n;     // merged
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        o0[(size_t)d * n] = 0.0f;
        o1[(size_t)d * n] = m * en[d] + om * ep[d];
    }
    float fr = 1.0f;
    for (int k = 0; k < NFREQ; ++k, fr *= 2.0f) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            float sn, cn, sp, cp;
            sincosf(en[d] * fr, &sn, &cn);
            sincosf(ep[d] * fr, &sp, &cp);
            const size_t cs = (size_t)(2 + 4 * k + d) * n, cc = (size_t)(2 + 4 * k + 2 + d) * n;
            o0[cs] = 0.0f;
            o0[cc] = 1.0f;
            o1[cs] = m * sn + om * sp;
            o1[cc] = m * cn + om * cp;
        }
    }
}

// grad_feat_n1 += scatter(m * g), grad_feat_p1 += scatter((1-m) * g), g = g_out[:, C+42 : 2C+42]
// grid (pixel blocks, nchunk, B); blockDim NT (whole wavefronts: the scatter hands taps over
// between neighbouring lanes)
__global__ void __launch_bounds__(NT) k_fusion_level_bwd(const float *__restrict__ g_out,
                                                         const float *__restrict__ prep,
                                                         const float *__restrict__ xs,
                                                         const float *__restrict__ ys,
                                                         float *__restrict__ g_fn1, float *__restrict__ g_fp1,
                                                         int C, int h, int w)
{
    const int n = h * w, b = blockIdx.z;
    const int i_raw = blockIdx.x * NT + threadIdx.x;
    const bool active = i_raw < n;
    const int i = active ? i_raw : n - 1;
    const int y = i / w, x = i - y * w;
    const float *pp = prep + (size_t)b * PREP * n + i;
    const float m = pp[8 * (size_t)n], om = 1.0f - m;
    const Tap tn = flow_tap_xy(pp[4 * (size_t)n], pp[5 * (size_t)n], xs, ys, x, y, h, w);
    const Tap tp = flow_tap_xy(pp[6 * (size_t)n], pp[7 * (size_t)n], xs, ys, x, y, h, w);
    const ScatterLinks ln = scatter_links(tn, active), lp = scatter_links(tp, active);
    const int CT = 2 * (C + EMB);
    const float *gb = g_out + ((size_t)b * CT + C + EMB) * n + i;
    const int c0 = blockIdx.y * CCH, c1 = min(c0 + CCH, C);
    for (int c = c0; c < c1; ++c) {
        const float g = active ? gb[(size_t)c * n] : 0.0f;
        const size_t pl = ((size_t)b * C + c) * n;
        if (g_fn1) scatter_taps_linked(g_fn1 + pl, w, tn, m * g, active, ln);
        if (g_fp1) scatter_taps_linked(g_fp1 + pl, w, tp, om * g, active, lp);
    }
}

}  // namespace

extern "C" {

size_t mvf_fusion_prep_floats(int B, int h, int w) { return (size_t)B * PREP * h * w; }

int mvf_fusion_prep(const float *flow_n1, const float *flow_p1, const float *mask, const float *prev_prep,
                    float *prep, int B, int h, int w, int Hf, int Wf, int prev_h, int prev_w, int halvings,
                    void *stream)
{
    if (B <= 0 || h <= 0 || w <= 0) return 0;
    if (!flow_n1 || !flow_p1 || !mask || !prep || (halvings != 1 && halvings != 2)) return (int)hipErrorInvalidValue;
    // source of the cascade: the previous level's e planes, or the full-resolution flows
    const float *e_n1 = prev_prep ? prev_prep : flow_n1, *e_p1 = prev_prep ? prev_prep + 2 * (size_t)prev_h * prev_w : flow_p1;
    const int eh = prev_prep ? prev_h : Hf, ew = prev_prep ? prev_w : Wf;
    const size_t bstride = prev_prep ? (size_t)PREP * prev_h * prev_w : (size_t)2 * Hf * Wf;
    hipLaunchKernelGGL(k_fusion_prep, dim3((unsigned)((h * w + NT - 1) / NT), (unsigned)B), dim3(NT), 0,
                       (hipStream_t)stream, flow_n1, flow_p1, mask, e_n1, e_p1, bstride, prep, h, w, Hf, Wf, eh, ew,
                       halvings);
    return hip_check_launch();
}

int mvf_fusion_level_fwd(const float *feat_0, const float *feat_n1, const float *feat_p1, const float *prep,
                         const float *xs, const float *ys, float *out, int B, int C, int h, int w, void *stream)
{
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
    if (!feat_0 || !feat_n1 || !feat_p1 || !prep || !xs || !ys || !out || h < 2 || w < 2)
        return (int)hipErrorInvalidValue;
    const int nchunk = (C + CCH - 1) / CCH;
    hipLaunchKernelGGL(k_fusion_level_fwd, dim3((unsigned)((h * w + NT - 1) / NT), (unsigned)(nchunk + 1), (unsigned)B),
                       dim3(NT), 0, (hipStream_t)stream, feat_0, feat_n1, feat_p1, prep, xs, ys, out, C, h, w);
    return hip_check_launch();
}

int mvf_fusion_level_bwd(const float *g_out, const float *prep, const float *xs, const float *ys, float *g_feat_n1,
                         float *g_feat_p1, int B, int C, int h, int w, void *stream)
{
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
    if (!g_out || !prep || !xs || !ys) return (int)hipErrorInvalidValue;
    if (!g_feat_n1 && !g_feat_p1) return 0;
    const int nchunk = (C + CCH - 1) / CCH;
    hipLaunchKernelGGL(k_fusion_level_bwd, dim3((unsigned)((h * w + NT - 1) / NT), (unsigned)nchunk, (unsigned)B),
                       dim3(NT), 0, (hipStream_t)stream, g_out, prep, xs, ys, g_feat_n1, g_feat_p1, C, h, w);
    return hip_check_launch();
}

}  // extern "C"
