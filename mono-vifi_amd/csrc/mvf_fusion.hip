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
#ifndef MVF_GATHER_CH
#define MVF_GATHER_CH 16
#endif
constexpr int GCH = MVF_GATHER_CH;   // channels per lane of the inverse-list gathers: a cell's list entries are read once per GCH channels
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
#ifdef MVF_ABL_FUS_NOGATHER      // timing ablation: the two warps read their own pixel (streaming loads instead of tap gathers)
            const float a = fn1[pl + i], q = fp1[pl + i];
#else
            const float a = bilerp(fn1 + pl, w, tn), q = bilerp(fp1 + pl, w, tp);
#endif
            ob[(size_t)(C + EMB + c) * n] = m * a + om * q;
        }
        return;
    }
#ifdef MVF_ABL_FUS_NOEMB         // timing ablation: the embedding channels are not written
    return;
#endif
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

// ---- deterministic backward: the scatter turned into a gather through an inverse tap list --------
// grad_feat[b,c,q] = sum over output pixels p whose bilinear taps touch q of  w(p->q) * m(p) * g[b,c,p].
// Which p touch q depends only on the flow (not on the channel), so the inverse map is built once
// per level and source -- a CSR list per destination pixel: count (integer atomics: order-free),
// exclusive scan, fill, then each (short) list is sorted by p -- and every channel then GATHERS
// through it in that fixed order: no float atomics, bit-reproducible, and the 64..512 channels of
// a level read g the way the forward warp reads its features.
// ws ints per level: cnt [2][B][N] | off [2][B][N+1] | entries (p, w bits) [2][B][4N][2]
struct InvWs {
    int *cnt, *off;
    int2 *ent;
};
MVF_DEV InvWs inv_ws(int *ws, int B, int N)
{
    InvWs r;
    r.cnt = ws;
    r.off = ws + (size_t)2 * B * N;
    r.ent = reinterpret_cast<int2 *>(r.off + (size_t)2 * B * (N + 1));
    return r;
}
// the (up to) four destination cells of output pixel (b, i) for source s, zero-weight taps dropped
struct Dest {
    int cell[4];
    float w[4];
    int n;
};
// FLOW: `prep` is a plain flow field [B,2,h,w] (standalone IFRNet.warp: one source, weight 1)
template <bool FLOW>
MVF_DEV Dest dest_of(const float *__restrict__ prep, const float *__restrict__ xs, const float *__restrict__ ys,
                     int b, int s, int i, int h, int w)
{
    const int n = h * w, y = i / w, x = i - y * w;
    const float *pp = prep + (size_t)b * (FLOW ? 2 : PREP) * n + i;
    const Tap t = FLOW ? flow_tap_xy(pp[0], pp[n], xs, ys, x, y, h, w)
                       : flow_tap_xy(pp[(size_t)(4 + 2 * s) * n], pp[(size_t)(5 + 2 * s) * n], xs, ys, x, y, h, w);
    const float m = FLOW ? 1.0f : pp[8 * (size_t)n];
    const float sc = (s == 0) ? m : 1.0f - m;            // merge weight of this source
    const float fw = t.wx, fe = 1.0f - fw, fn = t.wy, fs = 1.0f - fn;
    const float wt[4] = {fs * fe, fs * fw, fn * fe, fn * fw};
    const int cl[4] = {t.y0 * w + t.x0, t.y0 * w + t.x1, t.y1 * w + t.x0, t.y1 * w + t.x1};
    Dest d;
    d.n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // border clamping can map two taps to one cell (x1 == x0 at the right edge): the clamped
        // tap always carries weight 0 there and is dropped, so a pixel lists a cell at most once
        if (wt[k] != 0.0f) { d.cell[d.n] = cl[k]; d.w[d.n] = wt[k] * sc; ++d.n; }
    }
    return d;
}

// grid (pixel blocks, B, 2 sources | 1 for FLOW)
template <bool FLOW>
__global__ void __launch_bounds__(NT) k_inv_count(const float *__restrict__ prep, const float *__restrict__ xs,
                                                  const float *__restrict__ ys, int *__restrict__ ws, int B, int h, int w)
{
    const int n = h * w, i = blockIdx.x * NT + threadIdx.x, b = blockIdx.y, s = blockIdx.z;
    if (i >= n) return;
    const InvWs W = inv_ws(ws, B, n);
    const Dest d = dest_of<FLOW>(prep, xs, ys, b, s, i, h, w);
    int *cnt = W.cnt + ((size_t)s * B + b) * n;
    for (int k = 0; k < d.n; ++k) atomicAdd(cnt + d.cell[k], 1);
}

// exclusive scan of the counts of one (source, image) per block; off has N+1 entries.  Every lane
// owns a contiguous run of cells (local sums), ONE block-level scan of the 1024 run totals follows
__global__ void __launch_bounds__(1024) k_inv_scan(int *__restrict__ ws, int B, int n)
{
    __shared__ int sh[1024];
    const InvWs W = inv_ws(ws, B, n);
    const int *cnt = W.cnt + (size_t)blockIdx.x * n;
    int *off = W.off + (size_t)blockIdx.x * (n + 1);
    const int per = (n + 1023) / 1024;
    const int lo = min((int)threadIdx.x * per, n), hi = min(lo + per, n);
    int tot = 0;
    for (int i = lo; i < hi; ++i) tot += cnt[i];
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                   // Hillis-Steele inclusive scan of the run totals
        const int t = (threadIdx.x >= d) ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    int run = sh[threadIdx.x] - tot;                        // exclusive prefix of this lane's run
    for (int i = lo; i < hi; ++i) {
        off[i] = run;
        run += cnt[i];
    }
    if (threadIdx.x == 1023) off[n] = sh[1023];
}

template <bool FLOW>
__global__ void __launch_bounds__(NT) k_inv_fill(const float *__restrict__ prep, const float *__restrict__ xs,
                                                 const float *__restrict__ ys, int *__restrict__ ws, int B, int h, int w)
{
    const int n = h * w, i = blockIdx.x * NT + threadIdx.x, b = blockIdx.y, s = blockIdx.z;
    if (i >= n) return;
    const InvWs W = inv_ws(ws, B, n);
    const Dest d = dest_of<FLOW>(prep, xs, ys, b, s, i, h, w);
    const size_t sb = (size_t)s * B + b;
    int *cur = W.cnt + sb * n;                              // zeroed again by the caller: the cursor
    const int *off = W.off + sb * (n + 1);
    int2 *ent = W.ent + sb * 4 * n;
    for (int k = 0; k < d.n; ++k) {
        const int slot = off[d.cell[k]] + atomicAdd(cur + d.cell[k], 1);
        ent[slot] = make_int2(i, __float_as_int(d.w[k]));
    }
}

// every destination cell sorts its list by source pixel (insertion sort; lists average 4 entries)
__global__ void __launch_bounds__(NT) k_inv_sort(int *__restrict__ ws, int B, int n)
{
    const int q = blockIdx.x * NT + threadIdx.x;
    if (q >= n) return;
    const InvWs W = inv_ws(ws, B, n);
    const size_t sb = blockIdx.y;
    const int *off = W.off + sb * (n + 1);
    int2 *ent = W.ent + sb * 4 * n;
    const int lo = off[q], hi = off[q + 1];
    for (int a = lo + 1; a < hi; ++a) {
        const int2 v = ent[a];
        int c = a - 1;
        while (c >= lo && ent[c].x > v.x) { ent[c + 1] = ent[c]; --c; }
        ent[c + 1] = v;
    }
}

// One destination cell's (sorted) list applied to CCH channel planes: acc[k] += w_e * g[k][p_e] in list order.
// Round 4: the first GE entries of the list are fetched up front (clamped index; lists average four entries) and
// the channel loads of ALL of them are in flight before the first add -- the entry-at-a-time loop put two
// dependent memory latencies (entry, then its channel values) behind every list element: 0.12 of the HBM peak.
// Entries beyond the list's length are predicated off (never multiplied in: a 0 x inf would poison the sum);
// a list longer than GE finishes in the serial tail.  Same additions in the same order as before.
constexpr int GE = 6;
template <int NCH>
MVF_DEV void gather_list(const float *__restrict__ gb, size_t n, const int2 *__restrict__ ent, int lo, int hi, int nc,
                         float (&acc)[NCH])
{
    const int cnt = hi - lo, last = max(hi, 1) - 1;
    int2 e[GE];
#pragma unroll
    for (int j = 0; j < GE; ++j) e[j] = ent[min(lo + j, last)];
#pragma unroll
    for (int h0 = 0; h0 < NCH; h0 += 4) {
        if (h0 >= nc) break;            // (block-uniform: a chunk narrower than NCH, e.g. a 3-channel image)
        float v[GE][4];
#pragma unroll
        for (int j = 0; j < GE; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[j][k] = gb[(size_t)min(h0 + k, nc - 1) * n + e[j].x];
#pragma unroll
        for (int j = 0; j < GE; ++j)
            if (j < cnt) {
                const float wgt = __int_as_float(e[j].y);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[h0 + k] += wgt * v[j][k];
            }
    }
    for (int q = lo + GE; q < hi; ++q) {
        const int2 v = ent[q];
        const float wgt = __int_as_float(v.y);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
            if (k < nc) acc[k] += wgt * gb[(size_t)k * n + v.x];
    }
}

// grid (pixel blocks, channel chunks, B): both sources per lane
__global__ void __launch_bounds__(NT) k_fusion_level_bwd_gather(const float *__restrict__ g_out,
                                                                const int *__restrict__ ws,
                                                                float *__restrict__ g_fn1,
                                                                float *__restrict__ g_fp1, int B, int C, int h, int w)
{
    const int n = h * w, q = blockIdx.x * NT + threadIdx.x, b = blockIdx.z;
    if (q >= n) return;
    const InvWs W = inv_ws(const_cast<int *>(ws), B, n);
    const int CT = 2 * (C + EMB);
    const int c0 = blockIdx.y * GCH, nc = min(GCH, C - c0);
    const float *gb = g_out + ((size_t)b * CT + C + EMB + c0) * n;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float *dst = (s == 0) ? g_fn1 : g_fp1;
        if (!dst) continue;
        const size_t sb = (size_t)s * B + b;
        const int *off = W.off + sb * (n + 1);
        const int2 *ent = W.ent + sb * 4 * n;
        float acc[GCH];
#pragma unroll
        for (int k = 0; k < GCH; ++k) acc[k] = 0.0f;
        gather_list<GCH>(gb, (size_t)n, ent, off[q], off[q + 1], nc, acc);
#pragma unroll
        for (int k = 0; k < GCH; ++k)
            if (k < nc) dst[((size_t)b * C + c0 + k) * n + q] = acc[k];
    }
}

// standalone IFRNet.warp: g_img[b,c,q] = sum over the (sorted) list of q of w * g_out[b,c,p].
// grid (pixel blocks, channel chunks, B)
__global__ void __launch_bounds__(NT) k_flow_warp_bwd_gather(const float *__restrict__ g_out,
                                                             const int *__restrict__ ws, float *__restrict__ g_img,
                                                             int B, int C, int h, int w)
{
    const int n = h * w, q = blockIdx.x * NT + threadIdx.x, b = blockIdx.z;
    if (q >= n) return;
    const InvWs W = inv_ws(const_cast<int *>(ws), B, n);
    const int c0 = blockIdx.y * GCH, nc = min(GCH, C - c0);
    const float *gb = g_out + ((size_t)b * C + c0) * n;
    const int *off = W.off + (size_t)b * (n + 1);
    const int2 *ent = W.ent + (size_t)b * 4 * n;
    float acc[GCH];
#pragma unroll
    for (int k = 0; k < GCH; ++k) acc[k] = 0.0f;
    gather_list<GCH>(gb, (size_t)n, ent, off[q], off[q + 1], nc, acc);
#pragma unroll
    for (int k = 0; k < GCH; ++k)
        if (k < nc) g_img[((size_t)b * C + c0 + k) * n + q] = acc[k];
}

// ---- round 5: ANCHOR lists -----------------------------------------------------------------------------------
// The four cells an output pixel p touches are the 2x2 block at its top-left tap (y0, x0) -- its ANCHOR.  Listing p
// once, under its anchor, instead of once under each of the four cells makes the integer work of the inverse map a
// quarter (one counting atomic, one entry of 16 bytes per pixel and source instead of four of 8), the lists average
// ONE entry instead of four (nothing to sort, no long serial tails where the flow converges), and -- the point -- the
// channel values g[c][p] are fetched ONCE per pixel instead of once per touched cell:
//   S_k[a] = sum over the (sorted) list of anchor a of  w_k(p) * sc(p) * g[p],  k = the four corners of the block
//   grad[q] = ((S_0[q] + S_1[q - (0,1)]) + S_2[q - (1,0)]) + S_3[q - (1,1)]
// Deterministic like the cell lists: every list is sorted by p, the four terms are added in a fixed order.  Zero-weight
// corners are skipped exactly as dest_of skips them (a clamped tap carries weight 0: it would otherwise index a
// neighbour that does not exist, and 0 x inf would poison the sum).
// The lists of ALL pyramid levels of a step are built by one count / scan / fill / sort pass (they depend only on the
// teacher's flows: known when the forward pass ends), not by six launches per level inside backward.
#ifndef MVF_ANC_CH
#define MVF_ANC_CH 8
#endif
constexpr int ACH = MVF_ANC_CH;            // channels per pass of a lane over its two anchor lists
constexpr int MAXLV = 8;
struct AncLevel {
    const float *prep, *xs, *ys;
    int *cnt, *cur, *off;                  // [2][B][n], [2][B][n], [2][B][n+1]
    int4 *ent;                             // [2][B][n]: (p, wx bits, wy bits, merge weight bits)
    int h, w, blk0;                        // blk0: first block of the level in the flattened pixel grid
};
struct AncLevels {
    AncLevel l[MAXLV];
    int n, B;
};
MVF_DEV int level_of_block(const AncLevels &L, int blk)
{
    int lv = 0;
#pragma unroll
    for (int k = 1; k < MAXLV; ++k)
        if (k < L.n && blk >= L.l[k].blk0) lv = k;
    return lv;
}
struct AncTap {
    int anchor;
    float wx, wy, sc;
};
MVF_DEV AncTap anchor_of(const AncLevel &v, int b, int s, int i)
{
    const int n = v.h * v.w, y = i / v.w, x = i - y * v.w;
    const float *pp = v.prep + (size_t)b * PREP * n + i;
    const Tap t = flow_tap_xy(pp[(size_t)(4 + 2 * s) * n], pp[(size_t)(5 + 2 * s) * n], v.xs, v.ys, x, y, v.h, v.w);
    const float m = pp[8 * (size_t)n];
    AncTap a;
    a.anchor = t.y0 * v.w + t.x0;
    a.wx = t.wx; a.wy = t.wy;
    a.sc = (s == 0) ? m : 1.0f - m;
    return a;
}
// grid (pixel blocks of all levels, B, 2 sources)
__global__ void __launch_bounds__(NT) k_anc_count(AncLevels L)
{
    const int lv = level_of_block(L, blockIdx.x);
    const AncLevel &v = L.l[lv];
    const int n = v.h * v.w, i = (blockIdx.x - v.blk0) * NT + threadIdx.x, b = blockIdx.y, s = blockIdx.z;
    if (i >= n) return;
    const AncTap a = anchor_of(v, b, s, i);
    atomicAdd(v.cnt + ((size_t)s * L.B + b) * n + a.anchor, 1);
}
// block = (source-image pair, level): exclusive scan of the counts (the cell-list scan above, per level)
__global__ void __launch_bounds__(1024) k_anc_scan(AncLevels L)
{
    // chunks of 4,096 counts: a lane takes four consecutive ones (coalesced), the block scans the lane totals
    // (shuffle scan inside a wave, the 16 wave totals through LDS), the running prefix carries over
    __shared__ int wtot[16];
    __shared__ int carry_s;
    const AncLevel &v = L.l[blockIdx.y];
    const int n = v.h * v.w;
    const int *cnt = v.cnt + (size_t)blockIdx.x * n;
    int *off = v.off + (size_t)blockIdx.x * (n + 1);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    int carry = 0;
    for (int base = 0; base < n; base += 4096) {
        const int i0 = base + 4 * t;
        int c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = (i0 + k < n) ? cnt[i0 + k] : 0;
        const int tot = (c[0] + c[1]) + (c[2] + c[3]);
        int inc = tot;                                      // inclusive scan over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wtot[wv] = inc;
        __syncthreads();
        int wpre = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) wpre += (k < wv) ? wtot[k] : 0;
        int run = carry + wpre + inc - tot;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i0 + k < n) off[i0 + k] = run;
            run += c[k];
        }
        if (t == 1023) carry_s = run;
        __syncthreads();
        carry = carry_s;
    }
    if (t == 0) off[n] = carry;
}
__global__ void __launch_bounds__(NT) k_anc_fill(AncLevels L)
{
    const int lv = level_of_block(L, blockIdx.x);
    const AncLevel &v = L.l[lv];
    const int n = v.h * v.w, i = (blockIdx.x - v.blk0) * NT + threadIdx.x, b = blockIdx.y, s = blockIdx.z;
    if (i >= n) return;
    const AncTap a = anchor_of(v, b, s, i);
    const size_t sb = (size_t)s * L.B + b;
    const int slot = v.off[sb * (n + 1) + a.anchor] + atomicAdd(v.cur + sb * n + a.anchor, 1);
    v.ent[sb * n + slot] = make_int4(i, __float_as_int(a.wx), __float_as_int(a.wy), __float_as_int(a.sc));
}
// every anchor sorts its list by output pixel (lists average one entry: most lanes have nothing to do).  Lists of 3 .. 32
// entries: insertion sort by the anchor's lane.  Longer lists -- a flow field that leaves the image piles a row's pixels on
// that row's border anchor (the taps clamp) -- are sorted by the whole WAVE, one list after the other (one lane
// moved O(L^2) entries through global memory): rank sort, every lane holds up to ANC_KMAX of the list's entries, counts for
// each the entries with a smaller pixel index (the list is streamed through the wave 64 keys at a time) and writes it to
// its rank; L^2 / 64 compares per lane, no dependent memory traffic.  Lists beyond 64 * ANC_KMAX entries (only a flow
// that throws most of an image onto ONE anchor) keep the serial sort.
constexpr int ANC_SERIAL = 32, ANC_KMAX = 16;
MVF_DEV void anc_sort_serial(int4 *__restrict__ ent, int lo, int hi)
{
    for (int a = lo + 1; a < hi; ++a) {
        const int4 e = ent[a];
        int c = a - 1;
        while (c >= lo && ent[c].x > e.x) { ent[c + 1] = ent[c]; --c; }
        ent[c + 1] = e;
    }
}
MVF_DEV void anc_sort_wave(int4 *__restrict__ ent, int lo, int hi)      // all 64 lanes, wave-uniform arguments
{
    const int lane = threadIdx.x & (kWave - 1), len = hi - lo;
    int4 mine[ANC_KMAX];
    int rank[ANC_KMAX];
#pragma unroll
    for (int k = 0; k < ANC_KMAX; ++k) {
        const int a = lane + k * kWave;
        mine[k] = ent[lo + min(a, len - 1)];
        rank[k] = 0;
    }
    for (int c0 = 0; c0 < len; c0 += kWave) {
        const int key = (c0 + lane < len) ? ent[lo + c0 + lane].x : 0x7fffffff;
#pragma unroll 8
        for (int j = 0; j < kWave; ++j) {
            const int other = __shfl(key, j);
#pragma unroll
            for (int k = 0; k < ANC_KMAX; ++k) rank[k] += (other < mine[k].x) ? 1 : 0;      // pixel indices are distinct
        }
    }
    // every load of the list above has returned (the ranks depend on them) before the first store below is issued
#pragma unroll
    for (int k = 0; k < ANC_KMAX; ++k)
        if (lane + k * kWave < len) ent[lo + rank[k]] = mine[k];
}
__global__ void __launch_bounds__(NT) k_anc_sort(AncLevels L)
{
    const int lv = level_of_block(L, blockIdx.x);
    const AncLevel &v = L.l[lv];
    const int n = v.h * v.w, q = (blockIdx.x - v.blk0) * NT + threadIdx.x;
    const size_t sb = (size_t)blockIdx.z * L.B + blockIdx.y;
    const int *off = v.off + sb * (n + 1);
    int4 *ent = v.ent + sb * n;
    int lo = 0, hi = 0;
    if (q < n) { lo = off[q]; hi = off[q + 1]; }
    const int len = hi - lo;            // a pair is put in order by its reader (k_fusion_level_bwd_anchor)
    if ((len >= 3 && len <= ANC_SERIAL) || len > kWave * ANC_KMAX) anc_sort_serial(ent, lo, hi);
    unsigned long long todo = __ballot(len > ANC_SERIAL && len <= kWave * ANC_KMAX);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        anc_sort_wave(ent, __shfl(lo, src), __shfl(hi, src));
    }
}

// the four corner weights of an entry, as dest_of forms them (merge weight folded in), 0 where the corner is dropped
MVF_DEV void corner_weights(const int4 &e, float wt[4])
{
    const float fw = __int_as_float(e.y), fe = 1.0f - fw, fn = __int_as_float(e.z), fs = 1.0f - fn;
    const float sc = __int_as_float(e.w);
    const float raw[4] = {fs * fe, fs * fw, fn * fe, fn * fw};
#pragma unroll
    for (int k = 0; k < 4; ++k) wt[k] = (raw[k] != 0.0f) ? raw[k] * sc : 0.0f;
}
// lane t-1's value (0 into lane 0 of the wave): DPP wave_shr:1
MVF_DEV float from_prev_lane(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
// A workgroup owns a 32 x 8 region of anchors, one per lane, and emits the 31 x 7 cells whose four anchors lie inside
// it.  Per round of ACH channels a lane forms the four corner sums of its anchor (stage A: the channel values of an
// output pixel are fetched ONCE); its cell then needs S_0 (own), S_1 of the west anchor = lane t - 1 (one DPP wave
// shift: a region row is half a wave, and the lane a shift would take across a row end never emits), and S_2, S_3 of
// the row above: S_2 and the already west-shifted S_3 go through LDS (two planes, double buffered: ONE barrier per
// round).  Measured against an LDS-free form in which a lane evaluates the anchor above its cell as well (every g
// fetched by two lanes): equal at near-identity flows, 1.5 x faster at 6 px flows (that form's second readers sit on
// other XCDs: L2 hit rate 30 %, twice the HBM reads; its two list walks per lane diverge twice).
constexpr int ATW = 32, ATH = 8;           // anchors per region
constexpr int AGRP = 64;                   // channels per block (the list entries are read once per group)
#ifndef MVF_ANC_WAVES
#define MVF_ANC_WAVES 4
#endif
// grid (region tiles, channel groups, B * 2 sources)
__global__ void __launch_bounds__(NT, MVF_ANC_WAVES) k_fusion_level_bwd_anchor(const float *__restrict__ g_out,
                                                                const int *__restrict__ off_all,
                                                                const int4 *__restrict__ ent_all,
                                                                float *__restrict__ g_fn1, float *__restrict__ g_fp1,
                                                                int B, int C, int h, int w, int tiles_x)
{
    __shared__ float S[2][2][ACH][NT];
    const int n = h * w;
    const int b = blockIdx.z;
    // Round 6 (counters of level 0 at B 36, 6 px flows: L2 hit rate 34 %, 0.95 GB fetched for 0.32 GB of inputs --
    // profiles/r06_anchor_gather_pmc.csv): (i) BOTH sources in one workgroup, one after the other: they gather the same
    // channel planes of g a few pixels apart, so the second walk hits the lines the first one brought in (before: one
    // grid slice per source, dispatched far apart -- every plane fetched twice); (ii) every XCD owns one contiguous run
    // of tiles (the dispatcher deals consecutive workgroups to the eight XCDs, whose L2s do not talk to each other:
    // horizontally adjacent tiles read the same unaligned 128-byte lines and each fetched them for itself).
    const int ntile = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tile = xcd * (ntile >> 3) + min(xcd, ntile & 7) + slot;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int t = threadIdx.x, lx = t & (ATW - 1), ly = t / ATW;
    // region origin: one anchor row / column in front of the tile's cells
    const int ay = ty * (ATH - 1) - 1 + ly, ax = tx * (ATW - 1) - 1 + lx;
    const bool avalid = ay >= 0 && ay < h && ax >= 0 && ax < w;
    // the cell this lane emits: its own anchor position, if the three other anchors are in the region
    const bool emit = lx >= 1 && ly >= 1 && ay < h && ax < w;      // (ay, ax >= 0 follows)
    const int CT = 2 * (C + EMB);
    const int c_lo = blockIdx.y * AGRP, c_hi = min(c_lo + AGRP, C);
    const float *gb = uniform_ptr(g_out + ((size_t)b * CT + C + EMB) * n);
    const unsigned q4 = (unsigned)(max(ay, 0) * w + max(ax, 0)) * 4u;
    int buf = 0;
    for (int s = 0; s < 2; ++s) {
        float *dst = (s == 0) ? g_fn1 : g_fp1;
        if (!dst) continue;                                        // (workgroup-uniform)
        const size_t sb = (size_t)s * B + b;
        const int *off = off_all + sb * (n + 1);
        const int4 *ent = ent_all + sb * n;
        int lo = 0, hi = 0;
        if (avalid) { lo = off[ay * w + ax]; hi = off[ay * w + ax + 1]; }
        const int cnt = hi - lo;
        // the first two entries of the list up front (lists average one entry); clamped index, predicated use
        int4 e0 = ent[min(lo, n - 1)], e1 = ent[min(lo + 1, n - 1)];
        if (cnt == 2 && e0.x > e1.x) { const int4 tmp = e0; e0 = e1; e1 = tmp; }     // (k_anc_sort leaves pairs to the reader)
        float w0[4], w1[4];
        corner_weights(e0, w0);
        corner_weights(e1, w1);
        float *db = uniform_ptr(dst + (size_t)b * C * n);
        const unsigned o0 = (unsigned)e0.x * 4u, o1 = (unsigned)e1.x * 4u;
        for (int c0 = c_lo; c0 < c_hi; c0 += ACH, buf ^= 1) {
            const int nc = min(ACH, c_hi - c0);
            const float *gc = gb + (size_t)c0 * n;
            float acc[4][ACH];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int j = 0; j < ACH; ++j) acc[k][j] = 0.0f;
            if (cnt > 0) {
                float v0[ACH], v1[ACH];
#pragma unroll
                for (int j = 0; j < ACH; ++j) v0[j] = ldg_at(gc + (size_t)min(j, nc - 1) * n, o0);
                if (cnt > 1) {
#pragma unroll
                    for (int j = 0; j < ACH; ++j) v1[j] = ldg_at(gc + (size_t)min(j, nc - 1) * n, o1);
                }
#pragma unroll
                for (int j = 0; j < ACH; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k][j] += w0[k] * v0[j];
                if (cnt > 1) {
#pragma unroll
                    for (int j = 0; j < ACH; ++j)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k][j] += w1[k] * v1[j];
                }
                for (int q = lo + 2; q < hi; ++q) {
                    const int4 e = ent[q];
                    float wq[4];
                    corner_weights(e, wq);
#pragma unroll
                    for (int j = 0; j < ACH; ++j) {
                        const float v = ldg_at(gc + (size_t)min(j, nc - 1) * n, (unsigned)e.x * 4u);
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[k][j] += wq[k] * v;
                    }
                }
            }
            float west[ACH];
#pragma unroll
            for (int j = 0; j < ACH; ++j) {
                west[j] = from_prev_lane(acc[1][j]);
                S[buf][0][j][t] = acc[2][j];
                S[buf][1][j][t] = from_prev_lane(acc[3][j]);
            }
            __syncthreads();        // (the buffer written two rounds ago was read before the previous round's barrier)
            if (emit) {
#pragma unroll
                for (int j = 0; j < ACH; ++j) {
                    if (j >= nc) break;
                    const float r = ((acc[0][j] + west[j]) + S[buf][0][j][t - ATW]) + S[buf][1][j][t - ATW];
                    stg_at(db + (size_t)(c0 + j) * n, q4, r);
                }
            }
        }
    }
}

// count / scan / fill / sort of the inverse tap lists (see above); FLOW: one source
template <bool FLOW>
int build_inverse_lists(const float *field, const float *xs, const float *ys, int32_t *workspace, int B, int h, int w,
                        hipStream_t st)
{
    const int n = h * w;
    const dim3 gp((unsigned)((n + NT - 1) / NT), (unsigned)B, FLOW ? 1 : 2);
    const size_t cnt_bytes = (size_t)2 * B * n * sizeof(int);
    hipError_t e = hipMemsetAsync(workspace, 0, cnt_bytes, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_inv_count<FLOW>, gp, dim3(NT), 0, st, field, xs, ys, workspace, B, h, w);
    hipLaunchKernelGGL(k_inv_scan, dim3((unsigned)((FLOW ? 1 : 2) * B)), dim3(1024), 0, st, workspace, B, n);
    e = hipMemsetAsync(workspace, 0, cnt_bytes, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_inv_fill<FLOW>, gp, dim3(NT), 0, st, field, xs, ys, workspace, B, h, w);
    hipLaunchKernelGGL(k_inv_sort, dim3((unsigned)((n + NT - 1) / NT), (unsigned)((FLOW ? 1 : 2) * B)), dim3(NT), 0, st,
                       workspace, B, n);
    return 0;
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
    // three feature maps and the prep planes read once; feat_0 | emb(0) | merged (C + EMB each) written once
    ProfScope ps(MVF_PROF_FUSION_FWD, stream, 4LL * B * h * w * (3LL * C + PREP + 2LL * (C + EMB)));
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

size_t mvf_fusion_bwd_workspace_ints(int B, int h, int w)
{
    const size_t n = (size_t)h * w;
    return (size_t)2 * B * n + (size_t)2 * B * (n + 1) + (size_t)2 * B * 4 * n * 2 + 16;
}

int mvf_fusion_level_bwd_gather(const float *g_out, const float *prep, const float *xs, const float *ys,
                                float *g_feat_n1, float *g_feat_p1, int32_t *workspace, int B, int C, int h, int w,
                                void *stream)
{
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
    if (!g_out || !prep || !xs || !ys || !workspace || B > 32767) return (int)hipErrorInvalidValue;
    if (!g_feat_n1 && !g_feat_p1) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int n = h * w;
    // the merged half of g_out read once, two feature gradients written once (the inverse tap lists are this
    // build's own traffic, not priced)
    ProfScope ps(MVF_PROF_FUSION_BWD_GATHER, stream, 4LL * B * n * C * (1 + (g_feat_n1 ? 1 : 0) + (g_feat_p1 ? 1 : 0)));
    const int err = build_inverse_lists<false>(prep, xs, ys, workspace, B, h, w, st);
    if (err) return err;
    const int nchunk = (C + GCH - 1) / GCH;
    hipLaunchKernelGGL(k_fusion_level_bwd_gather, dim3((unsigned)((n + NT - 1) / NT), (unsigned)nchunk, (unsigned)B),
                       dim3(NT), 0, st, g_out, workspace, g_feat_n1, g_feat_p1, B, C, h, w);
    return hip_check_launch();
}

// ints of one level's anchor lists: entries [2][B][n] x 4, offsets [2][B][n+1]
size_t mvf_fusion_lists_level_ints(int B, int h, int w)
{
    const size_t n = (size_t)h * w;
    return (((size_t)2 * B * n * 4 + (size_t)2 * B * (n + 1)) + 3) & ~(size_t)3;
}
// ints of the scratch the build needs (counters + cursors of every level)
size_t mvf_fusion_lists_scratch_ints(int B, int n_levels, const int32_t *hs, const int32_t *ws)
{
    size_t t = 0;
    for (int i = 0; i < n_levels; ++i) t += (size_t)2 * 2 * B * hs[i] * ws[i];
    return t + 4;
}

int mvf_fusion_lists_build(const float *const *preps, const float *const *xs, const float *const *ys, const int32_t *hs,
                           const int32_t *ws, int n_levels, int B, int32_t *const *lists, int32_t *scratch,
                           void *stream)
{
    if (n_levels <= 0 || B <= 0) return 0;
    if (n_levels > MAXLV || B > 32767 || !preps || !xs || !ys || !hs || !ws || !lists || !scratch)
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    AncLevels L = {};
    L.n = n_levels; L.B = B;
    size_t so = 0;
    int blk = 0;
    for (int i = 0; i < n_levels; ++i) {
        if (!preps[i] || !xs[i] || !ys[i] || !lists[i] || hs[i] < 2 || ws[i] < 2 || (((uintptr_t)lists[i]) & 15))
            return (int)hipErrorInvalidValue;
        const size_t n = (size_t)hs[i] * ws[i];
        AncLevel &v = L.l[i];
        v.prep = preps[i]; v.xs = xs[i]; v.ys = ys[i]; v.h = hs[i]; v.w = ws[i];
        v.ent = reinterpret_cast<int4 *>(lists[i]);
        v.off = lists[i] + (size_t)2 * B * n * 4;
        v.cnt = scratch + so; so += (size_t)2 * B * n;
        v.cur = scratch + so; so += (size_t)2 * B * n;
        v.blk0 = blk;
        blk += (int)((n + NT - 1) / NT);
    }
    // (this build's own integer work: no algorithmic bytes, its time counts for the adjoint it serves)
    ProfScope ps(MVF_PROF_FUSION_BWD_GATHER, stream, 0);
    hipError_t e = hipMemsetAsync(scratch, 0, so * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    const dim3 gp((unsigned)blk, (unsigned)B, 2);
    hipLaunchKernelGGL(k_anc_count, gp, dim3(NT), 0, st, L);
    hipLaunchKernelGGL(k_anc_scan, dim3((unsigned)(2 * B), (unsigned)n_levels), dim3(1024), 0, st, L);
    hipLaunchKernelGGL(k_anc_fill, gp, dim3(NT), 0, st, L);
    hipLaunchKernelGGL(k_anc_sort, gp, dim3(NT), 0, st, L);
    return hip_check_launch();
}

int mvf_fusion_level_bwd_lists(const float *g_out, const int32_t *lists, float *g_feat_n1, float *g_feat_p1, int B,
                               int C, int h, int w, void *stream)
{
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0) return 0;
    if (!g_out || !lists || B > 32767 || h < 2 || w < 2) return (int)hipErrorInvalidValue;
    if (!g_feat_n1 && !g_feat_p1) return 0;
    const int n = h * w;
    if ((double)n * 4.0 * ACH >= 2147483648.0) return (int)hipErrorInvalidValue;    // 32-bit byte offsets inside a chunk of planes
    // the merged half of g_out read once, two feature gradients written once (the lists are this build's own traffic)
    ProfScope ps(MVF_PROF_FUSION_BWD_GATHER, stream, 4LL * B * n * C * (1 + (g_feat_n1 ? 1 : 0) + (g_feat_p1 ? 1 : 0)));
    const int tiles_x = (w + ATW - 2) / (ATW - 1), tiles_y = (h + ATH - 2) / (ATH - 1);
    hipLaunchKernelGGL(k_fusion_level_bwd_anchor, dim3((unsigned)(tiles_x * tiles_y), (unsigned)((C + AGRP - 1) / AGRP),
                                                      (unsigned)B),
                       dim3(NT), 0, (hipStream_t)stream, g_out, lists + (size_t)2 * B * n * 4,
                       reinterpret_cast<const int4 *>(lists), g_feat_n1, g_feat_p1, B, C, h, w, tiles_x);
    return hip_check_launch();
}

int mvf_flow_warp_bwd_gather(const float *flow, const float *xs, const float *ys, const float *g_out, float *g_img,
                             int32_t *workspace, int B, int C, int H, int W, void *stream)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    if (!flow || !xs || !ys || !g_out || !g_img || !workspace || B > 32767 || H < 2 || W < 2)
        return (int)hipErrorInvalidValue;
    hipStream_t st = (hipStream_t)stream;
    const int n = H * W;
    const int err = build_inverse_lists<true>(flow, xs, ys, workspace, B, H, W, st);
    if (err) return err;
    const int nchunk = (C + GCH - 1) / GCH;
    hipLaunchKernelGGL(k_flow_warp_bwd_gather, dim3((unsigned)((n + NT - 1) / NT), (unsigned)nchunk, (unsigned)B),
                       dim3(NT), 0, st, g_out, workspace, g_img, B, C, H, W);
    return hip_check_launch();
}

}  // extern "C"
