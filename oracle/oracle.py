"""ctypes front-end of the CPU parity oracle (oracle/mvf_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; the product package never imports this module.

Parity status: pinned against golden vectors captured from the reference
(tests/golden/*.npz, checked by tests/test_oracle_golden.py).

All functions take and return numpy fp32 arrays laid out like the reference's tensors.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmvf_oracle.so")

NO_SSIM, AVG_REPROJ, NO_AUTOMASK = 1, 2, 4


def build(force=False):
    src = os.path.join(_HERE, "mvf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.mvfo_smooth.restype = C.c_double
        _lib.mvfo_losses_base_fwd.restype = C.c_double
        _lib.mvfo_silog.restype = C.c_double
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(_fp)


def _pi(a):
    return None if a is None else a.ctypes.data_as(_ip)


def depth_consts(min_depth, max_depth):
    """(min_disp, range) rounded to fp32 the way the Python scalars of reference
    layers.py:21-23 are when they meet the fp32 tensor."""
    min_disp = 1 / max_depth
    max_disp = 1 / min_depth
    return np.float32(min_disp), np.float32(max_disp - min_disp)


def disp_to_depth(disp, min_depth=0.1, max_depth=100.0):
    disp = _f(disp)
    md, rg = depth_consts(min_depth, max_depth)
    scaled, depth = np.empty_like(disp), np.empty_like(disp)
    lib().mvfo_disp_to_depth(_p(disp), _p(scaled), _p(depth), C.c_long(disp.size),
                             C.c_float(md), C.c_float(rg))
    return scaled, depth


def backproject(depth, inv_K):
    depth, inv_K = _f(depth), _f(inv_K)
    B, _, H, W = depth.shape
    cam = np.empty((B, 4, H * W), np.float32)
    lib().mvfo_backproject(_p(depth), _p(inv_K), _p(cam), B, H, W)
    return cam


def proj_matrix(K, T):
    K, T = _f(K), _f(T)
    P = np.empty((K.shape[0], 3, 4), np.float32)
    lib().mvfo_proj_matrix(_p(K), _p(T), _p(P), K.shape[0])
    return P


def project(cam, K, T, H, W, eps=1e-7):
    cam, K, T = _f(cam), _f(K), _f(T)
    B = cam.shape[0]
    pix = np.empty((B, H, W, 2), np.float32)
    lib().mvfo_project(_p(cam), _p(K), _p(T), _p(pix), B, H, W, C.c_float(eps))
    return pix


def grid_sample(img, grid, want_idx=False):
    img, grid = _f(img), _f(grid)
    B, Cc, H, W = img.shape
    out = np.empty_like(img)
    x0 = np.empty((B, H, W), np.int32)
    y0 = np.empty((B, H, W), np.int32)
    lib().mvfo_grid_sample(_p(img), _p(grid), _p(out), _pi(x0), _pi(y0), B, Cc, H, W)
    return (out, x0, y0) if want_idx else out


def grid_sample_bwd(img, grid, gout):
    img, grid, gout = _f(img), _f(grid), _f(gout)
    B, Cc, H, W = img.shape
    gg = np.empty((B, H, W, 2), np.float32)
    lib().mvfo_grid_sample_bwd(_p(img), _p(grid), _p(gout), _p(gg), B, Cc, H, W)
    return gg


def warp_fwd(disp, inv_K, K, T, src, min_depth=0.1, max_depth=100.0, eps=1e-7):
    """-> dict(warped, pix, x0, y0) for one source (reference train.py:956-971)."""
    disp, inv_K, K, T, src = _f(disp), _f(inv_K), _f(K), _f(T), _f(src)
    B, _, H, W = disp.shape
    md, rg = depth_consts(min_depth, max_depth)
    warped = np.empty((B, 3, H, W), np.float32)
    pix = np.empty((B, H, W, 2), np.float32)
    x0 = np.empty((B, H, W), np.int32)
    y0 = np.empty((B, H, W), np.int32)
    lib().mvfo_warp_fwd(_p(disp), _p(inv_K), _p(K), _p(T), _p(src), _p(warped), _p(pix),
                        _pi(x0), _pi(y0), B, H, W, C.c_float(md), C.c_float(rg), C.c_float(eps))
    return dict(warped=warped, pix=pix, x0=x0, y0=y0)


def warp_bwd(disp, inv_K, K, T, src, gwarped, gdisp=None, min_depth=0.1, max_depth=100.0,
             eps=1e-7):
    """-> (grad_disp accumulated, grad_T [B,4,4])."""
    disp, inv_K, K, T, src, gwarped = map(_f, (disp, inv_K, K, T, src, gwarped))
    B, _, H, W = disp.shape
    md, rg = depth_consts(min_depth, max_depth)
    if gdisp is None:
        gdisp = np.zeros_like(disp)
    gT = np.empty((B, 4, 4), np.float32)
    lib().mvfo_warp_bwd(_p(disp), _p(inv_K), _p(K), _p(T), _p(src), _p(gwarped), _p(gdisp),
                        _p(gT), B, H, W, C.c_float(md), C.c_float(rg), C.c_float(eps))
    return gdisp, gT


def ssim(x, y):
    x, y = _f(x), _f(y)
    B, Cc, H, W = x.shape
    out = np.empty_like(x)
    lib().mvfo_ssim(_p(x), _p(y), _p(out), B, Cc, H, W)
    return out


def ssim_bwd(x, y, gout):
    x, y, gout = _f(x), _f(y), _f(gout)
    B, Cc, H, W = x.shape
    gx, gy = np.empty_like(x), np.empty_like(x)
    lib().mvfo_ssim_bwd(_p(x), _p(y), _p(gout), _p(gx), _p(gy), B, Cc, H, W)
    return gx, gy


def reprojection(pred, tgt, no_ssim=False):
    pred, tgt = _f(pred), _f(tgt)
    B, _, H, W = pred.shape
    out = np.empty((B, 1, H, W), np.float32)
    lib().mvfo_reprojection(_p(pred), _p(tgt), _p(out), B, H, W, int(bool(no_ssim)))
    return out


def smooth(disp, img, normalise=True):
    disp, img = _f(disp), _f(img)
    B, _, H, W = disp.shape
    mean = np.empty((B,), np.float32)
    v = lib().mvfo_smooth(_p(disp), _p(img), B, H, W, int(normalise), _p(mean))
    return float(v), mean


def smooth_bwd(disp, img, scale, normalise=True, gdisp=None):
    disp, img = _f(disp), _f(img)
    B, _, H, W = disp.shape
    if gdisp is None:
        gdisp = np.zeros_like(disp)
    lib().mvfo_smooth_bwd(_p(disp), _p(img), _p(gdisp), B, H, W, int(normalise), C.c_float(scale))
    return gdisp


def _ptr_array(arrs):
    return (_fp * len(arrs))(*[a.ctypes.data_as(_fp) for a in arrs])


def losses_base_fwd(tgt, warped, src, noise=None, mask_rec=None, flags=0):
    """-> dict(photo, rp, idl, to_opt, idx).  warped/src: sequences of [B,3,H,W]."""
    tgt = _f(tgt)
    warped = [_f(w) for w in warped]
    src = [_f(s) for s in src] if src is not None else warped
    S = len(warped)
    B, _, H, W = tgt.shape
    noise = _f(noise) if noise is not None else np.zeros((B, S, H, W), np.float32)
    mask = _f(mask_rec) if mask_rec is not None else None
    rp = np.empty((B, S, H, W), np.float32)
    idl = np.empty((B, S, H, W), np.float32)
    to_opt = np.empty((B, H, W), np.float32)
    idx = np.empty((B, H, W), np.int32)
    photo = lib().mvfo_losses_base_fwd(_p(tgt), _ptr_array(warped), _ptr_array(src), _p(noise),
                                       _p(mask), S, int(flags), _p(rp), _p(idl), _p(to_opt),
                                       _pi(idx), B, H, W)
    return dict(photo=float(photo), rp=rp, idl=idl, to_opt=to_opt, idx=idx)


def losses_base_bwd(tgt, warped, idx, mask_rec=None, flags=0, gloss=1.0):
    tgt = _f(tgt)
    warped = [_f(w) for w in warped]
    S = len(warped)
    B, _, H, W = tgt.shape
    mask = _f(mask_rec) if mask_rec is not None else None
    idx = np.ascontiguousarray(idx, np.int32)
    gw = [np.empty((B, 3, H, W), np.float32) for _ in range(S)]
    lib().mvfo_losses_base_bwd(_p(tgt), _ptr_array(warped), _pi(idx), _p(mask), S, int(flags),
                               C.c_float(gloss), _ptr_array(gw), B, H, W)
    return gw


def pose(axisangle, translation, invert=False):
    aa = _f(axisangle).reshape(-1, 3)
    tr = _f(translation).reshape(-1, 3)
    M = np.empty((aa.shape[0], 4, 4), np.float32)
    lib().mvfo_pose(_p(aa), _p(tr), int(bool(invert)), _p(M), aa.shape[0])
    return M


_dp = C.POINTER(C.c_double)


def _pd(a):
    return None if a is None else a.ctypes.data_as(_dp)


def unit_grads_f64(disp, tgt, src, T, K, inv_K, warped, idx, mask_rec=None, flags=0, smoothness=1e-3,
                   min_depth=0.1, max_depth=100.0, gloss=1.0, eps=1e-7):
    """The adjoint of a unit evaluated in DOUBLE (mvfo_*_bwd_f64: every decision -- argmin `idx`, bilinear cell, clamp,
    signs -- is the fp32 forward's, every value is formed in double): grad_disp [B,1,H,W] float64, grad_T [S,B,4,4]
    float64.  `warped`, `idx`: what `unit()` returned for the same inputs."""
    disp, tgt, inv_K, K = _f(disp), _f(tgt), _f(inv_K), _f(K)
    warped = [_f(w) for w in warped]
    S = len(warped)
    B, _, H, W = tgt.shape
    mask = _f(mask_rec) if mask_rec is not None else None
    idx = np.ascontiguousarray(idx, np.int32)
    gw = [np.empty((B, 3, H, W), np.float64) for _ in range(S)]
    lib().mvfo_losses_base_bwd_f64(_p(tgt), _ptr_array(warped), _pi(idx), _p(mask), S, int(flags), C.c_double(gloss),
                                   (_dp * S)(*[a.ctypes.data_as(_dp) for a in gw]), B, H, W)
    md, rg = depth_consts(min_depth, max_depth)
    gdisp = np.zeros((B, 1, H, W), np.float64)
    gT = np.empty((S, B, 4, 4), np.float64)
    for k in range(S):
        Tk, sk = _f(T[k]), _f(src[k])
        lib().mvfo_warp_bwd_f64(_p(disp), _p(inv_K), _p(K), _p(Tk), _p(sk), _pd(gw[k]), _pd(gdisp), _pd(gT[k]),
                                B, H, W, C.c_float(md), C.c_float(rg), C.c_float(eps))
    lib().mvfo_smooth_bwd_f64(_p(disp), _p(tgt), _pd(gdisp), B, H, W, 1, C.c_double(gloss * smoothness))
    return gdisp, gT


def unit(disp, tgt, src, T, K, inv_K, noise=None, mask_rec=None, flags=0,
         smoothness=1e-3, min_depth=0.1, max_depth=100.0, want_grads=False, gloss=1.0, adjoint64=False):
    """One hot-path unit end to end: S x generate_images_pred + compute_losses_base
    (reference train.py:956-1051).  src [S,B,3,H,W], T [S,B,4,4].  `adjoint64` (with `want_grads`): also
    `grad_disp64`, `grad_T64` -- the same adjoint evaluated in double (`unit_grads_f64`)."""
    S = len(src)
    w = [warp_fwd(disp, inv_K, K, T[k], src[k], min_depth, max_depth) for k in range(S)]
    warped = [x["warped"] for x in w]
    automask = not (flags & NO_AUTOMASK)
    fw = losses_base_fwd(tgt, warped, list(src) if automask else None, noise, mask_rec, flags)
    sm, _ = smooth(disp, tgt, normalise=True)
    loss = fw["photo"] + smoothness * sm
    out = dict(loss=loss, photo=fw["photo"], smooth=sm, warped=warped, to_opt=fw["to_opt"],
               idx=fw["idx"], rp=fw["rp"], idl=fw["idl"], x0=[x["x0"] for x in w],
               y0=[x["y0"] for x in w], pix=[x["pix"] for x in w])
    n_id = 0 if not automask else (1 if flags & AVG_REPROJ else S)
    if automask:
        out["auto_mask"] = (fw["idx"] > n_id - 1).astype(np.float32)[:, None]
    if want_grads:
        gw = losses_base_bwd(tgt, warped, fw["idx"], mask_rec, flags, gloss)
        gdisp = np.zeros_like(_f(disp))
        gT = []
        for k in range(S):
            gdisp, g = warp_bwd(disp, inv_K, K, T[k], src[k], gw[k], gdisp, min_depth, max_depth)
            gT.append(g)
        gdisp = smooth_bwd(disp, tgt, gloss * smoothness, True, gdisp)
        out.update(grad_disp=gdisp, grad_T=np.stack(gT, 0), grad_warped=gw)
        if adjoint64:
            g64, t64 = unit_grads_f64(disp, tgt, src, T, K, inv_K, warped, fw["idx"] if fw["idx"] is not None else None,
                                      mask_rec, flags, smoothness, min_depth, max_depth, gloss)
            out.update(grad_disp64=g64, grad_T64=t64)
    return out


def flow_warp(img, flow, xs, ys, want_idx=False):
    """IFRNet.warp (reference networks/IFRNet.py:7-15); xs/ys = torch.linspace(-1,1,size)."""
    img, flow, xs, ys = _f(img), _f(flow), _f(xs), _f(ys)
    B, Cc, H, W = img.shape
    out = np.empty_like(img)
    x0 = np.empty((B, H, W), np.int32)
    y0 = np.empty((B, H, W), np.int32)
    lib().mvfo_flow_warp(_p(img), _p(flow), _p(xs), _p(ys), _p(out), _pi(x0), _pi(y0), B, Cc, H, W)
    return (out, x0, y0) if want_idx else out


def flow_warp_bwd(img, flow, xs, ys, gout):
    img, flow, xs, ys, gout = _f(img), _f(flow), _f(xs), _f(ys), _f(gout)
    B, Cc, H, W = img.shape
    g_img = np.zeros_like(img)
    g_flow = np.empty_like(flow)
    lib().mvfo_flow_warp_bwd(_p(img), _p(flow), _p(xs), _p(ys), _p(gout), _p(g_img), _p(g_flow),
                             B, Cc, H, W)
    return g_img, g_flow


def silog(pred, target, mask=None, beta=0.5, gloss=1.0, want_grads=False):
    """Trainer.compute_SI_log_depth_loss (reference train.py:924-941)."""
    pred, target = _f(pred), _f(target)
    mask = _f(mask) if mask is not None else None
    B = pred.shape[0]
    N = pred[0].size
    gp = np.empty_like(pred) if want_grads else None
    gt = np.empty_like(target) if want_grads else None
    v = lib().mvfo_silog(_p(pred), _p(target), _p(mask), C.c_float(beta), C.c_float(gloss), _p(gp),
                         _p(gt), B, C.c_long(N))
    return (float(v), gp, gt) if want_grads else float(v)


def _box(box):
    return np.ascontiguousarray(box, dtype=np.int32)


def affine_transform(img, angle, box):
    """Trainer.affine_transform (reference train.py:888-902): rotate, crop the box, resize back."""
    img, angle, box = _f(img), _f(angle).reshape(-1), _box(box)
    B, Cc, H, W = img.shape
    out = np.empty_like(img)
    lib().mvfo_affine_transform(_p(img), _p(angle), _pi(box), _p(out), B, Cc, H, W)
    return out


def affine_restore(depth, angle, box, ratio):
    """depth_restore of compute_depth_consistency_loss_affine (reference train.py:909-916)."""
    depth, angle, box, ratio = _f(depth), _f(angle).reshape(-1), _box(box), _f(ratio).reshape(-1)
    B, Cc, H, W = depth.shape
    out = np.empty_like(depth)
    lib().mvfo_affine_restore(_p(depth), _p(angle), _pi(box), _p(ratio), _p(out), B, Cc, H, W)
    return out


def affine_restore_bwd(gout, angle, box, ratio):
    gout, angle, box, ratio = _f(gout), _f(angle).reshape(-1), _box(box), _f(ratio).reshape(-1)
    B, Cc, H, W = gout.shape
    g = np.empty_like(gout)
    lib().mvfo_affine_restore_bwd(_p(gout), _p(angle), _pi(box), _p(ratio), _p(g), B, Cc, H, W)
    return g


def set_threads(n):
    """OpenMP threads used by the oracle (bench.py's cpu_baseline calibrates this); returns
    the count in effect."""
    return int(lib().mvfo_set_threads(int(n)))


# --------------------------------------------------------------------------- f1: FusionModule
# reference: networks/fusion_module.py:65-130 (what enters the per-scale 1x1 convolutions).
def resize_bilinear(x, oh, ow, scale_factor=None, align_corners=False):
    """F.interpolate(x, mode="bilinear") as ATen's CPU kernel evaluates it.  align_corners=False:
    src = scale*(dst+0.5)-0.5 clamped at 0 (scale = in/out for size=..., 1/scale_factor when a
    scale factor was given); align_corners=True: src = dst*(in-1)/(out-1).
    out = wy0*(wx0*v00 + wx1*v01) + wy1*(wx0*v10 + wx1*v11)."""
    x = _f(x)
    H, W = x.shape[-2:]
    f32 = np.float32

    def axis(n_in, n_out):
        d = np.arange(n_out, dtype=f32)
        if align_corners:
            scale = f32(n_in - 1) / f32(n_out - 1) if n_out > 1 else f32(0.0)
            src = (scale * d).astype(f32)
        else:
            scale = f32(1.0 / scale_factor) if scale_factor is not None else f32(n_in) / f32(n_out)
            src = scale * (d + f32(0.5)) - f32(0.5)
            src = np.maximum(src, f32(0.0))
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(f32)).astype(f32)
        return i0, i1, (f32(1.0) - l1).astype(f32), l1

    y0, y1, wy0, wy1 = axis(H, oh)
    x0, x1, wx0, wx1 = axis(W, ow)
    top = wx0 * x[..., y0, :][..., x0] + wx1 * x[..., y0, :][..., x1]
    bot = wx0 * x[..., y1, :][..., x0] + wx1 * x[..., y1, :][..., x1]
    return (wy0[:, None] * top.astype(f32) + wy1[:, None] * bot.astype(f32)).astype(f32)


def resize_bilinear_bwd(g, ih, iw, scale_factor=None, align_corners=False):
    """Adjoint of resize_bilinear w.r.t. its input (fp64 accumulation): scatter of the four taps."""
    g = np.asarray(g, np.float64)
    oh, ow = g.shape[-2:]
    f32 = np.float32

    def axis(n_in, n_out):
        d = np.arange(n_out, dtype=f32)
        if align_corners:
            scale = f32(n_in - 1) / f32(n_out - 1) if n_out > 1 else f32(0.0)
            src = (scale * d).astype(f32)
        else:
            scale = f32(1.0 / scale_factor) if scale_factor is not None else f32(n_in) / f32(n_out)
            src = np.maximum(scale * (d + f32(0.5)) - f32(0.5), f32(0.0))
        i0 = np.minimum(src.astype(np.int64), n_in - 1)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(f32)).astype(np.float64)
        return i0, i1, 1.0 - l1, l1

    y0, y1, wy0, wy1 = axis(ih, oh)
    x0, x1, wx0, wx1 = axis(iw, ow)
    My = np.zeros((ih, oh))
    Mx = np.zeros((iw, ow))
    for d in range(oh):
        My[y0[d], d] += wy0[d]
        My[y1[d], d] += wy1[d]
    for d in range(ow):
        Mx[x0[d], d] += wx0[d]
        Mx[x1[d], d] += wx1[d]
    return np.einsum("yd,...de,xe->...yx", My, g, Mx)


def flow_embedding(x, num_freqs=10):
    """Embedder.embed (fusion_module.py:7-37): [x, sin(2^k x), cos(2^k x) ...] on dim 1."""
    x = _f(x)
    parts = [x]
    for k in range(num_freqs):
        f = np.float32(2.0 ** k)
        parts += [np.sin(x * f).astype(np.float32), np.cos(x * f).astype(np.float32)]
    return np.concatenate(parts, 1)


def embedding_flows(flow, levels, litemono=False):
    """get_embedding_flow (fusion_module.py:65-78): cascaded half-resolution flows, values
    halved with the resolution (LiteMono: the first level is halved twice)."""
    x, outs = _f(flow), []
    for i in range(levels):
        x = resize_bilinear(x, x.shape[-2] // 2, x.shape[-1] // 2, 0.5) * np.float32(0.5)
        if i == 0 and litemono:
            x = resize_bilinear(x, x.shape[-2] // 2, x.shape[-1] // 2, 0.5) * np.float32(0.5)
        outs.append(x.astype(np.float32))
    return outs


def _warp_flow(flow, H, W):
    """flow_ of warp_features (fusion_module.py:80-90): direct resize, scaled with the size."""
    fh, fw = flow.shape[-2:]
    fl = resize_bilinear(flow, H, W)
    fl = fl * np.array([W / fw, H / fh], np.float32).reshape(1, 2, 1, 1)
    return np.ascontiguousarray(fl.astype(np.float32))


def _linspace(n):
    import torch
    return torch.linspace(-1.0, 1.0, n).numpy()


def fusion_forward(feats, flows, mask, litemono=False):
    """feats = [pyramid_n1, pyramid_0, pyramid_p1], flows = [flow_0_n1, flow_0_p1] (full
    resolution), mask [B,1,H,W] -> per level cat[f0, emb(0), m*[warp(fn1), emb(e_n1)] +
    (1-m)*[warp(fp1), emb(e_p1)]]."""
    L = len(feats[1])
    e0 = embedding_flows(np.zeros_like(_f(flows[0])), L, litemono)
    en = embedding_flows(flows[0], L, litemono)
    ep = embedding_flows(flows[1], L, litemono)
    outs = []
    for i in range(L):
        f0 = _f(feats[1][i])
        B, Cc, H, W = f0.shape
        xs, ys = _linspace(W), _linspace(H)
        wn = flow_warp(feats[0][i], _warp_flow(flows[0], H, W), xs, ys)
        wp = flow_warp(feats[2][i], _warp_flow(flows[1], H, W), xs, ys)
        m = resize_bilinear(mask, H, W)
        a = np.concatenate([wn, flow_embedding(en[i])], 1)
        bq = np.concatenate([wp, flow_embedding(ep[i])], 1)
        merged = (m * a + (np.float32(1.0) - m) * bq).astype(np.float32)
        outs.append(np.concatenate([f0, flow_embedding(e0[i]), merged], 1))
    return outs


def fusion_backward(feats, flows, mask, gouts):
    """Gradients of sum_i <out_i, gouts_i> w.r.t. the three feature pyramids (flows and mask
    come from the frozen teacher and carry none)."""
    L = len(feats[1])
    g_n1, g_0, g_p1 = [], [], []
    for i in range(L):
        B, Cc, H, W = feats[1][i].shape
        E = (gouts[i].shape[1] - 2 * Cc) // 2
        xs, ys = _linspace(W), _linspace(H)
        m = resize_bilinear(mask, H, W)
        g = _f(gouts[i])
        g_0.append(np.ascontiguousarray(g[:, :Cc]))
        gw = g[:, Cc + E:2 * Cc + E]
        g_n1.append(flow_warp_bwd(feats[0][i], _warp_flow(flows[0], H, W), xs, ys,
                                  np.ascontiguousarray(m * gw))[0])
        g_p1.append(flow_warp_bwd(feats[2][i], _warp_flow(flows[1], H, W), xs, ys,
                                  np.ascontiguousarray((np.float32(1.0) - m) * gw))[0])
    return g_n1, g_0, g_p1


# --------------------------------------------------------------------------- f4: stem max pooling
def maxpool3s2(x):
    """nn.MaxPool2d(3, 2, 1) of the ResNet trunks (reference networks/monodepth2.py:39,
    networks/posenet.py:21, 87) restated as ATen evaluates it (MaxPoolKernel: scan kh then kw over the
    in-bounds part of the window, `val > max || isnan(val)`, start at the first in-bounds element).
    x [P,H,W] -> (out [P,OH,OW], code [P,OH,OW] uint8 = kh*3+kw relative to (2*oy-1, 2*ox-1)).
    Pinned to ATen's CPU kernel in tests/test_oracle_golden.py."""
    x = np.asarray(x, np.float32)
    P, H, W = x.shape
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    oy, ox = np.arange(OH)[:, None], np.arange(OW)[None, :]
    best = np.full((P, OH, OW), -np.inf, np.float32)
    code = np.broadcast_to(np.where(2 * oy - 1 < 0, 3, 0) + np.where(2 * ox - 1 < 0, 1, 0), (P, OH, OW)).copy()
    for kh in range(3):
        for kw in range(3):
            yy, xx = 2 * oy - 1 + kh, 2 * ox - 1 + kw
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            v = x[:, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
            with np.errstate(invalid="ignore"):
                take = ok[None] & ((v > best) | np.isnan(v))
            best = np.where(take, v, best)
            code = np.where(take, kh * 3 + kw, code)
    return best, code.astype(np.uint8)


def maxpool3s2_bwd(g, code, H, W):
    """Adjoint of maxpool3s2: an input pixel collects the (<= 2 x 2) windows whose maximum it is, in
    ATen's accumulation order (oy, then ox ascending; fp32 adds)."""
    g = np.asarray(g, np.float32)
    P, OH, OW = g.shape
    y, xq = np.arange(H)[:, None], np.arange(W)[None, :]
    gx = np.zeros((P, H, W), np.float32)
    for a in (0, 1):
        oy = (y >> 1) + a
        vy = (oy <= ((y + 1) >> 1)) & (oy < OH)
        for b in (0, 1):
            ox = (xq >> 1) + b
            vx = (ox <= ((xq + 1) >> 1)) & (ox < OW)
            want = (y - 2 * oy + 1) * 3 + (xq - 2 * ox + 1)
            oyc, oxc = np.clip(oy, 0, OH - 1), np.clip(ox, 0, OW - 1)
            hit = (vy & vx)[None] & (code[:, oyc, oxc] == want[None])
            gx = (gx + np.where(hit, g[:, oyc, oxc], np.float32(0))).astype(np.float32)
    return gx


# --------------------------------------------------------------------------- f4: colour augmentation
def color_jitter(img, factors, order, apply, flip, frames=1):
    """Flip + ColorJitter of datasets/mono_dataset.py:214-256 on float images, restating
    torchvision's published float-tensor algorithms (_blend, rgb_to_grayscale, _rgb2hsv, _hsv2rgb;
    torchvision is absent; pinned to PIL's ImageEnhance / 8-bit HSV implementation -- its PIL backend,
    what the reference's loader runs -- within uint8 quantisation: tests/test_pil_pins.py).  img [samples*frames,3,H,W] -> (raw flipped, augmented)."""
    f32 = np.float32
    img = _f(img)
    raw = img.copy()
    out = img.copy()

    def grey(c):
        return (f32(0.2989) * c[0] + f32(0.587) * c[1] + f32(0.114) * c[2]).astype(f32)

    def blend(a, o, r):
        return np.clip(f32(r) * a + (f32(1.0) - f32(r)) * o, 0, 1).astype(f32)

    def hue(c, f):
        r, g, b = c
        maxc, minc = np.maximum(r, np.maximum(g, b)), np.minimum(r, np.minimum(g, b))
        eq = maxc == minc
        cr = (maxc - minc).astype(f32)
        sat = (cr / np.where(eq, f32(1), maxc)).astype(f32)
        dv = np.where(eq, f32(1), cr).astype(f32)
        rc, gc, bc = ((maxc - r) / dv).astype(f32), ((maxc - g) / dv).astype(f32), ((maxc - b) / dv).astype(f32)
        h = np.where(maxc == r, bc - gc, np.where(maxc == g, f32(2) + rc - bc, f32(4) + gc - rc)).astype(f32)
        h = np.fmod(h / f32(6) + f32(1), f32(1)).astype(f32)
        h = (h + f32(f)).astype(f32)
        h = (h - np.floor(h)).astype(f32)
        h6 = (h * f32(6)).astype(f32)
        fi = np.floor(h6)
        fr = (h6 - fi).astype(f32)
        i = fi.astype(np.int64) % 6
        v = maxc
        p = np.clip(v * (f32(1) - sat), 0, 1).astype(f32)
        q = np.clip(v * (f32(1) - sat * fr), 0, 1).astype(f32)
        t = np.clip(v * (f32(1) - sat * (f32(1) - fr)), 0, 1).astype(f32)
        a1 = np.choose(i, [v, q, p, p, t, v])
        a2 = np.choose(i, [t, v, v, q, p, p])
        a3 = np.choose(i, [p, p, t, v, v, q])
        return np.stack([a1, a2, a3], 0).astype(f32)

    for n in range(img.shape[0]):
        s = n // frames
        c = img[n, :, :, ::-1].copy() if flip[s] else img[n].copy()
        raw[n] = c
        if apply[s]:
            for op in order[s]:
                if op == 0:
                    c = blend(c, np.zeros_like(c), factors[s][0])
                elif op == 1:
                    c = blend(c, f32(np.mean(grey(c), dtype=np.float64)), factors[s][1])
                elif op == 2:
                    c = blend(c, grey(c)[None], factors[s][2])
                else:
                    c = hue(c, factors[s][3])
        out[n] = c
    return raw, out
