"""Parity tests proper (-m gpu): the HIP path, called through the C ABI, against the CPU
oracle and the committed golden vectors.

Bar: bit-exact for the integer sampling indices and argmin (and, in exact mode, for every
per-pixel fp32 map whose arithmetic the reference's CPU run reproduces: depth, cam points,
grid, SSIM, reprojection, to_optimise); <= 1e-4 relative (tensor level: max|a-b|/max|b|)
for bilinear values, reductions and gradients -- the tolerance north_star states.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import affine_case, assert_grad_close, grad_guard, load_golden, rel_err, rel_l2
from oracle import oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from mono_vifi_amd import _native
    _native.lib()   # fail loudly if the HIP library is missing: no fallback
    return torch.device("cuda:0")


def T(a, dev, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.requires_grad_(True) if grad else t


def N(t):
    return t.detach().cpu().numpy()


class FakeSelf:
    """Carries `opt` like the object the golden capture calls the reference's methods on."""

    def __init__(self, **flags):
        from types import SimpleNamespace
        self.opt = SimpleNamespace(min_depth=0.1, max_depth=100.0, no_ssim=False,
                                   avg_reprojection=False, disable_automasking=False,
                                   disparity_smoothness=1e-3)
        for k, v in flags.items():
            setattr(self.opt, k, v)


def make_self(flags_arr, noise=None):
    from mono_vifi_amd.losses import HotPathLosses

    class S(FakeSelf, HotPathLosses):
        pass
    s = S(no_ssim=bool(flags_arr[0]), avg_reprojection=bool(flags_arr[1]),
          disable_automasking=bool(flags_arr[2]))
    s.tie_break_noise = noise
    return s


# ------------------------------------------------------------------ geometry (G1)
@pytest.mark.parametrize("case", ["seed0", "seed1", "seed2", "identity", "bigrot"])
def test_geometry_stages_bit_exact(dev, case):
    from mono_vifi_amd import layers, ops
    g = load_golden("g1_geom_" + case)
    B, _, H, W = g["disp"].shape
    _, depth = layers.disp_to_depth(T(g["disp"], dev), 0.1, 100.0)
    assert np.array_equal(N(depth), g["depth"])
    cam = layers.BackprojectDepth(B, H, W).to(dev)(depth, T(g["inv_K"], dev))
    assert np.array_equal(N(cam), g["cam_points"])
    proj = layers.Project3D(B, H, W).to(dev)
    for k in range(2):
        pix = proj(cam, T(g["K"], dev), T(g[f"T{k}"], dev))
        assert np.array_equal(N(pix), g[f"pix{k}"], equal_nan=True)
        idx = N(ops.grid_sample_indices((B, 3, H, W), pix))
        assert np.array_equal(idx[..., 0], g[f"x0_{k}"])
        assert np.array_equal(idx[..., 1], g[f"y0_{k}"])
        out = layers.grid_sample_border_ac(T(g["src"][k], dev), pix)
        assert np.max(np.abs(N(out) - g[f"warped{k}"])) <= 1e-6
        warped, pix2, idx2 = ops.warp_debug(T(g["disp"], dev), T(g[f"T{k}"], dev),
                                            T(g["src"][k], dev), T(g["K"], dev), T(g["inv_K"], dev))
        assert np.array_equal(N(pix2), g[f"pix{k}"], equal_nan=True)
        assert np.array_equal(N(idx2)[..., 0], g[f"x0_{k}"])
        assert np.array_equal(N(idx2)[..., 1], g[f"y0_{k}"])
        assert np.max(np.abs(N(warped) - g[f"warped{k}"])) <= 1e-6


def test_pose_glue(dev):
    from mono_vifi_amd import layers
    g = load_golden("g5_pose")
    for tag, inv in (("fwd", False), ("inv", True)):
        aa, tr = T(g["axisangle"], dev, True), T(g["translation"], dev, True)
        M = layers.transformation_from_parameters(aa, tr, invert=inv)
        assert np.max(np.abs(N(M) - g["M_" + tag])) <= 2e-6
        (M * T(g["weight"], dev)).sum().backward()
        assert rel_err(N(aa.grad), g["grad_axisangle_" + tag]) <= TOL
        assert rel_err(N(tr.grad), g["grad_translation_" + tag]) <= TOL
    R = layers.rot_from_axisangle(T(g["axisangle"], dev))
    assert np.max(np.abs(N(R)[:, :3, :3] - g["M_fwd"][:, :3, :3])) <= 2e-6


# ------------------------------------------------------------------ photometric (G2, G6)
def test_ssim_and_smooth_standalone(dev):
    from mono_vifi_amd import layers
    g = load_golden("g6_ssim_smooth")
    x, y = T(g["x"], dev, True), T(g["y"], dev, True)
    s = layers.SSIM().to(dev)(x, y)
    assert np.array_equal(N(s), g["ssim"])
    (s * T(g["weight"], dev)).sum().backward()
    assert rel_err(N(x.grad), g["grad_x"]) <= TOL
    assert rel_err(N(y.grad), g["grad_y"]) <= TOL
    s2 = layers.SSIM()(T(g["x_near"], dev), T(g["y"], dev))
    assert np.array_equal(N(s2), g["ssim_near"])   # catastrophic-cancellation regime
    disp = T(g["disp"], dev, True)
    sm = layers.get_smooth_loss(disp, T(g["y"], dev))
    assert abs(float(sm.detach()) - float(g["smooth"])) <= 2e-6 * abs(float(g["smooth"]))
    sm.backward()
    assert rel_err(N(disp.grad), g["grad_disp"]) <= 1e-5


@pytest.mark.parametrize("case", ["default", "mask", "no_ssim", "avg", "noauto", "noauto_mask"])
def test_losses_base_forward(dev, case):
    g = load_golden("g2_photo_" + case)
    noise = T(g["noise"], dev) if "noise" in g and not g["flags"][2] else None
    s = make_self(g["flags"], noise)
    tgt = T(g["tgt"], dev)
    warped = [T(g["warped"][k], dev) for k in range(2)]
    srcs = [T(g["src"][k], dev) for k in range(2)]
    for k in range(2):
        rp = s.compute_reprojection_loss(warped[k], tgt)
        assert np.array_equal(N(rp)[:, 0], g["rp"][:, k])
        idl = s.compute_reprojection_loss(srcs[k], tgt)
        assert np.array_equal(N(idl)[:, 0], g["idl"][:, k])
    mask = T(g["mask_rec"], dev) if case in ("mask", "noauto_mask") else None
    from mono_vifi_amd import ops
    disp = T(g["disp"], dev)
    S = 2
    src_args = srcs if not g["flags"][2] else []
    loss, auto_mask, to_opt, argmin = ops.LossesBase.apply(
        disp, tgt, mask, noise, S, s._loss_flags(), 1e-3, *warped, *src_args)
    assert np.array_equal(N(to_opt).reshape(g["to_opt"].shape), g["to_opt"])
    if "idxs" in g:
        assert np.array_equal(N(argmin).astype(np.int32), g["idxs"])
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    loss2, am2 = s.compute_losses_base({("disp", 0): disp}, tgt, warped, srcs, mask)
    assert float(loss2) == float(loss)
    if "auto_mask" in g:
        assert np.array_equal(N(am2), g["auto_mask"])
    else:
        assert am2 is None


# ------------------------------------------------------------------ gradients (G3)
@pytest.mark.parametrize("case", ["default", "mask", "no_ssim", "avg", "noauto"])
@pytest.mark.parametrize("path", ["staged", "fused"])
def test_unit_gradients(dev, case, path):
    from mono_vifi_amd import layers
    g = load_golden("g3_grad_" + case)
    noise = T(g["noise"], dev) if "noise" in g and not g["flags"][2] else None
    s = make_self(g["flags"], noise)
    aa, tr = T(g["axisangle"], dev, True), T(g["translation"], dev, True)
    disp = T(g["disp"], dev, True)
    tgt = T(g["tgt"], dev)
    srcs = [T(g["src"][k], dev) for k in range(2)]
    K, inv_K = T(g["K"], dev), T(g["inv_K"], dev)
    mask = T(g["mask_rec"], dev) if case == "mask" else None
    poses = []
    for k in range(2):
        M = layers.transformation_from_parameters(aa[k], tr[k], invert=(k == 1))
        M.retain_grad()
        poses.append(M)
    warped = []
    if path == "staged":
        for k in range(2):
            w = s.generate_images_pred({("disp", 0): disp}, poses[k], srcs[k], K, inv_K)
            w.retain_grad()
            warped.append(w)
        loss, auto_mask = s.compute_losses_base({("disp", 0): disp}, tgt, warped, srcs, mask)
    else:
        loss, auto_mask = s.compute_unit({("disp", 0): disp}, tgt, poses, srcs, K, inv_K, mask,
                                         want_auto_mask=True)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    if "auto_mask" in g:
        assert np.array_equal(N(auto_mask), g["auto_mask"])
    if path == "staged":
        for k in range(2):
            # poses here come from the GPU's sin/cos (<= 2e-6 from the CPU's), so the warped
            # values move by a few ulp more than in the stored-pose geometry test
            assert np.max(np.abs(N(warped[k]) - g["warped"][k])) <= 1e-5
            assert rel_err(N(warped[k].grad), g["grad_warped"][k]) <= TOL
    assert rel_err(N(disp.grad), g["grad_disp"]) <= TOL
    gT = np.stack([N(p.grad) for p in poses], 0)
    assert rel_err(gT, g["grad_T"]) <= TOL
    assert rel_err(N(aa.grad), g["grad_axisangle"]) <= TOL
    assert rel_err(N(tr.grad), g["grad_translation"]) <= TOL


# ------------------------------------------------------------------ full size (G4)
@pytest.fixture
def unit_kernel(request):
    """Selects which tile kernels a unit with gradients runs: "fwdbwd" = mvf_unit_fwdbwd (one
    kernel, what the training step runs), "separate" = mvf_unit_fwd + mvf_unit_bwd."""
    from mono_vifi_amd import ops
    old = ops.UNIT_FWDBWD
    ops.UNIT_FWDBWD = request.param == "fwdbwd"
    yield request.param
    ops.UNIT_FWDBWD = old


BOTH_KERNELS = pytest.mark.parametrize("unit_kernel", ["fwdbwd", "separate"], indirect=True)


@BOTH_KERNELS
@pytest.mark.parametrize("cfg", ["C1", "C2", "C4", "C5"])
def test_fullsize_unit(dev, cfg, unit_kernel):
    """BASELINE.json shapes through the fused unit -- with the training kernel
    (mvf_unit_fwdbwd) and with the separate forward / backward kernels: index-map SHA-256,
    loss, auto-mask and sampled gradients against what the reference produced on the same
    seeded inputs, and the whole tensors (argmin, grad_disp, grad_T) against the oracle."""
    from mono_vifi_amd import ops, synthetic
    g = load_golden("g4_full_" + cfg)
    B, H, W = (int(v) for v in g["shape"])
    inp = synthetic.unit_inputs(int(g["seed"]), B, H, W, with_mask=True)
    disp = T(inp["disp"], dev, True)
    Tt = T(g["T"], dev, True)
    cfgt = (2, 0, 1e-3, 0.1, 100.0, 1e-7, True, True)
    loss, auto_mask, argmin, idx, _ = ops.Unit.apply(
        disp, T(inp["tgt"], dev), Tt, T(inp["K"], dev), T(inp["inv_K"], dev),
        T(inp["mask_rec"], dev), T(inp["noise"], dev), cfgt, T(inp["src"][0], dev),
        T(inp["src"][1], dev))
    loss.backward()
    idx = N(idx)
    for k in range(2):
        assert hashlib.sha256(np.ascontiguousarray(idx[k, ..., 0]).tobytes()).hexdigest() == str(g[f"sha_x0_{k}"])
        assert hashlib.sha256(np.ascontiguousarray(idx[k, ..., 1]).tobytes()).hexdigest() == str(g[f"sha_y0_{k}"])
    n = B * H * W
    sidx = g["sample_idx"]
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    am = N(auto_mask).reshape(n)
    assert np.array_equal(am[sidx], g["auto_mask_s"])
    assert abs(am.mean() - float(g["auto_mask_mean"])) <= 1e-6
    gd = N(disp.grad)
    # the training kernel against the REFERENCE's own gradients: 1e-4 on the tensor AND per element (north_star's
    # bar; measured worst 5.3e-5 -- as close to the reference as the fp64-folding oracle is)
    assert_grad_close(gd.reshape(n)[sidx], g["grad_disp_s"], TOL, "grad_disp vs reference (sampled)",
                      max_tol=TOL if unit_kernel == "fwdbwd" else None)
    assert abs(np.linalg.norm(gd.astype(np.float64)) - float(g["grad_disp_norm"])) <= TOL * float(g["grad_disp_norm"])
    # whole tensors vs the oracle (fp64 reductions on both sides)
    ref = O.unit(inp["disp"], inp["tgt"], inp["src"], g["T"], inp["K"], inp["inv_K"], inp["noise"],
                 inp["mask_rec"], 0, want_grads=True)
    assert np.array_equal(N(argmin).astype(np.int32), ref["idx"])
    # whole tensor vs the oracle: two fp32 evaluation orders of a cancelling adjoint; the per-element guard is this
    # configuration's measured worst x 1.5 (conftest.GRAD_GUARD), the count of pixels beyond 1e-4 is tracked
    assert_grad_close(gd, ref["grad_disp"], TOL, "grad_disp vs oracle", max_tol=1e-3)
    if unit_kernel == "fwdbwd":
        grad_guard("oracle", cfg, gd, ref["grad_disp"], "grad_disp")
    assert rel_err(N(Tt.grad), ref["grad_T"]) <= TOL
    # the reference itself reduces grad_P in fp32; its own value is only good to ~5e-3
    assert rel_err(N(Tt.grad), g["grad_T"]) <= 5e-3


@BOTH_KERNELS
@pytest.mark.parametrize("name", ["no_ssim", "avg", "noauto"])
def test_fullsize_unit_flag_sets(dev, name, unit_kernel):
    """C2 (12 x 192 x 640) under --no_ssim / --avg_reprojection / --disable_automasking against what the REFERENCE
    produced on the same seeded inputs (loss, sampled auto-mask, 4,096 sampled gradients: 1e-4 on the tensor and,
    for the training kernel, per element) and the whole tensors against the oracle."""
    from mono_vifi_amd import ops, synthetic
    g = load_golden("g4_flags_C2_" + name)
    B, H, W = (int(v) for v in g["shape"])
    flags = int(g["flags"][0]) * 1 + int(g["flags"][1]) * 2 + int(g["flags"][2]) * 4
    use_mask = bool(int(g["use_mask"]))
    inp = synthetic.unit_inputs(int(g["seed"]), B, H, W, with_mask=use_mask)
    noise_np = np.ascontiguousarray(inp["noise"][:, :1] if flags & 2 else inp["noise"])
    mask_np = inp["mask_rec"] if use_mask else None
    disp = T(inp["disp"], dev, True)
    Tt = T(g["T"], dev, True)
    cfgt = (2, flags, 1e-3, 0.1, 100.0, 1e-7, True, False)
    loss, auto_mask, argmin, _, _ = ops.Unit.apply(
        disp, T(inp["tgt"], dev), Tt, T(inp["K"], dev), T(inp["inv_K"], dev),
        T(mask_np, dev) if use_mask else None, None if flags & 4 else T(noise_np, dev), cfgt,
        T(inp["src"][0], dev), T(inp["src"][1], dev))
    loss.backward()
    n = B * H * W
    sidx = g["sample_idx"]
    assert abs(float(loss) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    if "auto_mask_s" in g:
        am = N(auto_mask).reshape(n)
        assert np.array_equal(am[sidx], g["auto_mask_s"])
        assert abs(am.mean() - float(g["auto_mask_mean"])) <= 1e-6
    gd = N(disp.grad)
    # 1e-4 on the tensor (relative L2) and, for the training kernel, 1e-4 PER ELEMENT of the tensor max -- against the
    # reference's fp32 samples AND against the same reference code evaluated in float64 (g4_f64_C2_*, the arbiter).
    # Round 4 held this at 2e-4 and blamed the SSIM adjoint; the cause was the metric: `assert_grad_close` on 4,096
    # SAMPLES divides by the largest SAMPLE (2-2.6 x smaller than the tensor's largest element under these flag sets),
    # not by the tensor max the bar is stated in.  With the right denominator: kernel vs reference fp32 <= 5.7e-5, vs
    # float64 <= 4.4e-5; the reference's own fp32 run is 3.0-8.7e-5 from its float64 self (worst samples: far pixels,
    # depth 36-92 m, where grad_disp = -range depth^2 dL/d depth amplifies the rounding of a cancelling parallax term):
    # profiles/r05_grad_error_report.txt, tools/grad_error_report.py.
    d64 = load_golden("g4_f64_C2_" + name)
    assert np.array_equal(d64["sample_idx"], sidx)
    gmax = float(d64["grad_disp_max64"])
    got_s = gd.reshape(n)[sidx].astype(np.float64)
    assert rel_l2(got_s, g["grad_disp_s"]) <= TOL
    per_elem = TOL if unit_kernel == "fwdbwd" else 10 * TOL
    assert np.abs(got_s - g["grad_disp_s"]).max() <= per_elem * gmax, "vs reference fp32 (sampled), of the tensor max"
    same_fn = ~d64["selection_differs_near"].astype(bool)      # (a flipped argmin in the fp64 run: another function)
    assert np.abs(got_s - d64["grad_disp_s64"])[same_fn].max() <= per_elem * gmax, "vs reference float64 (sampled)"
    assert abs(np.linalg.norm(gd.astype(np.float64)) - float(g["grad_disp_norm"])) <= TOL * float(g["grad_disp_norm"])
    ref = O.unit(inp["disp"], inp["tgt"], inp["src"], g["T"], inp["K"], inp["inv_K"], noise_np, mask_np, flags,
                 want_grads=True)
    am = N(argmin).astype(np.int32)
    am[am == 255] = -1
    assert np.array_equal(am, ref["idx"])
    assert_grad_close(gd, ref["grad_disp"], TOL, "grad_disp vs oracle", max_tol=1e-3)
    if unit_kernel == "fwdbwd":
        grad_guard("oracle", "C2_" + name, gd, ref["grad_disp"], "grad_disp")
    assert rel_err(N(Tt.grad), ref["grad_T"]) <= TOL
    assert rel_err(N(Tt.grad), g["grad_T"]) <= 5e-3


@pytest.mark.parametrize("case", ["C1", "C2", "C4", "C5", "C2_no_ssim", "C2_avg", "C2_noauto"])
def test_fullsize_gradients_vs_double_adjoint(dev, case):
    """The training kernel's WHOLE gradient tensors at the BASELINE shapes against the oracle's adjoint evaluated in
    double (every decision the fp32 forward's, every value in double; pinned to the reference evaluated in float64 by
    tests/test_oracle_golden.py::test_double_adjoint_against_reference_float64).  Relative L2 <= 5e-5 (measured
    2.2-2.9e-5); per element of the tensor max: at most one pixel in 10,000 beyond 1e-4 (measured 0-48 of 0.5-2.6 M) and
    none beyond 5e-4 (measured worst 1.5e-4 at default flags, 4.1e-4 under --no_ssim).  The pixels beyond 1e-4 are not
    the kernel's: the fp32 oracle misses the double evaluation by as much (1.4-4.6e-4) at the same kind of pixel --
    far, high-contrast ones, where the adjoint's cancelling parallax terms amplify the fp32 rounding of the FORWARD
    values (u, v, z, the camera point) every fp32 evaluation, the reference's own autograd included, differentiates
    through (tools/grad_vs_f64_adjoint.py)."""
    from mono_vifi_amd import ops, synthetic
    flagset = case[3:] if "_" in case else None
    g = load_golden("g4_flags_C2_" + flagset if flagset else "g4_full_" + case)
    B, H, W = (int(v) for v in g["shape"])
    if flagset is None:
        flags, use_mask = 0, True
    else:
        flags = int(g["flags"][0]) * 1 + int(g["flags"][1]) * 2 + int(g["flags"][2]) * 4
        use_mask = bool(int(g["use_mask"]))
    inp = synthetic.unit_inputs(int(g["seed"]), B, H, W, with_mask=use_mask)
    noise_np = np.ascontiguousarray(inp["noise"][:, :1] if flags & 2 else inp["noise"])
    mask_np = inp["mask_rec"] if use_mask else None
    disp = T(inp["disp"], dev, True)
    Tt = T(g["T"], dev, True)
    loss, _, _, _, _ = ops.Unit.apply(
        disp, T(inp["tgt"], dev), Tt, T(inp["K"], dev), T(inp["inv_K"], dev),
        T(mask_np, dev) if use_mask else None, None if flags & 4 else T(noise_np, dev),
        (2, flags, 1e-3, 0.1, 100.0, 1e-7, True, False), T(inp["src"][0], dev), T(inp["src"][1], dev))
    loss.backward()
    ref = O.unit(inp["disp"], inp["tgt"], inp["src"], g["T"], inp["K"], inp["inv_K"], noise_np, mask_np, flags,
                 want_grads=True, adjoint64=True)
    gd = N(disp.grad).astype(np.float64)
    r64 = ref["grad_disp64"]
    mx = np.abs(r64).max()
    e = np.abs(gd - r64)
    assert np.linalg.norm(gd - r64) <= 5e-5 * np.linalg.norm(r64)
    _, beyond = grad_guard("f64", case, gd, r64, "grad_disp")
    assert beyond <= 1e-4 * e.size
    assert np.abs(N(Tt.grad) - ref["grad_T64"]).max() <= 1e-4 * np.abs(ref["grad_T64"]).max()


# ------------------------------------------------------------------ ragged shapes vs oracle
RAGGED = [(1, 2, 2), (1, 3, 5), (2, 14, 62), (1, 15, 63), (1, 16, 64), (2, 17, 65), (1, 31, 129),
          (3, 33, 70), (1, 64, 200)]


@BOTH_KERNELS
@pytest.mark.parametrize("shape", RAGGED)
@pytest.mark.parametrize("flags", [0, 1, 2, 4, 6])
def test_ragged_shapes_vs_oracle(dev, shape, flags, unit_kernel):
    """Tile-edge and tiny shapes (partial tiles, reflect halo == whole image, 1-tile
    images) for every flag combination, both kernel routes, against the oracle (indices,
    argmin, loss, gradients)."""
    from mono_vifi_amd import ops, synthetic
    B, H, W = shape
    inp = synthetic.unit_inputs(900 + H * W + flags, B, H, W, pose_scale=0.03, with_mask=True)
    use_mask = not (flags & 4 and flags & 2)
    mask_np = inp["mask_rec"] if use_mask else None
    T_np = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1))
                     for k in range(2)], 0)
    noise_np = inp["noise"][:, :1] if flags & 2 else inp["noise"]
    ref = O.unit(inp["disp"], inp["tgt"], inp["src"], T_np, inp["K"], inp["inv_K"],
                 np.ascontiguousarray(noise_np), mask_np, flags, want_grads=True)
    noise = None if flags & 4 else T(np.ascontiguousarray(noise_np), dev)
    mask = T(mask_np, dev) if use_mask else None
    disp = T(inp["disp"], dev, True)
    Tt = T(T_np, dev, True)
    cfgt = (2, flags, 1e-3, 0.1, 100.0, 1e-7, True, True)
    loss, auto_mask, argmin, idx, _ = ops.Unit.apply(
        disp, T(inp["tgt"], dev), Tt, T(inp["K"], dev), T(inp["inv_K"], dev), mask, noise, cfgt,
        T(inp["src"][0], dev), T(inp["src"][1], dev))
    loss.backward()
    idx = N(idx)
    for k in range(2):
        assert np.array_equal(idx[k, ..., 0], ref["x0"][k])
        assert np.array_equal(idx[k, ..., 1], ref["y0"][k])
    am = N(argmin).astype(np.int32)
    am[am == 255] = -1
    assert np.array_equal(am, ref["idx"])
    assert abs(float(loss) - ref["loss"]) <= 1e-5 * abs(ref["loss"])
    assert_grad_close(N(disp.grad), ref["grad_disp"], TOL, "grad_disp vs oracle")
    assert rel_err(N(Tt.grad), ref["grad_T"]) <= TOL


# ------------------------------------------------------------------ other source counts
@pytest.mark.parametrize("S", [1, 3, 4])
@pytest.mark.parametrize("flags", [0, 2, 4])
@pytest.mark.parametrize("path", ["fused", "staged"])
def test_source_counts_vs_oracle(dev, S, flags, path):
    """1, 3 and 4 source frames (odd counts exercise the mixed warped/identity candidate pair
    of the tile engine; the reference's loops are generic in len(frame_ids))."""
    from mono_vifi_amd import ops, synthetic
    B, H, W = 2, 33, 70
    inp = synthetic.unit_inputs(300 + 10 * S + flags, B, H, W, num_src=S, pose_scale=0.03, with_mask=True)
    T_np = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k % 2 == 1))
                     for k in range(S)], 0)
    noise_np = np.ascontiguousarray(inp["noise"][:, :1] if flags & 2 else inp["noise"])
    ref = O.unit(inp["disp"], inp["tgt"], inp["src"], T_np, inp["K"], inp["inv_K"], noise_np,
                 inp["mask_rec"], flags, want_grads=True)
    noise = None if flags & 4 else T(noise_np, dev)
    disp = T(inp["disp"], dev, True)
    Tt = T(T_np, dev, True)
    srcs = [T(inp["src"][k], dev) for k in range(S)]
    if path == "fused":
        cfgt = (S, flags, 1e-3, 0.1, 100.0, 1e-7, True, True)
        loss, _, argmin, idx, _ = ops.Unit.apply(disp, T(inp["tgt"], dev), Tt, T(inp["K"], dev),
                                                 T(inp["inv_K"], dev), T(inp["mask_rec"], dev), noise,
                                                 cfgt, *srcs)
        idx = N(idx)
        for k in range(S):
            assert np.array_equal(idx[k, ..., 0], ref["x0"][k])
            assert np.array_equal(idx[k, ..., 1], ref["y0"][k])
    else:
        s = make_self([flags & 1, (flags >> 1) & 1, (flags >> 2) & 1], noise)
        ws = [s.generate_images_pred({("disp", 0): disp}, Tt[k], srcs[k], T(inp["K"], dev),
                                     T(inp["inv_K"], dev)) for k in range(S)]
        if S == 1 and flags & 4:
            # single candidate + mask_rec: the reference's in-place `to_optimise *= mask_rec[:,0]`
            # cannot broadcast (train.py:1030-1036); the mirror raises the same error
            with pytest.raises(RuntimeError):
                s.compute_losses_base({("disp", 0): disp}, T(inp["tgt"], dev), ws, srcs,
                                      T(inp["mask_rec"], dev))
            return
        loss, _ = s.compute_losses_base({("disp", 0): disp}, T(inp["tgt"], dev), ws, srcs,
                                        T(inp["mask_rec"], dev))
        argmin = None
    loss.backward()
    if argmin is not None:
        am = N(argmin).astype(np.int32)
        am[am == 255] = -1
        assert np.array_equal(am.reshape(ref["idx"].shape), ref["idx"])
    assert abs(float(loss) - ref["loss"]) <= 1e-5 * abs(ref["loss"])
    assert_grad_close(N(disp.grad), ref["grad_disp"], TOL, "grad_disp vs oracle")
    assert rel_err(N(Tt.grad), ref["grad_T"]) <= TOL


# ------------------------------------------------------------------ forward+backward in one kernel
@pytest.mark.parametrize("shape", RAGGED + [(12, 192, 640)])
@pytest.mark.parametrize("flags", [0, 1, 2, 4, 6])
@pytest.mark.parametrize("S", [1, 2])
def test_unit_fwdbwd_equals_separate_kernels(dev, shape, flags, S):
    """mvf_unit_fwdbwd (what the training step runs: loss and gradients from one tile kernel)
    against mvf_unit_fwd + mvf_unit_bwd: same argmin and auto-mask, loss to 2e-6, grad_disp /
    grad_T within the gradient tolerance, also under a non-unit upstream gradient; and against
    the oracle at the small shapes."""
    from mono_vifi_amd import ops, synthetic
    B, H, W = shape
    inp = synthetic.unit_inputs(1300 + H * W + flags + S, B, H, W, num_src=S, pose_scale=0.03,
                                with_mask=True, disp_mode="smooth" if H * W > 10000 else "noise")
    use_mask = not (flags & 4 and (flags & 2 or S == 1))
    mask_np = inp["mask_rec"] if use_mask else None
    T_np = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1))
                     for k in range(S)], 0)
    noise_np = np.ascontiguousarray(inp["noise"][:, :1] if flags & 2 else inp["noise"])
    noise = None if flags & 4 else T(noise_np, dev)
    mask = T(mask_np, dev) if use_mask else None
    srcs = [T(inp["src"][k], dev) for k in range(S)]
    cfgt = (S, flags, 1e-3, 0.1, 100.0, 1e-7, True, False)
    res = {}
    for mode in (True, False):
        ops.UNIT_FWDBWD = mode
        try:
            disp, Tt = T(inp["disp"], dev, True), T(T_np, dev, True)
            loss, auto_mask, argmin, _, parts = ops.Unit.apply(
                disp, T(inp["tgt"], dev), Tt, T(inp["K"], dev), T(inp["inv_K"], dev), mask, noise,
                cfgt, *srcs)
            (loss * 1.0).backward()
            d1, t1 = N(disp.grad).copy(), N(Tt.grad).copy()
            disp.grad = None
            Tt.grad = None
            # a second graph with a non-unit upstream gradient
            disp2, Tt2 = T(inp["disp"], dev, True), T(T_np, dev, True)
            loss2 = ops.Unit.apply(disp2, T(inp["tgt"], dev), Tt2, T(inp["K"], dev), T(inp["inv_K"], dev),
                                   mask, noise, cfgt, *srcs)[0]
            (loss2 * 4.0).backward()
            res[mode] = (float(loss.detach()), N(argmin), N(auto_mask), d1, t1, N(disp2.grad), N(parts))
        finally:
            ops.UNIT_FWDBWD = True
    fb, sep = res[True], res[False]
    assert abs(fb[0] - sep[0]) <= 2e-6 * abs(sep[0])
    assert np.array_equal(fb[1], sep[1]) and np.array_equal(fb[2], sep[2])     # argmin, auto-mask: exact
    # gradients: the one-kernel route evaluates the (tolerance-level) adjoint in its own order
    # (SSIM partials from the forward's window statistics, reciprocal of z shared by the
    # perspective adjoint), so the two routes agree as two fp32 evaluation orders do
    assert_grad_close(fb[3], sep[3], TOL, "grad_disp: one kernel vs separate kernels")
    assert rel_err(fb[4], sep[4]) <= TOL
    # upstream gradient 4 (a power of two commutes with every rounding): scaled afterwards (one
    # kernel) == scaled inside (separate kernels)
    assert_grad_close(fb[5], sep[5], TOL, "grad_disp under upstream gradient 4")
    assert np.array_equal(fb[5], fb[3] * 4.0)
    assert np.allclose(fb[6], sep[6], rtol=2e-6, atol=1e-9)
    if H * W <= 20000:
        ref = O.unit(inp["disp"], inp["tgt"], inp["src"], T_np, inp["K"], inp["inv_K"], noise_np,
                     mask_np, flags, want_grads=True)
        am = fb[1].astype(np.int32)
        am[am == 255] = -1
        assert np.array_equal(am, ref["idx"])
        assert abs(fb[0] - ref["loss"]) <= 1e-5 * abs(ref["loss"])
        assert_grad_close(fb[3], ref["grad_disp"], TOL, "grad_disp vs oracle")
        assert rel_err(fb[4], ref["grad_T"]) <= TOL


# ------------------------------------------------------------------ properties at full size
def test_properties_fullsize(dev):
    """Size-independent properties at the benchmark shape (B12 640x192): determinism of the
    whole unit, linearity of the backward in the upstream gradient, identity-pose warp
    reproduces the source image, staged == fused."""
    from mono_vifi_amd import ops, synthetic
    B, H, W = 12, 192, 640
    inp = synthetic.unit_inputs(77, B, H, W, with_mask=True)
    T_np = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1))
                     for k in range(2)], 0)
    tens = dict(tgt=T(inp["tgt"], dev), K=T(inp["K"], dev), inv_K=T(inp["inv_K"], dev),
                mask=T(inp["mask_rec"], dev), noise=T(inp["noise"], dev),
                s0=T(inp["src"][0], dev), s1=T(inp["src"][1], dev))
    cfgt = (2, 0, 1e-3, 0.1, 100.0, 1e-7, True, False)

    def run(scale):
        disp = T(inp["disp"], dev, True)
        Tt = T(T_np, dev, True)
        loss, am, argmin, _, _ = ops.Unit.apply(disp, tens["tgt"], Tt, tens["K"], tens["inv_K"],
                                                tens["mask"], tens["noise"], cfgt, tens["s0"],
                                                tens["s1"])
        (loss * scale).backward()
        return float(loss), N(disp.grad), N(Tt.grad), N(argmin)

    l1, gd1, gT1, a1 = run(1.0)
    l2, gd2, gT2, a2 = run(1.0)
    assert l1 == l2 and np.array_equal(gd1, gd2) and np.array_equal(gT1, gT2) and np.array_equal(a1, a2)
    l3, gd3, gT3, _ = run(4.0)   # power of two: exact scaling
    assert np.array_equal(gd3, gd1 * 4.0) and rel_err(gT3, gT1 * 4.0) <= 1e-6

    # identity pose + any depth: every pixel samples itself
    eye = torch.eye(4, device=dev).repeat(B, 1, 1)
    warped, pix, idx = ops.warp_debug(T(inp["disp"], dev), eye, tens["s0"], tens["K"], tens["inv_K"])
    # inv_K = pinv(K) in fp32: the round trip lands within ~1e-3 px of the pixel centre
    assert float((warped - tens["s0"]).abs().max()) <= 1e-3

    # staged path == fused path
    from mono_vifi_amd.losses import HotPathLosses

    class S(FakeSelf, HotPathLosses):
        pass
    s = S()
    s.tie_break_noise = tens["noise"]
    disp = T(inp["disp"], dev, True)
    Tt = T(T_np, dev, True)
    ws = [s.generate_images_pred({("disp", 0): disp}, Tt[k], [tens["s0"], tens["s1"]][k], tens["K"],
                                 tens["inv_K"]) for k in range(2)]
    loss, _ = s.compute_losses_base({("disp", 0): disp}, tens["tgt"], ws, [tens["s0"], tens["s1"]],
                                    tens["mask"])
    loss.backward()
    assert abs(float(loss) - l1) <= 1e-6 * abs(l1)
    # two fp32 evaluation orders of the same adjoint (the one-kernel route derives the SSIM
    # partials from the forward's window statistics): gradient tolerance, not bit equality
    assert_grad_close(N(disp.grad), gd1, TOL, "grad_disp staged vs fused")
    assert rel_err(N(Tt.grad), gT1) <= TOL


def test_error_behaviour(dev):
    """Errors the reference raises at this boundary are kept."""
    from mono_vifi_amd import layers
    bp = layers.BackprojectDepth(2, 8, 8).to(dev)
    with pytest.raises(RuntimeError):
        bp(torch.rand(3, 1, 8, 8, device=dev), torch.eye(4, device=dev).repeat(3, 1, 1))
    with pytest.raises(RuntimeError):   # no CPU fallback
        layers.SSIM()(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))


# ------------------------------------------------------------------ f1: flow warp (G7)
@pytest.fixture
def flow_bwd(request):
    """grad_img route of the standalone flow warp: "gather" (deterministic, the default) or
    "scatter" (float atomics)."""
    from mono_vifi_amd import ops
    old = ops.FLOW_WARP_BWD_GATHER
    ops.FLOW_WARP_BWD_GATHER = request.param == "gather"
    yield request.param
    ops.FLOW_WARP_BWD_GATHER = old


BOTH_FLOW_BWD = pytest.mark.parametrize("flow_bwd", ["gather", "scatter"], indirect=True)


@BOTH_FLOW_BWD
@pytest.mark.parametrize("case", ["a", "b", "big"])
def test_flow_warp(dev, case, flow_bwd):
    from mono_vifi_amd import ops
    g = load_golden("g7_flow_" + case)
    assert np.array_equal(N(ops._linspace(g["img"].shape[3], dev)), g["xs"])
    img, flow = T(g["img"], dev, True), T(g["flow"], dev, True)
    out = ops.flow_warp(img, flow)
    idx = N(ops.flow_warp_indices(g["img"].shape, flow.detach()))
    assert np.array_equal(idx[..., 0], g["x0"]) and np.array_equal(idx[..., 1], g["y0"])
    assert np.max(np.abs(N(out) - g["out"])) <= 1e-6
    (out * T(g["weight"], dev)).sum().backward()
    assert rel_err(N(img.grad), g["grad_img"]) <= 1e-5
    assert rel_err(N(flow.grad), g["grad_flow"]) <= 1e-5


@BOTH_FLOW_BWD
def test_flow_warp_feature_pyramid_vs_oracle(dev, flow_bwd):
    """Feature-pyramid shapes of the fusion module: ResNet18 at 640x192 (64..512 channels, 96x320 ..
    6x20), the 512x192 Cityscapes pyramid of the HRNet18 encoder (C5) and Lite-Mono's 3-scale
    pyramid at 1024x320 (C4)."""
    from mono_vifi_amd import ops
    rng = np.random.default_rng(71)
    for (C, H, W) in ((64, 96, 320), (128, 24, 80), (512, 6, 20), (3, 192, 640),
                      (64, 96, 256), (18, 48, 128), (144, 6, 16),        # C5: 512x192
                      (48, 80, 256), (80, 40, 128), (128, 20, 64)):      # C4: Lite-Mono 1024x320
        img = rng.random((2, C, H, W)).astype(np.float32)
        flow = (4 * rng.standard_normal((2, 2, H, W))).astype(np.float32)
        wgt = rng.standard_normal((2, C, H, W)).astype(np.float32)
        xs, ys = N(ops._linspace(W, dev)), N(ops._linspace(H, dev))
        ref, x0, y0 = O.flow_warp(img, flow, xs, ys, want_idx=True)
        ti = T(img, dev, True)
        out = ops.flow_warp(ti, T(flow, dev))
        idx = N(ops.flow_warp_indices(img.shape, T(flow, dev)))
        assert np.array_equal(idx[..., 0], x0) and np.array_equal(idx[..., 1], y0)
        assert np.max(np.abs(N(out) - ref)) <= 1e-6
        (out * T(wgt, dev)).sum().backward()
        g_img, _ = O.flow_warp_bwd(img, flow, xs, ys, wgt)
        assert rel_err(N(ti.grad), g_img) <= 1e-5
        if flow_bwd == "gather":      # bit-reproducible: sorted inverse tap lists, no float atomics
            t2 = T(img, dev, True)
            (ops.flow_warp(t2, T(flow, dev)) * T(wgt, dev)).sum().backward()
            assert torch.equal(t2.grad, ti.grad)


# ------------------------------------------------------------------ f2: SI-log loss (G8)
def test_silog(dev):
    from mono_vifi_amd import ops
    g = load_golden("g8_silog")
    for tag, m in (("nomask", None), ("mask", g["mask"])):
        p, t = T(g["pred"], dev, True), T(g["target"], dev, True)
        loss = ops.silog_loss(p, t, T(m, dev) if m is not None else None, 0.5)
        (loss * 3.0).backward()
        assert abs(float(loss.detach()) - float(g["loss_" + tag])) <= 2e-6 * abs(float(g["loss_" + tag]))
        assert rel_err(N(p.grad), g["grad_pred_" + tag]) <= 1e-5
        assert rel_err(N(t.grad), g["grad_target_" + tag]) <= 1e-5
    # full size vs the oracle
    rng = np.random.default_rng(81)
    pred = (0.1 + 50 * rng.random((12, 1, 192, 640))).astype(np.float32)
    target = (pred * (0.5 + rng.random(pred.shape))).astype(np.float32)
    mask = (rng.random(pred.shape) > 0.2).astype(np.float32)
    ref, gp, gt = O.silog(pred, target, mask, 0.5, want_grads=True)
    p, t = T(pred, dev, True), T(target, dev, True)
    loss = ops.silog_loss(p, t, T(mask, dev))
    loss.backward()
    assert abs(float(loss.detach()) - ref) <= 1e-5 * abs(ref)
    assert rel_err(N(p.grad), gp) <= 1e-4 and rel_err(N(t.grad), gt) <= 1e-4


def test_silog_many_equals_single_calls(dev):
    """Round 5: the nine SI-log losses of a step (train.py:813-815, 868-882) as one launch each way
    (mvf_silog_many_fwd / _bwd): per job the bits of mvf_silog_fwd / _bwd, the total = the sum in job order;
    predictions that are strided views of one interleaved tensor are read in place; a tensor used by two jobs gets
    the sum of its two gradients."""
    from mono_vifi_amd import ops
    rng = np.random.default_rng(83)
    B, H, W, G = 3, 40, 72, 4
    base = T((0.1 + 50 * rng.random((B * G, 1, H, W))).astype(np.float32), dev, True)
    views = torch.unbind(base.view(B, G, 1, H, W), 1)                     # strided: image stride G*H*W
    tgt = [T((0.1 + 50 * rng.random((B, 1, H, W))).astype(np.float32), dev, True) for _ in range(2)]
    mask = T((rng.random((B, 1, H, W)) > 0.3).astype(np.float32), dev)
    jobs = [(views[0], tgt[0], None), (views[2], tgt[0], mask), (views[2], tgt[1], mask), (tgt[1], views[3], None)]
    total, losses = ops.silog_many(jobs, 0.5)
    (total * 1.25).backward()
    got = (float(total.detach()), N(losses), N(base.grad), [N(t.grad) for t in tgt])
    base2 = base.detach().clone().requires_grad_(True)
    views2 = torch.unbind(base2.view(B, G, 1, H, W), 1)
    tgt2 = [t.detach().clone().requires_grad_(True) for t in tgt]
    sub = {id(views[k]): views2[k] for k in range(G)}
    sub.update({id(tgt[k]): tgt2[k] for k in range(2)})
    singles = [ops.silog_loss(sub[id(p)], sub[id(t)], m, 0.5) for p, t, m in jobs]
    seq = np.float32(0.0)
    for l in singles:
        seq = np.float32(seq + np.float32(float(l.detach())))
    (sum(singles) * 1.25).backward()
    assert np.array_equal(got[1], np.array([float(l.detach()) for l in singles], np.float32))
    assert np.float32(got[0]) == seq
    assert rel_err(got[2], N(base2.grad)) <= 1e-6            # (two-job tensors: the order of the two adds may differ)
    for a, b in zip(got[3], tgt2):
        assert rel_err(a, N(b.grad)) <= 1e-6
    # group 1 of the interleaved tensor is read by no job
    assert float(base.grad.view(B, G, -1)[:, 1].abs().max()) == 0.0


def test_affine_batched_forms_equal_single_calls(dev):
    """Round 5: the two teacher frames of a step transformed by one launch (views per sample, train.py:832-833), the
    three affine depth maps restored by one launch as the channels of one image (train.py:868-882), read in place out
    of an interleaved decoder output: bit-identical to one call each, forward and adjoint."""
    from mono_vifi_amd import ops
    B, H, W, G = 4, 48, 80, 6
    x, angle, box, ratio = affine_case(25, B, 3, H, W)
    a, b, r = T(angle, dev), T(box, dev), T(ratio, dev)
    x2 = T(np.random.default_rng(26).random((B, 3, H, W)).astype(np.float32), dev)
    both = ops.affine_transform(torch.cat([T(x, dev), x2], 0), a, b)
    assert torch.equal(both[:B], ops.affine_transform(T(x, dev), a, b))
    assert torch.equal(both[B:], ops.affine_transform(x2, a, b))
    rng = np.random.default_rng(27)
    inter = T((0.1 + 20 * rng.random((B * G, 1, H, W))).astype(np.float32), dev, True)
    views = torch.unbind(inter.view(B, G, 1, H, W), 1)
    outs = ops.affine_restore_many(list(views[3:6]), a, b, r)
    wts = [torch.randn(B, 1, H, W, device=dev) for _ in range(3)]
    sum((o * w).sum() for o, w in zip(outs, wts)).backward()
    inter2 = inter.detach().clone().requires_grad_(True)
    views2 = torch.unbind(inter2.view(B, G, 1, H, W), 1)
    outs2 = [ops.affine_restore(v, a, b, r) for v in views2[3:6]]
    sum((o * w).sum() for o, w in zip(outs2, wts)).backward()
    for o, o2 in zip(outs, outs2):
        assert torch.equal(o, o2)
    assert torch.equal(inter.grad, inter2.grad)
    # maps that do not lie one plane apart are gathered first: same result
    outs3 = ops.affine_restore_many([views[5].detach(), views[3].detach(), views[4].detach()], a, b, r)
    assert torch.equal(outs3[0], outs2[2]) and torch.equal(outs3[1], outs2[0])


# ------------------------------------------------------------------ f2: affine glue
AFFINE_SHAPES = [(2, 3, 32, 64), (3, 1, 37, 101), (2, 2, 5, 7), (12, 3, 192, 640), (8, 1, 320, 1024),
                 (2, 6, 20, 70)]       # (six channels: two LDS chunks of the round-6 transform kernel)


@pytest.mark.parametrize("shape", AFFINE_SHAPES)
def test_affine_transform_vs_oracle(dev, shape):
    """Trainer.affine_transform (train.py:888-902): one launch vs the oracle's per-pixel
    restatement; positions are fp32 values up to W, so values agree to ~1e-4 px of the data's
    slope (i.i.d. data: the value range)."""
    from mono_vifi_amd import ops
    x, angle, box, _ = affine_case(21, *shape)
    want = O.affine_transform(x, angle, box)
    got = ops.affine_transform(T(x, dev), T(angle, dev), T(box, dev))
    assert np.max(np.abs(N(got) - want)) <= 1e-5
    with pytest.raises(RuntimeError):     # forward only
        ops.affine_transform(T(x, dev, True), T(angle, dev), T(box, dev))


@pytest.mark.parametrize("shape", AFFINE_SHAPES)
def test_affine_restore_vs_oracle(dev, shape):
    """depth_restore of compute_depth_consistency_loss_affine (train.py:909-916), forward and
    the deterministic two-pass adjoint."""
    from mono_vifi_amd import ops
    x, angle, box, ratio = affine_case(22, *shape, lo=0.1, hi=100.0)
    d = T(x, dev, True)
    out = ops.affine_restore(d, T(angle, dev), T(box, dev), T(ratio, dev))
    want = O.affine_restore(x, angle, box, ratio)
    assert np.max(np.abs(N(out) - want)) <= 1e-5 * 200.0
    g = np.random.default_rng(23).standard_normal(x.shape).astype(np.float32)
    out.backward(T(g, dev))
    gd = O.affine_restore_bwd(g, angle, box, ratio)
    assert rel_err(N(d.grad), gd) <= 1e-5
    # deterministic: a second backward gives the same bits
    d2 = T(x, dev, True)
    ops.affine_restore(d2, T(angle, dev), T(box, dev), T(ratio, dev)).backward(T(g, dev))
    assert np.array_equal(N(d.grad), N(d2.grad))


def test_affine_vs_torch_formulation(dev):
    """The same maths as batched torch ops on the GPU (tests/torch_affine.py: rotate / crop /
    paste as grid_sample calls), values and autograd gradient; plus degenerate boxes."""
    import torch_affine as ta
    from mono_vifi_amd import ops
    x, angle, box, ratio = affine_case(24, 4, 3, 96, 160, lo=0.1, hi=10.0)
    a, b, r = T(angle, dev), T(box, dev), T(ratio, dev)
    got = ops.affine_transform(T(x, dev), a, b)
    want = ta.crop_resize_bilinear(ta.rotate_bilinear(T(x, dev), a), b.long())
    assert float((got - want).abs().max()) <= 2e-4 * 10.0
    d1, d2 = T(x, dev, True), T(x, dev, True)
    o1 = ops.affine_restore(d1, a, b, r)
    o2 = ta.rotate_bilinear(ta.paste_resized(d2, b.long()), -a) * r.view(-1, 1, 1, 1)
    assert float((o1 - o2).abs().max()) <= 2e-4 * 20.0
    w = torch.randn_like(o1)
    (o1 * w).sum().backward()
    (o2 * w).sum().backward()
    assert float((d1.grad - d2.grad).abs().max()) <= 2e-4 * float(d2.grad.abs().max())
    # identity: angle 0, full box, ratio 1 (positions carry ~1e-5 px of rounding; data spans 10)
    z = torch.zeros(4, device=dev)
    full = torch.tensor([[0, 0, 160, 96]] * 4, dtype=torch.int32, device=dev)
    one = torch.ones(4, device=dev)
    assert float((ops.affine_transform(T(x, dev), z, full) - T(x, dev)).abs().max()) <= 1e-5 * 10.0
    assert float((ops.affine_restore(T(x, dev), z, full, one) - T(x, dev)).abs().max()) <= 1e-5 * 10.0
    with pytest.raises(RuntimeError):
        ops.affine_restore(T(x, dev), z[:2], full, one)


# ------------------------------------------------------------------ Conv3x3's reflection pad
@pytest.mark.parametrize("shape", [(1, 1, 2, 2), (2, 3, 3, 5), (3, 16, 24, 40), (12, 16, 192, 640), (2, 3, 4, 8),
                                   (1, 5, 5, 12)])
def test_reflect_pad1(dev, shape):
    """layers.Conv3x3 pads by reflection (layers.py:121-138): bit-identical copy forward,
    deterministic gather backward, against ATen on the same device."""
    import torch.nn.functional as F
    from mono_vifi_amd import layers, ops
    x = torch.randn(*shape, device=dev)
    a, b = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    pa, pb = ops.reflect_pad1(a), F.pad(b, (1, 1, 1, 1), mode="reflect")
    assert torch.equal(pa, pb)
    w = torch.randn_like(pa)
    (pa * w).sum().backward()
    (pb * w).sum().backward()
    assert float((a.grad - b.grad).abs().max()) <= 1e-5 * float(b.grad.abs().max())
    conv = layers.Conv3x3(shape[1], 4).to(dev)
    ref = conv.conv(conv.pad(x))
    assert torch.allclose(conv(x), ref, atol=1e-6)



# ------------------------------------------------------------------ in-kernel tie-break noise
@pytest.mark.parametrize("shape,flags,S", [((2, 33, 70), 0, 2), ((1, 64, 200), 2, 2), ((2, 17, 65), 0, 1),
                                           ((4, 192, 640), 0, 2), ((12, 192, 640), 0, 2), ((8, 320, 1024), 0, 2),
                                           ((12, 192, 512), 0, 2)])
def test_inkernel_noise_replayed_through_the_oracle(dev, shape, flags, S):
    """noise=None: the forward+backward kernel draws the tie-break noise of train.py:1023-1024
    itself (counter-based generator keyed by a per-call seed).  The draw is written out and
    replayed through the oracle: argmin / loss / gradients must be those of the oracle on that
    noise; the draw is standard normal, differs between seeds and repeats for a seed."""
    from mono_vifi_amd import ops, synthetic
    B, H, W = shape
    inp = synthetic.unit_inputs(1700 + H * W + flags + S, B, H, W, num_src=S, pose_scale=0.03,
                                with_mask=True, disp_mode="smooth" if H * W > 10000 else "noise")
    T_np = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1))
                     for k in range(S)], 0)
    n_id = 1 if flags & 2 else S
    srcs = [T(inp["src"][k], dev) for k in range(S)]

    def run(seed):
        torch.manual_seed(seed)
        noise_out = torch.zeros((B, n_id, H, W), device=dev)
        disp, Tt = T(inp["disp"], dev, True), T(T_np, dev, True)
        cfgt = (S, flags, 1e-3, 0.1, 100.0, 1e-7, True, False, noise_out)
        loss, am, argmin, _, _ = ops.Unit.apply(disp, T(inp["tgt"], dev), Tt, T(inp["K"], dev),
                                                T(inp["inv_K"], dev), T(inp["mask_rec"], dev), None, cfgt, *srcs)
        loss.backward()
        return float(loss.detach()), N(argmin), N(disp.grad), N(Tt.grad), N(noise_out)

    l1, a1, gd1, gT1, nz1 = run(5)
    l1b, a1b, gd1b, _, nz1b = run(5)
    _, _, _, _, nz2 = run(6)
    assert np.array_equal(nz1, nz1b) and l1 == l1b and np.array_equal(a1, a1b) and np.array_equal(gd1, gd1b)
    assert not np.array_equal(nz1, nz2)
    assert np.all(np.isfinite(nz1))
    if nz1.size >= 4000:
        assert abs(float(nz1.mean())) < 5.0 / np.sqrt(nz1.size) and abs(float(nz1.std()) - 1.0) < 0.05
        if n_id == 2:    # the two candidates of a pixel are uncorrelated
            assert abs(float(np.corrcoef(nz1[:, 0].ravel(), nz1[:, 1].ravel())[0, 1])) < 0.05
    ref = O.unit(inp["disp"], inp["tgt"], inp["src"], T_np, inp["K"], inp["inv_K"],
                 np.ascontiguousarray(nz1), inp["mask_rec"], flags, want_grads=True)
    am = a1.astype(np.int32)
    am[am == 255] = -1
    assert np.array_equal(am, ref["idx"])
    assert abs(l1 - ref["loss"]) <= 1e-5 * abs(ref["loss"])
    assert_grad_close(gd1, ref["grad_disp"], TOL, "grad_disp vs oracle (replayed noise)")
    assert rel_err(gT1, ref["grad_T"]) <= TOL


# ------------------------------------------------------------------ f1: fused fusion-module levels (G9)
def _fusion_case(g):
    L = len(g["chans"])
    feats = [[g[f"feat_{tag}_{i}"] for i in range(L)] for tag in ("n1", "0", "p1")]
    return L, feats, [g["flow_n1"], g["flow_p1"]], g["mask"], str(g["backbone"]) == "LiteMono"


@pytest.fixture
def fusion_bwd(request):
    """Backward route of the fused fusion levels: "gather" (deterministic inverse-tap-list gather,
    the default) or "scatter" (float atomics)."""
    from mono_vifi_amd import ops
    old = ops.FUSION_BWD_GATHER, ops.FUSION_BWD_ANCHOR
    ops.FUSION_BWD_GATHER = request.param != "scatter"
    ops.FUSION_BWD_ANCHOR = request.param == "gather"       # "cells": the round-4 per-level cell lists
    yield "gather" if request.param == "cells" else request.param
    ops.FUSION_BWD_GATHER, ops.FUSION_BWD_ANCHOR = old


BOTH_FUSION_BWD = pytest.mark.parametrize("fusion_bwd", ["gather", "cells", "scatter"], indirect=True)


@BOTH_FUSION_BWD
@pytest.mark.parametrize("case", ["resnet", "litemono", "dhrnet"])
def test_fusion_levels_vs_golden(dev, case, fusion_bwd):
    """mvf_fusion_prep + mvf_fusion_level_fwd/bwd against what the reference's FusionModule
    methods produced (G9): the tensor entering each 1x1 convolution and the gradients of the
    three feature pyramids.  Embedding bands multiply their argument by up to 2^9, so they get
    5e-4 (see tests/test_oracle_golden.py::test_fusion_oracle_vs_golden); the rest 1e-5."""
    from mono_vifi_amd import ops
    g = load_golden("g9_fusion_" + case)
    L, feats, flows, mask, lite = _fusion_case(g)
    tf = [[T(f, dev, True) for f in lvl] for lvl in feats]
    sizes = [tuple(f.shape[-2:]) for f in feats[1]]
    preps = ops.fusion_prep(T(flows[0], dev), T(flows[1], dev), T(mask, dev), sizes, lite)
    outs = [ops.fusion_level(tf[1][i], tf[0][i], tf[2][i], preps[i], lists=(preps.lists, i)) for i in range(L)]
    sum((o * T(g[f"weight_{i}"], dev)).sum() for i, o in enumerate(outs)).backward()
    en = O.embedding_flows(flows[0], L, lite)
    for i in range(L):
        Cc = int(g["chans"][i])
        want, got = g[f"out_{i}"], N(outs[i])
        assert got.shape == want.shape
        assert np.max(np.abs(N(preps[i])[:, 0:2] - en[i])) <= 1e-6                 # cascaded embedding flow
        assert np.array_equal(got[:, :Cc + 42], want[:, :Cc + 42])                 # feat_0 and emb(0): exact
        assert np.max(np.abs(got[:, Cc + 42:2 * Cc + 42] - want[:, Cc + 42:2 * Cc + 42])) <= 1e-5
        assert np.max(np.abs(got - want)) <= 5e-4
        assert rel_err(N(tf[0][i].grad), g[f"grad_n1_{i}"]) <= 1e-5
        assert rel_err(N(tf[1][i].grad), g[f"grad_0_{i}"]) <= 1e-6
        assert rel_err(N(tf[2][i].grad), g[f"grad_p1_{i}"]) <= 1e-5


@BOTH_FUSION_BWD
@pytest.mark.parametrize("pyr", ["resnet18_640x192", "dhrnet_512x192", "litemono_1024x320"])
def test_fusion_levels_full_pyramids_vs_oracle(dev, pyr, fusion_bwd):
    """The pyramids of BASELINE.json's configs (ResNet18 640x192, HRNet18 512x192 Cityscapes,
    Lite-Mono 1024x320) against the oracle, forward and feature gradients."""
    from mono_vifi_amd import ops
    chans, strides, (H, W), lite = {
        "resnet18_640x192": ([64, 64, 128, 256, 512], [2, 4, 8, 16, 32], (192, 640), False),
        "dhrnet_512x192": ([64, 18, 36, 72, 144], [2, 4, 8, 16, 32], (192, 512), False),
        "litemono_1024x320": ([48, 80, 128], [4, 8, 16], (320, 1024), True)}[pyr]
    rng = np.random.default_rng(91)
    B = 2
    feats = [[rng.standard_normal((B, c, H // s, W // s)).astype(np.float32) for c, s in zip(chans, strides)]
             for _ in range(3)]
    from mono_vifi_amd import synthetic
    flows = [synthetic._box3((8 * rng.standard_normal((B, 2, H, W))).astype(np.float32)) for _ in range(2)]
    mask = rng.random((B, 1, H, W)).astype(np.float32)
    want = O.fusion_forward(feats, flows, mask, lite)
    wts = [rng.standard_normal(w.shape).astype(np.float32) for w in want]
    gn, g0, gp = O.fusion_backward(feats, flows, mask, wts)
    tf = [[T(f, dev, True) for f in lvl] for lvl in feats]
    sizes = [tuple(f.shape[-2:]) for f in feats[1]]
    preps = ops.fusion_prep(T(flows[0], dev), T(flows[1], dev), T(mask, dev), sizes, lite)
    outs = [ops.fusion_level(tf[1][i], tf[0][i], tf[2][i], preps[i], lists=(preps.lists, i)) for i in range(len(chans))]
    sum((o * T(wts[i], dev)).sum() for i, o in enumerate(outs)).backward()
    if fusion_bwd == "gather":       # bit-reproducible: a second backward gives the same bits
        tf2 = [[T(f, dev, True) for f in lvl] for lvl in feats]
        outs2 = [ops.fusion_level(tf2[1][i], tf2[0][i], tf2[2][i], preps[i]) for i in range(len(chans))]
        sum((o * T(wts[i], dev)).sum() for i, o in enumerate(outs2)).backward()
        for i in range(len(chans)):
            assert torch.equal(tf[0][i].grad, tf2[0][i].grad) and torch.equal(tf[2][i].grad, tf2[2][i].grad)
    for i, Cc in enumerate(chans):
        got = N(outs[i])
        assert np.max(np.abs(got[:, :2 * Cc + 42] - want[i][:, :2 * Cc + 42])) <= 2e-5
        assert np.max(np.abs(got - want[i])) <= 1e-3          # 2^9-amplified embedding bands
        assert rel_err(N(tf[0][i].grad), gn[i]) <= 1e-5
        assert rel_err(N(tf[1][i].grad), g0[i]) <= 1e-6
        assert rel_err(N(tf[2][i].grad), gp[i]) <= 1e-5


@pytest.mark.parametrize("kind", ["leaves_right", "leaves_corner", "mixed"])
def test_fusion_adjoint_with_flows_that_leave_the_image(dev, kind):
    """a flow field that leaves the image clamps its taps to the border, so a whole row's (or, towards a
    corner, a whole image's) pixels pile on ONE border anchor: lists of hundreds of entries (wave-cooperative rank sort,
    `anc_sort_wave`) up to a whole level (serial fallback beyond 1,024) instead of the usual one or two.  The anchor
    lists' adjoint against the oracle, and bit-reproducible across two backward passes."""
    from mono_vifi_amd import ops, synthetic
    rng = np.random.default_rng(17)
    B, H, W = 2, 64, 160
    chans, strides = [16, 24], [2, 4]
    feats = [[rng.standard_normal((B, c, H // s, W // s)).astype(np.float32) for c, s in zip(chans, strides)] for _ in range(3)]
    base = synthetic._box3((3 * rng.standard_normal((B, 2, H, W))).astype(np.float32))
    flows = []
    for sgn in (1.0, -1.0):
        f = base.copy()
        if kind == "leaves_right":
            f[:, 0] += sgn * 2.0 * W                       # every tap beyond the right / left border: lists of w entries
        elif kind == "leaves_corner":
            f[:, 0] += sgn * 2.0 * W
            f[:, 1] += sgn * 2.0 * H                       # ... and beyond the bottom / top: one list of h * w entries
        else:
            f[:, 0, :, W // 2:] += sgn * 2.0 * W           # half of every row leaves
        flows.append(np.ascontiguousarray(f))
    mask = rng.random((B, 1, H, W)).astype(np.float32)
    want = O.fusion_forward(feats, flows, mask, False)
    wts = [rng.standard_normal(w.shape).astype(np.float32) for w in want]
    gn, g0, gp = O.fusion_backward(feats, flows, mask, wts)
    sizes = [tuple(f.shape[-2:]) for f in feats[1]]
    grads = []
    for _ in range(2):
        tf = [[T(f, dev, True) for f in lvl] for lvl in feats]
        preps = ops.fusion_prep(T(flows[0], dev), T(flows[1], dev), T(mask, dev), sizes, False)
        outs = [ops.fusion_level(tf[1][i], tf[0][i], tf[2][i], preps[i], lists=(preps.lists, i)) for i in range(len(chans))]
        sum((o * T(wts[i], dev)).sum() for i, o in enumerate(outs)).backward()
        grads.append([(tf[0][i].grad.clone(), tf[2][i].grad.clone()) for i in range(len(chans))])
    for i in range(len(chans)):
        assert torch.equal(grads[0][i][0], grads[1][i][0]) and torch.equal(grads[0][i][1], grads[1][i][1])
        # (thousands of terms pile on one cell: the fp32 sums of oracle and kernel run in the same sorted order)
        assert rel_err(N(grads[0][i][0]), gn[i]) <= 2e-5
        assert rel_err(N(grads[0][i][1]), gp[i]) <= 2e-5


def test_fusion_module_fused_equals_op_by_op(dev):
    """FusionModule on the device: the fused levels against the module's own op-by-op form
    (warp kernel + torch interpolate / sin / cos / cat), outputs and parameter / feature grads."""
    from types import SimpleNamespace
    import mono_vifi_amd.networks.fusion_module as fm
    torch.manual_seed(3)
    chans = [64, 64, 128, 256, 512]
    mod = fm.FusionModule(SimpleNamespace(backbone="ResNet18"), chans).to(dev)
    B, H, W = 2, 192, 640
    mk = lambda: [torch.randn(B, c, H >> (i + 1), W >> (i + 1), device=dev, requires_grad=True)  # noqa: E731
                  for i, c in enumerate(chans)]
    fl = [4 * torch.randn(B, 2, H, W, device=dev) for _ in range(2)]
    mask = torch.rand(B, 1, H, W, device=dev)
    res = {}
    feats = [mk(), mk(), mk()]
    for fused in (True, False):
        fm.FUSED_LEVELS = fused
        try:
            for lvl in feats:
                for f in lvl:
                    f.grad = None
            mod.zero_grad()
            outs = mod(feats, fl, mask)
            sum((o * o).sum() for o in outs).backward()
            res[fused] = ([o.detach().clone() for o in outs], [f.grad.clone() for lvl in feats for f in lvl],
                          [p.grad.clone() for p in mod.parameters()])
        finally:
            fm.FUSED_LEVELS = True
    for a, b in zip(res[True][0], res[False][0]):
        assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max())    # conv of 2^9-amplified bands
    for a, b in zip(res[True][1] + res[True][2], res[False][1] + res[False][2]):
        assert float((a - b).norm() / b.norm()) <= 2e-3


# ------------------------------------------------------------------ f4: step glue
@pytest.mark.parametrize("shape", [(2, 5, 3, 2, 3), (3, 16, 7, 6, 20), (2, 4, 0, 5, 9), (12, 32, 64, 48, 160),
                                   (2, 3, 5, 2, 4), (1, 6, 2, 3, 6)])      # smallest planes of the wide adjoint kernels
def test_up2cat_pad_vs_torch(dev, shape):
    """Decoder stage glue (monodepth2.py:84-90 + layers.py:121-138, 225-228): one pass ==
    ReflectionPad2d(1)(cat([upsample_nearest(x), skip])), bit-identical forward; the gather
    adjoint against autograd of the stock ops."""
    import torch.nn.functional as F
    from mono_vifi_amd import ops
    B, C1, C2, h, w = shape
    torch.manual_seed(1)
    x = torch.randn(B, C1, h, w, device=dev)
    skip = torch.randn(B, C2, 2 * h, 2 * w, device=dev) if C2 else None
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    sa = skip.clone().requires_grad_(True) if C2 else None
    sb = skip.clone().requires_grad_(True) if C2 else None
    got = ops.up2cat_pad(xa, sa)
    up = F.interpolate(xb, scale_factor=2, mode="nearest")
    want = F.pad(torch.cat([up, sb], 1) if C2 else up, (1, 1, 1, 1), mode="reflect")
    assert torch.equal(got, want)
    wgt = torch.randn_like(want)
    (got * wgt).sum().backward()
    (want * wgt).sum().backward()
    assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * float(xb.grad.abs().max())
    if C2:
        assert float((sa.grad - sb.grad).abs().max()) <= 1e-5 * float(sb.grad.abs().max())
    xc = x.clone().requires_grad_(True)          # deterministic
    (ops.up2cat_pad(xc, skip) * wgt).sum().backward()
    assert torch.equal(xc.grad, xa.grad)


def test_disp_head_vs_torch(dev):
    """sigmoid + disp_to_depth epilogue (monodepth2.py:93, layers.py:16-25) and its mean partials."""
    from mono_vifi_amd import layers, ops
    torch.manual_seed(2)
    logit = (3 * torch.randn(6, 1, 48, 160, device=dev))
    la, lb = logit.clone().requires_grad_(True), logit.clone().requires_grad_(True)
    disp, depth, part = ops.disp_head(la, 0.1, 100.0)
    d2 = torch.sigmoid(lb)
    _, dep2 = layers.disp_to_depth(d2, 0.1, 100.0)
    assert float((disp - d2).abs().max()) <= 2e-7
    assert float(((depth - dep2) / dep2).abs().max()) <= 2e-6
    assert float((part.sum(1) / (48 * 160) - d2.mean((1, 2, 3))).abs().max()) <= 1e-6
    wa, wb = torch.randn_like(disp), torch.randn_like(disp)
    ((disp * wa).sum() + (depth * wb).sum()).backward()
    ((d2 * wa).sum() + (dep2 * wb).sum()).backward()
    assert float((la.grad - lb.grad).abs().max()) <= 1e-5 * float(lb.grad.abs().max())
    # the partials are exactly what the unit kernel's own pre-pass computes: same loss bits
    from mono_vifi_amd import synthetic
    inp = synthetic.unit_inputs(31, 2, 48, 96, pose_scale=0.02)
    T_np = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1)) for k in range(2)], 0)
    lg = torch.logit(T(inp["disp"], dev).clamp(1e-4, 1 - 1e-4))
    dsp, _, prt = ops.disp_head(lg, 0.1, 100.0)
    res = []
    for mp in (None, prt):
        dd, Tt = dsp.detach().clone().requires_grad_(True), T(T_np, dev, True)
        cfgt = (2, 0, 1e-3, 0.1, 100.0, 1e-7, True, False, None, mp)
        loss = ops.Unit.apply(dd, T(inp["tgt"], dev), Tt, T(inp["K"], dev), T(inp["inv_K"], dev), None,
                              T(inp["noise"], dev), cfgt, T(inp["src"][0], dev), T(inp["src"][1], dev))[0]
        loss.backward()
        res.append((float(loss.detach()), dd.grad.clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])


def test_decoder_fused_glue_equals_stock_ops(dev):
    """DepthDecoder on the device: fused stage glue + disparity-head epilogue against the stock
    op-by-op form (outputs, input-feature and parameter gradients)."""
    import mono_vifi_amd.networks.monodepth2 as md
    torch.manual_seed(5)
    ch = [64, 64, 128, 256, 512]
    dec = md.DepthDecoder(np.array(ch), range(4)).to(dev)
    B, H, W = 2, 64, 96
    feats = [torch.randn(B, c, H >> (i + 1), W >> (i + 1), device=dev, requires_grad=True) for i, c in enumerate(ch)]
    res = {}
    from mono_vifi_amd import layers as L
    for fused in (True, False):
        md.FUSED_GLUE = L.FUSED_EPILOGUE = fused
        try:
            for f in feats:
                f.grad = None
            dec.zero_grad()
            out = dec(feats)
            sum((out[("disp", s)] ** 2).sum() for s in range(4)).backward()
            res[fused] = ([out[("disp", s)].detach().clone() for s in range(4)], [f.grad.clone() for f in feats],
                          [p.grad.clone() for p in dec.parameters()], out.get(("depth", 0)))
        finally:
            md.FUSED_GLUE = L.FUSED_EPILOGUE = True
    for a, b in zip(res[True][0], res[False][0]):
        assert float((a - b).abs().max()) <= 1e-6
    for a, b in zip(res[True][1] + res[True][2], res[False][1] + res[False][2]):
        assert float((a - b).norm() / b.norm()) <= 1e-4
    from mono_vifi_amd import layers
    assert float(((res[True][3] - layers.disp_to_depth(res[True][0][0], 0.1, 100.0)[1]) / res[True][3]).abs().max()) <= 1e-6


@pytest.mark.parametrize("act", ["none", "elu", "relu", "prelu"])
def test_bias_act_vs_torch(dev, act):
    """Convolution epilogue out = act(x + bias[c] (+ res)) (layers.py:106-118, IFRNet.py:128-157)
    against the stock ops in fp64; backward (act'(out) and the bias gradient) against autograd of
    the stock ops.  Shapes: float4-able planes, odd planes, a plane smaller than a block."""
    import torch.nn.functional as F
    from mono_vifi_amd import ops
    g = torch.Generator(device="cpu").manual_seed(9)
    for shape, with_res, one_slope in (((3, 5, 12, 20), False, False), ((2, 7, 9, 13), True, True),
                                       ((4, 16, 96, 160), True, False), ((2, 3, 1, 1), False, False),
                                       ((2, 3, 192, 640), True, False)):      # planes of several chunks (plane kernels)
        Nn, C = shape[0], shape[1]
        x = (2 * torch.randn(shape, generator=g)).to(dev)
        b = torch.randn(C, generator=g).to(dev)
        r = torch.randn(shape, generator=g).to(dev) if with_res else None
        sl = (0.25 * torch.rand(1 if one_slope else C, generator=g) + 0.05).to(dev) if act == "prelu" else None

        def stock(x_, b_, r_):
            v = x_ + b_.view(1, C, 1, 1)
            if r_ is not None:
                v = v + r_
            if act == "elu":
                return F.elu(v)
            if act == "relu":
                return F.relu(v)
            if act == "prelu":
                return F.prelu(v, sl.to(v.dtype))
            return v
        want = stock(x.double(), b.double(), None if r is None else r.double())
        got = ops.bias_act(x, b, act, sl, r)
        assert float((got.double() - want).abs().max()) <= 2e-6, (act, shape)
        # in place on x
        x2 = x.clone()
        got2 = ops.bias_act(x2, b, act, sl, r, inplace=True)
        assert got2.data_ptr() == x2.data_ptr() and torch.equal(got2, got)
        if act == "prelu":
            xr = x.clone().requires_grad_(True)
            with pytest.raises(NotImplementedError):
                ops.bias_act(xr, b, act, sl, r).sum().backward()
            continue
        xa, ba = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
        xb, bb = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
        ra = None if r is None else r.clone().requires_grad_(True)
        rb = None if r is None else r.clone().requires_grad_(True)
        w = torch.randn(shape, generator=g).to(dev)
        (ops.bias_act(xa, ba, act, None, ra) * w).sum().backward()
        (stock(xb, bb, rb) * w).sum().backward()
        assert float((xa.grad - xb.grad).abs().max()) <= 1e-6 * max(1.0, float(xb.grad.abs().max())), (act, shape)
        if act in ("elu", "relu"):
            # no bias: the activation's adjoint alone (no partial sums) through the same kernels
            xc_, xd_ = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            (ops.bias_act(xc_, None, act) * w).sum().backward()
            ((F.elu(xd_) if act == "elu" else F.relu(xd_)) * w).sum().backward()
            assert float((xc_.grad - xd_.grad).abs().max()) <= 1e-6 * max(1.0, float(xd_.grad.abs().max())), (act, shape)
        assert float((ba.grad - bb.grad).abs().max()) <= 2e-5 * max(1.0, float(bb.grad.abs().max())), (act, shape)
        if r is not None:
            assert torch.equal(ra.grad, xa.grad) and float((ra.grad - rb.grad).abs().max()) <= 1e-6 * max(1.0, float(rb.grad.abs().max()))
        # deterministic bias gradient
        xc, bc = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
        (ops.bias_act(xc, bc, act, None, None if r is None else r) * w).sum().backward()
        assert torch.equal(bc.grad, ba.grad)


def test_ifrnet_epilogue_equals_stock_ops(dev):
    """The frozen teacher with bias + PReLU (+ residual) in one epilogue pass per convolution
    against the stock op-by-op modules (reference: networks/IFRNet.py:128-157, 373-441)."""
    from mono_vifi_amd import layers as L
    from mono_vifi_amd.networks import ifrnet
    torch.manual_seed(4)
    net = ifrnet.IFRNet("small").to(dev).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.PReLU):
            m.weight.data.uniform_(0.05, 0.4)
    a, b = torch.rand(2, 3, 64, 128, device=dev), torch.rand(2, 3, 64, 128, device=dev)
    embt = torch.full((2, 1, 1, 1), 0.5, device=dev)
    out = {}
    with torch.no_grad():
        for fused in (True, False):
            L.FUSED_EPILOGUE = fused
            try:
                out[fused] = net(a, b, embt)
            finally:
                L.FUSED_EPILOGUE = True
    for u, v in zip(out[True], out[False]):
        assert float((u - v).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max()))
    # with a gradient required the stock modules run (PReLU epilogue is forward-only)
    a.requires_grad_(True)
    net(a, b, embt)[0].sum().backward()
    assert a.grad is not None and bool(torch.isfinite(a.grad).all())


def test_resnet_residual_epilogue_equals_stock_ops(dev):
    """relu(out + identity) of the residual blocks as one epilogue pass (in place on the batch-norm
    output) against add + relu: features, input and parameter gradients, grouped and plain."""
    from mono_vifi_amd import layers as L
    from mono_vifi_amd.networks import grouped, monodepth2
    torch.manual_seed(6)
    enc = grouped.convert_grouped_batchnorm(monodepth2.DepthEncoder(18, False)).to(dev).train()
    x = torch.rand(4, 3, 64, 96, device=dev)
    for G in (1, 2):
        res = {}
        state = {k: v.clone() for k, v in enc.state_dict().items()}
        for fused in (True, False):
            L.FUSED_EPILOGUE = fused
            try:
                enc.load_state_dict(state)
                enc.zero_grad()
                xi = x.clone().requires_grad_(True)
                with grouped.grouped(enc, G):
                    feats = enc(xi)
                sum((f ** 2).mean() for f in feats).backward()
                res[fused] = ([f.detach().clone() for f in feats], xi.grad.clone(),
                              torch.cat([p.grad.flatten() for p in enc.parameters()]))
            finally:
                L.FUSED_EPILOGUE = True
        for a, b in zip(res[True][0], res[False][0]):
            assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))
        assert float((res[True][1] - res[False][1]).norm() / res[False][1].norm()) <= 1e-4
        assert float((res[True][2] - res[False][2]).norm() / res[False][2].norm()) <= 1e-4


def test_bn_plan_shared_layer_and_stale_graph(dev):
    """The layer plan of a grouped call (one tiling launch when the call begins, one fold when it ends) against the
    per-layer path: (a) a BatchNorm layer that runs TWICE inside one grouped() call -- a block shared by
    two branches -- gives the sequential result (outputs, gradients, running statistics) instead of raising, and
    stays out of the plan afterwards; (b) a graph of an earlier grouped() call that still holds the prepared buffer
    fails loudly in backward once a later call has refilled it."""
    import torch.nn as nn
    from mono_vifi_amd.networks import grouped

    class Shared(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1, self.c2 = nn.Conv2d(3, 8, 3, padding=1), nn.Conv2d(8, 8, 3, padding=1)
            self.bn_a, self.bn_s = nn.BatchNorm2d(8), nn.BatchNorm2d(8)

        def forward(self, x):
            y = torch.relu(self.bn_a(self.c1(x)))
            u = self.bn_s(self.c2(y))                 # the shared layer, first run
            return self.bn_s(self.c2(torch.relu(u)))  # ... and second
    torch.manual_seed(4)
    G = 3
    x = torch.rand(2 * G, 3, 16, 24, device=dev)
    res = {}
    for plan in (True, False):
        torch.manual_seed(5)
        net = grouped.convert_grouped_batchnorm(Shared()).to(dev).train()
        grouped._BN_PLAN = plan
        try:
            for _ in range(2):                        # second step: the shared layer is out of the plan, the other in it
                net.zero_grad()
                with grouped.grouped(net, G):
                    out = net(x)
                (out ** 2).mean().backward()
        finally:
            grouped._BN_PLAN = True
        res[plan] = (out.detach().clone(), torch.cat([p.grad.flatten() for p in net.parameters()]),
                     torch.cat([b.flatten().float() for b in net.buffers()]))
        if plan:
            assert net.bn_s._plan_off and not net.bn_a._plan_off
    for a, b in zip(res[True], res[False]):
        assert float((a - b).abs().max()) <= 1e-5 * max(1.0, float(b.abs().max()))
    # (b) stale graph
    net = grouped.convert_grouped_batchnorm(nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8))).to(dev).train()
    with grouped.grouped(net, G):
        first = net(x)
    with grouped.grouped(net, G):
        net(x)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        first.sum().backward()


def test_resize_bilinear_vs_oracle(dev):
    """mvf_resize_bilinear_fwd/bwd (HRNet fuse layers: align_corners=True up to 8x; Lite-Mono
    decoder: scale_factor=2) against the oracle (itself pinned to ATen's CPU kernels), forward
    and the gather adjoint; the adjoint is bit-reproducible; drop-in for F.interpolate on the device."""
    import torch.nn.functional as F
    from mono_vifi_amd import ops
    rng = np.random.default_rng(13)
    cases = [(2, 18, 6, 20, 48, 160, None, True), (2, 36, 12, 40, 24, 80, None, True),
             (2, 18, 24, 80, 48, 160, None, True), (3, 40, 24, 80, 48, 160, 2.0, False),
             (1, 5, 5, 7, 13, 9, None, True), (1, 5, 5, 7, 13, 9, None, False),
             (2, 3, 16, 16, 8, 8, 0.5, False), (1, 2, 3, 4, 1, 1, None, True),
             (1, 7, 9, 11, 9, 11, None, True), (2, 16, 80, 256, 160, 512, 2.0, False)]
    for (B, C, ih, iw, oh, ow, sf, ac) in cases:
        x = rng.standard_normal((B, C, ih, iw)).astype(np.float32)
        w = rng.standard_normal((B, C, oh, ow)).astype(np.float32)
        xt = T(x, dev, True)
        out = ops.resize_bilinear(xt, size=None if sf is not None else (oh, ow), scale_factor=sf, align_corners=ac)
        assert tuple(out.shape) == (B, C, oh, ow)
        ref = O.resize_bilinear(x, oh, ow, sf, ac)
        assert np.abs(N(out) - ref).max() <= 2e-6, (ih, iw, oh, ow, sf, ac)
        (out * T(w, dev)).sum().backward()
        gref = O.resize_bilinear_bwd(w, ih, iw, sf, ac)
        assert rel_err(N(xt.grad), gref) <= 1e-5, (ih, iw, oh, ow, sf, ac)
        x2 = T(x, dev, True)
        (ops.resize_bilinear(x2, size=None if sf is not None else (oh, ow), scale_factor=sf, align_corners=ac)
         * T(w, dev)).sum().backward()
        assert torch.equal(x2.grad, xt.grad)
        # the one-pass gather (workspace-free form) and the separable two-pass adjoint agree
        os.environ["MVF_RESIZE_ONEPASS"] = "1"
        try:
            x4 = T(x, dev, True)
            (ops.resize_bilinear(x4, size=None if sf is not None else (oh, ow), scale_factor=sf, align_corners=ac)
             * T(w, dev)).sum().backward()
        finally:
            del os.environ["MVF_RESIZE_ONEPASS"]
        assert rel_err(N(x4.grad), gref) <= 1e-5 and rel_err(N(x4.grad), N(xt.grad)) <= 2e-6
        # ATen's own device kernel agrees
        x3 = T(x, dev, True)
        y3 = (F.interpolate(x3, scale_factor=sf, mode="bilinear", align_corners=ac) if sf is not None
              else F.interpolate(x3, size=(oh, ow), mode="bilinear", align_corners=ac))
        assert float((y3 - out).abs().max()) <= 2e-6


def test_maxpool3s2_vs_oracle(dev):
    """mvf_maxpool3s2_fwd/bwd (the ResNet trunks' nn.MaxPool2d(3, 2, 1)) against the oracle (pinned
    bit for bit to ATen's CPU kernels): values, the selected window element (ties, NaN, -inf
    windows, odd sizes, 1x1) and the gather adjoint are bit-identical; at the stem's full shape
    against ATen's device kernels."""
    import torch.nn.functional as F
    from mono_vifi_amd import ops
    from test_oracle_golden import _pool_cases
    rng = np.random.default_rng(22)
    for x in _pool_cases(rng):
        P, H, W = x.shape
        xt = T(x[None], dev, True)
        out = ops.maxpool3s2(xt)
        ref, code = O.maxpool3s2(x)
        assert np.array_equal(N(out)[0], ref, equal_nan=True), (P, H, W)
        w = rng.standard_normal(ref.shape).astype(np.float32)
        (out * T(w[None], dev)).sum().backward()
        assert np.array_equal(N(xt.grad)[0], O.maxpool3s2_bwd(w, code, H, W)), (P, H, W)
    # the depth encoder's grouped stem at the BASELINE shape: [96, 64, 96, 320] (post-ReLU values)
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn((24, 64, 96, 320), device=dev, generator=g).clamp_min_(0).requires_grad_(True)
    w = torch.randn((24, 64, 48, 160), device=dev, generator=g)
    out = ops.maxpool3s2(x)
    (out * w).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    ref = F.max_pool2d(x2, 3, 2, 1)
    (ref * w).sum().backward()
    assert torch.equal(out, ref)
    assert float((x.grad - x2.grad).abs().max()) <= 1e-6      # same elements selected; sums of <= 4 terms


def test_dhrnet_device_glue_equals_stock_ops(dev):
    """HRNet18 encoder + DHRNet decoder with the device glue (bilinear resize kernel, convolution
    epilogues) against the stock op-by-op form: disparities and parameter gradients."""
    from mono_vifi_amd import layers as L
    from mono_vifi_amd.networks import dhrnet
    torch.manual_seed(8)
    enc = dhrnet.DepthEncoder().to(dev).train()
    dec = dhrnet.DepthDecoder(enc.num_ch_enc).to(dev).train()
    x = torch.rand(4, 3, 128, 192, device=dev)
    params = list(enc.parameters()) + list(dec.parameters())
    res = {}
    state = ({k: v.clone() for k, v in enc.state_dict().items()}, {k: v.clone() for k, v in dec.state_dict().items()})
    for fused in (True, False, None):          # None: the stock form again (its own repeatability)
        L.FUSED_EPILOGUE = bool(fused)
        try:
            enc.load_state_dict(state[0]); dec.load_state_dict(state[1])
            for p in params:
                p.grad = None
            out = dec(enc(x))
            disp = out[("disp", 0)]
            (disp ** 2).mean().backward()
            res[fused] = (disp.detach().clone(), torch.cat([p.grad.flatten() for p in params if p.grad is not None]))
        finally:
            L.FUSED_EPILOGUE = True
    assert float((res[True][0] - res[False][0]).abs().max()) <= 2e-5
    # HRNet's backward is not run-to-run reproducible in the stock form (atomic scatter in ATen's
    # bilinear backward, MIOpen weight gradients): the bar is that repeatability
    noise = float((res[None][1] - res[False][1]).norm() / res[False][1].norm())
    dev_ = float((res[True][1] - res[False][1]).norm() / res[False][1].norm())
    assert dev_ <= 1e-4 + 3.0 * noise, (dev_, noise)


def test_upsample_nearest_vs_torch(dev):
    """mvf_upsample_nearest_fwd/bwd (layers.py:225-228) against F.interpolate(mode="nearest") and
    its autograd: exact forward, block-sum adjoint."""
    import torch.nn.functional as F
    from mono_vifi_amd import layers, ops
    g = torch.Generator(device="cpu").manual_seed(17)
    for (B, C, h, w, f) in ((2, 18, 24, 80, 2), (1, 5, 3, 7, 4), (2, 36, 12, 40, 8), (1, 1, 1, 1, 2), (2, 7, 9, 5, 1),
                            (3, 5, 7, 6, 2), (1, 3, 4, 5, 2)):
        x = torch.randn(B, C, h, w, generator=g).to(dev)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya = layers.upsample(xa, f)
        yb = F.interpolate(xb, scale_factor=f, mode="nearest")
        assert torch.equal(ya, yb)
        wgt = torch.randn(ya.shape, generator=g).to(dev)
        (ya * wgt).sum().backward()
        (yb * wgt).sum().backward()
        assert float((xa.grad - xb.grad).abs().max()) <= 1e-5 * max(1.0, float(xb.grad.abs().max()))
        # the wide factor-2 kernels and the element-per-lane kernels: same bits
        os.environ["MVF_NEAREST_NARROW"] = "1"
        try:
            xc = x.clone().requires_grad_(True)
            yc = layers.upsample(xc, f)
            (yc * wgt).sum().backward()
        finally:
            del os.environ["MVF_NEAREST_NARROW"]
        assert torch.equal(yc, ya) and torch.equal(xc.grad, xa.grad)


# ------------------------------------------------------------------ f4: on-device augmentation
def test_color_jitter_vs_oracle(dev):
    """mvf_color_jitter (flip + ColorJitter of datasets/mono_dataset.py:214-256 for a batch on the
    device) against the oracle's restatement of torchvision's float algorithm: every adjustment
    order, with / without jitter, with / without flip, three frames per sample sharing a draw."""
    import itertools
    from mono_vifi_amd import augment, ops
    rng = np.random.default_rng(51)
    perms = list(itertools.permutations(range(4)))
    S, F, H, W = len(perms) + 2, 3, 20, 36
    img = rng.random((S * F, 3, H, W)).astype(np.float32)
    img[0, :, :4] = 0.5                      # grey pixels: max == min in the hue conversion
    p = augment.draw_params(rng, S)
    p["aug_order"][:len(perms)] = np.array(perms, np.int32)
    p["aug_apply"][:len(perms)] = 1
    p["aug_apply"][len(perms)] = 0
    raw_w, aug_w = O.color_jitter(img, p["aug_factors"], p["aug_order"], p["aug_apply"], p["aug_flip"], F)
    raw, aug = ops.color_jitter(T(img, dev), T(p["aug_factors"], dev), T(p["aug_order"], dev),
                                T(p["aug_apply"], dev), T(p["aug_flip"], dev), frames=F, want_raw=True)
    assert np.array_equal(N(raw), raw_w)
    err = np.abs(N(aug) - aug_w)
    # the hue step is discontinuous where two channels tie for the maximum: allow a handful of pixels
    assert np.mean(err > 2e-5) <= 1e-4 and float(np.median(err)) <= 1e-6
    assert np.array_equal(N(aug)[len(perms) * F:(len(perms) + 1) * F], raw_w[len(perms) * F:(len(perms) + 1) * F])


def test_augment_on_device_fills_the_batch_contract(dev):
    """augment_on_device turns raw frames + the per-sample draw into the keys process_batch reads
    (SURVEY.md section 3.4): color / color_aug / color_affine / color_affine_aug per frame."""
    from mono_vifi_amd import augment, ops, synthetic
    B, H, W = 3, 64, 96
    b = synthetic.training_batch(9, B, H, W)
    rng = np.random.default_rng(2)
    inputs = {("color", f, 0): T(b[("color", f, 0)], dev) for f in (-1, 0, 1)}
    inputs.update({k: T(v, dev) for k, v in augment.draw_params(rng, B).items()})
    inputs["angle"], inputs["box"] = T(b["angle"], dev), T(b["box"].astype(np.int32), dev)
    src = {f: inputs[("color", f, 0)].clone() for f in (-1, 0, 1)}
    out = augment.augment_on_device(inputs, use_affine=True)
    flip = N(inputs["aug_flip"]).astype(bool)
    for f in (-1, 0, 1):
        want = torch.where(T(flip, dev).view(B, 1, 1, 1), src[f].flip(-1), src[f])
        assert torch.equal(out[("color", f, 0)], want)
        for key in ("color_aug", "color_affine", "color_affine_aug"):
            t = out[(key, f, 0)]
            assert tuple(t.shape) == (B, 3, H, W) and bool(torch.isfinite(t).all())
            assert float(t.min()) >= 0.0 and float(t.max()) <= 1.0
        same = N(inputs["aug_apply"]) == 0
        assert torch.equal(out[("color_aug", f, 0)][T(same, dev)], out[("color", f, 0)][T(same, dev)])
        ref_aff = ops.affine_transform(out[("color", f, 0)], inputs["angle"], inputs["box"])
        assert torch.equal(out[("color_affine", f, 0)], ref_aff)


@pytest.mark.parametrize("shape,G,plans", [
    ((3 * 8, 16, 6, 20), 8, ((0, 2, 1, 5, 6, 7), (3, 3, 0), (0, 1, 2), (4, 0, 4))),     # the trainer's plans
    ((2 * 5, 3, 5, 7), 5, ((0, 2, 1), (3, 3, 0), (0, 1, 2), (4, 0, 4))),                 # chunk % 4 != 0: scalar route
    ((4 * 3, 8), 3, ((2,), (0, 0, 0, 2))),                                               # group 1 unread; a group three times
    ((1 * 2, 4, 4), 2, ((1, 0),)),
])
def test_regroup_against_stack_of_views(shape, G, plans):
    """ops.regroup == merge_groups of split_groups views per plan (forward: a copy, bit-exact); adjoint == autograd's
    (sums of at most a handful of terms: 1e-6), zeros for unread groups, None gradients (an output nobody
    differentiated) skipped."""
    from mono_vifi_amd import ops
    from mono_vifi_amd.networks import grouped
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(shape, device="cuda", generator=g)
    xa = x.clone().requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    outs = ops.regroup(xa, G, plans)
    views = grouped.split_groups(xb, G)
    refs = [grouped.merge_groups([views[i] for i in p]) for p in plans]
    assert len(outs) == len(refs)
    for o, r in zip(outs, refs):
        assert o.shape == r.shape and torch.equal(o, r)
    ws = [torch.randn(o.shape, device="cuda", generator=g) for o in outs]
    skip = len(plans) - 1 if len(plans) > 1 else None           # the last output stays out of the loss
    la = sum((o * w).sum() for k, (o, w) in enumerate(zip(outs, ws)) if k != skip)
    lb = sum((r * w).sum() for k, (r, w) in enumerate(zip(refs, ws)) if k != skip)
    la.backward()
    lb.backward()
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-6, atol=1e-6)
    read = {i for k, p in enumerate(plans) if k != skip for i in p}
    for i in range(G):
        if i not in read:
            assert float(xa.grad.view(shape[0] // G, G, -1)[:, i].abs().max()) == 0.0


def test_regroup_rejects_bad_plans():
    from mono_vifi_amd import ops
    x = torch.zeros((6, 4), device="cuda")
    with pytest.raises(RuntimeError):
        ops.regroup(x, 4, ((0,),))            # 6 is not 4 interleaved groups
    with pytest.raises(RuntimeError):
        ops.regroup(x, 3, ((0, 3),))          # group index out of range
    with pytest.raises(RuntimeError):
        ops.regroup(x, 3, ((0,), ()))         # empty plan
    with pytest.raises(RuntimeError):
        ops.regroup(x.cpu(), 3, ((0,),))      # no CPU fallback


def test_regroup_adjoint_reads_channel_slices_in_place():
    """A gradient that is a channel slice of a wider tensor (what the fusion level hands back for its centre
    features) is read where it lies: same result as from a contiguous copy."""
    from mono_vifi_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    G, B, C, h, w = 5, 3, 8, 6, 12
    plans = ((0, 2, 1), (3, 3, 0), (0, 1, 2))
    x = torch.randn((B * G, C, h, w), device="cuda", generator=g)
    res = []
    for sliced in (True, False):
        xa = x.clone().requires_grad_(True)
        outs = ops.regroup(xa, G, plans)
        wide = torch.randn((B * 3, C + 7, h, w), device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
        g1 = wide[:, :C] if sliced else wide[:, :C].contiguous()
        assert g1.is_contiguous() != sliced
        torch.autograd.backward([outs[0], outs[1], outs[2]], [torch.ones_like(outs[0]), g1, 2 * torch.ones_like(outs[2])])
        res.append(xa.grad.clone())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("shape,groups,parts", [((12, 3, 16, 20), 6, 2), ((2, 3, 5, 7), 3, 2), ((4, 3, 8, 8), 8, 1)])
def test_interleave_groups_equals_cat_and_stack(shape, groups, parts):
    """One gather launch == merge_groups([cat(parts, 1) ...]) (a copy: bit-exact); chunk % 4 != 0 takes the scalar route."""
    from mono_vifi_amd import ops
    from mono_vifi_amd.networks import grouped
    g = torch.Generator(device="cuda").manual_seed(2)
    ts = [[torch.randn(shape, device="cuda", generator=g) for _ in range(parts)] for _ in range(groups)]
    ts[0][0] = ts[1][-1]                      # the same image in two groups (a frame shared by two pose pairs)
    ref = grouped.merge_groups([torch.cat(p, 1) for p in ts])
    got = ops.interleave_groups(ts)
    assert got.shape == ref.shape and torch.equal(got, ref)
    with pytest.raises(RuntimeError):
        ops.interleave_groups([[ts[0][0]], [ts[1][0][:, :2]]])          # groups of different widths
    with pytest.raises(RuntimeError):
        ops.interleave_groups([[ts[0][0].clone().requires_grad_(True)]])    # forward-only


@pytest.mark.parametrize("shape", [(3, 5, 16, 24), (2, 3, 9, 13), (1, 2, 8, 10)])
def test_maxpool_tap_adds_the_other_gradient_in_the_pass(shape):
    """maxpool3s2_tap(x) = (maxpool3s2(x), x); its adjoint = pooling adjoint + the tap's gradient in ONE pass,
    bit-identical to autograd's separate accumulation (a two-term sum); wide and narrow kernels; tap unused -> the
    plain adjoint."""
    from mono_vifi_amd import ops
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.randn(shape, device="cuda", generator=g)
    xa, xb, xc = (x.clone().requires_grad_(True) for _ in range(3))
    pa, ta = ops.maxpool3s2_tap(xa)
    pb = ops.maxpool3s2(xb)
    assert torch.equal(pa, pb) and torch.equal(ta, x)
    w1 = torch.randn(pa.shape, device="cuda", generator=g)
    w2 = torch.randn(shape, device="cuda", generator=g)
    ((pa * w1).sum() + (ta * w2).sum()).backward()
    ((pb * w1).sum() + (xb * w2).sum()).backward()
    assert torch.equal(xa.grad, xb.grad)
    pc, _ = ops.maxpool3s2_tap(xc)
    (pc * w1).sum().backward()
    xd = x.clone().requires_grad_(True)
    (ops.maxpool3s2(xd) * w1).sum().backward()
    assert torch.equal(xc.grad, xd.grad)


@pytest.mark.parametrize("shape,n", [((3, 18, 12, 40), 4), ((2, 5, 7, 9), 3), ((1, 4, 4, 4), 2), ((2, 3, 5, 7), 1)])
def test_sum_act_equals_term_at_a_time(shape, n):
    """ops.sum_act == relu(((t0 + t1) + t2) + ...) bit for bit (NaN and -0.0 like ATen's relu), gradient
    g * (out > 0) for every term; the scalar route for sizes that are no multiple of four."""
    from mono_vifi_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    ts = [torch.randn(shape, device="cuda", generator=g) for _ in range(n)]
    ts[0].view(-1)[0] = float("nan")
    ts[0].view(-1)[1] = -0.0
    for t in ts[1:]:
        t.view(-1)[1] = -0.0
    a = [t.clone().requires_grad_(True) for t in ts]
    b = [t.clone().requires_grad_(True) for t in ts]
    out = ops.sum_act(a, "relu")
    acc = None
    for t in b:
        acc = t if acc is None else acc + t
    ref = torch.relu(acc)
    assert torch.equal(torch.nan_to_num(out, nan=7.0), torch.nan_to_num(ref, nan=7.0))
    assert bool(torch.isnan(out.view(-1)[0])) and bool(torch.isnan(ref.view(-1)[0]))
    w = torch.randn(shape, device="cuda", generator=g)
    w.view(-1)[0] = 0.0
    (torch.nan_to_num(out) * w).sum().backward()
    (torch.nan_to_num(ref) * w).sum().backward()
    for x, y in zip(a, b):
        assert torch.equal(torch.nan_to_num(x.grad), torch.nan_to_num(y.grad))
    lin = ops.sum_act([t.clone() for t in ts[1:]] or [ts[0].clone()], "none")
    acc = None
    for t in (ts[1:] or [ts[0]]):
        acc = t if acc is None else acc + t
    assert torch.equal(torch.nan_to_num(lin, nan=7.0), torch.nan_to_num(acc, nan=7.0))


def test_decoder_epilogue_inside_the_pad_kernels(dev):
    """Round 5: a ConvBlock's bias + ELU applied by the pad kernel of its consumer
    (mvf_reflect_pad1_act_*, mvf_up2cat_pad_act_*; networks/monodepth2.py:84-93, layers.py:106-138) against the round-4
    form (epilogue pass, then pad): identical forward bits (same operations), input / weight gradients within the
    rounding of the bias-gradient folds; the ops alone against stock torch ops."""
    import torch.nn.functional as F
    from mono_vifi_amd import ops
    from mono_vifi_amd.networks import monodepth2
    torch.manual_seed(5)
    # the ops alone
    y = torch.randn(3, 6, 12, 20, device=dev, requires_grad=True)
    b = torch.randn(6, device=dev, requires_grad=True)
    skip = torch.randn(3, 5, 24, 40, device=dev, requires_grad=True)
    y2, b2, s2 = (t.detach().clone().requires_grad_(True) for t in (y, b, skip))
    got = ops.reflect_pad1_act(y, b)
    want = F.pad(F.elu(y2 + b2.view(1, -1, 1, 1)), (1, 1, 1, 1), mode="reflect")
    assert torch.equal(got, want)
    w = torch.randn_like(got)
    (got * w).sum().backward()
    (want * w).sum().backward()
    assert float((y.grad - y2.grad).abs().max()) <= 1e-6 * float(y2.grad.abs().max())
    assert float((b.grad - b2.grad).abs().max()) <= 1e-5 * float(b2.grad.abs().max())
    for t in (y, b, y2, b2):
        t.grad = None
    got = ops.up2cat_pad_act(y, b, skip)
    want = F.pad(torch.cat([F.interpolate(F.elu(y2 + b2.view(1, -1, 1, 1)), scale_factor=2, mode="nearest"), s2], 1),
                 (1, 1, 1, 1), mode="reflect")
    assert torch.equal(got, want)
    w = torch.randn_like(got)
    (got * w).sum().backward()
    (want * w).sum().backward()
    assert float((y.grad - y2.grad).abs().max()) <= 1e-6 * float(y2.grad.abs().max())
    assert float((b.grad - b2.grad).abs().max()) <= 1e-5 * float(b2.grad.abs().max())
    assert float((skip.grad - s2.grad).abs().max()) <= 1e-6 * float(s2.grad.abs().max())    # (ATen's pad adjoint adds in another order)
    # the decoder, both forms
    dec = monodepth2.DepthDecoder([64, 64, 128, 256, 512], range(1)).to(dev)
    feats = [torch.randn(2, c, 96 >> i, 160 >> i, device=dev) for i, c in enumerate([64, 64, 128, 256, 512])]
    res = {}
    for flag in (True, False):
        monodepth2.FUSE_EPILOGUE_INTO_PAD = flag
        try:
            fs = [f.clone().requires_grad_(True) for f in feats]
            dec.zero_grad()
            out = dec(fs, 0.1, 100.0)
            (out[("disp", 0)] * torch.linspace(0, 1, 192 * 320, device=dev).view(1, 1, 192, 320)).sum().backward()
            res[flag] = (out[("disp", 0)].detach().clone(), [f.grad.clone() for f in fs],
                         [p.grad.clone() for p in dec.parameters()])
        finally:
            monodepth2.FUSE_EPILOGUE_INTO_PAD = True
    # (the fused ops are bit-exact against the stock ops above; two passes through MIOpen's convolutions are not
    # bit-reproducible on this stack, so the two decoder runs are held to rounding)
    assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-6
    for a, b_ in zip(res[True][1] + res[True][2], res[False][1] + res[False][2]):
        assert float((a - b_).norm()) <= 1e-5 * float(b_.norm())


@pytest.mark.gpu
@pytest.mark.parametrize("narrow", [False, True])
def test_decoder_deferred_epilogue_with_four_scales(dev, narrow, monkeypatch):
    """with num_scales > 1 a ConvBlock output has TWO consumers (the disparity convolution's pad and
    the next level).  When the fused pad + epilogue kernel is refused (MVF_PAD_NARROW, tiny planes) the disparity branch
    must activate a COPY and leave the raw output to the next level -- rebinding it ran bias + ELU twice (wrong features
    in eval, an autograd version error in training).  All four scales, forward and backward, against the un-deferred
    decoder; with the wide kernels allowed and refused, and at a shape whose coarse levels are refused anyway."""
    from mono_vifi_amd.networks import monodepth2
    if narrow:
        monkeypatch.setenv("MVF_PAD_NARROW", "1")
    torch.manual_seed(9)
    for (Hh, Ww) in ((96, 160), (32, 48)):
        dec = monodepth2.DepthDecoder([64, 64, 128, 256, 512], range(4)).to(dev)
        feats = [torch.randn(2, c, max(1, Hh >> i), max(1, Ww >> i), device=dev) for i, c in enumerate([64, 64, 128, 256, 512])]
        res = {}
        for flag in (True, False):
            monodepth2.FUSE_EPILOGUE_INTO_PAD = flag
            try:
                fs = [f.clone().requires_grad_(True) for f in feats]
                dec.zero_grad()
                out = dec(fs, 0.1, 100.0)
                loss = sum((out[("disp", i)] * torch.linspace(0, 1, out[("disp", i)].shape[-1], device=dev)).sum()
                           for i in range(4))
                loss.backward()
                res[flag] = ([out[("disp", i)].detach().clone() for i in range(4)], [f.grad.clone() for f in fs],
                             [p.grad.clone() for p in dec.parameters()])
            finally:
                monodepth2.FUSE_EPILOGUE_INTO_PAD = True
        for a, b_ in zip(res[True][0], res[False][0]):
            assert float((a - b_).abs().max()) <= 2e-6
        for a, b_ in zip(res[True][1] + res[True][2], res[False][1] + res[False][2]):
            assert float((a - b_).norm()) <= 1e-5 * float(b_.norm())
