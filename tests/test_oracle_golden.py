"""Pins the CPU oracle (oracle/mvf_oracle.c) to golden vectors captured from the
reference (tests/golden/make_golden.py).  Bit-exact wherever SURVEY.md section 8a says the
reference's arithmetic is reproducible (indices, depth, cam points, grid, SSIM and
reprojection maps, argmin); tolerance elsewhere, stated per assert."""
import numpy as np
import pytest

from conftest import assert_grad_close, load_golden, rel_err
from oracle import oracle as O

G1 = ["seed0", "seed1", "seed2", "identity", "bigrot"]
FLAGS = {"default": 0, "mask": 0, "no_ssim": O.NO_SSIM, "avg": O.AVG_REPROJ,
         "noauto": O.NO_AUTOMASK, "noauto_mask": O.NO_AUTOMASK}


@pytest.mark.parametrize("case", G1)
def test_geometry_bit_exact(case):
    g = load_golden("g1_geom_" + case)
    _, depth = O.disp_to_depth(g["disp"])
    assert np.array_equal(depth, g["depth"])
    cam = O.backproject(g["depth"], g["inv_K"])
    assert np.array_equal(cam, g["cam_points"])
    B, _, H, W = g["disp"].shape
    for k in range(2):
        P = O.proj_matrix(g["K"], g[f"T{k}"])
        assert np.array_equal(P, g[f"P{k}"])
        pix = O.project(g["cam_points"], g["K"], g[f"T{k}"], H, W)
        assert np.array_equal(pix, g[f"pix{k}"], equal_nan=True)
        out, x0, y0 = O.grid_sample(g["src"][k], g[f"pix{k}"], want_idx=True)
        assert np.array_equal(x0, g[f"x0_{k}"])
        assert np.array_equal(y0, g[f"y0_{k}"])
        # bilinear values: ATen's vectorised kernel may contract; 1e-6 absolute on [0,1] data
        assert np.max(np.abs(out - g[f"warped{k}"])) <= 1e-6
        fused = O.warp_fwd(g["disp"], g["inv_K"], g["K"], g[f"T{k}"], g["src"][k])
        assert np.array_equal(fused["pix"], g[f"pix{k}"], equal_nan=True)
        assert np.array_equal(fused["x0"], g[f"x0_{k}"])
        assert np.array_equal(fused["y0"], g[f"y0_{k}"])
        assert np.max(np.abs(fused["warped"] - g[f"warped{k}"])) <= 1e-6


@pytest.mark.parametrize("case", list(FLAGS))
def test_photometric(case):
    g = load_golden("g2_photo_" + case)
    flags = FLAGS[case]
    assert flags == int(g["flags"][0]) * 1 + int(g["flags"][1]) * 2 + int(g["flags"][2]) * 4
    warped = [g["warped"][0], g["warped"][1]]
    src = [g["src"][0], g["src"][1]]
    assert np.array_equal(O.ssim(warped[0], g["tgt"]), g["ssim0"])
    for k in range(2):
        rp = O.reprojection(warped[k], g["tgt"], no_ssim=bool(flags & O.NO_SSIM))
        assert np.array_equal(rp[:, 0], g["rp"][:, k])
        idl = O.reprojection(src[k], g["tgt"], no_ssim=bool(flags & O.NO_SSIM))
        assert np.array_equal(idl[:, 0], g["idl"][:, k])
    mask = g.get("mask_rec") if case in ("mask", "noauto_mask") else None
    fw = O.losses_base_fwd(g["tgt"], warped, src, g.get("noise"), mask, flags)
    assert np.array_equal(fw["to_opt"].reshape(g["to_opt"].shape), g["to_opt"])
    if "idxs" in g:
        assert np.array_equal(fw["idx"], g["idxs"])
    sm, _ = O.smooth(g["disp"], g["tgt"], normalise=True)
    assert abs(sm - float(g["smooth"])) <= 2e-6 * abs(float(g["smooth"]))
    loss = fw["photo"] + 1e-3 * sm
    assert abs(loss - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    if "auto_mask" in g:
        n_id = 1 if flags & O.AVG_REPROJ else 2
        assert np.array_equal((fw["idx"] > n_id - 1).astype(np.float32)[:, None], g["auto_mask"])


@pytest.mark.parametrize("case", ["default", "mask", "no_ssim", "avg", "noauto"])
def test_gradients(case):
    g = load_golden("g3_grad_" + case)
    flags = int(g["flags"][0]) * 1 + int(g["flags"][1]) * 2 + int(g["flags"][2]) * 4
    mask = g.get("mask_rec") if case == "mask" else None
    out = O.unit(g["disp"], g["tgt"], g["src"], g["T"], g["K"], g["inv_K"], g.get("noise"),
                 mask, flags, want_grads=True)
    assert abs(out["loss"] - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    for k in range(2):
        assert rel_err(out["grad_warped"][k], g["grad_warped"][k]) <= 1e-4
    assert rel_err(out["grad_disp"], g["grad_disp"]) <= 1e-4
    assert rel_err(out["grad_T"], g["grad_T"]) <= 1e-4


def test_ssim_standalone_and_smooth():
    g = load_golden("g6_ssim_smooth")
    assert np.array_equal(O.ssim(g["x"], g["y"]), g["ssim"])
    # catastrophic-cancellation regime: still bit-equal in exact mode
    assert np.array_equal(O.ssim(g["x_near"], g["y"]), g["ssim_near"])
    gx, gy = O.ssim_bwd(g["x"], g["y"], g["weight"])
    assert rel_err(gx, g["grad_x"]) <= 1e-4
    assert rel_err(gy, g["grad_y"]) <= 1e-4
    sm, _ = O.smooth(g["disp"], g["y"], normalise=False)
    assert abs(sm - float(g["smooth"])) <= 2e-6 * abs(float(g["smooth"]))
    gd = O.smooth_bwd(g["disp"], g["y"], 1.0, normalise=False)
    assert rel_err(gd, g["grad_disp"]) <= 1e-5


def test_pose_glue():
    g = load_golden("g5_pose")
    for tag, inv in (("fwd", False), ("inv", True)):
        M = O.pose(g["axisangle"], g["translation"], invert=inv)
        assert np.max(np.abs(M - g["M_" + tag])) <= 2e-6


@pytest.mark.parametrize("cfg", ["C1", "C2", "C4", "C5"])
def test_fullsize_against_reference(cfg):
    """BASELINE.json shapes: inputs regenerated from the seed (pose matrices stored, since
    they come from sin/cos), compared with what the reference produced on them: SHA-256
    of the integer index maps, loss, sampled values, gradient norms."""
    import hashlib
    from mono_vifi_amd import synthetic
    g = load_golden("g4_full_" + cfg)
    B, H, W = (int(v) for v in g["shape"])
    inp = synthetic.unit_inputs(int(g["seed"]), B, H, W, with_mask=True)
    out = O.unit(inp["disp"], inp["tgt"], inp["src"], g["T"], inp["K"], inp["inv_K"],
                 inp["noise"], inp["mask_rec"], 0, want_grads=True)
    n = B * H * W
    sidx = g["sample_idx"]
    for k in range(2):
        assert hashlib.sha256(out["x0"][k].tobytes()).hexdigest() == str(g[f"sha_x0_{k}"])
        assert hashlib.sha256(out["y0"][k].tobytes()).hexdigest() == str(g[f"sha_y0_{k}"])
        w = out["warped"][k].transpose(1, 0, 2, 3).reshape(3, n)[:, sidx]
        assert np.max(np.abs(w - g[f"warped{k}_s"])) <= 1e-6
    assert abs(out["loss"] - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    assert np.array_equal(out["auto_mask"].reshape(n)[sidx], g["auto_mask_s"])
    assert abs(out["auto_mask"].mean() - float(g["auto_mask_mean"])) <= 1e-6
    assert_grad_close(out["grad_disp"].reshape(n)[sidx], g["grad_disp_s"], 1e-4, "grad_disp (sampled)")
    gnorm = np.linalg.norm(out["grad_disp"].astype(np.float64))
    assert abs(gnorm - float(g["grad_disp_norm"])) <= 1e-4 * float(g["grad_disp_norm"])
    # the reference reduces grad_P over H*W pixels in fp32 (BLAS); the oracle in fp64
    assert rel_err(out["grad_T"], g["grad_T"]) <= 5e-3


@pytest.mark.parametrize("name", ["no_ssim", "avg", "noauto"])
def test_fullsize_flag_sets_against_reference(name):
    """The other flag sets at the BASELINE shape C2 (12 x 192 x 640): loss, sampled auto-mask and sampled
    gradients the reference produced under --no_ssim / --avg_reprojection / --disable_automasking
    (tests/golden/make_golden.py::g4_flag_sets; until round 4 these flags were reference-checked at 24x40 only)."""
    from mono_vifi_amd import synthetic
    g = load_golden("g4_flags_C2_" + name)
    B, H, W = (int(v) for v in g["shape"])
    flags = int(g["flags"][0]) * 1 + int(g["flags"][1]) * 2 + int(g["flags"][2]) * 4
    use_mask = bool(int(g["use_mask"]))
    inp = synthetic.unit_inputs(int(g["seed"]), B, H, W, with_mask=use_mask)
    noise = np.ascontiguousarray(inp["noise"][:, :1] if flags & 2 else inp["noise"])
    out = O.unit(inp["disp"], inp["tgt"], inp["src"], g["T"], inp["K"], inp["inv_K"], noise,
                 inp["mask_rec"] if use_mask else None, flags, want_grads=True)
    n = B * H * W
    sidx = g["sample_idx"]
    assert abs(out["loss"] - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    if "auto_mask_s" in g:
        assert np.array_equal(out["auto_mask"].reshape(n)[sidx], g["auto_mask_s"])
        assert abs(out["auto_mask"].mean() - float(g["auto_mask_mean"])) <= 1e-6
    assert_grad_close(out["grad_disp"].reshape(n)[sidx], g["grad_disp_s"], 1e-4, "grad_disp (sampled)")
    gnorm = np.linalg.norm(out["grad_disp"].astype(np.float64))
    assert abs(gnorm - float(g["grad_disp_norm"])) <= 1e-4 * float(g["grad_disp_norm"])
    assert rel_err(out["grad_T"], g["grad_T"]) <= 5e-3      # the reference reduces grad_P in fp32


def _c2_case(name):
    """Inputs of the C2 fixtures (default flags: g4_full_C2; flag sets: g4_flags_C2_*) regenerated from the seed."""
    from mono_vifi_amd import synthetic
    g = load_golden("g4_full_C2" if name == "default" else "g4_flags_C2_" + name)
    B, H, W = (int(v) for v in g["shape"])
    if name == "default":
        flags, use_mask = 0, True
    else:
        flags = int(g["flags"][0]) * 1 + int(g["flags"][1]) * 2 + int(g["flags"][2]) * 4
        use_mask = bool(int(g["use_mask"]))
    inp = synthetic.unit_inputs(int(g["seed"]), B, H, W, with_mask=use_mask)
    noise = np.ascontiguousarray(inp["noise"][:, :1] if flags & 2 else inp["noise"])
    return g, inp, noise, (inp["mask_rec"] if use_mask else None), flags


@pytest.mark.parametrize("name", ["default", "no_ssim", "avg", "noauto"])
def test_double_adjoint_against_reference_float64(name):
    """The oracle's adjoint evaluated in DOUBLE (mvfo_*_bwd_f64: the fp32 forward's decisions, every value in double)
    against the REFERENCE evaluated in float64 on the same inputs (g4_f64_C2_*: 4,096 sampled gradients; samples near
    a pixel whose argmin flips in the float64 run belong to another function and are skipped): 3e-5 of the tensor max
    per sample (measured 2e-6 .. 2.0e-5; the fp32 oracle sits 1.6e-5 .. 1.0e-4 from the same samples).  This pins the
    arbiter the GPU suite holds the kernel's WHOLE gradient tensors to (test_fullsize_gradients_vs_double_adjoint)."""
    g, inp, noise, mask, flags = _c2_case(name)
    d64 = load_golden("g4_f64_C2_" + name)
    out = O.unit(inp["disp"], inp["tgt"], inp["src"], g["T"], inp["K"], inp["inv_K"], noise, mask, flags,
                 want_grads=True, adjoint64=True)
    sidx = d64["sample_idx"]
    gmax = float(d64["grad_disp_max64"])
    same = ~d64["selection_differs_near"].astype(bool)
    got = out["grad_disp64"].reshape(-1)[sidx]
    assert out["grad_disp64"].dtype == np.float64
    assert np.abs(got - d64["grad_disp_s64"])[same].max() <= 3e-5 * gmax
    assert abs(np.abs(out["grad_disp64"]).max() - gmax) <= 1e-4 * gmax
    # the fp32 oracle against its own double evaluation, whole tensor: the spread of fp32 evaluation orders
    e = np.abs(out["grad_disp"].astype(np.float64) - out["grad_disp64"]).max() / np.abs(out["grad_disp64"]).max()
    assert e <= 6e-4, e
    assert rel_err(out["grad_T"], out["grad_T64"]) <= 1e-4


@pytest.mark.parametrize("case", ["a", "b", "big"])
def test_flow_warp(case):
    """f1: IFRNet.warp -- indices bit-exact, values 1e-6, grads 1e-5 vs the reference."""
    g = load_golden("g7_flow_" + case)
    out, x0, y0 = O.flow_warp(g["img"], g["flow"], g["xs"], g["ys"], want_idx=True)
    assert np.array_equal(x0, g["x0"]) and np.array_equal(y0, g["y0"])
    assert np.max(np.abs(out - g["out"])) <= 1e-6
    g_img, g_flow = O.flow_warp_bwd(g["img"], g["flow"], g["xs"], g["ys"], g["weight"])
    assert rel_err(g_img, g["grad_img"]) <= 1e-5
    assert rel_err(g_flow, g["grad_flow"]) <= 1e-5


def test_silog():
    """f2: compute_SI_log_depth_loss incl. an all-masked image, vs the reference."""
    g = load_golden("g8_silog")
    for tag, m in (("nomask", None), ("mask", g["mask"])):
        loss, gp, gt = O.silog(g["pred"], g["target"], m, 0.5, gloss=3.0, want_grads=True)
        assert abs(loss - float(g["loss_" + tag])) <= 2e-6 * abs(float(g["loss_" + tag]))
        assert rel_err(gp, g["grad_pred_" + tag]) <= 1e-5
        assert rel_err(gt, g["grad_target_" + tag]) <= 1e-5


@pytest.mark.parametrize("case", ["default", "mask", "no_ssim", "avg", "noauto"])
def test_unfused_torch_vs_golden(case):
    """oracle/torch_unfused.py -- the un-fused ~130-ATen-op formulation timed as the second CPU
    baseline -- reproduces what the reference produced on G3: warped
    images, loss, gradients."""
    from oracle import torch_unfused as U
    g = load_golden("g3_grad_" + case)
    fl = g["flags"]
    flags = int(fl[0]) * U.NO_SSIM + int(fl[1]) * U.AVG_REPROJ + int(fl[2]) * U.NO_AUTOMASK
    noise = g["noise"] if "noise" in g and not fl[2] else None
    mask = g["mask_rec"] if case == "mask" else None
    out = U.unit(g["disp"], g["tgt"], g["src"], g["T"], g["K"], g["inv_K"], noise, mask, flags)
    for k in range(2):
        assert np.max(np.abs(out["warped"][k] - g["warped"][k])) <= 1e-6
    assert abs(out["loss"] - float(g["loss"])) <= 1e-6 * abs(float(g["loss"]))
    assert rel_err(out["grad_disp"], g["grad_disp"]) <= 1e-5
    assert rel_err(out["grad_T"], g["grad_T"]) <= 1e-5


def test_require_device_rejects_foreign_and_mixed_devices(monkeypatch):
    """kernels are enqueued on the CURRENT device's stream with raw pointers, so
    tensors of another GPU (or of two GPUs) must be rejected, not silently accessed across
    devices (one process per GPU: Trainer / bench.py select the rank's device)."""
    import torch
    from mono_vifi_amd import _native

    class FakeT:
        is_cuda = True

        def __init__(self, i):
            self.device = torch.device("cuda", i)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    _native.require_device(FakeT(0), None, FakeT(0))
    with pytest.raises(RuntimeError, match="different devices"):
        _native.require_device(FakeT(0), FakeT(1))
    with pytest.raises(RuntimeError, match="current device"):
        _native.require_device(FakeT(1), FakeT(1))


@pytest.mark.parametrize("case", ["resnet", "litemono", "dhrnet"])
def test_fusion_oracle_vs_golden(case):
    """f1: the oracle's restatement of FusionModule (what enters the 1x1 convolutions) against
    what the reference's own methods produced (G9).  The sin/cos embedding multiplies its
    argument by up to 2^9, so a 1-ulp difference in a down-sampled flow value moves the highest
    bands by ~1e-4: those channels get 5e-4, everything else 1e-6."""
    g = load_golden("g9_fusion_" + case)
    L = len(g["chans"])
    feats = [[g[f"feat_{tag}_{i}"] for i in range(L)] for tag in ("n1", "0", "p1")]
    flows = [g["flow_n1"], g["flow_p1"]]
    lite = str(g["backbone"]) == "LiteMono"
    en = O.embedding_flows(g["flow_n1"], L, lite)
    outs = O.fusion_forward(feats, flows, g["mask"], lite)
    for i in range(L):
        Cc = int(g["chans"][i])
        assert np.max(np.abs(O.flow_embedding(en[i])[:, :2] - g[f"emb_n1_{i}"][:, :2])) <= 1e-6
        assert np.max(np.abs(O.flow_embedding(en[i]) - g[f"emb_n1_{i}"])) <= 5e-4
        want = g[f"out_{i}"]
        assert outs[i].shape == want.shape
        # f0, emb(0) exact; merged warps of white-noise features: a 1-ulp flow difference moves a sample by ~1e-6 px
        assert np.max(np.abs(outs[i][:, :2 * Cc + 42] - want[:, :2 * Cc + 42])) <= 1e-5
        assert np.max(np.abs(outs[i] - want)) <= 5e-4
    gn, g0, gp = O.fusion_backward(feats, flows, g["mask"], [g[f"weight_{i}"] for i in range(L)])
    for i in range(L):
        assert rel_err(gn[i], g[f"grad_n1_{i}"]) <= 1e-5
        assert rel_err(g0[i], g[f"grad_0_{i}"]) <= 1e-6
        assert rel_err(gp[i], g[f"grad_p1_{i}"]) <= 1e-5


def test_resize_bilinear_oracle_vs_aten():
    """F.interpolate(mode="bilinear") IS the reference's function for the HRNet fuse layers
    (networks/hrnet_encoder.py:275-280, align_corners=True) and the Lite-Mono decoder
    (networks/LiteMono.py:495 via layers.py:225-228, scale_factor=2): the oracle's restatement and
    its adjoint against ATen's CPU kernels."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(3)
    for (ih, iw, oh, ow, sf, ac) in [(6, 20, 48, 160, None, True), (12, 40, 24, 80, None, True),
                                     (24, 80, 48, 160, 2.0, False), (5, 7, 13, 9, None, True),
                                     (5, 7, 13, 9, None, False), (16, 16, 8, 8, 0.5, False),
                                     (3, 4, 1, 1, None, True), (9, 11, 9, 11, None, True)]:
        x = rng.standard_normal((2, 3, ih, iw)).astype(np.float32)
        xt = torch.tensor(x, requires_grad=True)
        y = (F.interpolate(xt, scale_factor=sf, mode="bilinear", align_corners=ac) if sf is not None
             else F.interpolate(xt, size=(oh, ow), mode="bilinear", align_corners=ac))
        assert tuple(y.shape[-2:]) == (oh, ow)
        w = rng.standard_normal(tuple(y.shape)).astype(np.float32)
        (y * torch.tensor(w)).sum().backward()
        assert np.abs(O.resize_bilinear(x, oh, ow, sf, ac) - y.detach().numpy()).max() <= 2e-6
        assert np.abs(O.resize_bilinear_bwd(w, ih, iw, sf, ac) - xt.grad.numpy()).max() <= 2e-5


def _pool_cases(rng):
    cases = []
    for (P, H, W) in [(3, 8, 10), (2, 7, 9), (2, 1, 1), (1, 2, 5), (4, 96, 320 // 4), (1, 5, 2)]:
        x = rng.standard_normal((P, H, W)).astype(np.float32)
        cases.append(x)
        cases.append(np.maximum(x, 0))                        # post-ReLU input: ties at 0 everywhere
        q = np.round(x * 2) / 2                               # coarse values: many exact ties
        cases.append(q.astype(np.float32))
    s = rng.standard_normal((2, 6, 7)).astype(np.float32)
    s[0, 2, 3] = np.nan
    s[1, 0, 0] = np.nan
    s[1, 1, 1] = np.nan
    s[0, 4:, :] = -np.inf
    cases.append(s)
    # W % 4 == 0 (the device's wide kernels: two outputs / a 2 x 4 input block per lane) with NaN, -inf windows,
    # ties, an odd height (wide forward, one-pixel-per-lane backward) and the narrowest wide plane
    for (P, H, W) in [(2, 6, 8), (5, 7, 8), (1, 2, 4), (3, 10, 12)]:
        w = np.round(rng.standard_normal((P, H, W)) * 2).astype(np.float32) / 2
        w[0, min(2, H - 1), 3] = np.nan
        w[0, 0, 0] = np.nan
        w[P - 1, H - 2:, :] = -np.inf
        w[P - 1, :, W - 1] = np.nan if W > 4 else w[P - 1, :, W - 1]
        cases.append(w)
    return cases


def test_maxpool3s2_oracle_vs_aten():
    """nn.MaxPool2d(3, 2, 1) IS the reference's function between conv1 and layer1 of the ResNet
    trunks (networks/monodepth2.py:39, networks/posenet.py:87): the oracle's restatement (values,
    the selected element incl. ties / NaN / -inf windows) and its adjoint against ATen's CPU kernels,
    bit for bit."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(21)
    for x in _pool_cases(rng):
        P, H, W = x.shape
        xt = torch.tensor(x[None], requires_grad=True)
        y, ind = F.max_pool2d(xt, 3, 2, 1, return_indices=True)
        out, code = O.maxpool3s2(x)
        assert out.shape == tuple(y.shape[1:])
        assert np.array_equal(out, y[0].detach().numpy(), equal_nan=True)
        OH, OW = out.shape[1:]
        iy, ix = ind[0].numpy() // W, ind[0].numpy() % W
        want = (iy - (2 * np.arange(OH)[:, None] - 1)) * 3 + (ix - (2 * np.arange(OW)[None, :] - 1))
        assert np.array_equal(code.astype(np.int64), want)
        w = rng.standard_normal(out.shape).astype(np.float32)
        (y * torch.tensor(w[None])).sum().backward()
        assert np.array_equal(O.maxpool3s2_bwd(w, code, H, W), xt.grad[0].numpy())
