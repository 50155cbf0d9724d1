"""Host-side logic on CPU: option parsing of the reference's config format, network
state-dict keys, synthetic batch contract, batched affine helpers."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_options_defaults_and_config_file(tmp_path):
    from mono_vifi_amd import options
    o = options.default_options()
    assert (o.height, o.width, o.batch_size, o.frame_ids) == (192, 640, 12, [0, -1, 1])
    assert (o.learning_rate, o.weight_decay, o.clip_grad, o.decay_step) == (1e-4, 0.01, 5, [15])
    assert (o.min_depth, o.max_depth, o.disparity_smoothness, o.lamda) == (0.1, 100.0, 1e-3, 0.2)
    cfg = tmp_path / "ResNet18_KITTI_MR.txt"       # same format as the reference's configs/
    cfg.write_text("exp_name = ResNet18_KITTI_MR\n\ndataset = kitti\nwidth = 640\nheight = 192\n"
                   "backbone = ResNet18\nfuse_model_type = shared_encoder\nuse_affine = True\n\n"
                   "batch_size = 10\nnum_epochs = 20\nlamda = 0.2\nlr_sche_type = step\n"
                   "learning_rate = 1e-4\ndecay_rate = 0.1\ndecay_step = 15\nresume = True\n\n"
                   "log_frequency = 400\nsave_frequency = 400\n")
    o = options.parse_args(["-c", str(cfg), "--batch_size", "12"])
    assert o.exp_name == "ResNet18_KITTI_MR" and o.use_affine and o.resume
    assert o.batch_size == 12 and o.decay_step == [15] and o.log_frequency == 400
    with pytest.raises(SystemExit):
        bad = tmp_path / "bad.txt"
        bad.write_text("no_such_flag = 1\n")
        options.parse_args(["-c", str(bad)])


def test_network_state_dict_keys_match_reference_layout():
    from types import SimpleNamespace
    from mono_vifi_amd.networks import FusionModule, IFRNet, monodepth2, posenet
    enc = monodepth2.DepthEncoder(18)
    dec = monodepth2.DepthDecoder(enc.num_ch_enc, range(1))
    keys = set(enc.state_dict())
    assert {"encoder.conv1.weight", "encoder.bn1.running_mean", "encoder.layer1.0.conv1.weight",
            "encoder.layer2.0.downsample.0.weight", "encoder.layer4.1.bn2.weight"} <= keys
    assert not any(k.startswith("encoder.fc") for k in keys)
    dk = list(dec.state_dict())
    assert dk[0] == "decoder.0.conv.conv.weight" and "decoder.10.conv.weight" in dk and len(dk) == 22
    assert sum(p.numel() for p in dec.parameters()) == 3150705
    pe = posenet.ResnetEncoder(18, False, 2)
    assert pe.state_dict()["encoder.conv1.weight"].shape == (64, 6, 7, 7)
    pd = posenet.PoseDecoder(pe.num_ch_enc, 1, 2)
    assert list(pd.state_dict())[:2] == ["net.0.weight", "net.0.bias"]
    assert sum(p.numel() for p in pd.parameters()) == 1314572
    fm = FusionModule(SimpleNamespace(backbone="ResNet18"), enc.num_ch_enc)
    assert fm.state_dict()["fusion_conv.0.conv.conv.weight"].shape == (512, 2 * (512 + 42), 1, 1)
    assert sum(p.numel() for p in fm.parameters()) == 791552
    assert sum(p.numel() for p in IFRNet("large").parameters()) == 19699460
    assert sum(p.numel() for p in IFRNet("small").parameters()) == 2803610
    assert "decoder4.convblock.1.conv2.0.weight" in IFRNet("small").state_dict()


def test_networks_forward_shapes(cpu_warp):
    from types import SimpleNamespace
    from mono_vifi_amd.networks import FusionModule, IFRNet, monodepth2, posenet
    x = torch.rand(2, 3, 64, 96)
    enc = monodepth2.DepthEncoder(18)
    f = enc(x)
    assert [t.shape[1] for t in f] == [64, 64, 128, 256, 512]
    d = monodepth2.DepthDecoder(enc.num_ch_enc, range(1))(f)
    assert d[("disp", 0)].shape == (2, 1, 64, 96) and 0 <= float(d[("disp", 0)].detach().min())
    pe = posenet.ResnetEncoder(18, False, 2)
    aa, tr = posenet.PoseDecoder(pe.num_ch_enc, 1, 2)([pe(torch.cat([x, x], 1))])
    assert aa.shape == tr.shape == (2, 2, 1, 3)
    with torch.no_grad():
        img, f0, f1, m = IFRNet("small")(x, x.flip(3), torch.full((2, 1, 1, 1), 0.5))
    assert img.shape == x.shape and f0.shape == (2, 2, 64, 96) and m.shape == (2, 1, 64, 96)
    out = FusionModule(SimpleNamespace(backbone="ResNet18"), enc.num_ch_enc)([f, f, f], [f0, f1], m)
    assert [t.shape for t in out] == [t.shape for t in f]


def test_synthetic_batch_contract():
    from mono_vifi_amd import datasets, synthetic
    b = synthetic.training_batch(3, 2, 64, 96)
    for f in (0, -1, 1):
        for kind in ("color", "color_aug", "color_affine", "color_affine_aug"):
            t = b[(kind, f, 0)]
            assert t.shape == (2, 3, 64, 96) and t.dtype == np.float32 and 0 <= t.min() and t.max() <= 1
    assert b[("K", 0)].shape == (2, 4, 4) and np.allclose(b[("K", 0)][0, 0, 0], 0.58 * 96)
    assert np.allclose(b[("K", 0)][0] @ b[("inv_K", 0)][0], np.eye(4), atol=1e-4)
    assert b["box"].dtype == np.int64 and b["valid_mask_rec"].shape == (2, 1, 64, 96)
    item = datasets.SyntheticTripletDataset(64, 96, 10)[3]
    assert item[("color", 0, 0)].shape == (3, 64, 96) and item["Rc"].shape == (3, 3)


def test_batched_affine_helpers_match_per_sample_loops():
    """tests/torch_affine.py (the batched torch statement the GPU tests compare the affine
    kernels with) against the reference's per-sample loops (train.py:888-922) written with
    torch ops, and against the oracle."""
    import torch.nn.functional as F
    import torch_affine as tr
    torch.manual_seed(0)
    img = torch.rand(3, 2, 32, 48)
    box = torch.tensor([[3, 2, 30, 20], [0, 0, 48, 32], [10, 5, 24, 16]])
    out = tr.crop_resize_bilinear(img, box)
    for b in range(3):
        x0, y0, w, h = box[b].tolist()
        want = F.interpolate(img[b:b + 1, :, y0:y0 + h, x0:x0 + w], [32, 48], mode="bilinear",
                             align_corners=False)
        assert torch.allclose(out[b:b + 1], want, atol=1e-5)
    pasted = tr.paste_resized(img, box)
    for b in range(3):
        x0, y0, w, h = box[b].tolist()
        want = torch.zeros(1, 2, 32, 48)
        want[:, :, y0:y0 + h, x0:x0 + w] = F.interpolate(img[b:b + 1], [h, w], mode="bilinear",
                                                        align_corners=False)
        assert torch.allclose(pasted[b:b + 1], want, atol=1e-5)
    # rotation: 0 deg is the identity, 180 deg flips both axes, +a then -a returns (interior)
    z = tr.rotate_bilinear(img, torch.zeros(3, 1))
    assert torch.allclose(z, img, atol=1e-6)
    r = tr.rotate_bilinear(img, torch.full((3, 1), 180.0))
    assert torch.allclose(r, img.flip(2).flip(3), atol=1e-4)
    yy, xx = torch.meshgrid(torch.arange(32.0), torch.arange(48.0), indexing="ij")
    sm = (0.5 + 0.5 * torch.sin(xx / 9.0) * torch.cos(yy / 7.0)).expand(3, 2, 32, 48).contiguous()
    back = tr.rotate_bilinear(tr.rotate_bilinear(sm, torch.full((3, 1), 4.0)), torch.full((3, 1), -4.0))
    assert (back - sm)[:, :, 10:22, 14:34].abs().max() < 5e-3
    # +90 deg (counter-clockwise) of a square image == torch.rot90 with k=1
    sq = torch.rand(1, 1, 16, 16)
    assert torch.allclose(tr.rotate_bilinear(sq, torch.tensor([[90.0]])), torch.rot90(sq, 1, (2, 3)), atol=1e-4)
    # the composed forms against the oracle
    from oracle import oracle as O
    angle = torch.tensor([[3.0], [-4.5], [0.7]])
    ratio = torch.tensor([[1.6], [1.0], [2.0]])
    want = O.affine_transform(img.numpy(), angle.numpy(), box.numpy())
    got = tr.crop_resize_bilinear(tr.rotate_bilinear(img, angle), box)
    assert np.max(np.abs(got.numpy() - want)) <= 2e-5
    want = O.affine_restore(img.numpy(), angle.numpy(), box.numpy(), ratio.numpy())
    got = tr.rotate_bilinear(tr.paste_resized(img, box), -angle) * ratio.view(-1, 1, 1, 1)
    assert np.max(np.abs(got.numpy() - want)) <= 2e-5


def test_token_linear_backward_under_bf16_autocast():
    """`_TokenLinear` (Lite-Mono's per-image weight gradient) ran its backward outside the
    autocast context with a bf16 gradient against fp32 weight / activation -> dtype error under
    --amp_bf16.  The gate now skips it under autocast, and its backward computes in the weight's
    type whatever reaches it."""
    import torch
    import torch.nn as nn
    from mono_vifi_amd.networks import litemono as lm
    torch.manual_seed(0)
    lin = nn.Linear(8, 12)
    x = torch.randn(2, 40, 8, requires_grad=True)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        y = lm._TokenLinear.apply(x, lin.weight, lin.bias)
    assert y.dtype == torch.bfloat16
    y.float().pow(2).sum().backward()            # backward outside autocast, bf16 upstream gradient
    gx, gw = x.grad.clone(), lin.weight.grad.clone()
    x.grad = lin.weight.grad = lin.bias.grad = None
    with torch.autocast("cpu", dtype=torch.bfloat16):
        y2 = lin(x)
    y2.float().pow(2).sum().backward()
    assert gw.dtype == torch.float32 and gx.dtype == torch.float32
    assert torch.allclose(gx, x.grad, rtol=0.05, atol=0.05)
    assert torch.allclose(gw, lin.weight.grad, rtol=0.05, atol=0.2)


def test_dataset_augmentation_draw_changes_with_the_epoch():
    """the device-augment draw was a function of (seed, index) only: the same flip and
    jitter for an item in every epoch."""
    import torch
    from mono_vifi_amd import datasets
    ds = datasets.SyntheticTripletDataset(32, 64, 8, True, 3, device_augment=True)
    keys = [k for k, v in ds[1].items() if not isinstance(k, tuple) and torch.is_tensor(v)]
    a = ds[1]
    ds.set_epoch(1)
    b = ds[1]
    ds.set_epoch(0)
    c = ds[1]
    draw_keys = [k for k in keys if k not in ("Rc", "angle", "box", "ratio_local", "valid_mask_rec", "valid_mask_cons")]
    assert any(not torch.equal(a[k], b[k]) for k in draw_keys), draw_keys
    assert all(torch.equal(a[k], c[k]) for k in keys)
