"""Network restatements vs the reference's own modules, weights copied across.

Runs only where /root/reference exists (the build container); the reference never travels
to the GPU box, so there these tests skip.  The ResNet trunks cannot be compared: the
reference builds them from torchvision, which is absent ("parity
unpinned" -- tests/test_resnet_vs_hf.py holds them to an independent implementation, transformers'
ResNetModel, instead); everything importable is compared: DepthDecoder, PoseDecoder, FusionModule,
IFRNet (large and small)."""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference not present on this box")


@pytest.fixture(scope="module")
def ref():
    sys.dont_write_bytecode = True
    saved_path, saved_mods = list(sys.path), dict(sys.modules)
    sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision")
    tv.models = types.ModuleType("torchvision.models")
    tv.models.ResNet = torch.nn.Module
    tv.models.resnet18 = tv.models.resnet34 = tv.models.resnet50 = None
    tv.models.resnet101 = tv.models.resnet152 = None
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tv.models
    pkg = types.ModuleType("networks")
    pkg.__path__ = [os.path.join(REF, "networks")]
    sys.modules["networks"] = pkg
    sys.modules.pop("layers", None)
    mods = {n: importlib.import_module("networks." + n)
            for n in ("IFRNet", "fusion_module", "monodepth2", "posenet")}
    yield mods
    sys.path[:] = saved_path
    # remove what this module injected -- the stubs (no __file__) and the reference's own modules --
    # and nothing else: torch sub-modules imported lazily in the meantime must stay (deleting them
    # makes a later re-import register their dispatcher kernels twice)
    for k, m in list(sys.modules.items()):
        if k in saved_mods or k.split(".")[0] in ("torch", "numpy", "transformers"):
            continue
        f = getattr(m, "__file__", None)
        if k.split(".")[0] in ("torchvision", "networks", "layers", "timm", "yacs", "matplotlib", "hrnet_config") \
                or (f and os.path.abspath(f).startswith(REF)):
            del sys.modules[k]


def _copy(dst, src):
    sd, dd = src.state_dict(), dst.state_dict()
    assert list(sd) == list(dd), "state-dict keys / order differ from the reference"
    dst.load_state_dict(sd)


def test_depth_decoder(ref):
    from mono_vifi_amd.networks import monodepth2
    ch = np.array([64, 64, 128, 256, 512])
    torch.manual_seed(0)
    theirs = ref["monodepth2"].DepthDecoder(ch, range(1))
    ours = monodepth2.DepthDecoder(ch, range(1))
    _copy(ours, theirs)
    feats = [torch.randn(2, c, 32 // 2 ** i, 48 // 2 ** i) for i, c in enumerate(ch)]
    a, b = theirs(feats)[("disp", 0)], ours(feats)[("disp", 0)]
    assert torch.allclose(a, b, atol=1e-6)


def test_pose_decoder(ref):
    from mono_vifi_amd.networks import posenet
    ch = np.array([64, 64, 128, 256, 512])
    torch.manual_seed(1)
    theirs = ref["posenet"].PoseDecoder(ch, 1, 2)
    ours = posenet.PoseDecoder(ch, 1, 2)
    _copy(ours, theirs)
    feats = [[torch.randn(2, c, 6, 20) for c in ch]]
    for x, y in zip(theirs(feats), ours(feats)):
        assert torch.allclose(x, y, atol=1e-7)


@pytest.mark.parametrize("scale", ["small", "large"])
def test_ifrnet(ref, scale, cpu_warp):
    from mono_vifi_amd.networks import IFRNet
    torch.manual_seed(2)
    theirs = ref["IFRNet"].IFRNet(scale).eval()
    ours = IFRNet(scale).eval()
    _copy(ours, theirs)
    img0, img1 = torch.rand(1, 3, 64, 128), torch.rand(1, 3, 64, 128)
    embt = torch.full((1, 1, 1, 1), 0.5)
    with torch.no_grad():
        for x, y in zip(theirs(img0, img1, embt), ours(img0, img1, embt)):
            assert torch.allclose(x, y, atol=2e-5), float((x - y).abs().max())
        for x, y in zip(theirs(img0, img1, embt, onlyFlow=True), ours(img0, img1, embt, onlyFlow=True)):
            assert torch.allclose(x, y, atol=2e-5)
    f = torch.rand(1, 5, 16, 24)
    fl = 3 * torch.randn(1, 2, 16, 24)
    from mono_vifi_amd.networks.ifrnet import warp
    assert torch.allclose(ref["IFRNet"].warp(f, fl), warp(f, fl), atol=1e-6)


def test_fusion_module(ref, cpu_warp):
    from types import SimpleNamespace
    from mono_vifi_amd.networks import FusionModule
    ch = np.array([64, 64, 128, 256, 512])
    args = SimpleNamespace(backbone="ResNet18")
    torch.manual_seed(3)
    theirs = ref["fusion_module"].FusionModule(args, ch)
    ours = FusionModule(args, ch)
    _copy(ours, theirs)
    mk = lambda: [torch.randn(2, c, 32 // 2 ** i, 48 // 2 ** i) for i, c in enumerate(ch)]  # noqa: E731
    feats = [mk(), mk(), mk()]
    flows = [2 * torch.randn(2, 2, 64, 96), 2 * torch.randn(2, 2, 64, 96)]
    mask = torch.rand(2, 1, 64, 96)
    a = theirs([list(f) for f in feats], [fl.clone() for fl in flows], mask.clone())
    b = ours(feats, flows, mask)
    for x, y in zip(a, b):
        assert torch.allclose(x, y, atol=1e-5), float((x - y).abs().max())


def _import_ref_dhrnet():
    """The reference's DHRNet needs yacs (absent): a dict-backed CfgNode stand-in is enough
    for its read-only config tables.  matplotlib is stubbed if missing."""
    yacs = types.ModuleType("yacs")
    cfgmod = types.ModuleType("yacs.config")

    class CfgNode(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v
    cfgmod.CfgNode = CfgNode
    yacs.config = cfgmod
    sys.modules.setdefault("yacs", yacs)
    sys.modules.setdefault("yacs.config", cfgmod)
    try:
        import matplotlib.pyplot  # noqa: F401
    except Exception:
        mpl = types.ModuleType("matplotlib")
        mpl.pyplot = types.ModuleType("matplotlib.pyplot")
        sys.modules["matplotlib"] = mpl
        sys.modules["matplotlib.pyplot"] = mpl.pyplot
    return importlib.import_module("networks.DHRNet")


def test_dhrnet(ref):
    from mono_vifi_amd.networks import dhrnet
    theirs_mod = _import_ref_dhrnet()
    torch.manual_seed(4)
    their_enc = theirs_mod.DepthEncoder(18, False).eval()
    their_dec = theirs_mod.DepthDecoder(their_enc.num_ch_enc, range(1)).eval()
    enc, dec = dhrnet.DepthEncoder(18).eval(), dhrnet.DepthDecoder(their_enc.num_ch_enc, range(1)).eval()
    _copy(enc, their_enc)
    _copy(dec, their_dec)
    x = torch.rand(1, 3, 64, 96)
    with torch.no_grad():
        fa, fb = their_enc(x), enc(x)
        for a, b in zip(fa, fb):
            assert torch.allclose(a, b, atol=1e-5), float((a - b).abs().max())
        assert torch.allclose(their_dec(fa)[("disp", 0)], dec(fb)[("disp", 0)], atol=1e-6)
    assert sum(p.numel() for p in enc.parameters()) == 9562260
    assert sum(p.numel() for p in dec.parameters()) == 416589


def _import_ref_litemono():
    """The reference's LiteMono imports DropPath / trunc_normal_ from timm (absent): stub them
    with the textbook definitions so that the reference module itself can be imported."""
    timm = types.ModuleType("timm")
    timm.models = types.ModuleType("timm.models")
    layers_mod = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            return x      # compared in eval mode only
    layers_mod.DropPath = DropPath
    layers_mod.trunc_normal_ = torch.nn.init.trunc_normal_
    timm.models.layers = layers_mod
    for name, mod in (("timm", timm), ("timm.models", timm.models), ("timm.models.layers", layers_mod)):
        sys.modules.setdefault(name, mod)
    return importlib.import_module("networks.LiteMono")


@pytest.mark.parametrize("hw", [(192, 640), (320, 1024)])
def test_litemono(ref, hw):
    from mono_vifi_amd.networks import litemono
    theirs_mod = _import_ref_litemono()
    H, W = hw
    torch.manual_seed(5)
    their_enc = theirs_mod.DepthEncoder(model="lite-mono", drop_path_rate=0.2, width=W, height=H).eval()
    their_dec = theirs_mod.DepthDecoder(their_enc.num_ch_enc, range(1)).eval()
    enc = litemono.DepthEncoder(model="lite-mono", drop_path_rate=0.2, width=W, height=H).eval()
    dec = litemono.DepthDecoder(their_enc.num_ch_enc, range(1)).eval()
    _copy(enc, their_enc)
    _copy(dec, their_dec)
    x = torch.rand(1, 3, 64, 96)
    with torch.no_grad():
        fa, fb = their_enc(x), enc(x)
        for a, b in zip(fa, fb):
            assert torch.allclose(a, b, atol=2e-5), float((a - b).abs().max())
        assert torch.allclose(their_dec(fa)[("disp", 0)], dec(fb)[("disp", 0)], atol=1e-6)


def test_resume_from_a_checkpoint_written_in_the_reference_layout(ref, tmp_path):
    """ckpt.pth written the way the reference's Trainer.save_model does (train.py:1108-1136) --
    state dicts of the REFERENCE's own modules, plus what only its files contain: the dead
    ImageNet ``fc`` head under the ResNet encoders, and an optimizer state over its parameter
    list (aliased encoder_mf parameters twice, fc included) -- resumed by this build's Trainer
    (train.py:1138-1159 semantics): every weight this build owns is taken over, foreign keys
    are ignored, epoch / batch / step are restored, and the non-interchangeable optimizer state
    restarts with a warning instead of raising."""
    from types import SimpleNamespace
    from mono_vifi_amd.options import default_options
    from mono_vifi_amd.trainer import Trainer
    opts = default_options(batch_size=2, height=64, width=96, use_affine=True, num_workers=0, synthetic_len=16,
                           log_dir=str(tmp_path), exp_name="foreign", log_frequency=10 ** 9,
                           save_frequency=10 ** 9, resume=True)
    torch.manual_seed(11)
    ch = np.array([64, 64, 128, 256, 512])
    theirs = {"depth": ref["monodepth2"].DepthDecoder(ch, range(opts.num_scales)),
              "depth_mf": ref["monodepth2"].DepthDecoder(ch, range(opts.num_scales)),
              "pose": ref["posenet"].PoseDecoder(ch, 1, 2),
              "fusion_module": ref["fusion_module"].FusionModule(SimpleNamespace(backbone="ResNet18"), ch)}
    # the torchvision-built encoders cannot be instantiated here (torchvision is absent): their
    # state dicts are written with torchvision's key names (what this build's trunks use), random
    # values, plus the fc head every reference checkpoint carries
    probe = Trainer(default_options(batch_size=2, height=64, width=96, use_affine=True, num_workers=0,
                                    synthetic_len=16, log_dir=str(tmp_path), exp_name="probe",
                                    log_frequency=10 ** 9, save_frequency=10 ** 9))
    ckpt, params = {}, []
    for name in ("encoder", "pose_encoder"):
        sd = {k: torch.randn_like(v) if v.is_floating_point() else v.clone()
              for k, v in probe.models[name].state_dict().items()}
        sd["encoder.fc.weight"], sd["encoder.fc.bias"] = torch.randn(1000, 512), torch.randn(1000)
        ckpt[name] = sd
        params += [torch.nn.Parameter(v.clone()) for v in sd.values() if v.is_floating_point()]
    ckpt["encoder_mf"] = ckpt["encoder"]                      # shared_encoder: the same module saved twice
    for name, m in theirs.items():
        ckpt[name] = m.state_dict()
        params += list(m.parameters())
    params += [p for p in params[:20]]                          # the reference appends aliased parameters again
    opt = torch.optim.AdamW(params, lr=1e-4)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [15], 0.1)
    ckpt.update(height=64, width=96, use_stereo=False, epoch=3, step_in_total=1234, batch_idx=17,
                optimizer=opt.state_dict(), lr_scheduler=sched.state_dict())
    os.makedirs(os.path.join(str(tmp_path), "foreign"), exist_ok=True)
    torch.save(ckpt, os.path.join(str(tmp_path), "foreign", "ckpt.pth"))

    t = Trainer(opts)
    assert (t.ep_start, t.batch_start, t.step) == (3, 17, 1234)
    for name, m in theirs.items():
        mine = t.models[name].state_dict()
        assert list(mine) == list(m.state_dict())
        for k, v in m.state_dict().items():
            assert torch.equal(mine[k].cpu(), v), (name, k)
    for name in ("encoder", "pose_encoder"):
        mine = t.models[name].state_dict()
        assert "encoder.fc.weight" not in mine
        for k, v in mine.items():
            assert torch.equal(v.cpu(), ckpt[name][k]), (name, k)
    assert t.models["encoder_mf"] is t.models["encoder"]
    assert len(t.model_optimizer.state_dict()["state"]) == 0      # restarted, not crashed


def test_token_linear_weight_gradient_equals_stock_linear():
    """Lite-Mono's token MLPs form the weight gradient per image (batched GEMM) and sum: same
    function as nn.Linear's single tall-skinny GEMM."""
    import torch
    from mono_vifi_amd.networks import litemono
    torch.manual_seed(0)
    lin = torch.nn.Linear(12, 30)
    x = torch.randn(3, 5, 7, 12, requires_grad=True)
    w = torch.randn(3, 5, 7, 30)
    y = litemono._TokenLinear.apply(x, lin.weight, lin.bias)
    (y * w).sum().backward()
    got = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None
    lin.zero_grad()
    (lin(x) * w).sum().backward()
    for a, b in zip(got, (x.grad, lin.weight.grad, lin.bias.grad)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)


def test_litemono_nchw_mlp_equals_token_form():
    """The CDC block's channel MLP evaluated in NCHW (batched GEMMs over the images) is the same
    function as the reference's permute -> nn.Linear -> permute form: outputs and every gradient."""
    import torch
    from mono_vifi_amd.networks import litemono
    torch.manual_seed(1)
    blk = litemono.DilatedConv(12, 3, dilation=2, drop_path=0.0, layer_scale_init_value=0.5, expan_ratio=6)
    blk.train()
    x = torch.randn(3, 12, 9, 14, requires_grad=True)
    w = torch.randn(3, 12, 9, 14)
    res = {}
    try:
        for flag in (True, False):
            litemono.NCHW_MLP = flag
            x.grad = None
            blk.zero_grad()
            y = blk(x)
            (y * w).sum().backward()
            res[flag] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in blk.parameters()
                                                                 if p.grad is not None]
    finally:
        litemono.NCHW_MLP = True
    assert len(res[True]) == len(res[False]) >= 8
    for a, b in zip(res[True], res[False]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5), float((a - b).abs().max())


def test_samplers_equal_the_reference_classes(monkeypatch):
    """CustomSampler / CustomDistributedSampler (reference datasets/__init__.py:10-88) drive the data
    order and the mid-epoch resume: this build's classes against the reference's own, imported under
    another package name with its dataset sub-modules (cv2 / skimage users) stubbed."""
    import importlib.util
    from mono_vifi_amd import datasets as ours
    subs = {"kitti_dataset": ["KITTIRAWDataset", "KITTIOdomDataset", "KITTIDepthDataset"],
            "make3d_dataset": ["Make3DDataset"], "nyuv2_dataset": ["NYUDataset"],
            "cityscapes_dataset": ["CityscapesDataset"],
            "VFI_dataset": ["KITTI_VFI_Dataset", "Cityscapes_VFI_Dataset"]}
    injected = []
    try:
        for sub, names in subs.items():
            m = types.ModuleType("mvf_refdatasets." + sub)
            for n in names:
                setattr(m, n, object)
            sys.modules[m.__name__] = m
            injected.append(m.__name__)
        spec = importlib.util.spec_from_file_location(
            "mvf_refdatasets", os.path.join(REF, "datasets", "__init__.py"),
            submodule_search_locations=[os.path.join(REF, "datasets")])
        ref = importlib.util.module_from_spec(spec)
        sys.modules["mvf_refdatasets"] = ref
        injected.append("mvf_refdatasets")
        old = sys.dont_write_bytecode
        sys.dont_write_bytecode = True
        try:
            spec.loader.exec_module(ref)
        finally:
            sys.dont_write_bytecode = old
    finally:
        pass

    class D:
        def __len__(self):
            return 103

    try:
        a, b = ref.CustomSampler(D(), seed=7), ours.CustomSampler(D(), seed=7)
        for epoch, start in ((0, 0), (3, 0), (3, 17)):
            for s in (a, b):
                s.set_epoch(epoch)
                s.set_start_iter(start)
            assert list(a) == list(b) and len(a) == len(b)
        import torch.utils.data.distributed as tud
        world = 4
        for rank in range(world):
            monkeypatch.setattr(tud.dist, "is_available", lambda: True)
            monkeypatch.setattr(tud.dist, "get_world_size", lambda *a_, **k_: world)
            monkeypatch.setattr(tud.dist, "get_rank", lambda *a_, **k_: rank)
            theirs = ref.CustomDistributedSampler(D(), seed=7)
            mine = ours.CustomDistributedSampler(D(), seed=7, num_replicas=world, rank=rank)
            for epoch, start in ((0, 0), (2, 0), (2, 9)):
                for s in (theirs, mine):
                    s.set_epoch(epoch)
                    s.set_start_iter(start)
                assert list(theirs) == list(mine), (rank, epoch, start)
            assert len(theirs) == len(mine) == 103 // world
    finally:
        for n in injected:
            sys.modules.pop(n, None)


def _import_ref_options(cfg_path):
    """Execute the reference's options.py (module-level parser + parse_args) for one config file.
    configargparse is absent: a minimal stand-in reads the `key = value` file the way configargparse
    documents it (keys are long option names; `true` for a store_true flag sets it)."""
    import argparse
    import importlib.util

    class _Parser(argparse.ArgumentParser):
        def add_argument(self, *a, **k):
            k.pop("is_config_file", None)
            return super().add_argument(*a, **k)

        def parse_args(self, args=None, namespace=None):
            args = list(sys.argv[1:] if args is None else args)
            cfg = args[args.index("-c") + 1]
            flags = {o: act for act in self._actions for o in act.option_strings}
            extra = []
            for line in open(cfg):
                line = line.split("#")[0].strip()
                if not line:
                    continue
                key, val = [t.strip() for t in line.split("=", 1)]
                act = flags["--" + key]
                if isinstance(act, argparse._StoreTrueAction):
                    if val.lower() in ("true", "yes", "1"):
                        extra.append("--" + key)
                elif act.nargs in ("+", "*"):
                    extra += ["--" + key] + val.strip("[]").replace(",", " ").split()
                else:
                    extra += ["--" + key, val]
            return super().parse_args(extra + args, namespace)

    stub = types.ModuleType("configargparse")
    stub.ArgumentParser = _Parser
    saved_argv, saved_mod = sys.argv, sys.modules.get("configargparse")
    sys.modules["configargparse"] = stub
    sys.argv = ["train.py", "-c", cfg_path]
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        spec = importlib.util.spec_from_file_location("mvf_refoptions", os.path.join(REF, "options.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.opts
    finally:
        sys.dont_write_bytecode = old
        sys.argv = saved_argv
        if saved_mod is None:
            sys.modules.pop("configargparse", None)
        else:
            sys.modules["configargparse"] = saved_mod


def test_options_equal_the_reference_parser_on_its_own_config_files():
    """Every option the reference's options.py defines -- name, type, default -- and every training
    config file it ships (configs/resnet18, dhrnet, litemono) through this build's options.parse_args:
    the same namespace, key for key (paths that default to the reference's own directory aside)."""
    import glob
    from mono_vifi_amd import options
    files = sorted(glob.glob(os.path.join(REF, "configs", "resnet18", "*.txt")) +
                   glob.glob(os.path.join(REF, "configs", "dhrnet", "*.txt")) +
                   glob.glob(os.path.join(REF, "configs", "litemono", "*.txt")))
    assert len(files) >= 9
    for f in files:
        theirs = vars(_import_ref_options(f))
        mine = vars(options.parse_args(["-c", f]))
        skip = {"config", "data_path", "log_dir"}          # default to the reference's checkout / $HOME
        for k, v in theirs.items():
            if k in skip:
                continue
            assert k in mine, (os.path.basename(f), k)
            assert mine[k] == v, (os.path.basename(f), k, mine[k], v)


def test_layers_namespace_covers_the_reference(ref):
    """`from layers import *` must resolve every function / class name the reference's layers.py defines;
    the one function the reference itself never calls (get_smooth_loss_dyn, layers.py:244-258) equals the
    reference's on random inputs."""
    import ast
    from mono_vifi_amd import layers as mine
    tree = ast.parse(open(os.path.join(REF, "layers.py")).read())
    names = {n.name for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
    missing = sorted(n for n in names if not hasattr(mine, n))
    assert not missing, missing
    ref_layers = importlib.import_module("layers")
    g = torch.Generator().manual_seed(0)
    disp, img = torch.rand((2, 1, 12, 20), generator=g), torch.rand((2, 3, 12, 20), generator=g)
    mask = (torch.rand((2, 1, 12, 20), generator=g) > 0.7).float()
    a, b = mine.get_smooth_loss_dyn(disp, img, mask), ref_layers.get_smooth_loss_dyn(disp.clone(), img, mask)
    assert abs(float(a) - float(b)) <= 1e-6 * abs(float(b))
