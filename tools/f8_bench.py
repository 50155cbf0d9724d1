"""Same-box timings of the SURVEY 8f-1 kernels at the shapes of a ResNet18 training step (merged batch 3 x 12):
`mvf_fusion_level_fwd`, `mvf_fusion_level_bwd_lists` (lists prebuilt: the adjoint kernel alone) and `mvf_flow_warp_fwd`,
with HIP events around repeated launches; algorithmic bytes as the bench line prices them.  A/B of two builds:
    MVF_HOTPATH_LIB=<other libmvf_hotpath.so> python tools/f8_bench.py [flow sigma in px]
prints one JSON object (us per launch, fraction of the 8 TB/s HBM peak)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_vifi_amd import ops, _native  # noqa: E402

LEVELS = [(64, 2), (64, 4), (128, 8), (256, 16), (512, 32)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    amp = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    B, H, W = 36, 192, 640
    g = torch.Generator(device="cpu").manual_seed(5)
    up = lambda t: torch.nn.functional.interpolate(t, size=(H, W), mode="bilinear")  # noqa: E731
    fl_n1 = up(torch.randn(B, 2, H // 16, W // 16, generator=g) * amp).to(dev)
    fl_p1 = up(torch.randn(B, 2, H // 16, W // 16, generator=g) * amp).to(dev)
    mask = torch.sigmoid(up(torch.randn(B, 1, H // 16, W // 16, generator=g))).to(dev)
    sizes = [(H // s, W // s) for _, s in LEVELS]
    preps = ops.fusion_prep(fl_n1, fl_p1, mask, sizes)
    lists = ops.FusionLists(preps)
    out = {"lib": _native.LIB_PATH, "flow_sigma_px": amp, "fusion": [], "flow_warp": []}
    tot_f = tot_b = 0.0
    for k, ((C, s), prep) in enumerate(zip(LEVELS, preps)):
        h, w = H // s, W // s
        feats = [torch.randn(B, C, h, w, device=dev, requires_grad=(j > 0)) for j in range(3)]
        gout = torch.randn(B, 2 * (C + ops.EMB_CH), h, w, device=dev)

        def fwd():
            with torch.no_grad():
                ops.fusion_level(feats[0], feats[1], feats[2], prep)

        def both():
            for f in feats[1:]:
                f.grad = None
            ops.fusion_level(feats[0], feats[1], feats[2], prep, lists=(lists, k)).backward(gout)
        lists.level(k)
        tf = timed(fwd)
        tb = timed(both) - tf
        bf = 4.0 * B * h * w * (3 * C + 9 + 2 * (C + ops.EMB_CH))
        bb = 4.0 * B * h * w * C * 3
        out["fusion"].append({"C": C, "h": h, "w": w, "fwd_us": round(tf, 1), "fwd_frac": round(bf / tf / 1e-6 / 8e12, 3),
                              "bwd_us": round(tb, 1), "bwd_frac": round(bb / tb / 1e-6 / 8e12, 3)})
        tot_f += tf
        tot_b += tb
    out["fusion_fwd_us_per_step"], out["fusion_bwd_us_per_step"] = round(tot_f, 1), round(tot_b, 1)
    # the teacher's warps: frames and feature maps of IFRNet-L at batch 12 (reference networks/IFRNet.py:7-15)
    tot_w = 0.0
    for (Bw, C, s) in ((12, 3, 1), (12, 32, 2), (12, 48, 4), (12, 72, 8), (12, 96, 16)):
        h, w = H // s, W // s
        img = torch.randn(Bw, C, h, w, device=dev)
        fl = torch.nn.functional.interpolate(fl_n1[:Bw], size=(h, w), mode="bilinear") / s

        def warp():
            with torch.no_grad():
                ops.flow_warp(img, fl)
        t = timed(warp)
        by = 4.0 * Bw * h * w * (2 * C + 2)
        out["flow_warp"].append({"B": Bw, "C": C, "h": h, "w": w, "us": round(t, 1), "frac": round(by / t / 1e-6 / 8e12, 3)})
        tot_w += t
    out["flow_warp_us_sum"] = round(tot_w, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
