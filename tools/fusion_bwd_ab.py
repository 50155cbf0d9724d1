"""Same-box A/B of the two adjoints of a FusionModule level (reference: networks/fusion_module.py:80-103, whose own
backward is ATen's atomic grid_sampler_2d_backward): the atomic scatter `mvf_fusion_level_bwd` (memset of the two
gradient planes + float atomics, order of the additions not fixed) against the deterministic inverse-list gather
`mvf_fusion_level_bwd_gather` (count / scan / fill / sort + gather), at the feature pyramids of the three BASELINE
backbones with the merged batch of a training step (3 fusion jobs x batch).  Prints us per level and per step."""
import json
import sys
import os
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mono_vifi_amd import ops  # noqa: E402

PYRAMIDS = {
    "resnet18_640x192_B36": (36, 192, 640, [(64, 2), (64, 4), (128, 8), (256, 16), (512, 32)]),
    "dhrnet_640x192_B36": (36, 192, 640, [(64, 2), (18, 4), (36, 8), (72, 16), (144, 32)]),
    "litemono_1024x320_B24": (24, 320, 1024, [(48, 4), (80, 8), (128, 16)]),
}


def timed(fn, reps=10):
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
    amp = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0       # flow amplitude in pixels (sigma of the coarse field)
    out = {"flow_sigma_px": amp}
    for name, (B, H, W, levels) in PYRAMIDS.items():
        g = torch.Generator(device="cpu").manual_seed(5)
        # flows of a few pixels with smooth structure + noise, like the teacher's
        flow = lambda: (torch.randn(B, 2, H // 16, W // 16, generator=g) * amp)  # noqa: E731
        up = lambda t: torch.nn.functional.interpolate(t, size=(H, W), mode="bilinear")  # noqa: E731
        fl_n1, fl_p1 = up(flow()).to(dev), up(flow()).to(dev)
        mask = torch.sigmoid(up(torch.randn(B, 1, H // 16, W // 16, generator=g))).to(dev)
        sizes = [(H // s, W // s) for _, s in levels]
        preps = ops.fusion_prep(fl_n1, fl_p1, mask, sizes, litemono=name.startswith("litemono"))
        rows, tot = [], {"anchor": 0.0, "gather": 0.0, "atomic": 0.0}
        for (C, s), prep in zip(levels, preps):
            h, w = H // s, W // s
            feats = [torch.randn(B, C, h, w, device=dev, requires_grad=(k > 0)) for k in range(3)]
            gout = torch.randn(B, 2 * (C + ops.EMB_CH), h, w, device=dev)
            res = {}
            grads = {}
            for mode in ("anchor", "gather", "atomic"):
                ops.FUSION_BWD_GATHER = mode != "atomic"
                ops.FUSION_BWD_ANCHOR = mode == "anchor"

                def step():
                    for f in feats[1:]:
                        f.grad = None
                    o = ops.fusion_level(feats[0], feats[1], feats[2], prep)
                    o.backward(gout)
                def fwd_only():
                    with torch.no_grad():
                        ops.fusion_level(feats[0], feats[1], feats[2], prep)
                res[mode] = timed(step) - timed(fwd_only)
                step()
                grads[mode] = [f.grad.clone() for f in feats[1:]]
                tot[mode] += res[mode]
            err = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(grads["atomic"], grads["gather"]))
            err_a = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(grads["anchor"], grads["gather"]))
            alg = 4.0 * B * h * w * C * 3
            rows.append({"C": C, "h": h, "w": w, "anchor_us": round(res["anchor"], 1), "anchor_frac": round(alg / res["anchor"] / 1e-6 / 8e12, 3),
                         "anchor_vs_gather_max_rel_diff": err_a, "gather_us": round(res["gather"], 1), "atomic_us": round(res["atomic"], 1),
                         "algorithmic_MB": round(alg / 1e6, 1),
                         "gather_frac": round(alg / res["gather"] / 1e-6 / 8e12, 3),
                         "atomic_frac": round(alg / res["atomic"] / 1e-6 / 8e12, 3), "max_rel_diff": err})
        out[name] = {"levels": rows, "anchor_us_per_step (lists of one level per call)": round(tot["anchor"], 1), "gather_us_per_step": round(tot["gather"], 1), "atomic_us_per_step": round(tot["atomic"], 1)}
    ops.FUSION_BWD_GATHER = ops.FUSION_BWD_ANCHOR = True
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
