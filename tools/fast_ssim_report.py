"""Exact-mode vs the opt-in MVF_FAST_SSIM build (reciprocal-multiply means, v_rcp SSIM quotient):
deviation of one unit at the benchmark shape.  The library is chosen per process with
MVF_HOTPATH_LIB, so run once per build with --dump, then --compare:
    MVF_HOTPATH_LIB=.../var_exact/libmvf_hotpath.so python tools/fast_ssim_report.py --dump a.npz
    MVF_HOTPATH_LIB=.../var_fast/libmvf_hotpath.so  python tools/fast_ssim_report.py --dump b.npz
    python tools/fast_ssim_report.py --compare a.npz b.npz"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "--dump":
    import torch
    from mono_vifi_amd import layers, ops, synthetic
    dev = torch.device("cuda", 0)
    B, H, W = 12, 192, 640
    inp = synthetic.unit_inputs(4242, B, H, W, with_mask=False, disp_mode="smooth")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    aa, tr = t(inp["axisangle"]), t(inp["translation"])
    T = torch.stack([layers.transformation_from_parameters(aa[k], tr[k], invert=(k == 1)) for k in range(2)], 0).detach()
    disp, Tt = t(inp["disp"]).requires_grad_(True), T.clone().requires_grad_(True)
    cfg = (2, 0, 1e-3, 0.1, 100.0, 1e-7, True, True)
    loss, am, argmin, idx, _ = ops.Unit.apply(disp, t(inp["tgt"]), Tt, t(inp["K"]), t(inp["inv_K"]), None,
                                              t(inp["noise"]), cfg, t(inp["src"][0]), t(inp["src"][1]))
    loss.backward()
    np.savez(sys.argv[2], loss=float(loss.detach()), argmin=argmin.cpu().numpy(), idx=idx.cpu().numpy(),
             gd=disp.grad.cpu().numpy(), gT=Tt.grad.cpu().numpy())
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    gd_a, gd_b = a["gd"].astype(np.float64), b["gd"].astype(np.float64)
    rep = {"loss_exact": float(a["loss"]), "loss_fast": float(b["loss"]),
           "loss_rel_diff": abs(float(a["loss"]) - float(b["loss"])) / abs(float(a["loss"])),
           "sampling_indices_equal": bool(np.array_equal(a["idx"], b["idx"])),
           "argmin_flips": int((a["argmin"] != b["argmin"]).sum()), "pixels": int(a["argmin"].size),
           "grad_disp_rel_l2": float(np.linalg.norm(gd_a - gd_b) / np.linalg.norm(gd_a)),
           "grad_disp_max_over_max": float(np.abs(gd_a - gd_b).max() / np.abs(gd_a).max()),
           "grad_T_max_over_max": float(np.abs(a["gT"] - b["gT"]).max() / np.abs(a["gT"]).max())}
    print(json.dumps(rep))
