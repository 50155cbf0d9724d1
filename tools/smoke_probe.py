"""Probe: per-element and L2 gradient error of the smoke() case, separate kernels and FB."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from mono_vifi_amd import ops, synthetic
from oracle import oracle as O
dev = torch.device("cuda:0")
B, H, W = 2, 48, 96
inp = synthetic.unit_inputs(11, B, H, W, pose_scale=0.02, with_mask=True)
T_np = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1)) for k in range(2)], 0)
ref = O.unit(inp["disp"], inp["tgt"], inp["src"], T_np, inp["K"], inp["inv_K"], inp["noise"], inp["mask_rec"], 0, want_grads=True)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for want_idx in (True, False):
    disp = t(inp["disp"]).requires_grad_(True)
    T = t(T_np).requires_grad_(True)
    cfg = (2, 0, 1e-3, 0.1, 100.0, 1e-7, True, want_idx)
    loss = ops.Unit.apply(disp, t(inp["tgt"]), T, t(inp["K"]), t(inp["inv_K"]), t(inp["mask_rec"]), t(inp["noise"]), cfg, t(inp["src"][0]), t(inp["src"][1]))[0]
    loss.backward()
    g = disp.grad.cpu().numpy().astype(np.float64); r = ref["grad_disp"].astype(np.float64)
    d = np.abs(g - r)
    i = np.unravel_index(np.argmax(d), d.shape)
    print("want_idx", want_idx, "max rel", d.max() / np.abs(r).max(), "L2 rel", np.linalg.norm(g - r) / np.linalg.norm(r), "at", i, g[i], r[i], "count>1e-5:", int((d > 1e-5 * np.abs(r).max()).sum()))
