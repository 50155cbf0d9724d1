"""Whole-tensor gradient error of the training kernel against the oracle's adjoint evaluated in DOUBLE
(oracle.unit(..., adjoint64=True): every decision the fp32 forward's, every value in double; pinned to the reference
evaluated in float64 by tests/test_oracle_golden.py) at the BASELINE shapes and flag sets: max |kernel - f64| / max |f64|
over ALL pixels, relative L2, the same for the fp32 oracle, and for grad_T.   python tools/grad_vs_f64_adjoint.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from mono_vifi_amd import ops, synthetic  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda:0")
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
cases = [("g4_full_" + c, None) for c in ("C1", "C2", "C4", "C5")] + [("g4_flags_C2_" + n, n) for n in ("no_ssim", "avg", "noauto")]
print("case                 kernel-vs-f64 max   L2        oracle32-vs-f64 max   grad_T kernel-vs-f64")
for name, flagset in cases:
    g = load_golden(name)
    B, H, W = (int(v) for v in g["shape"])
    if flagset is None:
        flags, use_mask = 0, True
    else:
        flags = int(g["flags"][0]) * 1 + int(g["flags"][1]) * 2 + int(g["flags"][2]) * 4
        use_mask = bool(int(g["use_mask"]))
    inp = synthetic.unit_inputs(int(g["seed"]), B, H, W, with_mask=use_mask)
    noise_np = np.ascontiguousarray(inp["noise"][:, :1] if flags & 2 else inp["noise"])
    mask_np = inp["mask_rec"] if use_mask else None
    disp = t(inp["disp"]).requires_grad_(True)
    Tt = t(g["T"]).requires_grad_(True)
    loss, _, argmin, _, _ = ops.Unit.apply(disp, t(inp["tgt"]), Tt, t(inp["K"]), t(inp["inv_K"]),
                                           t(mask_np) if use_mask else None, None if flags & 4 else t(noise_np),
                                           (2, flags, 1e-3, 0.1, 100.0, 1e-7, True, False), t(inp["src"][0]), t(inp["src"][1]))
    loss.backward()
    ref = O.unit(inp["disp"], inp["tgt"], inp["src"], g["T"], inp["K"], inp["inv_K"], noise_np, mask_np, flags,
                 want_grads=True, adjoint64=True)
    gd = disp.grad.cpu().numpy().astype(np.float64)
    r64, r32 = ref["grad_disp64"], ref["grad_disp"].astype(np.float64)
    mx = np.abs(r64).max()
    e = np.abs(gd - r64)
    print(f"{name:20s} {e.max() / mx:.3e}        {np.linalg.norm(gd - r64) / np.linalg.norm(r64):.2e}  {np.abs(r32 - r64).max() / mx:.3e}"
          f"             {np.abs(Tt.grad.cpu().numpy() - ref['grad_T64']).max() / np.abs(ref['grad_T64']).max():.2e}", flush=True)
    i = int(e.argmax())
    b, rem = divmod(i, H * W)
    print(f"    worst pixel b={b} y={rem // W} x={rem % W}: kernel {gd.reshape(-1)[i]:.6e} f64 {r64.reshape(-1)[i]:.6e} fp32-oracle {r32.reshape(-1)[i]:.6e}; "
          f"pixels above 1e-4 of max: {(e > 1e-4 * mx).sum()} of {e.size}")
