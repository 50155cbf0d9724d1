#!/usr/bin/env python3
"""How far are the training kernel's gradients from the oracle and from the reference's samples at the BASELINE
shapes?  (north_star: 1e-4 relative; the full-size tests hold the tensor to 1e-4 relative L2 and each element to
1e-3 of the tensor max -- tests/conftest.py::assert_grad_close.)  GPU box: python tools/grad_error_report.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    from conftest import load_golden, rel_err, rel_l2
    from mono_vifi_amd import ops, synthetic
    from oracle import oracle as O
    dev = torch.device("cuda", 0)
    t = lambda a, g=False: torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(g)  # noqa: E731
    print("cfg    vs oracle: grad_disp relL2  max/tensor-max   grad_T max/max | vs reference samples: relL2  max/max")
    for cfg in ("C1", "C2", "C4", "C5"):
        g = load_golden("g4_full_" + cfg)
        B, H, W = (int(v) for v in g["shape"])
        inp = synthetic.unit_inputs(int(g["seed"]), B, H, W, with_mask=True)
        disp, Tt = t(inp["disp"], True), t(g["T"], True)
        cfgt = (2, 0, 1e-3, 0.1, 100.0, 1e-7, True, True)
        loss, _, _, _, _ = ops.Unit.apply(disp, t(inp["tgt"]), Tt, t(inp["K"]), t(inp["inv_K"]), t(inp["mask_rec"]),
                                          t(inp["noise"]), cfgt, t(inp["src"][0]), t(inp["src"][1]))
        loss.backward()
        gd = disp.grad.cpu().numpy()
        ref = O.unit(inp["disp"], inp["tgt"], inp["src"], g["T"], inp["K"], inp["inv_K"], inp["noise"], inp["mask_rec"],
                     0, want_grads=True)
        n = B * H * W
        s = g["sample_idx"]
        print(f"{cfg}     {rel_l2(gd, ref['grad_disp']):.2e}   {rel_err(gd, ref['grad_disp']):.2e}          "
              f"{rel_err(Tt.grad.cpu().numpy(), ref['grad_T']):.2e}     |  {rel_l2(gd.reshape(n)[s], g['grad_disp_s']):.2e}  "
              f"{rel_err(gd.reshape(n)[s], g['grad_disp_s']):.2e}   (oracle vs reference samples: "
              f"{rel_l2(ref['grad_disp'].reshape(n)[s], g['grad_disp_s']):.2e} {rel_err(ref['grad_disp'].reshape(n)[s], g['grad_disp_s']):.2e})")


if __name__ == "__main__":
    main()
