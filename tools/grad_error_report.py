#!/usr/bin/env python3
"""Where does the per-element gradient gap under --no_ssim / --avg_reprojection / --disable_automasking come from
?  The arbiter is the REFERENCE ITSELF evaluated in float64 on the same inputs
(tests/golden/g4_f64_C2_*.npz, written by tests/golden/make_golden.py g4d: gradients at the same 4,096 sample positions
as the fp32 captures, + a flag where its argmin / auto-mask differs from the fp32 run within 2 px: a flipped selection
is another function, not a rounding error -- excluded).  Against that truth, per flag set at the BASELINE shape C2:
the reference's own fp32 samples, the CPU oracle (fp32, fp64 folds) and -- on a GPU box -- the training kernel.
All errors are |d| / max |g| (the test's metric) AND absolute, because max |g| differs by flag set.

    python tools/grad_error_report.py            (CPU: reference fp32 and oracle; GPU box: + the HIP kernel)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

CASES = [("default", "g4_full_C2", 0, True), ("no_ssim", "g4_flags_C2_no_ssim", 1, True),
         ("avg", "g4_flags_C2_avg", 2, True), ("noauto", "g4_flags_C2_noauto", 4, False)]


def main():
    from conftest import load_golden
    from mono_vifi_amd import synthetic
    from oracle import oracle as O
    gpu = torch.cuda.is_available()
    if gpu:
        from mono_vifi_amd import ops
        dev = torch.device("cuda", 0)
        t = lambda a, g=False: torch.from_numpy(np.ascontiguousarray(a)).to(dev).requires_grad_(g)  # noqa: E731
    print("flag set   max|g| (fp64)   | reference fp32      | oracle (CPU)        | HIP kernel          | HIP vs reference fp32")
    print("                           | max|d|/max|g|  abs  | max|d|/max|g|  abs  | max|d|/max|g|  abs  | max|d|/max|g|")
    worst = []
    for name, f32name, flags, use_mask in CASES:
        g = load_golden(f32name)
        d = load_golden("g4_f64_C2_" + name)
        B, H, W = (int(v) for v in g["shape"])
        n = B * H * W
        s = g["sample_idx"]
        assert np.array_equal(s, d["sample_idx"])
        keep = ~d["selection_differs_near"].astype(bool)
        inp = synthetic.unit_inputs(int(g["seed"]), B, H, W, with_mask=use_mask)
        noise = inp["noise"][:, :1] if flags & 2 else inp["noise"]
        mask = inp["mask_rec"] if use_mask else None
        truth = d["grad_disp_s64"].astype(np.float64)
        gmax = float(d["grad_disp_max64"])
        ref32 = g["grad_disp_s"].astype(np.float64)
        ref = O.unit(inp["disp"], inp["tgt"], inp["src"], g["T"], inp["K"], inp["inv_K"], np.ascontiguousarray(noise), mask,
                     flags, want_grads=True)
        orc = ref["grad_disp"].reshape(n)[s].astype(np.float64)
        hip = None
        if gpu:
            disp, Tt = t(inp["disp"], True), t(g["T"], True)
            cfgt = (2, flags, 1e-3, 0.1, 100.0, 1e-7, True, True)
            loss, _, _, _, _ = ops.Unit.apply(disp, t(inp["tgt"]), Tt, t(inp["K"]), t(inp["inv_K"]),
                                              t(mask) if mask is not None else None,
                                              None if flags & 4 else t(np.ascontiguousarray(noise)), cfgt,
                                              t(inp["src"][0]), t(inp["src"][1]))
            loss.backward()
            hip = disp.grad.cpu().numpy().reshape(n)[s].astype(np.float64)

        def err(a):
            e = np.abs(a - truth)[keep]
            return f"{e.max() / gmax:.2e}  {e.max():.2e}"
        hv = f"{np.abs(hip - ref32)[keep].max() / gmax:.2e}" if hip is not None else "-"
        print(f"{name:9s}  {gmax:.3e}       | {err(ref32)} | {err(orc)} | {err(hip) if hip is not None else '-':19s} | {hv}"
              f"   (samples near a flipped selection, excluded: {int((~keep).sum())})")
        # the worst sample of the reference's fp32 run: how large are the values there?
        e32 = np.abs(ref32 - truth) * keep
        k = int(np.argmax(e32))
        b, rem = divmod(int(s[k]), H * W)
        y, x = divmod(rem, W)
        worst.append(f"{name}: reference-fp32's worst sample (b {b}, y {y}, x {x}): truth {truth[k]:+.4e}, reference fp32 "
                     f"{ref32[k]:+.4e}, oracle {orc[k]:+.4e}" + (f", HIP {hip[k]:+.4e}" if hip is not None else "") +
                     f"; disparity {float(inp['disp'][b, 0, y, x]):.4f} -> depth "
                     f"{1.0 / (0.01 + (10.0 - 0.01) * float(inp['disp'][b, 0, y, x])):.3f}")
    print()
    for w in worst:
        print(w)
    print("""
Reading.  (1) Against the float64 reference every fp32 evaluation -- the reference's own, the oracle, the kernel -- is
within 1e-4 of the TENSOR max per element; the kernel is also within 1e-4 of the reference's fp32 samples.  (2) The
1.2-1.4e-4 the round-4 test saw (and held at 2e-4, blaming "the cancelling SSIM adjoint" although --no_ssim has none)
was its metric: assert_grad_close on the 4,096 samples divides by the largest SAMPLE, which under these flag sets is
2-2.6 x smaller than the largest element of the tensor the 1e-4 bar is stated in.  (3) The term that produces the
largest errors is the same under every flag set: the worst samples are far pixels (disparity <= 0.002, depth 36-92 m)
where grad_disp = -range * depth^2 * dL/d depth multiplies by 1e4-1e5 a dL/d depth that is itself the difference of two
nearly cancelling terms of the perspective adjoint, d(u, v)/d depth = (P_row . r) / z - (P_2 . r) (x', y') / z^2 (for a
far point the parallax they leave is tiny); the reference's fp32 order loses up to 8.7e-5 of max |g| there, the oracle
up to 1.0e-4, the kernel up to 4.4e-5.  Nobody is the outlier.  The test now uses the tensor max (fixture
g4_f64_C2_*: grad_disp_max64) and holds the kernel to 1e-4 per element against both references.""")


if __name__ == "__main__":
    main()
