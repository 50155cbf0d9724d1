"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI load/export checks (CPU only).
`-m gpu`       : parity tests proper -- HIP path through the C ABI vs the oracle.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / max(|b|) -- the tensor-level relative error used for fp32 tolerances."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2 over the whole tensor."""
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (den if den > 0 else 1.0))


def assert_grad_close(a, b, tol=1e-4, what="grad", max_tol=None):
    """Tolerance for full-size gradient tensors (1e4..1e6 elements).

    The 1e-4 relative bar is applied to the tensor (relative L2 error).  Per element the
    bound is 1e-3 of the tensor max, NOT 1e-4: the SSIM adjoint cancels catastrophically
    (d sigma/dx = 2(x - mu)/9 with x ~ mu), so the reference's own fp32 autograd, the fp32
    oracle and this kernel -- three legitimate fp32 evaluation orders -- sit ~1e-4 of the
    tensor max apart at their worst pixel (measured: reference vs oracle 2.2e-4 at C1,
    DESIGN.md section 6).  The small golden fixtures are still held to 1e-4 per element.
    `max_tol` overrides the per-element bound: the training kernel's gradients are held to 1e-4 per element
    against the REFERENCE's own sampled gradients at the BASELINE shapes (measured worst 5.3e-5) and to 5e-4
    against the oracle's whole tensors (measured worst 2.7e-4 of 1.5-2.6 M pixels:
    profiles/r03_grad_error_report.txt)."""
    l2 = rel_l2(a, b)
    mx = rel_err(a, b)
    max_tol = 10 * tol if max_tol is None else max_tol
    assert l2 <= tol, f"{what}: relative L2 error {l2:.3e} > {tol}"
    assert mx <= max_tol, f"{what}: max error {mx:.3e} of tensor max > {max_tol}"


# Per-element guards (of the tensor max) of the WHOLE-tensor gradient comparisons at the BASELINE shapes, per configuration
# and arbiter: the worst this build measured in round 6 x 1.25 (profiles/r06_grad_error_report.txt; VERDICT r05 item 7: a
# flat 5e-4 let a regression hide).  "oracle": the fp32 oracle's whole tensor (two fp32 evaluation orders);
# "f64": the oracle's adjoint evaluated in double.  The 1e-4 bar of north_star is held where the reference's own
# numbers exist (its 4,096 samples, fp32 and float64); these guards bound the far, high-contrast pixels where every
# fp32 evaluation -- the reference's autograd included -- amplifies the rounding of a cancelling parallax term.
GRAD_GUARD = {
    "oracle": {"C1": 1.4e-4, "C2": 3.0e-4, "C4": 2.5e-4, "C5": 2.4e-4, "C2_no_ssim": 6.3e-4, "C2_avg": 2.9e-4, "C2_noauto": 3.7e-4},
    "f64": {"C1": 1.1e-4, "C2": 3.0e-4, "C4": 1.9e-4, "C5": 2.8e-4, "C2_no_ssim": 5.2e-4, "C2_avg": 2.5e-4, "C2_noauto": 3.4e-4},
}
# pixels beyond 1e-4 of the tensor max (of 0.5-2.6 M), same run: the tracked regression number; bound = 2 x measured + 5
GRAD_BEYOND = {
    "oracle": {"C1": 2, "C2": 5, "C4": 4, "C5": 3, "C2_no_ssim": 74, "C2_avg": 8, "C2_noauto": 57},
    "f64": {"C1": 0, "C2": 5, "C4": 2, "C5": 1, "C2_no_ssim": 42, "C2_avg": 4, "C2_noauto": 39},
}


def grad_guard(arbiter, cfg, got, want, what=""):
    """Assert the per-element guard of `cfg` and append the tracked regression numbers (worst pixel over the tensor max,
    pixels beyond 1e-4 of it) to gpurun_out/grad_error_counts.tsv (copied to profiles/ per round)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    mx = float(np.abs(want).max())
    e = np.abs(got - want)
    worst, beyond = float(e.max() / mx), int((e > 1e-4 * mx).sum())
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "grad_error_counts.tsv"), "a") as f:
            f.write(f"{arbiter}\t{cfg}\t{what}\t{worst:.3e}\t{beyond}\t{e.size}\t{GRAD_GUARD[arbiter][cfg]:.2e}\n")
    except OSError:
        pass
    assert worst <= GRAD_GUARD[arbiter][cfg], f"{what} {cfg} vs {arbiter}: worst pixel {worst:.3e} of the tensor max > {GRAD_GUARD[arbiter][cfg]:.2e}"
    cap = 2 * GRAD_BEYOND[arbiter][cfg] + 5
    assert beyond <= cap, f"{what} {cfg} vs {arbiter}: {beyond} pixels beyond 1e-4 of the tensor max (tracked bound {cap})"
    return worst, beyond


def torch_flow_warp(img, flow):
    """Test-only torch restatement of the reference's IFRNet.warp (networks/IFRNet.py:7-15),
    injected into mono_vifi_amd.networks.ifrnet.WARP_IMPL by the CPU tests that compare the
    network restatements with the reference's modules (the product uses the HIP kernel)."""
    import torch
    import torch.nn.functional as F
    B, _, H, W = flow.shape
    xs = torch.linspace(-1.0, 1.0, W).view(1, 1, 1, W).to(flow)
    ys = torch.linspace(-1.0, 1.0, H).view(1, 1, H, 1).to(flow)
    gx = xs + flow[:, 0:1] / ((W - 1.0) / 2.0)
    gy = ys + flow[:, 1:2] / ((H - 1.0) / 2.0)
    grid = torch.cat([gx.expand(B, 1, H, W), gy.expand(B, 1, H, W)], 1).permute(0, 2, 3, 1)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=True)


@pytest.fixture
def cpu_warp(monkeypatch):
    from mono_vifi_amd.networks import ifrnet
    monkeypatch.setattr(ifrnet, "WARP_IMPL", torch_flow_warp)
    import mono_vifi_amd.networks.fusion_module as fm
    return ifrnet


def affine_case(seed, B, C, H, W, lo=0.0, hi=1.0):
    rng = np.random.default_rng(seed)
    x = (lo + (hi - lo) * rng.random((B, C, H, W))).astype(np.float32)
    ratio = rng.uniform(1.2, 2.0, size=(B,)).astype(np.float32)
    angle = rng.uniform(-5.0, 5.0, size=(B,)).astype(np.float32)
    box = np.zeros((B, 4), np.int32)
    for b in range(B):                      # datasets/mono_dataset.py:110-149
        r = float(ratio[b])
        h_re, w_re = int(H * r), int(W * r)
        w0 = int((w_re - W) * rng.random())
        h0 = int((h_re - H) * rng.random())
        box[b] = (round(w0 / r), round(h0 / r), min(round(W / r), W - round(w0 / r)),
                  min(round(H / r), H - round(h0 / r)))
    return x, angle, box, ratio
