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
