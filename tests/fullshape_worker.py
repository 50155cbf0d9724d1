"""Child process of tests/test_fullshape_steps.py: ONE BASELINE.json training configuration at its full
per-GPU shape on cuda:0 -- fused unit kernels vs the staged generate_images_pred + compute_losses_base
pair on the same weights and batch, then one whole optimisation step.  Prints one JSON line.

A process of its own so that (a) MIOpen reads MIOPEN_FIND_MODE / MIOPEN_USER_DB_PATH from a clean
environment (it caches them at first use), (b) the activations of a batch-12 step are returned to the
system before the next configuration starts."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    backbone, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    log_dir = sys.argv[5]
    from mono_vifi_amd import synthetic
    from mono_vifi_amd.options import default_options
    from mono_vifi_amd.trainer import Trainer
    t0 = time.perf_counter()
    opts = default_options(batch_size=B, height=H, width=W, backbone=backbone, use_affine=True,
                           num_workers=0, synthetic_len=4 * B, log_dir=log_dir, exp_name="full",
                           log_frequency=10 ** 9, save_frequency=10 ** 9, inkernel_noise=False)
    t = Trainer(opts)
    t.set_train()
    b = synthetic.training_batch(77, B, H, W)
    batch = {k: torch.from_numpy(np.ascontiguousarray(v)).to(t.device) for k, v in b.items()}
    g = torch.Generator(device=t.device).manual_seed(3)
    t.tie_break_noise = torch.randn((B, 2, H, W), device=t.device, generator=g)
    state0 = {k: {n: v.clone() for n, v in m.state_dict().items()} for k, m in t.models.items()}
    out = {}
    # "warm": the first pass over a convolution shape may run with other solvers than the passes after it
    for tag, fused in (("warm", True), ("fused", True), ("staged", False), ("staged2", False)):
        for k, m in t.models.items():
            m.load_state_dict(state0[k])
        t.opt.fused_units = fused
        torch.manual_seed(0)
        _, losses = t.process_batch(dict(batch))
        t.reducer.zero_grad()
        losses["loss"].backward()
        t.reducer.finish()
        out[tag] = (float(losses["loss"]), float(losses["loss_base"]), float(losses["loss_dc"]),
                    torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten()
                               for p in t.parameters_to_train]).double().clone())
    ref = out["staged"][3]
    rn = float(ref.norm())
    res = {"backbone": backbone, "shape": [B, H, W],
           "loss": {k: out[k][0] for k in ("fused", "staged")},
           "loss_base": {k: out[k][1] for k in ("fused", "staged")},
           "loss_dc": {k: out[k][2] for k in ("fused", "staged")},
           "grad_dev": float((out["fused"][3] - ref).norm()) / rn,
           "grad_noise": float((out["staged2"][3] - ref).norm()) / rn,
           "grad_finite": bool(torch.isfinite(out["fused"][3]).all()), "grad_norm": rn}
    t.opt.fused_units = True
    t.tie_break_noise = None
    t.opt.inkernel_noise = True          # the variant bench.py times
    before = [p.detach().clone() for p in t.parameters_to_train[:8]]
    losses = t.optimisation_step(dict(batch))
    torch.cuda.synchronize()
    res["step"] = {k: float(losses[k].detach()) for k in ("loss", "loss_base", "loss_dc")}
    res["updated"] = any(not torch.equal(a, p.detach()) for a, p in zip(before, t.parameters_to_train[:8]))
    res["seconds"] = round(time.perf_counter() - t0, 1)
    res["peak_gb"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
