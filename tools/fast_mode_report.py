"""What the reference's evaluation order costs on CDNA4: exact mode vs the opt-in fast build of the unit kernel
(`make -C mono-vifi_amd/csrc fast` -> lib/libmvf_hotpath_fast.so: separable 3x3 window sums shared between the two
outputs of a lane, products folded in by fused multiply-adds, window / channel means by reciprocal multiplies, SSIM
quotient = v_rcp + one Newton step and ONE multiply, contracted formula; target statistics shared across the candidate
pairs as in exact mode).  The reference arithmetic being relaxed: layers.py:277-290 (nn.AvgPool2d sums its nine taps
row-major; products are rounded before they are summed; `/` is the IEEE divide).

Per BASELINE shape (C1 / C2 / C4 / C5) and build: us per unit, VALU instructions per pixel, fraction of the HBM peak
(bench.py --workload hotpath with its live PMC pass), and the deviation of one unit forward + backward (fused tile
kernel route): loss, argmin flips, sampling indices, grad_disp relative L2 / worst pixel over the tensor max, grad_T.

    python tools/fast_mode_report.py            # GPU box: builds nothing, needs both libraries -> JSON on stdout
    python tools/fast_mode_report.py --dump SHAPE out.npz      # (child: one unit through the library MVF_HOTPATH_LIB names)
Never the default, never in the parity suite."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "mono-vifi_amd", "lib")
SHAPES = {"C1": (4, 192, 640), "C2": (12, 192, 640), "C4": (8, 320, 1024), "C5": (12, 192, 512)}


def dump(shape, path):
    import torch
    from mono_vifi_amd import layers, synthetic
    dev = torch.device("cuda", 0)
    B, H, W = SHAPES[shape]
    inp = synthetic.unit_inputs(4242, B, H, W, with_mask=False, disp_mode="smooth")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    aa, tr = t(inp["axisangle"]), t(inp["translation"])
    T = torch.stack([layers.transformation_from_parameters(aa[k], tr[k], invert=(k == 1)) for k in range(2)], 0).detach()
    disp, Tt = t(inp["disp"]).requires_grad_(True), T.clone().requires_grad_(True)
    from mono_vifi_amd import ops
    cfg = dict(n=1, S=2, flags=ops._flags(False, False, False), smoothness=1e-3, min_depth=0.1, max_depth=100.0, eps=1e-7,
               want_mask=False, want_idx=True, want_ident=[False], want_sum=False, sum_in=False, sinks=None, n_tokens=0,
               mean_parts=None)
    res = ops.Units.apply(cfg, disp, t(inp["tgt"]), Tt, t(inp["K"]), t(inp["inv_K"]), None, t(inp["noise"]), None,
                          t(inp["src"][0]), t(inp["src"][1]))
    loss, argmin, idx = res[0][0], res[3], res[4]
    loss.backward()
    np.savez(path, loss=float(loss.detach()), argmin=argmin.cpu().numpy(), idx=idx.cpu().numpy(),
             gd=disp.grad.cpu().numpy(), gT=Tt.grad.cpu().numpy())
    # the SEPARATE forward and backward unit kernels (mvf_unit_fwd / mvf_unit_bwd, the non-training route: SURVEY 8d asks
    # for forward and backward on their own): library events around 20 launches each, 44 / 45 algorithmic B/px
    from mono_vifi_amd import _native as nat
    ops.UNIT_FWDBWD = False
    cfgt = (2, 0, 1e-3, 0.1, 100.0, 1e-7, False, False)
    args = (t(inp["tgt"]), Tt, t(inp["K"]), t(inp["inv_K"]), None, t(inp["noise"]), cfgt, t(inp["src"][0]), t(inp["src"][1]))

    def once():
        disp.grad = None
        Tt.grad = None
        ops.Unit.apply(disp, *args)[0].backward()
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    nat.check(nat.lib().mvf_profile_reset(), "reset")
    nat.check(nat.lib().mvf_profile_enable(1), "enable")
    for _ in range(20):
        once()
    torch.cuda.synchronize()
    nat.lib().mvf_profile_enable(0)
    px = B * H * W
    sep = {}
    for name, kid, bpp in (("fwd", nat.PROF_UNIT_FWD, 44), ("bwd", nat.PROF_UNIT_BWD, 45)):
        ms, n = nat.profile_read(kid)
        if n:
            us = ms / n * 1e3
            sep[name] = {"us_per_unit": round(us, 2), "frac": round(bpp * px / (us * 1e-6) / 8e12, 4), "launches": n}
    print("SEPARATE " + json.dumps(sep))


def compare(a, b):
    gd_a, gd_b = a["gd"].astype(np.float64), b["gd"].astype(np.float64)
    return {"loss_exact": float(a["loss"]), "loss_fast": float(b["loss"]),
            "loss_rel_diff": abs(float(a["loss"]) - float(b["loss"])) / abs(float(a["loss"])),
            "sampling_indices_equal": bool(np.array_equal(a["idx"], b["idx"])),
            "argmin_flips": int((a["argmin"] != b["argmin"]).sum()), "pixels": int(a["argmin"].size),
            "grad_disp_rel_l2": float(np.linalg.norm(gd_a - gd_b) / np.linalg.norm(gd_a)),
            "grad_disp_max_over_max": float(np.abs(gd_a - gd_b).max() / np.abs(gd_a).max()),
            "grad_disp_px_beyond_1e-4_of_max": int((np.abs(gd_a - gd_b) > 1e-4 * np.abs(gd_a).max()).sum()),
            "grad_T_max_over_max": float(np.abs(a["gT"] - b["gT"]).max() / np.abs(a["gT"]).max())}


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--dump":
        return dump(sys.argv[2], sys.argv[3])
    libs = {"exact": os.path.join(LIB, "libmvf_hotpath.so"), "fast": os.path.join(LIB, "libmvf_hotpath_fast.so")}
    for p in libs.values():
        assert os.path.exists(p), p + " (make -C mono-vifi_amd/csrc all fast)"
    rep = {"what": "exact mode vs the opt-in fast build of k_unit_fb (tools/fast_mode_report.py)", "shapes": {}}
    tmp = tempfile.mkdtemp(prefix="mvf_fast_", dir="/tmp")
    for name, (B, H, W) in SHAPES.items():
        row = {}
        for mode, lib in libs.items():
            env = dict(os.environ, MVF_HOTPATH_LIB=lib)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "hotpath", "--batch", str(B), "--height",
                                str(H), "--width", str(W), "--steps", "50", "--warmup", "5", "--no-cpu-baseline"],
                               env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                row[mode] = {"error": (r.stderr.strip().splitlines() or ["?"])[-1][:200]}
                continue
            rf = json.loads(line[-1])["roofline"]
            row[mode] = {k: rf.get(k) for k in ("us_per_unit", "frac", "achieved", "valu_instr_per_px", "valu_busy", "avg_us")}
            r2 = subprocess.run([sys.executable, os.path.abspath(__file__), "--dump", name, os.path.join(tmp, f"{name}_{mode}.npz")],
                                env=env, check=True, timeout=600, capture_output=True, text=True)
            for ln in r2.stdout.splitlines():
                if ln.startswith("SEPARATE "):
                    row[mode]["separate_kernels"] = json.loads(ln[9:])
        if "us_per_unit" in row.get("exact", {}) and "us_per_unit" in row.get("fast", {}):
            row["fast_over_exact_time"] = round(row["fast"]["us_per_unit"] / row["exact"]["us_per_unit"], 4)
            row["deviation"] = compare(np.load(os.path.join(tmp, f"{name}_exact.npz")), np.load(os.path.join(tmp, f"{name}_fast.npz")))
        rep["shapes"][name] = row
    c2 = rep["shapes"].get("C2", {})
    rep["frac_fast"] = c2.get("fast", {}).get("frac")
    rep["us_per_unit_fast"] = c2.get("fast", {}).get("us_per_unit")
    rep["frac_exact"] = c2.get("exact", {}).get("frac")
    rep["us_per_unit_exact"] = c2.get("exact", {}).get("us_per_unit")
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
