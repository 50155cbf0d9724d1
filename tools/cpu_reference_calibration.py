"""BUILD CONTAINER ONLY (needs /root/reference): the reference's own CPU path timed beside the two stand-ins that travel to
the GPU box -- the C / OpenMP port (oracle/mvf_oracle.c, `cpu_baseline` of the bench line) and the un-fused ATen form
(oracle/torch_unfused.py, `cpu_unfused` of the detail file) -- on the SAME cores, one unit forward + backward at C2
(batch 12, 640 x 192; reference: 2 x Trainer.generate_images_pred + Trainer.compute_losses_base + backward,
train.py:956-1051).  The ratios calibrate what the GPU box's stand-in numbers say about the reference there
(VERDICT r05 item 5).  Writes JSON on stdout.

    PYTHONDONTWRITEBYTECODE=1 python tools/cpu_reference_calibration.py [threads]"""
import importlib.util
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
import torch  # noqa: E402

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.set_num_threads(threads)
spec = importlib.util.spec_from_file_location("mvf_make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)               # imports the reference unmodified behind its six stub modules (no capture runs)
from mono_vifi_amd import synthetic  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import torch_unfused as U  # noqa: E402

B, H, W = 12, 192, 640
inp = synthetic.unit_inputs(4321, B, H, W, with_mask=True)
T_np = np.stack([O.pose(inp["axisangle"][k], inp["translation"][k], invert=(k == 1)) for k in range(2)], 0)
t = mg.t


def reference_unit():
    fs = mg.fake_self(B, H, W)
    disp = t(inp["disp"]).requires_grad_(True)
    Ts = [t(T_np[k]).requires_grad_(True) for k in range(2)]
    K, iK = t(inp["K"]), t(inp["inv_K"])
    srcs = [t(inp["src"][k]) for k in range(2)]
    with mg.FixedRandn(t(inp["noise"])):
        warped = [mg.Trainer.generate_images_pred(fs, {("disp", 0): disp}, Ts[k], srcs[k], K, iK) for k in range(2)]
        loss, _ = mg.Trainer.compute_losses_base(fs, {("disp", 0): disp}, t(inp["tgt"]), warped, srcs, t(inp["mask_rec"]))
    loss.backward()
    return float(loss.detach())


def unfused_unit():
    return U.unit(inp["disp"], inp["tgt"], inp["src"], T_np, inp["K"], inp["inv_K"], inp["noise"], inp["mask_rec"], 0)["loss"]


def port_unit():
    return O.unit(inp["disp"], inp["tgt"], inp["src"], T_np, inp["K"], inp["inv_K"], inp["noise"], inp["mask_rec"], 0,
                  want_grads=True)["loss"]


O.set_threads(threads)
libc = __import__("ctypes").CDLL(None)              # large buffers from the heap, reused (as bench.cpu_baseline does)
libc.mallopt(-3, 1 << 30), libc.mallopt(-1, (1 << 31) - 1)
fns = (("reference", reference_unit), ("torch_unfused", unfused_unit), ("c_port_openmp", port_unit))
times = {n: [] for n, _ in fns}
loss = {}
for n, f in fns:
    f()                                   # warm-up
for _ in range(7):                        # interleaved rounds: drift of the shared VM hits all three alike
    for n, f in fns:
        t0 = time.perf_counter()
        loss[n] = f()
        times[n].append(time.perf_counter() - t0)
out = {"what": "one unit forward + backward at C2 (B 12, 640x192), 7 interleaved rounds after one warm-up, this container; "
               "ratios from the per-formulation MINIMUM (the least disturbed run) and from the medians",
       "threads": threads, "cpus": len(os.sched_getaffinity(0))}
for n, _ in fns:
    mn, md = min(times[n]), statistics.median(times[n])
    out[n] = {"seconds_per_unit_min": round(mn, 4), "seconds_per_unit_median": round(md, 4), "samples_per_s": round(B / mn, 2),
              "images_per_s_hot_path_of_a_step": round(B / (9 * mn), 3), "loss": loss[n]}
for key, stat in (("min", "seconds_per_unit_min"), ("median", "seconds_per_unit_median")):
    out[f"torch_unfused_over_reference_{key}"] = round(out["reference"][stat] / out["torch_unfused"][stat], 3)
    out[f"c_port_over_reference_{key}"] = round(out["reference"][stat] / out["c_port_openmp"][stat], 3)
print(json.dumps(out, indent=1))
