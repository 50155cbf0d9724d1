"""torch.profiler view of one optimisation step: ATen ops by device time with input shapes
(which tensors the memcpy / cat / add / transpose time belongs to).
usage (GPU box): python tools/torch_prof.py [--backbone ResNet18] [--batch 12]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for _d in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _d, "0")
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="ResNet18")
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--height", type=int, default=192)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--rows", type=int, default=45)
args = ap.parse_args()
args.amp_bf16 = args.channels_last = False
args.noise = "kernel"
from mono_vifi_amd.bench_train import TrainStep  # noqa: E402
torch.cuda.set_device(0)
step = TrainStep(args, 0, 1, torch.device("cuda", 0))
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=args.rows,
                                                           max_name_column_width=40, max_shapes_column_width=70))
if os.environ.get("MVF_PROF_STACKS"):
    # python stacks of the large device copies / cats / adds
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof2:
        step()
        torch.cuda.synchronize()
    seen = {}
    for ev in prof2.events():
        if ev.name in ("aten::copy_", "aten::cat", "aten::add", "aten::add_", "aten::sum") and ev.input_shapes:
            dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
            if dt < 40:
                continue
            st = [s for s in (ev.stack or []) if "mono-vifi_amd" in s or "mono_vifi_amd" in s][:3]
            key = (ev.name, str(ev.input_shapes[:2]), tuple(st))
            seen.setdefault(key, [0, 0.0])
            seen[key][0] += 1
            seen[key][1] += dt
    for k, (n, t) in sorted(seen.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{t / 1e3:8.3f} ms x{n:3d} {k[0]:12s} {k[1]:60s} {' <- '.join(s.split('/')[-1] for s in k[2])}")
