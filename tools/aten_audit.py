"""Which ATen launches of a training step move how many bytes, and from which call site?

    python tools/aten_audit.py [--backbone ResNet18] [--top 40]

Runs the bench's training step (ResNet18, batch 12, 640x192 by default) under torch.profiler with
shapes and Python stacks recorded, and lists the element-wise / copy ATen operators (cat, stack, add,
copy_, clamp, mul, fill, ...) by device time: per (operator, input shapes) the launches per step,
microseconds per step, and the first frame of this package on the Python stack.  The conv / batch-norm
operators of the networks (MIOpen: out of scope) are summed in one line for reference.
Output: gpurun_out/aten_audit_<backbone>.txt."""
import argparse
import collections
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="ResNet18")
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    import mono_vifi_amd as pkg
    pkg.use_shipped_miopen_db()
    import torch
    from torch.profiler import ProfilerActivity, profile
    from mono_vifi_amd.bench_train import TrainStep
    dev = torch.device("cuda", 0)
    args = types.SimpleNamespace(batch=a.batch, height=a.height, width=a.width, backbone=a.backbone)
    step = TrainStep(args, 0, 1, dev)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
                 with_stack=True) as prof:
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0, ""])
    conv = [0, 0.0]
    for ev in prof.events():
        dt = getattr(ev, "self_device_time_total", 0.0) or 0.0
        if dt <= 0:
            continue
        name = ev.name
        if "conv" in name or "batch_norm" in name or "miopen" in name:
            conv[0] += 1
            conv[1] += dt
            continue
        shapes = str(ev.input_shapes)[:150]
        site = ""
        for fr in (ev.stack or []):
            if "mono" in fr and "torch/" not in fr:
                site = fr.strip()[-90:]
                break
        k = (name, shapes, site)
        agg[k][0] += 1
        agg[k][1] += dt
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    out = [f"# {a.backbone} B{a.batch} {a.width}x{a.height}: ATen / own operators by device time per step "
           f"({a.steps} profiled steps); conv + batch-norm operators (MIOpen): {conv[1] / a.steps / 1e3:.2f} ms/step "
           f"in {conv[0] // a.steps} launches",
           f"{'us/step':>9} {'n/step':>7}  operator | input shapes | call site"]
    tot = 0.0
    for (name, shapes, site), (n, dt, _) in rows[:a.top]:
        out.append(f"{dt / a.steps:9.1f} {n / a.steps:7.1f}  {name} | {shapes} | {site}")
    for _, (n, dt, _) in rows:
        tot += dt
    out.append(f"total non-conv device time: {tot / a.steps / 1e3:.2f} ms/step")
    by_op = collections.defaultdict(float)
    for (name, _, _), (n, dt, _) in rows:
        by_op[name] += dt
    out.append("by operator: " + ", ".join(f"{k} {v / a.steps / 1e3:.2f}" for k, v in
                                           sorted(by_op.items(), key=lambda kv: -kv[1])[:25]))
    text = "\n".join(out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", f"aten_audit_{a.backbone}.txt")
    open(path, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
