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


def trace_copies(step, a):
    """Forward-pass data movement by call site: bytes written by cat / stack / contiguous / clone / float()."""
    import traceback
    import torch
    log = collections.defaultdict(lambda: [0, 0])

    def site():
        frames = [f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                  for fr in reversed(traceback.extract_stack()[:-2])
                  if "mono" in fr.filename and "tools/" not in fr.filename]
        return " < ".join(frames[:3]) if frames else "?"

    def note(kind, out):
        if torch.is_tensor(out) and out.is_cuda:
            k = (kind, site(), tuple(out.shape))
            log[k][0] += 1
            log[k][1] += out.numel() * out.element_size()

    o_cat, o_stack, o_contig, o_clone = torch.cat, torch.stack, torch.Tensor.contiguous, torch.Tensor.clone

    def cat(*x, **k):
        r = o_cat(*x, **k)
        note("cat", r)
        return r

    def stack(*x, **k):
        r = o_stack(*x, **k)
        note("stack", r)
        return r

    def contiguous(self, *x, **k):
        was = self.is_contiguous(*x, **k) if not x else True
        r = o_contig(self, *x, **k)
        if not was:
            note("contiguous", r)
        return r

    def clone(self, *x, **k):
        r = o_clone(self, *x, **k)
        note("clone", r)
        return r

    torch.cat, torch.stack, torch.Tensor.contiguous, torch.Tensor.clone = cat, stack, contiguous, clone
    try:
        step()
        torch.cuda.synchronize()
    finally:
        torch.cat, torch.stack, torch.Tensor.contiguous, torch.Tensor.clone = o_cat, o_stack, o_contig, o_clone
    rows = sorted(log.items(), key=lambda kv: -kv[1][1])
    out = [f"# {a.backbone} B{a.batch} {a.width}x{a.height}: forward-pass copies of one step by call site (bytes WRITTEN; "
           f"each is read once too)", f"{'MB':>9} {'calls':>6}  kind | call site | shape"]
    for (kind, where, shape), (n, by) in rows[:a.top]:
        out.append(f"{by / 1e6:9.1f} {n:6d}  {kind} | {where} | {list(shape)}")
    out.append(f"total {sum(v[1] for v in log.values()) / 1e9:.2f} GB written by {sum(v[0] for v in log.values())} calls")
    text = "\n".join(out)
    path = os.path.join(ROOT, "gpurun_out", f"aten_copies_{a.backbone}.txt")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    open(path, "w").write(text + "\n")
    print(text)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="ResNet18")
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--copies", action="store_true",
                    help="instead of the profiler: log every torch.cat / stack / non-trivial .contiguous() / clone of the "
                         "forward pass with its bytes and Python call site")
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
    if a.copies:
        trace_copies(step, a)
        return
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
