#!/usr/bin/env python3
"""Per-layer convolution table of a training step (GPU box): which shapes the step spends its
MIOpen time on, and how far each is from the fp32 peaks -- the target list for a hand-written
convolution kernel.

    python tools/conv_table.py [--backbone ResNet18 --batch 12 --height 192 --width 640] [--top 25]

One eager step records every convolution call at the dispatcher (weight / input shape, calls per
step); every unique entry is then timed alone -- forward, and backward (data + weight gradient
together) -- with HIP events after a warm-up, and listed with its FLOPs, achieved TFLOP/s and share
of the step's convolution time.  Developer tool; never imported by the package, tests or bench."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def record(t, batch):
    """Every convolution call of one eager step, recorded at the dispatcher (most convolutions here run
    as F.conv2d inside layers.conv_bias_act, not through a module call): [description, calls]."""
    from torch.utils._python_dispatch import TorchDispatchMode
    seen = {}

    class Rec(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            if func.overloadpacket.__name__ == "convolution":
                x, w, bias, stride, pad, dil, transposed, opad, groups = args[:9]
                d = dict(tr=bool(transposed), shape=list(x.shape), wshape=list(w.shape), out=list(out.shape),
                         s=list(stride), p=list(pad), d=list(dil), op=list(opad), g=int(groups),
                         grad=bool(x.requires_grad or w.requires_grad))
                seen.setdefault(json.dumps(d, sort_keys=True), [d, 0])[1] += 1
            return out

    with Rec():
        _, losses = t.process_batch(dict(batch))
        losses["loss"].backward()
    return list(seen.values())


def time_layer(d, reps=10):
    import torch
    import torch.nn.functional as F
    dev = torch.device("cuda", 0)
    x = torch.randn(d["shape"], device=dev, requires_grad=True)
    w = torch.randn(d["wshape"], device=dev, requires_grad=True)
    g = torch.randn(d["out"], device=dev)
    if d["tr"]:
        conv = lambda v: F.conv_transpose2d(v, w, None, d["s"], d["p"], d["op"], d["g"], d["d"])   # noqa: E731
    else:
        conv = lambda v: F.conv2d(v, w, None, d["s"], d["p"], d["d"], d["g"])                      # noqa: E731

    def ev():
        return torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        y = conv(x)
        if d["grad"]:
            y.backward(g)
    torch.cuda.synchronize()
    f0, f1, b1 = ev(), ev(), ev()
    tf = tb = 0.0
    for _ in range(reps):
        x.grad = None
        w.grad = None
        f0.record()
        y = conv(x)
        f1.record()
        if d["grad"]:
            y.backward(g)
        b1.record()
        torch.cuda.synchronize()
        tf += f0.elapsed_time(f1)
        tb += f1.elapsed_time(b1)
    return tf / reps, tb / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="ResNet18")
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    from graph_bisect import make_trainer
    t, batch = make_trainer(a)
    layers = record(t, batch)
    del t, batch
    rows = []
    for d, n in layers:
        # multiply-accumulates of the forward
        kh, kw = d["wshape"][2], d["wshape"][3]
        if d["tr"]:     # weight [cin, cout/g, kh, kw]: every INPUT element feeds cout/g x kh x kw outputs
            macs = d["shape"][0] * d["shape"][1] * d["shape"][2] * d["shape"][3] * d["wshape"][1] * kh * kw
        else:           # weight [cout, cin/g, kh, kw]: every OUTPUT element reads cin/g x kh x kw inputs
            macs = d["out"][0] * d["out"][1] * d["out"][2] * d["out"][3] * d["wshape"][1] * kh * kw
        tf, tb = time_layer(d)
        rows.append((n * (tf + tb), n, tf, tb, 2 * macs / 1e9, d))
    rows.sort(key=lambda r: -r[0])
    total = sum(r[0] for r in rows)
    print(f"{len(rows)} unique convolutions, {sum(r[1] for r in rows)} calls per step, {total:.1f} ms per step alone")
    print("ms/step  calls  fwd_ms  bwd_ms  GFLOP(fwd)  TF/s fwd  TF/s bwd  layer")
    for tot, n, tf, tb, gf, d in rows[:a.top]:
        desc = (f"{'deconv' if d['tr'] else 'conv'} w{d['wshape']} s{d['s'][0]} d{d['d'][0]} g{d['g']} in {d['shape']}"
                f"{'' if d['grad'] else ' (no grad)'}")
        bw = (2 * gf / tb / 1e0) if tb > 0 else 0.0
        print(f"{tot:7.2f}  {n:5d}  {tf:6.3f}  {tb:6.3f}  {gf:10.1f}  {gf / tf:8.1f}  {bw:8.1f}  {desc}")


if __name__ == "__main__":
    main()
