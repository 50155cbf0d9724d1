#!/usr/bin/env python3
"""Per-layer convolution table of a training step (GPU box): which shapes the step spends its
MIOpen time on, and how far each is from the fp32 peaks -- the target list for a hand-written
convolution kernel.

    python tools/conv_table.py [--backbone ResNet18 --batch 12 --height 192 --width 640] [--top 25]

One eager step records every Conv2d / ConvTranspose2d call (module parameters, input shape, calls
per step); every unique entry is then timed alone -- forward, and backward (data + weight gradient
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
    import torch
    import torch.nn as nn
    seen, hooks = {}, []

    def hook(m, inp, out):
        x = inp[0]
        if not torch.is_tensor(x):
            return
        tr = isinstance(m, nn.ConvTranspose2d)
        d = dict(tr=tr, cin=m.in_channels, cout=m.out_channels, k=list(m.kernel_size), s=list(m.stride),
                 p=list(m.padding), d=list(m.dilation), g=m.groups, shape=list(x.shape), out=list(out.shape),
                 grad=bool(torch.is_grad_enabled() and (x.requires_grad or m.weight.requires_grad)))
        key = json.dumps(d, sort_keys=True)
        seen.setdefault(key, [d, 0])[1] += 1

    mods = list(t.models.values()) + [t.model_vfi_train]
    for mod in {id(m): m for m in mods}.values():
        for m in mod.modules():
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                hooks.append(m.register_forward_hook(hook))
    _, losses = t.process_batch(dict(batch))
    losses["loss"].backward()
    for h in hooks:
        h.remove()
    return list(seen.values())


def time_layer(d, reps=10):
    import torch
    import torch.nn as nn
    dev = torch.device("cuda", 0)
    cls = nn.ConvTranspose2d if d["tr"] else nn.Conv2d
    kw = dict(stride=d["s"], padding=d["p"], dilation=d["d"], groups=d["g"], bias=False)
    m = cls(d["cin"], d["cout"], d["k"], **kw).to(dev)
    x = torch.randn(d["shape"], device=dev, requires_grad=True)
    g = torch.randn(d["out"], device=dev)

    def ev():
        return torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        y = m(x)
        if d["grad"]:
            y.backward(g)
    torch.cuda.synchronize()
    f0, f1, b1 = ev(), ev(), ev()
    tf = tb = 0.0
    for _ in range(reps):
        x.grad = None
        m.weight.grad = None
        f0.record()
        y = m(x)
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
        N, _, _, _ = d["shape"]
        _, co, oh, ow = d["out"]
        macs = N * co * oh * ow * (d["cin"] // d["g"]) * d["k"][0] * d["k"][1]
        if d["tr"]:
            macs = d["shape"][0] * d["cin"] * d["shape"][2] * d["shape"][3] * (d["cout"] // d["g"]) * d["k"][0] * d["k"][1]
        tf, tb = time_layer(d)
        rows.append((n * (tf + tb), n, tf, tb, 2 * macs / 1e9, d))
    rows.sort(key=lambda r: -r[0])
    total = sum(r[0] for r in rows)
    print(f"{len(rows)} unique convolutions, {sum(r[1] for r in rows)} calls per step, {total:.1f} ms per step alone")
    print("ms/step  calls  fwd_ms  bwd_ms  GFLOP(fwd)  TF/s fwd  TF/s bwd  layer")
    for tot, n, tf, tb, gf, d in rows[:a.top]:
        desc = (f"{'deconv' if d['tr'] else 'conv'} {d['cin']}->{d['cout']} k{d['k'][0]} s{d['s'][0]} d{d['d'][0]} "
                f"g{d['g']} in {d['shape']}")
        bw = (2 * gf / tb / 1e0) if tb > 0 else 0.0
        print(f"{tot:7.2f}  {n:5d}  {tf:6.3f}  {tb:6.3f}  {gf:10.1f}  {gf / tf:8.1f}  {bw:8.1f}  {desc}")


if __name__ == "__main__":
    main()
