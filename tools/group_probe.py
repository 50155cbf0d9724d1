"""Probe (GPU): G encoder+decoder calls at batch B vs one call at batch G*B, fwd+bwd, fp32.
Indicates what grouping the independent encoder invocations of a step buys on
one GPU before any collective is involved.  Usage: python tools/group_probe.py [G] [B]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from mono_vifi_amd.networks import monodepth2  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda:0")
enc = monodepth2.DepthEncoder(18, False).to(dev)
dec = monodepth2.DepthDecoder(enc.num_ch_enc).to(dev)
x = torch.rand(G * B, 3, 192, 640, device=dev)


def run(chunks):
    tot = 0
    for c in x.chunk(chunks):
        out = dec(enc(c))
        tot = tot + out[("disp", 0)].mean()
    tot.backward()


for chunks in (G, 1, G, 1):
    for _ in range(2):
        run(chunks)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5):
        run(chunks)
    torch.cuda.synchronize()
    print(f"chunks={chunks}: {(time.time() - t0) / 5 * 1e3:.1f} ms per {G}x{B} images", flush=True)
