"""Which ATen calls of one optimisation step move whole tensors (copy_ / clone / contiguous / cat /
add on large operands), and from which Python line or autograd node.  A TorchDispatchMode, so it
also sees the calls the autograd engine makes in backward.
usage (GPU box): python tools/copy_trace.py [--backbone ResNet18] [--min-mb 8]"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for _d in ("FWD", "BWD", "WRW"):
    os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_" + _d, "0")
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="ResNet18")
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--height", type=int, default=192)
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--min-mb", type=float, default=8.0)
args = ap.parse_args()
args.amp_bf16 = args.channels_last = False
args.noise = "kernel"
from mono_vifi_amd.bench_train import TrainStep  # noqa: E402

WATCH = ("copy_", "clone", "contiguous", "cat", "stack", "add", "add_", "zeros_like", "zero_", "fill_",
         "new_zeros", "zeros", "sum", "_to_copy", "mul", "select_backward", "slice_backward")
seen = collections.defaultdict(lambda: [0, 0.0])


class Trace(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, fargs=(), kwargs=None):
        out = func(*fargs, **(kwargs or {}))
        name = func.__name__.split(".")[0]
        if name in WATCH:
            tens = [a for a in fargs if torch.is_tensor(a)]
            if fargs and isinstance(fargs[0], (list, tuple)):
                tens += [a for a in fargs[0] if torch.is_tensor(a)]
            res = out if torch.is_tensor(out) else None
            mb = max([t.numel() * t.element_size() for t in tens + ([res] if res is not None else [])] or [0]) / 2**20
            if mb >= args.min_mb:
                node = torch._C._current_autograd_node()
                if node is not None:
                    where = "bwd:" + node.name()
                else:
                    st = [f for f in traceback.extract_stack() if "mono-vifi_amd" in f.filename or "mono_vifi_amd" in f.filename]
                    where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in st[-3:][::-1])
                shp = [tuple(t.shape) for t in tens[:2]]
                contig = [t.is_contiguous() for t in tens[:2]]
                key = (name, str(shp), str(contig), where)
                seen[key][0] += 1
                seen[key][1] += mb
        return out


torch.cuda.set_device(0)
step = TrainStep(args, 0, 1, torch.device("cuda", 0))
for _ in range(2):
    step()
torch.cuda.synchronize()
with Trace():
    step()
torch.cuda.synchronize()
tot = 0.0
for k, (n, mb) in sorted(seen.items(), key=lambda kv: -kv[1][1])[:70]:
    tot += mb
    print(f"{mb:9.1f} MB x{n:3d} {k[0]:10s} {k[1]:52s} {k[2]:16s} {k[3]}")
print("total MB listed", tot)
