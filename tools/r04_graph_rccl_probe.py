"""Probe (round 4): RCCL collectives inside a captured HIP graph, group of one on one GPU.
Which issue pattern survives capture + replay with ProcessGroupNCCL's watchdog alive?
    python tools/r04_graph_rccl_probe.py main|hook|hook_sync [env tweaks via the environment]"""
import os
import sys
import time

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", str(29900 + os.getpid() % 90))
os.environ.update(RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

mode = sys.argv[1] if len(sys.argv) > 1 else "main"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0, device_id=dev)
w = torch.nn.Parameter(torch.randn(1024, 1024, device=dev))
x = torch.randn(64, 1024, device=dev)
buf = torch.zeros(1024 * 1024, device=dev)
w.grad = buf.view_as(w)
handles = []


def hook(p):
    cs = torch.cuda.current_stream()
    if mode == "hook_sync":
        dist.all_reduce(buf)
    else:
        handles.append(dist.all_reduce(buf, async_op=True))
    hook.info = (cs.cuda_stream, torch.cuda.is_current_stream_capturing())


if mode.startswith("hook"):
    w.register_post_accumulate_grad_hook(hook)


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        if mode in ("gather_main", "fn_both"):
            flat = torch.empty(t.numel() * 1, device=dev)
            dist.all_gather_into_tensor(flat, t.detach().reshape(-1).contiguous())
            t = flat.view_as(t) + 0 * t
        return t.clone()

    @staticmethod
    def backward(ctx, g):
        if mode in ("fn_bwd", "fn_both"):
            g = g.contiguous()
            dist.all_reduce(g)
        return g


if mode.startswith("gbn"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mono_vifi_amd.networks import grouped
    nlayers = int(mode[3:] or 1)
    net = torch.nn.Sequential(*[m for _ in range(nlayers) for m in
                                (torch.nn.Conv2d(8, 8, 3, padding=1, bias=False), torch.nn.BatchNorm2d(8), torch.nn.ReLU())]).to(dev)
    net = grouped.convert_grouped_batchnorm(net, sync=True, force_sync=True)
    net.train()
    xin = torch.randn(4, 8, 16, 16, device=dev)


def step():
    buf.zero_()
    if mode.startswith("gbn"):
        for p_ in net.parameters():
            p_.grad = None
        with grouped.grouped(net, 2):
            y = net(xin).sum()
        y.backward()
        return y
    h = x @ w
    if mode in ("gather_main", "fn_bwd", "fn_both"):
        h = _Fn.apply(h)
    y = h.sum()
    y.backward()
    if mode == "main":
        handles.append(dist.all_reduce(buf, async_op=True))
    for h in handles:
        h.wait()
    handles.clear()
    return y


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("eager ok; hook stream/capturing:", getattr(hook, "info", None), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    out = step()
print("captured; hook stream/capturing:", getattr(hook, "info", None), "capture stream", side.cuda_stream, flush=True)
for i in range(5):
    g.replay()
torch.cuda.synchronize()
time.sleep(3.0)          # let the watchdog look at whatever it holds
for i in range(5):
    g.replay()
torch.cuda.synchronize()
print("replayed 10x, grad sum", float(buf.sum()), flush=True)
dist.destroy_process_group()
print("PROBE OK", mode, flush=True)
