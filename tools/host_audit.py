"""Where does the HOST time of a training step go?  (eager step: the main thread's CPU time per step is close to
the step time, so launch count and Python overhead bound the eager step once the device work shrinks.)

    python tools/host_audit.py [--backbone ResNet18]

Three views of the same step (ResNet18, batch 12, 640x192): (1) wall time of forward / backward / optimiser on the
host with the device idle-waiting excluded (no synchronisation until the end), (2) cProfile of the Python side by
own time, (3) torch.profiler's CPU-side self time per operator and the number of device launches.
Output: gpurun_out/host_audit_<backbone>.txt"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="ResNet18")
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    a = ap.parse_args()
    import mono_vifi_amd as pkg
    pkg.use_shipped_miopen_db()
    import torch
    from torch.profiler import ProfilerActivity, profile
    from mono_vifi_amd.bench_train import TrainStep
    dev = torch.device("cuda", 0)
    args = types.SimpleNamespace(batch=a.batch, height=a.height, width=a.width, backbone=a.backbone)
    step = TrainStep(args, 0, 1, dev)
    tr = step.trainer
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    out = []
    # (1) host-side phases: enqueue time of each phase, device running behind
    ph = {"forward": 0.0, "backward": 0.0, "rest": 0.0}
    n = 5
    t_all0 = time.perf_counter()
    for _ in range(n):
        t0 = time.perf_counter()
        _, losses = tr.process_batch(dict(step.batch))
        t1 = time.perf_counter()
        tr.reducer.zero_grad()
        losses["loss"].backward()
        t2 = time.perf_counter()
        tr.reducer.finish()
        tr._update()
        t3 = time.perf_counter()
        ph["forward"] += t1 - t0
        ph["backward"] += t2 - t1
        ph["rest"] += t3 - t2
    t_enq = time.perf_counter() - t_all0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t_all0
    out.append(f"# {a.backbone} B{a.batch} {a.width}x{a.height}: host enqueue time per step (no sync inside): "
               + ", ".join(f"{k} {v / n * 1e3:.1f} ms" for k, v in ph.items())
               + f"; enqueue total {t_enq / n * 1e3:.1f} ms, with final sync {t_all / n * 1e3:.1f} ms per step")
    # (2) cProfile
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        step()
    pr.disable()
    torch.cuda.synchronize()
    sio = io.StringIO()
    pstats.Stats(pr, stream=sio).sort_stats("tottime").print_stats(45)
    out.append("## cProfile, 3 steps, by own time")
    out.append(sio.getvalue())
    # (3) torch profiler CPU self time
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    out.append("## torch.profiler, 2 steps, by CPU self time")
    out.append(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
    text = "\n".join(out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", f"host_audit_{a.backbone}.txt"), "w").write(text + "\n")
    print(text[:3000])


if __name__ == "__main__":
    main()
