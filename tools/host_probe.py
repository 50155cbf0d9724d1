"""Host enqueue time vs GPU time of the hot-path-only step (9 units fwd+bwd)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
torch.cuda.set_device(0)
step = bench.HotPathStep(args, 0, torch.device("cuda", 0))
for _ in range(10):
    step()
torch.cuda.synchronize()
h, g = [], []
for _ in range(30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    h.append(t1 - t0); g.append(t2 - t0)
h.sort(); g.sort()
print(f"host enqueue median {h[15]*1e3:.3f} ms, step (enqueue+drain) median {g[15]*1e3:.3f} ms, cpus {os.cpu_count()}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
