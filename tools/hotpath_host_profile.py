"""Host cost of the stand-alone hot-path loop (bench.py HotPathStep): wall and CPU time per step, cProfile of the
enqueue path (cumulative, top entries).  The loop is host-bound on the gpurun boxes (0.87 ms per step at batch 4, where
the kernels take 0.3 ms): this shows where the Python time goes.   python tools/hotpath_host_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
sys.argv = ["bench.py", "--workload", "hotpath", "--no-cpu-baseline"]
import torch  # noqa: E402
import bench  # noqa: E402

args = bench.parse()
dev = torch.device("cuda:0")
step = bench.HotPathStep(args, 0, dev)
for _ in range(20):
    step()
torch.cuda.synchronize()
t0, c0 = time.perf_counter(), time.process_time()
for _ in range(steps):
    step()
c1 = time.process_time()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"steps {steps}: wall {1e3 * (t1 - t0) / steps:.4f} ms/step, process CPU (enqueue) {1e3 * (c1 - c0) / steps:.4f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
print(s.getvalue()[:6000])
