"""kernel_stats.csv of `rocprofv3 --kernel-trace --stats -- python bench.py --steps K --warmup W`
-> per-step table (kernel, ms_per_step, launches_per_step), largest first.
usage: python tools/step_breakdown.py kernel_stats.csv TOTAL_STEPS [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
out = []
for r in rows:
    name = r.get("Name") or r.get("Kernel_Name")
    calls = float(r.get("Calls") or r.get("Count") or 0)
    tot = float(r.get("TotalDurationNs") or r.get("Total_Duration_Ns") or 0)
    out.append((tot / 1e6 / steps, calls / steps, name))
out.sort(reverse=True)
print("kernel,ms_per_step,launches_per_step")
for ms, n, name in out[:top]:
    print('"%s",%.3f,%.1f' % (name[:120].replace('"', "'"), ms, n))
print('"TOTAL (all kernels)",%.3f,%.1f' % (sum(o[0] for o in out), sum(o[1] for o in out)))
