"""Per-kernel MFMA / VALU utilisation from a rocprofv3 --pmc run of the training step.
usage: python tools/pmc_mfma_summary.py DIR TOTAL_STEPS
MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8
(the counter is summed over the 8 XCDs); VALU util = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / the same."""
import collections
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        did = r.get("Dispatch_Id")
        if (k, did) not in seen:
            seen.add((k, did))
            calls[k] += 1
rows = []
tot_cyc = tot_mfma = tot_valu = 0.0
for k, c in acc.items():
    cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0:
        continue
    simd_cyc = cyc * 1024.0
    mfma, valu = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0)
    rows.append((cyc, k, mfma / simd_cyc, valu / simd_cyc, c.get("SQ_INSTS_VALU_MFMA_F32", 0.0), calls[k]))
    tot_cyc += cyc
    tot_mfma += mfma
    tot_valu += valu
rows.sort(reverse=True)
print("kernel,share_of_gpu_cycles,mfma_util,valu_util,mfma_f32_insts_per_step,launches_per_step")
for cyc, k, mu, vu, mi, n in rows[:60]:
    print('"%s",%.4f,%.4f,%.4f,%.0f,%.1f' % (k[:110].replace('"', "'"), cyc / tot_cyc, mu, vu, mi / steps, n / steps))
print('"ALL KERNELS (cycle-weighted)",1.0000,%.4f,%.4f,,' % (tot_mfma / (tot_cyc * 1024.0), tot_valu / (tot_cyc * 1024.0)))
