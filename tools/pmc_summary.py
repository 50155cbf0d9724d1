"""Summarise rocprofv3 --pmc output directories: mean counter value per launch, per kernel.
usage: python tools/pmc_summary.py DIR [DIR ...]  -> CSV on stdout"""
import collections
import csv
import glob
import sys

print("pass,kernel,counter,mean_per_launch,launches")
for d in sys.argv[1:]:
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:70], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    tot = collections.defaultdict(float)
    for (k, c), (s, n) in acc.items():
        if c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"):
            tot[k] = max(tot[k], s)
    top = sorted(tot, key=tot.get, reverse=True)[:14] if tot else sorted({k for k, _ in acc})[:14]
    for k in top:
        for (kk, c), (s, n) in sorted(acc.items()):
            if kk == k:
                print('%s,"%s",%s,%.1f,%d' % (d.split("/")[-1], k.replace('"', "'"), c, s / n, n))
