"""Summarise a rocprofv3 --kernel-trace --stats run: python tools/prof_summary.py <dir> <n_steps_total>"""
import csv
import glob
import sys

d, steps = sys.argv[1], float(sys.argv[2])
f = glob.glob(d + "/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step: %.3f ms" % (tot / 1e6 / steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%-96s n/step=%5.1f  %7.3f ms/step  avg=%8.1f us %5.1f%%" % (
        r["Name"][:96], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps,
        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
