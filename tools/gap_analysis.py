"""GPU idle time inside the replayed step: from a rocprofv3 --kernel-trace CSV of `bench.py --steps K`, take the last
K * (kernels per step) dispatches on the replay stream and report busy time, wall time and the gaps between consecutive
kernels.  python tools/gap_analysis.py <kernel_trace.csv> <steps> <ms_per_step>"""
import csv
import sys

path, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed region: the window of `steps * ms` with the highest kernel count is the replay loop; find it by scanning
win = int(steps * ms * 1e6)
best, bi, j = 0, 0, 0
for i in range(len(rows)):
    while j < len(rows) and rows[j][0] < rows[i][0] + win:
        j += 1
    if j - i > best:
        best, bi = j - i, i
sel = rows[bi:bi + best]
busy = sum(e - s for s, e, _ in sel)
wall = sel[-1][1] - sel[0][0]
gaps = [max(0, sel[i + 1][0] - sel[i][1]) for i in range(len(sel) - 1)]
overlap = sum(max(0, sel[i][1] - sel[i + 1][0]) for i in range(len(sel) - 1))
print("kernels in window %d (%.1f per step), wall %.3f ms, busy %.3f ms (%.1f %%), gaps %.3f ms, overlap %.3f ms"
      % (len(sel), len(sel) / steps, wall / 1e6, busy / 1e6, 100.0 * busy / wall, sum(gaps) / 1e6, overlap / 1e6))
gs = sorted(gaps)
print("gap median %.2f us, p90 %.2f us, max %.1f us; gaps per step %.3f ms" % (gs[len(gs) // 2] / 1e3, gs[int(len(gs) * 0.9)] / 1e3,
                                                                               gs[-1] / 1e3, sum(gaps) / 1e6 / steps))
short = [(e - s) for s, e, _ in sel if e - s < 10000]
print("kernels shorter than 10 us: %d per step, %.3f ms per step" % (len(short) / steps, sum(short) / 1e6 / steps))
