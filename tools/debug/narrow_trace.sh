#!/bin/bash
# kernel-trace durations of gcn_wgrad_narrow's two kernels per shape (GPU box)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pt
rocprofv3 --kernel-trace -f csv -d /tmp/pt -- python3 $R/tools/wgrad_narrow_bench.py > /tmp/pt.log 2>&1
python3 - <<'P'
import csv,glob,collections
f=glob.glob("/tmp/pt/*/*kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "narrow" in n:
        d[(n.split("(")[0][-45:], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("LDS_Block_Size"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items()):
    v=v[5:]
    print(k, "n=%d avg %.1f us min %.1f" % (len(v), sum(v)/len(v)/1e3, min(v)/1e3))
P
