#!/bin/bash
# Per-kernel wave residency / VALU share of the eager bench step from SQ counters (GPU box): bash tools/debug/step_occupancy.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/po
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY -f csv -d /tmp/po -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full --no-graph > /tmp/po.log 2>&1
python3 - <<'P'
import csv,glob,collections
cf=glob.glob("/tmp/po/*/*counter_collection.csv")[0]
tf=glob.glob("/tmp/po/*/*kernel_trace.csv")[0]
dur={}
for r in csv.DictReader(open(tf)):
    dur[r["Dispatch_Id"]]=(r["Kernel_Name"], int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.Counter()
seen=set()
for r in csv.DictReader(open(cf)):
    did=r["Dispatch_Id"]
    n=r["Kernel_Name"].split("(")[0][-60:]
    agg[n][r["Counter_Name"]]+=float(r["Counter_Value"])
    if did not in seen:
        seen.add(did); cnt[n]+=1
        if did in dur: agg[n]["ns"]+=dur[did][1]
rows=[]
for n,v in agg.items():
    if v["ns"]<=0: continue
    cyc=v["ns"]*2.4
    rows.append((v["ns"]/1e3, n, cnt[n], v["SQ_WAVE_CYCLES"]*4/cyc/1024, v["SQ_INSTS_VALU"]*4/cyc/1024, v["SQ_WAIT_ANY"]/max(v["SQ_WAVE_CYCLES"],1)))
rows.sort(reverse=True)
print("%-62s %5s %9s %9s %9s %8s"%("kernel","calls","total us","waves/SIMD","VALU util","wait frac"))
for t,n,c,w,u,wa in rows[:45]:
    print("%-62s %5d %9.1f %9.2f %9.2f %8.2f"%(n,c,t,w,u,wa))
P
