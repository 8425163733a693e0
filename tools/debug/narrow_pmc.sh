#!/bin/bash
# PMC counters of gcn_wgrad_narrow at the step's shapes (GPU box): bash tools/debug/narrow_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/pn
  rocprofv3 --kernel-trace --pmc $c -f csv -d /tmp/pn -- python3 $R/tools/wgrad_narrow_bench.py > /tmp/pn.log 2>&1
  python3 - <<'P'
import csv,glob,collections
f=glob.glob("/tmp/pn/*/*counter_collection.csv")
if not f:
    print("no counters"); raise SystemExit
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n=r["Kernel_Name"]
    if "wgrad_narrow_kernel" in n:
        key=(n.split("<")[1].split(">")[0], r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X"))
        d[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()):
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
done
