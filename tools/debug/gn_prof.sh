cd /tmp && export TMPDIR=/tmp
for b in 128 256 512; do
  rm -rf /tmp/pg$b
  GCN_GN_BLOCKS=$b rocprofv3 --kernel-trace -f csv -d /tmp/pg$b -- python3 $GRAFT_REPO_ROOT/tools/gn_bench.py > /dev/null 2>&1
  echo blocks $b
  python3 - /tmp/pg$b <<'P'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+"/*/*_kernel_trace.csv")[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "gn_" in n or "fold" in n:
        d[(n.split("(")[0][-40:], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"))].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items()):
    v=v[len(v)//4:]
    print("  %-44s grid %-8s n=%3d avg %.1f us" % (k[0],k[1],len(v),sum(v)/len(v)/1e3))
P
done
