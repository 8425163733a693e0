"""Matrix-pipe utilisation per kernel from two rocprofv3 PMC passes of the bench command:

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -f csv -d /tmp/pm -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE          -f csv -d /tmp/pg -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full
  python tools/pmc_mfma.py <mfma counter_collection.csv> <gui counter_collection.csv> [out.json]

SQ_VALU_MFMA_BUSY_CYCLES counts the cycles the matrix pipes are busy, summed over the chip's 1024 SIMDs (32 per
v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE is the kernel's active cycles summed over the 8 XCDs
(MI355X_MICROARCH.md: units and DVFS sections).  mfma_busy = MFMA cycles / (1024 x GUI_ACTIVE / 8)."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and "gcn::" in r["Kernel_Name"]:
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            acc[name].append(float(r["Counter_Value"]))
    return acc


def main():
    m, g = per_kernel(sys.argv[1], "SQ_VALU_MFMA_BUSY_CYCLES"), per_kernel(sys.argv[2], "GRBM_GUI_ACTIVE")
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/r03_pmc_mfma.json"
    ks = {}
    for name in m:
        if name not in g:
            continue
        mf = sum(m[name]) / len(m[name])
        ga = sum(g[name]) / len(g[name])
        if mf <= 0 or ga <= 0:
            continue
        ks[name] = {"launches": len(m[name]), "SQ_VALU_MFMA_BUSY_CYCLES": round(mf), "GRBM_GUI_ACTIVE": round(ga),
                    "mfma_busy": round(mf / (1024.0 * ga / 8.0), 4)}
    doc = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / --pmc GRBM_GUI_ACTIVE (separate passes, --kernel-trace only) on `python3 bench.py "
           "--steps 2 --warmup 1 --no-cpu-baseline --no-full`, MI355X; means over launches.  mfma_busy = busy cycles of the matrix pipes / "
           "(1024 SIMDs x active cycles per XCD): the fraction of the chip's matrix-pipe cycles the kernel used.")
    json.dump({"_doc": doc, "kernels": ks}, open(out, "w"), indent=1)
    print("wrote", out, len(ks), "kernels")
    for n, r in sorted(ks.items(), key=lambda kv: -kv[1]["mfma_busy"])[:12]:
        print("%-70s %.3f" % (n[:70], r["mfma_busy"]))


if __name__ == "__main__":
    main()
