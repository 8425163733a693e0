"""Build profiles/r02_pmc_traffic.json from two rocprofv3 PMC passes of the bench command:

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d /tmp/pf -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d /tmp/pw -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-full
  python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [out.json]

Separate passes, no trace domains besides --kernel-trace (MI355X_MICROARCH.md, HBM section).  FETCH_SIZE / WRITE_SIZE are
in KB; gfx950 counts 16-B/lane streaming reads at half, hence hbm_bytes_corrected = (2*FETCH + WRITE)*1024."""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("gcn::") or \
           (r["Counter_Name"] == counter and "gcn::" in r["Kernel_Name"]):
            name = re.sub(r"^void ", "", r["Kernel_Name"])
            name = name.split("(")[0]
            acc[name].append(float(r["Counter_Value"]))
    return acc


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = sys.argv[3] if len(sys.argv) > 3 else "profiles/r02_pmc_traffic.json"
    ks = {}
    for name in f:
        fe = sum(f[name]) / len(f[name])
        wr = sum(w[name]) / len(w[name]) if name in w else 0.0
        ks[name] = {"launches": len(f[name]), "FETCH_SIZE_KB": round(fe, 1), "WRITE_SIZE_KB": round(wr, 1),
                    "hbm_bytes_corrected": int((2 * fe + wr) * 1024)}
    doc = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace on `python3 bench.py --steps 2 "
           "--warmup 1 --no-cpu-baseline --no-full`, MI355X. Units: KB per launch (mean over launches). hbm_bytes_corrected = "
           "(2*FETCH_SIZE + WRITE_SIZE)*1024 -- gfx950 FETCH_SIZE counts 16-B/lane streaming reads at half "
           "(MI355X_MICROARCH.md, HBM section); kernels that read with narrower accesses are over-corrected by up to 2x "
           "on the read side.")
    json.dump({"_doc": doc, "kernels": ks}, open(out, "w"), indent=1)
    print("wrote", out, len(ks), "kernels")


if __name__ == "__main__":
    main()
