"""Regenerate the measured tables of DESIGN.md from the committed profile files (profiles/r03_*), so that every number
in DESIGN.md is the current one and names the file it came from:

    python tools/design_tables.py            # rewrites the regions between <!-- BEGIN GENERATED:x --> / <!-- END GENERATED:x -->

Sources: r03_bench.json (the driver-style bench line), r03_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same
command: 13 replayed + 3 eager steps, plus the untimed north-star legs), r03_pmc_traffic.json (separate --pmc FETCH_SIZE
/ WRITE_SIZE passes, gfx950 read correction applied), r03_full_workload_kernel_stats.csv, r03_cfg5_kernel_stats.csv,
r03_bench_refshape.json, r03_bench_eager.json, r03_gemm_bench.log, r03_knn_fallback.log, r03_segdiam_bench.log,
r03_full_pmc_traffic.json."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
STEPS = 16            # steps inside the profiled bench command (3 warm-up + 10 timed replays... + 3 eager timing steps)


def load_stats(name):
    rows = list(csv.DictReader(open(os.path.join(P, name))))
    return [(r["Name"].strip('"'), int(r["Calls"]), int(r["TotalDurationNs"]), float(r["AverageNs"])) for r in rows]


def short(n):
    n = n.replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    n = n.replace("gcn::", "")
    n = re.sub(r"at::native::\(anonymous namespace\)::", "at::", n)
    return n[:70]


def group_of(n):
    if "gcn::" in n[:16]:
        return "own (libgcanet_hip.so)"
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"):
        return "hipBLASLt / rocBLAS"
    if "rocclr" in n:
        return "runtime copies / fills"
    if "rocprim" in n:
        return "rocPRIM"
    return "torch (at::native)"


def t_headline():
    d = json.load(open(os.path.join(P, "r03_bench.json")))
    ref = json.load(open(os.path.join(P, "r03_bench_refshape.json")))
    eag = json.load(open(os.path.join(P, "r03_bench_eager.json")))
    r, ns = d["roofline"], d["north_star"]
    L = ["| quantity | value | source |", "|---|---|---|"]
    L.append("| headline: clouds/s fwd+bwd+Adam, B=8/GPU, N=8192, k=64, bf16, 1 GPU | **%.0f clouds/s, %.3f ms/step** (host enqueue %.3f ms, %s) | `profiles/r03_bench.json` |"
             % (d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["step_launch"].split(" (")[0]))
    L.append("| the same step on eager launches (what multi-rank runs use) | %.3f ms/step, host enqueue %.3f ms | `profiles/r03_bench_eager.json` |"
             % (eag["ms_per_step"], eag["host_enqueue_ms_per_step"]))
    L.append("| reference default shape B=3, N=7000, k=80 (`option_new.py:58`) | %.3f ms/step, %.0f clouds/s | `profiles/r03_bench_refshape.json` |"
             % (ref["ms_per_step"], ref["value"]))
    L.append("| kNN Mpts/s (3 searches per step) | %.1f | `r03_bench.json: knn_mpts_per_s` |" % d["knn_mpts_per_s"])
    L.append("| `roofline` (dominant entry point %s) | algorithmic %.1f TF = **%.3f** of the f32 peak; executed (bf16 filter) %.0f TF = %.3f of the bf16 peak; %.3f ms per call; PMC traffic %.0f MB per call | `r03_bench.json: roofline`, `r03_pmc_traffic.json` |"
             % (r["kernel"], r["achieved"], r["frac"], r["executed_tflops"], r["executed_frac"], r["avg_launch_ms"], r["traffic"] / 1e6))
    kg, gm = ns["knn_gather"], ns["grouped_mlp"]
    L.append("| north star: kNN + gather vs HBM (%s) | %.3f + %.3f ms, %.0f GB/s = **%.1f %%** of 8 TB/s (gather alone %.1f %%) | `r03_bench.json: north_star.knn_gather` |"
             % (kg["shape"], kg["knn_ms"], kg["group_ms"], kg["achieved"], 100 * kg["frac"], 100 * kg["group_only_frac"]))
    pm = json.load(open(os.path.join(P, "r03_pmc_mfma.json")))["kernels"] if os.path.exists(os.path.join(P, "r03_pmc_mfma.json")) else {}
    busy = pm.get("gcn::edgeconv_fwd_q_kernel<8, 4, 2, true, true, true, false>", {}).get("mfma_busy")
    L.append("| north star: grouped MLP vs bf16 MFMA (%s) | %.3f + %.3f ms, %.0f TF algorithmic = **%.1f %%** of 2.5 PF; executed %.0f TF = %.1f %% of the 2.4-GHz peak%s | `r03_bench.json: north_star.grouped_mlp`, `r03_pmc_mfma.json` |"
             % (gm["shape"], gm["center_ms"], gm["grouped_ms"], gm["achieved"], 100 * gm["frac"], gm["executed_tflops"], 100 * gm["executed_frac"],
                ("; matrix pipes busy %.1f %% of the active cycles (PMC `SQ_VALU_MFMA_BUSY_CYCLES`)" % (100 * busy)) if busy else ""))
    if pm:
        L.append("| matrix-pipe busy fraction of other kernels (PMC) | %s | `r03_pmc_mfma.json` |"
                 % ", ".join("`%s` %.1f %%" % (k.replace("gcn::", "").replace(", true, true, true, false", ",…"), 100 * v["mfma_busy"])
                             for k, v in sorted(pm.items(), key=lambda kv: -kv[1]["mfma_busy"])[:8]))
    fw, c5, fg, cb = d["full_workload"], d["cfg5_workload"], d["forward_grouping"], d["cpu_baseline"]
    L.append("| literal full `forward_train` + losses + backward + Adam on blob clouds | %.2f ms/step, %.0f clouds/s, %d proposals, %d members | `r03_bench.json: full_workload` |"
             % (fw["ms_per_step"], fw["clouds_per_s"], fw["proposals"], fw["members"]))
    fr = fw.get("roofline")
    if fr and fr.get("achieved"):
        L.append("| its dominant own kernel (`%s`) | %.3f ms per launch, %.0f MB of HBM traffic (PMC) = %.0f GB/s = %.1f %% of 8 TB/s: an L2-resident gather, not an HBM stream | `r03_full_workload_kernel_stats.csv`, `r03_full_pmc_traffic.json` |"
                 % (fr["kernel"].replace("gcn::", ""), fr["avg_launch_ms"], fr["traffic"] / 1e6, fr["achieved"], 100 * fr["frac"]))
    L.append("| BASELINE configs[4], one GPU's share (4 clouds N=16384, C=256 EdgeConv, fp16 attention stacks) | %.2f ms/step, %.0f clouds/s | `r03_bench.json: cfg5_workload` |"
             % (c5["ms_per_step"], c5["clouds_per_s"]))
    L.append("| `forward_grouping`, device vs literal path | %.2f ms vs %.1f ms, %d proposals / %d members, identical: %s | `r03_bench.json: forward_grouping` |"
             % (fg["device_ms"], fg["literal_ms"], fg["proposals"], fg["members"], fg["identical"]))
    L.append("| CPU baseline (oracle port, %d threads) | %.3f clouds/s (%s) | `r03_bench.json: cpu_baseline` |" % (cb["cores"], cb["value"], cb["sample"].split(",")[0]))
    return "\n".join(L)


def t_step_kernels():
    rows = load_stats("r03_kernel_stats.csv")
    groups = {}
    for n, c, t, a in rows:
        g = group_of(n)
        gc, gt = groups.get(g, (0, 0))
        groups[g] = (gc + c, gt + t)
    tot = sum(t for _, _, t, _ in rows)
    L = ["| share of the profiled command (`profiles/r03_kernel_stats.csv`, %d steps + north-star legs) | launches/step | ms/step | %% of kernel time |" % STEPS,
         "|---|---|---|---|"]
    for g, (c, t) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
        L.append("| %s | %.0f | %.3f | %.1f |" % (g, c / STEPS, t / 1e6 / STEPS, 100.0 * t / tot))
    L.append("")
    L.append("| kernel (top 30 by time) | calls/step | avg us | ms/step |")
    L.append("|---|---|---|---|")
    for n, c, t, a in rows[:30]:
        L.append("| `%s` | %.1f | %.1f | %.3f |" % (short(n), c / STEPS, a / 1e3, t / 1e6 / STEPS))
    return "\n".join(L)


def t_traffic():
    d = json.load(open(os.path.join(P, "r03_pmc_traffic.json")))["kernels"]
    want = [("knnf_stream_kernel<4, 1>", "bf16 filter, all pairs", 8 * 8192 * 64 * 2 + 8 * 8192 * 8192 / 8),
            ("knnf_keys_kernel<64>", "exact keys of ~197 candidates per query", None),
            ("knnf_rank_kernel<64>", "ranking + proof", None),
            ("knnn_filter_kernel<1>", "xyz+normal filter, all pairs", 8 * 8192 * 32 + 8 * 8192 * 8192 / 8),
            ("edgeconv_fwd_q_kernel<4, 4, 2, true, true, true", "EdgeConv 64->128 forward", 8 * 8192 * (64 * 2 + 64 * 8 + 128 * 5)),
            ("group_points_lds_kernel<4, 1024>", "grouping_operation C=128 (north star)", 2.198e9),
            ("route_bwd_kernel", "EdgeConv backward routing", None),
            ("rsum_gather_kernel<1", "transposed aggregation", None)]
    L = ["| kernel | what | HBM bytes per launch (PMC, corrected) | algorithmic bytes |", "|---|---|---|---|"]
    for key, what, alg in want:
        for full, rec in d.items():
            if key in full:
                L.append("| `%s` | %s | %.1f MB | %s |" % (full.replace("gcn::", ""), what, rec["hbm_bytes_corrected"] / 1e6, ("%.1f MB" % (alg / 1e6)) if alg else "-"))
                break
    return "\n".join(L)


def t_other(name, steps, top=14):
    rows = load_stats(name)
    tot = sum(t for _, _, t, _ in rows)
    calls = sum(c for _, c, _, _ in rows)
    L = ["`profiles/%s`: %.2f ms of kernels and %.0f launches per step (%d profiled steps)." % (name, tot / 1e6 / steps, calls / steps, steps), "",
         "| kernel | calls/step | avg us | ms/step |", "|---|---|---|---|"]
    for n, c, t, a in rows[:top]:
        L.append("| `%s` | %.1f | %.1f | %.3f |" % (short(n), c / steps, a / 1e3, t / 1e6 / steps))
    return "\n".join(L)


def t_log(name):
    txt = open(os.path.join(P, name)).read().strip()
    return "```\n" + txt + "\n```"


GEN = {
    "headline": t_headline,
    "step_kernels": t_step_kernels,
    "traffic": t_traffic,
    "full_workload": lambda: t_other("r03_full_workload_kernel_stats.csv", 7),
    "cfg5": lambda: t_other("r03_cfg5_kernel_stats.csv", 7, top=10),
    "gemm": lambda: t_log("r03_gemm_bench.log"),
    "knn_cases": lambda: t_log("r03_knn_fallback.log"),
    "segdiam": lambda: t_log("r03_segdiam_bench.log"),
    "grouping": lambda: t_log("r03_grouping_bench.log"),
}


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    for key, fn in GEN.items():
        b, e = "<!-- BEGIN GENERATED:%s -->" % key, "<!-- END GENERATED:%s -->" % key
        if b not in s:
            continue
        i, j = s.index(b) + len(b), s.index(e)
        s = s[:i] + "\n" + fn() + "\n" + s[j:]
    open(path, "w").write(s)
    print("DESIGN.md tables regenerated from profiles/r03_*")


if __name__ == "__main__":
    main()
